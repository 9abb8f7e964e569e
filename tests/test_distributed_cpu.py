"""The N>1 path on CPU: world_size=2 over gloo.  The collective plumbing of
ensemble.aggregate_distributed is exercised with a torch moments function injected in place of
the HIP kernel (the kernel itself is covered by the GPU tests); results must equal the
reference-pinned single-process aggregation bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import golden


def _torch_moments(x):
    return x.mean(dim=0), x.var(dim=0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _member(g, tag, i):
    return {k[len(f"{tag}_in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_in{i}_")}


class _FakeMember:
    def __init__(self, outputs):
        self.outputs = outputs

    def get_outputs_for_camera(self, camera):
        assert camera == "camera"
        return self.outputs


def test_ensemble_pipeline_single_process_equals_aggregate():
    from uncertainty_nerf_gs_amd import ensemble
    g = golden("ensemble.npz")
    members = [_member(g, "alea", i) for i in range(4)]
    pipe = ensemble.EnsemblePipeline([_FakeMember(m) for m in members], moments_fn=_torch_moments)
    assert pipe.model is pipe.models[0]
    out = pipe.get_ensemble_outputs_for_camera_ray_bundle("camera")
    ref = ensemble.aggregate(members, moments_fn=_torch_moments)
    assert list(out) == list(ref) and all(torch.equal(out[k], ref[k]) for k in ref)
    with pytest.raises(AssertionError, match="at least two"):
        ensemble.EnsemblePipeline([_FakeMember(members[0])], moments_fn=_torch_moments) \
            .get_ensemble_outputs_for_camera_ray_bundle("camera")


def _worker(rank, world, port, tag, ret, per_rank=1):
    import torch.distributed as dist
    from uncertainty_nerf_gs_amd import ensemble
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = golden("ensemble.npz")
    if per_rank == 1:
        member = _member(g, tag, rank)
    else:  # rank-major member order: rank r holds members r*per_rank .. r*per_rank+per_rank-1
        member = [_member(g, tag, rank * per_rank + i) for i in range(per_rank)]
    if per_rank == 1:
        stages = {}      # bench.py's per-stage record: on CPU tensors no device times, but the byte counts of the exchange
        out = ensemble.aggregate_distributed(member, moments_fn=_torch_moments, stage_ms=stages)
        P, C = member["rgb"].shape[0] * member["rgb"].shape[1], stages["packed_image_bytes_per_member"] // (member["rgb"].shape[0] * member["rgb"].shape[1])
        assert stages["packed_image_bytes_per_member"] == P * C and C % 4 == 0
        # every rank sends (world - 1) / world of its packed image and receives world - 1 reduced (mean, var) slices
        assert abs(stages["bytes_all_to_all_sent_per_rank"] - P * C * (world - 1) // world) <= C
        assert stages["bytes_all_gather_received_per_rank"] == (world - 1) * 2 * C * (-(-P // world))
    else:  # the same through the EnsemblePipeline surface (fake members that return the golden images)
        pipe = ensemble.EnsemblePipeline([_FakeMember(m) for m in member], moments_fn=_torch_moments)
        out = pipe.get_ensemble_outputs_for_camera_ray_bundle("camera")
    ret[rank] = {k: v.numpy() for k, v in out.items()}
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["plain", "alea"])
def test_one_member_per_rank_matches_single_process(tag):
    from uncertainty_nerf_gs_amd import ensemble
    world = 2
    g = golden("ensemble.npz")
    members = [{k[len(f"{tag}_in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_in{i}_")}
               for i in range(world)]
    single = ensemble.aggregate(members, moments_fn=_torch_moments)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), tag, ret), nprocs=world, join=True)
        assert set(ret.keys()) == {0, 1}
        for r in range(world):
            assert set(ret[r]) == set(single)
            for k, v in single.items():
                assert np.array_equal(ret[r][k], v.numpy()), (r, k)


def _splat_worker(rank, world, port, ret):
    import torch.distributed as dist
    from uncertainty_nerf_gs_amd import ensemble
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = ensemble.aggregate_distributed(_splat_members()[rank], moments_fn=_torch_moments)
    ret[rank] = {k: v.numpy() for k, v in out.items()}
    dist.destroy_process_group()


def _splat_members(M=3, H=5, W=7):
    """active-splatfacto member dicts (activesplatfacto_model.py:359-367): images plus the [3] `background`"""
    g = torch.Generator().manual_seed(4)
    out = []
    for _ in range(M):
        unc = torch.rand(H, W, 1, generator=g)
        dv = torch.rand(H, W, 1, generator=g)
        out.append({"rgb": torch.rand(H, W, 3, generator=g), "depth": torch.rand(H, W, 1, generator=g) * 4,
                    "accumulation": torch.rand(H, W, 1, generator=g), "background": torch.tensor([0.1490, 0.1647, 0.2157]),
                    "uncertainty": unc, "rgb_var": unc ** 2, "rgb_std": unc, "depth_var": dv, "depth_std": dv.sqrt()})
    return out


def test_splat_member_ensemble_with_a_non_image_key():
    """EnsemblePipelineSplatfacto (ensemble_pipeline.py:210-300) inherits the aggregation: every key is stacked and
    averaged, `background` [3] included; 3 ranks x 35 pixels also exercises unequal pixel slices (12 / 12 / 11)."""
    from uncertainty_nerf_gs_amd import ensemble
    from oracle import nerf_oracle as O
    members = _splat_members()
    single = ensemble.aggregate(members, moments_fn=_torch_moments)
    ref = O.ensemble_aggregate(members)                 # the reference-pinned restatement of :159-189
    assert set(single) == set(ref)
    for k in ref:
        torch.testing.assert_close(single[k], ref[k], rtol=1e-6, atol=1e-7, msg=k)
    assert torch.allclose(single["background"], members[0]["background"])
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_splat_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
        for r in range(3):
            assert set(ret[r]) == set(single)
            for k, v in single.items():
                assert np.array_equal(ret[r][k], v.numpy()), (r, k)


def test_two_members_per_rank_matches_single_process():
    """M = 4 members over 2 ranks (the M > N case of bench.py --method ensemble)."""
    from uncertainty_nerf_gs_amd import ensemble
    g = golden("ensemble.npz")
    single = ensemble.aggregate([_member(g, "alea", i) for i in range(4)], moments_fn=_torch_moments)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), "alea", ret, 2), nprocs=2, join=True)
        for r in range(2):
            for k, v in single.items():
                assert np.array_equal(ret[r][k], v.numpy()), (r, k)


@pytest.mark.parametrize("tag", ["plain", "alea"])
def test_single_process_aggregate_matches_reference_golden(tag):
    """M=5 members on one device == the reference EnsemblePipeline output recorded in the fixture."""
    from uncertainty_nerf_gs_amd import ensemble
    g = golden("ensemble.npz")
    members = [{k[len(f"{tag}_in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_in{i}_")}
               for i in range(5)]
    out = ensemble.aggregate(members, moments_fn=_torch_moments)
    expect = {k[len(f"{tag}_out_"):]: g[k] for k in g.files if k.startswith(f"{tag}_out_")}
    assert set(out) == set(expect)
    for k, v in expect.items():
        np.testing.assert_allclose(out[k].numpy(), v, rtol=1e-6, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("P,world", [(2073600, 8), (40000, 8), (7, 3), (5, 8), (0, 2)])
def test_pixel_slices_tile_the_image_exactly(P, world):
    from uncertainty_nerf_gs_amd.ensemble import pixel_slice
    edges = [pixel_slice(P, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == P
    for (a0, b0), (a1, b1) in zip(edges, edges[1:]):
        assert b0 == a1 and b0 >= a0
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1


def test_views_for_rank_partition():
    from uncertainty_nerf_gs_amd.ensemble import views_for_rank
    all_views = sorted(v for r in range(8) for v in views_for_rank(24, r, 8))
    assert all_views == list(range(24))
