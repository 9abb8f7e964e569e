"""The N > 1 path on real GPUs: one process per GPU over NCCL (= RCCL on ROCm), the HIP moments kernel as the
reduction -- `ensemble.aggregate_distributed` against the reference-generated ensemble fixture.  Needs >= 2 GPUs
(the 1-GPU test box skips it; the gloo twin in test_distributed_cpu.py always runs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _member(g, tag, i):
    return {k[len(f"{tag}_in{i}_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_in{i}_")}


def _worker(rank, world, port, tag, ret):
    import torch.distributed as dist
    from uncertainty_nerf_gs_amd import ensemble, lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib.build_library()
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g = golden("ensemble.npz")
    member = {k: v.to(dev) for k, v in _member(g, tag, rank).items()}
    out = ensemble.aggregate_distributed(member)            # default moments_fn: the HIP kernel
    torch.cuda.synchronize()
    ret[rank] = {k: v.cpu().numpy() for k, v in out.items()}
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["plain", "alea"])
def test_one_member_per_gpu_over_rccl_matches_the_reference_fixture(tag):
    if torch.cuda.device_count() < 2:          # device_count() does not initialise the GPU in this process
        pytest.skip("needs >= 2 GPUs (one ensemble member per GPU)")
    from uncertainty_nerf_gs_amd import ensemble
    world = 2
    g = golden("ensemble.npz")
    single = ensemble.aggregate([_member(g, tag, i) for i in range(world)], moments_fn=lambda x: (x.mean(0), x.var(0)))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), tag, ret), nprocs=world, join=True)   # fresh processes, one per GPU
        for r in range(world):
            assert set(ret[r]) == set(single)
            for k, v in single.items():
                np.testing.assert_allclose(ret[r][k], v.numpy(), rtol=1e-6, atol=1e-7, err_msg=f"rank {r}: {k}")
            for k in single:
                assert np.array_equal(ret[r][k], ret[0][k]), f"ranks disagree on {k}"


def _solo_worker(rank, world, port, tag, ret):
    """world_size 1 over NCCL: all members on the one GPU, but the same collectives (all_to_all_single, all_gather)
    and the HIP moments kernel as in the multi-GPU run"""
    import torch.distributed as dist
    from uncertainty_nerf_gs_amd import ensemble, lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib.build_library()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = golden("ensemble.npz")
    members = [{k: v.to(dev) for k, v in _member(g, tag, i).items()} for i in range(5)]
    out = ensemble.aggregate_distributed(members)
    torch.cuda.synchronize()
    ret[0] = {k: v.cpu().numpy() for k, v in out.items()}
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["plain", "alea"])
def test_rccl_collectives_with_hip_moments_on_one_gpu_match_the_reference_fixture(tag):
    """The 1-GPU box cannot run one member per GPU, but it can run the same code path: an NCCL (= RCCL) process group
    of size 1, the all_to_all_single / all_gather calls of aggregate_distributed on device tensors, the HIP moments
    kernel -- 5 members against the REFERENCE's EnsemblePipeline output recorded in tests/golden/ensemble.npz."""
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    g = golden("ensemble.npz")
    expect = {k[len(f"{tag}_out_"):]: g[k] for k in g.files if k.startswith(f"{tag}_out_")}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_solo_worker, args=(1, _free_port(), tag, ret), nprocs=1, join=True)
        assert set(ret[0]) == set(expect)
        for k, v in expect.items():
            np.testing.assert_allclose(ret[0][k], v, rtol=1e-6, atol=1e-7, err_msg=k)
