"""Reading the reference's on-disk checkpoints (uncertainty-nerf-gs_amd/checkpoints.py): the `step-<9 digits>.ckpt`
discovery rule and the `pipeline` entry of models/ensemble/ensemble_utils.py:36-110, into the Model mirrors."""
import pytest
import torch

from uncertainty_nerf_gs_amd import checkpoints as C
from uncertainty_nerf_gs_amd import plugin


def _small(cfg):
    cfg.log2_hashmap_size = 6
    cfg.proposal_net_args_list = [dict(a, log2_hashmap_size=5) for a in cfg.proposal_net_args_list]
    return cfg


def _write_run(run_dir, model, steps, ddp=False, scale=1.0):
    """a run directory as ns-train leaves it: config.yml + nerfstudio_models/step-*.ckpt with the pipeline state
    dict under `_model.` (and `module.` for a DDP run), next to optimizer state the loader must ignore"""
    ckpts = run_dir / "nerfstudio_models"
    ckpts.mkdir(parents=True)
    (run_dir / "config.yml").write_text("# TrainerConfig\n")
    prefix = "_model.module." if ddp else "_model."
    for s in steps:
        sd = {prefix + k: v.clone() * scale + s for k, v in model.state_dict().items() if v.is_floating_point()}
        sd["datamanager.train_camera_optimizer.pose_adjustment"] = torch.zeros(4, 6)      # not a model key
        torch.save({"step": s, "pipeline": sd, "optimizers": {"fields": {"state": {}}}, "scalers": None}, ckpts / f"step-{s:09d}.ckpt")
    (ckpts / "notes.txt").write_text("not a checkpoint")
    return run_dir / "config.yml"


def test_latest_and_named_steps_load_into_a_model(tmp_path):
    cfg = _small(plugin.MODEL_CONFIGS["active-nerfacto"]())
    src = cfg._target(cfg, num_train_data=3)
    dst = cfg._target(cfg, num_train_data=3)
    config_yml = _write_run(tmp_path / "run0", src, steps=(10, 200, 1500))
    d = C.member_checkpoint_dir(config_yml)
    assert C.checkpoint_steps(d) == [10, 200, 1500]
    path, step = C.checkpoint_path(d)
    assert path.name == "step-000001500.ckpt" and step == 1500
    key = "field.mlp_base_mlp.layers.0.weight"
    path, step = C.load_model(dst, d)
    assert step == 1500 and torch.equal(dst.state_dict()[key], src.state_dict()[key] + 1500)
    path, step = C.load_model(dst, d, load_step=200)
    assert step == 200 and path.name == "step-000000200.ckpt" and torch.equal(dst.state_dict()[key], src.state_dict()[key] + 200)
    with pytest.raises(FileNotFoundError, match="does not exist"):
        C.load_model(dst, d, load_step=7)
    with pytest.raises(FileNotFoundError, match="No checkpoint directory"):
        C.checkpoint_steps(tmp_path / "nowhere")


def test_ensemble_members_come_from_the_directories_next_to_their_configs(tmp_path):
    cfg = _small(plugin.MODEL_CONFIGS["nerfacto-mcdropout"]())
    cfg.mc_samples = 0
    src = cfg._target(cfg, num_train_data=2)
    members = [cfg._target(cfg, num_train_data=2) for _ in range(3)]
    configs = [_write_run(tmp_path / f"member{i}", src, steps=(100 * (i + 1),), ddp=(i == 1), scale=1.0 + i) for i in range(3)]
    loaded = C.load_ensemble(members, configs)
    assert [s for _, s in loaded] == [100, 200, 300]
    key = "field.mlp_head.5.weight"
    for i, m in enumerate(members):       # the DDP member's `module.` prefix is stripped too
        assert torch.equal(m.state_dict()[key], src.state_dict()[key] * (1.0 + i) + 100 * (i + 1)), i
    with pytest.raises(ValueError):
        C.load_ensemble(members, configs[:2])
