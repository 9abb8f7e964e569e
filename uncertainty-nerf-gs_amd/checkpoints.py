"""Reading the reference's checkpoints from disk: the data format on the input side of the render path.

`ns-train` writes `<run>/nerfstudio_models/step-<9 digits>.ckpt`, a torch pickle `{"step", "pipeline", "optimizers",
"schedulers", ...}` whose `pipeline` entry is the pipeline's state dict (`_model.` prefix, `module.` under DDP).  The
reference finds and loads them in `models/ensemble/ensemble_utils.py:36-110` (`eval_load_ensemble_checkpoints`): the
latest step unless one is named, member i of an ensemble from the `nerfstudio_models` directory next to its
`config.yml`.  Same rules here, for the Model mirrors of `models.py` (whose `load_state_dict` strips the prefixes and
maps the state-dict names) -- no nerfstudio import, no GPU.
"""
from __future__ import annotations

import os
import re
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch

PathLike = Union[str, os.PathLike]
_STEP = re.compile(r"^step-(\d+)\.ckpt$")


def checkpoint_steps(load_dir: PathLike) -> List[int]:
    """steps of the `step-*.ckpt` files in `load_dir`, ascending (ensemble_utils.py:64-66 parses every directory
    entry as one; other files are ignored here instead of raising)"""
    load_dir = Path(load_dir)
    if not load_dir.is_dir():
        raise FileNotFoundError(f"No checkpoint directory found at {load_dir}: checkpoints are generated periodically "
                                "during training")
    return sorted(int(m.group(1)) for m in (_STEP.match(x) for x in os.listdir(load_dir)) if m)


def checkpoint_path(load_dir: PathLike, load_step: Optional[int] = None) -> Tuple[Path, int]:
    """-> (path, step): `load_step` or the latest one (ensemble_utils.py:50-70)"""
    if load_step is None:
        steps = checkpoint_steps(load_dir)
        if not steps:
            raise FileNotFoundError(f"no step-*.ckpt under {load_dir}")
        load_step = steps[-1]
    path = Path(load_dir) / f"step-{load_step:09d}.ckpt"
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint {path} does not exist")
    return path, int(load_step)


def read_pipeline_state(path: PathLike, trust_pickle: Optional[bool] = None) -> Tuple[Dict[str, torch.Tensor], int]:
    """-> (pipeline state dict, step) of one checkpoint file, on the CPU (ensemble_utils.py:71-72).
    Read with torch's safe unpickler (tensors and plain containers only).  The reference calls the unrestricted
    `torch.load(load_path, map_location="cpu")`, which runs whatever a pickle asks for; that is available here only as
    an explicit choice -- `trust_pickle=True` or UNERF_TRUST_CHECKPOINT_PICKLE=1 -- and only after the safe unpickler
    has refused the file's CONTENT (pickle.UnpicklingError: older nerfstudio versions stored config objects next to
    the tensors); I/O errors and corrupt files are reported as they are."""
    import pickle
    if trust_pickle is None:
        trust_pickle = os.environ.get("UNERF_TRUST_CHECKPOINT_PICKLE", "0") == "1"
    try:
        state = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not trust_pickle:
            raise pickle.UnpicklingError(
                f"{path}: holds objects the safe unpickler refuses ({e}); pass trust_pickle=True (or set "
                "UNERF_TRUST_CHECKPOINT_PICKLE=1) to unpickle a checkpoint you trust") from e
        state = torch.load(path, map_location="cpu", weights_only=False)
    if "pipeline" not in state:
        raise KeyError(f"{path}: no 'pipeline' entry (keys: {sorted(state)})")
    return state["pipeline"], int(state.get("step", -1))


def load_model(model, load_dir: PathLike, load_step: Optional[int] = None, strict: bool = False) -> Tuple[Path, int]:
    """Load one Model mirror from a `nerfstudio_models` directory.  -> (path loaded, step).  Raises (RuntimeError from
    the model's load_state_dict, with the path added) when the checkpoint does not cover the model's parameters: a
    run of another method, implementation or key layout must not render from random weights.  The load report
    (models.IncompatibleKeys: ignored keys, tensor counts) is kept as `model.last_load_report`."""
    path, step = checkpoint_path(load_dir, load_step)
    sd, saved_step = read_pipeline_state(path)
    try:
        report = model.load_state_dict(sd, strict=strict)
    except RuntimeError as e:
        raise RuntimeError(f"{path}: {e}") from e
    try:
        model.last_load_report = report
    except Exception:   # a frozen / slotted model object: the report is informational
        pass
    return path, saved_step if saved_step >= 0 else step


def member_checkpoint_dir(config_path: PathLike) -> Path:
    """`<run>/config.yml` -> `<run>/nerfstudio_models` (ensemble_utils.py:78)"""
    return Path(config_path).parent / "nerfstudio_models"


def load_ensemble(models: Sequence, config_paths: Iterable[PathLike], load_step: Optional[int] = None) -> List[Tuple[Path, int]]:
    """Member i from the checkpoint directory next to config_paths[i] (`eval_load_ensemble_checkpoints`,
    ensemble_utils.py:74-108; the reference takes member 0 from `config.load_dir`, which `eval_setup` points at the
    same place).  Under `torchrun` a rank passes only the members it holds, with their config paths."""
    config_paths = list(config_paths)
    if len(config_paths) != len(models):
        raise ValueError(f"{len(models)} models but {len(config_paths)} config paths")
    return [load_model(m, member_checkpoint_dir(c), load_step) for m, c in zip(models, config_paths)]
