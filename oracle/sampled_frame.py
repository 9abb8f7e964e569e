"""The CPU oracle on a SAMPLE of the rays of one full-size frame.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Used by tests/test_gpu_fullsize_parity.py and by bench.py's
`cpu_baseline` leg (which times it and keeps the outputs for the `parity_at_bench_size` record).

At the BASELINE size (1920x1080 rays, 16 x 2^19 x 2 tables) the oracle needs hours for a whole frame, but every ray of
the path is independent of every other one except through two frame-level quantities, and both can be handed over:
  * the counter RNG of the MC-dropout masks and of the Laplace depth draws is keyed by the ray's GLOBAL index
    (`mcdropout_outputs(ray_ids=...)`, `normal_noise(seed, draw, sample index)`), and
  * the Laplace weight samples are drawn per 32,768-ray chunk (laplace_model.py:432-443): a ray takes the set of
    its own chunk (`ws_*` [n_chunks, n, P]).
The one output that does depend on which rays share a chunk is `expected_depth`: DepthRenderer("expected") clips to
the min / max sample position of the CHUNK, which a sample of the chunk's rays cannot reproduce (it almost never
binds; the callers compare that key with a tolerance for it).
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from . import nerf_oracle as O


def ray_runs(total: int, n_runs: int, run: int, skew: int = 7) -> np.ndarray:
    """`n_runs` runs of `run` consecutive rays spread evenly over the `total` rays of a frame (run i starts at
    i * (total // n_runs) + i * skew, so the runs start at different columns and straddle image rows) -> ids [n_runs*run]"""
    starts = np.arange(n_runs, dtype=np.int64) * (total // n_runs) + np.arange(n_runs, dtype=np.int64) * skew
    ids = (starts[:, None] + np.arange(run, dtype=np.int64)[None, :]).reshape(-1)
    assert ids.max() < total
    return ids


def depth_noise_for(ids: np.ndarray, S: int, seed: int, draws: int) -> torch.Tensor:
    """the kernel's built-in depth draws (unerf_laplace_depth_weights without a noise tensor) for rays `ids` -> [D, n, S]"""
    sidx = (ids[:, None] * S + np.arange(S)[None, :]).reshape(-1)
    return torch.from_numpy(np.stack([O.normal_noise(seed, dd, sidx).reshape(len(ids), S) for dd in range(draws)]))


def reference_chunks(method: str, scene: O.NerfScene, origins: torch.Tensor, directions: torch.Tensor, ids: np.ndarray, *,
                     K: int = 0, mc_seed: int = 0, p_drop: float = 0.0, ws_density: Optional[torch.Tensor] = None,
                     ws_rgb: Optional[torch.Tensor] = None, depth_seed: int = 0, depth_draws: int = 100,
                     chunk_rays: int = 1 << 15, step: int = 1024, autocast=None,
                     diagnostics: Optional[dict] = None) -> Iterator[Tuple[np.ndarray, Dict[str, torch.Tensor]]]:
    """Yield (ids_part, outputs_part) for `ids` in pieces of at most `step` rays; origins / directions are the FULL
    frame's flattened rays [H*W, 3].  A piece never crosses a reference chunk (chunk_rays) so that one Laplace sample set
    serves it.  method: "active" | "mcdropout" | "laplace"."""
    ids = np.asarray(ids, dtype=np.int64)
    pos = 0
    while pos < len(ids):
        c = ids[pos] // chunk_rays
        end = pos
        while end < len(ids) and end - pos < step and ids[end] // chunk_rays == c:
            end += 1
        part = ids[pos:end]
        o, d = origins[part], directions[part]
        if method == "active":
            out = O.active_outputs(scene, o, d, autocast=autocast, diagnostics=diagnostics)
        elif method == "mcdropout":
            out = O.mcdropout_outputs(scene, o, d, K, mc_seed, p_drop, ray_ids=part, autocast=autocast, diagnostics=diagnostics)
        elif method == "laplace":
            wd = ws_density if ws_density.dim() == 2 else ws_density[c]
            wr = ws_rgb if ws_rgb.dim() == 2 else ws_rgb[c]
            noise = depth_noise_for(part, scene.num_nerf, depth_seed, depth_draws)
            out = O.laplace_outputs(scene, o, d, wd, wr, noise, autocast=autocast, diagnostics=diagnostics)
        else:
            raise ValueError(method)
        yield part, out
        pos = end


def reference_rays(method: str, scene: O.NerfScene, origins, directions, ids, **kw) -> Dict[str, torch.Tensor]:
    """all of `ids` -> outputs [len(ids), C] (with kw["diagnostics"]: "median_margin" is concatenated in the same order)"""
    lists: Dict[str, list] = {}
    for _, out in reference_chunks(method, scene, origins, directions, ids, **kw):
        for k, v in out.items():
            lists.setdefault(k, []).append(v)
    diag = kw.get("diagnostics")
    if diag is not None and "median_margin" in diag:
        diag["median_margin"] = torch.cat(diag["median_margin"])
    return {k: torch.cat(v) for k, v in lists.items()}
