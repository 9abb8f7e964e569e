"""Host-side helpers mirroring nerfuncertainty/utils.py."""
from __future__ import annotations

from typing import Iterable, Optional

import torch
from torch import nn


def create_mlp(in_dim: int, num_layers: int, layer_width: int, out_dim: int,
               skip_connections: Optional[Iterable[int]] = None, activation=nn.ReLU, out_activation=None,
               dropout_layers: Optional[Iterable[int]] = None, dropout_rate: Optional[float] = None,
               dtype: torch.dtype = torch.float32) -> nn.Sequential:
    """Same topology rule as the reference `create_mlp` (nerfuncertainty/utils.py:6-43), so that
    reference state-dicts (`field.mlp_base.{0,3}.*`, `field.mlp_head.{0,2,5}.*`) load unchanged.

    Layer i (0-based, i < num_layers-1) is `[Dropout if i in dropout_layers] Linear [activation]`;
    a skip index widens that Linear's input by `in_dim`; the output Linear is preceded by Dropout
    when `num_layers-1` or `-1` is listed.  Quirk kept on purpose: with `num_layers == 1` the
    result is a bare Linear and BOTH activations are dropped (utils.py:22-23) -- this is why the
    Laplace `base_mlp` has no ReLU."""
    skips = set(skip_connections or ())
    drops = set(dropout_layers or ())
    mods = []
    if num_layers == 1:
        return nn.Sequential(nn.Linear(in_dim, out_dim, dtype=dtype))
    for i in range(num_layers - 1):
        if i in drops:
            mods.append(nn.Dropout(p=dropout_rate))
        if i == 0:
            if i in skips:
                raise AssertionError("No skip connection for layer 0")
            fan_in = in_dim
        else:
            fan_in = layer_width + (in_dim if i in skips else 0)
        mods.append(nn.Linear(fan_in, layer_width, dtype=dtype))
        if activation:
            mods.append(activation())
    if (num_layers - 1) in drops or -1 in drops:
        mods.append(nn.Dropout(p=dropout_rate))
    mods.append(nn.Linear(layer_width, out_dim, dtype=dtype))
    if out_activation:
        mods.append(out_activation())
    return nn.Sequential(*mods)
