from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class RayBundle:
    origins: torch.Tensor
    directions: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None
    camera_indices: Optional[torch.Tensor] = None
    nears: Optional[torch.Tensor] = None
    fars: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self):
        return self.origins.shape[:-1].numel()
