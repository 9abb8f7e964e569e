from dataclasses import dataclass
from typing import Any


@dataclass
class VanillaDataManagerConfig:
    dataparser: Any = None
    train_num_rays_per_batch: int = 1024
    eval_num_rays_per_batch: int = 1024
