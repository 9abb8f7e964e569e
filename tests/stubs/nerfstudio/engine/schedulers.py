from dataclasses import dataclass
from typing import Optional


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_final: float = 1e-6
    max_steps: int = 100000
    warmup_steps: int = 0
    lr_pre_warmup: float = 1e-8
    ramp: Optional[str] = "cosine"
