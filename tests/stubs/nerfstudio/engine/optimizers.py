from dataclasses import dataclass


@dataclass
class AdamOptimizerConfig:
    lr: float = 5e-4
    eps: float = 1e-8
    weight_decay: float = 0.0
