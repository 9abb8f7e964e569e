from dataclasses import dataclass, field
from typing import Any, Dict, Optional


@dataclass
class TrainerConfig:
    method_name: Optional[str] = None
    steps_per_save: int = 1000
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_eval_all_images: int = 25000
    max_num_iterations: int = 1000000
    mixed_precision: bool = False
    pipeline: Any = None
    optimizers: Dict[str, Any] = field(default_factory=dict)
    viewer: Any = None
    vis: str = "wandb"
