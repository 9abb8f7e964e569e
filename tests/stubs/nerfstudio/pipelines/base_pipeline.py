from dataclasses import dataclass
from typing import Any


@dataclass
class VanillaPipelineConfig:
    datamanager: Any = None
    model: Any = None
