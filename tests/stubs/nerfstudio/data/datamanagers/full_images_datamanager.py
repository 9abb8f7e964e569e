from dataclasses import dataclass
from typing import Any


@dataclass
class FullImageDatamanagerConfig:
    dataparser: Any = None
    cache_images_type: str = "float32"
