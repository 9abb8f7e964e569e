from dataclasses import dataclass, field
from typing import Any, Type


@dataclass
class InstantiateConfig:
    """config whose `setup(**kwargs)` builds `_target(self, **kwargs)`"""
    _target: Type = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class ViewerConfig:
    num_rays_per_chunk: int = 32768
