from dataclasses import dataclass
from typing import Any


@dataclass
class MethodSpecification:
    config: Any
    description: str = ""
