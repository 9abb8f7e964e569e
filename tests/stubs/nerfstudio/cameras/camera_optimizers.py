from dataclasses import dataclass


@dataclass
class CameraOptimizerConfig:
    mode: str = "off"
