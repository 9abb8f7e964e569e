from dataclasses import dataclass


@dataclass
class NerfstudioDataParserConfig:
    load_3D_points: bool = False
