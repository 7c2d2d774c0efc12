"""MI355X-native `--com disco` (DiscoNet) collaborative-perception hot path.

Host side (this package) mirrors the reference's Python class surface; all
compute is in libdisconet_hip.so (disconet_amd/csrc, C ABI in
include/disconet_hip.h).  See DESIGN.md.
"""
from .config import Config
from .model import DiscoNet
from .seg import SegDiscoNet, SegModule
from .teacher import TeacherNet
from .train import CoDetModule, TrainEngine

__all__ = ["Config", "DiscoNet", "TeacherNet", "CoDetModule", "TrainEngine", "SegDiscoNet", "SegModule"]
