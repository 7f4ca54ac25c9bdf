"""diffusionmodel_amd — MI355X-native (gfx950) DDPM / ContextUnet hot path behind the class surface of
Shen-Yuuu/DiffusionModel's new_scripy.py: `ContextUnet`, `DDPM`, `ddpm_schedules`, `Cfg`/`Config` and the
block classes, with every tensor op a hand-written HIP kernel reached through the C ABI in
include/dm_amd.h (libdm_amd.so).  There is no CPU / ATen fallback: the package fails loudly without its
shared library or without a HIP device.  The MNIST ancestor lives in `diffusionmodel_amd.mnist`.
"""
from ._lib import DmError, LIB_PATH  # noqa: F401
from .config import Cfg, Config  # noqa: F401
from .ddpm import DDPM, SCHEDULE_KEYS, ddpm_schedules  # noqa: F401
from .modules import (ContextUnet, CoordAttn, EmbedFC, LocalEnhancer, ResConvBlock, ResidualConvBlock, SEBlock,  # noqa: F401
                      UnetDown, UnetUp)
from .optim import DmGradScaler, FusedAdamW  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401

__all__ = ["Cfg", "Config", "ContextUnet", "CoordAttn", "DDPM", "DmError", "DmGradScaler", "EmbedFC", "FusedAdamW", "GraphedTrainStep", "LocalEnhancer",
           "ResConvBlock", "ResidualConvBlock", "SEBlock", "UnetDown", "UnetUp", "ddpm_schedules"]
