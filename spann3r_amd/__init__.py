"""spann3r_amd: MI355X-native (gfx950) implementation of Spann3R's per-frame forward hot path.

Public surface mirrors /root/reference/spann3r/model.py: `Spann3R`, `SpatialMemory`.
Compute runs in hand-written HIP kernels behind a C-ABI library (include/spann3r_hip.h);
PyTorch only provides device memory, streams and torch.distributed.
"""
from .config import Spann3RConfig, FULL, TINY  # noqa: F401

__all__ = ["Spann3RConfig", "FULL", "TINY", "Spann3R", "SpatialMemory"]


def __getattr__(name):
    if name in ("Spann3R", "SpatialMemory"):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
