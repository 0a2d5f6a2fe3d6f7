"""touch_gs_amd -- MI355X (gfx950) native Gaussian-splatting hot path of Touch-GS.

Python host code on PyTorch-ROCm calling hand-written HIP kernels through the C ABI declared in
``include/tgs.h`` (``touch_gs_amd/lib/libtgs_hip.so``).  There is no CPU fallback: importing the
ops without the built library raises, and every op requires device tensors.
"""
from .camera import Camera  # noqa: F401

__version__ = "0.1.0"
