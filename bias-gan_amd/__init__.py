"""MI355X-native conv-GAN training step with the Bias-GAN module/loss/comm API.

Host side: Python over PyTorch-ROCm (device memory, streams, torch.distributed);
arithmetic: hand-written gfx950 HIP kernels in libbgamd.so behind the C ABI of
include/bgamd.h.  The sub-package names mirror the reference tree
(src/deepCam/{architecture/gpsro,utils,comm,data}) so call sites port by
changing the import root.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
