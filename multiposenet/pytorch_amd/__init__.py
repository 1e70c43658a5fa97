"""multiposenet.pytorch_amd — the poseNet forward/backward hot path of LiMeng95/MultiPoseNet.pytorch,
rebuilt for AMD MI355X (gfx950): hand-written HIP kernels behind a C-ABI (include/mpn.h), driven from
Python through a module that keeps the reference's boundary:

    from multiposenet.pytorch_amd.network.posenet import poseNet      # network/posenet.py:154
    from multiposenet.pytorch_amd.lib.nms.pth_nms import pth_nms      # lib/nms/pth_nms.py:5

Importing this package does not load the HIP library; the first kernel call does, and raises
``MpnError`` if ``libmpn_hip.so`` has not been built (there is no CPU fallback).
"""
import os as _os

# The training step uses three HIP streams at once (main chain, weight-gradient side stream, RCCL).  ROCclr maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; when two of ours share a queue a
# collective parked behind the side stream's event also parks the main chain (measured: 551 vs 622 images/s with
# one-rank RCCL).  Only effective if set before the HIP runtime initialises, hence here at import.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
