"""multiposenet.pytorch_amd — the poseNet forward/backward hot path of LiMeng95/MultiPoseNet.pytorch,
rebuilt for AMD MI355X (gfx950): hand-written HIP kernels behind a C-ABI (include/mpn.h), driven from
Python through a module that keeps the reference's boundary:

    from multiposenet.pytorch_amd.network.posenet import poseNet      # network/posenet.py:154
    from multiposenet.pytorch_amd.lib.nms.pth_nms import pth_nms      # lib/nms/pth_nms.py:5

Importing this package does not load the HIP library; the first kernel call does, and raises
``MpnError`` if ``libmpn_hip.so`` has not been built (there is no CPU fallback).
"""
__version__ = "0.1.0"
