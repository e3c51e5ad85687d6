# -*- coding: utf-8 -*-
"""quade_amd -- MI355X-native demultiplexing hot path behind Quade 0.3.2's interface.

The compute path is libquade_hip.so (hand-written HIP for gfx950, C ABI in include/quade_hip.h),
bound with ctypes in quade_amd.hip_backend.  There is no CPU fallback.
"""
__version__ = "0.1.0"
QUADE_VERSION = "Quade 0.3.2"  # reference version string kept by the CLI and the report header
