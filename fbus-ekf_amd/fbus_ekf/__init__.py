"""fbus_ekf -- host-side Python mirror of the C ABI in include/fbus_ekf.h.

The arithmetic lives in fbus-ekf_amd/lib/libfbus_ekf.so (hand-written HIP for
gfx950).  This package only marshals arguments: it binds the library with
ctypes, mirrors the reference's call contract (`predict` == ImuUpdate,
`correct` == MeasureUpdate, batched) and provides the seeded synthetic
streams and shard arithmetic the bench and tests share.  There is no CPU
fallback: importing works without a GPU (so that symbols can be checked), but
creating a filter raises if the library or a HIP device is missing.
"""
from .capi import (DIALECT_CPP, DIALECT_MATLAB, MODE_NEAREST, MODE_STACKED, COV_SIMPLE, COV_JOSEPH,
                   KERNEL_PREDICT, KERNEL_CORRECT, KERNEL_PREDICT_N, KERNEL_MARKER_POSE, KERNEL_FRAME,
                   VIS_REFRACTIVE, VIS_PINHOLE, VIS_CORNERS3D, POSE_INIT, POSE_RESET,
                   FbusError, FbusParams, default_params, declared_symbols, load_library, library_path)
from .filter import BatchedFilter
from .frame_batcher import FrameBatcher
from . import synth, shard

__all__ = [
    "DIALECT_CPP", "DIALECT_MATLAB", "MODE_NEAREST", "MODE_STACKED", "COV_SIMPLE", "COV_JOSEPH",
    "KERNEL_PREDICT", "KERNEL_CORRECT", "KERNEL_PREDICT_N", "KERNEL_MARKER_POSE", "KERNEL_FRAME",
    "VIS_REFRACTIVE", "VIS_PINHOLE", "VIS_CORNERS3D", "POSE_INIT", "POSE_RESET",
    "FbusError", "FbusParams", "default_params", "declared_symbols", "load_library", "library_path",
    "BatchedFilter", "FrameBatcher", "synth", "shard",
]
