"""ctypes binding of include/fbus_ekf.h (no torch, no numpy-side arithmetic)."""
import ctypes as C
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                       # fbus-ekf_amd/
_HEADER = os.path.join(os.path.dirname(_ROOT), "include", "fbus_ekf.h")

DIALECT_MATLAB, DIALECT_CPP = 0, 1
MODE_NEAREST, MODE_STACKED = 0, 1
COV_SIMPLE, COV_JOSEPH = 0, 1
KERNEL_PREDICT, KERNEL_CORRECT, KERNEL_PREDICT_N, KERNEL_MARKER_POSE, KERNEL_FRAME = 0, 1, 2, 3, 4
KERNEL_CORRECT_CORNERS = 5
VIS_REFRACTIVE, VIS_PINHOLE, VIS_CORNERS3D = 0, 1, 2
MEAS_PIXELS, MEAS_CORNERS = 0, 1   # fbus_ekf_frame_meas_fused_dev
POSE_INIT, POSE_RESET = 0, 1
L0_QUAT_MUL, L0_QUAT_TO_ROTMAT_M, L0_QUAT_TO_ROTMAT_E, L0_QUAT_NORMALIZE, L0_EXPM_SO3_NEG, L0_DTHETA_TO_QUAT, L0_SINCOS_HALF = range(7)
MAX_MARKERS, MAX_VISIBLE = 32, 16
MAX_WINDOW_FRAMES = 64           # fbus_ekf_frames_fused_dev
STREAM_OWN = (1 << 64) - 1         # FBUS_STREAM_OWN = (void*)-1
# fbus_ekf_launch_info (include/fbus_ekf.h)
(INFO_SIMDS, INFO_ONE_ROUND_FILTERS, INFO_TWO_WAVE_MIN_B, INFO_BIG_RECORDS_MB, INFO_MALL_MB, INFO_L2_KB, INFO_POLICY_BATCH,
 INFO_ROLES_PREDICT, INFO_ROLES_MEAS, INFO_TEAM_FRAMES, INFO_MEAS_SPLIT) = range(11)
ABI_VERSION = 6                    # FBUS_ABI_VERSION of the header this mirror was written against
ERR_ABI = 6


class FbusError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: status {code}" + (f" ({detail})" if detail else ""))


class FbusParams(C.Structure):
    _fields_ = [
        ("dialect", C.c_int32), ("cov_form", C.c_int32),
        ("q_diag", C.c_double * 4), ("r_pos", C.c_double), ("r_quat", C.c_double),
        ("p0_diag", C.c_double * 6),
        ("T_SC_left", C.c_double * 16), ("T_SC_right", C.c_double * 16),
        ("n_markers", C.c_int32), ("marker_id", C.c_int32 * MAX_MARKERS),
        ("marker_pos", (C.c_double * 3) * MAX_MARKERS), ("marker_rot", (C.c_double * 9) * MAX_MARKERS),
        ("switch_thres", C.c_double), ("max_dist", C.c_double),
        ("n_air", C.c_double), ("n_glass", C.c_double), ("n_water", C.c_double),
        ("d_air", C.c_double), ("d_glass", C.c_double), ("port_normal", C.c_double * 3),
        ("marker_size", C.c_double), ("r_pix", C.c_double),
    ]


def library_path():
    """in-tree build; FBUS_EKF_LIB overrides it (A/B runs of differently built kernels)"""
    return os.environ.get("FBUS_EKF_LIB") or os.path.join(_ROOT, "lib", "libfbus_ekf.so")


def declared_symbols():
    """Every function the public header declares (used by the export test)."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fbus_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library():
    """Loads the HIP library; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FbusError(-1, "load_library",
                        f"{path} is missing -- build it with `python fbus-ekf_amd/build.py` "
                        "(there is no CPU fallback)")
    # A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64.so.7; if this
    # library pulled in /opt/rocm's copy first, a later `import torch` would end up with a second
    # runtime and device discovery fails.  Loading torch first makes both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, ip, u8p = C.c_void_p, C.c_void_p, C.c_void_p
    H = C.c_void_p
    sig = {
        "fbus_params_default": ([C.POINTER(FbusParams), C.c_int], C.c_int),
        "fbus_params_validate": ([C.POINTER(FbusParams), C.c_char_p, C.c_size_t], C.c_int),
        "fbus_ekf_abi_version": ([], C.c_int),
        "fbus_params_size": ([], C.c_size_t),
        "fbus_ekf_create_checked": ([C.POINTER(H), C.POINTER(FbusParams), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
        "fbus_ekf_create": ([C.POINTER(H), C.POINTER(FbusParams), C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
        "fbus_ekf_destroy": ([H], C.c_int),
        "fbus_ekf_set_stream": ([H, vp], C.c_int),
        "fbus_ekf_set_team": ([H, C.c_int, C.c_int], C.c_int),
        "fbus_ekf_set_policy_batch": ([H, C.c_int], C.c_int),
        "fbus_ekf_launch_info": ([H, C.c_int, C.c_int, C.POINTER(C.c_int)], C.c_int),
        "fbus_ekf_wait_stream": ([H, vp], C.c_int),
        "fbus_ekf_signal_stream": ([H, vp], C.c_int),
        "fbus_ekf_sync": ([H], C.c_int),
        "fbus_ekf_last_error": ([H], C.c_char_p),
        "fbus_status_string": ([C.c_int], C.c_char_p),
        "fbus_ekf_set_state": ([H, vp, vp, vp, ip], C.c_int),
        "fbus_ekf_get_state": ([H, vp, vp, vp, ip], C.c_int),
        "fbus_ekf_set_state_dev": ([H, vp, vp, vp, ip], C.c_int),
        "fbus_ekf_get_state_dev": ([H, vp, vp, vp, ip], C.c_int),
        "fbus_ekf_reset_cov": ([H], C.c_int),
        "fbus_ekf_records": ([H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)], C.c_int),
        "fbus_ekf_attach_records": ([H, vp, C.c_size_t], C.c_int),
        "fbus_ekf_comm_unique_id": ([vp], C.c_int),
        "fbus_ekf_comm_init": ([H, vp, C.c_int, C.c_int], C.c_int),
        "fbus_ekf_comm_attach": ([H, vp, C.c_int, C.c_int], C.c_int),
        "fbus_ekf_comm_destroy": ([H], C.c_int),
        "fbus_ekf_gather": ([H, vp, C.POINTER(C.c_size_t)], C.c_int),
        "fbus_ekf_comm_init_all": ([C.POINTER(H), C.c_int], C.c_int),
        "fbus_ekf_gather_group": ([C.POINTER(H), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)], C.c_int),
        "fbus_ekf_copy_records": ([H, vp, C.c_int], C.c_int),
        "fbus_ekf_predict": ([H, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_predict_dev": ([H, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_predict_n": ([H, C.c_int, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_predict_n_dev": ([H, C.c_int, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_correct": ([H, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_correct_dev": ([H, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_correct_corners": ([H, C.c_int, ip, vp, vp, C.c_int, C.c_int, u8p], C.c_int),
        "fbus_ekf_correct_corners_dev": ([H, C.c_int, ip, vp, vp, C.c_int, C.c_int, u8p], C.c_int),
        "fbus_ekf_correct_pixels": ([H, C.c_int, ip, vp, vp, u8p], C.c_int),
        "fbus_ekf_correct_pixels_dev": ([H, C.c_int, ip, vp, vp, u8p], C.c_int),
        "fbus_ekf_predict_async": ([H, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_predict_n_async": ([H, C.c_int, vp, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_correct_async": ([H, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_correct_pixels_async": ([H, C.c_int, ip, vp, vp, u8p], C.c_int),
        "fbus_ekf_async_inputs_consumed": ([H], C.c_int),
        "fbus_ekf_async_stats": ([H, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
        "fbus_ekf_host_register": ([vp, C.c_size_t], C.c_int),
        "fbus_ekf_host_unregister": ([vp], C.c_int),
        "fbus_ekf_get_applied": ([H, u8p], C.c_int),
        "fbus_ekf_frame_dev": ([H, C.c_int, vp, vp, vp, C.c_int, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_frame_fused_dev": ([H, C.c_int, vp, vp, vp, C.c_int, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_frame_meas_fused_dev": ([H, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, ip, vp, vp, C.c_int, C.c_int, u8p], C.c_int),
        "fbus_ekf_frames_meas_fused_dev": ([H, C.c_int, ip, vp, vp, vp, C.c_int, C.c_int, C.c_int, ip, vp, vp, C.c_int, C.c_int, u8p], C.c_int),
        "fbus_ekf_frames_fused_dev": ([H, C.c_int, ip, vp, vp, vp, C.c_int, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_init_gravity_bias": ([H, C.c_int, vp, vp], C.c_int),
        "fbus_ekf_init_gravity_bias_dev": ([H, C.c_int, vp, vp], C.c_int),
        "fbus_ekf_pose_init": ([H, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_pose_init_dev": ([H, C.c_int, ip, vp, vp, C.c_int, u8p], C.c_int),
        "fbus_ekf_vision_only_pose": ([H, C.c_int, ip, vp, vp, vp], C.c_int),
        "fbus_ekf_vision_only_pose_dev": ([H, C.c_int, ip, vp, vp, vp], C.c_int),
        "fbus_ekf_imu_ema": ([H, C.c_int, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_imu_ema_dev": ([H, C.c_int, vp, vp, C.c_int], C.c_int),
        "fbus_ekf_marker_pose": ([H, C.c_int, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "fbus_ekf_marker_pose_dev": ([H, C.c_int, C.c_int, vp, vp, vp, vp, vp], C.c_int),
        "fbus_ekf_graph_begin": ([H], C.c_int),
        "fbus_ekf_graph_end": ([H, C.POINTER(C.c_int)], C.c_int),
        "fbus_ekf_graph_launch": ([H, C.c_int], C.c_int),
        "fbus_ekf_graph_destroy": ([H, C.c_int], C.c_int),
        "fbus_ekf_l0_eval": ([H, C.c_int, C.c_int, vp, vp, vp], C.c_int),
        "fbus_ekf_timing_enable": ([H, C.c_int], C.c_int),
        "fbus_ekf_timing_reset": ([H], C.c_int),
        "fbus_ekf_timing_read": ([H, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    # a binding that cannot use the header's create macro checks the ABI once after loading (include/fbus_ekf.h)
    if lib.fbus_ekf_abi_version() != ABI_VERSION or lib.fbus_params_size() != C.sizeof(FbusParams):
        raise FbusError(ERR_ABI, "load_library",
                        f"{path}: library ABI {lib.fbus_ekf_abi_version()} / fbus_params {lib.fbus_params_size()} B, "
                        f"this mirror expects ABI {ABI_VERSION} / {C.sizeof(FbusParams)} B -- rebuild the library")
    _lib = lib
    return lib


def default_params(dialect=DIALECT_MATLAB):
    lib = load_library()
    p = FbusParams()
    rc = lib.fbus_params_default(C.byref(p), dialect)
    if rc != 0:
        raise FbusError(rc, "fbus_params_default")
    return p
