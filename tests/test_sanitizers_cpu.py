"""CPU-side sanitizer runs (no GPU needed; GPU AddressSanitizer is not available on the pool, so the device code is covered by the
parity suite instead):

 * the C oracle built with -fsanitize=address,undefined (`make -C oracle asan`) runs the golden-vector, Jacobian and replay tests of
   tests/test_oracle_cpu.py / test_oracle_jacobians_cpu.py in a child process (gcc's libasan preloaded);
 * the HOST half of libfbus_ekf.so (handle, argument validation, parameter tables, ABI check, RCCL binding: fbus_ekf.hip) built with
   clang's -fsanitize=address,undefined -fno-gpu-sanitize is driven through every entry point that is reachable without a device:
   parameter defaults and validation (marker tables, out-of-range counts and ids), the ABI check, NULL handles and NULL pointers.
"""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _gcc_asan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_address_and_ub_sanitizers():
    rt = _gcc_asan()
    if rt is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               FBUS_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libfbus_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_cpu.py"), os.path.join(ROOT, "tests", "test_oracle_jacobians_cpu.py")],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail


_DRIVER = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "fbus-ekf_amd"))
os.environ["FBUS_EKF_LIB"] = sys.argv[2]
from fbus_ekf import capi
lib = capi.load_library()                      # ABI version and struct size are checked here
assert lib.fbus_ekf_abi_version() == capi.ABI_VERSION and lib.fbus_params_size() == C.sizeof(capi.FbusParams)
msg = C.create_string_buffer(256)
for dialect in (0, 1):
    p = capi.default_params(dialect)
    assert lib.fbus_params_validate(C.byref(p), msg, 256) == 0, msg.value
    assert p.n_markers == 12
# every way the host-side tables can be mis-sized
p = capi.default_params(0); p.n_markers = capi.MAX_MARKERS + 1
assert lib.fbus_params_validate(C.byref(p), msg, 256) == 1 and b"n_markers" in msg.value
p = capi.default_params(0); p.n_markers = -1
assert lib.fbus_params_validate(C.byref(p), msg, 256) == 1
p = capi.default_params(0); p.marker_id[3] = 5000
assert lib.fbus_params_validate(C.byref(p), msg, 256) == 1 and b"marker id" in msg.value
p = capi.default_params(0); p.marker_id[0] = -7
assert lib.fbus_params_validate(C.byref(p), msg, 256) == 1
p = capi.default_params(0); p.dialect = 9
assert lib.fbus_params_validate(C.byref(p), msg, 4) == 1 and len(msg.value) <= 3          # truncated message, still terminated
p = capi.default_params(0); p.r_pos = 0.0
assert lib.fbus_params_validate(C.byref(p), None, 0) == 1
assert lib.fbus_params_validate(None, msg, 256) == 1
p = capi.default_params(0); p.n_markers = capi.MAX_MARKERS                                 # the full table is legal
for k in range(capi.MAX_MARKERS):
    p.marker_id[k] = 1023 - k
    for i in range(9):
        p.marker_rot[k][i] = p.marker_rot[0][i]
assert lib.fbus_params_validate(C.byref(p), msg, 256) == 0, msg.value
# the ABI check refuses a caller built against another header BEFORE it reads the struct: a 16-byte buffer is enough to prove it
h = C.c_void_p()
small = C.create_string_buffer(16)
rc = lib.fbus_ekf_create_checked(C.byref(h), C.cast(small, C.POINTER(capi.FbusParams)), 16, capi.ABI_VERSION, 64, 0, 32, 18)
assert rc == capi.ERR_ABI and not h.value
p = capi.default_params(0)
assert lib.fbus_ekf_create_checked(C.byref(h), C.byref(p), C.sizeof(p), capi.ABI_VERSION - 1, 64, 0, 32, 18) == capi.ERR_ABI
# argument checks in front of the device query, then the loud no-device failure (this box has no GPU)
assert lib.fbus_ekf_create_checked(C.byref(h), C.byref(p), C.sizeof(p), capi.ABI_VERSION, 0, 0, 32, 18) == 1
assert lib.fbus_ekf_create_checked(C.byref(h), C.byref(p), C.sizeof(p), capi.ABI_VERSION, 64, 0, 16, 18) == 4
assert lib.fbus_ekf_create_checked(C.byref(h), C.byref(p), C.sizeof(p), capi.ABI_VERSION, 64, 0, 32, 17) == 4
assert lib.fbus_ekf_create_checked(None, C.byref(p), C.sizeof(p), capi.ABI_VERSION, 64, 0, 32, 18) == 1
rc = lib.fbus_ekf_create_checked(C.byref(h), C.byref(p), C.sizeof(p), capi.ABI_VERSION, 64, 0, 32, 18)
import torch
if not torch.cuda.is_available():
    assert rc == 2 and not h.value, rc
# NULL handles are refused everywhere (status 1), never dereferenced
null = C.c_void_p()
assert lib.fbus_ekf_predict_dev(null, None, None, None, 0) == 1
assert lib.fbus_ekf_correct_dev(null, 4, None, None, None, 0, None) == 1
assert lib.fbus_ekf_set_team(null, 0, 0) == 1 and lib.fbus_ekf_gather(null, None, None) == 1
assert lib.fbus_ekf_comm_unique_id(None) == 1 and lib.fbus_ekf_comm_destroy(null) == 1
assert lib.fbus_ekf_records(null, None, None, None) == 1 and lib.fbus_ekf_sync(null) == 1
assert lib.fbus_ekf_predict_async(null, None, None, None, 0) == 1 and lib.fbus_ekf_predict_n_async(null, 3, None, None, None, 0) == 1
assert lib.fbus_ekf_correct_async(null, 4, None, None, None, 0, None) == 1 and lib.fbus_ekf_correct_pixels_async(null, 4, None, None, None, None) == 1
assert lib.fbus_ekf_async_inputs_consumed(null) == 1 and lib.fbus_ekf_async_stats(null, None, None, None) == 1
assert lib.fbus_ekf_host_register(None, 0) == 1 and lib.fbus_ekf_host_unregister(None) == 1
assert lib.fbus_ekf_destroy(null) == 0
assert lib.fbus_status_string(6).decode().startswith("caller and library")
print("host-asan driver ok")
'''


def test_library_host_code_under_address_and_ub_sanitizers(tmp_path):
    spec = importlib.util.spec_from_file_location("fbus_build", os.path.join(ROOT, "fbus-ekf_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rt = b.asan_runtime()
    if rt is None:
        pytest.skip("clang's shared ASan runtime not found under /opt/rocm")
    lib = b.build_host_asan(str(tmp_path / "libfbus_ekf_asan.so"))
    drv = tmp_path / "drv.py"
    drv.write_text(_DRIVER)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(drv), ROOT, lib], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "host-asan driver ok" in r.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
