"""CPU suite, part 2: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU (no CPU fallback), and its host-side constants agree
with the oracle's.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import capi, synth


def test_library_is_built_in_tree():
    assert os.path.exists(capi.library_path()), "run `python fbus-ekf_amd/build.py` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported():
    lib = capi.load_library()
    names = capi.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_header_cites_the_reference_for_each_group():
    text = open(os.path.join(os.path.dirname(capi.library_path()), "..", "..", "include", "fbus_ekf.h")).read()
    for cite in ("ImuUpdate.m:36", "MeasureUpdate.m:37", "filter.cpp:588-616", "filter.cpp:622-754", "filter.hpp:63-137"):
        assert cite in text


def test_status_strings_and_null_handle_errors():
    lib = capi.load_library()
    assert lib.fbus_status_string(0) == b"ok"
    assert b"no CPU fallback" in lib.fbus_status_string(2)
    assert lib.fbus_ekf_sync(None) == 1
    assert lib.fbus_ekf_predict_dev(None, None, None, None, 0) == 1


def test_create_rejects_bad_arguments_before_touching_the_device():
    lib = capi.load_library()
    prm = capi.default_params(0)
    h = C.c_void_p()
    assert lib.fbus_ekf_create(C.byref(h), C.byref(prm), 0, 0, 32, 18) == 1        # batch <= 0
    assert lib.fbus_ekf_create(C.byref(h), C.byref(prm), 8, 0, 16, 18) == 4        # dtype
    assert lib.fbus_ekf_create(C.byref(h), C.byref(prm), 8, 0, 32, 12) == 4        # nstate
    assert lib.fbus_ekf_create(None, C.byref(prm), 8, 0, 32, 18) == 1


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fbus_ekf import BatchedFilter, FbusError
    with pytest.raises(FbusError) as e:
        BatchedFilter(64)
    assert e.value.code == 2


@pytest.mark.parametrize("dialect", [0, 1])
def test_default_params_agree_with_oracle_defaults(dialect):
    p = capi.default_params(dialect)
    o = oc.Oracle(dialect, 18).prm
    assert list(p.q_diag) == list(o.q_diag)
    assert (p.r_pos, p.r_quat, p.switch_thres) == (o.r_pos, o.r_quat, o.switch_thres)
    assert p.n_markers == o.n_markers == 12
    assert list(p.marker_id)[:12] == list(o.marker_id)[:12]
    R_IL, P_IL, Q_IL = synth.camera_constants(p)
    assert np.abs(R_IL.ravel() - np.array(list(o.R_IL))).max() < 1e-15
    assert np.abs(P_IL - np.array(list(o.P_IL))).max() < 1e-15
    assert np.abs(Q_IL - np.array(list(o.Q_IL))).max() < 1e-15
    ids, mpos, mquat = synth.marker_table(p)
    for k in range(12):
        assert np.abs(mpos[k] - np.array(list(o.marker_pos[k]))).max() == 0
        assert np.abs(mquat[k] - np.array(list(o.marker_quat[k]))).max() < 1e-15
    P0 = oc.Oracle(dialect, 18).P0()
    assert np.allclose(np.diag(P0), np.repeat(np.array(list(p.p0_diag)), 3))


def test_product_package_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(capi.library_path()), "..")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"oracle_capi|ekf_oracle|libfbus_oracle|vision_oracle", text), f


def test_cpp_host_class_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    """include/fbus/batched_filter.hpp is a usable C++ binding of the C ABI (plain g++, no HIP headers)."""
    import subprocess
    import torch
    root = os.path.abspath(os.path.join(os.path.dirname(capi.library_path()), "..", ".."))
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <fbus/batched_filter.hpp>
#include <cstdio>
int main() {
    try {
        fbus::BatchedFilter<float> f(64, FBUS_DIALECT_CPP);
        std::printf("created %d\n", f.batch());
        return 0;
    } catch (const fbus::Error& e) {
        std::printf("error %d: %s\n", e.code, e.what());
        return e.code == FBUS_ERR_NO_DEVICE ? 42 : 1;
    }
}
''')
    exe = tmp_path / "t"
    libdir = os.path.dirname(capi.library_path())
    subprocess.run(["g++", "-std=c++14", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lfbus_ekf", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 42, r.stdout + r.stderr
        assert "no CPU fallback" in r.stdout


def test_no_store_data_hazard_in_the_built_code_objects():
    """gfx950: a VALU write to a data register of a 128-bit store in the very next issue slot reaches memory on lanes 12-15
    of every 16 (measured, round 2).  LLVM pads it only for stores whose soffset is not a register; this scans the ISA
    of every built translation unit for an unpadded instance (tools/check_store_hazard.py)."""
    import glob
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    objs = sorted(glob.glob(os.path.join(root, "fbus-ekf_amd", "lib", "obj", "*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no objects / no llvm-objdump here (the objects do not travel to the GPU box)")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_store_hazard.py")] + objs,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "wide stores, 0 with" in r.stdout
