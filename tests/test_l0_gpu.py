"""GPU suite, row a3: the L0 helpers one by one.  The device inline functions the kernels are built from
(`ekf_device.hpp`: quat_mul, quat_to_rotmat_m / _e, quat_normalize, the closed-form expm(-[w]x dt) of predict_nominal, the
delta-theta -> quaternion of the injection, the polynomial / library sin-cos switch), evaluated through
`fbus_ekf_l0_eval`, against the oracle's restatement of matlab/quaternion_*.m, axisangle_to_quaternion.m,
vector_to_crossmat.m and C++/include/matrix_math.hpp:26-99 -- fp64 to rounding, fp32 to a few ulp."""
import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import BatchedFilter, capi

pytestmark = pytest.mark.gpu
N = 512


def _quats(rng, n=N):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


@pytest.mark.parametrize("dtype,tol", [(64, 2e-15), (32, 4e-7)])
def test_l0_helpers_against_the_oracle(dtype, tol):
    rng = np.random.default_rng(2)
    cast = (lambda a: a) if dtype == 64 else (lambda a: a.astype(np.float32).astype(np.float64))
    p, q = cast(_quats(rng)), cast(_quats(rng))
    with BatchedFilter(1, capi.default_params(0), dtype=dtype) as flt:
        # Hamilton product, [w x y z]   quaternion_add.m:22-28
        got = flt.l0_eval(capi.L0_QUAT_MUL, p, q)
        ref = np.array([oc.l0("quat_mul", a, b, out=4) for a, b in zip(p, q)])
        assert np.abs(got - ref).max() < tol
        # rotation matrices, both formulas   quaternion_to_rotmat.m:22-33 ; Eigen toRotationMatrix
        for op, name in ((capi.L0_QUAT_TO_ROTMAT_M, "quat_to_rotmat"), (capi.L0_QUAT_TO_ROTMAT_E, "quat_to_rotmat_eigen")):
            got = flt.l0_eval(op, q)
            ref = np.array([oc.l0(name, a, out=9) for a in q])
            assert np.abs(got - ref).max() < 2 * tol
            R = got.reshape(-1, 3, 3).astype(np.float64)
            assert np.abs(R @ np.swapaxes(R, 1, 2) - np.eye(3)).max() < 8 * tol      # orthonormal
            assert np.abs(np.linalg.det(R) - 1).max() < 8 * tol                       # right-handed
        # normalisation   quaternion_normalize.m:22-24
        raw = cast(rng.normal(size=(N, 4)) * rng.uniform(0.1, 10, (N, 1)))
        got = flt.l0_eval(capi.L0_QUAT_NORMALIZE, raw)
        assert np.abs(got - raw / np.linalg.norm(raw, axis=1, keepdims=True)).max() < tol
        # expm(-[w]x dt): the closed form of predict_nominal against the oracle's (ImuUpdate.m:68), small and large angles
        w = cast(rng.normal(size=(N, 3)) * rng.choice([1e-3, 0.05, 2.0, 30.0], (N, 1)))
        dt = cast(rng.uniform(1e-3, 2e-2, (N, 1)))
        got = flt.l0_eval(capi.L0_EXPM_SO3_NEG, w, dt)
        ref = np.array([oc.l0("expm_so3_neg", a, float(t[0]), out=9) for a, t in zip(w, dt)])
        assert np.abs(got - ref).max() < 2 * tol
        E = got.reshape(-1, 3, 3).astype(np.float64)
        assert np.abs(E @ np.swapaxes(E, 1, 2) - np.eye(3)).max() < 8 * tol
        # delta-theta -> quaternion of the injection (MeasureUpdate.m:94 ; filter.cpp:727-730), incl. the guarded zero
        dth = cast(rng.normal(size=(N, 3)) * rng.choice([0.0, 1e-6, 1e-2, 1.0], (N, 1)))
        got = flt.l0_eval(capi.L0_DTHETA_TO_QUAT, dth)
        ang = np.linalg.norm(dth, axis=1)
        ref = np.array([oc.l0("axisangle_to_quat", a, float(t), out=4) if t > 0 else [1.0, 0, 0, 0] for a, t in zip(dth, ang)])
        assert np.abs(got - ref).max() < tol and np.isfinite(got).all()
        # sin / cos of x and x / 2 on both sides of the 0.5 rad switch between the polynomial and the library
        x = cast(np.concatenate([np.linspace(-0.6, 0.6, N // 2), rng.uniform(-3, 3, N // 2)])[:, None])
        got = flt.l0_eval(capi.L0_SINCOS_HALF, x)
        ref = np.concatenate([np.sin(x), np.cos(x), np.sin(x / 2), np.cos(x / 2)], axis=1)
        assert np.abs(got - ref).max() < tol


def test_l0_eval_rejects_bad_arguments():
    with BatchedFilter(1, capi.default_params(0)) as flt:
        with pytest.raises(capi.FbusError):
            flt._check(flt._lib.fbus_ekf_l0_eval(flt._h, 99, 1, None, None, None), "l0_eval")
        with pytest.raises(ValueError):
            flt.l0_eval(capi.L0_QUAT_MUL, np.zeros((2, 4)), None)          # quat_mul needs b: the mirror refuses
        a = np.zeros((2, 4), np.float32)
        rc = flt._lib.fbus_ekf_l0_eval(flt._h, capi.L0_QUAT_MUL, 2, a.ctypes.data, None, a.ctypes.data)
        assert rc != 0                                                       # ... and so does the C ABI
