"""GPU suite, row f-1: marker pose from stereo corners (refractive / pin-hole triangulation + pose fit),
HIP kernel through the C ABI vs (a) the reference's own recorded data (tests/golden/vision_*.npz, slices of
matlab/dataset/*/corners.txt -> image.txt) and (b) the fp64 C oracle (oracle/vision_oracle.c)."""
import os

import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import BatchedFilter, capi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _qerr(a, b):
    """max component error between quaternions up to the common sign"""
    return float(np.minimum(np.abs(a - b).max(axis=-1), np.abs(a + b).max(axis=-1)).max())


def _oracle_refractive(left, right):
    p = oc.vision_params()
    pos, quat, c3 = [], [], []
    for l, r in zip(left, right):
        c = oc.refraction_triangulate(p, l, r)
        a, b, _ = oc.marker_pose(c)
        pos.append(a); quat.append(b); c3.append(c)
    return np.array(pos), np.array(quat), np.array(c3)


@pytest.mark.parametrize("dtype,tol_p,tol_q", [(64, 1.5e-5, 5e-5), (32, 2.5e-5, 8e-5)])
def test_water_recording_refractive_chain(dtype, tol_p, tol_q):
    """reference data: undistorted corner pairs -> logged marker pose (6 significant digits in the files)"""
    d = np.load(os.path.join(GOLD, "vision_water.npz"))
    c, im = d["corners"], d["image"]
    with BatchedFilter(1, capi.default_params(1), dtype=dtype) as flt:
        pos, quat = flt.marker_pose(c[:, 2:10], c[:, 10:18], capi.VIS_REFRACTIVE)
    assert np.abs(pos - im[:, 2:5]).max() < tol_p
    assert _qerr(quat.astype(np.float64), im[:, 5:9]) < tol_q


@pytest.mark.parametrize("dtype", [64, 32])
def test_land_recording_pose_fit(dtype):
    d = np.load(os.path.join(GOLD, "vision_land.npz"))
    c, im = d["corners"], d["image"]
    with BatchedFilter(1, capi.default_params(1), dtype=dtype) as flt:
        pos, quat = flt.marker_pose(c[:, 2:14], None, capi.VIS_CORNERS3D)
    assert np.abs(pos - im[:, 2:5]).max() < 1.5e-5
    assert np.abs(quat - im[:, 5:9]).max() < 5e-5


def _perturbed_inputs(n, seed=11):
    d = np.load(os.path.join(GOLD, "vision_water.npz"))["corners"]
    rng = np.random.default_rng(seed)
    base = d[rng.integers(0, len(d), n)]
    left = base[:, 2:10] + rng.normal(0, 0.02, (n, 8))
    right = base[:, 10:18] + rng.normal(0, 0.02, (n, 8))
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    return r32(left), r32(right)


@pytest.mark.parametrize("dtype,tol", [(64, 1e-10), (32, 2e-5)])
def test_refractive_matches_oracle_on_perturbed_corners(dtype, tol):
    left, right = _perturbed_inputs(2000)
    o_pos, o_quat, o_c3 = _oracle_refractive(left, right)
    with BatchedFilter(1, capi.default_params(1), dtype=dtype) as flt:
        pos, quat, c3 = flt.marker_pose(left, right, capi.VIS_REFRACTIVE, want_corners=True)
    scale = np.abs(o_c3).max()
    assert np.abs(c3 - o_c3).max() / scale < tol
    assert np.abs(pos - o_pos).max() / scale < tol
    assert _qerr(quat.astype(np.float64), o_quat) < 50 * tol        # noisy corners: the plane fit amplifies
    # Eigen's Quaterniond(Matrix3d) does not renormalise (vision.cpp:758): the norm carries the orthogonality error of the fitted
    # frame, which in fp32 depends on the last bits of the corners (r3: 1 ulp differences of the triangulation moved the
    # maximum over these 2000 markers from < 1e-6 to 3.9e-6; test_large_batch_properties allows 1e-4 on noisier corners)
    assert np.abs(np.linalg.norm(quat, axis=1) - 1).max() < (1e-5 if dtype == 32 else 1e-14)


@pytest.mark.parametrize("dtype,tol", [(64, 1e-9), (32, 5e-4)])
def test_pinhole_matches_oracle(dtype, tol):
    left, right = _perturbed_inputs(500, seed=5)
    p = oc.vision_params()
    lib = oc.load()
    ref = np.zeros((len(left), 12))
    import ctypes as C
    for i in range(len(left)):
        out = np.zeros(12)
        lib.fbv_normal_triangulate(C.byref(p), left[i].ctypes.data_as(C.POINTER(C.c_double)),
                                   right[i].ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
        ref[i] = out
    with BatchedFilter(1, capi.default_params(1), dtype=dtype) as flt:
        _, _, c3 = flt.marker_pose(left, right, capi.VIS_PINHOLE, want_corners=True)
    assert np.abs(c3.reshape(-1, 12) - ref).max() / np.abs(ref).max() < tol


def test_corners_to_correct_chain_on_device():
    """corners -> marker pose -> correct(), all on the device with torch tensors, vs the oracle chain"""
    import torch
    from fbus_ekf import synth
    from replay_ref import OracleEngine
    from util import COV_BLOCK_TOL, COV_BLOCK_TOL_F64, COV_TOL, STATE_TOL, cov_rel_err, cov_rel_err_blockwise, state_rel_err
    B = 512
    prm = capi.default_params(1)
    d = np.load(os.path.join(GOLD, "vision_water.npz"))["corners"]
    rng = np.random.default_rng(3)
    base = d[rng.integers(0, len(d), B)]
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    left, right = r32(base[:, 2:10]), r32(base[:, 10:18])
    o_pos, o_quat, _ = _oracle_refractive(left, right)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    # put every filter where marker 0 is seen as measured, plus a small offset -> modest innovations
    from fbus_ekf import replay
    for b in range(B):
        p, q, R = replay.pose_from_marker(np.concatenate([[0], o_pos[b], o_quat[b]]), prm)
        nom[b, 0:3], nom[b, 6:10], rot[b] = p + rng.normal(0, 0.01, 3), q, R.ravel()
    nom, rot, P = r32(nom), r32(rot), r32(P)
    dev = torch.device("cuda:0")
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        tl = torch.from_numpy(left.astype(np.float32)).to(dev)
        tr = torch.from_numpy(right.astype(np.float32)).to(dev)
        pos, quat = flt.marker_pose(tl, tr, capi.VIS_REFRACTIVE)
        ids = torch.zeros(B, dtype=torch.int32, device=dev)
        flt.correct(ids, pos, quat, capi.MODE_NEAREST)
        flt.sync()
        g = flt.get_state()
    eng = OracleEngine(B, 1, 18)
    eng.set_state(nom, rot, P, prev)
    eng.correct(np.zeros((B, 1), np.int32), o_pos[:, None, :], o_quat[:, None, :], 0)
    # (round 5) marker_pose_kernel<float> triangulates and fits the pose in double (rounds 1-4: in fp32, ~2e-6 m off in the corner
    # positions, 10x gates here): the measurement reaches correct() rounded to fp32 once, the standard single-step gates apply
    es, where = state_rel_err(g[0], eng.nominal, eng.P)
    ec, eb = cov_rel_err(g[2], eng.P), cov_rel_err_blockwise(g[2], eng.P)
    print(f"[parity] marker_pose -> correct chain fp32: sigma-aware {es:.2e} ({where}) cov {ec:.2e} cov block-wise {eb:.2e}")
    assert es <= STATE_TOL
    assert ec <= COV_TOL
    assert eb <= COV_BLOCK_TOL


def test_large_batch_properties():
    """n = 262 144 markers: finite, unit quaternions, right-handed frames, rigid marker geometry preserved"""
    n = 262144
    left, right = _perturbed_inputs(4096, seed=2)
    left = np.tile(left, (n // 4096, 1)); right = np.tile(right, (n // 4096, 1))
    with BatchedFilter(1, capi.default_params(1), dtype=32) as flt:
        pos, quat, c3 = flt.marker_pose(left, right, capi.VIS_REFRACTIVE, want_corners=True)
    assert np.isfinite(pos).all() and np.isfinite(quat).all()
    # Eigen's Quaterniond(Matrix3d) does not renormalise (vision.cpp:758): the norm carries the fp32
    # orthogonality error of the fitted frame on these noisy corners
    assert np.abs(np.linalg.norm(quat, axis=1) - 1).max() < 1e-4
    assert np.array_equal(pos[:4096], pos[-4096:])                    # deterministic across the batch
    side = np.linalg.norm(c3[:, 1] - c3[:, 0], axis=1)
    assert 0.1 < np.median(side) < 0.5                                # the 0.28 m marker, noisy corners


@pytest.mark.parametrize("cov_form", [0, 1])
@pytest.mark.parametrize("dtype,mult", [(64, 1e-4), (32, 1.0)])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("dialect", [0, 1])
def test_correct_from_stereo_corners_matches_oracle(dialect, mode, dtype, mult, cov_form):
    """correct_corners (north-star extension: triangulated corner positions as 3-row measurements, 12 rows per
    marker).  The reference has no counterpart -> parity unpinned by construction; validated against the fp64
    oracle chain (vision_oracle triangulation -> fbo_correct_corners).  Round 4 (csrc/ekf_meas.hpp): the kernel triangulates
    and folds in double whatever the record type, so the fp32 records meet the un-multiplied single-step gates (round 3
    triangulated in fp32, ~2e-6 m off in the corner positions, and had 10x); fp64 is tight.
    cov_form (round 6): the oracle's literal (I - K H) P (MeasureUpdate.m:101-102) and its Joseph form (north_star) -- the kernels'
    one-shot update is held to both (fbus_params::cov_form selects nothing in this entry point: include/fbus_ekf.h)."""
    from fbus_ekf import synth
    from replay_ref import OracleEngine
    from util import COV_BLOCK_TOL, COV_BLOCK_TOL_F64, COV_TOL, STATE_TOL, cov_rel_err, cov_rel_err_blockwise, state_rel_err
    B, M, size = 256, 3, 0.117
    prm = capi.default_params(dialect)
    prm.marker_size = size
    prm.cov_form = cov_form
    rng = np.random.default_rng(7)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    nom, rot, P, prev = synth.initial_state(300, 300 + B, list(prm.p0_diag), 18, mixed_cov=True)
    nom, rot, P = r32(nom), r32(rot), r32(P)
    ids, _, _ = synth.marker_frame(300, 300 + B, 0, M, nom, prm)
    ids[0] = -1
    ids[1, 0] = 9
    d = np.load(os.path.join(GOLD, "vision_water.npz"))["corners"]
    base = d[rng.integers(0, len(d), B * M)]
    left = r32(base[:, 2:10] + rng.normal(0, 0.003, (B * M, 8))).reshape(B, M, 8)
    right = r32(base[:, 10:18] + rng.normal(0, 0.003, (B * M, 8))).reshape(B, M, 8)
    p = oc.vision_params()
    corners = np.array([oc.refraction_triangulate(p, l, r) for l, r in zip(left.reshape(-1, 8), right.reshape(-1, 8))])
    corners = corners.reshape(B, M, 4, 3)
    # place every filter so that its first VALID marker is seen roughly where the corners are: modest innovations
    from fbus_ekf import replay
    for b in range(B):
        m = next((k for k in range(M) if ids[b, k] >= 0 and ids[b, k] != 9), None)
        if m is None:
            continue
        pos, quat, _ = oc.marker_pose(corners[b, m])
        pp, qq, RR = replay.pose_from_marker(np.concatenate([[ids[b, m]], pos, quat]), prm)
        nom[b, 0:3], nom[b, 6:10], rot[b] = pp + rng.normal(0, 0.005, 3), qq, RR.ravel()
    nom, rot = r32(nom), r32(rot)
    eng = OracleEngine(B, dialect, 18, cov_form=cov_form)
    eng.set_state(nom, rot, P, prev)
    ok = eng.orc.correct_corners(eng.nominal, eng.rot, eng.P, eng.prev, ids, corners, size, mode)
    # stacked mode at this batch size runs the four-role form of correct_corners2_kernel (the markers divided among four waves per tile): the
    # default, the one-wave kernel (set_team 1) and the two-role form all go through the same gate
    for roles in ((0, 1, 2) if (dtype == 32 and mode == 1) else (0,)):
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_team(0, roles)
            flt.set_state(nom, rot, P, prev)
            flt.correct_corners(ids, left, right, capi.VIS_REFRACTIVE, mode)
            g = flt.get_state()
            ap = flt.applied()
        assert (ap == ok).all() and ok[0] == 0 and ok[2:].all(), roles
        assert (g[3] == eng.prev).all()
        print(f"[parity] correct_corners dialect {dialect} mode {mode} fp{dtype} correct_roles {roles} vs {'Joseph' if cov_form else 'simple'}-form oracle: sigma-aware "
              f"{state_rel_err(g[0], eng.nominal, eng.P)[0]:.2e} cov {cov_rel_err(g[2], eng.P):.2e} cov block-wise {cov_rel_err_blockwise(g[2], eng.P):.2e}")
        assert state_rel_err(g[0], eng.nominal, eng.P)[0] <= STATE_TOL * mult, roles
        assert cov_rel_err(g[2], eng.P) <= COV_TOL * min(mult, 1.0), roles
        # (the oracle's own two forms differ by 1e-12 .. 1e-9 block-wise in double: tests/test_oracle_pixels_cpu.py)
        assert cov_rel_err_blockwise(g[2], eng.P) <= ((1e-9 if cov_form else COV_BLOCK_TOL_F64) if dtype == 64 else COV_BLOCK_TOL * mult), roles
        assert not np.array_equal(g[0][2:], nom[2:].astype(g[0].dtype))           # it did update


@pytest.mark.gpu
def test_corner_recordings_to_image_rows():
    """the data formats in front of the path: corners.txt rows in, image.txt rows out (fbus_ekf.replay), against the
    image.txt the reference recorded beside them -- water (refractive stereo) and land (3-D corners)"""
    from fbus_ekf import replay
    with BatchedFilter(1, capi.default_params(1), dtype=64) as flt:
        for name, geom in (("vision_water.npz", capi.VIS_REFRACTIVE), ("vision_land.npz", capi.VIS_CORNERS3D)):
            d = np.load(os.path.join(GOLD, name))
            rows = replay.image_from_corners(flt, d["corners"], geom)
            im = d["image"]
            assert rows.shape == (len(im), 9)
            assert np.array_equal(rows[:, 0], d["corners"][:, 0]) and np.array_equal(rows[:, 1], im[:, 1])
            assert np.abs(rows[:, 2:5] - im[:, 2:5]).max() < 1.5e-5
            assert _qerr(rows[:, 5:9], im[:, 5:9]) < 5e-5
