"""GPU suite: correct() from corner PIXELS -- the north star's "flat-port refractive stereo reprojection of ArUco corners,
per-corner 2 x N Jacobians" (`fbus_ekf_correct_pixels`).  No reference counterpart: the reference has only the
back-projection.  What pins it: the oracle's forward projection inverts the reference-pinned back-projection to 1e-12
(tests/test_oracle_cpu.py).  TWO oracles (round 6): `fbo_correct_pixels_analytic`, whose d pi / d X is the implicit-function-theorem
closed form written from the forward model (oracle/vision_oracle.c::fbv_project_camera_jac) -- the fp64 kernels are held to it at
1e-9 like every other fp64 kernel --, and `fbo_correct_pixels`, whose rows are CENTRAL DIFFERENCES of the projection (independent of
any closed form; 1e-6 as in round 5: its own finite-difference error); the two are cross-checked on the CPU
(tests/test_oracle_pixels_cpu.py).  fp32 within the standard single-step gates.  The covariance form: the kernels' one-shot update
against the oracle's literal (I - K H) P AND against its Joseph form (I - K H) P (I - K H)' + K R K' (MeasureUpdate.m:101-102 is the
simple form; north_star names Joseph)."""
import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import (COV_BLOCK_TOL, COV_TOL, PLAIN_TOL, STATE_TOL, WINDOW_TOL, assert_parity, parity_errors, pixel_scene)

pytestmark = pytest.mark.gpu
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
SIZE = 0.28


def _scene(B, M, dialect, seed, noise=5e-4):
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=seed, noise=noise, nominal=nom0)
    rng = np.random.default_rng(seed + 1)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))                      # innovations of a few mm / mrad
    dq = np.concatenate([np.ones((B, 1)), rng.normal(0, 0.002, (B, 3))], axis=1)
    nom[:, 6:10] = synth.qmul(nom[:, 6:10], dq)
    nom[:, 6:10] /= np.linalg.norm(nom[:, 6:10], axis=1, keepdims=True)
    nom = r32(nom)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    return prm, nom, rot, r32(P), prev, ids, r32(left), r32(right)


def _oracle_update(B, dialect, n, nom, rot, P, prev, ids, left, rgt, r_pix, skip=None, analytic=True, cov_form=oc.SIMPLE, vision=None):
    eng = OracleEngine(B, dialect, n, cov_form=cov_form)
    eng.set_state(nom, rot, P, prev)
    keep = eng.get_state()
    ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, rgt, SIZE, r_pix, vision=vision, analytic=analytic)
    if skip is not None:
        now = eng.get_state()
        for x, y in zip(now, keep):
            x[skip == 1] = y[skip == 1]
        eng.set_state(*now)
        ok[skip == 1] = 0
    return eng.get_state(), ok


F64_TOL = 1e-9          # every fp64 kernel against the oracle (tests/util.py); round 5 held the pixel rows to 1e-6 against the FD oracle


@pytest.mark.parametrize("cov_form", [oc.SIMPLE, oc.JOSEPH])
@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("dialect", [0, 1])
def test_correct_pixels_matches_the_oracle(dialect, stereo, cov_form, monkeypatch):
    B, M = 320, 4
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, dialect, seed=11 + dialect)
    prm.cov_form = cov_form                                        # (selects nothing in the kernels: their one form meets both oracles)
    ids[0] = -1                                                    # nothing visible
    ids[1, :] = 9                                                  # only ids outside the map
    skip = (np.arange(B) % 11 == 5).astype(np.uint8)
    rgt = right if stereo else None
    want, ok = _oracle_update(B, dialect, 18, nom, rot, P, prev, ids, left, rgt, prm.r_pix, skip, analytic=True, cov_form=cov_form)
    want_fd, ok_fd = _oracle_update(B, dialect, 18, nom, rot, P, prev, ids, left, rgt, prm.r_pix, skip, analytic=False, cov_form=cov_form)
    assert ok[0] == 0 and ok[1] == 0 and ok[2:][skip[2:] == 0].all() and np.array_equal(ok, ok_fd)
    # at this batch size the default is the divided-tail kernel (correct_pixels_split_kernel, csrc/ekf_meas_split.hpp: a solver wave and
    # an updater wave per tile; fp32, square port); the one-wave kernel (set_team correct_roles = 1), and -- with FBUS_MEAS_SPLIT=0, read
    # at fbus_ekf_create -- the two- and four-role forms of correct_pixels2_kernel (the markers divided among the waves of a tile, one
    # tail) go through the same gate
    for dtype, roles, split in ((64, 0, None), (64, 1, None), (32, 0, None), (32, 1, None), (32, 2, None), (32, 2, "0"), (32, 0, "0")):
        if split is not None:
            monkeypatch.setenv("FBUS_MEAS_SPLIT", split)
        else:
            monkeypatch.delenv("FBUS_MEAS_SPLIT", raising=False)
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_team(0, roles)
            if split == "0":
                assert flt.launch_info(capi.INFO_MEAS_SPLIT, M) == 0 and flt.launch_info(capi.INFO_ROLES_MEAS, M) > 1
            flt.set_state(nom, rot, P, prev)
            flt.correct_pixels(ids, left, rgt, skip)
            got = flt.get_state()
            assert (flt.applied() == ok).all()
        untouched = ok == 0
        assert np.array_equal(got[0][untouched], nom[untouched].astype(got[0].dtype))
        e = parity_errors(got, want)
        form = "Joseph" if cov_form == oc.JOSEPH else "simple"
        print(f"[parity] correct_pixels dialect {dialect} {'stereo' if stereo else 'left'} fp{dtype} correct_roles {roles}"
              f"{' one-tail' if split == '0' else ''} vs {form}-form oracle: literal {e['literal']:.2e} "
              f"sigma-aware {e['sigma']:.2e} ({e['sigma_block']}) plain {e['plain']:.2e} ({e['plain_block']}) cov {e['cov']:.2e} "
              f"cov block-wise {e['cov_block']:.2e}")
        if dtype == 64:
            # Against the Joseph-form oracle everything holds 1e-9 (measured: literal 3e-13, block-wise covariance 5e-13).  Against the
            # oracle's LITERAL (I - K H) P the state and the max-norm covariance do too, the block-wise covariance figure does not -- and it
            # is the ORACLE that is off: 32-64 rows at sigma_pix = 1e-3 shrink the pose variances by four decades, and (I - K H) P
            # subtracts nearly equal numbers there (block-wise 7e-10 left, 2e-8 stereo in double; the oracle's own two forms differ by
            # as much, tests/test_oracle_pixels_cpu.py), which the one-shot form G P(J,:) and Joseph's form never do.
            assert e["literal"] < F64_TOL and e["sigma"] < F64_TOL and e["cov"] < F64_TOL and e["asym"] == 0
            assert e["cov_block"] < (F64_TOL if cov_form == oc.JOSEPH else 1e-6)
            # and the independent central-difference oracle, to its own error (eps = 1e-6 m: ~1e-9 relative on H, times the gain of
            # 32-64 rows at r_pix = 1e-6)
            f = parity_errors(got, want_fd)
            assert f["literal"] < 1e-6 and f["sigma"] < 1e-6 and f["cov_block"] < 1e-6
        else:
            # (round 4) the single-step gates of every other parity test, un-widened: literal and sigma-aware 1e-5, plain 2e-4,
            # covariance 1e-4 max-norm and 1e-5 block-wise.  32-64 rows at sigma_pix = 1e-3 carry ~50x the information of a marker
            # pose and shrink the pose variances by four decades in ONE update; the round-3 kernel (fp32 throughout, six sequential
            # rank-1 passes) read literal 2e-5 .. 4e-5 and block-wise covariance 2e-4 .. 3e-3 here and had its gates at 5e-5 / 4e-3.
            # Measured now: literal <= 9e-7, sigma-aware <= 1.6e-6, block-wise covariance <= 6.4e-7 (csrc/ekf_meas.hpp).
            assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["plain"] <= PLAIN_TOL
            assert e["cov"] <= COV_TOL and e["cov_block"] <= COV_BLOCK_TOL and e["asym"] == 0
            # and the posterior is positive definite (checked on the correlation matrix: fp32 cannot hold eigenvalues 11 decades
            # apart to their own size)
            Ps = got[2][ok == 1].astype(np.float64)
            dg = np.sqrt(np.einsum("bii->bi", Ps))
            assert np.linalg.eigvalsh(Ps / (dg[:, :, None] * dg[:, None, :])).min() > 0


def test_a_corner_exactly_on_the_camera_axis():
    """The fold's on-axis case (rho = 0: t / rho is its limit 1 / L_t, and the camera-frame rows of the left-only kernel take e = (1, 0) there)
    is measure-zero in every other scene.  Here the extrinsics are the identity (T_SC_left = 1: McL = F R_IL = 1, P_IL = 0), the filter sits at the
    origin with R = 1 and marker 0's first corner at (0, 0, 1): its lateral offset is exactly zero in the kernels' arithmetic.  fp64 and fp32
    (one-wave left-only kernel, divided-tail kernel, stereo never: the right camera is somewhere else) against the analytic oracle."""
    B, M, dialect = 128, 2, 0
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    for i in range(16):
        prm.T_SC_left[i] = 1.0 if i % 5 == 0 else 0.0
    for i in range(3):
        prm.marker_pos[0][i] = (0.0, 0.0, 1.0)[i]
    for i in range(9):
        prm.marker_rot[0][i] = 1.0 if i % 4 == 0 else 0.0
    R_IL, P_IL, Q_IL = synth.camera_constants(prm)
    assert np.array_equal(np.diag([-1.0, -1.0, 1.0]) @ R_IL, np.eye(3)) and not P_IL.any()
    vp = oc.vision_params()
    for i in range(9):
        vp.R_IL[i] = 1.0 if i % 4 == 0 else 0.0
    for i in range(3):
        vp.P_LI[i] = 0.0
    mid0 = int(prm.marker_id[0])
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    nom = np.array(nom0, float)
    nom[:, 0:3] = 0.0
    nom[:, 6:10] = (1.0, 0.0, 0.0, 0.0)
    nom = r32(nom)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    assert np.array_equal(rot[0].reshape(3, 3), np.eye(3))
    cam = np.array([[0, 0, 1.0], [0, SIZE, 1.0], [SIZE, SIZE, 1.0], [SIZE, 0, 1.0]]) @ R_IL.T    # the marker's corners in the left camera frame: R_IL (c - p)
    uvL, _, ok = oc.project_stereo(vp, cam, stereo=False)
    assert ok.all() and not uvL[0].any()                                                        # corner 0 projects to the principal point
    rng = np.random.default_rng(5)
    ids = np.full((B, M), -1, np.int32); ids[:, 0] = mid0
    left = np.zeros((B, M, 8))
    left[:, 0, :] = uvL.reshape(8) + rng.normal(0, 5e-4, (B, 8))
    left, P = r32(left), r32(P)
    eng = OracleEngine(B, dialect, 18, cov_form=oc.JOSEPH)
    for i in range(9):
        eng.orc.prm.R_IL[i] = float(R_IL.reshape(9)[i])
    for i in range(3):
        eng.orc.prm.P_IL[i] = 0.0
        eng.orc.prm.marker_pos[0][i] = (0.0, 0.0, 1.0)[i]
    for i in range(4):
        eng.orc.prm.Q_IL[i] = float(Q_IL[i])
        eng.orc.prm.marker_quat[0][i] = (1.0, 0.0, 0.0, 0.0)[i]
    eng.set_state(nom, rot, P, prev)
    ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, None, SIZE, prm.r_pix, vision=vp, analytic=True)
    assert ok.all()
    want = eng.get_state()
    assert np.abs(want[0][:, 0:3] - nom[:, 0:3]).max() > 1e-5                                  # the update did something
    for dtype, roles in ((64, 1), (32, 1), (32, 0)):
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_team(0, roles)
            flt.set_state(nom, rot, P, prev)
            flt.correct_pixels(ids, left, None)
            got = flt.get_state()
            assert flt.applied().all()
        e = parity_errors(got, want)
        print(f"[parity] a corner on the camera axis, correct_pixels left fp{dtype} correct_roles {roles}: literal {e['literal']:.2e} "
              f"sigma-aware {e['sigma']:.2e} cov block-wise {e['cov_block']:.2e}")
        assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
        if dtype == 64:
            assert e["literal"] < F64_TOL and e["sigma"] < F64_TOL and e["cov_block"] < F64_TOL
        else:
            assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["plain"] <= PLAIN_TOL and e["cov_block"] <= COV_BLOCK_TOL


def test_the_fifteen_state_filter_takes_the_same_updates():
    """north_star's literal filter has 15 error states (per-corner 2 x 15 Jacobians): N = 18 without the gravity block.  The reprojection-row
    update (left, stereo) and the corner-row update through the N = 15 kernels against the N = 15 oracle, standard gates."""
    B, M, dialect, n = 192, 4, 0, 15
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=31, noise=5e-4, nominal=nom0)
    rng = np.random.default_rng(32)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))
    nom = r32(nom)
    rot, P, left, right = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9)), r32(P), r32(left), r32(right)
    assert P.shape[1:] == (n, n)
    vp = oc.vision_params()
    corners = np.zeros((B, M, 4, 3))
    for b in range(B):
        for m in range(M):
            if ids[b, m] >= 0:
                corners[b, m] = oc.refraction_triangulate(vp, left[b, m], right[b, m])
    for what in ("pixels left", "pixels stereo", "corners"):
        eng = OracleEngine(B, dialect, n)
        eng.set_state(nom, rot, P, prev)
        if what == "corners":
            ok = eng.orc.correct_corners(eng.nominal, eng.rot, eng.P, eng.prev, ids, corners, SIZE, capi.MODE_STACKED)
        else:
            ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, right if "stereo" in what else None, SIZE, prm.r_pix,
                                        analytic=True)
        assert ok.all()
        for dtype in (64, 32):
            with BatchedFilter(B, prm, dtype=dtype, nstate=n) as flt:
                flt.set_state(nom, rot, P, prev)
                if what == "corners":
                    flt.correct_corners(ids, left, right, capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                else:
                    flt.correct_pixels(ids, left, right if "stereo" in what else None)
                got = flt.get_state()
                assert (flt.applied() == ok).all()
            e = parity_errors(got, eng.get_state())
            print(f"[parity] N = 15, {what} fp{dtype}: literal {e['literal']:.2e} sigma-aware {e['sigma']:.2e} cov block-wise {e['cov_block']:.2e}")
            if dtype == 64:
                # (corner rows: r_pos = 1e-3 m leaves (I - K H) P well conditioned, 1e-9 holds; reprojection rows: the oracle's literal form
                # cancels -- see test_correct_pixels_matches_the_oracle, which holds the kernel to the Joseph-form oracle at 1e-9)
                assert e["literal"] < F64_TOL and e["sigma"] < F64_TOL and e["asym"] == 0
                assert e["cov_block"] < (F64_TOL if what == "corners" else 1e-6)
            else:
                assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["plain"] <= PLAIN_TOL
                assert e["cov"] <= COV_TOL and e["cov_block"] <= COV_BLOCK_TOL and e["asym"] == 0


def test_tilted_port_takes_the_general_normal_path():
    """The reference's configuration has the port square to the camera (normal_vector 0 0 1: paramconfig.yml:39-42, refractinfo.yml:10-13)
    and the kernels have a form specialised for it; a port tilted by two degrees goes through the general form: reprojection rows (left,
    stereo) and triangulated corners against the oracle with the same normal, standard gates."""
    B, M, dialect = 192, 4, 0
    nrm = np.array([0.03, -0.02, 1.0]); nrm /= np.linalg.norm(nrm)
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    vp = oc.vision_params()
    for i in range(3):
        prm.port_normal[i] = float(nrm[i]); vp.normal[i] = float(nrm[i])
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=21, noise=5e-4, nominal=nom0, vision=vp)
    rng = np.random.default_rng(22)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))
    nom = r32(nom)
    rot, P, left, right = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9)), r32(P), r32(left), r32(right)
    for stereo in (False, True):
        rgt = right if stereo else None
        eng = OracleEngine(B, dialect, 18, cov_form=oc.JOSEPH)       # (the form whose block-wise covariance is good to 1e-9 in double)
        eng.set_state(nom, rot, P, prev)
        ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, rgt, SIZE, prm.r_pix, vision=vp, analytic=True)
        assert ok.all()
        for dtype in (64, 32):
            with BatchedFilter(B, prm, dtype=dtype) as flt:
                flt.set_team(0, 1)
                flt.set_state(nom, rot, P, prev)
                flt.correct_pixels(ids, left, rgt)
                got = flt.get_state()
            e = parity_errors(got, eng.get_state())
            print(f"[parity] tilted port, correct_pixels {'stereo' if stereo else 'left'} fp{dtype}: literal {e['literal']:.2e} sigma-aware {e['sigma']:.2e} "
                  f"cov block-wise {e['cov_block']:.2e}")
            if dtype == 64:
                assert e["literal"] < F64_TOL and e["sigma"] < F64_TOL and e["cov_block"] < F64_TOL
            else:
                assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["plain"] <= PLAIN_TOL and e["cov_block"] <= COV_BLOCK_TOL
    # the corner-row update triangulates through the same port (general form of tri_corners_refractive)
    corners = np.zeros((B, M, 4, 3))
    for b in range(B):
        for m in range(M):
            if ids[b, m] >= 0:
                corners[b, m] = oc.refraction_triangulate(vp, left[b, m], right[b, m])
    eng = OracleEngine(B, dialect, 18)
    eng.set_state(nom, rot, P, prev)
    ok = eng.orc.correct_corners(eng.nominal, eng.rot, eng.P, eng.prev, ids, corners, SIZE, capi.MODE_STACKED)
    assert ok.all()
    for dtype in (64, 32):
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_state(nom, rot, P, prev)
            flt.correct_corners(ids, left, right, capi.VIS_REFRACTIVE, capi.MODE_STACKED)
            gc = flt.get_state()
        e = parity_errors(gc, eng.get_state())
        print(f"[parity] tilted port, correct_corners fp{dtype}: literal {e['literal']:.2e} sigma-aware {e['sigma']:.2e} cov block-wise {e['cov_block']:.2e}")
        if dtype == 64:
            assert e["literal"] < 1e-9 and e["sigma"] < 1e-9 and e["cov_block"] < 1e-9
        else:
            assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["plain"] <= PLAIN_TOL and e["cov_block"] <= COV_BLOCK_TOL
    # a square port gives a different answer on the same data (the tilt is not silently ignored)
    prm0 = capi.default_params(dialect)
    prm0.marker_size = SIZE
    with BatchedFilter(B, prm0, dtype=64) as flt:
        flt.set_state(nom, rot, P, prev)
        flt.correct_pixels(ids, left, None)
        sq = flt.get_state()
    assert np.abs(sq[0][:, 0:3] - got[0][:, 0:3]).max() > 1e-4


def test_correct_pixels_converges_on_the_true_pose():
    """repeated updates with the projections of the TRUE corners pull a displaced fp32 filter onto the truth (1 mm / 1 mrad),
    a different check from oracle parity: the Jacobian has the right sign and scale"""
    B, M = 256, 3
    prm = capi.default_params(0)
    prm.marker_size = SIZE
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=5)
    rng = np.random.default_rng(6)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.01, (B, 3))
    rot = synth.q2R(nom[:, 6:10]).reshape(B, 9)
    P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, np.zeros(B, np.int32))
        for _ in range(6):
            flt.correct_pixels(ids, left, right)
            # Gauss-Newton iteration: prior covariance again, and the carried rotation matrix refreshed from the corrected
            # quaternion (MeasureUpdate leaves it stale on purpose -- MeasureUpdate.m:92-98 -- and h(x) reads it)
            g = flt.get_state()
            flt.set_state(g[0], synth.q2R(g[0][:, 6:10].astype(np.float64)).reshape(B, 9), P, None)
        got = flt.get_state()
    perr = np.abs(got[0][:, 0:3] - truth[:, 0:3]).max(axis=1)
    dq = synth.qmul(truth[:, 6:10] * np.array([1, -1, -1, -1.0]), got[0][:, 6:10].astype(np.float64))
    qerr = np.abs(dq[:, 1:]).max(axis=1)
    print(f"[parity] correct_pixels Gauss-Newton: position error median {np.median(perr):.1e} 99% {np.quantile(perr, 0.99):.1e} max {perr.max():.1e} m; "
          f"attitude median {np.median(qerr):.1e} max {qerr.max():.1e}")
    # from 1e-2 m: most filters are on the truth after six steps; a single oblique marker converges slowly (still shrinking)
    assert np.median(perr) < 1e-5 and np.quantile(perr, 0.9) < 1e-4 and np.quantile(qerr, 0.9) < 1e-4
    assert perr.max() < 1e-2 and np.isfinite(got[0]).all()


def _wall_map(prm, orc_prm, size):
    """a 4 x 4 wall of 16 markers (ids 0..15, 0.3 m pitch, all with the orientation of the reference's marker 0) in the product's
    and in the oracle's parameters: the reference's 12-marker room never shows more than 2-3 markers to one camera pose"""
    _, mpos, mquat = synth.marker_table(prm)
    R0 = np.array(list(prm.marker_rot[0])).reshape(3, 3)
    q0 = mquat[0]
    prm.n_markers = orc_prm.n_markers = 16
    for k in range(16):
        off = R0 @ np.array([0.3 * (k % 4 - 1.5), 0.3 * (k // 4 - 1.5), 0.0])
        prm.marker_id[k] = orc_prm.marker_id[k] = k
        for i in range(3):
            prm.marker_pos[k][i] = orc_prm.marker_pos[k][i] = float(mpos[0][i] + off[i])
        for i in range(9):
            prm.marker_rot[k][i] = float(R0.ravel()[i])
        for i in range(4):
            orc_prm.marker_quat[k][i] = float(q0[i])


def test_config5_128_reprojection_rows_at_full_batch():
    """the literal north-star shape: B = 65 536, 16 markers x 4 corners x 2 rows = 128 stacked reprojection rows per filter (left
    camera; 256 with both), fp32: finite, symmetric, positive definite, unit quaternions on every filter, and the oracle on a
    strided subset.  The scene is a wall of 16 markers seen from 1.2 - 1.8 m."""
    import torch
    B, M, size = 65536, 16, 0.15
    prm = capi.default_params(0)
    prm.marker_size = size
    eng_probe = OracleEngine(1, 0, 18)
    _wall_map(prm, eng_probe.orc.prm, size)
    base, _, ids_s, left_s, right_s = pixel_scene(256, M, prm, size, seed=9, noise=5e-4, depth=(1.2, 1.8))
    nvis = (ids_s >= 0).sum(axis=1)
    assert nvis.mean() > 14 and nvis.max() == 16, nvis
    rep = B // 256
    nom = np.tile(base, (rep, 1)); ids = np.tile(ids_s, (rep, 1)); left = np.tile(left_s, (rep, 1, 1)); right = np.tile(right_s, (rep, 1, 1))
    rng = np.random.default_rng(10)
    nom[:, 0:3] += rng.normal(0, 0.003, (B, 3))
    nom = r32(nom); left = r32(left); right = r32(right)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
    prev = np.zeros(B, np.int32)
    dev = torch.device("cuda:0")
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    sub = np.arange(0, B, 1021)
    with BatchedFilter(B, prm) as flt:
        d = (torch.from_numpy(ids).to(dev), f32(left), f32(right))
        for stereo in (False, True):
            flt.timing_enable(True); flt.timing_reset()
            for _ in range(3):
                flt.set_state(nom, rot, P, prev)
                flt.correct_pixels(d[0], d[1], d[2] if stereo else None)
            flt.sync()
            ms, n = flt.timing_read(capi.KERNEL_CORRECT_CORNERS)
            got = flt.get_state()
            ap = flt.applied()
            assert (ap == 1).all(), (int((ap == 0).sum()), np.nonzero(ap == 0)[0][:10])
            rows = (2 if not stereo else 4) * 4
            print(f"[perf] correct_pixels B = 65536, {nvis.mean():.1f} of 16 markers visible = {nvis.mean() * rows:.0f} rows per filter "
                  f"({'stereo' if stereo else 'left camera'}): {ms / n * 1e3:.1f} us per launch")
            assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
            assert np.abs(np.linalg.norm(got[0][:, 6:10], axis=1) - 1).max() < 1e-6
            assert np.abs(got[2] - np.swapaxes(got[2], 1, 2)).max() == 0
            # 128 rows at sigma_pix = 1e-3 leave position variances of 1e-9 m^2 beside P_gg = 100: positive definiteness is checked
            # on the correlation matrix (fp32 cannot hold eigenvalues 11 decades apart to their own size) -- strictly positive
            # (round 3 admitted -1e-5)
            Ps = got[2][::97].astype(np.float64)
            dg = np.sqrt(np.einsum("bii->bi", Ps))
            assert np.linalg.eigvalsh(Ps / (dg[:, :, None] * dg[:, None, :])).min() > 0
            eng = OracleEngine(len(sub), 0, 18)
            _wall_map(capi.default_params(0), eng.orc.prm, size)
            eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
            eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids[sub], left[sub], right[sub] if stereo else None, size, prm.r_pix)
            e = parity_errors([x[sub] for x in got], eng.get_state())
            print(f"[parity] correct_pixels {rows * 16}-row shape, fp32 vs oracle: literal {e['literal']:.2e} sigma-aware {e['sigma']:.2e} "
                  f"plain {e['plain']:.2e} cov {e['cov']:.2e} cov block-wise {e['cov_block']:.2e}")
            # the single-step gates, un-widened (round 3: 5e-5 / 1e-4 / 5e-3; measured now 8e-9 / 8e-8 / 1.2e-7)
            assert e["literal"] <= STATE_TOL and e["sigma"] <= STATE_TOL and e["cov"] <= COV_TOL and e["cov_block"] <= COV_BLOCK_TOL


@pytest.mark.parametrize("route", ["one_wave", "split", "one_tail_roles", "fused", "fp64"])
def test_padding_of_absent_slots_may_be_anything(route, monkeypatch):
    """Slots without a marker of the map (id -1, or an id outside it) are ignored whatever their image points hold -- NaN included.
    (Round 6 folds every slot with a weight instead of branching around it: 0 x NaN must not reach the sums.)  The same call with zeros
    and with NaN / huge values in those slots: bit-equal results and flags, on every kernel route."""
    import torch
    B, M = 192, 4
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, seed=71)
    ids[:, 3] = -1                                                   # an absent slot in every filter
    ids[::3, 1] = 9                                                  # an id outside the map in every third
    ids[5, :] = -1                                                   # nothing at all
    junk_l, junk_r = left.copy(), right.copy()
    bad = (ids < 0) | (ids == 9)
    junk_l[bad] = np.nan
    junk_r[bad] = 1e30
    junk_r[5] = np.nan
    dtype = 64 if route == "fp64" else 32
    if route == "one_tail_roles":
        monkeypatch.setenv("FBUS_MEAS_SPLIT", "0")
    out = []
    for l, r in ((left, right), (junk_l, junk_r)):
        for stereo in (False, True):
            with BatchedFilter(B, prm, dtype=dtype) as flt:
                if route in ("one_wave", "fused", "fp64"):
                    flt.set_team(1, 1)
                flt.set_state(nom, rot, P, prev)
                if route == "fused":
                    tt = torch.float32
                    dev = torch.device("cuda:0")
                    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if np.asarray(a).dtype.kind in "iu" else \
                        torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev).to(tt)
                    flt.frame_meas(None, None, None, d(ids), d(l), d(r) if stereo else None, capi.MEAS_PIXELS, capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                    flt.sync()
                else:
                    flt.correct_pixels(ids, l, r if stereo else None)
                out.append((flt.get_state(), flt.applied()))
    for (sa, oka), (sb, okb) in zip(out[:2], out[2:]):
        assert np.array_equal(oka, okb) and oka[5] == 0 and oka.sum() >= B - 2
        for x, y in zip(sa, sb):
            assert np.isfinite(np.asarray(y, np.float64)).all() and np.array_equal(x, y)
