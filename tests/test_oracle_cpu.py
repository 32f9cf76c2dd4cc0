"""CPU suite, part 1: the oracle itself.

 * the vision oracle against the reference's own recorded corners.txt -> image.txt
   data (the only sharp golden vectors the reference holds);
 * the C oracle against the independently written numpy twin and against the
   committed golden vectors;
 * algebraic invariants of the filter (symmetry, PSD, N=15 inside N=18,
   Joseph == simple, one stacked marker == nearest marker).
"""
import os

import numpy as np
import pytest

import ekf_oracle_np as onp
import oracle_capi as oc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- vision (pinned)
def test_vision_water_matches_reference_recording():
    d = np.load(os.path.join(GOLD, "vision_water.npz"))
    p = oc.vision_params()
    for c, im in zip(d["corners"], d["image"]):
        corners = oc.refraction_triangulate(p, c[2:10], c[10:18])
        pos, quat, _ = oc.marker_pose(corners)
        assert np.abs(pos - im[2:5]).max() < 1.5e-5          # files carry 6 significant digits
        assert min(np.abs(quat - im[5:9]).max(), np.abs(quat + im[5:9]).max()) < 5e-5


def test_vision_water_is_sharp_in_refraction_index():
    """n_water = 1.33 instead of 1.32 must visibly break the match (the fixture is sharp)."""
    d = np.load(os.path.join(GOLD, "vision_water.npz"))
    p = oc.vision_params()
    p.n_water = 1.33
    c, im = d["corners"][0], d["image"][0]
    pos, _, _ = oc.marker_pose(oc.refraction_triangulate(p, c[2:10], c[10:18]))
    assert np.abs(pos - im[2:5]).max() > 1e-3


def test_vision_land_pose_fit_matches_reference_recording():
    d = np.load(os.path.join(GOLD, "vision_land.npz"))
    for c, im in zip(d["corners"], d["image"]):
        pos, quat, rot = oc.marker_pose(c[2:14])
        assert np.abs(pos - im[2:5]).max() < 1.5e-5
        assert np.abs(quat - im[5:9]).max() < 5e-5
        assert np.abs(rot @ rot.T - np.eye(3)).max() < 1e-12


# ---------------------------------------------------------------- EKF oracle
def _np_states(nom, rot, P, prev, n):
    out = []
    for b in range(nom.shape[0]):
        s = onp.State(n)
        s.p, s.v, s.q = nom[b, 0:3].copy(), nom[b, 3:6].copy(), nom[b, 6:10].copy()
        s.ba, s.bg, s.g = nom[b, 10:13].copy(), nom[b, 13:16].copy(), nom[b, 16:19].copy()
        s.R, s.P, s.prev_id = rot[b].reshape(3, 3).copy(), P[b].copy(), int(prev[b])
        out.append(s)
    return out


@pytest.mark.parametrize("dialect", [0, 1])
@pytest.mark.parametrize("n", [18, 15])
def test_c_oracle_matches_golden(dialect, n):
    g = np.load(os.path.join(GOLD, "ekf_random.npz"))
    t = f"d{dialect}_n{n}"
    orc = oc.Oracle(dialect, n)
    # predict (filters 0/1 hit the w == 0 / small-rate cases; matlab dialect NaNs at w == 0 like the reference)
    nom, rot, P, prev = g[t + "_nom"].copy(), g[t + "_rot"].copy(), g[t + "_P"].copy(), g[t + "_prev"].copy()
    with np.errstate(all="ignore"):
        orc.predict(nom, rot, P, prev, g[t + "_acc"], g[t + "_gyr"], g[t + "_dt"])
    sl = slice(1, None) if dialect == 0 else slice(None)
    assert np.abs(nom[sl] - g[t + "_pred_nom"][sl]).max() < 1e-12
    assert np.abs(rot[sl] - g[t + "_pred_rot"][sl]).max() < 1e-12
    assert np.abs(P - g[t + "_pred_P"]).max() / np.abs(P).max() < 1e-12
    if dialect == 0:
        assert np.isnan(nom[0, 6:10]).all() and np.isnan(g[t + "_pred_nom"][0, 6:10]).all()
    for mode, name in ((oc.NEAREST, "near"), (oc.STACKED, "stack")):
        nom, rot, P, prev = g[t + "_nom"].copy(), g[t + "_rot"].copy(), g[t + "_P"].copy(), g[t + "_prev"].copy()
        ok = orc.correct(nom, rot, P, prev, g[t + "_ids"], g[t + "_pos"], g[t + "_quat"], mode)
        assert (ok == g[f"{t}_{name}_ok"]).all()
        assert ok[2] == 0 and ok[3] == 0                     # nothing visible / id outside the map
        assert mode == oc.NEAREST or ok[6] == 1              # stacked: an unknown id beside known ones is skipped
        assert (prev == g[f"{t}_{name}_prev"]).all()
        assert np.abs(nom - g[f"{t}_{name}_nom"]).max() < 1e-9
        assert np.abs(P - g[f"{t}_{name}_P"]).max() / np.abs(P).max() < 1e-10


@pytest.mark.parametrize("dialect", [0, 1])
def test_c_oracle_matches_numpy_twin_free_running(dialect):
    rng = np.random.default_rng(7)
    n, B = 18, 6
    orc = oc.Oracle(dialect, n)
    prm = onp.Params(dialect, n)
    g = np.load(os.path.join(GOLD, "ekf_random.npz"))
    t = f"d{dialect}_n{n}"
    nom, rot, P, prev = [g[t + k][4:4 + B].copy() for k in ("_nom", "_rot", "_P", "_prev")]
    sts = _np_states(nom, rot, P, prev, n)
    for step in range(12):
        acc = rng.normal(0, 0.5, (B, 3)) + [0, 9.8, 0]
        gyr = rng.normal(0, 0.02, (B, 3))
        orc.predict(nom, rot, P, prev, acc, gyr, np.array([0.005]))
        for b in range(B):
            onp.predict(sts[b], prm, acc[b], gyr[b], 0.005)
        if step % 3 == 2:
            ids = np.stack([rng.choice([0, 1, 2, 16, 17], 3, replace=False) for _ in range(B)]).astype(np.int32)
            pos = rng.normal(0, 0.5, (B, 3, 3))
            quat = rng.normal(size=(B, 3, 4))
            quat /= np.linalg.norm(quat, axis=2, keepdims=True)
            mode = (step // 3) % 2
            orc.correct(nom, rot, P, prev, ids, pos, quat, mode)
            for b in range(B):
                onp.correct(sts[b], prm, ids[b], pos[b], quat[b], mode)
    for b in range(B):
        s = sts[b]
        assert np.abs(np.concatenate([s.p, s.v, s.q, s.ba, s.bg, s.g]) - nom[b]).max() < 1e-9
        assert np.abs(s.P - P[b]).max() / np.abs(P[b]).max() < 1e-10
        assert s.prev_id == prev[b]


def test_land_slice_replay_matches_golden_and_stays_psd():
    """config 1 plumbing: the land recording through the C oracle with the FBUS_EKF.m loop."""
    d = np.load(os.path.join(GOLD, "land_slice.npz"))
    from replay_ref import replay_with_oracle
    for dialect, key in ((0, "states_matlab"), (1, "states_cpp")):
        states, npred = replay_with_oracle(d["imu"], d["image"], dialect, len(d[key]))
        assert (npred == d["npredict"]).all()
        assert np.abs(states[:, :20] - d[key][:, :20]).max() < 1e-9
        Pg = d[key][:, 29:].reshape(-1, 18, 18)
        Pm = states[:, 29:].reshape(-1, 18, 18)
        assert np.abs(Pm - Pg).max() / np.abs(Pg).max() < 1e-10
        for Pk in Pm:
            assert np.abs(Pk - Pk.T).max() == 0.0
            assert np.linalg.eigvalsh(Pk).min() > 0
        if dialect == 0:
            # loose sanity against the reference's recorded (older-revision) fusion.txt: gyro-bias band
            assert np.abs(states[-1, 14:17] - np.array([-0.0017, 0.00026, -0.00058])).max() < 2e-4


# ---------------------------------------------------------------- invariants
def _random_batch(n, B, seed, dialect):
    g = np.load(os.path.join(GOLD, "ekf_random.npz"))
    t = f"d{dialect}_n{n}"
    return [g[t + k][4:4 + B].copy() for k in ("_nom", "_rot", "_P", "_prev")] + \
           [g[t + k][4:4 + B].copy() for k in ("_acc", "_gyr", "_dt", "_ids", "_pos", "_quat")]


@pytest.mark.parametrize("dialect", [0, 1])
def test_n15_is_n18_without_gravity_uncertainty(dialect):
    nom, rot, P15, prev, acc, gyr, dt, ids, pos, quat = _random_batch(15, 8, 0, dialect)
    B = nom.shape[0]
    P18 = np.zeros((B, 18, 18))
    P18[:, :15, :15] = P15
    o15, o18 = oc.Oracle(dialect, 15), oc.Oracle(dialect, 18)
    a = [nom.copy(), rot.copy(), P15.copy(), prev.copy()]
    b = [nom.copy(), rot.copy(), P18.copy(), prev.copy()]
    for _ in range(3):
        o15.predict(*a, acc, gyr, dt)
        o18.predict(*b, acc, gyr, dt)
        o15.correct(*a, ids, pos, quat, oc.NEAREST)
        o18.correct(*b, ids, pos, quat, oc.NEAREST)
    assert np.abs(a[0] - b[0]).max() < 1e-12
    assert np.abs(a[2] - b[2][:, :15, :15]).max() < 1e-12
    assert np.abs(b[2][:, 15:, :]).max() < 1e-15


@pytest.mark.parametrize("dialect", [0, 1])
def test_joseph_equals_simple_form(dialect):
    nom, rot, P, prev, acc, gyr, dt, ids, pos, quat = _random_batch(18, 8, 0, dialect)
    a = [nom.copy(), rot.copy(), P.copy(), prev.copy()]
    b = [nom.copy(), rot.copy(), P.copy(), prev.copy()]
    oc.Oracle(dialect, 18, oc.SIMPLE).correct(*a, ids, pos, quat, oc.STACKED)
    oc.Oracle(dialect, 18, oc.JOSEPH).correct(*b, ids, pos, quat, oc.STACKED)
    assert np.abs(a[0] - b[0]).max() == 0
    assert np.abs(a[2] - b[2]).max() / np.abs(a[2]).max() < 1e-10


@pytest.mark.parametrize("dialect", [0, 1])
def test_one_stacked_marker_is_the_nearest_marker_update(dialect):
    nom, rot, P, prev, acc, gyr, dt, ids, pos, quat = _random_batch(18, 8, 0, dialect)
    ids1, pos1, quat1 = ids[:, :1].copy(), pos[:, :1].copy(), quat[:, :1].copy()
    ids1[:] = 16
    prev[:] = 16
    a = [nom.copy(), rot.copy(), P.copy(), prev.copy()]
    b = [nom.copy(), rot.copy(), P.copy(), prev.copy()]
    orc = oc.Oracle(dialect, 18)
    orc.correct(*a, ids1, pos1, quat1, oc.NEAREST)
    orc.correct(*b, ids1, pos1, quat1, oc.STACKED)
    assert np.abs(a[0] - b[0]).max() == 0 and np.abs(a[2] - b[2]).max() == 0


def test_predict_keeps_covariance_symmetric_psd_and_biases_constant():
    nom, rot, P, prev, acc, gyr, dt, *_ = _random_batch(18, 8, 0, 0)
    orc = oc.Oracle(0, 18)
    before = nom.copy()
    for _ in range(20):
        orc.predict(nom, rot, P, prev, acc, gyr, dt)
    assert np.abs(nom[:, 10:19] - before[:, 10:19]).max() == 0       # ba, bg, g untouched by predict
    assert np.abs(np.linalg.norm(nom[:, 6:10], axis=1) - 1).max() < 1e-12
    for Pk in P:
        assert np.abs(Pk - Pk.T).max() == 0
        assert np.linalg.eigvalsh(Pk).min() > 0


# ---------------------------------------------------------------- the stacked kernel's algebra
def _info_compressed_update(P, H, res, Rdiag, drop_tol=4e-15):
    """numpy restatement of what the stacked correct kernel does (ekf_device.hpp: InfoAcc / joint_update /
    scalar_update_info): fold the rows into the 6x6 information matrix of the (p, theta) columns, L D L', six scalar
    updates in information form.  Returns dx, P'."""
    J = [0, 1, 2, 6, 7, 8]
    assert not np.any(np.delete(H, J, axis=1)), "rows must live in the p / theta columns"
    n = P.shape[0]
    Lam = np.zeros((6, 6)); b = np.zeros(6)
    for h, r, Rk in zip(H, res, Rdiag):
        Lam += np.outer(h[J], h[J]) / Rk
        b += h[J] * r / Rk
    A, beta, d, L = Lam.copy(), b.copy(), np.zeros(6), np.eye(6)
    dg0 = np.diag(Lam).copy()
    for a in range(6):
        ok = A[a, a] > drop_tol * dg0[a]
        d[a] = A[a, a] if ok else 0.0
        if not ok:
            beta[a] = 0.0
        L[a + 1:, a] = A[a, a + 1:] / A[a, a] if ok else 0.0
        for i in range(a + 1, 6):
            A[i, i:] -= L[i, a] * A[a, i:]
            A[i:, i] = A[i, i:]
            beta[i] -= L[i, a] * beta[a]
    P = P.copy(); dx = np.zeros(n)
    for a in range(6):
        h = np.zeros(n); h[J] = L[:, a]
        Ph = P @ h
        is_ = 1.0 / (1.0 + d[a] * (h @ Ph))
        dx = dx + Ph * (is_ * (beta[a] - d[a] * (h @ dx)))
        P = P - np.outer(Ph * (d[a] * is_), Ph)
    return dx, P


@pytest.mark.parametrize("dialect", [0, 1])
@pytest.mark.parametrize("M", [1, 3, 12])
def test_information_compressed_update_is_the_dense_stacked_update(dialect, M):
    """K = P H'(H P H' + R)^-1 (MeasureUpdate.m:84-90 with all visible markers stacked) against the 6-update form."""
    rng = np.random.default_rng(11 + M)
    prm = onp.Params(dialect, 18)
    g = np.load(os.path.join(GOLD, "ekf_random.npz"))
    t = f"d{dialect}_n18"
    nom, rot, P, prev = [g[t + k][:8].copy() for k in ("_nom", "_rot", "_P", "_prev")]
    for s in _np_states(nom, rot, P, prev, 18):
        mids = rng.choice(sorted(prm.markers), M, replace=False)
        rows, res = [], []
        for mid in mids:
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            H, r = onp._rows(s, prm, int(mid), rng.normal(0, 0.5, 3), q)
            rows.append(H); res.append(r)
        H, r = np.vstack(rows), np.concatenate(res)
        Rd = np.tile([prm.r_pos] * 3 + [prm.r_quat] * 4, M)
        K = s.P @ H.T @ np.linalg.inv(H @ s.P @ H.T + np.diag(Rd))
        dx_ref, P_ref = K @ r, (np.eye(18) - K @ H) @ s.P
        dx, Pn = _info_compressed_update(s.P, H, r, Rd)
        assert np.abs(dx - dx_ref).max() <= 1e-9 * max(1.0, np.abs(dx_ref).max())
        assert np.abs(Pn - P_ref).max() <= 1e-9 * np.abs(P_ref).max()


def test_information_compressed_update_drops_directions_without_information():
    """rows that span only part of the (p, theta) space: the L D L' pivots of the missing directions are zero and must
    be no-ops (position rows of one marker only: rank 3 of 6)"""
    prm = onp.Params(0, 18)
    g = np.load(os.path.join(GOLD, "ekf_random.npz"))
    nom, rot, P, prev = [g["d0_n18" + k][:4].copy() for k in ("_nom", "_rot", "_P", "_prev")]
    for s in _np_states(nom, rot, P, prev, 18):
        H, r = onp._rows(s, prm, 16, np.array([0.1, -0.2, 0.8]), np.array([1.0, 0, 0, 0]))
        H, r = np.vstack([H[:3], H[:3]]), np.concatenate([r[:3], r[:3] + 0.01])      # duplicated rows: still rank 3
        Rd = np.full(6, prm.r_pos)
        K = s.P @ H.T @ np.linalg.inv(H @ s.P @ H.T + np.diag(Rd))
        dx, Pn = _info_compressed_update(s.P, H, r, Rd)
        assert np.isfinite(dx).all() and np.isfinite(Pn).all()
        assert np.abs(dx - K @ r).max() <= 1e-9 * max(1.0, np.abs(K @ r).max())
        assert np.abs(Pn - (np.eye(18) - K @ H) @ s.P).max() <= 1e-9 * np.abs(s.P).max()


# ---------------------------------------------------------------- the parity gate itself (tests/util.py)
def test_parity_gate_passes_fp32_rounding_and_fails_on_block_mutations():
    """CPU twin of the GPU mutation test: the gate accepts an fp32-rounded copy of an oracle state and rejects the
    same copy with any one 3x3 covariance block scaled by 2, 1 + 1e-3 or 0, any nominal block scaled by 1 + 1e-3,
    or an asymmetric covariance."""
    from fbus_ekf import capi, synth
    from replay_ref import OracleEngine
    from util import assert_parity
    B = 64
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    eng = OracleEngine(B, 0, 18)
    eng.set_state(nom, rot, P, prev)
    acc, gyr = synth.imu_samples(0, B, 0, 1, nom)
    ids, pos, quat = synth.marker_frame(0, B, 0, 4, nom, prm)
    eng.predict(acc[0], gyr[0], np.array([0.005]))
    eng.correct(ids, pos, quat, 1)
    ref = eng.get_state()
    got = [x.astype(np.float32).astype(np.float64) for x in ref[:3]] + [ref[3].copy()]
    got[2] = (got[2] + np.swapaxes(got[2], 1, 2)) / 2
    assert_parity(got, ref, 32, "fp32-rounded oracle state", verbose=False)
    for bi in range(6):
        for bj in range(bi, 6):
            for factor in (2.0, 1.0 + 1e-3, 0.0):
                mut = [x.copy() for x in got]
                blk = mut[2][:, 3 * bi:3 * bi + 3, 3 * bj:3 * bj + 3] * factor
                mut[2][:, 3 * bi:3 * bi + 3, 3 * bj:3 * bj + 3] = blk
                mut[2][:, 3 * bj:3 * bj + 3, 3 * bi:3 * bi + 3] = np.swapaxes(blk, 1, 2)
                with pytest.raises(AssertionError):
                    assert_parity(mut, ref, 32, f"block ({bi},{bj}) x {factor}", verbose=False)
    for a, b in ((0, 3), (3, 6), (6, 10), (10, 13), (13, 16), (16, 19)):
        mut = [x.copy() for x in got]
        mut[0][:, a:b] *= 1.0 + 1e-3
        with pytest.raises(AssertionError):
            assert_parity(mut, ref, 32, f"nominal [{a}:{b}]", verbose=False)
    mut = [x.copy() for x in got]
    mut[2][:, 0, 1] *= 1.0 + 1e-6
    with pytest.raises(AssertionError):
        assert_parity(mut, ref, 32, "asymmetric", verbose=False)
    mut = [x.copy() for x in got]
    mut[3][0] += 1
    with pytest.raises(AssertionError):
        assert_parity(mut, ref, 32, "prev id", verbose=False)


# ---------------------------------------------------------------- forward flat-port projection (pixel-row model)
def test_forward_projection_inverts_the_pinned_back_projection():
    """The pixel-row measurement model needs the forward projection the reference does not have.  The oracle's is pinned
    through the reference's own (recording-pinned) back-projection: a corner position triangulated from the water
    recording, projected into both cameras and triangulated again, comes back to 1e-12; single rays built exactly as
    RefractionTriangulation builds them (vision.cpp:505-552) project back onto their pixel to 1e-14."""
    d = np.load(os.path.join(GOLD, "vision_water.npz"))
    p = oc.vision_params()
    worst = 0.0
    for c in d["corners"]:
        X = oc.refraction_triangulate(p, c[2:10], c[10:18])
        uvL, uvR, ok = oc.project_stereo(p, X)
        assert ok.all()
        for k in range(4):
            X2 = oc.refraction_triangulate(p, np.tile(uvL[k], 4), np.tile(uvR[k], 4))[0]
            worst = max(worst, np.abs(X2 - X[k]).max())
        # the recorded pixels themselves are recovered to the skewness of the two recorded rays (the triangulated point is
        # their mid-point): a few 1e-3 in normalised coordinates, far below the 0.1 a wrong index of refraction gives
        assert np.abs(uvL.ravel() - c[2:10]).max() < 1e-2 and np.abs(uvR.ravel() - c[10:18]).max() < 1e-2
    assert worst < 1e-12
    # a sharpness check of the same kind as the back-projection's: n_water 1.33 instead of 1.32 moves the pixels visibly
    q = oc.vision_params()
    q.n_water = 1.33
    X = oc.refraction_triangulate(p, d["corners"][0][2:10], d["corners"][0][10:18])
    assert np.abs(oc.project_stereo(q, X)[0] - oc.project_stereo(p, X)[0]).max() > 1e-4
    # behind the port: refused
    assert not oc.project_stereo(p, [[0.0, 0.0, 0.01]])[2][0]


def test_pixel_rows_pull_a_displaced_state_back():
    """oracle-level sanity of fbo_correct_pixels: with measurements generated by projecting the TRUE corners, one update from
    a displaced state moves the position towards the truth, the posterior covariance is symmetric positive definite and smaller
    than the prior on the observed blocks, and markers that are not in the map contribute nothing."""
    from fbus_ekf import capi
    from util import pixel_scene
    B, M, size = 24, 3, 0.28
    prm = capi.default_params(1)
    orc = oc.Oracle(1, 18)
    p = oc.vision_params()
    rng = np.random.default_rng(4)
    truth, _, ids, left, right = pixel_scene(B, M, prm, size, seed=3)
    assert (ids[:, 0] >= 0).all() and (ids >= 0).sum() > B               # some filters see several markers
    P = np.broadcast_to(orc.P0(), (B, 18, 18)).copy()
    prev = np.zeros(B, np.int32)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.01, (B, 3))                           # displaced prior (sigma_p = 1e-2)
    from fbus_ekf import synth
    rot = synth.q2R(nom[:, 6:10]).reshape(B, 9)
    for stereo in (False, True):
        n2, r2, P2, pv = nom.copy(), rot.copy(), P.copy(), prev.copy()
        ok = orc.correct_pixels(n2, r2, P2, pv, ids, left, right if stereo else None, size, 1e-6, p)
        assert ok.all()
        before = np.linalg.norm(nom[:, 0:3] - truth[:, 0:3], axis=1)
        after = np.linalg.norm(n2[:, 0:3] - truth[:, 0:3], axis=1)
        assert np.median(after / before) < (0.5 if stereo else 0.8)
        assert np.abs(P2 - np.swapaxes(P2, 1, 2)).max() < 1e-18 and np.linalg.eigvalsh(P2).min() > 0
        assert (np.einsum("bii->bi", P2)[:, [0, 1, 2, 6, 7, 8]] < np.einsum("bii->bi", P)[:, [0, 1, 2, 6, 7, 8]]).all()
    ids9 = np.full((B, M), 9, np.int32)
    n2, r2, P2, pv = nom.copy(), rot.copy(), P.copy(), prev.copy()
    assert not orc.correct_pixels(n2, r2, P2, pv, ids9, left, right, size, 1e-6, p).any()
    assert np.array_equal(n2, nom) and np.array_equal(P2, P)


# ---------------------------------------------------------------- the reference's recorded output, as far as it goes
@pytest.mark.parametrize("dialect", [0, 1])
def test_land_replay_follows_the_recorded_fused_trajectory(dialect):
    """The one recorded OUTPUT of the reference (matlab/dataset/landdata/dataset-02/fusion.txt) comes from an older revision whose
    world frame is the first IMU pose and whose marker map was estimated on line, so absolute poses cannot be compared -- but the
    relative motion of the IMU can: over the 50 s recording (0.85 m excursion and back) the restated filter, replaying imu.txt /
    image.txt through the loop of FBUS_EKF.m, stays within 0.15 m and 8 degrees of the recorded trajectory, and the travelled
    distances agree to 0.8 ... 1.15.  Loose (two different filters on the same sensor data), but it ties the whole chain -- file
    formats, init, window rule, predict, correct -- to something the reference itself produced."""
    from fbus_ekf import capi, replay
    from replay_ref import OracleEngine
    d = np.load(os.path.join(GOLD, "recordings.npz"))
    eng = OracleEngine(1, dialect, 18)
    states, _ = replay.replay(eng, d["land_imu"], d["land_image"], capi.default_params(dialect))
    from util import relative_motion_gap
    dp, dr, exc, ratio = relative_motion_gap(states, d["land_fusion_pose"])
    print(f"land, dialect {dialect}: relative-motion gap to fusion.txt {dp:.3f} m / {dr:.1f} deg over a {exc:.2f} m excursion, distance ratio {ratio[0]:.2f}..{ratio[1]:.2f}")
    assert exc > 0.8 and dp < 0.15 and dr < 8.0 and 0.8 < ratio[0] and ratio[1] < 1.15


def test_water_replay_follows_the_recorded_fused_trajectory():
    """the same frame-independent comparison on the underwater recording (waterdata/dataset-06, 44 s, 0.6 m excursion), C++
    dialect (the one that uses the quaternion residual, as the program that recorded fusion.txt did): within 0.2 m / 10 deg"""
    from fbus_ekf import capi, replay
    from replay_ref import OracleEngine
    from util import relative_motion_gap
    d = np.load(os.path.join(GOLD, "recordings.npz"))
    eng = OracleEngine(1, 1, 18)
    states, _ = replay.replay(eng, d["water_imu"], d["water_image"], capi.default_params(1))
    dp, dr, exc, ratio = relative_motion_gap(states, d["water_fusion_pose"])
    print(f"water, C++ dialect: relative-motion gap to fusion.txt {dp:.3f} m / {dr:.1f} deg over a {exc:.2f} m excursion, distance ratio {ratio[0]:.2f}..{ratio[1]:.2f}")
    assert exc > 0.5 and dp < 0.2 and dr < 10.0 and 0.7 < ratio[0] and ratio[1] < 1.15


def test_reprojection_rows_explain_the_water_recording():
    """The north star's measurement model against the reference's own DATA (oracle only, CPU): the water recording's corners.txt replayed
    through FBUS_EKF.m's frame loop with every update taken from the corner pixels (flat-port forward projection, 16 rows per frame).
    At the filter's posterior the projected corners of the marker reproduce the RECORDED pixels to ~1 px rms, and the trajectory stays
    within centimetres of the pose-row replay of the same recording -- with the marker side the recording itself triangulates to
    (0.1142 m; with vision.hpp:114's 0.28 m the same replay leaves the tank: the residual is what tells)."""
    import oracle_capi as oc
    from fbus_ekf import capi, replay, synth
    from replay_ref import OracleEngine
    d = np.load(os.path.join(GOLD, "recordings.npz"))
    imu, image, corners = d["water_imu"], d["water_image"], d["water_corners"]
    vp = oc.vision_params()
    sides = []
    for r in corners[::25]:
        t = oc.refraction_triangulate(vp, r[2:10], r[10:18])
        sides += [np.linalg.norm(t[(k + 1) % 4] - t[k]) for k in range(4)]
    side = float(np.median(sides))
    assert abs(side - 0.1142) < 1.5e-3 and np.std(sides) < 3e-3

    def run(size, nfr):
        prm = capi.default_params(0)
        prm.marker_size = size
        eng = OracleEngine(1, 0, 18)
        eng.marker_size, eng.r_pix = size, prm.r_pix
        st, n = replay.replay(eng, imu, image, prm, max_frames=nfr, corners=corners)
        R_IL, P_IL, _ = synth.camera_constants(prm)
        mids, mpos, mquat = synth.marker_table(prm)
        c = np.array([[0, 0, 0], [0, size, 0], [size, size, 0], [size, 0, 0.0]])
        res = []
        for row in st:
            cr = corners[np.argmin(np.abs(corners[:, 0] - row[0]))]
            R0 = synth.q2R(row[7:11])
            cam = (R_IL @ (R0.T @ (mpos[0] + (synth.q2R(mquat[0]) @ c.T).T - row[1:4] - R0 @ P_IL).T)).T
            uvL, uvR, ok = oc.project_stereo(vp, cam)
            res.append(np.concatenate([uvL.ravel() - cr[2:10], uvR.ravel() - cr[10:18]]))
        pose, _ = replay.replay(OracleEngine(1, 0, 18), imu, image, prm, max_frames=nfr)
        return float(np.sqrt((np.array(res) ** 2).mean())), float(np.linalg.norm(st[:, 1:4] - pose[:, 1:4], axis=1).max()), st

    rms, gap, st = run(side, 150)
    print(f"[oracle] water recording through the pixel rows: marker side {side:.4f} m, reprojection residual {rms:.2e} rms, gap to the pose-row replay {gap:.3f} m")
    assert np.isfinite(st).all() and rms < 5e-3 and gap < 0.05
    rms_wrong, gap_wrong, _ = run(0.28, 30)
    assert rms_wrong > 10 * rms                      # the wrong marker side shows in the residual at once
