"""GPU suite for the team kernels (fbus-ekf_amd/csrc/ekf_team.hpp): several waves per 64-filter tile.

predict / predict_n run the SAME device functions on the same operands as the one-wave kernels; the results are equal up to
the compiler's FMA contraction, which picks a different product of an `a*b + c*d` to fuse in a few expressions of the
differently specialised kernels (measured: 1 ulp on 5 of the 171 covariance elements of 17 of 311 filters, 1 ulp on single
nominal components) -- asserted to ulp-level bounds.  (The team form of the pose-row correct -- P - W W' in one step, divided over
2-4 waves -- was removed in round 4: parity-green, slower than the one-wave kernel at every batch size, never selected.  Its
one-shot update lives on inside the team frame-window kernel, tested below.)  Reference operations: matlab/ImuUpdate.m:63-81, matlab/MeasureUpdate.m:71-102,
C++/src/filter.cpp:588-616,622-741."""
import numpy as np
import pytest

from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import assert_parity, assert_window_parity, parity_errors

pytestmark = pytest.mark.gpu
DT = np.array([np.float64(np.float32(0.005))])


def _r32(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def _batch(B, dialect, n, seed_off=0):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(seed_off, seed_off + B, list(prm.p0_diag), n, mixed_cov=True)
    return prm, _r32(nom), _r32(rot), _r32(P), prev


def _same(a, b, what, ulps=2.0, nominal_exact=True):
    """nominal state, rotation bit-equal (or to `ulps` fp32 ulps), marker id equal; covariance equal to `ulps` fp32 ulps of
    sqrt(P_ii P_jj)"""
    for x, y, name in zip((a[0], a[1]), (b[0], b[1]), ("nominal", "rot")):
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        ok = np.array_equal(x, y) if nominal_exact else np.allclose(x, y, rtol=ulps * 1.2e-7, atol=ulps * 1.2e-7)
        assert ok, f"{what}: {name} differs (max |d| {np.abs(x - y).max():.3g})"
    assert np.array_equal(a[3], b[3]), f"{what}: prev id"
    Pa, Pb = np.asarray(a[2], np.float64), np.asarray(b[2], np.float64)
    d = np.sqrt(np.abs(np.einsum("bii->bi", Pb)))
    rel = np.abs(Pa - Pb) / (d[:, :, None] * d[:, None, :])
    assert rel.max() <= ulps * 1.2e-7, f"{what}: covariance differs by {rel.max() / 1.2e-7:.2f} ulp of sqrt(P_ii P_jj)"
    return float((Pa != Pb).mean())


@pytest.mark.parametrize("n", [18, 15])
@pytest.mark.parametrize("dialect", [0, 1])
def test_team_predict_equals_one_wave_predict(dialect, n):
    """one ImuUpdate per launch: 2, 3 and 4 roles against the one-wave kernel, ragged batch, per-filter and scalar dt,
    several steps in a row (every stored byte of the record is compared through get_state)"""
    B = 5 * 64 - 9
    prm, nom, rot, P, prev = _batch(B, dialect, n)
    acc, gyr = synth.imu_samples(0, B, 0, 3, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    dtb = _r32(np.random.default_rng(3).uniform(0.001, 0.01, B))
    ref = None
    for roles in (1, 2, 3, 4):
        with BatchedFilter(B, prm, nstate=n) as flt:
            flt.set_team(roles, 1)
            flt.set_state(nom, rot, P, prev)
            flt.predict(acc[0], gyr[0], dtb)
            flt.predict(acc[1], gyr[1], DT)
            flt.predict(acc[2], gyr[2], DT)
            got = flt.get_state()
        if ref is None:
            ref = got
            eng = OracleEngine(B, dialect, n)
            eng.set_state(nom, rot, P, prev)
            eng.predict(acc[0], gyr[0], dtb); eng.predict(acc[1], gyr[1], DT); eng.predict(acc[2], gyr[2], DT)
            assert_parity(ref, eng.get_state(), 32, f"one-wave predict x3 dialect {dialect} N {n}", plain_tol=5e-3)
        else:
            _same(got, ref, f"predict, {roles} roles, dialect {dialect}, N {n}", ulps=4.0, nominal_exact=False)


@pytest.mark.parametrize("n", [18, 15])
@pytest.mark.parametrize("dialect", [0, 1])
def test_team_predict_n_equals_one_wave_predict_n(dialect, n):
    """K samples per launch with the rows handed on through LDS between the steps: K = 1, 2, 3, 8 against the one-wave predict_n
    and against K per-call predicts"""
    B = 3 * 64 + 5
    prm, nom, rot, P, prev = _batch(B, dialect, n, seed_off=11)
    for K in (2, 3, 8):
        acc, gyr = synth.imu_samples(11, 11 + B, 0, K, nom)
        acc, gyr = _r32(acc), _r32(gyr)
        dts = _r32(np.random.default_rng(K).uniform(0.002, 0.008, K))
        out = {}
        for roles in (1, 4):
            with BatchedFilter(B, prm, nstate=n) as flt:
                flt.set_team(roles, 1)
                flt.set_state(nom, rot, P, prev)
                flt.predict_n(acc, gyr, dts)
                out[roles] = flt.get_state()
        _same(out[4], out[1], f"predict_n K = {K}, dialect {dialect}, N {n}", ulps=2.0 * K, nominal_exact=False)
        with BatchedFilter(B, prm, nstate=n) as flt:
            flt.set_team(1, 1)
            flt.set_state(nom, rot, P, prev)
            for k in range(K):
                flt.predict(acc[k], gyr[k], dts[k:k + 1])
            per_call = flt.get_state()
        e = parity_errors(out[4], per_call)
        assert e["literal"] < 2e-6 and e["cov_block"] < 2e-6, (K, e)   # resident vs streamed: same functions, fp32 rounding of reloads only


def test_team_is_the_default_for_small_batches():
    """the launcher's choice: with the default setting a 4096-filter predict must give the team kernel's result (bit-equal to
    an explicit set_team; correct stays on the one-wave kernel) -- and the whole per-call frame (K predicts + correct) stays
    inside the parity gate"""
    B, M, n, dialect = 4096, 4, 18, 0
    prm, nom, rot, P, prev = _batch(B, dialect, n, seed_off=3)
    acc, gyr = synth.imu_samples(3, 3 + B, 0, 7, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    ids, pos, quat = synth.marker_frame(3, 3 + B, 0, M, nom, prm)
    pos, quat = _r32(pos), _r32(quat)
    out = {}
    for setting in ("default", "explicit"):
        with BatchedFilter(B, prm) as flt:
            if setting == "explicit":
                flt.set_team(3, 1)
            flt.set_state(nom, rot, P, prev)
            for k in range(7):
                flt.predict(acc[k], gyr[k], DT)
            flt.correct(ids, pos, quat, capi.MODE_STACKED)
            out[setting] = flt.get_state()
    for x, y in zip(out["default"], out["explicit"]):
        assert np.array_equal(x, y), "default policy at 4096 filters is not the team kernels"
    eng = OracleEngine(B, dialect, n)
    eng.set_state(nom, rot, P, prev)
    sub = slice(0, B, 37)
    eng = OracleEngine(len(range(B)[sub]), dialect, n)
    eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
    for k in range(7):
        eng.predict(acc[k][sub], gyr[k][sub], DT)
    eng.correct(ids[sub], pos[sub], quat[sub], capi.MODE_STACKED)
    got = tuple(x[sub] for x in out["default"])
    assert_parity(got, eng.get_state(), 32, "frame of 7 + 1 steps, team kernels, 4096 filters", plain_tol=5e-3)


@pytest.mark.parametrize("B", [4096, 16384, 32768, 65536])
def test_team_kernels_parity_at_the_config_batch_sizes(B):
    """the round-2 review's batch sizes: team predict (3 roles) and team predict_n (K = 7, 4 roles) FORCED at 4096 / 16 384 /
    32 768 / 65 536 filters with the (one-wave) correct behind them, same gate and same oracle as every other kernel, on a strided
    subset; and the whole batch against the one-wave kernels (every filter)"""
    dialect, n, M = 0, 18, 4
    prm, nom, rot, P, prev = _batch(B, dialect, n, seed_off=21)
    acc, gyr = synth.imu_samples(21, 21 + B, 0, 8, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    ids, pos, quat = synth.marker_frame(21, 21 + B, 0, M, nom, prm)
    pos, quat = _r32(pos), _r32(quat)
    sub = np.arange(0, B, max(1, B // 97))
    dts = np.full(7, DT[0])

    def run(team):
        out = []
        with BatchedFilter(B, prm, nstate=n) as flt:
            flt.set_team(*team)
            flt.set_state(nom, rot, P, prev)
            flt.predict(acc[0], gyr[0], DT)
            out.append(flt.get_state())
            flt.predict_n(acc[1:8], gyr[1:8], dts)
            out.append(flt.get_state())
            flt.correct(ids, pos, quat, capi.MODE_STACKED)
            out.append(flt.get_state())
            flt.set_team(team[0], 3 if team[1] > 1 else 1)
            flt.predict(acc[0], gyr[0], DT)
            flt.correct(ids, pos, quat, capi.MODE_NEAREST)
            out.append(flt.get_state())
        return out

    team, one = run((3, 4)), run((1, 1))
    eng = OracleEngine(len(sub), dialect, n)
    eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
    eng.predict(acc[0][sub], gyr[0][sub], DT)
    assert_parity([x[sub] for x in team[0]], eng.get_state(), 32, f"team predict, {B} filters")
    for k in range(1, 8):
        eng.predict(acc[k][sub], gyr[k][sub], DT)
    assert_parity([x[sub] for x in team[1]], eng.get_state(), 32, f"team predict_n K = 7, {B} filters", plain_tol=5e-3)
    eng.correct(ids[sub], pos[sub], quat[sub], capi.MODE_STACKED)
    assert_parity([x[sub] for x in team[2]], eng.get_state(), 32, f"team correct stacked, {B} filters", plain_tol=5e-3)
    eng.predict(acc[0][sub], gyr[0][sub], DT)
    eng.correct(ids[sub], pos[sub], quat[sub], capi.MODE_NEAREST)
    assert_parity([x[sub] for x in team[3]], eng.get_state(), 32, f"team correct nearest, {B} filters", plain_tol=5e-3)
    _same(team[0], one[0], f"team vs one-wave predict, {B} filters", nominal_exact=False)
    _same(team[1], one[1], f"team vs one-wave predict_n, {B} filters", ulps=16.0, nominal_exact=False)
    for i in (2, 3):
        e = parity_errors(team[i], one[i])
        # the two paths run different instruction streams through the 8 predicts (team: roles, the packed forms of fewer stages) and
        # through the update; each sits ~3e-6 from the oracle (lines above), so do they from each other
        assert e["literal"] < 5e-6 and e["sigma"] < 1e-5 and e["cov_block"] < 1e-5 and e["prev_equal"], (B, i, e)


def _window_inputs(B, dialect, n, M, kcount, seed_off=0):
    prm, nom, rot, P, prev = _batch(B, dialect, n, seed_off)
    Kt = sum(kcount)
    acc, gyr = synth.imu_samples(seed_off, seed_off + B, 0, Kt, nom)
    frames = [synth.marker_frame(seed_off, seed_off + B, f, M, nom, prm) for f in range(len(kcount))]
    ids = np.stack([f[0] for f in frames]); pos = _r32(np.stack([f[1] for f in frames])); quat = _r32(np.stack([f[2] for f in frames]))
    return prm, nom, rot, P, prev, _r32(acc), _r32(gyr), ids, pos, quat


@pytest.mark.parametrize("n", [18, 15])
@pytest.mark.parametrize("mode", [capi.MODE_NEAREST, capi.MODE_STACKED])
@pytest.mark.parametrize("dialect", [0, 1])
def test_team_frame_window_equals_one_wave_window(dialect, mode, n):
    """frames_team_kernel (predict_n pipeline on four roles, MeasureUpdate on the nominal role, rows handed over in LDS) against
    the one-wave frame-window kernel on the same window -- ragged batch, a frame without IMU samples, an invisible marker, an
    unknown marker id, masked filters -- to FMA-contraction level; the one-wave result goes through the oracle gate; and the same
    window as F single-frame launches of the team kernel (records through HBM instead of LDS in between) is bit-equal"""
    import torch
    B, M = 7 * 64 - 5, 4
    kcount = [7, 0, 6, 3]
    F, Kt = len(kcount), sum(kcount)
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = _window_inputs(B, dialect, n, M, kcount, seed_off=5)
    ids[1, 5] = -1
    ids[2, 6, :] = 9
    skip = np.zeros((F, B), np.uint8); skip[2, 11] = 1; skip[3, 12] = 1
    dev = torch.device("cuda:0")
    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_acc, d_gyr, d_dt = dd(acc), dd(gyr), dd(np.full(Kt, DT[0]))
    d_ids, d_pos, d_quat, d_skip = torch.from_numpy(ids).to(dev), dd(pos), dd(quat), torch.from_numpy(skip).to(dev)
    out, app = {}, {}
    for what in ("one-wave", "team", "team-by-frame"):
        with BatchedFilter(B, prm, nstate=n) as flt:
            flt.set_team(1 if what == "one-wave" else 4, 1)
            flt.set_state(nom, rot, P, prev)
            if what == "team-by-frame":
                k0 = 0
                for f, K in enumerate(kcount):
                    flt.frame(d_acc[k0:k0 + K] if K else None, d_gyr[k0:k0 + K] if K else None, d_dt[k0:k0 + K] if K else None,
                              d_ids[f], d_pos[f], d_quat[f], mode, skip=d_skip[f], fused=True)
                    k0 += K
            else:
                flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, mode, skip=d_skip)
            flt.sync()
            out[what], app[what] = flt.get_state(), flt.applied().copy()
    assert np.array_equal(app["team"], app["one-wave"]) and np.array_equal(app["team-by-frame"], app["one-wave"])
    for k in range(4):
        assert np.array_equal(out["team"][k], out["team-by-frame"][k]), f"window vs frame by frame, element {k}"
    eng = OracleEngine(B, dialect, n)
    eng.set_state(nom, rot, P, prev)
    k0 = 0
    for f, K in enumerate(kcount):
        for k in range(K):
            eng.predict(acc[k0 + k], gyr[k0 + k], DT)
        k0 += K
        ids_f = ids[f].copy(); ids_f[skip[f] == 1] = -1
        eng.correct(ids_f, pos[f], quat[f], mode)
    # the window gate of tests/util.py (literal 1e-5; C++ dialect with N = 18: the stated 5e-5, see util.py / profiles/r05_window_quantisation.txt)
    for what in ("one-wave", "team"):
        assert_window_parity(out[what], eng.get_state(), f"{what} frame window d{dialect} mode {mode} N {n}", dialect, n)
    # team against one-wave: contraction differences of single steps (1 ulp), carried through 16 ImuUpdates and 4 MeasureUpdates --
    # two fp32 runs differ by what the window gate allows either of them against the oracle
    assert_window_parity(out["team"], out["one-wave"], f"team vs one-wave frame window d{dialect} mode {mode} N {n}", dialect, n)


def test_team_frame_is_the_default_for_small_batches_and_long_windows():
    """the launcher's choice: up to 512 tiles the fused frame / frame window entry points run the team kernel (bit-equal to the
    forced team run); a full 64-frame window with per-filter dt, K up to 9, against the oracle on a strided subset"""
    import torch
    B, M, n, dialect = 16384, 4, 18, 0
    rng = np.random.default_rng(11)
    kcount = [int(x) for x in rng.integers(0, 10, 64)]
    F, Kt = len(kcount), sum(kcount)
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = _window_inputs(B, dialect, n, M, kcount, seed_off=9)
    dtb = _r32(rng.uniform(0.003, 0.007, (Kt, B)))
    dev = torch.device("cuda:0")
    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_acc, d_gyr, d_dt = dd(acc), dd(gyr), dd(dtb)
    d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), dd(pos), dd(quat)
    out = {}
    for what in ("default", "team"):
        with BatchedFilter(B, prm, nstate=n) as flt:
            if what == "team":
                flt.set_team(4, 1)
            flt.set_state(nom, rot, P, prev)
            flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, capi.MODE_STACKED)
            flt.sync()
            out[what] = flt.get_state()
            if what == "team":
                for _ in range(3):                                   # three more windows: 256 frames, ~1300 ImuUpdates in all
                    flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, capi.MODE_STACKED)
                flt.sync()
                long_run = flt.get_state()
    for k in range(4):
        assert np.array_equal(out["default"][k], out["team"][k]), "default policy at 16 384 filters is not the team frame kernel"
    sub = np.arange(0, B, B // 61)
    eng = OracleEngine(len(sub), dialect, n)
    eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
    k0 = 0
    for f, K in enumerate(kcount):
        for k in range(K):
            eng.predict(acc[k0 + k][sub], gyr[k0 + k][sub], dtb[k0 + k][sub])
        k0 += K
        eng.correct(ids[f][sub], pos[f][sub], quat[f][sub], capi.MODE_STACKED)
    assert_window_parity(tuple(x[sub] for x in out["team"]), eng.get_state(), f"team frame window, 64 frames / {Kt} steps", dialect, n)
    # the long run: the one-shot form P - W W' of the divided MeasureUpdate, 256 times in a row between ~1300 fp32 predicts, must
    # stay with the oracle and keep every covariance symmetric positive definite
    for _ in range(3):
        k0 = 0
        for f, K in enumerate(kcount):
            for k in range(K):
                eng.predict(acc[k0 + k][sub], gyr[k0 + k][sub], dtb[k0 + k][sub])
            k0 += K
            eng.correct(ids[f][sub], pos[f][sub], quat[f][sub], capi.MODE_STACKED)
    Pl = np.asarray(long_run[2], np.float64)
    min_eig = np.linalg.eigvalsh(Pl[::16]).min(axis=1)
    dmin = np.einsum("bii->bi", Pl[::16]).min(axis=1)
    print(f"[parity] team frame window, 256 frames: min eigenvalue / min diagonal over 1024 filters {np.min(min_eig / dmin):.2e}")
    assert np.isfinite(Pl).all() and (min_eig > 0).all()
    # 256 frames are beyond the 100-frame window the gate is stated for; the run meets it all the same (measured: literal 1e-6,
    # sigma-aware 1.3e-5, covariance 5e-6)
    assert_window_parity(tuple(x[sub] for x in long_run), eng.get_state(), f"team frame window, 256 frames / {4 * Kt} steps", dialect, n)


def test_team_frame_window_at_512_tiles_two_workgroups_per_cu():
    """32 768 filters = 512 tiles: the largest launch the automatic choice gives to frames_team_kernel, two workgroups per CU (80 KiB of
    LDS and 250 registers each).  Two runs bit-equal (a race between the roles' LDS phases would show as a difference), both modes,
    C++ dialect (the previous marker id travels through the image), and parity with the oracle on a strided subset"""
    import torch
    B, M, n, dialect = 32768, 4, 18, 1
    kcount = [7, 7, 6, 0, 5]
    F, Kt = len(kcount), sum(kcount)
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = _window_inputs(B, dialect, n, M, kcount, seed_off=17)
    dev = torch.device("cuda:0")
    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_acc, d_gyr, d_dt = dd(acc), dd(gyr), dd(np.full(Kt, DT[0]))
    d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), dd(pos), dd(quat)
    runs = []
    for _ in range(2):
        with BatchedFilter(B, prm, nstate=n) as flt:
            flt.set_state(nom, rot, P, prev)
            flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, capi.MODE_NEAREST)
            flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, capi.MODE_STACKED)
            flt.sync()
            runs.append(flt.get_state())
    for k in range(4):
        assert np.array_equal(runs[0][k], runs[1][k]), f"two runs of the team frame window differ (element {k})"
    sub = np.arange(0, B, B // 53)
    eng = OracleEngine(len(sub), dialect, n)
    eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
    for mode in (capi.MODE_NEAREST, capi.MODE_STACKED):
        k0 = 0
        for f, K in enumerate(kcount):
            for k in range(K):
                eng.predict(acc[k0 + k][sub], gyr[k0 + k][sub], DT)
            k0 += K
            eng.correct(ids[f][sub], pos[f][sub], quat[f][sub], mode)
    assert_window_parity(tuple(x[sub] for x in runs[0]), eng.get_state(), "team frame windows at 32 768 filters", dialect, n)
