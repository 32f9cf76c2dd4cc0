"""GPU suite for the launch policy (round 4): thresholds derived from the device, whole rounds as launches of their own past one
wave per SIMD, and the policy batch that makes the kernel-family choice independent of the shard layout.
No reference counterpart for any of it (the reference runs one filter on one thread, C++/src/filter.cpp:190-250); the arithmetic
that must not change is ImuUpdate.m:36-82 / MeasureUpdate.m:37-103."""
import os

import numpy as np
import pytest

from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import PLAIN_WINDOW_TOL, assert_parity

pytestmark = pytest.mark.gpu
DT = np.array([np.float64(np.float32(0.005))])


def _r32(a):
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


class _env:
    """environment knobs are read ONCE, at fbus_ekf_create: set them around the construction of a handle"""

    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _inputs(B, M, dialect=0, seed=0, with_cov=True):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(seed, seed + B, list(prm.p0_diag), 18, mixed_cov=with_cov, with_cov=with_cov)
    nom, rot = _r32(nom), _r32(rot)
    P = _r32(P) if P is not None else None
    acc, gyr = synth.imu_samples(seed, seed + B, 0, 3, nom)
    ids, pos, quat = synth.marker_frame(seed, seed + B, 0, M, nom, prm)
    return prm, nom, rot, P, prev, _r32(acc), _r32(gyr), ids, _r32(pos), _r32(quat)


def _run(flt, nom, rot, P, prev, acc, gyr, ids, pos, quat):
    flt.set_state(nom, rot, P, prev)
    if P is None:
        flt.reset_cov()
    for k in range(2):
        flt.predict(acc[k], gyr[k], DT)
    flt.correct(ids, pos, quat, capi.MODE_STACKED)
    flt.predict(acc[2], gyr[2], DT)
    return flt.get_state()


def test_launch_policy_follows_the_device_and_halves_with_half_the_simds():
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    prm = capi.default_params(0)
    with BatchedFilter(16384, prm) as flt:
        p = flt.launch_policy(M=4, K=7)
    assert p["simds"] == 4 * cus and p["one_round_filters"] == 256 * cus and p["two_wave_min_b"] == 256 * cus + 1
    assert p["policy_batch"] == 16384
    quarter = p["simds"] // 4
    # 16 384 filters = 256 tiles: a quarter of a 1024-SIMD chip -> the per-call predict and the measurement folds divide the work
    assert (p["roles_predict"], p["roles_meas"], p["team_frames"]) == ((3, 4, True) if 256 <= quarter else (1, 2 if 256 <= 2 * quarter else 1, 256 <= 2 * quarter))
    with _env(FBUS_FAKE_SIMDS=p["simds"] // 2):
        with BatchedFilter(16384, prm) as flt:
            h = flt.launch_policy(M=4, K=7)
    assert h["simds"] == p["simds"] // 2 and h["one_round_filters"] == p["one_round_filters"] // 2
    assert h["two_wave_min_b"] == p["one_round_filters"] // 2 + 1 and h["mall_MB"] == p["mall_MB"] // 2
    assert h["big_records_MB"] == p["big_records_MB"] // 2
    if p["simds"] == 1024:
        assert (h["roles_predict"], h["roles_meas"]) == (1, 2)          # 256 tiles are half of a 512-SIMD device
        with BatchedFilter(8192, prm) as flt:
            assert flt.launch_info(capi.INFO_ROLES_PREDICT, 1) == 3      # 128 tiles: a quarter of this device ...
        with _env(FBUS_FAKE_SIMDS=512):
            with BatchedFilter(8192, prm) as flt:
                assert flt.launch_info(capi.INFO_ROLES_PREDICT, 1) == 3  # ... and exactly a quarter of the halved one
            with BatchedFilter(8256, prm) as flt:
                assert flt.launch_info(capi.INFO_ROLES_PREDICT, 1) == 1


def test_parity_holds_with_half_the_simds():
    """FBUS_FAKE_SIMDS = half the device: 40 000 filters are then more than one wave per SIMD and take the two-wave kernel forms:
    the same results as the default policy to an ulp, and parity with the oracle on a strided subset"""
    B, M = 40000, 4
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = _inputs(B, M, with_cov=False)
    with BatchedFilter(B, prm) as flt:
        simds = flt.launch_info(capi.INFO_SIMDS)
        ref = _run(flt, nom, rot, None, prev, acc, gyr, ids, pos, quat)
    with _env(FBUS_FAKE_SIMDS=simds // 2):
        with BatchedFilter(B, prm) as flt:
            assert flt.launch_info(capi.INFO_TWO_WAVE_MIN_B) == simds // 2 * 64 + 1
            got = _run(flt, nom, rot, None, prev, acc, gyr, ids, pos, quat)
    for a, b in zip(got[:3], ref[:3]):       # the two-wave forms against the one-wave forms: equal to an ulp (see below)
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = np.maximum(np.abs(b).max(axis=tuple(range(1, b.ndim)), keepdims=True), 1.0)
        assert np.abs((a - b) / scale).max() <= 8 * 1.2e-7
    assert np.array_equal(got[3], ref[3])
    sub = np.arange(0, B, 977)
    eng = OracleEngine(len(sub), 0, 18)
    P0 = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (len(sub), 18, 18)).copy()
    eng.set_state(nom[sub], rot[sub], P0, prev[sub])
    for k in range(2):
        eng.predict(acc[k][sub], gyr[k][sub], DT)
    eng.correct(ids[sub], pos[sub], quat[sub], capi.MODE_STACKED)
    eng.predict(acc[2][sub], gyr[2][sub], DT)
    assert_parity([x[sub] for x in got], eng.get_state(), 32, "half the SIMDs", plain_tol=PLAIN_WINDOW_TOL)


@pytest.mark.parametrize("B", [73728, 70001])
def test_more_filters_than_one_wave_per_simd(B):
    """past one wave per SIMD the launcher takes the <= 256-register kernel forms (row-split correct): against the one-wave forms
    (FBUS_TWO_WAVE_MIN_B out of reach) -- the same operations in the same order, compiled apart: equal to an ulp -- and against the
    oracle on a subset; 70 001 filters: ragged last tile"""
    M = 4
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = _inputs(B, M, with_cov=False)
    res = {}
    for two in (1 << 30, None):
        with _env(**({"FBUS_TWO_WAVE_MIN_B": two} if two else {})):
            with BatchedFilter(B, prm) as flt:
                one_round = flt.launch_info(capi.INFO_ONE_ROUND_FILTERS)
                assert flt.launch_info(capi.INFO_TWO_WAVE_MIN_B) == (two if two else one_round + 1)
                res[two] = _run(flt, nom, rot, None, prev, acc, gyr, ids, pos, quat)
                assert (flt.applied() == 1).all()
    for name, a, b in zip(("nominal", "rot", "P"), res[1 << 30][:3], res[None][:3]):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = np.maximum(np.abs(b).max(axis=tuple(range(1, b.ndim)), keepdims=True), 1.0)
        print(f"[two-wave forms] {B} {name}: max |d| / scale {np.abs((a - b) / scale).max():.2e}, elements that differ {(a != b).mean():.4f}")
        assert np.abs((a - b) / scale).max() <= 8 * 1.2e-7
    assert np.array_equal(res[1 << 30][3], res[None][3])
    sub = np.concatenate([np.arange(0, B, 1499), np.arange(B - 3, B)])
    eng = OracleEngine(len(sub), 0, 18)
    P0 = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (len(sub), 18, 18)).copy()
    eng.set_state(nom[sub], rot[sub], P0, prev[sub])
    for k in range(2):
        eng.predict(acc[k][sub], gyr[k][sub], DT)
    eng.correct(ids[sub], pos[sub], quat[sub], capi.MODE_STACKED)
    eng.predict(acc[2][sub], gyr[2][sub], DT)
    assert_parity([x[sub] for x in res[None]], eng.get_state(), 32, f"{B} filters", plain_tol=PLAIN_WINDOW_TOL)


def test_shards_equal_the_single_handle_bit_for_bit_under_the_policy_batch():
    """24 576 filters as ONE handle (384 tiles: four-role predict_n and frame kernels, one-wave per-call predict) and as three shards
    of 8 192 (128 tiles each: left alone they would take the three-role per-call predict as well).  With the policy batch set to
    the whole job -- what fbus::ShardedFilter and bench.py --total-batch do -- every shard runs the single handle's kernels and the
    gathered result is bit-equal; without it the results agree to fp32 rounding only."""
    total, M, K = 24576, 4, 7
    prm, nom, rot, P, prev, acc3, gyr3, ids, pos, quat = _inputs(total, M, with_cov=False)
    acc, gyr = synth.imu_samples(0, total, 0, K, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    dts = np.full(K, DT[0])

    def job(flt, lo, hi):
        flt.set_state(nom[lo:hi], rot[lo:hi], None, prev[lo:hi])
        flt.reset_cov()
        flt.predict(acc[0][lo:hi], gyr[0][lo:hi], DT)
        flt.predict_n(np.ascontiguousarray(acc[:, lo:hi]), np.ascontiguousarray(gyr[:, lo:hi]), dts)
        flt.correct(ids[lo:hi], pos[lo:hi], quat[lo:hi], capi.MODE_STACKED)
        import torch
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        flt.frame(f32(acc[:3, lo:hi]), f32(gyr[:3, lo:hi]), f32(dts[:3]), torch.from_numpy(np.ascontiguousarray(ids[lo:hi])).cuda(),
                  f32(pos[lo:hi]), f32(quat[lo:hi]), capi.MODE_STACKED, fused=True)
        flt.sync()
        return flt.get_state()

    with BatchedFilter(total, prm) as flt:
        if flt.launch_info(capi.INFO_SIMDS) != 1024:
            pytest.skip("thresholds of the 1024-SIMD device")
        single = job(flt, 0, total)
        fam = (flt.launch_info(capi.INFO_ROLES_PREDICT, 1), flt.launch_info(capi.INFO_ROLES_PREDICT, K), flt.launch_info(capi.INFO_TEAM_FRAMES))
    assert fam == (1, 4, 1)
    for pinned in (True, False):
        parts = []
        for r in range(3):
            lo, hi = r * 8192, (r + 1) * 8192
            with BatchedFilter(hi - lo, prm) as flt:
                if pinned:
                    flt.set_policy_batch(total)
                    assert (flt.launch_info(capi.INFO_ROLES_PREDICT, 1), flt.launch_info(capi.INFO_ROLES_PREDICT, K),
                            flt.launch_info(capi.INFO_TEAM_FRAMES)) == fam
                else:
                    assert flt.launch_info(capi.INFO_ROLES_PREDICT, 1) == 3
                parts.append(job(flt, lo, hi))
        gathered = [np.concatenate([p[i] for p in parts]) for i in range(4)]
        if pinned:
            for a, b in zip(gathered, single):
                assert np.array_equal(a, b)
        else:
            d0 = np.abs(gathered[0].astype(np.float64) - single[0]).max()
            print(f"[shards] un-pinned shards against the single handle: max |d nominal| {d0:.2e}, bit-equal {np.array_equal(gathered[0], single[0])}")
            assert d0 < 1e-4


def test_shards_of_a_job_above_one_round_equal_the_single_handle_bit_for_bit():
    """(round 5, advisor) 73 728 filters are more than one wave per SIMD: the single handle runs the <= 256-register forms (row-split
    correct, parked predict_n, frame2_kernel).  Its two shards of 36 864 filters are each BELOW that threshold -- left alone they take the
    one-wave forms, which agree with the others to an ulp only.  With the policy batch set to the job's total the shards run the job's
    forms (LaunchPolicy::policy_b) and the gathered result is bit-equal."""
    import torch
    total, M, K = 73728, 4, 5
    prm, nom, rot, P, prev, acc3, gyr3, ids, pos, quat = _inputs(total, M, with_cov=False)
    acc, gyr = synth.imu_samples(0, total, 0, K, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    dts = np.full(K, DT[0])
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()

    def job(flt, lo, hi):
        flt.set_state(nom[lo:hi], rot[lo:hi], None, prev[lo:hi])
        flt.reset_cov()
        flt.predict_n(np.ascontiguousarray(acc[:, lo:hi]), np.ascontiguousarray(gyr[:, lo:hi]), dts)       # parked above one round
        flt.correct(ids[lo:hi], pos[lo:hi], quat[lo:hi], capi.MODE_STACKED)                                   # row-split above one round
        flt.frame(f32(acc[:3, lo:hi]), f32(gyr[:3, lo:hi]), f32(dts[:3]), torch.from_numpy(np.ascontiguousarray(ids[lo:hi])).cuda(),
                  f32(pos[lo:hi]), f32(quat[lo:hi]), capi.MODE_STACKED, fused=True)                            # frame2_kernel above one round
        flt.sync()
        return flt.get_state()

    with BatchedFilter(total, prm) as flt:
        if flt.launch_info(capi.INFO_SIMDS) != 1024:
            pytest.skip("thresholds of the 1024-SIMD device")
        assert total >= flt.launch_info(capi.INFO_TWO_WAVE_MIN_B)
        single = job(flt, 0, total)
    half = total // 2
    for pinned in (True, False):
        parts = []
        for lo, hi in ((0, half), (half, total)):
            with BatchedFilter(hi - lo, prm) as flt:
                assert hi - lo < flt.launch_info(capi.INFO_TWO_WAVE_MIN_B)
                if pinned:
                    flt.set_policy_batch(total)
                parts.append(job(flt, lo, hi))
        gathered = [np.concatenate([p[i] for p in parts]) for i in range(4)]
        if pinned:
            for name, a, b in zip(("nominal", "rot", "P", "prev"), gathered, single):
                assert np.array_equal(a, b), f"pinned shards differ from the single handle in {name}"
        else:
            same = all(np.array_equal(a, b) for a, b in zip(gathered, single))
            d = np.abs(gathered[2].astype(np.float64) - single[2]).max() / np.abs(single[2]).max()
            print(f"[shards] un-pinned shards of 73 728 filters against the single handle: bit-equal {same}, max |d P| / max |P| {d:.2e}")
            assert d < 1e-6              # (the one-wave forms against the <= 256-register forms: to an ulp, not to the bit)


def test_config4_partition_262144_filters_as_eight_shards_of_32768():
    """BASELINE config 4 -- 262 144 filters sharded over 8 GPUs, 200 Hz IMU + 30 Hz stereo, 4 markers -- as its PARTITION on one device
    (round 6; no 8-GPU box in any round): the eight contiguous shards of 32 768 filters (fbus_ekf/shard.py) run one after
    the other as eight handles with the policy batch set to the job's total, step 0.1 s of the mixed schedule each (7 / 7 / 6 per-call
    ImuUpdates, a stacked 4-marker MeasureUpdate behind each run: 23 EKF steps per filter, ImuUpdate.m:36-82 / MeasureUpdate.m:37-103)
    plus one fused camera frame, and are gathered by fbus_ekf_copy_records (the peer-copy form of the gather, what
    fbus::NodeFilter::gather_to issues per shard) into the record buffer of a whole-batch handle.
    Asserted: the gathered records == the single 262 144-filter handle's BIT FOR BIT (records, and the unpacked state -- after the first
    camera frame and at the end); on EVERY filter the posterior is finite, symmetric, has a positive diagonal and a unit quaternion;
    positive definite on a strided subset; the fp64 oracle on a strided subset through the free-running window gate, as two windows:
    the first camera frame from the initial state (8 steps), then the remaining 19 steps with the oracle re-seeded from the device's
    state behind that frame (re-seeded windows; in ONE 27-step window from the P0 diagonal the start-up transient read
    literal 1.02e-5 against 1e-5 -- 3.9 x the fp32-record floor, i.e. the kernels' own fp32 arithmetic, not gated away but split where
    the north star splits it).  What stays untested is the 8-GPU hardware run itself."""
    import torch
    from fbus_ekf import shard
    from util import assert_window_parity, parity_errors
    total, world, M = 262144, 8, 4
    pattern = (7, 7, 6)
    Kt = sum(pattern)
    prm = capi.default_params(0)
    nom, rot, _, prev = synth.initial_state(0, total, list(prm.p0_diag), 18, with_cov=False)
    nom, rot = _r32(nom), _r32(rot)
    acc, gyr = synth.imu_samples(0, total, 0, Kt, nom)
    acc, gyr = _r32(acc), _r32(gyr)
    frames = [synth.marker_frame(0, total, f, M, nom, prm) for f in range(len(pattern) + 1)]
    ids = np.stack([f[0] for f in frames]); pos = _r32(np.stack([f[1] for f in frames])); quat = _r32(np.stack([f[2] for f in frames]))
    dev = torch.device("cuda:0")
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_dt = torch.full((max(pattern),), float(DT[0]), dtype=torch.float32, device=dev)

    def job(flt, lo, hi):
        """-> the state behind the first camera frame (the end state stays in the handle)"""
        flt.set_state(nom[lo:hi], rot[lo:hi], None, prev[lo:hi])
        flt.reset_cov()
        a, g = f32(acc[:, lo:hi]), f32(gyr[:, lo:hi])
        i, p, q = torch.from_numpy(np.ascontiguousarray(ids[:, lo:hi])).to(dev), f32(pos[:, lo:hi]), f32(quat[:, lo:hi])
        torch.cuda.synchronize()
        k, first = 0, None
        for f, K in enumerate(pattern):                                  # per-call API: one launch per EKF step
            for j in range(K):
                flt.predict(a[k + j], g[k + j], d_dt[:1])
            flt.correct(i[f], p[f], q[f], capi.MODE_STACKED)
            if f == 0:
                first = flt.get_state()
            k += K
        flt.frame(a[:3], g[:3], d_dt[:3], i[3], p[3], q[3], capi.MODE_STACKED, fused=True)      # and one fused camera frame
        flt.sync()
        return first

    with BatchedFilter(total, prm) as single, BatchedFilter(total, prm) as whole:
        if single.launch_info(capi.INFO_SIMDS) != 1024:
            pytest.skip("thresholds of the 1024-SIMD device")
        assert total >= single.launch_info(capi.INFO_TWO_WAVE_MIN_B)                            # the job runs the <= 256-register forms
        ref_first = job(single, 0, total)
        ref = single.get_state()
        ptr, bpf, tot = single.records()
        assert bpf == 800 and tot == total * 800
        rec_single = torch.empty(tot, dtype=torch.uint8, device=dev)
        single.copy_records(rec_single, 0)
        single.sync()
        wptr, _, wtot = whole.records()
        sizes = shard.record_bytes_of_ranks(total, world, bpf)
        assert sum(sizes) == wtot == tot
        off = 0
        firsts = []
        for r in range(world):
            lo, hi = shard.shard_range(total, r, world)
            assert (lo, hi) == (r * 32768, (r + 1) * 32768) and sizes[r] == 32768 * 800
            with BatchedFilter(hi - lo, prm) as flt:
                assert hi - lo < flt.launch_info(capi.INFO_TWO_WAVE_MIN_B)                      # left alone a shard would take the one-wave forms
                flt.set_policy_batch(total)
                assert flt.launch_info(capi.INFO_POLICY_BATCH) == total
                firsts.append(job(flt, lo, hi))
                flt.copy_records(wptr, 0, byte_offset=off)
                flt.sync()
            off += sizes[r]
        rec_whole = torch.empty(tot, dtype=torch.uint8, device=dev)
        whole.copy_records(rec_whole, 0)
        whole.sync()
        assert torch.equal(rec_whole, rec_single), "gathered shard records differ from the single handle's"
        got = whole.get_state()
    for name, a, b in zip(("nominal", "rot", "P", "prev"), got, ref):
        assert np.array_equal(a, b), f"config 4 partition differs from the single handle in {name}"
    for i, name in enumerate(("nominal", "rot", "P", "prev")):
        assert np.array_equal(np.concatenate([f[i] for f in firsts]), ref_first[i]), f"... behind the first camera frame in {name}"
    # size-independent properties on EVERY filter
    g_nom, g_rot, g_P = got[0], got[1], got[2]
    assert np.isfinite(g_nom).all() and np.isfinite(g_rot).all() and np.isfinite(g_P).all()
    assert np.abs(np.linalg.norm(g_nom[:, 6:10].astype(np.float64), axis=1) - 1).max() < 1e-6
    assert np.array_equal(g_P, np.swapaxes(g_P, 1, 2))
    assert (np.einsum("bii->bi", g_P) > 0).all()
    sub = np.concatenate([np.arange(0, total, 1499), np.arange(total - 3, total), np.arange(32768 - 2, 32768 + 2)])
    Ps = g_P[sub].astype(np.float64)
    dg = np.sqrt(np.einsum("bii->bi", Ps))
    assert np.linalg.eigvalsh(Ps / (dg[:, :, None] * dg[:, None, :])).min() > 0

    # the oracle on the strided subset, two windows; each also with its RECORD rounded to fp32 after every step (the floor of
    # tests/util.py::assert_window_parity: what an exact-arithmetic filter with fp32 records loses)
    def oracle_run(start, steps, fp32_records):
        eng = OracleEngine(len(sub), 0, 18)
        eng.set_state(*start)

        def q():
            if fp32_records:
                eng.nominal[...] = _r32(eng.nominal); eng.rot[...] = _r32(eng.rot); eng.P[...] = _r32(eng.P)
        for kind, a in steps:
            if kind == "p":
                eng.predict(acc[a][sub], gyr[a][sub], DT)
            else:
                eng.correct(ids[a][sub], pos[a][sub], quat[a][sub], capi.MODE_STACKED)
            q()
        return eng.get_state()

    P0 = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (len(sub), 18, 18)).copy()
    w1 = [("p", j) for j in range(7)] + [("c", 0)]
    w2 = [("p", 7 + j) for j in range(7)] + [("c", 1)] + [("p", 14 + j) for j in range(6)] + [("c", 2)] + [("p", j) for j in range(3)] + [("c", 3)]
    start1 = (nom[sub], rot[sub], P0, prev[sub])
    r1 = oracle_run(start1, w1, False)
    assert_window_parity([x[sub] for x in ref_first], r1, f"config 4 partition, first camera frame, {len(sub)} of {total} filters", 0, 18,
                         floor=parity_errors(oracle_run(start1, w1, True), r1))
    start2 = tuple(np.asarray(x[sub], np.float64) if x.dtype.kind == "f" else x[sub] for x in ref_first)
    r2 = oracle_run(start2, w2, False)
    assert_window_parity([x[sub] for x in got], r2, f"config 4 partition, the remaining 19 steps (re-seeded), {len(sub)} of {total} filters", 0, 18,
                         floor=parity_errors(oracle_run(start2, w2, True), r2))
