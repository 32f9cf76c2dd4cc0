"""GPU suite: one camera frame with the NORTH STAR's MeasureUpdate in one launch (`fbus_ekf_frame_meas_fused_dev`,
csrc/ekf_meas.hpp::frame_meas_kernel): K ImuUpdates (matlab/ImuUpdate.m:36-82) + correct() from corner pixels / from stereo
corners (the rows of MeasureUpdate.m:67,72-73 with a corner in place of the marker origin, through the flat-port model of
vision.cpp:496-599) with the record resident in registers / LDS in between.

What is asserted:
  * the UPDATE of the fused kernel alone (K = 0) == fbus_ekf_correct_pixels_dev / _corners_dev BIT FOR BIT (the same fold and update
    functions on the same values: nominal state, carried rotation, covariance, previous marker id, applied flags);
  * fused == the per-call sequence (one fbus_ekf_predict_n_dev launch, or K fbus_ekf_predict_dev launches, + the per-call update) to
    fp32 ROUNDING, through the single-step gate of tests/util.py -- N = 18 and N = 15, left camera and stereo, corner rows stacked and
    nearest (C++ dialect: hysteresis), with filters that see nothing, only unknown ids, or are skipped.  NOT bit for bit: the K-step
    loop is the same device function (predict_steps) in all three kernels, but which product of an a b + c d the compiler contracts
    into an FMA differs from kernel to kernel (measured: velocities near zero and a few small covariance elements by 1-2 ulp of the
    largest; tools/r5_diag.py) -- as between the fused pose frame and its per-call sequence (tests/test_parity_gpu.py);
  * fused against the fp64 oracle through the free-running window gate (K + 1 steps without re-seeding; tests/util.py);
  * the routes behind the same entry point that do NOT take the fused kernel (fp64 records, the team forms of small launches,
    M = 0) give the per-call results too;
  * at the bench's size (65 536 filters): size-independent properties -- fused == per-call on a strided subset, symmetric positive
    definite posterior."""
import ctypes

import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import PLAIN_WINDOW_TOL, assert_parity, assert_window_parity, pixel_scene

pytestmark = pytest.mark.gpu
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
SIZE = 0.28
DT = np.array([0.005])


def _scene(B, M, dialect, n, seed):
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=seed, noise=5e-4, nominal=nom0)
    rng = np.random.default_rng(seed + 1)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))
    nom[:, 3:6] = rng.normal(0, 0.02, (B, 3))                        # slow: K predicts must not carry the markers out of view
    nom = r32(nom)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    prev = np.where(ids[:, 0] >= 0, ids[:, 0], 0).astype(np.int32)   # C++ dialect: the previously used marker is one in view
    return prm, nom, rot, r32(P), prev, ids, r32(left), r32(right)


def _dev(torch, dtype):
    dev = torch.device("cuda:0")
    tt = torch.float32 if dtype == 32 else torch.float64
    return lambda a: (torch.from_numpy(np.ascontiguousarray(a)).to(dev) if np.asarray(a).dtype.kind in "iu"
                      else torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev).to(tt))


CASES = [("pixels", False, capi.MODE_STACKED), ("pixels", True, capi.MODE_STACKED),
         ("corners", True, capi.MODE_STACKED), ("corners", True, capi.MODE_NEAREST)]


def _run(flt, fused, d, K, what, stereo, mode):
    """one frame through the fused entry point ("fused"), as ONE predict_n launch + one per-call update ("n"), or as K per-call
    predicts + one per-call update ("k")"""
    kind = capi.MEAS_PIXELS if what == "pixels" else capi.MEAS_CORNERS
    rgt = d["right"] if stereo else None
    if fused is True or fused == "fused":
        a, g, t = (d["acc"][:K], d["gyr"][:K], d["dt"][:K]) if K > 0 else (None, None, None)
        flt.frame_meas(a, g, t, d["ids"], d["left"], rgt, kind, capi.VIS_REFRACTIVE, mode, skip=d["skip"])
    else:
        if fused == "n":
            if K > 0:
                flt.predict_n(d["acc"][:K], d["gyr"][:K], d["dt"][:K])
        else:
            for k in range(K):
                flt.predict(d["acc"][k], d["gyr"][k], d["dt"][:1])
        if what == "pixels":
            flt.correct_pixels(d["ids"], d["left"], rgt, d["skip"])
        else:
            flt.correct_corners(d["ids"], d["left"], rgt, capi.VIS_REFRACTIVE, mode, d["skip"])
    flt.sync()
    return flt.get_state(), flt.applied()


@pytest.mark.parametrize("n", [18, 15])
@pytest.mark.parametrize("dialect", [0, 1])
def test_fused_frame_equals_the_per_call_sequence_and_the_oracle(dialect, n):
    import torch
    B, M, K = 448 - 5, 4, 3                                           # ragged last tile
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, dialect, n, seed=41 + dialect)
    ids[0] = -1                                                       # nothing visible: the predicted record
    ids[1, :] = 9                                                     # only ids outside the map
    skip = (np.arange(B) % 13 == 7).astype(np.uint8)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    acc, gyr = r32(acc), r32(gyr)
    dd = _dev(torch, 32)
    d = {"acc": dd(acc), "gyr": dd(gyr), "dt": dd(np.full(K, DT[0])), "ids": dd(ids), "left": dd(left), "right": dd(right),
         "skip": dd(skip)}
    vp = oc.vision_params()
    corners = np.zeros((B, M, 4, 3))
    for b in range(B):
        for m in range(M):
            if ids[b, m] >= 0:
                corners[b, m] = oc.refraction_triangulate(vp, left[b, m], right[b, m])
    for what, stereo, mode in CASES:
        with BatchedFilter(B, prm, nstate=n) as fa, BatchedFilter(B, prm, nstate=n) as fb, BatchedFilter(B, prm, nstate=n) as fc:
            for f in (fa, fb, fc):
                f.set_team(1, 1)                                      # one wave per tile: the forms a full-chip launch runs
                f.set_state(nom, rot, P, prev)
            sa, oka = _run(fa, "fused", d, K, what, stereo, mode)
            sb, okb = _run(fb, "n", d, K, what, stereo, mode)
            sc, okc = _run(fc, "k", d, K, what, stereo, mode)
        assert (oka == okb).all() and (oka == okc).all()
        # to fp32 rounding against both per-call sequences (two fp32 runs of the same K + 1 steps: the single-step gate, the plain
        # per-block figure with its chain bound as in smoke())
        for other, name in ((sb, "predict_n + update"), (sc, "K per-call predicts + update")):
            assert_parity(sa, other, 32, f"fused frame vs {name}, N={n} dialect {dialect} {what} {'stereo' if stereo else 'left'} mode {mode}",
                          plain_tol=PLAIN_WINDOW_TOL)
        # the update alone (K = 0) through the fused kernel: bit for bit the per-call update
        with BatchedFilter(B, prm, nstate=n) as fa, BatchedFilter(B, prm, nstate=n) as fb:
            for f in (fa, fb):
                f.set_team(1, 1)
                f.set_state(nom, rot, P, prev)
            s0, ok0 = _run(fa, "fused", d, 0, what, stereo, mode)
            s1, ok1 = _run(fb, "n", d, 0, what, stereo, mode)
        assert (ok0 == ok1).all()
        for x, y, name in zip(s0, s1, ("nominal", "rot", "P", "prev")):
            assert np.array_equal(x, y), f"{what} stereo={stereo} mode={mode}: fused update (K = 0) != per-call update in {name}"
        # ... and the oracle: K ImuUpdates + the update on the fp64 side -- a free-running window of K + 1 steps (the predicted position
        # reaches the update rounded to fp32, and 32-64 rows at sigma_pix = 1e-3 turn that 6e-8 m into 1e-5 of the velocity's scale:
        # measured sigma-aware 1.5e-5 in block v, literal 5e-7, block-wise covariance 3e-6 -- the per-call sequence reads the same)
        eng = OracleEngine(B, dialect, n)
        eng.set_state(nom, rot, P, prev)
        for k in range(K):
            eng.predict(acc[k], gyr[k], DT)
        keep = eng.get_state()
        if what == "pixels":
            ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, right if stereo else None, SIZE, prm.r_pix)
        else:
            ok = eng.orc.correct_corners(eng.nominal, eng.rot, eng.P, eng.prev, ids, corners, SIZE, mode)
        now = eng.get_state()
        for x, y in zip(now, keep):
            x[skip == 1] = y[skip == 1]
        eng.set_state(*now)
        ok[skip == 1] = 0
        assert (oka == ok).all() and ok[2:][skip[2:] == 0].all() and not ok[0] and not ok[1]
        assert_window_parity(sa, eng.get_state(), f"fused frame vs oracle N={n} dialect {dialect} {what} {'stereo' if stereo else 'left'} mode {mode}",
                             dialect, n)


def test_routes_that_do_not_take_the_fused_kernel():
    """fp64 records, the automatic team forms of a small launch (roles > 1) and M = 0 behind the same entry point: the per-call
    results (fp64 / M = 0: bit for bit; team forms: the same launches as the per-call route, so bit for bit as well)"""
    import torch
    B, M, K = 256, 4, 4
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, 18, seed=77)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    for dtype in (64, 32):
        dd = _dev(torch, dtype)
        d = {"acc": dd(acc), "gyr": dd(gyr), "dt": dd(np.full(K, DT[0])), "ids": dd(ids), "left": dd(left), "right": dd(right), "skip": None}
        with BatchedFilter(B, prm, dtype=dtype) as fa, BatchedFilter(B, prm, dtype=dtype) as fb:
            fa.set_state(nom, rot, P, prev); fb.set_state(nom, rot, P, prev)
            # the per-call side of this comparison is predict_n + the update: what the entry point documents for these routes
            fa.frame_meas(d["acc"], d["gyr"], d["dt"], d["ids"], d["left"], d["right"], capi.MEAS_PIXELS)
            fb.predict_n(d["acc"], d["gyr"], d["dt"])
            fb.correct_pixels(d["ids"], d["left"], d["right"])
            sa, sb = fa.get_state(), fb.get_state()
            for x, y in zip(sa, sb):
                assert np.array_equal(x, y)
            # M = 0: predicts only
            fa.frame_meas(d["acc"], d["gyr"], d["dt"], None, None, None, capi.MEAS_PIXELS)
            fb.predict_n(d["acc"], d["gyr"], d["dt"])
            for x, y in zip(fa.get_state(), fb.get_state()):
                assert np.array_equal(x, y)


@pytest.mark.parametrize("route", ["resident", "auto_team", "fp64"])
def test_rejected_calls_leave_the_state_alone(route):
    """every argument is checked before the first launch on EVERY route (advisor, round 5: the alignment check used to sit in the
    resident route only -- an fp64 handle and a small handle on the automatic team / split route ran the K predicts first)"""
    import torch
    B, M, K = 128, 2, 2
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, 18, seed=5)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    dtype = 64 if route == "fp64" else 32
    dd = _dev(torch, dtype)
    with BatchedFilter(B, prm, dtype=dtype) as flt:
        if route == "resident":
            flt.set_team(1, 1)
        elif route == "auto_team":      # 2 tiles: the divided-tail / team forms, i.e. predict_n + the per-call update behind the entry point
            assert flt.launch_info(capi.INFO_MEAS_SPLIT, M) > 0 or flt.launch_info(capi.INFO_ROLES_MEAS, M) > 1
        flt.set_state(nom, rot, P, prev)
        before = flt.get_state()
        lib, h = flt._lib, flt._h
        a, g, t, i, l = dd(acc), dd(gyr), dd(np.full(K, DT[0])), dd(ids), dd(left)
        es = dtype // 8
        p = lambda x: x.data_ptr()
        assert lib.fbus_ekf_frame_meas_fused_dev(h, K, p(a), p(g), p(t), 0, 7, M, p(i), p(l), None, 0, 1, None) == 4      # kind
        assert lib.fbus_ekf_frame_meas_fused_dev(h, K, p(a), p(g), p(t), 0, capi.MEAS_CORNERS, M, p(i), p(l), None, 0, 1, None) == 1   # no right
        assert lib.fbus_ekf_frame_meas_fused_dev(h, K, p(a), p(g), p(t), 0, capi.MEAS_PIXELS, M, p(i), p(l) + es, None, 0, 1, None) == 1   # alignment
        assert lib.fbus_ekf_frame_meas_fused_dev(h, K, p(a), p(g), p(t), 0, capi.MEAS_CORNERS, M, p(i), p(l), p(l) + es, 0, 1, None) == 1   # alignment (right)
        assert lib.fbus_ekf_frame_meas_fused_dev(h, K, p(a), None, p(t), 0, capi.MEAS_PIXELS, M, p(i), p(l), None, 0, 1, None) == 1
        # the window entry point: two frames, image points one element off
        kc = (ctypes.c_int32 * 2)(1, 1)
        l2 = dd(np.concatenate([left, left]))
        i2 = dd(np.concatenate([ids, ids]))
        assert lib.fbus_ekf_frames_meas_fused_dev(h, 2, kc, p(a), p(g), p(t), 0, capi.MEAS_PIXELS, M, p(i2), p(l2) + es, None, 0, 1, None) == 1
        flt.sync()
        for x, y in zip(flt.get_state(), before):
            assert np.array_equal(x, y)
        # and the aligned call goes through on the same handle
        assert lib.fbus_ekf_frames_meas_fused_dev(h, 2, kc, p(a), p(g), p(t), 0, capi.MEAS_PIXELS, M, p(i2), p(l2), None, 0, 1, None) == 0
        flt.sync()
        assert not np.array_equal(flt.get_state()[0], before[0])


def test_fused_frame_at_the_bench_size():
    """65 536 filters x 4 marker slots (the `fused_frame_pixels_m4` row of bench.py): the automatic policy takes the fused kernel
    here; fused == predict_n + the per-call update to fp32 rounding on every filter, posterior symmetric positive definite, everything finite"""
    import torch
    B, M, K = 65536, 4, 7
    prm = capi.default_params(0)
    prm.marker_size = 0.15
    nom, rot, ids, left, right = synth.pixel_wall_scene(B, M, prm, 0.15, seed=9, stereo=True)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    dd = _dev(torch, 32)
    d = {"acc": dd(acc), "gyr": dd(gyr), "dt": dd(np.full(K, DT[0])), "ids": dd(ids), "left": dd(left), "right": dd(right), "skip": None}
    out = {}
    for stereo in (False, True):
        with BatchedFilter(B, prm) as fa, BatchedFilter(B, prm) as fb:
            for f in (fa, fb):
                f.set_state(nom, rot, None, np.zeros(B, np.int32))
                f.reset_cov()
            assert fa.launch_info(capi.INFO_ROLES_MEAS, M) == 1 and fa.launch_info(capi.INFO_MEAS_SPLIT, M) == 0
            sa, oka = _run(fa, "fused", d, K, "pixels", stereo, capi.MODE_STACKED)
            sb, okb = _run(fb, "n", d, K, "pixels", stereo, capi.MODE_STACKED)
        assert oka.all() and okb.all()
        assert_parity(sa, sb, 32, f"fused frame vs predict_n + update at 65 536 filters, {'stereo' if stereo else 'left'}", plain_tol=PLAIN_WINDOW_TOL)
        Ps = sa[2][::97].astype(np.float64)
        assert np.isfinite(sa[0]).all() and np.isfinite(Ps).all() and np.array_equal(Ps, np.swapaxes(Ps, 1, 2))
        dg = np.sqrt(np.einsum("bii->bi", Ps))
        assert np.linalg.eigvalsh(Ps / (dg[:, :, None] * dg[:, None, :])).min() > 0


def test_cpp_mirror_reaches_the_new_entry_point(tmp_path):
    """include/fbus/batched_filter.hpp::frame_meas_fused_dev compiled with plain g++ against the device library: K predicts through
    the new entry point (M = 0: predicts only) equal fbus_ekf_predict_n_dev bit for bit, and a bad `kind` is refused by exception."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(ROOT, "fbus-ekf_amd", "lib")
    src = tmp_path / "fm.cpp"
    src.write_text(r'''
#include <fbus/batched_filter.hpp>
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
    using BF = fbus::BatchedFilter<float>;
    const int B = 300, K = 4;
    std::vector<float> a(size_t(K) * B * 3), w(size_t(K) * B * 3), dt(K, 0.005f);
    for (size_t i = 0; i < a.size(); ++i) { a[i] = 0.05f * float(i % 11) - 0.2f; w[i] = 0.002f * float(i % 7) - 0.004f; }
    float *da, *dw, *ddt;
    if (hipMalloc((void**)&da, a.size() * 4) != hipSuccess || hipMalloc((void**)&dw, w.size() * 4) != hipSuccess || hipMalloc((void**)&ddt, K * 4) != hipSuccess) return 2;
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddt, dt.data(), K * 4, hipMemcpyHostToDevice);
    std::vector<char> r1, r2;
    for (int pass = 0; pass < 2; ++pass) {
        BF f(B, BF::defaults(FBUS_DIALECT_MATLAB), 0);
        f.reset_covariance();
        if (pass == 0) f.frame_meas_fused_dev(K, da, dw, ddt, FBUS_MEAS_PIXELS, 0, nullptr, nullptr);
        else if (fbus_ekf_predict_n_dev(f.handle(), K, da, dw, ddt, 0) != FBUS_OK) return 3;
        f.sync();
        void* recs = nullptr; size_t tot = 0;
        fbus_ekf_records(f.handle(), &recs, nullptr, &tot);
        std::vector<char>& r = pass ? r2 : r1;
        r.resize(tot);
        hipMemcpy(r.data(), recs, tot, hipMemcpyDeviceToHost);
        if (pass == 1) {
            bool threw = false;
            try { f.frame_meas_fused_dev(K, da, dw, ddt, 7, 0, nullptr, nullptr); } catch (const std::exception&) { threw = true; }
            if (!threw) return 4;
        }
    }
    std::printf("records equal %d\n", int(r1.size() == r2.size() && std::memcmp(r1.data(), r2.data(), r1.size()) == 0));
    return (r1.size() == r2.size() && std::memcmp(r1.data(), r2.data(), r1.size()) == 0) ? 0 : 5;
}
''')
    exe = tmp_path / "fm"
    subprocess.run(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    str(src), "-o", str(exe), "-L", libdir, "-lfbus_ekf", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "records equal 1" in r.stdout


@pytest.mark.parametrize("what,stereo,mode", [("pixels", False, capi.MODE_STACKED), ("pixels", True, capi.MODE_STACKED),
                                              ("corners", True, capi.MODE_STACKED), ("corners", True, capi.MODE_NEAREST)])
@pytest.mark.parametrize("dialect,n", [(0, 18), (1, 18), (0, 15)])
def test_window_of_frames_with_the_north_star_update(dialect, n, what, stereo, mode):
    """fbus_ekf_frames_meas_fused_dev: a window of camera frames -- { K_f predicts, the pixel / corner update } x F with the record resident
    -- against the same frames as F launches of the frame form (the same kernel body, instantiated with and without the frame loop), and
    against the oracle through the window gate.  A frame without IMU samples, a filter that sees nothing in one frame, unknown ids in
    another, masked filters."""
    import torch
    B, M = 448 - 5, 4
    kcount = [3, 0, 2, 4]
    F, Kt = len(kcount), sum(kcount)
    prm, nom, rot, P, prev, ids0, left0, right0 = _scene(B, M, dialect, n, seed=61 + dialect)
    acc, gyr = synth.imu_samples(0, B, 0, Kt, nom)
    acc, gyr = r32(acc), r32(gyr)
    rng = np.random.default_rng(7)
    ids = np.stack([ids0] * F); left = np.stack([left0] * F); right = np.stack([right0] * F)
    left = r32(left + rng.normal(0, 2e-4, left.shape)); right = r32(right + rng.normal(0, 2e-4, right.shape))
    ids[1, 5] = -1
    ids[2, 6, :] = 9
    skip = np.zeros((F, B), np.uint8); skip[2, 11] = 1; skip[3, 12] = 1; skip[0, 13] = 1
    dd = _dev(torch, 32)
    d_acc, d_gyr, d_dt = dd(acc), dd(gyr), dd(np.full(Kt, DT[0]))
    d_ids, d_left, d_right, d_skip = dd(ids), dd(left), dd(right), dd(skip)
    kind = capi.MEAS_PIXELS if what == "pixels" else capi.MEAS_CORNERS
    with BatchedFilter(B, prm, nstate=n) as fa, BatchedFilter(B, prm, nstate=n) as fb:
        for f in (fa, fb):
            f.set_team(1, 1)
            f.set_state(nom, rot, P, prev)
        fa.frames_meas(kcount, d_acc, d_gyr, d_dt, d_ids, d_left, d_right if stereo else None, kind, capi.VIS_REFRACTIVE, mode, skip=d_skip)
        k0 = 0
        for f, K in enumerate(kcount):
            a, g, t = (d_acc[k0:k0 + K], d_gyr[k0:k0 + K], d_dt[k0:k0 + K]) if K else (None, None, None)
            fb.frame_meas(a, g, t, d_ids[f], d_left[f], d_right[f] if stereo else None, kind, capi.VIS_REFRACTIVE, mode, skip=d_skip[f])
            k0 += K
        fa.sync(); fb.sync()
        sa, sb = fa.get_state(), fb.get_state()
        assert np.array_equal(fa.applied(), fb.applied())
    # bit for bit: the window form and the frame form are one kernel body, with and without the frame loop (measured: equal in every case)
    for x, y, name in zip(sa, sb, ("nominal", "rot", "P", "prev")):
        assert np.array_equal(x, y), f"window != frame by frame in {name}: {what} stereo={stereo} mode={mode}"
    # the oracle, frame by frame -- twice: fp64 throughout, and fp64 arithmetic with the RECORD rounded to fp32 after every step (the floor
    # of tests/util.py::assert_window_parity: four reprojection updates 15-20 ms apart pin the position to ~1e-4 m and differentiate the
    # fp32 position's 6e-8 m quantum into the velocity -- an exact-arithmetic filter with fp32 records is 1e-4 (literal) off the fp64 run
    # here, and so is the kernel)
    vp = oc.vision_params()
    corners = None
    if what == "corners":
        corners = np.zeros((F, B, M, 4, 3))
        for f in range(F):
            for b in range(B):
                for m in range(M):
                    if ids[f, b, m] >= 0:
                        corners[f, b, m] = oc.refraction_triangulate(vp, left[f, b, m], right[f, b, m])

    def oracle_run(fp32_records):
        eng = OracleEngine(B, dialect, n)
        eng.set_state(nom, rot, P, prev)

        def q():
            if fp32_records:
                eng.nominal[...] = r32(eng.nominal); eng.rot[...] = r32(eng.rot); eng.P[...] = r32(eng.P)
        k0 = 0
        for f, K in enumerate(kcount):
            for k in range(K):
                eng.predict(acc[k0 + k], gyr[k0 + k], DT); q()
            k0 += K
            keep = eng.get_state()
            if what == "pixels":
                eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids[f], left[f], right[f] if stereo else None, SIZE, prm.r_pix)
            else:
                eng.orc.correct_corners(eng.nominal, eng.rot, eng.P, eng.prev, ids[f], corners[f], SIZE, mode)
            now = eng.get_state()
            for x, y in zip(now, keep):
                x[skip[f] == 1] = y[skip[f] == 1]
            eng.set_state(*now)
            q()
        return eng.get_state()

    ref = oracle_run(False)
    from util import parity_errors
    floor = parity_errors(oracle_run(True), ref)
    assert_window_parity(sa, ref, f"window of {F} frames vs oracle N={n} d{dialect} {what} {'stereo' if stereo else 'left'} mode {mode}",
                         dialect, n, floor=floor)
