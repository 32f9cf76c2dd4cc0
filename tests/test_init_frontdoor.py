"""Rows f-2 / f-4: gravity+bias init, pose init / reset / vision-only pose, IMU EMA pre-filter.
CPU part: the C oracle against the independent host implementation in fbus_ekf/replay.py and closed forms.
GPU part (-m gpu): the HIP kernels through the C ABI against the oracle."""
import os

import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import capi, replay, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _meas(B, M, dialect, seed=0):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(seed, seed + B, list(prm.p0_diag), 18)
    ids, pos, quat = synth.marker_frame(seed, seed + B, 0, M, nom, prm)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    return prm, r32(nom), r32(rot), ids, r32(pos), r32(quat)


# ------------------------------------------------------------------ CPU
def test_oracle_gravity_bias_is_the_mean_of_the_first_rows():
    d = np.load(os.path.join(GOLD, "land_slice.npz"))["imu"][:500]
    g, bg = oc.init_gravity_bias(d[:, 1:4], d[:, 4:7])
    g2, bg2 = replay.init_gravity_gyrobias(d)
    assert np.allclose(g, g2, atol=1e-13) and np.allclose(bg, bg2, atol=1e-16)
    assert g[0] == 0 and g[1] == 0 and abs(-g[2] - 9.8) < 0.05


@pytest.mark.parametrize("dialect", [0, 1])
def test_oracle_pose_init_matches_host_replay_and_inverts_the_measurement_model(dialect):
    B, M = 64, 3
    prm, nom, rot, ids, pos, quat = _meas(B, M, dialect)
    orc = oc.Oracle(dialect, 18)
    n2, r2 = nom.copy(), rot.copy()
    n2[:, 0:3] += 0.3                                   # wrong pose before init
    ok, _ = orc.pose_init(n2, r2, ids, pos, quat, 0, 0.0)      # 0.0: no range check (markers up to 4 m away)
    assert ok.all()
    for b in range(B):
        k = replay.nearest(np.concatenate([ids[b][:, None], pos[b], quat[b]], axis=1))
        p, q, R = replay.pose_from_marker(np.concatenate([[ids[b, k]], pos[b, k], quat[b, k]]), prm)
        # C++ dialect: Eigen's toRotationMatrix (1 - 2(y^2+z^2)) vs the Matlab formula in replay.py differ by the
        # non-unit part of the fp32-rounded measurement quaternion (~1e-7)
        tol = 1e-12 if dialect == 0 else 1e-6
        assert np.abs(n2[b, 0:3] - p).max() < tol and np.abs(n2[b, 6:10] - q).max() < 1e-12
        assert np.abs(r2[b].reshape(3, 3) - R).max() < tol
    # the synthetic measurement is h(x0) + 1e-3 noise, so init must land on x0
    assert np.abs(n2[:, 0:3] - nom[:, 0:3]).max() < 2e-2
    assert (n2[:, 16:19] == [9.8, 0, 0]).all()
    # reset: v, ba zeroed; C++ also zeroes bg and leaves R stale, Matlab refreshes R
    n3, r3 = nom.copy(), rot.copy() * 0 + 7.0
    ok, _ = orc.pose_init(n3, r3, ids, pos, quat, 1, 0.0)
    assert (n3[:, 3:6] == 0).all() and (n3[:, 10:13] == 0).all()
    if dialect == 1:
        assert (n3[:, 13:16] == 0).all() and (r3 == 7.0).all()
    else:
        assert (n3[:, 13:16] == nom[:, 13:16]).all() and np.abs(r3 - r2).max() < 1e-12
    # vision-only: normalised quaternion, state untouched
    n4, r4 = nom.copy(), rot.copy()
    ok, out7 = orc.pose_init(n4, r4, ids, pos, quat, 2, 0.0)
    assert np.array_equal(n4, nom) and np.abs(np.linalg.norm(out7[:, 3:], axis=1) - 1).max() < 1e-14
    assert np.abs(out7[:, :3] - n2[:, 0:3]).max() < 1e-6


def test_oracle_pose_init_refuses_far_and_unknown_markers():
    B, M = 8, 2
    prm, nom, rot, ids, pos, quat = _meas(B, M, 1)
    orc = oc.Oracle(1, 18)
    pos2 = pos / np.linalg.norm(pos, axis=2, keepdims=True)      # every marker at 1 m: inside marker_max_dist
    pos2[0] *= 100.0                  # beyond both max_dist and the 10 m threshold
    pos2[1] = pos[1] / np.linalg.norm(pos[1], axis=1, keepdims=True) * 3.0   # 3 m > marker_max_dist 2
    ids2 = ids.copy(); ids2[2] = 9
    mask = np.ones(B, np.uint8); mask[3] = 0
    n2, r2 = nom.copy(), rot.copy()
    ok, _ = orc.pose_init(n2, r2, ids2, pos2, quat, 0, 2.0, mask)
    assert list(ok[:4]) == [0, 0, 0, 0] and ok[4:].all()
    assert np.array_equal(n2[:4], nom[:4])
    assert oc.Oracle(0, 18).pose_init(nom.copy(), rot.copy(), ids, pos2, quat, 0, 2.0)[0][1] == 1   # Matlab: no range check


def test_oracle_ema_matches_the_recurrence_and_carries_across_calls():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(50, 6))
    y, c = oc.imu_ema(x)
    ref = x.copy()
    for t in range(1, 50):
        ref[t] = 0.9 * ref[t - 1] + 0.1 * x[t]
    assert np.abs(y - ref).max() < 1e-15 and np.array_equal(c, y[-1])
    y1, c1 = oc.imu_ema(x[:20])
    y2, _ = oc.imu_ema(x[20:], c1)
    assert np.abs(np.concatenate([y1, y2]) - ref).max() < 1e-15


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(64, 1e-12), (32, 2e-6)])
@pytest.mark.parametrize("dialect", [0, 1])
def test_gpu_init_reset_vision_only(dialect, dtype, tol):
    from fbus_ekf import BatchedFilter
    B, M = 300, 3
    prm, nom, rot, ids, pos, quat = _meas(B, M, dialect, seed=2000)
    ids[0] = -1
    ids[1] = 9
    pos[2] = pos[2] / np.linalg.norm(pos[2], axis=1, keepdims=True) * 3.0
    mask = (np.arange(B) % 5 != 4).astype(np.uint8)
    P = np.broadcast_to(np.eye(18), (B, 18, 18)).copy()
    prev = np.zeros(B, np.int32)
    orc = oc.Oracle(dialect, 18)
    for what in (0, 1):
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_state(nom, rot, P, prev)
            flt.pose_init(ids, pos, quat, what, mask)
            g_nom, g_rot, _, _ = flt.get_state()
            ap = flt.applied()
        o_nom, o_rot = nom.copy(), rot.copy()
        ok, _ = orc.pose_init(o_nom, o_rot, ids, pos, quat, what, 2.0, mask)
        assert (ap == ok).all()
        assert np.abs(g_nom - o_nom).max() < tol * 10 and np.abs(g_rot - o_rot).max() < tol * 10
    with BatchedFilter(B, prm, dtype=dtype) as flt:
        flt.set_state(nom, rot, P, prev)
        out = flt.vision_only_pose(ids, pos, quat)
        assert np.array_equal(flt.get_state()[0], nom.astype(flt.np_dtype))
    ok, o7 = orc.pose_init(nom.copy(), rot.copy(), ids, pos, quat, 2)
    assert np.abs(out[ok == 1] - o7[ok == 1]).max() < tol * 10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(64, 1e-13), (32, 2e-6)])
def test_gpu_gravity_bias_and_ema(dtype, tol):
    from fbus_ekf import BatchedFilter
    B, T = 200, 64
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    acc, gyr = synth.imu_samples(0, B, 0, T, nom)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    acc, gyr = r32(acc), r32(gyr)
    with BatchedFilter(B, prm, dtype=dtype) as flt:
        flt.set_state(nom, rot, P, prev)
        flt.init_gravity_bias(acc, gyr)
        g_nom = flt.get_state()[0]
        fa1, fg1 = flt.imu_ema(acc[:24], gyr[:24], restart=True)
        fa2, fg2 = flt.imu_ema(acc[24:], gyr[24:])
    for b in (0, 57, B - 1):
        g, bg = oc.init_gravity_bias(acc[:, b], gyr[:, b])
        assert np.abs(g_nom[b, 16:19] - g).max() < tol * 10 and np.abs(g_nom[b, 13:16] - bg).max() < tol
        y, _ = oc.imu_ema(np.concatenate([acc[:, b], gyr[:, b]], axis=1))
        got = np.concatenate([np.concatenate([fa1[:, b], fa2[:, b]]), np.concatenate([fg1[:, b], fg2[:, b]])], axis=1)
        assert np.abs(got - y).max() < tol * 10
    assert np.array_equal(g_nom[:, :13], nom[:, :13].astype(g_nom.dtype))      # only bg and g were written


def test_frame_batcher_sequences_like_the_reference_filter_thread(tmp_path):
    """include/fbus/frame_batcher.hpp (EMA against the last buffered sample, window start <= t <= end,
    dt from the state time, buffer trimming) against a Python restatement of filter.cpp:24-55,483-531."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    src = tmp_path / "fb.cpp"
    src.write_text(r'''
#include <fbus/frame_batcher.hpp>
#include <cstdio>
#include <vector>
struct Recorder {
    void predict(const double* a, const double* w, double dt) { std::printf("P %.9f %.9f %.9f\n", dt, a[0], w[5]); }
    void correct(int M, const int32_t* ids, const double*, const double*, int mode, const unsigned char*) {
        std::printf("C %d %d %d\n", M, ids[0], mode); }
};
int main() {
    Recorder r;
    fbus::FrameBatcher<double, Recorder> fb(r, 2, 0.0105, true, 40, 10);
    double t = 0.0;
    int k = 0;
    for (int frame = 0; frame < 6; ++frame) {
        const int n = (frame == 3) ? 55 : 9;              // frame 3 overflows the 40-sample buffer once
        for (int i = 0; i < n; ++i, ++k) {
            t += 0.001;
            double a[3] = { 0.1 * k, 1, 2 }, w[3] = { 3, 4, 0.01 * k };
            fb.set_imu(t, a, w);
        }
        int32_t ids[2] = { frame, 7 };
        double pos[6] = {0}, quat[8] = {1,0,0,0,1,0,0,0};
        int used = fb.on_detections(t - 0.0005, 2, ids, pos, quat, 1);
        std::printf("F %d %zu %.9f\n", used, fb.buffered(), fb.state_time());
    }
}
''')
    exe = tmp_path / "fb"
    subprocess.run(["g++", "-std=c++14", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()

    exp = []
    buf, t_state, t, k = [], 0.0105, 0.0, 0
    for frame in range(6):
        n = 55 if frame == 3 else 9
        for _ in range(n):
            t += 0.001
            a0, w2 = 0.1 * k, 0.01 * k
            k += 1
            if buf:
                a0 = buf[-1][1] * 0.9 + a0 * 0.1
                w2 = buf[-1][2] * 0.9 + w2 * 0.1
            buf.append((t, a0, w2))
            if len(buf) > 40:
                del buf[:10]
        end, used, consumed = t - 0.0005, 0, 0
        for (ts, a0, w2) in buf:
            if ts < t_state:
                consumed += 1
                continue
            if ts > end:
                break
            consumed += 1
            exp.append(f"P {ts - t_state:.9f} {a0:.9f} {w2:.9f}")
            t_state = ts
            used += 1
        del buf[:consumed]
        exp.append(f"C 2 {frame} 1")
        exp.append(f"F {used} {len(buf)} {t_state:.9f}")
    assert got == exp


@pytest.mark.gpu
def test_cpp_host_mirror_runs_the_filter_thread_on_the_gpu(tmp_path):
    """The C++ side of the boundary end to end on the device: fbus::BatchedFilter<float> + fbus::FrameBatcher (what a
    maintainer links into C++/src/filter.cpp) fed an IMU stream and two detection frames;
    the resulting state must equal the oracle's after the same sequence (EMA pre-filter, window rule, dt from the state
    time, nearest-marker correct with hysteresis) and must equal what the Python mirror produces call for call."""
    import subprocess
    from fbus_ekf import BatchedFilter, synth
    from replay_ref import OracleEngine
    from util import COV_BLOCK_TOL, COV_BLOCK_TOL_F64, COV_TOL, STATE_TOL, cov_rel_err, cov_rel_err_blockwise, state_rel_err
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    B, M, dialect = 64, 3, 1
    prm = capi.default_params(dialect)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32)
    nom, rot, P, prev = synth.initial_state(40, 40 + B, list(prm.p0_diag), 18, mixed_cov=True)
    nom, rot, P = r32(nom), r32(rot), r32(P)
    rng = np.random.default_rng(17)
    n_imu = 23
    imu_a = r32(rng.normal(0, 0.3, (n_imu, 3)) + [0.1, 9.7, 0.2])
    imu_w = r32(rng.normal(0, 0.02, (n_imu, 3)))
    ids, pos, quat = synth.marker_frame(40, 40 + B, 0, M, nom.astype(np.float64), prm)
    # one robot, B hypotheses: the front door hands every filter the same detections (those of filter 0)
    ids0, pos0, quat0 = ids[0].astype(np.int32), r32(pos[0]), r32(quat[0])
    for name, arr in (("nom", nom), ("rot", rot), ("P", P), ("prev", prev.astype(np.int32)), ("imu_a", imu_a),
                      ("imu_w", imu_w), ("ids", ids0), ("pos", pos0), ("quat", quat0)):
        arr.tofile(tmp_path / f"{name}.bin")
    src = tmp_path / "thread.cpp"
    src.write_text(r'''
#include <fbus/batched_filter.hpp>
#include <fbus/frame_batcher.hpp>
#include <cstdio>
#include <string>
#include <vector>
template <typename T> std::vector<T> rd(const std::string& p, size_t n) {
    std::vector<T> v(n); FILE* f = std::fopen(p.c_str(), "rb"); if (!f || std::fread(v.data(), sizeof(T), n, f) != n) std::abort();
    std::fclose(f); return v; }
int main(int argc, char** argv) {
    const std::string d = argv[1];
    const int B = 64, M = 3, NI = 23;
    using F = fbus::BatchedFilter<float>;
    F flt(B, FBUS_DIALECT_CPP);
    flt.set_team(1, 1); flt.set_team(0, 0);          // the launch policy through the C++ class (back to the automatic choice)
    auto nom = rd<float>(d + "/nom.bin", B * 19), rot = rd<float>(d + "/rot.bin", B * 9), P = rd<float>(d + "/P.bin", B * 324);
    auto prev = rd<int32_t>(d + "/prev.bin", B);
    auto ia = rd<float>(d + "/imu_a.bin", NI * 3), iw = rd<float>(d + "/imu_w.bin", NI * 3);
    auto ids = rd<int32_t>(d + "/ids.bin", M); auto pos = rd<float>(d + "/pos.bin", M * 3), quat = rd<float>(d + "/quat.bin", M * 4);
    flt.set_state(nom.data(), rot.data(), P.data(), prev.data());
    fbus::FrameBatcher<float, F> fb(flt, B, 0.0);
    fb.set_async(argc > 2);                          // (round 6) the same sequence through fbus_ekf_*_async: queued, not waited for
    double t = 0.0;
    int k = 0, used = 0;
    for (int frame = 0; frame < 2; ++frame) {
        for (int i = 0; i < (frame ? 11 : 12); ++i, ++k) { t += 0.005; fb.set_imu(t, &ia[3 * k], &iw[3 * k]); }
        used += fb.on_detections(t - 0.001, M, ids.data(), pos.data(), quat.data(), F::Mode::Nearest);
    }
    flt.get_state(nom.data(), rot.data(), P.data(), prev.data());
    FILE* f = std::fopen((d + "/out.bin").c_str(), "wb");
    std::fwrite(nom.data(), 4, nom.size(), f); std::fwrite(rot.data(), 4, rot.size(), f); std::fwrite(P.data(), 4, P.size(), f);
    std::fwrite(prev.data(), 4, prev.size(), f); std::fclose(f);
    std::printf("used %d buffered %zu\n", used, fb.buffered());
    return 0;
}
''')
    exe = tmp_path / "thread"
    libdir = os.path.dirname(capi.library_path())
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lfbus_ekf", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    # the asynchronous front door (FrameBatcher::set_async: fbus_ekf_predict_async / _correct_async per sample / frame): the same bytes
    ra = subprocess.run([str(exe), str(tmp_path), "async"], capture_output=True, text=True)
    assert ra.returncode == 0 and ra.stdout == r.stdout, ra.stdout + ra.stderr
    assert np.array_equal(np.fromfile(tmp_path / "out.bin", np.uint8), out.view(np.uint8)), "asynchronous front door != synchronous"
    c_nom, c_rot, c_P = out[:B * 19].reshape(B, 19), out[B * 19:B * 28].reshape(B, 9), out[B * 28:B * 352].reshape(B, 18, 18)
    c_prev = np.fromfile(tmp_path / "out.bin", np.int32)[B * 352:]

    # the same sequence, restated: EMA 0.1 against the last BUFFERED sample; window start <= t <= end; dt from the state time
    eng = OracleEngine(B, dialect, 18)
    eng.set_state(nom, rot, P, prev)
    tile = lambda v: np.tile(np.asarray(v, np.float64), (B, 1))
    with BatchedFilter(B, prm) as py:
        py.set_state(nom, rot, P, prev)
        buf, t_state, t, k, used = [], 0.0, 0.0, 0, 0
        for frame in range(2):
            for _ in range(11 if frame else 12):
                t += 0.005
                a, w = imu_a[k].copy(), imu_w[k].copy()
                k += 1
                if buf:
                    c = np.float32(0.1)
                    a = buf[-1][1] * (np.float32(1) - c) + a * c
                    w = buf[-1][2] * (np.float32(1) - c) + w * c
                buf.append((t, a, w))
            end, consumed = t - 0.001, 0
            for (ts, a, w) in buf:
                if ts < t_state:
                    consumed += 1
                    continue
                if ts > end:
                    break
                consumed += 1
                dt = np.float32(ts - t_state)
                eng.predict(tile(a), tile(w), np.array([float(dt)]))
                py.predict(tile(a), tile(w), float(dt))
                t_state = ts
                used += 1
            del buf[:consumed]
            ok = eng.correct(np.tile(ids0, (B, 1)), np.tile(pos0, (B, 1, 1)), np.tile(quat0, (B, 1, 1)), 0)
            py.correct(np.tile(ids0, (B, 1)), np.tile(pos0, (B, 1, 1)), np.tile(quat0, (B, 1, 1)), 0)
        p_nom, p_rot, p_P, p_prev = py.get_state()
    # (round 6) the Python twin of the front door with set_async(True): every step queued through fbus_ekf_*_async, bit-equal
    from fbus_ekf import FrameBatcher
    with BatchedFilter(B, prm) as pa:
        pa.set_state(nom, rot, P, prev)
        fbp = FrameBatcher(pa, B, 0.0)
        fbp.set_async(True)
        ta, ka = 0.0, 0
        for frame in range(2):
            for _ in range(11 if frame else 12):
                ta += 0.005
                fbp.set_imu(ta, imu_a[ka], imu_w[ka])
                ka += 1
            fbp.on_detections(ta - 0.001, ids0, pos0, quat0, 0)
        assert pa.async_stats()["calls"] == used + 2
        a_nom, _, a_P, a_prev = pa.get_state()
    assert np.array_equal(a_nom, p_nom) and np.array_equal(a_P, p_P) and np.array_equal(a_prev, p_prev)
    assert r.stdout.strip() == f"used {used} buffered {len(buf)}"
    # C++ mirror == Python mirror, call for call (same library, same launches)
    assert np.array_equal(c_nom, p_nom.astype(np.float32)) and np.array_equal(c_P, p_P.astype(np.float32))
    assert np.array_equal(c_prev, p_prev)
    # ... and both follow the oracle
    assert ok.all()
    assert state_rel_err(c_nom, eng.nominal, eng.P)[0] <= STATE_TOL * 3           # 22 predicts + 2 corrects
    assert cov_rel_err(c_P, eng.P) <= COV_TOL and cov_rel_err_blockwise(c_P, eng.P) <= COV_BLOCK_TOL
    assert (c_prev == eng.prev).all()


def test_python_frame_batcher_issues_the_same_calls_as_the_cpp_header(tmp_path):
    """fbus_ekf.FrameBatcher (Python twin) against include/fbus/frame_batcher.hpp driven with the same stream: the
    recorded predict / correct calls must be identical (EMA, window rule, dt, trimming)."""
    import subprocess
    from fbus_ekf import FrameBatcher
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    src = tmp_path / "fb.cpp"
    src.write_text(r'''
#include <fbus/frame_batcher.hpp>
#include <cstdio>
struct Recorder {
    void predict(const float* a, const float* w, float dt) { std::printf("P %.9g %.9g %.9g %.9g\n", dt, a[0], a[4], w[5]); }
    void correct(int M, const int32_t* ids, const float* pos, const float* quat, int mode, const unsigned char*) {
        std::printf("C %d %d %d %.9g %.9g\n", M, ids[0], mode, pos[4], quat[7]); }
    void correct_pixels(int M, const int32_t* ids, const float* left, const float* right, const unsigned char*) {
        std::printf("X %d %d %.9g %.9g %d\n", M, ids[1], left[16 + 9], right ? right[16 + 15] : -1.f, right != nullptr); }
};
int main() {
    Recorder r;
    fbus::FrameBatcher<float, Recorder> fb(r, 2, 0.0105, true, 40, 10);
    double t = 0.0;
    int k = 0;
    for (int frame = 0; frame < 6; ++frame) {
        const int n = (frame == 3) ? 55 : 9;
        for (int i = 0; i < n; ++i, ++k) {
            t += 0.001;
            float a[3] = { 0.1f * k, 1.f + 0.01f * k, 2 }, w[3] = { 3, 4, 0.01f * k };
            fb.set_imu(t, a, w);
        }
        int32_t ids[2] = { frame, 7 };
        float pos[6] = {0, 1, 2, 3, 4.5f, 5}, quat[8] = {1, 0, 0, 0, 0.5f, 0.5f, 0.5f, 0.25f * frame};
        float left[16], right[16];
        for (int i = 0; i < 16; ++i) { left[i] = 0.01f * i + 0.1f * frame; right[i] = -0.02f * i; }
        int used = (frame % 2) ? fb.on_corner_pixels(t - 0.0005, 2, ids, left, frame == 3 ? nullptr : right)
                               : fb.on_detections(t - 0.0005, 2, ids, pos, quat, 1);
        std::printf("F %d %zu %.9f\n", used, fb.buffered(), fb.state_time());
    }
}
''')
    exe = tmp_path / "fb"
    subprocess.run(["g++", "-std=c++14", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    want = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()

    got = []

    class Recorder:
        def predict(self, a, w, dt):
            got.append("P %.9g %.9g %.9g %.9g" % (np.float32(dt), a[0, 0], a[1, 1], w[1, 2]))

        def correct(self, ids, pos, quat, mode):
            got.append("C %d %d %d %.9g %.9g" % (ids.shape[1], ids[0, 0], mode, pos[0, 1, 1], quat[0, 1, 3]))

        def correct_pixels(self, ids, left, right=None):
            got.append("X %d %d %.9g %.9g %d" % (ids.shape[1], ids[0, 1], left[1, 1, 1], right[1, 1, 7] if right is not None else -1.0,
                                                 right is not None))

    fb = FrameBatcher(Recorder(), 2, 0.0105, True, 40, 10)
    t, k = 0.0, 0
    for frame in range(6):
        for _ in range(55 if frame == 3 else 9):
            t += 0.001
            fb.set_imu(t, np.array([np.float32(0.1) * np.float32(k), np.float32(1) + np.float32(0.01) * np.float32(k), 2], np.float32),
                       np.array([3, 4, np.float32(0.01) * np.float32(k)], np.float32))
            k += 1
        left = (np.float32(0.01) * np.arange(16, dtype=np.float32) + np.float32(0.1) * np.float32(frame)).reshape(2, 8)
        right = (np.float32(-0.02) * np.arange(16, dtype=np.float32)).reshape(2, 8)
        if frame % 2:
            used = fb.on_corner_pixels(t - 0.0005, [frame, 7], left, None if frame == 3 else right)
        else:
            used = fb.on_detections(t - 0.0005, [frame, 7], [[0, 1, 2], [3, 4.5, 5]],
                                    [[1, 0, 0, 0], [0.5, 0.5, 0.5, 0.25 * frame]], 1)
        got.append("F %d %d %.9f" % (used, fb.buffered, fb.t_state))
    assert got == want


def test_window_plan_is_the_matlab_frame_loop():
    """replay.plan_windows (the host-side plan behind replay_windowed / fbus_ekf_frames_fused_dev) against replay() itself on
    the recorded land sequence with two stretches of camera frames removed: the same frames, the same IMU samples with the
    same dt in front of each, resets where the script resets (FBUS_EKF.m:168-171), windows of at most 64 frames."""
    from replay_ref import OracleEngine
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "recordings.npz"))
    imu, image = d["land_imu"], d["land_image"]
    t = image[:, 0]
    image = image[~(((t > t[0] + 8.0) & (t < t[0] + 8.4)) | ((t > t[0] + 20.0) & (t < t[0] + 20.25)))]
    calls = []

    class Spy(OracleEngine):                    # records what replay() asks of the engine
        def predict(self, a, g, dt):
            calls.append(("p", float(np.ravel(dt)[0]), np.array(a).ravel().copy()))
            return super().predict(a, g, dt)

        def correct(self, ids, pos, quat, mode=0, skip=None):
            calls.append(("c", np.array(ids).ravel().copy()))
            return super().correct(ids, pos, quat, mode)

        def pose_init(self, ids, pos, quat, reset):
            if reset:
                calls.append(("r", np.array(ids).ravel().copy()))
            return super().pose_init(ids, pos, quat, reset)

    prm = capi.default_params(0)
    nfr = 700
    ref, npred = replay.replay(Spy(1, 0, 18), imu, image, prm, max_frames=nfr)
    plan = replay.plan_windows(imu, image, max_frames=nfr)
    flat = []
    for item in plan:
        if item[0] == "reset":
            flat.append(("r", item[1][:, 0].astype(np.int32)))
            continue
        _, kcount, rows, dts, frames = item
        assert 1 <= len(kcount) <= 64 and len(frames) == len(kcount) and kcount.sum() == len(rows) == len(dts)
        k0 = 0
        for f, K in enumerate(kcount):
            for k in range(K):
                flat.append(("p", float(dts[k0 + k]), imu[rows[k0 + k], 1:4]))
            flat.append(("c", frames[f][:, 0].astype(np.int32)))
            k0 += K
    assert len(flat) == len(calls) and sum(1 for c in calls if c[0] == "r") == 2
    for a, b in zip(flat, calls):
        assert a[0] == b[0]
        if a[0] == "p":
            assert a[1] == b[1] and np.array_equal(a[2], b[2])
        else:
            assert np.array_equal(a[1], b[1])
    assert sum(len(p[1]) for p in plan if p[0] == "window") + 2 == nfr == len(ref)


def test_window_plan_long_gap_and_too_many_markers():
    """A frame with more than 255 IMU samples in front of it (1 kHz IMU across a vision gap, C++-style loop without the Matlab
    reset) must not produce a window entry fbus_ekf_frames_fused_dev rejects (kcount is 0..255): the leading samples come
    back as a ("predict", rows, dts) item, every sample exactly once and in order; a frame with more than MAX_VISIBLE markers
    is a clear ValueError, not an opaque C error in the middle of a replay."""
    t_imu = np.arange(0, 1.0, 1e-3)
    imu = np.concatenate([t_imu[:, None], np.zeros((len(t_imu), 6))], axis=1)
    image = np.array([[0.05, 0, 0, 0, 1, 1, 0, 0, 0], [0.10, 0, 0, 0, 1, 1, 0, 0, 0],
                      [0.50, 0, 0, 0, 1, 1, 0, 0, 0], [0.55, 0, 0, 0, 1, 1, 0, 0, 0], [0.60, 0, 0, 0, 1, 1, 0, 0, 0]])
    plan = replay.plan_windows(imu, image, matlab_reset=False)
    seen = []
    for item in plan:
        if item[0] == "predict":
            assert len(item[1]) == len(item[2]) > 0
            seen += list(item[1])
        elif item[0] == "window":
            assert item[1].max() <= replay.MAX_KCOUNT
            seen += list(item[2])
    assert any(i[0] == "predict" for i in plan)
    assert seen == sorted(seen) and len(set(seen)) == len(seen)
    # the matlab loop resets across that gap instead: no oversize entry, no predict item
    plan_m = replay.plan_windows(imu, image, matlab_reset=True)
    assert not any(i[0] == "predict" for i in plan_m) and any(i[0] == "reset" for i in plan_m)
    crowded = np.array([[0.05, k, 0, 0, 1, 1, 0, 0, 0] for k in range(capi.MAX_VISIBLE + 1)] + [[0.1, 0, 0, 0, 1, 1, 0, 0, 0]])
    with pytest.raises(ValueError, match="markers"):
        replay.plan_windows(imu, crowded)
