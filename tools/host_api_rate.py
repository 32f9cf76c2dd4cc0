#!/usr/bin/env python3
"""Rate of the HOST-pointer entry points, PCIe inclusive (informational: DESIGN.md / INTEGRATION.md 1d; never bench `value`):
    sync    fbus_ekf_predict / fbus_ekf_correct        stage pageable inputs, launch, WAIT (what rounds 1-5 had)
    async   fbus_ekf_predict_async / _correct_async    inputs by value into a pinned ring, copy stream + kernel stream, no wait
            ... from pageable numpy arrays (the calling thread copies them into the ring) and from PINNED arrays (in-place DMA)
at 65 536 filters (BASELINE configs[1]) and at B = 1 (the reference's own case: one filter, one thread -- beside the CPU oracle's
single-thread time per step).  Caller: FILTER::SetImuData / BatchImuProcessing, C++/src/filter.cpp:24-55,505-516."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

M = 4
prm = capi.default_params(0)
dt = np.array([0.005], np.float32)


def run(B, reps):
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    acc, gyr = synth.imu_samples(0, B, 0, 20, nom)
    acc, gyr = acc.astype(np.float32), gyr.astype(np.float32)
    ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
    pos, quat = pos.astype(np.float32), quat.astype(np.float32)
    pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory()
    p_acc, p_gyr, p_ids, p_pos, p_quat = pin(acc), pin(gyr), pin(ids), pin(pos), pin(quat)
    out = {}
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        for k in range(3):
            flt.predict(acc[k], gyr[k], dt); flt.predict_async(acc[k], gyr[k], dt); flt.predict_async(p_acc[k], p_gyr[k], dt)
        flt.correct(ids, pos, quat, 1); flt.correct_async(ids, pos, quat, 1); flt.correct_async(p_ids, p_pos, p_quat, 1)
        flt.sync()

        def timed(fn, n):
            """best of three passes behind a warm-up pass (the first DMA out of freshly pinned ring pages is slow: the first pass of the
            first asynchronous leg read 272 us per call where the steady state is 57)"""
            best = None
            for rep in range(4):
                flt.sync()
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                t_issue = time.perf_counter() - t0
                flt.sync()
                r = (t_issue / n * 1e6, (time.perf_counter() - t0) / n * 1e6)      # host time per call, wall time per call incl. the drain
                if rep and (best is None or r[1] < best[1]):
                    best = r
            return best

        n = 20 * reps
        out["predict sync"] = timed(lambda i: flt.predict(acc[i % 20], gyr[i % 20], dt), n)
        out["predict async pageable"] = timed(lambda i: flt.predict_async(acc[i % 20], gyr[i % 20], dt), n)
        out["predict async pinned"] = timed(lambda i: flt.predict_async(p_acc[i % 20], p_gyr[i % 20], dt), n)
        out["correct sync"] = timed(lambda i: flt.correct(ids, pos, quat, 1), 4 * reps)
        out["correct async pageable"] = timed(lambda i: flt.correct_async(ids, pos, quat, 1), 4 * reps)
        out["correct async pinned"] = timed(lambda i: flt.correct_async(p_ids, p_pos, p_quat, 1), 4 * reps)
        st = flt.async_stats()
        # the raw C call without the Python wrapper's argument conversion (what a C++ caller pays)
        import ctypes as C
        lib, h = flt._lib, flt._h
        pa, pg, pd = [C.c_void_p(p_acc[k].data_ptr()) for k in range(20)], [C.c_void_p(p_gyr[k].data_ptr()) for k in range(20)], dt.ctypes.data_as(C.c_void_p)
        out["predict async pinned, bare C call"] = timed(lambda i: lib.fbus_ekf_predict_async(h, pa[i % 20], pg[i % 20], pd, 0), n)
        qa, qg = [acc[k].ctypes.data_as(C.c_void_p) for k in range(20)], [gyr[k].ctypes.data_as(C.c_void_p) for k in range(20)]
        out["predict async pageable, bare C call"] = timed(lambda i: lib.fbus_ekf_predict_async(h, qa[i % 20], qg[i % 20], pd, 0), n)
        out["predict sync, bare C call"] = timed(lambda i: lib.fbus_ekf_predict(h, qa[i % 20], qg[i % 20], pd, 0), n)
    return out, st


if __name__ == "__main__":
    for B, reps in ((65536, 10), (4096, 20), (1, 50)):
        out, st = run(B, reps)
        print(f"B = {B}   (ring: {st['calls']} async calls, {st['waits']} waited for a slot, {st['direct_pieces']} pieces in place)")
        for name, (issue, wall) in out.items():
            print(f"   {name:38s} host {issue:8.1f} us/call   wall {wall:8.1f} us/call  -> {B / wall * 1e6:.3g} EKF steps/s")
    try:
        import oracle_capi as oc
        oc.build(native=True)
        orc = oc.Oracle(1, 18, native=True, nthreads=1)
        nom, rot, P, prev = synth.initial_state(0, 1, list(capi.default_params(1).p0_diag), 18)
        acc, gyr = synth.imu_samples(0, 1, 0, 20, nom)
        t0 = time.perf_counter()
        for r in range(2000):
            orc.predict(nom, rot, P, prev, acc[r % 20], gyr[r % 20], np.array([0.005]))
        print(f"CPU oracle (the reference's arithmetic, one thread, B = 1, through ctypes): {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per predict")
    except Exception as e:
        print("CPU oracle leg skipped:", e)
