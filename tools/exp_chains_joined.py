#!/usr/bin/env python3
"""Experiment: B filters as S launch chains (S handles of B/S filters, one stream each), per-call predict / correct launches,
(a) chains independent, (b) chains joined after every call (what a library-internal split of one handle would have to do:
every API call returns with all chains ordered behind it).  Prints EKF steps/s; S = 1 is the plain single handle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
PATTERN = (7, 7, 6)
B = int(os.environ.get("B", 1048576)); STEPS = int(os.environ.get("STEPS", 30))
dev = torch.device("cuda:0")
prm = capi.default_params(0)
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
for S, joined in ((1, False), (2, False), (2, True), (4, True), (1, False), (2, False), (2, True), (4, True)):
    hs, streams = [], [torch.cuda.Stream(dev) for _ in range(S)]
    for s in range(S):
        lo, hi = s * B // S, (s + 1) * B // S
        nom, rot, P, prev = synth.initial_state(lo, hi, list(prm.p0_diag), 18)
        acc, gyr = synth.imu_samples(lo, hi, 0, 20, nom)
        frames = [synth.marker_frame(lo, hi, f, 4, nom, prm) for f in range(3)]
        flt = BatchedFilter(hi - lo, prm)
        flt.order_streams = False
        flt.set_state(nom, rot, P, prev)
        flt.sync()
        flt.set_stream(streams[s])
        hs.append((flt, f32(acc), f32(gyr), [(torch.from_numpy(i).to(dev), f32(p), f32(q)) for i, p, q in frames]))
    d_dt = f32(np.full(1, 0.005))
    torch.cuda.synchronize()
    evs = [torch.cuda.Event() for _ in range(S)]
    def join():
        if not joined or S == 1: return
        for e, st in zip(evs, streams): e.record(st)
        for i, st in enumerate(streams):
            for j, e in enumerate(evs):
                if i != j: st.wait_event(e)
    def step():
        k = 0
        for f, K in enumerate(PATTERN):
            for kk in range(K):
                for flt, acc, gyr, frames in hs:
                    flt.predict(acc[k + kk], gyr[k + kk], d_dt)
                join()
            for flt, acc, gyr, frames in hs:
                ids, pos, quat = frames[f]
                flt.correct(ids, pos, quat, 1)
            join()
            k += K
    for _ in range(3): step()
    torch.cuda.synchronize()
    for h in hs: h[0]._keep.clear()
    t0 = time.perf_counter()
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"B {B} chains {S} {'joined per call' if joined else 'independent    '}: {B * 20 * STEPS / el:.4g} EKF steps/s, {el / STEPS * 1e3:.4f} ms per bench step", flush=True)
    for h in hs: h[0].close()
