#!/usr/bin/env python3
"""Experiment (round 5): a full-chip batch of the VALU-bound fused kernels as TWO half-chip handles on two streams, half a period apart.
One round of 1024 waves runs in lock step (every wave loads, computes, stores at the same time); past the cache, where rounds overlap, the
same kernels are 20-25 % faster per filter.  Two handles of 32 768 filters (policy batch 65 536: the full-chip kernel forms, one wave per
tile, 512 SIMDs each) launched alternately put two waves of each kernel on every CU; with the second stream delayed once by half a frame
their memory phases fall under each other's arithmetic.
    python tools/exp_two_phase.py [--frames 300]
prints EKF steps/s of: one handle; two handles in phase; two handles, the second delayed by ~half a frame."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--batch", type=int, default=65536)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, M, PATTERN = args.batch, 4, (7, 7, 6)
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def scene(kind):
    prm = capi.default_params(0)
    if kind == "pose":
        nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, with_cov=False)
        acc, gyr = synth.imu_samples(0, B, 0, 7, nom)
        ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
        return prm, nom, rot, prev, acc, gyr, (ids, pos, quat)
    prm.marker_size = 0.15
    nom, rot, ids, left, right = synth.pixel_wall_scene(B, M, prm, 0.15, seed=9, stereo=True)
    acc, gyr = synth.imu_samples(0, B, 0, 7, nom)
    return prm, nom, rot, np.zeros(B, np.int32), acc, gyr, (ids, left, right)


def run(kind, parts, delay_us):
    prm, nom, rot, prev, acc, gyr, meas = scene(kind)
    n = B // parts
    flts, data, streams = [], [], []
    for p in range(parts):
        lo, hi = p * n, (p + 1) * n
        s = torch.cuda.Stream()
        f = BatchedFilter(n, prm, order_streams=False)
        f.set_policy_batch(B)
        f.set_stream(s)
        f.set_state(nom[lo:hi], rot[lo:hi], None, prev[lo:hi]); f.reset_cov()
        d = dict(acc=f32(acc[:, lo:hi]), gyr=f32(gyr[:, lo:hi]), dt=f32(np.full(7, 0.005)), ids=torch.from_numpy(np.ascontiguousarray(meas[0][lo:hi])).to(dev),
                 a=f32(meas[1][lo:hi]), b=f32(meas[2][lo:hi]))
        flts.append(f); data.append(d); streams.append(s)
    torch.cuda.synchronize()

    def frame(p, K):
        f, d = flts[p], data[p]
        if kind == "pose":
            f.frame(d["acc"][:K], d["gyr"][:K], d["dt"][:K], d["ids"], d["a"], d["b"], capi.MODE_STACKED, fused=True)
        else:
            f.frame_meas(d["acc"][:K], d["gyr"][:K], d["dt"][:K], d["ids"], d["a"], None, capi.MEAS_PIXELS)

    def go(nframes):
        for i in range(nframes):
            K = PATTERN[i % 3]
            for p in range(parts):
                frame(p, K)
    go(30)
    torch.cuda.synchronize()
    if parts > 1 and delay_us > 0:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(delay_us * 2400))           # ~2.4 GHz: cycles
    t0 = time.perf_counter()
    go(args.frames)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    steps = B * sum(PATTERN[i % 3] + 1 for i in range(args.frames))
    ok = all(np.isfinite(f.get_state()[0]).all() for f in flts)
    for f in flts:
        f.close()
    return steps / el, el / args.frames * 1e6, ok


for kind in ("pose", "pixels"):
    for parts, delay in ((1, 0), (2, 0), (2, 12), (2, 20), (2, 28), (4, 0)):
        v, us, ok = run(kind, parts, delay)
        print(f"fused {kind:6s} frame, {B} filters as {parts} handle(s), second stream delayed {delay:2d} us: {v:.4g} EKF steps/s, {us:6.1f} us per frame of the whole batch, finite {ok}", flush=True)
