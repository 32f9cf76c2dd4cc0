#!/usr/bin/env python3
"""Informational runs of the BASELINE.json configs 2, 3 and 5 on one MI355X (config 1 = CPU plumbing and
config 4 = the multi-GPU bench are covered by tests/ and bench.py).  Prints one line per case; the contract
line stays bench.py's.  Kernel times are HIP-event brackets (library timing API) around back-to-back launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

dev = torch.device("cuda:0")
prm = capi.default_params(0)


def tensors(a, tdt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(tdt)


def timed(flt, kind, fn, reps):
    for _ in range(3):
        fn()
    flt.sync()
    flt.timing_enable(True); flt.timing_reset()
    for _ in range(reps):
        fn()
    ms, n = flt.timing_read(kind)
    flt.timing_enable(False)
    return ms / n * 1e3


def run_pixels(B, dtype, M, stereo, reps):
    """the reprojection-row update on the 4 x 4 wall scene of the bench (its own parameter set: the scene replaces the marker map)"""
    tdt = torch.float32 if dtype == 32 else torch.float64
    es = 4 if dtype == 32 else 8
    prm2 = capi.default_params(capi.DIALECT_MATLAB)
    prm2.marker_size = 0.15
    nom, rot, ids, left, right = synth.pixel_wall_scene(B, M, prm2, 0.15, seed=9, stereo=True)
    d = (torch.from_numpy(ids).to(dev), tensors(left, tdt), tensors(right, tdt) if stereo else None)
    prev0 = np.zeros(B, np.int32)
    with BatchedFilter(B, prm2, dtype=dtype, order_streams=False) as flt:
        def step():
            flt.set_state(nom, rot, None, prev0); flt.reset_cov()
            flt.correct_pixels(d[0], d[1], d[2])
        for _ in range(2):
            step()
        flt.sync(); flt.timing_enable(True); flt.timing_reset()
        for _ in range(reps):
            step()
        ms, n = flt.timing_read(capi.KERNEL_CORRECT_CORNERS)
        flt.timing_enable(False)
        roles = flt.launch_info(capi.INFO_ROLES_MEAS, M)
    us = ms / n * 1e3
    nrec = 28 + 171
    bytes_ = (2 * nrec * es + (1 + 8 * (2 if stereo else 1)) * es * M) * B          # record round trip + ids + image points
    return us, B / us * 1e6, bytes_ / us * 1e-3, roles


def run(B, dtype, M, what, K=1, mode=1, reps=200, n=18, stereo=False):
    if what == "correct_pixels":
        return run_pixels(B, dtype, M, stereo, min(reps, 12))
    tdt = torch.float32 if dtype == 32 else torch.float64
    es = 4 if dtype == 32 else 8
    nrec = 28 + n * (n + 1) // 2                    # record elements priced by SURVEY 8(d): nominal + R + packed covariance
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n)
    acc, gyr = synth.imu_samples(0, B, 0, max(K, 1), nom)
    d_acc, d_gyr, d_dt = tensors(acc, tdt), tensors(gyr, tdt), tensors(np.full(max(K, 1), 0.005), tdt)
    with BatchedFilter(B, prm, dtype=dtype, nstate=n, order_streams=False) as flt:
        flt.set_state(nom, rot, P, prev)
        if what == "predict":
            us = timed(flt, capi.KERNEL_PREDICT, lambda: flt.predict(d_acc[0], d_gyr[0], d_dt[:1]), reps)
            steps, bytes_ = B, (2 * nrec * es + 7 * es) * B
        elif what == "predict_n":
            us = timed(flt, capi.KERNEL_PREDICT_N, lambda: flt.predict_n(d_acc, d_gyr, d_dt), reps)
            steps, bytes_ = B * K, (2 * nrec * es + 7 * es) * B * K
        elif what == "correct_corners":
            # north-star B2: stereo corner pixels -> refractive triangulation -> 12 rows per marker, on the device
            ids, _, _ = synth.marker_frame(0, B, 0, min(M, 12), nom, prm)
            if M > 12:
                ids = np.concatenate([ids, np.full((B, M - 12), -1, np.int32)], axis=1)
            c = np.load(os.path.join(ROOT, "tests", "golden", "vision_water.npz"))["corners"]
            base = c[np.random.default_rng(1).integers(0, len(c), B * M)]
            left, right = base[:, 2:10].reshape(B, M, 8), base[:, 10:18].reshape(B, M, 8)
            d = (torch.from_numpy(ids).to(dev), tensors(left, tdt), tensors(right, tdt))
            us = timed(flt, capi.KERNEL_CORRECT_CORNERS,
                       lambda: flt.correct_corners(d[0], d[1], d[2], capi.VIS_REFRACTIVE, mode), reps)
            steps, bytes_ = B, (2 * nrec * es + 17 * es * M) * B
        else:
            ids, pos, quat = synth.marker_frame(0, B, 0, min(M, 12), nom, prm)
            if M > 12:
                pad = M - 12
                ids = np.concatenate([ids, np.full((B, pad), -1, np.int32)], axis=1)
                pos = np.concatenate([pos, np.zeros((B, pad, 3))], axis=1)
                quat = np.concatenate([quat, np.tile([1.0, 0, 0, 0], (B, pad, 1))], axis=1)
            d = (torch.from_numpy(ids).to(dev), tensors(pos, tdt), tensors(quat, tdt))
            us = timed(flt, capi.KERNEL_CORRECT, lambda: flt.correct(d[0], d[1], d[2], mode), reps)
            steps, bytes_ = B, (2 * nrec * es + 8 * es * M) * B
        roles = {"predict": flt.launch_info(capi.INFO_ROLES_PREDICT, 1), "predict_n": flt.launch_info(capi.INFO_ROLES_PREDICT, K),
                 "correct_corners": flt.launch_info(capi.INFO_ROLES_MEAS, M) if mode == 1 else 1}.get(what, 1)
    return us, steps / us * 1e6, bytes_ / us * 1e-3, roles


import json
CASES = []
print("config  case                                             us/launch     EKF steps/s   algorithmic GB/s   frac of 8 TB/s   waves/tile   (HIP-event bracket per launch: +2-3 us)")
for name, args in (
    ("2", dict(B=4096, dtype=32, M=0, what="predict")),
    ("2", dict(B=4096, dtype=32, M=0, what="predict_n", K=8)),
    ("3", dict(B=16384, dtype=32, M=4, what="correct", mode=0)),
    ("3", dict(B=16384, dtype=32, M=4, what="correct", mode=1)),
    ("3", dict(B=16384, dtype=32, M=4, what="correct_corners", mode=1)),
    # config 3 as the north star words it: stereo 4-marker MeasureUpdate through the flat-port refraction model (reprojection rows)
    ("3", dict(B=16384, dtype=32, M=4, what="correct_pixels", stereo=True)),
    ("3", dict(B=16384, dtype=32, M=4, what="correct_pixels", stereo=False)),
    ("5", dict(B=65536, dtype=32, M=16, what="correct", mode=1, reps=50)),
    ("5", dict(B=65536, dtype=32, M=16, what="correct_corners", mode=1, reps=30)),
    ("5", dict(B=65536, dtype=32, M=16, what="correct_pixels", stereo=False)),
    ("5", dict(B=65536, dtype=32, M=16, what="correct_pixels", stereo=True)),
    ("5", dict(B=65536, dtype=64, M=16, what="correct_pixels", stereo=False)),
    ("5", dict(B=65536, dtype=64, M=16, what="correct", mode=1, reps=20)),
    ("5", dict(B=65536, dtype=64, M=0, what="predict", reps=50)),
    ("-", dict(B=65536, dtype=32, M=1, what="correct", mode=0)),
    # the north star's literal 15-state filter (N = 18 without the gravity block): 608-byte records instead of 800
    ("n15", dict(B=65536, dtype=32, M=0, what="predict", n=15)),
    ("n15", dict(B=65536, dtype=32, M=4, what="correct", mode=1, n=15)),
    ("n18", dict(B=65536, dtype=32, M=0, what="predict")),
    ("n18", dict(B=65536, dtype=32, M=4, what="correct", mode=1)),
):
    us, sps, gbs, roles = run(**args)
    desc = f"B={args['B']} fp{args['dtype']} {'N=15 ' if args.get('n') == 15 else ''}{args['what']}" + (f" K={args['K']}" if "K" in args else "") + \
           (f" M={args['M']} " + (("stereo" if args.get("stereo") else "left") if args["what"] == "correct_pixels" else
                                  ("stacked" if args.get("mode", 1) else "nearest")) if args["what"].startswith("correct") else "")
    print(f"{name:>4}    {desc:<48} {us:9.2f}   {sps:13.4g}   {gbs:10.0f}   {gbs / 8000:14.2f}   {roles:10d}", flush=True)
    CASES.append({"config": name, "desc": desc, "us_hip_events": us, "steps_per_launch": sps * us * 1e-6, "bytes_per_launch": gbs * us * 1e3})
# sidecar for tools/run_configs_rocprof.py (kernel-trace durations of the same launches)
out = os.environ.get("FBUS_RUN_CONFIGS_JSON")
if out:
    json.dump(CASES, open(out, "w"), indent=1)
