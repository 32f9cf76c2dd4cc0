#!/usr/bin/env python3
"""Team kernels (several waves per 64-filter tile) against the one-wave kernels, launch by launch (HIP-event bracket per launch,
which adds ~2-3 us to every figure).  python tools/time_team.py [B ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

dev = torch.device("cuda:0")
prm = capi.default_params(0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.float32)


def timed(flt, kind, fn, reps):
    for _ in range(5):
        fn()
    flt.sync()
    flt.timing_enable(True); flt.timing_reset()
    for _ in range(reps):
        fn()
    ms, n = flt.timing_read(kind)
    flt.timing_enable(False)
    return ms / n * 1e3


Bs = [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 49152, 65536, 73728]
print("B        kernel                 " + "".join(f"roles={r:<8d}" for r in (1, 2, 3, 4)) + "   (us per launch)")
for B in Bs:
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    acc, gyr = synth.imu_samples(0, B, 0, 8, nom)
    d_acc, d_gyr, d_dt = up(acc), up(gyr), up(np.full(8, 0.005))
    ids, pos, quat = synth.marker_frame(0, B, 0, 4, nom, prm)
    d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), up(pos), up(quat)
    rows = {"predict": [], "predict_n K=8": [], "correct stacked M=4": [], "correct nearest M=4": []}
    with BatchedFilter(B, prm, order_streams=False) as flt:
        flt.set_state(nom, rot, P, prev)
        for r in (1, 2, 3, 4):
            flt.set_team(r, r)
            rows["predict"].append(timed(flt, capi.KERNEL_PREDICT, lambda: flt.predict(d_acc[0], d_gyr[0], d_dt[:1]), 300))
            if r in (1, 4):
                rows["predict_n K=8"].append(timed(flt, capi.KERNEL_PREDICT_N, lambda: flt.predict_n(d_acc, d_gyr, d_dt), 100))
            else:
                rows["predict_n K=8"].append(float("nan"))
            rows["correct stacked M=4"].append(timed(flt, capi.KERNEL_CORRECT, lambda: flt.correct(d_ids, d_pos, d_quat, 1), 200))
            rows["correct nearest M=4"].append(timed(flt, capi.KERNEL_CORRECT, lambda: flt.correct(d_ids, d_pos, d_quat, 0), 200))
    for name, v in rows.items():
        print(f"{B:<8d} {name:<22s} " + "".join(f"{x:<14.2f}" for x in v))
