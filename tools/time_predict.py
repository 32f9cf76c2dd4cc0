#!/usr/bin/env python3
"""Launch time of the per-call predict by batch size and state size (HIP events around runs of back-to-back launches):
    FBUS_PREDICT_LEAN=0|1 python tools/time_predict.py [N ...]      batches 65 536 .. 1 048 576"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
dev = torch.device("cuda:0")
prm = capi.default_params(0)
Ns = [int(a) for a in sys.argv[1:]] or [18, 15]
for n in Ns:
    for B in (65536, 131072, 262144, 524288, 1048576):
        nb = 65536
        nom, rot, P, prev = synth.initial_state(0, nb, list(prm.p0_diag), n, with_cov=False)
        acc, gyr = synth.imu_samples(0, nb, 0, 4, nom)
        rep = B // nb
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        d_acc, d_gyr = f32(acc).repeat(1, rep, 1).contiguous(), f32(gyr).repeat(1, rep, 1).contiguous()
        d_dt = f32(np.full(1, 0.005))
        with BatchedFilter(B, prm, nstate=n, order_streams=False) as flt:
            flt.set_state(np.tile(nom, (rep, 1)), np.tile(rot, (rep, 1)), None, np.tile(prev, rep))
            flt.reset_cov()
            for i in range(8):
                flt.predict(d_acc[i % 4], d_gyr[i % 4], d_dt)
            flt.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = torch.cuda.Stream()
            flt.set_stream(s)
            reps = 40
            with torch.cuda.stream(s):
                e0.record(s)
                for i in range(reps):
                    flt.predict(d_acc[i % 4], d_gyr[i % 4], d_dt)
                e1.record(s)
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            bytes_moved = (828 + 608) * B if n == 18 else (4 * (28 + 124) + 4 * (19 + 124) + 28) * B
            x = flt.get_state()
            print(f"LEAN={os.environ.get('FBUS_PREDICT_LEAN', '0')} N={n} B={B:8d}: predict {us:8.2f} us per launch, {bytes_moved / us / 1e6:5.2f} TB/s moved, finite {bool(np.isfinite(x[0]).all())}", flush=True)
