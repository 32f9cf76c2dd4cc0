#!/usr/bin/env python3
"""Time of ONE ImuUpdate inside the resident loop: predict_n at K = 4, 8, 16, 32 steps per launch (HIP events around runs of launches),
the slope is the step, the intercept the record's way in and out.  65 536 filters = one wave per SIMD.
    FBUS_EKF_LIB=... python tools/time_predict_n.py [N ...] [--dialect D] [--batch B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
args = sys.argv[1:]
dialect = int(args[args.index("--dialect") + 1]) if "--dialect" in args else 0
B = int(args[args.index("--batch") + 1]) if "--batch" in args else 65536
Ns = [int(a) for i, a in enumerate(args) if a.isdigit() and (i == 0 or not args[i - 1].startswith("--"))] or [18]
dev = torch.device("cuda:0")
prm = capi.default_params(dialect)
for n in Ns:
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n, with_cov=False)
    KMAX = 32
    acc, gyr = synth.imu_samples(0, B, 0, KMAX, nom)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_acc, d_gyr, d_dt = f32(acc), f32(gyr), f32(np.full(KMAX, 0.005))
    with BatchedFilter(B, prm, dialect=dialect, nstate=n, order_streams=False) as flt:
        if os.environ.get("TEAM_PREDICT"):
            flt.set_team(int(os.environ["TEAM_PREDICT"]), 1)           # 1 = one-wave kernels, 4 = four roles; 2 = the two-role predict_n of tools/patches/r05_predict_n_duo.diff (without the patch: four)
        flt.set_state(nom, rot, None, prev)
        flt.reset_cov()
        s = torch.cuda.Stream()
        flt.set_stream(s)
        pts = []
        for K in (4, 8, 16, 32):
            with torch.cuda.stream(s):
                for _ in range(5):
                    flt.predict_n(d_acc[:K], d_gyr[:K], d_dt[:K])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 30
                e0.record(s)
                for _ in range(reps):
                    flt.predict_n(d_acc[:K], d_gyr[:K], d_dt[:K])
                e1.record(s)
            e1.synchronize()
            pts.append((K, e0.elapsed_time(e1) * 1e3 / reps))
        ks, us = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts])
        slope, icpt = np.polyfit(ks, us, 1)
        x = flt.get_state()
        print(f"{os.path.basename(os.environ.get('FBUS_EKF_LIB', 'libfbus_ekf.so')):18s} roles {os.environ.get('TEAM_PREDICT', 'auto')} N={n} dialect {dialect} B={B}: " +
              "  ".join(f"K={k}: {u:7.2f} us" for k, u in pts) + f"   step {slope:.3f} us  in/out {icpt:.2f} us  finite {bool(np.isfinite(x[0]).all())}", flush=True)
