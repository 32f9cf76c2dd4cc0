#!/usr/bin/env python3
"""Cache policy of the per-call predict's record accesses by configuration: the bench pattern (7 / 7 / 6 predicts, a stacked
4-marker correct behind each run) with the policy forced through the handle's environment knobs, against the launcher's own
choice ("auto").  Prints EKF steps/s and the predict launch time (HIP events) per (dtype, N, filters, policy).
    python tools/policy_sweep.py [quick]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

dev = torch.device("cuda:0")
PATTERN = (7, 7, 6)
POLICIES = {"auto": {}, "nt/nt": {"FBUS_PREDICT_POLICY": "0"}, "default/nt": {"FBUS_PREDICT_POLICY": "1"},
            "default/default": {"FBUS_PREDICT_POLICY": "2"}}


def run(dtype, n, B, env, steps=3):
    for k in ("FBUS_PREDICT_LD", "FBUS_BIG_RECORDS_MB", "FBUS_PREDICT_POLICY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    prm = capi.default_params(0)
    tdt = torch.float32 if dtype == 32 else torch.float64
    base = B if B <= 131072 else 65536
    tile = B // base
    nom, rot, P, prev = synth.initial_state(0, base, list(prm.p0_diag), n, with_cov=False)
    up = lambda a, d=0: torch.cat([torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(tdt)] * tile, dim=d)
    acc, gyr = synth.imu_samples(0, base, 0, sum(PATTERN), nom)
    d_acc, d_gyr = up(acc, 1), up(gyr, 1)
    frames = []
    for f in range(3):
        ids, pos, quat = synth.marker_frame(0, base, f, 4, nom, prm)
        frames.append((torch.cat([torch.from_numpy(ids).to(dev)] * tile), up(pos), up(quat)))
    d_dt = torch.full((7,), 0.005, dtype=tdt, device=dev)
    with BatchedFilter(B, prm, dtype=dtype, nstate=n, order_streams=False) as flt:
        flt.set_state(np.tile(nom, (tile, 1)), np.tile(rot, (tile, 1)), None, np.tile(prev, tile))
        flt.reset_cov()

        def step():
            for _ in range(10):
                k = 0
                for f, K in enumerate(PATTERN):
                    flt.frame(d_acc[k:k + K], d_gyr[k:k + K], d_dt[:K], *frames[f], capi.MODE_STACKED)
                    k += K
        step(); flt.sync(); flt._keep.clear()
        flt.timing_enable(True, stride=4); flt.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        ms, cnt = flt.timing_read(capi.KERNEL_PREDICT)
        flt.timing_enable(False)
    return B * 230 * steps / el, ms / max(cnt, 1) * 1e3


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
grid = [(32, 18, b) for b in (49152, 65536, 73728, 98304, 131072, 262144, 393216, 524288, 786432, 1048576)] + \
       [(32, 15, b) for b in (49152, 65536, 98304, 131072)] + [(64, 18, b) for b in (32768, 49152, 65536, 98304)]
if quick:
    grid = [(32, 18, 65536), (32, 18, 1048576)]
print(f"{'dtype':>5} {'N':>3} {'filters':>8} {'records MB':>10}  " + "  ".join(f"{p:>24}" for p in POLICIES) + "   auto vs best")
for dtype, n, B in grid:
    rec_mb = B * (200 if n == 18 else 152) * (4 if dtype == 32 else 8) / 1e6
    res = {p: run(dtype, n, B, env, steps=2 if B >= 524288 else 3) for p, env in POLICIES.items()}
    best = max(v[0] for k, v in res.items() if k != "auto")
    print(f"{dtype:>5} {n:>3} {B:>8} {rec_mb:>10.0f}  " + "  ".join(f"{v[0]:.3e} ({v[1]:6.2f} us)" .rjust(24) for v in res.values()) +
          f"   {res['auto'][0] / best - 1:+.1%}", flush=True)
