#!/usr/bin/env python3
"""Launch times of the resident paths by K (samples per launch), fp64 or fp32 records:
    python3 tools/run_f64_fused.py [--dtype 64] [--batch 65536]
predict_n for K = 1, 2, 4, 8, 16 and the fused frame for K = 1, 7, 15 (M = 4, stacked) -- the slope is the cost of one resident
predict step, the intercept the record's way in and out (+ the correct).  HIP events on the handle's stream."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dtype", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from fbus_ekf import BatchedFilter, capi, synth
    dev = torch.device("cuda:0")
    prm = capi.default_params(capi.DIALECT_MATLAB)
    B, M = args.batch, 4
    tt = torch.float32 if args.dtype == 32 else torch.float64
    KM = 16
    nb = min(B, 4096)                                        # 4096 trajectories, repeated (timing only)
    rep = lambda a, ax: np.concatenate([a] * (B // nb), axis=ax) if B > nb else a
    nom0, rot0, _, prev0 = synth.initial_state(0, nb, list(prm.p0_diag), 18, with_cov=False)
    a_, g_ = synth.imu_samples(0, nb, 0, KM, nom0)
    i_, p_, q_ = synth.marker_frame(0, nb, 0, M, nom0, prm)
    f = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev).to(tt)
    acc, gyr = f(rep(a_, 1)), f(rep(g_, 1))
    dt = torch.full((KM,), 0.005, dtype=tt, device=dev)
    ids, pos, quat = torch.from_numpy(np.ascontiguousarray(rep(i_, 0))).to(dev), f(rep(p_, 0)), f(rep(q_, 0))
    nom0, rot0, prev0 = rep(nom0, 0), rep(rot0, 0), rep(prev0, 0)
    with BatchedFilter(B, prm, device=0, dtype=args.dtype, order_streams=False) as flt:
        def timed(kid, fn):
            for k in range(args.reps + 2):
                if k == 2:
                    flt.sync(); flt.timing_enable(True); flt.timing_reset()
                fn()
            ms, n = flt.timing_read(kid)
            flt.timing_enable(False)
            return ms / max(n, 1) * 1e3
        torch.cuda.synchronize()
        fresh = lambda: (flt.set_state(nom0, rot0, None, prev0), flt.reset_cov())
        fresh()
        for K in (1, 2, 4, 8, 16):
            kid = capi.KERNEL_PREDICT if K == 1 else capi.KERNEL_PREDICT_N
            us = timed(kid, lambda: flt.predict_n(acc[:K], gyr[:K], dt[:K], K=K))
            print(f"fp{args.dtype} B {B} predict_n K {K:2d}: {us:8.1f} us per launch, {us / K:6.2f} per step", flush=True)
        fresh()
        us = timed(capi.KERNEL_CORRECT, lambda: flt.correct(ids, pos, quat, capi.MODE_STACKED))
        print(f"fp{args.dtype} B {B} correct stacked M 4: {us:8.1f} us per launch", flush=True)
        for K in (1, 7, 15):
            fresh()
            us = timed(capi.KERNEL_FRAME, lambda: flt.frame(acc[:K], gyr[:K], dt[:K], ids, pos, quat, capi.MODE_STACKED, fused=True))
            print(f"fp{args.dtype} B {B} fused frame K {K:2d} + correct: {us:8.1f} us per launch", flush=True)
        g = flt.get_state()
        print("finite", bool(np.isfinite(g[0]).all() and np.isfinite(g[2]).all()))


if __name__ == "__main__":
    main()
