#!/usr/bin/env python3
"""Timing / profiling driver of the reprojection-row update (fbus_ekf_correct_pixels_dev) on the wall scene of the bench:
    python3 tools/run_pixels.py [--batch 65536] [--slots 16] [--stereo] [--reps 12] [--corners] [--dtype 32]
prints one line per case: us per launch (HIP events on the handle's stream).  Under rocprofv3 put python3 itself after `--`."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--stereo", action="store_true")
    ap.add_argument("--both", action="store_true", help="left camera, then stereo")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--dtype", type=int, default=32)
    ap.add_argument("--roles", type=int, default=0)
    ap.add_argument("--corners", action="store_true", help="fbus_ekf_correct_corners_dev (refractive, stacked) instead of the pixel rows")
    ap.add_argument("--fused-k0", action="store_true", help="the same update through fbus_ekf_frame_meas_fused_dev with K = 0 (frame_meas_kernel: "
                                                            "record loaded up front, covariance parked in LDS across the fold)")
    args = ap.parse_args()
    import torch
    from fbus_ekf import BatchedFilter, capi, synth
    dev = torch.device("cuda:0")
    prm = capi.default_params(capi.DIALECT_MATLAB)
    size = 0.15
    prm.marker_size = size
    B = args.batch
    nom, rot, ids, left, right = synth.pixel_wall_scene(B, args.slots, prm, size, seed=9, stereo=True)
    tt = torch.float32 if args.dtype == 32 else torch.float64
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(tt)
    d_ids, d_left, d_right = torch.from_numpy(ids).to(dev), f(left), f(right)
    nvis = float((ids >= 0).sum(axis=1).mean())
    prev0 = np.zeros(B, np.int32)
    with BatchedFilter(B, prm, device=0, dtype=args.dtype, order_streams=False) as flt:
        flt.set_team(0, args.roles)
        for stereo in ((True,) if args.corners else ((False, True) if args.both else (args.stereo,))):
            torch.cuda.synchronize()
            for k in range(args.reps + 2):
                if k == 2:
                    flt.sync(); flt.timing_enable(True); flt.timing_reset()
                flt.set_state(nom, rot, None, prev0)
                flt.reset_cov()
                if args.fused_k0:
                    flt.frame_meas(None, None, None, d_ids, d_left, d_right if stereo else None, capi.MEAS_CORNERS if args.corners else capi.MEAS_PIXELS,
                                   capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                elif args.corners:
                    flt.correct_corners(d_ids, d_left, d_right, capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                else:
                    flt.correct_pixels(d_ids, d_left, d_right if stereo else None)
            ms, n = flt.timing_read(capi.KERNEL_FRAME if args.fused_k0 else capi.KERNEL_CORRECT_CORNERS)
            flt.timing_enable(False)
            g = flt.get_state()
            ok = bool(np.isfinite(g[0]).all() and np.isfinite(g[2]).all())
            rows = nvis * (16 if stereo else 8)
            print(f"{'frame_meas K=0 ' if args.fused_k0 else ''}{'correct_corners' if args.corners else 'correct_pixels'} fp{args.dtype} B {B} slots {args.slots} ({nvis:.1f} in view, {rows:.0f} rows) {'stereo' if stereo else 'left'}: "
                  f"{ms / n * 1e3:.1f} us per launch, applied {float(flt.applied().mean()):.3f}, finite {ok}, "
                  f"posterior sigma_p {float(np.sqrt(g[2][:, 0, 0]).mean()):.2e}", flush=True)


if __name__ == "__main__":
    main()
