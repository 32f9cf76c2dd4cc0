#!/usr/bin/env python3
"""round 5 diagnostics (GPU):
  A  where the fused frame (fbus_ekf_frame_meas_fused_dev) and the per-call sequence part ways: predicts only (M = 0), the update only
     (K = 0), both -- elements that differ and by how many ulps
  B  the bench's north-star rows frame by frame: filters with a non-finite state and the smallest eigenvalue of the covariance's
     correlation matrix as the frames go by (left / stereo, N = 18 / 15, fp32 / fp64 records)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
from fbus_ekf import BatchedFilter, capi, synth

dev = torch.device("cuda:0")
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def dd(a, dtype=32):
    a = np.asarray(a)
    if a.dtype.kind in "iu":
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev).to(torch.float32 if dtype == 32 else torch.float64)


def ulps(x, y):
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    xi = x.view(np.int32).astype(np.int64); yi = y.view(np.int32).astype(np.int64)
    return np.abs(xi - yi)


def part_a():
    B, M, K = 448 - 5, 4, 3
    prm = capi.default_params(0); prm.marker_size = 0.15
    nom, rot, ids, left, right = synth.pixel_wall_scene(B, M, prm, 0.15, seed=9, stereo=True)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    d = dict(acc=dd(acc), gyr=dd(gyr), dt=dd(np.full(K, 0.005)), ids=dd(ids), left=dd(left), right=dd(right))
    for name, k, m in (("predicts only", K, 0), ("update only", 0, M), ("both", K, M)):
        out = []
        for fused in (True, False):
            with BatchedFilter(B, prm) as f:
                f.set_team(1, 1)
                f.set_state(nom, rot, None, np.zeros(B, np.int32)); f.reset_cov()
                a, g, t = (d["acc"][:k], d["gyr"][:k], d["dt"][:k]) if k else (None, None, None)
                i, l = (d["ids"], d["left"]) if m else (None, None)
                if fused:
                    f.frame_meas(a, g, t, i, l, None, capi.MEAS_PIXELS)
                else:
                    for j in range(k):
                        f.predict(d["acc"][j], d["gyr"][j], d["dt"][:1])
                    if m:
                        f.correct_pixels(d["ids"], d["left"], None)
                f.sync()
                out.append(f.get_state())
        for x, y, nm in zip(out[0], out[1], ("nominal", "rot", "P")):
            u = ulps(x, y)
            print(f"[A] {name:14s} {nm:8s}: {int((u > 0).sum())} of {u.size} elements differ, max {int(u.max())} ulp, "
                  f"filters affected {int((u.reshape(B, -1) > 0).any(axis=1).sum())} of {B}", flush=True)
            if nm == "nominal" and u.max() > 0:
                cols = np.unique(np.nonzero(u.reshape(B, -1))[1])
                print("      nominal columns that differ:", cols.tolist())
            if nm == "P" and u.max() > 0:
                ij = np.argwhere(u.reshape(B, 18, 18).max(axis=0) > 0)
                print("      P elements that differ (i, j) [first 24]:", ij[:24].tolist())


def part_b():
    B = 8192
    for slots, stereo, n, dtype in ((4, False, 18, 32), (4, True, 18, 32), (4, False, 15, 32), (4, True, 18, 64), (4, False, 15, 64), (16, True, 18, 32)):
        prm = capi.default_params(0); prm.marker_size = 0.15
        nom, rot, ids, left, right = synth.pixel_wall_scene(B, slots, prm, 0.15, seed=9, stereo=True)
        acc, gyr = synth.imu_samples(0, B, 0, 20, nom)
        d_acc, d_gyr, d_dt = dd(acc, dtype), dd(gyr, dtype), dd(np.full(1, 0.005), dtype)
        d_ids, d_left, d_right = dd(ids), dd(left, dtype), dd(right, dtype)
        with BatchedFilter(B, prm, dtype=dtype, nstate=n) as f:
            f.set_state(nom, rot, None, np.zeros(B, np.int32)); f.reset_cov()
            fr = 0
            for rep in range(12):
                k = 0
                for K in (7, 7, 6):
                    for j in range(K):
                        f.predict(d_acc[k + j], d_gyr[k + j], d_dt)
                    k += K
                    f.correct_pixels(d_ids, d_left, d_right if stereo else None)
                    fr += 1
                f.sync()
                g = f.get_state()
                bad = ~(np.isfinite(g[0]).all(axis=1) & np.isfinite(g[2]).reshape(B, -1).all(axis=1))
                Ps = g[2][~bad][::16].astype(np.float64)
                dg = np.sqrt(np.abs(np.einsum("bii->bi", Ps)))
                ev = np.linalg.eigvalsh(Ps / (dg[:, :, None] * dg[:, None, :])).min(axis=1) if len(Ps) else np.array([np.nan])
                print(f"[B] slots {slots} {'stereo' if stereo else 'left  '} N {n} fp{dtype} frame {fr:3d}: non-finite filters {int(bad.sum()):5d} of {B}, "
                      f"min diag {np.einsum('bii->bi', Ps).min():.2e}, min eig(corr) {ev.min():.2e}, filters with eig < 0: {int((ev < 0).sum())} of {len(ev)}, "
                      f"sigma_p {float(np.sqrt(np.abs(Ps[:, 0, 0])).mean()):.2e}", flush=True)
                if bad.sum() > B // 2:
                    break


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "ab"
    if "a" in what:
        part_a()
    if "b" in what:
        part_b()
