#!/usr/bin/env python3
"""BASELINE.json config 5: batch 65 536, M = 16 markers per frame (stacked, 112 rows of 7 per marker),
fp32 vs fp64 on the GPU: per-step and 1 s (30 frames) error of fp32 against fp64, and the fp64/fp32 timing.
(The per-corner 128 / 256-row pixel form of the same config is tests/test_pixels_gpu.py::test_config5_128_reprojection_rows_at_full_batch;
the wall times printed here include a host synchronisation per launch -- kernel times are in profiles/r02_run_configs.txt.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
from fbus_ekf import BatchedFilter, capi, synth
from util import cov_rel_err, state_rel_err, state_rel_err_literal

B, M = int(os.environ.get("SWEEP_B", 65536)), 12        # the map has 12 markers: M = 12 distinct + 4 absent slots = 16
PATTERN = (7, 7, 6)
prm = capi.default_params(0)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
dev = torch.device("cuda:0")
res = {}
for dtype, tdt in ((64, torch.float64), (32, torch.float32)):
    cv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(tdt)
    flt = BatchedFilter(B, prm, dtype=dtype)
    flt.set_state(nom, rot, P, prev)
    snaps = []
    step = 0
    t_corr = 0.0
    for frame in range(30):
        K = PATTERN[frame % 3]
        acc, gyr = synth.imu_samples(0, B, step, K, nom); step += K
        ids, pos, quat = synth.marker_frame(0, B, frame, M, nom, prm)
        ids16 = np.full((B, 16), -1, np.int32); ids16[:, :M] = ids
        pos16 = np.zeros((B, 16, 3)); pos16[:, :M] = pos
        quat16 = np.zeros((B, 16, 4)); quat16[:, :M] = quat; quat16[:, M:, 0] = 1
        d = [cv(acc), cv(gyr), cv(np.full(K, 0.005)), torch.from_numpy(ids16).to(dev), cv(pos16), cv(quat16)]
        flt.predict_n(d[0], d[1], d[2])
        flt.sync(); t0 = time.perf_counter()
        flt.correct(d[3], d[4], d[5], capi.MODE_STACKED)
        flt.sync(); t_corr += time.perf_counter() - t0
        if frame in (0, 29):
            snaps.append(flt.get_state())
    res[dtype] = (snaps, t_corr / 30)
    flt.close()
for i, name in ((0, "after 1 frame"), (1, "after 1 s (30 frames, 230 steps)")):
    a, b = res[32][0][i], res[64][0][i]
    print(f"{name}: fp32 vs fp64  state literal {state_rel_err_literal(a[0], b[0]):.2e}  per-block {state_rel_err(a[0], b[0], b[2])[0]:.2e}"
          f"  cov {cov_rel_err(a[2], b[2]):.2e}")
print(f"correct (12 markers of 16 slots, stacked) wall per launch: fp32 {res[32][1]*1e6:.1f} us, fp64 {res[64][1]*1e6:.1f} us, B = {B}")
