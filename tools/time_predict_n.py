#!/usr/bin/env python3
"""Launch time of predict_n (K ImuUpdates per launch, record resident in registers) by batch size.
  FBUS_EKF_LIB=... python tools/time_predict_n.py [K] [B ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 7
Bs = [int(a) for a in sys.argv[2:]] or [65536, 131072, 262144]
dev = torch.device("cuda:0")
prm = capi.default_params(0)
for B in Bs:
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, with_cov=False)
    acc, gyr = synth.imu_samples(0, B, 0, K, nom)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    d_acc, d_gyr, d_dt = f32(acc), f32(gyr), f32(np.full(K, 0.005))
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, None, prev)
        flt.reset_cov()
        for _ in range(5):
            flt.predict_n(d_acc, d_gyr, d_dt, K)
        flt.sync()
        flt.timing_enable(True); flt.timing_reset()
        for _ in range(40):
            flt.predict_n(d_acc, d_gyr, d_dt, K)
        flt.sync()
        ms, n = flt.timing_read(capi.KERNEL_PREDICT_N)
        x = flt.get_state()
        print(f"{os.path.basename(os.environ.get('FBUS_EKF_LIB', 'libfbus_ekf.so')):<20} B {B:>7} predict_n K={K}: {ms / n * 1e3:8.2f} us per launch, "
              f"{B * K / (ms / n * 1e-3):.3e} steps/s  finite={bool(np.isfinite(x[0]).all() and np.isfinite(x[2]).all())}")
