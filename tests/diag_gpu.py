#!/usr/bin/env python3
"""Not a test: prints the per-block GPU-vs-oracle error table (run on the GPU box)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
for sub in ("../fbus-ekf_amd", "../oracle", "."):
    sys.path.insert(0, os.path.join(HERE, sub))
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import cov_rel_err, cov_rel_err_blockwise, state_rel_err, _BLOCKS

r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)

def blocks(got, ref):
    out = []
    for name, a, b, floor in _BLOCKS:
        num = np.abs(got[:, a:b] - ref[:, a:b]).max(axis=1)
        den = np.maximum(np.abs(ref[:, a:b]).max(axis=1), floor)
        out.append(f"{name}={float((num/den).max()):.1e}")
    return " ".join(out)

B, M = 512, 4
for mixed in (False, True):
  for dialect in (0, 1):
    for n in (18, 15):
      for dtype in (32, 64):
        prm = capi.default_params(dialect)
        nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n, mixed_cov=mixed)
        nom, rot, P = r32(nom), r32(rot), r32(P)
        acc, gyr = synth.imu_samples(0, B, 0, 1, nom); acc, gyr = r32(acc), r32(gyr)
        ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
        pos = r32(pos + np.random.default_rng(5).normal(0, 0.02, pos.shape)); quat = r32(quat)
        dt = r32(np.array([0.005]))
        for what in ("predict", "near", "stack"):
            with BatchedFilter(B, prm, dtype=dtype, nstate=n) as flt:
                eng = OracleEngine(B, dialect, n)
                flt.set_state(nom, rot, P, prev); eng.set_state(nom, rot, P, prev)
                if what == "predict":
                    flt.predict(acc[0], gyr[0], dt); eng.predict(acc[0], gyr[0], dt)
                else:
                    mode = 0 if what == "near" else 1
                    flt.correct(ids, pos, quat, mode); eng.correct(ids, pos, quat, mode)
                g = flt.get_state()
            if dtype == 64 or n == 15: continue
            print(f"mixed={int(mixed)} d={dialect} n={n} f{dtype} {what:8s} sig={state_rel_err(g[0], eng.nominal, eng.P)[0]:.1e} {blocks(g[0], eng.nominal)} "
                  f"R={np.abs(g[1]-eng.rot).max():.1e} P={cov_rel_err(g[2], eng.P):.1e} Pblk={cov_rel_err_blockwise(g[2], eng.P):.1e}")

# ---- free running, per 10 frames ----
DT = r32(np.array([0.005]))
for dialect in (0, 1):
    B, M, n = 128, 4, 18
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n)
    nom, rot, P = r32(nom), r32(rot), r32(P)
    with BatchedFilter(B, prm, dtype=32, nstate=n) as flt:
        eng = OracleEngine(B, dialect, n)
        flt.set_state(nom, rot, P, prev); eng.set_state(nom, rot, P, prev)
        step = 0
        for frame in range(100):
            K = (7, 7, 6)[frame % 3]
            acc, gyr = synth.imu_samples(0, B, step, K, nom); acc, gyr = r32(acc), r32(gyr); step += K
            for k in range(K):
                flt.predict(acc[k], gyr[k], DT); eng.predict(acc[k], gyr[k], DT)
            ids, pos, quat = synth.marker_frame(0, B, frame, M, nom, prm); pos, quat = r32(pos), r32(quat)
            mode = 1 if frame % 2 else 0
            flt.correct(ids, pos, quat, mode); eng.correct(ids, pos, quat, mode)
            if frame % 10 == 9 or frame < 3:
                g = flt.get_state()
                print(f"free d={dialect} frame {frame:3d} sig={state_rel_err(g[0], eng.nominal, eng.P)[0]:.1e} {blocks(g[0], eng.nominal)} "
                      f"P={cov_rel_err(g[2], eng.P):.1e} Pblk={cov_rel_err_blockwise(g[2], eng.P):.1e} sigma_bg={np.sqrt(eng.P[:,12,12]).max():.1e}")
