#!/usr/bin/env python3
"""Free-running frame windows (tests/test_team_gpu.py::test_team_frame_window_equals_one_wave_window: 4 frames = 16 ImuUpdates + 4
MeasureUpdates without re-seeding): how much of the fp32-vs-fp64 gap is the fp32 REPRESENTATION of the record between steps, with
EXACT arithmetic inside every step?  Two runs of the fp64 oracle on the test's own inputs:
    A  fp64 throughout (what the GPU results are compared with)
    Q  the same, with the record (nominal state, carried rotation, covariance) rounded to fp32 after EVERY step -- the least any
       implementation that keeps fp32 records can lose (the north star's records are fp32; the resident kernels hold them in fp32
       registers between steps)
and a second quantised run Q' whose rounding is perturbed (round-to-nearest of x (1 + 2^-25)): two legitimate fp32 runs differ from
each other by what Q'-vs-Q shows -- the level "team vs one-wave" kernels (different FMA contraction) can differ by.
CPU only (the C oracle).   python tools/emul_window_quantisation.py [B]
Result (B = 443, the test's batch; literal = ||dx||_inf / ||x||_inf over the 19-vector, after frame 4):
    see profiles/r05_window_quantisation.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
from fbus_ekf import capi, synth
from replay_ref import OracleEngine
from util import parity_errors

r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
r32b = lambda a: (np.asarray(a, np.float64) * (1.0 + 2.0 ** -25)).astype(np.float32).astype(np.float64)
DT = np.array([0.005])


def batch(B, dialect, n, seed_off):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(seed_off, seed_off + B, list(prm.p0_diag), n, mixed_cov=True)
    return prm, r32(nom), r32(rot), r32(P), prev


def window_inputs(B, dialect, n, M, kcount, seed_off):
    prm, nom, rot, P, prev = batch(B, dialect, n, seed_off)
    Kt = sum(kcount)
    acc, gyr = synth.imu_samples(seed_off, seed_off + B, 0, Kt, nom)
    frames = [synth.marker_frame(seed_off, seed_off + B, f, M, nom, prm) for f in range(len(kcount))]
    ids = np.stack([f[0] for f in frames]); pos = r32(np.stack([f[1] for f in frames])); quat = r32(np.stack([f[2] for f in frames]))
    return prm, nom, rot, P, prev, r32(acc), r32(gyr), ids, pos, quat


def run(B, dialect, mode, n, kcount, quant):
    prm, nom, rot, P, prev, acc, gyr, ids, pos, quat = window_inputs(B, dialect, n, 4, kcount, 5)
    eng = OracleEngine(B, dialect, n)
    eng.set_state(nom, rot, P, prev)
    out = []

    def q():
        if quant is not None:
            eng.nominal[...] = quant(eng.nominal); eng.rot[...] = quant(eng.rot); eng.P[...] = quant(eng.P)
    k0 = 0
    for f, K in enumerate(kcount):
        for k in range(K):
            eng.predict(acc[k0 + k], gyr[k0 + k], DT); q()
        k0 += K
        eng.correct(ids[f], pos[f], quat[f], mode); q()
        out.append(eng.get_state())
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 7 * 64 - 5
    kcount = [7, 0, 6, 3]
    print(f"B = {B}, window {kcount} (16 ImuUpdates + 4 MeasureUpdates), M = 4; figures after each frame: "
          "literal / sigma-aware (block) / cov max-norm / cov block-wise")
    for n in (18, 15):
        for dialect in (0, 1):
            for mode in (0, 1):
                A = run(B, dialect, mode, n, kcount, None)
                Q = run(B, dialect, mode, n, kcount, r32)
                Q2 = run(B, dialect, mode, n, kcount, r32b)
                print(f"N {n} dialect {'matlab' if dialect == 0 else 'cpp   '} mode {'nearest' if mode == 0 else 'stacked'}")
                for f in range(len(kcount)):
                    e, d = parity_errors(Q[f], A[f]), parity_errors(Q2[f], Q[f])
                    print(f"   frame {f + 1}: fp32 records vs fp64  literal {e['literal']:.2e} sigma {e['sigma']:.2e} ({e['sigma_block']:2s}) cov {e['cov']:.2e} "
                          f"cov-block {e['cov_block']:.2e}   |  two fp32-record runs  literal {d['literal']:.2e} sigma {d['sigma']:.2e} cov-block {d['cov_block']:.2e}")
