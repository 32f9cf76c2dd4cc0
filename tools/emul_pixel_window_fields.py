#!/usr/bin/env python3
"""Which FIELD of an fp32 record sets the floor of the free-running reprojection-row windows?  (round 6, review item 2a)

tests/test_frame_meas_gpu.py::test_window_of_frames_with_the_north_star_update measures the fp32 kernels 1.4e-4 / 1.8e-4 (literal)
off the fp64 oracle after 4 camera frames / 9 ImuUpdates and grades them against a floor: the fp64 oracle with its whole record
rounded to fp32 after every step (1.1e-4).  That emulation rounds nominal state, carried rotation AND covariance together.  Here the
same window (the test's own inputs) with ONE group of fields rounded at a time, and with everything rounded EXCEPT one group:
    p, v, q, ba/bg/g, R (carried rotation), P (covariance)
CPU only (the C oracle).   python tools/emul_pixel_window_fields.py [B]     -> profiles/r06_window_fields.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import numpy as np
import oracle_capi as oc
from fbus_ekf import capi, synth
from replay_ref import OracleEngine
from util import parity_errors, pixel_scene

r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
SIZE, DT = 0.28, np.array([0.005])
NOM = {"p": slice(0, 3), "v": slice(3, 6), "q": slice(6, 10), "bias_g": slice(10, 19)}
GROUPS = ("p", "v", "q", "bias_g", "R", "P")


def scene(B, M, dialect, n, seed):
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), n, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=seed, noise=5e-4, nominal=nom0)
    rng = np.random.default_rng(seed + 1)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))
    dq = np.concatenate([np.ones((B, 1)), rng.normal(0, 0.002, (B, 3))], axis=1)
    nom[:, 6:10] = synth.qmul(nom[:, 6:10], dq)
    nom[:, 6:10] /= np.linalg.norm(nom[:, 6:10], axis=1, keepdims=True)
    nom = r32(nom)
    return prm, nom, r32(synth.q2R(nom[:, 6:10]).reshape(B, 9)), r32(P), prev, ids, r32(left), r32(right)


def run(B, dialect, n, stereo, kcount, rounded):
    """rounded: set of group names whose fields are rounded to fp32 after every step"""
    M = 4
    F, Kt = len(kcount), sum(kcount)
    prm, nom, rot, P, prev, ids0, left0, right0 = scene(B, M, dialect, n, 61 + dialect)
    acc, gyr = synth.imu_samples(0, B, 0, Kt, nom)
    acc, gyr = r32(acc), r32(gyr)
    rng = np.random.default_rng(7)
    left = np.stack([left0] * F); right = np.stack([right0] * F)
    left = r32(left + rng.normal(0, 2e-4, left.shape)); right = r32(right + rng.normal(0, 2e-4, right.shape))
    eng = OracleEngine(B, dialect, n)
    eng.set_state(nom, rot, P, prev)

    def q():
        for g in rounded:
            if g in NOM:
                eng.nominal[:, NOM[g]] = r32(eng.nominal[:, NOM[g]])
            elif g == "R":
                eng.rot[...] = r32(eng.rot)
            else:
                eng.P[...] = r32(eng.P)
    k0 = 0
    for f, K in enumerate(kcount):
        for k in range(K):
            eng.predict(acc[k0 + k], gyr[k0 + k], DT); q()
        k0 += K
        eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids0, left[f], right[f] if stereo else None, SIZE, prm.r_pix, analytic=True)
        q()
    return eng.get_state()


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 443
    kcount = [3, 0, 2, 4]
    print(f"B = {B}, window {kcount} (9 ImuUpdates + 4 reprojection-row updates, M = 4), fp64 oracle with fp32 rounding of selected record fields "
          "after every step; figures against the un-rounded fp64 run: literal / sigma-aware (block) / plain (block) / cov block-wise")
    for dialect, n in ((0, 18), (1, 18), (0, 15)):
        for stereo in (False, True):
            ref = run(B, dialect, n, stereo, kcount, ())
            print(f"dialect {'matlab' if dialect == 0 else 'cpp'} N {n} {'stereo' if stereo else 'left'}")
            cases = [("all", GROUPS)] + [(f"only {g}", (g,)) for g in GROUPS] + [(f"all but {g}", tuple(x for x in GROUPS if x != g)) for g in GROUPS] + \
                    [("all but p, v", tuple(x for x in GROUPS if x not in ("p", "v")))]
            for name, rounded in cases:
                e = parity_errors(run(B, dialect, n, stereo, kcount, rounded), ref)
                print(f"   {name:14s} literal {e['literal']:.2e}  sigma-aware {e['sigma']:.2e} ({e['sigma_block']:2s})  plain {e['plain']:.2e} ({e['plain_block']:2s})  "
                      f"cov block-wise {e['cov_block']:.2e}")
