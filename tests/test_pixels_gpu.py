"""GPU suite: correct() from corner PIXELS -- the north star's "flat-port refractive stereo reprojection of ArUco corners,
per-corner 2 x N Jacobians" (`fbus_ekf_correct_pixels`).  No reference counterpart: the reference has only the
back-projection.  What pins it: the oracle's forward projection inverts the reference-pinned back-projection to 1e-12
(tests/test_oracle_cpu.py); the oracle's Jacobian rows are CENTRAL DIFFERENCES of that projection while the device's are the
closed form, so agreement of the two posteriors also checks the analytic d pi / d X; fp64 device == oracle to 1e-7 (the
finite-difference error of the oracle's rows), fp32 within stated bounds."""
import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import (COV_BLOCK_TOL, COV_TOL, PLAIN_TOL, STATE_TOL, WINDOW_TOL, assert_parity, parity_errors, pixel_scene)

pytestmark = pytest.mark.gpu
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
SIZE = 0.28


def _scene(B, M, dialect, seed, noise=5e-4):
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=seed, noise=noise, nominal=nom0)
    rng = np.random.default_rng(seed + 1)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))                      # innovations of a few mm / mrad
    dq = np.concatenate([np.ones((B, 1)), rng.normal(0, 0.002, (B, 3))], axis=1)
    nom[:, 6:10] = synth.qmul(nom[:, 6:10], dq)
    nom[:, 6:10] /= np.linalg.norm(nom[:, 6:10], axis=1, keepdims=True)
    nom = r32(nom)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    return prm, nom, rot, r32(P), prev, ids, r32(left), r32(right)


@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("dialect", [0, 1])
def test_correct_pixels_matches_the_oracle(dialect, stereo):
    B, M = 320, 4
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, dialect, seed=11 + dialect)
    ids[0] = -1                                                    # nothing visible
    ids[1, :] = 9                                                  # only ids outside the map
    skip = (np.arange(B) % 11 == 5).astype(np.uint8)
    rgt = right if stereo else None
    eng = OracleEngine(B, dialect, 18)
    eng.set_state(nom, rot, P, prev)
    keep = eng.get_state()
    ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, rgt, SIZE, prm.r_pix)
    now = eng.get_state()
    for x, y in zip(now, keep):
        x[skip == 1] = y[skip == 1]
    eng.set_state(*now)
    ok[skip == 1] = 0
    assert ok[0] == 0 and ok[1] == 0 and ok[2:][skip[2:] == 0].all()
    for dtype in (64, 32):
        with BatchedFilter(B, prm, dtype=dtype) as flt:
            flt.set_state(nom, rot, P, prev)
            flt.correct_pixels(ids, left, rgt, skip)
            got = flt.get_state()
            assert (flt.applied() == ok).all()
        untouched = ok == 0
        assert np.array_equal(got[0][untouched], nom[untouched].astype(got[0].dtype))
        e = parity_errors(got, eng.get_state())
        print(f"[parity] correct_pixels dialect {dialect} {'stereo' if stereo else 'left'} fp{dtype}: literal {e['literal']:.2e} "
              f"sigma-aware {e['sigma']:.2e} ({e['sigma_block']}) plain {e['plain']:.2e} ({e['plain_block']}) cov {e['cov']:.2e} "
              f"cov block-wise {e['cov_block']:.2e}")
        if dtype == 64:
            # the oracle's rows are central differences (eps = 1e-6 m) of its projection: ~1e-9 relative on H
            assert e["literal"] < 1e-7 and e["sigma"] < 1e-7 and e["cov_block"] < 1e-7 and e["asym"] == 0
        else:
            # 32-64 rows with innovations of ~1e-3 in normalised coordinates known to 6e-8: 10x the single-step gates
            assert e["literal"] <= STATE_TOL and e["sigma"] <= WINDOW_TOL and e["plain"] <= 10 * PLAIN_TOL
            assert e["cov"] <= COV_TOL and e["cov_block"] <= 10 * COV_BLOCK_TOL and e["asym"] == 0


def test_correct_pixels_converges_on_the_true_pose():
    """repeated updates with the projections of the TRUE corners pull a displaced fp32 filter onto the truth (1 mm / 1 mrad),
    a different check from oracle parity: the Jacobian has the right sign and scale"""
    B, M = 256, 3
    prm = capi.default_params(0)
    prm.marker_size = SIZE
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=5)
    rng = np.random.default_rng(6)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.01, (B, 3))
    rot = synth.q2R(nom[:, 6:10]).reshape(B, 9)
    P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, np.zeros(B, np.int32))
        for _ in range(6):
            flt.correct_pixels(ids, left, right)
            flt.reset_cov()                                           # keep the gain up: this is a Gauss-Newton iteration
        got = flt.get_state()
    assert np.abs(got[0][:, 0:3] - truth[:, 0:3]).max() < 2e-3
    dq = synth.qmul(truth[:, 6:10] * np.array([1, -1, -1, -1.0]), got[0][:, 6:10].astype(np.float64))
    assert np.abs(dq[:, 1:]).max() < 2e-3


def test_config5_128_reprojection_rows_at_full_batch():
    """the literal north-star shape: B = 65 536, 16 marker slots x 4 corners x 2 rows = 128 stacked reprojection rows (left
    camera), fp32: finite, symmetric positive definite, unit quaternions on every filter, and the oracle on a strided
    subset.  Only the markers in front of the port contribute rows (the scene generator fills the slots it can)."""
    import torch
    B, M = 65536, 16
    prm = capi.default_params(0)
    prm.marker_size = SIZE
    base, _, ids_s, left_s, right_s = pixel_scene(512, M, prm, SIZE, seed=9, noise=5e-4)
    rep = B // 512
    nom = np.tile(base, (rep, 1)); ids = np.tile(ids_s, (rep, 1)); left = np.tile(left_s, (rep, 1, 1))
    rng = np.random.default_rng(10)
    nom[:, 0:3] += rng.normal(0, 0.003, (B, 3))
    nom = r32(nom); left = r32(left)
    rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
    P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
    prev = np.zeros(B, np.int32)
    dev = torch.device("cuda:0")
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        d = (torch.from_numpy(ids).to(dev), torch.from_numpy(left.astype(np.float32)).to(dev))
        flt.timing_enable(True); flt.timing_reset()
        for _ in range(3):
            flt.set_state(nom, rot, P, prev)
            flt.correct_pixels(d[0], d[1], None)
        flt.sync()
        ms, n = flt.timing_read(capi.KERNEL_CORRECT_CORNERS)
        got = flt.get_state()
        assert (flt.applied() == 1).all()
    print(f"[perf] correct_pixels B = 65536, 16 slots ({(ids_s >= 0).sum(axis=1).mean():.1f} visible on average), left camera: {ms / n * 1e3:.1f} us per launch")
    assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
    assert np.abs(np.linalg.norm(got[0][:, 6:10], axis=1) - 1).max() < 1e-6
    assert np.abs(got[2] - np.swapaxes(got[2], 1, 2)).max() == 0
    assert np.linalg.eigvalsh(got[2][::97].astype(np.float64)).min() > 0
    sub = np.arange(0, B, 1021)
    eng = OracleEngine(len(sub), 0, 18)
    eng.set_state(nom[sub], rot[sub], P[sub], prev[sub])
    eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids[sub], left[sub], None, SIZE, prm.r_pix)
    e = parity_errors([x[sub] for x in got], eng.get_state())
    print(f"[parity] correct_pixels 128-row shape, fp32 vs oracle: literal {e['literal']:.2e} sigma-aware {e['sigma']:.2e} "
          f"plain {e['plain']:.2e} cov block-wise {e['cov_block']:.2e}")
    assert e["literal"] <= 5e-5 and e["sigma"] <= WINDOW_TOL and e["cov_block"] <= 10 * COV_BLOCK_TOL
