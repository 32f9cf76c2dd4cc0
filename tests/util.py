"""Shared helpers for the tests (error metrics, state generation)."""
import numpy as np

# Tolerances from BASELINE.json north_star: fp32 GPU vs fp64 oracle,
# state rel-err <= 1e-5, covariance rel-err <= 1e-4 (max-norm relative).
STATE_TOL = 1e-5
COV_TOL = 1e-4

# Per-block denominators floors for the relative state error.  A relative error on
# a block whose true value is ~0 is ill-conditioned (SURVEY.md section 8(d) uses
# max(|v|, 1e-2) for velocity); the floors are the natural scales of each block:
# metres, m/s, unit quaternion, m/s^2 bias, rad/s bias, m/s^2 gravity.
_BLOCKS = (("p", 0, 3, 1e-1), ("v", 3, 6, 1e-2), ("q", 6, 10, 1.0),
           ("ba", 10, 13, 1e-2), ("bg", 13, 16, 1e-3), ("g", 16, 19, 1.0))


def state_rel_err(got, ref):
    """max over filters and blocks of ||got-ref||_inf / max(||ref||_inf, floor)."""
    got = np.asarray(got, np.float64).reshape(-1, 19)
    ref = np.asarray(ref, np.float64).reshape(-1, 19)
    worst, where = 0.0, None
    for name, a, b, floor in _BLOCKS:
        num = np.abs(got[:, a:b] - ref[:, a:b]).max(axis=1)
        den = np.maximum(np.abs(ref[:, a:b]).max(axis=1), floor)
        e = float((num / den).max())
        if e > worst:
            worst, where = e, name
    return worst, where


def rot_rel_err(got, ref):
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max())


def cov_rel_err(got, ref):
    """max over filters of max|dP| / max|P| (max-norm relative)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    B = ref.shape[0]
    num = np.abs(got - ref).reshape(B, -1).max(axis=1)
    den = np.abs(ref).reshape(B, -1).max(axis=1)
    return float((num / den).max())


def cov_rel_err_blockwise(got, ref):
    """stricter: every 3x3 block relative to the geometric scale sqrt(P_ii P_jj) of its rows/cols."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = np.sqrt(np.abs(np.einsum("bii->bi", ref)))
    scale = d[:, :, None] * d[:, None, :]
    return float((np.abs(got - ref) / scale).max())
