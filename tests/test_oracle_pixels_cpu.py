"""CPU suite: the two pixel oracles against each other.

`fbo_correct_pixels` differentiates the flat-port projection by central differences (eps = 1e-6 m); round 6 adds
`fbo_correct_pixels_analytic`, whose d pi / d X comes from the implicit-function theorem on the forward model
(oracle/vision_oracle.c::fbv_project_camera_jac; the ray geometry of vision.cpp:505-552 run forward) -- written from the model,
not from the device code.  Here: the closed-form Jacobian against a FOURTH-ORDER difference stencil of the projection (1e-9: the
closed form is exact, the stencil's own error is what is left), against the oracle's own second-order differences (1e-7: their
error), on the axis of the port, for both cameras and a tilted port; and the two posteriors of the update against each other.
The analytic oracle is what the fp64 kernels are held to at 1e-9 (tests/test_pixels_gpu.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle_capi as oc
from fbus_ekf import capi, synth
from replay_ref import OracleEngine
from util import parity_errors, pixel_scene

SIZE = 0.28


def _proj(lib, vp, X, which):
    uv = np.zeros(2)
    ok = lib.fbv_project_camera(C.byref(vp), oc._dp(np.ascontiguousarray(X, np.float64)), which, oc._dp(uv))
    return ok, uv


def _jac(lib, vp, X, which):
    uv, J = np.zeros(2), np.zeros(6)
    ok = lib.fbv_project_camera_jac(C.byref(vp), oc._dp(np.ascontiguousarray(X, np.float64)), which, oc._dp(uv), oc._dp(J))
    return ok, uv, J.reshape(2, 3)


def _stencil(lib, vp, X, which, h=2e-4):
    """five-point stencil: error O(h^4 f^(5)) ~ 1e-12 relative for this smooth map + rounding 1e-16 / h"""
    J = np.zeros((2, 3))
    for c in range(3):
        e = np.zeros(3)
        e[c] = h
        f = lambda s: _proj(lib, vp, X + s * e, which)[1]
        J[:, c] = (-f(2) + 8 * f(1) - 8 * f(-1) + f(-2)) / (12 * h)
    return J


@pytest.mark.parametrize("tilted", [False, True])
def test_closed_form_projection_jacobian(tilted):
    lib = oc.load()
    vp = oc.vision_params()
    if tilted:
        n = np.array([0.05, -0.03, 1.0])
        n /= np.linalg.norm(n)
        for i in range(3):
            vp.normal[i] = n[i]
    rng = np.random.default_rng(3)
    worst_stencil = worst_fd = 0.0
    seen = 0
    for _ in range(400):
        z = rng.uniform(0.3, 1.8)
        X = np.array([rng.uniform(-0.5, 0.5) * z, rng.uniform(-0.5, 0.5) * z, z])
        for which in (0, 1):
            ok, uv, J = _jac(lib, vp, X, which)
            ok0, uv0 = _proj(lib, vp, X, which)
            assert ok == ok0
            if not ok:
                continue
            seen += 1
            assert np.array_equal(uv, uv0)                           # the same projection, bit for bit
            scale = np.abs(J).max()
            worst_stencil = max(worst_stencil, np.abs(J - _stencil(lib, vp, X, which)).max() / scale)
            eps = 1e-6                                               # the stencil fbo_correct_pixels uses
            fd = np.stack([(_proj(lib, vp, X + eps * np.eye(3)[c], which)[1] - _proj(lib, vp, X - eps * np.eye(3)[c], which)[1]) / (2 * eps)
                           for c in range(3)], axis=1)
            worst_fd = max(worst_fd, np.abs(J - fd).max() / scale)
    print(f"[oracle] d pi / d X closed form vs 4th-order stencil {worst_stencil:.2e}, vs the FD oracle's 2nd-order differences {worst_fd:.2e} "
          f"({seen} projections, tilted port: {tilted})")
    assert seen > 500
    assert worst_stencil < 1e-9
    assert worst_fd < 1e-7


def test_on_the_axis_of_the_port():
    """rho -> 0: k = t / rho has a limit; the Jacobian is continuous through it"""
    lib = oc.load()
    vp = oc.vision_params()
    n = np.array([vp.normal[i] for i in range(3)])
    for z in (0.4, 1.0):
        Xp = z * n                                                   # a point on the axis, in the refraction frame
        Xcam = np.array([-Xp[0], -Xp[1], Xp[2]])                      # (left camera frame: the flip of vision.cpp:597-599)
        ok, uv, J0 = _jac(lib, vp, Xcam, 0)
        assert ok and np.allclose(uv, [n[0] / n[2], n[1] / n[2]], atol=1e-15)
        _, _, J1 = _jac(lib, vp, Xcam + np.array([1e-9, -2e-9, 0.0]), 0)
        assert np.abs(J0 - J1).max() < 1e-7 * np.abs(J0).max()
        assert np.abs(J0 - _stencil(lib, vp, Xcam, 0)).max() < 1e-8 * np.abs(J0).max()


@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("dialect", [0, 1])
def test_the_two_pixel_oracles_agree(dialect, stereo):
    """posterior of the analytic-Jacobian update vs the central-difference one: 1e-7 (the FD rows' error through the gain of 32-64
    rows at r_pix = 1e-6); applied flags equal; the Joseph form of either equals its simple form to rounding"""
    B, M = 96, 4
    prm = capi.default_params(dialect)
    prm.marker_size = SIZE
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    truth, _, ids, left, right = pixel_scene(B, M, prm, SIZE, seed=21 + dialect, noise=5e-4, nominal=nom0)
    rng = np.random.default_rng(5)
    nom = truth.copy()
    nom[:, 0:3] += rng.normal(0, 0.004, (B, 3))
    rot = synth.q2R(nom[:, 6:10]).reshape(B, 9)
    ids[0] = -1
    rgt = right if stereo else None
    out = {}
    for name, analytic, form in (("fd", False, oc.SIMPLE), ("an", True, oc.SIMPLE), ("an_joseph", True, oc.JOSEPH)):
        eng = OracleEngine(B, dialect, 18, cov_form=form)
        eng.set_state(nom, rot, P, prev)
        ok = eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, rgt, SIZE, prm.r_pix, analytic=analytic)
        out[name] = (eng.get_state(), ok)
    assert np.array_equal(out["fd"][1], out["an"][1]) and out["an"][1][0] == 0 and out["an"][1][1:].all()
    e = parity_errors(out["an"][0], out["fd"][0])
    print(f"[oracle] correct_pixels analytic vs central differences, dialect {dialect} {'stereo' if stereo else 'left'}: literal {e['literal']:.2e} "
          f"sigma-aware {e['sigma']:.2e} cov block-wise {e['cov_block']:.2e}")
    assert e["literal"] < 1e-7 and e["sigma"] < 1e-6 and e["cov_block"] < 1e-6
    assert e["literal"] > 1e-13                                      # they ARE different computations
    j = parity_errors(out["an_joseph"][0], out["an"][0])
    print(f"[oracle]   Joseph vs simple form of the same update: literal {j['literal']:.2e} cov block-wise {j['cov_block']:.2e}")
    assert j["literal"] == 0 and j["cov_block"] < 1e-9               # same gain, same injection; (I-KH)P(I-KH)'+KRK' == (I-KH)P to rounding
