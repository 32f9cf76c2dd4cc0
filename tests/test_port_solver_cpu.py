"""CPU suite: the forward flat-port projection as the measurement kernels solve it (csrc/ekf_meas.hpp::pixel_fold_marker) --
closed-form thin-port start, taken twice (in fp32 since round 6: port_start_f32), then ONE Halley step in double (two for fp64 records) with L_t and L_z carried along the
step to first order -- restated in numpy and compared with a converged Newton solution of the same port equation
    rho = L(t) = d_air t + d_glass tan(theta_glass) + z_w tan(theta_water),  t = tan(theta_air)
(the inverse of the ray construction of RefractionTriangulation, vision.cpp:505-552) over the whole admitted field of view: water depths
0.25 .. 2 m, rays up to 0.9 of the critical angle in water (tangent 3.16 in air -- far outside any lens; the recordings and the test
scenes stay below 0.8).  What the kernel comments claim is asserted here, by range of the tangent: the start (second pass) is within
1.2e-4 / 6e-4 / 3.5e-3 / 2e-2 of the root for tangents below 1.4 / 2 / 2.5 / 3.16; one Halley step leaves 2e-13 / 8e-12 / 1e-9 / 5e-8
(relative: 1.4e-8 at the very rim -- still below the fp32 rounding of the image point it is compared with), two reach double
precision; the carried L_t, L_z are the derivatives at the new point to second order in the step."""
import numpy as np

from fbus_ekf import capi


def _consts():
    prm = capi.default_params(capi.DIALECT_MATLAB)
    return dict(a0=prm.n_air / prm.n_glass, a1=prm.n_air / prm.n_water, d_air=prm.d_air, d_glass=prm.d_glass)


def _port(c, zw, t):
    """L, L_t, L_tt, L_z, L_zt at tangent t, as the kernel's port_eval_n computes them (round 6: in the tangent --
    tan(theta_m) = a_m t / sqrt(1 + (1 - a_m^2) t^2): two reciprocal square roots and no sine)"""
    qg, qw = 1.0 - c["a0"] ** 2, 1.0 - c["a1"] ** 2
    ig, iw = 1.0 / np.sqrt(1.0 + qg * t * t), 1.0 / np.sqrt(1.0 + qw * t * t)
    G, W = c["d_glass"] * c["a0"], zw * c["a1"]
    L = t * (c["d_air"] + G * ig + W * iw)
    Lt = c["d_air"] + G * ig ** 3 + W * iw ** 3
    Ltt = -3 * t * (qg * G * ig ** 5 + qw * W * iw ** 5)
    return L, Lt, Ltt, c["a1"] * t * iw, c["a1"] * iw ** 3


def _port_sine(c, zw, t):
    """the same five values through s = sin(theta_air) and 1 / cos(theta_m) (rounds 4-5's form; the textbook statement of the port)"""
    r = 1.0 / np.sqrt(1.0 + t * t)
    s = t * r
    icg = 1.0 / np.sqrt(1.0 - c["a0"] ** 2 * s * s)
    icw = 1.0 / np.sqrt(1.0 - c["a1"] ** 2 * s * s)
    G, W = c["d_glass"] * c["a0"], zw * c["a1"]
    L = c["d_air"] * t + s * (G * icg + W * icw)
    q3 = G * icg ** 3 + W * icw ** 3
    Lt = c["d_air"] + r ** 3 * q3
    Ltt = 3 * r ** 5 * (-t * q3 + s * r * (G * c["a0"] ** 2 * icg ** 5 + W * c["a1"] ** 2 * icw ** 5))
    Lz = c["a1"] * s * icw
    Lzt = c["a1"] * r ** 3 * icw ** 3
    return L, Lt, Ltt, Lz, Lzt


def _start(c, zw, rho, dtype=np.float32):
    """the kernel's start (port_start_f32: fp32 arithmetic since round 6 -- the start is good to 1e-4 by construction): thin port with an
    effective depth, then once more for the water alone with the port's offsets at t0"""
    f = dtype
    a1 = c["a1"]
    q1, a12, qg = f(1.0 - a1 * a1), f(a1 * a1), f(1.0 - c["a0"] ** 2)
    c0, dair, Gd0 = f((c["d_air"] + c["d_glass"] * c["a0"]) / a1), f(c["d_air"]), f(c["d_glass"] * c["a0"])
    zw, rho = zw.astype(f), rho.astype(f)
    u = rho / (zw + c0)
    t0 = u / np.sqrt(np.maximum(a12 - q1 * u * u, f(1e-6)))
    wg = f(1) / np.sqrt(f(1) + qg * t0 * t0)                             # tan(theta_glass) = a0 t0 wg
    u1 = np.maximum((rho - t0 * (dair + Gd0 * wg)) / zw, f(0))
    t1 = u1 / np.sqrt(np.maximum(a12 - q1 * u1 * u1, f(1e-6)))
    return t0.astype(np.float64), t1.astype(np.float64)


def _halley(c, zw, rho, t):
    L, Lt, Ltt, Lz, Lzt = _port(c, zw, t)
    f = L - rho
    dt = -2 * f * Lt / np.maximum(2 * Lt * Lt - f * Ltt, Lt * Lt)
    return np.maximum(t + dt, 0.0), Lt + Ltt * dt, Lz + Lzt * dt


def _grid(c):
    rng = np.random.default_rng(4)
    n = 400000
    zw = rng.uniform(0.25, 2.0, n)
    # in water no ray leans further than asin(a1); the kernels admit 0.9 of that tangent (the visibility test of the fold)
    tw_max = 0.9 * c["a1"] / np.sqrt(1 - c["a1"] ** 2)
    tw = rng.uniform(0.0, tw_max, n)                                      # tan(theta_water)
    sw = tw / np.sqrt(1 + tw * tw)
    sa = sw / c["a1"]                                                     # Snell back to air
    t_true = sa / np.sqrt(1 - sa * sa)
    rho = _port(c, zw, t_true)[0]
    return zw, rho, t_true


def test_closed_form_start_and_one_halley_step():
    c = _consts()
    zw, rho, t_true = _grid(c)
    assert t_true.max() > 2.5                                             # the rim of the field of view is in the sample
    t0, t1 = _start(c, zw, rho)
    e0, e1 = np.abs(t0 - t_true), np.abs(t1 - t_true)
    th, Lt_c, Lz_c = _halley(c, zw, rho, t1)
    eh = np.abs(th - t_true)
    bands = ((0.0, 1.4, 2e-3, 1.5e-4, 3e-13), (1.4, 2.0, 7e-3, 7e-4, 1e-11), (2.0, 2.5, 3.5e-2, 4e-3, 1.2e-9), (2.5, 4.0, 1.3e-1, 2.2e-2, 6e-8))
    for lo, hi, b0, b1, bh in bands:
        m = (t_true >= lo) & (t_true < hi)
        print(f"[port] tangent {lo:.1f} .. {hi:.1f}: start first pass {e0[m].max():.2e}, second pass {e1[m].max():.2e}, one Halley step {eh[m].max():.2e}")
        assert e0[m].max() < b0 and e1[m].max() < b1 and eh[m].max() < bh, (lo, hi)
    assert (eh / np.maximum(t_true, 1.0)).max() < 2e-8                    # below fp32 rounding (6e-8) everywhere
    th2, _, _ = _halley(c, zw, rho, th)
    e2 = np.abs(th2 - t_true) / np.maximum(t_true, 1.0)
    print(f"[port] two Halley steps: max {e2.max():.2e}")
    assert e2.max() < 1e-14
    # the derivatives the Jacobian uses: evaluated in front of the step, carried along it to first order
    _, Lt, _, Lz, _ = _port(c, zw, th)
    dLt, dLz = np.abs(Lt_c / Lt - 1), np.abs(Lz_c - Lz)
    print(f"[port] carried L_t relative {dLt.max():.2e}, L_z absolute {dLz.max():.2e}")
    inner = t_true < 2.0
    assert dLt[inner].max() < 2e-5 and dLz[inner].max() < 2e-5            # second order in a step of <= 7e-4
    # and without the carry they would be first order in the step: the carry is what keeps the Jacobian consistent
    _, Lt_s, _, Lz_s, _ = _port(c, zw, t1)
    assert np.abs(Lt_s / Lt - 1)[inner].max() > 10 * dLt[inner].max()


def test_the_tangent_form_is_the_port_equation():
    """port_eval_n's form against the sine form (the same function written the textbook way): values and derivatives to rounding"""
    c = _consts()
    zw, rho, t_true = _grid(c)
    t = t_true * (1 + 1e-4)
    for a, b, name in zip(_port(c, zw, t), _port_sine(c, zw, t), ("L", "L_t", "L_tt", "L_z", "L_zt")):
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
        print(f"[port] tangent form vs sine form, {name}: {err.max():.2e}")
        assert err.max() < 5e-13, name                                   # (the sine form cancels in 1 - a^2 s^2 towards the rim)
    # and the fp32 start is the double start to fp32 rounding: nothing the Halley step could notice
    t1_32, t1_64 = _start(c, zw, rho)[1], _start(c, zw, rho, np.float64)[1]
    assert np.abs(t1_32 - t1_64).max() < 2e-5 * max(1.0, t_true.max())


def test_forward_projection_inverts_the_ray_construction():
    """the solved tangent reproduces rho: L(t) - rho at rounding level relative to rho + the port's own offsets"""
    c = _consts()
    zw, rho, _ = _grid(c)
    t = _start(c, zw, rho)[1]
    for _ in range(2):
        t = _halley(c, zw, rho, t)[0]
    res = np.abs(_port(c, zw, t)[0] - rho) / (rho + c["d_air"])
    assert res.max() < 1e-13
