"""CPU suite: the algebra behind round 6's reprojection fold (csrc/ekf_meas.hpp), restated in numpy -- what the GPU parity tests confirm
end to end is pinned here identity by identity:
  * the rows of a projection rotated into the radial / tangential direction of the image point leave sum a a' and sum a res unchanged
    (isotropic pixel noise), and in the camera frame they are j_rad = (e0 / L_t, e1 / L_t, -c2), j_tan = (-k e1, k e0, 0);
  * sums folded in the camera frame (rows j, theta parts c' = j x Y, Y = M^-T r) give the IMU-frame information matrix through
    PixAcc::to_imu_frame + finish() with R M' in place of R -- for a general (non-orthogonal) M: (M'j) x r = adj(M) (j x M^-T r);
  * tri_corners_refractive's rays in the tangent, unnormalised, with the left one at z = 1: the mid-point of two rays does not depend on
    their lengths, and the tangent form is the ray of vision.cpp:505-552."""
import numpy as np

rng = np.random.default_rng(3)


def _skew(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0.0]])


def _info(rows_a, res, rs, R, w):
    """Lam, b of rows [-(R a)' , (a x r)'] as PixAcc + finish() form them: Lam_pp = w R S_aa R', Lam_pt = -w R S_ac, Lam_tt = w S_cc"""
    Saa = sum(np.outer(a, a) for a in rows_a)
    Sac = sum(np.outer(a, np.cross(a, r)) for a, r in zip(rows_a, rs))
    Scc = sum(np.outer(np.cross(a, r), np.cross(a, r)) for a, r in zip(rows_a, rs))
    sa = sum(a * e for a, e in zip(rows_a, res))
    sc = sum(np.cross(a, r) * e for a, r, e in zip(rows_a, rs, res))
    Lam = np.block([[w * R @ Saa @ R.T, -w * R @ Sac], [(-w * R @ Sac).T, w * Scc]])
    return Lam, np.concatenate([-w * R @ sa, w * sc])


def test_radial_tangential_rows_in_the_camera_frame():
    M = np.linalg.qr(rng.normal(size=(3, 3)))[0] + 1e-3 * rng.normal(size=(3, 3))        # a calibration's rotation: orthogonal to 1e-3 only
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    adj = np.linalg.det(M) * np.linalg.inv(M)
    MiT = np.linalg.inv(M).T
    rows_a, res_a, rs = [], [], []                      # the IMU-frame rows as rounds 4-5 formed them: a = M' J_r
    Sjj, Sjc, Scc, sj, sc = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3), np.zeros(3)
    for _ in range(16):
        e = rng.normal(size=2); e /= np.linalg.norm(e)
        iLt, k, c2 = rng.uniform(0.5, 2), rng.uniform(0.5, 2), rng.uniform(-0.5, 0.5)
        c1 = iLt - k
        r = rng.normal(size=3) + np.array([0, 0, 1.5])
        res = rng.normal(0, 1e-3, 2)
        J = np.array([[c1 * e[0] * e[0] + k, c1 * e[0] * e[1], -c2 * e[0]], [c1 * e[1] * e[0], c1 * e[1] * e[1] + k, -c2 * e[1]]])
        for rr in range(2):
            rows_a.append(M.T @ J[rr]); res_a.append(res[rr]); rs.append(r)
        # the kernel's camera-frame rows
        jrad, jtan = np.array([iLt * e[0], iLt * e[1], -c2]), np.array([-k * e[1], k * e[0], 0.0])
        rrad, rtan = e @ res, e[0] * res[1] - e[1] * res[0]
        assert np.allclose(np.outer(jrad, jrad) + np.outer(jtan, jtan), J.T @ J, atol=1e-14)
        assert np.allclose(jrad * rrad + jtan * rtan, J.T @ res, atol=1e-16)
        Y = MiT @ r
        for j, e_ in ((jrad, rrad), (jtan, rtan)):
            c = np.cross(j, Y)
            Sjj += np.outer(j, j); Sjc += np.outer(j, c); Scc += np.outer(c, c); sj += j * e_; sc += c * e_
    w = 1e6
    want_L, want_b = _info(rows_a, res_a, rs, R, w)
    # PixAcc::to_imu_frame (the adj(M) parts) + finish() with RM = R M'
    Sac2, Scc2, sc2 = Sjc @ adj.T, adj @ Scc @ adj.T, adj @ sc
    RM = R @ M.T
    got_L = np.block([[w * RM @ Sjj @ RM.T, -w * RM @ Sac2], [(-w * RM @ Sac2).T, w * Scc2]])
    got_b = np.concatenate([-w * RM @ sj, w * sc2])
    scale = np.sqrt(np.outer(np.diag(want_L), np.diag(want_L)))
    assert (np.abs(got_L - want_L) / scale).max() < 1e-13
    assert np.abs(got_b - want_b).max() < 1e-13 * np.abs(want_b).max()
    # and the identity itself, for a matrix that is nowhere near orthogonal
    A = rng.normal(size=(3, 3)); j = rng.normal(size=3); r = rng.normal(size=3)
    assert np.allclose(np.cross(A.T @ j, r), np.linalg.det(A) * np.linalg.inv(A) @ np.cross(j, np.linalg.inv(A).T @ r), atol=1e-12)


def _ray_reference(x, y, a0, a1g, d_air, d_glass):
    """vision.cpp:505-552 for the port square to the camera: unit ray in the water and exit point on the outer glass face"""
    r0 = np.array([x, y, 1.0]); r0 /= np.linalg.norm(r0)
    n = np.array([0, 0, 1.0])
    v0 = r0 @ n
    r1 = a0 * r0 + (np.sqrt(1 - a0 * a0 * (1 - v0 * v0)) - a0 * v0) * n
    v1 = r1 @ n
    r2 = a1g * r1 + (np.sqrt(1 - a1g * a1g * (1 - v1 * v1)) - a1g * v1) * n
    return r2, d_air / v0 * r0 + d_glass / v1 * r1


def _midpoint(PL, rL, PR, rR):
    cr = np.cross(rL, rR); dP = PR - PL; d3 = cr @ cr
    t1 = np.linalg.det(np.column_stack([cr, dP, rR])) / d3
    t2 = -np.linalg.det(np.column_stack([cr, rL, dP])) / d3
    return 0.5 * (PL + t1 * rL + PR + t2 * rR)


def test_triangulation_rays_in_the_tangent():
    n_air, n_glass, n_water, d_air, d_glass = 1.0, 1.49, 1.333, 0.002, 0.02
    a0, a1g = n_air / n_glass, n_glass / n_water
    a = a0 * a1g
    R_RL = np.linalg.qr(rng.normal(size=(3, 3)))[0]; R_RL *= np.sign(np.linalg.det(R_RL))
    P_LR = np.array([0.12, 0.001, -0.002])
    for _ in range(200):
        xl, yl, xr, yr = rng.uniform(-0.8, 0.8, 4)
        r2L, P1L = _ray_reference(xl, yl, a0, a1g, d_air, d_glass)
        r2R, P1R = _ray_reference(xr, yr, a0, a1g, d_air, d_glass)
        want = _midpoint(P1L, r2L, R_RL @ P1R + P_LR, R_RL @ r2R)
        # the kernel's form: t^2 = x^2 + y^2, two reciprocal square roots per ray, the left ray at z = 1, the right one from u = a R_RL(:, 0:1)(x, y)
        def pieces(x, y):
            t2 = x * x + y * y
            return 1 / np.sqrt(1 + (1 - a0 * a0) * t2), 1 / np.sqrt(1 + (1 - a * a) * t2), 1 + (1 - a * a) * t2
        igL, iwL, _ = pieces(xl, yl)
        igR, iwR, xwR = pieces(xr, yr)
        rL = np.array([a * iwL * xl, a * iwL * yl, 1.0])
        PL = np.array([(d_air + d_glass * a0 * igL) * xl, (d_air + d_glass * a0 * igL) * yl, d_air + d_glass])
        u = a * (R_RL[:, 0] * xr + R_RL[:, 1] * yr)
        rR = u + R_RL[:, 2] * (xwR * iwR)
        PR = ((d_air + d_glass * a0 * igR) / a) * u + (R_RL[:, 2] * (d_air + d_glass) + P_LR)
        assert np.allclose(np.cross(rL, r2L), 0, atol=1e-14) and np.allclose(PL, P1L, atol=1e-15)      # same ray, same exit point
        assert np.allclose(np.cross(rR, R_RL @ r2R), 0, atol=1e-14) and np.allclose(PR, R_RL @ P1R + P_LR, atol=1e-15)
        assert np.allclose(_midpoint(PL, rL, PR, rR), want, atol=1e-12)
