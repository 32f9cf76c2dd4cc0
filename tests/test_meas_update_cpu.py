"""CPU suite: the 6 x 6 stage and the covariance update of the round-4 measurement kernels (csrc/ekf_meas.hpp::info_solve,
direct_update) restated in numpy and compared with the dense Kalman update the reference performs (MeasureUpdate.m:84-102:
K = P H' (H P H' + R)^-1, P <- (I - K H) P, dx = K r).  The rows of a corner / pixel update touch only the position and attitude
columns J, so H = [H_J 0] and everything follows from the 6 x 6 information matrix Lam = H_J' R^-1 H_J and vector b = H_J' R^-1 r:
    Lam = Lc Lc' (pivots that carry no information dropped),  M = I + Lc' P_JJ Lc = C C',  Z = Lc C^-T,  Sinv = Z Z',
    G = I - P_JJ Sinv,  m = G' b;      dx = P(:, J) m,   P_rr -= x_a' Sinv x_c,   P(J, :) <- G P(J, :).
Asserted: (1) in double this IS the dense update (1e-8 block-wise against a 60-digit evaluation of the dense form), also when Lam is rank deficient (a single marker seen edge-on: 2 of 6
directions unobserved) where (H P H' + R)^-1 through Lam^-1 does not exist; (2) with the covariance STORED and UPDATED in float32
(what the fp32 kernels do; the 6 x 6 stage in double) the posterior matches the double one to 1e-6 block-wise although the update
shrinks the pose variances by four decades -- the form subtracts nothing on the rows the measurement shrinks; (3) the textbook
P - K H P in float32 does NOT (that is what round 3's six rank-1 passes amounted to and why their gates had to be widened)."""
import numpy as np

J = np.array([0, 1, 2, 6, 7, 8])
f32, f64 = np.float32, np.float64


def _chol_psd(A, tiny=4e-15):
    A = A.copy()
    L = np.zeros_like(A)
    d0 = np.diag(A).copy()
    for a in range(6):
        piv = A[a, a]
        s = 1.0 / np.sqrt(piv) if piv > tiny * d0[a] else 0.0
        L[a:, a] = A[a:, a] * s
        A[a:, a:] -= np.outer(L[a:, a], L[a:, a])
    return L


def _solve6(PJJ, Lam, b):
    Lc = _chol_psd(Lam)
    Cm = np.linalg.cholesky(np.eye(6) + Lc.T @ PJJ @ Lc)
    Z = np.linalg.solve(Cm, Lc.T).T                                        # Z Cm' = Lc
    Sinv = Z @ Z.T
    G = np.eye(6) - PJJ @ Sinv
    return G, Sinv, G.T @ b


def _direct_update(P, G, Sinv, m, u):
    """the kernel's update in arithmetic type u on a covariance stored in u"""
    N = P.shape[0]
    rr = np.array([i for i in range(N) if i not in J])
    P = P.astype(u)
    G, Sinv, m = G.astype(u), Sinv.astype(u), m.astype(u)
    X = P[J, :]                                                            # x_c = P(J, c), all columns
    dx = (P[:, J] @ m).astype(u)
    Pn = P.copy()
    Xr = X[:, rr]
    Pn[np.ix_(rr, rr)] = (P[np.ix_(rr, rr)] - (Xr.T @ (Sinv @ Xr)).astype(u)).astype(u)
    GP = (G @ X).astype(u)
    Pn[J, :] = GP
    Pn[:, J] = GP.T
    JJ = GP[:, J]
    Pn[np.ix_(J, J)] = np.triu(JJ) + np.triu(JJ, 1).T                      # the kernel writes the upper triangle of G P_JJ
    return dx.astype(f64), Pn.astype(f64)


def _case(rng, N, rows, sigma=1e-3, edge_on=False):
    A = np.eye(N) + 0.3 * rng.normal(size=(N, N))
    S = np.sqrt(np.array([1e-2] * 3 + [1e-2] * 3 + [1e-2] * 3 + [1e-4] * 3 + [1e-6] * 3 + [1e-1] * 3)[:N])
    P = S[:, None] * (A @ A.T) * S[None, :]
    P = (P + P.T) / 2
    HJ = rng.normal(size=(rows, 6))
    if edge_on:
        HJ[:, [2, 5]] = 0.0                                                # two directions nobody observes
    r = rng.normal(0, 3e-3, rows)
    return P, HJ, r, sigma ** 2


def _dense(P, HJ, r, rvar):
    """the reference's dense update, K = P H' (H P H' + R)^-1, P <- (I - K H) P, in 60-digit arithmetic (mpmath): in double the
    inverse (condition 1e5) times the cancellation of (I - K H) P (five decades) leaves only 3e-7 of the posterior"""
    import mpmath as mp
    N = P.shape[0]
    H = np.zeros((HJ.shape[0], N)); H[:, J] = HJ
    with mp.workdps(60):
        Pm, Hm = mp.matrix(P.tolist()), mp.matrix(H.tolist())
        S = Hm * Pm * Hm.T + mp.mpf(rvar) * mp.eye(len(r))
        K = Pm * Hm.T * mp.inverse(S)
        Pn = (mp.eye(N) - K * Hm) * Pm
        dx = K * mp.matrix(r.tolist())
        Kd = np.array([[float(K[i, j]) for j in range(K.cols)] for i in range(K.rows)])
        Pd = np.array([[float(Pn[i, j]) for j in range(N)] for i in range(N)])
        dxd = np.array([float(dx[i]) for i in range(N)])
    return dxd, (Pd + Pd.T) / 2, Kd, H


def _block_err(Pa, Pb):
    d = np.sqrt(np.diag(Pb))
    return np.abs((Pa - Pb) / (d[:, None] * d[None, :])).max()


def test_information_form_is_the_dense_update_in_double():
    rng = np.random.default_rng(0)
    for N in (18, 15):
        for rows, edge_on in ((8, False), (32, False), (48, False), (8, True), (2, False)):
            P, HJ, r, rvar = _case(rng, N, rows, edge_on=edge_on)
            Lam, b = HJ.T @ HJ / rvar, HJ.T @ r / rvar
            G, Sinv, m = _solve6(P[np.ix_(J, J)], Lam, b)
            dx, Pn = _direct_update(P, G, Sinv, m, f64)
            dx_ref, P_ref, _, _ = _dense(P, HJ, r, rvar)
            assert np.abs(dx - dx_ref).max() <= 1e-10 * max(1.0, np.abs(dx_ref).max()), (N, rows, edge_on)
            assert _block_err(Pn, P_ref) < 1e-8, (N, rows, edge_on, _block_err(Pn, P_ref))      # eps x the physical shrink (<= 1e6)
            assert np.linalg.eigvalsh(Pn).min() > 0


def test_float32_records_keep_the_posterior_where_the_textbook_form_loses_it():
    rng = np.random.default_rng(1)
    worst_direct, worst_textbook = 0.0, 0.0
    for trial in range(20):
        P, HJ, r, rvar = _case(rng, 18, 118, sigma=1e-3)                   # 16 slots in view: 118 rows at sigma_pix = 1e-3
        P32 = P.astype(f32).astype(f64)                                    # the record the kernel reads
        Lam, b = HJ.T @ HJ / rvar, HJ.T @ r / rvar
        G, Sinv, m = _solve6(P32[np.ix_(J, J)], Lam, b)                    # the 6 x 6 stage in double
        dx, Pn = _direct_update(P32, G, Sinv, m, f32)
        dx_ref, P_ref = _direct_update(P32, G, Sinv, m, f64)               # = the dense update (the test above)
        H = np.zeros((HJ.shape[0], 18)); H[:, J] = HJ
        K = P32 @ H.T @ np.linalg.inv(H @ P32 @ H.T + rvar * np.eye(len(r)))
        worst_direct = max(worst_direct, _block_err(Pn, P_ref))
        assert np.abs(dx - dx_ref).max() <= 2e-6 * max(np.abs(dx_ref).max(), 1e-3)
        d = np.sqrt(np.diag(Pn))
        assert np.linalg.eigvalsh(Pn / (d[:, None] * d[None, :])).min() > 0
        # the textbook form with the gain and the product rounded to float32
        KH = (K.astype(f32) @ H.astype(f32)).astype(f32)
        Pt = (P32.astype(f32) - (KH @ P32.astype(f32)).astype(f32)).astype(f64)
        worst_textbook = max(worst_textbook, _block_err((Pt + Pt.T) / 2, P_ref))
    print(f"[update] float32 records, 118 rows: one-shot form {worst_direct:.2e} block-wise, P - K H P in float32 {worst_textbook:.2e}")
    assert worst_direct < 2e-6
    assert worst_textbook > 50 * worst_direct


def test_constant_row_matrix_sums_expand_to_the_per_corner_sums():
    """csrc/ekf_meas.hpp::PixAcc::add_corner_const / expand_const: when every corner's rows share one N' (the corner-position update:
    N' = R_IL' R_IL), S_aa, S_ac and S_cc are linear in the count, sum r and sum r r'.  The formulas of expand_const transcribed,
    against the per-corner accumulation (add_corner: S_ac += N' [r]x, S_cc += [r]x' N' [r]x)."""
    rng = np.random.default_rng(3)
    A = rng.normal(size=(3, 3))
    Nm = A @ A.T

    def cross_mat(r):
        return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])

    rs, W, cnt = np.zeros(3), np.zeros((3, 3)), 0
    Saa, Sac, Scc = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 3))
    for _ in range(11):
        r = rng.normal(size=3)
        B = cross_mat(r)
        Saa += Nm; Sac += Nm @ B; Scc += B.T @ Nm @ B
        rs += r; W += np.outer(r, r); cnt += 1
    n00, n01, n02, n11, n12, n22 = Nm[0, 0], Nm[0, 1], Nm[0, 2], Nm[1, 1], Nm[1, 2], Nm[2, 2]
    w00, w01, w02, w11, w12, w22 = W[0, 0], W[0, 1], W[0, 2], W[1, 1], W[1, 2], W[2, 2]
    Scc_x = np.array([[n11 * w22 - 2 * n12 * w12 + n22 * w11, -n01 * w22 + n12 * w02 + n02 * w12 - n22 * w01, n01 * w12 - n11 * w02 - n02 * w11 + n12 * w01],
                      [0, n00 * w22 - 2 * n02 * w02 + n22 * w00, -n00 * w12 + n01 * w02 + n02 * w01 - n12 * w00],
                      [0, 0, n00 * w11 - 2 * n01 * w01 + n11 * w00]])
    Scc_x = Scc_x + np.triu(Scc_x, 1).T
    assert np.abs(Scc_x - Scc).max() < 1e-12 * np.abs(Scc).max()
    assert np.abs(Nm @ cross_mat(rs) - Sac).max() < 1e-12 * np.abs(Sac).max()
    assert np.abs(cnt * Nm - Saa).max() < 1e-12 * np.abs(Saa).max()


def test_square_port_row_is_the_general_row():
    """csrc/ekf_meas.hpp::pixel_fold_marker, NZ: with the port normal (0, 0, 1) the reprojection row
    a_r = alpha e'M + beta n'M + k (M_r - uv_r M_z),  alpha = c1 g_e, beta = -(c2 g_e + k g_n), g = (unit_r - uv_r unit_z) / D_z,
    loses its two k uv_r M_z terms:  a_r = c1 e_r (e'M) + k M_r - c2 e_r M_z  (D_z = 1, e_z = 0)."""
    rng = np.random.default_rng(5)
    M = rng.normal(size=(3, 3))
    n = np.array([0.0, 0.0, 1.0])
    for _ in range(50):
        lat = np.array([rng.normal(), rng.normal(), 0.0])
        rho = np.linalg.norm(lat)
        kk, c1, c2 = rng.uniform(0.3, 1.5), rng.normal(), rng.normal()
        e = lat / rho
        Dz = n[2] + kk * lat[2]
        uv = (n[:2] + kk * lat[:2]) / Dz
        eM, nM = e @ M, n @ M
        for r in range(2):
            ge, gn = (e[r] - uv[r] * e[2]) / Dz, (n[r] - uv[r] * n[2]) / Dz
            general = ge * c1 * eM - (c2 * ge + kk * gn) * nM + kk / Dz * (M[r] - uv[r] * M[2])
            square = c1 * e[r] * eM + kk * M[r] - c2 * e[r] * M[2]
            assert np.abs(general - square).max() < 1e-13 * max(1.0, np.abs(general).max())
