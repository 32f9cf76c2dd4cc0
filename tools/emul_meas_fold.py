#!/usr/bin/env python3
"""numpy model of the round-4 reprojection-row update (the algebra of csrc/ekf_meas.hpp), checked against the dense fp64 update:
  * rows regrouped per corner:  a = (J_q Mc)'  (IMU frame),  c = a x ru,  h_p = -R a,  h_theta = c
      Lam_pp = R S_aa R', Lam_pt = -R S_ac, Lam_tt = S_cc, b_p = -R s_a, b_t = s_c      (sums in double)
  * Jacobian as a combination of three vectors:  J_q = alpha_q e' + beta_q n' + k g_q'
  * 6 x 6 stage in double: Lam = Lc Lc' (dropped pivots), Mt = I + Lc' P_JJ Lc = Cm Cm', Z = Lc Cm^-T, Sinv = Z Z',
      G = I - P_JJ Sinv, m = G' b
  * update in fp32: dx = P(:,J) m ; W_c = Z' x_c ; P_rr -= W W' ; P_J,: = G P_J,:
    python tools/emul_meas_fold.py [room|wall] [stereo]
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emul_pixels_precision as E
from emul_pixels_precision import J6, r32

f32, f64 = np.float32, np.float64


def fold_regrouped(nom, rot, ids, left, right, prm, size, loop_dtype=f32, tol=1e-3, start="meas"):
    """-> Lam (B,6,6), b (B,6) in double, built the way the kernel builds them"""
    from fbus_ekf import synth
    B, M = ids.shape
    R_IL, P_IL, _ = synth.camera_constants(prm)
    mids, mpos, mquat = synth.marker_table(prm)
    slot_of = {int(i): k for k, i in enumerate(mids)}
    slot = np.array([[slot_of.get(int(i), -1) for i in row] for row in ids])
    vism = slot >= 0
    sl = np.where(vism, slot, 0)
    c = np.array([[0, 0, 0], [0, size, 0], [size, size, 0], [size, 0, 0.0]])
    Rm = synth.q2R(mquat)
    cw = mpos[sl][:, :, None, :] + np.einsum("bmij,kj->bmki", Rm[sl], c)
    p = nom[:, 0:3]; R = rot.reshape(B, 3, 3)
    u = cw - p[:, None, None, :]
    ru = np.einsum("bji,bmkj->bmki", R, u)                                 # R'u
    tI = ru - P_IL
    X = np.einsum("ij,bmkj->bmki", R_IL, tI)
    vc = E.vis_consts(f64)
    F = np.diag([-1.0, -1, 1])
    XL = X * np.array([-1, -1, 1.0])
    cams = [(XL, F @ R_IL, left.reshape(B, M, 4, 2))]
    if right is not None:
        XR = np.einsum("ij,bmkj->bmki", vc["R_RL_inv"], XL - vc["P_LR"])
        cams.append((XR, vc["R_RL_inv"] @ F @ R_IL, right.reshape(B, M, 4, 2)))
    n = vc["n"]
    Saa = np.zeros((B, 3, 3)); Sac = np.zeros((B, 3, 3)); Scc = np.zeros((B, 3, 3)); sa = np.zeros((B, 3)); sc = np.zeros((B, 3))
    its = []
    for Xc, Mc, y in cams:
        z = Xc @ n
        lat = Xc - z[..., None] * n
        rho = np.sqrt((lat * lat).sum(-1))
        zw = z - vc["d_air"] - vc["d_glass"]
        lim = 0.9 * zw * vc["a1"]
        ok = (zw > 0) & (rho * rho * (1 - vc["a1"] ** 2) < lim * lim) & vism[:, :, None]
        # Newton loop in loop_dtype from the measured ray, until the step is <= tol * t
        vcl = E.vis_consts(loop_dtype)
        D = np.concatenate([y, np.ones(y.shape[:-1] + (1,))], -1)
        zz = D @ n
        lm = D - zz[..., None] * n
        t = (np.sqrt((lm * lm).sum(-1)) / zz).astype(loop_dtype) if start == "meas" else \
            (rho / (vc["d_air"] + vc["a0"] * vc["d_glass"] + vc["a1"] * zw)).astype(loop_dtype)
        rho_l, zw_l = rho.astype(loop_dtype), np.where(ok, zw, 1.0).astype(loop_dtype)
        conv = ~ok
        nit = 0
        for it in range(12):
            L, Lt, Lz = E.port_ray(vcl, zw_l, t)
            dt = (rho_l - L) / Lt
            tn = np.maximum(t + dt, loop_dtype(0))
            t = np.where(conv, t, tn)
            conv = conv | ~(np.abs(dt) > loop_dtype(tol) * tn)
            nit += 1
            if conv.all():
                break
        its.append(nit)
        # final evaluation in double + one correction
        t = t.astype(f64)
        zwd = np.where(ok, zw, 1.0)
        L, Lt, Lz = E.port_ray(vc, zwd, t)
        t = np.maximum(t + (rho - L) / Lt, 0)
        irho = 1.0 / rho
        k = t * irho
        e = lat * irho[..., None]
        Dd = n + k[..., None] * lat
        iDz = 1.0 / Dd[..., 2]
        uv = Dd[..., :2] * iDz[..., None]
        res = y - uv
        eM = e @ Mc                                                      # e' Mc
        nM = n @ Mc
        for q in range(2):
            ge = (e[..., q] - uv[..., q] * e[..., 2]) * iDz
            gn = (n[q] - uv[..., q] * n[2]) * iDz
            alpha = ge * (1.0 / Lt - k)
            beta = -(Lz / Lt) * ge - k * gn
            gM = (Mc[q] - uv[..., q, None] * Mc[2]) * iDz[..., None]
            a = alpha[..., None] * eM + beta[..., None] * nM + k[..., None] * gM      # (B,M,4,3) = j' in the IMU frame
            a = np.where(ok[..., None], a, 0.0)
            cc = np.cross(a, ru)
            r = np.where(ok, res[..., q], 0.0)
            Saa += np.einsum("bmki,bmkj->bij", a, a); Sac += np.einsum("bmki,bmkj->bij", a, cc)
            Scc += np.einsum("bmki,bmkj->bij", cc, cc)
            sa += np.einsum("bmki,bmk->bi", a, r); sc += np.einsum("bmki,bmk->bi", cc, r)
    w = 1.0 / prm.r_pix
    Lam = np.zeros((B, 6, 6)); b = np.zeros((B, 6))
    Lam[:, :3, :3] = w * np.einsum("bij,bjk,blk->bil", R, Saa, R)
    Lam[:, :3, 3:] = -w * np.einsum("bij,bjk->bik", R, Sac)
    Lam[:, 3:, :3] = np.swapaxes(Lam[:, :3, 3:], 1, 2)
    Lam[:, 3:, 3:] = w * Scc
    b[:, :3] = -w * np.einsum("bij,bj->bi", R, sa)
    b[:, 3:] = w * sc
    return Lam, b, its


def chol_psd(A, tiny=4e-15):
    """lower Lc with A = Lc Lc'; a pivot that is not clearly positive relative to its original diagonal is dropped (zero column)"""
    A = A.copy(); B = A.shape[0]
    L = np.zeros_like(A)
    d0 = np.einsum("bii->bi", A).copy()
    for a in range(6):
        piv = A[:, a, a]
        ok = piv > tiny * d0[:, a]
        s = np.where(ok, 1.0 / np.sqrt(np.where(ok, piv, 1.0)), 0.0)
        L[:, a:, a] = A[:, a:, a] * s[:, None]
        A[:, a:, a:] -= L[:, a:, a, None] * L[:, None, a:, a]
    return L


def solve6(PJJ, Lam, b):
    Lc = chol_psd(Lam)
    Y = np.einsum("bij,bjk->bik", PJJ, Lc)
    Mt = np.eye(6) + np.einsum("bji,bjk->bik", Lc, Y)
    Cm = np.linalg.cholesky(Mt)
    Z = np.swapaxes(np.linalg.solve(Cm, np.swapaxes(Lc, 1, 2)), 1, 2)       # Z Cm' = Lc  ->  Z = Lc Cm^-T
    Sinv = np.einsum("bij,bkj->bik", Z, Z)
    G = np.eye(6) - np.einsum("bij,bjk->bik", PJJ, Sinv)
    m = np.einsum("bji,bj->bi", G, b)
    return G, Z, m


def update_direct(P32, G, Z, m, u=f32):
    B, N = P32.shape[:2]
    rr = np.array([i for i in range(N) if i not in J6])
    P = P32.astype(u).copy()
    G = G.astype(u); Z = Z.astype(u); m = m.astype(u)
    PJ = P[:, :, J6]                                                       # (B,N,6): column c of P restricted to J = x_c
    dx = np.einsum("bnj,bj->bn", PJ, m).astype(u)
    W = np.einsum("bnj,bjk->bnk", PJ[:, rr], Z).astype(u)                  # W_c = Z' x_c
    Prr = P[:, rr[:, None], rr[None, :]] - np.einsum("bnk,bmk->bnm", W, W).astype(u)
    GP = np.einsum("bij,bnj->bin", G, PJ).astype(u)                        # G P(J, :)
    Pn = P.copy()
    Pn[:, rr[:, None], rr[None, :]] = Prr
    Pn[:, J6, :] = GP
    Pn[:, :, J6] = np.swapaxes(GP, 1, 2)
    JJ = GP[:, :, J6]
    Pn[:, J6[:, None], J6[None, :]] = (JJ + np.swapaxes(JJ, 1, 2)) * u(0.5)
    return dx.astype(f64), Pn.astype(f64)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "room"
    stereo = "stereo" in sys.argv
    from fbus_ekf import capi, synth
    prm = capi.default_params(0)
    if which == "wall":
        from test_pixels_gpu import _wall_map
        from replay_ref import OracleEngine
        from util import pixel_scene
        size, M, B = 0.15, 16, 96
        prm.marker_size = size
        probe = OracleEngine(1, 0, 18)
        _wall_map(prm, probe.orc.prm, size)
        nom, _, ids, left, right = pixel_scene(B, M, prm, size, seed=9, noise=5e-4, depth=(1.2, 1.8))
        nom[:, 0:3] += np.random.default_rng(10).normal(0, 0.003, (B, 3))
        nom = r32(nom); left = r32(left); right = r32(right)
        rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
        P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
    else:
        from test_pixels_gpu import _scene, SIZE
        size, M, B = SIZE, 4, 192
        prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, seed=11)
    rgt = right if stereo else None
    w = 1.0 / prm.r_pix
    h0, res0, val = E.build_rows(nom, rot, ids, left, rgt, prm, size, f64, f64, iters=10)
    dx0, P0 = E.exact_update(P, h0, res0, val, w)
    Lam0, b0 = E.fold(h0, res0, val, w, f64)
    for loop_dtype, tol, start in ((f64, 1e-12, "meas"), (f32, 1e-3, "meas"), (f32, 1e-3, "paraxial"), (f32, 3e-3, "meas")):
        Lam, b, its = fold_regrouped(nom, rot, ids, left, rgt, prm, size, loop_dtype, tol, start)
        eL = np.abs(Lam - Lam0).max() / np.abs(Lam0).max(); eb = np.abs(b - b0).max() / np.abs(b0).max()
        G, Z, m = solve6(r32(P)[:, J6][:, :, J6], Lam, b)
        dx, Pn = update_direct(r32(P), G, Z, m, f32)
        print(f"loop {loop_dtype.__name__} tol {tol:g} start {start:8s} its {its}: Lam err {eL:.1e} b err {eb:.1e}   {E.figures(dx, Pn, dx0, P0, nom)}")
    dx, Pn = update_direct(r32(P), G, Z, m, f64)
    print(f"  same, update in fp64:   {E.figures(dx, Pn, dx0, P0, nom)}")




def halley_study():
    """iterations of Halley's method on the port equation from the measured ray, by exit criterion (numpy, double)"""
    from fbus_ekf import capi, synth
    from test_pixels_gpu import _scene, SIZE
    prm, nom, rot, P, prev, ids, left, right = _scene(192, 4, 0, seed=11)
    vc = E.vis_consts(f64)
    rng = np.random.default_rng(0)
    zw = rng.uniform(0.3, 1.8, 200000)
    tt = rng.uniform(0.01, 1.4, 200000)                        # true tan(theta_air)
    rho = E.port_ray(vc, zw, tt)[0]
    for innov in (3e-3, 1e-2, 3e-2, 1e-1):
        t0 = tt * (1 + rng.normal(0, innov, tt.shape))
        for method in ("newton", "halley"):
            t = t0.copy()
            errs = []
            for it in range(4):
                one = 1.0
                r = one / np.sqrt(one + t * t); s = t * r; s2 = s * s
                icg = one / np.sqrt(one - vc["a0"] ** 2 * s2); icw = one / np.sqrt(one - vc["a1"] ** 2 * s2)
                G, W = vc["d_glass"] * vc["a0"], zw * vc["a1"]
                L = vc["d_air"] * t + s * (G * icg + W * icw)
                q3 = G * icg ** 3 + W * icw ** 3
                Lt = vc["d_air"] + r ** 3 * q3
                Ltt = -3 * t * r ** 5 * q3 + 3 * s * r ** 6 * (G * vc["a0"] ** 2 * icg ** 5 + W * vc["a1"] ** 2 * icw ** 5)
                f = L - rho
                dt = -f / Lt if method == "newton" else -2 * f * Lt / (2 * Lt * Lt - f * Ltt)
                t = np.maximum(t + dt, 0)
                errs.append(np.abs(t / tt - 1).max())
            print(f"start error sigma {innov:g} {method:7s}: max relative error after 1..4 steps " + " ".join(f"{e:.1e}" for e in errs))


if __name__ == "__main__":
    if "halley" in sys.argv:
        halley_study()
    else:
        main()
