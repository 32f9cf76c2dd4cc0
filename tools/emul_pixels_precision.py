#!/usr/bin/env python3
"""CPU study (numpy, no GPU): which stage of the reprojection-row update (`fbus_ekf_correct_pixels`) sets its fp32 error.

Exact rows / residuals / posterior in float64 (cross-checked against the C oracle), then the device algorithm with each
stage switched between float32 and float64:
    geom   X_k, Hpp, Hpt of a corner            proj   flat-port projection + closed-form Jacobian
    fold   Lam = sum w h h', b = sum w h res     solve  the 6 x 6 algebra
    upd    the covariance / state update form:  'seq' six sequential rank-1 passes (round 3), 'oneshot' P - W W',
           'direct' J rows by the non-cancelling product G P_J, the rest by P - W W'
Prints literal / sigma-aware / block-wise covariance figures (tests/util.py metrics) per variant.
    python tools/emul_pixels_precision.py [wall|room] [stereo]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import oracle_capi as oc                                   # noqa: E402
from fbus_ekf import capi, synth                            # noqa: E402
from util import pixel_scene, cov_rel_err_blockwise, _BLOCKS, _SIGMA_IDX   # noqa: E402

r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
J6 = np.array([0, 1, 2, 6, 7, 8])


def vis_consts(dt):
    p = oc.vision_params()
    R_IL = np.array(list(p.R_IL)).reshape(3, 3); R_IR = np.array(list(p.R_IR)).reshape(3, 3)
    P_LI = np.array(list(p.P_LI)); P_RI = np.array(list(p.P_RI))
    R_RL = R_IL @ R_IR.T
    P_LR = P_LI - R_RL @ P_RI
    c = dict(R_RL_inv=np.linalg.inv(R_RL), P_LR=P_LR, a0=p.n_air / p.n_glass, a1=p.n_air / p.n_water,
             d_air=p.d_air, d_glass=p.d_glass, n=np.array(list(p.normal)))
    return {k: (np.asarray(v, dt) if isinstance(v, np.ndarray) else dt(v)) for k, v in c.items()}


def port_ray(vc, zw, t):
    one = t.dtype.type(1)
    r = one / np.sqrt(one + t * t)
    s = t * r
    s2 = s * s
    icg = one / np.sqrt(one - vc["a0"] * vc["a0"] * s2)
    icw = one / np.sqrt(one - vc["a1"] * vc["a1"] * s2)
    g = vc["d_glass"] * vc["a0"] * icg
    w = zw * vc["a1"] * icw
    L = vc["d_air"] * t + s * (g + w)
    Lt = vc["d_air"] + (g * icg * icg + w * icw * icw) * (r * r * r)
    Lz = vc["a1"] * s * icw
    return L, Lt, Lz


def project(vc, Xp, iters, t0=None, final_corr=False):
    """Xp (..., 3) in a camera's refraction frame -> uv (..., 2), J (..., 2, 3), ok; dtype of Xp"""
    dt = Xp.dtype.type
    n = vc["n"]
    z = Xp @ n
    lat = Xp - z[..., None] * n
    rho = np.sqrt((lat * lat).sum(-1))
    zw = z - vc["d_air"] - vc["d_glass"]
    lim = dt(0.9) * zw * vc["a1"]
    ok = (zw > 0) & ((rho * rho) * (dt(1) - vc["a1"] * vc["a1"]) < lim * lim)
    t = rho / (vc["d_air"] + vc["a0"] * vc["d_glass"] + vc["a1"] * zw) if t0 is None else t0.astype(Xp.dtype)
    for _ in range(iters):
        L, Lt, Lz = port_ray(vc, zw, t)
        t = np.maximum(t + (rho - L) / Lt, dt(0))
    L, Lt, Lz = port_ray(vc, zw, t)
    if final_corr:
        t = np.maximum(t + (rho - L) / Lt, dt(0))
    irho = dt(1) / rho
    k = t * irho
    e = lat * irho[..., None]
    D = n + k[..., None] * lat
    iDz = dt(1) / D[..., 2]
    uv = D[..., :2] * iDz[..., None]
    c1 = dt(1) / Lt - k
    c2 = Lz / Lt
    I = np.eye(3, dtype=Xp.dtype)
    dD = c1[..., None, None] * e[..., :, None] * e[..., None, :] - c2[..., None, None] * e[..., :, None] * n[None, :] \
        + k[..., None, None] * (I - n[:, None] * n[None, :])
    J = (dD[..., :2, :] - uv[..., :, None] * dD[..., 2:3, :]) * iDz[..., None, None]
    return uv, J, ok


def build_rows(nom, rot, ids, left, right, prm, size, dt_geom, dt_proj, iters=3, start="paraxial", final_corr=False):
    """-> hJ (B, R, 6) float64 holding values of precision dt_proj, res (B, R), valid (B, R)"""
    B, M = ids.shape
    R_IL, P_IL, _ = synth.camera_constants(prm)
    mids, mpos, mquat = synth.marker_table(prm)
    slot_of = {int(i): k for k, i in enumerate(mids)}
    slot = np.array([[slot_of.get(int(i), -1) for i in row] for row in ids])
    vism = slot >= 0
    sl = np.where(vism, slot, 0)
    g = dt_geom
    c = np.array([[0, 0, 0], [0, size, 0], [size, size, 0], [size, 0, 0.0]]).astype(g)
    Rm = synth.q2R(mquat.astype(g)).astype(g)                              # (nm,3,3)
    cw = mpos.astype(g)[sl][:, :, None, :] + np.einsum("bmij,kj->bmki", Rm[sl], c)     # (B,M,4,3)
    p = nom[:, 0:3].astype(g); R = rot.reshape(B, 3, 3).astype(g)
    RIL = R_IL.astype(g); PIL = P_IL.astype(g)
    u = cw - p[:, None, None, :]
    RP = np.einsum("bij,j->bi", R, PIL)
    d = u - RP[:, None, None, :]
    t = np.einsum("bji,bmkj->bmki", R, d)
    ru = np.einsum("bji,bmkj->bmki", R, u)
    X = np.einsum("ij,bmkj->bmki", RIL, t)
    Hpp = -np.einsum("ij,bkj->bik", RIL, R)                                # (B,3,3)
    skew = np.zeros(ru.shape[:-1] + (3, 3), g)
    skew[..., 0, 1] = -ru[..., 2]; skew[..., 0, 2] = ru[..., 1]; skew[..., 1, 0] = ru[..., 2]
    skew[..., 1, 2] = -ru[..., 0]; skew[..., 2, 0] = -ru[..., 1]; skew[..., 2, 1] = ru[..., 0]
    Hpt = np.einsum("ij,bmkjl->bmkil", RIL, skew)
    A = np.concatenate([np.broadcast_to(Hpp[:, None, None], Hpt.shape), Hpt], axis=-1)          # (B,M,4,3,6)
    pr = dt_proj
    vc = vis_consts(pr)
    F = np.diag([-1.0, -1, 1]).astype(pr)
    XL = (X.astype(pr)) * np.array([-1, -1, 1], pr)
    yl = left.reshape(B, M, 4, 2).astype(pr)
    cams = [(XL, F, yl)]
    if right is not None:
        XR = np.einsum("ij,bmkj->bmki", vc["R_RL_inv"], XL - vc["P_LR"])
        MR = vc["R_RL_inv"] @ F
        cams.append((XR, MR, right.reshape(B, M, 4, 2).astype(pr)))
    hs, rs, vs = [], [], []
    for Xc, Mx, y in cams:
        t0 = None
        if start == "meas":
            D = np.concatenate([y, np.ones(y.shape[:-1] + (1,), pr)], -1)
            zz = D @ vc["n"]
            latm = D - zz[..., None] * vc["n"]
            t0 = np.sqrt((latm * latm).sum(-1)) / zz
        uv, J, ok = project(vc, Xc, iters, t0, final_corr)
        jx = np.einsum("bmkqi,ij->bmkqj", J, Mx)
        h = np.einsum("bmkqi,bmkij->bmkqj", jx, A.astype(pr))
        hs.append(h); rs.append(y - uv); vs.append(np.broadcast_to((ok & vism[:, :, None])[..., None], uv.shape))
    h = np.concatenate(hs, axis=3).reshape(B, -1, 6).astype(np.float64)
    res = np.concatenate(rs, axis=3).reshape(B, -1).astype(np.float64)
    val = np.concatenate(vs, axis=3).reshape(B, -1)
    return h, res, val


def fold(h, res, val, w, dt):
    hh = np.where(val[..., None], h, 0).astype(dt); rr = np.where(val, res, 0).astype(dt)
    Lam = np.zeros((h.shape[0], 6, 6), dt); b = np.zeros((h.shape[0], 6), dt)
    wd = dt(w)
    for r in range(h.shape[1]):                                          # sequential accumulation, row by row
        wh = wd * hh[:, r, :]
        Lam += wh[:, :, None] * hh[:, r, None, :]
        b += wh * rr[:, r, None]
    return Lam, b


def ldl6(Lam):
    """unit lower L, d with Lam = L diag(d) L' (dtype of Lam)"""
    A = Lam.copy(); B = A.shape[0]
    L = np.tile(np.eye(6, dtype=A.dtype), (B, 1, 1)); d = np.zeros((B, 6), A.dtype)
    for a in range(6):
        d[:, a] = A[:, a, a]
        for i in range(a + 1, 6):
            L[:, i, a] = A[:, a, i] / d[:, a]
        for i in range(a + 1, 6):
            for j in range(i, 6):
                A[:, i, j] = A[:, i, j] - L[:, i, a] * A[:, a, j]
                A[:, j, i] = A[:, i, j]
    return L, d


def upd_seq(P, Lam, b, dt):
    """round 3: six sequential scalar updates (rows of L', information d, beta = L^-1 b), everything in dt"""
    P = P.astype(dt).copy(); Lam = Lam.astype(dt); b = b.astype(dt)
    L, d = ldl6(Lam)
    beta = np.linalg.solve(L.astype(np.float64), b.astype(np.float64)[..., None])[..., 0].astype(dt)   # forward substitution
    B, N = P.shape[0], P.shape[1]
    dx = np.zeros((B, N), dt)
    for a in range(6):
        h = np.zeros((B, N), dt)
        h[:, J6] = L[:, :, a]
        Ph = np.einsum("bij,bj->bi", P, h).astype(dt)
        hPh = (h * Ph).sum(1).astype(dt); hdx = (h * dx).sum(1).astype(dt)
        sp = dt(1) + d[:, a] * hPh
        g = (beta[:, a] - d[:, a] * hdx) / sp
        dk = d[:, a] / sp
        dx = dx + Ph * g[:, None]
        P = P - (Ph * dk[:, None])[:, :, None] * Ph[:, None, :]
    return dx.astype(np.float64), P.astype(np.float64)


def upd_forms(P32, Lam, b, dt_solve, dt_upd, form):
    """'oneshot': P - W W', W = P_J Z, Z Z' = (Lam^-1 + P_JJ)^-1 = Lam G;   'direct': J rows / columns as G P_J (no cancellation)
    6 x 6 algebra in dt_solve, N-sized products in dt_upd"""
    B, N = P32.shape[0], P32.shape[1]
    s = dt_solve
    PJJ = P32[:, J6][:, :, J6].astype(s); Lam = Lam.astype(s); b = b.astype(s)
    I6 = np.eye(6, dtype=s)
    A = I6 + np.einsum("bij,bjk->bik", PJJ, Lam)                          # I + P_JJ Lam
    G = np.linalg.inv(A.astype(np.float64)).astype(s) if s is np.float64 else inv_lu(A)       # (I + P_JJ Lam)^-1
    Sinv = np.einsum("bij,bjk->bik", Lam, G)                              # Lam G = (Lam^-1 + P_JJ)^-1, symmetric PSD
    Sinv = (Sinv + np.swapaxes(Sinv, 1, 2)) * s(0.5)
    m = np.einsum("bji,bj->bi", G, b)                                     # G' b
    u = dt_upd
    PJ = P32[:, :, J6].astype(u)                                          # (B,N,6) = P(:, J)
    dx = np.einsum("bnj,bj->bn", PJ, m.astype(u)).astype(u)
    P = P32.astype(u).copy()
    T = np.einsum("bnj,bjk->bnk", PJ, Sinv.astype(u)).astype(u)           # P_J Sinv
    P = P - np.einsum("bnk,bmk->bnm", T, PJ).astype(u)
    if form == "direct":
        GPJ = np.einsum("bij,bnj->bin", G.astype(u), PJ).astype(u)        # G P(J, :)  (6, N)
        P[:, J6, :] = GPJ
        P[:, :, J6] = np.swapaxes(GPJ, 1, 2)
        PJJn = GPJ[:, :, J6]
        P[:, J6[:, None], J6[None, :]] = (PJJn + np.swapaxes(PJJn, 1, 2)) * u(0.5)
    return dx.astype(np.float64), P.astype(np.float64)


def inv_lu(A):
    """Gauss-Jordan with partial pivoting in A's dtype (batched)"""
    B, n = A.shape[0], A.shape[1]
    M = np.concatenate([A.copy(), np.tile(np.eye(n, dtype=A.dtype), (B, 1, 1))], axis=2)
    ar = np.arange(B)
    for c in range(n):
        piv = c + np.abs(M[:, c:, c]).argmax(1)
        tmp = M[ar, c].copy(); M[ar, c] = M[ar, piv]; M[ar, piv] = tmp
        M[:, c] = M[:, c] / M[:, c, c][:, None]
        for r in range(n):
            if r != c:
                M[:, r] = M[:, r] - M[:, r, c][:, None] * M[:, c]
    return M[:, :, n:]


def exact_update(P, h, res, val, w):
    B, N = P.shape[0], P.shape[1]
    dx = np.zeros((B, N)); Pn = np.zeros_like(P)
    for bi in range(B):
        H = np.zeros((int(val[bi].sum()), N))
        H[:, J6] = h[bi][val[bi]]
        r = res[bi][val[bi]]
        S = H @ P[bi] @ H.T + np.eye(len(r)) / w
        K = np.linalg.solve(S, H @ P[bi]).T
        dx[bi] = K @ r
        Pn[bi] = (np.eye(N) - K @ H) @ P[bi]
        Pn[bi] = (Pn[bi] + Pn[bi].T) / 2
    return dx, Pn


def figures(dx, P, dx0, P0, nom):
    """literal / sigma-aware (on the increments: the nominal state is the same on both sides) / block-wise covariance"""
    lit = float((np.abs(dx - dx0).max(1) / 9.8).max())
    worst, where = 0.0, ""
    sig = np.sqrt(np.abs(np.einsum("bii->bi", P0)))
    for name, a, b_, floor in _BLOCKS:
        i0, i1 = _SIGMA_IDX[name]
        num = np.abs(dx[:, i0:i1] - dx0[:, i0:i1]).max(1) * (0.5 if name == "q" else 1.0)
        den = np.maximum(np.maximum(np.abs(nom[:, a:b_]).max(1), floor), sig[:, i0:i1].max(1))
        e = float((num / den).max())
        if e > worst:
            worst, where = e, name
    cb = cov_rel_err_blockwise(P, P0)
    dg = np.sqrt(np.abs(np.einsum("bii->bi", P)))
    ev = np.linalg.eigvalsh(P / (dg[:, :, None] * dg[:, None, :])).min()
    return f"literal {lit:.2e}  sigma {worst:.2e} ({where})  cov_block {cb:.2e}  min eig(corr) {ev:+.1e}"


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "room"
    stereo = "stereo" in sys.argv
    dialect = 0
    prm = capi.default_params(dialect)
    if which == "wall":
        size, M, B = 0.15, 16, 96
        prm.marker_size = size
        from test_pixels_gpu import _wall_map
        from replay_ref import OracleEngine
        probe = OracleEngine(1, 0, 18)
        _wall_map(prm, probe.orc.prm, size)
        nom, _, ids, left, right = pixel_scene(B, M, prm, size, seed=9, noise=5e-4, depth=(1.2, 1.8))
        rng = np.random.default_rng(10)
        nom[:, 0:3] += rng.normal(0, 0.003, (B, 3))
        nom = r32(nom); left = r32(left); right = r32(right)
        rot = r32(synth.q2R(nom[:, 6:10]).reshape(B, 9))
        P = np.broadcast_to(np.diag(np.repeat(np.array(list(prm.p0_diag)), 3)), (B, 18, 18)).copy()
        orc_prm = probe.orc.prm
    else:
        from test_pixels_gpu import _scene, SIZE
        size, M, B = SIZE, 4, 192
        prm, nom, rot, P, prev, ids, left, right = _scene(B, M, dialect, seed=11)
        orc_prm = None
    rgt = right if stereo else None
    w = 1.0 / prm.r_pix
    print(f"scene {which} B {B} M {M} stereo {stereo}: markers visible {float((ids >= 0).sum(1).mean()):.1f}")
    h0, res0, val = build_rows(nom, rot, ids, left, rgt, prm, size, np.float64, np.float64, iters=8)
    dx0, P0 = exact_update(P, h0, res0, val, w)
    # cross-check against the C oracle
    from replay_ref import OracleEngine
    eng = OracleEngine(B, dialect, 18)
    if orc_prm is not None:
        from test_pixels_gpu import _wall_map
        _wall_map(capi.default_params(0), eng.orc.prm, size)
    eng.set_state(nom, rot, P, np.zeros(B, np.int32))
    eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, rgt, size, prm.r_pix)
    print("  numpy exact vs C oracle: cov_block", f"{cov_rel_err_blockwise(P0, eng.get_state()[2]):.1e}",
          " dp", f"{np.abs(nom[:, :3] + dx0[:, :3] - eng.get_state()[0][:, :3]).max():.1e}")
    print(f"  rows per filter {val.sum(1).mean():.0f}; prior sigma p {np.sqrt(P[0, 0, 0]):.1e} -> posterior {np.sqrt(P0[:, 0, 0]).mean():.1e}")
    f32, f64 = np.float32, np.float64

    def run(label, geom, proj, fo, so, up, form, **kw):
        h, res, v = build_rows(nom, rot, ids, left, rgt, prm, size, geom, proj, **kw)
        v = v & val
        Lam, b = fold(h, res, v, w, fo)
        if form == "seq":
            dx, Pn = upd_seq(r32(P), Lam, b, up)
        else:
            dx, Pn = upd_forms(r32(P).astype(f32), Lam, b, so, up, form)
        print(f"  {label:58s} {figures(dx, Pn, dx0, P0, nom)}")

    run("all fp64, seq", f64, f64, f64, f64, f64, "seq", iters=8)
    run("all fp64, oneshot", f64, f64, f64, f64, f64, "oneshot", iters=8)
    run("all fp64, direct", f64, f64, f64, f64, f64, "direct", iters=8)
    print("  -- update form, everything in front exact")
    run("upd fp32 seq (round 3 form)", f64, f64, f64, f64, f32, "seq", iters=8)
    run("upd fp32 oneshot, solve fp64", f64, f64, f64, f64, f32, "oneshot", iters=8)
    run("upd fp32 direct,  solve fp64", f64, f64, f64, f64, f32, "direct", iters=8)
    run("upd fp32 direct,  solve fp32", f64, f64, f64, f32, f32, "direct", iters=8)
    print("  -- fold precision (update direct fp32, solve fp64)")
    run("fold fp32", f64, f64, f32, f64, f32, "direct", iters=8)
    print("  -- rows")
    run("proj fp32 (3 it paraxial), geom fp64, fold fp64", f64, f32, f64, f64, f32, "direct", iters=3)
    run("proj fp32 (2 it paraxial)", f64, f32, f64, f64, f32, "direct", iters=2)
    run("proj fp32 (1 it meas start + final corr)", f64, f32, f64, f64, f32, "direct", iters=1, start="meas", final_corr=True)
    run("proj fp32 (0 it meas start + final corr)", f64, f32, f64, f64, f32, "direct", iters=0, start="meas", final_corr=True)
    run("geom fp32, proj fp32 (1 it meas + corr), fold fp64", f32, f32, f64, f64, f32, "direct", iters=1, start="meas", final_corr=True)
    print("  -- candidates")
    run("ALL fp32 seq (round 3)", f32, f32, f32, f32, f32, "seq", iters=3)
    run("rows fp32, fold fp32, solve fp32, direct fp32", f32, f32, f32, f32, f32, "direct", iters=1, start="meas", final_corr=True)
    run("rows fp32, fold fp64, solve fp64, direct fp32", f32, f32, f64, f64, f32, "direct", iters=1, start="meas", final_corr=True)
    run("rows fp32, fold fp64, solve fp64, oneshot fp32", f32, f32, f64, f64, f32, "oneshot", iters=1, start="meas", final_corr=True)
    run("rows fp32, fold fp64, solve fp64, seq fp32", f32, f32, f64, f64, f32, "seq", iters=1, start="meas", final_corr=True)




def study_rows():
    """which part of the row construction matters: residual vs Jacobian precision"""
    from test_pixels_gpu import _scene, SIZE
    size, M, B = SIZE, 4, 192
    prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, seed=11)
    w = 1.0 / prm.r_pix
    f32, f64 = np.float32, np.float64
    h0, res0, val = build_rows(nom, rot, ids, left, None, prm, size, f64, f64, iters=8)
    dx0, P0 = exact_update(P, h0, res0, val, w)

    def go(label, h, res):
        Lam, b = fold(h, res, val, w, f64)
        dx, Pn = upd_forms(r32(P).astype(f32), Lam, b, f64, f32, "direct")
        print(f"  {label:58s} {figures(dx, Pn, dx0, P0, nom)}")
    go("exact rows", h0, res0)
    go("h rounded to fp32", r32(h0), res0)
    go("res rounded to fp32", h0, r32(res0))
    rng = np.random.default_rng(0)
    for e in (1e-9, 1e-8, 3e-8, 1e-7):
        go(f"res + uniform noise {e:.0e}", h0, res0 + rng.uniform(-e, e, res0.shape))
    for e in (1e-7, 1e-6, 1e-5):
        go(f"h (1 + {e:.0e} noise)", h0 * (1 + rng.uniform(-e, e, h0.shape)), res0)
    for it in (1, 2, 3, 4):
        h, res, v = build_rows(nom, rot, ids, left, None, prm, size, f64, f64, iters=it)
        go(f"fp64 rows, {it} Newton it from paraxial", h, res)
    for it in (0, 1, 2):
        h, res, v = build_rows(nom, rot, ids, left, None, prm, size, f64, f64, iters=it, start="meas", final_corr=True)
        go(f"fp64 rows, meas start, {it} it + final corr", h, res)
    h, res, v = build_rows(nom, rot, ids, left, None, prm, size, f32, f64, iters=8)
    go("geom fp32, proj fp64", h, res)
    h, res, v = build_rows(nom, rot, ids, left, None, prm, size, f64, f32, iters=8)
    go("geom fp64, proj fp32 converged", h, res)


if __name__ == "__main__":
    if "rows" in sys.argv:
        study_rows()
    else:
        main()
