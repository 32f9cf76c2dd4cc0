"""Seeded synthetic IMU + marker streams (host logic; numpy only).

Counter-based: every array is generated per block of 1024 filters from
Philox(key = [seed + stream id + step, block index]), so any shard [lo, hi) of the
batch reproduces exactly the values of the unsharded run.
Distributions follow the land recording of the reference
(matlab/dataset/landdata/dataset-02): accel noise std (0.66, 0.18, 0.53) m/s^2, gyro
std (0.022, 0.014, 0.008) rad/s, gravity [9.8, 0, 0] (InitPositionAndQuaternion.m:79).
"""
import numpy as np

BASE_SEED = 0xFB05EC0F
RNG_BLOCK = 1024
_S_STATE, _S_IMU, _S_MARK = 1 << 40, 2 << 40, 3 << 40

ACC_STD = np.array([0.66, 0.18, 0.53])
GYR_STD = np.array([0.022, 0.014, 0.008])
GRAVITY = np.array([9.8, 0.0, 0.0])


def _blocks(lo, hi):
    b0, b1 = lo // RNG_BLOCK, (hi - 1) // RNG_BLOCK
    for blk in range(b0, b1 + 1):
        s = max(lo, blk * RNG_BLOCK) - blk * RNG_BLOCK
        e = min(hi, (blk + 1) * RNG_BLOCK) - blk * RNG_BLOCK
        yield blk, s, e


def _rng(stream, step, blk, seed):
    return np.random.Generator(np.random.Philox(key=[(seed + stream + step) & (2**64 - 1), blk]))


def _draw(lo, hi, stream, step, seed, fn):
    parts = [fn(_rng(stream, step, blk, seed))[s:e] for blk, s, e in _blocks(lo, hi)]
    return np.concatenate(parts, axis=0)


# ---- small quaternion helpers for building consistent measurements (wxyz) -----------
def qmul(p, q):
    pw, px, py, pz = np.moveaxis(p, -1, 0)
    qw, qx, qy, qz = np.moveaxis(q, -1, 0)
    return np.stack([pw * qw - px * qx - py * qy - pz * qz,
                     pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx,
                     pw * qz + px * qy - py * qx + pz * qw], axis=-1)


def q2R(q):
    w, x, y, z = np.moveaxis(q, -1, 0)
    R = np.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], axis=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def R2q(R):
    """trace-based conversion of one 3x3 rotation matrix."""
    R = np.asarray(R, float)
    t = np.trace(R)
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1:] = [(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def camera_constants(params):
    """R_IL, P_IL, Q_IL from the raw left T_SC of an FbusParams (flip as FBUS_EKF.m:68)."""
    T = np.array(list(params.T_SC_left), float).reshape(4, 4)
    T = np.diag([-1.0, -1, 1, 1]) @ T
    R_IL = T[:3, :3]
    return R_IL, -R_IL.T @ T[:3, 3], R2q(R_IL)


def marker_table(params):
    n = params.n_markers
    ids = np.array(list(params.marker_id)[:n], np.int32)
    pos = np.array([list(params.marker_pos[k]) for k in range(n)], float)
    quat = np.array([R2q(np.array(list(params.marker_rot[k])).reshape(3, 3)) for k in range(n)], float)
    return ids, pos, quat


# ---- streams ----------------------------------------------------------------------------
def initial_state(lo, hi, p0_diag, nstate=18, seed=BASE_SEED, mixed_cov=False, with_cov=True):
    """nominal (n,19), rot (n,9), P (n,N,N), prev_id (n,) for filters [lo, hi).  with_cov=False returns P = None
    (large batches: the caller lets the device write the P0 diagonal instead of shipping n x N x N doubles)."""
    N = nstate
    mixed_cov = mixed_cov and with_cov

    def gen(r):
        p = r.uniform(-1, 1, (RNG_BLOCK, 3))
        v = r.normal(0, 0.1, (RNG_BLOCK, 3))
        q = r.normal(size=(RNG_BLOCK, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        ba = r.normal(0, 0.05, (RNG_BLOCK, 3))
        bg = r.normal(0, 0.002, (RNG_BLOCK, 3))
        g = np.broadcast_to(GRAVITY, (RNG_BLOCK, 3))
        if not mixed_cov:                              # A is drawn last: skipping it leaves p .. bg unchanged
            return np.concatenate([p, v, q, ba, bg, g], axis=1)
        A = r.normal(0, 0.05, (RNG_BLOCK, 18, 18))
        return np.concatenate([p, v, q, ba, bg, g, A.reshape(RNG_BLOCK, -1)], axis=1)

    raw = _draw(lo, hi, _S_STATE, 0, seed, gen)
    nominal = np.ascontiguousarray(raw[:, :19])
    rot = q2R(nominal[:, 6:10]).reshape(-1, 9)
    P0 = np.diag(np.repeat(np.asarray(p0_diag, float), 3)[:N])
    n = hi - lo
    if not with_cov:
        return nominal, rot, None, np.zeros(n, np.int32)
    if mixed_cov:   # S C S with S = sqrt(P0) and C = A A' a random correlation-like matrix (A = I + small):
        # dense, symmetric positive definite, every block at its physical scale
        A = np.eye(N) + raw[:, 19:].reshape(n, 18, 18)[:, :N, :N]
        S = np.sqrt(P0)
        P = S @ (A @ np.swapaxes(A, 1, 2)) @ S
        P = (P + np.swapaxes(P, 1, 2)) / 2
    else:
        P = np.broadcast_to(P0, (n, N, N)).copy()
    return nominal, rot, P, np.zeros(n, np.int32)


def imu_samples(lo, hi, step, K, nominal0, seed=BASE_SEED):
    """accel, gyro (K, n, 3) for IMU steps [step, step+K) of filters [lo, hi):
    specific force of a body at rest in the initial attitude plus recorded noise levels."""
    R0 = q2R(nominal0[:, 6:10])
    f0 = np.einsum("nji,j->ni", R0, -GRAVITY)           # R' (-g)
    acc, gyr = [], []
    for k in range(K):
        z = _draw(lo, hi, _S_IMU, step + k, seed, lambda r: r.normal(size=(RNG_BLOCK, 6)))
        acc.append(f0 + z[:, :3] * ACC_STD)
        gyr.append(z[:, 3:] * GYR_STD)
    return np.stack(acc), np.stack(gyr)


def marker_frame(lo, hi, frame, M, nominal0, params, seed=BASE_SEED, noise=1e-3):
    """ids (n,M) int32, pos (n,M,3), quat (n,M,4): M distinct map markers per filter,
    measurement = h(x0) + N(0, noise) with x0 the initial nominal state."""
    R_IL, P_IL, Q_IL = camera_constants(params)
    mids, mpos, mquat = marker_table(params)
    nm = len(mids)

    def gen(r):
        order = np.argsort(r.random((RNG_BLOCK, nm)), axis=1)[:, :M]
        z = r.normal(size=(RNG_BLOCK, M, 7))
        return np.concatenate([order[..., None].astype(float), z], axis=2)

    raw = _draw(lo, hi, _S_MARK, frame, seed, gen)
    slot = raw[..., 0].astype(np.int64)
    z = raw[..., 1:]
    p0, q0 = nominal0[:, 0:3], nominal0[:, 6:10]
    R0 = q2R(q0)
    Pm, Qm = mpos[slot], mquat[slot]                               # (n,M,3), (n,M,4)
    d = Pm - p0[:, None, :] - np.einsum("nij,j->ni", R0, P_IL)[:, None, :]
    hp = np.einsum("ij,nmj->nmi", R_IL, np.einsum("nji,nmj->nmi", R0, d))
    qc = q0 * np.array([1.0, -1, -1, -1])
    hq = qmul(qmul(np.broadcast_to(Q_IL, q0.shape), qc)[:, None, :], Qm)
    pos = hp + noise * z[..., :3]
    quat = hq + noise * z[..., 3:]
    quat /= np.linalg.norm(quat, axis=-1, keepdims=True)
    return mids[slot].astype(np.int32), pos, quat


def port_project(params, Xcam, right=False, iters=40):
    """Forward flat-port projection in numpy (the oracle's fbv_project_camera, vectorised; used to build realistic image
    measurements for the benches -- parity tests use the oracle itself): points (..., 3) of the LEFT camera frame as the
    triangulation returns them (vision.cpp:597-599: x and y flipped) -> normalised image points (..., 2) of the left or right
    camera and the visibility mask (in front of the port, inside 0.9 of its field of view).
    rho = d_air t + d_glass tan(theta_glass) + z_w tan(theta_water), t = tan(theta_air), Snell twice; Newton from the
    paraxial start (monotone: L is increasing and concave)."""
    X = np.asarray(Xcam, float) * np.array([-1.0, -1.0, 1.0])
    if right:
        TL = np.array(list(params.T_SC_left), float).reshape(4, 4)
        TR = np.array(list(params.T_SC_right), float).reshape(4, 4)
        R_RL = TL[:3, :3] @ TR[:3, :3].T
        P_LR = TL[:3, 3] - R_RL @ TR[:3, 3]
        X = np.einsum("ij,...j->...i", np.linalg.inv(R_RL), X - P_LR)
    n = np.array(list(params.port_normal), float)
    a0, a1 = params.n_air / params.n_glass, params.n_air / params.n_water
    z = X @ n
    lat = X - z[..., None] * n
    rho = np.sqrt((lat * lat).sum(-1))
    zw = z - params.d_air - params.d_glass
    ok = (zw > 0) & (rho < 0.9 * np.where(zw > 0, zw, 0) * a1 / np.sqrt(1 - a1 * a1))
    zs = np.where(ok, zw, 1.0)
    rs = np.where(ok, rho, 0.0)
    t = rs / (params.d_air + a0 * params.d_glass + a1 * zs)
    for _ in range(iters):
        r = 1.0 / np.sqrt(1.0 + t * t)
        s = t * r
        icg, icw = 1.0 / np.sqrt(1.0 - a0 * a0 * s * s), 1.0 / np.sqrt(1.0 - a1 * a1 * s * s)
        g, w = params.d_glass * a0 * icg, zs * a1 * icw
        L = params.d_air * t + s * (g + w)
        Lt = params.d_air + (g * icg * icg + w * icw * icw) * r ** 3
        t = np.maximum(t + (rs - L) / Lt, 0.0)
    k = np.where(rs > 0, t / np.where(rs > 0, rs, 1.0), 0.0)
    D = n + k[..., None] * np.where(ok[..., None], lat, 0.0)
    return D[..., :2] / D[..., 2:3], ok


def pixel_wall_scene(B, slots, params, size, seed=9, nbase=256, depth=(1.2, 1.8), noise=5e-4, stereo=False):
    """Scene for the reprojection-row update (fbus_ekf_correct_pixels) at batch scale, without the oracle: REPLACES the marker map
    in `params` by a 4 x 4 wall of 16 markers (ids 0..15, 0.3 m pitch, the orientation of the reference's marker 0:
    GetMarkerMap.m) and places `nbase` camera poses 1.2-1.8 m in front of it (repeated to B filters, positions jittered by 3 mm).
    Returns (nominal (B,19), rot (B,9), ids (B,slots) with -1 padding, left (B,slots,8)[, right (B,slots,8) with stereo=True]): the
    markers in front of the port and the FLAT-PORT projections of their corners (port_project) + noise as the measured image points
    (round 4: the kernel starts its Newton iteration from the measured point, so the timing legs need innovations of realistic
    size; the nominal positions are then jittered by 3 mm).  Parity tests use the oracle's projection (tests/util.py::pixel_scene)."""
    rng = np.random.default_rng(seed)
    _, mpos, mquat = marker_table(params)
    R0 = np.array(list(params.marker_rot[0])).reshape(3, 3)
    q0 = mquat[0].copy()
    params.n_markers = 16
    wall = np.zeros((16, 3))
    for k in range(16):
        wall[k] = mpos[0] + R0 @ np.array([0.3 * (k % 4 - 1.5), 0.3 * (k // 4 - 1.5), 0.0])
        params.marker_id[k] = k
        for i in range(3):
            params.marker_pos[k][i] = float(wall[k][i])
        for i in range(9):
            params.marker_rot[k][i] = float(R0.ravel()[i])
    R_IL, P_IL, Q_IL = camera_constants(params)
    c = np.array([[0, 0, 0], [0, size, 0], [size, size, 0], [size, 0, 0.0]])
    world = wall[:, None, :] + (R0 @ c.T).T[None]                     # (16, 4, 3) corners in the world
    nom = np.zeros((nbase, 19))
    nom[:, 16] = 9.8
    ids = np.full((nbase, slots), -1, np.int32)
    left = np.zeros((nbase, slots, 8))
    right = np.zeros((nbase, slots, 8))
    for b in range(nbase):
        while True:
            k0 = int(rng.integers(16))
            yq = np.array([0.0, 1.0, 0.0, 0.0]) + rng.normal(0, 0.15, 4)
            yq /= np.linalg.norm(yq)
            yp = np.array([rng.normal(0, 0.08), rng.normal(0, 0.08), rng.uniform(*depth)])
            q = qmul(qmul(q0, yq * np.array([1.0, -1, -1, -1])), Q_IL)     # InitPositionAndQuaternion.m:52-72
            q /= np.linalg.norm(q)
            R = q2R(q)
            p = -R @ R_IL.T @ yp + wall[k0] - R @ P_IL
            cam = np.einsum("ij,kcj->kci", R_IL @ R.T, world - p - R @ P_IL)        # (16, 4, 3)
            vis = (cam[:, :, 2].min(axis=1) > 0.25) & ((np.linalg.norm(cam[:, :, :2], axis=2) / cam[:, :, 2]).max(axis=1) < 0.8)
            uvL, okL = port_project(params, cam)
            vis &= okL.all(axis=1)
            if stereo:
                uvR, okR = port_project(params, cam, right=True)
                vis &= okR.all(axis=1)
            if vis[k0]:
                break
        nom[b, 0:3], nom[b, 6:10] = p, q
        order = [k0] + [k for k in rng.permutation(16) if k != k0 and vis[k]]
        for m, k in enumerate(order[:slots]):
            ids[b, m] = k
            left[b, m] = uvL[k].ravel() + rng.normal(0, noise, 8)
            if stereo:
                right[b, m] = uvR[k].ravel() + rng.normal(0, noise, 8)
    rep = (B + nbase - 1) // nbase
    nom = np.tile(nom, (rep, 1))[:B]
    ids = np.tile(ids, (rep, 1))[:B]
    left = np.tile(left, (rep, 1, 1))[:B]
    right = np.tile(right, (rep, 1, 1))[:B]
    nom[:, 0:3] += rng.normal(0, 0.003, (B, 3))
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
    nom = r32(nom)
    rot = r32(q2R(nom[:, 6:10]).reshape(B, 9))
    if stereo:
        return nom, rot, ids, r32(left), r32(right)
    return nom, rot, ids, r32(left)
