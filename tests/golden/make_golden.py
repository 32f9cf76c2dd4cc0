#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ (run in the dev container only).

  vision_water.npz / vision_land.npz
      slices of the reference's OWN recorded data (matlab/dataset/*/corners.txt ->
      image.txt; GPL-3.0, (c) the FBUS-EKF authors): inputs and the outputs the
      reference's C++ vision chain logged.  These pin oracle/vision_oracle.c.
  land_slice.npz
      the first seconds of landdata/dataset-02 (imu.txt / image.txt rows) plus the
      per-frame filter states produced by the numpy twin (oracle/ekf_oracle_np.py)
      driven with the loop of matlab/FBUS_EKF.m:118-210.
  ekf_random.npz
      seeded random single-step vectors (state in, inputs, state out) from the
      numpy twin for both dialects, N in {15, 18}, both correct() modes.

  recordings.npz
      the reference's two COMPLETE recordings (matlab/dataset/landdata/dataset-02 and
      waterdata/dataset-06: imu.txt, image.txt; GPL-3.0, (c) the FBUS-EKF authors) -- inputs only,
      float64, compressed -- for the full-length replay tests (the oracle runs beside
      the device in the test; no expected outputs are stored).  fusion.txt's gyro-bias
      columns and its pose columns ride along: the recorded C++ output comes from an older
      revision with its own world frame, so only frame-independent
      quantities can be compared -- the initial gyro bias and the RELATIVE motion of the IMU.

The EKF vectors are NOT reference outputs (the reference cannot run here): they pin
the two restatements and the HIP kernels to each other ("parity unpinned").
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ekf_oracle_np as onp  # noqa: E402

REF = "/root/reference/matlab/dataset"


def vision():
    cw = np.loadtxt(f"{REF}/waterdata/dataset-06/corners.txt")
    iw = np.loadtxt(f"{REF}/waterdata/dataset-06/image.txt")
    idx = np.r_[0:32, 500:516, len(cw) - 16:len(cw)]
    np.savez_compressed(os.path.join(HERE, "vision_water.npz"), corners=cw[idx], image=iw[idx])
    cl = np.loadtxt(f"{REF}/landdata/dataset-02/corners.txt")
    il = np.loadtxt(f"{REF}/landdata/dataset-02/image.txt")
    idx = np.r_[0:32, 600:616, len(cl) - 16:len(cl)]
    np.savez_compressed(os.path.join(HERE, "vision_land.npz"), corners=cl[idx], image=il[idx])


def state_vec(s):
    return np.concatenate([s.p, s.v, s.q, s.ba, s.bg, s.g])


def replay(imu, img, dialect, nframes):
    """matlab/FBUS_EKF.m:118-210 driven with the numpy twin."""
    prm = onp.Params(dialect, 18)
    s = onp.State(18)
    s.P = prm.P0()
    mean = imu[:500].mean(axis=0)                                  # InitGravityAndGyrobias.m:36-40
    s.g = -np.array([0, 0, np.linalg.norm(mean[1:4])])
    s.bg = mean[4:7].copy()
    # InitPositionAndQuaternion.m:38-80 (single visible marker)
    mid, yp, yq = int(img[0, 1]), img[0, 2:5], img[0, 5:9]
    Pm, Qm = prm.markers[mid]
    Q_IG = onp.qmul(onp.qmul(Qm, yq * np.array([1, -1, -1, -1.0])), prm.Q_IL)
    R_IG = onp.q2R(Q_IG) if dialect == onp.MATLAB else onp.q2R_eigen(Q_IG)   # filter.cpp:383 toRotationMatrix
    s.q, s.R = Q_IG, R_IG
    s.p = -R_IG @ prm.R_IL.T @ yp + Pm - R_IG @ prm.P_IL
    s.g = np.array([9.8, 0, 0])
    idx = int(np.argmax(imu[:, 0] > img[0, 0]))                    # FBUS_EKF.m:124-131 (0-based)
    pre_img, n_img = 0.0, 0
    out, npred = [], []
    while n_img < len(img) - 1 and len(out) < nframes:
        j = n_img + 1
        while j < len(img) and img[j, 0] == img[n_img, 0]:
            j += 1
        cur = img[n_img, 0]
        meas = img[n_img:j]
        n_img = j
        cnt = 0
        if cur - pre_img > 0.1 and pre_img != 0:
            raise RuntimeError("reset inside the golden slice is not expected")
        pre_imu = imu[idx - 1, 0]
        k = idx
        while k < len(imu):
            if imu[k, 0] > cur:
                break
            if imu[k, 0] < pre_img:
                pre_imu = imu[k, 0]
                k += 1
                continue
            dt = imu[k, 0] - pre_imu
            pre_imu = imu[k, 0]
            onp.predict(s, prm, imu[k, 1:4], imu[k, 4:7], dt)
            cnt += 1
            k += 1
        idx = k
        pre_img = cur
        onp.correct(s, prm, meas[:, 1].astype(int), meas[:, 2:5], meas[:, 5:9], onp.NEAREST)
        out.append(np.concatenate([[cur], state_vec(s), s.R.ravel(), s.P.ravel()]))
        npred.append(cnt)
    return np.array(out), np.array(npred)


def land_slice():
    imu = np.loadtxt(f"{REF}/landdata/dataset-02/imu.txt")
    img = np.loadtxt(f"{REF}/landdata/dataset-02/image.txt")
    nframes = 50
    t_end = img[nframes + 1, 0]
    imu_s = imu[imu[:, 0] <= t_end + 0.01]
    img_s = img[:nframes + 2]
    gm, nm = replay(imu_s, img_s, onp.MATLAB, nframes)
    gc, nc = replay(imu_s, img_s, onp.CPP, nframes)
    assert (nm == nc).all()
    np.savez_compressed(os.path.join(HERE, "land_slice.npz"), imu=imu_s, image=img_s,
                        states_matlab=gm, states_cpp=gc, npredict=nm)


def ekf_random():
    sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
    from fbus_ekf import capi, synth
    rng = np.random.default_rng(20261002)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)   # fp32-representable inputs
    out = {}
    B, M = 24, 4
    for dialect in (onp.MATLAB, onp.CPP):
        cprm = capi.default_params(dialect)
        for n in (18, 15):
            prm = onp.Params(dialect, n)
            tag = f"d{dialect}_n{n}"
            nom, rot, P, _ = synth.initial_state(5000, 5000 + B, list(cprm.p0_diag), n, mixed_cov=True)
            nom, rot, P = r32(nom), r32(rot), r32(P)
            acc, gyr = synth.imu_samples(5000, 5000 + B, 0, 1, nom)
            acc, gyr = r32(acc[0]), r32(gyr[0])
            gyr[0] = nom[0, 13:16]                     # w == 0 exactly: the reference's NaN case (guarded)
            gyr[1] = r32(nom[1, 13:16] + [3e-5, 0, 0])  # below the C++ 1e-4 small-rate switch
            dt = r32(rng.uniform(0.001, 0.01, B))
            ids, pos, quat = synth.marker_frame(5000, 5000 + B, 0, M, nom, cprm)
            pos = r32(pos + rng.normal(0, 0.03, pos.shape))        # innovations of a few cm
            quat = quat + rng.normal(0, 0.01, quat.shape)
            quat = r32(quat / np.linalg.norm(quat, axis=2, keepdims=True))
            quat[5] = -quat[5]                         # opposite-sign measurement quaternion: sign unification
            ids[2] = [-1, -1, -1, -1]                  # nothing visible
            ids[3] = [9, -1, 9, -1]                    # only an id outside the map
            ids[4, 1] = -1
            ids[6, 0] = 9
            prev = rng.choice([0, 1, 2, 16], B).astype(np.int32)

            def run(fn):
                o_nom, o_rot, o_P, o_prev, o_ok = nom.copy(), rot.copy(), P.copy(), prev.copy(), np.zeros(B, np.int32)
                for b in range(B):
                    s = onp.State(n)
                    s.p, s.v, s.q = nom[b, 0:3].copy(), nom[b, 3:6].copy(), nom[b, 6:10].copy()
                    s.ba, s.bg, s.g = nom[b, 10:13].copy(), nom[b, 13:16].copy(), nom[b, 16:19].copy()
                    s.R, s.P, s.prev_id = rot[b].reshape(3, 3).copy(), P[b].copy(), int(prev[b])
                    o_ok[b] = fn(s, b)
                    o_nom[b], o_rot[b], o_P[b], o_prev[b] = state_vec(s), s.R.ravel(), s.P, s.prev_id
                return o_nom, o_rot, o_P, o_prev, o_ok

            with np.errstate(all="ignore"):
                pn, pr, pP, _, _ = run(lambda s, b: (onp.predict(s, prm, acc[b], gyr[b], dt[b]), 1)[1])
            cA = run(lambda s, b: int(onp.correct(s, prm, ids[b], pos[b], quat[b], onp.NEAREST)))
            cS = run(lambda s, b: int(onp.correct(s, prm, ids[b], pos[b], quat[b], onp.STACKED)))
            out.update({f"{tag}_nom": nom, f"{tag}_rot": rot, f"{tag}_P": P, f"{tag}_prev": prev,
                        f"{tag}_acc": acc, f"{tag}_gyr": gyr, f"{tag}_dt": dt,
                        f"{tag}_ids": ids.astype(np.int32), f"{tag}_pos": pos, f"{tag}_quat": quat,
                        f"{tag}_pred_nom": pn, f"{tag}_pred_rot": pr, f"{tag}_pred_P": pP})
            for name, c in (("near", cA), ("stack", cS)):
                out.update({f"{tag}_{name}_nom": c[0], f"{tag}_{name}_P": c[2], f"{tag}_{name}_prev": c[3],
                            f"{tag}_{name}_ok": c[4]})
    np.savez_compressed(os.path.join(HERE, "ekf_random.npz"), **out)


def recordings():
    out = {}
    for tag, d in (("land", "landdata/dataset-02"), ("water", "waterdata/dataset-06")):
        out[f"{tag}_imu"] = np.loadtxt(f"{REF}/{d}/imu.txt")
        out[f"{tag}_image"] = np.loadtxt(f"{REF}/{d}/image.txt")
        fus = np.loadtxt(f"{REF}/{d}/fusion.txt")
        out[f"{tag}_fusion_bg"] = fus[:, [0, 14, 15, 16]]
        out[f"{tag}_fusion_pose"] = fus[:, 0:8]          # t, p(3), q(wxyz): the recorded fused trajectory (older revision)
    # the water recording's corners.txt in full (`t id` + 8 left + 8 right undistorted normalised corner coordinates, vision.cpp:111-119):
    # what the cameras saw, row for row with image.txt -- the input of the replay through the north star's reprojection rows
    out["water_corners"] = np.loadtxt(f"{REF}/waterdata/dataset-06/corners.txt")
    np.savez_compressed(os.path.join(HERE, "recordings.npz"), **out)


if __name__ == "__main__":
    recordings()
    vision()
    land_slice()
    ekf_random()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
