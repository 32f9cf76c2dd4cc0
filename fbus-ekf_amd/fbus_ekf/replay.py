"""Dataset replay driver: the frame loop of matlab/FBUS_EKF.m:118-210 around predict/correct.

Host-side sequencing only (which IMU rows belong to which camera frame, when to
reset); the filter arithmetic is delegated to an `engine` exposing the
BatchedFilter interface (set_state / get_state / predict / correct) with B = 1.
File formats: SURVEY.md App. C (imu.txt `t ax ay az gx gy gz`, image.txt
`t id px py pz qw qx qy qz`).

The initialisation (InitGravityAndGyrobias.m:36-40, InitPositionAndQuaternion.m:38-80)
and the reset after a vision gap (ResetState.m:37-80) are engine calls too
(`init_gravity_bias`, `pose_init`): HIP kernels for BatchedFilter.  The small numpy
helpers below restate the same formulas for the tests.
"""
import numpy as np

from . import synth
from .capi import MAX_VISIBLE

MAX_KCOUNT = 255      # IMU samples a window entry can name (fbus_ekf_frames_fused_dev: kcount is 0..255)


def init_gravity_gyrobias(imu_rows):
    """InitGravityAndGyrobias.m:36-40 : mean of the first IMU rows."""
    mean = np.asarray(imu_rows, float).mean(axis=0)
    return -np.array([0.0, 0.0, np.linalg.norm(mean[1:4])]), mean[4:7].copy()


def nearest(meas):
    """index of the nearest marker, start threshold 10 (MeasureUpdate.m:51-60)."""
    best, bi = 10.0, -1
    for i, row in enumerate(meas):
        d = np.linalg.norm(row[1:4])
        if d < best:
            best, bi = d, i
    return bi


def pose_from_marker(meas_row, params):
    """InitPositionAndQuaternion.m:52-72 / ResetState.m:52-72 : IMU pose from one marker."""
    R_IL, P_IL, Q_IL = synth.camera_constants(params)
    ids, mpos, mquat = synth.marker_table(params)
    k = int(np.nonzero(ids == int(meas_row[0]))[0][0])
    yp, yq = meas_row[1:4], meas_row[4:8]
    q = synth.qmul(synth.qmul(mquat[k], yq * np.array([1.0, -1, -1, -1])), Q_IL)
    R = synth.q2R(q)
    p = -R @ R_IL.T @ yp + mpos[k] - R @ P_IL
    return p, q, R


def replay(engine, imu, image, params, max_frames=None, matlab_reset=True, corners=None, stereo=True):
    """Runs the recording through `engine` (B = 1).  Returns (states, npredict):
    states[k] = [t, nominal(19), rot(9), P(N*N)] after frame k.
    corners (optional; round 5): the rows of corners.txt that belong to the rows of image.txt (`t id` + 8 left + 8 right undistorted
    normalised corner coordinates, vision.cpp:111-119).  The frame loop, the initialisation and the resets stay as they are (they use
    the marker poses of image.txt, as the reference does), but every MeasureUpdate becomes the north star's: correct() from the corner
    PIXELS through the flat-port model (engine.correct_pixels; stereo=False: the left camera's rows only)."""
    imu = np.asarray(imu, float)
    image = np.asarray(image, float)
    if corners is not None:
        corners = np.asarray(corners, float)
        if len(corners) != len(image) or np.abs(corners[:, 0] - image[:, 0]).max() > 1e-9:
            raise ValueError("corners rows must be the rows of image (same time stamps)")
    N = engine.N
    P0 = np.diag(np.repeat(np.array(list(params.p0_diag)), 3)[:N])[None]
    engine.set_state(np.zeros((1, 19)), np.zeros((1, 9)), P0, np.zeros(1, np.int32))
    # FBUS_EKF.m:118 (gravity, gyro bias from the first 500 IMU rows) and :124-132 (pose from the first frame):
    # both run inside the engine (device kernels for BatchedFilter)
    engine.init_gravity_bias(imu[:500, None, 1:4], imu[:500, None, 4:7])
    meas0 = image[0:1, 1:9]
    engine.pose_init(meas0[:, 0].astype(np.int32)[None], meas0[None, :, 1:4], meas0[None, :, 4:8], 0)
    idx = int(np.argmax(imu[:, 0] > image[0, 0]))
    pre_img, n_img = 0.0, 0
    out, npred = [], []
    while n_img < len(image) - 1 and (max_frames is None or len(out) < max_frames):
        j = n_img + 1                                               # FBUS_EKF.m:155-164
        while j < len(image) and image[j, 0] == image[n_img, 0]:
            j += 1
        cur = image[n_img, 0]
        meas = image[n_img:j, 1:9]
        crn = corners[n_img:j] if corners is not None else None
        n_img = j
        cnt = 0
        if cur - pre_img > 0.1 and pre_img != 0 and matlab_reset:   # FBUS_EKF.m:168-171, ResetState.m:75-79
            engine.pose_init(meas[:, 0].astype(np.int32)[None], meas[None, :, 1:4], meas[None, :, 4:8], 1)
            pre_img = cur
        else:
            pre_imu = imu[idx - 1, 0]                               # FBUS_EKF.m:175-191
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
                engine.predict(imu[k:k + 1, 1:4], imu[k:k + 1, 4:7], np.array([dt]))
                cnt += 1
                k += 1
            idx = k
            pre_img = cur
            if crn is None:
                engine.correct(meas[:, 0].astype(np.int32)[None], meas[None, :, 1:4], meas[None, :, 4:8], 0)
            else:
                engine.correct_pixels(crn[:, 1].astype(np.int32)[None], crn[None, :, 2:10], crn[None, :, 10:18] if stereo else None)
        nominal, rot, P, _ = engine.get_state()
        out.append(np.concatenate([[cur], nominal.ravel(), rot.ravel(), P.ravel()]).astype(np.float64))
        npred.append(cnt)
    return np.array(out), np.array(npred)


def plan_windows(imu, image, max_frames=None, matlab_reset=True, max_window=64):
    """The frame loop of FBUS_EKF.m:151-210 as a PLAN: which IMU samples (with their dt) go in front of which frame, where the
    script would reset instead (a vision gap > 0.1 s, FBUS_EKF.m:168-171), cut into windows of consecutive non-reset frames
    of at most `max_window` frames.  Everything here depends on the time stamps only, so a whole recording can be planned
    before the first launch.  Returns a list of ("reset", meas), ("window", kcount, imu_rows, dts, [meas per frame]) and --
    only when a frame has more than 255 IMU samples in front of it -- ("predict", imu_rows, dts) for the leading samples
    that do not fit a window entry.  Raises ValueError for a frame with more than MAX_VISIBLE markers."""
    imu = np.asarray(imu, float)
    image = np.asarray(image, float)
    idx = int(np.argmax(imu[:, 0] > image[0, 0]))
    pre_img, n_img, nframes = 0.0, 0, 0
    plan, cur_win = [], None

    def flush():
        nonlocal cur_win
        if cur_win is not None and cur_win[1]:
            plan.append(("window", np.array(cur_win[1], np.int32), np.array(cur_win[2], int), np.array(cur_win[3], float), cur_win[4]))
        cur_win = None

    while n_img < len(image) - 1 and (max_frames is None or nframes < max_frames):
        j = n_img + 1
        while j < len(image) and image[j, 0] == image[n_img, 0]:
            j += 1
        cur = image[n_img, 0]
        meas = image[n_img:j, 1:9]
        n_img = j
        nframes += 1
        if cur - pre_img > 0.1 and pre_img != 0 and matlab_reset:
            flush()
            plan.append(("reset", meas))
            pre_img = cur
            continue
        rows, dts = [], []
        pre_imu = imu[idx - 1, 0]
        k = idx
        while k < len(imu):
            if imu[k, 0] > cur:
                break
            if imu[k, 0] < pre_img:
                pre_imu = imu[k, 0]
                k += 1
                continue
            dts.append(imu[k, 0] - pre_imu)
            pre_imu = imu[k, 0]
            rows.append(k)
            k += 1
        idx = k
        pre_img = cur
        if len(meas) > MAX_VISIBLE:
            raise ValueError(f"frame at t = {cur}: {len(meas)} markers, at most {MAX_VISIBLE} per frame")
        if len(rows) > MAX_KCOUNT:
            # more IMU samples in front of this frame than a window entry can name (kcount is 0..255; a 1 kHz IMU after a
            # long vision gap with matlab_reset = False): the leading samples run as plain predicts in front of the window
            flush()
            cut = len(rows) - MAX_KCOUNT
            plan.append(("predict", np.array(rows[:cut], int), np.array(dts[:cut], float)))
            rows, dts = rows[cut:], dts[cut:]
        if cur_win is None:
            cur_win = ["window", [], [], [], []]
        if len(cur_win[1]) == max_window:
            flush()
            cur_win = ["window", [], [], [], []]
        cur_win[1].append(len(rows)); cur_win[2] += rows; cur_win[3] += dts; cur_win[4].append(meas)
    flush()
    return plan


def replay_windowed(flt, imu, image, params, max_frames=None, max_window=64):
    """The recording through the Matlab loop on `flt` (a BatchedFilter of B filters, every one of them fed the same recording:
    config 1 at batch scale) with each stretch of consecutive frames as ONE launch of the frame-window kernel
    (fbus_ekf_frames_fused_dev) instead of one launch per EKF step.  Returns the number of EKF steps per filter; the state is
    read with flt.get_state().  Same arithmetic as replay(); the results agree to fp32 rounding (resident vs streamed predict)."""
    import torch
    imu = np.asarray(imu, float)
    image = np.asarray(image, float)
    B, N = flt.B, flt.N
    dev = torch.device("cuda", flt.device) if hasattr(flt, "device") else torch.device("cuda:0")
    tt = torch.float32 if flt.np_dtype == np.float32 else torch.float64
    P0 = np.diag(np.repeat(np.array(list(params.p0_diag)), 3)[:N])[None]
    flt.set_state(np.zeros((B, 19)), np.zeros((B, 9)), np.repeat(P0, B, 0), np.zeros(B, np.int32))
    rep = lambda a: np.ascontiguousarray(np.broadcast_to(a, (B,) + a.shape[1:]))
    flt.init_gravity_bias(np.ascontiguousarray(np.broadcast_to(imu[:500, None, 1:4], (500, B, 3))),
                          np.ascontiguousarray(np.broadcast_to(imu[:500, None, 4:7], (500, B, 3))))
    meas0 = image[0:1, 1:9]
    flt.pose_init(rep(meas0[:, 0].astype(np.int32)[None]), rep(meas0[None, :, 1:4]), rep(meas0[None, :, 4:8]), 0)
    steps = 0
    for item in plan_windows(imu, image, max_frames, True, max_window):
        if item[0] == "reset":
            meas = item[1]
            flt.pose_init(rep(meas[:, 0].astype(np.int32)[None]), rep(meas[None, :, 1:4]), rep(meas[None, :, 4:8]), 1)
            continue
        up = lambda a, t=tt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(t)
        if item[0] == "predict":
            _, rows, dts = item
            for c0 in range(0, len(rows), 64):              # bounded uploads: (64, B, 3) per call
                r = rows[c0:c0 + 64]
                d_acc = up(imu[r, 1:4])[:, None, :].expand(len(r), B, 3).contiguous()
                d_gyr = up(imu[r, 4:7])[:, None, :].expand(len(r), B, 3).contiguous()
                d_dt = up(dts[c0:c0 + 64])
                flt.wait_stream(torch.cuda.current_stream())
                flt.predict_n(d_acc, d_gyr, d_dt, K=len(r))
            steps += len(rows)
            continue
        _, kcount, rows, dts, frames = item
        F, M = len(frames), max(len(m) for m in frames)
        ids = np.full((F, 1, M), -1, np.int32); pos = np.zeros((F, 1, M, 3)); quat = np.zeros((F, 1, M, 4)); quat[..., 0] = 1
        for f, m in enumerate(frames):
            ids[f, 0, :len(m)] = m[:, 0].astype(np.int32); pos[f, 0, :len(m)] = m[:, 1:4]; quat[f, 0, :len(m)] = m[:, 4:8]
        d_acc = up(imu[rows, 1:4])[:, None, :].expand(len(rows), B, 3).contiguous()
        d_gyr = up(imu[rows, 4:7])[:, None, :].expand(len(rows), B, 3).contiguous()
        d_ids = torch.from_numpy(ids).to(dev).expand(F, B, M).contiguous()
        d_pos = up(pos).expand(F, B, M, 3).contiguous()
        d_quat = up(quat).expand(F, B, M, 4).contiguous()
        d_dt = up(dts)
        flt.wait_stream(torch.cuda.current_stream())          # the uploads above ran on torch's stream
        flt.frames(kcount, d_acc, d_gyr, d_dt, d_ids, d_pos, d_quat, 0)
        steps += int(kcount.sum()) + F
    flt.sync()
    return steps


def replay_cpp_loop(engine, imu, image, params, max_frames=None, n_init=500):
    """The recording through the loop of FILTER::FilterThreadFunction (C++/src/filter.cpp:190-250) instead of the
    Matlab script's: per camera frame

        ResetSystemState()      filter.cpp:405-477  if the detection is more than 0.1 s past the state's time stamp the
                                nominal pose is reset from the nearest marker (v = ba = bg = 0, covariance and the
                                carried rotation matrix untouched, state time := detection time) -- and the frame
                                CONTINUES: it is not skipped as in FBUS_EKF.m:168-171;
        BatchImuProcessing()    filter.cpp:483-531  every buffered IMU sample with state time <= t <= detection time,
                                dt = t - state time (after a reset that window is empty or a single sample);
        ObservationUpdate()     filter.cpp:622-754  nearest marker (hysteresis in the C++ dialect).

    The state's time stamp is the time of the last IMU sample used (filter.cpp:516), which is also what the trace
    row carries (filter.cpp:241).  Gravity / gyro bias come from the first `n_init` IMU rows (the live filter averages
    what arrived during its first second, filter.cpp:256-285), the pose from the first frame (filter.cpp:291-399).
    Returns (states, npredict, nreset) with states[k] = [t_state, nominal(19), rot(9), P(N*N)] after frame k."""
    imu = np.asarray(imu, float)
    image = np.asarray(image, float)
    N = engine.N
    P0 = np.diag(np.repeat(np.array(list(params.p0_diag)), 3)[:N])[None]
    engine.set_state(np.zeros((1, 19)), np.zeros((1, 9)), P0, np.zeros(1, np.int32))
    engine.init_gravity_bias(imu[:n_init, None, 1:4], imu[:n_init, None, 4:7])
    j = 1
    while j < len(image) and image[j, 0] == image[0, 0]:
        j += 1
    meas = image[0:j, 1:9]
    engine.pose_init(meas[:, 0].astype(np.int32)[None], meas[None, :, 1:4], meas[None, :, 4:8], 0)
    t_state = image[0, 0]                                           # filter.cpp:375
    idx = int(np.searchsorted(imu[:, 0], t_state, side="right"))    # samples up to the detection are consumed (:300-305,388)
    n_img = j
    out, npred, nreset = [], [], 0
    while n_img < len(image) and (max_frames is None or len(out) < max_frames):
        j = n_img + 1
        while j < len(image) and image[j, 0] == image[n_img, 0]:
            j += 1
        t_det = image[n_img, 0]
        meas = image[n_img:j, 1:9]
        n_img = j
        ids, pos, quat = meas[:, 0].astype(np.int32)[None], meas[None, :, 1:4], meas[None, :, 4:8]
        if t_det - t_state > 0.1:                                   # filter.cpp:462
            ok = engine.pose_init(ids, pos, quat, 1)
            if ok is None or np.all(ok):                            # a marker out of range / not in the map: no reset (:432-447)
                t_state = t_det
                nreset += 1
        cnt = 0
        while idx < len(imu) and imu[idx, 0] <= t_det:              # filter.cpp:493-517
            if imu[idx, 0] >= t_state:
                engine.predict(imu[idx:idx + 1, 1:4], imu[idx:idx + 1, 4:7], np.array([imu[idx, 0] - t_state]))
                t_state = imu[idx, 0]
                cnt += 1
            idx += 1
        engine.correct(ids, pos, quat, 0)
        nominal, rot, P, _ = engine.get_state()
        out.append(np.concatenate([[t_state], nominal.ravel(), rot.ravel(), P.ravel()]).astype(np.float64))
        npred.append(cnt)
    return np.array(out), np.array(npred), nreset


# ---- recording files either side of the path ---------------------------------------
def load_recording(directory):
    """imu.txt (`t ax ay az gx gy gz`, main.cpp IMU callback) and image.txt (`t id px py pz qw qx qy qz`,
    vision.cpp:101-110; rows with equal t belong to one frame) of a dataset directory."""
    import os
    imu = np.loadtxt(os.path.join(directory, "imu.txt"), ndmin=2)
    image = np.loadtxt(os.path.join(directory, "image.txt"), ndmin=2)
    if imu.shape[1] != 7 or image.shape[1] != 9:
        raise ValueError(f"unexpected column counts: imu {imu.shape[1]} (want 7), image {image.shape[1]} (want 9)")
    return imu, image


def fusion_rows(states):
    """The fused trace the reference appends to data/fusion.txt (filter.cpp:238-248): one row per camera frame,
    `t p(3) q(wxyz) v(3) ba(3) bg(3)` = 17 columns, from the rows replay() returns
    (nominal layout p v q ba bg g)."""
    s = np.asarray(states, float)
    t, nom = s[:, 0:1], s[:, 1:20]
    return np.concatenate([t, nom[:, 0:3], nom[:, 6:10], nom[:, 3:6], nom[:, 10:13], nom[:, 13:16]], axis=1)


def save_fusion(path, states):
    np.savetxt(path, fusion_rows(states), fmt="%.9f")


def image_from_corners(engine, corners, geometry):
    """corners.txt rows -> image.txt rows through the device (`marker_pose_kernel`): what VISION::GetMarkerPose
    hands the filter.  corners: water `t id` + 8 left + 8 right normalised coordinates (vision.cpp:111-119), geometry
    VIS_REFRACTIVE / VIS_PINHOLE; land `t id` + four 3-D corners (`:120-124`), geometry VIS_CORNERS3D.
    Returns rows `t id px py pz qw qx qy qz`."""
    from . import capi
    c = np.asarray(corners, float)
    if geometry == capi.VIS_CORNERS3D:
        pos, quat = engine.marker_pose(c[:, 2:14], None, geometry)
    else:
        pos, quat = engine.marker_pose(c[:, 2:10], c[:, 10:18], geometry)
    return np.concatenate([c[:, 0:2], np.asarray(pos, float), np.asarray(quat, float)], axis=1)
