/*
 * fbus_ekf.h -- C ABI of the MI355X-native batched error-state EKF.
 *
 * Drop-in boundary for ONE path of CASIA-RoboticFish/FBUS-EKF: the filter's
 * predict (ImuUpdate) and correct (MeasureUpdate) steps, batched over B
 * independent filters that live in device memory.  The reference exposes no
 * FFI of its own (predict/correct are private members / Matlab functions), so
 * each entry point cites the reference call contract it replaces.  Paths are
 * relative to the upstream repository.
 *
 * Conventions
 *   - one handle = one device, B filters, one dtype (32|64) and one error-state
 *     size (18 = reference, 15 = gravity block removed);
 *   - calls are stream-ordered and asynchronous until fbus_ekf_sync();
 *     a handle is not thread-safe;
 *   - every function returns FBUS_OK (0) or a negative/positive fbus_status;
 *     fbus_ekf_last_error() gives the text of the last failure on a handle;
 *   - "host" entry points take host pointers in the handle's dtype (float for
 *     32, double for 64) and stage them through library-owned device buffers;
 *     "_dev" entry points take device pointers and add no copies -- these are
 *     the hot path;
 *   - arrays are dense row-major: nominal B x 19 (p3 v3 q4[wxyz] ba3 bg3 g3),
 *     rot B x 9 (carried rotation matrix, see below), P B x N x N,
 *     accel/gyro B x 3, marker ids B x M (int32, -1 = absent),
 *     marker pos B x M x 3, marker quat B x M x 4 (wxyz);
 *   - there is NO CPU fallback: without a HIP device create() fails with
 *     FBUS_ERR_NO_DEVICE.
 */
#ifndef FBUS_EKF_H
#define FBUS_EKF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBUS_MAX_MARKERS 32     /* marker-map entries                     */
#define FBUS_MAX_VISIBLE 16     /* markers per frame and filter (M)       */
#define FBUS_MAX_MARKER_ID 1023 /* ArUco ids handled by the id->slot table */

typedef struct fbus_ekf* fbus_ekf_t;

typedef enum fbus_status {
    FBUS_OK = 0,
    FBUS_ERR_INVALID = 1,       /* bad argument                          */
    FBUS_ERR_NO_DEVICE = 2,     /* no usable HIP device                  */
    FBUS_ERR_HIP = 3,           /* a HIP runtime call failed             */
    FBUS_ERR_UNSUPPORTED = 4,   /* dtype / nstate / mode not built       */
    FBUS_ERR_NOMEM = 5,
    FBUS_ERR_ABI = 6            /* caller built against another header version (see FBUS_ABI_VERSION) */
} fbus_status;

/* Which of the reference's two implementations is reproduced. */
enum { FBUS_DIALECT_MATLAB = 0,   /* matlab/ImuUpdate.m, MeasureUpdate.m             */
       FBUS_DIALECT_CPP = 1 };    /* C++/src/filter.cpp                              */
/* Measurement handling in correct(). */
enum { FBUS_MODE_NEAREST = 0,     /* reference: nearest marker (C++: + hysteresis), 7 rows */
       FBUS_MODE_STACKED = 1 };   /* extension: all visible markers, 7M rows, one linearisation point */
/* Stereo geometry of corner inputs (marker_pose, correct_corners). */
enum { FBUS_VIS_REFRACTIVE = 0,   /* flat-port Snell ray trace (vision.cpp:472-618)      */
       FBUS_VIS_PINHOLE = 1,      /* 6x4 DLT (vision.cpp:395-466)                        */
       FBUS_VIS_CORNERS3D = 2 };  /* corner positions given, no triangulation            */
/* Covariance correction form. */
enum { FBUS_COV_SIMPLE = 0,       /* (I-KH)P then symmetrise (MeasureUpdate.m:101-102)  */
       FBUS_COV_JOSEPH = 1 };     /* (I-KH)P(I-KH)' + K R K'                            */

/* Replaces: EkfParam (C++/include/common.hpp:32-54), the constants built in the
 * FILTER ctor (C++/include/filter.hpp:63-125), matlab/FBUS_EKF.m:32-39,83-112,
 * MarkerPoseServer (common.hpp:19-29 / matlab/GetMarkerMap.m), CameraInfo::T_SC
 * and RefractInfo (common.hpp:58-103). */
typedef struct fbus_params {
    int32_t dialect;            /* FBUS_DIALECT_*                                       */
    int32_t cov_form;           /* FBUS_COV_*                                           */
    double  q_diag[4];          /* process noise added to the v, theta, ba, bg diagonals (not scaled by dt) */
    double  r_pos, r_quat;      /* measurement noise: position rows, quaternion rows    */
    double  p0_diag[6];         /* initial covariance per block p v theta ba bg g (used by fbus_ekf_reset_cov) */
    double  T_SC_left[16];      /* left camera T_SC, row-major 4x4, RAW (the diag(-1,-1,1,1) flip is applied inside) */
    double  T_SC_right[16];     /* right camera T_SC, RAW (triangulation only)          */
    int32_t n_markers;
    int32_t marker_id[FBUS_MAX_MARKERS];
    double  marker_pos[FBUS_MAX_MARKERS][3];
    double  marker_rot[FBUS_MAX_MARKERS][9];    /* row-major rotation marker->world     */
    double  switch_thres;       /* C++ dialect marker hysteresis (paramconfig.yml:57)   */
    double  max_dist;           /* marker_max_dist (paramconfig.yml:56), init/reset only */
    double  n_air, n_glass, n_water;            /* flat-port refraction (paramconfig.yml:31-42) */
    double  d_air, d_glass;
    double  port_normal[3];
    double  marker_size;        /* side of the square marker [m] (vision.hpp:114: 0.28); corner-row model only */
    double  r_pix;              /* noise of one normalised image coordinate (pixel-row model only); default 1e-6 */
} fbus_params;

/* Fills prm with the reference's constants for the given dialect
 * (FBUS_EKF.m / paramconfig.yml / camerainfo1.yml / GetMarkerMap.m). */
int fbus_params_default(fbus_params* prm, int dialect);
/* Host-side check of a parameter set (no device needed): dialect / cov_form values, positive noise, marker count and ids in
 * range.  FBUS_OK, or FBUS_ERR_INVALID with the reason in msg (may be NULL).  fbus_ekf_create applies the same checks. */
int fbus_params_validate(const fbus_params* prm, char* msg, size_t msg_len);

/* ---- ABI version ---------------------------------------------------------- */
/* fbus_params is passed by pointer and has grown (round 2 added r_pix); fbus_ekf_set_stream(h, NULL) changed
 * meaning in round 2 (NULL is HIP's legacy default stream now, FBUS_STREAM_OWN the handle's own stream).  A caller built
 * against an older header must not run silently against a newer library: fbus_ekf_create below is a macro that hands
 * the caller's compile-time sizeof(fbus_params) and FBUS_ABI_VERSION to fbus_ekf_create_checked, which refuses a
 * mismatch with FBUS_ERR_ABI.  (Bindings that cannot use the macro -- ctypes, loadlibrary -- call
 * fbus_ekf_abi_version() / fbus_params_size() once after loading and compare; the Python mirror does.)
 *   5  round 5: fbus_ekf_frame_meas_fused_dev, fbus_ekf_frames_meas_fused_dev (struct unchanged)
 *   4  round 4: fbus_ekf_set_policy_batch, fbus_ekf_launch_info (struct unchanged)
 *   3  round 3: FBUS_ERR_ABI, create_checked, team kernels (fbus_ekf_set_team), fbus_ekf_gather
 *   2  round 2: r_pix in fbus_params, set_stream(NULL) = legacy default stream
 *   1  round 1 */
#define FBUS_ABI_VERSION 6
int fbus_ekf_abi_version(void);
size_t fbus_params_size(void);

/* ---- lifetime ------------------------------------------------------------- */
/* Replaces: FILTER::FILTER (filter.hpp:63-137) for a batch of filters.
 * dtype 32|64, nstate 15|18.  device = HIP ordinal. */
int fbus_ekf_create_checked(fbus_ekf_t* out, const fbus_params* prm, size_t params_size, int abi_version,
                            int batch, int device, int dtype, int nstate);
int fbus_ekf_create(fbus_ekf_t* out, const fbus_params* prm, int batch, int device, int dtype, int nstate);
#ifndef FBUS_EKF_NO_ABI_CHECK
#define fbus_ekf_create(out, prm, batch, device, dtype, nstate) \
    fbus_ekf_create_checked((out), (prm), sizeof(fbus_params), FBUS_ABI_VERSION, (batch), (device), (dtype), (nstate))
#endif
int fbus_ekf_destroy(fbus_ekf_t h);
/* Run all work of this handle on an existing hipStream_t (e.g. the caller's
 * framework stream).  hip_stream is the hipStream_t itself: NULL (0) is HIP's
 * legacy default stream, as everywhere in HIP -- a caller whose framework is
 * on the default stream passes 0 and gets launches ordered with its own work.
 * FBUS_STREAM_OWN restores the handle's own stream, which is non-blocking:
 * it is ordered against NO other stream, so a caller that stays on it orders
 * its device buffers itself (fbus_ekf_sync / device sync, or the two calls
 * below). */
#define FBUS_STREAM_OWN ((void*)(intptr_t)-1)
int fbus_ekf_set_stream(fbus_ekf_t h, void* hip_stream);
/* Waves per 64-filter tile ("team" kernels; no reference counterpart -- the reference runs one filter on one thread,
 * filter.cpp:190-250).  predict_roles governs predict / predict_n and the fused frame / frame window entry points (fp32): 0 (default):
 * chosen per launch from how much of the chip a one-wave launch would leave idle -- predict: 3 waves per tile up to a quarter of the
 * chip's SIMDs in tiles (16 384 filters on MI355X), predict_n and the fused entry points: 4 up to half (32 768); 1: always one wave
 * per tile; 2..4: always that many.  correct_roles governs fbus_ekf_correct_corners (stacked mode) and fbus_ekf_correct_pixels, whose
 * markers are divided among the waves of a tile (both record types): 0 = four waves up to a quarter of the chip, two up to half;
 * 1 = never; 2 = two; 3..4 = four.  The pose-row fbus_ekf_correct always runs one wave per tile (its team form was measured slower at
 * every batch size and removed in round 4).  Results agree to fp32 rounding whatever the choice (predict: the same arithmetic, 1 ulp
 * on a few covariance elements where the compiler fuses a different product; the folds: double-precision sums in a different
 * order), so a caller that compares runs BIT FOR BIT across batch sizes pins the value -- or, for shards of one job, names the job's
 * size with fbus_ekf_set_policy_batch. */
int fbus_ekf_set_team(fbus_ekf_t h, int predict_roles, int correct_roles);
/* (round 4) The batch the automatic kernel-FAMILY choice above is keyed on.  0 (default): this handle's own batch.  A job that is
 * cut into shards over several handles / GPUs (fbus::ShardedFilter, bench.py --total-batch) names the WHOLE job here, so that every
 * shard layout runs the same kernels and produces the same bits as the single-handle run (the team and the one-wave kernels agree
 * to fp32 rounding only).  No reference counterpart (one filter, one thread: filter.cpp:190-250). */
int fbus_ekf_set_policy_batch(fbus_ekf_t h, int total_filters);
/* What the launch policy of this handle is (decided at create from the device: CU count, cache sizes; environment overrides are
 * read there once -- no entry point reads the environment afterwards).  `arg`: K for FBUS_INFO_ROLES_PREDICT, M for
 * FBUS_INFO_ROLES_MEAS, else ignored. */
enum { FBUS_INFO_SIMDS = 0,            /* SIMDs of the device = CUs x 4 (FBUS_FAKE_SIMDS overrides it: tests)              */
       FBUS_INFO_ONE_ROUND_FILTERS = 1,/* filters of one wave per SIMD = SIMDs x 64                                         */
       FBUS_INFO_TWO_WAVE_MIN_B = 2,   /* from this many filters on the <= 256-register kernel forms run                   */
       FBUS_INFO_BIG_RECORDS_MB = 3,   /* records above this take the default cache policy in predict                      */
       FBUS_INFO_MALL_MB = 4, FBUS_INFO_L2_KB = 5,
       FBUS_INFO_POLICY_BATCH = 6,     /* see fbus_ekf_set_policy_batch                                                    */
       FBUS_INFO_ROLES_PREDICT = 7,    /* waves per tile the next predict (arg = 1) / predict_n (arg = K) would use       */
       FBUS_INFO_ROLES_MEAS = 8,       /* ... correct_corners (stacked) / correct_pixels with arg = M marker slots          */
       FBUS_INFO_TEAM_FRAMES = 9,      /* 1: the fused frame / frame window entry points use the team kernel              */
       FBUS_INFO_MEAS_SPLIT = 10 };    /* (round 5) correct_pixels with arg = M: 0 = one wave applies the update, 2 / 4 = the update divided
                                          between a solver and an updater wave, that many waves per tile (fp32, square port) */
int fbus_ekf_launch_info(fbus_ekf_t h, int what, int arg, int* value);
/* Cross-stream ordering without a host sync (hipEventRecord + hipStreamWaitEvent):
 * wait_stream   -- work submitted to the handle's stream after this call starts
 *                  only when everything already submitted to other_stream is done
 *                  (inputs produced on the caller's stream);
 * signal_stream -- the reverse (outputs consumed on the caller's stream). */
int fbus_ekf_wait_stream(fbus_ekf_t h, void* other_stream);
int fbus_ekf_signal_stream(fbus_ekf_t h, void* other_stream);
int fbus_ekf_sync(fbus_ekf_t h);
const char* fbus_ekf_last_error(fbus_ekf_t h);
const char* fbus_status_string(int status);

/* ---- state I/O (also serves as checkpoint / resume) ------------------------ */
/* Replaces: direct member access to NominalState/ErrorState (common.hpp:205-247)
 * and the Matlab State struct (FBUS_EKF.m:74-85).  Any pointer may be NULL to
 * skip that part.  prev_id is the C++ dialect's preUsedMarkerID_ (filter.hpp). */
int fbus_ekf_set_state(fbus_ekf_t h, const void* nominal, const void* rot, const void* P, const int32_t* prev_id);
int fbus_ekf_get_state(fbus_ekf_t h, void* nominal, void* rot, void* P, int32_t* prev_id);
int fbus_ekf_set_state_dev(fbus_ekf_t h, const void* nominal, const void* rot, const void* P, const int32_t* prev_id);
int fbus_ekf_get_state_dev(fbus_ekf_t h, void* nominal, void* rot, void* P, int32_t* prev_id);
/* P <- diag(p0_diag) for every filter. */
int fbus_ekf_reset_cov(fbus_ekf_t h);
/* The packed device-resident records (what a multi-GPU gather ships): base
 * pointer, bytes per filter and total bytes.  Layout: DESIGN.md section 3. */
int fbus_ekf_records(fbus_ekf_t h, void** dev_ptr, size_t* bytes_per_filter, size_t* total_bytes);
/* ---- multi-GPU: one process (rank) per GPU, one handle per rank, ONE collective -------------------------------
 * No reference counterpart: the reference runs one filter on one CPU thread (filter.cpp:190-250).  Filters are independent, so
 * the ranks step their shards with no communication; the single exchange of the path is the gather of the packed records at the
 * end, over RCCL (xGMI inside a node).  RCCL is bound at first use (dlopen of librccl.so.1); the single-GPU path never needs it.
 *   comm_unique_id  rank 0 creates the 128-byte ncclUniqueId and hands it to the other ranks (any out-of-band channel);
 *   comm_init       every rank: ncclCommInitRank on the handle's device;
 *   comm_attach     alternatively adopt an existing ncclComm_t of the caller's (not destroyed by the library);
 *   gather          every rank's records into out_dev on every rank, on the handle's stream (asynchronous until
 *                   fbus_ekf_sync).  bytes_of_rank = NULL: all ranks hold the same number of bytes (ncclAllGather, rank k at
 *                   offset k * bytes); otherwise world entries (entry [rank] = this handle's fbus_ekf_records total): ragged
 *                   shards, rank k at the running offset (grouped ncclBroadcast). */
int fbus_ekf_comm_unique_id(void* id128);
int fbus_ekf_comm_init(fbus_ekf_t h, const void* id128, int rank, int world);
int fbus_ekf_comm_attach(fbus_ekf_t h, void* nccl_comm, int rank, int world);
int fbus_ekf_comm_destroy(fbus_ekf_t h);
int fbus_ekf_gather(fbus_ekf_t h, void* out_dev, const size_t* bytes_of_rank);
/* (round 4) ONE process that owns several devices (fbus::NodeFilter -- the closest thing to the reference's one stateful object,
 * filter.cpp:190-250, owning a whole node).  comm_init_all: ncclCommInitAll over the handles' devices, rank k = handles[k] (one handle
 * per device).  gather_group: the fbus_ekf_gather of all n ranks inside one RCCL group (out_dev[k] on handles[k]'s device).
 * copy_records: this handle's packed records to a buffer on any device of the process, on the handle's stream (hipMemcpyPeerAsync) --
 * the gather when the consumer sits on one device, or when the shards share a device; no communicator needed. */
int fbus_ekf_comm_init_all(fbus_ekf_t* handles, int n);
int fbus_ekf_gather_group(fbus_ekf_t* handles, int n, void* const* out_dev, const size_t* bytes_of_rank);
int fbus_ekf_copy_records(fbus_ekf_t h, void* dst, int dst_device);
/* Point the handle at caller-owned device storage of total_bytes (as reported
 * by fbus_ekf_records) so that a framework tensor can alias the records. */
int fbus_ekf_attach_records(fbus_ekf_t h, void* dev_ptr, size_t total_bytes);

/* ---- predict == ImuUpdate -------------------------------------------------- */
/* Replaces: State = ImuUpdate(State, accel, gyro, dt) (matlab/ImuUpdate.m:36) and
 * FILTER::UpdateCovariance + UpdateNominalState (filter.cpp:588-616,533-582) as
 * called per IMU sample from BatchImuProcessing (filter.cpp:505-516).
 * dt_per_filter = 0: dt points to ONE value; 1: B values. */
int fbus_ekf_predict(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter);
int fbus_ekf_predict_dev(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter);
/* K consecutive IMU samples in one launch (state stays in registers between
 * samples).  accel/gyro are K x B x 3, dt is K (dt_per_filter=0) or K x B.
 * Same arithmetic, same results as K fbus_ekf_predict calls. */
int fbus_ekf_predict_n(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter);
int fbus_ekf_predict_n_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter);

/* ---- (round 6) the same host-pointer signatures WITHOUT the wait -------------- */
/* Replaces the same calls as fbus_ekf_predict / _predict_n / _correct / _correct_pixels, as the reference's caller issues them: one
 * call per IMU sample / camera frame that returns at once (FILTER::SetImuData hands the sample to the filter thread under a mutex,
 * filter.cpp:24-55; BatchImuProcessing then runs one predict per sample, :505-516; SetDetectionResult copies the detections,
 * :61-65).  Arguments, checks, arithmetic and results are those of the synchronous entry points; what differs:
 *   - every HOST array is taken by value at the call.  Pageable memory is copied into a pinned ring slot of the handle's by the
 *     calling thread -- the caller's buffer is free again on return.  Memory that is already pinned (hipHostMalloc, hipHostRegister,
 *     fbus_ekf_host_register below; pieces of >= 4 KiB) is transferred IN PLACE and must stay unchanged until
 *     fbus_ekf_async_inputs_consumed() or fbus_ekf_sync() returns;
 *   - nothing waits for the device: the H2D copy runs on a copy stream, the kernel on the handle's stream behind it, so the copy of
 *     call i + 1 overlaps the kernel of call i.  The ring has 8 slots; the ninth call in flight waits for the first one's kernel
 *     (back-pressure; counted by fbus_ekf_async_stats);
 *   - completion and device-side errors surface at fbus_ekf_sync() or at any host-pointer result (fbus_ekf_get_state,
 *     fbus_ekf_get_applied), which are unchanged.  Not capturable into a HIP graph (FBUS_ERR_INVALID between graph_begin / _end).
 * Measured (tools/host_api_rate.py). */
int fbus_ekf_predict_async(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter);
int fbus_ekf_predict_n_async(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter);
int fbus_ekf_correct_async(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip);
int fbus_ekf_correct_pixels_async(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right /* may be NULL */,
                                  const uint8_t* skip);
/* every H2D copy issued so far has completed: pinned input arrays handed to the _async calls may be rewritten (cheaper than
 * fbus_ekf_sync: does not wait for the kernels) */
int fbus_ekf_async_inputs_consumed(fbus_ekf_t h);
/* counters since create: _async calls, calls that had to wait for a ring slot, input pieces transferred in place (any may be NULL) */
int fbus_ekf_async_stats(fbus_ekf_t h, int64_t* calls, int64_t* waits, int64_t* direct_pieces);
/* page-lock / release a host range for in-place transfers (hipHostRegister / hipHostUnregister for callers without HIP headers) */
int fbus_ekf_host_register(void* ptr, size_t bytes);
int fbus_ekf_host_unregister(void* ptr);

/* ---- correct == MeasureUpdate ---------------------------------------------- */
/* Replaces: State = MeasureUpdate(State, visionMeas[8xM], markerMap, cameraInfo)
 * (matlab/MeasureUpdate.m:37) and FILTER::ObservationUpdate (filter.cpp:622-754)
 * fed by SetDetectionResult (filter.cpp:61-65).  ids B x M (-1 = slot unused),
 * pos B x M x 3, quat B x M x 4.  skip (B bytes, may be NULL): non-zero = leave
 * that filter untouched.  Filters with no usable marker are left untouched
 * (the reference's silent early return, filter.cpp:671-673). */
int fbus_ekf_correct(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat,
                     int mode, const uint8_t* skip);
int fbus_ekf_correct_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat,
                         int mode, const uint8_t* skip);
/* 1 where the last correct() applied an update, 0 where it returned early. */
int fbus_ekf_get_applied(fbus_ekf_t h, uint8_t* applied_host);

/* ---- correct from stereo corners (north-star extension; NO reference counterpart) ------------ */
/* The per-corner measurement model BASELINE.json's north_star describes, in the form SURVEY.md section 0.1 (B2)
 * fixes: the marker's four corners are triangulated on the device (flat-port refractive or pin-hole, the
 * arithmetic of fbus_ekf_marker_pose: vision.cpp:472-618 / :395-466) and each corner position is a 3-row
 * measurement h_k = R_IL R'(P_m + R_m c_k - p - R P_IL), c_k = corner k in the marker frame of
 * VISION::ComputeMarkerPose (vision.cpp:736-759), noise r_pos per row: 12 rows per marker, Jacobians of the
 * reference's position rows (MeasureUpdate.m:72-73), same gain / injection / covariance algebra as correct().
 * left/right: B x M x 8 normalised corner coordinates (FBUS_VIS_REFRACTIVE / FBUS_VIS_PINHOLE), or
 * left = B x M x 12 corner positions (FBUS_VIS_CORNERS3D).  mode as in correct() (nearest = by corner 0).
 * The reference has nothing to compare this with: it is validated against the fp64 oracle only. */
int fbus_ekf_correct_corners(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                             int geometry, int mode, const uint8_t* skip);
int fbus_ekf_correct_corners_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                                 int geometry, int mode, const uint8_t* skip);

/* ---- one camera frame: K predicts then one correct -------------------------- */
/* Replaces one iteration of the frame loop (matlab/FBUS_EKF.m:175-196;
 * filter.cpp:232-235): enqueues K per-sample predict launches followed by one
 * correct launch without returning to the caller in between.  Device pointers. */
int fbus_ekf_frame_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                       int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip);

/* Same frame in ONE launch: the records stay in registers between the K predicts and the correct
 * (one HBM round trip per frame instead of one per EKF step).  Same arithmetic and results as
 * fbus_ekf_frame_dev.  M = 0: predicts only. */
int fbus_ekf_frame_fused_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                             int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip);

/* (round 5) The same frame with the NORTH STAR's MeasureUpdate -- correct() from corner pixels (FBUS_MEAS_PIXELS: the rows of
 * fbus_ekf_correct_pixels_dev; right may be NULL = left camera; geometry and mode are ignored) or from stereo corners
 * (FBUS_MEAS_CORNERS: fbus_ekf_correct_corners_dev with its geometry and mode) -- in place of the pose rows: K predicts
 * (matlab/ImuUpdate.m:36-82 ; filter.cpp:505-516) and the update (matlab/MeasureUpdate.m:84-102 with the reprojection rows of the
 * flat-port model, vision.cpp:496-599 run forward) in ONE launch, the record resident in registers / LDS in between.  Same
 * arithmetic as K fbus_ekf_predict_dev calls + one fbus_ekf_correct_pixels_dev / _corners_dev call, and equal results TO FP32
 * ROUNDING (the single-step gate of tests/util.py: which product of an a b + c d becomes an FMA differs between the specialised
 * kernels); bit-equal are the update alone (K = 0) to the per-call update and a window to its sequence of frames -- where that
 * update runs one wave per tile: more than half a chip of tiles, or fbus_ekf_set_team(h, ., 1).  Smaller launches and fp64
 * records run as fbus_ekf_predict_n_dev + the per-call update.  M = 0: predicts only.  left / right: 16-byte aligned. */
enum { FBUS_MEAS_PIXELS = 0, FBUS_MEAS_CORNERS = 1 };
int fbus_ekf_frame_meas_fused_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                                  int kind, int M, const int32_t* ids, const void* left, const void* right, int geometry, int mode,
                                  const uint8_t* skip);

/* (round 5) A WINDOW of such frames in ONE launch (offline replay of a recorded stretch of corners.txt through the north star's model: the
 * frame loop of matlab/FBUS_EKF.m:151-210 with the reprojection / corner rows): nframes times { kcount[f] predicts, the update }, the
 * records resident from the first load to the last store.  Same arithmetic and results as nframes calls of
 * fbus_ekf_frame_meas_fused_dev (bit-equal where that takes its resident kernel; elsewhere this entry point runs frame by frame).
 *   kcount HOST array, nframes entries (<= FBUS_MAX_WINDOW_FRAMES), each 0..255;  accel, gyro [sum kcount][B][3], dt [sum kcount] or [..][B]
 *   ids [nframes][B][M], left / right [nframes][B][M][8] ([..][12] corner positions for FBUS_VIS_CORNERS3D), skip [nframes][B] or NULL
 * fbus_ekf_get_applied afterwards reports the LAST frame of the window. */
int fbus_ekf_frames_meas_fused_dev(fbus_ekf_t h, int nframes, const int32_t* kcount, const void* accel, const void* gyro, const void* dt,
                                   int dt_per_filter, int kind, int M, const int32_t* ids, const void* left, const void* right, int geometry,
                                   int mode, const uint8_t* skip);

/* A WINDOW of camera frames in ONE launch (offline replay of a recorded stretch: the frame loop of
 * matlab/FBUS_EKF.m:151-210 / FilterThreadFunction, filter.cpp:229-235): nframes times { kcount[f] predicts, one
 * correct } with the records resident in registers from the first load to the last store.  Same arithmetic and
 * results as nframes calls of fbus_ekf_frame_fused_dev.
 *   kcount  HOST array, nframes entries (<= FBUS_MAX_WINDOW_FRAMES), each 0..255: IMU samples in front of frame f
 *   accel, gyro  [sum kcount][B][3], dt [sum kcount] or [sum kcount][B] (dt_per_filter)       device
 *   ids [nframes][B][M], pos [nframes][B][M][3], quat [nframes][B][M][4], skip [nframes][B] or NULL   device
 * fbus_ekf_get_applied afterwards reports the LAST frame of the window. */
#define FBUS_MAX_WINDOW_FRAMES 64
int fbus_ekf_frames_fused_dev(fbus_ekf_t h, int nframes, const int32_t* kcount, const void* accel, const void* gyro,
                              const void* dt, int dt_per_filter, int M, const int32_t* ids, const void* pos,
                              const void* quat, int mode, const uint8_t* skip);

/* ---- initialisation / reset (the callers' side of the path) ------------------- */
/* Replaces: InitGravityAndGyrobias (matlab/InitGravityAndGyrobias.m:36-40) /
 * FILTER::InitializeGravityAndBias (filter.cpp:256-285).  accel, gyro: T x B x 3.
 * Writes g = (0, 0, -|mean accel|) and bg = mean gyro of every filter. */
int fbus_ekf_init_gravity_bias(fbus_ekf_t h, int T, const void* accel, const void* gyro);
int fbus_ekf_init_gravity_bias_dev(fbus_ekf_t h, int T, const void* accel, const void* gyro);
/* Replaces: InitPositionAndQuaternion (matlab/InitPositionAndQuaternion.m:38-80) /
 * FILTER::InitializePose (filter.cpp:291-399) [FBUS_POSE_INIT: p, q, R from the nearest
 * marker, g = (9.8, 0, 0)] and ResetState (matlab/ResetState.m:37-80) /
 * FILTER::ResetSystemState (filter.cpp:405-477) [FBUS_POSE_RESET: p, q, v = 0, ba = 0;
 * Matlab dialect also refreshes R, C++ dialect also zeroes bg and leaves R stale].
 * mask (B bytes, may be NULL = all): non-zero selects the filters to (re)initialise.
 * C++ dialect: markers farther than params.max_dist are refused (filter.cpp:343-347).
 * fbus_ekf_get_applied() tells which filters were changed. */
enum { FBUS_POSE_INIT = 0, FBUS_POSE_RESET = 1 };
int fbus_ekf_pose_init(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                       const uint8_t* mask);
int fbus_ekf_pose_init_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                           const uint8_t* mask);
/* Replaces: ComputeVisionOnlyResults (matlab/ComputeVisionOnlyResults.m:39-79) /
 * positionOnlyVisual, quaternionOnlyVisual (filter.cpp:449-456).  out_pose: B x 7 (p3, q4 wxyz);
 * the state is not touched. */
int fbus_ekf_vision_only_pose(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat,
                              void* out_pose);
int fbus_ekf_vision_only_pose_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat,
                                  void* out_pose);
/* Replaces: the IMU pre-filter of FILTER::SetImuData (filter.cpp:36-47):
 * y[t] = 0.9 y[t-1] + 0.1 x[t] per filter, in place on T x B x 3 accel / gyro arrays; the last
 * filtered sample is carried in the handle to the next call (restart != 0 forgets it: the next
 * sample passes through, as the first buffered sample does in the reference). */
int fbus_ekf_imu_ema(fbus_ekf_t h, int T, void* accel, void* gyro, int restart);
int fbus_ekf_imu_ema_dev(fbus_ekf_t h, int T, void* accel, void* gyro, int restart);

/* ---- marker pose from stereo corners (the step in front of correct) --------- */
/* Replaces: VISION::RefractionTriangulation (C++/src/vision.cpp:472-618) or
 * VISION::NormalTriangulation (:395-466) followed by VISION::ComputeMarkerPose
 * (:624-759), for n markers at once (n = B x M when the output feeds correct()).
 * geometry FBUS_VIS_REFRACTIVE / FBUS_VIS_PINHOLE: left/right are n x 8 undistorted
 * normalised image coordinates of the four corners (x0 y0 .. x3 y3, the corners.txt
 * layout written at vision.cpp:111-119).  FBUS_VIS_CORNERS3D: `left` is n x 12 corner
 * positions in the left camera frame (the alternative log at vision.cpp:120-124) and
 * `right` is ignored.  Outputs: pos n x 3, quat n x 4 (wxyz) -- exactly the arrays
 * correct() takes -- and, if not NULL, corners3d n x 12.  Uses the handle's
 * T_SC_left/right and refraction constants, dtype and stream. */
int fbus_ekf_marker_pose(fbus_ekf_t h, int n, int geometry, const void* left, const void* right,
                         void* pos, void* quat, void* corners3d);
int fbus_ekf_marker_pose_dev(fbus_ekf_t h, int n, int geometry, const void* left, const void* right,
                             void* pos, void* quat, void* corners3d);

/* ---- HIP graphs -------------------------------------------------------------- */
/* Capture any sequence of *_dev calls on this handle once (between graph_begin and graph_end: the calls are
 * recorded, not executed; no host-pointer entry points, no sync, no timing inside) and replay it with ONE
 * launch.  The captured launches keep the device pointers they were given: replay reads the same input
 * buffers (refill them in place between replays).  For launch-bound inner loops: small batches, long runs of
 * per-step launches.  New: the reference runs one filter on one CPU thread and has nothing comparable. */
int fbus_ekf_graph_begin(fbus_ekf_t h);
int fbus_ekf_graph_end(fbus_ekf_t h, int* graph_id);
int fbus_ekf_graph_launch(fbus_ekf_t h, int graph_id);
int fbus_ekf_graph_destroy(fbus_ekf_t h, int graph_id);

/* ---- measurement support (bench / profiling) -------------------------------- */
enum { FBUS_KERNEL_PREDICT = 0, FBUS_KERNEL_CORRECT = 1, FBUS_KERNEL_PREDICT_N = 2, FBUS_KERNEL_MARKER_POSE = 3,
       FBUS_KERNEL_FRAME = 4, FBUS_KERNEL_CORRECT_CORNERS = 5, FBUS_KERNEL_COUNT = 6 };
/* When enabled (on >= 1), launches of the listed kernels are bracketed by HIP events on
 * the handle's stream; read() synchronises and returns the summed device time and launch
 * count since the last reset.  fbus_ekf_frame_dev uses ONE pair around its run of K
 * back-to-back predict launches (a pair per launch costs ~8 us of stream time and reads
 * ~3 us long) and brackets only every `on`-th frame (on = 1: every frame). */
int fbus_ekf_timing_enable(fbus_ekf_t h, int on);
int fbus_ekf_timing_reset(fbus_ekf_t h);
int fbus_ekf_timing_read(fbus_ekf_t h, int kernel, double* total_ms, int64_t* launches);

/* ---- correct() from corner PIXELS: the north star's reprojection rows ------------ */
/* No reference counterpart (the reference has only the back-projection VISION::RefractionTriangulation,
 * vision.cpp:472-618).  Measurement = the normalised image points of the four corners of every visible marker in the left
 * camera (right == NULL: 2 rows per corner, 128 rows at 16 markers) or in both cameras (4 rows per corner).
 * h = pi(X_k(x)): X_k = R_IL R'(P_m + R_m c_k - p - R P_IL) the corner in the left camera frame (the geometry of
 * MeasureUpdate.m:67,72-73 with the corner in place of the marker origin), pi = the flat-port forward projection
 * (air -> glass -> water, the inverse of the ray construction of vision.cpp:505-552, in the plane of the port normal and the
 * point: a closed-form thin-port start taken twice, then ONE Halley step in double -- two for fp64 records --; no iteration).
 * Per-corner Jacobian rows (2 x N) = d pi/dX (closed form) x [ -R_IL R' | R_IL [R'(c_w - p)]x ];
 * all rows of all visible markers at one linearisation point, folded into the 6x6 information matrix; the update is the one-shot
 * form  P(J,:) <- G P(J,:),  P_rr -= x_a' S^-1 x_c  (symmetric by construction, nothing cancels on the rows the measurement
 * shrinks).  fbus_params::cov_form DOES NOT APPLY to this entry point nor to fbus_ekf_correct_corners: FBUS_COV_JOSEPH selects
 * nothing here (the one-shot form already has what Joseph's form is chosen for); it governs fbus_ekf_correct only.  The one form
 * is checked against BOTH forms of the oracle -- (I - K H) P, MeasureUpdate.m:101-102, and Joseph's (I - K H) P (I - K H)' + K R K',
 * what north_star names: fp64 records 1e-9, fp32 the standard gates (tests/test_pixels_gpu.py::test_correct_pixels_matches_the_oracle,
 * tests/test_vision_gpu.py::test_correct_from_stereo_corners_matches_oracle).
 * left / right must be 16-byte aligned (any allocation is): the slots are fetched with 16-byte loads, FBUS_ERR_INVALID otherwise.
 * ids (B, M), left / right (B, M, 8) = x0 y0 .. x3 y3 (the column layout of corners.txt, vision.cpp:111-119). */
int fbus_ekf_correct_pixels(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right /* may be NULL */,
                            const uint8_t* skip);
int fbus_ekf_correct_pixels_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                                const uint8_t* skip);

/* ---- L0 helpers on the device (unit-test hook) -------------------------------- */
/* Evaluates ONE of the device inline helpers the kernels are built from for n independent inputs -- what
 * matlab/quaternion_*.m, vector_to_crossmat.m, axisangle_to_quaternion.m and C++/include/matrix_math.hpp:26-99
 * compute -- so that each helper can be tested against the oracle's restatement on its own, not only through
 * predict/correct.  a, b, out: HOST arrays of the handle's scalar type, n rows each.
 *   FBUS_L0_QUAT_MUL         a (n,4) (x) b (n,4) -> out (n,4)     quaternion_add.m:22-28 / Eigen Quaterniond::operator*
 *   FBUS_L0_QUAT_TO_ROTMAT_M a (n,4)             -> out (n,9)     quaternion_to_rotmat.m:22-33
 *   FBUS_L0_QUAT_TO_ROTMAT_E a (n,4)             -> out (n,9)     Eigen toRotationMatrix (filter.cpp:542,562,564)
 *   FBUS_L0_QUAT_NORMALIZE   a (n,4)             -> out (n,4)     quaternion_normalize.m:22-24
 *   FBUS_L0_EXPM_SO3_NEG     a (n,3) = w, b (n,1) = dt -> out (n,9) = expm(-[w]x dt)   ImuUpdate.m:68 (closed form of predict_nominal)
 *   FBUS_L0_DTHETA_TO_QUAT   a (n,3) = dtheta    -> out (n,4)     axisangle_to_quaternion.m:22-29 as used by the injection (MeasureUpdate.m:94)
 *   FBUS_L0_SINCOS_HALF      a (n,1) = x         -> out (n,4) = sin x, cos x, sin x/2, cos x/2 (the polynomial / library switch) */
enum { FBUS_L0_QUAT_MUL = 0, FBUS_L0_QUAT_TO_ROTMAT_M = 1, FBUS_L0_QUAT_TO_ROTMAT_E = 2, FBUS_L0_QUAT_NORMALIZE = 3,
       FBUS_L0_EXPM_SO3_NEG = 4, FBUS_L0_DTHETA_TO_QUAT = 5, FBUS_L0_SINCOS_HALF = 6 };
int fbus_ekf_l0_eval(fbus_ekf_t h, int op, int n, const void* a, const void* b, void* out);

#ifdef __cplusplus
}
#endif
#endif /* FBUS_EKF_H */
