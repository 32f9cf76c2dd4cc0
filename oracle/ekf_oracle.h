/*
 * ekf_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of the FBUS-EKF filter hot path (predict = ImuUpdate,
 * correct = MeasureUpdate), written from scratch in plain C with dense n x n
 * arithmetic that follows the reference operation for operation.  It is the
 * checker the HIP kernels are compared against and the "port" CPU baseline
 * bench.py times.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never calls into it.
 *
 * PARITY STATUS: "parity unpinned" for the EKF arithmetic.  The reference
 * cannot be compiled (needs Eigen/OpenCV/aruco/glog) or run (no Matlab/Octave)
 * in this image and ships no known-answer test for ImuUpdate/MeasureUpdate;
 * its recorded fusion.txt comes from an older revision and does not
 * reproduce.  What pins this file instead: an independently written numpy twin
 * (oracle/ekf_oracle_np.py) that agrees to <=1e-12; finite-difference checks of the
 * two Jacobians against the oracle's own non-linear functions
 * (tests/test_oracle_jacobians_cpu.py, via fbo_transition / fbo_measurement);
 * algebraic invariants; and two loose bands of the recorded fusion.txt -- the
 * initial gyro bias and the RELATIVE motion of the IMU over the land recording
 * (0.1 m / 6 deg over a 0.85 m excursion; tests/test_oracle_cpu.py).  The neighbouring vision chain
 * (oracle/vision_oracle.c) IS pinned by the reference's recorded
 * corners.txt -> image.txt data.
 *
 * Reference files followed (paths relative to the upstream repository):
 *   predict : matlab/ImuUpdate.m:36-82 ; C++/src/filter.cpp:533-616
 *   correct : matlab/MeasureUpdate.m:37-103 ; C++/src/filter.cpp:622-741
 *   helpers : matlab/quaternion_*.m, axisangle_to_quaternion.m,
 *             vector_to_crossmat.m, rotmat_to_quaternion.m ;
 *             C++/include/matrix_math.hpp:26-99
 *   consts  : matlab/FBUS_EKF.m:32-39,68,83-112 ; C++/include/filter.hpp:28-34,63-125
 *             C++/config/paramconfig.yml:44-57 ; matlab/GetMarkerMap.m:1-63
 */
#ifndef FBUS_EKF_ORACLE_H
#define FBUS_EKF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define FBO_NMAX 18
#define FBO_MAX_MARKERS 32      /* map entries */
#define FBO_MAX_VISIBLE 16      /* markers per frame */
#define FBO_STACK_ROWS 32                 /* dense updates of up to this many rows keep their scratch on the stack (<= 29 KiB) */
#define FBO_MMAX (16 * FBO_MAX_VISIBLE)  /* stacked measurement rows (7 per marker pose, 12 per marker corners, 16 per marker stereo pixels) */

enum { FBO_DIALECT_MATLAB = 0, FBO_DIALECT_CPP = 1 };
enum { FBO_MODE_NEAREST = 0,    /* reference behaviour: one 7-row update, nearest marker */
       FBO_MODE_STACKED = 1 };  /* extension: all visible markers, 7M stacked rows       */
enum { FBO_COV_SIMPLE = 0,      /* (I-KH)P then symmetrise (reference)                   */
       FBO_COV_JOSEPH = 1 };    /* (I-KH)P(I-KH)' + K R K'                               */

typedef struct {
    int    dialect;             /* FBO_DIALECT_*                                          */
    int    nstate;              /* 18 (reference) or 15 (gravity block removed)           */
    double q_diag[4];           /* process noise added to v, theta, ba, bg diagonals      */
    double r_pos, r_quat;       /* measurement noise (pos rows, quat rows)                */
    double R_IL[9];             /* rotation part of flipped left T_SC (row-major)         */
    double P_IL[3];             /* -R_IL' * t                                             */
    double Q_IL[4];             /* quaternion of R_IL, wxyz                               */
    int    n_markers;
    int    marker_id[FBO_MAX_MARKERS];
    double marker_pos[FBO_MAX_MARKERS][3];
    double marker_quat[FBO_MAX_MARKERS][4];
    double switch_thres;        /* C++ marker hysteresis (paramconfig.yml:57)             */
    int    cov_form;            /* FBO_COV_*                                              */
} fbo_params;

typedef struct {
    double p[3], v[3], q[4], ba[3], bg[3], g[3];
    double R[9];                /* carried rotation matrix (possibly stale), row-major    */
    double P[FBO_NMAX * FBO_NMAX];  /* n x n row-major in the leading n*n entries         */
    int    prev_id;             /* C++ dialect: preUsedMarkerID_                          */
} fbo_state;

/* ---- L0 helpers (exported for unit tests) ---- */
void fbo_quat_mul(const double p[4], const double q[4], double out[4]);
void fbo_axisangle_to_quat(const double axis[3], double angle, double q[4]);
void fbo_quat_to_rotmat(const double q[4], double R[9]);          /* matlab formula   */
void fbo_quat_to_rotmat_eigen(const double q[4], double R[9]);    /* Eigen formula    */
void fbo_rotmat_to_quat(const double R[9], double q[4]);          /* trace based      */
void fbo_quat_left_matrix(const double q[4], double L[16]);
void fbo_quat_right_matrix(const double q[4], double Rm[16]);
void fbo_skew(const double v[3], double M[9]);
void fbo_expm_so3_neg(const double w[3], double dt, double E[9]); /* expm(-[w]x dt)   */

/* ---- constants ---- */
/* fills R_IL/P_IL/Q_IL from the raw (un-flipped) 4x4 row-major left T_SC,
 * applying diag(-1,-1,1,1) first (FBUS_EKF.m:68, filter.hpp:67-70).        */
void fbo_set_camera(fbo_params* prm, const double TSC_raw[16]);
/* adds one marker (position + row-major rotation matrix) to the map.       */
int  fbo_add_marker(fbo_params* prm, int id, const double pos[3], const double rot[9]);
/* reference defaults per dialect (Q, R, switch threshold, 12-marker map, camera). */
void fbo_default_params(fbo_params* prm, int dialect, int nstate);
/* P0 diag per dialect. */
void fbo_default_P0(const fbo_params* prm, double* P /* n*n */);

/* ---- the hot path ---- */
void fbo_predict(fbo_state* s, const fbo_params* prm,
                 const double accel[3], const double gyro[3], double dt);
/* returns 1 if an update was applied, 0 if skipped (no usable marker).     */
int  fbo_correct(fbo_state* s, const fbo_params* prm, int M,
                 const int* ids, const double* pos /*Mx3*/, const double* quat /*Mx4 wxyz*/,
                 int mode);

/* ---- the linearisation the hot path uses, exported so that tests can pin it by finite differences ---- */
/* Fx (n x n row-major) of ImuUpdate.m:63-69 / filter.cpp:597-604 at state s.                      */
void fbo_transition(const fbo_state* s, const fbo_params* prm, const double accel[3], const double gyro[3],
                    double dt, double* Fx);
/* h (7: position, quaternion after the sign unification against yq), H (7 x n), r (7) of marker `id`
 * (MeasureUpdate.m:67-88 / filter.cpp:684-721).  Returns 0 for an id outside the map.            */
int  fbo_measurement(const fbo_state* s, const fbo_params* prm, int id, const double* yp, const double* yq,
                     double* h, double* H, double* r);

/* ---- batched drivers (flat arrays; used by tests and the CPU baseline) ----
 * nominal: B x 19 (p v q ba bg g), rot: B x 9, P: B x n x n, prev: B ints   */
void fbo_predict_batch(int B, double* nominal, double* rot, double* P, int* prev,
                       const fbo_params* prm, const double* accel, const double* gyro,
                       const double* dt, int dt_stride, int nthreads);
void fbo_correct_batch(int B, double* nominal, double* rot, double* P, int* prev,
                       const fbo_params* prm, int M, const int* ids, const double* pos,
                       const double* quat, int mode, int* applied, int nthreads);

/* ---- corner-row measurement model: north-star extension, no reference counterpart (parity unpinned) ---- */
int  fbo_correct_corners(fbo_state* s, const fbo_params* prm, int M, const int* ids, const double* corners /*Mx12*/,
                         double size, int mode);
void fbo_correct_corners_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                               int M, const int* ids, const double* corners, double size, int mode, int* applied);

/* ---- pixel-row measurement model: flat-port reprojection of the corners, 2 (left) or 4 (stereo) rows per corner;
 *      north-star extension, no reference counterpart (parity unpinned); vision_params = const fbv_params*        ---- */
int  fbo_correct_pixels(fbo_state* s, const fbo_params* prm, const void* vision_params, int M, const int* ids,
                        const double* left /*Mx8*/, const double* right /*Mx8 or NULL*/, double size, double r_pix);
void fbo_correct_pixels_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                              const void* vision_params, int M, const int* ids, const double* left, const double* right,
                              double size, double r_pix, int* applied);
/* the same update with d pi / d X in closed form (vision_oracle.c::fbv_project_camera_jac, implicit-function theorem on the forward
 * model) instead of central differences: the second, independent pixel oracle -- exact to rounding, what the fp64 kernels are held
 * to at 1e-9.  Cross-checked against the central-difference one on the CPU. */
int  fbo_correct_pixels_analytic(fbo_state* s, const fbo_params* prm, const void* vision_params, int M, const int* ids,
                                 const double* left, const double* right, double size, double r_pix);
void fbo_correct_pixels_analytic_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                                       const void* vision_params, int M, const int* ids, const double* left, const double* right,
                                       double size, double r_pix, int* applied);

/* ---- init / reset / front door ---- */
void fbo_init_gravity_bias(int T, const double* accel, const double* gyro, double g[3], double bg[3]);
int  fbo_pose_init(fbo_state* s, const fbo_params* prm, int M, const int* ids, const double* pos,
                   const double* quat, int what, double max_dist, double* out7);
void fbo_pose_init_batch(int B, double* nominal, double* rot, const fbo_params* prm, int M, const int* ids,
                         const double* pos, const double* quat, int what, double max_dist,
                         const unsigned char* mask, double* out7, int* applied);
void fbo_imu_ema(int T, double* x, double* carry, int have_carry);

/* one camera frame per thread range: K predicts (accel/gyro K x B x 3, dt K values)
 * followed by one correct; threads are spawned once per call (CPU-baseline driver). */
void fbo_frame_batch(int B, double* nominal, double* rot, double* P, int* prev,
                     const fbo_params* prm, int K, const double* accel, const double* gyro,
                     const double* dt, int M, const int* ids, const double* pos,
                     const double* quat, int mode, int nthreads);

void fbo_schedule_batch(int B, double* nominal, double* rot, double* P, int* prev, const fbo_params* prm,
                        int nframes, const int* Ks, int reps, const double* accel, const double* gyro,
                        const double* dt, int M, const int* ids, const double* pos, const double* quat,
                        int mode, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
