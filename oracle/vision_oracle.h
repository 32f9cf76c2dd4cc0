/*
 * vision_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of the step immediately in front of correct():
 * flat-port refractive stereo triangulation of the four ArUco corners and the
 * marker pose fit.  PINNED by the reference's own
 * recorded data: waterdata/dataset-06/corners.txt -> image.txt (refractive
 * triangulation + pose, <=1e-5 pos / 3e-5 quat at the files' 6-digit text
 * precision) and landdata/dataset-02/corners.txt -> image.txt (pose fit only).
 *
 * Reference lines followed (C++/src/vision.cpp):
 *   RefractionTriangulation : 472-618
 *   NormalTriangulation     : 395-466
 *   ComputeMarkerPose       : 624-759
 */
#ifndef FBUS_VISION_ORACLE_H
#define FBUS_VISION_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double R_IL[9], P_LI[3];    /* left  T_SC rotation / translation, RAW (vision.cpp uses the un-flipped matrices) */
    double R_IR[9], P_RI[3];    /* right T_SC rotation / translation, RAW */
    double n_air, n_glass, n_water;
    double d_air, d_glass;
    double normal[3];
} fbv_params;

/* camerainfo1.yml extrinsics + paramconfig.yml:31-42 refraction constants */
void fbv_default_params(fbv_params* p);

/* one marker: 4 corners, left/right undistorted normalised image coordinates
 * (x0 y0 x1 y1 x2 y2 x3 y3 each) -> 4 corner positions in the left camera frame. */
void fbv_refraction_triangulate(const fbv_params* p, const double left[8], const double right[8],
                                double corners[12]);
void fbv_normal_triangulate(const fbv_params* p, const double left[8], const double right[8],
                            double corners[12]);
/* 4 corner positions -> marker position, quaternion (wxyz), rotation (row-major) */
void fbv_marker_pose(const double corners[12], double pos[3], double quat[4], double rot[9]);

/* Forward flat-port projection (not in the reference; see vision_oracle.c): point in a camera's refraction frame
 * -> normalised image point; and a point of the left camera frame (as triangulated, flipped) -> left / right pixels.
 * Return 0 if the point is behind the port. */
int fbv_refraction_project(const fbv_params* p, const double Xp[3], double uv[2]);
int fbv_project_camera(const fbv_params* p, const double Xcam[3], int which /* 0 left, 1 right */, double uv[2]);
/* analytic Jacobians of the two projections above (J = d uv / d X, 2 x 3 row-major), from the implicit-function theorem */
int fbv_refraction_project_jac(const fbv_params* p, const double Xp[3], double uv[2], double J[6]);
int fbv_project_camera_jac(const fbv_params* p, const double Xcam[3], int which, double uv[2], double J[6]);
int fbv_project_stereo(const fbv_params* p, const double Xcam[3], double uvL[2], double uvR[2] /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
