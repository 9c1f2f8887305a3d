/*
 * rgbdslam_oracle.h -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the tracking-and-fusion
 * hot path of diegothomas/RGB-D-SLAM (DemoStep1).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and
 * cannot be compiled in this image (nvcc / OpenCV 2.4 gpu / Eigen / GL / Win32 are all missing,
 * SURVEY.md section 8c).  Every function below therefore restates the reference's arithmetic from
 * its source text (file:line cited per function), under the fixed definition
 * "C promotion rules, round-to-nearest, no FMA contraction" (build with -ffp-contract=off).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product (librgbdslam_hip.so) never links or calls it.
 *
 * Conventions (reference: D1 = DemoStep1/DemoStep1):
 *   n = rows (height), m = cols (width); images row-major idx = i*m + j.
 *   VMap/NMap  f32x3 interleaved; RGB f32x3; RGBD f32x4 (.w = depth/100); Mask/Label u8.
 *   4x4 poses column-major (pose[0],pose[4],pose[8],pose[12] = row 0, D1/Kernel.cu:365).
 *   3x3 rotations column-major like Eigen (R[0],R[3],R[6] = row 0, D1/ICP.cu:98).
 */
#ifndef RGBDSLAM_ORACLE_H
#define RGBDSLAM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Stage 1: preprocessing ------------------------------------------------------------- */
/* D1/Kernel.cu:33-52 (VertexMapKinectKernel). depth_pitch in ELEMENTS (u16). */
void orc_vertex_map_kinect(float *VMap, const uint16_t *depth, size_t depth_pitch,
                           const float *calib, int n, int m);
/* D1/Kernel.cu:739-753 (CpyKernelChar3). bgr_pitch in BYTES per row. */
void orc_cpy_char3(const uint8_t *bgr, size_t bgr_pitch, float *rgb, int n, int m);
/* D1/Kernel.cu:123-307 (NormalKernel). */
void orc_compute_normal(float *NMap, const float *VMap, int n, int m, int inverse);
/* D1/Kernel.cu:670-682 (CpyKernel): img = (b+g+r)/3.0 */
void orc_intensity(float *img, const float *rgb, int n, int m);
/* cv::gpu::Sobel call sites D1/RG.cpp:809-810 (OpenCV 2.4 separable 3x3, reflect-101). */
void orc_sobel(const float *img, float *gx, float *gy, int n, int m);

/* ---- Stage 2+3: ICP normal equations --------------------------------------------------- */
/* Number of 8x8 blocks the reference launches (D1/ICP.cu:258-261). */
void orc_icp_grid(int n, int m, int fact, int *grid_rows, int *grid_cols);
/* D1/ICP.cu:72-249 (searchGauss + ProcessMatchKernelGaussNewto). buf = f32[27][nblk] laid out
 * buf[blk + term*nblk], blk = brow + bcol*grid_rows.  corr (optional, may be NULL) = int32[n*m]
 * indexed by the model pixel; -1 = no correspondence, else v*m+u of the matched current pixel. */
void orc_icp_match(const float *intr, const float *Rcurr, const float *tcurr,
                   const float *VMap, const float *NMap, const float *RGB,
                   const float *Gx, const float *Gy,
                   const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                   float distThres, float angleThres, int n, int m, int fact,
                   float *buf, int32_t *corr);
/* D1/ICP.cu:50-70 (ReduceKernel, STRIDE 512). */
void orc_icp_reduce(const float *buf, int nblk, float *out27);
/* D1/ICP.cu:254-286 (EstimateSystemGaussNewton): A row-major 6x6 symmetric, b[6]. */
void orc_estimate_system(const float *intr, const float *Rcurr, const float *tcurr,
                         const float *VMap, const float *NMap, const float *RGB,
                         const float *Gx, const float *Gy,
                         const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                         float distThres, float angleThres, int n, int m, int fact,
                         float *A, float *b, float *sums27);

/* Small dense helpers standing in for Eigen (D1/RG.cpp:834,862,873,923; versions unpinned). */
float orc_det6(const float *A);                          /* partial-pivot LU determinant       */
int   orc_llt_solve6(const float *A, const float *b, float *x); /* 0 ok, 1 = non-positive pivot */
void  orc_inverse3(const float *Rcm, float *out_cm);     /* cofactor inverse, column-major     */
/* D1/Utilities.cpp:1145-1167 */
void  orc_quaternion2matrix(const double *q, double m[3][3]);

typedef struct orc_icp_params {
    int   lvl;            /* 3 (D1/GP-demo.cpp:409) */
    int   max_iter[3];    /* {3,2,1} (D1/RG.h:124-126) */
    float distThres, angleThres;
    float thres_cvg_alpha, thres_cvg_tr; /* 1e-6 (D1/RG.h:111) */
} orc_icp_params;

/* D1/RG.cpp:797-932 (RG::AlignFrameGaussNewton).  poseR/poset are updated in place when tracking
 * succeeds. Returns 1 = ok, 0 = lost/singular.  trace (optional) receives, per executed iteration,
 * 27 sums + 6 solution floats (33 floats); *n_iters receives the count. */
int orc_align_gauss_newton(const orc_icp_params *prm, const float *intr,
                           const float *VMap, const float *NMap, const float *RGB,
                           const float *VMap_prev, const float *NMap_prev, const float *RGBD_prev,
                           int n, int m,
                           float *poseR, float *poset,
                           const float *poseRPrev, const float *posetPrev,
                           float *trace, int *n_iters);

/* ---- Stage 4: fusion -------------------------------------------------------------------- */
/* D1/Kernel.cu:440-526 (MergeDepthOffKernel). */
void orc_merge_depth_off(float *RGBD, const uint16_t *depth, size_t depth_pitch, const float *RGB,
                         uint8_t *mask, const float *calib, int n, int m);
/* D1/Kernel.cu:54-72 (VertexMapGLKernel). */
void orc_vertex_map_gl(float *VMap, const float *RGBD, const float *calib, int n, int m);
/* D1/Primitive.cu:267-357 (UpdateKernel) under the deterministic winner rule of DESIGN.md:
 * snapshot read of the cell mask; gate-passing writers beat eroders; among writers highest
 * mask_curr, ties -> lowest frame pixel index; a single erode step when only eroders hit a cell.
 * claim = scratch u64 per plane cell (claim_tab[L][cell]); stamp must grow every call.
 * plane_lo/plane_hi restrict to labels owned by this shard (plane-partitioned multi-GPU). */
void orc_update(uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                uint64_t **ClaimTab, uint32_t stamp,
                const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                const float *pose, int N_prim, int M_prim, int n, int m,
                int plane_lo, int plane_hi);
/* D1/Primitive.cu:31-93 (VertexFromBumpKernel); n,m = bump image size, outputs (n/lvl)x(m/lvl). */
void orc_vertex_from_bump(float *VMap, float *RGBA, float *MaskOut, const uint16_t *Bump,
                          const uint8_t *RGB_bump, const uint8_t *Mask, const float *param,
                          int n, int m, int lvl);

/* ---- Stage 5: prediction ---------------------------------------------------------------- */
/* D1/Kernel.cu:345-379 (TransformVertexMapKernel). */
void orc_transform_vertex_map(const float *pose, float *VMap, float *RGBD, int n, int m);
/* D1/Kernel.cu:381-438 (QuadTrimFrameKernel). */
/* D1/Primitive.cu:95-152 (variant 0), D2/Primitive.cu:367-424 (variant 1) */
void orc_quad_trim(uint32_t *index, const float *VMap, const uint8_t *Mask, unsigned short min_conf,
                   unsigned n, unsigned m, float thresh, int lvl, int variant);
void orc_quad_trim_frame(uint32_t *index, const float *VMap, int n, int m, float thresh);
/* Software restatement of the GL_QUADS draw (D1/GP-demo.cpp:98-113, D1/Frame.cpp:247-279):
 * see DESIGN.md "compute rasteriser".  index may be NULL (quads trimmed inline with thresh).
 * tex = f32x4[n*m] (r,g,b,depth/100), zero where nothing was drawn.  zbuf = scratch u64[n*m]. */
void orc_raster(float *tex, uint64_t *zbuf, const float *VMap, const float *RGBD,
                const uint32_t *index, float thresh, const float *calib,
                float znear, float zfar, int n, int m);
/* D1/Kernel.cu:309-343,885-907 (ReadKernel, incl. the memset of RGBD). */
void orc_read_frame(float *VMap, float *RGBD, const float *tex, const float *calib, int n, int m);
/* D1/Kernel.cu:634-667 (ProjectMaskKernel); negative projections count as out of view. */
void orc_project_mask(const float *pose, const float *calib, const uint8_t *mask,
                      uint8_t *mask_swap, const float *VMap, int n, int m);
/* D1/RG.cu:252-320,900-915 (3 x SegmentKernel), double-buffered per pass. Result in Label. */
void orc_segment(uint8_t *Label, uint8_t *Label_swap, const float *VMap, const float *NMap,
                 const float *pose, const float *Equations, int nbPlans, float epsilon, float alpha,
                 int n, int m);
/* D1/Kernel.cu:783-802 (AddZBUFFOffKernel) and 616-632 (InitMaskKernel). */
void orc_add_zbuff_off(const uint16_t *depth, size_t depth_pitch, const float *rgb_in, float *RGBD,
                       const float *calib, int n, int m);
void orc_init_mask(uint8_t *mask, const float *RGBD, int n, int m);

/* Test-harness stand-in for plane detection: label every pixel with a non-zero
 * normal by the first plane whose |n.p - d| <= eps and n.n_p >= alpha (world frame). */
void orc_seed_labels(uint8_t *Label, const float *VMap, const float *NMap, const float *pose,
                     const float *Equations, int nbPlans, float epsilon, float alpha, int n, int m);

/* ---- Whole frame loop: D1/GP-demo.cpp:138-296 (display) -------------------------------- */
typedef struct orc_slam orc_slam;
orc_slam *orc_slam_create(int n, int m, const float *calib, const orc_icp_params *prm,
                          int max_planes, int N_prim, int M_prim, float znear, float zfar);
void orc_slam_destroy(orc_slam *s);
/* Inject plane equations (n,e1,e2,d) x nb + centres (xyz) and create primitives as
 * RG::CreatePrimitives does (D1/RG.cpp:699-793); seeds labels on the current model frame. */
void orc_slam_set_planes(orc_slam *s, const float *equations, const float *centers, int nb);
int orc_slam_detect_planes(orc_slam *s, int min_size, int min_normals, float *equations_out, float *centers_out);
/* One display() pass. Returns 1 tracked, 0 lost (model re-seeded), 2 first frame (model seeded). */
int orc_slam_step(orc_slam *s, const uint16_t *depth, const uint8_t *bgr);
/* optional stage (Frame.cpp:50-54, commented out in the reference): cv::bilateralFilter(depth, 5, 100, 100) before everything */
void orc_bilateral_depth(uint16_t *dst, const uint16_t *src, size_t pitch, int d, double sigma_color, double sigma_space, int n, int m);
void orc_slam_set_bilateral(orc_slam *s, int on);
/* RG::Finish schedule (RG.cpp:149-271), run at the end of every tracked step: first detection when count > first_after (30),
 * re-detection when count % period == 0 (100; 0 = never); min_size / min_normals >= 0 override 5000 / 1000 */
void orc_slam_set_finish(orc_slam *s, int enabled, int first_after, int period, int min_size, int min_normals);
int orc_slam_finish_detections(const orc_slam *s);
int orc_slam_nb_planes(const orc_slam *s);
void orc_slam_get_pose(const orc_slam *s, float *R9_cm, float *t3);
/* Raw buffer access for parity tests: name in {"VMap","NMap","RGBD","Mask","Label",
 * "frame.VMap","frame.NMap","frame.RGB","Bump<k>","RGBb<k>","Maskb<k>"} */
const void *orc_slam_buffer(const orc_slam *s, const char *name, size_t *bytes);

#ifdef __cplusplus
}
#endif
/* ---- plane detection (detect_oracle.c) -- D1/RG.cpp:277-694, D1/Kernel.cu:684-781,
 *      D1/Utilities.cpp:1052-1092; Canny / erode / flood fill restated from OpenCV 2.4's published algorithms */
void orc_cpy_n(uint8_t *mat, const float *NMap, int n, int m, int chnl);
void orc_canny(const uint8_t *img, uint8_t *edges, double low_thresh, double high_thresh, int n, int m);
void orc_sum(uint8_t *out, const uint8_t *r, const uint8_t *g, const uint8_t *b, int n, int m);
void orc_discont(uint8_t *mat, const float *VMap, int n, int m, float thresh);
void orc_thresh(uint8_t *out, const uint8_t *in, uint8_t thresh, const uint8_t *Label, int n, int m);
void orc_erode_ellipse5(uint8_t *dst, const uint8_t *src, int n, int m);
void orc_find_blobs(const uint8_t *binary, int32_t *blob, int32_t *sizes, int n, int m);
int orc_fit_planes(const int32_t *blob, const int32_t *sizes, const float *VMap, const float *NMap, const float *pose,
                   uint8_t *Label, float *equations, float *centers, int n, int m, int min_size, int min_normals,
                   int max_planes);

/* threads the row loops run on: 1 in the default build, omp_get_max_threads() in librgbdslam_oracle_omp.so */
int orc_threads(void);

/* ---- the remaining operator functions the reference's host code calls (checker of csrc/compat.hip) ---- */
/* Kernel.cu:528-614 MergeDepthKinectKernel */
void orc_merge_depth_kinect(float *RGBD, const uint16_t *depth, const float *RGB, uint8_t *mask, const float *calib, int n, int m);
/* Kernel.cu:804-823 AddZBUFFKinectKernel */
void orc_add_zbuff_kinect(const uint16_t *depth, const float *rgb_in, float *RGBD, const float *calib, int n, int m);
/* Kernel.cu:670-682 CpyKernel (pitch in floats) */
void orc_gpu_cpy(float *img, size_t pitch, const float *rgb, int n, int m);
/* Kernel.cu:74-121 MapData2VBOKernel */
void orc_map_data_2_vbo(const uint8_t *color_bgra, const uint16_t *depth, const int32_t *color_coord, float *VMap, float *RGB,
                        const float *calib, int n, int m);
/* RG.cu:508-537 VotePlanKernel, with a defined summation order (rows left to right, then rows top to bottom) */
void orc_vote_plan(const uint8_t *Label, const float *VMap, float *centers, float *count, const float *pose, int nbPlans, int n, int m);
/* RG.cu:539-555 AffectPlanKernel */
void orc_affect_plan(uint8_t *Label, const int32_t *Buff, int n, int m);
/* Primitive.cu:502-556 + the wrapper's ratio (counter[0] / counter[1], 1 when nothing is occupied) */
float orc_overlap(const uint16_t *Bump, const uint8_t *Mask, const float *param, const float *param_in, int n, int m, int N_prim,
                  int M_prim, uint32_t *counter2);

/* 16-bit labels (models of more than 255 planes) and the tiled-model labelling stand-in; checkers of
 * rgbdslam_update_l16 / rgbdslam_label_planes_l16 */
void orc_update_l16(uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                    const float *VMap, const float *RGBD, const uint8_t *Mask, const uint16_t *Label16, const float *pose,
                    int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi);
void orc_label_planes_l16(uint16_t *Label16, const float *VMap, const float *NMap, const float *pose, float **ParamTab, int nb,
                          int N_prim, int M_prim, float epsilon, float alpha, int n, int m);

/* 16-bit labels through the tracker (the build's extension; the reference's label image is unsigned char, Frame.h:197): the
 * statements of orc_project_mask / orc_segment / orc_seed_labels on u16 images; checkers of rgbdslam_set_label_width(ctx, 2) */
void orc_project_label_l16(const float *pose, const float *calib, const uint16_t *label, uint16_t *label_swap, const float *VMap,
                           int n, int m);
void orc_segment_l16(uint16_t *Label, uint16_t *Label_swap, const float *VMap, const float *NMap, const float *pose,
                     const float *Equations, int nbPlans, float epsilon, float alpha, int n, int m);
void orc_seed_labels_l16(uint16_t *Label, const float *VMap, const float *NMap, const float *pose, const float *Equations,
                         int nbPlans, float epsilon, float alpha, int n, int m);
void orc_slam_set_label16(orc_slam *s, int on);

/* prediction from the planes: the Bump meshes (VertexFromBump Primitive.cu:31-93 + QuadTrim Primitive.cu:95-152) drawn with the
 * rasteriser rules of orc_raster; checkers of rgbdslam_planes_draw / rgbdslam_planes_resolve */
void orc_planes_draw(uint64_t *zbuf, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, int plane_lo,
                     int plane_hi, const float *pose_world_to_cam, int N, int M, int min_conf, float thresh, const float *calib,
                     float znear, float zfar, int n, int m);
void orc_planes_resolve(const uint64_t *zbuf, float *VMap, float *RGBD, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                        float **ParamTab, int plane_lo, int plane_hi, const float *pose_world_to_cam, int N, int M,
                        const float *calib, int n, int m);

#endif
