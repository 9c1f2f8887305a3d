// rgbdslam_shim.cpp -- built INTO THE REFERENCE TREE in place of Kernel.cu / ICP.cu / Primitive.cu / RG.cu and linked with
// librgbdslam_hip.so: it defines, with the reference's own names and signatures, every free function of the four operator
// headers that the reference's host code (Frame.cpp, Graph.cpp, RG.cpp, RG.h) calls, and forwards each to the C ABI of
// include/rgbdslam.h.  Frame.cpp / Graph.cpp / RG.cpp therefore compile and link unchanged, with ONE exception that is a
// change of data, not of code: the cudaArray* handed to ReadFrame_cu is the f32x4 image written by PredictReplaceGL()
// below instead of a GL texture (the OpenGL prediction pass, GP-demo.cpp:98-113, is replaced by a compute rasteriser).
//
// tests/test_shim_compiles.py compiles this file against the reference's headers (plus integration/stub/cudaTypes.cuh for
// the CUDA / OpenCV / Windows type names) and links it against librgbdslam_hip.so together with a translation unit that
// takes the address of every function the reference's host code calls: an undefined symbol fails the test.
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "Kernel.cuh"
#include "ICP.cuh"
#include "Primitive.cuh"
#include "RG.cuh"

#include "rgbdslam.h"

static rgbdslam_ctx *g;          // the reference's file-scope device globals -> one context (one GPU, like cudaGLSetGLDevice(0))
static int g_nb_planes;          // entries of Graph's pointer tables, learnt from the Affect* calls (Graph.cpp:62-69)

// = checkCudaErrors (cudaTypes.cuh:36): print and exit
static void ck(int rc, const char *what) {
    if (rc) { fprintf(stderr, "rgbdslam: %s failed with code %d\n", what, rc); exit(EXIT_FAILURE); }
}
#define CK(call) ck((call), #call)
static rgbdslam_ctx *ctx() {
    if (!g) CK(rgbdslam_create(&g, 0, NULL));
    return g;
}
static void sync() { CK(rgbdslam_synchronize(ctx())); }   // every reference wrapper ends in cudaDeviceSynchronize()

// ---- Kernel.cuh ----------------------------------------------------------------------------------------------------------
void SetCalibrationMatrix(float *Calib_CPU) { CK(rgbdslam_set_calibration_matrix(ctx(), Calib_CPU)); }
void FreeCalibrationMatrix() {}
void AllocPoseMatrix() { (void)ctx(); }
void SetPoseMatrix(float *pose_CPU) { CK(rgbdslam_set_pose_matrix(ctx(), pose_CPU)); }
void FreePoseMatrix() {}

void VertexMapKinect(float *VMap, cv::gpu::DevMem2D_<unsigned short> depth, int n, int m) {
    CK(rgbdslam_vertex_map_kinect(ctx(), VMap, depth.data, depth.step, n, m)); sync();
}
void VertexMapGL(float *VMap, float *RGB, int n, int m) { CK(rgbdslam_vertex_map_gl(ctx(), VMap, RGB, n, m)); sync(); }
void MapData2VBO_cu(BYTE *d_colorFrame, USHORT *d_depthFrame, LONG *d_colorCoord, float *VMap, float *RGB, int n, int m) {
    CK(rgbdslam_map_data_2_vbo(ctx(), d_colorFrame, d_depthFrame, (const int32_t *)d_colorCoord, VMap, RGB, n, m)); sync();
}
void ComputeNormal(float *NMap, float *VMap, int n, int m, bool inverse) {
    CK(rgbdslam_compute_normal(ctx(), NMap, VMap, n, m, inverse ? 1 : 0)); sync();
}
// Array = the f32x4[n*m] image written by PredictReplaceGL (see the head of this file); NMap is untouched here as in the
// reference (Kernel.cu:885-907) -- PredictedFrame::ReadFrame clears and recomputes it itself (Frame.cpp:241-242)
void ReadFrame_cu(float *Vmap, float *RGBD, float *NMap, cudaArray *Array, int n, int m) {
    (void)NMap;
    CK(rgbdslam_read_frame(ctx(), Vmap, RGBD, (const float *)(const void *)Array, n, m)); sync();
}
void TransformVertexMap(float *VMap, float *RGBD, int n, int m) { CK(rgbdslam_transform_vertex_map(ctx(), VMap, RGBD, n, m)); sync(); }
void QuadTrimFrame(float *VMap, unsigned int *indices_dev, int n, int m, float thresh) {
    CK(rgbdslam_quad_trim_frame(ctx(), VMap, indices_dev, n, m, thresh)); sync();
}
void MergeDepthOff(float *RGB_predicted, cv::gpu::DevMem2D_<unsigned short> depth, float *RGB, unsigned char *mask, int n, int m) {
    CK(rgbdslam_merge_depth_off(ctx(), RGB_predicted, depth.data, depth.step, RGB, mask, n, m)); sync();
}
void MergeDepthKinect(float *RGB_predicted, unsigned short *depth, float *RGB, unsigned char *mask, int n, int m) {
    CK(rgbdslam_merge_depth_kinect(ctx(), RGB_predicted, depth, RGB, mask, n, m)); sync();
}
void InitMask_cu(unsigned char *mask, float *RGB, int n, int m) { CK(rgbdslam_init_mask(ctx(), mask, RGB, n, m)); sync(); }
void ProjectMask_cu(unsigned char *mask, unsigned char *mask_swap, float *VMap, int n, int m) {
    CK(rgbdslam_project_mask(ctx(), mask, mask_swap, VMap, n, m)); sync();
}
void gpu_cpy(cv::gpu::DevMem2D_<float> mat, float *rgb, int n, int m) { CK(rgbdslam_gpu_cpy(ctx(), mat.data, mat.step, rgb, n, m)); sync(); }
void gpu_cpyN(cv::gpu::DevMem2D_<unsigned char> mat, float *NMap, int n, int m, int chnl) {
    CK(rgbdslam_gpu_cpy_n(ctx(), mat.data, mat.step, NMap, n, m, chnl)); sync();
}
void gpu_Discont(cv::gpu::DevMem2D_<unsigned char> mat, float *VMap, int n, int m, float thresh) {
    CK(rgbdslam_gpu_discont(ctx(), mat.data, mat.step, VMap, n, m, thresh)); sync();
}
void gpu_cpy_char3(cv::gpu::DevMem2D_<uchar[3]> mat, float *rgb, int n, int m) {
    CK(rgbdslam_gpu_cpy_char3(ctx(), (const uint8_t *)mat.data, mat.step, rgb, n, m)); sync();
}
void gpu_Sum(cv::gpu::DevMem2D_<unsigned char> Out, cv::gpu::DevMem2D_<unsigned char> InR, cv::gpu::DevMem2D_<unsigned char> InG,
             cv::gpu::DevMem2D_<unsigned char> InB, int n, int m) {
    CK(rgbdslam_gpu_sum(ctx(), Out.data, Out.step, InR.data, InR.step, InG.data, InG.step, InB.data, InB.step, n, m)); sync();
}
void gpu_Thresh(cv::gpu::DevMem2D_<unsigned char> Out, cv::gpu::DevMem2D_<unsigned char> In, unsigned char thresh,
                unsigned char *Label, int n, int m) {
    CK(rgbdslam_gpu_thresh(ctx(), Out.data, Out.step, In.data, In.step, thresh, Label, n, m)); sync();
}
void gpu_add_zbuffOff(cv::gpu::DevMem2D_<unsigned short> depth, float *rgb_in, float *rgb, int n, int m) {
    CK(rgbdslam_gpu_add_zbuff_off(ctx(), depth.data, depth.step, rgb_in, rgb, n, m)); sync();
}
void gpu_add_zbuffKinect(unsigned short *depth, float *rgb_in, float *rgb, int n, int m) {
    CK(rgbdslam_gpu_add_zbuff_kinect(ctx(), depth, rgb_in, rgb, n, m)); sync();
}

// ---- ICP.cuh -------------------------------------------------------------------------------------------------------------
void AllocBuffers(int n, int m) { CK(rgbdslam_alloc_buffers(ctx(), n, m)); }
void FreeBuffers() { CK(rgbdslam_free_buffers(ctx())); }
void EstimateSystemGaussNewton(float *Rcurr_dev, float *tcurr_dev, float *VMap, float *NMap, float *RGB,
                               cv::gpu::DevMem2D_<float> Gx, cv::gpu::DevMem2D_<float> Gy, float *VMap_prev, float *NMap_prev,
                               float *RGB_prev, float *intr, float distThres, float angleThres, int n_row, int m_col, float *A,
                               float *b, int fact) {
    CK(rgbdslam_estimate_system_gauss_newton(ctx(), Rcurr_dev, tcurr_dev, VMap, NMap, RGB, Gx.data, Gx.step, Gy.data, Gy.step,
                                             VMap_prev, NMap_prev, RGB_prev, intr, distThres, angleThres, n_row, m_col, A, b, fact,
                                             NULL));   // blocking, like the reference's D2H copy (ICP.cu:271)
}

// ---- Primitive.cuh -------------------------------------------------------------------------------------------------------
void AllocPosePrimMatrix() { (void)ctx(); }   // pose_prim only feeds kernels that D1 never launches (UpdateBump, DetectContour)
void SetPosePrimMatrix(float *pose_CPU) { (void)pose_CPU; }
void FreePosePrimMatrix() {}
void AllocBufPrim() { (void)ctx(); }          // res_dev: the counter of Count_cu (no callers); Overlap_cu's counters are the library's
void FreeBufPrim() {}
void VertexFromBump(float *VMap_VBO, float *RGB_VBO, float *Mask_VBO, unsigned short *Bump_dev, unsigned char *RGB_bump_dev,
                    unsigned char *Mask_dev, float *param_dev, int n, int m, int lvl) {
    CK(rgbdslam_vertex_from_bump(ctx(), VMap_VBO, RGB_VBO, Mask_VBO, Bump_dev, RGB_bump_dev, Mask_dev, param_dev, n, m, lvl)); sync();
}
void QuadTrim(float *VMap, unsigned char *Mask, unsigned int *indices_dev, unsigned short min_conf, unsigned int n, unsigned int m,
              float thresh, int lvl) {
    CK(rgbdslam_quad_trim(ctx(), VMap, Mask, indices_dev, min_conf, n, m, thresh, lvl, 0)); sync();
}
void Update_cu(unsigned short **BumpTab, unsigned char **RGBTab, unsigned char **MaskTab, float **ParamTab, float *VMap, float *RGB,
               unsigned char *Mask, unsigned char *Label, float *pose, int N_prim, int M_prim, int n, int m) {
    CK(rgbdslam_update_cu(ctx(), BumpTab, RGBTab, MaskTab, ParamTab, VMap, RGB, Mask, Label, pose, N_prim, M_prim, n, m, g_nb_planes));
    sync();
}
float Overlap_cu(unsigned short *Bump, unsigned char *Mask, float *param, float *param_in, int n, int m, int N_prim, int M_prim) {
    float ratio = 1.0f;
    CK(rgbdslam_overlap(ctx(), Bump, Mask, param, param_in, n, m, N_prim, M_prim, &ratio, NULL));
    return ratio;
}

// ---- RG.cuh --------------------------------------------------------------------------------------------------------------
void Segment_cu(unsigned char *Label, float *VMap, float *NMap, float *pose, float *Equations, int nbPlans, float epsilon,
                float alpha, int n, int m) {
    CK(rgbdslam_segment_cu(ctx(), Label, VMap, NMap, pose, Equations, nbPlans, epsilon, alpha, n, m)); sync();
}
void VotePlan_cu(unsigned char *Label, float *VMap, float *centers, float *count, float *pose, int n, int m) {
    // the reference's kernel indexes centers / count by whatever label it meets; so does the library's: entries of labels
    // that do not occur in the image are never touched, hence the full u8 label range is safe to pass
    CK(rgbdslam_vote_plan(ctx(), Label, VMap, centers, count, pose, 255, n, m)); sync();
}
void AffectPlan_cu(unsigned char *Label, int *Buff, int n, int m) { CK(rgbdslam_affect_plan(ctx(), Label, Buff, n, m)); sync(); }
static void table_store(void **tab, const void *ptr, int indx) {
    CK(rgbdslam_table_store(ctx(), tab, ptr, indx));
    if (indx + 1 > g_nb_planes) g_nb_planes = indx + 1;
}
void Affect(float **Tab, float *in, int indx) { table_store((void **)Tab, in, indx); }
void AffectChar(unsigned char **Tab, unsigned char *in, int indx) { table_store((void **)Tab, in, indx); }
void AffectShort(unsigned short **Tab, unsigned short *in, int indx) { table_store((void **)Tab, in, indx); }

// ---- the one call site whose shape changes: DisplayPredictedFBO() (GP-demo.cpp:98-113) -------------------------------------
// Instead of drawing the quad mesh with OpenGL into the RGBA32F FBO and mapping its texture as a cudaArray, GP-demo.cpp calls
// this and passes the returned pointer to PredictedFrame::ReadFrame -> ReadFrame_cu above.  tex: f32x4[n*m]; zbuf: u64[n*m],
// filled with 0xFF once (rgbdslam_raster_clear); indices_dev may be NULL (quads trimmed inline with thresh).
cudaArray *PredictReplaceGL(float *tex, unsigned long long *zbuf, float *VMap, float *RGBD, unsigned int *indices_dev, float thresh,
                            float znear, float zfar, int n, int m) {
    CK(rgbdslam_predict_raster(ctx(), tex, (uint64_t *)zbuf, VMap, RGBD, indices_dev, thresh, znear, zfar, n, m)); sync();
    return (cudaArray *)(void *)tex;
}
