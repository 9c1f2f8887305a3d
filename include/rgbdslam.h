/*
 * rgbdslam.h -- C ABI of librgbdslam_hip.so: the MI355X (gfx950) implementation of the
 * tracking-and-fusion hot path of diegothomas/RGB-D-SLAM (DemoStep1).
 *
 * Every entry point replaces one free host function that the reference declares in its four
 * operator headers (Kernel.cuh, ICP.cuh, Primitive.cuh, RG.cuh) and that its L3/L4 classes call
 * with raw device pointers; the reference declaration is cited per function
 * (paths relative to DemoStep1/DemoStep1/).  Differences from the reference signatures:
 *   - the file-scope globals (calib_dev, pose_dev, buf_dev, outbuf_dev ...) live in an opaque
 *     rgbdslam_ctx, one per GPU;
 *   - cv::gpu::DevMem2D_<T> arguments become (pointer, row pitch in bytes);
 *   - functions return int: 0 = ok, < 0 = -(hipError_t), > 0 = RGBDSLAM_E_* argument errors;
 *     nothing aborts the process (the reference's checkCudaErrors exits);
 *   - work is enqueued on the context's stream and is asynchronous unless stated; call
 *     rgbdslam_synchronize() (the reference ends every wrapper with cudaDeviceSynchronize()).
 *
 * Layout conventions: n = rows, m = cols, images row-major idx = i*m + j; VMap/NMap f32x3;
 * RGB f32x3; RGBD f32x4 (.w = depth/100); Mask/Label u8; 4x4 poses column-major;
 * 3x3 rotations column-major (Eigen), all exactly as in the reference.
 */
#ifndef RGBDSLAM_H
#define RGBDSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGBDSLAM_OK 0
#define RGBDSLAM_E_ARG 1      /* null pointer / non-positive size / unsupported value */
#define RGBDSLAM_E_STATE 2    /* call order (e.g. buffers not allocated) */
#define RGBDSLAM_E_LIMIT 3    /* size beyond a documented maximum */

typedef struct rgbdslam_ctx rgbdslam_ctx;

/* ---- context ------------------------------------------------------------------------------ */
/* Creates the per-GPU context on `device`; stream = hipStream_t (NULL = the null stream). */
int rgbdslam_create(rgbdslam_ctx **out, int device, void *stream);
int rgbdslam_destroy(rgbdslam_ctx *ctx);
int rgbdslam_set_stream(rgbdslam_ctx *ctx, void *stream);
int rgbdslam_get_stream(rgbdslam_ctx *ctx, void **stream);
int rgbdslam_synchronize(rgbdslam_ctx *ctx);
/* rgbdslam_synchronize() polls the stream for up to `microseconds` before it blocks (default 5000, or RGBDSLAM_SYNC_SPIN_US
 * at create; 0 = block at once).  A blocked wait is woken tens of microseconds after the stream drains; the frame is 112 us.
 * No reference counterpart (the reference blocks in cudaDeviceSynchronize after every wrapper). */
int rgbdslam_set_sync_spin(rgbdslam_ctx *ctx, int microseconds);
/* Auxiliary ICP records of a frame (no reference counterpart: MatchKernelGaussNewton forms these four values per sample and
 * iteration, ICP.cu:167-173): f32x4 per pixel = Gx / 8, Gy / 8, (r + g + b) / 3 and the depth weight 0.0012 / (0.0012 +
 * 0.0019 (z' - 0.4)^2), z' = -VMap.z -- each the reference's expression on the frame's stored maps, so the sums are unchanged.
 * aux_write (16-byte aligned, n*m*16 bytes, or NULL): the NEXT launch that preprocesses a frame (rgbdslam_frame_preprocess,
 * ..._model_segment_merge_claim_preprocess, ..._update_apply_tracked_preprocess) leaves that frame's records there.  aux_read
 * (or NULL): the NEXT rgbdslam_track_frame[_sharded], in either launch form, reads them (one 16-byte gather instead of five
 * words from three maps, ~60 fewer issue slots per sample); they must belong to the maps passed to that call.  One call each. */
int rgbdslam_set_frame_aux(rgbdslam_ctx *ctx, float *aux_write, const float *aux_read);
/* The same records from maps that already exist (callers of the per-operator entry points): aux[n*m] f32x4, 16-byte aligned. */
int rgbdslam_frame_aux(rgbdslam_ctx *ctx, const float *VMap, const float *RGB, const float *Gx, size_t gx_pitch, const float *Gy,
                       size_t gy_pitch, float *aux, int n, int m);
/* Stores up to RGBDSLAM_COPY_MAX_SEGMENTS device ranges into memory the device can write -- pinned host memory, or device
 * memory -- with ONE launch on the context's stream (no reference counterpart: the reference reads its poses back with blocking
 * cudaMemcpy, RG.cpp:898-903; a hipMemcpyAsync per range is a blit launch each).  Addresses and sizes are multiples of 4
 * bytes; asynchronous, the data is on the host after rgbdslam_synchronize(). */
#define RGBDSLAM_COPY_MAX_SEGMENTS 4
int rgbdslam_copy_segments(rgbdslam_ctx *ctx, const void *const *src_dev, void *const *dst, const size_t *bytes, int count);
const char *rgbdslam_version(void);
/* Instrumentation (no reference counterpart; the reference's only timing is clock() prints behind
 * `#define TIMING`, Utilities.h:82): brackets every later launch of the kernel called `kernel_name`
 * (e.g. "k_icp_match") with a HIP event pair on the context's stream.  NULL switches it off.
 * profile_read synchronises, returns the summed durations and the launch count, and resets. */
int rgbdslam_profile_kernel(rgbdslam_ctx *ctx, const char *kernel_name);
int rgbdslam_profile_read(rgbdslam_ctx *ctx, double *total_ms, int *launches);
/* mean elapsed time of `pairs` event pairs with nothing between their two records (the markers' own cost) */
int rgbdslam_profile_event_overhead(rgbdslam_ctx *ctx, int pairs, double *mean_ms);
/* Measurement probe, no counterpart in the reference: the rate a plain streaming kernel reaches on this card's HBM
 * (SURVEY 8(d) asks for the per-kernel fractions against a measured figure as well as the nominal one).  `dst`, `src`:
 * caller's device buffers of `bytes` bytes, 16-byte aligned, src holding finite floats.  mode 0 = copy (counts
 * 2 x bytes), 1 = fill dst (bytes), 2 = read src (bytes).  One untimed launch, then `reps` timed ones; blocking. */
int rgbdslam_profile_hbm_stream(rgbdslam_ctx *ctx, void *dst, const void *src, size_t bytes, int mode, int reps,
                                double *gbytes_per_s);
/* Debug: int64[7] device words that receive wall_clock64 stamps (10 ns ticks) of the phases of the
 * last ICP iteration kernel: [0] first workgroup start, [1..6] the tail workgroup's start, rounds
 * done, its line of sums stored, (unused), all lines arrived + tree done, solve done.  NULL switches it off. */
int rgbdslam_debug_icp_timing(rgbdslam_ctx *ctx, long long *timing7_dev);

/* Kernel.cuh:7-8   void SetCalibrationMatrix(float *Calib_CPU); FreeCalibrationMatrix()
 * calib = {fx, fy, cx, cy, k1..k5 (unused), ds, factor} (GP-demo.cpp:17,404). */
int rgbdslam_set_calibration_matrix(rgbdslam_ctx *ctx, const float *calib11);
/* Kernel.cuh:9-11  AllocPoseMatrix / SetPoseMatrix(float *pose_CPU) / FreePoseMatrix
 * 16 floats, column-major; used by transform_vertex_map and project_mask. */
int rgbdslam_set_pose_matrix(rgbdslam_ctx *ctx, const float *pose16);

/* ---- stage 1: preprocessing --------------------------------------------------------------- */
/* Kernel.cuh:14  void VertexMapKinect(float *VMap, DevMem2D_<unsigned short> depth, int n, int m) */
int rgbdslam_vertex_map_kinect(rgbdslam_ctx *ctx, float *VMap, const uint16_t *depth,
                               size_t depth_pitch_bytes, int n, int m);
/* Kernel.cuh:32  void gpu_cpy_char3(DevMem2D_<uchar[3]> mat, float *rgb, int n, int m) */
int rgbdslam_gpu_cpy_char3(rgbdslam_ctx *ctx, const uint8_t *bgr, size_t bgr_pitch_bytes,
                           float *rgb, int n, int m);
/* Kernel.cuh:17  void ComputeNormal(float *NMap, float *VMap, int n, int m, bool inverse) */
int rgbdslam_compute_normal(rgbdslam_ctx *ctx, float *NMap, const float *VMap, int n, int m,
                            int inverse);
/* Kernel.cuh:28  void gpu_cpy(DevMem2D_<float> mat, float *rgb, int n, int m)  followed by the two
 * cv::gpu::Sobel calls of RG.cpp:809-810, fused.  img may be NULL.  Pitches in bytes. */
int rgbdslam_intensity_sobel(rgbdslam_ctx *ctx, const float *rgb, float *img, size_t img_pitch,
                             float *gx, size_t gx_pitch, float *gy, size_t gy_pitch, int n, int m);

/* OPTIONAL, default-off stage of OffLineFrame::LoadFrame: the filter the reference carries commented out at Frame.cpp:50-54
 * (depth.convertTo(CV_32FC1); bilateralFilter(tmp, out, 5, 100.0, 100.0); out.convertTo(CV_16UC1)).  cv::bilateralFilter is
 * third-party (OpenCV 2.4): restated from its published float algorithm (circular window of radius d/2, reflect-101 border,
 * 4096-bin interpolated exponential table for the range weights, spanning the full u16 range).  dst != src; radius <= 4. */
int rgbdslam_bilateral_depth(rgbdslam_ctx *ctx, uint16_t *dst, size_t dst_pitch_bytes, const uint16_t *src,
                             size_t src_pitch_bytes, int d, double sigma_color, double sigma_space, int n, int m);

/* ---- stage 2+3: ICP normal equations ------------------------------------------------------ */
/* ICP.cuh:6-7  void AllocBuffers(int n, int m); void FreeBuffers()  (n, m = 8x8-block grid; the
 * scratch here is a fixed 27 x 512 floats, so the arguments only have to be positive) */
int rgbdslam_alloc_buffers(rgbdslam_ctx *ctx, int grid_rows, int grid_cols);
int rgbdslam_free_buffers(rgbdslam_ctx *ctx);
/* ICP.cuh:9-12  void EstimateSystemGaussNewton(float *Rcurr_dev, float *tcurr_dev, float *VMap,
 *   float *NMap, float *RGB, DevMem2D_<float> Gx, DevMem2D_<float> Gy, float *VMap_prev,
 *   float *NMap_prev, float *RGB_prev, float *intr, float distThres, float angleThres,
 *   int n_row, int m_col, float *A, float *b, int fact)
 * Rcurr_dev(9) tcurr_dev(3) intr(4) are DEVICE pointers; A(36, row-major) b(6) HOST outputs.
 * BLOCKING (like the reference's cudaMemcpy D2H).  One launch does what the reference's two
 * kernels do; the 27 sums are accumulated in exactly the reference's order (8x8-block tree, then
 * 512-strided chains + tree) and are therefore bit-identical to the oracle's.  corr_dev (optional, may be NULL): int32[n*m] debug map,
 * -1 or the matched current-frame pixel index, indexed by model pixel. */
int rgbdslam_estimate_system_gauss_newton(rgbdslam_ctx *ctx, const float *Rcurr_dev,
                                          const float *tcurr_dev, const float *VMap,
                                          const float *NMap, const float *RGB,
                                          const float *Gx, size_t gx_pitch,
                                          const float *Gy, size_t gy_pitch,
                                          const float *VMap_prev, const float *NMap_prev,
                                          const float *RGB_prev, const float *intr_dev,
                                          float distThres, float angleThres, int n_row, int m_col,
                                          float *A, float *b, int fact, int32_t *corr_dev);

/* RG.cpp:797-932  bool RG::AlignFrameGaussNewton -- the whole coarse-to-fine Gauss-Newton loop
 * resident on the device (match + reduce + 6x6 solve + pose update per iteration, no host round
 * trips).  pose_state_dev = f32[24] device memory: poseR(9, col-major) poset(3) poseRPrev(9)
 * posetPrev(3); poseR/poset are updated in place on success.  status_dev = int32[4] device
 * memory: [0] 1 tracked / 0 lost, [1] iterations executed, [2] the same per level
 * (8 bits each, level 0 = full resolution in the low byte), [3] reserved.  Asynchronous. */
typedef struct rgbdslam_icp_params {
    int   lvl;                 /* 3 */
    int   max_iter[3];         /* {3,2,1}  RG.h:124-126 */
    float distThres, angleThres;
    float thres_cvg_alpha, thres_cvg_tr;   /* 1e-6  RG.h:111 */
} rgbdslam_icp_params;
int rgbdslam_align_gauss_newton(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm,
                                const float *VMap, const float *NMap, const float *RGB,
                                const float *Gx, size_t gx_pitch, const float *Gy, size_t gy_pitch,
                                const float *VMap_prev, const float *NMap_prev,
                                const float *RGB_prev, int n_row, int m_col,
                                float *pose_state_dev, int32_t *status_dev, float *trace_dev);

/* Persistent-loop mode of rgbdslam_track_frame: all Gauss-Newton iterations of a frame (the loop of
 * D1/RG.cpp:838-903) run inside ONE kernel launch, the iterations separated by a device-wide epoch word
 * instead of kernel boundaries.  Same arithmetic, same results.  The kernel needs all of its 512
 * workgroups on the device at once, i.e. this process has the GPU to itself and the frame loop uses one
 * stream; the mode is therefore off unless the caller opts in (or RGBDSLAM_ICP_LOOP=1 is set when the
 * buffers are allocated).  Returns RGBDSLAM_E_STATE when the device cannot hold the grid.  A launch
 * whose workgroups give up waiting (50 ms) marks the frame as lost and raises the fault count. */
int rgbdslam_set_icp_loop(rgbdslam_ctx *ctx, int on);
/* faults[0] = launches that gave up since the last call; synchronises the stream.  A fault switches the mode off. */
int rgbdslam_icp_loop_faults(rgbdslam_ctx *ctx, int *faults);

/* The ICP kernels hand 128-byte lines (chain sums, loop state) between workgroups with relaxed agent-scope stores of one wave
 * instruction and accept a line when the sequence word in each of its four 32-byte sectors is current -- which is sufficient
 * only if gfx950 writes and reads such a sector indivisibly (measured so, not documented).  Every line also
 * carries an integrity word (XOR of its payload); with the check switched on (or RGBDSLAM_ICP_LINE_CHECK=1 when the buffers are
 * allocated) every accepted line is verified, a failing one is counted and read again, and the result no longer depends on that
 * assumption (+ ~0.5 us per Gauss-Newton iteration).  rgbdslam_icp_line_rejections: count since the last call; synchronises. */
int rgbdslam_set_icp_line_check(rgbdslam_ctx *ctx, int on);
int rgbdslam_icp_line_rejections(rgbdslam_ctx *ctx, unsigned int *rejections);
/* Debug: give-up time of the device-side waits in 10 ns ticks (default 5 000 000 = 50 ms; 0 = every waiter leaves at once). */
int rgbdslam_debug_set_icp_timeout(rgbdslam_ctx *ctx, int ticks);
/* Debug (no reference counterpart): this rank's tail waits `ticks` x 10 ns between publishing its line of the row-sharded
 * exchange and its first poll -- the rank that falls behind in the tests of the double-buffered exchange slots. */
int rgbdslam_debug_set_exchange_delay(rgbdslam_ctx *ctx, int ticks);

/* Row-sharded form of the same loop (no reference counterpart: the reference is single-GPU;
 * SURVEY.md 8e / BASELINE.json configs[3]).  Each rank owns the 8x8-block rows
 * [block_row_lo, block_row_hi) of the level's block grid (divUp(n/2^lvl, 8) rows) and produces its
 * own 27 partial sums; the caller all-gathers them (RCCL, stream-ordered) and every rank runs the
 * identical solve on the rank-ordered total, so all ranks hold bit-identical poses.  Sequence:
 * align_begin; for lvl = prm->lvl-1..0, iter = 0..max_iter[lvl]+1: align_band_system, <all-gather>,
 * align_solve_gathered; align_end.  All asynchronous; converged/failed iterations exit on device. */
int rgbdslam_align_begin(rgbdslam_ctx *ctx, const float *pose_state_dev, int32_t *status_dev);
int rgbdslam_align_band_system(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap,
                               const float *NMap, const float *RGB, const float *Gx, size_t gx_pitch,
                               const float *Gy, size_t gy_pitch, const float *VMap_prev,
                               const float *NMap_prev, const float *RGB_prev, int n_row, int m_col,
                               int lvl, int iter, int block_row_lo, int block_row_hi, float *sums27_dev);
int rgbdslam_align_solve_gathered(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm,
                                  const float *gathered_dev, int parts, const float *pose_state_dev,
                                  int lvl, int iter, float *trace_dev);
int rgbdslam_align_end(rgbdslam_ctx *ctx, float *pose_state_dev, int32_t *status_dev);

/* The same row-sharded iteration with the exchange INSIDE the launch (one launch per iteration per rank instead of three plus
 * a host-driven collective): every rank's kernel writes its 27 band sums as one checksummed 128-byte line into its slot of
 * every peer's exchange block (peer-mapped fine-grained device memory: xGMI stores, no RCCL call), waits until all ranks'
 * lines of this iteration have arrived in its own block, adds them in rank order and runs the identical solve.
 * Set-up: rgbdslam_exchange_create on every rank (returns an IPC handle), all-gather the handles by any means,
 * rgbdslam_exchange_connect.  world <= 8; every rank must issue the same sequence of rgbdslam_align_band_exchange calls and own
 * a non-empty band.  A rank that waits 50 ms in vain marks the frame lost and raises the fault count (rgbdslam_icp_loop_faults). */
#define RGBDSLAM_EXCHANGE_HANDLE_BYTES 64
int rgbdslam_exchange_create(rgbdslam_ctx *ctx, int world, int rank, void *handle_out);
int rgbdslam_exchange_connect(rgbdslam_ctx *ctx, const void *all_handles);
int rgbdslam_exchange_destroy(rgbdslam_ctx *ctx);
int rgbdslam_align_band_exchange(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                                 const float *RGB, const float *Gx, size_t gx_pitch, const float *Gy, size_t gy_pitch,
                                 const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, int n_row, int m_col,
                                 int lvl, int iter, int block_row_lo, int block_row_hi, const float *pose_state_dev,
                                 float *trace_dev);


/* ---- fused frame pipeline ------------------------------------------------------------------
 * One entry point per METHOD of the reference's L3/L4 classes instead of one per operator; same
 * per-pixel arithmetic (results are bit-identical to the per-operator entry points).  The tracker
 * state is a device-resident rgbdslam_pose_block: rgbdslam_track_frame writes it, the model_*
 * entry points read their poses from it and do nothing when `tracked` is 0 (model_reseed: nothing
 * when it is 1), so a whole display() pass (GP-demo.cpp:138-296) is a fixed, sync-free launch
 * sequence.  The model is double-buffered (set "in" -> set "out"). */
typedef struct rgbdslam_pose_block {
    float pose_state[24];  /* poseR(9, col-major) poset(3) poseRPrev(9) posetPrev(3)  RG.h:55-58 */
    int32_t status[4];     /* [0] 1 tracked / 0 lost, [1] iterations, [2] per level (8 bits each), [3] frames tracked so far */
    float P_rel[16];       /* model -> current camera, col-major 4x4 (pose_inv of GP-demo.cpp:201-211) */
    float P_rel_inv[16];   /* GP-demo.cpp:225-236 */
    float P_glob[16];      /* current camera -> world (pose of GP-demo.cpp:230-235,254-259) */
} rgbdslam_pose_block;
/* OffLineFrame::LoadFrame (Frame.cpp:43-77: gpu_cpy_char3 + VertexMapKinect + ComputeNormal) plus
 * gpu_cpy and the two cv::gpu::Sobel calls of RG.cpp:807-810, one kernel. */
int rgbdslam_frame_preprocess(rgbdslam_ctx *ctx, const uint16_t *depth, size_t depth_pitch_bytes,
                              const uint8_t *bgr, size_t bgr_pitch_bytes, float *VMap, float *NMap,
                              float *RGB, float *gx, size_t gx_pitch, float *gy, size_t gy_pitch,
                              int n, int m);
/* RG::AlignFrameGaussNewton (RG.cpp:797-932) followed by the pose algebra of display()
 * (GP-demo.cpp:201-211,225-236,244-245,192-193): fills pose_block_dev on the device.
 * history_dev (optional): f32[16] per frame -- poseR(9) poset(3) tracked iterations level_iters 0 --
 * written at slot frame_index. */
int rgbdslam_track_frame(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap,
                         const float *NMap, const float *RGB, const float *Gx, size_t gx_pitch,
                         const float *Gy, size_t gy_pitch, const float *VMap_prev,
                         const float *NMap_prev, const float *RGB_prev, int n_row, int m_col,
                         rgbdslam_pose_block *pose_block_dev, float *history_dev, int frame_index);
/* The whole row-sharded Gauss-Newton loop of a frame in ONE persistent launch per rank, the exchange inside it: the sharded
 * counterpart of rgbdslam_track_frame with rgbdslam_set_icp_loop (no reference counterpart; RG::AlignFrameGaussNewton,
 * RG.cpp:797-932, for the loop it shards).  band_lo3 / band_hi3 [lvl] = this rank's block rows of level lvl (non-empty at
 * every level); poses stay bit-identical across ranks and to the multi-launch peer form.  RGBDSLAM_E_STATE without
 * rgbdslam_set_icp_loop or a connected exchange; at most 15 Gauss-Newton steps per frame (sum of max_iter + 2). */
int rgbdslam_track_frame_sharded(rgbdslam_ctx *ctx, const rgbdslam_icp_params *prm, const float *VMap, const float *NMap,
                                 const float *RGB, const float *Gx, size_t gx_pitch, const float *Gy, size_t gy_pitch,
                                 const float *VMap_prev, const float *NMap_prev, const float *RGB_prev, int n_row, int m_col,
                                 rgbdslam_pose_block *pose_block_dev, float *history_dev, int frame_index,
                                 const int *band_lo3, const int *band_hi3);
/* PredictedFrame::Cpy + InitMask (Frame.cpp:335-370).  Runs if force != 0 or the block says lost. */
int rgbdslam_model_reseed(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, int force,
                          const float *frame_VMap, const float *frame_NMap, const float *frame_RGB,
                          const uint16_t *depth, size_t depth_pitch_bytes, float *VMap, float *NMap,
                          float *RGBD, uint8_t *Mask, int n, int m);
/* PredictedFrame::Transform + DisplayPredictedFBO + ReadFrame (without its normals) + ProjectMask +
 * the gather half of ProjectLabel (Frame.cpp:281-298,229-245,372-393; GP-demo.cpp:98-113). */
int rgbdslam_model_predict(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                           const float *VMap_in, const float *RGBD_in, const uint8_t *Mask_in,
                           const uint8_t *Label_in, uint64_t *zbuf, float *VMap_out, float *RGBD_out,
                           uint8_t *Mask_out, uint8_t *Label_out, float thresh, float znear,
                           float zfar, int n, int m,
                           /* optional (frame_VMap != NULL): when the block says LOST, the same launch
                            * performs rgbdslam_model_reseed on set "in" (VMap_in, NMap_reseed, RGBD_in,
                            * Mask_in are then written) -- GP-demo.cpp:189-193 without an extra launch */
                           const float *frame_VMap, const float *frame_NMap, const float *frame_RGB,
                           const uint16_t *depth, size_t depth_pitch_bytes, float *NMap_reseed,
                           /* optional scratch, RGBDSLAM_PREDICT_SCRATCH_BYTES(n, m) bytes, 16-byte aligned (contents
                            * need not survive the call): the z-buffer pass leaves every vertex's projection there and
                            * the resolve shades from it instead of projecting the winner's vertices again; NULL: it
                            * recomputes them.  Same bits either way. */
                           void *vertex_scratch);
#define RGBDSLAM_PREDICT_SCRATCH_BYTES(n, m) ((size_t)20 * (size_t)(n) * (size_t)(m))
/* The same, with the APPLY pass of the PREVIOUS frame's Graph::Update riding on the draw launch (a caller that enqueues a
 * recorded sequence may defer that pass by one frame: it reads the model the previous frame left -- VMap_in, RGBD_in, Mask_in,
 * Label_in -- and is posed and gated by that frame's immutable history entry, rgbdslam_track_frame's history row). */
int rgbdslam_model_predict_apply(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                                 const float *VMap_in, const float *RGBD_in, const uint8_t *Mask_in,
                                 const uint8_t *Label_in, uint64_t *zbuf, float *VMap_out, float *RGBD_out,
                                 uint8_t *Mask_out, uint8_t *Label_out, float thresh, float znear, float zfar,
                                 int n, int m, const float *frame_VMap, const float *frame_NMap,
                                 const float *frame_RGB, const uint16_t *depth, size_t depth_pitch_bytes,
                                 float *NMap_reseed, void *vertex_scratch, const float *prev_history_entry_dev, uint16_t **BumpTab,
                                 uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                 uint32_t prev_stamp, int N_prim, int M_prim, int plane_lo, int plane_hi);
/* The apply pass alone, posed and gated by a history entry (what a deferred apply pass becomes when no next frame comes). */
int rgbdslam_update_apply_from_history(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab,
                                       uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                       uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                       const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                       int plane_hi);
/* ComputeNormal of ReadFrame (Frame.cpp:242) + Segment_cu (Frame.cpp:395).  The result is left in
 * Label_swap (one pass when nbPlans == 0, three otherwise: always an odd number of swaps). */
int rgbdslam_model_normal_segment(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                                  float *NMap, const float *VMap, uint8_t *Label, uint8_t *Label_swap,
                                  const float *Equations_dev, int nbPlans, float epsilon,
                                  float alpha, int n, int m);
/* PredictedFrame::Merge (Frame.cpp:300-333): MergeDepthOff + VertexMapGL + ComputeNormal. */
int rgbdslam_model_merge(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                         const float *RGBD_in, const uint8_t *Mask_in, const uint16_t *depth,
                         size_t depth_pitch_bytes, const float *RGB, float *RGBD_out,
                         uint8_t *Mask_out, float *VMap_out, float *NMap_out, int n, int m);
/* rgbdslam_model_normal_segment followed by rgbdslam_model_merge_claim, as ONE launch instead of four: Merge reads no
 * labels and the label growth reads nothing Merge writes, so the two are independent kinds of tile of one grid; the three
 * growth passes of a tile run out of LDS on the tile + 3 rings of labels, and the claim pass of Graph::Update follows in the
 * same tile (GP-demo.cpp:232-249,260; Frame.cpp:242,300-333,395; RG.cu:906-912).  Same labels
 * (left in Label_swap, as after three passes), same model maps, same claim words as the operator sequence.  Needs
 * nbPlans > 0.  Label's contents after the call are unspecified (the operator sequence leaves the second pass there).
 * NMap_seg is scratch (3 * n * m floats, != NMap_out) that only the three-launch form of the same work touches
 * (RGBDSLAM_SEG3=0, kept for A/B measurements); the model's normals are NMap_out. */
int rgbdslam_model_segment_merge_claim(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap_seg,
                                       const float *VMap_pred, uint8_t *Label, uint8_t *Label_swap, const float *Equations_dev,
                                       int nbPlans, float epsilon, float alpha, const float *RGBD_in, const uint8_t *Mask_in,
                                       const uint16_t *depth, size_t depth_pitch_bytes, const float *RGB, float *RGBD_out,
                                       uint8_t *Mask_out, float *VMap_out, float *NMap_out, uint8_t **MaskTab, float **ParamTab,
                                       uint64_t **ClaimTab, uint32_t stamp, int N_prim, int M_prim, int n, int m, int plane_lo,
                                       int plane_hi);
/* The same, with OffLineFrame::LoadFrame (rgbdslam_frame_preprocess) of the NEXT input frame riding on the launch:
 * that job depends on nothing here and is hidden under this launch's own dependent steps.  The f* / gx / gy buffers must be
 * ones nothing of the current frame reads (the host library keeps two sets of input-frame maps for this). */
int rgbdslam_model_segment_merge_claim_preprocess(
    rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, float *NMap_seg, const float *VMap_pred, uint8_t *Label,
    uint8_t *Label_swap, const float *Equations_dev, int nbPlans, float epsilon, float alpha, const float *RGBD_in,
    const uint8_t *Mask_in, const uint16_t *depth, size_t depth_pitch_bytes, const float *RGB, float *RGBD_out, uint8_t *Mask_out,
    float *VMap_out, float *NMap_out, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, int N_prim,
    int M_prim, int n, int m, int plane_lo, int plane_hi, const uint16_t *next_depth, size_t next_depth_pitch_bytes,
    const uint8_t *next_bgr, size_t next_bgr_pitch_bytes, float *fVMap, float *fNMap, float *fRGB, float *gx, size_t gx_pitch,
    float *gy, size_t gy_pitch);
/* PredictedFrame::Merge + the CLAIM pass of the Graph::Update that follows it in display() (GP-demo.cpp:249-260): the
 * claim of a pixel needs only its own merged vertex and mask, so it rides on the merge launch.  Follow with
 * rgbdslam_update_apply_tracked (same stamp, same tables) to finish the update. */
int rgbdslam_model_merge_claim(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                               const float *RGBD_in, const uint8_t *Mask_in, const uint16_t *depth,
                               size_t depth_pitch_bytes, const float *RGB, float *RGBD_out,
                               uint8_t *Mask_out, float *VMap_out, float *NMap_out, const uint8_t *Label,
                               uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                               int N_prim, int M_prim, int n, int m, int plane_lo, int plane_hi);
int rgbdslam_update_apply_tracked(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                                  uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                                  uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD,
                                  const uint8_t *Mask, const uint8_t *Label, int N_prim, int M_prim, int n,
                                  int m, int plane_lo, int plane_hi);
/* Graph::Update (Graph.cpp:50-95) with the pose taken from the block (P_glob) and gated on it. */
int rgbdslam_update_tracked(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev,
                            uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                            uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD,
                            const uint8_t *Mask, const uint8_t *Label, int N_prim, int M_prim, int n,
                            int m, int plane_lo, int plane_hi);
/* rgbdslam_update_apply_tracked for frame f and rgbdslam_frame_preprocess for frame f+1 (same n x m) in ONE launch.
 * The two are independent once frame f's merge has run -- nothing in flight reads frame f's maps any more, so the
 * caller may pass the same frame buffers -- and both are latency-bound, so the preprocessing tiles run in the
 * issue slots the update leaves idle.  For callers that know the next frame while they enqueue this one
 * (a recorded sequence, a capture queue); Graph::Update (Graph.cpp:74-84) + OffLineFrame::LoadFrame (Frame.cpp:43-77). */
int rgbdslam_update_apply_tracked_preprocess(rgbdslam_ctx *ctx, const rgbdslam_pose_block *pose_block_dev, uint16_t **BumpTab,
                                             uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                             uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                             const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                             int plane_hi, const uint16_t *next_depth, size_t depth_pitch_bytes,
                                             const uint8_t *next_bgr, size_t bgr_pitch_bytes, float *fVMap, float *fNMap,
                                             float *fRGB, float *gx, size_t gx_pitch, float *gy, size_t gy_pitch);
/* The same, gated on and posed by the frame's entry of the pose history ring (16 floats written by
 * rgbdslam_track_frame: poseR(9, column-major) poset(3) tracked ...).  Unlike the pose block, the entry is
 * never rewritten by the next frame, so this call may run on another stream concurrently with the next
 * frame's tracking (the caller orders it after rgbdslam_model_merge and before the model is written again). */
int rgbdslam_update_from_history(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab,
                                 uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab,
                                 uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                                 const uint8_t *Label, int N_prim, int M_prim, int n, int m, int plane_lo,
                                 int plane_hi);

/* ---- stage 4: fusion ---------------------------------------------------------------------- */
/* Kernel.cuh:21  void MergeDepthOff(float *RGB_predicted, DevMem2D_<unsigned short> depth,
 *                                   float *RGB, unsigned char *mask, int n, int m) */
int rgbdslam_merge_depth_off(rgbdslam_ctx *ctx, float *RGBD_predicted, const uint16_t *depth,
                             size_t depth_pitch_bytes, const float *RGB, uint8_t *mask,
                             int n, int m);
/* Kernel.cuh:15  void VertexMapGL(float *VMap, float *RGB, int n, int m) */
int rgbdslam_vertex_map_gl(rgbdslam_ctx *ctx, float *VMap, const float *RGBD, int n, int m);
/* Primitive.cuh:23-24  void Update_cu(unsigned short **BumpTab, unsigned char **RGBTab,
 *   unsigned char **MaskTab, float **ParamTab, float *VMap, float *RGB, unsigned char *Mask,
 *   unsigned char *Label, float *pose, int N_prim, int M_prim, int n, int m)
 * Tables are DEVICE arrays of DEVICE pointers; pose is a HOST pointer (16 floats).
 * ClaimTab (device array of device u64 pointers, one N_prim*M_prim slab per plane, zeroed once)
 * and stamp (strictly increasing per call, > 0) implement the deterministic winner rule that
 * replaces the reference's data race.  Only labels in [plane_lo, plane_hi) are
 * processed (plane-partitioned multi-GPU); pass 0, INT_MAX for all. */
int rgbdslam_update(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                    float **ParamTab, uint64_t **ClaimTab, uint32_t stamp,
                    const float *VMap, const float *RGBD, const uint8_t *Mask,
                    const uint8_t *Label, const float *pose16_host, int N_prim, int M_prim,
                    int n, int m, int plane_lo, int plane_hi);
/* Primitive.cuh:18  void QuadTrim(float *VMap, unsigned char *Mask, unsigned int *indices_dev,
 *   unsigned short min_conf, unsigned int n, unsigned int m, float thresh, int lvl)
 * Quad list of a plane mesh decoded at level lvl: 4 vertex ids per quad of the (n/lvl-1) x (m/lvl-1)
 * grid, all 0 when the quad is dropped (a corner whose mask is neither 1 nor >= 10 + min_conf, or an
 * edge/diagonal longer than thresh).  variant 0 = D1/Primitive.cu:95-152; variant 1 = the DemoStep2
 * form (D2/Primitive.cu:367-424: corner rule mask < 11, ids emitted in the order 0,3,2,1). */
int rgbdslam_quad_trim(rgbdslam_ctx *ctx, const float *VMap, const uint8_t *Mask,
                       uint32_t *indices_dev, unsigned short min_conf, unsigned int n,
                       unsigned int m, float thresh, int lvl, int variant);

/* Whole-model mesh in ONE launch (lvl = 1), planes [plane_lo, plane_hi): VertexFromBump
 * (Primitive.cuh:16) + ComputeNormal on the decoded vertices (D2/Primitive.cuh:22) + QuadTrim
 * (Primitive.cuh:18), fused so that the Bump images are read once and the vertices never re-read.
 * Tables are device arrays of per-plane device pointers as in Update_cu; NMapTab[k] holds 3*n*m
 * floats, IndexTab[k] 4*(n-1)*(m-1) ids. */
int rgbdslam_bump_mesh_batch(rgbdslam_ctx *ctx, float **VMapTab, float **RGBATab, float **MaskOutTab,
                             float **NMapTab, uint32_t **IndexTab, uint16_t **BumpTab,
                             uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab, int plane_lo,
                             int plane_hi, int n, int m, unsigned short min_conf, float thresh,
                             int inverse_normals, int variant);

/* Primitive.cuh:16  void VertexFromBump(float *VMap_VBO, float *RGB_VBO, float *Mask_VBO,
 *   unsigned short *Bump_dev, unsigned char *RGB_bump_dev, unsigned char *Mask_dev,
 *   float *param_dev, int n, int m, int lvl) */
int rgbdslam_vertex_from_bump(rgbdslam_ctx *ctx, float *VMap, float *RGBA, float *MaskOut,
                              const uint16_t *Bump, const uint8_t *RGB_bump, const uint8_t *Mask,
                              const float *param_dev, int n, int m, int lvl);

/* DemoStep2's forms of the decode (D2/Primitive.cuh): a cell is empty when mask < 11.
 * variant 0 = void VertexFromBump(float *VMap_VBO, unsigned short *Bump_dev, unsigned char *RGB_bump_dev,
 *   unsigned short *Mask_dev, float *param_dev, int n, int m, int lvl) and VertexOnlyFromBump (the same without the unused
 *   colour argument), D2/Primitive.cu:31-121,704-730: Mask = uint16_t[n*m], output VMap of (n/lvl) x (m/lvl), RGBA / RGB_bump
 *   unused (may be NULL).
 * variant 1 = void VertexRGBFromBump(float *VMap_VBO, float *RGB_VBO, unsigned short *Bump_dev, unsigned char *RGB_bump_dev,
 *   unsigned char *Mask_dev, float *param_dev, int n, int m, int lvl), D2/Primitive.cu:123-179,732-742: Mask = uint8_t[n*m],
 *   VMap + RGBA (r, g, b, 1) of n x m, cell (i, j) read and written at i*m + j, positioned at (lvl*i, lvl*j).
 * The reference's kernels test i, j against n, m instead of the output size (a grid rounded up writes past the
 * sub-sampled output for lvl > 1): the output size is the bound here. */
int rgbdslam_vertex_from_bump_d2(rgbdslam_ctx *ctx, float *VMap, float *RGBA, const uint16_t *Bump,
                                 const uint8_t *RGB_bump, const void *Mask, const float *param_dev,
                                 int n, int m, int lvl, int variant);

/* The same decode for planes [plane_lo, plane_hi) in ONE launch (lvl = 1): the tables are
 * device arrays of per-plane device pointers like Update_cu's; outputs one buffer per plane (the
 * reference keeps one VBO triple per Primitive, Primitive.h:60-66).  For models of hundreds of
 * planes (BASELINE configs[2]) where one launch per plane would be launch-bound. */
int rgbdslam_vertex_from_bump_batch(rgbdslam_ctx *ctx, float **VMapTab, float **RGBATab,
                                    float **MaskOutTab, uint16_t **BumpTab, uint8_t **RGBTab,
                                    uint8_t **MaskTab, float **ParamTab, int plane_lo, int plane_hi,
                                    int n, int m);

/* ---- stage 5: prediction ------------------------------------------------------------------ */
/* Kernel.cuh:19  void TransformVertexMap(float *VMap, float *RGBD, int n, int m)  (uses pose) */
int rgbdslam_transform_vertex_map(rgbdslam_ctx *ctx, float *VMap, float *RGBD, int n, int m);
/* Kernel.cuh:20  void QuadTrimFrame(float *VMap, unsigned int *indices_dev, int n, int m, float thresh) */
int rgbdslam_quad_trim_frame(rgbdslam_ctx *ctx, const float *VMap, uint32_t *indices, int n, int m,
                             float thresh);
/* Replaces the OpenGL pass DisplayPredictedFBO (GP-demo.cpp:98-113) + PredictedFrame::Draw(true,
 * true) (Frame.cpp:247-279): z-buffered rasterisation of the quad mesh into tex = f32x4[n*m]
 * (r, g, b, depth/100; 0 = nothing drawn).  indices may be NULL: quads are then trimmed inline
 * with `thresh` exactly as quad_trim_frame would.  zbuf = u64[n*m] scratch that must hold
 * 0xFF..FF on entry (rgbdslam_raster_clear) and holds it again on return. */
int rgbdslam_raster_clear(rgbdslam_ctx *ctx, uint64_t *zbuf, int n, int m);
int rgbdslam_predict_raster(rgbdslam_ctx *ctx, float *tex, uint64_t *zbuf, const float *VMap,
                            const float *RGBD, const uint32_t *indices, float thresh,
                            float znear, float zfar, int n, int m);
/* No counterpart in the reference: rgbdslam_predict_raster for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels, same result, but the z-buffer
 * is a static device array inside the library (one copy per GPU, 8 bytes x RGBDSLAM_NOALLOC_MAX_PIXELS = 16.6 MB, held
 * until the process ends; its address on a GPU is resolved by rgbdslam_create on that GPU) instead of a caller
 * workspace.  The call itself performs no hipMalloc / hipFree / allocating library call.  n * m beyond
 * RGBDSLAM_NOALLOC_MAX_PIXELS returns RGBDSLAM_E_LIMIT.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two streams, two contexts) race -- serialise them. */
#define RGBDSLAM_NOALLOC_MAX_PIXELS (1920 * 1080)
int rgbdslam_predict_raster_noalloc(rgbdslam_ctx *ctx, float *tex, const float *VMap, const float *RGBD,
                                    const uint32_t *indices, float thresh, float znear, float zfar, int n, int m);
/* Kernel.cuh:18  void ReadFrame_cu(float *Vmap, float *RGBD, float *NMap, cudaArray *Array, int n, int m)
 * The GL texture becomes the plain f32x4 buffer written by rgbdslam_predict_raster. */
int rgbdslam_read_frame(rgbdslam_ctx *ctx, float *VMap, float *RGBD, const float *tex, int n, int m);
/* Kernel.cuh:24  void ProjectMask_cu(unsigned char *mask, unsigned char *mask_swap, float *VMap, int n, int m) */
int rgbdslam_project_mask(rgbdslam_ctx *ctx, const uint8_t *mask, uint8_t *mask_swap,
                          const float *VMap, int n, int m);
/* RG.cuh:11  void Segment_cu(unsigned char *Label, float *VMap, float *NMap, float *pose,
 *   float *Equations, int nbPlans, float epsilon, float alpha, int n, int m)
 * pose = HOST pointer; Equations = DEVICE pointer.  Label_swap = u8[n*m] scratch: the three
 * passes are double-buffered; the result is left in Label. */
int rgbdslam_segment(rgbdslam_ctx *ctx, uint8_t *Label, uint8_t *Label_swap, const float *VMap,
                     const float *NMap, const float *pose16_host, const float *Equations_dev,
                     int nbPlans, float epsilon, float alpha, int n, int m);
/* Kernel.cuh:35  void gpu_add_zbuffOff(DevMem2D_<unsigned short> depth, float *rgb_in, float *rgb, int n, int m) */
int rgbdslam_gpu_add_zbuff_off(rgbdslam_ctx *ctx, const uint16_t *depth, size_t depth_pitch_bytes,
                               const float *rgb_in, float *RGBD, int n, int m);
/* Kernel.cuh:23  void InitMask_cu(unsigned char *mask, float *RGB, int n, int m) */
int rgbdslam_init_mask(rgbdslam_ctx *ctx, uint8_t *mask, const float *RGBD, int n, int m);
/* Test-harness stand-in for plane detection (not in the reference): labels by
 * plane membership.  pose = HOST pointer. */
int rgbdslam_seed_labels(rgbdslam_ctx *ctx, uint8_t *Label, const float *VMap, const float *NMap,
                         const float *pose16_host, const float *Equations_dev, int nbPlans,
                         float epsilon, float alpha, int n, int m);

/* ---- plane detection front end: RG::DetectPrimitives, D1/RG.cpp:277-338 --------------- */
/* Kernel.cuh: void gpu_cpyN(cv::gpu::DevMem2D_<unsigned char> mat, float *NMap, int n, int m, int chnl) */
int rgbdslam_gpu_cpy_n(rgbdslam_ctx *ctx, uint8_t *mat, size_t pitch, const float *NMap, int n, int m, int chnl);
/* cv::gpu::Canny(img, edges, low, high, 3) as called at RG.cpp:296-302 (OpenCV 2.4, third party, restated from
 * its published algorithm: 3x3 Sobel, L1 magnitude, thresholds swapped if low > high, 8-connected hysteresis).
 * work_labels: n*m int32, work_bytes: 2*n*m bytes of scratch. */
int rgbdslam_canny(rgbdslam_ctx *ctx, const uint8_t *img, size_t img_pitch, uint8_t *edges, size_t edges_pitch,
                   double low_thresh, double high_thresh, int32_t *work_labels, uint8_t *work_bytes, int n, int m);
/* Kernel.cuh: void gpu_Sum(DevMem2D_<uchar> Out, InR, InG, InB, int n, int m) */
int rgbdslam_gpu_sum(rgbdslam_ctx *ctx, uint8_t *out, size_t out_pitch, const uint8_t *inR, size_t r_pitch,
                     const uint8_t *inG, size_t g_pitch, const uint8_t *inB, size_t b_pitch, int n, int m);
/* Kernel.cuh: void gpu_Discont(DevMem2D_<uchar> mat, float *VMap, int n, int m, float thresh) */
int rgbdslam_gpu_discont(rgbdslam_ctx *ctx, uint8_t *mat, size_t pitch, const float *VMap, int n, int m, float thresh);
/* Kernel.cuh: void gpu_Thresh(DevMem2D_<uchar> Out, DevMem2D_<uchar> In, unsigned char thresh, unsigned char *Label, int n, int m) */
int rgbdslam_gpu_thresh(rgbdslam_ctx *ctx, uint8_t *out, size_t out_pitch, const uint8_t *in, size_t in_pitch,
                        uint8_t thresh, const uint8_t *Label, int n, int m);
/* cv::erode(src, dst, getStructuringElement(MORPH_ELLIPSE, Size(5,5))) of RG.cpp:320-326 (CPU in the reference) */
int rgbdslam_erode_ellipse5(rgbdslam_ctx *ctx, uint8_t *dst, size_t dst_pitch, const uint8_t *src, size_t src_pitch,
                            int n, int m);
/* Utilities.h: void FindBlobs(const cv::Mat &binary, std::vector<std::vector<cv::Point2i>> &blobs) (CPU flood fill
 * in the reference).  blob[p] = raster index of the first pixel of p's 4-connected component (-1 = background):
 * ascending first-pixel index IS FindBlobs' blob order, and a blob's points are the pixels carrying its index in
 * raster order.  sizes[first pixel] = blob size, 0 elsewhere.  Both n*m int32. */
int rgbdslam_find_blobs(rgbdslam_ctx *ctx, const uint8_t *binary, size_t pitch, int32_t *blob, int32_t *sizes,
                        int n, int m);
/* RG.cpp:293-338 in one call (3 x (gpu_cpyN + Canny) + gpu_Sum + gpu_Discont + gpu_Thresh + erode + FindBlobs) without
 * the intermediate images.  Label may be NULL (the reference clears it just before, RG.cpp:313).
 * work_labels: 3*n*m int32, work_bytes: 8*n*m bytes of scratch. */
int rgbdslam_detect_regions(rgbdslam_ctx *ctx, const float *NMap, const float *VMap, const uint8_t *Label,
                            double canny_low, double canny_high, float thresh_discont, uint8_t thresh_edge,
                            int32_t *blob, int32_t *sizes, int32_t *work_labels, uint8_t *work_bytes, int n, int m);

/* The per-blob median plane fit of RG::DetectPrimitives (RG.cpp:469-646: three sorted normal-component lists and one sorted
 * distance list per blob, their middle elements, the in-plane basis, the blob centre) on the device: the middle element of a
 * sorted list is found by a three-pass radix select over the image, so the floats are the ones `sort` + `[size / 2]` returns.
 * blob / sizes: rgbdslam_find_blobs' / rgbdslam_detect_regions' outputs; VMap / NMap: the model frame; pose: HOST pointer,
 * column-major 4x4.  workspace: rgbdslam_fit_planes_workspace_bytes(n, m) bytes of device memory, 256-byte aligned, owned by
 * the caller; everything is enqueued on the context's stream.
 * rgbdslam_fit_planes_read (BLOCKING): *nb_blobs = blobs of more than min_size pixels (RG.cpp:342-347); when it exceeds
 * RGBDSLAM_FIT_MAX_BLOBS the records are incomplete and the caller fits on the host.  records_host: RGBDSLAM_FIT_MAX_BLOBS x
 * RGBDSLAM_FIT_RECORD floats, in NO particular blob order: [0] root pixel (int bits) = FindBlobs' blob order, [1] size (int), [2]
 * non-zero normals (int), [3] accepted (int: [2] >= min_normals, RG.cpp:500), [4..13] equation n(3) e1(3) e2(3) d, [17..19] centre.
 * rgbdslam_fit_planes_label: Label[p] = label_of_slot_host[record index of p's blob], 0 elsewhere (RG.cpp:590). */
#define RGBDSLAM_FIT_MAX_BLOBS 64
#define RGBDSLAM_FIT_RECORD 20
size_t rgbdslam_fit_planes_workspace_bytes(int n, int m);
int rgbdslam_fit_planes(rgbdslam_ctx *ctx, const int32_t *blob, const int32_t *sizes, const float *VMap, const float *NMap,
                        const float *pose16_host, int min_size, int min_normals, void *workspace, int n, int m);
int rgbdslam_fit_planes_read(rgbdslam_ctx *ctx, const void *workspace, int *nb_blobs, float *records_host);
int rgbdslam_fit_planes_label(rgbdslam_ctx *ctx, uint8_t *Label, const int32_t *blob, void *workspace,
                              const uint8_t *label_of_slot_host, int n, int m);

/* ---- prediction of the model directly from the Bump images (BASELINE configs[2] "fusion +
 * raycast partitioned by plane").  No call site in DemoStep1's loop (it predicts from the previous depth image); the meshes are
 * the reference's: VertexFromBump (Primitive.cuh:16, Primitive.cu:31-93) + QuadTrim (Primitive.cuh:18, Primitive.cu:95-152).
 * planes_draw rasterises the trimmed quad meshes of planes [plane_lo, plane_hi) -- transformed by the world -> camera pose
 * (HOST pointer, 16 floats column-major) -- into zbuf (u64[n*m], 0xFF-filled by rgbdslam_raster_clear; NOT cleared here) with
 * the rasteriser rules of rgbdslam_predict_raster; key = f32 depth << 32 | ~(((plane*(N-1)*(M-1) + quad)*2 + t)).  Z-buffers of
 * disjoint plane ranges merge by an elementwise u64 minimum (multi-GPU: one all-reduce).  planes_resolve turns the pixels won by
 * planes [plane_lo, plane_hi) into VMap (f32x3) and RGBD (r, g, b, depth/100) as ReadFrame_cu does (Kernel.cu:309-343); empty
 * pixels become zeros (written when plane_lo == 0); other pixels are left alone.  min_conf >= 1 (Primitive.cu:135). */
int rgbdslam_planes_draw(rgbdslam_ctx *ctx, uint64_t *zbuf, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab,
                         float **ParamTab, int plane_lo, int plane_hi, const float *pose_world_to_cam16_host, int N_prim,
                         int M_prim, unsigned short min_conf, float thresh, float znear, float zfar, int n, int m);
int rgbdslam_planes_resolve(rgbdslam_ctx *ctx, uint64_t *zbuf, float *VMap, float *RGBD, uint16_t **BumpTab, uint8_t **RGBTab,
                            uint8_t **MaskTab, float **ParamTab, int plane_lo, int plane_hi,
                            const float *pose_world_to_cam16_host, int N_prim, int M_prim, int reset_zbuf, int n, int m);

/* ---- 16-bit labels: models of more than 255 planes (BASELINE configs[2], 256 planes).  No reference counterpart: the
 * reference's label image is u8 (Frame.h:197; label 0 = none, L = plane L-1, Primitive.cu:279-281), which caps a model at 255
 * planes.  rgbdslam_update / rgbdslam_update_from_history with a u16 label image; same kernels, same winner rule. */
int rgbdslam_update_l16(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                        uint64_t **ClaimTab, uint32_t stamp, const float *VMap, const float *RGBD, const uint8_t *Mask,
                        const uint16_t *Label16, const float *pose16_host, int N_prim, int M_prim, int n, int m, int plane_lo,
                        int plane_hi);
int rgbdslam_update_from_history_l16(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab, uint8_t **RGBTab,
                                     uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, const float *VMap,
                                     const float *RGBD, const uint8_t *Mask, const uint16_t *Label16, int N_prim, int M_prim,
                                     int n, int m, int plane_lo, int plane_hi);
/* Test / bench harness stand-in for plane detection on TILED models (like rgbdslam_seed_labels, not in the reference): a
 * pixel takes the first plane k in [0, nb_planes) whose equation it satisfies (|n.p - d| <= epsilon, n.n_p >= alpha) and whose
 * N_prim x M_prim Bump image contains its cell.  Pose: pose16_host (HOST pointer) or, when that is NULL, the device-resident
 * history entry poseR(9, column-major) poset(3) ... written by rgbdslam_track_frame. */
int rgbdslam_label_planes_l16(rgbdslam_ctx *ctx, uint16_t *Label16, const float *VMap, const float *NMap,
                              const float *pose16_host, const float *history_entry_dev, float **ParamTab, int nb_planes,
                              int N_prim, int M_prim, float epsilon, float alpha, int n, int m);
/* 16-bit labels through the TRACKER (round 3; no reference counterpart: PredictedFrame::_Label_dev is unsigned char, Frame.h:197,
 * and so are the label arguments of ProjectMask_cu / Segment_cu / Update_cu).  rgbdslam_set_label_width(ctx, 2) makes every
 * label image this context's entry points read or write a uint16_t image of n*m elements; the pointers keep their uint8_t type
 * in the signatures (the u8 entry points stay the boundary, 1 = default).  It applies to: rgbdslam_model_predict[_apply]
 * (Label_in / Label_out), rgbdslam_model_normal_segment, rgbdslam_model_segment_merge_claim[_preprocess],
 * rgbdslam_model_merge_claim, rgbdslam_update / _update_tracked / _update_apply_tracked[_preprocess] / _update_from_history /
 * _update_apply_from_history / _update_cu, rgbdslam_segment / _segment_cu, rgbdslam_seed_labels, rgbdslam_project_label,
 * rgbdslam_fit_planes_label, rgbdslam_gpu_thresh / _detect_regions (Label argument), rgbdslam_vote_plan / _affect_plan;
 * nbPlans may then reach 65535.  The *_l16 entry points above take u16 labels whatever the width.  Same kernels (templates on
 * the label type), same arithmetic: a model of at most 255 planes gives the same labels at either width. */
int rgbdslam_set_label_width(rgbdslam_ctx *ctx, int bytes_per_label);
/* One POD for what the reference spreads over compile-time #defines and the RG constructor's arguments (cudaTypes.cuh:41-51,
 * Utilities.h:75-85, GP-demo.cpp:326-451: RG(n, m, path, Calib[11], lvl, distThres, angleThres, ...); RG.h:111,124-126) plus this
 * library's own switches.  rgbdslam_config_defaults fills the reference's TUM fr1/xyz values (GP-demo.cpp:404-409);
 * rgbdslam_configure applies calib / switches to the context (the ICP parameters are per call: pass cfg->icp to
 * rgbdslam_track_frame / rgbdslam_align_gauss_newton); rgbdslam_get_config reads the context's current values back (icp is left
 * untouched).  The individual setters remain. */
typedef struct rgbdslam_config {
    float calib[11];               /* fx fy cx cy k1..k5 ds factor (GP-demo.cpp:17,404) */
    rgbdslam_icp_params icp;       /* lvl 3, max_iter {3,2,1}, distThres 0.05, angleThres sin 20 deg, thres_cvg 1e-6 */
    int icp_loop;                  /* rgbdslam_set_icp_loop */
    int icp_line_check;            /* rgbdslam_set_icp_line_check */
    int label_width;               /* rgbdslam_set_label_width: 1 (Frame.h:197) or 2 */
} rgbdslam_config;
void rgbdslam_config_defaults(rgbdslam_config *cfg);
int rgbdslam_configure(rgbdslam_ctx *ctx, const rgbdslam_config *cfg);
int rgbdslam_get_config(const rgbdslam_ctx *ctx, rgbdslam_config *cfg);
int rgbdslam_get_label_width(const rgbdslam_ctx *ctx);
/* Kernel.cuh:24 ProjectMask_cu applied to the LABEL image (Frame.cpp:389, the gather of PredictedFrame::ProjectLabel):
 * rgbdslam_project_mask with the context's label width (uses the context's pose, like rgbdslam_project_mask). */
int rgbdslam_project_label(rgbdslam_ctx *ctx, const uint8_t *Label, uint8_t *Label_swap, const float *VMap, int n, int m);

/* Thin frame pack for the plane-partitioned multi-GPU form (no reference counterpart: the reference is single-GPU; SURVEY.md
 * 8e row 1).  After PredictedFrame::Merge (Frame.cpp:300-333) the model VMap is VertexMapGL's function of RGBD.w
 * (Kernel.cu:54-72) and RGBD.rgb is the frame's colour byte / 255 (Kernel.cu:463-465, 739-753), so the plane owners only need
 * W = RGBD.w (f32), the colour bytes, Mask and the labels: 10 B per pixel instead of 31.  rgbdslam_export_thin writes those
 * fields of a merged model frame; rgbdslam_update_from_history_thin_l16 is rgbdslam_update_from_history_l16 reading them
 * (vertex and colour rebuilt with the reference's operations: the planes receive the same bits). */
int rgbdslam_export_thin(rgbdslam_ctx *ctx, const float *RGBD, const uint8_t *Mask, float *W_out, uint8_t *RGB8_out,
                         uint8_t *Mask_out, int n, int m);
/* rgbdslam_export_thin + the frame's u16 label image (as the tracker maintains it at label width 2) and its 64-byte history
 * entry, in the same launch; either pair (Label16, Label16_out) / (history_entry_dev, history_entry_out) may be NULL */
int rgbdslam_export_thin_l16(rgbdslam_ctx *ctx, const float *RGBD, const uint8_t *Mask, const uint16_t *Label16,
                             const float *history_entry_dev, float *W_out, uint8_t *RGB8_out, uint8_t *Mask_out,
                             uint16_t *Label16_out, float *history_entry_out, int n, int m);
int rgbdslam_update_from_history_thin_l16(rgbdslam_ctx *ctx, const float *history_entry_dev, uint16_t **BumpTab, uint8_t **RGBTab,
                                          uint8_t **MaskTab, float **ParamTab, uint64_t **ClaimTab, uint32_t stamp, const float *W,
                                          const uint8_t *RGB8, const uint8_t *Mask, const uint16_t *Label16, int N_prim, int M_prim,
                                          int n, int m, int plane_lo, int plane_hi);

/* ---- the rest of the reference's operator headers: everything else its unchanged host code calls ---------------------
 * (so that integration/rgbdslam_shim.cpp defines every symbol Frame.cpp / Graph.cpp / RG.cpp reference) */
/* Kernel.cuh:22  void MergeDepthKinect(float *RGB_predicted, unsigned short *depth, float *RGB, unsigned char *mask, int n, int m)
 * (Kernel.cu:528-614; live-Kinect layout: depth[(n-1-i)*m + (m-1-j)], no pitch; Frame.cpp:315) */
int rgbdslam_merge_depth_kinect(rgbdslam_ctx *ctx, float *RGBD_predicted, const uint16_t *depth, const float *RGB,
                                uint8_t *mask, int n, int m);
/* Kernel.cuh:36  void gpu_add_zbuffKinect(unsigned short *depth, float *rgb_in, float *rgb, int n, int m)  (Kernel.cu:804-823; Frame.cpp:353) */
int rgbdslam_gpu_add_zbuff_kinect(rgbdslam_ctx *ctx, const uint16_t *depth, const float *rgb_in, float *RGBD, int n, int m);
/* Kernel.cuh:28  void gpu_cpy(cv::gpu::DevMem2D_<float> mat, float *rgb, int n, int m)  (Kernel.cu:670-682; RG.cpp:807) on its own,
 * for callers that keep cv::gpu::Sobel; rgbdslam_intensity_sobel is the fused form */
int rgbdslam_gpu_cpy(rgbdslam_ctx *ctx, float *img, size_t img_pitch_bytes, const float *rgb, int n, int m);
/* Kernel.cuh:16  void MapData2VBO_cu(BYTE *d_colorFrame, USHORT *d_depthFrame, LONG *d_colorCoord, float *VMap, float *RGB, int n, int m)
 * (Kernel.cu:74-121; Frame.cpp:103; BGRA8 colour, LONG = int32).  VMap / RGB are only written where the colour coordinate is valid. */
int rgbdslam_map_data_2_vbo(rgbdslam_ctx *ctx, const uint8_t *colorFrame, const uint16_t *depthFrame, const int32_t *colorCoord,
                            float *VMap, float *RGB, int n, int m);
/* RG.cuh:23  void VotePlan_cu(unsigned char *Label, float *VMap, float *centers, float *count, float *pose, int n, int m)
 * (RG.cu:508-537,999-1014; RG.cpp:946).  centers (3*nbPlans) and count (nbPlans) are DEVICE buffers that are added to; pose is
 * a HOST pointer.  The reference's unordered float atomics are replaced by a fixed summation order. */
int rgbdslam_vote_plan(rgbdslam_ctx *ctx, const uint8_t *Label, const float *VMap, float *centers_dev, float *count_dev,
                       const float *pose16_host, int nbPlans, int n, int m);
/* RG.cuh:25  void AffectPlan_cu(unsigned char *Label, int *Buff, int n, int m)  (RG.cu:539-555,1016-1025; RG.cpp:965) */
int rgbdslam_affect_plan(rgbdslam_ctx *ctx, uint8_t *Label, const int32_t *Buff_dev, int n, int m);
/* Primitive.cuh:32  float Overlap_cu(unsigned short *Bump, unsigned char *Mask, float *param, float *param_in, int n, int m, int N_prim, int M_prim)
 * (Primitive.cu:502-556; Graph.cpp:341).  BLOCKING.  counters2 (optional, host): [0] cells landing on the other plane, [1] occupied cells. */
int rgbdslam_overlap(rgbdslam_ctx *ctx, const uint16_t *Bump, const uint8_t *Mask, const float *param_dev, const float *param_in_dev,
                     int n, int m, int N_prim, int M_prim, float *ratio, uint32_t *counters2);
/* RG.cuh:27-29  void Affect(float **Tab, float *in, int indx) / AffectChar / AffectShort  (RG.cu:557-567,1027-1049; Graph.cpp:64-68):
 * Tab[indx] = ptr in a device pointer table.  BLOCKING, like the reference's one-thread kernels. */
int rgbdslam_table_store(rgbdslam_ctx *ctx, void **tab_dev, const void *ptr, int indx);
/* Primitive.cuh:23-24  Update_cu with exactly the reference's arguments: rgbdslam_update with the claim slabs (one u64 per cell and
 * plane, allocated on first use and kept in the context) and the call stamp owned by the library.  nb_planes = valid table entries. */
#define RGBDSLAM_COMPAT_MAX_PLANES 255
int rgbdslam_update_cu(rgbdslam_ctx *ctx, uint16_t **BumpTab, uint8_t **RGBTab, uint8_t **MaskTab, float **ParamTab,
                       const float *VMap, const float *RGBD, const uint8_t *Mask, const uint8_t *Label,
                       const float *pose16_host, int N_prim, int M_prim, int n, int m, int nb_planes);
/* RG.cuh:11  Segment_cu with exactly the reference's arguments: rgbdslam_segment with the swap image owned by the library */
int rgbdslam_segment_cu(rgbdslam_ctx *ctx, uint8_t *Label, const float *VMap, const float *NMap, const float *pose16_host,
                        const float *Equations_dev, int nbPlans, float epsilon, float alpha, int n, int m);

#ifdef __cplusplus
}
#endif
#endif
