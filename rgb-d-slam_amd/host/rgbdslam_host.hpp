// rgbdslam_host.hpp -- C++ host mirror of the reference's L3/L4 classes for the hot path:
//   InputFrame / OffLineFrame / PredictedFrame   (D1/Frame.h, D1/Frame.cpp)
//   Primitive                                    (D1/Primitive.h)
//   Graph                                        (D1/Graph.h, D1/Graph.cpp)
//   RG                                           (D1/RG.h, D1/RG.cpp)  + the display() loop of D1/GP-demo.cpp
// Same entry-point names and argument meaning; all device work goes through the C ABI of
// include/rgbdslam.h (no kernels here).  GL/VBO interop, file I/O and the Kinect path are dropped
// (out of scope).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdexcept>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/rgbdslam.h"
#include "dataset_io.hpp"
#include "model_io.hpp"

namespace slam {

struct Error : std::runtime_error {
    int code;
    Error(const std::string &what, int c) : std::runtime_error(what + " (code " + std::to_string(c) + ")"), code(c) {}
};
void check(int rc, const char *what);
void hip_check(hipError_t e, const char *what);

// 3x3 float, column-major like Eigen::Matrix3f (D1/RG.h:55-58); arithmetic restates the Eigen
// expressions the reference uses (products left to right, cofactor inverse).
struct Mat3 {
    float a[9];
    float &operator()(int r, int c) { return a[c * 3 + r]; }
    float operator()(int r, int c) const { return a[c * 3 + r]; }
    static Mat3 Identity();
    Mat3 operator*(const Mat3 &b) const;
    Mat3 inverse() const;
    Mat3 neg() const;
};
struct Vec3 {
    float v[3];
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    static Vec3 Zero() { return Vec3{{0.f, 0.f, 0.f}}; }
    Vec3 operator+(const Vec3 &o) const { return Vec3{{v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}}; }
    float norm() const;
};
Vec3 operator*(const Mat3 &m, const Vec3 &x);
float norm_minus_identity(const Mat3 &m);
// D1/Utilities.cpp:1145-1167
void quaternion2matrix(double *q, double m[3][3]);
// 16-float column-major packing used throughout display() (D1/GP-demo.cpp:206-211)
void pack_pose(const Mat3 &R, const Vec3 &t, float *pose16);
// Stand-ins for Eigen's 6x6 determinant and LLT (D1/RG.cpp:862,873), parity unpinned.
float det6(const float *A);
bool llt_solve6(const float *A, const float *b, float *x);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    void alloc(size_t n, bool zero = true);
    void release();
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// ---- D1/Frame.h:19-134 ------------------------------------------------------------------------
// device pointer tables + stamp of one Graph::Update (Graph.cpp:56-70), handed to the fused merge+claim / apply pair
struct UpdateTables {
    uint16_t **Bump; uint8_t **RGB, **Mask; float **Param; uint64_t **Claim;
    uint32_t stamp;
    int N_prim, M_prim, plane_lo, plane_hi;
};

class InputFrame {
public:
    InputFrame(rgbdslam_ctx *ctx, int n, int m);
    virtual ~InputFrame() {}
    int getN() const { return _n; }
    int getM() const { return _m; }
    float *getVMap() { return _VMap_dev[_cur].p; }
    float *getNMap() { return _NMap_dev[_cur].p; }
    float *getRGB() { return _RGB_dev[_cur].p; }
    bool _offline = true;
protected:
    rgbdslam_ctx *_ctx;
    int _n, _m;
    // two buffer sets: with a load stream attached (fused pipeline) the next frame is preprocessed into the
    // other set while the current one is still being tracked and merged; otherwise only set 0 is used
    int _cur = 0;
    DevBuf<float> _VMap_dev[2], _NMap_dev[2], _RGB_dev[2];
};

class OffLineFrame : public InputFrame {
public:
    OffLineFrame(rgbdslam_ctx *ctx, int n, int m);
    ~OffLineFrame();
    // D1/Frame.cpp:43-77 with the cv::imread + GpuMat::upload pair replaced by the caller's
    // decoded images: host buffers (uploaded here) or device-resident ones (used in place).
    void LoadFrame(const uint16_t *depth_host, const uint8_t *bgr_host);
    void LoadFrame(const std::string &filename_depth, const std::string &filename_color);   // Frame.cpp:43-77 (image files)
    void LoadFrameDevice(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch);
    // The maps of the frame (depth_dev, bgr_dev) were already computed into this object's buffers by a launch the caller
    // enqueued (rgbdslam_update_apply_tracked_preprocess): the LoadFrameDevice of exactly that frame launches nothing.
    // other_set: they were computed into the OTHER buffer set (other_*(), below), which that LoadFrameDevice switches to.
    void MarkPreprocessed(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch, bool other_set = false);
    // The buffer set the current frame does NOT use (allocated on first use; no load stream attached): a launch that still
    // reads the current frame's maps can preprocess the next frame into it (rgbdslam_model_segment_merge_claim_preprocess).
    void PrepareOtherSet();
    float *otherVMap() { return _VMap_dev[_cur ^ 1].p; }
    float *otherNMap() { return _NMap_dev[_cur ^ 1].p; }
    float *otherRGB() { return _RGB_dev[_cur ^ 1].p; }
    float *otherGx() { return _Gx[_cur ^ 1].p; }
    float *otherGy() { return _Gy[_cur ^ 1].p; }
    // fused pipeline: with gradients enabled, LoadFrame* runs rgbdslam_frame_preprocess (one kernel for
    // Frame.cpp:65-71 + RG.cpp:807-810) and the tracker skips its own Sobel pass
    void EnableGradients(bool on);
    // OPTIONAL stage, off by default: cv::bilateralFilter(depth, 5, 100.0, 100.0) before everything else -- the lines the
    // reference carries commented out (Frame.cpp:50-54).  The filtered image replaces the depth for the whole frame (Merge too).
    void EnableBilateral(bool on) { _bilateral = on; }
    bool bilateral() const { return _bilateral; }
    bool has_gradients() const { return _gradients; }
    float *getGx() { return _Gx[_cur].p; }
    float *getGy() { return _Gy[_cur].p; }
    // Auxiliary ICP records of a buffer set (rgbdslam_set_frame_aux: gradients / 8, intensity, depth weight per pixel, written by
    // the fused preprocessing next to the maps, read by the persistent ICP kernel).  announce_aux: call right before the launch
    // that preprocesses into the current (other = false) or the other buffer set; getAux: the current set's records, or NULL when
    // its maps were not produced that way.  RGBDSLAM_NO_AUX=1 switches the records off (A/B).
    void announce_aux(bool other);
    const float *getAux() const { return _aux_valid[_cur] ? _Aux[_cur].p : nullptr; }
    const uint16_t *depth() const { return _depth_cur[_cur]; }
    size_t depth_pitch() const { return _depth_pitch[_cur]; }
    // Overlap: loads go to `load_stream` and alternate between the two buffer sets.  wait_ready(consumer) = the current
    // set is preprocessed (the consumer stream waits on it); release(consumer) = the consumer's work enqueued
    // so far is the last that reads the current set (the load that next reuses the set waits on it).
    // Ordering primitive: stream value operations by default, events on request or when signal memory is refused; see
    // set_stream_events below for the profiler hang the default can run into.
    void AttachLoadStream(hipStream_t load_stream);
    hipStream_t load_stream() const { return _load_stream; }
    void wait_ready(hipStream_t consumer);                         // the consumer stream waits for the current set's maps
    void release(hipStream_t consumer);
private:
    void begin_load();
    void preprocess(const uint8_t *bgr, size_t bgr_pitch);
    DevBuf<uint16_t> _depth_dev_test[2];
    DevBuf<uint8_t> _color_dev_test[2];
    DevBuf<float> _Gx[2], _Gy[2], _Aux[2];
    bool _aux_valid[2] = {false, false};
    bool _aux_enabled = getenv("RGBDSLAM_NO_AUX") == nullptr;
    const uint16_t *_depth_cur[2] = {nullptr, nullptr};
    size_t _depth_pitch[2] = {0, 0};
    bool _gradients = false, _bilateral = false;
    DevBuf<uint16_t> _depth_filtered[2];
    void filter_depth();
    hipStream_t _load_stream = nullptr;
    // The two streams are ordered by stream VALUE operations on two words of signal memory per buffer set (ready: written
    // by the load stream, awaited by the consumer; free: the reverse), sequence numbers counting up: measured on this
    // system (tools/microbench/stream_sync.hip) an event record + wait costs ~16 us per hand-over, a value write + wait ~6.
    // RGBDSLAM_STREAM_EVENTS=1, RG::set_stream_events(true) / slamhost_rg_set_stream_events select events, and so does a refused
    // signal-memory allocation.  CAUTION: a profiler that collects hardware counters (rocprofv3 --pmc) serialises the
    // queues; the producer stream then never runs while the consumer sits in hipStreamWaitValue32, and the process hangs --
    // profile load-stream applications with events.
    bool _use_events = getenv("RGBDSLAM_STREAM_EVENTS") != nullptr;
public:
    void set_stream_events(bool on) {
        if (_load_stream) throw Error("set_stream_events: the load stream is already attached", RGBDSLAM_E_STATE);
        _use_events = on;
    }
    bool stream_events() const { return _use_events; }
private:
    hipEvent_t _ready[2] = {nullptr, nullptr}, _free[2] = {nullptr, nullptr};
    uint32_t *_sig_ready[2] = {nullptr, nullptr}, *_sig_free[2] = {nullptr, nullptr};
    uint32_t _seq_ready[2] = {0, 0}, _seq_free[2] = {0, 0};
    bool _free_recorded[2] = {false, false};
    bool _loaded_once = false;
    struct { const uint16_t *depth = nullptr; size_t depth_pitch = 0; const uint8_t *bgr = nullptr; size_t bgr_pitch = 0; bool valid = false, other = false; } _pre;
};

// ---- D1/Frame.h:188-282 -----------------------------------------------------------------------
class PredictedFrame {
public:
    PredictedFrame(rgbdslam_ctx *ctx, int n, int m);
    void SetLabelWidth(int bytes);                                // 1 = unsigned char (Frame.h:197), 2 = u16 (> 255 planes)
    int label_width() const { return _label_width; }
    void Transform(float *pose, bool quad);                       // Frame.cpp:281-298
    void ReadFrame(float znear, float zfar);                      // GP-demo.cpp:98-113 + Frame.cpp:229-245
    void Merge(OffLineFrame *frame);                              // Frame.cpp:300-333
    void Cpy(OffLineFrame *frame);                                // Frame.cpp:335-359
    void InitMask();                                              // Frame.cpp:361-370
    void ProjectMask(float *pose);                                // Frame.cpp:372-383
    void ProjectLabel(float *pose_loc, float *pose_glob, float *Equations_dev, int nbPlans);  // Frame.cpp:385-405
    // fused frame pipeline (include/rgbdslam.h "fused frame pipeline"): the same methods as one
    // launch each, poses taken from the device-resident block, model set A -> B -> A
    void ReseedFused(const rgbdslam_pose_block *pb, int force, OffLineFrame *frame);       // Cpy + InitMask
    // Transform..ProjectLabel gather; prev_apply != nullptr: the previous frame's deferred apply pass rides on the draw launch
    void PredictFused(const rgbdslam_pose_block *pb, float znear, float zfar, OffLineFrame *reseed_from,
                      const UpdateTables *prev_apply = nullptr, const float *prev_history_entry = nullptr);
    void NormalSegmentFused(const rgbdslam_pose_block *pb, float *Equations_dev, int nbPlans);
    // claim != null: the claim pass of the Graph::Update that follows rides on the merge launch
    // NormalSegmentFused + MergeFused(claim) in one launch (rgbdslam_model_segment_merge_claim); needs planes
    // next_depth != nullptr: the preprocessing of that next device frame rides on the launch (into frame's other buffer set)
    void SegmentMergeClaimFused(const rgbdslam_pose_block *pb, float *Equations_dev, int nbPlans, OffLineFrame *frame, const UpdateTables &claim,
                                const uint16_t *next_depth = nullptr, size_t next_depth_pitch = 0, const uint8_t *next_bgr = nullptr,
                                size_t next_bgr_pitch = 0);
    void MergeFused(const rgbdslam_pose_block *pb, OffLineFrame *frame, const struct UpdateTables *claim = nullptr);
    float *getVMap() { return _VMap_dev.p; }
    float *getNMap() { return _NMap_dev.p; }
    float *getRGB() { return _RGB_dev.p; }
    int _n, _m;
    uint8_t *_Mask_dev, *_Mask_dev_swap, *_Label_dev, *_Label_dev_swap;   // labels: label_width() bytes per pixel
    bool use_index_buffer = false;   // true: QuadTrimFrame writes the GL_QUADS index buffer first
private:
    rgbdslam_ctx *_ctx;
    int _label_width = 1;
    DevBuf<float> _VMap_dev, _NMap_dev, _RGB_dev, _tex;
    DevBuf<float> _VMap_alt, _RGB_alt;   // model set B of the fused pipeline
    DevBuf<float> _NMap_seg;             // scratch of SegmentMergeClaimFused's three-launch form (RGBDSLAM_SEG3=0)
    DevBuf<uint8_t> _mask_a, _mask_b, _label_a, _label_b;
    DevBuf<uint32_t> _Index_dev;
    DevBuf<uint64_t> _zbuf;
    DevBuf<float> _vertex_scratch;       // rgbdslam_model_predict's optional scratch (RGBDSLAM_PREDICT_SCRATCH_BYTES)
};

// ---- D1/Primitive.h:17-237 (data model of one Bump image) ---------------------------------------
#define RES_PLANE 0.004
class Primitive {
public:
    Primitive(float *nmle, float *e1, float *e2, float dist, float *scale, int *Size, float *Shift, float *anchor);
    void Allocate();
    void SetParam();                                              // Primitive.h:180-190
    float _nmle[3], _e1[3], _e2[3], _dist, _scale[2], _Shift[2], _anchor[3];
    int _Size[2];
    uint8_t *_Mask_dev = nullptr;
    uint16_t *_Bump_dev = nullptr;
    uint8_t *_RGB_dev = nullptr;
    uint64_t *_Claim_dev = nullptr;
    DevBuf<float> _param_dev;
};

// ---- D1/Graph.h:17-128 ----------------------------------------------------------------------------
class Graph {
public:
    Graph(rgbdslam_ctx *ctx, int frag_n, int frag_m, int max_planes);
    ~Graph();
    bool AddPrimitive(Primitive *Prim);                           // Graph.h:100-116
    void Update(PredictedFrame *frame, float *pose);              // Graph.cpp:50-95
    void UpdateTracked(PredictedFrame *frame, const rgbdslam_pose_block *pb);   // same, pose from the device block
    void UpdateFromHistory(PredictedFrame *frame, const float *history_entry_dev);   // same, pose from the frame's history entry
    // split form: begin_update (tables + new stamp; false without planes), claim pass inside PredictedFrame::MergeFused,
    // then ApplyTracked
    bool begin_update(UpdateTables *t);
    void ApplyTracked(PredictedFrame *frame, const rgbdslam_pose_block *pb, const UpdateTables &t);
    // ApplyTracked + the preprocessing of the NEXT frame into `next`'s buffers, one launch
    void ApplyTrackedAndPreprocess(PredictedFrame *frame, const rgbdslam_pose_block *pb, const UpdateTables &t, OffLineFrame *next,
                                   const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch);
    void Clear();
    // keeps primitive i as number Assoc[i] - 1 (Assoc[i] = 0: dropped and freed); device tables rebuilt on the next update
    void Compact(const std::vector<int32_t> &Assoc);
    std::vector<Primitive *> _nodes;
    int _SizeVBUFF[2];
    int plane_lo = 0, plane_hi = 0x7FFFFFFF;                       // plane-partitioned multi-GPU shard
private:
    rgbdslam_ctx *_ctx;
    int _max_planes;
    std::vector<DevBuf<uint16_t> *> _data_Bump_dev;
    std::vector<DevBuf<uint8_t> *> _data_RGB_dev, _data_Mask_dev;
    std::vector<DevBuf<uint64_t> *> _data_Claim_dev;
    DevBuf<void *> _BumpTab, _RGBTab, _MaskTab, _ParamTab, _ClaimTab;
    bool _tables_built = false;
    void build_tables();
    uint32_t _stamp = 0;
};

// ---- D1/RG.h:29-330 ---------------------------------------------------------------------------------
class RG {
public:
    // n = height, m = width (RG.h:110-111)
    RG(int device, void *stream, int n, int m, const float *Calib, int lvl, float distThres, float angleThres,
       int max_planes = 50, int N_prim = 1024, int M_prim = 1024);
    ~RG();
    bool LoadFrame(const uint16_t *depth_host, const uint8_t *bgr_host);   // RG.cpp:78-147 minus file I/O
    bool LoadFrameDevice(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch);
    // The device-resident frame that the NEXT LoadFrameDevice will name, if the caller knows it already (a recorded
    // sequence): the fused pipeline then computes its maps in the launch of this frame's Bump update.
    void HintNextDeviceFrame(const uint16_t *depth_dev, size_t depth_pitch, const uint8_t *bgr_dev, size_t bgr_pitch);
    // file sources: the dataset part of the constructor (RG.h:145-156), RG::LoadFrame() as in
    // RG.cpp:78-147 (false once the sequence is exhausted), and the TUM trajectory of the frames processed so far
    void OpenDataset(const char *path, int Sizedata, bool real = false, bool RGBDRep = true);
    bool LoadFrame();
    void SaveTrajectory(const char *filename);                     // Utilities.cpp:500-546 over the recorded poses
    // model dump: the saving half of RG::Finish + SaveThread (RG.cpp:16-59,209-232) -- planes are
    // copied to the host (Primitive::LoadOnCPU), cleaned, and written with the key frame; the live model is untouched
    void SaveModel(const char *dir, int indx);
    Dataset *_dataset = nullptr;
    bool _first_out = true;
    bool AlignFrameGaussNewton();                                  // RG.cpp:797-932
    // plane injection: stands in for DetectPrimitives then runs CreatePrimitives
    void SetPlanes(const float *equations, const float *centers, int nbPlan);
    // The reference's TIMING switch (Utilities.h:82; clock() round every stage of display(), GP-demo.cpp:153-277) for an
    // asynchronous pipeline: HIP events at the stage boundaries on the frame stream, read back on demand.  Costs an event
    // record per boundary (~2 us each), hence opt-in.  StageTimes: mean us per frame of the frames since the last read (at
    // most the last 64): [0] stage 1 preprocessing (0 when it rode on the previous frame's merge launch), [1] stages 2+3 ICP,
    // [2] stage 5 prediction, [3] stage 4 fusion (Segment + Merge + Graph::Update and whatever rides on it); returns the frames.
    void SetStageTiming(bool on);
    int StageTimes(double us[4]);
    void SetLabel16(bool on);                                     // u16 label images (models of more than 255 planes)
    // tracker of a model whose Bump images live elsewhere (plane-partitioned form): equations for ProjectLabel + Segment only
    void SetPlaneEquations(const float *equations, int nbPlan);
    void SeedLabelsTiled(float **ParamTab_dev, int nbPlan, int N_prim, int M_prim);   // detection stand-in for tiled models (u16)
    void ExportLabels(void *dst_dev, size_t off_label);
    void CreatePrimitives(float *pose, bool labels_from_detection = false);   // RG.cpp:699-793
    // plane detection: RG.cpp:277-694.  The image stages run on the GPU (rgbdslam_detect_regions);
    // the median plane fit is host code as in the reference.  DetectPlanes = the first branch of RG::Finish
    // (RG.cpp:154-175): DetectPrimitives + CreatePrimitives on the current model frame; returns the plane count.
    void DetectPrimitives(float *pose);
    int DetectPlanes();
    // RG.cpp:934-1006: drops the planes that hold at most MIN_SIZE_PLAN labelled pixels, renumbers the rest (labels, equations,
    // centres = mean labelled point, the Graph's primitives); returns the planes kept.  RG::Finish calls it when
    // finish_reorganize is set (in the reference everything it produces is overwritten by the detection that follows).
    int ReorganizeEqua(float *pose);
    bool finish_reorganize = false;
    // RG::Finish (RG.cpp:149-271), step 12 of display() (GP-demo.cpp:274): the first plane detection once _count > 30, then
    // every 100 frames the dump of the model (SaveModel into save_dir; synchronous -- the reference's Win32 SaveThread is
    // out of scope), the reset of the plane set and a new detection.  As in RG.cpp:177-187 the WHOLE periodic block is
    // skipped while save_dir holds a file Dummy.txt; it is also skipped when there is no save_dir (nothing could be dumped, and
    // the fused planes are not thrown away silently) unless finish_reset_without_dump is set.
    // Step() / StepAsync() call it; a frame whose tracking was lost skips it (GP-demo.cpp:196).
    void Finish();
    bool finish_reset_without_dump = false;
    int finish_skipped = 0;                                        // periods that were due but could not be saved
    bool finish_enabled = true;
    int finish_first_after = 30, finish_period = 100;              // RG.cpp:154,187
    std::string save_dir;
    bool _first_in = true;                                         // RG.h:74,194
    int _indx = 0;                                                 // RG.h:49,164
    int finish_detections = 0, finish_dumps = 0;
    int min_size_plan = 5000;                                      // MIN_SIZE_PLAN (Utilities.h:85)
    int min_normals_plan = 1000;                                   // RG.cpp:500
    bool _Detect = false;
    // one pass of display() (GP-demo.cpp:138-296): 2 = model seeded, 1 = tracked, 0 = lost
    int Step();
    // fused pipeline (default): the whole pass is enqueued without a host sync; the pose, status and
    // iteration counts stay on the device until Sync().  Returns 2 for the seeding frame, else 1.
    int StepAsync();
    int Sync();                                                    // -> last frame's status: 1 tracked / 0 lost
    bool fused = true;
    // fused pipeline only, opt-in: frame k+1's preprocessing and frame k's Bump update run on two auxiliary
    // streams beside the tracking of frame k+1.  Measured SLOWER than the single stream at 640x480 (4260 vs
    // 4610 frames/s: the six cross-stream event operations per frame cost more than the 24 us of kernels they
    // take off the critical path), hence off by default.
    bool overlap = false;
    void SyncAll();                                                // main + auxiliary streams
    void SyncIfStale();                                            // Sync() unless nothing was enqueued since the last one
    static const int HISTORY_CAP = 8192;                           // per-frame pose ring (16 floats each)
    int frames_done = 0;
    long tracked_total = 0;                                        // frames tracked so far (as of the last Sync)
    const float *history_host(int frame_index);                    // valid after Sync(); frame_index in the last HISTORY_CAP
    float *history_dev() { return _history_dev.p; }                // the device ring itself (HISTORY_CAP entries of 16 floats)
    // Plane-partitioned multi-GPU: copies what the plane owners need of the frame just enqueued -- the merged
    // model VMap, RGBD, Mask and the frame's history entry (pose + tracked flag) -- into one packed device buffer at the
    // given byte offsets, stream-ordered after the frame's kernels; the caller broadcasts that buffer.
    void ExportFrame(void *dst_dev, size_t off_vmap, size_t off_rgbd, size_t off_mask, size_t off_hist);
    // the same for the thin pack (10 B per pixel): RGBD.w, colour bytes, Mask + the history entry; VMap and RGBD.rgb are
    // functions of those (Kernel.cu:54-72, 463-465) and every plane owner rebuilds them
    void ExportFrameThin(void *dst_dev, size_t off_w, size_t off_rgb8, size_t off_mask, size_t off_hist, size_t off_label = (size_t)-1);

    OffLineFrame *_frame;
    PredictedFrame *_predicted_frame;
    Graph *_graph;
    Mat3 _poseR, _poseRPrev;
    Vec3 _poset, _posetPrev;
    int _count, _nbPlan, _lvl;
    float _distThres, _angleThres, _thres_cvg_alpha, _thres_cvg_tr;
    int _max_iter[3];
    float Znear = 0.5f, Zfar = 10.0f;                              // GP-demo.cpp:18-19
    bool device_loop = true;   // true: whole Gauss-Newton loop on the GPU; false: per-iteration host solve
    int last_iterations = 0;
    int last_level_iterations = 0;   // packed per level, 8 bits each (device loop only)
    rgbdslam_ctx *ctx() { return _ctx; }
    struct { const uint16_t *depth = nullptr; size_t depth_pitch = 0; const uint8_t *bgr = nullptr; size_t bgr_pitch = 0; bool valid = false; } _next_hint;
    // where the hinted next frame is preprocessed: on the merge launch (default) or on the apply launch (RGBDSLAM_NEXT_ON_APPLY=1)
    bool _next_on_merge = getenv("RGBDSLAM_NEXT_ON_APPLY") == nullptr;
    // The apply pass of Graph::Update may wait for the NEXT frame's draw launch (it reads the model this frame leaves and the
    // planes; nothing before the next merge writes either): only while a caller that will certainly step the hinted next
    // frame allows it (run_device_sequence), and any Sync() runs a pending pass first.  RGBDSLAM_APPLY_NOW=1: never.
    bool _defer_apply = false;
    bool _defer_apply_enabled = getenv("RGBDSLAM_APPLY_NOW") == nullptr;
    struct { UpdateTables t; const float *hist = nullptr; bool valid = false; } _deferred;
    void FlushDeferredApply();
    std::vector<float> _Equations, _Centers;
    DevBuf<float> _Equations_dev;
    bool _equations_only = false;                                  // SetPlaneEquations: labels are maintained, nothing is fused here
private:
    rgbdslam_ctx *_ctx;
    float _calib[11];
    DevBuf<float> _intr_dev, _Rcurr_dev, _tcurr_dev, _img1, _Gx, _Gy, _pose_state_dev, _trace_dev;
    hipStream_t _load_stream = nullptr, _update_stream = nullptr;
    hipEvent_t _ev_merged = nullptr, _ev_updated = nullptr;
    // per-stage timers (SetStageTiming): HIP events at the stage boundaries of the last STAGE_RING frames
    static const int STAGE_RING = 64, STAGE_MARKS = 6;
    bool _stage_timing = false;
    std::vector<hipEvent_t> _stage_ev;                             // [STAGE_RING][STAGE_MARKS]
    std::vector<unsigned char> _stage_set;                         // which marks of a ring entry were recorded
    int _stage_frames = 0;
    void stage_mark(int mark);
    bool _update_pending = false;
    int _overlap_mode = 3;   // bit 0: loads on the load stream, bit 1: Bump update on the update stream
    bool _overlap_mode_set = false;
public:
    void set_overlap_mode(int mode) { _overlap_mode = mode; _overlap_mode_set = true; }
private:
    DevBuf<int32_t> _status_dev;
    float *_pinned = nullptr;   // 24 floats pose state + 4 ints status
    rgbdslam_pose_block *_pb_dev = nullptr, *_pb_host = nullptr;   // device block + pinned mirror
    DevBuf<float> _history_dev;
    float *_history_host = nullptr;
    int _history_synced = 0;
    DevBuf<int32_t> _blob_dev, _blob_sizes_dev, _detect_labels;
    DevBuf<uint8_t> _detect_bytes, _fit_ws;                        // _fit_ws: workspace of rgbdslam_fit_planes
    std::vector<int> _traj_count;                                  // dataset index of every processed frame
    std::vector<Mat3> _traj_R;                                     // unfused path: poses pushed by Step()
    std::vector<Vec3> _traj_t;
    rgbdslam_icp_params icp_params() const;
    void configure_streams();
};

}  // namespace slam
