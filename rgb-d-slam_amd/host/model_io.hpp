// On-disk model format of the reference: what DemoStep1 dumps every 100 frames and
// DemoStep2 reads back.
//   writer  Primitive::Clean       D1/Primitive.cpp:4-134     (crop every plane to the bounding box of its observed cells)
//           Graph::save            D1/Graph.cpp:97-248        (Pose<i+1>.txt, Graph-eq<i>.txt, Bump/Mask/RGB<i>-<k>.tif, Dummy<i>.txt)
//           PredictedFrame::save   D1/Frame.cpp:191-227       (Depth<i>.tiff, RGB<i>.tiff key-frame images)
//           SaveThread             D1/RG.cpp:16-59            (the order: Clean, frame->save(indx+1), graph->save(indx))
//   reader  Model3D::Update        D2/Model3D.cpp:1050-1184   (plane list + images -> patches)
//           KeyFrame::KeyFrame     D2/KeyFrame.h:49-104       (pose + key-frame images)
// Host code; the TIFF container is host/tiff_io.cpp.  Paths are joined with '/' (the reference hard-codes '\\').
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "dataset_io.hpp"

namespace slam {

// host copy of one Primitive's data model (D1/Primitive.h:17-60)
struct PlanarPatch {
    float nmle[3] = {0, 0, 0}, e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0}, dist = 0, scale[2] = {0, 0}, Shift[2] = {0, 0}, anchor[3] = {0, 0, 0};
    int Size[2] = {0, 0};
    bool has_data = false;                   // false = the reference's "_Bump == NULL" state after a failed Clean
    std::vector<uint16_t> Bump;              // 3 per cell
    std::vector<uint8_t> RGB;                // 3 per cell
    std::vector<uint16_t> Mask;              // 1 per cell (bytes in DemoStep1, shorts once loaded by DemoStep2)
    bool Clean();                            // D1/Primitive.cpp:4-134
};

void SaveGraph(const std::string &dir, int indx, const float *pose16_cm, const std::vector<PlanarPatch> &nodes);   // Graph::save
void SaveFrame(const std::string &dir, int indx, const float *rgbd, int n, int m);                                  // PredictedFrame::save

struct LoadedModel {
    float pose[16];                          // column-major like Eigen (KeyFrame.h:92-95)
    std::vector<PlanarPatch> patches;        // planes whose size is 0 are skipped (Model3D.cpp:1107)
    Image depth, rgb;                        // key-frame images as stored (row-flipped, x5000 / x200 scaled)
};
// false while Dummy<indx>.txt does not exist yet (the writer's ready flag, Model3D.cpp:1061-1066)
bool LoadModel(const std::string &dir, int indx, LoadedModel *out);

}  // namespace slam
