// Stand-in for the reference's cudaTypes.cuh, used ONLY to syntax- and link-check integration/rgbdslam_shim.cpp in this
// repository's CPU tests (tests/test_shim_compiles.py).  It supplies the handful of type names the four operator headers
// (Kernel.cuh, ICP.cuh, Primitive.cuh, RG.cuh) mention; a maintainer building inside the reference tree uses the real
// header (CUDA / OpenCV / Windows types) instead and never sees this file.
#ifndef RGBDSLAM_STUB_CUDATYPES
#define RGBDSLAM_STUB_CUDATYPES
#include <stddef.h>
#include <vector>
typedef unsigned char uchar;
typedef unsigned char BYTE;     // Windows: 8 bit
typedef unsigned short USHORT;  // Windows: 16 bit
typedef int LONG;               // Windows LONG is 32 bit on every Windows ABI
struct cudaArray;               // opaque in the headers
namespace cv { namespace gpu {
template <typename T> struct DevMem2D_ {   // OpenCV 2.4 core/devmem2d.hpp: {cols, rows, data, step}
    int cols, rows;
    T *data;
    size_t step;                           // bytes per row
};
} }
using namespace std;
#endif
