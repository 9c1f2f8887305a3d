// Dataset I/O and trajectory output of the reference's host side.
//   LoadTimeStamps / LoadAssociatedTimeStamps   D1/Utilities.cpp:171-276, 278-304
//   frame file naming rules of RG::LoadFrame    D1/RG.cpp:105-139   (and the dataset sizing of RG.h:145-156)
//   matrix2quaternion / SaveTrajectory          D1/Utilities.cpp:11-68, 500-546
//   cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)    D1/Frame.cpp:48,58  (PNG container, written here over zlib)
// Host code only; the frames it produces enter the GPU path through OffLineFrame::LoadFrame.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace slam {

struct Mat3;
struct Vec3;

// ---- image files ----------------------------------------------------------------------------------------
// What cv::imread(file, CV_LOAD_IMAGE_UNCHANGED) hands the reference: rows x cols x channels, 8 or 16 bit
// samples in host byte order, colour in B,G,R(,A) order, rows top to bottom, no row padding.
struct Image {
    int rows = 0, cols = 0, channels = 0, bits = 0;
    std::vector<uint8_t> data;
    size_t row_bytes() const { return (size_t)cols * channels * (bits / 8); }
};
Image read_png(const std::string &filename);                       // throws slam::Error
void write_png(const std::string &filename, const Image &img);    // 8/16 bit, 1/3/4 channels (B,G,R order in memory)
// Baseline TIFF 6.0 (the container of the reference's "real" input frames RG.cpp:134-138 and of its model
// dumps, Graph.cpp:97-248): strips, 8/16 bit, 1/3/4 samples, uncompressed / LZW / PackBits, either byte order.
Image read_tiff(const std::string &filename);
void write_tiff(const std::string &filename, const Image &img, bool lzw = true);   // cv::imwrite default: LZW

// ---- TUM RGB-D directory listing ------------------------------------------------------------------------
// Utilities.cpp:171-276: rgb.txt x depth.txt association, 20 ms window; names are the file columns.
void LoadTimeStamps(std::vector<std::string> *timestamp_depth, std::vector<std::string> *timestamp_color, const char *path);
// Utilities.cpp:278-304: pre-associated "associated.txt" (t_rgb file_rgb t_depth file_depth per line)
void LoadAssociatedTimeStamps(std::vector<std::string> *timestamp_depth, std::vector<std::string> *timestamp_color, const char *path);

// The three sources RG::LoadFrame knows (RG.cpp:105-139) and the sequence length rule of RG.h:145-156.
class Dataset {
public:
    // real: Depth_%d.tiff / RGB_%d.tiff; else RGBDRep: TUM listing; else numbered depth/color PNGs
    Dataset(const char *path, int Sizedata, bool real, bool RGBDRep);
    int size() const { return _Size_data; }
    // file names of the frame the reference loads when its counter is `count` (first call: count == 1)
    void frame_files(int count, std::string *filename_depth, std::string *filename_color) const;
    const std::string &timestamp_depth(int count) const { return _timestamp_depth.at(count); }
    std::string _path;
    bool _real, _RGBDRep;
    std::vector<std::string> _timestamp_depth, _timestamp_color;
private:
    int _Size_data;
};

// ---- trajectory -----------------------------------------------------------------------------------------
void matrix2quaternion(const Mat3 &currR, float *res);             // res = (w, x, y, z)
// TUM trajectory text: "<stamp> tx ty tz qx qy qz qw"; poses whose t(0) == 100 (lost marker) are skipped.
void SaveTrajectory(const char *filename, const std::vector<Mat3> &PosesR, const std::vector<Vec3> &Posest,
                    const std::vector<std::string> &timestamp);

}  // namespace slam
