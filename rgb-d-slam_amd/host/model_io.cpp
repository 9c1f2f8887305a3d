// On-disk model format; see model_io.hpp for the reference map.
#include "model_io.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

#include "rgbdslam_host.hpp"

namespace slam {

bool PlanarPatch::Clean() {
    if (!has_data) return false;
    const int n = Size[0], m = Size[1];
    int count = 0;
    int BBox[4] = {10000, -10000, 10000, -10000};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            const size_t c = (size_t)i * m + j;
            if (Mask[c] > 10) {                                  // Primitive.cpp:14-19
                if (i < BBox[0]) BBox[0] = i;
                if (i > BBox[1]) BBox[1] = i;
                if (j < BBox[2]) BBox[2] = j;
                if (j > BBox[3]) BBox[3] = j;
                count++;
            } else {                                             // Primitive.cpp:20-27: unobserved cells lose colour and geometry
                RGB[3 * c] = RGB[3 * c + 1] = RGB[3 * c + 2] = 0;
                Bump[3 * c] = Bump[3 * c + 1] = Bump[3 * c + 2] = 0;
            }
        }
    if (count < 10) {                                            // Primitive.cpp:31-36
        has_data = false;
        Bump.clear(); RGB.clear(); Mask.clear();
        return false;
    }
    const int nn = BBox[1] - BBox[0] + 1, mm = BBox[3] - BBox[2] + 1;
    std::vector<uint16_t> TheBump((size_t)3 * nn * mm, 0), TheMask((size_t)nn * mm, 0);
    std::vector<uint8_t> TheRGB((size_t)3 * nn * mm, 0);
    Shift[0] = (float)((double)Shift[0] + (double)(float)BBox[0] * RES_PLANE);   // Primitive.cpp:50-51 (RES_PLANE is a double literal)
    Shift[1] = (float)((double)Shift[1] + (double)(float)BBox[2] * RES_PLANE);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            const size_t c = (size_t)i * m + j;
            if (Mask[c] < 11) continue;
            const size_t d = (size_t)(i - BBox[0]) * mm + (j - BBox[2]);
            for (int k = 0; k < 3; k++) { TheBump[3 * d + k] = Bump[3 * c + k]; TheRGB[3 * d + k] = RGB[3 * c + k]; }
            TheMask[d] = Mask[c];
        }
    Size[0] = nn; Size[1] = mm;
    Bump.swap(TheBump); RGB.swap(TheRGB); Mask.swap(TheMask);
    return true;
}

namespace {
std::ofstream open_out(const std::string &name) {
    std::ofstream f(name.c_str(), std::ios::out);
    if (!f.is_open()) throw Error("Could not open " + name, RGBDSLAM_E_ARG);
    return f;
}
}  // namespace

void SaveGraph(const std::string &dir, int indx, const float *P, const std::vector<PlanarPatch> &nodes) {
    {   // Graph.cpp:107-119: the pose goes out column by column under the NEXT index
        std::ofstream f = open_out(dir + "/Pose" + std::to_string(indx + 1) + ".txt");
        for (int k = 0; k < 16; k++) f << P[k] << (k == 15 ? "" : " ");
        f << std::endl;
    }
    if (indx == -1) return;                                      // Graph.cpp:121-122
    std::ofstream f = open_out(dir + "/Graph-eq" + std::to_string(indx) + ".txt");
    f << nodes.size() << std::endl;
    int count = 0;
    for (const PlanarPatch &p : nodes) {
        if (!p.has_data) {                                       // Graph.cpp:137-145: 14 zeros and a 0 x 0 size, no anchor
            for (int k = 0; k < 14; k++) f << 0.0 << " ";
            f << 0 << " " << 0 << std::endl;
            count++;
            continue;
        }
        f << p.nmle[0] << " " << p.nmle[1] << " " << p.nmle[2] << " " << p.dist << " " << p.e1[0] << " " << p.e1[1] << " " << p.e1[2] << " "
          << p.e2[0] << " " << p.e2[1] << " " << p.e2[2] << " " << p.Shift[0] << " " << p.Shift[1] << " " << p.scale[0] << " " << p.scale[1]
          << " " << p.Size[0] << " " << p.Size[1] << " " << p.anchor[0] << " " << p.anchor[1] << " " << p.anchor[2] << std::endl;
        const int n = p.Size[0], m = p.Size[1];
        Image img;
        img.rows = n; img.cols = m; img.channels = 3; img.bits = 16;
        img.data.assign((size_t)n * m * 6, 0);
        uint16_t *px = (uint16_t *)img.data.data();
        const std::string tail = std::to_string(indx) + "-" + std::to_string(count) + ".tif";
        // Graph.cpp:167-179: rows flipped, channel 2 <- component 0 (memory order B,G,R = components 2,1,0)
        for (int i = 0, k = n - 1; i < n; i++, k--)
            for (int j = 0; j < m; j++)
                for (int c = 0; c < 3; c++) px[3 * ((size_t)k * m + j) + (2 - c)] = p.Bump[3 * ((size_t)i * m + j) + c];
        write_tiff(dir + "/Bump" + tail, img);
        for (int i = 0, k = n - 1; i < n; i++, k--)              // Graph.cpp:190-202: mask x 200 in all three channels
            for (int j = 0; j < m; j++)
                for (int c = 0; c < 3; c++) px[3 * ((size_t)k * m + j) + c] = (uint16_t)(200 * p.Mask[(size_t)i * m + j]);
        write_tiff(dir + "/Mask" + tail, img);
        for (int i = 0, k = n - 1; i < n; i++, k--)              // Graph.cpp:213-222: colour x 200
            for (int j = 0; j < m; j++)
                for (int c = 0; c < 3; c++) px[3 * ((size_t)k * m + j) + (2 - c)] = (uint16_t)(200 * p.RGB[3 * ((size_t)i * m + j) + c]);
        write_tiff(dir + "/RGB" + tail, img);
        count++;
    }
    f.close();
    open_out(dir + "/Dummy" + std::to_string(indx) + ".txt");    // Graph.cpp:240-246: the ready flag, written last
}

void SaveFrame(const std::string &dir, int indx, const float *rgbd, int n, int m) {
    Image d;
    d.rows = n; d.cols = m; d.channels = 1; d.bits = 16;
    d.data.assign((size_t)n * m * 2, 0);
    uint16_t *pd = (uint16_t *)d.data.data();
    for (int i = 0, k = n - 1; i < n; i++, k--)                  // Frame.cpp:196-204: depth in 1/5000 m
        for (int j = 0; j < m; j++)
            pd[(size_t)k * m + j] = (uint16_t)(long)((double)rgbd[4 * ((size_t)i * m + j) + 3] * 100.0 * 5000.0);
    write_tiff(dir + "/Depth" + std::to_string(indx) + ".tiff", d);
    Image c;
    c.rows = n; c.cols = m; c.channels = 3; c.bits = 16;
    c.data.assign((size_t)n * m * 6, 0);
    uint16_t *pc = (uint16_t *)c.data.data();
    for (int i = 0, k = n - 1; i < n; i++, k--)                  // Frame.cpp:212-218
        for (int j = 0; j < m; j++)
            for (int ch = 0; ch < 3; ch++)
                pc[3 * ((size_t)k * m + j) + (2 - ch)] = (uint16_t)(200 * (int)(uint16_t)(long)((double)rgbd[4 * ((size_t)i * m + j) + ch] * 255.0));
    write_tiff(dir + "/RGB" + std::to_string(indx) + ".tiff", c);
}

bool LoadModel(const std::string &dir, int indx, LoadedModel *out) {
    {
        std::ifstream flag((dir + "/Dummy" + std::to_string(indx) + ".txt").c_str());
        if (!flag.is_open()) return false;
    }
    // KeyFrame(path, indx + 1)
    const std::string k = std::to_string(indx + 1);
    {
        std::ifstream f((dir + "/Pose" + k + ".txt").c_str());
        if (!f.is_open()) throw Error("Could not open " + dir + "/Pose" + k + ".txt", RGBDSLAM_E_ARG);
        std::string line;
        std::getline(f, line);
        float *p = out->pose;
        for (int q = 0; q < 16; q++) p[q] = (q % 5 == 0) ? 1.0f : 0.0f;
        sscanf(line.c_str(), "%f %f %f %f %f %f %f %f %f %f %f %f %f %f %f %f", p, p + 1, p + 2, p + 3, p + 4, p + 5, p + 6, p + 7, p + 8,
               p + 9, p + 10, p + 11, p + 12, p + 13, p + 14, p + 15);
    }
    out->rgb = read_tiff(dir + "/RGB" + k + ".tiff");
    out->depth = read_tiff(dir + "/Depth" + k + ".tiff");
    std::ifstream f((dir + "/Graph-eq" + std::to_string(indx) + ".txt").c_str());
    if (!f.is_open()) throw Error("Could not open " + dir + "/Graph-eq" + std::to_string(indx) + ".txt", RGBDSLAM_E_ARG);
    std::string line;
    std::getline(f, line);
    int nbPrim = 0;
    sscanf(line.c_str(), "%d", &nbPrim);
    out->patches.clear();
    for (int p = 0; p < nbPrim; p++) {
        PlanarPatch P;
        std::getline(f, line);
        sscanf(line.c_str(), "%f %f %f %f %f %f %f %f %f %f %f %f %f %f %d %d %f %f %f", &P.nmle[0], &P.nmle[1], &P.nmle[2], &P.dist,
               &P.e1[0], &P.e1[1], &P.e1[2], &P.e2[0], &P.e2[1], &P.e2[2], &P.Shift[0], &P.Shift[1], &P.scale[0], &P.scale[1], &P.Size[0],
               &P.Size[1], &P.anchor[0], &P.anchor[1], &P.anchor[2]);
        if (P.Size[0] * P.Size[1] == 0) continue;                // Model3D.cpp:1107-1108
        const int n = P.Size[0], m = P.Size[1];
        const std::string tail = std::to_string(indx) + "-" + std::to_string(p) + ".tif";
        Image ib = read_tiff(dir + "/Bump" + tail), ic = read_tiff(dir + "/RGB" + tail), im = read_tiff(dir + "/Mask" + tail);
        for (const Image *I : {&ib, &ic, &im})
            if (I->rows != n || I->cols != m || I->channels != 3 || I->bits != 16)
                throw Error("plane image does not match the size in Graph-eq: " + tail, RGBDSLAM_E_ARG);
        const uint16_t *pb = (const uint16_t *)ib.data.data(), *pc = (const uint16_t *)ic.data.data(), *pm = (const uint16_t *)im.data.data();
        P.Bump.assign((size_t)3 * n * m, 0); P.RGB.assign((size_t)3 * n * m, 0); P.Mask.assign((size_t)n * m, 0);
        for (int i = 0, kk = n - 1; i < n; i++, kk--)            // Model3D.cpp:1154-1166
            for (int j = 0; j < m; j++) {
                for (int c = 0; c < 3; c++) {
                    P.Bump[3 * ((size_t)kk * m + j) + c] = pb[3 * ((size_t)i * m + j) + (2 - c)];
                    P.RGB[3 * ((size_t)kk * m + j) + c] = (uint8_t)(pc[3 * ((size_t)i * m + j) + (2 - c)] / 200);
                }
                P.Mask[(size_t)kk * m + j] = (uint16_t)(pm[3 * ((size_t)i * m + j)] / 200);
            }
        P.has_data = true;
        out->patches.push_back(P);
    }
    return true;
}

}  // namespace slam
