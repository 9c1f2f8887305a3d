// Dataset I/O and trajectory output; see dataset_io.hpp for the reference map.
#include "dataset_io.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rgbdslam_host.hpp"

namespace slam {

// =========================================================================================================
// PNG container (ISO/IEC 15948) over zlib.  Output convention = cv::imread(CV_LOAD_IMAGE_UNCHANGED) of
// OpenCV 2.4 (D1/Frame.cpp:48,58): grey stays 1 channel, 16-bit stays 16-bit (host byte order), RGB -> BGR,
// RGBA -> BGRA, palette -> BGR, sub-byte grey scaled to 8 bit.  Grey+alpha is refused (OpenCV's handling
// of it differs between versions and the reference never reads one).
// =========================================================================================================
namespace {

const uint8_t PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

std::vector<uint8_t> slurp(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) throw Error("cannot open " + filename, RGBDSLAM_E_ARG);
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t k;
    while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
    fclose(f);
    return buf;
}

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the scanline filters of one (sub)image in place; `src` holds rows of (1 + stride) bytes
void unfilter(uint8_t *src, size_t rows, size_t stride, size_t bpp, const std::string &filename) {
    for (size_t y = 0; y < rows; y++) {
        uint8_t *line = src + y * (stride + 1), ft = line[0], *cur = line + 1;
        const uint8_t *up = y ? line - stride : nullptr;      // previous row's pixel bytes
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, add;
            switch (ft) {
            case 0: add = 0; break;
            case 1: add = a; break;
            case 2: add = b; break;
            case 3: add = (a + b) >> 1; break;
            case 4: add = paeth(a, b, c); break;
            default: throw Error("bad PNG filter type in " + filename, RGBDSLAM_E_ARG);
            }
            cur[x] = (uint8_t)(cur[x] + add);
        }
    }
}

}  // namespace

Image read_png(const std::string &filename) {
    std::vector<uint8_t> file = slurp(filename);
    if (file.size() < 8 || memcmp(file.data(), PNG_SIG, 8)) throw Error("not a PNG file: " + filename, RGBDSLAM_E_ARG);
    uint32_t W = 0, H = 0;
    int bitdepth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool end = false;
    for (size_t off = 8; !end;) {
        if (off + 12 > file.size()) throw Error("truncated PNG: " + filename, RGBDSLAM_E_ARG);
        uint32_t len = be32(&file[off]);
        if (off + 12 + (size_t)len > file.size()) throw Error("truncated PNG chunk: " + filename, RGBDSLAM_E_ARG);
        const uint8_t *tag = &file[off + 4], *body = &file[off + 8];
        if ((uint32_t)crc32(0L, tag, 4 + len) != be32(body + len)) throw Error("PNG chunk CRC mismatch: " + filename, RGBDSLAM_E_ARG);
        if (!memcmp(tag, "IHDR", 4)) {
            if (len != 13) throw Error("bad IHDR: " + filename, RGBDSLAM_E_ARG);
            W = be32(body); H = be32(body + 4);
            bitdepth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] || body[11] || interlace > 1) throw Error("unsupported PNG method: " + filename, RGBDSLAM_E_ARG);
        } else if (!memcmp(tag, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(tag, "IEND", 4)) {
            end = true;
        }
        off += 12 + (size_t)len;
    }
    int samples;
    switch (ctype) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 6: samples = 4; break;
    default: throw Error("unsupported PNG colour type in " + filename, RGBDSLAM_E_ARG);
    }
    bool depth_ok = (ctype == 0 && (bitdepth == 1 || bitdepth == 2 || bitdepth == 4 || bitdepth == 8 || bitdepth == 16)) ||
                    (ctype == 3 && (bitdepth == 1 || bitdepth == 2 || bitdepth == 4 || bitdepth == 8)) ||
                    ((ctype == 2 || ctype == 6) && (bitdepth == 8 || bitdepth == 16));
    if (!W || !H || W > 65535u || H > 65535u || !depth_ok) throw Error("bad PNG header: " + filename, RGBDSLAM_E_ARG);
    if (ctype == 3 && (plte.empty() || plte.size() % 3)) throw Error("PNG palette missing: " + filename, RGBDSLAM_E_ARG);

    const size_t bits_px = (size_t)samples * bitdepth, bpp = bits_px >= 8 ? bits_px / 8 : 1;
    // pass geometry: one pass for a plain image, the 7 Adam7 passes otherwise
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    size_t pw[7], ph[7], total = 0;
    for (int p = 0; p < npass; p++) {
        pw[p] = interlace ? (W > (uint32_t)ax0[p] ? (W - ax0[p] + adx[p] - 1) / adx[p] : 0) : W;
        ph[p] = interlace ? (H > (uint32_t)ay0[p] ? (H - ay0[p] + ady[p] - 1) / ady[p] : 0) : H;
        if (pw[p] && ph[p]) total += ph[p] * (1 + (pw[p] * bits_px + 7) / 8);
    }
    std::vector<uint8_t> raw(total);
    uLongf got = (uLongf)total;
    int zrc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || got != total) throw Error("PNG data stream corrupt or of the wrong size: " + filename, RGBDSLAM_E_ARG);

    Image img;
    img.rows = (int)H; img.cols = (int)W;
    img.bits = bitdepth == 16 ? 16 : 8;
    img.channels = ctype == 3 ? 3 : samples;
    img.data.assign((size_t)H * img.row_bytes(), 0);
    const size_t obytes = (size_t)img.channels * (img.bits / 8);

    size_t roff = 0;
    for (int p = 0; p < npass; p++) {
        if (!pw[p] || !ph[p]) continue;
        const size_t stride = (pw[p] * bits_px + 7) / 8;
        unfilter(&raw[roff], ph[p], stride, bpp, filename);
        for (size_t y = 0; y < ph[p]; y++) {
            const uint8_t *row = &raw[roff + y * (stride + 1) + 1];
            const size_t oy = interlace ? ay0[p] + y * ady[p] : y;
            for (size_t x = 0; x < pw[p]; x++) {
                const size_t ox = interlace ? ax0[p] + x * adx[p] : x;
                uint8_t *o = &img.data[oy * img.row_bytes() + ox * obytes];
                if (bitdepth < 8) {
                    int v = (row[(x * bitdepth) >> 3] >> (8 - bitdepth - ((x * bitdepth) & 7))) & ((1 << bitdepth) - 1);
                    if (ctype == 3) {
                        if ((size_t)v * 3 + 2 >= plte.size()) throw Error("PNG palette index out of range: " + filename, RGBDSLAM_E_ARG);
                        o[0] = plte[v * 3 + 2]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3];
                    } else {
                        o[0] = (uint8_t)(v * 255 / ((1 << bitdepth) - 1));
                    }
                } else if (ctype == 3) {
                    int v = row[x];
                    if ((size_t)v * 3 + 2 >= plte.size()) throw Error("PNG palette index out of range: " + filename, RGBDSLAM_E_ARG);
                    o[0] = plte[v * 3 + 2]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3];
                } else if (bitdepth == 8) {
                    const uint8_t *s = row + x * samples;
                    if (samples == 1) o[0] = s[0];
                    else { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; if (samples == 4) o[3] = s[3]; }
                } else {   // 16 bit, big endian in the file
                    const uint8_t *s = row + x * samples * 2;
                    uint16_t v[4];
                    for (int c = 0; c < samples; c++) v[c] = (uint16_t)((s[2 * c] << 8) | s[2 * c + 1]);
                    if (samples >= 3) { uint16_t t = v[0]; v[0] = v[2]; v[2] = t; }
                    memcpy(o, v, 2 * samples);
                }
            }
        }
        roff += ph[p] * (stride + 1);
    }
    return img;
}

void write_png(const std::string &filename, const Image &img) {
    if (!(img.bits == 8 || img.bits == 16) || !(img.channels == 1 || img.channels == 3 || img.channels == 4) || img.rows <= 0 ||
        img.cols <= 0 || img.data.size() != (size_t)img.rows * img.row_bytes())
        throw Error("write_png: unsupported image", RGBDSLAM_E_ARG);
    const int samples = img.channels, bps = img.bits / 8;
    const size_t stride = img.row_bytes();
    std::vector<uint8_t> raw((size_t)img.rows * (stride + 1));
    for (int y = 0; y < img.rows; y++) {
        uint8_t *line = &raw[(size_t)y * (stride + 1)];
        line[0] = 0;   // filter "none": deflate does the work, the reader handles all five
        for (int x = 0; x < img.cols; x++)
            for (int c = 0; c < samples; c++) {
                int sc = (samples >= 3 && c < 3) ? 2 - c : c;   // B,G,R in memory -> R,G,B in the file
                const uint8_t *s = &img.data[(size_t)y * stride + ((size_t)x * samples + sc) * bps];
                uint8_t *d = line + 1 + ((size_t)x * samples + c) * bps;
                if (bps == 1) d[0] = s[0];
                else { uint16_t v; memcpy(&v, s, 2); d[0] = v >> 8; d[1] = v & 0xFF; }
            }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw Error("write_png: deflate failed", RGBDSLAM_E_ARG);
    FILE *f = fopen(filename.c_str(), "wb");
    if (!f) throw Error("cannot create " + filename, RGBDSLAM_E_ARG);
    auto chunk = [&](const char *tag, const uint8_t *body, uint32_t len) {
        std::vector<uint8_t> c(12 + (size_t)len);
        put_be32(&c[0], len);
        memcpy(&c[4], tag, 4);
        if (len) memcpy(&c[8], body, len);
        put_be32(&c[8 + len], (uint32_t)crc32(0L, &c[4], 4 + len));
        fwrite(c.data(), 1, c.size(), f);
    };
    fwrite(PNG_SIG, 1, 8, f);
    uint8_t ihdr[13];
    put_be32(ihdr, (uint32_t)img.cols); put_be32(ihdr + 4, (uint32_t)img.rows);
    ihdr[8] = (uint8_t)img.bits; ihdr[9] = samples == 1 ? 0 : (samples == 3 ? 2 : 6); ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    fclose(f);
}

// =========================================================================================================
// TUM listings
// =========================================================================================================
namespace {

// The reference reads with `getline(buf, 256); while (!eof()) { parse(buf); getline(buf, 256); }`
// (Utilities.cpp:190-202): a line is parsed only if the getline that produced it did not hit end of file,
// i.e. a final line without '\n' is dropped; `skip` header lines are consumed first.
bool listing_lines(const std::string &filename, int skip, std::vector<std::string> *out) {
    std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
    if (!f.is_open()) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string all = ss.str();
    size_t pos = 0;
    int lineno = 0;
    while (true) {
        size_t nl = all.find('\n', pos);
        if (nl == std::string::npos) break;                    // unterminated tail (or nothing): eof() was set by this read
        std::string line = all.substr(pos, nl - pos);
        if (line.size() > 255) throw Error("listing line longer than the reference's 256-byte buffer: " + filename, RGBDSLAM_E_LIMIT);
        if (lineno++ >= skip) out->push_back(line);
        pos = nl + 1;
    }
    return true;
}

struct Stamped {
    std::vector<double> t;
    std::vector<std::string> name;
};

// "%d.%d %s": the fraction is read as an INTEGER and divided by 1e6 (Utilities.cpp:198-199), which is right
// for TUM's 6-digit fractions only; a line that does not parse leaves the previous values in place.
void parse_listing(const std::vector<std::string> &lines, Stamped *s) {
    int tmp1 = 0, tmp2 = 0;
    char str[256] = {0};
    for (const std::string &l : lines) {
        sscanf(l.c_str(), "%d.%d %255s", &tmp1, &tmp2, str);
        s->t.push_back(double(tmp1) + double(tmp2) / 1000000.0);
        s->name.push_back(std::string(str));
    }
}

}  // namespace

void LoadTimeStamps(std::vector<std::string> *timestamp_depth, std::vector<std::string> *timestamp_color, const char *path) {
    std::vector<std::string> ld, lc;
    Stamped D, Cc;
    if (!listing_lines(std::string(path) + "//depth.txt", 3, &ld)) return;      // reference prints and returns
    if (!listing_lines(std::string(path) + "//rgb.txt", 3, &lc)) return;
    parse_listing(ld, &D);
    parse_listing(lc, &Cc);

    // Utilities.cpp:232-270.  For every colour stamp (in order) the closest depth stamp strictly inside
    // the 20 ms window; matched entries leave BOTH lists.  The reference walks the colour list with an
    // iterator AND an index; erasing re-seats the iterator one element ahead of the index, so after the
    // first match the loop ends one element early (the last colour stamp is never examined).  `it_pos`
    // restates the iterator, `indx_color` the index.  Where the reference would step the iterator past
    // end() (a match on the very last element before any other match: undefined behaviour), this stops.
    const double max_difference_depth = 0.02;
    size_t it_pos = 0;
    long indx_color = -1;
    while (it_pos < Cc.t.size()) {
        indx_color++;
        const double val = Cc.t[(size_t)indx_color];
        double min_val = max_difference_depth;
        size_t position = 0;
        for (size_t k = 0; k < D.t.size(); k++) {
            const double val2 = D.t[k];
            if (val2 > (val + max_difference_depth)) break;
            const double diff = fabs(val - val2);
            if (diff < min_val && diff < max_difference_depth) { min_val = diff; position = k; }
        }
        if (min_val == max_difference_depth) { it_pos++; continue; }
        timestamp_depth->push_back(D.name[position]);
        timestamp_color->push_back(Cc.name[(size_t)indx_color]);
        Cc.t.erase(Cc.t.begin() + indx_color);
        Cc.name.erase(Cc.name.begin() + indx_color);
        D.t.erase(D.t.begin() + position);
        D.name.erase(D.name.begin() + position);
        it_pos = (size_t)indx_color + 1;                       // it = erase(...) ; it++
        indx_color--;
    }
}

void LoadAssociatedTimeStamps(std::vector<std::string> *timestamp_depth, std::vector<std::string> *timestamp_color, const char *path) {
    std::vector<std::string> lines;
    if (!listing_lines(std::string(path) + "//associated.txt", 0, &lines)) return;
    int t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    char str1[256] = {0}, str2[256] = {0};
    for (const std::string &l : lines) {
        sscanf(l.c_str(), "%d.%d %255s %d.%d %255s", &t1, &t2, str1, &t3, &t4, str2);
        timestamp_color->push_back(std::string(str1));         // first column pair is the colour image (Utilities.cpp:297)
        timestamp_depth->push_back(std::string(str2));
    }
}

Dataset::Dataset(const char *path, int Sizedata, bool real, bool RGBDRep) : _path(path), _real(real), _RGBDRep(RGBDRep) {
    if (!real && RGBDRep) {                                    // RG.h:146-149
        LoadTimeStamps(&_timestamp_depth, &_timestamp_color, path);
        _Size_data = (int)_timestamp_depth.size();
    } else {
        _Size_data = Sizedata;                                 // RG.h:151,154
    }
}

void Dataset::frame_files(int count, std::string *filename_depth, std::string *filename_color) const {
    char buff[32];
    if (!_real) {
        if (_RGBDRep) {                                        // RG.cpp:106-108
            *filename_depth = _path + "//" + _timestamp_depth.at((size_t)count);
            *filename_color = _path + "//" + _timestamp_color.at((size_t)count);
        } else {                                               // RG.cpp:110-131: 1-based, zero-padded to 6 digits below 10000
            if (count + 1 >= 10000) throw Error("numbered dataset: the reference names no file past 9999", RGBDSLAM_E_LIMIT);
            snprintf(buff, sizeof(buff), "%06d.png", count + 1);
            *filename_depth = _path + "//depth//" + buff;
            *filename_color = _path + "//color//" + buff;
        }
    } else {                                                   // RG.cpp:134-138: 0-based TIFF pairs
        snprintf(buff, sizeof(buff), "%d.tiff", count);
        *filename_depth = _path + "//Depth_" + buff;
        *filename_color = _path + "//RGB_" + buff;
    }
}

// =========================================================================================================
// Trajectory
// =========================================================================================================
void matrix2quaternion(const Mat3 &R, float *res) {
    // Utilities.cpp:11-68: largest-diagonal branches in float (0.5/r and 0.5*r go through double), and a
    // trace branch in double when no diagonal element is strictly largest.
    const float r00 = R(0, 0), r11 = R(1, 1), r22 = R(2, 2);
    if (r00 > r11 && r00 > r22) {
        float r = sqrtf(1 + r00 - r11 - r22);
        float s = (float)(0.5 / r);
        res[0] = (R(2, 1) - R(1, 2)) * s;
        res[1] = (float)(0.5 * r);
        res[2] = (R(0, 1) + R(1, 0)) * s;
        res[3] = (R(2, 0) + R(0, 2)) * s;
        return;
    }
    if (r11 > r00 && r11 > r22) {
        float r = sqrtf(1 + r11 - r00 - r22);
        float s = (float)(0.5 / r);
        res[0] = (R(2, 0) - R(0, 2)) * s;
        res[1] = (R(0, 1) + R(1, 0)) * s;
        res[2] = (float)(0.5 * r);
        res[3] = (R(2, 1) + R(1, 2)) * s;
        return;
    }
    if (r22 > r00 && r22 > r11) {
        float r = sqrtf(1 + r22 - r00 - r11);
        float s = (float)(0.5 / r);
        res[0] = (R(1, 0) - R(0, 1)) * s;
        res[1] = (R(0, 2) + R(2, 0)) * s;
        res[2] = (R(2, 1) + R(1, 2)) * s;
        res[3] = (float)(0.5 * r);
        return;
    }
    float qw = (float)(sqrt(1.0 + r00 + r11 + r22) / 2.0);
    res[0] = qw;
    res[1] = (float)((R(2, 1) - R(1, 2)) / (4.0 * qw));
    res[2] = (float)((R(0, 2) - R(2, 0)) / (4.0 * qw));
    res[3] = (float)((R(1, 0) - R(0, 1)) / (4.0 * qw));
}

void SaveTrajectory(const char *filename, const std::vector<Mat3> &PosesR, const std::vector<Vec3> &Posest,
                    const std::vector<std::string> &timestamp) {
    if (PosesR.size() < Posest.size() || timestamp.size() < Posest.size())
        throw Error("SaveTrajectory: fewer rotations/timestamps than translations", RGBDSLAM_E_ARG);
    std::ofstream f(filename, std::ios::out);
    if (!f.is_open()) throw Error(std::string("Could not open ") + filename, RGBDSLAM_E_ARG);
    f << "# Etimated camera trajectory\n";                     // header text as the reference writes it (Utilities.cpp:513)
    float q[4];
    for (size_t i = 0; i < Posest.size(); i++) {
        if (Posest[i](0) == 100.0) continue;                   // lost-frame marker
        // characters 6..22 of the listing entry: "depth/1305031102.160407.png" -> "1305031102.160407"
        if (timestamp[i].size() < 6) throw Error("SaveTrajectory: timestamp entry shorter than 6 characters", RGBDSLAM_E_ARG);
        const std::string stamp = timestamp[i].substr(6, 17);
        matrix2quaternion(PosesR[i], q);
        f << stamp << " " << Posest[i](0) << " " << Posest[i](1) << " " << Posest[i](2) << " " << q[1] << " " << q[2] << " " << q[3]
          << " " << q[0] << std::endl;                         // default ostream float format (= %g, 6 digits)
    }
}

}  // namespace slam
