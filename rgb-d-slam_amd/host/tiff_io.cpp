// Baseline TIFF 6.0 reader/writer.  The reference moves two kinds of TIFF through cv::imread / cv::imwrite
// (OpenCV 2.4 over libtiff): the "real" capture frames Depth_%d.tiff / RGB_%d.tiff (D1/RG.cpp:134-138) and
// the model dump Bump/Mask/RGB%d-%d.tif (D1/Graph.cpp:97-248, read back by D2/Model3D.cpp:1050-1209).
// Neither OpenCV nor libtiff headers exist in this image, so the container is written out here from the
// TIFF 6.0 specification: strips, 8/16-bit unsigned samples, 1/3/4 samples per pixel, chunky planar
// configuration, compression none / LZW (with or without horizontal predictor) / PackBits.
// Memory convention as dataset_io.hpp: B,G,R(,A) order, host byte order, top-down rows.
#include <cstdio>
#include <cstring>

#include "dataset_io.hpp"
#include "rgbdslam_host.hpp"

namespace slam {
namespace {

struct Reader {
    const std::vector<uint8_t> &b;
    bool big;
    const std::string &name;
    void need(size_t off, size_t len) const {
        if (off + len > b.size() || off + len < off) throw Error("truncated TIFF: " + name, RGBDSLAM_E_ARG);
    }
    uint16_t u16(size_t off) const { need(off, 2); return big ? (uint16_t)((b[off] << 8) | b[off + 1]) : (uint16_t)((b[off + 1] << 8) | b[off]); }
    uint32_t u32(size_t off) const {
        need(off, 4);
        return big ? ((uint32_t)b[off] << 24) | ((uint32_t)b[off + 1] << 16) | ((uint32_t)b[off + 2] << 8) | b[off + 3]
                   : ((uint32_t)b[off + 3] << 24) | ((uint32_t)b[off + 2] << 16) | ((uint32_t)b[off + 1] << 8) | b[off];
    }
    // values of one directory entry (types BYTE, SHORT, LONG)
    std::vector<uint32_t> values(size_t entry) const {
        uint16_t type = u16(entry + 2);
        uint32_t cnt = u32(entry + 4);
        size_t sz = type == 1 ? 1 : type == 3 ? 2 : type == 4 ? 4 : 0;
        if (!sz) throw Error("unsupported TIFF field type: " + name, RGBDSLAM_E_ARG);
        if (cnt > (1u << 24)) throw Error("implausible TIFF field count: " + name, RGBDSLAM_E_ARG);
        size_t off = (size_t)cnt * sz <= 4 ? entry + 8 : u32(entry + 8);
        std::vector<uint32_t> v(cnt);
        for (uint32_t k = 0; k < cnt; k++) {
            size_t o = off + k * sz;
            need(o, sz);
            v[k] = sz == 1 ? b[o] : sz == 2 ? u16(o) : u32(o);
        }
        return v;
    }
};

// TIFF flavour of LZW: MSB-first codes of 9..12 bits, 256 = clear, 257 = end, width grows one code early.
void lzw_decode(const uint8_t *src, size_t n, std::vector<uint8_t> *out, size_t want, const std::string &name) {
    struct Entry { uint16_t prev; uint16_t len; uint8_t first, last; };
    std::vector<Entry> tab(4096);
    for (int k = 0; k < 256; k++) tab[k] = Entry{0xFFFF, 1, (uint8_t)k, (uint8_t)k};
    int nbits = 9, next = 258, old = -1;
    uint32_t acc = 0;
    int have = 0;
    size_t pos = 0;
    out->clear();
    out->reserve(want);
    auto emit = [&](int code) {
        size_t len = tab[code].len, at = out->size();
        out->resize(at + len);
        for (int c = code; len; c = tab[c].prev) (*out)[at + --len] = tab[c].last;
    };
    while (out->size() < want) {
        while (have < nbits) {
            if (pos >= n) return;                              // stream ended without an EOI code: keep what we have
            acc = (acc << 8) | src[pos++];
            have += 8;
        }
        int code = (acc >> (have - nbits)) & ((1 << nbits) - 1);
        have -= nbits;
        if (code == 257) return;
        if (code == 256) { nbits = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code > 255) throw Error("corrupt LZW stream: " + name, RGBDSLAM_E_ARG);
            emit(code);
            old = code;
            continue;
        }
        if (code > next || next > 4095) throw Error("corrupt LZW stream: " + name, RGBDSLAM_E_ARG);
        const uint8_t first = code < next ? tab[code].first : tab[old].first;
        tab[next] = Entry{(uint16_t)old, (uint16_t)(tab[old].len + 1), tab[old].first, first};
        next++;
        emit(code);
        old = code;
        if (next == (1 << nbits) - 1 && nbits < 12) nbits++;
    }
}

void packbits_decode(const uint8_t *src, size_t n, std::vector<uint8_t> *out, size_t want) {
    out->clear();
    for (size_t p = 0; p < n && out->size() < want;) {
        int8_t h = (int8_t)src[p++];
        if (h >= 0) {
            size_t k = (size_t)h + 1;
            if (p + k > n) k = n - p;
            out->insert(out->end(), src + p, src + p + k);
            p += k;
        } else if (h != -128) {
            if (p >= n) break;
            out->insert(out->end(), (size_t)(1 - h), src[p++]);
        }
    }
}

// encoder: the string table is a hash of (prefix code, next byte) pairs, reset when 4094 codes are in use
void lzw_encode(const uint8_t *src, size_t n, std::vector<uint8_t> *out) {
    const int HSIZE = 9001;
    std::vector<int32_t> hkey(HSIZE, -1);
    std::vector<uint16_t> hval(HSIZE);
    int nbits = 9, next = 258;
    uint32_t acc = 0;
    int have = 0;
    auto put = [&](int code) {
        acc = (acc << nbits) | (uint32_t)code;
        have += nbits;
        while (have >= 8) { out->push_back((uint8_t)(acc >> (have - 8))); have -= 8; }
    };
    put(256);
    if (n) {
        int cur = src[0];
        for (size_t k = 1; k < n; k++) {
            const int c = src[k];
            const int32_t key = (cur << 8) | c;
            int h = (int)(((uint32_t)key * 2654435761u) % HSIZE);
            while (hkey[h] != -1 && hkey[h] != key) h = (h + 1) % HSIZE;
            if (hkey[h] == key) { cur = hval[h]; continue; }
            put(cur);
            hkey[h] = key; hval[h] = (uint16_t)next++;
            cur = c;
            if (next == 4094) {
                put(256);
                std::fill(hkey.begin(), hkey.end(), -1);
                nbits = 9; next = 258;
            } else if (next > (1 << nbits) - 1) {
                nbits++;
            }
        }
        put(cur);
        // the decoder's table runs one entry behind: the width it expects for the end code follows its own count
        if (next + 1 > (1 << nbits) - 1 && nbits < 12) nbits++;
    }
    put(257);
    if (have) out->push_back((uint8_t)(acc << (8 - have)));
}

}  // namespace

Image read_tiff(const std::string &filename) {
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) throw Error("cannot open " + filename, RGBDSLAM_E_ARG);
    std::vector<uint8_t> file;
    uint8_t tmp[1 << 16];
    size_t k;
    while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) file.insert(file.end(), tmp, tmp + k);
    fclose(f);
    if (file.size() < 8 || !((file[0] == 'I' && file[1] == 'I') || (file[0] == 'M' && file[1] == 'M')))
        throw Error("not a TIFF file: " + filename, RGBDSLAM_E_ARG);
    Reader r{file, file[0] == 'M', filename};
    if (r.u16(2) != 42) throw Error("not a TIFF file: " + filename, RGBDSLAM_E_ARG);
    size_t ifd = r.u32(4);
    int nent = r.u16(ifd);
    uint32_t W = 0, H = 0, comp = 1, photo = 1, spp = 1, rps = 0xFFFFFFFFu, planar = 1, pred = 1, fmt = 1;
    std::vector<uint32_t> bps(1, 1), offs, counts;
    for (int e = 0; e < nent; e++) {
        size_t ent = ifd + 2 + 12 * (size_t)e;
        uint16_t tag = r.u16(ent);
        switch (tag) {
        case 256: W = r.values(ent).at(0); break;
        case 257: H = r.values(ent).at(0); break;
        case 258: bps = r.values(ent); break;
        case 259: comp = r.values(ent).at(0); break;
        case 262: photo = r.values(ent).at(0); break;
        case 273: offs = r.values(ent); break;
        case 277: spp = r.values(ent).at(0); break;
        case 278: rps = r.values(ent).at(0); break;
        case 279: counts = r.values(ent); break;
        case 284: planar = r.values(ent).at(0); break;
        case 317: pred = r.values(ent).at(0); break;
        case 339: fmt = r.values(ent).at(0); break;
        case 322: case 323: case 324: case 325: throw Error("tiled TIFF is not supported: " + filename, RGBDSLAM_E_ARG);
        default: break;
        }
    }
    if (!W || !H || W > 65535u || H > 65535u || bps.empty()) throw Error("bad TIFF header: " + filename, RGBDSLAM_E_ARG);
    for (uint32_t b : bps) if (b != bps[0]) throw Error("mixed TIFF sample widths: " + filename, RGBDSLAM_E_ARG);
    const uint32_t bits = bps[0];
    if (!(bits == 8 || bits == 16) || !(spp == 1 || spp == 3 || spp == 4) || planar != 1 || fmt != 1 || photo > 2 ||
        !(comp == 1 || comp == 5 || comp == 32773) || !(pred == 1 || pred == 2) || offs.empty() || offs.size() != counts.size())
        throw Error("unsupported TIFF layout: " + filename, RGBDSLAM_E_ARG);
    if (rps == 0 || rps > H) rps = H;
    if ((size_t)offs.size() != ((size_t)H + rps - 1) / rps) throw Error("TIFF strip table does not match the image: " + filename, RGBDSLAM_E_ARG);

    Image img;
    img.rows = (int)H; img.cols = (int)W; img.channels = (int)spp; img.bits = (int)bits;
    const size_t rowb = img.row_bytes(), bpsb = bits / 8;
    img.data.assign((size_t)H * rowb, 0);
    std::vector<uint8_t> strip;
    for (size_t s = 0; s < offs.size(); s++) {
        const size_t row0 = s * rps, nrows = row0 + rps > H ? H - row0 : rps, want = nrows * rowb;
        r.need(offs[s], counts[s]);
        const uint8_t *src = &file[offs[s]];
        if (comp == 1) strip.assign(src, src + (counts[s] < want ? counts[s] : want));
        else if (comp == 5) lzw_decode(src, counts[s], &strip, want, filename);
        else packbits_decode(src, counts[s], &strip, want);
        if (strip.size() < want) throw Error("TIFF strip shorter than its rows: " + filename, RGBDSLAM_E_ARG);
        for (size_t y = 0; y < nrows; y++) {
            const uint8_t *in = &strip[y * rowb];
            uint8_t *o = &img.data[(row0 + y) * rowb];
            // file byte order -> host values, undo horizontal differencing, RGB -> BGR
            std::vector<uint16_t> v((size_t)W * spp);
            for (size_t i = 0; i < v.size(); i++)
                v[i] = bpsb == 1 ? in[i] : (r.big ? (uint16_t)((in[2 * i] << 8) | in[2 * i + 1]) : (uint16_t)((in[2 * i + 1] << 8) | in[2 * i]));
            if (pred == 2)
                for (size_t i = spp; i < v.size(); i++) v[i] = (uint16_t)((v[i] + v[i - spp]) & (bpsb == 1 ? 0xFF : 0xFFFF));
            for (size_t x = 0; x < W; x++)
                for (uint32_t c = 0; c < spp; c++) {
                    uint32_t sc = (spp >= 3 && c < 3) ? 2 - c : c;
                    uint16_t val = v[x * spp + sc];
                    if (photo == 0) val = (uint16_t)((bpsb == 1 ? 0xFF : 0xFFFF) - val);   // white-is-zero
                    if (bpsb == 1) o[x * spp + c] = (uint8_t)val; else memcpy(o + 2 * (x * spp + c), &val, 2);
                }
        }
    }
    return img;
}

void write_tiff(const std::string &filename, const Image &img, bool lzw) {
    if (!(img.bits == 8 || img.bits == 16) || !(img.channels == 1 || img.channels == 3 || img.channels == 4) || img.rows <= 0 ||
        img.cols <= 0 || img.data.size() != (size_t)img.rows * img.row_bytes())
        throw Error("write_tiff: unsupported image", RGBDSLAM_E_ARG);
    const uint32_t W = img.cols, H = img.rows, spp = img.channels, bpsb = img.bits / 8;
    const size_t rowb = img.row_bytes();
    uint32_t rps = (uint32_t)((1u << 13) / rowb);              // ~8 KB strips like cv::imwrite's TIFF encoder
    if (rps < 1) rps = 1;
    if (rps > H) rps = H;
    const uint32_t nstrips = (H + rps - 1) / rps;

    std::vector<uint8_t> body;                                 // strip data, placed right after the 8-byte header
    std::vector<uint32_t> offs(nstrips), counts(nstrips);
    std::vector<uint8_t> raw;
    for (uint32_t s = 0; s < nstrips; s++) {
        const uint32_t row0 = s * rps, nrows = row0 + rps > H ? H - row0 : rps;
        raw.assign((size_t)nrows * rowb, 0);
        for (uint32_t y = 0; y < nrows; y++) {
            const uint8_t *in = &img.data[(size_t)(row0 + y) * rowb];
            std::vector<uint16_t> v((size_t)W * spp);
            for (size_t x = 0; x < W; x++)
                for (uint32_t c = 0; c < spp; c++) {
                    uint32_t sc = (spp >= 3 && c < 3) ? 2 - c : c;                          // BGR -> RGB
                    uint16_t val;
                    if (bpsb == 1) val = in[x * spp + sc]; else memcpy(&val, in + 2 * (x * spp + sc), 2);
                    v[x * spp + c] = val;
                }
            if (lzw)                                                                         // horizontal predictor
                for (size_t i = v.size() - 1; i >= spp; i--) v[i] = (uint16_t)((v[i] - v[i - spp]) & (bpsb == 1 ? 0xFF : 0xFFFF));
            uint8_t *o = &raw[(size_t)y * rowb];
            for (size_t i = 0; i < v.size(); i++) {
                if (bpsb == 1) o[i] = (uint8_t)v[i];
                else { o[2 * i] = (uint8_t)(v[i] & 0xFF); o[2 * i + 1] = (uint8_t)(v[i] >> 8); }   // little-endian file
            }
        }
        offs[s] = 8 + (uint32_t)body.size();
        if (lzw) { std::vector<uint8_t> z; lzw_encode(raw.data(), raw.size(), &z); body.insert(body.end(), z.begin(), z.end()); counts[s] = (uint32_t)z.size(); }
        else { body.insert(body.end(), raw.begin(), raw.end()); counts[s] = (uint32_t)raw.size(); }
        if (body.size() & 1) body.push_back(0);                // word alignment
    }

    std::vector<uint8_t> out;
    auto p16 = [&](uint16_t v) { out.push_back(v & 0xFF); out.push_back(v >> 8); };
    auto p32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) out.push_back((v >> (8 * k)) & 0xFF); };
    out.push_back('I'); out.push_back('I'); p16(42);
    // layout: header | strips | BitsPerSample array | StripOffsets | StripByteCounts | IFD
    const uint32_t extra0 = 8 + (uint32_t)body.size();
    const uint32_t bps_off = extra0, offs_off = bps_off + 2 * spp + ((2 * spp) & 2), counts_off = offs_off + 4 * nstrips,
                   ifd_off = counts_off + 4 * nstrips;
    p32(ifd_off);
    out.insert(out.end(), body.begin(), body.end());
    for (uint32_t c = 0; c < spp; c++) p16((uint16_t)img.bits);
    if ((2 * spp) & 2) p16(0);
    for (uint32_t v : offs) p32(v);
    for (uint32_t v : counts) p32(v);
    struct Ent { uint16_t tag, type; uint32_t cnt, val; };
    std::vector<Ent> ents = {
        {256, 4, 1, W}, {257, 4, 1, H},
        {258, 3, spp, spp == 1 ? (uint32_t)img.bits : bps_off},
        {259, 3, 1, lzw ? 5u : 1u},
        {262, 3, 1, spp == 1 ? 1u : 2u},
        {273, 4, nstrips, nstrips == 1 ? offs[0] : offs_off},
        {277, 3, 1, spp},
        {278, 4, 1, rps},
        {279, 4, nstrips, nstrips == 1 ? counts[0] : counts_off},
        {284, 3, 1, 1},
    };
    if (lzw) ents.push_back({317, 3, 1, 2});
    if (spp == 4) ents.push_back({338, 3, 1, 2});              // ExtraSamples: unassociated alpha
    ents.push_back({339, 3, 1, 1});                            // unsigned integer samples
    p16((uint16_t)ents.size());
    for (const Ent &e : ents) {
        p16(e.tag); p16(e.type); p32(e.cnt);
        if (e.type == 3 && e.cnt == 1) { p16((uint16_t)e.val); p16(0); } else p32(e.val);
    }
    p32(0);
    FILE *fo = fopen(filename.c_str(), "wb");
    if (!fo) throw Error("cannot create " + filename, RGBDSLAM_E_ARG);
    fwrite(out.data(), 1, out.size(), fo);
    fclose(fo);
}

}  // namespace slam
