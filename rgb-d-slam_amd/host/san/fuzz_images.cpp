// Sanitizer harness for the host-side file parsers: slam::read_png (PNG over zlib incl. Adam7 and all filters)
// and slam::read_tiff (baseline TIFF: strips, LZW + predictor, PackBits, both byte orders) parse UNTRUSTED files -- the datasets
// RG::LoadFrame names (D1/RG.cpp:105-139) and the model dumps DemoStep2 reads (D2/Model3D.cpp:1050-1209).  Built with
// -fsanitize=address,undefined as a plain executable (no GPU, no preload); every seed file is truncated at every length
// and mutated (bit flips, bytes set to 0x00 / 0xFF / 0x7F, 16- and 32-bit fields blown up): each input must either decode or
// throw slam::Error / std::exception -- never read or write out of bounds, never overflow, never hang.
//   usage: fuzz_images <work file> <seed files...>      (the extension of a seed selects the parser)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../dataset_io.hpp"

static std::vector<unsigned char> slurp(const char *name) {
    std::vector<unsigned char> v;
    FILE *f = fopen(name, "rb");
    if (!f) return v;
    unsigned char buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned int rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned int)(rng_state >> 16); }

static long decoded = 0, rejected = 0;
static void try_one(const std::string &work, const std::vector<unsigned char> &bytes, bool tiff) {
    FILE *f = fopen(work.c_str(), "wb");
    if (!f) { perror("work file"); exit(2); }
    if (!bytes.empty()) fwrite(bytes.data(), 1, bytes.size(), f);
    fclose(f);
    try {
        slam::Image img = tiff ? slam::read_tiff(work) : slam::read_png(work);
        // a decoded image must be self-consistent: the sample buffer is what the header promises
        if (img.rows < 0 || img.cols < 0 || img.data.size() != (size_t)img.rows * img.row_bytes()) { fprintf(stderr, "inconsistent image\n"); exit(3); }
        volatile unsigned long long sum = 0;
        for (size_t k = 0; k < img.data.size(); k += 97) sum += img.data[k];      // touch what was decoded
        decoded++;
    } catch (const std::exception &) { rejected++; }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <work file> <seeds...>\n", argv[0]); return 2; }
    const std::string work = argv[1];
    for (int a = 2; a < argc; a++) {
        const std::string name = argv[a];
        const bool tiff = name.size() > 4 && (name.rfind(".tif") != std::string::npos);
        const std::vector<unsigned char> seed = slurp(argv[a]);
        if (seed.empty()) { fprintf(stderr, "empty seed %s\n", argv[a]); return 2; }
        try_one(work, seed, tiff);
        if (decoded == 0) { fprintf(stderr, "seed %s does not decode\n", argv[a]); return 4; }
        // every prefix (every length for small files, a stride for larger ones)
        const size_t stride = seed.size() > 4096 ? seed.size() / 2048 : 1;
        for (size_t len = 0; len < seed.size(); len += stride) try_one(work, std::vector<unsigned char>(seed.begin(), seed.begin() + len), tiff);
        // mutations, biased towards the headers (first 256 bytes) where sizes, offsets and counts live
        for (int it = 0; it < 1500; it++) {
            std::vector<unsigned char> m = seed;
            const int edits = 1 + (int)(rnd() % 4);
            for (int e = 0; e < edits; e++) {
                const size_t span = (rnd() & 1) ? (m.size() < 256 ? m.size() : 256) : m.size();
                const size_t pos = rnd() % span;
                switch (rnd() % 6) {
                case 0: m[pos] ^= (unsigned char)(1u << (rnd() % 8)); break;
                case 1: m[pos] = 0x00; break;
                case 2: m[pos] = 0xFF; break;
                case 3: m[pos] = 0x7F; break;
                case 4: for (size_t k = pos; k < pos + 4 && k < m.size(); k++) m[k] = 0xFF; break;       // a huge 32-bit field
                default: for (size_t k = pos; k < pos + 2 && k < m.size(); k++) m[k] = (unsigned char)rnd(); break;
                }
            }
            if ((rnd() % 8) == 0) m.resize(rnd() % (m.size() + 1));
            try_one(work, m, tiff);
        }
    }
    printf("fuzz_images: %ld inputs decoded, %ld rejected, no sanitizer report\n", decoded, rejected);
    return 0;
}
