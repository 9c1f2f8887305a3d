/* Sanitizer harness for the CPU oracle (test infrastructure): the oracle's display() loop (orc_slam_step: every
 * restated kernel of SURVEY 8a, the Gauss-Newton loop, the software rasteriser, the Bump fusion) incl. RG::Finish's plane
 * detection (detect_oracle.c) over a short recorded sequence, built as a plain executable with -fsanitize=address,undefined.
 *   usage: oracle_san <frames.bin>    frames.bin = int32 n, m, frames; float calib[11]; then per frame u16 depth[n*m], u8 bgr[n*m*3]
 * Prints the final pose as hex words (the caller compares them with the regular build's). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../rgbdslam_oracle.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s frames.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t hdr[3];
    float calib[11];
    if (fread(hdr, 4, 3, f) != 3 || fread(calib, 4, 11, f) != 11) { fprintf(stderr, "short header\n"); return 2; }
    const int n = hdr[0], m = hdr[1], frames = hdr[2];
    if (n < 8 || m < 8 || n > 4096 || m > 4096 || frames < 1 || frames > 10000) { fprintf(stderr, "bad header\n"); return 2; }
    const size_t P = (size_t)n * m;
    uint16_t *depth = (uint16_t *)malloc(2 * P);
    uint8_t *bgr = (uint8_t *)malloc(3 * P);
    orc_icp_params prm = {3, {3, 2, 1}, 0.05f, 0.3420201433256687f, 1e-6f, 1e-6f};
    orc_slam *s = orc_slam_create(n, m, calib, &prm, 50, 128, 128, 0.5f, 10.0f);
    orc_slam_set_finish(s, 1, 3, 8, 200, 100);        /* first detection after 3 frames, a periodic cycle every 8 */
    int tracked = 0;
    for (int k = 0; k < frames; k++) {
        if (fread(depth, 2, P, f) != P || fread(bgr, 1, 3 * P, f) != 3 * P) { fprintf(stderr, "short frame %d\n", k); return 2; }
        tracked += orc_slam_step(s, depth, bgr) != 0;
    }
    float R[9], t[3];
    orc_slam_get_pose(s, R, t);
    uint32_t w[12];
    memcpy(w, R, 36); memcpy(w + 9, t, 12);
    printf("tracked %d planes %d detections %d pose", tracked, orc_slam_nb_planes(s), orc_slam_finish_detections(s));
    for (int k = 0; k < 12; k++) printf(" %08x", w[k]);
    printf("\n");
    size_t nb = 0;
    const uint8_t *lab = (const uint8_t *)orc_slam_buffer(s, "Label", &nb);
    unsigned long sum = 0;
    for (size_t k = 0; k < nb; k++) sum += lab[k];
    printf("label_sum %lu\n", sum);
    orc_slam_destroy(s);
    free(depth); free(bgr);
    fclose(f);
    return 0;
}
