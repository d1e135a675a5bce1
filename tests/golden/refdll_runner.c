/*
 * tests/golden/refdll_runner.c -- generator tooling for tests/golden/refdll_grid_index.npz (not product, not oracle).
 *
 * Runs two functions of the REFERENCE ITSELF: GMSMatcher::getGridIndexLeft (RVA 0x47bc0) and
 * GMSMatcher::getGridIndexRight (RVA 0x47d60) straight out of the reference's
 * SfM-GMS/bin/opencv_xfeatures2d452.dll. Both are leaf functions (no calls, no imports; the only data they touch
 * is `this` and, RIP-relative, the 0.5 constant in .rdata), so the PE image is simply mapped section by section
 * into executable memory and they are called with the Microsoft x64 convention on a zeroed stand-in object that
 * holds nothing but the four grid dimensions they read ([this+0x50..0x5c], see the disassembly in SURVEY.md 8a).
 * Nothing of the DLL is copied into the repository: only the inputs and the integers it returns are kept.
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin>
 *   in.bin : int32 n, then n x (float nx, float ny)
 *   out.bin: n x 9 int32: left cell for grid types 1..4, right cell for right grids 20, 10, 14, 28, 40
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

typedef int(__attribute__((ms_abi)) * left_fn)(void* self, const float* pt, int type);
typedef int(__attribute__((ms_abi)) * right_fn)(void* self, const float* pt);

static uint32_t rd32(const unsigned char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const unsigned char* p) { uint16_t v; memcpy(&v, p, 2); return v; }

int main(int argc, char** argv)
{
    if (argc != 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* file = malloc((size_t)fsz);
    if (fread(file, 1, (size_t)fsz, f) != (size_t)fsz) return 3;
    fclose(f);

    const uint32_t pe = rd32(file + 0x3c);
    if (memcmp(file + pe, "PE\0\0", 4) != 0) return 4;
    const int nsec = rd16(file + pe + 6);
    const int optsz = rd16(file + pe + 20);
    const unsigned char* opt = file + pe + 24;
    if (rd16(opt) != 0x20b) return 4; /* PE32+ */
    const uint32_t size_image = rd32(opt + 56), size_headers = rd32(opt + 60);
    unsigned char* img = mmap(NULL, size_image, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (img == MAP_FAILED) return 5;
    memcpy(img, file, size_headers);
    const unsigned char* sec = opt + optsz;
    for (int i = 0; i < nsec; i++, sec += 40) {
        const uint32_t vsize = rd32(sec + 8), va = rd32(sec + 12), rsize = rd32(sec + 16), rptr = rd32(sec + 20);
        const uint32_t n = rsize < vsize ? rsize : vsize;
        if ((uint64_t)va + n > size_image || (uint64_t)rptr + n > (uint64_t)fsz) return 4;
        memcpy(img + va, file + rptr, n);
    }
    left_fn get_left = (left_fn)(img + 0x47bc0);
    right_fn get_right = (right_fn)(img + 0x47d60);

    FILE* in = fopen(argv[2], "rb");
    FILE* out = fopen(argv[3], "wb");
    if (!in || !out) return 6;
    int32_t n = 0;
    if (fread(&n, 4, 1, in) != 1) return 6;
    static const int right_dims[5] = {20, 10, 14, 28, 40};
    unsigned char self[0x200];
    for (int i = 0; i < n; i++) {
        float pt[2];
        if (fread(pt, 4, 2, in) != 2) return 6;
        int32_t res[9];
        memset(self, 0, sizeof self);
        *(int32_t*)(self + 0x50) = 20; /* mGridSizeLeft */
        *(int32_t*)(self + 0x54) = 20;
        for (int t = 1; t <= 4; t++) res[t - 1] = get_left(self, pt, t);
        for (int s = 0; s < 5; s++) {
            *(int32_t*)(self + 0x58) = right_dims[s]; /* mGridSizeRight */
            *(int32_t*)(self + 0x5c) = right_dims[s];
            res[4 + s] = get_right(self, pt);
        }
        fwrite(res, 4, 9, out);
    }
    fclose(in);
    fclose(out);
    return 0;
}
