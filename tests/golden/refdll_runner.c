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
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> assign
 *   A third function, GMSMatcher::assignMatchPairs (RVA 0x47880; its only call is getGridIndexLeft), driven for grid
 *   types 1..4 the way GMSMatcher::run does (motion matrix and per-cell counts zeroed before each type). The object it
 *   works on is laid out by hand from the disassembly: mvP1.begin @+0x00, mvP2.begin @+0x18, mvMatches.begin @+0x30,
 *   mNumberMatches @+0x48, grid sizes @+0x50..0x5c, Mat data pointer @+0x78, pointer to the row step @+0xb0,
 *   mNumberPointsInPerCellLeft.begin @+0xc8, mvMatchPairs.begin @+0xf8.
 *   in.bin : int32 wr, hr, n1, n2, m; n1 x float2 (normalised), n2 x float2, m x (int32 query, int32 train)
 *   out.bin: per grid type: m x 2 int32 (mvMatchPairs), 400 int32 (per-cell counts), 400 * wr * hr int32 (motion)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> verify
 *   The body of GMSMatcher::verifyCellPairs (RVA 0x48d10) for one left cell at a time. The function's first act per
 *   cell is cv::sum(row) through opencv_core (an import this image does not have), so it cannot be called from its
 *   entry; but everything after that test -- the arg-max scan, the nine neighbour sums through the rotation pattern,
 *   sqrt(T / n) * factor and the '>' -- is self-contained code from RVA 0x48e12 on. The trampoline below rebuilds the
 *   function's own prologue state (frame, saved registers, security cookie, rbx = this, rsi = cell, r13d = cell + 1,
 *   rbp = 9 * rotation, xmm7 = 0) and jumps there; with mGridNumberLeft set to cell + 1 the loop ends after that one
 *   cell and the function's own epilogue returns. The "row sum == 0 -> -1" branch is the driver's (cells with no
 *   match never reach the fragment). Object layout from the disassembly: mGridNumberLeft @+0x60, mGridNumberRight
 *   @+0x64, motion data @+0x78 / row-step pointer @+0xb0, per-cell counts @+0xc8, mCellPairs.begin @+0xe0, left
 *   neighbour table data @+0x140 / step pointer @+0x178, right neighbour table data @+0x1a0 / step pointer @+0x1d8,
 *   mThresholdFactor @+0x1f0. The two neighbour tables are filled by the DLL's own GMSMatcher::initalizeNeighbors (see "nb9").
 *   in.bin : int32 wr, hr; double factor; 400 int32 counts; 400 * wr * hr int32 motion
 *   out.bin: 8 x 400 int32 mCellPairs (rotation types 1..8)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> nb9
 *   GMSMatcher::initalizeNeighbors (RVA 0x48180) and, through it, GMSMatcher::getNB9 (RVA 0x48030), run out of the DLL. getNB9
 *   builds a std::vector<int>(9, -1): operator new (RVA 0x81650) loops on the CRT's malloc through the import slot at RVA
 *   0x901b0, the vector's release ends in free through the slot at RVA 0x901b8 (api-ms-win-crt-heap-l1-1-0.dll: _callnewh
 *   0x901a8, malloc 0x901b0, free 0x901b8). The image's imports are unresolved here, so those two slots are pointed at
 *   ms_abi wrappers of this process's malloc / free -- an allocator, no arithmetic. initalizeNeighbors(this, Mat& neighbor,
 *   const Size& grid) reads neighbor.rows @+0x08, neighbor.data @+0x10 and the pointer to the row step @+0x48.
 *   in.bin : int32 n, then n x (int32 width, int32 height)
 *   out.bin: per grid: width * height x 9 int32 (the neighbour table)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> normalize
 *   GMSMatcher::normalizePoints (RVA 0x48420): (this, const std::vector<cv::KeyPoint>& kp, const cv::Size& size,
 *   std::vector<cv::Point2f>& npts). It resizes npts to kp.size() first; handed a vector that already has that size it
 *   allocates nothing and is a leaf. Vectors are {begin, end, capacity-end} pointer triples; KeyPoint stride 0x1c.
 *   in.bin : int32 n, width, height; n x 28-byte cv::KeyPoint records
 *   out.bin: n x (float nx, float ny)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> setscale
 *   The head of GMSMatcher::setScale (RVA 0x48c10): mGridSizeRight = cvRound(mGridSizeLeft * mScaleRatios[scale]) per axis
 *   (cvtdq2pd, mulsd, cvtsd2si), mGridNumberRight = their product -- everything up to its first import, cv::Mat::zeros(rows,
 *   cols, type) through the slot at RVA 0x903f8. That slot is pointed at a function of this file that records the three
 *   arguments and takes control back (longjmp): nothing of opencv_core is emulated, the call simply ends the experiment.
 *   Two of the five mScaleRatios entries (.data RVA 0x2c5018 / 0x2c5020) are written by a static initialiser of the DLL
 *   (RVA 0x10b0, a leaf: sqrtpd of the constant 2.0, then 1.0 / that); it is run first, as the loader would.
 *   in.bin : int32 left_w, left_h
 *   out.bin: 5 doubles (mScaleRatios after the initialiser), then per scale 0..4: int32 right_w, right_h, n_right, and the
 *            rows, cols, type handed to cv::Mat::zeros (the right neighbour table: n_right x 9, CV_32SC1 = 4)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> mark
 *   The tail of GMSMatcher::run (RVA 0x48630) from RVA 0x48acd on: the loop that marks the inliers of one grid type
 *   (pairs[i].first >= 0 && mCellPairs[pairs[i].first] == pairs[i].second -> bts into mvbInlierMask, RVA 0x48ae0-0x48b26), the
 *   "next grid type" test, the count of the mask's set bits (RVA 0x48b46-0x48bd4) and the function's own epilogue. A trampoline
 *   rebuilds run()'s frame (seven pushes, 0x1f0 bytes, cookie, [rsp+0x50] = &mvbInlierMask, rdi = this, r12 = &mCellPairs) and
 *   enters with r13d = 4, so that the code leaves its grid-type loop after this one marking pass and returns the count; the
 *   driver calls it once per grid type on the state of that type, the mask accumulating as in run(). Object layout:
 *   mNumberMatches @+0x48, mCellPairs.begin @+0xe0, mvMatchPairs.begin @+0xf8, mvbInlierMask {word begin, end, cap, bit size} @+0x110.
 *   in.bin : int32 m; per grid type 1..4: m x 2 int32 (mvMatchPairs), 400 int32 (mCellPairs)
 *   out.bin: per grid type: (m + 31) / 32 uint32 (the mask's words after the pass), int32 (the count the code returns)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> select
 *   GMSMatcher::getInlierMask (RVA 0x47dc0) itself, whole, for any flag combination. It reaches setScale (RVA 0x48c10) and run
 *   (RVA 0x48630) through eight call rel32 instructions; those eight displacements are re-pointed at two functions of this file
 *   that PLAY BACK a script: setScale does nothing but log its argument, run logs its argument, puts the scripted mask of the
 *   (scale, rotation) at hand into mvbInlierMask (this + 0x110) and returns the scripted count. Everything else -- the loop
 *   nest and its order, the strict '>' against the best count so far, the copy of the mask (std::vector<bool> assignment, inline
 *   and through RVA 0x46fb0, with operator new / delete / memmove reached through the import slots 0x901b0 / 0x901b8 / 0x90188,
 *   pointed at this process's malloc / free / memmove) and the value returned -- is the DLL's own code.
 *   in.bin : int32 m, with_rotation, with_scale; 40 int32 counts [scale][rotation]; 40 x (m + 31) / 32 uint32 masks
 *   out.bin: int32 returned count; int64 size in bits of the caller's mask afterwards; (m + 31) / 32 uint32 its words (zeros when it
 *            stayed empty); int32 number of calls; 96 int32 call log (100 + s = setScale(s), r = run(r), -1 = unused)
 *
 * usage: refdll_runner <dll> <in.bin> <out.bin> chain
 *   matchGMS's whole computation as a CHAIN OF THE DLL'S OWN PIECES on one pair: normalizePoints (RVA 0x48420) for both images;
 *   getInlierMask (RVA 0x47dc0) with its eight calls re-pointed as above -- but now the two targets run the DLL too: "setScale"
 *   runs the head of the real setScale (right grid and cell count, up to its first import) and fills the right neighbour table by
 *   initalizeNeighbors (RVA 0x48180); "run" does, per grid type, assignMatchPairs (RVA 0x47880), the body of verifyCellPairs (RVA
 *   0x48e12, per cell) and the marking / counting tail of run (RVA 0x48acd). What this file supplies between the pieces is storage
 *   and the four things the DLL does through opencv_core / std::vector members: zeroing the motion matrix (Mat::setTo(0)), the
 *   three assign() fills (mask false, pairs (0, 0), cell pairs -1, counts 0), the allocation of the matrices (Mat::zeros), and the
 *   "row sum == 0" test in front of a cell's verification (cv::sum) -- no arithmetic of the algorithm. convertMatches is the copy
 *   of (queryIdx, trainIdx) the driver lays out.
 *   in.bin : int32 n1, n2, m, w1, h1, w2, h2, with_rotation, with_scale; double factor; n1 + n2 28-byte cv::KeyPoint records;
 *            m x (int32 queryIdx, int32 trainIdx)
 *   out.bin: int32 returned count; int64 size in bits of the mask; (m + 31) / 32 uint32 its words
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <setjmp.h>
#include <sys/mman.h>

/* see "verify" above: emulates verifyCellPairs' prologue, then jumps into its body */
void __attribute__((ms_abi)) verify_fragment(void* self, int rotation_type, long long cell, void* body, void* cookie);
__asm__(".intel_syntax noprefix\n"
        ".globl verify_fragment\n"
        "verify_fragment:\n"
        "  mov rax, rsp\n"
        "  mov [rax+0x10], rbx\n"
        "  mov [rax+0x18], rbp\n"
        "  mov [rax+0x20], rsi\n"
        "  push rdi\n  push r12\n  push r13\n  push r14\n  push r15\n"
        "  sub rsp, 0x100\n"
        "  movaps [rax-0x38], xmm6\n"
        "  movaps [rax-0x48], xmm7\n"
        "  mov r10, [rax+0x28]\n"          /* fifth argument: address of the image's security cookie */
        "  mov r10, [r10]\n"
        "  xor r10, rsp\n"
        "  mov [rsp+0xd0], r10\n"
        "  mov rbx, rcx\n"
        "  xor r15d, r15d\n"
        "  mov [rsp+0x20], r15d\n"
        "  movsxd rax, edx\n"
        "  lea rbp, [rax+8*rax]\n"
        "  mov rsi, r8\n"
        "  lea r13d, [r8d+1]\n"
        "  xorps xmm7, xmm7\n"
        "  jmp r9\n"
        ".att_syntax prefix\n");

/* see "setscale" above: what setScale hands to its first import, and the way back */
static jmp_buf g_back;
static int32_t g_zeros_args[3];
static void __attribute__((ms_abi)) stop_at_mat_zeros(void* ret_slot, int rows, int cols, int type)
{
    (void)ret_slot;
    g_zeros_args[0] = rows;
    g_zeros_args[1] = cols;
    g_zeros_args[2] = type;
    longjmp(g_back, 1);
}

/* the two CRT imports getNB9 needs (see "nb9" above) */
static void* __attribute__((ms_abi)) crt_malloc(size_t n) { return malloc(n); }
static void __attribute__((ms_abi)) crt_free(void* p) { free(p); }
static void* __attribute__((ms_abi)) crt_memmove(void* d, const void* s, size_t n) { return memmove(d, s, n); }
static void* __attribute__((ms_abi)) crt_memset(void* d, int c, size_t n) { return memset(d, c, n); }
typedef void(__attribute__((ms_abi)) * init_nb_fn)(void* self, void* mat, const int32_t* grid_size);

/* see "mark" above: emulates run()'s prologue, then jumps behind its call of verifyCellPairs with "this is the last grid type" */
int __attribute__((ms_abi)) mark_fragment(void* self, void* body, void* cookie);
__asm__(".intel_syntax noprefix\n"
        ".globl mark_fragment\n"
        "mark_fragment:\n"
        "  mov [rsp+0x18], rbx\n"
        "  push rbp\n  push rsi\n  push rdi\n  push r12\n  push r13\n  push r14\n  push r15\n"
        "  sub rsp, 0x1f0\n"
        "  movaps [rsp+0x1e0], xmm6\n"
        "  mov rax, [r8]\n"               /* third argument: address of the image's security cookie */
        "  xor rax, rsp\n"
        "  mov [rsp+0x1d0], rax\n"
        "  mov rdi, rcx\n"
        "  lea rcx, [rdi+0x110]\n"
        "  mov [rsp+0x50], rcx\n"
        "  lea r12, [rdi+0xe0]\n"
        "  mov r13d, 4\n"
        "  jmp rdx\n"
        ".att_syntax prefix\n");

/* GMSMatcher::initalizeNeighbors out of the image: fills table[w * h][9] */
static void dll_neighbors(unsigned char* img, int32_t* table, int w, int h)
{
    *(void**)(img + 0x901b0) = (void*)crt_malloc;
    *(void**)(img + 0x901b8) = (void*)crt_free;
    uint64_t step = 36;
    unsigned char mat[0x60];
    memset(mat, 0, sizeof mat);
    *(int32_t*)(mat + 0x08) = w * h;      /* rows */
    *(int32_t*)(mat + 0x0c) = 9;          /* cols */
    *(void**)(mat + 0x10) = table;        /* data */
    *(void**)(mat + 0x48) = &step;        /* step.p */
    const int32_t size[2] = {w, h};
    unsigned char self[0x200];
    memset(self, 0, sizeof self);
    ((init_nb_fn)(img + 0x48180))(self, mat, size);
}


/* ---- "select" and "chain": getInlierMask with its calls of setScale / run re-pointed at this file ------------------------- */
static unsigned char* g_img;
static int g_chain;                  /* 0: play the script back, 1: run the DLL's own pieces */
static int32_t g_m, g_words;
static const int32_t* g_counts;      /* [5][8] */
static const uint32_t* g_masks;      /* [5][8][words] */
static int32_t g_scale;              /* what setScale was last called with */
static int32_t g_log[96], g_n_log;
static uint32_t* g_mask_words;       /* storage of mvbInlierMask (this + 0x110) */
/* chain: the object's storage */
static int32_t *g_motion, *g_nleft, *g_cell_pairs, *g_pairs, *g_nb_left, *g_nb_right;
static uint64_t g_step_motion, g_step_nb = 36;
static uint32_t g_size_image;

static void put_mask(unsigned char* self, uint32_t* words)
{
    *(void**)(self + 0x110) = words;
    *(void**)(self + 0x118) = words + g_words;
    *(void**)(self + 0x120) = words + g_words;
    *(uint64_t*)(self + 0x128) = (uint64_t)g_m;
}

static void __attribute__((ms_abi)) stub_set_scale(unsigned char* self, int scale)
{
    if (g_n_log < 96) g_log[g_n_log++] = 100 + scale;
    g_scale = scale;
    if (!g_chain) return;
    /* the head of the real setScale: right grid size and cell count; it ends at its first import (cv::Mat::zeros) */
    typedef void(__attribute__((ms_abi)) * scale_fn)(void* self, int scale);
    *(void**)(g_img + 0x903f8) = (void*)stop_at_mat_zeros;
    if (setjmp(g_back) == 0) {
        ((scale_fn)(g_img + 0x48c10))(self, scale);
        abort(); /* it must not get past cv::Mat::zeros */
    }
    const int wr = *(int32_t*)(self + 0x58), hr = *(int32_t*)(self + 0x5c), nr = *(int32_t*)(self + 0x64);
    if (nr != wr * hr || g_zeros_args[0] != nr || g_zeros_args[1] != 9) abort();
    /* what setScale goes on to do through opencv_core: the matrices of this right grid (storage), the neighbour table by the DLL */
    free(g_motion);
    free(g_nb_right);
    g_motion = malloc(sizeof(int32_t) * 400 * (size_t)nr);
    g_nb_right = malloc(sizeof(int32_t) * 9 * (size_t)nr);
    dll_neighbors(g_img, g_nb_right, wr, hr);
    g_step_motion = (uint64_t)nr * 4;
    *(void**)(self + 0x78) = g_motion;
    *(void**)(self + 0xb0) = &g_step_motion;
    *(void**)(self + 0x1a0) = g_nb_right;
    *(void**)(self + 0x1d8) = &g_step_nb;
}

static int __attribute__((ms_abi)) stub_run(unsigned char* self, int rotation)
{
    typedef void(__attribute__((ms_abi)) * assign_fn)(void* self, int grid_type);
    if (g_n_log < 96) g_log[g_n_log++] = rotation;
    if (!g_chain) {
        const int idx = g_scale * 8 + (rotation - 1);
        memcpy(g_mask_words, g_masks + (size_t)idx * g_words, 4 * (size_t)g_words);
        put_mask(self, g_mask_words);
        return g_counts[idx];
    }
    const int nr = *(int32_t*)(self + 0x64);
    /* run(): mvbInlierMask.assign(M, false); mvMatchPairs.assign(M, (0, 0)) */
    memset(g_mask_words, 0, 4 * (size_t)g_words);
    put_mask(self, g_mask_words);
    memset(g_pairs, 0, 8 * (size_t)g_m);
    int count = 0;
    for (int t = 1; t <= 4; t++) {
        /* mMotionStatistics.setTo(0); mCellPairs.assign(N, -1); mNumberPointsInPerCellLeft.assign(N, 0) */
        memset(g_motion, 0, sizeof(int32_t) * 400 * (size_t)nr);
        for (int i = 0; i < 400; i++) g_cell_pairs[i] = -1, g_nleft[i] = 0;
        *(int32_t*)(self + 0x60) = 400;
        ((assign_fn)(g_img + 0x47880))(self, t);
        for (int i = 0; i < 400; i++) {
            long long rowsum = 0;   /* cv::sum(row) == 0: the cell keeps -1 */
            for (int j = 0; j < nr; j++) rowsum += g_motion[(size_t)i * nr + j];
            if (rowsum == 0) continue;
            *(int32_t*)(self + 0x60) = i + 1; /* mGridNumberLeft: verifyCellPairs' loop ends after this cell */
            verify_fragment(self, rotation, i, g_img + 0x48e12, g_img + 0x2c5068);
        }
        *(int32_t*)(self + 0x60) = 400;
        count = mark_fragment(self, g_img + 0x48acd, g_img + 0x2c5068);
    }
    return count;
}

/* re-points the eight call rel32 of getInlierMask at the two functions above, through absolute jumps placed behind the image */
static int patch_get_inlier_mask(unsigned char* img)
{
    static const uint32_t to_scale[4] = {0x47de3, 0x47e12, 0x47e8b, 0x47fe8}, to_run[4] = {0x47def, 0x47e26, 0x47e95, 0x47ff2};
    unsigned char* stub = img + g_size_image;
    void* fn[2] = {(void*)stub_set_scale, (void*)stub_run};
    for (int k = 0; k < 2; k++) {
        unsigned char* s = stub + 16 * k;
        s[0] = 0x48; s[1] = 0xB8;                 /* mov rax, imm64 */
        memcpy(s + 2, &fn[k], 8);
        s[10] = 0xFF; s[11] = 0xE0;               /* jmp rax */
    }
    for (int k = 0; k < 4; k++) {
        const uint32_t sites[2] = {to_scale[k], to_run[k]};
        const uint32_t target[2] = {0x48c10, 0x48630};
        for (int w = 0; w < 2; w++) {
            unsigned char* at = img + sites[w];
            int32_t rel;
            memcpy(&rel, at + 1, 4);
            if (at[0] != 0xE8 || sites[w] + 5 + (uint32_t)rel != target[w]) return 0;   /* not the call this file expects */
            rel = (int32_t)((int64_t)(g_size_image + 16 * w) - (int64_t)(sites[w] + 5));
            memcpy(at + 1, &rel, 4);
        }
    }
    *(void**)(img + 0x901b0) = (void*)crt_malloc;
    *(void**)(img + 0x901b8) = (void*)crt_free;
    *(void**)(img + 0x90188) = (void*)crt_memmove;
    *(void**)(img + 0x90180) = (void*)crt_memmove;   /* memcpy */
    *(void**)(img + 0x90178) = (void*)crt_memset;
    return 1;
}

typedef int(__attribute__((ms_abi)) * inlier_fn)(void* self, void* mask_vec, int with_rotation, int with_scale);

typedef int(__attribute__((ms_abi)) * left_fn)(void* self, const float* pt, int type);
typedef int(__attribute__((ms_abi)) * right_fn)(void* self, const float* pt);

static uint32_t rd32(const unsigned char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const unsigned char* p) { uint16_t v; memcpy(&v, p, 2); return v; }

int main(int argc, char** argv)
{
    if (argc != 4 && argc != 5) return 2;
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
    unsigned char* img = mmap(NULL, (size_t)size_image + 0x1000, PROT_READ | PROT_WRITE | PROT_EXEC, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (img == MAP_FAILED) return 5;
    memcpy(img, file, size_headers);
    const unsigned char* sec = opt + optsz;
    for (int i = 0; i < nsec; i++, sec += 40) {
        const uint32_t vsize = rd32(sec + 8), va = rd32(sec + 12), rsize = rd32(sec + 16), rptr = rd32(sec + 20);
        const uint32_t n = rsize < vsize ? rsize : vsize;
        if ((uint64_t)va + n > size_image || (uint64_t)rptr + n > (uint64_t)fsz) return 4;
        memcpy(img + va, file + rptr, n);
    }
    typedef void(__attribute__((ms_abi)) * assign_fn)(void* self, int grid_type);
    if (argc == 5 && strcmp(argv[4], "assign") == 0) {
        assign_fn assign = (assign_fn)(img + 0x47880);
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t hdr[5];
        if (fread(hdr, 4, 5, in) != 5) return 6;
        const int wr = hdr[0], hr = hdr[1], n1 = hdr[2], n2 = hdr[3], m = hdr[4];
        float* p1 = malloc(sizeof(float) * 2 * (size_t)n1);
        float* p2 = malloc(sizeof(float) * 2 * (size_t)n2);
        int32_t* mt = malloc(sizeof(int32_t) * 2 * (size_t)m);
        int32_t* mp = calloc(2 * (size_t)m, sizeof(int32_t));
        int32_t* motion = malloc(sizeof(int32_t) * 400 * (size_t)wr * hr);
        int32_t nleft[400];
        if (fread(p1, 8, (size_t)n1, in) != (size_t)n1 || fread(p2, 8, (size_t)n2, in) != (size_t)n2 ||
            fread(mt, 8, (size_t)m, in) != (size_t)m)
            return 6;
        uint64_t step = (uint64_t)wr * hr * 4;  /* bytes per motion row */
        unsigned char self[0x200];
        memset(self, 0, sizeof self);
        *(void**)(self + 0x00) = p1;
        *(void**)(self + 0x18) = p2;
        *(void**)(self + 0x30) = mt;
        *(uint64_t*)(self + 0x48) = (uint64_t)m;
        *(int32_t*)(self + 0x50) = 20;
        *(int32_t*)(self + 0x54) = 20;
        *(int32_t*)(self + 0x58) = wr;
        *(int32_t*)(self + 0x5c) = hr;
        *(void**)(self + 0x78) = motion;
        *(void**)(self + 0xb0) = &step;
        *(void**)(self + 0xc8) = nleft;
        *(void**)(self + 0xf8) = mp;
        for (int t = 1; t <= 4; t++) {
            memset(motion, 0, sizeof(int32_t) * 400 * (size_t)wr * hr);
            memset(nleft, 0, sizeof nleft);
            assign(self, t);
            fwrite(mp, 8, (size_t)m, out);
            fwrite(nleft, 4, 400, out);
            fwrite(motion, 4, 400 * (size_t)wr * hr, out);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && strcmp(argv[4], "verify") == 0) {
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t hdr[2];
        double factor;
        if (fread(hdr, 4, 2, in) != 2 || fread(&factor, 8, 1, in) != 1) return 6;
        const int wr = hdr[0], hr = hdr[1], nr = wr * hr;
        int32_t nleft[400], cell_pairs[400];
        int32_t* motion = malloc(sizeof(int32_t) * 400 * (size_t)nr);
        if (fread(nleft, 4, 400, in) != 400 || fread(motion, 4, 400 * (size_t)nr, in) != 400 * (size_t)nr) return 6;
        int32_t* nb_left = malloc(sizeof(int32_t) * 9 * 400);
        int32_t* nb_right = malloc(sizeof(int32_t) * 9 * (size_t)nr);
        /* both neighbour tables by the DLL's own initalizeNeighbors / getNB9 (DLL@0x180048180 / 0x180048030) */
        dll_neighbors(img, nb_left, 20, 20);
        dll_neighbors(img, nb_right, wr, hr);
        uint64_t step_motion = (uint64_t)nr * 4, step_nb = 36;
        unsigned char self[0x200];
        memset(self, 0, sizeof self);
        *(int32_t*)(self + 0x64) = nr;
        *(void**)(self + 0x78) = motion;
        *(void**)(self + 0xb0) = &step_motion;
        *(void**)(self + 0xc8) = nleft;
        *(void**)(self + 0xe0) = cell_pairs;
        *(void**)(self + 0x140) = nb_left;
        *(void**)(self + 0x178) = &step_nb;
        *(void**)(self + 0x1a0) = nb_right;
        *(void**)(self + 0x1d8) = &step_nb;
        *(double*)(self + 0x1f0) = factor;
        for (int rot = 1; rot <= 8; rot++) {
            for (int i = 0; i < 400; i++) {
                cell_pairs[i] = -1; /* run(): mCellPairs.assign(400, -1) */
                long long rowsum = 0;
                for (int j = 0; j < nr; j++) rowsum += motion[(size_t)i * nr + j];
                if (rowsum == 0) continue; /* the cv::sum(row) == 0 branch keeps -1 */
                *(int32_t*)(self + 0x60) = i + 1; /* mGridNumberLeft: the loop ends after this cell */
                verify_fragment(self, rot, i, img + 0x48e12, img + 0x2c5068);
            }
            fwrite(cell_pairs, 4, 400, out);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && strcmp(argv[4], "nb9") == 0) {
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t n = 0;
        if (fread(&n, 4, 1, in) != 1) return 6;
        for (int i = 0; i < n; i++) {
            int32_t wh[2];
            if (fread(wh, 4, 2, in) != 2) return 6;
            int32_t* table = malloc(sizeof(int32_t) * 9 * (size_t)wh[0] * wh[1]);
            memset(table, 0x55, sizeof(int32_t) * 9 * (size_t)wh[0] * wh[1]);
            dll_neighbors(img, table, wh[0], wh[1]);
            fwrite(table, 4, 9 * (size_t)wh[0] * wh[1], out);
            free(table);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && strcmp(argv[4], "setscale") == 0) {
        typedef void(__attribute__((ms_abi)) * void_fn)(void);
        typedef void(__attribute__((ms_abi)) * scale_fn)(void* self, int scale);
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t left[2];
        if (fread(left, 4, 2, in) != 2) return 6;
        ((void_fn)(img + 0x10b0))();                       /* the static initialiser of mScaleRatios[2], [3] */
        fwrite(img + 0x2c5008, 8, 5, out);
        *(void**)(img + 0x903f8) = (void*)stop_at_mat_zeros;
        for (volatile int s = 0; s < 5; s++) {
            static unsigned char self[0x400];
            memset(self, 0, sizeof self);
            *(int32_t*)(self + 0x50) = left[0];
            *(int32_t*)(self + 0x54) = left[1];
            memset(g_zeros_args, 0xff, sizeof g_zeros_args);
            if (setjmp(g_back) == 0) {
                ((scale_fn)(img + 0x48c10))(self, s);
                return 7;                                   /* it must not get past cv::Mat::zeros */
            }
            int32_t res[6] = {*(int32_t*)(self + 0x58), *(int32_t*)(self + 0x5c), *(int32_t*)(self + 0x64),
                              g_zeros_args[0], g_zeros_args[1], g_zeros_args[2]};
            fwrite(res, 4, 6, out);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && strcmp(argv[4], "mark") == 0) {
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t m = 0;
        if (fread(&m, 4, 1, in) != 1 || m < 0) return 6;
        const size_t words = ((size_t)m + 31) / 32;
        int32_t* pairs = malloc(8 * ((size_t)m ? (size_t)m : 1));
        uint32_t* mask = calloc(words ? words : 1, 4);
        int32_t cell_pairs[400];
        unsigned char self[0x200];
        memset(self, 0, sizeof self);
        *(uint64_t*)(self + 0x48) = (uint64_t)m;
        *(void**)(self + 0xe0) = cell_pairs;
        *(void**)(self + 0xf8) = pairs;
        *(void**)(self + 0x110) = mask;
        *(void**)(self + 0x118) = mask + words;
        *(void**)(self + 0x120) = mask + words;
        *(uint64_t*)(self + 0x128) = (uint64_t)m;
        for (int t = 1; t <= 4; t++) {
            if (fread(pairs, 8, (size_t)m, in) != (size_t)m || fread(cell_pairs, 4, 400, in) != 400) return 6;
            const int32_t count = mark_fragment(self, img + 0x48acd, img + 0x2c5068);
            fwrite(mask, 4, words, out);
            fwrite(&count, 4, 1, out);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && (strcmp(argv[4], "select") == 0 || strcmp(argv[4], "chain") == 0)) {
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        g_img = img;
        g_size_image = size_image;
        g_chain = strcmp(argv[4], "chain") == 0;
        if (!patch_get_inlier_mask(img)) return 8;
        static unsigned char self[0x400];
        memset(self, 0, sizeof self);
        int with_rotation, with_scale;
        if (!g_chain) {
            int32_t hdr[3];
            if (fread(hdr, 4, 3, in) != 3) return 6;
            g_m = hdr[0];
            with_rotation = hdr[1];
            with_scale = hdr[2];
            g_words = (g_m + 31) / 32;
            int32_t* counts = malloc(4 * 40);
            uint32_t* masks = malloc(4 * 40 * (size_t)(g_words ? g_words : 1));
            if (fread(counts, 4, 40, in) != 40 || fread(masks, 4, 40 * (size_t)g_words, in) != 40 * (size_t)g_words) return 6;
            g_counts = counts;
            g_masks = masks;
        } else {
            int32_t hdr[9];
            double factor;
            if (fread(hdr, 4, 9, in) != 9 || fread(&factor, 8, 1, in) != 1) return 6;
            const size_t n1 = (size_t)hdr[0], n2 = (size_t)hdr[1];
            g_m = hdr[2];
            with_rotation = hdr[7];
            with_scale = hdr[8];
            g_words = (g_m + 31) / 32;
            unsigned char* kp = malloc(28 * (n1 + n2 + 1));
            int32_t* mt = malloc(8 * ((size_t)g_m + 1));
            if (fread(kp, 28, n1 + n2, in) != n1 + n2 || fread(mt, 8, (size_t)g_m, in) != (size_t)g_m) return 6;
            /* the constructor: normalizePoints for both images (the DLL's), convertMatches (the copy above), the 20 x 20 left grid and
             * its neighbour table (the DLL's initalizeNeighbors) */
            typedef void(__attribute__((ms_abi)) * norm_fn)(void* self, void* kp_vec, const int32_t* size, void* out_vec);
            float* p1 = malloc(8 * (n1 + 1));
            float* p2 = malloc(8 * (n2 + 1));
            void* kv1[3] = {kp, kp + 28 * n1, kp + 28 * n1};
            void* ov1[3] = {p1, (unsigned char*)p1 + 8 * n1, (unsigned char*)p1 + 8 * n1};
            void* kv2[3] = {kp + 28 * n1, kp + 28 * (n1 + n2), kp + 28 * (n1 + n2)};
            void* ov2[3] = {p2, (unsigned char*)p2 + 8 * n2, (unsigned char*)p2 + 8 * n2};
            ((norm_fn)(img + 0x48420))(self, kv1, hdr + 3, ov1);
            ((norm_fn)(img + 0x48420))(self, kv2, hdr + 5, ov2);
            if (ov1[0] != (void*)p1 || ov2[0] != (void*)p2) return 7;
            typedef void(__attribute__((ms_abi)) * void_fn)(void);
            ((void_fn)(img + 0x10b0))();   /* the static initialiser of mScaleRatios[2], [3] */
            g_nleft = malloc(4 * 400);
            g_cell_pairs = malloc(4 * 400);
            g_pairs = malloc(8 * ((size_t)g_m + 1));
            g_nb_left = malloc(4 * 9 * 400);
            dll_neighbors(img, g_nb_left, 20, 20);
            *(void**)(self + 0x00) = p1;
            *(void**)(self + 0x18) = p2;
            *(void**)(self + 0x30) = mt;
            *(uint64_t*)(self + 0x48) = (uint64_t)g_m;
            *(int32_t*)(self + 0x50) = 20;
            *(int32_t*)(self + 0x54) = 20;
            *(int32_t*)(self + 0x60) = 400;
            *(void**)(self + 0xc8) = g_nleft;
            *(void**)(self + 0xe0) = g_cell_pairs;
            *(void**)(self + 0xf8) = g_pairs;
            *(void**)(self + 0x140) = g_nb_left;
            *(void**)(self + 0x178) = &g_step_nb;
            *(double*)(self + 0x1f0) = factor;
        }
        g_mask_words = calloc(g_words ? (size_t)g_words : 1, 4);
        memset(g_log, 0xff, sizeof g_log);
        void* mask_vec[4] = {NULL, NULL, NULL, NULL};   /* the caller's std::vector<bool>: empty */
        const int32_t ret = ((inlier_fn)(img + 0x47dc0))(self, mask_vec, with_rotation, with_scale);
        const int64_t bits = (int64_t)(uint64_t)(uintptr_t)mask_vec[3];
        fwrite(&ret, 4, 1, out);
        fwrite(&bits, 8, 1, out);
        uint32_t* res = calloc(g_words ? (size_t)g_words : 1, 4);
        if (bits != 0) {
            if (bits != g_m) return 9;
            memcpy(res, mask_vec[0], 4 * (size_t)g_words);
        }
        fwrite(res, 4, (size_t)g_words, out);
        if (!g_chain) {
            fwrite(&g_n_log, 4, 1, out);
            fwrite(g_log, 4, 96, out);
        }
        fclose(in);
        fclose(out);
        return 0;
    }
    if (argc == 5 && strcmp(argv[4], "normalize") == 0) {
        typedef void(__attribute__((ms_abi)) * norm_fn)(void* self, void* kp_vec, const int32_t* size, void* out_vec);
        FILE* in = fopen(argv[2], "rb");
        FILE* out = fopen(argv[3], "wb");
        if (!in || !out) return 6;
        int32_t hdr[3];
        if (fread(hdr, 4, 3, in) != 3) return 6;
        const size_t n = (size_t)hdr[0];
        unsigned char* kp = malloc(28 * (n ? n : 1));
        float* npts = malloc(8 * (n ? n : 1));
        if (fread(kp, 28, n, in) != n) return 6;
        memset(npts, 0x55, 8 * (n ? n : 1));
        void* kp_vec[3] = {kp, kp + 28 * n, kp + 28 * n};
        void* out_vec[3] = {npts, (unsigned char*)npts + 8 * n, (unsigned char*)npts + 8 * n};
        unsigned char self[0x200];
        memset(self, 0, sizeof self);
        ((norm_fn)(img + 0x48420))(self, kp_vec, hdr + 1, out_vec);
        if (out_vec[0] != (void*)npts) return 7; /* it must not have reallocated */
        fwrite(npts, 8, n, out);
        fclose(in);
        fclose(out);
        return 0;
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
