/*
 * oracle/gms_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of cv::xfeatures2d::matchGMS as shipped in the reference's
 * SfM-GMS/bin/opencv_xfeatures2d452.dll (opencv_contrib xfeatures2d 4.5.2, class GMSMatcher).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * path (sfm-gms_amd/csrc) never links, calls or falls back to it.
 *
 * PARITY: PINNED ONLY IN PART. The reference holds no tests, golden vectors or recorded outputs for this path
 *  and its implementation exists only as a Windows PE32+ DLL whose imports
 * (opencv_core452.dll, ...) are not vendored, so the function as a whole cannot be run here: end-to-end
 * parity is UNPINNED. What is pinned against the reference itself:
 *   - grid_index_left / grid_index_right (the float -> cell arithmetic, where bit-exactness is decided): the two
 *     corresponding leaf functions of the DLL were executed here on 6.7k points and their outputs are committed as
 *     tests/golden/refdll_grid_index.npz (generator: tests/golden/refdll_runner.c, make_refdll_vectors.py);
 *     tests/test_oracle_pins.py requires this file to reproduce every integer;
 *   - assign_match_pairs (binning: the skip rule, the right cell cached by grid type 1, motion[l][r]++ and the
 *     per-cell counts): GMSMatcher::assignMatchPairs was executed out of the DLL (its only call is
 *     getGridIndexLeft) for grid types 1..4 on three right grids; tests/golden/refdll_assign_pairs.npz holds
 *     what it wrote, and gms_ref_assign_pairs() must reproduce it;
 *   - verify_cell_pairs (arg-max, rotated 3 x 3 neighbour sums, sqrt(T / n) * factor, '>'): the body of
 *     GMSMatcher::verifyCellPairs after its cv::sum(row) test (DLL@0x180048e12) was executed out of the DLL one
 *     cell at a time for rotation types 1..8 on nine motion matrices (all five right grids, four factors);
 *     tests/golden/refdll_verify_cells.npz holds mCellPairs and gms_ref_verify_cells() must reproduce it;
 *   - init_neighbors (getNB9 / initalizeNeighbors): the DLL's own initalizeNeighbors was executed (its operator new / delete
 *     pointed at this process's allocator) for the left grid, the five right grids of setScale and three odd grids;
 *     tests/golden/refdll_nb9.npz holds the tables, gms_ref_neighbors() must reproduce them -- and the verify fragment
 *     above consumed THOSE tables, not restated ones;
 *   - the rotation-pattern table, the scale-ratio table and the 0.5 constant: compared byte for byte with the DLL
 *     by tests/test_oracle_pins.py when /root/reference exists.
 *   - normalize_points: GMSMatcher::normalizePoints executed out of the DLL on cv::KeyPoint records of eight image sizes
 *     (tests/golden/refdll_normalize.npz); set_scale's arithmetic: the head of GMSMatcher::setScale executed up to its first
 *     import, after the DLL's own static initialiser of mScaleRatios (tests/golden/refdll_setscale.npz).
 * What is left (the loops of run and getInlierMask, matchGMS's copy-out) follows the DLL's
 * disassembly address by address (cited per function, "DLL@0x..." = virtual address in that DLL, image base
 * 0x180000000) and keeps the reference's dense 400 x N_right motion matrix and
 * loop order on purpose, so that it is an obviously faithful, structurally independent checker for the sparse
 * GPU formulation; a second restatement (gms_ref_sparse.py) must agree with it bit for bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (x86-64 SSE2: float ops are true fp32, no x87).
 */
#include "gms_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* mRotationPatterns: DLL .rdata 0x18012f520 (file offset 0x12df20), 8 x 9 int32. */
const int gms_ref_rotation_patterns[8][9] = {
    {1, 2, 3, 4, 5, 6, 7, 8, 9}, {4, 1, 2, 7, 5, 3, 8, 9, 6}, {7, 4, 1, 8, 5, 2, 9, 6, 3},
    {8, 7, 4, 9, 5, 1, 6, 3, 2}, {9, 8, 7, 6, 5, 4, 3, 2, 1}, {6, 9, 8, 3, 5, 7, 2, 1, 4},
    {3, 6, 9, 2, 5, 8, 1, 4, 7}, {2, 3, 6, 1, 5, 9, 4, 7, 8}};

/* mScaleRatios: DLL .data 0x1802c5008 = {1.0, 0.5, <dyn>, <dyn>, 2.0}; the two dynamic slots are
 * 1/sqrt(2) and sqrt(2), written by the static initialiser. */
double gms_ref_scale_ratio(int s)
{
    switch (s) {
    case 0: return 1.0;
    case 1: return 1.0 / 2;
    case 2: return 1.0 / sqrt(2.0);
    case 3: return sqrt(2.0);
    default: return 2.0;
    }
}

/* cvFloor(float) / cvFloor(double) as compiled into the DLL: cvttss2si / cvttsd2si, then
 * "i -= (i > v)" (DLL@0x180047be3-0x180047c01, 0x180047c65-0x180047c77). */
static int floor_f32(float v)
{
    int i = (int)v;
    return i - ((float)i > v);
}
static int floor_f64(double v)
{
    int i = (int)v;
    return i - ((double)i > v);
}
/* cvRound(double): cvtsd2si, round-half-even in the default MXCSR mode (DLL@0x180048c10). */
static int round_f64(double v) { return (int)lrint(v); }

typedef struct {
    int n_matches;
    float *p1, *p2;            /* normalised points, 2 floats each            (normalizePoints) */
    int n1, n2;
    const gms_dmatch* matches; /* (queryIdx, trainIdx) read in place            (convertMatches) */
    int wl, hl, n_left;        /* left grid 20 x 20                     (DLL@0x180046ac6-ace)   */
    int wr, hr, n_right;       /* right grid, per scale                                (setScale) */
    int* nb_left;              /* [n_left][9]                                                  */
    int* nb_right;             /* [n_right][9]                                                 */
    int* motion;               /* [n_left][n_right] dense                                      */
    int* n_per_cell_left;      /* [n_left]                                                     */
    int* cell_pairs;           /* [n_left]                                                     */
    int *pair_first, *pair_second; /* mvMatchPairs                                             */
    unsigned char* mask;       /* mvbInlierMask                                                */
    double threshold_factor;
    int domain_error;
    unsigned char* best_mask;  /* getInlierMask's copy of the best mask                        */
    /* allocation sizes (bytes) of the buffers above, in the order of buf_slot() below: a state can be kept and
     * reused from call to call (gms_ref_scratch), buffers only ever grow -- storage only, no arithmetic */
    size_t cap[11];
} gms_ref_state;

/* Grow-only buffer: makes *p hold at least `bytes` bytes. Contents are NOT preserved (every user initialises what it
 * reads, exactly as after the reference's fresh allocations). */
static int ensure(void** p, size_t* cap, size_t bytes)
{
    if (bytes == 0) bytes = 1;
    if (*cap >= bytes) return 0;
    free(*p);
    *p = malloc(bytes);
    *cap = *p ? bytes : 0;
    return *p ? 0 : -1;
}
#define ENSURE(st, field, slot, bytes) ensure((void**)&(st)->field, &(st)->cap[slot], (bytes))

/* GMSMatcher::normalizePoints, DLL@0x180048420: cvtdq2ps on width/height, divss per coordinate. */
static void normalize_points(const gms_keypoint* kp, int n, int width, int height, float* out)
{
    for (int i = 0; i < n; i++) {
        out[2 * i + 0] = kp[i].x / (float)width;
        out[2 * i + 1] = kp[i].y / (float)height;
    }
}

/* GMSMatcher::getNB9 / initalizeNeighbors, DLL@0x180048030 / 0x180048180. */
static void init_neighbors(int* nb, int gw, int gh)
{
    for (int idx = 0; idx < gw * gh; idx++) {
        int* nb9 = nb + 9 * idx;
        for (int k = 0; k < 9; k++) nb9[k] = -1;
        int ix = idx % gw, iy = idx / gw;
        for (int yi = -1; yi <= 1; yi++)
            for (int xi = -1; xi <= 1; xi++) {
                int xx = ix + xi, yy = iy + yi;
                if (xx < 0 || xx >= gw || yy < 0 || yy >= gh) continue;
                nb9[xi + 4 + yi * 3] = xx + yy * gw;
            }
    }
}

/* GMSMatcher::getGridIndexLeft, DLL@0x180047bc0-0x180047d55. The product (float)W * nx is rounded
 * to fp32 (mulss) before it is widened (cvtps2pd) and 0.5 added in fp64 (addsd [0x18012dfc0]).
 * One common bounds test for all four grid types: x >= W || y >= H -> -1; no lower-bound test. */
static int grid_index_left(const gms_ref_state* st, const float* pt, int type)
{
    int x = 0, y = 0;
    float fx = (float)st->wl * pt[0];
    float fy = (float)st->hl * pt[1];
    if (type == 1) {
        x = floor_f32(fx);
        y = floor_f32(fy);
    } else if (type == 2) {
        x = floor_f64((double)fx + 0.5);
        y = floor_f32(fy);
    } else if (type == 3) {
        x = floor_f32(fx);
        y = floor_f64((double)fy + 0.5);
    } else if (type == 4) {
        x = floor_f64((double)fx + 0.5);
        y = floor_f64((double)fy + 0.5);
    }
    if (x >= st->wl || y >= st->hl) return -1;
    return x + y * st->wl;
}

/* GMSMatcher::getGridIndexRight, DLL@0x180047d60 (inlined at 0x1800478e7): no bounds test. */
static int grid_index_right(const gms_ref_state* st, const float* pt)
{
    int x = floor_f32((float)st->wr * pt[0]);
    int y = floor_f32((float)st->hr * pt[1]);
    return x + y * st->wr;
}

/* GMSMatcher::setScale, DLL@0x180048c10. */
static int set_scale(gms_ref_state* st, int scale)
{
    st->wr = round_f64(st->wl * gms_ref_scale_ratio(scale));
    st->hr = round_f64(st->hl * gms_ref_scale_ratio(scale));
    st->n_right = st->wr * st->hr;
    if (ENSURE(st, nb_right, 0, sizeof(int) * 9 * (size_t)st->n_right) ||
        ENSURE(st, motion, 1, sizeof(int) * (size_t)st->n_left * (size_t)st->n_right))
        return -1;
    init_neighbors(st->nb_right, st->wr, st->hr);
    return 0;
}

/* A coordinate is inside the parity domain when the float->int conversions of the reference are
 * defined and non-negative (SURVEY.md section 8a "valid input domain"): finite, >= 0, < 2^20. */
static int coord_ok(float v) { return v >= 0.0f && v < 1048576.0f; }

/* GMSMatcher::assignMatchPairs, DLL@0x180047880 (inlined in run at 0x1800489e0-0x180048ab4). */
static void assign_match_pairs(gms_ref_state* st, int grid_type)
{
    for (int i = 0; i < st->n_matches; i++) {
        const float* lp = st->p1 + 2 * (size_t)st->matches[i].queryIdx;
        const float* rp = st->p2 + 2 * (size_t)st->matches[i].trainIdx;
        int lgidx = st->pair_first[i] = grid_index_left(st, lp, grid_type);
        int rgidx;
        if (grid_type == 1)
            rgidx = st->pair_second[i] = grid_index_right(st, rp);
        else
            rgidx = st->pair_second[i];
        if (lgidx < 0 || rgidx < 0) continue;
        if (rgidx >= st->n_right) { /* the reference would write outside its matrix row/allocation */
            st->domain_error = 1;
            continue;
        }
        st->motion[(size_t)lgidx * st->n_right + rgidx]++;
        st->n_per_cell_left[lgidx]++;
    }
}

/* GMSMatcher::verifyCellPairs, DLL@0x180048d10-0x1800491ec. */
static void verify_cell_pairs(gms_ref_state* st, int rotation_type)
{
    const int* rp = gms_ref_rotation_patterns[rotation_type - 1];
    for (int i = 0; i < st->n_left; i++) {
        const int* row = st->motion + (size_t)i * st->n_right;
        /* cv::sum(row)[0] == 0 (fp64 compare at DLL@0x180048dd7) */
        double s = 0;
        for (int j = 0; j < st->n_right; j++) s += row[j];
        if (s == 0) {
            st->cell_pairs[i] = -1;
            continue;
        }
        /* arg-max, strict '>' from 0, ascending j: lowest index among maxima (DLL@0x180048e20-e4f) */
        int max_number = 0;
        for (int j = 0; j < st->n_right; j++)
            if (row[j] > max_number) {
                st->cell_pairs[i] = j;
                max_number = row[j];
            }
        int idx_grid_rt = st->cell_pairs[i];
        const int* nb9_lt = st->nb_left + 9 * i;
        const int* nb9_rt = st->nb_right + 9 * idx_grid_rt;
        int score = 0;
        double thresh = 0;
        int numpair = 0;
        for (int j = 0; j < 9; j++) {
            int ll = nb9_lt[j];
            int rr = nb9_rt[rp[j] - 1];
            if (ll == -1 || rr == -1) continue;
            score += st->motion[(size_t)ll * st->n_right + rr];
            thresh += (double)st->n_per_cell_left[ll];
            numpair++;
        }
        /* divsd, sqrtsd, mulsd [this+0x1f0]; comisd thresh,score; reject iff thresh > score
         * (DLL@0x180049171-0x18004919d) */
        thresh = sqrt(thresh / (double)numpair) * st->threshold_factor;
        if (thresh > (double)score) st->cell_pairs[i] = -2;
    }
}

/* GMSMatcher::run, DLL@0x180048630-0x180048c08. */
static int run(gms_ref_state* st, int rotation_type)
{
    memset(st->mask, 0, (size_t)st->n_matches);
    for (int i = 0; i < st->n_matches; i++) st->pair_first[i] = st->pair_second[i] = 0;
    for (int grid_type = 1; grid_type <= 4; grid_type++) {
        memset(st->motion, 0, sizeof(int) * (size_t)st->n_left * (size_t)st->n_right);
        for (int i = 0; i < st->n_left; i++) {
            st->cell_pairs[i] = -1;
            st->n_per_cell_left[i] = 0;
        }
        assign_match_pairs(st, grid_type);
        verify_cell_pairs(st, rotation_type);
        for (int i = 0; i < st->n_matches; i++)
            if (st->pair_first[i] >= 0 && st->cell_pairs[st->pair_first[i]] == st->pair_second[i])
                st->mask[i] = 1;
    }
    int c = 0;
    for (int i = 0; i < st->n_matches; i++) c += st->mask[i];
    return c;
}

/* matchGMS (DLL@0x180048280) on a caller-kept state: `st` holds nothing but storage between calls. */
static int match_with_state(gms_ref_state* st, const gms_keypoint* kp1, int n1, int w1, int h1,
                            const gms_keypoint* kp2, int n2, int w2, int h2,
                            const gms_dmatch* matches, int m,
                            int with_rotation, int with_scale, double threshold_factor,
                            gms_dmatch* out, int* n_out, unsigned char* mask_out, gms_pair_result* result)
{
    if (n_out) *n_out = 0;
    if (result) {
        result->n_inliers = 0;
        result->best_scale = -1;
        result->best_rot = -1;
        result->status = GMS_OK;
    }
    if (n1 < 0 || n2 < 0 || m < 0 || w1 <= 0 || h1 <= 0 || w2 <= 0 || h2 <= 0) return GMS_ERR_BAD_ARG;
    if ((n1 > 0 && !kp1) || (n2 > 0 && !kp2) || (m > 0 && (!matches || !out)) || !n_out)
        return GMS_ERR_BAD_ARG;
    if (mask_out && m > 0) memset(mask_out, 0, (size_t)m);

    st->n_matches = m;
    st->n1 = n1;
    st->n2 = n2;
    st->matches = matches;
    st->threshold_factor = threshold_factor;
    st->domain_error = 0;
    st->wl = 20;
    st->hl = 20;
    st->n_left = 400;

    int rc = GMS_OK;
    unsigned char* best_mask = NULL;
    if (ENSURE(st, p1, 2, sizeof(float) * 2 * (size_t)n1) || ENSURE(st, p2, 3, sizeof(float) * 2 * (size_t)n2) ||
        ENSURE(st, nb_left, 4, sizeof(int) * 9 * 400) || ENSURE(st, n_per_cell_left, 5, sizeof(int) * 400) ||
        ENSURE(st, cell_pairs, 6, sizeof(int) * 400) || ENSURE(st, pair_first, 7, sizeof(int) * (size_t)m) ||
        ENSURE(st, pair_second, 8, sizeof(int) * (size_t)m) || ENSURE(st, mask, 9, (size_t)m) ||
        ENSURE(st, best_mask, 10, (size_t)m)) {
        rc = GMS_ERR_BAD_ARG;
        goto done;
    }
    best_mask = st->best_mask;
    memset(best_mask, 0, (size_t)(m ? m : 1));

    normalize_points(kp1, n1, w1, h1, st->p1);
    normalize_points(kp2, n2, w2, h2, st->p2);
    init_neighbors(st->nb_left, st->wl, st->hl);

    /* Domain check (the reference has none, DLL@0x180048280): indices in range, matched points
     * finite and non-negative. Outside it the reference reads/writes out of bounds. */
    for (int i = 0; i < m; i++) {
        int q = matches[i].queryIdx, t = matches[i].trainIdx;
        if (q < 0 || q >= n1 || t < 0 || t >= n2) {
            rc = GMS_ERR_DOMAIN;
            goto done;
        }
        if (!coord_ok(st->p1[2 * q]) || !coord_ok(st->p1[2 * q + 1]) || !coord_ok(st->p2[2 * t]) ||
            !coord_ok(st->p2[2 * t + 1])) {
            rc = GMS_ERR_DOMAIN;
            goto done;
        }
    }

    /* GMSMatcher::getInlierMask, DLL@0x180047dc0-0x180047fe1: scale outer (0..4), rotation inner
     * (1..8); keep (mask, count) on strict '>' starting from 0. Without either flag: setScale(0),
     * run(1), take the mask unconditionally. */
    int max_inlier = 0, best_scale = -1, best_rot = -1;
    int n_scales = with_scale ? 5 : 1;
    int n_rots = with_rotation ? 8 : 1;
    for (int scale = 0; scale < n_scales; scale++) {
        if (set_scale(st, scale)) {
            rc = GMS_ERR_BAD_ARG;
            goto done;
        }
        for (int rot = 1; rot <= n_rots; rot++) {
            int num_inlier = run(st, rot);
            if (st->domain_error) {
                rc = GMS_ERR_DOMAIN;
                goto done;
            }
            if (num_inlier > max_inlier) {
                memcpy(best_mask, st->mask, (size_t)m);
                max_inlier = num_inlier;
                best_scale = scale;
                best_rot = rot;
            }
        }
    }

    /* matchGMS copy-out, DLL@0x18004831e-0x180048371: clear, then push every masked match. */
    {
        int k = 0;
        for (int i = 0; i < m; i++)
            if (best_mask[i]) out[k++] = matches[i];
        *n_out = k;
        if (mask_out && m > 0) memcpy(mask_out, best_mask, (size_t)m);
        if (result) {
            result->n_inliers = k;
            result->best_scale = best_scale;
            result->best_rot = best_rot;
        }
    }

done:
    if (result) result->status = rc;
    return rc;
}

static void state_release(gms_ref_state* st)
{
    free(st->p1);
    free(st->p2);
    free(st->nb_left);
    free(st->nb_right);
    free(st->motion);
    free(st->n_per_cell_left);
    free(st->cell_pairs);
    free(st->pair_first);
    free(st->pair_second);
    free(st->mask);
    free(st->best_mask);
    memset(st, 0, sizeof *st);
}

/* Whole call with fresh storage, as the reference's stack GMSMatcher has it (DLL@0x180048280). */
int gms_ref_match(const gms_keypoint* kp1, int n1, int w1, int h1,
                  const gms_keypoint* kp2, int n2, int w2, int h2,
                  const gms_dmatch* matches, int m,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, int* n_out, unsigned char* mask_out, gms_pair_result* result)
{
    gms_ref_state st;
    memset(&st, 0, sizeof st);
    int rc = match_with_state(&st, kp1, n1, w1, h1, kp2, n2, w2, h2, matches, m, with_rotation, with_scale,
                              threshold_factor, out, n_out, mask_out, result);
    state_release(&st);
    return rc;
}

/* The same call on storage kept from call to call (one scratch per host thread): identical arithmetic and loop
 * order, no allocator traffic once the buffers have reached the largest shape. For the multi-threaded CPU baseline
 * (gms_ref_mt.c): glibc serves the 640 KB .. 2.5 MB motion matrix by mmap/munmap, which serialises every thread of
 * the process on one kernel lock. */
struct gms_ref_scratch {
    gms_ref_state st;
};
gms_ref_scratch* gms_ref_scratch_create(void) { return (gms_ref_scratch*)calloc(1, sizeof(gms_ref_scratch)); }
void gms_ref_scratch_destroy(gms_ref_scratch* s)
{
    if (!s) return;
    state_release(&s->st);
    free(s);
}
int gms_ref_match_ws(gms_ref_scratch* s, const gms_keypoint* kp1, int n1, int w1, int h1,
                     const gms_keypoint* kp2, int n2, int w2, int h2,
                     const gms_dmatch* matches, int m,
                     int with_rotation, int with_scale, double threshold_factor,
                     gms_dmatch* out, int* n_out, unsigned char* mask_out, gms_pair_result* result)
{
    if (!s) return GMS_ERR_BAD_ARG;
    return match_with_state(&s->st, kp1, n1, w1, h1, kp2, n2, w2, h2, matches, m, with_rotation, with_scale,
                            threshold_factor, out, n_out, mask_out, result);
}

/* Exposed pieces, so tests can pin them one at a time. */
int gms_ref_grid_index_left(float nx, float ny, int type)
{
    gms_ref_state st;
    memset(&st, 0, sizeof st);
    st.wl = st.hl = 20;
    float pt[2] = {nx, ny};
    return grid_index_left(&st, pt, type);
}
int gms_ref_grid_index_right(float nx, float ny, int wr, int hr)
{
    gms_ref_state st;
    memset(&st, 0, sizeof st);
    st.wr = wr;
    st.hr = hr;
    float pt[2] = {nx, ny};
    return grid_index_right(&st, pt);
}
void gms_ref_right_grid(int scale, int* wr, int* hr)
{
    *wr = round_f64(20 * gms_ref_scale_ratio(scale));
    *hr = round_f64(20 * gms_ref_scale_ratio(scale));
}
/* setScale's grid arithmetic for any left grid (DLL@0x180048c37-0x180048c71): pinned by tests/golden/refdll_setscale.npz */
void gms_ref_right_grid_from(int left_w, int left_h, int scale, int* wr, int* hr)
{
    *wr = round_f64(left_w * gms_ref_scale_ratio(scale));
    *hr = round_f64(left_h * gms_ref_scale_ratio(scale));
}
float gms_ref_normalize(float v, int extent) { return v / (float)extent; }
int gms_ref_threshold_rejects(int T, int n, int score, double factor)
{
    double thresh = sqrt((double)T / (double)n) * factor;
    return thresh > (double)score;
}

/* initalizeNeighbors / getNB9 for a gw x gh grid: out[gw * gh][9]. Pinned against the DLL's own initalizeNeighbors
 * (DLL@0x180048180, tests/golden/refdll_nb9.npz). */
void gms_ref_neighbors(int gw, int gh, int* out) { init_neighbors(out, gw, gh); }

/* assignMatchPairs for grid types 1..4 in sequence, exactly as run() drives it (motion and nLeft zeroed before each
 * type, the right cell cached by type 1), on ALREADY NORMALISED points. Outputs per grid type t (0-based):
 * pairs[t][m][2] = mvMatchPairs, nleft[t][400], motion[t][400 * wr * hr] (dense). Returns 0, or -1 on allocation
 * failure / -2 if a right cell leaves the grid. Pinned against the DLL's own assignMatchPairs (DLL@0x180047880). */
int gms_ref_assign_pairs(const float* p1, const float* p2, const int* matches, int m, int wr, int hr,
                         int* pairs, int* nleft, int* motion)
{
    gms_ref_state st;
    memset(&st, 0, sizeof st);
    st.wl = st.hl = 20;
    st.n_left = 400;
    st.wr = wr;
    st.hr = hr;
    st.n_right = wr * hr;
    st.n_matches = m;
    st.p1 = (float*)p1;
    st.p2 = (float*)p2;
    gms_dmatch* dm = (gms_dmatch*)calloc((size_t)(m ? m : 1), sizeof(gms_dmatch));
    st.pair_first = (int*)calloc((size_t)(m ? m : 1), sizeof(int));
    st.pair_second = (int*)calloc((size_t)(m ? m : 1), sizeof(int));
    if (!dm || !st.pair_first || !st.pair_second) return -1;
    for (int i = 0; i < m; i++) {
        dm[i].queryIdx = matches[2 * i];
        dm[i].trainIdx = matches[2 * i + 1];
    }
    st.matches = dm;
    for (int t = 1; t <= 4; t++) {
        st.motion = motion + (size_t)(t - 1) * 400 * st.n_right;
        st.n_per_cell_left = nleft + (t - 1) * 400;
        memset(st.motion, 0, sizeof(int) * 400 * (size_t)st.n_right);
        memset(st.n_per_cell_left, 0, sizeof(int) * 400);
        assign_match_pairs(&st, t);
        for (int i = 0; i < m; i++) {
            pairs[((size_t)(t - 1) * m + i) * 2 + 0] = st.pair_first[i];
            pairs[((size_t)(t - 1) * m + i) * 2 + 1] = st.pair_second[i];
        }
    }
    free(dm);
    free(st.pair_first);
    free(st.pair_second);
    return st.domain_error ? -2 : 0;
}

/* verifyCellPairs for one rotation type on a given dense motion matrix and per-cell counts (what assignMatchPairs
 * left behind): cell_pairs_out[400] as run() would see them (-1 empty, -2 rejected, else the right cell).
 * Pinned against the body of the DLL's own verifyCellPairs (DLL@0x180048e12 onwards, see tests/golden/refdll_runner.c). */
int gms_ref_verify_cells(const int* motion, const int* nleft, int wr, int hr, int rotation_type, double factor,
                         int* cell_pairs_out)
{
    gms_ref_state st;
    memset(&st, 0, sizeof st);
    st.wl = st.hl = 20;
    st.n_left = 400;
    st.wr = wr;
    st.hr = hr;
    st.n_right = wr * hr;
    st.threshold_factor = factor;
    st.motion = (int*)motion;
    st.n_per_cell_left = (int*)nleft;
    st.cell_pairs = cell_pairs_out;
    st.nb_left = (int*)malloc(sizeof(int) * 9 * 400);
    st.nb_right = (int*)malloc(sizeof(int) * 9 * (size_t)st.n_right);
    if (!st.nb_left || !st.nb_right) return -1;
    init_neighbors(st.nb_left, 20, 20);
    init_neighbors(st.nb_right, wr, hr);
    for (int i = 0; i < 400; i++) cell_pairs_out[i] = -1;
    verify_cell_pairs(&st, rotation_type);
    free(st.nb_left);
    free(st.nb_right);
    return 0;
}
