/*
 * oracle/gms_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement of cv::xfeatures2d::matchGMS as shipped in the reference's
 * SfM-GMS/bin/opencv_xfeatures2d452.dll (opencv_contrib xfeatures2d 4.5.2, class GMSMatcher).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * path (sfm-gms_amd/csrc) never links, calls or falls back to it.
 *
 * PARITY: PINNED BY THE REFERENCE BINARY'S OWN CODE, PIECE BY PIECE AND AS A CHAIN -- not by a run of the unmodified function.
 * The reference holds no tests, golden vectors or recorded outputs for this path and its
 * implementation exists only as a Windows PE32+ DLL whose imports (opencv_core452.dll, ...) are not vendored, so matchGMS cannot
 * be called as it stands. What tests/golden/refdll_runner.c does instead is map the DLL and execute its functions -- whole where
 * they are self-contained, from behind their first import where they are not -- on inputs built here; only inputs and returned
 * values are committed (tests/golden/refdll_*.npz, generator make_refdll_vectors.py), and tests/test_oracle_pins.py /
 * tests/test_golden.py require this file to reproduce every one of them:
 *   - grid_index_left / grid_index_right (the float -> cell arithmetic): the DLL's two leaf functions on 6.7k points;
 *   - assign_match_pairs: GMSMatcher::assignMatchPairs for grid types 1..4 on three right grids;
 *   - verify_cell_pairs: the body of GMSMatcher::verifyCellPairs behind its cv::sum(row) test, one cell at a time, rotation
 *     types 1..8, nine motion matrices (all five right grids, four factors), on neighbour tables the DLL filled itself;
 *   - init_neighbors: GMSMatcher::initalizeNeighbors / getNB9 (operator new / delete pointed at this process's allocator);
 *   - normalize_points: GMSMatcher::normalizePoints on cv::KeyPoint records of eight image sizes; set_scale: the head of
 *     GMSMatcher::setScale up to its first import, after the DLL's own static initialiser of mScaleRatios;
 *   - mark_inliers / count_mask (run()'s marking loop, its grid-type loop exit, its return value): the tail of GMSMatcher::run
 *     from behind its verifyCellPairs call, for grid types 1..4 in sequence (refdll_mark.npz);
 *   - select_hypothesis (getInlierMask: loop nest and order, strict '>', the mask copy, the value returned):
 *     GMSMatcher::getInlierMask executed WHOLE with its eight calls of setScale / run re-pointed at a script player, every flag
 *     combination, eleven scripts -- ties, all zero, maximum first / last (refdll_select.npz);
 *   - the rotation-pattern table, the scale-ratio table and the 0.5 constant, byte for byte (when /root/reference exists);
 *   - END TO END (refdll_chain.npz): getInlierMask executed whole with those calls re-pointed at drivers that run the DLL's own
 *     pieces (setScale's head + initalizeNeighbors; per grid type assignMatchPairs, the verifyCellPairs body per cell, run's
 *     marking / counting tail), behind the DLL's normalizePoints -- on the inputs of all fourteen golden cases (BASELINE configs 1
 *     and 2 among them), all four flag combinations: the masks and counts equal this file's, bit for bit.
 * What the runner supplies between the DLL's pieces, and what therefore stays pinned by reading (DLL addresses cited per
 * function below): storage and Mat::zeros' allocations, the zero fills of run() (Mat::setTo(0), the three vector::assign calls),
 * the "row sum == 0" test in front of a cell's verification (cv::sum), convertMatches' copy of (queryIdx, trainIdx), and
 * matchGMS's copy-out (clear + push_back of every masked match). No arithmetic of the algorithm is among them.
 * The dense 400 x N_right motion matrix and the reference's loop order are kept on purpose, so that this file is an obviously
 * faithful, structurally independent checker for the sparse GPU formulation; a second restatement (gms_ref_sparse.py) must
 * agree with it bit for bit.
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

/* The tail of one grid type in GMSMatcher::run, DLL@0x180048ae0-0x180048b26: a match whose left cell is on the grid and whose
 * right cell is that cell's verified partner becomes an inlier (bts: OR-accumulated over the grid types). */
static void mark_inliers(const int* pair_first, const int* pair_second, const int* cell_pairs, int m, unsigned char* mask)
{
    for (int i = 0; i < m; i++)
        if (pair_first[i] >= 0 && cell_pairs[pair_first[i]] == pair_second[i]) mask[i] = 1;
}

/* run()'s return value, DLL@0x180048b46-0x180048bd4: the number of set bits of mvbInlierMask. */
static int count_mask(const unsigned char* mask, int m)
{
    int c = 0;
    for (int i = 0; i < m; i++) c += mask[i];
    return c;
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
        mark_inliers(st->pair_first, st->pair_second, st->cell_pairs, st->n_matches, st->mask);
    }
    return count_mask(st->mask, st->n_matches);
}

/* GMSMatcher::getInlierMask, DLL@0x180047dc0-0x180047fe1, on abstract setScale / run (the real ones below; scripted ones in
 * gms_ref_selftest_select): scale outer (0..4 with withScale, else 0 alone), rotation inner (1..8 with withRotation, else 1
 * alone); a hypothesis replaces the best one -- its mask is copied -- on a strictly larger count, starting from 0. Without either
 * flag the DLL copies the mask of its one run unconditionally (DLL@0x180047df4-0x180047e00): with a count of 0 that mask is all
 * false, which is what best_mask starts as. Returns 0, or the error a callback reports. */
typedef struct {
    void* ctx;
    int (*set_scale)(void* ctx, int scale);                                /* 0 = ok */
    int (*run)(void* ctx, int rotation_type, const unsigned char** mask);  /* the count, *mask = the run's mask; < 0 = error */
} hypothesis_ops;

static int select_hypothesis(const hypothesis_ops* ops, int with_rotation, int with_scale, int m, unsigned char* best_mask,
                             int* max_inlier, int* best_scale, int* best_rot)
{
    *max_inlier = 0;
    *best_scale = *best_rot = -1;
    if (m > 0) memset(best_mask, 0, (size_t)m);
    const int n_scales = with_scale ? 5 : 1, n_rots = with_rotation ? 8 : 1;
    for (int scale = 0; scale < n_scales; scale++) {
        const int rc = ops->set_scale(ops->ctx, scale);
        if (rc) return rc;
        for (int rot = 1; rot <= n_rots; rot++) {
            const unsigned char* mask = NULL;
            const int num_inlier = ops->run(ops->ctx, rot, &mask);
            if (num_inlier < 0) return num_inlier;
            if (num_inlier > *max_inlier) {
                memcpy(best_mask, mask, (size_t)m);
                *max_inlier = num_inlier;
                *best_scale = scale;
                *best_rot = rot;
            } else if (!with_rotation && !with_scale && m > 0) {
                memcpy(best_mask, mask, (size_t)m);  /* the one run's mask, whatever its count (all false when the count is 0) */
            }
        }
    }
    return 0;
}

static int real_set_scale(void* ctx, int scale) { return set_scale((gms_ref_state*)ctx, scale) ? GMS_ERR_BAD_ARG : 0; }
static int real_run(void* ctx, int rotation_type, const unsigned char** mask)
{
    gms_ref_state* st = (gms_ref_state*)ctx;
    const int c = run(st, rotation_type);
    *mask = st->mask;
    return st->domain_error ? GMS_ERR_DOMAIN : c;
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

    /* GMSMatcher::getInlierMask (select_hypothesis above) on this pair's setScale / run */
    int max_inlier = 0, best_scale = -1, best_rot = -1;
    {
        const hypothesis_ops ops = {st, real_set_scale, real_run};
        rc = select_hypothesis(&ops, with_rotation, with_scale, m, best_mask, &max_inlier, &best_scale, &best_rot);
        if (rc) goto done;
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

/* Test hook: run()'s marking loop and its return value for one grid type (mark_inliers / count_mask above) on a hand-laid-out
 * state, the mask accumulating over calls -- compared with the DLL's own tail of run() (tests/golden/refdll_mark.npz). */
int gms_ref_selftest_mark(const int* pairs /* m x 2 */, const int* cell_pairs /* 400 */, int m, unsigned char* mask /* m, in/out */)
{
    int* first = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int* second = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    if (!first || !second) {
        free(first);
        free(second);
        return -1;
    }
    for (int i = 0; i < m; i++) {
        first[i] = pairs[2 * i];
        second[i] = pairs[2 * i + 1];
    }
    mark_inliers(first, second, cell_pairs, m, mask);
    free(first);
    free(second);
    return count_mask(mask, m);
}

/* Test hook: getInlierMask's loop nest (select_hypothesis above, the one the real path runs) on SCRIPTED run() results --
 * counts[5][8] and masks[5][8][m] -- with a log of the calls it makes (100 + s = setScale(s), r = run(r)); compared with the DLL's
 * own getInlierMask driven by the same script (tests/golden/refdll_select.npz). */
typedef struct {
    const int* counts;
    const unsigned char* masks;
    int m, scale, n_log;
    int* log;
} script_ctx;
static int script_set_scale(void* ctx, int scale)
{
    script_ctx* c = (script_ctx*)ctx;
    c->scale = scale;
    if (c->n_log < 96) c->log[c->n_log++] = 100 + scale;
    return 0;
}
static int script_run(void* ctx, int rotation_type, const unsigned char** mask)
{
    script_ctx* c = (script_ctx*)ctx;
    if (c->n_log < 96) c->log[c->n_log++] = rotation_type;
    const int idx = c->scale * 8 + rotation_type - 1;
    *mask = c->masks + (size_t)idx * (size_t)c->m;
    return c->counts[idx];
}
int gms_ref_selftest_select(int with_rotation, int with_scale, int m, const int* counts, const unsigned char* masks,
                            unsigned char* best_mask, int* best /* count, scale, rot */, int* log /* 96 */, int* n_log)
{
    script_ctx c = {counts, masks, m, 0, 0, log};
    const hypothesis_ops ops = {&c, script_set_scale, script_run};
    for (int i = 0; i < 96; i++) log[i] = -1;
    const int rc = select_hypothesis(&ops, with_rotation, with_scale, m, best_mask, &best[0], &best[1], &best[2]);
    *n_log = c.n_log;
    return rc;
}
