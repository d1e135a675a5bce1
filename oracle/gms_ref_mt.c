/*
 * oracle/gms_ref_mt.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * Batch driver for the CPU restatement: pairs are independent, so the CPU
 * baseline runs one pair per host thread; each pair is the serial reference algorithm (gms_ref.c, unchanged
 * arithmetic and loop order). Every thread keeps ONE scratch state for all its pairs (gms_ref_match_ws) and takes
 * its work from an atomic counter: no allocator call and no lock per pair, so the baseline scales with the cores
 * (round 2's version malloc'ed a 640 KB .. 2.5 MB motion matrix per hypothesis -- served by mmap/munmap, one
 * process-wide kernel lock -- and took a mutex per pair: 256 threads were slower than 16).
 */
#include "gms_ref.h"

#include <pthread.h>
#include <stdlib.h>

typedef struct {
    const gms_keypoint* kp_all;
    const int64_t* frame_off;
    const int32_t* wh;
    const gms_pair* pairs;
    int n_pairs;
    const gms_dmatch* matches;
    int rot, scale;
    double thr;
    gms_dmatch* out;
    gms_pair_result* results;
    unsigned char* mask;
    int next;         /* shared work counter (atomic fetch-add) */
    int failed;
} batch_job;

static void* worker(void* arg)
{
    batch_job* job = (batch_job*)arg;
    gms_ref_scratch* ws = gms_ref_scratch_create();
    if (!ws) {
        __atomic_fetch_add(&job->failed, 1, __ATOMIC_RELAXED);
        return NULL;
    }
    for (;;) {
        int i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (i >= job->n_pairs) break;
        const gms_pair* p = &job->pairs[i];
        int64_t oa = job->frame_off[p->frame_a], ob = job->frame_off[p->frame_b];
        int na = (int)(job->frame_off[p->frame_a + 1] - oa), nb = (int)(job->frame_off[p->frame_b + 1] - ob);
        int n_out = 0;
        gms_pair_result r;
        int rc = gms_ref_match_ws(ws, job->kp_all + oa, na, job->wh[2 * p->frame_a], job->wh[2 * p->frame_a + 1],
                                  job->kp_all + ob, nb, job->wh[2 * p->frame_b], job->wh[2 * p->frame_b + 1],
                                  job->matches + p->match_off, p->m, job->rot, job->scale, job->thr,
                                  job->out + p->match_off, &n_out,
                                  job->mask ? job->mask + p->match_off : NULL, &r);
        if (job->results) job->results[i] = r;
        if (rc != GMS_OK) __atomic_fetch_add(&job->failed, 1, __ATOMIC_RELAXED);
    }
    gms_ref_scratch_destroy(ws);
    return NULL;
}

int gms_ref_batch(const gms_keypoint* kp_all, const int64_t* frame_off, const int32_t* wh, int n_frames,
                  const gms_pair* pairs, int n_pairs, const gms_dmatch* matches,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, gms_pair_result* results, unsigned char* mask, int n_threads)
{
    (void)n_frames;
    batch_job job = {kp_all, frame_off, wh, pairs, n_pairs, matches, with_rotation, with_scale,
                     threshold_factor, out, results, mask, 0, 0};
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 512) n_threads = 512;
    pthread_t th[512];
    int started = 1;
    for (int t = 1; t < n_threads; t++) {
        if (pthread_create(&th[started], NULL, worker, &job) != 0) break;  /* fewer threads, same work */
        started++;
    }
    worker(&job);
    for (int t = 1; t < started; t++) pthread_join(th[t], NULL);
    return job.failed;
}
