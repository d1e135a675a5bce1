/*
 * oracle/gms_ref_mt.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * Batch driver for the CPU restatement: pairs are independent, so the CPU
 * baseline runs one pair per host thread; each pair is the serial reference algorithm.
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
    int next;         /* shared work counter */
    int failed;
    pthread_mutex_t mu;
} batch_job;

static void* worker(void* arg)
{
    batch_job* job = (batch_job*)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        int i = job->next++;
        pthread_mutex_unlock(&job->mu);
        if (i >= job->n_pairs) break;
        const gms_pair* p = &job->pairs[i];
        int64_t oa = job->frame_off[p->frame_a], ob = job->frame_off[p->frame_b];
        int na = (int)(job->frame_off[p->frame_a + 1] - oa), nb = (int)(job->frame_off[p->frame_b + 1] - ob);
        int n_out = 0;
        gms_pair_result r;
        int rc = gms_ref_match(job->kp_all + oa, na, job->wh[2 * p->frame_a], job->wh[2 * p->frame_a + 1],
                               job->kp_all + ob, nb, job->wh[2 * p->frame_b], job->wh[2 * p->frame_b + 1],
                               job->matches + p->match_off, p->m, job->rot, job->scale, job->thr,
                               job->out + p->match_off, &n_out,
                               job->mask ? job->mask + p->match_off : NULL, &r);
        if (job->results) job->results[i] = r;
        if (rc != GMS_OK) {
            pthread_mutex_lock(&job->mu);
            job->failed++;
            pthread_mutex_unlock(&job->mu);
        }
    }
    return NULL;
}

int gms_ref_batch(const gms_keypoint* kp_all, const int64_t* frame_off, const int32_t* wh, int n_frames,
                  const gms_pair* pairs, int n_pairs, const gms_dmatch* matches,
                  int with_rotation, int with_scale, double threshold_factor,
                  gms_dmatch* out, gms_pair_result* results, unsigned char* mask, int n_threads)
{
    (void)n_frames;
    batch_job job = {kp_all, frame_off, wh, pairs, n_pairs, matches, with_rotation, with_scale,
                     threshold_factor, out, results, mask, 0, 0, PTHREAD_MUTEX_INITIALIZER};
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &job);
    worker(&job);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
    return job.failed;
}
