/* mslam_cpu_bench.c — the timed CPU leg (bench.py: cpu_baseline): the oracle's detect + match loop on the host cores.
 *
 * TEST INFRASTRUCTURE (part of oracle/libmslam_oracle.so), not product code.
 *
 * What is timed is what the reference's frontend does per frame (rgbd_feature_frontend.cpp:187,237): detect(frame t),
 * then match(from = keypoints t, to = keypoints t-1), built the way the reference builds it: scalar, one thread per
 * stream (distributed_cv_feature.cpp:855-857: USE_OPENMP is not defined by its CMake).  Frames are sharded over plain
 * pthreads — every thread owns a contiguous block of the stream and runs one mso_detect + mso_match loop over it, no
 * shared state, no locks, results kept per thread; all threads leave a start gate together and the wall time is taken from
 * the opening of that gate to the last join.
 *
 * The per-frame buffers of mso_detect are 0.3-1 MB each: above glibc's default mmap threshold every malloc / free of them
 * is an mmap / munmap, i.e. a write lock on the process's address space shared by all threads, plus fresh page faults per
 * frame.  mso_bench_stream() raises the thresholds once (mallopt) so that each thread re-uses its own arena instead: with
 * the defaults the all-core figure measures the kernel's mm lock, not the algorithm.
 */
#define _GNU_SOURCE
#include "mslam_oracle.h"
#include <malloc.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* start gate: the threads allocate their buffers, then wait here; the caller opens it (go = 1) once every thread exists, or
 * sends them home (go = -1) when one could not be created */
typedef struct mso_bench_gate
{
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int go, ready;
} mso_bench_gate;

typedef struct
{
    const uint8_t* frames; /* n_unique back-to-back BGR8 frames */
    int n_unique, W, H;
    int first, count;      /* this thread's block: stream positions first .. first + count - 1 (taken mod n_unique) */
    const mso_orb_params* p;
    const mso_cvorb_params* cvp; /* non-NULL: the cv::ORB detector mode */
    int max_kp;
    double ratio;
    struct mso_bench_gate* gate;
    /* results */
    double keypoints, matches, seconds;
    int rc;
} mso_bench_job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* bench_thread(void* arg)
{
    mso_bench_job* j = (mso_bench_job*)arg;
    const int K = j->max_kp;
    float* xy[2];
    uint8_t* desc[2];
    int n[2] = {0, 0};
    int32_t* octave = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
    float* angle = (float*)malloc(sizeof(float) * (size_t)K);
    float* resp = (float*)malloc(sizeof(float) * (size_t)K);
    int32_t* fi = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
    for(int s = 0; s < 2; ++s)
    {
        xy[s] = (float*)malloc(sizeof(float) * 2 * (size_t)K);
        desc[s] = (uint8_t*)malloc(32 * (size_t)K);
    }
    const size_t fb = (size_t)j->W * j->H * 3;
    j->rc = 0;
    /* an allocation that failed: the thread still goes through the gate (the caller counts it) but runs no frame and reports -3 */
    const int alloc_ok = octave && angle && resp && fi && ti && xy[0] && xy[1] && desc[0] && desc[1];
    if(!alloc_ok)
        j->rc = -3;
    pthread_mutex_lock(&j->gate->mu);
    ++j->gate->ready;
    pthread_cond_broadcast(&j->gate->cv);
    while(j->gate->go == 0)
        pthread_cond_wait(&j->gate->cv, &j->gate->mu);
    const int go = j->gate->go;
    pthread_mutex_unlock(&j->gate->mu);
    const double t0 = now_s();
    for(int i = 0; go > 0 && alloc_ok && i < j->count; ++i)
    {
        const int cur = i & 1, prev = cur ^ 1;
        const uint8_t* f = j->frames + (size_t)((j->first + i) % j->n_unique) * fb;
        int rc;
        if(j->cvp)
            rc = mso_cvorb_detect(f, j->W, j->H, j->cvp, K, xy[cur], desc[cur], octave, angle, resp, &n[cur]);
        else
            rc = mso_detect(f, j->W, j->H, j->p, K, xy[cur], desc[cur], octave, angle, resp, &n[cur]);
        if(rc != 0)
        {
            j->rc = rc;
            n[cur] = K;
        }
        j->keypoints += n[cur];
        if(i > 0) /* orb_feature.cpp:84-117 through rgbd_feature_frontend.cpp:237: match(from = current, to = previous) */
            j->matches += mso_match(desc[cur], n[cur], desc[prev], n[prev], j->ratio, fi, ti);
    }
    j->seconds = now_s() - t0;
    for(int s = 0; s < 2; ++s)
    {
        free(xy[s]);
        free(desc[s]);
    }
    free(octave);
    free(angle);
    free(resp);
    free(fi);
    free(ti);
    return NULL;
}

/* Runs n_threads x frames_per_thread frames of the stream (thread t takes positions t * frames_per_thread ..., modulo
 * n_unique).  out[0] = keypoints detected, out[1] = matches kept, out[2] = wall seconds (gate -> last join),
 * out[3] / out[4] = shortest / longest thread loop in seconds.  Returns 0, -1 on a capacity overflow in some thread,
 * -2 when threads could not be started. */
int mso_bench_stream(const uint8_t* frames, int n_unique, int W, int H, const mso_orb_params* p,
                     const mso_cvorb_params* cvp, int n_threads, int frames_per_thread, int max_kp, double out[5])
{
    if(n_threads < 1 || frames_per_thread < 1 || n_unique < 1)
        return -2;
    /* process-wide allocator settings, for the duration of this call only (the caller is a Python process that goes on to do
     * other things): raised here, put back to glibc's documented defaults before returning (128 KB thresholds, arena limit
     * chosen by the library).  What cannot be restored is glibc's DYNAMIC adjustment of the mmap threshold, which any explicit
     * setting switches off for the rest of the process: a footnote for a test / bench process, stated here. */
    mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024); /* glibc's upper limit for this knob */
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_ARENA_MAX, 4096);
#define MSO_RESTORE_MALLOPT()                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        mallopt(M_MMAP_THRESHOLD, 128 * 1024);                                                                         \
        mallopt(M_TRIM_THRESHOLD, 128 * 1024);                                                                         \
        mallopt(M_ARENA_MAX, 0);                                                                                       \
    } while(0)
    mso_bench_job* jobs = (mso_bench_job*)calloc((size_t)n_threads, sizeof(mso_bench_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    if(!jobs || !th)
    {
        free(jobs);
        free(th);
        MSO_RESTORE_MALLOPT();
        return -2;
    }
    mso_bench_gate gate;
    pthread_mutex_init(&gate.mu, NULL);
    pthread_cond_init(&gate.cv, NULL);
    gate.go = 0;
    gate.ready = 0;
    int started = 0;
    for(int t = 0; t < n_threads; ++t)
    {
        mso_bench_job* j = &jobs[t];
        j->frames = frames;
        j->n_unique = n_unique;
        j->W = W;
        j->H = H;
        j->first = (int)(((long long)t * frames_per_thread) % n_unique);
        j->count = frames_per_thread;
        j->p = p;
        j->cvp = cvp;
        j->max_kp = max_kp;
        j->ratio = 0.7; /* orb_feature.cpp:101 */
        j->gate = &gate;
        if(pthread_create(&th[t], NULL, bench_thread, j) != 0)
            break;
        ++started;
    }
    /* wait until every started thread stands at the gate (its buffers allocated), then open it */
    pthread_mutex_lock(&gate.mu);
    while(gate.ready < started)
        pthread_cond_wait(&gate.cv, &gate.mu);
    gate.go = started == n_threads ? 1 : -1;
    pthread_cond_broadcast(&gate.cv);
    pthread_mutex_unlock(&gate.mu);
    const double t0 = now_s();
    if(started != n_threads)
    {
        for(int t = 0; t < started; ++t)
            pthread_join(th[t], NULL);
        pthread_cond_destroy(&gate.cv);
        pthread_mutex_destroy(&gate.mu);
        free(jobs);
        free(th);
        MSO_RESTORE_MALLOPT();
        return -2;
    }
    for(int t = 0; t < n_threads; ++t)
        pthread_join(th[t], NULL);
    const double wall = now_s() - t0;
    int rc = 0;
    double kp = 0, m = 0, smin = 1e300, smax = 0;
    for(int t = 0; t < n_threads; ++t)
    {
        kp += jobs[t].keypoints;
        m += jobs[t].matches;
        if(jobs[t].seconds < smin)
            smin = jobs[t].seconds;
        if(jobs[t].seconds > smax)
            smax = jobs[t].seconds;
        if(jobs[t].rc == -3)
            rc = -2; /* a thread could not allocate its buffers */
        else if(jobs[t].rc != 0 && rc == 0)
            rc = -1;
    }
    out[0] = kp;
    out[1] = m;
    out[2] = wall;
    out[3] = smin;
    out[4] = smax;
    pthread_cond_destroy(&gate.cv);
    pthread_mutex_destroy(&gate.mu);
    free(jobs);
    free(th);
    MSO_RESTORE_MALLOPT();
#undef MSO_RESTORE_MALLOPT
    return rc;
}
