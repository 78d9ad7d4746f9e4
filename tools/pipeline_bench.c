/* pipeline_bench.c -- the headline measurement from a C host: no Python, no torch, only include/rmcv_abi.h.
 *
 * What a maintainer's process loop does with the library (the reference's own: executable/main.cpp:163-209), in batch form:
 *     rmcv_pipeline_create -> { rmcv_pipeline_submit } x steps -> rmcv_pipeline_drain, lists through rmcv_pipeline_collect.
 * Same workload, same frame sets, same schedule and same timing discipline as bench.py (exactly `steps` steps between two drains,
 * `repeats` such regions, the median reported), so the two figures can be held side by side (within 2 %).
 *
 *   gcc -O2 -I include tools/pipeline_bench.c -o /tmp/pipeline_bench -L rmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib -lpthread
 *   /tmp/pipeline_bench [--steps 20] [--warmup 5] [--repeats 7] [--frames 256] [--width 1280] [--height 1024] [--depth 8]
 *                       [--pixel-streams 2] [--sparse-streams 4] [--sets 8] [--variant 0] [--warmup-seconds 0.4]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rmcv_abi.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

typedef struct {
    uint8_t* host;
    int w, h, first, count, variant;
    uint64_t base;
} synth_job;

static void* synth_worker(void* arg)
{
    synth_job* j = (synth_job*)arg;
    for (int f = j->first; f < j->first + j->count; f++)
        rmcv_synth_frame(j->host + (size_t)f * 3 * j->w * j->h, j->w, j->h, 3 * j->w, j->base + (uint64_t)f, RMCV_CAMP_BLUE, j->variant);
    return NULL;
}

static int cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y;
}

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc__ = (call);                                                            \
        if (rc__ != RMCV_OK) {                                                        \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc__, pl ? rmcv_pipeline_last_error(pl) : ""); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv)
{
    int steps = 20, warmup = 5, repeats = 7, n = 256, w = 1280, h = 1024, depth = 8, pixs = 2, sps = 4, sets = 8, variant = 0, threads = 16;
    double warm_s = 0.4;
    for (int i = 1; i + 1 < argc; i += 2) {
        const char* k = argv[i];
        const char* v = argv[i + 1];
        if (!strcmp(k, "--steps")) steps = atoi(v);
        else if (!strcmp(k, "--warmup")) warmup = atoi(v);
        else if (!strcmp(k, "--repeats")) repeats = atoi(v);
        else if (!strcmp(k, "--frames")) n = atoi(v);
        else if (!strcmp(k, "--width")) w = atoi(v);
        else if (!strcmp(k, "--height")) h = atoi(v);
        else if (!strcmp(k, "--depth")) depth = atoi(v);
        else if (!strcmp(k, "--pixel-streams")) pixs = atoi(v);
        else if (!strcmp(k, "--sparse-streams")) sps = atoi(v);
        else if (!strcmp(k, "--sets")) sets = atoi(v);
        else if (!strcmp(k, "--variant")) variant = atoi(v);
        else if (!strcmp(k, "--threads")) threads = atoi(v);
        else if (!strcmp(k, "--warmup-seconds")) warm_s = atof(v);
        else { fprintf(stderr, "unknown option %s\n", k); return 2; }
    }
    rmcv_hw_queues_hint(); /* GPU_MAX_HW_QUEUES=12 unless set: before the process's first HIP call (the library does not touch the environment by itself) */
    if (sets < depth) sets = depth; /* batches that overlap in time must not share input (the 256 MB Infinity Cache would serve the second) */
    if (steps < 1 || repeats < 1 || repeats > 64 || threads < 1 || threads > 64) return 2;
    rmcv_pipeline* pl = NULL;
    const size_t frame_bytes = (size_t)3 * w * h, set_bytes = frame_bytes * (size_t)n;
    uint8_t* host = (uint8_t*)malloc(set_bytes);
    void** d_frames = (void**)calloc((size_t)sets, sizeof(void*));
    if (!host || !d_frames) return 3;
    /* frame set k of bench.py: stream indices k * 1000003 + 0 .. n-1 (rank 0) */
    for (int k = 0; k < sets; k++) {
        pthread_t th[64];
        synth_job jobs[64];
        const int per = (n + threads - 1) / threads;
        int nt = 0;
        for (int t = 0; t < threads && t * per < n; t++, nt++) {
            jobs[t] = (synth_job){host, w, h, t * per, (t + 1) * per <= n ? per : n - t * per, variant, (uint64_t)k * 1000003ull};
            pthread_create(&th[t], NULL, synth_worker, &jobs[t]);
        }
        for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
        CHECK(rmcv_device_alloc(0, (int64_t)set_bytes, &d_frames[k]));
        CHECK(rmcv_device_upload(0, d_frames[k], host, (int64_t)set_bytes)); /* resident in HBM before any timing */
    }
    free(host);

    rmcv_limits lim;
    rmcv_default_limits(&lim);
    lim.max_frames = n;
    lim.max_width = w;
    lim.max_height = h;
    if (variant >= 10) lim.max_contours = 4096;
    rmcv_pipeline_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.depth = depth;
    cfg.pixel_streams = pixs;
    cfg.sparse_streams = sps;
    CHECK(rmcv_pipeline_create(0, &lim, &cfg, &pl));
    rmcv_pipeline_info info;
    CHECK(rmcv_pipeline_get_info(pl, &info));
    rmcv_params p;
    rmcv_default_params(&p);
    uint64_t t = 0, step_no = 0;
#define STEP()                                                                                                                     \
    do {                                                                                                                           \
        CHECK(rmcv_pipeline_submit(pl, d_frames[step_no % (uint64_t)sets], n, w, h, 3 * w, (int64_t)frame_bytes, &p, RMCV_STAGE_ALL, &t)); \
        step_no++;                                                                                                                 \
    } while (0)
    for (int i = 0; i < warmup; i++) STEP();
    CHECK(rmcv_pipeline_drain(pl));
    int warm_steps = 0;
    for (double t0 = now_s(); now_s() - t0 < warm_s;) { /* by time as well: the clocks ramp */
        for (int i = 0; i < steps; i++) STEP();
        warm_steps += steps;
        CHECK(rmcv_pipeline_drain(pl));
    }
    double dt[64], enq[64];
    for (int r = 0; r < repeats; r++) {
        CHECK(rmcv_pipeline_drain(pl));
        const double t0 = now_s();
        for (int i = 0; i < steps; i++) STEP();
        enq[r] = now_s() - t0;
        CHECK(rmcv_pipeline_drain(pl));
        dt[r] = now_s() - t0;
    }
    /* one long region: the pipeline's fill and drain amortised (not the metric) */
    const int long_steps = 25 * steps;
    CHECK(rmcv_pipeline_drain(pl));
    const double tl = now_s();
    for (int i = 0; i < long_steps; i++) STEP();
    CHECK(rmcv_pipeline_drain(pl));
    const double dl = now_s() - tl;
    /* what was computed: the lists of the last `depth` batches (the tickets still in flight), by frame set */
    rmcv_armour* arm = (rmcv_armour*)malloc((size_t)info.armour_cap * sizeof(rmcv_armour));
    int32_t* offs = (int32_t*)malloc((size_t)(n + 1) * 4);
    long arm_set0 = -1, arm_all = 0;
    for (uint64_t q = step_no >= (uint64_t)depth ? step_no - (uint64_t)depth : 0; q < step_no; q++) {
        int32_t total = 0;
        CHECK(rmcv_pipeline_collect(pl, q, arm, info.armour_cap, offs, &total));
        if (q % (uint64_t)sets == 0) arm_set0 = total;
        arm_all += total;
    }
    qsort(dt, (size_t)repeats, sizeof(double), cmp_double);
    qsort(enq, (size_t)repeats, sizeof(double), cmp_double);
    const double med = repeats % 2 ? dt[repeats / 2] : 0.5 * (dt[repeats / 2 - 1] + dt[repeats / 2]);
    const double ms = med / steps * 1e3, fps = (double)n * steps / med;
    printf("{\"host\": \"C (tools/pipeline_bench.c)\", \"metric\": \"frames/sec (%dx%d BGR) armour detect\", \"value\": %.1f, \"unit\": \"frames/s\", "
           "\"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"ms_per_step_min\": %.4f, \"repeats\": %d, "
           "\"host_enqueue_ms_per_step\": %.4f, \"steady_state_ms_per_step\": %.4f, \"path_hbm_frac\": %.4f, "
           "\"depth\": %d, \"pixel_streams\": %d, \"sparse_streams\": %d, \"frame_sets\": %d, \"gpu_max_hw_queues\": %d, "
           "\"warmup_steps_by_time\": %d, \"armours_set0\": %ld, \"armours_last_%d_batches\": %ld}\n",
           w, h, fps, steps, warmup, ms, dt[0] / steps * 1e3, repeats, enq[repeats / 2] / steps * 1e3, dl / long_steps * 1e3,
           fps * 4.0 * w * h / 8e12, info.depth, info.pixel_streams, info.sparse_streams, sets, info.hw_queues_env, warm_steps, arm_set0,
           info.depth, arm_all);
    free(arm);
    free(offs);
    rmcv_pipeline_destroy(pl);
    for (int k = 0; k < sets; k++) rmcv_device_free(0, d_frames[k]);
    free(d_frames);
    return 0;
}
