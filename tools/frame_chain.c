/* dev tool: the per-frame drop-in chain timed from C -- what a C/C++ host sees, without the Python call overhead of tools/frame_chain.py.
 *   gcc -O2 -Iinclude tools/frame_chain.c -o /tmp/frame_chain -Lrmcv_amd/lib -lrmcv_hip -Wl,-rpath,$PWD/rmcv_amd/lib && /tmp/frame_chain [W H]
 * Three calls per frame exactly as include/rmcv_shim.hpp issues them for executable/main.cpp:172-176, results on the host after each. */
#define _GNU_SOURCE
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

#include "rmcv_abi.h"

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}
/* FC_ALLOC=malloc (default) | huge | nohuge: how the caller's frame and byte-image buffers are backed (2 MiB-aligned + madvise) -- the
 * runtime's pageable copies pin or stage them, and what that costs depends on the pages behind them */
static void* buf_alloc(size_t bytes)
{
    const char* m = getenv("FC_ALLOC");
    if (!m || !strcmp(m, "malloc")) return malloc(bytes);
    void* p = NULL;
    const size_t sz = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (posix_memalign(&p, 2u << 20, sz)) return NULL;
    madvise(p, sz, !strcmp(m, "huge") ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
    memset(p, 0, sz);
    return p;
}
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

int main(int argc, char** argv)
{
    const int W = argc > 2 ? atoi(argv[1]) : 1280, H = argc > 2 ? atoi(argv[2]) : 1024, N = 300;
    rmcv_limits lim;
    rmcv_default_limits(&lim);
    lim.max_frames = 1; lim.max_width = W; lim.max_height = H;
    rmcv_ctx* c = NULL;
    if (rmcv_ctx_create(0, &lim, &c)) { fprintf(stderr, "no context\n"); return 2; }
    uint8_t* img[4];
    for (int i = 0; i < 4; i++) { img[i] = buf_alloc((size_t)3 * W * H); rmcv_synth_frame(img[i], W, H, 3 * W, (uint64_t)i, 1, 0); }
    uint8_t* binary = buf_alloc((size_t)W * H);
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        sched_getaffinity(0, sizeof(set), &set);
        printf("cpu %d of %d allowed, FC_ALLOC=%s, binary at %p\n", sched_getcpu(), CPU_COUNT(&set), getenv("FC_ALLOC") ? getenv("FC_ALLOC") : "malloc", (void*)binary);
    }
    rmcv_point* pts = malloc(sizeof(rmcv_point) * 65536);
    int32_t* offs = malloc(4 * 2049);
    rmcv_lightblob* blobs = malloc(sizeof(rmcv_lightblob) * 256);
    int32_t* neg = malloc(4 * 2048);
    rmcv_armour* arms = malloc(sizeof(rmcv_armour) * 256);
    static double tot[300], ec[300], st[7][300];
    static const char* st_name[7] = {"sync+bind+upload", "enqueue kernels", "first image chunk", "image chunks", "runtime copy", "wait kernels", "hand over"};
    for (int mode = 0; mode <= 2; mode += 2) {
        rmcv_ctx_set_option(c, RMCV_OPT_FRAME_UPLOAD, mode == 0 ? (getenv("RMCV_FRAME_UPLOAD") ? atoi(getenv("RMCV_FRAME_UPLOAD")) : 3) : mode);
        int32_t nc = 0, np = 0, nb = 0, nn = 0, na = 0;
        int n_up1 = 0, n_img1 = 0; /* chains that took the pinned staging buffer / the export kernel */
        for (int i = -8; i < N; i++) {
            const uint8_t* f = img[(i + 8) % 4];
            const double t0 = now_ms();
            int rc = rmcv_extract_color(c, f, W, H, 3 * W, RMCV_CAMP_BLUE, 80, RMCV_MORPH_CLOSE, binary, pts, 65536, offs, 2048, &nc, &np);
            const double t1 = now_ms();
            double us[9] = {0};
            rmcv_ctx_frame_timing(c, us, 9);
            if (i >= 0) { n_up1 += us[7] == 1; n_img1 += us[8] == 1; }
            if (i >= 0) for (int k = 0; k < 7; k++) st[k][i] = us[k];
            rc |= rmcv_filter_lightblobs(c, pts, offs, nc, 70.0f, 1.5f, 80.0f, 10.0, 99999.0, RMCV_CAMP_BLUE, blobs, 256, &nb, NULL, neg, &nn);
            rc |= rmcv_filter_armours(c, blobs, nb, 12.0f, 22.0f, 0.4f, RMCV_CAMP_BLUE, arms, 256, &na);
            const double t2 = now_ms();
            if (rc) { fprintf(stderr, "rc %d: %s\n", rc, rmcv_last_error(c)); return 3; }
            if (i >= 0) { tot[i] = t2 - t0; ec[i] = t1 - t0; }
        }
        qsort(tot, N, sizeof(double), cmp);
        qsort(ec, N, sizeof(double), cmp);
        printf("%-20s median %.4f  min %.4f  p90 %.4f ms | extract_color median %.4f | contours %d blobs %d armours %d\n",
               mode == 0 ? "default_upload" : "registered_in_place", tot[N / 2], tot[0], tot[9 * N / 10], ec[N / 2], nc, nb, na);
        printf("    extract_color on the host, medians (p90) in us:");
        for (int k = 0; k < 7; k++) {
            qsort(st[k], N, sizeof(double), cmp);
            printf("  %s %.1f (%.1f)", st_name[k], st[k][N / 2], st[k][9 * N / 10]);
        }
        printf("  | chains on the staging buffer %d, on the export kernel %d of %d\n", n_up1, n_img1, N);
    }
    rmcv_ctx_destroy(c);
    return 0;
}
