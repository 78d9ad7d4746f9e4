// membw3.hip -- follow-up to membw2: which property makes the 3:1 stream fast?  (dev tool)
// Factors: coalesced vs per-lane-48B loads, non-temporal loads/stores, waves per CU (grid x block), loads in flight (U),
// and how far apart the workgroups' addresses are at any moment (WINDOW: the grid sweeps the buffer in windows of that many bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride sweep; a thread block of B threads; COAL: wave reads 3 x 1 KiB contiguous, else lane reads 48 contiguous bytes
template <int U, bool NTL, bool NTS, bool COAL>
__global__ void k31(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += step * U) {
        u32x4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) {
                if (COAL) { const size_t w0 = (j - lane) * 3; a[u] = ld<NTL>(in + w0 + lane); b[u] = ld<NTL>(in + w0 + 64 + lane); c[u] = ld<NTL>(in + w0 + 128 + lane); }
                else { a[u] = ld<NTL>(in + 3 * j); b[u] = ld<NTL>(in + 3 * j + 1); c[u] = ld<NTL>(in + 3 * j + 2); }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) st<NTS>(out + j, a[u] ^ b[u] ^ c[u]);
        }
    }
}
static hipEvent_t e0, e1;
template <typename F> static void timeit(const char* name, F launch, double bytes)
{
    std::vector<float> t;
    for (int r = 0; r < 15; r++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s median %.4f ms %5.0f GB/s   min %.4f ms %5.0f GB/s\n", name, t[7], bytes / t[7] / 1e6, t[0], bytes / t[0] / 1e6);
}
int main()
{
    const size_t out_bytes = 256ull * 1280 * 1024, in_bytes = 3 * out_bytes, n = out_bytes / 16;
    u32x4 *in, *out;
    hipMalloc(&in, in_bytes + (1 << 20));
    hipMalloc(&out, in_bytes + (1 << 20));
    hipMemset(in, 1, in_bytes);
    hipMemset(out, 0, in_bytes);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double mix = (double)in_bytes + out_bytes;
    char nm[128];
#define RUN(U, NTL, NTS, COAL, G, B)                                                                                   \
    snprintf(nm, sizeof nm, "%s U%d %s%s grid %5d x %4d", COAL ? "coalesced" : "strided  ", U, NTL ? "ntL " : "    ", NTS ? "ntS " : "    ", G, B); \
    timeit(nm, [&] { hipLaunchKernelGGL((k31<U, NTL, NTS, COAL>), dim3(G), dim3(B), 0, 0, in, n, out); }, mix);
    for (int rep = 0; rep < 2; rep++) {
        RUN(4, true, true, true, 256, 256)
        RUN(4, true, true, true, 256, 512)
        RUN(4, true, true, true, 256, 1024)
        RUN(4, true, true, true, 128, 256)
        RUN(4, true, true, true, 384, 256)
        RUN(4, true, true, true, 512, 256)
        RUN(4, true, true, true, 768, 256)
        RUN(4, true, true, true, 1024, 256)
        RUN(2, true, true, true, 256, 256)
        RUN(2, true, true, true, 512, 256)
        RUN(8, true, true, true, 256, 256)
        RUN(1, true, true, true, 1024, 256)
        RUN(4, true, false, true, 256, 256)
        RUN(4, false, true, true, 256, 256)
        RUN(4, false, false, true, 256, 256)
        RUN(4, true, true, false, 256, 256)
        RUN(4, false, true, false, 256, 256)
        RUN(4, false, true, false, 512, 256)
        RUN(4, false, true, false, 1024, 256)
        RUN(2, false, true, false, 1024, 256)
    }
    return 0;
}
