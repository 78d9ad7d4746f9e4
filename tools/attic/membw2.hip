// membw2.hip -- HBM ceilings of this part for k_binary's traffic (dev tool): a sweep over launch shape, loads in flight,
// streaming hints and the workgroup->address mapping, for  pure read / pure write / 1:1 copy / the 3:1 read:write mix.
//   hipcc -O3 --offload-arch=gfx950 -o membw2 membw2.hip && ./membw2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// item i: read 48 B at 48*i (lane stride 48 B), write 16 B at 16*i.  U items in flight per thread.
// MAP 0: grid-stride (all workgroups sweep the buffer together); MAP 1: each workgroup owns a contiguous range (k_binary's strips)
template <int U, bool NTL, bool NTS, int MAP>
__global__ __launch_bounds__(256) void k31_strided(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    size_t i0, step, end;
    if (MAP == 0) { i0 = blockIdx.x * (size_t)256 + threadIdx.x; step = (size_t)gridDim.x * 256; end = n; }
    else { const size_t per = (n + gridDim.x - 1) / gridDim.x; i0 = blockIdx.x * per + threadIdx.x; step = 256; end = (blockIdx.x + 1) * per < n ? (blockIdx.x + 1) * per : n; }
    for (size_t i = i0; i < end; i += step * U) {
        u32x4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < end) { a[u] = ld<NTL>(in + 3 * j); b[u] = ld<NTL>(in + 3 * j + 1); c[u] = ld<NTL>(in + 3 * j + 2); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < end) st<NTS>(out + j, a[u] ^ b[u] ^ c[u]);
        }
    }
}
// wave-coalesced: a wave reads 3 x 1 KiB contiguous (lane stride 16 B) and writes 1 KiB
template <int U, bool NTL, bool NTS, int MAP>
__global__ __launch_bounds__(256) void k31_coal(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    size_t i0, step, end;
    if (MAP == 0) { i0 = blockIdx.x * (size_t)256 + threadIdx.x; step = (size_t)gridDim.x * 256; end = n; }
    else { const size_t per = ((n + gridDim.x - 1) / gridDim.x + 255) & ~(size_t)255; i0 = blockIdx.x * per + threadIdx.x; step = 256; end = (blockIdx.x + 1) * per < n ? (blockIdx.x + 1) * per : n; }
    for (size_t i = i0; i < end; i += step * U) {
        u32x4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            const size_t w0 = (j - lane) * 3;
            if (j < end) { a[u] = ld<NTL>(in + w0 + lane); b[u] = ld<NTL>(in + w0 + 64 + lane); c[u] = ld<NTL>(in + w0 + 128 + lane); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < end) st<NTS>(out + j, a[u] ^ b[u] ^ c[u]);
        }
    }
}
template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    u32x4 acc = {0, 0, 0, 0};
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += step * U) {
        u32x4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * step < n) a[u] = ld<NTL>(in + i + u * step); else a[u] = acc;
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= a[u];
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}
template <bool NTS>
__global__ __launch_bounds__(256) void k_write(size_t n, u32x4* __restrict__ out)
{
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += step) { u32x4 v = {(uint32_t)i, 1, 2, 3}; st<NTS>(out + i, v); }
}
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += step * U) {
        u32x4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * step < n) a[u] = ld<NTL>(in + i + u * step);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * step < n) st<NTS>(out + i + u * step, a[u]);
    }
}

static hipEvent_t e0, e1;
template <typename F> static float best_ms(F launch, int reps = 7)
{
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char** argv)
{
    const size_t out_bytes = 256ull * 1280 * 1024, in_bytes = 3 * out_bytes; // one batch of k_binary
    const size_t n_out = out_bytes / 16;
    u32x4 *in, *out;
    CK(hipMalloc(&in, in_bytes + (1 << 20)));
    CK(hipMalloc(&out, in_bytes + (1 << 20)));
    CK(hipMemset(in, 1, in_bytes));
    CK(hipMemset(out, 0, in_bytes));
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384};
    printf("# one batch = 1.0066 GB read + 0.3355 GB written; GB/s = (bytes moved) / best-of-7 time\n");
    for (int g : grids) {
        float r1 = best_ms([&] { hipLaunchKernelGGL((k_read<1, false>), dim3(g), dim3(256), 0, 0, in, in_bytes / 16, out); });
        float r4 = best_ms([&] { hipLaunchKernelGGL((k_read<4, false>), dim3(g), dim3(256), 0, 0, in, in_bytes / 16, out); });
        float r4n = best_ms([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(g), dim3(256), 0, 0, in, in_bytes / 16, out); });
        float w = best_ms([&] { hipLaunchKernelGGL((k_write<false>), dim3(g), dim3(256), 0, 0, n_out, out); });
        float wn = best_ms([&] { hipLaunchKernelGGL((k_write<true>), dim3(g), dim3(256), 0, 0, n_out, out); });
        float c1 = best_ms([&] { hipLaunchKernelGGL((k_copy<1, false, false>), dim3(g), dim3(256), 0, 0, in, 2 * n_out, out); });
        float c4 = best_ms([&] { hipLaunchKernelGGL((k_copy<4, false, false>), dim3(g), dim3(256), 0, 0, in, 2 * n_out, out); });
        float c4n = best_ms([&] { hipLaunchKernelGGL((k_copy<4, false, true>), dim3(g), dim3(256), 0, 0, in, 2 * n_out, out); });
        float c4nn = best_ms([&] { hipLaunchKernelGGL((k_copy<4, true, true>), dim3(g), dim3(256), 0, 0, in, 2 * n_out, out); });
        printf("grid %5d  read U1 %5.0f U4 %5.0f U4nt %5.0f | write %5.0f nt %5.0f | copy(0.67+0.67GB) U1 %5.0f U4 %5.0f U4/ntS %5.0f U4/ntLS %5.0f GB/s\n", g,
               in_bytes / r1 / 1e6, in_bytes / r4 / 1e6, in_bytes / r4n / 1e6, out_bytes / w / 1e6, out_bytes / wn / 1e6, 4.0 * out_bytes / c1 / 1e6,
               4.0 * out_bytes / c4 / 1e6, 4.0 * out_bytes / c4n / 1e6, 4.0 * out_bytes / c4nn / 1e6);
    }
    const double mix = (double)in_bytes + out_bytes;
#define ROW(NAME, K)                                                                                               \
    for (int g : grids) {                                                                                          \
        float t = best_ms([&] { hipLaunchKernelGGL((K), dim3(g), dim3(256), 0, 0, in, n_out, out); });            \
        printf("3:1 %-34s grid %5d  %.4f ms  %5.0f GB/s\n", NAME, g, t, mix / t / 1e6);                            \
    }
    ROW("strided U1 plain", (k31_strided<1, false, false, 0>))
    ROW("strided U2 plain", (k31_strided<2, false, false, 0>))
    ROW("strided U4 plain", (k31_strided<4, false, false, 0>))
    ROW("strided U4 ntS", (k31_strided<4, false, true, 0>))
    ROW("strided U4 ntL+ntS", (k31_strided<4, true, true, 0>))
    ROW("strided U8 ntS", (k31_strided<8, false, true, 0>))
    ROW("strided U4 ntS contiguous-per-WG", (k31_strided<4, false, true, 1>))
    ROW("coalesced U1 plain", (k31_coal<1, false, false, 0>))
    ROW("coalesced U2 ntS", (k31_coal<2, false, true, 0>))
    ROW("coalesced U4 ntS", (k31_coal<4, false, true, 0>))
    ROW("coalesced U4 ntL+ntS", (k31_coal<4, true, true, 0>))
    ROW("coalesced U8 ntS", (k31_coal<8, false, true, 0>))
    ROW("coalesced U4 ntS contiguous-per-WG", (k31_coal<4, false, true, 1>))
    // output buffer offset against the input (channel/bank alignment of the two streams)
    for (size_t off : {(size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)(1 << 19)}) {
        float t = best_ms([&] { hipLaunchKernelGGL((k31_coal<4, false, true, 0>), dim3(2048), dim3(256), 0, 0, in, n_out, out + off / 16); });
        printf("3:1 coalesced U4 ntS grid 2048, out + %7zu B: %.4f ms %5.0f GB/s\n", off, t, mix / t / 1e6);
    }
    return 0;
}
