// membw4.hip -- the 3:1 read:write stream of one batch (1.007 GB in, 0.336 GB out) WARM and COLD.  (dev tool)
// membw2/membw3 launched every repetition over the SAME buffers: part of a gigabyte that was read a moment ago is still in the
// 256 MB Infinity Cache, and so were their "ceilings".  Here the launches rotate over SETS distinct buffer pairs (SETS = 1: the old
// way), on one stream back to back and alternating over two streams (what bench.py's pixel streams do).
//   hipcc -O3 --offload-arch=gfx950 tools/membw4.hip -o tools/membw4 && tools/membw4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// wave reads 3 x 1 KiB contiguous, writes 1 KiB; grid-stride
template <int U, bool NT>
__global__ void k31(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += step * U) {
        u32x4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) { const size_t w0 = (j - lane) * 3; a[u] = ld<NT>(in + w0 + lane); b[u] = ld<NT>(in + w0 + 64 + lane); c[u] = ld<NT>(in + w0 + 128 + lane); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) st<NT>(out + j, a[u] ^ b[u] ^ c[u]);
        }
    }
}
// the same traffic, but every workgroup walks its OWN contiguous chunk of CH 16-byte outputs (3 x CH inputs) at a time, chunks dealt
// round-robin -- the shape of k_binary's strips (a strip = 36 contiguous rows = 138 KB of input): hundreds of separate address streams
template <int U, bool NT>
__global__ void k31_chunk(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out, size_t CH)
{
    const int lane = threadIdx.x & 63;
    for (size_t c0 = blockIdx.x * CH; c0 < n; c0 += (size_t)gridDim.x * CH) {
        const size_t c1 = c0 + CH < n ? c0 + CH : n;
        for (size_t i = c0 + threadIdx.x; i < c1; i += (size_t)blockDim.x * U) {
            u32x4 a[U], b[U], c[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t j = i + u * blockDim.x;
                if (j < c1) { const size_t w0 = (j - lane) * 3; a[u] = ld<NT>(in + w0 + lane); b[u] = ld<NT>(in + w0 + 64 + lane); c[u] = ld<NT>(in + w0 + 128 + lane); }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t j = i + u * blockDim.x;
                if (j < c1) st<NT>(out + j, a[u] ^ b[u] ^ c[u]);
            }
        }
    }
}
// pure read (sum into one word per workgroup) and pure write, same traffic shapes
template <int U>
__global__ void kread(const u32x4* __restrict__ in, size_t n, uint32_t* __restrict__ out)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += step * U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) { u32x4 v = ld<true>(in + j); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
int main()
{
    const size_t out_bytes = 256ull * 1280 * 1024, in_bytes = 3 * out_bytes, n = out_bytes / 16;
    const int MAXS = 4;
    u32x4 *in[MAXS], *out[MAXS];
    for (int s = 0; s < MAXS; s++) {
        hipMalloc(&in[s], in_bytes + (1 << 20));
        hipMalloc(&out[s], out_bytes + (1 << 20));
        hipMemset(in[s], s + 1, in_bytes);
        hipMemset(out[s], 0, out_bytes);
    }
    hipStream_t st[2];
    hipStreamCreate(&st[0]);
    hipStreamCreate(&st[1]);
    hipDeviceSynchronize();
    const double bytes = (double)in_bytes + out_bytes;
    auto run = [&](const char* name, int sets, int streams, int grid, int block, int which) {
        double best = 1e9, med;
        std::vector<double> t;
        for (int rep = 0; rep < 7; rep++) {
            hipDeviceSynchronize();
            const int K = 40;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; i++) {
                const int s = i % sets;
                hipStream_t q = st[i % streams];
                if (which == 0) hipLaunchKernelGGL((k31<2, true>), dim3(grid), dim3(block), 0, q, in[s], n, out[s]);
                else if (which == 1) hipLaunchKernelGGL((k31<4, true>), dim3(grid), dim3(block), 0, q, in[s], n, out[s]);
                else if (which == 2) hipLaunchKernelGGL((kread<4>), dim3(grid), dim3(block), 0, q, in[s], 3 * n, (uint32_t*)out[s]);
                else hipLaunchKernelGGL((k31_chunk<4, true>), dim3(grid), dim3(block), 0, q, in[s], n, out[s], (size_t)(which == 3 ? 2560 : which == 4 ? 640 : 20480));
            }
            hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / K;
            t.push_back(ms);
            if (ms < best) best = ms;
        }
        std::sort(t.begin(), t.end());
        med = t[3];
        const double b = which == 2 ? (double)in_bytes : bytes;
        printf("%-34s sets %d streams %d grid %4d x %3d: median %.4f ms %5.0f GB/s   min %.4f ms %5.0f GB/s\n", name, sets, streams, grid, block, med,
               b / med / 1e6, best, b / best / 1e6);
        fflush(stdout);
    };
    for (int sets : {1, 2, 4}) {
        run("3:1 copy U2 nt", sets, 1, 256, 256, 0);
        run("3:1 copy U2 nt", sets, 1, 512, 256, 0);
        run("3:1 copy U4 nt", sets, 1, 256, 256, 1);
        run("3:1 copy U4 nt", sets, 1, 768, 256, 1);
        run("3:1 copy U2 nt", sets, 2, 256, 256, 0);
        run("3:1 copy U2 nt", sets, 2, 512, 256, 0);
        run("3:1 chunks of 40 KB out U4 nt", sets, 1, 768, 256, 3);   // 2560 x 16 B = 32 rows x 1280 B of output: one strip
        run("3:1 chunks of 40 KB out U4 nt", sets, 2, 512, 256, 3);
        run("3:1 chunks of 10 KB out U4 nt", sets, 1, 768, 256, 4);
        run("3:1 chunks of 320 KB out U4 nt", sets, 1, 768, 256, 5);
        run("read only U4 nt", sets, 1, 512, 256, 2);
        run("read only U4 nt", sets, 1, 1024, 256, 2);
    }
    return 0;
}
