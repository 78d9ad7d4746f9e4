// membw5.hip -- would ONE large workgroup per CU, loader waves running a chunk ahead of storer waves, lift the 3:1 stream of a batch
// above what k_binary's shape reaches?  (dev tool; VERDICT r3 item 4's proposal, priced with a mock before any rewrite)
// The mock: chunks of CH 16-byte outputs (3 CH inputs) are dealt round-robin to the workgroups; NL loader waves reduce chunk k+1
// into one half of a double buffer in LDS while NS storer waves -- after an optional pause that stands for the morphology's
// latency-bound phases -- write chunk k from the other half; one barrier per chunk.  Next to it: the grid-stride and the chunk copies
// of membw4 at the same grids.  Cold: launches rotate over 4 buffer pairs.
//   hipcc -O3 --offload-arch=gfx950 tools/membw5.hip -o tools/membw5 && tools/membw5
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ldnt(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }

template <int U>
__global__ void k31(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += step * U) {
        u32x4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) { const size_t w0 = (j - lane) * 3; a[u] = ldnt(in + w0 + lane); b[u] = ldnt(in + w0 + 64 + lane); c[u] = ldnt(in + w0 + 128 + lane); }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t j = i + u * step;
            if (j < n) stnt(out + j, a[u] ^ b[u] ^ c[u]);
        }
    }
}
template <int U>
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
                if (j < c1) { const size_t w0 = (j - lane) * 3; a[u] = ldnt(in + w0 + lane); b[u] = ldnt(in + w0 + 64 + lane); c[u] = ldnt(in + w0 + 128 + lane); }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t j = i + u * blockDim.x;
                if (j < c1) stnt(out + j, a[u] ^ b[u] ^ c[u]);
            }
        }
    }
}
// loader waves a chunk ahead of storer waves, double buffer in LDS
template <int NL, int NS, int U>
__global__ __launch_bounds__((NL + NS) * 64) void k31_pipe(const u32x4* __restrict__ in, size_t n, u32x4* __restrict__ out, int CH, int pause)
{
    extern __shared__ u32x4 buf[]; // 2 x CH
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t n_chunks = (n + CH - 1) / CH;
    const int mine = blockIdx.x < n_chunks ? (int)((n_chunks - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    for (int k = 0; k <= mine; k++) {
        if (wave < NL) {
            if (k < mine) {
                const size_t c0 = ((size_t)blockIdx.x + (size_t)k * gridDim.x) * CH;
                const int len = (int)(c0 + CH < n ? CH : n - c0);
                u32x4* dst = buf + (k & 1) * CH;
                for (int i = wave * 64 + lane; i < len; i += NL * 64 * U) {
                    u32x4 a[U], b[U], c[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int j = i + u * NL * 64;
                        if (j < len) { const size_t w0 = (c0 + j - lane) * 3; a[u] = ldnt(in + w0 + lane); b[u] = ldnt(in + w0 + 64 + lane); c[u] = ldnt(in + w0 + 128 + lane); }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int j = i + u * NL * 64;
                        if (j < len) dst[j] = a[u] ^ b[u] ^ c[u];
                    }
                }
            }
        } else if (k > 0) {
            const size_t c0 = ((size_t)blockIdx.x + (size_t)(k - 1) * gridDim.x) * CH;
            const int len = (int)(c0 + CH < n ? CH : n - c0);
            const u32x4* src = buf + ((k - 1) & 1) * CH;
            for (int p = 0; p < pause; p++) __builtin_amdgcn_s_sleep(8); // ~512 clocks each
            for (int i = (wave - NL) * 64 + lane; i < len; i += NS * 64) stnt(out + c0 + i, src[i]);
        }
        __syncthreads();
    }
}
int main(int argc, char** argv)
{
    const size_t out_bytes = 256ull * 1280 * 1024, in_bytes = 3 * out_bytes, n = out_bytes / 16;
    const int MAXS = argc > 1 ? atoi(argv[1]) : 4; // buffer pairs the launches rotate over
    const bool quick = argc > 2;
    u32x4 *in[16], *out[16];
    printf("%d buffer pairs (%.1f GB)\n", MAXS, MAXS * (in_bytes + out_bytes) / 1e9);
    for (int s = 0; s < MAXS; s++) {
        hipMalloc(&in[s], in_bytes + (1 << 20));
        hipMalloc(&out[s], out_bytes + (1 << 20));
        hipMemset(in[s], s + 1, in_bytes);
        hipMemset(out[s], 0, out_bytes);
    }
    hipStream_t st[2];
    hipStreamCreate(&st[0]);
    hipStreamCreate(&st[1]);
    hipFuncSetAttribute((const void*)k31_pipe<8, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k31_pipe<12, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k31_pipe<4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k31_pipe<4, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k31_pipe<6, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k31_pipe<2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipDeviceSynchronize();
    const double bytes = (double)in_bytes + out_bytes;
    auto run = [&](const char* name, int streams, int grid, int block, int which, int CH, int pause) {
        std::vector<double> t;
        for (int rep = 0; rep < 7; rep++) {
            hipDeviceSynchronize();
            const int K = 40;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < K; i++) {
                const int s = i % MAXS;
                hipStream_t q = st[i % streams];
                const size_t lds = 2 * (size_t)CH * 16;
                switch (which) {
                case 0: hipLaunchKernelGGL((k31<2>), dim3(grid), dim3(block), 0, q, in[s], n, out[s]); break;
                case 1: hipLaunchKernelGGL((k31_chunk<4>), dim3(grid), dim3(block), 0, q, in[s], n, out[s], (size_t)CH); break;
                case 2: hipLaunchKernelGGL((k31_pipe<8, 8, 2>), dim3(grid), dim3(1024), lds, q, in[s], n, out[s], CH, pause); break;
                case 3: hipLaunchKernelGGL((k31_pipe<12, 4, 2>), dim3(grid), dim3(1024), lds, q, in[s], n, out[s], CH, pause); break;
                case 4: hipLaunchKernelGGL((k31_pipe<4, 4, 2>), dim3(grid), dim3(512), lds, q, in[s], n, out[s], CH, pause); break;
                case 5: hipLaunchKernelGGL((k31_pipe<4, 4, 4>), dim3(grid), dim3(512), lds, q, in[s], n, out[s], CH, pause); break;
                case 6: hipLaunchKernelGGL((k31_pipe<6, 2, 2>), dim3(grid), dim3(512), lds, q, in[s], n, out[s], CH, pause); break;
                case 7: hipLaunchKernelGGL((k31_pipe<2, 2, 4>), dim3(grid), dim3(256), lds, q, in[s], n, out[s], CH, pause); break;
                }
            }
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return; }
            t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / K);
        }
        std::sort(t.begin(), t.end());
        printf("%-40s streams %d grid %4d x %4d CH %5d pause %2d: median %.4f ms %5.0f GB/s   min %.4f\n", name, streams, grid, block, CH, pause, t[3],
               bytes / t[3] / 1e6, t[0]);
        fflush(stdout);
    };
    // verify the pipe kernel once against the grid-stride copy
    {
        std::vector<uint32_t> a(1 << 16), b(1 << 16);
        hipLaunchKernelGGL((k31<2>), dim3(256), dim3(256), 0, 0, in[0], n, out[0]);
        hipMemcpy(a.data(), (char*)out[0] + out_bytes - a.size() * 4, a.size() * 4, hipMemcpyDeviceToHost);
        hipMemset(out[0], 0, out_bytes);
        hipLaunchKernelGGL((k31_pipe<8, 8, 2>), dim3(256), dim3(1024), 2 * 2560 * 16, 0, in[0], n, out[0], 2560, 0);
        hipMemcpy(b.data(), (char*)out[0] + out_bytes - b.size() * 4, b.size() * 4, hipMemcpyDeviceToHost);
        printf("pipe == copy on the last 256 KB: %s\n", a == b ? "yes" : "NO");
    }
    run("3:1 copy U2 grid-stride", 1, 256, 256, 0, 0, 0);
    run("3:1 copy U2 grid-stride", 1, 256, 1024, 0, 0, 0);
    run("3:1 copy U2 grid-stride", 1, 768, 256, 0, 0, 0);
    run("3:1 chunks U4", 1, 768, 256, 1, 2560, 0);
    run("3:1 chunks U4", 1, 256, 256, 1, 2560, 0);
    run("3:1 chunks U4", 1, 256, 512, 1, 2560, 0);
    run("3:1 chunks U4", 1, 256, 1024, 1, 2560, 0);
    run("3:1 chunks U4", 1, 512, 512, 1, 2560, 0);
    if (quick) {
        run("pipe 8 loaders + 8 storers U2", 1, 256, 1024, 2, 2560, 0);
        run("pipe 8 loaders + 8 storers U2", 1, 256, 1024, 2, 2560, 8);
        run("pipe 8 loaders + 8 storers U2", 2, 256, 1024, 2, 2560, 8);
        return 0;
    }
    for (int pause : {0, 8, 16}) {
        run("pipe 8 loaders + 8 storers U2", 1, 256, 1024, 2, 2560, pause);
        run("pipe 12 loaders + 4 storers U2", 1, 256, 1024, 3, 2560, pause);
        run("pipe 4 + 4 U2, two per CU", 1, 512, 512, 4, 1280, pause);
        run("pipe 4 + 4 U4, two per CU", 1, 512, 512, 5, 1280, pause);
        run("pipe 6 + 2 U2, two per CU", 1, 512, 512, 6, 1280, pause);
        run("pipe 2 + 2 U4, four per CU", 1, 1024, 256, 7, 640, pause);
        run("pipe 2 + 2 U4, four per CU CH 2560", 1, 768, 256, 7, 1280, pause);
    }
    run("pipe 4 + 4 U2, two streams x 1 per CU", 2, 256, 512, 4, 2560, 8);
    run("pipe 2 + 2 U4, two streams x 2 per CU", 2, 512, 256, 7, 1280, 8);
    return 0;
}
