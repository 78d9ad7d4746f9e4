// membw.hip -- reference streaming kernels with k_binary's traffic mix (dev tool, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const uint4* __restrict__ in, size_t n, uint4* out)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678) out[0] = acc;
}
__global__ void k_copy(const uint4* __restrict__ in, size_t n, uint4* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// 3:1 : each thread reads 48 contiguous bytes (3 x uint4, lane stride 48) and writes 16 (lane stride 16)
__global__ void k_3to1_strided(const uint4* __restrict__ in, size_t n_out, uint4* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
        uint4 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
        uint4 r;
        r.x = a.x ^ b.y ^ c.z; r.y = a.y ^ b.z ^ c.w; r.z = a.z ^ b.w ^ c.x; r.w = a.w ^ b.x ^ c.y;
        out[i] = r;
    }
}
// 3:1 coalesced: wave reads 3 x 1 KiB contiguous
__global__ void k_3to1_coal(const uint4* __restrict__ in, size_t n_out, uint4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out; i += (size_t)gridDim.x * blockDim.x) {
        size_t w0 = (i - lane) * 3;
        uint4 a = in[w0 + lane], b = in[w0 + 64 + lane], c = in[w0 + 128 + lane];
        uint4 r;
        r.x = a.x ^ b.y ^ c.z; r.y = a.y ^ b.z ^ c.w; r.z = a.z ^ b.w ^ c.x; r.w = a.w ^ b.x ^ c.y;
        out[i] = r;
    }
}
__global__ void k_write(size_t n, uint4* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(i, 1, 2, 3);
}
int main()
{
    const size_t in_bytes = 256ull * 1280 * 1024 * 3, out_bytes = 256ull * 1280 * 1024;
    uint4 *in, *out;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, in_bytes));
    CK(hipMemset(in, 1, in_bytes)); CK(hipMemset(out, 0, in_bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192}) for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, in, in_bytes / 16, out);
            if (mode == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, in, out_bytes * 2 / 16, out);
            if (mode == 2) hipLaunchKernelGGL(k_3to1_strided, dim3(grid), dim3(256), 0, 0, in, out_bytes / 16, out);
            if (mode == 3) hipLaunchKernelGGL(k_3to1_coal, dim3(grid), dim3(256), 0, 0, in, out_bytes / 16, out);
            if (mode == 4) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, out_bytes / 16, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const char* nm[5] = {"read 1.0GB", "copy 0.67GB+0.67GB", "3:1 strided", "3:1 coalesced", "write 0.34GB"};
        double bytes = mode == 0 ? in_bytes : mode == 1 ? out_bytes * 4.0 : mode == 4 ? out_bytes : (double)in_bytes + out_bytes;
        printf("grid %5d %-20s %.4f ms  %.0f GB/s\n", grid, nm[mode], best, bytes / best / 1e6);
    }
    return 0;
}
