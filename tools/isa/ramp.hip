// dev probe: how fast does the hardware place the workgroups of a launch?  Every workgroup stamps its start and then stays (300 us),
// so the whole grid must be resident at once; printed: the spread of the start stamps, by grid, block size, LDS and registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int VG>
__global__ void ramp(long long* stamp, long long ticks)
{
    extern __shared__ int lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamp[blockIdx.x] = t0;
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = threadIdx.x * 0.5f + i;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(4);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < VG; i++) s += acc[i];
    if (s == 12345.678f) lds[threadIdx.x] = 1, stamp[0] = lds[0];
}
template <int VG>
int run(const char* name, int grid, int block, int ldsb, long long* d)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ramp<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    std::vector<long long> h(grid);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(ramp<VG>, dim3(grid), dim3(block), ldsb, 0, d, 30000ll);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, grid * sizeof(long long), hipMemcpyDeviceToHost));
    }
    std::sort(h.begin(), h.end());
    printf("%-22s grid %5d x %4d  lds %6d B: first .. median .. last workgroup start  0 .. %.1f .. %.1f us\n", name, grid, block, ldsb,
           (h[grid / 2] - h[0]) / 100.0, (h[grid - 1] - h[0]) / 100.0);
    return 0;
}
int main()
{
    long long* d = nullptr;
    CK(hipMalloc((void**)&d, 1 << 20));
    run<8>("few regs", 256, 256, 0, d);
    run<8>("few regs", 768, 256, 0, d);
    run<8>("few regs", 768, 256, 14 * 1024, d);
    run<8>("few regs", 2048, 256, 0, d);
    run<64>("~80 regs", 768, 256, 14 * 1024, d);
    run<64>("~80 regs", 256, 768, 42 * 1024, d);
    run<64>("~80 regs", 256, 1024, 56 * 1024, d);
    run<64>("~80 regs", 512, 512, 28 * 1024, d);
    run<150>("~168 regs", 256, 256, 88 * 1024, d);
    run<150>("~168 regs", 256, 512, 88 * 1024, d);
    return 0;
}
