// dev probe: how long after a cross-stream dependency is satisfied does the dependent kernel start?  (1) event recorded on stream a
// BEFORE a long kernel, stream b waits on it and launches a stamp kernel: b should start at once; (2) event AFTER a short kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long* stamp, long long ticks)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = wall_clock64();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void stampk(long long* stamp) { if (threadIdx.x == 0) *stamp = wall_clock64(); }
int main()
{
    long long* st = nullptr;
    CK(hipMalloc((void**)&st, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    for (int round = 0; round < 4; round++) {
        CK(hipMemset(st, 0, 64));
        CK(hipDeviceSynchronize());
        // (1) b depends on a point of a that lies BEFORE the long kernel
        CK(hipEventRecord(ev, a));
        CK(hipStreamWaitEvent(b, ev, 0));
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, a, st, 30000ll);         // 0.3 ms
        hipLaunchKernelGGL(stampk, dim3(1), dim3(64), 0, b, st + 2);
        // (2) then b depends on the END of that kernel
        CK(hipEventRecord(ev2, a));
        CK(hipStreamWaitEvent(b, ev2, 0));
        hipLaunchKernelGGL(stampk, dim3(1), dim3(64), 0, b, st + 3);
        // (3) same-stream successor for comparison
        hipLaunchKernelGGL(stampk, dim3(1), dim3(64), 0, a, st + 4);
        CK(hipDeviceSynchronize());
        long long h[5];
        CK(hipMemcpy(h, st, 40, hipMemcpyDeviceToHost));
        printf("round %d: long kernel 0 .. %.1f us | dependent-on-earlier-point kernel on the other stream ran at %+.1f us | dependent-on-its-end kernel on the other stream at end %+.1f us | same-stream successor at end %+.1f us\n",
               round, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[3] - h[1]) / 100.0, (h[4] - h[1]) / 100.0);
    }
    return 0;
}
