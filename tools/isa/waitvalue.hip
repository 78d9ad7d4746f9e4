// dev probe: does hipStreamWaitValue32 work on this part / runtime, on memory a kernel writes?  (frame-level hand-over: the sparse
// kernel's stream must not start before the pixel kernel has STARTED.)   hipcc --offload-arch=gfx950 waitvalue.hip -o waitvalue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(unsigned* flag, unsigned v, long long spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(10);
}
__global__ void consumer(long long* stamp) { if (threadIdx.x == 0) *stamp = wall_clock64(); }
__global__ void stampk(long long* stamp) { if (threadIdx.x == 0) *stamp = wall_clock64(); }
int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* flag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { CK(hipMalloc((void**)&flag, 8)); printf("(plain hipMalloc instead)\n"); }
    CK(hipMemset(flag, 0, 8));
    long long* st = nullptr;
    CK(hipMalloc((void**)&st, 32));
    CK(hipMemset(st, 0, 32));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (unsigned round = 1; round <= 3; round++) {
        // stream b waits for flag == round, then stamps; stream a: stamp, producer (writes the flag after 1 ms, runs 1 ms more)
        e = hipStreamWaitValue32(b, flag, round, hipStreamWaitValueEq, 0xFFFFFFFFu);
        printf("round %u: hipStreamWaitValue32 -> %s\n", round, hipGetErrorString(e));
        if (e != hipSuccess) return 2;
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, st + 1);
        usleep(20000); // the waiting stream has been ready for 20 ms before the producer is even enqueued
        hipLaunchKernelGGL(stampk, dim3(1), dim3(64), 0, a, st + 0);
        hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, flag, round, 100000ll);
        hipLaunchKernelGGL(stampk, dim3(1), dim3(64), 0, a, st + 2);
        CK(hipDeviceSynchronize());
        long long h[3];
        CK(hipMemcpy(h, st, 24, hipMemcpyDeviceToHost));
        printf("   producer enqueued at 0, consumer ran at %+.3f ms, producer finished at %+.3f ms  (expected: consumer ~ +1 ms, before the producer's end at ~ +2 ms)\n",
               (h[1] - h[0]) / 1e5, (h[2] - h[0]) / 1e5);
    }
    return 0;
}
