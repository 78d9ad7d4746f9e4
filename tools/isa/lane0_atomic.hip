// lane0_atomic.hip -- the work-queue pattern that hung in round 1, rebuilt small so that its gfx950 ISA can be read
// (hipcc -S, no GPU needed:  make -C tools/isa  ->  lane0_atomic.s) and run (make -C tools/isa run, on the GPU box).
// Every loop is capped at CAP iterations, so no variant can hang the GPU: a variant that "re-runs candidate 0 forever"
// shows up as iterations == CAP.
//
//   A  if (lane == 0) i = atomicAdd(&next, 1);  i = readfirstlane(i);        <- the form that "re-ran candidate 0 forever"
//   B  i = readfirstlane(atomicAdd(lane == 0 ? &next : &sink[lane], lane == 0)) <- the form the kernels use now
//   C  A with the consumer loop body containing a lane-divergent early exit       <- what the round-1 loop body looked like
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CAP 100000

__device__ __noinline__ int body(int i, int lane, const int* __restrict__ in) { return in[i * 64 + lane]; }

extern "C" __global__ __launch_bounds__(256) void queue_A(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int next;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) next = 0;
    __syncthreads();
    int acc = 0, it = 0;
    for (;; it++) {
        if (it >= CAP) break;
        int i = 0;
        if (lane == 0) i = atomicAdd(&next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        acc += body(i, lane, in);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = it;
}

extern "C" __global__ __launch_bounds__(256) void queue_B(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int next;
    __shared__ int sink[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) next = 0;
    __syncthreads();
    int acc = 0, it = 0;
    for (;; it++) {
        if (it >= CAP) break;
        const int i = __builtin_amdgcn_readfirstlane(atomicAdd(lane == 0 ? &next : &sink[lane], lane == 0 ? 1 : 0));
        if (i >= n) break;
        acc += body(i, lane, in);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = it;
}

// C: the body leaves lanes behind.  A lane that takes the divergent `continue` is parked by the structurizer until the loop
// latch; if the compiler proves `i` uniform it may keep the loop's exit test scalar -- but the readfirstlane of the NEXT
// iteration then executes under whatever EXEC the latch restored.
extern "C" __global__ __launch_bounds__(256) void queue_C(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int next;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) next = 0;
    __syncthreads();
    int acc = 0, it = 0;
    for (;; it++) {
        if (it >= CAP) break;
        int i = 0;
        if (lane == 0) i = atomicAdd(&next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        const int v = in[i * 64 + lane];
        if (v < 0) continue; // lane-divergent
        acc += body(i, lane, in) + v;
        if (acc == 12345) break; // lane-divergent exit: from here on the loop runs with a partial EXEC mask
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = it;
}

// D: C, but the lane that leaves early is lane 0 (the producer): nobody draws any more, the first ACTIVE lane's i = 0 is
// what v_readfirstlane returns -> candidate 0 for ever (here: until CAP).  This is the failure mode.
extern "C" __global__ __launch_bounds__(256) void queue_D(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int next;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) next = 0;
    __syncthreads();
    int acc = 0, it = 0;
    for (;; it++) {
        if (it >= CAP) break;
        int i = 0;
        if (lane == 0) i = atomicAdd(&next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        acc += in[i * 64 + lane];
        if (lane == 0 && i == 3) break; // the producer lane leaves the loop; the others go on
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = it;
}

// E: the robust form: the producer is the first ACTIVE lane, whichever that is
extern "C" __global__ __launch_bounds__(256) void queue_E(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int next;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) next = 0;
    __syncthreads();
    int acc = 0, it = 0;
    for (;; it++) {
        if (it >= CAP) break;
        int i = 0;
        if (lane == __builtin_amdgcn_readfirstlane(lane)) i = atomicAdd(&next, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= n) break;
        acc += in[i * 64 + lane];
        if (lane == 0 && i == 3) break;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = it;
}

int main()
{
    const int n = 64, blocks = 4;
    int *in, *out;
    hipMalloc(&in, n * 64 * sizeof(int));
    hipMalloc(&out, 2 * blocks * 256 * sizeof(int));
    int* h = new int[n * 64];
    for (int i = 0; i < n * 64; i++) h[i] = (i * 2654435761u >> 20) & 1023; // >= 0: C's divergent exits are not taken by data
    hipMemcpy(in, h, n * 64 * sizeof(int), hipMemcpyHostToDevice);
    int* o = new int[2 * blocks * 256];
    const char* names[5] = {"A if(lane==0)+readfirstlane", "B all lanes, sink words", "C A + divergent body", "D producer lane leaves",
                            "E first-active-lane producer"};
    void (*ks[5])(const int*, int, int*) = {queue_A, queue_B, queue_C, queue_D, queue_E};
    for (int k = 0; k < 5; k++) {
        hipMemset(out, 0, 2 * blocks * 256 * sizeof(int));
        hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(256), 0, 0, in, n, out);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o, out, 2 * blocks * 256 * sizeof(int), hipMemcpyDeviceToHost);
        // every candidate 0..n-1 must have been consumed exactly once per workgroup: sum over a block's 4 waves of acc(lane) ==
        // sum_i in[i*64+lane]
        int bad = 0, itmax = 0;
        for (int b = 0; b < blocks; b++)
            for (int lane = 0; lane < 64; lane++) {
                long want = 0, got = 0;
                for (int i = 0; i < n; i++) want += h[i * 64 + lane];
                for (int w = 0; w < 4; w++) got += o[b * 256 + w * 64 + lane];
                bad += want != got;
            }
        for (int t = 0; t < blocks * 256; t++) itmax = o[blocks * 256 + t] > itmax ? o[blocks * 256 + t] : itmax;
        printf("%-32s %s  lanes with a wrong sum: %d  max iterations of a lane: %d%s\n", names[k], hipGetErrorString(e), bad, itmax,
               itmax >= CAP ? "  <- ran until the cap: candidate 0 for ever" : "");
    }
    return 0;
}
