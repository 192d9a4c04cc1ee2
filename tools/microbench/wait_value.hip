// GPU box: does hipStreamWaitValue32 on a flag written by a RUNNING kernel release a second stream before that kernel ends?
// (the mechanism behind starting the next render_kernel at the previous one's end-of-launch tail)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void producer(unsigned *flag, long long *stamps, long long spin_before, long long spin_after) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_before) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        stamps[0] = wall_clock64();
    }
    while (wall_clock64() - t0 < spin_before + spin_after) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void consumer(long long *stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall_clock64();
}
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned *flag = nullptr;
    CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    long long *stamps = nullptr;
    CK(hipMalloc((void **)&stamps, 64));
    CK(hipMemset(stamps, 0, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamWriteValue32(a, flag, 0, 0));
        CK(hipStreamSynchronize(a));
        hipLaunchKernelGGL(producer, dim3(64), dim3(64), 0, a, flag, stamps, 100000LL, 400000LL); // wall_clock64: 100 MHz -> 1 ms + 4 ms
        CK(hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueEq, 0xFFFFFFFFu));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, stamps);
        CK(hipDeviceSynchronize());
        long long h[3];
        CK(hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost));
        printf("rep %d: flag set at 0, producer ended at %+.3f ms, consumer ran at %+.3f ms  -> %s\n", rep, (h[1] - h[0]) / 1e5, (h[2] - h[0]) / 1e5,
               h[2] < h[1] ? "released DURING the producer" : "released only after it");
    }
    return 0;
}
