// Does CDNA4 skip the 32-lane pass of a wave64 VALU instruction whose EXEC half is all zero?
// Times a dependent f64 FMA chain with different active-lane masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain(double *out, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-9, b = 1.0000001, c = 1e-7;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a = a * b + c; a2 = a2 * b + c; a3 = a3 * b + c; a4 = a4 * b + c;
            a = a * b + c; a2 = a2 * b + c; a3 = a3 * b + c; a4 = a4 * b + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
}
__global__ void chainf(float *out, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-6f, b = 1.0000001f, c = 1e-7f;
    float a2 = a + 1, a3 = a + 2, a4 = a + 3;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a = a * b + c; a2 = a2 * b + c; a3 = a3 * b + c; a4 = a4 * b + c;
            a = a * b + c; a2 = a2 * b + c; a3 = a3 * b + c; a4 = a4 * b + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4;
}
int main() {
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    double *d; float *f;
    hipMalloc(&d, sizeof(double) * blocks * threads); hipMalloc(&f, sizeof(float) * blocks * threads);
    struct { const char *name; unsigned long long m; } masks[] = {
        {"all 64", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"high 32", 0xFFFFFFFF00000000ull}, {"even lanes (32)", 0x5555555555555555ull},
        {"low 16", 0xFFFFull}, {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull}, {"1 lane", 1ull}, {"lane 0 + lane 32", 0x100000001ull}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass)
    for (auto &mk : masks) {
        float ms64 = 0, ms32 = 0;
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, mk.m, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, mk.m, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms64, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(chainf, dim3(blocks), dim3(threads), 0, 0, f, mk.m, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms32, e0, e1);
        if (pass) printf("%-22s f64 %.3f ms   f32 %.3f ms\n", mk.name, ms64, ms32);
    }
    return 0;
}
