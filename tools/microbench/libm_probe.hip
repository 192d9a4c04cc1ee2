// libm_probe.hip -- the DEVICE's log / sin / atan2 / acos on given arguments (the four libm functions render_kernel calls:
// rt_lane.h log_cold, checker_sine_cold, sphere_uv_cold), compiled like the kernels (-O3 -ffp-contract=off -fno-fast-math).
// tests/sweeps/libm_attribution.py feeds it the arguments a sample's path passed to those functions on the host and compares the
// results bit by bit: the device's libm is accurate to about an ulp, not correctly rounded.
//   hipcc -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 libm_probe.hip -o libm_probe
//   libm_probe in.bin out.bin      in: n x {fn, a, b} doubles (fn 0 log, 1 sin, 2 atan2(a, b), 3 acos), out: n doubles
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const double *in, double *out, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int fn = (int)in[3 * i];
    const double a = in[3 * i + 1], b = in[3 * i + 2];
    double r;
    if (fn == 0)
        r = log(a);
    else if (fn == 1)
        r = sin(a);
    else if (fn == 2)
        r = atan2(a, b);
    else
        r = acos(a);
    out[i] = r;
}

#define CHECK(e)                                                                  \
    do {                                                                          \
        hipError_t err_ = (e);                                                    \
        if (err_ != hipSuccess) {                                                 \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_));        \
            return 2;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const int n = (int)(bytes / (3 * (long)sizeof(double)));
    std::vector<double> in((size_t)n * 3), out((size_t)n);
    if (n > 0 && std::fread(in.data(), sizeof(double), (size_t)n * 3, f) != (size_t)n * 3) return 1;
    std::fclose(f);
    if (n > 0) {
        double *din = nullptr, *dout = nullptr;
        CHECK(hipMalloc(&din, (size_t)n * 3 * sizeof(double)));
        CHECK(hipMalloc(&dout, (size_t)n * sizeof(double)));
        CHECK(hipMemcpy(din, in.data(), (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, din, dout, n);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out.data(), dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
    f = std::fopen(argv[2], "wb");
    if (!f) return 1;
    std::fwrite(out.data(), sizeof(double), (size_t)n, f);
    std::fclose(f);
    return 0;
}
