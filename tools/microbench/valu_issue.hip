// valu_issue.hip -- issue cost (shader cycles per wave64 instruction per SIMD) of the instruction classes the
// render kernels are made of.  The VALU-issue roofline of bench.py / tools/summarize_profile.py prices the
// rocprofv3 SQ_INSTS_VALU_* class counts with these numbers (profiles/r02_valu_issue.json).
//
// Method: ONE workgroup of 256*W threads on one CU = W waves on each of its 4 SIMDs.  Every wave runs `iters`
// loop trips of 64 copies of one instruction over 8 independent register sets (8 instructions per asm block, so the
// compiler puts nothing between them; the loop overhead is 1 s_add + 1 s_cmp + 1 branch per 64 instructions),
// brackets the loop with s_memtime and stores the elapsed ticks.
// cycles per instruction per SIMD = max ticks over the waves / (64 * iters * W).
// W = 1 shows what ONE wave can issue (its own instruction cadence), W = 4 / 8 the pipe's throughput.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue > out.json
#include <hip/hip_runtime.h>

#include <cstdio>

#define BLOCK8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define KERNEL(NAME, ACC_T, SRC_T, INSTR, ...)                                                                   \
    __global__ void k_##NAME(unsigned long long *ticks, int iters, double seed) {                                \
        ACC_T a0 = (ACC_T)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
              a7 = a0 + 7;                                                                                       \
        SRC_T b = (SRC_T)(1.25 + (double)threadIdx.x * 1e-3), c = (SRC_T)3;                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
        for (int i = 0; i < iters; ++i) {                                                                        \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                      \
                asm volatile(BLOCK8(INSTR)                                                                       \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                             : "v"(b), "v"(c)                                                                    \
                             : __VA_ARGS__);                                                                     \
            }                                                                                                    \
        }                                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
        if ((threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;                                          \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == (ACC_T)123) ticks[63] = 1; /* keep the results alive */     \
    }

#define I_fma_f64(A) "v_fma_f64 %" #A ", %" #A ", %8, %9\n"
KERNEL(fma_f64, double, double, I_fma_f64, "memory")
#define I_add_f64(A) "v_add_f64 %" #A ", %" #A ", %9\n"
KERNEL(add_f64, double, double, I_add_f64, "memory")
#define I_mul_f64(A) "v_mul_f64 %" #A ", %" #A ", %8\n"
KERNEL(mul_f64, double, double, I_mul_f64, "memory")
#define I_max_f64(A) "v_max_f64 %" #A ", %" #A ", %8\n"
KERNEL(max_f64, double, double, I_max_f64, "memory")
#define I_rcp_f64(A) "v_rcp_f64 %" #A ", %" #A "\n"
KERNEL(rcp_f64, double, double, I_rcp_f64, "memory")
#define I_rsq_f64(A) "v_rsq_f64 %" #A ", %" #A "\n"
KERNEL(rsq_f64, double, double, I_rsq_f64, "memory")
#define I_sqrt_f64(A) "v_sqrt_f64 %" #A ", %" #A "\n"
KERNEL(sqrt_f64, double, double, I_sqrt_f64, "memory")
#define I_div_scale_f64(A) "v_div_scale_f64 %" #A ", s[20:21], %" #A ", %8, %" #A "\n"
KERNEL(div_scale_f64, double, double, I_div_scale_f64, "s20", "s21")
#define I_div_fmas_f64(A) "v_div_fmas_f64 %" #A ", %" #A ", %8, %9\n"
KERNEL(div_fmas_f64, double, double, I_div_fmas_f64, "memory")
#define I_div_fixup_f64(A) "v_div_fixup_f64 %" #A ", %" #A ", %8, %9\n"
KERNEL(div_fixup_f64, double, double, I_div_fixup_f64, "memory")
#define I_cmp_f64_sgpr(A) "v_cmp_lt_f64 s[20:21], %" #A ", %8\n"
KERNEL(cmp_f64_sgpr, double, double, I_cmp_f64_sgpr, "s20", "s21")
#define I_cmp_f64_vcc(A) "v_cmp_lt_f64 vcc, %" #A ", %8\n"
KERNEL(cmp_f64_vcc, double, double, I_cmp_f64_vcc, "vcc")
#define I_ldexp_f64(A) "v_ldexp_f64 %" #A ", %" #A ", 1\n"
KERNEL(ldexp_f64, double, double, I_ldexp_f64, "memory")
#define I_fract_f64(A) "v_fract_f64 %" #A ", %" #A "\n"
KERNEL(fract_f64, double, double, I_fract_f64, "memory")
#define I_trig_preop_f64(A) "v_trig_preop_f64 %" #A ", %" #A ", 1\n"
KERNEL(trig_preop_f64, double, double, I_trig_preop_f64, "memory")
#define I_cvt_f32_f64(A) "v_cvt_f32_f64 %" #A ", %8\n"
KERNEL(cvt_f32_f64, float, double, I_cvt_f32_f64, "memory")
#define I_cvt_u32_f64(A) "v_cvt_u32_f64 %" #A ", %8\n"
KERNEL(cvt_u32_f64, unsigned, double, I_cvt_u32_f64, "memory")
#define I_cvt_f64_u32(A) "v_cvt_f64_u32 %" #A ", %8\n"
KERNEL(cvt_f64_u32, double, unsigned, I_cvt_f64_u32, "memory")
#define I_cvt_f64_f32(A) "v_cvt_f64_f32 %" #A ", %8\n"
KERNEL(cvt_f64_f32, double, float, I_cvt_f64_f32, "memory")
#define I_fma_f32(A) "v_fma_f32 %" #A ", %" #A ", %8, %9\n"
KERNEL(fma_f32, float, float, I_fma_f32, "memory")
#define I_add_f32(A) "v_add_f32 %" #A ", %" #A ", %9\n"
KERNEL(add_f32, float, float, I_add_f32, "memory")
#define I_mul_f32(A) "v_mul_f32 %" #A ", %" #A ", %8\n"
KERNEL(mul_f32, float, float, I_mul_f32, "memory")
#define I_max_f32(A) "v_max_f32 %" #A ", %" #A ", %8\n"
KERNEL(max_f32, float, float, I_max_f32, "memory")
#define I_min_f32(A) "v_min_f32 %" #A ", %" #A ", %8\n"
KERNEL(min_f32, float, float, I_min_f32, "memory")
#define I_max3_f32(A) "v_max3_f32 %" #A ", %" #A ", %8, %9\n"
KERNEL(max3_f32, float, float, I_max3_f32, "memory")
#define I_rcp_f32(A) "v_rcp_f32 %" #A ", %" #A "\n"
KERNEL(rcp_f32, float, float, I_rcp_f32, "memory")
#define I_cmp_f32_sgpr(A) "v_cmp_lt_f32 s[20:21], %" #A ", %8\n"
KERNEL(cmp_f32_sgpr, float, float, I_cmp_f32_sgpr, "s20", "s21")
#define I_add_u32(A) "v_add_u32 %" #A ", %" #A ", %8\n"
KERNEL(add_u32, unsigned, unsigned, I_add_u32, "memory")
#define I_sub_u32(A) "v_sub_u32 %" #A ", %" #A ", %8\n"
KERNEL(sub_u32, unsigned, unsigned, I_sub_u32, "memory")
#define I_add_co_u32(A) "v_add_co_u32 %" #A ", s[20:21], %" #A ", %8\n"
KERNEL(add_co_u32, unsigned, unsigned, I_add_co_u32, "s20", "s21")
#define I_addc_co_u32(A) "v_addc_co_u32 %" #A ", s[20:21], %" #A ", %8, s[22:23]\n"
KERNEL(addc_co_u32, unsigned, unsigned, I_addc_co_u32, "s20", "s21")
#define I_xor_b32(A) "v_xor_b32 %" #A ", %" #A ", %8\n"
KERNEL(xor_b32, unsigned, unsigned, I_xor_b32, "memory")
#define I_and_b32(A) "v_and_b32 %" #A ", %" #A ", %8\n"
KERNEL(and_b32, unsigned, unsigned, I_and_b32, "memory")
#define I_or_b32(A) "v_or_b32 %" #A ", %" #A ", %8\n"
KERNEL(or_b32, unsigned, unsigned, I_or_b32, "memory")
#define I_lshlrev_b32(A) "v_lshlrev_b32 %" #A ", 3, %" #A "\n"
KERNEL(lshlrev_b32, unsigned, unsigned, I_lshlrev_b32, "memory")
#define I_lshrrev_b32(A) "v_lshrrev_b32 %" #A ", 3, %" #A "\n"
KERNEL(lshrrev_b32, unsigned, unsigned, I_lshrrev_b32, "memory")
#define I_alignbit_b32(A) "v_alignbit_b32 %" #A ", %" #A ", %8, %9\n"
KERNEL(alignbit_b32, unsigned, unsigned, I_alignbit_b32, "memory")
#define I_cndmask_b32_sgpr(A) "v_cndmask_b32 %" #A ", %" #A ", %8, s[22:23]\n"
KERNEL(cndmask_b32_sgpr, unsigned, unsigned, I_cndmask_b32_sgpr, "memory")
#define I_cndmask_b32_vcc(A) "v_cndmask_b32 %" #A ", %" #A ", %8, vcc\n"
KERNEL(cndmask_b32_vcc, unsigned, unsigned, I_cndmask_b32_vcc, "memory")
#define I_mov_b32(A) "v_mov_b32 %" #A ", %8\n"
KERNEL(mov_b32, unsigned, unsigned, I_mov_b32, "memory")
#define I_mul_lo_u32(A) "v_mul_lo_u32 %" #A ", %" #A ", %8\n"
KERNEL(mul_lo_u32, unsigned, unsigned, I_mul_lo_u32, "memory")
#define I_mul_hi_u32(A) "v_mul_hi_u32 %" #A ", %" #A ", %8\n"
KERNEL(mul_hi_u32, unsigned, unsigned, I_mul_hi_u32, "memory")
#define I_mad_u32_u24(A) "v_mad_u32_u24 %" #A ", %" #A ", %8, %9\n"
KERNEL(mad_u32_u24, unsigned, unsigned, I_mad_u32_u24, "memory")
#define I_cmp_u32_sgpr(A) "v_cmp_lt_u32 s[20:21], %" #A ", %8\n"
KERNEL(cmp_u32_sgpr, unsigned, unsigned, I_cmp_u32_sgpr, "s20", "s21")
#define I_bfe_u32(A) "v_bfe_u32 %" #A ", %" #A ", 3, 5\n"
KERNEL(bfe_u32, unsigned, unsigned, I_bfe_u32, "memory")
#define I_and_or_b32(A) "v_and_or_b32 %" #A ", %" #A ", %8, %9\n"
KERNEL(and_or_b32, unsigned, unsigned, I_and_or_b32, "memory")
#define I_lshl_or_b32(A) "v_lshl_or_b32 %" #A ", %" #A ", 3, %8\n"
KERNEL(lshl_or_b32, unsigned, unsigned, I_lshl_or_b32, "memory")
#define I_lshlrev_b64(A) "v_lshlrev_b64 %" #A ", %8, %" #A "\n"
KERNEL(lshlrev_b64, unsigned long long, unsigned, I_lshlrev_b64, "memory")
#define I_lshrrev_b64(A) "v_lshrrev_b64 %" #A ", %8, %" #A "\n"
KERNEL(lshrrev_b64, unsigned long long, unsigned, I_lshrrev_b64, "memory")
#define I_mad_u64_u32(A) "v_mad_u64_u32 %" #A ", s[20:21], %8, %9, %" #A "\n"
KERNEL(mad_u64_u32, unsigned long long, unsigned, I_mad_u64_u32, "s20", "s21")
#define I_cmp_u64_sgpr(A) "v_cmp_lt_u64 s[20:21], %" #A ", %" #A "\n"
KERNEL(cmp_u64_sgpr, unsigned long long, unsigned, I_cmp_u64_sgpr, "s20", "s21")
#define I_s_nop0(A) "s_nop 0\n"
KERNEL(s_nop0, unsigned, unsigned, I_s_nop0, "memory")

typedef void (*Kern)(unsigned long long *, int, double);
struct Entry {
    const char *name;
    Kern k;
};
#define E(NAME) {#NAME, k_##NAME}

int main() {
    const Entry entries[] = {
        E(fma_f64),
        E(add_f64),
        E(mul_f64),
        E(max_f64),
        E(rcp_f64),
        E(rsq_f64),
        E(sqrt_f64),
        E(div_scale_f64),
        E(div_fmas_f64),
        E(div_fixup_f64),
        E(cmp_f64_sgpr),
        E(cmp_f64_vcc),
        E(ldexp_f64),
        E(fract_f64),
        E(trig_preop_f64),
        E(cvt_f32_f64),
        E(cvt_u32_f64),
        E(cvt_f64_u32),
        E(cvt_f64_f32),
        E(fma_f32),
        E(add_f32),
        E(mul_f32),
        E(max_f32),
        E(min_f32),
        E(max3_f32),
        E(rcp_f32),
        E(cmp_f32_sgpr),
        E(add_u32),
        E(sub_u32),
        E(add_co_u32),
        E(addc_co_u32),
        E(xor_b32),
        E(and_b32),
        E(or_b32),
        E(lshlrev_b32),
        E(lshrrev_b32),
        E(alignbit_b32),
        E(cndmask_b32_sgpr),
        E(cndmask_b32_vcc),
        E(mov_b32),
        E(mul_lo_u32),
        E(mul_hi_u32),
        E(mad_u32_u24),
        E(cmp_u32_sgpr),
        E(bfe_u32),
        E(and_or_b32),
        E(lshl_or_b32),
        E(lshlrev_b64),
        E(lshrrev_b64),
        E(mad_u64_u32),
        E(cmp_u64_sgpr),
        E(s_nop0)};
    const int iters = 4000;
    unsigned long long *d = nullptr;
    if (hipMalloc((void **)&d, 64 * sizeof(unsigned long long)) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    printf("{\n \"method\": \"one workgroup of 256*W threads (W waves per SIMD of one CU), 64 copies of the instruction per loop trip over 8 "
           "independent register sets (8 per asm block), %d trips, s_memtime around the loop; cycles = max ticks over the waves / (64 * trips * W)\",\n"
           " \"unit\": \"shader cycles per wave64 instruction per SIMD\",\n \"instructions\": {\n",
           iters);
    bool first = true;
    for (const Entry &e : entries) {
        double cyc[4] = {0, 0, 0, 0};
        const int ws[4] = {1, 2, 3, 4};
        for (int wi = 0; wi < 4; ++wi) {
            const int W = ws[wi];
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(d, 0, 64 * sizeof(unsigned long long));
                hipLaunchKernelGGL(e.k, dim3(1), dim3(256 * W), 0, 0, d, iters, 1.5);
                if (hipDeviceSynchronize() != hipSuccess) {
                    fprintf(stderr, "kernel %s failed\n", e.name);
                    return 1;
                }
                unsigned long long h[64];
                (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
                unsigned long long mx = 0;
                for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
                if (rep > 0 && mx < best) best = mx;
            }
            cyc[wi] = (double)best / (64.0 * iters * W);
        }
        printf("%s  \"%s\": {\"w1\": %.3f, \"w2\": %.3f, \"w3\": %.3f, \"w4\": %.3f}", first ? "" : ",\n", e.name, cyc[0], cyc[1], cyc[2], cyc[3]);
        first = false;
    }
    printf("\n }\n}\n");
    (void)hipFree(d);
    return 0;
}
