// Micro-test (gfx950): is one wait state enough between a packed-fp32 VALU producer and a dependent consumer?
// LLVM puts ONE `s_nop 0` between v_pk_fma_f32 and a v_cvt_pk_bf16_f32 that reads its result when nothing else separates them.
// Each case: sentinel -> v[200:201], producer writes v[200:201], GAP, consumer reads v200 / v201; results checked per lane quarter.
//   hipcc --offload-arch=gfx950 -O2 scripts/hazard/pkfma_hazard.hip -o build_variants/pkfma_hazard && ./pkfma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CASE_KERNEL(NAME, PRODUCER, GAP, CONSUMER)                                                                       \
    __global__ void NAME(const float* in, uint32_t* out, int iters) {                                                    \
        const int i = blockIdx.x * blockDim.x + threadIdx.x;                                                             \
        f2 x = {in[i * 6 + 0], in[i * 6 + 1]}, y = {in[i * 6 + 2], in[i * 6 + 3]}, z = {in[i * 6 + 4], in[i * 6 + 5]};  \
        uint32_t bad = 0, r = 0, last = 0;                                                                               \
        const float sentinel = 1000.0f;                                                                                  \
        for (int k = 0; k < iters; ++k) {                                                                                \
            asm volatile("v_mov_b32 v200, %4\n v_mov_b32 v201, %4\n s_nop 7\n s_nop 7\n" PRODUCER "\n" GAP CONSUMER "\n s_nop 7\n s_nop 7\n" \
                         : "=v"(r) : "v"(x), "v"(y), "v"(z), "v"(sentinel) : "v200", "v201");                           \
            if (k == 0) last = r;                                                                                        \
            bad += (r != last);                                                                                          \
        }                                                                                                                \
        out[i * 2] = last;                                                                                               \
        out[i * 2 + 1] = bad;                                                                                            \
    }

#define P_FMA "v_pk_fma_f32 v[200:201], %1, %2, %3"
#define P_MUL "v_pk_mul_f32 v[200:201], %1, %2"
#define P_ADD "v_pk_add_f32 v[200:201], %1, %3"
#define P_FMA32 "v_fma_f32 v200, %1, %2, %3\n v_fma_f32 v201, %1, %2, %3"   /* not comparable numerically; run-to-run only */
#define C_CVT "v_cvt_pk_bf16_f32 %0, v200, v201"
#define C_ADD "v_add_f32 %0, v200, v201"
#define G0 ""
#define G1 "s_nop 0\n"
#define G2 "s_nop 1\n"
#define G3 "s_nop 2\n"
#define GV "v_mov_b32 v202, v203\n"

CASE_KERNEL(fma_g0_cvt, P_FMA, G0, C_CVT)
CASE_KERNEL(fma_g1_cvt, P_FMA, G1, C_CVT)
CASE_KERNEL(fma_g2_cvt, P_FMA, G2, C_CVT)
CASE_KERNEL(fma_g3_cvt, P_FMA, G3, C_CVT)
CASE_KERNEL(fma_gv_cvt, P_FMA, GV, C_CVT)
CASE_KERNEL(mul_g0_cvt, P_MUL, G0, C_CVT)
CASE_KERNEL(mul_g1_cvt, P_MUL, G1, C_CVT)
CASE_KERNEL(add_g0_cvt, P_ADD, G0, C_CVT)
CASE_KERNEL(add_g1_cvt, P_ADD, G1, C_CVT)
CASE_KERNEL(fma_g0_add, P_FMA, G0, C_ADD)
CASE_KERNEL(fma_g1_add, P_FMA, G1, C_ADD)
CASE_KERNEL(fma_g2_add, P_FMA, G2, C_ADD)

static uint32_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

typedef void (*kern_t)(const float*, uint32_t*, int);
struct Case { const char* name; kern_t k; int prod; int cons; };

int main() {
    const int waves_per_block = 8, blocks = 1024, n = blocks * waves_per_block * 64, iters = 64;
    std::vector<float> h(n * 6);
    uint32_t s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 4096.0f - 8.0f; }
    float* d_in; uint32_t* d_out;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, n * 8);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    Case cases[] = {{"pk_fma            -> cvt_pk_bf16", fma_g0_cvt, 0, 0}, {"pk_fma  s_nop 0   -> cvt_pk_bf16", fma_g1_cvt, 0, 0},
                    {"pk_fma  s_nop 1   -> cvt_pk_bf16", fma_g2_cvt, 0, 0}, {"pk_fma  s_nop 2   -> cvt_pk_bf16", fma_g3_cvt, 0, 0},
                    {"pk_fma  v_mov     -> cvt_pk_bf16", fma_gv_cvt, 0, 0}, {"pk_mul            -> cvt_pk_bf16", mul_g0_cvt, 1, 0},
                    {"pk_mul  s_nop 0   -> cvt_pk_bf16", mul_g1_cvt, 1, 0}, {"pk_add            -> cvt_pk_bf16", add_g0_cvt, 2, 0},
                    {"pk_add  s_nop 0   -> cvt_pk_bf16", add_g1_cvt, 2, 0}, {"pk_fma            -> v_add_f32", fma_g0_add, 0, 1},
                    {"pk_fma  s_nop 0   -> v_add_f32", fma_g1_add, 0, 1}, {"pk_fma  s_nop 1   -> v_add_f32", fma_g2_add, 0, 1}};
    std::vector<uint32_t> o(n * 2);
    for (int wpb : {1, 4, 8}) {   // 1 wave per block (few waves per SIMD) .. 8 waves per block (several waves per SIMD)
        for (auto& c : cases) {
            hipMemset(d_out, 0, n * 8);
            hipLaunchKernelGGL(c.k, dim3(n / (wpb * 64)), dim3(wpb * 64), 0, 0, d_in, d_out, iters);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(o.data(), d_out, n * 8, hipMemcpyDeviceToHost);
            long wrong[4] = {0, 0, 0, 0}, flaky[4] = {0, 0, 0, 0}, stale[4] = {0, 0, 0, 0};
            for (int i = 0; i < n; ++i) {
                const float* p = &h[i * 6];
                float lo, hi;
                if (c.prod == 0) { lo = fmaf(p[0], p[2], p[4]); hi = fmaf(p[1], p[3], p[5]); }
                else if (c.prod == 1) { lo = p[0] * p[2]; hi = p[1] * p[3]; }
                else { lo = p[0] + p[4]; hi = p[1] + p[5]; }
                uint32_t want, stale_lo, stale_hi;
                if (c.cons == 0) { want = bf16_rne(lo) | (bf16_rne(hi) << 16); stale_lo = bf16_rne(1000.0f) | (bf16_rne(hi) << 16); stale_hi = bf16_rne(lo) | (bf16_rne(1000.0f) << 16); }
                else { float w = lo + hi; memcpy(&want, &w, 4); float a = 1000.0f + hi, b = lo + 1000.0f; memcpy(&stale_lo, &a, 4); memcpy(&stale_hi, &b, 4); }
                const int q = (i & 63) >> 4;
                wrong[q] += o[i * 2] != want;
                stale[q] += (o[i * 2] != want) && (o[i * 2] == stale_lo || o[i * 2] == stale_hi || o[i * 2] == (bf16_rne(1000.0f) * 0x10001u));
                flaky[q] += o[i * 2 + 1] != 0;
            }
            printf("%d waves/block  %-34s wrong per lane quarter %6ld %6ld %6ld %6ld  (stale sentinel %ld %ld %ld %ld)  run-to-run flaky %ld %ld %ld %ld\n", wpb, c.name,
                   wrong[0], wrong[1], wrong[2], wrong[3], stale[0], stale[1], stale[2], stale[3], flaky[0], flaky[1], flaky[2], flaky[3]);
        }
    }
    printf("lanes per quarter: %d\n", n / 4);
    return 0;
}
