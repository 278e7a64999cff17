// Follow-up to ln_chain.hip (gfx950): which part of the chain has to change for the low result of lanes 48..63 to come out right
// while companion waves on the same SIMD issue MFMAs?  One kernel per variant; see VARIANTS below.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CVT_WAR(n) "v_cvt_pk_bf16_f32 %[r" #n "], v200, v201\n"
#define ADD_OPSEL(n) "v_pk_add_f32 v[200:201], %[x" #n "], %[ms] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"     /* mean = ms.y (odd register) */
#define ADD_PLAIN(n) "v_pk_add_f32 v[200:201], %[x" #n "], %[mm] neg_lo:[0,1] neg_hi:[0,1]\n"                  /* mm = {mean, mean} */
#define ADD_OPSELHI(n) "v_pk_add_f32 v[200:201], %[x" #n "], %[sm] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" /* mean = sm.x (even register) */
#define MUL "v_pk_mul_f32 v[200:201], v[200:201], %[rs] op_sel_hi:[1,0]\n"
#define FMA "v_pk_fma_f32 v[200:201], v[200:201], %[g], %[b]\n"
#define N0 "s_nop 0\n"
#define N1 "s_nop 1\n"
#define N3 "s_nop 3\n"
#define VM "v_mov_b32 v202, v203\n"

__device__ __forceinline__ uint32_t bf16_rne(float f) { uint32_t u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }

#define CHAIN_KERNEL(NAME, S0, S1, S2, S3)                                                                                              \
    __global__ __launch_bounds__(512) void NAME(const float* in, uint32_t* out, int iters) {                                            \
        const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                          \
        uint32_t bad_lo = 0, bad_hi = 0, skipped = 0;                                                                                   \
        if (wave >= 4) {                                                                                                                \
            f4 acc = {0.f, 0.f, 0.f, 0.f};                                                                                              \
            bf16x8 a, b;                                                                                                                \
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(e - lane); }                         \
            for (int k = 0; k < iters * 24; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);                     \
            out[i * 3] = 0; out[i * 3 + 1] = 0; out[i * 3 + 2] = acc[0] == 12345.f;                                                     \
            return;                                                                                                                     \
        }                                                                                                                               \
        f2 x[4];                                                                                                                        \
        for (int j = 0; j < 4; ++j) x[j] = f2{in[i * 8 + 2 * j], in[i * 8 + 2 * j + 1]};                                                \
        for (int k = 0; k < iters; ++k) {                                                                                               \
            const float mean = 0.03125f * (float)((lane + k) % 17 + 1);                                                                 \
            const f2 ms = {2.0f + k, mean}, mm = {mean, mean}, sm = {mean, 2.0f + k};                                                   \
            const f2 rs = {0.75f + 0.001f * (float)(k & 63), 1e30f};                                                                    \
            const f2 g = {1.0625f, 0.9375f}, b = {0.125f, -0.0625f};                                                                    \
            uint32_t r0, r1, r2, r3, r4;                                                                                                \
            asm volatile("v_mov_b32 v200, 0\n v_mov_b32 v201, 0\n" S0 S1 S2 S3 "s_nop 7\n v_cvt_pk_bf16_f32 %[r4], v200, v201\n"        \
                         : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4)                               \
                         : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [ms] "v"(ms), [mm] "v"(mm), [sm] "v"(sm),    \
                           [rs] "v"(rs), [g] "v"(g), [b] "v"(b)                                                                         \
                         : "v200", "v201", "v202");                                                                                     \
            const uint32_t rr[4] = {r1, r2, r3, r4};                                                                                    \
            for (int j = 0; j < 4; ++j) {                                                                                               \
                const float lo = fmaf((x[j].x - mean) * rs.x, g.x, b.x), hi = fmaf((x[j].y - mean) * rs.x, g.y, b.y);                   \
                const float lo_nomean = fmaf(x[j].x * rs.x, g.x, b.x);                                                                  \
                bad_lo += (rr[j] & 0xffffu) != bf16_rne(lo);                                                                            \
                bad_hi += (rr[j] >> 16) != bf16_rne(hi);                                                                                \
                skipped += ((rr[j] & 0xffffu) != bf16_rne(lo)) && ((rr[j] & 0xffffu) == bf16_rne(lo_nomean));                           \
            }                                                                                                                           \
        }                                                                                                                               \
        out[i * 3] = bad_lo; out[i * 3 + 1] = bad_hi; out[i * 3 + 2] = skipped;                                                         \
    }

#define ST(n, PRE, ADD, GAP1, GAP2, GAP3) PRE ADD GAP1 MUL GAP2 FMA GAP3
// the cvt of step n+1 converts step n's result; "r0" of the first step converts zeros
#define CHAIN4(NAME, PRE0, PRE1, PRE2, PRE3, ADD0, ADD1, ADD2, ADD3, G1, G2, G3) \
    CHAIN_KERNEL(NAME, ST(0, PRE0, ADD0, G1, G2, G3), ST(1, PRE1, ADD1, G1, G2, G3), ST(2, PRE2, ADD2, G1, G2, G3), ST(3, PRE3, ADD3, G1, G2, G3))

CHAIN4(k_base, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N0, N0, N0)
CHAIN4(k_gap1_2, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N1, N0, N0)
CHAIN4(k_gap1_4, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N3, N0, N0)
CHAIN4(k_gap1_valu, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), VM, N0, N0)
CHAIN4(k_gap3_4, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N0, N0, N3)
CHAIN4(k_pre_nop, CVT_WAR(0) N1, CVT_WAR(1) N1, CVT_WAR(2) N1, CVT_WAR(3) N1, ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N0, N0, N0)
CHAIN4(k_plain, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_PLAIN(0), ADD_PLAIN(1), ADD_PLAIN(2), ADD_PLAIN(3), N0, N0, N0)
CHAIN4(k_opselhi, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSELHI(0), ADD_OPSELHI(1), ADD_OPSELHI(2), ADD_OPSELHI(3), N0, N0, N0)
CHAIN4(k_nogaps, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), "", "", "")
CHAIN4(k_allgap2, CVT_WAR(0), CVT_WAR(1), CVT_WAR(2), CVT_WAR(3), ADD_OPSEL(0), ADD_OPSEL(1), ADD_OPSEL(2), ADD_OPSEL(3), N1, N1, N1)

typedef void (*kern_t)(const float*, uint32_t*, int);
int main() {
    const int n = 512 * 2048, iters = 200;
    std::vector<float> h((size_t)n * 8);
    uint32_t s = 4242u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 8192.0f - 4.0f; }
    float* d_in; uint32_t* d_out;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, (size_t)n * 12);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    struct { const char* name; kern_t k; } v[] = {
        {"as scheduled (s_nop 0 gaps)", k_base}, {"pk_add -> s_nop 1 -> pk_mul", k_gap1_2}, {"pk_add -> s_nop 3 -> pk_mul", k_gap1_4},
        {"pk_add -> v_mov -> pk_mul", k_gap1_valu}, {"pk_fma -> s_nop 3 -> cvt", k_gap3_4}, {"cvt -> s_nop 1 -> pk_add", k_pre_nop},
        {"pk_add without op_sel ({m,m})", k_plain}, {"pk_add op_sel_hi:[1,0] (mean in the even reg)", k_opselhi},
        {"no s_nop at all", k_nogaps}, {"every gap s_nop 1", k_allgap2}};
    std::vector<uint32_t> o((size_t)n * 3);
    for (auto& c : v) {
        hipMemset(d_out, 0, (size_t)n * 12);
        hipLaunchKernelGGL(c.k, dim3(n / 512), dim3(512), 0, 0, d_in, d_out, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(o.data(), d_out, (size_t)n * 12, hipMemcpyDeviceToHost);
        long lo[4] = {0}, hi[4] = {0}, sk[4] = {0};
        for (int i = 0; i < n; ++i) { if (((i & 511) >> 6) >= 4) continue; const int q = (i & 63) >> 4; lo[q] += o[(size_t)i * 3]; hi[q] += o[(size_t)i * 3 + 1]; sk[q] += o[(size_t)i * 3 + 2]; }
        printf("%-46s wrong LOW per lane quarter %ld %ld %ld %ld | wrong HIGH %ld %ld %ld %ld | low without the mean %ld %ld %ld %ld\n", c.name, lo[0], lo[1], lo[2], lo[3], hi[0],
               hi[1], hi[2], hi[3], sk[0], sk[1], sk[2], sk[3]);
    }
    printf("packed results checked per lane quarter and variant: %ld\n", (long)n / 2 / 4 * iters * 4);
    return 0;
}
