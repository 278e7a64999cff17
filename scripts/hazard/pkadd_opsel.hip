// Micro-test (gfx950): v_pk_add_f32 whose LOW lane takes the HIGH register of the src1 pair (op_sel:[0,1], negated) -- the form
// LLVM emits for "x - mean" when mean sits in the odd register of a pair -- executed while vector-memory loads of the same wave
// are returning into other registers.  In the fused Transformer's LayerNorm this instruction occasionally returned x - 0 in the low
// result of lanes 48..63.  Each lane checks every result against x - m computed with plain v_sub_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

#define PK(n, d) "v_pk_add_f32 %" #d ", %[x" #n "], %[m] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
#define LD(r, o) "global_load_dwordx4 v[" #r "], %[voff], %[base] offset:" #o "\n"

template <int MODE>   // 0: loads in flight while the packed adds run; 1: loads waited for first; 2: no loads
__global__ void pkadd_kernel(const float* in, const float* tab, uint32_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    f2 x0 = {in[i * 8 + 0], in[i * 8 + 1]}, x1 = {in[i * 8 + 2], in[i * 8 + 3]}, x2 = {in[i * 8 + 4], in[i * 8 + 5]}, x3 = {in[i * 8 + 6], in[i * 8 + 7]};
    uint32_t bad_lo = 0, bad_hi = 0, lo_is_x = 0;
    for (int k = 0; k < iters; ++k) {
        f2 m = {1.0f + k, 0.25f + 0.001f * (float)(lane + k)};   // .y is the "mean" (odd register of the pair), .x something else
        f2 d0, d1, d2, d3, d4, d5, d6, d7;
        const int voff = ((i * 131 + k * 7919) & 0xffff) * 16;   // scattered 16-byte loads: varying return times
        if (MODE == 0)
            asm volatile(LD(100:103, 0) LD(104:107, 1024) LD(108:111, 2048) LD(112:115, 3072) LD(116:119, 16) LD(120:123, 1040) LD(124:127, 2064) LD(128:131, 3088)
                         "s_nop 3\n"
                         PK(0, 0) PK(1, 1) PK(2, 2) PK(3, 3) PK(0, 4) PK(1, 5) PK(2, 6) PK(3, 7)
                         "s_waitcnt vmcnt(0)\n"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                         : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "v"(m), [voff] "v"(voff), [base] "s"(tab)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                           "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
        else if (MODE == 1)
            asm volatile(LD(100:103, 0) LD(104:107, 1024) LD(108:111, 2048) LD(112:115, 3072) LD(116:119, 16) LD(120:123, 1040) LD(124:127, 2064) LD(128:131, 3088)
                         "s_waitcnt vmcnt(0)\n"
                         PK(0, 0) PK(1, 1) PK(2, 2) PK(3, 3) PK(0, 4) PK(1, 5) PK(2, 6) PK(3, 7)
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                         : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "v"(m), [voff] "v"(voff), [base] "s"(tab)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                           "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131");
        else
            asm volatile(PK(0, 0) PK(1, 1) PK(2, 2) PK(3, 3) PK(0, 4) PK(1, 5) PK(2, 6) PK(3, 7)
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
                         : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "v"(m), [voff] "v"(voff), [base] "s"(tab));
        const f2 xs[8] = {x0, x1, x2, x3, x0, x1, x2, x3};
        const f2 ds[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bad_lo += ds[j].x != xs[j].x - m.y;
            bad_hi += ds[j].y != xs[j].y - m.y;
            lo_is_x += ds[j].x == xs[j].x;
        }
    }
    out[i * 3] = bad_lo; out[i * 3 + 1] = bad_hi; out[i * 3 + 2] = lo_is_x;
}

template <int MODE>
static void run(const char* name, const float* d_in, const float* d_tab, uint32_t* d_out, int n, int wpb, int iters) {
    std::vector<uint32_t> o(n * 3);
    hipMemset(d_out, 0, n * 12);
    hipLaunchKernelGGL(pkadd_kernel<MODE>, dim3(n / (wpb * 64)), dim3(wpb * 64), 0, 0, d_in, d_tab, d_out, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    hipMemcpy(o.data(), d_out, n * 12, hipMemcpyDeviceToHost);
    long lo[4] = {0}, hi[4] = {0}, isx[4] = {0};
    for (int i = 0; i < n; ++i) { const int q = (i & 63) >> 4; lo[q] += o[i * 3]; hi[q] += o[i * 3 + 1]; isx[q] += o[i * 3 + 2]; }
    printf("%d waves/block %-28s wrong LOW results per lane quarter %ld %ld %ld %ld | wrong HIGH %ld %ld %ld %ld | low == x (mean not subtracted) %ld %ld %ld %ld\n", wpb,
           name, lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3], isx[0], isx[1], isx[2], isx[3]);
}

int main() {
    const int n = 256 * 8 * 64 * 4, iters = 200;
    std::vector<float> h(n * 8);
    uint32_t s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 8192.0f - 4.0f; }
    float *d_in, *d_tab; uint32_t* d_out;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_tab, (size_t)(0x10000 * 16 + 8192)); hipMalloc(&d_out, n * 12);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_tab, 0, (size_t)(0x10000 * 16 + 8192));
    for (int wpb : {1, 4, 8, 16}) {
        run<0>("loads in flight", d_in, d_tab, d_out, n, wpb, iters);
        run<1>("loads waited for", d_in, d_tab, d_out, n, wpb, iters);
        run<2>("no loads", d_in, d_tab, d_out, n, wpb, iters);
    }
    printf("results checked per lane quarter and variant: %ld\n", (long)n / 4 * iters * 8);
    return 0;
}
