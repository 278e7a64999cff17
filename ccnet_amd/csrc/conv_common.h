// Pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_patch.hip).
#pragma once
#include "common.h"

namespace {

constexpr int BKB = 128;  // bytes of K per row per step

// native vector type for everything staged through registers (arrays of HIP's uint4 struct end up in scratch/LDS)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SegQ {
    const unsigned char* ptr;  // first element of the slice (channel offset applied)
    unsigned ldcB;             // pixel stride in bytes
    int shift, Hs, Ws;
    unsigned span;             // bytes addressable from ptr (< 2^31)
};

struct ConvQ {
    SegQ seg[2];
    int nseg, Ctot, C0;
    int H, W, kh, kw, stride, pad, Ho, Wo, Cout;
    unsigned M, ntiles;
    unsigned magicW, shiftW, magicH, shiftH;  // n / Wo, n / Ho by multiply-high
    const unsigned char* weight;
    int Kpad, nk;
    const float* bias;
    int act;
    const unsigned char* res_ptr;
    int res_ldc, res_coff;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int vec_out;
    int flat;  // 1x1 stride-1 unpadded, no upsampled operand: input pixel index == output pixel index
    int xcd;   // tile order, see cc_tile_walk
};

template <typename T> struct Frag;
template <> struct Frag<__bf16> {
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Frag<float> {
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& acc) {
        f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BKB + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned magic, unsigned shift) { return (__umulhi(n, magic) + n) >> shift; }


// ACT: activation compiled into the epilogue (CCNET_ACT_*), or ACT_RUNTIME to branch on p.act per element.
constexpr int ACT_RUNTIME = -1;

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")

#ifdef CC_STAMP  // diagnostic build only: per-phase cycle totals of the main loop (scripts/conv_bench.py prints them)
__device__ unsigned long long g_stamps[8];
#define STAMP(i)                                                      \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += now_ - st_last;                                  \
        st_last = now_;                                               \
    } while (0)
#else
#define STAMP(i)
#endif


}  // namespace
