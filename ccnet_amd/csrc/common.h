// Shared host/device helpers for libccnet_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
#include "../../include/ccnet_hip.h"

// ---- error plumbing -------------------------------------------------------------------------------------
int ccnet_set_error(int code, const char* fmt, ...);

#define CC_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return ccnet_set_error(CCNET_EINVAL, __VA_ARGS__); \
    } while (0)

#define CC_HIP(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return ccnet_set_error(CCNET_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define CC_LAUNCH_CHECK()                                                                               \
    do {                                                                                                \
        hipError_t e_ = hipGetLastError();                                                              \
        if (e_ != hipSuccess) return ccnet_set_error(CCNET_EHIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

// A recorded or immediate launch.  x_override replaces the stem image pointer when not null.
using ccnet_launch_fn = std::function<int(hipStream_t, const float* x_override)>;
// Runs `fn` now on `stream`, or appends it to the calling thread's open plan.  kind/flops/bytes describe the op for
// ccnet_plan_op_info (algorithmic work: 2*MAC flops; each distinct input and output element counted once).
int ccnet_submit(ccnet_launch_fn fn, ccnet_stream stream, int kind = 0, double flops = 0.0, double bytes = 0.0);

static inline bool cc_dtype_ok(int d) { return d == CCNET_F32 || d == CCNET_BF16; }
static inline int cc_esize(int d) { return d == CCNET_BF16 ? 2 : 4; }
// read operands: 16-byte vector access on channel groups
static inline bool cc_slice_vec_ok(const ccnet_slice& s, int dtype) {
    int g = 16 / cc_esize(dtype);
    return s.ptr && s.C > 0 && s.ldc >= s.coff + s.C && s.coff >= 0 && (s.ldc % g) == 0 && (s.coff % g) == 0 &&
           (s.C % g) == 0 && ((uintptr_t)s.ptr % 16) == 0 && s.shift >= 0 && s.shift <= 3;
}

// CCNET_NO_XCD_REMAP=1 gives the persistent kernels their plain tile order back (tuning / bisecting aid), see cc_tile_walk
static inline int cc_xcd_remap() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_NO_XCD_REMAP");
        v = (e && atoi(e) != 0) ? 0 : 1;
    }
    return v;
}

// Tile slots (blocks per output-channel slice) of a persistent grid of at most `resident` blocks split into ny slices: as many
// as fit, no more than there are tiles, and a multiple of 8 where possible so that cc_tile_walk_sliced can keep tile
// neighbours and the slices of a tile on one XCD (surplus blocks find their range empty and leave).
static inline int cc_grid_x(int resident, int ny, unsigned ntiles) {
    int gx = resident / ny;
    if (gx < 1) gx = 1;
    if ((unsigned)gx > ntiles) gx = (int)ntiles;
    if (gx >= 8) {
        const int up = (gx + 7) & ~7;
        gx = up * ny <= resident ? up : (gx & ~7);
    }
    return gx;
}

// ---- device helpers -------------------------------------------------------------------------------------
#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Tile walk of a persistent block: tiles first, first + stride, ... below end.  Plain order: block b takes b, b + grid, ...
// The 8 XCDs have an L2 each and blocks are dealt to them round-robin (blocks b and b + 8 share one; observed, not promised:
// a speed matter only), so in plain order the tiles in flight on one L2 are 8 apart and every halo row or column that two
// neighbouring tiles share is fetched through two L2s.  With `xcd` set (and the grid a multiple of 8) the blocks of one
// label walk a contiguous eighth of the tile range together: neighbours in x and y are in flight on the same L2.
// The number of rounds does not change: ceil(ceil(n / 8) / (grid / 8)) == ceil(n / grid).
__device__ __forceinline__ void cc_tile_walk(unsigned ntiles, int xcd, unsigned& first, unsigned& stride, unsigned& end) {
    const unsigned g = gridDim.x, b = blockIdx.x;
    if (xcd && (g & 7u) == 0u) {
        const unsigned per = (ntiles + 7u) >> 3, x = b & 7u;
        stride = g >> 3;
        first = x * per + (b >> 3);
        end = min((x + 1u) * per, ntiles);
    } else {
        first = b;
        stride = g;
        end = ntiles;
    }
}

// The same walk for kernels whose workgroups also split the output channels into ny slices, on a ONE-dimensional grid of
// gx * ny blocks (gx from cc_grid_x: a multiple of 8).  The slices of one tile slot are the blocks L, L + 8, .., L + 8 (ny - 1):
// same XCD, dispatched next to each other, so the input rows they all read cross the fabric once and are L2 hits for the others.
// (With the slices on blockIdx.y they were gx blocks apart in dispatch order, and on different L2s whenever gx % 8 != 0: the
// 512-channel layers read their input 4-6 times from memory, rocprofv3 FETCH_SIZE.)
__device__ __forceinline__ void cc_tile_walk_sliced(unsigned ntiles, int xcd, unsigned ny, unsigned& first, unsigned& stride, unsigned& end,
                                                    unsigned& slice) {
    const unsigned L = blockIdx.x, gx = gridDim.x / ny;
    if (xcd && (gx & 7u) == 0u) {
        const unsigned per = (ntiles + 7u) >> 3, x = L & 7u, idx = L >> 3;
        slice = idx % ny;
        stride = gx >> 3;
        first = x * per + idx / ny;
        end = min((x + 1u) * per, ntiles);
    } else {
        slice = L % ny;
        first = L / ny;
        stride = gx;
        end = ntiles;
    }
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
    __bf16 h = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
// Two floats -> one packed pair with ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN).  Round 2 saw the fused Transformer become
// run-to-run different with this form and kept two scalar conversions; the cause was not the conversion but register allocation:
// the shorter code put LayerNorm's mean into an odd register and LLVM broadcast it with `op_sel:[0,1]`, a packed-fp32 operand form
// that reads zero in lanes 48..63 beside another wave's MFMAs (scripts/hazard/, scripts/check_isa.py lints every build).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const bf16x2_t h = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}
// 8 consecutive channels <-> fp32
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf16_bits_to_f32(v.x & 0xFFFFu); f[1] = bf16_bits_to_f32(v.x >> 16);
    f[2] = bf16_bits_to_f32(v.y & 0xFFFFu); f[3] = bf16_bits_to_f32(v.y >> 16);
    f[4] = bf16_bits_to_f32(v.z & 0xFFFFu); f[5] = bf16_bits_to_f32(v.z >> 16);
    f[6] = bf16_bits_to_f32(v.w & 0xFFFFu); f[7] = bf16_bits_to_f32(v.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// Load / store a group of 8 consecutive channels as fp32, for either storage type.
template <typename T> struct Chan8;
template <> struct Chan8<__bf16> {
    static __device__ __forceinline__ void load(const void* p, float* f) { unpack8(*(const uint4*)p, f); }
    static __device__ __forceinline__ void store(void* p, const float* f) { *(uint4*)p = pack8(f); }
};
template <> struct Chan8<float> {
    static __device__ __forceinline__ void load(const void* p, float* f) {
        float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void store(void* p, const float* f) {
        ((float4*)p)[0] = make_float4(f[0], f[1], f[2], f[3]);
        ((float4*)p)[1] = make_float4(f[4], f[5], f[6], f[7]);
    }
};
template <typename T> __device__ __forceinline__ float load1(const T* p);
template <> __device__ __forceinline__ float load1<__bf16>(const __bf16* p) { return (float)*p; }
template <> __device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <typename T> __device__ __forceinline__ void store1(T* p, float v);
template <> __device__ __forceinline__ void store1<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }
template <> __device__ __forceinline__ void store1<float>(float* p, float v) { *p = v; }

// 1 / (1 + e^-x) from v_exp_f32 and v_rcp_f32 (both ~1 ulp): 5 VALU instructions.  A true IEEE division here costs ~15
// instructions per element and made the conv epilogue as expensive as its MFMAs.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int ACT>
__device__ __forceinline__ float act_fn(float x) {
    if constexpr (ACT == CCNET_ACT_SILU) return x * sigmoidf_(x);
    else if constexpr (ACT == CCNET_ACT_RELU) return fmaxf(x, 0.0f);
    else if constexpr (ACT == CCNET_ACT_RELU6) return fminf(fmaxf(x, 0.0f), 6.0f);
    else if constexpr (ACT == CCNET_ACT_LRELU) return x > 0.0f ? x : 0.1f * x;
    else return x;
}
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case CCNET_ACT_SILU: return act_fn<CCNET_ACT_SILU>(x);
        case CCNET_ACT_RELU: return act_fn<CCNET_ACT_RELU>(x);
        case CCNET_ACT_RELU6: return act_fn<CCNET_ACT_RELU6>(x);
        case CCNET_ACT_LRELU: return act_fn<CCNET_ACT_LRELU>(x);
        default: return x;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif  // __HIPCC__
