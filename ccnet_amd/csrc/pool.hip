// Max pooling kernels: the SPPF cascade (three 5x5 stride-1 pools == windows 5/9/13) and 3x3 stride-2.
// Both are HBM-bound: SPPF reads C channels and writes 3C per pixel, everything else stays in LDS.
#include "common.h"

namespace {

// 16 bytes of channels.  The SPPF cascade compares in an ORDER-PRESERVING INTEGER domain: a finite float's bit pattern becomes
// a monotonic signed integer by flipping its magnitude bits when the sign is set (x ^ ((x >> sign) & magnitude mask), an
// involution), so a max over bf16 pairs is one v_pk_max_i16 per dword instead of unpack / fmaxf / repack (~30 VALU per
// 16 bytes, which made the kernel VALU-bound: a wave64 VALU instruction takes four cycles on a SIMD).  Planes live in LDS
// in that domain; values are mapped on the way in and on the way out.  (-0.0 orders below +0.0; NaNs are not expected.)
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
template <typename T> struct Vec16;
template <> struct Vec16<__bf16> {
    static __device__ __forceinline__ unsigned key1(unsigned x) {
        const s16x2_t sx = __builtin_bit_cast(s16x2_t, x);
        const s16x2_t sg = sx >> (short)15;  // 0 or -1 per half
        return x ^ (__builtin_bit_cast(unsigned, sg) & 0x7FFF7FFFu);
    }
    static __device__ __forceinline__ unsigned max1(unsigned a, unsigned b) {
        const s16x2_t x = __builtin_bit_cast(s16x2_t, a), y = __builtin_bit_cast(s16x2_t, b);
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
    }
    static __device__ __forceinline__ uint4 key(const uint4& a) { return make_uint4(key1(a.x), key1(a.y), key1(a.z), key1(a.w)); }
    static __device__ __forceinline__ uint4 kmax(const uint4& a, const uint4& b) {
        return make_uint4(max1(a.x, b.x), max1(a.y, b.y), max1(a.z, b.z), max1(a.w, b.w));
    }
    static __device__ __forceinline__ uint4 vmax(const uint4& a, const uint4& b) {
        float fa[8], fb[8];
        unpack8(a, fa); unpack8(b, fb);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = fmaxf(fa[i], fb[i]);
        return pack8(fa);
    }
};
template <> struct Vec16<float> {
    static __device__ __forceinline__ unsigned key1(unsigned x) { return x ^ ((unsigned)((int)x >> 31) & 0x7FFFFFFFu); }
    static __device__ __forceinline__ unsigned max1(unsigned a, unsigned b) { return (unsigned)max((int)a, (int)b); }
    static __device__ __forceinline__ uint4 key(const uint4& a) { return make_uint4(key1(a.x), key1(a.y), key1(a.z), key1(a.w)); }
    static __device__ __forceinline__ uint4 kmax(const uint4& a, const uint4& b) {
        return make_uint4(max1(a.x, b.x), max1(a.y, b.y), max1(a.z, b.z), max1(a.w, b.w));
    }
    static __device__ __forceinline__ uint4 vmax(const uint4& a, const uint4& b) {
        float4 x = __builtin_bit_cast(float4, a), y = __builtin_bit_cast(float4, b);
        return __builtin_bit_cast(uint4, make_float4(fmaxf(x.x, y.x), fmaxf(x.y, y.y), fmaxf(x.z, y.z), fmaxf(x.w, y.w)));
    }
};

// One block = one image x `gpb` adjacent 16-byte channel groups; the H x W planes of those groups live in LDS.  Adjacent
// lanes take adjacent groups of one pixel, so a wave's loads and stores cover gpb*16 contiguous bytes per pixel (one group
// per block touched 16 bytes out of every 2 KB pixel row: a quarter of each 64-byte memory access was used).
// torch max_pool2d pads with -inf, i.e. out-of-image taps are skipped.  NaN handling differs from torch
// (fmaxf drops NaN, torch propagates); inputs here are finite activations.
template <typename T>
__global__ __launch_bounds__(256) void sppf_kernel(unsigned char* buf, int ldc, int coff, int C, int H, int W, int lg, int G) {
    extern __shared__ __align__(16) unsigned char dyn[];
    constexpr int EPC = 16 / (int)sizeof(T);
    const int ng = 1 << lg;  // groups per block, a power of two that divides G
    const int g0 = blockIdx.x * ng, b = blockIdx.y;
    const int HW = H * W, items = HW << lg;
    uint4* cur = (uint4*)dyn;          // [HW][ng]: adjacent lanes read adjacent 16-byte slots
    uint4* tmp = cur + items;
    unsigned char* base = buf + ((size_t)b * HW * ldc + coff + g0 * EPC) * sizeof(T);
    const size_t pstride = (size_t)ldc * sizeof(T);
    const float rw = 1.0f / (float)W;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int pix = i >> lg, sub = i & (ng - 1);
        cur[i] = Vec16<T>::key(*(const uint4*)(base + pix * pstride + sub * 16));
    }
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int i = threadIdx.x; i < items; i += 256) {
            const int pix = i >> lg;
            const int y = (int)(((float)pix + 0.5f) * rw), x = pix - y * W;  // exact for H*W <= 5120
            uint4 m = cur[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && (unsigned)(x + d) < (unsigned)W) m = Vec16<T>::kmax(m, cur[i + d * ng]);
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < items; i += 256) {
            const int pix = i >> lg, sub = i & (ng - 1);
            const int y = (int)(((float)pix + 0.5f) * rw);
            uint4 m = tmp[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && (unsigned)(y + d) < (unsigned)H) m = Vec16<T>::kmax(m, tmp[i + d * W * ng]);
            // cur[i] is only read by this thread in this pass (the pass reads tmp), so update in place
            cur[i] = m;
            *(uint4*)(base + pix * pstride + sub * 16 + (size_t)level * C * sizeof(T)) = Vec16<T>::key(m);
        }
        __syncthreads();
    }
}

// Fallback for planes that do not fit in LDS (H*W > 5120, i.e. inputs beyond ~2290 px per side): one cascade level per launch,
// y_level = MaxPool2d(5,1,2)(y_{level-1}) read straight from global memory (25 taps, neighbours hit L1/L2).  Same values as
// the LDS kernel (max is exact); three launches instead of one.
template <typename T>
__global__ __launch_bounds__(256) void sppf_level_kernel(unsigned char* buf, int ldc, int coff, int C, int B, int H, int W, int level) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int G = C / EPC;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W * G;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    const size_t src_c = (size_t)coff + (size_t)(level - 1) * C + (size_t)g * EPC;
    uint4 m = *(const uint4*)(buf + ((size_t)pix * ldc + src_c) * sizeof(T));
    for (int dy = -2; dy <= 2; ++dy) {
        if ((unsigned)(y + dy) >= (unsigned)H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            if ((unsigned)(x + dx) >= (unsigned)W || (dx == 0 && dy == 0)) continue;
            const size_t q = ((size_t)b * H + (y + dy)) * W + (x + dx);
            m = Vec16<T>::vmax(m, *(const uint4*)(buf + (q * ldc + src_c) * sizeof(T)));
        }
    }
    *(uint4*)(buf + ((size_t)pix * ldc + src_c + C) * sizeof(T)) = m;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const unsigned char* in, int in_ldc, int in_coff, int B, int H, int W, int C,
                                                         unsigned char* out, int out_ldc, int out_coff, int Ho, int Wo) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int G = C / EPC;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    long long total = (long long)B * Ho * Wo * G;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    long long pix = idx / G;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    bool have = false;
    uint4 m = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            uint4 v = *(const uint4*)(in + ((((size_t)b * H + iy) * W + ix) * in_ldc + in_coff + g * EPC) * sizeof(T));
            m = have ? Vec16<T>::vmax(m, v) : v;
            have = true;
        }
    *(uint4*)(out + ((size_t)pix * out_ldc + out_coff + g * EPC) * sizeof(T)) = m;
}

}  // namespace

extern "C" int ccnet_sppf_pool(ccnet_slice buf, int dtype, int B, int H, int W, ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && cc_slice_vec_ok(buf, dtype) && buf.shift == 0, "sppf_pool: bad slice");
    CC_REQUIRE(buf.ldc >= buf.coff + 4 * buf.C, "sppf_pool: buffer must hold 4*C channels from coff (ldc=%d coff=%d C=%d)", buf.ldc, buf.coff, buf.C);
    CC_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0, "sppf_pool: bad geometry");
    const int G = buf.C / (16 / cc_esize(dtype));
    const double sppf_bytes = 4.0 * B * H * W * (double)buf.C * cc_esize(dtype);
    if ((size_t)H * W * 32 > 160 * 1024 - 1024) {
        // plane too large for LDS: three global-memory passes (inputs beyond ~2290 px per side)
        const long long total = (long long)B * H * W * G;
        CC_REQUIRE(total < (1ll << 31) * 256, "sppf_pool: too many elements");
        const unsigned blocks = (unsigned)((total + 255) / 256);
        return ccnet_submit(
            [=](hipStream_t st, const float*) -> int {
                for (int level = 1; level <= 3; ++level) {
                    if (dtype == CCNET_BF16)
                        hipLaunchKernelGGL(sppf_level_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, B, H, W, level);
                    else
                        hipLaunchKernelGGL(sppf_level_kernel<float>, dim3(blocks), dim3(256), 0, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, B, H, W, level);
                    CC_LAUNCH_CHECK();
                }
                return CCNET_OK;
            },
            stream, CCNET_OP_SPPF, 0.0, sppf_bytes);
    }
    // channel groups per block: a power of two that divides G, as many as keep a block within ~52 KB of LDS (three blocks per
    // CU), at most 8 (128 contiguous bytes per pixel)
    int lg = 0;
    while (lg < 3 && (G % (2 << lg)) == 0 && (size_t)H * W * 32 * (2 << lg) <= 52 * 1024) ++lg;
    const int gpb = 1 << lg;
    const size_t lds = (size_t)H * W * 32 * gpb;
    const int gx = (G + gpb - 1) / gpb;
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16) {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)sppf_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(sppf_kernel<__bf16>, dim3(gx, B), dim3(256), lds, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, H, W, lg, G);
            } else {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)sppf_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(sppf_kernel<float>, dim3(gx, B), dim3(256), lds, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, H, W, lg, G);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_SPPF, 0.0, sppf_bytes);
}

extern "C" int ccnet_maxpool3x3s2(ccnet_slice in, int dtype, int B, int H, int W, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && cc_slice_vec_ok(in, dtype) && cc_slice_vec_ok(out, dtype) && in.C == out.C && in.shift == 0 && out.shift == 0,
               "maxpool3x3s2: bad slices");
    CC_REQUIRE(B > 0 && H > 0 && W > 0, "maxpool3x3s2: bad geometry");
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    long long total = (long long)B * Ho * Wo * (in.C / (16 / cc_esize(dtype)));
    unsigned blocks = (unsigned)((total + 255) / 256);
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16)
                hipLaunchKernelGGL(maxpool3s2_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)in.ptr, in.ldc, in.coff, B, H, W,
                                   in.C, (unsigned char*)out.ptr, out.ldc, out.coff, Ho, Wo);
            else
                hipLaunchKernelGGL(maxpool3s2_kernel<float>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)in.ptr, in.ldc, in.coff, B, H, W,
                                   in.C, (unsigned char*)out.ptr, out.ldc, out.coff, Ho, Wo);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_MAXPOOL, 0.0, ((double)B * H * W + (double)B * Ho * Wo) * in.C * cc_esize(dtype));
}
