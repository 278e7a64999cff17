// Max pooling kernels: the SPPF cascade (three 5x5 stride-1 pools == windows 5/9/13) and 3x3 stride-2.
// Both are HBM-bound: SPPF reads C channels and writes 3C per pixel, everything else stays in LDS.
#include "common.h"

namespace {

template <typename T> struct Vec16;  // 16 bytes of channels, max in the element domain
template <> struct Vec16<__bf16> {
    static __device__ __forceinline__ uint4 vmax(const uint4& a, const uint4& b) {
        float fa[8], fb[8];
        unpack8(a, fa); unpack8(b, fb);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = fmaxf(fa[i], fb[i]);
        return pack8(fa);
    }
};
template <> struct Vec16<float> {
    static __device__ __forceinline__ uint4 vmax(const uint4& a, const uint4& b) {
        float4 x = __builtin_bit_cast(float4, a), y = __builtin_bit_cast(float4, b);
        return __builtin_bit_cast(uint4, make_float4(fmaxf(x.x, y.x), fmaxf(x.y, y.y), fmaxf(x.z, y.z), fmaxf(x.w, y.w)));
    }
};

// One block = one image x one 16-byte channel group; the whole H x W plane of that group lives in LDS.
// torch max_pool2d pads with -inf, i.e. out-of-image taps are skipped.  NaN handling differs from torch
// (fmaxf drops NaN, torch propagates); inputs here are finite activations.
template <typename T>
__global__ __launch_bounds__(256) void sppf_kernel(unsigned char* buf, int ldc, int coff, int C, int H, int W) {
    extern __shared__ __align__(16) unsigned char dyn[];
    uint4* cur = (uint4*)dyn;
    uint4* tmp = cur + H * W;
    constexpr int EPC = 16 / (int)sizeof(T);
    const int g = blockIdx.x, b = blockIdx.y;
    const int HW = H * W;
    unsigned char* base = buf + ((size_t)b * HW * ldc + coff + g * EPC) * sizeof(T);
    const size_t pstride = (size_t)ldc * sizeof(T);
    for (int i = threadIdx.x; i < HW; i += 256) cur[i] = *(const uint4*)(base + i * pstride);
    __syncthreads();
    for (int level = 1; level <= 3; ++level) {
        for (int i = threadIdx.x; i < HW; i += 256) {
            int y = i / W, x = i - y * W;
            uint4 m = cur[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                int xx = x + d;
                if (d != 0 && (unsigned)xx < (unsigned)W) m = Vec16<T>::vmax(m, cur[y * W + xx]);
            }
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += 256) {
            int y = i / W, x = i - y * W;
            uint4 m = tmp[i];
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                int yy = y + d;
                if (d != 0 && (unsigned)yy < (unsigned)H) m = Vec16<T>::vmax(m, tmp[yy * W + x]);
            }
            // cur[i] is only read by this thread in this pass (the pass reads tmp), so update in place
            cur[i] = m;
            *(uint4*)(base + i * pstride + (size_t)level * C * sizeof(T)) = m;
        }
        __syncthreads();
    }
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
    size_t lds = (size_t)H * W * 16 * 2;
    CC_REQUIRE(lds <= 160 * 1024, "sppf_pool: %dx%d plane does not fit in LDS (max H*W = 5120)", H, W);
    int G = buf.C / (16 / cc_esize(dtype));
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16) {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)sppf_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(sppf_kernel<__bf16>, dim3(G, B), dim3(256), lds, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, H, W);
            } else {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)sppf_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(sppf_kernel<float>, dim3(G, B), dim3(256), lds, st, (unsigned char*)buf.ptr, buf.ldc, buf.coff, buf.C, H, W);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_SPPF, 0.0, 4.0 * B * H * W * (double)buf.C * cc_esize(dtype));
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
