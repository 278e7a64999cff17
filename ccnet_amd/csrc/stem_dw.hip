// Network stem (6x6 stride-2 pad-2 conv on the NCHW fp32 image, Cin = 3) and the depthwise 3x3 of MobileViT.
//
// Stem: one block = 8 x 32 output pixels.  The 20 x 68 x 3 input patch is staged once in LDS (as bf16 or fp32),
// the folded weights [Cout][160] too; K is laid out as (c, ky) pairs x 8 kx slots (slots 6,7 and pairs 18,19 carry
// zero weights) so that a lane's 8-element MFMA fragment is 8 consecutive patch columns.  Memory bound:
// 12 B in + 2*Cout B out per output pixel.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;            // output tile
constexpr int PH = 2 * TH + 4;            // 20 patch rows
constexpr int PW = 2 * TW + 4;            // 68 patch cols
constexpr int PWP = 72;                   // padded row (elements)
constexpr int KS = 160;                   // padded K
constexpr int NPAIR = 18;

struct StemP {
    const float* x;
    int B, H, W, Ho, Wo;
    const float* w;     // [Cout][160] fp32
    const float* bias;  // [Cout]
    int Cout, act;
    unsigned char* out;
    int out_ldc, out_coff;
};

template <typename T, typename OutT, int NT>
__global__ __launch_bounds__(256) void stem_kernel(const StemP p, const float* x) {
    constexpr int CP = NT * 16;
    constexpr int EPS = CP + 4;
    __shared__ __align__(16) T patch[3 * PH * PWP];
    __shared__ __align__(16) T wl[CP * KS];
    __shared__ __align__(16) float ep[4 * 16 * EPS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;
    const float* xb = x + (size_t)b * 3 * p.H * p.W;
    for (int e = t; e < 3 * PH * PWP; e += 256) {
        int c = e / (PH * PWP), r = (e / PWP) % PH, col = e % PWP;
        int iy = iy0 + r, ix = ix0 + col;
        float v = 0.0f;
        if (col < PW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = xb[((size_t)c * p.H + iy) * p.W + ix];
        patch[e] = (T)v;
    }
    for (int e = t; e < CP * KS; e += 256) {
        int n = e / KS;
        wl[e] = (T)(n < p.Cout ? p.w[e] : 0.0f);
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    // wave w owns output rows 2w, 2w+1; M tile mt -> (row 2w + mt/2, cols (mt&1)*16 ..)
    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            int pair = ks * 4 + fq;
            if (pair >= NPAIR) pair = 0;  // zero weights there; read something finite from this pixel's own window
            const int c = pair / 6, ky = pair - c * 6;
            uint4 fb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(wl + (j * 16 + fr) * KS + ks * 32 + fq * 8);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ly = 2 * wave + (mt >> 1), lx = (mt & 1) * 16 + fr;
                const uint32_t* src = (const uint32_t*)(patch + (c * PH + 2 * ly + ky) * PWP + 2 * lx);
                uint4 fa = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[j]), acc[mt][j], 0, 0, 0);
            }
        }
    } else {
        for (int s = 0; s < KS / 4; ++s) {
            const int k = 4 * s + fq;
            int pair = k >> 3;
            const int kx = k & 7;
            if (pair >= NPAIR) pair = 0;
            const int c = pair / 6, ky = pair - c * 6;
            float fb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = (float)wl[(j * 16 + fr) * KS + k];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ly = 2 * wave + (mt >> 1), lx = (mt & 1) * 16 + fr;
                float fa = (float)patch[(c * PH + 2 * ly + ky) * PWP + 2 * lx + kx];
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb[j], acc[mt][j], 0, 0, 0);
            }
        }
    }

    float* epw = ep + wave * 16 * EPS;
    constexpr int UPR = CP / 8;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) epw[(fq * 4 + r) * EPS + j * 16 + fr] = acc[mt][j][r];
        __syncthreads();
        const int oy = oy0 + 2 * wave + (mt >> 1);
        for (int u = lane; u < 16 * UPR; u += 64) {
            const int row = u / UPR, cu = u - row * UPR;
            const int ox = ox0 + (mt & 1) * 16 + row;
            const int n = cu * 8;
            if (oy >= p.Ho || ox >= p.Wo || n >= p.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act(epw[row * EPS + n + e] + p.bias[n + e], p.act);
            size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
            Chan8<OutT>::store(p.out + (pix * p.out_ldc + p.out_coff + n) * sizeof(OutT), v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const unsigned char* in, int in_ldc, int in_coff, int B, int H, int W, int C,
                                                        const float* w, const float* bias, unsigned char* out, int out_ldc,
                                                        int out_coff) {
    const int G = C / 8;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    long long total = (long long)B * H * W * G;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    long long pix = idx / G;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias[g * 8 + e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int iy = y + ky - 1, ix = x + kx - 1;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            float v[8];
            Chan8<T>::load(in + ((((size_t)b * H + iy) * W + ix) * in_ldc + in_coff + g * 8) * sizeof(T), v);
            const float* wt = w + (ky * 3 + kx) * C + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(v[e], wt[e], acc[e]);
        }
    Chan8<T>::store(out + ((size_t)pix * out_ldc + out_coff + g * 8) * sizeof(T), acc);
}

template <typename T, typename OutT>
int launch_stem(const StemP& p, const float* x, hipStream_t st) {
    dim3 grid((p.Wo + TW - 1) / TW, (p.Ho + TH - 1) / TH, p.B);
    int nt = (p.Cout + 15) / 16;
    switch (nt) {
        case 1: hipLaunchKernelGGL((stem_kernel<T, OutT, 1>), grid, dim3(256), 0, st, p, x); break;
        case 2: hipLaunchKernelGGL((stem_kernel<T, OutT, 2>), grid, dim3(256), 0, st, p, x); break;
        case 3: hipLaunchKernelGGL((stem_kernel<T, OutT, 3>), grid, dim3(256), 0, st, p, x); break;
        case 4: hipLaunchKernelGGL((stem_kernel<T, OutT, 4>), grid, dim3(256), 0, st, p, x); break;
        default: hipLaunchKernelGGL((stem_kernel<T, OutT, 5>), grid, dim3(256), 0, st, p, x); break;
    }
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

}  // namespace

extern "C" int ccnet_stem_conv(const float* x_dev, int B, int H, int W, const float* weight, const float* bias, int Cout, int act,
                               ccnet_slice out, int out_dtype, ccnet_stream stream) {
    CC_REQUIRE(x_dev && weight && bias, "stem_conv: null pointer");
    CC_REQUIRE(B > 0 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0, "stem_conv: H,W must be even (got %d,%d)", H, W);
    CC_REQUIRE(B <= 65535, "stem_conv: batch too large for one launch");
    CC_REQUIRE(Cout > 0 && Cout <= 80 && (Cout % 8) == 0, "stem_conv: Cout must be a multiple of 8, at most 80 (got %d)", Cout);
    CC_REQUIRE(cc_dtype_ok(out_dtype) && cc_slice_vec_ok(out, out_dtype) && (out.ldc % 8) == 0 && (out.coff % 8) == 0 && out.C == Cout &&
                   out.shift == 0, "stem_conv: bad output slice");
    CC_REQUIRE(act >= CCNET_ACT_NONE && act <= CCNET_ACT_LRELU, "stem_conv: unknown activation");
    StemP p{x_dev, B, H, W, H / 2, W / 2, weight, bias, Cout, act, (unsigned char*)out.ptr, out.ldc, out.coff};
    return ccnet_submit(
        [p, out_dtype](hipStream_t st, const float* xo) -> int {
            const float* x = xo ? xo : p.x;
            return out_dtype == CCNET_BF16 ? launch_stem<__bf16, __bf16>(p, x, st) : launch_stem<float, float>(p, x, st);
        },
        stream, CCNET_OP_STEM, 2.0 * B * (H / 2) * (W / 2) * (double)Cout * 108.0,
        (double)B * 3 * H * W * 4 + (double)B * (H / 2) * (W / 2) * Cout * cc_esize(out_dtype));
}

extern "C" int ccnet_dwconv3x3(ccnet_slice in, int dtype, int B, int H, int W, const float* weight, const float* bias, ccnet_slice out,
                               ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && weight && bias, "dwconv3x3: bad arguments");
    CC_REQUIRE(cc_slice_vec_ok(in, dtype) && cc_slice_vec_ok(out, dtype) && in.C == out.C && (in.C % 8) == 0 && in.shift == 0 && out.shift == 0 &&
                   (in.ldc % 8) == 0 && (in.coff % 8) == 0 && (out.ldc % 8) == 0 && (out.coff % 8) == 0, "dwconv3x3: bad slices");
    CC_REQUIRE(B > 0 && H > 0 && W > 0, "dwconv3x3: bad geometry");
    long long total = (long long)B * H * W * (in.C / 8);
    unsigned blocks = (unsigned)((total + 255) / 256);
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16)
                hipLaunchKernelGGL(dwconv3x3_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)in.ptr, in.ldc, in.coff, B, H, W,
                                   in.C, weight, bias, (unsigned char*)out.ptr, out.ldc, out.coff);
            else
                hipLaunchKernelGGL(dwconv3x3_kernel<float>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)in.ptr, in.ldc, in.coff, B, H, W,
                                   in.C, weight, bias, (unsigned char*)out.ptr, out.ldc, out.coff);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_DWCONV, 2.0 * B * H * W * (double)in.C * 9.0, 2.0 * B * H * W * (double)in.C * cc_esize(dtype));
}
