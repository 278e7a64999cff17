// Network stem (6x6 stride-2 pad-2 conv on the NCHW fp32 image, Cin = 3) and the depthwise 3x3 of MobileViT.
//
// Stem: persistent blocks walk tiles of 8 x 32 output pixels.  The folded weights [Cout][160] are staged in LDS once per
// block; per tile the 3 x 20 x 68 input patch is fetched as float2 pairs into registers one tile ahead, converted and
// written to one of two LDS patch buffers (one barrier per tile).  K is laid out as (c, ky) pairs x 8 kx slots (slots
// 6,7 and pairs 18,19 carry zero weights) so that a lane's 8-element MFMA fragment is 8 consecutive patch columns.
// The weights are the MFMA "A" operand: a lane ends up with 4 consecutive channels of one pixel and stores them
// straight from registers.  Memory bound: 12 B in + sizeof(OutT)*Cout B out per output pixel.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;            // output tile
constexpr int PH = 2 * TH + 4;            // 20 patch rows
constexpr int PW = 2 * TW + 4;            // 68 patch cols
constexpr int PWP = 72;                   // padded row (elements)
constexpr int KS = 160;                   // padded K
constexpr int NPAIR = 18;
constexpr int NSLOT = 8;                  // float2 fetches per thread and tile: 3 * 20 * 34 = 2040 <= 8 * 256

struct StemP {
    const float* x;
    int B, H, W, Ho, Wo;
    const float* w;     // [Cout][160] fp32
    const float* bias;  // [Cout]
    int Cout, act;
    unsigned char* out;
    int out_ldc, out_coff;
    int tiles_x, tiles_y;
    unsigned ntiles;
    int xcd;  // tile order, see cc_tile_walk
};

// Four waves per SIMD for the narrow stems (second launch-bound argument: waves per SIMD): the kernel waits on its fetches,
// and a fourth resident block per CU is worth more than the 4 registers it spills (0.226 -> 0.214 ms at 64 x 640 x 640).
template <typename T, typename OutT, int NT>
__global__ __launch_bounds__(256, NT <= 2 ? 4 : 1) void stem_kernel(const StemP p, const float* x) {
    constexpr int CP = NT * 16;
    __shared__ __align__(16) T patch[2][3 * PH * PWP];
    __shared__ __align__(16) T wl[CP * KS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // PAIR8 (two n-tiles, 2-byte outputs -- the s-width stem): LDS row j*16 + q*4 + r holds output channel q*8 + j*4 + r, so that a
    // lane's registers of the two tiles are the 8 consecutive channels q*8 .. q*8+7 of its pixel: one 16-byte store per pixel, and a
    // wave's store instruction covers 16 consecutive pixels x 64 bytes = 1 KB of whole lines (separately the two tiles wrote 32-byte
    // pieces at a 64-byte stride: 13 M write requests of 32 bytes for 419 MB)
    constexpr bool PAIR8 = NT == 2 && sizeof(OutT) == 2;
    auto chan_of = [](int n) { return PAIR8 ? ((n >> 2) & 3) * 8 + (n >> 4) * 4 + (n & 3) : n; };
    for (int e = t; e < CP * KS; e += 256) {
        const int n = e / KS, k = e - n * KS;
        const int c = chan_of(n);
        wl[e] = (T)(c < p.Cout ? p.w[c * KS + k] : 0.0f);
    }
    // columns PW..PWP-1 of every patch row are never fetched but are read under the zero-weight kx slots 6,7
    static_assert(PW + 4 == PWP, "pad columns");
    for (int e = t; e < 2 * 3 * PH * 4; e += 256) patch[e / (3 * PH * 4)][((e >> 2) % (3 * PH)) * PWP + PW + (e & 3)] = (T)0.0f;
    float breg[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = chan_of(j * 16 + fq * 4 + r);
            breg[j][r] = n < p.Cout ? p.bias[n] : 0.0f;
        }

    // this thread's fetch slots: pair index e -> (channel, patch row, column pair)
    int s_src[NSLOT], s_lds[NSLOT], s_r[NSLOT], s_c2[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int e = t + 256 * i;
        const int c = e / (PH * 34), rem = e - c * (PH * 34);
        const int r = rem / 34, cp = rem - r * 34;
        const bool live = e < 3 * PH * 34;
        s_src[i] = (c * p.H + r) * p.W + 2 * cp;
        s_lds[i] = live ? (c * PH + r) * PWP + 2 * cp : -1;
        s_r[i] = r;
        s_c2[i] = 2 * cp;
    }
    const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
    auto origin = [&](unsigned tile, int& b, int& oy0, int& ox0) {
        b = (int)(tile / per_img);
        const unsigned rem = tile - (unsigned)b * per_img;
        const int ty = (int)(rem / (unsigned)p.tiles_x);
        oy0 = ty * TH;
        ox0 = (int)(rem - (unsigned)ty * p.tiles_x) * TW;
    };
    float2 regs[NSLOT];
    auto fetch = [&](unsigned tile) {
        int b, oy0, ox0;
        origin(tile, b, oy0, ox0);
        const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;  // both even: a pair is inside or outside as a whole
        const float* base = x + (size_t)b * 3 * p.H * p.W + (long long)iy0 * p.W + ix0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const bool ok = s_lds[i] >= 0 && (unsigned)(iy0 + s_r[i]) < (unsigned)p.H && (unsigned)(ix0 + s_c2[i]) < (unsigned)p.W;
            regs[i] = ok ? *(const float2*)(base + s_src[i]) : make_float2(0.0f, 0.0f);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i)
            if (s_lds[i] >= 0) {
                if constexpr (sizeof(T) == 2) *(uint32_t*)(&patch[buf][s_lds[i]]) = pack_bf16x2(regs[i].x, regs[i].y);
                else *(float2*)(&patch[buf][s_lds[i]]) = regs[i];
            }
    };

    unsigned tile, tstride, tend;
    cc_tile_walk(p.ntiles, p.xcd, tile, tstride, tend);
    if (tile >= tend) return;
    fetch(tile);
    for (int it = 0; tile < tend; tile += tstride, ++it) {
        const int buf = it & 1;
        stash(buf);
        __syncthreads();  // also orders the weight fill; the other buffer was last read before the previous barrier
        const unsigned next = tile + tstride;
        if (next < tend) fetch(next);  // in flight during the MFMAs

        const T* pt = patch[buf];
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
                    const uint32_t* src = (const uint32_t*)(pt + (c * PH + 2 * ly + ky) * PWP + 2 * lx);
                    uint4 fa = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa), acc[mt][j], 0, 0, 0);
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
                    float fa = (float)pt[(c * PH + 2 * ly + ky) * PWP + 2 * lx + kx];
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa, acc[mt][j], 0, 0, 0);
                }
            }
        }

        // lane (fr, fq) of accumulator (mt, j) holds channels j*16 + fq*4 .. +3 of pixel (row 2w + mt/2, col (mt&1)*16 + fr)
        int b, oy0, ox0;
        origin(tile, b, oy0, ox0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + 2 * wave + (mt >> 1), ox = ox0 + (mt & 1) * 16 + fr;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const size_t pix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
            if constexpr (PAIR8) {
                if (fq * 8 >= p.Cout) continue;
                uint32_t pk[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = acc[mt][j][r] + breg[j][r];
                        v[r] = p.act == CCNET_ACT_SILU ? act_fn<CCNET_ACT_SILU>(a) : apply_act(a, p.act);
                    }
                    pk[2 * j] = pack_bf16x2(v[0], v[1]);
                    pk[2 * j + 1] = pack_bf16x2(v[2], v[3]);
                }
                *(uint4*)(p.out + (pix * p.out_ldc + p.out_coff + fq * 8) * 2) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                continue;
            }
            unsigned char* orow = p.out + (pix * p.out_ldc + p.out_coff + fq * 4) * sizeof(OutT);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j * 16 + fq * 4 >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[mt][j][r] + breg[j][r];
                    v[r] = p.act == CCNET_ACT_SILU ? act_fn<CCNET_ACT_SILU>(a) : apply_act(a, p.act);
                }
                if constexpr (sizeof(OutT) == 2) *(uint2*)(orow + j * 16 * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else *(float4*)(orow + j * 16 * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
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

template <typename T, typename OutT, int NT>
int stem_resident_blocks() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)stem_kernel<T, OutT, NT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached[dev] = per_cu * cus;
    }
    return cached[dev];
}

template <typename T, typename OutT, int NT>
int launch_stem_nt(const StemP& p, const float* x, hipStream_t st) {
    unsigned grid = (unsigned)stem_resident_blocks<T, OutT, NT>();
    if (grid > p.ntiles) grid = p.ntiles;
    hipLaunchKernelGGL((stem_kernel<T, OutT, NT>), dim3(grid), dim3(256), 0, st, p, x);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

template <typename T, typename OutT>
int launch_stem(const StemP& p, const float* x, hipStream_t st) {
    switch ((p.Cout + 15) / 16) {
        case 1: return launch_stem_nt<T, OutT, 1>(p, x, st);
        case 2: return launch_stem_nt<T, OutT, 2>(p, x, st);
        case 3: return launch_stem_nt<T, OutT, 3>(p, x, st);
        case 4: return launch_stem_nt<T, OutT, 4>(p, x, st);
        default: return launch_stem_nt<T, OutT, 5>(p, x, st);
    }
}

}  // namespace

extern "C" int ccnet_stem_conv(const float* x_dev, int B, int H, int W, const float* weight, const float* bias, int Cout, int act,
                               ccnet_slice out, int out_dtype, ccnet_stream stream) {
    CC_REQUIRE(x_dev && weight && bias, "stem_conv: null pointer");
    CC_REQUIRE(B > 0 && H >= 2 && W >= 2 && (H % 2) == 0 && (W % 2) == 0, "stem_conv: H,W must be even (got %d,%d)", H, W);
    CC_REQUIRE((long long)B * 3 * H * W < (1ll << 31), "stem_conv: image batch of 2^31 elements or more");
    CC_REQUIRE(Cout > 0 && Cout <= 80 && (Cout % 8) == 0, "stem_conv: Cout must be a multiple of 8, at most 80 (got %d)", Cout);
    CC_REQUIRE(cc_dtype_ok(out_dtype) && cc_slice_vec_ok(out, out_dtype) && (out.ldc % 8) == 0 && (out.coff % 8) == 0 && out.C == Cout &&
                   out.shift == 0, "stem_conv: bad output slice");
    CC_REQUIRE(act >= CCNET_ACT_NONE && act <= CCNET_ACT_LRELU, "stem_conv: unknown activation");
    StemP p{x_dev, B, H, W, H / 2, W / 2, weight, bias, Cout, act, (unsigned char*)out.ptr, out.ldc, out.coff, 0, 0, 0};
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    p.ntiles = (unsigned)((long long)B * p.tiles_x * p.tiles_y);
    p.xcd = cc_xcd_remap();
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
