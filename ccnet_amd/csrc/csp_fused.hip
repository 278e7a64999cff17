// CSPLayer with ONE bottleneck in one launch (bf16; network/model_block.py:86-116 CSPLayer.forward, :193-218 Bottleneck.forward):
//
//     x_1 = conv1(x);  x_2 = conv2(x);  x_1 = [x_1 +] conv_b2(conv_b1(x_1));  out = conv3(cat(x_1, x_2))
//
// with every conv a BaseConv (BN folded, SiLU).  As separate launches the two big-map layers of this shape (dark2 at 160^2,
// dark3 / C3_p3 at 80^2 for a 640^2 input) are five HBM-bound passes over tensors of 100-400 MB; here a workgroup owns an 8x16
// tile of output pixels, fetches the (8+2)x(16+2) halo patch of x once and keeps x_1, x_2, the bottleneck's hidden map and the
// updated x_1 in LDS: HBM sees x once and out once.
//
// Stages per tile (256 threads = 4 waves; all GEMMs on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand, so a
// lane's four accumulators are four consecutive channels of one pixel and every epilogue is register-only):
//   1  [x_1 | x_2] = SiLU(W12 x + b)    x_1 on all 180 patch pixels (12 m-tiles of 16 consecutive patch rows), x_2 on the 128
//                                       inner pixels; K = Cin in chunks of 64 channels staged through LDS (XB)
//   2  t = SiLU(Wb1 x_1 + b)            on the patch, written over XB; ZERO outside the image (conv_b2 pads t, not x)
//   3  x_1' = SiLU(Wb2 * t + b) [+ x_1] nine taps = the same LDS image read at shifted rows; written in place over x_1
//   4  out = SiLU(W3 [x_1' | x_2] + b)  -> global
// bf16 rounding points are those of the separate launches (every intermediate is rounded when it is stored).
// Weights: MFMA fragments ([n-tile][k-step][64 lanes][8]) packed by ccnet_amd/ops.py:pack_csp, read from L2 by the wave that
// owns the n-tile, one k-step ahead of the MFMAs.
#include <mutex>
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef CSP_WAVES
#define CSP_WAVES 4
#endif
constexpr int TH = 8, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW, MROWS = 192;  // 180 patch pixels, 12 m-tiles
constexpr int XB_BYTES = MROWS * 128;                                            // one 64-channel chunk of the patch

struct CspSeg {
    const unsigned char* ptr;  // first channel of the slice at (b=0, y=0, x=0)
    unsigned ld;               // pixel stride in bytes
    int C, shift, Hs, Ws;      // channels; logical (y, x) is stored (y >> shift, x >> shift) of an Hs x Ws map
};
struct CspP {
    CspSeg seg[2];
    int nseg, cin;
    unsigned char* out;
    unsigned ldo;  // bytes
    const u32x4* w;
    const float* bias;
    int B, H, W, tiles_x, tiles_y;
    unsigned ntiles;
    int xcd;  // tile order, see cc_tile_walk
};

// 16-byte chunk c of row r.  RB = 128: eight chunks per row; RB = 64: four (two rows share a 128-byte line)
template <int RB>
__device__ __forceinline__ int lds_off(int r, int c) {
    if constexpr (RB == 128) return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    else return r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}

__device__ __forceinline__ f32x4 mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Where a wave finds its weight fragments: fragment i of the packed blob is 64 lanes x 16 bytes, in global memory (L2-resident)
// or, when the whole blob fits beside the activations, in LDS (staged once per persistent workgroup).
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <bool WLDS>
struct WSrc {
    const u32x4* g;
    const unsigned char* l;
    __device__ __forceinline__ u32x4 ld(int frag, int lane) const {
        if constexpr (WLDS) return *(const u32x4*)(l + ((size_t)frag * 64 + lane) * 16);
        else return g[(size_t)frag * 64 + lane];
    }
};

// ring[d][n] = fragment (n-tile n, k-step d) for d < min(D, NKS); a fragment (n, k) of a matrix is blob fragment f0 + n * KS + k
template <int NT, int NKS, int D, bool WLDS>
__device__ __forceinline__ void preload(u32x4 (&ring)[D][NT], const WSrc<WLDS>& w, int f0, int KS, int lane) {
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int n = 0; n < NT; ++n)
            if (d < NKS) ring[d][n] = w.ld(f0 + n * KS + d, lane);
}

// acc[m][n] += W[n][k-steps 0 .. NKS) x B(m, ks), fully unrolled; the ring holds k-steps 0..D-1 on entry and is refilled D steps
// ahead.  bsrc(m, ks) = address of the lane's 16 bytes of the pixel operand.
template <int MT, int NT, int NKS, int D, bool WLDS, class BSrc>
__device__ __forceinline__ void mm(f32x4 (*acc)[NT], u32x4 (&ring)[D][NT], const WSrc<WLDS>& w, int f0, int KS, int lane, BSrc bsrc) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const u32x4 bq = *(const u32x4*)bsrc(m, ks);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = mfma(ring[ks % D][n], bq, acc[m][n]);
        }
        if (ks + D < NKS) {
#pragma unroll
            for (int n = 0; n < NT; ++n) ring[ks % D][n] = w.ld(f0 + n * KS + ks + D, lane);
        }
    }
}

#ifdef CSP_NOACT  // timing experiment: what the SiLU epilogues cost
#define CSP_ACT(x) (x)
#else
#define CSP_ACT(x) act_fn<CCNET_ACT_SILU>(x)
#endif
__device__ __forceinline__ uint2 silu4(const f32x4& v, const float4& b) {
    const float x0 = v[0] + b.x, x1 = v[1] + b.y, x2 = v[2] + b.z, x3 = v[3] + b.w;
    return make_uint2(pack_bf16x2(CSP_ACT(x0), CSP_ACT(x1)), pack_bf16x2(CSP_ACT(x2), CSP_ACT(x3)));
}

#ifdef CSP_STAMP  // diagnostic build: cycles per phase of wave 0 of block 0 (0 load, 1 mm1, 2 epi1, 3 stage 2, 4 stage 3, 5 stage 4)
__device__ unsigned long long g_csp_stamps[8];
#define CSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_csp_stamps[i], n_ - st_last); st_last = n_; } } while (0)
#else
#define CSTAMP(i)
#endif


// H16 = hidden / 16 (2 or 4).  LDS: XB | P1 (x_1 patch, later x_1' on its inner rows) | P2 (x_2, inner pixels) [| weights]
// NWV waves per workgroup (4 or 8); every stage splits its (m-tile, n-tile) grid over them, n first.
template <int H16, bool SHORTCUT, bool WLDS, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 2) void csp_fused_kernel(const CspP p) {
    constexpr int HID = 16 * H16, RB = 2 * HID, KH = HID / 32;  // bytes per row of the hidden-width images; k-steps of a hidden-wide K
    constexpr int NTH = NWV * 64;
    constexpr int NLD = (PROWS * 8 + NTH - 1) / NTH;            // 16-byte pieces of one 64-channel patch chunk per thread
    // stage 1: the 2*H16 n-tiles of [x_1 | x_2] over the waves; WPN waves share an n-tile and split its m-tiles
    constexpr int N1 = 2 * H16, NTW = N1 >= NWV ? N1 / NWV : 1, WPN = N1 >= NWV ? 1 : NWV / N1, M1A = 12 / WPN, M1B = 8 / WPN;
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char* XB = lds;
    unsigned char* P1 = lds + XB_BYTES;
    unsigned char* P2 = P1 + MROWS * RB;
    unsigned char* WL = P2 + 128 * RB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, px16_0 = lane & 15, kg_0 = lane >> 4;
    const int KS1 = p.cin / 32;
    // first fragment of the four weight matrices in the blob, and their biases
    const int f12 = 0, fb1 = 2 * H16 * KS1, fb2 = fb1 + H16 * KH, f3 = fb2 + H16 * 9 * KH;
    const float* b12 = p.bias;
    const float* bb1 = b12 + 2 * HID;
    const float* bb2 = bb1 + HID;
    const float* b3 = bb2 + HID;
    const WSrc<WLDS> w{p.w, WL};
    if constexpr (WLDS) {
        const int nfr16 = (f3 + 2 * H16 * 2 * KH) * 64;
        for (int i = t; i < nfr16; i += NTH) ((u32x4*)WL)[i] = p.w[i];
        __syncthreads();  // the first fragments are read before the tile loop's first barrier
    }

    // the patch chunk (kc) of a tile: global -> registers (xload), registers -> XB (xstore)
    u32x4 xv[NLD];
    auto xload = [&](unsigned tile, int kc) {
        const int tx = (int)(tile % (unsigned)p.tiles_x);
        const int rest = (int)(tile / (unsigned)p.tiles_x);
        const int ty = rest % p.tiles_y, b = rest / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        const int ch0 = kc * 64;
        const int sgi = (p.nseg == 2 && ch0 >= p.seg[0].C) ? 1 : 0;
        const CspSeg& sg = p.seg[sgi];
        const int cseg = ch0 - (sgi ? p.seg[0].C : 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = t + i * NTH;
            const int r = idx >> 3, c = idx & 7;
            const int py = (r * 3641) >> 16, pxx = r - py * PW;  // r / 18 for r < 180
            const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
            xv[i] = u32x4{0u, 0u, 0u, 0u};
            if (idx < PROWS * 8 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                xv[i] = *(const u32x4*)(sg.ptr + ((size_t)((size_t)b * sg.Hs + (gy >> sg.shift)) * sg.Ws + (gx >> sg.shift)) * sg.ld +
                                        (size_t)(cseg + c * 8) * 2);
        }
    };
    auto xstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = t + i * NTH;
            if (idx < PROWS * 8) *(u32x4*)(XB + lds_off<128>(idx >> 3, idx & 7)) = xv[i];
        }
    };

#ifdef CSP_STAMP
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const int nchunks = p.cin / 64;
    const int nt0 = (wave / WPN) * NTW;  // stage 1: first n-tile of the wave among the 2*H16 of [x_1 | x_2]
    const int mp1 = wave % WPN;          // and its share of that n-tile's m-tiles
    const bool isx1 = nt0 < H16;         // x_1 on the whole patch (12 m-tiles), x_2 on the inner pixels (8)
    unsigned tile0, tstride, tend;
    cc_tile_walk(p.ntiles, p.xcd, tile0, tstride, tend);
    if (tile0 < tend) xload(tile0, 0);
    for (unsigned tile = tile0; tile < tend; tile += tstride) {
        const int tx = (int)(tile % (unsigned)p.tiles_x);
        const int rest = (int)(tile / (unsigned)p.tiles_x);
        const int ty = rest % p.tiles_y, b = rest / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
        // The swizzled LDS addresses below are the same for every tile; left alone, the compiler hoists dozens of them out of
        // the tile loop and spills them.  Laundered lane terms make it recompute them (a few VALU instructions per use).
        // (hidden 64 only: hidden 32 has the registers to keep them)
        const int px16 = H16 == 4 ? opaque(px16_0) : px16_0, kg = H16 == 4 ? opaque(kg_0) : kg_0;

        // ---- stage 1: [x_1 | x_2] over K = cin, 64 channels at a time through XB.  Chunk 0 was fetched during the previous
        // tile's stage 4 (XB has been free since that tile's stage 3 ended).
        f32x4 acc1[M1A][NTW];
#pragma unroll
        for (int m = 0; m < M1A; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc1[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nchunks; ++kc) {
            u32x4 r1[2][NTW];
            preload<NTW, 2, 2>(r1, w, f12 + nt0 * KS1 + kc * 2, KS1, lane);  // in flight across the barriers below
            if (kc > 0) {
                __syncthreads();  // the previous chunk's reads of XB are done
                xload(tile, kc);
            }
            xstore();
            __syncthreads();
            CSTAMP(0);
            if (isx1)
                mm<M1A, NTW, 2, 2>(acc1, r1, w, f12 + nt0 * KS1 + kc * 2, KS1, lane,
                                   [&](int m, int ks) { return XB + lds_off<128>((mp1 * M1A + m) * 16 + px16, ks * 4 + kg); });
            else
                mm<M1B, NTW, 2, 2>(acc1, r1, w, f12 + nt0 * KS1 + kc * 2, KS1, lane,
                                   [&](int m, int ks) { return XB + lds_off<128>((mp1 * M1B + m + 1) * PW + 1 + px16, ks * 4 + kg); });
            CSTAMP(1);
        }
        // stage 2's fragments travel under epilogue 1
        constexpr int NSPLIT2 = H16 < NWV ? H16 : NWV, MT2 = 12 / (NWV / NSPLIT2);
        const int nt2 = wave % NSPLIT2, m02 = (wave / NSPLIT2) * MT2;
        u32x4 r2[KH][1];
        preload<1, KH, KH>(r2, w, fb1 + nt2 * KH, KH, lane);
        // epilogue 1: x_1 -> P1 (row = patch row), x_2 -> P2 (row = inner pixel)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int ch = (nt0 + n) * 16 + kg * 4;  // channel within [x_1 | x_2]
            const float4 bv = *(const float4*)(b12 + ch);
            if (isx1) {
#pragma unroll
                for (int m = 0; m < M1A; ++m)
                    *(uint2*)(P1 + lds_off<RB>((mp1 * M1A + m) * 16 + px16, ch >> 3) + (ch & 7) * 2) = silu4(acc1[m][n], bv);
            } else {
                const int c2 = ch - HID;
#pragma unroll
                for (int m = 0; m < M1B; ++m)
                    *(uint2*)(P2 + lds_off<RB>((mp1 * M1B + m) * 16 + px16, c2 >> 3) + (c2 & 7) * 2) = silu4(acc1[m][n], bv);
            }
        }
        __syncthreads();
        CSTAMP(2);

        // ---- stage 2: t = SiLU(Wb1 x_1 + b) on the patch -> T (= XB), zero outside the image
        unsigned char* T = XB;
        constexpr int NSPLIT3 = H16 / 2, MT3 = 8 / (NWV / NSPLIT3), NKS3 = 9 * KH, D3 = 4;
        const int nh3 = wave % NSPLIT3, m03 = (wave / NSPLIT3) * MT3;
        const int fw3 = fb2 + nh3 * 2 * NKS3;
        u32x4 r3[D3][2];
        {
            f32x4 acc[MT2][1];
#pragma unroll
            for (int m = 0; m < MT2; ++m) acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            mm<MT2, 1, KH, KH>(acc, r2, w, fb1 + nt2 * KH, KH, lane,
                               [&](int m, int ks) { return P1 + lds_off<RB>((m02 + m) * 16 + px16, ks * 4 + kg); });
            preload<2, NKS3, D3>(r3, w, fw3, NKS3, lane);  // stage 3's first fragments travel under epilogue 2
            const int ch = nt2 * 16 + kg * 4;
            const float4 bv = *(const float4*)(bb1 + ch);
#pragma unroll
            for (int m = 0; m < MT2; ++m) {
                const int r = (m02 + m) * 16 + px16;
                const int py = (r * 3641) >> 16, pxx = r - py * PW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
                const bool inside = r < PROWS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const uint2 v = silu4(acc[m][0], bv);
                *(uint2*)(T + lds_off<RB>(r, ch >> 3) + (ch & 7) * 2) = inside ? v : make_uint2(0u, 0u);
            }
        }
        __syncthreads();
        CSTAMP(3);

        // ---- stage 3: x_1' = SiLU(Wb2 * t + b) [+ x_1], 3x3 taps as shifted rows of T; in place over P1's inner rows
        constexpr int NT4 = H16, MT4 = 8 / (NWV / 2), NKS4 = 2 * KH, D4 = NKS4 < 2 ? NKS4 : 2;  // stage 4: 2*H16 n-tiles in two halves
        const int nh4 = wave & 1, m04 = (wave >> 1) * MT4;
        const int fw4 = f3 + nh4 * NT4 * NKS4;
        u32x4 r4[D4][NT4];
        {
            f32x4 acc[MT3][2];
#pragma unroll
            for (int m = 0; m < MT3; ++m) acc[m][0] = acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            // k loop kept rolled in blocks of D3 (unrolled, the scheduler hoists every tap's loads and spills)
#pragma unroll 1
            for (int ks0 = 0; ks0 < NKS3; ks0 += D3) {
#pragma unroll
                for (int u = 0; u < D3; ++u) {
                    const int ks = ks0 + u;
                    if (ks < NKS3) {
                        const int tap = ks / KH, j = ks - tap * KH;
                        const int dy = (tap * 11) >> 5, dx = tap - dy * 3;  // tap / 3 for tap < 9
#pragma unroll
                        for (int m = 0; m < MT3; ++m) {
                            const u32x4 bq = *(const u32x4*)(T + lds_off<RB>((m03 + m + dy) * PW + dx + px16, j * 4 + kg));
                            acc[m][0] = mfma(r3[u][0], bq, acc[m][0]);
                            acc[m][1] = mfma(r3[u][1], bq, acc[m][1]);
                        }
                        if (ks + D3 < NKS3) {
                            r3[u][0] = w.ld(fw3 + ks + D3, lane);
                            r3[u][1] = w.ld(fw3 + NKS3 + ks + D3, lane);
                        }
                    }
                }
            }
            preload<NT4, NKS4, D4>(r4, w, fw4, NKS4, lane);  // stage 4's first fragments travel under epilogue 3
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int ch = (nh3 * 2 + n) * 16 + kg * 4;
                const float4 bv = *(const float4*)(bb2 + ch);
#pragma unroll
                for (int m = 0; m < MT3; ++m) {
                    unsigned char* q = P1 + lds_off<RB>((m03 + m + 1) * PW + 1 + px16, ch >> 3) + (ch & 7) * 2;
                    const f32x4 a = acc[m][n];
                    float v0 = CSP_ACT(a[0] + bv.x), v1 = CSP_ACT(a[1] + bv.y);
                    float v2 = CSP_ACT(a[2] + bv.z), v3 = CSP_ACT(a[3] + bv.w);
                    if constexpr (SHORTCUT) {  // the residual is added after the activation (model_block.py:213-218)
                        const uint2 o = *(const uint2*)q;
                        v0 += bf16_bits_to_f32(o.x & 0xFFFFu); v1 += bf16_bits_to_f32(o.x >> 16);
                        v2 += bf16_bits_to_f32(o.y & 0xFFFFu); v3 += bf16_bits_to_f32(o.y >> 16);
                    }
                    *(uint2*)q = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
            }
        }
        __syncthreads();
        CSTAMP(4);

        // ---- stage 4: out = SiLU(W3 [x_1' | x_2] + b) -> global; the next tile's first patch chunk is fetched underneath
        {
            const unsigned next = tile + tstride;
            if (next < tend) xload(next, 0);
            f32x4 acc[MT4][NT4];
#pragma unroll
            for (int m = 0; m < MT4; ++m)
#pragma unroll
                for (int n = 0; n < NT4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            mm<MT4, NT4, NKS4, D4>(acc, r4, w, fw4, NKS4, lane, [&](int m, int ks) {
                return ks < KH ? P1 + lds_off<RB>((m04 + m + 1) * PW + 1 + px16, ks * 4 + kg)
                               : P2 + lds_off<RB>((m04 + m) * 16 + px16, (ks - KH) * 4 + kg);
            });
#pragma unroll
            for (int n = 0; n < NT4; ++n) {
                const int ch = (nh4 * NT4 + n) * 16 + kg * 4;
                const float4 bv = *(const float4*)(b3 + ch);
#pragma unroll
                for (int m = 0; m < MT4; ++m) {
                    const int gy = y0 + m04 + m, gx = x0 + px16;
                    *(uint2*)(p.out + ((size_t)((size_t)b * p.H + gy) * p.W + gx) * p.ldo + (size_t)ch * 2) = silu4(acc[m][n], bv);
                }
            }
        }
        CSTAMP(5);
    }
}

template <int H16, bool SHORTCUT, bool WLDS, int NWV = CSP_WAVES>
int launch_csp(const CspP& p, hipStream_t st, int wbytes) {
    auto kernel = csp_fused_kernel<H16, SHORTCUT, WLDS, NWV>;
    const int LDS = XB_BYTES + MROWS * 32 * H16 + 128 * 32 * H16 + (WLDS ? wbytes : 0);
    // The function's dynamic-LDS limit is set ONCE per device to the most this instantiation can ever ask for (WLDS layers differ in
    // cin and so in LDS bytes: setting it per launch would let two host threads interleave set(small) / set(big) / launch(big)).
    // The resident-workgroup count is cached per LDS size, under a lock.
    constexpr int LDS_CAP = WLDS ? 80 * 1024 : XB_BYTES + MROWS * 32 * H16 + 128 * 32 * H16;
    if (LDS > LDS_CAP) return ccnet_set_error(CCNET_EINVAL, "csp_fused: %d bytes of LDS exceed the %d this kernel is set up for", LDS, LDS_CAP);
    struct Entry { int lds, grid; };
    static std::mutex mu;
    static bool attr_set[16] = {false};
    static Entry cache[16][4] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int grid = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            CC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CAP));
            attr_set[dev] = true;
        }
        int slot = 0;
        for (; slot < 4 && cache[dev][slot].lds != 0 && cache[dev][slot].lds != LDS; ++slot) {}
        if (slot == 4) slot = 0;   // more than four distinct cin per device: recompute into the first slot
        if (cache[dev][slot].lds != LDS) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, NWV * 64, LDS) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            cache[dev][slot] = Entry{LDS, per_cu * cus};
        }
        grid = cache[dev][slot].grid;
    }
    if ((unsigned)grid > p.ntiles) grid = (int)p.ntiles;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NWV * 64), LDS, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

}  // namespace

#ifdef CSP_STAMP
extern "C" int ccnet_debug_csp_stamps(unsigned long long* out8) {
    CC_HIP(hipDeviceSynchronize());
    CC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_csp_stamps), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0};
    CC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_csp_stamps), z, sizeof(z)));
    return CCNET_OK;
}
#endif

extern "C" size_t ccnet_csp_fused_weight_elems(int cin, int hidden) {
    if (cin <= 0 || hidden <= 0) return 0;
    return (size_t)2 * hidden * cin + (size_t)hidden * hidden + (size_t)hidden * 9 * hidden + (size_t)2 * hidden * 2 * hidden;
}

extern "C" int ccnet_csp_fused(const ccnet_slice* in, int nseg, int B, int H, int W, int hidden, int shortcut, int act, const void* weights_dev,
                               const float* bias_dev, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(in && (nseg == 1 || nseg == 2) && weights_dev && bias_dev, "csp_fused: null pointer / bad segment count");
    CC_REQUIRE(hidden == 32 || hidden == 64, "csp_fused: hidden width %d not built (32 or 64)", hidden);
    CC_REQUIRE(act == CCNET_ACT_SILU, "csp_fused: only SiLU is built");
    CC_REQUIRE(B > 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0, "csp_fused: map %dx%d is not a multiple of the %dx%d tile", H, W, TH, TW);
    CC_REQUIRE((long long)B * H * W < (1ll << 31), "csp_fused: too many pixels");
    CC_REQUIRE(cc_slice_vec_ok(out, CCNET_BF16) && out.C == 2 * hidden && out.shift == 0, "csp_fused: bad output slice");
    CC_REQUIRE(((uintptr_t)weights_dev % 16) == 0 && ((uintptr_t)bias_dev % 16) == 0, "csp_fused: weights / bias must be 16-byte aligned");
    CspP p{};
    int cin = 0;
    for (int s = 0; s < nseg; ++s) {
        CC_REQUIRE(cc_slice_vec_ok(in[s], CCNET_BF16) && (in[s].C % 64) == 0, "csp_fused: input slice %d must be bf16 with a multiple of 64 channels", s);
        const int sh = in[s].shift;
        CC_REQUIRE((H % (1 << sh)) == 0 && (W % (1 << sh)) == 0, "csp_fused: upsampled input %d does not divide the map", s);
        p.seg[s] = CspSeg{(const unsigned char*)in[s].ptr + (size_t)in[s].coff * 2, (unsigned)in[s].ldc * 2u, in[s].C, sh, H >> sh, W >> sh};
        cin += in[s].C;
    }
    if (nseg == 1) p.seg[1] = p.seg[0];
    p.nseg = nseg;
    p.cin = cin;
    p.out = (unsigned char*)out.ptr + (size_t)out.coff * 2;
    p.ldo = (unsigned)out.ldc * 2u;
    p.w = (const u32x4*)weights_dev;
    p.bias = bias_dev;
    p.B = B; p.H = H; p.W = W;
    p.tiles_x = W / TW; p.tiles_y = H / TH;
    p.ntiles = (unsigned)((long long)B * p.tiles_x * p.tiles_y);
    p.xcd = cc_xcd_remap();
    const double px = (double)B * H * W;
    const double flops = 2.0 * px * ((double)cin * 2 * hidden + (double)hidden * hidden + 9.0 * hidden * hidden + 4.0 * hidden * hidden);
    double bytes = px * 2.0 * hidden * 2.0 + 2.0 * (double)ccnet_csp_fused_weight_elems(cin, hidden);
    for (int s = 0; s < nseg; ++s) bytes += px / (double)(1 << (2 * in[s].shift)) * in[s].C * 2.0;
    const bool sc = shortcut != 0;
    // hidden 32 with 64 input channels: the whole weight blob (36 KB) fits beside the activations with two workgroups per CU
    const int wbytes = (int)(2 * ccnet_csp_fused_weight_elems(cin, hidden));
    const bool wlds = hidden == 32 && XB_BYTES + MROWS * 64 + 128 * 64 + wbytes <= 80 * 1024;
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (hidden == 64) return sc ? launch_csp<4, true, false>(p, st, wbytes) : launch_csp<4, false, false>(p, st, wbytes);
            if (wlds) return sc ? launch_csp<2, true, true>(p, st, wbytes) : launch_csp<2, false, true>(p, st, wbytes);
            return sc ? launch_csp<2, true, false>(p, st, wbytes) : launch_csp<2, false, false>(p, st, wbytes);
        },
        stream, CCNET_OP_CSP, flops, bytes);
}
