// One Bottleneck of a CSPLayer in ONE launch (bf16): y = [x +] SiLU(conv3x3(SiLU(conv1x1(x)))), both convs BaseConv with BN folded,
// C -> C -> C channels (network/model_block.py:193-218 with expansion 1.0, as CSPLayer builds it, :86-116), C = 64, 128 or 256.
//
// Why: at 20 x 20 ... 80 x 80 the pair is a 15 us 1x1 launch (3 GFLOP, latency-bound) whose output the 3x3 launch reads straight
// back.  Here the hidden map h never leaves the CU.  Per tile of 256 output pixels (conv_halo.hip's two geometries, 16 x 16 or 256
// linear padded positions for narrow maps; C = 256: 128 linear positions, maps up to 21 wide):
//   1. x of the tile and its halo (352 rows x C; 176 at C = 256) comes in by LDS-DMA as C/32 planes of 32 channels (64-byte rows,
//      conv_halo's row swizzle), all at once -- the planes hold the whole K of the 1x1;
//   2. phase 1: h = SiLU(W1 x + b1) for all rows: wave w owns the 16-row m-tiles w, w + 4, ..; accumulators in registers;
//   3. the waves overwrite the planes with h (bf16, ZERO where the row is outside the image: the 3x3 pads h, not x);
//   4. phase 2: the nine taps over the C/32 resident planes, wave tiles of 64 x 64 (C = 64) or 128 pixels x 64 channels, no
//      barrier inside; the next tap's pixel fragments are requested before the current tap's MFMAs;
//   5. the tile's shortcut rows and the next tile's x are requested, then the epilogue (bias, SiLU, + x, store) runs while they arrive.
// Weights never touch LDS: one stream per 64 output channels of the 3x3 in consumption order -- the 1x1's fragments (all of
// them: every wave needs every hidden channel for its rows), then (plane, tap, n-tile) of the 3x3 -- behind a register ring of 8
// (C = 64) or 16 fragments that runs ahead across phases and tiles (the stream ends with a copy of its head).
// LDS: C = 64 51 KB, two workgroups per CU; C >= 128 96 KB, one workgroup of four waves per CU (up to 512 registers per lane) --
// there every SiLU is exposed (47 % of a tile's time, -DBK_STAMP), which is why graph.py takes those shapes only while every tile
// finds a CU of its own.
// Same MFMA shape, k order, bias placement and rounding points as the two separate launches: results are bit-identical to
// ccnet_conv2d / ccnet_conv_pw followed by ccnet_conv_halo (tests/test_ops_gpu.py:test_bneck_fused_...).
// The output must not alias the input: a tile's halo is its neighbours' output pixels.
#include <stdlib.h>
#include <mutex>
#include "conv_common.h"

#ifndef BK_ABL
#define BK_ABL 0   // timing ablations (wrong results): 1 = no SiLU on h, 2 = no SiLU in the output epilogue, 4 = no ring refills, 8 = pixel fragments read once per tile, 16 = no x DMA for the second and later tiles
#endif

#ifdef BK_STAMP   // diagnostic build only: time (s_memtime ticks) block 0 spends per phase, summed over its tiles (scripts/bneck_bench.py)
__device__ unsigned long long g_bk_stamps[8];
#define BKSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_bk_stamps[i], n_ - st_last); st_last = n_; } } while (0)
#else
#define BKSTAMP(i)
#endif

namespace {

constexpr int FROWB = 64;                 // bytes per plane row (32 channels)
constexpr int FLINMAXWP = 47;
// LDS-DMA pieces (16 rows) per plane: 22 = 352 rows >= 18 * 18 (16 x 16 tiles) and >= 256 + 2 * 47 + 2 (256 linear positions).
// C = 256 (the 20 x 20 maps) works on tiles of 128 linear positions, 11 pieces = 176 rows >= 128 + 2 * 23 + 2, so that its eight
// planes are the same 88 KB: one workgroup per CU either way.
constexpr int bneck_pieces(int C) { return C == 256 ? 11 : 22; }
constexpr int bneck_tile_px(int C) { return C == 256 ? 128 : 256; }
constexpr int FLINMAXWP256 = 23;
// weight fragments in flight per wave.  C >= 128 runs one workgroup per CU (96 KB of LDS), i.e. one wave per SIMD with nobody to cover
// a late fragment, and has the registers: 16.  C = 64 runs two: 8.
constexpr int bneck_ring(int C) { return C >= 128 ? 16 : 8; }

struct BneckQ {
    const unsigned char* in_ptr;   // first element of the x slice
    unsigned in_ldcB, in_span;
    int H, W, B;
    int tiles_x, tiles_y;          // 2D
    int Wp, Hp;                    // LIN: padded width W + 2, rows per image H + 1
    unsigned magicWp, shiftWp, magicHp, shiftHp;
    unsigned ntiles;
    const unsigned char* wstream;  // [C / 64][(C / 32) * (C / 16) + (C / 32) * 36 + bneck_ring(C)][64 lanes][16 B]
    unsigned long long wstride;
    const float* bias1;
    const float* bias2;
    int shortcut;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int xcd;
};

// pixel fragments of tap TAP of the plane at PT into FA
#define BK_LOAD(FA, PT, TAP)                                                                                                       \
    {                                                                                                                              \
        constexpr int ky_ = (TAP) / 3, kx_ = (TAP) - ky_ * 3;                                                                      \
        if constexpr (LIN) {                                                                                                       \
            const int r0_ = lin_base + ky_ * lin_wp + kx_;                                                                         \
            const int a0_ = r0_ * FROWB + ((fq ^ ((r0_ >> 1) & 3)) << 4);                                                          \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) FA[i] = *(const u32x4*)((PT) + a0_ + i * MSTR);                         \
        } else {                                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                FA[i] = *(const u32x4*)((PT) + lanec[(2 * (i + ky_) + kx_) & 7] + ((i + ky_) * 18 + kx_) * FROWB);                 \
        }                                                                                                                          \
    }

// tap TAP of plane CC of the current group of planes (at pt): the NEXT tap's pixel fragments are requested first (one wave per
// SIMD at C >= 128: nothing else hides the LDS latency), then NT x MT MFMAs on the fragments requested a tap ago
#define BK_TAP(CC, TAP)                                                                                                            \
    {                                                                                                                              \
        constexpr int idx_ = (CC) * 9 + (TAP), cur_ = idx_ & 1, nidx_ = idx_ + 1;                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if constexpr (nidx_ < PG * 9 && !(BK_ABL & 8)) { BK_LOAD(fa[cur_ ^ 1], pt + (nidx_ / 9) * PLANE, nidx_ % 9) }              \
        else if constexpr (NG > 1 && !(BK_ABL & 8)) { BK_LOAD(fa[cur_ ^ 1], ptn, 0) }                                              \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                           \
            const int slot = ((CC) * 36 + (TAP) * NT + j) % FRING;                                                                 \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot]), __builtin_bit_cast(bf16x8, fa[(BK_ABL & 8) ? 0 : cur_][i]), acc[i][j], 0, 0, 0); \
            if (!(BK_ABL & 4)) ring[slot] = *(const u32x4*)(wp + FRING * 1024 + loff);                                             \
            wp += 1024;                                                                                                            \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    }

template <int C, bool LIN>
__global__ __launch_bounds__(256, (C == 64 ? 2 : 1)) void bneck_fused_kernel(const BneckQ p) {
    static_assert(C != 256 || LIN, "256 channels: linear tiles only");
    constexpr int NP = C / 32;             // planes
    constexpr int PIECES = bneck_pieces(C), PLANE = PIECES * 16 * FROWB, TPX = bneck_tile_px(C);
    constexpr int NPI = (PIECES + 3) / 4;  // pieces (and m-tiles of the 1x1) per wave: 22 over four waves = 6, 6, 5, 5; 11 = 3, 3, 3, 2
    constexpr int NT1 = C / 16;            // n-tiles of the 1x1: every wave computes all of them for its rows
    constexpr int MT1 = NPI;
    constexpr int MT = C == 64 ? 4 : 8, NT = 4;   // wave tile of the 3x3: 64 x 64 (C = 64), 128 x 64
    constexpr int PXW = MT * 16;
    constexpr int MSTR = LIN ? 16 * FROWB : 18 * FROWB;
    constexpr int F1 = NP * NT1;           // fragments of the 1x1 per tile
    constexpr int FRING = bneck_ring(C);
    constexpr int PG = NP < 4 ? NP : 4, NG = NP / PG;   // the 3x3 walks the planes in groups of PG (unrolled), NG groups
    static_assert(F1 % FRING == 0 && (PG * 36) % FRING == 0 && (PG * 9) % 2 == 0, "ring slots are compile-time constants");
    extern __shared__ __align__(16) unsigned char smem[];   // planes[NP] | pixtab | bias1 | bias2
    int* pixtab = (int*)(smem + NP * PLANE);
    float* b1s = (float*)(smem + NP * PLANE + NPI * 256 * 4);
    float* b2s = b1s + C;

    unsigned tile0, tstride, tend;
    cc_tile_walk(p.ntiles, p.xcd, tile0, tstride, tend);
    if (tile0 >= tend) return;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // waves over (pixels, 64-channel slices of the 3x3's output): 4 x 1 (C = 64), 2 x 2 (128), 1 x 4 (256)
    const int wm = C == 64 ? wave : C == 128 ? (wave >> 1) : 0, wn = C == 64 ? 0 : C == 128 ? (wave & 1) : wave;
    const int fr = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.in_ptr, p.in_span);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned ldcB = p.in_ldcB;

    for (int c = t; c < C; c += 256) { b1s[c] = p.bias1[c]; b2s[c] = p.bias2[c]; }

    // ---- geometry (conv_halo.hip)
    const unsigned prc = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) * 16);
    auto lin_pixel = [&](int q) -> int {
        if (q < 0) return -1;
        const int yp = (int)fastdiv((unsigned)q, p.magicWp, p.shiftWp), xp = q - yp * p.Wp;
        const int b = (int)fastdiv((unsigned)yp, p.magicHp, p.shiftHp), yy = yp - b * p.Hp;
        const bool real = b < p.B && yy >= 1 && xp >= 1 && xp <= p.W;
        return real ? (b * p.H + yy - 1) * p.W + xp - 1 : -1;
    };
    struct TileAt { int tile, b, y0, x0; bool valid; };   // wave uniform
    auto tile_at = [&](unsigned tile) -> TileAt {
        TileAt ta{(int)tile, 0, 0, 0, tile < tend};
        if constexpr (!LIN) {
            const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
            const unsigned b = tile / per_img, rem = tile - b * per_img;
            const int ty = (int)(rem / (unsigned)p.tiles_x);
            ta.b = (int)b; ta.y0 = ty * 16 - 1; ta.x0 = (int)(rem - (unsigned)ty * p.tiles_x) * 16 - 1;
        }
        return ta;
    };
    // pixel behind patch row r of a tile, -1 = outside the image (zero)
    auto patch_pixel = [&](const TileAt& ta, int r) -> int {
        if (!ta.valid) return -1;
        if constexpr (LIN) {
            return r < TPX + 2 * p.Wp + 2 ? lin_pixel(ta.tile * TPX - p.Wp - 1 + r) : -1;
        } else {
            const int py = r / 18, px = r - py * 18;
            const int vy = ta.y0 + py, vx = ta.x0 + px;
            const bool ok = r < 324 && (unsigned)vy < (unsigned)p.H && (unsigned)vx < (unsigned)p.W;
            return ok ? (ta.b * p.H + vy) * p.W + vx : -1;
        }
    };
    // a lane reads back only what it wrote itself: no barrier around the table
    auto fill_pixels = [&](const TileAt& ta) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int pc = wave + 4 * i;
            if (pc < PIECES) pixtab[i * 256 + t] = patch_pixel(ta, 16 * pc + (lane >> 2));
        }
    };
    auto issue_x = [&]() {
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int pc = wave + 4 * i;
            if (pc >= PIECES) continue;   // wave uniform
            const int px = pixtab[i * 256 + t];
            const unsigned v = px >= 0 ? (unsigned)px * ldcB + prc : OOB;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + pl * PLANE + pc * 1024), 16, (int)v, pl * 64, 0, 0);
        }
    };

    // ---- per-lane LDS offsets
    const int a1 = wave * 1024 + fr * FROWB + ((fq ^ ((fr >> 1) & 3)) << 4);                           // phase 1 reads: rows (wave + 4 m) * 16 + fr
    const int w1o = wave * 1024 + fr * FROWB + ((((fq >> 1)) ^ ((fr >> 1) & 3)) << 4) + (fq & 1) * 8;   // h writes: 4 channels = 8 bytes
    int lanec[8];
    const int lin_base = wm * PXW + fr;
    const int lin_wp = __builtin_amdgcn_readfirstlane(p.Wp);
    if constexpr (!LIN) {
#pragma unroll
        for (int m = 0; m < 8; ++m) lanec[m] = (wm * MT * 18 + fr) * FROWB + ((fq ^ (((m + fr) >> 1) & 3)) << 4);
    }

    // ---- weight ring
    const unsigned char* wp = p.wstream + (size_t)wn * p.wstride;   // wave uniform
    const int loff = lane * 16;
    u32x4 ring[FRING];
#pragma unroll
    for (int k = 0; k < FRING; ++k) ring[k] = *(const u32x4*)(wp + k * 1024 + loff);
    constexpr long long wrewind = (long long)(F1 + NP * 36) * 1024;

    const int nbase = wn * 64 + fq * 4;

    TileAt ta = tile_at(tile0);
    fill_pixels(ta);
    issue_x();
#ifdef BK_STAMP
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif

    for (unsigned tile = tile0; tile < tend; tile += tstride) {
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): the planes (and the bias table) have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {   // every tile starts at the head of the stream again: hidden from the optimiser, which otherwise hoists the ~200 64-bit
            // fragment addresses of a tile out of the tile loop and spills them
            unsigned long long wpo = (unsigned long long)wp;
            asm volatile("" : "+s"(wpo));
            wp = (const unsigned char*)wpo;
        }
        BKSTAMP(0);

        // ---- phase 1: acc1[m][n] = W1 x over the rows of m-tiles wave + 4 m
        {
            f32x4 acc1[MT1][NT1];
#pragma unroll
            for (int m = 0; m < MT1; ++m)
#pragma unroll
                for (int n = 0; n < NT1; ++n) acc1[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4 fx[2][MT1];
#pragma unroll
            for (int m = 0; m < MT1; ++m) fx[0][m] = *(const u32x4*)(smem + a1 + m * 4096);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                __builtin_amdgcn_sched_barrier(0);
                if (pl + 1 < NP) {
#pragma unroll
                    for (int m = 0; m < MT1; ++m) fx[(pl + 1) & 1][m] = *(const u32x4*)(smem + (pl + 1) * PLANE + a1 + m * 4096);
                }
#pragma unroll
                for (int n = 0; n < NT1; ++n) {
                    const int slot = (pl * NT1 + n) % FRING;
                    // (the waves that own one m-tile less multiply whatever lies behind the plane -- still inside this workgroup's
                    // LDS -- and drop it below: cheaper than a branch per fragment)
#pragma unroll
                    for (int m = 0; m < MT1; ++m)
                        acc1[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot]), __builtin_bit_cast(bf16x8, fx[pl & 1][m]), acc1[m][n], 0, 0, 0);
                    if (!(BK_ABL & 4)) ring[slot] = *(const u32x4*)(wp + FRING * 1024 + loff);
                    wp += 1024;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every wave's reads of x are done before h overwrites the planes
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            BKSTAMP(1);

            // ---- h = SiLU(acc1 + b1) as bf16 into the planes, zero outside the image
            float4 bv[NT1];
#pragma unroll
            for (int n = 0; n < NT1; ++n) bv[n] = *(const float4*)(b1s + n * 16 + fq * 4);
#pragma unroll
            for (int m = 0; m < MT1; ++m) {
                if (wave + 4 * m >= PIECES) break;   // wave uniform: the last waves own one m-tile less
                const bool real = patch_pixel(ta, (wave + 4 * m) * 16 + fr) >= 0;
#pragma unroll
                for (int n = 0; n < NT1; ++n) {
                    constexpr int A1 = (BK_ABL & 1) ? CCNET_ACT_NONE : CCNET_ACT_SILU;
                    float v0 = act_fn<A1>(acc1[m][n][0] + bv[n].x), v1 = act_fn<A1>(acc1[m][n][1] + bv[n].y);
                    float v2 = act_fn<A1>(acc1[m][n][2] + bv[n].z), v3 = act_fn<A1>(acc1[m][n][3] + bv[n].w);
                    uint2 h = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                    if (!real) h = make_uint2(0u, 0u);
                    *(uint2*)(smem + (n >> 1) * PLANE + m * 4096 + (w1o ^ ((n & 1) * 32))) = h;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        BKSTAMP(2);

        // ---- phase 2: nine taps over the resident planes of h
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            u32x4 fa[2][MT];
            BK_LOAD(fa[0], smem, 0)
            for (int g = 0; g < NG; ++g) {
                const unsigned char* pt = smem + g * PG * PLANE;
                const unsigned char* ptn = smem + (g + 1 < NG ? g + 1 : 0) * PG * PLANE;
                (void)ptn;
                BK_TAP(0, 0) BK_TAP(0, 1) BK_TAP(0, 2) BK_TAP(0, 3) BK_TAP(0, 4) BK_TAP(0, 5) BK_TAP(0, 6) BK_TAP(0, 7) BK_TAP(0, 8)
                BK_TAP(1, 0) BK_TAP(1, 1) BK_TAP(1, 2) BK_TAP(1, 3) BK_TAP(1, 4) BK_TAP(1, 5) BK_TAP(1, 6) BK_TAP(1, 7) BK_TAP(1, 8)
                if constexpr (PG == 4) {
                    BK_TAP(2, 0) BK_TAP(2, 1) BK_TAP(2, 2) BK_TAP(2, 3) BK_TAP(2, 4) BK_TAP(2, 5) BK_TAP(2, 6) BK_TAP(2, 7) BK_TAP(2, 8)
                    BK_TAP(3, 0) BK_TAP(3, 1) BK_TAP(3, 2) BK_TAP(3, 3) BK_TAP(3, 4) BK_TAP(3, 5) BK_TAP(3, 6) BK_TAP(3, 7) BK_TAP(3, 8)
                }
            }
        }
        wp -= wrewind;   // the stream ends with a copy of its first FRING fragments: the ring already holds the next tile's
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): every wave's reads of h are done before the next x overwrites the planes
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        BKSTAMP(3);

        // ---- this tile's shortcut rows are requested, then the next tile's x, then the epilogue runs while both arrive (vector
        // memory returns in order: requested after the x planes, the shortcut rows would hold the epilogue back behind them)
        {
            float4 bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = *(const float4*)(b2s + nbase + j * 16);
            unsigned b = 0;
            int ty = 0, tx = 0;
            if constexpr (!LIN) {
                const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
                b = tile / per_img;
                const unsigned rem = tile - b * per_img;
                ty = (int)(rem / (unsigned)p.tiles_x);
                tx = (int)(rem - (unsigned)ty * p.tiles_x);
            }
            int mpix[MT];
            uint2 rv[MT][NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (LIN) {
                    mpix[i] = lin_pixel((int)tile * TPX + wm * PXW + i * 16 + fr);
                } else {
                    const int oy = ty * 16 + wm * MT + i, ox = tx * 16 + fr;
                    mpix[i] = (oy < p.H && ox < p.W) ? ((int)b * p.H + oy) * p.W + ox : -1;
                }
                if (p.shortcut && mpix[i] >= 0) {
                    const unsigned char* rrow = p.in_ptr + (size_t)mpix[i] * ldcB + nbase * 2;
#pragma unroll
                    for (int j = 0; j < NT; ++j) rv[i][j] = *(const uint2*)(rrow + j * 32);
                } else {
#pragma unroll
                    for (int j = 0; j < NT; ++j) rv[i][j] = make_uint2(0u, 0u);
                }
            }
            const TileAt tn = tile_at(tile + tstride);
            if (tn.valid) {
                fill_pixels(tn);
                if (!(BK_ABL & 16)) issue_x();
            }
            ta = tn;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (mpix[i] < 0) continue;
                unsigned char* orow = p.out_ptr + ((size_t)mpix[i] * p.out_ldc + p.out_coff + nbase) * 2;
                float v[NT][4];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    constexpr int A2 = (BK_ABL & 2) ? CCNET_ACT_NONE : CCNET_ACT_SILU;
                    v[j][0] = act_fn<A2>(acc[i][j][0] + bv[j].x); v[j][1] = act_fn<A2>(acc[i][j][1] + bv[j].y);
                    v[j][2] = act_fn<A2>(acc[i][j][2] + bv[j].z); v[j][3] = act_fn<A2>(acc[i][j][3] + bv[j].w);
                }
                if (p.shortcut) {   // added after the activation (model_block.py:213-218)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        v[j][0] += bf16_bits_to_f32(rv[i][j].x & 0xFFFFu); v[j][1] += bf16_bits_to_f32(rv[i][j].x >> 16);
                        v[j][2] += bf16_bits_to_f32(rv[i][j].y & 0xFFFFu); v[j][3] += bf16_bits_to_f32(rv[i][j].y >> 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) *(uint2*)(orow + j * 32) = make_uint2(pack_bf16x2(v[j][0], v[j][1]), pack_bf16x2(v[j][2], v[j][3]));
            }
        }
        BKSTAMP(4);
    }
}

template <int C, bool LIN>
int launch_bneck(const BneckQ& p, hipStream_t st) {
    auto kernel = bneck_fused_kernel<C, LIN>;
    constexpr int LDS = (C / 32) * bneck_pieces(C) * 16 * FROWB + ((bneck_pieces(C) + 3) / 4) * 256 * 4 + 2 * C * 4;
    static std::mutex mu;
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int resident = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (cached[dev] == 0) {
            CC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, LDS) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            cached[dev] = per_cu * cus;
        }
        resident = cached[dev];
    }
    const int gx = cc_grid_x(resident, 1, p.ntiles);
    hipLaunchKernelGGL(kernel, dim3(gx), dim3(256), LDS, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

void bneck_magic(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *shift = s;
    *magic = (unsigned)((((1ull << s) - d) << 32) / d + 1);
}

}  // namespace

#ifdef BK_STAMP
extern "C" int ccnet_debug_bneck_stamps(unsigned long long* out8, int reset) {
    CC_HIP(hipDeviceSynchronize());
    CC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bk_stamps), sizeof(unsigned long long) * 8));
    if (reset) {
        unsigned long long z[8] = {0};
        CC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bk_stamps), z, sizeof(z)));
    }
    return CCNET_OK;
}
#endif

extern "C" size_t ccnet_bneck_fused_stream_elems(int C) {
    if (C != 64 && C != 128 && C != 256) return 0;
    return (size_t)(C / 64) * ((size_t)(C / 32) * (C / 16) + (size_t)(C / 32) * 36 + bneck_ring(C)) * 512;
}

extern "C" int ccnet_bneck_fused(ccnet_slice in, int B, int H, int W, const void* wstream_dev, const float* bias1_dev, const float* bias2_dev,
                                 int shortcut, int act, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(wstream_dev && bias1_dev && bias2_dev, "bneck_fused: null pointer");
    CC_REQUIRE(B > 0 && H > 0 && W > 0, "bneck_fused: bad sizes (B=%d H=%d W=%d)", B, H, W);
    CC_REQUIRE(act == CCNET_ACT_SILU, "bneck_fused: SiLU only (act=%d)", act);
    const int Cc = in.C;
    CC_REQUIRE(Cc == 64 || Cc == 128 || Cc == 256, "bneck_fused: 64, 128 or 256 channels (C=%d)", Cc);
    CC_REQUIRE(Cc != 256 || W + 2 <= FLINMAXWP256, "bneck_fused: 256 channels on maps up to %d wide only (W=%d)", FLINMAXWP256 - 2, W);
    CC_REQUIRE(cc_slice_vec_ok(in, CCNET_BF16) && in.shift == 0, "bneck_fused: input slice must be bf16, un-shifted, 16-byte aligned (C=%d ldc=%d coff=%d)",
               in.C, in.ldc, in.coff);
    CC_REQUIRE(((uintptr_t)wstream_dev % 16) == 0 && ((uintptr_t)bias1_dev % 16) == 0 && ((uintptr_t)bias2_dev % 16) == 0,
               "bneck_fused: weights / biases must be 16-byte aligned");
    CC_REQUIRE(out.ptr && out.shift == 0 && out.C == Cc && out.coff >= 0 && out.ldc >= out.coff + out.C && (out.ldc % 4) == 0 && (out.coff % 4) == 0 &&
                   ((uintptr_t)out.ptr % 8) == 0, "bneck_fused: bad output slice (C=%d ldc=%d coff=%d)", out.C, out.ldc, out.coff);
    const double span = (((double)B * H * W - 1.0) * in.ldc + in.C) * 2.0;
    CC_REQUIRE(span < 2147483648.0, "bneck_fused: input tensor of 2 GiB or more");
    {   // a tile's halo is its neighbours' output: the output slice must not overlap the input slice
        const unsigned char* i0 = (const unsigned char*)in.ptr;
        const unsigned char* i1 = i0 + (size_t)B * H * W * in.ldc * 2;
        const unsigned char* o0 = (const unsigned char*)out.ptr;
        const unsigned char* o1 = o0 + (size_t)B * H * W * out.ldc * 2;
        const bool same_rows = o0 < i1 && i0 < o1;
        const bool chan_overlap = in.ptr == out.ptr && in.ldc == out.ldc ? (out.coff < in.coff + in.C && in.coff < out.coff + out.C) : true;
        CC_REQUIRE(!(same_rows && chan_overlap), "bneck_fused: the output slice overlaps the input slice");
    }
    BneckQ p{};
    p.in_ptr = (const unsigned char*)in.ptr + (size_t)in.coff * 2;
    p.in_ldcB = (unsigned)(in.ldc * 2);
    p.in_span = (unsigned)span;
    p.H = H; p.W = W; p.B = B;
    const long long M = (long long)B * H * W;
    CC_REQUIRE(M < (1ll << 30), "bneck_fused: too many pixels");
    const int tx = (W + 15) / 16, ty = (H + 15) / 16;
    const double waste2d = (double)tx * ty * 256.0 / ((double)H * W);
    const double wastelin = (double)(W + 2) * (H + 1) / ((double)H * W);
    const bool lin = Cc == 256 || (W + 2 <= FLINMAXWP && wastelin < waste2d);
    if (lin) {
        p.Wp = W + 2; p.Hp = H + 1;
        const long long q = ((long long)B * p.Hp + 1) * p.Wp;
        CC_REQUIRE(q < (1ll << 30), "bneck_fused: too many padded positions");
        const int tpx = bneck_tile_px(Cc);
        p.ntiles = (unsigned)((q + tpx - 1) / tpx);
        bneck_magic((unsigned)p.Wp, &p.magicWp, &p.shiftWp);
        bneck_magic((unsigned)p.Hp, &p.magicHp, &p.shiftHp);
    } else {
        p.tiles_x = tx; p.tiles_y = ty;
        p.ntiles = (unsigned)((long long)B * tx * ty);
        p.Wp = 18;
    }
    p.wstream = (const unsigned char*)wstream_dev;
    p.wstride = ((unsigned long long)(Cc / 32) * (Cc / 16) + (unsigned long long)(Cc / 32) * 36 + bneck_ring(Cc)) * 1024ull;
    p.bias1 = bias1_dev; p.bias2 = bias2_dev;
    p.shortcut = shortcut ? 1 : 0;
    p.out_ptr = (unsigned char*)out.ptr; p.out_ldc = out.ldc; p.out_coff = out.coff;
    p.xcd = cc_xcd_remap();
    const double flops = 2.0 * (double)M * Cc * (10.0 * Cc);
    const double bytes = (double)M * Cc * 4.0 + 10.0 * Cc * Cc * 2.0;
    return ccnet_submit(
        [p, lin, Cc](hipStream_t st, const float*) -> int {
            if (Cc == 256) return launch_bneck<256, true>(p, st);
            if (Cc == 128) return lin ? launch_bneck<128, true>(p, st) : launch_bneck<128, false>(p, st);
            return lin ? launch_bneck<64, true>(p, st) : launch_bneck<64, false>(p, st);
        },
        stream, CCNET_OP_CONV, flops, bytes);
}
