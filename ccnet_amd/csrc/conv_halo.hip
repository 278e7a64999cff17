// 3x3 stride-1 pad-1 Conv-BN-act, and SpaceToDepth + 3x3 (model_block.py:37-58, 152-175), bf16, with
//   * the input staged as an LDS halo patch (as conv_patch.hip), 32 channels (64-byte rows) per chunk,
//   * the WEIGHTS NOT IN LDS AT ALL: every wave streams its MFMA A-operand fragments straight from L2 into a register ring,
//     in the order it consumes them (packed on the host, ops.py:PackedHalo),
//   * wave tiles of 128 pixels x 64 output channels (8 x 4 MFMA tiles) instead of 64 x 64.
//
// Why (round 3): in conv_patch.hip a wave reads 16 KB of LDS (8 pixel + 8 weight fragments) per 32 MFMAs; two waves per SIMD
// need 1024 MFMA cycles for what costs the CU's one LDS (128 B/clk, shared by four SIMDs) 8 x 128 = 1024 cycles to deliver --
// before the weight DMA writes 32 KB more into it.  The LDS port, not the matrix pipe, was the roof (the pipe measured ~50 %
// busy).  Here a wave reads 8 KB of LDS per 32 MFMAs (no weight fragments, no weight DMA) and each weight fragment is used by
// 8 MFMAs instead of 4: LDS and the vector-memory path are both at about half their rate with the matrix pipe full.
// One barrier per CHUNK (288 MFMAs per wave) instead of one per tap: the patch is the only thing the waves share.
//
// Geometries (template parameter LIN):
//   2D  : a block owns 16 x 16 output pixels; patch row r = py * 18 + px of the (16+2) x (16+2) halo; m-tile i of a wave is
//         output row i (16 pixels), tap (ky, kx) reads the same image (ky * 18 + kx) rows further down.  Partial tiles are masked.
//   LIN : the batch as ONE zero-padded image of width Wp = W + 2 with a zero row between images (conv_patch.hip's linear
//         geometry); a block owns 256 consecutive padded positions, tap (ky, kx) reads ky * Wp + kx rows further down.  For
//         narrow maps (Wp <= 47: the 40 x 40 and 20 x 20 levels), where 16 x 16 tiles would waste up to 44 %.
// LDS row r keeps its four 16-byte chunks at slot c ^ ((r >> 1) & 3): 8 consecutive rows x one logical chunk cover all 32
// banks once, for any starting row (fragments start at arbitrary rows: tap shifts).
// SpaceToDepth: the conv is 3x3 over the virtual (H/2, W/2, 4C) tensor whose channel (2dy+dx)*C + c is stored pixel
// (2y+dy, 2x+dx), channel c; a 32-channel chunk lies inside one (dy, dx), which only changes the patch fill's source address.
// Stride 2 (MODE 2: BaseConv k3 s2 p1, the downsampling convs of net_deploy / net_small_target): out(y, x) reads stored rows
// 2y-1, 2y, 2y+1 = virtual row y-1 phase dy=1, virtual row y phases dy=0 and dy=1 -- a 2x2 conv over the same virtual tensor in
// which a chunk of phase (dy, dx) only has the taps its phase reaches: (0,0) one, (0,1) and (1,0) two, (1,1) four.  Nine
// (tap, phase) pairs, exactly the nine taps of the 3x3.  To keep the chunk loop free of divergent code paths (a four-way branch
// over differently unrolled tap sequences spilled 600 registers) every pass over a staged chunk is TWO taps with run-time
// offsets: passes vq = 0..4 are phase (0,0) [its tap + a zero-weight one], (0,1), (1,0), (1,1) first half, (1,1) second half
// (that phase's chunks are staged twice); 8 fragments per pass, an 8-deep ring, ten taps multiplied for nine needed.
#include <stdlib.h>
#include <mutex>
#include "conv_common.h"

namespace {

constexpr int HCH = 32;                   // channels per chunk
constexpr int HROWB = 64;                 // bytes per patch row
constexpr int HPIECES = 22;               // LDS-DMA pieces (16 rows) per chunk
constexpr int HPROWS = HPIECES * 16;      // 352 >= 18 * 18 = 324 (2D) and >= 256 + 2 * 47 + 2 (LIN)
constexpr int HPATCH = HPROWS * HROWB;    // 22528
constexpr int HRING = 9;                  // weight fragments in flight per wave (two taps ahead); divides 36
constexpr int HFRAGS = 36;                // fragments per chunk and 64 output channels: 9 taps x 4 n-tiles
constexpr int HLINMAXWP = 47;

struct HaloQ {
    SegQ seg[2];
    int nseg, C0;             // C0: channels of segment 0 (plain mode)
    int Creal;                // s2d: channels of the stored tensor
    int Hv, Wv;               // output size = (virtual) input size
    int Hin, Win;             // stored input size
    int B, Cout, nchunks;
    int tiles_x, tiles_y;     // 2D
    int Wp, Hp;               // LIN: padded width W + 2, rows per image H + 1
    unsigned magicWp, shiftWp, magicHp, shiftHp;
    unsigned ntiles;
    const unsigned char* wstream;   // [Cout / 64][nchunks * 36 + HRING][64 lanes][16 B]
    unsigned long long wstride;     // bytes per 64-channel stream
    const float* bias;
    int act;
    const unsigned char* res_ptr;
    int res_ldc, res_coff;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int xcd;
};

// One tap of the current chunk: MT pixel fragments from the patch at the tap's row shift, then NT x MT MFMAs, each weight
// fragment leaving the ring as it is used and its slot refilled with the fragment HRING further down the stream.
#define HALO_TAP(TAP)                                                                                                              \
    {                                                                                                                              \
        constexpr int ky = (TAP) / 3, kx = (TAP) - ky * 3;                                                                         \
        u32x4 fa[MT];                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if (HALO_ABL & 1) { if ((TAP) == 0) { _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pt + lanec[i] * 0 + i * 1024 + loff); } else { _Pragma("unroll") for (int i = 0; i < MT; ++i) { fa[i] = fakeep[i]; } } } else \
        if constexpr (LIN) {                                                                                                       \
            const int r0 = lin_base + ky * lin_wp + kx;                                                                            \
            const int a0 = r0 * HROWB + ((fq ^ ((r0 >> 1) & 3)) << 4);                                                             \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pt + a0 + i * MSTR);                            \
        } else {                                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                fa[i] = *(const u32x4*)(pt + lanec[(2 * (i + ky) + kx) & 7] + ((i + ky) * 18 + kx) * HROWB);                       \
        }                                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                           \
            const int slot = ((TAP) * NT + j) % HRING;                                                                             \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0); \
            if (!(HALO_ABL & 2)) ring[slot] = *(const u32x4*)(wp + HRING * 1024 + loff);                                           \
            wp += 1024;                                                                                                            \
        }                                                                                                                          \
        if (HALO_ABL & 1) { _Pragma("unroll") for (int i = 0; i < MT; ++i) fakeep[i] = fa[i]; }                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    }

#ifndef HALO_ABL
#define HALO_ABL 0   // timing ablations (wrong results): 1 = pixel fragments read once per chunk, 2 = no weight refills
#endif

// One tap of a stride-2 pass: patch offset (KY, KX) in {0,1}^2 as wave-uniform run-time values, ring slots SB .. SB + 3
#define HALO_TAP2(KY, KX, SB)                                                                                                      \
    {                                                                                                                              \
        u32x4 fa[MT];                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        if constexpr (LIN) {                                                                                                       \
            const int r0 = lin_base + (KY) * lin_wp + (KX);                                                                        \
            const int a0 = r0 * HROWB + ((fq ^ ((r0 >> 1) & 3)) << 4);                                                             \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(pt + a0 + i * MSTR);                            \
        } else {                                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                       \
                const int r0 = (wm * MT + i + (KY)) * 18 + (KX) + fr;                                                              \
                fa[i] = *(const u32x4*)(pt + r0 * HROWB + ((fq ^ ((r0 >> 1) & 3)) << 4));                                          \
            }                                                                                                                      \
        }                                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                           \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[(SB) + j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0); \
            ring[(SB) + j] = *(const u32x4*)(wp + RING * 1024 + loff);                                                             \
            wp += 1024;                                                                                                            \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    }

// MODE: 0 = 3x3 s1 p1, 1 = SpaceToDepth + 3x3, 2 = 3x3 s2 p1
template <int BN, bool LIN, int MODE, int ACT>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const HaloQ p) {
    constexpr bool S2D = MODE != 0, ST2 = MODE == 2;   // both read the un-shuffled tensor through the virtual (H/2, W/2, 4C) one
    constexpr int RING = ST2 ? 8 : HRING;
    constexpr int MT = BN == 128 ? 8 : 4, NT = 4;
    constexpr int PXW = MT * 16;                             // pixels per wave
    constexpr int MSTR = LIN ? 16 * HROWB : 18 * HROWB;      // LDS bytes between a wave's m-tiles
    extern __shared__ __align__(16) unsigned char smem[];    // patch[2]

    unsigned tile0, tstride, tend, nslice;
    cc_tile_walk_sliced(p.ntiles, p.xcd, (unsigned)(p.Cout / BN), tile0, tstride, tend, nslice);
    if (tile0 >= tend) return;
    const unsigned my_tiles = (tend - tile0 + tstride - 1) / tstride;
    const unsigned total_chunks = my_tiles * (unsigned)p.nchunks;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = BN == 128 ? (wave >> 1) : wave, wn = BN == 128 ? (wave & 1) : 0;
    const int fr = lane & 15, fq = lane >> 4;
    const int n64 = (int)nslice * (BN / 64) + wn;

    const __amdgpu_buffer_rsrc_t rsrcA0 = make_rsrc(p.seg[0].ptr, p.seg[0].span);
    const __amdgpu_buffer_rsrc_t rsrcA1 = make_rsrc(p.seg[1].ptr, p.seg[1].span);
    constexpr unsigned OOB = 0x80000000u;

    // ---- patch fill.  Piece pc covers LDS rows 16 pc .. 16 pc + 15; lane l writes row 16 pc + (l >> 2), physical slot l & 3, and
    // so fetches logical chunk (l & 3) ^ ((row >> 1) & 3) -- the row's swizzle does not depend on pc (16 pc is a multiple of 8)
    const unsigned prc = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) * 16);
    auto lin_pixel = [&](int q, bool src) -> int {
        if (q < 0) return -1;
        const int yp = (int)fastdiv((unsigned)q, p.magicWp, p.shiftWp), xp = q - yp * p.Wp;
        const int b = (int)fastdiv((unsigned)yp, p.magicHp, p.shiftHp), yy = yp - b * p.Hp;
        const bool real = b < p.B && yy >= 1 && xp >= 1 && xp <= p.Wv;
        if (!real) return -1;
        if (S2D && src) return (b * p.Hin + 2 * (yy - 1)) * p.Win + 2 * (xp - 1);
        return (b * p.Hv + yy - 1) * p.Wv + xp - 1;
    };
    // pixel (index in the stored tensor; s2d: top-left pixel of the 2x2 group; -1 = zero) behind patch row r of a tile: recomputed
    // for every piece that is issued (a dozen VALU instructions beside 288 MFMAs) rather than kept in registers per tile
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
    auto patch_pixel = [&](const TileAt& ta, int r) -> int {
        if (!ta.valid) return -1;
        if constexpr (LIN) {
            return r < 256 + 2 * p.Wp + 2 ? lin_pixel(ta.tile * 256 - p.Wp - 1 + r, true) : -1;
        } else {
            const int py = r / 18, px = r - py * 18;
            const int vy = ta.y0 + py, vx = ta.x0 + px;
            const bool ok = r < 324 && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv;
            return ok ? (ta.b * p.Hin + (S2D ? 2 * vy : vy)) * p.Win + (S2D ? 2 * vx : vx) : -1;
        }
    };
    // chunk cc -> (segment, wave-uniform byte offset), kept in scalar registers explicitly (conv_patch.hip: left to the compiler
    // the resource choice became a waterfall loop)
    const unsigned ldcB0 = p.seg[0].ldcB, ldcB1 = p.seg[1].ldcB;
    auto chunk_src = [&](int cc, bool& s1, unsigned& soff) {
        if (S2D) {
            const int v0 = cc * HCH;
            int q = v0 / p.Creal;                                  // q = 2 dy + dx (stride 2: pass 0..4, passes 3 and 4 are phase 3)
            const int c0 = v0 - q * p.Creal;
            if (ST2 && q == 4) q = 3;
            s1 = false;
            soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((q >> 1) * p.Win + (q & 1)) * ldcB0 + (unsigned)c0 * 2u));
        } else {
            const int c = cc * HCH;
            s1 = __builtin_amdgcn_readfirstlane((int)(c >= p.C0)) != 0;
            soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(c - (s1 ? p.C0 : 0)) * 2u));
        }
    };
    // The pixel behind a lane's row of piece i is the same for every chunk of a tile: it is worked out once per tile (fill_pixels,
    // ~30 VALU instructions per piece) into a table of 6 x 256 words behind the two patch buffers and read back per chunk.  A lane
    // reads only what it wrote itself, so no barrier is involved.  (Kept in registers the six values spilled; recomputed per
    // chunk they were 190 VALU + 80 scalar instructions beside the chunk's 288 MFMAs.)
    int* pixtab = (int*)(smem + 2 * HPATCH);
    auto fill_pixels = [&](const TileAt& ta) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pc = wave + 4 * i;
            if (pc < HPIECES) pixtab[i * 256 + t] = patch_pixel(ta, 16 * pc + (lane >> 2));
        }
    };
    auto patch_piece = [&](int i, int buf, bool s1, unsigned soff) {
        const int pc = wave + 4 * i;
        if (pc >= HPIECES) return;   // wave-uniform
        const int px = pixtab[i * 256 + t];
        const unsigned v = px >= 0 ? (unsigned)px * (s1 ? ldcB1 : ldcB0) + prc : OOB;
        if (s1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA1, (lds_void*)(smem + buf * HPATCH + pc * 1024), 16, (int)v, (int)soff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA0, (lds_void*)(smem + buf * HPATCH + pc * 1024), 16, (int)v, (int)soff, 0, 0);
    };

    // ---- pixel-fragment addresses.  2D: row = (wm * MT + i + ky) * 18 + kx + fr; wm * MT * 18 is a multiple of 8, so the row's
    // swizzle term depends on ((i + ky) * 18 + kx) mod 8 and fr only: eight per-lane constants, everything else an immediate.
    int lanec[8];
    const int lin_base = wm * PXW + fr;
    const int lin_wp = __builtin_amdgcn_readfirstlane(p.Wp);
    if constexpr (!LIN) {
#pragma unroll
        for (int m = 0; m < 8; ++m) lanec[m] = (wm * MT * 18 + fr) * HROWB + ((fq ^ (((m + fr) >> 1) & 3)) << 4);
    }

    // ---- weight ring
    const unsigned char* wbase = p.wstream + (size_t)n64 * p.wstride;   // wave uniform
    const unsigned char* wp = wbase;
    const int loff = lane * 16;
    u32x4 ring[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k) ring[k] = *(const u32x4*)(wp + k * 1024 + loff);
    const long long wrewind = (long long)p.nchunks * (ST2 ? 8 : HFRAGS) * 1024;   // stride 2: 8 per pass, nchunks = passes
    const int cpp = ST2 ? p.Creal / HCH : 1;   // chunks per phase

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fakeep[MT];   // ablation builds only
    (void)fakeep;

    const int nbase = n64 * 64 + fq * 4;
    auto epilogue = [&](unsigned tile) {
        float4 bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = p.bias ? *(const float4*)(p.bias + nbase + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned b = 0;
        int ty = 0, tx = 0;
        if constexpr (!LIN) {
            const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
            b = tile / per_img;
            const unsigned rem = tile - b * per_img;
            ty = (int)(rem / (unsigned)p.tiles_x);
            tx = (int)(rem - (unsigned)ty * p.tiles_x);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int m;
            if constexpr (LIN) {
                m = lin_pixel((int)tile * 256 + wm * PXW + i * 16 + fr, false);
            } else {
                const int oy = ty * 16 + wm * MT + i, ox = tx * 16 + fr;
                m = (oy < p.Hv && ox < p.Wv) ? ((int)b * p.Hv + oy) * p.Wv + ox : -1;
            }
            if (m >= 0) {
                unsigned char* orow = p.out_ptr + ((size_t)m * p.out_ldc + p.out_coff + nbase) * 2;
                uint2 rv[NT];
                if (p.res_ptr) {   // Bottleneck shortcut: the m-tile's four loads go out together, ahead of the SiLU arithmetic
                    const unsigned char* rrow = p.res_ptr + ((size_t)m * p.res_ldc + p.res_coff + nbase) * 2;
#pragma unroll
                    for (int j = 0; j < NT; ++j) rv[j] = *(const uint2*)(rrow + j * 32);
                }
                float v[NT][4];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    v[j][0] = acc[i][j][0] + bv[j].x; v[j][1] = acc[i][j][1] + bv[j].y; v[j][2] = acc[i][j][2] + bv[j].z; v[j][3] = acc[i][j][3] + bv[j].w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (ACT == ACT_RUNTIME) v[j][r] = apply_act(v[j][r], p.act);
                        else v[j][r] = act_fn<ACT>(v[j][r]);
                    }
                }
                if (p.res_ptr) {   // added after the activation (model_block.py:213-218)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        v[j][0] += bf16_bits_to_f32(rv[j].x & 0xFFFFu); v[j][1] += bf16_bits_to_f32(rv[j].x >> 16);
                        v[j][2] += bf16_bits_to_f32(rv[j].y & 0xFFFFu); v[j][3] += bf16_bits_to_f32(rv[j].y >> 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) *(uint2*)(orow + j * 32) = make_uint2(pack_bf16x2(v[j][0], v[j][1]), pack_bf16x2(v[j][2], v[j][3]));
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Nothing this lambda loaded may still be pending where its path joins the main loop: the compiler would otherwise put a
        // full vmcnt(0) in front of EVERY chunk's first LDS read (the bias registers are reused there), draining the weight ring and
        // the patch DMA each time.  Here it costs one drain per tile, before the next chunk's DMA has been issued.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as an instruction the compiler's own wait-count bookkeeping sees
    };

    // ---- prologue: chunk 0 of the first tile
    int pf_cc = 0;
    unsigned pf_tile = tile0;
    bool pf_s1;
    unsigned pf_soff;
    fill_pixels(tile_at(pf_tile));
    chunk_src(pf_cc, pf_s1, pf_soff);
#pragma unroll
    for (int i = 0; i < 6; ++i) patch_piece(i, 0, pf_s1, pf_soff);
    auto pf_advance = [&]() {
        if (++pf_cc == p.nchunks) {
            pf_cc = 0;
            pf_tile += tstride;
            fill_pixels(tile_at(pf_tile));
        }
        chunk_src(pf_cc, pf_s1, pf_soff);
    };
    pf_advance();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0).  (Waits are builtins, not inline asm: the compiler's wait-count
                                          // bookkeeping does not see an asm wait and answers with full vmcnt(0) drains of its own.)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur_cc = 0;
    unsigned cur_tile = tile0, pend_tile = 0;
    bool pending = false;
    for (unsigned ci = 0; ci < total_chunks; ++ci) {
        const bool next_chunk = ci + 1 < total_chunks;
        const unsigned char* pt = smem + (ci & 1) * HPATCH;
        if (pending) {
            if (!(HALO_ABL & 8) || cur_tile + tstride >= tend) epilogue(pend_tile);
            pending = false;
        }
        if (next_chunk && !(HALO_ABL & 4)) {
#pragma unroll
            for (int i = 0; i < 6; ++i) patch_piece(i, (int)((ci + 1) & 1), pf_s1, pf_soff);
        }
        if constexpr (ST2) {
            // pass vq over a staged chunk: two taps (offsets into the 2 x 2 window).  vq 0: (1,1) + the zero-weight tap; 1: (1,0)
            // (1,1); 2: (0,1) (1,1); 3: (0,0) (0,1); 4: (1,0) (1,1)
            const int vq = __builtin_amdgcn_readfirstlane(cur_cc / cpp);
            const int kya = (vq == 2 || vq == 3) ? 0 : 1, kxa = (vq == 0 || vq == 2) ? 1 : 0;
            const int kyb = vq == 3 ? 0 : 1, kxb = 1;
            HALO_TAP2(kya, kxa, 0) HALO_TAP2(kyb, kxb, 4)
        } else {
            HALO_TAP(0) HALO_TAP(1) HALO_TAP(2) HALO_TAP(3) HALO_TAP(4) HALO_TAP(5) HALO_TAP(6) HALO_TAP(7) HALO_TAP(8)
        }
        if (next_chunk) pf_advance();
        if (++cur_cc == p.nchunks) {
            cur_cc = 0;
            pending = true;
            pend_tile = cur_tile;
            cur_tile += tstride;
            wp -= wrewind;   // the stream ends with a copy of its first HRING fragments: the ring already holds the next tile's
        }
        // The next chunk's patch pieces were issued before this chunk's 36 ring loads, and vector-memory operations return in
        // order: with at most HRING of them outstanding the pieces have landed.  (This chunk's last MFMAs already waited for
        // ring loads issued after the pieces, so the wait is normally satisfied on arrival.)
        static_assert(HRING == 9, "the immediates below encode vmcnt(RING)");
        if constexpr (ST2) __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): every chunk issues at least 8 ring loads
        else __builtin_amdgcn_s_waitcnt(0x0079);                 // vmcnt(9) lgkmcnt(0)
        if (!(HALO_ABL & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (pending) epilogue(pend_tile);
}

template <int BN, bool LIN, int MODE, int ACT>
int launch_halo_k(const HaloQ& p, hipStream_t st) {
    auto kernel = conv_halo_kernel<BN, LIN, MODE, ACT>;
    constexpr int LDS = 2 * HPATCH + 6 * 256 * 4;   // two patch buffers + the per-tile pixel table
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
    const int gy = p.Cout / BN;
    const int gx = cc_grid_x(resident, gy, p.ntiles);
    hipLaunchKernelGGL(kernel, dim3(gx * gy), dim3(256), LDS, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

template <int BN, bool LIN, int MODE>
int launch_halo_act(const HaloQ& p, hipStream_t st) {
    if (p.act == CCNET_ACT_SILU) return launch_halo_k<BN, LIN, MODE, CCNET_ACT_SILU>(p, st);
    if (MODE == 0 && p.act == CCNET_ACT_NONE) return launch_halo_k<BN, LIN, MODE, CCNET_ACT_NONE>(p, st);
    return launch_halo_k<BN, LIN, MODE, ACT_RUNTIME>(p, st);
}

template <int BN>
int launch_halo(const HaloQ& p, bool lin, int mode, hipStream_t st) {
    if (mode == 2) return lin ? launch_halo_act<BN, true, 2>(p, st) : launch_halo_act<BN, false, 2>(p, st);
    if (lin) return mode ? launch_halo_act<BN, true, 1>(p, st) : launch_halo_act<BN, true, 0>(p, st);
    return mode ? launch_halo_act<BN, false, 1>(p, st) : launch_halo_act<BN, false, 0>(p, st);
}

void halo_magic(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *shift = s;
    *magic = (unsigned)((((1ull << s) - d) << 32) / d + 1);
}

}  // namespace

extern "C" size_t ccnet_conv_halo_stream_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || (Cin % HCH) != 0 || (Cout % 64) != 0) return 0;
    return (size_t)(Cout / 64) * ((size_t)(Cin / HCH) * HFRAGS + HRING) * 512;
}

// stride-2 stream: five passes of two taps over the chunks of C, 8 fragments per pass and chunk
extern "C" size_t ccnet_conv_halo_s2_stream_elems(int C, int Cout) {
    if (C <= 0 || Cout <= 0 || (C % HCH) != 0 || (Cout % 64) != 0) return 0;
    return (size_t)(Cout / 64) * ((size_t)(C / HCH) * 40 + 8) * 512;
}

extern "C" int ccnet_conv_halo(const ccnet_slice* in, int nseg, int s2d, int B, int H, int W, int Cout, const void* wstream_dev,
                               const float* bias_dev, int act, ccnet_slice residual, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(in && (nseg == 1 || nseg == 2) && wstream_dev, "conv_halo: null pointer / bad segment count");
    CC_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && (Cout % 64) == 0, "conv_halo: bad sizes (B=%d H=%d W=%d Cout=%d; Cout must be a multiple of 64)", B, H, W, Cout);
    CC_REQUIRE(act >= CCNET_ACT_NONE && act <= CCNET_ACT_LRELU, "conv_halo: unknown activation %d", act);
    CC_REQUIRE(s2d >= 0 && s2d <= 2, "conv_halo: mode must be 0 (3x3), 1 (SpaceToDepth + 3x3) or 2 (3x3 stride 2)");
    CC_REQUIRE(!s2d || (nseg == 1 && (H % 2) == 0 && (W % 2) == 0), "conv_halo: SpaceToDepth / stride-2 modes take one input of even size");
    CC_REQUIRE(((uintptr_t)wstream_dev % 16) == 0 && ((uintptr_t)bias_dev % 16) == 0, "conv_halo: weights / bias must be 16-byte aligned");
    HaloQ p{};
    int ctot = 0;
    for (int s = 0; s < nseg; ++s) {
        const ccnet_slice& sl = in[s];
        CC_REQUIRE(cc_slice_vec_ok(sl, CCNET_BF16) && sl.shift == 0 && (sl.C % HCH) == 0,
                   "conv_halo: input slice %d must be bf16, un-shifted, 16-byte aligned with a multiple of 32 channels (C=%d ldc=%d coff=%d)", s, sl.C,
                   sl.ldc, sl.coff);
        const double span = (((double)B * H * W - 1.0) * sl.ldc + sl.C) * 2.0;
        CC_REQUIRE(span < 2147483648.0, "conv_halo: input tensor of 2 GiB or more");
        p.seg[s] = SegQ{(const unsigned char*)sl.ptr + (size_t)sl.coff * 2, (unsigned)(sl.ldc * 2), 0, H, W, (unsigned)span};
        ctot += sl.C;
    }
    if (nseg == 1) p.seg[1] = p.seg[0];
    p.nseg = nseg;
    p.C0 = nseg == 2 ? in[0].C : ctot;
    p.Creal = in[0].C;
    const int cvirt = s2d ? 4 * ctot : ctot;
    p.Hin = H; p.Win = W;
    p.Hv = s2d ? H / 2 : H; p.Wv = s2d ? W / 2 : W;
    p.B = B; p.Cout = Cout;
    p.nchunks = s2d == 2 ? 5 * ctot / HCH : cvirt / HCH;   // stride 2: five passes over the four phases' chunks
    const long long M = (long long)B * p.Hv * p.Wv;
    CC_REQUIRE(M < (1ll << 30), "conv_halo: too many pixels");
    // geometry: 16 x 16 tiles, or the linear padded space where that wastes less (narrow maps)
    const int tx = (p.Wv + 15) / 16, ty = (p.Hv + 15) / 16;
    const double waste2d = (double)tx * ty * 256.0 / ((double)p.Hv * p.Wv);
    const double wastelin = (double)(p.Wv + 2) * (p.Hv + 1) / ((double)p.Hv * p.Wv);
    const bool lin = p.Wv + 2 <= HLINMAXWP && wastelin < waste2d;
    if (lin) {
        p.Wp = p.Wv + 2; p.Hp = p.Hv + 1;
        const long long q = ((long long)B * p.Hp + 1) * p.Wp;
        CC_REQUIRE(q < (1ll << 30), "conv_halo: too many padded positions");
        p.ntiles = (unsigned)((q + 255) / 256);
        halo_magic((unsigned)p.Wp, &p.magicWp, &p.shiftWp);
        halo_magic((unsigned)p.Hp, &p.magicHp, &p.shiftHp);
    } else {
        p.tiles_x = tx; p.tiles_y = ty;
        p.ntiles = (unsigned)((long long)B * tx * ty);
        p.Wp = 18;
    }
    p.wstream = (const unsigned char*)wstream_dev;
    p.wstride = s2d == 2 ? ((unsigned long long)p.nchunks * 8 + 8) * 1024ull : ((unsigned long long)p.nchunks * HFRAGS + HRING) * 1024ull;
    p.bias = bias_dev;
    p.act = act;
    CC_REQUIRE(out.ptr && out.shift == 0 && out.C == Cout && out.coff >= 0 && out.ldc >= out.coff + out.C && (out.ldc % 4) == 0 && (out.coff % 4) == 0 &&
                   ((uintptr_t)out.ptr % 8) == 0, "conv_halo: bad output slice (C=%d ldc=%d coff=%d)", out.C, out.ldc, out.coff);
    p.out_ptr = (unsigned char*)out.ptr; p.out_ldc = out.ldc; p.out_coff = out.coff;
    if (residual.ptr) {
        CC_REQUIRE(residual.shift == 0 && residual.C == Cout && residual.coff >= 0 && residual.ldc >= residual.coff + residual.C && (residual.ldc % 4) == 0 &&
                       (residual.coff % 4) == 0 && ((uintptr_t)residual.ptr % 8) == 0, "conv_halo: bad residual slice");
        p.res_ptr = (const unsigned char*)residual.ptr; p.res_ldc = residual.ldc; p.res_coff = residual.coff;
    }
    p.xcd = cc_xcd_remap();
    const double kreal = s2d == 2 ? 9.0 * ctot : 9.0 * cvirt;   // stride 2: the nine taps of the 3x3 over the stored channels
    const double flops = 2.0 * (double)M * Cout * kreal;
    const double bytes = ((double)B * H * W * ctot + (double)Cout * kreal) * 2.0 + (double)M * Cout * (2.0 + (residual.ptr ? 2.0 : 0.0));
    const int s2 = s2d;
    // 128 output channels per workgroup wherever Cout allows.  A rule that switched to 64-channel tiles when the 128-channel ones
    // leave workgroup slots empty looked better layer by layer in a microbenchmark and cost 5 % of the MFMA-bound group inside
    // the network (same-box A/B, scripts/ab_env.sh: 2.35 against 2.23 ms per forward).  Both shapes accumulate a pixel's sum in the
    // same order, so the choice never changes a bit of the result.  CCNET_HALO_BN=64 forces the small tile (measurement, test).
    static const char* force_bn = getenv("CCNET_HALO_BN");
    const int bn = ((Cout % 128) == 0 && !(force_bn && force_bn[0] == '6')) ? 128 : 64;
    return ccnet_submit(
        [p, lin, s2, bn](hipStream_t st, const float*) -> int { return bn == 128 ? launch_halo<128>(p, lin, s2, st) : launch_halo<64>(p, lin, s2, st); },
        stream, CCNET_OP_CONV, flops, bytes);
}
