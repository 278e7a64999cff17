// 3x3 stride-1 pad-1 convolution (and SpaceToDepth + 3x3, given in its 6x6 stride-2 pad-2 form) with the input staged as
// an LDS halo patch instead of an im2col gather.
//
// Why: in conv_igemm every K step pulls 128 pixels x 128 B of A through the vector-memory path although eight of the
// nine taps re-read what the previous ones fetched; that path (LDS-DMA issue, 64 B/clk/CU) and not the MFMA pipe is what
// the 3x3 layers wait on.  Here a block owns an 8 x 16 tile of output pixels and loops K as (64-channel chunk) x (9 taps):
// the (8+2) x (16+2) input pixels of a chunk are fetched ONCE (23 KB instead of 9 x 16 KB), and each tap's MFMA operand
// is the same LDS image read at a shifted row.  The weights still stream one 128-B K slice per step.
//
//  * LDS patch: row r = py*18 + px holds 128 B (64 bf16 / 32 fp32 channels) of patch pixel (py, px); 16-B chunk c sits
//    at slot c ^ ((px >> 1) & 7).  The swizzle depends on px only (18 is even), so with the tap loop unrolled every
//    fragment address is a per-lane constant (6 registers: 3 kx x 2 K halves) plus an immediate.
//  * Fill: 24 LDS-DMA pieces of 8 rows per chunk (6 per wave), issued one per K step during the previous chunk behind
//    that step's weight pieces, so `s_waitcnt vmcnt(1 + stores)` waits for the weights and leaves the piece in flight.
//    Out-of-image pixels use an out-of-range buffer offset and arrive as zeros.
//  * SpaceToDepth mode: the conv is 3x3 over the virtual (H/2, W/2, 4C) tensor whose channel (2dy+dx)*C + c is input
//    pixel (2y+dy, 2x+dx), channel c (model_block.py:169-173).  A 64-channel chunk is one (dy,dx) (C >= 64) or the pair
//    dx = 0,1 of one dy (C = 32); its weights are contiguous in the 6x6 packing, so the same packed tensor is used.
//  * Epilogue, weight tiles, MFMA operand order: as conv_igemm (registers only, weights are the "A" operand).
//  * Linear geometry (LIN, output maps narrower than 30 pixels whose width is not a multiple of 8 -- the 20 x 20 maps, plain
//    3x3 and SpaceToDepth layers alike): the batch is
//    laid out as ONE zero-padded image of width Wp = W + 2 with a zero row between consecutive images; a tile is 128
//    consecutive positions q = Yp * Wp + Xp of that padded space (pad positions are computed and thrown away: 13 % at 20 x 20),
//    its patch the 128 + 2 Wp + 2 positions from q0 - Wp - 1 on, and tap (ky, kx) is the same LDS image read ky * Wp + kx rows
//    further down.  LDS row r keeps 16-B chunk c at slot c ^ ((r >> 1) & 7) (conv_igemm's swizzle; 16-row steps leave it alone).
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

namespace {

// Tile geometry, template parameter MC: an MFMA m-tile is MR x MC output pixels (MR * MC = 16) and a block stacks 8 of them
// vertically: 8 x 16 pixels (MC = 16, widths % 16 == 0) or 16 x 8 (MC = 8, widths % 8 == 0).  Both have a 180-pixel patch.
constexpr int PPIX = 180;
constexpr int PIECES = 6;                                          // LDS-DMA pieces per wave and chunk (24 x 8 rows = 192 >= 180)
constexpr int PROWS = 4 * PIECES * 8;
constexpr int PATCH_BYTES = PROWS * BKB;                           // 24576

struct PatchQ {
    SegQ seg[2];
    int nseg, Ctot, C0;       // Ctot: channels the 3x3 sees (4*C in s2d mode)
    int s2d, Creal;           // s2d: input is (B, 2*Hv, 2*Wv, Creal)
    int Hv, Wv;               // (virtual) input = output size
    int Hin, Win;             // stored input size
    int Cout;
    int tiles_x, tiles_y;
    unsigned ntiles;
    const unsigned char* weight;
    int Kpad, nchunks;
    const float* bias;
    int act;
    const unsigned char* res_ptr;
    int res_ldc, res_coff;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int vec_out;
    int mc;  // tile geometry (16 or 8), see PPIX
    int xcd; // tile order, see cc_tile_walk
    int Wp, Hp;  // linear geometry: padded width W + 2, padded rows per image H + 1
    int B;
};

// MFMAs of one tap: patch fragments at the tap's shifted rows, weight fragments from the 128-byte slice at BSP
#define PATCH_TAP(TAP, BSP)                                                                                             \
            { \
                constexpr int tky = (TAP) / 3, tkx = (TAP) - tky * 3; \
                const unsigned char* Bs = (BSP); \
                u32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT]; \
                int pa0, pa1; \
                if constexpr (LIN) { \
                    const int r0 = lin_row + tky * lin_wp + tkx, sw = (r0 >> 1) & 7; \
                    pa0 = r0 * BKB + ((fq ^ sw) << 4); \
                    pa1 = r0 * BKB + (((4 + fq) ^ sw) << 4); \
                } else { \
                    pa0 = abase[tkx][0] + tky * PW * BKB; \
                    pa1 = abase[tkx][1] + tky * PW * BKB; \
                } \
_Pragma("unroll") \
                for (int i = 0; i < MT; ++i) fa0[i] = *(const u32x4*)(pt + pa0 + i * PSTEP); \
_Pragma("unroll") \
                for (int j = 0; j < NT; ++j) fb0[j] = *(const u32x4*)(Bs + bbase[j][0]); \
_Pragma("unroll") \
                for (int i = 0; i < MT; ++i) fa1[i] = *(const u32x4*)(pt + pa1 + i * PSTEP); \
_Pragma("unroll") \
                for (int j = 0; j < NT; ++j) fb1[j] = *(const u32x4*)(Bs + bbase[j][1]); \
                __builtin_amdgcn_sched_barrier(0); \
                __builtin_amdgcn_s_setprio(1); \
_Pragma("unroll") \
                for (int i = 0; i < MT; ++i) \
_Pragma("unroll") \
                    for (int j = 0; j < NT; ++j) Frag<T>::mma(fb0[j], fa0[i], acc[i][j]); \
_Pragma("unroll") \
                for (int i = 0; i < MT; ++i) \
_Pragma("unroll") \
                    for (int j = 0; j < NT; ++j) Frag<T>::mma(fb1[j], fa1[i], acc[i][j]); \
                __builtin_amdgcn_s_setprio(0); \
                __builtin_amdgcn_sched_barrier(0); \
            }

// One K step (chunk, tap) of the main loop with the tap as a compile-time constant; expanded nine times in the kernel so
// that every LDS fragment address is a per-lane constant plus an immediate (a lambda here kept the accumulators in memory).
#define PATCH_STEP(TAP)                                                                                                 \
    {                                                                                                               \
        constexpr int tap = TAP;                                                                                    \
            const int bpar = (cpar ^ tap) & 1; \
            int after = 0; \
            if (tap < 8) b_tile(bpar ^ 1, b_koff(kb_cur, tap + 1)); \
            else if (next_chunk) b_tile(bpar ^ 1, b_koff(kb_next, 0)); \
            if (tap < PIECES && next_chunk) { \
                patch_piece(tap, cpar ^ 1, pf_s1, pf_soff); \
                after = 1; \
            } \
            bool counted = true; \
            if (tap == 0 && pending) { \
                const bool full = epilogue(pend_tile); \
                if (full) after += S; \
                else counted = false; \
                pending = false; \
            } \
            PATCH_TAP(TAP, bt0 + bpar * B_BYTES) \
            if (!counted) WAIT_VM(0); \
            else if (after == 0) WAIT_VM(0); \
            else if (after == 1) WAIT_VM(1); \
            else if (after == S) WAIT_VM(S); \
            else WAIT_VM(S + 1); \
            __builtin_amdgcn_s_barrier(); \
            asm volatile("" ::: "memory"); \
    }

// 64-column tiles (BN == 64): a tap is only 16 MFMAs per wave, and a K step costs about as much in DMA issue, waits and the
// barrier as those take.  There the taps go in pairs -- steps (0,1) (2,3) (4,5) (6,7) (8) of a chunk, five barriers instead of
// nine -- with both taps' weight slices side by side in a buffer twice as wide (same LDS as the 128-column kernel).
#define PATCH_DSTEP(SI)                                                                                                 \
    {                                                                                                               \
        constexpr int si = SI, t0 = 2 * si, t1 = 2 * si + 1;                                                        \
            const int bpar = (cpar ^ si) & 1; \
            int after = 0; \
            if (si < 4) { \
                b_tile2(bpar ^ 1, 0, b_koff(kb_cur, t0 + 2)); \
                if (t1 + 2 <= 8) b_tile2(bpar ^ 1, 1, b_koff(kb_cur, t1 + 2 <= 8 ? t1 + 2 : 0)); \
            } else if (next_chunk) { \
                b_tile2(bpar ^ 1, 0, b_koff(kb_next, 0)); \
                b_tile2(bpar ^ 1, 1, b_koff(kb_next, 1)); \
            } \
            if (si < 3 && next_chunk) { \
                patch_piece(2 * si, cpar ^ 1, pf_s1, pf_soff); \
                patch_piece(2 * si + 1, cpar ^ 1, pf_s1, pf_soff); \
                after = 2; \
            } \
            bool counted = true; \
            if (si == 0 && pending) { \
                const bool full = epilogue(pend_tile); \
                if (full) after += S; \
                else counted = false; \
                pending = false; \
            } \
            PATCH_TAP(t0, bt0 + (bpar * 2) * B_BYTES) \
            if (t1 <= 8) PATCH_TAP((t1 <= 8 ? t1 : 0), bt0 + (bpar * 2 + 1) * B_BYTES) \
            if (!counted) WAIT_VM(0); \
            else if (after == 0) WAIT_VM(0); \
            else if (after == 2) WAIT_VM(2); \
            else if (after == S) WAIT_VM(S); \
            else WAIT_VM(S + 2); \
            __builtin_amdgcn_s_barrier(); \
            asm volatile("" ::: "memory"); \
    }

template <typename T, typename OutT, int BN, int MC, int ACT, bool S2D, bool LIN = false>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(const PatchQ p) {
    constexpr int MR = 16 / MC, TH = 8 * MR, TW = MC, PW = MC + 2;
    static_assert((TH + 2) * PW == PPIX && (PW % 2) == 0, "patch geometry");
    constexpr int PSTEP = LIN ? 16 * BKB : MR * PW * BKB;  // LDS bytes between a wave's m-tiles
    constexpr int ES = (int)sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int BK = BKB / ES;  // channels per chunk
    constexpr int WTN = BN / 2, MT = 4, NT = WTN / 16;
    constexpr int B_LOADS = BN / 32;
    constexpr int B_BYTES = BN * BKB;
    constexpr int S = MT * NT;  // epilogue stores per lane of a full tile
    extern __shared__ __align__(16) unsigned char smem[];  // patch[2] | B[2]
    unsigned char* const patch0 = smem;
    unsigned char* const bt0 = smem + 2 * PATCH_BYTES;

    unsigned tile0, tstride, tend, nslice;
    cc_tile_walk_sliced(p.ntiles, p.xcd, (unsigned)((p.Cout + BN - 1) / BN), tile0, tstride, tend, nslice);
    if (tile0 >= tend) return;
    const unsigned my_tiles = (tend - tile0 + tstride - 1) / tstride;
    const unsigned total_chunks = my_tiles * (unsigned)p.nchunks;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = (int)nslice * BN;

    const __amdgpu_buffer_rsrc_t rsrcA0 = make_rsrc(p.seg[0].ptr, p.seg[0].span);
    const __amdgpu_buffer_rsrc_t rsrcA1 = make_rsrc(p.seg[1].ptr, p.seg[1].span);
    const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.weight + (size_t)n0 * p.Kpad * ES, (unsigned)BN * (unsigned)p.Kpad * ES);
    constexpr unsigned OOB = 0x80000000u;

    // ---- weight tile: as conv_igemm (row lr + 32 i, logical chunk under the row swizzle)
    const int lr = t >> 3;
    const int lcb = (t & 7) ^ ((lr >> 1) & 7);
    unsigned boff[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) boff[i] = (unsigned)((lr + 32 * i) * p.Kpad + lcb * EPC) * ES;

    // ---- patch fill: piece (wave + 4 i) covers LDS rows 8*(wave + 4 i) .. +7; this lane owns row r_i, physical slot lane & 7
    // pixel offsets (in pixels of the stored tensor, -1 = zero) and the byte offset of the lane's logical chunk inside the
    // pixel's channel window, for the tile being prefetched (recomputed per tile: a few divisions by constants)
    int pix[PIECES];
    unsigned pr_c[PIECES];
    // linear geometry: padded position -> pixel index (-1 for pad positions) of the output, or of the stored input tensor
    // (SpaceToDepth: the top-left pixel of the 2x2 group, plus dx when a chunk holds both columns)
    auto lin_pixel = [&](int q, bool src, int dx) -> int {
        if (q < 0) return -1;
        const int yp = q / p.Wp, xp = q - yp * p.Wp;
        const int b = yp / p.Hp, yy = yp - b * p.Hp;
        const bool real = b < p.B && yy >= 1 && xp >= 1 && xp <= p.Wv;
        if (!real) return -1;
        if (S2D && src) return (b * p.Hin + 2 * (yy - 1)) * p.Win + 2 * (xp - 1) + dx;
        return (b * p.Hv + yy - 1) * p.Wv + xp - 1;
    };
    const int lin_wp = __builtin_amdgcn_readfirstlane(p.Wp);
    const int lin_row = wm * 64 + fr;  // LDS row of the lane's pixel in m-tile 0 under tap (0, 0)
    auto setup_tile = [&](unsigned tile) {
        if constexpr (LIN) {
            const bool tv = tile < tend;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int r = 8 * (wave + 4 * i) + (lane >> 3);
                const int lc = (lane & 7) ^ ((r >> 1) & 7);
                int ch = lc * EPC, dx = 0;
                if (S2D && p.Creal < BK) { dx = ch >= p.Creal ? 1 : 0; ch -= dx * p.Creal; }
                pr_c[i] = (unsigned)ch * ES;
                const int px = (tv && r < 128 + 2 * p.Wp + 2) ? lin_pixel((int)tile * 128 - p.Wp - 1 + r, true, dx) : -1;
                pix[i] = px;
            }
            return;
        }
        const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
        const unsigned b = tile / per_img, rem = tile - b * per_img;
        const int ty = (int)(rem / (unsigned)p.tiles_x), tx = (int)(rem - (unsigned)ty * p.tiles_x);
        const bool tv = tile < tend;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int r = 8 * (wave + 4 * i) + (lane >> 3);
            const int py = r / PW, px = r - py * PW;
            const int lc = (lane & 7) ^ ((px >> 1) & 7);
            int ch = lc * EPC;  // first channel of the lane's chunk within the BK-channel window
            int dx = 0;
            if (S2D && p.Creal < BK) { dx = ch >= p.Creal ? 1 : 0; ch -= dx * p.Creal; }  // Creal == BK/2 there
            pr_c[i] = (unsigned)ch * ES;
            const int vy = ty * TH - 1 + py, vx = tx * TW - 1 + px;
            const bool ok = tv && r < PPIX && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv;
            const int sy = S2D ? 2 * vy : vy, sx = S2D ? 2 * vx + dx : vx;
            pix[i] = ok ? ((int)b * p.Hin + sy) * p.Win + sx : -1;
        }
    };
    // chunk cc of the prefetch tile -> (segment, uniform byte offset); every quantity here is wave-uniform and is kept in
    // scalar registers explicitly (readfirstlane): left to the compiler, the resource / offset choice of the patch DMA became
    // a waterfall loop and the pixel stride a scalar load followed by a full lgkmcnt(0) wait, once per piece
    const unsigned ldcB0 = p.seg[0].ldcB, ldcB1 = p.seg[1].ldcB;
    auto chunk_src = [&](int cc, bool& s1, unsigned& soff) {
        if (S2D) {
            const int v0 = cc * BK;
            const int q = v0 / p.Creal, c0 = v0 - q * p.Creal;  // q = 2*dy + dx of the first channel
            s1 = false;
            soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((q >> 1) * p.Win + (p.Creal < BK ? 0 : (q & 1))) * ldcB0 + (unsigned)c0 * ES));
        } else {
            const int c = cc * BK;
            s1 = __builtin_amdgcn_readfirstlane((int)(c >= p.C0)) != 0;
            soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(c - (s1 ? p.C0 : 0)) * ES));
        }
    };
    auto patch_piece = [&](int i, int buf, bool s1, unsigned soff) {
        const unsigned ldcB = s1 ? ldcB1 : ldcB0;
        const unsigned v = pix[i] >= 0 ? (unsigned)pix[i] * ldcB + pr_c[i] : OOB;
        if (s1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA1, (lds_void*)(patch0 + buf * PATCH_BYTES + (wave + 4 * i) * 1024), 16, (int)v, (int)soff, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA0, (lds_void*)(patch0 + buf * PATCH_BYTES + (wave + 4 * i) * 1024), 16, (int)v, (int)soff, 0, 0);
    };
    // weights of K step (cc, tap): byte offset of its 128-B slice in a packed row = kbase(cc) + tapmul(tap) * tstride, where
    // only kbase depends on the chunk (computed once per chunk, scalar) and tapmul is a compile-time constant of the step
    const int kstride = __builtin_amdgcn_readfirstlane((S2D ? p.Creal : p.Ctot) * ES);
    auto kbase = [&](int cc) -> int {
        if (S2D) {
            const int v0 = cc * BK;
            const int q = v0 / p.Creal, c0 = v0 - q * p.Creal;
            return __builtin_amdgcn_readfirstlane(((((q >> 1) * 6 + (q & 1)) * p.Creal) + c0) * ES);
        }
        return __builtin_amdgcn_readfirstlane(cc * BK * ES);
    };
    auto b_koff = [&](int kb, int tap) -> int {
        const int u = tap / 3, v = tap - u * 3;
        return kb + (S2D ? 12 * u + 2 * v : tap) * kstride;
    };
    constexpr bool DBL = BN == 64;  // taps in pairs, see PATCH_DSTEP
    auto b_tile2 = [&](int buf, int slice, int koff) {
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(bt0 + (buf * 2 + slice) * B_BYTES + wave * 1024 + i * 4096), 16, (int)boff[i], koff, 0, 0);
    };
    auto b_tile = [&](int buf, int koff) {
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(bt0 + buf * B_BYTES + wave * 1024 + i * 4096), 16, (int)boff[i], koff, 0, 0);
    };

    // ---- MFMA operand addresses
    int abase[3][2];  // [kx][K half]: lane part of the patch fragment address
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int h = 0; h < 2; ++h) abase[kx][h] = ((wm * 4 * MR + fr / MC) * PW + fr % MC + kx) * BKB + (((4 * h + fq) ^ (((fr % MC + kx) >> 1) & 7)) << 4);
    int bbase[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) bbase[j][h] = lds_off(wn * WTN + j * 16 + fr, 4 * h + fq);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nbase = n0 + wn * WTN + fq * 4;
    float breg[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nbase + j * 16 + r;
            breg[j][r] = (p.bias && n < p.Cout) ? p.bias[n] : 0.0f;
        }
    const bool n_full = p.vec_out && (n0 + BN <= p.Cout);

    auto finish4 = [&](int i, int j, float* v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = acc[i][j][r] + breg[j][r];
            if constexpr (ACT == ACT_RUNTIME) v[r] = apply_act(x, p.act);
            else v[r] = act_fn<ACT>(x);
        }
    };
    auto store4 = [&](unsigned char* op, const float* v) {
        if constexpr (sizeof(OutT) == 2) *(uint2*)op = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        else *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
    };
    // returns true when exactly S store instructions per lane were issued (and no loads)
    auto epilogue = [&](unsigned tile) -> bool {
        if constexpr (LIN) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = lin_pixel((int)tile * 128 + wm * 64 + i * 16 + fr, false, 0);
                if (m >= 0) {
                    unsigned char* orow = p.out_ptr + ((size_t)m * p.out_ldc + p.out_coff) * sizeof(OutT);
                    const unsigned char* rrow = p.res_ptr ? p.res_ptr + ((size_t)m * p.res_ldc + p.res_coff) * ES : nullptr;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n = nbase + j * 16;
                        if (n >= p.Cout) continue;
                        float v[4];
                        finish4(i, j, v);
                        if (p.vec_out && n + 4 <= p.Cout) {
                            if (rrow) {
                                if constexpr (ES == 2) {
                                    const uint2 rv = *(const uint2*)(rrow + (size_t)n * 2);
                                    v[0] += bf16_bits_to_f32(rv.x & 0xFFFFu); v[1] += bf16_bits_to_f32(rv.x >> 16);
                                    v[2] += bf16_bits_to_f32(rv.y & 0xFFFFu); v[3] += bf16_bits_to_f32(rv.y >> 16);
                                } else {
                                    const float4 rv = *(const float4*)(rrow + (size_t)n * 4);
                                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                                }
                            }
                            store4(orow + (size_t)n * sizeof(OutT), v);
                        } else {
                            for (int r = 0; r < 4 && n + r < p.Cout; ++r) {
                                float x = v[r];
                                if (rrow) x += load1<T>((const T*)rrow + n + r);
                                store1<OutT>((OutT*)orow + n + r, x);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return false;
        }
        const unsigned per_img = (unsigned)(p.tiles_x * p.tiles_y);
        const unsigned b = tile / per_img, rem = tile - b * per_img;
        const int ty = (int)(rem / (unsigned)p.tiles_x), tx = (int)(rem - (unsigned)ty * p.tiles_x);
        const int oy0 = ty * TH + wm * 4 * MR + fr / MC, ox = tx * TW + fr % MC;
        const bool full = n_full && !p.res_ptr && ty * TH + TH <= p.Hv && tx * TW + TW <= p.Wv;
        if (full) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const size_t m = ((size_t)b * p.Hv + oy0 + i * MR) * p.Wv + ox;
                unsigned char* orow = p.out_ptr + (m * p.out_ldc + p.out_coff + nbase) * sizeof(OutT);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float v[4];
                    finish4(i, j, v);
                    store4(orow + j * 16 * sizeof(OutT), v);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int oy = oy0 + i * MR;
                if (oy < p.Hv && ox < p.Wv) {
                    const size_t m = ((size_t)b * p.Hv + oy) * p.Wv + ox;
                    unsigned char* orow = p.out_ptr + (m * p.out_ldc + p.out_coff) * sizeof(OutT);
                    const unsigned char* rrow = p.res_ptr ? p.res_ptr + (m * p.res_ldc + p.res_coff) * ES : nullptr;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n = nbase + j * 16;
                        if (n >= p.Cout) continue;
                        float v[4];
                        finish4(i, j, v);
                        if (p.vec_out && n + 4 <= p.Cout) {
                            if (rrow) {
                                if constexpr (ES == 2) {
                                    const uint2 rv = *(const uint2*)(rrow + (size_t)n * 2);
                                    v[0] += bf16_bits_to_f32(rv.x & 0xFFFFu); v[1] += bf16_bits_to_f32(rv.x >> 16);
                                    v[2] += bf16_bits_to_f32(rv.y & 0xFFFFu); v[3] += bf16_bits_to_f32(rv.y >> 16);
                                } else {
                                    const float4 rv = *(const float4*)(rrow + (size_t)n * 4);
                                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                                }
                            }
                            store4(orow + (size_t)n * sizeof(OutT), v);
                        } else {
                            for (int r = 0; r < 4 && n + r < p.Cout; ++r) {
                                float x = v[r];
                                if (rrow) x += load1<T>((const T*)rrow + n + r);
                                store1<OutT>((OutT*)orow + n + r, x);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        return full;
    };

    // ---- prologue: chunk 0 of the first tile and the weights of its first tap
    int cur_cc = 0;              // chunk being computed
    unsigned cur_tile = tile0;
    int pf_cc = 0;               // chunk the patch prefetch points at
    unsigned pf_tile = tile0;
    bool pf_s1;
    unsigned pf_soff;
    setup_tile(pf_tile);
    chunk_src(pf_cc, pf_s1, pf_soff);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) patch_piece(i, 0, pf_s1, pf_soff);
    if constexpr (DBL) {
        b_tile2(0, 0, b_koff(kbase(0), 0));
        b_tile2(0, 1, b_koff(kbase(0), 1));
    } else {
        b_tile(0, b_koff(kbase(0), 0));
    }
    // advance the prefetch cursor to chunk 1
    auto pf_advance = [&]() {
        if (++pf_cc == p.nchunks) {
            pf_cc = 0;
            pf_tile += tstride;
            setup_tile(pf_tile);
        }
        chunk_src(pf_cc, pf_s1, pf_soff);
    };
    pf_advance();
    WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    bool pending = false;
    unsigned pend_tile = 0;
    for (unsigned ci = 0; ci < total_chunks; ++ci) {
        const bool next_chunk = ci + 1 < total_chunks;
        const int cpar = (int)(ci & 1);
        const unsigned char* pt = patch0 + cpar * PATCH_BYTES;
        // weight offsets of this chunk and of the one after it (for the weights of its first tap)
        const int kb_cur = kbase(cur_cc);
        const int kb_next = kbase(cur_cc + 1 == p.nchunks ? 0 : cur_cc + 1);
        if constexpr (DBL) {
            PATCH_DSTEP(0) PATCH_DSTEP(1) PATCH_DSTEP(2) PATCH_DSTEP(3) PATCH_DSTEP(4)
        } else {
            PATCH_STEP(0) PATCH_STEP(1) PATCH_STEP(2) PATCH_STEP(3) PATCH_STEP(4) PATCH_STEP(5) PATCH_STEP(6) PATCH_STEP(7) PATCH_STEP(8)
        }
        if (next_chunk) pf_advance();
        if (++cur_cc == p.nchunks) {
            cur_cc = 0;
            pending = true;
            pend_tile = cur_tile;
            cur_tile += tstride;
        }
    }
    if (pending) epilogue(pend_tile);
}

template <typename T, typename OutT, int BN, int MC, int ACT, bool S2D, bool LIN = false>
int launch_patch_s2d(const PatchQ& p, hipStream_t st) {
    auto kernel = conv_patch_kernel<T, OutT, BN, MC, ACT, S2D, LIN>;
    constexpr int LDS = 2 * PATCH_BYTES + (BN == 64 ? 4 : 2) * BN * BKB;
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        CC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached[dev] = per_cu * cus;
    }
    const int gy = (p.Cout + BN - 1) / BN;
    const int gx = cc_grid_x(cached[dev], gy, p.ntiles);
    hipLaunchKernelGGL(kernel, dim3(gx * gy), dim3(256), LDS, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

template <typename T, typename OutT, int BN, int MC, int ACT>
int launch_patch_act(const PatchQ& p, hipStream_t st) {
    return p.s2d ? launch_patch_s2d<T, OutT, BN, MC, ACT, true>(p, st) : launch_patch_s2d<T, OutT, BN, MC, ACT, false>(p, st);
}

template <typename T, typename OutT, int BN, int MC>
int launch_patch_mc(const PatchQ& p, hipStream_t st) {
    if (p.act == CCNET_ACT_SILU) return launch_patch_act<T, OutT, BN, MC, CCNET_ACT_SILU>(p, st);
    if (p.act == CCNET_ACT_NONE) return launch_patch_act<T, OutT, BN, MC, CCNET_ACT_NONE>(p, st);
    return launch_patch_act<T, OutT, BN, MC, ACT_RUNTIME>(p, st);
}
template <typename T, typename OutT, int BN>
int launch_patch_bn(const PatchQ& p, hipStream_t st) {
    // linear geometry: only built for the bf16 network path (ccnet_conv2d_patch_try sets Wp for nothing else)
    if constexpr (std::is_same<T, __bf16>::value && std::is_same<OutT, __bf16>::value)
    if (p.Wp) {
        if (p.s2d) {
            if (p.act == CCNET_ACT_SILU) return launch_patch_s2d<T, OutT, BN, 16, CCNET_ACT_SILU, true, true>(p, st);
            return launch_patch_s2d<T, OutT, BN, 16, ACT_RUNTIME, true, true>(p, st);
        }
        if (p.act == CCNET_ACT_SILU) return launch_patch_s2d<T, OutT, BN, 16, CCNET_ACT_SILU, false, true>(p, st);
        if (p.act == CCNET_ACT_NONE) return launch_patch_s2d<T, OutT, BN, 16, CCNET_ACT_NONE, false, true>(p, st);
        return launch_patch_s2d<T, OutT, BN, 16, ACT_RUNTIME, false, true>(p, st);
    }
    return p.mc == 16 ? launch_patch_mc<T, OutT, BN, 16>(p, st) : launch_patch_mc<T, OutT, BN, 8>(p, st);
}

bool lin_enabled() {  // CCNET_NO_LINEAR_PATCH=1: maps whose width is no multiple of 8 stay on conv_igemm
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_NO_LINEAR_PATCH");
        v = (e && atoi(e) != 0) ? 0 : 1;
    }
    return v == 1;
}

bool patch_enabled() {  // CCNET_NO_PATCH_CONV=1 sends every conv to conv_igemm (tuning / bisecting aid)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_NO_PATCH_CONV");
        v = (e && atoi(e) != 0) ? 0 : 1;
    }
    return v == 1;
}

}  // namespace

// Returns 1 and fills `launch` when the descriptor is one this kernel takes, 0 when conv_igemm should run it.
// Called by ccnet_conv2d after it has validated the descriptor.
int ccnet_conv2d_patch_try(const ccnet_conv_desc* d, const double* span, int kpad, ccnet_launch_fn* launch) {
    if (!patch_enabled()) return 0;
    const int es = cc_esize(d->dtype), bk = BKB / es;
    const bool plain = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1;
    const bool s2d = d->kh == 6 && d->kw == 6 && d->stride == 2 && d->pad == 2 && d->nseg == 1 && (d->H % 2) == 0 && (d->W % 2) == 0;
    if (!plain && !s2d) return 0;
    if (d->Cout < 64) return 0;
    // widths that are no multiple of 8: the linear geometry, for narrow maps and the bf16 network path only
    const bool lin = (d->Wo % 8) != 0;
    if (lin && !(d->Wo <= 29 && d->dtype == CCNET_BF16 && d->out_dtype == CCNET_BF16 && lin_enabled())) return 0;
    const int mc = (d->Wo % 16) == 0 ? 16 : 8, th = 8 * (16 / mc);
    int ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->in[s].shift != 0 || span[s] >= 2147483648.0) return 0;
        if (!s2d && (d->in[s].C % bk) != 0) return 0;  // s2d: its own channel rule below
        ctot += d->in[s].C;
    }
    if (s2d) {
        const int c = d->in[0].C;
        if (!((c % bk) == 0 || 2 * c == bk)) return 0;
    }
    if ((long long)d->B * d->H * d->W >= (1ll << 31)) return 0;
    PatchQ p{};
    for (int s = 0; s < d->nseg; ++s) {
        const ccnet_slice& sl = d->in[s];
        p.seg[s] = SegQ{(const unsigned char*)sl.ptr + (size_t)sl.coff * es, (unsigned)(sl.ldc * es), 0, d->H, d->W, (unsigned)span[s]};
    }
    if (d->nseg == 1) p.seg[1] = p.seg[0];
    p.nseg = d->nseg;
    p.s2d = s2d ? 1 : 0;
    p.Creal = d->in[0].C;
    p.Ctot = s2d ? 4 * ctot : ctot;
    p.C0 = d->nseg == 2 ? d->in[0].C : p.Ctot;
    p.Hv = d->Ho; p.Wv = d->Wo; p.Hin = d->H; p.Win = d->W;
    p.Cout = d->Cout;
    p.mc = mc;
    p.xcd = cc_xcd_remap();
    p.tiles_x = (d->Wo + mc - 1) / mc;
    p.tiles_y = (d->Ho + th - 1) / th;
    p.ntiles = (unsigned)((long long)d->B * p.tiles_x * p.tiles_y);
    p.B = d->B;
    if (lin) {
        p.Wp = d->Wo + 2;
        p.Hp = d->Ho + 1;
        const long long q = ((long long)d->B * p.Hp + 1) * p.Wp;
        if (q >= (1ll << 30)) return 0;
        p.ntiles = (unsigned)((q + 127) / 128);
    }
    p.weight = (const unsigned char*)d->weight;
    p.Kpad = kpad;
    p.nchunks = p.Ctot / bk;
    p.bias = d->bias;
    p.act = d->act;
    const ccnet_slice& o = d->out;
    p.out_ptr = (unsigned char*)o.ptr; p.out_ldc = o.ldc; p.out_coff = o.coff;
    const int og = 16 / cc_esize(d->out_dtype);
    bool vec = (o.ldc % og) == 0 && (o.coff % og) == 0 && ((uintptr_t)o.ptr % 16) == 0 && (o.ldc % 8) == 0 && (o.coff % 8) == 0;
    if (d->residual.ptr) {
        const ccnet_slice& r = d->residual;
        p.res_ptr = (const unsigned char*)r.ptr; p.res_ldc = r.ldc; p.res_coff = r.coff;
        const int rg = 16 / es;
        vec = vec && (r.ldc % rg) == 0 && (r.coff % rg) == 0 && ((uintptr_t)r.ptr % 16) == 0 && (r.ldc % 8) == 0 && (r.coff % 8) == 0;
    }
    p.vec_out = vec ? 1 : 0;
    const int bn = d->Cout <= 64 ? 64 : 128;
    const int dt = d->dtype, odt = d->out_dtype;
    *launch = [p, bn, dt, odt](hipStream_t st, const float*) -> int {
        if (dt == CCNET_BF16) {
            if (odt == CCNET_BF16) return bn == 64 ? launch_patch_bn<__bf16, __bf16, 64>(p, st) : launch_patch_bn<__bf16, __bf16, 128>(p, st);
            return bn == 64 ? launch_patch_bn<__bf16, float, 64>(p, st) : launch_patch_bn<__bf16, float, 128>(p, st);
        }
        return bn == 64 ? launch_patch_bn<float, float, 64>(p, st) : launch_patch_bn<float, float, 128>(p, st);
    };
    return 1;
}
