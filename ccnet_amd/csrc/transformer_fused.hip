// Whole nn.Transformer(src, src) of MobileViT (model_block.py:270-271, 290) in ONE launch, bf16.
//
// The reference builds nn.Transformer with batch_first=False and feeds it (B, HW, d): the sequence axis is the image
// batch (S = B) and the H*W spatial positions are the "batch" (SURVEY fact 3).  Attention therefore only ever mixes the
// S tokens that share a spatial position, and every other op (linear, LayerNorm, residual) is per token: the 6 encoder
// + 6 decoder layers decompose into H*W INDEPENDENT problems of S <= 64 tokens x 256 channels.
//
// One workgroup of FOUR waves owns one position and needs 80 KB of LDS, so TWO workgroups share a CU: 400 positions (a
// 640x640 batch) are all resident at once on the 512 slots, and a workgroup's barrier / softmax / LayerNorm phases run
// beside the other workgroup's GEMMs (the previous version held 156 KB per position: one workgroup per CU, two rounds of
// positions, the matrix pipe idle whenever the one workgroup synchronised).  LDS per workgroup:
//   X  [64][256] bf16, 32 KB : the activation stream; pre-LayerNorm sums and LayerNorm results are written IN PLACE
//   A, B, C  [64][128] bf16, 16 KB each : Q_h / K_h / V_h of the head being processed; the head's attention output reuses A
//                                        (out_proj is accumulated per head); A|B holds an FFN hidden chunk; B|C stages the
//                                        encoder memory for the cross-attention K/V projections; C holds the LayerNorm table
// The encoder memory itself lives in global memory -- in the rows of the OUTPUT tensor, which nothing else touches until
// the final decoder.norm overwrites them.
//
// Weights are packed on the host into one stream PER WAVE, in the exact order that wave consumes them: fragment =
// 16 output rows x 32 k as the 64-lane A operand of v_mfma_f32_16x16x32_bf16 (1 KB, lane l: row l&15, k (l>>4)*8..+7).
// A wave keeps a ring of 16 fragments in registers and refills each slot as it is consumed.  The weights are the MFMA
// "A" operand, so a lane ends up with 4 consecutive output channels of one token and every epilogue is register-only.
//
// Attention core without a probability tile: scores are computed TRANSPOSED (S^T = K Q^T, keys on the accumulator rows), so
// a lane's 16 score registers are 16 keys of ONE query; after the softmax they ARE the B operand of O^T = V^T P^T under a
// permuted key order that the V fragments (transposing LDS reads) follow.  Rounding points (bf16) are those of the per-op
// path: projections, probabilities, attention output, pre-LayerNorm sums, LayerNorm outputs, the FFN hidden layer; all
// accumulation, softmax and LayerNorm statistics are fp32.
//
// Fixed geometry: d_model 256, 2 heads of 128, FF 1024 (the s-width model), S <= 64.  Other shapes use the per-op path.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int E = 256, HD = 128, FF = 1024, SM = 64;
#ifndef TF_RING
#define TF_RING 8
#endif
constexpr int RING = TF_RING;
constexpr int STREAM_PAD = 16;             // zero fragments behind each wave's stream (host layout, ops.py:pack_transformer)
static_assert(RING <= STREAM_PAD, "the ring prefetches RING fragments past the end of the stream");
constexpr int NW = 4;                    // waves per workgroup
constexpr int XB = SM * E * 2;           // 32768: a swizzled [64][256] bf16 GEMM operand (512-byte rows)
constexpr int HB = SM * HD * 2;          // 16384: a [64][128] bf16 tile (256-byte rows)
constexpr int OFF_X = 0, OFF_A = XB, OFF_B = XB + HB, OFF_C = XB + 2 * HB;
constexpr int LDS_BYTES = XB + 3 * HB;   // 81920: two workgroups fill a CU's 160 KB exactly

struct TfP {
    const unsigned char* src;  // tokens: row (s*N + n), E bf16 each, row stride src_ld elements
    unsigned char* out;
    int src_ld, out_ld, S, N;
    const unsigned char* wstream;  // [NW waves][frags_per_wave + 16][64 lanes][16 B]
    long long frags_per_wave;
    const float* params;
    int n_enc, n_dec;
    float scale;
};

// byte offset of (row, 8-element chunk) in a swizzled [64][256] operand: conflict-free ds_read_b128 B-operand fragments
__device__ __forceinline__ int xoff(int row, int chunk) { return row * 512 + (((chunk & 16) | ((chunk ^ row) & 15)) << 4); }
// the same for a [64][128] tile (16 chunks per row)
__device__ __forceinline__ int qoff(int row, int chunk) { return row * 256 + (((chunk ^ row) & 15) << 4); }
// V tile [key][128]: 32-byte column blocks XOR-ed with the row, so that the 4-row x 32-byte groups of ds_read_b64_tr_b16
// (and the two such groups of a 32-lane half) fall on distinct banks
__device__ __forceinline__ int voff(int row, int col) { return row * 256 + ((((col >> 4) ^ row) & 7) << 5) + (col & 15) * 2; }

template <int NTW>
__device__ __forceinline__ void zero(f32x4 (&acc)[4][NTW]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc[mt][j] += W(16 rows of n-tile j) . src(tokens of m-tile mt) over KT k-tiles of 32, consuming NTW*KT fragments of the
// wave's stream (k-tile major, n-tile minor).  src is a swizzled operand with ROWB-byte rows; lk[kt & 3] = this lane's byte
// offset of chunk (4*(kt&3) + fq) in row fr, the other address terms are immediates.  Every call consumes a multiple of RING
// fragments, so the ring slot of each fragment is a compile-time constant.
template <int NTW, int KT, int ROWB>
__device__ __forceinline__ void mm(f32x4 (&acc)[4][NTW], const unsigned char* src, u32x4 (&ring)[RING], const unsigned char*& wp, int loff,
                                   const int (&lk)[4]) {
    static_assert((NTW * KT) % RING == 0, "a GEMM call must consume whole ring turns");
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        u32x4 xb[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xb[mt] = *(const u32x4*)(src + lk[kt & 3] + mt * 16 * ROWB + (kt >> 2) * 256);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int slot = (kt * NTW + j) % RING;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot]), __builtin_bit_cast(bf16x8, xb[mt]), acc[mt][j], 0, 0, 0);
            ring[slot] = *(const u32x4*)(wp + RING * 1024 + loff);  // refill with the fragment RING ahead (wp is wave uniform)
            wp += 1024;
        }
        // bound how far the scheduler hoists the operand reads of later k-tiles
        if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// The same with a 16-deep ring: slots 0..7 in `ring`, 8..15 in `hi` (the FFN's look-ahead; see ffn()).  Every call consumes a
// multiple of 16 fragments.
#ifndef TF_FFN_RING16
#define TF_FFN_RING16 1
#endif
template <int NTW, int KT, int ROWB>
__device__ __forceinline__ void mm16(f32x4 (&acc)[4][NTW], const unsigned char* src, u32x4 (&ring)[RING], u32x4 (&hi)[8], const unsigned char*& wp,
                                     int loff, const int (&lk)[4]) {
    static_assert(RING == 8 && (NTW * KT) % 16 == 0, "a GEMM call must consume whole turns of the 16-deep ring");
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        u32x4 xb[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xb[mt] = *(const u32x4*)(src + lk[kt & 3] + mt * 16 * ROWB + (kt >> 2) * 256);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int slot = (kt * NTW + j) % 16;
            u32x4& w = slot < 8 ? ring[slot & 7] : hi[slot & 7];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, xb[mt]), acc[mt][j], 0, 0, 0);
            w = *(const u32x4*)(wp + 16 * 1024 + loff);
            wp += 1024;
        }
        if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// An opaque copy of a lane-derived value.  The swizzled LDS addresses of the epilogues are invariant across the layer loops;
// left alone, the compiler hoists dozens of them out of those loops and then spills them (hundreds of bytes of scratch per
// lane).  Recomputing them per call from a laundered lane index costs a few VALU instructions beside thousands of MFMAs.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint2 pack4(const float* v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }

// A scalar broadcast into BOTH registers of a real 64-bit pair, opaque to the compiler.  Left alone, LLVM broadcasts a scalar into a
// packed-fp32 op with op_sel; when the scalar sits in the odd register of a pair that is `op_sel:[0,1]` (low lane reads the high
// dword), and that form reads ZERO in lanes 48..63 while another wave's MFMA runs on the SIMD (scripts/hazard/ln_chain_min.hip,
// scripts/check_isa.py): LayerNorm's x - mean then silently becomes x - 0 for tokens 48..63.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pair2(float v) {
    f32x2 p = {v, v};
    asm volatile("" : "+v"(p));
    return p;
}

// workgroup barrier that waits for this wave's LDS traffic only: the weight ring's global loads stay in flight across it
#define TSYNC() __syncthreads()

// Four waves.  Wave w owns output columns [64w, 64w+64) of every 256-wide GEMM, columns [32w, 32w+32) of each per-head
// Q/K/V projection, and query tile w (16 queries) of the attention core.
#ifndef TF_PAIR
#define TF_PAIR 0
#endif
// TF_PAIR=1 (experiment, off): a workgroup is EIGHT waves = two positions side by side (waves 0-3 and 4-7, 80 KB of LDS each,
// one workgroup per CU).  Nothing is shared between the halves but the barriers: waves w and w + 4 consume the same weight
// stream in step, in the hope that the second request for a fragment meets the first one's line in the CU's L1.  Measured on
// one box: 0.889 ms against 0.884 ms for two independent workgroups per CU -- no gain, so the default stays unpaired.
__global__ __launch_bounds__(NW * 64 * (TF_PAIR ? 2 : 1), TF_PAIR ? 1 : 2) void transformer_fused_kernel(const TfP p) {
    extern __shared__ __align__(16) unsigned char sm_all[];
    const int half = TF_PAIR ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    unsigned char* sm = sm_all + half * LDS_BYTES;
    const int t = threadIdx.x & 255, lane = t & 63, lane0 = lane;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4, fr0 = fr, fq0 = fq;
    // an odd position count leaves the last workgroup's second half without a position: it repeats the first half's (same
    // values to the same addresses) so that it takes part in every barrier
    const int n = TF_PAIR ? min((int)blockIdx.x * 2 + half, p.N - 1) : (int)blockIdx.x;
    unsigned char* X = sm + OFF_X;
    unsigned char* RA = sm + OFF_A;
    unsigned char* RB = sm + OFF_B;
    unsigned char* RC = sm + OFF_C;
    int lx[4], lq[4];  // B-operand chunk offsets of this lane in a 512-byte-row / 256-byte-row operand
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        lx[k4] = fr * 512 + ((((k4 * 4) ^ fr ^ fq) & 15) << 4);
        lq[k4] = fr * 256 + ((((k4 * 4 + fq) ^ fr) & 15) << 4);
    }

    // ---- weight ring
    const unsigned char* wp = p.wstream + (size_t)wave * (size_t)(p.frags_per_wave + STREAM_PAD) * 1024;  // wave uniform
    const int loff = lane * 16;
    u32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = *(const u32x4*)(wp + i * 1024 + loff);
    const float* pp = p.params;
    // rows of p.out as a buffer resource (the encoder memory is read back through it; checked < 2 GiB on the host)
    const __amdgpu_buffer_rsrc_t mem_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(((size_t)p.S * p.N - 1) * p.out_ld * 2 + E * 2), 0x00020000);

    // tokens of this position -> X (rows >= S are zero)
    auto load_src = [&]() __attribute__((always_inline)) {
        for (int e = t; e < SM * 32; e += NW * 64) {
            const int row = e >> 5, ch = e & 31;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (row < p.S) v = *(const u32x4*)(p.src + ((size_t)((size_t)row * p.N + n) * p.src_ld + ch * 8) * 2);
            *(u32x4*)(X + xoff(row, ch)) = v;
        }
    };

    // LayerNorm of the 64 rows of X, in place (and optionally to the global output).  Lane = row, wave = 64-channel slice:
    // each lane sums its slice serially and the four slices of a row meet through a 2 KB table in region C -- no cross-lane
    // shuffles.  Statistics in fp32, one pass (sum, sum of squares).  A wave's slice is exactly the columns its own GEMM
    // epilogue wrote, so nothing but the wave's own LDS traffic has to be waited for before phase 1.
    float2* lnpart = (float2*)RC;  // [64 rows][4 slices]
    auto layernorm = [&](const float* gb, bool to_lds, bool to_global) __attribute__((always_inline)) {
        const int row = opaque(lane);
        float v[64];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32x4 raw = *(const u32x4*)(X + xoff(row, 8 * wave + q));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * q + 2 * e] = bf16_bits_to_f32(raw[e] & 0xFFFFu);
                v[8 * q + 2 * e + 1] = bf16_bits_to_f32(raw[e] >> 16);
            }
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 64; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
        lnpart[row * NW + wave] = make_float2(s1, s2);
        TSYNC();
        const float4 p01 = *(const float4*)(&lnpart[row * NW]);
        const float4 p23 = *(const float4*)(&lnpart[row * NW + 2]);
        const float t1 = (p01.x + p01.z) + (p23.x + p23.z);
        const float t2 = (p01.y + p01.w) + (p23.y + p23.w);
        const float mean = t1 * (1.0f / E);
        const float var = fmaxf(t2 * (1.0f / E) - mean * mean, 0.0f);
        const float rs = rsqrtf(var + 1e-5f);
        const float* g = gb + 64 * wave;       // wave-uniform
        const float* bt = gb + E + 64 * wave;
        const f32x2 mean2 = pair2(mean), rs2 = pair2(rs);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 8 * q + 2 * e;
                const f32x2 y = (f32x2{v[c], v[c + 1]} - mean2) * rs2 * f32x2{g[c], g[c + 1]} + f32x2{bt[c], bt[c + 1]};
                pk[e] = pack_bf16x2(y.x, y.y);
            }
            if (to_lds) *(u32x4*)(X + xoff(row, 8 * wave + q)) = pk;
            if (to_global && row < p.S) *(u32x4*)(p.out + ((size_t)((size_t)row * p.N + n) * p.out_ld + 64 * wave + 8 * q) * 2) = pk;
        }
    };
    auto add_ln = [&](bool to_lds, bool to_global) __attribute__((always_inline)) {
        layernorm(pp, to_lds, to_global);
        pp += 2 * E;
        TSYNC();
    };

    // acc (+ bias + residual from X) -> X in place (each element is read and written by the same lane)
    auto to_z = [&](f32x4 (&acc)[4][4], const float4 (&bias4)[4]) __attribute__((always_inline)) {
        const int fr = opaque(fr0), fq = opaque(fq0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = wave * 64 + j * 16 + fq * 4;
            const float4 bv = bias4[j];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = mt * 16 + fr;
                unsigned char* px = X + xoff(row, n0 >> 3) + (n0 & 4) * 2;
                const uint2 rx = *(const uint2*)px;
                float v[4] = {acc[mt][j][0] + bv.x + bf16_bits_to_f32(rx.x & 0xFFFFu), acc[mt][j][1] + bv.y + bf16_bits_to_f32(rx.x >> 16),
                              acc[mt][j][2] + bv.z + bf16_bits_to_f32(rx.y & 0xFFFFu), acc[mt][j][3] + bv.w + bf16_bits_to_f32(rx.y >> 16)};
                *(uint2*)px = pack4(v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LayerNorm that follows reads this wave's own columns
    };

    // multi-head attention block.  Queries from X; keys / values from X (self) or from the encoder memory kept in the rows of
    // p.out (cross).  Result (+ bias + X residual) lands in X.
    auto attention = [&](bool cross) __attribute__((always_inline)) {
        const int fr = opaque(fr0), fq = opaque(fq0), lane = opaque(lane0);
        const float* in_bias = pp;         // q | k | v, 3E
        const float* out_bias = pp + 3 * E;
        pp += 4 * E;
        uint2 ao0[8];  // head 0's attention output of this lane (bf16), parked in registers while head 1 uses regions A, B, C
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // encoder memory of this position -> B|C by LDS-DMA, requested now and awaited after the Q projection.  One wave
            // instruction fills 1 KB = two 512-byte rows in lane order, so the operand's swizzle goes on the SOURCE address:
            // the lane at physical chunk c' of row r fetches logical chunk (c' & 16) | ((c' ^ r) & 15).  Rows >= S lie beyond the
            // tensor: the buffer resource's range check returns zeros for them.
            if (cross) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = (wave * 8 + k) * 2 + (lane >> 5), pc = lane & 31;
                    const int ch = (pc & 16) | ((pc ^ row) & 15);
                    const unsigned voffset = row < p.S ? (unsigned)(((size_t)row * p.N + n) * p.out_ld + ch * 8) * 2u : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(mem_rsrc, (__attribute__((address_space(3))) void*)(RB + (wave * 8 + k) * 1024), 16,
                                                             (int)voffset, 0, 0, 0);
                }
            }
            // projections of this head: wave w produces columns [32w, 32w+32) of Q_h, K_h and V_h
            const int c0 = wave * 32 + fq * 4;  // column inside the head (n-tile j adds 16)
            // biases are fetched BEFORE the GEMM they follow: vector-memory results return in issue order, so a load issued
            // after it would sit behind the 16 weight fragments the ring has in flight
            auto ld_bias = [&](int part, float4 (&bv2)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bv2[j] = *(const float4*)(in_bias + part * E + h * HD + c0 + j * 16);
            };
            auto put_qk = [&](unsigned char* dst, f32x4 (&acc)[4][2], const float4 (&bv2)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 bv = bv2[j];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float v[4] = {acc[mt][j][0] + bv.x, acc[mt][j][1] + bv.y, acc[mt][j][2] + bv.z, acc[mt][j][3] + bv.w};
                        const int row = mt * 16 + fr, col = c0 + j * 16;
                        *(uint2*)(dst + qoff(row, col >> 3) + (col & 4) * 2) = pack4(v);
                    }
                }
            };
            auto put_v = [&](f32x4 (&acc)[4][2], const float4 (&bv2)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 bv = bv2[j];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float v[4] = {acc[mt][j][0] + bv.x, acc[mt][j][1] + bv.y, acc[mt][j][2] + bv.z, acc[mt][j][3] + bv.w};
                        *(uint2*)(RC + voff(mt * 16 + fr, c0 + j * 16)) = pack4(v);
                    }
                }
            };
            float4 bq[2], bk[2], bvv[2];
            {
                ld_bias(0, bq);
                f32x4 acc[4][2];
                zero<2>(acc);
                mm<2, 8, 512>(acc, X, ring, wp, loff, lx);
                put_qk(RA, acc, bq);
            }
            ld_bias(1, bk);
            if (cross) {
                // every vector-memory operation of this wave so far, the 8 DMA pieces included.  (A counted wait -- vmcnt(RING),
                // "the Q projection issued 16 ring refills behind the DMA" -- held the ring's newest loads in flight, but it is only
                // right while the compiler leaves at least RING vector-memory instructions between the DMA and this point; the
                // spill reloads around here drain the counter anyway, so the exact form costs nothing.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TSYNC();  // the staged memory is complete
                f32x4 acck[4][2], accv[4][2];
                zero<2>(acck);
                zero<2>(accv);
                mm<2, 8, 512>(acck, RB, ring, wp, loff, lx);
                ld_bias(2, bvv);
                mm<2, 8, 512>(accv, RB, ring, wp, loff, lx);
                TSYNC();  // every wave has read the staged memory: K_h and V_h may overwrite it
                put_qk(RB, acck, bk);
                put_v(accv, bvv);
            } else {
                f32x4 acc[4][2];
                zero<2>(acc);
                mm<2, 8, 512>(acc, X, ring, wp, loff, lx);
                put_qk(RB, acc, bk);
                ld_bias(2, bvv);
                zero<2>(acc);
                mm<2, 8, 512>(acc, X, ring, wp, loff, lx);
                put_v(acc, bvv);
            }
            TSYNC();
            // transposed scores of query tile `wave`: s[i][r] = K[key i*16 + fq*4 + r] . Q[query wave*16 + fr]
            f32x4 s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                const u32x4 qf = *(const u32x4*)(RA + wave * 16 * 256 + lq[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 kf = *(const u32x4*)(RB + i * 16 * 256 + lq[ks]);
                    s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf), s[i], 0, 0, 0);
                }
            }
            // softmax over the keys of this lane's query: 16 registers, then the four lanes that share fr; keys >= S are masked
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (i * 16 + fq * 4 + r < p.S) ? s[i][r] * p.scale : -INFINITY;
                    s[i][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // the probabilities are rounded to bf16 where the per-op path rounds them (the P operand of PV); the
                    // normalising sum is taken over the fp32 values, as there
                    const float pv = __expf(s[i][r] - mx);
                    s[i][r] = pv;
                    sum += pv;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            TSYNC();  // every wave is done with Q_h and K_h
            // O^T = V^T P^T.  k-step ks covers keys {(2ks + (j>>2))*16 + fq*4 + (j&3)}: exactly the lane's registers s[2ks][*], s[2ks+1][*]
            f32x4 o[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
            // transposing reads: lane 4q+p of each 16-lane group addresses key row (base + q), d columns 4p..4p+3.  The row's
            // low three bits do not depend on ks or on lo / hi, so the swizzled block of d-tile d is one per-lane constant
            const int krow = fq * 4 + (fr >> 2);
            const unsigned char* vbase = RC + krow * 256 + (fr & 3) * 8;
            int vblk[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) vblk[d] = ((d ^ krow) & 7) << 5;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 pf;
                pf[0] = pack_bf16x2(s[2 * ks][0], s[2 * ks][1]);
                pf[1] = pack_bf16x2(s[2 * ks][2], s[2 * ks][3]);
                pf[2] = pack_bf16x2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
                pf[3] = pack_bf16x2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const unsigned char* pv = vbase + vblk[d] + (2 * ks) * 16 * 256;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pv);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pv + 16 * 256));
                    const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), o[d], 0, 0, 0);
                    if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // O (query wave*16 + fr, channels h*128 + d*16 + fq*4 + r): head 0 waits in registers, then both heads go to A|B as
            // the [64][256] B operand of out_proj
            if (h == 0) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float v[4] = {o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
                    ao0[d] = pack4(v);
                }
                TSYNC();  // every wave is done with V_0: the next head may overwrite regions A, B, C
            } else {
                TSYNC();  // every wave is done with V_1 (and, since the barrier after the scores, with Q_1 and K_1)
                const int row = wave * 16 + fr;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float v[4] = {o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
                    const int c0h = d * 16 + fq * 4;
                    *(uint2*)(RA + xoff(row, c0h >> 3) + (c0h & 4) * 2) = ao0[d];
                    *(uint2*)(RA + xoff(row, (HD + c0h) >> 3) + (c0h & 4) * 2) = pack4(v);
                }
                TSYNC();  // the attention output is complete
            }
        }
        f32x4 zacc[4][4];
        zero<4>(zacc);
        mm<4, 8, 512>(zacc, RA, ring, wp, loff, lx);
        float4 ob[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[j] = *(const float4*)(out_bias + wave * 64 + j * 16 + fq * 4);
        to_z(zacc, ob);
    };

    // FFN in eight chunks of 128 hidden channels: linear1 chunk (wave w owns 32 of its columns) -> ReLU -> [64][128] tile in
    // region A or B (alternating, so one barrier per chunk is enough: a wave that has passed chunk c's barrier has finished
    // reading the tile of chunk c-1, which is the buffer chunk c+1 writes) -> that K slice of linear2 into the 64-column
    // accumulator the wave keeps across the chunks.  (Chunks of 256 held 64 + 64 accumulator registers beside the ring and
    // spilled ring slots; the k order of linear2 is the same, results are bit-identical.)
    auto ffn = [&]() __attribute__((always_inline)) {
        const int fr = opaque(fr0), fq = opaque(fq0);
        const float* b1 = pp;
        const float* b2 = pp + FF;
        pp += FF + E;
        f32x4 acc2[4][4];
        zero<4>(acc2);
        // Inside the FFN (two thirds of the kernel's MFMAs, and the part with registers to spare since it runs in 128-wide chunks) the
        // weight ring is 16 deep: eight more fragments are requested here, every refill looks 16 ahead, and at the end the upper
        // half is simply dropped -- the stream position is a multiple of 16 fragments later, the lower half holds what an 8-deep
        // ring would hold, and the next phase requests the upper half's fragments again (8 KB per wave and FFN).
        u32x4 hi[8];
        if (TF_FFN_RING16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) hi[i] = *(const u32x4*)(wp + (8 + i) * 1024 + loff);
        }
#pragma unroll 2
        for (int c = 0; c < FF / 128; ++c) {
            unsigned char* HT = (c & 1) ? RB : RA;
            float4 hb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) hb[j] = *(const float4*)(b1 + c * 128 + wave * 32 + j * 16 + fq * 4);
            f32x4 acc1[4][2];
            zero<2>(acc1);
            if (TF_FFN_RING16) mm16<2, 8, 512>(acc1, X, ring, hi, wp, loff, lx);
            else mm<2, 8, 512>(acc1, X, ring, wp, loff, lx);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = wave * 32 + j * 16 + fq * 4;  // column inside the chunk
                const float4 bv = hb[j];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4] = {fmaxf(acc1[mt][j][0] + bv.x, 0.f), fmaxf(acc1[mt][j][1] + bv.y, 0.f), fmaxf(acc1[mt][j][2] + bv.z, 0.f),
                                  fmaxf(acc1[mt][j][3] + bv.w, 0.f)};
                    *(uint2*)(HT + qoff(mt * 16 + fr, n0 >> 3) + (n0 & 4) * 2) = pack4(v);
                }
            }
            TSYNC();
            if (TF_FFN_RING16) mm16<4, 4, 256>(acc2, HT, ring, hi, wp, loff, lq);
            else mm<4, 4, 256>(acc2, HT, ring, wp, loff, lq);
        }
        TSYNC();  // region A / B (and the LayerNorm table in C) are free again
        float4 b2v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b2v[j] = *(const float4*)(b2 + wave * 64 + j * 16 + fq * 4);
        to_z(acc2, b2v);
    };

    load_src();
    TSYNC();
    for (int l = 0; l < p.n_enc; ++l) {
        attention(false);
        add_ln(true, false);
        ffn();
        add_ln(true, false);
    }
    add_ln(false, true);  // encoder.norm: the memory goes to this position's rows of p.out (global), not to LDS
    // the memory rows are read back by other waves of this workgroup: make the stores visible (they leave through L2) first
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the ring was drained by that wait: nothing is lost, its slots are simply complete
    load_src();  // nn.Transformer(src, src): the decoder starts from the same tokens
    TSYNC();
    for (int l = 0; l < p.n_dec; ++l) {
        attention(false);
        add_ln(true, false);
        attention(true);
        add_ln(true, false);
        ffn();
        add_ln(true, false);
    }
    add_ln(false, true);  // decoder.norm: rows go to the output (overwriting the memory rows, which are no longer needed)
}

}  // namespace

extern "C" int ccnet_transformer_fused_frags(int n_enc, int n_dec, long long* frags_per_wave, long long* param_floats) {
    CC_REQUIRE(n_enc >= 0 && n_dec >= 0 && frags_per_wave && param_floats, "transformer_fused_frags: bad arguments");
    // per wave (4 waves): attention = 2 heads x (3 projections x 16 + out_proj slice 16), FFN = 8 chunks x (16 + 16)
    const long long attn = 2 * (3 * 16 + 16), ffn = (FF / 128) * 32;
    *frags_per_wave = (long long)n_enc * (attn + ffn) + (long long)n_dec * (2 * attn + ffn);
    const long long pa = 4 * E, pl = 2 * E, pf = FF + E;
    *param_floats = (long long)n_enc * (pa + pl + pf + pl) + pl + (long long)n_dec * (2 * (pa + pl) + pf + pl) + pl;
    return CCNET_OK;
}

extern "C" int ccnet_transformer_fused(const void* src, int src_ld, void* out, int out_ld, int S, int N, int d_model, int nhead, int dim_ff,
                                       const void* wstream, long long frags_per_wave, const float* params, int n_enc, int n_dec,
                                       ccnet_stream stream) {
    CC_REQUIRE(src && out && wstream && params, "transformer_fused: null pointer");
    CC_REQUIRE(d_model == E && nhead == 2 && dim_ff == FF, "transformer_fused: built for d_model 256, 2 heads, FF 1024 (got %d, %d, %d)", d_model,
               nhead, dim_ff);
    CC_REQUIRE(S >= 1 && S <= SM && N >= 1, "transformer_fused: 1 <= S <= 64 tokens per position (got S=%d N=%d)", S, N);
    CC_REQUIRE(src_ld >= E && out_ld >= E && (src_ld % 8) == 0 && (out_ld % 8) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                   ((uintptr_t)wstream % 16) == 0 && ((uintptr_t)params % 16) == 0, "transformer_fused: rows must be 16-byte aligned");
    CC_REQUIRE((double)S * N * out_ld * 2.0 < 2147483648.0, "transformer_fused: output tensor of 2 GiB or more");
    CC_REQUIRE(src != out, "transformer_fused: src and out must be different buffers (out also holds the encoder memory while the kernel runs)");
    long long need_f = 0, need_p = 0;
    ccnet_transformer_fused_frags(n_enc, n_dec, &need_f, &need_p);
    CC_REQUIRE(frags_per_wave == need_f, "transformer_fused: weight stream has %lld fragments per wave, expected %lld", frags_per_wave, need_f);
    TfP p{(const unsigned char*)src, (unsigned char*)out, src_ld, out_ld, S, N, (const unsigned char*)wstream, frags_per_wave, params, n_enc, n_dec,
          1.0f / sqrtf((float)HD)};
    const double rows = (double)S * N;
    const double fl_layer_attn = 2.0 * rows * E * (4.0 * E) + 4.0 * rows * S * E;  // projections + QK^T + PV
    const double fl_ffn = 4.0 * rows * E * FF;
    return ccnet_submit(
        [p](hipStream_t st, const float*) -> int {
            static bool attr_set[16] = {false};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
            if (!attr_set[dev]) {
                CC_HIP(hipFuncSetAttribute((const void*)transformer_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDS_BYTES * (TF_PAIR ? 2 : 1)));
                attr_set[dev] = true;
            }
            if (TF_PAIR)
                hipLaunchKernelGGL(transformer_fused_kernel, dim3((p.N + 1) / 2), dim3(NW * 128), LDS_BYTES * 2, st, p);
            else
                hipLaunchKernelGGL(transformer_fused_kernel, dim3(p.N), dim3(NW * 64), LDS_BYTES, st, p);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_TRANSFORMER, n_enc * (fl_layer_attn + fl_ffn) + n_dec * (2 * fl_layer_attn + fl_ffn),
        2.0 * rows * E * 2 + (double)need_f * NW * 1024);
}
