// Whole nn.Transformer(src, src) of MobileViT (model_block.py:270-271, 290) in ONE launch, bf16.
//
// The reference builds nn.Transformer with batch_first=False and feeds it (B, HW, d): the sequence axis is the image
// batch (S = B) and the H*W spatial positions are the "batch" (SURVEY fact 3).  Attention therefore only ever mixes the
// S tokens that share a spatial position, and every other op (linear, LayerNorm, residual) is per token: the 6 encoder
// + 6 decoder layers decompose into H*W INDEPENDENT problems of S <= 64 tokens x 256 channels.  One workgroup (8 waves)
// owns one position, keeps its 64 x 256 activation stream, the encoder memory, the attention output and one head's
// Q/K/V in LDS (156 KB of the CU's 160 KB) and runs all twelve layers without touching HBM for activations; what it
// streams is the weights (22 MB bf16, L2/Infinity-Cache resident, read by every workgroup).
//
// Weights are packed on the host into one stream PER WAVE, in the exact order that wave consumes them: fragment =
// 16 output rows x 32 k as the 64-lane A operand of v_mfma_f32_16x16x32_bf16 (1 KB, lane l: row l&15, k (l>>4)*8..+7).
// A wave keeps a ring of 16 fragments in registers and refills each slot as it is consumed (128 KB per CU in flight).
// The weights are the MFMA "A" operand, so a lane ends up with 4 consecutive output channels of one token.
// Rounding points (bf16) are those of the per-op path: projections, attention output, pre-LayerNorm sums, LayerNorm
// outputs, the FFN hidden layer; all accumulation, softmax and LayerNorm statistics are fp32.
//
// Fixed geometry: d_model 256, 2 heads of 128, FF 1024 (the s-width model), S <= 64.  Other shapes use the per-op path.
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int E = 256, HD = 128, FF = 1024, SM = 64;
constexpr int RING = 16;
constexpr int NW = 8;                   // waves per workgroup
constexpr int XB = SM * E * 2;           // 32768: a swizzled [64][256] bf16 GEMM operand
constexpr int QLD = HD + 8;              // padded row of the per-head Q/K/V tiles (elements)
constexpr int QB = SM * QLD * 2;         // 17408
constexpr int ZLD = E + 8;               // padded row of the pre-LayerNorm tile (elements)
constexpr int PLD = 72;
constexpr int OFF_X = 0, OFF_MEM = XB, OFF_AO = 2 * XB, OFF_T = 3 * XB;  // T: Q|K|V of one head, or hidden chunk, or Z
constexpr int T_BYTES = 3 * QB;          // 52224 >= 64*ZLD*2 = 33792 and >= XB
constexpr int OFF_P = OFF_T + T_BYTES;
constexpr int LDS_BYTES = OFF_P + 4 * 16 * PLD * 2;  // 159744

struct TfP {
    const unsigned char* src;  // tokens: row (s*N + n), E bf16 each, row stride src_ld elements
    unsigned char* out;
    int src_ld, out_ld, S, N;
    const unsigned char* wstream;  // [NW waves][frags_per_wave + RING][64 lanes][16 B]
    long long frags_per_wave;
    const float* params;
    int n_enc, n_dec;
    float scale;
};

// byte offset of (row, 8-element chunk) in a swizzled [64][256] operand: conflict-free ds_read_b128 B-operand fragments
__device__ __forceinline__ int xoff(int row, int chunk) { return row * 512 + (((chunk & 16) | ((chunk ^ row) & 15)) << 4); }

// acc[mt][j] += W(16 rows of n-tile j) . src(tokens of m-tile mt), K = 256, consuming NTW*8 fragments of the wave's stream.
// lx[k4] = this lane's byte offset of B-operand chunk (4*k4 + fq) in row fr of a swizzled operand (see xoff): the other
// address terms are immediates.  `start` = ring slot of the first fragment; a compile-time constant after unrolling.
template <int NTW>
__device__ __forceinline__ void mm256(f32x4 (&acc)[4][NTW], const unsigned char* src, u32x4 (&ring)[RING], const unsigned char*& wp,
                                      int loff, const int (&lx)[4], int start = 0) {
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        u32x4 xb[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xb[mt] = *(const u32x4*)(src + lx[kt & 3] + mt * 8192 + (kt >> 2) * 256);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int slot = (start + kt * NTW + j) % RING;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot]), __builtin_bit_cast(bf16x8, xb[mt]), acc[mt][j], 0, 0, 0);
            ring[slot] = *(const u32x4*)(wp + RING * 1024 + loff);  // refill with the fragment RING ahead (wp is wave uniform)
#ifndef TF_NOADVANCE  // perf ablation: re-read the same fragments (cache hits) instead of streaming
            wp += 1024;
#endif
        }
    }
}

template <int NTW>
__device__ __forceinline__ void zero(f32x4 (&acc)[4][NTW]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

#ifdef TF_STAMP  // diagnostic build: cycles per phase (0 GEMM, 1 epilogue, 2 barrier, 3 attention core, 4 LayerNorm)
__device__ unsigned long long g_tf_stamps[8];
#define TS(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_last; st_last = n_; } while (0)
#define TPH(i) st_ph = (i)
#define TSYNC() do { TS(st_ph); __syncthreads(); TS(2); } while (0)
#else
#define TS(i)
#define TPH(i)
#define TSYNC() __syncthreads()
#endif

// all-reduce over the 16 lanes of a DPP row (the lanes that share fq): xor-1 and xor-2 inside quads, then the two mirror
// steps -- four DPP-modified VALU instructions instead of four ds_bpermute round trips
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141>(v));  // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}

__device__ __forceinline__ uint2 pack4(const float* v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }

// 8 waves (two per SIMD, so that one wave's LDS / VALU / barrier time hides under the other's MFMAs).  Wave w owns
// output columns [32w, 32w+32) of every 256-wide GEMM and columns [16w, 16w+16) of each per-head Q/K/V projection.
__global__ __launch_bounds__(NW * 64, 2) void transformer_fused_kernel(const TfP p) {
    extern __shared__ __align__(16) unsigned char sm[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n = blockIdx.x;
    unsigned char* X = sm + OFF_X;
    unsigned char* MEM = sm + OFF_MEM;
    unsigned char* AO = sm + OFF_AO;
    unsigned char* Tb = sm + OFF_T;
    const int qt = wave & 3, half = wave >> 2;  // attention core: query tile and half of the head's columns
    unsigned short* Pw = (unsigned short*)(sm + OFF_P) + qt * 16 * PLD;
    int lx[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) lx[k4] = fr * 512 + ((((k4 * 4) ^ fr ^ fq) & 15) << 4);

    // ---- weight ring
    const unsigned char* wp = p.wstream + (size_t)wave * (size_t)(p.frags_per_wave + RING) * 1024;  // wave uniform
    const int loff = lane * 16;
    u32x4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = *(const u32x4*)(wp + i * 1024 + loff);
    const float* pp = p.params;
#ifdef TF_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    int st_ph = 1;
#endif

    // tokens of this position -> X (rows >= S are zero)
    auto load_src = [&]() __attribute__((always_inline)) {
        for (int e = t; e < SM * 32; e += NW * 64) {
            const int row = e >> 5, ch = e & 31;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (row < p.S) v = *(const u32x4*)(p.src + ((size_t)((size_t)row * p.N + n) * p.src_ld + ch * 8) * 2);
            *(u32x4*)(X + xoff(row, ch)) = v;
        }
    };

    // LayerNorm of the 64 rows of Z (pre-LayerNorm sums, padded rows in T) or of X itself -> dst (swizzled operand) and
    // optionally the global output.  Lane = row, wave = 32-channel slice: each lane sums its slice serially, the 8 slices
    // of a row meet through a 4 KB LDS table (one barrier) -- no cross-lane shuffles (a 64-lane butterfly is 6 dependent
    // ds_bpermute per reduction and made LayerNorm a quarter of the kernel).  Statistics in fp32, one pass (sum, sum of
    // squares).
    float2* lnpart = (float2*)(sm + OFF_P);  // [64 rows][8 slices]; the P tiles are idle outside the attention core
    auto layernorm = [&](unsigned char* dst, const float* gb, bool from_x, bool to_global) __attribute__((always_inline)) {
        TPH(4);
        const int row = lane;
        float v[32];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 raw = from_x ? *(const u32x4*)(X + xoff(row, 4 * wave + q)) : *(const u32x4*)(Tb + (row * ZLD + 32 * wave + 8 * q) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[8 * q + 2 * e] = bf16_bits_to_f32(raw[e] & 0xFFFFu);
                v[8 * q + 2 * e + 1] = bf16_bits_to_f32(raw[e] >> 16);
            }
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 32; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
        lnpart[row * NW + wave] = make_float2(s1, s2);
        TSYNC();
        TPH(4);
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w += 2) {
            const float4 pr = *(const float4*)(&lnpart[row * NW + w]);
            t1 += pr.x + pr.z;
            t2 += pr.y + pr.w;
        }
        const float mean = t1 * (1.0f / E);
        const float var = fmaxf(t2 * (1.0f / E) - mean * mean, 0.0f);
        const float rs = rsqrtf(var + 1e-5f);
        const float* g = gb + 32 * wave;       // wave-uniform: scalar loads
        const float* bt = gb + E + 32 * wave;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 8 * q + 2 * e;
                pk[e] = pack_bf16x2((v[c] - mean) * rs * g[c] + bt[c], (v[c + 1] - mean) * rs * g[c + 1] + bt[c + 1]);
            }
            *(u32x4*)(dst + xoff(row, 4 * wave + q)) = pk;
            if (to_global && row < p.S) *(u32x4*)(p.out + ((size_t)((size_t)row * p.N + n) * p.out_ld + 32 * wave + 8 * q) * 2) = pk;
        }
    };

    // acc (+ bias + residual from X) -> Z
    auto to_z = [&](f32x4 (&acc)[4][2], const float4 (&bias2)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n0 = wave * 32 + j * 16 + fq * 4;
            const float4 bv = bias2[j];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = mt * 16 + fr;
                const uint2 rx = *(const uint2*)(X + xoff(row, n0 >> 3) + (n0 & 4) * 2);
                float v[4] = {acc[mt][j][0] + bv.x + bf16_bits_to_f32(rx.x & 0xFFFFu), acc[mt][j][1] + bv.y + bf16_bits_to_f32(rx.x >> 16),
                              acc[mt][j][2] + bv.z + bf16_bits_to_f32(rx.y & 0xFFFFu), acc[mt][j][3] + bv.w + bf16_bits_to_f32(rx.y >> 16)};
                *(uint2*)(Tb + (row * ZLD + n0) * 2) = pack4(v);
            }
        }
    };

    // multi-head attention block: queries from XQ, keys/values from XKV; result (+ bias + X residual) lands in Z
    auto attention = [&](const unsigned char* XQ, const unsigned char* XKV) __attribute__((always_inline)) {
        const float* in_bias = pp;         // q | k | v, 3E
        const float* out_bias = pp + 3 * E;
        pp += 4 * E;
        unsigned short* Qs = (unsigned short*)Tb;
        unsigned short* Ks = Qs + SM * QLD;
        unsigned short* Vs = Ks + SM * QLD;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // projections of this head: wave w produces columns [16w, 16w+16) of Q_h, K_h and V_h
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                // biases are fetched BEFORE the GEMM: vector-memory results return in issue order, so a load issued after
                // it would sit behind the 16 weight fragments the ring has in flight
                const int c0 = wave * 16 + fq * 4;  // column inside the head
                const float4 bv = *(const float4*)(in_bias + part * E + h * HD + c0);
                f32x4 acc[4][1];
                zero<1>(acc);
                mm256<1>(acc, part == 0 ? XQ : XKV, ring, wp, loff, lx, ((h * 3 + part) * 8) % RING); TS(0); TPH(1);
                unsigned short* dstT = part == 0 ? Qs : part == 1 ? Ks : Vs;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4] = {acc[mt][0][0] + bv.x, acc[mt][0][1] + bv.y, acc[mt][0][2] + bv.z, acc[mt][0][3] + bv.w};
                    *(uint2*)(dstT + (mt * 16 + fr) * QLD + c0) = pack4(v);
                }
            }
            TSYNC();
            TPH(3);
            // scores of query tile qt against all 64 keys (both waves of a query tile compute them)
            f32x4 s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                const u32x4 a = *(const u32x4*)(Qs + (qt * 16 + fr) * QLD + ks * 32 + fq * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 b = *(const u32x4*)(Ks + (i * 16 + fr) * QLD + ks * 32 + fq * 8);
                    s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), s[i], 0, 0, 0);
                }
            }
            // softmax over keys (lane: rows fq*4+r, keys i*16+fr); keys >= S are masked
            float inv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = (i * 16 + fr < p.S) ? s[i][r] * p.scale : -INFINITY;
                    s[i][r] = v;
                    mx = fmaxf(mx, v);
                }
                mx = row16_max(mx);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __expf(s[i][r] - mx);
                    s[i][r] = pv;
                    sum += pv;
                }
                sum = row16_sum(sum);
                inv[r] = 1.0f / sum;
            }
            // P: C layout -> LDS tile of this query tile -> A fragments.  The two waves of a query tile write identical
            // values to the same addresses and each reads back after its own writes, so no barrier is needed.
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pw[(fq * 4 + r) * PLD + i * 16 + fr] = (unsigned short)f32_to_bf16_bits(s[i][r]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            f32x4 o[4];  // columns half*64 + i*16 + fr of the head
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const u32x4 a = *(const u32x4*)(Pw + fr * PLD + k2 * 32 + fq * 8);
                // transposed V block: lane 4q+p of each 16-lane group addresses key row (base + q), d columns 4p..4p+3
                const unsigned short* vrow = Vs + (k2 * 32 + fq * 8 + (fr >> 2)) * QLD + half * 64 + (fr & 3) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + i * 16));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 4 * QLD + i * 16));
                    const s16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), o[i], 0, 0, 0);
                }
            }
            // O (rows fq*4+r of query tile qt) -> AO[:, h*HD + half*64 ..]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = qt * 16 + fq * 4 + r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = h * HD + half * 64 + i * 16 + fr;
                    *(unsigned short*)(AO + xoff(row, col >> 3) + (col & 7) * 2) = (unsigned short)f32_to_bf16_bits(o[i][r] * inv[r]);
                }
            }
            TSYNC();  // every wave is done with this head's K/V (and AO is complete after the second head)
        }
        float4 ob[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) ob[j] = *(const float4*)(out_bias + wave * 32 + j * 16 + fq * 4);
        f32x4 acc[4][2];
        zero<2>(acc);
        mm256<2>(acc, AO, ring, wp, loff, lx); TS(0); TPH(1);
        to_z(acc, ob);
        TSYNC();
    };

    auto ffn = [&]() __attribute__((always_inline)) {
        const float* b1 = pp;
        const float* b2 = pp + FF;
        pp += FF + E;
        float4 b2v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b2v[j] = *(const float4*)(b2 + wave * 32 + j * 16 + fq * 4);
        f32x4 acc2[4][2];
        zero<2>(acc2);
        for (int c = 0; c < FF / 256; ++c) {
            float4 hb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) hb[j] = *(const float4*)(b1 + c * 256 + wave * 32 + j * 16 + fq * 4);
            f32x4 acc1[4][2];
            zero<2>(acc1);
            mm256<2>(acc1, X, ring, wp, loff, lx); TS(0); TPH(1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = wave * 32 + j * 16 + fq * 4;  // column inside the chunk
                const float4 bv = hb[j];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4] = {fmaxf(acc1[mt][j][0] + bv.x, 0.f), fmaxf(acc1[mt][j][1] + bv.y, 0.f), fmaxf(acc1[mt][j][2] + bv.z, 0.f),
                                  fmaxf(acc1[mt][j][3] + bv.w, 0.f)};
                    *(uint2*)(Tb + xoff(mt * 16 + fr, n0 >> 3) + (n0 & 4) * 2) = pack4(v);
                }
            }
            TSYNC();
            mm256<2>(acc2, Tb, ring, wp, loff, lx); TS(0); TPH(1);
            TSYNC();  // before the next chunk (or Z) overwrites the hidden tile
        }
        to_z(acc2, b2v);
        TSYNC();
    };
    auto add_ln = [&](unsigned char* dst, bool from_x, bool to_global) __attribute__((always_inline)) {
        layernorm(dst, pp, from_x, to_global);
        pp += 2 * E;
        TSYNC();
    };

    load_src();
    TSYNC();
    for (int l = 0; l < p.n_enc; ++l) {
        attention(X, X);
        add_ln(X, false, false);
        ffn();
        add_ln(X, false, false);
    }
    add_ln(MEM, true, false);  // encoder.norm
    load_src();  // nn.Transformer(src, src): the decoder starts from the same tokens
    TSYNC();
    for (int l = 0; l < p.n_dec; ++l) {
        attention(X, X);
        add_ln(X, false, false);
        attention(X, MEM);
        add_ln(X, false, false);
        ffn();
        add_ln(X, false, false);
    }
    add_ln(X, true, true);  // decoder.norm (dst X is scratch here; rows go to the output)
#ifdef TF_STAMP
    if (lane == 0 && wave == 1)
        for (int i = 0; i < 5; ++i) atomicAdd(&g_tf_stamps[i], st_acc[i]);
#endif
}

}  // namespace

#ifdef TF_STAMP
extern "C" int ccnet_debug_tf_stamps(unsigned long long* out8) {
    CC_HIP(hipDeviceSynchronize());
    CC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tf_stamps), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0};
    CC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tf_stamps), z, sizeof(z)));
    return CCNET_OK;
}
#endif

extern "C" int ccnet_transformer_fused_frags(int n_enc, int n_dec, long long* frags_per_wave, long long* param_floats) {
    CC_REQUIRE(n_enc >= 0 && n_dec >= 0 && frags_per_wave && param_floats, "transformer_fused_frags: bad arguments");
    const long long attn = 2 * 3 * 8 + 16, ffn = 2 * (FF / 256) * 16;  // per wave, 8 waves
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
                CC_HIP(hipFuncSetAttribute((const void*)transformer_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                attr_set[dev] = true;
            }
            hipLaunchKernelGGL(transformer_fused_kernel, dim3(p.N), dim3(NW * 64), LDS_BYTES, st, p);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_TRANSFORMER, n_enc * (fl_layer_attn + fl_ffn) + n_dec * (2 * fl_layer_attn + fl_ffn),
        2.0 * rows * E * 2 + (double)need_f * NW * 1024);
}
