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

    // tokens of this position -> X (rows >= S are zero)
    auto load_src = [&]() __attribute__((always_inline)) {
        for (int e = t; e < SM * 32; e += NW * 64) {
            const int row = e >> 5, ch = e & 31;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (row < p.S) v = *(const u32x4*)(p.src + ((size_t)((size_t)row * p.N + n) * p.src_ld + ch * 8) * 2);
            *(u32x4*)(X + xoff(row, ch)) = v;
        }
    };

    // Z (pre-LayerNorm, padded rows in T) -> LayerNorm -> dst (swizzled operand) and optionally global rows
    auto layernorm = [&](unsigned char* dst, const float* gb, bool to_global) __attribute__((always_inline)) {
        const float4 g = *(const float4*)(gb + lane * 4);
        const float4 b = *(const float4*)(gb + E + lane * 4);
#pragma unroll 2
        for (int r = 0; r < SM / NW; ++r) {
            const int row = wave * (SM / NW) + r;
            const uint2 raw = *(const uint2*)(Tb + (row * ZLD + lane * 4) * 2);
            float v[4] = {bf16_bits_to_f32(raw.x & 0xFFFFu), bf16_bits_to_f32(raw.x >> 16), bf16_bits_to_f32(raw.y & 0xFFFFu), bf16_bits_to_f32(raw.y >> 16)};
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / E);
            float d[4] = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
            const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / E);
            const float rs = rsqrtf(var + 1e-5f);
            float o[4] = {d[0] * rs * g.x + b.x, d[1] * rs * g.y + b.y, d[2] * rs * g.z + b.z, d[3] * rs * g.w + b.w};
            const uint2 pk = pack4(o);
            *(uint2*)(dst + xoff(row, lane >> 1) + (lane & 1) * 8) = pk;
            if (to_global && row < p.S) *(uint2*)(p.out + ((size_t)((size_t)row * p.N + n) * p.out_ld + lane * 4) * 2) = pk;
        }
    };

    // acc (+ bias + residual from X) -> Z
    auto to_z = [&](f32x4 (&acc)[4][2], const float* bias) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n0 = wave * 32 + j * 16 + fq * 4;
            const float4 bv = *(const float4*)(bias + n0);
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
                f32x4 acc[4][1];
                zero<1>(acc);
                mm256<1>(acc, part == 0 ? XQ : XKV, ring, wp, loff, lx, ((h * 3 + part) * 8) % RING);
                unsigned short* dstT = part == 0 ? Qs : part == 1 ? Ks : Vs;
                const int c0 = wave * 16 + fq * 4;  // column inside the head
                const float4 bv = *(const float4*)(in_bias + part * E + h * HD + c0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4] = {acc[mt][0][0] + bv.x, acc[mt][0][1] + bv.y, acc[mt][0][2] + bv.z, acc[mt][0][3] + bv.w};
                    *(uint2*)(dstT + (mt * 16 + fr) * QLD + c0) = pack4(v);
                }
            }
            __syncthreads();
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
#pragma unroll
                for (int o_ = 8; o_ > 0; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_, 64));
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __expf(s[i][r] - mx);
                    s[i][r] = pv;
                    sum += pv;
                }
#pragma unroll
                for (int o_ = 8; o_ > 0; o_ >>= 1) sum += __shfl_xor(sum, o_, 64);
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
            __syncthreads();  // every wave is done with this head's K/V (and AO is complete after the second head)
        }
        f32x4 acc[4][2];
        zero<2>(acc);
        mm256<2>(acc, AO, ring, wp, loff, lx);
        to_z(acc, out_bias);
        __syncthreads();
    };

    auto ffn = [&]() __attribute__((always_inline)) {
        const float* b1 = pp;
        const float* b2 = pp + FF;
        pp += FF + E;
        f32x4 acc2[4][2];
        zero<2>(acc2);
        for (int c = 0; c < FF / 256; ++c) {
            f32x4 acc1[4][2];
            zero<2>(acc1);
            mm256<2>(acc1, X, ring, wp, loff, lx);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = wave * 32 + j * 16 + fq * 4;  // column inside the chunk
                const float4 bv = *(const float4*)(b1 + c * 256 + n0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4] = {fmaxf(acc1[mt][j][0] + bv.x, 0.f), fmaxf(acc1[mt][j][1] + bv.y, 0.f), fmaxf(acc1[mt][j][2] + bv.z, 0.f),
                                  fmaxf(acc1[mt][j][3] + bv.w, 0.f)};
                    *(uint2*)(Tb + xoff(mt * 16 + fr, n0 >> 3) + (n0 & 4) * 2) = pack4(v);
                }
            }
            __syncthreads();
            mm256<2>(acc2, Tb, ring, wp, loff, lx);
            __syncthreads();  // before the next chunk (or Z) overwrites the hidden tile
        }
        to_z(acc2, b2);
        __syncthreads();
    };
    auto add_ln = [&](unsigned char* dst, bool to_global) __attribute__((always_inline)) {
        layernorm(dst, pp, to_global);
        pp += 2 * E;
        __syncthreads();
    };
    // plain LayerNorm of X (encoder.norm / decoder.norm): copy X to Z first
    auto ln_of_x = [&](unsigned char* dst, bool to_global) __attribute__((always_inline)) {
        for (int r = 0; r < SM / NW; ++r) {
            const int row = wave * (SM / NW) + r;
            *(uint2*)(Tb + (row * ZLD + lane * 4) * 2) = *(const uint2*)(X + xoff(row, lane >> 1) + (lane & 1) * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // rows are wave private
        add_ln(dst, to_global);
    };

    load_src();
    __syncthreads();
    for (int l = 0; l < p.n_enc; ++l) {
        attention(X, X);
        add_ln(X, false);
        ffn();
        add_ln(X, false);
    }
    ln_of_x(MEM, false);
    load_src();  // nn.Transformer(src, src): the decoder starts from the same tokens
    __syncthreads();
    for (int l = 0; l < p.n_dec; ++l) {
        attention(X, X);
        add_ln(X, false);
        attention(X, MEM);
        add_ln(X, false);
        ffn();
        add_ln(X, false);
    }
    ln_of_x(X, true);
}

}  // namespace

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
