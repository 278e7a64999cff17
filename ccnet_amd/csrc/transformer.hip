// Transformer pieces for MobileViT's nn.Transformer: residual + LayerNorm, and scaled-dot-product attention over
// the image-batch axis (batch_first=False quirk: sequence = images, "batch" = spatial positions).
// The linear layers run on the implicit-GEMM conv kernel (1x1 conv over S*N "pixels").
#include <math.h>
#include "common.h"

namespace {

// LP lanes per row (power of two >= E/8 capped at 64), up to 2 channel groups per lane (E <= 1024).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const unsigned char* x, const unsigned char* res, const float* gamma, const float* beta,
                                                        unsigned char* out, long long rows, int E, int LP, float eps) {
    const int G = E / 8;
    const int t = threadIdx.x;
    const int sub = t % LP;
    const long long row = ((long long)blockIdx.x * 256 + t) / LP;
    const bool live = row < rows;
    float v[2][8];
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = sub + q * LP;
        if (live && g < G) {
            Chan8<T>::load(x + ((size_t)row * E + g * 8) * sizeof(T), v[q]);
            if (res) {
                float r[8];
                Chan8<T>::load(res + ((size_t)row * E + g * 8) * sizeof(T), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[q][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = 0.0f;
        }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)E;
    float ss = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = sub + q * LP;
        if (g < G) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float d = v[q][e] - mean;
                ss += d * d;
            }
        }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = 1.0f / sqrtf(ss / (float)E + eps);
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int g = sub + q * LP;
        if (g < G) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (v[q][e] - mean) * rstd * gamma[g * 8 + e] + beta[g * 8 + e];
            Chan8<T>::store(out + ((size_t)row * E + g * 8) * sizeof(T), o8);
        }
    }
}

// One block per (position n, head h); 4 waves, each wave owns queries wave, wave+4, ...
// Keys/values are streamed through LDS in chunks of KC with an online softmax, so any Sk works.
constexpr int KC = 64;
constexpr int HD_MAX = 256;

struct AttnP {
    const unsigned char *q, *k, *v;
    unsigned char* o;
    int q_ld, q_off, k_ld, k_off, v_ld, v_off, o_ld, o_off;
    int Sq, Sk, N, hd;
    float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const AttnP p) {
    extern __shared__ __align__(16) float sm[];
    const int hd = p.hd, hdp = hd + 1;
    float* Ks = sm;                    // [KC][hd+1]
    float* Vs = Ks + KC * hdp;         // [KC][hd]
    float* Qs = Vs + KC * hd;          // [4][hd]
    const int n = blockIdx.x, h = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int DPL = HD_MAX / 64;  // output dims per lane
    const int nq_iter = (p.Sq + 3) / 4;
    const bool single = p.Sk <= KC;
    for (int qi = 0; qi < nq_iter; ++qi) {
        const int sq = qi * 4 + wave;
        const bool qlive = sq < p.Sq;
        __syncthreads();
        if (qlive) {
            const T* qp = (const T*)(p.q + ((size_t)((size_t)sq * p.N + n) * p.q_ld + p.q_off + h * hd) * sizeof(T));
            for (int d = lane; d < hd; d += 64) Qs[wave * hd + d] = load1<T>(qp + d) * p.scale;
        }
        float m_run = -INFINITY, l_run = 0.0f;
        float o[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) o[i] = 0.0f;
        for (int k0 = 0; k0 < p.Sk; k0 += KC) {
            const int kn = min(KC, p.Sk - k0);
            if (!single || qi == 0) {  // with a single key chunk K/V stay resident for every query
                if (!single) __syncthreads();
                for (int e = t; e < kn * hd; e += 256) {
                    const int j = e / hd, d = e - j * hd;
                    const size_t rowi = (size_t)(k0 + j) * p.N + n;
                    Ks[j * hdp + d] = load1<T>((const T*)(p.k + (rowi * p.k_ld + p.k_off + h * hd + d) * sizeof(T)));
                    Vs[j * hd + d] = load1<T>((const T*)(p.v + (rowi * p.v_ld + p.v_off + h * hd + d) * sizeof(T)));
                }
            }
            __syncthreads();
            if (qlive) {
                float sc = -INFINITY;
                if (lane < kn) {
                    float a = 0.0f;
                    const float* kr = Ks + lane * hdp;
                    const float* qr = Qs + wave * hd;
                    for (int d = 0; d < hd; ++d) a = fmaf(qr[d], kr[d], a);
                    sc = a;
                }
                const float m_new = fmaxf(m_run, wave_max(sc));
                const float pj = (lane < kn) ? __expf(sc - m_new) : 0.0f;
                const float alpha = __expf(m_run - m_new);  // 0 on the first chunk (m_run = -inf)
                l_run = l_run * alpha + wave_sum(pj);
#pragma unroll
                for (int i = 0; i < DPL; ++i) o[i] *= alpha;
                for (int j = 0; j < kn; ++j) {
                    const float pb = __shfl(pj, j, 64);
#pragma unroll
                    for (int i = 0; i < DPL; ++i) {
                        const int d = lane + 64 * i;
                        if (d < hd) o[i] = fmaf(pb, Vs[j * hd + d], o[i]);
                    }
                }
                m_run = m_new;
            }
        }
        if (qlive) {
            const float inv = 1.0f / l_run;
            T* op = (T*)(p.o + ((size_t)((size_t)sq * p.N + n) * p.o_ld + p.o_off + h * hd) * sizeof(T));
#pragma unroll
            for (int i = 0; i < DPL; ++i) {
                const int d = lane + 64 * i;
                if (d < hd) store1<T>(op + d, o[i] * inv);
            }
        }
    }
}

// ---- bf16 MFMA attention: one block per (position n, head h), 4 waves x 16 queries, keys in chunks of 64 with an
// online softmax.  Q/K/V tiles are staged row-major in LDS; S = Q K^T and O = P V both run on v_mfma_f32_16x16x32_bf16.
// K fragments are plain ds_read_b128 (K is k-contiguous); V is consumed through ds_read_b64_tr_b16, gfx950's transposing
// LDS read, so V needs no transposed copy.  P goes C-layout -> A-layout through a wave-private LDS tile.
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int APAD = 8;  // bf16 elements of padding per LDS row (16 B: shifts consecutive rows by 4 banks)

template <int HDT>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const AttnP p) {
    extern __shared__ __align__(16) unsigned char smraw[];
    const int hd = p.hd, ldr = hd + APAD;  // row stride in elements
    unsigned short* Qs = (unsigned short*)smraw;          // [64][ldr]
    unsigned short* Ks = Qs + 64 * ldr;                   // [64][ldr]
    unsigned short* Vs = Ks + 64 * ldr;                   // [64][ldr]
    unsigned short* Ps = Vs + 64 * ldr;                   // [4][16][72]
    constexpr int PLD = 72;
    const int n = blockIdx.x, h = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int cpr = hd >> 3;  // 16-B chunks per row
    constexpr int NTMAX = HDT / 16;
    const int ntiles = hd >> 4, ksteps = hd >> 5;
    const bool single = p.Sk <= 64;
    unsigned short* Pw = Ps + wave * 16 * PLD;

    auto stage = [&](unsigned short* dst, const unsigned char* src, int ld, int off, int r0, int rmax) {
        for (int e = t; e < 64 * cpr; e += 256) {
            const int row = e / cpr, ch = e - row * cpr;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r0 + row < rmax) v = *(const uint4*)(src + (((size_t)(r0 + row) * p.N + n) * ld + off + h * hd + ch * 8) * 2);
            *(uint4*)(dst + row * ldr + ch * 8) = v;
        }
    };

    for (int q0 = 0; q0 < p.Sq; q0 += 64) {
        __syncthreads();
        stage(Qs, p.q, p.q_ld, p.q_off, q0, p.Sq);
        float m_run[4], l_run[4];
        f32x4 o[NTMAX];
#pragma unroll
        for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.0f; }
#pragma unroll
        for (int i = 0; i < NTMAX; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < p.Sk; k0 += 64) {
            if (!single || q0 == 0) {
                if (!single) __syncthreads();
                stage(Ks, p.k, p.k_ld, p.k_off, k0, p.Sk);
                stage(Vs, p.v, p.v_ld, p.v_off, k0, p.Sk);
            }
            __syncthreads();
            // S[16 q][64 keys]
            f32x4 s[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < ksteps; ++ks) {
                const uint4 a = *(const uint4*)(Qs + (wave * 16 + fr) * ldr + ks * 32 + fq * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 b = *(const uint4*)(Ks + (i * 16 + fr) * ldr + ks * 32 + fq * 8);
                    s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), s[i], 0, 0, 0);
                }
            }
            // online softmax; this lane holds rows fq*4+r, keys i*16+fr
            float alpha[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = (k0 + i * 16 + fr < p.Sk) ? s[i][r] * p.scale : -INFINITY;
                    s[i][r] = v;
                    mx = fmaxf(mx, v);
                }
#pragma unroll
                for (int o_ = 8; o_ > 0; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_, 64));
                const float m_new = fmaxf(m_run[r], mx);
                alpha[r] = __expf(m_run[r] - m_new);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __expf(s[i][r] - m_new);
                    s[i][r] = pv;
                    sum += pv;
                }
#pragma unroll
                for (int o_ = 8; o_ > 0; o_ >>= 1) sum += __shfl_xor(sum, o_, 64);
                l_run[r] = l_run[r] * alpha[r] + sum;
                m_run[r] = m_new;
            }
#pragma unroll
            for (int i = 0; i < NTMAX; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[i][r] *= alpha[r];
            // P: C layout -> LDS [16][PLD] -> A fragments
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Pw[(fq * 4 + r) * PLD + i * 16 + fr] = (unsigned short)f32_to_bf16_bits(s[i][r]);
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const uint4 a = *(const uint4*)(Pw + fr * PLD + k2 * 32 + fq * 8);
                // transposed V block: lane 4q+p of each 16-lane group addresses key row (base + q), d columns 4p..4p+3
                const unsigned short* vrow = Vs + (k2 * 32 + fq * 8 + (fr >> 2)) * ldr + (fr & 3) * 4;
#pragma unroll
                for (int i = 0; i < NTMAX; ++i) {
                    if (i < ntiles) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + i * 16));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 4 * ldr + i * 16));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), o[i], 0, 0, 0);
                    }
                }
            }
        }
        // normalise and store: rows fq*4+r, columns i*16+fr
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sq = q0 + wave * 16 + fq * 4 + r;
            if (sq < p.Sq) {
                const float inv = 1.0f / l_run[r];
                __bf16* op = (__bf16*)(p.o + (((size_t)sq * p.N + n) * p.o_ld + p.o_off + h * hd) * 2);
#pragma unroll
                for (int i = 0; i < NTMAX; ++i)
                    if (i < ntiles) op[i * 16 + fr] = (__bf16)(o[i][r] * inv);
            }
        }
    }
}

}  // namespace

extern "C" int ccnet_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* out, int dtype, int64_t rows, int E,
                               float eps, ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && x && gamma && beta && out, "layernorm: null pointer / bad dtype");
    CC_REQUIRE(rows > 0 && E > 0 && (E % 8) == 0 && E <= 1024, "layernorm: E must be a multiple of 8, at most 1024 (got %d)", E);
    CC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!res || ((uintptr_t)res % 16) == 0), "layernorm: pointers must be 16-byte aligned");
    int LP = 1;
    while (LP < E / 8 && LP < 64) LP <<= 1;
    const unsigned blocks = (unsigned)(((long long)rows * LP + 255) / 256);
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16)
                hipLaunchKernelGGL(layernorm_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)x, (const unsigned char*)res, gamma,
                                   beta, (unsigned char*)out, (long long)rows, E, LP, eps);
            else
                hipLaunchKernelGGL(layernorm_kernel<float>, dim3(blocks), dim3(256), 0, st, (const unsigned char*)x, (const unsigned char*)res, gamma,
                                   beta, (unsigned char*)out, (long long)rows, E, LP, eps);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_LAYERNORM, 0.0, (double)rows * E * cc_esize(dtype) * (res ? 3.0 : 2.0));
}

extern "C" int ccnet_attention(ccnet_slice q, ccnet_slice k, ccnet_slice v, int dtype, int Sq, int Sk, int N, int heads, int hd, ccnet_slice out,
                               ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype), "attention: bad dtype");
    CC_REQUIRE(Sq > 0 && Sk > 0 && N > 0 && heads > 0 && hd > 0 && hd <= HD_MAX, "attention: bad sizes (hd=%d, max %d)", hd, HD_MAX);
    CC_REQUIRE(heads <= 65535, "attention: too many heads");
    const int C = heads * hd;
    const ccnet_slice* sl[4] = {&q, &k, &v, &out};
    for (int i = 0; i < 4; ++i)
        CC_REQUIRE(sl[i]->ptr && sl[i]->C == C && sl[i]->coff >= 0 && sl[i]->ldc >= sl[i]->coff + C && sl[i]->shift == 0, "attention: bad slice %d", i);
    AttnP p{(const unsigned char*)q.ptr, (const unsigned char*)k.ptr, (const unsigned char*)v.ptr, (unsigned char*)out.ptr,
            q.ldc, q.coff, k.ldc, k.coff, v.ldc, v.coff, out.ldc, out.coff, Sq, Sk, N, hd, 1.0f / sqrtf((float)hd)};
    const size_t lds = (size_t)(KC * (hd + 1) + KC * hd + 4 * hd) * sizeof(float);
    // bf16 with hd a multiple of 32 runs on the matrix cores
    const bool mfma = dtype == CCNET_BF16 && (hd % 32) == 0 && (q.ldc % 8) == 0 && (k.ldc % 8) == 0 && (v.ldc % 8) == 0 && (q.coff % 8) == 0 &&
                      (k.coff % 8) == 0 && (v.coff % 8) == 0 && ((uintptr_t)q.ptr % 16) == 0 && ((uintptr_t)k.ptr % 16) == 0 && ((uintptr_t)v.ptr % 16) == 0;
    const size_t lds_m = ((size_t)3 * 64 * (hd + APAD) + 4 * 16 * 72) * 2;
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (mfma) {
                if (hd <= 128) {
                    if (lds_m > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)attention_mfma_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
                    hipLaunchKernelGGL(attention_mfma_kernel<128>, dim3(N, heads), dim3(256), lds_m, st, p);
                } else {
                    if (lds_m > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)attention_mfma_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
                    hipLaunchKernelGGL(attention_mfma_kernel<256>, dim3(N, heads), dim3(256), lds_m, st, p);
                }
            } else if (dtype == CCNET_BF16) {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)attention_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(attention_kernel<__bf16>, dim3(N, heads), dim3(256), lds, st, p);
            } else {
                if (lds > 64 * 1024) CC_HIP(hipFuncSetAttribute((const void*)attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(attention_kernel<float>, dim3(N, heads), dim3(256), lds, st, p);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_ATTENTION, 4.0 * (double)Sq * Sk * N * heads * hd,
        ((double)Sq * 2 + (double)Sk * 2) * N * (double)C * cc_esize(dtype));
}
