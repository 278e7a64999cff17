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
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16) {
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
