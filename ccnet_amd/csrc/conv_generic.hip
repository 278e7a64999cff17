// Generic (fallback) form of the implicit-GEMM convolution: one 128 x BN tile per block, per-tap address arithmetic that
// also handles an upsampled (shift > 0) operand under a k > 1 window.  conv_igemm.hip holds the fast persistent kernel and
// dispatches here only for the cases it does not cover.
// BN-folded NHWC implicit-GEMM convolution on gfx950 MFMA.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Ctot.  A (pixels x K) is gathered on the fly from
// one or two NHWC channel slices (second slice = torch.cat, shift = nearest upsampling), B is the packed
// weight [Cout_pad][Kpad].  Block = 256 threads (4 waves), tile 128 x BN x 128 bytes of K (64 bf16 / 32 f32),
// LDS double buffered, next tile prefetched into registers while the current one feeds the MFMAs
// (register staging because out-of-image taps must read as zero).
// LDS rows are 128 B = 8 chunks of 16 B; chunk c of row r lives at chunk c ^ ((r>>1)&7), which makes both the
// ds_write_b128 fill and the ds_read_b128 fragment reads bank-conflict free for the gfx950 lane groups.
// bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact fp32, k order permuted identically
// for A and B).  Epilogue: bias + activation (+ residual) through a wave-private LDS transpose so that global
// stores are 16 B per lane along the channel axis.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BKB = 128;  // bytes of K per row per step
constexpr int NTHREADS = 256;

struct SegP {
    const unsigned char* ptr;
    int ldc, coff, C, shift, Hs, Ws;
};

struct ConvP {
    SegP seg[2];
    int Ctot, C0;
    int B, H, W, kh, kw, stride, pad, Ho, Wo, Cout;
    long long M;
    const unsigned char* weight;
    int Kpad;  // elements
    int nk;    // K steps
    const float* bias;
    int act;
    const unsigned char* res_ptr;
    int res_ldc, res_coff;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int vec_out;  // 1: channel groups of 8 are aligned in out/residual
};

// native vector type for everything staged through registers (HIP's uint4 is a struct wrapping a union, which LLVM
// sometimes leaves in scratch when held in small arrays)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Frag;
template <> struct Frag<__bf16> {
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Frag<float> {
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& acc) {
        f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BKB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, typename OutT, int BN, int WM, int WN>
__global__ __launch_bounds__(NTHREADS) void conv_generic_kernel(const ConvP p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int EPC = 16 / ES;   // elements per 16-B chunk
    constexpr int BK = BKB / ES;   // elements of K per step
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int A_LOADS = BM / 32;                   // 4
    constexpr int B_LOADS = (BN >= 32) ? BN / 32 : 1;  // rows of B per thread
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB;
    constexpr int EP_STRIDE = WTN + 4;  // floats per row in the epilogue staging tile
    constexpr int EP_BYTES = 4 * 16 * EP_STRIDE * 4;
    constexpr int SMEM = (2 * (A_BYTES + B_BYTES) > EP_BYTES) ? 2 * (A_BYTES + B_BYTES) : EP_BYTES;
    __shared__ __align__(16) unsigned char smem[SMEM];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread gather state: chunk column `lc`, rows lr + 32*i
    const int lc = t & 7, lr = t >> 3;
    int iy0[A_LOADS], ix0[A_LOADS], bb[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        long long m = m0 + lr + 32 * i;
        if (m < p.M) {
            int ox = (int)(m % p.Wo);
            long long r = m / p.Wo;
            int oy = (int)(r % p.Ho);
            bb[i] = (int)(r / p.Ho);
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
        } else {
            bb[i] = 0; iy0[i] = -(1 << 28); ix0[i] = 0;  // row past M: every tap is out of the image
        }
    }
    // K cursor of this thread's chunk: channel c within the concatenated input, tap (ky, kx)
    int kc = lc * EPC;
    int tap = kc / p.Ctot;
    int c = kc - tap * p.Ctot;
    int ky = tap / p.kw, kx = tap - ky * p.kw;
    const bool b_active = (BN >= 32) || (t < BN * 8);
    const unsigned char* wrow = p.weight + ((size_t)(n0 + lr) * p.Kpad + lc * EPC) * ES;

    u32x4 ra[A_LOADS], rb[B_LOADS];
    unsigned okbits = 0u;
    auto gather = [&](int kstep) {
        const int s = (c >= p.C0) ? 1 : 0;
        const SegP& sg = p.seg[s];
        const int cs = c - (s ? p.C0 : 0);
        const bool tap_ok = ky < p.kh;
        const unsigned char* sbase = sg.ptr + (size_t)(sg.coff + cs) * ES;
        okbits = 0u;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // branch-free: out-of-image taps load a clamped (always mapped) pixel and are zeroed when the tile is
            // written to LDS, so the loads of a K step issue back to back and stay in flight across the MFMAs
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            const size_t pix = ((size_t)bb[i] * sg.Hs + (iyc >> sg.shift)) * sg.Ws + (ixc >> sg.shift);
            ra[i] = *(const u32x4*)(sbase + pix * sg.ldc * ES);
            okbits |= (ok ? 1u : 0u) << i;
        }
        if constexpr (BN >= 32) {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                rb[i] = *(const u32x4*)(wrow + ((size_t)(32 * i) * p.Kpad + (size_t)kstep * BK) * ES);
        } else {
            // BN == 16: only 128 threads own a weight chunk; the others re-read row 0's chunk and never store it
            rb[0] = *(const u32x4*)((b_active ? wrow : wrow - (size_t)lr * p.Kpad * ES) + (size_t)kstep * BK * ES);
        }
    };
    auto advance = [&]() {
        c += BK;
        while (c >= p.Ctot) {
            c -= p.Ctot;
            if (++kx == p.kw) { kx = 0; ++ky; }
        }
    };
    auto stage = [&](int buf) {
        unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *(u32x4*)(As + lds_off(lr + 32 * i, lc)) = ((okbits >> i) & 1u) ? ra[i] : u32x4{0u, 0u, 0u, 0u};
        if (b_active) {
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i) *(u32x4*)(Bs + lds_off(lr + 32 * i, lc)) = rb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(As + lds_off(wm * WTM + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *(const u32x4*)(Bs + lds_off(wn * WTN + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) Frag<T>::mma(fa[i], fb[j], acc[i][j]);
        }
    };
    // software pipeline: the global loads of tile k+1 are in flight while tile k feeds the MFMAs; one barrier per tile
    gather(0);
    stage(0);
    __syncthreads();
    for (int k = 0; k + 1 < p.nk; ++k) {
        advance();
        gather(k + 1);
        compute(k & 1);
        stage((k + 1) & 1);
        __syncthreads();
    }
    compute((p.nk - 1) & 1);
    __syncthreads();

    // ---- epilogue: per wave, one 16-row slab at a time through LDS
    float* ep = (float*)smem + wave * (16 * EP_STRIDE);
    constexpr int UPR = WTN / 8;          // 8-channel units per row
    constexpr int UNITS = 16 * UPR;       // per slab
    const int nbase = n0 + wn * WTN;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ep[(fq * 4 + r) * EP_STRIDE + j * 16 + fr] = acc[i][j][r];
        __syncthreads();
        for (int u = lane; u < UNITS; u += 64) {
            const int row = u / UPR, cu = u - row * UPR;
            const long long m = m0 + wm * WTM + i * 16 + row;
            const int n = nbase + cu * 8;
            if (m >= p.M || n >= p.Cout) continue;
            float v[8];
            const float4 x0 = *(const float4*)(ep + row * EP_STRIDE + cu * 8);
            const float4 x1 = *(const float4*)(ep + row * EP_STRIDE + cu * 8 + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
            const int nv = (p.Cout - n) < 8 ? (p.Cout - n) : 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float b = (p.bias && e < nv) ? p.bias[n + e] : 0.0f;
                v[e] = apply_act(v[e] + b, p.act);
            }
            unsigned char* optr = p.out_ptr + ((size_t)m * p.out_ldc + p.out_coff + n) * sizeof(OutT);
            if (p.vec_out && nv == 8) {
                if (p.res_ptr) {
                    float rv[8];
                    Chan8<T>::load(p.res_ptr + ((size_t)m * p.res_ldc + p.res_coff + n) * ES, rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                Chan8<OutT>::store(optr, v);
            } else {
                for (int e = 0; e < nv; ++e) {
                    float x = v[e];
                    if (p.res_ptr) x += load1<T>((const T*)(p.res_ptr + ((size_t)m * p.res_ldc + p.res_coff + n + e) * ES));
                    store1<OutT>((OutT*)optr + e, x);
                }
            }
        }
    }
}

int pick_bn(int Cout) { return Cout <= 16 ? 16 : Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128; }

template <typename T, typename OutT>
int launch(const ConvP& p, int bn, hipStream_t st) {
    dim3 block(NTHREADS);
    long long mt = (p.M + BM - 1) / BM;
    int coutp = ((p.Cout + bn - 1) / bn) * bn;
    dim3 grid((unsigned)mt, (unsigned)(coutp / bn));
    switch (bn) {
        case 128: hipLaunchKernelGGL((conv_generic_kernel<T, OutT, 128, 2, 2>), grid, block, 0, st, p); break;
        case 64: hipLaunchKernelGGL((conv_generic_kernel<T, OutT, 64, 2, 2>), grid, block, 0, st, p); break;
        case 32: hipLaunchKernelGGL((conv_generic_kernel<T, OutT, 32, 4, 1>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((conv_generic_kernel<T, OutT, 16, 4, 1>), grid, block, 0, st, p); break;
    }
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

}  // namespace

int ccnet_conv2d_generic(const ccnet_conv_desc* d, ccnet_stream stream) {
    CC_REQUIRE(d != nullptr, "conv2d: null descriptor");
    CC_REQUIRE(cc_dtype_ok(d->dtype) && cc_dtype_ok(d->out_dtype), "conv2d: bad dtype");
    CC_REQUIRE(!(d->dtype == CCNET_F32 && d->out_dtype == CCNET_BF16), "conv2d: fp32 in / bf16 out is not built");
    CC_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv2d: nseg must be 1 or 2");
    CC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad geometry");
    CC_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->kh) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->kw) / d->stride + 1,
               "conv2d: Ho/Wo (%d,%d) do not match H,W,k,stride,pad", d->Ho, d->Wo);
    CC_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), "conv2d: too many input pixels");
    ConvP p{};
    int ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const ccnet_slice& sl = d->in[s];
        CC_REQUIRE(cc_slice_vec_ok(sl, d->dtype), "conv2d: input slice %d is not 16-byte channel aligned (C=%d ldc=%d coff=%d)", s, sl.C, sl.ldc, sl.coff);
        CC_REQUIRE((d->H % (1 << sl.shift)) == 0 && (d->W % (1 << sl.shift)) == 0, "conv2d: H,W not divisible by the upsampling factor");
        p.seg[s] = SegP{(const unsigned char*)sl.ptr, sl.ldc, sl.coff, sl.C, sl.shift, d->H >> sl.shift, d->W >> sl.shift};
        ctot += sl.C;
    }
    if (d->nseg == 1) p.seg[1] = p.seg[0];
    p.Ctot = ctot;
    p.C0 = d->in[0].C;
    int coutp = 0, kpad = 0;
    ccnet_conv_pack_dims(d->dtype, d->Cout, d->kh * d->kw * ctot, &coutp, &kpad);
    CC_REQUIRE(d->weight && ((uintptr_t)d->weight % 16) == 0, "conv2d: weight pointer must be 16-byte aligned");
    CC_REQUIRE(d->Kpad == kpad && d->Cout_pad == coutp, "conv2d: weight packed as [%d][%d], expected [%d][%d]", d->Cout_pad, d->Kpad, coutp, kpad);
    p.B = d->B; p.H = d->H; p.W = d->W; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.M = (long long)d->B * d->Ho * d->Wo;
    p.weight = (const unsigned char*)d->weight;
    p.Kpad = kpad;
    p.nk = kpad / (BKB / cc_esize(d->dtype));
    p.bias = d->bias;
    p.act = d->act;
    CC_REQUIRE(d->act >= CCNET_ACT_NONE && d->act <= CCNET_ACT_LRELU, "conv2d: unknown activation %d", d->act);
    const ccnet_slice& o = d->out;
    CC_REQUIRE(o.ptr && o.shift == 0 && o.C == d->Cout && o.coff >= 0 && o.ldc >= o.coff + o.C, "conv2d: bad output slice");
    p.out_ptr = (unsigned char*)o.ptr; p.out_ldc = o.ldc; p.out_coff = o.coff;
    int og = 16 / cc_esize(d->out_dtype);  // channels per 16 B of output
    bool vec = (o.ldc % og) == 0 && (o.coff % og) == 0 && ((uintptr_t)o.ptr % 16) == 0 && (o.ldc % 8) == 0 && (o.coff % 8) == 0;
    if (d->residual.ptr) {
        const ccnet_slice& r = d->residual;
        CC_REQUIRE(r.shift == 0 && r.C == d->Cout && r.coff >= 0 && r.ldc >= r.coff + r.C, "conv2d: bad residual slice");
        p.res_ptr = (const unsigned char*)r.ptr; p.res_ldc = r.ldc; p.res_coff = r.coff;
        int rg = 16 / cc_esize(d->dtype);
        vec = vec && (r.ldc % rg) == 0 && (r.coff % rg) == 0 && ((uintptr_t)r.ptr % 16) == 0 && (r.ldc % 8) == 0 && (r.coff % 8) == 0;
    }
    p.vec_out = vec ? 1 : 0;
    const int bn = pick_bn(d->Cout);
    const int dt = d->dtype, odt = d->out_dtype;
    return ccnet_submit(
        [p, bn, dt, odt](hipStream_t st, const float*) -> int {
            if (dt == CCNET_BF16) return odt == CCNET_BF16 ? launch<__bf16, __bf16>(p, bn, st) : launch<__bf16, float>(p, bn, st);
            return launch<float, float>(p, bn, st);
        },
        stream, CCNET_OP_CONV, 2.0 * (double)p.M * d->Cout * d->kh * d->kw * ctot,
        ((double)d->B * d->H * d->W * ctot + (double)d->Cout * d->kh * d->kw * ctot) * cc_esize(d->dtype) +
            (double)p.M * d->Cout * (cc_esize(d->out_dtype) + (d->residual.ptr ? cc_esize(d->dtype) : 0)));
}
