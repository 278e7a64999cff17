// 1x1 stride-1 convolution of the thin layers (K <= 256 input and N <= 256 output channels, bf16): weights resident in LDS,
// pixels streamed global -> registers -> MFMA, no barrier after the prologue.
//
// Why a second kernel for these: through conv_igemm a thin layer costs (fixed) + (bytes at ~7 TB/s) + (its tiles' MFMA and
// SiLU time) ADDED UP -- measured over the 1x1 layers of `net` at 40x40 and 80x80 -- because a block of four waves takes a
// tile through load, barrier, MFMA and a 64-value-per-lane epilogue in sequence, and re-fetches the (identical) weight tile
// for every pixel tile.  A device copy of the same bytes takes a third of the time.  Here
//  * the folded weights are put into LDS once per workgroup, already in MFMA fragment order (fragment (n-tile j, k-step ks)
//    = 64 lanes x 16 B, conflict-free ds_read_b128); one workgroup of 8 waves per CU, up to 128 KB of weights;
//  * every wave walks its own blocks of 32 pixels: the lane's 16 bytes of pixel (fr) and k-quarter (fq) come straight from
//    global memory (a block of a contiguous NHWC map is one contiguous run), and the registers of k-step ks are refilled with
//    the NEXT block's data right after their last use, so a whole block of loads is in flight per wave while it computes;
//  * two neighbouring n-tiles hold the two halves of 8 consecutive channels (the fragment rows are dealt accordingly when the
//    weights are staged), so the epilogue stores 16 B per lane;
//  * more than 128 output channels run as two passes over the same pixel registers.
// Arithmetic is conv_igemm's: bf16 operands, fp32 accumulation from zero over k in the same MFMA shape and k order, then bias,
// activation and rounding -- results are bit-identical to it, so which kernel a layer gets (it depends on the pixel count) does
// not show in the output (tests/test_ops_gpu.py:test_conv1x1_stream_equals_implicit_gemm_bitwise and the batch-independence
// test of tests/test_forward_gpu.py).
#include <stdlib.h>
#include "conv_common.h"

namespace {

struct StreamQ {
    const unsigned char* in0;
    const unsigned char* in1;
    unsigned ldc0B, ldc1B;  // pixel strides in bytes
    int KS0;                // k-steps (32 channels) that come from segment 0
    int shift0, shift1;     // nearest-neighbour upsampling of a segment (nn.Upsample before the torch.cat): pixel (y, x) reads (y >> s, x >> s)
    int H, W;               // output = unshifted input size
    unsigned magicW, shiftW, magicH, shiftH;
    unsigned M, nblocks;
    int Cout;               // multiple of 64
    const unsigned char* weight;  // packed [Cout_pad][Kpad] bf16
    int Kpad;
    const float* bias;
    unsigned char* out;
    unsigned out_ldcB;      // bytes
    int out_coff;
};

constexpr int NTP = 8;           // n-tiles per pass (128 channels)

// channel of row a (0..15) of n-tile j: tiles 2q and 2q+1 interleave in groups of four, see the header
__device__ __forceinline__ int frag_channel(int j, int a) { return (j >> 1) * 32 + ((a >> 2) << 3) + ((j & 1) << 2) + (a & 3); }

// PXT pixel tiles (of 16) per block, SW waves per workgroup (SW / 4 per SIMD: the register budget follows)
// and PD register sets of pixel data (blocks in flight per wave)
template <int KS, int ACT, int PXT, int SW, int PD>
__global__ __launch_bounds__(SW * 64, SW / 4) void conv_stream_kernel(const StreamQ p) {
    constexpr int PXB = 16 * PXT;
    extern __shared__ __align__(16) unsigned char smem[];  // weight fragments [NT][KS][64][16 B] | bias [Cout] fp32
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int NT = p.Cout >> 4;
    float* const bias_l = (float*)(smem + (size_t)NT * KS * 1024);

    // ---- prologue: weights -> LDS in fragment order (global reads run along the packed rows)
    {
        const int cpr = KS * 4;  // 16-B chunks per weight row
        const int total = p.Cout * cpr;
        for (int e = t; e < total; e += SW * 64) {
            const int n = e / cpr, c = e - n * cpr;
            const int ks = c >> 2, kq = c & 3;
            // channel n sits in n-tile j, row a with frag_channel(j, a) == n
            const int g = n >> 5, w = n & 31;
            const int j = 2 * g + ((w >> 2) & 1), a = ((w >> 3) << 2) + (w & 3);
            const u32x4 v = *(const u32x4*)(p.weight + ((size_t)n * p.Kpad + ks * 32 + kq * 8) * 2);
            *(u32x4*)(smem + ((size_t)(j * KS + ks) * 64 + kq * 16 + a) * 16) = v;
        }
        for (int e = t; e < p.Cout; e += SW * 64) bias_l[e] = p.bias ? p.bias[e] : 0.0f;
    }
    __syncthreads();

    const unsigned wstride = gridDim.x * SW;
    unsigned blk = blockIdx.x * SW + wave;
    if (blk >= p.nblocks) return;

    // the lane's 16 bytes of pixel tile tt, k-step ks of block b (rows beyond M read the last pixel; they are not stored)
    const bool shifted = (p.shift0 | p.shift1) != 0;  // uniform
    auto xload = [&](unsigned b, int tt, int ks) -> u32x4 {
        const unsigned m = min(b * PXB + tt * 16 + fr, p.M - 1u);
        unsigned m0 = m, m1 = m;
        if (shifted) {
            const unsigned q1 = fastdiv(m, p.magicW, p.shiftW), xx = m - q1 * (unsigned)p.W;
            const unsigned bi = fastdiv(q1, p.magicH, p.shiftH), yy = q1 - bi * (unsigned)p.H;
            m0 = (bi * (unsigned)(p.H >> p.shift0) + (yy >> p.shift0)) * (unsigned)(p.W >> p.shift0) + (xx >> p.shift0);
            m1 = (bi * (unsigned)(p.H >> p.shift1) + (yy >> p.shift1)) * (unsigned)(p.W >> p.shift1) + (xx >> p.shift1);
        }
        const unsigned char* src = ks < p.KS0 ? p.in0 + (size_t)m0 * p.ldc0B + ks * 64 : p.in1 + (size_t)m1 * p.ldc1B + (ks - p.KS0) * 64;
        return *(const u32x4*)(src + fq * 16);
    };
    const int npass = (NT + NTP - 1) / NTP;
    // one block out of the register set x; its registers are refilled with block `nxt` (if below nblocks) as they fall free
    auto process = [&](u32x4 (&x)[PXT][KS], unsigned blk, unsigned nxt) {
        const bool more = nxt < p.nblocks;
        for (int ps = 0; ps < npass; ++ps) {
            const int j0 = ps * NTP;
            const int nj = min(NTP, NT - j0);  // 4 or 8 (Cout is a multiple of 64)
            const bool last = ps == npass - 1;
            f32x4 acc[PXT][NTP];
#pragma unroll
            for (int j = 0; j < NTP; ++j)
#pragma unroll
                for (int tt = 0; tt < PXT; ++tt) acc[tt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned char* wl = smem + ((size_t)j0 * KS * 64 + lane) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 wf[NTP];
#pragma unroll
                for (int j = 0; j < NTP; ++j)
                    if (j < 4 || nj == NTP) wf[j] = *(const u32x4*)(wl + (size_t)(j * KS + ks) * 1024);
#pragma unroll
                for (int j = 0; j < NTP; ++j)
                    if (j < 4 || nj == NTP) {
#pragma unroll
                        for (int tt = 0; tt < PXT; ++tt) Frag<__bf16>::mma(wf[j], x[tt][ks], acc[tt][j]);
                    }
                // the registers of this k-step are free after the last pass: the next data goes in at once
                if (last && more) {
#pragma unroll
                    for (int tt = 0; tt < PXT; ++tt) x[tt][ks] = xload(nxt, tt, ks);
                }
            }
            // epilogue: lane holds channels 32 q + 8 fq .. + 7 of pixel (tt, fr) in acc[tt][2q], acc[tt][2q + 1]; the bias is
            // added to the finished sum, as conv_igemm does (so the two kernels agree bit for bit)
#pragma unroll
            for (int tt = 0; tt < PXT; ++tt) {
                const unsigned m = blk * PXB + tt * 16 + fr;
                if (m < p.M) {
                    unsigned char* orow = p.out + (size_t)m * p.out_ldcB + (size_t)(p.out_coff + j0 * 16 + fq * 8) * 2;
#pragma unroll
                    for (int q = 0; q < NTP / 2; ++q)
                        if (q < 2 || nj == NTP) {
                            const float* bq = bias_l + j0 * 16 + q * 32 + fq * 8;
                            const f32x4 b0 = *(const f32x4*)bq, b1 = *(const f32x4*)(bq + 4);
                            float v[8];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                v[r] = act_fn<ACT>(acc[tt][2 * q][r] + b0[r]);
                                v[4 + r] = act_fn<ACT>(acc[tt][2 * q + 1][r] + b1[r]);
                            }
                            *(uint4*)(orow + q * 64) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                                  pack_bf16x2(v[6], v[7]));
                        }
                }
            }
        }
    };
    auto fill = [&](u32x4 (&x)[PXT][KS], unsigned b) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int tt = 0; tt < PXT; ++tt) x[tt][ks] = xload(min(b, p.nblocks - 1u), tt, ks);
    };
    if constexpr (PD == 1) {
        u32x4 xa[PXT][KS];
        fill(xa, blk);
        for (; blk < p.nblocks; blk += wstride) process(xa, blk, blk + wstride);
    } else {
        // two register sets: while one block is computed the next one is complete or landing and the one after is requested
        u32x4 xa[PXT][KS], xb[PXT][KS];
        fill(xa, blk);
        fill(xb, blk + wstride);
        for (; blk < p.nblocks; blk += 2 * wstride) {
            process(xa, blk, blk + 2 * wstride);
            if (blk + wstride < p.nblocks) process(xb, blk + wstride, blk + 3 * wstride);
        }
    }
}

bool stream_enabled() {  // CCNET_NO_STREAM_CONV=1 sends the thin layers to conv_igemm (tuning / bisecting aid)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_NO_STREAM_CONV");
        v = (e && atoi(e) != 0) ? 0 : 1;
    }
    return v == 1;
}

template <int KS, int ACT, int PXT, int SW, int PD>
int launch_stream(StreamQ p, hipStream_t st) {
    auto kernel = conv_stream_kernel<KS, ACT, PXT, SW, PD>;
    p.nblocks = (p.M + 16 * PXT - 1) / (16 * PXT);
    const int lds = (p.Cout >> 4) * KS * 1024 + p.Cout * 4;
    static int cus[16] = {0};
    static int attr[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[dev] = n;
    }
    if (attr[dev] < lds) {  // the largest request so far is what the function is allowed
        CC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));
        attr[dev] = 132 * 1024;
    }
    unsigned grid = (unsigned)cus[dev];
    const unsigned need = (p.nblocks + SW - 1) / SW;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(SW * 64), lds, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

// CCNET_STREAM_SHAPE: 0 = 32-pixel blocks, 8 waves; 1 (default) = 16-pixel blocks, 16 waves; 2 = as 1 with two blocks in flight
// per wave (12 waves for K = 256).  Measured on the 1x1 layers of `net` (64 x 640 x 640, sum over the eligible ones): 0 is 6 %,
// 1 is 12 % faster than conv_igemm, 2 no better than 0 -- resident waves count for more than bytes in flight per wave.
int stream_shape() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CCNET_STREAM_SHAPE");
        v = e ? atoi(e) : 1;
    }
    return v;
}

template <int KS, int ACT>
int launch_stream_shape(const StreamQ& p, hipStream_t st) {
    if (stream_shape() == 0) return launch_stream<KS, ACT, 2, 8, 1>(p, st);
    if (stream_shape() == 1) return launch_stream<KS, ACT, 1, 16, 1>(p, st);
    // the second register set costs 4 KS registers: 16 waves still fit for K <= 128, 12 for K = 256
    if constexpr (KS <= 4) return launch_stream<KS, ACT, 1, 16, 2>(p, st);
    else return launch_stream<KS, ACT, 1, 12, 2>(p, st);
}

template <int KS>
int launch_stream_act(const StreamQ& p, int act, hipStream_t st) {
    return act == CCNET_ACT_SILU ? launch_stream_shape<KS, CCNET_ACT_SILU>(p, st) : launch_stream_shape<KS, CCNET_ACT_NONE>(p, st);
}

}  // namespace

// Returns 1 and fills `launch` when the descriptor is one this kernel takes, 0 when conv_igemm should run it.  Called by
// ccnet_conv2d after it has validated the descriptor.
int ccnet_conv2d_stream_try(const ccnet_conv_desc* d, int kpad, ccnet_launch_fn* launch) {
    if (!stream_enabled()) return 0;
    if (d->dtype != CCNET_BF16 || d->out_dtype != CCNET_BF16) return 0;
    if (!(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0) || d->residual.ptr) return 0;
    if (d->act != CCNET_ACT_SILU && d->act != CCNET_ACT_NONE) return 0;
    int ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const ccnet_slice& sl = d->in[s];
        if (sl.shift < 0 || sl.shift > 3 || (d->H % (1 << sl.shift)) != 0 || (d->W % (1 << sl.shift)) != 0 || (sl.C % 32) != 0 || (sl.ldc % 8) != 0 || (sl.coff % 8) != 0 || ((uintptr_t)sl.ptr % 16) != 0) return 0;
        ctot += sl.C;
    }
    if (ctot != 128 && ctot != 256) return 0;  // K = 64 (one k-step pair) measures no better here than in conv_igemm
    if ((d->Cout % 64) != 0 || d->Cout > 256 || kpad != ctot) return 0;  // n-tiles come in fours: a pass has 4 or 8
    const ccnet_slice& o = d->out;
    if ((o.ldc % 8) != 0 || (o.coff % 8) != 0 || ((uintptr_t)o.ptr % 16) != 0) return 0;
    const long long M = (long long)d->B * d->H * d->W;
    // below ~64k pixels the prologue (every workgroup stages all the weights) outweighs the gain: 256 -> 256 at 64 x 20 x 20 is
    // 16.8 us in conv_igemm and 20.5 us here
    if (M < 65536 || M >= (1ll << 31) - 64) return 0;
    StreamQ p{};
    p.in0 = (const unsigned char*)d->in[0].ptr + (size_t)d->in[0].coff * 2;
    p.ldc0B = (unsigned)d->in[0].ldc * 2;
    p.KS0 = d->in[0].C / 32;
    if (d->nseg == 2) {
        p.in1 = (const unsigned char*)d->in[1].ptr + (size_t)d->in[1].coff * 2;
        p.ldc1B = (unsigned)d->in[1].ldc * 2;
    } else {
        p.in1 = p.in0;
        p.ldc1B = p.ldc0B;
    }
    p.M = (unsigned)M;
    p.shift0 = d->in[0].shift;
    p.shift1 = d->nseg == 2 ? d->in[1].shift : d->in[0].shift;
    p.H = d->H; p.W = d->W;
    {
        auto magic = [](unsigned dv, unsigned* mg, unsigned* sh) {
            unsigned s = 0;
            while ((1ull << s) < dv) ++s;
            *sh = s;
            *mg = (unsigned)((((1ull << s) - dv) << 32) / dv + 1);
        };
        magic((unsigned)d->W, &p.magicW, &p.shiftW);
        magic((unsigned)d->H, &p.magicH, &p.shiftH);
    }
    p.Cout = d->Cout;
    p.weight = (const unsigned char*)d->weight;
    p.Kpad = kpad;
    p.bias = d->bias;
    p.out = (unsigned char*)o.ptr;
    p.out_ldcB = (unsigned)o.ldc * 2;
    p.out_coff = o.coff;
    const int ks = ctot / 32, act = d->act;
    *launch = [p, ks, act](hipStream_t st, const float*) -> int {
        if (ks == 4) return launch_stream_act<4>(p, act, st);
        return launch_stream_act<8>(p, act, st);
    };
    return 1;
}
