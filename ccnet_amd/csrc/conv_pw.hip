// 1x1 Conv-BN-act (BaseConv k1 s1, model_block.py:37-58; torch.cat / nn.Upsample on its input side, network/net.py:208-210), bf16,
// built like conv_halo.hip: the WEIGHTS never touch LDS -- every wave streams its MFMA A-operand fragments from L2 into a register
// ring in consumption order (ops.py:PackedPw) -- and only the pixels are staged in LDS, where the two channel-waves of a workgroup
// share them.  Workgroup tile 256 pixels x 128 (or 64) output channels, wave tile 128 x 64 (8 x 4 MFMA tiles), K in chunks of 64
// input channels = two k-steps per barrier.
//
// Why (round 3): the 1x1 layers with K >= 256 on the 20 x 20 and 40 x 40 maps ran at 350-450 TFLOP/s on conv_igemm / conv_stream.
// conv_igemm's K step is 32 MFMAs per wave between two barriers, with 16 LDS fragment reads and a weight tile that goes
// L2 -> LDS -> registers; its K loop of 4-16 steps with one step of DMA look-ahead is bound by the latency of that chain.  Here a
// barrier interval is 64 MFMAs per wave with 16 pixel-fragment reads, the weight ring holds two k-steps of fragments in flight
// (8 KB per wave), and the next chunk's pixels are requested a whole chunk ahead.
//
// LDS: two chunk buffers of [2 k-steps][256 rows][64 B] (row r keeps its four 16-byte pieces at slot c ^ ((r >> 1) & 3)) and a
// per-tile table of source pixels (a lane's four rows x two segments), 72 KB per workgroup, two workgroups per CU.
// Inputs: one or two bf16 slices (channel concat), each possibly nearest-upsampled (shift), each a multiple of 64 channels.
// Accumulation order: k ascending in steps of 32 with fp32 accumulators from zero, bias added to the finished sum -- the order
// of conv_igemm / conv_stream, so the kernels agree bit for bit and the choice between them never shows in an output.
#include <stdlib.h>
#include <mutex>
#include "conv_common.h"

namespace {

constexpr int PW_ROWS = 256;                  // pixels per workgroup tile
constexpr int PW_ROWB = 64;                   // bytes per LDS row (32 channels)
constexpr int PW_SUB = PW_ROWS * PW_ROWB;     // one k-step of the tile: 16 KB
constexpr int PW_BUF = 2 * PW_SUB;            // one chunk (64 channels)
constexpr int PW_RING = 8;                    // weight fragments in flight per wave: two k-steps x four n-tiles
constexpr int PW_TAB = 2 * 4 * 256 * 4;       // [segment][row slot][thread] source pixels
constexpr int PW_LDS = 2 * PW_BUF + PW_TAB;

struct PwQ {
    SegQ seg[2];
    int nseg, C0;
    int B, H, W;
    unsigned M, HW;
    int Cout, nchunks;                        // nchunks = Cin / 64
    unsigned magicW, shiftW, magicHW, shiftHW;
    unsigned ntiles;
    const unsigned char* wstream;             // [Cout / 64][Cin / 32 * 4 + PW_RING][64 lanes][16 B]
    unsigned long long wstride;
    const float* bias;
    int act;
    const unsigned char* res_ptr;
    int res_ldc, res_coff;
    unsigned char* out_ptr;
    int out_ldc, out_coff;
    int any_shift;
    int xcd;
};

// One k-step of the current chunk: MT pixel fragments, then NT x MT MFMAs, each weight fragment leaving the ring as it is used
#define PW_STEP(S)                                                                                                                 \
    {                                                                                                                              \
        u32x4 fa[MT];                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[i] = *(const u32x4*)(ptw + (S) * PW_SUB + i * 16 * PW_ROWB);             \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                           \
            constexpr int slot = (S) * NT;                                                                                         \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ring[slot + j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0); \
            ring[slot + j] = *(const u32x4*)(wp + PW_RING * 1024 + loff);                                                          \
            wp += 1024;                                                                                                            \
        }                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
    }

template <int BN, int ACT>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const PwQ p) {
    constexpr int MT = BN == 128 ? 8 : 4, NT = 4;
    constexpr int PXW = MT * 16;   // pixels per wave
    extern __shared__ __align__(16) unsigned char smem[];

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

    // ---- tile fill.  Wave w fetches rows 64 w .. 64 w + 63 of both k-steps: piece (i, s) = 16 rows x 64 B; lane l writes row
    // 64 w + 16 i + (l >> 2), physical slot l & 3, and so fetches logical piece (l & 3) ^ ((row >> 1) & 3)
    const unsigned prc = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) * 16);
    int* pixtab = (int*)(smem + 2 * PW_BUF);
    auto fill_pixels = [&](unsigned tile) {   // source pixel of the lane's four rows, per segment (-1: zero row)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned m = tile * (unsigned)PW_ROWS + (unsigned)(wave * 64 + i * 16 + (lane >> 2));
            int v0 = -1, v1 = -1;
            if (tile < tend && m < p.M) {
                if (p.any_shift) {
                    const unsigned b = fastdiv(m, p.magicHW, p.shiftHW), rem = m - b * p.HW;
                    const unsigned y = fastdiv(rem, p.magicW, p.shiftW), x = rem - y * (unsigned)p.W;
                    v0 = (int)((b * (unsigned)p.seg[0].Hs + (y >> p.seg[0].shift)) * (unsigned)p.seg[0].Ws + (x >> p.seg[0].shift));
                    v1 = (int)((b * (unsigned)p.seg[1].Hs + (y >> p.seg[1].shift)) * (unsigned)p.seg[1].Ws + (x >> p.seg[1].shift));
                } else {
                    v0 = v1 = (int)m;
                }
            }
            pixtab[i * 256 + t] = v0;
            pixtab[(4 + i) * 256 + t] = v1;
        }
    };
    const unsigned ldcB0 = p.seg[0].ldcB, ldcB1 = p.seg[1].ldcB;
    auto chunk_src = [&](int cc, bool& s1, unsigned& soff) {   // wave-uniform, kept in scalar registers
        const int c = cc * 64;
        s1 = __builtin_amdgcn_readfirstlane((int)(p.nseg == 2 && c >= p.C0)) != 0;
        soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(c - (s1 ? p.C0 : 0)) * 2u));
    };
    auto fill_chunk = [&](int buf, bool s1, unsigned soff) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = pixtab[((s1 ? 4 : 0) + i) * 256 + t];
            const unsigned v = px >= 0 ? (unsigned)px * (s1 ? ldcB1 : ldcB0) + prc : OOB;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                lds_void* dst = (lds_void*)(smem + buf * PW_BUF + s * PW_SUB + (wave * 4 + i) * 1024);
                if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA1, dst, 16, (int)v, (int)(soff + s * 64), 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA0, dst, 16, (int)v, (int)(soff + s * 64), 0, 0);
            }
        }
    };

    // ---- weight ring
    const unsigned char* wp = p.wstream + (size_t)n64 * p.wstride;   // wave uniform
    const int loff = lane * 16;
    u32x4 ring[PW_RING];
#pragma unroll
    for (int k = 0; k < PW_RING; ++k) ring[k] = *(const u32x4*)(wp + k * 1024 + loff);
    const long long wrewind = (long long)p.nchunks * PW_RING * 1024;   // 8 fragments per chunk

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // pixel fragments: row = wm * PXW + i * 16 + fr; the row's swizzle term depends on fr only (the rest is a multiple of 16)
    const int lanec = (wm * PXW + fr) * PW_ROWB + ((fq ^ ((fr >> 1) & 3)) << 4);

    const int nbase = n64 * 64 + fq * 4;
    auto epilogue = [&](unsigned tile) {
        float4 bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = p.bias ? *(const float4*)(p.bias + nbase + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const unsigned m = tile * (unsigned)PW_ROWS + (unsigned)(wm * PXW + i * 16 + fr);
            if (m < p.M) {
                unsigned char* orow = p.out_ptr + ((size_t)m * p.out_ldc + p.out_coff + nbase) * 2;
                uint2 rv[NT];
                if (p.res_ptr) {
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
        // nothing this lambda loaded may still be pending where its path joins the main loop (conv_halo.hip): one drain per tile
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    };

    // ---- prologue: chunk 0 of the first tile
    int pf_cc = 0;
    unsigned pf_tile = tile0;
    bool pf_s1;
    unsigned pf_soff;
    fill_pixels(pf_tile);
    chunk_src(pf_cc, pf_s1, pf_soff);
    fill_chunk(0, pf_s1, pf_soff);
    auto pf_advance = [&]() {
        if (++pf_cc == p.nchunks) {
            pf_cc = 0;
            pf_tile += tstride;
            fill_pixels(pf_tile);
        }
        chunk_src(pf_cc, pf_s1, pf_soff);
    };
    pf_advance();
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0) (builtins, not inline asm: the compiler's wait bookkeeping must see them)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int cur_cc = 0;
    unsigned cur_tile = tile0, pend_tile = 0;
    bool pending = false;
    for (unsigned ci = 0; ci < total_chunks; ++ci) {
        const bool next_chunk = ci + 1 < total_chunks;
        const unsigned char* ptw = smem + (ci & 1) * PW_BUF + lanec;
        if (pending) {
            epilogue(pend_tile);
            pending = false;
        }
        if (next_chunk) fill_chunk((int)((ci + 1) & 1), pf_s1, pf_soff);
        PW_STEP(0) PW_STEP(1)
        if (next_chunk) pf_advance();
        if (++cur_cc == p.nchunks) {
            cur_cc = 0;
            pending = true;
            pend_tile = cur_tile;
            cur_tile += tstride;
            wp -= wrewind;   // the stream ends with a copy of its first PW_RING fragments: the ring already holds the next tile's
        }
        // The next chunk's pieces were issued before this chunk's 8 ring loads, and vector-memory operations return in order: with
        // at most PW_RING of them outstanding the pieces have landed.
        static_assert(PW_RING == 8, "the immediate below encodes vmcnt(PW_RING)");
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (pending) epilogue(pend_tile);
}

template <int BN, int ACT>
int launch_pw_k(const PwQ& p, hipStream_t st) {
    auto kernel = conv_pw_kernel<BN, ACT>;
    static std::mutex mu;
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int resident = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (cached[dev] == 0) {
            CC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS));
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 256, PW_LDS) != hipSuccess || per_cu < 1) per_cu = 1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            cached[dev] = per_cu * cus;
        }
        resident = cached[dev];
    }
    const int gy = p.Cout / BN;
    const int gx = cc_grid_x(resident, gy, p.ntiles);
    hipLaunchKernelGGL(kernel, dim3(gx * gy), dim3(256), PW_LDS, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

template <int BN>
int launch_pw(const PwQ& p, hipStream_t st) {
    if (p.act == CCNET_ACT_SILU) return launch_pw_k<BN, CCNET_ACT_SILU>(p, st);
    if (p.act == CCNET_ACT_NONE) return launch_pw_k<BN, CCNET_ACT_NONE>(p, st);
    return launch_pw_k<BN, ACT_RUNTIME>(p, st);
}

void pw_magic(unsigned d, unsigned* magic, unsigned* shift) {   // n / d for n < 2^31 as (mulhi(n, magic) + n) >> shift
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *shift = s;
    *magic = (unsigned)((((1ull << s) - d) << 32) / d + 1);
}

}  // namespace

extern "C" size_t ccnet_conv_pw_stream_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0 || (Cin % 64) != 0 || (Cout % 64) != 0) return 0;
    return (size_t)(Cout / 64) * ((size_t)(Cin / 32) * 4 + PW_RING) * 512;
}

extern "C" int ccnet_conv_pw(const ccnet_slice* in, int nseg, int B, int H, int W, int Cout, const void* wstream_dev, const float* bias_dev,
                             int act, ccnet_slice residual, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(in && (nseg == 1 || nseg == 2) && wstream_dev, "conv_pw: null pointer / bad segment count");
    CC_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && (Cout % 64) == 0, "conv_pw: bad sizes (B=%d H=%d W=%d Cout=%d; Cout must be a multiple of 64)", B, H, W, Cout);
    CC_REQUIRE(act >= CCNET_ACT_NONE && act <= CCNET_ACT_LRELU, "conv_pw: unknown activation %d", act);
    CC_REQUIRE(((uintptr_t)wstream_dev % 16) == 0 && ((uintptr_t)bias_dev % 16) == 0, "conv_pw: weights / bias must be 16-byte aligned");
    const long long M = (long long)B * H * W;
    CC_REQUIRE(M < (1ll << 30), "conv_pw: too many pixels");
    PwQ p{};
    int ctot = 0;
    for (int s = 0; s < nseg; ++s) {
        const ccnet_slice& sl = in[s];
        CC_REQUIRE(cc_slice_vec_ok(sl, CCNET_BF16) && sl.shift <= 2 && (sl.C % 64) == 0 && (H % (1 << sl.shift)) == 0 && (W % (1 << sl.shift)) == 0,
                   "conv_pw: input slice %d must be bf16, 16-byte aligned, a multiple of 64 channels, upsampled by at most 4 (C=%d ldc=%d coff=%d shift=%d)",
                   s, sl.C, sl.ldc, sl.coff, sl.shift);
        const int Hs = H >> sl.shift, Ws = W >> sl.shift;
        const double span = (((double)B * Hs * Ws - 1.0) * sl.ldc + sl.C) * 2.0;
        CC_REQUIRE(span < 2147483648.0, "conv_pw: input tensor of 2 GiB or more");
        p.seg[s] = SegQ{(const unsigned char*)sl.ptr + (size_t)sl.coff * 2, (unsigned)(sl.ldc * 2), sl.shift, Hs, Ws, (unsigned)span};
        p.any_shift |= sl.shift != 0;
        ctot += sl.C;
    }
    if (nseg == 1) p.seg[1] = p.seg[0];
    p.nseg = nseg;
    p.C0 = nseg == 2 ? in[0].C : ctot;
    p.B = B; p.H = H; p.W = W;
    p.M = (unsigned)M; p.HW = (unsigned)(H * W);
    p.Cout = Cout;
    p.nchunks = ctot / 64;
    pw_magic((unsigned)W, &p.magicW, &p.shiftW);
    pw_magic(p.HW, &p.magicHW, &p.shiftHW);
    p.ntiles = (unsigned)((M + PW_ROWS - 1) / PW_ROWS);
    p.wstream = (const unsigned char*)wstream_dev;
    p.wstride = ((unsigned long long)(ctot / 32) * 4 + PW_RING) * 1024ull;
    p.bias = bias_dev;
    p.act = act;
    CC_REQUIRE(out.ptr && out.shift == 0 && out.C == Cout && out.coff >= 0 && out.ldc >= out.coff + out.C && (out.ldc % 4) == 0 && (out.coff % 4) == 0 &&
                   ((uintptr_t)out.ptr % 8) == 0, "conv_pw: bad output slice (C=%d ldc=%d coff=%d)", out.C, out.ldc, out.coff);
    p.out_ptr = (unsigned char*)out.ptr; p.out_ldc = out.ldc; p.out_coff = out.coff;
    if (residual.ptr) {
        CC_REQUIRE(residual.shift == 0 && residual.C == Cout && residual.coff >= 0 && residual.ldc >= residual.coff + residual.C && (residual.ldc % 4) == 0 &&
                       (residual.coff % 4) == 0 && ((uintptr_t)residual.ptr % 8) == 0, "conv_pw: bad residual slice");
        p.res_ptr = (const unsigned char*)residual.ptr; p.res_ldc = residual.ldc; p.res_coff = residual.coff;
    }
    p.xcd = cc_xcd_remap();
    const double flops = 2.0 * (double)M * Cout * ctot;
    double bytes = (double)Cout * ctot * 2.0 + (double)M * Cout * (2.0 + (residual.ptr ? 2.0 : 0.0));
    for (int s = 0; s < nseg; ++s) bytes += (double)M / (double)(1 << (2 * in[s].shift)) * in[s].C * 2.0;
    const int bn = (Cout % 128) == 0 ? 128 : 64;
    return ccnet_submit([p, bn](hipStream_t st, const float*) -> int { return bn == 128 ? launch_pw<128>(p, st) : launch_pw<64>(p, st); }, stream,
                        CCNET_OP_CONV, flops, bytes);
}
