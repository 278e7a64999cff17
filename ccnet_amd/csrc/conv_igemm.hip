// BN-folded NHWC implicit-GEMM convolution on gfx950 MFMA -- persistent, software-pipelined kernel.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = kh*kw*Ctot.  A (pixels x K) is gathered on the fly from one or
// two NHWC channel slices (second slice = torch.cat, shift = nearest upsampling), B is the packed weight [Cout_pad][Kpad].
// Block = 256 threads (4 waves), tile 128 x BN x 128 bytes of K per step (64 bf16 / 32 f32).
//
// Pipeline: a block walks a strided list of M tiles; the (tile, K-step) pairs form ONE flattened sequence with two LDS
// buffers.  In step j the block (1) issues the global loads of step j+1 -- which may already belong to the next tile --,
// (2) if step j-1 finished a tile, runs that tile's epilogue (registers -> global) while those loads fly,
// (3) feeds step j's MFMAs, (4) writes step j+1's registers to the idle buffer, (5) one barrier.  So neither the
// prologue of a tile nor its epilogue leaves the memory pipe empty, which is what the thin 1x1 layers (1-2 K steps)
// need, and the deep 3x3 layers pay one prologue per block instead of one per tile.
//
// Gather cost: per tile each thread precomputes, for its 4 rows, a byte pointer to tap (0,0) per input segment and a
// 64-bit tap-validity mask; a K step then costs a shift/test, a 64-bit add and a select per row.  Out-of-image taps load
// a safe address and are zeroed when written to LDS (no exec-mask branches, loads stay in flight across the MFMAs).
// LDS rows are 128 B = 8 chunks of 16 B; chunk c of row r lives at c ^ ((r>>1)&7): conflict free for the gfx950
// ds_read_b128 lane groups and for the ds_write_b128 fill.  bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32
// (exact fp32, k order permuted identically for A and B).  The weights are the MFMA "A" operand, so each lane ends up
// with 4 consecutive channels of one pixel per accumulator tile: the epilogue (bias + activation + residual) is
// register-only and stores 8 B (bf16) / 16 B (fp32) per lane -- an LDS-transposed epilogue cost as much as all MFMAs.
#include <stdlib.h>
#include <type_traits>
#include "conv_common.h"

int ccnet_conv2d_generic(const ccnet_conv_desc* d, ccnet_stream stream);  // conv_generic.hip
int ccnet_conv2d_patch_try(const ccnet_conv_desc* d, const double* span, int kpad, ccnet_launch_fn* launch);  // conv_patch.hip
int ccnet_conv2d_stream_try(const ccnet_conv_desc* d, int kpad, ccnet_launch_fn* launch);                      // conv_stream.hip

namespace {

// Tile BM x BN, WM x WN waves of 64 lanes, NST LDS stages (the DMA runs NST-1 K steps ahead of the MFMAs).
template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NST, int ACT>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvQ p) {
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int ES = (int)sizeof(T);
    constexpr int EPC = 16 / ES;  // elements per 16-B chunk
    constexpr int BK = BKB / ES;  // elements of K per step
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int RPI = NTHREADS / 8;  // tile rows covered by one DMA instruction of the whole block
    constexpr int A_LOADS = BM / RPI;
    constexpr int B_LOADS = (BN >= RPI) ? BN / RPI : 1;
    constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB, BUF_BYTES = A_BYTES + B_BYTES;
    static_assert(NST == 2 || BN >= RPI, "deep pipelines count DMA instructions per wave: every wave must load B");
    static_assert(A_LOADS + B_LOADS + 2 * MT * NT < 64, "vmcnt is a 6-bit counter");
    __shared__ __align__(16) unsigned char smem[NST * BUF_BYTES];

    unsigned tile0, tstride, tend, nslice;
    cc_tile_walk_sliced(p.ntiles, p.xcd, (unsigned)((p.Cout + BN - 1) / BN), tile0, tstride, tend, nslice);
    if (tile0 >= tend) return;
    const unsigned my_tiles = (tend - tile0 + tstride - 1) / tstride;
    const unsigned total = my_tiles * (unsigned)p.nk;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = (int)nslice * BN;
    // Tiles are filled by LDS-DMA (buffer_load_dwordx4 ... lds): one wave instruction writes 1 KB = 8 rows x 128 B of the
    // tile in lane order, so lane l of wave w owns PHYSICAL slot (l & 7) of rows 8w + (l >> 3) + RPI i and fetches the
    // LOGICAL chunk that the swizzle puts there (the swizzle goes on the source address); RPI is a multiple of 16, so
    // the swizzle term is the same for all of a thread's rows.
    const int lr = t >> 3;
    const int lc = (t & 7) ^ ((lr >> 1) & 7);
    const bool b_active = (BN >= 32) || (wave < BN / 8);  // wave-uniform

    // buffer resources: a fetch whose offset is out of range returns zeros, which is how padding taps, rows beyond M
    // and the zero-padded tail of K are filled -- no select, no exec mask
    const __amdgpu_buffer_rsrc_t rsrcA0 = make_rsrc(p.seg[0].ptr, p.seg[0].span);
    const __amdgpu_buffer_rsrc_t rsrcA1 = make_rsrc(p.seg[1].ptr, p.seg[1].span);
    const __amdgpu_buffer_rsrc_t rsrcB = make_rsrc(p.weight + (size_t)n0 * p.Kpad * ES, (unsigned)BN * (unsigned)p.Kpad * ES);
    constexpr unsigned OOB = 0x80000000u;  // spans are < 2^31 (checked on the host)
    unsigned boff[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) boff[i] = (unsigned)((lr + RPI * i) * p.Kpad + lc * EPC) * ES;

    // K cursor of this thread's chunk at step 0 of a tile
    int c_init, ky_init, kx_init;
    {
        const int kc = lc * EPC;
        const int tap = kc / p.Ctot;
        c_init = kc - tap * p.Ctot;
        ky_init = tap / p.kw;
        kx_init = tap - ky_init * p.kw;
    }

    // ---- gather side state (runs one step ahead of the MFMAs)
    unsigned rowoff[2][A_LOADS];  // byte offset of tap (0,0), channel 0 of each row in each segment (wraps for negative)
    unsigned long long rmask[A_LOADS];
    unsigned g_tile = tile0;
    int g_k = 0, c = c_init, ky = ky_init, kx = kx_init;

    auto setup_rows = [&](unsigned tile) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const unsigned m = tile * BM + lr + RPI * i;
            const bool valid = m < p.M;
            const unsigned mm = valid ? m : 0u;
            if (p.flat) {  // the thin 1x1 layers change tile every 1-4 K steps: no divisions, no tap loops for them
                rowoff[0][i] = mm * p.seg[0].ldcB;
                rowoff[1][i] = mm * p.seg[1].ldcB;
                rmask[i] = valid ? 1ull : 0ull;
                continue;
            }
            const unsigned q1 = fastdiv(mm, p.magicW, p.shiftW);
            const int ox = (int)(mm - q1 * (unsigned)p.Wo);
            const unsigned b = fastdiv(q1, p.magicH, p.shiftH);
            const int oy = (int)(q1 - b * (unsigned)p.Ho);
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const SegQ& sg = p.seg[s];
                const int pix = ((int)b * sg.Hs + (iy0 >> sg.shift)) * sg.Ws + (ix0 >> sg.shift);
                rowoff[s][i] = (unsigned)pix * sg.ldcB;
            }
            unsigned xbits = 0u;
            for (int x = 0; x < p.kw; ++x) xbits |= ((unsigned)(ix0 + x) < (unsigned)p.W ? 1u : 0u) << x;
            unsigned long long mk = 0ull;
            for (int y = 0; y < p.kh; ++y)
                if ((unsigned)(iy0 + y) < (unsigned)p.H) mk |= (unsigned long long)xbits << (y * p.kw);
            rmask[i] = valid ? mk : 0ull;
        }
    };
    // issue the LDS-DMA of the (tile, K step) under the gather cursor into buffer `buf`
    auto gather = [&](int buf) {
        // the host only sends two-segment problems here when a 128-B window never straddles the segments
        const bool s1 = __builtin_amdgcn_readfirstlane((int)(c >= p.C0)) != 0;
        const int cs = c - (s1 ? p.C0 : 0);
        const unsigned ldcB = s1 ? p.seg[1].ldcB : p.seg[0].ldcB;
        const unsigned off = (unsigned)(ky * p.W + kx) * ldcB + (unsigned)cs * ES;
        // in the zero-padded tail of K the tap index runs past kh*kw <= 63, where every mask bit is 0
        const int tap = min(ky * p.kw + kx, 63);
        const __amdgpu_buffer_rsrc_t rsrcA = s1 ? rsrcA1 : rsrcA0;
        unsigned char* As = smem + buf * BUF_BYTES + wave * 1024;
        unsigned char* Bs = As + A_BYTES;
#ifdef CC_EXP
        if (!((CC_EXP & 1) && __builtin_amdgcn_readfirstlane(tap) != 0))
#endif
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const unsigned ok = (unsigned)(rmask[i] >> tap) & 1u;
            const unsigned v = ok ? (s1 ? rowoff[1][i] : rowoff[0][i]) + off : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(As + i * RPI * BKB), 16, (int)v, 0, 0, 0);
        }
#ifdef CC_EXP
        if (!((CC_EXP & 2) && g_k != 0))
#endif
        if (b_active) {
            const int koff = g_k * BKB;
#pragma unroll
            for (int i = 0; i < B_LOADS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(Bs + i * RPI * BKB), 16, (int)boff[i], koff, 0, 0);
        }
    };
    // move the gather cursor to the next (tile, K step)
    auto advance = [&]() {
        if (++g_k == p.nk) {
            g_k = 0;
            g_tile += tstride;
            c = c_init; ky = ky_init; kx = kx_init;
            setup_rows(g_tile);
        } else {
            c += BK;
            while (c >= p.Ctot) {
                c -= p.Ctot;
                if (++kx == p.kw) { kx = 0; ++ky; }
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * BUF_BYTES;
        const unsigned char* Bs = As + A_BYTES;
        // both halves of the K step are requested up front: the second half's LDS latency hides behind the first MFMAs
        u32x4 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
        {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa0[i] = *(const u32x4*)(As + lds_off(wm * WTM + i * 16 + fr, fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb0[j] = *(const u32x4*)(Bs + lds_off(wn * WTN + j * 16 + fr, fq));
#pragma unroll
        for (int i = 0; i < MT; ++i) fa1[i] = *(const u32x4*)(As + lds_off(wm * WTM + i * 16 + fr, 4 + fq));
#pragma unroll
        for (int j = 0; j < NT; ++j) fb1[j] = *(const u32x4*)(Bs + lds_off(wn * WTN + j * 16 + fr, 4 + fq));
        }
        // keep the 16 LDS reads ahead of the MFMAs (the scheduler otherwise sinks each read next to its first use and
        // exposes the LDS latency four times per step)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) Frag<T>::mma(fb0[j], fa0[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) Frag<T>::mma(fb1[j], fa1[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // The MFMAs run with the weights as the "A" operand, so a lane's 4 accumulator registers of tile (i, j) are 4
    // CONSECUTIVE CHANNELS (n = nbase + j*16 + fq*4 + r) of ONE pixel (m = ... + i*16 + fr): bias + activation
    // (+ residual) happen in registers and each lane stores 4 channels (8 B bf16 / 16 B fp32) straight to global --
    // no LDS round trip, no barrier, and the epilogue overlaps the next tile's loads that are already in flight.
    const int nbase = n0 + wn * WTN + fq * 4;
    float breg[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nbase + j * 16 + r;
            breg[j][r] = (p.bias && n < p.Cout) ? p.bias[n] : 0.0f;
        }
    auto finish4 = [&](int i, int j, float* v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = acc[i][j][r] + breg[j][r];
            if constexpr (ACT == ACT_RUNTIME) v[r] = apply_act(x, p.act);
            else v[r] = act_fn<ACT>(x);
        }
    };
    auto add_res4 = [&](const unsigned char* rp, float* v) {
        if constexpr (ES == 2) {
            const uint2 rv = *(const uint2*)rp;
            v[0] += bf16_bits_to_f32(rv.x & 0xFFFFu); v[1] += bf16_bits_to_f32(rv.x >> 16);
            v[2] += bf16_bits_to_f32(rv.y & 0xFFFFu); v[3] += bf16_bits_to_f32(rv.y >> 16);
        } else {
            const float4 rv = *(const float4*)rp;
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
    };
    auto store4 = [&](unsigned char* op, const float* v) {
        if constexpr (sizeof(OutT) == 2) *(uint2*)op = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        else *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
    };
    // `full`: every row and channel of the tile exists and the vector path applies -> exactly MT*NT store instructions
    // per lane, which lets the main loop wait for the DMA issued before them with a counted vmcnt instead of draining
    // the stores (a drained store costs its whole round trip to L2 once per tile: ruinous for 1-4 K-step problems).
    auto epilogue = [&](unsigned tile, bool full) {
        if (full) {
            const unsigned m0 = tile * BM + wm * WTM + fr;
            unsigned char* obase = p.out_ptr + ((size_t)m0 * p.out_ldc + p.out_coff + nbase) * sizeof(OutT);
            const size_t ostep = (size_t)16 * p.out_ldc * sizeof(OutT);
            if (p.res_ptr) {
                // all residual loads first, so that they cost one round trip and not MT*NT
                const unsigned char* rbase = p.res_ptr + ((size_t)m0 * p.res_ldc + p.res_coff + nbase) * ES;
                const size_t rstep = (size_t)16 * p.res_ldc * ES;
                using ResV = typename std::conditional<ES == 2, uint2, float4>::type;
                ResV rv[MT][NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) rv[i][j] = *(const ResV*)(rbase + i * rstep + j * 16 * ES);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float v[4];
                        finish4(i, j, v);
                        if constexpr (ES == 2) {
                            v[0] += bf16_bits_to_f32(rv[i][j].x & 0xFFFFu); v[1] += bf16_bits_to_f32(rv[i][j].x >> 16);
                            v[2] += bf16_bits_to_f32(rv[i][j].y & 0xFFFFu); v[3] += bf16_bits_to_f32(rv[i][j].y >> 16);
                        } else {
                            v[0] += rv[i][j].x; v[1] += rv[i][j].y; v[2] += rv[i][j].z; v[3] += rv[i][j].w;
                        }
                        store4(obase + i * ostep + j * 16 * sizeof(OutT), v);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        float v[4];
                        finish4(i, j, v);
                        store4(obase + i * ostep + j * 16 * sizeof(OutT), v);
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const unsigned m = tile * BM + wm * WTM + i * 16 + fr;
                if (m < p.M) {
                    unsigned char* orow = p.out_ptr + ((size_t)m * p.out_ldc + p.out_coff) * sizeof(OutT);
                    const unsigned char* rrow = p.res_ptr ? p.res_ptr + ((size_t)m * p.res_ldc + p.res_coff) * ES : nullptr;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n = nbase + j * 16;
                        if (n >= p.Cout) continue;
                        float v[4];
                        finish4(i, j, v);
                        if (p.vec_out && n + 4 <= p.Cout) {
                            if (rrow) add_res4(rrow + (size_t)n * ES, v);
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
    };

    const bool n_full = p.vec_out && (n0 + BN <= p.Cout);
    constexpr int L = A_LOADS + B_LOADS;  // DMA instructions per wave and K step
    constexpr int S = MT * NT;            // store instructions per lane of a full tile's epilogue

    // prologue: the first NST-1 steps are put in flight, the first one is awaited
    setup_rows(g_tile);
    gather(0);
    unsigned issued = 1;  // K steps whose DMA has been issued
    if (NST == 3 && total > 1) {
        advance();
        gather(1);
        issued = 2;
        WAIT_VM(L);
    } else {
        WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    unsigned c_tile = tile0;  // tile the MFMAs are working on
    int c_k = 0;
    bool pending = false;
    unsigned pend_tile = 0;
    int stores_prev = 0;  // 1: the previous trip issued exactly S stores; 0: none; -1: an uncounted number
#ifdef CC_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    for (unsigned j = 0;; ++j) {  // one extra trip drains the last tile's epilogue (single inlined epilogue body)
        // step j + NST - 1 goes into the buffer that step j - 1 read, behind that step's barrier
        bool dma_now = false;
        if (issued < total) {
            advance();
            gather((int)(issued % NST));
            ++issued;
            dma_now = true;
        }
        STAMP(0);
        int stores_now = 0;
        if (pending) {
            const bool full = n_full && (pend_tile + 1) * BM <= p.M;
            epilogue(pend_tile, full);
            stores_now = full ? 1 : -1;
            pending = false;
        }
        STAMP(1);
        if (j == total) break;
        compute((int)(j % NST));
        STAMP(2);
        // Step j+1 must have landed before the barrier that lets every wave read it.  Vector memory operations retire
        // in order, so it is enough to allow as many outstanding operations as were issued AFTER its DMA: with two
        // stages that is this trip's stores; with three, also the previous trip's stores and this trip's DMA.
        // Raw s_barrier: __syncthreads() would add a full vmcnt(0).
        if (NST == 2) {
            if (stores_now == 1) WAIT_VM(S);
            else WAIT_VM(0);
        } else {
            const bool counted = stores_now >= 0 && stores_prev >= 0;
            const int ns = stores_now + stores_prev;
            if (!counted) WAIT_VM(0);
            else if (dma_now) {
                if (ns == 0) WAIT_VM(L);
                else if (ns == 1) WAIT_VM(L + S);
                else WAIT_VM(L + 2 * S);
            } else {
                if (ns == 0) WAIT_VM(0);
                else if (ns == 1) WAIT_VM(S);
                else WAIT_VM(2 * S);
            }
        }
        stores_prev = stores_now;
        STAMP(3);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(4);
        if (++c_k == p.nk) {
            c_k = 0;
            pending = true;
            pend_tile = c_tile;
            c_tile += tstride;
        }
    }
#ifdef CC_STAMP
    if (lane == 0 && wave == 1)
        for (int i = 0; i < 5; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
#endif
}

int pick_bn(int Cout) { return Cout <= 16 ? 16 : Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128; }

void magic_div(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *shift = s;
    *magic = (unsigned)((((1ull << s) - d) << 32) / d + 1);
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NST, int ACT>
int resident_blocks() {
    auto kernel = conv_igemm_kernel<T, OutT, BM, BN, WM, WN, NST, ACT>;
    // blocks of this instantiation that fit on the chip at once (cached per instantiation and device)
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, WM * WN * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cached[dev] = per_cu * cus;
    }
    return cached[dev];
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NST, int ACT>
int launch_act(ConvQ p, hipStream_t st) {
    auto kernel = conv_igemm_kernel<T, OutT, BM, BN, WM, WN, NST, ACT>;
    p.ntiles = (p.M + BM - 1) / BM;
    const int gy = (p.Cout + BN - 1) / BN;
    const int gx = cc_grid_x(resident_blocks<T, OutT, BM, BN, WM, WN, NST, ACT>(), gy, p.ntiles);
    hipLaunchKernelGGL(kernel, dim3(gx * gy), dim3(WM * WN * 64), 0, st, p);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}

template <typename T, typename OutT, int BM, int BN, int WM, int WN, int NST>
int launch_tile(const ConvQ& p, hipStream_t st) {
    // SiLU (every BaseConv), identity (prediction / attention layers) and ReLU (transformer FFN) get their own code
    if (p.act == CCNET_ACT_SILU) return launch_act<T, OutT, BM, BN, WM, WN, NST, CCNET_ACT_SILU>(p, st);
    if (p.act == CCNET_ACT_NONE) return launch_act<T, OutT, BM, BN, WM, WN, NST, CCNET_ACT_NONE>(p, st);
    if (p.act == CCNET_ACT_RELU) return launch_act<T, OutT, BM, BN, WM, WN, NST, CCNET_ACT_RELU>(p, st);
    return launch_act<T, OutT, BM, BN, WM, WN, NST, ACT_RUNTIME>(p, st);
}

// big: 256-row tiles, 8 waves, one block per CU with a three-stage ring (two K steps of DMA in flight); chosen when the
// problem has enough of them to fill the chip a few times over.  Otherwise 128-row tiles, 4 waves, two blocks per CU.
template <typename T, typename OutT>
int launch(const ConvQ& p, int bn, bool big, hipStream_t st) {
    if (big) {
        if (bn == 128) return launch_tile<T, OutT, 256, 128, 4, 2, 3>(p, st);
        return launch_tile<T, OutT, 256, 64, 4, 2, 3>(p, st);
    }
    switch (bn) {
        case 128: return launch_tile<T, OutT, 128, 128, 2, 2, 2>(p, st);
        case 64: return launch_tile<T, OutT, 128, 64, 2, 2, 2>(p, st);
        case 32: return launch_tile<T, OutT, 128, 32, 4, 1, 2>(p, st);
        default: return launch_tile<T, OutT, 128, 16, 4, 1, 2>(p, st);
    }
}

int conv_big_policy() {  // CCNET_CONV_BIG=0/1 forces the tile choice (tuning aid); unset = automatic
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("CCNET_CONV_BIG");
        v = e ? atoi(e) : -1;
    }
    return v;
}

}  // namespace

#ifdef CC_STAMP
extern "C" int ccnet_debug_stamps(unsigned long long* out8, int reset) {
    CC_HIP(hipDeviceSynchronize());
    CC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8));
    if (reset) {
        unsigned long long z[8] = {0};
        CC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
    }
    return CCNET_OK;
}
#endif

extern "C" int ccnet_conv_pack_dims(int dtype, int Cout, int K, int* Cout_pad, int* Kpad) {
    CC_REQUIRE(cc_dtype_ok(dtype) && Cout > 0 && K > 0 && Cout_pad && Kpad, "conv_pack_dims: bad arguments");
    int bn = pick_bn(Cout);
    int bk = BKB / cc_esize(dtype);
    *Cout_pad = ((Cout + bn - 1) / bn) * bn;
    *Kpad = ((K + bk - 1) / bk) * bk;
    return CCNET_OK;
}

extern "C" int ccnet_conv2d(const ccnet_conv_desc* d, ccnet_stream stream) {
    CC_REQUIRE(d != nullptr, "conv2d: null descriptor");
    CC_REQUIRE(cc_dtype_ok(d->dtype) && cc_dtype_ok(d->out_dtype), "conv2d: bad dtype");
    CC_REQUIRE(!(d->dtype == CCNET_F32 && d->out_dtype == CCNET_BF16), "conv2d: fp32 in / bf16 out is not built");
    CC_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv2d: nseg must be 1 or 2");
    CC_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "conv2d: bad geometry");
    CC_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->kh) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->kw) / d->stride + 1,
               "conv2d: Ho/Wo (%d,%d) do not match H,W,k,stride,pad", d->Ho, d->Wo);
    CC_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), "conv2d: too many input pixels");
    // cases the persistent kernel does not cover go to the generic one: an upsampled operand under a window or padding,
    // more than 63 taps, byte offsets beyond 32 bits
    // more than 63 taps, a tensor of 2 GiB or more (32-bit buffer offsets), or two segments whose boundary falls inside
    // a 128-byte K window (the LDS-DMA picks its segment per window, not per lane)
    bool generic = d->kh * d->kw > 63;
    const int es = cc_esize(d->dtype);
    const int bk = BKB / es;
    double span[2] = {0.0, 0.0};
    for (int s = 0; s < d->nseg; ++s) {
        const ccnet_slice& sl = d->in[s];
        if (sl.shift < 0 || sl.shift > 3) continue;  // rejected below
        if (sl.shift > 0 && (d->kh > 1 || d->kw > 1 || d->pad > 0)) generic = true;
        span[s] = (((double)d->B * (d->H >> sl.shift) * (d->W >> sl.shift) - 1.0) * sl.ldc + sl.C) * es;
        if (span[s] >= 2147483648.0) generic = true;
    }
    if (d->nseg == 2 && ((d->in[0].C % bk) != 0 || (d->in[1].C % bk) != 0)) generic = true;
    if (generic) return ccnet_conv2d_generic(d, stream);

    ConvQ p{};
    int ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const ccnet_slice& sl = d->in[s];
        CC_REQUIRE(cc_slice_vec_ok(sl, d->dtype), "conv2d: input slice %d is not 16-byte channel aligned (C=%d ldc=%d coff=%d)", s, sl.C, sl.ldc, sl.coff);
        CC_REQUIRE((d->H % (1 << sl.shift)) == 0 && (d->W % (1 << sl.shift)) == 0, "conv2d: H,W not divisible by the upsampling factor");
        p.seg[s] = SegQ{(const unsigned char*)sl.ptr + (size_t)sl.coff * es, (unsigned)(sl.ldc * es), sl.shift, d->H >> sl.shift, d->W >> sl.shift,
                        (unsigned)span[s]};
        ctot += sl.C;
    }
    if (d->nseg == 1) p.seg[1] = p.seg[0];
    p.nseg = d->nseg;
    p.Ctot = ctot;
    p.C0 = d->nseg == 2 ? d->in[0].C : ctot;
    int coutp = 0, kpad = 0;
    ccnet_conv_pack_dims(d->dtype, d->Cout, d->kh * d->kw * ctot, &coutp, &kpad);
    CC_REQUIRE(d->weight && ((uintptr_t)d->weight % 16) == 0, "conv2d: weight pointer must be 16-byte aligned");
    CC_REQUIRE(d->Kpad == kpad && d->Cout_pad == coutp, "conv2d: weight packed as [%d][%d], expected [%d][%d]", d->Cout_pad, d->Kpad, coutp, kpad);
    CC_REQUIRE((double)coutp * kpad * es < 2147483648.0, "conv2d: packed weight of 2 GiB or more");
    p.H = d->H; p.W = d->W; p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    const long long M = (long long)d->B * d->Ho * d->Wo;
    CC_REQUIRE(M > 0 && M < (1ll << 31) - 256, "conv2d: too many output pixels");
    p.M = (unsigned)M;
    magic_div((unsigned)d->Wo, &p.magicW, &p.shiftW);
    magic_div((unsigned)d->Ho, &p.magicH, &p.shiftH);
    p.weight = (const unsigned char*)d->weight;
    p.Kpad = kpad;
    p.nk = kpad / (BKB / es);
    CC_REQUIRE((double)M / 128 * p.nk < 4.0e9, "conv2d: problem too large");
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
        int rg = 16 / es;
        vec = vec && (r.ldc % rg) == 0 && (r.coff % rg) == 0 && ((uintptr_t)r.ptr % 16) == 0 && (r.ldc % 8) == 0 && (r.coff % 8) == 0;
    }
    p.vec_out = vec ? 1 : 0;
    p.xcd = cc_xcd_remap();
    p.flat = (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && p.seg[0].shift == 0 && p.seg[1].shift == 0) ? 1 : 0;
    const double flops = 2.0 * (double)M * d->Cout * d->kh * d->kw * ctot;
    const double bytes = ((double)d->B * d->H * d->W * ctot + (double)d->Cout * d->kh * d->kw * ctot) * es +
                         (double)M * d->Cout * (cc_esize(d->out_dtype) + (d->residual.ptr ? es : 0));
    {   // thin 1x1 layers (K, N <= 256): weights resident in LDS, pixels streamed
        ccnet_launch_fn stream_fn;
        if (o.ptr && o.shift == 0 && ccnet_conv2d_stream_try(d, kpad, &stream_fn)) return ccnet_submit(stream_fn, stream, CCNET_OP_CONV, flops, bytes);
    }
    {   // 3x3 s1 p1 and SpaceToDepth (6x6 s2 p2) layers on 16-aligned widths: LDS halo-patch kernel
        ccnet_launch_fn patch_fn;
        if (ccnet_conv2d_patch_try(d, span, kpad, &patch_fn)) return ccnet_submit(patch_fn, stream, CCNET_OP_CONV, flops, bytes);
    }
    const int bn = pick_bn(d->Cout);
    const int dt = d->dtype, odt = d->out_dtype;
    const long long tiles256 = ((M + 255) / 256) * ((d->Cout + bn - 1) / bn);
    (void)tiles256;  // the 256-row / 3-stage variant measures slower than two 128-row blocks per CU so far: opt-in only
    const bool big = bn >= 64 && conv_big_policy() == 1;
    return ccnet_submit(
        [p, bn, big, dt, odt](hipStream_t st, const float*) -> int {
            if (dt == CCNET_BF16) return odt == CCNET_BF16 ? launch<__bf16, __bf16>(p, bn, big, st) : launch<__bf16, float>(p, bn, big, st);
            return launch<float, float>(p, bn, big, st);
        },
        stream, CCNET_OP_CONV, flops, bytes);
}
