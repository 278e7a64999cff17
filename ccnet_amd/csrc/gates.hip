// Attention gates: CBAM (channel + spatial), ECA, and the ASFF softmax fusion.  All HBM-bound elementwise /
// reduction kernels over NHWC; reductions are two-stage with a fixed order, so results are run-to-run identical.
//
// CBAM passes (dependency-forced 3 reads + 1 write of x):
//   1 pool_partial : per (image, pixel chunk) sum and max per channel            -> scratch
//   2 cbam_mlp     : finish avg/max, shared MLP, sigmoid (one block per image)    -> ca[B][C]
//   3 cbam_stats   : per pixel mean_c / max_c of x*ca                             -> stats[B][H][W][2]
//   4 cbam_apply   : sa = sigmoid(conv kxk(stats)); out = x*ca*sa
// ECA: pool_partial + eca_gate + scale.
// Pass 2 was tried inside pass 1 (round 3: the block that finishes an image last, found through a per-image agent-scope counter,
// runs the MLP).  Publishing the partials across XCDs (sc1 stores, the wait for them, the counter's round trip) holds every
// pooling block's CU slot about twice as long: pool 22 -> 42 us at 80x80x128 x 64 images, against the 11 us launch it saves.
// Measured slower in total (0.217 against 0.192 ms for the three CBAMs of `net`), so the gate stays a launch of its own.
#include <math.h>
#include <mutex>
#include "common.h"

namespace {

constexpr int POOL_CHUNK = 128;  // pixels per partial block (small: a 20x20 map must still spread over the chip)

// scratch layout (floats): partial[B][nchunk][2][C] | ca[B][C] | stats[B][H*W][2]
struct GateScratch {
    size_t partial, ca, stats, total;
    int nchunk;
};
GateScratch gate_scratch(int B, int H, int W, int C) {
    GateScratch s;
    s.nchunk = (H * W + POOL_CHUNK - 1) / POOL_CHUNK;
    s.partial = 0;
    s.ca = s.partial + (size_t)B * s.nchunk * 2 * C;
    s.stats = s.ca + (size_t)B * C;
    s.total = s.stats + (size_t)B * H * W * 2;
    s.total = (s.total + 3) & ~(size_t)3;
    return s;
}

// avg[c] (and mx[c] when given) of one image from its chunk partials [chunk][2][C].  A thread owns a (sum | max, 4 channels)
// pair; with few pairs the chunks are split over nt / pairs thread groups and combined in group order through tmp (4 * nt
// floats, may be null: no split).  The summation order depends on C and nt only.
__device__ inline void finish_pool(const float* pb, int nchunk, int C, int HW, float* avg, float* mx, float* tmp, int t, int nt) {
    const int Q = C / 4, pairs = mx ? 2 * Q : Q;
    const int parts = (tmp != nullptr && 2 * pairs <= nt) ? nt / pairs : 1;
    for (int pr0 = 0; pr0 < pairs; pr0 += nt) {   // one pass unless there are more pairs than threads
        const int pr = parts > 1 ? t % pairs : pr0 + t;
        const int part = parts > 1 ? t / pairs : 0;
        const bool live = pr < pairs && part < parts;
        const int kind = live ? pr / Q : 0, quad = live ? pr - kind * Q : 0;
        float4 acc = kind ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live) {
            // 8 chunk rows in flight (the index is clamped so that no load sits behind a branch; surplus rows are not used)
            for (int k0 = part; k0 < nchunk; k0 += 8 * parts) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = min(k0 + u * parts, nchunk - 1);
                    v[u] = *(const float4*)(pb + (size_t)(k * 2 + kind) * C + quad * 4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u * parts < nchunk) {
                        if (kind) {
                            acc.x = fmaxf(acc.x, v[u].x); acc.y = fmaxf(acc.y, v[u].y); acc.z = fmaxf(acc.z, v[u].z); acc.w = fmaxf(acc.w, v[u].w);
                        } else {
                            acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                        }
                    }
                }
            }
        }
        if (parts > 1) {
            if (live) *(float4*)(tmp + (size_t)(part * pairs + pr) * 4) = acc;
            __syncthreads();
            if (t < pairs) {
                acc = *(const float4*)(tmp + (size_t)t * 4);
                for (int q = 1; q < parts; ++q) {
                    const float4 v = *(const float4*)(tmp + (size_t)(q * pairs + t) * 4);
                    if (kind) {
                        acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w);
                    } else {
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                }
            }
        }
        if (live && part == 0) {
            float* dst = kind ? mx + quad * 4 : avg + quad * 4;
            if (kind) { dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w; }
            else { dst[0] = acc.x / (float)HW; dst[1] = acc.y / (float)HW; dst[2] = acc.z / (float)HW; dst[3] = acc.w / (float)HW; }
        }
    }
    __syncthreads();
}

// CBAM channel gate of one image: ca = sigmoid(fc2 relu(fc1 avg) + fc2 relu(fc1 max)).  sm: avg[C] max[C] hid[2*Cr]
__device__ inline void cbam_mlp_body(const float* pb, int nchunk, int C, int Cr, int HW, const float* fc1, const float* fc2, float* ca_b, float* sm,
                                     float* tmp, int t, int nt) {
    float* avg = sm;
    float* mx = sm + C;
    float* hid = sm + 2 * C;
    finish_pool(pb, nchunk, C, HW, avg, mx, tmp, t, nt);
    const int lane = t & 63, wave = t >> 6;
    // hidden unit j (< Cr: on the average, >= Cr: on the maximum); 8 units per wave pass so that 8 rows of fc1 are in flight
    for (int j0 = wave * 8; j0 < 2 * Cr; j0 += (nt >> 6) * 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.0f;
        // row pointers first (clamped, so that no load sits behind a branch: a load under `if` is waited for on the spot)
        const float* wrow[8];
        const float* vsrc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = min(j0 + u, 2 * Cr - 1);
            wrow[u] = fc1 + (size_t)(j < Cr ? j : j - Cr) * C;
            vsrc[u] = j < Cr ? avg : mx;
        }
#pragma unroll 4
        for (int c = lane; c < C; c += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fmaf(wrow[u][c], vsrc[u][c], a[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float r = wave_sum(a[u]);
            if (lane == 0 && j0 + u < 2 * Cr) hid[j0 + u] = fmaxf(r, 0.0f);
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += nt) {
        const float* wr = fc2 + (size_t)c * Cr;
        float a = 0.0f, m = 0.0f;
        if ((Cr & 3) == 0) {  // rows of fc2 are 16-byte aligned then
#pragma unroll 4
            for (int j = 0; j < Cr; j += 4) {
                const float4 w4 = *(const float4*)(wr + j);
                a = fmaf(w4.x, hid[j], a); a = fmaf(w4.y, hid[j + 1], a); a = fmaf(w4.z, hid[j + 2], a); a = fmaf(w4.w, hid[j + 3], a);
                m = fmaf(w4.x, hid[Cr + j], m); m = fmaf(w4.y, hid[Cr + j + 1], m); m = fmaf(w4.z, hid[Cr + j + 2], m); m = fmaf(w4.w, hid[Cr + j + 3], m);
            }
        } else {
            for (int j = 0; j < Cr; ++j) { a = fmaf(wr[j], hid[j], a); m = fmaf(wr[j], hid[Cr + j], m); }
        }
        ca_b[c] = sigmoidf_(a + m);
    }
}

// ECA gate of one image: avg over pixels -> 3-tap conv over the channel axis (zero padded) -> sigmoid.  sm: avg[C]
__device__ inline void eca_gate_body(const float* pb, int nchunk, int C, int HW, const float* w3, float* gate_b, float* sm, float* tmp, int t, int nt) {
    float* avg = sm;
    finish_pool(pb, nchunk, C, HW, avg, nullptr, tmp, t, nt);
    for (int c = t; c < C; c += nt) {
        float l = c > 0 ? avg[c - 1] : 0.0f, r = c + 1 < C ? avg[c + 1] : 0.0f;
        gate_b[c] = sigmoidf_(w3[0] * l + w3[1] * avg[c] + w3[2] * r);
    }
}

// Sum and max per channel over the pixels of one (image, chunk): the work of one group of 256 threads (t = index in the group).
// `red`: the group's [2][256][8] floats.  Contains block-wide barriers: every thread of the block calls it the same number of times;
// a group without a chunk passes active = false.  dst: [2][C] partials of this chunk.
template <typename T>
__device__ inline void pool_partial_body(const unsigned char* img, int ldc, int C, int HW, int chunk, bool active, float* red, float* dst, int t) {
    const int G = C / 8;
    const int p0 = chunk * POOL_CHUNK;
    const int p1 = min(HW, p0 + POOL_CHUNK);
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int TG = min(G - g0, 256);
        const int PL = 256 / TG;
        const int g = g0 + t % TG, pl = t / TG;
        float s[8], m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.0f; m[e] = -INFINITY; }
        if (active && pl < PL) {
            const unsigned char* src = img + (size_t)(g * 8) * sizeof(T);
            // 4 pixels in flight per thread (clamped index: no load behind a branch; a surplus pixel is loaded and not used)
            for (int pb = p0 + pl; pb < p1; pb += 4 * PL) {
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Chan8<T>::load(src + (size_t)min(pb + u * PL, p1 - 1) * ldc * sizeof(T), v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (pb + u * PL < p1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s[e] += v[u][e]; m[e] = fmaxf(m[e], v[u][e]); }
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[t * 8 + e] = s[e]; red[(256 + t) * 8 + e] = m[e]; }
        __syncthreads();
        // every thread finishes output values (sum | max, channel): PL reads each, in pixel-lane order
        const int nout = TG * 8;
        if (active) {
            for (int o = t; o < 2 * nout; o += 256) {
                const int kind = o >= nout ? 1 : 0, rem = o - kind * nout;
                const float* r = red + kind * 256 * 8 + rem;
                float acc = r[0];
                if (kind) {
                    for (int q = 1; q < PL; ++q) acc = fmaxf(acc, r[q * nout]);
                } else {
                    for (int q = 1; q < PL; ++q) acc += r[q * nout];
                }
                dst[kind * C + g0 * 8 + rem] = acc;
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pool_partial_kernel(const unsigned char* in, int ldc, int coff, int C, int HW, int nchunk, float* partial) {
    __shared__ float red[2 * 256 * 8];
    const int b = blockIdx.y, chunk = blockIdx.x;
    pool_partial_body<T>(in + (((size_t)b * HW) * ldc + coff) * sizeof(T), ldc, C, HW, chunk, true, red, partial + (((size_t)b * nchunk + chunk) * 2) * C,
                         threadIdx.x);
}

// one block per image; dynamic LDS: avg[C] max[C] hid[2*Cr] | 4 * blockDim floats for the split chunk reduction
__global__ __launch_bounds__(1024) void cbam_mlp_kernel(const float* partial, int nchunk, int C, int Cr, int HW, const float* fc1, const float* fc2,
                                                       float* ca) {
    extern __shared__ float sm[];
    const int b = blockIdx.x;
    // the weights do not depend on the pooling result: one load per 128-byte line starts them on their way into L2 / L1 while the
    // partials are reduced (every stage of this kernel is a chain of cold misses otherwise: 17 us at C = 512)
    float warm = 0.0f;
    for (int i = threadIdx.x * 32; i < Cr * C; i += blockDim.x * 32) warm += fc1[i] + fc2[i];
    cbam_mlp_body(partial + (size_t)b * nchunk * 2 * C, nchunk, C, Cr, HW, fc1, fc2, ca + (size_t)b * C, sm, sm + ((2 * C + 2 * Cr + 3) & ~3), threadIdx.x,
                  blockDim.x);
    asm volatile("" ::"v"(warm));
}

__global__ __launch_bounds__(256) void eca_gate_kernel(const float* partial, int nchunk, int C, int HW, const float* w3, float* gate) {
    extern __shared__ float sm[];
    const int b = blockIdx.x;
    eca_gate_body(partial + (size_t)b * nchunk * 2 * C, nchunk, C, HW, w3, gate + (size_t)b * C, sm, sm + C, threadIdx.x, 256);
}

// LP lanes cooperate on one pixel (LP = power of two >= C/8, at most 64); a thread works on GP_PIX pixels of image blockIdx.y
// at once (pixel slot + u * slots), so that it has GP_PIX 16-byte loads in flight and reads its channels' gate once.
// One load per thread left the reads at 3.9 TB/s; the 64-bit index divisions of the first version cost as much as the loads.
constexpr int GP_PIX = 4;
// (mean, max) over the channels of x * ca for the GP_PIX pixels pbase + u * slots of one image; sub = the thread's lane among the LP
// of its pixels.  src: the image's first element, ca_b / stats_b: the image's gate and its [HW][2] output.
template <typename T>
__device__ inline void cbam_stats_body(const unsigned char* src, int ldc, int C, int HW, int lps, int slots, int pbase, int sub, const float* ca_b,
                                       float* stats_b) {
    const int G = C / 8, LP = 1 << lps;
    float s[GP_PIX], m[GP_PIX];
#pragma unroll
    for (int u = 0; u < GP_PIX; ++u) { s[u] = 0.0f; m[u] = -INFINITY; }
    for (int g = sub; g < G; g += LP) {
        float v[GP_PIX][8];
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) Chan8<T>::load(src + ((size_t)min(pbase + u * slots, HW - 1) * ldc + g * 8) * sizeof(T), v[u]);
        const float4 c0 = *(const float4*)(ca_b + g * 8), c1 = *(const float4*)(ca_b + g * 8 + 4);
        const float cg[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[u][e] * cg[e];
                s[u] += x;
                m[u] = fmaxf(m[u], x);
            }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) {
            s[u] += __shfl_xor(s[u], o, 64);
            m[u] = fmaxf(m[u], __shfl_xor(m[u], o, 64));
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) {
            const int pix = pbase + u * slots;
            if (pix < HW) *(float2*)(stats_b + (size_t)pix * 2) = make_float2(s[u] / (float)C, m[u]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cbam_stats_kernel(const unsigned char* in, int ldc, int coff, int C, int HW, int lps, const float* ca,
                                                         float* stats) {
    const int LP = 1 << lps, slots = 256 >> lps;
    const int t = threadIdx.x, b = blockIdx.y;
    cbam_stats_body<T>(in + (((size_t)b * HW) * ldc + coff) * sizeof(T), ldc, C, HW, lps, slots, blockIdx.x * (slots * GP_PIX) + (t >> lps), t & (LP - 1),
                       ca + (size_t)b * C, stats + (size_t)b * HW * 2);
}

// sa = sigmoid(conv k x k over the (mean, max) planes), out = x * ca * sa for the same GP_PIX pixels
template <typename T>
__device__ inline void cbam_apply_body(const unsigned char* src, unsigned char* dst, int ldc, int out_ldc, int C, int H, int W, int lps, int slots,
                                       int pbase, int sub, const float* ca_b, const float2* sp, const float* wk, int k) {
    const int G = C / 8, LP = 1 << lps, HW = H * W;
    int py[GP_PIX], px[GP_PIX];
    float a[GP_PIX];
#pragma unroll
    for (int u = 0; u < GP_PIX; ++u) {
        const int pix = min(pbase + u * slots, HW - 1);
        py[u] = pix / W;
        px[u] = pix - py[u] * W;
        a[u] = 0.0f;
    }
    // spatial gate: the k x k taps of the (mean, max) planes are spread over the LP lanes of a pixel; a tap's weights and offsets
    // are found once for the thread's GP_PIX pixels; a tap outside the image reads pixel 0 with zero weights (no branch)
    const int pad = k / 2, kk = k * k;
    for (int tap = sub; tap < kk; tap += LP) {
        const int ky = tap / k, kx = tap - ky * k;
        const float wa = wk[tap], wm = wk[kk + tap];
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) {
            const int iy = py[u] + ky - pad, ix = px[u] + kx - pad;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const float2 sv = sp[ok ? iy * W + ix : 0];
            a[u] = fmaf(ok ? wa : 0.0f, sv.x, a[u]);
            a[u] = fmaf(ok ? wm : 0.0f, sv.y, a[u]);
        }
    }
    for (int o = LP >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) a[u] += __shfl_xor(a[u], o, 64);
    }
    float sa[GP_PIX];
#pragma unroll
    for (int u = 0; u < GP_PIX; ++u) sa[u] = sigmoidf_(a[u]);
    for (int g = sub; g < G; g += LP) {
        float v[GP_PIX][8];
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) Chan8<T>::load(src + ((size_t)min(pbase + u * slots, HW - 1) * ldc + g * 8) * sizeof(T), v[u]);
        const float4 c0 = *(const float4*)(ca_b + g * 8), c1 = *(const float4*)(ca_b + g * 8 + 4);
        const float cg[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int u = 0; u < GP_PIX; ++u) {
            const int pix = pbase + u * slots;
            if (pix < HW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = (v[u][e] * cg[e]) * sa[u];
                Chan8<T>::store(dst + ((size_t)pix * out_ldc + g * 8) * sizeof(T), v[u]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_kernel(const unsigned char* in, int ldc, int coff, int C, int H, int W, int lps, const float* ca,
                                                         const float* stats, const float* wk, int k, unsigned char* out, int out_ldc,
                                                         int out_coff) {
    const int LP = 1 << lps, slots = 256 >> lps, HW = H * W;
    const int t = threadIdx.x, b = blockIdx.y;
    cbam_apply_body<T>(in + (((size_t)b * HW) * ldc + coff) * sizeof(T), out + (((size_t)b * HW) * out_ldc + out_coff) * sizeof(T), ldc, out_ldc, C, H, W,
                       lps, slots, blockIdx.x * (slots * GP_PIX) + (t >> lps), t & (LP - 1), ca + (size_t)b * C, (const float2*)stats + (size_t)b * HW, wk, k);
}

// The whole CBAM of one image in ONE workgroup of 1024 threads, for small maps (the 20 x 20 gate sits on every net's critical path as
// four dependent launches of a few microseconds of work each; measured SLOWER than those, see ccnet_cbam: opt-in): the four passes above, back to back, with the partials, the gate and
// the (mean, max) planes in LDS and x re-read from L2.  The same device functions in the same roles -- a group of 256 threads per
// pixel chunk in the pooling, all 1024 in the MLP as cbam_mlp_kernel runs it, LP lanes per pixel in the two pixel passes -- so the
// result is bit-identical to the four launches.  Dynamic LDS (floats): partial[nchunk][2][C] | red[4][2][256][8] | sm | tmp[4096] |
// ca[C] | stats[HW][2].
template <typename T>
__global__ __launch_bounds__(1024) void cbam_image_kernel(const unsigned char* in, int ldc, int coff, int C, int Cr, int H, int W, int nchunk, int lps,
                                                          const float* fc1, const float* fc2, const float* wk, int k, unsigned char* out, int out_ldc,
                                                          int out_coff) {
    extern __shared__ __align__(16) float gsm[];
    const int HW = H * W, t = threadIdx.x, b = blockIdx.x;
    float* partial = gsm;
    float* red = partial + (size_t)nchunk * 2 * C;
    float* sm = red + 4 * 2 * 256 * 8;
    float* tmp = sm + ((2 * C + 2 * Cr + 3) & ~3);
    float* ca_s = tmp + 4 * 1024;
    float* stats_s = ca_s + C;
    const unsigned char* src = in + (((size_t)b * HW) * ldc + coff) * sizeof(T);
    unsigned char* dst = out + (((size_t)b * HW) * out_ldc + out_coff) * sizeof(T);
    float warm = 0.0f;   // the MLP's weights on their way while the image is pooled (cbam_mlp_kernel)
    for (int i = t * 32; i < Cr * C; i += 1024 * 32) warm += fc1[i] + fc2[i];
    const int grp = t >> 8, tl = t & 255;
    for (int c0 = 0; c0 < nchunk; c0 += 4) {
        const int chunk = c0 + grp;
        pool_partial_body<T>(src, ldc, C, HW, min(chunk, nchunk - 1), chunk < nchunk, red + grp * (2 * 256 * 8), partial + (size_t)min(chunk, nchunk - 1) * 2 * C, tl);
    }
    cbam_mlp_body(partial, nchunk, C, Cr, HW, fc1, fc2, ca_s, sm, tmp, t, 1024);
    asm volatile("" ::"v"(warm));
    __syncthreads();
    const int LP = 1 << lps, slots = 1024 >> lps, sub = t & (LP - 1), slot = t >> lps;
    for (int p0 = 0; p0 < HW; p0 += slots * GP_PIX) cbam_stats_body<T>(src, ldc, C, HW, lps, slots, p0 + slot, sub, ca_s, stats_s);
    __syncthreads();
    for (int p0 = 0; p0 < HW; p0 += slots * GP_PIX)
        cbam_apply_body<T>(src, dst, ldc, out_ldc, C, H, W, lps, slots, p0 + slot, sub, ca_s, (const float2*)stats_s, wk, k);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const unsigned char* in, int ldc, int coff, int C, int HW, long long total,
                                                    const float* gate, unsigned char* out, int out_ldc, int out_coff) {
    const int G = C / 8;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const int b = (int)(pix / HW);
    float v[8];
    Chan8<T>::load(in + ((size_t)pix * ldc + coff + g * 8) * sizeof(T), v);
    const float* cg = gate + (size_t)b * C + g * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= cg[e];
    Chan8<T>::store(out + ((size_t)pix * out_ldc + out_coff + g * 8) * sizeof(T), v);
}

struct SrcP {
    const unsigned char* ptr;
    int ldc, coff, shift;
};

template <typename T>
__global__ __launch_bounds__(256) void asff_fuse_kernel(SrcP s0, SrcP s1, SrcP s2, const unsigned char* logit, int l_ldc, int l_coff, int C, int H,
                                                        int W, long long total, unsigned char* out, int out_ldc, int out_coff) {
    const int G = C / 8;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int b = (int)(pix / ((long long)W * H));
    const T* lp = (const T*)(logit + ((size_t)pix * l_ldc + l_coff) * sizeof(T));
    float a0 = load1<T>(lp), a1 = load1<T>(lp + 1), a2 = load1<T>(lp + 2);
    float mx = fmaxf(a0, fmaxf(a1, a2));
    float e0 = __expf(a0 - mx), e1 = __expf(a1 - mx), e2 = __expf(a2 - mx);
    float inv = 1.0f / (e0 + e1 + e2);
    float w[3] = {e0 * inv, e1 * inv, e2 * inv};
    const SrcP* src[3] = {&s0, &s1, &s2};
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const SrcP& s = *src[k];
        size_t sp = ((size_t)b * (H >> s.shift) + (y >> s.shift)) * (W >> s.shift) + (x >> s.shift);
        float v[8];
        Chan8<T>::load(s.ptr + (sp * s.ldc + s.coff + g * 8) * sizeof(T), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e] * w[k];
    }
    Chan8<T>::store(out + ((size_t)pix * out_ldc + out_coff + g * 8) * sizeof(T), acc);
}

bool slice8_ok(const ccnet_slice& s, int dtype) {
    return cc_slice_vec_ok(s, dtype) && (s.ldc % 8) == 0 && (s.coff % 8) == 0 && (s.C % 8) == 0;
}
int lanes_per_pixel(int C) {
    int g = C / 8, lp = 1;
    while (lp < g && lp < 64) lp <<= 1;
    return lp;
}

}  // namespace

extern "C" size_t ccnet_gate_scratch_floats(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return gate_scratch(B, H, W, C).total;
}

extern "C" int ccnet_cbam(ccnet_slice in, int dtype, int B, int H, int W, const float* fc1, const float* fc2, int Cr, const float* w7, int k7,
                          float* scratch_dev, ccnet_slice out, ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && slice8_ok(in, dtype) && slice8_ok(out, dtype) && in.C == out.C && in.shift == 0 && out.shift == 0,
               "cbam: bad slices");
    CC_REQUIRE(fc1 && fc2 && w7 && scratch_dev && ((uintptr_t)scratch_dev & 15) == 0 && Cr > 0 && (k7 == 7 || k7 == 3),
               "cbam: bad weights or scratch not 16-byte aligned (Cr=%d k=%d)", Cr, k7);
    CC_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && (long long)B * H * W < (1ll << 31), "cbam: bad geometry");
    const int C = in.C, HW = H * W;
    const GateScratch gs = gate_scratch(B, H, W, C);
    const int LP = lanes_per_pixel(C);
    int lps = 0;
    while ((1 << lps) < LP) ++lps;
    const int ppb = (256 / LP) * GP_PIX;  // pixels per block
    const dim3 pgrid((unsigned)((HW + ppb - 1) / ppb), (unsigned)B);
    // small maps: the four passes in one workgroup per image (cbam_image_kernel; bit-identical) -- OPT-IN, CCNET_CBAM_IMAGE=1: at
    // 64 x 20 x 20 x 512 it takes 67 us where the four launches take 46-49 (one workgroup per image is 64 CUs walking three passes at
    // one CU's bandwidth; the four launches each use the whole chip).  Kept as the measured answer to "fewer launches" and for
    // callers with many small gates in flight.  Shapes: maps up to 24 x 24 whose MLP runs on 1024 threads as a launch of its own
    // (the summation order depends on the thread count).
    const char* img_e = getenv("CCNET_CBAM_IMAGE");   // read per call (a call records or launches once): tests switch it
    const size_t img_lds = ((size_t)gs.nchunk * 2 * C + 4 * 2 * 256 * 8 + ((2 * C + 2 * Cr + 3) & ~3) + 4 * 1024 + C + 2 * (size_t)HW) * sizeof(float);
    const bool per_image = img_e && img_e[0] == '1' && HW <= 576 && 2 * Cr >= 64 && img_lds <= 150 * 1024;
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            float* partial = scratch_dev + gs.partial;
            float* ca = scratch_dev + gs.ca;
            float* stats = scratch_dev + gs.stats;
            const unsigned char* ip = (const unsigned char*)in.ptr;
            unsigned char* op = (unsigned char*)out.ptr;
            const int mlp_threads = 2 * Cr >= 64 ? 1024 : 256;  // 8 hidden units per wave pass
            const size_t mlp_lds = (size_t)(((2 * C + 2 * Cr + 3) & ~3) + 4 * mlp_threads) * sizeof(float);
            if (per_image) {
                static std::mutex mu;
                static size_t allowed[2][16] = {{0}};   // largest dynamic LDS size set per (dtype, device)
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
                const int ti = dtype == CCNET_BF16 ? 0 : 1;
                {
                    std::lock_guard<std::mutex> lock(mu);
                    if (allowed[ti][dev] < 150 * 1024) {   // once, to the largest size the dispatch rule lets through
                        const void* fn = dtype == CCNET_BF16 ? (const void*)cbam_image_kernel<__bf16> : (const void*)cbam_image_kernel<float>;
                        CC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                        allowed[ti][dev] = 150 * 1024;
                    }
                }
                if (dtype == CCNET_BF16)
                    hipLaunchKernelGGL(cbam_image_kernel<__bf16>, dim3(B), dim3(1024), img_lds, st, ip, in.ldc, in.coff, C, Cr, H, W, gs.nchunk, lps, fc1, fc2,
                                       w7, k7, op, out.ldc, out.coff);
                else
                    hipLaunchKernelGGL(cbam_image_kernel<float>, dim3(B), dim3(1024), img_lds, st, ip, in.ldc, in.coff, C, Cr, H, W, gs.nchunk, lps, fc1, fc2,
                                       w7, k7, op, out.ldc, out.coff);
                CC_LAUNCH_CHECK();
                return CCNET_OK;
            }
            if (dtype == CCNET_BF16) {
                hipLaunchKernelGGL(pool_partial_kernel<__bf16>, dim3(gs.nchunk, B), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, gs.nchunk, partial);
                hipLaunchKernelGGL(cbam_mlp_kernel, dim3(B), dim3(mlp_threads), mlp_lds, st, partial, gs.nchunk, C, Cr, HW, fc1, fc2, ca);
                hipLaunchKernelGGL(cbam_stats_kernel<__bf16>, pgrid, dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, lps, ca, stats);
                hipLaunchKernelGGL(cbam_apply_kernel<__bf16>, pgrid, dim3(256), 0, st, ip, in.ldc, in.coff, C, H, W, lps, ca, stats, w7, k7, op,
                                   out.ldc, out.coff);
            } else {
                hipLaunchKernelGGL(pool_partial_kernel<float>, dim3(gs.nchunk, B), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, gs.nchunk, partial);
                hipLaunchKernelGGL(cbam_mlp_kernel, dim3(B), dim3(mlp_threads), mlp_lds, st, partial, gs.nchunk, C, Cr, HW, fc1, fc2, ca);
                hipLaunchKernelGGL(cbam_stats_kernel<float>, pgrid, dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, lps, ca, stats);
                hipLaunchKernelGGL(cbam_apply_kernel<float>, pgrid, dim3(256), 0, st, ip, in.ldc, in.coff, C, H, W, lps, ca, stats, w7, k7, op,
                                   out.ldc, out.coff);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_CBAM, 0.0, 2.0 * B * H * W * (double)C * cc_esize(dtype));
}

extern "C" int ccnet_eca(ccnet_slice in, int dtype, int B, int H, int W, const float* w3, float* scratch_dev, ccnet_slice out,
                         ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && slice8_ok(in, dtype) && slice8_ok(out, dtype) && in.C == out.C && in.shift == 0 && out.shift == 0,
               "eca: bad slices");
    CC_REQUIRE(w3 && scratch_dev && ((uintptr_t)scratch_dev & 15) == 0, "eca: null pointer or scratch not 16-byte aligned");
    CC_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && (long long)B * H * W < (1ll << 31), "eca: bad geometry");
    const int C = in.C, HW = H * W;
    const GateScratch gs = gate_scratch(B, H, W, C);
    const long long total = (long long)B * HW * (C / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            float* partial = scratch_dev + gs.partial;
            float* gate = scratch_dev + gs.ca;
            const unsigned char* ip = (const unsigned char*)in.ptr;
            unsigned char* op = (unsigned char*)out.ptr;
            if (dtype == CCNET_BF16) {
                hipLaunchKernelGGL(pool_partial_kernel<__bf16>, dim3(gs.nchunk, B), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, gs.nchunk, partial);
                hipLaunchKernelGGL(eca_gate_kernel, dim3(B), dim3(256), (C + 1024) * sizeof(float), st, partial, gs.nchunk, C, HW, w3, gate);
                hipLaunchKernelGGL(scale_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, total, gate, op, out.ldc, out.coff);
            } else {
                hipLaunchKernelGGL(pool_partial_kernel<float>, dim3(gs.nchunk, B), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, gs.nchunk, partial);
                hipLaunchKernelGGL(eca_gate_kernel, dim3(B), dim3(256), (C + 1024) * sizeof(float), st, partial, gs.nchunk, C, HW, w3, gate);
                hipLaunchKernelGGL(scale_kernel<float>, dim3(blocks), dim3(256), 0, st, ip, in.ldc, in.coff, C, HW, total, gate, op, out.ldc, out.coff);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_ECA, 0.0, 2.0 * B * H * W * (double)C * cc_esize(dtype));
}

extern "C" int ccnet_asff_fuse(ccnet_slice l0, ccnet_slice l1, ccnet_slice l2, ccnet_slice logits, int dtype, int B, int H, int W, ccnet_slice out,
                               ccnet_stream stream) {
    CC_REQUIRE(cc_dtype_ok(dtype) && slice8_ok(l0, dtype) && slice8_ok(l1, dtype) && slice8_ok(l2, dtype) && slice8_ok(out, dtype) && out.shift == 0,
               "asff_fuse: bad slices");
    CC_REQUIRE(l0.C == out.C && l1.C == out.C && l2.C == out.C, "asff_fuse: channel mismatch");
    CC_REQUIRE(logits.ptr && logits.C == 3 && logits.shift == 0 && logits.ldc >= logits.coff + 3, "asff_fuse: logits slice must have 3 channels");
    CC_REQUIRE(B > 0 && H > 0 && W > 0, "asff_fuse: bad geometry");
    const ccnet_slice* ls[3] = {&l0, &l1, &l2};
    for (int k = 0; k < 3; ++k)
        CC_REQUIRE((H % (1 << ls[k]->shift)) == 0 && (W % (1 << ls[k]->shift)) == 0, "asff_fuse: H,W not divisible by upsampling factor");
    const long long total = (long long)B * H * W * (out.C / 8);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    SrcP s0{(const unsigned char*)l0.ptr, l0.ldc, l0.coff, l0.shift}, s1{(const unsigned char*)l1.ptr, l1.ldc, l1.coff, l1.shift},
        s2{(const unsigned char*)l2.ptr, l2.ldc, l2.coff, l2.shift};
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (dtype == CCNET_BF16)
                hipLaunchKernelGGL(asff_fuse_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, s0, s1, s2, (const unsigned char*)logits.ptr, logits.ldc,
                                   logits.coff, out.C, H, W, total, (unsigned char*)out.ptr, out.ldc, out.coff);
            else
                hipLaunchKernelGGL(asff_fuse_kernel<float>, dim3(blocks), dim3(256), 0, st, s0, s1, s2, (const unsigned char*)logits.ptr, logits.ldc,
                                   logits.coff, out.C, H, W, total, (unsigned char*)out.ptr, out.ldc, out.coff);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_ASFF, 0.0,
        ((double)B * H * W * out.C + (double)B * (H >> l0.shift) * (W >> l0.shift) * out.C + (double)B * (H >> l1.shift) * (W >> l1.shift) * out.C +
         (double)B * (H >> l2.shift) * (W >> l2.shift) * out.C) * cc_esize(dtype));
}
