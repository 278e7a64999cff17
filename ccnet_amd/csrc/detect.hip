// decode_outputs + post_process on the GPU (compile this file with -ffp-contract=off: the box arithmetic must
// round exactly like the reference's separate fp32 tensor ops, so no FMA contraction anywhere in here).
//
//   decode   : sigmoid / exp / grid, one thread per anchor                       (utils/bbox.py:36-76)
//   filter   : xyxy, class max, conf mask, stable compaction, one block per image (utils/bbox.py:98-130)
//   nms      : coordinate-trick batched NMS                                       (utils/bbox.py:133-138 -> torchvision)
//              prepare (max coordinate, offset boxes, scores) -> rank sort (stable, descending)
//              -> 64x64 IoU bitmask tiles (one wave per tile, one 64-bit word per lane)
//              -> sequential sweep, 64 boxes at a time resolved in registers -> gather + rescale (utils/bbox.py:145-158)
#include <math.h>
#include <mutex>
#include "common.h"

namespace {

constexpr int MAXLEV = 8;
struct DecodeP {
    const float* head[MAXLEV];
    long long sb[MAXLEV], sc[MAXLEV], sp[MAXLEV];
    int h[MAXLEV], w[MAXLEV], a0[MAXLEV + 1];
    float stride[MAXLEV], rw[MAXLEV];
    float rnch;
    int fast;
    int nlev, B, A, nch;
    float in_h, in_w;
    float* out;
};

// one thread per output ELEMENT (anchor, channel) of image blockIdx.y: with NHWC heads both the load and the store of a wave
// are contiguous.  Index arithmetic is 32-bit (64-bit divisions made this kernel VALU-bound: 31 us for 47 MB).
__global__ __launch_bounds__(256) void decode_kernel(const DecodeP p) {
    // blockIdx.z = level, so every per-level quantity is a scalar; the two index divisions go through a float reciprocal
    // when every level has fewer than 2^22 elements (two roundings of 2^-24 each keep the product inside its integer cell),
    // not the ~30-instruction integer sequence
    const int l = blockIdx.z, b = blockIdx.y;
    const int wl = p.w[l];
    const unsigned e = blockIdx.x * 256u + threadIdx.x;      // element inside this level: r * nch + c
    const unsigned n_l = (unsigned)(p.a0[l + 1] - p.a0[l]) * (unsigned)p.nch;
    if (e >= n_l) return;
    const int r = p.fast ? (int)(((float)e + 0.5f) * p.rnch) : (int)(e / (unsigned)p.nch);
    const int c = (int)e - r * p.nch;
    const int gy = p.fast ? (int)(((float)r + 0.5f) * p.rw[l]) : r / wl;
    const int gx = r - gy * wl;
    const unsigned per_img = (unsigned)p.A * (unsigned)p.nch;
    const float raw = p.head[l][(long long)b * p.sb[l] + (long long)r * p.sp[l] + (long long)c * p.sc[l]];
    const float s = p.stride[l];
    // ((raw + grid) * stride) / dim, (exp(raw) * stride) / dim, 1 / (1 + exp(-raw)): same operations on the same operands as
    // bbox.py:39,70-76, arranged as ONE exponential and ONE division per element with selects around them -- a wave holds all
    // channel kinds, and five divergent branches (each with its own IEEE division) made this kernel VALU-bound
    const bool is_xy = c < 2, is_box = c < 4;
    const float ex = expf(is_box ? raw : -raw);
    const float g = (c == 0) ? (float)gx : (float)gy;
    const float num = is_xy ? (raw + g) * s : (is_box ? ex * s : 1.0f);
    const float den = is_box ? ((c & 1) ? p.in_h : p.in_w) : 1.0f + ex;
    const float v = num / den;
    p.out[(size_t)b * per_img + (unsigned)p.a0[l] * (unsigned)p.nch + e] = v;
}

// ---- filter: one block (1024 threads) per image, anchors in order, stable compaction.
// Per super-chunk of 32 x 1024 anchors: pass 1 evaluates the keep test of every anchor (independent loads, no barrier between
// them, so their latencies overlap) and leaves one flag bit per anchor in a register and one count per (iteration, wave) in
// LDS; one scan turns the counts into output offsets; pass 2 re-reads only the kept rows (L2-hot) and writes them.  The
// previous form paid a global-load round trip and three barriers per 1024 anchors (23 us for 8400 anchors x 64 images).
constexpr int F_IT = 32;
__global__ __launch_bounds__(1024) void filter_kernel(const float* pred, int A, int nc, float conf, float* cand, int* counts) {
    __shared__ int cnt[F_IT * 16];
    __shared__ int base_s;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nch = 5 + nc;
    const float* pb = pred + (size_t)b * A * nch;
    float* cb = cand + (size_t)b * A * 7;
    if (t == 0) base_s = 0;
    for (int s0 = 0; s0 < A; s0 += F_IT * 1024) {
        const int iters = min(F_IT, (A - s0 + 1023) / 1024);
        unsigned flags = 0u;
        for (int k = 0; k < iters; ++k) {
            const int a = s0 + k * 1024 + t;
            bool keep = false;
            if (a < A) {
                const float* r = pb + (size_t)a * nch;
                const float obj = r[4];
                float cc = r[5];
                for (int c = 1; c < nc; ++c) {  // the value of the maximum, by the same comparisons as pass 2 (its index is found there)
                    const float v = r[5 + c];
                    if (v > cc) cc = v;
                }
                keep = (obj * cc) >= conf;
            }
            flags |= keep ? (1u << k) : 0u;
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) cnt[k * 16 + wave] = __popcll(bal);
        }
        __syncthreads();
        // exclusive scan of cnt in (iteration, wave) order = anchor order; 512 entries at most, one wave does it
        if (t < 64) {
            int carry = base_s;
            for (int i0 = 0; i0 < iters * 16; i0 += 64) {
                const int i = i0 + t;
                const int v = i < iters * 16 ? cnt[i] : 0;
                int incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int u = __shfl_up(incl, o, 64);
                    if (t >= o) incl += u;
                }
                if (i < iters * 16) cnt[i] = carry + incl - v;
                carry += __shfl(incl, 63, 64);
            }
            if (t == 0) base_s = carry;
        }
        __syncthreads();
        for (int k = 0; k < iters; ++k) {
            const bool keep = (flags >> k) & 1u;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int a = s0 + k * 1024 + t;
                const float* r = pb + (size_t)a * nch;
                const float cx = r[0], cy = r[1], w = r[2], h = r[3];
                float cc = r[5];
                int cp = 0;
                for (int c = 1; c < nc; ++c) {
                    const float v = r[5 + c];
                    if (v > cc) { cc = v; cp = c; }  // strict >: first maximal index on ties (torch.max)
                }
                float* d = cb + (size_t)(cnt[k * 16 + wave] + __popcll(bal & ((1ull << lane) - 1ull))) * 7;
                d[0] = cx - w / 2; d[1] = cy - h / 2; d[2] = cx + w / 2; d[3] = cy + h / 2; d[4] = r[4]; d[5] = cc; d[6] = (float)cp;
            }
        }
        __syncthreads();  // cnt is rewritten by the next super-chunk
    }
    if (t == 0) counts[b] = base_s;
}

// ---- NMS workspace (per image, nmax boxes; W = ceil(nmax/64) words)
struct NmsWs {
    size_t boxes, scores, order, mask, total;  // byte offsets
    int W;
};
NmsWs nms_ws(int B, int nmax) {
    NmsWs w;
    w.W = (nmax + 63) / 64;
    size_t np = (size_t)w.W * 64;
    w.boxes = 0;                                         // float4 [B][np]  offset boxes, later sorted boxes
    w.scores = w.boxes + (size_t)B * np * 16 * 2;        // two float4 planes: unsorted, sorted
    w.order = w.scores + (size_t)B * np * 4;             // float [B][np]
    w.mask = w.order + (size_t)B * np * 4;               // int [B][np]
    w.mask = (w.mask + 255) & ~(size_t)255;
    w.total = w.mask + (size_t)B * np * w.W * 8;         // u64 [B][np][W]
    return w;
}

// word held by lane j (j wave-uniform) as a scalar: two v_readlane_b32 instead of two ds_bpermute round trips -- the greedy
// sweep reads 64 of them in a dependent loop, and the LDS latency of the shuffles was most of its time
__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int j) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
    return ((unsigned long long)hi << 32) | lo;
}

// torchvision nms: areas and the intersection in fp32, ovr = inter / (area_i + area_j - inter) compared as a double against
// the threshold; a NaN overlap (0/0 of two empty boxes) compares false and never suppresses.  Shared by both NMS paths.
__device__ __forceinline__ bool suppresses(const float4& bi, float ai, const float4& bj, double thr) {
    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
    const float w = fmaxf(0.0f, fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x));
    const float h = fmaxf(0.0f, fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y));
    const float inter = w * h;
    const float ovr = inter / ((ai + aj) - inter);
    return (double)ovr > thr;
}

// det row = candidate row rescaled like the numpy tail of post_process (bbox.py:145-158): fp32 centre / size arithmetic, then a
// float64 multiply by the image size.  Shared by both NMS paths.
__device__ __forceinline__ void write_det(const float* r, const int* image_hw, int b, float* d) {
    const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
    const float w = x2 - x1, h = y2 - y1;
    const float top = cy - (h / 2.f), left = cx - (w / 2.f), bottom = cy + (h / 2.f), right = cx + (w / 2.f);
    const double ih = (double)image_hw[b * 2], iw = (double)image_hw[b * 2 + 1];
    d[0] = (float)((double)top * ih);
    d[1] = (float)((double)left * iw);
    d[2] = (float)((double)bottom * ih);
    d[3] = (float)((double)right * iw);
    d[4] = r[4]; d[5] = r[5]; d[6] = r[6];
}

// offset boxes = boxes + cls * (max_coordinate + 1)  (torchvision batched_nms coordinate trick); scores = obj * cls_conf
__global__ __launch_bounds__(256) void nms_prepare_kernel(const float* cand, const int* counts, int A, int np, int nmax, float4* boxes, float* scores) {
    __shared__ float red[4];
    __shared__ float maxc_s;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = min(counts[b], nmax);  // an image with more candidates than the caller sized for is cut, never overrun
    const float* cb = cand + (size_t)b * A * 7;
    float m = -INFINITY;
    for (int i = t; i < n; i += 256) {
        const float* r = cb + (size_t)i * 7;
        m = fmaxf(m, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
    }
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    if (t == 0) maxc_s = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float step = maxc_s + 1.0f;
    for (int i = t; i < n; i += 256) {
        const float* r = cb + (size_t)i * 7;
        const float off = r[6] * step;
        boxes[(size_t)b * np + i] = make_float4(r[0] + off, r[1] + off, r[2] + off, r[3] + off);
        scores[(size_t)b * np + i] = r[4] * r[5];
    }
}

// stable descending rank: rank(i) = #{j : s_j > s_i or (s_j == s_i and j < i)}
__global__ __launch_bounds__(256) void nms_rank_kernel(const int* counts, int np, int nmax, const float4* boxes, const float* scores, float4* sboxes,
                                                       int* order) {
    __shared__ float tile[256];
    const int b = blockIdx.y, t = threadIdx.x;
    const int n = min(counts[b], nmax);
    if ((int)(blockIdx.x * 256) >= n) return;
    const int i = blockIdx.x * 256 + t;
    const float* sb = scores + (size_t)b * np;
    const float si = i < n ? sb[i] : 0.0f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        tile[t] = (j0 + t < n) ? sb[j0 + t] : -INFINITY;
        __syncthreads();
        const int jn = min(256, n - j0);
        for (int j = 0; j < jn; ++j) {
            const float sj = tile[j];
            rank += (sj > si || (sj == si && (j0 + j) < i)) ? 1 : 0;
        }
    }
    if (i < n) {
        order[(size_t)b * np + rank] = i;
        sboxes[(size_t)b * np + rank] = boxes[(size_t)b * np + i];
    }
}

// one wave per 64x64 tile of the (sorted) pair matrix; lane = row, bit j = column
__global__ __launch_bounds__(64) void nms_mask_kernel(const int* counts, int np, int W, int nmax, const float4* sboxes, double thr, unsigned long long* mask) {
    __shared__ float4 cols[64];
    const int b = blockIdx.z, rb = blockIdx.y, cbk = blockIdx.x;
    const int n = min(counts[b], nmax);
    if (cbk < rb || rb * 64 >= n || cbk * 64 >= n) return;
    const int lane = threadIdx.x;
    const float4* sb = sboxes + (size_t)b * np;
    const int ci = cbk * 64 + lane;
    cols[lane] = ci < n ? sb[ci] : make_float4(0, 0, 0, 0);
    __syncthreads();
    const int i = rb * 64 + lane;
    if (i >= n) return;
    const float4 bi = sb[i];
    const float ai = (bi.z - bi.x) * (bi.w - bi.y);
    unsigned long long bits = 0;
    const int jn = min(64, n - cbk * 64);
    for (int j = 0; j < jn; ++j) {
        const int gj = cbk * 64 + j;
        if (gj <= i) continue;
        if (suppresses(bi, ai, cols[j], thr)) bits |= 1ull << j;
    }
    mask[((size_t)b * np + i) * W + cbk] = bits;
}

// one block per image: sweep the sorted boxes 64 at a time
__global__ __launch_bounds__(256) void nms_sweep_kernel(const int* counts, int np, int W, const unsigned long long* mask, const int* order, int nmax,
                                                        int* keep_idx, int* nkeep) {
    extern __shared__ unsigned long long removed[];  // [W]
    __shared__ int kept_l[64];  // positions (0..63) of the boxes kept in the current chunk
    __shared__ int kept_n;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = min(counts[b], nmax);
    const int nw = (n + 63) / 64;
    for (int w = t; w < nw; w += 256) removed[w] = 0ull;
    __syncthreads();
    const unsigned long long* mb = mask + (size_t)b * np * W;
    const int* ob = order + (size_t)b * np;
    int total = 0;  // meaningful in wave 0
    for (int c = 0; c < nw; ++c) {
        if (t < 64) {
            const int i = c * 64 + t;
            const unsigned long long diag = i < n ? mb[(size_t)i * W + c] : 0ull;
            unsigned long long rem = removed[c];
            unsigned long long kept = 0ull;
            const int jn = min(64, n - c * 64);
            for (int j = 0; j < jn; ++j) {
                const unsigned long long dj = lane_word(diag, j);
                if (!((rem >> j) & 1ull)) {
                    kept |= 1ull << j;
                    rem |= dj;
                }
            }
            if ((kept >> t) & 1ull) {
                const int pos = total + __popcll(kept & ((1ull << t) - 1ull));
                keep_idx[(size_t)b * nmax + pos] = ob[i];
            }
            total += __popcll(kept);
            if ((kept >> t) & 1ull) kept_l[__popcll(kept & ((1ull << t) - 1ull))] = t;
            if (t == 0) kept_n = __popcll(kept);
        }
        __syncthreads();
        // OR the mask rows of the kept boxes into the words still ahead.  The loads are what costs (L2 round trips), so
        // they are spread over the whole block -- wave q takes every fourth kept row -- and issued 8 at a time; rows are
        // picked with a clamped index (a repeated OR is harmless), so no load sits behind a branch.
        const int nk = kept_n;
        if (nk > 0) {
            const int wave = t >> 6, lane = t & 63;
            const int mine = (nk - wave + 3) / 4;  // kept rows wave, wave+4, ...
            for (int w = c + 1 + lane; w < nw; w += 64) {
                unsigned long long acc = 0ull;
                for (int q = 0; q < mine; q += 8) {
                    unsigned long long v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = wave + 4 * min(q + u, mine - 1);
                        v[u] = mb[(size_t)(c * 64 + kept_l[idx]) * W + w];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc |= v[u];
                }
                if (mine > 0 && acc) atomicOr(&removed[w], acc);
            }
        }
        __syncthreads();
    }
    if (t == 0) nkeep[b] = total;
}

// det row = cand[keep] rescaled like the numpy tail of post_process (bbox.py:145-158)
__global__ __launch_bounds__(256) void nms_gather_kernel(const float* cand, int A, int nmax, const int* keep_idx, const int* nkeep, const int* image_hw,
                                                         float* det) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nkeep[b]) return;
    write_det(cand + ((size_t)b * A + keep_idx[(size_t)b * nmax + k]) * 7, image_hw, b, det + ((size_t)b * nmax + k) * 7);
}

// ---- the whole NMS of one image in ONE block, for nmax <= SMALL_NMAX candidates per image: prepare, rank, mask, sweep and
// gather as above, with the offset boxes, the sorted boxes, the order and the pair bitmask in LDS instead of the
// workspace.  Five dependent launches of microsecond kernels become one; the arithmetic is the same device functions.
constexpr int SMALL_NMAX = 256, SMALL_W = SMALL_NMAX / 64, SMALL_T = 1024;  // beyond this the tiled kernels' many blocks win

#ifdef NMS_STAMP  // diagnostic build: cycles per phase of block 0 (0 prepare, 1 rank, 2 mask, 3 sweep, 4 gather)
__device__ unsigned long long g_nms_stamps[8];
#define NSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_nms_stamps[i], n_ - st_last); st_last = n_; } } while (0)
#else
#define NSTAMP(i)
#endif
__global__ __launch_bounds__(SMALL_T) void nms_small_kernel(const float* cand, const int* counts, int A, int nmax, double thr, const int* image_hw, float* det,
                                                        int* keep_idx, int* nkeep) {
    __shared__ float4 boxes[SMALL_NMAX];
    __shared__ float4 sboxes[SMALL_NMAX];
    __shared__ float scores[SMALL_NMAX];
    __shared__ int order[SMALL_NMAX];
    __shared__ int kept_s[SMALL_NMAX];  // candidate indices of the kept boxes, score order
    __shared__ unsigned long long mask[SMALL_NMAX * SMALL_W];
    __shared__ float red[SMALL_T / 64];
    __shared__ float maxc_s;
    __shared__ int total_s;
    __shared__ unsigned long long removed_s[SMALL_W];
#ifdef NMS_STAMP
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = min(min(counts[b], nmax), SMALL_NMAX);
    const int nw = (n + 63) / 64;
    const float* cb = cand + (size_t)b * A * 7;
    // prepare
    float m = -INFINITY;
    for (int i = t; i < n; i += SMALL_T) {
        const float* r = cb + (size_t)i * 7;
        m = fmaxf(m, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
    }
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    if (t == 0) {
        float mm = red[0];
        for (int q = 1; q < SMALL_T / 64; ++q) mm = fmaxf(mm, red[q]);
        maxc_s = mm;
    }
    __syncthreads();
    const float step = maxc_s + 1.0f;
    for (int i = t; i < n; i += SMALL_T) {
        const float* r = cb + (size_t)i * 7;
        const float off = r[6] * step;
        boxes[i] = make_float4(r[0] + off, r[1] + off, r[2] + off, r[3] + off);
        scores[i] = r[4] * r[5];
    }
    __syncthreads();
    NSTAMP(0);
    // stable descending rank: one wave per box, lanes over the other boxes (a thread per box would walk n LDS reads in a row)
    const int lane = t & 63, wave = t >> 6;
    // (four boxes / four (row, word) items per wave at a time: each item is a long dependent chain -- LDS read, IEEE division,
    // ballot -- and independent chains side by side are what hides its latency)
    constexpr int NWV = SMALL_T / 64, UNR = 4;
    for (int i0 = wave * UNR; i0 < n; i0 += NWV * UNR) {
        float si[UNR];
        int rank[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            si[u] = scores[min(i0 + u, n - 1)];
            rank[u] = 0;
        }
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const float sj = j < n ? scores[j] : -INFINITY;
#pragma unroll
            for (int u = 0; u < UNR; ++u) rank[u] += __popcll(__ballot(j < n && (sj > si[u] || (sj == si[u] && j < i0 + u))));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (lane == 0 && i0 + u < n) {
                order[rank[u]] = i0 + u;
                sboxes[rank[u]] = boxes[i0 + u];
            }
    }
    __syncthreads();
    NSTAMP(1);
    // pair bitmask: one wave per (row i, word w >= the row's own), one pair per lane -- the IoU's IEEE division is ~40 VALU
    // instructions, and a thread per row would run 64 of them back to back
    const int items = n * nw;
    for (int it0 = wave * UNR; it0 < items; it0 += NWV * UNR) {
        unsigned long long bits[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int it = min(it0 + u, items - 1);
            const int i = it / nw, w = it - i * nw;
            const float4 bi = sboxes[i];
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            const int j = w * 64 + lane;
            const bool hit = j < n && j > i && suppresses(bi, ai, sboxes[min(j, n - 1)], thr);
            bits[u] = __ballot(hit);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (lane == 0 && it0 + u < items) mask[((it0 + u) / nw) * SMALL_W + (it0 + u) % nw] = bits[u];
    }
    __syncthreads();
    NSTAMP(2);
    // greedy sweep by wave 0, 64 sorted boxes at a time.  Everything the inner loop touches is wave-uniform and is kept in
    // scalar registers on purpose (readfirstlane / readlane): as vector code the 64 dependent iterations of shifts, tests and
    // ORs cost four cycles per instruction and took 27 of this kernel's 33 us.  Rows of kept boxes reach the words ahead
    // through LDS atomics (one wave: LDS executes its instructions in order, so the next chunk's read sees them).
    if (t < 64) {
        volatile unsigned long long* rm = removed_s;
        if (t < SMALL_W) rm[t] = 0ull;
        int total = 0;
        for (int c = 0; c < nw; ++c) {
            const int i = c * 64 + t;
            const unsigned long long diag = i < n ? mask[i * SMALL_W + c] : 0ull;
            const unsigned long long r0 = rm[c];
            unsigned long long rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(r0 >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)r0);
            unsigned long long kept = 0ull;
            const int jn = min(64, n - c * 64);
            for (int j = 0; j < jn; ++j) {
                const unsigned long long dj = lane_word(diag, j);
                if (!((rem >> j) & 1ull)) {
                    kept |= 1ull << j;
                    rem |= dj;
                }
            }
            const bool mine = (kept >> t) & 1ull;
            if (mine) kept_s[total + __popcll(kept & ((1ull << t) - 1ull))] = order[i];
            total += __popcll(kept);
            if (mine)
                for (int w = c + 1; w < nw; ++w) {
                    const unsigned long long v = mask[i * SMALL_W + w];
                    if (v) atomicOr(&removed_s[w], v);
                }
        }
        if (t == 0) {
            nkeep[b] = total;
            total_s = total;
        }
    }
    __syncthreads();
    NSTAMP(3);
    for (int k = t; k < total_s; k += SMALL_T) {
        const int src = kept_s[k];
        keep_idx[(size_t)b * nmax + k] = src;
        write_det(cb + (size_t)src * 7, image_hw, b, det + ((size_t)b * nmax + k) * 7);
    }
    NSTAMP(4);
}

}  // namespace

#ifdef NMS_STAMP
extern "C" int ccnet_debug_nms_stamps(unsigned long long* out8) {
    CC_HIP(hipDeviceSynchronize());
    CC_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_nms_stamps), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0};
    CC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_nms_stamps), z, sizeof(z)));
    return CCNET_OK;
}
#endif

extern "C" int ccnet_decode(const float* const* head_dev, const int64_t* sb, const int64_t* sc, const int64_t* sp, const int32_t* h, const int32_t* w,
                            int nlev, int B, int nc, float in_h, float in_w, float* out_dev, ccnet_stream stream) {
    CC_REQUIRE(head_dev && sb && sc && sp && h && w && out_dev, "decode: null pointer");
    CC_REQUIRE(nlev > 0 && nlev <= MAXLEV && B > 0 && nc > 0 && in_h > 0 && in_w > 0, "decode: bad sizes (nlev=%d)", nlev);
    DecodeP p{};
    int A = 0;
    for (int l = 0; l < nlev; ++l) {
        CC_REQUIRE(head_dev[l] && h[l] > 0 && w[l] > 0, "decode: bad level %d", l);
        p.head[l] = head_dev[l]; p.sb[l] = sb[l]; p.sc[l] = sc[l]; p.sp[l] = sp[l]; p.h[l] = h[l]; p.w[l] = w[l];
        p.a0[l] = A;
        // python: input_shape[0] / h is a double, stored into an fp32 tensor by torch.full (bbox.py:57)
        p.stride[l] = (float)((double)in_h / (double)h[l]);
        A += h[l] * w[l];
    }
    p.a0[nlev] = A;
    p.nlev = nlev; p.B = B; p.A = A; p.nch = 5 + nc; p.in_h = in_h; p.in_w = in_w; p.out = out_dev;
    CC_REQUIRE((double)A * (5 + nc) < 2.0e9 && B <= 65535, "decode: too many elements per image or too many images");
    long long biggest = 0;
    for (int l = 0; l < nlev; ++l) {
        p.rw[l] = 1.0f / (float)w[l];
        const long long n_l = (long long)h[l] * w[l] * (5 + nc);
        biggest = n_l > biggest ? n_l : biggest;
    }
    p.rnch = 1.0f / (float)(5 + nc);
    p.fast = biggest < (1ll << 22) ? 1 : 0;
    const unsigned blocks = (unsigned)((biggest + 255) / 256);
    return ccnet_submit(
        [p, blocks, B, nlev](hipStream_t st, const float*) -> int {
            hipLaunchKernelGGL(decode_kernel, dim3(blocks, B, nlev), dim3(256), 0, st, p);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_DECODE, 0.0, 2.0 * B * (double)A * (5 + nc) * 4.0);
}

extern "C" int ccnet_filter(const float* pred_dev, int B, int A, int nc, float conf_thres, float* cand_dev, int32_t* counts_dev,
                            ccnet_stream stream) {
    CC_REQUIRE(pred_dev && cand_dev && counts_dev, "filter: null pointer");
    CC_REQUIRE(B > 0 && B <= 65535 && A > 0 && nc > 0, "filter: bad sizes");
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            hipLaunchKernelGGL(filter_kernel, dim3(B), dim3(1024), 0, st, pred_dev, A, nc, conf_thres, cand_dev, counts_dev);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_FILTER, 0.0, (double)B * A * (5 + nc) * 4.0);
}

extern "C" size_t ccnet_nms_workspace_bytes(int B, int nmax) {
    if (B <= 0 || nmax <= 0) return 256;
    return nms_ws(B, nmax).total;
}

// Shared by ccnet_nms (caller's workspace) and ccnet_nms_noalloc (the library's static scratch): arguments are validated,
// then the kernels are submitted with `base` as their workspace.
static int nms_submit(const char* who, const float* cand_dev, const int32_t* counts_dev, int B, int A, int nmax, double nms_thres,
                      const int32_t* image_hw_dev, float* det_dev, int32_t* keep_idx_dev, int32_t* nkeep_dev, unsigned char* base, ccnet_stream stream) {
    CC_REQUIRE(cand_dev && counts_dev && image_hw_dev && det_dev && keep_idx_dev && nkeep_dev && base, "%s: null pointer", who);
    CC_REQUIRE(B > 0 && B <= 65535 && A > 0 && nmax > 0 && nmax <= A, "%s: bad sizes (B=%d A=%d nmax=%d)", who, B, A, nmax);
    CC_REQUIRE(((uintptr_t)base % 256) == 0, "%s: workspace must be 256-byte aligned", who);
    const NmsWs ws = nms_ws(B, nmax);
    CC_REQUIRE(ws.W <= 65535, "%s: nmax too large for one launch (%d)", who, nmax);
    CC_REQUIRE((size_t)ws.W * 8 <= 64 * 1024, "%s: nmax too large (%d)", who, nmax);
    const int np = ws.W * 64;
    float4* boxes = (float4*)(base + ws.boxes);
    float4* sboxes = boxes + (size_t)B * np;
    float* scores = (float*)(base + ws.scores);
    int* order = (int*)(base + ws.order);
    unsigned long long* mask = (unsigned long long*)(base + ws.mask);
    const int W = ws.W;
    if (nmax <= SMALL_NMAX)
        return ccnet_submit(
            [=](hipStream_t st, const float*) -> int {
                hipLaunchKernelGGL(nms_small_kernel, dim3(B), dim3(SMALL_T), 0, st, cand_dev, counts_dev, A, nmax, nms_thres, image_hw_dev, det_dev,
                                   keep_idx_dev, nkeep_dev);
                CC_LAUNCH_CHECK();
                return CCNET_OK;
            },
            stream, CCNET_OP_NMS, 0.0, 0.0);
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            hipLaunchKernelGGL(nms_prepare_kernel, dim3(B), dim3(256), 0, st, cand_dev, counts_dev, A, np, nmax, boxes, scores);
            hipLaunchKernelGGL(nms_rank_kernel, dim3((nmax + 255) / 256, B), dim3(256), 0, st, counts_dev, np, nmax, boxes, scores, sboxes, order);
            hipLaunchKernelGGL(nms_mask_kernel, dim3(W, W, B), dim3(64), 0, st, counts_dev, np, W, nmax, sboxes, nms_thres, mask);
            hipLaunchKernelGGL(nms_sweep_kernel, dim3(B), dim3(256), (size_t)W * 8, st, counts_dev, np, W, mask, order, nmax, keep_idx_dev, nkeep_dev);
            hipLaunchKernelGGL(nms_gather_kernel, dim3((nmax + 255) / 256, B), dim3(256), 0, st, cand_dev, A, nmax, keep_idx_dev, nkeep_dev,
                               image_hw_dev, det_dev);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_NMS, 0.0, 0.0);
}

extern "C" int ccnet_nms(const float* cand_dev, const int32_t* counts_dev, int B, int A, int nmax, double nms_thres, const int32_t* image_hw_dev,
                         float* det_dev, int32_t* keep_idx_dev, int32_t* nkeep_dev, void* workspace_dev, ccnet_stream stream) {
    return nms_submit("nms", cand_dev, counts_dev, B, A, nmax, nms_thres, image_hw_dev, det_dev, keep_idx_dev, nkeep_dev, (unsigned char*)workspace_dev,
                      stream);
}

// ---- the entry point that allocates nothing (include/ccnet_hip.h: ccnet_nms_noalloc) ---------------------------------------
// The NMS workspace (offset boxes, sorted boxes, scores, order, pair bitmask) lives in a file-scope device array instead of
// caller memory.  Each GPU the library runs on holds its own copy at its own address for the life of the process; the array's
// NAME is only a host-side stand-in, so the address is resolved once per device with hipGetSymbolAddress and only that pointer
// is ever handed to a kernel.  Nothing in the array has to start at zero: every word a kernel reads was written earlier in the
// same call (mask tiles, order and box slots are all produced for the rows < counts[b] that the later kernels read).
namespace {
constexpr size_t NMS_SCRATCH_BYTES = (size_t)32 << 20;
constexpr int NMS_SCRATCH_MAX_DEV = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[NMS_SCRATCH_BYTES];
namespace {
void* g_scratch_ptr[NMS_SCRATCH_MAX_DEV] = {nullptr};
std::mutex g_scratch_mu;
// address of THIS device's copy of `scratch` (resolved on first use on the device, with that device current)
int scratch_for_current_device(unsigned char** out) {
    int dev = -1;
    CC_HIP(hipGetDevice(&dev));
    CC_REQUIRE(dev >= 0 && dev < NMS_SCRATCH_MAX_DEV, "nms_noalloc: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        CC_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (ptr == nullptr) return ccnet_set_error(CCNET_EHIP, "nms_noalloc: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = ptr;
    }
    *out = (unsigned char*)g_scratch_ptr[dev];
    return CCNET_OK;
}
}  // namespace

extern "C" size_t ccnet_nms_noalloc_max_bytes(void) { return NMS_SCRATCH_BYTES; }

extern "C" int ccnet_nms_noalloc_init(void) {
    unsigned char* ptr = nullptr;
    return scratch_for_current_device(&ptr);
}

extern "C" int ccnet_nms_noalloc(const float* cand_dev, const int32_t* counts_dev, int B, int A, int nmax, double nms_thres,
                                 const int32_t* image_hw_dev, float* det_dev, int32_t* keep_idx_dev, int32_t* nkeep_dev, ccnet_stream stream) {
    CC_REQUIRE(B > 0 && B <= 65535 && A > 0 && nmax > 0 && nmax <= A, "nms_noalloc: bad sizes (B=%d A=%d nmax=%d)", B, A, nmax);
    const size_t need = nms_ws(B, nmax).total;
    CC_REQUIRE(need <= NMS_SCRATCH_BYTES,
               "nms_noalloc: B=%d x nmax=%d needs %zu bytes of workspace, the static scratch holds %zu; use ccnet_nms with a caller workspace", B,
               nmax, need, (size_t)NMS_SCRATCH_BYTES);
    unsigned char* ptr = nullptr;
    const int rc = scratch_for_current_device(&ptr);
    if (rc != CCNET_OK) return rc;
    return nms_submit("nms_noalloc", cand_dev, counts_dev, B, A, nmax, nms_thres, image_hw_dev, det_dev, keep_idx_dev, nkeep_dev, ptr, stream);
}
