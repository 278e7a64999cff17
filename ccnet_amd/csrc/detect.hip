// decode_outputs + post_process on the GPU (compile this file with -ffp-contract=off: the box arithmetic must
// round exactly like the reference's separate fp32 tensor ops, so no FMA contraction anywhere in here).
//
//   decode   : sigmoid / exp / grid, one thread per anchor                       (utils/bbox.py:36-76)
//   filter   : xyxy, class max, conf mask, stable compaction, one block per image (utils/bbox.py:98-130)
//   nms      : coordinate-trick batched NMS                                       (utils/bbox.py:133-138 -> torchvision)
//              bucket by class (max coordinate, offset boxes, scores, class groups) -> rank sort inside a group (stable,
//              descending) -> 64x64 IoU bitmask tiles inside a group -> per-group sweep, 64 boxes at a time resolved in
//              registers -> merge of the groups' kept lists by score + rescale (utils/bbox.py:145-158)
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

// ((raw + grid) * stride) / dim, (exp(raw) * stride) / dim, 1 / (1 + exp(-raw)): same operations on the same operands as
// bbox.py:39,70-76, arranged as ONE exponential and ONE division per element with selects around them -- a wave holds all
// channel kinds, and five divergent branches (each with its own IEEE division) made this VALU-bound.  Every per-level quantity
// is a scalar (blockIdx.z = level); the two index divisions go through a float reciprocal when every level has fewer than 2^22
// elements (two roundings of 2^-24 each keep the product inside its integer cell), not the ~30-instruction integer sequence.
// e = element inside the level: r * nch + c
__device__ __forceinline__ float decode_value(const DecodeP& p, int l, int b, unsigned e) {
    const int wl = p.w[l];
    const int r = p.fast ? (int)(((float)e + 0.5f) * p.rnch) : (int)(e / (unsigned)p.nch);
    const int c = (int)e - r * p.nch;
    const int gy = p.fast ? (int)(((float)r + 0.5f) * p.rw[l]) : r / wl;
    const int gx = r - gy * wl;
    const float raw = p.head[l][(long long)b * p.sb[l] + (long long)r * p.sp[l] + (long long)c * p.sc[l]];
    const float s = p.stride[l];
    const bool is_xy = c < 2, is_box = c < 4;
    const float ex = expf(is_box ? raw : -raw);
    const float g = (c == 0) ? (float)gx : (float)gy;
    const float num = is_xy ? (raw + g) * s : (is_box ? ex * s : 1.0f);
    const float den = is_box ? ((c & 1) ? p.in_h : p.in_w) : 1.0f + ex;
    return num / den;
}

// one thread per output ELEMENT (anchor, channel) of image blockIdx.y: with NHWC heads both the load and the store of a wave
// are contiguous.  Index arithmetic is 32-bit (64-bit divisions made this kernel VALU-bound: 31 us for 47 MB).
__global__ __launch_bounds__(256) void decode_kernel(const DecodeP p) {
    const int l = blockIdx.z, b = blockIdx.y;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    const unsigned n_l = (unsigned)(p.a0[l + 1] - p.a0[l]) * (unsigned)p.nch;
    if (e >= n_l) return;
    const unsigned per_img = (unsigned)p.A * (unsigned)p.nch;
    p.out[(size_t)b * per_img + (unsigned)p.a0[l] * (unsigned)p.nch + e] = decode_value(p, l, b, e);
}

// decode that also leaves, per anchor, the value post_process tests against its threshold (obj * max class probability, the
// fp32 product filter_keep forms): a block takes 256 anchors of one level, its 256 x nch elements go through the arithmetic
// above (loads and stores contiguous across the block) and through LDS, from where thread t reads anchor t's row.  The filter
// then reads 4 bytes per anchor instead of the whole row.
__global__ __launch_bounds__(256) void decode_scored_kernel(const DecodeP p, float* score) {
    extern __shared__ float sv[];  // [256][nch]
    const int l = blockIdx.z, b = blockIdx.y, t = threadIdx.x;
    const int n_r = p.a0[l + 1] - p.a0[l];
    const int r0 = blockIdx.x * 256;
    if (r0 >= n_r) return;
    const int rows = min(256, n_r - r0);
    const unsigned nel = (unsigned)rows * (unsigned)p.nch, e0 = (unsigned)r0 * (unsigned)p.nch;
    float* ob = p.out + (size_t)b * ((unsigned)p.A * (unsigned)p.nch) + (unsigned)p.a0[l] * (unsigned)p.nch + e0;
#pragma unroll 4
    for (unsigned i = t; i < nel; i += 256u) {
        const float v = decode_value(p, l, b, e0 + i);
        ob[i] = v;
        sv[i] = v;
    }
    __syncthreads();
    if (t < rows) {
        const float* r = sv + t * p.nch;
        const float obj = r[4];
        float cc = r[5];
        for (int c = 6; c < p.nch; ++c) {
            const float v = r[c];
            if (v > cc) cc = v;
        }
        score[(size_t)b * p.A + p.a0[l] + r0 + t] = obj * cc;
    }
}

// ---- filter: one block (1024 threads) per image, anchors in order, stable compaction.
// Per super-chunk of 32 x 1024 anchors: pass 1 evaluates the keep test of every anchor (independent loads, no barrier between
// them, so their latencies overlap) and leaves one flag bit per anchor in a register and one count per (iteration, wave) in
// LDS; one scan turns the counts into output offsets; pass 2 re-reads only the kept rows (L2-hot) and writes them.  The
// previous form paid a global-load round trip and three barriers per 1024 anchors (23 us for 8400 anchors x 64 images).
constexpr int F_IT = 32, F_SUPER = F_IT * 1024;
// keep test of one anchor: obj * max class score >= conf (fp32, the comparisons of pass 2)
__device__ __forceinline__ bool filter_keep(const float* r, int nc, float conf) {
    const float obj = r[4];
    float cc = r[5];
    for (int c = 1; c < nc; ++c) {
        const float v = r[5 + c];
        if (v > cc) cc = v;
    }
    return (obj * cc) >= conf;
}
// kept anchors per (image, super-chunk): lets filter_kernel run one block per super-chunk on maps with many anchors
__global__ __launch_bounds__(1024) void filter_count_kernel(const float* pred, const float* score, int A, int nc, float conf, int nsuper,
                                                           int* chunk_counts) {
    __shared__ int tot;
    const int b = blockIdx.x, sc = blockIdx.y, t = threadIdx.x;
    const int nch = 5 + nc;
    const float* pb = pred + (size_t)b * A * nch;
    if (t == 0) tot = 0;
    __syncthreads();
    int mine = 0;
    for (int a = sc * F_SUPER + t; a < min(A, (sc + 1) * F_SUPER); a += 1024)
        mine += (score ? score[(size_t)b * A + a] >= conf : filter_keep(pb + (size_t)a * nch, nc, conf)) ? 1 : 0;
    const unsigned long long any = __ballot(mine != 0);
    if (any) {
        int v = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((t & 63) == 0) atomicAdd(&tot, v);
    }
    __syncthreads();
    if (t == 0) chunk_counts[b * nsuper + sc] = tot;
}
// chunk_counts == nullptr: one block per image walks all super-chunks; otherwise block (image, y) does super-chunk y and starts
// at the sum of the earlier chunks' counts.  score (may be null): [B][A] values of obj * max class probability from
// decode_scored_kernel, the same fp32 product the keep test forms from the row.
__global__ __launch_bounds__(1024) void filter_kernel(const float* pred, const float* score, int A, int nc, float conf, float* cand, int* counts,
                                                     int nsuper, const int* chunk_counts) {
    __shared__ int cnt[F_IT * 16];
    __shared__ int base_s;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nch = 5 + nc;
    const float* pb = pred + (size_t)b * A * nch;
    float* cb = cand + (size_t)b * A * 7;
    const int s_first = chunk_counts ? (int)blockIdx.y * F_SUPER : 0;
    const int s_end = chunk_counts ? min(A, s_first + F_SUPER) : A;
    if (t == 0) {
        int base = 0;
        if (chunk_counts)
            for (int y = 0; y < (int)blockIdx.y; ++y) base += chunk_counts[b * nsuper + y];
        base_s = base;
    }
    for (int s0 = s_first; s0 < s_end; s0 += F_IT * 1024) {
        const int iters = min(F_IT, (A - s0 + 1023) / 1024);
        unsigned flags = 0u;
        for (int k = 0; k < iters; ++k) {
            const int a = s0 + k * 1024 + t;
            bool keep = false;
            if (a < A) {
                if (score) {
                    keep = score[(size_t)b * A + a] >= conf;
                } else {
                    const float* r = pb + (size_t)a * nch;
                    const float obj = r[4];
                    float cc = r[5];
                    for (int c = 1; c < nc; ++c) {  // the value of the maximum, by the same comparisons as pass 2 (its index is found there)
                        const float v = r[5 + c];
                        if (v > cc) cc = v;
                    }
                    keep = (obj * cc) >= conf;
                }
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
    if (t == 0 && (!chunk_counts || (int)blockIdx.y == nsuper - 1)) counts[b] = base_s;
}

// ---- NMS workspace and per-image tables
// torchvision's batched_nms adds cls * (max_coordinate + 1) to every box so that one global NMS never lets boxes of different
// classes meet.  The tiled path below keeps that arithmetic (the offset boxes are what every IoU is computed from, so the bits
// cannot change) and uses its consequence: the candidates of an image are bucketed by class, adjacent classes are merged into
// one GROUP unless the offset boxes of everything before the boundary and everything after it are disjoint in x or in y (then
// every cross pair has a zero intersection and can never suppress), and rank sort, pair bitmask and greedy sweep run per group.
// The kept boxes of the groups are merged back into one list by (score descending, candidate index ascending), which is the
// order the global algorithm emits.  A negative threshold (a zero overlap would suppress), a non-finite coordinate or a class
// id outside 0..255 makes the image one group: that is the global algorithm itself.
//
// Pair bitmask storage: row i (sorted position) of a group that ends in 64-box chunk `we` holds the words of chunks
// word(i)..we only, rows back to back (row_cum() is the closed form of the running total).  An image's rows are cut into
// at most NMS_PANELS panels of <= cap words, a chunk never straddles two; mask and sweep are launched once per panel on the
// same `cap` words, the sweep's state (removed bits, kept lists) persists in the workspace.  One class per image in the worst
// case needs all four panels; six even classes need one.
constexpr int NMS_MAXC = 256;    // class ids 0..255 are partitioned
constexpr int NMS_TAB = 1024;    // ints of table per image
constexpr int NMS_PANELS = 4;
constexpr int NMS_GSLOTS = 32;   // sweep blocks per image (a block takes groups slot, slot + 32, ...)
constexpr int T_G = 0, T_START = 8, T_CUM = 8 + 264, T_KEEP = 8 + 528;  // [T_START + g] first row of group g (.. G inclusive), [T_CUM + g] words before it, [T_KEEP + g] kept so far
struct NmsWs {
    size_t boxes, sboxes, bkey, skey, kkey, rowgrp, table, removed, mask, total;  // byte offsets
    int W;
    unsigned capq, cap;  // panel quantum and panel capacity in words per image
};
NmsWs nms_ws(int B, int nmax) {
    NmsWs w;
    w.W = (nmax + 63) / 64;
    const size_t np = (size_t)w.W * 64, nb = (size_t)B * np;
    size_t o = 0;
    w.boxes = o;  o += nb * 16;   // float4 [B][np]  offset boxes in bucket order
    w.sboxes = o; o += nb * 16;   // float4 [B][np]  offset boxes in sorted order (group-major, score descending inside a group)
    w.bkey = o;   o += nb * 8;    // sort key (score, candidate index) of a bucket / sorted position, see nms_key()
    w.skey = o;   o += nb * 8;
    w.kkey = o;   o += nb * 8;    // [first row of group g + k] = key of the group's k-th kept box
    w.rowgrp = o; o += nb * 2;    // group of a position (the same for bucket and sorted order)
    o = (o + 255) & ~(size_t)255;
    w.table = o;  o += (size_t)B * NMS_TAB * 4;
    w.removed = o; o += (size_t)B * (w.W + NMS_MAXC) * 8;  // each group's span of removed words, private (offset by its index)
    o = (o + 255) & ~(size_t)255;
    // all rows of an image hold at most 32 W (W + 1) words (one group); a quarter of that per panel, plus one chunk of slack
    // (small inputs: everything in one panel, up to 512 KB per image)
    const unsigned all = 32u * (unsigned)w.W * (unsigned)(w.W + 1);
    w.capq = all / 4u > (1u << 16) ? all / 4u : (all < (1u << 16) ? all : (1u << 16));
    w.cap = w.capq + 64u * (unsigned)w.W;
    w.mask = o;   o += (size_t)B * w.cap * 8;
    w.total = o;
    return w;
}

// sum of (r >> 6) over r < m
__device__ __forceinline__ unsigned tri64(unsigned m) {
    const unsigned k = m >> 6;
    return 32u * k * (k - 1u) + (m & 63u) * k;
}
// mask words of the image that precede sorted row i of group g (rows s..e-1, row r holds words (r >> 6)..((e-1) >> 6))
__device__ __forceinline__ unsigned row_cum(const int* tab, int g, int i) {
    const int s = tab[T_START + g], e = tab[T_START + g + 1];
    const unsigned we = (unsigned)(e - 1) >> 6;
    return (unsigned)tab[T_CUM + g] + (unsigned)(i - s) * (we + 1u) - (tri64((unsigned)i) - tri64((unsigned)s));
}
__device__ __forceinline__ int fkey(float f) {  // order-preserving float -> int (finite values)
    const int v = __float_as_int(f);
    return v >= 0 ? v : v ^ 0x7fffffff;
}
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
// One unsigned key per candidate: box j comes before box i in the stable descending order (s_j > s_i, or s_j == s_i and
// idx_j < idx_i) iff key_j > key_i.  High word: the score's bits made monotone (-0 counts as +0, like the float compare);
// low word: ~idx.
__device__ __forceinline__ unsigned long long nms_key(float score, int idx) {
    const unsigned u = __float_as_uint(score == 0.0f ? 0.0f : score);
    const unsigned m = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)m << 32) | (unsigned long long)(0xffffffffu - (unsigned)idx);
}
__device__ __forceinline__ int nms_key_idx(unsigned long long k) { return (int)(0xffffffffu - (unsigned)k); }
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
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

// bucket the candidates of an image by class, form the groups, write the tables.  One block per image.
//   offset boxes = boxes + cls * (max_coordinate + 1)  (torchvision batched_nms coordinate trick); scores = obj * cls_conf
__global__ __launch_bounds__(1024) void nms_bucket_kernel(const float* __restrict__ cand, const int* __restrict__ counts, int A, int np, int nmax, int W,
                                                          int split_ok, float4* boxes, unsigned long long* bkey, unsigned short* rowgrp, int* table,
                                                          unsigned long long* removed) {
    __shared__ int hist[NMS_MAXC], cursor[NMS_MAXC], gid[NMS_MAXC], cstart[NMS_MAXC + 1];
    __shared__ int lox[NMS_MAXC], loy[NMS_MAXC], hix[NMS_MAXC], hiy[NMS_MAXC];
    __shared__ float sfx[NMS_MAXC + 1], sfy[NMS_MAXC + 1];
    __shared__ float red[16];
    __shared__ int redw[16], redc[16];
    __shared__ float maxc_s;
    __shared__ int weird_s, ncls_s;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = min(counts[b], nmax);  // an image with more candidates than the caller sized for is cut, never overrun
    int* tab = table + (size_t)b * NMS_TAB;
    const float* cb = cand + (size_t)b * A * 7;
    float4* bx = boxes + (size_t)b * np;
    unsigned short* rg = rowgrp + (size_t)b * np;
    for (int c = t; c < NMS_MAXC; c += 1024) {
        hist[c] = 0;
        cursor[c] = 0;
        lox[c] = loy[c] = 0x7fffffff;
        hix[c] = hiy[c] = (int)0x80000000;
        tab[T_KEEP + c] = 0;
    }
    for (int w = t; w < W + NMS_MAXC; w += 1024) removed[(size_t)b * (W + NMS_MAXC) + w] = 0ull;
    __syncthreads();
    // pass 1: largest coordinate, class histogram, can the image be partitioned at all
    float m = -INFINITY;
    int weird = 0, cmax = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        const bool act = i < n;
        int ci = -1;
        if (act) {
            const float* r = cb + (size_t)i * 7;
            const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3], cl = r[6];
            m = fmaxf(m, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
            const float z = ((x1 - x1) + (y1 - y1)) + ((x2 - x2) + (y2 - y2));  // 0 iff every coordinate is finite
            const int c = (cl >= 0.0f && cl < (float)NMS_MAXC) ? (int)cl : -1;
            if (z == 0.0f && c >= 0 && (float)c == cl) {
                ci = c;
                cmax = max(cmax, c);
            } else {
                weird = 1;
            }
        }
        unsigned long long todo = __ballot(ci >= 0);
        while (todo) {  // one LDS atomic per (wave, class)
            const int leader = __ffsll((long long)todo) - 1;
            const int c = __builtin_amdgcn_readlane(ci, leader);
            const unsigned long long same = __ballot(ci == c);
            if (lane == leader) atomicAdd(&hist[c], __popcll(same));
            todo &= ~same;
        }
    }
    m = wave_max(m);
    const float cmf = wave_max((float)cmax);
    const unsigned long long anyw = __ballot(weird != 0);
    if (lane == 0) {
        red[wave] = m;
        redc[wave] = (int)cmf;
        redw[wave] = anyw != 0ull;
    }
    __syncthreads();
    if (t == 0) {
        float mm = red[0];
        int cm = redc[0], wd = redw[0];
        for (int q = 1; q < 16; ++q) {
            mm = fmaxf(mm, red[q]);
            cm = max(cm, redc[q]);
            wd |= redw[q];
        }
        wd |= !split_ok;
        maxc_s = mm;
        weird_s = wd;
        const int ncls = wd ? 1 : cm + 1;
        ncls_s = ncls;
        if (wd) hist[0] = n;
        int acc = 0;
        for (int c = 0; c < ncls; ++c) {
            cstart[c] = acc;
            acc += hist[c];
        }
        cstart[ncls] = acc;
    }
    __syncthreads();
    const float step = maxc_s + 1.0f;
    const int wd = weird_s, ncls = ncls_s;
    // pass 2: placement by class (any order inside a class: the rank below breaks score ties by candidate index)
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        const bool act = i < n;
        float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0;
        if (act) {
            const float* r = cb + (size_t)i * 7;
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; r4 = r[4]; r5 = r[5]; r6 = r[6];
        }
        const int ci = act ? (wd ? 0 : (int)r6) : -1;
        int pos = 0;
        unsigned long long todo = __ballot(ci >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int c = __builtin_amdgcn_readlane(ci, leader);
            const unsigned long long same = __ballot(ci == c);
            int base = 0;
            if (lane == leader) base = atomicAdd(&cursor[c], __popcll(same));
            base = __builtin_amdgcn_readlane(base, leader);
            if (ci == c) pos = cstart[c] + base + __popcll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        if (act) {
            const float off = r6 * step;
            bx[pos] = make_float4(r0 + off, r1 + off, r2 + off, r3 + off);
            bkey[(size_t)b * np + pos] = nms_key(r4 * r5, i);
            rg[pos] = (unsigned short)ci;
        }
    }
    __syncthreads();
    // pass 3: extent of each class's offset boxes (a wave whose 64 positions are one class reduces in registers)
    if (!wd) {
        for (int p0 = wave * 64; p0 < n; p0 += 1024) {
            const int p = p0 + lane;
            const bool act = p < n;
            const float4 v = bx[act ? p : p0];
            const int c = rg[act ? p : p0];
            const int c0 = __builtin_amdgcn_readfirstlane(c);
            if (__ballot(c != c0) == 0ull) {
                const float ax = -wave_max(-v.x), ay = -wave_max(-v.y), zx = wave_max(v.z), zy = wave_max(v.w);
                if (lane == 0) {
                    atomicMin(&lox[c0], fkey(ax));
                    atomicMin(&loy[c0], fkey(ay));
                    atomicMax(&hix[c0], fkey(zx));
                    atomicMax(&hiy[c0], fkey(zy));
                }
            } else {
                atomicMin(&lox[c], fkey(v.x));
                atomicMin(&loy[c], fkey(v.y));
                atomicMax(&hix[c], fkey(v.z));
                atomicMax(&hiy[c], fkey(v.w));
            }
        }
    }
    __syncthreads();
    // groups: a boundary in front of class c iff everything before it and everything from it on are disjoint in x or in y
    if (t == 0) {
        int G = 0;
        if (n > 0) {
            if (wd) {
                gid[0] = 0;
                tab[T_START] = 0;
                G = 1;
            } else {
                sfx[ncls] = sfy[ncls] = INFINITY;
                for (int c = ncls - 1; c >= 0; --c) {
                    sfx[c] = hist[c] ? fminf(sfx[c + 1], fkey_inv(lox[c])) : sfx[c + 1];
                    sfy[c] = hist[c] ? fminf(sfy[c + 1], fkey_inv(loy[c])) : sfy[c + 1];
                }
                float phx = -INFINITY, phy = -INFINITY;
                int g = -1;
                for (int c = 0; c < ncls; ++c) {
                    if (hist[c] == 0) {
                        gid[c] = max(g, 0);
                        continue;
                    }
                    if (g < 0 || phx <= sfx[c] || phy <= sfy[c]) {
                        ++g;
                        tab[T_START + g] = cstart[c];
                    }
                    gid[c] = g;
                    phx = fmaxf(phx, fkey_inv(hix[c]));
                    phy = fmaxf(phy, fkey_inv(hiy[c]));
                }
                G = g + 1;
            }
        }
        tab[T_G] = G;
        tab[T_START + G] = n;
        unsigned cum = 0;
        for (int g = 0; g < G; ++g) {
            const int s = tab[T_START + g], e = tab[T_START + g + 1];
            const unsigned we = (unsigned)(e - 1) >> 6;
            tab[T_CUM + g] = (int)cum;
            cum += (unsigned)(e - s) * (we + 1u) - (tri64((unsigned)e) - tri64((unsigned)s));
        }
        tab[T_CUM + G] = (int)cum;
    }
    __syncthreads();
    for (int p = t; p < n; p += 1024) rg[p] = (unsigned short)gid[rg[p]];
}

// stable descending rank inside the group: rank(i) = #{j in the group : key_j > key_i}.  A wave owns 16 rows whose keys sit in
// scalar registers and streams the group's keys 64 at a time, one per lane: a pair costs 1/64 of a 64-bit compare plus two
// scalar instructions (count the ballot, add), instead of half a dozen vector instructions per pair and thread.
constexpr int RANK_R = 16;
__global__ __launch_bounds__(256) void nms_rank_kernel(const int* __restrict__ counts, int np, int nmax, const float4* __restrict__ boxes,
                                                       const unsigned long long* __restrict__ bkey, const unsigned short* __restrict__ rowgrp,
                                                       const int* __restrict__ table, float4* sboxes, unsigned long long* skey) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int n = min(counts[b], nmax);
    const int p0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * RANK_R);
    if (p0 >= n) return;
    const int* tab = table + (size_t)b * NMS_TAB;
    const size_t ib = (size_t)b * np;
    const unsigned long long* kb = bkey + ib;
    const int gf = rowgrp[ib + p0], gl = rowgrp[ib + min(p0 + RANK_R - 1, n - 1)];
    const int lo = tab[T_START + gf], hi = tab[T_START + gl + 1];
    unsigned long long ki[RANK_R];
    int acc[RANK_R];
#pragma unroll
    for (int r = 0; r < RANK_R; ++r) {
        ki[r] = uniform64(kb[min(p0 + r, n - 1)]);
        acc[r] = 0;
    }
    if (gf == gl) {
        for (int j0 = lo; j0 < hi; j0 += 256) {
            unsigned long long kj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kj[u] = (j0 + u * 64 + lane < hi) ? kb[j0 + u * 64 + lane] : 0ull;  // 0 is above no key
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < RANK_R; ++r) acc[r] += __popcll(__ballot(kj[u] > ki[r]));
        }
    } else {  // the 16 rows straddle a group boundary: each row counts inside its own group only
        int rs[RANK_R], re[RANK_R];
#pragma unroll
        for (int r = 0; r < RANK_R; ++r) {
            const int g = rowgrp[ib + min(p0 + r, n - 1)];
            rs[r] = tab[T_START + g];
            re[r] = tab[T_START + g + 1];
        }
        for (int j0 = lo; j0 < hi; j0 += 64) {
            const int j = j0 + lane;
            const unsigned long long kj = j < hi ? kb[j] : 0ull;
#pragma unroll
            for (int r = 0; r < RANK_R; ++r) acc[r] += __popcll(__ballot(kj > ki[r] && j >= rs[r] && j < re[r]));
        }
    }
    int mine = 0;
#pragma unroll
    for (int r = 0; r < RANK_R; ++r) mine = lane == r ? acc[r] : mine;
    if (lane < RANK_R && p0 + lane < n) {
        const int g = rowgrp[ib + p0 + lane];
        const size_t q = ib + tab[T_START + g] + mine;
        sboxes[q] = boxes[ib + p0 + lane];
        skey[q] = kb[p0 + lane];
    }
}

// pair bitmask of one panel.  Block = one 64-row chunk of the sorted list, lane = row; the four waves share the chunk's column
// tiles (its own chunk .. the last chunk of the last group it touches).  A tile is done in two phases: a FILTER over all 64
// columns -- four compares per pair: the boxes overlap in x and in y, which every pair with a non-zero intersection passes --
// and the exact torchvision arithmetic (suppresses()) on the pairs that passed, each lane walking its own candidate bits.  With
// a threshold >= 0 a pair with a zero (or NaN) intersection can never suppress, so the filter cannot change a bit.
__global__ __launch_bounds__(256) void nms_mask_kernel(const int* __restrict__ counts, int np, int nmax, unsigned capq, unsigned cap, int panel,
                                                       const float4* __restrict__ sboxes, const unsigned short* __restrict__ rowgrp,
                                                       const int* __restrict__ table, double thr, int skip_zero, unsigned long long* mask) {
    __shared__ float4 cols_s[4][64];
    const int b = blockIdx.y, c = blockIdx.x;
    const int n = min(counts[b], nmax);
    const int r0 = c * 64;
    if (r0 >= n) return;
    const int* tab = table + (size_t)b * NMS_TAB;
    const float4* sb = sboxes + (size_t)b * np;
    const unsigned short* rg = rowgrp + (size_t)b * np;
    if ((int)(row_cum(tab, rg[r0], r0) / capq) != panel) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = r0 + lane, il = min(i, n - 1);
    const int g = rg[il];
    const int we = (tab[T_START + g + 1] - 1) >> 6;
    const int wmax = (tab[T_START + rg[min(r0 + 63, n - 1)] + 1] - 1) >> 6;
    unsigned long long* mrow = mask + (size_t)b * cap + (row_cum(tab, g, il) - (unsigned)panel * capq);
    const float4 bi = sb[il];
    const float ai = (bi.z - bi.x) * (bi.w - bi.y);
    float4* cols = cols_s[wave];
    for (int ct = c + wave; ct <= wmax; ct += 4) {
        const int jn = min(64, n - ct * 64);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cols[lane] = sb[min(ct * 64 + lane, n - 1)];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned lo = 0u, hi = 0u;
        if (skip_zero) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float4 bj = cols[j];
                const bool hit = (bi.z > bj.x) & (bj.z > bi.x) & (bi.w > bj.y) & (bj.w > bi.y);
                if (j < 32) lo |= hit ? (1u << j) : 0u; else hi |= hit ? (1u << (j - 32)) : 0u;
            }
        } else {
            lo = hi = ~0u;
        }
        unsigned long long todo = ((unsigned long long)hi << 32) | lo;
        if (jn < 64) todo &= (1ull << jn) - 1ull;
        // the diagonal tile is kept SYMMETRIC (all j != i: the overlap test is symmetric bit for bit), so that the sweep has every
        // row's predecessors without a transpose
        if (ct == c) todo &= ~(1ull << lane);
        unsigned long long bits = 0ull;
        while (__ballot(todo != 0ull)) {
            if (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                if (suppresses(bi, ai, cols[j], thr)) bits |= 1ull << j;
            }
        }
        if (i < n && ct <= we) mrow[ct - c] = bits;
    }
}

// greedy sweep of one panel: a block takes whole groups, 64 sorted boxes (one chunk) per step.  Wave 0 walks the dependent
// chain: the chunk's diagonal words and the words of the NEXT chunk were fetched one step ahead, the 64 boxes are resolved in
// scalar registers, and what the kept ones remove from the next chunk is OR-ed in at once.  The fifteen other waves run one
// step behind: they OR the mask rows of the previous chunk's kept boxes into the removed words two or more chunks ahead (L2
// round trips that the chain no longer waits for).
constexpr int SWEEP_T = 1024, SWEEP_H = SWEEP_T / 64 - 1;
__global__ __launch_bounds__(SWEEP_T) void nms_sweep_kernel(const int* __restrict__ counts, int np, int W, int nmax, unsigned capq, unsigned cap, int panel,
                                                            const unsigned long long* __restrict__ mask, const unsigned short* __restrict__ rowgrp,
                                                            const unsigned long long* __restrict__ skey, int* table, unsigned long long* removed_g,
                                                            unsigned long long* kkey) {
    extern __shared__ unsigned long long removed[];  // the group's span of words (<= W)
    __shared__ int kept_l[2][64];  // lanes (0..63) of the boxes kept in the current / previous chunk
    __shared__ int kept_n[2];
    const int b = blockIdx.y, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int n = min(counts[b], nmax);
    if (n <= 0) return;
    int* tab = table + (size_t)b * NMS_TAB;
    const int G = tab[T_G];
    const unsigned long long* mb = mask + (size_t)b * cap;
    const unsigned long long* sk = skey + (size_t)b * np;
    const unsigned pbase = (unsigned)panel * capq;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const int s = tab[T_START + g], e = tab[T_START + g + 1];
        const int c0 = s >> 6, we = (e - 1) >> 6, nwg = we - c0 + 1;
        const unsigned gcum = (unsigned)tab[T_CUM + g];
        // words before row r of this group / before the first row of chunk c (a row of an earlier group iff c == c0 and s is mid-chunk)
        auto own_cum = [&](int r) -> unsigned { return gcum + (unsigned)(r - s) * (unsigned)(we + 1) - (tri64((unsigned)r) - tri64((unsigned)s)); };
        auto chunk_panel = [&](int c) -> int {
            const int r0 = c * 64;
            return (int)((r0 >= s ? own_cum(r0) : row_cum(tab, rowgrp[(size_t)b * np + r0], r0)) / capq);
        };
        if (chunk_panel(c0) > panel || chunk_panel(we) < panel) continue;
        unsigned long long* rem_g = removed_g + (size_t)b * (W + NMS_MAXC) + c0 + g;
        __syncthreads();
        for (int w = t; w < nwg; w += SWEEP_T) removed[w] = rem_g[w];
        __syncthreads();
        int total = tab[T_KEEP + g];  // meaningful in wave 0
        // first and last chunk of the group inside this panel
        int cf = c0;
        while (chunk_panel(cf) < panel) ++cf;
        int cl = cf;
        while (cl < we && chunk_panel(cl + 1) == panel) ++cl;
        // rows of chunk c that belong to the group: lanes a..z-1, `cnt` words each, the first at `rows`
        auto lanes_a = [&](int c) { return max(s - c * 64, 0); };
        auto lanes_z = [&](int c) { return min(e - c * 64, 64); };
        auto rows_of = [&](int c) { return mb + (own_cum(c * 64 + lanes_a(c)) - pbase); };
        unsigned long long d0 = 0ull, d1 = 0ull, k0 = 0ull;  // wave 0: diagonal word, next chunk's word and key of lane's row
        if (wave == 0) {
            const int a = lanes_a(cf), z = lanes_z(cf);
            if (lane >= a && lane < z) {
                const unsigned long long* r = rows_of(cf) + (size_t)(lane - a) * (unsigned)(we - cf + 1);
                d0 = r[0];
                d1 = cf < we ? r[1] : 0ull;
                k0 = sk[cf * 64 + lane];
            }
        }
        for (int c = cf; c <= cl + 1; ++c) {  // step cl + 1: only the helpers, for the last chunk's kept rows
            const int cur = (c - cf) & 1;
            if (wave == 0) {
                if (c <= cl) {
                    unsigned long long e0 = 0ull, e1 = 0ull, k1 = 0ull;
                    if (c < cl) {  // fetch one step ahead
                        const int a = lanes_a(c + 1), z = lanes_z(c + 1);
                        if (lane >= a && lane < z) {
                            const unsigned long long* r = rows_of(c + 1) + (size_t)(lane - a) * (unsigned)(we - c);
                            e0 = r[0];
                            e1 = c + 1 < we ? r[1] : 0ull;
                            k1 = sk[(c + 1) * 64 + lane];
                        }
                    }
                    const int a = lanes_a(c), z = lanes_z(c);
                    // Greedy over the chunk without walking it box by box: d0 is the row of the symmetric overlap matrix, so
                    // its bits below the lane are the boxes that would suppress this one if kept.  A box is removed as soon as
                    // one of them is kept, kept as soon as none of them is kept or undecided; the lowest undecided box always
                    // decides, and typical chunks settle in a few rounds instead of 64 dependent steps.
                    const unsigned long long pred = d0 & ((1ull << lane) - 1ull);
                    const unsigned long long span = (z >= 64 ? ~0ull : (1ull << z) - 1ull) & ~((1ull << a) - 1ull);
                    unsigned long long und = span & ~uniform64(removed[c - c0]);
                    unsigned long long kept = 0ull;
                    while (und) {
                        const bool u = (und >> lane) & 1ull;
                        const unsigned long long out = __ballot(u && (pred & kept) != 0ull);
                        const unsigned long long in = __ballot(u && (pred & (kept | und)) == 0ull);
                        kept |= in;
                        und &= ~(in | out);
                    }
                    const bool mine = (kept >> lane) & 1ull;
                    if (mine && d1) atomicOr(&removed[c + 1 - c0], d1);
                    const int before = __popcll(kept & ((1ull << lane) - 1ull));
                    if (mine) {
                        kkey[(size_t)b * np + s + total + before] = k0;
                        kept_l[cur][before] = lane;
                    }
                    total += __popcll(kept);
                    if (lane == 0) kept_n[cur] = __popcll(kept);
                    d0 = e0; d1 = e1; k0 = k1;
                }
            } else if (c > cf) {
                // helpers: rows kept in chunk c - 1, words of chunks c + 1 .. we.  Helper h takes kept rows h, h + 15, ...; all of
                // a lane's loads are issued before the first OR (clamped index: a repeated OR is harmless)
                const int pc = c - 1, prev = cur ^ 1;
                const int nk = kept_n[prev];
                const int h = wave - 1;
                const int mine = (nk - h + SWEEP_H - 1) / SWEEP_H;
                if (mine > 0 && pc + 2 <= we) {
                    const int a = lanes_a(pc);
                    const unsigned cnt = (unsigned)(we - pc + 1);
                    const unsigned long long* rows = rows_of(pc);
                    for (int w = pc + 2 + lane; w <= we; w += 64) {
                        unsigned long long v[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) v[u] = rows[(size_t)(kept_l[prev][h + SWEEP_H * min(u, mine - 1)] - a) * cnt + (w - pc)];
                        const unsigned long long acc = (v[0] | v[1]) | (v[2] | v[3]) | v[4];
                        if (acc) atomicOr(&removed[w - c0], acc);
                    }
                }
            }
            __syncthreads();
        }
        for (int w = t; w < nwg; w += SWEEP_T) rem_g[w] = removed[w];
        if (t == 0) tab[T_KEEP + g] = total;
    }
}

// merge the groups' kept lists into the global order (score descending, candidate index ascending: descending key) and write
// the rows: det row = cand[keep] rescaled like the numpy tail of post_process (bbox.py:145-158).  The output slot of a kept box
// is its place in its own list plus, for every other group, the number of that group's kept keys above it -- binary searches,
// eight groups side by side so that their (dependent) loads overlap.
__global__ __launch_bounds__(256) void nms_merge_kernel(const float* __restrict__ cand, const int* __restrict__ counts, int A, int np, int nmax,
                                                        const unsigned short* __restrict__ rowgrp, const int* __restrict__ table,
                                                        const unsigned long long* __restrict__ kkey, const int* __restrict__ image_hw, float* det,
                                                        int* keep_idx, int* nkeep) {
    const int b = blockIdx.y;
    const int n = min(counts[b], nmax);
    const int* tab = table + (size_t)b * NMS_TAB;
    const int G = n > 0 ? tab[T_G] : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int tot = 0;
        for (int g = 0; g < G; ++g) tot += tab[T_KEEP + g];
        nkeep[b] = tot;
    }
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const size_t ib = (size_t)b * np;
    const int g = rowgrp[ib + q];
    const int k = q - tab[T_START + g];
    if (k >= tab[T_KEEP + g]) return;
    const unsigned long long key = kkey[ib + q];
    const int id = nms_key_idx(key);
    int out = k;
    for (int h0 = 0; h0 < G; h0 += 8) {
        const unsigned long long* kl[8];
        int lo[8], hi[8];
        int open = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int h = h0 + u;
            const bool on = h < G && h != g;
            kl[u] = kkey + ib + tab[T_START + (h < G ? h : 0)];
            lo[u] = 0;
            hi[u] = on ? tab[T_KEEP + h] : 0;
            open |= hi[u];
        }
        while (open) {  // kept boxes of group h in front of this one
            open = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int mid = (lo[u] + hi[u]) >> 1;
                const bool go = lo[u] < hi[u];
                const unsigned long long km = go ? kl[u][mid] : 0ull;
                if (go && km > key) lo[u] = mid + 1; else if (go) hi[u] = mid;
                open |= hi[u] - lo[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) out += lo[u];
    }
    keep_idx[(size_t)b * nmax + out] = id;
    write_det(cand + ((size_t)b * A + id) * 7, image_hw, b, det + ((size_t)b * nmax + out) * 7);
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

extern "C" int ccnet_decode_scored(const float* const* head_dev, const int64_t* sb, const int64_t* sc, const int64_t* sp, const int32_t* h,
                                   const int32_t* w, int nlev, int B, int nc, float in_h, float in_w, float* out_dev, float* score_dev,
                                   ccnet_stream stream) {
    CC_REQUIRE(head_dev && sb && sc && sp && h && w && out_dev, "decode: null pointer");
    CC_REQUIRE(!score_dev || (5 + nc) * 1024 <= 65536, "decode: scores need 5 + nc <= 64 (got nc=%d)", nc);
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
    const unsigned rblocks = (unsigned)((biggest / (5 + nc) + 255) / 256);
    return ccnet_submit(
        [p, blocks, rblocks, B, nlev, score_dev](hipStream_t st, const float*) -> int {
            if (score_dev)
                hipLaunchKernelGGL(decode_scored_kernel, dim3(rblocks, B, nlev), dim3(256), (size_t)p.nch * 1024, st, p, score_dev);
            else
                hipLaunchKernelGGL(decode_kernel, dim3(blocks, B, nlev), dim3(256), 0, st, p);
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_DECODE, 0.0, 2.0 * B * (double)A * (5 + nc) * 4.0);
}

extern "C" int ccnet_decode(const float* const* head_dev, const int64_t* sb, const int64_t* sc, const int64_t* sp, const int32_t* h, const int32_t* w,
                            int nlev, int B, int nc, float in_h, float in_w, float* out_dev, ccnet_stream stream) {
    return ccnet_decode_scored(head_dev, sb, sc, sp, h, w, nlev, B, nc, in_h, in_w, out_dev, nullptr, stream);
}

extern "C" size_t ccnet_filter_workspace_ints(int B, int A) {
    if (B <= 0 || A <= F_SUPER) return 0;
    return (size_t)B * ((A + F_SUPER - 1) / F_SUPER);
}

extern "C" int ccnet_filter_scored(const float* pred_dev, const float* score_dev, int B, int A, int nc, float conf_thres, float* cand_dev,
                                   int32_t* counts_dev, int32_t* workspace_dev, ccnet_stream stream) {
    CC_REQUIRE(pred_dev && cand_dev && counts_dev, "filter: null pointer");
    CC_REQUIRE(B > 0 && B <= 65535 && A > 0 && nc >= 1 && nc <= 1024, "filter: bad sizes (B=%d A=%d nc=%d)", B, A, nc);
    const int nsuper = (A + F_SUPER - 1) / F_SUPER;
    CC_REQUIRE(nsuper <= 65535, "filter: too many anchors (%d)", A);
    int32_t* cc = nsuper > 1 ? workspace_dev : nullptr;  // without a workspace one block per image walks every super-chunk
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            if (cc) {
                hipLaunchKernelGGL(filter_count_kernel, dim3(B, nsuper), dim3(1024), 0, st, pred_dev, score_dev, A, nc, conf_thres, nsuper, cc);
                hipLaunchKernelGGL(filter_kernel, dim3(B, nsuper), dim3(1024), 0, st, pred_dev, score_dev, A, nc, conf_thres, cand_dev, counts_dev,
                                   nsuper, cc);
            } else {
                hipLaunchKernelGGL(filter_kernel, dim3(B), dim3(1024), 0, st, pred_dev, score_dev, A, nc, conf_thres, cand_dev, counts_dev, nsuper,
                                   (const int*)nullptr);
            }
            CC_LAUNCH_CHECK();
            return CCNET_OK;
        },
        stream, CCNET_OP_FILTER, 0.0, (double)B * A * (score_dev ? 1 : 5 + nc) * 4.0);
}

extern "C" int ccnet_filter_ws(const float* pred_dev, int B, int A, int nc, float conf_thres, float* cand_dev, int32_t* counts_dev,
                               int32_t* workspace_dev, ccnet_stream stream) {
    return ccnet_filter_scored(pred_dev, nullptr, B, A, nc, conf_thres, cand_dev, counts_dev, workspace_dev, stream);
}

extern "C" int ccnet_filter(const float* pred_dev, int B, int A, int nc, float conf_thres, float* cand_dev, int32_t* counts_dev,
                            ccnet_stream stream) {
    return ccnet_filter_ws(pred_dev, B, A, nc, conf_thres, cand_dev, counts_dev, nullptr, stream);
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
    float4* boxes = (float4*)(base + ws.boxes);
    float4* sboxes = (float4*)(base + ws.sboxes);
    unsigned long long* bkey = (unsigned long long*)(base + ws.bkey);
    unsigned long long* skey = (unsigned long long*)(base + ws.skey);
    unsigned long long* kkey = (unsigned long long*)(base + ws.kkey);
    unsigned short* rowgrp = (unsigned short*)(base + ws.rowgrp);
    int* table = (int*)(base + ws.table);
    unsigned long long* removed = (unsigned long long*)(base + ws.removed);
    unsigned long long* mask = (unsigned long long*)(base + ws.mask);
    const unsigned capq = ws.capq, cap = ws.cap;
    const int split_ok = nms_thres >= 0.0 ? 1 : 0;  // below zero a ZERO overlap suppresses, across classes too: one group
    const int panels = 32u * (unsigned)W * (unsigned)(W + 1) <= capq ? 1 : NMS_PANELS;  // small inputs always fit one panel
    return ccnet_submit(
        [=](hipStream_t st, const float*) -> int {
            const dim3 rows((nmax + 255) / 256, B);
            hipLaunchKernelGGL(nms_bucket_kernel, dim3(B), dim3(1024), 0, st, cand_dev, counts_dev, A, np, nmax, W, split_ok, boxes, bkey, rowgrp, table,
                               removed);
            hipLaunchKernelGGL(nms_rank_kernel, dim3((nmax + 4 * RANK_R - 1) / (4 * RANK_R), B), dim3(256), 0, st, counts_dev, np, nmax, boxes, bkey,
                               rowgrp, table, sboxes, skey);
            for (int panel = 0; panel < panels; ++panel) {
                hipLaunchKernelGGL(nms_mask_kernel, dim3(W, B), dim3(256), 0, st, counts_dev, np, nmax, capq, cap, panel, sboxes, rowgrp, table,
                                   nms_thres, split_ok, mask);
                hipLaunchKernelGGL(nms_sweep_kernel, dim3(NMS_GSLOTS, B), dim3(SWEEP_T), (size_t)W * 8, st, counts_dev, np, W, nmax, capq, cap, panel,
                                   mask, rowgrp, skey, table, removed, kkey);
            }
            hipLaunchKernelGGL(nms_merge_kernel, rows, dim3(256), 0, st, cand_dev, counts_dev, A, np, nmax, rowgrp, table, kkey, image_hw_dev, det_dev,
                               keep_idx_dev, nkeep_dev);
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
// The NMS workspace (offset boxes, sorted boxes, sort keys, group tables, pair bitmask panel) lives in a file-scope device array
// instead of caller memory.  Each GPU the library runs on holds its own copy at its own address for the life of the process; the
// array's NAME is only a host-side stand-in, so the address is resolved once per device with hipGetSymbolAddress and only that
// pointer is ever handed to a kernel.  What has to start at zero -- the sweep's removed bits and kept counts -- is cleared by the
// call's first kernel (nms_bucket_kernel); every other word a kernel reads was written earlier in the same call.
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
    if (stream != nullptr) {   // the scratch pointer below belongs to the CURRENT device: a stream of another GPU would get the wrong one
        int cur = -1;
        hipDevice_t sdev = -1;
        CC_HIP(hipGetDevice(&cur));
        CC_HIP(hipStreamGetDevice((hipStream_t)stream, &sdev));
        CC_REQUIRE((int)sdev == cur, "nms_noalloc: the stream belongs to device %d but device %d is current; make the stream's device current first",
                   (int)sdev, cur);
    }
    unsigned char* ptr = nullptr;
    const int rc = scratch_for_current_device(&ptr);
    if (rc != CCNET_OK) return rc;
    return nms_submit("nms_noalloc", cand_dev, counts_dev, B, A, nmax, nms_thres, image_hw_dev, det_dev, keep_idx_dev, nkeep_dev, ptr, stream);
}
