// Image preparation in front of the hot path: Pillow-exact bicubic resize of 8-bit images, the
// reference's normalisation and HWC -> CHW, on device.
//
// Replaces tools/eval.py:134-140 (`image.resize((W, H), Image.BICUBIC)`, /255, mean/std, transpose) and the letterbox of
// utils/datasets.py:123-134 (+ :73-77).  The resize is Pillow's ImagingResample (src/libImaging/Resample.c, third party):
// per axis a table of (first tap, tap count) and 22-bit fixed-point weights; a horizontal pass into an 8-bit intermediate
// image, then a vertical pass, each clip8((2^21 + sum(pixel*k)) >> 22).  Integer work: results are bit-exact.
//
// Two kernels, both HBM-bound byte work (no MFMA): resample_h writes the intermediate as RGBX words, resample_v reads it
// back (tap re-reads hit L2: consecutive output rows share source rows), maps each 8-bit result through a 3x256 fp32
// table (the normalisation has only 256 possible inputs per channel; the host fills the table with the reference's own
// numpy arithmetic) and writes the three planes with 16-byte stores.
#include <math.h>
#include "common.h"

#pragma clang fp contract(off)  // the weight tables must round like Pillow's plain C doubles

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c
constexpr int KS_MAX = 4096;

inline double bicubic_filter(double x) {  // Resample.c: bicubic_filter, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

inline void axis_scale(int in_size, int out_size, double* scale, double* filterscale, double* support, int* ksize) {
    *filterscale = *scale = (double)in_size / out_size;
    if (*filterscale < 1.0) *filterscale = 1.0;
    *support = 2.0 * *filterscale;  // bicubic support 2
    *ksize = (int)ceil(*support) * 2 + 1;
}

__device__ __forceinline__ uint32_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint32_t)min(max(v, 0), 255);
}

struct HP {
    const uint8_t* src; long long img_stride, row_stride; int pix_stride, chan_step;
    int ih, iw, nw, ks;
    const int32_t* bounds; const int32_t* kk;
    uint32_t* tmp;
};

// one thread per intermediate pixel: tmp[n][y][x] = (r, g, b, 0)
__global__ __launch_bounds__(256) void resample_h_kernel(HP p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.ih * p.nw) return;
    const int y = (int)(idx / p.nw), x = (int)(idx - (long long)y * p.nw);
    const int n = blockIdx.y;
    const int first = min(max(p.bounds[2 * x], 0), p.iw - 1);
    const int cnt = min(max(p.bounds[2 * x + 1], 0), p.ks);
    const uint8_t* row = p.src + n * p.img_stride + y * p.row_stride;
    const int32_t* k = p.kk + (long long)x * p.ks;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int j = 0; j < cnt; ++j) {
        const uint8_t* px = row + (long long)min(first + j, p.iw - 1) * p.pix_stride;
        const int w = k[j];
        s0 += (int)px[0] * w;
        s1 += (int)px[p.chan_step] * w;
        s2 += (int)px[2 * p.chan_step] * w;
    }
    p.tmp[((long long)n * p.ih + y) * p.nw + x] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16);
}

// The same pass with the block's operands staged in LDS: a block owns H_XC consecutive output columns of H_RB source
// rows.  Its weight records (H_XC x ks words, contiguous in the table) and the byte span of each source row that those
// columns touch are fetched with coalesced loads; the per-tap byte reads then hit LDS instead of issuing one vector
// memory instruction per byte (which bound the direct kernel: 22 memory instructions per output pixel, now ~3).
constexpr int H_XC = 256, H_RB = 4;

__global__ __launch_bounds__(256) void resample_h_lds_kernel(HP p, int pitch) {
    extern __shared__ __align__(16) unsigned char hsm[];
    int32_t* lk = (int32_t*)hsm;                         // [H_XC][ks]
    unsigned char* ls = hsm + (size_t)H_XC * p.ks * 4;   // [H_RB][pitch]
    const int x0 = blockIdx.x * H_XC, y0 = blockIdx.y * H_RB, n = blockIdx.z;
    const int ncol = min(H_XC, p.nw - x0), nrow = min(H_RB, p.ih - y0);
    const int32_t* kg = p.kk + (long long)x0 * p.ks;
    for (int i = threadIdx.x; i < ncol * p.ks; i += 256) lk[i] = kg[i];
    const int smin = min(max(p.bounds[2 * x0], 0), p.iw - 1);
    const int last = x0 + ncol - 1;
    const int smax = min(max(p.bounds[2 * last] + p.bounds[2 * last + 1], smin + 1), p.iw);
    const int span = min((smax - smin) * p.pix_stride, pitch);
    const uint8_t* img = p.src + n * p.img_stride + (long long)smin * p.pix_stride;
    for (int r = 0; r < nrow; ++r) {
        const uint8_t* row = img + (y0 + r) * p.row_stride;
        for (int i = threadIdx.x; i < span; i += 256) ls[r * pitch + i] = row[i];
    }
    __syncthreads();
    const int xl = threadIdx.x;
    if (xl >= ncol) return;
    const int x = x0 + xl;
    const int first = min(max(p.bounds[2 * x], smin), p.iw - 1);
    const int cnt = min(max(p.bounds[2 * x + 1], 0), p.ks);
    const int32_t* k = lk + xl * p.ks;
    for (int r = 0; r < nrow; ++r) {
        const unsigned char* base = ls + r * pitch;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int j = 0; j < cnt; ++j) {
            const int off = min((min(first + j, p.iw - 1) - smin) * p.pix_stride, span - p.pix_stride);
            const unsigned char* px = base + off;
            const int w = k[j];
            s0 += (int)px[0] * w;
            s1 += (int)px[p.chan_step] * w;
            s2 += (int)px[2 * p.chan_step] * w;
        }
        p.tmp[((long long)n * p.ih + y0 + r) * p.nw + x] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16);
    }
}

struct VP {
    const uint32_t* tmp; int ih, nw, nh, dx, dy, ks;
    const int32_t* bounds; const int32_t* kk;
    const float* lut; int fill;
    float* out; long long out_img_stride, out_chan_stride; int H, W;
};

// VEC output pixels of one row per thread and V_ITEMS such items per thread (the table load and the barrier are paid
// once per block); the paste rectangle is VEC-aligned when VEC == 4
constexpr int V_ITEMS = 8;

template <int VEC>
__global__ __launch_bounds__(256) void resample_v_kernel(VP p) {
    __shared__ float lut[768];
    for (int i = threadIdx.x; i < 768; i += 256) lut[i] = p.lut[i];
    __syncthreads();
    const int wq = p.W / VEC;
    const long long total = (long long)p.H * wq;
    const int n = blockIdx.y;
    const uint32_t* __restrict__ tmp = p.tmp + (long long)n * p.ih * p.nw;
    float* __restrict__ outn = p.out + n * p.out_img_stride;
#pragma unroll 2
    for (int it = 0; it < V_ITEMS; ++it) {
        const long long idx = ((long long)blockIdx.x * V_ITEMS + it) * 256 + threadIdx.x;
        if (idx >= total) return;
        const int y = (int)(idx / wq), x0 = (int)(idx - (long long)y * wq) * VEC;
        uint32_t v[VEC][3];
        const int yy = y - p.dy, xx = x0 - p.dx;
        if (yy >= 0 && yy < p.nh && xx >= 0 && xx < p.nw) {
            const int first = min(max(p.bounds[2 * yy], 0), p.ih - 1);
            const int cnt = min(max(p.bounds[2 * yy + 1], 0), p.ks);
            const int32_t* k = p.kk + (long long)yy * p.ks;
            const uint32_t* col = tmp + xx;
            int s[VEC][3];
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e][0] = s[e][1] = s[e][2] = 1 << (PRECISION_BITS - 1);
            for (int j = 0; j < cnt; ++j) {
                const uint32_t* q = col + (long long)min(first + j, p.ih - 1) * p.nw;
                const int w = k[j];
                uint32_t px[VEC];
                if constexpr (VEC == 4) {
                    const uint4 t = *(const uint4*)q;
                    px[0] = t.x; px[1] = t.y; px[2] = t.z; px[3] = t.w;
                } else {
                    px[0] = *q;
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    s[e][0] += (int)(px[e] & 255u) * w;
                    s[e][1] += (int)((px[e] >> 8) & 255u) * w;
                    s[e][2] += (int)((px[e] >> 16) & 255u) * w;
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[e][c] = clip8(s[e][c]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e][0] = v[e][1] = v[e][2] = (uint32_t)p.fill;
        }
        float* o = outn + (long long)y * p.W + x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (VEC == 4) {
                f32x4 r = {lut[c * 256 + v[0][c]], lut[c * 256 + v[1][c]], lut[c * 256 + v[2][c]], lut[c * 256 + v[3][c]]};
                __builtin_nontemporal_store(r, (f32x4*)(o + c * p.out_chan_stride));
            } else {
                o[c * p.out_chan_stride] = lut[c * 256 + v[0][c]];
            }
        }
    }
}

}  // namespace

extern "C" int ccnet_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return ccnet_set_error(CCNET_EINVAL, "resample: sizes must be positive (%d -> %d)", in_size, out_size);
    double scale, fs, support;
    int ks;
    axis_scale(in_size, out_size, &scale, &fs, &support, &ks);
    if (ks > KS_MAX) return ccnet_set_error(CCNET_EINVAL, "resample: %d -> %d needs %d taps (limit %d)", in_size, out_size, ks, KS_MAX);
    return ks;
}

// Resample.c: precompute_coeffs + normalize_coeffs_8bpc for the full box (0, in_size)
extern "C" int ccnet_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize) {
    CC_REQUIRE(bounds && kk, "resample: null table");
    const int need = ccnet_resample_ksize(in_size, out_size);
    if (need < 0) return need;
    CC_REQUIRE(ksize == need, "resample: table has %d taps per pixel, %d -> %d needs %d", ksize, in_size, out_size, need);
    double scale, filterscale, support;
    int ks;
    axis_scale(in_size, out_size, &scale, &filterscale, &support, &ks);
    const double ss = 1.0 / filterscale;
    double k[KS_MAX];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        int32_t* o = kk + (long long)xx * ks;
        for (int x = 0; x < ks; ++x) {
            double w = 0.0;
            if (x < xmax) w = ww != 0.0 ? k[x] / ww : k[x];
            o[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return CCNET_OK;
}

extern "C" size_t ccnet_preprocess_workspace_bytes(int n, int ih, int nw) {
    if (n <= 0 || ih <= 0 || nw <= 0) return 0;
    return (size_t)n * ih * nw * 4;
}

extern "C" int ccnet_preprocess(const uint8_t* src_dev, int n, int ih, int iw, int channels, long long src_image_stride,
                                long long src_row_stride, int src_pixel_stride, float* out_dev, int H, int W,
                                long long out_image_stride, long long out_chan_stride, int dx, int dy, int nw, int nh,
                                const int32_t* hbounds_dev, const int32_t* hk_dev, int hks, const int32_t* vbounds_dev,
                                const int32_t* vk_dev, int vks, const float* lut_dev, int fill, void* workspace_dev,
                                size_t workspace_bytes, ccnet_stream stream) {
    CC_REQUIRE(src_dev && out_dev && hbounds_dev && hk_dev && vbounds_dev && vk_dev && lut_dev && workspace_dev, "preprocess: null pointer");
    CC_REQUIRE(n > 0 && n <= 65535 && ih > 0 && iw > 0 && H > 0 && W > 0, "preprocess: bad sizes (n=%d, %dx%d -> %dx%d)", n, ih, iw, H, W);
    CC_REQUIRE(channels == 1 || channels == 3, "preprocess: %d source channels (1 = grey replicated as convert('RGB') does, or 3)", channels);
    CC_REQUIRE(src_pixel_stride >= channels && src_row_stride >= (long long)iw * src_pixel_stride && src_image_stride >= ih * src_row_stride,
               "preprocess: source strides overlap");
    CC_REQUIRE(nw > 0 && nh > 0 && dx >= 0 && dy >= 0 && dx + nw <= W && dy + nh <= H, "preprocess: paste rectangle %dx%d+%d+%d outside %dx%d", nw, nh, dx,
               dy, W, H);
    CC_REQUIRE(out_chan_stride >= (long long)H * W && out_image_stride >= 3 * out_chan_stride, "preprocess: output strides overlap");
    CC_REQUIRE(hks == ccnet_resample_ksize(iw, nw) && vks == ccnet_resample_ksize(ih, nh), "preprocess: tap counts %d/%d do not belong to %dx%d -> %dx%d",
               hks, vks, ih, iw, nh, nw);
    CC_REQUIRE(fill >= 0 && fill <= 255, "preprocess: fill %d", fill);
    CC_REQUIRE(workspace_bytes >= ccnet_preprocess_workspace_bytes(n, ih, nw) && ((uintptr_t)workspace_dev % 16) == 0, "preprocess: workspace too small or unaligned");
    CC_REQUIRE((double)ih * nw < 2.0e9 && (double)H * W < 2.0e9, "preprocess: image too large");
    HP hp{src_dev, src_image_stride, src_row_stride, src_pixel_stride, channels == 3 ? 1 : 0, ih, iw, nw, hks, hbounds_dev, hk_dev, (uint32_t*)workspace_dev};
    VP vp{(const uint32_t*)workspace_dev, ih, nw, nh, dx, dy, vks, vbounds_dev, vk_dev, lut_dev, fill, out_dev, out_image_stride, out_chan_stride, H, W};
    const bool vec = (W % 4) == 0 && (dx % 4) == 0 && (nw % 4) == 0 && ((uintptr_t)out_dev % 16) == 0 && (out_image_stride % 4) == 0 && (out_chan_stride % 4) == 0;
    hipStream_t st = (hipStream_t)stream;
    // bytes of one source row that H_XC consecutive output columns can touch (Resample.c: taps lie within +-support of the
    // centre), rounded up generously; the kernel clamps to it, so an under-estimate could only misplace reads inside LDS
    const double hscale = (double)iw / nw, hsupport = 2.0 * (hscale < 1.0 ? 1.0 : hscale);
    const long long span_px = (long long)(H_XC * hscale + 2.0 * hsupport) + 4;
    const long long pitch = ((span_px < iw ? span_px : iw) * src_pixel_stride + 15) / 16 * 16;
    const long long lds = (long long)H_XC * hks * 4 + H_RB * pitch;
    const unsigned rb = (unsigned)((ih + H_RB - 1) / H_RB);
    if (lds <= 60 * 1024 && rb <= 65535u && src_pixel_stride <= 16) {
        hipLaunchKernelGGL(resample_h_lds_kernel, dim3((nw + H_XC - 1) / H_XC, rb, n), dim3(256), (size_t)lds, st, hp, (int)pitch);
    } else {
        const unsigned hb = (unsigned)(((long long)ih * nw + 255) / 256);
        hipLaunchKernelGGL(resample_h_kernel, dim3(hb, n), dim3(256), 0, st, hp);
    }
    CC_LAUNCH_CHECK();
    const unsigned vb = (unsigned)(((long long)H * (W / (vec ? 4 : 1)) + 256 * V_ITEMS - 1) / (256 * V_ITEMS));
    if (vec) hipLaunchKernelGGL(resample_v_kernel<4>, dim3(vb, n), dim3(256), 0, st, vp);
    else hipLaunchKernelGGL(resample_v_kernel<1>, dim3(vb, n), dim3(256), 0, st, vp);
    CC_LAUNCH_CHECK();
    return CCNET_OK;
}
