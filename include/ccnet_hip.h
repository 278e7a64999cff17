/* ccnet_hip.h -- C ABI of libccnet_hip.so, the MI355X (gfx950) compute library under the CCNet Python API.
 *
 * The reference (ICtrlCV/CCNet) is pure Python on torch/torchvision and has no FFI of its own; each entry
 * point below names the reference code it stands in for (paths relative to the reference repo root).
 * The Python host (ccnet_amd/) binds these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev / inside a ccnet_slice is DEVICE memory owned by the
 *     caller (PyTorch's allocator in our host); the library allocates no device memory of its own;
 *   - every launch goes to the caller's stream (hipStream_t passed as void*), nothing synchronises
 *     unless its comment says so;
 *   - return 0 on success, a negative CCNET_E* code on failure; ccnet_last_error() gives the thread-local
 *     message.  Nothing throws or aborts.  Arguments are validated on the host BEFORE any launch.
 *   - activations are NHWC; a ccnet_slice is a channel range of a [B][H][W][ldc] buffer, so torch.cat on
 *     channels never copies (producers write their slice of the consumer's buffer).
 */
#ifndef CCNET_HIP_H
#define CCNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ccnet_stream; /* hipStream_t */

enum { CCNET_F32 = 0, CCNET_BF16 = 1 };
/* network/model_block.py:14-34 get_activation_function */
enum { CCNET_ACT_NONE = 0, CCNET_ACT_SILU = 1, CCNET_ACT_RELU = 2, CCNET_ACT_RELU6 = 3, CCNET_ACT_LRELU = 4 };
enum {
    CCNET_OK = 0,
    CCNET_EINVAL = -1,  /* bad argument / shape the kernels do not support */
    CCNET_EHIP = -2,    /* a HIP runtime call failed */
    CCNET_ENOGPU = -3
};

typedef struct {
    void* ptr;     /* element (b=0, y=0, x=0, channel 0) of the WHOLE buffer */
    int32_t ldc;   /* channels of the whole buffer = pixel stride in elements */
    int32_t coff;  /* first channel of this slice */
    int32_t C;     /* channels in this slice */
    int32_t shift; /* nearest-neighbour upsampling by 2^shift: logical pixel (y,x) is stored pixel (y>>shift, x>>shift)
                      (nn.Upsample(scale_factor=2, mode="nearest"), network/net.py:103; F.interpolate x2/x4,
                      network/model_block.py:385-392).  Only read operands may have shift != 0. */
} ccnet_slice;

const char* ccnet_last_error(void);
int ccnet_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int ccnet_device_count(void);

/* ---- Conv-BN-act as BN-folded NHWC implicit GEMM on MFMA ------------------------------------------------
 * Replaces BaseConv.forward (network/model_block.py:57-58), FocusReplaceConv/ASFFConv.forward (:186-187,
 * :318-319), the SpaceToDepth slice+cat+conv (:166-175; expressed as its exact 6x6 stride-2 pad-2 form),
 * the bare nn.Conv2d/nn.Linear calls of MobileViT / nn.Transformer / the head preds (:265-277,
 * network/net.py:178-192), and torch.cat / nn.Upsample on its input side (network/net.py:208-210).
 *   out[b,oy,ox,n] = act( bias[n] + sum_{ky,kx,c} in[b, oy*stride-pad+ky, ox*stride-pad+kx, c] * W[n][ky][kx][c] ) (+ residual)
 * in = channel concat of in[0..nseg-1]; zero padding outside the image.
 * weight: [Cout_pad][Kpad] row-major in `dtype`, K index = (ky*kw + kx)*Ctot + c, zero padded; pad sizes from
 * ccnet_conv_pack_dims.  bias: fp32[Cout] or NULL.  residual (ptr may be NULL) is added AFTER the activation
 * (Bottleneck.forward, model_block.py:213-218) and may alias `out`. */
typedef struct {
    int32_t dtype;     /* storage type of in / weight / residual: CCNET_F32 or CCNET_BF16; accumulation is fp32 */
    int32_t out_dtype; /* CCNET_F32 or CCNET_BF16 (fp32 out from bf16 in is used for the head logits) */
    int32_t B, H, W;   /* logical input size (after `shift`) */
    int32_t kh, kw, stride, pad;
    int32_t Ho, Wo, Cout;
    int32_t nseg;      /* 1 or 2 */
    ccnet_slice in[2];
    const void* weight;
    int32_t Kpad, Cout_pad;
    const float* bias;
    int32_t act;
    ccnet_slice residual;
    ccnet_slice out;
} ccnet_conv_desc;

int ccnet_conv_pack_dims(int dtype, int Cout, int K, int* Cout_pad, int* Kpad);
int ccnet_conv2d(const ccnet_conv_desc* d, ccnet_stream stream);

/* ---- 3x3 Conv-BN-act with register-streamed weights (bf16) -------------------------------------------------
 * The same operator as ccnet_conv2d for the two shapes that carry most of the network's FLOPs -- BaseConv k3 s1 p1
 * (network/model_block.py:37-58: Bottleneck.conv2, the head's cls/reg convs, MobileViT.conv4) and SpaceToDepth + BaseConv k3
 * (:152-175; s2d = 1: `in` is the un-shuffled (B, H, W, C) tensor, the conv runs over the virtual (H/2, W/2, 4C) one with
 * channel (2dy+dx)*C + c = pixel (2y+dy, 2x+dx), channel c) -- on a different kernel: the halo patch of the input in LDS,
 * the weights streamed from L2 into registers in consumption order (conv_halo.hip).
 *   in: 1-2 bf16 slices (channel concat; s2d: exactly one), un-shifted, each a multiple of 32 channels; H, W: stored input size;
 *   Cout % 64 == 0; out / residual (ptr may be NULL; added after the activation) bf16 with channel offsets % 4 == 0.
 *   wstream_dev: bf16, ccnet_conv_halo_stream_elems(Cin_virtual, Cout) elements: for each block of 64 output channels, for each
 *   32-channel chunk, for each tap (ky*3+kx), for each 16-row n-tile: one MFMA A-operand fragment [64 lanes][8] (lane l: row
 *   l & 15, k (l >> 4)*8..+7), then a copy of the block's first 9 fragments (ccnet_amd/ops.py:PackedHalo).  bias_dev: fp32[Cout] or NULL.
 * s2d = 2: BaseConv k3 s2 p1 (the downsampling convs of net_deploy / net_small_target, network/net_deploy.py:44-60) on the same
 *   kernel: `in` is the (B, H, W, C) tensor (one slice, H and W even), the output is (H/2, W/2); wstream_dev has
 *   ccnet_conv_halo_s2_stream_elems(C, Cout) elements: per block of 64 output channels, five passes over the 32-channel chunks
 *   of C, each pass two taps (ky, kx) of four n-tile fragments per chunk -- pass 0: (1,1) and four zero fragments; 1: (1,0),
 *   (1,2); 2: (0,1), (2,1); 3: (0,0), (0,2); 4: (2,0), (2,2) -- then a copy of the block's first 8 fragments
 *   (ccnet_amd/ops.py:PackedHaloS2).
 * Anything else: CCNET_EINVAL (callers use ccnet_conv2d).  Rounding points are ccnet_conv2d's; the accumulation order differs. */
size_t ccnet_conv_halo_stream_elems(int Cin_virtual, int Cout);
size_t ccnet_conv_halo_s2_stream_elems(int C, int Cout);
int ccnet_conv_halo(const ccnet_slice* in, int nseg, int s2d, int B, int H, int W, int Cout, const void* wstream_dev,
                    const float* bias_dev, int act, ccnet_slice residual, ccnet_slice out, ccnet_stream stream);

/* ---- 1x1 Conv-BN-act with register-streamed weights (bf16) ---------------------------------------------------
 * The same operator as ccnet_conv2d for BaseConv k1 s1 (network/model_block.py:37-58) over a channel concat of one or two
 * inputs (torch.cat, network/net.py:208-210), each possibly nearest-upsampled (slice.shift, nn.Upsample network/net.py:103),
 * on a kernel built like ccnet_conv_halo: pixels staged in LDS, weights streamed from L2 into registers (conv_pw.hip).
 *   in: 1-2 bf16 slices, each a multiple of 64 channels, shift <= 2; H, W: logical (upsampled) size = output size;
 *   Cout % 64 == 0; out / residual (ptr may be NULL; added after the activation) bf16 with channel offsets % 4 == 0.
 *   wstream_dev: bf16, ccnet_conv_pw_stream_elems(Cin, Cout) elements: for each block of 64 output channels, for each 32-channel
 *   k-step, for each 16-row n-tile: one MFMA A-operand fragment [64 lanes][8], then a copy of the block's first 8 fragments
 *   (ccnet_amd/ops.py:PackedPw).  bias_dev: fp32[Cout] or NULL.
 * Anything else: CCNET_EINVAL (callers use ccnet_conv2d).  Results are bit-identical to ccnet_conv2d's (same k order). */
size_t ccnet_conv_pw_stream_elems(int Cin, int Cout);
int ccnet_conv_pw(const ccnet_slice* in, int nseg, int B, int H, int W, int Cout, const void* wstream_dev, const float* bias_dev,
                  int act, ccnet_slice residual, ccnet_slice out, ccnet_stream stream);

/* ---- one Bottleneck in one launch (bf16) ----------------------------------------------------------------------
 * Replaces Bottleneck.forward as CSPLayer builds it (network/model_block.py:193-218 inside :86-116, expansion 1.0):
 *   y = [x +] conv2(conv1(x)),  conv1 = BaseConv k1 (C -> C), conv2 = BaseConv k3 s1 p1 (C -> C), BN folded, SiLU, C = 64, 128 or 256
 *   (256: maps up to 21 wide).
 * The hidden map stays in LDS (ccnet_amd/csrc/bneck_fused.hip).  `in`: one un-shifted bf16 slice of C channels; `out`: C channels,
 * must NOT overlap `in` (a tile's halo is its neighbours' output pixels) -- CCNET_EINVAL otherwise.  shortcut != 0 adds x after
 * the activation.  wstream_dev: bf16, ccnet_bneck_fused_stream_elems(C) elements: for each block of 64 output channels of conv2,
 * the C/32 x C/16 MFMA A-operand fragments of conv1 (k-step major, then 16-row n-tile; ALL output channels), then conv2's
 * fragments in ccnet_conv_halo's order (32-channel chunk, tap, n-tile), then a copy of the block's first 8 (C = 64) or 16 (C >= 128)
 * fragments (ccnet_amd/ops.py:PackedBneck).  bias1_dev / bias2_dev: fp32[C].  Results are bit-identical to ccnet_conv2d (conv1) followed
 * by ccnet_conv_halo (conv2).  Anything else: CCNET_EINVAL (callers use the two launches). */
size_t ccnet_bneck_fused_stream_elems(int C);
int ccnet_bneck_fused(ccnet_slice in, int B, int H, int W, const void* wstream_dev, const float* bias1_dev, const float* bias2_dev,
                      int shortcut, int act, ccnet_slice out, ccnet_stream stream);

/* ---- a whole CSPLayer with one bottleneck in one launch (bf16) ---------------------------------------------
 * Replaces CSPLayer.forward with n = 1 (network/model_block.py:86-116) and the Bottleneck inside it (:193-218):
 *   x_1 = conv1(x); x_2 = conv2(x); x_1 = [x_1 +] conv_b2(conv_b1(x_1)); out = conv3(cat(x_1, x_2))
 * every conv a BaseConv (BN folded, SiLU); x = channel concat of in[0..nseg-1] (each a multiple of 64 channels, may carry
 * an upsampling shift).  hidden = 32 or 64; out has 2*hidden channels; H % 8 == 0 and W % 16 == 0.  Intermediates are
 * rounded to bf16 where the separate launches round them.  weights_dev: bf16 MFMA fragments of
 * [conv1;conv2] (2h x Cin) | conv_b1 (h x h) | conv_b2 (h x 9h, K = (ky*3+kx)*h + c) | conv3 (2h x 2h), each
 * [n-tile][k-step][64 lanes][8] (ccnet_amd/ops.py:pack_csp), ccnet_csp_fused_weight_elems(cin, hidden) elements;
 * bias_dev: fp32 [2h | h | h | 2h].  Anything else: CCNET_EINVAL (callers use ccnet_conv2d per layer). */
size_t ccnet_csp_fused_weight_elems(int cin, int hidden);
int ccnet_csp_fused(const ccnet_slice* in, int nseg, int B, int H, int W, int hidden, int shortcut, int act,
                    const void* weights_dev, const float* bias_dev, ccnet_slice out, ccnet_stream stream);

/* ---- network stem: 6x6 stride-2 pad-2 conv on the NCHW fp32 image, folded BN, activation ------------------
 * Replaces FocusReplaceConv (network/net_deploy.py:41) and SpaceToDepth(3,..)+BaseConv k3 (network/net.py:40)
 * in its 6x6 form.  x_dev: fp32 [B][3][H][W] (the tensor the reference's callers build, tools/demo.py:56-65).
 * weight: fp32 [Cout][160] with k = (c*6 + ky)*8 + kx (kx 6,7 and rows 18,19 zero).  Cout must be <= 80 and % 8 == 0. */
int ccnet_stem_conv(const float* x_dev, int B, int H, int W, const float* weight, const float* bias, int Cout,
                    int act, ccnet_slice out, int out_dtype, ccnet_stream stream);

/* depthwise 3x3 stride-1 pad-1 with bias, no activation: MobileViT.conv1 (network/model_block.py:265-266).
 * weight fp32 [9][C] (tap-major), bias fp32 [C]. */
int ccnet_dwconv3x3(ccnet_slice in, int dtype, int B, int H, int W, const float* weight, const float* bias,
                    ccnet_slice out, ccnet_stream stream);

/* ---- pooling ---------------------------------------------------------------------------------------------
 * SPPFBottleneck's three cascaded MaxPool2d(5,1,2) (network/model_block.py:75-81) == windows 5/9/13 of x:
 * reads channels [coff, coff+C) of buf and writes y1,y2,y3 to [coff+C, coff+4C) of the same buffer. */
int ccnet_sppf_pool(ccnet_slice buf, int dtype, int B, int H, int W, ccnet_stream stream);
/* F.max_pool2d(x, 3, stride=2, padding=1) (network/model_block.py:381) */
int ccnet_maxpool3x3s2(ccnet_slice in, int dtype, int B, int H, int W, ccnet_slice out, ccnet_stream stream);

/* ---- attention gates -------------------------------------------------------------------------------------
 * CBAM.forward (network/attention_block.py:19-22): x*ca then *sa.  fc1: fp32 [C/r][C], fc2: fp32 [C][C/r],
 * w7: fp32 [2][k][k] (avg plane first), k = 7 or 3.  scratch_dev: fp32, 16-byte aligned, ccnet_gate_scratch_floats(B,H,W,C)
 * floats; calls sharing a scratch must not overlap. */
size_t ccnet_gate_scratch_floats(int B, int H, int W, int C);
int ccnet_cbam(ccnet_slice in, int dtype, int B, int H, int W, const float* fc1, const float* fc2, int Cr,
               const float* w7, int k7, float* scratch_dev, ccnet_slice out, ccnet_stream stream);
/* ECA.forward (network/attention_block.py:95-102): global avg -> Conv1d(1,1,3,pad 1) over channels -> sigmoid -> scale */
int ccnet_eca(ccnet_slice in, int dtype, int B, int H, int W, const float* w3, float* scratch_dev, ccnet_slice out,
              ccnet_stream stream);
/* ASFF fusion (network/model_block.py:401-405): softmax over the 3 weight logits (fp32/bf16 [B][H][W][ldc], 3 ch)
 * and the weighted sum of the three resized levels (each may carry an upsampling shift). */
int ccnet_asff_fuse(ccnet_slice l0, ccnet_slice l1, ccnet_slice l2, ccnet_slice logits, int dtype, int B, int H, int W,
                    ccnet_slice out, ccnet_stream stream);

/* ---- Transformer pieces (torch.nn.Transformer as used by MobileViT, network/model_block.py:270-271,290) ----
 * out = LayerNorm(x (+ res)) * gamma + beta over the last dim E; rows of E elements, eps as nn.LayerNorm (1e-5). */
int ccnet_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* out, int dtype,
                    int64_t rows, int E, float eps, ccnet_stream stream);
/* scaled-dot-product attention over axis S of (S, N, heads*hd) operands (batch_first=False: S is the image batch).
 * q/k/v are channel slices of [S*N][ld] row buffers: q rows are (s*N + n); Sq queries attend Sk keys.  C = heads*hd. */
int ccnet_attention(ccnet_slice q, ccnet_slice k, ccnet_slice v, int dtype, int Sq, int Sk, int N, int heads, int hd,
                    ccnet_slice out, ccnet_stream stream);

/* The whole torch.nn.Transformer(src, src) of MobileViT (network/model_block.py:270-271, 290; torch's
 * nn/modules/transformer.py encoder/decoder stacks, post-norm, ReLU, eval mode) in one launch, bf16 only.
 * With batch_first=False the sequence axis is the image batch, so the stack is N independent problems of S tokens:
 * one workgroup per spatial position keeps its tokens in LDS through all layers and streams the weights.
 * src/out: rows (s*N + n) of d_model bf16, row strides src_ld/out_ld elements.  Built for d_model 256, 2 heads,
 * dim_ff 1024, 1 <= S <= 64 (anything else: CCNET_EINVAL; callers fall back to the per-op entry points above).
 * wstream: bf16 MFMA fragments in per-wave consumption order, [8][frags_per_wave + 16][64][8]; params: fp32 biases and
 * LayerNorm gamma/beta in program order (layout: ccnet_amd/ops.py:pack_transformer, transformer_fused.hip).
 * ccnet_transformer_fused_frags reports the stream lengths for a given depth. */
int ccnet_transformer_fused_frags(int n_enc, int n_dec, long long* frags_per_wave, long long* param_floats);
int ccnet_transformer_fused(const void* src, int src_ld, void* out, int out_ld, int S, int N, int d_model, int nhead,
                            int dim_ff, const void* wstream, long long frags_per_wave, const float* params, int n_enc,
                            int n_dec, ccnet_stream stream);

/* ---- decode + post-process --------------------------------------------------------------------------------
 * decode_outputs (utils/bbox.py:13-79).  heads: nlev raw fp32 maps, element (b,c,y,x) of level l at
 * head_dev[l][b*sb[l] + c*sc[l] + (y*w[l]+x)*sp[l]].  out_dev: fp32 [B][A][5+nc] contiguous.
 * stride of a level = in_h / h[l] for BOTH axes (bbox.py:57); x,w are divided by in_w and y,h by in_h. */
int ccnet_decode(const float* const* head_dev, const int64_t* sb, const int64_t* sc, const int64_t* sp, const int32_t* h,
                 const int32_t* w, int nlev, int B, int nc, float in_h, float in_w, float* out_dev, ccnet_stream stream);

/* The same, and per anchor the value post_process compares with its threshold: score_dev fp32 [B][A] = obj * max class
 * probability (the fp32 product of utils/bbox.py:104-107).  Needs 5 + nc <= 64.  score_dev NULL = ccnet_decode. */
int ccnet_decode_scored(const float* const* head_dev, const int64_t* sb, const int64_t* sc, const int64_t* sp,
                        const int32_t* h, const int32_t* w, int nlev, int B, int nc, float in_h, float in_w, float* out_dev,
                        float* score_dev, ccnet_stream stream);

/* post_process part 1 (utils/bbox.py:98-130): cxcywh->xyxy, class max (first index on ties), keep
 * obj*cls >= conf_thres (fp32 compare), order-preserving compaction.
 * cand_dev: fp32 [B][A][7] rows [x1,y1,x2,y2,obj,cls_conf,cls_id]; counts_dev: int32 [B]. */
int ccnet_filter(const float* pred_dev, int B, int A, int nc, float conf_thres, float* cand_dev, int32_t* counts_dev,
                 ccnet_stream stream);
/* The same with a small workspace (ccnet_filter_workspace_ints(B, A) int32; 0 for A <= 32768): maps with more anchors than
 * one workgroup's super-chunk (a 1280x1280 input has 136 000) then run one workgroup per (image, 32768 anchors) after a
 * counting pass instead of one workgroup per image; results are identical.  workspace_dev may be NULL (= ccnet_filter). */
size_t ccnet_filter_workspace_ints(int B, int A);
int ccnet_filter_ws(const float* pred_dev, int B, int A, int nc, float conf_thres, float* cand_dev, int32_t* counts_dev,
                    int32_t* workspace_dev, ccnet_stream stream);
/* ccnet_filter_ws whose keep test reads score_dev (from ccnet_decode_scored on the same pred_dev: 4 bytes per anchor instead of
 * the row); rows are read for kept anchors only.  Results are identical.  score_dev NULL = ccnet_filter_ws. */
int ccnet_filter_scored(const float* pred_dev, const float* score_dev, int B, int A, int nc, float conf_thres, float* cand_dev,
                        int32_t* counts_dev, int32_t* workspace_dev, ccnet_stream stream);
/* post_process part 2 (utils/bbox.py:133-158): torchvision.ops.boxes.batched_nms (0.9.0 coordinate trick; third party,
 * see DESIGN.md) + gather + rescale to image pixels.  Above 256 rows per image the work is partitioned by class group
 * (classes whose offset boxes cannot meet), which leaves every bit of the global algorithm's result unchanged.
 * nmax = rows provided per image: results are exact when
 * nmax >= max(counts); an image with more candidates is processed on its first nmax only (never overrun), so a caller
 * may launch with an estimate and look at counts afterwards.  image_hw_dev: int32 [B][2] = (ih, iw).
 * det_dev: fp32 [B][nmax][7] rows [top,left,bottom,right,obj,cls_conf,cls_id] in descending score order;
 * keep_idx_dev: int32 [B][nmax] indices into the image's candidate list; nkeep_dev: int32 [B].
 * workspace_dev: ccnet_nms_workspace_bytes(B, nmax) bytes. */
size_t ccnet_nms_workspace_bytes(int B, int nmax);
int ccnet_nms(const float* cand_dev, const int32_t* counts_dev, int B, int A, int nmax, double nms_thres,
              const int32_t* image_hw_dev, float* det_dev, int32_t* keep_idx_dev, int32_t* nkeep_dev, void* workspace_dev,
              ccnet_stream stream);

/* ccnet_nms_noalloc: ccnet_nms without a caller workspace.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory at the call site -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  The workspace (offset boxes, sorted boxes, sort keys, group tables, pair bitmask panel) is a file-scope device array
 * of ccnet_nms_noalloc_max_bytes() = 32 MiB inside the library; each GPU the library runs on holds its own copy until the
 * process ends.  Same arguments, results and kernels as ccnet_nms; a (B, nmax) whose ccnet_nms_workspace_bytes exceeds the
 * maximum is rejected with CCNET_EINVAL (use ccnet_nms then).  The call allocates nothing (no hipMalloc / hipFree / allocating
 * library routine); the array's per-device address is resolved on the first call on a device, or ahead of time by
 * ccnet_nms_noalloc_init() with that device current (do that before a graph capture).  Every call on one GPU shares that
 * GPU's array: calls that overlap in time on one GPU (two streams, two threads) race -- serialise them or use ccnet_nms. */
size_t ccnet_nms_noalloc_max_bytes(void);
int ccnet_nms_noalloc_init(void);
int ccnet_nms_noalloc(const float* cand_dev, const int32_t* counts_dev, int B, int A, int nmax, double nms_thres,
                      const int32_t* image_hw_dev, float* det_dev, int32_t* keep_idx_dev, int32_t* nkeep_dev,
                      ccnet_stream stream);

/* ---- image preparation in front of the path ----------------------------------------------
 * Replaces tools/eval.py:134-140 (`image.resize((W, H), Image.BICUBIC)`, float32, /255, mean/std, HWC->CHW) and the
 * letterbox of utils/datasets.py:123-134 (+ :73-77).  The resize is Pillow's ImagingResample for 8-bit images (third
 * party; src/libImaging/Resample.c): integer arithmetic, results are bit-exact.  The cv2.INTER_CUBIC flavour of
 * tools/demo.py:53 is a different filter and is not provided.
 *
 * ccnet_resample_ksize: taps per output pixel of one axis (negative CCNET_E* on bad sizes).
 * ccnet_resample_coeffs (host only, no GPU needed): Resample.c's precompute_coeffs + normalize_coeffs_8bpc for the full
 * box; bounds[out_size][2] = (first source index, tap count), kk[out_size][ksize] 22-bit fixed-point weights. */
int ccnet_resample_ksize(int in_size, int out_size);
int ccnet_resample_coeffs(int in_size, int out_size, int32_t* bounds_host, int32_t* kk_host, int ksize);
/* ccnet_preprocess: n images of one size.  src_dev: uint8, pixel (i,y,x) channel c at
 * src_dev[i*src_image_stride + y*src_row_stride + x*src_pixel_stride + c], channels 3 (RGB) or 1 (grey, replicated like
 * convert("RGB")).  Each image is resized to nw x nh, pasted at (dx, dy) on an H x W canvas of value `fill` (the plain
 * resize of eval.py is dx = dy = 0, nw = W, nh = H), every 8-bit value v of channel c becomes lut_dev[c*256 + v] (the host
 * fills the table with the reference's numpy arithmetic) and is written to
 * out_dev[i*out_image_stride + c*out_chan_stride + y*W + x] (fp32).  The hbounds/hk and vbounds/vk tables are device copies of
 * ccnet_resample_coeffs(iw, nw) and (ih, nh).  workspace_dev: ccnet_preprocess_workspace_bytes(n, ih, nw) bytes, 16-byte
 * aligned (the 8-bit intermediate image between the two passes). */
size_t ccnet_preprocess_workspace_bytes(int n, int ih, int nw);
int ccnet_preprocess(const uint8_t* src_dev, int n, int ih, int iw, int channels, long long src_image_stride,
                     long long src_row_stride, int src_pixel_stride, float* out_dev, int H, int W,
                     long long out_image_stride, long long out_chan_stride, int dx, int dy, int nw, int nh,
                     const int32_t* hbounds_dev, const int32_t* hk_dev, int hks, const int32_t* vbounds_dev,
                     const int32_t* vk_dev, int vks, const float* lut_dev, int fill, void* workspace_dev,
                     size_t workspace_bytes, ccnet_stream stream);

/* ---- plan: a recorded sequence of the calls above, replayed with one C call per forward --------------------
 * Replaces the Python-level op-by-op dispatch of Net.forward (network/net.py:194-275 and siblings).
 * Recording: ccnet_plan_begin(&plan) makes the CALLING THREAD record instead of launch until ccnet_plan_end();
 * every ccnet_* compute call in between is validated and appended.  A plan is immutable afterwards; one plan is run
 * by one thread at a time, distinct plans are independent (no global mutable state besides the thread-local recorder). */
typedef struct ccnet_plan ccnet_plan;
int ccnet_plan_begin(ccnet_plan** plan);
int ccnet_plan_end(void);
int ccnet_plan_num_ops(const ccnet_plan* plan);
/* Concurrency inside a plan (recording thread only).  Ops are recorded on the current lane (0..3, default 0 = the stream
 * given to ccnet_plan_run; lanes 1..3 are side streams owned by the plan).  ccnet_plan_wait(dst, src) makes everything
 * recorded later on lane dst wait for everything recorded so far on lane src.  The recorder must join every side lane back
 * into lane 0 before ccnet_plan_end().  Used for the independent head levels / PAN branches (network/net.py:225-275). */
int ccnet_plan_set_lane(int lane);
int ccnet_plan_wait(int dst_lane, int src_lane);
/* x_dev, if not NULL, replaces the image pointer of every recorded ccnet_stem_conv.  Lane 0 ops go to `stream`, side-lane ops
 * to streams owned by the plan, joined back into `stream` before the call returns (nothing synchronises the host).
 * Environment: CCNET_LANES=0 runs everything on `stream`; CCNET_GRAPH=1 replays the plan as a HIP graph from the second
 * call with the same x_dev.  One thread at a time per plan (the library serialises concurrent callers of one plan). */
int ccnet_plan_run(const ccnet_plan* plan, const float* x_dev, ccnet_stream stream);
void ccnet_plan_destroy(ccnet_plan* plan);
/* Measurement aids (bench.py): what op i is and its algorithmic work (flops = 2*MAC; bytes = every distinct input and
 * output element once), and a run that brackets every op with HIP events ON `stream` and synchronises it. */
enum {
    CCNET_OP_OTHER = 0, CCNET_OP_CONV = 1, CCNET_OP_STEM = 2, CCNET_OP_DWCONV = 3, CCNET_OP_SPPF = 4, CCNET_OP_MAXPOOL = 5,
    CCNET_OP_CBAM = 6, CCNET_OP_ECA = 7, CCNET_OP_ASFF = 8, CCNET_OP_LAYERNORM = 9, CCNET_OP_ATTENTION = 10,
    CCNET_OP_DECODE = 11, CCNET_OP_FILTER = 12, CCNET_OP_NMS = 13, CCNET_OP_TRANSFORMER = 14,
    CCNET_OP_CSP = 15 /* a whole CSPLayer in one launch (ccnet_csp_fused) */
};
int ccnet_plan_op_info(const ccnet_plan* plan, int i, int* kind, double* flops, double* bytes);
int ccnet_plan_run_timed(const ccnet_plan* plan, const float* x_dev, ccnet_stream stream, float* ms_per_op);

#ifdef __cplusplus
}
#endif
#endif /* CCNET_HIP_H */
