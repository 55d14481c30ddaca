/* cf_hip.h - C ABI of libcfhip.so: the MI355X (gfx950) kernels of the CenterFusion forward path.
 *
 * The reference (HengWeiBin/CenterFusionDetect3D) is pure Python on PyTorch; its "FFI" for this
 * path is the set of ATen / torchvision operators its modules dispatch to.  Each entry point below
 * names the reference interface it replaces (paths relative to /root/reference/src/lib).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are asynchronous;
 *   - no hidden allocation, no host synchronisation, re-entrant per stream (graph-capturable);
 *   - return value: 0 on success, negative CF_E* code on failure (never throws across the ABI);
 *     cf_last_error() returns a static host string describing the last failure of this thread.
 *   - activations between kernels are NHWC fp32 ("channels-last"); the module boundary tensors
 *     (images in, head maps out) are NCHW fp32 as in the reference.
 */
#ifndef CF_HIP_H
#define CF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 6 /* 2: cf_dcn_args.mask_activated, cf_nchw_to_nhwc, cf_spin_us; 3: cf_conv3x3_root_f16x3, stride 2 in cf_conv3x3_f16x3; 4: cf_head_fused_args.mx / first_scale, cf_pack_feat_mx, cf_dcn_args.out_mx; 5: cf_conv3x3_proj_f16x3, cf_stem_args.out_pool, cf_pack_conv_f16x3, cf_pack_dcn_f16; 6: in_scale (per-layer activation pre-scale of the f16x3 kernels) in cf_conv_args / cf_dcn_args / cf_stem_args, cf_dcn_args.mx_scale, cf_pack_feat_mx_scaled, cf_absmax_f32, cf_checksum64, cf_topk_peaks_if_changed, cf_topk_frustum */

#define CF_OK 0
#define CF_EINVAL (-22)
#define CF_ELAUNCH (-5)

#define CF_MAX_SRC 4

/* activation / epilogue modes of cf_conv2d_fused and cf_dcn_v2_fused */
#define CF_ACT_NONE 0
#define CF_ACT_RELU 1
#define CF_ACT_SIGMOID_CLAMP 2 /* clamp(sigmoid(x), 1e-4, 1-1e-4)  (networks/detectHeads.py:21-23) */
#define CF_ACT_RAW_AND_SIGDEPTH 3 /* out = x, out2 = 1/(sigmoid(x)+1e-6)-1 (model/utils.py:131-141) */

#define CF_LAYOUT_NHWC 0
#define CF_LAYOUT_NCHW 1
#define CF_LAYOUT_NHWC_SPLIT_BF16 2 /* per pixel [C hi][C lo] bf16, x = hi + lo (cf_conv2d_bf16x3) */

/* One 16-byte K-slot of the implicit GEMM: 4 consecutive channels of one source at one tap. */
typedef struct cf_slot {
  int32_t src;   /* index into cf_conv_args.src (uniform inside a 32-wide K chunk); -1: zero chunk */
  int32_t dy;    /* tap row offset relative to ho*stride (already includes -pad)                    */
  int32_t dx;    /* tap col offset relative to wo*stride                                            */
  int32_t c_off; /* first channel inside the source row; <0: zero padding slot                      */
} cf_slot;

/* cf_conv2d_fused: implicit-GEMM convolution on the fp32 MFMA pipe with a fused epilogue
 *   out = act( conv(cat(src...), W) + bias [+ residual] )
 * replaces aten::conv2d + aten::batch_norm(eval, folded into W/bias) + add_ + relu_ of
 *   model/networks/dla.py:21-41 (Root: the channel concat is never materialised - up to 4 sources),
 *   dla.py:124-161 (BasicBlock), dla.py:177-192,250-269 (stem/level0/level1), dla.py:98-103 (project),
 *   dla.py:426-433 (conv_offset_mask), model/networks/detectHeads.py:59-98 (head convs) and
 *   model/networks/fusionModules.py:18-35 (ConcateCombiner: feat || pc_hm as two sources).
 * GEMM view: M = B*Ho*Wo, N = n_out, K = 4 * n_slots, weights pre-packed [N_pad][K_pad]
 * (K contiguous, BN folded) in slot order by the host (centerfusiondetect3d_amd/packing.py). */
typedef struct cf_conv_args {
  const float* src[CF_MAX_SRC]; /* NHWC sources sharing B,H,W                                        */
  int32_t src_c[CF_MAX_SRC];    /* channel stride (floats per pixel) of each source                  */
  int32_t n_src;
  int32_t B, H, W;              /* input geometry                                                     */
  int32_t Ho, Wo;               /* output geometry                                                    */
  int32_t stride;               /* spatial stride                                                     */
  const float* weight;          /* [N_pad][K_pad]                                                     */
  const cf_slot* slots;         /* [K_pad/4]                                                          */
  const float* bias;            /* [N_pad]                                                            */
  int32_t K_pad;                /* multiple of 32                                                     */
  int32_t N;                    /* real output channels                                               */
  int32_t N_pad;                /* multiple of 32, >= N                                               */
  const float* residual;        /* optional NHWC [M][res_stride], added before the activation         */
  int32_t res_stride;
  float* out;                   /* NHWC: [M][out_stride] ; NCHW: [B][N][Ho*Wo]                        */
  float* out2;                  /* second output for CF_ACT_RAW_AND_SIGDEPTH (same layout) or NULL    */
  int32_t out_stride;           /* floats per pixel of the NHWC destination (>= N)                    */
  int32_t out_layout;           /* CF_LAYOUT_*                                                        */
  int32_t act;                  /* CF_ACT_*                                                           */
  int32_t precise;              /* !=0: two-level (per-32-K-chunk) fp32 summation, see cf_gemm.hip    */
  float out_scale;              /* f16x3 kernels only: 2^-s / in_scale, s = weight scale exponent (2^-(s+4) at the
                                   default in_scale)                                                   */
  float in_scale;               /* (ABI 6) f16x3 kernels only: the power of two every source value (and, in the fused
                                   Root / projection forms, every operand of that GEMM) is multiplied by before
                                   the split into fp16 hi + lo.  0 = the default 16.  |x| * in_scale must stay
                                   below 65504: larger values are CLAMPED (see "Dynamic range" below)  */
} cf_conv_args;
int cf_conv2d_fused(const cf_conv_args* a, void* stream);

/* cf_stem_fused: the DLA-34 stem in one launch - base_layer 7x7 (C -> 16), level0 3x3 (16 -> 16) and
 * level1 3x3 stride 2 (16 -> 32), each followed by its folded BatchNorm and ReLU
 * (model/networks/dla.py:237-262: DLA.base_layer, level0, level1) - straight from the NCHW image batch
 * to the half-resolution fp32 NHWC level1 map; the full-resolution intermediates stay in LDS
 * (cf_stem.hip).  f16x3 arithmetic (fp32-level accuracy).  Weights: MFMA 16x16x32 fragment order as
 * produced by packing.pack_stem; scale_* = 2^-(s+4) of the layer. */
typedef struct cf_stem_args {
  const float* x;                 /* images, fp32 NCHW (B, C, H, W), C <= 3                    */
  int32_t B, C, H, W;             /* H, W even                                                 */
  const void* w_base;   const float* b_base;   float scale_base;
  const void* w_level0; const float* b_level0; float scale_level0;
  const void* w_level1; const float* b_level1; float scale_level1;
  float* out;                     /* fp32 NHWC (B, H/2, W/2, 32)                               */
  float* out_pool;                /* (ABI 5) optional: MaxPool2d(2, 2) of `out`, fp32 NHWC (B, H/4, W/4, 32) - the level-2
                                     Tree's downsample (dla.py:96), written from the same registers; NULL = not written */
  float in_scale[3];              /* (ABI 6) activation pre-scale (power of two, 0 = 16) of the image, of base_layer's output
                                     and of level0's output; scale_* = 2^-s / in_scale[i] of the layer that reads it */
} cf_stem_args;
int cf_stem_fused(const cf_stem_args* a, void* stream);

/* cf_conv2d_bf16x3: the same implicit GEMM on the bf16 MFMA pipe with split operands
 * (x = hi + lo, both bf16; a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulate; <= ~2^-17
 * relative error per product at 5.3x the fp32-MFMA rate).  Used for the head convolutions
 * (model/networks/detectHeads.py:59-98, 165-191), which are not followed by the error-amplifying
 * DCN neck.  Same argument block as cf_conv2d_fused, read as follows: src[] are split-bf16 NHWC
 * tensors (src_c = channels per plane, multiple of 8); one slot = 8 channels; weight is
 * [N_pad][2][K_pad] bf16 (hi plane, lo plane); out is either CF_LAYOUT_NHWC_SPLIT_BF16
 * (out_stride = channels per plane) or CF_LAYOUT_NCHW fp32; residual / precise are ignored. */
int cf_conv2d_bf16x3(const cf_conv_args* a, void* stream);
/* (ABI 6) LEGACY: the unfused heads of rounds 1-2.  Compiled only with -DCF_LEGACY_HEADS; the default library keeps the export and
 * answers it with CF_EINVAL ("legacy kernel path") - every bf16x3 layer of the path runs inside cf_head_fused. */

/* cf_conv2d_f16x3: cf_conv2d_fused semantics (fp32 NHWC sources / residual / output, bias, ReLU) with
 * the products evaluated on the f16 MFMA pipe from split operands (x = hi + lo fp16 after a
 * power-of-two scale): fp32-level accuracy at ~3x the fp32-MFMA rate (cf_gemm_f16.hip).  Differences
 * in the argument block: one slot = 8 channels (src_c multiples of 8, K_pad = 8 * n_slots, multiple
 * of 32); weight = fragment-packed fp16 hi/lo planes of 2^s * W (packing.pack_conv_f16);
 * out_scale = 2^-(s+4); N_pad is 32 or a multiple of 64; output layout NHWC only; act NONE / RELU. */
int cf_conv2d_f16x3(const cf_conv_args* a, void* stream);

/* cf_conv3x3_f16x3: the 3x3 / stride 1 / pad 1 / single-source case of cf_conv2d_f16x3 with LDS patch
 * reuse (cf_conv3x3_f16.hip): each 16-channel slice of the input rows a pixel tile needs is fetched
 * and split once and serves all 9 taps.  Same argument block and same result as cf_conv2d_f16x3 for
 * weights packed slice-major (k = (16-channel slice, tap, channel): K_pad = 144 * C/16 rounded up to
 * 32; the slot table lists that order, so it is only read when the call falls back).  Feature maps too
 * wide for the patch (about W > 250) are forwarded to cf_conv2d_f16x3.  Carries the BasicBlock
 * convolutions (model/networks/dla.py:42-62) and DeformConv.conv_offset_mask (dla.py:406-414). */
int cf_conv3x3_f16x3(const cf_conv_args* a, void* stream);
/* (ABI 3) also the 3x3 / stride 2 / pad 1 case - the BasicBlock conv1 that opens a DLA level (dla.py:124-145): odd / even
 * input columns as two planes of the LDS patch; same packing, same result as cf_conv2d_f16x3. */

/* cf_conv3x3_root_f16x3: BasicBlock.conv2 (+ residual + ReLU; dla.py:33-41) of a one-level Tree's tree2 and the Tree's
 * Root (1x1 convolution + BN + ReLU over cat(x2, x1, *children); dla.py:105-118, 81-96) in ONE launch:
 *   x2 = ReLU(conv3x3(t, W2) + b2 + x1);  out = act(W_root . [x2; x1] + b_root)
 * `conv` is conv2's argument block as for cf_conv3x3_f16x3 (residual = x1, act = RELU; `out` = a buffer for x2, written
 * only when the call falls back); `root` is the Root's block as for cf_conv2d_f16x3 with src[0] = conv->out,
 * src[1] = conv->residual and, behind them, the Tree's children (dla.py:109-117; level_root's pooled input, earlier
 * tree outputs); `root_channels[i]` (host array, n_src entries) = the channels the Root reads from source i.  Where a workgroup holds every channel of its pixels (64 / 128 / 256-channel layers) x2 never leaves the
 * chip: it is split to fp16 hi / lo into LDS as the B operand of the Root's GEMM, whose products and order are those of
 * cf_conv2d_f16x3 - so the result equals the two launches bit for bit, which is what runs for every other shape. */
int cf_conv3x3_root_f16x3(const cf_conv_args* conv, const cf_conv_args* root, const int32_t* root_channels, void* stream);

/* (ABI 5) cf_conv3x3_proj_f16x3: BasicBlock.conv2 of the sub-tree that opens a DLA level TOGETHER WITH the Tree's `project`
 * (1x1 convolution + BN of the 2x2-max-pooled level input), which the reference computes as a tensor and hands to the
 * block as its residual (model/networks/dla.py:96-107 Tree.forward: bottom = downsample(x); residual = project(bottom);
 * x1 = tree1(x, residual); dla.py:56-62 BasicBlock: out = bn2(conv2(.)) + residual; relu):
 *   out = act( conv3x3(src[0], W2) + W_p . src[1] + (b2 + b_p) )
 * One argument block as for cf_conv3x3_f16x3 with n_src = 2: src[0] = the 3x3 input, src[1] = the pooled tensor (same
 * B, H, W as the output), weights = packing.pack_conv_f16(proj=...) - the projection's k-steps (slots: source 1, tap
 * (0, 0)) behind the slice-major 3x3 part, both scaled by one 2^s, bias = the sum.  `src_channels` (host array, 2
 * entries) = the real channels of the two sources (multiples of 32).  The projection's products go into the same
 * accumulators, last: no residual tensor, one launch less per DLA level.  A geometry no patch tiling fits runs
 * cf_conv2d_f16x3 on the same slot table (the same products in the same order; the same bits wherever the patch tiling
 * does not split K over waves - maps of at most 512 pixels with 256+ channels do, as in cf_conv3x3_f16x3). */
int cf_conv3x3_proj_f16x3(const cf_conv_args* a, const int32_t* src_channels, void* stream);

/* ---- (ABI 5) host-side weight preparation: SURVEY 8(b) "cf_pack_weights" (one-time BN fold + layout) -------------------------
 * Pure CPU functions (no stream, no device memory): HOST pointers in, HOST buffers out; the caller copies the results to the
 * device once.  They do what centerfusiondetect3d_amd/packing.py does for the Python host - the same arithmetic operation for
 * operation, the same bytes (tests/test_cabi.py) - so a host in any language can feed the f16x3 operators.  Reference: the
 * eval-mode BatchNorm the reference runs as its own op behind every convolution (model/networks/dla.py:29, 36-39, 151-159;
 * DeformConv: dla.py:399-404) is folded here; the heads / stem / upsample packers stay in packing.py. */
typedef struct cf_pack_src {
  int32_t channels;   /* real channels this source contributes to the convolution's Cin                  */
  int32_t stride;     /* floats per pixel of the NHWC tensor holding it (multiple of 8)                  */
  int32_t c_base;     /* first channel inside that tensor (multiple of 8)                                */
} cf_pack_src;
typedef struct cf_pack_bn {        /* eval-mode BatchNorm to fold (per output channel); gamma == NULL: none */
  const float* gamma; const float* beta; const float* mean; const float* var;
  float eps;
} cf_pack_bn;
typedef struct cf_pack_conv_desc {
  const float* weight;             /* (cout, sum of src channels, kh, kw), fp32, host                       */
  const float* bias;               /* (cout) or NULL                                                        */
  cf_pack_bn bn;
  int32_t cout, kh, kw, stride;
  int32_t pad;                     /* < 0: (kh - 1) / 2 * dilation                                          */
  int32_t dilation;
  const cf_pack_src* src; int32_t n_src;
  const float* proj_weight;        /* optional (cout, proj.channels) 1x1 projection summed into the same    */
  const float* proj_bias;          /*   accumulators (cf_conv3x3_proj_f16x3), with its own BatchNorm        */
  cf_pack_bn proj_bn;
  cf_pack_src proj;
} cf_pack_conv_desc;
typedef struct cf_pack_info {
  int32_t n_pad, k_pad;            /* cf_conv_args.N_pad / K_pad (cf_dcn_args: N_pad; K = 9 * C)            */
  int32_t n_slots;                 /* entries of the slot table (k_pad / 8); 0 for the DCN                  */
  int32_t patch;                   /* slice-major 3x3 packing: cf_conv3x3_f16x3 / _root / _proj may run it  */
  float out_scale;                 /* 2^-(s+4): cf_conv_args.out_scale / cf_dcn_args.out_scale              */
  size_t weight_bytes;             /* size of the packed weight buffer                                      */
} cf_pack_info;
/* sizes first (out_scale is not known before the weights are read: 0 here) ... */
int cf_pack_conv_f16x3_info(const cf_pack_conv_desc* d, cf_pack_info* info);
/* ... then the packing: weight_out (weight_bytes), slots_out (n_slots), bias_out (n_pad floats), all host memory */
int cf_pack_conv_f16x3(const cf_pack_conv_desc* d, void* weight_out, cf_slot* slots_out, float* bias_out, cf_pack_info* info);
/* DeformConv main weights (cout, cin, 3, 3) -> cf_dcn_v2_f16x3's layout (K order tap-major, fp16 hi / lo fragments) */
int cf_pack_dcn_f16_info(int32_t cout, int32_t cin, cf_pack_info* info);
int cf_pack_dcn_f16(const float* weight, const float* bias, const cf_pack_bn* bn, int32_t cout, int32_t cin,
                    void* weight_out, float* bias_out, cf_pack_info* info);

/* cf_split_bf16: fp32 NHWC [M][in_stride] (C used) -> split-bf16 [M][2][Cs], channels C..Cs-1 zero. */
int cf_split_bf16(const float* x, void* out, long M, int C, int in_stride, int Cs, void* stream);

/* cf_head_tail: fused tail of up to CF_MAX_HEADS sibling heads:  x -> [ReLU(W x + b)] x n_hidden -> W_out x
 * + b_out (+ head activation), hidden width 256, on the bf16 MFMA pipe with split operands.  The
 * hidden maps stay in LDS (64-pixel tiles); only the (B, n_out, H, W) fp32 NCHW maps are written.
 * replaces the 1x1 layers of model/networks/detectHeads.py:64-71, 80-90 (one launch instead of
 * 3 per secondary head / 1 per primary head, and no hidden-map round trips through HBM).
 * x: split-bf16 NHWC (B,H,W,2,x_stride); head i reads channels [c_base[i], c_base[i]+256).
 * w_hidden / w_out: weights in MFMA fragment order (centerfusiondetect3d_amd/packing.py:
 * pack_fragments): [row tile of 32][k step of 16][hi,lo][lane 64][8 bf16]; w_out is one row tile
 * (n_out <= 32, zero padded); b_out has 32 floats. */
#define CF_MAX_HEADS 12
typedef struct cf_head_tail_args {
  const void* x;
  int32_t x_stride;
  int32_t B, H, W;
  int32_t n_heads;
  int32_t n_hidden;                        /* 0..2 hidden 256->256 layers per head */
  const void* w_hidden[CF_MAX_HEADS][2];
  const float* b_hidden[CF_MAX_HEADS][2];
  const void* w_out[CF_MAX_HEADS];
  const float* b_out[CF_MAX_HEADS];
  float* out[CF_MAX_HEADS];
  float* out2[CF_MAX_HEADS];               /* second output of CF_ACT_RAW_AND_SIGDEPTH heads, else NULL */
  int32_t c_base[CF_MAX_HEADS];
  int32_t n_out[CF_MAX_HEADS];
  int32_t act[CF_MAX_HEADS];
} cf_head_tail_args;
int cf_head_tail(const cf_head_tail_args* a, void* stream);
/* (ABI 6) LEGACY as a stand-alone launch (CF_LEGACY_HEADS builds only; CF_EINVAL otherwise): the argument block lives on as
 * cf_head_fused_args.tail, whose layers run inside the fused launch. */

/* cf_head_fused: a whole head group in one launch: 3x3 conv (sources -> 256) + ReLU, then the tail
 * of cf_head_tail (tail.x / tail.x_stride / tail.c_base are ignored: the hidden tile is produced in
 * LDS and never written to HBM).  replaces model/networks/detectHeads.py:59-98 end to end.
 * src[]: split-bf16 NHWC sources (feat [, pc_hm]); slots: 8-channel slots as for cf_conv2d_bf16x3,
 * K_pad a multiple of 64; w_first[i]: fragment-packed [256/32][K_pad/16] rows of head i in slot
 * order; b_first[i]: 256 floats. */
typedef struct cf_head_fused_args {
  cf_head_tail_args tail;
  const void* src[2];
  int32_t src_c[2];
  int32_t n_src;
  const cf_slot* slots;
  int32_t K_pad;
  const void* w_first[CF_MAX_HEADS];
  const float* b_first[CF_MAX_HEADS];
  int32_t layout3x3;                       /* 1: the slot order is the canonical one - src[0] (64 channels): 9 taps
                                              x 8 slots, then src[1] (8 channels): 9 taps x 1 slot - so the launch
                                              may run on the 2-D patch kernel (no slot table reads)                 */
  const void* w_out_perm[CF_MAX_HEADS];    /* layout3x3 && n_hidden == 0: w_out with the k order of an accumulator
                                              register group (position 8h+j of a 16-group = channel 4h+(j&3)+8(j>>2)) */
  int32_t mfma16;                          /* layout3x3 only.  != 0: w_first[] and w_out_perm[] are packed for
                                              v_mfma_f32_16x16x32_bf16 - [16-row tile][k step of 32][hi,lo][lane 64][8 bf16],
                                              lane l = row l & 15, k 8 (l >> 4) + j (packing.pack_fragments16); w_out_perm is
                                              ONE 16-row tile whose k position (step ks, group g, j) holds hidden channel
                                              64 (ks >> 1) + 16 (2 (ks & 1) + (j >> 2)) + 4 g + (j & 3); n_out <= 16; tail.w_hidden[][] ([16 tiles][8 k steps])
                                              and tail.w_out[] (one 16-row tile, natural k order) are 16x16x32 fragments too.  The launch
                                              then runs on the 16x16x32 patch kernel (1.14x the 32x32x16 rate under load) */
  int32_t mx;                              /* layout3x3 && mfma16 only.  != 0: the FIRST layer runs as "fp16 main term + block-scaled FP6
                                              cross terms" (v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4, 1.5 passes per
                                              product instead of 3): src[0] is the 272-byte-per-pixel image cf_pack_feat_mx writes
                                              (src_c[0] = 64), w_first[i] the operand stream of packing.pack_head_first_mx (slots / K_pad
                                              are not read), src[1] (optional) the split-bf16 pc_hm planes as before; the tail layers are
                                              unchanged.  ABI 4 */
  float first_scale[CF_MAX_HEADS];         /* mx: 2^-(s+4) of head i's first layer (applied where b_first is added) */
} cf_head_fused_args;
int cf_head_fused(const cf_head_fused_args* a, void* stream);
/* (ABI 6) the default library runs the forms the host dispatches - layout3x3 = 1 with mfma16 = 1 (16x16x32 fragments, n_out <= 16),
 * mx = 0 or 1; the 32x32x16 patch kernel (mfma16 = 0) and the slot-table kernel (layout3x3 = 0) are CF_LEGACY_HEADS builds only
 * (CF_EINVAL otherwise). */

/* cf_pack_feat_mx: fp32 NHWC feature map [M][in_stride] (64 channels used) -> [M][272] bytes for cf_head_fused with mx = 1:
 * per pixel four 64-byte segments g = 0..3 - [8 fp16: hi channels 8g..8g+7][8 fp16: hi channels 32+8g..32+8g+7][FP6 block g:
 * 32 e2m3 fields (element j in bits 6j..6j+5, 24 B) + 8 B zero], block 0 / 1 = lo channels 0-31 / 32-63, block 2 / 3 = hi
 * channels 0-31 / 32-63, where hi = fp16(clamp(16 x)), lo = 16 x - hi - then [4 E8M0 scale bytes, block order][12 B zero].
 * A block's scale is 2^e with e the smallest integer such that max|.| <= 7.5 * 2^e.
 * replaces: nothing in the reference - it is the operand preparation of model/networks/detectHeads.py:64-79 on this path
 * (the fp32 -> split conversion the bf16x3 heads take from cf_split_bf16 / the DCN epilogue).  ABI 4 */
int cf_pack_feat_mx(const float* x, int in_stride, void* rows, long M, void* stream);
/* (ABI 6) the same rows with hi = fp16(clamp(scale x)), lo = scale x - hi for a power-of-two `scale` (cf_pack_feat_mx: 16); the
 * head launch that reads them gets first_scale = 2^-s / scale.  For feature maps whose values exceed 65504 / 16 / 2. */
int cf_pack_feat_mx_scaled(const float* x, int in_stride, void* rows, long M, float scale, void* stream);

/* cf_dcn_v2_fused: modulated deformable 3x3 convolution (stride 1, pad 1, dil 1, groups 1) with
 * the bilinear gather fused into the GEMM A-tile staging, + bias(BN folded) + ReLU.
 * replaces torchvision.ops.deform_conv2d + BN + ReLU of model/networks/dla.py:456-472.
 * `offmask` is the raw output of conv_offset_mask, NHWC with `om_stride` floats per pixel:
 * channels 2k / 2k+1 = dy / dx of tap k, channels 18+k = mask logit of tap k (sigmoid applied
 * here unless mask_activated) - the chunk/cat of dla.py:457-459 is the identity on the first 18 channels. */
typedef struct cf_dcn_args {
  const float* x;       /* NHWC [B][H][W][C]                    */
  const float* offmask; /* NHWC [B][H][W][om_stride], 27 used   */
  int32_t om_stride;
  int32_t B, H, W, C;   /* C multiple of 32                     */
  const float* weight;  /* [N_pad][9*C], k = tap*C + c          */
  const float* bias;    /* [N_pad]                              */
  int32_t N, N_pad;
  float* out;           /* NHWC [B][H][W][out_stride]           */
  int32_t out_stride;
  int32_t act;
  int32_t precise;      /* as cf_conv_args.precise              */
  float out_scale;      /* cf_dcn_v2_f16x3 only: 2^-s / in_scale (2^-(s+4) at the default in_scale) */
  void* out_split_bf16; /* cf_dcn_v2_f16x3 only, optional: the same result additionally as split-bf16 NHWC
                           [B][H][W][2 (hi, lo)][split_stride] - what the head kernels read (saves cf_split_bf16) */
  int32_t split_stride; /* channels per plane of out_split_bf16 (>= N, multiple of 8) */
  void* workspace;      /* cf_dcn_v2_f16x3 only, optional: cf_dcn_v2_workspace_bytes(...) bytes.  With it, small
                           maps (<= 2048 pixels per image) split K over 2-4 workgroups per tile and a second
                           launch adds the partial sums in fixed order; without it K is never split */
  size_t workspace_bytes; /* size of `workspace`; checked against cf_dcn_v2_workspace_bytes(...) when K is split */
  int32_t mask_activated; /* 0: offmask channels 18..26 are mask LOGITS, sigmoid applied here (the fused DeformConv of
                             the module path); != 0: they already are the modulation factors, used as they are - the
                             contract of torchvision.ops.deform_conv2d(mask=...) (dla.py:460: the caller's sigmoid) */
  void* out_mx;           /* cf_dcn_v2_f16x3 only, optional, N = N_pad = 64, not on K-split maps: the same result additionally
                             as the [B*H*W][272] byte rows of cf_pack_feat_mx (bit-identical to packing `out`): the operand
                             format of cf_head_fused with mx = 1 - saves that pass over the feature map.  ABI 4 */
  float in_scale;         /* (ABI 6) cf_dcn_v2_f16x3 only: power of two the sampled values are multiplied by (through the
                             modulation factor) before the fp16 split; 0 = 16; as cf_conv_args.in_scale */
  float mx_scale;         /* (ABI 6) with out_mx: the rows' pre-scale (cf_pack_feat_mx_scaled's `scale`); 0 = 16 */
} cf_dcn_args;
int cf_dcn_v2_fused(const cf_dcn_args* a, void* stream);

/* cf_dcn_v2_f16x3: cf_dcn_v2_fused with the main GEMM on the f16 MFMA pipe from split operands (see
 * cf_conv2d_f16x3); weight = fragment-packed fp16 hi/lo planes of 2^s * W in (tap, channel) K order
 * (packing.pack_dcn_f16), out_scale = 2^-(s+4).  x / offmask / out stay fp32 NHWC. */
int cf_dcn_v2_f16x3(const cf_dcn_args* a, void* stream);
size_t cf_dcn_v2_workspace_bytes(int B, int H, int W, int C, int N_pad);

/* cf_upsample_dw: depthwise transposed conv (k = 2f, stride f, pad f/2, groups = C, no bias),
 * optionally fused with the IDA skip add:  out = convT(x) [+ skip].
 * replaces aten::conv_transpose2d + add of model/networks/dla.py:502-511, 518-524.
 * weight is repacked [k][k][C]; all tensors NHWC; output is (H*f, W*f). */
int cf_upsample_dw(const float* x, const float* weight, const float* skip, float* out, int B,
                   int H, int W, int C, int f, void* stream);

/* cf_maxpool2x2: 2x2 stride-2 max pooling, NHWC.  replaces aten::max_pool2d of dla.py:96,107. */
int cf_maxpool2x2(const float* x, float* out, int B, int H, int W, int C, void* stream);

/* cf_nchw_to_nhwc4: (B,3,H,W) NCHW image -> (B,H,W,4) NHWC with a zero 4th channel (stem input). */
int cf_nchw_to_nhwc4(const float* x, float* out, int B, int C, int H, int W, void* stream);

/* cf_nhwc_to_nchw: generic layout change used to hand NHWC intermediates back in NCHW. */
int cf_nhwc_to_nchw(const float* x, float* out, int B, int H, int W, int C, int c_stride,
                    void* stream);

/* cf_nchw_to_nhwc: (B,C,H,W) -> NHWC with `out_stride` floats per pixel, written at channel `out_offset`
 * (the other channels of the destination are left alone).  Entry side of the operator-level deform_conv2d
 * drop-in (ops.deform_conv2d: NCHW input / offset / mask of dla.py:461-470 -> the NHWC buffers of cf_dcn_v2_*). */
int cf_nchw_to_nhwc(const float* x, float* out, int B, int C, int H, int W, int out_stride, int out_offset,
                    void* stream);

/* cf_radar_ingest: raw radar sweeps -> what cf_pillar_expand consumes (detector.py:257-283,
 * datasets/nuscenes.py:171-199, utils/pointcloud.py:17-49): depth <= max_dist
 * (if > 0), y -= z_offset, pinhole projection with the 3x3 intrinsic and division by the projected z,
 * keep depth > 0 and 1 < u < img_w - 1 and 1 < v < img_h - 1, order by depth (ties: original index;
 * descending = exact reverse).  pc (B, n_rows, max_n) f64 padded sweeps with counts_in (B) points each
 * (rows 0..2 = x, y, z in the camera frame); intrinsics (B, 3, 3) f64.  Outputs: pc_2d (B, 3, max_n)
 * [u, v, depth], pc_3d (B, n_rows, max_n) (row 1 carries the offset y), counts_out (B); padding zeroed. */
int cf_radar_ingest(const double* pc, const int32_t* counts_in, int B, int n_rows, int max_n,
                    const double* intrinsics, int img_w, int img_h, double max_dist, double z_offset,
                    int descending, double* pc_2d, double* pc_3d, int32_t* counts_out, void* stream);

/* cf_preprocess_images: the image side of Detector.pre_process (detector.py:226-234; SURVEY §8(f)
 * rank 2): cv2.warpAffine(INTER_LINEAR, border 0) of uint8 HWC camera frames to the network input
 * size in OpenCV's fixed-point arithmetic, ((v / 255 - mean) / std) evaluated in float64, fp32 NCHW out.
 * src (B, Hs, Ws, 3) uint8 on the device; map_dst_to_src, mean, stdv: HOST pointers (six doubles: the
 * already inverted 2x3 matrix, as cv::warpAffine forms it; three floats each). */
int cf_preprocess_images(const uint8_t* src, int B, int Hs, int Ws, const double* map_dst_to_src,
                         const float* mean, const float* stdv, int Hd, int Wd, float* out, void* stream);

/* cf_topk_peaks: per-image top-K over a (B,C,H,W) NCHW score map, optionally after the 3x3
 * equality NMS, ordered by (score desc, class asc, pixel asc).
 * replaces model/utils.py:6-38 (topk) [+ model/utils.py:112-128 (nms) when nms != 0].
 * outputs: scores (B,K) f32, inds (B,K) i32 pixel index in [0,H*W), classes (B,K) i32.
 * workspace: cf_topk_workspace_bytes(B, K) bytes of device memory (per-slice candidate keys). */
size_t cf_topk_workspace_bytes(int B, int K);
/* nms == 2: the same result as nms == 1, but the suppressed map is first written by its own fully
 * parallel pass into scratch memory behind the keys - workspace must then hold
 * cf_topk_workspace_bytes_nms(B, C, H, W, K) bytes.  (nms == 1 suppresses on the fly, no scratch.) */
size_t cf_topk_workspace_bytes_nms(int B, int C, int H, int W, int K);
/* (ABI 6) cf_checksum64: a position-weighted checksum of a device buffer read as 32-bit words, as CF_CHECKSUM_PARTS partial sums
 * over a fixed partition of the words: parts[b] = sum of word[i] * w(i) mod 2^64, w(i) = ((uint32)i * 2654435761) | 1 (exact
 * integer arithmetic: the same bits give the same parts; one launch, no atomics, nothing to zero).
 * cf_topk_peaks_if_changed: the guard of peaks computed earlier - sums = [expected parts | actual parts] (2 x
 * CF_CHECKSUM_PARTS words in device memory, compared ON THE DEVICE: no host sync).  All equal: the launch does nothing and
 * scores / inds / classes keep what they hold.  Any part differs: it computes cf_topk_peaks(heat, nms) into them - one
 * workgroup per image, ~0.3 ms: the rare path.  nms = 1 or 2 (same result); workspace: cf_topk_workspace_bytes(B, K) bytes.
 * Together they let the host keep the peaks the forward computed beside its own launches (model.py: heads_lanes) and still
 * honour ANY later change of the heat map - also one made through `tensor.data`, which no version counter sees - as the
 * reference's fusionDecode would (model/decode.py:38-57: it always reads the map it is given).  No reference counterpart. */
#define CF_CHECKSUM_PARTS 256
int cf_checksum64(const void* x, long n_words, unsigned long long* parts, void* stream);
int cf_topk_peaks_if_changed(const float* heat, int B, int C, int H, int W, int K, int nms, float* scores, int32_t* inds,
                             int32_t* classes, void* workspace, const unsigned long long* sums, void* stream);
int cf_topk_peaks(const float* heat, int B, int C, int H, int W, int K, int nms, float* scores,
                  int32_t* inds, int32_t* classes, void* workspace, void* stream);

/* cf_frustum_assoc: radar frustum association for K peaks per image.
 * replaces utils/pointcloud.py:331-394 (getPcFrustumHeatmap, after its topk) and 397-481
 * (cvtPcDepthToHeatmap) incl. get_alpha / cvtAlphaToYaw / get3DCorners / getDistanceThresh
 * (pointcloud.py:195-328).  All maps NCHW fp32.  pc_hm (B,3,H,W) is fully written (zero where
 * nothing is painted); pc_hm_nhwc4 (B,H,W,4) fp32 and pc_hm_split8 (B,H,W,2,8) split-bf16, if not
 * NULL, receive the same data channels-last for the secondary-head convolution. */
int cf_frustum_assoc(const int32_t* inds, int K, const float* depth, const float* wh,
                     const float* dim, const float* rot, const float* calib, const float* pc_dep,
                     int B, int H, int W, float max_pc_dist, float* pc_hm, float* pc_hm_nhwc4,
                     void* pc_hm_split8, void* stream);

/* (ABI 6) cf_topk_frustum: cf_topk_peaks(heat, nms = 0) followed by cf_frustum_assoc on its peaks - the chain between the
 * primary and the secondary head launches (utils/pointcloud.py:347-392: topk of the un-NMS'd heat map, then the association
 * loop) - as TWO launches instead of three: the slice top-K, then the association kernel, which merges the slices' sorted
 * lists in its prologue.  heat (B,C,H,W); workspace: cf_topk_workspace_bytes(B, K) bytes; scores / inds / classes (B,K):
 * optional (all three or none) - the peaks as cf_topk_peaks would return them.  Same results as the two calls, bit for bit. */
int cf_topk_frustum(const float* heat, int C, int K, const float* depth, const float* wh, const float* dim,
                    const float* rot, const float* calib, const float* pc_dep, int B, int H, int W, float max_pc_dist,
                    float* pc_hm, float* pc_hm_nhwc4, void* pc_hm_split8, float* scores, int32_t* inds, int32_t* classes,
                    void* workspace, void* stream);

/* cf_pillar_expand: radar points -> pc_dep (B,3,H,W) by pillar expansion (fp64 geometry).
 * replaces dataset/generic_dataset.py:738-942 (processPointCloud / transformPointCloud /
 * getPcPillarsSize) + dataset/datasets/nuscenes.py:221-263 (getDepthMap / drawPcHeat).
 * pc_2d: (B, 3, max_n) f64 [u, v, depth] in source-image pixels, depth-ascending per frame;
 * pc_3d: (B, n_rows>=10, max_n) f64 camera-frame radar rows (row 8 = vx, row 9 = vz);
 * counts: (B) i32 valid points per frame; calib (B,3,4) f64; trans (B,2,3) f64 (source image ->
 * output map affine); pillar = (h, w, l) metres.  keep_mask (B,max_n) u8 and xy_out (B,2,max_n)
 * f64 may be NULL; they receive the transformPointCloud filter mask / transformed coordinates. */
int cf_pillar_expand(const double* pc_2d, const double* pc_3d, const int32_t* counts, int B,
                     int max_n, int n_rows, const double* calib, const double* trans, int H, int W,
                     double pillar_h, double pillar_w, double pillar_l, float* pc_dep,
                     uint8_t* keep_mask, double* xy_out, void* stream);

/* cf_decode_gather: from K (index, class, score) peaks produce the 33-float detection rows of
 * model/decode.py:10-174 (fusionDecode): [score, classId, cx_n, cy_n, x1, y1, x2, y2, rot8, dim3,
 * amodal2, att8, vel3, depth].  Any map pointer may be NULL (its columns are zero-filled; reg NULL
 * means the +0.5 centre of decode.py:139-141).  det: (B,K,33) f32. */
typedef struct cf_decode_args {
  const float* scores;  /* (B,K) */
  const int32_t* inds;  /* (B,K) */
  const int32_t* classes;
  const float *reg, *wh, *depth, *rot, *dim, *amodal, *att, *vel; /* NCHW maps */
  int32_t B, K, H, W;
  int32_t out_h, out_w; /* outputSize */
  int32_t norm2d;
  float* det;           /* (B,K,33) */
} cf_decode_args;
int cf_decode_gather(const cf_decode_args* a, void* stream);

/* cf_post_process: 2D -> 3D post-processing of decoded detections, replaces utils/postProcess.py:13-85
 * (+ utils/ddd.py:8-23,122-199, utils/pointcloud.py:195-296) for the inference case.
 * det (B,K,33) as written by cf_decode_gather; calib (B,3,4) f32; trans_inv (2,3) f32 device = the
 * output-map -> source-image affine (getAffineTransform(..., inverse=True)); out (B,K,54) f32:
 * [score, classId+1, centre xy (source px), bbox x1 y1 x2 y2 (source px), depth, alpha, dim hwl,
 *  amodal_offset xy, nuscenes_att 8, velocity 3 (re-projected on the heading), location xyz, yaw,
 *  8 box corners xyz (all zero when a dimension is <= 0)]. */
int cf_post_process(const float* det, const float* calib, const float* trans_inv, int B, int K,
                    int out_h, int out_w, float* out, void* stream);

/* cf_decode_post: cf_decode_gather and cf_post_process in ONE launch (the 33-float row never leaves
 * registers): model/decode.py:10-174 followed by utils/postProcess.py:13-85, as Detector.process +
 * Detector.post_process chain them (detector.py:343-349, 397-426) and as the evaluation loop does
 * (model/progressBar.py:95-110).  post (B,K,54) as cf_post_process writes it; a->det (B,K,33) is
 * optional (NULL: only the final rows).  Bit-identical to the two separate calls. */
int cf_decode_post(const cf_decode_args* a, const float* calib, const float* trans_inv, float* post,
                   void* stream);

/* cf_serialize_nuscenes: post-processed detections -> the numeric content of the nuScenes result
 * file, replaces dataset/datasets/nuscenes.py:416-482 (getEvalFormatItem) and the per-sample merge +
 * top-500 of nuscenes.py:536-553 (convert_eval_format); SURVEY §8(f) rank 4.
 *   rows (B*K, 12) f32: [translation xyz = trans_matrix @ (location - (0, h, 0), 1), size (w, l, h),
 *                        velocity xy = (velocity_matrix @ (v, 0))[:2], score, class index 0..9,
 *                        attribute id 0..8 (0 = ""), keep = score > -1 && all dimensions > 0]
 *   rotation (B*K, 4) f64 (optional): pose_rot * cs_rot * R_y(yaw), (w, x, y, z); NULL cs/pose: R_y(yaw)
 *   order (n_samples, max_per_sample) i32: row indices (b*K + k) of each sample's best rows, best first,
 *     -1 padded; counts (n_samples).  A sample's frames are sample_frames[sample_ptr[s] .. sample_ptr[s+1])
 *     in image order (CSR); at most cf_serialize_max_candidates() kept rows per sample take part.
 * The strings of the file (sample_token, detection_name, attribute_name) are joined on the host. */
typedef struct cf_serialize_args {
  const float* post;             /* (B,K,54) from cf_post_process / cf_decode_post                */
  int32_t B, K;
  const float* trans_matrix;     /* (B,4,4) f32 camera -> global (image_info["trans_matrix"])     */
  const float* velocity_matrix;  /* (B,4,4) f32 (image_info["velocity_trans_matrix"])             */
  const double* cs_rot;          /* (B,4) f64 calibrated-sensor quaternion, or NULL               */
  const double* pose_rot;        /* (B,4) f64 ego-pose quaternion, or NULL                        */
  float* rows;                   /* (B*K,12)                                                      */
  double* rotation;              /* (B*K,4) or NULL                                               */
  int32_t n_samples;             /* 0: rows only                                                  */
  const int32_t* sample_ptr;     /* (n_samples+1)                                                 */
  const int32_t* sample_frames;  /* (sample_ptr[n_samples]) frame indices                         */
  int32_t max_per_sample;        /* 500 in the reference                                          */
  int32_t* order;                /* (n_samples, max_per_sample)                                   */
  int32_t* counts;               /* (n_samples)                                                   */
} cf_serialize_args;
int cf_serialize_nuscenes(const cf_serialize_args* a, void* stream);
int cf_serialize_max_candidates(void);

/* Weight packing runs once per load_state_dict (BatchNorm fold, slot tables, split hi / lo planes, MFMA fragment order).
 * For the f16x3 convolution and DCN operators it is part of this ABI (cf_pack_conv_f16x3, cf_pack_dcn_f16 above: host-side C,
 * byte-identical to the Python packers); the heads', the stem's and the upsample's weights are packed by
 * centerfusiondetect3d_amd/packing.py only (the reference's own host side is Python).  The layouts the kernels expect
 * are documented at each argument block above and in DESIGN.md section 3. */

/* Dynamic range of the f16x3 / mx kernels.  They evaluate fp32 products from operands split into two fp16 values after a
 * power-of-two pre-scale: weights per layer (chosen by the packer from max|W|), activations by `in_scale` (default 16).  An
 * activation with |x| * in_scale > 65504 is CLAMPED to +-65504 / in_scale - silently, the kernels carry no overflow flag.
 * The reference's fp32 convolutions accept any magnitude (model/networks/dla.py:124-159), so a caller must either know
 * its activations stay below 65504 / in_scale (4094 at the default) or measure them - cf_absmax_f32 on the source buffers -
 * and pass a smaller in_scale (out_scale = 2^-s / in_scale).  The Python host does exactly that: DLASeg.check_ranges /
 * calibrate / activation_ranges (centerfusiondetect3d_amd/model.py), Detector(range_policy=...).
 *
 * cf_absmax_f32: out[0] = max |x[m][c]| over m < M, c < C of an fp32 buffer with `stride` floats per row, as a float
 * (NaN if any element is NaN, +inf if any is infinite).  `out` (device, one float) is zeroed by the call (hipMemsetAsync on
 * `stream`) before the reduction; M = 0 leaves 0 there.  replaces: nothing in the reference (range guard of this path). */
int cf_absmax_f32(const float* x, long M, int C, int stride, float* out, void* stream);

/* cf_spin_us: diagnostic - ONE 64-thread workgroup that stays resident for `microseconds` (<= 100000) of the
 * constant 100 MHz clock and then exits.  Two of them on two streams finish in ~1x the time when the streams run
 * concurrently and in ~2x when HIP has put them on one hardware queue: the host's one-time stream-pair probe
 * (centerfusiondetect3d_amd/model.py, _side_streams).  No reference counterpart (the reference leaves stream
 * placement to PyTorch: it never forks streams). */
int cf_spin_us(int microseconds, void* stream);

const char* cf_last_error(void);
int cf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CF_HIP_H */
