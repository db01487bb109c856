/*
 * polyhead.h -- C ABI of libpolyhead.so: the MI355X (gfx950) implementation of PolyphonicFormer's
 * unified-query decode hot path.  Plain pointers, sizes and a hipStream_t (passed as void*); no
 * torch types.  Every entry point
 *   - returns 0 on success or a negative PH_E* code (never throws, see ph_last_error_string),
 *   - never allocates and never synchronises: the caller owns every buffer and passes the stream,
 *   - is thread-compatible (no mutable globals besides a thread-local error string).
 *
 * The reference is 100 % Python: there is no FFI in it to mirror.  Each function
 * below therefore cites the reference *Python* lines whose device work it replaces; the Python
 * classes in polyphonicformer_amd/ keep the reference's registry names / kwargs / state_dict keys
 * and call these through ctypes (INTEGRATION.md shows the binding).
 *
 * Internal device formats:
 *   feature planes  : uint16 (bf16 bits) [P][B][256][HWp], HWp = HW rounded up to 128, zero padded.
 *                     P = 1 for PH_PREC_BF16; P = 2 (hi, lo with x ~= hi + lo to 2^-17) for
 *                     PH_PREC_SPLIT, the fp32-grade mode used for the 1e-3 parity runs.
 *   mask bits       : uint32 [B][Npad][HWp/32], bit j of word w = 1[logit(pixel 32w+j) > 0];
 *                     Npad = N rounded up to 32; rows >= N and pixels >= HW are 0.
 *   query matrices  : fp32 [B][N][256] row major at the API; bf16 planes [P][...][Npad][256] inside.
 */
#ifndef POLYHEAD_H_
#define POLYHEAD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PH_VERSION 100

enum { PH_OK = 0, PH_EINVAL = -1, PH_EUNSUPPORTED = -2, PH_ELAUNCH = -3, PH_EWORKSPACE = -4 };
/* Arithmetic of an operator = element format of its 16-bit operands x planes per operand:
 *   PH_PREC_BF16         bf16, one plane each                          (1 MFMA per product, ~2^-9 per operand)
 *   PH_PREC_BF16_KSPLIT  ph_dynconv only: bf16 features in ONE plane, the dynamic kernels as hi + lo planes (2 MFMAs):
 *                        exact for features that ARE bf16 values, kernels to 2^-17
 *   PH_PREC_SPLIT        bf16, hi + lo planes of both operands, a.b ~= ah.bh + ah.bl + al.bh (3 MFMAs, ~2^-16)
 *   PH_PREC_F16          IEEE fp16, one plane each (1 MFMA, ~2^-12 per operand; |values| < 65504)
 *   PH_PREC_BF16_KF16    ph_dynconv only: bf16 features in ONE plane, the dynamic kernels as ONE fp16 plane (PH_KERN_F16); the
 *                        feature fragments are converted to fp16 in registers (exact inside fp16's normal range) and the
 *                        product is one f16 MFMA: ~2^-12 on the kernels only -- 2.5e-4 per stage, single-plane speed.
 *   PH_PREC_QHYBRID      ph_query_stage only: the updator half (pooled sums, gate products: unbounded operands) as
 *                        PH_PREC_SPLIT, the attention / FFN / fc-tower half -- every product there has a LayerNorm- or softmax-
 *                        bounded operand -- as ONE fp16 plane of weights and activations (1 MFMA, 2^-12 per operand, half the
 *                        weight stream); weights packed accordingly (pack.py), dynamic kernels out as PH_KERN_F16. */
enum { PH_PREC_BF16 = 1, PH_PREC_BF16_KSPLIT = 2, PH_PREC_SPLIT = 3, PH_PREC_F16 = 5, PH_PREC_BF16_KF16 = 6, PH_PREC_QHYBRID = 7 };
/* flag OR-ed into `prec` of ph_nhwc_ingest (output) and ph_conv_nhwc (input, 3x3 stride 2, one-plane formats): the planes are
   chunk-major [frame][256 / 16][HW][16] instead of channels-last [frame][HW][256].  The stride-2 kernel stages 16 channels of
   a 9 x 129 pixel patch at a time; from channels-last planes that is 32 bytes of every 512, i.e. a quarter of each 128-byte
   line per stage and (the lines do not survive in L2 between stages) four times the plane's bytes from HBM. */
enum { PH_PLANES_C16 = 0x100 };
enum { PH_OUT_F32 = 0, PH_OUT_BF16 = 1, PH_OUT_F16 = 2 };
enum { PH_KERN_BF16_PLANES = 0, PH_KERN_F16 = 1 };   /* ph_query_stage: format of the dynamic conv kernels it emits */
enum { PH_GN_TO_PLANES = 0, PH_GN_UP2_PLANES = 1, PH_GN_ACCUM = 2, PH_GN_TO_NCHW = 3, PH_GN_TO_CPLANES = 4 };   /* ph_gn_apply modes */
enum { PH_IN_F32_NCHW = 0, PH_IN_PLANES = 1 };   /* ph_khead_fused input_format */

#define PH_C 256          /* channels: in_channels == out_channels == feat_channels == 256 */
#define PH_HEADS 8        /* num_heads (head dim 32) */

int         ph_version(void);
const char* ph_last_error_string(void);

/* ---- geometry helpers ------------------------------------------------------------------- */
static inline int64_t ph_hw_padded(int64_t hw) { return (hw + 127) / 128 * 128; }
static inline int     ph_n_padded(int n) { return (n + 31) / 32 * 32; }

/* ---- feature-map ingest ------------------------------------------------------------------
 * fp32 NCHW [B][256][HW] (what KernelHead hands over, kernel_head.py:347 x_feats/depth_feats)
 * -> bf16 planes [P][B][256][HWp].  Replaces nothing arithmetic in the reference; it is the
 * format change at the boundary. */
int ph_ingest_features(const float* src, uint16_t* planes, int B, int64_t HW, int prec, void* stream);

/* fp32 mask logits [B][N][HW] -> mask bits.  kernel_update_head.py:236-238
 * (sigmoid -> > hard_mask_thr(0.5) -> float), stated as logit > 0. */
int ph_binarize(const float* logits, int64_t logits_batch_stride /* elements; 0 = N*HW (contiguous) */, uint32_t* bits,
                int B, int N, int64_t HW, void* stream);
/* the same for fp32 or fp16 logits (dtype PH_OUT_F32 / PH_OUT_F16), optionally predicated on a device word: when run_if is
 * non-NULL and *run_if == 0 at execution time the launch returns at once (see ph_khead_fused_if) */
int ph_binarize_if(const void* logits, int dtype, int64_t logits_batch_stride, uint32_t* bits, int B, int N, int64_t HW,
                   const uint32_t* run_if /* nullable */, void* stream);

/* ---- A7: masked pooling -------------------------------------------------------------------
 * kernel_update_head.py:241-242  einsum('bnhw,bchw->bnc') for x and depth_feats in one pass
 * (and kernel_head.py:320 with only `xplanes`).  Split over `nsplit` pixel ranges; the
 * deterministic partial sums land in partial[B][nsplit][Npad][512] (cols 0..255 = x,
 * 256..511 = depth_feats) and are summed in fixed order by the consumer. `dplanes` may be NULL. */
int ph_pool(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, float* partial,
            int B, int N, int64_t HW, int nsplit, int prec, void* stream);
/* the same over the first ph_n_padded(N) rows of a bits tensor that has `bits_rows` (>= that) rows per frame:
 * kernel_head.py:314-320 pools over the THING rows of the full mask tensor */
int ph_pool_rows(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, int bits_rows, float* partial,
                 int B, int N, int64_t HW, int nsplit, int prec, void* stream);
/* ph_pool that also writes pcount[B][nsplit][ph_n_padded(N)] (int32): the set bits of every mask row in every pixel range -- the
 * `count(M)` of the folded feat_transform bias (kernel_update_head.py:225,241), handed to ph_query_stage_counts */
int ph_pool_counts(const uint16_t* xplanes, const uint16_t* dplanes, const uint32_t* bits, float* partial, int32_t* pcount,
                   int B, int N, int64_t HW, int nsplit, int prec, void* stream);

/* ---- packed per-stage weights -------------------------------------------------------------
 * One KernelUpdateHead stage (kernel_update_head.py:21-191 parameters) packed by the host into
 *   wb : uint16 [P][wb_plane_elems]   bf16 MFMA B-fragments, tile-major
 *   wf : float  [..]                  biases, LayerNorm affine, folded vectors
 * with offsets (in elements) indexed by the enums below; branch 0 = mask, 1 = depth. */
enum {
    PH_W_DYN = 0,   /* dynamic_layer folded with feat_transform: [512][256]   kernel_updator.py:58, kernel_update_head.py:225 */
    PH_W_INP,       /* input_layer [512][256]                                  kernel_updator.py:64 */
    PH_W_IG,        /* input_gate  [256][256]                                  :73 */
    PH_W_UG,        /* update_gate [256][256]                                  :74 */
    PH_W_FC,        /* fc_layer    [256][256]                                  :89 */
    PH_W_QKV,       /* attn.in_proj_weight [768][256]                          kernel_update_head.py:112-115 */
    PH_W_OUT,       /* attn.out_proj [256][256] */
    PH_W_FFN1,      /* ffn.layers.0.0 [F][256]                                 :146-151 */
    PH_W_FFN2,      /* ffn.layers.1   [256][F] */
    PH_W_H0A,       /* cls_fcs.0 (mask branch) / depth_regs.0 (depth branch)   :161-187 */
    PH_W_H0B,       /* mask_fcs.0 (mask branch only) */
    PH_W_CLS,       /* fc_cls [Lpad][256] (mask branch only)                   :169-172 */
    PH_W_KERN,      /* fc_mask / fc_depth folded with feat_(depth_)transform: [272][256]; row 256 = bias dot  :189-190,225-226 */
    PH_W_COUNT
};
enum {
    PH_V_DYN_CNT = 0, /* [512] dynamic_layer.weight @ feat_transform.bias (multiplies the pixel count) */
    PH_V_DYN_B, PH_V_INP_B, PH_V_IG_B, PH_V_UG_B,
    PH_V_LN_IG_G, PH_V_LN_IG_B,   /* input_norm_in   kernel_updator.py:73 */
    PH_V_LN_UG_G, PH_V_LN_UG_B,   /* norm_in         :74 */
    PH_V_LN_PO_G, PH_V_LN_PO_B,   /* norm_out        :78 */
    PH_V_LN_IO_G, PH_V_LN_IO_B,   /* input_norm_out  :79 */
    PH_V_FC_B, PH_V_LN_FC_G, PH_V_LN_FC_B,
    PH_V_QKV_B, PH_V_OUT_B, PH_V_LN_ATT_G, PH_V_LN_ATT_B,
    PH_V_FFN1_B, PH_V_FFN2_B, PH_V_LN_FFN_G, PH_V_LN_FFN_B,
    PH_V_LN_H0A_G, PH_V_LN_H0A_B, PH_V_LN_H0B_G, PH_V_LN_H0B_B,
    PH_V_CLS_B, PH_V_KERN_B,       /* [272] folded fc_mask/fc_depth bias (entry 256 = bias . transform bias) */
    PH_V_COUNT
};
typedef struct {
    int64_t w[2][PH_W_COUNT];   /* element offsets into one plane of wb */
    int64_t v[2][PH_V_COUNT];   /* element offsets into wf */
    int64_t wb_plane_elems;     /* distance between the hi and lo plane */
    int32_t ffn_dim;            /* F, multiple of 256 */
    int32_t num_classes;        /* L */
} ph_stage_layout;

/* ---- A8-A12: the query side of one stage ---------------------------------------------------
 * kernel_update_head.py:245-288 + funcs/kernel_updator.py:55-93 for both branches:
 *   pre : partial-sum reduce, KernelUpdator x2, attention in-projection
 *   post: self-attention per branch, out-proj + LN, FFN + LN, cls / mask-kernel / depth-kernel heads
 * Inputs  k_in, q_in fp32 [B][N][256] (proposal_feat, depth_proposal); bits for the pixel counts.
 * Outputs obj, dobj fp32 [B][N][256]; cls fp32 [B][N][L]; kern planes [P][2][B][Npad][256] and
 *         kbias fp32 [2][B][Npad]: the dynamic 1x1 conv kernels already folded with
 *         feat_transform, i.e. new_mask_logits = kern[0] . x + kbias[0]  (kernel_update_head.py:317-329).
 * Workspace (ph_query_workspace_bytes): q/k/v planes + residual.
 * phases: PH_QUERY_PRE | PH_QUERY_POST, optionally | PH_QUERY_WIDE: as many rows per workgroup as divide the padded row
 *         count (fewest CU-seconds; for launches that share the GPU with other streams).  Without it the launch is shaped
 *         to fill the chip by itself (lowest latency of a single launch). */
enum { PH_QUERY_PRE = 1, PH_QUERY_POST = 2, PH_QUERY_BOTH = 3, PH_QUERY_WIDE = 0x100 };
size_t ph_query_workspace_bytes(int B, int N, int prec);
/* byte offset, inside the workspace, of the fp32 [B][2][Npad][256] KernelUpdator outputs
 * (funcs/kernel_updator.py:93) that the PRE phase leaves behind for the POST phase. */
size_t ph_query_workspace_updator_offset(int B, int N, int prec);
int ph_query_stage(const float* partial, int nsplit, const uint32_t* bits,
                   const float* k_in, const float* q_in,
                   const uint16_t* wb, const float* wf, const ph_stage_layout* layout,
                   float* obj, float* dobj, float* cls, int cls_sigmoid /* kernel_update.py:396-397 */,
                   uint16_t* kern, float* kbias,
                   void* workspace, size_t workspace_bytes,
                   int B, int N, int64_t HW, int prec /* PH_PREC_BF16 | PH_PREC_SPLIT */,
                   int kern_format /* PH_KERN_BF16_PLANES: [P][2][B][Npad][256]; PH_KERN_F16: one fp16 plane */,
                   int phases, void* stream);
/* the same, taking the hard masks' pixel counts from ph_pool_counts instead of counting the bit rows again; prec may also
 * be PH_PREC_QHYBRID (with PH_KERN_F16) in both entry points */
int ph_query_stage_counts(const float* partial, int nsplit, const uint32_t* bits, const int32_t* pcount, const float* k_in,
                          const float* q_in, const uint16_t* wb, const float* wf, const ph_stage_layout* layout,
                          float* obj, float* dobj, float* cls, int cls_sigmoid, uint16_t* kern, float* kbias,
                          void* workspace, size_t workspace_bytes, int B, int N, int64_t HW, int prec, int kern_format,
                          int phases, void* stream);

/* ---- A13: dynamic 1x1 convolution -----------------------------------------------------------
 * kernel_update_head.py:317-329: logits[b][n][hw] = sum_c kern[b][n][c] * feat[b][c][hw] + kbias[b][n].
 * Either writes the mask bits the next stage pools with (bits_out != NULL; the logits of a
 * non-final stage are consumed only through `> 0`, kernel_update_head.py:236-238) or the logits
 * themselves (logits_out, dtype out_dtype = PH_OUT_F32 / BF16 / F16).  `prec` = PH_PREC_BF16 (1 feature plane, 1 kernel
 * plane), PH_PREC_BF16_KSPLIT (1, 2), PH_PREC_SPLIT (2, 2), PH_PREC_F16 (fp16 planes, 1, 1) or PH_PREC_BF16_KF16 (bf16 plane, one fp16 kernel plane).
 * `kern` is planes [P][..][Npad][256] (plane stride
 * given), `kern_batch_stride` / `kbias_batch_stride` / `out_batch_stride` are the element distances
 * between frames: Npad*256 / Npad / N*HW for per-frame dynamic kernels; 0 / 0 / rows*HW when the
 * same static 1x1 conv weights serve every frame (kernel_head.py:256,285,295 init_kernels,
 * conv_direct_depth, conv_seg), which also lets the output be a row slice of a larger tensor. */
int ph_dynconv(const uint16_t* planes, const uint16_t* kern, int64_t kern_plane_stride, int64_t kern_batch_stride,
               const float* kbias, int64_t kbias_batch_stride, uint32_t* bits_out, void* logits_out, int out_dtype,
               int64_t out_batch_stride, int B, int N, int64_t HW, int prec, void* stream);

/* ---- A14: x2 bilinear upsample, align_corners=False (kernel_update.py:131-143); dtype PH_OUT_* ------------- */
int ph_upsample2x(const void* src, void* dst, int dtype, int64_t planes /* B*N */, int H, int W, void* stream);

/* ---- A13 of the final stage + A14 fused (round 4): up_out[b][n] = bilinear x2 (align_corners=False) of the 16-bit logits
 * kern[b][n] . feat[b] + kbias[b][n], from ONE read of the feature plane; logits_out (nullable) additionally receives the
 * low-resolution logits [B][N][H][W] themselves (kernel_update.py:131-143 returns both for the mask branch; the depth
 * branch's low-resolution logits are never returned: kernel_update.py:338-345,401).  Same values as ph_dynconv followed by
 * ph_upsample2x up to fp32 rounding inside the interpolation.  `kern`: ONE 16-bit plane [B][Npad][256] (frame stride given),
 * prec PH_PREC_BF16 (out_dtype PH_OUT_BF16), PH_PREC_F16 or PH_PREC_BF16_KF16 (out_dtype PH_OUT_F16).
 * ph_dynconv_up2_supported: W == 256 (tiles of 64 pixels must not straddle image rows; 2048 / 8), H * W % 128 == 0,
 * 65 <= N <= 224; otherwise callers use the two-kernel form. */
int ph_dynconv_up2_supported(int N, int H, int W, int prec, int out_dtype);
int ph_dynconv_up2(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                   int64_t kbias_batch_stride, void* logits_out /* nullable */, void* up_out, int out_dtype, int B, int N, int H,
                   int W, int prec, void* stream);
/* the same with the launch geometry chosen by the caller (round 6): `workgroups` contiguous ranges of image rows, 0 = one per CU.  A
 * launch that shares the GPU with other streams' kernels ends sooner with 1.5 per CU (engine.DecodePlan: plans marked `shares_gpu`);
 * the values do not depend on it. */
int ph_dynconv_up2_wgs(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                       int64_t kbias_batch_stride, void* logits_out /* nullable */, void* up_out, int out_dtype, int B, int N, int H,
                       int W, int prec, int workgroups, void* stream);

/* ---- A13 of a NON-final stage fused with the x half of the next stage's A7 (round 6): the mask bits of ph_dynconv AND
 * partial[b][split][n][0 .. 255] = sum over the pixels of range `split` of bits[n][px] * x[c][px] -- ph_pool's output for the x map
 * (kernel_update_head.py:317-329, :236-241) -- from ONE read of the x plane.  The caller pools depth_feats alone afterwards
 * (ph_pool_counts with xplanes = the depth plane, dplanes = NULL, partial + 256: columns 256 .. 511 and the pixel counts).
 * `kern`: ONE 16-bit plane [B][Npad][256]; prec PH_PREC_BF16, PH_PREC_F16 or PH_PREC_BF16_KF16; 33 <= N <= 192
 * (ph_dynconv_poolx_supported); grid = nsplit x B workgroups, one per CU when nsplit * B is about the CU count.  The bits are
 * ph_dynconv's bit for bit.  The pixel ranges are ph_pool's for the same nsplit (whole pairs of 64-pixel chunks; an empty range
 * writes zeros), and in the PH_PREC_BF16 / PH_PREC_F16 grades every range's sums are ph_pool's BIT FOR BIT (same operands, same
 * order); in the PH_PREC_BF16_KF16 grade the tile is pooled after its conversion to fp16 (exact for every bf16 value inside
 * fp16's normal range: 1e-6 apart on N(0, 1) data).  Replaces: kernel_update_head.py:317-329 + the x half of :241. */
int ph_dynconv_poolx_supported(int N, int prec);
int ph_dynconv_poolx(const uint16_t* planes, const uint16_t* kern, int64_t kern_batch_stride, const float* kbias,
                     int64_t kbias_batch_stride, uint32_t* bits_out, float* partial, int nsplit, int B, int N, int64_t HW, int prec,
                     void* stream);

/* ---- A1-A5: KernelHead after localization_fpn (kernel_head.py:245-347) ------------------------
 * ph_khead_conv_gn: loc/sem/dfe = ReLU(GN(conv1x1(f0/f1/f2))) and x = sem + loc, from the three fp32
 *   post-neck maps [B][256][HW] to bf16 planes (+ optional fp32 NCHW x_feats / depth_feats).
 *   wplanes: bf16 planes [P][3][256][256] of {loc,seg,depth}_convs.0.conv.weight;
 *   gn_affine: fp32 [3][2][256] (gamma, beta) of the three GroupNorms; eps 1e-5.
 * The remaining 1x1 convs (init_kernels :256, conv_seg :295, conv_direct_depth :285) are ph_dynconv
 * calls with static weights, the object pooling (:314-320) is ph_pool, and
 * ph_khead_proposals forms proposal_feats = [init_kernels.weight + pooled ; conv_seg.weight[stuff]]
 * (:299-300,324-335) as fp32 [B][n_thing_queries + n_stuff][256]. */
size_t ph_khead_workspace_bytes(int B, int64_t HW, int groups);
int ph_khead_conv_gn(const float* f0, const float* f1, const float* f2, const uint16_t* wplanes,
                     const float* gn_affine, int groups, float eps,
                     uint16_t* loc_planes, uint16_t* sem_planes, uint16_t* x_planes, uint16_t* dfe_planes,
                     float* x_f32 /* nullable */, float* dfe_f32 /* nullable */,
                     void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, void* stream);
/* ph_khead_fused: the same two passes with the three static 1x1 convs applied to the normalised tile while it is
 * still in LDS (second GEMM of the apply pass): loc and sem never reach HBM, and the mask / seg / depth logits leave
 * the kernel directly.  Replaces ph_khead_conv_gn + 3 x ph_dynconv + the stuff-row copy (kernel_head.py:250-331).
 *   w2_init / w2_seg / w2_dd: bf16 planes [P][rows32/32][16][64][8] = MFMA 32x32x16 A fragments of
 *     init_kernels.weight [n_init][256], conv_seg.weight [n_seg][256], conv_direct_depth.weight [1][256], rows zero
 *     padded to a multiple of 32 (block (row tile, k-step): lane l holds row (l & 31), k = 16*step + 8*(l >> 5) .. +8);
 *   bias_seg [rows32(n_seg)], bias_dd [32] fp32 (init_kernels has no bias);
 *   mask_preds fp32 [B][n_init + n_stuff][HW]: rows [0, n_init) = init_kernels(loc), rows [n_init, ..) = seg rows
 *     [stuff_lo, stuff_lo + n_stuff) (cat_stuff_mask, :329-331; n_stuff = 0: none);
 *   seg_preds fp32 [B][n_seg][HW]; depth_pred fp32 [B][1][HW]; x_planes / dfe_planes / x_f32 / dfe_f32 as above
 *   (dfe_planes doubles as the scratch that carries loc from the first to the second apply launch).
 *   input_format: PH_IN_F32_NCHW -- f0/f1/f2 are the fp32 maps [B][256][HW] of the reference boundary;
 *     PH_IN_PLANES -- they are bf16 planes [P][B][256][HWp], zero in [HW, HWp) (what ph_gn_apply PH_GN_TO_CPLANES
 *     writes: the neck hands its outputs over at 2 bytes per element; in bf16 precision the results are bit-identical
 *     to feeding the fp32 maps, whose first use is the same rounding).
 *   The mask bits of these logits (use_binary, :310-314) are ph_binarize's: emitting them from this epilogue (ballot per
 *   accumulator register) measured slower than that separate pass. */
int ph_khead_fused(const void* f0, const void* f1, const void* f2, const uint16_t* wplanes,
                   const float* gn_affine, int groups, float eps,
                   const uint16_t* w2_init, int n_init, const uint16_t* w2_seg, const float* bias_seg, int n_seg,
                   const uint16_t* w2_dd, const float* bias_dd, int stuff_lo, int n_stuff,
                   uint16_t* x_planes, uint16_t* dfe_planes, float* x_f32 /* nullable */, float* dfe_f32 /* nullable */,
                   float* mask_preds, float* seg_preds, float* depth_pred,
                   void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, int input_format, void* stream);
/* ph_khead_fused with the logit dtype as an argument (PH_OUT_F32 / PH_OUT_F16) and a device predicate: when run_if is non-NULL
 * and *run_if == 0 at execution time every launch of the call returns at once.  Issued behind ph_khead_onepass with run_if =
 * that call's status word (the first 4 bytes of its workspace) it is the in-call fallback of a one-pass launch that gave up:
 * same buffers, same results to the two forms' summation-order difference, no host round trip, capturable in a HIP graph. */
int ph_khead_fused_if(const void* f0, const void* f1, const void* f2, const uint16_t* wplanes,
                      const float* gn_affine, int groups, float eps,
                      const uint16_t* w2_init, int n_init, const uint16_t* w2_seg, const float* bias_seg, int n_seg,
                      const uint16_t* w2_dd, const float* bias_dd, int stuff_lo, int n_stuff,
                      uint16_t* x_planes, uint16_t* dfe_planes, float* x_f32 /* nullable */, float* dfe_f32 /* nullable */,
                      void* mask_preds, void* seg_preds, void* depth_pred, int logits_dtype, const uint32_t* run_if /* nullable */,
                      void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, int input_format, void* stream);
/* ph_khead_onepass (round 3): ph_khead_fused's results from ONE read of the three maps.  The conv output of a 128-pixel
 * slice stays in the accumulator registers of one workgroup per CU while the GroupNorm sums of the whole (frame, map)
 * are exchanged between the workgroups inside the launch (persistent grid, bounded spins); loc waits in LDS for
 * x = sem + loc; the mask bits of `mask_preds` (kernel_head.py:314-317; what ph_binarize would write) come from the
 * second GEMM's accumulators.  Same arithmetic as ph_khead_fused's PH_PREC_BF16 / PH_PREC_F16 grades.
 *   conv_frags: {loc,seg,depth}_convs.0.conv.weight as MFMA 32x32x16 A fragments [3][8][16][64][8] (one 16-bit plane);
 *   mask_preds / seg_preds / depth_pred: fp32 or fp16 (`out_dtype` PH_OUT_F32 / PH_OUT_F16), shapes as ph_khead_fused;
 *   bits: nullable, uint32 [B][bits_rows][HWp/32], rows >= n_init + n_stuff are cleared;
 *   workspace: ph_khead_onepass_workspace_bytes(B, HW) bytes; the caller zeroes it ONCE after allocation, every call clears all
 *     of it but the last 256 bytes (a memset node ahead of the kernel): those hold the sticky time-out words.
 * ph_khead_onepass_supported: 1 when the geometry fits (HWp / 128 <= #CUs, 32 groups, one-plane grade, HW % 4 == 0 for
 * fp32 inputs); otherwise callers use ph_khead_fused.
 * The persistent grid needs one workgroup resident on every CU it uses.  When that does not happen within the hand-off bound
 * (ph_khead_onepass_set_timeout_us, default 20 ms: another kernel holds CUs that long, or two one-pass launches starve each
 * other) the launch GIVES UP: it raises the per-call status word (first 4 bytes of the workspace), its workgroups leave, and
 * the tensors it was writing are incomplete.  Round 4: the caller issues ph_khead_fused_if + ph_binarize_if predicated on that
 * word directly behind the call (engine.KernelHeadPlan.run does), so the SAME call still ends with the right results -- there
 * is no undefined-result mode, no host synchronisation, and it holds inside HIP graphs.  ph_khead_onepass_status: 1 if the last
 * call gave up; ph_khead_onepass_timeouts: workgroup time-outs since the workspace was zeroed (both synchronise the stream). */
int ph_khead_onepass_supported(int B, int64_t HW, int groups, int prec, int input_format);
size_t ph_khead_onepass_workspace_bytes(int B, int64_t HW);
int ph_khead_onepass(const void* f0, const void* f1, const void* f2, const uint16_t* conv_frags,
                     const float* gn_affine, int groups, float eps,
                     const uint16_t* w2_init, int n_init, const uint16_t* w2_seg, const float* bias_seg, int n_seg,
                     const uint16_t* w2_dd, const float* bias_dd, int stuff_lo, int n_stuff,
                     uint16_t* x_planes, uint16_t* dfe_planes, float* x_f32 /* nullable */, float* dfe_f32 /* nullable */,
                     void* mask_preds, void* seg_preds, void* depth_pred, int out_dtype,
                     uint32_t* bits /* nullable */, int bits_rows,
                     void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec, int input_format, void* stream);
int ph_khead_onepass_status(const void* workspace, int B, void* stream);
int ph_khead_onepass_timeouts(const void* workspace, int B, int64_t HW, void* stream);
void ph_khead_onepass_set_timeout_us(int microseconds /* <= 0: the default */);
/* debugging aid: device buffer [grid][2][3 * rounds][16] uint64 filled with s_memtime stamps by the following launches (NULL: off) */
void ph_khead_onepass_set_timeline(void* buf);
int ph_khead_proposals(const float* partial, int nsplit, const float* w_init /*[Nq][256]*/,
                       const float* w_stuff /*[n_stuff][256]*/, float* proposal_feats,
                       int B, int n_thing_queries, int n_stuff, void* stream);

/* ---- N4 (first part): matching costs of the mask Hungarian assigner (polyphonic/funcs/assigner.py:113-129 DiceCost,
 * :164-194 MaskCost with pred_act = sigmoid, gt_valid as MaskHungarianAssignerWithDepth.assign :478-498 passes it) --
 * every pixel sum of B images in one pass: logits fp32 [B][N][HW] (mask logits, sigmoid applied inside), gt fp32
 * [B][G][HW] (soft masks; pad unused rows with zeros), valid fp32 [B][HW] of 0/1 or NULL.  N <= 256, G <= 127.
 * partial: fp32 [B][ph_match_nsplit(HW, B)][ph_match_record_floats(N, G)]; the caller adds the records of one image in
 * order.  Record (Npad = ph_n_padded(N), Gpad = ph_n_padded(G + 1)):
 *   A[Npad][Gpad] (A[n][g] = sum p t v; column G = S[n] = sum p v) | Q[Npad] = sum p^2 v | C[Gpad] = sum t^2 v |
 *   T[Gpad] = sum t v | V = sum v. */
int64_t ph_match_record_floats(int N, int G);
int ph_match_nsplit(int64_t HW, int B);
int ph_match_sums(const float* logits, const float* gt, const float* valid, float* partial, int B, int N, int G,
                  int64_t HW, void* stream);

/* ---- N4 (training path): the pixel passes of one stage's losses, kernel_update_head.py:355-441 ------------------------------
 * Every *_sums entry writes fixed-order partial records (doubles, one per workgroup) that the caller adds up in index
 * order; every *_grad entry writes d loss / d logits given the coefficients the caller derives from those sums.
 *  mask : positive prediction rows `pos_rows[P]` of pred / target / weight [rows][HW]; out [P][nsplit][5] = sum BCE-with-logits,
 *         pixel count, a = sum sig t, b = sum sig^2, c = sum t^2 over the pixels with weight != 0 (loss_mask: mmdet
 *         cross_entropy_loss.py:74-113, loss_dice: dice_loss.py:9-46).  grad += coef[p][0] (sig - t) + (coef[p][1] t + coef[p][2] sig)
 *         sig (1 - sig) on those pixels.
 *  rank : softmax cross entropy over the N mask channels of each pixel against rank_target [B][HW] (int32, ignore_index),
 *         cross_entropy_loss.py:9-47; out [B * ph_rank_loss_blocks(HW)]; grad (overwritten, all N channels) = scale (softmax - onehot).
 *  depth: DepthLoss (polyphonic/losses/depth_loss.py:9-65) over 0 < target < 80, weight != 0; out [blocks][5] = n, sum lm^2, sum lm,
 *         sum r^2, sum |r| with lm = (log p - log t) w, r = (p - t) w / t, p = depth_act(pred); grad (overwritten) =
 *         ((c0 lm + c1) w / p + (c2 r + c3 sign r) w / t) depth_act'(pred).
 *  focal: py_sigmoid_focal_loss (focal_loss.py:12-60) on pred [R][L], labels [R] (>= L: background), weight [R][L]. */
int ph_mask_loss_sums(const float* pred, const float* target, const float* weight, const int32_t* pos_rows, int P, int64_t HW,
                      int nsplit, double* out, void* stream);
int ph_mask_loss_grad(const float* pred, const float* target, const float* weight, const int32_t* pos_rows, int P, int64_t HW,
                      const float* coef, float* grad, void* stream);
int ph_rank_loss_blocks(int64_t HW);
int ph_rank_loss_sum(const float* pred, const int32_t* rank_target, int B, int N, int64_t HW, int ignore_index, double* out,
                     void* stream);
int ph_rank_loss_grad(const float* pred, const int32_t* rank_target, int B, int N, int64_t HW, int ignore_index, float scale,
                      float* grad, void* stream);
int ph_depth_loss_blocks(int64_t total);
int ph_depth_loss_sums(const float* pred, const float* target, const float* weight, int64_t total, int depth_mode, double* out,
                       void* stream);
int ph_depth_loss_grad(const float* pred, const float* target, const float* weight, int64_t total, int depth_mode, float c0,
                       float c1, float c2, float c3, float* grad, void* stream);
int ph_focal_loss_blocks(int64_t total);
int ph_focal_loss_sum(const float* pred, const int64_t* labels, const float* weight, int64_t R, int L, float gamma, float alpha,
                      double* out, void* stream);
int ph_focal_loss_grad(const float* pred, const int64_t* labels, const float* weight, int64_t R, int L, float gamma, float alpha,
                       float scale, float* grad, void* stream);
/* the same focal loss over a class-major map: pred [B][L][HW], target int32 [B][HW] (== L: pixel not selected) -- KernelHead's
 * loss_rpn_seg, kernel_head.py:538-551; out [B * ph_rank_loss_blocks(HW)], grad [B][L][HW] (overwritten) */
int ph_seg_focal_sum(const float* pred, const int32_t* target, int B, int L, int64_t HW, float gamma, float alpha, double* out,
                     void* stream);
int ph_seg_focal_grad(const float* pred, const int32_t* target, int B, int L, int64_t HW, float gamma, float alpha, float scale,
                      float* grad, void* stream);

/* target assembly of the losses above.  rank_target: out[b][p] = the last j with pos[b*N + j] and mask_targets[b*N + j][p] != 0,
 * else ignore_index (kernel_update_head.py:420-432).  seg_target (one image): L, then sem_cls[s] where sem_seg[s] != 0 in order,
 * then pos_labels[i] where pos_masks[i] != 0 in order (kernel_head.py:590-605). */
int ph_rank_target(const float* mask_targets, const uint8_t* pos, int B, int N, int64_t HW, int ignore_index, int32_t* out,
                   void* stream);
int ph_seg_target(const float* sem_seg, const int64_t* sem_cls, int S, const float* pos_masks, const int64_t* pos_labels, int P,
                  int L, int64_t HW, int64_t* out, void* stream);

/* DepthCost (funcs/assigner.py:17-80): out[n][g] = { sum lm^2, sum lm, sum r^2, sum |r| } over the pixels with
 * gt_depth * gt_masks[g] > 0, lm = log(depth_act(z_n) + eps) - log(gt_depth * gt_masks[g] + eps), r = (d - t) / t;
 * nvalid[g] = number of those pixels.  depth_mode: 0 'sigmoid', 1 'monodepth' (funcs/depth_utils.py). */
int ph_depth_cost_sums(const float* depth_logits, const float* gt_depth, const float* gt_masks, int N, int G, int64_t HW,
                       int depth_mode, float eps, float* out, float* nvalid, void* stream);

/* ---- N4, backward: the map-sized products on fp32 NCHW maps (csrc/ph_train.hip); hi+lo bf16 MFMA, fp32 grade ----------------
 * rows_x_map : Y[b][m][p] = sum_k A[b][m][k] X[b][k][p].  A [B or 1][Mpad][lda] zero padded (Mpad % 16 == 0, lda % 8 == 0),
 *              a_batch_stride in elements (0: one A for every image).  binarize_x: X is used as (X > 1.5 * 2^-24) ? 1 : 0.
 *              Replaces F.conv2d with 1x1 static / dynamic kernels (kernel_head.py:250-295, kernel_update_head.py:317-329)
 *              and autograd's grad_input of those and of the einsum pooling (kernel_update_head.py:241-242).
 * map_x_mapT : O[b][m][k] = sum_p G[b][m][p] X[b][k][p], K <= 256.  nsplit from ph_map_x_map_t_nsplit; partial holds
 *              B * nsplit * M * K floats, summed in a fixed order.  binarize_g as above (the hard-mask pooling forward).
 * upsample2x_bwd : transpose of ph_upsample2x (F.interpolate x2 bilinear, align_corners=False). */
int ph_rows_x_map(const float* A, int64_t a_batch_stride, int lda, int Mpad, int M, int K, const float* X, float* Y, int B, int64_t HW,
                  int binarize_x, void* stream);
int ph_map_x_map_t_nsplit(int B, int M, int64_t HW);
int ph_map_x_map_t(const float* G, const float* X, float* partial, float* out, int B, int M, int K, int64_t HW, int nsplit,
                  int binarize_g, void* stream);
int ph_upsample2x_bwd(const float* grad_out, float* grad_in, int64_t planes, int H, int W, void* stream);
/* round 5: _ex forms.  rows_x_map: bias [B][M] (nullable) added to every pixel of row m -- the scalar bias of a folded dynamic
 * kernel, kernel_update_head.py:317-329 --, add [B][M][HW] (nullable, may be Y itself): Y = A X + add (a second gradient
 * contribution lands where the first one lies).
 * map_x_mapT: rowsum [B][M] = sum_p G[b][m][p] (binarised: the pixel count of each hard mask; otherwise the gradient of a dynamic
 * kernel's scalar bias), rs_partial [B][nsplit][M] its per-split scratch; both null or both given.
 * hard_count: out[r] = #{p : sigmoid(logits[r][p]) > 0.5}. */
int ph_rows_x_map_ex(const float* A, int64_t a_batch_stride, int lda, int Mpad, int M, int K, const float* X, float* Y, int B,
                     int64_t HW, int binarize_x, const float* bias, const float* add, void* stream);
int ph_map_x_map_t_ex(const float* G, const float* X, float* partial, float* out, int B, int M, int K, int64_t HW, int nsplit,
                      int binarize_g, float* rs_partial, float* rowsum, int sum_batch, void* stream);
/* GroupNorm + ReLU of a ConvModule in training mode, fp32 NCHW (csrc/ph_gntrain.hip; kernel_head.py:250-278, semantic_fpn.py:75-178).
 * fwd: out = relu(GN(y)) (nullable when out_sum is given); out_sum = out + add (both null or both given: KernelHead's
 *      x_feats = sem + loc, the neck's sum over its towers); stats [B][groups][2]
 *      (mean, rstd) kept for the backward; partial: B * groups * ph_gn_train_nsplit(HW, C / groups) * 2 doubles of scratch.
 * bwd: dy = dyA (+ dyB) masked by out > 0 (recomputed from y); dx, dgamma [C], dbeta [C] are overwritten;
 *      partial: B * C * ph_gn_train_bwd_nsplit(HW) * 2 doubles of scratch.  sum_batch != 0 in ph_map_x_map_t_ex: out [M][K] and
 *      rowsum [M] are summed over the images as well (the gradient of a static 1x1 kernel and of its bias). */
/* 3x3 convolutions of SemanticFPNWrapper's towers in training (fp32 NCHW, csrc/ph_train.hip; semantic_fpn.py:75-150):
 * conv3x3_taps : W [M][K][3][3] <-> tap-major [9][M][K] (transpose: [9][K][M]; flip: tap 8 - t -- together the operand of the input
 *                gradient of a stride-1 conv); to_weight != 0 runs the inverse (a tap-major weight gradient back to [M][K][3][3]).
 * conv3x3_train: Y [B][M][Ho][Wo] = sum_tap taps[tap] X[.., src_tap(.)], X [B][K][Hi][Wi]; mode 0: forward with `stride` (also the
 *                input gradient of a stride-1 conv with flipped / transposed taps), mode 1: input gradient of the stride-2 conv
 *                (X = dL/dY [Hi][Wi], Y = dL/dX [Ho][Wo], taps transposed, NOT flipped).
 * conv3x3_wgrad: tap-major dW [9][M][K] summed over the batch; partial: B * nsplit * M * K floats (nsplit: ph_map_x_map_t_nsplit). */
int ph_conv3x3_taps(const float* W, float* taps, int M, int K, int transpose, int flip, int to_weight, void* stream);
int ph_conv3x3_train(const float* taps, int M, int K, const float* X, float* Y, int B, int Hi, int Wi, int Ho, int Wo, int stride,
                     int mode, void* stream);
int ph_conv3x3_wgrad(const float* dY, const float* X, float* partial, float* taps_out, int B, int M, int K, int Hi, int Wi, int Ho,
                     int Wo, int stride, int nsplit, void* stream);
int ph_gn_train_nsplit(int64_t HW, int cpg);
int ph_gn_train_bwd_nsplit(int64_t HW);
int ph_gn_train_fwd(const float* y, const float* gamma, const float* beta, int groups, float eps, const float* add, float* out,
                    float* out_sum, float* stats, double* partial, int B, int C, int64_t HW, void* stream);
int ph_gn_train_bwd(const float* y, const float* stats, const float* gamma, const float* beta, int groups, const float* dyA,
                    const float* dyB, float* dx, float* dgamma, float* dbeta, double* partial, int B, int C, int64_t HW, void* stream);
int ph_hard_count(const float* logits, float* out, int64_t rows, int64_t HW, void* stream);

/* ---- N4: targets + losses + d(losses)/d(predictions) of one head / stage in ONE call, from DESCRIPTORS (csrc/ph_loss.hip) ------
 * Replaces `get_targets` + `loss` of both heads on the training path (kernel_update_head.py:355-591, kernel_head.py:456-698): every
 * target / weight row the reference materialises is a ground-truth mask, the image's valid map, its depth map, ones or zeros,
 * so rows are described by device POINTERS (int64 addresses of fp32 rows of HW pixels):
 *   pos_rows int32 [P] (rows with a label in [0, L)), pos_u8 [B*N]; tptr / wptr int64 [B*N]: mask target / weight row (0 = zeros);
 *   depth items grouped per prediction row: dstart int32 [depth_rows + 1], dit_t (target row), dit_w (weight row; 1 = ones),
 *   dit_s (scale; the weight is scale * row, and the reference's `* (gt_depth > 0)` is the kernel's own `0 < target < 80`);
 *   labels int64 [B*N], label_w fp32 [B*N][L] (nullable together with cls_score: KernelHead has no classification);
 *   KernelHead only: seg_pred [B][seg_L][HW] with the paint lists of the dense semantic target (kernel_head.py:590-605):
 *   sstart int32 [B + 1], sit_m (mask row), sit_l (class), painted in order over background = seg_L.
 * mask_pred [B][N][HW]; depth_pred [depth_rows][HW] (KernelHead: the ONE direct depth map per image, all its items summed).
 * losses [8] (device): loss_depth, loss_cls, mask BCE, dice, rank, seg focal, pos_acc, number of labelled seg pixels -- weighted as
 * the reference's loss modules weight them.  g_* nullable all together: d(sum of the losses)/d(prediction), overwritten.
 * Loss values and gradient coefficients are formed on the device (fp64, fixed order): no host round trip inside the call. */
typedef struct {
    int32_t B, N, L, P, depth_rows, seg_L, has_rank, ignore, depth_mode;
    int64_t HW;
    float lw_mask, lw_dice, dice_eps, lw_rank, lw_depth, dw_si, dw_sq, dw_abs, lw_cls, cls_gamma, cls_alpha, cls_avg, lw_seg, seg_gamma,
        seg_alpha;
} ph_loss_cfg;
size_t ph_train_losses_scratch_bytes(const ph_loss_cfg* cfg);
int ph_train_losses(const ph_loss_cfg* cfg, const float* mask_pred, const float* cls_score, const float* depth_pred,
                    const float* seg_pred, const int32_t* pos_rows, const uint8_t* pos_u8, const int64_t* tptr, const int64_t* wptr,
                    const int32_t* dstart, const int64_t* dit_t, const int64_t* dit_w, const float* dit_s, const int64_t* labels,
                    const float* label_w, const int32_t* sstart, const int64_t* sit_m, const int32_t* sit_l, float* losses,
                    float* g_mask, float* g_cls, float* g_depth, float* g_seg, void* scratch, size_t scratch_bytes, void* stream);

/* ---- N4, device half: the QUERY SIDE of one KernelUpdateHead stage in TRAINING mode (csrc/ph_qtrain.hip) -------------------
 * Replaces what autograd records for kernel_update_head.py:245-288 (KernelUpdator x2, funcs/kernel_updator.py:55-93; mmcv
 * MultiheadAttention + LayerNorm x2; FFN + LayerNorm x2; cls_fcs / mask_fcs / depth_regs + fc_cls / fc_mask / fc_depth) and its
 * backward.  Rows r = b * N + n of 256 features; two branches (0 mask, 1 depth).  fp32 in, fp32 out, fp32 MFMA products.
 * params: HOST array [2][PH_QTRAIN_NPARAM] of DEVICE pointers to the fp32 parameters as nn.Parameter stores them ([out][in]
 * weights), per branch in this order (mask-branch names; the depth branch: *_depth, depth_regs / fc_depth for 34-38, 39-43 null):
 *   0 feat_transform.conv.weight  1 .bias | kernel_update_conv: 2 dynamic_layer.weight 3 .bias 4 input_layer.weight 5 .bias
 *   6 input_gate.weight 7 .bias 8 update_gate.weight 9 .bias 10 input_norm_in.weight 11 .bias 12 norm_in.weight 13 .bias
 *   14 norm_out.weight 15 .bias 16 input_norm_out.weight 17 .bias 18 fc_layer.weight 19 .bias 20 fc_norm.weight 21 .bias |
 *   22 attention.attn.in_proj_weight 23 in_proj_bias 24 out_proj.weight 25 out_proj.bias 26 attention_norm.weight 27 .bias |
 *   28 ffn.layers.0.0.weight 29 .bias 30 ffn.layers.1.weight 31 .bias 32 ffn_norm.weight 33 .bias |
 *   34 mask_fcs.0.weight 35 mask_fcs.1.weight 36 mask_fcs.1.bias 37 fc_mask.weight 38 fc_mask.bias |
 *   39 cls_fcs.0.weight 40 cls_fcs.1.weight 41 cls_fcs.1.bias 42 fc_cls.weight 43 fc_cls.bias
 * forward : pooled [2][R][256] = the hard-mask pooling of x / depth_feats BEFORE feat_transform (folded: pooling is linear in it),
 *           cnt [R] = pixels of each hard mask, k [R][256] kernels, q [R][256] depth kernels (the kernel k.detach() is added inside,
 *           :250).  Writes cls [R][L], kern [2][R][256] = fc_mask / fc_depth outputs folded with feat_transform's weight,
 *           kbias [2][R] = their product with its bias, obj [2][R][256] = the updated kernels; `saved` (ph_qtrain_saved_floats)
 *           keeps the pre-normalisation rows, activations and attention probabilities the backward needs.
 * backward: gradients w.r.t. cls, kern, kbias, obj in; writes grads [2][PH_QTRAIN_NPARAM] (HOST array of device pointers, each
 *           the size of its parameter; every one is overwritten, none accumulated), g_pooled [2][R][256], g_k, g_q [R][256].
 *           `scratch`: ph_qtrain_scratch_floats.  Fixed summation orders throughout (no atomics). */
/* ph_gemm32: one product of the training side's fp32-MFMA tile GEMM on its own (tests, timing): C [M][N] = A(m,k) B(k,n) (+ bias[n]);
 * kcA / kcB != 0: the operand is k-contiguous (X [row][k] / W [out][in]), else row-contiguous ([k][row]); ksplit > 1: C [ksplit][M][ldc] */
int ph_gemm32(const float* A, int lda, int kcA, const float* B, int ldb, int kcB, float* C, int ldc, int M, int N, int K, int ksplit,
              const float* bias, void* stream);
#define PH_QTRAIN_NPARAM 44
size_t ph_qtrain_saved_floats(int B, int N, int L, int F);
size_t ph_qtrain_scratch_floats(int B, int N, int L, int F);
int ph_qtrain_forward(const float* const* params, const float* pooled, const float* cnt, const float* k, const float* q, float* cls,
                      float* kern, float* kbias, float* obj, float* saved, int B, int N, int L, int F, void* stream);
int ph_qtrain_backward(const float* const* params, const float* pooled, const float* cnt, const float* k, const float* q,
                       const float* saved, const float* g_cls, const float* g_kern, const float* g_kbias, const float* g_obj,
                       float* const* grads, float* g_pooled, float* g_k, float* g_q, float* scratch, int B, int N, int L, int F,
                       void* stream);

/* ---- A16-A18: panoptic merge (kernel_update.py:421-535, kernel_update_head.py:593-626) -------
 * geom = {sh, sw, Hb, Wb, h, w, Ho, Wo}: stride-4 source size, batch_input_shape, img_shape, ori_shape.
 * activate: act_mask[k] = sigmoid(mask_up[q_idx[k]]), act_depth[k] = depth_act(depth_up[q_idx[k]]),
 *           act_depth0 = depth_act(depth_init_up); logits fp32 or bf16 ([N][sh][sw]), outputs fp32.
 * argmax  : ids[px] = first argmax_k scores[k] * rescale(act_mask[k])(px); counts[0][k] = #{ids == k},
 *           counts[1][k] = #{rescale(act_mask[k]) >= 0.5} (counts zeroed by the call).
 * paste   : pan = newid[ids]; depth_final = newid[ids] > 0 ? rescale(act_depth[ids]) : rescale(act_depth0).
 * from_probs != 0: act_* are already full-resolution [K][Ho][Wo] maps (no resampling) -- the integer
 * semantics in isolation. The accept loop between argmax and paste (:500-533) is host logic. */
/* select  : the segment candidates of kernel_update.py:428-434 / :448-459 on the device, for B frames: the top max_per_img
 *           (query, thing class) pairs of cls_scores [B][N][L] (post-sigmoid) in descending order, then the stuff queries'
 *           own-class scores (the diagonal of the [N - num_proposals] x [L - num_thing_classes] block) in descending order;
 *           q_idx / labels / scores [B][out_batch_stride], K = max_per_img + stuff entries each.  Ties: ascending index. */
int ph_panoptic_select(const float* cls_scores, int64_t cls_batch_stride, int B, int N, int L, int num_proposals,
                       int num_thing_classes, int max_per_img, int32_t* q_idx, int32_t* labels, float* scores,
                       int64_t out_batch_stride, void* stream);
int ph_panoptic_activate(const void* mask_up, const void* depth_up, int dtype, const float* depth_init_up,
                         const int32_t* q_idx, int K, int h2, int w2, int depth_mode /*0 sigmoid, 1 monodepth*/,
                         float* act_mask, float* act_depth, float* act_depth0, void* stream);
int ph_panoptic_argmax(const float* act_mask, const float* scores, int K, const int32_t* geom, int from_probs,
                       int32_t* ids, int32_t* counts /*[2][K]*/, void* stream);
int ph_panoptic_paste(const int32_t* ids, const int32_t* newid, const float* act_depth, const float* act_depth0,
                      const int32_t* geom, int from_probs, int32_t* pan, float* depth_basic, float* depth_final,
                      void* stream);

/* ---- device side of the video association step (polyphonic_former_video.py:359-396) ----------
 * ph_segment_boxes : int32 panoptic id map [H][W], segment ids 1..nseg -> rois [nseg][5] = (0, x1, y1, x2, y2)
 *                    (centre +- 2 * mean |deviation| per axis, clamped at 0: polyphonic/video/utils.py:39-83,
 *                    polyphonic_former_video.py:413-415) and tight extent boxes [nseg][4] (funcs/utils.py:4-22).
 * ph_roi_align_fpn : mmdet SingleRoIExtractor (level = floor(log2(sqrt(area)/56 + 1e-6))) + mmcv RoIAlign(7,
 *                    sampling_ratio 2, avg, aligned) over `nlev` fp32 maps [1][256][H_l][W_l] (feats = host array of
 *                    device pointers, hw = {H_0, W_0, H_1, ...}, scales = 1/stride) -> channels-last bf16 planes
 *                    [P][n][49][256] (+ optional fp32 [n][256][7][7]).
 * ph_gemm_rows     : Y[M][N] = act(X[M][K] W^T + b), X bf16 planes [P][M][K], W packed B fragments (plane stride
 *                    given), Y fp32 and/or bf16 planes; the track head's conv3x3-as-GEMM, fc and fc_embed
 *                    (polyphonic/video/track_heads.py:92-102).
 * ph_im2col7       : channels-last [P][n][49][256] -> 3x3/pad-1 patches [P][n*49][2304], K order (tap, channel).
 * ph_gn_relu_cl    : per-RoI GroupNorm + ReLU of fp32 [n*49][256] -> channels-last bf16 planes. */
size_t ph_segment_boxes_workspace_bytes(int nseg);
int ph_segment_boxes(const int32_t* pan, int H, int W, int nseg, float* rois, float* ext_boxes,
                     void* workspace, size_t workspace_bytes, void* stream);
int ph_roi_align_fpn(const float* const* feats, const int32_t* hw, const float* scales, int nlev, const float* rois,
                     int n, float finest_scale, uint16_t* out_cl, float* out_f32 /* nullable */, int prec, void* stream);
int ph_gemm_rows(const uint16_t* X, const uint16_t* Wp, int64_t w_plane_elems, const float* bias /* nullable */, int relu,
                 float* Yf /* nullable */, uint16_t* Yp /* nullable */, int M, int N, int K, int prec, void* stream);
int ph_im2col7(const uint16_t* in, uint16_t* out, int n, int prec, void* stream);
/* ph_gemm_rows for the association step's small M (a dozen RoIs): K split over several hundred workgroups, partial sums in
 * `workspace`, added in a fixed order (deterministic; the split depends on M, N, K only).  im2col7 = 1: X is the channels-last
 * [P][M / 49][49][256] maps and the 3x3 / pad-1 patches are gathered by the operand loads (K = 2304; what ph_im2col7 +
 * ph_gemm_rows compute through a materialised matrix). */
size_t ph_gemm_rows_workspace_bytes(int M, int N, int K);
int ph_gemm_rows_splitk(const uint16_t* X, int im2col7, const uint16_t* Wp, int64_t w_plane_elems, const float* bias /* nullable */,
                        int relu, float* Yf /* nullable */, uint16_t* Yp /* nullable */, int M, int N, int K, int prec,
                        void* workspace, size_t workspace_bytes, void* stream);
/* Tracker affinity (quasi_dense_embed_tracker.py:165-182) on the device, next to the embeddings: score[n][m] between the n
 * detections (emb fp32 [n][256], labels int32 [n]) and the m memory columns (memo_emb fp32 [m][256], memo_labels int32 [m]);
 * metric 0 = bisoftmax, 1 = softmax, 2 = cosine; with_cats: zero where the labels differ.  n <= 128, m <= 4096.  The greedy
 * assignment that consumes the matrix is host logic (video.QuasiDenseEmbedTracker). */
size_t ph_track_affinity_workspace_bytes(int n, int m);
int ph_track_affinity(const float* emb, const int32_t* labels, const float* memo_emb, const int32_t* memo_labels, int n, int m,
                      int metric, int with_cats, float* score, void* workspace, size_t workspace_bytes, void* stream);
int ph_gn_relu_cl(const float* y, const float* gamma, const float* beta, int groups, float eps, uint16_t* out, int n,
                  int prec, void* stream);

/* ---- the step before the path, SemanticFPNWrapper.forward (polyphonic/funcs/semantic_fpn.py:198-235,
 * configs/_base_/models/polyphonic_former.py:78-96).  Inside the neck every map is channels-last (NHWC):
 * ph_nhwc_ingest : fp32 NCHW [B][256][HW] (+ add[256][HW], nullable: SinePositionalEncoding on level 3, :202-208)
 *                  -> bf16 NHWC planes [P][B][HW][256].
 * ph_conv_nhwc   : ConvModule's conv (no bias): KSxKS, pad KS/2, stride 1 or 2 (3x3) / 1 (1x1), 256 -> 256 channels;
 *                  X bf16 NHWC planes, Wp = pack.pack_b32 fragments of W[n][tap * 256 + c], Y fp32 NHWC [B][Ho][Wo][256];
 *                  partial = per-channel (sum, sum of squares) of every (row pair, column tile) [B][nwg][256][2]
 *                  (ph_conv_nhwc_partial_floats) for the GroupNorm that follows.
 * ph_gn_finalize : partial -> stats [B][groups][2] = (mean, rstd), fp64 combine (also used by ph_khead_conv_gn).
 * ph_gn_apply    : GroupNorm affine + ReLU on fp32 NHWC (stats == NULL: plain copy/convert), then per `mode`:
 *                  bf16 NHWC planes | x2 bilinear (align_corners=False, nn.Upsample :131-134) bf16 NHWC planes |
 *                  fp32 NHWC accumulate | fp32 NCHW (what KernelHead takes; LDS transpose).
 * ph_gn_sum_planes: sum over `nlev` <= 4 levels of ReLU(GroupNorm(y_l)) -> bf16 NHWC planes (the sum over levels, :221,
 *                  without an fp32 sum buffer); ys / stats / gammas / betas are HOST arrays of `nlev` device pointers. */
int ph_nhwc_ingest(const float* src, const float* add /* nullable */, uint16_t* dst, int B, int64_t HW, int prec, void* stream);
size_t ph_conv_nhwc_partial_floats(int B, int Ho, int Wo);                     /* upper bound, any instantiation */
/* `nwg` for ph_gn_finalize = entries per frame of ph_conv_nhwc's `partial` output: one per (pair of output rows, 64-pixel column
   tile) whatever tile form the launch takes (round 6: batch-invariant partial sums; rounds 3-5 it was the workgroups of the
   chosen form, hence the arguments) */
int ph_conv_nhwc_workgroups_b(int ksize, int stride, int Ho, int Wo, int prec, int B);
int ph_conv_nhwc(const uint16_t* X, const uint16_t* Wp, int64_t w_plane_elems, float* Y, float* partial, int ksize, int stride,
                 int B, int H, int W, int prec, void* stream);
int ph_gn_finalize(const float* partial, float* stats, int nwg, int groups, int64_t HW, float eps, int B, void* stream);
int ph_gn_sum_planes(const float* const* ys, const float* const* stats, const float* const* gammas, const float* const* betas,
                     int nlev, int groups, uint16_t* planes, int B, int64_t HW, int prec, void* stream);
/* ph_gn_sum_cplanes: ph_gn_sum_planes with the result as CHANNEL planes [P][B][256][HWp] (zero in [HW, HWp)), the input format of
 * ph_neck_out_convs: conv_pred + the two aux convs (semantic_fpn.py:156-178,223-231; each 1x1 conv + GN + ReLU of the level sum)
 * in two passes over that sum -- statistics by recomputation, then normalise + ReLU + store -- without an fp32 conv output in
 * memory.  in_channels_last != 0: the sum is ph_gn_sum_planes' [P][B][HW][256] instead (a 64-pixel tile is 32 KiB of consecutive
 * bytes and its B fragments are plain 16-byte LDS reads: no transposition anywhere -- the form NeckPlan uses).
 * wplanes: 16-bit planes [P][3][256][256] (out, in) of the three conv weights; gn_affine fp32 [3][2][256];
 * out_planes_m: 16-bit planes [P][B][256][HWp] and / or out_f32_m: fp32 NCHW [B][256][HW], at least one per map;
 * workspace: ph_neck_out_convs_workspace_bytes(B, HW, groups).  Up to 3 frames per launch a frame's outputs are bit-identical to
 * those of a one-frame launch (tile runs and summation order do not depend on B). */
size_t ph_neck_out_convs_workspace_bytes(int B, int64_t HW, int groups);
int ph_gn_sum_cplanes(const float* const* ys, const float* const* stats, const float* const* gammas, const float* const* betas,
                      int nlev, int groups, uint16_t* planes, int B, int64_t HW, int prec, void* stream);
int ph_neck_out_convs(const uint16_t* in_planes, int in_channels_last, const uint16_t* wplanes, const float* gn_affine, int groups,
                      float eps, uint16_t* out_planes0, uint16_t* out_planes1, uint16_t* out_planes2, float* out_f32_0,
                      float* out_f32_1, float* out_f32_2, void* workspace, size_t workspace_bytes, int B, int64_t HW, int prec,
                      void* stream);
int ph_gn_apply(const float* y, const float* stats /* nullable */, const float* gamma, const float* beta, int groups, int mode,
                int accumulate, uint16_t* planes /* nullable */, float* outf /* nullable */, int B, int H, int W, int prec,
                void* stream);

/* ---- N1: the quasi-dense embedding tracker as a native object (csrc/ph_tracker.hip; polyphonic/video/qdtrack/trackers/
 * quasi_dense_embed_tracker.py:47-207).  Host bookkeeping in C++ (boxes, labels, ids, ages), embeddings in a device POOL of
 * `capacity` rows of 256 floats inside `device_mem` (ph_tracker_device_bytes; owned by the caller, alive as long as the tracker).
 * match: boxes [n][5] (x1, y1, x2, y2, score) and labels [n] on the HOST, embeds [n][256] on the DEVICE; writes the kept detections
 * in descending-score order -- kept_out [k] (indices into the input), ids_out [k] (>= 0 track id, -1 unmatched, -2 suppressed) -- and
 * updates the memory.  Returns k >= 0 or a negative error.  One stream synchronisation inside (the [n x m] score download).
 * thresholds are compared in fp32 like the reference's tensors; memo_momentum / one_minus_momentum: fp32(m) and fp32(1 - m) with
 * 1 - m evaluated in double (what `(1 - momentum) * tensor` does). */
typedef struct {
    float init_score_thr, obj_score_thr, match_score_thr, memo_momentum, one_minus_momentum, nms_conf_thr, nms_backdrop_iou_thr,
        nms_class_iou_thr;
    int32_t memo_tracklet_frames, memo_backdrop_frames, with_cats, metric;   /* metric: 0 bisoftmax, 1 softmax, 2 cosine */
} ph_tracker_cfg;
typedef struct ph_tracker ph_tracker;
size_t ph_tracker_device_bytes(int capacity, int max_dets);
ph_tracker* ph_tracker_create(const ph_tracker_cfg* cfg, void* device_mem, size_t device_bytes, int capacity, int max_dets);
void ph_tracker_destroy(ph_tracker* t);
void ph_tracker_reset(ph_tracker* t);
int64_t ph_tracker_num_tracklets(const ph_tracker* t);
int ph_tracker_rows(const ph_tracker* t);
void ph_tracker_debug_times(const ph_tracker* t, double* out6);   /* accumulated host seconds per phase of `match` (csrc/ph_tracker.hip) */
int ph_tracker_match(ph_tracker* t, const float* boxes, const int64_t* labels, const float* embeds_dev, int n, int64_t frame_id,
                     int32_t* kept_out, int64_t* ids_out, void* stream);
/* a whole step's frames in one call (`video.replay_tracking` after the all-gather): frame f's rows of boxes / labels / kept_out /
   ids_out start at sum(counts[0..f-1]); embeds_dev[f]: that frame's [counts[f]][256] device rows; frames without detections are
   skipped and do not advance the frame counter (polyphonic_former_video.py:391-402).  Returns the number of frames matched. */
int ph_tracker_match_frames(ph_tracker* t, const float* boxes, const int64_t* labels, const float* const* embeds_dev, const int32_t* counts,
                            int nframes, int64_t first_frame_id, int32_t* kept_out, int64_t* ids_out, int32_t* kept_counts, void* stream);

/* ---- self tests of the gfx950 fragment layouts the kernels rely on (tests/test_gpu_selftest.py) */
int ph_selftest_mfma16(const uint16_t* a /*[16][32]*/, const uint16_t* bt /*[16][32]*/, float* d /*[16][16]*/, void* stream);
int ph_selftest_mfma32(const uint16_t* a /*[32][16]*/, const uint16_t* bt /*[32][16]*/, float* d /*[32][32]*/, void* stream);
int ph_selftest_readbw(const void* p, int64_t bytes, int blocks, void* out /*4 B*/, void* stream);   /* streaming-read yardstick */
int ph_selftest_trread(const uint16_t* src /*[16][16]*/, uint16_t* out /*[64][4]*/, void* stream);
/* holds `blocks` workgroup slots with `lds_bytes` of LDS each for `microseconds` (a CU-hogging neighbour for the time-out tests) */
int ph_selftest_hog(int blocks, int lds_bytes, int microseconds, void* scratch4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POLYHEAD_H_ */
