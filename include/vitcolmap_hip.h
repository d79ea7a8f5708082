/*
 * vitcolmap_hip.h — C ABI of libvitcolmap_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary for the hot path of randyjhc/vit-colmap: everything the reference
 * computes between "ViT patch tokens" and "rows in the COLMAP database", plus the exhaustive
 * descriptor matcher.  The reference has no FFI of its own (it is pure Python calling torch and
 * pycolmap), so each entry point cites the reference *method* it replaces; INTEGRATION.md shows
 * the ctypes stub a maintainer of the reference would add at that call site.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller unless marked [host].
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *     work on it; they never synchronise, allocate or free (safe under hipGraph capture).
 *   - Return value: VC_OK (0) or a negative VC_ERR_* code.  No exceptions cross the boundary,
 *     there is no global mutable state, and calls on distinct streams are independent.
 *   - Layouts are row-major, densely packed.  "tokens" = (images, H*W, C) float32, i.e. the
 *     ViT's own output order (token index = y*W + x, reference vit_extractor.py:150-156 builds
 *     its (C,H,W) view from exactly this tensor).
 */
#ifndef VITCOLMAP_HIP_H_
#define VITCOLMAP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_ABI_VERSION 1

#define VC_OK 0
#define VC_ERR_INVALID_ARG (-1) /* null pointer, negative size, misaligned buffer            */
#define VC_ERR_UNSUPPORTED (-2) /* size outside what the kernels cover (see each function)   */
#define VC_ERR_LAUNCH (-3)      /* hipLaunchKernel reported an error (see vc_last_hip_error) */
#define VC_ERR_WORKSPACE (-4)   /* caller's workspace is too small                           */

#define VC_MAX_KEYPOINTS 2048 /* rows per image the matcher kernels accept                  */
#define VC_MAX_DESC_DIM 1024  /* descriptor bytes per row (255^2 * D must stay below 2^26)  */

typedef void* vc_stream_t;

int vc_abi_version(void);
const char* vc_status_string(int status);
/* hipError_t of the last failed launch on the calling thread (0 if none). */
int vc_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * Matcher — replaces pycolmap.match_exhaustive's per-pair arithmetic
 * (reference call site: vit_colmap/pipeline/run_pipeline.py:351-363; options
 * vit_colmap/utils/config.py:64-96).  Specification: oracle/matcher_oracle.py.
 * ------------------------------------------------------------------------------------------ */

/* Bytes of the MFMA-ready ("prepared") copy of n_images descriptor blocks of n_max x d uint8. */
size_t vc_prepared_bytes(int n_images, int n_max, int d);

/*
 * Re-tile uint8 descriptor blocks for the matcher: desc [n_images][n_max][d] uint8,
 * counts [n_images] int32 (valid rows per image, 0..n_max) -> prepared (vc_prepared_bytes).
 * Rows >= counts[i] are neutralised (similarity 0 with everything).
 * Limits: 1 <= n_max <= VC_MAX_KEYPOINTS, 1 <= d <= VC_MAX_DESC_DIM.
 */
int vc_prepare_descriptors(const uint8_t* desc, const int32_t* counts, int n_images, int n_max,
                           int d, void* prepared, vc_stream_t stream);

/*
 * Match image pairs: for every p < n_pairs, images a = pairs[2p], b = pairs[2p+1]:
 * int32 similarities, per-row and per-column best / second best (lowest index wins ties),
 * angle + ratio tests, optional cross check; writes the matches ordered by row index to
 * out_matches[p][0..out_counts[p]) as (row in a, row in b) uint32 pairs.
 * out_matches: [n_pairs][n_max][2] uint32; out_counts: [n_pairs] int32.
 * out_counts[p] == VC_COUNT_SELFCHECK_FAILED (-1): the persistent kernel's consistency check failed for that pair — the
 * waves of its workgroup disagreed on the tile-ring cursors, which the source makes identical by construction (a
 * toolchain fault of this kind is documented in profiles/r03_matcher_looped_miscompile.md); the pair's list is undefined.
 * Never produced by a correct build: callers treat it as an error (the Python host raises HipLibraryError).
 */
#define VC_COUNT_SELFCHECK_FAILED (-1)
int vc_match_pairs_u8(const void* prepared, const int32_t* counts, int n_images, int n_max, int d,
                      const int32_t* pairs, int n_pairs, float max_ratio, float max_distance,
                      int cross_check, uint32_t* out_matches, int32_t* out_counts,
                      vc_stream_t stream);

/*
 * One-way search on raw descriptors: for each of the n1 rows of d1, index of the most similar
 * of the n2 rows of d2 (-1 if every similarity is 0), its similarity and the runner-up's.
 * workspace: at least vc_knn_workspace_bytes(n1, n2, d) bytes, 16-byte aligned.
 */
size_t vc_knn_workspace_bytes(int n1, int n2, int d);
int vc_knn_top2_u8(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int d, int32_t* out_idx,
                   int32_t* out_best, int32_t* out_second, void* workspace,
                   size_t workspace_bytes, vc_stream_t stream);

/*
 * Angle / ratio tests and cross check on two one-way results (12 = rows of image 1 against
 * image 2, 21 = the transpose).  out_pairs: [n1][2] uint32, out_count: [1] int32.
 * With cross_check == 0 the *21 pointers may be NULL.
 */
int vc_mutual_ratio(const int32_t* idx12, const int32_t* best12, const int32_t* second12, int n1,
                    const int32_t* idx21, const int32_t* best21, const int32_t* second21, int n2,
                    float max_ratio, float max_distance, int cross_check, uint32_t* out_pairs,
                    int32_t* out_count, vc_stream_t stream);

/* Test hooks: out[s] = theta(s) = angle assigned to integer similarity s, for s in [0, n).
 * vc_theta_table reads the table the pair kernel's acceptance tests use (generated at build time by
 * csrc/gen_theta_table.c, clamped at s = 512^2); vc_theta_eval evaluates the same expression on the device. */
int vc_theta_table(float* out, int n, vc_stream_t stream);
int vc_theta_eval(float* out, int n, vc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Two-view geometric verification, scoring half — the step that follows descriptor matching inside
 * pycolmap.match_exhaustive (reference call site vit_colmap/pipeline/run_pipeline.py:351-363) and fills the
 * two_view_geometries table the reference's metrics read (vit_colmap/utils/metrics.py:207-243).
 * Specification: oracle/two_view_oracle.py.  Hypotheses are produced above the ABI (8x8 linear solves);
 * everything that is O(pairs x hypotheses x matches) runs here.
 *   pts      [total][4] float32 (x1, y1, x2, y2): the matched keypoints of all pairs, concatenated; 16-byte aligned
 *   offsets  [n_pairs + 1] int32: pair p owns pts[offsets[p] .. offsets[p+1])
 *   model    VC_MODEL_FUNDAMENTAL: row-major F, inlier iff (x2' F x1)^2 <= e^2 (|F x1|_xy^2 + |F' x2|_xy^2)  (Sampson)
 *            VC_MODEL_HOMOGRAPHY : row-major H, inlier iff |(H x1)_xy - x2 (H x1)_w|^2 <= e^2 (H x1)_w^2      (transfer)
 * vc_two_view_score:   hypotheses [n_pairs][n_hyp][9] -> out_counts [n_pairs][n_hyp] (NaN hypotheses count 0)
 * vc_two_view_inliers: models [n_pairs][9] -> out_mask [total] uint8
 * ------------------------------------------------------------------------------------------ */
#define VC_MODEL_FUNDAMENTAL 0
#define VC_MODEL_HOMOGRAPHY 1
int vc_two_view_score(const float* pts, const int32_t* offsets, int n_pairs, const float* hypotheses, int n_hyp,
                      int model, float max_error, int32_t* out_counts, vc_stream_t stream);
int vc_two_view_inliers(const float* pts, const int32_t* offsets, int n_pairs, const float* models, int model,
                        float max_error, uint8_t* out_mask, vc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Keypoint selection + descriptors over the ViT token grid — replaces
 * ViTExtractor._dense_to_sparse and helpers (reference vit_colmap/features/vit_extractor.py:168-653).
 * Specification: oracle/select_oracle.py.  All functions are batched over n_images.
 * ------------------------------------------------------------------------------------------ */
#define VC_DTYPE_F32 0
#define VC_DTYPE_BF16 1

#define VC_METHOD_HARRIS 0   /* vit_extractor.py:281-348 */
#define VC_METHOD_DOG 1      /* vit_extractor.py:350-394 */
#define VC_METHOD_COMBINED 2 /* vit_extractor.py:271-277 */

/*
 * tokens [n_images][H*W][C] (float32 or bfloat16, token = y*W + x) -> st [n_images][4][H*W]:
 * channel means of gx^2, gy^2, gx*gy (forward differences, zero at the last column / row) and
 * the channel mean of the features (vit_extractor.py:298-309, 365).
 */
int vc_structure_tensor(const void* tokens, int token_dtype, int n_images, int H, int W, int C,
                        float* st, vc_stream_t stream);

/* st -> score [n_images][H*W] in [0,1] (method: VC_METHOD_*).  H*W <= 16384. */
int vc_score_map(const float* st, int n_images, int H, int W, int method, float* score,
                 vc_stream_t stream);

/*
 * score -> kept keypoints in score order: spatial binning with per-bin top-k (bin_size cells),
 * global top-`target`, greedy NMS at `nms_radius` cells (vit_extractor.py:404-543).
 * Total order everywhere: score descending, then position ascending.
 * out_yx [n_images][kmax][2] (y, x), out_score [n_images][kmax], out_count [n_images]; the kernel writes every slot
 * (zeros behind the kept points), so the buffers may be handed over uninitialised.
 * dbg_cand_* (all NULL or all non-NULL, same shapes): the candidate list before NMS (slots behind it are left as they were).
 * Limits: target <= 4096, H*W*8 + 80 KiB <= 159 KiB, nms_radius <= 8; kmax >= min(target, candidates).
 */
int vc_select_keypoints(const float* score, int n_images, int H, int W, int target, int bin_size,
                        float nms_radius, int kmax, int32_t* out_yx, float* out_score,
                        int32_t* out_count, int32_t* dbg_cand_yx, float* dbg_cand_score,
                        int32_t* dbg_cand_count, vc_stream_t stream);

/*
 * Descriptors at the kept grid points: bilinear gather with the reference's grid_sample arithmetic
 * (vit_extractor.py:545-586), optional projection desc @ proj ([C][dd] float32, NULL = none;
 * vit_extractor.py:651), L2 normalisation (:243), uint8 quantisation clip(d*512, 0, 255) truncated
 * (:250), and pixel coordinates (x + 0.5) * (resized_w / W) * (orig_w / resized_w) (:229-236).
 * out_kp [n_images][kmax][2] float32 (x, y); out_desc_f32 (may be NULL) and out_desc_u8
 * [n_images][kmax][dd or C]; rows >= count[i] are zero-filled (vc_prepare_descriptors reads them).
 */
int vc_describe(const void* tokens, int token_dtype, int n_images, int H, int W, int C,
                const int32_t* yx, const int32_t* count, int kmax, const float* proj, int dd,
                int resized_w, int resized_h, int orig_w, int orig_h, float* out_kp,
                float* out_desc_f32, uint8_t* out_desc_u8, vc_stream_t stream);

/*
 * Descriptors at GIVEN sub-pixel keypoints — replaces ViTExtractor._extract_descriptors_at_keypoints of the reference's
 * hybrid extractor (vit_colmap/features/hybrid_extractor.py:224-294; the OpenCV detectors that produce the keypoints
 * stay on the host).  keypoints_xy [n_images][kmax][2] float32 (x, y) in original-image pixels; the sampling position is
 * x * (feat_w / orig_w) * (W / feat_w) as the reference computes it, then the same grid_sample arithmetic, optional
 * projection and quantiser as vc_describe.  normalisation: VC_NORM_L2, or VC_NORM_ROOTSIFT = L1-normalise, sqrt(clamp(., 1e-8)),
 * L2-normalise (:285-288).  out_desc_f32 may be NULL; rows >= count[i] are zero-filled.
 */
#define VC_NORM_L2 0
#define VC_NORM_ROOTSIFT 1
int vc_describe_at(const void* tokens, int token_dtype, int n_images, int H, int W, int C, const float* keypoints_xy,
                   const int32_t* count, int kmax, const float* proj, int dd, int feat_w, int feat_h, int orig_w,
                   int orig_h, int normalisation, float* out_desc_f32, uint8_t* out_desc_u8, vc_stream_t stream);

/* The quantiser alone: out[i] = (uint8) clip(in[i] * 512, 0, 255)   (vit_extractor.py:250). */
int vc_quantize_u8(const float* in, uint8_t* out, size_t n, vc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Keypoints + descriptors from dense head outputs — replaces the post-model part of
 * TrainableViTExtractor._run_inference (reference vit_colmap/features/trainable_vit_extractor.py:170-267;
 * _simple_nms :114-138).  Specification: oracle/trainable_oracle.py.  Batched over same-size images.
 *   kp_map        [n_images][4][H][W] float32: score logit, dx, dy, orientation (model output "keypoints")
 *   desc_map      float32 descriptor map (model output "descriptors", unit norm), element (image i, channel c,
 *                 cell p = y W + x) at i desc_image_stride + c desc_channel_stride + p desc_pixel_stride, so both
 *                 the reference's [D][H][W] (strides D H W, H W, 1) and a channels-last map (H W D, 1, D) are accepted
 *   score = sigmoid(logit) (correctly rounded float32); candidate iff score equals the maximum of its
 *   (2 nms_radius + 1)^2 window (-inf outside the map) and score > score_threshold; the kmax best by (score
 *   descending, position ascending) are kept (trainable_vit_extractor.py:181-210).
 *   out_keypoints [n_images][kmax][6] float32: x = clamp((((col + dx) + 0.5) 4) scale_x, 0, x_max), y likewise,
 *                 1, orientation, score, 0  (:219-254; scale_x = float32(orig_w / resized_w), x_max = orig_w - 1)
 *   out_desc      [n_images][kmax][D] uint8 = trunc(clip((d + 1) 127.5, 0, 255))  (:265-267)
 *   out_count     [n_images] int32; rows >= count are zero (whole blocks for vc_prepare_descriptors)
 *   workspace     vc_heatmap_workspace_bytes(n_images, H, W, kmax) bytes, 16-byte aligned.  kmax <= 65536.
 */
size_t vc_heatmap_workspace_bytes(int n_images, int H, int W, int kmax);
int vc_heatmap_keypoints(const float* kp_map, const float* desc_map, long long desc_image_stride,
                         long long desc_channel_stride, long long desc_pixel_stride, int n_images, int H, int W,
                         int D, int nms_radius, float score_threshold, int kmax, float scale_x, float scale_y,
                         float x_max, float y_max, void* workspace, float* out_keypoints, uint8_t* out_desc,
                         int32_t* out_count, vc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Image preprocessing — replaces reference vit_extractor.py:117-132 (BGR->RGB, resize to
 * multiples of 14 with cv2.INTER_LINEAR, /255, ImageNet mean/std), batched.
 * ------------------------------------------------------------------------------------------ */
#define VC_LAYOUT_NCHW 0    /* out [n][3][out_h][out_w]                                        */
#define VC_LAYOUT_PATCHES 1 /* out [n][(out_h/14)*(out_w/14)][3*14*14], element (c, dy, dx)    */
#define VC_LAYOUT_PATCHES_PAD 2 /* the same with rows padded to VC_PATCH_K_PADDED elements, zeros in the
                                   padding: the A operand of vc_patch_embed_bf16                  */
#define VC_PATCH_K_PADDED 640

/*
 * images_bgr [n_images][h][w][3] uint8 -> model input (float32 or bfloat16, VC_DTYPE_*).
 * The frame is resized only if (out_h, out_w) != (h, w).  resized_bgr_or_null (optional,
 * [n_images][out_h][out_w][3] uint8) receives the resized 8-bit frame for inspection.
 */
int vc_preprocess_u8(const uint8_t* images_bgr, int n_images, int h, int w, int out_h, int out_w,
                     int out_dtype, int layout, void* out, uint8_t* resized_bgr_or_null,
                     vc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ViT glue (bf16): y = LayerNorm(x + residual) in one pass — the two memory-bound ops between
 * the GEMMs of a DINOv2 block (the model behind reference vit_extractor.py:135-146).
 * x, residual, y: [rows][C] bfloat16; gamma, beta: [C] bfloat16; statistics in float32.
 * residual_or_null == NULL: plain LayerNorm.  sum_out_or_null (may alias neither input) receives
 * x + residual rounded to bfloat16, i.e. the new residual stream.  C % 8 == 0, C <= 2048,
 * all pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
int vc_add_layernorm_bf16(const void* x, const void* residual_or_null, const void* gamma,
                          const void* beta, float eps, int rows, int C, void* sum_out_or_null,
                          void* y_out, vc_stream_t stream);

/* Plain LayerNorm over rows that come in groups of group_rows (one image's tokens), dropping the FIRST row of every group
 * (the class token, which nothing downstream reads: reference vit_extractor.py:140-142 takes x_norm_patchtokens):
 * x [n_groups][group_rows][C] -> y [n_groups][group_rows - 1][C], dense.  Same arithmetic as vc_add_layernorm_bf16. */
int vc_layernorm_drop_first_bf16(const void* x, const void* gamma, const void* beta, float eps, int n_groups,
                                 int group_rows, int C, void* y_out, vc_stream_t stream);

/*
 * Multi-head self-attention forward, head_dim 64 (DINOv2 ViT-S/B/L): softmax(Q K^T / 8) V.
 * qkv [batch][n_tokens][3][n_heads][64] bfloat16 (the fused qkv projection's output as it is),
 * out [batch][n_tokens][n_heads*64] bfloat16 (what the output projection reads).  16-byte aligned.
 * q_prescaled != 0: the q part of qkv already carries the factor (1/8) * log2(e) (fold it into the rows of the qkv
 * projection that produce q, weights and bias): the kernel then subtracts the running row maximum inside the matrix
 * product and evaluates exp2 of the accumulator directly — one float instruction less per score, same softmax.
 */
int vc_attention_bf16(const void* qkv, int batch, int n_tokens, int n_heads, int head_dim, int q_prescaled,
                      void* out, vc_stream_t stream);

/*
 * Linear layer with fused epilogue (bf16 in, float32 accumulate on MFMA, bf16 out):
 *   out = epi(x W^T + bias),  x [rows][k_in], weight [n_out][k_in] (torch.nn.Linear layout),
 *   bias [n_out], out [rows][n_out].
 * epilogue: VC_EPI_BIAS     out = x W^T + b                    (attn.qkv)
 *           VC_EPI_GELU     out = gelu_erf(x W^T + b)          (mlp.fc1; exact GELU, erf to 1.5e-7)
 *           VC_EPI_RESIDUAL out = residual + x W^T + b         (attn.proj, mlp.fc2: the block's
 *                           residual-stream update; residual [rows][n_out], may alias out)
 * The pre-activation is NOT rounded to bf16 before GELU / the residual add (one rounding less than
 * the unfused sequence).  n_out % 128 == 0, k_in % 64 == 0, pointers 16-byte aligned,
 * residual_or_null non-NULL exactly for VC_EPI_RESIDUAL.
 * Two tile forms behind the one entry: a 256 x 256 persistent tile (one workgroup per CU, 128 KiB staging ring) when
 * n_out % 256 == 0 and rows >= 1024 — the ViT-B / ViT-L layers — and a 128 x 128 tile otherwise; same arithmetic, the
 * accumulation order over k_in is the same in both.
 * Replaces the nn.Linear / GELU / residual-add calls inside the hub model's blocks
 * (reference vit_extractor.py:135-146).
 */
#define VC_EPI_BIAS 0
#define VC_EPI_GELU 1
#define VC_EPI_RESIDUAL 2
int vc_linear_bf16(const void* x, const void* weight, const void* bias, const void* residual_or_null,
                   void* out, int rows, int n_out, int k_in, int epilogue, vc_stream_t stream);

/*
 * Convolution over a channels-last image batch as an implicit GEMM on the 256 x 256 tile (the convolutional heads of the
 * trainable extractor: reference vit_colmap/model/vit_feature_model.py:12-29 UpsampleBlock — ConvTranspose2d(4, stride 2,
 * pad 1) as four 2 x 2-tap products, one per output parity — and :96-120 trunk / heads, Conv2d 3 x 3 pad 1; eval-mode
 * BatchNorm folded into weight and bias by the caller):
 *   out[(b, y, x)][n] = epi( sum over taps t = (ty, tx), channels c of
 *                            x[b][y + dy0 + ty][x + dx0 + tx][c] * weight[n][(ty * kw + tx) * c_in + c]  + bias[n] ),
 * pixels outside the image contribute zero.  x [batch * height * width + 1][c_in] bf16: the image batch followed by ONE spare
 * row, which this call overwrites with zeros (taps outside the image read it); weight [n_out][kh * kw * c_in] bf16, bias
 * [n_out] bf16, out [batch * height * width][n_out] bf16.  epilogue VC_EPI_BIAS or VC_EPI_GELU.  n_out % 256 == 0,
 * c_in % 64 == 0, kh * kw <= 16, the batch below 4 GiB, 16-byte aligned pointers.  A 3 x 3 pad-1 convolution is
 * (kh, kw, dy0, dx0) = (3, 3, -1, -1).
 * out_parity: -1 = out as above.  2 i + j (0..3) = this call is parity class (i, j) of a stride-2 transposed convolution: row
 * (b, y, x) is written to pixel (2 y + i, 2 x + j) of out [batch][2 height][2 width][n_out]; the four calls of a layer
 * (taps (kh, kw, dy0, dx0) = (2, 2, i - 1, j - 1)) fill that tensor without an interleaving copy.
 */
int vc_conv_taps_bf16(void* x, const void* weight, const void* bias, void* out, int batch, int height, int width, int c_in,
                      int n_out, int kh, int kw, int dy0, int dx0, int out_parity, int epilogue, vc_stream_t stream);

/*
 * The same linear layer for k_in == 384 (ViT-S: attn.qkv, attn.proj, mlp.fc1), "x-stationary":
 * each wave keeps its 32 token rows in registers for the whole launch and only the weights stream
 * (csrc/gemm.hip).  Optionally fuses the LayerNorm that precedes the layer in a pre-norm block:
 *   out = epi(LayerNorm(x) W^T + b)   computed as   epi(((x - mean) rstd) (W diag(gamma))^T + (b + W beta)).
 *
 * vc_linear_xs_prepare (once per layer): weight [n_out][384] float32, bias [n_out] float32 or NULL,
 *   ln_gamma / ln_beta [384] float32 or both NULL  ->  weight_tiled (vc_linear_xs_weight_bytes bytes:
 *   bf16 in MFMA fragment order, piece (nb, ks) = features 32 nb..+32 x k 16 ks..+16, lane 32 h + r holds
 *   W'[32 nb + r][16 ks + 8 h .. +8]) and bias_folded [n_out] float32.
 * vc_linear_xs_bf16: x [rows][384] bf16, out [rows][n_out] bf16, epilogue as vc_linear_bf16;
 *   fuse_layernorm != 0 normalises each x row first (two-pass float32 statistics, eps = ln_eps; the
 *   weights must have been prepared with that LayerNorm's gamma / beta).  residual may alias out.
 *   n_out % 32 == 0, n_out <= 4096, 16-byte aligned pointers.  Launches one persistent workgroup per CU.
 */
size_t vc_linear_xs_weight_bytes(int n_out, int k_in);
int vc_linear_xs_prepare(const float* weight, const float* bias_or_null, const float* ln_gamma_or_null,
                         const float* ln_beta_or_null, int n_out, int k_in, void* weight_tiled,
                         float* bias_folded, vc_stream_t stream);
int vc_linear_xs_bf16(const void* x, const void* weight_tiled, const float* bias_folded,
                      const void* residual_or_null, void* out, int rows, int n_out, int k_in, int epilogue,
                      int fuse_layernorm, float ln_eps, const void* gelu_table_or_null, vc_stream_t stream);
/*
 * GELU table for VC_EPI_GELU (optional): with a table the epilogue evaluates the GELU as the standard bf16
 * pipeline does — on the bf16-ROUNDED pre-activation, result rounded to bf16, bit for bit
 * bf16(0.5 x (1 + erff(x / sqrt 2))) — by an LDS lookup: a handful of integer instructions and one ds_read_u16 per value instead of
 * ~150 float instructions per 32x32 block (fewer issue slots beside the MFMAs; float and integer VALU overlap
 * with the matrix pipe alike, profiles/r02_overlap_probe.md).  Without a table
 * (NULL) the GELU is evaluated in float32 on the unrounded pre-activation.  vc_gelu_table_bytes() bytes,
 * 16-byte aligned, filled once by vc_gelu_table_bf16; used when n_out * 4 + table <= 16 KiB.
 */
size_t vc_gelu_table_bytes(void);
int vc_gelu_table_bf16(void* table, vc_stream_t stream);

/*
 * DINOv2 patch embedding + position embedding in one GEMM (the conv 14x14 / 14 of the hub model as a
 * matrix product over padded patches):
 *   out[b][1 + t][:] = patches[b][t][:] W^T + bias + pos_embed[1 + t][:]      t = 0 .. tokens-1
 * patches [n_images][tokens][k_in] bf16 (vc_preprocess_u8 with VC_LAYOUT_PATCHES_PAD, k_in = VC_PATCH_K_PADDED),
 * weight [n_out][k_in] bf16 (conv weight flattened (c, dy, dx), zero padded), bias [n_out] bf16,
 * pos_embed [1 + tokens][n_out] bf16 (already interpolated to this grid), out [n_images][1 + tokens][n_out] bf16.
 * Row 0 of every image (the class token + pos_embed[0]) is NOT written here.  n_out % 128 == 0, k_in % 64 == 0.
 */
int vc_patch_embed_bf16(const void* patches, const void* weight, const void* bias, const void* pos_embed,
                        void* out, int n_images, int tokens, int n_out, int k_in, vc_stream_t stream);

/*
 * Fused MLP of a pre-norm block for dim == 384 (ViT-S):  x += fc2(gelu(fc1(LayerNorm(x))))  in one kernel — the
 * hidden tensor never exists in memory (csrc/gemm.hip, mlp_kernel).  GELU as in vc_linear_xs_bf16 with a table.
 * vc_mlp_prepare (once per block): w1 [n_hidden][384], b1 [n_hidden], LayerNorm gamma / beta [384] (or both NULL),
 *   w2 [384][n_hidden], b2 [384], all float32  ->  weights_tiled (vc_mlp_weight_bytes bytes, stage order), b1_folded
 *   [n_hidden] float32, b2_out [384] float32.  n_hidden % 32 == 0, n_hidden <= 2048.
 * vc_mlp_bf16: x_inout [rows][384] bf16, updated in place.  gelu_table from vc_gelu_table_bf16.
 */
size_t vc_mlp_weight_bytes(int n_hidden, int dim);
int vc_mlp_prepare(const float* w1, const float* b1_or_null, const float* ln_gamma_or_null,
                   const float* ln_beta_or_null, const float* w2, const float* b2_or_null, int n_hidden, int dim,
                   void* weights_tiled, float* b1_folded, float* b2_out, vc_stream_t stream);
int vc_mlp_bf16(void* x_inout, const void* weights_tiled, const float* b1_folded, const float* b2,
                const void* gelu_table, int rows, int n_hidden, int dim, float ln_eps, vc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITCOLMAP_HIP_H_ */
