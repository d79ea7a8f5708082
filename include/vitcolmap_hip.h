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
 */
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

/* Test hook: out[s] = theta(s) = angle assigned to integer similarity s, for s in [0, n). */
int vc_theta_table(float* out, int n, vc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VITCOLMAP_HIP_H_ */
