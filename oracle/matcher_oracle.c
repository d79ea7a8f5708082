/*
 * CPU ORACLE (test infrastructure, NOT product code) — exhaustive descriptor matcher in C.
 *
 * Same specification as oracle/matcher_oracle.py (read its header): the brute-force matcher
 * semantics behind pycolmap.match_exhaustive, which the reference calls at
 * /root/reference/vit_colmap/pipeline/run_pipeline.py:351-363 with the options of
 * /root/reference/vit_colmap/utils/config.py:64-96.  COLMAP's source is not in the container
 * (pycolmap==3.12.6 is an absent wheel, third_party/colmap an empty directory), so this is a
 * restatement of its published algorithm; PARITY UNPINNED for match contents.
 *
 * Used by: tests/ (cross-check of the numpy oracle and of the HIP kernels) and bench.py's
 * cpu_baseline leg (timed on the GPU box's host cores, OpenMP over image pairs).
 * Never linked into or loaded by vit_colmap_amd/.
 *
 * Build: make -C oracle   ->  oracle/libvco_oracle.so   (plain gcc, no -march=native: the
 * .so is built in the build container and travels to a different host)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VCO_API __attribute__((visibility("default")))

/* theta(s) = RN_f32(acos_f64(min(f32(s) * 2^-18, 1)))  (matcher_oracle.py: theta_f32) */
static inline float theta_f32(int32_t s) {
  float x = (float)s * (1.0f / (512.0f * 512.0f));
  if (x > 1.0f) x = 1.0f;
  return (float)acos((double)x);
}

static inline int accept(int32_t best, int32_t second, float max_ratio, float max_distance) {
  if (best <= 0) return 0;
  const float tb = theta_f32(best);
  if (tb > max_distance) return 0;
  const float ts = theta_f32(second);
  if (tb >= max_ratio * ts) return 0;
  return 1;
}

/* int32 dot product of two uint8 rows; written so gcc vectorises it at -O3. */
__attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
static int32_t dot_u8(const uint8_t* __restrict a, const uint8_t* __restrict b, int d) {
  int32_t acc = 0;
  for (int k = 0; k < d; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
  return acc;
}

/*
 * One pass over the similarity matrix keeps the row scan (columns ascending) and the column
 * scan (rows ascending) of the two one-way searches at the same time.
 * Outputs (any may be NULL): idx/best/second for rows (n1) and for columns (n2).
 */
VCO_API void vco_top2_both(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int d,
                           int32_t* ridx, int32_t* rbest, int32_t* rsecond,
                           int32_t* cidx, int32_t* cbest, int32_t* csecond) {
  for (int j = 0; j < n2; ++j) { cidx[j] = -1; cbest[j] = 0; csecond[j] = 0; }
  for (int i = 0; i < n1; ++i) {
    const uint8_t* a = d1 + (size_t)i * d;
    int32_t bi = -1, b = 0, s2 = 0;
    for (int j = 0; j < n2; ++j) {
      const int32_t s = dot_u8(a, d2 + (size_t)j * d, d);
      if (s > b) { bi = j; s2 = b; b = s; } else if (s > s2) { s2 = s; }
      if (s > cbest[j]) { cidx[j] = i; csecond[j] = cbest[j]; cbest[j] = s; }
      else if (s > csecond[j]) { csecond[j] = s; }
    }
    ridx[i] = bi; rbest[i] = b; rsecond[i] = s2;
  }
}

/* Matches of one image pair -> out_pairs (uint32 [<=min(n1,n2)... up to n1][2]); returns count. */
VCO_API int vco_match_pair_u8(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int d,
                              float max_ratio, float max_distance, int cross_check,
                              uint32_t* out_pairs) {
  if (n1 <= 0 || n2 <= 0) return 0;
  int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * 3 * ((size_t)n1 + n2));
  int32_t *ridx = buf, *rbest = buf + n1, *rsec = buf + 2 * n1;
  int32_t *cidx = buf + 3 * n1, *cbest = cidx + n2, *csec = cidx + 2 * n2;
  vco_top2_both(d1, n1, d2, n2, d, ridx, rbest, rsec, cidx, cbest, csec);
  int m = 0;
  for (int i = 0; i < n1; ++i) {
    if (!accept(rbest[i], rsec[i], max_ratio, max_distance)) continue;
    const int j = ridx[i];
    if (cross_check) {
      if (!accept(cbest[j], csec[j], max_ratio, max_distance)) continue;
      if (cidx[j] != i) continue;
    }
    out_pairs[2 * m] = (uint32_t)i;
    out_pairs[2 * m + 1] = (uint32_t)j;
    ++m;
  }
  free(buf);
  return m;
}

/*
 * Batch over image pairs (OpenMP).  desc: [n_images][n_max][d] uint8, counts[n_images],
 * pairs: [n_pairs][2] image indices.  out_matches: [n_pairs][n_max][2], out_counts[n_pairs].
 * Returns the number of threads used.
 */
VCO_API int vco_match_pairs_u8(const uint8_t* desc, const int32_t* counts, int n_images,
                               int n_max, int d, const int32_t* pairs, int n_pairs,
                               float max_ratio, float max_distance, int cross_check,
                               uint32_t* out_matches, int32_t* out_counts, int num_threads) {
  (void)n_images;
  int used = 1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
  used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int p = 0; p < n_pairs; ++p) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    out_counts[p] = vco_match_pair_u8(desc + (size_t)a * n_max * d, counts[a],
                                      desc + (size_t)b * n_max * d, counts[b], d, max_ratio,
                                      max_distance, cross_check,
                                      out_matches + (size_t)p * n_max * 2);
  }
  return used;
}

/* theta table entry, for the exhaustive GPU-vs-oracle check. */
VCO_API float vco_theta(int32_t s) { return theta_f32(s); }
