// Keypoint selection and descriptor extraction over the ViT token grid, batched over images.
//
// Replaces ViTExtractor._dense_to_sparse and its helpers
// (reference vit_colmap/features/vit_extractor.py:168-653); specification:
// oracle/select_oracle.py.  Input is the ViT's own output order, tokens (images, H*W, C):
// the (C, H, W) view the reference builds (vit_extractor.py:150-156) is never materialised.
//
//   structure_tensor_kernel  one wave per token: forward differences to the right / lower
//                            neighbour, channel means of gx^2, gy^2, gx*gy and the channel mean
//                            (HBM-bound: one coalesced read of C*4 bytes per token, neighbours
//                            come from L2)
//   score_kernel             one workgroup per image: 3x3 Gaussian, Harris + edge mix / DoG,
//                            min-max normalisation with block reductions
//   select_kernel            one workgroup per image, score map in LDS: per-bin top-k, global
//                            top-k, greedy NMS, all as rank computations with the total order
//                            (score desc, position asc); bit-exact against the oracle
//   describe_kernel          one wave per keypoint: bilinear gather (the reference's
//                            grid_sample arithmetic), optional projection, L2 normalise,
//                            truncating uint8 quantiser, pixel coordinates
//
// Floating-point stages round like the oracle wherever the order of operations is defined
// (built with -ffp-contract=off); channel sums use a different association than torch's, so
// score maps agree to ~1e-6 and the tests feed identical score maps to the integer stages.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitcolmap_hip.h"
#include "common.h"

namespace {

constexpr int kMaxCells = 16384;      // H*W of one score map held in LDS (64 KiB)
constexpr int kMaxCandidates = 4096;  // keypoint candidates per image
constexpr int kSelThreads = 512;

__device__ inline float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

template <typename T>
__device__ inline float load_token(const T* p, int i);
template <>
__device__ inline float load_token<float>(const float* p, int i) { return p[i]; }
template <>
__device__ inline float load_token<uint16_t>(const uint16_t* p, int i) { return bf16_to_f32(p[i]); }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------
// structure tensor: st[img][0..3][cell] = mean_c gx^2, mean_c gy^2, mean_c gx*gy, mean_c f
// (vit_extractor.py:298-309, 365)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void structure_tensor_kernel(const T* __restrict__ tokens, int H, int W, int C,
                                                               float* __restrict__ st, bool vec_ok) {
  const int lane = threadIdx.x & 63;
  const int cells = H * W;
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int img = blockIdx.y;
  if (cell >= cells) return;
  const int y = cell / W, x = cell - y * W;
  const T* f = tokens + ((size_t)img * cells + cell) * C;
  const bool has_r = x + 1 < W, has_d = y + 1 < H;
  const T* fr = f + C;
  const T* fd = f + (size_t)W * C;
  float sxx = 0.f, syy = 0.f, sxy = 0.f, sm = 0.f;
  if (vec_ok) {                                        // C % 8 == 0 and a 16-byte aligned base (host check)
    // a lane takes 8 consecutive channels per step for either token type (same summation order for bf16 and float32
    // tokens): one 16-byte load per token row for bf16, two for float32
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    constexpr int NQ = (int)sizeof(T) / 2;             // 16-byte loads per 8 channels
    for (int c0 = lane * 8; c0 < C; c0 += 64 * 8) {
      const v4u zero = {0u, 0u, 0u, 0u};
      v4u qv[NQ], qr[NQ], qd[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        qv[q] = *((const v4u*)(f + c0) + q);
        qr[q] = has_r ? *((const v4u*)(fr + c0) + q) : zero;
        qd[q] = has_d ? *((const v4u*)(fd + c0) + q) : zero;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v, r, d;
        if (sizeof(T) == 4) {
          v = __uint_as_float(qv[j >> 2][j & 3]); r = __uint_as_float(qr[j >> 2][j & 3]); d = __uint_as_float(qd[j >> 2][j & 3]);
        } else {
          const int sh = (j & 1) * 16;
          v = __uint_as_float(((qv[0][j >> 1] >> sh) & 0xffffu) << 16);
          r = __uint_as_float(((qr[0][j >> 1] >> sh) & 0xffffu) << 16);
          d = __uint_as_float(((qd[0][j >> 1] >> sh) & 0xffffu) << 16);
        }
        const float gx = has_r ? r - v : 0.f;
        const float gy = has_d ? d - v : 0.f;
        sxx += gx * gx;
        syy += gy * gy;
        sxy += gx * gy;
        sm += v;
      }
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      const float v = load_token<T>(f, c);
      const float gx = has_r ? load_token<T>(fr, c) - v : 0.f;
      const float gy = has_d ? load_token<T>(fd, c) - v : 0.f;
      sxx += gx * gx;
      syy += gy * gy;
      sxy += gx * gy;
      sm += v;
    }
  }
  sxx = wave_sum(sxx); syy = wave_sum(syy); sxy = wave_sum(sxy); sm = wave_sum(sm);
  if (lane == 0) {
    float* o = st + (size_t)img * 4 * cells + cell;
    const float inv = (float)C;
    o[0] = sxx / inv;
    o[cells] = syy / inv;
    o[2 * cells] = sxy / inv;
    o[3 * cells] = sm / inv;
  }
}

// ---------------------------------------------------------------------------------------
// score map
// ---------------------------------------------------------------------------------------
// 1-D Gaussian taps as the reference builds them (vit_extractor.py:396-402): exp(-x^2/(2 s^2))
// in float32, normalised by their float32 sum; the 2-D kernel is their outer product.
struct Taps {
  float w[11];
  int k;
};

__device__ inline float zero_pad(const float* m, int H, int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? m[y * W + x] : 0.f;
}

// correlation with the outer-product kernel, accumulated in the oracle's order (dy major)
__device__ inline float smooth(const float* m, int H, int W, int y, int x, const Taps& t) {
  const int r = t.k / 2;
  float acc = 0.f;
  for (int dy = 0; dy < t.k; ++dy)
    for (int dx = 0; dx < t.k; ++dx) acc += (t.w[dx] * t.w[dy]) * zero_pad(m, H, W, y + dy - r, x + dx - r);
  return acc;
}

__device__ inline float block_reduce(float v, bool is_max, float* scratch) {
  // all threads get the result; scratch: >= 16 floats of LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float u = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, u) : fminf(v, u);
  }
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = scratch[0];
  for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, scratch[w]) : fminf(r, scratch[w]);
  return r;
}

// in-place (x - min) and, if the new max is > 0, / max   (vit_extractor.py:344-346, 390-392)
__device__ inline void minmax01(float* m, int cells, float* scratch) {
  float lo = INFINITY;
  for (int i = threadIdx.x; i < cells; i += blockDim.x) lo = fminf(lo, m[i]);
  lo = block_reduce(lo, false, scratch);
  float hi = -INFINITY;
  for (int i = threadIdx.x; i < cells; i += blockDim.x) {
    const float v = m[i] - lo;
    m[i] = v;
    hi = fmaxf(hi, v);
  }
  hi = block_reduce(hi, true, scratch);
  if (hi > 0.f)
    for (int i = threadIdx.x; i < cells; i += blockDim.x) m[i] = m[i] / hi;
  __syncthreads();
}

// (x - min) / (max - min + 1e-8)   (vit_extractor.py:275-276)
__device__ inline void minmax_eps(float* m, int cells, float* scratch) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < cells; i += blockDim.x) { lo = fminf(lo, m[i]); hi = fmaxf(hi, m[i]); }
  lo = block_reduce(lo, false, scratch);
  hi = block_reduce(hi, true, scratch);
  const float den = (hi - lo) + 1e-8f;
  for (int i = threadIdx.x; i < cells; i += blockDim.x) m[i] = (m[i] - lo) / den;
  __syncthreads();
}

__global__ __launch_bounds__(kSelThreads) void score_kernel(const float* __restrict__ st, int H, int W, int method,
                                                            Taps g3, Taps g7, Taps g11, float* __restrict__ score) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cells = H * W;
  float* a = lds;            // harris (or the only map)
  float* b = lds + cells;    // dog when method == combined
  float* scratch = lds + 2 * cells;
  const int img = blockIdx.x;
  const float* ixx = st + (size_t)img * 4 * cells;
  const float* iyy = ixx + cells;
  const float* ixy = iyy + cells;
  const float* avg = ixy + cells;

  if (method == 0 || method == 2) {  // harris (vit_extractor.py:312-348)
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
      const int y = i / W, x = i - y * W;
      const float sxx = smooth(ixx, H, W, y, x, g3);
      const float syy = smooth(iyy, H, W, y, x, g3);
      const float sxy = smooth(ixy, H, W, y, x, g3);
      const float det = sxx * syy - sxy * sxy;
      const float tr = sxx + syy;
      const float corner = det - 0.04f * (tr * tr);
      const float edge = sqrtf(sxx + syy);
      a[i] = 0.7f * corner + 0.3f * edge;
    }
    __syncthreads();
    minmax01(a, cells, scratch);
  }
  if (method == 1 || method == 2) {  // difference of Gaussians (vit_extractor.py:350-394)
    float* o = method == 1 ? a : b;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
      const int y = i / W, x = i - y * W;
      o[i] = fabsf(smooth(avg, H, W, y, x, g7) - smooth(avg, H, W, y, x, g11));
    }
    __syncthreads();
    minmax01(o, cells, scratch);
  }
  if (method == 2) {  // vit_extractor.py:272-277
    minmax_eps(a, cells, scratch);
    minmax_eps(b, cells, scratch);
    for (int i = threadIdx.x; i < cells; i += blockDim.x) a[i] = 0.5f * a[i] + 0.5f * b[i];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < cells; i += blockDim.x) score[(size_t)img * cells + i] = a[i];
}

// ---------------------------------------------------------------------------------------
// selection: per-bin top-k, global top-k, greedy NMS   (vit_extractor.py:404-543)
// ---------------------------------------------------------------------------------------
// "a before b" in the total order (score desc, position asc)
__device__ inline bool before(float sa, int pa, float sb, int pb) { return sa > sb || (sa == sb && pa < pb); }

__global__ __launch_bounds__(kSelThreads) void select_kernel(const float* __restrict__ score_g, int H, int W,
                                                             int target, int bin, float nms_radius, int kmax,
                                                             int32_t* __restrict__ out_yx,
                                                             float* __restrict__ out_score,
                                                             int32_t* __restrict__ out_count,
                                                             int32_t* __restrict__ dbg_cand_yx,
                                                             float* __restrict__ dbg_cand_score,
                                                             int32_t* __restrict__ dbg_cand_count) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int cells = H * W;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int img = blockIdx.x;
  float* score = lds;                                  // [cells]
  int* cand_cell = (int*)(lds + cells);                // [kMaxCandidates] candidate list (bin order)
  float* cand_score = (float*)(cand_cell + kMaxCandidates);
  int* tmp_cell = (int*)(cand_score + kMaxCandidates);    // [kMaxCandidates]
  float* tmp_score = (float*)(tmp_cell + kMaxCandidates);
  int* state = (int*)(tmp_score + kMaxCandidates);        // [kMaxCandidates] NMS: 0 open, 1 kept, 2 gone
  int* cell_rank = state + kMaxCandidates;                // [cells] rank of the candidate on a cell, or -1
  __shared__ int s_wave_tot[kSelThreads / 64];

  for (int i = tid; i < cells; i += nt) score[i] = score_g[(size_t)img * cells + i];
  __syncthreads();

  // ---- per-bin top-k: the rank of a cell inside its bin IS its output position -------------
  const int nbh = max(1, H / bin), nbw = max(1, W / bin);
  const int per_bin = max(1, target / (nbh * nbw));
  // All bins at once: every bin has the same extent (bins tile the first nbh * bin rows / nbw * bin columns; a map smaller
  // than one bin is a single clipped bin per axis), so bin b = (bi, bj) owns the output slots [b k, b k + k) and a thread
  // ranks one cell inside its own bin.  (Bin after bin — 4 bins of 256 cells at 34 x 45 with the reference's bin size 16 —
  // left three quarters of the workgroup idle in each of four rounds.)
  const int bh = min(bin, H), bw = min(bin, W), n = bh * bw;
  const int k = min(per_bin, n);
  const int K_bins = nbh * nbw * k;
  for (int idx = tid; idx < nbh * nbw * n; idx += nt) {
    const int b = idx / n, e = idx - b * n;
    const int bi = b / nbw, bj = b - bi * nbw;
    const int y0 = bi * bin, x0 = bj * bin;
    const int ey = e / bw, ex = e - ey * bw;
    const float s = score[(y0 + ey) * W + x0 + ex];
    int rank = 0;
    for (int oy = 0; oy < bh; ++oy)
      for (int ox = 0; ox < bw; ++ox)
        rank += before(score[(y0 + oy) * W + x0 + ox], oy * bw + ox, s, e) ? 1 : 0;
    if (rank < k) {
      cand_cell[b * k + rank] = (y0 + ey) * W + x0 + ex;
      cand_score[b * k + rank] = s;
    }
  }
  int K = K_bins;  // candidates produced by all bins (same value in every thread)
  __syncthreads();

  // ---- more than target: keep the global top `target`, in order ----------------------------
  // The candidate list is nbh * nbw runs of k entries, each run already in the total order (a bin's ranks; ties inside a run
  // are in candidate-index order, which is what `before` breaks ties by).  The rank of an entry in the whole list is its
  // position in its own run plus, per other run, the length of that run's prefix that comes before it — a binary search,
  // not a pass over all K candidates.
  int run_len = k, n_runs = nbh * nbw;
  auto merged_rank = [&](int e) {
    const float s = cand_score[e];
    const int mine = e / run_len;
    int rank = e - mine * run_len;
    for (int q = 0; q < n_runs; ++q) {
      if (q == mine) continue;
      int lo = 0, hi = run_len;                  // first x in run q that does NOT come before e
      while (lo < hi) {
        const int mid = (lo + hi) >> 1, o = q * run_len + mid;
        if (before(cand_score[o], o, s, e)) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    return rank;
  };
  if (K > target) {
    for (int e = tid; e < K; e += nt) {
      const int rank = merged_rank(e);
      if (rank < target) { tmp_cell[rank] = cand_cell[e]; tmp_score[rank] = cand_score[e]; }
    }
    __syncthreads();
    K = target;
    for (int e = tid; e < K; e += nt) { cand_cell[e] = tmp_cell[e]; cand_score[e] = tmp_score[e]; }
    __syncthreads();
    run_len = K;                                 // one run now, in order
    n_runs = 1;
  }
  if (dbg_cand_count) {
    for (int e = tid; e < K; e += nt) {
      const int cell = cand_cell[e];
      dbg_cand_yx[((size_t)img * kmax + e) * 2 + 0] = cell / W;
      dbg_cand_yx[((size_t)img * kmax + e) * 2 + 1] = cell % W;
      dbg_cand_score[(size_t)img * kmax + e] = cand_score[e];
    }
    if (tid == 0) dbg_cand_count[img] = K;
  }

  // ---- NMS: stable sort by score (rank), then the greedy rule as a fixed point -------------
  for (int i = tid; i < cells; i += nt) cell_rank[i] = -1;
  __syncthreads();
  for (int e = tid; e < K; e += nt) {
    const int rank = merged_rank(e);
    tmp_cell[rank] = cand_cell[e];
    tmp_score[rank] = cand_score[e];
    state[rank] = 0;
  }
  __syncthreads();
  for (int r = tid; r < K; r += nt) cell_rank[tmp_cell[r]] = r;
  __syncthreads();
  const int R = (int)floorf(nms_radius);
  const float r2 = nms_radius * nms_radius;
  // point r is kept iff no KEPT point of smaller rank lies at distance 0 < d < radius
  // (distances compare exactly on squared integers: sqrt is monotone and exact at 0)
  // (one barrier per round: __syncthreads_or carries the "somebody decided" flag; a point reads its neighbours' states while
  // others write theirs, which is harmless — a state changes once, 0 -> 1 or 0 -> 2, and a point decides only when every
  // lower-ranked neighbour has, so it sees their final values whichever side of a write the read falls)
  for (int iter = 0; iter < K + 1; ++iter) {
    int progressed = 0;
    for (int r = tid; r < K; r += nt) {
      if (state[r] != 0) continue;
      const int cell = tmp_cell[r];
      const int y = cell / W, x = cell - y * W;
      bool dead = false, wait = false;
      for (int dy = -R; dy <= R; ++dy)
        for (int dx = -R; dx <= R; ++dx) {
          const int d2 = dy * dy + dx * dx;
          if (d2 == 0 || !((float)d2 < r2)) continue;
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const int o = cell_rank[yy * W + xx];
          if (o < 0 || o > r) continue;  // no candidate there, or it ranks after this one
          const int so = state[o];
          if (so == 1) dead = true;
          else if (so == 0) wait = true;
        }
      if (dead) { state[r] = 2; progressed = 1; }
      else if (!wait) { state[r] = 1; progressed = 1; }
    }
    if (!__syncthreads_or(progressed)) break;
  }
  // ---- ordered compaction of the kept points (rank order = score order) ---------------------
  int base = 0;
  const int lane = tid & 63, wave = tid >> 6;
  for (int r0 = 0; r0 < K; r0 += nt) {
    const int r = r0 + tid;
    const bool keep = r < K && state[r] == 1;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_wave_tot[wave] = __popcll(mask);
    __syncthreads();
    int before_me = base, tot = 0;
    for (int w = 0; w < nt / 64; ++w) { if (w < wave) before_me += s_wave_tot[w]; tot += s_wave_tot[w]; }
    if (keep) {
      const int pos = before_me + __popcll(mask & ((1ull << lane) - 1ull));
      if (pos < kmax) {
        const int cell = tmp_cell[r];
        out_yx[((size_t)img * kmax + pos) * 2 + 0] = cell / W;
        out_yx[((size_t)img * kmax + pos) * 2 + 1] = cell % W;
        out_score[(size_t)img * kmax + pos] = tmp_score[r];
      }
    }
    base += tot;
    __syncthreads();
  }
  // slots behind the kept points are zero (the host hands over uninitialised buffers: three fill launches per batch less)
  const int total = min(base, kmax);
  for (int p = total + tid; p < kmax; p += nt) {
    out_yx[((size_t)img * kmax + p) * 2 + 0] = 0;
    out_yx[((size_t)img * kmax + p) * 2 + 1] = 0;
    out_score[(size_t)img * kmax + p] = 0.f;
  }
  if (tid == 0) out_count[img] = total;
}

// ---------------------------------------------------------------------------------------
// describe: gather + projection + normalise + quantise + pixel coordinates
// (vit_extractor.py:226-250, 545-586, 651)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void describe_kernel(const T* __restrict__ tokens, int H, int W, int C,
                                                       const int32_t* __restrict__ yx,
                                                       const int32_t* __restrict__ count, int kmax,
                                                       const float* __restrict__ proj, int dd, float sx1,
                                                       float sx2, float sy1, float sy2,
                                                       float* __restrict__ out_kp, float* __restrict__ out_f32,
                                                       uint8_t* __restrict__ out_u8,
                                                       // hybrid extractor (hybrid_extractor.py:224-294): descriptors at given
                                                       // sub-pixel keypoints (original-image pixels) instead of grid points,
                                                       // RootSIFT normalisation instead of L2
                                                       const float* __restrict__ kp_in, int rootsift) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int img = blockIdx.y;
  const int m = blockIdx.x * 4 + wave;
  const int n = min(count[img], kmax);
  const int out_dim = proj ? dd : C;
  float* g = lds + wave * (C + out_dim);  // gathered descriptor, then the projected one
  float* o = g + C;
  uint8_t* u8row = out_u8 + ((size_t)img * kmax + m) * out_dim;
  if (m >= kmax) return;
  if (m >= n) {  // rows beyond the count are zero: the matcher reads whole blocks
    for (int j = lane; j < out_dim; j += 64) u8row[j] = 0;
    if (out_f32) for (int j = lane; j < out_dim; j += 64) out_f32[((size_t)img * kmax + m) * out_dim + j] = 0.f;
    if (out_kp && lane < 2) out_kp[((size_t)img * kmax + m) * 2 + lane] = 0.f;
    return;
  }
  int cy = 0, cx = 0;
  float fy, fx;
  if (kp_in) {
    // kp * (w_feat / w_orig) * (W / w_feat): a float32 array times two Python doubles, one after the other
    // (hybrid_extractor.py:249-254); sx1 / sx2 carry those factors here
    fx = (kp_in[((size_t)img * kmax + m) * 2 + 0] * sx1) * sx2;
    fy = (kp_in[((size_t)img * kmax + m) * 2 + 1] * sy1) * sy2;
  } else {
    cy = yx[((size_t)img * kmax + m) * 2 + 0];
    cx = yx[((size_t)img * kmax + m) * 2 + 1];
    fy = (float)cy;
    fx = (float)cx;
  }
  // grid_sample(bilinear, border, align_corners=True) at the normalised coordinate,
  // with torch's float32 steps (oracle: gather_descriptors)
  const float gy = 2.0f * fy / (float)(H - 1) - 1.0f;
  const float gx = 2.0f * fx / (float)(W - 1) - 1.0f;
  float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
  float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  const float y0f = floorf(iy), x0f = floorf(ix);
  const float wy1 = iy - y0f, wx1 = ix - x0f;
  const float wy0 = 1.0f - wy1, wx0 = 1.0f - wx1;
  const int y0 = (int)y0f, x0 = (int)x0f, y1 = y0 + 1, x1 = x0 + 1;
  const bool in_y1 = y1 < H, in_x1 = x1 < W;
  const T* base = tokens + (size_t)img * H * W * C;
  const T* p00 = base + ((size_t)y0 * W + x0) * C;
  const T* p01 = base + ((size_t)y0 * W + min(x1, W - 1)) * C;
  const T* p10 = base + ((size_t)min(y1, H - 1) * W + x0) * C;
  const T* p11 = base + ((size_t)min(y1, H - 1) * W + min(x1, W - 1)) * C;
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
  for (int c = lane; c < C; c += 64) {
    const float v00 = load_token<T>(p00, c) * w00;
    const float v01 = (in_x1 ? load_token<T>(p01, c) : 0.f) * w01;
    const float v10 = (in_y1 ? load_token<T>(p10, c) : 0.f) * w10;
    const float v11 = (in_y1 && in_x1 ? load_token<T>(p11, c) : 0.f) * w11;
    g[c] = ((v00 + v01) + v10) + v11;
  }
  // wave-private LDS: in-order DS execution makes the writes visible to the reads below
  const float* d = g;
  if (proj) {
    for (int j = lane; j < dd; j += 64) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc += g[c] * proj[(size_t)c * dd + j];
      o[j] = acc;
    }
    d = o;
  }
  float* dw = proj ? o : g;   // the wave's own copy, rewritten in place by the RootSIFT steps
  if (rootsift) {
    // hybrid_extractor.py:285-288: L1-normalise (eps 1e-12), sqrt(clamp(., 1e-8)), then the L2 step below
    float s1 = 0.f;
    for (int j = lane; j < out_dim; j += 64) s1 += fabsf(d[j]);
    s1 = fmaxf(wave_sum(s1), 1e-12f);
    for (int j = lane; j < out_dim; j += 64) dw[j] = sqrtf(fmaxf(d[j] / s1, 1e-8f));
    d = dw;
  }
  float ss = 0.f;
  for (int j = lane; j < out_dim; j += 64) ss += d[j] * d[j];
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  for (int j = lane; j < out_dim; j += 64) {
    const float v = d[j] / nrm;
    if (out_f32) out_f32[((size_t)img * kmax + m) * out_dim + j] = v;
    const float q = fminf(fmaxf(v * 512.0f, 0.f), 255.f);
    u8row[j] = (uint8_t)q;  // truncation, negatives -> 0 (vit_extractor.py:250)
  }
  if (out_kp && lane == 0) {
    out_kp[((size_t)img * kmax + m) * 2 + 0] = (((float)cx + 0.5f) * sx1) * sx2;
    out_kp[((size_t)img * kmax + m) * 2 + 1] = (((float)cy + 0.5f) * sy1) * sy2;
  }
}

// u8 = clip(f * 512, 0, 255) truncated — the quantiser alone, for bit-exactness tests
__global__ void quantize_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)fminf(fmaxf(in[i] * 512.0f, 0.f), 255.f);
}

Taps make_taps(int k, float sigma) {
  Taps t;
  t.k = k;
  float sum = 0.f;
  for (int i = 0; i < k; ++i) {
    const float x = (float)i - (float)(k / 2);
    t.w[i] = expf(-(x * x) / (float)(2.0 * (double)sigma * (double)sigma));
    sum += t.w[i];
  }
  for (int i = 0; i < k; ++i) t.w[i] = t.w[i] / sum;
  for (int i = k; i < 11; ++i) t.w[i] = 0.f;
  return t;
}

}  // namespace

extern "C" {

int vc_structure_tensor(const void* tokens, int token_dtype, int n_images, int H, int W, int C, float* st,
                        vc_stream_t stream) {
  if (!tokens || !st || n_images < 0 || H <= 0 || W <= 0 || C <= 0) return VC_ERR_INVALID_ARG;
  if (token_dtype != VC_DTYPE_F32 && token_dtype != VC_DTYPE_BF16) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const dim3 grid((H * W + 3) / 4, n_images);
  const bool vec_ok = C % 8 == 0 && ((uintptr_t)tokens) % 16 == 0;
  if (token_dtype == VC_DTYPE_F32)
    hipLaunchKernelGGL(structure_tensor_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)tokens, H, W, C, st, vec_ok);
  else
    hipLaunchKernelGGL(structure_tensor_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)tokens, H, W, C, st, vec_ok);
  return vc::check_launch();
}

int vc_score_map(const float* st, int n_images, int H, int W, int method, float* score, vc_stream_t stream) {
  if (!st || !score || n_images < 0 || H <= 0 || W <= 0) return VC_ERR_INVALID_ARG;
  if (method < 0 || method > 2) return VC_ERR_INVALID_ARG;
  if ((long)H * W > kMaxCells) return VC_ERR_UNSUPPORTED;
  if (n_images == 0) return VC_OK;
  const size_t smem = ((size_t)2 * H * W + 32) * sizeof(float);
  static vc::PerDeviceOnce configured;
  if (int st = configured.run([] {
        return hipFuncSetAttribute((const void*)score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((2 * kMaxCells + 32) * sizeof(float)));
      }))
    return st;
  // kernel sizes as the reference derives them: k = int(6 sigma + 1), made odd (vit_extractor.py:371-374)
  const Taps g3 = make_taps(3, 1.0f), g7 = make_taps(7, 1.0f), g11 = make_taps(11, 1.6f);
  hipLaunchKernelGGL(score_kernel, dim3(n_images), dim3(kSelThreads), smem, (hipStream_t)stream, st, H, W,
                     method, g3, g7, g11, score);
  return vc::check_launch();
}

int vc_select_keypoints(const float* score, int n_images, int H, int W, int target, int bin_size,
                        float nms_radius, int kmax, int32_t* out_yx, float* out_score, int32_t* out_count,
                        int32_t* dbg_cand_yx, float* dbg_cand_score, int32_t* dbg_cand_count,
                        vc_stream_t stream) {
  if (!score || !out_yx || !out_score || !out_count) return VC_ERR_INVALID_ARG;
  if (n_images < 0 || H <= 0 || W <= 0 || target <= 0 || bin_size <= 0 || kmax <= 0) return VC_ERR_INVALID_ARG;
  if (!(nms_radius >= 0.f) || nms_radius > 8.f) return VC_ERR_UNSUPPORTED;
  if ((dbg_cand_count != nullptr) != (dbg_cand_yx != nullptr) || (dbg_cand_count != nullptr) != (dbg_cand_score != nullptr))
    return VC_ERR_INVALID_ARG;
  const long cells = (long)H * W;
  const int nb = (H / bin_size > 0 ? H / bin_size : 1) * (W / bin_size > 0 ? W / bin_size : 1);
  const long cand = (long)nb * (target / nb > 0 ? target / nb : 1);   // before the global cut
  if (cells > kMaxCells || cand > kMaxCandidates || target > kMaxCandidates) return VC_ERR_UNSUPPORTED;
  if (kmax < (cand < target ? cand : target)) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const size_t smem = (size_t)cells * 4 * 2 + (size_t)kMaxCandidates * 4 * 5;
  constexpr int kDynMax = 160 * 1024 - 1024;  // the kernel also has a few static LDS words
  if (smem > (size_t)kDynMax) return VC_ERR_UNSUPPORTED;
  static vc::PerDeviceOnce configured;
  if (int st = configured.run([] {
        return hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDynMax);
      }))
    return st;
  hipLaunchKernelGGL(select_kernel, dim3(n_images), dim3(kSelThreads), smem, (hipStream_t)stream, score, H, W,
                     target, bin_size, nms_radius, kmax, out_yx, out_score, out_count, dbg_cand_yx,
                     dbg_cand_score, dbg_cand_count);
  return vc::check_launch();
}

int vc_describe(const void* tokens, int token_dtype, int n_images, int H, int W, int C, const int32_t* yx,
                const int32_t* count, int kmax, const float* proj, int dd, int resized_w, int resized_h,
                int orig_w, int orig_h, float* out_kp, float* out_desc_f32, uint8_t* out_desc_u8,
                vc_stream_t stream) {
  if (!tokens || !yx || !count || !out_kp || !out_desc_u8) return VC_ERR_INVALID_ARG;
  if (n_images < 0 || H <= 1 || W <= 1 || C <= 0 || kmax <= 0) return VC_ERR_INVALID_ARG;
  if (token_dtype != VC_DTYPE_F32 && token_dtype != VC_DTYPE_BF16) return VC_ERR_INVALID_ARG;
  if (proj && dd <= 0) return VC_ERR_INVALID_ARG;
  if (resized_w <= 0 || resized_h <= 0 || orig_w <= 0 || orig_h <= 0) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const int out_dim = proj ? dd : C;
  const size_t smem = (size_t)4 * (C + out_dim) * sizeof(float);
  if (smem > 64 * 1024) return VC_ERR_UNSUPPORTED;
  // scale factors are Python doubles in the reference, applied one after the other to a float32
  // tensor (vit_extractor.py:229-236)
  const float sx1 = (float)((double)resized_w / (double)W), sx2 = (float)((double)orig_w / (double)resized_w);
  const float sy1 = (float)((double)resized_h / (double)H), sy2 = (float)((double)orig_h / (double)resized_h);
  const dim3 grid((kmax + 3) / 4, n_images);
  if (token_dtype == VC_DTYPE_F32)
    hipLaunchKernelGGL(describe_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, (const float*)tokens,
                       H, W, C, yx, count, kmax, proj, dd, sx1, sx2, sy1, sy2, out_kp, out_desc_f32, out_desc_u8,
                       (const float*)nullptr, 0);
  else
    hipLaunchKernelGGL(describe_kernel<uint16_t>, grid, dim3(256), smem, (hipStream_t)stream,
                       (const uint16_t*)tokens, H, W, C, yx, count, kmax, proj, dd, sx1, sx2, sy1, sy2, out_kp,
                       out_desc_f32, out_desc_u8, (const float*)nullptr, 0);
  return vc::check_launch();
}

int vc_describe_at(const void* tokens, int token_dtype, int n_images, int H, int W, int C, const float* keypoints_xy,
                   const int32_t* count, int kmax, const float* proj, int dd, int feat_w, int feat_h, int orig_w,
                   int orig_h, int normalisation, float* out_desc_f32, uint8_t* out_desc_u8, vc_stream_t stream) {
  if (!tokens || !keypoints_xy || !count || !out_desc_u8) return VC_ERR_INVALID_ARG;
  if (n_images < 0 || H <= 1 || W <= 1 || C <= 0 || kmax <= 0) return VC_ERR_INVALID_ARG;
  if (token_dtype != VC_DTYPE_F32 && token_dtype != VC_DTYPE_BF16) return VC_ERR_INVALID_ARG;
  if (proj && dd <= 0) return VC_ERR_INVALID_ARG;
  if (normalisation != VC_NORM_L2 && normalisation != VC_NORM_ROOTSIFT) return VC_ERR_INVALID_ARG;
  if (feat_w <= 0 || feat_h <= 0 || orig_w <= 0 || orig_h <= 0) return VC_ERR_INVALID_ARG;
  if (n_images == 0) return VC_OK;
  const int out_dim = proj ? dd : C;
  const size_t smem = (size_t)4 * (C + out_dim) * sizeof(float);
  if (smem > 64 * 1024) return VC_ERR_UNSUPPORTED;
  // kp_x = x * (w_feat / w_orig) * (W / w_feat): two Python doubles applied in turn to a float32 array
  const float sx1 = (float)((double)feat_w / (double)orig_w), sx2 = (float)((double)W / (double)feat_w);
  const float sy1 = (float)((double)feat_h / (double)orig_h), sy2 = (float)((double)H / (double)feat_h);
  const dim3 grid((kmax + 3) / 4, n_images);
  if (token_dtype == VC_DTYPE_F32)
    hipLaunchKernelGGL(describe_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, (const float*)tokens, H, W, C,
                       (const int32_t*)nullptr, count, kmax, proj, dd, sx1, sx2, sy1, sy2, (float*)nullptr, out_desc_f32,
                       out_desc_u8, keypoints_xy, normalisation == VC_NORM_ROOTSIFT ? 1 : 0);
  else
    hipLaunchKernelGGL(describe_kernel<uint16_t>, grid, dim3(256), smem, (hipStream_t)stream, (const uint16_t*)tokens, H, W,
                       C, (const int32_t*)nullptr, count, kmax, proj, dd, sx1, sx2, sy1, sy2, (float*)nullptr, out_desc_f32,
                       out_desc_u8, keypoints_xy, normalisation == VC_NORM_ROOTSIFT ? 1 : 0);
  return vc::check_launch();
}

int vc_quantize_u8(const float* in, uint8_t* out, size_t n, vc_stream_t stream) {
  if (!in || !out) return VC_ERR_INVALID_ARG;
  if (n == 0) return VC_OK;
  hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  return vc::check_launch();
}

}  // extern "C"
