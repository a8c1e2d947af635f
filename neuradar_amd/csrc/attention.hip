// Single-head self-attention of the radar decoder's transformer encoder layer (detr/models/transformer.py:
// 176-189 through nn.MultiheadAttention(d_model, nhead = 1), models/neuradar.py:250,463-491): out = dropout(softmax(Q K^T /
// sqrt(D))) V over the n tokens of one scan (107 x 33 = 3 531 rays), D = d_model = 48, forward and backward in fp32.
// The score matrix (12.5 M entries per scan) never exists: one lane owns one query (forward, dQ) or one key (dK, dV) row in
// registers and walks the rows of the other side, which every lane of the wave reads at the same address (scalar loads);
// softmax is the online form, the backward recomputes the probabilities from the forward's log-sum-exp.  The work is 100
// multiply-adds per (query, key) pair on the vector ALUs, exact fp32 with a dropout mask that the backward can reproduce.
// Measured at the radar scan's size: forward 238 us, forward + backward 1.25 ms, against 140 us / 0.39 ms of torch's
// fused attention (a matrix-core flash kernel) -- this is the dependency-free, bit-reproducible implementation of the op, not
// the fast one: `decoders.Transformer(attention="hip")` selects it, the default stays torch's.  An MFMA version (32 x 32 x 2
// fp32 tiles for Q K^T and P V) is what would close the gap; the decoder is outside the path the metric times.
// Dropout on the probabilities (training, p = 0.1 in the reference) is a counter-based hash of (seed, scan, query, key),
// identical in the forward and both backward passes; a caller may pass the keep mask explicitly instead (tests).
#include <math.h>

#include "nr_common.h"

namespace {

constexpr int kTile = 64;  // rows of the other side per LDS tile

__device__ __forceinline__ uint32_t att_hash(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

struct Drop {
  const float* mask;  // [N, n, n] keep values (0 / 1) or NULL
  uint32_t seed;
  float p;            // drop probability
};

// keep(i, j) / (1 - p): the factor torch's dropout applies to a probability
__device__ __forceinline__ float drop_factor(const Drop& d, int64_t scan, int64_t n, int i, int j) {
  if (d.p <= 0.0f) return 1.0f;
  float keep;
  if (d.mask != nullptr) {
    keep = d.mask[(scan * n + i) * n + j];
  } else {
    const uint32_t h = att_hash((uint32_t)j ^ att_hash((uint32_t)i ^ att_hash(d.seed + (uint32_t)scan * 0x9E3779B9u)));
    keep = (float)(h >> 8) * (1.0f / 16777216.0f) >= d.p ? 1.0f : 0.0f;
  }
  return keep / (1.0f - d.p);
}

// The rows of the other side are read straight from global memory at WAVE-UNIFORM addresses: the compiler turns them
// into scalar loads (s_load_dwordx8/16 through the scalar cache) and the multiply-adds take them as SGPR operands.  The first
// version staged 64-row tiles in LDS and read them back as broadcasts: 24 ds_read_b128 per (query, key) pair kept the LDS
// pipe busier than the 96 multiply-adds kept the vector ALUs (forward 227 us, 2.1 ms with the backward).
template <int D>
__device__ __forceinline__ float dot_row(const float (&a)[D], const float* __restrict__ b) {
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < D; ++c) s += a[c] * b[c];
  return s;
}

template <int D>
__device__ __forceinline__ void axpy_row(float (&acc)[D], float a, const float* __restrict__ b) {
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] += a * b[c];
}

// forward: lane = query, blockIdx.z = part of the key range (a scan has 56 waves' worth of queries: the key range is
// split so that the launch fills the chip).  Every part leaves (acc, m, l) of its keys; attention_merge_kernel combines them.
template <int D>
__global__ void __launch_bounds__(64)
attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t n, float scale,
                     Drop drop, int64_t keys_per_part, float* __restrict__ part) {
  const int64_t scan = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const float* qs = q + scan * n * D;
  const float* ks = k + scan * n * D;
  const float* vs = v + scan * n * D;
  float qi[D], acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    qi[c] = valid ? qs[(int64_t)i * D + c] * scale : 0.0f;
    acc[c] = 0.0f;
  }
  float m = -INFINITY, l = 0.0f;
  const int64_t j_begin = (int64_t)blockIdx.z * keys_per_part, j_end = j_begin + keys_per_part < n ? j_begin + keys_per_part : n;
  for (int64_t j0 = j_begin; j0 < j_end; j0 += kTile) {
    const int tj = (int)(j_end - j0 < kTile ? j_end - j0 : kTile);
    // the tile's scores first, then ONE rescale of the accumulator per tile
    float s[kTile], tmax = m;
#pragma unroll
    for (int j = 0; j < kTile; ++j) {
      s[j] = j < tj ? dot_row<D>(qi, ks + (j0 + j) * D) : -INFINITY;
      tmax = fmaxf(tmax, s[j]);
    }
    const float corr = m == -INFINITY ? 0.0f : expf(m - tmax);
    l *= corr;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] *= corr;
    m = tmax;
#pragma unroll
    for (int j = 0; j < kTile; ++j) {
      if (j >= tj) continue;
      const float p = expf(s[j] - m);
      l += p;
      const float pd = drop.p > 0.0f && valid ? p * drop_factor(drop, scan, n, i, (int)(j0 + j)) : p;
      axpy_row<D>(acc, pd, vs + (j0 + j) * D);
    }
  }
  if (valid) {
    float* o = part + (((int64_t)blockIdx.z * gridDim.y + scan) * n + i) * (D + 2);
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = acc[c];
    o[D] = m;
    o[D + 1] = l;
  }
}

template <int D>
__global__ void __launch_bounds__(256)
attention_merge_kernel(const float* __restrict__ part, int parts, int64_t rows, float* __restrict__ out, float* __restrict__ lse) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float m = -INFINITY;
  for (int p = 0; p < parts; ++p) m = fmaxf(m, part[((int64_t)p * rows + r) * (D + 2) + D]);
  float l = 0.0f, acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] = 0.0f;
  for (int p = 0; p < parts; ++p) {
    const float* o = part + ((int64_t)p * rows + r) * (D + 2);
    const float w = o[D] == -INFINITY ? 0.0f : expf(o[D] - m);
    l += w * o[D + 1];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] += w * o[c];
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int c = 0; c < D; ++c) out[r * D + c] = acc[c] * inv;
  lse[r] = m + logf(l);
}

// backward, query side: lane = query.  delta_i = dO_i . out_i;  dS_ij = P_ij (dP_ij - delta_i);  dQ_i = scale sum_j dS_ij K_j
template <int D>
__global__ void __launch_bounds__(64)
attention_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                       const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ g_out, int64_t n,
                       float scale, Drop drop, int64_t keys_per_part, float* __restrict__ g_q, float* __restrict__ delta) {
  const int64_t scan = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  const int64_t row = scan * n + i;
  float qi[D], go[D], acc[D], dl = 0.0f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    qi[c] = valid ? q[row * D + c] * scale : 0.0f;
    go[c] = valid ? g_out[row * D + c] : 0.0f;
    dl += valid ? go[c] * out[row * D + c] : 0.0f;
    acc[c] = 0.0f;
  }
  const float li = valid ? lse[row] : 0.0f;
  if (valid && blockIdx.z == 0) delta[row] = dl;
  const int64_t j_begin = (int64_t)blockIdx.z * keys_per_part, j_end = j_begin + keys_per_part < n ? j_begin + keys_per_part : n;
  const float* ks = k + scan * n * D;
  const float* vs = v + scan * n * D;
#pragma unroll 2
  for (int64_t j = j_begin; j < j_end; ++j) {
    const float p = expf(dot_row<D>(qi, ks + j * D) - li);
    float dp = dot_row<D>(go, vs + j * D);
    if (drop.p > 0.0f && valid) dp *= drop_factor(drop, scan, n, i, (int)j);
    axpy_row<D>(acc, p * (dp - dl), ks + j * D);
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < D; ++c) unsafeAtomicAdd(g_q + row * D + c, acc[c] * scale);  // (one addend per key part)
  }
}

// backward, key side: lane = key.  dV_j = sum_i Pd_ij dO_i;  dK_j = scale sum_i dS_ij Q_i
template <int D>
__global__ void __launch_bounds__(64)
attention_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                        const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ g_out, int64_t n,
                        float scale, Drop drop, int64_t rows_per_part, float* __restrict__ g_k, float* __restrict__ g_v) {
  const int64_t scan = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = j < n;
  const int64_t row = scan * n + j;
  float kj[D], vj[D], gk[D], gv[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    kj[c] = valid ? k[row * D + c] * scale : 0.0f;
    vj[c] = valid ? v[row * D + c] : 0.0f;
    gk[c] = gv[c] = 0.0f;
  }
  const int64_t i_begin = (int64_t)blockIdx.z * rows_per_part, i_end = i_begin + rows_per_part < n ? i_begin + rows_per_part : n;
  const float* qs = q + scan * n * D;
  const float* gs = g_out + scan * n * D;
#pragma unroll 2
  for (int64_t i = i_begin; i < i_end; ++i) {
    const float p = expf(dot_row<D>(kj, qs + i * D) - lse[scan * n + i]);
    const float f = drop.p > 0.0f && valid ? drop_factor(drop, scan, n, (int)i, j) : 1.0f;
    axpy_row<D>(gv, p * f, gs + i * D);
    const float dp = dot_row<D>(vj, gs + i * D) * f;
    axpy_row<D>(gk, p * (dp - delta[scan * n + i]), qs + i * D);
  }
  if (valid) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      unsafeAtomicAdd(g_k + row * D + c, gk[c] * scale);  // (one addend per query part)
      unsafeAtomicAdd(g_v + row * D + c, gv[c]);
    }
  }
}

inline bool bad(const void* a, const void* b, const void* c) { return !a || !b || !c; }

inline int att_parts(int64_t n_scans, int64_t n) {  // parts of the other side's range: ~1 024 waves in the launch
  const int64_t waves = nr_cdiv(n, 64) * n_scans;  // (4 096 waves: forward 238 -> 262 us, backward twice as long: one atomic
  int64_t parts = nr_cdiv(1024, waves);             //  addend per part and gradient element)
  const int64_t max_parts = nr_cdiv(n, kTile);
  parts = parts < 1 ? 1 : parts > 32 ? 32 : parts;
  return (int)(parts > max_parts ? max_parts : parts);
}

}  // namespace

extern "C" int64_t nr_attention_workspace_floats(int64_t n_scans, int64_t n, int d) {
  if (n_scans < 0 || n < 0 || (d != 32 && d != 48 && d != 64)) return -1;
  if (n_scans == 0 || n == 0) return 0;
  return (int64_t)att_parts(n_scans, n) * n_scans * n * (d + 2) + n_scans * n;  // forward partials | backward delta
}

extern "C" int nr_attention_fwd(const float* q, const float* k, const float* v, int64_t n_scans, int64_t n, int d, float dropout_p,
                                uint32_t seed, const float* keep_mask, float* out, float* lse, float* workspace,
                                nr_stream_t stream) {
  if (n_scans == 0 || n == 0) return 0;
  if (bad(q, k, v) || !out || !lse || !workspace || n_scans < 0 || n < 0 || n > INT_MAX / 2 || !(dropout_p >= 0.0f) || dropout_p >= 1.0f)
    return NR_EINVAL;
  const Drop drop = {keep_mask, seed, dropout_p};
  const float scale = 1.0f / sqrtf((float)d);
  const int parts = att_parts(n_scans, n);
  const int64_t per = nr_cdiv(nr_cdiv(n, parts), kTile) * kTile;
  dim3 grid((unsigned)nr_cdiv(n, 64), (unsigned)n_scans, (unsigned)nr_cdiv(n, per)), block(64);
  const int64_t rows = n_scans * n;
#define CALL(DD)                                                                                                              \
  {                                                                                                                            \
    hipLaunchKernelGGL(attention_fwd_kernel<DD>, grid, block, 0, nr_s(stream), q, k, v, n, scale, drop, per, workspace);        \
    hipLaunchKernelGGL(attention_merge_kernel<DD>, dim3((unsigned)nr_cdiv(rows, 256)), dim3(256), 0, nr_s(stream), workspace, \
                       (int)grid.z, rows, out, lse);                                                                           \
  }
  switch (d) {
    case 32: CALL(32) break;
    case 48: CALL(48) break;
    case 64: CALL(64) break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse,
                                const float* g_out, int64_t n_scans, int64_t n, int d, float dropout_p, uint32_t seed,
                                const float* keep_mask, float* g_q, float* g_k, float* g_v, float* workspace, nr_stream_t stream) {
  if (n_scans == 0 || n == 0) return 0;
  if (bad(q, k, v) || bad(out, lse, g_out) || bad(g_q, g_k, g_v) || !workspace || n_scans < 0 || n < 0 || n > INT_MAX / 2 ||
      !(dropout_p >= 0.0f) || dropout_p >= 1.0f)
    return NR_EINVAL;
  const Drop drop = {keep_mask, seed, dropout_p};
  const float scale = 1.0f / sqrtf((float)d);
  const int parts = att_parts(n_scans, n);
  const int64_t per = nr_cdiv(nr_cdiv(n, parts), kTile) * kTile;
  dim3 grid((unsigned)nr_cdiv(n, 64), (unsigned)n_scans, (unsigned)nr_cdiv(n, per)), block(64);
  float* delta = workspace + (int64_t)parts * n_scans * n * (d + 2);
#define CALL(DD)                                                                                                             \
  {                                                                                                                           \
    hipLaunchKernelGGL(attention_bwd_q_kernel<DD>, grid, block, 0, nr_s(stream), q, k, v, out, lse, g_out, n, scale, drop, per, \
                       g_q, delta);                                                                                           \
    hipLaunchKernelGGL(attention_bwd_kv_kernel<DD>, grid, block, 0, nr_s(stream), q, k, v, lse, delta, g_out, n, scale, drop,  \
                       per, g_k, g_v);                                                                                        \
  }
  switch (d) {
    case 32: CALL(32) break;
    case 48: CALL(48) break;
    case 64: CALL(64) break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
