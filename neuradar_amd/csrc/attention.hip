// Single-head self-attention of the radar decoder's transformer encoder layer (detr/models/transformer.py:
// 176-189 through nn.MultiheadAttention(d_model, nhead = 1), models/neuradar.py:250,463-491): out = dropout(softmax(Q K^T /
// sqrt(D))) V over the n tokens of one scan (107 x 33 = 3 531 rays), D = d_model = 48, forward and backward in fp32 on the
// matrix cores (v_mfma_f32_16x16x4_f32: fp32 operands, the products are exact fp32 -- this op needs no reduced precision).
// The score matrix (12.5 M entries per scan) never exists: flash-style tiles with the online softmax in the forward, the
// probabilities recomputed from the forward's log-sum-exp in the backward.  Measured at the radar scan's size (kernel
// trace): forward 36 us + 8 us merge of the key parts, backward 63 us (query side) + 93 us (key side): forward + backward
// 200 us against 373-424 us of torch's attention (hipBLASLt GEMMs + softmax kernels on this build) and 1 250 us of this
// file's first version (one lane per query on the vector ALUs).  PMC (tools/pmc_attention.sh): the matrix cores are busy
// 39 / 34 / 32 % of the three kernels' cycles (12.5 M scores x 24 / 36 / 48 MFMA steps = 16 / 24 / 31 us at the 155 TF/s
// fp32 matrix peak); the rest is waits on LDS and on the staged loads at ~2 light waves per SIMD.  Tile size, waves per
// block, query groups per wave and the part count are build-time knobs (NR_ATT_*): 64 keys x 4 waves x 1 group measured
// fastest at this size (128-key tiles: +12 %; two groups per wave: +20 %).
// Dropout on the probabilities (training, p = 0.1 in the reference) is a counter-based hash of (seed, scan, query, key),
// identical in the forward and both backward passes; a caller may pass the keep mask explicitly instead (tests).
#include <math.h>

#include "nr_common.h"

namespace {

__device__ __forceinline__ uint32_t att_hash(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

struct Drop {
  const float* mask;  // [N, n, n] keep values (0 / 1) or NULL
  uint32_t seed;
  float p;            // drop probability
  const float* epoch; // device-resident step counter (nullable): folded into the seed by the kernel, so that replays of a captured
                      // graph draw a new mask every step (the host-side seed is baked into the graph)
};
__device__ __forceinline__ Drop drop_of_step(Drop d) {
  if (d.epoch != nullptr) d.seed += (uint32_t)d.epoch[0] * 0x85EBCA6Bu;
  return d;
}

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

// ---- tiling.  v_mfma_f32_16x16x4_f32 tiles (fp32 operands: the op stays exact fp32), one wave = QW groups of
// 16 queries, the block's waves share 64-key tiles of K and V in LDS.  Everything is kept TRANSPOSED so that no value ever
// changes lanes between the two products:
//   S^T [16 keys x 16 queries]  = K_tile (A operand: lane = key l%16, contraction slot l/16 = a quarter of the d range,
//                                 read as D/4 contiguous floats of the key's LDS row) x Q^T (B operand: registers, loaded once)
//   the result leaves lane l with keys 4 (l/16) + r, r = 0..3, of query l%16 -- exactly the B operand P^T[key slot l/16][query]
//   of   O^T [16 d x 16 queries] += V^T (A operand: lane = d, key 4 (l/16) + r from LDS) x P^T, step r = 0..3.
// The contraction order over keys / d is a permutation of the natural one, which a sum does not mind.  The online softmax
// needs the maximum over a query's keys: 16 values in the lane, then two cross-lane exchanges (lanes l, l^16, l^32, l^48
// hold one query) per 64 keys; the row sums stay per lane until the end.
typedef float att_f4 __attribute__((ext_vector_type(4)));
#ifndef NR_ATT_QW
#define NR_ATT_QW 1
#endif
#ifndef NR_ATT_W
#define NR_ATT_W 4
#endif
#ifndef NR_ATT_KT
#define NR_ATT_KT 64
#endif
#ifndef NR_ATT_WAVES
#define NR_ATT_WAVES 2048
#endif

template <int D>
struct AttMfma {
  static constexpr int QW = NR_ATT_QW;  // groups of 16 queries per wave
  static constexpr int W = NR_ATT_W;  // waves per block
  static constexpr int KT = NR_ATT_KT;  // keys per LDS tile
  static constexpr int STR = D + 4;   // LDS row stride in floats (16-byte aligned rows, rows 4 apart on different banks)
  static constexpr int DS = D / 4;    // contraction steps of S (per slot: D/4 consecutive d)
  static constexpr int DT = D / 16;   // 16-row tiles of O^T
  static constexpr int ROWS = 16 * QW * W;  // queries per block
  static constexpr int PS = D + 4;    // floats per row of a part's partial result: acc [D] | m | l | pad (16-byte rows)
};

__device__ __forceinline__ float att_xor_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  return fmaxf(x, __shfl_xor(x, 32));
}
__device__ __forceinline__ float att_xor_sum(float x) {
  x += __shfl_xor(x, 16);
  return x + __shfl_xor(x, 32);
}

// rows [j0, j0 + KT) of a [n, D] matrix -> registers (zero rows past j_end) -> LDS tile: the loads of the NEXT tile are issued
// before the current one is multiplied and land while the matrix cores work
template <int D, int THREADS>
struct AttStage {
  static constexpr int KT = AttMfma<D>::KT, STR = AttMfma<D>::STR, C4 = D / 4, NS = KT * C4 / THREADS;
  static_assert(KT * C4 % THREADS == 0, "tile does not divide among the threads");
  float4 r[NS];
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int64_t j0, int64_t j_end, int tid) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int idx = tid + u * THREADS, row = idx / C4, c4 = idx - row * C4;
      r[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (j0 + row < j_end) r[u] = *reinterpret_cast<const float4*>(src + (j0 + row) * D + c4 * 4);
    }
  }
  __device__ __forceinline__ void commit(float* __restrict__ tile, int tid) const {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int idx = tid + u * THREADS, row = idx / C4, c4 = idx - row * C4;
      *reinterpret_cast<float4*>(tile + row * STR + c4 * 4) = r[u];
    }
  }
};

template <int D>
__global__ void __launch_bounds__(AttMfma<D>::W * 64)
attention_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int64_t n, float scale,
                          Drop drop_in, int64_t keys_per_part, float* __restrict__ part) {
  const Drop drop = drop_of_step(drop_in);
  using C = AttMfma<D>;
  constexpr int QW = C::QW, KT = C::KT, STR = C::STR, DS = C::DS, DT = C::DT, NKT = KT / 16;
  __shared__ __attribute__((aligned(16))) float ks[KT * STR];
  __shared__ __attribute__((aligned(16))) float vs[KT * STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, slot = lane >> 4;
  const int64_t scan = blockIdx.y;
  const float* qs = q + scan * n * D;
  const float* kp = k + scan * n * D;
  const float* vp = v + scan * n * D;
  const int q0 = (blockIdx.x * C::W + wave) * (16 * QW);
  float qreg[QW][DS];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int i = q0 + 16 * g + lq;
#pragma unroll
    for (int s = 0; s < DS; s += 4) {
      float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (i < n) t = *reinterpret_cast<const float4*>(qs + (int64_t)i * D + slot * DS + s);
      qreg[g][s] = t.x * scale; qreg[g][s + 1] = t.y * scale; qreg[g][s + 2] = t.z * scale; qreg[g][s + 3] = t.w * scale;
    }
  }
  att_f4 o[QW][DT];
  float m[QW], l[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    m[g] = -INFINITY;
    l[g] = 0.0f;
#pragma unroll
    for (int t = 0; t < DT; ++t) o[g][t] = att_f4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  const int64_t j_begin = (int64_t)blockIdx.z * keys_per_part, j_end = j_begin + keys_per_part < n ? j_begin + keys_per_part : n;
  AttStage<D, C::W * 64> sk, sv;
  sk.fetch(kp, j_begin, j_end, tid);
  sv.fetch(vp, j_begin, j_end, tid);
  for (int64_t j0 = j_begin; j0 < j_end; j0 += KT) {
    __syncthreads();
    sk.commit(ks, tid);
    sv.commit(vs, tid);
    __syncthreads();
    if (j0 + KT < j_end) {
      sk.fetch(kp, j0 + KT, j_end, tid);
      sv.fetch(vp, j0 + KT, j_end, tid);
    }
    // 1. S^T for the tile's four key groups, every query group
    att_f4 s[QW][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      float a[DS];
      const float* kr = ks + (kt * 16 + lq) * STR + slot * DS;
#pragma unroll
      for (int c = 0; c < DS; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(kr + c);
        a[c] = t.x; a[c + 1] = t.y; a[c + 2] = t.z; a[c + 3] = t.w;
      }
#pragma unroll
      for (int g = 0; g < QW; ++g) {
        att_f4 c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int st = 0; st < DS; ++st) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], qreg[g][st], c, 0, 0, 0);
        s[g][kt] = c;
      }
    }
    // 2. online softmax per query group; s becomes the (dropped) probabilities
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      float tmax = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t j = j0 + kt * 16 + 4 * slot + r;
          if (j >= j_end) s[g][kt][r] = -INFINITY;
          tmax = fmaxf(tmax, s[g][kt][r]);
        }
      tmax = att_xor_max(tmax);
      const float m_new = fmaxf(m[g], tmax);
      const float corr = m[g] == -INFINITY ? 0.0f : __expf(m[g] - m_new);
      m[g] = m_new;
      float lsum = 0.0f;
      const int i = q0 + 16 * g + lq;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(s[g][kt][r] - m_new);  // (-inf - finite: 0)
          lsum += p;
          const int64_t j = j0 + kt * 16 + 4 * slot + r;
          s[g][kt][r] = drop.p > 0.0f && i < n && j < j_end ? p * drop_factor(drop, scan, n, i, (int)j) : p;
        }
      l[g] = l[g] * corr + lsum;
#pragma unroll
      for (int t = 0; t < DT; ++t) o[g][t] *= corr;
    }
    // 3. O^T += V^T P^T
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = vs + (kt * 16 + 4 * slot + r) * STR + lq;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const float a = vr[16 * t];
#pragma unroll
          for (int g = 0; g < QW; ++g) o[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s[g][kt][r], o[g][t], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int i = q0 + 16 * g + lq;
    const float lt = att_xor_sum(l[g]);
    if (i >= n) continue;
    float* op = part + (((int64_t)blockIdx.z * gridDim.y + scan) * n + i) * C::PS;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) op[16 * t + 4 * slot + r] = o[g][t][r];
    if (slot == 0) {
      op[D] = m[g];
      op[D + 1] = lt;
    }
  }
}

// The two products every pass is made of, for the QW groups of a wave at once (operands read from LDS once per tile group):
// rows:  c[g] = TILE[16 rows of group kt] x B[g]   (A operand: lane = row l%16, its slot's D/4 consecutive columns)
// cols:  acc[g][t] += TILE^T[16 columns 16 t.., 4 rows of the slot] x b[g]   (A operand: lane = column, row 4 (l/16) + r)
template <int D, int QW>
__device__ __forceinline__ void att_rows(const float* __restrict__ tile, int kt, int lq, int slot,
                                         const float (&breg)[QW][AttMfma<D>::DS], att_f4 (&c)[QW]) {
  constexpr int DS = AttMfma<D>::DS, STR = AttMfma<D>::STR;
  float a[DS];
  const float* row = tile + (kt * 16 + lq) * STR + slot * DS;
#pragma unroll
  for (int x = 0; x < DS; x += 4) {
    const float4 t = *reinterpret_cast<const float4*>(row + x);
    a[x] = t.x; a[x + 1] = t.y; a[x + 2] = t.z; a[x + 3] = t.w;
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    att_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int st = 0; st < DS; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], breg[g][st], acc, 0, 0, 0);
    c[g] = acc;
  }
}

// backward, query side on the matrix cores: the forward's structure with two row products (S^T = K Q^T, dP^T = V dO^T) and
// K in the place of V for the accumulation: dQ^T += K^T dS^T.  delta_i = dO_i . out_i is computed here and left for the
// key side.
template <int D>
__global__ void __launch_bounds__(AttMfma<D>::W * 64)
attention_bwd_q_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                            const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ g_out, int64_t n,
                            float scale, Drop drop_in, int64_t keys_per_part, float* __restrict__ g_q, float* __restrict__ delta) {
  const Drop drop = drop_of_step(drop_in);
  using C = AttMfma<D>;
  constexpr int QW = C::QW, KT = C::KT, STR = C::STR, DS = C::DS, DT = C::DT, NKT = KT / 16;
  __shared__ __attribute__((aligned(16))) float ks[KT * STR];
  __shared__ __attribute__((aligned(16))) float vs[KT * STR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, slot = lane >> 4;
  const int64_t scan = blockIdx.y;
  const float* kp = k + scan * n * D;
  const float* vp = v + scan * n * D;
  const int q0 = (blockIdx.x * C::W + wave) * (16 * QW);
  float qreg[QW][DS], goreg[QW][DS], li[QW], dl[QW];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int i = q0 + 16 * g + lq;
    const int64_t row = scan * n + i;
    float d = 0.0f;
#pragma unroll
    for (int x = 0; x < DS; x += 4) {
      float4 tq = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tg = tq, to = tq;
      if (i < n) {
        tq = *reinterpret_cast<const float4*>(q + row * D + slot * DS + x);
        tg = *reinterpret_cast<const float4*>(g_out + row * D + slot * DS + x);
        to = *reinterpret_cast<const float4*>(out + row * D + slot * DS + x);
      }
      qreg[g][x] = tq.x * scale; qreg[g][x + 1] = tq.y * scale; qreg[g][x + 2] = tq.z * scale; qreg[g][x + 3] = tq.w * scale;
      goreg[g][x] = tg.x; goreg[g][x + 1] = tg.y; goreg[g][x + 2] = tg.z; goreg[g][x + 3] = tg.w;
      d += tg.x * to.x + tg.y * to.y + tg.z * to.z + tg.w * to.w;
    }
    dl[g] = att_xor_sum(d);
    li[g] = i < n ? lse[row] : 0.0f;
    if (i < n && slot == 0 && blockIdx.z == 0) delta[row] = dl[g];
  }
  att_f4 acc[QW][DT];
#pragma unroll
  for (int g = 0; g < QW; ++g)
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[g][t] = att_f4{0.0f, 0.0f, 0.0f, 0.0f};
  const int64_t j_begin = (int64_t)blockIdx.z * keys_per_part, j_end = j_begin + keys_per_part < n ? j_begin + keys_per_part : n;
  AttStage<D, C::W * 64> sk, sv;
  sk.fetch(kp, j_begin, j_end, tid);
  sv.fetch(vp, j_begin, j_end, tid);
  for (int64_t j0 = j_begin; j0 < j_end; j0 += KT) {
    __syncthreads();
    sk.commit(ks, tid);
    sv.commit(vs, tid);
    __syncthreads();
    if (j0 + KT < j_end) {
      sk.fetch(kp, j0 + KT, j_end, tid);
      sv.fetch(vp, j0 + KT, j_end, tid);
    }
    att_f4 s[NKT][QW], dp[NKT][QW];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      att_rows<D, QW>(ks, kt, lq, slot, qreg, s[kt]);
      att_rows<D, QW>(vs, kt, lq, slot, goreg, dp[kt]);
    }
#pragma unroll
    for (int g = 0; g < QW; ++g) {
      const int i = q0 + 16 * g + lq;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t j = j0 + kt * 16 + 4 * slot + r;
          const float p = j < j_end ? __expf(s[kt][g][r] - li[g]) : 0.0f;
          float d = dp[kt][g][r];
          if (drop.p > 0.0f && i < n && j < j_end) d *= drop_factor(drop, scan, n, i, (int)j);
          s[kt][g][r] = p * (d - dl[g]);
        }
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* kr = ks + (kt * 16 + 4 * slot + r) * STR + lq;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const float a = kr[16 * t];
#pragma unroll
          for (int g = 0; g < QW; ++g) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s[kt][g][r], acc[g][t], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int i = q0 + 16 * g + lq;
    if (i >= n) continue;
    float* gp = g_q + (scan * n + i) * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(gp + 16 * t + 4 * slot + r, acc[g][t][r] * scale);  // (one addend per key part)
  }
}

// backward, key side on the matrix cores: the wave owns groups of 16 keys (K^T and V^T in registers as B operands), the
// queries' rows of Q and dO come through LDS with their log-sum-exp and delta.  S = Q K^T and dP = dO V^T leave lane l with
// queries 4 (l/16) + r of key l%16 -- the B operands of dV^T += dO^T Pd and dK^T += Q^T dS.
template <int D>
__global__ void __launch_bounds__(AttMfma<D>::W * 64)
attention_bwd_kv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                             const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ g_out, int64_t n,
                             float scale, Drop drop_in, int64_t rows_per_part, float* __restrict__ g_k, float* __restrict__ g_v) {
  const Drop drop = drop_of_step(drop_in);
  using C = AttMfma<D>;
  constexpr int QW = C::QW, KT = C::KT, STR = C::STR, DS = C::DS, DT = C::DT, NKT = KT / 16;
  __shared__ __attribute__((aligned(16))) float qt_[KT * STR];
  __shared__ __attribute__((aligned(16))) float gt_[KT * STR];
  __shared__ __attribute__((aligned(16))) float lse_t[KT];
  __shared__ __attribute__((aligned(16))) float dl_t[KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, slot = lane >> 4;
  const int64_t scan = blockIdx.y;
  const float* qp = q + scan * n * D;
  const float* gp = g_out + scan * n * D;
  const int k0 = (blockIdx.x * C::W + wave) * (16 * QW);
  float kreg[QW][DS], vreg[QW][DS];
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int j = k0 + 16 * g + lq;
    const int64_t row = scan * n + j;
#pragma unroll
    for (int x = 0; x < DS; x += 4) {
      float4 tk = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tv = tk;
      if (j < n) {
        tk = *reinterpret_cast<const float4*>(k + row * D + slot * DS + x);
        tv = *reinterpret_cast<const float4*>(v + row * D + slot * DS + x);
      }
      kreg[g][x] = tk.x * scale; kreg[g][x + 1] = tk.y * scale; kreg[g][x + 2] = tk.z * scale; kreg[g][x + 3] = tk.w * scale;
      vreg[g][x] = tv.x; vreg[g][x + 1] = tv.y; vreg[g][x + 2] = tv.z; vreg[g][x + 3] = tv.w;
    }
  }
  att_f4 acck[QW][DT], accv[QW][DT];
#pragma unroll
  for (int g = 0; g < QW; ++g)
#pragma unroll
    for (int t = 0; t < DT; ++t) acck[g][t] = accv[g][t] = att_f4{0.0f, 0.0f, 0.0f, 0.0f};
  const int64_t i_begin = (int64_t)blockIdx.z * rows_per_part, i_end = i_begin + rows_per_part < n ? i_begin + rows_per_part : n;
  AttStage<D, C::W * 64> sq, sg;
  float pl = 0.0f;  // threads 0..KT-1: the row's lse (+inf past the end: its probabilities vanish); KT..2KT-1: its delta
  auto fetch_rows = [&](int64_t i0) {
    sq.fetch(qp, i0, i_end, tid);
    sg.fetch(gp, i0, i_end, tid);
    if (tid < KT) pl = i0 + tid < i_end ? lse[scan * n + i0 + tid] : INFINITY;
    else if (tid < 2 * KT) pl = i0 + tid - KT < i_end ? delta[scan * n + i0 + tid - KT] : 0.0f;
  };
  fetch_rows(i_begin);
  for (int64_t i0 = i_begin; i0 < i_end; i0 += KT) {
    __syncthreads();
    sq.commit(qt_, tid);
    sg.commit(gt_, tid);
    if (tid < KT) lse_t[tid] = pl;
    else if (tid < 2 * KT) dl_t[tid - KT] = pl;
    __syncthreads();
    if (i0 + KT < i_end) fetch_rows(i0 + KT);
    att_f4 s[NKT][QW], dp[NKT][QW];
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
      att_rows<D, QW>(qt_, qt, lq, slot, kreg, s[qt]);
      att_rows<D, QW>(gt_, qt, lq, slot, vreg, dp[qt]);
    }
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
      const float4 l4 = *reinterpret_cast<const float4*>(lse_t + qt * 16 + 4 * slot);
      const float4 d4 = *reinterpret_cast<const float4*>(dl_t + qt * 16 + 4 * slot);
      const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int g = 0; g < QW; ++g) {
        const int j = k0 + 16 * g + lq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t i = i0 + qt * 16 + 4 * slot + r;
          const float p = __expf(s[qt][g][r] - lr[r]);
          const float f = drop.p > 0.0f && j < n && i < i_end ? drop_factor(drop, scan, n, (int)i, j) : 1.0f;
          s[qt][g][r] = p * f;                               // Pd
          dp[qt][g][r] = p * (dp[qt][g][r] * f - dr[r]);     // dS
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = (qt * 16 + 4 * slot + r) * STR + lq;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const float ag = gt_[off + 16 * t], aq = qt_[off + 16 * t];
#pragma unroll
          for (int g = 0; g < QW; ++g) {
            accv[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, s[qt][g][r], accv[g][t], 0, 0, 0);
            acck[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, dp[qt][g][r], acck[g][t], 0, 0, 0);
          }
        }
      }
  }
#pragma unroll
  for (int g = 0; g < QW; ++g) {
    const int j = k0 + 16 * g + lq;
    if (j >= n) continue;
    float* pk = g_k + (scan * n + j) * D;
    float* pv = g_v + (scan * n + j) * D;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsafeAtomicAdd(pk + 16 * t + 4 * slot + r, acck[g][t][r] * scale);  // (one addend per query part)
        unsafeAtomicAdd(pv + 16 * t + 4 * slot + r, accv[g][t][r]);
      }
  }
}

// combines the parts' (acc, m, l): one thread per (row, four output columns)
template <int D>
__global__ void __launch_bounds__(256)
attention_merge_kernel(const float* __restrict__ part, int parts, int64_t rows, float* __restrict__ out, float* __restrict__ lse) {
  constexpr int C4 = D / 4, PS = AttMfma<D>::PS;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C4) return;
  const int64_t r = idx / C4;
  const int c = (int)(idx - r * C4);
  float m = -INFINITY;
  for (int p = 0; p < parts; ++p) m = fmaxf(m, part[((int64_t)p * rows + r) * PS + D]);
  float l = 0.0f;
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int p = 0; p < parts; ++p) {
    const float* o = part + ((int64_t)p * rows + r) * PS;
    const float w = o[D] == -INFINITY ? 0.0f : expf(o[D] - m);
    l += w * o[D + 1];
    const float4 t = *reinterpret_cast<const float4*>(o + 4 * c);
    acc.x += w * t.x; acc.y += w * t.y; acc.z += w * t.z; acc.w += w * t.w;
  }
  const float inv = 1.0f / l;
  *reinterpret_cast<float4*>(out + r * D + 4 * c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (c == 0) lse[r] = m + logf(l);
}

inline bool bad(const void* a, const void* b, const void* c) { return !a || !b || !c; }

constexpr int kMaxParts = 32;

// parts of the other side's range: a scan is n / 16 / QW waves' worth of rows, the launch wants ~2 per SIMD
template <int D>
inline int att_parts_mfma(int64_t n_scans, int64_t n) {
  const int64_t waves = nr_cdiv(n, AttMfma<D>::ROWS) * AttMfma<D>::W * n_scans;
  int64_t parts = nr_cdiv(NR_ATT_WAVES, waves);
  const int64_t max_parts = nr_cdiv(n, AttMfma<D>::KT);
  parts = parts < 1 ? 1 : parts > kMaxParts ? kMaxParts : parts;
  return (int)(parts > max_parts ? max_parts : parts);
}

}  // namespace

extern "C" int64_t nr_attention_workspace_floats(int64_t n_scans, int64_t n, int d) {
  if (n_scans < 0 || n < 0 || (d != 32 && d != 48 && d != 64)) return -1;
  if (n_scans == 0 || n == 0) return 0;
  return (int64_t)kMaxParts * n_scans * n * (d + 4) + n_scans * n;  // forward partials (at most kMaxParts parts) | backward delta
}

extern "C" int nr_attention_fwd(const float* q, const float* k, const float* v, int64_t n_scans, int64_t n, int d, float dropout_p,
                                uint32_t seed, const float* seed_epoch, const float* keep_mask, float* out, float* lse,
                                float* workspace, nr_stream_t stream) {
  if (n_scans == 0 || n == 0) return 0;
  if (bad(q, k, v) || !out || !lse || !workspace || n_scans < 0 || n < 0 || n > INT_MAX / 2 || !(dropout_p >= 0.0f) || dropout_p >= 1.0f)
    return NR_EINVAL;
  const Drop drop = {keep_mask, seed, dropout_p, seed_epoch};
  const float scale = 1.0f / sqrtf((float)d);
  const int64_t rows = n_scans * n;
#define CALL(DD)                                                                                                              \
  {                                                                                                                            \
    using A = AttMfma<DD>;                                                                                                     \
    const int64_t per = nr_cdiv(nr_cdiv(n, att_parts_mfma<DD>(n_scans, n)), A::KT) * A::KT;                                    \
    const dim3 grid((unsigned)nr_cdiv(n, A::ROWS), (unsigned)n_scans, (unsigned)nr_cdiv(n, per));                              \
    hipLaunchKernelGGL(attention_fwd_mfma_kernel<DD>, grid, dim3(A::W * 64), 0, nr_s(stream), q, k, v, n, scale, drop, per,    \
                       workspace);                                                                                             \
    hipLaunchKernelGGL(attention_merge_kernel<DD>, dim3((unsigned)nr_cdiv(rows * (DD / 4), 256)), dim3(256), 0, nr_s(stream), \
                       workspace, (int)grid.z, rows, out, lse);                                                                \
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
                                const float* seed_epoch, const float* keep_mask, float* g_q, float* g_k, float* g_v, float* workspace,
                                nr_stream_t stream) {
  if (n_scans == 0 || n == 0) return 0;
  if (bad(q, k, v) || bad(out, lse, g_out) || bad(g_q, g_k, g_v) || !workspace || n_scans < 0 || n < 0 || n > INT_MAX / 2 ||
      !(dropout_p >= 0.0f) || dropout_p >= 1.0f)
    return NR_EINVAL;
  const Drop drop = {keep_mask, seed, dropout_p, seed_epoch};
  const float scale = 1.0f / sqrtf((float)d);
  float* delta = workspace + (int64_t)kMaxParts * n_scans * n * (d + 4);
#define CALL(DD)                                                                                                              \
  {                                                                                                                            \
    using A = AttMfma<DD>;                                                                                                     \
    const int64_t per = nr_cdiv(nr_cdiv(n, att_parts_mfma<DD>(n_scans, n)), A::KT) * A::KT;                                    \
    const dim3 grid((unsigned)nr_cdiv(n, A::ROWS), (unsigned)n_scans, (unsigned)nr_cdiv(n, per));                              \
    hipLaunchKernelGGL(attention_bwd_q_mfma_kernel<DD>, grid, dim3(A::W * 64), 0, nr_s(stream), q, k, v, out, lse, g_out, n,   \
                       scale, drop, per, g_q, delta);                                                                          \
    hipLaunchKernelGGL(attention_bwd_kv_mfma_kernel<DD>, grid, dim3(A::W * 64), 0, nr_s(stream), q, k, v, lse, delta, g_out,   \
                       n, scale, drop, per, g_k, g_v);                                                                         \
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
