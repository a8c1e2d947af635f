// The radar decoder's pre-norm encoder layer around the attention, four launches instead of ~67 library ones (SURVEY 8f-2;
// detr/models/transformer.py:176-189 forward_pre with nhead = 1, + the encoder's final LayerNorm :66-68):
//
//   pre  fwd   x2 = LN1(x); [q | k] = (x2 + pos) Wqk^T + b; v = x2 Wv^T + b
//   post fwd   x1 = x + drop1(att Wo^T + bo); x3 = x1 + drop3(W2 drop2(relu(W1 LN2(x1) + b1)) + b2); out = LNf(x3)
//   post bwd   recomputes post fwd from (x, att); grad_out -> grad_att, grad_x1, d{Wo, W1, W2, LN2, LNf}
//   pre  bwd   recomputes LN1; grad_q/k/v, grad_x1 -> grad_x, d{Wqk, Wv, LN1}
//
// Everything outside the attention is row-wise in the tokens, so the layer rides on the MLP kernels' building blocks
// (mlp_tiles.h): a wave owns 32 tokens, activations live in registers in the MFMA C/D layout (lane = token, registers x lane
// half = features), Linear layers chain through v_mfma_f32_32x32x2_f32 with the weights in LDS, weight gradients contract
// over the tokens through the wave's LDS scratch.  A token's C features sit in the registers of lanes c and c + 32: LayerNorm
// statistics are a register sum + one cross-half exchange; its parameter gradients accumulate per lane across the wave's tiles
// and are reduced over the lanes once per kernel.  Dropout keep decisions: a counter-based hash of (seed + *epoch, site, row,
// feature), identical in forward and backward -- nothing is stored between them but x1's inputs (x, att), which exist anyway.
#include "mlp_tiles.h"
#include "nr_common.h"

namespace {
using namespace nrmlp;

__device__ __forceinline__ uint32_t enc_hash(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

struct EncDrop {
  uint32_t seed;  // seed of this step (the epoch folded in)
  float p, inv_keep;
};
__device__ __forceinline__ EncDrop enc_drop_of(const nr_encoder_t& e) {
  EncDrop d;
  d.seed = e.seed;
  if (e.seed_epoch != nullptr) d.seed += (uint32_t)e.seed_epoch[0] * 0x85EBCA6Bu;
  d.p = e.p_drop;
  d.inv_keep = e.p_drop > 0.0f ? 1.0f / (1.0f - e.p_drop) : 1.0f;
  return d;
}
// per-token part of the hash for dropout site `site` (1..3)
__device__ __forceinline__ uint32_t enc_drop_row(const EncDrop& d, int site, int64_t row) {
  return enc_hash((uint32_t)row ^ enc_hash(d.seed + 0x51ED270Bu * (uint32_t)site));
}
__device__ __forceinline__ float enc_drop_factor(const EncDrop& d, uint32_t hrow, int k) {
  const uint32_t h = enc_hash((uint32_t)k ^ hrow);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= d.p ? d.inv_keep : 0.0f;
}
template <int K>
__device__ __forceinline__ void enc_dropout(f32x16 (&t)[(K + 31) / 32], const EncDrop& d, int site, int64_t row, int h) {
  if (d.p <= 0.0f) return;
  const uint32_t hrow = enc_drop_row(d, site, row);
#pragma unroll
  for (int kt = 0; kt < (K + 31) / 32; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt * 32 + rowmap(r, 0) + 4 * h;
      if (k < K) t[kt][r] *= enc_drop_factor(d, hrow, k);
    }
}

// rows of a row-major [n, dim] array <-> tiles (lane = token i of the tile, features rowmap(r, h)); 16-byte pieces
template <int K>
__device__ __forceinline__ void load_rows(f32x16 (&t)[(K + 31) / 32], const float* __restrict__ p, int64_t smp, bool valid, int h) {
#pragma unroll
  for (int kt = 0; kt < (K + 31) / 32; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k0 = kt * 32 + 8 * q + 4 * h;
      float4 v4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (valid && k0 < K) v4 = *reinterpret_cast<const float4*>(p + smp * K + k0);
      t[kt][4 * q + 0] = v4.x; t[kt][4 * q + 1] = v4.y; t[kt][4 * q + 2] = v4.z; t[kt][4 * q + 3] = v4.w;
    }
}
// rows [m0, m0 + K) of the tiles' row index go to p[smp * K + (m - m0)]
template <int K, int MT>
__device__ __forceinline__ void store_rows(const f32x16 (&t)[MT], float* __restrict__ p, int64_t smp, bool valid, int h, int m0 = 0) {
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = mt * 32 + 8 * q + 4 * h - m0;
      if (m >= 0 && m < K)
        *reinterpret_cast<float4*>(p + smp * K + m) = make_float4(t[mt][4 * q], t[mt][4 * q + 1], t[mt][4 * q + 2], t[mt][4 * q + 3]);
    }
}

// LayerNorm over the C features of each token.  gb: LDS [2C] = gamma | beta.  xhat = (x - mean) * rstd (padding rows 0).
template <int C>
__device__ __forceinline__ void ln_fwd(const f32x16 (&x)[(C + 31) / 32], f32x16 (&xhat)[(C + 31) / 32], f32x16 (&y)[(C + 31) / 32],
                                       const float* gb, float eps, float& rstd, int h) {
  constexpr int KT = (C + 31) / 32;
  float s = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += (kt * 32 + rowmap(r, 0) + 4 * h < C) ? x[kt][r] : 0.0f;
  s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.0f / C);
  float v = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool on = kt * 32 + rowmap(r, 0) + 4 * h < C;
      const float d = on ? x[kt][r] - mean : 0.0f;
      xhat[kt][r] = d;
      v += d * d;
    }
  v += __shfl_xor(v, 32, 64);
  rstd = rsqrtf(v * (1.0f / C) + eps);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt * 32 + rowmap(r, 0) + 4 * h;
      const bool on = k < C;
      xhat[kt][r] *= rstd;
      y[kt][r] = on ? xhat[kt][r] * gb[on ? k : 0] + gb[C + (on ? k : 0)] : 0.0f;
    }
}

// dy -> dx (in place in `dy`); per-lane sums of dy * xhat and dy into dgam / dbet
template <int C>
__device__ __forceinline__ void ln_bwd(f32x16 (&dy)[(C + 31) / 32], const f32x16 (&xhat)[(C + 31) / 32], const float* gb, float rstd,
                                       f32x16 (&dgam)[(C + 31) / 32], f32x16 (&dbet)[(C + 31) / 32], int h) {
  constexpr int KT = (C + 31) / 32;
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt * 32 + rowmap(r, 0) + 4 * h;
      const bool on = k < C;
      const float d = on ? dy[kt][r] : 0.0f;
      dgam[kt][r] += d * xhat[kt][r];
      dbet[kt][r] += d;
      const float g = d * gb[on ? k : 0];
      dy[kt][r] = g;
      s1 += g;
      s2 += g * xhat[kt][r];
    }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  s1 *= (1.0f / C);
  s2 *= (1.0f / C);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool on = kt * 32 + rowmap(r, 0) + 4 * h < C;
      dy[kt][r] = on ? rstd * (dy[kt][r] - s1 - xhat[kt][r] * s2) : 0.0f;
    }
}

// the wave's per-lane LayerNorm parameter-gradient sums -> the block's LDS image lg[2C] (gamma | beta): reduce over the 32
// lanes of each half, lane 0 of the half adds
template <int C>
__device__ __forceinline__ void ln_grads_to_lds(const f32x16 (&dgam)[(C + 31) / 32], const f32x16 (&dbet)[(C + 31) / 32], float* lg, int i, int h) {
#pragma unroll
  for (int kt = 0; kt < (C + 31) / 32; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt * 32 + rowmap(r, 0) + 4 * h;
      float a = dgam[kt][r], b = dbet[kt][r];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
      }
      if (i == 0 && k < C) {
        atomicAdd(&lg[k], a);
        atomicAdd(&lg[C + k], b);
      }
    }
}

template <int C>
__device__ __forceinline__ void load_ln(float* lds, const float* __restrict__ w, const float* __restrict__ b) {
  for (int k = threadIdx.x; k < C; k += blockDim.x) {
    lds[k] = w[k];
    lds[C + k] = b[k];
  }
}
template <int C>
__device__ __forceinline__ void flush_ln(const float* lg, float* __restrict__ gw, float* __restrict__ gb) {
  for (int k = threadIdx.x; k < C; k += blockDim.x) {
    if (gw) unsafeAtomicAdd(gw + k, lg[k]);
    if (gb) unsafeAtomicAdd(gb + k, lg[C + k]);
  }
}

template <int N>
__device__ __forceinline__ void add_tiles(f32x16 (&a)[N], const f32x16 (&b)[N]) {
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[n][r] += b[n][r];
}

#ifndef NR_ENC_WAVES
#define NR_ENC_WAVES 4
#endif
constexpr int kWaves = NR_ENC_WAVES;

// ------------------------------------------------------------------------------------------------ pre: LN1 + in-projection
template <int C>
struct PreLds {
  using Lqk = Layer<C, 2 * C>;
  using Lv = Layer<C, C>;
  static constexpr int o_qk = 0, o_v = o_qk + Lqk::SIZE, o_ln = o_v + Lv::SIZE, W_TOTAL = o_ln + 2 * C;
  static constexpr int g_qk = 0, g_v = g_qk + Lqk::G_SIZE, g_ln = g_v + Lv::G_SIZE, G_TOTAL = g_ln + 2 * C;
};

template <int C>
__device__ __forceinline__ void pre_load(float* lw, const nr_encoder_t& e) {
  using P = PreLds<C>;
  load_layer<C, 2 * C>(lw + P::o_qk, e.in_proj_weight, e.in_proj_bias, 0);
  load_layer<C, C>(lw + P::o_v, e.in_proj_weight, e.in_proj_bias, 2 * C);
  load_ln<C>(lw + P::o_ln, e.norm1_weight, e.norm1_bias);
  __syncthreads();
}

template <int C>
__global__ void __launch_bounds__(kWaves * 64)
encoder_pre_fwd_kernel(nr_encoder_t e, const float* __restrict__ x, const float* __restrict__ pos, int64_t n, float* __restrict__ q,
                       float* __restrict__ k, float* __restrict__ v) {
  using P = PreLds<C>;
  constexpr int KT = (C + 31) / 32, QT = (2 * C + 31) / 32;
  extern __shared__ float enc_lds[];
  float* lw = enc_lds;
  pre_load<C>(lw, e);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[KT], ps[KT], xh[KT], x2[KT], qk[QT], vv[KT];
    load_rows<C>(x0, x, smp, valid, h);
    load_rows<C>(ps, pos, smp, valid, h);
    float rstd;
    ln_fwd<C>(x0, xh, x2, lw + P::o_ln, e.eps, rstd, h);
    dense_fwd<C, C, false>(x2, vv, lw + P::o_v, i, h);
    add_tiles(ps, x2);  // q and k read x2 + pos
    dense_fwd<C, 2 * C, false>(ps, qk, lw + P::o_qk, i, h);
    store_rows<C, QT>(qk, q, smp, valid, h, 0);
    store_rows<C, QT>(qk, k, smp, valid, h, C);
    store_rows<C, KT>(vv, v, smp, valid, h, 0);
  }
}

template <int C>
__global__ void __launch_bounds__(kWaves * 64)
encoder_pre_bwd_kernel(nr_encoder_t e, const float* __restrict__ x, const float* __restrict__ pos, const float* __restrict__ g_q,
                       const float* __restrict__ g_k, const float* __restrict__ g_v, const float* __restrict__ g_x1, int64_t n,
                       float* __restrict__ g_x, nr_encoder_grads_t grads) {
  using P = PreLds<C>;
  constexpr int KT = (C + 31) / 32, QT = (2 * C + 31) / 32;
  extern __shared__ float enc_lds[];
  float* lw = enc_lds;
  float* lg = lw + P::W_TOTAL;
  float* scr_all = lg + P::G_TOTAL;
  for (int t = threadIdx.x; t < P::G_TOTAL; t += blockDim.x) lg[t] = 0.0f;
  pre_load<C>(lw, e);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  float* scrA = scr_all + wave * (2 * kScrTile + 32);
  float* scrB = scrA + kScrTile;
  float* scrE = scrB + kScrTile;
  f32x16 dgam[KT], dbet[KT];
  zero_tiles(dgam);
  zero_tiles(dbet);
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[KT], ps[KT], xh[KT], x2[KT], dqk[QT], dv[KT], dx2[KT], dt[KT];
    load_rows<C>(x0, x, smp, valid, h);
    load_rows<C>(ps, pos, smp, valid, h);
    float rstd;
    ln_fwd<C>(x0, xh, x2, lw + P::o_ln, e.eps, rstd, h);
    add_tiles(ps, x2);  // the in-projection's input for q | k
    // grad of [q | k] as one 2C-row block: rows < C from grad_q, rows >= C from grad_k
#pragma unroll
    for (int mt = 0; mt < QT; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int m = mt * 32 + 8 * qd + 4 * h;
        float4 v4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (valid && m < C) v4 = *reinterpret_cast<const float4*>(g_q + smp * C + m);
        else if (valid && m < 2 * C) v4 = *reinterpret_cast<const float4*>(g_k + smp * C + (m - C));
        dqk[mt][4 * qd] = v4.x; dqk[mt][4 * qd + 1] = v4.y; dqk[mt][4 * qd + 2] = v4.z; dqk[mt][4 * qd + 3] = v4.w;
      }
    load_rows<C>(dv, g_v, smp, valid, h);
    dense_bwd_dw<C, 2 * C, false>(dqk, ps, lg + P::g_qk, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    dense_bwd_dx<C, 2 * C>(dqk, dx2, lw + P::o_qk, i, h);
    dense_bwd_dw<C, C, false>(dv, x2, lg + P::g_v, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    dense_bwd_dx<C, C>(dv, dt, lw + P::o_v, i, h);
    add_tiles(dx2, dt);
    ln_bwd<C>(dx2, xh, lw + P::o_ln, rstd, dgam, dbet, h);
    load_rows<C>(dt, g_x1, smp, valid, h);
    add_tiles(dx2, dt);
    store_rows<C, KT>(dx2, g_x, smp, valid, h, 0);
  }
  ln_grads_to_lds<C>(dgam, dbet, lg + P::g_ln, i, h);
  __syncthreads();
  flush_layer_grads<C, 2 * C>(lg + P::g_qk, grads.in_proj_weight, grads.in_proj_bias, 0);
  flush_layer_grads<C, C>(lg + P::g_v, grads.in_proj_weight, grads.in_proj_bias, 2 * C);
  flush_ln<C>(lg + P::g_ln, grads.norm1_weight, grads.norm1_bias);
}

// ------------------------------------------------------------------------------------------------ post: out-projection, FFN, norms
template <int C, int FF>
struct PostLds {
  using Lo = Layer<C, C>;
  using L1 = Layer<C, FF>;
  using L2 = Layer<FF, C>;
  static constexpr int o_o = 0, o_1 = o_o + Lo::SIZE, o_2 = o_1 + L1::SIZE, o_ln2 = o_2 + L2::SIZE, o_lnf = o_ln2 + 2 * C,
                       W_TOTAL = o_lnf + 2 * C;
  static constexpr int g_o = 0, g_1 = g_o + Lo::G_SIZE, g_2 = g_1 + L1::G_SIZE, g_ln2 = g_2 + L2::G_SIZE, g_lnf = g_ln2 + 2 * C,
                       G_TOTAL = g_lnf + 2 * C;
};

template <int C, int FF>
__device__ __forceinline__ void post_load(float* lw, const nr_encoder_t& e) {
  using P = PostLds<C, FF>;
  load_layer<C, C>(lw + P::o_o, e.out_proj_weight, e.out_proj_bias, 0);
  load_layer<C, FF>(lw + P::o_1, e.linear1_weight, e.linear1_bias, 0);
  load_layer<FF, C>(lw + P::o_2, e.linear2_weight, e.linear2_bias, 0);
  load_ln<C>(lw + P::o_ln2, e.norm2_weight, e.norm2_bias);
  load_ln<C>(lw + P::o_lnf, e.norm_weight, e.norm_bias);
  __syncthreads();
}

// the forward of the post part up to x3; everything the backward needs stays in the caller's registers
template <int C, int FF>
__device__ __forceinline__ void post_forward(const float* lw, const nr_encoder_t& e, const EncDrop& d, int64_t smp,
                                             const f32x16 (&att)[(C + 31) / 32], f32x16 (&x1)[(C + 31) / 32] /* in: x, out: x1 */,
                                             f32x16 (&xh2)[(C + 31) / 32], f32x16 (&y2)[(C + 31) / 32], float& rstd2,
                                             f32x16 (&hid)[(FF + 31) / 32] /* relu(W1 y2 + b1), dropped */, f32x16 (&x3)[(C + 31) / 32],
                                             int i, int h) {
  using P = PostLds<C, FF>;
  constexpr int KT = (C + 31) / 32;
  f32x16 o[KT];
  dense_fwd<C, C, false>(att, o, lw + P::o_o, i, h);
  enc_dropout<C>(o, d, 1, smp, h);
  add_tiles(x1, o);
  ln_fwd<C>(x1, xh2, y2, lw + P::o_ln2, e.eps, rstd2, h);
  dense_fwd<C, FF, true>(y2, hid, lw + P::o_1, i, h);
  enc_dropout<FF>(hid, d, 2, smp, h);
  dense_fwd<FF, C, false>(hid, x3, lw + P::o_2, i, h);
  enc_dropout<C>(x3, d, 3, smp, h);
  add_tiles(x3, x1);
}

template <int C, int FF>
__global__ void __launch_bounds__(kWaves * 64)
encoder_post_fwd_kernel(nr_encoder_t e, const float* __restrict__ x, const float* __restrict__ att, int64_t n, float* __restrict__ out) {
  using P = PostLds<C, FF>;
  constexpr int KT = (C + 31) / 32, FT = (FF + 31) / 32;
  extern __shared__ float enc_lds[];
  float* lw = enc_lds;
  post_load<C, FF>(lw, e);
  const EncDrop d = enc_drop_of(e);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 a[KT], x1[KT], xh2[KT], y2[KT], hid[FT], x3[KT], xhf[KT], y[KT];
    load_rows<C>(a, att, smp, valid, h);
    load_rows<C>(x1, x, smp, valid, h);
    float rstd2, rstdf;
    post_forward<C, FF>(lw, e, d, smp, a, x1, xh2, y2, rstd2, hid, x3, i, h);
    ln_fwd<C>(x3, xhf, y, lw + P::o_lnf, e.eps, rstdf, h);
    store_rows<C, KT>(y, out, smp, valid, h, 0);
  }
}

template <int C, int FF>
__global__ void __launch_bounds__(kWaves * 64)
encoder_post_bwd_kernel(nr_encoder_t e, const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ g_out,
                        int64_t n, float* __restrict__ g_att, float* __restrict__ g_x1, nr_encoder_grads_t grads) {
  using P = PostLds<C, FF>;
  constexpr int KT = (C + 31) / 32, FT = (FF + 31) / 32;
  extern __shared__ float enc_lds[];
  float* lw = enc_lds;
  float* lg = lw + P::W_TOTAL;
  float* scr_all = lg + P::G_TOTAL;
  for (int t = threadIdx.x; t < P::G_TOTAL; t += blockDim.x) lg[t] = 0.0f;
  post_load<C, FF>(lw, e);
  const EncDrop d = enc_drop_of(e);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  float* scrA = scr_all + wave * (2 * kScrTile + 32);
  float* scrB = scrA + kScrTile;
  float* scrE = scrB + kScrTile;
  f32x16 dgam2[KT], dbet2[KT], dgamf[KT], dbetf[KT];
  zero_tiles(dgam2);
  zero_tiles(dbet2);
  zero_tiles(dgamf);
  zero_tiles(dbetf);
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < tiles; tile += (int64_t)gridDim.x * kWaves) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 a[KT], x1[KT], xh2[KT], y2[KT], hid[FT], x3[KT], xhf[KT], y[KT];
    load_rows<C>(a, att, smp, valid, h);
    load_rows<C>(x1, x, smp, valid, h);
    float rstd2, rstdf;
    post_forward<C, FF>(lw, e, d, smp, a, x1, xh2, y2, rstd2, hid, x3, i, h);
    ln_fwd<C>(x3, xhf, y, lw + P::o_lnf, e.eps, rstdf, h);
    // ---- backward
    f32x16 dx3[KT], dff[KT], dhid[FT], dy2[KT];
    load_rows<C>(dx3, g_out, smp, valid, h);
    ln_bwd<C>(dx3, xhf, lw + P::o_lnf, rstdf, dgamf, dbetf, h);  // dx3 = d loss / d x3: the residual's share of d x1 as well
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) dff[kt] = dx3[kt];
    enc_dropout<C>(dff, d, 3, smp, h);
    dense_bwd_dw<FF, C, false>(dff, hid, lg + P::g_2, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    dense_bwd_dx<FF, C>(dff, dhid, lw + P::o_2, i, h);
    enc_dropout<FF>(dhid, d, 2, smp, h);
    // relu: `hid` (after dropout) is positive exactly where the activation was positive AND kept; a dropped entry's gradient
    // is already zero, so masking by hid > 0 is masking by the activation's sign
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) dhid[ft][r] = hid[ft][r] > 0.0f ? dhid[ft][r] : 0.0f;
    dense_bwd_dw<C, FF, false>(dhid, y2, lg + P::g_1, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    dense_bwd_dx<C, FF>(dhid, dy2, lw + P::o_1, i, h);
    ln_bwd<C>(dy2, xh2, lw + P::o_ln2, rstd2, dgam2, dbet2, h);
    add_tiles(dx3, dy2);  // d loss / d x1
    store_rows<C, KT>(dx3, g_x1, smp, valid, h, 0);
    enc_dropout<C>(dx3, d, 1, smp, h);  // d loss / d (att Wo^T + bo)
    dense_bwd_dw<C, C, false>(dx3, a, lg + P::g_o, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    dense_bwd_dx<C, C>(dx3, dy2, lw + P::o_o, i, h);
    store_rows<C, KT>(dy2, g_att, smp, valid, h, 0);
  }
  ln_grads_to_lds<C>(dgam2, dbet2, lg + P::g_ln2, i, h);
  ln_grads_to_lds<C>(dgamf, dbetf, lg + P::g_lnf, i, h);
  __syncthreads();
  flush_layer_grads<C, C>(lg + P::g_o, grads.out_proj_weight, grads.out_proj_bias, 0);
  flush_layer_grads<C, FF>(lg + P::g_1, grads.linear1_weight, grads.linear1_bias, 0);
  flush_layer_grads<FF, C>(lg + P::g_2, grads.linear2_weight, grads.linear2_bias, 0);
  flush_ln<C>(lg + P::g_ln2, grads.norm2_weight, grads.norm2_bias);
  flush_ln<C>(lg + P::g_lnf, grads.norm_weight, grads.norm_bias);
}

bool enc_ok(const nr_encoder_t* e) {
  return e && e->in_proj_weight && e->in_proj_bias && e->out_proj_weight && e->out_proj_bias && e->linear1_weight && e->linear1_bias &&
         e->linear2_weight && e->linear2_bias && e->norm1_weight && e->norm1_bias && e->norm2_weight && e->norm2_bias &&
         e->norm_weight && e->norm_bias && e->p_drop >= 0.0f && e->p_drop < 1.0f;
}
unsigned enc_blocks(int64_t n) {
  const int64_t b = nr_cdiv(nr_cdiv(n, 32), kWaves);
  return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
constexpr size_t kScrBytes = (size_t)kWaves * (2 * kScrTile + 32) * sizeof(float);

}  // namespace

// the four kernels' dynamic-LDS sizes per instantiated width (nr_init; the backward ones exceed the default 64 KB)
template <int C, int FF>
static int enc_init_width() {
  if (int rc = nr_raise_lds(encoder_pre_fwd_kernel<C>, PreLds<C>::W_TOTAL * sizeof(float))) return rc;
  if (int rc = nr_raise_lds(encoder_post_fwd_kernel<C, FF>, PostLds<C, FF>::W_TOTAL * sizeof(float))) return rc;
  if (int rc = nr_raise_lds(encoder_post_bwd_kernel<C, FF>, (PostLds<C, FF>::W_TOTAL + PostLds<C, FF>::G_TOTAL) * sizeof(float) + kScrBytes)) return rc;
  return nr_raise_lds(encoder_pre_bwd_kernel<C>, (PreLds<C>::W_TOTAL + PreLds<C>::G_TOTAL) * sizeof(float) + kScrBytes);
}
int nr_init_encoder() {
  if (int rc = enc_init_width<48, 64>()) return rc;
  if (int rc = enc_init_width<32, 64>()) return rc;
  return enc_init_width<64, 64>();
}

// widths the kernels are instantiated for: (d_model, dim_feedforward) = (48, 64) -- NeuRadar's radar decoder -- and (32, 64), (64, 64)
#define NR_ENC_DISPATCH(CALL)                                              \
  if (enc->d_model == 48 && enc->dim_feedforward == 64) { CALL(48, 64); }  \
  else if (enc->d_model == 32 && enc->dim_feedforward == 64) { CALL(32, 64); } \
  else if (enc->d_model == 64 && enc->dim_feedforward == 64) { CALL(64, 64); } \
  else return NR_EINVAL;

extern "C" int nr_encoder_pre_fwd(const nr_encoder_t* enc, const float* x, const float* pos, int64_t n, float* q, float* k, float* v,
                                  nr_stream_t stream) {
  if (n == 0) return 0;
  if (!enc_ok(enc) || !x || !pos || !q || !k || !v || n < 0) return NR_EINVAL;
#define CALL(C, FF)                                                                                                     \
  {                                                                                                                     \
    const size_t lds = PreLds<C>::W_TOTAL * sizeof(float);                                                              \
    hipLaunchKernelGGL(encoder_pre_fwd_kernel<C>, dim3(enc_blocks(n)), dim3(kWaves * 64), lds, nr_s(stream), *enc, x, pos, n, q, k, v); \
  }
  NR_ENC_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_encoder_post_fwd(const nr_encoder_t* enc, const float* x, const float* att, int64_t n, float* out, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!enc_ok(enc) || !x || !att || !out || n < 0) return NR_EINVAL;
#define CALL(C, FF)                                                                                                     \
  {                                                                                                                     \
    const size_t lds = PostLds<C, FF>::W_TOTAL * sizeof(float);                                                         \
    hipLaunchKernelGGL((encoder_post_fwd_kernel<C, FF>), dim3(enc_blocks(n)), dim3(kWaves * 64), lds, nr_s(stream), *enc, x, att, n, out); \
  }
  NR_ENC_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_encoder_post_bwd(const nr_encoder_t* enc, const float* x, const float* att, const float* grad_out, int64_t n,
                                   float* grad_att, float* grad_x1, const nr_encoder_grads_t* grads, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!enc_ok(enc) || !x || !att || !grad_out || !grad_att || !grad_x1 || !grads || n < 0) return NR_EINVAL;
#define CALL(C, FF)                                                                                                     \
  {                                                                                                                     \
    const size_t lds = (PostLds<C, FF>::W_TOTAL + PostLds<C, FF>::G_TOTAL) * sizeof(float) + kScrBytes;                 \
    hipLaunchKernelGGL((encoder_post_bwd_kernel<C, FF>), dim3(enc_blocks(n)), dim3(kWaves * 64), lds, nr_s(stream), *enc, x, att,  \
                       grad_out, n, grad_att, grad_x1, *grads);                                                         \
  }
  NR_ENC_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_encoder_pre_bwd(const nr_encoder_t* enc, const float* x, const float* pos, const float* grad_q, const float* grad_k,
                                  const float* grad_v, const float* grad_x1, int64_t n, float* grad_x, const nr_encoder_grads_t* grads,
                                  nr_stream_t stream) {
  if (n == 0) return 0;
  if (!enc_ok(enc) || !x || !pos || !grad_q || !grad_k || !grad_v || !grad_x1 || !grad_x || !grads || n < 0) return NR_EINVAL;
#define CALL(C, FF)                                                                                                     \
  {                                                                                                                     \
    const size_t lds = (PreLds<C>::W_TOTAL + PreLds<C>::G_TOTAL) * sizeof(float) + kScrBytes;                           \
    hipLaunchKernelGGL(encoder_pre_bwd_kernel<C>, dim3(enc_blocks(n)), dim3(kWaves * 64), lds, nr_s(stream), *enc, x, pos, grad_q, \
                       grad_k, grad_v, grad_x1, n, grad_x, *grads);                                                     \
  }
  NR_ENC_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
