// NeuRADField MLP stack on MFMA: mlp_geo -> (sdf, embedding) -> SH -> mlp_feature (+residual) ->
// sigmoid density, forward and backward, one launch each (replaces 2x tcnn FullyFusedMLP + tcnn SH).
// See mlp_tiles.h for the register/LDS layout.  Persistent 4-wave workgroups, one 32-sample tile per
// wave per iteration; weights are loaded into LDS once per workgroup, weight gradients are summed in
// LDS (ds_add_f32) over all tiles of the workgroup and flushed once with contiguous global atomics.
#include "field_common.h"

using namespace nrmlp;
using namespace nrfield;

namespace {

// Every block of the three field kernels needs the whole weight image in LDS.  Building it from the
// torch-layout matrices costs five zero-fill + gather passes and barriers PER BLOCK (measured: ~20 us
// of a 54 us forward); nr_field_pack builds it once per optimizer step into global memory and the
// blocks copy it with 16-byte loads (f.packed; NULL -> build it here as before).
template <int COUNT>
__device__ __forceinline__ void copy_image(float* lw, const float* __restrict__ image) {
  static_assert(COUNT % 4 == 0, "image prefix must be a multiple of 4 floats");
  const float4* src = reinterpret_cast<const float4*>(image);
  float4* dst = reinterpret_cast<float4*>(lw);
  for (int i = threadIdx.x; i < COUNT / 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

template <int IN, int HID>
__device__ __forceinline__ void load_field_weights(float* lw, const nr_field_t& f) {
  using I = FieldImage<IN, HID>;
  if (f.packed != nullptr) {
    copy_image<I::W_TOTAL>(lw, f.packed);
    return;
  }
  load_layer<IN, HID>(lw + I::oG1, f.geo.weight[0], f.geo.bias[0], 0);
  load_layer<HID, kC>(lw + I::oG2, f.geo.weight[1], f.geo.bias[1], 1);
  for (int k = threadIdx.x; k < I::SDF; k += blockDim.x)
    lw[I::oSdf + k] = k < HID ? f.geo.weight[1][k] : (k == HID ? f.geo.bias[1][0] : 0.0f);
  load_layer<kC + kSH, HID>(lw + I::oF1, f.feat.weight[0], f.feat.bias[0], 0);
  load_layer<HID, HID>(lw + I::oF2, f.feat.weight[1], f.feat.bias[1], 0);
  load_layer<HID, kC>(lw + I::oF3, f.feat.weight[2], f.feat.bias[2], 0);
  __syncthreads();
}


#ifndef NR_MLP_FWD_WAVES
#define NR_MLP_FWD_WAVES 1
#endif
#ifndef NR_MLP_BWD_WAVES
#define NR_MLP_BWD_WAVES 1
#endif

// Activation stash (nr_field_t.stash): what the feature half of the backward would otherwise recompute -- e,
// the two hidden activations of mlp_feature (after their ReLU) and sdf -- kept per 32-sample tile in the
// registers' own layout: float (reg * 64 + lane) of the tile's block, so every store / load is one coalesced
// 256-byte access.  The backward walks the same tiles (tile t = samples 32t..32t+31), any grid size.
template <int HID> struct Stash {
  static constexpr int HT = (HID + 31) / 32;
  static constexpr int oE = 0, oF1 = 16, oF2 = 16 + 16 * HT, oSdf = 16 + 32 * HT, REGS = oSdf + 1;
  static constexpr int64_t kTile = (int64_t)REGS * 64;
};
template <int NT>
__device__ __forceinline__ void stash_put(float* __restrict__ p, const f32x16 (&t)[NT], int lane) {
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[(k * 16 + r) * 64 + lane] = t[k][r];
}
template <int NT>
__device__ __forceinline__ void stash_get(const float* __restrict__ p, f32x16 (&t)[NT], int lane) {
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[k][r] = p[(k * 16 + r) * 64 + lane];
}

template <int IN, int HID, int FW, bool STASH>  // FW: feature width F as a compile-time constant (0: runtime)
__global__ void __launch_bounds__(256, NR_MLP_FWD_WAVES)
field_fwd_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F,
                 const float* __restrict__ dirs, int S, int rows_sm, int64_t n, float* __restrict__ feature,
                 float* __restrict__ sdf_out, float* __restrict__ alpha_out) {
  using I = FieldImage<IN, HID>;
  __shared__ __attribute__((aligned(16))) float lw[I::W_TOTAL];
  load_field_weights<IN, HID>(lw, fld);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const float beta = fabsf(fld.beta[0]) + kBetaMin;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  // the next tile's features are requested before the current tile's chain of layers starts (one dependent chain per wave)
  auto foff = [&](int k) { const int Fq = FW > 0 ? FW : F; return (int64_t)(k / Fq) * sl + (k % Fq); };
  f32x16 xn[I::IT];
  auto request = [&](int64_t t) {
    const int64_t s_ = t * 32 + i;
    const bool v = t < tiles && s_ < n;
    load_rows<IN>(xn, feats + (v ? s_ * sn : 0), v, h, foff);
  };
  request((int64_t)blockIdx.x * 4 + wave);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[I::IT], h1[I::HT], e[1], cat[2], f1[I::HT], f2[I::HT], o[1];
#pragma unroll
    for (int t = 0; t < I::IT; ++t) x0[t] = xn[t];
    request(tile + (int64_t)gridDim.x * 4);
    dense_fwd<IN, HID, true>(x0, h1, lw + I::oG1, i, h);
    dense_fwd<HID, kC, false>(h1, e, lw + I::oG2, i, h);
    const float sdf = sdf_row<HID>(h1, lw + I::oSdf, h);
    cat[0] = e[0];
    const NrRowMap rm = nr_row_map(valid ? smp : 0, n, S, rows_sm);
    cat[1] = (fld.sample_dirs != nullptr ? sh_tile(fld.sample_dirs, rm.out, h) : sh_tile(dirs, rm.ray, h));
    dense_fwd<kC + kSH, HID, true>(cat, f1, lw + I::oF1, i, h);
    dense_fwd<HID, HID, true>(f1, f2, lw + I::oF2, i, h);
    dense_fwd<HID, kC, false>(f2, o, lw + I::oF3, i, h);
    if (STASH) {
      using St = Stash<HID>;
      float* sp = fld.stash + tile * St::kTile;
      stash_put<1>(sp + St::oE * 64, e, lane);
      stash_put<I::HT>(sp + St::oF1 * 64, f1, lane);
      stash_put<I::HT>(sp + St::oF2 * 64, f2, lane);
      sp[St::oSdf * 64 + lane] = sdf;
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // rows 8q+4h .. +3 are registers 4q..4q+3
        float4 v = make_float4(e[0][4 * q] + o[0][4 * q], e[0][4 * q + 1] + o[0][4 * q + 1],
                               e[0][4 * q + 2] + o[0][4 * q + 2], e[0][4 * q + 3] + o[0][4 * q + 3]);
        *reinterpret_cast<float4*>(feature + rm.out * kC + 8 * q + 4 * h) = v;
      }
      if (h == 0) {
        sdf_out[rm.out] = sdf;
        alpha_out[rm.out] = 1.0f / (1.0f + expf(sdf * beta));  // sigmoid(-sdf*beta), utils.py:38
      }
    }
  }
}

template <int ROWS>
__device__ __forceinline__ void relu_mask(f32x16 (&g)[(ROWS + 31) / 32], const f32x16 (&y)[(ROWS + 31) / 32]) {
#pragma unroll
  for (int t = 0; t < (ROWS + 31) / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g[t][r] = y[t][r] > 0.0f ? g[t][r] : 0.0f;
}

// Backward is split into two launches so that each keeps its weight-gradient accumulators AND its
// activations in registers (one kernel for all five layers needs > 512 registers per lane at width 64):
//   A (feature half): recompute geo fwd -> e, sdf; feat fwd; backward through mlp_feature and the
//     sigmoid; writes d_e [n,C] and d_sdf [n] to the workspace; accumulates dV1..dV3, d_beta.
//   B (geometry half): recompute h1; backward through mlp_geo from the workspace; accumulates dW1, dW2
//     (the sdf row via per-lane partial products); writes grad_feats.

template <int IN, int HID, int FW, bool STASH>  // FW: feature width F as a compile-time constant (0: runtime)
__global__ void __launch_bounds__(256, NR_MLP_BWD_WAVES)
field_bwd_feat_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F,
                      const float* __restrict__ dirs, int S, int rows_sm, int64_t n, const float* __restrict__ g_feature,
                      const float* __restrict__ g_alpha, const float* __restrict__ g_sdf, float* __restrict__ ws,
                      float* __restrict__ slab) {
  using I = FieldImage<IN, HID>;
  constexpr int kScrPerWave = kBwdScrTiles * kScrTile;
  constexpr int kImg = I::F1::G_SIZE + I::F2::G_SIZE + I::F3::G_SIZE + 2;
  constexpr int kScrTotal = 4 * kScrPerWave > kImg ? 4 * kScrPerWave : kImg;
  constexpr int oF1 = 0, oF2 = oF1 + I::F1::G_SIZE, oF3 = oF2 + I::F2::G_SIZE, oBeta = oF3 + I::F3::G_SIZE;
  __shared__ __attribute__((aligned(16))) float lw[I::W_TOTAL];
  __shared__ float scr_all[kScrTotal];  // per-wave staging; reused as the block's gradient image at the end
  load_field_weights<IN, HID>(lw, fld);  // ends with __syncthreads()
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * kScrPerWave;
  const float beta_raw = fld.beta[0];
  const float beta = fabsf(beta_raw) + kBetaMin;
  f32x16 aF1[I::HT][2], aF2[I::HT][I::HT], aF3[1][I::HT];
  float bF1[I::HT], bF2[I::HT], bF3[1], d_beta = 0.0f;
#pragma unroll
  for (int t = 0; t < I::HT; ++t) { zero_tiles(aF1[t]); zero_tiles(aF2[t]); bF1[t] = bF2[t] = 0.0f; }
  zero_tiles(aF3[0]);
  bF3[0] = 0.0f;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 cat[2], f1[I::HT], f2[I::HT];
    float sdf;
    const NrRowMap rm = nr_row_map(valid ? smp : 0, n, S, rows_sm);
    if (STASH) {  // the forward left e, f1, f2, sdf of this tile in the stash: 81 loads instead of 176 MFMAs
      using St = Stash<HID>;
      const float* sp = fld.stash + tile * St::kTile;
      f32x16 e[1];
      stash_get<1>(sp + St::oE * 64, e, lane);
      cat[0] = e[0];
      stash_get<I::HT>(sp + St::oF1 * 64, f1, lane);
      stash_get<I::HT>(sp + St::oF2 * 64, f2, lane);
      sdf = sp[St::oSdf * 64 + lane];
      cat[1] = (fld.sample_dirs != nullptr ? sh_tile(fld.sample_dirs, rm.out, h) : sh_tile(dirs, rm.ray, h));
    } else {
      {
        f32x16 x0[I::IT], h1[I::HT], e[1];
        load_rows<IN>(x0, feats + (valid ? smp * sn : 0), valid, h,
                      [&](int k) { const int Fq = FW > 0 ? FW : F; return (int64_t)(k / Fq) * sl + (k % Fq); });
        dense_fwd<IN, HID, true>(x0, h1, lw + I::oG1, i, h);
        dense_fwd<HID, kC, false>(h1, e, lw + I::oG2, i, h);
        sdf = sdf_row<HID>(h1, lw + I::oSdf, h);
        cat[0] = e[0];
      }
      cat[1] = (fld.sample_dirs != nullptr ? sh_tile(fld.sample_dirs, rm.out, h) : sh_tile(dirs, rm.ray, h));
      dense_fwd<kC + kSH, HID, true>(cat, f1, lw + I::oF1, i, h);
      dense_fwd<HID, HID, true>(f1, f2, lw + I::oF2, i, h);
    }
    f32x16 d_o[1], d_f2[I::HT], d_f1[I::HT], d_cat[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // rows 8q+4h .. +3 of the sample's [C] row are registers 4q..4q+3: 16-byte loads
      const float4 v = valid ? *reinterpret_cast<const float4*>(g_feature + rm.out * kC + 8 * q + 4 * h) : make_float4(0, 0, 0, 0);
      d_o[0][4 * q] = v.x; d_o[0][4 * q + 1] = v.y; d_o[0][4 * q + 2] = v.z; d_o[0][4 * q + 3] = v.w;
    }
    dense_bwd_dw_reg<HID, kC>(d_o, f2, aF3, bF3, scr, i, h);          // layers[2]: o = V3 f2 + b
    dense_bwd_dx<HID, kC>(d_o, d_f2, lw + I::oF3, i, h);
    relu_mask<HID>(d_f2, f2);
    dense_bwd_dw_reg<HID, HID>(d_f2, f1, aF2, bF2, scr, i, h);        // layers[1]
    dense_bwd_dx<HID, HID>(d_f2, d_f1, lw + I::oF2, i, h);
    relu_mask<HID>(d_f1, f1);
    dense_bwd_dw_reg<kC + kSH, HID>(d_f1, cat, aF1, bF1, scr, i, h);  // layers[0]: input [e ; sh]
    dense_bwd_dx<kC + kSH, HID>(d_f1, d_cat, lw + I::oF1, i, h);
    // alpha = sigmoid(-sdf * beta)
    const float ga = valid ? g_alpha[rm.out] : 0.0f;
    const float a = 1.0f / (1.0f + expf(sdf * beta));
    const float dsig = ga * a * (1.0f - a);
    float d_sdf = dsig * (-beta);
    if (g_sdf != nullptr && valid) d_sdf += g_sdf[rm.out];
    if (h == 0) d_beta += dsig * (-sdf) * (beta_raw >= 0.0f ? 1.0f : -1.0f);
    {  // d_e = d_o + d_cat[0] (residual; SH carries no gradient) and d_sdf -> workspace (zeros for rows >= n)
      float* w = ws + tile * kWsTile;
#pragma unroll
      for (int r = 0; r < 16; ++r) w[r * 64 + lane] = valid ? d_o[0][r] + d_cat[0][r] : 0.0f;
      w[16 * 64 + lane] = valid ? d_sdf : 0.0f;
    }
  }
  d_beta = nr_wave_sum(d_beta);
  float* img = scr_all;
  __syncthreads();  // every wave is done with its staging scratch
  for (int w = 0; w < 4; ++w) {  // wave 0 assigns, waves 1..3 add: plain LDS stores, no atomics
    if (wave == w) {
      const bool first = w == 0;
      merge_dw<kC + kSH, HID>(aF1, bF1, img + oF1, first, i, h);
      merge_dw<HID, HID>(aF2, bF2, img + oF2, first, i, h);
      merge_dw<HID, kC>(aF3, bF3, img + oF3, first, i, h);
      if (lane == 0) img[oBeta] = first ? d_beta : img[oBeta] + d_beta;
    }
    __syncthreads();
  }
  // the block's partial sums go to its own slab with plain stores; field_grad_reduce_kernel adds the
  // slabs up (256 blocks flushing 13.6k floats each with atomics onto the SAME addresses cost ~50 us)
  float* out = slab + (int64_t)blockIdx.x * I::G_TOTAL + I::gF1;
  if (threadIdx.x == 0) img[oBeta + 1] = 0.0f;
  __syncthreads();
  for (int k = threadIdx.x; k < kImg; k += blockDim.x) out[k] = img[k];
}

template <int IN, int HID, int FW>  // FW: feature width F as a compile-time constant (0: runtime)
__global__ void __launch_bounds__(256, NR_MLP_BWD_WAVES)
field_bwd_geo_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F, int64_t n,
                     const float* __restrict__ ws, float* __restrict__ g_feats, float* __restrict__ slab) {
  using I = FieldImage<IN, HID>;
  constexpr int kScrPerWave = kBwdScrTiles * kScrTile;
  constexpr int kImg = I::G1::G_SIZE + I::G2::G_SIZE + I::SDF;
  constexpr int kScrTotal = 4 * kScrPerWave > kImg ? 4 * kScrPerWave : kImg;
  constexpr int oG1 = 0, oG2 = oG1 + I::G1::G_SIZE, oSdf = oG2 + I::G2::G_SIZE;
  __shared__ __attribute__((aligned(16))) float lw[I::oF1];  // only the geometry MLP's weights: a prefix of the image
  __shared__ float scr_all[kScrTotal];
  if (fld.packed != nullptr) {
    copy_image<I::oF1>(lw, fld.packed);
  } else {
    load_layer<IN, HID>(lw + I::oG1, fld.geo.weight[0], fld.geo.bias[0], 0);
    load_layer<HID, kC>(lw + I::oG2, fld.geo.weight[1], fld.geo.bias[1], 1);
    for (int k = threadIdx.x; k < I::SDF; k += blockDim.x)
      lw[I::oSdf + k] = k < HID ? fld.geo.weight[1][k] : (k == HID ? fld.geo.bias[1][0] : 0.0f);
    __syncthreads();
  }
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  float* scr = scr_all + wave * kScrPerWave;
  f32x16 aG1[I::HT][I::IT], aG2[1][I::HT], aSdf[I::HT];
  float bG1[I::HT], bG2[1], bSdf = 0.0f;
#pragma unroll
  for (int t = 0; t < I::HT; ++t) { zero_tiles(aG1[t]); bG1[t] = 0.0f; }
  zero_tiles(aG2[0]); zero_tiles(aSdf);
  bG2[0] = 0.0f;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  // the next tile's features and its d_e / d_sdf rows from the feature half are requested one tile ahead
  auto foff = [&](int k) { const int Fq = FW > 0 ? FW : F; return (int64_t)(k / Fq) * sl + (k % Fq); };
  f32x16 xn[I::IT], wn;
  float wn_sdf = 0.0f;
  auto request = [&](int64_t t) {
    const int64_t s_ = t * 32 + i;
    const bool v = t < tiles && s_ < n;
    load_rows<IN>(xn, feats + (v ? s_ * sn : 0), v, h, foff);
    const float* wt = ws + (t < tiles ? t : 0) * kWsTile;
#pragma unroll
    for (int r = 0; r < 16; ++r) wn[r] = wt[r * 64 + lane];
    wn_sdf = wt[16 * 64 + lane];
  };
  request((int64_t)blockIdx.x * 4 + wave);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[I::IT], h1[I::HT], d_e[1], d_h1[I::HT], d_x0[I::IT];
#pragma unroll
    for (int t = 0; t < I::IT; ++t) x0[t] = xn[t];
    d_e[0] = wn;
    const float d_sdf = wn_sdf;
    request(tile + (int64_t)gridDim.x * 4);
    dense_fwd<IN, HID, true>(x0, h1, lw + I::oG1, i, h);
    if (h == 0) bSdf += d_sdf;
    // mlp_geo.layers[1]: rows 1..C -> e (MFMA), row 0 -> sdf (per-lane partial products, reduced at the end)
    dense_bwd_dw_reg<HID, kC>(d_e, h1, aG2, bG2, scr, i, h);
    dense_bwd_dx<HID, kC>(d_e, d_h1, lw + I::oG2, i, h);
#pragma unroll
    for (int t = 0; t < I::HT; ++t)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int k = t * 32 + rowmap(s, 0) + 4 * h;
        aSdf[t][s] += d_sdf * h1[t][s];
        if (k < HID) d_h1[t][s] += lw[I::oSdf + k] * d_sdf;
      }
    relu_mask<HID>(d_h1, h1);
    dense_bwd_dw_reg<IN, HID>(d_h1, x0, aG1, bG1, scr, i, h);  // mlp_geo.layers[0]
    dense_bwd_dx<IN, HID>(d_h1, d_x0, lw + I::oG1, i, h);
    if (valid) {
      float* gf = g_feats + smp * sn;
#pragma unroll
      for (int kt = 0; kt < I::IT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = kt * 32 + rowmap(r, 0) + 4 * h;
          if (k < IN) gf[foff(k)] = d_x0[kt][r];
        }
    }
  }
  // sdf row: sum the per-lane partials over the 32 sample lanes of each half
#pragma unroll
  for (int t = 0; t < I::HT; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float v = aSdf[t][s];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, NR_WAVE);
      aSdf[t][s] = v;
    }
  bSdf = nr_wave_sum(bSdf);
  float* img = scr_all;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      const bool first = w == 0;
      merge_dw<IN, HID>(aG1, bG1, img + oG1, first, i, h);
      merge_dw<HID, kC>(aG2, bG2, img + oG2, first, i, h);
      if (i == 0) {
#pragma unroll
        for (int t = 0; t < I::HT; ++t)
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const int k = t * 32 + rowmap(s, 0) + 4 * h;
            if (k < HID) img[oSdf + k] = first ? aSdf[t][s] : img[oSdf + k] + aSdf[t][s];
          }
        if (h == 0) img[oSdf + HID] = first ? bSdf : img[oSdf + HID] + bSdf;
      }
    }
    __syncthreads();
  }
  float* out = slab + (int64_t)blockIdx.x * I::G_TOTAL + I::gG1;
  for (int k = HID + 1 + threadIdx.x; k < I::SDF; k += blockDim.x) img[oSdf + k] = 0.0f;  // padding
  __syncthreads();
  for (int k = threadIdx.x; k < kImg; k += blockDim.x) out[k] = img[k];
}

// destination of gradient-image element e in the torch-layout gradients (nullptr: padding)
template <int IN, int HID>
__device__ __forceinline__ float* grad_dst(int e, const nr_field_grads_t& g) {
  using I = FieldImage<IN, HID>;
  auto dense = [&](int j, int K, int M, float* w, float* b, int row0) -> float* {  // [M][K] then [M]
    return j < M * K ? w + (int64_t)row0 * K + j : b + row0 + (j - M * K);
  };
  if (e < I::gG2) return dense(e - I::gG1, IN, HID, g.geo.weight[0], g.geo.bias[0], 0);
  if (e < I::gSdf) return dense(e - I::gG2, HID, kC, g.geo.weight[1], g.geo.bias[1], 1);
  if (e < I::gF1) {
    const int j = e - I::gSdf;
    return j < HID ? g.geo.weight[1] + j : (j == HID ? g.geo.bias[1] : nullptr);
  }
  if (e < I::gF2) return dense(e - I::gF1, kC + kSH, HID, g.feat.weight[0], g.feat.bias[0], 0);
  if (e < I::gF3) return dense(e - I::gF2, HID, HID, g.feat.weight[1], g.feat.bias[1], 0);
  if (e < I::gBeta) return dense(e - I::gF3, HID, kC, g.feat.weight[2], g.feat.bias[2], 0);
  return e == I::gBeta ? g.beta : nullptr;
}

// grads += sum over the blocks' slabs (one writer per element: plain read-modify-write).  Block =
// 64 elements x 16 slab groups: every thread adds 16 slabs with independent loads, LDS combines the groups.
template <int IN, int HID>
__global__ void __launch_bounds__(1024)
field_grad_reduce_kernel(const float* __restrict__ slab, int n_slabs, nr_field_grads_t grads) {
  using I = FieldImage<IN, HID>;
  __shared__ float part[16][64];
  const int ex = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + ex;
  float sum = 0.0f;
  if (e < I::G_TOTAL) {
#pragma unroll 8
    for (int b = g; b < n_slabs; b += 16) sum += slab[(int64_t)b * I::G_TOTAL + e];
  }
  part[g][ex] = sum;
  __syncthreads();
  if (g == 0 && e < I::G_TOTAL) {
#pragma unroll
    for (int k = 1; k < 16; ++k) sum += part[k][ex];
    float* dst = grad_dst<IN, HID>(e, grads);
    if (dst != nullptr) *dst += sum;
  }
}

// weight image for load_field_weights' fast path
template <int IN, int HID>
__global__ void __launch_bounds__(1024)
field_pack_kernel(nr_field_t fld, float* __restrict__ image) {
  using I = FieldImage<IN, HID>;
  __shared__ __attribute__((aligned(16))) float lw[I::W_TOTAL];
  fld.packed = nullptr;
  load_field_weights<IN, HID>(lw, fld);
  for (int k = threadIdx.x; k < I::W_TOTAL; k += blockDim.x) image[k] = lw[k];
}

// ---- generic MLP (drop-in for field_components/mlp.py:MLP, e.g. the lidar decoder 48->32->32->2) ----
// Runs zero-padded on <KP, HP, MP> in {32,64}^3 with NL in {2,3} Linear layers.
template <int KP, int HP, int MP, int NL>
struct MlpImage {
  using L0 = Layer<KP, HP>;
  using L1 = Layer<HP, HP>;   // only when NL == 3
  using L2 = Layer<HP, MP>;
  static constexpr int o0 = 0, o1 = o0 + L0::SIZE, o2 = o1 + (NL == 3 ? L1::SIZE : 0), W_TOTAL = o2 + L2::SIZE;
  static constexpr int g0 = 0, g1 = g0 + L0::G_SIZE, g2 = g1 + (NL == 3 ? L1::G_SIZE : 0), G_TOTAL = g2 + L2::G_SIZE;
};

template <int KP, int HP, int MP, int NL>
__device__ __forceinline__ void load_mlp_weights(float* lw, const nr_mlp_t& m) {
  using I = MlpImage<KP, HP, MP, NL>;
  load_layer<KP, HP>(lw + I::o0, m.weight[0], m.bias[0], 0, m.in_dim, m.width);
  if (NL == 3) load_layer<HP, HP>(lw + I::o1, m.weight[1], m.bias[1], 0, m.width, m.width);
  load_layer<HP, MP>(lw + I::o2, m.weight[NL - 1], m.bias[NL - 1], 0, m.width, m.out_dim);
  __syncthreads();
}

template <int KP, int HP, int MP, int NL>
__global__ void __launch_bounds__(256)
mlp_fwd_kernel(nr_mlp_t mlp, const float* __restrict__ x, int64_t n, float* __restrict__ y) {
  using I = MlpImage<KP, HP, MP, NL>;
  __shared__ float lw[I::W_TOTAL];
  load_mlp_weights<KP, HP, MP, NL>(lw, mlp);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int in_dim = mlp.in_dim, out_dim = mlp.out_dim;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[KP / 32], a1[HP / 32], a2[HP / 32], out[MP / 32];
#pragma unroll
    for (int kt = 0; kt < KP / 32; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kt * 32 + rowmap(r, 0) + 4 * h;
        x0[kt][r] = (valid && k < in_dim) ? x[smp * in_dim + k] : 0.0f;
      }
    dense_fwd<KP, HP, true>(x0, a1, lw + I::o0, i, h);
    if (NL == 3) {
      dense_fwd<HP, HP, true>(a1, a2, lw + I::o1, i, h);
      dense_fwd<HP, MP, false>(a2, out, lw + I::o2, i, h);
    } else {
      dense_fwd<HP, MP, false>(a1, out, lw + I::o2, i, h);
    }
    if (valid) {
#pragma unroll
      for (int mt = 0; mt < MP / 32; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mt * 32 + rowmap(r, 0) + 4 * h;
          if (m < out_dim) y[smp * out_dim + m] = out[mt][r];
        }
    }
  }
}

template <int KP, int HP, int MP, int NL>
__global__ void __launch_bounds__(256)
mlp_bwd_kernel(nr_mlp_t mlp, const float* __restrict__ x, const float* __restrict__ g_y, int64_t n,
               float* __restrict__ g_x, nr_mlp_grads_t grads) {
  using I = MlpImage<KP, HP, MP, NL>;
  __shared__ float lw[I::W_TOTAL];
  __shared__ float lg[I::G_TOTAL];
  __shared__ float scr[4][2 * kScrTile + 32];
  for (int k = threadIdx.x; k < I::G_TOTAL; k += blockDim.x) lg[k] = 0.0f;
  load_mlp_weights<KP, HP, MP, NL>(lw, mlp);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  float* scrA = scr[wave];
  float* scrB = scrA + kScrTile;
  float* scrE = scrB + kScrTile;
  const int in_dim = mlp.in_dim, out_dim = mlp.out_dim;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[KP / 32], a1[HP / 32], a2[HP / 32], d_out[MP / 32], d_a2[HP / 32], d_a1[HP / 32], d_x0[KP / 32];
#pragma unroll
    for (int kt = 0; kt < KP / 32; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kt * 32 + rowmap(r, 0) + 4 * h;
        x0[kt][r] = (valid && k < in_dim) ? x[smp * in_dim + k] : 0.0f;
      }
#pragma unroll
    for (int mt = 0; mt < MP / 32; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + rowmap(r, 0) + 4 * h;
        d_out[mt][r] = (valid && m < out_dim) ? g_y[smp * out_dim + m] : 0.0f;
      }
    dense_fwd<KP, HP, true>(x0, a1, lw + I::o0, i, h);
    if (NL == 3) {
      dense_fwd<HP, HP, true>(a1, a2, lw + I::o1, i, h);
      dense_bwd_dw<HP, MP, false>(d_out, a2, lg + I::g2, 0.0f, nullptr, scrA, scrB, scrE, i, h);
      dense_bwd_dx<HP, MP>(d_out, d_a2, lw + I::o2, i, h);
      relu_mask<HP>(d_a2, a2);
      dense_bwd_dw<HP, HP, false>(d_a2, a1, lg + I::g1, 0.0f, nullptr, scrA, scrB, scrE, i, h);
      dense_bwd_dx<HP, HP>(d_a2, d_a1, lw + I::o1, i, h);
    } else {
      dense_bwd_dw<HP, MP, false>(d_out, a1, lg + I::g2, 0.0f, nullptr, scrA, scrB, scrE, i, h);
      dense_bwd_dx<HP, MP>(d_out, d_a1, lw + I::o2, i, h);
    }
    relu_mask<HP>(d_a1, a1);
    dense_bwd_dw<KP, HP, false>(d_a1, x0, lg + I::g0, 0.0f, nullptr, scrA, scrB, scrE, i, h);
    if (g_x != nullptr) {
      dense_bwd_dx<KP, HP>(d_a1, d_x0, lw + I::o0, i, h);
      if (valid) {
#pragma unroll
        for (int kt = 0; kt < KP / 32; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = kt * 32 + rowmap(r, 0) + 4 * h;
            if (k < in_dim) g_x[smp * in_dim + k] = d_x0[kt][r];
          }
      }
    }
  }
  __syncthreads();
  flush_layer_grads<KP, HP>(lg + I::g0, grads.weight[0], grads.bias[0], 0, mlp.in_dim, mlp.width);
  if (NL == 3) flush_layer_grads<HP, HP>(lg + I::g1, grads.weight[1], grads.bias[1], 0, mlp.width, mlp.width);
  flush_layer_grads<HP, MP>(lg + I::g2, grads.weight[NL - 1], grads.bias[NL - 1], 0, mlp.width, mlp.out_dim);
}

int check_mlp(const nr_mlp_t* m) {
  if (!m || (m->num_layers != 2 && m->num_layers != 3)) return NR_EINVAL;
  if (m->in_dim < 1 || m->in_dim > 64 || m->width < 1 || m->width > 64 || m->out_dim < 1 || m->out_dim > 64) return NR_EINVAL;
  for (int l = 0; l < m->num_layers; ++l) if (!m->weight[l] || !m->bias[l]) return NR_EINVAL;
  return 0;
}

#define NR_MLP_DISPATCH(KERNEL, ...)                                                                         \
  do {                                                                                                       \
    const int kp = mlp->in_dim > 32, hp = mlp->width > 32, mp = mlp->out_dim > 32, nl3 = mlp->num_layers == 3; \
    const int key = kp | (hp << 1) | (mp << 2) | (nl3 << 3);                                                 \
    switch (key) {                                                                                           \
      case 0: hipLaunchKernelGGL((KERNEL<32, 32, 32, 2>), __VA_ARGS__); break;                              \
      case 1: hipLaunchKernelGGL((KERNEL<64, 32, 32, 2>), __VA_ARGS__); break;                              \
      case 2: hipLaunchKernelGGL((KERNEL<32, 64, 32, 2>), __VA_ARGS__); break;                              \
      case 3: hipLaunchKernelGGL((KERNEL<64, 64, 32, 2>), __VA_ARGS__); break;                              \
      case 4: hipLaunchKernelGGL((KERNEL<32, 32, 64, 2>), __VA_ARGS__); break;                              \
      case 5: hipLaunchKernelGGL((KERNEL<64, 32, 64, 2>), __VA_ARGS__); break;                              \
      case 6: hipLaunchKernelGGL((KERNEL<32, 64, 64, 2>), __VA_ARGS__); break;                              \
      case 7: hipLaunchKernelGGL((KERNEL<64, 64, 64, 2>), __VA_ARGS__); break;                              \
      case 8: hipLaunchKernelGGL((KERNEL<32, 32, 32, 3>), __VA_ARGS__); break;                              \
      case 9: hipLaunchKernelGGL((KERNEL<64, 32, 32, 3>), __VA_ARGS__); break;                              \
      case 10: hipLaunchKernelGGL((KERNEL<32, 64, 32, 3>), __VA_ARGS__); break;                             \
      case 11: hipLaunchKernelGGL((KERNEL<64, 64, 32, 3>), __VA_ARGS__); break;                             \
      case 12: hipLaunchKernelGGL((KERNEL<32, 32, 64, 3>), __VA_ARGS__); break;                             \
      case 13: hipLaunchKernelGGL((KERNEL<64, 32, 64, 3>), __VA_ARGS__); break;                             \
      case 14: hipLaunchKernelGGL((KERNEL<32, 64, 64, 3>), __VA_ARGS__); break;                             \
      default: hipLaunchKernelGGL((KERNEL<64, 64, 64, 3>), __VA_ARGS__); break;                             \
    }                                                                                                        \
  } while (0)

}  // namespace

extern "C" int nr_field_fwd(const nr_field_t* field, const float* feats, int64_t sn, int64_t sl, int F,
                            const float* dirs, int S, int rows_sample_major, int64_t n, float* feature, float* sdf,
                            float* alpha, nr_stream_t stream) {
  if (n == 0) return 0;
  int hid = 0;
  if (check_field(field, &hid) != 0 || !feats || !dirs || !feature || !sdf || !alpha || S < 0 || F < 1 || n < 0) return NR_EINVAL;
  if (rows_sample_major < 0 || (rows_sample_major && (S < 1 || n % S != 0 || rows_sample_major > n / S))) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, 32);
  unsigned blocks = (unsigned)(nr_cdiv(tiles, 4) < 512 ? nr_cdiv(tiles, 4) : 512);
  if (const int v = nr_tuning().field_fwd_blocks; v > 0 && (int64_t)v < nr_cdiv(tiles, 4)) blocks = (unsigned)v;
  if (field->dtype != NR_DTYPE_F32)  // bf16 / fp16 operands (mlp_lp.hip)
    return field_fwd_lp(field, hid, feats, sn, sl, F, dirs, S, rows_sample_major, n, feature, sdf, alpha, blocks, nr_s(stream));
#define LAUNCH_FWD2(HIDC, FWC, ST)                                                                                    \
  hipLaunchKernelGGL((field_fwd_kernel<32, HIDC, FWC, ST>), dim3(blocks), dim3(256), 0, nr_s(stream), *field, feats, sn, sl, \
                     F, dirs, S, rows_sample_major, n, feature, sdf, alpha)
#define LAUNCH_FWD(FWC)                                                                                               \
  {                                                                                                                    \
    if (hid == 32) {                                                                                                   \
      if (field->stash) LAUNCH_FWD2(32, FWC, true); else LAUNCH_FWD2(32, FWC, false);                                  \
    } else {                                                                                                           \
      if (field->stash) LAUNCH_FWD2(64, FWC, true); else LAUNCH_FWD2(64, FWC, false);                                  \
    }                                                                                                                  \
  }
  if (F == 2) LAUNCH_FWD(2) else if (F == 4) LAUNCH_FWD(4) else LAUNCH_FWD(0)
#undef LAUNCH_FWD
#undef LAUNCH_FWD2
  NR_LAUNCH_CHECK();
  return 0;
}


extern "C" int nr_field_fwd_gather(const nr_field_t* field, const float* x01, const float* std01, const float* table,
                                   const float* scalings, int L, int F, int log2T, float* feats_out, int64_t sl, const float* dirs,
                                   int S, int rows_sample_major, int64_t n, float* feature, float* sdf, float* alpha,
                                   nr_stream_t stream) {
  if (n == 0) return 0;
  int hid = 0;
  if (check_field(field, &hid) != 0 || !x01 || !std01 || !table || !scalings || !dirs || !feature || !sdf || !alpha || S < 0 || n < 0 ||
      log2T < 1 || log2T > 30)
    return NR_EINVAL;
  if (rows_sample_major < 0 || (rows_sample_major && (S < 1 || n % S != 0 || rows_sample_major > n / S))) return NR_EINVAL;
  // built for NeuRadar's main grid feeding the 32-wide stack on 16-bit operands; anything else: nr_hash_encode_fwd + nr_field_fwd
  if (L != 8 || F != 4 || hid != 32 || field->dtype == NR_DTYPE_F32) return NR_EINVAL;
  if (feats_out != nullptr && (((uintptr_t)feats_out & 15u) != 0 || (sl & 3) != 0)) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, 32);
  unsigned blocks = (unsigned)(nr_cdiv(tiles, 4) < 512 ? nr_cdiv(tiles, 4) : 512);
  if (const int v = nr_tuning().field_fwd_blocks; v > 0 && (int64_t)v < nr_cdiv(tiles, 4)) blocks = (unsigned)v;
  return field_fwd_gather_lp(field, hid, x01, std01, table, scalings, log2T, feats_out, sl, dirs, S, rows_sample_major, n, feature,
                             sdf, alpha, blocks, nr_s(stream));
}

extern "C" int nr_field_bwd(const nr_field_t* field, const float* feats, int64_t sn, int64_t sl, int F,
                            const float* dirs, int S, int rows_sample_major, int64_t n, const float* g_feature,
                            const float* g_alpha, const float* g_sdf, float* g_feats, const nr_field_grads_t* grads,
                            float* workspace, nr_stream_t stream) {
  if (n == 0) return 0;
  int hid = 0;
  if ((((uintptr_t)workspace | (uintptr_t)g_feature) & 15u) != 0) return NR_EINVAL;  // read / written with 16-byte accesses
  if (check_field(field, &hid) != 0 || !feats || !dirs || !g_feature || !g_alpha || !g_feats || !workspace || S < 0 ||
      F < 1 || n < 0)
    return NR_EINVAL;
  if (rows_sample_major < 0 || (rows_sample_major && (S < 1 || n % S != 0 || rows_sample_major > n / S))) return NR_EINVAL;
  if (grads) {  // NULL: the per-block gradient slabs stay in the workspace for a later nr_field_grad_reduce
    for (int l = 0; l < 2; ++l) if (!grads->geo.weight[l] || !grads->geo.bias[l]) return NR_EINVAL;
    for (int l = 0; l < 3; ++l) if (!grads->feat.weight[l] || !grads->feat.bias[l]) return NR_EINVAL;
  }
  const unsigned blocks = field_bwd_blocks(n, field, hid);
  float* slab = workspace + ws_floats(n);  // [blocks][G_TOTAL] after the d_e / d_sdf tiles
  if (field->dtype != NR_DTYPE_F32) {  // bf16 / fp16 operands (mlp_lp.hip); same workspace / slab layout, same reduce
    const int rc = field_bwd_lp(field, hid, feats, sn, sl, F, dirs, S, rows_sample_major, n, g_feature, g_alpha, g_sdf, g_feats,
                                workspace, slab, blocks, nr_s(stream));
    if (rc != 0) return rc;
    if (grads) {
      if (hid == 32)
        hipLaunchKernelGGL((field_grad_reduce_kernel<32, 32>), dim3((unsigned)nr_cdiv(FieldImage<32, 32>::G_TOTAL, 64)), dim3(1024),
                           0, nr_s(stream), slab, (int)blocks, *grads);
      else
        hipLaunchKernelGGL((field_grad_reduce_kernel<32, 64>), dim3((unsigned)nr_cdiv(FieldImage<32, 64>::G_TOTAL, 64)), dim3(1024),
                           0, nr_s(stream), slab, (int)blocks, *grads);
      NR_LAUNCH_CHECK();
    }
    return 0;
  }
#define LAUNCH_BWD(HIDC, FWC)                                                                                          \
  {                                                                                                                     \
    using I = FieldImage<32, HIDC>;                                                                                     \
    if (field->stash)                                                                                                   \
      hipLaunchKernelGGL((field_bwd_feat_kernel<32, HIDC, FWC, true>), dim3(blocks), dim3(256), 0, nr_s(stream), *field, feats, \
                         sn, sl, F, dirs, S, rows_sample_major, n, g_feature, g_alpha, g_sdf, workspace, slab);         \
    else                                                                                                                \
      hipLaunchKernelGGL((field_bwd_feat_kernel<32, HIDC, FWC, false>), dim3(blocks), dim3(256), 0, nr_s(stream), *field,   \
                         feats, sn, sl, F, dirs, S, rows_sample_major, n, g_feature, g_alpha, g_sdf, workspace, slab);  \
    hipLaunchKernelGGL((field_bwd_geo_kernel<32, HIDC, FWC>), dim3(blocks), dim3(256), 0, nr_s(stream), *field, feats, sn,  \
                       sl, F, n, workspace, g_feats, slab);                                                             \
    if (grads)                                                                                                          \
      hipLaunchKernelGGL((field_grad_reduce_kernel<32, HIDC>), dim3((unsigned)nr_cdiv(I::G_TOTAL, 64)), dim3(1024), 0,   \
                         nr_s(stream), slab, (int)blocks, *grads);                                                      \
  }
  if (hid == 32) {
    if (F == 2) LAUNCH_BWD(32, 2) else if (F == 4) LAUNCH_BWD(32, 4) else LAUNCH_BWD(32, 0)
  } else {
    if (F == 2) LAUNCH_BWD(64, 2) else if (F == 4) LAUNCH_BWD(64, 4) else LAUNCH_BWD(64, 0)
  }
#undef LAUNCH_BWD
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_field_grad_reduce(const nr_field_t* field, const float* workspace, int64_t n, const nr_field_grads_t* grads,
                                    nr_stream_t stream) {
  if (n == 0) return 0;
  int hid = 0;
  if (check_field(field, &hid) != 0 || !workspace || !grads || n < 0) return NR_EINVAL;
  for (int l = 0; l < 2; ++l) if (!grads->geo.weight[l] || !grads->geo.bias[l]) return NR_EINVAL;
  for (int l = 0; l < 3; ++l) if (!grads->feat.weight[l] || !grads->feat.bias[l]) return NR_EINVAL;
  const unsigned blocks = field_bwd_blocks(n, field, hid);
  const float* slab = workspace + ws_floats(n);
  if (hid == 32)
    hipLaunchKernelGGL((field_grad_reduce_kernel<32, 32>), dim3((unsigned)nr_cdiv(FieldImage<32, 32>::G_TOTAL, 64)), dim3(1024), 0,
                       nr_s(stream), slab, (int)blocks, *grads);
  else
    hipLaunchKernelGGL((field_grad_reduce_kernel<32, 64>), dim3((unsigned)nr_cdiv(FieldImage<32, 64>::G_TOTAL, 64)), dim3(1024), 0,
                       nr_s(stream), slab, (int)blocks, *grads);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t nr_field_bwd_workspace_floats(const nr_field_t* field, int64_t n) {
  int hid = 0;
  if (check_field(field, &hid) != 0 || n < 0) return -1;
  const int64_t g_total = hid == 32 ? FieldImage<32, 32>::G_TOTAL : FieldImage<32, 64>::G_TOTAL;
  return ws_floats(n) + kMaxBwdBlocks * g_total;  // d_e / d_sdf tiles + one gradient slab per block
}

extern "C" int64_t nr_field_stash_floats(const nr_field_t* field, int64_t n) {
  int hid = 0;
  if (check_field(field, &hid) != 0 || n < 0) return -1;
  if (field->dtype != NR_DTYPE_F32) return 0;  // the reduced-precision backward recomputes the forward
  return nr_cdiv(n, 32) * (hid == 32 ? Stash<32>::kTile : Stash<64>::kTile);
}

extern "C" int64_t nr_field_image_floats(const nr_field_t* field) {
  int hid = 0;
  if (field && field->dtype != NR_DTYPE_F32 && field->packed == nullptr) {  // size query before the image exists
    nr_field_t probe = *field;
    probe.packed = reinterpret_cast<const float*>(&probe);
    if (check_field(&probe, &hid) != 0) return -1;
    return field_image_bytes_lp(hid) / 4;
  }
  if (check_field(field, &hid) != 0) return -1;
  if (field->dtype != NR_DTYPE_F32) return field_image_bytes_lp(hid) / 4;
  return hid == 32 ? FieldImage<32, 32>::W_TOTAL : FieldImage<32, 64>::W_TOTAL;
}

extern "C" int nr_field_pack(const nr_field_t* field, float* image, nr_stream_t stream) {
  int hid = 0;
  if (!field || !image || ((uintptr_t)image & 15u) != 0) return NR_EINVAL;
  if (field->dtype != NR_DTYPE_F32) {
    nr_field_t probe = *field;
    probe.packed = image;
    if (check_field(&probe, &hid) != 0) return NR_EINVAL;
    return field_pack_lp(field, hid, image, nr_s(stream));
  }
  if (check_field(field, &hid) != 0) return NR_EINVAL;
  if (hid == 32)
    hipLaunchKernelGGL((field_pack_kernel<32, 32>), dim3(1), dim3(1024), 0, nr_s(stream), *field, image);
  else
    hipLaunchKernelGGL((field_pack_kernel<32, 64>), dim3(1), dim3(1024), 0, nr_s(stream), *field, image);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_mlp_fwd(const nr_mlp_t* mlp, const float* x, int64_t n, float* y, nr_stream_t stream) {
  if (n == 0) return 0;
  if (check_mlp(mlp) != 0 || !x || !y || n < 0) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, 32);
  const unsigned blocks = (unsigned)(nr_cdiv(tiles, 4) < 512 ? nr_cdiv(tiles, 4) : 512);
  NR_MLP_DISPATCH(mlp_fwd_kernel, dim3(blocks), dim3(256), 0, nr_s(stream), *mlp, x, n, y);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_mlp_bwd(const nr_mlp_t* mlp, const float* x, const float* g_y, int64_t n, float* g_x,
                          const nr_mlp_grads_t* grads, nr_stream_t stream) {
  if (n == 0) return 0;
  if (check_mlp(mlp) != 0 || !x || !g_y || !grads || n < 0) return NR_EINVAL;
  for (int l = 0; l < mlp->num_layers; ++l) if (!grads->weight[l] || !grads->bias[l]) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, 32);
  const unsigned blocks = (unsigned)(nr_cdiv(tiles, 4) < 256 ? nr_cdiv(tiles, 4) : 256);
  NR_MLP_DISPATCH(mlp_bwd_kernel, dim3(blocks), dim3(256), 0, nr_s(stream), *mlp, x, g_y, n, g_x, *grads);
  NR_LAUNCH_CHECK();
  return 0;
}
