// Alpha compositing (fwd + bwd) and expected depth.  One wavefront per ray: lanes = samples for the
// transmittance scan (wave shuffles, no LDS), lanes = channels for the feature reduction.
#include "nr_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

// exclusive product of (1 - alpha) over earlier lanes
__device__ __forceinline__ float excl_transmittance(float one_minus, int lane) {
  const float incl = nr_wave_incl_prod(one_minus);
  const float up = __shfl_up(incl, 1, NR_WAVE);
  return lane == 0 ? 1.0f : up;
}

__global__ void __launch_bounds__(256)
composite_fwd_kernel(const float* __restrict__ alpha, const float* __restrict__ feature, const float* __restrict__ euclid,
                     int64_t n_rays, int S, int C, float* __restrict__ weights, float* __restrict__ accumulation,
                     float* __restrict__ features, float* __restrict__ depth) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const float a = lane < S ? alpha[ray * S + lane] : 0.0f;
  const float T = excl_transmittance(1.0f - a, lane);
  float w = lane < S ? a * T : 0.0f;                 // nerfacc.render_weight_from_alpha (neuradar.py:1016)
  const float acc = nr_wave_sum(w);                  // AccumulationRenderer (renderers.py:349)
  if (lane == S - 1) w = w + 1.0f - acc;             // sky fix-up (neuradar.py:508)
  if (lane < S) weights[ray * S + lane] = w;
  const float* e = euclid + ray * (S + 1);
  float d = lane < S - 1 ? w * ((e[lane] + e[lane + 1]) / 2.0f) : 0.0f;  // render_depth_simple, sky dropped
  d = nr_wave_sum(d);
  if (lane == 0) {
    accumulation[ray] = acc;
    depth[ray] = d;
  }
  // FeatureRenderer (renderers.py:85): lanes = channels, loop over samples
  for (int c0 = 0; c0 < C; c0 += NR_WAVE) {
    const int c = c0 + lane;
    float sum = 0.0f;
    for (int s = 0; s < S; ++s) {
      const float ws = __shfl(w, s, NR_WAVE);
      if (c < C) sum += feature[(ray * S + s) * C + c] * ws;
    }
    if (c < C) features[ray * C + c] = sum;
  }
}

__global__ void __launch_bounds__(256)
composite_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ feature, const float* __restrict__ euclid,
                     const float* __restrict__ weights, const float* __restrict__ g_features,
                     const float* __restrict__ g_depth, const float* __restrict__ g_acc, const float* __restrict__ g_w,
                     int64_t n_rays, int S, int C, float* __restrict__ g_alpha, float* __restrict__ g_feature) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const float wfin = lane < S ? weights[ray * S + lane] : 0.0f;  // weights AFTER the sky fix-up
  // G_s = dL/d(final weight s):  sum_c gF_c f[s][c] + gD * mid_s (s < S-1) + gW_s
  float G = 0.0f;
  for (int c0 = 0; c0 < C; c0 += NR_WAVE) {
    const int c = c0 + lane;
    const float gf = (c < C && g_features) ? g_features[ray * C + c] : 0.0f;
    for (int s = 0; s < S; ++s) {
      const float ws = __shfl(wfin, s, NR_WAVE);
      float part = 0.0f;
      if (c < C) {
        const int64_t o = (ray * S + s) * C + c;
        part = gf * feature[o];
        g_feature[o] = gf * ws;
      }
      part = nr_wave_sum(part);
      if (lane == s) G += part;
    }
  }
  const float* e = euclid + ray * (S + 1);
  if (g_depth && lane < S - 1) G += g_depth[ray] * ((e[lane] + e[lane + 1]) / 2.0f);
  if (g_w && lane < S) G += g_w[ray * S + lane];
  // final_j = w_j (j < S-1), final_{S-1} = w_{S-1} + 1 - sum_j w_j  =>  dw_j = G_j - G_{S-1} + gAcc
  const float G_last = __shfl(G, S - 1, NR_WAVE);
  const float dw = lane < S ? G - G_last + (g_acc ? g_acc[ray] : 0.0f) : 0.0f;
  // w_j = a_j T_j, T_j = prod_{i<j}(1-a_i):
  //   d a_j = T_j * (dw_j - R_j),  R_j = sum_{s>j} dw_s a_s prod_{j<i<s}(1-a_i)
  // R is the suffix scan of the affine maps R_j = B_j + A_j R_{j+1} with A_j = 1-a_{j+1},
  // B_j = dw_{j+1} a_{j+1}  (division-free, safe for alpha -> 1).
  const float a = lane < S ? alpha[ray * S + lane] : 0.0f;
  const float T = excl_transmittance(1.0f - a, lane);
  float A = __shfl_down(1.0f - a, 1, NR_WAVE), Bv = __shfl_down(dw * a, 1, NR_WAVE);
  if (lane >= S - 1) { A = 0.0f; Bv = 0.0f; }  // R_{S-1} = 0
#pragma unroll
  for (int o = 1; o < NR_WAVE; o <<= 1) {
    // compose f_lane with f_{lane+o}: (A,B) o (A',B') = (A*A', B + A*B')
    const float A2 = __shfl_down(A, o, NR_WAVE), B2 = __shfl_down(Bv, o, NR_WAVE);
    if (lane + o < NR_WAVE) {
      Bv = Bv + A * B2;
      A = A * A2;
    }
  }
  if (lane < S) g_alpha[ray * S + lane] = T * (dw - Bv);
}

// ---- fused render + loss + render-backward of the training step -------------------------------
// composite_fwd -> supervision loss -> distortion loss -> composite_bwd of ONE ray never leave its
// wavefront: the five launches (and the weights / g_features / g_depth / g_w round trips between
// them) become one.  Same arithmetic as the separate kernels, which remain the reference for it
// (tests/test_gpu_parity.py::test_fused_render_matches_separate_kernels).
// Lanes = samples for the scans; for the [S, C] feature block lane = (s parity, channel): 2 x 32
// coalesced 128-B rows per load, kept in registers between the forward sum and the backward dot.
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int KH>  // S <= 2 * KH
__global__ void __launch_bounds__(256)
render_train_kernel(const float* __restrict__ alpha, const float* __restrict__ feature, const float* __restrict__ euclid,
                    const float* __restrict__ spacing, const float* __restrict__ target_f,
                    const float* __restrict__ target_d, int64_t n_rays, int S, int C, float rgb_mult, float depth_mult,
                    float dist_mult, float* __restrict__ weights, float* __restrict__ accumulation,
                    float* __restrict__ features, float* __restrict__ depth, float* __restrict__ g_alpha,
                    float* __restrict__ g_feature, float* __restrict__ loss, const float* __restrict__ g_features_extra,
                    const float* __restrict__ g_depth_extra, nr_lidar_sup_t lidar) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  // ---- composite forward (neuradar.py:1016, renderers.py:349, neuradar.py:508, neurad.py:721-728) ----
  const float a = lane < S ? alpha[ray * S + lane] : 0.0f;
  const float T = excl_transmittance(1.0f - a, lane);
  float w = lane < S ? a * T : 0.0f;
  const float acc = nr_wave_sum(w);
  if (lane == S - 1) w = w + 1.0f - acc;
  if (lane < S) weights[ray * S + lane] = w;
  const float* e = euclid + ray * (S + 1);
  const float mid_e = lane < S - 1 ? (e[lane] + e[lane + 1]) / 2.0f : 0.0f;
  const float d = nr_wave_sum(lane < S - 1 ? w * mid_e : 0.0f);
  const int sh = lane >> 5, c = lane & 31;
  const bool cvalid = c < C;
  float fr[KH];  // feature[s][c] for s = 2k + sh
  float fsum = 0.0f;
  const float* frow = feature + ray * S * C + c;
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    const int s = 2 * k + sh;
    const float w0 = readlane_f(w, 2 * k), w1 = readlane_f(w, 2 * k + 1);  // lanes past S hold 0
    fr[k] = (cvalid && s < S) ? frow[(int64_t)s * C] : 0.0f;
    fsum += fr[k] * (sh ? w1 : w0);
  }
  fsum += nr_xor32_f(fsum);  // FeatureRenderer (renderers.py:85)
  if (lane == 0) {
    accumulation[ray] = acc;
    depth[ray] = d;
  }
  if (lane < C) features[ray * C + lane] = fsum;
  // ---- supervision: rgb_mult * mean((f - t)^2) + depth_mult * mean(|d - t|) ----
  const float kf = rgb_mult / (float)(n_rays * C), kd = depth_mult / (float)n_rays;
  const float diff = (cvalid && target_f != nullptr) ? fsum - target_f[ray * C + c] : 0.0f;
  float lossv = sh == 0 ? kf * diff * diff : 0.0f;
  float gF = 2.0f * kf * diff;  // d loss / d features[c], in both halves
  if (g_features_extra != nullptr && cvalid) gF += g_features_extra[ray * C + c];
  const float dd = target_d != nullptr ? d - target_d[ray] : 0.0f;
  if (lane == 0) lossv += kd * fabsf(dd);
  float gD = dd > 0.0f ? kd : (dd < 0.0f ? -kd : 0.0f);
  if (g_depth_extra != nullptr) gD += g_depth_extra[ray];
  // ---- distortion loss on the first S-1 samples (losses.py:137-157; sky sample dropped) ----
  const int n_used = S - 1;
  const bool on = lane < n_used;
  const float* sp = spacing + ray * (S + 1);
  const float c0 = on ? sp[lane] : 0.0f, c1 = on ? sp[lane + 1] : 0.0f;
  const float wi = on ? w : 0.0f;
  const float mid_c = (c1 + c0) / 2.0f;
  float inner = 0.0f;
#pragma unroll 1
  for (int j = 0; j < n_used; ++j) inner += readlane_f(wi, j) * fabsf(mid_c - readlane_f(mid_c, j));
  const float kdist = dist_mult / (float)n_rays;
  if (on) lossv += kdist * (wi * inner + wi * wi * (c1 - c0) / 3.0f);
  float gW = on ? kdist * (2.0f * inner + 2.0f * wi * (c1 - c0) / 3.0f) : 0.0f;
  if (lidar.is_lidar != nullptr && lidar.is_lidar[ray]) {  // carving: the weights away from the measured return (neuradar.py:537-541)
    const bool ret = lidar.did_return[ray] != 0;
    const bool close = ret ? fabsf(lidar.range[ray] - mid_e) < lidar.carving_epsilon : mid_e < lidar.non_return_distance;
    if (on && !close) {
      lossv += lidar.weight * wi * wi;
      gW += 2.0f * lidar.weight * wi;
    }
  }
  lossv = nr_wave_sum(lossv);
  if (lane == 0 && lossv != 0.0f) unsafeAtomicAdd(loss + nr_loss_slot_index(), lossv);
  // ---- composite backward ----
  // g_feature[s][c] = gF_c * w_s ; dot_s = sum_c gF_c f[s][c]: 32 values reduced over the 32 lanes of
  // each half by a halving butterfly (31 exchanges), after which lane (sh, j) holds dot_{2j+sh}
  float* grow = g_feature + ray * S * C + c;
  float v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < KH; ++k) {
    v[k] = gF * fr[k];
    const int s = 2 * k + sh;
    const float w0 = readlane_f(w, 2 * k), w1 = readlane_f(w, 2 * k + 1);
    if (cvalid && s < S) grow[(int64_t)s * C] = gF * (sh ? w1 : w0);
  }
#pragma unroll
  for (int half = 16; half >= 1; half >>= 1) {
    const bool upper = (lane & half) != 0;
#pragma unroll
    for (int k = 0; k < half; ++k) {
      float lo = v[k], hi = v[k + half];
      // opaque to the optimiser: otherwise the two selects become ONE dynamically indexed read of v[],
      // which it lowers to a 32-way compare/select chain per element (measured: 4,800 VALU per ray)
      asm volatile("" : "+v"(lo), "+v"(hi));
      const float send = upper ? lo : hi;
      const float keep = upper ? hi : lo;
      v[k] = keep + __shfl_xor(send, half, NR_WAVE);
    }
  }
  float G = __shfl(v[0], (lane & 1) * 32 + (lane >> 1), NR_WAVE);  // lane s <- dot_s
  if (lane >= S) G = 0.0f;
  G += gD * mid_e + gW;
  const float G_last = __shfl(G, S - 1, NR_WAVE);
  const float dw = lane < S ? G - G_last : 0.0f;
  float A = __shfl_down(1.0f - a, 1, NR_WAVE), Bv = __shfl_down(dw * a, 1, NR_WAVE);
  if (lane >= S - 1) { A = 0.0f; Bv = 0.0f; }
#pragma unroll
  for (int o = 1; o < NR_WAVE; o <<= 1) {
    const float A2 = __shfl_down(A, o, NR_WAVE), B2 = __shfl_down(Bv, o, NR_WAVE);
    if (lane + o < NR_WAVE) {
      Bv = Bv + A * B2;
      A = A * A2;
    }
  }
  if (lane < S) g_alpha[ray * S + lane] = T * (dw - Bv);
}

__global__ void __launch_bounds__(256)
depth_kernel(const float* __restrict__ weights, const float* __restrict__ euclid, int64_t n_rays, int S,
             float* __restrict__ depth) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const float* e = euclid + ray * (S + 1);
  float d = 0.0f;
  for (int s = lane; s < S; s += NR_WAVE) d += weights[ray * S + s] * ((e[s] + e[s + 1]) / 2.0f);
  d = nr_wave_sum(d);
  if (lane == 0) depth[ray] = d;
}

// ---- nerfacc's batched helpers by themselves (models/neuradar.py:1016-1022, models/neurad.py:727-728) ----
// One wavefront per ray, any S: chunks of 64 samples with the running transmittance carried from chunk to chunk.
// density mode (t_starts != nullptr): in = sigma, alpha = 1 - exp(-sigma * (t_end - t_start)).
__global__ void __launch_bounds__(256)
weights_from_alpha_fwd_kernel(const float* __restrict__ in, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
                              int64_t n_rays, int S, float* __restrict__ weights, float* __restrict__ trans,
                              float* __restrict__ alphas) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  float carry = 1.0f;
  for (int s0 = 0; s0 < S; s0 += NR_WAVE) {
    const int s = s0 + lane;
    const int64_t o = ray * S + s;
    float a = 0.0f;
    if (s < S) a = t_starts ? 1.0f - expf(-in[o] * (t_ends[o] - t_starts[o])) : in[o];
    const float incl = nr_wave_incl_prod(1.0f - a);
    const float up = __shfl_up(incl, 1, NR_WAVE);
    const float T = carry * (lane == 0 ? 1.0f : up);
    if (s < S) {
      weights[o] = a * T;
      trans[o] = T;
      if (alphas) alphas[o] = a;
    }
    carry *= __shfl(incl, NR_WAVE - 1, NR_WAVE);
  }
}

// w_i = a_i T_i, T_i = prod_{j<i}(1 - a_j).  With H_s = g_w_s a_s + g_T_s (everything that reaches T_s):
//   d a_j = g_w_j T_j - T_j R_j,   R_j = sum_{s>j} H_s prod_{j<i<s}(1 - a_i) = H_{j+1} + (1 - a_{j+1}) R_{j+1}
// (division-free: safe for alpha -> 1).  Chunks are walked from the far end with R carried across them.
__global__ void __launch_bounds__(256)
weights_from_alpha_bwd_kernel(const float* __restrict__ in, const float* __restrict__ t_starts, const float* __restrict__ t_ends,
                              const float* __restrict__ trans, const float* __restrict__ g_w, const float* __restrict__ g_T,
                              const float* __restrict__ g_a, int64_t n_rays, int S, float* __restrict__ g_in) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  float R_next = 0.0f, A_next = 0.0f, H_next = 0.0f;  // R, (1 - a), H of the first sample of the chunk behind this one
  for (int s0 = (S - 1) / NR_WAVE * NR_WAVE; s0 >= 0; s0 -= NR_WAVE) {
    const int s = s0 + lane;
    const int64_t o = ray * S + s;
    const bool on = s < S;
    float a = 0.0f, delta = 0.0f;
    if (on) {
      if (t_starts) {
        delta = t_ends[o] - t_starts[o];
        a = 1.0f - expf(-in[o] * delta);
      } else {
        a = in[o];
      }
    }
    const float gw = (on && g_w) ? g_w[o] : 0.0f;
    const float H = gw * a + ((on && g_T) ? g_T[o] : 0.0f);
    // affine map of lane j: R_j = B_j + A_j R_{j+1}, with (A_j, B_j) = (1 - a_{j+1}, H_{j+1}); the last lane's successor is
    // the first sample of the chunk behind
    float A = __shfl_down(1.0f - a, 1, NR_WAVE), Bv = __shfl_down(H, 1, NR_WAVE);
    if (lane == NR_WAVE - 1) { A = A_next; Bv = H_next; }
    if (s + 1 >= S) { A = 0.0f; Bv = 0.0f; }
#pragma unroll
    for (int k = 1; k < NR_WAVE; k <<= 1) {
      const float A2 = __shfl_down(A, k, NR_WAVE), B2 = __shfl_down(Bv, k, NR_WAVE);
      if (lane + k < NR_WAVE) {
        Bv = Bv + A * B2;
        A = A * A2;
      }
    }
    // lane j now maps R_{s0+64} (= R of the next chunk's first sample) to R_j
    const float R = Bv + A * R_next;
    if (on) {
      const float T = trans[o];
      float ga = T * (gw - R) + (g_a ? g_a[o] : 0.0f);
      g_in[o] = t_starts ? ga * delta * (1.0f - a) : ga;
    }
    R_next = __shfl(R, 0, NR_WAVE);
    A_next = __shfl(1.0f - a, 0, NR_WAVE);
    H_next = __shfl(H, 0, NR_WAVE);
  }
}

// accumulate_along_rays, batched branch: out[b][c] = sum_s w[b][s] * values[b][s][c] (values == nullptr: C = 1, sum_s w).
__global__ void __launch_bounds__(256)
accumulate_fwd_kernel(const float* __restrict__ w, const float* __restrict__ values, int64_t n_rays, int S, int C,
                      float* __restrict__ out) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  if (values == nullptr) {
    float sum = 0.0f;
    for (int s = lane; s < S; s += NR_WAVE) sum += w[ray * S + s];
    sum = nr_wave_sum(sum);
    if (lane == 0) out[ray] = sum;
    return;
  }
  for (int c0 = 0; c0 < C; c0 += NR_WAVE) {
    const int c = c0 + lane;
    float sum = 0.0f;
    for (int s0 = 0; s0 < S; s0 += NR_WAVE) {
      const float wl = s0 + lane < S ? w[ray * S + s0 + lane] : 0.0f;
      const int ns = min(NR_WAVE, S - s0);
      for (int k = 0; k < ns; ++k) {
        const float ws = __shfl(wl, k, NR_WAVE);
        if (c < C) sum += values[(ray * S + s0 + k) * C + c] * ws;
      }
    }
    if (c < C) out[ray * C + c] = sum;
  }
}

__global__ void __launch_bounds__(256)
accumulate_bwd_kernel(const float* __restrict__ w, const float* __restrict__ values, const float* __restrict__ g_out,
                      int64_t n_rays, int S, int C, float* __restrict__ g_w, float* __restrict__ g_values) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  if (values == nullptr) {
    const float g = g_out[ray];
    for (int s = lane; s < S; s += NR_WAVE) g_w[ray * S + s] = g;
    return;
  }
  for (int s0 = 0; s0 < S; s0 += NR_WAVE) {
    const float wl = s0 + lane < S ? w[ray * S + s0 + lane] : 0.0f;
    const int ns = min(NR_WAVE, S - s0);
    float gws = 0.0f;
    for (int c0 = 0; c0 < C; c0 += NR_WAVE) {
      const int c = c0 + lane;
      const float g = c < C ? g_out[ray * C + c] : 0.0f;
      for (int k = 0; k < ns; ++k) {
        const float ws = __shfl(wl, k, NR_WAVE);
        float part = 0.0f;
        if (c < C) {
          const int64_t o = (ray * S + s0 + k) * C + c;
          part = g * values[o];
          if (g_values) g_values[o] = g * ws;
        }
        part = nr_wave_sum(part);
        if (lane == k) gws += part;
      }
    }
    if (g_w && s0 + lane < S) g_w[ray * S + s0 + lane] = gws;
  }
}

}  // namespace

extern "C" int nr_render_weights_fwd(const float* alphas_or_sigmas, const float* t_starts, const float* t_ends, int64_t n_rays,
                                     int S, float* weights, float* transmittance, float* alphas, nr_stream_t stream) {
  if (n_rays == 0 || S == 0) return 0;
  if (!alphas_or_sigmas || !weights || !transmittance || (t_starts == nullptr) != (t_ends == nullptr) || S < 0 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(weights_from_alpha_fwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     alphas_or_sigmas, t_starts, t_ends, n_rays, S, weights, transmittance, alphas);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_render_weights_bwd(const float* alphas_or_sigmas, const float* t_starts, const float* t_ends,
                                     const float* transmittance, const float* grad_weights, const float* grad_transmittance,
                                     const float* grad_alphas, int64_t n_rays, int S, float* grad_in, nr_stream_t stream) {
  if (n_rays == 0 || S == 0) return 0;
  if (!alphas_or_sigmas || !transmittance || !grad_in || (t_starts == nullptr) != (t_ends == nullptr) || S < 0 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(weights_from_alpha_bwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     alphas_or_sigmas, t_starts, t_ends, transmittance, grad_weights, grad_transmittance, grad_alphas, n_rays, S,
                     grad_in);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_accumulate_fwd(const float* weights, const float* values, int64_t n_rays, int S, int C, float* out,
                                 nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!weights || !out || S < 0 || n_rays < 0 || (values && C < 1)) return NR_EINVAL;
  hipLaunchKernelGGL(accumulate_fwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream), weights,
                     values, n_rays, S, C, out);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_accumulate_bwd(const float* weights, const float* values, const float* grad_out, int64_t n_rays, int S, int C,
                                 float* grad_weights, float* grad_values, nr_stream_t stream) {
  if (n_rays == 0 || S == 0) return 0;
  if (!weights || !grad_out || S < 0 || n_rays < 0 || (values && C < 1) || (!values && !grad_weights)) return NR_EINVAL;
  hipLaunchKernelGGL(accumulate_bwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream), weights,
                     values, grad_out, n_rays, S, C, grad_weights, grad_values);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_composite_fwd(const float* alpha, const float* feature, const float* euclid, int64_t n_rays, int S,
                                int C, float* weights, float* accumulation, float* features, float* depth,
                                nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!alpha || !feature || !euclid || !weights || !accumulation || !features || !depth || S < 1 || S > NR_WAVE ||
      C < 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     alpha, feature, euclid, n_rays, S, C, weights, accumulation, features, depth);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_composite_bwd(const float* alpha, const float* feature, const float* euclid, const float* weights,
                                const float* g_features, const float* g_depth, const float* g_acc, const float* g_w,
                                int64_t n_rays, int S, int C, float* g_alpha, float* g_feature, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!alpha || !feature || !euclid || !weights || !g_alpha || !g_feature || S < 1 || S > NR_WAVE || C < 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     alpha, feature, euclid, weights, g_features, g_depth, g_acc, g_w, n_rays, S, C, g_alpha, g_feature);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_depth_from_weights(const float* weights, const float* euclid, int64_t n_rays, int S, float* depth,
                                     nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!weights || !euclid || !depth || S < 1 || n_rays < 0) return NR_EINVAL;
  hipLaunchKernelGGL(depth_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream), weights,
                     euclid, n_rays, S, depth);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_render_train(const float* alpha, const float* feature, const float* euclid, const float* spacing,
                               const float* target_features, const float* target_depth, int64_t n_rays, int S, int C,
                               float rgb_mult, float depth_mult, float distortion_mult, float* weights,
                               float* accumulation, float* features, float* depth, float* g_alpha, float* g_feature,
                               float* loss, const float* g_features_extra, const float* g_depth_extra, const nr_lidar_sup_t* lidar,
                               nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if ((!target_features && rgb_mult != 0.0f) || (!target_depth && depth_mult != 0.0f)) return NR_EINVAL;
  if (!alpha || !feature || !euclid || !spacing || !weights || !accumulation ||
      !features || !depth || !g_alpha || !g_feature || !loss || S < 2 || S > NR_WAVE || C < 1 || C > 32 || n_rays < 0)
    return NR_EINVAL;
  nr_lidar_sup_t lid = {};
  if (lidar != nullptr) {
    if (!lidar->is_lidar || !lidar->did_return || !lidar->range) return NR_EINVAL;
    lid = *lidar;
  }
  const dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock));
  if (S <= 32)
    hipLaunchKernelGGL(render_train_kernel<16>, grid, dim3(256), 0, nr_s(stream), alpha, feature, euclid, spacing,
                       target_features, target_depth, n_rays, S, C, rgb_mult, depth_mult, distortion_mult, weights,
                       accumulation, features, depth, g_alpha, g_feature, loss, g_features_extra, g_depth_extra, lid);
  else
    hipLaunchKernelGGL(render_train_kernel<32>, grid, dim3(256), 0, nr_s(stream), alpha, feature, euclid, spacing,
                       target_features, target_depth, n_rays, S, C, rgb_mult, depth_mult, distortion_mult, weights,
                       accumulation, features, depth, g_alpha, g_feature, loss, g_features_extra, g_depth_extra, lid);
  NR_LAUNCH_CHECK();
  return 0;
}
