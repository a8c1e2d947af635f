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

}  // namespace

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
