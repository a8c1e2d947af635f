// Proposal-sampling kernels: power-law initial bins, density -> weights (wavefront scan, fwd+bwd)
// and inverse-CDF resampling.  One wavefront (64 lanes) owns one ray; lanes hold consecutive
// samples so the transmittance scan is a shuffle scan, and the CDF lives in LDS for the binary search.
#include "nr_common.h"
#include "weights_dev.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxS = 256;  // samples per ray supported by the per-ray kernels

__global__ void __launch_bounds__(256)
power_bins_kernel(const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ t_rand,
                  int64_t n_rays, int S, float lam, float scaling, float* __restrict__ spacing,
                  float* __restrict__ euclid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int E = S + 1;
  if (i >= n_rays * E) return;
  const int64_t b = i / E;
  const int j = (int)(i - b * E);
  // ray_samplers.py:102-115
  float bin = nr_linspace(0.0f, 1.0f, E, j);
  if (t_rand != nullptr) {
    const float first = nr_linspace(0.0f, 1.0f, E, 0), last = nr_linspace(0.0f, 1.0f, E, S);
    const float lower = j == 0 ? first : (bin + nr_linspace(0.0f, 1.0f, E, j - 1)) / 2.0f;
    const float upper = j == S ? last : (nr_linspace(0.0f, 1.0f, E, j + 1) + bin) / 2.0f;
    bin = lower + (upper - lower) * t_rand[i];
  }
  const float s_near = nr_power_fn(nears[b] * scaling, lam), s_far = nr_power_fn(fars[b] * scaling, lam);
  spacing[i] = bin;
  euclid[i] = nr_spacing_to_euclid(bin, s_near, s_far, lam, scaling);
}

// ---- RaySamples.get_weights (cameras/rays.py:188-210) -----------------------------------------
// ITEMS consecutive samples per lane; S <= 64*ITEMS.
template <int ITEMS>
__global__ void __launch_bounds__(256)
weights_fwd_kernel(const float* __restrict__ density, const float* __restrict__ euclid, int64_t n_rays, int S,
                   float* __restrict__ weights) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const float* e = euclid + ray * (S + 1);
  float dd[ITEMS], local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    dd[k] = s < S ? (e[s + 1] - e[s]) * density[ray * S + s] : 0.0f;
    local += dd[k];
  }
  float excl = nr_wave_excl_sum(local);  // sum of dd over earlier lanes
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    if (s < S) weights[ray * S + s] = nr_nan_to_num((1.0f - expf(-dd[k])) * expf(-excl));
    excl += dd[k];
  }
}

template <int ITEMS>
__global__ void __launch_bounds__(256)
weights_bwd_kernel(const float* __restrict__ density, const float* __restrict__ euclid, const float* __restrict__ gw,
                   int64_t n_rays, int S, float* __restrict__ gdensity) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float* g = gw + ray * S;
  nr_weights_bwd_ray<ITEMS>(density + ray * S, euclid + ray * (S + 1), [&](int s) { return g[s]; }, S, gdensity + ray * S);
}

// ---- PDFSampler (ray_samplers.py:305-376) -----------------------------------------------------
template <int ITEMS>
__global__ void __launch_bounds__(256)
pdf_resample_kernel(const float* __restrict__ weights, const float* __restrict__ spacing_in,
                    const float* __restrict__ jitter, const float* __restrict__ nears, const float* __restrict__ fars,
                    int64_t n_rays, int S, int S_out, float lam, float scaling, float sky_distance,
                    float* __restrict__ spacing_out, float* __restrict__ euclid_out) {
  __shared__ float s_cdf[kWavesPerBlock][64 * ITEMS + 1];
  __shared__ float s_bins[kWavesPerBlock][64 * ITEMS + 1];
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (ray >= n_rays) return;  // whole wave exits together; no block barrier is used below
  const int lane = nr_lane();
  float* cdf = s_cdf[wave];
  float* bins = s_bins[wave];

  float w[ITEMS], local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    w[k] = s < S ? weights[ray * S + s] + 0.01f : 0.0f;  // histogram_padding (:309)
    local += w[k];
  }
  float w_sum = nr_wave_sum(local);
  const float padding = fmaxf(1e-5f - w_sum, 0.0f);  // :312-315
  w_sum += padding;
  local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    w[k] = s < S ? (w[k] + padding / (float)S) / w_sum : 0.0f;  // pdf (:317)
    local += w[k];
  }
  float run = nr_wave_excl_sum(local);
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    run += w[k];
    if (s < S) cdf[s + 1] = fminf(1.0f, run);  // :318-319
  }
  for (int j = lane; j <= S; j += NR_WAVE) bins[j] = spacing_in[ray * (S + 1) + j];
  // the wave's own LDS writes are read back by other lanes of the same wave: LDS ops of one
  // wave execute in order, the fence only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int num_bins = S_out + 1;
  const float s_near = nr_power_fn(nears[ray] * scaling, lam), s_far = nr_power_fn(fars[ray] * scaling, lam);
  for (int j = lane; j < num_bins; j += NR_WAVE) {
    float u = nr_linspace(0.0f, (float)(1.0 - (1.0 / (double)num_bins)), num_bins, j);  // :323 / :332
    u = jitter != nullptr ? u + jitter[ray] / (float)num_bins : u + (float)(1.0 / (double)(2 * num_bins));
    // searchsorted(cdf, u, side="right"): number of entries <= u
    int lo = 0, hi = S + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
    const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
    float t = (u - c0) / (c1 - c0);
    t = isnan(t) ? 0.0f : nr_nan_to_num(t);  // nan_to_num(., 0)
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float nb = b0 + t * (b1 - b0);
    float eu = nr_spacing_to_euclid(nb, s_near, s_far, lam, scaling);
    float sp = nb;
    if (sky_distance > 0.0f && j == S_out) {  // "sky field" (models/neuradar.py:578-582)
      eu = eu + (sky_distance - eu);
      sp = 1.0f - 1e-7f;
    }
    spacing_out[ray * num_bins + j] = sp;
    euclid_out[ray * num_bins + j] = eu;
  }
}

}  // namespace

extern "C" int nr_power_bins(const float* nears, const float* fars, const float* t_rand, int64_t n_rays, int S,
                             float lam, float scaling, float* spacing, float* euclid, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!nears || !fars || !spacing || !euclid || S < 1 || n_rays < 0 || lam == 0.0f || lam == 1.0f) return NR_EINVAL;
  const int64_t n = n_rays * (S + 1);
  hipLaunchKernelGGL(power_bins_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), nears, fars,
                     t_rand, n_rays, S, lam, scaling, spacing, euclid);
  NR_LAUNCH_CHECK();
  return 0;
}

#define NR_DISPATCH_ITEMS(S, CALL)                   \
  do {                                               \
    if ((S) <= 64) { CALL(1); }                      \
    else if ((S) <= 128) { CALL(2); }                \
    else { CALL(4); }                                \
  } while (0)

extern "C" int nr_weights_from_density_fwd(const float* density, const float* euclid, int64_t n_rays, int S,
                                           float* weights, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!density || !euclid || !weights || S < 1 || S > kMaxS || n_rays < 0) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), block(256);
#define CALL(I) hipLaunchKernelGGL(weights_fwd_kernel<I>, grid, block, 0, nr_s(stream), density, euclid, n_rays, S, weights)
  NR_DISPATCH_ITEMS(S, CALL);
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_weights_from_density_bwd(const float* density, const float* euclid, const float* gw, int64_t n_rays,
                                           int S, float* gdensity, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!density || !euclid || !gw || !gdensity || S < 1 || S > kMaxS || n_rays < 0) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), block(256);
#define CALL(I) hipLaunchKernelGGL(weights_bwd_kernel<I>, grid, block, 0, nr_s(stream), density, euclid, gw, n_rays, S, gdensity)
  NR_DISPATCH_ITEMS(S, CALL);
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_pdf_resample(const float* weights, const float* spacing_in, const float* jitter, const float* nears,
                               const float* fars, int64_t n_rays, int S, int S_out, float lam, float scaling,
                               float sky_distance, float* spacing_out, float* euclid_out, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!weights || !spacing_in || !nears || !fars || !spacing_out || !euclid_out || S < 1 || S > kMaxS || S_out < 1 ||
      n_rays < 0 || lam == 0.0f || lam == 1.0f)
    return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), block(256);
#define CALL(I) hipLaunchKernelGGL(pdf_resample_kernel<I>, grid, block, 0, nr_s(stream), weights, spacing_in, jitter, nears, fars, n_rays, S, S_out, lam, scaling, sky_distance, spacing_out, euclid_out)
  NR_DISPATCH_ITEMS(S, CALL);
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
