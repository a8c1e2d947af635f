// Proposal-sampling kernels: power-law initial bins, density -> weights (wavefront scan, fwd+bwd)
// and inverse-CDF resampling.  One wavefront (64 lanes) owns one ray; lanes hold consecutive
// samples so the transmittance scan is a shuffle scan, and the CDF lives in LDS for the binary search.
#include "nr_common.h"
#include "weights_dev.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxS = 256;  // samples per ray supported by the per-ray kernels

// s-space position of edge j of a ray (ray_samplers.py:102-115); t = the ray's jitter for this edge or < 0 (eval)
__device__ __forceinline__ float power_bin(int S, int j, const float* __restrict__ t_rand_ray) {
  const int E = S + 1;
  float bin = nr_linspace(0.0f, 1.0f, E, j);
  if (t_rand_ray != nullptr) {
    const float first = nr_linspace(0.0f, 1.0f, E, 0), last = nr_linspace(0.0f, 1.0f, E, S);
    const float lower = j == 0 ? first : (bin + nr_linspace(0.0f, 1.0f, E, j - 1)) / 2.0f;
    const float upper = j == S ? last : (nr_linspace(0.0f, 1.0f, E, j + 1) + bin) / 2.0f;
    bin = lower + (upper - lower) * t_rand_ray[j];
  }
  return bin;
}

// CONTRACT: also emit the contracted Gaussians of the S samples (nr_contract_gaussians' values) so the
// first hash-grid launch can follow directly.
template <bool CONTRACT>
__global__ void __launch_bounds__(256)
power_bins_kernel(const float* __restrict__ nears, const float* __restrict__ fars, const float* __restrict__ t_rand,
                  int64_t n_rays, int S, float lam, float scaling, float* __restrict__ spacing,
                  float* __restrict__ euclid, const float* __restrict__ origins, const float* __restrict__ directions,
                  const float* __restrict__ pixel_area, float scale, int sample_major, float* __restrict__ x01,
                  float* __restrict__ std01) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int E = S + 1;
  if (i >= n_rays * E) return;
  const int64_t b = i / E;
  const int j = (int)(i - b * E);
  const float* tr = t_rand != nullptr ? t_rand + b * E : nullptr;
  const float bin = power_bin(S, j, tr);
  const float s_near = nr_power_fn(nears[b] * scaling, lam), s_far = nr_power_fn(fars[b] * scaling, lam);
  const float e0 = nr_spacing_to_euclid(bin, s_near, s_far, lam, scaling);
  spacing[i] = bin;
  euclid[i] = e0;
  if constexpr (CONTRACT) {
    if (j < S) {
      const float e1 = nr_spacing_to_euclid(power_bin(S, j + 1, tr), s_near, s_far, lam, scaling);
      float x[3], sd;
      nr_contract_sample(origins + b * 3, directions + b * 3, pixel_area[b], e0, e1, scale, x, sd);
      const int64_t row = nr_row_of(b, j, S, sample_major);
#pragma unroll
      for (int a = 0; a < 3; ++a) x01[row * 3 + a] = x[a];
      std01[row] = sd;
    }
  }
}

// ---- RaySamples.get_weights (cameras/rays.py:188-210) -----------------------------------------
// ITEMS consecutive samples per lane; S <= 64*ITEMS.
// weights of one ray into w[k] (sample lane * ITEMS + k); returns nothing else
template <int ITEMS>
__device__ __forceinline__ void weights_fwd_ray(const float* __restrict__ density, const float* __restrict__ e, int S,
                                                float (&w)[ITEMS]) {
  const int lane = nr_lane();
  float dd[ITEMS], local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    dd[k] = s < S ? (e[s + 1] - e[s]) * density[s] : 0.0f;
    local += dd[k];
  }
  float excl = nr_wave_excl_sum(local);  // sum of dd over earlier lanes
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    w[k] = nr_nan_to_num((1.0f - expf(-dd[k])) * expf(-excl));
    excl += dd[k];
  }
}

template <int ITEMS>
__global__ void __launch_bounds__(256)
weights_fwd_kernel(const float* __restrict__ density, const float* __restrict__ euclid, int64_t n_rays, int S,
                   float* __restrict__ weights) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  float w[ITEMS];
  weights_fwd_ray<ITEMS>(density + ray * S, euclid + ray * (S + 1), S, w);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = nr_lane() * ITEMS + k;
    if (s < S) weights[ray * S + s] = w[k];
  }
}

template <int ITEMS>
__global__ void __launch_bounds__(256)
weights_bwd_kernel(const float* __restrict__ density, const float* __restrict__ euclid, const float* __restrict__ gw,
                   int64_t n_rays, int S, float* __restrict__ gdensity) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const float* g = gw + ray * S;
  float dens[ITEMS], delta[ITEMS];
  nr_weights_bwd_load<ITEMS>(density + ray * S, euclid + ray * (S + 1), S, dens, delta);
  nr_weights_bwd_ray<ITEMS>(dens, delta, [&](int s) { return g[s]; }, S, gdensity + ray * S);
}

// ---- PDFSampler (ray_samplers.py:305-376) -----------------------------------------------------
// One ray: win[k] = weight of sample lane * ITEMS + k.  cdf/bins: the wave's LDS rows (64*ITEMS+1 floats;
// bins is left holding the NEW euclidean edges when keep_edges is set).  Writes the S_out+1 new edges.
template <int ITEMS>
__device__ __forceinline__ void pdf_resample_ray(const float (&win)[ITEMS], const float* __restrict__ spacing_in_ray,
                                                 const float* __restrict__ jitter_ray, float near, float far, int S,
                                                 int S_out, float lam, float scaling, float sky_distance, float* cdf,
                                                 float* bins, float* __restrict__ spacing_out_ray,
                                                 float* __restrict__ euclid_out_ray, bool keep_edges) {
  const int lane = nr_lane();
  float w[ITEMS], local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    w[k] = s < S ? win[k] + 0.01f : 0.0f;  // histogram_padding (:309)
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
  for (int j = lane; j <= S; j += NR_WAVE) bins[j] = spacing_in_ray[j];
  // the wave's own LDS writes are read back by other lanes of the same wave: LDS ops of one
  // wave execute in order, the fence only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

  const int num_bins = S_out + 1;
  const float s_near = nr_power_fn(near * scaling, lam), s_far = nr_power_fn(far * scaling, lam);
  constexpr int kOutPerLane = 64 * ITEMS / NR_WAVE + 1;
  float eu_new[kOutPerLane];
#pragma unroll
  for (int t = 0; t < kOutPerLane; ++t) {
    const int j = lane + t * NR_WAVE;
    eu_new[t] = 0.0f;
    if (j < num_bins) {
      float u = nr_linspace(0.0f, (float)(1.0 - (1.0 / (double)num_bins)), num_bins, j);  // :323 / :332
      u = jitter_ray != nullptr ? u + jitter_ray[0] / (float)num_bins : u + (float)(1.0 / (double)(2 * num_bins));
      // searchsorted(cdf, u, side="right"): number of entries <= u
      int lo = 0, hi = S + 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
      }
      const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
      const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
      float tt = (u - c0) / (c1 - c0);
      tt = isnan(tt) ? 0.0f : nr_nan_to_num(tt);  // nan_to_num(., 0)
      tt = fminf(fmaxf(tt, 0.0f), 1.0f);
      const float nb = b0 + tt * (b1 - b0);
      float eu = nr_spacing_to_euclid(nb, s_near, s_far, lam, scaling);
      float sp = nb;
      if (sky_distance > 0.0f && j == S_out) {  // "sky field" (models/neuradar.py:578-582)
        eu = eu + (sky_distance - eu);
        sp = 1.0f - 1e-7f;
      }
      spacing_out_ray[j] = sp;
      euclid_out_ray[j] = eu;
      eu_new[t] = eu;
    }
  }
  if (keep_edges) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();  // every lane is done searching the old bins
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
    for (int t = 0; t < kOutPerLane; ++t) {
      const int j = lane + t * NR_WAVE;
      if (j < num_bins) bins[j] = eu_new[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  }
}

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
  float w[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = nr_lane() * ITEMS + k;
    w[k] = s < S ? weights[ray * S + s] : 0.0f;
  }
  pdf_resample_ray<ITEMS>(w, spacing_in + ray * (S + 1), jitter ? jitter + ray : nullptr, nears[ray], fars[ray], S, S_out, lam,
                          scaling, sky_distance, s_cdf[wave], s_bins[wave], spacing_out + ray * (S_out + 1),
                          euclid_out + ray * (S_out + 1), false);
}

// ---- one proposal round after its density launch, in one launch: get_weights -> expected depth ->
// PDF resampling -> contracted Gaussians of the NEW samples (what the next hash-grid launch reads).
// Same arithmetic as nr_weights_from_density_fwd + nr_depth_from_weights + nr_pdf_resample +
// nr_contract_gaussians, which remain the reference for it.
template <int ITEMS>
__global__ void __launch_bounds__(256)
proposal_round_kernel(const float* __restrict__ density, const float* __restrict__ euclid, const float* __restrict__ spacing_in,
                      const float* __restrict__ jitter, const float* __restrict__ nears, const float* __restrict__ fars,
                      const float* __restrict__ origins, const float* __restrict__ directions,
                      const float* __restrict__ pixel_area, int64_t n_rays, int S, int S_out, float lam, float scaling,
                      float sky_distance, float scale, int sample_major, float* __restrict__ weights,
                      float* __restrict__ depth, float* __restrict__ spacing_out, float* __restrict__ euclid_out,
                      float* __restrict__ x01, float* __restrict__ std01) {
  __shared__ float s_cdf[kWavesPerBlock][64 * ITEMS + 1];
  __shared__ float s_bins[kWavesPerBlock][64 * ITEMS + 1];
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const float* e = euclid + ray * (S + 1);
  float w[ITEMS];
  weights_fwd_ray<ITEMS>(density + ray * S, e, S, w);
  float d = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    if (s < S) {
      weights[ray * S + s] = w[k];
      d += w[k] * ((e[s] + e[s + 1]) / 2.0f);  // render_depth_simple (models/neurad.py:721-728)
    } else {
      w[k] = 0.0f;
    }
  }
  d = nr_wave_sum(d);
  if (lane == 0) depth[ray] = d;
  float* bins = s_bins[wave];
  pdf_resample_ray<ITEMS>(w, spacing_in + ray * (S + 1), jitter ? jitter + ray : nullptr, nears[ray], fars[ray], S, S_out, lam,
                          scaling, sky_distance, s_cdf[wave], bins, spacing_out + ray * (S_out + 1),
                          euclid_out + ray * (S_out + 1), true);
  const float area = pixel_area[ray];
  for (int j = lane; j < S_out; j += NR_WAVE) {
    float x[3], sd;
    nr_contract_sample(origins + ray * 3, directions + ray * 3, area, bins[j], bins[j + 1], scale, x, sd);
    const int64_t row = nr_row_of(ray, j, S_out, sample_major);
#pragma unroll
    for (int a = 0; a < 3; ++a) x01[row * 3 + a] = x[a];
    std01[row] = sd;
  }
}

}  // namespace

extern "C" int nr_power_bins(const float* nears, const float* fars, const float* t_rand, int64_t n_rays, int S,
                             float lam, float scaling, float* spacing, float* euclid, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!nears || !fars || !spacing || !euclid || S < 1 || n_rays < 0 || lam == 0.0f || lam == 1.0f) return NR_EINVAL;
  const int64_t n = n_rays * (S + 1);
  hipLaunchKernelGGL(power_bins_kernel<false>, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), nears, fars,
                     t_rand, n_rays, S, lam, scaling, spacing, euclid, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, nullptr);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_power_bins_contract(const float* nears, const float* fars, const float* t_rand, const float* origins,
                                      const float* directions, const float* pixel_area, int64_t n_rays, int S, float lam,
                                      float scaling, float scale, int sample_major_rows, float* spacing, float* euclid,
                                      float* x01, float* std01, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!nears || !fars || !origins || !directions || !pixel_area || !spacing || !euclid || !x01 || !std01 || S < 1 ||
      n_rays < 0 || lam == 0.0f || lam == 1.0f || !(scale > 0))
    return NR_EINVAL;
  const int64_t n = n_rays * (S + 1);
  hipLaunchKernelGGL(power_bins_kernel<true>, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), nears, fars,
                     t_rand, n_rays, S, lam, scaling, spacing, euclid, origins, directions, pixel_area, scale,
                     sample_major_rows, x01, std01);
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

extern "C" int nr_proposal_round(const float* density, const float* euclid, const float* spacing_in, const float* jitter,
                                 const float* nears, const float* fars, const float* origins, const float* directions,
                                 const float* pixel_area, int64_t n_rays, int S, int S_out, float lam, float scaling,
                                 float sky_distance, float scale, int sample_major_rows, float* weights, float* depth,
                                 float* spacing_out, float* euclid_out, float* x01, float* std01, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!density || !euclid || !spacing_in || !nears || !fars || !origins || !directions || !pixel_area || !weights || !depth ||
      !spacing_out || !euclid_out || !x01 || !std01 || S < 1 || S > kMaxS || S_out < 1 || S_out > S || n_rays < 0 ||
      lam == 0.0f || lam == 1.0f || !(scale > 0))
    return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), block(256);
#define CALL(I)                                                                                                              \
  hipLaunchKernelGGL(proposal_round_kernel<I>, grid, block, 0, nr_s(stream), density, euclid, spacing_in, jitter, nears, fars, \
                     origins, directions, pixel_area, n_rays, S, S_out, lam, scaling, sky_distance, scale, sample_major_rows,  \
                     weights, depth, spacing_out, euclid_out, x01, std01)
  NR_DISPATCH_ITEMS(S, CALL);
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
