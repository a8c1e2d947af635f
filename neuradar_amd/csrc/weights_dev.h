// Backward of RaySamples.get_weights (cameras/rays.py:188-210) for one ray per wavefront, shared by
// nr_weights_from_density_bwd and the fused inter-level-loss launch.  ITEMS consecutive samples per lane.
#pragma once
#include "nr_common.h"

// inputs of one lane: sample lane * ITEMS + k has density dens[k] and width delta[k] (0 past S)
template <int ITEMS>
__device__ __forceinline__ void nr_weights_bwd_load(const float* __restrict__ density, const float* __restrict__ e, int S,
                                                    float (&dens)[ITEMS], float (&delta)[ITEMS]) {
  const int lane = nr_lane();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    delta[k] = s < S ? e[s + 1] - e[s] : 0.0f;
    dens[k] = s < S ? density[s] : 0.0f;
  }
}

// gw(s) -> dL/dw_s of this ray; writes gdensity[s] for s < S
template <int ITEMS, typename GwFn>
__device__ __forceinline__ void nr_weights_bwd_ray(const float (&dens)[ITEMS], const float (&delta)[ITEMS], GwFn gw, int S,
                                                   float* __restrict__ gdensity) {
  const int lane = nr_lane();
  float dd[ITEMS], local = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    dd[k] = delta[k] * dens[k];
    local += dd[k];
  }
  float excl = nr_wave_excl_sum(local);
  // w_s = a_s T_s, T_s = exp(-P_s), P_s = sum_{j<s} dd_j
  //   d dd_s = gw_s T_s exp(-dd_s)  -  sum_{j>s} gw_j w_j
  float gT[ITEMS], gww[ITEMS], tail = 0.0f;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int s = lane * ITEMS + k;
    const float T = expf(-excl), ex = expf(-dd[k]);
    const float w = (1.0f - ex) * T;
    float g = s < S ? gw(s) : 0.0f;
    if (isnan(w) || isinf(w)) g = 0.0f;  // nan_to_num passes no gradient there
    gT[k] = g * T * ex;
    gww[k] = g * w;
    tail += gww[k];
    excl += dd[k];
  }
  float after = nr_wave_excl_suffix_sum(tail);  // sum of g*w over later lanes
#pragma unroll
  for (int k = ITEMS - 1; k >= 0; --k) {
    const int s = lane * ITEMS + k;
    if (s < S) gdensity[s] = (gT[k] - after) * delta[k];
    after += gww[k];
  }
}
