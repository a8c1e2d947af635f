// Device helpers of the hash-grid kernels shared by grid.hip and actors.hip: corner arithmetic and the per-level
// gather + trilinear interpolation in the reference's order (field_components/encodings.py:406-466).
#pragma once
#include "nr_common.h"

namespace nrgrid {

template <int F>
struct VecF;
template <>
struct VecF<1> { using type = float; };
template <>
struct VecF<2> { using type = float2; };
template <>
struct VecF<4> { using type = float4; };

template <int F>
__device__ __forceinline__ void load_entry(const float* p, float (&v)[F]) {
  if constexpr (F == 1) {
    v[0] = *p;
  } else if constexpr (F == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else if constexpr (F == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) v[f] = p[f];
  }
}

// sample index handled by thread i (identity, or ray-major storage walked sample-major)
__device__ __forceinline__ int64_t sample_of_thread(int64_t i, int64_t n, int S) {
  if (S <= 0) return i;
  const int64_t B = n / S;
  const int64_t b = i % B, s = i / B;
  return b * S + s;
}

struct Corner {
  int lo[3], hi[3];
  float w[3];  // weight of the CEIL corner per axis (encodings.py:434,454-464)
};

__device__ __forceinline__ Corner make_corner(const float* x, int64_t idx, float scale) {
  Corner c;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = x[idx * 3 + a] * scale;
    const float fl = floorf(p);
    c.lo[a] = (int)fl;
    c.hi[a] = (int)ceilf(p);
    c.w[a] = p - fl;
  }
  return c;
}

// one level of one sample: 8 gathers, trilinear interpolation in the reference's order, per-level rescale
template <int F>
__device__ __forceinline__ void encode_level(const float* __restrict__ x, const float* __restrict__ std,
                                             const float* __restrict__ table, float scale, int level, int log2T, int64_t idx,
                                             float (&feat)[F]) {
  const Corner c = make_corner(x, idx, scale);
  const uint32_t mask = (1u << log2T) - 1u;
  const float* base = table + (((int64_t)level << log2T) * F);
  float acc_z[2][F];
#pragma unroll
  for (int zs = 0; zs < 2; ++zs) {  // zs = 0: ceil z, 1: floor z
    const int iz = zs == 0 ? c.hi[2] : c.lo[2];
    float acc_y[2][F];
#pragma unroll
    for (int ys = 0; ys < 2; ++ys) {
      const int iy = ys == 0 ? c.hi[1] : c.lo[1];
      float vh[F], vl[F];
      load_entry<F>(base + (int64_t)nr_hash3(c.hi[0], iy, iz, mask) * F, vh);
      load_entry<F>(base + (int64_t)nr_hash3(c.lo[0], iy, iz, mask) * F, vl);
#pragma unroll
      for (int f = 0; f < F; ++f) acc_y[ys][f] = vh[f] * c.w[0] + vl[f] * (1.0f - c.w[0]);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc_z[zs][f] = acc_y[0][f] * c.w[1] + acc_y[1][f] * (1.0f - c.w[1]);
  }
  float r = 1.0f;
  if (std != nullptr) r = 1.0f / fmaxf(scale * 2.0f * std[idx], 1.0f);  // neurad_encoding.py:314
#pragma unroll
  for (int f = 0; f < F; ++f) feat[f] = (acc_z[0][f] * c.w[2] + acc_z[1][f] * (1.0f - c.w[2])) * r;
}

}  // namespace nrgrid
