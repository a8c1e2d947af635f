// tiny-cuda-nn-compatible multiresolution hash grid, 3-D and 4-D inputs: the function and the parameter
// layout of `tcnn.Encoding{HashGrid, interpolation Linear}` as the reference configures it
// (field_components/encodings.py:361-373,386-401), so that tables trained through the reference's tcnn path -- incl. the
// 4-D (xyz + actor id) grid of field_components/neurad_encoding.py:112-133,282-293 -- can be evaluated and trained here.
// The algorithm is restated from tiny-cuda-nn's published sources (encodings/grid.h: grid_scale, grid_resolution, pos_fract,
// grid_index, coherent_prime_hash; the offset table of GridEncodingTemplated's constructor); tiny-cuda-nn itself is an
// unpinned dependency that is not in the image: parity is pinned by oracle/tcnn_grid.py and the known parameter count
// of the instant-ngp default grid only.
// One thread per (sample, level); HBM/L2-bound gathers, plain float atomics in the backward (the grids this serves --
// actor boxes -- see a few percent of the samples).
#include <math.h>

#include "nr_common.h"

namespace {

constexpr int kMaxLevels = 32;

struct TcnnGeom {
  float scale[kMaxLevels];
  uint32_t res[kMaxLevels];
  uint32_t off[kMaxLevels + 1];  // entries
};

inline bool make_geom(int D, int L, int log2T, int base_res, float per_level_scale, TcnnGeom* g) {
  if ((D != 3 && D != 4) || L < 1 || L > kMaxLevels || log2T < 1 || log2T > 30 || base_res < 1 || !(per_level_scale > 0.0f))
    return false;
  const float log2_pls = log2f(per_level_scale);
  uint64_t off = 0;
  const uint32_t max_params = 0xFFFFFFFFu / 2;
  g->off[0] = 0;
  for (int l = 0; l < L; ++l) {
    const float scale = exp2f((float)l * log2_pls) * (float)base_res - 1.0f;
    const uint32_t r = (uint32_t)ceilf(scale) + 1u;
    uint64_t n = powf((float)r, (float)D) > (float)max_params ? max_params : 1;
    if (n == 1)
      for (int d = 0; d < D; ++d) n *= r;
    n = (n + 7) / 8 * 8;
    const uint64_t cap = 1ull << log2T;
    n = n < cap ? n : cap;
    g->scale[l] = scale;
    g->res[l] = r;
    off += n;
    if (off > 0xFFFFFFFFull) return false;
    g->off[l + 1] = (uint32_t)off;
  }
  return true;
}

template <int D>
__device__ __forceinline__ uint32_t grid_index(const uint32_t (&c)[D], uint32_t res, uint32_t size) {
  constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
  uint32_t stride = 1, index = 0;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (stride <= size) {
      index += c[d] * stride;
      stride *= res;
    }
  }
  if (size < stride) {
    index = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) index ^= c[d] * primes[d];
  }
  return index % size;
}

template <int D, int F, bool BWD>
__global__ void __launch_bounds__(256)
tcnn_grid_kernel(const float* __restrict__ x, const float* __restrict__ params, TcnnGeom g, int L, float* __restrict__ out,
                 const float* __restrict__ grad_out, float* __restrict__ grad_params, int64_t n) {
  const int level = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float scale = g.scale[level];
  const uint32_t res = g.res[level], off = g.off[level], size = g.off[level + 1] - off;
  uint32_t cell[D];
  float w[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float pos = fmaf(scale, x[i * D + d], 0.5f);
    const float fl = floorf(pos);
    cell[d] = (uint32_t)(int)fl;
    w[d] = pos - fl;
  }
  float acc[F], go[F], gmag = 0.0f;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    acc[f] = 0.0f;
    go[f] = BWD ? grad_out[i * (int64_t)(L * F) + level * F + f] : 0.0f;
    gmag += fabsf(go[f]);
  }
  if (BWD && gmag == 0.0f) return;  // masked rows: no gradient, whatever their position holds
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    float weight = 1.0f;
    uint32_t c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool hi = corner & (1 << d);
      weight *= hi ? w[d] : 1.0f - w[d];
      c[d] = cell[d] + (hi ? 1u : 0u);
    }
    const int64_t at = ((int64_t)off + grid_index<D>(c, res, size)) * F;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      if (BWD) {
        const float v = weight * go[f];
        if (v != 0.0f) unsafeAtomicAdd(grad_params + at + f, v);
      } else {
        acc[f] += weight * params[at + f];
      }
    }
  }
  if (!BWD) {
#pragma unroll
    for (int f = 0; f < F; ++f) out[i * (int64_t)(L * F) + level * F + f] = acc[f];
  }
}

template <bool BWD>
int launch(const float* x, const float* params, int D, int L, int F, int log2T, int base_res, float pls, float* out,
           const float* grad_out, float* grad_params, int64_t n, nr_stream_t stream) {
  if (n == 0) return 0;
  TcnnGeom g;
  if (!x || n < 0 || !make_geom(D, L, log2T, base_res, pls, &g)) return NR_EINVAL;
  if (BWD ? (!grad_out || !grad_params) : (!params || !out)) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n, 256), (unsigned)L), block(256);
#define CALL(DD, FF) \
  hipLaunchKernelGGL((tcnn_grid_kernel<DD, FF, BWD>), grid, block, 0, nr_s(stream), x, params, g, L, out, grad_out, grad_params, n)
#define BYF(DD)                   \
  switch (F) {                    \
    case 1: CALL(DD, 1); break;   \
    case 2: CALL(DD, 2); break;   \
    case 4: CALL(DD, 4); break;   \
    case 8: CALL(DD, 8); break;   \
    default: return NR_EINVAL;    \
  }
  if (D == 3) { BYF(3) } else { BYF(4) }
#undef BYF
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int64_t nr_tcnn_grid_param_count(int D, int L, int F, int log2T, int base_res, float per_level_scale) {
  TcnnGeom g;
  if (F < 1 || !make_geom(D, L, log2T, base_res, per_level_scale, &g)) return -1;
  return (int64_t)g.off[L] * F;
}

extern "C" int nr_tcnn_grid_geometry(int D, int L, int log2T, int base_res, float per_level_scale, float* scales,
                                     uint32_t* resolutions, uint32_t* offsets) {
  TcnnGeom g;
  if (!scales || !resolutions || !offsets || !make_geom(D, L, log2T, base_res, per_level_scale, &g)) return NR_EINVAL;
  for (int l = 0; l < L; ++l) {
    scales[l] = g.scale[l];
    resolutions[l] = g.res[l];
  }
  for (int l = 0; l <= L; ++l) offsets[l] = g.off[l];
  return 0;
}

extern "C" int nr_tcnn_grid_fwd(const float* x, const float* params, int D, int L, int F, int log2T, int base_res,
                                float per_level_scale, float* out, int64_t n, nr_stream_t stream) {
  return launch<false>(x, params, D, L, F, log2T, base_res, per_level_scale, out, nullptr, nullptr, n, stream);
}

extern "C" int nr_tcnn_grid_bwd(const float* x, int D, int L, int F, int log2T, int base_res, float per_level_scale,
                                const float* grad_out, float* grad_params, int64_t n, nr_stream_t stream) {
  return launch<true>(x, nullptr, D, L, F, log2T, base_res, per_level_scale, nullptr, grad_out, grad_params, n, stream);
}
