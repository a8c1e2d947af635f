// Hash-grid kernels: frustum -> contracted Gaussian, multiresolution hash gather (fwd) and
// scatter-add (bwd).  HBM/L2-bound integer+gather work: no MFMA here by design.
//
// Launch shape: blockIdx.y = level, so every wave gathers from ONE level's table region; with
// `sample_major` the 64 lanes of a wave are 64 neighbouring RAYS at the same sample slot, which for
// camera patches are centimetres apart -> the texture path merges most of a wave-instruction's
// addresses into a few cache lines on all but the finest levels.
#include <limits.h>

#include "nr_common.h"

namespace {

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

template <int F>
__global__ void __launch_bounds__(256)
hash_encode_fwd_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ table,
                       const float* __restrict__ scalings, int log2T, float* __restrict__ out, int64_t sn, int64_t sl,
                       int64_t n, int S) {
  const int level = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t idx = sample_of_thread(i, n, S);
  const float scale = scalings[level];
  const Corner c = make_corner(x, idx, scale);
  const uint32_t mask = (1u << log2T) - 1u;
  const float* base = table + (((int64_t)level << log2T) * F);

  // x-pairs for the four (y,z) selections, reduced x -> y -> z exactly like the reference
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
  float* o = out + idx * sn + (int64_t)level * sl;
#pragma unroll
  for (int f = 0; f < F; ++f) o[f] = (acc_z[0][f] * c.w[2] + acc_z[1][f] * (1.0f - c.w[2])) * r;
}

// Backward scatter-add.  Memory-side float atomics are the scarce resource here (about 20 G
// requests/s chip-wide, far less when many lanes hit one address -- and coherent camera rays do exactly
// that), so contributions are summed on chip first:
//   1. lanes l and l^32 (the same pixel column of two adjacent patch rows under the sample-major
//      mapping) that fall in the same grid cell are folded into one lane;
//   2. a segmented wave scan sums every run of consecutive lanes that share a cell (a cell fixes all
//      eight corner slots), and only the run's last lane issues the 8*F atomics.
// Incoherent rays (lidar points, radar grids) simply form runs of length one: the cost is a fixed
// ~(9 + 8F) shuffles per scan step and no loss against plain atomics.
template <int F>
__global__ void __launch_bounds__(256)
hash_encode_bwd_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings,
                       int log2T, const float* __restrict__ gout, int64_t sn, int64_t sl,
                       float* __restrict__ gtable, int64_t n, int S) {
  const int level = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;  // no early exit: every lane takes part in the wave scans
  const int lane = nr_lane();
  const float scale = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u;
  float* base = gtable + (((int64_t)level << log2T) * F);

  Corner c;
  float v[8][F];
  if (valid) {
    const int64_t idx = sample_of_thread(i, n, S);
    c = make_corner(x, idx, scale);
    float r = 1.0f;
    if (std != nullptr) r = 1.0f / fmaxf(scale * 2.0f * std[idx], 1.0f);
    const float* gi = gout + idx * sn + (int64_t)level * sl;
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gi[f] * r;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
      const float w = (hx ? c.w[0] : 1.0f - c.w[0]) * (hy ? c.w[1] : 1.0f - c.w[1]) * (hz ? c.w[2] : 1.0f - c.w[2]);
#pragma unroll
      for (int f = 0; f < F; ++f) v[corner][f] = g[f] * w;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) { c.lo[a] = INT_MIN + lane; c.hi[a] = INT_MIN + lane; c.w[a] = 0.0f; }
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) v[corner][f] = 0.0f;
  }
  // a lane's cell is (lo, hi) per axis: hi == lo exactly on grid planes, so both are compared
  auto same_cell = [&](int lx, int ly, int lz, int ux, int uy, int uz) {
    return lx == c.lo[0] && ly == c.lo[1] && lz == c.lo[2] && ux == c.hi[0] && uy == c.hi[1] && uz == c.hi[2];
  };
  // 1. fold the upper half-wave onto the lower one where the cells agree
  {
    const bool same = same_cell(__shfl_xor(c.lo[0], 32, NR_WAVE), __shfl_xor(c.lo[1], 32, NR_WAVE),
                                __shfl_xor(c.lo[2], 32, NR_WAVE), __shfl_xor(c.hi[0], 32, NR_WAVE),
                                __shfl_xor(c.hi[1], 32, NR_WAVE), __shfl_xor(c.hi[2], 32, NR_WAVE));
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float o = __shfl_xor(v[corner][f], 32, NR_WAVE);
        if (same) v[corner][f] = lane < 32 ? v[corner][f] + o : 0.0f;
      }
  }
  // 2. segmented inclusive scan over runs of equal cells
  const bool head = lane == 0 || !same_cell(__shfl_up(c.lo[0], 1, NR_WAVE), __shfl_up(c.lo[1], 1, NR_WAVE),
                                            __shfl_up(c.lo[2], 1, NR_WAVE), __shfl_up(c.hi[0], 1, NR_WAVE),
                                            __shfl_up(c.hi[1], 1, NR_WAVE), __shfl_up(c.hi[2], 1, NR_WAVE));
  int flag = head ? 1 : 0;
#pragma unroll
  for (int d = 1; d < NR_WAVE; d <<= 1) {
    const int fprev = __shfl_up(flag, d, NR_WAVE);
    const bool take = lane >= d && !flag;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float t = __shfl_up(v[corner][f], d, NR_WAVE);
        if (take) v[corner][f] += t;
      }
    if (take) flag = fprev;
  }
  const int next_head = __shfl_down(head ? 1 : 0, 1, NR_WAVE);
  const bool tail = lane == NR_WAVE - 1 || next_head;
  if (tail) {
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
      bool nz = false;
#pragma unroll
      for (int f = 0; f < F; ++f) nz |= v[corner][f] != 0.0f;
      if (!nz) continue;  // zero-weight corners (exact grid planes), folded or padding lanes
      const uint32_t slot = nr_hash3(hx ? c.hi[0] : c.lo[0], hy ? c.hi[1] : c.lo[1], hz ? c.hi[2] : c.lo[2], mask);
      float* dst = base + (int64_t)slot * F;
#pragma unroll
      for (int f = 0; f < F; ++f) unsafeAtomicAdd(dst + f, v[corner][f]);
    }
  }
}

__global__ void __launch_bounds__(256)
contract_gaussians_kernel(const float* __restrict__ origins, const float* __restrict__ directions,
                          const float* __restrict__ pixel_area, const float* __restrict__ edges, int64_t n_rays,
                          int S, float scale, float* __restrict__ x01, float* __restrict__ std01) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * S) return;
  const int64_t b = i / S;
  const int s = (int)(i - b * S);
  const float e0 = edges[b * (S + 1) + s], e1 = edges[b * (S + 1) + s + 1];
  // Frustums.get_fast_isotropic_gaussian, one multisample (cameras/rays.py:118-123)
  const float half = (e1 - e0) / 2.0f;
  const float t = e0 + 1.0f * half;
  const float cross = pixel_area[b] * (t * t);
  float sd = powf(cross * half, 1.0f / 3.0f);
  float m[3];
  float mag = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    m[a] = (origins[b * 3 + a] + directions[b * 3 + a] * t) / scale;  // spatial_distortions.py:133
    mag = fmaxf(mag, fabsf(m[a]));
  }
  sd = sd / scale;
  if (!(mag < 1.0f)) {  // spatial_distortions.py:107-112 (L-inf norm)
    const float cm = fmaxf(mag, 1.0f);
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = (2.0f - (1.0f / cm)) * (m[a] / cm);
    const float k = powf(2.0f * cm - 1.0f, 1.0f / 3.0f) / cm;
    sd = sd * (k * k);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) x01[i * 3 + a] = (m[a] + 2.0f) / 4.0f;  // :135
  std01[i] = sd / 4.0f;                                               // :136
}

}  // namespace

extern "C" int nr_hash_encode_fwd(const float* x, const float* std, const float* table, const float* scalings,
                                  int L, int F, int log2T, float* out, int64_t sn, int64_t sl, int64_t n,
                                  int sample_major, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !table || !scalings || !out || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  if (sample_major > 0 && n % sample_major != 0) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n, 256), (unsigned)L), block(256);
  switch (F) {
    case 1: hipLaunchKernelGGL(hash_encode_fwd_kernel<1>, grid, block, 0, nr_s(stream), x, std, table, scalings, log2T, out, sn, sl, n, sample_major); break;
    case 2: hipLaunchKernelGGL(hash_encode_fwd_kernel<2>, grid, block, 0, nr_s(stream), x, std, table, scalings, log2T, out, sn, sl, n, sample_major); break;
    case 4: hipLaunchKernelGGL(hash_encode_fwd_kernel<4>, grid, block, 0, nr_s(stream), x, std, table, scalings, log2T, out, sn, sl, n, sample_major); break;
    case 8: hipLaunchKernelGGL(hash_encode_fwd_kernel<8>, grid, block, 0, nr_s(stream), x, std, table, scalings, log2T, out, sn, sl, n, sample_major); break;
    default: return NR_EINVAL;
  }
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_hash_encode_bwd(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                  const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n,
                                  int sample_major, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !gout || !scalings || !gtable || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  if (sample_major > 0 && n % sample_major != 0) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n, 256), (unsigned)L), block(256);
  switch (F) {
    case 1: hipLaunchKernelGGL(hash_encode_bwd_kernel<1>, grid, block, 0, nr_s(stream), x, std, scalings, log2T, gout, sn, sl, gtable, n, sample_major); break;
    case 2: hipLaunchKernelGGL(hash_encode_bwd_kernel<2>, grid, block, 0, nr_s(stream), x, std, scalings, log2T, gout, sn, sl, gtable, n, sample_major); break;
    case 4: hipLaunchKernelGGL(hash_encode_bwd_kernel<4>, grid, block, 0, nr_s(stream), x, std, scalings, log2T, gout, sn, sl, gtable, n, sample_major); break;
    case 8: hipLaunchKernelGGL(hash_encode_bwd_kernel<8>, grid, block, 0, nr_s(stream), x, std, scalings, log2T, gout, sn, sl, gtable, n, sample_major); break;
    default: return NR_EINVAL;
  }
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_contract_gaussians(const float* origins, const float* directions, const float* pixel_area,
                                     const float* edges, int64_t n_rays, int S, float scale, float* x01, float* std01,
                                     nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!origins || !directions || !pixel_area || !edges || !x01 || !std01 || S < 1 || n_rays < 0 || !(scale > 0)) return NR_EINVAL;
  const int64_t n = n_rays * S;
  hipLaunchKernelGGL(contract_gaussians_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), origins,
                     directions, pixel_area, edges, n_rays, S, scale, x01, std01);
  NR_LAUNCH_CHECK();
  return 0;
}
