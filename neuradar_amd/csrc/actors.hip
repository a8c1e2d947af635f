// Dynamic actors on the device, without host synchronisation.
//
// The reference finds the samples inside actor boxes with two `nonzero`s (host syncs) and loops over the actors in
// Python (field_components/neurad_encoding.py:231-275,295-307).  Here every step is a fixed-shape launch:
//   nr_actor_candidates   per ray: the actors whose bounding sphere the ray's sample line passes (:237-246), at most K
//   (torch, fixed shape)  world->box transforms of those (ray, actor) pairs from the learnable trajectories
//                         (model_components/dynamic_actors.py:183-197, utils/poses.py:90-149): tiny tensors, autograd
//   nr_actor_assign       per sample: sphere test (:254-258), exact box test (:263-267) -> the actor it belongs to,
//                         box-frame position through the actor contraction, box-frame view direction, per-ray flip
//   nr_actor_encode_fwd   the actor's 3-D hash grid written over the static features of those samples (:186-187)
//   nr_actor_encode_bwd   gradients: actor tables (+=), zero for the overwritten static features, and -- for the
//                         trajectory optimisation -- d loss / d (world->box transform) per (ray, candidate)
// Samples outside every box cost one 4-byte read in the last two kernels.
#include "grid_dev.h"

using namespace nrgrid;

namespace {

constexpr float kEps = 1.0e-7f;  // neurad_encoding.py:33

// Keyframe interval of every ray's time: right = searchsorted(timestamps, t) (first stamp >= t), left = max(right - 1, 0),
// right clamped to the last stamp, frac = clamp((t - ts[left]) / (ts[right] - ts[left] + 1e-6), 0, 1) -- the arithmetic of
// utils/poses.py:108-121 -- in one launch (the same in torch ops is a dozen tiny launches at the top of every step).
__global__ void __launch_bounds__(256)
actor_keyframes_kernel(const float* __restrict__ times, int64_t n, const float* __restrict__ stamps, int n_stamps,
                       int64_t* __restrict__ left, int64_t* __restrict__ right, float* __restrict__ frac) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const float t = times[b];
  int lo = 0, hi = n_stamps;
  while (lo < hi) {  // first index with stamps[i] >= t
    const int m = (lo + hi) >> 1;
    if (stamps[m] < t) lo = m + 1; else hi = m;
  }
  const int l = lo - 1 < 0 ? 0 : lo - 1, r = lo > n_stamps - 1 ? n_stamps - 1 : lo;
  const float f = (t - stamps[l]) / (stamps[r] - stamps[l] + 1.0e-6f);
  left[b] = l;
  right[b] = r;
  frac[b] = fminf(fmaxf(f, 0.0f), 1.0f);
}

__global__ void __launch_bounds__(256)
actor_candidates_kernel(const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ euclid,
                        int S, const int64_t* __restrict__ left, const int64_t* __restrict__ right, const float* __restrict__ frac,
                        const float* __restrict__ positions, const uint8_t* __restrict__ present, const float* __restrict__ bounds,
                        int n_actors, int64_t n_rays, int K, int* __restrict__ cand, int* __restrict__ overflow) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_rays) return;
  const float* e = euclid + b * (S + 1);
  const float t0 = e[0] + (e[1] - e[0]) / 2.0f, t1 = e[S - 1] + (e[S] - e[S - 1]) / 2.0f;
  float o[3], line[3], p0[3], len = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o[a] = origins[b * 3 + a];
    const float d = directions[b * 3 + a];
    p0[a] = o[a] + d * t0;
    line[a] = (o[a] + d * t1) - p0[a];  // first -> last sample (:239-240)
    len += line[a] * line[a];
  }
  len = sqrtf(len) + kEps;
#pragma unroll
  for (int a = 0; a < 3; ++a) line[a] /= len;
  const int64_t l = left[b], r = right[b];
  const float f = frac[b];
  int n = 0;
  for (int a = 0; a < n_actors; ++a) {
    const bool valid = present[l * n_actors + a] || present[r * n_actors + a];
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float pl = positions[(l * n_actors + a) * 3 + k], pr = positions[(r * n_actors + a) * 3 + k];
      c[k] = (pl + (pr - pl) * f) - p0[k];
    }
    const float cx = c[1] * line[2] - c[2] * line[1], cy = c[2] * line[0] - c[0] * line[2], cz = c[0] * line[1] - c[1] * line[0];
    const float dist = sqrtf(cx * cx + cy * cy + cz * cz);
    const float radius = sqrtf(bounds[a * 3] * bounds[a * 3] + bounds[a * 3 + 1] * bounds[a * 3 + 1] + bounds[a * 3 + 2] * bounds[a * 3 + 2]);
    if (valid && dist < radius) {
      if (n < K) cand[b * K + n] = a;
      ++n;
    }
  }
  for (int k = n < K ? n : K; k < K; ++k) cand[b * K + k] = -1;
  if (n > K) atomicMax(overflow, n);
}

// ScaledSceneContraction(order = inf, scale) on (position, std): spatial_distortions.py:103-113,126-136
__device__ __forceinline__ void contract_actor(const float (&p)[3], float sd, float scale, float (&x01)[3], float& std01) {
  float m[3], mag = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    m[a] = p[a] / scale;
    mag = fmaxf(mag, fabsf(m[a]));
  }
  sd = sd / scale;
  if (!(mag < 1.0f)) {
    const float cm = fmaxf(mag, 1.0f);
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = (2.0f - (1.0f / cm)) * (m[a] / cm);
    const float k = powf(2.0f * cm - 1.0f, 1.0f / 3.0f) / cm;
    sd = sd * (k * k);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) x01[a] = (m[a] + 2.0f) / 4.0f;
  std01 = sd / 4.0f;
}

struct SampleGeom {
  int64_t ray;
  int64_t out;   // ray-major sample index b * S + s
  float pos[3];  // world position of the sample's centre
  float sd;      // isotropic std (cameras/rays.py:109-124)
};

__device__ __forceinline__ SampleGeom sample_geom(int64_t row, int64_t n, int S, int sm, const float* __restrict__ origins,
                                                  const float* __restrict__ directions, const float* __restrict__ pixel_area,
                                                  const float* __restrict__ euclid) {
  const NrRowMap rm = nr_row_map(row, n, S, sm);
  SampleGeom g;
  g.ray = rm.ray;
  g.out = rm.out;
  const int s = (int)(rm.out - rm.ray * S);
  const float e0 = euclid[rm.ray * (S + 1) + s], e1 = euclid[rm.ray * (S + 1) + s + 1];
  const float half = (e1 - e0) / 2.0f;
  const float t = e0 + 1.0f * half;
#pragma unroll
  for (int a = 0; a < 3; ++a) g.pos[a] = origins[rm.ray * 3 + a] + directions[rm.ray * 3 + a] * t;
  g.sd = powf(pixel_area[rm.ray] * (t * t) * half, 1.0f / 3.0f);
  return g;
}

__global__ void __launch_bounds__(256)
actor_assign_kernel(const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ pixel_area,
                    const float* __restrict__ euclid, int64_t n_rays, int S, int sm, const int* __restrict__ cand, int K,
                    const float* __restrict__ w2b, const float* __restrict__ centres, const float* __restrict__ bounds,
                    float actor_scale, const float* __restrict__ flip, int* __restrict__ slot_of_row,
                    float* __restrict__ x01a, float* __restrict__ std01a, float* __restrict__ dirs_sample) {
  const int64_t n = n_rays * S;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const SampleGeom g = sample_geom(row, n, S, sm, origins, directions, pixel_area, euclid);
  int slot = -1;
  for (int k = 0; k < K; ++k) {
    const int a = cand[g.ray * K + k];
    if (a < 0) break;
    const float* c = centres + (g.ray * K + k) * 3;
    const float dx = g.pos[0] - c[0], dy = g.pos[1] - c[1], dz = g.pos[2] - c[2];
    const float* bd = bounds + a * 3;
    const float radius = sqrtf(bd[0] * bd[0] + bd[1] * bd[1] + bd[2] * bd[2]);
    if (!(sqrtf(dx * dx + dy * dy + dz * dz) < radius)) continue;  // :254-258
    const float* w = w2b + (g.ray * K + k) * 12;
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float pb = (g.pos[0] * w[i * 4] + g.pos[1] * w[i * 4 + 1] + g.pos[2] * w[i * 4 + 2]) + w[i * 4 + 3];
      inside = inside && fabsf(pb) < bd[i];
    }
    if (inside) slot = k;  // candidates ascend by actor: the last match wins, like the reference's index_put order
  }
  slot_of_row[row] = slot;
  float dvec[3] = {directions[g.ray * 3], directions[g.ray * 3 + 1], directions[g.ray * 3 + 2]};
  if (slot >= 0) {
    const float* w = w2b + (g.ray * K + slot) * 12;
    const float sign = flip != nullptr ? flip[g.ray] : 1.0f;
    float pb[3], db[3], dn = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      pb[i] = (g.pos[0] * w[i * 4] + g.pos[1] * w[i * 4 + 1] + g.pos[2] * w[i * 4 + 2]) + w[i * 4 + 3];
      db[i] = dvec[0] * w[i * 4] + dvec[1] * w[i * 4 + 1] + dvec[2] * w[i * 4 + 2];
      dn += db[i] * db[i];
    }
    dn = sqrtf(dn) + kEps;
    pb[0] *= sign;  // per-ray random flip around the x axis (:218-225)
    float x01[3], s01;
    contract_actor(pb, g.sd, actor_scale, x01, s01);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      x01a[row * 3 + i] = x01[i];
      dvec[i] = db[i] / dn;
    }
    dvec[0] *= sign;
    std01a[row] = s01;
  }
  if (dirs_sample != nullptr) {
#pragma unroll
    for (int i = 0; i < 3; ++i) dirs_sample[g.out * 3 + i] = dvec[i];
  }
}

template <int F>
__global__ void __launch_bounds__(256)
actor_encode_fwd_kernel(const float* __restrict__ x01a, const float* __restrict__ std01a, const int* __restrict__ slot_of_row,
                        const int* __restrict__ cand, int K, int S, int sm, int64_t n, const float* __restrict__ tables,
                        const int* __restrict__ table_of_actor, const float* __restrict__ scalings, int L, int log2T,
                        float* __restrict__ feats, int64_t sn, int64_t sl, int static_levels) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int slot = slot_of_row[row];
  if (slot < 0) return;
  const NrRowMap rm = nr_row_map(row, n, S, sm);
  const int a = cand[rm.ray * K + slot];
  const int tid = table_of_actor ? table_of_actor[a] : a;  // actor_to_id: the actor's hash grid (neurad_encoding.py:183)
  const float* table = tables + (((int64_t)tid * L) << log2T) * F;
  for (int level = 0; level < static_levels; ++level) {
    float feat[F];
    if (level < L) {
      encode_level<F>(x01a, std01a, table, scalings[level], level, log2T, row, feat);
    } else {  // actor features are zero-padded to the static width (:186)
#pragma unroll
      for (int f = 0; f < F; ++f) feat[f] = 0.0f;
    }
    float* o = feats + row * sn + (int64_t)level * sl;
#pragma unroll
    for (int f = 0; f < F; ++f) o[f] = feat[f];
  }
}

// backward of contract_actor's position path (the std path carries no gradient: the grid's rescale weight is treated as
// a constant of the sample, as in ops._HashEncode): g wrt x01 -> g wrt the box-frame position
__device__ __forceinline__ void contract_actor_bwd(const float (&p)[3], float scale, const float (&g01)[3], float (&gp)[3]) {
  float x[3], mag = 0.0f;
  int jmax = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    x[a] = p[a] / scale;
    if (fabsf(x[a]) > mag) { mag = fabsf(x[a]); jmax = a; }
  }
  float gx[3];
  if (mag < 1.0f) {
#pragma unroll
    for (int a = 0; a < 3; ++a) gx[a] = g01[a] / 4.0f;
  } else {  // x' = (2 - 1/m) x / m, m = |x_jmax|
    const float m = mag, k = (2.0f - 1.0f / m) / m;
    float gm = 0.0f;  // d loss / d m
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      gx[a] = (g01[a] / 4.0f) * k;
      gm += (g01[a] / 4.0f) * x[a] * (-2.0f / (m * m) + 2.0f / (m * m * m));
    }
    gx[jmax] += gm * (x[jmax] >= 0.0f ? 1.0f : -1.0f);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) gp[a] = gx[a] / scale;
}

template <int F>
__global__ void __launch_bounds__(256)
actor_encode_bwd_kernel(const float* __restrict__ x01a, const float* __restrict__ std01a, const int* __restrict__ slot_of_row,
                        const int* __restrict__ cand, int K, int S, int sm, int64_t n, const float* __restrict__ tables,
                        const int* __restrict__ table_of_actor, const float* __restrict__ scalings, int L, int log2T,
                        float* __restrict__ g_feats, int64_t sn, int64_t sl,
                        int static_levels, float* __restrict__ g_tables, const float* __restrict__ origins,
                        const float* __restrict__ directions, const float* __restrict__ pixel_area, const float* __restrict__ euclid,
                        const float* __restrict__ w2b, float actor_scale, const float* __restrict__ flip, float* __restrict__ g_w2b) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int slot = slot_of_row[row];
  if (slot < 0) return;
  const NrRowMap rm = nr_row_map(row, n, S, sm);
  const int a = cand[rm.ray * K + slot];
  const int tid = table_of_actor ? table_of_actor[a] : a;  // actor_to_id (neurad_encoding.py:183)
  const uint32_t mask = (1u << log2T) - 1u;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int level = 0; level < L; ++level) {
    const float scale = scalings[level];
    const Corner c = make_corner(x01a, row, scale);
    const int64_t base = ((((int64_t)tid * L + level) << log2T)) * F;
    float r = 1.0f / fmaxf(scale * 2.0f * std01a[row], 1.0f);
    float g[F];
    float* gi = g_feats + row * sn + (int64_t)level * sl;
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gi[f] * r;
    float t[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
      const int64_t e = base + (int64_t)nr_hash3(hx ? c.hi[0] : c.lo[0], hy ? c.hi[1] : c.lo[1], hz ? c.hi[2] : c.lo[2], mask) * F;
      const float w = (hx ? c.w[0] : 1.0f - c.w[0]) * (hy ? c.w[1] : 1.0f - c.w[1]) * (hz ? c.w[2] : 1.0f - c.w[2]);
      float d = 0.0f;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (g[f] != 0.0f) unsafeAtomicAdd(g_tables + e + f, g[f] * w);
        if (g_w2b != nullptr) d += tables[e + f] * g[f];
      }
      t[corner] = d;
    }
    if (g_w2b != nullptr) {  // d interp / d position, as hash_encode_bwd_input_kernel
      const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
      acc[0] += scale * (((t[7] - t[6]) * wy + (t[5] - t[4]) * (1.0f - wy)) * wz + ((t[3] - t[2]) * wy + (t[1] - t[0]) * (1.0f - wy)) * (1.0f - wz));
      acc[1] += scale * (((t[7] - t[5]) * wx + (t[6] - t[4]) * (1.0f - wx)) * wz + ((t[3] - t[1]) * wx + (t[2] - t[0]) * (1.0f - wx)) * (1.0f - wz));
      acc[2] += scale * (((t[7] - t[3]) * wx + (t[6] - t[2]) * (1.0f - wx)) * wy + ((t[5] - t[1]) * wx + (t[4] - t[0]) * (1.0f - wx)) * (1.0f - wy));
    }
  }
  // the static grid was overwritten at this sample: it receives no gradient (:186-187)
  for (int level = 0; level < static_levels; ++level) {
    float* gi = g_feats + row * sn + (int64_t)level * sl;
#pragma unroll
    for (int f = 0; f < F; ++f) gi[f] = 0.0f;
  }
  if (g_w2b != nullptr) {
    const SampleGeom gm = sample_geom(row, n, S, sm, origins, directions, pixel_area, euclid);
    const float* w = w2b + (rm.ray * K + slot) * 12;
    const float sign = flip != nullptr ? flip[rm.ray] : 1.0f;
    float pb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pb[i] = (gm.pos[0] * w[i * 4] + gm.pos[1] * w[i * 4 + 1] + gm.pos[2] * w[i * 4 + 2]) + w[i * 4 + 3];
    pb[0] *= sign;
    float gp[3];
    contract_actor_bwd(pb, actor_scale, acc, gp);
    gp[0] *= sign;
    float* gw = g_w2b + (rm.ray * K + slot) * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) unsafeAtomicAdd(gw + i * 4 + j, gp[i] * gm.pos[j]);
      unsafeAtomicAdd(gw + i * 4 + 3, gp[i]);
    }
  }
}

// ---- world->box transforms of the (ray, candidate) pairs from the learnable trajectories, forward and backward ----------
// dynamic_actors.py:183-197 over interpolate_trajectories_6d (utils/poses.py:90-149), rotation_6d_to_matrix
// (cameras/camera_utils.py:422-443) and pose_inverse (utils/poses.py:35-49): per keyframe the stored 6-D rotation is
// orthonormalised (Gram-Schmidt), the 9-vector (b1, b2, position) is interpolated linearly at the ray's time,
// orthonormalised again into the rotation's ROWS, and inverted.  One thread per pair, everything in registers.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float nrm(V3 a) { return fmaxf(sqrtf(dot(a, a)), 1e-12f); }  // F.normalize's clamp

struct Gs { V3 b1, b2, v; float n1, nv; };  // b1 = r1/|r1|, v = r2 - (b1.r2) b1, b2 = v/|v|
__device__ __forceinline__ Gs gram_schmidt(V3 r1, V3 r2) {
  Gs g;
  g.n1 = nrm(r1);
  g.b1 = r1 * (1.0f / g.n1);
  g.v = r2 - g.b1 * dot(g.b1, r2);
  g.nv = nrm(g.v);
  g.b2 = g.v * (1.0f / g.nv);
  return g;
}
// gradients wrt (r1, r2) from gradients wrt (b1, b2)
__device__ __forceinline__ void gram_schmidt_bwd(const Gs& g, V3 r2, V3 gb1, V3 gb2, V3& gr1, V3& gr2) {
  const V3 gv = (gb2 - g.b2 * dot(gb2, g.b2)) * (1.0f / g.nv);
  gr2 = gv - g.b1 * dot(g.b1, gv);
  gb1 = gb1 - r2 * dot(g.b1, gv) - gv * dot(g.b1, r2);
  gr1 = (gb1 - g.b1 * dot(gb1, g.b1)) * (1.0f / g.n1);
}

struct PairPose { Gs kl, kr, m; V3 u1, u2, c, b3; };
__device__ __forceinline__ PairPose pair_pose(const float* __restrict__ rot6, const float* __restrict__ pos, int64_t il, int64_t ir, float f) {
  PairPose p;
  p.kl = gram_schmidt(v3(rot6 + il * 6), v3(rot6 + il * 6 + 3));
  p.kr = gram_schmidt(v3(rot6 + ir * 6), v3(rot6 + ir * 6 + 3));
  p.u1 = p.kl.b1 + (p.kr.b1 - p.kl.b1) * f;
  p.u2 = p.kl.b2 + (p.kr.b2 - p.kl.b2) * f;
  const V3 pl = v3(pos + il * 3), pr = v3(pos + ir * 3);
  p.c = pl + (pr - pl) * f;
  p.m = gram_schmidt(p.u1, p.u2);
  p.b3 = cross(p.m.b1, p.m.b2);
  return p;
}

__global__ void __launch_bounds__(256)
actor_w2b_fwd_kernel(const int* __restrict__ cand, int64_t n_pairs, int K, int n_actors, const int64_t* __restrict__ left,
                     const int64_t* __restrict__ right, const float* __restrict__ frac, const float* __restrict__ rot6,
                     const float* __restrict__ pos, float* __restrict__ w2b, float* __restrict__ centres) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const int64_t b = i / K;
  int a = cand[i];
  a = a < 0 ? 0 : a;  // padding slots get actor 0's transform (never used: nr_actor_assign stops at -1)
  const PairPose p = pair_pose(rot6, pos, left[b] * n_actors + a, right[b] * n_actors + a, frac[b]);
  const V3 rows[3] = {p.m.b1, p.m.b2, p.b3};  // b2w rotation rows; w2b = [R^T | -R^T c]
  float* w = w2b + i * 12;
  const float cv[3] = {p.c.x, p.c.y, p.c.z};
#pragma unroll
  for (int r = 0; r < 3; ++r) {  // row r of R^T = column r of R
    const float col[3] = {r == 0 ? rows[0].x : (r == 1 ? rows[0].y : rows[0].z), r == 0 ? rows[1].x : (r == 1 ? rows[1].y : rows[1].z),
                          r == 0 ? rows[2].x : (r == 1 ? rows[2].y : rows[2].z)};
    w[r * 4] = col[0]; w[r * 4 + 1] = col[1]; w[r * 4 + 2] = col[2];
    w[r * 4 + 3] = -(col[0] * cv[0] + col[1] * cv[1] + col[2] * cv[2]);
    centres[i * 3 + r] = cv[r];
  }
}

__global__ void __launch_bounds__(256)
actor_w2b_bwd_kernel(const int* __restrict__ cand, int64_t n_pairs, int K, int n_actors, const int64_t* __restrict__ left,
                     const int64_t* __restrict__ right, const float* __restrict__ frac, const float* __restrict__ rot6,
                     const float* __restrict__ pos, const float* __restrict__ g_w2b, float* __restrict__ g_rot6,
                     float* __restrict__ g_pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const int a = cand[i];
  if (a < 0) return;
  const float* G = g_w2b + i * 12;
  bool any = false;
#pragma unroll
  for (int k = 0; k < 12; ++k) any = any || G[k] != 0.0f;
  if (!any) return;
  const int64_t b = i / K;
  const float f = frac[b];
  const int64_t il = left[b] * n_actors + a, ir = right[b] * n_actors + a;
  const PairPose p = pair_pose(rot6, pos, il, ir, f);
  // w2b[r][j] = R[j][r] (j < 3), w2b[r][3] = -sum_j R[j][r] c_j      (R rows b1, b2, b3)
  const float c[3] = {p.c.x, p.c.y, p.c.z};
  float gR[3][3], gc[3] = {0.0f, 0.0f, 0.0f};
  const V3 rows[3] = {p.m.b1, p.m.b2, p.b3};
  const float R[3][3] = {{rows[0].x, rows[0].y, rows[0].z}, {rows[1].x, rows[1].y, rows[1].z}, {rows[2].x, rows[2].y, rows[2].z}};
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      gR[j][r] = G[r * 4 + j] - c[j] * G[r * 4 + 3];
      gc[j] -= R[j][r] * G[r * 4 + 3];
    }
  V3 gb1 = {gR[0][0], gR[0][1], gR[0][2]}, gb2 = {gR[1][0], gR[1][1], gR[1][2]};
  const V3 gb3 = {gR[2][0], gR[2][1], gR[2][2]};
  gb1 = gb1 + cross(p.m.b2, gb3);  // b3 = b1 x b2
  gb2 = gb2 + cross(gb3, p.m.b1);
  V3 gu1, gu2;
  gram_schmidt_bwd(p.m, p.u2, gb1, gb2, gu1, gu2);
  const float wl = 1.0f - f, wr = f;
  V3 gr1, gr2;
  gram_schmidt_bwd(p.kl, v3(rot6 + il * 6 + 3), gu1 * wl, gu2 * wl, gr1, gr2);
  const float gl[6] = {gr1.x, gr1.y, gr1.z, gr2.x, gr2.y, gr2.z};
  gram_schmidt_bwd(p.kr, v3(rot6 + ir * 6 + 3), gu1 * wr, gu2 * wr, gr1, gr2);
  const float gr_[6] = {gr1.x, gr1.y, gr1.z, gr2.x, gr2.y, gr2.z};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (gl[k] != 0.0f) unsafeAtomicAdd(g_rot6 + il * 6 + k, gl[k]);
    if (gr_[k] != 0.0f) unsafeAtomicAdd(g_rot6 + ir * 6 + k, gr_[k]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (gc[k] * wl != 0.0f) unsafeAtomicAdd(g_pos + il * 3 + k, gc[k] * wl);
    if (gc[k] * wr != 0.0f) unsafeAtomicAdd(g_pos + ir * 3 + k, gc[k] * wr);
  }
}

}  // namespace

extern "C" int nr_actor_w2b_fwd(const int* cand, int64_t n_rays, int K, int n_actors, const int64_t* left, const int64_t* right,
                                const float* frac, const float* rotations_6d, const float* positions, float* w2b, float* centres,
                                nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!cand || !left || !right || !frac || !rotations_6d || !positions || !w2b || !centres || K < 1 || n_actors < 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(actor_w2b_fwd_kernel, dim3((unsigned)nr_cdiv(n_rays * K, 256)), dim3(256), 0, nr_s(stream), cand, n_rays * K, K,
                     n_actors, left, right, frac, rotations_6d, positions, w2b, centres);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_w2b_bwd(const int* cand, int64_t n_rays, int K, int n_actors, const int64_t* left, const int64_t* right,
                                const float* frac, const float* rotations_6d, const float* positions, const float* grad_w2b,
                                float* grad_rotations_6d, float* grad_positions, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!cand || !left || !right || !frac || !rotations_6d || !positions || !grad_w2b || !grad_rotations_6d || !grad_positions ||
      K < 1 || n_actors < 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(actor_w2b_bwd_kernel, dim3((unsigned)nr_cdiv(n_rays * K, 256)), dim3(256), 0, nr_s(stream), cand, n_rays * K, K,
                     n_actors, left, right, frac, rotations_6d, positions, grad_w2b, grad_rotations_6d, grad_positions);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_candidates(const float* origins, const float* directions, const float* euclid, int64_t n_rays, int S,
                                   const int64_t* left, const int64_t* right, const float* frac, const float* positions,
                                   const uint8_t* present, const float* bounds, int n_actors, int K, int* cand, int* overflow,
                                   nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!origins || !directions || !euclid || !left || !right || !frac || !positions || !present || !bounds || !cand || !overflow ||
      S < 1 || n_actors < 1 || K < 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(actor_candidates_kernel, dim3((unsigned)nr_cdiv(n_rays, 256)), dim3(256), 0, nr_s(stream), origins, directions,
                     euclid, S, left, right, frac, positions, present, bounds, n_actors, n_rays, K, cand, overflow);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_keyframes(const float* times, int64_t n_rays, const float* timestamps, int n_timestamps, int64_t* left,
                                  int64_t* right, float* frac, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!times || !timestamps || !left || !right || !frac || n_timestamps < 1 || n_rays < 0) return NR_EINVAL;
  hipLaunchKernelGGL(actor_keyframes_kernel, dim3((unsigned)nr_cdiv(n_rays, 256)), dim3(256), 0, nr_s(stream), times, n_rays,
                     timestamps, n_timestamps, left, right, frac);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_assign(const float* origins, const float* directions, const float* pixel_area, const float* euclid,
                               int64_t n_rays, int S, int sample_major_rows, const int* cand, int K, const float* w2b,
                               const float* centres, const float* bounds, float actor_scale, const float* flip, int* slot_of_row,
                               float* x01a, float* std01a, float* dirs_sample, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!origins || !directions || !pixel_area || !euclid || !cand || !w2b || !centres || !bounds || !slot_of_row || !x01a ||
      !std01a || S < 1 || K < 1 || n_rays < 0 || !(actor_scale > 0) || sample_major_rows < 0 || sample_major_rows > n_rays)
    return NR_EINVAL;
  hipLaunchKernelGGL(actor_assign_kernel, dim3((unsigned)nr_cdiv(n_rays * S, 256)), dim3(256), 0, nr_s(stream), origins, directions,
                     pixel_area, euclid, n_rays, S, sample_major_rows, cand, K, w2b, centres, bounds, actor_scale, flip, slot_of_row,
                     x01a, std01a, dirs_sample);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_encode_fwd(const float* x01a, const float* std01a, const int* slot_of_row, const int* cand, int K,
                                   int64_t n_rays, int S, int sample_major_rows, const float* tables, const int* table_of_actor,
                                   const float* scalings, int L, int F, int log2T, float* feats, int64_t sn, int64_t sl,
                                   int static_levels, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!x01a || !std01a || !slot_of_row || !cand || !tables || !scalings || !feats || L < 1 || static_levels < L || log2T < 1 ||
      log2T > 30 || S < 1 || K < 1 || n_rays < 0)
    return NR_EINVAL;
  const int64_t n = n_rays * S;
  dim3 grid((unsigned)nr_cdiv(n, 256)), block(256);
#define CALL(FF) hipLaunchKernelGGL(actor_encode_fwd_kernel<FF>, grid, block, 0, nr_s(stream), x01a, std01a, slot_of_row, cand, K, S, \
                                    sample_major_rows, n, tables, table_of_actor, scalings, L, log2T, feats, sn, sl, static_levels)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_actor_encode_bwd(const float* x01a, const float* std01a, const int* slot_of_row, const int* cand, int K,
                                   int64_t n_rays, int S, int sample_major_rows, const float* tables, const int* table_of_actor,
                                   const float* scalings, int L, int F, int log2T, float* g_feats, int64_t sn, int64_t sl,
                                   int static_levels, float* g_tables,
                                   const float* origins, const float* directions, const float* pixel_area, const float* euclid,
                                   const float* w2b, float actor_scale, const float* flip, float* g_w2b, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!x01a || !std01a || !slot_of_row || !cand || !tables || !scalings || !g_feats || !g_tables || L < 1 || static_levels < L ||
      log2T < 1 || log2T > 30 || S < 1 || K < 1 || n_rays < 0)
    return NR_EINVAL;
  if (g_w2b != nullptr && (!origins || !directions || !pixel_area || !euclid || !w2b || !(actor_scale > 0))) return NR_EINVAL;
  const int64_t n = n_rays * S;
  dim3 grid((unsigned)nr_cdiv(n, 256)), block(256);
#define CALL(FF) hipLaunchKernelGGL(actor_encode_bwd_kernel<FF>, grid, block, 0, nr_s(stream), x01a, std01a, slot_of_row, cand, K, S, \
                                    sample_major_rows, n, tables, table_of_actor, scalings, L, log2T, g_feats, sn, sl, static_levels, g_tables, \
                                    origins, directions, pixel_area, euclid, w2b, actor_scale, flip, g_w2b)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
