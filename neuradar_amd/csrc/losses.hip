// Loss tail of the training step as three kernels: direct supervision of the
// rendered features/depth, mip-NeRF-360 distortion and the ZipNeRF anti-aliased inter-level loss
// (reference model_components/losses.py:137-156,626-705).  Each kernel produces the loss value AND
// the gradient w.r.t. the weights it consumes in one pass (the inter-level target is detached in the
// reference, so its gradient is elementwise in the proposal weights).  One wavefront per ray.
#include "nr_common.h"
#include "weights_dev.h"

namespace {

constexpr int kWavesPerBlock = 4;
// loss values are summed into NR_LOSS_SLOTS partial sums (slot = global wave index mod slots): thousands of
// atomics on ONE address serialise at the memory side (~15 ns each), spread over 64 they do not.
__device__ __forceinline__ float* loss_slot(float* loss) { return loss + nr_loss_slot_index(); }

__global__ void __launch_bounds__(256)
supervision_loss_kernel(const float* __restrict__ features, int feat_stride, const float* __restrict__ target_f, int C,
                        const float* __restrict__ depth, const float* __restrict__ target_d, int64_t n_rays,
                        float rgb_mult, float depth_mult, float* __restrict__ g_features, float* __restrict__ g_depth,
                        float* __restrict__ loss) {
  // rgb_mult * mean((f - t)^2) + depth_mult * mean(|d - t|)
  const int64_t total = n_rays * C;
  float acc = 0.0f;
  const float kf = rgb_mult / (float)total, kd = depth_mult / (float)n_rays;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / C;
    const int c = (int)(i - b * C);
    const float diff = features[b * feat_stride + c] - target_f[i];
    acc += kf * diff * diff;
    g_features[b * feat_stride + c] = 2.0f * kf * diff;
    if (c == 0) {
      const float dd = depth[b] - target_d[b];
      acc += kd * fabsf(dd);
      g_depth[b] = dd > 0.0f ? kd : (dd < 0.0f ? -kd : 0.0f);
    }
  }
  acc = nr_wave_sum(acc);
  if (nr_lane() == 0 && acc != 0.0f) unsafeAtomicAdd(loss_slot(loss), acc);
}

// losses.py:137-157.  c [n_rays, c_stride] s-space edges, w [n_rays, w_stride]; the first n_used
// samples take part (the sky sample is dropped, neuradar.py:515,534).  g_w [n_rays, w_stride] is
// overwritten (entries >= n_used get 0).
__global__ void __launch_bounds__(256)
distortion_loss_kernel(const float* __restrict__ c, int c_stride, const float* __restrict__ w, int w_stride, int n_used,
                       int64_t n_rays, float mult, float* __restrict__ g_w, float* __restrict__ loss) {
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  const bool on = lane < n_used;
  const float c0 = on ? c[ray * c_stride + lane] : 0.0f, c1 = on ? c[ray * c_stride + lane + 1] : 0.0f;
  const float wi = on ? w[ray * w_stride + lane] : 0.0f;
  const float mid = (c1 + c0) / 2.0f;
  float inner = 0.0f;
  for (int j = 0; j < n_used; ++j) inner += __shfl(wi, j, NR_WAVE) * fabsf(mid - __shfl(mid, j, NR_WAVE));
  const float k = mult / (float)n_rays;
  float l = wi * inner + wi * wi * (c1 - c0) / 3.0f;
  l = nr_wave_sum(on ? l : 0.0f);
  if (lane < w_stride) g_w[ray * w_stride + lane] = on ? k * (2.0f * inner + 2.0f * wi * (c1 - c0) / 3.0f) : 0.0f;
  if (lane == 0) unsafeAtomicAdd(loss_slot(loss), k * l);
}

// losses.py:626-705 for ONE proposal level.  Final level: c [.., n_used+1 edges], w [.., n_used]
// (detached); proposal level: cp [n_rays, Sp+1], wp [n_rays, Sp].  n_used <= 31 so that the 2*(n_used+1)
// blur knots fit one wavefront.  Outputs g_wp [n_rays, Sp] (overwritten) and loss +=.
// inclusive prefix sum over lanes 0..n-1 in lane order, one rounding per addition (what a sequential cumsum does)
__device__ __forceinline__ float seq_incl_sum(float v, int n) {
  float acc = 0.0f, out = 0.0f;
  const int lane = nr_lane();
  for (int j = 0; j < n; ++j) {
    acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    out = lane == j ? acc : out;
  }
  return lane >= n ? acc : out;
}

constexpr int kMaxKnots = 66;
constexpr int kMaxProp = 256;

// ITEMS > 0: the gradient is carried on through RaySamples.get_weights of the proposal level in the
// same launch (density_p, euclid_p -> g_density_p; nr_weights_from_density_bwd's arithmetic), the
// proposal-weight gradient never leaves the wavefront's LDS.
template <int ITEMS>
__global__ void __launch_bounds__(256)
interlevel_loss_kernel(const float* __restrict__ c, int c_stride, const float* __restrict__ w, int w_stride, int n_used,
                       const float* __restrict__ cp, const float* __restrict__ wp, int Sp, int64_t n_rays, float pulse,
                       float mult, float* __restrict__ g_wp, float* __restrict__ loss,
                       const float* __restrict__ density_p, const float* __restrict__ euclid_p,
                       float* __restrict__ g_density_p, nr_lidar_sup_t lidar) {
  __shared__ float s_knot[kWavesPerBlock][kMaxKnots];   // c_  : [0, sorted knots, 1]
  __shared__ float s_val[kWavesPerBlock][kMaxKnots];    // w_  : blurred density at the knots
  __shared__ float s_cdf[kWavesPerBlock][kMaxKnots];    // cdf : integral of the piecewise-linear density
  __shared__ float s_q[kWavesPerBlock][kMaxProp + 1];   // cdf interpolated at the proposal edges
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (ray >= n_rays) return;
  const int lane = nr_lane();
  float* knot = s_knot[wave];
  float* val = s_val[wave];
  float* cdf = s_cdf[wave];
  float* q = s_q[wave];
  const int E = n_used + 1;  // edges of the final level
  const int K = 2 * E;       // blur knots
  // every global input of the ray is requested up front: the wave is latency-bound, and the LDS fences
  // below would otherwise pin each load to the phase that consumes it
  constexpr int kPerLane = kMaxProp / NR_WAVE;
  float xq[kPerLane + 1], pw[kPerLane];
#pragma unroll
  for (int t = 0; t <= kPerLane; ++t) {
    const int j = lane + t * NR_WAVE;
    xq[t] = j <= Sp ? cp[ray * (Sp + 1) + j] : 0.0f;
    if (t < kPerLane) pw[t] = j < Sp ? wp[ray * Sp + j] : 0.0f;
  }
  constexpr int NI = ITEMS > 0 ? ITEMS : 1;
  float dens_p[NI], delta_p[NI];
  if constexpr (ITEMS > 0) nr_weights_bwd_load<ITEMS>(density_p + ray * Sp, euclid_p + ray * (Sp + 1), Sp, dens_p, delta_p);
  // ---- final-level histogram, remaining mass on the last kept sample (:663-664) ----
  const float wi = lane < n_used ? w[ray * w_stride + lane] : 0.0f;
  const float acc = nr_wave_sum(wi);
  const float ci = lane < E ? c[ray * c_stride + lane] : 0.0f;
  const float c_next = __shfl_down(ci, 1, NR_WAVE);
  float dens = 0.0f;  // w_norm (:666)
  if (lane < n_used) dens = (lane == n_used - 1 ? wi + (1.0f - acc) : wi) / (c_next - ci);
  // ---- box blur (:626-635): knots c-r (lanes 0..E-1) and c+r (lanes E..2E-1), merged by rank ----
  const int src = lane < E ? lane : lane - E;                 // which edge this lane's knot comes from
  const float edge = __shfl(ci, src, NR_WAVE);
  const float mine = lane < E ? edge - pulse : edge + pulse;
  const float d_prev = __shfl_up(dens, 1, NR_WAVE);
  const float jump_e = ((lane < n_used ? dens : 0.0f) - (lane >= 1 && lane <= n_used ? d_prev : 0.0f)) / (2.0f * pulse);
  const float jump = __shfl(jump_e, src, NR_WAVE) * (lane < E ? 1.0f : -1.0f);  // slope change at this knot
  // rank in the merged order: own index + number of knots of the OTHER list that sort before it.
  // The edges go through LDS for the search: lanes leave the loop at different times, and a shuffle
  // cannot read a lane that has already dropped out of the loop.
  if (lane < E) q[lane] = ci;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  int other = 0;
  if (lane < K) {
    int lo = 0, hi = E;
    while (lo < hi) {  // the other list is sorted ascending: c[k] -+ pulse
      const int m = (lo + hi) >> 1;
      const float o = q[m] + (lane < E ? pulse : -pulse);
      const bool before = lane < E ? (o < mine) : (o <= mine);
      if (before) lo = m + 1; else hi = m;
    }
    other = lo;
  }
  const int rank = src + other;
  if (lane < K) {
    knot[1 + rank] = mine;
    val[1 + rank] = jump;  // temporarily: slope jump at this knot
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  // sorted order: lane j holds knot j (j < K)
  const float xj = lane < K ? knot[1 + lane] : 0.0f;
  const float sj = lane < K ? val[1 + lane] : 0.0f;
  const float x_next = __shfl_down(xj, 1, NR_WAVE);
  // y2 is defined on the first K-1 sorted knots (:632); slope after knot j = cumsum(y2)[j]
  // The three running sums (slope, value, cdf) are SEQUENTIAL in lane order, like torch.cumsum on the reference's side:
  // the slope jumps are O(w / dc / pulse) ~ 1e5 with alternating signs, and a tree scan's different rounding order moved
  // the gradient of the fine pulse by up to 3e-2 of its value on near-empty proposal bins.  One loop of <= 64 steps carries
  // all three sums as wave-uniform values (every lane computes them from the broadcast inputs; lane j keeps step j's).
  const float s_in = lane < K - 1 ? sj : 0.0f;
  const float dx_in = lane < K - 1 ? x_next - xj : 0.0f;
  float slope_run = 0.0f, y_run = 0.0f, cdf_run = 0.0f, y_prev_c = 0.0f;
  float yr = 0.0f, y_at = 0.0f, cdf_incl = 0.0f;
#pragma unroll 4
  for (int j = 0; j < K; ++j) {
    const float sj_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s_in), j));
    const float dx_b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dx_in), j));
    slope_run += sj_b;                         // cumsum(y2)[j]                      (:632)
    y_run += dx_b * slope_run;                 // cumsum((x[1:] - x[:-1]) * slope)[j] (:633)
    const float y_c = fmaxf(y_run, 0.0f);      // clamped value at knot j+1
    cdf_run += 0.5f * (y_c + y_prev_c) * dx_b; // integral up to knot j+1            (:685)
    const bool me = lane == j;
    yr = me ? y_c : yr;
    y_at = me ? y_prev_c : y_at;
    cdf_incl = me ? cdf_run : cdf_incl;
    y_prev_c = y_c;
  }
  float cdf_at = __shfl_up(cdf_incl, 1, NR_WAVE);
  if (lane == 0) cdf_at = 0.0f;
  if (lane < K) {
    val[1 + lane] = y_at;
    cdf[1 + lane] = cdf_at;
  }
  if (lane == 0) {  // padding (:688-691)
    knot[0] = 0.0f; val[0] = 0.0f; cdf[0] = 0.0f;
    knot[K + 1] = 1.0f; val[K + 1] = 0.0f; cdf[K + 1] = 1.0f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  // ---- query the piecewise-quadratic cdf at the proposal edges (:638-651) ----
  const int NK = K + 2;
#pragma unroll
  for (int tq = 0; tq <= kPerLane; ++tq) {
    const int j = lane + tq * NR_WAVE;
    if (j > Sp) continue;
    const float x = xq[tq];
    int lo = 0, hi = NK;  // searchsorted(left): first index with knot >= x
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (knot[m] < x) lo = m + 1; else hi = m;
    }
    const int left = max(lo - 1, 0), right = min(lo, NK - 1);
    const float x0 = knot[left], x1 = knot[right], v0 = val[left], v1 = val[right], f0 = cdf[left];
    float t = (x - x0) / (x1 - x0);
    t = isnan(t) ? 0.0f : nr_nan_to_num(t);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    q[j] = f0 + (x - x0) * (v0 + v1 * t + v0 * (1.0f - t)) * 0.5f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  // ---- loss and its gradient w.r.t. the proposal weights (:700-704) ----
  const float k = mult / (float)n_rays;
  float l = 0.0f, lc = 0.0f;
  const bool carve = lidar.is_lidar != nullptr && euclid_p != nullptr && lidar.is_lidar[ray] != 0;
  const bool carve_ret = carve && lidar.did_return[ray] != 0;
  const float carve_range = carve ? lidar.range[ray] : 0.0f;
  // the level's own lidar depth loss (depth_loss_i, neuradar.py:641-648): depth = sum_j w_j mid_j over ALL its samples
  float g_pdepth = 0.0f, l_pdepth = 0.0f;
  if (carve && lidar.depth_weight > 0.0f) {
    float dsum = 0.0f;
#pragma unroll
    for (int t = 0; t < kPerLane; ++t) {
      const int j = lane + t * NR_WAVE;
      if (j < Sp) dsum += pw[t] * ((euclid_p[ray * (Sp + 1) + j] + euclid_p[ray * (Sp + 1) + j + 1]) / 2.0f);
    }
    dsum = nr_wave_sum(dsum);
    const float scale = carve_ret ? 1.0f : lidar.non_return_loss_mult;
    const float diff = dsum - (carve_ret ? carve_range : fmaxf(dsum, lidar.non_return_distance));
    l_pdepth = lidar.depth_weight * scale * fabsf(diff);
    g_pdepth = lidar.depth_weight * scale * (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
  }
  float gj[kPerLane];
#pragma unroll
  for (int t = 0; t < kPerLane; ++t) {
    const int j = lane + t * NR_WAVE;
    gj[t] = 0.0f;
    if (j < Sp) {
      const float target = q[j + 1] - q[j];
      const float p = pw[t];
      const float ex = fmaxf(target - p, 0.0f), den = p + 1e-5f;
      l += ex * ex / den;
      gj[t] = k * (-2.0f * ex / den - ex * ex / (den * den));
      if (carve) {  // prop_weights_loss_i = sum((w * mask)^2), neuradar.py:529-531
        const float mid = (euclid_p[ray * (Sp + 1) + j] + euclid_p[ray * (Sp + 1) + j + 1]) * 0.5f;
        const bool close = carve_ret ? fabsf(carve_range - mid) < lidar.carving_epsilon : mid < lidar.non_return_distance;
        if (!close) {
          lc += p * p;
          gj[t] += 2.0f * lidar.weight * p;
        }
        gj[t] += g_pdepth * ((euclid_p[ray * (Sp + 1) + j] + euclid_p[ray * (Sp + 1) + j + 1]) / 2.0f);
      }
      if (g_wp != nullptr) g_wp[ray * Sp + j] = gj[t];
    }
  }
  l = k * nr_wave_sum(l);
  if (carve) l += lidar.weight * nr_wave_sum(lc) + l_pdepth;
  if (lane == 0) unsafeAtomicAdd(loss_slot(loss), l);
  if constexpr (ITEMS > 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();  // every lane has read its q[j], q[j+1]
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
    for (int t = 0; t < kPerLane; ++t) {
      const int j = lane + t * NR_WAVE;
      if (j < Sp) q[j] = gj[t];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    nr_weights_bwd_ray<ITEMS>(dens_p, delta_p, [&](int s) { return q[s]; }, Sp, g_density_p + ray * Sp);
  }
}

// Temporal appearance embedding (neuradar.py:556-565) next to the rendered features of rays [row0, row0 + n_rows)
__device__ __forceinline__ void appearance_index(float t, int64_t sensor, float duration, int E, int64_t& before, int64_t& after,
                                                 float& ratio) {
  const float ti = t / duration * (float)E;
  float b = floorf(ti);
  b = fminf(fmaxf(b, 0.0f), (float)(E - 1));
  const float a = fminf(fmaxf(b + 1.0f, 0.0f), (float)(E - 1));
  ratio = ti - b;
  before = (int64_t)b + sensor * E;
  after = (int64_t)a + sensor * E;
}

__global__ void __launch_bounds__(256)
appearance_concat_fwd_kernel(const float* __restrict__ features, int C, const float* __restrict__ table, int A,
                             const float* __restrict__ times, const int64_t* __restrict__ sensor_idx, float duration, int E,
                             int64_t row0, int64_t n_rows, float* __restrict__ out) {
  const int W = C + A;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * W) return;
  const int64_t r = i / W;
  const int c = (int)(i - r * W);
  const int64_t ray = row0 + r;
  if (c < C) {
    out[i] = features[ray * C + c];
    return;
  }
  int64_t b, a;
  float ratio;
  appearance_index(times[ray], sensor_idx[ray], duration, E, b, a, ratio);
  out[i] = table[b * A + (c - C)] * (1.0f - ratio) + table[a * A + (c - C)] * ratio;
}

constexpr int kAppLds = 4096;  // table gradients of up to this many floats are summed per block in LDS first
__global__ void __launch_bounds__(256)
appearance_concat_bwd_kernel(const float* __restrict__ g_out, int C, int A, const float* __restrict__ times,
                             const int64_t* __restrict__ sensor_idx, float duration, int E, int64_t row0, int64_t n_rows,
                             float* __restrict__ g_features, float* __restrict__ g_table, int64_t table_rows) {
  __shared__ float acc[kAppLds];
  const bool use_lds = table_rows * A <= kAppLds;
  if (use_lds) {
    for (int k = threadIdx.x; k < table_rows * A; k += blockDim.x) acc[k] = 0.0f;
    __syncthreads();
  }
  const int W = C + A;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * W; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    const int64_t ray = row0 + r;
    const float g = g_out[i];
    if (c < C) {
      g_features[ray * C + c] = g;
      continue;
    }
    if (g == 0.0f) continue;
    int64_t b, a;
    float ratio;
    appearance_index(times[ray], sensor_idx[ray], duration, E, b, a, ratio);
    if (use_lds) {
      atomicAdd(&acc[b * A + (c - C)], g * (1.0f - ratio));
      atomicAdd(&acc[a * A + (c - C)], g * ratio);
    } else {
      unsafeAtomicAdd(g_table + b * A + (c - C), g * (1.0f - ratio));
      unsafeAtomicAdd(g_table + a * A + (c - C), g * ratio);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < table_rows * A; k += blockDim.x)
      if (acc[k] != 0.0f) unsafeAtomicAdd(g_table + k, acc[k]);
  }
}

// neuradar.py:432-452,624-636: intensity MSE on returning rays, ray-drop BCE with logits on all lidar rays
__global__ void __launch_bounds__(256)
lidar_head_loss_kernel(const float* __restrict__ y, const float* __restrict__ target_i, const uint8_t* __restrict__ did_return,
                       int64_t n, const float* __restrict__ inv_n_ret, float ki, float kd, float* __restrict__ g_y,
                       float* __restrict__ loss) {
  float acc = 0.0f;
  const float inv_ret = inv_n_ret[0], inv_n = 1.0f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float y0 = y[i * 2], y1 = y[i * 2 + 1];
    const bool ret = did_return[i] != 0;
    float g0 = 0.0f;
    if (ret) {
      const float s = 1.0f / (1.0f + expf(-y0));
      const float d = s - target_i[i];
      acc += ki * inv_ret * d * d;
      g0 = 2.0f * ki * inv_ret * d * s * (1.0f - s);
    }
    // BCE with logits, target z = !did_return: max(y,0) - y z + log(1 + exp(-|y|))
    const float z = ret ? 0.0f : 1.0f;
    acc += kd * inv_n * (fmaxf(y1, 0.0f) - y1 * z + log1pf(expf(-fabsf(y1))));
    const float sg = 1.0f / (1.0f + expf(-y1));
    g_y[i * 2] = g0;
    g_y[i * 2 + 1] = kd * inv_n * (sg - z);
  }
  acc = nr_wave_sum(acc);
  if (nr_lane() == 0 && acc != 0.0f) unsafeAtomicAdd(loss_slot(loss), acc);
}

// ---- lidar depth / intensity / ray-drop losses with the batch quantile (neuradar.py:612-636) ----
__device__ __forceinline__ float lidar_unreduced(const float* __restrict__ depth, const nr_lidar_losses_t& c, int64_t i) {
  const int64_t ray = c.row0 + i;
  const float d = depth[ray];
  const bool ret = c.did_return[ray] != 0;
  const float target = ret ? c.range[ray] : fmaxf(d, c.non_return_distance);
  return fabsf(target - d) * (ret ? 1.0f : c.non_return_loss_mult);
}

__global__ void __launch_bounds__(256)
lidar_unreduced_kernel(const float* __restrict__ depth, nr_lidar_losses_t c, float* __restrict__ unreduced) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) unreduced[i] = lidar_unreduced(depth, c, i);
}

// Order statistics without a sort: rank_i = #{j : x_j < x_i} + #{j < i : x_j == x_i}; the elements whose ranks are
// floor / ceil of quantile * (n - 1) are the two values torch.quantile interpolates between.  The values are non-negative
// floats, whose bit patterns order like unsigned integers: key = (bits << 32 | index) turns the rank into ONE 64-bit compare
// per pair.  A block ranks 16 elements, sixteen threads per element each walking a sixteenth of every LDS tile of keys
// (n^2 / 64 wave-compares spread over n / 16 blocks: 4 661 lidar rays keep every SIMD of the chip busy for a few microseconds).
constexpr int kRankTile = 4096, kRankElems = 16, kRankParts = 256 / kRankElems;
__global__ void __launch_bounds__(256)
lidar_rank_kernel(const float* __restrict__ unreduced, nr_lidar_losses_t c, float* __restrict__ stats) {
  __shared__ unsigned long long tile[kRankTile];
  __shared__ int ranks[kRankElems];
  const int64_t n = c.n;
  const int e = threadIdx.x & (kRankElems - 1), part = threadIdx.x / kRankElems;
  const int64_t i = (int64_t)blockIdx.x * kRankElems + e;  // my element
  const float xi = i < n ? unreduced[i] : 0.0f;
  const unsigned long long ki = ((unsigned long long)__float_as_uint(xi) << 32) | (unsigned)i;
  if (threadIdx.x < kRankElems) ranks[threadIdx.x] = 0;
  int rank = 0;
  for (int64_t t0 = 0; t0 < n; t0 += kRankTile) {
    const int len = (int)min((int64_t)kRankTile, n - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < len; k += 256)
      tile[k] = ((unsigned long long)__float_as_uint(unreduced[t0 + k]) << 32) | (unsigned)(t0 + k);
    __syncthreads();
    const int q0 = part * (kRankTile / kRankParts), q1 = min(len, q0 + kRankTile / kRankParts);
#pragma unroll 8
    for (int k = q0; k < q1; ++k) rank += tile[k] < ki ? 1 : 0;
  }
  if (i < n && rank) atomicAdd(&ranks[e], rank);
  __syncthreads();
  if (part == 0 && i < n) {
    const float pos = c.quantile * (float)(n - 1);  // ATen computes the rank in the input's dtype
    const int lo = (int)floorf(pos), hi = (int)ceilf(pos);
    if (ranks[e] == lo) stats[0] = xi;
    if (ranks[e] == hi) stats[1] = xi;
  }
}

__global__ void __launch_bounds__(1024)
lidar_quantile_counts_kernel(const float* __restrict__ unreduced, nr_lidar_losses_t c, float* __restrict__ stats) {
  const float pos = c.quantile * (float)(c.n - 1);
  const float w = pos - floorf(pos), a = stats[0], b = stats[1];
  const float q = w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);  // at::lerp
  int cm = 0, cr = 0;
  for (int64_t i = threadIdx.x; i < c.n; i += blockDim.x) {
    const bool m = unreduced[i] < q;
    cm += m ? 1 : 0;
    cr += (m && c.did_return[c.row0 + i] != 0) ? 1 : 0;
  }
  __shared__ int s_m[16], s_r[16];
  const float fm = nr_wave_sum((float)cm), fr = nr_wave_sum((float)cr);  // (counts < 2^24: exact in float)
  if (nr_lane() == 0) { s_m[threadIdx.x >> 6] = (int)fm; s_r[threadIdx.x >> 6] = (int)fr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tm = 0, tr = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { tm += s_m[k]; tr += s_r[k]; }
    stats[2] = q;
    stats[3] = tm > 0 ? 1.0f / (float)tm : 0.0f;
    stats[4] = tr > 0 ? 1.0f / (float)tr : 0.0f;
    stats[5] = (float)tm;
    stats[6] = (float)tr;
  }
}

__global__ void __launch_bounds__(256)
lidar_losses_kernel(const float* __restrict__ depth, const float* __restrict__ y, nr_lidar_losses_t c,
                    const float* __restrict__ unreduced, const float* __restrict__ stats, float* __restrict__ g_depth,
                    float* __restrict__ g_y, float* __restrict__ loss) {
  const float q = stats[2], inv_m = stats[3], inv_r = stats[4], inv_n = 1.0f / (float)c.n;
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ray = c.row0 + i;
    const bool ret = c.did_return[ray] != 0;
    const float un = unreduced[i];
    const bool in = un < q;
    const float d = depth[ray];
    const float diff = d - (ret ? c.range[ray] : fmaxf(d, c.non_return_distance));
    float gd = 0.0f;
    if (in) {  // depth_mult * mean(unreduced[mask])
      acc += c.depth_mult * inv_m * un;
      gd = c.depth_mult * inv_m * (ret ? 1.0f : c.non_return_loss_mult) * (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f));
    }
    g_depth[ray] = gd;
    const float y0 = y[i * 2], y1 = y[i * 2 + 1];
    float g0 = 0.0f;
    if (in && ret) {  // intensity MSE on the returning rays inside the quantile (:627-632)
      const float s = 1.0f / (1.0f + expf(-y0));
      const float e = s - c.target_intensity[ray];
      acc += c.intensity_mult * inv_r * e * e;
      g0 = 2.0f * c.intensity_mult * inv_r * e * s * (1.0f - s);
    }
    const float z = ret ? 0.0f : 1.0f;  // BCE with logits, target = !did_return (:634-636)
    acc += c.ray_drop_mult * inv_n * (fmaxf(y1, 0.0f) - y1 * z + log1pf(expf(-fabsf(y1))));
    g_y[i * 2] = g0;
    g_y[i * 2 + 1] = c.ray_drop_mult * inv_n * (1.0f / (1.0f + expf(-y1)) - z);
  }
  acc = nr_wave_sum(acc);
  if (nr_lane() == 0 && acc != 0.0f) unsafeAtomicAdd(loss_slot(loss), acc);
}

}  // namespace

inline bool lidar_cfg_ok(const nr_lidar_losses_t* c) {
  return c && c->did_return && c->range && c->target_intensity && c->row0 >= 0 && c->n >= 0 && c->quantile >= 0.0f && c->quantile <= 1.0f;
}

extern "C" int nr_lidar_depth_quantile(const float* depth, const nr_lidar_losses_t* cfg, float* unreduced, float* stats,
                                       nr_stream_t stream) {
  if (!lidar_cfg_ok(cfg) || !depth || !unreduced || !stats) return NR_EINVAL;
  if (cfg->n == 0) return 0;
  hipLaunchKernelGGL(lidar_unreduced_kernel, dim3((unsigned)nr_cdiv(cfg->n, 256)), dim3(256), 0, nr_s(stream), depth, *cfg, unreduced);
  hipLaunchKernelGGL(lidar_rank_kernel, dim3((unsigned)nr_cdiv(cfg->n, kRankElems)), dim3(256), 0, nr_s(stream), unreduced, *cfg, stats);
  hipLaunchKernelGGL(lidar_quantile_counts_kernel, dim3(1), dim3(1024), 0, nr_s(stream), unreduced, *cfg, stats);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_lidar_losses(const float* depth, const float* y, const nr_lidar_losses_t* cfg, const float* unreduced,
                               const float* stats, float* grad_depth, float* grad_y, float* loss, nr_stream_t stream) {
  if (!lidar_cfg_ok(cfg) || !depth || !y || !unreduced || !stats || !grad_depth || !grad_y || !loss) return NR_EINVAL;
  if (cfg->n == 0) return 0;
  const int64_t want = nr_cdiv(cfg->n, 256);
  hipLaunchKernelGGL(lidar_losses_kernel, dim3((unsigned)(want < 256 ? want : 256)), dim3(256), 0, nr_s(stream), depth, y, *cfg,
                     unreduced, stats, grad_depth, grad_y, loss);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_appearance_concat_fwd(const float* features, int C, const float* table, int A, const float* times,
                                        const int64_t* sensor_idx, float duration, int E, int64_t row0, int64_t n_rows, float* out,
                                        nr_stream_t stream) {
  if (n_rows == 0) return 0;
  if (!features || !table || !times || !sensor_idx || !out || C < 1 || A < 1 || E < 1 || !(duration > 0) || row0 < 0 || n_rows < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(appearance_concat_fwd_kernel, dim3((unsigned)nr_cdiv(n_rows * (C + A), 256)), dim3(256), 0, nr_s(stream), features,
                     C, table, A, times, sensor_idx, duration, E, row0, n_rows, out);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_appearance_concat_bwd(const float* g_out, int C, int A, const float* times, const int64_t* sensor_idx, float duration,
                                        int E, int64_t row0, int64_t n_rows, float* g_features, float* g_table, int64_t table_rows,
                                        nr_stream_t stream) {
  if (n_rows == 0) return 0;
  if (!g_out || !times || !sensor_idx || !g_features || !g_table || C < 1 || A < 1 || E < 1 || !(duration > 0) || row0 < 0 ||
      n_rows < 0 || table_rows < 1)
    return NR_EINVAL;
  const int64_t want = nr_cdiv(n_rows * (C + A), 256);
  hipLaunchKernelGGL(appearance_concat_bwd_kernel, dim3((unsigned)(want < 256 ? want : 256)), dim3(256), 0, nr_s(stream), g_out, C, A,
                     times, sensor_idx, duration, E, row0, n_rows, g_features, g_table, table_rows);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_lidar_head_loss(const float* y, const float* target_intensity, const uint8_t* did_return, int64_t n,
                                  const float* inv_n_returning, float intensity_mult, float ray_drop_mult, float* g_y, float* loss,
                                  nr_stream_t stream) {
  if (n == 0) return 0;
  if (!y || !target_intensity || !did_return || !inv_n_returning || !g_y || !loss || n < 0) return NR_EINVAL;
  const int64_t want = nr_cdiv(n, 256);
  hipLaunchKernelGGL(lidar_head_loss_kernel, dim3((unsigned)(want < 256 ? want : 256)), dim3(256), 0, nr_s(stream), y, target_intensity,
                     did_return, n, inv_n_returning, intensity_mult, ray_drop_mult, g_y, loss);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_supervision_loss(const float* features, int feat_stride, const float* target_f, int C,
                                   const float* depth, const float* target_d, int64_t n_rays, float rgb_mult,
                                   float depth_mult, float* g_features, float* g_depth, float* loss,
                                   nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!features || !target_f || !depth || !target_d || !g_features || !g_depth || !loss || C < 1 || feat_stride < C || n_rays < 0)
    return NR_EINVAL;
  const int64_t want = nr_cdiv(n_rays * C, 256);
  hipLaunchKernelGGL(supervision_loss_kernel, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, nr_s(stream),
                     features, feat_stride, target_f, C, depth, target_d, n_rays, rgb_mult, depth_mult, g_features,
                     g_depth, loss);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_distortion_loss(const float* c, int c_stride, const float* w, int w_stride, int n_used, int64_t n_rays,
                                  float mult, float* g_w, float* loss, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!c || !w || !g_w || !loss || n_used < 1 || n_used > NR_WAVE || w_stride < n_used || w_stride > NR_WAVE ||
      c_stride < n_used + 1 || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(distortion_loss_kernel, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     c, c_stride, w, w_stride, n_used, n_rays, mult, g_w, loss);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_interlevel_loss(const float* c, int c_stride, const float* w, int w_stride, int n_used, const float* cp,
                                  const float* wp, int Sp, int64_t n_rays, float pulse, float mult, float* g_wp,
                                  float* loss, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!c || !w || !cp || !wp || !g_wp || !loss || n_used < 1 || n_used > 31 || w_stride < n_used ||
      c_stride < n_used + 1 || Sp < 1 || Sp > kMaxProp || !(pulse > 0.0f) || n_rays < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(interlevel_loss_kernel<0>, dim3((unsigned)nr_cdiv(n_rays, kWavesPerBlock)), dim3(256), 0, nr_s(stream),
                     c, c_stride, w, w_stride, n_used, cp, wp, Sp, n_rays, pulse, mult, g_wp, loss, nullptr, nullptr, nullptr,
                     nr_lidar_sup_t{});
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_interlevel_loss_to_density(const float* c, int c_stride, const float* w, int w_stride, int n_used,
                                             const float* cp, const float* wp, const float* density_p,
                                             const float* euclid_p, int Sp, int64_t n_rays, float pulse, float mult,
                                             float* g_density_p, float* loss, const nr_lidar_sup_t* lidar,
                                             nr_stream_t stream) {
  if (n_rays == 0) return 0;
  nr_lidar_sup_t lid = {};
  if (lidar != nullptr) {
    if (!lidar->is_lidar || !lidar->did_return || !lidar->range) return NR_EINVAL;
    lid = *lidar;
  }
  if (!c || !w || !cp || !wp || !density_p || !euclid_p || !g_density_p || !loss || n_used < 1 || n_used > 31 ||
      w_stride < n_used || c_stride < n_used + 1 || Sp < 1 || Sp > kMaxProp || !(pulse > 0.0f) || n_rays < 0)
    return NR_EINVAL;
  const dim3 grid((unsigned)nr_cdiv(n_rays, kWavesPerBlock));
#define CALL(I)                                                                                                          \
  hipLaunchKernelGGL(interlevel_loss_kernel<I>, grid, dim3(256), 0, nr_s(stream), c, c_stride, w, w_stride, n_used, cp, wp, \
                     Sp, n_rays, pulse, mult, nullptr, loss, density_p, euclid_p, g_density_p, lid)
  if (Sp <= 64) { CALL(1); } else if (Sp <= 128) { CALL(2); } else { CALL(4); }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
