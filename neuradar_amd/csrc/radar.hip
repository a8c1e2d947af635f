// Radar point-set loss of the training step on the device (model_components/radar_utils.py:54-168): the reference builds the
// detection-to-prediction cost matrix with torch ops, copies it to the host, runs scipy's linear_sum_assignment per scan and
// assembles the loss from boolean-mask indexing -- a host round trip in the middle of every step.  Here:
//   radar_cost_kernel   cost[m detections][n predictions] of one scan (+ every detection's cheapest prediction)
//   lsa_kernel          the rectangular linear sum assignment (shortest augmenting paths with dual variables, Crouse 2016 -- the
//                       algorithm scipy implements), ONE workgroup (512 threads) per scan, duals and path state in registers / LDS, float64
//   radar_loss_kernel   Hungarian-matched loss ("euclidean" | "nll") and its gradient w.r.t. the 7 outputs per prediction
// so the whole chain is three launches with no host read and can sit inside a captured graph.
#include "nr_common.h"

namespace {

constexpr float kEps = 1e-6f, kMinVar = 1e-3f, kMaxCost = 1e9f;  // radar_utils.py:30-32
constexpr int kLsaThreads = 512, kLsaWaves = kLsaThreads / NR_WAVE;
constexpr int kLsaMaxCols = 8192, kLsaMaxRows = 1024, kLsaColsPerThread = kLsaMaxCols / kLsaThreads;

__device__ __forceinline__ float clamp_ep(float r) { return fminf(fmaxf(r, kEps), 1.0f - kEps); }

// cost of (prediction k, detection j): "euclidean" |xyz_k - gt_j| - log r_k (radar_utils.py:99-103); "nll"
// log(1 - r_k) - log r_k - sum_axis Laplace(mu, b).log_prob(gt) (:105-118).  inf -> MAX_COST (:120-122).
__device__ __forceinline__ float radar_cost(const float* __restrict__ p, const float* __restrict__ g, int nll) {
  const float r = clamp_ep(p[0]);
  float c;
  if (!nll) {
    const float dx = p[1] - g[0], dy = p[2] - g[1], dz = p[3] - g[2];
    c = sqrtf(dx * dx + dy * dy + dz * dz) - logf(r);
  } else {
    c = logf(1.0f - r) - logf(r);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float b = fmaxf(p[4 + a], kMinVar);
      c += logf(2.0f * b) + fabsf(g[a] - p[1 + a]) / b;
    }
  }
  return (isinf(c) || c != c) ? kMaxCost : c;  // (a NaN prediction -- a diverged fp16 run -- must not poison the search: scipy raises there)
}

// One block per (detection, scan): cost row [n] + its minimum / arg-minimum (the row duals the assignment starts from).
__global__ void __launch_bounds__(256)
radar_cost_kernel(const float* __restrict__ pred, int64_t n, const float* __restrict__ gt, int gt_stride,
                  const int* __restrict__ seg, int m_cap, int nll, float* __restrict__ cost, float* __restrict__ row_min,
                  int* __restrict__ row_arg) {
  const int scan = blockIdx.y, j = blockIdx.x;
  const int m = seg[scan + 1] - seg[scan];
  if (j >= m) return;
  const float* g = gt + (int64_t)(seg[scan] + j) * gt_stride;
  const float g3[3] = {g[0], g[1], g[2]};
  const float* p = pred + (int64_t)scan * n * 7;
  float* out = cost + ((int64_t)scan * m_cap + j) * n;
  float best = INFINITY;
  int arg = 0;
  for (int64_t k = threadIdx.x; k < n; k += blockDim.x) {
    const float c = radar_cost(p + k * 7, g3, nll);
    out[k] = c;
    if (c < best) { best = c; arg = (int)k; }
  }
  // block arg-min (smallest index among equal minima)
  __shared__ float sv[256];
  __shared__ int si[256];
  sv[threadIdx.x] = best;
  si[threadIdx.x] = best < INFINITY ? arg : 0x7fffffff;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float v2 = sv[threadIdx.x + o];
      const int i2 = si[threadIdx.x + o];
      if (v2 < sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    row_min[(int64_t)scan * m_cap + j] = sv[0];
    row_arg[(int64_t)scan * m_cap + j] = si[0] < (int)n ? si[0] : 0;
  }
}

// ---- linear sum assignment ----------------------------------------------------------------------------------------
// Rows = the smaller side (nr <= nc).  Thread t owns columns t, t + 1024, ...: their dual v, shortest-path cost and
// "scanned" flag live in its registers; row4col / path (who reaches the column) in LDS for the augmentation walk.
// Every iteration of the shortest-path search: all threads relax their columns against the current row and the block
// takes the arg-min over (cost, assigned?, column) -- free columns win ties like in scipy, then the lower index.
struct LsaKey {
  double v;
  unsigned k;  // bit 31: column already assigned, low bits: column
};
__device__ __forceinline__ bool lsa_less(const LsaKey& a, const LsaKey& b) { return a.v < b.v || (a.v == b.v && a.k < b.k); }
__device__ __forceinline__ LsaKey lsa_shfl_xor(const LsaKey& a, int o) {
  LsaKey r;
  r.v = __shfl_xor(a.v, o, NR_WAVE);
  r.k = (unsigned)__shfl_xor((int)a.k, o, NR_WAVE);
  return r;
}
// one step of the wave arg-min on DPP operands (VALU only: a ds_bpermute round trip per dword and step made the reduction a
// third of an iteration): lanes without a source keep their own value
template <int CTRL, int ROWMASK>
__device__ __forceinline__ LsaKey lsa_dpp_min(const LsaKey& a) {
  const int hi = __double2hiint(a.v), lo = __double2loint(a.v);
  LsaKey o;
  o.v = __hiloint2double(nr_dpp_i<CTRL, ROWMASK>(hi, hi), nr_dpp_i<CTRL, ROWMASK>(lo, lo));
  o.k = (unsigned)nr_dpp_i<CTRL, ROWMASK>((int)a.k, (int)a.k);
  return lsa_less(o, a) ? o : a;
}
// arg-min over the wave, valid in lane 63 (prefix minima inside the rows of 16, then the two row broadcasts)
__device__ __forceinline__ LsaKey lsa_wave_min_to_lane63(LsaKey a) {
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 1, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 2, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 4, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 8, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_BCAST15, 0xA>(a);
  a = lsa_dpp_min<NR_DPP_ROW_BCAST31, 0xC>(a);
  return a;
}

__global__ void __launch_bounds__(kLsaThreads)
lsa_kernel(const float* __restrict__ cost_all, const float* __restrict__ row_min_all, const int* __restrict__ row_arg_all,
           const int* __restrict__ seg, int m_cap, int n_pred, int* __restrict__ assoc_all, int* __restrict__ status) {
  const int scan = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = seg[scan + 1] - seg[scan];
  const float* C = cost_all + (int64_t)scan * m_cap * n_pred;  // [m][n_pred]
  int* assoc = assoc_all + (int64_t)scan * n_pred;             // per prediction: detection index or -1
  // orientation: rows = detections when m <= n_pred (the usual case: a few hundred detections, thousands of rays)
  const bool tr = m > n_pred;
  const int nr = tr ? n_pred : m, nc = tr ? m : n_pred;
  for (int k = tid; k < n_pred; k += kLsaThreads) assoc[k] = -1;
  if (nr == 0) return;
  if (nr > kLsaMaxRows || nc > kLsaMaxCols) {
    if (tid == 0) status[scan] = 2;  // beyond the kernel's static limits
    return;
  }
  auto c_at = [&](int i, int j) -> double { return (double)(tr ? C[(int64_t)j * n_pred + i] : C[(int64_t)i * n_pred + j]); };

  __shared__ int row4col[kLsaMaxCols];
  __shared__ int path[kLsaMaxCols];
  __shared__ double u[kLsaMaxRows];
  __shared__ int col4row[kLsaMaxRows];
  __shared__ int sr_row[kLsaMaxRows + 1];
  __shared__ double sr_val[kLsaMaxRows + 1];
  __shared__ double red_v[2][kLsaWaves];
  __shared__ unsigned red_k[2][kLsaWaves];

  double v[kLsaColsPerThread], spc[kLsaColsPerThread];
#pragma unroll
  for (int q = 0; q < kLsaColsPerThread; ++q) v[q] = 0.0;
  for (int j = tid; j < nc; j += kLsaThreads) row4col[j] = -1;
  for (int i = tid; i < nr; i += kLsaThreads) {
    col4row[i] = -1;
    u[i] = tr ? 0.0 : (double)row_min_all[(int64_t)scan * m_cap + i];
  }
  __syncthreads();
  if (!tr) {
    // start from the duals u_i = min_j c_ij, v = 0 (feasible; an edge (i, argmin_i) is tight): every column claimed by exactly
    // one row's minimum is assigned at once -- the lowest row wins a contested column -- and only the losers need a search
    for (int i = tid; i < nr; i += kLsaThreads)  // (-1 is the largest unsigned value: any row index replaces it)
      atomicMin(reinterpret_cast<unsigned*>(&row4col[row_arg_all[(int64_t)scan * m_cap + i]]), (unsigned)i);
    __syncthreads();
    for (int j = tid; j < nc; j += kLsaThreads)
      if (row4col[j] != -1) col4row[row4col[j]] = j;
    __syncthreads();
  }

  int parity = 0;
  for (int cur = 0; cur < nr; ++cur) {
    if (col4row[cur] != -1) continue;  // (uniform: LDS value read by every thread after a barrier)
    unsigned scanned = 0;              // bit q: my column q is in SC
#pragma unroll
    for (int q = 0; q < kLsaColsPerThread; ++q) spc[q] = INFINITY;
    double min_val = 0.0;
    int i = cur, n_sr = 0, sink = -1;
    while (sink == -1) {
      if (tid == 0) {
        sr_row[n_sr] = i;
        sr_val[n_sr] = min_val;
      }
      ++n_sr;
      const double ui = u[i];
      LsaKey best = {INFINITY, 0xffffffffu};
#pragma unroll
      for (int q = 0; q < kLsaColsPerThread; ++q) {
        const int j = tid + q * kLsaThreads;
        if (j < nc && !((scanned >> q) & 1u)) {
          const double r = min_val + c_at(i, j) - ui - v[q];
          if (r < spc[q]) {
            spc[q] = r;
            path[j] = i;
          }
          const LsaKey cand = {spc[q], (unsigned)j | (row4col[j] != -1 ? 0x80000000u : 0u)};
          if (lsa_less(cand, best)) best = cand;
        }
      }
      best = lsa_wave_min_to_lane63(best);
      if (lane == NR_WAVE - 1) {
        red_v[parity][wave] = best.v;
        red_k[parity][wave] = best.k;
      }
      __syncthreads();
      best.v = red_v[parity][0];
      best.k = red_k[parity][0];
#pragma unroll
      for (int w = 1; w < kLsaWaves; ++w) {
        const LsaKey other = {red_v[parity][w], red_k[parity][w]};
        if (lsa_less(other, best)) best = other;
      }
      parity ^= 1;
      if (!(best.v < INFINITY)) {  // infeasible (cannot happen with finite costs)
        if (tid == 0) status[scan] = 1;
        return;
      }
      min_val = best.v;
      const int j = (int)(best.k & 0x7fffffffu);
      if ((j & (kLsaThreads - 1)) == tid) scanned |= 1u << (j / kLsaThreads);
      if (best.k & 0x80000000u) i = row4col[j]; else sink = j;
    }
    // dual updates (scipy: u[cur] += minVal; u[i] += minVal - spc[col4row[i]] for the other rows of SR; v[j] -= minVal - spc[j] on SC)
    __syncthreads();  // sr_row / sr_val of the last iteration are visible
    for (int t = tid; t < n_sr; t += kLsaThreads) u[sr_row[t]] += min_val - sr_val[t];
#pragma unroll
    for (int q = 0; q < kLsaColsPerThread; ++q)
      if ((scanned >> q) & 1u) v[q] -= min_val - spc[q];
    // the sink itself was added to SC last with spc = min_val: its v is unchanged (min_val - spc = 0)
    if (tid == 0) {  // augment along the alternating path
      int j = sink;
      while (true) {
        const int i2 = path[j];
        row4col[j] = i2;
        const int prev = col4row[i2];
        col4row[i2] = j;
        j = prev;
        if (i2 == cur) break;
      }
    }
    __syncthreads();
  }
  // association per prediction
  if (!tr) {
    for (int i = tid; i < nr; i += kLsaThreads) assoc[col4row[i]] = i;
  } else {
    for (int i = tid; i < nr; i += kLsaThreads) assoc[i] = col4row[i];
  }
}

// get_radar_loss (radar_utils.py:130-168) + calculate_radar_loss' mean over scans (:93) + radar_mult, value and gradient:
// unmatched prediction: -log(1 - r); matched: -log r + |xyz - gt| ("euclidean") or -log r - sum Laplace log-likelihood ("nll").
__global__ void __launch_bounds__(256)
radar_loss_kernel(const float* __restrict__ pred, int64_t n, int n_scans, const float* __restrict__ gt, int gt_stride,
                  const int* __restrict__ seg, const int* __restrict__ assoc, int nll, float mult, float* __restrict__ g_pred,
                  float* __restrict__ loss) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  if (idx < n * n_scans) {
    const int scan = (int)(idx / n);
    const float* p = pred + idx * 7;
    float* gp = g_pred + idx * 7;
    const float k = mult / ((float)n * (float)n_scans);
    const float raw = p[0], r = clamp_ep(raw);
    const bool live = raw >= kEps && raw <= 1.0f - kEps;  // clamp's gradient
    const int a = assoc[idx];
    float g[7] = {0, 0, 0, 0, 0, 0, 0};
    if (a < 0) {
      acc = -logf(1.0f - r);
      if (live) g[0] = 1.0f / (1.0f - r);
    } else {
      const float* t = gt + (int64_t)(seg[scan] + a) * gt_stride;
      acc = -logf(r);
      if (live) g[0] = -1.0f / r;
      if (!nll) {
        const float dx = p[1] - t[0], dy = p[2] - t[1], dz = p[3] - t[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        acc += d;
        if (d > 0.0f) { g[1] = dx / d; g[2] = dy / d; g[3] = dz / d; }
      } else {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float braw = p[4 + ax], b = fmaxf(braw, kMinVar);
          const float diff = p[1 + ax] - t[ax], ad = fabsf(diff);
          acc += logf(2.0f * b) + ad / b;
          g[1 + ax] = (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f)) / b;
          if (braw >= kMinVar) g[4 + ax] = 1.0f / b - ad / (b * b);
        }
      }
    }
    acc *= k;
#pragma unroll
    for (int q = 0; q < 7; ++q) gp[q] = k * g[q];
  }
  acc = nr_wave_sum(acc);
  if (nr_lane() == 0 && acc != 0.0f) unsafeAtomicAdd(loss + nr_loss_slot_index(), acc);
}

}  // namespace

extern "C" int64_t nr_radar_assign_workspace_bytes(int n_scans, int64_t n_pred, int max_detections) {
  if (n_scans < 0 || n_pred < 0 || max_detections < 0) return -1;
  // cost [scans][m_cap][n] floats, row minima + arg-minima [scans][m_cap], status [scans]
  return ((int64_t)n_scans * max_detections * n_pred + (int64_t)n_scans * max_detections) * 4 + (int64_t)n_scans * max_detections * 4 +
         (int64_t)n_scans * 4 + 64;
}

extern "C" int nr_radar_assign(const float* pred, int n_scans, int64_t n_pred, const float* detections, int det_stride,
                               const int* seg, int max_detections, int cost_type, int* assoc, void* workspace, nr_stream_t stream) {
  if (n_scans == 0 || n_pred == 0) return 0;
  if (!pred || !detections || !seg || !assoc || !workspace || n_scans < 0 || n_pred < 0 || max_detections < 0 || det_stride < 3 ||
      (cost_type != 0 && cost_type != 1) || n_pred > 0x3fffffff)
    return NR_EINVAL;
  const int64_t small = max_detections < n_pred ? max_detections : n_pred, large = max_detections < n_pred ? n_pred : max_detections;
  if (small > kLsaMaxRows || large > kLsaMaxCols) return NR_EINVAL;
  float* cost = reinterpret_cast<float*>(workspace);
  float* row_min = cost + (int64_t)n_scans * max_detections * n_pred;
  int* row_arg = reinterpret_cast<int*>(row_min + (int64_t)n_scans * max_detections);
  int* status = row_arg + (int64_t)n_scans * max_detections;
  hipError_t e = hipMemsetAsync(status, 0, sizeof(int) * n_scans, nr_s(stream));
  if (e != hipSuccess) return (int)e;
  if (max_detections > 0) {
    hipLaunchKernelGGL(radar_cost_kernel, dim3((unsigned)max_detections, (unsigned)n_scans), dim3(256), 0, nr_s(stream), pred, n_pred,
                       detections, det_stride, seg, max_detections, cost_type, cost, row_min, row_arg);
    NR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(lsa_kernel, dim3((unsigned)n_scans), dim3(kLsaThreads), 0, nr_s(stream), cost, row_min, row_arg, seg,
                     max_detections, (int)n_pred, assoc, status);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_radar_loss(const float* pred, int n_scans, int64_t n_pred, const float* detections, int det_stride, const int* seg,
                             const int* assoc, int loss_type, float mult, float* grad_pred, float* loss, nr_stream_t stream) {
  if (n_scans == 0 || n_pred == 0) return 0;
  if (!pred || !detections || !seg || !assoc || !grad_pred || !loss || n_scans < 0 || n_pred < 0 || det_stride < 3 ||
      (loss_type != 0 && loss_type != 1))
    return NR_EINVAL;
  hipLaunchKernelGGL(radar_loss_kernel, dim3((unsigned)nr_cdiv(n_pred * n_scans, 256)), dim3(256), 0, nr_s(stream), pred, n_pred,
                     n_scans, detections, det_stride, seg, assoc, loss_type, mult, grad_pred, loss);
  NR_LAUNCH_CHECK();
  return 0;
}
