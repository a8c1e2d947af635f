// Radar point-set loss of the training step on the device (model_components/radar_utils.py:54-168): the reference builds the
// detection-to-prediction cost matrix with torch ops, copies it to the host, runs scipy's linear_sum_assignment per scan and
// assembles the loss from boolean-mask indexing -- a host round trip in the middle of every step.  Here:
//   radar_cost_sort_kernel  cost[rows][columns] of one scan + every row's cheapest columns in cost order
//   lsa_kernel          the rectangular linear sum assignment (shortest augmenting paths with dual variables, Crouse 2016 -- the
//                       algorithm scipy implements), ONE wave per scan over the assigned columns only, duals and path state in
//                       registers / LDS, float64
//   radar_loss_kernel   Hungarian-matched loss ("euclidean" | "nll") and its gradient w.r.t. the 7 outputs per prediction
// so the whole chain is three launches with no host read and can sit inside a captured graph.
#include "nr_common.h"

namespace {

constexpr float kEps = 1e-6f, kMinVar = 1e-3f, kMaxCost = 1e9f;  // radar_utils.py:30-32
constexpr int kLsaMaxCols = 8192, kLsaMaxRows = 1024;

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

// ---- linear sum assignment ----------------------------------------------------------------------------------------
// Orientation per scan: rows = the smaller side (nr <= nc; detections when there are fewer detections than rays -- the usual
// case).  Workspace per scan: sorted keys [small][small] u64, cost [small][large] float, row-by-slot costs [small][64 Q] float
// (only when they do not fit the LDS); small / large = min / max of (max_detections, n_pred).
//
// The search never needs more than a row's CHEAPEST FREE column: free columns keep the dual v = 0 (v only changes on scanned
// columns, and a free column is scanned only as the sink), so through row i the best free column is the first free entry of the
// row's columns in cost order, and at most nr - 1 columns are ever assigned -- the row's nr cheapest columns are enough.  The
// shortest-path tree is therefore kept over the ASSIGNED columns only ("slots", <= nr of them, numbered as they are assigned)
// plus one running best free candidate: an iteration relaxes <= nr slots instead of nc columns (3 531 -> <= 200 on a
// neuradar radar scan), which one wave does without a barrier.  Same iterations, duals and result as the full-width search.
constexpr int kSortThreads = 512, kSortWaves = kSortThreads / NR_WAVE, kSortPerThread = kLsaMaxCols / kSortThreads;

__device__ __forceinline__ unsigned lsa_ord(float f) {  // float -> unsigned with the same order
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float lsa_unord(unsigned o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu)); }

// One block per (row, scan): the row's costs -> cost[row][:] and its nr cheapest columns in (cost, column) order ->
// sorted[row][:nr].  The nr-th smallest cost is found by bisection on the ordered bit patterns (32 counting rounds over the
// thread's registers), the columns up to it are compacted in column order and only those (P2 = nr rounded up to a power of two,
// <= 1 024) go through a bitonic sort in LDS -- a full sort of the 3 531-column row was LDS-bandwidth-bound at 52 us.
__global__ void __launch_bounds__(kSortThreads)
radar_cost_sort_kernel(const float* __restrict__ pred, int n_pred, const float* __restrict__ gt, int gt_stride,
                       const int* __restrict__ seg, int nll, int small, int large, float* __restrict__ cost_all,
                       unsigned long long* __restrict__ sorted_all) {
  __shared__ unsigned long long keys[kLsaMaxRows];
  __shared__ int part[2][kSortWaves];
  __shared__ int tot[kSortPerThread][kSortWaves];  // per (register slot, wave): taken below the threshold | at it << 16
  const int scan = blockIdx.y, r = blockIdx.x, tid = threadIdx.x, lane = tid & (NR_WAVE - 1), wave = tid / NR_WAVE;
  const int m = seg[scan + 1] - seg[scan];
  const bool tr = m > n_pred;
  const int nr = tr ? n_pred : m, nc = tr ? m : n_pred;
  if (r >= nr || nr > small || nc > large) return;
  const float* p0 = pred + (int64_t)scan * n_pred * 7;
  const float* g0 = gt + (int64_t)seg[scan] * gt_stride;
  float* out = cost_all + ((int64_t)scan * small + r) * large;
  unsigned key[kSortPerThread];  // column c = tid + q * kSortThreads; 0xffffffff past the row (no cost maps there: NaN -> MAX_COST)
#pragma unroll
  for (int q = 0; q < kSortPerThread; ++q) {
    const int c = tid + q * kSortThreads;
    key[q] = 0xffffffffu;
    if (c < nc) {
      const float* g = g0 + (int64_t)(tr ? c : r) * gt_stride;
      const float g3[3] = {g[0], g[1], g[2]};
      const float cst = radar_cost(p0 + (int64_t)(tr ? r : c) * 7, g3, nll);
      out[c] = cst;
      key[q] = lsa_ord(cst);
    }
  }
  // the smallest T with #(key <= T) >= nr
  unsigned lo = 0u, hi = 0xfffffffeu;
  for (int it = 0; lo < hi; ++it) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kSortPerThread; ++q) cnt += key[q] <= mid ? 1 : 0;
    cnt = (int)nr_wave_sum((float)cnt);  // (<= 8 192: exact in float)
    if (lane == 0) part[it & 1][wave] = cnt;
    __syncthreads();
    int all = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) all += part[it & 1][w];
    if (all >= nr) hi = mid; else lo = mid + 1;
  }
  const unsigned T = lo;
  // ordered compaction (column order = register slot, then thread): everything below T, then the first columns AT T
#pragma unroll
  for (int q = 0; q < kSortPerThread; ++q) {
    const unsigned long long below = __ballot(key[q] < T), at = __ballot(key[q] == T);
    if (lane == 0) tot[q][wave] = __popcll(below) | (__popcll(at) << 16);
  }
  __syncthreads();
  int n_below = 0;
#pragma unroll
  for (int q = 0; q < kSortPerThread; ++q)
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) n_below += tot[q][w] & 0xffff;
  const int need_at = nr - n_below;  // >= 1
  int P2 = 2;
  while (P2 < nr) P2 <<= 1;
  for (int k = nr + tid; k < P2; k += kSortThreads) keys[k] = ~0ull;
  int base_below = 0, base_at = 0;
#pragma unroll
  for (int q = 0; q < kSortPerThread; ++q) {
    const unsigned long long below = __ballot(key[q] < T), at = __ballot(key[q] == T);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int b0 = base_below, a0 = base_at;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const int t = tot[q][w];
      if (w < wave) {
        b0 += t & 0xffff;
        a0 += t >> 16;
      }
      base_below += t & 0xffff;
      base_at += t >> 16;
    }
    const unsigned long long full = ((unsigned long long)key[q] << 32) | (unsigned)(tid + q * kSortThreads);
    if (key[q] < T) {
      keys[b0 + __popcll(below & lt)] = full;
    } else if (key[q] == T) {
      const int rank = a0 + __popcll(at & lt);
      if (rank < need_at) keys[n_below + rank] = full;
    }
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P2 >> 1); t += kSortThreads) {
        const int l2 = ((t & ~(j - 1)) << 1) | (t & (j - 1)), h2 = l2 | j;
        const unsigned long long a = keys[l2], b = keys[h2];
        if ((a > b) == ((l2 & k) == 0)) {
          keys[l2] = b;
          keys[h2] = a;
        }
      }
      __syncthreads();
    }
  }
  unsigned long long* dst = sorted_all + ((int64_t)scan * small + r) * small;
  for (int k = tid; k < nr; k += kSortThreads) dst[k] = keys[k];
}

// minimum of a double over the wave, in every lane (DPP operands: VALU only)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double lsa_dpp_min(double a) {
  const int hi = __double2hiint(a), lo = __double2loint(a);
  return fmin(a, __hiloint2double(nr_dpp_i<CTRL, ROWMASK>(hi, hi), nr_dpp_i<CTRL, ROWMASK>(lo, lo)));
}
__device__ __forceinline__ double lsa_wave_min(double a) {
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 1, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 2, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 4, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_SHR + 8, 0xF>(a);
  a = lsa_dpp_min<NR_DPP_ROW_BCAST15, 0xA>(a);
  a = lsa_dpp_min<NR_DPP_ROW_BCAST31, 0xC>(a);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 63), __builtin_amdgcn_readlane(__double2loint(a), 63));
}

// Per-slot state lives in registers: entry x of such an array belongs to lane (x & 63), register (x >> 6).
// A lane's own entry by a per-lane register index / predicated write of it (VALU selects: no branches):
template <int Q, typename T>
__device__ __forceinline__ T lsa_pick(const T (&a)[Q], int q) {
  // (every element passes through an empty asm: left alone, the compiler folds the select chain over the array into ONE
  // dynamically indexed load, which puts the array -- all the per-slot state -- into scratch memory: 2 100 cycles per iteration)
  T r = a[0];
  asm("" : "+v"(r));
#pragma unroll
  for (int k = 1; k < Q; ++k) {
    T t = a[k];
    asm("" : "+v"(t));
    r = q == k ? t : r;
  }
  return r;
}
template <int Q, typename T>
__device__ __forceinline__ void lsa_put(T (&a)[Q], int q, T val, bool pred) {
#pragma unroll
  for (int k = 0; k < Q; ++k) a[k] = (pred && q == k) ? val : a[k];
}
__device__ __forceinline__ int lsa_lane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double lsa_lane(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// read of entry x for a wave-uniform x
template <int Q, typename T>
__device__ __forceinline__ T lsa_rd(const T (&a)[Q], int x) { return lsa_lane(lsa_pick<Q, T>(a, x >> 6), x & 63); }

// One wave per scan (shortest augmenting paths with dual variables, Crouse 2016 -- the algorithm scipy implements; float64
// duals).  All state of the search is per SLOT and in registers: the slot's column and dual v, and -- moving with it along an
// augmenting path -- the row matched to it with that row's dual u, its cached cheapest free column (and where in its sorted
// columns that was found) and two flags; an unmatched row exists only as the root of its own search (wave-uniform values).
// cm[row][slot]: the cost of (row, the slot's column) -- in LDS when small * small floats fit beside the assigned-columns
// bitmap (<= 200 rows), else in the workspace through L2-coherent accesses (a plain store followed by a plain load of the same
// address by the same wave returned stale L1 data on gfx950) -- filled for a row when it first enters a tree and extended for
// all such rows when a column becomes a slot: an iteration of the search touches no global memory and is ~150 instructions.
template <int Q, bool LDS_CM>
__global__ void __launch_bounds__(NR_WAVE)
lsa_kernel(const float* __restrict__ cost_all, const unsigned long long* __restrict__ sorted_all, float* cm_all,
           const int* __restrict__ seg, int small, int large, int n_pred, int* __restrict__ assoc_all, int* __restrict__ status) {
  extern __shared__ unsigned lsa_lds[];
  unsigned* abits = lsa_lds;                                           // [kLsaMaxCols / 32] column is assigned
  int* claim = reinterpret_cast<int*>(lsa_lds + kLsaMaxCols / 32);     // start only: [nc] lowest row whose minimum it is
  int* start_col = claim + kLsaMaxCols;                                // start only: [Q * 64] per slot: column, row, minimum
  int* start_row = start_col + Q * NR_WAVE;
  int* start_min = start_row + Q * NR_WAVE;
  float* cm_l = reinterpret_cast<float*>(lsa_lds + kLsaMaxCols / 32);  // afterwards (LDS_CM): [small][small]
  const int scan = blockIdx.x, lane = threadIdx.x;
  // ONE wave per scan on the step's critical chain, sharing its SIMD with whatever else the step has in flight: highest issue priority
  __builtin_amdgcn_s_setprio(3);
  const int m = seg[scan + 1] - seg[scan];
  int* assoc = assoc_all + (int64_t)scan * n_pred;  // per prediction: detection index or -1
  const bool tr = m > n_pred;
  const int nr = tr ? n_pred : m, nc = tr ? m : n_pred;
  for (int k = lane; k < n_pred; k += NR_WAVE) assoc[k] = -1;
  if (nr == 0) return;
  if (nr > Q * NR_WAVE || nc > kLsaMaxCols || nr > small || nc > large) {
    if (lane == 0) status[scan] = 2;  // beyond the kernel's static limits / more detections than max_detections
    return;
  }
  const float* C = cost_all + (int64_t)scan * small * large;
  const unsigned long long* S = sorted_all + (int64_t)scan * small * small;
  float* cm_g = cm_all + (int64_t)scan * small * (Q * NR_WAVE);
  auto cm_at = [&](int row, int s) -> float* { return LDS_CM ? cm_l + row * small + s : cm_g + (int64_t)row * (Q * NR_WAVE) + s; };
  auto cm_store = [&](int row, int s, float x) {
    if (LDS_CM) *cm_at(row, s) = x; else __hip_atomic_store(cm_at(row, s), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto cm_load = [&](int row, int s) -> float {
    return LDS_CM ? *cm_at(row, s) : __hip_atomic_load(cm_at(row, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto is_assigned = [&](unsigned c) -> bool { return (abits[c >> 5] >> (c & 31)) & 1u; };
  constexpr int kActive = 1, kHasFree = 2;  // flags: the row has its cm row | its cheapest free column is cached

  for (int w = lane; w < kLsaMaxCols / 32; w += NR_WAVE) abits[w] = 0u;
  for (int j = lane; j < nc; j += NR_WAVE) claim[j] = -1;
  __syncthreads();
  // start from the duals u_i = min_j c_ij, v = 0 (feasible; the edge (i, argmin_i) is tight): every column that is some row's
  // minimum is assigned at once -- the lowest row wins a contested column -- and only the losers need a search
  int c0[Q], min0[Q];  // per row (lane + 64 q): its cheapest column and that cost's ordered bits
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = lane + q * NR_WAVE;
    c0[q] = min0[q] = 0;
    if (i < nr) {
      const unsigned long long k0 = S[(int64_t)i * small];
      min0[q] = (int)(unsigned)(k0 >> 32);
      c0[q] = (int)(unsigned)k0;
      atomicMin(reinterpret_cast<unsigned*>(&claim[c0[q]]), (unsigned)i);  // (-1 is the largest unsigned value)
    }
  }
  __syncthreads();
  int n_slots = 0;
  unsigned unmatched = 0;  // bit q: my row q needs a search
#pragma unroll
  for (int q = 0; q < Q; ++q) {  // (slots are numbered in row order)
    const int i = lane + q * NR_WAVE;
    const bool win = i < nr && claim[c0[q]] == i;
    const unsigned long long mask = __ballot(win);
    if (win) {
      const int s = n_slots + __popcll(mask & ((1ull << lane) - 1ull));
      start_col[s] = c0[q];
      start_row[s] = i;
      start_min[s] = min0[q];
      atomicOr(&abits[(unsigned)c0[q] >> 5], 1u << (c0[q] & 31));
    } else if (i < nr) {
      unmatched |= 1u << q;
    }
    n_slots += __popcll(mask);
  }
  __syncthreads();
  // slots (entry s: lane s & 63, register s >> 6)
  double u_s[Q], v[Q], vw[Q], spc[Q], cand[Q];  // u of the slot's row; v of its column; per search: v or -inf (closed), path cost, open path cost
  int col[Q], row_of[Q], path[Q], bf_cost[Q], bf_col[Q], bf_ptr[Q], flags[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int s = lane + q * NR_WAVE;
    col[q] = s < n_slots ? start_col[s] : 0;
    row_of[q] = s < n_slots ? start_row[s] : 0;
    u_s[q] = s < n_slots ? (double)lsa_unord((unsigned)start_min[s]) : 0.0;
    path[q] = bf_cost[q] = bf_col[q] = bf_ptr[q] = flags[q] = 0;
    v[q] = 0.0;
  }
  __syncthreads();  // (the start arrays' LDS becomes cm)

  for (int cur = 0; cur < nr; ++cur) {
    if (!((lsa_lane((int)unmatched, cur & 63) >> (cur >> 6)) & 1)) continue;
    // the root: an unmatched row (never in a tree before)
    double cur_u = (double)lsa_unord((unsigned)lsa_rd<Q, int>(min0, cur));
    int cur_bf_cost = 0, cur_bf_col = 0, cur_bf_ptr = 0, cur_flags = 0;
    unsigned scanned = 0;  // bit q: my slot q is in the tree
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      vw[q] = (lane + q * NR_WAVE) < n_slots ? v[q] : -INFINITY;
      spc[q] = cand[q] = INFINITY;
    }
    double min_val = 0.0, free_val = INFINITY;  // free_val: cheapest way to a free column so far ...
    unsigned free_col = 0xffffffffu;            // ... the column ...
    int free_via = -1;                          // ... and the slot of the row it is reached from (-1: the root)
    // the row being scanned (wave-uniform): its slot (-1: the root), dual, cached free column, flags
    int i = cur, i_slot = -1, r_bf_cost = 0, r_bf_col = 0, r_flags = 0;
    double ui = cur_u;
    while (true) {
      const bool mine = i_slot >= 0 && lane == (i_slot & 63);
      if (!(r_flags & kActive)) {  // first time in a tree: the row's costs of all slots so far
        const float* Ci = C + (int64_t)i * large;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int s = lane + q * NR_WAVE;
          if (s < n_slots) cm_store(i, s, Ci[col[q]]);
        }
        r_flags |= kActive;
        if (i_slot < 0) cur_flags = r_flags; else lsa_put<Q, int>(flags, i_slot >> 6, r_flags, mine);
      }
      float cq[Q];  // (slots past n_slots: any value -- their vw is -inf)
#pragma unroll
      for (int q = 0; q < Q; ++q) cq[q] = cm_load(i, LDS_CM ? min(lane + q * NR_WAVE, small - 1) : lane + q * NR_WAVE);
      if (!(r_flags & kHasFree)) {  // the first free entry of the row's sorted columns
        const unsigned long long* Si = S + (int64_t)i * small;
        int p = 0;
        while (true) {
          const unsigned long long key = (p + lane) < nr ? Si[p + lane] : ~0ull;
          const bool is_free = (p + lane) < nr && !is_assigned((unsigned)key);
          const unsigned long long mask = __ballot(is_free);
          if (mask) {
            const int first = __builtin_ctzll(mask);
            r_bf_cost = lsa_lane((int)(key >> 32), first);
            r_bf_col = lsa_lane((int)(unsigned)key, first);
            p += first;
            break;
          }
          p += NR_WAVE;
          if (p >= nr) {  // cannot happen: fewer than nr columns are assigned
            if (lane == 0) status[scan] = 1;
            return;
          }
        }
        r_flags |= kHasFree;
        if (i_slot < 0) {
          cur_bf_cost = r_bf_cost;
          cur_bf_col = r_bf_col;
          cur_bf_ptr = p;
        } else {
          lsa_put<Q, int>(bf_cost, i_slot >> 6, r_bf_cost, mine);
          lsa_put<Q, int>(bf_col, i_slot >> 6, r_bf_col, mine);
          lsa_put<Q, int>(bf_ptr, i_slot >> 6, p, mine);
        }
        if (i_slot < 0) cur_flags = r_flags; else lsa_put<Q, int>(flags, i_slot >> 6, r_flags, mine);
      }
      {
        const double r = min_val + (double)lsa_unord((unsigned)r_bf_cost) - ui;
        if (r < free_val || (r == free_val && (unsigned)r_bf_col < free_col)) {
          free_val = r;
          free_col = (unsigned)r_bf_col;
          free_via = i_slot;
        }
      }
      double best = INFINITY;
      int best_q = 0;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const double r = min_val + (double)cq[q] - ui - vw[q];  // (+inf on closed slots)
        const bool upd = r < spc[q];
        spc[q] = upd ? r : spc[q];
        cand[q] = upd ? r : cand[q];
        path[q] = upd ? i_slot : path[q];
        if (cand[q] < best) {
          best = cand[q];
          best_q = q;
        }
      }
      const double wmin = lsa_wave_min(best);
      if (!(wmin < free_val)) {  // a free column is at least as near as every open assigned one: the sink (free wins ties)
        min_val = free_val;
        break;
      }
      min_val = wmin;
      // the owner of the minimum: the lowest lane holding it (its lowest slot); its row is scanned next
      const int owner = __builtin_ctzll(__ballot(best == wmin));
      const int sel_row = lsa_pick<Q, int>(row_of, best_q), sel_cost = lsa_pick<Q, int>(bf_cost, best_q),
                sel_col = lsa_pick<Q, int>(bf_col, best_q), sel_flags = lsa_pick<Q, int>(flags, best_q);
      const double sel_u = lsa_pick<Q, double>(u_s, best_q);
      const bool own = lane == owner;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const bool hit = own && q == best_q;
        cand[q] = hit ? INFINITY : cand[q];
        vw[q] = hit ? -INFINITY : vw[q];
      }
      scanned |= own ? (1u << best_q) : 0u;
      i_slot = owner + NR_WAVE * lsa_lane(best_q, owner);
      i = lsa_lane(sel_row, owner);
      ui = lsa_lane(sel_u, owner);
      r_bf_cost = lsa_lane(sel_cost, owner);
      r_bf_col = lsa_lane(sel_col, owner);
      r_flags = lsa_lane(sel_flags, owner);
    }
    if (!(min_val < INFINITY)) {  // infeasible (cannot happen with finite costs)
      if (lane == 0) status[scan] = 1;
      return;
    }
    // dual updates (scipy: u[cur] += minVal; u[i] += minVal - spc[col4row[i]] for the other rows of SR; v[j] -= minVal - spc[j] on SC)
    const int s_new = n_slots;
    const unsigned sink = free_col;
    cur_u += min_val;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if ((scanned >> q) & 1u) {
        const double d = min_val - spc[q];
        u_s[q] += d;
        v[q] -= d;
      }
      if (lane + q * NR_WAVE == s_new) {  // the sink becomes a slot (its v stays 0)
        col[q] = (int)sink;
        v[q] = 0.0;
      }
    }
    if (lane == 0) abits[sink >> 5] |= 1u << (sink & 31);
    {  // augment: along the alternating path every row moves to the slot it reached (the root ends in the last one)
      int js = s_new, src = free_via;
      while (true) {
        int m_row = cur, m_cost = cur_bf_cost, m_col = cur_bf_col, m_ptr = cur_bf_ptr, m_flags = cur_flags;
        double m_u = cur_u;
        if (src >= 0) {
          m_row = lsa_rd<Q, int>(row_of, src);
          m_cost = lsa_rd<Q, int>(bf_cost, src);
          m_col = lsa_rd<Q, int>(bf_col, src);
          m_ptr = lsa_rd<Q, int>(bf_ptr, src);
          m_flags = lsa_rd<Q, int>(flags, src);
          m_u = lsa_rd<Q, double>(u_s, src);
        }
        const bool dst = lane == (js & 63);
        lsa_put<Q, int>(row_of, js >> 6, m_row, dst);
        lsa_put<Q, int>(bf_cost, js >> 6, m_cost, dst);
        lsa_put<Q, int>(bf_col, js >> 6, m_col, dst);
        lsa_put<Q, int>(bf_ptr, js >> 6, m_ptr, dst);
        lsa_put<Q, int>(flags, js >> 6, m_flags, dst);
        lsa_put<Q, double>(u_s, js >> 6, m_u, dst);
        if (src < 0) break;
        js = src;
        src = lsa_rd<Q, int>(path, src);
      }
    }
    ++n_slots;
    __syncthreads();  // (the bitmap's new bit)
    // rows that have been in a tree: their cost of the new slot; rows whose cached cheapest free column was the sink: the next one
    // (both loads requested before either is used)
    float c_new[Q];
    unsigned long long k_new[Q];
    bool refresh[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool live = (lane + q * NR_WAVE) < n_slots;
      c_new[q] = 0.0f;
      k_new[q] = 0ull;
      refresh[q] = live && (flags[q] & kHasFree) && (unsigned)bf_col[q] == sink;
      if (live && (flags[q] & kActive)) c_new[q] = C[(int64_t)row_of[q] * large + sink];
      if (refresh[q]) k_new[q] = S[(int64_t)row_of[q] * small + bf_ptr[q] + 1];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool live = (lane + q * NR_WAVE) < n_slots;
      if (live && (flags[q] & kActive)) cm_store(row_of[q], s_new, c_new[q]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (refresh[q]) {
        const unsigned long long* Sr = S + (int64_t)row_of[q] * small;
        int p = bf_ptr[q] + 1;
        unsigned long long key = k_new[q];
        while (is_assigned((unsigned)key)) key = Sr[++p];  // (ends before nr: fewer than nr columns are assigned)
        bf_ptr[q] = p;
        bf_cost[q] = (int)(key >> 32);
        bf_col[q] = (int)(unsigned)key;
      }
    }
    __syncthreads();
  }
  // association per prediction
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if ((lane + q * NR_WAVE) < n_slots) {
      if (!tr) assoc[col[q]] = row_of[q]; else assoc[row_of[q]] = col[q];
    }
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


// Radar rays -> rendered points + their sine position embedding (neuradar.py:470-476, detr/models/position_encoding_3d.py:56-103)
// in ONE launch (the torch expression of it is ~50 elementwise launches at the head of the radar chain).  Channel c of the
// embedding: axis code[c] >> 1, sin (code & 1 == 0) or cos of  xyz[axis] * 2 pi / dim_t[c]  (dim_t from the host, in torch's
// arithmetic).  dirs = d xyz / d depth.
__global__ void __launch_bounds__(256)
radar_points_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ sph, int64_t n, const float* __restrict__ dim_t,
                        const int* __restrict__ code, int C, float* __restrict__ xyz, float* __restrict__ dirs, float* __restrict__ pos) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const int64_t r = idx / C;
  const int c = (int)(idx - r * C);
  const float d = depth[r], phi = sph[r * 2], theta = sph[r * 2 + 1];
  const float cp = cosf(phi), sp = sinf(phi), ct = cosf(theta), st = sinf(theta);
  const float p3[3] = {d * cp * ct, d * sp * ct, d * st};
  if (c < 3) {
    xyz[r * 3 + c] = p3[c];
    dirs[r * 3 + c] = c == 0 ? cp * ct : (c == 1 ? sp * ct : st);
  }
  const int k = code[c];
  const float a = p3[k >> 1] * 6.283185307179586f / dim_t[c];
  pos[idx] = (k & 1) ? cosf(a) : sinf(a);
}

__global__ void __launch_bounds__(256)
radar_points_bwd_kernel(const float* __restrict__ g_xyz, const float* __restrict__ dirs, int64_t n, float* __restrict__ g_depth) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  g_depth[r] = g_xyz[r * 3] * dirs[r * 3] + g_xyz[r * 3 + 1] * dirs[r * 3 + 1] + g_xyz[r * 3 + 2] * dirs[r * 3 + 2];
}


// ---- the three radar heads + the assembly of radar_output in one launch each way (models/neuradar.py:252-278,480-491) --------
// offset / existence / uncertainty head: MLP in -> 16 -> 16 -> {3, 1, 3}, ReLU hidden (field_components/mlp.py), then
//   radar_output = [sigmoid(e), xyz + 1.5 tanh(o), softplus(u)]                                   (7 floats per ray).
// torch: three MLP launches + tanh, scale, sigmoid, softplus, add, cat forward; three MLP backward launches (a fixed ~40 us
// each at a few thousand rays) + the activations' backwards and the slicing of the 7-wide gradient.  Block = 64 rays x 3 heads
// (wave h = head h, lane = ray; a fourth wave helps with the parameter gradients); parameters in LDS.
constexpr int kRhHid = 16, kRhRays = 64, kRhThreads = 256, kRhMaxIn = 64;
__device__ __constant__ const int kRhOut[3] = {3, 1, 3};
struct RhLayout {  // float offsets of one head's parameters inside the block's LDS image
  int w1, b1, w2, b2, w3, b3, size;
};
__host__ __device__ inline RhLayout rh_layout(int C, int out) {
  RhLayout l;
  l.w1 = 0;
  l.b1 = l.w1 + kRhHid * C;
  l.w2 = l.b1 + kRhHid;
  l.b2 = l.w2 + kRhHid * kRhHid;
  l.w3 = l.b2 + kRhHid;
  l.b3 = l.w3 + out * kRhHid;
  l.size = l.b3 + out;
  return l;
}
__device__ __forceinline__ int rh_head_base(int C, int h) {
  int base = 0;
  for (int k = 0; k < h; ++k) base += rh_layout(C, kRhOut[k]).size;
  return base;
}
__device__ __forceinline__ void rh_load_params(const nr_radar_heads_t& hd, int C, float* __restrict__ lw) {
  for (int h = 0; h < 3; ++h) {
    const RhLayout l = rh_layout(C, kRhOut[h]);
    float* dst = lw + rh_head_base(C, h);
    for (int k = threadIdx.x; k < kRhHid * C; k += blockDim.x) dst[l.w1 + k] = hd.weight[h][0][k];
    for (int k = threadIdx.x; k < kRhHid * kRhHid; k += blockDim.x) dst[l.w2 + k] = hd.weight[h][1][k];
    for (int k = threadIdx.x; k < kRhOut[h] * kRhHid; k += blockDim.x) dst[l.w3 + k] = hd.weight[h][2][k];
    for (int k = threadIdx.x; k < kRhHid; k += blockDim.x) {
      dst[l.b1 + k] = hd.bias[h][0][k];
      dst[l.b2 + k] = hd.bias[h][1][k];
    }
    for (int k = threadIdx.x; k < kRhOut[h]; k += blockDim.x) dst[l.b3 + k] = hd.bias[h][2][k];
  }
}
// one head of one ray: hidden activations h1, h2 and the raw outputs y (x: the ray's C inputs, read from LDS row xs)
__device__ __forceinline__ void rh_head_fwd(const float* __restrict__ lw, const RhLayout& l, int C, int out, const float* __restrict__ xs,
                                            float (&h1)[kRhHid], float (&h2)[kRhHid], float (&y)[3]) {
#pragma unroll
  for (int j = 0; j < kRhHid; ++j) h1[j] = lw[l.b1 + j];
  for (int k = 0; k < C; ++k) {
    const float xv = xs[k];
#pragma unroll
    for (int j = 0; j < kRhHid; ++j) h1[j] += lw[l.w1 + j * C + k] * xv;
  }
#pragma unroll
  for (int j = 0; j < kRhHid; ++j) h1[j] = fmaxf(h1[j], 0.0f);
#pragma unroll
  for (int j = 0; j < kRhHid; ++j) {
    float a = lw[l.b2 + j];
#pragma unroll
    for (int k = 0; k < kRhHid; ++k) a += lw[l.w2 + j * kRhHid + k] * h1[k];
    h2[j] = fmaxf(a, 0.0f);
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    float a = 0.0f;
    if (o < out) {
      a = lw[l.b3 + o];
#pragma unroll
      for (int k = 0; k < kRhHid; ++k) a += lw[l.w3 + o * kRhHid + k] * h2[k];
    }
    y[o] = a;
  }
}
__device__ __forceinline__ float rh_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float rh_softplus(float v) { return v > 20.0f ? v : log1pf(expf(v)); }  // torch.nn.Softplus(beta=1, threshold=20)

__global__ void __launch_bounds__(kRhThreads)
radar_heads_fwd_kernel(nr_radar_heads_t hd, const float* __restrict__ x, int C, const float* __restrict__ xyz, int64_t n,
                       float* __restrict__ out) {
  extern __shared__ float rh_lds[];
  float* lw = rh_lds;                                                         // the three heads' parameters
  float* xs = lw + rh_head_base(C, 3);                                        // [64 rays][C + 1]
  rh_load_params(hd, C, lw);
  const int64_t r0 = (int64_t)blockIdx.x * kRhRays;
  for (int k = threadIdx.x; k < kRhRays * C; k += blockDim.x) {
    const int t = k / C, c = k - t * C;
    xs[t * (C + 1) + c] = (r0 + t) < n ? x[(r0 + t) * C + c] : 0.0f;
  }
  __syncthreads();
  const int h = threadIdx.x >> 6, t = threadIdx.x & 63;
  const int64_t r = r0 + t;
  if (h >= 3 || r >= n) return;
  const RhLayout l = rh_layout(C, kRhOut[h]);
  float h1[kRhHid], h2[kRhHid], y[3];
  rh_head_fwd(lw + rh_head_base(C, h), l, C, kRhOut[h], xs + t * (C + 1), h1, h2, y);
  float* o = out + r * 7;
  if (h == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) o[1 + a] = xyz[r * 3 + a] + 1.5f * tanhf(y[a]);
  } else if (h == 1) {
    o[0] = rh_sigmoid(y[0]);
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) o[4 + a] = rh_softplus(y[a]);
  }
}

// Backward: wave h recomputes its head for its ray, backpropagates, leaves (d y, d h2, d h1) and (h2, h1) in LDS; then all four
// waves sum the outer products over the block's 64 rays -- every thread owns parameter entries -- and add them to the gradients.
__global__ void __launch_bounds__(kRhThreads)
radar_heads_bwd_kernel(nr_radar_heads_t hd, const float* __restrict__ x, int C, const float* __restrict__ g_out, int64_t n,
                       float* __restrict__ g_x, float* __restrict__ g_xyz, nr_radar_heads_grads_t gr) {
  extern __shared__ float rh_lds[];
  float* lw = rh_lds;
  float* xs = lw + rh_head_base(C, 3);                  // [64][C + 1]
  float* act = xs + kRhRays * (C + 1);                  // [3 heads][64 rays][h1 16 | h2 16 | dh1 16 | dh2 16 | dy 3 | pad 1] = 68 floats
  float* gxs = act + 3 * kRhRays * 68;                  // [3 heads][64 rays][C + 1]: every head's d x
  rh_load_params(hd, C, lw);
  const int64_t r0 = (int64_t)blockIdx.x * kRhRays;
  for (int k = threadIdx.x; k < kRhRays * C; k += blockDim.x) {
    const int t = k / C, c = k - t * C;
    xs[t * (C + 1) + c] = (r0 + t) < n ? x[(r0 + t) * C + c] : 0.0f;
  }
  __syncthreads();
  const int h = threadIdx.x >> 6, t = threadIdx.x & 63;
  const int64_t r = r0 + t;
  if (h < 3) {
    const int out = kRhOut[h];
    const RhLayout l = rh_layout(C, out);
    const float* w = lw + rh_head_base(C, h);
    float* a = act + (h * kRhRays + t) * 68;
    float* gx = gxs + (h * kRhRays + t) * (C + 1);
    float h1[kRhHid], h2[kRhHid], y[3], dy[3] = {0.0f, 0.0f, 0.0f}, d2[kRhHid], d1[kRhHid];
    const bool live = r < n;
    if (live) {
      rh_head_fwd(w, l, C, out, xs + t * (C + 1), h1, h2, y);
      const float* g = g_out + r * 7;
      if (h == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float th = tanhf(y[q]);
          dy[q] = 1.5f * (1.0f - th * th) * g[1 + q];
          g_xyz[r * 3 + q] = g[1 + q];
        }
      } else if (h == 1) {
        const float e = rh_sigmoid(y[0]);
        dy[0] = e * (1.0f - e) * g[0];
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) dy[q] = (y[q] > 20.0f ? 1.0f : rh_sigmoid(y[q])) * g[4 + q];
      }
#pragma unroll
      for (int k = 0; k < kRhHid; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int o = 0; o < 3; ++o)
          if (o < out) s += w[l.w3 + o * kRhHid + k] * dy[o];
        d2[k] = h2[k] > 0.0f ? s : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < kRhHid; ++k) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < kRhHid; ++j) s += w[l.w2 + j * kRhHid + k] * d2[j];
        d1[k] = h1[k] > 0.0f ? s : 0.0f;
      }
      for (int c = 0; c < C; ++c) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < kRhHid; ++j) s += w[l.w1 + j * C + c] * d1[j];
        gx[c] = s;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kRhHid; ++k) h1[k] = h2[k] = d1[k] = d2[k] = 0.0f;
      for (int c = 0; c < C; ++c) gx[c] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kRhHid; ++k) {
      a[k] = h1[k];
      a[16 + k] = h2[k];
      a[32 + k] = d1[k];
      a[48 + k] = d2[k];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) a[64 + q] = dy[q];
  }
  __syncthreads();
  // d x = sum over the heads
  for (int k = threadIdx.x; k < kRhRays * C; k += blockDim.x) {
    const int tt = k / C, c = k - tt * C;
    if (r0 + tt < n)
      g_x[(r0 + tt) * C + c] = gxs[(0 * kRhRays + tt) * (C + 1) + c] + gxs[(1 * kRhRays + tt) * (C + 1) + c] + gxs[(2 * kRhRays + tt) * (C + 1) + c];
  }
  // parameter gradients: entry e of a head's image = sum over the 64 rays of (left factor) * (right factor)
  for (int hh = 0; hh < 3; ++hh) {
    const int out = kRhOut[hh];
    const RhLayout l = rh_layout(C, out);
    const float* a0 = act + hh * kRhRays * 68;
    for (int e = threadIdx.x; e < l.size; e += blockDim.x) {
      int lo, ro;      // offsets of the left (gradient) and right (activation) factor inside a ray's 68 floats; ro < 0: bias;
      bool from_x = false;  // right factor = the ray's input x
      int xc = 0;
      float* dst;
      if (e < l.b1) { const int j = e / C; xc = e - j * C; lo = 32 + j; ro = 0; from_x = true; dst = gr.weight[hh][0] + e; }
      else if (e < l.w2) { lo = 32 + (e - l.b1); ro = -1; dst = gr.bias[hh][0] + (e - l.b1); }
      else if (e < l.b2) { const int q = e - l.w2, j = q / kRhHid, k2 = q - j * kRhHid; lo = 48 + j; ro = k2; dst = gr.weight[hh][1] + q; }
      else if (e < l.w3) { lo = 48 + (e - l.b2); ro = -1; dst = gr.bias[hh][1] + (e - l.b2); }
      else if (e < l.b3) { const int q = e - l.w3, o = q / kRhHid, k2 = q - o * kRhHid; lo = 64 + o; ro = 16 + k2; dst = gr.weight[hh][2] + q; }
      else { lo = 64 + (e - l.b3); ro = -1; dst = gr.bias[hh][2] + (e - l.b3); }
      float s = 0.0f;
      for (int tt = 0; tt < kRhRays; ++tt) {
        const float left = a0[tt * 68 + lo];
        const float right = from_x ? xs[tt * (C + 1) + xc] : (ro < 0 ? 1.0f : a0[tt * 68 + ro]);
        s += left * right;
      }
      if (s != 0.0f) unsafeAtomicAdd(dst, s);
    }
  }
}

}  // namespace

static int lsa_q(int64_t small) {  // slots per lane: 1, 2, 4, 8, 16
  int q = 1;
  while ((int64_t)q * NR_WAVE < small) q <<= 1;
  return q;
}
constexpr int64_t kLsaLdsBytes = 160 * 1024, kLsaLdsFixed = kLsaMaxCols / 8;  // the CU's LDS; the assigned-columns bitmap
inline bool lsa_cm_in_lds(int64_t small) { return kLsaLdsFixed + small * small * 4 <= kLsaLdsBytes; }
static int64_t lsa_lds_bytes(int64_t small, int Q) {
  const int64_t start = (int64_t)kLsaMaxCols * 4 + 3 * Q * NR_WAVE * 4, cm = lsa_cm_in_lds(small) ? small * small * 4 : 0;
  return kLsaLdsFixed + (start > cm ? start : cm);
}

extern "C" int64_t nr_radar_assign_workspace_bytes(int n_scans, int64_t n_pred, int max_detections) {
  if (n_scans < 0 || n_pred < 0 || max_detections < 0) return -1;
  const int64_t small = max_detections < n_pred ? max_detections : n_pred, large = max_detections < n_pred ? n_pred : max_detections;
  // per scan: sorted keys [small][small] u64, cost [small][large] float, row-by-slot costs [small][64 Q] float; status [scans]
  return (int64_t)n_scans * small * small * 8 + (int64_t)n_scans * small * (large + lsa_q(small) * NR_WAVE) * 4 + (int64_t)n_scans * 4 + 64;
}

extern "C" int64_t nr_radar_assign_status_offset(int n_scans, int64_t n_pred, int max_detections) {
  if (n_scans < 0 || n_pred < 0 || max_detections < 0) return -1;
  const int64_t small = max_detections < n_pred ? max_detections : n_pred, large = max_detections < n_pred ? n_pred : max_detections;
  return (int64_t)n_scans * small * small * 8 + (int64_t)n_scans * small * (large + lsa_q(small) * NR_WAVE) * 4;
}

template <int Q, bool LDS_CM>
static int lsa_launch(const float* cost, const unsigned long long* sorted, float* cm, const int* seg, int small, int large, int n_pred,
                      int* assoc, int* status, int n_scans, nr_stream_t stream) {
  const int64_t lds = lsa_lds_bytes(small, Q);
  hipLaunchKernelGGL((lsa_kernel<Q, LDS_CM>), dim3((unsigned)n_scans), dim3(NR_WAVE), (size_t)lds, nr_s(stream), cost, sorted, cm, seg,
                     small, large, n_pred, assoc, status);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_radar_assign(const float* pred, int n_scans, int64_t n_pred, const float* detections, int det_stride,
                               const int* seg, int max_detections, int cost_type, int* assoc, void* workspace, nr_stream_t stream) {
  if (n_scans == 0 || n_pred == 0) return 0;
  if (!pred || !detections || !seg || !assoc || !workspace || n_scans < 0 || n_pred < 0 || max_detections < 0 || det_stride < 3 ||
      (cost_type != 0 && cost_type != 1) || n_pred > 0x3fffffff || ((uintptr_t)workspace & 7))
    return NR_EINVAL;
  const int64_t small = max_detections < n_pred ? max_detections : n_pred, large = max_detections < n_pred ? n_pred : max_detections;
  if (small > kLsaMaxRows || large > kLsaMaxCols) return NR_EINVAL;
  const int Q = lsa_q(small);
  unsigned long long* sorted = reinterpret_cast<unsigned long long*>(workspace);
  float* cost = reinterpret_cast<float*>(sorted + (int64_t)n_scans * small * small);
  float* cm = cost + (int64_t)n_scans * small * large;
  int* status = reinterpret_cast<int*>(cm + (int64_t)n_scans * small * Q * NR_WAVE);
  hipError_t e = hipMemsetAsync(status, 0, sizeof(int) * n_scans, nr_s(stream));
  if (e != hipSuccess) return (int)e;
  if (small > 0) {
    hipLaunchKernelGGL(radar_cost_sort_kernel, dim3((unsigned)small, (unsigned)n_scans), dim3(kSortThreads), 0, nr_s(stream), pred,
                       (int)n_pred, detections, det_stride, seg, cost_type, (int)small, (int)large, cost, sorted);
    NR_LAUNCH_CHECK();
  }
#define NR_LSA_LAUNCH(QQ, IN_LDS) \
  return lsa_launch<QQ, IN_LDS>(cost, sorted, cm, seg, (int)small, (int)large, (int)n_pred, assoc, status, n_scans, stream)
  switch (Q) {
    case 1: NR_LSA_LAUNCH(1, true);
    case 2: NR_LSA_LAUNCH(2, true);
    case 4:
      if (lsa_cm_in_lds(small)) NR_LSA_LAUNCH(4, true);
      NR_LSA_LAUNCH(4, false);
    case 8: NR_LSA_LAUNCH(8, false);
    default: NR_LSA_LAUNCH(16, false);
  }
#undef NR_LSA_LAUNCH
}

int nr_init_radar() {  // the six lsa_kernel variants stage up to the CU's whole LDS; the heads' backward 160 KB at 64 inputs
  if (int rc = nr_raise_lds(lsa_kernel<1, true>, kLsaLdsBytes)) return rc;
  if (int rc = nr_raise_lds(lsa_kernel<2, true>, kLsaLdsBytes)) return rc;
  if (int rc = nr_raise_lds(lsa_kernel<4, true>, kLsaLdsBytes)) return rc;
  if (int rc = nr_raise_lds(lsa_kernel<4, false>, kLsaLdsBytes)) return rc;
  if (int rc = nr_raise_lds(lsa_kernel<8, false>, kLsaLdsBytes)) return rc;
  if (int rc = nr_raise_lds(lsa_kernel<16, false>, kLsaLdsBytes)) return rc;
  return nr_raise_lds(radar_heads_bwd_kernel, 160 * 1024);
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

extern "C" int nr_radar_points_fwd(const float* depth, const float* dirs_spher, int64_t n, const float* dim_t, const int* code, int C,
                                   float* xyz, float* dirs, float* pos, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!depth || !dirs_spher || !dim_t || !code || !xyz || !dirs || !pos || n < 0 || C < 3) return NR_EINVAL;
  hipLaunchKernelGGL(radar_points_fwd_kernel, dim3((unsigned)nr_cdiv(n * C, 256)), dim3(256), 0, nr_s(stream), depth, dirs_spher, n, dim_t,
                     code, C, xyz, dirs, pos);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_radar_points_bwd(const float* g_xyz, const float* dirs, int64_t n, float* g_depth, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!g_xyz || !dirs || !g_depth || n < 0) return NR_EINVAL;
  hipLaunchKernelGGL(radar_points_bwd_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), g_xyz, dirs, n, g_depth);
  NR_LAUNCH_CHECK();
  return 0;
}

static size_t rh_lds_bytes(int C, bool bwd) {
  size_t f = 0;
  const int outs[3] = {3, 1, 3};
  for (int h = 0; h < 3; ++h) f += (size_t)rh_layout(C, outs[h]).size;
  f += (size_t)kRhRays * (C + 1);
  if (bwd) f += (size_t)3 * kRhRays * 68 + (size_t)3 * kRhRays * (C + 1);
  return f * sizeof(float);
}

extern "C" int nr_radar_heads_fwd(const nr_radar_heads_t* heads, const float* x, int in_dim, const float* xyz, int64_t n, float* out,
                                  nr_stream_t stream) {
  if (n == 0) return 0;
  if (!heads || !x || !xyz || !out || n < 0 || in_dim < 1 || in_dim > kRhMaxIn) return NR_EINVAL;
  for (int h = 0; h < 3; ++h)
    for (int l = 0; l < 3; ++l)
      if (!heads->weight[h][l] || !heads->bias[h][l]) return NR_EINVAL;
  hipLaunchKernelGGL(radar_heads_fwd_kernel, dim3((unsigned)nr_cdiv(n, kRhRays)), dim3(kRhThreads), rh_lds_bytes(in_dim, false), nr_s(stream),
                     *heads, x, in_dim, xyz, n, out);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_radar_heads_bwd(const nr_radar_heads_t* heads, const float* x, int in_dim, const float* grad_out, int64_t n,
                                  float* grad_x, float* grad_xyz, const nr_radar_heads_grads_t* grads, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!heads || !x || !grad_out || !grad_x || !grad_xyz || !grads || n < 0 || in_dim < 1 || in_dim > kRhMaxIn) return NR_EINVAL;
  for (int h = 0; h < 3; ++h)
    for (int l = 0; l < 3; ++l)
      if (!heads->weight[h][l] || !heads->bias[h][l] || !grads->weight[h][l] || !grads->bias[h][l]) return NR_EINVAL;
  const size_t lds = rh_lds_bytes(in_dim, true);  // (64 rays x 3 heads of activations: > 64 KB at 64 inputs; raised in nr_init)
  hipLaunchKernelGGL(radar_heads_bwd_kernel, dim3((unsigned)nr_cdiv(n, kRhRays)), dim3(kRhThreads), lds, nr_s(stream), *heads, x, in_dim,
                     grad_out, n, grad_x, grad_xyz, *grads);
  NR_LAUNCH_CHECK();
  return 0;
}
