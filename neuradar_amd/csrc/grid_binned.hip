// Hash-grid scatter-add for INCOHERENT rows (lidar rays: neighbouring rows share no fine-level cell, so the on-chip
// dedup of hash_encode_bwd_kernel finds nothing to merge and the launch runs at the memory side's float-atomic rate,
// one 64-byte request per lane: 20 G requests/s).  Two passes, no atomics on single table entries:
//   1. bin:   every (row, level, corner) contribution becomes an (entry index, F values) pair, routed by the table slice
//             ("bucket", 128 KB of table) its entry falls into.  A wave owns 1 024 rows of one level and a private region
//             of every bucket's queue, so appending needs no global counter: pairs collect in wave-private LDS bins and
//             leave in runs of >= half a bin with plain stores.
//   2. apply: one workgroup per (level, bucket) keeps its slice of the gradient table in LDS (128 KB), streams the
//             bucket's queue and accumulates with LDS float adds, then adds the slice to the table with CONTIGUOUS
//             atomics (256 B per wave-instruction: the shape the memory side applies at ~1.3 TB/s).
// Anything that does not fit (a bin or a region overflowing: many rows in one cell) falls back to a direct atomic.
#include <limits.h>

#include <type_traits>

#include "nr_common.h"

namespace {

constexpr int kWaves = 4;           // waves per binning block
constexpr int kRowsPerWave = 1024;  // rows of one level binned by one wave = one queue region per bucket
constexpr int kMaxBuckets = 32;
constexpr int kSliceFloats = 32768;  // 128 KB of gradient table per bucket

template <int F> struct BinCfg { static constexpr int CAP = F == 1 ? 64 : 32; };  // entries per wave-private LDS bin

// Orders the wave's LDS traffic for the compiler.  LDS operations of one wave execute in program order, so no wait is
// needed between a lane's store and another lane's later load; a __builtin_amdgcn_fence here would also drain the
// wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)) -- after every queue flush: measured 11 k cycles per round.
__device__ __forceinline__ void wfence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

struct BinGeom {
  int rb_log2;   // log2(rows per bucket)
  int nb;        // buckets per level
  int cap;       // entries per (bucket, region)
  int64_t regions;
};

inline bool bin_geom(int F, int log2T, int64_t n, BinGeom* g) {
  if (F != 1 && F != 2 && F != 4) return false;
  int rows_log2 = 15;  // 32768 floats
  for (int f = F; f > 1; f >>= 1) --rows_log2;
  g->rb_log2 = log2T < rows_log2 ? log2T : rows_log2;
  const int64_t nb = (int64_t)1 << (log2T - g->rb_log2);
  if (nb > kMaxBuckets) return false;
  g->nb = (int)nb;
  const int expect = kRowsPerWave * 8 / g->nb;
  int cap = (2 * expect + 63) / 64 * 64;
  if (cap > kRowsPerWave * 8) cap = kRowsPerWave * 8;
  g->cap = cap;
  g->regions = nr_cdiv(n, kRowsPerWave);
  return true;
}

template <int F>
__global__ void __launch_bounds__(kWaves * 64)
bin_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int log2T,
           const float* __restrict__ gout, int64_t sn, int64_t sl, float* __restrict__ gtable, int64_t n, int nb, int rb_log2,
           int cap, int64_t regions, uint32_t* __restrict__ qidx, float* __restrict__ qval, uint32_t* __restrict__ qcnt) {
  constexpr int CAP = BinCfg<F>::CAP, LINE = 32;  // a flush moves exactly LINE entries: one aligned 128-B line of indices
  __shared__ uint32_t s_idx[kWaves][kMaxBuckets][CAP];
  __shared__ float s_val[kWaves][kMaxBuckets][CAP * F];
  __shared__ uint32_t s_cnt[kWaves][kMaxBuckets];
  __shared__ uint32_t s_off[kWaves][kMaxBuckets];
  const int level = blockIdx.y, lane = nr_lane(), wave = threadIdx.x >> 6;
  const int64_t region = (int64_t)blockIdx.x * kWaves + wave;
  if (region >= regions) return;
  uint32_t (*bidx)[CAP] = s_idx[wave];
  float (*bval)[CAP * F] = s_val[wave];
  uint32_t* cnt = s_cnt[wave];
  uint32_t* off = s_off[wave];
  if (lane < kMaxBuckets) { cnt[lane] = 0u; off[lane] = 0u; }
  wfence();
  const float scale = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u, lmask = (1u << rb_log2) - 1u;
  float* tbase = gtable + (((int64_t)level << log2T) * F);
  const float* gl = gout + (int64_t)level * sl;
  auto qbase = [&](int b) { return (((int64_t)level * nb + b) * regions + region) * cap; };

  // moves the first `take` (<= 64) entries of bin b to the wave's region of bucket b; the rest slides to the front
  auto flush_bin = [&](int b, uint32_t take) {
    const uint32_t have = cnt[b] < (uint32_t)CAP ? cnt[b] : (uint32_t)CAP;  // the same address in every lane: broadcast
    const uint32_t o = off[b];
    const int64_t dst = qbase(b) + o;
    uint32_t li = 0u, li2 = 0u;
    float v[F], v2[F];
    const bool mine = (uint32_t)lane < take, rest = take + lane < have;
    if (mine) {
      li = bidx[b][lane];
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = bval[b][lane * F + f];
    }
    if (rest) {
      li2 = bidx[b][take + lane];
#pragma unroll
      for (int f = 0; f < F; ++f) v2[f] = bval[b][(take + lane) * F + f];
    }
    wfence();
    if (mine) {
      if (o + lane < (uint32_t)cap) {
        qidx[dst + lane] = li;
#pragma unroll
        for (int f = 0; f < F; ++f) qval[(dst + lane) * F + f] = v[f];
      } else {  // this wave's region of the bucket is full: straight to the table
#pragma unroll
        for (int f = 0; f < F; ++f) unsafeAtomicAdd(tbase + ((((int64_t)b << rb_log2) + li) * F + f), v[f]);
      }
    }
    if (rest) {
      bidx[b][lane] = li2;
#pragma unroll
      for (int f = 0; f < F; ++f) bval[b][lane * F + f] = v2[f];
    }
    if (lane == 0) {
      off[b] = o + take < (uint32_t)cap ? o + take : (uint32_t)cap;
      cnt[b] = have - take;
    }
    wfence();
  };
  auto flush_lines = [&]() {  // until no bin holds a full line (a bin can hold two)
    while (true) {
      const uint32_t c = lane < nb ? cnt[lane] : 0u;
      unsigned long long full = __ballot(c >= (uint32_t)LINE);
      if (!full) break;
      while (full) {
        const int b2 = __builtin_ctzll(full);
        full &= full - 1ull;
        flush_bin(b2, LINE);
      }
    }
  };

  const int64_t r0 = region * kRowsPerWave;
  float nx[3] = {0.0f, 0.0f, 0.0f}, nstd = 0.0f, ng[F];
#pragma unroll
  for (int f = 0; f < F; ++f) ng[f] = 0.0f;
  auto fetch = [&](int64_t row) {  // the NEXT round's inputs are requested before this round is processed
    if (row < n) {
#pragma unroll
      for (int a = 0; a < 3; ++a) nx[a] = x[row * 3 + a];
      if (std != nullptr) nstd = std[row];
#pragma unroll
      for (int f = 0; f < F; ++f) ng[f] = gl[row * sn + f];
    }
  };
  fetch(r0 + lane);
#pragma unroll 1
  for (int it = 0; it < kRowsPerWave / 64; ++it) {
    const int64_t row = r0 + it * 64 + lane;
    if (r0 + it * 64 >= n) break;  // wave-uniform
    const bool valid = row < n;
    float cx[3] = {nx[0], nx[1], nx[2]}, cstd = nstd, cg[F];
#pragma unroll
    for (int f = 0; f < F; ++f) cg[f] = ng[f];
    if (it + 1 < kRowsPerWave / 64) fetch(row + 64);
    int lo[3] = {INT_MIN + lane, 0, 0};  // invalid lanes: a cell of their own
    float v[8][F];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) v[corner][f] = 0.0f;
    if (valid) {
      float cw[3], g[F];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float p = cx[a] * scale;
        const float fl = floorf(p);
        lo[a] = (int)fl;
        cw[a] = p - fl;
      }
      float r = 1.0f;
      if (std != nullptr) r = 1.0f / fmaxf(scale * 2.0f * cstd, 1.0f);
#pragma unroll
      for (int f = 0; f < F; ++f) g[f] = cg[f] * r;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
        const float w = (hx ? cw[0] : 1.0f - cw[0]) * (hy ? cw[1] : 1.0f - cw[1]) * (hz ? cw[2] : 1.0f - cw[2]);
#pragma unroll
        for (int f = 0; f < F; ++f) v[corner][f] = g[f] * w;
      }
    }
    // runs of consecutive lanes in one cell (neighbouring samples of a ray at the coarse levels) are summed on the
    // wave: segmented inclusive scan on DPP moves, the run's last lane carries the sum (as hash_encode_bwd_kernel)
    const bool head = lane == 0 || !(nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[0]) == lo[0] &&
                                     nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[1]) == lo[1] &&
                                     nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[2]) == lo[2]);
    // ballot, not a DPP move: the compiler may re-evaluate a cheap DPP inside the divergent loops below, where a
    // disabled source lane makes it return its `old` operand (seen: lanes in the middle of a run appended as tails)
    const unsigned long long heads = __ballot(head);
    const bool tail = lane == NR_WAVE - 1 || ((heads >> (lane + 1)) & 1ull) != 0ull;
    int flag = head ? 1 : 0;
    auto scan_step = [&](auto ctrl, auto rowmask) {
      constexpr int C = decltype(ctrl)::value, R = decltype(rowmask)::value;
      const float take = flag ? 0.0f : 1.0f;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
#pragma unroll
        for (int f = 0; f < F; ++f) v[corner][f] = __builtin_fmaf(nr_dpp_f<C, R>(0.0f, v[corner][f]), take, v[corner][f]);
      flag |= nr_dpp_i<C, R>(0, flag);
    };
    scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 1>{}, std::integral_constant<int, 0xF>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 2>{}, std::integral_constant<int, 0xF>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 4>{}, std::integral_constant<int, 0xF>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 8>{}, std::integral_constant<int, 0xF>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x2>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x4>{});
    scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x8>{});
    // all eight corners of a run tail are appended in one round: eight LDS counter bumps in flight, one fence
    uint32_t hs[8];
    unsigned pend = 0u;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      hs[corner] = nr_hash3(lo[0] + (corner & 1), lo[1] + ((corner >> 1) & 1), lo[2] + ((corner >> 2) & 1), mask);
      bool nz = false;
#pragma unroll
      for (int f = 0; f < F; ++f) nz = nz || v[corner][f] != 0.0f;
      if (nz && tail && valid) pend |= 1u << corner;
    }
#pragma unroll 1
    while (__any(pend != 0u)) {
      uint32_t pos[8];
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
        pos[corner] = (pend >> corner) & 1u ? atomicAdd(&cnt[hs[corner] >> rb_log2], 1u) : 0xFFFFFFFFu;  // LDS, wave-private
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
        if (pos[corner] < (uint32_t)CAP) {  // a full bin is retried after the flush
          const int b = (int)(hs[corner] >> rb_log2);
          bidx[b][pos[corner]] = hs[corner] & lmask;
#pragma unroll
          for (int f = 0; f < F; ++f) bval[b][pos[corner] * F + f] = v[corner][f];
          pend &= ~(1u << corner);
        }
      wfence();
      flush_lines();
    }
  }
  {  // what is left: partial lines
    const uint32_t c = lane < nb ? cnt[lane] : 0u;
    unsigned long long some = __ballot(c > 0u);
    while (some) {
      const int b2 = __builtin_ctzll(some);
      some &= some - 1ull;
      const uint32_t have = cnt[b2] < (uint32_t)CAP ? cnt[b2] : (uint32_t)CAP;
      flush_bin(b2, have);
    }
  }
  if (lane < nb) qcnt[((int64_t)level * nb + lane) * regions + region] = off[lane];
}

template <int F>
__global__ void __launch_bounds__(1024)
apply_kernel(const uint32_t* __restrict__ qidx, const float* __restrict__ qval, const uint32_t* __restrict__ qcnt,
             int64_t regions, int cap, int nb, int rb_log2, int log2T, float* __restrict__ gtable) {
  __shared__ float slice[kSliceFloats];
  const int level = blockIdx.y, b = blockIdx.x, lane = nr_lane(), wave = threadIdx.x >> 6;
  const int count = (1 << rb_log2) * F;
  for (int i = threadIdx.x; i < count; i += blockDim.x) slice[i] = 0.0f;
  __syncthreads();
  const int64_t q0 = ((int64_t)level * nb + b) * regions;
  // A lane takes 4 CONSECUTIVE entries (16-byte loads): rows of one cell sit next to each other in the queue, and
  // neighbouring lanes adding to one LDS address serialise inside the instruction; the lane sums equal neighbours itself.
  const int waves = blockDim.x >> 6;
  constexpr int kInFlight = 2;  // regions per wave whose loads are issued before the first LDS add
  for (int64_t reg0 = wave; reg0 < regions; reg0 += (int64_t)waves * kInFlight) {
    uint32_t have[kInFlight];
#pragma unroll
    for (int r = 0; r < kInFlight; ++r) have[r] = reg0 + r * waves < regions ? qcnt[q0 + reg0 + r * waves] : 0u;
    uint32_t most = 0u;
#pragma unroll
    for (int r = 0; r < kInFlight; ++r) most = have[r] > most ? have[r] : most;
    for (uint32_t i0 = 0; i0 < most; i0 += 256) {
      uint4 li[kInFlight];
      float v[kInFlight][4][F];
#pragma unroll
      for (int r = 0; r < kInFlight; ++r) {
        const int64_t base = (q0 + reg0 + r * waves) * cap;  // cap is a multiple of 64 entries: 16-byte aligned
        const uint32_t i = i0 + lane * 4;
        li[r] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int f = 0; f < F; ++f) v[r][e][f] = 0.0f;
        if (i < have[r]) {  // entries past `have` inside the 4-group are stale memory: masked below
          li[r] = *reinterpret_cast<const uint4*>(qidx + base + i);
          const float4* pv = reinterpret_cast<const float4*>(qval + (base + i) * F);
#pragma unroll
          for (int q = 0; q < F; ++q) {
            const float4 t4 = pv[q];
            const float tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[r][(q * 4 + j) / F][(q * 4 + j) % F] = tt[j];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kInFlight; ++r) {
        const uint32_t i = i0 + lane * 4;
        const uint32_t idx4[4] = {li[r].x, li[r].y, li[r].z, li[r].w};
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = i + e < have[r];
#pragma unroll
          for (int f = 0; f < F; ++f) acc[f] += ok ? v[r][e][f] : 0.0f;
          const bool last = e == 3 || !(i + e + 1 < have[r]) || idx4[e + 1 < 4 ? e + 1 : 3] != idx4[e];
          if (ok && last) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
              if (acc[f] != 0.0f) atomicAdd(&slice[idx4[e] * F + f], acc[f]);  // ds_add_f32
              acc[f] = 0.0f;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  float* out = gtable + ((((int64_t)level << log2T) + ((int64_t)b << rb_log2)) * F);
  for (int i = threadIdx.x; i < count; i += blockDim.x) {
    const float v = slice[i];
    if (v != 0.0f) unsafeAtomicAdd(out + i, v);  // other launches add into the same table concurrently
  }
}

}  // namespace

extern "C" int64_t nr_hash_encode_bwd_binned_workspace_bytes(int L, int F, int log2T, int64_t n) {
  BinGeom g;
  if (L < 1 || log2T < 1 || log2T > 30 || n < 0 || !bin_geom(F, log2T, n, &g)) return -1;
  const int64_t slots = (int64_t)L * g.nb * g.regions;
  return slots * 4 + slots * g.cap * 4 + slots * g.cap * 4 * F + 256;
}

extern "C" int nr_hash_encode_bwd_binned(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                         const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, void* workspace,
                                         nr_stream_t stream) {
  if (n == 0) return 0;
  BinGeom g;
  if (!x || !gout || !scalings || !gtable || !workspace || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  if (!bin_geom(F, log2T, n, &g) || ((uintptr_t)workspace & 15u) != 0) return NR_EINVAL;
  const int64_t slots = (int64_t)L * g.nb * g.regions;
  uint32_t* qcnt = static_cast<uint32_t*>(workspace);
  uint32_t* qidx = qcnt + (slots + 3) / 4 * 4;
  float* qval = reinterpret_cast<float*>(qidx + slots * g.cap);
  dim3 grid1((unsigned)nr_cdiv(g.regions, kWaves), (unsigned)L), grid2((unsigned)g.nb, (unsigned)L);
#define CALL(FF)                                                                                                          \
  {                                                                                                                        \
    hipLaunchKernelGGL(bin_kernel<FF>, grid1, dim3(kWaves * 64), 0, nr_s(stream), x, std, scalings, log2T, gout, sn, sl,   \
                       gtable, n, g.nb, g.rb_log2, g.cap, g.regions, qidx, qval, qcnt);                                       \
    hipLaunchKernelGGL(apply_kernel<FF>, grid2, dim3(1024), 0, nr_s(stream), qidx, qval, qcnt, g.regions, g.cap, g.nb,      \
                       g.rb_log2, log2T, gtable);                                                                          \
  }
  switch (F) {
    case 1: CALL(1) break;
    case 2: CALL(2) break;
    case 4: CALL(4) break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}
