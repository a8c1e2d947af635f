// Hash-grid scatter-add by TABLE SLICE OWNERSHIP, for levels whose table is hit densely (the NeuRadar proposal grid:
// 12 M contributions per level onto 1 M entries -- every 64-byte line of the level is touched ~160 times per step, and
// the merging kernel of grid.hip, which can only fold neighbouring rows, runs at the memory side's float-atomic rate,
// 20 G requests/s).  Two passes, no atomics on single table entries:
//   1. bin:   a block takes ROWS rows of one level, one thread per row.  The 8 corners of a row are 4 x-PAIRS
//             (x, x+1 | y, z): the slice (64 KB of table = the top bits of the entry index) of a pair depends on (y, z)
//             only, so a pair is ONE record {key, 2 F sums}.  Neighbouring lanes in one cell are summed across the wave
//             first (VALU segmented scan); the run sums are then MERGED block-wide in an LDS table keyed by the pair and
//             partitioned by slice: insert-or-add with LDS INTEGER atomics (ds_cmpst_rtn_b32 + ds_add_u64 on 64-bit
//             fixed-point sums scaled by the block's largest |value|; integer LDS atomics run at several lanes/clk/CU,
//             ds_add_f32 at 0.33 -- tools/lds_lab.hip).  A partition then leaves as one contiguous run of raw records
//             (key + integer sums), the "sub-bin" of (level, slice, tile), with the tile's exponent beside it: no global
//             counter, no global atomic, no conversion.
//   2. apply: one workgroup per (level, slice) streams the slice's sub-bins and accumulates them in LDS, shifting every
//             record from its tile's exponent to the level's (36 bits below the level's largest value are kept, 2^12 times
//             finer than fp32; the sum of the records does not depend on their order).  The slice is then added to the table
//             with CONTIGUOUS float atomics (other launches add into the same table concurrently; 256 B per
//             wave-instruction, the shape the memory side takes at ~1.3 TB/s).
// A record that finds its partition full after four probes, a pair that straddles two slices (negative corner) and
// non-finite values go to the table directly with float atomics.
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "nr_common.h"

namespace {

constexpr int kSliceFloats = 16384;  // 128 KB of int64 accumulators per apply block
constexpr int kMaxSlices = 64;
// A tile's sums in the bin pass's LDS table (SumT):
//   unsigned long long  64-bit fixed point 44 bits below the tile's largest value (2 048 addends: < 2^55) -- exact to fp32's
//                       resolution for every addend; the fp32 path.
//   uint32_t            32-bit fixed point 21 bits below it (a slot collects at most one addend per run of equal cells: <= 512
//                       per tile, < 2^31): 12 instead of 20 bytes of LDS per slot (three blocks per CU instead of two),
//                       ds_add_u32 at twice the rate of ds_add_u64, 12-byte records.  An addend is rounded to 2^-22..2^-21 of
//                       the TILE's largest contribution (smaller ones vanish): for the steps whose MLPs already run on
//                       16-bit operands (u = 2^-8 / 2^-11) -- the fused step picks it there; fresh step 2.55 -> 2.38 ms.
template <typename SumT> struct FixBits;
template <> struct FixBits<unsigned long long> { static constexpr int value = 44; };
template <> struct FixBits<uint32_t> { static constexpr int value = 21; };
constexpr int kLevelBits = 36;  // the level's sums (apply pass, always 64-bit): 36 bits below the level's largest (2^14 tiles: < 2^62)
constexpr int kApplyThreads = 1024;
constexpr int kNoRecords = INT_MIN;

// rows per bin block (one thread each) and slots of the block's LDS merge table.  The table is partitioned by slice
// (M / slices slots each: 64 at 64 slices, against an expected ROWS * 4 / 64 = M / 128 distinct records per slice when
// nothing merges), so a partition IS the sub-bin the block leaves for that slice.
template <int F> struct BinCfg;
#ifndef NR_BIN_F1
#define NR_BIN_F1 512, 4096
#endif
template <int R, int MM> struct BinCfgT { static constexpr int ROWS = R, M = MM; };
template <> struct BinCfg<1> : BinCfgT<NR_BIN_F1> {};
template <> struct BinCfg<2> { static constexpr int ROWS = 256, M = 2048; };
template <> struct BinCfg<4> { static constexpr int ROWS = 128, M = 1024; };
constexpr uint32_t kEmpty = 0xFFFFFFFFu, kReserved = 0xFFFFFFFEu;

struct BinGeom {
  int shift;   // entry index >> shift = slice
  int ns;      // slices per level
  int cap;     // records per sub-bin = slots per table partition (power of two)
  int cap_log2;
  int64_t nb;  // row tiles
};

inline bool bin_geom(int F, int log2T, int64_t n, BinGeom* g) {
  if (F != 1 && F != 2 && F != 4) return false;
  if (log2T > 20) return false;  // (entry index, trailing ones of x) is the 24-bit merge key
  int entries_log2 = 14;  // kSliceFloats / F entries per slice
  for (int f = F; f > 1; f >>= 1) --entries_log2;
  g->shift = log2T < entries_log2 ? log2T : entries_log2;
  const int64_t ns = (int64_t)1 << (log2T - g->shift);
  if (ns > kMaxSlices) return false;
  g->ns = (int)ns;
  const int rows = F == 1 ? BinCfg<1>::ROWS : F == 2 ? BinCfg<2>::ROWS : BinCfg<4>::ROWS;
  const int m = F == 1 ? BinCfg<1>::M : F == 2 ? BinCfg<2>::M : BinCfg<4>::M;
  g->cap = m / g->ns;
  g->cap_log2 = 0;
  while ((1 << g->cap_log2) < g->cap) ++g->cap_log2;
  g->nb = nr_cdiv(n, rows);
  return true;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a workgroup-scope fence for GLOBAL memory
// (s_waitcnt vmcnt(0)): it would wait for the prefetched loads of the next tile and for the record stores of the last
// flush on every barrier.  Nothing here communicates through global memory.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// y (|y| < 2^44) -> nearest int64, without the 64-bit conversion sequence of the compiler's float -> long long:
// the high part rounds to an integer of at most 25 bits, the remainder is exact in fp32
__device__ __forceinline__ long long fix_i64(float y) {
  const float hi = rintf(y * 0x1p-20f);
  const float lo = fmaf(hi, -0x1p20f, y);
  return ((long long)(int)hi << 20) + (long long)(int)rintf(lo);
}
template <typename SumT> __device__ __forceinline__ SumT fix_sum(float y);
template <> __device__ __forceinline__ unsigned long long fix_sum<unsigned long long>(float y) { return (unsigned long long)fix_i64(y); }
template <> __device__ __forceinline__ uint32_t fix_sum<uint32_t>(float y) { return (uint32_t)(int)rintf(y); }
template <typename SumT> __device__ __forceinline__ long long sum_to_i64(SumT v);
template <> __device__ __forceinline__ long long sum_to_i64<unsigned long long>(unsigned long long v) { return (long long)v; }
template <> __device__ __forceinline__ long long sum_to_i64<uint32_t>(uint32_t v) { return (long long)(int)v; }

// Persistent blocks: a block walks row tiles (ROWS rows) and, per tile, up to kLevelChunk levels; the LDS table is set up
// once and every flush leaves the slots it read empty again.  Two barriers per (tile, level): A -- the wave maxima are
// visible and the previous flush is complete; B -- all inserts are done.  Everything a tile needs from memory (position,
// std, the gradients of all its levels) is requested one tile ahead.
constexpr int kLevelChunk = 8;

// HEAD: the proposal field's density head is folded in (nr_prop_density_scatter_binned).  `gout` then holds the FORWARD
// features; the gradient of every level's feature is g * w[k] with g = g_density * d trunc_exp(feats . w), computed here
// from the tile's own loads exactly as prop_density_bwd_kernel does -- the [L, n, F] gradient buffer is never written or
// read -- and the head's weight gradient sum_rows g * feats leaves as one partial per block (folded by the apply kernel).
struct DensityHead {
  const float* w;          // [L * F]
  const float* g_density;  // [B, S] ray-major
  float* partials;         // [blocks, kLevelChunk * F] (workspace)
  int n_samples;
  int64_t sm_rays;
};

// -DNR_BIN_CLOCKS: wave-cycles per phase of the bin pass, summed over all waves (tools/probe_bin_phases.py reads them
// through nr_debug_bin_clocks, which only this build exports)
#ifdef NR_BIN_CLOCKS
__device__ unsigned long long g_bin_clocks[8];
#define NR_CLK(i) { const long long t_ = clock64(); clk[i] += t_ - tlast; tlast = t_; }
#else
#define NR_CLK(i)
#endif

// A second row SEGMENT for the same table (the other proposal round: both rounds scatter into proposal_fields[1]'s table,
// models/neuradar.py:302): its tiles follow the first segment's in the tile numbering, so ONE bin pass and ONE apply pass
// serve both -- the apply pass then walks each table slice once per step instead of once per round.
struct BinSegment {
  const float* x;
  const float* std;
  const float* gout;
  const float* g_density;
  int64_t n, sl, sm_rays;
  int n_samples;
};

// (F = 1 on 32-bit sums: 48 KB of LDS per block = THREE blocks per CU, which needs six waves per SIMD = at most 80 VGPRs; left alone
// the compiler took 84 -- five waves per SIMD, two resident blocks per CU, and the launch's third block per CU ran as a tail)
template <int F, bool HEAD, typename SumT>
__global__ void __launch_bounds__(BinCfg<F>::ROWS) __attribute__((amdgpu_waves_per_eu((F == 1 && sizeof(SumT) == 4) ? 6 : 1)))
bin_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int L, int level0,
           int log2T, const float* __restrict__ gout, int64_t sn, int64_t sl, float* __restrict__ gtable, int64_t n, int ns,
           int shift, int cap_log2, int64_t nb, uint32_t* __restrict__ rkey, SumT* __restrict__ rsum,
           uint32_t* __restrict__ cntg, int* __restrict__ tile_exp, DensityHead head, BinSegment seg2, int64_t nb1) {
  constexpr int ROWS = BinCfg<F>::ROWS, M = BinCfg<F>::M, WAVES = ROWS / NR_WAVE;
  // keys [M] | 64-bit sums [M][2][F].  Exactly half a CU's LDS at F = 1 (two blocks per CU), so the WAVES floats of the
  // block maximum live in the sums of the table's last slots, which are taken out of service (kReserved never matches)
  constexpr int kSumWords = sizeof(SumT) / 4, kFixBits = FixBits<SumT>::value;
  __shared__ __attribute__((aligned(16))) uint32_t lds[M + M * 2 * kSumWords * F];
  uint32_t* keys = lds;
  SumT* acc = reinterpret_cast<SumT*>(lds + M);
  constexpr int kScratchSlots = (WAVES * 4 + 8 * kSumWords * F - 1) / (8 * kSumWords * F);
  float* wmax = reinterpret_cast<float*>(acc + (size_t)(M - kScratchSlots) * 2 * F);
  const int tid = threadIdx.x, lane = tid & (NR_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the flush's sub-bin addresses stay in SGPRs)
  const int cap = 1 << cap_log2;
  const uint32_t mask = (1u << log2T) - 1u, pmask = (uint32_t)cap - 1u;
  const int nl = L - level0 < kLevelChunk ? L - level0 : kLevelChunk;
  for (int i = tid; i < M; i += ROWS) keys[i] = i < M - kScratchSlots ? kEmpty : kReserved;
  for (int i = tid; i < M * 2 * F; i += ROWS) acc[i] = 0;
  __syncthreads();

  int64_t tile = blockIdx.x;
  if (tile >= nb) return;
  float px[3] = {0.0f, 0.0f, 0.0f}, pstd = 0.0f, pg[kLevelChunk][F], pgd = 0.0f;
  float wk[kLevelChunk][F], part[kLevelChunk][F];
#pragma unroll
  for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
    for (int f = 0; f < F; ++f) {
      wk[l][f] = HEAD && l < nl ? head.w[(level0 + l) * F + f] : 0.0f;
      part[l][f] = 0.0f;
    }
  auto fetch = [&](int64_t t) {
    const bool second = t >= nb1;  // (uniform) tiles [nb1, nb) belong to the second segment
    const float* sx = second ? seg2.x : x;
    const float* sstd = second ? seg2.std : std;
    const float* sgout = second ? seg2.gout : gout;
    const int64_t sn_rows = second ? seg2.n : n, ssl = second ? seg2.sl : sl;
    const int64_t row = (second ? t - nb1 : t) * ROWS + tid;
    const bool in = row < sn_rows;
    if (in) {
#pragma unroll
      for (int a = 0; a < 3; ++a) px[a] = sx[row * 3 + a];
      if (std != nullptr) pstd = sstd[row];
    }
    if (HEAD)
      pgd = !in ? 0.0f
                : second ? seg2.g_density[nr_row_map(row, sn_rows, seg2.n_samples, seg2.sm_rays).out]
                         : head.g_density[nr_row_map(row, sn_rows, head.n_samples, head.sm_rays).out];
#pragma unroll
    for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
      for (int f = 0; f < F; ++f) pg[l][f] = in && l < nl ? sgout[(int64_t)(level0 + l) * ssl + row * sn + f] : 0.0f;
  };
  fetch(tile);
#ifdef NR_BIN_CLOCKS
  long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
#pragma unroll 1
  for (; tile < nb; tile += gridDim.x) {
    float cx[3], cstd, cg[kLevelChunk][F];
#pragma unroll
    for (int a = 0; a < 3; ++a) cx[a] = px[a];
    cstd = pstd;
#pragma unroll
    for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
      for (int f = 0; f < F; ++f) cg[l][f] = pg[l][f];
    if (HEAD) {  // prop_density_bwd_kernel's arithmetic, in its order
      float xs = 0.0f;
#pragma unroll
      for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
        for (int f = 0; f < F; ++f) xs += cg[l][f] * wk[l][f];
      const float g = pgd * expf(fminf(fmaxf(xs, -15.0f), 15.0f));  // activations.py:38-41
#pragma unroll
      for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          part[l][f] += g * cg[l][f];
          cg[l][f] = g * wk[l][f];
        }
    }
    if (tile + gridDim.x < nb) fetch(tile + gridDim.x);
    NR_CLK(0)
#pragma unroll 1
    for (int li = 0; li < nl; ++li) {
      const int level = level0 + li;
      float g[F], mag = 0.0f;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        g[f] = cg[0][f];
#pragma unroll
        for (int l = 1; l < kLevelChunk; ++l) g[f] = li == l ? cg[l][f] : g[f];  // (li is uniform: a select, no indexing)
        mag += fabsf(g[f]);
      }
      const float scale = scalings[level];
      float* base = gtable + (((int64_t)level << log2T) * F);
      // ---- 1. the row's four x-pairs, in registers
      uint32_t ia[4], tz = 0;
      float va[4][F], vb[4][F], mq[4];
      float vmax = 0.0f;
      int lo[3] = {INT_MIN + lane, INT_MIN + lane, INT_MIN + lane};  // rows without a gradient: a cell of their own
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ia[q] = 0;
#pragma unroll
        for (int f = 0; f < F; ++f) va[q][f] = vb[q][f] = 0.0f;
      }
      if (mag != 0.0f) {  // zero gradients (masked samples, rows past n) are skipped, like in the merging kernel
        float cw[3];
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
        for (int f = 0; f < F; ++f) g[f] *= r;
        tz = (uint32_t)__builtin_ctz(~(uint32_t)lo[0] | 0x80000000u);  // trailing ones of x: (x + 1) ^ x = 2^(tz + 1) - 1
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool hy = q & 1, hz = q & 2;
          const uint32_t h = ((uint32_t)(lo[1] + (hy ? 1 : 0)) * 2654435761u) ^ ((uint32_t)(lo[2] + (hz ? 1 : 0)) * 805459861u);
          ia[q] = ((uint32_t)lo[0] ^ h) & mask;
          // the weights in the merging kernel's order: wx * wy * wz, then g * w
          const float wa = (1.0f - cw[0]) * (hy ? cw[1] : 1.0f - cw[1]) * (hz ? cw[2] : 1.0f - cw[2]);
          const float wb = cw[0] * (hy ? cw[1] : 1.0f - cw[1]) * (hz ? cw[2] : 1.0f - cw[2]);
#pragma unroll
          for (int f = 0; f < F; ++f) {
            va[q][f] = g[f] * wa;
            vb[q][f] = g[f] * wb;
          }
        }
      }
      // ---- 1b. neighbouring lanes in one cell (camera pixels at a coarse level, samples of one ray in one cell) are
      // summed across the wave first, VALU only (the segmented scan of grid.hip): the last lane of a run carries the
      // run's sum to the table, so the LDS atomics see one lane per cell and no two lanes of a wave on one address
      const bool head = lane == 0 || !(nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[0]) == lo[0] &&
                                       nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[1]) == lo[1] &&
                                       nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[2]) == lo[2]);
      const unsigned long long heads = __ballot(head);
      if (heads != ~0ull) {  // (uniform) some run is longer than one lane
        int flag = head ? 1 : 0;
        auto scan_step = [&](auto ctrl, auto rowmask) {  // (nr_common.h: two vector instructions per element and in-row step)
          constexpr int C = decltype(ctrl)::value, R = decltype(rowmask)::value;
          int f2 = flag;
          nr_seg_scan_step<C, R, 4 * F>(&va[0][0], f2, lane);
          nr_seg_scan_step<C, R, 4 * F>(&vb[0][0], flag, lane);
        };
        // (the scan stops once every lane has reached its run's head: the remaining steps would add nothing)
        do {
          scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 1>{}, std::integral_constant<int, 0xF>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 2>{}, std::integral_constant<int, 0xF>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 4>{}, std::integral_constant<int, 0xF>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_SHR + 8>{}, std::integral_constant<int, 0xF>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x2>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x4>{});
          if (__ballot(flag == 0) == 0ull) break;
          scan_step(std::integral_constant<int, NR_DPP_ROW_BCAST15>{}, std::integral_constant<int, 0x8>{});
        } while (false);
      }
      const bool tail = lane == NR_WAVE - 1 || ((heads >> (lane + 1)) & 1ull);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float m = 0.0f;
        bool finite = true;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          finite = finite && fabsf(va[q][f]) <= 3.0e38f && fabsf(vb[q][f]) <= 3.0e38f;  // (false for NaN)
          m = fmaxf(m, fmaxf(fabsf(va[q][f]), fabsf(vb[q][f])));
        }
        mq[q] = !tail ? 0.0f : finite ? m : -1.0f;  // 0: nothing to add; -1: goes to the table directly
        if (tail && finite) vmax = fmaxf(vmax, m);
      }
      // ---- 2. block maximum -> the tile's fixed-point scale on this level
      vmax = nr_wave_max_to_lane63(vmax);
      if (lane == NR_WAVE - 1) wmax[wave] = vmax;
      NR_CLK(1)
      lds_barrier();  // A
      NR_CLK(2)
      float bmax = 0.0f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) bmax = fmaxf(bmax, wmax[w]);
      // bmax < 2^e (exponent field of the float; denormal maxima count as 2^-126); sums in units of 2^(e - 44)
      int e = (int)((__float_as_uint(bmax) >> 23) & 0xFFu) - 126;
      e = e < kFixBits - 126 ? kFixBits - 126 : e;  // (the scale stays a normal float)
      const float fix = __uint_as_float((uint32_t)(kFixBits - e + 127) << 23);
      // ---- 3. merge: insert-or-add into the slice's partition of the table (LDS integer atomics)
      const uint32_t pair = tz < 14u ? (2u << tz) - 1u : 0u;  // ia ^ ib
      const uint32_t key_hi = tz << 20;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (mq[q] == 0.0f) continue;
        const uint32_t ib = (ia[q] ^ pair) & mask;
        int slot = -1;
        if (mq[q] > 0.0f && pair != 0u && (ia[q] >> shift) == (ib >> shift)) {
          const uint32_t key = ia[q] | key_hi, part = (ia[q] >> shift) << cap_log2;
          uint32_t s0 = ia[q] & pmask;
#pragma unroll 1
          for (int probe = 0; probe < 4 && slot < 0; ++probe) {
            const uint32_t old = atomicCAS(&keys[part + s0], kEmpty, key);  // ds_cmpst_rtn_b32
            if (old == kEmpty || old == key) slot = (int)(part + s0);
            s0 = (s0 + 1u) & pmask;
          }
        }
        if (slot >= 0) {
#pragma unroll
          for (int f = 0; f < F; ++f) {
            atomicAdd(&acc[(slot * 2 + 0) * F + f], fix_sum<SumT>(va[q][f] * fix));  // ds_add_u64 / ds_add_u32
            atomicAdd(&acc[(slot * 2 + 1) * F + f], fix_sum<SumT>(vb[q][f] * fix));
          }
        } else {
          // partition full, pair across two slices (negative corner) or non-finite values: straight to the table
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (va[q][f] != 0.0f) unsafeAtomicAdd(base + (int64_t)ia[q] * F + f, va[q][f]);
          const uint32_t ib2 = (ia[q] ^ ((2u << tz) - 1u)) & mask;  // tz <= 31: 2u << 31 wraps to 0, all ones
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (vb[q][f] != 0.0f) unsafeAtomicAdd(base + (int64_t)ib2 * F + f, vb[q][f]);
        }
      }
      NR_CLK(3)
      lds_barrier();  // B
      NR_CLK(4)
      // ---- 4. every partition leaves as one contiguous run of raw records, its sub-bin, and is empty again afterwards
      const int64_t tl = (int64_t)level * nb + tile;
      const int64_t sub_stride = nb << cap_log2;
      int64_t sub = ((int64_t)level * ns + wave) * sub_stride + (tile << cap_log2);
      constexpr int NP = M / (NR_WAVE * WAVES);  // partitions per wave when a partition is one wave-width of slots
      if (cap_log2 == 6) {
        // (the usual geometry: ns = M / 64.)  The wave's partitions are read NB at a time -- keys first, then the sums of
        // the occupied slots, then the stores: two LDS round trips per BATCH instead of per partition (the flush was 44 %
        // of the kernel's wave-cycles with one partition in flight)
        constexpr int NB = F == 1 ? 4 : 2;  // (registers: NB * 2F 64-bit sums; the kernel must stay at 128 VGPRs for two blocks per CU)
        static_assert(NP % NB == 0, "partitions per wave");
#pragma unroll 1
        for (int j0 = 0; j0 < NP; j0 += NB) {
          uint32_t kk[NB];
          unsigned long long bal[NB];
          SumT av[NB][2 * F];
#pragma unroll
          for (int j = 0; j < NB; ++j) kk[j] = keys[((wave + (j0 + j) * WAVES) << 6) + lane];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const bool occ = kk[j] < kReserved;
            bal[j] = __ballot(occ);
            const int sl_ = ((wave + (j0 + j) * WAVES) << 6) + lane;
#pragma unroll
            for (int k = 0; k < 2 * F; ++k) av[j][k] = occ ? acc[sl_ * 2 * F + k] : (SumT)0;
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int sp_ = wave + (j0 + j) * WAVES;
            const int sl_ = (sp_ << 6) + lane;
            const int64_t subj = sub + (int64_t)(j0 + j) * WAVES * sub_stride;
            if (kk[j] < kReserved) {
              const uint32_t at = (uint32_t)__popcll(bal[j] & ((1ull << lane) - 1ull));
              rkey[subj + at] = kk[j];
#pragma unroll
              for (int k = 0; k < 2 * F; ++k) {
                rsum[(subj + at) * (2 * F) + k] = av[j][k];
                acc[sl_ * 2 * F + k] = 0;
              }
              keys[sl_] = kEmpty;
            }
            if (lane == 0) cntg[tl * ns + sp_] = (uint32_t)__popcll(bal[j]);
          }
        }
      } else {
      for (int s = wave; s < ns; s += WAVES, sub += WAVES * sub_stride) {
        uint32_t* dk = rkey + sub;                      // (scalar bases, 32-bit lane offsets)
        SumT* ds = rsum + sub * (2 * F);
        uint32_t count = 0;
        for (int c0 = 0; c0 < cap; c0 += NR_WAVE) {
          const int sl_ = (s << cap_log2) + c0 + lane;
          const uint32_t key = c0 + lane < cap ? keys[sl_] : kEmpty;
          const bool occ = key < kReserved;
          const unsigned long long bal = __ballot(occ);
          if (bal == 0ull) continue;
          if (occ) {
            const uint32_t at = count + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            dk[at] = key;
#pragma unroll
            for (int k = 0; k < 2 * F; ++k) {
              ds[at * (2 * F) + k] = acc[sl_ * 2 * F + k];
              acc[sl_ * 2 * F + k] = 0;
            }
            keys[sl_] = kEmpty;
          }
          count += (uint32_t)__popcll(bal);
        }
        if (lane == 0) cntg[tl * ns + s] = count;
      }
      }
      if (tid == 0) tile_exp[tl] = bmax > 0.0f ? e : kNoRecords;
      NR_CLK(5)
    }
  }
#ifdef NR_BIN_CLOCKS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_bin_clocks[i], (unsigned long long)clk[i]);
#endif
  if (HEAD) {  // the block's share of the head's weight gradient (the table is idle: its first words carry the wave sums)
    lds_barrier();
    float* red = reinterpret_cast<float*>(acc);
#pragma unroll
    for (int l = 0; l < kLevelChunk; ++l)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float t = nr_wave_sum(part[l][f]);
        if (lane == 0) red[wave * (kLevelChunk * F) + l * F + f] = t;
      }
    lds_barrier();
    if (tid < kLevelChunk * F) {
      float t = 0.0f;
      for (int w = 0; w < WAVES; ++w) t += red[w * (kLevelChunk * F) + tid];
      head.partials[(int64_t)blockIdx.x * (kLevelChunk * F) + tid] = t;
    }
  }
}

template <int F, typename SumT>
__global__ void __launch_bounds__(kApplyThreads)
apply_kernel(const uint32_t* __restrict__ rkey, const SumT* __restrict__ rsum, const uint32_t* __restrict__ cntg,
             const int* __restrict__ tile_exp, int64_t nb, int ns, int shift, int cap_log2, int log2T, float* __restrict__ gtable,
             const float* __restrict__ head_partials, int head_blocks, int head_dim, float* __restrict__ g_w) {
  __shared__ unsigned long long acc[kSliceFloats];
  __shared__ uint32_t cnts[kApplyThreads];  // count | right shift << 16
  __shared__ int red[kApplyThreads / NR_WAVE];
  const int tid = threadIdx.x, slice = blockIdx.x, level = blockIdx.y;
  if (head_partials != nullptr && slice == 0 && level == 0 && blockIdx.z == 0 && tid < head_dim) {  // the density head's weight gradient
    float t = 0.0f;
    for (int b = 0; b < head_blocks; ++b) t += head_partials[(int64_t)b * (kLevelChunk * F) + tid];
    unsafeAtomicAdd(g_w + tid, t);
  }
  const int slice_floats = (1 << shift) * F;
  for (int i = tid; i < slice_floats; i += kApplyThreads) acc[i] = 0ull;
  // the level's exponent: the largest of its tiles'
  int e_l = kNoRecords;
  for (int64_t b = tid; b < nb; b += kApplyThreads) e_l = max(e_l, tile_exp[(int64_t)level * nb + b]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e_l = max(e_l, __shfl_xor(e_l, o, NR_WAVE));
  if ((tid & (NR_WAVE - 1)) == 0) red[tid >> 6] = e_l;
  __syncthreads();
  e_l = kNoRecords;
  for (int w = 0; w < kApplyThreads / NR_WAVE; ++w) e_l = max(e_l, red[w]);
  if (e_l == kNoRecords) return;  // nothing was binned on this level
  const uint32_t mask = (1u << log2T) - 1u, in_mask = (1u << shift) - 1u;
  const int64_t sub0 = (((int64_t)level * ns + slice) * nb) << cap_log2;
  // Sub-bins are mostly far from full (a third at the finest level, a few records at the coarse ones), so a wave takes
  // EIGHT sub-bins per pass, eight lanes each, kBatch records per lane requested before the first is accumulated; walking
  // every slot of every sub-bin with one lane per slot made this kernel as long as the bin kernel.  blockIdx.z halves the
  // tile range (768 blocks at one per CU = three even rounds instead of one and a half).
  constexpr int kBatch = 4, kWaves = kApplyThreads / NR_WAVE;
  const int lane = tid & (NR_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane >> 3, l8 = lane & 7;
  const int64_t half = (nb + gridDim.z - 1) / gridDim.z, t_begin = (int64_t)blockIdx.z * half;
  const int64_t t_end = t_begin + half < nb ? t_begin + half : nb;
  for (int64_t b0 = t_begin; b0 < t_end; b0 += kApplyThreads) {
    __syncthreads();
    {
      uint32_t c = 0;
      if (b0 + tid < t_end) {
        const int e_b = tile_exp[(int64_t)level * nb + b0 + tid];
        if (e_b != kNoRecords) {
          c = cntg[((int64_t)level * nb + b0 + tid) * ns + slice];
          int sh = e_l - e_b + (FixBits<SumT>::value - kLevelBits);  // (32-bit records: -15 + ..., a left shift where the tile's exponent is near the level's)
          sh = sh > 63 ? 63 : sh;
          c |= (uint32_t)(sh + 64) << 16;
        }
      }
      cnts[tid] = c;
    }
    __syncthreads();
    const int tiles_here = (int)(t_end - b0 < kApplyThreads ? t_end - b0 : kApplyThreads);
    for (int tg = wave * 8; tg < tiles_here; tg += kWaves * 8) {
      const int t = tg + sub;
      const uint32_t c = t < tiles_here ? cnts[t] : 0u;
      const uint32_t cnt = c & 0xFFFFu;
      const int sh = (int)(c >> 16) - 64;
      const int64_t at0 = sub0 + ((b0 + t) << cap_log2);
      for (uint32_t r0 = 0; __any(r0 < cnt); r0 += kBatch * 8) {
        uint32_t key[kBatch];
        SumT q[kBatch][2 * F];
        bool ok[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          const uint32_t r = r0 + k * 8 + l8;
          ok[k] = r < cnt;
          key[k] = 0;
          if (ok[k]) {
            key[k] = rkey[at0 + r];
#pragma unroll
            for (int j = 0; j < 2 * F; ++j) q[k][j] = rsum[(at0 + r) * (2 * F) + j];
          }
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          if (!ok[k]) continue;
          const uint32_t a = key[k] & mask, tz = key[k] >> 20;
          const uint32_t b = (a ^ ((2u << tz) - 1u)) & mask;
          const uint32_t ea = (a & in_mask) * F, eb = (b & in_mask) * F;
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const long long ra = sum_to_i64<SumT>(q[k][f]), rb = sum_to_i64<SumT>(q[k][F + f]);
            const long long qa = sh >= 0 ? ra >> sh : ra << -sh, qb = sh >= 0 ? rb >> sh : rb << -sh;
            if (qa != 0) atomicAdd(&acc[ea + f], (unsigned long long)qa);  // ds_add_u64
            if (qb != 0) atomicAdd(&acc[eb + f], (unsigned long long)qb);
          }
        }
      }
    }
  }
  __syncthreads();
  const float unfix = ldexpf(1.0f, e_l - kLevelBits);
  float* out = gtable + ((((int64_t)level << log2T) + ((int64_t)slice << shift)) * F);
  for (int i = tid; i < slice_floats; i += kApplyThreads) {
    const long long a = (long long)acc[i];
    if (a != 0) unsafeAtomicAdd(out + i, (float)a * unfix);  // other launches add into the same table concurrently
  }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

inline int nr_num_cus() {
  int dev = 0, v = 0;  // (two runtime look-ups, ~0.1 us: no cache to keep)
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  return hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
}

struct Workspace {
  uint32_t* cntg;
  int* tile_exp;
  uint32_t* rkey;
  void* rsum;  // SumT records
  float* partials;  // [kMaxBinBlocks][kLevelChunk * F]: the density head's weight-gradient partials
  int64_t bytes;
};
constexpr int kMaxBinBlocks = 2048;

inline Workspace carve(void* workspace, int L, int F, const BinGeom& g) {
  Workspace w;
  char* p = static_cast<char*>(workspace);
  const int64_t slots = (int64_t)L * g.ns * g.nb * g.cap;
  w.cntg = reinterpret_cast<uint32_t*>(p);
  p += align_up((int64_t)L * g.nb * g.ns * 4, 256);
  w.tile_exp = reinterpret_cast<int*>(p);
  p += align_up((int64_t)L * g.nb * 4, 256);
  w.rkey = reinterpret_cast<uint32_t*>(p);
  p += align_up(slots * 4, 256);
  w.rsum = p;
  p += align_up(slots * 2 * F * 8, 256);  // (sized for the 64-bit records; the 32-bit ones use the first half)
  w.partials = reinterpret_cast<float*>(p);
  p += align_up((int64_t)kMaxBinBlocks * kLevelChunk * F * 4, 256);
  w.bytes = p - static_cast<char*>(workspace);
  return w;
}

}  // namespace

extern "C" int64_t nr_hash_encode_bwd_binned_workspace_bytes(int L, int F, int log2T, int64_t n) {
  BinGeom g;
  if (L < 1 || log2T < 1 || log2T > 30 || n < 0 || !bin_geom(F, log2T, n, &g)) return -1;
  return carve(nullptr, L, F, g).bytes;
}

// tiles of a launch over two segments: each segment's rows rounded up to whole tiles
static int64_t tiles_of(int F, int64_t n) {
  const int rows = F == 1 ? BinCfg<1>::ROWS : F == 2 ? BinCfg<2>::ROWS : BinCfg<4>::ROWS;
  return nr_cdiv(n, rows);
}

static int launch_binned(const float* x, const float* std, const float* scalings, int L, int F, int log2T, const float* gout,
                         int64_t sn, int64_t sl, float* gtable, int64_t n, void* workspace, nr_stream_t stream,
                         const DensityHead* head_in, float* g_w, const BinSegment* second = nullptr, int sum_bits = 64) {
  if (n == 0 && (second == nullptr || second->n == 0)) return 0;
  if (sum_bits != 32 && sum_bits != 64) return NR_EINVAL;
  BinGeom g;
  if (!x || !gout || !scalings || !gtable || !workspace || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  if (second != nullptr && (!second->x || !second->gout || second->n < 0 || (std != nullptr) != (second->std != nullptr))) return NR_EINVAL;
  const int64_t nb1 = tiles_of(F, n), nb2 = second != nullptr ? tiles_of(F, second->n) : 0;
  if (!bin_geom(F, log2T, n, &g) || ((uintptr_t)workspace & 15u) != 0 || nb1 + nb2 > INT_MAX) return NR_EINVAL;
  g.nb = nb1 + nb2;
  if (head_in != nullptr && (L > kLevelChunk || !head_in->w || !head_in->g_density || !g_w)) return NR_EINVAL;
  const Workspace w = carve(workspace, L, F, g);
  // persistent blocks: as many merge tables as fit a CU's 160 KB of LDS (20 bytes per slot with 64-bit sums at F = 1, 12 with
  // 32-bit ones), at most 8 waves per SIMD
  const int m_slots = F == 1 ? BinCfg<1>::M : F == 2 ? BinCfg<2>::M : BinCfg<4>::M;
  const int rows_blk = F == 1 ? BinCfg<1>::ROWS : F == 2 ? BinCfg<2>::ROWS : BinCfg<4>::ROWS;
  const int64_t lds_bytes = (int64_t)m_slots * (4 + 2 * F * (sum_bits / 8));
  int64_t per_cu = (160 * 1024) / lds_bytes;
  if (per_cu * rows_blk > 2048) per_cu = 2048 / rows_blk;
  if (per_cu < 1) per_cu = 1;
  int64_t persistent = per_cu * (int64_t)nr_num_cus();
  if (nr_tuning().bin_blocks_per_cu > 0) persistent = (int64_t)nr_tuning().bin_blocks_per_cu * nr_num_cus() / 2 > 0 ? (int64_t)nr_tuning().bin_blocks_per_cu * nr_num_cus() / 2 : persistent;  // (halves of a CU)
  if (persistent > kMaxBinBlocks) persistent = kMaxBinBlocks;
  const unsigned blocks = (unsigned)(g.nb < persistent ? g.nb : persistent);
  dim3 grid1(blocks), grid2((unsigned)g.ns, (unsigned)L, g.nb >= 64 ? 2u : 1u);
  DensityHead head = {nullptr, nullptr, w.partials, 0, 0};
  if (head_in != nullptr) {
    head = *head_in;
    head.partials = w.partials;
  }
  BinSegment seg2 = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (second != nullptr) seg2 = *second;
#define CALL_T(FF, ST)                                                                                                      \
  {                                                                                                                          \
    ST* rs = static_cast<ST*>(w.rsum);                                                                                       \
    if (head_in != nullptr)                                                                                                  \
      hipLaunchKernelGGL((bin_kernel<FF, true, ST>), grid1, dim3(BinCfg<FF>::ROWS), 0, nr_s(stream), x, std, scalings, L, 0,  \
                         log2T, gout, sn, sl, gtable, n, g.ns, g.shift, g.cap_log2, g.nb, w.rkey, rs, w.cntg, w.tile_exp,     \
                         head, seg2, nb1);                                                                                   \
    else                                                                                                                     \
      for (int l0 = 0; l0 < L; l0 += kLevelChunk)                                                                            \
        hipLaunchKernelGGL((bin_kernel<FF, false, ST>), grid1, dim3(BinCfg<FF>::ROWS), 0, nr_s(stream), x, std, scalings, L,  \
                           l0, log2T, gout, sn, sl, gtable, n, g.ns, g.shift, g.cap_log2, g.nb, w.rkey, rs, w.cntg,           \
                           w.tile_exp, head, seg2, nb1);                                                                     \
    hipLaunchKernelGGL((apply_kernel<FF, ST>), grid2, dim3(kApplyThreads), 0, nr_s(stream), w.rkey, rs, w.cntg, w.tile_exp,  \
                       g.nb, g.ns, g.shift, g.cap_log2, log2T, gtable, head_in != nullptr ? w.partials : nullptr,            \
                       (int)blocks, L * FF, g_w);                                                                            \
  }
#define CALL(FF)                                          \
  if (sum_bits == 32) CALL_T(FF, uint32_t) else CALL_T(FF, unsigned long long)
  switch (F) {
    case 1: CALL(1) break;
    case 2: CALL(2) break;
    case 4: CALL(4) break;
    default: return NR_EINVAL;
  }
#undef CALL
#undef CALL_T
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_hash_encode_bwd_binned(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                         const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, void* workspace,
                                         nr_stream_t stream) {
  return launch_binned(x, std, scalings, L, F, log2T, gout, sn, sl, gtable, n, workspace, stream, nullptr, nullptr);
}

extern "C" int nr_hash_encode_bwd_binned_lp(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                            const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, void* workspace,
                                            int sum_bits, nr_stream_t stream) {
  return launch_binned(x, std, scalings, L, F, log2T, gout, sn, sl, gtable, n, workspace, stream, nullptr, nullptr, nullptr, sum_bits);
}

extern "C" int nr_prop_density_scatter_binned(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                              const float* feats, int64_t sn, int64_t sl, const float* w, const float* g_density,
                                              int n_samples, int64_t rows_sample_major, float* gtable, float* g_w, int64_t n,
                                              void* workspace, nr_stream_t stream) {
  return nr_prop_density_scatter_binned_lp(x, std, scalings, L, F, log2T, feats, sn, sl, w, g_density, n_samples, rows_sample_major,
                                           gtable, g_w, n, workspace, 64, stream);
}

extern "C" int nr_prop_density_scatter_binned_lp(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                                 const float* feats, int64_t sn, int64_t sl, const float* w, const float* g_density,
                                                 int n_samples, int64_t rows_sample_major, float* gtable, float* g_w, int64_t n,
                                                 void* workspace, int sum_bits, nr_stream_t stream) {
  if (n == 0) return 0;
  if (rows_sample_major < 0 || (rows_sample_major && (n_samples < 1 || n % n_samples != 0 || rows_sample_major > n / n_samples)))
    return NR_EINVAL;
  const DensityHead head = {w, g_density, nullptr, n_samples, rows_sample_major};
  return launch_binned(x, std, scalings, L, F, log2T, feats, sn, sl, gtable, n, workspace, stream, &head, g_w, nullptr, sum_bits);
}

extern "C" int nr_prop_density_scatter_binned2(const float* x1, const float* std1, const float* feats1, const float* g_density1,
                                               int n_samples1, int64_t n1, const float* x2, const float* std2, const float* feats2,
                                               const float* g_density2, int n_samples2, int64_t n2, int64_t rows_sample_major,
                                               const float* scalings, int L, int F, int log2T, int64_t sn, const float* w,
                                               float* gtable, float* g_w, void* workspace, nr_stream_t stream) {
  if (n1 == 0 && n2 == 0) return 0;
  if (n1 <= 0 || n2 <= 0 || n_samples1 < 1 || n_samples2 < 1 || n1 % n_samples1 != 0 || n2 % n_samples2 != 0 ||
      n1 / n_samples1 != n2 / n_samples2 || rows_sample_major < 0 || rows_sample_major > n1 / n_samples1 || !g_density2)
    return NR_EINVAL;
  const DensityHead head = {w, g_density1, nullptr, n_samples1, rows_sample_major};
  const BinSegment second = {x2, std2, feats2, g_density2, n2, n2 * F, rows_sample_major, n_samples2};
  return launch_binned(x1, std1, scalings, L, F, log2T, feats1, sn, n1 * F, gtable, n1, workspace, stream, &head, g_w, &second);
}

#ifdef NR_BIN_CLOCKS
extern "C" int nr_debug_bin_clocks(unsigned long long* out8, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bin_clocks), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bin_clocks), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
