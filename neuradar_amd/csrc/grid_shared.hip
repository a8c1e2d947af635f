// Hash-grid scatter-add through a BLOCK-SHARED, vertex-keyed LDS table on 32-bit integer atomics: the main grid's scatter of
// steps whose MLPs run on 16-bit operands (NeuRadar's grid: 8 levels x 2^22 entries x 4 floats = 64 MB per level, far too
// many 64-KB slices for the slice-owner kernels of grid_binned.hip).  Backward of HashEncoding.pytorch_fwd
// (field_components/encodings.py:406-466) = a dense index_put(accumulate=True) into the [L*T, F] gradient.
//
// Why: the merging kernel of grid.hip keeps a wave-PRIVATE table of cells (8 corners x F floats = 128 B per
// cell, 35 KB per wave): one wave per SIMD, ~2 500 instructions per 64 rows and level (a 32-float segmented scan + an owner
// protocol instead of atomics), 9 us per iteration with nothing else on the SIMD to hide a dependent instruction behind.
// LDS integer atomics (ds_cmpst_rtn_b32 / ds_add_u32: 7-12 lane-ops per clock and CU, tools/lds_lab.hip) make a table that
// the whole block shares cheap: keyed by VERTEX (one entry of the level = one 16-byte record), so the rows of all the block's
// waves and the corners that neighbouring cells share fold into one record; 20 bytes per slot; no scan, no owner words.
//
//   block = 256 threads = one tile of 256 rows (as stored: 256 neighbouring rays at one sample slot for camera / radar rows,
//   8 rays x 32 samples for lidar rows), levels one after the other; per (tile, level):
//     1. every thread: its row's gradient (one float4, level-major), per-level rescale, block maximum -> the tile's
//        fixed-point scale (21 bits below the largest |gradient|: 256 addends of one sign fit 31 bits);
//     2. insert-or-add its 8 corners: ds_cmpst on the key (open addressing, 3 840 slots for at most 2 048 distinct vertices),
//        4 x ds_add_u32; a lane that claims an empty slot appends the slot to the block's list of occupied slots;
//     3. flush: lane = (occupied slot, feature) -- the 4 floats of a vertex are 4 adjacent lanes of ONE global atomic
//        instruction = one 16-byte request -- and the slot is empty again.
//   Two LDS-only barriers per (tile, level); everything a tile needs from memory is requested one tile ahead.
// Rounding: an addend is rounded to 2^-22 ... 2^-21 of the largest gradient entry of its 256-row tile on that level (smaller
// contributions vanish) -- the companion of 16-bit MLP operands (u = 2^-8 / 2^-11), like the 32-bit tile sums of the bin
// pass; sums of integers do not depend on the order of the addends.  Non-finite gradients go to the table directly.
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "grid_dev.h"
#include "nr_common.h"

namespace {

// R6 experiments, measured and NOT adopted (code in git history, commit bbf50ab): the table keyed by the
// 64-byte LINE of the gradient table (one atomic request per line instead of per vertex: -21 % requests, but +6 % wave-cycles:
// 870 vs 698 us alone, step 2.70 vs 2.31 ms); two blocks per CU on a 3 328-slot table (alone 441 -> 292 us, in the step +2..5 %:
// the bin blocks of the proposal scatters lose their LDS); the three scatters one after the other (+15 %); two features' sums in one
// 64-bit integer (16 ds_add_u64 instead of 32 ds_add_u32 per row: alone -4 % / -1.5 %, step +-0).
constexpr int kRows = 256;          // rows per tile
// SPLIT (template parameter of the kernel; NR_TUNE_SHARED_SPLIT): a row's 8 corners shared out to SPLIT threads -- part p of the
// block = threads [256 p, 256 p + 256) takes corners [p kCpt, p kCpt + kCpt) of row tid & 255.  The table allows one block per CU,
// and with one thread per row that is ONE wave per SIMD with nothing to hide an LDS round trip behind.  Every part loads the row
// and computes its cell (cheap, L1 hits); the addends, the segmented scan, the inserts and the adds -- the work -- are split, and
// there are SPLIT waves per SIMD to overlap.  R6, same box: the kernel by itself 680 -> 558 (2) -> 544 us (4); in the step with the
// three scatters side by side (one GPU) 2.13 -> 2.25 ms on a fresh model (the proposal scatters speed up, this kernel and the
// Adam behind it -- the critical path -- fall back) and 2.11 -> 2.03-2.11 after 600 steps; with this scatter by itself first (the
// data-parallel order: the main table's exchange follows it) 2.39 -> 2.33 fresh, 2.32 -> 2.14-2.22 after 600 steps.  Default: 1
// beside the other scatters, 2 in the data-parallel order (fused_step sets the knob).
constexpr int kWaves = kRows / NR_WAVE;  // waves per part
#ifndef NR_SHARED_SLOTS
#define NR_SHARED_SLOTS 3840
#endif
#ifndef NR_SHARED_PROBES
#define NR_SHARED_PROBES NR_SHARED_SLOTS
#endif
constexpr int kSlots = NR_SHARED_SLOTS;  // 3 840 = 15 x 256: load factor <= 0.54 when no two of the tile's 2 048 corners coincide.  (A/B
                                         // builds: a smaller table -- more blocks per CU in the same LDS -- with NR_SHARED_PROBES bounding the
                                         // linear probing; a vertex that finds no slot goes to the table directly)
constexpr int kMaxProbes = NR_SHARED_PROBES;
constexpr int kMaxOcc = kRows * 8 < kSlots ? kRows * 8 : kSlots;  // distinct vertices of a tile on one level (at most one per slot)
constexpr int kFixBits = 21;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kMaxLevels = 8;
constexpr int kPlane = kSlots + 16;  // floats of one feature's plane of sums: slot s of feature f at f * kPlane + s -- an insert's 64
                                     // lanes (one feature, random slots) spread over all LDS banks, and so do a flush's lanes
                                     // (4 features of one slot: banks s, s + 16, s + 32, s + 48); [slot][feature] rows put the
                                     // 64 lanes of every insert on a quarter of the banks (PMC: 4x the bank-conflict cycles)

__device__ __forceinline__ void lds_barrier() {  // orders LDS traffic only (no vmcnt(0): the next tile's loads stay in flight)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// -DNR_SHARED_CLOCKS: wave-cycles per phase, summed over all waves (tools/probe_main_shared.py reads them through
// nr_debug_shared_clocks, which only that build exports)
#ifdef NR_SHARED_CLOCKS
__device__ unsigned long long g_shared_clocks[8];
#define NR_CLK(i) { const long long t_ = clock64(); clk[i] += t_ - tlast; tlast = t_; }
#else
#define NR_CLK(i)
#endif

__device__ __forceinline__ uint32_t slot_of(uint32_t key) { return __umulhi(key * 2654435761u, (uint32_t)kSlots); }

template <bool MARK, int SPLIT>
__global__ void __launch_bounds__(kRows * SPLIT)
scatter_shared_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int L,
                      int log2T, const float* __restrict__ gout, int64_t sl, float* __restrict__ gtable, int64_t n,
                      int64_t n_tiles, unsigned char* __restrict__ seen) {
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "corners per thread: 8, 4 or 2");
  constexpr int F = 4, kThreads = kRows * SPLIT, kCpt = 8 / SPLIT;
  __shared__ __attribute__((aligned(16))) uint32_t keys[kSlots];
  __shared__ __attribute__((aligned(16))) uint32_t vals[kPlane * F];
  __shared__ uint16_t occ[kMaxOcc];
  __shared__ uint32_t count[2];
  __shared__ float wmax[kWaves];
  const int tid = threadIdx.x, lane = tid & (NR_WAVE - 1), rtid = tid & (kRows - 1), part = tid / kRows, wave = rtid >> 6;
  const uint32_t mask = (1u << log2T) - 1u;
#ifdef NR_SHARED_PRIO
  __builtin_amdgcn_s_setprio(NR_SHARED_PRIO);
#endif
  for (int i = tid; i < kSlots; i += kThreads) keys[i] = kEmpty;
  for (int i = tid; i < kPlane * F; i += kThreads) vals[i] = 0u;
  if (tid < 2) count[tid] = 0u;
  __syncthreads();

  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  // Everything a tile needs from memory -- position, std, the gradient rows of all its levels -- is requested one TILE ahead
  // and waited for once per tile: gfx9 has one counter for loads and atomics, and behind a flush loop of unknown length the
  // compiler can only wait for "everything", i.e. for every float atomic of the flush to come back from the memory side.
  // (A wait per (tile, level) made every unit cost the same 15 us whatever its level held.)  The level loop is unrolled so
  // that the register array of gradients is indexed statically (a dynamic index moves it to scratch memory).
  float px[3] = {0.0f, 0.0f, 0.0f}, pstd = 0.0f;
  float4 pg[kMaxLevels];
  auto fetch = [&](int64_t t) {
    const int64_t row = t * kRows + rtid;
    const bool in = row < n;
#pragma unroll
    for (int a = 0; a < 3; ++a) px[a] = in ? x[row * 3 + a] : 0.0f;
    pstd = (in && std != nullptr) ? std[row] : 0.0f;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
      pg[l] = (in && l < L) ? *reinterpret_cast<const float4*>(gout + (int64_t)l * sl + row * F) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  };
  fetch(tile);
  uint32_t unit = 0;
#ifdef NR_SHARED_CLOCKS
  long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
#pragma unroll 1
  for (; tile < n_tiles; tile += gridDim.x) {
    float cx[3], cstd;
    float4 cg[kMaxLevels];
#pragma unroll
    for (int a = 0; a < 3; ++a) cx[a] = px[a];
    cstd = pstd;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) cg[l] = pg[l];
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
    NR_CLK(0)
#pragma unroll
    for (int level = 0; level < kMaxLevels; ++level) {
      if (level >= L) break;
      float g[F] = {cg[level].x, cg[level].y, cg[level].z, cg[level].w};
      const float scale = scalings[level];
      float* base = gtable + (((int64_t)level << log2T) * F);
      const bool live = g[0] != 0.0f || g[1] != 0.0f || g[2] != 0.0f || g[3] != 0.0f;  // (true for NaN)
      float r = 1.0f;
      if (std != nullptr) r = 1.0f / fmaxf(scale * 2.0f * cstd, 1.0f);  // neurad_encoding.py:314
      float mag = 0.0f;
      bool finite = true;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        g[f] *= r;
        finite = finite && fabsf(g[f]) <= 3.0e38f;  // (false for NaN)
        mag = fmaxf(mag, fabsf(g[f]));
      }
      // ---- the row's 8 corners: entry index and trilinear weight.  The reference takes the ceil / floor corners with weight
      // `offset` on the ceil side (encodings.py:434,454-464); floor + 1 is the same vertex wherever its weight is not zero
      // (an integer coordinate has ceil = floor and offset 0: nothing is added on that side either way), and makes the 8
      // entries a function of the CELL alone -- what the merge across lanes below relies on
      uint32_t idx[kCpt];
      float w[kCpt];
      int lo[3];
      {
        float o[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float p = cx[a] * scale;
          const float fl = floorf(p);
          lo[a] = (int)fl;
          o[a] = p - fl;
        }
#pragma unroll
        for (int c = 0; c < kCpt; ++c) {
          const int cc = part * kCpt + c;  // (the corner's number: bit 0 / 1 / 2 = the high side in x / y / z)
          const bool hx = cc & 1, hy = cc & 2, hz = cc & 4;
          idx[c] = nr_hash3(lo[0] + (hx ? 1 : 0), lo[1] + (hy ? 1 : 0), lo[2] + (hz ? 1 : 0), mask);
          w[c] = (hx ? o[0] : 1.0f - o[0]) * (hy ? o[1] : 1.0f - o[1]) * (hz ? o[2] : 1.0f - o[2]);
        }
      }
      // ---- block maximum -> fixed-point scale of (tile, level)
      float vmax = nr_wave_max_to_lane63(live && finite ? mag : 0.0f);
      if (lane == NR_WAVE - 1 && part == 0) wmax[wave] = vmax;
      NR_CLK(1)
      lds_barrier();  // A: the wave maxima are visible; the previous unit's flush is complete
      NR_CLK(2)
      float bmax = wmax[0];
#pragma unroll
      for (int k = 1; k < kWaves; ++k) bmax = fmaxf(bmax, wmax[k]);
      // bmax < 2^e (exponent field of the float); sums in units of 2^(e - kFixBits)
      int e = (int)((__float_as_uint(bmax) >> 23) & 0xFFu) - 126;
      e = e < kFixBits - 126 ? kFixBits - 126 : e;  // (the scale stays a normal float)
      const float fix = __uint_as_float((uint32_t)(kFixBits - e + 127) << 23);
      const float inv_fix = __uint_as_float((uint32_t)(e - kFixBits + 127) << 23);
      uint32_t* cnt = &count[unit & 1u];
      if (live && !finite) {
        // inf / NaN gradients (an overflowed 16-bit operand upstream): straight to the table, like torch's index_put would
#pragma unroll
        for (int c = 0; c < kCpt; ++c)
#pragma unroll
          for (int f = 0; f < F; ++f) unsafeAtomicAdd(base + (int64_t)idx[c] * F + f, g[f] * w[c]);
        if (MARK)
#pragma unroll
          for (int c = 0; c < kCpt; ++c) seen[((int64_t)level << log2T) + idx[c]] = 1;
      }
      // ---- the row's 32 addends in fixed point, then merged across the wave's lanes: neighbouring lanes in ONE cell (camera
      // pixels at a coarse level -- or at every level while a fresh model's samples sit within a metre of the camera; samples
      // of one lidar ray in a coarse cell) are summed by a segmented scan on DPP operands, and only the last lane of a run goes
      // to the table.  Integer sums: the result is the same whatever is merged where; without the scan 64 lanes of a wave
      // queue on one LDS address (PMC: 4x the bank-conflict cycles of the bin pass, the insert phase 68 % of the kernel).
      const bool ins_row = live && finite;
      int q[kCpt][F];
#pragma unroll
      for (int c = 0; c < kCpt; ++c)
#pragma unroll
        for (int f = 0; f < F; ++f) q[c][f] = ins_row ? (int)rintf(g[f] * w[c] * fix) : 0;
      int cell[3] = {ins_row ? lo[0] : INT_MIN + lane, ins_row ? lo[1] : INT_MIN + lane, ins_row ? lo[2] : INT_MIN + lane};
      const bool head = lane == 0 || !(nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, cell[0]) == cell[0] &&
                                       nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, cell[1]) == cell[1] &&
                                       nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, cell[2]) == cell[2]);
      const unsigned long long heads = __ballot(head);
      if (heads != ~0ull) {  // (uniform) some run is longer than one lane
        int flag = head ? 1 : 0;
        auto scan_step = [&](auto ctrl, auto rowmask) {  // (nr_seg_scan_step: two vector instructions per element and in-row step)
          nr_seg_scan_step<decltype(ctrl)::value, decltype(rowmask)::value, kCpt * F>(&q[0][0], flag, lane);
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
      // ---- insert-or-add of the run sums.  The first probe of all 8 corners is issued before any is looked at (independent
      // LDS round trips); the adds return nothing; the slots a lane has claimed join the block's list of occupied slots with
      // ONE returning atomic per wave for all 8 corners
      NR_CLK(6)
      uint32_t nz = 0u;  // bit c: corner c has something to add
#pragma unroll
      for (int c = 0; c < kCpt; ++c) nz |= ((q[c][0] | q[c][1] | q[c][2] | q[c][3]) != 0 ? 1u : 0u) << c;
      const bool ins = ins_row && tail;
      uint32_t s[kCpt], old[kCpt];
#pragma unroll
      for (int c = 0; c < kCpt; ++c) {
        s[c] = slot_of(idx[c]);
        old[c] = (ins && ((nz >> c) & 1u)) ? atomicCAS(&keys[s[c]], kEmpty, idx[c]) : idx[c];  // ds_cmpst_rtn_b32
      }
      uint32_t claimed = 0u;  // bit c: this lane created the slot of corner c
#pragma unroll
      for (int c = 0; c < kCpt; ++c) {
        const bool act = ins && ((nz >> c) & 1u);
        if (act && old[c] == kEmpty) claimed |= 1u << c;
        if (act && old[c] != kEmpty && old[c] != idx[c]) {  // taken by another vertex: linear probing
          uint32_t sc = s[c];
          int probes = 0;
          while (true) {
            sc = sc + 1u == (uint32_t)kSlots ? 0u : sc + 1u;
            const uint32_t o2 = atomicCAS(&keys[sc], kEmpty, idx[c]);
            if (o2 == kEmpty) { claimed |= 1u << c; break; }
            if (o2 == idx[c]) break;
            if (++probes >= kMaxProbes) { sc = kEmpty; break; }  // (default build: cannot happen, at most 2 048 keys for 3 840 slots)
          }
          s[c] = sc;
        }
        if (act && s[c] != kEmpty) {
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (q[c][f] != 0) atomicAdd(&vals[f * kPlane + s[c]], (uint32_t)q[c][f]);  // ds_add_u32
        } else if (act) {  // (no slot: see above)
#pragma unroll
          for (int f = 0; f < F; ++f) unsafeAtomicAdd(base + (int64_t)idx[c] * F + f, (float)q[c][f] * inv_fix);
          if (MARK) seen[((int64_t)level << log2T) + idx[c]] = 1;
        }
      }
      NR_CLK(7)
      {
        // One returning atomic per wave reserves the span of the list for all 8 corners of all its lanes; inside it the entries
        // go lane-major.  (Corner-major -- neighbouring rays' same-numbered corners, i.e. the vertices of one 64-byte table line, on
        // adjacent lanes of the flush -- was measured 6 % SLOWER per step, same call: 2.44 vs 2.28-2.32 ms.)
        unsigned long long cm[kCpt];
        int total = 0;
#pragma unroll
        for (int c = 0; c < kCpt; ++c) {
          cm[c] = __ballot((claimed >> c) & 1u);
          total += (int)__popcll(cm[c]);
        }
        if (total > 0) {  // (uniform)
          uint32_t at = 0;
          if (lane == 0) at = atomicAdd(cnt, (uint32_t)total);  // ds_add_rtn_u32
          at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
          const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
          for (int c = 0; c < kCpt; ++c) at += (uint32_t)__popcll(cm[c] & below);
#pragma unroll
          for (int c = 0; c < kCpt; ++c)
            if ((claimed >> c) & 1u) occ[at++] = (uint16_t)s[c];
        }
      }
      NR_CLK(3)
      lds_barrier();  // B: every insert is done
      NR_CLK(4)
      // ---- flush: lane = (occupied slot, feature); the slot is empty again afterwards.  Four slots per lane and trip, their LDS
      // reads issued together
      const uint32_t n_items = *cnt * F;
      if (tid == 0) count[(unit + 1u) & 1u] = 0u;  // (the other counter: last read in the previous unit's flush)
      for (uint32_t item0 = (uint32_t)tid; item0 < n_items; item0 += kThreads * 4) {
        uint32_t sc[4], key[4];
        int q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t item = item0 + (uint32_t)k * kThreads;
          sc[k] = item < n_items ? occ[item >> 2] : kEmpty;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t f = (item0 + (uint32_t)k * kThreads) & 3u;
          key[k] = sc[k] != kEmpty ? keys[sc[k]] : 0u;
          q[k] = sc[k] != kEmpty ? (int)vals[f * kPlane + sc[k]] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t f = (item0 + (uint32_t)k * kThreads) & 3u;
          if (sc[k] == kEmpty) continue;
          vals[f * kPlane + sc[k]] = 0u;
          if (f == 0u) keys[sc[k]] = kEmpty;  // (the 4 lanes of a slot sit in one wave instruction: all have read the key)
          if (q[k] != 0) {
            unsafeAtomicAdd(base + (int64_t)key[k] * F + f, (float)q[k] * inv_fix);
            if (MARK) seen[((int64_t)level << log2T) + key[k]] = 1;
          }
        }
      }
      NR_CLK(5)
      ++unit;
    }
  }
#ifdef NR_SHARED_CLOCKS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_shared_clocks[i], (unsigned long long)clk[i]);
#endif
}

// Stamp every table group (entry) that the rows' gradients can reach on any level: the 8 vertices (floor, floor + 1 per axis --
// this file's scatter's) of every row's cell.  One byte per entry of the [L * T] table; the value comes from a device-resident step
// counter.  A superset of what the scatter writes (vertices with weight 0, rows with a zero gradient) is fine: see adam_split_kernel.
__global__ void __launch_bounds__(256)
hash_mark_kernel(const float* __restrict__ x, const float* __restrict__ scalings, int L, int log2T, int64_t n, uint8_t* __restrict__ stamp,
                 const float* __restrict__ epoch) {
  const uint8_t now = (uint8_t)nr_stamp_value(epoch);
  const uint32_t mask = (1u << log2T) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float px = x[i * 3], py = x[i * 3 + 1], pz = x[i * 3 + 2];
    for (int l = 0; l < L; ++l) {
      const float s = scalings[l];
      const int lx = (int)floorf(px * s), ly = (int)floorf(py * s), lz = (int)floorf(pz * s);
      uint8_t* base = stamp + ((int64_t)l << log2T);
#pragma unroll
      for (int c = 0; c < 8; ++c) base[nr_hash3(lx + (c & 1), ly + ((c >> 1) & 1), lz + (c >> 2), mask)] = now;
    }
  }
}

}  // namespace

#ifdef NR_SHARED_CLOCKS
extern "C" int nr_debug_shared_clocks(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_shared_clocks), 64);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_shared_clocks), z, 64);
  }
  return (int)e;
}
#endif

extern "C" int nr_hash_encode_bwd_shared_split(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                               const float* grad_out, int64_t sn, int64_t sl, float* grad_table, int64_t n,
                                               unsigned char* seen_grad, int threads_per_row, nr_stream_t stream) {
  if (threads_per_row != 0 && threads_per_row != 1 && threads_per_row != 2 && threads_per_row != 4) return NR_EINVAL;
  if (n == 0) return 0;
  if (!x || !scalings || !grad_out || !grad_table || n < 0 || L < 1 || L > kMaxLevels || log2T < 1 || log2T > 30) return NR_EINVAL;
  // built for 4-float entries read as one float4 per row and level (the level-major [L, n, 4] gradient of the fused step)
  if (F != 4 || sn != 4 || (sl & 3) != 0 || (((uintptr_t)grad_out | (uintptr_t)grad_table) & 15u) != 0) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, kRows);
  // ONE block per CU (81 KB of its 160 KB of LDS): beside the step's other scatters -- whose bin blocks take 48 KB each -- that
  // beats two per CU by 4.5 % per step, same call (2.27 -> 2.17 ms fresh; 384 / 192 / 128 blocks: 2.24 / 2.20 / 2.45 ms)
  const int cap = nr_tuning().shared_blocks > 0 ? nr_tuning().shared_blocks : 256;
  const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
  int split = threads_per_row;  // (the A/B knob overrides the caller)
  if (nr_tuning().shared_split == 1 || nr_tuning().shared_split == 2 || nr_tuning().shared_split == 4) split = nr_tuning().shared_split;
  if (split == 0) split = 1;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kRows * split), 0, nr_s(stream), x, std, scalings, L, log2T, grad_out, sl, grad_table, n,
                       tiles, seen_grad);
  };
  const bool mark = seen_grad != nullptr;
  if (split == 1) mark ? launch(scatter_shared_kernel<true, 1>) : launch(scatter_shared_kernel<false, 1>);
  else if (split == 2) mark ? launch(scatter_shared_kernel<true, 2>) : launch(scatter_shared_kernel<false, 2>);
  else mark ? launch(scatter_shared_kernel<true, 4>) : launch(scatter_shared_kernel<false, 4>);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_hash_encode_bwd_shared(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                         const float* grad_out, int64_t sn, int64_t sl, float* grad_table, int64_t n,
                                         unsigned char* seen_grad, nr_stream_t stream) {
  return nr_hash_encode_bwd_shared_split(x, std, scalings, L, F, log2T, grad_out, sn, sl, grad_table, n, seen_grad, 0, stream);
}

extern "C" int nr_hash_mark_vertices(const float* x, const float* scalings, int L, int log2T, int64_t n, uint8_t* stamp, const float* epoch,
                                     nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !scalings || !stamp || !epoch || n < 0 || L < 1 || log2T < 1 || log2T > 30) return NR_EINVAL;
  const int64_t want = nr_cdiv(n, 256);
  hipLaunchKernelGGL(hash_mark_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, nr_s(stream), x, scalings, L, log2T, n, stamp,
                     epoch);
  NR_LAUNCH_CHECK();
  return 0;
}
