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

#include "grid_dev.h"
#include "nr_common.h"

namespace {

constexpr int kRows = 256;          // threads per block = rows per tile
constexpr int kWaves = kRows / NR_WAVE;
constexpr int kSlots = 3840;        // 15 x 256: load factor <= 0.54 when no two of the tile's 2 048 corners coincide
constexpr int kMaxOcc = kRows * 8;  // distinct vertices of a tile on one level
constexpr int kFixBits = 21;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kMaxLevels = 8;

__device__ __forceinline__ void lds_barrier() {  // orders LDS traffic only (no vmcnt(0): the next tile's loads stay in flight)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ uint32_t slot_of(uint32_t key) { return __umulhi(key * 2654435761u, (uint32_t)kSlots); }

template <bool MARK>
__global__ void __launch_bounds__(kRows)
scatter_shared_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int L,
                      int log2T, const float* __restrict__ gout, int64_t sl, float* __restrict__ gtable, int64_t n,
                      int64_t n_tiles, unsigned char* __restrict__ seen) {
  constexpr int F = 4;
  __shared__ __attribute__((aligned(16))) uint32_t keys[kSlots];
  __shared__ __attribute__((aligned(16))) uint32_t vals[kSlots * F];
  __shared__ uint16_t occ[kMaxOcc];
  __shared__ uint32_t count[2];
  __shared__ float wmax[kWaves];
  const int tid = threadIdx.x, lane = tid & (NR_WAVE - 1), wave = tid >> 6;
  const uint32_t mask = (1u << log2T) - 1u;
  for (int i = tid; i < kSlots; i += kRows) keys[i] = kEmpty;
  for (int i = tid; i < kSlots * F; i += kRows) vals[i] = 0u;
  if (tid < 2) count[tid] = 0u;
  __syncthreads();

  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  float px[3] = {0.0f, 0.0f, 0.0f}, pstd = 0.0f;
  float4 pg[kMaxLevels];
  auto fetch = [&](int64_t t) {
    const int64_t row = t * kRows + tid;
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
#pragma unroll 1
    for (int level = 0; level < L; ++level, ++unit) {
      float4 g4 = cg[0];
#pragma unroll
      for (int l = 1; l < kMaxLevels; ++l) g4 = level == l ? cg[l] : g4;  // (level is uniform: selects, no indexing)
      float g[F] = {g4.x, g4.y, g4.z, g4.w};
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
      // ---- the row's 8 corners: entry index and trilinear weight (the reference's ceil / floor corners, weight `offset` on
      // the ceil side, encodings.py:434,454-464)
      uint32_t idx[8];
      float w[8];
      {
        int lo[3], hi[3];
        float o[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float p = cx[a] * scale;
          const float fl = floorf(p);
          lo[a] = (int)fl;
          hi[a] = (int)ceilf(p);
          o[a] = p - fl;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool hx = c & 1, hy = c & 2, hz = c & 4;
          idx[c] = nr_hash3(hx ? hi[0] : lo[0], hy ? hi[1] : lo[1], hz ? hi[2] : lo[2], mask);
          w[c] = (hx ? o[0] : 1.0f - o[0]) * (hy ? o[1] : 1.0f - o[1]) * (hz ? o[2] : 1.0f - o[2]);
        }
      }
      // ---- block maximum -> fixed-point scale of (tile, level)
      float vmax = nr_wave_max_to_lane63(live && finite ? mag : 0.0f);
      if (lane == NR_WAVE - 1) wmax[wave] = vmax;
      lds_barrier();  // A: the wave maxima are visible; the previous unit's flush is complete
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
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int f = 0; f < F; ++f) unsafeAtomicAdd(base + (int64_t)idx[c] * F + f, g[f] * w[c]);
        if (MARK)
#pragma unroll
          for (int c = 0; c < 8; ++c) seen[((int64_t)level << log2T) + idx[c]] = 1;
      }
      // ---- insert-or-add.  The first probe of all 8 corners is issued before any is looked at (independent LDS round trips)
      const bool ins = live && finite;
      uint32_t s[8], old[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        s[c] = slot_of(idx[c]);
        old[c] = ins ? atomicCAS(&keys[s[c]], kEmpty, idx[c]) : idx[c];  // ds_cmpst_rtn_b32
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        bool claimed = ins && old[c] == kEmpty;
        if (ins && old[c] != kEmpty && old[c] != idx[c]) {  // taken by another vertex: linear probing
          uint32_t sc = s[c];
          int probes = 0;
          while (true) {
            sc = sc + 1u == (uint32_t)kSlots ? 0u : sc + 1u;
            const uint32_t o2 = atomicCAS(&keys[sc], kEmpty, idx[c]);
            if (o2 == kEmpty) { claimed = true; break; }
            if (o2 == idx[c]) break;
            if (++probes >= kSlots) { sc = kEmpty; break; }  // (cannot happen: at most 2 048 keys for 3 840 slots)
          }
          s[c] = sc;
        }
        // a claimed slot joins the list of occupied slots: one returning atomic per wave instruction
        const unsigned long long cm = __ballot(claimed);
        if (cm != 0ull) {
          uint32_t at = 0;
          if (lane == (int)__builtin_ctzll(cm)) at = atomicAdd(cnt, (uint32_t)__popcll(cm));  // ds_add_rtn_u32
          at = (uint32_t)__builtin_amdgcn_readlane((int)at, (int)__builtin_ctzll(cm));
          if (claimed) occ[at + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = (uint16_t)s[c];
        }
        if (ins && s[c] != kEmpty) {
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const int q = (int)rintf(g[f] * w[c] * fix);
            if (q != 0) atomicAdd(&vals[s[c] * F + f], (uint32_t)q);  // ds_add_u32
          }
        } else if (ins) {  // (no slot: see above)
#pragma unroll
          for (int f = 0; f < F; ++f) unsafeAtomicAdd(base + (int64_t)idx[c] * F + f, g[f] * w[c]);
          if (MARK) seen[((int64_t)level << log2T) + idx[c]] = 1;
        }
      }
      lds_barrier();  // B: every insert is done
      // ---- flush: lane = (occupied slot, feature); the slot is empty again afterwards
      const uint32_t n_occ = *cnt;
      if (tid == 0) count[(unit + 1u) & 1u] = 0u;  // (the other counter: last read in the previous unit's flush)
      for (uint32_t item = (uint32_t)tid; item < n_occ * F; item += kRows) {
        const uint32_t sc = occ[item >> 2], f = item & 3u;
        const uint32_t key = keys[sc];
        const int q = (int)vals[sc * F + f];
        vals[sc * F + f] = 0u;
        if (f == 0u) keys[sc] = kEmpty;  // (the 4 lanes of a slot sit in one wave instruction: all have read the key)
        if (q != 0) {
          unsafeAtomicAdd(base + (int64_t)key * F + f, (float)q * inv_fix);
          if (MARK) seen[((int64_t)level << log2T) + key] = 1;
        }
      }
    }
  }
}

}  // namespace

extern "C" int nr_hash_encode_bwd_shared(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                         const float* grad_out, int64_t sn, int64_t sl, float* grad_table, int64_t n,
                                         unsigned char* seen_grad, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !scalings || !grad_out || !grad_table || n < 0 || L < 1 || L > kMaxLevels || log2T < 1 || log2T > 30) return NR_EINVAL;
  // built for 4-float entries read as one float4 per row and level (the level-major [L, n, 4] gradient of the fused step)
  if (F != 4 || sn != 4 || (sl & 3) != 0 || (((uintptr_t)grad_out | (uintptr_t)grad_table) & 15u) != 0) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(n, kRows);
  int cap = 512;  // two blocks per CU (81 KB of LDS each)
  if (const char* e = getenv("NR_SHARED_BLOCKS")) cap = atoi(e) > 0 ? atoi(e) : cap;  // tuning knob
  const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
  if (seen_grad != nullptr)
    hipLaunchKernelGGL(scatter_shared_kernel<true>, dim3(blocks), dim3(kRows), 0, nr_s(stream), x, std, scalings, L, log2T, grad_out,
                       sl, grad_table, n, tiles, seen_grad);
  else
    hipLaunchKernelGGL(scatter_shared_kernel<false>, dim3(blocks), dim3(kRows), 0, nr_s(stream), x, std, scalings, L, log2T, grad_out,
                       sl, grad_table, n, tiles, seen_grad);
  NR_LAUNCH_CHECK();
  return 0;
}
