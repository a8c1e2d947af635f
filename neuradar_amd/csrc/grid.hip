// Hash-grid kernels: frustum -> contracted Gaussian, multiresolution hash gather (fwd) and
// scatter-add (bwd).  HBM/L2-bound integer+gather work: no MFMA here by design.
//
// Launch shape: blockIdx.y = level, so every wave gathers from ONE level's table region; with
// `sample_major` the 64 lanes of a wave are 64 neighbouring RAYS at the same sample slot, which for
// camera patches are centimetres apart -> the texture path merges most of a wave-instruction's
// addresses into a few cache lines on all but the finest levels.
#include <limits.h>

#include <type_traits>

#include "grid_dev.h"

using namespace nrgrid;

namespace {

// Proposal field forward in one launch (NeuRADProposalField.get_density, neurad_field.py:208-213): all levels of
// a sample in one thread, features stored for the backward, density = trunc_exp(feats . w) written [B,S].
template <int F>
__global__ void __launch_bounds__(256)
prop_field_fwd_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ table,
                      const float* __restrict__ scalings, int L, int log2T, const float* __restrict__ w,
                      float* __restrict__ out, int64_t sn, int64_t sl, int64_t n, int n_samples, int sm_rays,
                      float* __restrict__ density) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xs = 0.0f;
  for (int level = 0; level < L; ++level) {
    float feat[F];
    encode_level<F>(x, std, table, scalings[level], level, log2T, i, feat);
    float* o = out + i * sn + (int64_t)level * sl;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      o[f] = feat[f];
      xs += feat[f] * w[level * F + f];
    }
  }
  density[nr_row_map(i, n, n_samples, sm_rays).out] = expf(xs);
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
  float feat[F];
  encode_level<F>(x, std, table, scalings[level], level, log2T, idx, feat);
  float* o = out + idx * sn + (int64_t)level * sl;
#pragma unroll
  for (int f = 0; f < F; ++f) o[f] = feat[f];
}

// Backward scatter-add.  Memory-side float atomics are the scarce resource here (about 20 G
// requests/s chip-wide, far less when many lanes hit one address -- and coherent camera rays do exactly
// that: a 131k-sample batch touches only 10^1..10^4 distinct cells per level), so contributions are
// summed on chip first, keyed by grid CELL (a cell fixes all eight corner slots):
//   1. lanes l and l^32 (the same pixel column of two adjacent patch rows under the sample-major
//      mapping) that fall in the same cell are folded into one lane;
//   2. a segmented wave scan sums every run of consecutive lanes that share a cell;
//   3. the run's last lane adds its 8*F partial sums into its WAVE's private LDS hash table.  LDS
//      atomics (ds_add_f32 / ds_cmpst) measured an order of magnitude slower than plain LDS traffic,
//      so the table uses none: a wave owns its table exclusively, slots are claimed by
//      write-then-read-back, and lanes that target the same slot take turns through an owner word;
//   4. before an insert would push the table past three quarters full, and after the wave's CHUNK
//      samples, the occupied entries are flushed: their slots are compacted into a list first, then
//      lane = (entry, corner, feature) with the feature index fastest, so the F floats of a corner are F
//      adjacent lanes of ONE atomic wave-instruction and leave the L2 as one request (consecutive
//      instructions of one lane never merge: the F = 4 flush went from 189 to 87 us, incoherent lidar
//      rays from 2.0 to 0.55 ms).  A contribution that still finds no slot in four probes goes straight
//      to memory.
// The wave's inputs for iteration i+1 are requested before iteration i is processed (the loop holds no
// global atomics on its common path, so the loads stay in flight across it), and the sample-major
// index is advanced incrementally instead of divided out per sample.  PMC (profiles/): with the step's
// real gradients the kernel is bound by per-wave latency / VALU issue, not by the atomic rate.
// Per feature width: samples per wave (CHUNK), cells per wave table (CAP), waves per block (W; bounded by
// the 64 KB static LDS of a block).  Tuned on the bench's camera patches AND on incoherent lidar rays
// (tools/scatter_real.py, tools/probe_incoherent.py).
template <int F> struct BwdCfg;
#ifndef NR_BWD_F1
#define NR_BWD_F1 512, 256, 4
#endif
#ifndef NR_BWD_F2
#define NR_BWD_F2 512, 128, 4
#endif
template <int C, int P, int WV> struct BwdCfgT { static constexpr int CHUNK = C, CAP = P, W = WV; };
template <> struct BwdCfg<1> : BwdCfgT<NR_BWD_F1> {};
template <> struct BwdCfg<2> : BwdCfgT<NR_BWD_F2> {};
#ifndef NR_BWD_F4
#define NR_BWD_F4 256, 128, 2
#endif
#ifndef NR_BWD_F4_WIDE
#define NR_BWD_F4_WIDE 512, 256, 1
#endif
template <> struct BwdCfg<4> : BwdCfgT<NR_BWD_F4> {};
template <> struct BwdCfg<8> { static constexpr int CHUNK = 256, CAP = 64, W = 2; };
constexpr unsigned long long kEmptyKey = ~0ull;

__device__ __forceinline__ unsigned long long pack_cell(const int* lo) {
  return ((unsigned long long)(uint32_t)lo[0] & 0x1FFFFFull) | (((unsigned long long)(uint32_t)lo[1] & 0x1FFFFFull) << 21) |
         (((unsigned long long)(uint32_t)lo[2] & 0x1FFFFFull) << 42);
}

__device__ __forceinline__ void wave_fence() {
  // LDS operations of one wave execute in order; this only pins the compiler's ordering
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// seen: optional, one byte per four floats of this level's table region: set where a sum is added (the optimizer then leaves
// never-marked groups alone without reading their gradient, nr_adam_step_marked)
template <int F, int CAP>
__device__ __forceinline__ void flush_table(unsigned long long* keys, float* vals, int* list, float* base, uint32_t mask,
                                            int lane, bool reset, uint8_t* seen = nullptr) {
  // occupied slots first (list = the owner words, idle outside the insert phase): the flush then costs
  // occupied * 8 * F / 64 wave-instructions instead of CAP * 8 * F / 64
  int n_occ = 0;
#pragma unroll 1
  for (int k0 = 0; k0 < CAP; k0 += NR_WAVE) {
    const int k = k0 + lane;
    const bool occ = k < CAP && keys[k] != kEmptyKey;
    const unsigned long long m = __ballot(occ);
    if (occ) list[n_occ + __popcll(m & ((1ull << lane) - 1ull))] = k;
    n_occ += __popcll(m);
  }
  wave_fence();
#pragma unroll 1
  for (int kk = lane; kk < n_occ * 8 * F; kk += NR_WAVE) {
    const int e = kk / (8 * F), r = kk - e * (8 * F);
    const int corner = r / F, f = r - corner * F;
    const int slot = list[e];
    const unsigned long long key = keys[slot];
    const int cx = ((int)((uint32_t)(key & 0x1FFFFF) << 11)) >> 11;
    const int cy = ((int)((uint32_t)((key >> 21) & 0x1FFFFF) << 11)) >> 11;
    const int cz = ((int)((uint32_t)((key >> 42) & 0x1FFFFF) << 11)) >> 11;
    const uint32_t hs = nr_hash3(cx + (corner & 1), cy + ((corner >> 1) & 1), cz + ((corner >> 2) & 1), mask);
    const float t = vals[slot * 8 * F + r];
    if (t != 0.0f) {
      unsafeAtomicAdd(base + (int64_t)hs * F + f, t);
      if (seen != nullptr) seen[((int64_t)hs * F + f) >> 2] = 1;
    }
    if (reset) vals[slot * 8 * F + r] = 0.0f;
  }
  if (reset) {
    wave_fence();
    for (int e = lane; e < n_occ; e += NR_WAVE) keys[list[e]] = kEmptyKey;
    wave_fence();
  }
}

template <int F, int CHUNK, int CAP, int W>
__global__ void __launch_bounds__(W * 64)
hash_encode_bwd_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int log2T,
           const float* __restrict__ gout, int64_t sn, int64_t sl, float* __restrict__ gtable, int64_t n, int S,
           uint8_t* __restrict__ seen_all) {
  constexpr int NV = 8 * F;
  __shared__ unsigned long long s_key[W][CAP];
  __shared__ float s_val[W][CAP * NV];
  __shared__ int s_owner[W][CAP];
  const int level = blockIdx.y;
  const int lane = nr_lane(), wave = threadIdx.x >> 6;
  unsigned long long* keys = s_key[wave];
  float* vals = s_val[wave];
  int* owner = s_owner[wave];
  const float scale = scalings[level];
  const uint32_t mask = (1u << log2T) - 1u;
  float* base = gtable + (((int64_t)level << log2T) * F);
  uint8_t* seen = seen_all != nullptr ? seen_all + ((((int64_t)level << log2T) * F) >> 2) : nullptr;
  for (int k = lane; k < CAP; k += NR_WAVE) keys[k] = kEmptyKey;
  for (int k = lane; k < CAP * NV; k += NR_WAVE) vals[k] = 0.0f;
  wave_fence();

  const int64_t chunk0 = ((int64_t)blockIdx.x * W + wave) * CHUNK;
  if (chunk0 >= n) return;
  // storage index of this lane's sample, advanced incrementally: thread i = s * B + b reads sample b * S + s
  const int64_t B = S > 0 ? n / S : 0;
  int64_t idx;
  int64_t rb = 0;  // ray index of the lane's current sample (sample-major walk only)
  if (S > 0) {
    const int64_t i = chunk0 + lane;
    rb = i % B;
    idx = rb * S + i / B;
  } else {
    idx = chunk0 + lane;
  }
  const float* gl = gout + (int64_t)level * sl;
  float nx[3] = {0.0f, 0.0f, 0.0f}, nstd = 0.0f, ng[F];
#pragma unroll
  for (int f = 0; f < F; ++f) ng[f] = 0.0f;
  auto fetch = [&](int64_t i) {
    if (i < n) {
#pragma unroll
      for (int a = 0; a < 3; ++a) nx[a] = x[idx * 3 + a];
      if (std != nullptr) nstd = std[idx];
#pragma unroll
      for (int f = 0; f < F; ++f) ng[f] = gl[idx * sn + f];
    }
  };
  auto advance = [&]() {
    if (S > 0) {
      rb += NR_WAVE;
      idx += (int64_t)NR_WAVE * S;
      while (rb >= B) {  // wrapped past the last ray: next sample slot
        rb -= B;
        idx -= B * S - 1;
      }
    } else {
      idx += NR_WAVE;
    }
  };
  fetch(chunk0 + lane);
  int fill = 0;
#pragma unroll 1
  for (int64_t i = chunk0 + lane; i < chunk0 + CHUNK; i += NR_WAVE) {
    const bool valid = i < n;
    float cx[3], cstd, cg[F];
#pragma unroll
    for (int a = 0; a < 3; ++a) cx[a] = nx[a];
    cstd = nstd;
#pragma unroll
    for (int f = 0; f < F; ++f) cg[f] = ng[f];
    advance();
    if (i + NR_WAVE < chunk0 + CHUNK) fetch(i + NR_WAVE);
    int lo[3];
    float v[8][F];
    if (valid) {
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
      float g[F];
#pragma unroll
      for (int f = 0; f < F; ++f) g[f] = cg[f] * r;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
        const float w = (hx ? cw[0] : 1.0f - cw[0]) * (hy ? cw[1] : 1.0f - cw[1]) * (hz ? cw[2] : 1.0f - cw[2]);
#pragma unroll
        for (int f = 0; f < F; ++f) v[corner][f] = g[f] * w;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) lo[a] = INT_MIN + lane;
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
#pragma unroll
        for (int f = 0; f < F; ++f) v[corner][f] = 0.0f;
    }
    auto same_cell = [&](int lx, int ly, int lz) { return lx == lo[0] && ly == lo[1] && lz == lo[2]; };
    {
      const bool same = same_cell(nr_xor32_i(lo[0]), nr_xor32_i(lo[1]), nr_xor32_i(lo[2]));
      if (__ballot(same) != 0ull) {  // (uniform: no exchange where no lane shares its cell with lane ^ 32)
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const float o = nr_xor32_f(v[corner][f]);
            if (same) v[corner][f] = lane < 32 ? v[corner][f] + o : 0.0f;
          }
      }
    }
    const bool head = lane == 0 || !same_cell(nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[0]),
                                              nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[1]),
                                              nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[2]));
    int flag = head ? 1 : 0;
    auto scan_step = [&](auto ctrl, auto rowmask) {  // (nr_common.h: two vector instructions per element and in-row step)
      nr_seg_scan_step<decltype(ctrl)::value, decltype(rowmask)::value, 8 * F>(&v[0][0], flag, lane);
    };
    // a step changes nothing once every lane's partial sum has reached its run's head (flag = 1 everywhere: take = 0), so
    // the scan stops there -- no step at all where every lane is a cell of its own (the fine levels), all seven only where
    // a run crosses a 16-lane row.  Same sums bit for bit; 2 x 8 x F vector instructions per skipped step.
    do {
      if (__ballot(flag == 0) == 0ull) break;
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
    const int next_head = nr_dpp_i<NR_DPP_WAVE_SHL1, 0xF>(1, head ? 1 : 0);
    uint32_t any_bits = 0u;  // "some |v| is not zero" from the values' bits: one three-input OR per two values instead of an add each
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) any_bits |= __float_as_uint(v[corner][f]);
    const bool nz = (any_bits & 0x7FFFFFFFu) != 0u;
    const bool want = (lane == NR_WAVE - 1 || next_head) && nz;
    {  // make room BEFORE inserting: with <= 3/4 load the probes (almost) always succeed
      const int need = __popcll(__ballot(want));
      if (fill + need > CAP * 3 / 4 && fill > 0) {
        flush_table<F, CAP>(keys, vals, owner, base, mask, lane, true, seen);
        fill = 0;
      }
    }
    const unsigned long long key = pack_cell(lo);
    uint32_t s0 = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (CAP - 1);
    int slot = -1;
    bool fresh = false;
#pragma unroll 1
    for (int probe = 0; probe < 4; ++probe) {
      const bool searching = want && slot < 0;
      if (!__any(searching)) break;
      const bool was_empty = searching && keys[s0] == kEmptyKey;
      if (was_empty) keys[s0] = key;
      wave_fence();
      if (searching) {
        if (keys[s0] == key) { slot = (int)s0; fresh = was_empty; } else s0 = (s0 + 1) & (CAP - 1);
      }
      wave_fence();
    }
    fill += __popcll(__ballot(fresh));  // upper bound: lanes of one new cell may each count it
    // 3b. lanes sharing a slot take turns (owner word), everyone else adds concurrently: plain RMW
    bool pending = slot >= 0;
#pragma unroll 1
    while (__any(pending)) {
      if (pending) owner[slot] = lane;
      wave_fence();
      if (pending && owner[slot] == lane) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
#pragma unroll
          for (int f = 0; f < F; ++f) vals[slot * NV + corner * F + f] += v[corner][f];
        pending = false;
      }
      wave_fence();
    }
    const bool spill = want && slot < 0;
    if (__any(spill)) {
      if (spill) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
          const uint32_t hs = nr_hash3(lo[0] + (corner & 1), lo[1] + ((corner >> 1) & 1), lo[2] + ((corner >> 2) & 1), mask);
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (v[corner][f] != 0.0f) {
              unsafeAtomicAdd(base + (int64_t)hs * F + f, v[corner][f]);
              if (seen != nullptr) seen[((int64_t)hs * F + f) >> 2] = 1;
            }
        }
      }
      fill = CAP;  // crowded around some hash: make room
    }
  }
  wave_fence();
  flush_table<F, CAP>(keys, vals, owner, base, mask, lane, false, seen);
}

// Gradient w.r.t. the input positions (needed only where positions depend on parameters: samples
// inside dynamic-actor boxes, whose box-frame coordinates depend on the learnable trajectories;
// field_components/neurad_encoding.py:176,205-207).  d out / d x_a = scale_l * d interp / d w_a:
// floor/ceil are piecewise constant, the interpolation weight w = p - floor(p) has slope 1.
template <int F>
__global__ void __launch_bounds__(256)
hash_encode_bwd_input_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ table,
                             const float* __restrict__ scalings, int L, int log2T, const float* __restrict__ gout,
                             int64_t sn, int64_t sl, float* __restrict__ gx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask = (1u << log2T) - 1u;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int level = 0; level < L; ++level) {
    const float scale = scalings[level];
    const Corner c = make_corner(x, i, scale);
    const float* base = table + (((int64_t)level << log2T) * F);
    float r = scale;
    if (std != nullptr) r = scale / fmaxf(scale * 2.0f * std[i], 1.0f);
    float g[F];
    const float* gi = gout + i * sn + (int64_t)level * sl;
#pragma unroll
    for (int f = 0; f < F; ++f) g[f] = gi[f];
    // t[corner] = <table[corner], g>
    float t[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const bool hx = corner & 1, hy = corner & 2, hz = corner & 4;
      float v[F];
      load_entry<F>(base + (int64_t)nr_hash3(hx ? c.hi[0] : c.lo[0], hy ? c.hi[1] : c.lo[1], hz ? c.hi[2] : c.lo[2], mask) * F, v);
      float d = 0.0f;
#pragma unroll
      for (int f = 0; f < F; ++f) d += v[f] * g[f];
      t[corner] = d;
    }
    const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
    // corner index bits: 1 = x hi, 2 = y hi, 4 = z hi
    const float dx = ((t[7] - t[6]) * wy + (t[5] - t[4]) * (1.0f - wy)) * wz + ((t[3] - t[2]) * wy + (t[1] - t[0]) * (1.0f - wy)) * (1.0f - wz);
    const float dy = ((t[7] - t[5]) * wx + (t[6] - t[4]) * (1.0f - wx)) * wz + ((t[3] - t[1]) * wx + (t[2] - t[0]) * (1.0f - wx)) * (1.0f - wz);
    const float dz = ((t[7] - t[3]) * wx + (t[6] - t[2]) * (1.0f - wx)) * wy + ((t[5] - t[1]) * wx + (t[4] - t[0]) * (1.0f - wx)) * (1.0f - wy);
    acc[0] += dx * r; acc[1] += dy * r; acc[2] += dz * r;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) gx[i * 3 + a] = acc[a];
}

__global__ void __launch_bounds__(256)
contract_gaussians_kernel(const float* __restrict__ origins, const float* __restrict__ directions,
                          const float* __restrict__ pixel_area, const float* __restrict__ edges, int64_t n_rays,
                          int S, float scale, int sample_major, float* __restrict__ x01, float* __restrict__ std01) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output row
  if (i >= n_rays * S) return;
  const NrRowMap rm = nr_row_map(i, n_rays * S, S, sample_major);  // sample_major = rays stored sample-major
  const int64_t b = rm.ray;
  const int s = (int)(rm.out - b * S);
  float x[3], sd;
  nr_contract_sample(origins + b * 3, directions + b * 3, pixel_area[b], edges[b * (S + 1) + s], edges[b * (S + 1) + s + 1],
                     scale, x, sd);
#pragma unroll
  for (int a = 0; a < 3; ++a) x01[i * 3 + a] = x[a];
  std01[i] = sd;
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

extern "C" int nr_prop_field_fwd(const float* x, const float* std, const float* table, const float* scalings, int L, int F,
                                 int log2T, const float* w, float* feats, int64_t sn, int64_t sl, int64_t n, int n_samples,
                                 int rows_sample_major, float* density, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !table || !scalings || !w || !feats || !density || L < 1 || log2T < 1 || log2T > 30 || n < 0 || n_samples < 1 ||
      n % n_samples != 0 || rows_sample_major < 0 || rows_sample_major > n / n_samples)
    return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n, 256)), block(256);
#define CALL(FF) hipLaunchKernelGGL(prop_field_fwd_kernel<FF>, grid, block, 0, nr_s(stream), x, std, table, scalings, L, log2T, w, feats, sn, sl, n, n_samples, rows_sample_major, density)
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

// wave_cells: cells of a wave's private table; 0 = the default of the feature width.  F = 4 knows a second, wide
// configuration (512 samples per wave, 256 cells, one wave per block) for batches with incoherent rows: it issues fewer
// atomics per sample and leaves more of the chip to the kernels that run beside it (mixed batch: step -3 % fresh, -6 %
// after 1 500 steps; camera-only 16 384 rays +3 %, which is why it is the caller's choice).
static int hash_encode_bwd_launch(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                  const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, int sample_major,
                                  int wave_cells, uint8_t* seen, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !gout || !scalings || !gtable || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  if (sample_major > 0 && n % sample_major != 0) return NR_EINVAL;
  if (wave_cells != 0 && wave_cells != 128 && wave_cells != 256) return NR_EINVAL;
#define LAUNCH(FF, CHUNK, CAP, W)                                                                                  \
  {                                                                                                               \
    dim3 grid((unsigned)nr_cdiv(n, (int64_t)(W) * (CHUNK)), (unsigned)L), block((W) * 64);                         \
    hipLaunchKernelGGL((hash_encode_bwd_kernel<FF, CHUNK, CAP, W>), grid, block, 0, nr_s(stream), x, std, scalings, \
                       log2T, gout, sn, sl, gtable, n, sample_major, seen);                                       \
  }
#define LAUNCH_X(...) LAUNCH(__VA_ARGS__)
#define CALL(FF) LAUNCH(FF, BwdCfg<FF>::CHUNK, BwdCfg<FF>::CAP, BwdCfg<FF>::W)
  switch (F) {
    case 1: CALL(1) break;
    case 2: CALL(2) break;
    case 4:
      if (wave_cells == 256) LAUNCH_X(4, NR_BWD_F4_WIDE) else CALL(4)
      break;
    case 8: CALL(8) break;
    default: return NR_EINVAL;
  }
#undef CALL
#undef LAUNCH_X
#undef LAUNCH
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_hash_encode_bwd_tuned(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                        const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n,
                                        int sample_major, int wave_cells, nr_stream_t stream) {
  return hash_encode_bwd_launch(x, std, scalings, L, F, log2T, gout, sn, sl, gtable, n, sample_major, wave_cells, nullptr, stream);
}

extern "C" int nr_hash_encode_bwd_marked(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                         const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, int sample_major,
                                         int wave_cells, unsigned char* seen_grad, nr_stream_t stream) {
  if (!seen_grad) return NR_EINVAL;
  return hash_encode_bwd_launch(x, std, scalings, L, F, log2T, gout, sn, sl, gtable, n, sample_major, wave_cells, seen_grad, stream);
}

extern "C" int nr_hash_encode_bwd(const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                                  const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n,
                                  int sample_major, nr_stream_t stream) {
  return nr_hash_encode_bwd_tuned(x, std, scalings, L, F, log2T, gout, sn, sl, gtable, n, sample_major, 0, stream);
}

extern "C" int nr_hash_encode_bwd_input(const float* x, const float* std, const float* table, const float* scalings, int L,
                                        int F, int log2T, const float* gout, int64_t sn, int64_t sl, float* gx, int64_t n,
                                        nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !table || !scalings || !gout || !gx || L < 1 || log2T < 1 || log2T > 30 || n < 0) return NR_EINVAL;
  dim3 grid((unsigned)nr_cdiv(n, 256)), block(256);
  switch (F) {
    case 1: hipLaunchKernelGGL(hash_encode_bwd_input_kernel<1>, grid, block, 0, nr_s(stream), x, std, table, scalings, L, log2T, gout, sn, sl, gx, n); break;
    case 2: hipLaunchKernelGGL(hash_encode_bwd_input_kernel<2>, grid, block, 0, nr_s(stream), x, std, table, scalings, L, log2T, gout, sn, sl, gx, n); break;
    case 4: hipLaunchKernelGGL(hash_encode_bwd_input_kernel<4>, grid, block, 0, nr_s(stream), x, std, table, scalings, L, log2T, gout, sn, sl, gx, n); break;
    case 8: hipLaunchKernelGGL(hash_encode_bwd_input_kernel<8>, grid, block, 0, nr_s(stream), x, std, table, scalings, L, log2T, gout, sn, sl, gx, n); break;
    default: return NR_EINVAL;
  }
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_contract_gaussians(const float* origins, const float* directions, const float* pixel_area,
                                     const float* edges, int64_t n_rays, int S, float scale, int sample_major, float* x01,
                                     float* std01, nr_stream_t stream) {
  if (n_rays == 0) return 0;
  if (!origins || !directions || !pixel_area || !edges || !x01 || !std01 || S < 1 || n_rays < 0 || !(scale > 0)) return NR_EINVAL;
  const int64_t n = n_rays * S;
  hipLaunchKernelGGL(contract_gaussians_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), origins,
                     directions, pixel_area, edges, n_rays, S, scale, sample_major, x01, std01);
  NR_LAUNCH_CHECK();
  return 0;
}
