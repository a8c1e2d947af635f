// Development lab for the hash-grid scatter kernel: candidate designs behind one C entry so a single
// GPU session can time them against the production kernel on the bench's real sample positions.
// Not part of the product library.
#include <limits.h>

#include <type_traits>

#include "../neuradar_amd/csrc/nr_common.h"

namespace {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kEmptyVert = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned long long pack_cell(const int* lo) {
  return ((unsigned long long)(uint32_t)lo[0] & 0x1FFFFFull) | (((unsigned long long)(uint32_t)lo[1] & 0x1FFFFFull) << 21) |
         (((unsigned long long)(uint32_t)lo[2] & 0x1FFFFFull) << 42);
}
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int64_t sample_of_thread(int64_t i, int64_t n, int S) {
  if (S <= 0) return i;
  const int64_t B = n / S;
  const int64_t b = i % B, s = i / B;
  return b * S + s;
}

// Wave-collective add of (slot hs, F values) into the wave's vertex table; `active` lanes take part.
template <int F, int VCAP, int OCAP>
__device__ __forceinline__ void vertex_add(uint32_t* vkeys, float* vvals, int* owner, float* base, uint32_t hs,
                                           const float (&val)[F], bool active, int lane, int flags) {
  uint32_t s0 = (hs * 2654435761u) >> (32 - __builtin_ctz(VCAP));
  int slot = -1;
#pragma unroll 1
  for (int probe = 0; probe < 8; ++probe) {
    const bool searching = active && slot < 0;
    if (!__any(searching)) break;
    if (searching && vkeys[s0] == kEmptyVert) vkeys[s0] = hs;
    wave_fence();
    if (searching) {
      if (vkeys[s0] == hs) slot = (int)s0; else s0 = (s0 + 1) & (VCAP - 1);
    }
    wave_fence();
  }
  bool pending = slot >= 0;
#pragma unroll 1
  while (__any(pending)) {
    if (pending) owner[slot & (OCAP - 1)] = lane;
    wave_fence();
    if (pending && owner[slot & (OCAP - 1)] == lane) {
#pragma unroll
      for (int f = 0; f < F; ++f) vvals[slot * F + f] += val[f];
      pending = false;
    }
    wave_fence();
  }
  if (active && slot < 0 && !(flags & 1)) {
#pragma unroll
    for (int f = 0; f < F; ++f)
      if (val[f] != 0.0f) unsafeAtomicAdd(base + (int64_t)hs * F + f, val[f]);
  }
}

// v8: production design + (a) all inputs of iteration i+1 requested before iteration i is processed,
// (b) no division in the loop, (c) the table is flushed when it fills up instead of spilling every
// later contribution straight to memory.  flags bit0: skip global atomics (timing ablation).
// v9 flush: lane = (entry, corner, f) with f fastest -> the F floats of a corner are F adjacent lanes of ONE
// atomic wave-instruction, which leaves the L2 as one request (a lane's own consecutive instructions never merge)
template <int F, int CAP>
__device__ __forceinline__ void flush_table_v9(unsigned long long* keys, float* vals, int* list, float* base, uint32_t mask,
                                               int lane, int flags, bool reset) {
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
    if (t != 0.0f && !(flags & 1)) unsafeAtomicAdd(base + (int64_t)hs * F + f, t);
    if (reset) vals[slot * 8 * F + r] = 0.0f;
  }
  if (reset) {
    wave_fence();
    for (int e = lane; e < n_occ; e += NR_WAVE) keys[list[e]] = kEmptyKey;
    wave_fence();
  }
}

template <int F, int CAP>
__device__ __forceinline__ void flush_table(unsigned long long* keys, float* vals, float* base, uint32_t mask, int lane,
                                            int flags, bool reset) {
#pragma unroll 1
  for (int k = lane; k < CAP * 8; k += NR_WAVE) {
    const int slot = k >> 3, corner = k & 7;
    const unsigned long long key = keys[slot];
    if (key == kEmptyKey) continue;
    const int cx = ((int)((uint32_t)(key & 0x1FFFFF) << 11)) >> 11;
    const int cy = ((int)((uint32_t)((key >> 21) & 0x1FFFFF) << 11)) >> 11;
    const int cz = ((int)((uint32_t)((key >> 42) & 0x1FFFFF) << 11)) >> 11;
    const uint32_t hs = nr_hash3(cx + (corner & 1), cy + ((corner >> 1) & 1), cz + ((corner >> 2) & 1), mask);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const float t = vals[k * F + f];
      if (t != 0.0f && !(flags & 1)) unsafeAtomicAdd(base + (int64_t)hs * F + f, t);
      if (reset) vals[k * F + f] = 0.0f;
    }
  }
  if (reset) {
    wave_fence();
    for (int k = lane; k < CAP; k += NR_WAVE) keys[k] = kEmptyKey;
    wave_fence();
  }
}

template <int F, int CHUNK, int CAP, int W, bool LDSATOMIC = false, bool V9 = false>
__global__ void __launch_bounds__(W * 64)
scatter_v8(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ scalings, int log2T,
           const float* __restrict__ gout, int64_t sn, int64_t sl, float* __restrict__ gtable, int64_t n, int S, int flags) {
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
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float o = nr_xor32_f(v[corner][f]);
          if (same) v[corner][f] = lane < 32 ? v[corner][f] + o : 0.0f;
        }
    }
    const bool head = lane == 0 || !same_cell(nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[0]),
                                              nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[1]),
                                              nr_dpp_i<NR_DPP_WAVE_SHR1, 0xF>(INT_MIN, lo[2]));
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
    const int next_head = nr_dpp_i<NR_DPP_WAVE_SHL1, 0xF>(1, head ? 1 : 0);
    float mag = 0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner)
#pragma unroll
      for (int f = 0; f < F; ++f) mag += fabsf(v[corner][f]);
    const bool nz = mag != 0.0f;
    const bool want = (lane == NR_WAVE - 1 || next_head) && nz;
    if constexpr (V9) {  // make room BEFORE inserting: with <= 3/4 load the probes (almost) always succeed
      const int need = __popcll(__ballot(want));
      if (fill + need > CAP * 3 / 4 && fill > 0) {
        flush_table_v9<F, CAP>(keys, vals, owner, base, mask, lane, flags, true);
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
    if constexpr (LDSATOMIC) {
      if (slot >= 0) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner)
#pragma unroll
          for (int f = 0; f < F; ++f) unsafeAtomicAdd(&s_val[wave][slot * NV + corner * F + f], v[corner][f]);
      }
      wave_fence();
    } else {
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
    }
    const bool spill = want && slot < 0;
    if (__any(spill)) {
      if (spill && !(flags & 1)) {
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
          const uint32_t hs = nr_hash3(lo[0] + (corner & 1), lo[1] + ((corner >> 1) & 1), lo[2] + ((corner >> 2) & 1), mask);
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (v[corner][f] != 0.0f) unsafeAtomicAdd(base + (int64_t)hs * F + f, v[corner][f]);
        }
      }
      fill = CAP;  // crowded around some hash: make room
    }
    if constexpr (!V9) {
      if (fill >= CAP * 3 / 4 && i + NR_WAVE < chunk0 + CHUNK) {
        flush_table<F, CAP>(keys, vals, base, mask, lane, flags, true);
        fill = 0;
      }
    }
  }
  wave_fence();
  if constexpr (V9) flush_table_v9<F, CAP>(keys, vals, owner, base, mask, lane, flags, false);
  else flush_table<F, CAP>(keys, vals, base, mask, lane, flags, false);
}

template <int F, int CHUNK, int CAP, int W, bool LDSATOMIC = false, bool V9 = false>
int launch_v8(const float* x, const float* std, const float* scalings, int L, int log2T, const float* gout, int64_t sn,
              int64_t sl, float* gtable, int64_t n, int S, int flags, hipStream_t stream) {
  dim3 grid((unsigned)nr_cdiv(n, (int64_t)W * CHUNK), (unsigned)L), block(W * 64);
  hipLaunchKernelGGL((scatter_v8<F, CHUNK, CAP, W, LDSATOMIC, V9>), grid, block, 0, stream, x, std, scalings, log2T, gout, sn, sl,
                     gtable, n, S, flags);
  return (int)hipGetLastError();
}

}  // namespace

// variant: 100*chunk_code + 10*table_code + waves_code
extern "C" int lab_scatter(int variant, const float* x, const float* std, const float* scalings, int L, int F, int log2T,
                           const float* gout, int64_t sn, int64_t sl, float* gtable, int64_t n, int S, int flags,
                           void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define CASE(id, f, chunk, cap, w, v9) \
  if (variant == id && F == f) return launch_v8<f, chunk, cap, w, false, v9>(x, std, scalings, L, log2T, gout, sn, sl, gtable, n, S, flags, st);
  CASE(0, 1, 512, 256, 4, false)   // production
  CASE(1, 1, 512, 256, 4, true)
  CASE(2, 1, 512, 128, 4, true)
  CASE(0, 2, 512, 128, 4, false)   // production
  CASE(1, 2, 512, 128, 4, true)
  CASE(2, 2, 512, 256, 2, true)
  CASE(0, 4, 512, 64, 4, false)    // production
  CASE(1, 4, 512, 64, 4, true)
  CASE(2, 4, 512, 128, 2, true)
  CASE(3, 4, 256, 128, 2, true)
#undef CASE
  return -1;
}
