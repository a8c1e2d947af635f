// Proposal-field head (Linear(L*F,1,bias=False) + trunc_exp), SH degree-4 encoding, dense Adam.
#include <stdlib.h>

#include "nr_common.h"
#include "sh4.h"

namespace {

__device__ __forceinline__ float feat_at(const float* feats, int64_t i, int k, int64_t sn, int64_t sl, int F) {
  return feats[i * sn + (int64_t)(k / F) * sl + (k % F)];
}

__global__ void __launch_bounds__(256)
prop_density_fwd_kernel(const float* __restrict__ feats, int64_t sn, int64_t sl, int F, const float* __restrict__ w,
                        int in_dim, int64_t n, int S, int rows_sample_major, float* __restrict__ density) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = 0.0f;
  for (int k = 0; k < in_dim; ++k) x += feat_at(feats, i, k, sn, sl, F) * w[k];
  density[nr_row_map(i, n, S, rows_sample_major).out] = expf(x);  // trunc_exp forward (activations.py:33-35)
}

constexpr int kPropMaxIn = 64;

// IN = compile-time in_dim (0: runtime, up to kPropMaxIn): the weight-gradient partials stay in
// registers over the thread's whole grid-stride loop and are reduced once per block.
template <int IN>
__global__ void __launch_bounds__(256)
prop_density_bwd_kernel(const float* __restrict__ feats, int64_t sn, int64_t sl, int F, const float* __restrict__ w,
                        int in_dim_rt, int64_t n, int S, int rows_sample_major, const float* __restrict__ g_density,
                        float* __restrict__ g_feats, float* __restrict__ g_w) {
  constexpr int NI = IN > 0 ? IN : kPropMaxIn;
  const int in_dim = IN > 0 ? IN : in_dim_rt;
  __shared__ float s_gw[4][NI];
  float wk[NI], part[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    wk[k] = k < in_dim ? w[k] : 0.0f;
    part[k] = 0.0f;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float f[NI], x = 0.0f;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      f[k] = k < in_dim ? feat_at(feats, i, k, sn, sl, F) : 0.0f;
      x += f[k] * wk[k];
    }
    const float g = g_density[nr_row_map(i, n, S, rows_sample_major).out] * expf(fminf(fmaxf(x, -15.0f), 15.0f));  // activations.py:38-41
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if (k < in_dim) g_feats[i * sn + (int64_t)(k / F) * sl + (k % F)] = g * wk[k];
      part[k] += g * f[k];
    }
  }
  const int lane = nr_lane(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const float t = nr_wave_sum(part[k]);
    if (lane == 0) s_gw[wave][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < in_dim)
    unsafeAtomicAdd(g_w + threadIdx.x, s_gw[0][threadIdx.x] + s_gw[1][threadIdx.x] + s_gw[2][threadIdx.x] + s_gw[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256)
sh4_kernel(const float* __restrict__ dirs, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sh[16];
  nr_sh4(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], sh);
  float4* o = reinterpret_cast<float4*>(out + i * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = make_float4(sh[q * 4], sh[q * 4 + 1], sh[q * 4 + 2], sh[q * 4 + 3]);
}

// Dense Adam / AdamW, torch.optim semantics (single tensor, no amsgrad/maximize), grad zeroed in
// the same pass: 4 reads + 4 writes of 4 B per parameter, pure HBM streaming (float4 per lane).
//
// Exact shortcuts for hash tables: an entry with grad == 0, exp_avg == 0 and exp_avg_sq == 0 is a fixed
// point of Adam without weight decay (update = lr * 0 / (0 + eps) = 0, moments stay 0).  A lane whose four
// entries are all at that point reads (g, m, v) only -- no parameter read, no stores: 12 instead of 32 bytes per
// parameter wherever neighbouring lanes skip together, i.e. in the never-touched rows of the coarse levels (a level
// of resolution R has (R+1)^3 vertices for T rows: for R < ~80 at T = 2^19 most rows stay untouched for the whole
// training) and in most of every level early on.  A gradient that is already zero is not zeroed again (28 B).  With the caller's `seen_grad` bytes (one per
// four parameters, zero-initialised together with the moments; set here the first time a gradient arrives) a group
// that never had a gradient is recognised from its byte and its gradient alone: 4 B.  Bit-identical to the dense update.
// Measured on the 537 MB NeuRadar table: step 1.22 -> 1.04 ms; headline step -4 %.
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
            int64_t n, float lr, float beta1, float beta2, float eps, float wd, int adamw, float bc1, float bc2_sqrt,
            float grad_scale, int zero_grad, const float* __restrict__ dev_hyper, uint8_t* __restrict__ seen_grad,
            const float* __restrict__ skip, __bf16* __restrict__ delta16, int ss) {
  // ss: stride of the moments in 16-byte groups -- 1: two arrays; 2: ONE array of [exp_avg x4 | exp_avg_sq x4] records (v = m + 4
  // floats): a live group between never-touched neighbours then costs three cache lines (parameter, gradient, moments) instead of four
  if (skip != nullptr && skip[0] != 0.0f) {  // found-inf (GradScaler.step): no update; the gradient is still cleared
    if (delta16 != nullptr)  // (nothing moved: zero deltas for the replicas)
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) delta16[i] = (__bf16)0.0f;
    if (skip[0] == 2.0f) return;  // 2: the caller keeps the gradient for the next step (an overflowed row-list exchange)
    if (zero_grad) {
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4* g4z = reinterpret_cast<float4*>(grad);
      const int64_t n4z = n / 4;
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4z; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g = g4z[i];
        if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f) g4z[i] = z;  // (NaN != 0: cleared as well)
      }
      if (blockIdx.x == 0)
        for (int64_t i = n4z * 4 + threadIdx.x; i < n; i += blockDim.x) grad[i] = 0.0f;
    }
    return;
  }
  if (dev_hyper != nullptr) {  // graph-replay friendly: {lr, 1-beta1^t, sqrt(1-beta2^t)} live on the device
    lr = dev_hyper[0];
    bc1 = dev_hyper[1];
    bc2_sqrt = dev_hyper[2];
  }
  const int64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(param);
  float4* g4 = reinterpret_cast<float4*>(grad);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const float step_size = lr / bc1;
  const bool can_skip = wd == 0.0f;
  auto upd = [&](float& p, float& g, float& mm, float& vv) {
    float gr = g * grad_scale;
    if (wd != 0.0f) {
      if (adamw) p = p * (1.0f - lr * wd); else gr = gr + wd * p;
    }
    mm = mm + (gr - mm) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1-beta1)
    vv = vv * beta2 + (1.0f - beta2) * gr * gr;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p = p - step_size * (mm / denom);
    if (zero_grad) g = 0.0f;
  };
#ifndef NR_ADAM_TEMPORAL  // the 4.3-GB stream of the NeuRadar table bypasses L2 retention: what the scatters and the
  //                         sampling rounds beside / after it keep there survives (step -1.5 % fresh, -3 % after 1 500 steps)
  typedef float f4 __attribute__((ext_vector_type(4)));
  auto ld = [](const float4* q) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(q)); return make_float4(t.x, t.y, t.z, t.w); };
  auto stv = [](float4* q, float4 x) { f4 t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4*>(q)); };
#else
  auto ld = [](const float4* q) { return *q; };
  auto stv = [](float4* q, float4 x) { *q = x; };
#endif
  // Memory-level parallelism: a thread walks kU groups of four parameters per trip.  Their gradients and flags are requested
  // together, then the moments AND the parameters of the live ones together -- two dependent round trips per kU groups (the
  // first cut took three per group: g -> (m, v) -> p; PMC: 3.7 TB/s of HBM-side traffic, 60 % of what streams reach).
#ifndef NR_ADAM_U
#define NR_ADAM_U 1
#endif
  constexpr int kU = NR_ADAM_U;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * kU) {
    float4 g[kU];
    uint8_t seen[kU];
    bool live[kU], had[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = i0 + k * stride;
      const bool in = i < n4;
      g[k] = in ? ld(g4 + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      seen[k] = (in && seen_grad != nullptr) ? seen_grad[i] : (uint8_t)1;
      live[k] = in;
    }
    float4 mm[kU], vv[kU], pp[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = i0 + k * stride;
      had[k] = g[k].x != 0.0f || g[k].y != 0.0f || g[k].z != 0.0f || g[k].w != 0.0f;
      if (delta16 != nullptr && live[k]) reinterpret_cast<uint2*>(delta16)[i] = make_uint2(0u, 0u);  // (rewritten below if the group moves)
      // never had a gradient: m = v = 0 without reading them, the update is the identity (4 B/param)
      if (can_skip && !had[k] && seen[k] == 0) live[k] = false;
      if (live[k]) {
        mm[k] = ld(m4 + i * ss);
        vv[k] = ld(v4 + i * ss);
        pp[k] = ld(p4 + i);
      }
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      if (!live[k]) continue;
      const int64_t i = i0 + k * stride;
      if (can_skip && !had[k] && mm[k].x == 0.0f && mm[k].y == 0.0f && mm[k].z == 0.0f && mm[k].w == 0.0f && vv[k].x == 0.0f &&
          vv[k].y == 0.0f && vv[k].z == 0.0f && vv[k].w == 0.0f)
        continue;  // fixed point (weights restored from a checkpoint without moments): no stores
      if (seen_grad != nullptr && seen[k] == 0) seen_grad[i] = 1;
      const float4 p_old = pp[k];
      upd(pp[k].x, g[k].x, mm[k].x, vv[k].x); upd(pp[k].y, g[k].y, mm[k].y, vv[k].y);
      upd(pp[k].z, g[k].z, mm[k].z, vv[k].z); upd(pp[k].w, g[k].w, mm[k].w, vv[k].w);
      if (delta16 != nullptr) {
        // sharded data-parallel step: the replicas receive the update as a bf16 DELTA, and the owner applies the same rounded
        // delta to its own copy -- every replica computes p_old + float(delta) from identical inputs: bit-identical tables
        __bf16 d[4] = {(__bf16)(pp[k].x - p_old.x), (__bf16)(pp[k].y - p_old.y), (__bf16)(pp[k].z - p_old.z), (__bf16)(pp[k].w - p_old.w)};
        pp[k] = make_float4(p_old.x + (float)d[0], p_old.y + (float)d[1], p_old.z + (float)d[2], p_old.w + (float)d[3]);
        uint2 raw;
        __builtin_memcpy(&raw, d, 8);
        reinterpret_cast<uint2*>(delta16)[i] = raw;
      }
      stv(p4 + i, pp[k]); stv(m4 + i * ss, mm[k]); stv(v4 + i * ss, vv[k]);
      if (zero_grad && had[k]) stv(g4 + i, g[k]);  // a gradient that is already zero is not zeroed again (28 B/param)
    }
  }
  if (blockIdx.x == 0)  // (n % 4 tail: separate arrays only, checked by the entry point)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float p_old = param[i];
      upd(param[i], grad[i], m[i], v[i]);
      if (delta16 != nullptr) {
        delta16[i] = (__bf16)(param[i] - p_old);
        param[i] = p_old + (float)delta16[i];
      }
    }
}

// Sharded data-parallel table step, receiving side: p[i] += float(delta[i]) for the elements OUTSIDE this rank's own shard
// [lo, hi) (the owner's Adam launch has applied its deltas already); zero groups -- most of a table in any one step -- cost
// their 8 bytes of delta only.
__global__ void __launch_bounds__(256)
apply_delta16_kernel(float* __restrict__ p, const __bf16* __restrict__ delta, int64_t n, int64_t lo, int64_t hi,
                     uint32_t* __restrict__ generation) {
  if (blockIdx.x == 0 && threadIdx.x == 0) nr_bump_generation(generation);
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    if (i * 4 >= lo && i * 4 < hi) continue;  // (shards are whole 16-byte groups)
    const uint2 raw = reinterpret_cast<const uint2*>(delta)[i];
    if ((raw.x | raw.y) == 0u) continue;
    __bf16 d[4];
    __builtin_memcpy(d, &raw, 8);
    float4 v = reinterpret_cast<float4*>(p)[i];
    v.x += (float)d[0]; v.y += (float)d[1]; v.z += (float)d[2]; v.w += (float)d[3];
    reinterpret_cast<float4*>(p)[i] = v;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x)
      if (i < lo || i >= hi) p[i] += (float)delta[i];
}

// ... sending side of the reduce-scatter: low[i] = bf16(g[i]) and g[i] = 0 in one pass (the reduced shard comes back from the
// collective; everything else of the local gradient is spent) -- instead of a conversion pass plus a 537-MB memset
__global__ void __launch_bounds__(256)
grad_to16_clear_kernel(float* __restrict__ g, __bf16* __restrict__ low, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    uint2 raw = make_uint2(0u, 0u);
    if (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f) {
      __bf16 d[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      __builtin_memcpy(&raw, d, 8);
      reinterpret_cast<float4*>(g)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    reinterpret_cast<uint2*>(low)[i] = raw;
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      low[i] = (__bf16)g[i];
      g[i] = 0.0f;
    }
}

// The same update where the SCATTER marks `seen_grad` (nr_hash_encode_bwd_marked: a byte is set wherever a gradient sum is
// added): a never-marked group has g = m = v = 0 by construction, so it is left alone after reading its byte alone -- adam_kernel
// above still reads the group's gradient (4 B/param: 537 MB of the NeuRadar table per step) to notice a first arrival.  The
// thread's next byte is requested one trip ahead, and a marked group's g, m, v and p in ONE trip (adam_kernel: g, then m / v / p).
// No weight decay (a decayed parameter moves without a gradient).  Bit-identical to the dense update.
__global__ void __launch_bounds__(256)
adam_marked_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, int64_t n,
                   float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt, float grad_scale, int zero_grad,
                   const float* __restrict__ dev_hyper, const uint8_t* __restrict__ seen_grad, const float* __restrict__ skip, int ss) {
  if (dev_hyper != nullptr) {
    lr = dev_hyper[0];
    bc1 = dev_hyper[1];
    bc2_sqrt = dev_hyper[2];
  }
  const int64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(param);
  float4* g4 = reinterpret_cast<float4*>(grad);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  if (skip != nullptr && skip[0] != 0.0f) {  // found-inf: no update; marked groups' gradients are cleared (their moments stay
    if (zero_grad) {                         // zero: the next step finds them at the fixed point or with a fresh gradient)
      const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        if (seen_grad[i] == 0) continue;
        const float4 g = g4[i];
        if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f) g4[i] = z;
      }
    }
    return;
  }
  const float step_size = lr / bc1;
  auto upd = [&](float& p, float& g, float& mm, float& vv) {
    const float gr = g * grad_scale;
    mm = mm + (gr - mm) * (1.0f - beta1);
    vv = vv * beta2 + (1.0f - beta2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p = p - step_size * (mm / denom);
    if (zero_grad) g = 0.0f;
  };
  typedef float f4 __attribute__((ext_vector_type(4)));
  auto ld = [](const float4* q) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(q)); return make_float4(t.x, t.y, t.z, t.w); };
#ifdef NR_ADAM_TEMPORAL_STORES  // (A/B build: plain stores, which the L2 may merge into whole lines before they leave)
  auto stv = [](float4* q, float4 x) { *q = x; };
#else
  auto stv = [](float4* q, float4 x) { f4 t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4*>(q)); };
#endif
#ifndef NR_ADAMM_U
#define NR_ADAMM_U 1
#endif
  constexpr int kU = NR_ADAMM_U;  // groups per trip: their four loads each are requested together
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint8_t mark[kU];
#pragma unroll
  for (int k = 0; k < kU; ++k) mark[k] = (i0 + k * stride) < n4 ? seen_grad[i0 + k * stride] : (uint8_t)0;
  for (; i0 < n4; i0 += stride * kU) {
    uint8_t cur[kU];
    float4 g[kU], mm[kU], vv[kU], pp[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = i0 + k * stride, nx = i + stride * kU;
      cur[k] = mark[k];
      mark[k] = nx < n4 ? seen_grad[nx] : (uint8_t)0;
      if (cur[k] != 0) {
        g[k] = ld(g4 + i);
        mm[k] = ld(m4 + i * ss);
        vv[k] = ld(v4 + i * ss);
        pp[k] = ld(p4 + i);
      }
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      if (cur[k] == 0) continue;
      const int64_t i = i0 + k * stride;
      const bool had = g[k].x != 0.0f || g[k].y != 0.0f || g[k].z != 0.0f || g[k].w != 0.0f;
      if (!had && mm[k].x == 0.0f && mm[k].y == 0.0f && mm[k].z == 0.0f && mm[k].w == 0.0f && vv[k].x == 0.0f && vv[k].y == 0.0f &&
          vv[k].z == 0.0f && vv[k].w == 0.0f)
        continue;  // fixed point: no stores
      upd(pp[k].x, g[k].x, mm[k].x, vv[k].x); upd(pp[k].y, g[k].y, mm[k].y, vv[k].y);
      upd(pp[k].z, g[k].z, mm[k].z, vv[k].z); upd(pp[k].w, g[k].w, mm[k].w, vv[k].w);
      stv(p4 + i, pp[k]); stv(m4 + i * ss, mm[k]); stv(v4 + i * ss, vv[k]);
      if (zero_grad && had) stv(g4 + i, g[k]);
    }
  }
}

// ---- the marked update in TWO launches around the scatter -----------------------------------------
// The step's critical path ended scatter (700 us) -> Adam (514 us) with nothing beside the Adam.  Which groups of the table THIS
// step's gradient can reach is known long before the backward: the vertices of the step's samples, i.e. what the forward gather
// reads (nr_hash_mark_vertices stamps them with the step's epoch value as soon as the sampling rounds are done).  So:
//   PHASE 1  groups with a history (seen) that this step does NOT touch (stamp != value): their gradient is exactly 0 -- the
//            zero-gradient update (moments decay, the parameter moves) runs BESIDE the forward and the field backward, on its own
//            stream; the gradient is neither read nor written (96 instead of 128 bytes per group);
//   PHASE 2  the groups stamped this step, after the scatter: the full update.
// Every group gets exactly one update per step, with the arithmetic of adam_marked_kernel on the same operands: bit-identical to
// the single launch (tests/test_gpu_adam_split.py).  A stale stamp (the 8-bit value comes round every 255 steps; a vertex whose
// weight is 0) only sends a group to phase 2 with a zero gradient.  Phase 1 writes only parameters the step's forward does not
// read.  No loss-scaler skip (the flags are not known when phase 1 runs): the caller keeps the single launch under a scaler.
template <int PHASE>
__global__ void __launch_bounds__(256)
adam_split_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v, int64_t n,
                  float beta1, float beta2, float eps, float grad_scale, const float* __restrict__ dev_hyper,
                  const uint8_t* __restrict__ seen_grad, const uint8_t* __restrict__ stamp, const float* __restrict__ epoch, int ss) {
  const float lr = dev_hyper[0], bc1 = dev_hyper[1], bc2_sqrt = dev_hyper[2];
  const uint8_t now = (uint8_t)nr_stamp_value(epoch);
  const int64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(param);
  float4* g4 = reinterpret_cast<float4*>(grad);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const float step_size = lr / bc1;
  auto upd = [&](float& p, float g, float& mm, float& vv) {  // adam_marked_kernel's arithmetic, operand for operand
    const float gr = g * grad_scale;
    mm = mm + (gr - mm) * (1.0f - beta1);
    vv = vv * beta2 + (1.0f - beta2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p = p - step_size * (mm / denom);
  };
  typedef float f4 __attribute__((ext_vector_type(4)));
  auto ld = [](const float4* q) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(q)); return make_float4(t.x, t.y, t.z, t.w); };
  auto stv = [](float4* q, float4 x) { f4 t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4*>(q)); };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto live = [&](int64_t j) -> bool {  // (the thread's next two bytes are requested one trip ahead)
    if (j >= n4) return false;
    const uint8_t st = stamp[j];
    return PHASE == 1 ? (st != now && seen_grad[j] != 0) : st == now;
  };
  bool cur = live(i);
  for (; i < n4; i += stride) {
    const bool mine = cur;
    cur = live(i + stride);
    if (!mine) continue;
    float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (PHASE == 2) g = ld(g4 + i);
    float4 mm = ld(m4 + i * ss), vv = ld(v4 + i * ss), pp = ld(p4 + i);
    const bool had = g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f;
    if (!had && mm.x == 0.0f && mm.y == 0.0f && mm.z == 0.0f && mm.w == 0.0f && vv.x == 0.0f && vv.y == 0.0f && vv.z == 0.0f && vv.w == 0.0f)
      continue;  // fixed point: no stores
    upd(pp.x, g.x, mm.x, vv.x); upd(pp.y, g.y, mm.y, vv.y); upd(pp.z, g.z, mm.z, vv.z); upd(pp.w, g.w, mm.w, vv.w);
    stv(p4 + i, pp); stv(m4 + i * ss, mm); stv(v4 + i * ss, vv);
    if (PHASE == 2 && had) stv(g4 + i, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
  }
}

// One-thread kernel: advances the optimizer step counter and refreshes {lr, 1-beta1^t, sqrt(1-beta2^t)}
// (ExponentialDecayScheduler, engine/schedulers.py:112-143; LambdaLR applies func(k-1) to step k).
__global__ void adam_hyper_kernel(float* __restrict__ step_t, float* __restrict__ hyper, float lr, float lr_final,
                                  int warmup, int max_steps, float beta1, float beta2, const float* __restrict__ amp,
                                  int amp_group, uint32_t* __restrict__ generation) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  nr_bump_generation(generation);  // an optimizer step is about to rewrite parameters (graph replays pass here too)
  // step_t[0]: scheduler steps so far, step_t[1]: optimizer updates so far.  A step the loss scaler skipped is not counted:
  // the counters were advanced at the top of that step, before its gradients were known, so they are taken back here.
  double step = (double)step_t[0], upd = (double)step_t[1];
  if (amp != nullptr) {
    if (amp[NR_AMP_SKIPPED_PREV] != 0.0f && step > 0.0) step -= 1.0;
    if (amp[NR_AMP_FOUND_PREV + amp_group] != 0.0f && upd > 0.0) upd -= 1.0;
  }
  double cur;
  if (step < (double)warmup) {
    const double pre = 1e-8;
    double r = step / (double)warmup;
    r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
    cur = pre + ((double)lr - pre) * sin(0.5 * 3.14159265358979323846 * r);
  } else {
    const int span = max_steps - warmup > 1 ? max_steps - warmup : 1;
    double t = (step - (double)warmup) / (double)span;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    cur = exp(log((double)lr) * (1.0 - t) + log((double)lr_final) * t);
  }
  const double k = upd + 1.0;
  step_t[0] = (float)(step + 1.0);
  step_t[1] = (float)k;
  hyper[0] = (float)cur;
  hyper[1] = (float)(1.0 - pow((double)beta1, k));
  hyper[2] = (float)sqrt(1.0 - pow((double)beta2, k));
}

// ---- dynamic loss scale (GradScaler) --------------------------------------------------------------------------------
__global__ void amp_init_kernel(float* __restrict__ amp, float init_scale) {
  const int i = threadIdx.x;
  if (i < NR_AMP_FLOATS) amp[i] = i == NR_AMP_SCALE ? init_scale : (i == NR_AMP_INV_SCALE ? 1.0f / init_scale : 0.0f);
}

__global__ void amp_update_kernel(float* __restrict__ amp, int n_groups, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bool any = false;
  for (int g = 0; g < NR_AMP_MAX_GROUPS; ++g) {
    const float f = g < n_groups ? amp[NR_AMP_FOUND + g] : 0.0f;
    any |= f != 0.0f;
    amp[NR_AMP_FOUND_PREV + g] = f != 0.0f ? 1.0f : 0.0f;
    amp[NR_AMP_FOUND + g] = 0.0f;
  }
  float scale = amp[NR_AMP_SCALE], tracker = amp[NR_AMP_GROWTH_TRACKER];
  if (any) {  // torch.cuda.amp.GradScaler.update (_amp_update_scale_)
    scale *= backoff;
    tracker = 0.0f;
    amp[NR_AMP_SKIPPED_TOTAL] += 1.0f;
  } else {
    tracker += 1.0f;
    if (tracker >= (float)interval) {
      const float grown = scale * growth;
      if (grown < 3.0e38f) scale = grown;  // (torch keeps the old scale when the grown one is not finite)
      tracker = 0.0f;
    }
  }
  amp[NR_AMP_SCALE] = scale;
  amp[NR_AMP_INV_SCALE] = 1.0f / scale;
  amp[NR_AMP_GROWTH_TRACKER] = tracker;
  amp[NR_AMP_SKIPPED_PREV] = any ? 1.0f : 0.0f;
}

__global__ void __launch_bounds__(256)
nonfinite_check_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ flag) {
  bool bad = false;
  const int64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) bad |= !isfinite(x[i]);
  if (__any(bad) && nr_lane() == 0) flag[0] = 1.0f;  // (every writer stores the same value: no atomic needed)
}

template <typename E>
__global__ void __launch_bounds__(256)
unscale_add_16_kernel(float* __restrict__ dst, E* __restrict__ src, int64_t n, const float* __restrict__ inv_scale,
                      float* __restrict__ flag) {
  const float inv = inv_scale != nullptr ? inv_scale[0] : 1.0f;
  bool bad = false;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 d = reinterpret_cast<float4*>(dst)[i];
    const uint2 raw = reinterpret_cast<const uint2*>(src)[i];
    E e[4];
    __builtin_memcpy(e, &raw, 8);
    d.x = (d.x + (float)e[0]) * inv; d.y = (d.y + (float)e[1]) * inv; d.z = (d.z + (float)e[2]) * inv; d.w = (d.w + (float)e[3]) * inv;
    bad |= !(isfinite(d.x) && isfinite(d.y) && isfinite(d.z) && isfinite(d.w));
    reinterpret_cast<float4*>(dst)[i] = d;
    reinterpret_cast<uint2*>(src)[i] = make_uint2(0u, 0u);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float d = (dst[i] + (float)src[i]) * inv;
      bad |= !isfinite(d);
      dst[i] = d;
      src[i] = (E)0.0f;
    }
  if (flag != nullptr && __any(bad) && nr_lane() == 0) flag[0] = 1.0f;
}

// ---- sparse exchange of a hash table's gradient between data-parallel ranks ----------------------
// A step touches ~1 % of the main table's rows, so the ranks exchange (row, values) lists instead of
// all-reducing the dense table.  compact: every non-zero row is appended to the list (wave-aggregated
// counter) and cleared in the table; rows that do not fit `cap` stay where they are and `count` exceeds
// `cap` (the caller then falls back to the dense all-reduce).  apply: table[row] += values, one list
// at a time in rank order with plain adds -> every rank computes bit-identical sums.
// Each WAVE owns a contiguous range of rows: pass A counts its non-zero rows, ONE returning atomic per block
// reserves the block's span of the list (the waves' spans follow one another inside it), pass B re-reads the
// range and writes the rows in ascending order -- no barrier inside either pass.  (A returning atomic per wave
// and iteration -- 70 k of them on one address -- took 650 us: same-address atomics are applied one by one.)
template <int F>
__global__ void __launch_bounds__(256)
grad_compact_kernel(float* __restrict__ grad, int64_t rows, int64_t cap, int* __restrict__ idx, float* __restrict__ val,
                    int* __restrict__ count) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int lane = nr_lane(), wave = threadIdx.x >> 6;
  const int64_t per_wave = nr_cdiv_dev(nr_cdiv_dev(rows, (int64_t)gridDim.x * 4), NR_WAVE) * NR_WAVE;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * per_wave, r1 = r0 + per_wave < rows ? r0 + per_wave : rows;
  auto nonzero = [&](int64_t r, float (&v)[F]) {
    bool nz = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      v[f] = r < r1 ? grad[r * F + f] : 0.0f;
      nz |= v[f] != 0.0f;
    }
    return nz;
  };
  int mine = 0;  // wave-uniform
  for (int64_t base = r0; base < r1; base += NR_WAVE) {
    float v[F];
    mine += __popcll(__ballot(nonzero(base + lane, v)));
  }
  if (lane == 0) s_wave[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total > 0 ? atomicAdd(count, total) : 0;
  }
  __syncthreads();
  int64_t pos = s_base;
  for (int k = 0; k < wave; ++k) pos += s_wave[k];
  if (mine == 0) return;
  for (int64_t base = r0; base < r1; base += NR_WAVE) {
    const int64_t r = base + lane;
    float v[F];
    const bool nz = nonzero(r, v);
    const unsigned long long m = __ballot(nz);
    const int64_t at = pos + __popcll(m & ((1ull << lane) - 1ull));
    if (nz && at < cap) {
      idx[at] = (int)r;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        val[at * F + f] = v[f];
        grad[r * F + f] = 0.0f;
      }
    }
    pos += __popcll(m);
  }
}

template <int F>
__global__ void __launch_bounds__(256)
grad_apply_kernel(const int* __restrict__ idx, const float* __restrict__ val, const int* __restrict__ count, int64_t cap,
                  float* __restrict__ grad) {
  const int64_t n = *count < cap ? *count : cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
#pragma unroll
    for (int f = 0; f < F; ++f) grad[r * F + f] += val[i * F + f];
  }
}

// The same for the list of rank `list_rank` out of `world` gathered lists of capacity m, GUARDED by the ranks' true row counts
// (counts[r] may exceed m: rank r then could not move all its rows into its list): if ANY list overflowed, only the rank's OWN
// list is applied -- its rows go back where they came from, the local gradient is whole again -- and flag[0] = 2 tells the
// table's optimizer launch to skip this step and KEEP the gradient (nr_adam_step's skip = 2); otherwise every list is applied
// (the caller launches this once per rank, in rank order: bit-identical sums on every rank) and flag[0] = 0.  Every rank sees
// the same counts, so every rank takes the same branch: no host read, replicas stay identical.
template <int F>
__global__ void __launch_bounds__(256)
grad_apply_guarded_kernel(const int* __restrict__ idx, const float* __restrict__ val, const int* __restrict__ counts, int world,
                          int list_rank, int own_rank, int64_t m, float* __restrict__ grad, float* __restrict__ flag) {
  bool ovf = false;
  for (int r = 0; r < world; ++r) ovf = ovf || (int64_t)counts[r] > m;
  if (flag != nullptr && list_rank == 0 && blockIdx.x == 0 && threadIdx.x == 0) flag[0] = ovf ? 2.0f : 0.0f;
  if (ovf && list_rank != own_rank) return;
  const int64_t n = (int64_t)counts[list_rank] < m ? (int64_t)counts[list_rank] : m;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
#pragma unroll
    for (int f = 0; f < F; ++f) grad[r * F + f] += val[i * F + f];
  }
}

// ---- row lists to the SHARD OWNERS (sharded data-parallel table step, parallel.GradAllReducer.shard_step) ---------------
// The table's rows are owned in `world` contiguous shards of rows_per_shard rows.  compact_shards: one launch, blockIdx.y = the
// destination shard d; the non-zero rows of that shard go, as (row - d * rows_per_shard, values), into segment d of the send
// lists -- segment d starts at sum(caps[0..d)) and holds caps[d] rows -- and are cleared in the table; counts[d] = ALL non-zero
// rows of shard d (rows beyond caps[d] stay in the table).  Same two-pass scheme as grad_compact_kernel.
template <int F>
__global__ void __launch_bounds__(256)
grad_compact_shards_kernel(float* __restrict__ grad_all, int64_t rows, const int* __restrict__ caps, int* __restrict__ idx_all,
                           float* __restrict__ val_all, int* __restrict__ counts) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int d = blockIdx.y;
  int64_t off = 0;
  for (int k = 0; k < d; ++k) off += caps[k];
  const int64_t cap = caps[d];
  float* __restrict__ grad = grad_all + (int64_t)d * rows * F;
  int* __restrict__ idx = idx_all + off;
  float* __restrict__ val = val_all + off * F;
  int* __restrict__ count = counts + d;
  const int lane = nr_lane(), wave = threadIdx.x >> 6;
  const int64_t per_wave = nr_cdiv_dev(nr_cdiv_dev(rows, (int64_t)gridDim.x * 4), NR_WAVE) * NR_WAVE;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * per_wave, r1 = r0 + per_wave < rows ? r0 + per_wave : rows;
  auto nonzero = [&](int64_t r, float (&v)[F]) {
    bool nz = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      v[f] = r < r1 ? grad[r * F + f] : 0.0f;
      nz |= v[f] != 0.0f;
    }
    return nz;
  };
  int mine = 0;  // wave-uniform
  for (int64_t base = r0; base < r1; base += NR_WAVE) {
    float v[F];
    mine += __popcll(__ballot(nonzero(base + lane, v)));
  }
  if (lane == 0) s_wave[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = total > 0 ? atomicAdd(count, total) : 0;
  }
  __syncthreads();
  int64_t pos = s_base;
  for (int k = 0; k < wave; ++k) pos += s_wave[k];
  if (mine == 0) return;
  for (int64_t base = r0; base < r1; base += NR_WAVE) {
    const int64_t r = base + lane;
    float v[F];
    const bool nz = nonzero(r, v);
    const unsigned long long m = __ballot(nz);
    const int64_t at = pos + __popcll(m & ((1ull << lane) - 1ull));
    if (nz && at < cap) {
      idx[at] = (int)r;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        val[at * F + f] = v[f];
        grad[r * F + f] = 0.0f;
      }
    }
    pos += __popcll(m);
  }
}

// did ANY (source, destination) list overflow?  counts [world, world] (source-major, as all-gathered), caps [world] (per destination)
__device__ __forceinline__ bool nr_lists_overflowed(const int* __restrict__ counts, const int* __restrict__ caps, int world) {
  bool ovf = false;
  for (int k = 0; k < world * world; ++k) ovf = ovf || counts[k] > caps[k % world];
  return ovf;
}

// Owner side: the list that source rank `src` sent to this rank (`own`), added onto the rank's shard of the gradient with plain
// adds -- the caller launches it for src = 0 ... world-1 IN THAT ORDER, so the owner's sums do not depend on anything but the
// lists.  Nothing is applied if any list of the exchange overflowed (every rank sees the same counts: the same branch everywhere);
// flag[0] = 2 then makes the owner's Adam launch skip the step (nr_adam_step's skip = 2), and restore (below) puts every rank's
// own rows back into its local gradient, which the next step's scatter adds onto.
template <int F>
__global__ void __launch_bounds__(256)
grad_lists_apply_kernel(const int* __restrict__ idx, const float* __restrict__ val, const int* __restrict__ counts,
                        const int* __restrict__ caps, int world, int src, int own, float* __restrict__ shard, float* __restrict__ flag) {
  const bool ovf = nr_lists_overflowed(counts, caps, world);
  if (flag != nullptr && src == 0 && blockIdx.x == 0 && threadIdx.x == 0) flag[0] = ovf ? 2.0f : 0.0f;
  if (ovf) return;
  const int64_t n = counts[src * world + own];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
#pragma unroll
    for (int f = 0; f < F; ++f) shard[r * F + f] += val[i * F + f];
  }
}

// Sender side, overflow only: every segment of the rank's own send lists goes back where it came from (blockIdx.y = destination).
template <int F>
__global__ void __launch_bounds__(256)
grad_lists_restore_kernel(const int* __restrict__ idx_all, const float* __restrict__ val_all, const int* __restrict__ counts,
                          const int* __restrict__ caps, int world, int own, int64_t rows, float* __restrict__ grad_all,
                          const float* __restrict__ found_inf) {
  if (!nr_lists_overflowed(counts, caps, world)) return;
  if (found_inf != nullptr && found_inf[0] != 0.0f) return;  // (a gradient the loss scaler rejects is not kept: discard below)
  const int d = blockIdx.y;
  int64_t off = 0;
  for (int k = 0; k < d; ++k) off += caps[k];
  const int64_t have = counts[own * world + d], n = have < caps[d] ? have : caps[d];
  const int* __restrict__ idx = idx_all + off;
  const float* __restrict__ val = val_all + off * F;
  float* __restrict__ grad = grad_all + (int64_t)d * rows * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
#pragma unroll
    for (int f = 0; f < F; ++f) grad[r * F + f] += val[i * F + f];
  }
}

// Overflowed exchange of a step the loss scaler rejects (found-inf raised on some rank): the optimizer skips the step and CLEARS
// its gradient instead of keeping it -- so must every rank's local gradient, of which the compaction has left the rows beyond the
// lists' capacities in place (an inf / NaN loss makes every touched vertex non-zero, zero-weight corners included: the row count
// jumps exactly on such a step).  A no-op unless both conditions hold.
__global__ void __launch_bounds__(256)
grad_lists_discard_kernel(float* __restrict__ grad, int64_t n, const int* __restrict__ counts, const int* __restrict__ caps, int world,
                          const float* __restrict__ found_inf) {
  if (found_inf[0] == 0.0f || !nr_lists_overflowed(counts, caps, world)) return;
  float4* g4 = reinterpret_cast<float4*>(grad);
  const int64_t n4 = n / 4;
  const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g = g4[i];
    if (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f) g4[i] = z;  // (NaN != 0)
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) grad[i] = 0.0f;
}

}  // namespace

extern "C" int nr_grad_apply_guarded(const int* idx, const float* val, const int* counts, int world, int list_rank, int own_rank,
                                     int64_t m, int F, float* grad, float* flag, nr_stream_t stream) {
  if (!idx || !val || !counts || !grad || m < 1 || world < 1 || list_rank < 0 || list_rank >= world || own_rank < 0 || own_rank >= world)
    return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(m, 256) < 1024 ? nr_cdiv(m, 256) : 1024);
#define CALL(FF) hipLaunchKernelGGL(grad_apply_guarded_kernel<FF>, dim3(blocks), dim3(256), 0, nr_s(stream), idx, val, counts, world, \
                                    list_rank, own_rank, m, grad, flag)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_compact(float* grad, int64_t rows, int F, int64_t cap, int* idx, float* val, int* count,
                               nr_stream_t stream) {
  if (rows == 0) return 0;
  if (!grad || !idx || !val || !count || rows < 0 || rows > 0x7fffffff || cap < 1) return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(rows, 256) < 1024 ? nr_cdiv(rows, 256) : 1024);
#define CALL(FF) hipLaunchKernelGGL(grad_compact_kernel<FF>, dim3(blocks), dim3(256), 0, nr_s(stream), grad, rows, cap, idx, val, count)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_compact_shards(float* grad, int64_t rows_per_shard, int F, int world, const int* caps, int* idx, float* val,
                                      int* counts, nr_stream_t stream) {
  if (rows_per_shard == 0) return 0;
  if (!grad || !caps || !idx || !val || !counts || rows_per_shard < 0 || rows_per_shard > 0x7fffffff || world < 1 || world > 64)
    return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(rows_per_shard, 256) < 1024 ? nr_cdiv(rows_per_shard, 256) : 1024);
#define CALL(FF) hipLaunchKernelGGL(grad_compact_shards_kernel<FF>, dim3(blocks, (unsigned)world), dim3(256), 0, nr_s(stream), grad, \
                                    rows_per_shard, caps, idx, val, counts)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_lists_apply(const int* idx, const float* val, int64_t list_cap, const int* counts, const int* caps, int world,
                                   int src_rank, int own_rank, int F, float* shard, float* flag, nr_stream_t stream) {
  if (!idx || !val || !counts || !caps || !shard || list_cap < 1 || world < 1 || world > 64 || src_rank < 0 || src_rank >= world ||
      own_rank < 0 || own_rank >= world)
    return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(list_cap, 256) < 1024 ? nr_cdiv(list_cap, 256) : 1024);
#define CALL(FF) hipLaunchKernelGGL(grad_lists_apply_kernel<FF>, dim3(blocks), dim3(256), 0, nr_s(stream), idx, val, counts, caps, \
                                    world, src_rank, own_rank, shard, flag)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_lists_restore(const int* idx, const float* val, int64_t max_cap, const int* counts, const int* caps, int world,
                                     int own_rank, int64_t rows_per_shard, int F, float* grad, const float* found_inf,
                                     nr_stream_t stream) {
  if (!idx || !val || !counts || !caps || !grad || max_cap < 1 || world < 1 || world > 64 || own_rank < 0 || own_rank >= world ||
      rows_per_shard < 1 || (((uintptr_t)grad) & 15u) != 0)
    return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(max_cap, 256) < 256 ? nr_cdiv(max_cap, 256) : 256);
#define CALL(FF) hipLaunchKernelGGL(grad_lists_restore_kernel<FF>, dim3(blocks, (unsigned)world), dim3(256), 0, nr_s(stream), idx, val, \
                                    counts, caps, world, own_rank, rows_per_shard, grad, found_inf)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  if (found_inf != nullptr)
    hipLaunchKernelGGL(grad_lists_discard_kernel, dim3(4096), dim3(256), 0, nr_s(stream), grad, (int64_t)world * rows_per_shard * F, counts,
                       caps, world, found_inf);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_apply(const int* idx, const float* val, const int* count, int64_t cap, int F, float* grad,
                             nr_stream_t stream) {
  if (!idx || !val || !count || !grad || cap < 1) return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(cap, 256) < 1024 ? nr_cdiv(cap, 256) : 1024);
#define CALL(FF) hipLaunchKernelGGL(grad_apply_kernel<FF>, dim3(blocks), dim3(256), 0, nr_s(stream), idx, val, count, cap, grad)
  switch (F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    case 8: CALL(8); break;
    default: return NR_EINVAL;
  }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_adam_step_split(float* param, float* grad, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                                  float grad_scale, const float* dev_hyper, const uint8_t* seen_grad, const uint8_t* stamp,
                                  const float* epoch, int phase, int state_stride, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !grad || !m || !v || !dev_hyper || !seen_grad || !stamp || !epoch || n < 0 || (n & 3) != 0 || (phase != 1 && phase != 2))
    return NR_EINVAL;
  if (state_stride != 1 && !(state_stride == 2 && v == m + 4)) return NR_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15u) != 0) return NR_EINVAL;
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  const int64_t cap = nr_tuning().adam_blocks > 0 ? nr_tuning().adam_blocks : 4096;
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  if (phase == 1)
    hipLaunchKernelGGL(adam_split_kernel<1>, dim3(blocks), dim3(256), 0, nr_s(stream), param, grad, m, v, n, beta1, beta2, eps, grad_scale,
                       dev_hyper, seen_grad, stamp, epoch, state_stride);
  else
    hipLaunchKernelGGL(adam_split_kernel<2>, dim3(blocks), dim3(256), 0, nr_s(stream), param, grad, m, v, n, beta1, beta2, eps, grad_scale,
                       dev_hyper, seen_grad, stamp, epoch, state_stride);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_adam_hyper(float* step_t, float* hyper, float lr, float lr_final, int warmup, int max_steps,
                             float beta1, float beta2, const float* amp, int amp_group, nr_stream_t stream) {
  if (!step_t || !hyper || !(lr > 0.0f) || !(lr_final > 0.0f) || warmup < 0 || max_steps < 1) return NR_EINVAL;
  if (amp != nullptr && (amp_group < 0 || amp_group >= NR_AMP_MAX_GROUPS)) return NR_EINVAL;
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, nr_s(stream), step_t, hyper, lr, lr_final, warmup, max_steps,
                     beta1, beta2, amp, amp_group, nr_generation_ptr());
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_amp_init(float* amp, float init_scale, nr_stream_t stream) {
  if (!amp || !(init_scale > 0.0f)) return NR_EINVAL;
  hipLaunchKernelGGL(amp_init_kernel, dim3(1), dim3(64), 0, nr_s(stream), amp, init_scale);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_amp_update(float* amp, int n_groups, float growth_factor, float backoff_factor, int growth_interval,
                             nr_stream_t stream) {
  if (!amp || n_groups < 1 || n_groups > NR_AMP_MAX_GROUPS || !(growth_factor >= 1.0f) || !(backoff_factor > 0.0f) ||
      !(backoff_factor <= 1.0f) || growth_interval < 1)
    return NR_EINVAL;
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, nr_s(stream), amp, n_groups, growth_factor, backoff_factor,
                     growth_interval);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_nonfinite_check(const float* x, int64_t n, float* flag, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!x || !flag || n < 0 || ((uintptr_t)x & 15u) != 0) return NR_EINVAL;
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  hipLaunchKernelGGL(nonfinite_check_kernel, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, nr_s(stream), x, n, flag);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_unscale_add_16(float* dst, void* src, int64_t n, int src_dtype, const float* inv_scale, float* flag,
                                 nr_stream_t stream) {
  if (n == 0) return 0;
  if (!dst || !src || n < 0 || ((uintptr_t)dst & 15u) != 0 || ((uintptr_t)src & 7u) != 0) return NR_EINVAL;
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
  if (src_dtype == NR_DTYPE_F16)
    hipLaunchKernelGGL(unscale_add_16_kernel<_Float16>, dim3(blocks), dim3(256), 0, nr_s(stream), dst, static_cast<_Float16*>(src), n,
                       inv_scale, flag);
  else if (src_dtype == NR_DTYPE_BF16)
    hipLaunchKernelGGL(unscale_add_16_kernel<__bf16>, dim3(blocks), dim3(256), 0, nr_s(stream), dst, static_cast<__bf16*>(src), n,
                       inv_scale, flag);
  else
    return NR_EINVAL;
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_prop_density_fwd(const float* feats, int64_t sn, int64_t sl, int F, const float* w, int in_dim,
                                   int64_t n, int n_samples, int rows_sample_major, float* density,
                                   nr_stream_t stream) {
  if (n == 0) return 0;
  if (!feats || !w || !density || in_dim < 1 || in_dim > 64 || F < 1 || n < 0) return NR_EINVAL;
  if (rows_sample_major < 0 || (rows_sample_major && (n_samples < 1 || n % n_samples != 0 || rows_sample_major > n / n_samples))) return NR_EINVAL;
  hipLaunchKernelGGL(prop_density_fwd_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), feats, sn,
                     sl, F, w, in_dim, n, n_samples, rows_sample_major, density);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_prop_density_bwd(const float* feats, int64_t sn, int64_t sl, int F, const float* w, int in_dim,
                                   int64_t n, int n_samples, int rows_sample_major, const float* density,
                                   const float* g_density, float* g_feats, float* g_w, nr_stream_t stream) {
  (void)density;  // the clamped backward needs the pre-activation, which is recomputed from feats
  if (n == 0) return 0;
  if (!feats || !w || !g_density || !g_feats || !g_w || in_dim < 1 || in_dim > 64 || F < 1 || n < 0) return NR_EINVAL;
  if (rows_sample_major < 0 || (rows_sample_major && (n_samples < 1 || n % n_samples != 0 || rows_sample_major > n / n_samples))) return NR_EINVAL;
  // every block ends with ONE atomic request onto the same line (grad_w): keep the queue short
  const int cap_blocks = nr_tuning().pdbwd_blocks > 0 ? nr_tuning().pdbwd_blocks : 256;
  const unsigned blocks = (unsigned)(nr_cdiv(n, 256) < cap_blocks ? nr_cdiv(n, 256) : cap_blocks);
#define CALL(IN)                                                                                                       \
  hipLaunchKernelGGL(prop_density_bwd_kernel<IN>, dim3(blocks), dim3(256), 0, nr_s(stream), feats, sn, sl, F, w, in_dim, n, \
                     n_samples, rows_sample_major, g_density, g_feats, g_w)
  if (in_dim == 6) { CALL(6); } else if (in_dim == 4) { CALL(4); } else if (in_dim == 8) { CALL(8); } else { CALL(0); }
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_sh4_fwd(const float* dirs, int64_t n, float* out, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!dirs || !out || n < 0) return NR_EINVAL;
  hipLaunchKernelGGL(sh4_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), dirs, n, out);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_adam_step(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float wd, int adamw, int step, float grad_scale, int zero_grad,
                            const float* dev_hyper, uint8_t* seen_grad, const float* skip, void* delta16, int state_stride,
                            nr_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !grad || !m || !v || n < 0 || step < 1) return NR_EINVAL;
  if (state_stride != 1 && !(state_stride == 2 && (n & 3) == 0 && v == m + 4)) return NR_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15u) != 0 || ((uintptr_t)delta16 & 7u) != 0) return NR_EINVAL;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  const int64_t cap = nr_tuning().adam_blocks > 0 ? nr_tuning().adam_blocks : 4096;
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, nr_s(stream), param, grad, m, v, n, lr, beta1, beta2, eps,
                     wd, adamw, bc1, bc2_sqrt, grad_scale, zero_grad, dev_hyper, seen_grad, skip, static_cast<__bf16*>(delta16), state_stride);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_apply_delta16(float* param, const void* delta16, int64_t n, int64_t lo, int64_t hi, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !delta16 || n < 0 || lo < 0 || hi < lo || hi > n || (lo & 3) != 0 || ((hi & 3) != 0 && hi != n) ||
      ((uintptr_t)param & 15u) != 0 || ((uintptr_t)delta16 & 7u) != 0)
    return NR_EINVAL;
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  hipLaunchKernelGGL(apply_delta16_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, nr_s(stream), param,
                     static_cast<const __bf16*>(delta16), n, lo, hi, nr_generation_ptr());
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_grad_to16_clear(float* grad, void* low16, int64_t n, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!grad || !low16 || n < 0 || ((uintptr_t)grad & 15u) != 0 || ((uintptr_t)low16 & 7u) != 0) return NR_EINVAL;
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  hipLaunchKernelGGL(grad_to16_clear_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, nr_s(stream), grad,
                     static_cast<__bf16*>(low16), n);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_adam_step_marked(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                   float eps, int step, float grad_scale, int zero_grad, const float* dev_hyper,
                                   const uint8_t* seen_grad, const float* skip, int state_stride, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!param || !grad || !m || !v || !seen_grad || n < 0 || step < 1 || (n & 3) != 0) return NR_EINVAL;
  if (state_stride != 1 && !(state_stride == 2 && v == m + 4)) return NR_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15u) != 0) return NR_EINVAL;
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const int64_t want = nr_cdiv(n / 4 + 1, 256);
  const int64_t cap = nr_tuning().adam_blocks > 0 ? nr_tuning().adam_blocks : 4096;
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL(adam_marked_kernel, dim3(blocks), dim3(256), 0, nr_s(stream), param, grad, m, v, n, lr, beta1, beta2, eps, bc1,
                     bc2_sqrt, grad_scale, zero_grad, dev_hyper, seen_grad, skip, state_stride);
  NR_LAUNCH_CHECK();
  return 0;
}
