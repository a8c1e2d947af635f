// Lab: what would Adam's marked update cost if the gradient and both moments of a 4-float group
// lived in ONE 64-byte record [g | m | v | pad] instead of three arrays (+ the parameter array either way)?  Same arithmetic, same
// live pattern (a byte per group), non-temporal 16-byte accesses; live groups drawn at random with a given density, or in runs.
// build: hipcc -O3 --offload-arch=gfx950 tools/adam_layout_lab.hip -o tools/adam_layout_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ldnt(const f4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(f4* p, f4 v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void upd(f4& p, f4& g, f4& m, f4& v) {
  const float b1 = 0.9f, b2 = 0.999f, step = 1e-2f, bc2 = 0.1f, eps = 1e-15f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = m[k] + (g[k] - m[k]) * (1.0f - b1);
    v[k] = v[k] * b2 + (1.0f - b2) * g[k] * g[k];
    p[k] = p[k] - step * (m[k] / (sqrtf(v[k]) / bc2 + eps));
    g[k] = 0.0f;
  }
}

// (a) today's layout: four arrays
__global__ void __launch_bounds__(256) adam_arrays(f4* p, f4* g, f4* m, f4* v, const uint8_t* seen, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint8_t mk = i < n4 ? seen[i] : 0;
  for (; i < n4; i += stride) {
    const uint8_t cur = mk;
    mk = i + stride < n4 ? seen[i + stride] : 0;
    if (!cur) continue;
    f4 gg = ldnt(g + i), mm = ldnt(m + i), vv = ldnt(v + i), pp = ldnt(p + i);
    const bool had = gg[0] != 0.0f || gg[1] != 0.0f || gg[2] != 0.0f || gg[3] != 0.0f;
    upd(pp, gg, mm, vv);
    stnt(p + i, pp); stnt(m + i, mm); stnt(v + i, vv);
    if (had) stnt(g + i, gg);
  }
}
// (b) one 64-byte record per group: [g | m | v | pad]; parameters stay an array of their own (the forward gathers them)
__global__ void __launch_bounds__(256) adam_records(f4* p, f4* rec, const uint8_t* seen, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint8_t mk = i < n4 ? seen[i] : 0;
  for (; i < n4; i += stride) {
    const uint8_t cur = mk;
    mk = i + stride < n4 ? seen[i + stride] : 0;
    if (!cur) continue;
    f4* r = rec + i * 4;
    f4 gg = ldnt(r), mm = ldnt(r + 1), vv = ldnt(r + 2), pp = ldnt(p + i);
    upd(pp, gg, mm, vv);
    stnt(p + i, pp); stnt(r, gg); stnt(r + 1, mm); stnt(r + 2, vv);
  }
}

// (c) today's arrays, but a lane owns a whole 64-byte LINE (4 groups, their 4 seen bytes as one word): 4 x 16 B per array and lane
__global__ void __launch_bounds__(256) adam_arrays_line(f4* p, f4* g, f4* m, f4* v, const uint32_t* seen4, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mk = i < n16 ? seen4[i] : 0u;
  for (; i < n16; i += stride) {
    const uint32_t cur = mk;
    mk = i + stride < n16 ? seen4[i + stride] : 0u;
    if (!cur) continue;
    f4 gg[4], mm[4], vv[4], pp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((cur >> (8 * k)) & 0xFFu) { gg[k] = ldnt(g + 4 * i + k); mm[k] = ldnt(m + 4 * i + k); vv[k] = ldnt(v + 4 * i + k); pp[k] = ldnt(p + 4 * i + k); }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((cur >> (8 * k)) & 0xFFu) {
        const bool had = gg[k][0] != 0.0f || gg[k][1] != 0.0f || gg[k][2] != 0.0f || gg[k][3] != 0.0f;
        upd(pp[k], gg[k], mm[k], vv[k]);
        stnt(p + 4 * i + k, pp[k]); stnt(m + 4 * i + k, mm[k]); stnt(v + 4 * i + k, vv[k]);
        if (had) stnt(g + 4 * i + k, gg[k]);
      }
  }
}
// (d) as (a) with TWO groups per lane and trip, a whole wave-width apart (more loads in flight per lane)
__global__ void __launch_bounds__(256) adam_arrays_x2(f4* p, f4* g, f4* m, f4* v, const uint8_t* seen, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    const uint8_t a = seen[i], b = j < n4 ? seen[j] : 0;
    f4 g0, m0, v0, p0, g1, m1, v1, p1;
    if (a) { g0 = ldnt(g + i); m0 = ldnt(m + i); v0 = ldnt(v + i); p0 = ldnt(p + i); }
    if (b) { g1 = ldnt(g + j); m1 = ldnt(m + j); v1 = ldnt(v + j); p1 = ldnt(p + j); }
    if (a) { upd(p0, g0, m0, v0); stnt(p + i, p0); stnt(m + i, m0); stnt(v + i, v0); stnt(g + i, g0); }
    if (b) { upd(p1, g1, m1, v1); stnt(p + j, p1); stnt(m + j, m1); stnt(v + j, v1); stnt(g + j, g1); }
  }
}

int main(int argc, char** argv) {
  const int64_t n4 = 33554432;  // NeuRadar's main table: 8 levels x 2^22 entries
  const size_t bytes = (size_t)n4 * 16;
  f4 *p, *g, *m, *v, *rec;
  uint8_t* seen;
  hipMalloc(&p, bytes); hipMalloc(&g, bytes); hipMalloc(&m, bytes); hipMalloc(&v, bytes); hipMalloc(&rec, bytes * 4); hipMalloc(&seen, n4);
  hipMemset(p, 0, bytes); hipMemset(g, 0, bytes); hipMemset(m, 0, bytes); hipMemset(v, 0, bytes); hipMemset(rec, 0, bytes * 4);
  std::vector<uint8_t> h(n4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pattern = 0; pattern < 2; ++pattern)
    for (double dens : {0.15, 0.39, 0.70}) {
      uint64_t s = 88172645463325252ull;
      auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
      int64_t live = 0;
      if (pattern == 0) {
        for (int64_t i = 0; i < n4; ++i) live += (h[i] = rnd() < dens);
      } else {  // runs: whole 64-byte lines of the four-array layout (4 groups) live or not -- coarse levels, coherent rays
        for (int64_t i = 0; i < n4; i += 4) { const uint8_t b = rnd() < dens; for (int k = 0; k < 4; ++k) h[i + k] = b; live += 4 * b; }
      }
      hipMemcpy(seen, h.data(), n4, hipMemcpyHostToDevice);
      for (int variant = 0; variant < 4; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
          hipEventRecord(e0);
          if (variant == 0) hipLaunchKernelGGL(adam_arrays, dim3(4096), dim3(256), 0, 0, p, g, m, v, seen, n4);
          else if (variant == 1) hipLaunchKernelGGL(adam_records, dim3(4096), dim3(256), 0, 0, p, rec, seen, n4);
          else if (variant == 2) hipLaunchKernelGGL(adam_arrays_line, dim3(4096), dim3(256), 0, 0, p, g, m, v, reinterpret_cast<const uint32_t*>(seen), n4 / 4);
          else hipLaunchKernelGGL(adam_arrays_x2, dim3(4096), dim3(256), 0, 0, p, g, m, v, seen, n4);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        const double moved = (double)live * (variant == 1 ? 128.0 : 112.0) + (double)n4;  // bytes the kernel asks for
        printf("%s live %.2f (%s): %-26s %7.1f us  %6.2f TB/s of requested bytes\n", pattern == 0 ? "random groups" : "whole lines  ", dens,
               pattern == 0 ? "16-B granules" : "64-B granules", variant == 0 ? "four arrays (today)" : variant == 1 ? "p + [g|m|v|pad] records" : variant == 2 ? "four arrays, lane = line" : "four arrays, 2 groups/trip", best * 1e3, moved / (best * 1e-3) / 1e12);
      }
    }
  return 0;
}
