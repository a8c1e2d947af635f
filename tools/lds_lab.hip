// GPU lab (development tool): LDS atomic throughput on gfx950 by address pattern -- what an LDS-resident accumulator costs.
//   ops: 0 ds_write_b32   1 ds_add_f32 (no return)   2 ds_add_rtn_u32   3 ds_cmpst_rtn_b32 + ds_add_f32 (tagged slot)
//        5 ds_add_u64   6 ds_add_u32 (no return)   7 ds_add_rtn_u32 on 64 counters (ranking)
//        4 plain read-add-write (no atomics; only valid for conflict-free patterns, rate reference)
//   patterns: 0 lane-distinct consecutive   1 random over `words`   2 all lanes one address   3 runs of 4 lanes per address
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lds_lab tools/lds_lab.hip ; run: tools/lds_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

constexpr int kWords = 8192;  // 32 KB of floats + 32 KB of tags per block

template <int OP, int PAT>
__global__ void __launch_bounds__(256) lab(float* out, int iters) {
  __shared__ float vals[kWords];
  __shared__ uint32_t tags[kWords];
  for (int i = threadIdx.x; i < kWords; i += 256) { vals[i] = 0.0f; tags[i] = 0xFFFFFFFFu; }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, gid = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.0f;
  uint32_t r = pcg(gid);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      r = r * 1664525u + 1013904223u;
      uint32_t a;
      if (PAT == 0) a = (threadIdx.x + 256 * ((it * 8 + k) & 31)) & (kWords - 1);
      else if (PAT == 1) a = (r >> 8) & (kWords - 1);
      else if (PAT == 2) a = (__builtin_amdgcn_readfirstlane(r) >> 8) & (kWords - 1);
      else a = ((__builtin_amdgcn_readfirstlane(r) >> 8) + (lane >> 2) * 33) & (kWords - 1);
      if (OP == 0) vals[a] = (float)k;
      else if (OP == 1) __hip_atomic_fetch_add(&vals[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (OP == 2) acc += (float)__hip_atomic_fetch_add(&tags[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (OP == 3) {
        uint32_t expected = 0xFFFFFFFFu;
        __hip_atomic_compare_exchange_strong(&tags[a], &expected, a, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (expected == 0xFFFFFFFFu || expected == a) __hip_atomic_fetch_add(&vals[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else acc += 1.0f;
      } else if (OP == 5) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(vals) + (a >> 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (OP == 6) __hip_atomic_fetch_add(&tags[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (OP == 7) acc += (float)__hip_atomic_fetch_add(&tags[a & 63], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else vals[a] += 1.0f;
    }
  }
  __syncthreads();
  float s = acc;
  for (int i = threadIdx.x; i < kWords; i += 256) s += vals[i];
  if (s == 123.456f) out[gid] = s;
}

template <int OP, int PAT> void run(const char* name, float* out) {
  const int blocks = 2048, iters = 256;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((lab<OP, PAT>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((lab<OP, PAT>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)blocks * 256 * iters * 8;
  printf("%-44s %8.3f ms  %8.1f G lane-ops/s  (%.2f lane-ops/clk/CU at 2.4 GHz, 256 CUs)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 2.4 / 256);
}

int main() {
  float* out; CK(hipMalloc(&out, 2048 * 256 * 4));
  run<0, 0>("ds_write      consecutive", out);
  run<0, 1>("ds_write      random", out);
  run<4, 0>("read-add-write consecutive", out);
  run<1, 0>("ds_add_f32    consecutive", out);
  run<1, 1>("ds_add_f32    random", out);
  run<1, 3>("ds_add_f32    runs of 4 lanes", out);
  run<1, 2>("ds_add_f32    one address", out);
  run<2, 0>("ds_add_rtn_u32 consecutive", out);
  run<2, 1>("ds_add_rtn_u32 random", out);
  run<2, 2>("ds_add_rtn_u32 one address", out);
  run<5, 0>("ds_add_u64 (no rtn) consecutive", out);
  run<5, 1>("ds_add_u64 (no rtn) random", out);
  run<5, 3>("ds_add_u64 (no rtn) runs of 4", out);
  run<5, 2>("ds_add_u64 (no rtn) one address", out);
  run<6, 1>("ds_add_u32 (no rtn) random", out);
  run<7, 1>("ds_add_rtn_u32 random over 64 counters", out);
  run<3, 0>("cmpst+add_f32 consecutive", out);
  run<3, 1>("cmpst+add_f32 random", out);
  run<3, 3>("cmpst+add_f32 runs of 4 lanes", out);
  run<3, 2>("cmpst+add_f32 one address", out);
  return 0;
}
