// GPU lab (development tool): float atomic-add throughput by memory scope and by destination locality.
//   agent scope     : what atomicAdd emits; executed memory-side, coherent across the 8 XCDs
//   workgroup scope : executed in the issuing XCD's L2 -- only correct when every accessor of an address runs on
//                     the same XCD, which the "slice" modes guarantee by construction (slice = f(XCC_ID))
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o tools/atomic_lab tools/atomic_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

// MODE 0: agent scope, random over the whole table.  1: agent scope, random inside the XCD's slice.
// 2: workgroup scope, random inside the XCD's slice.   3: workgroup scope, whole table (WRONG results expected; rate only)
template <int MODE>
__global__ void __launch_bounds__(256) atom_kernel(float* __restrict__ table, uint32_t words, int per_thread, uint32_t* __restrict__ xcc_hist) {
  const uint32_t x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xcc_hist[x], 1u);
  const uint32_t slice = words / 8;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t r = pcg(gid * 9781u + i * 6271u + 17u);
    uint32_t idx;
    if (MODE == 0 || MODE == 3) idx = r % words; else idx = x * slice + r % slice;
    if (MODE == 0 || MODE == 1) __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(&table[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__global__ void sum_kernel(const float* __restrict__ t, uint32_t words, double* out) {
  double s = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) s += t[i];
  for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

template <int MODE> void run(const char* name, float* table, uint32_t words, uint32_t* hist, double* dsum, int blocks = 8192) {
  const int per_thread = 16 * 8192 / blocks;
  const double total = (double)blocks * 256 * per_thread;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  double got = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(table, 0, (size_t)words * 4));
    CK(hipMemset(hist, 0, 64));
    CK(hipMemset(dsum, 0, 8));
    CK(hipEventRecord(a));
    atom_kernel<MODE><<<blocks, 256>>>(table, words, per_thread, hist);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
    sum_kernel<<<1024, 256>>>(table, words, dsum);
    CK(hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost));
  }
  uint32_t h[16];
  CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
  printf("%-34s table %7.2f MB: %8.3f ms  %7.2f G atomics/s   sum %s (%.0f of %.0f)  blocks/xcc:", name, words * 4 / 1048576.0, best,
         total / best / 1e6, got == total ? "OK " : "BAD", got, total);
  for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
  printf("\n");
}

int main() {
  float* table; uint32_t* hist; double* dsum;
  const uint32_t max_words = 64u << 20;
  CK(hipMalloc(&table, (size_t)max_words * 4)); CK(hipMalloc(&hist, 64)); CK(hipMalloc(&dsum, 8));
  for (uint32_t mb : {4u, 24u, 64u, 256u}) {
    const uint32_t words = mb << 18;
    run<0>("agent scope, whole table", table, words, hist, dsum);
    run<1>("agent scope, XCD slice", table, words, hist, dsum);
    run<2>("workgroup scope, XCD slice", table, words, hist, dsum);
    run<3>("workgroup scope, whole table", table, words, hist, dsum);
  }
  // how many CUs does it take to saturate the atomic path?  (one 256-thread block per CU up to 256 blocks)
  for (int blocks : {32, 64, 128, 256, 512, 1024, 2048}) {
    char name[64];
    snprintf(name, sizeof name, "agent scope, %d blocks", blocks);
    run<0>(name, table, 24u << 18, hist, dsum, blocks);
  }
  return 0;
}
