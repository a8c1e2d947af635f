// GPU lab (development tool): float atomic-add throughput when the lanes of ONE wave instruction come in groups of G adjacent
// floats (G = 1, 4, 8, 16 -- a 4-, 16-, 32-, 64-byte chunk at a random G-aligned place of the table): is the memory side's atomic
// rate a rate of lanes, of 16-byte entries, or of (instruction, line) requests?  Decides whether the main grid's scatter
// (grid_shared.hip: 4 adjacent lanes = one 16-byte vertex) gains from leaving x-pair vertices -- adjacent 16-byte entries when x is
// even -- on adjacent lane quads (VERDICT r05 next #4b).  Also: a quad pair in REVERSED order (entry e^1 first) and a quad pair
// split over two instructions.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o tools/atomic_group_lab tools/atomic_group_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

// G lanes share one random chunk; VARIANT 0: ascending inside the chunk; 1: the two halves of the chunk swapped (G >= 8);
// 2: the second half leaves in a SECOND instruction (same lanes, next iteration) -- what a flush does today when the pair's two
// vertices were claimed by different lanes
template <int G, int VARIANT>
__global__ void __launch_bounds__(256) group_kernel(float* __restrict__ table, uint32_t words, int per_thread) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t grp = gid / G, in = gid % G;
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t r = pcg(grp * 9781u + i * 6271u + 17u);
    const uint32_t chunk = (r % (words / G)) * G;
    uint32_t off = in;
    if (VARIANT == 1) off = in ^ (G / 2);
    if (VARIANT == 2) {
      // lanes [0, G/2) of the group: first half now; lanes [G/2, G): the first half of ANOTHER chunk now.  The second halves
      // follow in the next iteration's instruction
      const uint32_t r2 = pcg((grp * 2 + (in >= G / 2 ? 1u : 0u)) * 9781u + (i >> 1) * 6271u + 29u);
      const uint32_t c2 = (r2 % (words / G)) * G;
      const uint32_t o2 = (in % (G / 2)) + ((i & 1) ? G / 2 : 0);
      unsafeAtomicAdd(&table[c2 + o2], 1.0f);
      continue;
    }
    unsafeAtomicAdd(&table[chunk + off], 1.0f);
  }
}

template <int G, int VARIANT> void run(const char* name, float* table, uint32_t words) {
  const int blocks = 8192, per_thread = 16;
  const double total = (double)blocks * 256 * per_thread;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(table, 0, (size_t)words * 4));
    CK(hipEventRecord(a));
    group_kernel<G, VARIANT><<<blocks, 256>>>(table, words, per_thread);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-44s table %4u MB: %8.3f ms  %7.2f G lane-atomics/s  %7.2f G groups/s  %7.1f GB/s\n", name, words >> 18, best, total / best / 1e6,
         total / G / best / 1e6, total * 4 / best / 1e6);
}

int main() {
  float* table;
  const uint32_t max_words = 128u << 20;
  CK(hipMalloc(&table, (size_t)max_words * 4));
  for (uint32_t mb : {64u, 512u}) {
    const uint32_t words = mb << 18;
    run<1, 0>("G=1  (4 B, random)", table, words);
    run<4, 0>("G=4  (16 B = one vertex)", table, words);
    run<8, 0>("G=8  (32 B = an aligned x-pair)", table, words);
    run<8, 1>("G=8  halves swapped (odd entry first)", table, words);
    run<8, 2>("G=8  halves in two instructions", table, words);
    run<16, 0>("G=16 (64 B)", table, words);
    run<32, 0>("G=32 (128 B)", table, words);
    run<64, 0>("G=64 (256 B, a whole wave contiguous)", table, words);
  }
  return 0;
}
