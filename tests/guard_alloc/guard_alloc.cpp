// Test infrastructure (tests/test_gpu_redzone.py): a torch.cuda.memory.CUDAPluggableAllocator that turns EVERY out-of-bounds access
// of EVERY kernel a process launches -- this repo's, MIOpen's, ATen's -- into a GPU page fault, i.e. into the ROCr runtime's
// "Memory access fault by GPU node ..." + abort of the (child) process.
//
// torch's caching allocator hands out pieces of 2-MB / 20-MB segments: an overrun of a few bytes lands in mapped memory unless the
// tensor happens to be the last block of its segment -- which depends on everything the process allocated before (the round-4
// suite aborted in 2 of 8 runs, no test alone ever did).  Here every allocation is its own hipMalloc'ed region, rounded up to a
// multiple of 2 MB, and the tensor is its LAST bytes (start rounded down to 16 bytes: vector accesses stay aligned).  The runtime
// spaces such regions 2 MB apart with nothing mapped in between (tools/probe_redzone.py: an overrun of 8 bytes faults).
// 64 canary bytes in front of every tensor are checked when it is freed (writes before the start): guard_canary_failures().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
constexpr size_t kGranule = 2u << 20;
constexpr size_t kCanary = 64;
std::mutex g_mutex;
std::unordered_map<void*, void*> g_base;  // tensor address -> region base
long g_failures = 0, g_allocs = 0;
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t stream) {
  (void)stream;
  if (size <= 0) size = 1;
  const size_t span = ((size_t)size + 15) & ~(size_t)15;
  const size_t region = (span + kCanary + kGranule - 1) / kGranule * kGranule;
  int prev = 0;
  hipGetDevice(&prev);
  if (prev != device) hipSetDevice(device);
  void* base = nullptr;
  if (hipMalloc(&base, region) != hipSuccess) {
    if (prev != device) hipSetDevice(prev);
    return nullptr;
  }
  char* p = static_cast<char*>(base) + region - span;
  hipMemset(p - kCanary, 0xA5, kCanary);  // (synchronous enough: the null stream orders it in front of later work)
  if (prev != device) hipSetDevice(prev);
  std::lock_guard<std::mutex> lock(g_mutex);
  g_base[p] = base;
  ++g_allocs;
  return p;
}

extern "C" void guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
  (void)size; (void)device; (void)stream;
  if (!ptr) return;
  void* base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_base.find(ptr);
    if (it == g_base.end()) return;
    base = it->second;
    g_base.erase(it);
  }
  unsigned char canary[kCanary];
  if (hipMemcpy(canary, static_cast<char*>(ptr) - kCanary, kCanary, hipMemcpyDeviceToHost) == hipSuccess) {  // (waits for the device)
    for (size_t i = 0; i < kCanary; ++i)
      if (canary[i] != 0xA5) {
        std::lock_guard<std::mutex> lock(g_mutex);
        ++g_failures;
        fprintf(stderr, "guard_alloc: bytes in FRONT of the allocation at %p were overwritten\n", ptr);
        break;
      }
  }
  hipFree(base);
}

extern "C" long guard_canary_failures(void) { return g_failures; }
extern "C" long guard_allocations(void) { return g_allocs; }
