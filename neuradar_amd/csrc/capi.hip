// ABI bookkeeping entry points of libneuradar_hip.so (see include/neuradar_hip.h): version, one-time device setup, tuning knobs.
#include "nr_common.h"

namespace {
NrTuning g_tuning = {};
}  // namespace

const NrTuning& nr_tuning() { return g_tuning; }

namespace {
constexpr int kMaxDevices = 64;
uint32_t* g_generation[kMaxDevices] = {};  // one word per device, allocated by nr_init on that device (process lifetime)
}  // namespace

uint32_t* nr_generation_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return g_generation[dev];
}

extern "C" const uint32_t* nr_param_generation(void) { return nr_generation_ptr(); }

extern "C" int nr_abi_version(void) { return NR_ABI_VERSION; }
extern "C" const char* nr_target_arch(void) { return "gfx950"; }

// Once per device (the CURRENT one), before the first launch: the dynamic-LDS attributes of the kernels that stage more than 64 KB.
extern "C" int nr_init(void) {
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < kMaxDevices && g_generation[dev] == nullptr) {
    uint32_t* w = nullptr;
    if (hipError_t e = hipMalloc(&w, 64); e != hipSuccess) return (int)e;
    if (hipError_t e = hipMemset(w, 0, 64); e != hipSuccess) return (int)e;
    g_generation[dev] = w;
  }
  if (int rc = nr_init_conv7()) return rc;
  if (int rc = nr_init_encoder()) return rc;
  if (int rc = nr_init_radar()) return rc;
  return 0;
}

// Launch-shape knobs for A/B measurements (names: the NR_TUNE_* enum of the header).  Call it at load time, not while another thread
// launches: the table is plain process memory.  value 0 restores the default.
extern "C" int nr_set_tuning(int knob, int value) {
  if (value < 0) return NR_EINVAL;
  switch (knob) {
    case NR_TUNE_CONV7_BLOCKS: g_tuning.conv7_blocks = value; break;
    case NR_TUNE_BIN_BLOCKS_PER_CU: g_tuning.bin_blocks_per_cu = value; break;
    case NR_TUNE_SHARED_BLOCKS: g_tuning.shared_blocks = value; break;
    case NR_TUNE_FIELD_FWD_BLOCKS: g_tuning.field_fwd_blocks = value; break;
    case NR_TUNE_FIELD_BWD_BLOCKS: g_tuning.field_bwd_blocks = value; break;
    case NR_TUNE_PDBWD_BLOCKS: g_tuning.pdbwd_blocks = value; break;
    case NR_TUNE_ADAM_BLOCKS: g_tuning.adam_blocks = value; break;
    case NR_TUNE_PW_MFMA_OFF: g_tuning.pw_mfma_off = value; break;
    case NR_TUNE_SHARED_SPLIT: g_tuning.shared_split = value; break;
    default: return NR_EINVAL;
  }
  return 0;
}
