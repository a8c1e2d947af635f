// Shared device helpers for the NeuRadar gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neuradar_hip.h"

#define NR_WAVE 64

#define NR_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

static inline hipStream_t nr_s(nr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- launch tuning + one-time kernel attributes (capi.hip) --------------------------------------------
// No entry point reads the environment or keeps lazily-initialised state: the knobs below are set explicitly through
// nr_set_tuning() (the Python binding forwards its NR_* variables ONCE, when it loads the library), and the kernels that need
// more than the default 64 KB of dynamic LDS get their attribute in nr_init() (once per device, before the first launch;
// a launch without it fails with hipErrorInvalidValue -- loudly).  0 = the built-in default.
struct NrTuning {
  int conv7_blocks;        // NR_TUNE_CONV7_BLOCKS: persistent blocks of nr_conv7_fwd
  int bin_blocks_per_cu;   // NR_TUNE_BIN_BLOCKS_PER_CU: bin blocks per HALF CU of the binned scatters
  int shared_blocks;       // NR_TUNE_SHARED_BLOCKS: blocks of nr_hash_encode_bwd_shared
  int field_fwd_blocks;    // NR_TUNE_FIELD_FWD_BLOCKS
  int field_bwd_blocks;    // NR_TUNE_FIELD_BWD_BLOCKS
  int pdbwd_blocks;        // NR_TUNE_PDBWD_BLOCKS: blocks of nr_prop_density_bwd
  int adam_blocks;         // NR_TUNE_ADAM_BLOCKS
  int pw_mfma_off;         // NR_TUNE_PW_MFMA_OFF: 1 = the transposed convolution on the generic pointwise kernels
  int shared_split;        // NR_TUNE_SHARED_SPLIT: threads per row of nr_hash_encode_bwd_shared (1, 2 or 4; 0 = 1)
};
const NrTuning& nr_tuning();
// The device's PARAMETER GENERATION word (capi.hip; allocated by nr_init, NULL before): bumped on the device by every launch that
// writes parameters through raw pointers (the optimizers' schedule kernel nr_adam_hyper -- once per optimizer step --,
// nr_apply_delta16, nr_bn_act_fwd's running statistics), graph replays included.  Consumers that cache something derived from
// parameters (nr_conv7_fold_pack) compare it ON THE DEVICE: no host read, no version counter that raw-pointer writes bypass.
uint32_t* nr_generation_ptr();
__device__ __forceinline__ void nr_bump_generation(uint32_t* gen) {
  if (gen != nullptr) atomicAdd(gen, 1u);
}
// per-file attribute setup, called by nr_init()
int nr_init_conv7();
int nr_init_encoder();
int nr_init_radar();
template <typename Kern>
inline int nr_raise_lds(Kern kern, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static inline int64_t nr_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__device__ __forceinline__ int64_t nr_cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- spatial hash (field_components/encodings.py:406-421) -------------------------------------
// The reference multiplies int32 corners by {1, 2654435761, 805459861} in int64, xors and takes
// mod T.  T is a power of two, so the slot is the low log2(T) bits, which uint32 wrap-around
// arithmetic reproduces exactly (also for negative corners, two's complement).
__device__ __forceinline__ uint32_t nr_hash3(int ix, int iy, int iz, uint32_t mask) {
  return (((uint32_t)ix * 1u) ^ ((uint32_t)iy * 2654435761u) ^ ((uint32_t)iz * 805459861u)) & mask;
}

// ---- the step's stamp value (nr_hash_mark_vertices / nr_adam_step_split): 1 ... 255 from a device-resident step counter, so
// that a replayed hipGraph stamps every step differently; 0 = never stamped
__device__ __forceinline__ int nr_stamp_value(const float* __restrict__ epoch) { return ((int)epoch[0]) % 255 + 1; }

// ---- loss partial sums ---------------------------------------------------------------------------
// Every wave adds its loss into slot (global wave index mod NR_LOSS_SLOTS).  Same-address float atomics
// are applied one after the other at the memory side (~0.4 us each, measured): with 64 slots the
// 4 096 waves of a 4 096-ray launch queued 64 deep and that queue WAS the kernel (inter-level loss
// 30 us, flat in its proposal size; 1 024 slots: 4 deep).
__device__ __forceinline__ int nr_loss_slot_index() {
  return (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (NR_LOSS_SLOTS - 1));
}

// ---- per-sample row order ------------------------------------------------------------------------
// The per-sample arrays that only the hash grid and the MLPs touch (positions, per-level features and
// their gradients) may keep the rows of the first `sm` rays SAMPLE-major, row = s * sm + b: a wave's 64
// rows are then 64 neighbouring rays at one sample slot -- coalesced loads AND coherent grid cells, which
// is what camera patches want.  Rays b >= sm (and everything when sm = 0) keep the reference's ray-major
// row b * S + s, which is what incoherent lidar / radar rays want (neighbouring samples of ONE ray share
// coarse cells).  Per-ray scans (weights, resampling, compositing) always use b * S + s.
__device__ __forceinline__ int64_t nr_row_of(int64_t b, int s, int S, int64_t sm) {
  return b < sm ? (int64_t)s * sm + b : b * S + s;
}
struct NrRowMap {
  int64_t ray, out;  // ray index, ray-major sample index
};
__device__ __forceinline__ NrRowMap nr_row_map(int64_t j, int64_t n, int S, int64_t sm) {
  NrRowMap m;
  if (S <= 0) {
    m.ray = j; m.out = j;
  } else if (j >= sm * S) {
    m.ray = j / S; m.out = j;
  } else {
    const int64_t s = j / sm;
    m.ray = j - s * sm;
    m.out = m.ray * S + s;
  }
  return m;
}

// ---- frustum sample -> contracted isotropic Gaussian --------------------------------------------
// Frustums.get_fast_isotropic_gaussian, one multisample (cameras/rays.py:109-124) followed by
// ScaledSceneContraction(order=inf) on the GaussiansStd (spatial_distortions.py:103-113,126-136).
__device__ __forceinline__ void nr_contract_sample(const float* __restrict__ origin, const float* __restrict__ direction,
                                                   float pixel_area, float e0, float e1, float scale, float (&x01)[3],
                                                   float& std01) {
  const float half = (e1 - e0) / 2.0f;
  const float t = e0 + 1.0f * half;
  const float cross = pixel_area * (t * t);
  float sd = powf(cross * half, 1.0f / 3.0f);
  float m[3];
  float mag = 0.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    m[a] = (origin[a] + direction[a] * t) / scale;  // spatial_distortions.py:133
    mag = fmaxf(mag, fabsf(m[a]));
  }
  sd = sd / scale;
  if (!(mag < 1.0f)) {  // spatial_distortions.py:107-112 (L-inf norm)
    const float cm = fmaxf(mag, 1.0f);
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = (2.0f - (1.0f / cm)) * (m[a] / cm);
    const float k = powf(2.0f * cm - 1.0f, 1.0f / 3.0f) / cm;
    sd = sd * (k * k);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) x01[a] = (m[a] + 2.0f) / 4.0f;  // :135
  std01 = sd / 4.0f;                                           // :136
}

// ---- ZipNeRF power transform (utils/math.py:541-579), finite-lambda branch ---------------------
// torch.pow special-cases exponent -1 as a reciprocal; NeuRadar uses lambda = -1.
__device__ __forceinline__ float nr_pow_lam(float base, float lam) {
  return lam == -1.0f ? 1.0f / base : powf(base, lam);
}
__device__ __forceinline__ float nr_power_fn(float x, float lam) {
  const float lam_1 = fabsf(lam - 1.0f);
  return (lam_1 / lam) * (nr_pow_lam(x / lam_1 + 1.0f, lam) - 1.0f);
}
__device__ __forceinline__ float nr_inv_power_fn(float x, float lam) {
  const float lam_1 = fabsf(lam - 1.0f);
  const float b = fmaxf(x * lam / lam_1 + 1.0f, 1e-10f);
  return (nr_pow_lam(b, 1.0f / lam) - 1.0f) * lam_1;
}
// spacing_to_euclidean_fn of SpacedSampler (model_components/ray_samplers.py:119-120) with
// PowerSampler's spacing fns (:846-851)
__device__ __forceinline__ float nr_spacing_to_euclid(float s, float s_near, float s_far, float lam, float scaling) {
  return nr_inv_power_fn(s * s_far + (1.0f - s) * s_near, lam) / scaling;
}

// torch.linspace(start, end, steps)[idx] in fp32 (symmetric two-sided evaluation, like ATen)
__device__ __forceinline__ float nr_linspace(float start, float end, int steps, int idx) {
  if (steps == 1) return start;
  const float step = (end - start) / (float)(steps - 1);
  const int halfway = steps / 2;
  return idx < halfway ? start + step * (float)idx : end - step * (float)(steps - idx - 1);
}

// torch.nan_to_num defaults: nan -> 0, +-inf -> +-FLT_MAX
__device__ __forceinline__ float nr_nan_to_num(float v) {
  if (isnan(v)) return 0.0f;
  if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

// ---- DPP / permlane cross-lane moves (VALU only; ds_bpermute-based __shfl_* goes through the LDS
// crossbar and measured ~16 cycles per wave-instruction per CU, which made the scatter shuffle-bound) --
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int nr_dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROWMASK, 0xF, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float nr_dpp_f(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROWMASK, 0xF, false));
}
constexpr int NR_DPP_ROW_SHR = 0x110;   // + n: lane i <- lane i-n inside its row of 16
constexpr int NR_DPP_ROW_SHL = 0x100;   // + n: lane i <- lane i+n inside its row of 16
constexpr int NR_DPP_WAVE_SHL1 = 0x130; // lane i <- lane i+1 (whole wave)
constexpr int NR_DPP_WAVE_SHR1 = 0x138; // lane i <- lane i-1 (whole wave)
constexpr int NR_DPP_ROW_BCAST15 = 0x142;  // lane 15 of row r -> every lane of row r+1
constexpr int NR_DPP_ROW_BCAST31 = 0x143;  // lane 31 -> every lane of rows 2 and 3
// value of lane (l ^ 32): v_permlane32_swap exchanges the upper half of one register with the lower half of another
__device__ __forceinline__ int nr_xor32_i(int x) {
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
}
__device__ __forceinline__ float nr_xor32_f(float x) { return __int_as_float(nr_xor32_i(__float_as_int(x))); }

// One step of a SEGMENTED inclusive scan over N values per lane (runs of neighbouring lanes; `flag` != 0 once a lane's partial
// sums have reached its run's head): lane i adds lane (i - d)'s values unless its flag is set, then the flags propagate.  Used
// by the three scatter kernels to fold neighbouring lanes in one grid cell before anything touches LDS or memory.
// R6: an element costs TWO vector instructions per in-row step and ONE per row-broadcast step (round 5: five issue slots --
// v_mov 0 / s_nop / v_mov_dpp / v_cndmask-or-fma / v_add: the scan was two thirds of the main scatter's instructions):
//   * in-row steps (C = row_shr:d): the mask moves to the SOURCE side -- lane j offers its values only if lane j + d of its row
//     still accumulates (flags shifted the other way: one DPP per step) -- so the add itself carries the DPP operand
//     (v_cndmask + v_add_{u32,f32}_dpp); all offers are formed before the adds (a DPP operand written by the previous
//     instruction costs two wait states);
//   * row-broadcast steps (C = row_bcast:15, one source, sixteen destinations with flags of their own): the destinations are
//     selected by EXEC -- the step's ONE source lane (15 / 31 / 47 for destination row 1 / 2 / 3) stays active (DPP reads active
//     lanes only; it lies outside the row mask, so it is not written) -- and the add is a bare v_add_*_dpp.
// Same additions in the same order as the masked form: identical sums (floats: x + 1 * y and x + 0 are exact).
template <int C, int R> __device__ __forceinline__ int nr_dpp_zero(int src) { return nr_dpp_i<C, R>(0, src); }
template <int C, int R> __device__ __forceinline__ float nr_dpp_zero(float src) { return nr_dpp_f<C, R>(0.0f, src); }
template <int C, int R, int N, typename T>
__device__ __forceinline__ void nr_seg_scan_step(T* __restrict__ x, int& flag, int lane) {
  if constexpr (C >= NR_DPP_ROW_SHR && C < NR_DPP_ROW_SHR + 16) {
    constexpr int D = C - NR_DPP_ROW_SHR;
    const bool offer = nr_dpp_i<NR_DPP_ROW_SHL + D, 0xF>(1, flag) == 0;  // lane j + d of my row still accumulates
    T m[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      m[k] = offer ? x[k] : (T)0;
      asm volatile("" : "+v"(m[k]));
    }
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] += nr_dpp_zero<C, R>(m[k]);
  } else {
    static_assert(C == NR_DPP_ROW_BCAST15 && (R == 0x2 || R == 0x4 || R == 0x8), "row-broadcast steps of the scan: one destination row each");
    constexpr int kSource = R == 0x2 ? 15 : R == 0x4 ? 31 : 47;  // (a lane of the row BEFORE the destination row: never written)
    if (flag == 0 || lane == kSource) {
#pragma unroll
      for (int k = 0; k < N; ++k) x[k] += nr_dpp_zero<C, R>(x[k]);
    }
  }
  flag |= nr_dpp_i<C, R>(0, flag);
}

// maximum over the wave, valid in LANE 63 only: six v_max_f32 with DPP operands (prefix maxima inside the rows of 16, then
// the two row broadcasts) instead of six ds_bpermute round trips
__device__ __forceinline__ float nr_wave_max_to_lane63(float v) {
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_SHR + 1, 0xF>(v, v));
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_SHR + 2, 0xF>(v, v));
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_SHR + 4, 0xF>(v, v));
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_SHR + 8, 0xF>(v, v));
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_BCAST15, 0xA>(v, v));
  v = fmaxf(v, nr_dpp_f<NR_DPP_ROW_BCAST31, 0xC>(v, v));
  return v;
}

// ---- wave64 scans / reductions ----------------------------------------------------------------
__device__ __forceinline__ int nr_lane() { return threadIdx.x & (NR_WAVE - 1); }

// sum over the wave, in every lane: prefix sums inside the rows of 16 and the two row broadcasts on DPP operands (lane 63
// ends up with the total), then one readlane -- seven vector instructions instead of six ds_bpermute round trips through
// the LDS crossbar.  (The order of the additions is fixed: rows left to right, then rows 0+1, 2+3, then the halves.)
__device__ __forceinline__ float nr_wave_sum(float v) {
  v += nr_dpp_f<NR_DPP_ROW_SHR + 1, 0xF>(0.0f, v);
  v += nr_dpp_f<NR_DPP_ROW_SHR + 2, 0xF>(0.0f, v);
  v += nr_dpp_f<NR_DPP_ROW_SHR + 4, 0xF>(0.0f, v);
  v += nr_dpp_f<NR_DPP_ROW_SHR + 8, 0xF>(0.0f, v);
  v += nr_dpp_f<NR_DPP_ROW_BCAST15, 0xA>(0.0f, v);
  v += nr_dpp_f<NR_DPP_ROW_BCAST31, 0xC>(0.0f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), NR_WAVE - 1));
}

// inclusive prefix sum across the wave
__device__ __forceinline__ float nr_wave_incl_sum(float v) {
  const int lane = nr_lane();
#pragma unroll
  for (int o = 1; o < NR_WAVE; o <<= 1) {
    float t = __shfl_up(v, o, NR_WAVE);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive suffix sum across the wave
__device__ __forceinline__ float nr_wave_incl_suffix_sum(float v) {
  const int lane = nr_lane();
#pragma unroll
  for (int o = 1; o < NR_WAVE; o <<= 1) {
    float t = __shfl_down(v, o, NR_WAVE);
    if (lane + o < NR_WAVE) v += t;
  }
  return v;
}
// exclusive prefix / suffix sums.  Shifted inclusive scans, NOT "inclusive - own": subtracting a
// lane's own (possibly huge) value back out would cancel away the small prefix it sits on.
__device__ __forceinline__ float nr_wave_excl_sum(float v) {
  const float incl = nr_wave_incl_sum(v);
  const float up = __shfl_up(incl, 1, NR_WAVE);
  return nr_lane() == 0 ? 0.0f : up;
}
__device__ __forceinline__ float nr_wave_excl_suffix_sum(float v) {
  const float incl = nr_wave_incl_suffix_sum(v);
  const float dn = __shfl_down(incl, 1, NR_WAVE);
  return nr_lane() == NR_WAVE - 1 ? 0.0f : dn;
}
// inclusive prefix product
__device__ __forceinline__ float nr_wave_incl_prod(float v) {
  const int lane = nr_lane();
#pragma unroll
  for (int o = 1; o < NR_WAVE; o <<= 1) {
    float t = __shfl_up(v, o, NR_WAVE);
    if (lane >= o) v *= t;
  }
  return v;
}
