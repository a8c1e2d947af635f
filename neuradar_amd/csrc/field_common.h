// Shared pieces of the NeuRADField MLP kernels (fp32 in mlp.hip, bf16 / fp16 in mlp_lp.hip): dimensions, the LDS
// weight / gradient image layouts, per-tile workspace layout and the input helpers.
#pragma once
#include <stdlib.h>

#include "nr_common.h"
#include "mlp_tiles.h"
#include "sh4.h"

namespace nrfield {
using namespace nrmlp;

constexpr int kC = 32;         // geo_feat_dim == nff_out_dim (fields/neurad_field.py:64,97)
constexpr int kSH = 16;        // SHEncoding(levels=4)
constexpr float kBetaMin = 1e-4f;  // model_components/utils.py:24

template <int IN, int HID>
struct FieldImage {
  using G1 = Layer<IN, HID>;       // mlp_geo.layers[0]
  using G2 = Layer<HID, kC>;       // mlp_geo.layers[1], embedding rows 1..C
  using F1 = Layer<kC + kSH, HID>; // mlp_feature.layers[0]
  using F2 = Layer<HID, HID>;
  using F3 = Layer<HID, kC>;
  static constexpr int SDF = (HID + 1 + 3) / 4 * 4;  // row 0 of mlp_geo.layers[1] + its bias (padded to 16 bytes)
  // weight image offsets
  static constexpr int oG1 = 0, oG2 = oG1 + G1::SIZE, oSdf = oG2 + G2::SIZE, oF1 = oSdf + SDF,
                       oF2 = oF1 + F1::SIZE, oF3 = oF2 + F2::SIZE, W_TOTAL = oF3 + F3::SIZE;
  // gradient image offsets
  static constexpr int gG1 = 0, gG2 = gG1 + G1::G_SIZE, gSdf = gG2 + G2::G_SIZE, gF1 = gSdf + SDF,
                       gF2 = gF1 + F1::G_SIZE, gF3 = gF2 + F2::G_SIZE, gBeta = gF3 + F3::G_SIZE,
                       G_TOTAL = gBeta + 2;
  static constexpr int HT = (HID + 31) / 32, IT = (IN + 31) / 32;
};

// load a [rows x 32 samples] block given per-sample element addressing elem(k) -> offset
template <int ROWS, typename OffFn>
__device__ __forceinline__ void load_rows(f32x16 (&t)[(ROWS + 31) / 32], const float* __restrict__ base, bool valid,
                                          int h, OffFn off) {
#pragma unroll
  for (int kt = 0; kt < (ROWS + 31) / 32; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt * 32 + rowmap(r, 0) + 4 * h;
      t[kt][r] = (valid && k < ROWS) ? base[off(k)] : 0.0f;
    }
}

__device__ __forceinline__ f32x16 sh_tile(const float* __restrict__ dirs, int64_t ray, int h) {
  // SH of the [0,1]-mapped direction (fields/base_field.py:135-141 + encodings.py:797-800)
  float sh[16];
  nr_sh4((dirs[ray * 3 + 0] + 1.0f) / 2.0f, (dirs[ray * 3 + 1] + 1.0f) / 2.0f, (dirs[ray * 3 + 2] + 1.0f) / 2.0f, sh);
  f32x16 t;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = r < 8 ? (h ? sh[rowmap(r, 0) + 4] : sh[rowmap(r, 0)]) : 0.0f;
  return t;
}

template <int HID>
__device__ __forceinline__ float sdf_row(const f32x16 (&h1)[(HID + 31) / 32], const float* wsdf, int h) {
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < (HID + 31) / 32; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = t * 32 + rowmap(s, 0) + 4 * h;
      if (k < HID) part += wsdf[k] * h1[t][s];
    }
  return part + __shfl_xor(part, 32, NR_WAVE) + wsdf[HID];
}

constexpr int kBwdScrTiles = 4;  // KT + MT <= 4 staged tiles per layer
// d_e / d_sdf handed from the feature half to the geometry half: per 32-sample tile 17 registers x 64 lanes in the
// registers' own layout (float reg * 64 + lane), one coalesced 256-byte access per register on either side
constexpr int64_t kWsTile = 17 * 64;
__host__ __device__ constexpr int64_t ws_floats(int64_t n) { return (n + 31) / 32 * kWsTile; }

inline int check_field(const nr_field_t* f, int* hid) {
  if (!f || !f->beta) return NR_EINVAL;
  const nr_mlp_t &g = f->geo, &m = f->feat;
  if (g.num_layers != 2 || m.num_layers != 3) return NR_EINVAL;
  if (g.in_dim != 32 || g.out_dim != kC + 1 || m.in_dim != kC + kSH || m.out_dim != kC || g.width != m.width) return NR_EINVAL;
  if (g.width != 32 && g.width != 64) return NR_EINVAL;
  if (f->dtype != NR_DTYPE_F32 && f->dtype != NR_DTYPE_BF16 && f->dtype != NR_DTYPE_F16) return NR_EINVAL;
  if (f->dtype != NR_DTYPE_F32 && f->packed == nullptr) return NR_EINVAL;
  for (int l = 0; l < 2; ++l) if (!g.weight[l] || !g.bias[l]) return NR_EINVAL;
  for (int l = 0; l < 3; ++l) if (!m.weight[l] || !m.bias[l]) return NR_EINVAL;
  *hid = g.width;
  return 0;
}

// Blocks of the backward launches = gradient slabs in the workspace.  One block per CU (one wave per SIMD) for the fp32
// kernels, whose dW accumulators leave room for no second wave; the 16-bit kernels at width 32 fit two waves per SIMD
// (<= 256 registers), so they get two blocks per CU.
constexpr int kMaxBwdBlocks = 512;
inline unsigned field_bwd_blocks(int64_t n, const nr_field_t* field, int hid) {
  const int64_t tiles = nr_cdiv(n, 32);
  int cap = field->dtype != NR_DTYPE_F32 && hid == 32 ? kMaxBwdBlocks : 256;
  if (const int v = nr_tuning().field_bwd_blocks; v > 0 && v <= kMaxBwdBlocks) cap = v;
  return (unsigned)(nr_cdiv(tiles, 4) < cap ? nr_cdiv(tiles, 4) : cap);
}

// reduced-precision variants (mlp_lp.hip), dispatched from the entry points in mlp.hip on nr_field_t.dtype
int64_t field_image_bytes_lp(int hid);
int field_pack_lp(const nr_field_t* field, int hid, void* image, hipStream_t st);
int field_fwd_lp(const nr_field_t* field, int hid, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S,
                 int rows_sm, int64_t n, float* feature, float* sdf, float* alpha, unsigned blocks, hipStream_t st);
int field_fwd_gather_lp(const nr_field_t* field, int hid, const float* x01, const float* std01, const float* table,
                        const float* scalings, int log2T, float* feats_out, int64_t sl, const float* dirs, int S, int rows_sm,
                        int64_t n, float* feature, float* sdf, float* alpha, unsigned blocks, hipStream_t st);
int field_bwd_lp(const nr_field_t* field, int hid, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S,
                 int rows_sm, int64_t n, const float* g_feature, const float* g_alpha, const float* g_sdf, float* g_feats,
                 float* ws, float* slab, unsigned blocks, hipStream_t st);

}  // namespace nrfield
