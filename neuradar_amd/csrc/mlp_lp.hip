// NeuRADField MLP stack in reduced precision: bf16 or fp16 operands on v_mfma_f32_32x32x16_{bf16,f16}, fp32
// accumulation -- what the reference trains with (torch.autocast + tcnn FullyFusedMLP, engine/trainer.py:189-200,564,
// field_components/mlp.py:109-127; BASELINE configs[2] "bf16", configs[4] "fp16 MFMA").  Same math, tiling and
// gradient-slab layout as the fp32 kernels in mlp.hip; 16x the matrix rate.
//
// Layout.  A wave owns 32 samples.  A layer's fp32 accumulator tile (MFMA C/D layout: lane = sample, 16 registers =
// 16 of 32 rows) is converted pairwise to 16-bit: registers 8s..8s+7 become the 8-element B fragment of k-step s of
// the next layer, element j of lane half h being row 16s + 8(j>>2) + 4h + (j&3).  The weights are packed once per
// optimizer step (nr_field_pack) into fragment order for exactly that k permutation, forward ([mt][kstep][lane][8])
// and transposed (for dX = W^T dZ), so every MFMA's A operand is ONE conflict-free ds_read_b128.  Weight gradients
// contract over samples (lanes): the two operands are staged per wave as [sample][row] 16-bit images (8-byte
// granules, XOR-swizzled) and read back transposed with ds_read_b64_tr_b16.  The sdf row, biases, the sigmoid and all
// reductions stay fp32.  The backward recomputes the forward (22 MFMAs per tile) instead of reading a stash.
#include "field_common.h"
#include "grid_dev.h"

using namespace nrmlp;
using namespace nrfield;

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Bf16 {
  using elem = __bf16;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float sum2(uint32_t v, float acc) {  // acc + lo + hi, fp32
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, v), __builtin_bit_cast(bf16x2, 0x3f803f80u), acc, false);
  }
};
struct Fp16 {
  using elem = _Float16;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2 v = {(_Float16)lo, (_Float16)hi};  // v_cvt_pk_f16_f32
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float sum2(uint32_t v, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v), __builtin_bit_cast(f16x2, 0x3c003c00u), acc, false);
  }
};

// a [32 rows x 32 samples] activation block as two 16-row B fragments
struct PTile {
  u32x4 s[2];
};

template <typename T>
__device__ __forceinline__ PTile to_ptile(const f32x16& a) {
  PTile p;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) p.s[s][q] = T::pack2(a[8 * s + 2 * q], a[8 * s + 2 * q + 1]);
  return p;
}

__device__ __forceinline__ void relu_tile(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = __int_as_float(max(__float_as_int(a[r]), 0));
}

// g <- g where the (post-ReLU, 16-bit) activation y is non-zero
__device__ __forceinline__ void relu_mask_packed(f32x16& g, const PTile& y) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t v = y.s[s][q];
      if ((v & 0xffffu) == 0u) g[8 * s + 2 * q] = 0.0f;
      if ((v & 0xffff0000u) == 0u) g[8 * s + 2 * q + 1] = 0.0f;
    }
}

constexpr int kFragBytes = 1024;  // one MFMA A operand for all 64 lanes

// ---- weight image (bytes).  Forward blocks [mt][kstep], transposed blocks [kt][mstep], then the fp32 block ----
template <int HID>
struct LpImage {
  static constexpr int HT = HID / 32, HS = HID / 16;
  static constexpr int oG1f = 0, szG1f = HT * 2 * kFragBytes;          // mlp_geo.layers[0]: 32 -> HID
  static constexpr int oG2f = oG1f + szG1f, szG2f = HS * kFragBytes;   // mlp_geo.layers[1] rows 1..32: HID -> 32
  static constexpr int oF1f = oG2f + szG2f, szF1f = HT * 3 * kFragBytes;  // mlp_feature.layers[0]: 48 -> HID
  static constexpr int oF2f = oF1f + szF1f, szF2f = HT * HS * kFragBytes;
  static constexpr int oF3f = oF2f + szF2f, szF3f = HS * kFragBytes;   // HID -> 32
  static constexpr int oF1t = oF3f + szF3f, szF1t = HS * kFragBytes;   // d_e rows only (SH carries no gradient)
  static constexpr int oF2t = oF1t + szF1t, szF2t = HT * HS * kFragBytes;
  static constexpr int oF3t = oF2t + szF2t, szF3t = HT * 2 * kFragBytes;
  static constexpr int oG1t = oF3t + szF3t, szG1t = HS * kFragBytes;
  static constexpr int oG2t = oG1t + szG1t, szG2t = HT * 2 * kFragBytes;
  static constexpr int oF32 = oG2t + szG2t;
  // fp32 block (float offsets)
  static constexpr int bG1 = 0, bG2 = bG1 + HT * 32, bF1 = bG2 + 32, bF2 = bF1 + HT * 32, bF3 = bF2 + HT * 32,
                       wSdf = bF3 + 32, nF32 = wSdf + HID + 4;  // sdf row (HID weights + bias), padded
  static constexpr int BYTES = oF32 + nF32 * 4;
  static_assert(BYTES % 16 == 0, "image must be a multiple of 16 bytes");
};

template <int BYTES>
__device__ __forceinline__ void copy_bytes(unsigned char* dst, const unsigned char* __restrict__ src) {
  static_assert(BYTES % 16 == 0, "16-byte pieces");
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (int i = threadIdx.x; i < BYTES / 16; i += blockDim.x) d[i] = s[i];
}

// y[mt] = bias + W x, x given as KS k-steps (x[ks >> 1].s[ks & 1])
template <typename T, int KS, int MT>
__device__ __forceinline__ void dense_lp(const PTile* x, f32x16 (&y)[MT], const unsigned char* wimg, const float* bias,
                                         int lane, int h) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x16 acc;
    if (bias != nullptr) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3 are rows 8g+4h .. +3
        const float4 b = *reinterpret_cast<const float4*>(bias + mt * 32 + 8 * g + 4 * h);
        acc[4 * g] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(wimg + (mt * KS + ks) * kFragBytes + lane * 16);
      acc = T::mfma(a, x[ks >> 1].s[ks & 1], acc);
    }
    y[mt] = acc;
  }
}

// ---- weight gradients: per-wave transposition through LDS ---------------------------------------------------------
constexpr int kStageTile = 2048;  // [32 samples][32 rows] x 16 bit

__device__ __forceinline__ void stage_ptile(unsigned char* img, const PTile& t, int lane) {
  const int c = lane & 31, h = lane >> 5, sw = (c >> 1) & 7;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int u = 4 * s + 2 * pp + h;  // granule = rows 4u .. 4u+3
      u32x2 v = {t.s[s][2 * pp], t.s[s][2 * pp + 1]};
      *reinterpret_cast<u32x2*>(img + c * 64 + 8 * (u ^ sw)) = v;
    }
}

// fragment of sample k-step t (samples 16t .. 16t+15) for the lane's row (lane & 31): 8 consecutive samples 8h + j
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* img, int t, int lane) {
  const int h = lane >> 5, q = (lane & 15) >> 2, p = lane & 3, u = 4 * ((lane >> 4) & 1) + p;
  u32x4 f;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int c = 16 * t + 8 * h + 4 * jj + q;
    const unsigned char* a = img + c * 64 + 8 * (u ^ ((c >> 1) & 7));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
    const u32x2 w = __builtin_bit_cast(u32x2, v);
    f[2 * jj] = w[0];
    f[2 * jj + 1] = w[1];
  }
  return f;
}

// acc[mt][kt] += dz[mt] x[kt]^T over the tile's 32 samples; rowsum[mt] += per-lane partial of sum_samples dz (row = lane & 31)
template <typename T, int KT, int MT>
__device__ __forceinline__ void dense_dw_lp(const PTile (&dz)[MT], const PTile (&x)[KT], f32x16 (&acc)[MT][KT],
                                            float (&rowsum)[MT], unsigned char* scr, int lane) {
  wave_lds_fence();  // the previous layer's transposed reads are done before the scratch is overwritten
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) stage_ptile(scr + kt * kStageTile, x[kt], lane);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) stage_ptile(scr + (KT + mt) * kStageTile, dz[mt], lane);
  wave_lds_fence();
  u32x4 b[KT][2];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int t = 0; t < 2; ++t) b[kt][t] = tr_frag(scr + kt * kStageTile, t, lane);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const u32x4 a = tr_frag(scr + (KT + mt) * kStageTile, t, lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) rowsum[mt] = T::sum2(a[q], rowsum[mt]);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[mt][kt] = T::mfma(a, b[kt][t], acc[mt][kt]);
    }
  }
}

template <int MT, int KT>
__device__ __forceinline__ void scale_acc(f32x16 (&acc)[MT][KT], float (&rowsum)[MT], float k) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    rowsum[mt] *= k;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][kt][r] *= k;
  }
}

template <typename T>
__device__ __forceinline__ PTile sh_ptile(const float* __restrict__ dirs, int64_t ray, int h) {
  const f32x16 t = sh_tile(dirs, ray, h);  // rows 0..15 in registers 0..7
  PTile p = to_ptile<T>(t);
  p.s[1] = u32x4{0u, 0u, 0u, 0u};
  return p;
}

template <int HID>
__device__ __forceinline__ float sdf_row_lp(const f32x16 (&h1)[HID / 32], const float* wsdf, int h) {
  float part = 0.0f;
#pragma unroll
  for (int t = 0; t < HID / 32; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 w = *reinterpret_cast<const float4*>(wsdf + t * 32 + 8 * g + 4 * h);
      part += w.x * h1[t][4 * g] + w.y * h1[t][4 * g + 1] + w.z * h1[t][4 * g + 2] + w.w * h1[t][4 * g + 3];
    }
  return part + __shfl_xor(part, 32, NR_WAVE) + wsdf[HID];
}

template <int FW>
struct FeatOff {
  int64_t sl;
  int F;
  __device__ __forceinline__ int64_t operator()(int k) const {
    const int Fq = FW > 0 ? FW : F;
    return (int64_t)(k / Fq) * sl + (k % Fq);
  }
};

// ---- forward --------------------------------------------------------------------------------------------------------
template <typename T, int HID, int FW>
__global__ void __launch_bounds__(256)
field_fwd_lp_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F,
                    const float* __restrict__ dirs, int S, int rows_sm, int64_t n, float* __restrict__ feature,
                    float* __restrict__ sdf_out, float* __restrict__ alpha_out) {
  using I = LpImage<HID>;
  constexpr int HT = I::HT, HS = I::HS;
  constexpr int oF32 = I::oF1t;  // LDS: forward images, then the fp32 block
  __shared__ __attribute__((aligned(16))) unsigned char lds[oF32 + I::nF32 * 4];
  const unsigned char* image = reinterpret_cast<const unsigned char*>(fld.packed);
  copy_bytes<I::oF1t>(lds, image);
  copy_bytes<I::nF32 * 4>(lds + oF32, image + I::oF32);
  __syncthreads();
  const float* fb = reinterpret_cast<const float*>(lds + oF32);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const float beta = fabsf(fld.beta[0]) + kBetaMin;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  const FeatOff<FW> foff{sl, F};
  // the next tile's features are requested before the current tile's chain of five layers starts: the chain is one
  // dependent sequence per wave, a tile's 16 loads would otherwise be exposed latency in front of it
  f32x16 xn[1];
  {
    const int64_t t0 = (int64_t)blockIdx.x * 4 + wave;
    const int64_t s0 = t0 * 32 + i;
    load_rows<32>(xn, feats + (t0 < tiles && s0 < n ? s0 * sn : 0), t0 < tiles && s0 < n, h, foff);
  }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[1], h1[HT], e[1], f1[HT], f2[HT], o[1];
    x0[0] = xn[0];
    {
      const int64_t tn = tile + (int64_t)gridDim.x * 4, sn_ = tn * 32 + i;
      const bool vn = tn < tiles && sn_ < n;
      load_rows<32>(xn, feats + (vn ? sn_ * sn : 0), vn, h, foff);
    }
    PTile xp[1] = {to_ptile<T>(x0[0])};
    dense_lp<T, 2, HT>(xp, h1, lds + I::oG1f, fb + I::bG1, lane, h);
    PTile h1p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(h1[t]); h1p[t] = to_ptile<T>(h1[t]); }
    const float sdf = sdf_row_lp<HID>(h1, fb + I::wSdf, h);
    dense_lp<T, HS, 1>(h1p, e, lds + I::oG2f, fb + I::bG2, lane, h);
    const NrRowMap rm = nr_row_map(valid ? smp : 0, n, S, rows_sm);
    PTile cat[2] = {to_ptile<T>(e[0]), (fld.sample_dirs != nullptr ? sh_ptile<T>(fld.sample_dirs, rm.out, h) : sh_ptile<T>(dirs, rm.ray, h))};
    dense_lp<T, 3, HT>(cat, f1, lds + I::oF1f, fb + I::bF1, lane, h);
    PTile f1p[HT], f2p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(f1[t]); f1p[t] = to_ptile<T>(f1[t]); }
    dense_lp<T, HS, HT>(f1p, f2, lds + I::oF2f, fb + I::bF2, lane, h);
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(f2[t]); f2p[t] = to_ptile<T>(f2[t]); }
    dense_lp<T, HS, 1>(f2p, o, lds + I::oF3f, fb + I::bF3, lane, h);
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(e[0][4 * q] + o[0][4 * q], e[0][4 * q + 1] + o[0][4 * q + 1],
                               e[0][4 * q + 2] + o[0][4 * q + 2], e[0][4 * q + 3] + o[0][4 * q + 3]);
        *reinterpret_cast<float4*>(feature + rm.out * kC + 8 * q + 4 * h) = v;
      }
      if (h == 0) {
        sdf_out[rm.out] = sdf;
        alpha_out[rm.out] = 1.0f / (1.0f + expf(sdf * beta));
      }
    }
  }
}

// ---- forward with the main grid's gather inside (SURVEY north star: "LDS-staged per-level features" -- here they never
// leave the registers).  NeuRadar's main grid: L = 8 levels x F = 4 features = the MLP's 32 inputs.  Lane (sample i, half h) of
// the 32-sample tile holds input rows {0-3, 8-11, 16-19, 24-27} + 4 h in the MFMA layout = levels {h, h + 2, h + 4, h + 6}:
// it gathers exactly those four levels of its sample (4 x 8 corners x 16 bytes), interpolates them with encode_level's
// arithmetic -- the same values nr_hash_encode_fwd would store -- and feeds the chain of five layers directly; the level-major
// [L, n, F] copy the backward recomputes from is written on the way (16 bytes per lane and level) instead of being written by
// one launch and read back by the next.
template <typename T, int HID>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
field_fwd_gather_lp_kernel(nr_field_t fld, const float* __restrict__ x01, const float* __restrict__ std01,
                           const float* __restrict__ table, const float* __restrict__ scalings, int log2T,
                           float* __restrict__ feats_out, int64_t sl, const float* __restrict__ dirs, int S, int rows_sm, int64_t n,
                           float* __restrict__ feature, float* __restrict__ sdf_out, float* __restrict__ alpha_out) {
  using I = LpImage<HID>;
  constexpr int HT = I::HT, HS = I::HS;
  constexpr int oF32 = I::oF1t;
  __shared__ __attribute__((aligned(16))) unsigned char lds[oF32 + I::nF32 * 4];
  const unsigned char* image = reinterpret_cast<const unsigned char*>(fld.packed);
  copy_bytes<I::oF1t>(lds, image);
  copy_bytes<I::nF32 * 4>(lds + oF32, image + I::oF32);
  __syncthreads();
  const float* fb = reinterpret_cast<const float*>(lds + oF32);
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const float beta = fabsf(fld.beta[0]) + kBetaMin;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  float scal[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) scal[j] = scalings[2 * j + h];
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[1], h1[HT], e[1], f1[HT], f2[HT], o[1];
    {
      // all 32 corner loads of the lane's four levels are requested before the first is used (written level by level the
      // compiler waits for each level's eight loads and its store before requesting the next: four exposed round trips per
      // tile at two waves per SIMD); the arithmetic is encode_level's, in its order
      const uint32_t mask = (1u << log2T) - 1u;
      const int64_t sq = valid ? smp : 0;
      nrgrid::Corner cr[4];
      float4 cv[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cr[j] = nrgrid::make_corner(x01, sq, scal[j]);
        const float* base = table + (((int64_t)(2 * j + h) << log2T) * 4);
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // q = zs * 4 + ys * 2 + xs; s = 0: ceil corner, 1: floor corner (encode_level's loops)
          const int iz = (q & 4) ? cr[j].lo[2] : cr[j].hi[2], iy = (q & 2) ? cr[j].lo[1] : cr[j].hi[1];
          const int ix = (q & 1) ? cr[j].lo[0] : cr[j].hi[0];
          cv[j][q] = *reinterpret_cast<const float4*>(base + (int64_t)nr_hash3(ix, iy, iz, mask) * 4);
        }
      }
      const float sd = std01[sq];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float wx = cr[j].w[0], wy = cr[j].w[1], wz = cr[j].w[2];
        float feat[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          auto el = [&](int q) { const float4 t = cv[j][q]; return f == 0 ? t.x : f == 1 ? t.y : f == 2 ? t.z : t.w; };
          float az[2];
#pragma unroll
          for (int zs = 0; zs < 2; ++zs) {
            const float ay0 = el(zs * 4 + 0) * wx + el(zs * 4 + 1) * (1.0f - wx);  // ys = 0 (ceil y): x ceil * w + x floor * (1 - w)
            const float ay1 = el(zs * 4 + 2) * wx + el(zs * 4 + 3) * (1.0f - wx);
            az[zs] = ay0 * wy + ay1 * (1.0f - wy);
          }
          const float r = 1.0f / fmaxf(scal[j] * 2.0f * sd, 1.0f);  // neurad_encoding.py:314
          feat[f] = valid ? (az[0] * wz + az[1] * (1.0f - wz)) * r : 0.0f;
          x0[0][4 * j + f] = feat[f];
        }
        if (valid && feats_out != nullptr)
          *reinterpret_cast<float4*>(feats_out + (int64_t)(2 * j + h) * sl + smp * 4) = make_float4(feat[0], feat[1], feat[2], feat[3]);
      }
    }
    PTile xp[1] = {to_ptile<T>(x0[0])};
    dense_lp<T, 2, HT>(xp, h1, lds + I::oG1f, fb + I::bG1, lane, h);
    PTile h1p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(h1[t]); h1p[t] = to_ptile<T>(h1[t]); }
    const float sdf = sdf_row_lp<HID>(h1, fb + I::wSdf, h);
    dense_lp<T, HS, 1>(h1p, e, lds + I::oG2f, fb + I::bG2, lane, h);
    const NrRowMap rm = nr_row_map(valid ? smp : 0, n, S, rows_sm);
    PTile cat[2] = {to_ptile<T>(e[0]), (fld.sample_dirs != nullptr ? sh_ptile<T>(fld.sample_dirs, rm.out, h) : sh_ptile<T>(dirs, rm.ray, h))};
    dense_lp<T, 3, HT>(cat, f1, lds + I::oF1f, fb + I::bF1, lane, h);
    PTile f1p[HT], f2p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(f1[t]); f1p[t] = to_ptile<T>(f1[t]); }
    dense_lp<T, HS, HT>(f1p, f2, lds + I::oF2f, fb + I::bF2, lane, h);
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(f2[t]); f2p[t] = to_ptile<T>(f2[t]); }
    dense_lp<T, HS, 1>(f2p, o, lds + I::oF3f, fb + I::bF3, lane, h);
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(e[0][4 * q] + o[0][4 * q], e[0][4 * q + 1] + o[0][4 * q + 1],
                               e[0][4 * q + 2] + o[0][4 * q + 2], e[0][4 * q + 3] + o[0][4 * q + 3]);
        *reinterpret_cast<float4*>(feature + rm.out * kC + 8 * q + 4 * h) = v;
      }
      if (h == 0) {
        sdf_out[rm.out] = sdf;
        alpha_out[rm.out] = 1.0f / (1.0f + expf(sdf * beta));
      }
    }
  }
}

// ---- backward, feature half: recompute the forward, backward through mlp_feature and the sigmoid ---------------------
template <typename T, int HID, int FW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HID == 32 ? 2 : 1)))
field_bwd_feat_lp_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F,
                         const float* __restrict__ dirs, int S, int rows_sm, int64_t n, const float* __restrict__ g_feature,
                         const float* __restrict__ g_alpha, const float* __restrict__ g_sdf, float* __restrict__ ws,
                         float* __restrict__ slab) {
  using I = LpImage<HID>;
  using G = FieldImage<32, HID>;
  constexpr int HT = I::HT, HS = I::HS;
  // LDS: [G1f G2f F1f F2f][F1t F2t F3t][fp32 block][staging / gradient image]
  constexpr int szA = I::oF3f, szB = I::oG1t - I::oF1t, oB = szA, oF32 = oB + szB, oScr = oF32 + I::nF32 * 4;
  constexpr int kImg = G::F1::G_SIZE + G::F2::G_SIZE + G::F3::G_SIZE + 2;
  constexpr int kScrPerWave = 4 * kStageTile;
  constexpr int kScrBytes = 4 * kScrPerWave > kImg * 4 ? 4 * kScrPerWave : kImg * 4;
  constexpr int oF1 = 0, oF2 = oF1 + G::F1::G_SIZE, oF3 = oF2 + G::F2::G_SIZE, oBeta = oF3 + G::F3::G_SIZE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[oScr + kScrBytes];
  const unsigned char* image = reinterpret_cast<const unsigned char*>(fld.packed);
  copy_bytes<szA>(lds, image);
  copy_bytes<szB>(lds + oB, image + I::oF1t);
  copy_bytes<I::nF32 * 4>(lds + oF32, image + I::oF32);
  __syncthreads();
  const float* fb = reinterpret_cast<const float*>(lds + oF32);
  const unsigned char* wF1t = lds + oB, *wF2t = wF1t + I::szF1t, *wF3t = wF2t + I::szF2t;
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  unsigned char* scr = lds + oScr + wave * kScrPerWave;
  const float beta_raw = fld.beta[0];
  const float beta = fabsf(beta_raw) + kBetaMin;
  // loss scale of the 16-bit domain: the caller's dynamic scale (nr_amp state) when it is given, the static one otherwise
  const float gs = fld.amp != nullptr ? fld.amp[NR_AMP_SCALE] : (fld.grad_scale > 0.0f ? fld.grad_scale : 1.0f), inv_gs = 1.0f / gs;
  f32x16 aF1[HT][2], aF2[HT][HT], aF3[1][HT];
  float bF1[HT], bF2[HT], bF3[1], d_beta = 0.0f;
#pragma unroll
  for (int t = 0; t < HT; ++t) { zero_tiles(aF1[t]); zero_tiles(aF2[t]); bF1[t] = bF2[t] = 0.0f; }
  zero_tiles(aF3[0]);
  bF3[0] = 0.0f;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  const FeatOff<FW> foff{sl, F};
  f32x16 xn[1];  // the next tile's features, requested one tile ahead (see field_fwd_lp_kernel)
  {
    const int64_t t0 = (int64_t)blockIdx.x * 4 + wave, s0 = t0 * 32 + i;
    load_rows<32>(xn, feats + (t0 < tiles && s0 < n ? s0 * sn : 0), t0 < tiles && s0 < n, h, foff);
  }
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    const NrRowMap rm = nr_row_map(valid ? smp : 0, n, S, rows_sm);
    PTile cat[2], f1p[HT], f2p[HT];
    float sdf;
    {  // forward again
      f32x16 x0[1], h1[HT], e[1], f1[HT], f2[HT];
      x0[0] = xn[0];
      {
        const int64_t tn = tile + (int64_t)gridDim.x * 4, sn_ = tn * 32 + i;
        const bool vn = tn < tiles && sn_ < n;
        load_rows<32>(xn, feats + (vn ? sn_ * sn : 0), vn, h, foff);
      }
      PTile xp[1] = {to_ptile<T>(x0[0])};
      dense_lp<T, 2, HT>(xp, h1, lds + I::oG1f, fb + I::bG1, lane, h);
      PTile h1p[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) { relu_tile(h1[t]); h1p[t] = to_ptile<T>(h1[t]); }
      sdf = sdf_row_lp<HID>(h1, fb + I::wSdf, h);
      dense_lp<T, HS, 1>(h1p, e, lds + I::oG2f, fb + I::bG2, lane, h);
      cat[0] = to_ptile<T>(e[0]);
      cat[1] = (fld.sample_dirs != nullptr ? sh_ptile<T>(fld.sample_dirs, rm.out, h) : sh_ptile<T>(dirs, rm.ray, h));
      dense_lp<T, 3, HT>(cat, f1, lds + I::oF1f, fb + I::bF1, lane, h);
#pragma unroll
      for (int t = 0; t < HT; ++t) { relu_tile(f1[t]); f1p[t] = to_ptile<T>(f1[t]); }
      dense_lp<T, HS, HT>(f1p, f2, lds + I::oF2f, fb + I::bF2, lane, h);
#pragma unroll
      for (int t = 0; t < HT; ++t) { relu_tile(f2[t]); f2p[t] = to_ptile<T>(f2[t]); }
    }
    f32x16 d_o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = valid ? *reinterpret_cast<const float4*>(g_feature + rm.out * kC + 8 * q + 4 * h) : make_float4(0, 0, 0, 0);
      d_o[4 * q] = v.x * gs; d_o[4 * q + 1] = v.y * gs; d_o[4 * q + 2] = v.z * gs; d_o[4 * q + 3] = v.w * gs;
    }
    PTile d_op[1] = {to_ptile<T>(d_o)};
    dense_dw_lp<T, HT, 1>(d_op, f2p, aF3, bF3, scr, lane);                 // layers[2]: o = V3 f2 + b
    f32x16 d_f2[HT], d_f1[HT], d_cat[1];
    dense_lp<T, 2, HT>(d_op, d_f2, wF3t, nullptr, lane, h);
    PTile d_f2p[HT], d_f1p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_mask_packed(d_f2[t], f2p[t]); d_f2p[t] = to_ptile<T>(d_f2[t]); }
    dense_dw_lp<T, HT, HT>(d_f2p, f1p, aF2, bF2, scr, lane);               // layers[1]
    dense_lp<T, HS, HT>(d_f2p, d_f1, wF2t, nullptr, lane, h);
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_mask_packed(d_f1[t], f1p[t]); d_f1p[t] = to_ptile<T>(d_f1[t]); }
    dense_dw_lp<T, 2, HT>(d_f1p, cat, aF1, bF1, scr, lane);                // layers[0]: input [e ; sh]
    dense_lp<T, HS, 1>(d_f1p, d_cat, wF1t, nullptr, lane, h);
    // alpha = sigmoid(-sdf * beta)
    const float ga = valid ? g_alpha[rm.out] : 0.0f;
    const float a = 1.0f / (1.0f + expf(sdf * beta));
    const float dsig = ga * a * (1.0f - a);
    float d_sdf = dsig * (-beta);
    if (g_sdf != nullptr && valid) d_sdf += g_sdf[rm.out];
    if (h == 0) d_beta += dsig * (-sdf) * (beta_raw >= 0.0f ? 1.0f : -1.0f);
    {
      float* w = ws + tile * kWsTile;
#pragma unroll
      for (int r = 0; r < 16; ++r) w[r * 64 + lane] = valid ? (d_o[r] + d_cat[0][r]) * inv_gs : 0.0f;
      w[16 * 64 + lane] = valid ? d_sdf : 0.0f;
    }
  }
  scale_acc(aF1, bF1, inv_gs);
  scale_acc(aF2, bF2, inv_gs);
  scale_acc(aF3, bF3, inv_gs);
  d_beta = nr_wave_sum(d_beta);
  float* img = reinterpret_cast<float*>(lds + oScr);
  __syncthreads();  // every wave is done with its staging scratch
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      const bool first = w == 0;
      merge_dw<kC + kSH, HID>(aF1, bF1, img + oF1, first, i, h);
      merge_dw<HID, HID>(aF2, bF2, img + oF2, first, i, h);
      merge_dw<HID, kC>(aF3, bF3, img + oF3, first, i, h);
      if (lane == 0) img[oBeta] = first ? d_beta : img[oBeta] + d_beta;
    }
    __syncthreads();
  }
  float* out = slab + (int64_t)blockIdx.x * G::G_TOTAL + G::gF1;
  if (threadIdx.x == 0) img[oBeta + 1] = 0.0f;
  __syncthreads();
  for (int k = threadIdx.x; k < kImg; k += blockDim.x) out[k] = img[k];
}

// ---- backward, geometry half ------------------------------------------------------------------------------------------
template <typename T, int HID, int FW>
__global__ void __launch_bounds__(256)
field_bwd_geo_lp_kernel(nr_field_t fld, const float* __restrict__ feats, int64_t sn, int64_t sl, int F, int64_t n,
                        const float* __restrict__ ws, float* __restrict__ g_feats, float* __restrict__ slab) {
  using I = LpImage<HID>;
  using G = FieldImage<32, HID>;
  constexpr int HT = I::HT;
  // LDS: [G1f][G1t G2t][fp32 block][staging / gradient image]
  constexpr int szA = I::szG1f, szB = I::oF32 - I::oG1t, oB = szA, oF32 = oB + szB, oScr = oF32 + I::nF32 * 4;
  constexpr int kImg = G::G1::G_SIZE + G::G2::G_SIZE + G::SDF;
  constexpr int kScrPerWave = 4 * kStageTile;
  constexpr int kScrBytes = 4 * kScrPerWave > kImg * 4 ? 4 * kScrPerWave : kImg * 4;
  constexpr int oG1 = 0, oG2 = oG1 + G::G1::G_SIZE, oSdf = oG2 + G::G2::G_SIZE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[oScr + kScrBytes];
  const unsigned char* image = reinterpret_cast<const unsigned char*>(fld.packed);
  copy_bytes<szA>(lds, image + I::oG1f);
  copy_bytes<szB>(lds + oB, image + I::oG1t);
  copy_bytes<I::nF32 * 4>(lds + oF32, image + I::oF32);
  __syncthreads();
  const float* fb = reinterpret_cast<const float*>(lds + oF32);
  const unsigned char* wG1t = lds + oB, *wG2t = wG1t + I::szG1t;
  const int lane = nr_lane(), i = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  unsigned char* scr = lds + oScr + wave * kScrPerWave;
  // loss scale of the 16-bit domain: the caller's dynamic scale (nr_amp state) when it is given, the static one otherwise
  const float gs = fld.amp != nullptr ? fld.amp[NR_AMP_SCALE] : (fld.grad_scale > 0.0f ? fld.grad_scale : 1.0f), inv_gs = 1.0f / gs;
  f32x16 aG1[HT][1], aG2[1][HT], aSdf[HT];
  float bG1[HT], bG2[1], bSdf = 0.0f;
#pragma unroll
  for (int t = 0; t < HT; ++t) { zero_tiles(aG1[t]); bG1[t] = 0.0f; }
  zero_tiles(aG2[0]); zero_tiles(aSdf);
  bG2[0] = 0.0f;
  const int64_t tiles = nr_cdiv_dev(n, 32);
  const FeatOff<FW> foff{sl, F};
  f32x16 xn[1], wn;  // the next tile's features and its d_e / d_sdf rows from the feature half, one tile ahead
  float wn_sdf = 0.0f;
  bool bad = false;
  auto request = [&](int64_t t) {
    const int64_t s_ = t * 32 + i;
    const bool v = t < tiles && s_ < n;
    load_rows<32>(xn, feats + (v ? s_ * sn : 0), v, h, foff);
    const float* wt = ws + (t < tiles ? t : 0) * kWsTile;
#pragma unroll
    for (int r = 0; r < 16; ++r) wn[r] = wt[r * 64 + lane];
    wn_sdf = wt[16 * 64 + lane];
  };
  request((int64_t)blockIdx.x * 4 + wave);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t smp = tile * 32 + i;
    const bool valid = smp < n;
    f32x16 x0[1], h1[HT], d_e, d_h1[HT], d_x0[1];
    x0[0] = xn[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) d_e[r] = wn[r] * gs;
    const float d_sdf = wn_sdf;
    request(tile + (int64_t)gridDim.x * 4);
    PTile xp[1] = {to_ptile<T>(x0[0])};
    dense_lp<T, 2, HT>(xp, h1, lds, fb + I::bG1, lane, h);
    PTile h1p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) { relu_tile(h1[t]); h1p[t] = to_ptile<T>(h1[t]); }
    if (h == 0) bSdf += d_sdf;
    PTile d_ep[1] = {to_ptile<T>(d_e)};
    dense_dw_lp<T, HT, 1>(d_ep, h1p, aG2, bG2, scr, lane);   // mlp_geo.layers[1] rows 1..C
    dense_lp<T, 2, HT>(d_ep, d_h1, wG2t, nullptr, lane, h);
    const float d_sdf_s = d_sdf * gs;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w = *reinterpret_cast<const float4*>(fb + I::wSdf + t * 32 + 8 * g + 4 * h);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          aSdf[t][4 * g + k] += d_sdf * h1[t][4 * g + k];
          d_h1[t][4 * g + k] += wv[k] * d_sdf_s;
        }
      }
    PTile d_h1p[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) d_h1[t][r] = h1[t][r] > 0.0f ? d_h1[t][r] : 0.0f;
      d_h1p[t] = to_ptile<T>(d_h1[t]);
    }
    dense_dw_lp<T, 1, HT>(d_h1p, xp, aG1, bG1, scr, lane);   // mlp_geo.layers[0]
    dense_lp<T, I::HS, 1>(d_h1p, d_x0, wG1t, nullptr, lane, h);
    if (valid) {
      float* gf = g_feats + smp * sn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = d_x0[0][r] * inv_gs;
        bad |= !isfinite(v);
        gf[foff(rowmap(r, 0) + 4 * h)] = v;
      }
    }
  }
  // found-inf (GradScaler): an overflowed 16-bit operand anywhere upstream of this row shows here as inf / NaN -- the flag makes
  // the optimizers of fld.amp_groups skip the step (every writer stores the same value: no atomic)
  if (fld.amp != nullptr && __any(bad) && lane == 0)
    for (int g = 0; g < NR_AMP_MAX_GROUPS; ++g)
      if ((fld.amp_groups >> g) & 1u) fld.amp[NR_AMP_FOUND + g] = 1.0f;
  scale_acc(aG1, bG1, inv_gs);
  scale_acc(aG2, bG2, inv_gs);
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float v = aSdf[t][s];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, NR_WAVE);
      aSdf[t][s] = v;
    }
  bSdf = nr_wave_sum(bSdf);
  float* img = reinterpret_cast<float*>(lds + oScr);
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      const bool first = w == 0;
      merge_dw<32, HID>(aG1, bG1, img + oG1, first, i, h);
      merge_dw<HID, kC>(aG2, bG2, img + oG2, first, i, h);
      if (i == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const int k = t * 32 + rowmap(s, 0) + 4 * h;
            img[oSdf + k] = first ? aSdf[t][s] : img[oSdf + k] + aSdf[t][s];
          }
        if (h == 0) img[oSdf + HID] = first ? bSdf : img[oSdf + HID] + bSdf;
      }
    }
    __syncthreads();
  }
  float* out = slab + (int64_t)blockIdx.x * G::G_TOTAL + G::gG1;
  for (int k = HID + 1 + threadIdx.x; k < G::SDF; k += blockDim.x) img[oSdf + k] = 0.0f;  // padding
  __syncthreads();
  for (int k = threadIdx.x; k < kImg; k += blockDim.x) out[k] = img[k];
}

// ---- weight image ---------------------------------------------------------------------------------------------------
// forward block (mt, ks), lane (r, h), element j: W[row0 + 32 mt + r][16 ks + 8 (j>>2) + 4 h + (j&3)]
template <typename E>
__device__ void pack_fwd(unsigned char* dst, const float* __restrict__ W, int ld, int row0, int m_act, int k_act, int MT, int KS) {
  E* d = reinterpret_cast<E*>(dst);
  for (int idx = threadIdx.x; idx < MT * KS * 512; idx += blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9, ks = blk % KS, mt = blk / KS;
    const int m = mt * 32 + (lane & 31), k = 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    d[idx] = (E)((m < m_act && k < k_act) ? W[(int64_t)(row0 + m) * ld + k] : 0.0f);
  }
}
// transposed block (kt, ms), lane (r, h), element j: W[row0 + 16 ms + 8 (j>>2) + 4 h + (j&3)][32 kt + r]
template <typename E>
__device__ void pack_tr(unsigned char* dst, const float* __restrict__ W, int ld, int row0, int m_act, int k_act, int KT, int MS) {
  E* d = reinterpret_cast<E*>(dst);
  for (int idx = threadIdx.x; idx < KT * MS * 512; idx += blockDim.x) {
    const int j = idx & 7, lane = (idx >> 3) & 63, blk = idx >> 9, ms = blk % MS, kt = blk / MS;
    const int k = kt * 32 + (lane & 31), m = 16 * ms + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    d[idx] = (E)((m < m_act && k < k_act) ? W[(int64_t)(row0 + m) * ld + k] : 0.0f);
  }
}

template <typename T, int HID>
__global__ void __launch_bounds__(1024)
field_pack_lp_kernel(nr_field_t f, unsigned char* __restrict__ image) {
  using I = LpImage<HID>;
  using E = typename T::elem;
  constexpr int HT = I::HT, HS = I::HS;
  pack_fwd<E>(image + I::oG1f, f.geo.weight[0], 32, 0, HID, 32, HT, 2);
  pack_fwd<E>(image + I::oG2f, f.geo.weight[1], HID, 1, kC, HID, 1, HS);
  pack_fwd<E>(image + I::oF1f, f.feat.weight[0], kC + kSH, 0, HID, kC + kSH, HT, 3);
  pack_fwd<E>(image + I::oF2f, f.feat.weight[1], HID, 0, HID, HID, HT, HS);
  pack_fwd<E>(image + I::oF3f, f.feat.weight[2], HID, 0, kC, HID, 1, HS);
  pack_tr<E>(image + I::oF1t, f.feat.weight[0], kC + kSH, 0, HID, kC, 1, HS);
  pack_tr<E>(image + I::oF2t, f.feat.weight[1], HID, 0, HID, HID, HT, HS);
  pack_tr<E>(image + I::oF3t, f.feat.weight[2], HID, 0, kC, HID, HT, 2);
  pack_tr<E>(image + I::oG1t, f.geo.weight[0], 32, 0, HID, 32, 1, HS);
  pack_tr<E>(image + I::oG2t, f.geo.weight[1], HID, 1, kC, HID, HT, 2);
  float* fb = reinterpret_cast<float*>(image + I::oF32);
  for (int k = threadIdx.x; k < I::nF32; k += blockDim.x) {
    float v = 0.0f;
    if (k < I::bG2) v = k - I::bG1 < HID ? f.geo.bias[0][k - I::bG1] : 0.0f;
    else if (k < I::bF1) v = f.geo.bias[1][1 + k - I::bG2];
    else if (k < I::bF2) v = k - I::bF1 < HID ? f.feat.bias[0][k - I::bF1] : 0.0f;
    else if (k < I::bF3) v = k - I::bF2 < HID ? f.feat.bias[1][k - I::bF2] : 0.0f;
    else if (k < I::wSdf) v = f.feat.bias[2][k - I::bF3];
    else if (k < I::wSdf + HID) v = f.geo.weight[1][k - I::wSdf];
    else if (k == I::wSdf + HID) v = f.geo.bias[1][0];
    fb[k] = v;
  }
}

template <typename T, int HID>
int launch_fwd(const nr_field_t* field, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S, int rows_sm,
               int64_t n, float* feature, float* sdf, float* alpha, unsigned blocks, hipStream_t st) {
#define NR_LP_FWD(FWC)                                                                                                   \
  hipLaunchKernelGGL((field_fwd_lp_kernel<T, HID, FWC>), dim3(blocks), dim3(256), 0, st, *field, feats, sn, sl, F, dirs, S,   \
                     rows_sm, n, feature, sdf, alpha)
  if (F == 2) NR_LP_FWD(2); else if (F == 4) NR_LP_FWD(4); else NR_LP_FWD(0);
#undef NR_LP_FWD
  NR_LAUNCH_CHECK();
  return 0;
}

template <typename T, int HID>
int launch_bwd(const nr_field_t* field, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S, int rows_sm,
               int64_t n, const float* g_feature, const float* g_alpha, const float* g_sdf, float* g_feats, float* ws, float* slab,
               unsigned blocks, hipStream_t st) {
#define NR_LP_BWD(FWC)                                                                                                   \
  {                                                                                                                      \
    hipLaunchKernelGGL((field_bwd_feat_lp_kernel<T, HID, FWC>), dim3(blocks), dim3(256), 0, st, *field, feats, sn, sl, F,    \
                       dirs, S, rows_sm, n, g_feature, g_alpha, g_sdf, ws, slab);                                          \
    hipLaunchKernelGGL((field_bwd_geo_lp_kernel<T, HID, FWC>), dim3(blocks), dim3(256), 0, st, *field, feats, sn, sl, F, n,  \
                       ws, g_feats, slab);                                                                                \
  }
  if (F == 2) NR_LP_BWD(2) else if (F == 4) NR_LP_BWD(4) else NR_LP_BWD(0)
#undef NR_LP_BWD
  NR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace nrfield {

int64_t field_image_bytes_lp(int hid) { return hid == 32 ? LpImage<32>::BYTES : LpImage<64>::BYTES; }

int field_pack_lp(const nr_field_t* field, int hid, void* image, hipStream_t st) {
  unsigned char* img = static_cast<unsigned char*>(image);
  if (field->dtype == NR_DTYPE_BF16) {
    if (hid == 32) hipLaunchKernelGGL((field_pack_lp_kernel<Bf16, 32>), dim3(1), dim3(1024), 0, st, *field, img);
    else hipLaunchKernelGGL((field_pack_lp_kernel<Bf16, 64>), dim3(1), dim3(1024), 0, st, *field, img);
  } else {
    if (hid == 32) hipLaunchKernelGGL((field_pack_lp_kernel<Fp16, 32>), dim3(1), dim3(1024), 0, st, *field, img);
    else hipLaunchKernelGGL((field_pack_lp_kernel<Fp16, 64>), dim3(1), dim3(1024), 0, st, *field, img);
  }
  NR_LAUNCH_CHECK();
  return 0;
}

int field_fwd_lp(const nr_field_t* field, int hid, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S,
                 int rows_sm, int64_t n, float* feature, float* sdf, float* alpha, unsigned blocks, hipStream_t st) {
  if (field->dtype == NR_DTYPE_BF16)
    return hid == 32 ? launch_fwd<Bf16, 32>(field, feats, sn, sl, F, dirs, S, rows_sm, n, feature, sdf, alpha, blocks, st)
                     : launch_fwd<Bf16, 64>(field, feats, sn, sl, F, dirs, S, rows_sm, n, feature, sdf, alpha, blocks, st);
  return hid == 32 ? launch_fwd<Fp16, 32>(field, feats, sn, sl, F, dirs, S, rows_sm, n, feature, sdf, alpha, blocks, st)
                   : launch_fwd<Fp16, 64>(field, feats, sn, sl, F, dirs, S, rows_sm, n, feature, sdf, alpha, blocks, st);
}

int field_fwd_gather_lp(const nr_field_t* field, int hid, const float* x01, const float* std01, const float* table,
                        const float* scalings, int log2T, float* feats_out, int64_t sl, const float* dirs, int S, int rows_sm,
                        int64_t n, float* feature, float* sdf, float* alpha, unsigned blocks, hipStream_t st) {
  if (hid != 32) return NR_EINVAL;
  if (field->dtype == NR_DTYPE_BF16)
    hipLaunchKernelGGL((field_fwd_gather_lp_kernel<Bf16, 32>), dim3(blocks), dim3(256), 0, st, *field, x01, std01, table, scalings, log2T,
                       feats_out, sl, dirs, S, rows_sm, n, feature, sdf, alpha);
  else
    hipLaunchKernelGGL((field_fwd_gather_lp_kernel<Fp16, 32>), dim3(blocks), dim3(256), 0, st, *field, x01, std01, table, scalings, log2T,
                       feats_out, sl, dirs, S, rows_sm, n, feature, sdf, alpha);
  NR_LAUNCH_CHECK();
  return 0;
}

int field_bwd_lp(const nr_field_t* field, int hid, const float* feats, int64_t sn, int64_t sl, int F, const float* dirs, int S,
                 int rows_sm, int64_t n, const float* g_feature, const float* g_alpha, const float* g_sdf, float* g_feats,
                 float* ws, float* slab, unsigned blocks, hipStream_t st) {
  if (field->dtype == NR_DTYPE_BF16)
    return hid == 32 ? launch_bwd<Bf16, 32>(field, feats, sn, sl, F, dirs, S, rows_sm, n, g_feature, g_alpha, g_sdf, g_feats, ws, slab, blocks, st)
                     : launch_bwd<Bf16, 64>(field, feats, sn, sl, F, dirs, S, rows_sm, n, g_feature, g_alpha, g_sdf, g_feats, ws, slab, blocks, st);
  return hid == 32 ? launch_bwd<Fp16, 32>(field, feats, sn, sl, F, dirs, S, rows_sm, n, g_feature, g_alpha, g_sdf, g_feats, ws, slab, blocks, st)
                   : launch_bwd<Fp16, 64>(field, feats, sn, sl, F, dirs, S, rows_sm, n, g_feature, g_alpha, g_sdf, g_feats, ws, slab, blocks, st);
}

}  // namespace nrfield
