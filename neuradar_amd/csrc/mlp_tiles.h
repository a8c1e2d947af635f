// MFMA building blocks for the small NeuRadar MLPs on gfx950 (fp32 in / fp32 accumulate:
// v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain -> parity with the torch reference).
//
// Layout ("D-layout"): a wave owns a tile of 32 SAMPLES.  A [rows x 32 samples] activation block is
// kept in registers as f32x16 tiles of 32 rows: lane l = (c = l & 31, h = l >> 5) holds, for sample c,
// rows rowmap(r, h) = (r & 3) + 8 (r >> 2) + 4 h, r = 0..15 -- exactly the MFMA C/D layout.
// Because the 32x32x2 B operand wants B[k = lane>>5][col = lane&31], accumulator register s of one
// layer IS the B operand of k-step s of the next layer, provided the weights are fed in the matching
// k order.  Layers therefore chain with no LDS traffic and no cross-lane movement; weights sit in
// LDS ([row][k], odd leading dimension -> conflict-free ds_read_b32 for both W and W^T reads).
// Only weight gradients contract over samples (the lane axis) and go through a per-wave LDS scratch.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Keeps hipcc from hoisting every LDS weight read of a fully unrolled layer to the top of the kernel
// (which costs 500+ registers and spills): reads stay within one 16-MFMA group of their use.
#ifndef NR_MLP_NO_SCHED_FENCE
#define NR_MLP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NR_MLP_SCHED_FENCE() ((void)0)
#endif

namespace nrmlp {

__host__ __device__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// number of k-steps s = 0..n-1 that touch a row < valid (rowmap(s,0) is increasing in s)
__host__ __device__ constexpr int nsteps(int valid) {
  int n = 0;
  for (int s = 0; s < 16; ++s) n += rowmap(s, 0) < valid ? 1 : 0;
  return n;
}
__host__ __device__ constexpr int imin(int a, int b) { return a < b ? a : b; }

constexpr int kScrLd = 33;                 // scratch tile leading dimension (32 samples + 1 pad)
constexpr int kScrTile = 32 * kScrLd;      // floats per staged tile

// LDS image of one Linear(K -> M): zero-padded weights [MT*32][KT*32+1] followed by bias [MT*32];
// gradient image: dense [M][K] followed by [M].
template <int K, int M>
struct Layer {
  static constexpr int KT = (K + 31) / 32, MT = (M + 31) / 32;
  static constexpr int LDW = KT * 32 + 1;
  static constexpr int W_SIZE = MT * 32 * LDW;
  static constexpr int SIZE = W_SIZE + MT * 32;
  static constexpr int G_SIZE = M * K + M;
};

// cooperative (whole block) load of torch-layout weights into the LDS image.  The torch matrix has
// leading dimension k_act (<= K) and rows [row0, row0+m_act) are taken (m_act <= M), so a Linear can
// be split by output rows and narrower layers can run zero-padded on a wider template.
template <int K, int M>
__device__ __forceinline__ void load_layer(float* lds, const float* __restrict__ w, const float* __restrict__ b,
                                           int row0, int k_act = K, int m_act = M) {
  using L = Layer<K, M>;
  for (int i = threadIdx.x; i < L::SIZE; i += blockDim.x) lds[i] = 0.0f;
  __syncthreads();
  for (int i = threadIdx.x; i < m_act * k_act; i += blockDim.x) {
    const int m = i / k_act, k = i - m * k_act;
    lds[m * L::LDW + k] = w[(row0 + m) * k_act + k];
  }
  for (int i = threadIdx.x; i < m_act; i += blockDim.x) lds[L::W_SIZE + i] = b ? b[row0 + i] : 0.0f;
}

// cooperative flush of the LDS gradient image ([M][K] then [M]) into the torch-layout global
// gradients (+=).  Consecutive lanes add to consecutive addresses: contiguous atomic wave-instructions.
template <int K, int M>
__device__ __forceinline__ void flush_layer_grads(const float* lds, float* __restrict__ gw, float* __restrict__ gb,
                                                  int row0, int k_act = K, int m_act = M) {
  for (int i = threadIdx.x; i < m_act * k_act; i += blockDim.x) {
    const int m = i / k_act, k = i - m * k_act;
    unsafeAtomicAdd(gw + (row0 + m) * k_act + k, lds[m * K + k]);
  }
  if (gb)
    for (int i = threadIdx.x; i < m_act; i += blockDim.x) unsafeAtomicAdd(gb + row0 + i, lds[M * K + i]);
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// y = act(W x + b).  x: KT tiles, y: MT tiles.
template <int K, int M, bool RELU>
__device__ __forceinline__ void dense_fwd(const f32x16 (&x)[(K + 31) / 32], f32x16 (&y)[(M + 31) / 32],
                                          const float* lw, int i, int h) {
  using L = Layer<K, M>;
  const float* bias = lw + L::W_SIZE;
#pragma unroll
  for (int mt = 0; mt < L::MT; ++mt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[mt * 32 + rowmap(r, 0) + 4 * h];
    const float* wrow = lw + (mt * 32 + i) * L::LDW + 4 * h;
#pragma unroll
    for (int kt = 0; kt < L::KT; ++kt) {
      const int ns = nsteps(imin(32, K - 32 * kt));
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < ns) acc = mfma(wrow[kt * 32 + rowmap(s, 0)], x[kt][s], acc);
      NR_MLP_SCHED_FENCE();
    }
    if (RELU) {
#pragma unroll
      // relu as ONE v_max_i32 on the bit pattern (negative floats are negative ints; -0.0 -> +0.0);
      // fmaxf costs two v_max_f32 (the IEEE-mode canonicalisation of its operand comes first)
      for (int r = 0; r < 16; ++r) acc[r] = __int_as_float(max(__float_as_int(acc[r]), 0));
    }
    y[mt] = acc;
  }
}

// dx = W^T dz.  dz: MT tiles (rows >= M must be zero), dx: KT tiles.
template <int K, int M>
__device__ __forceinline__ void dense_bwd_dx(const f32x16 (&dz)[(M + 31) / 32], f32x16 (&dx)[(K + 31) / 32],
                                             const float* lw, int i, int h) {
  using L = Layer<K, M>;
#pragma unroll
  for (int kt = 0; kt < L::KT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < L::MT; ++mt) {
      const int ns = nsteps(imin(32, M - 32 * mt));
      const float* wcol = lw + (mt * 32 + 4 * h) * L::LDW + kt * 32 + i;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < ns) acc = mfma(wcol[rowmap(s, 0) * L::LDW], dz[mt][s], acc);
      NR_MLP_SCHED_FENCE();
    }
    dx[kt] = acc;
  }
}

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in order; this only pins the compiler's ordering.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__device__ __forceinline__ void stage_tile(float* scr, const f32x16& t, int i, int h) {
#pragma unroll
  for (int r = 0; r < 16; ++r) scr[(rowmap(r, 0) + 4 * h) * kScrLd + i] = t[r];
}

// dW += dz x^T, db += rowsum(dz), accumulated into the block's LDS gradient image `lg` with ds_add.
// The contraction runs over the 32 samples of the tile, i.e. over lanes, so both operands are
// staged through the wave's scratch (scrA, scrB: kScrTile floats each) and read back transposed.
// EXTRA: one more output row whose dz is a per-sample scalar (`extra`, identical in both lane
// halves); its gradient row lands at lg_extra[k] and its bias gradient at lg_extra[K].
template <int K, int M, bool EXTRA>
__device__ __forceinline__ void dense_bwd_dw(const f32x16 (&dz)[(M + 31) / 32], const f32x16 (&x)[(K + 31) / 32],
                                             float* lg, float extra, float* lg_extra, float* scrA, float* scrB,
                                             float* scrE, int i, int h) {
  using L = Layer<K, M>;
  if (EXTRA) {
    if (h == 0) scrE[i] = extra;
  }
#pragma unroll
  for (int kt = 0; kt < L::KT; ++kt) {
    wave_lds_fence();
    stage_tile(scrB, x[kt], i, h);
#pragma unroll
    for (int mt = 0; mt < L::MT; ++mt) {
      wave_lds_fence();
      stage_tile(scrA, dz[mt], i, h);
      wave_lds_fence();
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      float rowsum = 0.0f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = scrA[i * kScrLd + 2 * s + h];
        const float b = scrB[i * kScrLd + 2 * s + h];
        acc = mfma(a, b, acc);
        rowsum += a;
      }
      const int k = kt * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + rowmap(r, 0) + 4 * h;
        if (m < M && k < K) atomicAdd(&lg[m * K + k], acc[r]);
      }
      if (kt == 0 && mt * 32 + i < M) atomicAdd(&lg[M * K + mt * 32 + i], rowsum);
    }
    if (EXTRA) {
      wave_lds_fence();
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      float rowsum = 0.0f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float e = scrE[2 * s + h];
        const float a = i == 0 ? e : 0.0f;
        const float b = scrB[i * kScrLd + 2 * s + h];
        acc = mfma(a, b, acc);
        rowsum += a;
      }
      const int k = kt * 32 + i;
      if (h == 0 && k < K) atomicAdd(&lg_extra[k], acc[0]);  // D row 0 lives in (h = 0, reg 0)
      if (kt == 0 && i == 0) atomicAdd(&lg_extra[K], rowsum);
    }
  }
  wave_lds_fence();
}

// Register-accumulating variant: acc[mt][kt] (MFMA C/D tiles of dW) and rowsum[mt] (bias-gradient
// partials) persist in registers across all tiles of a wave -- LDS float atomics (ds_add_f32) turned
// out to cost more than the MFMAs themselves.  All KT input tiles and MT gradient tiles of the layer
// are staged once (scr: (KT+MT) * kScrTile floats), one fence, then MT*KT*16 MFMAs.
template <int K, int M>
__device__ __forceinline__ void dense_bwd_dw_reg(const f32x16 (&dz)[(M + 31) / 32], const f32x16 (&x)[(K + 31) / 32],
                                                 f32x16 (&acc)[(M + 31) / 32][(K + 31) / 32],
                                                 float (&rowsum)[(M + 31) / 32], float* scr, int i, int h) {
  using L = Layer<K, M>;
  wave_lds_fence();  // the previous layer's transposed reads are done before the scratch is overwritten
#pragma unroll
  for (int kt = 0; kt < L::KT; ++kt) stage_tile(scr + kt * kScrTile, x[kt], i, h);
#pragma unroll
  for (int mt = 0; mt < L::MT; ++mt) stage_tile(scr + (L::KT + mt) * kScrTile, dz[mt], i, h);
  wave_lds_fence();
#pragma unroll
  for (int mt = 0; mt < L::MT; ++mt) {
    const float* A = scr + (L::KT + mt) * kScrTile + i * kScrLd + h;
    float a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      a[s] = A[2 * s];
      rowsum[mt] += a[s];
    }
#pragma unroll
    for (int kt = 0; kt < L::KT; ++kt) {
      const float* B = scr + kt * kScrTile + i * kScrLd + h;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[mt][kt] = mfma(a[s], B[2 * s], acc[mt][kt]);
      NR_MLP_SCHED_FENCE();
    }
  }
}

// Merge one wave's register accumulators into the block's gradient image `img` ([M][K] then [M]),
// plain stores: the caller serialises the waves (first wave assigns, later waves add).
template <int K, int M>
__device__ __forceinline__ void merge_dw(const f32x16 (&acc)[(M + 31) / 32][(K + 31) / 32],
                                         const float (&rowsum)[(M + 31) / 32], float* img, bool first, int i, int h) {
  using L = Layer<K, M>;
#pragma unroll
  for (int mt = 0; mt < L::MT; ++mt) {
#pragma unroll
    for (int kt = 0; kt < L::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + rowmap(r, 0) + 4 * h, k = kt * 32 + i;
        if (m < M && k < K) img[m * K + k] = first ? acc[mt][kt][r] : img[m * K + k] + acc[mt][kt][r];
      }
    const float rs = rowsum[mt] + __shfl_xor(rowsum[mt], 32, 64);  // lane halves hold even / odd samples
    const int m = mt * 32 + i;
    if (h == 0 && m < M) img[M * K + m] = first ? rs : img[M * K + m] + rs;
  }
}

template <int N>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[N]) {
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[n][r] = 0.0f;
}

}  // namespace nrmlp
