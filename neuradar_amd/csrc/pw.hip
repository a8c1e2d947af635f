// The RGB decoder's POINTWISE convolutions on channels-last activations (models/neuradar.py:225-240: Conv2d(C_in, 32, 1) + ReLU
// at its head, ConvTranspose2d(32, 32, 3, stride=3) between the block pairs -- every input pixel owns its 3 x 3 output block, so
// it is a 32 -> 9 * 32 pointwise layer with scattered rows -- and Conv2d(32, 3, 1) + Sigmoid at its tail): y[p, n] = act(b[n] +
// sum_k x[p, k] W[n, k]).  Together < 1 % of the CNN's FLOPs (0.2 GFLOP per step: no MFMA needed -- one lane per pixel, weights
// broadcast from LDS in fp32, fp32 accumulation) but 30 of its library launches per step (igemm / CK kernels with their
// zero-fills, bias adds, bias reductions, casts, and the activations' own kernels).
//   nr_pw_fwd         y = act(x W^T + b); x fp32 or 16-bit, y 16-bit or fp32; act none / ReLU / sigmoid
//   nr_pw_bwd_data    dx = (dy * act'(y)) W  (x's type; optionally times a device scalar: the inverse loss scale)
//   nr_pw_bwd_weight  dW += (dy * act'(y))^T x, db += column sums: per-block partials + a reduce launch (like nr_conv7_wgrad)
// `transposed`: W is the ConvTranspose2d weight in its channels-last memory [k][ky][kx][o]; y / dy then have 3H x 3W pixels.
#include "nr_common.h"

namespace {

enum { kF32 = 0, kBf16 = 1, kF16 = 2 };
enum { kActNone = 0, kActRelu = 1, kActSigmoid = 2 };

template <typename E>
__device__ __forceinline__ float to_f(E v) { return (float)v; }

// 8 consecutive elements starting at p (16-byte aligned for 16-bit types, 32-byte for fp32) as floats
template <typename E>
__device__ __forceinline__ void load8(const E* __restrict__ p, float (&v)[8]) {
  if constexpr (sizeof(E) == 4) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    typedef E e8 __attribute__((ext_vector_type(8)));
    const e8 a = *reinterpret_cast<const e8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
  }
}
template <typename E>
__device__ __forceinline__ void store8(E* __restrict__ p, const float (&v)[8]) {
  if constexpr (sizeof(E) == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    typedef E e8 __attribute__((ext_vector_type(8)));
    e8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (E)v[j];
    *reinterpret_cast<e8*>(p) = a;
  }
}

// derivative factor of the activation from its OUTPUT
__device__ __forceinline__ float act_grad(int act, float y) {
  return act == kActRelu ? (y > 0.0f ? 1.0f : 0.0f) : (act == kActSigmoid ? y * (1.0f - y) : 1.0f);
}

struct PwShape {
  int P;            // input pixels (images * H * W); every element offset below fits 32 bits (checked on the host)
  int K, N;         // in / out channels per pixel (transposed: N = 9 * out channels)
  int transposed;   // ConvTranspose2d(3, stride 3): output pixel (3y + ky, 3x + kx), channels o <- n = (ky * 3 + kx) * O + o
  int H, W;         // input image size (transposed only)
  int O;            // out channels (= N, or N / 9)
};

// element offset in y / dy of output n of input pixel p = pix_base(p) + tap_off(n): all 32-bit (a 64-bit division is ~100 instructions)
__device__ __forceinline__ int pix_base(const PwShape& s, int p) {
  if (!s.transposed) return p * s.N;
  const int hw = s.H * s.W, img = p / hw, rem = p - img * hw, yy = rem / s.W, xx = rem - yy * s.W;
  return ((img * 3 * s.H + 3 * yy) * 3 * s.W + 3 * xx) * s.O;
}
__device__ __forceinline__ int tap_off(const PwShape& s, int n) {
  if (!s.transposed) return n;
  const int tap = n / s.O, o = n - tap * s.O, ky = tap / 3, kx = tap - 3 * ky;
  return (ky * 3 * s.W + kx) * s.O + o;
}

// weights into LDS as fp32 [n][K] (+ bias [O] behind them); W16: [N][K], or transposed-convolution memory [K][9][O] = [K][N]
template <typename E>
__device__ __forceinline__ void load_weights(float* lw, const E* __restrict__ w, const E* __restrict__ b, const PwShape& s) {
  for (int e = threadIdx.x; e < s.N * s.K; e += blockDim.x) {
    const int n = e / s.K, k = e - n * s.K;
    lw[e] = (float)(s.transposed ? w[k * s.N + n] : w[e]);
  }
  for (int o = threadIdx.x; o < s.O; o += blockDim.x) lw[s.N * s.K + o] = b ? (float)b[o] : 0.0f;
  __syncthreads();
}

constexpr int kMaxK = 48;

// ------------------------------------------------------------------------------------------------ forward
// one thread per (pixel, group of 8 outputs): item = g * P + p -- consecutive lanes are consecutive pixels of ONE output group, so
// the weight reads are LDS broadcasts; the pixel's row is re-read by every group (from L2: 64-192 bytes)
template <typename EX, typename EY, typename EW, int K>
__global__ void __launch_bounds__(256)
pw_fwd_kernel(const EX* __restrict__ x, const EW* __restrict__ w, const EW* __restrict__ b, EY* __restrict__ y, PwShape s, int act) {
  extern __shared__ float pw_lds[];
  load_weights(pw_lds, w, b, s);
  const float* bias = pw_lds + s.N * K;
  const int G = (s.N + 7) / 8, items = s.P * G;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
    const int g = it / s.P, p = it - g * s.P;
    const int n0 = 8 * g, nn = s.N - n0 < 8 ? s.N - n0 : 8;
    const int o0 = n0 % s.O;  // (groups of 8 do not straddle a tap: O is a multiple of 8 when transposed)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = j < nn ? bias[o0 + j] : 0.0f;
#pragma unroll
    for (int k8 = 0; k8 < K / 8; ++k8) {
      float v[8];
      load8(x + p * K + 8 * k8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < nn) {
          const float* wr = pw_lds + (n0 + j) * K + 8 * k8;
          const float4 wa = *reinterpret_cast<const float4*>(wr), wb = *reinterpret_cast<const float4*>(wr + 4);
          acc[j] += v[0] * wa.x + v[1] * wa.y + v[2] * wa.z + v[3] * wa.w + v[4] * wb.x + v[5] * wb.y + v[6] * wb.z + v[7] * wb.w;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (act == kActRelu) acc[j] = fmaxf(acc[j], 0.0f);
      else if (act == kActSigmoid) acc[j] = 1.0f / (1.0f + __expf(-acc[j]));
    }
    EY* dst = y + pix_base(s, p) + tap_off(s, n0);
    if (nn == 8) store8(dst, acc);
    else
      for (int j = 0; j < nn; ++j) dst[j] = (EY)acc[j];
  }
}

// ------------------------------------------------------------------------------------------------ data gradient
// dx[p, k] = scale * sum_n dy'[p, n] W[n, k], dy' = dy * act'(y); one thread per (pixel, group of 8 input channels)
template <typename EX, typename EY, typename EW, int K>
__global__ void __launch_bounds__(256)
pw_bwd_data_kernel(const EY* __restrict__ dy, const EY* __restrict__ y, const EW* __restrict__ w, EX* __restrict__ dx, PwShape s, int act,
                   const float* __restrict__ scale_ptr) {
  extern __shared__ float pw_lds[];
  load_weights(pw_lds, w, static_cast<const EW*>(nullptr), s);
  const float scale = scale_ptr ? scale_ptr[0] : 1.0f;
  constexpr int G = K / 8;
  const int items = s.P * G;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
    const int kg = it / s.P, p = it - kg * s.P;
    const int pb = pix_base(s, p);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    int o = 0, row = 0, kx = 0;  // (transposed) channel offset inside the tap, element offset of the tap's output pixel, tap column
    for (int n0 = 0; n0 < s.N; n0 += 8) {
      const int nn = s.N - n0 < 8 ? s.N - n0 : 8;
      const int at = pb + (s.transposed ? row + o : n0);
      if (s.transposed) {
        o += 8;
        if (o == s.O) {
          o = 0;
          ++kx;
          row += s.O;
          if (kx == 3) { kx = 0; row += (3 * s.W - 3) * s.O; }
        }
      }
      float g[8], yv[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (nn == 8) {
        load8(dy + at, g);
        if (act != kActNone) load8(y + at, yv);
      } else {
        for (int j = 0; j < 8; ++j) {
          g[j] = j < nn ? (float)dy[at + j] : 0.0f;
          yv[j] = (j < nn && act != kActNone) ? (float)y[at + j] : 0.0f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j >= nn) continue;
        const float gj = g[j] * act_grad(act, yv[j]);
        const float* wr = pw_lds + (n0 + j) * K + 8 * kg;
        const float4 wa = *reinterpret_cast<const float4*>(wr), wb = *reinterpret_cast<const float4*>(wr + 4);
        acc[0] += gj * wa.x; acc[1] += gj * wa.y; acc[2] += gj * wa.z; acc[3] += gj * wa.w;
        acc[4] += gj * wb.x; acc[5] += gj * wb.y; acc[6] += gj * wb.z; acc[7] += gj * wb.w;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= scale;
    store8(dx + p * K + 8 * kg, acc);
  }
}

// ------------------------------------------------------------------------------------------------ weight / bias gradient
// (a) N and K multiples of 4: a block stages tiles of kTP pixels (x [kTP][K], dy' [kTP][N]) in LDS; a thread owns 4 x 4 blocks of
//     dW (pairs e = t, t + 256, ...: n-block e / (K / 4), k-block e % (K / 4)): 2 float4 LDS reads per 16 FMAs; the bias sums ride
//     on the k-block-0 threads.  (b) tiny layers (the tail, 3 x 32): one lane per pixel keeps all N * K + N sums in registers,
//     waves reduce once at the end.  One partial [N * K | O] per block either way, summed by pw_reduce_kernel.
constexpr int kTP = 32;
constexpr int kWPairs = 3;      // 4 x 4 blocks per thread: ceil(72 * 8 / 256)
constexpr int kSmall = 3 * 32;  // N * K up to which (b) is used

template <typename EX, typename EY>
__global__ void __launch_bounds__(256)
pw_bwd_weight_kernel(const EX* __restrict__ x, const EY* __restrict__ dy, const EY* __restrict__ y, float* __restrict__ partial, PwShape s,
                     int act, int n_tiles) {
  extern __shared__ float pw_lds[];
  const int K = s.K, N = s.N;
  float* xs = pw_lds;          // [kTP][K]
  float* gs = xs + kTP * K;    // [kTP][N]
  const int n_out = N * K + s.O, KB = K / 4, pairs = (N / 4) * KB;
  float acc[kWPairs][16], bsum[kWPairs][4];
#pragma unroll
  for (int j = 0; j < kWPairs; ++j) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) bsum[j][q] = 0.0f;
  }
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int p0 = tile * kTP;
    __syncthreads();
    for (int e = threadIdx.x; e < kTP * K; e += blockDim.x) {
      const int t = e / K, k = e - t * K;
      xs[e] = (p0 + t) < s.P ? (float)x[(p0 + t) * K + k] : 0.0f;
    }
    const int G8 = N / 8;  // (N is a multiple of 8 here: 32 or 288)
    for (int e = threadIdx.x; e < kTP * G8; e += blockDim.x) {
      const int t = e / G8, n0 = 8 * (e - t * G8);
      float g[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (p0 + t < s.P) {
        const int at = pix_base(s, p0 + t) + tap_off(s, n0);
        load8(dy + at, g);
        if (act != kActNone) {
          float yv[8];
          load8(y + at, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] *= act_grad(act, yv[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) gs[t * N + n0 + j] = g[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kWPairs; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e >= pairs) break;
      const int nb = e / KB, kb = e - nb * KB;
#pragma unroll 4
      for (int t = 0; t < kTP; ++t) {
        const float4 g4 = *reinterpret_cast<const float4*>(gs + t * N + 4 * nb);
        const float4 x4 = *reinterpret_cast<const float4*>(xs + t * K + 4 * kb);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[j][4 * a + c] += gv[a] * xv[c];
          bsum[j][a] += gv[a];
        }
      }
    }
  }
  float* out = partial + blockIdx.x * n_out;
  // the bias sums of out channel o collect the n with n % O == o (transposed: nine taps): the k-block-0 threads add them in LDS
  __syncthreads();
  float* bl = pw_lds;  // [O]
  for (int o = threadIdx.x; o < s.O; o += blockDim.x) bl[o] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kWPairs; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e >= pairs) break;
    const int nb = e / KB, kb = e - nb * KB;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int c = 0; c < 4; ++c) out[(4 * nb + a) * K + 4 * kb + c] = acc[j][4 * a + c];
      if (kb == 0) atomicAdd(&bl[(4 * nb + a) % s.O], bsum[j][a]);
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < s.O; o += blockDim.x) out[N * K + o] = bl[o];
}

// (b): N * K <= kSmall, not transposed.  Lane = pixel; NK = N * K sums + N bias sums in registers.
template <typename EX, typename EY, int K>
__global__ void __launch_bounds__(256)
pw_bwd_weight_small_kernel(const EX* __restrict__ x, const EY* __restrict__ dy, const EY* __restrict__ y, float* __restrict__ partial,
                           PwShape s, int act) {
  constexpr int NMAX = kSmall / 32;  // 3
  __shared__ float red[4][NMAX * K + NMAX];
  float acc[NMAX][K], bs[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    bs[n] = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) acc[n][k] = 0.0f;
  }
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < s.P; p += gridDim.x * blockDim.x) {
    float xr[K], g[NMAX];
#pragma unroll
    for (int k8 = 0; k8 < K / 8; ++k8) {
      float v[8];
      load8(x + p * K + 8 * k8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[8 * k8 + j] = v[j];
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      g[n] = 0.0f;
      if (n < s.N) {
        g[n] = (float)dy[p * s.N + n];
        if (act != kActNone) g[n] *= act_grad(act, (float)y[p * s.N + n]);
      }
      bs[n] += g[n];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[n][k] += g[n] * xr[k];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float a = acc[n][k];
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
      if (lane == 0) red[wave][n * K + k] = a;
    }
    float a = bs[n];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) red[wave][NMAX * K + n] = a;
  }
  __syncthreads();
  float* out = partial + blockIdx.x * (s.N * K + s.O);
  for (int e = threadIdx.x; e < s.N * K + s.O; e += blockDim.x) {
    const int src = e < s.N * K ? e : NMAX * K + (e - s.N * K);
    out[e] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
  }
}

// ------------------------------------------------------------------------------------------------ transposed convolution on MFMA
// ConvTranspose2d(32, 32, 3, stride 3) is 9 GEMMs [pixels x 32] x [32 x 32] (one per tap); v_mfma_f32_32x32x16_{bf16,f16}, fp32
// accumulation.  Operand lane maps (cdna_hip_programming.md): lane (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; D: col = l & 31, rows (reg & 3) + 8 (reg >> 2) + 4h.
//   forward        D[pixel][o] = sum_i X[pixel][i] W[i][tap][o]: A = 16 contiguous bytes of the pixel's row; B = 8 scalars of the
//                  weight memory [i][tap][o] (stride 288 elements; consecutive lanes = consecutive o: coalesced)
//   data gradient  D[pixel][i] = sum_(tap, o) dY[out pixel(tap)][o] W[i][tap][o]: A = 16 contiguous bytes of the output pixel's
//                  row, B = 16 contiguous bytes of the weight memory -- no transposition anywhere
// A wave takes tiles of 32 input pixels; D's rows are pixels, its columns the 32 channels: every store is 64 contiguous bytes.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename E> struct Mma;
template <> struct Mma<__bf16> {
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<_Float16> {
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
__device__ __forceinline__ int mma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <typename E>
__global__ void __launch_bounds__(256)
convt_fwd_mfma_kernel(const E* __restrict__ x, const E* __restrict__ w, const E* __restrict__ b, E* __restrict__ y, PwShape s) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int n_tiles = (s.P + 31) / 32;
  const float bias = b ? (float)b[r] : 0.0f;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int p = tile * 32 + r;
    u32x4 a[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) a[c] = p < s.P ? *reinterpret_cast<const u32x4*>(x + p * 32 + 16 * c + 8 * h) : zero;
    int base[16];  // output offsets of this lane's 16 pixel rows
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pp = tile * 32 + mma_row(q, h);
      base[q] = pp < s.P ? pix_base(s, pp) : -1;
    }
    for (int tap = 0; tap < 9; ++tap) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = bias;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        typedef E e8 __attribute__((ext_vector_type(8)));
        e8 bv;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = w[(16 * c + 8 * h + j) * 288 + tap * 32 + r];
        acc = Mma<E>::mfma(a[c], __builtin_bit_cast(u32x4, bv), acc);
      }
      const int off = ((tap / 3) * 3 * s.W + tap % 3) * 32 + r;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (base[q] >= 0) y[base[q] + off] = (E)acc[q];
    }
  }
}

template <typename E>
__global__ void __launch_bounds__(256)
convt_bwd_data_mfma_kernel(const E* __restrict__ dy, const E* __restrict__ w, E* __restrict__ dx, PwShape s) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const int n_tiles = (s.P + 31) / 32;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int p = tile * 32 + r;
    const int pb = p < s.P ? pix_base(s, p) : -1;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    for (int tap = 0; tap < 9; ++tap) {
      const int off = ((tap / 3) * 3 * s.W + tap % 3) * 32;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const u32x4 av = pb >= 0 ? *reinterpret_cast<const u32x4*>(dy + pb + off + 16 * c + 8 * h) : zero;
        const u32x4 bv = *reinterpret_cast<const u32x4*>(w + r * 288 + tap * 32 + 16 * c + 8 * h);
        acc = Mma<E>::mfma(av, bv, acc);
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pp = tile * 32 + mma_row(q, h);
      if (pp < s.P) dx[pp * 32 + r] = (E)acc[q];
    }
  }
}

// grads (16-bit, the parameters' memory: [N][K] or the transposed convolution's [K][N]; bias [O]) (+)= sum of the partials
template <typename EW>
__global__ void __launch_bounds__(256)
pw_reduce_kernel(const float* __restrict__ partial, int n_blocks, EW* __restrict__ gw, EW* __restrict__ gb, PwShape s, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_out = s.N * s.K + s.O;
  if (e >= n_out) return;
  float a = 0.0f;
  int b = 0;
  for (; b + 8 <= n_blocks; b += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = partial[(int64_t)(b + j) * n_out + e];
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j];
  }
  for (; b < n_blocks; ++b) a += partial[(int64_t)b * n_out + e];
  EW* dst;
  if (e < s.N * s.K) {
    const int n = e / s.K, k = e - n * s.K;
    dst = gw + (s.transposed ? (int64_t)k * s.N + n : e);
  } else {
    if (gb == nullptr) return;
    dst = gb + (e - s.N * s.K);
  }
  *dst = (EW)(accumulate ? (float)*dst + a : a);
}

constexpr int kMaxBlocksW = 128;

bool shape_ok(const PwShape& s) {
  return s.P >= 0 && s.K > 0 && s.K <= kMaxK && s.K % 8 == 0 && s.N > 0 && s.N <= 9 * 64 && s.O > 0 &&
         (s.transposed ? (s.N == 9 * s.O && s.O % 8 == 0 && s.H > 0 && s.W > 0 && s.P % (s.H * s.W) == 0) : s.N == s.O) &&
         s.N * s.K + s.O <= 256 * 40;
}
// the shape the MFMA kernels are written for (NR_TUNE_PW_MFMA_OFF: the generic kernels)
bool convt_mfma(const PwShape& s, int x_f32, int y_f32, int act) {
  return !nr_tuning().pw_mfma_off && s.transposed && s.K == 32 && s.O == 32 && !x_f32 && !y_f32 && act == kActNone;
}
unsigned pw_blocks(int64_t items) {
  const int64_t b = nr_cdiv(items, 256);
  return (unsigned)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace

// dtype codes: NR_DTYPE_F32 (0), NR_DTYPE_BF16 (1), NR_DTYPE_F16 (2).  dtype16: the parameters' (and 16-bit activations') type.
#define PW_DISPATCH_K(CALL, EX, EY, EW)                     \
  switch (s.K) {                                            \
    case 32: CALL(EX, EY, EW, 32); break;                   \
    case 48: CALL(EX, EY, EW, 48); break;                   \
    default: return NR_EINVAL;                              \
  }
#define PW_DISPATCH(CALL)                                                                                       \
  if (dtype16 == NR_DTYPE_BF16) {                                                                               \
    if (x_f32 && !y_f32) { PW_DISPATCH_K(CALL, float, __bf16, __bf16) }                                         \
    else if (!x_f32 && !y_f32) { PW_DISPATCH_K(CALL, __bf16, __bf16, __bf16) }                                  \
    else if (!x_f32 && y_f32) { PW_DISPATCH_K(CALL, __bf16, float, __bf16) }                                    \
    else return NR_EINVAL;                                                                                      \
  } else if (dtype16 == NR_DTYPE_F16) {                                                                         \
    if (x_f32 && !y_f32) { PW_DISPATCH_K(CALL, float, _Float16, _Float16) }                                     \
    else if (!x_f32 && !y_f32) { PW_DISPATCH_K(CALL, _Float16, _Float16, _Float16) }                            \
    else if (!x_f32 && y_f32) { PW_DISPATCH_K(CALL, _Float16, float, _Float16) }                                \
    else return NR_EINVAL;                                                                                      \
  } else return NR_EINVAL;

inline bool pw_fits(int64_t n_pixels, int in_channels, int out_channels, int transposed) {
  const int64_t n = transposed ? 9 * (int64_t)out_channels : out_channels;
  return n_pixels >= 0 && n_pixels * (n > in_channels ? n : in_channels) < (int64_t)1 << 30;
}
static PwShape pw_shape(int64_t n_pixels, int in_channels, int out_channels, int transposed, int height, int width) {
  PwShape s;
  s.P = (int)n_pixels; s.K = in_channels; s.O = out_channels; s.transposed = transposed ? 1 : 0;
  s.N = transposed ? 9 * out_channels : out_channels; s.H = height; s.W = width;
  return s;
}

extern "C" int nr_pw_fwd(const void* x, int x_f32, const void* w16, const void* b16, void* y, int y_f32, int64_t n_pixels, int in_channels,
                         int out_channels, int act, int transposed, int height, int width, int dtype16, nr_stream_t stream) {
  if (n_pixels == 0) return 0;
  const PwShape s = pw_shape(n_pixels, in_channels, out_channels, transposed, height, width);
  if (!x || !w16 || !y || !pw_fits(n_pixels, in_channels, out_channels, transposed) || !shape_ok(s) || act < 0 || act > 2 || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w16) & 15u) != 0) return NR_EINVAL;
  const size_t lds = (size_t)(s.N * s.K + s.O) * sizeof(float);
  if (convt_mfma(s, x_f32, y_f32, act)) {  // the decoder's ConvTranspose2d(32, 32, 3, stride 3): 9 GEMMs on the matrix cores
    const unsigned blocks = (unsigned)nr_cdiv(nr_cdiv(s.P, 32), 4);
    if (dtype16 == NR_DTYPE_BF16)
      hipLaunchKernelGGL(convt_fwd_mfma_kernel<__bf16>, dim3(blocks), dim3(256), 0, nr_s(stream), static_cast<const __bf16*>(x),
                         static_cast<const __bf16*>(w16), static_cast<const __bf16*>(b16), static_cast<__bf16*>(y), s);
    else
      hipLaunchKernelGGL(convt_fwd_mfma_kernel<_Float16>, dim3(blocks), dim3(256), 0, nr_s(stream), static_cast<const _Float16*>(x),
                         static_cast<const _Float16*>(w16), static_cast<const _Float16*>(b16), static_cast<_Float16*>(y), s);
    NR_LAUNCH_CHECK();
    return 0;
  }
#define CALL(EX, EY, EW, KK)                                                                                                  \
  hipLaunchKernelGGL((pw_fwd_kernel<EX, EY, EW, KK>), dim3(pw_blocks((int64_t)s.P * ((s.N + 7) / 8))), dim3(256), lds, nr_s(stream), static_cast<const EX*>(x), \
                     static_cast<const EW*>(w16), static_cast<const EW*>(b16), static_cast<EY*>(y), s, act)
  PW_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_pw_bwd_data(const void* grad_y, const void* y, int y_f32, const void* w16, void* grad_x, int x_f32, int64_t n_pixels,
                              int in_channels, int out_channels, int act, int transposed, int height, int width, const float* scale,
                              int dtype16, nr_stream_t stream) {
  if (n_pixels == 0) return 0;
  const PwShape s = pw_shape(n_pixels, in_channels, out_channels, transposed, height, width);
  if (!grad_y || !w16 || !grad_x || (act != 0 && !y) || !pw_fits(n_pixels, in_channels, out_channels, transposed) || !shape_ok(s) || act < 0 || act > 2 ||
      (((uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)w16) & 15u) != 0)
    return NR_EINVAL;
  const size_t lds = (size_t)(s.N * s.K + s.O) * sizeof(float);
  if (convt_mfma(s, x_f32, y_f32, act) && scale == nullptr) {
    const unsigned blocks = (unsigned)nr_cdiv(nr_cdiv(s.P, 32), 4);
    if (dtype16 == NR_DTYPE_BF16)
      hipLaunchKernelGGL(convt_bwd_data_mfma_kernel<__bf16>, dim3(blocks), dim3(256), 0, nr_s(stream), static_cast<const __bf16*>(grad_y),
                         static_cast<const __bf16*>(w16), static_cast<__bf16*>(grad_x), s);
    else
      hipLaunchKernelGGL(convt_bwd_data_mfma_kernel<_Float16>, dim3(blocks), dim3(256), 0, nr_s(stream),
                         static_cast<const _Float16*>(grad_y), static_cast<const _Float16*>(w16), static_cast<_Float16*>(grad_x), s);
    NR_LAUNCH_CHECK();
    return 0;
  }
#define CALL(EX, EY, EW, KK)                                                                                                       \
  hipLaunchKernelGGL((pw_bwd_data_kernel<EX, EY, EW, KK>), dim3(pw_blocks((int64_t)s.P * (s.K / 8))), dim3(256), lds, nr_s(stream),                       \
                     static_cast<const EY*>(grad_y), static_cast<const EY*>(y), static_cast<const EW*>(w16), static_cast<EX*>(grad_x), s, \
                     act, scale)
  PW_DISPATCH(CALL)
#undef CALL
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t nr_pw_workspace_bytes(void) { return (int64_t)kMaxBlocksW * 256 * 40 * 4; }

extern "C" int nr_pw_bwd_weight(const void* x, int x_f32, const void* grad_y, const void* y, int y_f32, void* grad_w16, void* grad_b16,
                                int accumulate, void* workspace, int64_t n_pixels, int in_channels, int out_channels, int act,
                                int transposed, int height, int width, int dtype16, nr_stream_t stream) {
  const PwShape s = pw_shape(n_pixels, in_channels, out_channels, transposed, height, width);
  if (!x || !grad_y || !grad_w16 || !workspace || (act != 0 && !y) || !pw_fits(n_pixels, in_channels, out_channels, transposed) || !shape_ok(s) || act < 0 || act > 2 || n_pixels <= 0) return NR_EINVAL;
  const bool small = !s.transposed && s.N * s.K <= kSmall && s.K == 32;
  if (!small && ((s.N & 7) != 0 || (s.K & 3) != 0 || ((s.N / 4) * (s.K / 4) + 255) / 256 > kWPairs)) return NR_EINVAL;
  const int64_t tiles = nr_cdiv(s.P, kTP);
  const unsigned blocks = small ? (unsigned)(nr_cdiv(s.P, 256) < kMaxBlocksW ? nr_cdiv(s.P, 256) : kMaxBlocksW)
                                : (unsigned)(tiles < kMaxBlocksW ? tiles : kMaxBlocksW);
  const size_t lds = (size_t)(kTP * s.K + kTP * s.N) * sizeof(float);
  float* part = static_cast<float*>(workspace);
  const int n_out = s.N * s.K + s.O;
#define CALL_W(EX, EY)                                                                                                           \
  if (small)                                                                                                                     \
    hipLaunchKernelGGL((pw_bwd_weight_small_kernel<EX, EY, 32>), dim3(blocks), dim3(256), 0, nr_s(stream), static_cast<const EX*>(x), \
                       static_cast<const EY*>(grad_y), static_cast<const EY*>(y), part, s, act);                                 \
  else                                                                                                                           \
    hipLaunchKernelGGL((pw_bwd_weight_kernel<EX, EY>), dim3(blocks), dim3(256), lds, nr_s(stream), static_cast<const EX*>(x),     \
                       static_cast<const EY*>(grad_y), static_cast<const EY*>(y), part, s, act, (int)tiles)
#define CALL_R(EW)                                                                                                               \
  hipLaunchKernelGGL(pw_reduce_kernel<EW>, dim3((n_out + 255) / 256), dim3(256), 0, nr_s(stream), part, (int)blocks,             \
                     static_cast<EW*>(grad_w16), static_cast<EW*>(grad_b16), s, accumulate)
  if (dtype16 == NR_DTYPE_BF16) {
    if (x_f32 && !y_f32) { CALL_W(float, __bf16); }
    else if (!x_f32 && !y_f32) { CALL_W(__bf16, __bf16); }
    else if (!x_f32 && y_f32) { CALL_W(__bf16, float); }
    else return NR_EINVAL;
    CALL_R(__bf16);
  } else if (dtype16 == NR_DTYPE_F16) {
    if (x_f32 && !y_f32) { CALL_W(float, _Float16); }
    else if (!x_f32 && !y_f32) { CALL_W(_Float16, _Float16); }
    else if (!x_f32 && y_f32) { CALL_W(_Float16, float); }
    else return NR_EINVAL;
    CALL_R(_Float16);
  } else return NR_EINVAL;
#undef CALL_W
#undef CALL_R
  NR_LAUNCH_CHECK();
  return 0;
}
