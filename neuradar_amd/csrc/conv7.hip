// 7 x 7 convolution, 32 -> 32 channels, stride 1, padding 3, on channels-last 16-bit activations: the convolutions of the RGB
// decoder's BasicBlocks (model_components/cnns.py:21-47, models/neuradar.py:225-240: eight of the CNN's eleven convolutions and
// > 95 % of its FLOPs) as an implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulation.
//
//   D[out channel o][pixel] = sum over (tap t = (ky, kx), in channel i)  W[o][t][i] * X[pixel + t - 3][i]
//   A = W (rows = out channels), B = X (columns = pixels): per (tap, half of the in channels) ONE MFMA per 32 pixels,
//   98 per tile row; lane (r, h) holds A[row r][k = 8h + j] / B[k = 8h + j][col r] (cdna_hip_programming.md, operand lane
//   maps), i.e. for the weights W[o = r][t][16 c + 8h .. + 7] and for the activations X[pixel r][16 c + 8h .. + 7]: both are 16
//   contiguous bytes of the channels-last tensors.
//   * the weights sit in LDS as a 98-KB image in exactly that fragment order (nr_conv7_pack writes it once per optimizer
//     step, forward orientation and the flipped / transposed one for the data gradient): one conflict-free ds_read_b128 per MFMA;
//   * a block of 8 waves takes a tile of 8 rows x 32 pixels: the 14 x 38-pixel input halo is staged in LDS with a pixel stride
//     of 80 bytes (64 of channels + 16 of padding: the b128 reads of 16 neighbouring pixels then cover all 64 banks), the next
//     tile's halo is requested into registers before the current tile's MFMAs and written to LDS after them;
//   * persistent blocks (one per CU: 98 KB + 42 KB of LDS) walk the tiles; the epilogue adds the bias, rounds to 16 bits and
//     stores 4 x 8 bytes per lane (D: column = pixel on the lane, rows = channels (reg & 3) + 8 (reg >> 2) + 4 h).
// The same kernel computes the DATA GRADIENT: dX = conv7(dY, W') with W'[i][t][o] = W[o][48 - t][i] (the other image).
// Rendering a 1920 x 1080 image is where it pays most: 0.83 TFLOP of 7 x 7 convolutions that MIOpen runs at ~6 % of the
// bf16 MFMA rate (tools/probe_render_entry.py).
#include "nr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct CBf16 {
  using elem = __bf16;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct CFp16 {
  using elem = _Float16;
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

constexpr int kC = 32;                    // channels in and out
constexpr int kTaps = 49;
constexpr int kFrag = 1024;               // bytes of one MFMA A operand for all 64 lanes
constexpr int kImg = kTaps * 2 * kFrag;   // 100 352 bytes: [tap][channel half][lane][8 x 16 bit]
constexpr int kImgAll = kImg + kC * 4;    // ... followed by the 32 biases in fp32: one orientation of one convolution
constexpr int kTH = 8, kTW = 32;          // output tile: 8 rows (one per wave) x 32 pixels
constexpr int kHH = kTH + 6, kHW = kTW + 6;
constexpr int kPix = 80;                  // bytes per staged pixel: 64 of channels + 16 of padding
constexpr int kHalo = kHH * kHW * kPix;   // 42 560 bytes
constexpr int kThreads = 64 * kTH;
constexpr int kChunks = kHH * kHW * 4;    // 16-byte pieces of a halo
constexpr int kPerThread = (kChunks + kThreads - 1) / kThreads;

// image[orientation]: forward  A[o][t][i]          = W[o][t][i]
//                     backward A[i][t][o] (of dX)  = W[o][48 - t][i]
// w: [32][49][32] 16-bit, contiguous (the channels-last memory of a [O, I, 7, 7] parameter)
template <typename E>
__global__ void __launch_bounds__(256)
conv7_pack_kernel(const E* __restrict__ wbase, nr_conv7_list_t list, unsigned char* __restrict__ images) {
  const int conv = blockIdx.y, orient = blockIdx.z;
  const E* w = wbase + list.offset[conv];
  unsigned char* dst = images + ((int64_t)conv * 2 + orient) * kImgAll;
  E* img = reinterpret_cast<E*>(dst);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kImg / 2; e += gridDim.x * blockDim.x) {
    const int j = e & 7, lane = (e >> 3) & 63, frag = e >> 9, c = frag & 1, t = frag >> 1;
    const int r = lane & 31, h = lane >> 5, k = 16 * c + 8 * h + j;
    img[e] = orient == 0 ? w[(r * kTaps + t) * kC + k] : w[(k * kTaps + (kTaps - 1 - t)) * kC + r];
  }
  if (blockIdx.x == 0 && threadIdx.x < kC) {  // the convolution's bias (fp32 in the image); none for the data gradient
    const bool has = orient == 0 && list.bias_offset[conv] >= 0;
    reinterpret_cast<float*>(dst + kImg)[threadIdx.x] = has ? (float)wbase[list.bias_offset[conv] + threadIdx.x] : 0.0f;
  }
}

// Rendering: BN(conv(x)) of a BasicBlock in eval mode as ONE convolution -- W'[o] = W[o] gamma[o] / sqrt(var[o] + eps),
// b' = (b - mean) gamma / sqrt(var + eps) + beta (model_components/cnns.py:21-47, torch BatchNorm2d eval) -- folded from the fp32
// master parameters and written as the forward weight image in one launch, with an EARLY OUT on the device: state[0] holds the
// parameter generation (+ 1) the images were built from; unless `force`, nothing is done while it still matches the device's
// generation word (nr_common.h) -- no host read, no version counter that raw-pointer optimizers bypass.  state[1] counts the
// rebuilds (tests).  A second one-thread launch commits state[0] after every block of the first has read it.
template <typename E>
__global__ void __launch_bounds__(256)
conv7_fold_pack_kernel(nr_conv7_fold_t list, unsigned char* __restrict__ images, const uint32_t* __restrict__ generation,
                       const uint32_t* __restrict__ state, int force) {
  if (!force && generation != nullptr && state[0] == *generation + 1u) return;
  const int conv = blockIdx.y;
  const float* w = list.weight[conv];
  const int64_t so = list.stride_o[conv], st = list.stride_t[conv], si = list.stride_i[conv];
  unsigned char* dst = images + (int64_t)conv * 2 * kImgAll;  // (orientation 0 of the [n, 2, bytes] image array)
  E* img = reinterpret_cast<E*>(dst);
  __shared__ float kf[kC];
  if (threadIdx.x < kC)
    kf[threadIdx.x] = list.gamma[conv][threadIdx.x] / sqrtf(list.var[conv][threadIdx.x] + list.eps[conv]);
  __syncthreads();
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kImg / 2; e += gridDim.x * blockDim.x) {
    const int j = e & 7, lane = (e >> 3) & 63, frag = e >> 9, c = frag & 1, t = frag >> 1;
    const int r = lane & 31, h = lane >> 5, k = 16 * c + 8 * h + j;
    img[e] = (E)(w[r * so + t * st + k * si] * kf[r]);
  }
  if (blockIdx.x == 0 && threadIdx.x < kC) {
    const int o = threadIdx.x;
    const float b = list.bias[conv] != nullptr ? list.bias[conv][o] : 0.0f;
    // (the bias passes through the 16-bit type like the weights do: what the torch-side fold this replaces handed to nr_conv7_pack)
    reinterpret_cast<float*>(dst + kImg)[o] = (float)(E)((b - list.mean[conv][o]) * kf[o] + list.beta[conv][o]);
  }
}

__global__ void conv7_fold_commit_kernel(const uint32_t* __restrict__ generation, uint32_t* __restrict__ state, int force) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t now = generation != nullptr ? *generation + 1u : 0u;
  if (force || generation == nullptr || state[0] != now) {
    state[0] = now;
    state[1] += 1u;
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
conv7_kernel(const typename T::elem* __restrict__ x, const unsigned char* __restrict__ image,
             const typename T::elem* __restrict__ residual, int relu, typename T::elem* __restrict__ y, int P, int H, int W,
             int tiles_y, int tiles_x, int64_t n_tiles) {
  using E = typename T::elem;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* wl = lds;          // weight image
  unsigned char* xt = lds + kImgAll;  // input halo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < kImgAll / 16; i += kThreads) reinterpret_cast<uint4*>(wl)[i] = reinterpret_cast<const uint4*>(image)[i];
  float bv[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) bv[q] = reinterpret_cast<const float*>(image + kImg)[(q & 3) + 8 * (q >> 2) + 4 * h];

  uint4 pre[kPerThread];
  auto request = [&](int64_t tile) {  // the halo of `tile` into registers (zeros outside the image = the padding)
    const int p = (int)(tile / ((int64_t)tiles_y * tiles_x)), rem = (int)(tile % ((int64_t)tiles_y * tiles_x));
    const int y0 = (rem / tiles_x) * kTH - 3, x0 = (rem % tiles_x) * kTW - 3;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int c = tid + k * kThreads;
      const int pix = c >> 2, q = c & 3, ry = pix / kHW, rx = pix - ry * kHW, gy = y0 + ry, gx = x0 + rx;
      const bool in = c < kChunks && tile < n_tiles && gy >= 0 && gy < H && gx >= 0 && gx < W;
      pre[k] = in ? *reinterpret_cast<const uint4*>(x + (((int64_t)p * H + gy) * W + gx) * kC + q * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
      const int c = tid + k * kThreads;
      if (c < kChunks) *reinterpret_cast<uint4*>(xt + (c >> 2) * kPix + (c & 3) * 16) = pre[k];
    }
  };
  int64_t tile = blockIdx.x;
  request(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();  // every wave is done with the previous halo (and, the first time, the weight image is complete)
    commit();
    __syncthreads();
    request(tile + gridDim.x);  // in flight during this tile's MFMAs
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = bv[q];
    const unsigned char* xrow = xt + ((wave * kHW + r) * kPix) + h * 16;
#pragma unroll 7
    for (int t = 0; t < kTaps; ++t) {
      const int ky = t / 7, kx = t - ky * 7;
      const unsigned char* xp = xrow + (ky * kHW + kx) * kPix;
      const unsigned char* wp = wl + (t * 2) * kFrag + lane * 16;
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(wp), b0 = *reinterpret_cast<const u32x4*>(xp);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(wp + kFrag), b1 = *reinterpret_cast<const u32x4*>(xp + 32);
      acc = T::mfma(a0, b0, acc);
      acc = T::mfma(a1, b1, acc);
    }
    const int p = (int)(tile / ((int64_t)tiles_y * tiles_x)), rem = (int)(tile % ((int64_t)tiles_y * tiles_x));
    const int gy = (rem / tiles_x) * kTH + wave, gx = (rem % tiles_x) * kTW + r;
    if (gy < H && gx < W) {
      const int64_t at = (((int64_t)p * H + gy) * W + gx) * kC + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if (residual != nullptr) {  // (+ x) of the block's second convolution
          const uint2 rr = *reinterpret_cast<const uint2*>(residual + at + 8 * g);
          E re[4];
          __builtin_memcpy(re, &rr, 8);
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] += (float)re[k];
        }
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.0f);
        }
        *reinterpret_cast<uint2*>(y + at + 8 * g) = make_uint2(T::pack2(v[0], v[1]), T::pack2(v[2], v[3]));
      }
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
//   dW[o][t][i] = sum over pixels  dY[pixel][o] * X[pixel + t - 3][i],   db[o] = sum over pixels dY[pixel][o]
// Per tap a 32 x 32 product that contracts over PIXELS: A[row o][k = pixel], B[k = pixel][col i] -- both operands are columns of
// the channels-last tiles, read from LDS with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered
// column-major: the lane's channel for 4 consecutive pixels; two reads per operand).  Pixel rows of 64 bytes: the 4 pixels of a
// read are 256 contiguous bytes = all 64 banks once, whatever the tap's shift.  A block stages a tile of dY (8 x 32 pixels) and
// the 14 x 38 halo of X; its 8 waves split the 49 taps (wave w: taps w, w + 8, ...: 7 accumulators of 16 registers), the tile's
// 16 A fragments stay in registers for all of a wave's taps.  Persistent blocks keep their accumulators across tiles and leave
// ONE partial [49][32][32] (+ bias row) each; conv7_wgrad_reduce_kernel sums the partials.
constexpr int kWPix = 64;                       // bytes per staged pixel (no padding: transposed reads)
constexpr int kWHalo = kHH * kHW * kWPix;       // 34 048
constexpr int kWTile = kTH * kTW * kWPix;       // 16 384
constexpr int kWPart = kTaps * kC * kC + kC;    // floats of one block's partial: dW [t][o][i], then db [o]
constexpr int kWMaxBlocks = 64;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 8 consecutive pixels (8h + j) of the lane's channel (lane & 31), starting at pixel `pix0` of an image with 64-byte pixel rows
__device__ __forceinline__ u32x4 tr_pixels(const unsigned char* img, int pix0, int lane) {
  const int h = lane >> 5, q = (lane & 15) >> 2, p = lane & 3, u = 4 * ((lane >> 4) & 1) + p;
  u32x4 f;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const unsigned char* a = img + (pix0 + 8 * h + 4 * jj + q) * kWPix + 8 * u;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
    const u32x2 w = __builtin_bit_cast(u32x2, v);
    f[2 * jj] = w[0];
    f[2 * jj + 1] = w[1];
  }
  return f;
}

template <typename T>
__global__ void __launch_bounds__(kThreads)
conv7_wgrad_kernel(const typename T::elem* __restrict__ x, const typename T::elem* __restrict__ gy, float* __restrict__ partial,
                   int P, int H, int W, int tiles_y, int tiles_x, int64_t n_tiles) {
  using E = typename T::elem;
  __shared__ __attribute__((aligned(16))) unsigned char xs[kWHalo];
  __shared__ __attribute__((aligned(16))) unsigned char gs[kWTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  constexpr int kMine = (kTaps + kTH - 1) / kTH;  // taps per wave (7)
  f32x16 acc[kMine];
#pragma unroll
  for (int m = 0; m < kMine; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;
  float bsum = 0.0f;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int p = (int)(tile / ((int64_t)tiles_y * tiles_x)), rem = (int)(tile % ((int64_t)tiles_y * tiles_x));
    const int ty0 = (rem / tiles_x) * kTH, tx0 = (rem % tiles_x) * kTW;
    __syncthreads();  // the previous tile's reads are done
    for (int c = tid; c < kChunks; c += kThreads) {  // X halo: 14 x 38 pixels x 4 pieces of 16 bytes
      const int pix = c >> 2, q = c & 3, ry = pix / kHW, rx = pix - ry * kHW, gy_ = ty0 - 3 + ry, gx_ = tx0 - 3 + rx;
      const bool in = gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W;
      *reinterpret_cast<uint4*>(xs + pix * kWPix + q * 16) =
          in ? *reinterpret_cast<const uint4*>(x + (((int64_t)p * H + gy_) * W + gx_) * kC + q * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    for (int c = tid; c < kTH * kTW * 4; c += kThreads) {  // dY tile
      const int pix = c >> 2, q = c & 3, ry = pix / kTW, rx = pix - ry * kTW, gy_ = ty0 + ry, gx_ = tx0 + rx;
      const bool in = gy_ < H && gx_ < W;
      *reinterpret_cast<uint4*>(gs + pix * kWPix + q * 16) =
          in ? *reinterpret_cast<const uint4*>(gy + (((int64_t)p * H + gy_) * W + gx_) * kC + q * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#pragma unroll 2
    for (int kb = 0; kb < 16; ++kb) {  // k-block = (tile row, half of the row): 16 pixels
      const u32x4 a = tr_pixels(gs, (kb >> 1) * kTW + (kb & 1) * 16, lane);  // dY, shared by the wave's taps
      if (wave == 0) {  // db: every lane sums its channel over its 8 pixels of the k-block
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t word = a[q];
          E e2[2];
          __builtin_memcpy(e2, &word, 4);
          bsum += (float)e2[0] + (float)e2[1];
        }
      }
#pragma unroll
      for (int m = 0; m < kMine; ++m) {
        const int t = wave + m * kTH;
        if (t < kTaps) {  // (uniform per wave)
          const int ky = t / 7, kx = t - ky * 7;
          const u32x4 b = tr_pixels(xs, ((kb >> 1) + ky) * kHW + (kb & 1) * 16 + kx, lane);
          acc[m] = T::mfma(a, b, acc[m]);
        }
      }
    }
  }
  // D[row o][col i]: column on the lane, rows (reg & 3) + 8 (reg >> 2) + 4 h
  float* out = partial + (int64_t)blockIdx.x * kWPart;
  const int i = lane & 31;
#pragma unroll
  for (int m = 0; m < kMine; ++m) {
    const int t = wave + m * kTH;
    if (t < kTaps) {
#pragma unroll
      for (int q = 0; q < 16; ++q) out[(t * kC + (q & 3) + 8 * (q >> 2) + 4 * h) * kC + i] = acc[m][q];
    }
  }
  if (wave == 0) {
    bsum += nr_xor32_f(bsum);  // the two pixel halves of the lane's channel
    if (lane < kC) out[kTaps * kC * kC + lane] = bsum;
  }
}

// dW [o][t][i] (the parameter's channels-last memory) and db [o] = sum of the blocks' partials [t][o][i] | [o]
template <typename E>
__global__ void __launch_bounds__(256)
conv7_wgrad_reduce_kernel(const float* __restrict__ partial, int n_blocks, E* __restrict__ gw, E* __restrict__ gb, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kWPart) return;
  float s = 0.0f;
  int b = 0;
  for (; b + 8 <= n_blocks; b += 8) {  // eight independent loads in flight (a dependent chain of 64 loads was 13.6 us)
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = partial[(int64_t)(b + k) * kWPart + e];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; b < n_blocks; ++b) s += partial[(int64_t)b * kWPart + e];
  if (e < kTaps * kC * kC) {
    const int i = e % kC, o = (e / kC) % kC, t = e / (kC * kC);
    E* dst = gw + (o * kTaps + t) * kC + i;
    *dst = (E)(accumulate ? (float)*dst + s : s);
  } else if (gb != nullptr) {
    E* dst = gb + (e - kTaps * kC * kC);
    *dst = (E)(accumulate ? (float)*dst + s : s);
  }
}

}  // namespace

int nr_init_conv7() {
  if (int rc = nr_raise_lds(conv7_kernel<CBf16>, (size_t)kImgAll + kHalo)) return rc;
  return nr_raise_lds(conv7_kernel<CFp16>, (size_t)kImgAll + kHalo);
}

extern "C" int64_t nr_conv7_wgrad_workspace_bytes(void) { return (int64_t)kWMaxBlocks * kWPart * 4; }

extern "C" int nr_conv7_wgrad(const void* x16, const void* grad_y16, void* grad_w16, void* grad_b16, int accumulate, void* workspace,
                              int n_images, int height, int width, int dtype, nr_stream_t stream) {
  if (!x16 || !grad_y16 || !grad_w16 || !workspace || n_images < 0 || height < 0 || width < 0 ||
      (dtype != NR_DTYPE_BF16 && dtype != NR_DTYPE_F16) || (((uintptr_t)x16 | (uintptr_t)grad_y16 | (uintptr_t)workspace) & 15u) != 0)
    return NR_EINVAL;
  const int ty = (height + kTH - 1) / kTH, tx = (width + kTW - 1) / kTW;
  const int64_t tiles = (int64_t)n_images * ty * tx;
  const unsigned blocks = (unsigned)(tiles < kWMaxBlocks ? (tiles > 0 ? tiles : 1) : kWMaxBlocks);
  float* part = static_cast<float*>(workspace);
  const unsigned rblocks = (kWPart + 255) / 256;
  if (dtype == NR_DTYPE_BF16) {
    hipLaunchKernelGGL(conv7_wgrad_kernel<CBf16>, dim3(blocks), dim3(kThreads), 0, nr_s(stream), static_cast<const __bf16*>(x16),
                       static_cast<const __bf16*>(grad_y16), part, n_images, height, width, ty, tx, tiles);
    hipLaunchKernelGGL(conv7_wgrad_reduce_kernel<__bf16>, dim3(rblocks), dim3(256), 0, nr_s(stream), part, (int)blocks,
                       static_cast<__bf16*>(grad_w16), static_cast<__bf16*>(grad_b16), accumulate);
  } else {
    hipLaunchKernelGGL(conv7_wgrad_kernel<CFp16>, dim3(blocks), dim3(kThreads), 0, nr_s(stream), static_cast<const _Float16*>(x16),
                       static_cast<const _Float16*>(grad_y16), part, n_images, height, width, ty, tx, tiles);
    hipLaunchKernelGGL(conv7_wgrad_reduce_kernel<_Float16>, dim3(rblocks), dim3(256), 0, nr_s(stream), part, (int)blocks,
                       static_cast<_Float16*>(grad_w16), static_cast<_Float16*>(grad_b16), accumulate);
  }
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t nr_conv7_image_bytes(void) { return 2 * (int64_t)kImgAll; }

extern "C" int nr_conv7_pack(const void* weights16, const nr_conv7_list_t* list, int dtype, void* images, nr_stream_t stream) {
  if (!list || list->n == 0) return 0;
  if (!weights16 || !images || list->n < 0 || list->n > NR_CONV7_MAX || (dtype != NR_DTYPE_BF16 && dtype != NR_DTYPE_F16) ||
      ((uintptr_t)images & 15u) != 0)
    return NR_EINVAL;
  const dim3 grid(8, (unsigned)list->n, 2);
  if (dtype == NR_DTYPE_BF16)
    hipLaunchKernelGGL(conv7_pack_kernel<__bf16>, grid, dim3(256), 0, nr_s(stream), static_cast<const __bf16*>(weights16), *list,
                       static_cast<unsigned char*>(images));
  else
    hipLaunchKernelGGL(conv7_pack_kernel<_Float16>, grid, dim3(256), 0, nr_s(stream), static_cast<const _Float16*>(weights16), *list,
                       static_cast<unsigned char*>(images));
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_conv7_fold_pack(const nr_conv7_fold_t* list, int dtype, void* images, uint32_t* state, int force, nr_stream_t stream) {
  if (!list || list->n == 0) return 0;
  if (!images || !state || list->n < 0 || list->n > NR_CONV7_MAX || (dtype != NR_DTYPE_BF16 && dtype != NR_DTYPE_F16) ||
      ((uintptr_t)images & 15u) != 0)
    return NR_EINVAL;
  for (int k = 0; k < list->n; ++k)
    if (!list->weight[k] || !list->gamma[k] || !list->beta[k] || !list->mean[k] || !list->var[k]) return NR_EINVAL;
  const uint32_t* gen = nr_generation_ptr();
  const dim3 grid(8, (unsigned)list->n);
  if (dtype == NR_DTYPE_BF16)
    hipLaunchKernelGGL(conv7_fold_pack_kernel<__bf16>, grid, dim3(256), 0, nr_s(stream), *list, static_cast<unsigned char*>(images), gen, state,
                       force);
  else
    hipLaunchKernelGGL(conv7_fold_pack_kernel<_Float16>, grid, dim3(256), 0, nr_s(stream), *list, static_cast<unsigned char*>(images), gen,
                       state, force);
  hipLaunchKernelGGL(conv7_fold_commit_kernel, dim3(1), dim3(64), 0, nr_s(stream), gen, state, force);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_conv7_fwd(const void* x16, const void* image, const void* residual16, int relu, void* y16, int n_images, int height,
                            int width, int dtype, nr_stream_t stream) {
  if (n_images == 0 || height == 0 || width == 0) return 0;
  if (!x16 || !image || !y16 || n_images < 0 || height < 0 || width < 0 || (dtype != NR_DTYPE_BF16 && dtype != NR_DTYPE_F16) ||
      (((uintptr_t)x16 | (uintptr_t)y16 | (uintptr_t)image | (uintptr_t)residual16) & 15u) != 0)
    return NR_EINVAL;
  const int ty = (height + kTH - 1) / kTH, tx = (width + kTW - 1) / kTW;
  const int64_t tiles = (int64_t)n_images * ty * tx;
  const int cap = nr_tuning().conv7_blocks > 0 ? nr_tuning().conv7_blocks : 256;  // one block per CU (141 KB of LDS)
  const unsigned blocks = (unsigned)(tiles < cap ? tiles : cap);
  const size_t lds = (size_t)kImgAll + kHalo;  // (> 64 KB: the attribute is raised in nr_init)
  if (dtype == NR_DTYPE_BF16)
    hipLaunchKernelGGL(conv7_kernel<CBf16>, dim3(blocks), dim3(kThreads), lds, nr_s(stream), static_cast<const __bf16*>(x16),
                       static_cast<const unsigned char*>(image), static_cast<const __bf16*>(residual16), relu, static_cast<__bf16*>(y16),
                       n_images, height, width, ty, tx, tiles);
  else
    hipLaunchKernelGGL(conv7_kernel<CFp16>, dim3(blocks), dim3(kThreads), lds, nr_s(stream), static_cast<const _Float16*>(x16),
                       static_cast<const unsigned char*>(image), static_cast<const _Float16*>(residual16), relu,
                       static_cast<_Float16*>(y16), n_images, height, width, ty, tx, tiles);
  NR_LAUNCH_CHECK();
  return 0;
}
