// Training-mode batch normalisation of a channels-last activation [M pixels, C channels] with the residual add and the ReLU
// around it in the same launches -- the RGB decoder's BasicBlock (model_components/cnns.py:21-47: conv -> BN -> ReLU -> conv -> BN
// -> (+ x) -> ReLU; torch.nn.BatchNorm2d in train mode: biased variance for the normalisation, unbiased for running_var).
// torch runs this as three MIOpen kernels per batch norm each way plus the clamp, the add and their backwards (11 launches per
// normalisation, 8 normalisations per step: a third of the CNN chain's launches, each ~5 us of a launch-bound chain).  Here:
//   forward   bn_stats (per-block partial sums of x, x^2) -> bn_apply (mean / rstd from the partials, running statistics,
//             y = relu(gamma * xhat + beta + residual))
//   backward  bn_bwd_stats (g' = g masked by y > 0; partial sums of g', g' * xhat) -> bn_bwd_apply (d gamma, d beta,
//             dx = gamma * rstd * (g' - mean(g') - xhat * mean(g' xhat)), d residual = g')
// Activations in fp32, bf16 or fp16; parameters, statistics and all sums in fp32.  C = 8, 16, 32 or 64.
#include "nr_common.h"

namespace {

constexpr int kBnThreads = 256, kBnMaxBlocks = 128, kV = 8;  // a thread moves kV consecutive channels of a pixel (16 bytes in 16 bit)

template <typename T> struct BnIo;
template <> struct BnIo<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[kV]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[kV]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <typename E>
struct BnIo16 {
  typedef E vec __attribute__((ext_vector_type(kV)));
  static __device__ __forceinline__ void ld(const E* p, float (&v)[kV]) {
    const vec t = *reinterpret_cast<const vec*>(p);
#pragma unroll
    for (int k = 0; k < kV; ++k) v[k] = (float)t[k];
  }
  static __device__ __forceinline__ void st(E* p, const float (&v)[kV]) {
    vec t;
#pragma unroll
    for (int k = 0; k < kV; ++k) t[k] = (E)v[k];
    *reinterpret_cast<vec*>(p) = t;
  }
};
template <> struct BnIo<_Float16> : BnIo16<_Float16> {};
template <> struct BnIo<__bf16> : BnIo16<__bf16> {};

// Thread t of a block: channel vector t % CV (channels 8 (t % CV) ... + 7), row slot t / CV; R = 256 / CV rows per pass.
// partial [blocks][2][C]: this block's sums of a, b over its rows.
__device__ __forceinline__ void bn_block_partials(const float (&a)[kV], const float (&b)[kV], int C, float* __restrict__ partial) {
  __shared__ float red[2][kBnThreads * kV];  // [slot][channel]
  const int tid = threadIdx.x, CV = C / kV, cv = tid % CV, slot = tid / CV, R = kBnThreads / CV;
#pragma unroll
  for (int k = 0; k < kV; ++k) {
    red[0][slot * C + cv * kV + k] = a[k];
    red[1][slot * C + cv * kV + k] = b[k];
  }
  __syncthreads();
  for (int j = tid; j < 2 * C; j += kBnThreads) {
    const int which = j / C, c = j % C;
    float s = 0.0f;
    for (int g = 0; g < R; ++g) s += red[which][g * C + c];
    partial[((int64_t)blockIdx.x * 2 + which) * C + c] = s;
  }
}

// sums over all blocks' partials -> LDS tot[2][C] (every block does it: nblk * 2 * C floats out of the L2, all 256 threads
// loading: thread t sums the partials of (sum, channel) t % 2C over blocks t / 2C, t / 2C + 256 / 2C, ...)
__device__ __forceinline__ void bn_total(const float* __restrict__ partial, int nblk, int C, float (*tot)[64]) {
  __shared__ float part[kBnThreads];
  const int tid = threadIdx.x, G = kBnThreads / (2 * C) > 0 ? kBnThreads / (2 * C) : 1;
  if (tid < G * 2 * C) {
    const int k = tid % (2 * C);
    float s = 0.0f;
    for (int b = tid / (2 * C); b < nblk; b += G) s += partial[(int64_t)b * 2 * C + k];
    part[tid] = s;
  }
  __syncthreads();
  if (tid < 2 * C) {
    float t = 0.0f;
    for (int g = 0; g < G; ++g) t += part[g * 2 * C + tid];
    tot[tid / C][tid % C] = t;
  }
  __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_stats_kernel(const T* __restrict__ x, int64_t M, int C, float* __restrict__ partial) {
  const int tid = threadIdx.x, CV = C / kV, cv = tid % CV, R = kBnThreads / CV;
  float s[kV], ss[kV];
#pragma unroll
  for (int k = 0; k < kV; ++k) s[k] = ss[k] = 0.0f;
  for (int64_t r = (int64_t)blockIdx.x * R + tid / CV; r < M; r += (int64_t)gridDim.x * R) {
    float v[kV];
    BnIo<T>::ld(x + r * C + cv * kV, v);
#pragma unroll
    for (int k = 0; k < kV; ++k) {
      s[k] += v[k];
      ss[k] += v[k] * v[k];
    }
  }
  bn_block_partials(s, ss, C, partial);
}

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ residual, int64_t M, int C, const float* __restrict__ partial, int nblk,
                const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                float* __restrict__ running_mean, float* __restrict__ running_var, int relu, T* __restrict__ y,
                float* __restrict__ save_mean, float* __restrict__ save_rstd, uint32_t* __restrict__ generation) {
  __shared__ float tot[2][64];
  if (running_mean != nullptr && blockIdx.x == 0 && threadIdx.x == 0) nr_bump_generation(generation);  // (the running statistics move)
  __shared__ float scale[64], shift[64];
  bn_total(partial, nblk, C, tot);
  const int tid = threadIdx.x;
  if (tid < C) {
    const float mean = tot[0][tid] / (float)M;
    const float var = fmaxf(tot[1][tid] / (float)M - mean * mean, 0.0f);  // biased (the normalisation's)
    const float rstd = 1.0f / sqrtf(var + eps);
    scale[tid] = gamma[tid] * rstd;
    shift[tid] = beta[tid] - mean * gamma[tid] * rstd;
    if (blockIdx.x == 0) {
      save_mean[tid] = mean;
      save_rstd[tid] = rstd;
      if (running_mean != nullptr) {  // torch: running = (1 - momentum) * running + momentum * batch (unbiased variance)
        running_mean[tid] = (1.0f - momentum) * running_mean[tid] + momentum * mean;
        running_var[tid] = (1.0f - momentum) * running_var[tid] + momentum * var * ((float)M / fmaxf((float)M - 1.0f, 1.0f));
      }
    }
  }
  __syncthreads();
  const int CV = C / kV, cv = tid % CV, R = kBnThreads / CV;
  for (int64_t r = (int64_t)blockIdx.x * R + tid / CV; r < M; r += (int64_t)gridDim.x * R) {
    float v[kV], q[kV];
    BnIo<T>::ld(x + r * C + cv * kV, v);
    if (residual != nullptr) BnIo<T>::ld(residual + r * C + cv * kV, q);
#pragma unroll
    for (int k = 0; k < kV; ++k) {
      v[k] = v[k] * scale[cv * kV + k] + shift[cv * kV + k];
      if (residual != nullptr) v[k] += q[k];
      if (relu) v[k] = fmaxf(v[k], 0.0f);
    }
    BnIo<T>::st(y + r * C + cv * kV, v);
  }
}

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_stats_kernel(const T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ x, int64_t M, int C,
                    const float* __restrict__ save_mean, const float* __restrict__ save_rstd, int relu, float* __restrict__ partial) {
  const int tid = threadIdx.x, CV = C / kV, cv = tid % CV, R = kBnThreads / CV;
  float mean[kV], rstd[kV], s1[kV], s2[kV];
#pragma unroll
  for (int k = 0; k < kV; ++k) {
    mean[k] = save_mean[cv * kV + k];
    rstd[k] = save_rstd[cv * kV + k];
    s1[k] = s2[k] = 0.0f;
  }
  for (int64_t r = (int64_t)blockIdx.x * R + tid / CV; r < M; r += (int64_t)gridDim.x * R) {
    float gv[kV], yv[kV], xv[kV];
    BnIo<T>::ld(g + r * C + cv * kV, gv);
    BnIo<T>::ld(x + r * C + cv * kV, xv);
    if (relu) BnIo<T>::ld(y + r * C + cv * kV, yv);
#pragma unroll
    for (int k = 0; k < kV; ++k) {
      if (relu && !(yv[k] > 0.0f)) gv[k] = 0.0f;
      s1[k] += gv[k];
      s2[k] += gv[k] * ((xv[k] - mean[k]) * rstd[k]);
    }
  }
  bn_block_partials(s1, s2, C, partial);
}

template <typename T>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y, const T* __restrict__ x, int64_t M, int C,
                    const float* __restrict__ partial, int nblk, const float* __restrict__ gamma, const float* __restrict__ save_mean,
                    const float* __restrict__ save_rstd, int relu, T* __restrict__ dx, T* __restrict__ d_residual,
                    float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  __shared__ float tot[2][64];
  bn_total(partial, nblk, C, tot);
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid < C) {  // "+=": the parameters' gradient buffers
    g_beta[tid] += tot[0][tid];
    g_gamma[tid] += tot[1][tid];
  }
  const int CV = C / kV, cv = tid % CV, R = kBnThreads / CV;
  float mean[kV], rstd[kV], kk[kV], m1[kV], m2[kV];
#pragma unroll
  for (int k = 0; k < kV; ++k) {
    const int c = cv * kV + k;
    mean[k] = save_mean[c];
    rstd[k] = save_rstd[c];
    kk[k] = gamma[c] * rstd[k];
    m1[k] = tot[0][c] / (float)M;
    m2[k] = tot[1][c] / (float)M;
  }
  for (int64_t r = (int64_t)blockIdx.x * R + tid / CV; r < M; r += (int64_t)gridDim.x * R) {
    float gv[kV], yv[kV], xv[kV], dv[kV];
    BnIo<T>::ld(g + r * C + cv * kV, gv);
    BnIo<T>::ld(x + r * C + cv * kV, xv);
    if (relu) BnIo<T>::ld(y + r * C + cv * kV, yv);
#pragma unroll
    for (int k = 0; k < kV; ++k) {
      if (relu && !(yv[k] > 0.0f)) gv[k] = 0.0f;
      dv[k] = kk[k] * (gv[k] - m1[k] - ((xv[k] - mean[k]) * rstd[k]) * m2[k]);
    }
    BnIo<T>::st(dx + r * C + cv * kV, dv);
    if (d_residual != nullptr) BnIo<T>::st(d_residual + r * C + cv * kV, gv);
  }
}

int bn_blocks(int64_t M, int C) {
  const int64_t R = kBnThreads / (C / kV), want = nr_cdiv(M, R * 2);  // ~2 pixels per thread
  return (int)(want < 1 ? 1 : (want < kBnMaxBlocks ? want : kBnMaxBlocks));
}
bool bn_ok(int64_t M, int C, int dtype) {
  return M >= 1 && C >= kV && C <= 64 && C % kV == 0 && kBnThreads % (2 * C) == 0 && (dtype == NR_DTYPE_F32 || dtype == NR_DTYPE_BF16 || dtype == NR_DTYPE_F16);
}

}  // namespace

extern "C" int64_t nr_bn_act_workspace_floats(int64_t M, int C) { return (C < 1 || C > 64) ? -1 : (int64_t)kBnMaxBlocks * 2 * C; }

extern "C" int nr_bn_act_fwd(const void* x, const void* residual, int64_t M, int C, int dtype, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, int relu, void* y,
                             float* save_mean, float* save_rstd, float* workspace, nr_stream_t stream) {
  if (M == 0) return 0;
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !workspace || !bn_ok(M, C, dtype) ||
      (running_mean == nullptr) != (running_var == nullptr))
    return NR_EINVAL;
  const int nblk = bn_blocks(M, C);
#define NR_BN_FWD(T)                                                                                                                  \
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(nblk), dim3(kBnThreads), 0, nr_s(stream), (const T*)x, M, C, workspace);                 \
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(nblk), dim3(kBnThreads), 0, nr_s(stream), (const T*)x, (const T*)residual, M, C,         \
                     workspace, nblk, gamma, beta, eps, momentum, running_mean, running_var, relu, (T*)y, save_mean, save_rstd, \
                     nr_generation_ptr())
  if (dtype == NR_DTYPE_F32) { NR_BN_FWD(float); } else if (dtype == NR_DTYPE_BF16) { NR_BN_FWD(__bf16); } else { NR_BN_FWD(_Float16); }
#undef NR_BN_FWD
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_bn_act_bwd(const void* grad_y, const void* y, const void* x, int64_t M, int C, int dtype, const float* gamma,
                             const float* save_mean, const float* save_rstd, int relu, void* grad_x, void* grad_residual,
                             float* grad_gamma, float* grad_beta, float* workspace, nr_stream_t stream) {
  if (M == 0) return 0;
  if (!grad_y || !y || !x || !gamma || !save_mean || !save_rstd || !grad_x || !grad_gamma || !grad_beta || !workspace ||
      !bn_ok(M, C, dtype))
    return NR_EINVAL;
  const int nblk = bn_blocks(M, C);
#define NR_BN_BWD(T)                                                                                                                  \
  hipLaunchKernelGGL(bn_bwd_stats_kernel<T>, dim3(nblk), dim3(kBnThreads), 0, nr_s(stream), (const T*)grad_y, (const T*)y,             \
                     (const T*)x, M, C, save_mean, save_rstd, relu, workspace);                                                       \
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(nblk), dim3(kBnThreads), 0, nr_s(stream), (const T*)grad_y, (const T*)y,             \
                     (const T*)x, M, C, workspace, nblk, gamma, save_mean, save_rstd, relu, (T*)grad_x, (T*)grad_residual, grad_gamma, \
                     grad_beta)
  if (dtype == NR_DTYPE_F32) { NR_BN_BWD(float); } else if (dtype == NR_DTYPE_BF16) { NR_BN_BWD(__bf16); } else { NR_BN_BWD(_Float16); }
#undef NR_BN_BWD
  NR_LAUNCH_CHECK();
  return 0;
}
