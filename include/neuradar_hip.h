/*
 * neuradar_hip.h -- C ABI of libneuradar_hip.so: the MI355X (gfx950) volumetric-rendering hot path
 * of NeuRadar.  This is the drop-in boundary: each entry point stands in for a
 * call the reference makes into tiny-cuda-nn / nerfacc / torch on this path, and is what a binding
 * behind the reference's `implementation=` switch would call (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (torch); the library allocates nothing
 *     persistent and keeps no global state beyond the nr_set_tuning() table (set once at load).  All tensors are contiguous fp32 unless stated.
 *   - `stream` is a hipStream_t passed as void* (torch's current stream); calls are asynchronous.
 *   - Return value: 0 on success, a hipError_t (>0) for a runtime failure, NR_EINVAL (-1) for a bad
 *     argument.  Nothing is thrown across the boundary.
 *   - "+=" outputs accumulate (atomically) into caller-zeroed buffers -- parameter gradients keep
 *     the torch layouts (hash_table [L*T,F]; Linear.weight [out,in]; bias [out]) so optimizers,
 *     state_dict and DDP-style all-reduce see ordinary nn.Parameters.
 *   - Paths cited as file:line are relative to /root/reference/nerfstudio/.
 */
#ifndef NEURADAR_HIP_H
#define NEURADAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NR_EINVAL (-1)
#define NR_MAX_LAYERS 8
#define NR_ABI_VERSION 28
#define NR_DTYPE_F32 0
#define NR_DTYPE_BF16 1
#define NR_DTYPE_F16 2
#define NR_LOSS_SLOTS 1024 /* loss kernels add into loss[0..1023]; the loss value is the sum of the slots */

typedef void* nr_stream_t;

/* Library / ABI version (NR_ABI_VERSION) and the code-object target ("gfx950"). */
int nr_abi_version(void);
const char* nr_target_arch(void);

/* ABI v25: explicit one-time setup instead of lazily-initialised state inside the entry points.
 *   nr_init()        once per DEVICE (the current one: call it after hipSetDevice, before the first launch, from any thread):
 *                    raises hipFuncAttributeMaxDynamicSharedMemorySize for the kernels that stage more than the default 64 KB of
 *                    LDS (nr_conv7_fwd, nr_encoder_*, nr_radar_assign, nr_radar_heads_bwd).  Idempotent.  Without it those entry
 *                    points return hipErrorInvalidValue -- they no longer set the attribute on first use (v24 kept a
 *                    `static bool` per kernel and set it from whichever thread launched first).
 *   nr_set_tuning()  launch-shape knobs for A/B measurements (value 0 = the built-in default, < 0 = NR_EINVAL).  No entry point
 *                    reads the environment any more (v24 called getenv() on every launch); the Python binding forwards its
 *                    NR_* variables once at load (neuradar_amd/_lib.py).  The table is plain process memory: set it at load
 *                    time, not while another thread launches.  This is the library's only process-level state. */
enum {
  NR_TUNE_CONV7_BLOCKS = 0,      /* persistent blocks of nr_conv7_fwd (default 256) */
  NR_TUNE_BIN_BLOCKS_PER_CU = 1, /* bin blocks per HALF CU of the binned scatters (default: what the LDS allows) */
  NR_TUNE_SHARED_BLOCKS = 2,     /* blocks of nr_hash_encode_bwd_shared (default 256) */
  NR_TUNE_FIELD_FWD_BLOCKS = 3,  /* blocks of nr_field_fwd / nr_field_fwd_gather (default min(tiles / 4, 512)) */
  NR_TUNE_FIELD_BWD_BLOCKS = 4,  /* blocks (= gradient slabs) of nr_field_bwd* (default 256 fp32 / 512 16-bit) */
  NR_TUNE_PDBWD_BLOCKS = 5,      /* blocks of nr_prop_density_bwd (default 256) */
  NR_TUNE_ADAM_BLOCKS = 6,       /* blocks of nr_adam_step* (default 4096) */
  NR_TUNE_PW_MFMA_OFF = 7,       /* 1: ConvTranspose2d on the generic pointwise kernels instead of the MFMA ones */
  NR_TUNE_SHARED_SPLIT = 8       /* v28: threads per row of nr_hash_encode_bwd_shared: 1 (default), 2 or 4 share a row's 8 corners */
};                               /* (v27: knobs 8 / 9 of v26 -- NR_TUNE_PROP_SHARED_* -- are gone: nothing read them) */
int nr_init(void);
int nr_set_tuning(int knob, int value);

/* ------------------------------------------------------------------------------------------------
 * Multiresolution hash grid   -- replaces tcnn.Encoding{HashGrid} fwd/bwd reached from
 * field_components/encodings.py:370-373,468-471 (K1/K2/K3 in SURVEY 2a) with the semantics of the
 * torch path encodings.py:406-466: every level hashed, ceil/floor corners, primes
 * {1, 2654435761, 805459861}, one [L*T, F] table, `scalings[l] = floor(min_res * g^l)` (fp32).
 *
 *   x         [n,3]   positions in [0,1]
 *   std       [n] or NULL.  When given, level l of the output is multiplied by
 *             1/max(1, 2*scalings[l]*std)  (NeuRADHashEncoding._rescale_grid_features,
 *             field_components/neurad_encoding.py:309-316), fused into the gather.
 *   table     [L*T, F], T = 2^log2_hashmap_size, F in {1,2,4,8}
 *   out       element (i, l, f) at out[i*out_stride_n + l*out_stride_l + f]; the torch layout
 *             [n, L*F] is (out_stride_n = L*F, out_stride_l = F); the level-major layout the fused
 *             kernels prefer is (F, n*F).
 *   sample_major  0: thread i handles sample i.  S>0: samples are (ray b, sample s) pairs stored
 *             ray-major (i = b*S + s) and lanes are assigned to consecutive RAYS of one sample slot,
 *             which makes the lanes of a wave spatially coherent for camera patches.  Results are
 *             identical either way.
 * ---------------------------------------------------------------------------------------------- */
int nr_hash_encode_fwd(const float* x, const float* std, const float* table, const float* scalings,
                       int num_levels, int features_per_level, int log2_hashmap_size,
                       float* out, int64_t out_stride_n, int64_t out_stride_l,
                       int64_t n, int sample_major, nr_stream_t stream);

/* grad_table [L*T, F] += scatter of grad_out (same strides as `out`).  Backward of the gather
 * (torch autograd of encodings.py:445-466; tcnn's kernel_grid_backward).  Positions get no grad
 * (static scene: sample positions are detached, model_components/ray_samplers.py:364). */
int nr_hash_encode_bwd(const float* x, const float* std, const float* scalings,
                       int num_levels, int features_per_level, int log2_hashmap_size,
                       const float* grad_out, int64_t out_stride_n, int64_t out_stride_l,
                       float* grad_table, int64_t n, int sample_major, nr_stream_t stream);
/* The same with the size of a wave's private merge table chosen by the caller: wave_cells = 0 (the default of the feature
 * width), 128 or 256 cells.  Only F = 4 has two configurations; 256 (512 samples per wave) is for batches that contain
 * incoherent rows (lidar / radar rays): fewer atomics per sample, more of the chip left to the kernels running beside it. */
int nr_hash_encode_bwd_tuned(const float* x, const float* std, const float* scalings,
                             int num_levels, int features_per_level, int log2_hashmap_size,
                             const float* grad_out, int64_t out_stride_n, int64_t out_stride_l,
                             float* grad_table, int64_t n, int sample_major, int wave_cells, nr_stream_t stream);

/* The same, and seen_grad[(row * F + f) / 4] = 1 (bytes, one per four floats of the table, as nr_adam_step's) wherever a sum is
 * added: the optimizer step that follows (nr_adam_step_marked) then leaves never-marked groups alone without reading their
 * gradient.  Single-GPU steps only: a gradient that arrives by a collective marks nothing. */
int nr_hash_encode_bwd_marked(const float* x, const float* std, const float* scalings,
                              int num_levels, int features_per_level, int log2_hashmap_size,
                              const float* grad_out, int64_t out_stride_n, int64_t out_stride_l,
                              float* grad_table, int64_t n, int sample_major, int wave_cells, unsigned char* seen_grad,
                              nr_stream_t stream);

/* The same scatter-add through a BLOCK-SHARED, vertex-keyed LDS table on 32-bit integer atomics (grid_shared.hip): a block of
 * 256 threads takes 256 rows as stored and, level by level, merges their 8 corners each in one table (insert-or-add on the
 * entry index; 32-bit fixed-point sums scaled by the tile's largest gradient entry on that level), then adds every occupied
 * record to grad_table with ONE 16-byte atomic request.  For tables too large for the slice-owner kernels below (NeuRadar's
 * main grid) in steps whose MLPs already run on 16-bit operands: an addend is rounded to 2^-22 ... 2^-21 of the largest
 * gradient entry of its 256-row tile on that level (smaller ones vanish); otherwise equal to nr_hash_encode_bwd up to
 * summation order.  features_per_level = 4, num_levels <= 8, grad_out level-major with out_stride_n = 4 (one float4 per row
 * and level); NR_EINVAL otherwise.  seen_grad: NULL, or the optimizer's bytes as for nr_hash_encode_bwd_marked. */
int nr_hash_encode_bwd_shared(const float* x01, const float* std01, const float* scalings, int num_levels,
                              int features_per_level, int log2_hashmap_size, const float* grad_out, int64_t out_stride_n,
                              int64_t out_stride_l, float* grad_table, int64_t n, unsigned char* seen_grad,
                              nr_stream_t stream);

/* v28: the same with the launch shape chosen by the caller: threads_per_row = 1, 2 or 4 threads share a row's 8 corners (blocks of
 * 256 / 512 / 1 024 threads on the same 256-row tile and the same table: 1 / 2 / 4 waves per SIMD to hide the LDS round trips
 * behind); 0 = the library's default (NR_TUNE_SHARED_SPLIT, else 1).  Same sums whatever the shape (integer addition).  Which
 * is faster depends on what runs beside it: by itself 2 (-18 %), beside the step's proposal scatters 1. */
int nr_hash_encode_bwd_shared_split(const float* x01, const float* std01, const float* scalings, int num_levels,
                                    int features_per_level, int log2_hashmap_size, const float* grad_out, int64_t out_stride_n,
                                    int64_t out_stride_l, float* grad_table, int64_t n, unsigned char* seen_grad,
                                    int threads_per_row, nr_stream_t stream);

/* The same scatter-add by TABLE SLICE OWNERSHIP, for tables a step hits densely (the proposal grids) and for incoherent rows
 * (lidar / radar rays), where the merging of nr_hash_encode_bwd finds little and the launch runs at the memory side's rate for
 * single-entry float atomics.  Two passes: every block merges the contributions of its rows in an LDS table (64-bit
 * fixed-point sums, LDS integer atomics) and leaves one run of records per 64-KB table slice in `workspace`; then one
 * workgroup per slice accumulates its records in LDS and adds the slice to grad_table with contiguous atomics.  Rows are
 * walked as stored; results equal nr_hash_encode_bwd's up to summation order (sums are kept 36 bits below the level's largest
 * value: finer than fp32).  Supported when a level's table is at most 64 slices (T * F <= 2^20 floats: the NeuRadar proposal
 * grids, the L16/F2/T=2^19 main grid); nr_hash_encode_bwd_binned_workspace_bytes returns -1 otherwise (use
 * nr_hash_encode_bwd).  workspace: that many bytes, 16-byte aligned, caller-owned; no initialisation needed. */
int64_t nr_hash_encode_bwd_binned_workspace_bytes(int num_levels, int features_per_level, int log2_hashmap_size, int64_t n);
int nr_hash_encode_bwd_binned(const float* x, const float* std, const float* scalings,
                              int num_levels, int features_per_level, int log2_hashmap_size,
                              const float* grad_out, int64_t out_stride_n, int64_t out_stride_l,
                              float* grad_table, int64_t n, void* workspace, nr_stream_t stream);

/* The proposal field's backward from d loss / d density to its table in ONE pass (NeuRADProposalField.get_density,
 * neurad_field.py:208-213, backward of `trunc_exp(features @ w)` + HashEncoding backward): equal to nr_prop_density_bwd
 * followed by nr_hash_encode_bwd_binned, without the [L, n, F] feature-gradient buffer in between -- every row's
 * g = g_density * d trunc_exp(feats . w) is recomputed from the forward features where the scatter reads them.
 * feats: the forward features in the layout of nr_hash_encode_fwd's `out` (strides as there); w [L*F]; g_density [B,S]
 * ray-major with the row order arguments of nr_prop_density_bwd; grad_table += scatter; g_w [L*F] += sum_rows g * feats.
 * num_levels <= 8; table limits and workspace as for nr_hash_encode_bwd_binned. */
int nr_prop_density_scatter_binned(const float* x, const float* std, const float* scalings,
                                   int num_levels, int features_per_level, int log2_hashmap_size,
                                   const float* feats, int64_t out_stride_n, int64_t out_stride_l,
                                   const float* w, const float* g_density, int n_samples, int64_t rows_sample_major,
                                   float* grad_table, float* g_w, int64_t n, void* workspace, nr_stream_t stream);

/* The same two entry points with the width of the bin pass's fixed-point tile sums as an argument: sum_bits = 64 (the functions
 * above: every addend exact to fp32's resolution) or 32 -- an addend is then rounded to 2^-22..2^-21 of the LARGEST contribution
 * of its 512-row tile (smaller ones vanish), 12 instead of 20 bytes of LDS per slot, ds_add_u32 instead of ds_add_u64, 12-byte
 * records.  Meant for steps whose MLPs already run on 16-bit MFMA operands (nr_field_t.dtype != NR_DTYPE_F32); the same
 * workspace size. */
int nr_hash_encode_bwd_binned_lp(const float* x01, const float* std01, const float* scalings, int num_levels,
                                 int features_per_level, int log2_hashmap_size, const float* grad_out, int64_t out_stride_n,
                                 int64_t out_stride_l, float* grad_table, int64_t n, void* workspace, int sum_bits,
                                 nr_stream_t stream);
int nr_prop_density_scatter_binned_lp(const float* x01, const float* std01, const float* scalings, int num_levels,
                                      int features_per_level, int log2_hashmap_size, const float* feats, int64_t out_stride_n,
                                      int64_t out_stride_l, const float* w, const float* g_density, int n_samples,
                                      int64_t rows_sample_major, float* grad_table, float* g_w, int64_t n, void* workspace,
                                      int sum_bits, nr_stream_t stream);

/* Both proposal rounds in ONE bin pass and ONE apply pass: the rounds evaluate the same field (proposal_fields[1],
 * models/neuradar.py:302), so their gradients scatter into the same table; round 2's row tiles follow round 1's and the
 * apply pass walks every table slice once per step instead of once per round.  Arguments as
 * nr_prop_density_scatter_binned, per round (x, std, feats [L, n, F] level-major with feat_stride_n = sn, g_density [B, S],
 * n_samples, n = B * S); the workspace holds both rounds: nr_hash_encode_bwd_binned_workspace_bytes(L, F, log2T, n1p + n2p)
 * with each n rounded up to a multiple of 512. */
int nr_prop_density_scatter_binned2(const float* x1, const float* std1, const float* feats1, const float* g_density1,
                                    int n_samples1, int64_t n1, const float* x2, const float* std2, const float* feats2,
                                    const float* g_density2, int n_samples2, int64_t n2, int64_t rows_sample_major,
                                    const float* scalings, int num_levels, int features_per_level, int log2_hashmap_size,
                                    int64_t feat_stride_n, const float* w, float* grad_table, float* g_w, void* workspace,
                                    nr_stream_t stream);

/* Single-head self-attention of the radar decoder's transformer encoder layer (detr/models/transformer.py:
 * 176-189 via nn.MultiheadAttention(d_model, 1), models/neuradar.py:250,463-491): out = dropout(softmax(q k^T / sqrt(d))) v
 * per scan, fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products).  q, k, v, out, grad_* [n_scans, n, d]
 * (d in {32, 48, 64}, rows 16-byte aligned); lse [n_scans, n] (the forward's log-sum-exp, kept
 * for the backward).  Dropout on the probabilities: dropout_p in [0, 1), the keep decisions are a hash of (seed, scan, query,
 * key) -- the same in the forward and the backward -- or, when keep_mask [n_scans, n, n] (0 / 1) is not NULL, taken from it.
 * seed_epoch (device float, nullable): a step counter the kernels fold into the seed, so that a captured graph draws new masks
 * on every replay (as nr_uniform_fill's epoch).
 * The backward ACCUMULATES into grad_q / grad_k / grad_v (+=; the caller zeroes them).  workspace:
 * nr_attention_workspace_floats(n_scans, n, d) floats, the same buffer for the forward and its backward is fine. */
int64_t nr_attention_workspace_floats(int64_t n_scans, int64_t n, int d);
int nr_attention_fwd(const float* q, const float* k, const float* v, int64_t n_scans, int64_t n, int d, float dropout_p,
                     uint32_t seed, const float* seed_epoch, const float* keep_mask, float* out, float* lse, float* workspace,
                     nr_stream_t stream);
int nr_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* grad_out,
                     int64_t n_scans, int64_t n, int d, float dropout_p, uint32_t seed, const float* seed_epoch,
                     const float* keep_mask, float* grad_q, float* grad_k, float* grad_v, float* workspace, nr_stream_t stream);

/* 7 x 7 convolution, 32 -> 32 channels, stride 1, padding 3, channels-last 16-bit activations: the convolutions of the RGB
 * decoder's BasicBlocks (model_components/cnns.py:21-47 as built by models/neuradar.py:225-240) on the matrix cores (conv7.hip).
 *   nr_conv7_pack: weights of up to NR_CONV7_MAX convolutions -- each [32][7][7][32] 16-bit, CONTIGUOUS in that order (the
 *     channels-last memory of a torch [O, I, 7, 7] parameter), at element offsets list->offset[k] from weights16, biases [32]
 *     16-bit at list->bias_offset[k] (-1: none) -- into the kernels' LDS images: images[k][0] for the convolution itself
 *     (weights in MFMA fragment order + the bias in fp32), images[k][1] (flipped taps, in / out swapped, no bias) for its data
 *     gradient; nr_conv7_image_bytes() bytes per convolution (both orientations, each half of it), 16-byte aligned.  Once per
 *     optimizer step.
 *   nr_conv7_fwd: y[p, r, c, :] = act( bias + sum_taps W[:, ky, kx, :] x[p, r + ky - 3, c + kx - 3, :] (+ residual[p, r, c, :]) )
 *     (zero padding; act = ReLU when `relu`, residual nullable) for x, y, residual [n_images, height, width, 32] of dtype
 *     NR_DTYPE_BF16 | NR_DTYPE_F16, fp32 accumulation; `image` = one orientation of one convolution's images.  With the
 *     data-gradient image and grad_y as x it returns grad_x. */
#define NR_CONV7_MAX 16
typedef struct nr_conv7_list {
  int n;
  int64_t offset[NR_CONV7_MAX];
  int64_t bias_offset[NR_CONV7_MAX];
} nr_conv7_list_t;
/* Rendering (ABI v27): the eval-mode batch norm that follows each of the n convolutions FOLDED into its weights and bias --
 * W'[o] = W[o] gamma[o] / sqrt(var[o] + eps), b' = (b - mean) gamma / sqrt(var + eps) + beta (model_components/cnns.py:21-47 in
 * eval mode) -- straight from the fp32 master parameters into images[k][0] (the forward orientation of nr_conv7_pack's array), in
 * one launch.  weight[k]: fp32, element (o, tap t = ky * 7 + kx, i) at o * stride_o + t * stride_t + i * stride_i (any memory
 * format); bias[k] nullable.  state: two device uint32 owned by the caller, zero-initialised: the launch EARLY-OUTS on the device
 * while state[0] matches the device's parameter-generation word (nr_param_generation: bumped by every optimizer step, sharded
 * delta apply and batch-norm running-statistics update of this library, graph replays included) unless force != 0 (the caller saw
 * a torch-side change: a version counter, a rebound tensor); state[1] counts the rebuilds. */
typedef struct {
  int n;
  const float* weight[NR_CONV7_MAX];
  int64_t stride_o[NR_CONV7_MAX], stride_t[NR_CONV7_MAX], stride_i[NR_CONV7_MAX];
  const float* bias[NR_CONV7_MAX];
  const float* gamma[NR_CONV7_MAX];
  const float* beta[NR_CONV7_MAX];
  const float* mean[NR_CONV7_MAX];
  const float* var[NR_CONV7_MAX];
  float eps[NR_CONV7_MAX];
} nr_conv7_fold_t;
int nr_conv7_fold_pack(const nr_conv7_fold_t* list, int dtype, void* images, uint32_t* state, int force, nr_stream_t stream);
/* The device's parameter-generation word (device pointer; NULL before nr_init on this device). */
const uint32_t* nr_param_generation(void);
int64_t nr_conv7_image_bytes(void);
int nr_conv7_pack(const void* weights16, const nr_conv7_list_t* list, int dtype, void* images, nr_stream_t stream);
int nr_conv7_fwd(const void* x16, const void* image, const void* residual16, int relu, void* y16, int n_images, int height,
                 int width, int dtype, nr_stream_t stream);
/* Weight and bias gradient of the same convolution: grad_w [32][7][7][32] (the parameter's channels-last memory, 16-bit;
 * overwritten, or "+=" when `accumulate`) = sum over pixels of grad_y[pixel][o] x[pixel + tap - 3][i], grad_b [32] (nullable) =
 * sum over pixels of grad_y;
 * fp32 accumulation on the matrix cores (operands read column-wise from LDS with ds_read_b64_tr_b16), per-block partials in
 * `workspace` (nr_conv7_wgrad_workspace_bytes() bytes, 16-byte aligned, caller-owned) summed by a second launch. */
int64_t nr_conv7_wgrad_workspace_bytes(void);
int nr_conv7_wgrad(const void* x16, const void* grad_y16, void* grad_w16, void* grad_b16, int accumulate, void* workspace,
                   int n_images, int height, int width, int dtype, nr_stream_t stream);

/* The RGB decoder's pointwise convolutions on channels-last activations (pw.hip; models/neuradar.py:225-240: Conv2d(C_in, 32, 1)
 * + ReLU at its head, ConvTranspose2d(32, 32, 3, stride=3) between the block pairs, Conv2d(32, 3, 1) + Sigmoid at its tail):
 *   y[p, n] = act(b[n] + sum_k x[p, k] W[n, k]),  act: 0 none, 1 ReLU, 2 sigmoid;  fp32 accumulation.
 * x [n_pixels, in_channels] fp32 (x_f32 != 0) or 16-bit; y [n_pixels, out_channels] 16-bit or fp32 (y_f32 != 0); w16 / b16 the
 * parameters in `dtype16` (NR_DTYPE_BF16 / NR_DTYPE_F16): a Conv2d weight [out, in, 1, 1] (= [out][in] in memory).
 * transposed != 0: the ConvTranspose2d(3, stride 3) weight [in, out, 3, 3] in its CHANNELS-LAST memory [in][ky][kx][out]; x are
 * n_pixels = images * height * width input pixels, y / grad_y have images * 3 height * 3 width pixels of out_channels, pixel
 * (y, x) of an image owning output pixels (3y + ky, 3x + kx).  in_channels 32 or 48, out_channels <= 32 (a multiple of 8 when
 * transposed).  16-byte aligned arrays.
 *   nr_pw_bwd_data:   grad_x = scale[0] * (grad_y * act'(y)) W  (scale: nullable device float -- the inverse loss scale)
 *   nr_pw_bwd_weight: grad_w16 / grad_b16 (the parameters' memory; nullable bias) = or += (accumulate) the sums over the pixels,
 *                     via per-block partials in `workspace` (nr_pw_workspace_bytes() bytes) + a reduce launch. */
int nr_pw_fwd(const void* x, int x_f32, const void* w16, const void* b16, void* y, int y_f32, int64_t n_pixels, int in_channels,
              int out_channels, int act, int transposed, int height, int width, int dtype16, nr_stream_t stream);
int nr_pw_bwd_data(const void* grad_y, const void* y, int y_f32, const void* w16, void* grad_x, int x_f32, int64_t n_pixels,
                   int in_channels, int out_channels, int act, int transposed, int height, int width, const float* scale, int dtype16,
                   nr_stream_t stream);
int64_t nr_pw_workspace_bytes(void);
int nr_pw_bwd_weight(const void* x, int x_f32, const void* grad_y, const void* y, int y_f32, void* grad_w16, void* grad_b16,
                     int accumulate, void* workspace, int64_t n_pixels, int in_channels, int out_channels, int act, int transposed,
                     int height, int width, int dtype16, nr_stream_t stream);

/* The radar decoder's encoder layer around the attention (detr/models/transformer.py:176-189 `forward_pre` of
 * TransformerEncoderLayer(d_model = C, nhead = 1, dim_feedforward = FF, dropout = p, normalize_before = True), followed by the
 * encoder's final LayerNorm, transformer.py:66-68 -- what models/neuradar.py:463-491 runs per radar scan), encoder.hip:
 *   nr_encoder_pre_fwd   x2 = LN1(x);  [q | k] = (x2 + pos) Wqk^T + bqk;  v = x2 Wv^T + bv        (in_proj rows 0..2C | 2C..3C)
 *   (nr_attention_fwd:   att = softmax(q k^T / sqrt(C)) v, dropout on the probabilities)
 *   nr_encoder_post_fwd  x1 = x + drop(att Wo^T + bo);  x3 = x1 + drop(W2 drop(relu(W1 LN2(x1) + b1)) + b2);  out = LNf(x3)
 *   nr_encoder_post_bwd  grad_out -> grad_att, grad_x1 (the residual path's share of d x) and the gradients of Wo, W1, W2,
 *                        LN2, LNf; recomputes the forward from (x, att)
 *   (nr_attention_bwd:   grad_att -> grad_q, grad_k, grad_v)
 *   nr_encoder_pre_bwd   grad_q / k / v, grad_x1 -> grad_x and the gradients of the in-projection and LN1 (pos: a constant)
 * All arrays fp32 [n, C] row-major (n = tokens of all scans; the layer is row-wise outside the attention); fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: an fp32 fma chain, like the MLP kernels).  C <= 64 (multiple of 4), FF <= 64.
 * Dropout (training): keep decisions are a counter-based hash of (seed + *seed_epoch, site, row, feature) that forward and
 * backward reproduce; p_drop = 0 in eval mode.  Gradients are "+=". */
typedef struct nr_encoder {
  const float* in_proj_weight;  /* [3C, C] */
  const float* in_proj_bias;    /* [3C] */
  const float* out_proj_weight; /* [C, C] */
  const float* out_proj_bias;   /* [C] */
  const float* linear1_weight;  /* [FF, C] */
  const float* linear1_bias;    /* [FF] */
  const float* linear2_weight;  /* [C, FF] */
  const float* linear2_bias;    /* [C] */
  const float* norm1_weight; const float* norm1_bias;   /* [C] each */
  const float* norm2_weight; const float* norm2_bias;
  const float* norm_weight;  const float* norm_bias;    /* the encoder's final LayerNorm */
  int d_model, dim_feedforward;
  float eps;                    /* LayerNorm eps (1e-5) */
  float p_drop;
  uint32_t seed;
  const float* seed_epoch;      /* nullable device-resident step counter folded into the seed (captured graphs) */
} nr_encoder_t;

typedef struct nr_encoder_grads {  /* all "+=", any pointer may be NULL */
  float* in_proj_weight; float* in_proj_bias; float* out_proj_weight; float* out_proj_bias;
  float* linear1_weight; float* linear1_bias; float* linear2_weight; float* linear2_bias;
  float* norm1_weight; float* norm1_bias; float* norm2_weight; float* norm2_bias; float* norm_weight; float* norm_bias;
} nr_encoder_grads_t;

int nr_encoder_pre_fwd(const nr_encoder_t* enc, const float* x, const float* pos, int64_t n, float* q, float* k, float* v,
                       nr_stream_t stream);
int nr_encoder_post_fwd(const nr_encoder_t* enc, const float* x, const float* att, int64_t n, float* out, nr_stream_t stream);
int nr_encoder_post_bwd(const nr_encoder_t* enc, const float* x, const float* att, const float* grad_out, int64_t n,
                        float* grad_att, float* grad_x1, const nr_encoder_grads_t* grads, nr_stream_t stream);
int nr_encoder_pre_bwd(const nr_encoder_t* enc, const float* x, const float* pos, const float* grad_q, const float* grad_k,
                       const float* grad_v, const float* grad_x1, int64_t n, float* grad_x, const nr_encoder_grads_t* grads,
                       nr_stream_t stream);

/* Radar point-set loss on the device (model_components/radar_utils.py:54-168, called from
 * models/neuradar.py:652-662): the reference copies a cost matrix to the host and runs scipy's linear_sum_assignment per scan
 * in the middle of every training step.
 *   pred        [n_scans, n_pred, 7] = (existence probability, x, y, z, three Laplace scales) -- decode_features' radar_output
 *   detections  rows of det_stride >= 3 floats (x, y, z first), scan after scan; seg [n_scans + 1] (device, int32) = first row of
 *               every scan (the reference derives it from radar_indices[:, 1] == 0 with a host sync, :59-61)
 *   max_detections  upper bound of a scan's detection count: sizes the launch and the workspace, the counts themselves are
 *               read on the device.  Limits: min(detections, n_pred) <= 1024, max(...) <= 8192 per scan.
 * nr_radar_assign: cost matrix (cost_type 0 = "euclidean": distance - log r, the one training always uses, :77-78; 1 = "nll",
 * :105-118; MultiBernoulli's clamps :38-45 applied; inf -> 1e9) + rectangular linear sum assignment (shortest augmenting
 * paths, float64 duals -- the algorithm behind scipy.optimize.linear_sum_assignment) -> assoc [n_scans, n_pred] int32:
 * index of the matched detection inside its scan, or -1.  Where the optimum is not unique (exactly equal sums) the choice
 * among the optima may differ from scipy's.
 * nr_radar_loss: loss (NR_LOSS_SLOTS partial sums) += mult * mean_scans( sum_k l_k / n_pred ) with l_k = -log(1 - r_k) for an
 * unmatched prediction and -log r_k + |xyz_k - det| (loss_type 0, :156-164) or -log r_k - Laplace log-likelihood of det
 * (loss_type 1, :132-154) for a matched one; grad_pred [n_scans, n_pred, 7] = its gradient (overwritten). */
int64_t nr_radar_assign_workspace_bytes(int n_scans, int64_t n_pred, int max_detections);
/* nr_radar_assign leaves one int32 STATUS word per scan in the workspace, at this byte offset (-1: bad arguments):
 *   0  assigned;
 *   1  the search gave up (its iteration cap: not observed on finite costs) -- assoc holds the partial matching found so far;
 *   2  the scan has more detections than max_detections, or exceeds the static limits above: NOTHING was assigned -- assoc is -1
 *      for the whole scan, and nr_radar_loss then treats every prediction of it as unmatched (all existence probabilities are
 *      pushed towards 0).  scipy's linear_sum_assignment, which the reference calls, has no such limit: size max_detections
 *      from the data (the largest scan of the dataset) and check the words outside the captured step -- e.g. when a batch's
 *      segments are built, or with a periodic host read (ops.radar_status / DecoderLossHead.check_radar_status). */
int64_t nr_radar_assign_status_offset(int n_scans, int64_t n_pred, int max_detections);
int nr_radar_assign(const float* pred, int n_scans, int64_t n_pred, const float* detections, int det_stride, const int* seg,
                    int max_detections, int cost_type, int* assoc, void* workspace, nr_stream_t stream);
int nr_radar_loss(const float* pred, int n_scans, int64_t n_pred, const float* detections, int det_stride, const int* seg,
                  const int* assoc, int loss_type, float mult, float* grad_pred, float* loss, nr_stream_t stream);

/* Radar rays -> rendered points and their sine position embedding, the head of decode_features' radar branch
 * (models/neuradar.py:470-476: spherical (azimuth, elevation) * depth -> xyz; detr/models/position_encoding_3d.py:56-103,
 * evaluated under no_grad): one launch instead of ~50 elementwise ones.
 *   depth [n], dirs_spher [n, 2] = (azimuth phi, elevation theta); xyz [n, 3] = depth * (cos phi cos theta, sin phi cos theta,
 *   sin theta); dirs [n, 3] = d xyz / d depth (kept for the backward); pos [n, C]: channel c = sin (code[c] & 1 == 0) or cos of
 *   xyz[code[c] >> 1] * 2 pi / dim_t[c], dim_t [C] = temperature^(2 floor(k / 2) / cdim) per axis as the reference computes it.
 * nr_radar_points_bwd: g_depth [n] = <g_xyz, dirs>. */
int nr_radar_points_fwd(const float* depth, const float* dirs_spher, int64_t n, const float* dim_t, const int* code, int C,
                        float* xyz, float* dirs, float* pos, nr_stream_t stream);
int nr_radar_points_bwd(const float* g_xyz, const float* dirs, int64_t n, float* g_depth, nr_stream_t stream);

/* The radar decoder's three heads and the assembly of radar_output in one launch each way (models/neuradar.py:252-278,480-491):
 * head h = MLP in_dim -> 16 -> 16 -> {3, 1, 3} with ReLU hidden layers (h = 0 offset, 1 existence probability, 2 uncertainty;
 * weight[h][l] is layer l's [out, in] matrix, bias[h][l] its bias), x [n, in_dim] the transformer's output, xyz [n, 3] the rendered points:
 *   out [n, 7] = [sigmoid(e), xyz + 1.5 tanh(o), softplus(u)]      (torch.nn.Softplus: identity above 20); in_dim <= 64.
 * nr_radar_heads_bwd: grad_x [n, in_dim] and grad_xyz [n, 3] (overwritten), parameter gradients "+=". */
typedef struct nr_radar_heads {
  const float* weight[3][3];
  const float* bias[3][3];
} nr_radar_heads_t;
typedef struct nr_radar_heads_grads {
  float* weight[3][3];
  float* bias[3][3];
} nr_radar_heads_grads_t;
int nr_radar_heads_fwd(const nr_radar_heads_t* heads, const float* x, int in_dim, const float* xyz, int64_t n, float* out,
                       nr_stream_t stream);
int nr_radar_heads_bwd(const nr_radar_heads_t* heads, const float* x, int in_dim, const float* grad_out, int64_t n, float* grad_x,
                       float* grad_xyz, const nr_radar_heads_grads_t* grads, nr_stream_t stream);

/* Training-mode batch normalisation of a channels-last activation x [M, C] (M = batch * height * width pixels, C = 8, 16, 32
 * or 64; x 16-byte aligned) with the residual add and the ReLU behind it, as the RGB decoder's BasicBlock chains them
 * (model_components/cnns.py:21-47; torch.nn.BatchNorm2d, train mode):
 *   y = act(gamma * (x - mean) / sqrt(var + eps) + beta + residual),  act = ReLU (relu != 0) or identity; residual nullable
 *   mean / var over the M pixels (biased variance); running_mean / running_var (nullable pair) <- (1 - momentum) * running +
 *   momentum * (mean | unbiased variance); save_mean / save_rstd [C] for the backward.
 * nr_bn_act_bwd: from grad_y, with g' = grad_y where y > 0 (relu) -- grad_x = gamma * rstd * (g' - mean(g') - xhat * mean(g' xhat)),
 * grad_residual = g' (nullable), grad_gamma += sum g' xhat, grad_beta += sum g'.
 * dtype of x / residual / y / the gradients: NR_DTYPE_F32 | NR_DTYPE_BF16 | NR_DTYPE_F16; parameters, statistics, sums: float.
 * workspace: nr_bn_act_workspace_floats(M, C) floats, reusable by launches on one stream.  Two launches each way (torch: three
 * MIOpen kernels per normalisation each way + clamp + add + their backwards). */
int64_t nr_bn_act_workspace_floats(int64_t M, int C);
int nr_bn_act_fwd(const void* x, const void* residual, int64_t M, int C, int dtype, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, int relu, void* y, float* save_mean, float* save_rstd,
                  float* workspace, nr_stream_t stream);
int nr_bn_act_bwd(const void* grad_y, const void* y, const void* x, int64_t M, int C, int dtype, const float* gamma,
                  const float* save_mean, const float* save_rstd, int relu, void* grad_x, void* grad_residual, float* grad_gamma,
                  float* grad_beta, float* workspace, nr_stream_t stream);

/* tiny-cuda-nn-compatible multiresolution hash grid for 3-D and 4-D inputs: what
 * `tcnn.Encoding(n_input_dims, {"otype": "HashGrid", n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
 * per_level_scale})` computes with interpolation "Linear" (field_components/encodings.py:361-373,386-401,468-471; the 4-D
 * actor grid: neurad_encoding.py:112-133,282-293), on tcnn's own parameter layout (levels packed back to back, dense
 * indexing while resolution^D <= 2^log2_hashmap_size, hashed above, entry = F contiguous values), so that
 * `tcnn_encoding.params` of a reference checkpoint is usable as it is (as fp32).  x [n,D] in [0,1]; out / grad_out
 * [n, L*F]; grad_params += (not zeroed).  D in {3,4}, F in {1,2,4,8}, L <= 32.  nr_tcnn_grid_param_count: floats of
 * `params` (-1: unsupported configuration).  Restated from tiny-cuda-nn's published sources -- the library is not in the
 * image: see DESIGN.md section 9 for what pins it. */
int64_t nr_tcnn_grid_param_count(int n_dims, int num_levels, int features_per_level, int log2_hashmap_size,
                                 int base_resolution, float per_level_scale);
/* The level geometry the kernels use, into HOST arrays: scales[L] (x * scale + 0.5 = grid position), resolutions[L],
 * offsets[L+1] (entries; level l owns [offsets[l], offsets[l+1])).  scale_l = exp2f(l * log2f(per_level_scale)) * base - 1 is
 * evaluated in float32 on the host; exp2f implementations differ by a few ulp, so a table trained elsewhere sees its
 * positions shifted by up to ~1e-6 of a cell per unit of resolution. */
int nr_tcnn_grid_geometry(int n_dims, int num_levels, int log2_hashmap_size, int base_resolution, float per_level_scale,
                          float* scales, uint32_t* resolutions, uint32_t* offsets);
int nr_tcnn_grid_fwd(const float* x, const float* params, int n_dims, int num_levels, int features_per_level,
                     int log2_hashmap_size, int base_resolution, float per_level_scale, float* out, int64_t n,
                     nr_stream_t stream);
int nr_tcnn_grid_bwd(const float* x, int n_dims, int num_levels, int features_per_level, int log2_hashmap_size,
                     int base_resolution, float per_level_scale, const float* grad_out, float* grad_params, int64_t n,
                     nr_stream_t stream);

/* grad_x [n,3] = d(sum out*grad_out)/dx (overwritten).  Only needed where positions depend on
 * parameters: samples inside dynamic-actor boxes, whose box-frame coordinates follow the learnable
 * trajectories (require_actor_grad, field_components/neurad_encoding.py:83,176). */
int nr_hash_encode_bwd_input(const float* x, const float* std, const float* table, const float* scalings,
                             int num_levels, int features_per_level, int log2_hashmap_size,
                             const float* grad_out, int64_t out_stride_n, int64_t out_stride_l,
                             float* grad_x, int64_t n, nr_stream_t stream);

/* Keyframe interval of each ray's time (utils/poses.py:108-121): right = searchsorted(timestamps, t), left = max(right-1, 0),
 * right clamped to the last stamp, frac = clamp((t - ts[left]) / (ts[right] - ts[left] + 1e-6), 0, 1).  left / right int64 [n]. */
int nr_actor_keyframes(const float* times, int64_t n_rays, const float* timestamps, int n_timestamps, int64_t* left,
                       int64_t* right, float* frac, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dynamic actors without host synchronisation (reference: two `nonzero`s and a Python loop over
 * actors, field_components/neurad_encoding.py:231-275,295-307).  Fixed-shape launches:
 *   nr_actor_candidates  cand [n_rays,K] int32 <- ascending ids of the actors whose bounding sphere (radius = |bounds|) the
 *                        ray's first->last-sample line passes and that exist at the ray's time (:237-246), -1 padded; actors
 *                        beyond K are dropped and *overflow (device int, caller-zeroed) receives max(count).  left/right [n_rays]
 *                        int64 and frac [n_rays] are the keyframe interval of the ray's time (utils/poses.py:117-131), positions
 *                        [T,A,3], present [T,A] uint8, bounds [A,3] = sizes/2 + padding (dynamic_actors.py:95-96).
 *   nr_actor_assign      per sample row (row order as nr_contract_gaussians: sample_major_rows): sphere test (:254-258) and
 *                        exact box test (:263-267) against the ray's candidates with w2b [n_rays,K,3,4] (world->box, e.g. from
 *                        torch so that autograd reaches the trajectories) and centres [n_rays,K,3]; slot_of_row [n] <- index into
 *                        cand (-1: static sample; the LAST matching candidate wins).  For actor samples: box-frame position,
 *                        x *= flip[ray] (+-1 or NULL, :218-225), ScaledSceneContraction(actor_scale) -> x01a [n,3], std01a [n];
 *                        dirs_sample [n_rays*S,3] (ray-major, nullable) <- the view direction of every sample: rotated into the
 *                        box, normalised and flipped for actor samples (:210-215), the ray's direction otherwise.
 *   nr_actor_encode_fwd  feats rows of actor samples <- the actor's grid (tables [A][L*T,F], shared scalings [L]; the grid of
 *                        actor a is tables[table_of_actor[a]] -- DynamicActors.actor_to_id, neurad_encoding.py:183, which the
 *                        closed-loop server rewrites -- or tables[a] when table_of_actor is NULL) with the per-level rescale, levels L..static_levels-1 zeroed (:186-187); F must equal the static grid's F.
 *   nr_actor_encode_bwd  g_tables += scatter of those rows of g_feats, which are then ZEROED (the static grid was overwritten
 *                        there); with g_w2b [n_rays,K,3,4] (caller-zeroed, nullable) += d loss / d w2b through the grid's input
 *                        gradient, the contraction and the flip (require_actor_grad, :83,176).
 * ---------------------------------------------------------------------------------------------- */
int nr_actor_candidates(const float* origins, const float* directions, const float* euclid, int64_t n_rays, int n_samples,
                        const int64_t* left, const int64_t* right, const float* frac, const float* positions,
                        const uint8_t* present, const float* bounds, int n_actors, int K, int* cand, int* overflow,
                        nr_stream_t stream);
/* w2b [n_rays,K,3,4], centres [n_rays,K,3] <- world->box transform and box centre of every (ray, candidate) pair from the
 * learnable trajectories rotations_6d [T,A,6], positions [T,A,3] (dynamic_actors.py:183-197, utils/poses.py:35-49,90-149,
 * cameras/camera_utils.py:422-443): keyframe Gram-Schmidt, linear interpolation at the ray's time, Gram-Schmidt, inverse.
 * Backward: grad_rotations_6d / grad_positions += from grad_w2b (pairs with cand < 0 or a zero gradient are skipped). */
int nr_actor_w2b_fwd(const int* cand, int64_t n_rays, int K, int n_actors, const int64_t* left, const int64_t* right,
                     const float* frac, const float* rotations_6d, const float* positions, float* w2b, float* centres,
                     nr_stream_t stream);
int nr_actor_w2b_bwd(const int* cand, int64_t n_rays, int K, int n_actors, const int64_t* left, const int64_t* right,
                     const float* frac, const float* rotations_6d, const float* positions, const float* grad_w2b,
                     float* grad_rotations_6d, float* grad_positions, nr_stream_t stream);
int nr_actor_assign(const float* origins, const float* directions, const float* pixel_area, const float* euclid,
                    int64_t n_rays, int n_samples, int sample_major_rows, const int* cand, int K, const float* w2b,
                    const float* centres, const float* bounds, float actor_scale, const float* flip, int* slot_of_row,
                    float* x01a, float* std01a, float* dirs_sample, nr_stream_t stream);
int nr_actor_encode_fwd(const float* x01a, const float* std01a, const int* slot_of_row, const int* cand, int K,
                        int64_t n_rays, int n_samples, int sample_major_rows, const float* tables, const int* table_of_actor,
                        const float* scalings, int num_levels, int features_per_level, int log2_hashmap_size, float* feats,
                        int64_t feat_stride_n, int64_t feat_stride_l, int static_levels, nr_stream_t stream);
int nr_actor_encode_bwd(const float* x01a, const float* std01a, const int* slot_of_row, const int* cand, int K,
                        int64_t n_rays, int n_samples, int sample_major_rows, const float* tables, const int* table_of_actor,
                        const float* scalings, int num_levels, int features_per_level, int log2_hashmap_size, float* grad_feats,
                        int64_t feat_stride_n, int64_t feat_stride_l, int static_levels, float* grad_tables, const float* origins,
                        const float* directions, const float* pixel_area, const float* euclid, const float* w2b,
                        float actor_scale, const float* flip, float* grad_w2b, nr_stream_t stream);

/* Frustums.get_fast_isotropic_gaussian(1) (cameras/rays.py:109-124) followed by
 * ScaledSceneContraction(order=inf, scale) on the GaussiansStd
 * (field_components/spatial_distortions.py:103-113,126-136):
 *   origins, directions [n_rays,3]; pixel_area [n_rays]; edges [n_rays, n_samples+1] (metres)
 *   -> x01 [n_rays*n_samples, 3] in [0,1], std01 [n_rays*n_samples].
 * Row order of the outputs: sample_major_rows = number of LEADING rays whose rows are stored
 * sample-major, row s*sm+b (b < sm); the remaining rays keep row b*S+s (the reference's [B,S]
 * flattening; 0 = all of them).  Sample-major rows are what the fused step feeds the hash grid and the
 * MLP kernels for camera patches (rows_sample_major below, same meaning): 64 consecutive rows are then
 * 64 neighbouring rays at one sample slot, which makes every per-sample load coalesced and the touched
 * grid cells coherent; incoherent lidar / radar rays stay ray-major (neighbouring samples of one ray
 * share coarse cells).  Per-ray arrays (densities, alphas, weights, rendered features) are always [B,S]. */
int nr_contract_gaussians(const float* origins, const float* directions, const float* pixel_area,
                          const float* edges, int64_t n_rays, int n_samples, float scale,
                          int sample_major_rows, float* x01, float* std01, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small MLPs on MFMA  -- replaces tcnn.Network{FullyFusedMLP} (field_components/mlp.py:109-113,
 * 180-183; K4/K5/K7) with the torch semantics mlp.py:159-178: Linear+ReLU hidden layers, linear
 * output.  weight[i] is [out_i, in_i] row-major, bias[i] is [out_i] (nn.Linear layouts).
 * Widths: in_dim <= 64, layer width in {16,32,64}, out_dim <= 64.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nr_mlp {
  int num_layers;                         /* number of Linear layers (>=1) */
  int in_dim, width, out_dim;             /* MLP(in_dim, num_layers, layer_width, out_dim) */
  const float* weight[NR_MAX_LAYERS];
  const float* bias[NR_MAX_LAYERS];
} nr_mlp_t;

typedef struct nr_mlp_grads {             /* all "+=" */
  float* weight[NR_MAX_LAYERS];
  float* bias[NR_MAX_LAYERS];
} nr_mlp_grads_t;

/* y [n,out_dim] = MLP(x [n,in_dim]).  MLP.forward, mlp.py:180-183. */
int nr_mlp_fwd(const nr_mlp_t* mlp, const float* x, int64_t n, float* y, nr_stream_t stream);
/* grad_x [n,in_dim] (nullable) and parameter grads += from grad_y [n,out_dim]. */
int nr_mlp_bwd(const nr_mlp_t* mlp, const float* x, const float* grad_y, int64_t n,
               float* grad_x, const nr_mlp_grads_t* grads, nr_stream_t stream);

/* NeuRADField.forward after the hash grid (fields/neurad_field.py:137-148), one launch:
 *   h = mlp_geo(feats); sdf = h[0]; e = h[1:1+C]; sh = SH4((dir+1)/2) (no grad,
 *   encodings.py:797-800 + fields/base_field.py:135-141 + utils/math.py:31-78);
 *   feature = e + mlp_feature([e, sh]); alpha = sigmoid(-sdf*(|beta|+1e-4))
 *   (model_components/utils.py:30-46).
 * feats: element (i, k) at feats[i*feat_stride_n + (k/F)*feat_stride_l + k%F] (same convention as
 * nr_hash_encode_fwd's `out`; F = feat_f).  directions [n_rays,3] per RAY, sample i belongs to
 * ray i / n_samples; n_samples == 0 means directions are per SAMPLE [n,3] (dynamic actors rotate the
 * view direction of the samples inside their boxes, neurad_encoding.py:210-215).
 * Outputs feature [n,C], sdf [n], alpha [n].
 * rows_sample_major = sm > 0 (needs n_samples > 0): the first sm rays' rows of feats / grad_feats are
 * sample-major (row s*sm+b), see nr_contract_gaussians; outputs and their gradients stay at b*n_samples+s. */
typedef struct nr_field {
  nr_mlp_t geo;        /* in_dim = L*F, out_dim = 1 + C */
  nr_mlp_t feat;       /* in_dim = C + 16, out_dim = C  */
  const float* beta;   /* SigmoidDensity.beta, 1 float */
  const float* packed; /* NULL, or the weight image written by nr_field_pack for these weights:
                          lets every block load the weights with one 16-byte-wide copy */
  float* stash;        /* NULL, or nr_field_stash_floats(field, n) floats of caller-owned scratch: nr_field_fwd
                          then leaves the activations the backward needs there (e, mlp_feature's two hidden
                          layers, sdf: 392 B / 648 B per sample at width 32 / 64) and nr_field_bwd, called with the
                          same field, n and inputs, reads them instead of recomputing the forward */
  int dtype;           /* NR_DTYPE_F32 (0): v_mfma_f32_32x32x2_f32, the 1e-4 parity path.  NR_DTYPE_BF16 / NR_DTYPE_F16:
                          16-bit operands on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation (what the reference
                          trains with: torch.autocast + tcnn FullyFusedMLP, engine/trainer.py:189-200,564,
                          field_components/mlp.py:109-127).  Inputs, outputs, parameters and gradients stay fp32 tensors;
                          `packed` is then REQUIRED (nr_field_pack converts the weights) and `stash` is ignored. */
  const float* sample_dirs; /* NULL, or view directions per SAMPLE [n,3] indexed by the ray-major sample index b*S+s (they
                          replace `directions`): dynamic actors rotate the view direction of the samples inside their boxes
                          (nr_actor_assign writes this array); row order of feats is unaffected */
  float grad_scale;    /* reduced precision only: gradients are multiplied by this factor where they enter the 16-bit
                          domain and divided back where they leave it (a static GradScaler, trainer.py:200,585-595);
                          <= 0 means 1.  bf16 needs none; fp16 gradients underflow without it. */
  float* amp;          /* NULL, or the device-resident state of a DYNAMIC loss scale (nr_amp_*, below): the 16-bit backward then
                          takes its scale from amp[NR_AMP_SCALE] instead of grad_scale and stores 1.0f to amp[NR_AMP_FOUND + g] for
                          every optimizer group g of amp_groups when a gradient it writes is inf / NaN (GradScaler's found_inf) */
  unsigned amp_groups; /* bit g set: group g's optimizer consumes gradients that pass through this backward */
} nr_field_t;

typedef struct nr_field_grads {
  nr_mlp_grads_t geo, feat;
  float* beta;
} nr_field_grads_t;

int nr_field_fwd(const nr_field_t* field, const float* feats, int64_t feat_stride_n, int64_t feat_stride_l,
                 int feat_f, const float* directions, int n_samples, int rows_sample_major, int64_t n,
                 float* feature, float* sdf, float* alpha, nr_stream_t stream);
/* Backward: grad_feature [n,C], grad_alpha [n], grad_sdf [n] (nullable) ->
 * grad_feats (same strides as feats, overwritten) and parameter grads +=.
 * workspace: nr_field_bwd_workspace_floats(field, n) floats of caller-owned, 16-byte aligned scratch
 * (d_e / d_sdf rows handed from the feature half to the geometry half, and one weight-gradient slab
 * per block that a last launch sums into the gradients; the library allocates nothing). */
int64_t nr_field_bwd_workspace_floats(const nr_field_t* field, int64_t n);
/* Weight image for nr_field_t.packed: nr_field_image_floats(field) floats, 16-byte aligned; rebuild it
 * (nr_field_pack) whenever the weights change, i.e. once per optimizer step. */
int64_t nr_field_image_floats(const nr_field_t* field);
int64_t nr_field_stash_floats(const nr_field_t* field, int64_t n);
/* nr_hash_encode_fwd of the main grid + nr_field_fwd as ONE launch (fields/neurad_field.py:128-152 with
 * neurad_encoding.py:277-280,309-316 in front): the per-level features go from the gather to the first layer in registers.
 * x01 [n,3], std01 [n] (nr_contract_gaussians, the step's row order), table [L*T, F], scalings [L]; feats_out (nullable): the
 * level-major [L, n, F] copy (level stride feat_stride_l floats) nr_field_bwd recomputes from -- the values
 * nr_hash_encode_fwd(..., std01, ...) stores.  Other arguments and outputs as nr_field_fwd.  Built for num_levels = 8,
 * features_per_level = 4, width 32, 16-bit operands (nr_field_t.dtype != NR_DTYPE_F32); NR_EINVAL otherwise: use the two
 * launches. */
int nr_field_fwd_gather(const nr_field_t* field, const float* x01, const float* std01, const float* table,
                        const float* scalings, int num_levels, int features_per_level, int log2_hashmap_size,
                        float* feats_out, int64_t feat_stride_l, const float* directions, int n_samples,
                        int rows_sample_major, int64_t n, float* feature, float* sdf, float* alpha, nr_stream_t stream);
int nr_field_pack(const nr_field_t* field, float* image, nr_stream_t stream);
int nr_field_bwd(const nr_field_t* field, const float* feats, int64_t feat_stride_n, int64_t feat_stride_l,
                 int feat_f, const float* directions, int n_samples, int rows_sample_major, int64_t n,
                 const float* grad_feature, const float* grad_alpha, const float* grad_sdf,
                 float* grad_feats, const nr_field_grads_t* grads, float* workspace, nr_stream_t stream);
/* grads may be NULL: the per-block weight-gradient slabs then stay in `workspace`, and the caller adds them
 * into the gradients later -- on any stream ordered after nr_field_bwd, e.g. beside the grid scatter that only
 * needs grad_feats -- with nr_field_grad_reduce(field, workspace, n, grads) (same field, workspace and n). */
int nr_field_grad_reduce(const nr_field_t* field, const float* workspace, int64_t n, const nr_field_grads_t* grads,
                         nr_stream_t stream);

/* Degree-4 real spherical harmonics, 16 components: SHEncoding.forward torch path
 * (encodings.py:797-805).  in [n,3] -> out [n,16].  (The caller applies (d+1)/2.) */
int nr_sh4_fwd(const float* dirs, int64_t n, float* out, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal field head  -- NeuRADProposalField.get_density after the grid
 * (fields/neurad_field.py:211-212): density = trunc_exp(feats . w), w [in_dim] (Linear(L*F,1,
 * bias=False)).  Backward uses exp(clamp(x,-15,15)) (field_components/activations.py:28-41).
 * rows_sample_major = sm (needs n_samples): feats / grad_feats rows of the first sm rays are s*sm+b,
 * density and grad_density stay [B,S] (see nr_contract_gaussians).
 * ---------------------------------------------------------------------------------------------- */
int nr_prop_density_fwd(const float* feats, int64_t feat_stride_n, int64_t feat_stride_l, int feat_f,
                        const float* w, int in_dim, int64_t n, int n_samples, int rows_sample_major,
                        float* density, nr_stream_t stream);
/* nr_hash_encode_fwd (sample rows walked as stored, level-major or any strides) followed by
 * nr_prop_density_fwd in ONE launch: feats (kept for the backward) and density [n_rays, n_samples]. */
int nr_prop_field_fwd(const float* x, const float* std, const float* table, const float* scalings,
                      int num_levels, int features_per_level, int log2_hashmap_size, const float* w,
                      float* feats, int64_t feat_stride_n, int64_t feat_stride_l, int64_t n, int n_samples,
                      int rows_sample_major, float* density, nr_stream_t stream);
int nr_prop_density_bwd(const float* feats, int64_t feat_stride_n, int64_t feat_stride_l, int feat_f,
                        const float* w, int in_dim, int64_t n, int n_samples, int rows_sample_major,
                        const float* density, const float* grad_density, float* grad_feats, float* grad_w,
                        nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sampling  (model_components/ray_samplers.py)
 * ---------------------------------------------------------------------------------------------- */
/* PowerSampler/SpacedSampler.generate_ray_samples (ray_samplers.py:98-132,838-852) with
 * power_fn / inv_power_fn (utils/math.py:541-579).  nears, fars [n_rays]; t_rand [n_rays, S+1] or
 * NULL (eval).  -> spacing [n_rays,S+1] (s-space edges), euclid [n_rays,S+1] (metres). */
int nr_power_bins(const float* nears, const float* fars, const float* t_rand, int64_t n_rays, int n_samples,
                  float lambda, float scaling, float* spacing, float* euclid, nr_stream_t stream);
/* nr_power_bins followed by nr_contract_gaussians of the same samples in one launch (the step's first
 * hash-grid launch can follow directly). */
int nr_power_bins_contract(const float* nears, const float* fars, const float* t_rand, const float* origins,
                           const float* directions, const float* pixel_area, int64_t n_rays, int n_samples,
                           float lambda, float scaling, float contraction_scale, int sample_major_rows,
                           float* spacing, float* euclid, float* x01, float* std01, nr_stream_t stream);

/* RaySamples.get_weights (cameras/rays.py:188-210): density [n_rays,S], euclid [n_rays,S+1] ->
 * weights [n_rays,S] (wavefront scan). */
int nr_weights_from_density_fwd(const float* density, const float* euclid, int64_t n_rays, int n_samples,
                                float* weights, nr_stream_t stream);
int nr_weights_from_density_bwd(const float* density, const float* euclid, const float* grad_weights,
                                int64_t n_rays, int n_samples, float* grad_density, nr_stream_t stream);

/* PDFSampler.generate_ray_samples, include_original=False, histogram_padding=0.01
 * (ray_samplers.py:305-376): weights [n_rays,S] over spacing_in [n_rays,S+1] -> n_out+1 new edges.
 * jitter [n_rays] in [0,1) (single_jitter, :325-326) or NULL (eval, :332-334).  nears/fars give
 * the ray's spacing_to_euclidean_fn (:119-120).  Outputs are detached by construction.
 * sky_distance > 0 additionally applies the "sky field" stretch of models/neuradar.py:578-582 to the
 * LAST edge (euclid += sky_distance - euclid, spacing := 1 - 1e-7); pass 0 for plain PDF sampling. */
int nr_pdf_resample(const float* weights, const float* spacing_in, const float* jitter,
                    const float* nears, const float* fars, int64_t n_rays, int n_in, int n_out,
                    float lambda, float scaling, float sky_distance, float* spacing_out, float* euclid_out,
                    nr_stream_t stream);
/* One proposal round after its density launch (ProposalNetworkSampler.generate_ray_samples,
 * ray_samplers.py:623-666, one iteration) in ONE launch: nr_weights_from_density_fwd ->
 * nr_depth_from_weights -> nr_pdf_resample -> nr_contract_gaussians of the n_out new samples.
 * density [n_rays,S], euclid/spacing_in [n_rays,S+1] -> weights [n_rays,S], depth [n_rays],
 * spacing_out/euclid_out [n_rays,n_out+1], x01 [n_rays*n_out,3], std01 [n_rays*n_out]. */
int nr_proposal_round(const float* density, const float* euclid, const float* spacing_in, const float* jitter,
                      const float* nears, const float* fars, const float* origins, const float* directions,
                      const float* pixel_area, int64_t n_rays, int n_samples, int n_out, float lambda,
                      float scaling, float sky_distance, float contraction_scale, int sample_major_rows,
                      float* weights, float* depth, float* spacing_out, float* euclid_out, float* x01,
                      float* std01, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Compositing  -- replaces nerfacc.render_weight_from_alpha + accumulate_along_rays (batched
 * branch; models/neuradar.py:1016, models/neurad.py:727-728) and the renderers
 * (model_components/renderers.py:59-90,322-350) in the order of neuradar.py:504-517:
 *   w_i = alpha_i prod_{j<i}(1-alpha_j); acc = sum w; w_last += 1-acc; features = sum w*feature;
 *   depth = sum_{i<S-1} w_i*(start_i+end_i)/2.
 * alpha [n_rays,S]; feature [n_rays,S,C]; euclid [n_rays,S+1].
 * -> weights [n_rays,S] (after the sky fix-up), accumulation [n_rays], features [n_rays,C], depth [n_rays]
 * ---------------------------------------------------------------------------------------------- */
int nr_composite_fwd(const float* alpha, const float* feature, const float* euclid, int64_t n_rays,
                     int n_samples, int n_channels, float* weights, float* accumulation, float* features,
                     float* depth, nr_stream_t stream);
/* grad_weights (nullable) is an extra upstream gradient on the returned weights (regularisers). */
int nr_composite_bwd(const float* alpha, const float* feature, const float* euclid, const float* weights,
                     const float* grad_features, const float* grad_depth, const float* grad_accumulation,
                     const float* grad_weights, int64_t n_rays, int n_samples, int n_channels,
                     float* grad_alpha, float* grad_feature, nr_stream_t stream);
/* render_depth_simple alone (proposal depths, neuradar.py:527-528). */
int nr_depth_from_weights(const float* weights, const float* euclid, int64_t n_rays, int n_samples,
                          float* depth, nr_stream_t stream);

/* nerfacc's batched helpers by themselves (nerfacc==0.5.2, packed_info=None / ray_indices=None branches), for callers that
 * use them one by one: models/neuradar.py:1016 `render_weight_from_alpha(alphas) -> (weights, transmittance)`,
 * :1018-1022 `render_weight_from_density(t_starts, t_ends, sigmas) -> (weights, transmittance, alphas)`,
 * models/neurad.py:727-728 / model_components/renderers.py:88,345 `accumulate_along_rays(weights, values)`.
 *   T_i = prod_{j<i}(1 - alpha_j) (also where alpha_i = 0), w_i = alpha_i T_i; no sky fix-up; any n_samples.
 * Density mode (t_starts, t_ends non-null): the input is sigma, alpha = 1 - exp(-sigma (t_end - t_start)), `alphas` receives it.
 * Backward: grad_weights / grad_transmittance / grad_alphas are each nullable; grad_in is d/d(alpha) or d/d(sigma). */
int nr_render_weights_fwd(const float* alphas_or_sigmas, const float* t_starts, const float* t_ends, int64_t n_rays,
                          int n_samples, float* weights, float* transmittance, float* alphas, nr_stream_t stream);
int nr_render_weights_bwd(const float* alphas_or_sigmas, const float* t_starts, const float* t_ends,
                          const float* transmittance, const float* grad_weights, const float* grad_transmittance,
                          const float* grad_alphas, int64_t n_rays, int n_samples, float* grad_in, nr_stream_t stream);
/* out[b][c] = sum_s weights[b][s] * values[b][s][c]; values == NULL: out[b] = sum_s weights[b][s] (n_channels ignored).
 * Backward: grad_weights [n_rays,S] and grad_values [n_rays,S,C] are each nullable. */
int nr_accumulate_fwd(const float* weights, const float* values, int64_t n_rays, int n_samples, int n_channels,
                      float* out, nr_stream_t stream);
int nr_accumulate_bwd(const float* weights, const float* values, const float* grad_out, int64_t n_rays, int n_samples,
                      int n_channels, float* grad_weights, float* grad_values, nr_stream_t stream);

/* Lidar supervision of the weights (the "carving" terms of models/neuradar.py:529-531,537-541,637-638,650 with
 * _compute_is_close_to_lidar :971-994): for a lidar ray, a sample is "close" when |range - t_mid| < carving_epsilon (the ray
 * returned) or t_mid < non_return_distance (it did not); the loss adds weight * sum over the ray's other samples of w_s^2,
 * i.e. sum((w * mask)^2) * weight with mask = is_lidar & !close.  `weight` already contains the level's multiplier and
 * 1 / (number of lidar rays in the batch).  All arrays are per ray [n_rays]. */
typedef struct nr_lidar_sup {
  const uint8_t* is_lidar;
  const uint8_t* did_return;
  const float* range;          /* RayBundle.metadata["directions_norm"]: the measured distance of a lidar ray */
  float carving_epsilon;       /* 0.1   (LossSettings.carving_epsilon, neuradar.py:94) */
  float non_return_distance;   /* 150 m (non_return_lidar_distance, :102) */
  float weight;
  /* lidar depth loss on the level's own rendered depth (the proposal levels' depth_loss_i, neuradar.py:641-648,679-688; no
   * quantile mask there): depth_weight * |target - depth| per lidar ray, target = range (returned) or max(depth, 150 m)
   * (not returned, loss x non_return_loss_mult).  depth_weight already holds prop_lidar_loss_mult * depth_mult / n_lidar;
   * 0 = off.  Read by nr_interlevel_loss_to_density only (the main level's depth loss needs the batch quantile:
   * nr_lidar_depth_quantile / nr_lidar_losses below). */
  float depth_weight;
  float non_return_loss_mult;  /* 0.1 (:104) */
} nr_lidar_sup_t;

/* Lidar losses of the training step on the lidar rays' rendered depth and decoder outputs (models/neuradar.py:612-636 with the
 * multipliers of :690-700), without the reference's boolean-mask indexing and torch.quantile host paths:
 *   unreduced_i = |target_i - depth_i|, target = range (returned ray) or max(depth_i, non_return_distance) (not returned: the
 *   loss x non_return_loss_mult); q = torch.quantile(unreduced, quantile) (linear interpolation between the two order statistics
 *   around quantile * (n - 1)); mask = unreduced < q
 *   depth_loss = depth_mult * mean(unreduced[mask]);  intensity_loss = intensity_mult * mean((sigmoid(y0) - target_intensity)^2
 *   over mask & returned);  ray_drop_loss = ray_drop_mult * BCEWithLogits(y1, !returned) over all lidar rays.
 * The lidar rays are rows [row0, row0 + n) of the per-ray arrays (depth, did_return, range, target_intensity: [n_rays]); y
 * [n, 2] are the lidar decoder's outputs (intensity logit, ray-drop logit).
 * nr_lidar_depth_quantile: unreduced [n] and stats [8] = (x_lo, x_hi, q, 1 / count(mask), 1 / count(mask & returned), ...)
 * (an empty mask -- the reference's mean of an empty tensor is nan -- contributes 0 here).
 * nr_lidar_losses: loss +=; grad_depth [n_rays]: rows of the lidar segment are written; grad_y [n, 2] overwritten. */
typedef struct nr_lidar_losses {
  const uint8_t* did_return;
  const float* range;
  const float* target_intensity;
  int64_t row0, n;
  float non_return_distance;   /* 150 m */
  float non_return_loss_mult;  /* 0.1 */
  float quantile;              /* 0.95 (quantile_threshold, :96) */
  float depth_mult, intensity_mult, ray_drop_mult;
} nr_lidar_losses_t;
int nr_lidar_depth_quantile(const float* depth, const nr_lidar_losses_t* cfg, float* unreduced, float* stats, nr_stream_t stream);
int nr_lidar_losses(const float* depth, const float* y, const nr_lidar_losses_t* cfg, const float* unreduced, const float* stats,
                    float* grad_depth, float* grad_y, float* loss, nr_stream_t stream);

/* One-launch training tail of a ray batch: nr_composite_fwd, then the bench loss
 *   rgb_mult * mean((features - target_features)^2) + depth_mult * mean(|depth - target_depth|)
 *   + distortion_mult * distortion(spacing, weights[:, :S-1])      (losses.py:137-157)
 * and its backward through nr_composite_bwd, all inside the ray's wavefront (the separate entry
 * points above compute the same values; this one saves four launches and their round trips).
 * alpha [n_rays,S], feature [n_rays*S,C], euclid/spacing [n_rays,S+1], targets [n_rays,C]/[n_rays];
 * S <= 64, C <= 32.  Outputs: weights [n_rays,S], accumulation, depth [n_rays], features [n_rays,C],
 * grad_alpha [n_rays,S], grad_feature [n_rays*S,C]; loss (NR_LOSS_SLOTS partial sums) +=.
 * grad_features_extra [n_rays,C] / grad_depth_extra [n_rays] (nullable): gradients arriving from the decoders behind the
 * rendered features / depth, added to the supervision's own; target_features / target_depth may be NULL when rgb_mult /
 * depth_mult is 0 (the decoders' losses supervise instead). */
int nr_render_train(const float* alpha, const float* feature, const float* euclid, const float* spacing,
                    const float* target_features, const float* target_depth, int64_t n_rays, int n_samples,
                    int n_channels, float rgb_mult, float depth_mult, float distortion_mult,
                    float* weights, float* accumulation, float* features, float* depth,
                    float* grad_alpha, float* grad_feature, float* loss, const float* grad_features_extra,
                    const float* grad_depth_extra,
                    const nr_lidar_sup_t* lidar, nr_stream_t stream);
/* grad_features_extra [n_rays,C] (nullable): an upstream gradient on the rendered features from a consumer the launch does
 * not contain (per-ray decoders), added to the supervision term's.  lidar (nullable): carving term on the first S-1
 * samples (the sky sample is dropped before the reference computes non_nearby_weights, neuradar.py:515,537-541). */

/* ------------------------------------------------------------------------------------------------
 * Sensor ray generation (device-side; the reference runs these on CPU workers)
 * ---------------------------------------------------------------------------------------------- */
/* Cameras._generate_rays_from_coords (cameras/cameras.py:596-660,782-804,887-949) behind
 * RayGenerator.forward (model_components/ray_generators.py:47-62).
 * ray_indices [n,3] int64 (camera,row,col); c2w [n_cams,3,4]; fx,fy,cx,cy,cam_times [n_cams];
 * velocities [n_cams,3], rs_offsets [n_cams,2], heights [n_cams] nullable together (rolling shutter);
 * distortion [n_cams,6] = [k1,k2,k3,k4,p1,p2] or NULL: iterative undistortion
 * (cameras/camera_utils.py:655-758); camera_type [n_cams] int32 or NULL: 0 = PERSPECTIVE (:782-787),
 * 1 = FISHEYE (:789-804, the ZOD camera model).
 * -> origins, directions [n,3]; pixel_area, times, directions_norm [n]. */
int nr_gen_rays_camera(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy,
                       const float* cx, const float* cy, const float* cam_times, const float* velocities,
                       const float* rs_offsets, const float* heights, const float* distortion,
                       const int* camera_type, int64_t n,
                       float* origins, float* directions, float* pixel_area, float* times,
                       float* directions_norm, nr_stream_t stream);
/* Lidars._generate_rays_from_points (cameras/lidars.py:356-417).  lidar_indices [n] int64;
 * points [n,point_dim>=5]; l2w [n_lidars,3,4]; scan_times [n_lidars]; velocities [n_lidars,3] or NULL.
 * -> ... did_return [n] uint8 (distance < 1e3). */
int nr_gen_rays_lidar(const int64_t* lidar_indices, const float* points, int point_dim, const float* l2w,
                      const float* scan_times, const float* velocities, int64_t n,
                      float* origins, float* directions, float* pixel_area, float* times,
                      float* directions_norm, uint8_t* did_return, nr_stream_t stream);
/* LidarPointSampler.collate_image_dataset_batch (data/pixel_samplers.py:538-577) followed by nr_gen_rays_lidar, one
 * launch (the reference samples on CPU workers and generates rays there).  Ray i belongs to
 * lidar lidar_order[i / rays_per_lidar] (the reference's randperm shuffle, supplied by the caller; rays_per_lidar =
 * ceil(num_rays / num_lidars)) and looks at point floor(u[i] * points_per_lidar[lidar]) of that lidar's scan (u in
 * [0,1); the product is formed in double like the reference's float64 draw).  points [sum points_per_lidar, point_dim]
 * holds the scans back to back, cum_points[l] = first row of lidar l.  indices [n,2] int64 (nullable) receives
 * (lidar, point).  Outputs as nr_gen_rays_lidar. */
int nr_gen_rays_lidar_sampled(const float* u, int64_t n, int rays_per_lidar, const int64_t* lidar_order,
                              const int64_t* points_per_lidar, const int64_t* cum_points, const float* points, int point_dim,
                              const float* l2w, const float* scan_times, const float* velocities, float* origins,
                              float* directions, float* pixel_area, float* times, float* directions_norm,
                              uint8_t* did_return, int64_t* indices, nr_stream_t stream);
/* order [n] int64 <- the permutation that sorts u [n] (ties by index): a uniformly random permutation of n sensors
 * from n uniform numbers, standing in for torch.randperm in LidarPointSampler (data/pixel_samplers.py:550,560-563)
 * without leaving the device or the captured graph.  n <= 65536 (one workgroup; n is a sensor count). */
int nr_permutation_from_uniform(const float* u, int n, int64_t* order, nr_stream_t stream);
/* RadarPointSampler's choice of scans (data/pixel_samplers.py:640-649): scan_indices [n_scans] = 0..num_radars-1 padded
 * with 0 when num_radars <= n_scans, else floor(u[k] * (num_radars - 1)) -- like the reference's randint(0,
 * num_radars - 1) the LAST scan is never drawn.  Feed the result to nr_gen_rays_radar. */
int nr_sample_radar_scans(const float* u, int n_scans, int64_t num_radars, int64_t* scan_indices, nr_stream_t stream);

/* Radars._generate_rays_from_fov (cameras/radars.py:268-358).  scan_indices [n_scans] int64;
 * the FOV grid is n_az x n_el (azimuth-major) with az_i = (float)(min_az + i*d_az) evaluated in
 * double like torch.arange.  r2w [n_radars,3,4]; scan_times [n_radars].
 * -> rays [n_scans*n_az*n_el]: origins, directions [.,3]; pixel_area, times [.]; directions_spher [.,2]. */
int nr_gen_rays_radar(const int64_t* scan_indices, int64_t n_scans, const float* r2w, const float* scan_times,
                      float min_az, float d_az, int n_az, float min_el, float d_el, int n_el,
                      float* origins, float* directions, float* pixel_area, float* times,
                      float* directions_spher, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Loss tail of a training step.  Each entry adds its (already weighted)
 * loss value into the NR_LOSS_SLOTS partial sums loss[0..NR_LOSS_SLOTS-1] (their sum is the loss) and writes the gradient w.r.t. the tensors it consumes, so no autograd graph
 * is needed between compositing and the backward kernels.
 * ---------------------------------------------------------------------------------------------- */
/* rgb_mult*mean((features[:, :C] - target_f)^2) + depth_mult*mean(|depth - target_d|): the stand-in
 * for the reference's decoders + image/lidar losses (models/neuradar.py:672-704).
 * features rows have stride feat_stride >= C.  -> g_features (same stride, first C columns), g_depth. */
int nr_supervision_loss(const float* features, int feat_stride, const float* target_f, int n_channels,
                        const float* depth, const float* target_d, int64_t n_rays, float rgb_mult,
                        float depth_mult, float* g_features, float* g_depth, float* loss, nr_stream_t stream);
/* distortion_loss (model_components/losses.py:137-157) of the final level: c [n_rays,c_stride]
 * s-space edges, w [n_rays,w_stride] weights, first n_used samples (sky sample dropped,
 * neuradar.py:515,534).  mult * mean over rays.  -> g_w [n_rays,w_stride] (overwritten). */
int nr_distortion_loss(const float* c, int c_stride, const float* w, int w_stride, int n_used, int64_t n_rays,
                       float mult, float* g_w, float* loss, nr_stream_t stream);
/* zipnerf_interlevel_loss (losses.py:626-705) for one proposal level: the final level (c, w, first
 * n_used <= 31 samples, detached) is blurred with a box of half-width `pulse`, integrated and
 * resampled at the proposal edges cp [n_rays,Sp+1]; loss = mult * mean_rays sum_j
 * relu(target_j - wp_j)^2 / (wp_j + 1e-5).  -> g_wp [n_rays,Sp] (overwritten). */
int nr_interlevel_loss(const float* c, int c_stride, const float* w, int w_stride, int n_used, const float* cp,
                       const float* wp, int n_prop_samples, int64_t n_rays, float pulse, float mult, float* g_wp,
                       float* loss, nr_stream_t stream);
/* nr_interlevel_loss followed by nr_weights_from_density_bwd of the proposal level in one launch:
 * density_p [n_rays,Sp], euclid_p [n_rays,Sp+1] -> grad_density_p [n_rays,Sp] (overwritten). */
int nr_interlevel_loss_to_density(const float* c, int c_stride, const float* w, int w_stride, int n_used,
                                  const float* c_prop, const float* w_prop, const float* density_prop,
                                  const float* euclid_prop, int n_prop, int64_t n_rays, float pulse_width,
                                  float mult, float* grad_density_prop, float* loss, const nr_lidar_sup_t* lidar,
                                  nr_stream_t stream);
/* lidar (nullable): + weight * sum_j (w_prop_j * mask_j)^2 over ALL samples of the proposal level (prop_weights_loss_i,
 * neuradar.py:529-531). */

/* Temporal appearance embedding (models/neuradar.py:550-568) concatenated to the rendered features of rays
 * [row0, row0 + n_rows): out [n_rows, C + A] = [features[row] | lerp(table[sensor*E + floor(t/duration*E)], next, ratio)]
 * -- the input rows of a per-ray decoder (:510-512).  table [n_sensors*E, A]; times [n_rays]; sensor_idx [n_rays] int64.
 * Backward: grad_features [n_rays, C] rows [row0, row0+n_rows) <- the first C columns of grad_out; grad_table +=. */
int nr_appearance_concat_fwd(const float* features, int n_channels, const float* table, int app_dim, const float* times,
                             const int64_t* sensor_idx, float duration, int embeds_per_sensor, int64_t row0, int64_t n_rows,
                             float* out, nr_stream_t stream);
int nr_appearance_concat_bwd(const float* grad_out, int n_channels, int app_dim, const float* times, const int64_t* sensor_idx,
                             float duration, int embeds_per_sensor, int64_t row0, int64_t n_rows, float* grad_features,
                             float* grad_table, int64_t table_rows, nr_stream_t stream);
/* Losses of the lidar decoder's two outputs y [n,2] (models/neuradar.py:432-452,624-636 with the multipliers of :690-700):
 * intensity_mult * mean over RETURNING rays of (sigmoid(y0) - target_intensity)^2 (inv_n_returning: device scalar, 1 / their
 * number) + ray_drop_mult * mean BCE-with-logits(y1, !did_return).  The reference additionally masks by the 95 % quantile of
 * the depth residual (a sort): not part of this launch.  -> grad_y [n,2] (overwritten), loss +=. */
int nr_lidar_head_loss(const float* y, const float* target_intensity, const uint8_t* did_return, int64_t n,
                       const float* inv_n_returning, float intensity_mult, float ray_drop_mult, float* grad_y, float* loss,
                       nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: dense Adam/AdamW over a flat parameter buffer, one pass (param, grad, m, v), grad
 * zeroed in the same pass.  Semantics of torch.optim.Adam(W) as configured in
 * configs/method_configs.py:384-409 (eps=1e-15; weight_decay decoupled when `adamw`).
 * `step` (1-based) sets the bias corrections; when `dev_hyper` (device float[3] = {lr, 1-beta1^t,
 * sqrt(1-beta2^t)}) is non-NULL it overrides lr/step so that a captured hipGraph can be replayed
 * while the schedule advances.  Pointers must be 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
int nr_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int adamw,
                 int step, float grad_scale, int zero_grad, const float* dev_hyper, uint8_t* seen_grad,
                 const float* skip, void* delta16, int state_stride, nr_stream_t stream);
/* skip: NULL, or a device float (an nr_amp found-inf flag): when it is non-zero the update is SKIPPED -- parameters and moments
 * stay as they are, the gradient is still cleared when zero_grad (GradScaler.step: "optimizer.step() is skipped if the
 * gradients contain infs or NaNs", engine/optimizers.py:154-166); the value 2.0f skips WITHOUT clearing the gradient (the caller
 * carries it into the next step: an overflowed row-list exchange, GradAllReducer.reduce_sparse).
 * delta16: NULL, or n bf16 values (8-byte aligned): the sharded data-parallel table step -- the update of every element is
 * rounded to bf16, applied as param = param_old + float(delta) and stored (0 where nothing moved); the other replicas apply the
 * same deltas with nr_apply_delta16 and end up bit-identical to the owner (pipelines/base_pipeline.py:305-307's DDP keeps
 * replicas identical by all-reducing gradients; here the table's optimizer runs on 1/world of the rows per rank instead). */
/* param[i] += float(delta16[i]) for i outside [lo, hi) (this rank's own shard: whole 16-byte groups). */
int nr_apply_delta16(float* param, const void* delta16, int64_t n, int64_t lo, int64_t hi, nr_stream_t stream);
/* low16[i] = bf16(grad[i]); grad[i] = 0 -- the reduce-scatter's bf16 send buffer and the clearing of the spent local gradient
 * in one pass. */
int nr_grad_to16_clear(float* grad, void* low16, int64_t n, nr_stream_t stream);
/* seen_grad: NULL, or n/4 bytes owned by the caller, zeroed when exp_avg / exp_avg_sq are zeroed.  Byte i is set
 * the first time parameters 4i..4i+3 receive a non-zero gradient; while it is 0 their moments are known to be
 * zero, and (weight_decay == 0) a zero gradient leaves them untouched without reading the moments.  Exact. */

/* nr_adam_step (Adam, no weight decay) over a buffer whose seen_grad bytes are set by the SCATTER (nr_hash_encode_bwd_marked):
 * a group whose byte is 0 has g = exp_avg = exp_avg_sq = 0 by construction and is skipped after reading the byte alone (nr_adam_step
 * reads its gradient to notice a first arrival: 4 B per parameter of the whole table per step); n % 4 == 0.  Exact. */
int nr_adam_step_marked(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                        float eps, int step, float grad_scale, int zero_grad, const float* dev_hyper,
                        const unsigned char* seen_grad, const float* skip, int state_stride, nr_stream_t stream);
/* state_stride (both entry points): 1 -- exp_avg and exp_avg_sq are two arrays of n floats; 2 -- they are the two halves of ONE
 * array of n/4 records [exp_avg x 4 | exp_avg_sq x 4] (exp_avg_sq == exp_avg + 4, n % 4 == 0): the hash tables' layout, where the
 * live 16-byte groups are scattered and every array touched costs a cache line per live group. */

/* Advance the optimizer step counter step_t[0] (device float, 0-based scheduler step) and refresh
 * dev_hyper = {lr(step), 1-beta1^(step+1), sqrt(1-beta2^(step+1))} with the reference's
 * ExponentialDecayScheduler (engine/schedulers.py:112-143: cosine ramp from 1e-8 over `warmup` steps,
 * then log-linear decay lr -> lr_final until max_steps).  One launch, graph-replayable. */
int nr_adam_hyper(float* step_t, float* dev_hyper, float lr, float lr_final, int warmup, int max_steps,
                  float beta1, float beta2, const float* amp, int amp_group, nr_stream_t stream);
/* step_t: TWO device floats -- [0] the scheduler's step count, [1] the optimizer's own step count (bias corrections); without
 * `amp` they advance together.  With amp (nr_amp state) and amp_group (this optimizer's group): a step that the previous
 * nr_amp_update recorded as skipped is not counted -- the scheduler's when ANY group was skipped (the reference steps its
 * schedulers only if the scale did not decrease, engine/trainer.py:590-594), the optimizer's own when ITS step was skipped
 * (torch.optim.Adam's `step` counts performed updates). */

/* ------------------------------------------------------------------------------------------------
 * Dynamic loss scale with found-inf guard: torch.cuda.amp.GradScaler (engine/trainer.py:200,572-594;
 * engine/optimizers.py:154-166) as device-resident state -- no host read anywhere, graph-replayable.
 *   amp [NR_AMP_FLOATS] floats:
 *     [NR_AMP_SCALE] the loss scale S        [NR_AMP_GROWTH_TRACKER] steps without inf / NaN since the last change
 *     [NR_AMP_INV_SCALE] 1 / S               [NR_AMP_SKIPPED_PREV] 1 if the last finished step skipped any optimizer
 *     [NR_AMP_SKIPPED_TOTAL] skipped steps so far
 *     [NR_AMP_FOUND + g] found_inf of optimizer group g in the CURRENT step: producers (the 16-bit field backward,
 *        nr_nonfinite_check, nr_unscale_add_16) store 1.0f; the group's Adam launches take it as `skip`
 *     [NR_AMP_FOUND_PREV + g] the same flags of the previous step (nr_adam_hyper reads them)
 *   nr_amp_update (once per step, after every optimizer launch): found anywhere -> S *= backoff, tracker = 0; else tracker += 1 and
 *   S *= growth when it reaches growth_interval (GradScaler.update); copies the flags to FOUND_PREV and clears them.
 * ---------------------------------------------------------------------------------------------- */
#define NR_AMP_MAX_GROUPS 8
#define NR_AMP_SCALE 0
#define NR_AMP_GROWTH_TRACKER 1
#define NR_AMP_INV_SCALE 2
#define NR_AMP_SKIPPED_PREV 3
#define NR_AMP_SKIPPED_TOTAL 4
#define NR_AMP_FOUND 8
#define NR_AMP_FOUND_PREV (NR_AMP_FOUND + NR_AMP_MAX_GROUPS)
#define NR_AMP_FLOATS (NR_AMP_FOUND + 2 * NR_AMP_MAX_GROUPS)
int nr_amp_init(float* amp, float init_scale, nr_stream_t stream);
int nr_amp_update(float* amp, int n_groups, float growth_factor, float backoff_factor, int growth_interval, nr_stream_t stream);
/* flag[0] = 1.0f if any of x[0..n) is inf / NaN (left alone otherwise): found_inf of a gradient buffer that no flagging
 * producer covers (torch's _amp_foreach_non_finite_check_and_unscale_ without the unscale). */
int nr_nonfinite_check(const float* x, int64_t n, float* flag, nr_stream_t stream);
/* Gradients of 16-bit working copies folded into the fp32 gradient buffer they mirror, unscaled:
 *   dst[i] = (dst[i] + float(src[i])) * inv_scale[0];  src[i] = 0;  flag[0] = 1.0f if a result is inf / NaN.
 * src: n 16-bit floats (src_dtype NR_DTYPE_BF16 / NR_DTYPE_F16); inv_scale: device float or NULL (= 1); flag nullable.
 * dst must hold only this step's (scaled) contributions (the fused optimizers clear gradients every step). */
int nr_unscale_add_16(float* dst, void* src, int64_t n, int src_dtype, const float* inv_scale, float* flag, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse exchange of a hash table's gradient between data-parallel ranks (replaces the dense DDP
 * all-reduce of pipelines/base_pipeline.py:305-307 for tables of which a step touches ~1 % of the rows).
 * nr_grad_compact: grad [rows, F]; every non-zero row is moved to (idx[pos], val[pos,F]) and zeroed;
 *   *count += number of non-zero rows (the caller zeroes it); rows beyond `cap` stay in grad.
 * nr_grad_apply: grad[idx[i]] += val[i] for i < min(*count, cap); indices of one list are unique, the
 *   caller applies the ranks' lists one after the other so that every rank adds in the same order.
 * ---------------------------------------------------------------------------------------------- */
int nr_grad_compact(float* grad, int64_t rows, int row_width, int64_t cap, int* idx, float* val, int* count,
                    nr_stream_t stream);
int nr_grad_apply(const int* idx, const float* val, const int* count, int64_t cap, int row_width, float* grad,
                  nr_stream_t stream);
/* nr_grad_apply of the list of rank `list_rank`, guarded by the TRUE row counts of all `world` ranks (device, as gathered;
 * counts[r] > m means rank r's list of capacity m overflowed): if any list overflowed, only the OWN list is applied (the local
 * gradient is whole again) and flag[0] = 2.0f -- the value that makes nr_adam_step skip the step and keep the gradient for the
 * next one; otherwise the list is applied and flag[0] = 0.  Launch once per rank, in rank order.  No host read: every rank sees
 * the same counts and takes the same branch. */
int nr_grad_apply_guarded(const int* idx, const float* val, const int* counts, int world, int list_rank, int own_rank, int64_t m,
                          int row_width, float* grad, float* flag, nr_stream_t stream);

/* Row lists to the SHARD OWNERS: the gradient half of the sharded data-parallel table step (ABI v27).  The table's rows are
 * owned in `world` contiguous shards of rows_per_shard rows; instead of a dense reduce-scatter every rank sends each owner the
 * rows of that owner's shard it has a gradient for (9-22 % of the rows per rank on the mixed batch), the owner adds the lists
 * in rank order with plain adds (fp32, exact, the same sums whatever the arrival order) -- the reduction half of the
 * reference's DDP all-reduce (pipelines/base_pipeline.py:305-307) for a gradient that is sparse.
 * nr_grad_compact_shards: grad [world * rows_per_shard, F]; for every destination d the non-zero rows of shard d move to segment
 *   d of (idx, val) -- segment d starts at row sum(caps[0..d)) of the lists and holds caps[d] rows; idx = row - d * rows_per_shard
 *   -- and are zeroed in grad; counts[d] += ALL non-zero rows of the shard (the caller zeroes counts; rows beyond caps[d] stay
 *   in grad).  caps, counts: device int32 [world].
 * nr_grad_lists_apply: on the owner `own_rank`: the list received from `src_rank` (capacity list_cap = caps[own_rank]) is added
 *   onto `shard` (the rank's own rows of the gradient), guarded by the all-gathered counts [world, world] (source-major): if ANY
 *   list of the exchange overflowed its destination's capacity nothing is applied and flag[0] = 2.0f (nr_adam_step's "skip and
 *   keep the gradient"); otherwise flag[0] = 0.  Launch once per source, in rank order.
 * nr_grad_lists_restore: overflow only (a no-op otherwise): every segment of the rank's own send lists is added back onto its
 *   local gradient, which is then whole again and carries over into the next step.  max_cap = max(caps).  found_inf (nullable;
 *   the loss scaler's flag of the table's optimizer, already summed over the ranks): when it is raised the step's gradient is
 *   rejected, not kept -- nothing is restored and a second launch clears the whole local gradient (the rows the lists had no
 *   room for are still in it), so that the skipped step leaves no inf / NaN behind on any rank. */
int nr_grad_compact_shards(float* grad, int64_t rows_per_shard, int row_width, int world, const int* caps, int* idx, float* val,
                           int* counts, nr_stream_t stream);
int nr_grad_lists_apply(const int* idx, const float* val, int64_t list_cap, const int* counts, const int* caps, int world,
                        int src_rank, int own_rank, int row_width, float* shard, float* flag, nr_stream_t stream);
int nr_grad_lists_restore(const int* idx, const float* val, int64_t max_cap, const int* counts, const int* caps, int world,
                          int own_rank, int64_t rows_per_shard, int row_width, float* grad, const float* found_inf,
                          nr_stream_t stream);

/* On-device batch assembly for camera patches (the reference samples patches
 * in data/pixel_samplers.py and generates rays on CPU workers): u [n_patches,3] uniform [0,1) ->
 * patch (camera, y0, x0) = (floor(u0*n_cams), floor(u1*(H-span)), floor(u2*(W-span))), span =
 * patch*stride; ray k of patch p is pixel (y0 + stride*(k / patch), x0 + stride*(k % patch)).
 * Rays are generated exactly as nr_gen_rays_camera; pixel_area is multiplied by area_scale
 * (_scale_pixel_area, models/neuradar.py:996-1008).  ray_indices [n,3] int64 (nullable) receives the
 * (camera,row,col) triples.  n = n_patches*patch*patch. */
int nr_gen_rays_camera_patches(const float* u, int64_t n_patches, int n_cams, int height, int width, int patch,
                               int stride, float area_scale, const float* c2w, const float* fx, const float* fy,
                               const float* cx, const float* cy, const float* cam_times, const float* velocities,
                               const float* rs_offsets, const float* heights, const float* distortion,
                               const int* camera_type, float* origins, float* directions,
                               float* pixel_area, float* times, float* directions_norm, int64_t* ray_indices,
                               nr_stream_t stream);

/* U[0,1) numbers for the per-step jitters (the reference uses torch.rand: ray_samplers.py:111,326,
 * pixel_samplers.py): counter-based, value i of draw number *epoch (a device-resident float counter, e.g.
 * the optimizer's step; NULL = 0) for this seed.  24 random bits per value, like torch's float32 rand. */
int nr_uniform_fill(float* out, int64_t n, uint32_t seed, const float* epoch, nr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ABI v26: the main table's optimizer step in two launches around the scatter.
 *   nr_hash_mark_vertices   stamp[level * T + entry] = value for the 8 vertices (floor, floor + 1 per axis) of every row's cell on
 *                           every level: the entries this step's gradient can reach (a superset of what the scatter writes).
 *                           x [n,3] in [0,1], scalings [L], stamp uint8 [L * T] (one byte per 4-float group of an F = 4 table).
 *   nr_adam_step_split      nr_adam_step_marked's update (no weight decay, no skip flag), phase 1: groups with seen_grad != 0 whose
 *                           stamp differs from the step's value -- zero gradient, the gradient buffer is neither read nor written;
 *                           phase 2: groups stamped with the step's value -- full update, gradient cleared.  Together: exactly one
 *                           update per group, bit-identical to the single launch.  Phase 1 may run before / beside the step's
 *                           forward and backward (it writes only groups the step does not read).
 *   value = (int)epoch[0] % 255 + 1, epoch = a device float that counts steps (FlatAdam.step_t[1]): graph replays stamp every step
 *   differently without a host-side argument.  lr and the bias corrections come from dev_hyper (nr_adam_hyper).
 * ---------------------------------------------------------------------------------------------- */
int nr_hash_mark_vertices(const float* x, const float* scalings, int L, int log2T, int64_t n, uint8_t* stamp, const float* epoch,
                          nr_stream_t stream);
int nr_adam_step_split(float* param, float* grad, float* m, float* v, int64_t n, float beta1, float beta2, float eps, float grad_scale,
                       const float* dev_hyper, const uint8_t* seen_grad, const uint8_t* stamp, const float* epoch, int phase,
                       int state_stride, nr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEURADAR_HIP_H */
