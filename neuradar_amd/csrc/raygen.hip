// Sensor ray generation on the device (the reference runs these in CPU worker processes).
// One thread per ray; tiny, bandwidth-trivial kernels whose job is to keep the batch on the GPU.
#include "nr_common.h"

namespace {

constexpr float kEps = 1e-7f;            // camera_utils._EPS
constexpr float kLidarHDiv = 3.0e-3f;    // lidars.py:41
constexpr float kLidarVDiv = 1.5e-3f;    // lidars.py:42
constexpr float kLidarValid = 1.0e3f;    // valid_lidar_distance_threshold

// camera_utils.normalize_with_norm (camera_utils.py:596-610)
__device__ __forceinline__ float normalize3(float (&v)[3]) {
  const float norm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), kEps);
  v[0] /= norm; v[1] /= norm; v[2] /= norm;
  return norm;
}

// radial_and_tangential_undistort (cameras/camera_utils.py:655-758): 10 Newton steps on the OpenCV
// model [k1,k2,k3,k4,p1,p2], step suppressed where |det J| <= 1e-3.
__device__ __forceinline__ void undistort(float& u, float& v, const float* __restrict__ k) {
  const float xd = u, yd = v;
  float x = xd, y = yd;
#pragma unroll 1
  for (int it = 0; it < 10; ++it) {
    const float r = x * x + y * y;
    const float d = 1.0f + r * (k[0] + r * (k[1] + r * (k[2] + r * k[3])));
    const float fx = d * x + 2.0f * k[4] * x * y + k[5] * (r + 2.0f * x * x) - xd;
    const float fy = d * y + 2.0f * k[5] * x * y + k[4] * (r + 2.0f * y * y) - yd;
    const float d_r = k[0] + r * (2.0f * k[1] + r * (3.0f * k[2] + r * 4.0f * k[3]));
    const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
    const float fx_x = d + d_x * x + 2.0f * k[4] * y + 6.0f * k[5] * x;
    const float fx_y = d_y * x + 2.0f * k[4] * x + 2.0f * k[5] * y;
    const float fy_x = d_x * y + 2.0f * k[5] * y + 2.0f * k[4] * x;
    const float fy_y = d + d_y * y + 2.0f * k[5] * x + 6.0f * k[4] * y;
    const float den = fy_x * fx_y - fx_x * fy_y;
    const bool ok = fabsf(den) > 1e-3f;
    x = x + (ok ? (fx * fy_y - fy * fx_y) / den : 0.0f);
    y = y + (ok ? (fy * fx_x - fx * fy_x) / den : 0.0f);
  }
  u = x;
  v = y;
}

__device__ __forceinline__ void camera_ray(int64_t i, int64_t cam, int64_t row, int64_t col, const float* __restrict__ c2w,
                                           const float* __restrict__ fx, const float* __restrict__ fy,
                                           const float* __restrict__ cx, const float* __restrict__ cy,
                                           const float* __restrict__ cam_times, const float* __restrict__ velocities,
                                           const float* __restrict__ rs_offsets, const float* __restrict__ heights,
                                           const float* __restrict__ distortion, const int* __restrict__ camera_type,
                                           float area_scale, float* __restrict__ origins, float* __restrict__ directions,
                                           float* __restrict__ pixel_area, float* __restrict__ times,
                                           float* __restrict__ directions_norm) {
  const float y = (float)row + 0.5f;  // pixel centres (cameras.py:313)
  const float x = (float)col + 0.5f;
  const float fxv = fx[cam], fyv = fy[cam], cxv = cx[cam], cyv = cy[cam];
  const float* pose = c2w + cam * 12;
  // the pixel and its +1 neighbours in x and y (cameras.py:622-624), lens undistortion (:636-653),
  // OpenCV -> OpenGL flip (:656), perspective z = -1 (:782-787) or fisheye (:789-804), rotate (:892-894)
  const bool fisheye = camera_type != nullptr && camera_type[cam] == 1;
  float d[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float u = k == 1 ? (x - cxv + 1.0f) / fxv : (x - cxv) / fxv;
    float v = k == 2 ? (y - cyv + 1.0f) / fyv : (y - cyv) / fyv;
    if (distortion != nullptr) undistort(u, v, distortion + cam * 6);
    v = -v;
    float w = -1.0f;
    if (fisheye) {
      const float theta = fminf(fmaxf(sqrtf(u * u + v * v), 0.0f), 3.14159265358979323846f);
      const float st = sinf(theta);
      u = u * st / theta;
      v = v * st / theta;
      w = -cosf(theta);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) d[k][r] = u * pose[r * 4 + 0] + v * pose[r * 4 + 1] + w * pose[r * 4 + 2];
  }
  const float norm0 = normalize3(d[0]);
  normalize3(d[1]);
  normalize3(d[2]);
  float dx = 0.0f, dy = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    dx += (d[0][r] - d[1][r]) * (d[0][r] - d[1][r]);
    dy += (d[0][r] - d[2][r]) * (d[0][r] - d[2][r]);
  }
  float o[3] = {pose[3], pose[7], pose[11]};
  float t = cam_times[cam];
  if (velocities != nullptr) {  // top-to-bottom rolling shutter (cameras.py:922-939)
    const float off0 = rs_offsets[cam * 2], duration = rs_offsets[cam * 2 + 1] - off0;
    const float dt = y / heights[cam] * duration + off0;
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = o[r] + velocities[cam * 3 + r] * dt;
    t = t + dt;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    origins[i * 3 + r] = o[r];
    directions[i * 3 + r] = d[0][r];
  }
  pixel_area[i] = (sqrtf(dx) * sqrtf(dy)) * area_scale;
  times[i] = t;
  if (directions_norm != nullptr) directions_norm[i] = norm0;
}

__global__ void __launch_bounds__(256)
gen_rays_camera_kernel(const int64_t* __restrict__ ray_indices, const float* __restrict__ c2w,
                       const float* __restrict__ fx, const float* __restrict__ fy, const float* __restrict__ cx,
                       const float* __restrict__ cy, const float* __restrict__ cam_times,
                       const float* __restrict__ velocities, const float* __restrict__ rs_offsets,
                       const float* __restrict__ heights, const float* __restrict__ distortion,
                       const int* __restrict__ camera_type, int64_t n, float* __restrict__ origins,
                       float* __restrict__ directions, float* __restrict__ pixel_area, float* __restrict__ times,
                       float* __restrict__ directions_norm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  camera_ray(i, ray_indices[i * 3 + 0], ray_indices[i * 3 + 1], ray_indices[i * 3 + 2], c2w, fx, fy, cx, cy, cam_times,
             velocities, rs_offsets, heights, distortion, camera_type, 1.0f, origins, directions, pixel_area, times,
             directions_norm);
}

__global__ void __launch_bounds__(256)
gen_rays_camera_patches_kernel(const float* __restrict__ u, int64_t n_patches, int n_cams, int H, int W, int patch, int stride,
                               float area_scale, const float* __restrict__ c2w, const float* __restrict__ fx,
                               const float* __restrict__ fy, const float* __restrict__ cx, const float* __restrict__ cy,
                               const float* __restrict__ cam_times, const float* __restrict__ velocities,
                               const float* __restrict__ rs_offsets, const float* __restrict__ heights,
                               const float* __restrict__ distortion, const int* __restrict__ camera_type,
                               float* __restrict__ origins, float* __restrict__ directions, float* __restrict__ pixel_area,
                               float* __restrict__ times, float* __restrict__ directions_norm,
                               int64_t* __restrict__ ray_indices) {
  const int per = patch * patch;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_patches * per) return;
  const int64_t p = i / per;
  const int k = (int)(i - p * per);
  const int span = patch * stride;
  const int64_t cam = min((int64_t)(u[p * 3 + 0] * (float)n_cams), (int64_t)n_cams - 1);
  const int64_t y0 = min((int64_t)(u[p * 3 + 1] * (float)(H - span)), (int64_t)(H - span - 1));
  const int64_t x0 = min((int64_t)(u[p * 3 + 2] * (float)(W - span)), (int64_t)(W - span - 1));
  const int64_t row = y0 + (int64_t)stride * (k / patch), col = x0 + (int64_t)stride * (k % patch);
  if (ray_indices != nullptr) {
    ray_indices[i * 3 + 0] = cam; ray_indices[i * 3 + 1] = row; ray_indices[i * 3 + 2] = col;
  }
  camera_ray(i, cam, row, col, c2w, fx, fy, cx, cy, cam_times, velocities, rs_offsets, heights, distortion, camera_type,
             area_scale, origins, directions, pixel_area, times, directions_norm);
}

__global__ void __launch_bounds__(256)
gen_rays_lidar_kernel(const int64_t* __restrict__ lidar_indices, const float* __restrict__ points, int point_dim,
                      const float* __restrict__ l2w, const float* __restrict__ scan_times,
                      const float* __restrict__ velocities, int64_t n, float* __restrict__ origins,
                      float* __restrict__ directions, float* __restrict__ pixel_area, float* __restrict__ times,
                      float* __restrict__ directions_norm, uint8_t* __restrict__ did_return) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t li = lidar_indices[i];
  const float* pose = l2w + li * 12;
  const float* p = points + i * point_dim;
  const float dt = p[4];
  float v[3], o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float world = (p[0] * pose[r * 4 + 0] + p[1] * pose[r * 4 + 1] + p[2] * pose[r * 4 + 2]) + pose[r * 4 + 3];
    o[r] = pose[r * 4 + 3];
    if (velocities != nullptr) o[r] = o[r] + dt * velocities[li * 3 + r];  // lidars.py:378-380
    v[r] = world - o[r];
  }
  const float dist = normalize3(v);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    origins[i * 3 + r] = o[r];
    directions[i * 3 + r] = v[r];
  }
  pixel_area[i] = kLidarHDiv * kLidarVDiv;
  times[i] = scan_times[li] + dt;
  directions_norm[i] = dist;
  did_return[i] = dist < kLidarValid ? 1 : 0;
}

// LidarPointSampler.collate_image_dataset_batch (data/pixel_samplers.py:538-577) + LidarRayGenerator in one launch:
// ray i belongs to slot i / rays_per_lidar of the shuffled lidar order, its point is floor(u_i * n_points[lidar]).
__global__ void __launch_bounds__(256)
gen_rays_lidar_sampled_kernel(const float* __restrict__ u, int64_t n, int rays_per_lidar, const int64_t* __restrict__ order,
                              const int64_t* __restrict__ points_per_lidar, const int64_t* __restrict__ cum_points,
                              const float* __restrict__ points, int point_dim, const float* __restrict__ l2w,
                              const float* __restrict__ scan_times, const float* __restrict__ velocities,
                              float* __restrict__ origins, float* __restrict__ directions, float* __restrict__ pixel_area,
                              float* __restrict__ times, float* __restrict__ directions_norm, uint8_t* __restrict__ did_return,
                              int64_t* __restrict__ indices) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t li = order[i / rays_per_lidar];
  const int64_t np = points_per_lidar[li];
  int64_t pt = (int64_t)floor((double)u[i] * (double)np);  // the reference draws float64 rands (:553-554)
  pt = pt < np - 1 ? pt : np - 1;
  if (indices != nullptr) { indices[i * 2] = li; indices[i * 2 + 1] = pt; }
  const float* pose = l2w + li * 12;
  const float* p = points + (cum_points[li] + pt) * point_dim;
  const float dt = p[4];
  float v[3], o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float world = (p[0] * pose[r * 4 + 0] + p[1] * pose[r * 4 + 1] + p[2] * pose[r * 4 + 2]) + pose[r * 4 + 3];
    o[r] = pose[r * 4 + 3];
    if (velocities != nullptr) o[r] = o[r] + dt * velocities[li * 3 + r];  // lidars.py:378-380
    v[r] = world - o[r];
  }
  const float dist = normalize3(v);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    origins[i * 3 + r] = o[r];
    directions[i * 3 + r] = v[r];
  }
  pixel_area[i] = kLidarHDiv * kLidarVDiv;
  times[i] = scan_times[li] + dt;
  directions_norm[i] = dist;
  did_return[i] = dist < kLidarValid ? 1 : 0;
}

// RadarPointSampler's scan choice (data/pixel_samplers.py:640-649): every radar once when there are at most n_scans of
// them (padded with scan 0), else n_scans draws of randint(0, num_radars - 1) -- the last scan is never drawn.
__global__ void sample_radar_scans_kernel(const float* __restrict__ u, int n_scans, int64_t num_radars, int64_t* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_scans) return;
  if (num_radars <= n_scans) {
    out[k] = k < num_radars ? k : 0;
  } else {
    const int64_t hi = num_radars - 1;
    int64_t s = (int64_t)floor((double)u[k] * (double)hi);
    out[k] = s < hi - 1 ? s : hi - 1;
  }
}

// order <- the permutation that sorts u (ties by index): torch.randperm's role in LidarPointSampler
// (data/pixel_samplers.py:550,560-563), by counting ranks -- n is a number of sensors (hundreds), O(n^2 / threads).
__global__ void __launch_bounds__(256)
permutation_kernel(const float* __restrict__ u, int n, int64_t* __restrict__ order) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float ui = u[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (u[j] < ui || (u[j] == ui && j < i)) ? 1 : 0;
    order[rank] = i;
  }
}

__global__ void __launch_bounds__(256)
gen_rays_radar_kernel(const int64_t* __restrict__ scan_indices, int64_t n_scans, const float* __restrict__ r2w,
                      const float* __restrict__ scan_times, float min_az, float d_az, int n_az, float min_el,
                      float d_el, int n_el, float* __restrict__ origins, float* __restrict__ directions,
                      float* __restrict__ pixel_area, float* __restrict__ times, float* __restrict__ spher) {
  const int64_t per_scan = (int64_t)n_az * n_el;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_scans * per_scan) return;
  const int64_t k = i / per_scan;
  const int rem = (int)(i - k * per_scan);
  const int ia = rem / n_el, ie = rem - ia * n_el;  // azimuth-major meshgrid (radars.py:293-296)
  const int64_t scan = scan_indices[k];
  // torch.arange evaluates start + i*step in double, then rounds to fp32
  const float az = (float)((double)min_az + (double)ia * (double)d_az);
  const float el = (float)((double)min_el + (double)ie * (double)d_el);
  const float* pose = r2w + scan * 12;
  const float l[3] = {cosf(el) * cosf(az), cosf(el) * sinf(az), sinf(el)};  // radars.py:313-316
  float v[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // transform WITH translation, then subtract the origin again (radars.py:317-320)
    const float world = (l[0] * pose[r * 4 + 0] + l[1] * pose[r * 4 + 1] + l[2] * pose[r * 4 + 2]) + pose[r * 4 + 3];
    v[r] = world - pose[r * 4 + 3];
    origins[i * 3 + r] = pose[r * 4 + 3];
  }
  normalize3(v);
#pragma unroll
  for (int r = 0; r < 3; ++r) directions[i * 3 + r] = v[r];
  pixel_area[i] = (d_az / 5.0f) * (d_el / 5.0f);  // radars.py:324-328
  times[i] = scan_times[scan];
  spher[i * 2 + 0] = az;
  spher[i * 2 + 1] = el;
}

// ---- counter-based uniform numbers for the per-step jitters ---------------------------------------
// The reference draws them with torch.rand (ray_samplers.py:111,326; pixel_samplers.py); any U[0,1)
// source serves.  Inside a replayed hipGraph torch's generator costs two extra bookkeeping launches per
// replay, so the step uses this: value i of draw `epoch` = 24 random bits of a PCG-style hash.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

__global__ void __launch_bounds__(256)
uniform_fill_kernel(float* __restrict__ out, int64_t n, uint32_t seed, const float* __restrict__ epoch) {
  const uint32_t e = epoch != nullptr ? (uint32_t)epoch[0] : 0u;
  const uint32_t key = pcg_hash(seed ^ pcg_hash(e * 0x9E3779B9u + 0x85EBCA6Bu));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = pcg_hash((uint32_t)i ^ pcg_hash((uint32_t)(i >> 32) + key));
    out[i] = (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}

}  // namespace

extern "C" int nr_uniform_fill(float* out, int64_t n, uint32_t seed, const float* epoch, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!out || n < 0) return NR_EINVAL;
  const unsigned blocks = (unsigned)(nr_cdiv(n, 1024) < 2048 ? nr_cdiv(n, 1024) : 2048);
  hipLaunchKernelGGL(uniform_fill_kernel, dim3(blocks), dim3(256), 0, nr_s(stream), out, n, seed, epoch);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_gen_rays_camera(const int64_t* ray_indices, const float* c2w, const float* fx, const float* fy,
                                  const float* cx, const float* cy, const float* cam_times, const float* velocities,
                                  const float* rs_offsets, const float* heights, const float* distortion,
                                  const int* camera_type, int64_t n, float* origins, float* directions,
                                  float* pixel_area, float* times, float* directions_norm, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!ray_indices || !c2w || !fx || !fy || !cx || !cy || !cam_times || !origins || !directions || !pixel_area ||
      !times || !directions_norm || n < 0)
    return NR_EINVAL;
  if ((velocities != nullptr) != (rs_offsets != nullptr) || (velocities != nullptr) != (heights != nullptr)) return NR_EINVAL;
  hipLaunchKernelGGL(gen_rays_camera_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), ray_indices,
                     c2w, fx, fy, cx, cy, cam_times, velocities, rs_offsets, heights, distortion, camera_type, n, origins,
                     directions, pixel_area, times, directions_norm);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_gen_rays_camera_patches(const float* u, int64_t n_patches, int n_cams, int H, int W, int patch, int stride,
                                          float area_scale, const float* c2w, const float* fx, const float* fy,
                                          const float* cx, const float* cy, const float* cam_times, const float* velocities,
                                          const float* rs_offsets, const float* heights, const float* distortion,
                                          const int* camera_type, float* origins, float* directions, float* pixel_area,
                                          float* times, float* directions_norm, int64_t* ray_indices, nr_stream_t stream) {
  if (n_patches == 0) return 0;
  if (!u || !c2w || !fx || !fy || !cx || !cy || !cam_times || !origins || !directions || !pixel_area || !times ||
      n_patches < 0 || n_cams < 1 || patch < 1 || stride < 1 || H <= patch * stride || W <= patch * stride)
    return NR_EINVAL;
  if ((velocities != nullptr) != (rs_offsets != nullptr) || (velocities != nullptr) != (heights != nullptr)) return NR_EINVAL;
  const int64_t n = n_patches * patch * patch;
  hipLaunchKernelGGL(gen_rays_camera_patches_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), u, n_patches,
                     n_cams, H, W, patch, stride, area_scale, c2w, fx, fy, cx, cy, cam_times, velocities, rs_offsets, heights,
                     distortion, camera_type, origins, directions, pixel_area, times, directions_norm, ray_indices);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_gen_rays_lidar(const int64_t* lidar_indices, const float* points, int point_dim, const float* l2w,
                                 const float* scan_times, const float* velocities, int64_t n, float* origins,
                                 float* directions, float* pixel_area, float* times, float* directions_norm,
                                 uint8_t* did_return, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!lidar_indices || !points || point_dim < 5 || !l2w || !scan_times || !origins || !directions || !pixel_area ||
      !times || !directions_norm || !did_return || n < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(gen_rays_lidar_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), lidar_indices,
                     points, point_dim, l2w, scan_times, velocities, n, origins, directions, pixel_area, times,
                     directions_norm, did_return);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_gen_rays_lidar_sampled(const float* u, int64_t n, int rays_per_lidar, const int64_t* lidar_order,
                                         const int64_t* points_per_lidar, const int64_t* cum_points, const float* points,
                                         int point_dim, const float* l2w, const float* scan_times, const float* velocities,
                                         float* origins, float* directions, float* pixel_area, float* times,
                                         float* directions_norm, uint8_t* did_return, int64_t* indices, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!u || !lidar_order || !points_per_lidar || !cum_points || !points || point_dim < 5 || rays_per_lidar < 1 || !l2w ||
      !scan_times || !origins || !directions || !pixel_area || !times || !directions_norm || !did_return || n < 0)
    return NR_EINVAL;
  hipLaunchKernelGGL(gen_rays_lidar_sampled_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), u, n,
                     rays_per_lidar, lidar_order, points_per_lidar, cum_points, points, point_dim, l2w, scan_times, velocities,
                     origins, directions, pixel_area, times, directions_norm, did_return, indices);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_permutation_from_uniform(const float* u, int n, int64_t* order, nr_stream_t stream) {
  if (n == 0) return 0;
  if (!u || !order || n < 0 || n > 65536) return NR_EINVAL;
  hipLaunchKernelGGL(permutation_kernel, dim3(1), dim3(256), 0, nr_s(stream), u, n, order);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_sample_radar_scans(const float* u, int n_scans, int64_t num_radars, int64_t* scan_indices, nr_stream_t stream) {
  if (n_scans == 0) return 0;
  if (!scan_indices || n_scans < 0 || num_radars < 1 || (num_radars > n_scans && (!u || num_radars < 2))) return NR_EINVAL;
  hipLaunchKernelGGL(sample_radar_scans_kernel, dim3((unsigned)nr_cdiv(n_scans, 64)), dim3(64), 0, nr_s(stream), u, n_scans,
                     num_radars, scan_indices);
  NR_LAUNCH_CHECK();
  return 0;
}

extern "C" int nr_gen_rays_radar(const int64_t* scan_indices, int64_t n_scans, const float* r2w, const float* scan_times,
                                 float min_az, float d_az, int n_az, float min_el, float d_el, int n_el, float* origins,
                                 float* directions, float* pixel_area, float* times, float* spher, nr_stream_t stream) {
  if (n_scans == 0) return 0;
  if (!scan_indices || !r2w || !scan_times || !origins || !directions || !pixel_area || !times || !spher ||
      n_az < 1 || n_el < 1 || n_scans < 0)
    return NR_EINVAL;
  const int64_t n = n_scans * n_az * n_el;
  hipLaunchKernelGGL(gen_rays_radar_kernel, dim3((unsigned)nr_cdiv(n, 256)), dim3(256), 0, nr_s(stream), scan_indices,
                     n_scans, r2w, scan_times, min_az, d_az, n_az, min_el, d_el, n_el, origins, directions, pixel_area,
                     times, spher);
  NR_LAUNCH_CHECK();
  return 0;
}
