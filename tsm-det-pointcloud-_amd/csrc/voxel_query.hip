// voxel_query.hip — neighbour search on the voxel grid for the consumers of multi_scale_3d_features (SURVEY.md §8 row
// f-4).  Replaces voxel_query_kernel_stack and voxel_query_dilated_kernel_stack, reference
// pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-100 and :125-215 (wrappers voxel_query_utils.py:12-50,
// 117-158): one thread per query voxel scans the (2*z_range+1)(2*y_range+1)(2*x_range+1) cells around it in
// (dz, dy, dx) order through the dense voxel -> row table (generate_voxel2pinds), keeps neighbours within `radius`; the
// first `nsample` fill the slots in order, later ones replace a random slot with probability nsample / found (the fork's
// reservoir step).  The dilated form steps the scan by a stride per axis and also drops neighbours closer than
// `former_radius`.  The reference draws from cuRAND's default XORWOW generator seeded with the query index; the
// generator here follows the same published algorithm — that step is PARITY UNPINNED (no CUDA device or vector in this
// environment to confirm the stream), the rest is integer / comparison logic and is checked exactly against the oracle.
#include "spx_common.h"

namespace {

struct Xorwow {
  uint32_t d, v[5];
};

__device__ __forceinline__ void xorwow_init(Xorwow& s, uint64_t seed) {
  const uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
  const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  const uint32_t t0 = 1099087573u * s0;
  const uint32_t t1 = 2591861531u * s1;
  s.d = 6615241u + t1 + t0;
  s.v[0] = 123456789u + t0;
  s.v[1] = 362436069u ^ t0;
  s.v[2] = 521288629u + t1;
  s.v[3] = 88675123u ^ t1;
  s.v[4] = 5783321u + t0;
}

__device__ __forceinline__ uint32_t xorwow_next(Xorwow& s) {
  const uint32_t t = s.v[0] ^ (s.v[0] >> 2);
  s.v[0] = s.v[1];
  s.v[1] = s.v[2];
  s.v[2] = s.v[3];
  s.v[3] = s.v[4];
  s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
  s.d += 362437u;
  return s.v[4] + s.d;
}

__device__ __forceinline__ float xorwow_uniform(Xorwow& s) {
  return __fmaf_rn((float)xorwow_next(s), 2.3283064e-10f, 2.3283064e-10f / 2.0f);
}

__global__ __launch_bounds__(256) void k_voxel_query(const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                     const int32_t* __restrict__ new_coords,
                                                     const int32_t* __restrict__ point_indices, int64_t M, int B, int R1,
                                                     int R2, int R3, int nsample, float radius, int z_range, int y_range,
                                                     int x_range, int z_stride, int y_stride, int x_stride,
                                                     float former_radius, int32_t* __restrict__ idx,
                                                     int32_t* __restrict__ cnt_unique, int32_t* __restrict__ idx_cnt) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= M) return;
  const float qx = new_xyz[3 * p], qy = new_xyz[3 * p + 1], qz = new_xyz[3 * p + 2];
  const int4 c = reinterpret_cast<const int4*>(new_coords)[p];        // (b, z, y, x)
  int32_t* out = idx + p * nsample;
  for (int l = 0; l < nsample; ++l) out[l] = 0;
  const float radius2 = radius * radius, former2 = former_radius * former_radius;
  Xorwow st;
  xorwow_init(st, (uint64_t)p);
  int cnt = 0, cnt2 = 0, in_range = 0;
  if (c.x >= 0 && c.x < B) {
    for (int dz = -z_range; dz <= z_range; dz += z_stride) {
      const int z = c.y + dz;
      if (z < 0 || z >= R1) continue;
      for (int dy = -y_range; dy <= y_range; dy += y_stride) {
        const int y = c.z + dy;
        if (y < 0 || y >= R2) continue;
        for (int dx = -x_range; dx <= x_range; dx += x_stride) {
          const int x = c.w + dx;
          if (x < 0 || x >= R3) continue;
          const int32_t nb = point_indices[(((int64_t)c.x * R1 + z) * R2 + y) * R3 + x];
          if (nb < 0) continue;
          ++in_range;
          const float ddx = xyz[3 * (int64_t)nb] - qx, ddy = xyz[3 * (int64_t)nb + 1] - qy, ddz = xyz[3 * (int64_t)nb + 2] - qz;
          // same association as the reference's expression (a*a + b*b) + c*c, no contraction
          const float dist2 = __fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz));
          if (dist2 > radius2 || dist2 < former2) continue;   // former_radius = 0: the plain ball
          ++cnt2;
          if (cnt < nsample) {
            if (cnt == 0)
              for (int l = 0; l < nsample; ++l) out[l] = nb;
            out[cnt] = nb;
            ++cnt;
          } else {
            const float rnd = xorwow_uniform(st);
            if (rnd < __fdiv_rn((float)nsample, (float)cnt2)) {
              const int ins = (int)ceilf(__fmul_rn(xorwow_uniform(st), (float)nsample)) - 1;
              out[ins] = nb;
            }
          }
        }
      }
    }
  }
  cnt_unique[p] = in_range;
  if (idx_cnt) idx_cnt[p] = cnt;
  if (cnt == 0) out[0] = -1;
  for (int l = 0; cnt < nsample; ++l, ++cnt) out[cnt] = out[l];
}


int launch_query(const float* new_xyz, const float* xyz, const int32_t* new_coords, const int32_t* point_indices, int64_t m,
                 int batch, const int32_t* shape3, int nsample, float former_radius, float radius, const int32_t* range3,
                 const int32_t* stride3, int32_t* idx, int32_t* cnt_unique, int32_t* idx_cnt, spx_stream_t stream) {
  if (!shape3 || !range3 || !stride3 || m < 0 || batch <= 0 || nsample <= 0 || !(radius >= 0.f) || !(former_radius >= 0.f))
    return SPX_ERR_INVALID_ARG;
  if (m == 0) return SPX_OK;
  if (!new_xyz || !xyz || !new_coords || !point_indices || !idx || !cnt_unique) return SPX_ERR_INVALID_ARG;
  for (int j = 0; j < 3; ++j)
    if (shape3[j] <= 0 || range3[j] < 0 || stride3[j] <= 0) return SPX_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_voxel_query, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, spx_s(stream), new_xyz, xyz, new_coords,
                     point_indices, m, batch, shape3[0], shape3[1], shape3[2], nsample, radius, range3[0], range3[1], range3[2],
                     stride3[0], stride3[1], stride3[2], former_radius, idx, cnt_unique, idx_cnt);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

}  // namespace

extern "C" int spx_voxel_query(const float* new_xyz, const float* xyz, const int32_t* new_coords,
                               const int32_t* point_indices, int64_t m, int batch, const int32_t* shape3, int nsample,
                               float radius, const int32_t* range3, int32_t* idx, int32_t* cnt_unique,
                               spx_stream_t stream) {
  const int32_t one[3] = {1, 1, 1};
  return launch_query(new_xyz, xyz, new_coords, point_indices, m, batch, shape3, nsample, 0.f, radius, range3, one, idx,
                      cnt_unique, nullptr, stream);
}

extern "C" int spx_voxel_query_dilated(const float* new_xyz, const float* xyz, const int32_t* new_coords,
                                       const int32_t* point_indices, int64_t m, int batch, const int32_t* shape3,
                                       int nsample, float former_radius, float radius, const int32_t* range3,
                                       const int32_t* stride3, int32_t* idx, int32_t* cnt_unique, int32_t* idx_cnt,
                                       spx_stream_t stream) {
  if (!idx_cnt && m > 0) return SPX_ERR_INVALID_ARG;
  return launch_query(new_xyz, xyz, new_coords, point_indices, m, batch, shape3, nsample, former_radius, radius, range3,
                      stride3, idx, cnt_unique, idx_cnt, stream);
}
