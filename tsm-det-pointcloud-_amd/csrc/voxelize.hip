// voxelize.hip — PointToVoxel hard voxelisation on the GPU, bit-exact with the sequential CPU semantics
// (first-occurrence voxel order, <= max_points points per voxel in input order, <= max_voxels voxels per
// frame), with MeanVFE fused into the gather.  SURVEY.md §8a rows a1, a5.
//
// The sequential rules are restated as order statistics so that they parallelise without any sort:
//   * a voxel's creation order is the order of its SMALLEST point index ("creator");
//     exclusive prefix-sum of creator flags over the point axis = first-occurrence voxel id;
//   * the points a voxel keeps are its max_points smallest point indices.  They are collected by a
//     lock-free sorted insert: slot t of a voxel receives atomicMin(slot, p) and the larger of
//     (old, p) is carried to slot t+1.  Each step preserves the multiset {slot} U {carried}, so
//     slot t ends as the (t+1)-th smallest index regardless of interleaving — deterministic.
//   * per-frame voxel caps are applied from per-frame creator counts (frames are contiguous).
// Replaces spconv.utils.Point2VoxelCPU3d.point_to_voxel — reference call sites
// pcdet/datasets/processor/data_processor.py:37-43,55 — plus the concatenation of
// pcdet/datasets/dataset.py:161-229 and MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:26-29).
#include "spx_common.h"

namespace {

constexpr uint64_t kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int32_t kNoPoint = 0x7F7F7F7F;  // memset(0x7F) pattern, larger than any point index we accept
constexpr int kBlock = 256;

struct VoxGeom {
  float lo[3], vs[3];
  int32_t grid[3];  // (gx, gy, gz)
};

struct VoxWs {
  uint64_t* keys;     // [slots]
  int32_t* top;       // [slots * T] sorted smallest point indices per voxel
  int32_t* pt_slot;   // [n] hash slot of each point (-1: dropped)
  uint32_t* crank;    // [n] exclusive creator rank (valid for creators)
  uint32_t* blocksum; // [nblk + 1]
  int32_t* first;     // [B + 1] global creator rank at the first point of each frame
  int32_t* obase;     // [B + 1] output row base of each frame
  int64_t slots, nblk;
  int log2size;
  size_t total;
};

static inline VoxWs vox_layout(void* ws, int64_t n, int batch, int T) {
  VoxWs w;
  w.slots = 1024;
  while (w.slots < 2 * n) w.slots <<= 1;
  w.log2size = 0;
  while ((int64_t(1) << w.log2size) < w.slots) ++w.log2size;
  w.nblk = (n + kBlock - 1) / kBlock;
  char* p = reinterpret_cast<char*>(ws);
  size_t o = 0;
  w.keys = reinterpret_cast<uint64_t*>(p + o);
  o += spx_align((size_t)w.slots * 8);
  w.top = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)w.slots * T * 4);
  w.pt_slot = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)(n + 1) * 4);
  w.crank = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)(n + 1) * 4);
  w.blocksum = reinterpret_cast<uint32_t*>(p + o);
  o += spx_align((size_t)(w.nblk + 1) * 4);
  w.first = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)(batch + 1) * 4);
  w.obase = reinterpret_cast<int32_t*>(p + o);
  o += spx_align((size_t)(batch + 1) * 4);
  w.total = o;
  return w;
}

__device__ __forceinline__ int point_frame(const float* pts, int64_t p, int stride, int batch_col) {
  return batch_col < 0 ? 0 : (int)pts[p * stride + batch_col];
}

// fp32 floor((p - lo) / vsize) with a correctly rounded division: same bits as the CPU semantics
__device__ __forceinline__ bool point_cell(const float* pt, int xyz_col, const VoxGeom& g, int* cx, int* cy, int* cz) {
  int c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float f = floorf(__fdiv_rn(__fsub_rn(pt[xyz_col + j], g.lo[j]), g.vs[j]));
    if (!(f >= 0.0f) || !(f < (float)g.grid[j])) return false;
    c[j] = (int)f;
  }
  *cx = c[0];
  *cy = c[1];
  *cz = c[2];
  return true;
}

__global__ void k_vox_insert(const float* __restrict__ pts, int64_t n, int stride, int xyz_col, int batch_col,
                             int batch, VoxGeom g, int T, VoxWs w, int32_t* status) {
  int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  int cx, cy, cz;
  int b = point_frame(pts, p, stride, batch_col);
  bool ok = (unsigned)b < (unsigned)batch && point_cell(pts + p * stride, xyz_col, g, &cx, &cy, &cz);
  if (!ok) {
    w.pt_slot[p] = -1;
    return;
  }
  uint64_t key = ((((uint64_t)b * g.grid[2] + cz) * g.grid[1] + cy) * g.grid[0]) + cx;
  uint64_t mask = (uint64_t)w.slots - 1;
  uint64_t slot = spx_hash64(key) >> (64 - w.log2size);
  // The probe sequence is bounded by the slot count: the table is sized >= 2n and cleared by the library, so the bound is
  // only reached on a workspace the caller declared pre-cleared (SPX_WS_PRECLEARED) that holds stale keys.  The point is
  // then dropped and the status word receives SPX_ERR_TABLE_FULL — not a wave that spins for ever.
  bool placed = false;
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&w.keys[slot]), kEmpty, key);
    if (old == kEmpty || old == key) {
      placed = true;
      break;
    }
    slot = (slot + 1) & mask;
  }
  if (!placed) {
    w.pt_slot[p] = -1;
    if (status) atomicMin(status, (int32_t)SPX_ERR_TABLE_FULL);
    return;
  }
  w.pt_slot[p] = (int32_t)slot;
  int32_t carry = (int32_t)p;
  int32_t* top = w.top + slot * T;
  for (int t = 0; t < T; ++t) {
    int32_t old = atomicMin(&top[t], carry);
    if (old == kNoPoint) break;       // slot was empty: the carried index now lives there
    if (old > carry) carry = old;     // we displaced a larger index: carry it on
  }
}

__device__ __forceinline__ uint32_t is_creator(const VoxWs& w, int64_t p, int64_t n, int T) {
  if (p >= n) return 0u;
  int32_t s = w.pt_slot[p];
  return (s >= 0 && w.top[(int64_t)s * T] == (int32_t)p) ? 1u : 0u;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[4];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int x = 0; x < wave; ++x) base += wsum[x];
  if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

__global__ void k_vox_blocksum(int64_t n, int T, VoxWs w) {
  int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t total;
  block_excl_scan(is_creator(w, p, n, T), &total);
  if (threadIdx.x == 0) w.blocksum[blockIdx.x] = total;
}

__global__ void k_vox_scan_top(VoxWs w) {
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < w.nblk; base += kBlock) {
    int64_t j = base + threadIdx.x;
    uint32_t v = j < w.nblk ? w.blocksum[j] : 0;
    uint32_t total;
    uint32_t ex = block_excl_scan(v, &total);
    uint32_t carry = carry_s;
    if (j < w.nblk) w.blocksum[j] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) w.blocksum[w.nblk] = carry_s;  // total creators
}

__global__ void k_vox_rank(const float* __restrict__ pts, int64_t n, int stride, int batch_col, int batch, int T,
                           VoxWs w) {
  int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t f = is_creator(w, p, n, T);
  uint32_t ex = w.blocksum[blockIdx.x] + block_excl_scan(f, nullptr);
  if (p >= n) return;
  w.crank[p] = ex;
  int b = point_frame(pts, p, stride, batch_col);
  bool frame_first = p == 0 || point_frame(pts, p - 1, stride, batch_col) != b;
  if (frame_first && (unsigned)b < (unsigned)batch) w.first[b] = (int32_t)ex;
}

// one thread: empty-frame fix-up, per-frame caps, output bases, total M
__global__ void k_vox_frames(int batch, int max_voxels, VoxWs w, int64_t* d_num_voxels) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int32_t total = (int32_t)w.blocksum[w.nblk];
  w.first[batch] = total;
  for (int f = batch - 1; f >= 0; --f)
    if (w.first[f] < 0) w.first[f] = w.first[f + 1];
  int32_t base = 0;
  for (int f = 0; f < batch; ++f) {
    w.obase[f] = base;
    int32_t cnt = w.first[f + 1] - w.first[f];
    base += cnt < max_voxels ? cnt : max_voxels;
  }
  w.obase[batch] = base;
  *d_num_voxels = (int64_t)base;
}

__global__ void k_vox_emit(const float* __restrict__ pts, int64_t n, int stride, int xyz_col, int feat_col, int C,
                           int batch_col, VoxGeom g, int T, int max_voxels, VoxWs w, float* __restrict__ voxels,
                           int32_t* __restrict__ coords, int32_t* __restrict__ num_points, float* __restrict__ mean,
                           int64_t cap) {
  int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (!is_creator(w, p, n, T)) return;
  int b = point_frame(pts, p, stride, batch_col);
  int32_t local = (int32_t)w.crank[p] - w.first[b];
  if (local >= max_voxels) return;  // voxel budget of this frame exhausted: voxel and its points dropped
  int64_t row = (int64_t)w.obase[b] + local;
  if (row >= cap) return;
  int cx, cy, cz;
  point_cell(pts + p * stride, xyz_col, g, &cx, &cy, &cz);
  reinterpret_cast<int4*>(coords)[row] = make_int4(b, cz, cy, cx);
  const int32_t* top = w.top + (int64_t)w.pt_slot[p] * T;
  int cnt = 0;
  for (int t = 0; t < T; ++t) cnt += top[t] != kNoPoint ? 1 : 0;
  num_points[row] = cnt;
  for (int f = 0; f < C; ++f) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
      float v = 0.f;
      if (t < cnt) v = pts[(int64_t)top[t] * stride + feat_col + f];
      if (voxels) voxels[(row * T + t) * C + f] = v;
      s += v;  // same summation order as voxels.sum(dim=1) over the zero-padded slots
    }
    if (mean) mean[row * C + f] = __fdiv_rn(s, (float)(cnt < 1 ? 1 : cnt));
  }
}

__global__ void k_mean_vfe(const float* __restrict__ voxels, const int32_t* __restrict__ num, int64_t n,
                           const int64_t* d_n, int T, int C, float* __restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int64_t v = t / C;
  int f = (int)(t % C);
  if (v >= spx_live_n(d_n, n)) return;
  float s = 0.f;
  for (int j = 0; j < T; ++j) s += voxels[(v * T + j) * C + f];
  int cnt = num[v];
  out[t] = __fdiv_rn(s, (float)(cnt < 1 ? 1 : cnt));
}

}  // namespace

extern "C" size_t spx_voxelize_ws_bytes(int64_t n_points, int batch, int max_points) {
  return vox_layout(nullptr, n_points < 1 ? 1 : n_points, batch < 1 ? 1 : batch, max_points < 1 ? 1 : max_points).total;
}

extern "C" int spx_voxelize(const float* points, int64_t n_points, int point_stride, int xyz_col, int feat_col, int c,
                            int batch_col, int batch, const float* range, const float* vsize, const int32_t* grid,
                            int max_points, int max_voxels, float* voxels, int32_t* coords, int32_t* num_points,
                            float* mean, int64_t* d_num_voxels, int64_t cap, int flags, int32_t* d_status, void* ws,
                            size_t ws_bytes, spx_stream_t stream) {
  if ((!points && n_points > 0) || !range || !vsize || !grid || !coords || !num_points || !d_num_voxels || n_points < 0 ||
      point_stride <= 0 || c <= 0 || xyz_col < 0 || feat_col < 0 || xyz_col + 3 > point_stride ||
      feat_col + c > point_stride || batch_col >= point_stride || batch <= 0 || max_points <= 0 ||
      max_points > 64 || max_voxels <= 0 || cap <= 0 || (flags & ~SPX_WS_PRECLEARED))
    return SPX_ERR_INVALID_ARG;
  if (n_points >= (int64_t)kNoPoint) return SPX_ERR_TOO_LARGE;
  if ((int64_t)batch * grid[0] * grid[1] * grid[2] >= (int64_t(1) << 62)) return SPX_ERR_TOO_LARGE;
  int64_t need = (int64_t)batch * max_voxels;
  if (n_points < need) need = n_points;
  if (cap < need) return SPX_ERR_INVALID_ARG;
  if (!ws || ws_bytes < spx_voxelize_ws_bytes(n_points, batch, max_points)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  if (n_points == 0) {
    spx_fill_async(d_num_voxels, 0, sizeof(int64_t), s);
    return SPX_OK;
  }
  VoxWs w = vox_layout(ws, n_points, batch, max_points);
  VoxGeom g;
  for (int j = 0; j < 3; ++j) {
    g.lo[j] = range[j];
    g.vs[j] = vsize[j];
    g.grid[j] = grid[j];
  }
  if (!(flags & SPX_WS_PRECLEARED)) {
    spx_fill_async(w.keys, 0xFF, (size_t)w.slots * 8, s);
    spx_fill_async(w.top, 0x7F, (size_t)w.slots * max_points * 4, s);
  }
  spx_fill_async(w.first, 0xFF, (size_t)(batch + 1) * 4, s);
  unsigned nb = (unsigned)w.nblk;
  hipLaunchKernelGGL(k_vox_insert, dim3(nb), dim3(kBlock), 0, s, points, n_points, point_stride, xyz_col, batch_col,
                     batch, g, max_points, w, d_status);
  hipLaunchKernelGGL(k_vox_blocksum, dim3(nb), dim3(kBlock), 0, s, n_points, max_points, w);
  hipLaunchKernelGGL(k_vox_scan_top, dim3(1), dim3(kBlock), 0, s, w);
  hipLaunchKernelGGL(k_vox_rank, dim3(nb), dim3(kBlock), 0, s, points, n_points, point_stride, batch_col, batch,
                     max_points, w);
  hipLaunchKernelGGL(k_vox_frames, dim3(1), dim3(64), 0, s, batch, max_voxels, w, d_num_voxels);
  hipLaunchKernelGGL(k_vox_emit, dim3(nb), dim3(kBlock), 0, s, points, n_points, point_stride, xyz_col, feat_col, c,
                     batch_col, g, max_points, max_voxels, w, voxels, coords, num_points, mean, cap);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int spx_mean_vfe(const float* voxels, const int32_t* num_points, int64_t n, const int64_t* d_n,
                            int max_points, int c, float* out, spx_stream_t stream) {
  if (!voxels || !num_points || !out || n < 0 || max_points <= 0 || c <= 0) return SPX_ERR_INVALID_ARG;
  if (n == 0) return SPX_OK;
  int64_t total = n * c;
  hipLaunchKernelGGL(k_mean_vfe, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, spx_s(stream),
                     voxels, num_points, n, d_n, max_points, c, out);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
