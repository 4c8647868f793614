// assign.hip — anchor target assignment in two launches.
//
// Restates AxisAlignedTargetAssigner.assign_targets / assign_targets_single of the reference
// (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-210, with POS_FRACTION < 0,
// NORM_BY_NUM_EXAMPLES False, MATCH_HEIGHT False — the SECOND configuration), including boxes3d_nearest_bev_iou
// (pcdet/utils/box_utils.py:249-298) and ResidualCoder.encode_torch (pcdet/utils/box_coder_utils.py:13-43):
//   m = max_j IoU(a, j), g = first argmax;   forced(a) = exists j: IoU(a, j) == max_a' IoU(a', j) > 0
//   label = class(g) if (m >= matched or forced) ; 0 if m < unmatched ; -1 otherwise ; all 0 when the frame has no gt
//   target = encode(gt[g], a) where label > 0.
// The reference runs ~60 small torch kernels per (sample, class) with host syncs in between; here
//   k_gt_max : one thread per (frame, anchor set, anchor): IoU against the frame's gt boxes, per-gt maximum by
//              LDS atomicMax then one global atomicMax per block (IoU >= 0, so the float bits order like ints;
//              max is order independent => deterministic);
//   k_assign : recomputes the same IoUs (bitwise the same code path, so the == test is exact), applies the rule and
//              writes labels / targets / weights directly in the head's (y, x, class, rot) anchor order.
// fp32 operations are kept in torch's order with contraction off so that threshold comparisons agree.
#include "spx_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxGt = 256;
constexpr float kPi = 3.14159265358979323846f;

struct Box4 {
  float x1, y1, x2, y2;
};

// boxes3d_lidar_to_aligned_bev_boxes: swap dx/dy when |limit_period(heading, 0.5, pi)| >= pi/4
__device__ __forceinline__ Box4 aligned_bev(const float* b) {
  float v = b[6];
  float rot = fabsf(v - floorf(__fdiv_rn(v, kPi) + 0.5f) * kPi);
  float dx = rot < (kPi / 4) ? b[3] : b[4];
  float dy = rot < (kPi / 4) ? b[4] : b[3];
  Box4 r;
  r.x1 = b[0] - dx / 2;
  r.y1 = b[1] - dy / 2;
  r.x2 = b[0] + dx / 2;
  r.y2 = b[1] + dy / 2;
  return r;
}

__device__ __forceinline__ float iou_normal(const Box4& a, const Box4& b) {
  float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
  float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
  float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
  float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  float inter = w * h;
  return __fdiv_rn(inter, fmaxf(area_a + area_b - inter, 1e-6f));
}

struct AssignArgs {
  const float* anchors;   // [n_sets][A][7]
  const float* gt;        // [B][M][8]
  const int32_t* set_cls; // [n_sets] 0-based index into class_names of each anchor set
  const float* matched;   // [n_sets]
  const float* unmatched; // [n_sets]
  int n_sets, n_classes, per_loc, B, M;
  int64_t A;
};

// valid gt rows of frame b for anchor set s: row <= last row with a non-zero box (row 0 always), class matches.
// class id 0 (padding) indexes class_names[-1], as the reference's numpy indexing does.
__device__ __forceinline__ bool gt_valid(const AssignArgs& g, const int32_t* nkeep, int b, int j, int s) {
  if (j >= nkeep[b]) return false;
  int cls = (int)g.gt[((int64_t)b * g.M + j) * 8 + 7];
  int name_idx = ((cls - 1) % g.n_classes + g.n_classes) % g.n_classes;
  return name_idx == g.set_cls[s];
}

__global__ void k_gt_keep(AssignArgs g, int32_t* nkeep) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= g.B) return;
  int last = 0;
  for (int j = 0; j < g.M; ++j) {
    const float* p = g.gt + ((int64_t)b * g.M + j) * 8;
    float s = 0.f;
    for (int t = 0; t < 7; ++t) s += p[t];
    if (s != 0.f) last = j;
  }
  nkeep[b] = last + 1;
}

__global__ __launch_bounds__(256) void k_gt_max(AssignArgs g, const int32_t* __restrict__ nkeep, int* __restrict__ gmax) {
  __shared__ int smax[kMaxGt];
  const int s = blockIdx.y, b = blockIdx.z;
  for (int j = threadIdx.x; j < g.M; j += 256) smax[j] = 0;
  __syncthreads();
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a < g.A) {
    Box4 ab = aligned_bev(g.anchors + ((int64_t)s * g.A + a) * 7);
    for (int j = 0; j < g.M; ++j) {
      if (!gt_valid(g, nkeep, b, j, s)) continue;   // block-uniform
      float v = iou_normal(ab, aligned_bev(g.gt + ((int64_t)b * g.M + j) * 8));
      if (v > 0.f) atomicMax(&smax[j], __float_as_int(v));
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < g.M; j += 256)
    if (smax[j] > 0) atomicMax(&gmax[((int64_t)b * g.n_sets + s) * g.M + j], smax[j]);
}

__global__ __launch_bounds__(256) void k_assign(AssignArgs g, const int32_t* __restrict__ nkeep,
                                                const int* __restrict__ gmax, int32_t* __restrict__ labels,
                                                float* __restrict__ targets, float* __restrict__ weights) {
  const int s = blockIdx.y, b = blockIdx.z;
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (a >= g.A) return;
  const float* an = g.anchors + ((int64_t)s * g.A + a) * 7;
  Box4 ab = aligned_bev(an);
  float a_max = -1.f;
  int a_arg = 0;
  bool forced = false, has_gt = false;
  for (int j = 0; j < g.M; ++j) {
    float v = -1.f;
    if (gt_valid(g, nkeep, b, j, s)) {
      has_gt = true;
      v = iou_normal(ab, aligned_bev(g.gt + ((int64_t)b * g.M + j) * 8));
      float gm = __int_as_float(gmax[((int64_t)b * g.n_sets + s) * g.M + j]);
      if (gm > 0.f && v == gm) forced = true;
    }
    if (v > a_max) {   // strict: first maximum wins, like torch.max / argmax
      a_max = v;
      a_arg = j;
    }
  }
  const float* gb = g.gt + ((int64_t)b * g.M + a_arg) * 8;
  int label = -1;
  if (a_max < g.unmatched[s] || !has_gt) label = 0;
  if ((forced || a_max >= g.matched[s]) && has_gt) label = (int)gb[7];
  // output position: (location, set, within-location) — the head's (y, x, class, rot) order
  const int64_t loc = a / g.per_loc, within = a % g.per_loc;
  const int64_t o = (int64_t)b * g.A * g.n_sets + (loc * g.n_sets + s) * g.per_loc + within;
  labels[o] = label;
  const bool pos = label > 0;
  weights[o] = pos ? 1.f : 0.f;
  float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pos) {   // ResidualCoder.encode_torch (sizes clamped at 1e-5)
    float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
    float dxg = fmaxf(gb[3], 1e-5f), dyg = fmaxf(gb[4], 1e-5f), dzg = fmaxf(gb[5], 1e-5f);
    float diag = sqrtf(dxa * dxa + dya * dya);
    t[0] = __fdiv_rn(gb[0] - an[0], diag);
    t[1] = __fdiv_rn(gb[1] - an[1], diag);
    t[2] = __fdiv_rn(gb[2] - an[2], dza);
    t[3] = logf(__fdiv_rn(dxg, dxa));
    t[4] = logf(__fdiv_rn(dyg, dya));
    t[5] = logf(__fdiv_rn(dzg, dza));
    t[6] = gb[6] - an[6];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) targets[o * 7 + k] = t[k];
}

}  // namespace

extern "C" size_t spx_assign_targets_ws_bytes(int batch, int n_sets, int max_gt) {
  return spx_align((size_t)batch * n_sets * max_gt * 4) + spx_align((size_t)batch * 4);
}

extern "C" int spx_assign_targets(const float* anchors, int n_sets, int64_t anchors_per_set, int per_location,
                                  const float* gt_boxes, int batch, int max_gt, const int32_t* d_set_class,
                                  int n_classes, const float* d_matched, const float* d_unmatched, int32_t* labels,
                                  float* targets, float* weights, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!anchors || !gt_boxes || !d_set_class || !d_matched || !d_unmatched || !labels || !targets || !weights ||
      n_sets <= 0 || anchors_per_set <= 0 || per_location <= 0 || anchors_per_set % per_location != 0 || batch <= 0 ||
      max_gt <= 0 || n_classes <= 0)
    return SPX_ERR_INVALID_ARG;
  if (max_gt > kMaxGt) return SPX_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spx_assign_targets_ws_bytes(batch, n_sets, max_gt)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  int* gmax = reinterpret_cast<int*>(ws);
  int32_t* nkeep = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + spx_align((size_t)batch * n_sets * max_gt * 4));
  AssignArgs g;
  g.anchors = anchors;
  g.gt = gt_boxes;
  g.set_cls = d_set_class;
  g.matched = d_matched;
  g.unmatched = d_unmatched;
  g.n_sets = n_sets;
  g.n_classes = n_classes;
  g.per_loc = per_location;
  g.B = batch;
  g.M = max_gt;
  g.A = anchors_per_set;
  spx_fill_async(gmax, 0, (size_t)batch * n_sets * max_gt * 4, s);
  hipLaunchKernelGGL(k_gt_keep, dim3((batch + 63) / 64), dim3(64), 0, s, g, nkeep);
  dim3 grid((unsigned)((anchors_per_set + 255) / 256), n_sets, batch);
  hipLaunchKernelGGL(k_gt_max, grid, dim3(256), 0, s, g, nkeep, gmax);
  hipLaunchKernelGGL(k_assign, grid, dim3(256), 0, s, g, nkeep, gmax, labels, targets, weights);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
