// nms.hip — rotated-BEV IoU and NMS.
//
// IoU follows the reference's algorithm, quirks included, because NMS decisions (an integer result) depend on them:
// intersection polygon = {edge/edge crossings with STRICT straddle tests} U {corners of one box inside the other with a
// 1e-2 margin}, vertices sorted by angle around their centroid, shoelace area
// (reference pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:30-235, CPU twin src/iou3d_cpu.cpp:30-252).
// NMS: 64x64 suppression bitmask tiles as iou3d_nms_kernel.cu:267-325, but only the upper triangle, and the greedy
// reduction that the reference does on the HOST after a synchronous cudaMemcpy of the whole mask
// (src/iou3d_nms.cpp:90-135) runs on the device: one wave resolves 64 boxes at a time from the diagonal tile held in
// registers, then ORs the kept rows into the running removed-mask; keep indices and their count stay on the device.
#include "spx_common.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-2f;

struct P2 {
  float x, y;
};

__device__ __forceinline__ float cross3(const P2& p1, const P2& p2, const P2& p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool bbox_overlap(const P2& p1, const P2& p2, const P2& q1, const P2& q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool in_box(const float* box, const P2& p) {
  float ca = cosf(-box[6]), sa = sinf(-box[6]);
  float rx = (p.x - box[0]) * ca + (p.y - box[1]) * (-sa);
  float ry = (p.x - box[0]) * sa + (p.y - box[1]) * ca;
  return fabsf(rx) < box[3] / 2 + kMargin && fabsf(ry) < box[4] / 2 + kMargin;
}

// segment p0-p1 x segment q0-q1 (strict straddling), intersection point in *ans
__device__ __forceinline__ bool seg_cross(const P2& p1, const P2& p0, const P2& q1, const P2& q0, P2* ans) {
  if (!bbox_overlap(p0, p1, q0, q1)) return false;
  float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ void corners_of(const float* b, P2* c) {
  float hx = b[3] / 2, hy = b[4] / 2, ca = cosf(b[6]), sa = sinf(b[6]);
  const float ox[4] = {-hx, hx, hx, -hx}, oy[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // rotate (center + offset) around the center
    float px = b[0] + ox[k], py = b[1] + oy[k];
    c[k].x = (px - b[0]) * ca + (py - b[1]) * (-sa) + b[0];
    c[k].y = (px - b[0]) * sa + (py - b[1]) * ca + b[1];
  }
  c[4] = c[0];
}

__device__ float overlap_area(const float* a, const float* b) {
  P2 ca[5], cb[5], pts[16];
  corners_of(a, ca);
  corners_of(b, cb);
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], &x)) {
        pts[cnt++] = x;
        sx += x.x;
        sy += x.y;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box(a, cb[k])) {
      sx += cb[k].x;
      sy += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (in_box(b, ca[k])) {
      sx += ca[k].x;
      sy += ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  if (cnt == 0) return 0.f;   // (the reference divides by zero here and sums an empty polygon: area 0)
  float cx = sx / cnt, cy = sy / cnt;
  float ang[16];
  for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - cy, pts[i].x - cx);
  // bubble sort ascending by angle, exactly the reference's pass structure (stable w.r.t. ties)
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        float t = ang[i];
        ang[i] = ang[i + 1];
        ang[i + 1] = t;
        P2 p = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = p;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float so = overlap_area(a, b);
  return so / fmaxf(sa + sb - so, kEps);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  float inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kEps);
}

__global__ void k_iou_bev(const float* __restrict__ a, int64_t n, const float* __restrict__ b, int64_t m,
                          float* __restrict__ out, int overlap_only) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * m) return;
  const float* pa = a + (t / m) * 7;
  const float* pb = b + (t % m) * 7;
  out[t] = overlap_only ? overlap_area(pa, pb) : iou_bev(pa, pb);
}

// grid (col_blocks, col_blocks), 64 threads; only tiles with col >= row are computed
template <bool NORMAL>
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, int64_t n, float thresh,
                                                 unsigned long long* __restrict__ mask, int col_blocks) {
  const int row_start = blockIdx.y, col_start = blockIdx.x;
  if (col_start < row_start) return;
  __shared__ float cb[64 * 7];
  const int row_size = (int)min((int64_t)64, n - (int64_t)row_start * 64);
  const int col_size = (int)min((int64_t)64, n - (int64_t)col_start * 64);
  if ((int)threadIdx.x < col_size)
    for (int j = 0; j < 7; ++j) cb[threadIdx.x * 7 + j] = boxes[((int64_t)col_start * 64 + threadIdx.x) * 7 + j];
  __syncthreads();
  if ((int)threadIdx.x < row_size) {
    const int64_t cur = (int64_t)row_start * 64 + threadIdx.x;
    float me[7];
    for (int j = 0; j < 7; ++j) me[j] = boxes[cur * 7 + j];
    unsigned long long t = 0;
    int start = row_start == col_start ? threadIdx.x + 1 : 0;
    for (int i = start; i < col_size; ++i) {
      float v = NORMAL ? iou_normal(me, cb + i * 7) : iou_bev(me, cb + i * 7);
      if (v > thresh) t |= 1ull << i;
    }
    mask[cur * col_blocks + col_start] = t;
  }
}

// one wave: greedy reduction in score order, 64 boxes per step
__global__ __launch_bounds__(64) void k_nms_reduce(const unsigned long long* __restrict__ mask, int64_t n, int col_blocks,
                                                   int64_t* __restrict__ keep, int64_t* __restrict__ d_num_keep) {
  extern __shared__ unsigned long long remv[];   // [col_blocks] running "removed" mask, lives in LDS
  const int lane = threadIdx.x;
  for (int j = lane; j < col_blocks; j += 64) remv[j] = 0ull;
  __syncthreads();
  int64_t cnt = 0;
  for (int c = 0; c < col_blocks; ++c) {
    const int size = (int)min((int64_t)64, n - (int64_t)c * 64);
    // diagonal tile: lane i holds which later boxes of this chunk box i suppresses
    unsigned long long diag = lane < size ? mask[((int64_t)c * 64 + lane) * col_blocks + c] : 0ull;
    unsigned long long alive = ~remv[c];
    if (size < 64) alive &= (1ull << size) - 1ull;
    unsigned long long kept = 0ull;
    for (int i = 0; i < size; ++i) {
      unsigned long long di = __shfl(diag, i, 64);   // wave-uniform loop: every lane tracks the same alive/kept
      if ((alive >> i) & 1ull) {
        kept |= 1ull << i;
        alive &= ~di;
      }
    }
    // emit kept indices in order (lane i writes its own slot) and fold their rows into the removed mask
    if ((kept >> lane) & 1ull) keep[cnt + __popcll(kept & ((1ull << lane) - 1ull))] = (int64_t)c * 64 + lane;
    cnt += __popcll(kept);
    for (int j = c + 1 + lane; j < col_blocks; j += 64) {
      unsigned long long acc = remv[j];
      unsigned long long kk = kept;
      while (kk) {
        int i = __ffsll(kk) - 1;
        kk &= kk - 1;
        acc |= mask[((int64_t)c * 64 + i) * col_blocks + j];
      }
      remv[j] = acc;
    }
    __syncthreads();
  }
  if (lane == 0) *d_num_keep = cnt;
}

}  // namespace

extern "C" int spx_boxes_iou_bev(const float* boxes_a, int64_t n, const float* boxes_b, int64_t m, int overlap_only,
                                 float* out, spx_stream_t stream) {
  if ((!boxes_a && n > 0) || (!boxes_b && m > 0) || !out || n < 0 || m < 0) return SPX_ERR_INVALID_ARG;
  if (n * m == 0) return SPX_OK;
  if (n * m >= (int64_t(1) << 40)) return SPX_ERR_TOO_LARGE;
  hipLaunchKernelGGL(k_iou_bev, dim3((unsigned)((n * m + 255) / 256)), dim3(256), 0, spx_s(stream), boxes_a, n, boxes_b, m,
                     out, overlap_only);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" size_t spx_nms_ws_bytes(int64_t n) {
  int64_t cb = (n + 63) / 64;
  return spx_align((size_t)(n < 1 ? 1 : n) * cb * 8);
}

extern "C" int spx_nms_bev(const float* boxes, int64_t n, float thresh, int axis_aligned, int64_t* keep,
                           int64_t* d_num_keep, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if ((!boxes && n > 0) || !keep || !d_num_keep || n < 0) return SPX_ERR_INVALID_ARG;
  if (n > (int64_t(1) << 20)) return SPX_ERR_TOO_LARGE;   // removed-mask must fit LDS: 2^20 / 64 words = 128 KiB
  hipStream_t s = spx_s(stream);
  if (n == 0) {
    spx_fill_async(d_num_keep, 0, sizeof(int64_t), s);
    return SPX_OK;
  }
  if (!ws || ws_bytes < spx_nms_ws_bytes(n)) return SPX_ERR_WORKSPACE;
  int cb = (int)((n + 63) / 64);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws);
  if (axis_aligned)
    hipLaunchKernelGGL((k_nms_mask<true>), dim3(cb, cb), dim3(64), 0, s, boxes, n, thresh, mask, cb);
  else
    hipLaunchKernelGGL((k_nms_mask<false>), dim3(cb, cb), dim3(64), 0, s, boxes, n, thresh, mask, cb);
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(64), (size_t)cb * 8, s, mask, n, cb, keep, d_num_keep);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
