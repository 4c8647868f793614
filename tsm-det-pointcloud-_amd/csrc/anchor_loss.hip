// anchor_loss.hip — the anchor head's three losses AND their gradients in one pass.
//
// Restates AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss
// (reference pcdet/models/dense_heads/anchor_head_template.py:101-224) with
// SigmoidFocalClassificationLoss(alpha 0.25, gamma 2), WeightedSmoothL1Loss(beta 1/9, code weights 1) after
// add_sin_difference, and WeightedCrossEntropyLoss on the direction bins (pcdet/utils/loss_utils.py:9-77,140-209,
// 310-338; direction target anchor_head_template.py:151-165).  The reference spends ~100 elementwise / reduction
// launches forward and ~150 backward on [B, 211200, *] tensors; here one thread per anchor computes its three loss terms
// and d(loss)/d(prediction) at once, block sums go to a partial array and a last kernel adds them in fixed order (no
// float atomics: bitwise reproducible).  Normalisers are the per-frame positive counts, as in the reference.
#include "spx_common.h"

namespace {

constexpr int kMaxC = 8;
constexpr float kTwoPi = 6.28318530717958647692f;

struct LossArgs {
  const float* cls;      // [B][A][NC] logits
  const float* box;      // [B][A][7]
  const float* dir;      // [B][A][NB] or null
  const int32_t* labels; // [B][A]
  const float* tgt;      // [B][A][7]
  const float* anchors;  // [A][7]
  int B, NC, NB;
  int64_t A;
  float dir_offset, cls_w, loc_w, dir_w, beta, alpha;
};

__global__ __launch_bounds__(1024) void k_count_pos(const int32_t* __restrict__ labels, int64_t A, int32_t* __restrict__ npos) {
  __shared__ int s[16];
  const int b = blockIdx.x;
  int c = 0;
  const int32_t* row = labels + (int64_t)b * A;
  if ((A & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    // 16-byte loads, four in flight per thread: one block per frame walks 211 200 labels, and with one dependent 4-byte
    // load per trip this kernel took 34 us
    const int4* row4 = reinterpret_cast<const int4*>(row);
    const int64_t A4 = A >> 2;
#pragma unroll 4
    for (int64_t a = threadIdx.x; a < A4; a += 1024) {
      const int4 v = row4[a];
      c += (v.x > 0) + (v.y > 0) + (v.z > 0) + (v.w > 0);
    }
  } else {
    for (int64_t a = threadIdx.x; a < A; a += 1024) c += row[a] > 0 ? 1 : 0;
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < 16; ++i) t += s[i];
    npos[b] = t;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) t = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(256) void k_anchor_loss(LossArgs g, const int32_t* __restrict__ npos, float* __restrict__ partial,
                                                     float* __restrict__ dcls, float* __restrict__ dbox,
                                                     float* __restrict__ ddir) {
  __shared__ float sm[4];
  const int b = blockIdx.y;
  const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float l_cls = 0.f, l_loc = 0.f, l_dir = 0.f;
  if (a < g.A) {
    const int64_t o = (int64_t)b * g.A + a;
    const int label = g.labels[o];
    const float norm = 1.0f / fmaxf((float)npos[b], 1.0f);
    const float invB = 1.0f / (float)g.B;
    // ---- classification: sigmoid focal loss, weight (label >= 0) / max(#pos, 1)
    const float wc = label >= 0 ? norm : 0.f;
    for (int c = 0; c < g.NC; ++c) {
      const float x = g.cls[o * g.NC + c];
      const float t = (label == c + 1) ? 1.f : 0.f;
      const float p = 1.f / (1.f + expf(-x));
      const float aw = t * g.alpha + (1.f - t) * (1.f - g.alpha);
      const float pt = t * (1.f - p) + (1.f - t) * p;
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
      l_cls += aw * pt * pt * bce * wc;
      const float dpt = (1.f - 2.f * t) * p * (1.f - p);
      dcls[o * g.NC + c] = aw * (2.f * pt * dpt * bce + pt * pt * (p - t)) * wc * invB * g.cls_w;
    }
    // ---- box regression: smooth L1 on the sin-difference code, weight (label > 0) / max(#pos, 1)
    const float wr = label > 0 ? norm : 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float pr = g.box[o * 7 + k];
      float tg = g.tgt[o * 7 + k];
      float diff, dd = 1.f;
      if (k == 6) {
        if (tg != tg) tg = pr;            // NaN targets are ignored (loss_utils.py:190)
        diff = sinf(pr) * cosf(tg) - cosf(pr) * sinf(tg);
        dd = cosf(pr) * cosf(tg) + sinf(pr) * sinf(tg);
      } else {
        diff = (tg != tg) ? 0.f : pr - tg;
      }
      const float nrm = fabsf(diff);
      float l, dl;
      if (g.beta < 1e-5f) {
        l = nrm;
        dl = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      } else if (nrm < g.beta) {
        l = 0.5f * nrm * nrm / g.beta;
        dl = diff / g.beta;
      } else {
        l = nrm - 0.5f * g.beta;
        dl = diff > 0.f ? 1.f : -1.f;
      }
      l_loc += l * wr;
      dbox[o * 7 + k] = dl * dd * wr * invB * g.loc_w;
    }
    // ---- direction bins: cross entropy, weight (label > 0) / max(#pos, 1)
    if (g.dir != nullptr) {
      const float rot_gt = g.tgt[o * 7 + 6] + g.anchors[a * 7 + 6];
      const float v = rot_gt - g.dir_offset;
      const float off = v - floorf(v / kTwoPi + 0.f) * kTwoPi;
      int bin = (int)floorf(off / (kTwoPi / (float)g.NB));
      bin = bin < 0 ? 0 : (bin > g.NB - 1 ? g.NB - 1 : bin);
      float lg[kMaxC], mx = -3.4e38f;
      for (int c = 0; c < g.NB; ++c) {
        lg[c] = g.dir[o * g.NB + c];
        mx = fmaxf(mx, lg[c]);
      }
      float se = 0.f;
      for (int c = 0; c < g.NB; ++c) se += expf(lg[c] - mx);
      const float lse = mx + logf(se);
      l_dir += (lse - lg[bin]) * wr;
      for (int c = 0; c < g.NB; ++c)
        ddir[o * g.NB + c] = (expf(lg[c] - lse) - (c == bin ? 1.f : 0.f)) * wr * invB * g.dir_w;
    }
  }
  const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  float s0 = block_sum(l_cls, sm), s1 = block_sum(l_loc, sm), s2 = block_sum(l_dir, sm);
  if (threadIdx.x == 0) {
    partial[blk * 3 + 0] = s0;
    partial[blk * 3 + 1] = s1;
    partial[blk * 3 + 2] = s2;
  }
}

__global__ __launch_bounds__(256) void k_loss_final(const float* __restrict__ partial, int64_t nblk, LossArgs g,
                                                    float* __restrict__ out) {
  __shared__ float sm[4];
  float s[3] = {0.f, 0.f, 0.f};
  for (int64_t i = threadIdx.x; i < nblk; i += 256)
    for (int c = 0; c < 3; ++c) s[c] += partial[i * 3 + c];
  const float scale[3] = {g.cls_w / (float)g.B, g.loc_w / (float)g.B, g.dir_w / (float)g.B};
  for (int c = 0; c < 3; ++c) {
    float t = block_sum(s[c], sm);
    if (threadIdx.x == 0) out[c] = t * scale[c];
  }
}

}  // namespace

extern "C" size_t spx_anchor_loss_ws_bytes(int batch, int64_t anchors) {
  int64_t nblk = (int64_t)batch * ((anchors + 255) / 256);
  return spx_align((size_t)nblk * 3 * 4) + spx_align((size_t)batch * 4);
}

extern "C" int spx_anchor_loss(const float* cls_preds, const float* box_preds, const float* dir_preds,
                               const int32_t* labels, const float* reg_targets, const float* anchors, int batch,
                               int64_t n_anchors, int num_class, int num_dir_bins, float dir_offset, float cls_weight,
                               float loc_weight, float dir_weight, float beta, float alpha, float* losses, float* dcls,
                               float* dbox, float* ddir, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!cls_preds || !box_preds || !labels || !reg_targets || !anchors || !losses || !dcls || !dbox || batch <= 0 ||
      n_anchors <= 0 || num_class <= 0 || num_class > kMaxC || (dir_preds && (num_dir_bins <= 0 || num_dir_bins > kMaxC || !ddir)))
    return SPX_ERR_INVALID_ARG;
  if (!ws || ws_bytes < spx_anchor_loss_ws_bytes(batch, n_anchors)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  unsigned nbx = (unsigned)((n_anchors + 255) / 256);
  int64_t nblk = (int64_t)batch * nbx;
  float* partial = reinterpret_cast<float*>(ws);
  int32_t* npos = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + spx_align((size_t)nblk * 3 * 4));
  LossArgs g;
  g.cls = cls_preds;
  g.box = box_preds;
  g.dir = dir_preds;
  g.labels = labels;
  g.tgt = reg_targets;
  g.anchors = anchors;
  g.B = batch;
  g.NC = num_class;
  g.NB = dir_preds ? num_dir_bins : 0;
  g.A = n_anchors;
  g.dir_offset = dir_offset;
  g.cls_w = cls_weight;
  g.loc_w = loc_weight;
  g.dir_w = dir_preds ? dir_weight : 0.f;
  g.beta = beta;
  g.alpha = alpha;
  hipLaunchKernelGGL(k_count_pos, dim3(batch), dim3(1024), 0, s, labels, n_anchors, npos);
  hipLaunchKernelGGL(k_anchor_loss, dim3(nbx, batch), dim3(256), 0, s, g, npos, partial, dcls, dbox, ddir);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, s, partial, nblk, g, losses);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
