// wino_wgrad.hip — weight gradient of the dense 3x3 / stride 1 / pad 1 convolution (csrc/wino_conv2d.hip) in the Winograd
// F(2x2, 3x3) domain, on the exact-fp32 MFMA.
//
// With Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile, the filter gradient is
//     dg = G^T [ sum over tiles of (B^T d B) . (A dY A^T) ] G
// i.e. 16 independent [Cin x tiles] x [tiles x Cout] GEMMs (one per transform position) in place of the direct form's 9
// [Cin x pixels] x [pixels x Cout]: 16 multiplies per (tile, ci, co) instead of 36, same fp32 arithmetic.
// replaces: the weight half of cuDNN / MIOpen convolution_backward for Conv2d(c, c', 3, padding=1, bias=False), reference
// pcdet/models/backbones_2d/base_bev_backbone.py:38-49 (10 such layers in the KITTI model).
//
// Kernel 1 (k_wino_wgrad): one workgroup (512 threads, 8 waves) = one ROW i of the 4x4 transform positions x a 128 x 128
// block of (ci, co) x a contiguous range of tiles (the K dimension, split over workgroups so the launch fills the chip).
//   * wave (j, half): position (i, j); accumulators 4 ci-blocks x 2 co-blocks of 32x32 = 128 registers.  A fragment lane
//     m' holds channels 4m'..4m'+3 of ONE tile as one ds_read_b128: component e is MFMA row-block e, i.e. row-block e
//     is the channels = e (mod 4) — the permutation is undone when the partial result is stored.  (A wave's 64 output
//     channels are contiguous: lane n' holds the pair 2n', 2n'+1.)  So both
//     transformed operands sit in LDS in their natural [tile][channel] order: no transpose anywhere.
//   * per step of 8 tiles: every thread fetches 2 patch rows x 16 bytes for two (tile, patch column, channel quad) items
//     and 2 dY rows x 16 bytes for one (tile, dY column, channel quad) item (buffer loads: out-of-map = 0 = the zero
//     padding), one step ahead; row combination in registers, column pass across lanes with DPP, result to LDS (double
//     buffered, one barrier per step); 32 MFMAs per wave per step.
//   * the partial sums go to part[split][pos][Cin][Cout].
// Kernel 2 (k_wino_wgrad_final): sums the splits, applies G^T . G and writes dw with the weight tensor's own strides.
#include "spx_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStepTiles = 8;
constexpr int kBlk = 128;       // ci and co per workgroup
constexpr int kThreads = 512;
constexpr uint32_t kOob = 0xFFFFFFF0u;

struct WgradArgs {
  const float* x;       // [N, H, W] pixels, x_ld floats apart
  const float* dy;      // [N, H, W] pixels, dy_ld floats apart
  float* part;          // [n_split][16][Cin][Cout]
  int64_t x_ld, dy_ld;
  uint32_t x_bytes, dy_bytes;
  int32_t n, h, w, cin, cout;
  int32_t tiles_x, tiles_y;
  int64_t n_tiles;
  int32_t tiles_per_split;   // multiple of kStepTiles
  int32_t n_split;
};

constexpr int qp(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__global__ void __launch_bounds__(kThreads) k_wino_wgrad(WgradArgs a) {
  // Vs[buf 2][j 4][tile 8][128 ci] then Zs[buf 2][j 4][tile 8][128 co]: 2 x 32 KiB
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 4 * kStepTiles * kBlk];
  float* const vs = smem;
  float* const zs = smem + 2 * 4 * kStepTiles * kBlk;
  constexpr int kBufFloats = 4 * kStepTiles * kBlk;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  // blockIdx.x = (split low 3 bits) | (row i << 3) | (split high bits << 5): the four rows of one tile range run on the
  // same XCD (ids equal mod 8) and share its L2 for the pixels they all read
  const int bid = blockIdx.x;
  const int row = (bid >> 3) & 3;
  const int split = (bid & 7) | ((bid >> 5) << 3);
  const int cib = blockIdx.y, cob = blockIdx.z;

  const int64_t t_begin = (int64_t)split * a.tiles_per_split;
  int64_t t_end = t_begin + a.tiles_per_split;
  if (t_end > a.n_tiles) t_end = a.n_tiles;
  const int nsteps = t_begin < t_end ? (int)((t_end - t_begin + kStepTiles - 1) / kStepTiles) : 0;

  // row i of B^T d = d[ra] + sb * d[rb]; row i of A dY = a0 * dY[0] + a1 * dY[1]
  const int ra = (row == 0) ? 0 : (row == 2 ? 2 : 1);
  const int rb = (row == 0) ? 2 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
  const float sb = (row == 1) ? 1.0f : -1.0f;
  const float a0 = (row == 3) ? 0.0f : 1.0f;
  const float a1 = (row == 0) ? 0.0f : (row == 1 ? 1.0f : -1.0f);

  // the x resource starts ONE PIXEL BEFORE the map: patch column `col` of tile column tx is pixel 2tx - 1 + col, and with
  // that shift its offset is (scalar: pixel 2tx of the row) + (vector: col pixels) with both parts non-negative — buffer
  // offsets are unsigned and are not wrapped.  The pixel before the map is never read (those lanes carry the
  // out-of-range offset).
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) - a.x_ld, 0, (int)(a.x_bytes + (uint32_t)a.x_ld * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);

  // ---- loader roles
  // x: items (tile, patch column, channel quad) = 8 x 4 x 32, two per thread (tiles tlo and tlo + 4)
  const int x_col = lane & 3;
  const int x_qd = (lane >> 2) + 16 * (wave & 1);
  const int x_tlo = wave >> 1;
  // dy: items (tile, dY column, channel quad) = 8 x 2 x 32, one per thread
  const int z_c = lane & 1;
  const int z_qd = lane >> 1;
  const int z_tl = wave;
  const uint32_t x_chan = (uint32_t)((cib * kBlk + x_qd * 4) * 4);
  const uint32_t z_chan = (uint32_t)((cob * kBlk + z_qd * 4) * 4);
  const float sgn = (x_col == 1) ? 1.0f : -1.0f;

  // tile -> (frame, tile row, tile column) of the step's first tile, advanced by 8 per step in scalar registers; a thread's
  // tile is at most 7 further, so one wrap is enough (host: tiles_x >= 8)
  int btx, bty, bnn;
  {
    const int64_t tt = t_begin < a.n_tiles ? t_begin : 0;
    btx = (int)(tt % a.tiles_x);
    bty = (int)((tt / a.tiles_x) % a.tiles_y);
    bnn = (int)(tt / ((int64_t)a.tiles_x * a.tiles_y));
  }
  int64_t bt = t_begin;

  const uint32_t hw = (uint32_t)a.h * (uint32_t)a.w, x_ld4 = (uint32_t)a.x_ld * 4u, dy_ld4 = (uint32_t)a.dy_ld * 4u;
  // A wave's x items share their tile (x_tlo, x_tlo + 4) and its dy item its tile (z_tl): tile -> pixel arithmetic and the
  // row / tile validity are wave-uniform and run on the scalar unit; per lane there is only a constant byte offset (patch
  // column and channels) that goes in the buffer load's vector offset, the tile's base goes in its scalar offset, and the
  // lanes whose patch column falls off the left / right edge of the map get the out-of-range offset (= zero padding).
  const uint32_t xv_const = (uint32_t)x_col * x_ld4 + x_chan;     // + scalar: ((nn*H + py)*W + 2tx) * ld4, resource shifted
  const uint32_t zv_const = (uint32_t)z_c * dy_ld4 + z_chan;      // + scalar: ((nn*H + 2ty)*W + 2tx) * ld4
  f32x4 xa[2], xb[2], dz[2];
  auto issue_loads = [&]() {
    // the step starting at tile bt
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int tl = x_tlo + 4 * it;
      int tx = btx + tl, ty = bty, nn = bnn;
      if (tx >= a.tiles_x) {
        tx -= a.tiles_x;
        ty += 1;
        if (ty >= a.tiles_y) {
          ty = 0;
          nn += 1;
        }
      }
      const bool tv = bt + tl < t_end;
      const int pya = 2 * ty - 1 + ra, pyb = 2 * ty - 1 + rb;
      // scalar byte offset of pixel (row 0, 2tx) of the frame (in the shifted resource: of pixel 2tx - 1)
      const uint32_t sbase = ((uint32_t)nn * hw + (uint32_t)(2 * tx)) * x_ld4;
      const uint32_t rowb = (uint32_t)a.w * x_ld4;
      const bool oka = tv && pya >= 0 && pya < a.h, okb = tv && pyb >= 0 && pyb < a.h;
      // patch columns inside the map: 2tx - 1 + col in [0, W)
      const int lo = (tx == 0) ? 1 : 0, hi = a.w - (2 * tx - 1);      // valid columns: lo <= col < hi
      const bool okc = x_col >= lo && x_col < hi;
      const uint32_t va = (oka && okc) ? xv_const : kOob, vb = (okb && okc) ? xv_const : kOob;
      xa[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)va, (int)(sbase + (uint32_t)pya * rowb), 0));
      xb[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vb, (int)(sbase + (uint32_t)pyb * rowb), 0));
    }
    {
      int tx = btx + z_tl, ty = bty, nn = bnn;
      if (tx >= a.tiles_x) {
        tx -= a.tiles_x;
        ty += 1;
        if (ty >= a.tiles_y) {
          ty = 0;
          nn += 1;
        }
      }
      const bool tv = bt + z_tl < t_end;
      const int py = 2 * ty;
      const uint32_t rowb = (uint32_t)a.w * dy_ld4;
      const uint32_t sbase = (((uint32_t)nn * (uint32_t)a.h + (uint32_t)py) * (uint32_t)a.w + (uint32_t)(2 * tx)) * dy_ld4;
      const bool okc = 2 * tx + z_c < a.w;
      const uint32_t v0 = (tv && okc && row != 3) ? zv_const : kOob;                     // row 3 of A dY does not use dY[0]
      const uint32_t v1 = (tv && okc && row != 0 && py + 1 < a.h) ? zv_const : kOob;     // row 0 does not use dY[1]
      dz[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)v0, (int)sbase, 0));
      dz[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)v1, (int)(sbase + rowb), 0));
    }
    // advance the scalar tile cursor
    bt += kStepTiles;
    btx += kStepTiles;
    if (btx >= a.tiles_x) {
      btx -= a.tiles_x;
      bty += 1;
      if (bty >= a.tiles_y) {
        bty = 0;
        bnn += 1;
      }
    }
  };
  float* const v_dst = vs + (x_col * kStepTiles + x_tlo) * kBlk + x_qd * 4;          // + buf, + 4 tiles for item 1
  float* const z_dst = zs + ((2 * z_c) * kStepTiles + z_tl) * kBlk + z_qd * 4;       // + buf, + one j plane for the second
  // the transform of the loaded step into LDS buffer `buf`, in three pieces (the K loop puts one between each pair of MFMA
  // groups, where it issues in the shadow of the partner wave's MFMAs)
  auto xform_x = [&](int buf, int it) {
    const f32x4 t = xa[it] + sb * xb[it];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = dpp<qp(0, 1, 2, 1)>(t[e]) + sgn * dpp<qp(2, 2, 1, 3)>(t[e]);
    *reinterpret_cast<f32x4*>(v_dst + buf * kBufFloats + it * (4 * kBlk)) = v;
  };
  auto xform_dy = [&](int buf) {
    const f32x4 r = a0 * dz[0] + a1 * dz[1];
    f32x4 p, za, zb;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] = dpp<qp(1, 0, 3, 2)>(r[e]);
    // dY column 0 lane: Z[i][0] = r0, Z[i][1] = r0 + r1; column 1 lane: Z[i][2] = r0 - r1, Z[i][3] = -r1
    za = z_c ? (p - r) : r;
    zb = z_c ? (-r) : (r + p);
    *reinterpret_cast<f32x4*>(z_dst + buf * kBufFloats) = za;
    *reinterpret_cast<f32x4*>(z_dst + buf * kBufFloats + kStepTiles * kBlk) = zb;
  };
  auto transform_store = [&](int buf) {
    xform_x(buf, 0);
    xform_x(buf, 1);
    xform_dy(buf);
  };

  // ---- MFMA role: position (row, j), output-channel half
  const int j = wave & 3, half = wave >> 2;
  const float* const a_src = vs + (j * kStepTiles + lh) * kBlk + l31 * 4;
  const float* const b_src = zs + (j * kStepTiles + lh) * kBlk + half * 64 + l31 * 2;
  f32x16 acc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
  // the 32 MFMAs of step `buf`; after each group of 8: one piece of the NEXT step's transform (into the other buffer), after
  // the last the loads of the step after next
  auto mma = [&](int buf) {
    const float* pa = a_src + buf * kBufFloats;
    const float* pb = b_src + buf * kBufFloats;
    f32x4 av = *reinterpret_cast<const f32x4*>(pa);
    f32x2 bv = *reinterpret_cast<const f32x2*>(pb);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 an = av;
      f32x2 bn = bv;
      if (q < 3) {
        an = *reinterpret_cast<const f32x4*>(pa + (q + 1) * 2 * kBlk);
        bn = *reinterpret_cast<const f32x2*>(pb + (q + 1) * 2 * kBlk);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[f], acc[e][f], 0, 0, 0);
      if (q == 0) xform_x(buf ^ 1, 0);
      if (q == 1) xform_x(buf ^ 1, 1);
      if (q == 2) xform_dy(buf ^ 1);
      if (q == 3) issue_loads();
      __builtin_amdgcn_sched_barrier(0);
      av = an;
      bv = bn;
    }
  };

  if (nsteps > 0) {
    issue_loads();
    transform_store(0);
    issue_loads();       // step 1 (past the end: every offset out of range, zeros)
    __syncthreads();
    // one barrier per step
    for (int s = 0; s < nsteps; ++s) {
      mma(s & 1);
      __syncthreads();
    }
  }

  // ---- partial result: acc[e][f][r] is (ci = 4 m' + e, co = 64 half + 2 n' + f), m' = row of register r, n' = lane & 31: a
  // wave's store instruction covers 256 contiguous bytes of a row (whole 128-byte lines: with the two halves interleaved at
  // 8 bytes, as first written, the PMC write size was twice the 64 MB of partials)
  float* const out = a.part + (((int64_t)split * 16 + row * 4 + j) * a.cin + cib * kBlk) * a.cout + cob * kBlk + half * 64 + l31 * 2;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mrow = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const f32x2 v = {acc[e][0][r], acc[e][1][r]};
      *reinterpret_cast<f32x2*>(out + (int64_t)(4 * mrow + e) * a.cout) = v;
    }
}

// dw[co][ci][a][b] = (G^T [sum over splits of part[.][pos][ci][co]] G)[a][b]; blockDim (64 co, 4 split groups)
__global__ void __launch_bounds__(256) k_wino_wgrad_final(const float* __restrict__ part, int n_split, int cin, int cout,
                                                          float* __restrict__ dw, int64_t s_o, int64_t s_i, int64_t s_a,
                                                          int64_t s_b) {
  __shared__ float red[3][16][64];
  const int co = blockIdx.x * 64 + threadIdx.x;
  const int ci = blockIdx.y;
  const int g = threadIdx.y;
  float u[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) u[p] = 0.f;
  const int64_t plane = (int64_t)cin * cout;
  for (int s = g; s < n_split; s += 4) {
    const float* src = part + (int64_t)s * 16 * plane + (int64_t)ci * cout + co;
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] += src[p * plane];
  }
  if (g > 0) {
#pragma unroll
    for (int p = 0; p < 16; ++p) red[g - 1][p][threadIdx.x] = u[p];
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] += red[0][p][threadIdx.x] + red[1][p][threadIdx.x] + red[2][p][threadIdx.x];
    float t[3][4];   // G^T U
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      t[0][jj] = u[jj] + 0.5f * (u[4 + jj] + u[8 + jj]);
      t[1][jj] = 0.5f * (u[4 + jj] - u[8 + jj]);
      t[2][jj] = 0.5f * (u[4 + jj] + u[8 + jj]) + u[12 + jj];
    }
    float* d = dw + (int64_t)co * s_o + (int64_t)ci * s_i;
#pragma unroll
    for (int aa = 0; aa < 3; ++aa) {
      d[aa * s_a + 0 * s_b] = t[aa][0] + 0.5f * (t[aa][1] + t[aa][2]);
      d[aa * s_a + 1 * s_b] = 0.5f * (t[aa][1] - t[aa][2]);
      d[aa * s_a + 2 * s_b] = 0.5f * (t[aa][1] + t[aa][2]) + t[aa][3];
    }
  }
}

static int wgrad_splits(int cin, int cout) {
  int s = 64 / ((cin / kBlk) * (cout / kBlk));
  s &= ~7;
  return s < 8 ? 8 : s;
}

}  // namespace

extern "C" size_t spx_wino_wgrad_ws_bytes(int32_t cin, int32_t cout) {
  if (cin <= 0 || cout <= 0 || cin % kBlk || cout % kBlk) return 0;
  return (size_t)wgrad_splits(cin, cout) * 16 * (size_t)cin * cout * sizeof(float);
}

extern "C" int spx_conv2d_wino_wgrad(const float* x, int64_t x_ld, const float* dy, int64_t dy_ld, int32_t n, int32_t h,
                                     int32_t w, int32_t cin, int32_t cout, float* dw, int64_t s_o, int64_t s_i, int64_t s_a,
                                     int64_t s_b, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0) return SPX_ERR_INVALID_ARG;
  if (cin % kBlk || cout % kBlk || x_ld < cin || dy_ld < cout || (x_ld & 3) || (dy_ld & 3)) return SPX_ERR_INVALID_ARG;
  const int tiles_x = (w + 1) / 2, tiles_y = (h + 1) / 2;
  if (tiles_x < kStepTiles) return SPX_ERR_UNSUPPORTED;
  const int64_t x_bytes = ((int64_t)n * h * w - 1) * x_ld * 4 + (int64_t)cin * 4;
  const int64_t dy_bytes = ((int64_t)n * h * w - 1) * dy_ld * 4 + (int64_t)cout * 4;
  if (x_bytes + x_ld * 4 >= 0xFFFFFFF0ll || dy_bytes >= 0xFFFFFFF0ll) return SPX_ERR_TOO_LARGE;
  const int ns = wgrad_splits(cin, cout);
  if (ws == nullptr || ws_bytes < spx_wino_wgrad_ws_bytes(cin, cout)) return SPX_ERR_WORKSPACE;
  WgradArgs a;
  a.x = x; a.dy = dy; a.part = static_cast<float*>(ws);
  a.x_ld = x_ld; a.dy_ld = dy_ld;
  a.x_bytes = (uint32_t)x_bytes; a.dy_bytes = (uint32_t)dy_bytes;
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
  a.tiles_x = tiles_x; a.tiles_y = tiles_y;
  a.n_tiles = (int64_t)n * tiles_x * tiles_y;
  const int64_t per = (a.n_tiles + ns - 1) / ns;
  a.tiles_per_split = (int32_t)((per + kStepTiles - 1) / kStepTiles * kStepTiles);
  a.n_split = ns;
  hipLaunchKernelGGL(k_wino_wgrad, dim3((unsigned)(4 * ns), (unsigned)(cin / kBlk), (unsigned)(cout / kBlk)), dim3(kThreads), 0,
                     spx_s(stream), a);
  SPX_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_wino_wgrad_final, dim3((unsigned)(cout / 64), (unsigned)cin), dim3(64, 4), 0, spx_s(stream),
                     static_cast<const float*>(ws), ns, cin, cout, dw, s_o, s_i, s_a, s_b);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
