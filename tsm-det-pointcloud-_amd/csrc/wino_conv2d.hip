// wino_conv2d.hip — dense 3x3 / stride 1 / pad 1 convolution over channels-last BEV maps, Winograd F(2x2, 3x3) on the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), transforms fused into the one kernel.
//
// Replaces, for the 3x3 stride-1 layers of BaseBEVBackbone (reference pcdet/models/backbones_2d/base_bev_backbone.py:38-49:
// Conv2d(c, c, kernel_size=3, padding=1, bias=False) x LAYER_NUMS per block) the vendor implicit-GEMM kernels that take
// 2/3 of the training step: the direct form needs 9 multiplies per (pixel, ci, co), F(2x2, 3x3) needs 4 — per 2x2 output
// tile, 16 element-wise products in the transformed domain, each of them a [tiles x Cin] x [Cin x Cout] GEMM:
//     Y = A^T [ (G g G^T) . (B^T d B) ] A          (Lavin & Gray 2015, correlation form — what Conv2d computes)
// The data gradient is the same kernel over dY with the taps rotated by 180 degrees and ci/co swapped (spx_wino_weight's
// `flip`).  Arithmetic stays fp32 end to end (the transforms are +/- and *0.5, the products run on the f32 MFMA); the
// difference to the direct sum is reassociation, ~1e-6 relative (tests/test_gpu_wino.py states the bar).
//
// Work decomposition (one workgroup = 512 threads = 8 waves, 2 per SIMD):
//   * 32 consecutive 2x2 tiles (row-major over frame, tile row, tile column) x 128 output channels;
//   * wave (w, half): the 4 transform positions of row i = w (p = 4w .. 4w+3), output channels half*64 .. +63:
//     accumulators 4 positions x 2 column blocks x 16 = 128 registers; A = transformed input (tiles are the MFMA rows),
//     B = transformed weights (output channels are the MFMA columns);
//   * K loop in chunks of 8 input channels: threads 0..255 load one column (4 pixels x 4 channels) of one tile's 4x4 input
//     patch straight from global memory (the 4x overlap of neighbouring patches is served by L1/L2), transform it (column
//     pass in registers, row pass across the 4 lanes of a quad with DPP) and write the 16 positions to LDS, double
//     buffered; every wave reads its A fragments back with ds_read_b128.  B fragments never touch LDS: a position's
//     weights are used by exactly one wave, which reads them from the (L2-resident, pre-transformed, fragment-ordered)
//     weight image with one coalesced 1 KiB load per (position, column block);
//   * the k index of an MFMA step is (lane half h, sub-step s) -> channel 4h+s for A and B alike, so one 16-byte load
//     feeds four MFMA steps;
//   * output transform: each wave reduces its row over j in registers ((M A)[i][c], c = 0,1), the four rows meet in LDS
//     (64 KiB per c) and all 512 threads combine them, apply the optional per-channel scale / shift / ReLU and store
//     channels-last rows (512 contiguous bytes per pixel).
#include "spx_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTiles = 32;    // 2x2 output tiles per workgroup = MFMA rows
constexpr int kKc = 8;        // input channels per K chunk
constexpr int kSc = 16;       // input channels per super-chunk (2 chunks): one 64-byte half line of a pixel
constexpr int kCoWg = 128;    // output channels per workgroup
constexpr int kThreads = 512;
constexpr int kPosFloats = kCoWg * kKc;   // floats of one (chunk, position) of the weight image
constexpr int kPlane = kTiles * kKc + 8;  // floats of one position's [tile][k] plane in LDS: + 32 bytes, so that the four
                                          // patch-column lanes of a quad (four planes) write to different banks

struct WinoArgs {
  const float* x;      // [N, H, W] pixels, x_ld floats apart, Cin used
  const float* u;      // transformed weights, see k_wino_weight
  float* y;            // [N, H, W] pixels, y_ld floats apart
  const float* scale;  // [Cout] or null
  const float* shift;  // [Cout] or null
  float* stats;        // [n_blocks][2][Cout] per-tile-block sums of y and y*y (BatchNorm statistics), or null
  int64_t x_ld, y_ld;
  uint32_t x_bytes, u_bytes;   // extents of x and u for the buffer resources
  int32_t n, h, w, cin, cout;
  int32_t tiles_x, tiles_y;
  int64_t n_tiles;
  int32_t relu;
  int32_t n_blocks;    // gridDim.x
};

// quad_perm[a,b,c,d] DPP control
constexpr int qp(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

__device__ __forceinline__ float dpp_p(float v) {   // lanes (0,1,2,3) of a quad read lanes (0,1,2,1)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), qp(0, 1, 2, 1), 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_q(float v) {   // lanes (0,1,2,3) of a quad read lanes (2,2,1,3)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), qp(2, 2, 1, 3), 0xF, 0xF, true));
}

// Which input channel sits in k-slot `slot` (0..7) of chunk `chunk`.  A loader lane fetches 16 bytes (4 channels) per
// pixel and super-chunk; chunk kk (0, 1) of the super-chunk takes components 2kk, 2kk+1 of every such fetch, so the 8
// k-slots of a chunk are channels 16*S + 4*g + 2*kk + e  (g = slot >> 1: lane group, e = slot & 1) — see the loader below.
__host__ __device__ constexpr int wino_channel(int chunk, int slot) {
  return (chunk >> 1) * kSc + (slot >> 1) * 4 + (chunk & 1) * 2 + (slot & 1);
}

// Transformed weight image u[cb][chunk][pos 16][nb 4][h 2][col 32][s 4]: cb = output-channel block of 128, nb = 32-column
// block, k-slot = 4h + s — the 1 KiB of one (chunk, pos, nb) is exactly one wave's B-fragment load in lane order.
// U_pos = (G g G^T)[pos] with g[a][b] = w[(co, ci, a, b)] (flip = 0) or w[(ci', co', 2-a, 2-b)] read as the filter of the
// data gradient (flip = 1: the kernel's "input channels" are the layer's output channels).  Strides are in floats, so
// OIHW-contiguous and channels_last weights both work.
__global__ void k_wino_weight(const float* __restrict__ w, int64_t s_o, int64_t s_i, int64_t s_a, int64_t s_b, int cin,
                              int cout, int flip, float* __restrict__ u) {
  // cin / cout are the KERNEL's input / output channel counts (already swapped by the host for flip = 1)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)cin * cout;
  if (idx >= total) return;
  const int nchunk = cin / kKc;
  const int sidx = (int)(idx & 3);
  const int col = (int)((idx >> 2) & 31);
  const int hh = (int)((idx >> 7) & 1);
  const int nb = (int)((idx >> 8) & 3);
  const int chunk = (int)((idx >> 10) % nchunk);
  const int cb = (int)((idx >> 10) / nchunk);
  const int ci = wino_channel(chunk, 4 * hh + sidx), co = cb * kCoWg + nb * 32 + col;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      g[a][b] = flip ? w[(int64_t)ci * s_o + (int64_t)co * s_i + (2 - a) * s_a + (2 - b) * s_b]
                     : w[(int64_t)co * s_o + (int64_t)ci * s_i + a * s_a + b * s_b];
  float t[4][3];   // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  float* dst = u + ((int64_t)(cb * nchunk + chunk) * 16) * kPosFloats + (idx & 1023);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(int64_t)(i * 4 + 0) * kPosFloats] = t[i][0];
    dst[(int64_t)(i * 4 + 1) * kPosFloats] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    dst[(int64_t)(i * 4 + 2) * kPosFloats] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    dst[(int64_t)(i * 4 + 3) * kPosFloats] = t[i][2];
  }
}

// NB = 32-column blocks per wave: 2 -> a workgroup covers 128 output channels, 256 registers, one workgroup per CU;
// 1 -> 64 output channels, 128 registers, TWO workgroups per CU (the epilogue / prologue of one runs under the K loop of the
// other, and a launch whose workgroup count is a poor multiple of 256 loses less in its last round).
template <int NB>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 4 : 2, NB == 1 ? 4 : 2)))
k_wino_conv(WinoArgs a) {
  constexpr int kCoW = 64 * NB;      // output channels of this workgroup
  // V[buf 2][pos 16][tile 32][k 8] during the K loop (32 KiB); R[row i 4][tile 32][co] in the output transform
  constexpr int kSmemFloats = 4 * kTiles * kCoW > 2 * 16 * kPlane ? 4 * kTiles * kCoW : 2 * 16 * kPlane;
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wrow = wave & 3;    // transform row i owned by this wave
  const int half = wave >> 2;   // which half of the workgroup's output channels
  const int l31 = lane & 31, lh = lane >> 5;

  // Workgroup id -> (tile block, output-channel block).  Consecutive ids go round-robin to the 8 XCDs, each with its own L2.
  // An XCD gets a CONTIGUOUS range of tile blocks — vertically adjacent tiles share two of their four patch rows, and a
  // tile row is 2-3 blocks long, so the sharers run on the same L2 within a few workgroup slots — and, for each tile
  // block, all its output-channel blocks back to back (they read the same pixels).  Round-robin over tile blocks fetched
  // every map 2.6 (128 channels) to 6 times (256 channels) into the L2s.
  const int ncb = a.cout / kCoW;
  const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int per = (a.n_blocks + 7) >> 3;         // tile blocks per XCD (the grid is padded to 8 * per * ncb)
  const int cbw = rest % ncb;                    // output-channel block of width kCoW
  const int bid = xcd * per + rest / ncb;        // tile block
  if (bid >= a.n_blocks) return;
  const int64_t tile0 = (int64_t)bid * kTiles;
  const int nbg = (cbw * 2 + half) * NB;         // this wave's first 32-column block, counted over all of Cout
  const int nchunk = a.cin / kKc;
  const int nsuper = a.cin / kSc;

  // ---- loader role (every thread): one patch column (4 pixels) of one tile, lane group g of 4.  Per super-chunk (16
  // channels = a 64-byte half line of each pixel) a thread fetches 16 bytes (channels 4g..4g+3) of each of its 4 pixels:
  // the four lane groups of a pixel cover the 64 contiguous bytes in one instruction.  Components 2kk, 2kk+1 of the four
  // fetched vectors are the thread's share of chunk kk of the super-chunk (wino_channel), so one round of loads feeds two
  // K chunks and every fetched byte is used.
  const int ld_col = tid & 3;           // patch column j — the 4 lanes of a DPP quad
  const int ld_g = (tid >> 2) & 3;      // lane group: channels 4g..4g+3 of each 64-byte half
  const int ld_tile = tid >> 4;         // tile inside the workgroup
  // Both operands come in through buffer loads: a scalar resource + a 32-bit per-lane byte offset (+ a scalar offset for
  // the weights) instead of 64-bit per-lane pointers, and an out-of-range offset reads as 0 — which IS the zero padding:
  // pixels outside the map (and the tiles past the end) get offset 0xFFFFFFF0, no mask, no branch, same load count on
  // every lane.  (host: both buffers < 4 GiB)
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_u =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, (int)a.u_bytes, 0x00020000);
  uint32_t xoff[4];   // byte offset of the thread's 4 pixels (+ its lane group's 16 bytes)
  {
    const int64_t t = tile0 + ld_tile;
    const bool tv = t < a.n_tiles;
    const int64_t tt = tv ? t : 0;
    const int tx = (int)(tt % a.tiles_x);
    const int ty = (int)((tt / a.tiles_x) % a.tiles_y);
    const int nn = (int)(tt / ((int64_t)a.tiles_x * a.tiles_y));
    const int px = 2 * tx - 1 + ld_col;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int py = 2 * ty - 1 + i;
      const bool ok = tv && px >= 0 && px < a.w && py >= 0 && py < a.h;
      const int64_t pix = ((int64_t)nn * a.h + py) * a.w + px;
      xoff[i] = ok ? (uint32_t)((pix * a.x_ld + ld_g * 4) * 4) : 0xFFFFFFF0u;
    }
  }
  const float sgn = (ld_col == 1) ? 1.0f : -1.0f;
  float* const vdst = smem + ld_col * kPlane + ld_tile * kKc + ld_g * 2;   // + buf * 16 planes + i * 4 planes

  // ---- MFMA role
  // weights of this wave: positions 4*wrow .. +3, column blocks nbg .. nbg+NB-1 (block nbg>>2 of 128, sub-block nbg&3)
  const uint32_t u_wave = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)(((((int64_t)(nbg >> 2) * nchunk * 16 + wrow * 4) * 4 + (nbg & 3)) * 256) * 4));   // bytes, wave-uniform
  const uint32_t u_lane = (uint32_t)lane * 16u;
  const float* const asrc = smem + wrow * 4 * kPlane + l31 * kKc + lh * 4;

  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

  f32x4 xr[4];   // [patch row]
  auto load_x = [&](int sc) {
    sc = sc < nsuper ? sc : nsuper - 1;   // past the end: repeat the last one (loaded, never used)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xoff[i], sc * (kSc * 4), 0));
  };
  // Transform of chunk kk of the super-chunk in xr into V buffer `buf`, in five pieces that the K loop spreads over the
  // chunk: the column pass (t = B^T d over the patch rows, registers), then one row of positions at a time (row pass across
  // the quad's lanes with DPP: lane j' gets (t B)[i][j']; 8 bytes to LDS).  Spread out, each piece issues in the shadow of
  // the partner wave's MFMAs; done in one block at either end of the iteration it was exposed (the f32 MFMA leaves about
  // four vector issue slots per 64 cycles).
  f32x2 tcol[4];
  auto xform_cols = [&](int kk) {
    f32x2 d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = f32x2{xr[i][2 * kk], xr[i][2 * kk + 1]};
    tcol[0] = d[0] - d[2];
    tcol[1] = d[1] + d[2];
    tcol[2] = d[2] - d[1];
    tcol[3] = d[1] - d[3];
  };
  auto xform_row = [&](int buf, int i) {
    f32x2 v;
#pragma unroll
    for (int e = 0; e < 2; ++e) v[e] = dpp_p(tcol[i][e]) + sgn * dpp_q(tcol[i][e]);
    *reinterpret_cast<f32x2*>(vdst + buf * (16 * kPlane) + i * (4 * kPlane)) = v;
  };
  auto transform_store = [&](int buf, int kk) {
    xform_cols(kk);
#pragma unroll
    for (int i = 0; i < 4; ++i) xform_row(buf, i);
  };
  // B fragments of the current chunk; position j's pair is re-loaded for the NEXT chunk right after the MFMAs that read it
  // were issued, so a chunk's weight fetch has a whole chunk of MFMAs to land in and only 8 fragment registers per
  // position are live
  f32x4 bw[4][NB];
  auto load_b = [&](int chunk, int j) {
    chunk = chunk < nchunk ? chunk : nchunk - 1;   // the last chunk re-reads itself: constant load count per iteration
    const int so = (int)u_wave + (chunk * 16 + j) * (kPosFloats * 4);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      bw[j][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)u_lane, so + nb * 1024, 0));
  };
  // MFMAs of chunk `buf`'s positions; between the positions: the re-load of that position's weights for the next chunk and
  // one row of the NEXT chunk's transform (into the other buffer)
  auto mma = [&](int buf, int next_chunk, int kk_next) {
    const float* s = asrc + buf * (16 * kPlane);
    f32x4 av = *reinterpret_cast<const f32x4*>(s);
    xform_cols(kk_next);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 an = av;
      if (j < 3) an = *reinterpret_cast<const f32x4*>(s + (j + 1) * kPlane);   // next position's A fragment
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], bw[j][nb][st], acc[j][nb], 0, 0, 0);
      load_b(next_chunk, j);
      xform_row(buf ^ 1, j);
      // keep the re-load and the transform row HERE: left alone, the scheduler sinks all eight loads below the last MFMA of
      // the chunk and the next chunk then starts by waiting out their L2 latency
      __builtin_amdgcn_sched_barrier(0);
      av = an;
    }
  };

  // prologue: chunk 0 into buffer 0.  Issue order as in the loop (x, then the weights): loads retire in order, and the
  // loop's waits for x must find the weight loads YOUNGER than what they wait for on every path into the loop.
  load_x(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) load_b(0, j);
  transform_store(0, 0);
  __syncthreads();

  // One barrier per chunk: iteration c multiplies chunk c out of buffer c&1 while chunk c+1 is transformed, row by row
  // between the positions, into the other buffer (last read in iteration c-1, i.e. before the barrier that ended it).  The x
  // registers of super-chunk S are dead after chunk 2S+1 has been transformed (iteration 2S); super-chunk S+1 is fetched
  // right there and first used one iteration later.  The transform after the last chunk rewrites a buffer nobody reads
  // again.
  for (int sc = 0; sc < nsuper; ++sc) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = sc * 2 + kk;
      mma(kk, c + 1, kk ^ 1);
      if (kk == 0) load_x(sc + 1);
      __syncthreads();
    }
  }

  // ---- output transform.  This wave holds M[i = wrow][j = 0..3]; (M A)[i][0] = M0 + M1 + M2, (M A)[i][1] = M1 - M2 - M3;
  // Y[0][c] = R0 + R1 + R2, Y[1][c] = R1 - R2 - R3 over the four rows i (= waves), which meet in LDS.
  constexpr int kQuads = kCoW / 4;             // output-channel quads of the workgroup
  constexpr int kTilesPass = kThreads / kQuads;  // tiles combined per pass by the 512 threads
  const int o_q = tid % kQuads;
  const int o_t = tid / kQuads;
  const int co0 = cbw * kCoW + o_q * 4;
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};
  if (a.scale) sc4 = *reinterpret_cast<const f32x4*>(a.scale + co0);
  if (a.shift) sh4 = *reinterpret_cast<const f32x4*>(a.shift + co0);
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    float* r = smem + (wrow * kTiles) * kCoW + half * (32 * NB) + l31;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int trow = (g & 3) + 8 * (g >> 2) + 4 * lh;
        const float v = (c2 == 0) ? (acc[0][nb][g] + acc[1][nb][g] + acc[2][nb][g])
                                  : (acc[1][nb][g] - acc[2][nb][g] - acc[3][nb][g]);
        r[trow * kCoW + nb * 32] = v;
      }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < kTiles / kTilesPass; ++ps) {
      const int tl = o_t + ps * kTilesPass;
      const int64_t t = tile0 + tl;
      const float* s = smem + tl * kCoW + o_q * 4;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(s);
      const f32x4 r1 = *reinterpret_cast<const f32x4*>(s + kTiles * kCoW);
      const f32x4 r2 = *reinterpret_cast<const f32x4*>(s + 2 * kTiles * kCoW);
      const f32x4 r3 = *reinterpret_cast<const f32x4*>(s + 3 * kTiles * kCoW);
      if (t < a.n_tiles) {
        const int tx = (int)(t % a.tiles_x);
        const int ty = (int)((t / a.tiles_x) % a.tiles_y);
        const int nn = (int)(t / ((int64_t)a.tiles_x * a.tiles_y));
        const int px = 2 * tx + c2;
        f32x4 y0 = (r0 + r1 + r2) * sc4 + sh4;
        f32x4 y1 = (r1 - r2 - r3) * sc4 + sh4;
        if (a.stats) {   // sums over the pixels this workgroup stores (the statistics of the training-mode BatchNorm that follows)
          const float m0 = (px < a.w) ? 1.f : 0.f, m1 = (px < a.w && 2 * ty + 1 < a.h) ? 1.f : 0.f;
          st_s += y0 * m0 + y1 * m1;
          st_q += y0 * y0 * m0 + y1 * y1 * m1;
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y0[e] = fmaxf(y0[e], 0.f);
            y1[e] = fmaxf(y1[e], 0.f);
          }
        }
        if (px < a.w) {
          const int py = 2 * ty;
          float* dst = a.y + (((int64_t)nn * a.h + py) * a.w + px) * a.y_ld + co0;
          *reinterpret_cast<f32x4*>(dst) = y0;
          if (py + 1 < a.h) *reinterpret_cast<f32x4*>(dst + (int64_t)a.w * a.y_ld) = y1;
        }
      }
    }
    if (c2 == 0) __syncthreads();
  }
  if (a.stats) {
    // the kTilesPass threads that share an output-channel quad meet in LDS; thread ch of the first kCoW adds them up in a
    // fixed order and writes the block's row of partial sums (k_bn_finalize combines the rows in fp64)
    __syncthreads();
    float* red = smem;                                  // [kTilesPass][kQuads][8]
    *reinterpret_cast<f32x4*>(red + (o_t * kQuads + o_q) * 8) = st_s;
    *reinterpret_cast<f32x4*>(red + (o_t * kQuads + o_q) * 8 + 4) = st_q;
    __syncthreads();
    if (tid < kCoW) {
      const int q = tid >> 2, e = tid & 3;
      float s1 = 0.f, s2 = 0.f;
      for (int t = 0; t < kTilesPass; ++t) {
        s1 += red[(t * kQuads + q) * 8 + e];
        s2 += red[(t * kQuads + q) * 8 + 4 + e];
      }
      float* dst = a.stats + (int64_t)bid * 2 * a.cout + cbw * kCoW + tid;
      dst[0] = s1;
      dst[a.cout] = s2;
    }
  }
}

}  // namespace

extern "C" int64_t spx_wino_weight_floats(int32_t cin, int32_t cout) { return (int64_t)16 * cin * cout; }

extern "C" int spx_wino_weight(const float* w, int64_t s_o, int64_t s_i, int64_t s_a, int64_t s_b, int32_t cin, int32_t cout,
                               int flip, float* u, spx_stream_t stream) {
  if (cin <= 0 || cout <= 0 || cin % kSc != 0 || cout % kCoWg != 0) return SPX_ERR_INVALID_ARG;
  const int64_t total = (int64_t)cin * cout;
  hipLaunchKernelGGL(k_wino_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, spx_s(stream), w, s_o, s_i, s_a,
                     s_b, cin, cout, flip, u);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" int64_t spx_wino_stat_rows(int32_t n, int32_t h, int32_t w) {
  const int64_t tiles = (int64_t)n * ((w + 1) / 2) * ((h + 1) / 2);
  return (tiles + kTiles - 1) / kTiles;
}

extern "C" int spx_conv2d_wino(const float* x, int64_t x_ld, const float* u, int32_t n, int32_t h, int32_t w, int32_t cin,
                               int32_t cout, const float* scale, const float* shift, int relu, float* y, int64_t y_ld,
                               float* stat_partials, spx_stream_t stream) {
  if (n <= 0 || h <= 0 || w <= 0) return SPX_OK;
  if (cin % kSc != 0 || cout % kCoWg != 0 || x_ld < cin || y_ld < cout || (x_ld & 3) || (y_ld & 3)) return SPX_ERR_INVALID_ARG;
  const int64_t x_bytes = ((int64_t)n * h * w - 1) * x_ld * 4 + (int64_t)cin * 4, u_bytes = (int64_t)16 * cin * cout * 4;
  if (x_bytes >= 0xFFFFFFF0ll || u_bytes >= 0x7FFFFFFFll) return SPX_ERR_TOO_LARGE;
  WinoArgs a;
  a.x = x; a.u = u; a.y = y; a.scale = scale; a.shift = shift; a.stats = stat_partials;
  a.x_ld = x_ld; a.y_ld = y_ld;
  a.x_bytes = (uint32_t)x_bytes; a.u_bytes = (uint32_t)u_bytes;
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
  a.tiles_x = (w + 1) / 2; a.tiles_y = (h + 1) / 2;
  a.n_tiles = (int64_t)n * a.tiles_x * a.tiles_y;
  a.relu = relu;
  const int64_t nb = (a.n_tiles + kTiles - 1) / kTiles;
  if (nb > 0x7fffffff) return SPX_ERR_INVALID_ARG;
  a.n_blocks = (int32_t)nb;
#ifdef WINO_FORCE_NB
  const bool narrow = WINO_FORCE_NB == 1;
#else
  const bool narrow = true;
#endif
  const int64_t nb8 = (nb + 7) / 8 * 8;
  if (narrow)
    hipLaunchKernelGGL(k_wino_conv<1>, dim3((unsigned)(nb8 * (cout / 64))), dim3(kThreads), 0, spx_s(stream), a);
  else
    hipLaunchKernelGGL(k_wino_conv<2>, dim3((unsigned)(nb8 * (cout / 128))), dim3(kThreads), 0, spx_s(stream), a);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}
