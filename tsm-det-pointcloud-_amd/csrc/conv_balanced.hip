// conv_balanced.hip — sparse convolution forward / dgrad with an MFMA-work-balanced, persistent, block-lockstep schedule.
//
// Same arithmetic as conv_gemm.hip (output-stationary implicit GEMM, v_mfma_f32_16x16x4_f32, same packed weights, same
// gather map).  What changes is WHO does which work and where the weights come from.  In-kernel stamps on k_conv_mfma
// (tools/conv_diag.py) showed (a) a wave waiting out five memory round trips per (tile, offset) unit, ~14,000 cycles
// against 2,048 of MFMA, (b) 1.9 GB of weight fragments moving L2 -> CU per 64x64 launch (every wave re-reads the
// 16 KB slice W_k per unit), and (c) a launch that ends in a long low-occupancy tail because tiles differ in their
// number of non-empty offsets.  Here
//   1. spx_conv_plan records a 27-bit offset mask per 16-row tile, counts the non-empty (64-row super-tile, offset)
//      super-units, prefix-sums them and deals the flat list in equal ranges to a persistent grid of workgroups
//      (4 per CU);
//   2. k_conv_mfma_pbl: the four waves of a workgroup hold the accumulators of the four 16-row tiles of a super-tile and
//      walk the workgroup's super-units in lockstep: W_k is copied global -> LDS once per workgroup and super-unit
//      (global_load_lds_dwordx4, double buffered, in flight while the previous super-unit is multiplied), the rows of
//      the next super-unit are gathered and the rule entries of the one after are read meanwhile; one barrier per
//      super-unit.  A super-tile whose offsets all belong to one workgroup is written to `dst` directly; one shared with
//      neighbouring workgroups goes to a scratch slot as a partial sum;
//   3. the partial sums of a shared super-tile are combined INSIDE the launch: every owner draws a ticket from the tile's
//      arrival counter (agent-scope release before, cdna_hip_programming.md §split-K recipe); the last arriver reads all
//      partials back and adds them in workgroup order (= offset order, a FIXED order whoever arrives last), applies the
//      epilogue and writes the rows.  No float atomics: results are bitwise reproducible from run to run.  (Round 1 did
//      this in a second kernel, k_conv_fixup: 13 us + a launch gap per call.)
// The plan depends only on the rule table, so the python layer caches it per rulebook (forward, dgrad and the second
// layer of a submanifold pair all reuse it).
//
// Serves the same reference call sites as conv_gemm.hip (spconv_backbone.py:86-121 and their autograd).
#include <stdlib.h>

#include "spx_common.h"

#ifdef SPX_CV_DIAG
// diagnostic build only (never shipped): per-wave cycle stamps, see tools/conv_diag.py --balanced
__device__ unsigned long long* g_cb_diag = nullptr;
extern "C" int spx_diag_set_balanced(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cb_diag), &p, sizeof(p)); }
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef SPX_CB_WPB
#define SPX_CB_WPB 4
#endif
#ifndef SPX_CB_TPW
#define SPX_CB_TPW 1
#endif
constexpr int kWpb = SPX_CB_WPB;     // waves per workgroup
constexpr int kTpw = SPX_CB_TPW;     // 16-row tiles per wave (dev builds: 2 = twice the MFMA work between two barriers)
constexpr int kTps = kWpb * kTpw;    // 16-row tiles per super-tile
constexpr int kRows = 16 * kTps;     // destination rows per super-tile
constexpr int kBlocks = 4096 / kWpb;   // persistent workgroups: 4 waves per SIMD in total (256 CUs x 4 SIMDs)
constexpr int kLdsPerBlock = 36 * 1024;   // LDS footprint forced per workgroup: four fit a CU's 160 KiB, a fifth not
constexpr int kPreLds = 12 * 1024;  // prefix entries staged in LDS by the plan kernel (786k rows)
constexpr int kHdr = 4;             // plan header: [0] active workgroups nb, [1] super-units U, [2] super-tiles T4, [3] -

// plan layout (int32): hdr[kHdr] | wstart[kBlocks + 1, padded] | mask[4 * T4cap] | pre4[T4cap + 1] | arrivals[T4cap]
// arrivals: per-super-tile ticket counters of the in-launch combine; zeroed by spx_conv_plan and reset by each last
// arriver, so a plan serves any number of launches — one at a time (forward, dgrad, the sibling layer run in stream order).
__host__ __device__ inline int64_t plan_off_wstart() { return kHdr; }
__host__ __device__ inline int64_t plan_off_mask() { return kHdr + (kBlocks + 1 + 3) / 4 * 4; }   // 16-byte aligned
__host__ __device__ inline int64_t plan_off_pre(int64_t t4cap) { return plan_off_mask() + kTps * t4cap; }
__host__ __device__ inline int64_t plan_off_arr(int64_t t4cap) { return plan_off_pre(t4cap) + t4cap + 1; }
__host__ __device__ inline int64_t plan_ints(int64_t t4cap) { return plan_off_arr(t4cap) + t4cap; }

// first super-unit of workgroup b when U super-units are dealt to nb workgroups
__device__ __forceinline__ int block_u0(int b, int U, int nb) { return (int)((int64_t)b * U / nb); }

// ---------------------------------------------------------------- plan, pass 1: offset mask per 16-row tile
// One wave looks at 4 tiles (lane = 16*sub + r): bit k of mask[t] = some row of tile t has a neighbour at table row k.
__global__ __launch_bounds__(256) void k_plan_mask(const int32_t* __restrict__ pair, int64_t ld, int K, int64_t n,
                                                   const int64_t* d_n, int64_t t4cap, int32_t* __restrict__ plan) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, r = lane & 15;
  const int64_t nlive = spx_live_n(d_n, n);
  const int64_t T = (nlive + 15) / 16;
  const int64_t tile = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + sub;
  const int64_t row = tile * 16 + r;
  const bool ok = tile < T && row < nlive;
  uint32_t m = 0;
  for (int k0 = 0; k0 < K; k0 += 9) {
    int32_t id[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) id[u] = (ok && k0 + u < K) ? pair[(int64_t)(k0 + u) * ld + row] : -1;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const unsigned long long b = __ballot(id[u] >= 0);
      if ((b >> (16 * sub)) & 0xFFFFull) m |= 1u << (k0 + u);
    }
  }
  if (r == 0 && tile < kTps * t4cap) plan[plan_off_mask() + tile] = tile < T ? (int32_t)m : 0;
}

// ---------------------------------------------------------------- plan, pass 2 (one block): prefix over super-tiles, cuts
__global__ __launch_bounds__(1024) void k_plan_scan(int64_t n, const int64_t* d_n, int64_t t4cap, int32_t* __restrict__ plan) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t nlive = spx_live_n(d_n, n);
  const int T4 = (int)((nlive + kRows - 1) / kRows);
  const int32_t* mask = plan + plan_off_mask();
  int32_t* pre = plan + plan_off_pre(t4cap);
  int32_t* wstart = plan + plan_off_wstart();
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < T4; base += 1024) {          // block-wide exclusive scan, 1024 super-tiles per trip
    const int S = base + tid;
    int v = 0;
    if (S < T4) {
      int many = 0;
#pragma unroll
      for (int i = 0; i < kTps; i += 4) {
        const int4 m = *reinterpret_cast<const int4*>(mask + (size_t)kTps * S + i);
        many |= m.x | m.y | m.z | m.w;
      }
      v = __popc((unsigned)many);                        // super-units: offsets present in any tile of the super-tile
    }
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int off = s_carry;
    for (int i = 0; i < wv; ++i) off += s_wave[i];
    if (S < T4) {
      pre[S] = off + incl - v;
      plan[plan_off_arr(t4cap) + S] = 0;
    }
    __syncthreads();
    if (tid == 1023) s_carry = off + incl;
    __syncthreads();
  }
  const int U = s_carry;
  if (tid == 0) {
    pre[T4] = U;
    int nb = U / 8;                       // at least 8 super-units per workgroup
    if (nb > kBlocks) nb = kBlocks;
    if (nb >= 8) nb &= ~7;                // a multiple of 8: the XCD-contiguous numbering of the main kernel needs it
    if (nb < 1) nb = 1;                   // also when U == 0: workgroup 0 still writes the empty super-tiles
    plan[0] = nb;
    plan[1] = U;
    plan[2] = T4;
    plan[3] = 0;
  }
  __syncthreads();
  const int nb = plan[0];
  // wstart[b] = the super-tile that holds super-unit u0(b): smallest S with pre[S + 1] > u0.  The searches read the
  // prefix from LDS when it fits (11 dependent reads each: from L2 they were most of this kernel's 20 us).
  __shared__ int s_pre[kPreLds];
  const bool in_lds = T4 + 1 <= kPreLds;
  if (in_lds)
    for (int i = tid; i <= T4; i += 1024) s_pre[i] = pre[i];
  __syncthreads();
  for (int b = tid; b <= kBlocks; b += 1024) {
    int s0 = T4;
    if (b < nb) {
      const int u0 = block_u0(b, U, nb);
      int lo = 0, hi = T4;
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        const int v = in_lds ? s_pre[mid + 1] : pre[mid + 1];
        if (v > u0) hi = mid; else lo = mid + 1;
      }
      s0 = lo;
    }
    wstart[b] = s0;
  }
}

// ---------------------------------------------------------------- in-launch combine of shared super-tiles
// owners of super-tile S = the workgroups whose unit range meets [p0, p0 + work): b_lo = last one starting at or before
// p0, b_hi = last one starting before p0 + work
__device__ __forceinline__ void tile_owners(int p0, int work, int U, int nb, int& b_lo, int& b_hi) {
  int a = 0, b = nb - 1;
  while (a < b) {
    const int mid = (a + b + 1) >> 1;
    if (block_u0(mid, U, nb) <= p0) a = mid; else b = mid - 1;
  }
  b_lo = a;
  b = nb - 1;
  while (a < b) {
    const int mid = (a + b + 1) >> 1;
    if (block_u0(mid, U, nb) < p0 + work) a = mid; else b = mid - 1;
  }
  b_hi = a;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Partial sums travel through `scratch` in the MFMA accumulator's own layout, slab[wave][nt][lane] = one f32x4 (rows
// 16*wave + 4*(lane>>4) + e, column 16*nt + (lane&15)): every store and load is one coalesced 16-byte access per lane.
// They are handed from workgroup to workgroup inside the launch, so (cdna_hip_programming.md §6 Guideline 16, R1) every
// store is WRITE-THROUGH (`sc1`: no agent-scope release fence, which would write back the XCD's whole L2 once per
// workgroup — measured: +25 us per launch), every storing wave drains its stores, the workgroup meets at a barrier, ONE
// lane draws the tile's ticket with an agent-scope atomic; the workgroup that draws the last ticket acquires once
// (invalidates its CU's L1) and reads every slab with `sc1` loads.
__device__ __forceinline__ int slab_byte_off(int slot_index, int tile_floats, int NT, int wave, int nt, int lane, int t = 0) {
  return (slot_index * (tile_floats / 4) + ((wave * kTpw + t) * NT + nt) * 64 + lane) * 16;
}

// Called by ALL threads of a workgroup right after they stored their partial of shared super-tile S (sc1 stores).
// Returns 0, or (b_lo | b_hi << 12) + 1 in the one workgroup that arrived last (then every partial of S is visible to it).
// `flag` is one LDS word no wave is reading or writing at this point.
__device__ __forceinline__ int tile_arrive(int S, int p0, int work, int U, int nb, int32_t* arrivals, volatile int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    int b_lo, b_hi;
    tile_owners(p0, work, U, nb, b_lo, b_hi);
    const int t = __hip_atomic_fetch_add(&arrivals[S], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int f = 0;
    if (t == b_hi - b_lo) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&arrivals[S], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      f = (b_lo | (b_hi << 12)) + 1;
    }
    *flag = f;
  }
  __syncthreads();
  return *flag;
}

// last arriver: this wave's 16 rows of super-tile S = sum of the owners' partials in workgroup order (a fixed order,
// whoever arrives last), epilogue, scatter through perm — the same code shape as the whole-tile write of the main loop
template <int CD>
__device__ __forceinline__ void tile_combine(int S, int owners, const int32_t* __restrict__ wstart,
                                             __amdgpu_buffer_rsrc_t rs, int64_t nlive, const float* __restrict__ scale,
                                             const float* __restrict__ shift, int relu, const int32_t* __restrict__ perm,
                                             float* __restrict__ dst) {
  constexpr int NT = CD / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int b_lo = (owners - 1) & 0xFFF, b_hi = (owners - 1) >> 12;
  for (int t = 0; t < kTpw; ++t) {
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = b_lo; b <= b_hi; ++b) {
      const int slot = 2 * b + (wstart[b] == S ? 0 : 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, slab_byte_off(slot, kRows * CD, NT, wave, nt, lane, t), 0, 16);
        acc[nt] += __builtin_bit_cast(f32x4, u);
      }
    }
    int64_t drow[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t orow = (int64_t)S * kRows + 16 * (kTpw * wave + t) + 4 * q + e;
      drow[e] = orow < nlive ? (perm ? (int64_t)perm[orow] : orow) : -1;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = 16 * nt + r;
      const float sc = scale ? scale[col] : 1.0f;
      const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (drow[e] >= 0) {
          float v = acc[nt][e];
          if (scale || shift) v = v * sc + sh;
          if (relu) v = v > 0.f ? v : 0.f;
          dst[drow[e] * CD + col] = v;
        }
      }
    }
  }
}

// super-tiles with no rule pair at all (possible on backward tables): rows = epilogue(0); dealt round-robin
template <int CD, int NTHR>
__device__ __forceinline__ void write_empty_tiles(int blk, int nb, int T4, const int32_t* __restrict__ pre, int64_t nlive,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  int relu, const int32_t* __restrict__ perm, float* __restrict__ dst) {
  constexpr int V = kRows * CD / 4;
  for (int S = blk; S < T4; S += nb) {
    if (pre[S + 1] != pre[S]) continue;
    for (int i = threadIdx.x; i < V; i += NTHR) {
      const int rr = (4 * i) / CD, c0 = (4 * i) % CD;
      const int64_t orow = (int64_t)S * kRows + rr;
      if (orow >= nlive) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = shift ? shift[c0 + e] : 0.0f;
        if (relu) x = x > 0.f ? x : 0.f;
        v[e] = x;
      }
      const int64_t drow = perm ? (int64_t)perm[orow] : orow;
      *reinterpret_cast<f32x4*>(dst + drow * CD + c0) = v;
    }
  }
}

// ---------------------------------------------------------------- persistent, balanced, block-lockstep implicit GEMM
template <int CS, int CD>
__global__ __launch_bounds__(64 * kWpb) __attribute__((amdgpu_waves_per_eu(kTpw == 1 ? 4 : 3, kTpw == 1 ? 4 : 4))) void k_conv_mfma_pbl(const float* __restrict__ src, const float* __restrict__ wp,
                                                       const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
                                                       int64_t n, const int64_t* d_n, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu,
                                                       const int32_t* __restrict__ plan, int64_t t4cap,
                                                       const int32_t* __restrict__ perm, float* __restrict__ dst,
                                                       float* __restrict__ scratch, int32_t* arrivals) {
  constexpr int NT = CD / 16;
  constexpr int JG = CS / 16;
  constexpr int NF = NT * JG;               // 1 KiB weight fragments per offset
  __shared__ f32x4 sBF[2 * NF * 64 + 1];    // two weight buffers + one word for the "arrived last" broadcast (ONE object)
  f32x4 (*sB)[NF * 64] = reinterpret_cast<f32x4 (*)[NF * 64]>(sBF);
  volatile int* s_flag = reinterpret_cast<volatile int*>(&sBF[2 * NF * 64]);
  extern __shared__ char occupancy_pad[];   // tops the footprint up to kLdsPerBlock
  (void)occupancy_pad;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nb = plan[0];
  if ((int)blockIdx.x >= nb) return;        // whole workgroup: no barrier has been reached yet
  // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  Renumber so that XCD x walks a CONTIGUOUS eighth of the
  // super-unit list: neighbouring destination rows gather overlapping source rows, and each XCD's private L2 then
  // holds one eighth of the feature matrix (plus halo) instead of a sample of all of it.
#ifndef SPX_CB_NO_XCD
  const int blk = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#else
  const int blk = blockIdx.x;
#endif
  const int U = plan[1], T4 = plan[2];
  const int64_t nlive = spx_live_n(d_n, n);
  const int32_t* mask = plan + plan_off_mask();
  const int32_t* pre = plan + plan_off_pre(t4cap);
  const int u0 = block_u0(blk, U, nb), u1 = block_u0(blk + 1, U, nb);
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const int s_first = plan[plan_off_wstart() + blk];
  const __amdgpu_buffer_rsrc_t rs_scratch =
      __builtin_amdgcn_make_buffer_rsrc(scratch, 0, 2 * kBlocks * kRows * CD * (int)sizeof(float), 0x00020000);
  write_empty_tiles<CD, 64 * kWpb>(blk, nb, T4, pre, nlive, scale, shift, relu, perm, dst);

  // ---- the workgroup's super-units in order (all values wave-uniform and equal in the four waves)
  struct Cur {
    int S, k;            // super-tile, loop index of the offset (weights W_k, table row trow(k)); S < 0: past the end
    uint32_t mself[kTpw];   // offset masks of THIS wave's tiles of S
    int whole;           // S belongs to this workgroup alone
  };
  int it_S = s_first - 1;
  uint32_t it_todo = 0, it_mself[kTpw];
#pragma unroll
  for (int t = 0; t < kTpw; ++t) it_mself[t] = 0;
  int it_whole = 0;
  auto next = [&]() -> Cur {
    Cur out;
    out.S = -1, out.k = 0, out.whole = 0;
#pragma unroll
    for (int t = 0; t < kTpw; ++t) out.mself[t] = 0u;
    while (it_todo == 0) {
      ++it_S;
      if (it_S >= T4) return out;
      const int p0 = pre[it_S];
      if (p0 >= u1) return out;
      const int work = pre[it_S + 1] - p0;
      if (work == 0) continue;                                   // empty super-tile: written up front (write_empty_tiles)
      uint32_t many = 0;
#pragma unroll
      for (int i = 0; i < kTps; i += 4) {
        const int4 m4 = *reinterpret_cast<const int4*>(mask + (size_t)kTps * it_S + i);
        many |= (uint32_t)(m4.x | m4.y | m4.z | m4.w);
      }
#pragma unroll
      for (int t = 0; t < kTpw; ++t) it_mself[t] = (uint32_t)mask[(size_t)kTps * it_S + kTpw * wave + t];
      it_whole = (u0 <= p0 && p0 + work <= u1) ? 1 : 0;
      int c = p0;
      for (int k = 0; k < K; ++k) {                              // owned: super-units whose index lies in [u0, u1)
        const int tr = flip ? K - 1 - k : k;
        if (!((many >> tr) & 1u)) continue;
        if (c >= u0 && c < u1) it_todo |= 1u << k;
        ++c;
      }
    }
    out.S = it_S;
    out.k = __ffs((int)it_todo) - 1;
    it_todo &= it_todo - 1;
#pragma unroll
    for (int t = 0; t < kTpw; ++t) out.mself[t] = it_mself[t];
    out.whole = it_whole;
    return out;
  };

  auto copy_w = [&](int k, int buf) {
#pragma unroll
    for (int i = 0; i < (NF + kWpb - 1) / kWpb; ++i) {
      const int f = wave + i * kWpb;        // wave-uniform
      if (f < NF)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wp4 + ((size_t)k * NF + f) * 64 + lane),
            (__attribute__((address_space(3))) void*)(&sB[buf][f * 64]), 16, 0, 0);
    }
  };
  // (Returning the entry raw and applying its validity at the consumers removes the wait the compiler puts right behind
  // this load, but the kernel then runs 5 % SLOWER — measured, same call — so the select stays here.)
  auto load_id = [&](const Cur& c, int t) -> int32_t {       // unconditional (row clamped); -1 where not applicable
    const int S = c.S >= 0 ? c.S : 0;
    const int64_t row = (int64_t)S * kRows + 16 * (kTpw * wave + t) + r;
    const int64_t rc = row < nlive ? row : nlive - 1;
    const int32_t v = pair[(int64_t)(flip ? K - 1 - c.k : c.k) * ld + rc];
    return (c.S >= 0 && row < nlive) ? v : -1;
  };
  auto gather = [&](int32_t id, f32x4 (&a)[JG]) {            // unconditional (index clamped); masked when used
    const float* p = src + (size_t)(id > 0 ? id : 0) * CS + 4 * q;
#pragma unroll
    for (int jg = 0; jg < JG; ++jg) a[jg] = *reinterpret_cast<const f32x4*>(p + 16 * jg);
  };

  f32x4 acc[kTpw][NT];
#pragma unroll
  for (int t = 0; t < kTpw; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Pipeline: while super-unit c0 is multiplied, the rows of c1 and the weight slice of c1 are in flight and the rule
  // entries of c2 are read; the barrier at the top of each trip is the one point where they are waited for.
#ifdef SPX_CV_DIAG
  const unsigned long long d_t0 = __builtin_amdgcn_s_memtime(), d_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long d_bar = 0, d_mma = 0, d_units = 0, d_act = 0;
#endif
  Cur c0 = next(), c1 = next(), c2 = next();
  int32_t id0[kTpw], id1[kTpw];
  f32x4 a_cur[kTpw][JG], a_nxt[kTpw][JG];
#pragma unroll
  for (int t = 0; t < kTpw; ++t) id0[t] = load_id(c0, t), id1[t] = load_id(c1, t);
  if (c0.S >= 0) copy_w(c0.k, 0);
#pragma unroll
  for (int t = 0; t < kTpw; ++t) gather(id0[t], a_cur[t]);
  int it = 0;
  while (c0.S >= 0) {
#ifdef SPX_CV_DIAG
    const unsigned long long d_a = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();   // W(c0) landed for every wave, a_cur / id1 arrived; the other weight buffer is free again
#ifdef SPX_CV_DIAG
    d_bar += __builtin_amdgcn_s_memtime() - d_a;
    d_units += 1;
#endif
    int32_t id2[kTpw];
#pragma unroll
    for (int t = 0; t < kTpw; ++t) id2[t] = load_id(c2, t);
#pragma unroll
    for (int t = 0; t < kTpw; ++t) gather(id1[t], a_nxt[t]);
    if (c1.S >= 0) copy_w(c1.k, (it + 1) & 1);
    // everything above is only ISSUED here; keep it above the MFMAs (the scheduler would otherwise sink the gathers
    // next to their use in the next trip and serialise latency and arithmetic again)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int tr0 = flip ? K - 1 - c0.k : c0.k;
    bool act[kTpw];
    bool any_act = false;
#pragma unroll
    for (int t = 0; t < kTpw; ++t) {
      act[t] = (c0.mself[t] >> tr0) & 1u;   // wave-uniform: this tile has the offset
      any_act |= act[t];
    }
#ifdef SPX_CV_DIAG
    const unsigned long long d_c = __builtin_amdgcn_s_memtime();
    if (any_act) d_act += 1;
#endif
    if (any_act) {
      const f32x4* B = sB[it & 1];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        f32x4 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = B[(nt * JG + jg) * 64 + lane];
#pragma unroll
        for (int t = 0; t < kTpw; ++t) {
          if (!act[t]) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = id0[t] >= 0 ? a_cur[t][jg][e] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt][e], acc[t][nt], 0, 0, 0);
          }
        }
      }
    }
#ifdef SPX_CV_DIAG
    d_mma += __builtin_amdgcn_s_memtime() - d_c;
#endif
    if (c1.S != c0.S) {
      // last owned offset of super-tile c0.S: write this wave's rows.  C layout: col = lane&15, row = 4*(lane>>4) + e
      if (c0.whole) {
#pragma unroll
        for (int t = 0; t < kTpw; ++t) {
          const int64_t row_base = (int64_t)c0.S * kRows + 16 * (kTpw * wave + t);
          int64_t drow[4];                     // where the rows of this position go: perm[] under a grouped row order
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t orow = row_base + 4 * q + e;
            drow[e] = orow < nlive ? (perm ? (int64_t)perm[orow] : orow) : -1;
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = 16 * nt + r;
            const float sc = scale ? scale[col] : 1.0f;
            const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (drow[e] >= 0) {
                float v = acc[t][nt][e];
                if (scale || shift) v = v * sc + sh;
                if (relu) v = v > 0.f ? v : 0.f;
                dst[drow[e] * CD + col] = v;
              }
            }
          }
        }
      } else {
        // partial sums of a super-tile shared with neighbouring workgroups: slot 0 if it is this workgroup's first
        // super-tile, else slot 1
        const int slot = 2 * blk + (c0.S == s_first ? 0 : 1);
#pragma unroll
        for (int t = 0; t < kTpw; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][nt]), rs_scratch,
                                                   slab_byte_off(slot, kRows * CD, NT, wave, nt, lane, t), 0, 16);
        const int p0 = pre[c0.S];
        const int owners = tile_arrive(c0.S, p0, pre[c0.S + 1] - p0, U, nb, arrivals, s_flag);
        if (owners) tile_combine<CD>(c0.S, owners, plan + plan_off_wstart(), rs_scratch, nlive, scale, shift, relu, perm, dst);
      }
#pragma unroll
      for (int t = 0; t < kTpw; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < kTpw; ++t) {
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a_cur[t][jg] = a_nxt[t][jg];
      id0[t] = id1[t];
      id1[t] = id2[t];
    }
    c0 = c1;
    c1 = c2;
    c2 = next();
    ++it;
  }
#ifdef SPX_CV_DIAG
  if (g_cb_diag && lane == 0) {
    unsigned long long* o = g_cb_diag + ((size_t)blk * kWpb + wave) * 8;
    o[0] = __builtin_amdgcn_s_memtime() - d_t0;   // wave lifetime, core cycles
    o[1] = d_units;                               // super-units walked
    o[2] = d_act;                                 // super-units in which this wave multiplied
    o[3] = d_bar;                                 // cycles inside __syncthreads (incl. its vmcnt(0) drain)
    o[4] = d_mma;                                 // cycles in the LDS-read + MFMA phase
    o[5] = d_r0;                                  // start, 100 MHz ticks
    o[6] = __builtin_amdgcn_s_memrealtime();      // end
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[7] = (unsigned long long)hw | ((unsigned long long)xcc << 32);   // where the wave ran (SIMD / CU / SE / XCC)
  }
#endif
}

// ---------------------------------------------------------------- the same schedule for wide layers (128 channels)
// A 128x128 weight slice is 64 KiB: two of them do not fit the static LDS budget, and a one-tile-per-wave kernel re-reads
// the slice per 16 rows (1.7 GB L2 -> CU per launch for the BEV entry conv, its bound).  Here the destination columns
// are multiplied in two halves: half h of W_k lives in LDS buffer h (32 KiB each), a super-unit is two sub-steps
//   barrier ; issue {rule entries of c2, rows of c1, W_k half 1 -> buffer 1} ; MFMA half 0 from buffer 0
//   barrier ; issue {W_k' half 0 of the next super-unit -> buffer 0}          ; MFMA half 1 from buffer 1
// so every copy has a whole sub-step (4096 MFMA cycles) to land.  64 KiB of LDS per workgroup = two workgroups per CU;
// the plan still deals its ranges to up to 1024 virtual workgroups, each physical workgroup walks two of them one after
// the other (same partial-sum slots, same fix-up kernel, same bits as a 1024-workgroup launch would give).
template <int CS, int CD>
__global__ __launch_bounds__(64 * kWpb) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv_mfma_pbl2(
    const float* __restrict__ src, const float* __restrict__ wp, const int32_t* __restrict__ pair, int64_t ld, int K, int flip,
    int64_t n, const int64_t* d_n, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
    const int32_t* __restrict__ plan, int64_t t4cap, const int32_t* __restrict__ perm, float* __restrict__ dst,
    float* __restrict__ scratch, int32_t* arrivals) {
  static_assert(kWpb == 4, "two-half kernel is written for 4 waves per workgroup");
  if constexpr (kTpw != 1) return;          // dev builds with two tiles per wave do not cover the 128-channel kernel
  constexpr int NT = CD / 16;
  constexpr int NTH = NT / 2;               // column tiles per half
  constexpr int JG = CS / 16;
  constexpr int NF = NT * JG;               // 1 KiB weight fragments per offset
  constexpr int NFH = NTH * JG;             // per half
  __shared__ f32x4 sBF[2 * NFH * 64 + 1];   // two half-slice buffers + the "arrived last" word (ONE object)
  f32x4 (*sB)[NFH * 64] = reinterpret_cast<f32x4 (*)[NFH * 64]>(sBF);
  volatile int* s_flag = reinterpret_cast<volatile int*>(&sBF[2 * NFH * 64]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nb = plan[0];
  const int np = (nb + 1) >> 1;             // physical workgroups that have work
  if ((int)blockIdx.x >= np) return;        // whole workgroup: no barrier has been reached yet
  const int lb = (np & 7) == 0 ? (int)(blockIdx.x & 7) * (np >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // XCD-contiguous
  const int U = plan[1], T4 = plan[2];
  const int64_t nlive = spx_live_n(d_n, n);
  const int32_t* mask = plan + plan_off_mask();
  const int32_t* pre = plan + plan_off_pre(t4cap);
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const __amdgpu_buffer_rsrc_t rs_scratch =
      __builtin_amdgcn_make_buffer_rsrc(scratch, 0, 2 * kBlocks * kRows * CD * (int)sizeof(float), 0x00020000);
  write_empty_tiles<CD, 64 * kWpb>(lb, np, T4, pre, nlive, scale, shift, relu, perm, dst);

  for (int vv = 0; vv < 2; ++vv) {
    const int blk = 2 * lb + vv;            // virtual workgroup of the plan
    if (blk >= nb) break;
    __syncthreads();                        // the previous virtual workgroup's last reads of the LDS buffers are done
    const int u0 = block_u0(blk, U, nb), u1 = block_u0(blk + 1, U, nb);
    const int s_first = plan[plan_off_wstart() + blk];
    struct Cur {
      int S, k;
      uint32_t mself;
      int whole;
    };
    int it_S = s_first - 1;
    uint32_t it_todo = 0, it_mself = 0;
    int it_whole = 0;
    auto next = [&]() -> Cur {
      while (it_todo == 0) {
        ++it_S;
        if (it_S >= T4) return Cur{-1, 0, 0u, 0};
        const int p0 = pre[it_S];
        if (p0 >= u1) return Cur{-1, 0, 0u, 0};
        const int work = pre[it_S + 1] - p0;
        if (work == 0) continue;
        const int4 m4 = *reinterpret_cast<const int4*>(mask + (size_t)kWpb * it_S);
        const uint32_t many = (uint32_t)(m4.x | m4.y | m4.z | m4.w);
        it_mself = (uint32_t)mask[(size_t)kWpb * it_S + wave];
        it_whole = (u0 <= p0 && p0 + work <= u1) ? 1 : 0;
        int c = p0;
        for (int k = 0; k < K; ++k) {
          const int tr = flip ? K - 1 - k : k;
          if (!((many >> tr) & 1u)) continue;
          if (c >= u0 && c < u1) it_todo |= 1u << k;
          ++c;
        }
      }
      const int k = __ffs((int)it_todo) - 1;
      it_todo &= it_todo - 1;
      return Cur{it_S, k, it_mself, it_whole};
    };
    auto copy_w = [&](int k, int h) {       // half h of W_k -> LDS buffer h
#pragma unroll
      for (int i = 0; i < NFH / kWpb; ++i) {
        const int f = wave + i * kWpb;      // wave-uniform
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wp4 + ((size_t)k * NF + (size_t)h * NFH + f) * 64 + lane),
            (__attribute__((address_space(3))) void*)(&sB[h][f * 64]), 16, 0, 0);
      }
    };
    auto load_id = [&](const Cur& c) -> int32_t {
      const int S = c.S >= 0 ? c.S : 0;
      const int64_t row = (int64_t)S * kRows + 16 * wave + r;
      const int64_t rc = row < nlive ? row : nlive - 1;
      const int32_t v = pair[(int64_t)(flip ? K - 1 - c.k : c.k) * ld + rc];
      return (c.S >= 0 && row < nlive) ? v : -1;
    };
    auto gather = [&](int32_t id, f32x4 (&a)[JG]) {
      const float* p = src + (size_t)(id > 0 ? id : 0) * CS + 4 * q;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a[jg] = *reinterpret_cast<const f32x4*>(p + 16 * jg);
    };
    auto mma_half = [&](int h, const f32x4 (&a)[JG], bool live, f32x4 (&acc)[NT]) {
      const f32x4* B = sB[h];
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        f32x4 b[NTH];
#pragma unroll
        for (int nt = 0; nt < NTH; ++nt) b[nt] = B[(nt * JG + jg) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = live ? a[jg][e] : 0.f;
#pragma unroll
          for (int nt = 0; nt < NTH; ++nt)
            acc[h * NTH + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt][e], acc[h * NTH + nt], 0, 0, 0);
        }
      }
    };

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    Cur c0 = next(), c1 = next(), c2 = next();
    int32_t id0 = load_id(c0), id1 = load_id(c1);
    if (c0.S >= 0) copy_w(c0.k, 0);
    f32x4 a_cur[JG], a_nxt[JG];
    gather(id0, a_cur);
    while (c0.S >= 0) {
      const int tr0 = flip ? K - 1 - c0.k : c0.k;
      const bool active = (c0.mself >> tr0) & 1u;      // wave-uniform: this wave's tile has the offset
      __syncthreads();   // half 0 of W(c0) landed for every wave, a_cur / id1 arrived; buffer 1 is free again
      const int32_t id2 = load_id(c2);
      gather(id1, a_nxt);
      copy_w(c0.k, 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bool live0 = id0 >= 0;
      if (active) mma_half(0, a_cur, live0, acc);
      __syncthreads();   // half 1 of W(c0) landed; buffer 0 is free again
      if (c1.S >= 0) copy_w(c1.k, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (active) mma_half(1, a_cur, live0, acc);
      if (c1.S != c0.S) {
        const int64_t row_base = (int64_t)c0.S * kRows + 16 * wave;
        if (c0.whole) {
          int64_t drow[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t orow = row_base + 4 * q + e;
            drow[e] = orow < nlive ? (perm ? (int64_t)perm[orow] : orow) : -1;
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = 16 * nt + r;
            const float sc = scale ? scale[col] : 1.0f;
            const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (drow[e] >= 0) {
                float v = acc[nt][e];
                if (scale || shift) v = v * sc + sh;
                if (relu) v = v > 0.f ? v : 0.f;
                dst[drow[e] * CD + col] = v;
              }
            }
          }
        } else {
          const int slot = 2 * blk + (c0.S == s_first ? 0 : 1);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[nt]), rs_scratch,
                                                   slab_byte_off(slot, kRows * CD, NT, wave, nt, lane), 0, 16);
          const int p0 = pre[c0.S];
          const int owners = tile_arrive(c0.S, p0, pre[c0.S + 1] - p0, U, nb, arrivals, s_flag);
          if (owners) tile_combine<CD>(c0.S, owners, plan + plan_off_wstart(), rs_scratch, nlive, scale, shift, relu, perm, dst);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) a_cur[jg] = a_nxt[jg];
      id0 = id1;
      id1 = id2;
      c0 = c1;
      c1 = c2;
      c2 = next();
    }
  }
}

static inline int64_t tiles4_cap(int64_t n) { return (n + kRows - 1) / kRows + 1; }

template <int CS, int CD>
static void launch_pb(const float* src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip, int64_t n,
                      const int64_t* d_n, const float* scale, const float* shift, int relu, const int32_t* plan,
                      const int32_t* perm, float* dst, float* scratch, hipStream_t s) {
  const int64_t t4cap = tiles4_cap(n);
  constexpr int kStatic = 2 * (CS / 16) * (CD / 16) * 1024 + 16;  // the two weight buffers + the flag word
  // (a 16-wave workgroup fills a CU's wave slots at this register count by itself: no padding needed)
  const int pad = (kWpb == 4 && kLdsPerBlock > kStatic) ? kLdsPerBlock - kStatic : 0;
  hipLaunchKernelGGL((k_conv_mfma_pbl<CS, CD>), dim3(kBlocks), dim3(64 * kWpb), pad, s, src, wp, pair, ld, K, flip, n, d_n, scale,
                     shift, relu, plan, t4cap, perm, dst, scratch, const_cast<int32_t*>(plan) + plan_off_arr(t4cap));
}

template <int CS, int CD>
static void launch_pb2(const float* src, const float* wp, const int32_t* pair, int64_t ld, int K, int flip, int64_t n,
                       const int64_t* d_n, const float* scale, const float* shift, int relu, const int32_t* plan,
                       const int32_t* perm, float* dst, float* scratch, hipStream_t s) {
  const int64_t t4cap = tiles4_cap(n);
  hipLaunchKernelGGL((k_conv_mfma_pbl2<CS, CD>), dim3(kBlocks / 2), dim3(64 * kWpb), 0, s, src, wp, pair, ld, K, flip, n, d_n,
                     scale, shift, relu, plan, t4cap, perm, dst, scratch, const_cast<int32_t*>(plan) + plan_off_arr(t4cap));
}

}  // namespace

extern "C" size_t spx_conv_plan_bytes(int64_t n_dst) {
  return spx_align((size_t)plan_ints(tiles4_cap(n_dst < 0 ? 0 : n_dst)) * sizeof(int32_t));
}

extern "C" int spx_conv_plan(const int32_t* pair, int64_t pair_ld, int kvol, int64_t n_dst, const int64_t* d_n_dst,
                             int32_t* plan, spx_stream_t stream) {
  if (!pair || !plan || kvol <= 0 || kvol > SPX_MAX_KVOL || n_dst <= 0 || pair_ld < n_dst) return SPX_ERR_INVALID_ARG;
  if (n_dst >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  hipStream_t s = spx_s(stream);
  const int64_t t4cap = tiles4_cap(n_dst);
  hipLaunchKernelGGL(k_plan_mask, dim3((unsigned)((kTps * t4cap + 15) / 16)), dim3(256), 0, s, pair, pair_ld, kvol, n_dst,
                     d_n_dst, t4cap, plan);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, s, n_dst, d_n_dst, t4cap, plan);
  SPX_CHECK_LAUNCH();
  return SPX_OK;
}

extern "C" size_t spx_conv_gemm_balanced_ws_bytes(int c_dst, int64_t n_dst) {
  if (c_dst <= 0 || n_dst <= 0) return 0;
  return spx_align((size_t)2 * kBlocks * kRows * c_dst * sizeof(float));
}

#define SPX_PB_CASE(A, B)                                                                                            \
  if (c_src == A && c_dst == B) {                                                                                    \
    launch_pb<A, B>(src, w_packed, pair, pair_ld, kvol, flip_k, n_dst, d_n_dst, scale, shift, relu, plan, perm, dst,    \
                    scratch, s);                                                                                        \
    SPX_CHECK_LAUNCH();                                                                                              \
    return SPX_OK;                                                                                                   \
  }

extern "C" int spx_conv_gemm_balanced(const float* src, int c_src, const float* w_packed, int c_dst, int kvol, int flip_k,
                                      const int32_t* pair, int64_t pair_ld, int64_t n_dst, const int64_t* d_n_dst,
                                      const float* scale, const float* shift, int relu, int32_t* plan,
                                      const int32_t* perm, float* dst, void* ws, size_t ws_bytes, spx_stream_t stream) {
  if (!src || !w_packed || !pair || !dst || !plan || c_src <= 0 || c_dst <= 0 || kvol <= 0 || kvol > SPX_MAX_KVOL ||
      n_dst <= 0 || pair_ld < n_dst)
    return SPX_ERR_INVALID_ARG;
  if (n_dst >= (int64_t(1) << 31)) return SPX_ERR_TOO_LARGE;
  if (!ws || ws_bytes < spx_conv_gemm_balanced_ws_bytes(c_dst, n_dst)) return SPX_ERR_WORKSPACE;
  hipStream_t s = spx_s(stream);
  float* scratch = reinterpret_cast<float*>(ws);
  SPX_PB_CASE(32, 32)
  SPX_PB_CASE(32, 64)
  SPX_PB_CASE(64, 32)
  SPX_PB_CASE(64, 64)
  if (c_src == 128 && c_dst == 128 && kWpb == 4 && kTpw == 1) {
    launch_pb2<128, 128>(src, w_packed, pair, pair_ld, kvol, flip_k, n_dst, d_n_dst, scale, shift, relu, plan, perm, dst, scratch,
                         s);
    SPX_CHECK_LAUNCH();
    return SPX_OK;
  }
  return SPX_ERR_UNSUPPORTED;   // other channel pairs: use spx_conv_gemm
}
