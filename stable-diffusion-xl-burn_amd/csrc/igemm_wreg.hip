// Linear-layer GEMM with the weights streamed STRAIGHT INTO REGISTERS (round 4).
//
// Why a third structure next to igemm_pipe_kernel / igemm_wide_kernel.  The timeline of the 96x128 pipe kernel (the M = 2048 x
// N = 1280 linears: 257 launches of a UNet step) shows a k-tile of 1279 cycles for 384 cycles of matrix-pipe work: six waves on
// four SIMDs (two SIMDs carry twice the MFMAs), both operands staged through LDS (3 ds_read_b128 per 2 MFMAs) and a rendezvous
// of all waves per 64-deep k-tile with nothing else resident on the CU to fill it (140 KiB of LDS per workgroup).  Here:
//   * the WEIGHTS never touch LDS.  They are re-packed once at model build in MFMA A-operand order (launch_repack_wfrag: per
//     32-column block and 64-deep k-tile four 1-KiB fragments, lane l = weight row l&31, k = 16 kk + 8 (l>>5) .. +7), so a wave
//     streams its own 32 columns as ONE contiguous run with plain `global_load_dwordx4` (1 KiB per instruction, whole lines), L + 1
//     k-tiles deep in registers (3 in production).  A weight fragment is private to its wave: no barrier, no LDS bytes, no ds_read for this operand.
//   * only the ACTIVATIONS go through LDS (global_load_lds, the pipe kernel's swizzled 128-byte rows): BM x 128 B per k-tile, so
//     a ring slot is 8 - 16 KiB instead of 28 - 48 (L + 1 = 3 slots per k-group in production: two workgroups fit a CU).
//   * wave tile = BM rows x 32 columns (TM = BM / 32 accumulator tiles): TM ds_read_b128 per TM MFMAs, every wave of a group on
//     its own SIMD with the same work (no 6-waves-on-4-SIMDs imbalance).
//   * TWO k-groups per workgroup (waves 0-3: even k-tiles, waves 4-7: odd k-tiles of the SAME output tile): two waves per SIMD
//     that share nothing but the barrier -- what "two workgroups per CU" would give, on grids of < 256 workgroups that can never
//     be two per CU.  The groups' fp32 partial sums meet once, through LDS, in the fixed order (even + odd): bit-reproducible,
//     independent of BM and of the batch (the selection rule is static per layer, so an entry is bit-identical alone or batched).
// Hand-kept invariants (the compiler sees neither the asm loads nor the DMA completion):
//   * VMEM queue of a wave, per (virtual) iteration i: W(i + L) x 4, then the wave's PP activation pieces of tile i + L, in that
//     order, iff tile i + L exists.  Returns are in order, so every wait is a constant: with d = tiles left in the group,
//     W(j) has landed at  vmcnt(PP + min(L - 1, d - 1) * U),  the wave's pieces of tile j + 1 at  vmcnt(min(L - 1, max(d - 2, 0)) * U),
//     U = PP + 4.
//   * slot / register stage of tile j = j % (L + 1); tile j + L goes into the stage of tile j - 1, which the barrier of
//     iteration j - 1 (activations) and the last MFMAs of iteration j - 1 (weights) have freed.
//   * one s_barrier per iteration, between the third and the fourth kk-step: before it the wave's own pieces of tile j + 1 have
//     landed and all its reads of tile j are complete (lgkmcnt(0)).
// tools/asm_lint.py checks the asm-load rule (no instruction touches the destination of a pending asm global_load before a
// counted vmcnt wait covers it) on the device assembly of this file.
#include "igemm_common.h"
#include <atomic>

namespace sdxl {

// weight fragment load.  Inline asm: the compiler's own vmcnt bookkeeping of a plain load is exact in straight-line code only -- at
// the loop header it falls back to vmcnt(0) in front of the first MFMA, which serialises the whole prefetch -- so these loads
// are waited for by hand like the DMA pieces (counts in the file header; tools/asm_lint.py replays the queue over the assembly).
template <int OFF> __device__ __forceinline__ half8 wreg_gload128(const half_t* ptr) {
  half8 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(ptr), "n"(OFF));
  return v;
}

// [Npad][Kpad] row-major packed weights (f16, k contiguous) -> fragment order: 16-byte unit ((nb * nk + kt) * 4 + kk) * 64 + lane
// = row nb * 32 + (lane & 31), 16-byte chunk kt * 8 + kk * 2 + (lane >> 5)
__global__ void repack_wfrag_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int nblocks, int nk) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nblocks * nk * 256;
  if (i >= total) return;
  const int lane = (int)(i & 63), kk = (int)((i >> 6) & 3);
  const size_t t = i >> 8;
  const int kt = (int)(t % nk);
  const int nb = (int)(t / nk);
  dst[i] = src[(size_t)(nb * 32 + (lane & 31)) * (nk * 8) + kt * 8 + kk * 2 + (lane >> 5)];
}
void launch_repack_wfrag(const void* w, void* wf, int Npad, int Kpad, hipStream_t s) {
  const int nblocks = Npad / 32, nk = Kpad / 64;
  const size_t total = (size_t)nblocks * nk * 256;
  hipLaunchKernelGGL(repack_wfrag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(w),
                     reinterpret_cast<f32x4*>(wf), nblocks, nk);
}

// direct row-per-lane epilogue of a wave's share of a BM x 32 wave tile.  BOTH k-groups finalize: group t of the wave pair that
// computed columns nw .. nw + 31 (even / odd k-tiles) takes the 16 columns nw + 16 t .. + 15 -- the accumulator groups q = 2t, 2t + 1, whose
// other-group partial sums it has just added -- one v_permlane32_swap per value pair leaves a lane 8 consecutive columns of its row:
// one 16-byte residual load and one 16-byte store per 32-row block.  Row statistics for the next folded LayerNorm: a 64-column slot is
// held by FOUR waves here (columns: wave pair (w & ~1, w | 1); halves of 16: the two groups), so the slot's pivot (its first stored
// value: even wave, group 0) and three partial sums cross through LDS.  The arithmetic -- per 8-column piece shifted sums, pieces
// paired across the lane halves, then ((p0+p1) + (p2+p3)) + ((p4+p5) + (p6+p7)) -- is the tree of igemm_epilogue_rows / the staged
// epilogue, so the (mean, M2) bits do not depend on which kernel produced the rows.  Called by all 8 waves (two barriers when p.stat_out).
// the epilogue's operands that do not depend on the accumulators -- residual rows, bias -- are requested BEFORE the partial sums
// cross LDS (~1.7 k cycles of barriers and LDS traffic that would otherwise precede an exposed L2 / Infinity-Cache round trip)
template <int TM> struct WregEpiOperands { half8 rh[TM]; f32x4 rf[TM][2]; f32x4 bz0, bz1, sg0, sg1; };
template <int TM>
__device__ __forceinline__ void wreg_epilogue_request(const IgemmParams& p, int mw, int nw, int lane, int t, const void* zeros, WregEpiOperands<TM>& op) {
  const int fr = lane & 31, fh = lane >> 5;
  const f32x4* zv = reinterpret_cast<const f32x4*>(zeros);
  const int n0 = nw + t * 16 + 8 * fh;
  const bool r16 = p.R && p.r_dt == DT_F16, r32 = p.R && p.r_dt == DT_F32;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mw + i * 32 + fr;
    const bool ok = m < p.M;
    const size_t o = (size_t)(ok ? m : 0) * p.ldr + n0;
    op.rh[i] = *((r16 && ok) ? reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.R) + o) : reinterpret_cast<const half8*>(zeros));
    if (r32) {
      op.rf[i][0] = *(ok ? reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + o) : zv);
      op.rf[i][1] = *(ok ? reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + o + 4) : zv);
    }
  }
  // (no time-embedding bias here: it rides on the 3x3 conv_in of a ResBlock only -- the launcher refuses ebias)
  op.bz0 = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nw + 16 * t + 4 * fh) : zv);
  op.bz1 = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nw + 16 * t + 8 + 4 * fh) : zv);
  // f16 shadow of an fp32 stream (IgemmParams::shadow): the next LayerNorm's gamma over this lane's 8 output columns
  op.sg0 = *(p.shadow ? reinterpret_cast<const f32x4*>(p.shadow_gamma + n0) : zv);
  op.sg1 = *(p.shadow ? reinterpret_cast<const f32x4*>(p.shadow_gamma + n0 + 4) : zv);
}
template <int TM>
__device__ __forceinline__ void wreg_epilogue(const IgemmParams& p, const f32x16 (&acc)[TM], int mw, int nw, int lane, int w, int t,
                                              float* xch, const WregEpiOperands<TM>& op) {
  const int fr = lane & 31, fh = lane >> 5;
  float sv[TM][8];      // what the row statistics are taken of: the stored values (f16 outputs: the rounded ones the consumer will read; fp32 rows as they are)
  int m[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) { m[i] = mw + i * 32 + fr; mok[i] = m[i] < p.M; }
  const int n0 = nw + t * 16 + 8 * fh;            // this lane's 8 consecutive output columns
  {
    const bool r16 = p.R && p.r_dt == DT_F16, r32 = p.R && p.r_dt == DT_F32;
    const half8 (&rh)[TM] = op.rh;
    const f32x4 (&rf)[TM][2] = op.rf;
    const f32x4 bz0 = op.bz0, bz1 = op.bz1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f32x4 v0, v1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {               // accumulator groups q = 2t and 2t + 1 (wave-uniform t: a select, not an index)
        v0[r] = t ? acc[i][8 + r] : acc[i][r];
        v1[r] = t ? acc[i][12 + r] : acc[i][4 + r];
      }
      v0 = v0 + bz0; v1 = v1 + bz1;
      float wv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0[r]), __float_as_uint(v1[r]), false, false);
        wv[r] = __uint_as_float(sw[0]);
        wv[4 + r] = __uint_as_float(sw[1]);
      }
      if (r16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] += (float)rh[i][e];
      } else if (r32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[e] += rf[i][0][e]; wv[4 + e] += rf[i][1][e]; }
      }
      if (p.c_dt == DT_F16) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (half_t)wv[e];
        if (mok[i]) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.C) + (size_t)m[i] * p.ldc + n0) = h;
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[i][e] = (float)h[e];
      } else {
        if (mok[i]) {
          float* cp = reinterpret_cast<float*>(p.C) + (size_t)m[i] * p.ldc + n0;
          *reinterpret_cast<f32x4*>(cp) = f32x4{wv[0], wv[1], wv[2], wv[3]};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{wv[4], wv[5], wv[6], wv[7]};
        }
        if (p.shadow && p.shadow_lo_scale < 0.f) {      // (uniform) HL16 shadow: (hi, lo) of x * gamma in the split-operand layout, read as f16 by the GEMM behind the next LayerNorm
          float xs[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { xs[e] = wv[e] * op.sg0[e]; xs[4 + e] = wv[4 + e] * op.sg1[e]; }
          if (mok[i]) store_hl8(reinterpret_cast<float*>(p.shadow) + (size_t)m[i] * p.shadow_ld, n0, xs);
        } else
        if (p.shadow) {      // (kernel argument: uniform) f16(x * gamma of the next LayerNorm): the operand of the GEMM behind it
          half8 hs;
#pragma unroll
          for (int e = 0; e < 4; ++e) { hs[e] = (half_t)(wv[e] * op.sg0[e]); hs[4 + e] = (half_t)(wv[4 + e] * op.sg1[e]); }
          if (mok[i]) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.shadow) + (size_t)m[i] * p.shadow_ld + n0) = hs;
          if (p.shadow_lo_scale > 0.f) {      // (uniform) the lo halves of the same products behind the N hi columns
            half8 hl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x0 = wv[e] * op.sg0[e], x1 = wv[4 + e] * op.sg1[e];
              asm("" : "+v"(x0)); asm("" : "+v"(x1));      // the SAME rounded products the hi halves came from
              hl[e] = (half_t)((x0 - (float)(half_t)x0) * p.shadow_lo_scale); hl[4 + e] = (half_t)((x1 - (float)(half_t)x1) * p.shadow_lo_scale);
            }
            if (mok[i]) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.shadow) + (size_t)m[i] * p.shadow_ld + p.N + n0) = hl;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[i][e] = wv[e];
      }
    }
  }
  if (!p.stat_out) return;          // (kernel argument: uniform over the workgroup)
  // ---- row statistics of the stored (rounded) values, slot = the 64 columns of the wave pair (w & ~1, w | 1) x both groups
  const int pair = w >> 1;
  const int part = (w & 1) * 2 + t;                      // position of this wave's 16 columns in the slot: pieces 2 part, 2 part + 1
  float* xpiv = xch + pair * (TM * 32);                  // [pair][row]
  float* xsum = xch + 2 * (TM * 32) + pair * (3 * TM * 64);   // [pair][part - 1][row][2]
  float piv[TM];
  if (part == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      piv[i] = __shfl(sv[i][0], fr);              // the slot's first stored value (lanes 0..31 of the first wave hold it)
      if (fh == 0) xpiv[i * 32 + fr] = piv[i];
    }
  }
  wait_lgkmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float A1[TM], A2[TM];
  if (part != 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) piv[i] = xpiv[i * 32 + fr];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = sv[i][e] - piv[i]; s1 += d; s2 = fmaf(d, d, s2); }
    A1[i] = s1 + __shfl_xor(s1, 32); A2[i] = s2 + __shfl_xor(s2, 32);        // pieces 2 part (lanes 0..31) + 2 part + 1 (lanes 32..63)
    if (part != 0 && fh == 0) { xsum[((part - 1) * TM * 32 + i * 32 + fr) * 2] = A1[i]; xsum[((part - 1) * TM * 32 + i * 32 + fr) * 2 + 1] = A2[i]; }
  }
  wait_lgkmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (part == 0 && fh == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r2 = (i * 32 + fr) * 2;
      const float s1 = (A1[i] + xsum[r2]) + (xsum[TM * 64 + r2] + xsum[2 * TM * 64 + r2]);
      const float s2 = (A2[i] + xsum[r2 + 1]) + (xsum[TM * 64 + r2 + 1] + xsum[2 * TM * 64 + r2 + 1]);
      if (mok[i]) {
        float* dst = p.stat_out + ((size_t)(nw >> 6) * p.M + m[i]) * 2;
        dst[0] = piv[i] + s1 * (1.0f / 64.0f);
        dst[1] = fmaxf(s2 - s1 * s1 * (1.0f / 64.0f), 0.f);
      }
    }
  }
}

#ifdef SDXL_MEASURE
// coarse s_memtime stamps (tools/wreg_timeline.py): [workgroup][wave][8] = entry, prologue issued, tile 0 landed, k-loop done, partial
// sums exchanged, epilogue issued, stores drained; word 7 = shader cycles per 100 MHz tick x 100 (clock) is derived by the tool from
// words 6 / 7 = s_memrealtime at entry / exit
__device__ unsigned* g_wreg_tl = nullptr;
void igemm_set_wreg_timeline(void* buf) {
  unsigned* b = reinterpret_cast<unsigned*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wreg_tl), &b, sizeof(b)) != hipSuccess) throw std::runtime_error("igemm_wreg: cannot set the timeline buffer");
}
#define WREG_STAMP(i) do { wtl[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WREG_STAMP(i) do { } while (0)
#endif
// MODE (measure builds, forced variants 63 ..): 0 production; knock-outs that time one resource alone (results are garbage):
// 1 no MFMAs, 2 operand pointers frozen (every fetch after the first hits the L1 / L2: same instruction stream, no fabric traffic),
// 3 no VMEM at all in the k-loop, 4 no k-loop barriers; 5 = production arithmetic with the other XCD ownership (row tiles);
// 6 weight stream only (no activation pieces), 7 activation pieces only (no weight loads), 8 activation pieces read as CONTIGUOUS
// 1-KiB runs (what a k-tile-major activation layout would give) instead of 8 rows x 128 B at the row stride -- waits stay exact in 6 / 7
// MODE >= 16: a bit set of the same knock-outs (16 no MFMAs, 32 frozen pointers, 64 no VMEM, 128 no barriers, 256 other XCD ownership,
// 512 no activation pieces, 1024 no weight loads, 2048 contiguous activation pieces, 4096 no LDS fragment reads)
constexpr int wreg_mode_flags(int m) {
  return m >= 16 ? m : m == 1 ? 16 : m == 2 ? 32 : m == 3 ? 64 : m == 4 ? 128 : m == 5 ? 256 : m == 6 ? 512 : m == 7 ? 1024 : m == 8 ? 2048 : 0;
}
template <int BM, int L, int MODE = 0>
__global__ __launch_bounds__(512) void igemm_wreg_kernel(const IgemmParams p, const void* zeros) {
#ifdef SDXL_MEASURE
  unsigned wtl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  wtl[6] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
  WREG_STAMP(0);
  kernarg_prefetch<(int)sizeof(IgemmParams) + 8>();
  constexpr int NG = 2;                       // k-groups per workgroup
  constexpr int NSG = L + 1;                  // ring slots / weight register stages per group
  constexpr int TM = BM / 32;                 // accumulator tiles per wave (wave tile BM x 32)
  constexpr int PP = BM / 32;                 // activation DMA pieces (8 rows x 128 B) per wave and k-tile
  constexpr int FL = wreg_mode_flags(MODE);
  constexpr bool NOMFMA = FL & 16, FROZEN = FL & 32, NOVMEM = FL & 64, NOBAR = FL & 128, XCDALT = FL & 256, NOA = FL & 512, NOWL = FL & 1024,
                 ACONTIG = FL & 2048, NOLDS = FL & 4096;
  constexpr int NW = NOWL ? 0 : 4;            // weight fragment loads per wave and k-tile
  constexpr int NA = NOA ? 0 : PP;            // activation pieces the wave issues per k-tile
  constexpr int U = NA + NW;                  // VMEM operations per wave and k-tile
  constexpr int NOPS = U;
  constexpr int SLOT = BM * 128;              // bytes per ring slot
  constexpr int RING = NSG * SLOT;            // bytes per group
  constexpr int KSTEP = 64 * NG;              // elements between a group's consecutive k-tiles
  static_assert(BM % 32 == 0 && TM >= 2 && TM <= 4, "wave tile: 64 / 96 / 128 rows");
  static_assert((NSG - 1) * SLOT + (TM - 1) * 4096 < 65536, "fragment offsets must fit the ds_read immediate");
  static_assert(PP + (L - 1) * U < 64 && L >= 2, "vmcnt is 6 bits");
  static_assert(NOPS <= 3 * TM, "one VMEM operation per MFMA gap before the rendezvous");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w = wave & 3;

  const int tilesM = (p.M + BM - 1) / BM;
  const int nwg = tilesM * (p.N >> 7);
  // workgroups behind the tile grid (CUs this launch would leave idle) only read the weights of a later GEMM (IgemmParams::warm)
  if ((int)blockIdx.x >= nwg) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (p.warm[r]) igemm_warm_body<512>(p.warm[r], p.warm_bytes[r], (int)blockIdx.x - nwg, (int)gridDim.x - nwg);
    return;
  }
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // an XCD (private L2) owns a run of consecutive remapped ids = a few ROW tiles and sweeps every weight column tile under them:
  // with M >= N (every shape this kernel is selected for) that is the ownership that brings fewer bytes into an XCD's L2 --
  // (M / 8 + N) K instead of (N / 8 + M) K -- and the weights a second XCD asks for come out of the memory-side Infinity Cache,
  // not out of HBM again.  Measured against column ownership (MODE 5) on the step's shapes, cold / warm weights: 16.0 / 14.2 vs
  // 16.4 / 15.3 us (out-projection), 42.0 / 37.3 vs 42.7 / 39.2 us (FF-out), 142.8 / 123.7 vs 147.8 / 131.5 us at K = 20480
  // (profiles/r04_wreg_knockout.txt).
  const int tilesN = p.N >> 7;
  int tm = bid / tilesN, tn = bid - tm * tilesN;
  if constexpr (XCDALT) { tn = bid / tilesM; tm = bid - tn * tilesM; }
  // round 5 A/B (knob wreg_xcd2d): 2-D ownership.  With whole row-tile runs an XCD streams EVERY weight column tile under its ~tilesM / 8 row tiles
  // (M = 2048 x N = 1280: 3 x 96 rows + 1280 columns of operands per XCD, 26 % of its fetches compulsory misses -- the 73 % L2 hit rate of DESIGN 10.2);
  // four patches of (tilesM / 2 row tiles) x (tilesN / 2 column tiles), two XCDs per patch in row-major order inside it, make that ~6 x 96 rows + 640 columns.
  if (p.wreg_xcd2d && !(tilesM & 1) && !(tilesN & 1) && tilesN >= 8) {
    const int ch = tilesN >> 1, prow = tilesM >> 1, ptiles = prow * ch;
    const int patch = bid / ptiles, idx = bid - patch * ptiles;
    const int r = idx / ch;
    tm = (patch >> 1) * prow + r;
    tn = (patch & 1) * ch + (idx - r * ch);
  }
  const int m0 = tm * BM, n0 = tn * 128;
  const int nk = p.Kpad >> 6;
  const int nkg = (nk - g + 1) >> 1;          // k-tiles of this group: g, g + 2, ...

  // ---- weight stream: this wave's 32 columns, fragment order, contiguous over k
  const half_t* wp = reinterpret_cast<const half_t*>(p.Wf) + ((size_t)((n0 >> 5) + w) * nk + g) * 2048 + lane * 8;
  half8 Wr[NSG][4];
  if constexpr (NOVMEM || NOWL) { static_for<NSG>([&](auto S_) { static_for<4>([&](auto K_) { Wr[decltype(S_)::value][decltype(K_)::value] = half8{1, 1, 1, 1, 1, 1, 1, 1}; }); }); }
  auto loadW = [&](auto S, auto KK) {
    constexpr int s = decltype(S)::value, kk = decltype(KK)::value;
    if constexpr (!NOVMEM && !NOWL) Wr[s][kk] = wreg_gload128<kk * 1024>(wp);
    if constexpr (kk == 3 && !FROZEN) wp += NG * 2048;
  };
  // ---- activation pieces: piece q of this wave = tile rows (4 q + w) * 8 .. + 7 of the group's k-tile; lane -> (row, 16-byte slot)
  const half_t* Ag = reinterpret_cast<const half_t*>(p.A);
  const half_t* aptr[PP];
  int aadv[PP];
  char* const ring = smem + g * RING;
  auto setupA = [&]() {
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const int row = (q * 4 + w) * 8 + (lane >> 3);
      const int m = m0 + row;
      const bool ok = m < p.M;
      aptr[q] = ok ? Ag + (size_t)m * p.lda + g * 64 + (((lane & 7) ^ ((row >> 1) & 7)) << 3) : reinterpret_cast<const half_t*>(zeros);
      aadv[q] = ok ? KSTEP : 0;
      if constexpr (ACONTIG) {       // the same bytes per tile and the same sharing between the column tiles of a row tile, one run per piece
        aptr[q] = Ag + ((size_t)(tm % (p.M / BM)) * nk + g) * (BM * 64) + (q * 4 + w) * 512 + lane * 8;
        aadv[q] = NG * BM * 64;
      }
    }
  };
  auto pieceA = [&](auto S, auto Q) {
    constexpr int s = decltype(S)::value, q = decltype(Q)::value;
    if constexpr (!NOVMEM && !NOA) __builtin_amdgcn_global_load_lds((gptr_t)aptr[q], (lptr_t)(ring + s * SLOT + (q * 4 + w) * 1024), 16, 0, 0);
    if constexpr (!FROZEN) aptr[q] += aadv[q];
  };
  // operation x of a k-tile's VMEM sequence: 0..3 weight fragments, 4.. activation pieces
  auto vmem_op = [&](auto S, auto X) {
    constexpr int x = decltype(X)::value;
    if constexpr (x < NW) loadW(S, X);
    else pieceA(S, std::integral_constant<int, x - NW>{});
  };

  const int fr = lane & 31, fh = lane >> 5;
  unsigned fa[4];
  {
    const unsigned basea = lds0 + g * RING + fr * 128 + ((fh ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa[kk] = basea ^ (kk << 5);
  }
  half8 fA[2][TM];
  if constexpr (NOLDS) { static_for<TM>([&](auto I) { fA[0][decltype(I)::value] = fA[1][decltype(I)::value] = half8{1, 1, 1, 1, 1, 1, 1, 1}; }); }
  auto ldsA = [&](auto SET, auto S, auto KK) {
    constexpr int set = decltype(SET)::value, s = decltype(S)::value, kk = decltype(KK)::value;
    if constexpr (!NOLDS) static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<s * SLOT + decltype(I)::value * 4096>(fa[kk]); });
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  // Slot rotation.  Everything below is straight-line code with compile-time slots, register stages and wait counts: the LAST
  // tile of a group always sits in slot L, so the final L tiles ("drain": nothing left to issue, the waits shrink with the tiles
  // left) are one static sequence over slots 1 .. L, the steady tiles before them are whole passes over slots 1, 2, .., L, 0
  // behind one partial pass, and only the prologue (and that partial pass) is entered by the run-time first slot s0.
  const int s0 = ((L - (nkg - 1)) % NSG + NSG) % NSG;         // slot of tile 0
  const int ns = nkg > L ? nkg - L : 0;                       // steady tiles (tile j + L exists)

  // ---- prologue = virtual iterations -L .. -1: W(i), pieces(i) for the first L tiles into slots s0 + i.  Tile 0's weights go
  // out before the activation geometry is worked out (inside a UNet step the weights are the operand that arrives cold from HBM).
  static_for<NSG>([&](auto C) {
    constexpr int c = decltype(C)::value;
    if (s0 == c && nkg > 0) static_for<4>([&](auto KK) { loadW(C, KK); });
  });
  setupA();
  static_for<NSG>([&](auto C) {
    constexpr int c = decltype(C)::value;
    if (s0 == c) {
      static_for<L>([&](auto I) {
        constexpr int i = decltype(I)::value;
        using SI = std::integral_constant<int, (c + i) % NSG>;
        if (i < nkg) {
          if constexpr (i > 0) static_for<4>([&](auto KK) { loadW(SI{}, KK); });
          static_for<PP>([&](auto Q) { pieceA(SI{}, Q); });
        }
      });
    }
  });
  __builtin_amdgcn_sched_barrier(0);
  WREG_STAMP(1);
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // rendezvous of virtual iteration -1: own pieces of tile 0 landed (tiles 1 .. min(L, nkg) - 1 may stay in flight), then everybody's
  if (nkg >= L) wait_vmcnt<(L - 1) * U>();
  else static_for<L - 1>([&](auto D) { constexpr int dd = decltype(D)::value + 1; if (nkg == dd) wait_vmcnt<(dd - 1) * U>(); });
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  static_for<NSG>([&](auto C) { if (s0 == decltype(C)::value && nkg > 0) ldsA(I0{}, C, I0{}); });
  __builtin_amdgcn_sched_barrier(0);
  WREG_STAMP(2);

  // TM MFMAs of kk-step KK from fragment set SET; steady tiles put VMEM operations X0 .. of tile j + L behind MFMAs 0, 1, ...
  auto mma = [&](auto SET, auto S, auto KK, auto SF, auto X0) {
    constexpr int set = decltype(SET)::value, s = decltype(S)::value, kk = decltype(KK)::value, x0 = decltype(X0)::value;
    static_for<TM>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (!NOMFMA) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wr[s][kk], fA[set][i], acc[i], 0, 0, 0);
      else asm volatile("" ::"v"(Wr[s][kk]), "v"(fA[set][i]));      // (keeps the in-flight destination registers allocated up to here)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (x0 >= 0 && x0 + i < NOPS) {
        vmem_op(SF, std::integral_constant<int, x0 + i>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  using NOX = std::integral_constant<int, -1>;
  // one k-tile in slot S.  D = 0: steady (more than L tiles left: tile j + L is issued, constant waits); D = 1 .. L: drain tile
  // with D tiles left, itself included
  auto ktile = [&](auto S, auto DD) {
    constexpr int c = decltype(S)::value, cn = (c + 1) % NSG, cf = (c + L) % NSG, D = decltype(DD)::value;
    constexpr bool steady = D == 0;
    using SF = std::integral_constant<int, cf>;
    using SN = std::integral_constant<int, cn>;
    using X0 = std::integral_constant<int, steady ? 0 : -1>;
    using X1 = std::integral_constant<int, steady ? TM : -1>;
    using X2 = std::integral_constant<int, steady ? 2 * TM : -1>;
    ldsA(I1{}, S, I1{});
    wait_vmcnt<steady ? NA + (L - 1) * U : NA + (D - 1) * U>();        // W(j) has landed
    wait_lgkmcnt<TM>();
    mma(I0{}, S, I0{}, SF{}, X0{});
    ldsA(I0{}, S, I2{});
    wait_lgkmcnt<TM>();
    mma(I1{}, S, I1{}, SF{}, X1{});
    ldsA(I1{}, S, I3{});
    wait_lgkmcnt<TM>();
    mma(I0{}, S, I2{}, SF{}, X2{});
    if constexpr (steady || D > 1) {
      wait_vmcnt<steady ? (L - 1) * U : (D - 2) * U>();                // own pieces of tile j + 1 have landed
      wait_lgkmcnt<0>();
      if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ldsA(I0{}, SN{}, I0{});
    } else {
      wait_lgkmcnt<0>();
    }
    mma(I1{}, S, I3{}, SF{}, NOX{});
  };
  using STEADY = std::integral_constant<int, 0>;
  if (ns > 0) {
    // first (partial) pass: slots s0, s0 + 1, .., L, 0 -- then whole passes 1, .., L, 0
    const int first = s0 == 0 ? 1 : NSG - s0 + 1;
    static_for<L>([&](auto C) { constexpr int c = decltype(C)::value + 1; if (s0 != 0 && s0 <= c) ktile(std::integral_constant<int, c>{}, STEADY{}); });
    ktile(I0{}, STEADY{});
    for (int pass = (ns - first) / NSG; pass > 0; --pass) {
      static_for<L>([&](auto C) { ktile(std::integral_constant<int, decltype(C)::value + 1>{}, STEADY{}); });
      ktile(I0{}, STEADY{});
    }
  }
  // drain: D = L .. 1 tiles left, slots 1 .. L (groups with fewer than L tiles enter in the middle)
  static_for<L>([&](auto X) {
    constexpr int D = L - decltype(X)::value;
    if (D <= nkg) ktile(std::integral_constant<int, L + 1 - D>{}, std::integral_constant<int, D>{});
  });
  WREG_STAMP(3);
  // odd k-tile count: group 1 has one tile -- and one rendezvous -- less; barrier counts must match across the workgroup
  if constexpr (!NOBAR) { if ((nk & 1) && g == 1) __builtin_amdgcn_s_barrier(); }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // No rendezvous here: the LAST tile of a group always sits in slot L, and every wave has passed the last k-loop rendezvous (all reads of
  // the other slots complete in front of it), so slots 0 .. L - 1 of BOTH rings are dead while slower waves still read their slot L --
  // the exchange area lives in those slots.

  // ---- the two groups' partial sums meet in LDS.  Each group finalizes HALF of every wave tile's columns (group t: accumulator
  // groups q = 2t, 2t + 1), so it parks the other half for its partner (lane-linear 16-byte pieces) and adds the partner's half
  // to its own: even k-tiles + odd k-tiles whichever group does the add (fp32 addition commutes bit for bit)
  WregEpiOperands<TM> eop;
  wreg_epilogue_request<TM>(p, m0, n0 + w * 32, lane, g, zeros, eop);
  static_assert(4 * TM * 2 * 64 * 16 <= L * SLOT, "a group's four exchange pieces must fit its dead slots 0 .. L - 1");
  f32x4* xw = reinterpret_cast<f32x4*>(smem + g * RING) + (size_t)w * (TM * 2 * 64) + lane;                    // written by (g, w): own group's slots 0 ..
  const f32x4* xo = reinterpret_cast<const f32x4*>(smem + (1 - g) * RING) + (size_t)w * (TM * 2 * 64) + lane;  // partner's
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = g ? acc[i][4 * qq + r] : acc[i][8 + 4 * qq + r];     // group 1 parks q = 0, 1; group 0 parks q = 2, 3
      xw[(i * 2 + qq) * 64] = v;
    }
  wait_lgkmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const f32x4 o = xo[(i * 2 + qq) * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (g) acc[i][8 + 4 * qq + r] = o[r] + acc[i][8 + 4 * qq + r];      // (even-tile partial + odd-tile partial in both groups)
        else acc[i][4 * qq + r] = acc[i][4 * qq + r] + o[r];
      }
    }
  WREG_STAMP(4);
  static_assert((2 * BM + 6 * 2 * BM) * 4 <= SLOT, "statistics exchange must fit one slot");
  float* xch = reinterpret_cast<float*>(smem + L * SLOT);           // statistics exchange: slot L of ring 0 (dead behind the exchange rendezvous)
  wreg_epilogue<TM>(p, acc, m0, n0 + w * 32, lane, w, g, xch, eop);
#ifdef SDXL_MEASURE
  WREG_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wtl[7] = (unsigned)__builtin_amdgcn_s_memrealtime();
  { const unsigned t6 = (unsigned)__builtin_amdgcn_s_memtime();
    if (g_wreg_tl && lane == 0) { unsigned* d = g_wreg_tl + ((size_t)blockIdx.x * 8 + wave) * 16; for (int i = 0; i < 8; ++i) d[i] = wtl[i]; d[8] = t6; } }
#endif
}

// ---------------------------------------------------------------------------------------------------------
static std::atomic<int> g_wreg_enable{1};
void igemm_set_wreg(int v) { g_wreg_enable = v; }
static std::atomic<int> g_wreg_xcd2d{0};
void igemm_set_wreg_xcd2d(int v) { g_wreg_xcd2d = v; }

// warming workgroups of a launch: the CU slots its tile grid leaves empty in its (single) round -- one workgroup per CU up to 256 tiles,
// two up to 512 -- at most 64 (a warmer pulls ~30 GB/s out of HBM: 36 of them move 13 MB inside an out-projection's 15 us)
static int wreg_warm_groups(const IgemmParams& p, int ntiles) {
  if (!p.warm[0] || !p.warm_bytes[0] || ntiles > 512) return 0;
  const int spare = (ntiles <= 256 ? 256 : 512) - ntiles;
  return spare < 8 ? 0 : spare > 64 ? 64 : spare;
}
template <int BM, int L, int MODE = 0>
static void launch_wreg_t(const IgemmParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (L + 1) * BM * 128;
  static_assert(lds >= (size_t)8 * (BM / 32) * 2048 + (size_t)(2 * BM + 6 * 2 * BM) * 4, "exchange areas must fit the dead rings");
  static bool attr_set[kIgemmMaxDev] = {};
  const int dev = igemm_current_device();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wreg_kernel<BM, L, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      throw std::runtime_error("igemm_wreg: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    attr_set[dev] = true;
  }
  const int tilesM = (p.M + BM - 1) / BM, tilesN = p.N / 128;
  const int ntiles = tilesM * tilesN;
  IgemmParams q = p;
  q.wreg_xcd2d = g_wreg_xcd2d.load();
  hipLaunchKernelGGL((igemm_wreg_kernel<BM, L, MODE>), dim3(ntiles + wreg_warm_groups(p, ntiles)), dim3(512), lds, s, q, igemm_zero_page());
}

// shapes this kernel takes: plain f16 linear layers / 1x1 convolutions whose weights were also packed in fragment order
bool igemm_wreg_ok(const IgemmParams& p) {
  if (!p.Wf || !igemm_zero_page()) return false;
  if (p.a_dt != DT_F16 || (p.c_dt != DT_F16 && p.c_dt != DT_F32)) return false;
  if (p.ksize != 1 || p.stride != 1 || p.up != 0 || p.pad != 0 || p.Hin != p.Hout || p.Win != p.Wout) return false;
  if (p.N % 128 != 0 || p.Cin != p.K || p.K != p.Kpad || p.Kpad % 64 != 0 || p.Kpad < 128) return false;
  if (p.act != 0 || p.n_split < p.N || p.xa_k || p.gn_part || p.ln_stat || p.acc_scale) return false;
  if ((p.lda & 7) != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15) != 0) return false;
  if ((p.ldc & 7) != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0) return false;
  if (p.R && ((p.ldr & 7) != 0 || (reinterpret_cast<uintptr_t>(p.R) & 15) != 0)) return false;
  if (p.ebias) return false;
  // f16 shadow of an fp32 output (+ the fp32 rows' statistics): whole 16-byte pieces
  if (p.shadow && (p.c_dt != DT_F32 || !p.shadow_gamma || (p.shadow_ld & (p.shadow_lo_scale < 0.f ? 15 : 7)) != 0 || (reinterpret_cast<uintptr_t>(p.shadow) & 15) != 0)) return false;
  return true;
}
// variant 0: rows per tile from the grid it makes on 256 CUs (the k-summation order does not depend on it); 60 / 62 force 96 / 64
// rows.  (A 128-row tile -- 64 accumulators + 4 weight stages + 2 x 4 fragments -- spilled fragment registers that were still in
// flight and was 30 - 75 % slower than the 96-row tile on every shape of the step: removed, profiles/r04_wreg_first_ab.txt.)
static std::atomic<int> g_warm_enable{1};
void igemm_set_warm(int v) { g_warm_enable = v; }
int igemm_warm_enabled() { return g_warm_enable.load(); }
bool igemm_wreg_selected(const IgemmParams& p) {
  if (!igemm_wreg_ok(p) || !g_wreg_enable.load()) return false;
  const long rows = 2L * (p.rpb > 0 ? p.rpb : p.M);
  return ((rows + 95) / 96) * (long)(p.N / 128) <= 512;
}
bool launch_igemm_wreg(const IgemmParams& p, int variant, hipStream_t s) {
  if (!igemm_wreg_ok(p)) return false;
  if (variant == 0) {
    if (!g_wreg_enable.load()) return false;
    // where it pays (profiles/r04_wreg_first_ab.txt, r04_wreg_ab_l2.txt): grids of at most TWO 96-row tiles per CU -- with the
    // shallow prefetch (L = 2: 72 KiB of LDS) two workgroups share a CU, so the 64^2 level's 430 tiles are one resident round
    // (out-projection 19.5 -> 16.9 us, FF-out 44.5 -> 41.6 us, 1920 -> 640 skip 36.1 -> 33.1 us, cold weights) like the 32^2 level's 220
    // (18.2 -> 15.0, 47.8 -> 41.2 us).  Evaluated on the CFG PAIR's shape (2 entries of rpb rows) whatever the actual batch: the two
    // structures sum k in different orders, and an entry must come out bit-identical alone or batched.
    const long rows = 2L * (p.rpb > 0 ? p.rpb : p.M);
    if (((rows + 95) / 96) * (long)(p.N / 128) > 512) return false;
  }
#ifdef SDXL_MEASURE
  if (variant == 68) { launch_wreg_t<96, 3>(p, s); return true; }     // prefetch-depth A/B partners of the 96-row kernel (L = 2 in production)
  if (variant == 69) { launch_wreg_t<96, 4>(p, s); return true; }
  if (variant == 61) { launch_wreg_t<128, 2>(p, s); return true; }    // 128-row tile (64 accumulators) at the shallow depth
  if (variant >= 63 && variant <= 67) {     // knock-out timing modes of the 96-row kernel (garbage results except 67)
    if (variant == 63) launch_wreg_t<96, 2, 1>(p, s); else if (variant == 64) launch_wreg_t<96, 2, 2>(p, s);
    else if (variant == 65) launch_wreg_t<96, 2, 3>(p, s); else if (variant == 66) launch_wreg_t<96, 2, 4>(p, s);
    else launch_wreg_t<96, 2, 5>(p, s);
    return true;
  }
  if (variant == 70) { launch_wreg_t<96, 2, 6>(p, s); return true; }     // per-stream knock-outs: weights only / activations only / contiguous activations
  if (variant == 71) { launch_wreg_t<96, 2, 7>(p, s); return true; }
  if (variant == 72) { launch_wreg_t<96, 2, 8>(p, s); return true; }
  if (variant == 73) { launch_wreg_t<96, 2, 16 | 32>(p, s); return true; }                 // fetch stream + LDS reads, every byte an L2 hit
  if (variant == 74) { launch_wreg_t<96, 2, 16 | 32 | 4096>(p, s); return true; }          // ... without the LDS reads
  if (variant == 75) { launch_wreg_t<96, 2, 16 | 32 | 4096 | 128>(p, s); return true; }    // ... and without the rendezvous
  if (variant == 76) { launch_wreg_t<96, 2, 16 | 4096>(p, s); return true; }               // fetch stream alone, real pointers
  if (variant == 77) { launch_wreg_t<96, 2, 4096>(p, s); return true; }                    // MFMAs + fetch stream, no LDS reads
#endif
  int bm = variant == 60 ? 96 : variant == 62 ? 64 : 0;
  if (!bm) {      // 64 rows where that is still a single round of one tile per CU (small M: 512^2 images, single entries), else 96
    const long t64 = (long)((p.M + 63) / 64) * (p.N / 128);
    bm = t64 <= 256 ? 64 : 96;
  }
  if (bm == 64) launch_wreg_t<64, 2>(p, s);
  else launch_wreg_t<96, 2>(p, s);
  return true;
}

}  // namespace sdxl
