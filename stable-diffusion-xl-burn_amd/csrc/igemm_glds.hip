// Fast-path implicit GEMM for gfx950: f16 operands, direct-to-LDS staging through an NS-deep ring.  PRODUCTION kernels only:
// the A/B partners, rolled loops, measurement modes and the loader-wave experiment live in igemm_measure.hip (measure builds).
//
// Same contract as igemm_kernel (igemm.hip) for the shapes that dominate the SDXL step (f16 activations, Cin % 64 == 0,
// 16-byte aligned rows); everything else stays on the generic kernel.  What every kernel in this file shares (CDNA4-specific):
//   * HBM/L2 -> LDS without a VGPR round trip: every wave issues `global_load_lds_dwordx4` (16 B per lane, 1 KiB per wave
//     instruction = 8 tile rows of 128 B).  The LDS image is lane-linear, so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address (chunk ^ f(row)) and again on the fragment read; halo / tail rows fetch from a zero page.
//     The conv gather (tap, stride, fused nearest-2x upsample) is still just a per-lane source address.
//   * NS-deep LDS ring with COUNTED `s_waitcnt vmcnt(N)` across a raw s_barrier; one barrier per k-tile.
//   * v_mfma_f32_32x32x16_f16 with the operand roles swapped (weights = MFMA A operand, activations = B operand): the
//     accumulator layout then gives each lane 4 CONSECUTIVE output columns of one row, and GEGLU pairs (x, gate) sit in the
//     same lane of the same accumulator tile.
//   * swizzle f(row) = (row>>1)&7 makes the ds_read_b128 fragment reads of 32 rows conflict-free across the four 16-lane
//     service groups (two 128-byte tile rows share one 256-byte bank row); measured SQ_LDS_BANK_CONFLICT = 0.
//   * XCD-aware block -> tile mapping; epilogue staged through LDS so loads/stores are whole row segments; folded LayerNorm
//     (igemm_common.h).
// Kernels, in file order (selection: launch_igemm_glds at the bottom; measurements: DESIGN.md section 3.1):
//   igemm_glds_kernel  4 waves, 2-slot ring, several co-resident blocks per CU           (forced fallback variants 4 / 6)
//   igemm_pipe_kernel  6 / 8 waves, hand-ordered k-loop unrolled by the ring depth: counted lgkmcnt / vmcnt, register-double-
//                      buffered fragments, DMA pieces between the MFMAs; tiles 256x128, 128x128, 96x128, 256x160, 128x160;
//                      f16, strict f32 (4 x v_mfma_f32_32x32x2_f32) and the fused cross-attention instantiations
//   igemm_wide_kernel  256x320 tile, k-tile 32                                           (GEGLU projections)
#include "igemm_common.h"
#include <atomic>

namespace sdxl {

template <int BM, int BN, int NS, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void igemm_glds_kernel(const IgemmParams p, const void* zeros) {
  kernarg_prefetch<(int)sizeof(IgemmParams) + 8>();   // every argument line in flight at once (one wait instead of five)   // >= MINB blocks per CU
  constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA tiles per wave
  constexpr int AJ = BM / 32, BJ = BN / 32;   // DMA instructions per wave per k-tile (8 rows each, 4 waves)
  constexpr int PER = AJ + BJ;                // DMA instructions per wave per stage
  constexpr int KT = 64;                      // f16 elements per k-tile = one 128-byte row
  constexpr int STAGE = (BM + BN) * 128;      // bytes per ring slot: A tile then B tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tilesN = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Each XCD (private 4 MiB L2) owns a contiguous run of remapped ids.  Walk that run so the LARGER operand is read from
  // HBM by one XCD only: weights bigger than activations (the M=2048 transformer GEMMs) -> an XCD owns a range of weight
  // column tiles and sweeps all row tiles; otherwise (convs at 64^2/128^2, VAE) it owns row tiles and sweeps the weights.
  const int tilesM = (p.M + BM - 1) / BM;
  int tm, tn;
  if ((size_t)p.N * p.K > (size_t)p.M * p.Cin) { tn = bid / tilesM; tm = bid - tn * tilesM; }
  else { tm = bid / tilesN; tn = bid - tm * tilesN; }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA geometry: instruction j of this wave covers tile rows (j*4 + wave)*8 .. +7; lane -> (row, slot)
  const int lrow = lane >> 3, slot = lane & 7;
  const int HWo = p.Hout * p.Wout;
  const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
  int rb[AJ], ry[AJ], rx[AJ], rsw[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;
    const int m = m0 + row;
    rsw[j] = (slot ^ ((row >> 1) & 7)) * 8;       // source chunk (elements) that lands in this lane's LDS slot
    if (m < p.M) {
      const int b = m / HWo;
      const int rem = m - b * HWo;
      const int oy = rem / p.Wout;
      rb[j] = b; ry[j] = oy * p.stride - p.pad; rx[j] = (rem - oy * p.Wout) * p.stride - p.pad;
    } else { rb[j] = -1; ry[j] = -(1 << 28); rx[j] = 0; }
  }
  const half_t* Ag = reinterpret_cast<const half_t*>(p.A);
  // Incremental DMA source pointers: stage() is called for k-tiles 0,1,2,... in order, so the per-lane source address of
  // every tile row is a running pointer that advances by 64 elements per k-tile and is recomputed (bounds test, pixel
  // address) only when the k-tile crosses into the next filter tap -- no per-tile integer division or 64-bit multiply.
  const half_t* wptr[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;
    wptr[j] = reinterpret_cast<const half_t*>(p.W) + (size_t)(n0 + row) * p.Kpad + (slot ^ ((row >> 1) & 7)) * 8;
  }
  const half_t* aptr[AJ];
  int aadv[AJ];
  int s_c0 = 0, s_dy = 0, s_dx = 0;      // wave-uniform tap walk state
  auto retap = [&]() {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int iy = ry[j] + s_dy, ix = rx[j] + s_dx;
      const bool ok = (unsigned)iy < (unsigned)Hup && (unsigned)ix < (unsigned)Wup;   // rows beyond M carry iy << 0
      const size_t off = (((size_t)(rb[j] < 0 ? 0 : rb[j]) * p.Hin + ((ok ? iy : 0) >> p.up)) * p.Win + ((ok ? ix : 0) >> p.up)) * p.lda + rsw[j];
      aptr[j] = ok ? Ag + off : reinterpret_cast<const half_t*>(zeros);
      aadv[j] = ok ? KT : 0;
    }
  };
  retap();

  auto stage = [&](int buf) {
    char* la = smem + buf * STAGE + wave * 1024;
    char* lb = la + BM * 128;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(la + j * 4096), 16, 0, 0);
      aptr[j] += aadv[j];
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(lb + j * 4096), 16, 0, 0);
      wptr[j] += KT;
    }
    s_c0 += KT;
    if (s_c0 == p.Cin) {                 // next k-tile starts a new tap (uniform branch)
      s_c0 = 0;
      if (++s_dx == p.ksize) { s_dx = 0; ++s_dy; }
      retap();
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.Kpad / KT;
  const int fr = lane & 31, fh = lane >> 5;
  // prologue: NS-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s);
  float lnA[TM], lnC[TM];
  ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  int cur = 0;                 // ring slot of tile kt
  int nxt = NS - 1;            // ring slot tile kt+NS-1 goes to (= slot of tile kt-1)
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; tiles kt+1 .. kt+NS-2 may stay in flight (only if they were really issued)
    if (kt + NS - 2 < nk) wait_vmcnt<PER * (NS - 2)>(); else wait_vmcnt<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();          // all waves: tile kt visible, compute(kt-1) finished -> slot `nxt` is free
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) stage(nxt);
    const char* a = smem + cur * STAGE;
    const char* b = a + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = kk * 2 + fh;
      half8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 32 + fr;
        fa[i] = *reinterpret_cast<const half8*>(a + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 32 + fr;
        fb[j] = *reinterpret_cast<const half8*>(b + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)   // weights as the A operand (rows = n), activations as B (cols = m)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    nxt = cur;
    cur = cur + 1 == NS ? 0 : cur + 1;
  }

  __syncthreads();                                 // every wave is done reading the ring: it becomes the staging area
  igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * (WM * WN * 4), lnA, lnC, zeros);
}

// ---------------------------------------------------------------------------------------------------------
// 8-wave software-pipelined kernel (one 512-thread workgroup per CU, two waves per SIMD).
//
// Why a second structure: in the 4-wave kernel above every kk-step is {4 ds_read_b128 -> lgkmcnt(0) -> 4 MFMA} on ONE
// fragment register set and every k-tile starts with vmcnt(0), so LDS latency and DMA latency are both exposed and the
// matrix pipe idles ~2/3 of the time (measured 25..35 % MFMA utilisation).  Here the k-loop is hand ordered:
//   * fragments are double buffered in registers: the ds_reads of step kk+1 are issued before the MFMAs of step kk and
//     waited for with a COUNTED lgkmcnt (hipcc only emits lgkmcnt(0) across the loop back edge, so the reads are inline
//     asm and every wait is followed by sched_barrier(0) so no MFMA is hoisted above it);
//   * the ring is NS >= 3 deep and the DMA wait is counted too: at the single barrier of k-tile kt (between its third and
//     fourth kk-step) a wave waits only for its own pieces of tile kt+1; the pieces of tile kt+NS-1 -- issued BETWEEN the
//     MFMAs of the first three steps, into the slot the previous barrier freed -- stay in flight across the raw s_barrier.
//     After the barrier the first fragments of tile kt+1 are prefetched under the fourth step;
//   * WGM = waves along M (4: 4x2 wave grid, 8: 8x1 -- the 256x160 tile, 3: the 6-wave 96x128 tile).  BN need not be a
//     multiple of 64: the weight tile's BN/8 eight-row pieces are dealt round-robin, waves below REM carry one more piece
//     and wait on their own count.
//   * S2 ("two k-tiles per rendezvous", rings of >= 4 slots): the vmcnt wait + s_barrier run in every SECOND k-tile only and cover the
//     next TWO tiles; the DMA lead is one tile shorter (tile kt + NS - 2 goes into the slot of tile kt - 2, which the last barrier
//     -- in tile kt - 1 or kt - 2 -- has freed).  Half the rendezvous of a launch for one tile less in flight.
// TSW ("transposed by operand swap", f16 256x128 only): workgroups whose whole tile lies in the transposed part of the output (the V^T
// columns of a fused QKV projection) run a copy of the k-loop with the MFMA operand roles swapped -- activations as the A operand -- so a
// lane's accumulators are 4 x 4 consecutive ROWS (keys) of one column: the transposed store is then the direct row-per-lane epilogue
// (permlane32 half swap, 16-byte stores along the key axis) instead of the LDS-staged transpose (12.6 k cycles per 64x64 wave tile, the
// longest epilogue of the launch: tools/timeline_probe.py).  Same products, same k order: bit-identical values.
// XH (with XA, round 6): the fused cross-attention at split precision (xattn_inplace_hl: hi / lo context images, three MFMAs per product)
template <int BM, int BN, int NS, int WGM = 4, int NW = 8, typename T = half_t, bool XA = false, bool S2 = false, bool TSW = false, bool XH = false>
__global__ __launch_bounds__(64 * NW) void igemm_pipe_kernel(const IgemmParams p, const void* zeros) {
  kernarg_prefetch<(int)sizeof(IgemmParams) + 8>();   // every argument line in flight at once (one wait instead of five)
  typedef typename PipeElem<T>::frag frag_t;
  constexpr int CE = 16 / (int)sizeof(T);     // elements per 16-byte chunk: 8 (f16) or 4 (f32, strict mode)
  constexpr int WGN = NW / WGM;               // NW waves per workgroup (8, or 4 with twice the wave tile)
  constexpr int WM = BM / WGM, WN = BN / WGN; // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NF = TM + TN;                 // ds_read_b128 per kk-step
  constexpr int BPC = BN / 8;                 // 8-row pieces of the weight tile
  constexpr int AJ = BM / (8 * NW), BJ = (BPC + NW - 1) / NW;   // DMA pieces per wave per k-tile (8 rows each, NW waves)
  constexpr int REM = BPC % NW;               // waves >= REM (when REM != 0) have no last weight piece
  constexpr int PER = AJ + BJ;
  static_assert(BM % (8 * NW) == 0 && WM % 32 == 0 && WN % 32 == 0 && BN % 8 == 0, "bad tile");
  constexpr int KT = 8 * CE;                  // elements per k-tile = one 128-byte row (64 f16 / 32 f32)
  constexpr int STAGE = (BM + BN) * 128;
  constexpr bool LIN = XA;                    // linear-only instantiation: scalar-base DMA addressing, no tap walk
  static_assert(NS >= 3, "counted-wait pipeline needs a ring of at least 3 slots");
  static_assert(BM * 128 + (TN - 1) * 4096 < 65536, "fragment offsets must fit the ds_read immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const bool lastb = REM == 0 || wave < REM;  // this wave carries weight piece BJ-1

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (grid = tiles x SK): consecutive remapped ids = the SK k-slices of one tile, so a tile's slices share an XCD
  // (its L2 then serves the partial slabs to the reducing workgroup at the same-XCD rate; placement is speed only)
  const int SK = p.splitk > 1 ? p.splitk : 1;
  const int slice = SK > 1 ? bid % SK : 0;
  if (SK > 1) bid /= SK;
  const int tile_id = bid;
  int tm, tn;
  if ((size_t)p.N * p.K > (size_t)p.M * p.Cin) { tn = bid / tilesM; tm = bid - tn * tilesM; }
  else { tm = bid / tilesN; tn = bid - tm * tilesN; }
  const int m0 = tm * BM, n0 = tn * BN;
  // TSW: the whole tile in the transposed part, whole 8-key pieces inside one batch entry, f16 out -> operand-swapped k-loop + direct transposed store
  bool tsw_swapped = false;
  if constexpr (TSW) {
    const int ok = n0 >= p.n_split && n0 + BN <= p.N && p.c_dt == DT_F16 && p.act == 0 && !p.epi_staged && (p.rpb % BM) == 0 && m0 + BM <= p.M &&
                   (p.ct_ld & 7) == 0 && (reinterpret_cast<uintptr_t>(p.Ct) & 15) == 0 && (!p.ln_stat || p.ln_slots <= 24);
    tsw_swapped = __builtin_amdgcn_readfirstlane(ok) != 0;
  }
  // counted DMA wait: K tiles of this wave's pieces may stay in flight
  auto wait_tiles = [&](auto KK) {
    constexpr int k = decltype(KK)::value;
    if constexpr (REM == 0) wait_vmcnt<PER * k>();
    else { if (lastb) wait_vmcnt<PER * k>(); else wait_vmcnt<(PER - 1) * k>(); }
  };

  // ---- DMA geometry: piece j of this wave covers tile rows (j*8 + wave)*8 .. +7; lane -> (row, slot)
  const int lrow = lane >> 3, slot = lane & 7;
  // weight rows first: they need no pixel arithmetic, and inside a UNet step the weights are the operand that arrives cold (the
  // activations were written by the previous launch) -- tile 0's weight pieces go out BEFORE the activation geometry below is
  // worked out (~1.5 k cycles earlier).  The in-order vmcnt accounting is unchanged: tile 0's pieces are still this wave's oldest.
  // Measured on the bench line, two libraries on one box: 23.30 / 23.27 -> 23.23 / 23.24 ms per step (profiles/r03_weights_first_ab.txt).
  const T* wptr[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (j * NW + wave) * 8 + lrow;
    wptr[j] = reinterpret_cast<const T*>(p.W) + (size_t)(n0 + row) * p.Kpad + (slot ^ ((row >> 1) & 7)) * CE;
  }
  // this workgroup's k-tiles [kbeg, kbeg + nk) of the Kpad / KT of the contraction (split-K: slice `slice` of SK)
  const int nk_all = p.Kpad / KT;
  const int kbeg = SK > 1 ? (int)((long)slice * nk_all / SK) : 0;
  const int nk = SK > 1 ? (int)((long)(slice + 1) * nk_all / SK) - kbeg : nk_all;
  if (kbeg > 0) {
#pragma unroll
    for (int j = 0; j < BJ; ++j) wptr[j] += kbeg * KT;
  }
  constexpr bool B0_EARLY = !LIN;      // (the scalar-base form of the fused cross-attention projections keeps the plain order)
  if constexpr (B0_EARLY) {
    if (nk > 0) {
      char* lb0 = smem + BM * 128 + wave * 1024;       // ring slot 0
      static_for<BJ>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if (j < BJ - 1 || lastb) {
          __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(lb0 + j * (NW * 1024)), 16, 0, 0);
          wptr[j] += KT;
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int HWo = p.Hout * p.Wout;
  const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
  // (prologue cost, tools/timeline_probe.py: the two integer divisions per tile row below took 1.5 - 2.7 k cycles of every launch.
  // Linear layers / 1x1 convs -- output row m IS source row m -- skip the pixel decomposition altogether; power-of-two image
  // sizes (every SDXL level) use shifts; anything else keeps the divisions.)
  const bool lin_rows = p.ksize == 1 && p.stride == 1 && p.up == 0 && p.pad == 0;
  const bool pow2 = (HWo & (HWo - 1)) == 0 && (p.Wout & (p.Wout - 1)) == 0;
  const int sh_hw = __builtin_ctz((unsigned)HWo | 0x40000000u), sh_w = __builtin_ctz((unsigned)p.Wout | 0x40000000u);
  int rb[AJ], ry[AJ], rx[AJ], rsw[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (j * NW + wave) * 8 + lrow;
    const int m = m0 + row;
    rsw[j] = (slot ^ ((row >> 1) & 7)) * CE;
    rb[j] = m < p.M ? m : -1; ry[j] = 0; rx[j] = 0;          // lin_rows: rb carries the row itself
    if (!lin_rows) {
      if (m < p.M) {
        int b, rem, oy;
        if (pow2) { b = m >> sh_hw; rem = m & (HWo - 1); oy = rem >> sh_w; }
        else { b = m / HWo; rem = m - b * HWo; oy = rem / p.Wout; }
        rb[j] = b; ry[j] = oy * p.stride - p.pad; rx[j] = (rem - oy * p.Wout) * p.stride - p.pad;
      } else { rb[j] = -1; ry[j] = -(1 << 28); rx[j] = 0; }
    }
  }
  const T* Ag = reinterpret_cast<const T*>(p.A);
  const T* aptr[AJ];
  int aadv[AJ];
  int s_c0 = 0, s_dy = 0, s_dx = 0;
  if (kbeg > 0) {   // start the tap walk inside the contraction
    const int e0 = kbeg * KT, tap = e0 / p.Cin;
    s_c0 = e0 - tap * p.Cin; s_dy = tap / p.ksize; s_dx = tap - s_dy * p.ksize;
  }
  auto retap = [&]() {
    if (lin_rows) {       // one tap, rows are contiguous K-runs: no bounds / pixel arithmetic
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const bool ok = rb[j] >= 0 && s_dy == 0;
        aptr[j] = ok ? Ag + (size_t)rb[j] * p.lda + rsw[j] + s_c0 : reinterpret_cast<const T*>(zeros);
        aadv[j] = ok ? KT : 0;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int iy = ry[j] + s_dy, ix = rx[j] + s_dx;
      const bool ok = (unsigned)iy < (unsigned)Hup && (unsigned)ix < (unsigned)Wup;
      const size_t off = (((size_t)(rb[j] < 0 ? 0 : rb[j]) * p.Hin + ((ok ? iy : 0) >> p.up)) * p.Win + ((ok ? ix : 0) >> p.up)) * p.lda + rsw[j] + s_c0;
      aptr[j] = ok ? Ag + off : reinterpret_cast<const T*>(zeros);
      aadv[j] = ok ? KT : 0;
    }
  };
  retap();
  // Linear layers (LIN; today = the fused cross-attention projections): scalar-base DMA addressing.  A row of the tile is a
  // contiguous K-run, so piece q reads {wave-uniform 64-bit base in SGPRs} + {loop-invariant 32-bit lane offset}: the k-loop
  // advances TWO scalar bases per k-tile (s_add_u32 / s_addc_u32) instead of PER 64-bit VGPR pointers (2 VALU each) and drops
  // the tap walk.  Rows past M read row M - 1 (never stored).
  unsigned long long abase = 0, wbase = 0;
  unsigned aoff[AJ], woff[BJ];
  if constexpr (LIN) {
    abase = (unsigned long long)(uintptr_t)(Ag + (size_t)m0 * p.lda + (size_t)kbeg * KT);
    wbase = (unsigned long long)(uintptr_t)(reinterpret_cast<const T*>(p.W) + (size_t)n0 * p.Kpad + (size_t)kbeg * KT);
    abase = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(abase >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)abase);
    wbase = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(wbase >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)wbase);
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      int row = (j * NW + wave) * 8 + lrow;
      const int sw = (slot ^ ((row >> 1) & 7)) * 16;
      if (m0 + row >= p.M) row = p.M - 1 - m0;
      aoff[j] = (unsigned)row * (unsigned)(p.lda * (int)sizeof(T)) + sw;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int row = (j * NW + wave) * 8 + lrow;
      woff[j] = (unsigned)row * (unsigned)(p.Kpad * (int)sizeof(T)) + (slot ^ ((row >> 1) & 7)) * 16;
    }
  }
  // pieces q of one k-tile: q < AJ -> activation piece q, else weight piece q - AJ.  PH selects the pieces with q % 3 == PH
  // (PH < 0: all of them); the tap walk advances once per k-tile, after the last piece (tile_done).
  auto issue = [&](int buf, auto PH) {
    constexpr int ph = decltype(PH)::value;
    char* la = smem + buf * STAGE + wave * 1024;
    char* lb = la + BM * 128;
    static_for<PER>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      if constexpr (ph == -1 || (ph >= 0 && q % 3 == ph) || (ph == -3 && q < AJ)) {      // -1: all pieces; -3: the activation pieces only
        if constexpr (LIN) {
          // saddr form: global_load_lds_dwordx4 voffset, sbase -- M0 = LDS byte address of this wave's 1-KiB piece
          // (hand-written, so the hazards are ours to keep: an LDS-DMA instruction must not issue in the wait state right behind
          // the SALU write of M0 -- it would take the PREVIOUS piece's LDS address -- and a VMEM instruction needs 5 wait
          // states behind a VALU write (v_readlane / v_readfirstlane restore) of its scalar base; s_mov + s_nop 3 covers both.
          // The compiler's hazard recogniser does this for the builtin form and does not look inside inline asm.)
          if constexpr (q < AJ) {
            const unsigned m = lds0 + buf * STAGE + wave * 1024 + q * (NW * 1024), vo = aoff[q];
            const unsigned long long sb = abase;      // (asm operands do not capture into the generic lambda by themselves)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(vo), "s"(sb) : "memory");
          } else if (q - AJ < BJ - 1 || lastb) {
            const unsigned m = lds0 + buf * STAGE + BM * 128 + wave * 1024 + (q - AJ) * (NW * 1024), vo = woff[q - AJ];
            const unsigned long long sb = wbase;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(vo), "s"(sb) : "memory");
          }
        } else if constexpr (q < AJ) {
          __builtin_amdgcn_global_load_lds((gptr_t)aptr[q], (lptr_t)(la + q * (NW * 1024)), 16, 0, 0);
          aptr[q] += aadv[q];
        } else if (q - AJ < BJ - 1 || lastb) {     // ragged weight tile: wave-uniform predicate on the last piece
          __builtin_amdgcn_global_load_lds((gptr_t)wptr[q - AJ], (lptr_t)(lb + (q - AJ) * (NW * 1024)), 16, 0, 0);
          wptr[q - AJ] += KT;
        }
      }
    });
  };
  auto tile_done = [&]() {
    if constexpr (LIN) { abase += KT * sizeof(T); wbase += KT * sizeof(T); return; }
    s_c0 += KT;
    if (s_c0 == p.Cin) {
      s_c0 = 0;
      if (++s_dx == p.ksize) { s_dx = 0; ++s_dy; }
      retap();
    }
  };

  f32x16 acc[TM][TN];   // (zeroed behind the prologue's DMA issue: the writes ride under the ring fill)

  const int fr = lane & 31, fh = lane >> 5;
  // per-lane fragment address inside a stage: A rows wm*WM + i*32 + fr (i -> +4096 B immediate), B rows likewise behind
  // the A tile.  sw(row) = (row>>1)&7 is the same for rows 32 apart, so one base per operand; step kk flips chunk bits
  // 1..2:  chunk(kk) = (kk*2 + fh) ^ sw = (fh ^ sw) ^ (kk << 1)  ->  byte offset ^ (kk << 5)
  unsigned basea, baseb;
  {
    const int ra = wm * WM + fr, rbw = wn * WN + fr;
    basea = lds0 + ra * 128 + ((fh ^ ((ra >> 1) & 7)) << 4);
    baseb = lds0 + BM * 128 + rbw * 128 + ((fh ^ ((rbw >> 1) & 7)) << 4);
  }
  constexpr bool HL = is_hl<T>::value;        // split-operand element: four fragment sets (hi0, lo0, hi1, lo1), three MFMAs per pair
  frag_t fA[HL ? 4 : 2][TM], fB[HL ? 4 : 2][TN];
  auto ldfrag = [&](unsigned so, int kk, auto SET) {
    constexpr int set = decltype(SET)::value;
    const unsigned aa = (basea ^ (kk << 5)) + so, ab = (baseb ^ (kk << 5)) + so;
    static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<decltype(I)::value * 4096, frag_t>(aa); });
    static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<decltype(J)::value * 4096, frag_t>(ab); });
  };
  // MFMAs of one kk-step from fragment set SET; DMA pieces PH (or none, PH = 3) are issued between them
  auto mma = [&](auto SET, int buf, auto PH, bool more, auto SWT) {
    constexpr int set = decltype(SET)::value;
    constexpr int ph = decltype(PH)::value;
    constexpr bool sw = decltype(SWT)::value;      // operand roles swapped (TSW): D^T tiles
    if constexpr (!HL) {
    if constexpr (sw) acc[0][0] = PipeElem<T>::mma(fA[set][0], fB[set][0], acc[0][0]);
    else acc[0][0] = PipeElem<T>::mma(fB[set][0], fA[set][0], acc[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ph < 3) {
      if (more) issue(buf, PH);            // wave-uniform branch around the DMA pieces only, never around MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<TM * TN - 1>([&](auto X) {
      constexpr int x = decltype(X)::value + 1, i = x / TN, j = x % TN;
      if constexpr (sw) acc[i][j] = PipeElem<T>::mma(fA[set][i], fB[set][j], acc[i][j]);
      else acc[i][j] = PipeElem<T>::mma(fB[set][j], fA[set][i], acc[i][j]);
    });
    __builtin_amdgcn_sched_barrier(0);
    }
  };
  // split-operand MFMAs.  CROSS = false: hi x hi of fragment set SH (TM*TN MFMAs); CROSS = true: the two cross terms
  // w_hi x a_lo and w_lo x a_hi of the pair (SH, SL) (2*TM*TN MFMAs).  DMA pieces PH go behind the first MFMA, as above.
  // WX = every packed weight of this layer is ONE f16 (its lo half is zero: what a real SDXL record holds): the w_lo x a_hi
  // MFMAs would add exact zeros and are left out, as are the ds_reads of the weights' lo fragments -- bit-identical results.
  auto mma_hl = [&](auto SHI, auto SLO, auto CROSS, auto WXT, int buf, auto PH, bool more) {
    if constexpr (HL) {
      constexpr int sh = decltype(SHI)::value, sl = decltype(SLO)::value, ph = decltype(PH)::value;
      constexpr bool cross = decltype(CROSS)::value, wx = decltype(WXT)::value;
      auto one = [&](auto X) {
        constexpr int x = decltype(X)::value, i = x / TN, j = x % TN;
        if constexpr (!cross) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[sh][j], fA[sh][i], acc[i][j], 0, 0, 0);
        else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[sh][j], fA[sl][i], acc[i][j], 0, 0, 0);
          if constexpr (!wx) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[sl][j], fA[sh][i], acc[i][j], 0, 0, 0);
        }
      };
      one(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ph < 3) {
        if (more) issue(buf, PH);
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<TM * TN - 1>([&](auto X) { one(std::integral_constant<int, decltype(X)::value + 1>{}); });
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using IALL = std::integral_constant<int, -1>;
  using IAONLY = std::integral_constant<int, -3>;
  constexpr int LEAD = S2 ? NS - 2 : NS - 1;   // tile kt + LEAD is issued during tile kt
  constexpr int NPRO = LEAD;                   // tiles staged by the prologue
  static_assert(!S2 || (NS >= 4 && !HL), "two tiles per rendezvous: ring of >= 4 slots, f16 / f32 elements");

  // fused cross-attention, one-MFMA-row wave tiles: the 24 context fragments (96 VGPRs -- these kernels have the room) are
  // requested BEFORE the first DMA piece, so they are the oldest entries of the in-order vmcnt queue and ride under the
  // prologue's wait for tile 0 instead of adding a memory round trip to the epilogue
  constexpr bool XA_EARLY = XA && TM == 1 && !XH;      // (the split-precision form needs 48 fragments per head: loaded inside the epilogue)
  half8 xkf[XA ? 3 : 1][4], xvf[XA ? 2 : 1][6];
  if constexpr (XA_EARLY) xattn_load_frags(p, m0 + wm * WM, n0 + wn * WN, lane, xkf, xvf);
  // folded LayerNorm: the tile's row coefficients, evaluated once per workgroup (LnCoop) where 2 KiB of LDS are left behind the ring
  typedef LnCoop<BM, 64 * NW> LnC;
  constexpr bool LN_COOP = LnC::OK && pipe_lds_total(NS * STAGE, 0) > NS * STAGE;
  float* ln_coef = reinterpret_cast<float*>(smem + NS * STAGE);
  LnC lnc;
  if constexpr (LN_COOP) { lnc.load(p, m0, tid); __builtin_amdgcn_sched_barrier(0); }
  // ---- prologue: tiles 0 .. NPRO-1 in flight, wait for tile 0 only
#pragma unroll
  for (int s = 0; s < NPRO; ++s)
    if (s < nk) {
      if (B0_EARLY && s == 0) issue(0, IAONLY{}); else issue(s, IALL{});      // (tile 0's weight pieces are already on their way)
      tile_done();
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float lnA[TM], lnC[TM];
  bool hl_wexact = false;        // split-operand launches: every packed weight is one f16 (its lo half is zero)
  const bool ln_coop = LN_COOP && p.ln_slots <= 24;
  if constexpr (LN_COOP) lnc.finish(p, m0, ln_coef);
  if (!ln_coop) ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  // (S2: the first rendezvous covers tiles 0 AND 1)
  if (NPRO <= nk) wait_tiles(std::integral_constant<int, NPRO - (S2 ? 2 : 1)>{}); else wait_vmcnt<0>();
  // finish()'s ds_write of the row coefficients must have LANDED before the barrier releases their readers: a raw s_barrier
  // carries no wait of its own, and a wave whose tile 0 is already there reaches it a few cycles behind the write
  if constexpr (LN_COOP) wait_lgkmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (ln_coop) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      lnA[i] = p.ln_stat ? ln_coef[(wm * WM + i * 32 + fr) * 2] : 1.f;
      lnC[i] = p.ln_stat ? ln_coef[(wm * WM + i * 32 + fr) * 2 + 1] : 0.f;
    }
  }
  if constexpr (HL) {   // split-operand weights are packed times a power of two (exact): the row coefficient of the epilogue undoes it
    // (the A operand's factor is per batch entry of the row: IgemmParams::a_scale_rpb)
    const float wsc = p.acc_scale ? *p.acc_scale : 1.f;                                           // powers of two: the products are exact
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WM + i * 32 + fr;
      const int be = p.a_scale_rpb > 0 ? (m < p.M ? m : p.M - 1) / p.a_scale_rpb : 0;
      lnA[i] = wsc * (p.a_scale ? p.a_scale[be] : 1.f); lnC[i] = 0.f;
    }
    // (readfirstlane: the compiler must SEE that the choice is wave-uniform -- as a vector condition it runs both k-loops one after
    // the other under complementary EXEC masks, and MFMAs, barriers and the M0-addressed DMA do not honour EXEC)
    const int wx = (p.acc_scale && p.hl_wexact_ok && p.acc_scale[1] != 0.f) ? 1 : 0;
    hl_wexact = __builtin_amdgcn_readfirstlane(wx) != 0;
  }
  ldfrag(0, 0, I0{});
  // The k-loop is unrolled by the ring depth: ring slots become compile-time constants, so every
  // fragment read is {one of 16 loop-invariant lane addresses} + immediate and the DMA destinations fold into M0
  // constants -- the rolled loop re-derives them with ~25 VALU / SALU instructions per k-tile, and instruction issue (not
  // LDS or DMA bandwidth) is what fills this kernel's SIMDs.  ds_read immediates are 16 bit: slots
  // beyond 64 KiB go through a second address set (+ 65536).
  // wave tiles of up to 4 MFMA tiles per operand: two address sets (+0, +64 KiB); wider ones (256x160: 5): one set per slot
  constexpr bool PERSLOT = TM > 4 || TN > 4 || NS * STAGE > 131072;
  constexpr int NSET = PERSLOT ? NS : 2;
  unsigned fa[NSET][4], fb[NSET][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
      fa[q][kk] = (basea ^ (kk << 5)) + (PERSLOT ? q * STAGE : q * 65536u);
      fb[q][kk] = (baseb ^ (kk << 5)) + (PERSLOT ? q * STAGE : q * 65536u);
    }
  auto ldf = [&](auto SO, auto KK, auto SET, auto NOB) {          // NOB: leave out the weight fragments (split-operand lo sets of exact-f16 weights)
    constexpr unsigned so = decltype(SO)::value;
    constexpr int kk = decltype(KK)::value, set = decltype(SET)::value;
    constexpr bool nob = decltype(NOB)::value;
    constexpr int hi = PERSLOT ? (int)(so / STAGE) : (so >= 65536u ? 1 : 0);
    constexpr unsigned lo = PERSLOT ? 0u : so - hi * 65536u;
    static_assert(lo + (TM - 1) * 4096 < 65536u && lo + (TN - 1) * 4096 < 65536u, "fragment immediate out of range");
    static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<lo + decltype(I)::value * 4096, frag_t>(fa[hi][kk]); });
    if constexpr (!nob) static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<lo + decltype(J)::value * 4096, frag_t>(fb[hi][kk]); });
  };
  using NB0 = std::false_type;
  auto ktile = [&](int kt, auto CUR, auto WXT) {
    constexpr bool wx = decltype(WXT)::value;
    using WX = std::integral_constant<bool, wx>;
    constexpr int NFL = wx ? TM : NF;          // ds_reads of a lo fragment set
    constexpr int c = decltype(CUR)::value;
    constexpr int nslot = (c + 1) % NS, fl = (c + NS - (S2 ? 2 : 1)) % NS;      // fl: the slot tile kt + LEAD goes into
    using SO = std::integral_constant<unsigned, (unsigned)c * STAGE>;
    using SN = std::integral_constant<unsigned, (unsigned)nslot * STAGE>;
    const bool more = kt + LEAD < nk;
    if constexpr (HL) {
      // one 32-deep k-tile = the pairs (hi0, lo0) and (hi1, lo1) in sets 0..3; set 0 was requested behind the previous barrier
      using F = std::false_type; using Tt = std::true_type;
      ldf(SO{}, I1{}, I1{}, WX{});
      wait_lgkmcnt<NFL>();
      mma_hl(I0{}, I0{}, F{}, WX{}, fl, I0{}, more);      // w_hi0 x a_hi0
      ldf(SO{}, I2{}, I2{}, NB0{});
      wait_lgkmcnt<NF>();
      mma_hl(I0{}, I1{}, Tt{}, WX{}, fl, I1{}, more);     // w_hi0 x a_lo0 (+ w_lo0 x a_hi0)
      ldf(SO{}, I3{}, I3{}, WX{});
      wait_lgkmcnt<NFL>();
      mma_hl(I2{}, I2{}, F{}, WX{}, fl, I2{}, more);      // w_hi1 x a_hi1
      if (more) tile_done();
      if (kt + 1 < nk) {
        if (more) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ldf(SN{}, I0{}, I0{}, NB0{});
      } else {
        wait_lgkmcnt<0>();
      }
      mma_hl(I2{}, I3{}, Tt{}, WX{}, fl, I3{}, false);    // w_hi1 x a_lo1 (+ w_lo1 x a_hi1)
      return;
    }
    using SWT = std::integral_constant<bool, !HL && wx>;     // (non-split kernels: the flag selects the operand-swapped copy, TSW)
    ldf(SO{}, I1{}, I1{}, NB0{});
    wait_lgkmcnt<NF>();
    mma(I0{}, fl, I0{}, more, SWT{});
    ldf(SO{}, I2{}, I0{}, NB0{});
    wait_lgkmcnt<NF>();
    mma(I1{}, fl, I1{}, more, SWT{});
    ldf(SO{}, I3{}, I1{}, NB0{});
    wait_lgkmcnt<NF>();
    mma(I0{}, fl, I2{}, more, SWT{});
    if (more) tile_done();
    if (kt + 1 < nk) {
      if (!S2 || (kt & 1)) {
        // own pieces of the next tile (S2: the next two) landed, then the rendezvous; tiles beyond may stay in flight
        if (more) wait_tiles(std::integral_constant<int, LEAD - (S2 ? 2 : 1)>{}); else wait_vmcnt<0>();
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ldf(SN{}, I0{}, I0{}, NB0{});
      } else {
        // S2, even tile: tile kt + 1 was covered by the rendezvous of tile kt - 1 (or the prologue's) -- no wait, no barrier
        ldf(SN{}, I0{}, I0{}, NB0{});
        wait_lgkmcnt<NF>();
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      wait_lgkmcnt<0>();
    }
    mma(I1{}, fl, I3{}, false, SWT{});
  };
  int kt = 0;
  auto kloop = [&](auto WXT) {
    for (; kt + NS <= nk; kt += NS) static_for<NS>([&](auto S) { ktile(kt + decltype(S)::value, S, WXT); });
    static_for<NS - 1>([&](auto S) { if (kt + decltype(S)::value < nk) ktile(kt + decltype(S)::value, S, WXT); });
  };
  if constexpr (HL) {
    // two copies of the k-loop: the compiler moves the first fragments (asm ds_reads it believes complete) into each copy's
    // registers at the loop entry -- they must have LANDED before that (one LDS latency per launch)
    wait_lgkmcnt<0>();
    if (hl_wexact) kloop(std::true_type{}); else kloop(std::false_type{});       // wave-uniform: a device scalar next to the weight scale
  } else if constexpr (TSW) {
    // (two copies of the k-loop, as the split-operand kernel's: the choice is made wave-uniform for the compiler, and the first
    // fragments -- asm reads it believes complete -- must have landed before either copy takes them over)
    wait_lgkmcnt<0>();
    if (tsw_swapped) kloop(std::true_type{}); else kloop(std::false_type{});
  } else kloop(std::false_type{});
  __builtin_amdgcn_s_barrier();                    // every wave is done reading the ring: it becomes the staging area
  asm volatile("" ::: "memory");
  if constexpr (BN == 128) {
    if (SK > 1) {
      // ---- split-K combine inside the launch.  Every slice parks its fp32 accumulators in its slab (register order: 16-byte
      // stores, lane-contiguous), then ONE agent-scope release + ticket; the workgroup that draws the last ticket acquires
      // once and sums the SK slabs in slice order 0..SK-1 -- its own included, so the result does not depend on which slice
      // arrived last (bit-reproducible) -- and runs the normal epilogue.  Correct for any placement of the slices
      // (cdna_hip_programming.md section 6 guideline 16: plain stores -> vmcnt(0) -> barrier -> lane-0 release -> asm vmcnt(0)
      // -> relaxed agent ticket; consumer: acquire once -> barrier -> plain loads).  The last arriver re-arms the counter.
      constexpr int NV = TM * TN * 4;                                    // f32x4 vectors per lane
      f32x4* slab = reinterpret_cast<f32x4*>(p.splitk_ws) + ((size_t)tile_id * SK + slice) * (size_t)(NW * NV * 64);
      static_for<TM * TN>([&](auto X) {
        constexpr int x = decltype(X)::value, i = x / TN, j = x % TN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (round 4: write-through stores instead of plain stores + an agent-scope release -- the release writes the XCD's L2 back once per
          //  workgroup, 240 times per launch; the write-through pieces leave as they are issued and vmcnt(0) covers them: g_splitk_wt = 0 restores the fence)
          const f32x4 v = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          f32x4* dst = slab + (size_t)((wave * NV + x * 4 + q) * 64 + lane);
          if (p.splitk_wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
          else *dst = v;
        }
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      volatile int* flag = reinterpret_cast<volatile int*>(smem + NS * STAGE - 16);   // inside the one LDS array, beyond the staging regions
      if (tid == 0) {
        if (!p.splitk_wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(p.splitk_cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)(SK - 1);
        if (last) {
          __hip_atomic_store(p.splitk_cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
      }
      __syncthreads();
      if (!*flag) return;
      const f32x4* s0 = reinterpret_cast<const f32x4*>(p.splitk_ws) + (size_t)tile_id * SK * (size_t)(NW * NV * 64);
      static_for<TM * TN>([&](auto X) {
        constexpr int x = decltype(X)::value, i = x / TN, j = x % TN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 sum = s0[(size_t)((wave * NV + x * 4 + q) * 64 + lane)];
          for (int sl = 1; sl < SK; ++sl) sum += s0[(size_t)sl * (NW * NV * 64) + (size_t)((wave * NV + x * 4 + q) * 64 + lane)];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] = sum[r];
        }
      });
      __syncthreads();          // every wave has read the flag before the staging regions are written
    }
  }
  if constexpr (XA) {
    static_assert(TN == 2 && sizeof(T) == 2, "fused cross-attention: wave tile = one 64-wide head, f16");
    static_assert(NW * WM * WN * 4 <= NS * STAGE, "staging regions must fit the dead ring");
    if constexpr (XH) {
      static_assert(!XH || (XA && TM == 1), "split-precision fused cross-attention: XA kernels with one-MFMA-row wave tiles");
      xattn_inplace_hl<TM>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros);
    } else {
    if constexpr (!XA_EARLY) xattn_load_frags(p, m0 + wm * WM, n0 + wn * WN, lane, xkf, xvf);
    xattn_inplace<TM>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros, xkf, xvf);
    }
    IgemmParams pe = p;                       // bias and the LayerNorm affine went into q: the store adds nothing
    pe.bias = nullptr; pe.ln_stat = nullptr;
    if (!p.epi_staged && igemm_rows_ok<TM, TN, false>(pe, n0 + wn * WN))
      igemm_epilogue_rows<TM, TN, false>(pe, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros);
    else
      igemm_epilogue_staged<TM, TN>(pe, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * (WM * WN * 4), lnA, lnC, zeros);
    return;
  }
  // whole wave tiles take the direct row-per-lane epilogue (registers -> permlane32 half swap -> 16-byte stores); tiles cut by N
  // or n_split, the transposed V^T part, GroupNorm-statistics producers and unaligned outputs keep the LDS-staged one.  The
  // choice depends on N / alignment only (never on the batch), and every wave makes it for itself (staging regions are private).
  // (GEGLU projections keep the staged epilogue: measured per shape in the step, tools/launch_ab.py, the direct form is 1 - 5 us
  // faster on every plain shape and 23 - 50 us SLOWER on the 256x320 GEGLU kernel, whose 160 accumulator registers leave no
  // room for the per-column vectors -- profiles/r03_epilogue_ab.txt)
  if constexpr (TSW) {
    if (tsw_swapped) {
      igemm_epilogue_swapped<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, ln_coop ? ln_coef + (wm * WM) * 2 : nullptr, zeros);
      return;
    }
  }
  if (!p.epi_staged && p.act != 1 && igemm_rows_ok<TM, TN, false>(p, n0 + wn * WN)) {
    igemm_epilogue_rows<TM, TN, false>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros);
    return;
  }
  constexpr bool FITS = NW * WM * WN * 4 <= NS * STAGE;        // full-width staging regions fit the dead ring
  if (FITS || p.act == 1) {
    const int region = p.act == 1 ? WM * (WN / 2) * 4 : WM * WN * 4;   // GEGLU halves the staged width
    if constexpr (BM == 256 && BN == 128 && NW == 8 && WGM == 4 && (sizeof(T) == 2 || HL)) {      // (f16, and -- round 6 -- the split-operand kernel: fp32 rows, same epilogue)
      static_assert(NW * WM * WN * 4 + 4096 <= NS * STAGE, "GroupNorm-statistics scratch must fit behind the staging regions");
      const GnCtx gc{smem + NW * WM * WN * 4, wave, wm, wn, m0, n0};
      igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * region, lnA, lnC, zeros, &gc);
    } else
    igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * region, lnA, lnC, zeros);
  } else {
    igemm_epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, fr, fh, lnA, lnC);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Wide-tile variant for the GEGLU projections: block tile 256 x 320, k-tile 32.
//
// The global->LDS path sustains ~22 B/clk/CU whatever issues it, so the reachable MFMA rate of a tile is
// BM*BN/(BM+BN) flop per DMA byte: 85 for 256x128, 98 for 256x160, 142 for 256x320.  A 64-deep k-tile of that tile
// would not fit a 3-slot ring (72 KiB per slot); with KT = 32 a slot is 36 KiB and FOUR slots fit (144 KiB).  N = 10240 at
// M = 2048 is then 8 x 32 = 256 tiles: one workgroup per CU, ONE round (256x160 needs two, 256x128 three), so the per-tile
// fixed cost is paid once.  8 waves as 4 (M) x 2 (N), wave tile 64 x 160 = 2 x 5 MFMA tiles: 7 ds_read_b128 per 10 MFMA.
// LDS rows are 64 bytes (32 halfs): a 1-KiB DMA piece covers 16 rows, source chunk = slot ^ ((row>>2)&3), which makes the
// 16-lane ds_read_b128 service groups hit 16 distinct 16-byte slots of the 256-byte bank row.
// Per k-tile: {frags kk=0 -> 10 MFMA} {frags kk=1 -> wait own pieces of tile kt+1 -> barrier -> 10 MFMA with the pieces of
// tile kt+NS (slot just freed) issued between them}.  GEGLU epilogue only (staged through LDS in two 32-row passes).
#ifdef SDXL_MEASURE
// coarse s_memtime stamps of the wide kernel (tools/wide_timeline.py): [workgroup][wave][16], words 0..7 = entry, ring fill issued, tile 0 landed
// (first barrier passed), k-loop done, ring dead (barrier), first / second epilogue pass issued, stores drained
__device__ unsigned* g_wide_tl = nullptr;
void igemm_set_wide_timeline(void* buf) {
  unsigned* b = reinterpret_cast<unsigned*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_tl), &b, sizeof(b)) != hipSuccess) throw std::runtime_error("igemm: cannot set the wide-kernel timeline buffer");
}
#define WIDE_STAMP(i) do { wtl[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WIDE_STAMP(i) do { } while (0)
#endif
// DB (round 5): fragments of the NEXT kk-step are requested before the MFMAs of the current one (two register sets, counted lgkmcnt, the pipe
// kernels' scheme): the rolled form above waits for its seven ds_read_b128 in front of every 10-MFMA burst with nothing of its own to issue
// meanwhile -- 1.85 - 1.97 k cycles per 32-deep k-tile for 1.28 k cycles of matrix-pipe work (profiles/r04_wide_geglu_timeline.txt).
template <int NS, bool DB = false>
__global__ __launch_bounds__(512) void igemm_wide_kernel(const IgemmParams p, const void* zeros) {
#ifdef SDXL_MEASURE
  unsigned wtl[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  wtl[8] = (unsigned)__builtin_amdgcn_s_memrealtime();     // 100 MHz constant clock: shader clock of this launch = d(memtime) / d(realtime)
#endif
  WIDE_STAMP(0);
  kernarg_prefetch<(int)sizeof(IgemmParams) + 8>();   // every argument line in flight at once (one wait instead of five)
  constexpr int BM = 256, BN = 320, KT = 32;
  constexpr int WM = 64, WN = 160, TM = 2, TN = 5, NF = TM + TN;
  constexpr int ROWB = KT * 2;                 // 64 bytes per tile row
  constexpr int STAGE = (BM + BN) * ROWB;      // 36864
  constexpr int AJ = BM / 16 / 8;              // 2 pieces of 16 rows per wave
  constexpr int BPC = BN / 16, BJ = (BPC + 7) / 8, REM = BPC % 8;   // 20 pieces: waves 0..3 carry 3, waves 4..7 carry 2
  constexpr int PER = AJ + BJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool lastb = wave < REM;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  if ((size_t)p.N * p.K > (size_t)p.M * p.Cin) { tn = bid / tilesM; tm = bid - tn * tilesM; }
  else { tm = bid / tilesN; tn = bid - tm * tilesN; }
  const int m0 = tm * BM, n0 = tn * BN;
  auto wait_tiles = [&](auto KK) {
    constexpr int k = decltype(KK)::value;
    if (lastb) wait_vmcnt<PER * k>(); else wait_vmcnt<(PER - 1) * k>();
  };

  // ---- DMA geometry (linear layers only: ksize 1, stride 1): piece j of this wave = tile rows (j*8 + wave)*16 .. +15
  const int lrow = lane >> 2, slot = lane & 3;
  const half_t* aptr[AJ];
  int aadv[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (j * 8 + wave) * 16 + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    aptr[j] = ok ? reinterpret_cast<const half_t*>(p.A) + (size_t)m * p.lda + (slot ^ ((row >> 2) & 3)) * 8
                 : reinterpret_cast<const half_t*>(zeros);
    aadv[j] = ok ? KT : 0;
  }
  const half_t* wptr[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int row = (j * 8 + wave) * 16 + lrow;
    wptr[j] = reinterpret_cast<const half_t*>(p.W) + (size_t)(n0 + row) * p.Kpad + (slot ^ ((row >> 2) & 3)) * 8;
  }
  auto issue = [&](int buf, auto Q) {            // piece Q of the next tile into ring slot buf
    constexpr int q = decltype(Q)::value;
    char* la = smem + buf * STAGE + wave * 1024;
    char* lb = la + BM * ROWB;
    if constexpr (q < AJ) {
      __builtin_amdgcn_global_load_lds((gptr_t)aptr[q], (lptr_t)(la + q * 8192), 16, 0, 0);
      aptr[q] += aadv[q];
    } else if (q - AJ < BJ - 1 || lastb) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[q - AJ], (lptr_t)(lb + (q - AJ) * 8192), 16, 0, 0);
      wptr[q - AJ] += KT;
    }
  };

  // the two 32-row halves of the wave tile live in SEPARATE accumulator arrays: the epilogue runs once per half, and handing
  // it `&acc[i]` of one [2][5] array made hipcc address the accumulators through scratch (the round-1 build of this kernel
  // lost 35 us in its epilogue to exactly that)
  f32x16 acc0[1][TN], acc1[1][TN];
  const int nk = p.Kpad / KT;
  const int fr = lane & 31, fh = lane >> 5;
  unsigned basea, baseb;
  {
    const int ra = wm * WM + fr, rbw = wn * WN + fr;
    basea = lds0 + ra * ROWB + ((fh ^ ((ra >> 2) & 3)) << 4);
    baseb = lds0 + BM * ROWB + rbw * ROWB + ((fh ^ ((rbw >> 2) & 3)) << 4);
  }
  constexpr int NSET = DB ? 2 : 1;
  half8 fA[NSET][TM], fB[NSET][TN];
  auto ldfrag = [&](auto SET, unsigned so, int kk) {       // chunk(kk) = (kk*2 + fh) ^ sw = (fh ^ sw) ^ (kk << 1) -> byte offset ^ (kk << 5)
    constexpr int set = decltype(SET)::value;
    const unsigned aa = (basea ^ (kk << 5)) + so, ab = (baseb ^ (kk << 5)) + so;
    static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<decltype(I)::value * 32 * ROWB>(aa); });
    static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<decltype(J)::value * 32 * ROWB>(ab); });
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, NSET - 1>;
  // ten MFMAs of one kk-step; with DMA: the PER pieces of the next tile go behind MFMAs 0, 2, 4, 6, 8
  auto mma = [&](auto SET, int buf, bool dma) {
    constexpr int set = decltype(SET)::value;
    __builtin_amdgcn_s_setprio(1);
    static_for<TM * TN>([&](auto X) {
      constexpr int x = decltype(X)::value, i = x / TN, j = x % TN;
      if constexpr (i == 0) acc0[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[set][j], fA[set][0], acc0[0][j], 0, 0, 0);
      else acc1[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[set][j], fA[set][1], acc1[0][j], 0, 0, 0);
      if constexpr ((x & 1) == 0 && x / 2 < PER) {
        __builtin_amdgcn_sched_barrier(0);
        if (dma) issue(buf, std::integral_constant<int, x / 2>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // folded LayerNorm: the tile's 256 row coefficients once per workgroup (LnCoop, 2 KiB of LDS behind the ring)
  typedef LnCoop<BM, 512> LnC;
  float* ln_coef = reinterpret_cast<float*>(smem + NS * STAGE);
  LnC lnc;
  lnc.load(p, m0, tid);
  __builtin_amdgcn_sched_barrier(0);
  // ---- prologue: NS - 1 tiles in flight, wait for tile 0 only.  The last ring slot is filled from inside k-tile 0 (its first kk-step
  // carries the pieces of tile NS - 1): the first MFMA does not wait behind the issue of a whole ring (every wave's VMEM issue queues
  // behind the CU's fetch rate: entry -> "ring fill issued" was 7.5 - 9 k cycles with three tiles, profiles/r04_wide_geglu_timeline.txt)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) static_for<PER>([&](auto Q) { issue(s, Q); });
  WIDE_STAMP(1);
  float lnA[TM], lnC[TM];
  const bool ln_coop = p.ln_slots <= 24;
  lnc.finish(p, m0, ln_coef);
  if (!ln_coop) ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  if (NS - 1 <= nk) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
  wait_lgkmcnt<0>();                             // the coefficients' ds_write has landed (raw s_barrier waits for nothing)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  WIDE_STAMP(2);
  if (ln_coop) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      lnA[i] = p.ln_stat ? ln_coef[(wm * WM + i * 32 + fr) * 2] : 1.f;
      lnC[i] = p.ln_stat ? ln_coef[(wm * WM + i * 32 + fr) * 2 + 1] : 0.f;
    }
  }
  // (zeroed only now: keeping 160 accumulator registers live across the statistics loads of ln_prologue spills)
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[0][j][r] = 0.f; acc1[0][j][r] = 0.f; }
  int cur = 0;
  if constexpr (DB) {
    // set 0 = kk-step 0, set 1 = kk-step 1 of the current tile.  Per tile: {request set 1 | wait set 0 | 10 MFMAs} {own pieces of tile kt + 1,
    // lgkmcnt(0) = every read of tile kt complete, barrier | request set 0 of tile kt + 1 | 10 MFMAs + the pieces of tile kt + NS}
    ldfrag(S0{}, 0, 0);
    wait_lgkmcnt<0>();       // (the compiler believes an asm read complete when the statement ends: nothing in flight across the loop entry)
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned so = cur * STAGE;
      const int nxt = cur + 1 == NS ? 0 : cur + 1;
      ldfrag(S1{}, so, 1);
      wait_lgkmcnt<NF>();                         // set 0 has landed (the seven reads just issued may still be in flight)
      if (kt == 0) mma(S0{}, NS - 1, NS - 1 < nk); else mma(S0{}, cur, false);
      const bool more = kt + NS < nk;
      if (kt + 1 < nk) {
        if (kt + NS - 1 < nk) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
        wait_lgkmcnt<0>();                        // set 1 has landed: all of this wave's reads of tile kt are complete
        __builtin_amdgcn_s_barrier();             // tile kt+1 visible; slot of tile kt free for tile kt+NS
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(S0{}, nxt * STAGE, 0);             // in flight under the MFMAs of kk-step 1
      } else {
        wait_lgkmcnt<0>();
      }
      mma(S1{}, cur, more);
      cur = nxt;
    }
    wait_lgkmcnt<0>();
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned so = cur * STAGE;
    ldfrag(S0{}, so, 0);
    wait_lgkmcnt<0>();
    if (kt == 0) mma(S0{}, NS - 1, NS - 1 < nk); else mma(S0{}, cur, false);      // (k-tile 0 completes the ring: slot NS - 1 has never been read)
    ldfrag(S0{}, so, 1);
    wait_lgkmcnt<0>();                          // own reads of tile kt complete
    const bool more = kt + NS < nk;
    if (kt + 1 < nk) {
      if (kt + NS - 1 < nk) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();             // tile kt+1 visible; slot of tile kt free for tile kt+NS
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(S0{}, cur, more);
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  WIDE_STAMP(3);
  __builtin_amdgcn_s_barrier();                  // ring dead -> staging area
  asm volatile("" ::: "memory");
  WIDE_STAMP(4);
  if (p.act == 1) {
    // two passes of 32 rows: a full 64 x 80 fp32 staging region per wave would not fit next to seven others
    char* region = smem + wave * (32 * (WN / 2) * 4);
    {
      const float la1[1] = {lnA[0]}, lc1[1] = {lnC[0]};
      igemm_epilogue_staged_impl<1, TN, true>(p, acc0, m0 + wm * WM, n0 + wn * WN, lane, region, false, la1, lc1, zeros);
    }
    WIDE_STAMP(5);
    {
      const float la1[1] = {lnA[1]}, lc1[1] = {lnC[1]};
      igemm_epilogue_staged_impl<1, TN, true>(p, acc1, m0 + wm * WM + 32, n0 + wn * WN, lane, region, false, la1, lc1, zeros);
    }
  }   // (the launcher only admits GEGLU projections)
#ifdef SDXL_MEASURE
  WIDE_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WIDE_STAMP(7);
  wtl[9] = (unsigned)__builtin_amdgcn_s_memrealtime();
  if (g_wide_tl && lane == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) g_wide_tl[((size_t)blockIdx.x * 8 + wave) * 16 + i] = wtl[i];
  }
#endif
}

// Per-DEVICE state: the zero page the DMA reads halo / tail rows from lives on the device that launches, and the
// dynamic-LDS attribute (up to 147 KiB) is set once per (kernel, device).  A second sdxl_ctx on another GPU of the same
// process gets its own.
constexpr int kMaxDev = 64;
static const void* g_zero_pages[kMaxDev] = {};
static int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) throw std::runtime_error("igemm: no current HIP device");
  return d;
}
const void* igemm_zero_page() { return g_zero_pages[current_device()]; }
int igemm_current_device() { return current_device(); }
static_assert(kMaxDev == kIgemmMaxDev, "per-device tables of the two translation units must agree");
#ifdef SDXL_MEASURE
bool launch_igemm_measure(const IgemmParams& p, int variant, hipStream_t s);   // igemm_measure.hip
#endif
void igemm_glds_init() {
  const int d = current_device();
  if (g_zero_pages[d]) return;
  void* z = nullptr;
  if (hipMalloc(&z, 4096) != hipSuccess || hipMemset(z, 0, 4096) != hipSuccess)
    throw std::runtime_error("igemm: cannot allocate the zero page");
  g_zero_pages[d] = z;
}
template <typename K> static void set_lds_attr(K kernel, size_t lds, bool (&done)[kMaxDev], int dev) {
  if (done[dev]) return;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    throw std::runtime_error("igemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  done[dev] = true;
}

template <int BM, int BN, int NS, int MINB = 2>
static void launch_glds(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  static bool attr_set[kMaxDev] = {};
  const int dev = current_device();
  set_lds_attr(&igemm_glds_kernel<BM, BN, NS, MINB>, lds, attr_set, dev);
  hipLaunchKernelGGL((igemm_glds_kernel<BM, BN, NS, MINB>), dim3(tilesM * tilesN), dim3(256), lds, s, p, g_zero_pages[dev]);
}

template <int BM, int BN, int NS, int WGM = 4, int NW = 8, typename T = half_t, bool XA = false, bool S2 = false, bool TSW = false, bool XH = false>
static void launch_pipe(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)pipe_lds_total(NS * (BM + BN) * 128, 0);   // ring + LayerNorm coefficients
  static bool attr_set[kMaxDev] = {};
  const int dev = current_device();
  set_lds_attr(&igemm_pipe_kernel<BM, BN, NS, WGM, NW, T, XA, S2, TSW, XH>, lds, attr_set, dev);
  const int sk = (BN == 128 && p.splitk > 1) ? p.splitk : 1;
  if (sk > 1 && (size_t)tilesM * tilesN * sk * BM * BN * 4 > p.splitk_ws_bytes) throw std::runtime_error("igemm: split-K workspace too small");
  IgemmParams q = p;
  q.splitk = sk;
  hipLaunchKernelGGL((igemm_pipe_kernel<BM, BN, NS, WGM, NW, T, XA, S2, TSW, XH>), dim3(tilesM * tilesN * sk), dim3(64 * NW), lds, s, q, g_zero_pages[dev]);
}

static std::atomic<int> g_wide_db{0};     // measure builds, knob (sdxl_debug_set "wide_db"): 1 = the wide GEGLU kernel with register-double-buffered fragments
void igemm_set_wide_db(int v) { g_wide_db = v; }
static void launch_wide(const IgemmParams& p, hipStream_t s) {
  // three slots (was four): the k-loop is not latency-bound and the whole ring is requested before the first MFMA -- 108 instead of
  // 144 KiB per CU; same-box library A/B on the bench line 21.19 / 21.12 -> 21.07 / 21.09 ms per step (profiles/r04_ring_depth_lib_ab.txt)
  constexpr int NS = 3;
  const int tilesM = (p.M + 255) / 256, tilesN = (p.N + 319) / 320;
  const size_t lds = (size_t)NS * (256 + 320) * 64 + 2048;   // ring + LayerNorm coefficients
  static bool attr_set[2][kMaxDev] = {};
  const int dev = current_device();
#ifdef SDXL_MEASURE
  // register-double-buffered fragment reads (DB): measured on the bench line A/B/A/B, 21.213 / 21.179 (DB) vs 21.209 / 21.190 ms per step -- no
  // difference (profiles/r05_wide_double_buffer_ab.txt): the k-tile is LDS-bandwidth + matrix-pipe co-bound (112 KiB of fragment reads + 36 KiB of
  // DMA writes = 1156 LDS cycles against 1280 MFMA cycles), not latency-bound.  Measure builds only.
  if (g_wide_db.load()) {
    set_lds_attr(&igemm_wide_kernel<NS, true>, lds, attr_set[1], dev);
    hipLaunchKernelGGL((igemm_wide_kernel<NS, true>), dim3(tilesM * tilesN), dim3(512), lds, s, p, g_zero_pages[dev]);
    return;
  }
#endif
  set_lds_attr(&igemm_wide_kernel<NS, false>, lds, attr_set[0], dev);
  hipLaunchKernelGGL((igemm_wide_kernel<NS, false>), dim3(tilesM * tilesN), dim3(512), lds, s, p, g_zero_pages[dev]);
}


// variant: 0 auto; 1 = 128x128 ring 3; 2 = 128x64 ring 4; 3 = 64x128 ring 4; 4 = 128x128 ring 2; 5 = 128x64 ring 2;
// 6 = 64x128 ring 2; 7 = 128x128 ring 4; 8 = 64x128 ring 3.  Returns false when the shape needs the generic kernel.
static std::atomic<int> g_splitk_wt{1};   // A/B knob (sdxl_debug_set "splitk_wt"): 0 = split-K slabs by plain stores + agent-scope release (the round-2 form)
void igemm_set_splitk_wt(int v) { g_splitk_wt = v; }
static std::atomic<int> g_tsw{1};      // A/B knob (sdxl_debug_set "igemm_tsw"): 0 = the V^T part of a fused QKV projection keeps the LDS-staged transposed epilogue
void igemm_set_tsw(int v) { g_tsw = v; }
static bool g_igemm_unrolled = true;
void igemm_set_unrolled(int v) { g_igemm_unrolled = v != 0; }

// Split-K rule.  It depends on ONE batch entry's shape (rows per entry, N, K) only -- never on the batch size -- so an entry
// comes out bit-identical whether it runs alone or next to others (the CFG pair as one batch-2 forward, split-CFG chains).
int igemm_splitk_slices(const IgemmParams& p) {
  if (!p.splitk_ws || !p.splitk_cnt) return 1;
  if (p.act != 0 || p.n_split < p.N || p.ln_stat) return 1;
  const int nk = p.Kpad / 64;
  // measured (profiles/r02_splitk_sweep.txt): pays from K ~ 11520 (the 32^2 convs: +4 % at K = 11520, +22 % at K = 23040); at
  // K = 5120 (FF-out) the serial combine costs more than the 24 % fewer bytes moved buy (57 vs 44 us), so the bar is 160 k-tiles
  // (round 6: N up to 1536 -- the refiner's 32^2 / 16^2 levels were excluded by the base model's 1280 and ran their K = 13824 ... 27648 convolutions on 12 - 48
  //  workgroups; and EIGHT slices where an entry has one 256-row tile: the 16^2 level of the refiner / of a 512^2 base image streams 42 - 85 MB of weights through
  //  12 x 8 workgroups instead of 12 x 3 -- profiles/r06_refiner_shape_profile.txt)
  if (p.rpb <= 0 || p.rpb > 1024 || p.N > 1536 || nk < 160) return 1;
  return p.rpb <= 256 ? 8 : 3;
}
size_t igemm_splitk_ws_bytes(int batch, int rows_per_entry, int n_max) {
  if (n_max > 1536) n_max = 1536;          // wider outputs never split (igemm_splitk_slices)
  const long rows3 = (long)batch * (rows_per_entry < 1024 ? rows_per_entry : 1024), rows8 = (long)batch * (rows_per_entry < 256 ? rows_per_entry : 256);
  const size_t slabs3 = (size_t)((rows3 + 255) / 256) * 3, slabs8 = (size_t)((rows8 + 255) / 256) * 8;
  return (slabs3 > slabs8 ? slabs3 : slabs8) * (size_t)((n_max + 127) / 128) * 256 * 128 * 4;
}

// Cost of a grid of bm x bn tiles over an M x N output with nk k-tiles (arbitrary units, ~ns).  A workgroup's k-loop time goes
// with the bytes it stages per k-tile, (bm + bn) x 128 -- operand staging, not MFMA issue, bounds these kernels --
// and that holds per CU: a single round costs a full tile time however few CUs it fills (128x128 over 2048 x 1280 = 160
// workgroups takes as long per k-tile as 256 would).  Rounds after the first overlap with their predecessors' tails: a partly
// filled last round costs its fill fraction, but never less than 2/3.  FIXED ~ 8 us of launch / prologue / epilogue per round.
// w = 1.2 for the 8x1-wave 256x160 tile (6 fragment reads per 5 MFMAs), 1.05 for 256x320.
static double tile_cost(int M, int N, int nk, int bm, int bn, double w) {
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const long full = tiles / 256, rem = tiles % 256;
  const double frac = rem == 0 ? 0.0 : (full == 0 ? 1.0 : (rem / 256.0 > 2.0 / 3.0 ? rem / 256.0 : 2.0 / 3.0));
  return ((double)full + frac) * (bm + bn) * nk * w + (double)(full + (rem ? 1 : 0)) * 3000.0;
}

// Tile choice by a two-term cost model fitted to the sweeps (profiles/r01_igemm_sweep.txt, r02_tile_sweep_256x160_256x320.txt,
// r02_tile_96x128.txt) -- tile_cost() above.  What the model buys: N = 320 / 1280 convs at 128^2 / 64^2 get 256x160 tiles =
// exactly one round (conv128 320: 95 -> 69 us, conv64 1280up: 353 -> 235 us), the GEGLU projections the one-round 256x320 tile
// (lin64 geglu 88 -> 77 us, lin32 geglu 67 -> 63 us), and the M = 2048 x N = 1280 linears (attention out / query projections,
// FF-out: 240 launches per step) 96x128 tiles -- 220 workgroups that each stage 12.5 % fewer bytes than the 160 of 128x128
// (19 -> 17 us, 54 -> 47 us), and the M = 8192 x N = 640 shapes of the 64^2 level 128x160 tiles (4 waves, 32x160 wave tiles) = exactly
// 256 workgroups where 256x128 made 160 (conv64 640 82 -> 70 us, 1920>640 243 -> 201, lin64 ff 44 -> 39; profiles/r02_tile_128x160.txt).
// Returns the production variant id (35, 36 / 44, 45, 38, 49, 26).
static int pick_tile(const IgemmParams& p, bool allow_128x160 = true) {
  const int nk = p.Kpad / 64;
  struct Cand { int v, bm, bn; double w; bool ok; };
  const bool lin = p.ksize == 1 && p.stride == 1 && p.up == 0;
  const Cand cands[6] = {
      {35, 256, 128, 1.0, true},
      {36, 128, 128, 1.0, true},
      {45, 96, 128, 1.0, true},
      {38, 256, 160, 1.2, p.N % 160 == 0 && !p.stat_out},
      {49, 128, 160, 1.15, allow_128x160 && p.N % 160 == 0 && !p.stat_out && p.act == 0},
      {26, 256, 320, 1.05, p.act == 1 && lin && p.N % 320 == 0}};
  double best = 1e300;
  int variant = 35;
  for (const Cand& c : cands) {
    if (!c.ok) continue;
    const double cost = tile_cost(p.M, p.N, nk, c.bm, c.bn, c.w);
    if (cost < best) { best = cost; variant = c.v; }
  }
  if (variant == 36 && nk >= 40) variant = 44;   // long contractions: the 5-slot ring (4 tiles in flight) is 3-6 % faster (profiles/r02_ring5_ab.txt)
  return variant;
}

static std::atomic<int> g_hl_tile96{29};     // bit 0: 96x128 for linears, bit 1: ... for 3x3 convs too (not selected), bit 2: 4-wave 128x160 for N % 160 == 0, N % 128 != 0 layers, bit 3: ... wherever the cost model prefers it, bit 4: in-launch split-K for the K >= 10240 convolutions
void igemm_set_hl_tile96(int v) { g_hl_tile96 = v; }
// GroupNorm statistics from the producing GEMM's epilogue (IgemmParams::gn_part): taken by the 256x128 kernel (plain or split-K)
// when that is the tile the selection picks anyway, whole 256-row tiles inside one batch entry, f16 operands, plain epilogue.
bool igemm_gn_part_ok(const IgemmParams& p) {
  if (p.a_dt == DT_HL) {
    // split-operand convolutions (round 6): where the selection runs the 256x128 kernel anyway -- the in-launch split-K of the K >= 10240 convolutions
    // of the 32^2 level (launch_igemm_hl_pipe; one entry's shape only) -- its staged epilogue leaves the statistics of the fp32 rows
    if (p.c_dt != DT_F32 || (p.Cin % 32) != 0 || (p.lda % 4) != 0 || (p.Kpad % 32) != 0) return false;
    if (p.act != 0 || p.n_split < p.N || p.stat_out || p.ln_stat || p.xa_k) return false;
    if (p.M % 256 != 0 || p.rpb <= 0 || p.rpb % 256 != 0 || p.N % 64 != 0) return false;
    if (p.ebias && (p.ebias_ld & 3) != 0) return false;
    if ((p.ldc & 3) != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0) return false;
    return (g_hl_tile96.load() & 16) && igemm_splitk_slices(p) > 1;
  }
  if (p.a_dt != DT_F16 || p.c_dt != DT_F16 || (p.Cin % 64) != 0 || (p.lda % 8) != 0 || (p.Kpad % 64) != 0) return false;
  if (p.act != 0 || p.n_split < p.N || p.stat_out || p.ln_stat || p.xa_k) return false;
  if (p.M % 256 != 0 || p.rpb <= 0 || p.rpb % 256 != 0 || p.N % 64 != 0) return false;
  if (p.ebias && (p.ebias_ld & 3) != 0) return false;
  // Like the split-K rule this must depend on ONE batch entry's shape only -- the statistics path rounds differently from the
  // statistics kernel, and an entry has to come out bit-identical alone, in the CFG pair or in a larger batch.  The tile
  // preference is therefore evaluated for the CFG pair (2 entries), whatever the actual batch.
  IgemmParams q = p;
  q.M = 2 * p.rpb;
  if (igemm_splitk_slices(p) > 1) return true;
  if (pick_tile(q, false) != 35) return false;
  // 128x160 tiles cannot leave the statistics (32x160 wave tiles): keep 256x128 + statistics unless the other tile saves more than
  // the statistics launch it brings back (~13 us ~ 6000 cost units)
  const int nk = p.Kpad / 64;
  const bool t160 = p.N % 160 == 0;
  return !t160 || tile_cost(q.M, q.N, nk, 256, 128, 1.0) - tile_cost(q.M, q.N, nk, 128, 160, 1.15) < 6000.0;
}

bool igemm_xattn_ok(int a_dt, int c_dt, int M, int N, int K, int rpb, int n_ctx) {
  // (c_dt: f16 rows; HL16 rows for the split-precision form behind an fp32-class out-projection -- the row / staged epilogues store either)
  return a_dt == DT_F16 && (c_dt == DT_F16 || c_dt == DT_HL) && M > 0 && N % 64 == 0 && K % 64 == 0 && rpb > 0 && rpb % 64 == 0 && M % rpb == 0 &&
         n_ctx >= 1 && n_ctx <= 96;
}

// Context K [B][n_ctx][C] / V^T [B][C][vt_ld] (f16) -> the operand-order image xattn_inplace reads: per (batch entry, head) 24
// fragments of 64 lanes x 8 halfs -- 12 of K (key tile t, k-step s4: key = 32t + lane&31, d = 16 s4 + 8(e>>2) + 4(lane>>5) + (e&3))
// then 12 of V^T (d tile dt, k-step s6: d = 32dt + lane&31, key = 16 s6 + 8(e>>2) + 4(lane>>5) + (e&3)); keys >= n_ctx are zero.
__global__ void xattn_pack_kernel(const half_t* K, const half_t* Vt, half8* out, int B, int C, int nctx, int vt_ld) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = C >> 6;
  if (i >= (size_t)B * H * 24 * 64) return;
  const int lane = (int)(i & 63), f = (int)((i >> 6) % 24);
  const int bh = (int)(i / (24 * 64)), b = bh / H, h = bh - b * H;
  const int fr = lane & 31, fh = lane >> 5;
  half8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int sub = 8 * (e >> 2) + 4 * fh + (e & 3);
    half_t x = (half_t)0.f;
    if (f < 12) {
      const int key = 32 * (f >> 2) + fr, d = 16 * (f & 3) + sub;
      if (key < nctx) x = K[((size_t)b * nctx + key) * C + h * 64 + d];
    } else {
      const int g = f - 12, d = 32 * (g / 6) + fr, key = 16 * (g % 6) + sub;
      if (key < nctx) x = Vt[((size_t)b * C + h * 64 + d) * vt_ld + key];
    }
    v[e] = x;
  }
  out[i] = v;
}
size_t xattn_pack_bytes(int B, int C) { return (size_t)B * (C / 64) * 24 * 64 * 16; }
void launch_xattn_pack(const void* K, const void* Vt, void* out, int B, int C, int nctx, int vt_ld, hipStream_t s) {
  const size_t n = (size_t)B * (C / 64) * 24 * 64;
  hipLaunchKernelGGL(xattn_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const half_t*>(K),
                     reinterpret_cast<const half_t*>(Vt), reinterpret_cast<half8*>(out), B, C, nctx, vt_ld);
}

bool launch_igemm_wreg(const IgemmParams& p, int variant, hipStream_t s);   // igemm_wreg.hip
bool launch_igemm_glds(const IgemmParams& p, int variant, hipStream_t s) {
  if (!g_zero_pages[current_device()]) return false;
  if (p.act > 1) return false;   // GELU / QuickGELU epilogues (CLIP MLP, once per prompt) live in the generic kernel
  if (p.a_dt != DT_F16 || (p.Cin % 64) != 0 || (p.lda % 8) != 0 || (p.Kpad % 64) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.A) & 15) != 0) return false;
  if (p.n_split < p.N && (p.n_split & 3) != 0) return false;
  if (p.ebias && (p.ebias_ld & 3) != 0) return false;
  if (p.stat_out && (variant == 2 || variant == 5 || variant == 19 || variant == 38)) return false;   // wave tiles narrower / other than 64 columns
  // the DMA reads weight rows up to the tile edge: Npad is a multiple of 128 for every packed weight (pack_* kernels)
  const bool was_auto = variant == 0;
  IgemmParams psk = p;
  psk.splitk = 1;
  psk.splitk_wt = g_splitk_wt.load();
  if (p.xa_k) {
    // fused cross-attention: wave tiles of 64 columns only (256x128 / 128x128), chosen by the same cost model
    if (p.act != 0 || p.n_split < p.N || p.stat_out || p.R || p.ebias ||
        !igemm_xattn_ok(p.a_dt, p.c_dt, p.M, p.N, p.K, p.rpb, p.xa_nctx))
      throw std::runtime_error("igemm: fused cross-attention needs a plain f16 projection (no residual / split outputs)");
    if (!p.xa_k_lo && p.c_dt != DT_F16) throw std::runtime_error("igemm: the f16 fused cross-attention writes f16 rows");
    const int nk = p.Kpad / 64;
    if (p.xa_k_lo) {     // split precision: the one-MFMA-row tiles only (96x128 / 128x128), same cost model
      const double c128 = tile_cost(p.M, p.N, nk, 128, 128, 1.0), c96 = tile_cost(p.M, p.N, nk, 96, 128, 1.0);
      if (c96 < c128) launch_pipe<96, 128, 3, 3, 6, half_t, true, false, false, true>(psk, s);
      else launch_pipe<128, 128, 4, 4, 8, half_t, true, false, false, true>(psk, s);
      return true;
    }
    int v = variant;
    if (v != 35 && v != 36 && v != 44 && v != 45) {
      const double c256 = tile_cost(p.M, p.N, nk, 256, 128, 1.0), c128 = tile_cost(p.M, p.N, nk, 128, 128, 1.0), c96 = tile_cost(p.M, p.N, nk, 96, 128, 1.0);
      v = c256 <= c128 && c256 <= c96 ? 35 : (c96 < c128 ? 45 : (nk >= 40 ? 44 : 36));
    }
    if (v == 35) launch_pipe<256, 128, 3, 4, 8, half_t, true>(psk, s);
    else if (v == 36) launch_pipe<128, 128, 4, 4, 8, half_t, true>(psk, s);
    else if (v == 44) launch_pipe<128, 128, 5, 4, 8, half_t, true>(psk, s);
    else launch_pipe<96, 128, 3, 3, 6, half_t, true>(psk, s);     // (three slots, was five: 21.47 -> 21.39 / 21.38 ms per step with three / four, profiles/r04_ring_depth_lib_ab.txt)
    return true;
  }
  if (p.gn_part) {
    if (!igemm_gn_part_ok(p)) throw std::runtime_error("igemm: GroupNorm statistics requested from a shape the 256x128 epilogue does not take");
    variant = 0;            // (a forced test variant must not drop the statistics)
  }
  // plain linear layers / 1x1 convs whose weights also exist in fragment order: the weights-in-registers kernel (igemm_wreg.hip).
  // The rule is static per layer (never the batch), so its k-summation order (even + odd k-tiles) is what such a layer always gets.
  if (p.shadow) {      // f16 shadow of an fp32 output: written by the weights-in-registers epilogue only (run_conv asks igemm_wreg_selected before it sets the field)
    if (!launch_igemm_wreg(psk, 0, s)) throw std::runtime_error("igemm: an f16 shadow output needs the weights-in-registers kernel");
    return true;
  }
  if ((variant == 0 || (variant >= 60 && variant <= 77)) && launch_igemm_wreg(psk, variant, s)) return true;
  if (variant >= 60 && variant <= 77) return false;
  if (variant == 0 && igemm_splitk_slices(p) > 1) {
    // long contractions over a small output (FF-out and the 32^2 convs of the CFG pair: M = 2048, N = 1280 is 80 tiles of
    // 256x128 on 256 CUs): three k-slices per tile fill the chip with the tile shape that moves the fewest bytes per flop
    psk.splitk = igemm_splitk_slices(p);
    launch_pipe<256, 128, 3, 4, 8>(psk, s);
    return true;
  }
  if (variant == 0) variant = p.gn_part ? 35 : pick_tile(p);
#ifdef SDXL_MEASURE
  if (was_auto && !g_igemm_unrolled) {   // A/B against the rolled loops (profiles/r01_igemm_unrolled_ab.txt)
    if (variant == 35) variant = 11; else if (variant == 36) variant = 13; else if (variant == 38) variant = 19;
  }
#else
  (void)was_auto;
#endif
  switch (variant) {
    // ---- production kernels (what the auto selection launches)
    case 4: launch_glds<128, 128, 2>(psk, s); break;                          // 4 waves, 2-3 co-resident blocks: ragged multi-round grids
    case 6: launch_glds<64, 128, 2>(psk, s); break;
    case 35:    // 8 waves, hand-ordered k-loop unrolled by the ring depth; outputs with a transposed part (fused QKV: V^T) take the operand-swap twin
      if (p.n_split < p.N && g_tsw.load() && p.c_dt == DT_F16) launch_pipe<256, 128, 3, 4, 8, half_t, false, false, true>(psk, s);      // (the operand-swapped V^T epilogue writes f16 rows)
      else launch_pipe<256, 128, 3, 4, 8>(psk, s);
      break;
    case 36: launch_pipe<128, 128, 4, 4, 8>(psk, s); break;
    case 38:                                                                // 256x160 GEGLU tile (8x1 waves)
      if (p.N % 160 != 0) return false;
      launch_pipe<256, 160, 3, 8, 8>(psk, s); break;
    case 44: launch_pipe<128, 128, 5, 4, 8>(psk, s); break;   // 5-slot ring = all 160 KiB of LDS: 4 tiles in flight
    case 45: launch_pipe<96, 128, 5, 3, 6>(psk, s); break;    // 6 waves (3 x 2), 96-row tile: M = 2048 x N = 1280 -> 220 workgroups
    case 46: launch_pipe<96, 128, 4, 3, 6>(psk, s); break;
    // 45 with ONE rendezvous per two k-tiles (S2).  Alone, on L2-resident operands, 2 - 4 % faster on every shape of the step; inside
    // the step, where the weights stream in cold, the tile less in flight costs more than the rendezvous saved: GEMM class 20.95 -
    // 21.04 vs 20.76 - 20.82 ms on one box (profiles/r03_two_tiles_per_rendezvous.txt).  Kept as the A/B partner, not selected.
    case 47: launch_pipe<96, 128, 5, 3, 6, half_t, false, true>(psk, s); break;
    case 49:                                                                // 4 waves 4x1 (32x160 wave tiles): N = 640 at 64^2 -> exactly 256 tiles
      if (p.N % 160 != 0 || p.stat_out) return false;
      launch_pipe<128, 160, 3, 4, 4>(psk, s); break;
    case 26:                                                                // 256x320, k-tile 32: linear GEGLU projections only
      if (p.act != 1 || p.ksize != 1 || p.stride != 1 || p.up != 0 || p.N % 320 != 0 || p.Kpad % 32 != 0) return false;
      launch_wide(psk, s); break;
#ifdef SDXL_MEASURE
    default: return launch_igemm_measure(psk, variant, s);   // A/B partners, measurement modes (igemm_measure.hip)
#else
    default: return false;
#endif
  }
  return true;
}

// Strict-fp32 mode on the same direct-to-LDS pipeline (the VAE at the reference's precision, sample/main.rs:121,273, and the
// parity configuration of the UNet): fp32 operands staged as 128-byte rows of 32 k-values, four v_mfma_f32_32x32x2_f32 per
// fragment pair.  The f32 MFMA runs at 1/16 of the f16 rate, so these launches are matrix-pipe bound and the tile choice
// only has to keep the rounds of 256 CUs full.  Returns false for shapes the generic kernel must take.
bool launch_igemm_f32_pipe(const IgemmParams& p, hipStream_t s) {
  if (!g_zero_pages[current_device()]) return false;
  if (p.act > 1 || p.ln_stat || p.stat_out) return false;
  if (p.a_dt != DT_F32 || (p.Cin % 32) != 0 || (p.lda % 4) != 0 || (p.Kpad % 32) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.A) & 15) != 0) return false;
  if (p.n_split < p.N && (p.n_split & 3) != 0) return false;
  if (p.ebias && (p.ebias_ld & 3) != 0) return false;
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  const double eff128 = (double)t128 / (double)(((t128 + 255) / 256) * 256);
  const double eff256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
  IgemmParams q = p;
  q.splitk = 1;
  if (eff256 >= eff128) launch_pipe<256, 128, 3, 4, 8, float>(q, s);
  else launch_pipe<128, 128, 4, 4, 8, float>(q, s);
  return true;
}

// Split-operand mode (DT_HL; igemm_common.h): HL16 operands on the same direct-to-LDS pipeline, 3 f16 MFMAs per 16-deep product.
// Returns false for shapes the generic kernel must take (none in the VAE: its Cin % 32 != 0 layers are packed fp32).
bool launch_igemm_hl_pipe(const IgemmParams& p, hipStream_t s) {
  if (!g_zero_pages[current_device()]) return false;
  if (p.act > 1 || p.ln_stat || p.stat_out || p.xa_k) return false;
  if (p.a_dt != DT_HL || (p.Cin % 32) != 0 || (p.lda % 4) != 0 || (p.Kpad % 32) != 0) return false;
  if (p.gn_part && !igemm_gn_part_ok(p)) throw std::runtime_error("igemm: GroupNorm statistics requested from a split-operand shape the 256x128 split-K epilogue does not take");
  if ((reinterpret_cast<uintptr_t>(p.A) & 15) != 0) return false;
  if (p.n_split < p.N && (p.n_split & 3) != 0) return false;
  if (p.c_dt == DT_HL) {
    // HL16 outputs are written as whole 8-column (8-key) pieces: plain outputs with aligned rows, a fully transposed one (the
    // VAE's V^T) whose batch entries are whole pieces, or a split at a multiple of 128 columns (the UNet's q | k | V^T: a wave
    // tile is then wholly on one side); ragged splits go out as fp32
    if (p.n_split < p.N && p.n_split != 0 && ((p.n_split & 127) != 0 || p.act != 0)) return false;
    const int nout = p.act == 1 ? (p.N >> 1) : p.N;
    if (p.n_split != 0 && ((nout & 7) != 0 || (p.ldc & 15) != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0)) return false;
    if (p.n_split < p.N && ((p.ct_ld & 15) != 0 || (p.rpb & 7) != 0 || (p.M % 8) != 0 || (reinterpret_cast<uintptr_t>(p.Ct) & 15) != 0)) return false;
  }
  if (p.ebias && (p.ebias_ld & 3) != 0) return false;
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128);
  const double eff128 = (double)t128 / (double)(((t128 + 255) / 256) * 256);
  const double eff256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
  IgemmParams q = p;
  q.splitk = 1;
  // round 5: the 96-row tile of the f16 engine for the shapes where it fills more of the chip's single round -- the M = 2048 x N = 1280 linears of the
  // 32^2 level (out-projections, cross-attention query projection, FF-out: 220 workgroups instead of 160).  Same k order in every tile shape, so the
  // choice never changes a result bit (A/B knob: sdxl_debug_set "hl_tile96").
  const long t96 = (long)((p.M + 95) / 96) * ((p.N + 127) / 128);
  const int t96mode = g_hl_tile96.load();      // 1: linear layers / 1x1 only (default), 2: 3x3 convolutions too (measured -0.15 % on the mixed mode's step: not selected)
  // (never a shape the in-launch split-K below would take: split-K depends on ONE entry's shape, this tile choice on the batched M -- testing it first
  //  would give a K >= 10240 linear layer one summation order at B <= 2 and another at B >= 3)
  if ((t96mode & 1) && (p.ksize == 1 || (t96mode & 2)) && p.n_split >= p.N && t128 < 256 && t96 <= 256 && t96 > t128 && eff256 < (double)t96 / 256.0 &&
      !((t96mode & 16) && igemm_splitk_slices(p) > 1)) {
    launch_pipe<96, 128, 5, 3, 6, hl16_t>(q, s);
    return true;
  }
  // (a 256x160 HL tile for the N = 320 convolutions of the 128^2 level -- no 17 % of column padding -- spills DMA pointers inside its k-loop: scratch
  //  loads in the VM queue break the hand-counted vmcnt waits.  Not instantiated.)
  // long contractions over a small output (the K >= 10240 convolutions of the 32^2 level: 80 tiles of 256x128): three k-slices per tile combined inside the
  // launch, as in the f16 engine (igemm_splitk_slices: depends on one batch entry's shape only; slabs summed in slice order: bit-reproducible) -- 240
  // workgroups instead of the 160 of 128x128 tiles.  Mixed-mode step 41.18 / 41.22 -> 40.60 / 40.56 ms, A/B/A/B; F32_SPLIT forward 2.16e-6 (was 2.12e-6),
  // config-2 final latent 2.4e-4 (2.1e-4) from the oracle (profiles/r05_hl_tile160_ab.txt; knob hl_tile96 bit 4)
  if ((t96mode & 16) && igemm_splitk_slices(p) > 1) {
    q.splitk = igemm_splitk_slices(p);
    q.splitk_wt = g_splitk_wt.load();
    launch_pipe<256, 128, 3, 4, 8, hl16_t>(q, s);
    return true;
  }
  // the 4-wave 128x160 tile (one wave per SIMD, up to 512 registers each) for layers whose width is a multiple of 160 but not of 128 -- the N = 320
  // convolutions of the 128^2 level: two exact column tiles instead of three 128-wide ones with 17 % of padding.  Mixed-mode step 44.58 / 44.36 ->
  // 44.11 / 44.02 ms, A/B/A/B, results bit-identical (profiles/r05_hl_tile160_ab.txt; knob hl_tile96 bit 2)
  if ((t96mode & 4) && p.N % 160 == 0 && p.N % 128 != 0 && p.n_split >= p.N && !p.stat_out && p.act == 0) {
    launch_pipe<128, 160, 3, 4, 4, hl16_t>(q, s);
    return true;
  }
  // ... and where it fills the chip's rounds better than the 128-wide tiles (cost model of the f16 selection, tile_cost): the M = 8192 x N = 640 shapes of
  // the 64^2 level are exactly 256 tiles of 128x160 where 256x128 makes 160 and 128x128 320.  Mixed-mode step 41.14 / 41.32 -> 40.46 / 40.49 ms, A/B/A/B, bit-identical
  // (profiles/r05_hl_tile160_ab.txt; knob hl_tile96 bit 3)
  if ((t96mode & 8) && p.N % 160 == 0 && p.n_split >= p.N && !p.stat_out && p.act == 0) {
    const int nk = p.Kpad / 32;
    const double c160 = tile_cost(p.M, p.N, nk, 128, 160, 1.15);
    const double cbest = std::min(tile_cost(p.M, p.N, nk, 256, 128, 1.0), tile_cost(p.M, p.N, nk, 128, 128, 1.0));
    if (c160 < cbest) { launch_pipe<128, 160, 3, 4, 4, hl16_t>(q, s); return true; }
  }
  if (eff256 >= eff128) launch_pipe<256, 128, 3, 4, 8, hl16_t>(q, s);
  else launch_pipe<128, 128, 4, 4, 8, hl16_t>(q, s);
  return true;
}

}  // namespace sdxl
