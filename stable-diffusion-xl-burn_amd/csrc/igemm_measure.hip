// Measurement build only (build.py --measure, -DSDXL_MEASURE): the A/B partners, dead-end variants and measurement modes of
// the implicit-GEMM kernels -- rolled k-loops, other rings, DMA issue modes (DMODE 1..7: results WRONG by construction for
// 2, 3, 5, 6, 7), L2-prefetch touches (PF), s_setprio (PRIO), loader-wave specialisation (igemm_ws_kernel).  None of this is
// in the release library: the production kernels (igemm_glds.hip) carry only the schedule that ships.  The numbers these
// variants produced are under profiles/ (r01_igemm_dma_modes.txt, r02_l2_prefetch_ab.txt, r02_ring5_ab.txt, ...).
#include "igemm_common.h"

#ifdef SDXL_MEASURE
namespace sdxl {

template <int BM, int BN, int NS, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void igemm_glds_m_kernel(const IgemmParams p, const void* zeros) {   // >= MINB blocks per CU
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

// DMODE: 0 = pieces of tile kt+NS-1 spread over the first three kk-steps of k-tile kt (slot freed by the previous barrier);
//        1 = pieces of tile kt+NS issued right after the barrier of k-tile kt, between the MFMAs of its fourth kk-step
//            (slot freed by THIS barrier; prefetch distance ~NS-1 full k-tiles instead of ~NS-2 + 1/3);
//        2 = measurement only: as 1 but the DMA sources never advance along k (every k-tile re-reads the block's first
//            one from L2) -- the compute-only ceiling of the loop.  Results are wrong by construction.
//        3 = measurement only: as 1 but NO DMA is issued inside the loop at all (ds_read + MFMA + barrier only).
// WGM = waves along M (4: 4x2 wave grid, 8: 8x1 -- the 256x160 GEGLU tile: N = 10240 / 5120 gives 512 / 1024 tiles = whole
// rounds of 256 CUs where 256x128 leaves the last round 44 % empty).  BN need not be a multiple of 64: the weight tile's
// BN/8 eight-row pieces are dealt round-robin, waves below REM carry one more piece and wait on their own count.
// PF > 0 (linear layers): L2 PREFETCH PF k-tiles ahead of the DMA.  Measured (tools/l2_probe.hip, profiles/r02_l2_probe.txt): a CU
// pulls L2-RESIDENT data at ~145 GB/s through this same LDS-DMA path, yet the k-loop only streams ~45 GB/s per CU -- every
// workgroup of an XCD asks for a new operand line at about the same time, so nobody finds it in the L2: all of them wait out
// the Infinity-Cache / HBM latency (~2 us under load) with only NS-1 tiles in flight.  So each wave touches, one k-tile-row
// line per lane (a 4-byte LDS-DMA into a scratch slot: no register, no compiler-visible hazard), the lines the DMA will ask
// for PF k-tiles later; by then they are L2 hits.
// TL (timeline): the production schedule (UNR, DMODE 0) with s_memtime stamps -- per wave: T0 kernel entry, T1 prologue DMA issued,
// T2 first barrier passed, per k-tile {A: arrived at the DMA wait, B: own pieces of tile kt+1 landed, C: barrier passed}, T3 k-loop
// done, T4 epilogue done.  Stamps are parked in LDS behind the ring (ds_write: no vmcnt traffic) and dumped at the end to
// g_tl_buf[workgroup][wave][kTlWords] (igemm_set_timeline).  s_memtime returns through lgkmcnt, so the A stamp is taken behind a
// lgkmcnt(0) that the schedule itself only issues a few instructions later: ~3 scalar round trips per k-tile of overhead.
constexpr int kTlTiles = 116, kTlWords = 3 * kTlTiles + 16;   // words 0..15: phase stamps + meta, then 3 per k-tile
__device__ unsigned* g_tl_buf = nullptr;
template <int BM, int BN, int NS, bool PRIO, int DMODE = 0, int WGM = 4, int NW = 8, bool UNR = false, typename T = half_t, int PF = 0, bool XA = false, bool TL = false>
__global__ __launch_bounds__(64 * NW) void igemm_pipe_m_kernel(const IgemmParams p, const void* zeros) {
  kernarg_prefetch<(int)sizeof(IgemmParams) + 8>();   // every argument line in flight at once (one wait instead of five)
  typedef typename PipeElem<T>::frag frag_t;
  constexpr int CE = 16 / (int)sizeof(T);     // elements per 16-byte chunk: 8 (f16) or 4 (f32, strict mode)
  static_assert(sizeof(T) == 2 || DMODE == 0, "measurement modes exist for the f16 kernel only");
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
  // measurement-only modes (results wrong by construction): 5 = schedule of mode 0 WITHOUT ds_reads / MFMAs (DMA-only
  // ceiling), 6 = 5 with every DMA piece reading 1 KiB CONTIGUOUS (operands as if pre-tiled [rows/8][K/64][8][64]),
  // 7 = mode 0 (full compute) with the contiguous sources of 6
  constexpr bool SCHED0 = DMODE == 0 || DMODE >= 5;
  constexpr bool NOMMA = DMODE == 5 || DMODE == 6;
  constexpr bool CONTIG = DMODE == 6 || DMODE == 7;
  constexpr int WADV = CONTIG ? 512 : KT;
  static_assert(NS >= 3, "counted-wait pipeline needs a ring of at least 3 slots");
  static_assert(BM * 128 + (TN - 1) * 4096 < 65536, "fragment offsets must fit the ds_read immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const bool lastb = REM == 0 || wave < REM;  // this wave carries weight piece BJ-1
  typedef __attribute__((address_space(3))) volatile unsigned* lds_u32_t;
  lds_u32_t tl = (lds_u32_t)(lptr_t)(smem + pipe_lds_total(NS * (BM + BN) * 128, PF > 0 ? NW * 256 : 0)) + wave * kTlWords;
  auto stamp = [&](int idx) {
    if constexpr (TL) {
      const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
      if (lane == 0 && idx < kTlWords) tl[idx] = t;
    }
  };
  stamp(0);

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
  if constexpr (TL) { asm volatile("" :: "s"(m0), "s"(n0)); stamp(8); }   // kernel arguments have arrived
  // counted DMA wait: K tiles of this wave's pieces may stay in flight
  auto wait_tiles = [&](auto KK) {
    constexpr int k = decltype(KK)::value;
    if constexpr (REM == 0) wait_vmcnt<PER * k>();
    else { if (lastb) wait_vmcnt<PER * k>(); else wait_vmcnt<(PER - 1) * k>(); }
  };
  static_assert(PF == 0 || (REM == 0 && UNR && DMODE == 0 && NW * 64 >= BM + BN), "L2 prefetch: unrolled production kernels only");

  // ---- DMA geometry: piece j of this wave covers tile rows (j*8 + wave)*8 .. +7; lane -> (row, slot)
  const int lrow = lane >> 3, slot = lane & 7;
  const int HWo = p.Hout * p.Wout;
  const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
  const bool lin_rows = p.ksize == 1 && p.stride == 1 && p.up == 0 && p.pad == 0 && !CONTIG;   // as the production kernel
  const bool pow2 = (HWo & (HWo - 1)) == 0 && (p.Wout & (p.Wout - 1)) == 0;
  const int sh_hw = __builtin_ctz((unsigned)HWo | 0x40000000u), sh_w = __builtin_ctz((unsigned)p.Wout | 0x40000000u);
  int rb[AJ], ry[AJ], rx[AJ], rsw[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (j * NW + wave) * 8 + lrow;
    const int m = m0 + row;
    rsw[j] = (slot ^ ((row >> 1) & 7)) * CE;
    rb[j] = m < p.M ? m : -1; ry[j] = 0; rx[j] = 0;
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
  const T* aptr[AJ];
  int aadv[AJ];
  int s_c0 = 0, s_dy = 0, s_dx = 0;
  if (kbeg > 0) {   // start the tap walk inside the contraction
    const int e0 = kbeg * KT, tap = e0 / p.Cin;
    s_c0 = e0 - tap * p.Cin; s_dy = tap / p.ksize; s_dx = tap - s_dy * p.ksize;
#pragma unroll
    for (int j = 0; j < BJ; ++j) wptr[j] += e0;
  }
  stamp(9);    // DMA geometry (row -> pixel divisions) done
  auto retap = [&]() {
    if (lin_rows) {
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
  stamp(10);   // first tap's source pointers
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
  // L2 prefetch: line L = wave * 64 + lane of the tile's BM activation rows then BN weight rows (one 128-byte line per k-tile)
  const T* pfp = reinterpret_cast<const T*>(zeros);
  int pfadv = 0;
  if constexpr (PF > 0) {
    const int L = wave * 64 + lane;
    if (p.ksize == 1 && p.stride == 1 && p.up == 0) {
      if (L < BM) { if (m0 + L < p.M) { pfp = Ag + (size_t)(m0 + L) * p.lda; pfadv = KT; } }
      else if (L < BM + BN) { pfp = reinterpret_cast<const T*>(p.W) + (size_t)(n0 + L - BM) * p.Kpad; pfadv = KT; }
    }
    if (pfadv) pfp += (size_t)(kbeg + NS - 1 + PF) * KT;
  }
  if constexpr (CONTIG) {
    const int nkc = p.Kpad / KT;
#pragma unroll
    for (int j = 0; j < AJ; ++j) { aptr[j] = Ag + (size_t)((m0 >> 3) + j * NW + wave) * nkc * 512 + lane * 8; aadv[j] = 512; }
#pragma unroll
    for (int j = 0; j < BJ; ++j) wptr[j] = reinterpret_cast<const T*>(p.W) + (size_t)((n0 >> 3) + j * NW + wave) * nkc * 512 + lane * 8;
  }
  // pieces q of one k-tile: q < AJ -> activation piece q, else weight piece q - AJ.  PH selects the pieces with q % 3 == PH
  // (PH < 0: all of them); the tap walk advances once per k-tile, after the last piece (tile_done).
  auto issue = [&](int buf, auto PH) {
    constexpr int ph = decltype(PH)::value;
    char* la = smem + buf * STAGE + wave * 1024;
    char* lb = la + BM * 128;
    static_for<PER>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      constexpr int NM = TM * TN;                                   // MFMAs per kk-step
      constexpr int PPG = (PER + (NM > 1 ? NM - 2 : 0)) / (NM > 1 ? NM - 1 : 1);   // pieces per MFMA gap (early mode)
      constexpr int HALF = (PER + 1) / 2;
      if constexpr (ph < 0 || (ph < 3 && q % 3 == ph) || (ph >= 10 && q / PPG == ph - 10) || (ph == 5 && q < HALF) ||
                    (ph == 6 && q >= HALF)) {
        if constexpr (LIN) {
          // saddr form: global_load_lds_dwordx4 voffset, sbase -- M0 = LDS byte address of this wave's 1-KiB piece
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
          if constexpr (DMODE != 2) aptr[q] += aadv[q];
        } else if (q - AJ < BJ - 1 || lastb) {     // ragged weight tile: wave-uniform predicate on the last piece
          __builtin_amdgcn_global_load_lds((gptr_t)wptr[q - AJ], (lptr_t)(lb + (q - AJ) * (NW * 1024)), 16, 0, 0);
          if constexpr (DMODE != 2) wptr[q - AJ] += WADV;
        }
      }
    });
  };
  auto tile_done = [&]() {
    if constexpr (DMODE == 2 || CONTIG) return;
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
  frag_t fA[DMODE == 4 ? 4 : 2][TM], fB[DMODE == 4 ? 4 : 2][TN];
  auto ldfrag = [&](unsigned so, int kk, auto SET) {
    constexpr int set = decltype(SET)::value;
    const unsigned aa = (basea ^ (kk << 5)) + so, ab = (baseb ^ (kk << 5)) + so;
    if constexpr (NOMMA) return;
    static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<decltype(I)::value * 4096, frag_t>(aa); });
    static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<decltype(J)::value * 4096, frag_t>(ab); });
  };
  // MFMAs of one kk-step from fragment set SET; DMA pieces PH (or none, PH = 3) are issued between them
  auto mma = [&](auto SET, int buf, auto PH, bool more) {
    constexpr int set = decltype(SET)::value;
    constexpr int ph = decltype(PH)::value;
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (!NOMMA) acc[0][0] = PipeElem<T>::mma(fB[set][0], fA[set][0], acc[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ph < 3) {
      if (more) issue(buf, PH);            // wave-uniform branch around the DMA pieces only, never around MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ph == 5 || ph == 6) {    // lookahead-2 mode: half of the next tile's pieces behind the first MFMA
      if (more) issue(buf, PH);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ph == 4) {               // early mode: gap 0 pieces here, gap g pieces after MFMA g
      if (more) issue(buf, std::integral_constant<int, 10>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<TM * TN - 1>([&](auto X) {
      constexpr int x = decltype(X)::value + 1, i = x / TN, j = x % TN;
      if constexpr (!NOMMA) acc[i][j] = PipeElem<T>::mma(fB[set][j], fA[set][i], acc[i][j]);
      if constexpr (ph == 4 && x < TM * TN - 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(buf, std::integral_constant<int, 10 + x>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using IALL = std::integral_constant<int, -1>;
  using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>;
  constexpr int NPRO = SCHED0 ? NS - 1 : NS;   // tiles staged by the prologue

  // fused cross-attention, one-MFMA-row wave tiles: the 24 context fragments (96 VGPRs -- these kernels have the room) are
  // requested BEFORE the first DMA piece, so they are the oldest entries of the in-order vmcnt queue and ride under the
  // prologue's wait for tile 0 instead of adding a memory round trip to the epilogue
  constexpr bool XA_EARLY = XA && TM == 1;
  half8 xkf[XA ? 3 : 1][4], xvf[XA ? 2 : 1][6];
  if constexpr (XA_EARLY) xattn_load_frags(p, m0 + wm * WM, n0 + wn * WN, lane, xkf, xvf);
  // folded LayerNorm: the tile's row coefficients, evaluated once per workgroup (LnCoop) where 2 KiB of LDS are left behind the ring
  typedef LnCoop<BM, 64 * NW> LnC;
  constexpr bool LN_COOP = LnC::OK && pipe_lds_total(NS * STAGE, PF > 0 ? NW * 256 : 0) > NS * STAGE + (PF > 0 ? NW * 256 : 0);
  float* ln_coef = reinterpret_cast<float*>(smem + NS * STAGE + (PF > 0 ? NW * 256 : 0));
  LnC lnc;
  if constexpr (LN_COOP) { lnc.load(p, m0, tid); __builtin_amdgcn_sched_barrier(0); }
  stamp(11);   // accumulators zeroed, fragment addresses, LayerNorm statistics requested
  // ---- prologue: tiles 0 .. NPRO-1 in flight, wait for tile 0 only
#pragma unroll
  for (int s = 0; s < NPRO; ++s)
    if (s < nk) { issue(s, IALL{}); tile_done(); }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  stamp(1);
  float lnA[TM], lnC[TM];
  const bool ln_coop = LN_COOP && p.ln_slots <= 24;
  if constexpr (LN_COOP) lnc.finish(p, m0, ln_coef);
  if (!ln_coop) ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  if (NPRO <= nk) wait_tiles(std::integral_constant<int, NPRO - 1>{}); else wait_vmcnt<0>();
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
  stamp(2);
  ldfrag(0, 0, I0{});
  int cur = 0;                      // ring slot of tile kt
  int fill = NS - 1;                // ring slot tile kt+NS-1 goes to (the slot tile kt-1 occupied)
  if constexpr (UNR) {
    // The schedule of DMODE 0 with the k-loop unrolled by the ring depth: ring slots become compile-time constants, so every
    // fragment read is {one of 16 loop-invariant lane addresses} + immediate and the DMA destinations fold into M0
    // constants -- the rolled loop re-derives them with ~25 VALU / SALU instructions per k-tile, and instruction issue (not
    // LDS or DMA bandwidth) is what fills this kernel's SIMDs.  ds_read immediates are 16 bit: slots
    // beyond 64 KiB go through a second address set (+ 65536).
    static_assert(DMODE == 0, "unrolled ring: production schedule only");
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
    auto ldf = [&](auto SO, auto KK, auto SET) {
      constexpr unsigned so = decltype(SO)::value;
      constexpr int kk = decltype(KK)::value, set = decltype(SET)::value;
      constexpr int hi = PERSLOT ? (int)(so / STAGE) : (so >= 65536u ? 1 : 0);
      constexpr unsigned lo = PERSLOT ? 0u : so - hi * 65536u;
      static_assert(lo + (TM - 1) * 4096 < 65536u && lo + (TN - 1) * 4096 < 65536u, "fragment immediate out of range");
      static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<lo + decltype(I)::value * 4096, frag_t>(fa[hi][kk]); });
      static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<lo + decltype(J)::value * 4096, frag_t>(fb[hi][kk]); });
    };
    auto ktile = [&](int kt, auto CUR) {
      constexpr int c = decltype(CUR)::value;
      constexpr int nslot = (c + 1) % NS, fl = (c + NS - 1) % NS;
      using SO = std::integral_constant<unsigned, (unsigned)c * STAGE>;
      using SN = std::integral_constant<unsigned, (unsigned)nslot * STAGE>;
      const bool more = kt + NS - 1 < nk;
      ldf(SO{}, I1{}, I1{});
      wait_lgkmcnt<NF>();
      mma(I0{}, fl, I0{}, more);
      ldf(SO{}, I2{}, I0{});
      wait_lgkmcnt<NF>();
      mma(I1{}, fl, I1{}, more);
      ldf(SO{}, I3{}, I1{});
      wait_lgkmcnt<NF>();
      mma(I0{}, fl, I2{}, more);
      if constexpr (PF > 0) {
        if (more) {   // one more VM op per tile and wave: the line touches of tile kt + NS - 1 + PF (zero page beyond the end)
          const T* q = kt + NS - 1 + PF < nk ? pfp : reinterpret_cast<const T*>(zeros);
          __builtin_amdgcn_global_load_lds((gptr_t)q, (lptr_t)(smem + NS * STAGE + wave * 256), 4, 0, 0);
          pfp += pfadv;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more) tile_done();
      if (kt + 1 < nk) {
        if constexpr (PF > 0) {
          // in flight stay tiles kt+2 .. kt+NS-1: NS-2 tiles of PER pieces, each loop-issued one with its prefetch op
          if (!more) wait_vmcnt<0>();
          else if (NS == 3 || kt >= NS - 3) wait_vmcnt<(PER + 1) * (NS - 2)>();
          else wait_vmcnt<PER * (NS - 2) + 1>();
        } else {
          if constexpr (TL) { wait_lgkmcnt<0>(); stamp(16 + 3 * kt); }
          if (more) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
          stamp(17 + 3 * kt);
        }
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        stamp(18 + 3 * kt);
        ldf(SN{}, I0{}, I0{});
      } else {
        wait_lgkmcnt<0>();
      }
      mma(I1{}, fl, I3{}, false);
    };
    int kt = 0;
    for (; kt + NS <= nk; kt += NS) static_for<NS>([&](auto S) { ktile(kt + decltype(S)::value, S); });
    static_for<NS - 1>([&](auto S) { if (kt + decltype(S)::value < nk) ktile(kt + decltype(S)::value, S); });
  } else if constexpr (DMODE == 4) {
    // lookahead-2 schedule: one fragment set per kk-step, the ds_reads of step kk+2 are issued before the MFMAs of step kk,
    // so an LDS stall of a whole kk-step (DMA write bursts into the same LDS) does not starve the matrix pipe.  The
    // barrier moves between steps 1 and 2 (all reads of tile kt are issued by then); behind it: the first two fragment
    // sets of tile kt+1 and the DMA pieces of tile kt+NS (slot just freed), half behind each of the last two steps.
    ldfrag(0, 1, I1{});
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned so = cur * STAGE;
      const int nslot = cur + 1 == NS ? 0 : cur + 1;
      const bool more1 = kt + NS < nk;
      ldfrag(so, 2, I2{});
      wait_lgkmcnt<2 * NF>();
      mma(I0{}, cur, I3{}, false);
      ldfrag(so, 3, I3{});
      wait_lgkmcnt<2 * NF>();
      mma(I1{}, cur, I3{}, false);
      wait_lgkmcnt<0>();                        // own reads of tile kt complete
      const bool has_next = kt + 1 < nk;        // (MFMAs stay outside the branches: hipcc would clone the accumulators)
      if (has_next) {
        if (kt + NS - 1 < nk) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(nslot * STAGE, 0, I0{});
      }
      mma(I2{}, cur, I5{}, more1);
      if (has_next) ldfrag(nslot * STAGE, 1, I1{});
      mma(I3{}, cur, I6{}, more1);
      if (more1) tile_done();
      cur = nslot;
    }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned so = cur * STAGE;
    const int nslot = cur + 1 == NS ? 0 : cur + 1;
    const bool more = kt + NS - 1 < nk;           // tile kt+NS-1 exists (uniform): DMODE 0 stages it during steps 0..2
    const bool more1 = DMODE == 3 ? false : kt + NS < nk;   // tile kt+NS exists: DMODE 1 stages it after this k-tile's barrier
    ldfrag(so, 1, I1{});
    wait_lgkmcnt<NF>();
    if constexpr (SCHED0) mma(I0{}, fill, I0{}, more); else mma(I0{}, fill, I3{}, false);
    ldfrag(so, 2, I0{});
    wait_lgkmcnt<NF>();
    if constexpr (SCHED0) mma(I1{}, fill, I1{}, more); else mma(I1{}, fill, I3{}, false);
    ldfrag(so, 3, I1{});
    wait_lgkmcnt<NF>();
    if constexpr (SCHED0) { mma(I0{}, fill, I2{}, more); if (more) tile_done(); } else mma(I0{}, fill, I3{}, false);
    if (kt + 1 < nk) {
      // own pieces of tile kt+1 landed (tiles kt+2 .. kt+NS-1 may stay in flight); own reads of tile kt complete
      if (more) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
      wait_lgkmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ldfrag(nslot * STAGE, 0, I0{});
    } else {
      wait_lgkmcnt<0>();
    }
    if constexpr (SCHED0) mma(I1{}, fill, I3{}, false);
    else { mma(I1{}, cur, I4{}, more1); if (more1) tile_done(); }
    fill = cur;
    cur = nslot;
  }
  stamp(3);
  __builtin_amdgcn_s_barrier();                    // every wave is done reading the ring: it becomes the staging area
  asm volatile("" ::: "memory");
  if constexpr (BN == 128 && DMODE == 0) {
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
        for (int q = 0; q < 4; ++q)
          slab[(size_t)((wave * NV + x * 4 + q) * 64 + lane)] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      volatile int* flag = reinterpret_cast<volatile int*>(smem + NS * STAGE - 16);   // inside the one LDS array, beyond the staging regions
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
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
    if constexpr (!XA_EARLY) xattn_load_frags(p, m0 + wm * WM, n0 + wn * WN, lane, xkf, xvf);
    xattn_inplace<TM>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros, xkf, xvf);
    IgemmParams pe = p;                       // bias and the LayerNorm affine went into q: the store adds nothing
    pe.bias = nullptr; pe.ln_stat = nullptr;
    igemm_epilogue_staged<TM, TN>(pe, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * (WM * WN * 4), lnA, lnC, zeros);
    return;
  }
  bool rows_done = false;
  if (!p.epi_staged) {       // the production kernel's direct row-per-lane epilogue
    if (p.act != 1 && igemm_rows_ok<TM, TN, false>(p, n0 + wn * WN)) {
      stamp(13);     // barrier passed, path chosen
      auto est = [&](int i) { if (i == 0) stamp(14); else if (i == 1) stamp(15); else if (i == 7) stamp(7 - 1); };   // words 14, 15, 6 (tile_id moves out)
      igemm_epilogue_rows<TM, TN, false>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, lnA, lnC, zeros, est);
      rows_done = true;
    }
  }
  constexpr bool FITS = NW * WM * WN * 4 <= NS * STAGE;        // full-width staging regions fit the dead ring
  if (rows_done) {
  } else if (FITS || p.act == 1) {
    const int region = p.act == 1 ? WM * (WN / 2) * 4 : WM * WN * 4;   // GEGLU halves the staged width
    if constexpr (BM == 256 && BN == 128 && NW == 8 && WGM == 4 && sizeof(T) == 2) {
      static_assert(NW * WM * WN * 4 + 4096 <= NS * STAGE, "GroupNorm-statistics scratch must fit behind the staging regions");
      const GnCtx gc{smem + NW * WM * WN * 4, wave, wm, wn, m0, n0};
      igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * region, lnA, lnC, zeros, &gc);
    } else
    igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * region, lnA, lnC, zeros);
  } else {
    igemm_epilogue<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, fr, fh, lnA, lnC);
  }
  if constexpr (TL) {
    stamp(12);                                           // epilogue instructions issued (stores in flight)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the epilogue's stores have left the wave
    stamp(4);
    if (lane == 0) { tl[5] = (unsigned)nk; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); tl[7] = xcc; }
    __builtin_amdgcn_s_waitcnt(0);
    if (g_tl_buf)
      for (int i = lane; i < kTlWords; i += 64) g_tl_buf[((size_t)blockIdx.x * NW + wave) * kTlWords + i] = tl[i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Warp-specialised variant: 8 compute waves + NL loader waves per workgroup.
//
// Measured on the pipelined kernel above (tools/igemm_ksweep.py): with the DMA pieces issued by the computing waves the
// k-loop runs at ~1000 TFLOP/s; the same loop with NO DMA issue runs at ~1300-1440, and pointing every piece at
// L2-resident data changes nothing -- the cost is the ISSUE of global_load_lds (~60+ cycles of the issuing wave per
// 1-KiB piece, right between its MFMAs), not latency or bandwidth.  So the pieces move to dedicated loader waves: they
// own the tap walk, the source pointers, the counted vmcnt waits and nothing else; the compute waves run ds_read + MFMA
// + one barrier per k-tile.  Protocol per k-tile kt (all waves meet at the same raw s_barrier):
//   loader : wait own pieces of tile kt+1 (vmcnt leaves tiles kt+2.. in flight) -> barrier -> issue tile kt+NS into the
//            slot of tile kt (free: every compute wave finished reading it before the barrier)
//   compute: kk-steps 0..2 of tile kt (fragments double buffered, counted lgkmcnt) -> lgkmcnt(0) -> barrier -> prefetch
//            the first fragments of tile kt+1 -> kk-step 3
template <int BM, int BN, int NS, int NL>
__global__ __launch_bounds__(512 + 64 * NL) void igemm_ws_kernel(const IgemmParams p, const void* zeros) {
  constexpr int WM = BM / 4, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NF = TM + TN;
  constexpr int APC = BM / 8, BPC = BN / 8;            // 8-row DMA pieces per k-tile
  constexpr int AJ = APC / NL, BJ = BPC / NL;          // per loader wave
  constexpr int PER = AJ + BJ;
  constexpr int KT = 64;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(NS >= 3 && APC % NL == 0 && BPC % NL == 0, "bad loader split");
  static_assert(PER * (NS - 1) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
  const int nk = p.Kpad / KT;

  if (wave >= 8) {
    // =============================================================== loader wave lw: pieces pc = j*NL + lw
    const int lw = wave - 8;
    const int lrow = lane >> 3, slot = lane & 7;
    const int HWo = p.Hout * p.Wout;
    const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
    int rb[AJ], ry[AJ], rx[AJ], rsw[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int row = (j * NL + lw) * 8 + lrow;
      const int m = m0 + row;
      rsw[j] = (slot ^ ((row >> 1) & 7)) * 8;
      if (m < p.M) {
        const int b = m / HWo;
        const int rem = m - b * HWo;
        const int oy = rem / p.Wout;
        rb[j] = b; ry[j] = oy * p.stride - p.pad; rx[j] = (rem - oy * p.Wout) * p.stride - p.pad;
      } else { rb[j] = -1; ry[j] = -(1 << 28); rx[j] = 0; }
    }
    const half_t* Ag = reinterpret_cast<const half_t*>(p.A);
    const half_t* wptr[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int row = (j * NL + lw) * 8 + lrow;
      wptr[j] = reinterpret_cast<const half_t*>(p.W) + (size_t)(n0 + row) * p.Kpad + (slot ^ ((row >> 1) & 7)) * 8;
    }
    const half_t* aptr[AJ];
    int aadv[AJ];
    int s_c0 = 0, s_dy = 0, s_dx = 0;
    auto retap = [&]() {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int iy = ry[j] + s_dy, ix = rx[j] + s_dx;
        const bool ok = (unsigned)iy < (unsigned)Hup && (unsigned)ix < (unsigned)Wup;
        const size_t off = (((size_t)(rb[j] < 0 ? 0 : rb[j]) * p.Hin + ((ok ? iy : 0) >> p.up)) * p.Win + ((ok ? ix : 0) >> p.up)) * p.lda + rsw[j];
        aptr[j] = ok ? Ag + off : reinterpret_cast<const half_t*>(zeros);
        aadv[j] = ok ? KT : 0;
      }
    };
    retap();
    auto issue_tile = [&](int buf) {
      char* la = smem + buf * STAGE + lw * 1024;
      char* lb = la + BM * 128;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(la + j * NL * 1024), 16, 0, 0);
        aptr[j] += aadv[j];
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(lb + j * NL * 1024), 16, 0, 0);
        wptr[j] += KT;
      }
      s_c0 += KT;
      if (s_c0 == p.Cin) {
        s_c0 = 0;
        if (++s_dx == p.ksize) { s_dx = 0; ++s_dy; }
        retap();
      }
    };
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < nk) issue_tile(s);
    if (NS <= nk) wait_vmcnt<PER * (NS - 1)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
      if (kt + NS - 1 < nk) wait_vmcnt<PER * (NS - 2)>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NS < nk) issue_tile(cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
    }
    __builtin_amdgcn_s_barrier();               // the compute waves' "ring is dead" barrier before the staged epilogue
    return;
  }

  // ================================================================= compute wave
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;
  unsigned basea, baseb;
  {
    const int ra = wm * WM + fr, rbw = wn * WN + fr;
    basea = lds0 + ra * 128 + ((fh ^ ((ra >> 1) & 7)) << 4);
    baseb = lds0 + BM * 128 + rbw * 128 + ((fh ^ ((rbw >> 1) & 7)) << 4);
  }
  half8 fA[2][TM], fB[2][TN];
  auto ldfrag = [&](unsigned so, int kk, auto SET) {
    constexpr int set = decltype(SET)::value;
    const unsigned aa = (basea ^ (kk << 5)) + so, ab = (baseb ^ (kk << 5)) + so;
    static_for<TM>([&](auto I) { fA[set][decltype(I)::value] = lds_read128<decltype(I)::value * 4096>(aa); });
    static_for<TN>([&](auto J) { fB[set][decltype(J)::value] = lds_read128<decltype(J)::value * 4096>(ab); });
  };
  auto mma = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[set][j], fA[set][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  float lnA[TM], lnC[TM];
  ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  __builtin_amdgcn_s_barrier();                 // tile 0 landed (loaders waited for their pieces)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  ldfrag(0, 0, I0{});
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned so = cur * STAGE;
    const int nslot = cur + 1 == NS ? 0 : cur + 1;
    ldfrag(so, 1, I1{});
    wait_lgkmcnt<NF>();
    mma(I0{});
    ldfrag(so, 2, I0{});
    wait_lgkmcnt<NF>();
    mma(I1{});
    ldfrag(so, 3, I1{});
    wait_lgkmcnt<NF>();
    mma(I0{});
    wait_lgkmcnt<0>();                          // own reads of tile kt complete
    if (kt + 1 < nk) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ldfrag(nslot * STAGE, 0, I0{});
    }
    mma(I1{});
    cur = nslot;
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  igemm_epilogue_staged<TM, TN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, smem + wave * (WM * WN * 4), lnA, lnC, zeros);
}

template <typename K> static void set_lds_attr_m(K kernel, size_t lds, bool (&done)[kIgemmMaxDev], int dev) {
  if (done[dev]) return;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    throw std::runtime_error("igemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  done[dev] = true;
}
template <int BM, int BN, int NS, int MINB = 2>
static void launch_glds_m(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  static bool attr_set[kIgemmMaxDev] = {};
  const int dev = igemm_current_device();
  set_lds_attr_m(&igemm_glds_m_kernel<BM, BN, NS, MINB>, lds, attr_set, dev);
  hipLaunchKernelGGL((igemm_glds_m_kernel<BM, BN, NS, MINB>), dim3(tilesM * tilesN), dim3(256), lds, s, p, igemm_zero_page());
}
template <int BM, int BN, int NS, bool PRIO, int DMODE = 0, int WGM = 4, int NW = 8, bool UNR = false, typename T = half_t, int PF = 0, bool XA = false, bool TL = false>
static void launch_pipe_m(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)pipe_lds_total(NS * (BM + BN) * 128, PF > 0 ? NW * 256 : 0) + (TL ? NW * kTlWords * 4 : 0);
  static_assert(!TL || pipe_lds_total(NS * (BM + BN) * 128, PF > 0 ? NW * 256 : 0) + NW * kTlWords * 4 <= 163840, "timeline stamps do not fit behind the ring");
  static bool attr_set[kIgemmMaxDev] = {};
  const int dev = igemm_current_device();
  set_lds_attr_m(&igemm_pipe_m_kernel<BM, BN, NS, PRIO, DMODE, WGM, NW, UNR, T, PF, XA, TL>, lds, attr_set, dev);
  IgemmParams q = p;
  q.splitk = 1;
  hipLaunchKernelGGL((igemm_pipe_m_kernel<BM, BN, NS, PRIO, DMODE, WGM, NW, UNR, T, PF, XA, TL>), dim3(tilesM * tilesN), dim3(64 * NW), lds, s, q, igemm_zero_page());
}
template <int BM, int BN, int NS, int NL>
static void launch_ws(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  static bool attr_set[kIgemmMaxDev] = {};
  const int dev = igemm_current_device();
  set_lds_attr_m(&igemm_ws_kernel<BM, BN, NS, NL>, lds, attr_set, dev);
  hipLaunchKernelGGL((igemm_ws_kernel<BM, BN, NS, NL>), dim3(tilesM * tilesN), dim3(512 + 64 * NL), lds, s, p, igemm_zero_page());
}

// timeline buffer of the TL variants: [workgroups][waves][kTlWords] unsigned (device memory owned by the caller; null = off)
void igemm_set_timeline(void* buf) {
  unsigned* b = reinterpret_cast<unsigned*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &b, sizeof(b)) != hipSuccess) throw std::runtime_error("igemm: cannot set the timeline buffer");
}
int igemm_timeline_words() { return kTlWords; }

// variant ids of the measurement build (tools/igemm_sweep.py, tools/dma_modes.py, the tests' measure-only parametrisations)
bool launch_igemm_measure(const IgemmParams& psk, int variant, hipStream_t s) {
  const IgemmParams& p = psk;
  switch (variant) {
    case 40: launch_pipe_m<256, 128, 3, false, 0, 4, 8, true, half_t, 4>(psk, s); break;   // + L2 prefetch touches 4 / 8 k-tiles ahead:
    case 41: launch_pipe_m<128, 128, 4, false, 0, 4, 8, true, half_t, 4>(psk, s); break;   //   measured SLOWER (profiles/r02_l2_prefetch_ab.txt)
    case 42: launch_pipe_m<128, 128, 4, false, 0, 4, 8, true, half_t, 8>(psk, s); break;
    case 43: launch_pipe_m<256, 128, 3, false, 0, 4, 8, true, half_t, 8>(psk, s); break;
    // timeline (s_memtime-stamped) twins of the production 96x128 / 128x128 / 256x128 kernels: variants 145 / 136 / 135
    case 145: launch_pipe_m<96, 128, 5, false, 0, 3, 6, true, half_t, 0, false, true>(psk, s); break;
    case 136: launch_pipe_m<128, 128, 4, false, 0, 4, 8, true, half_t, 0, false, true>(psk, s); break;
    case 135: launch_pipe_m<256, 128, 3, false, 0, 4, 8, true, half_t, 0, false, true>(psk, s); break;
    case 1: launch_glds_m<128, 128, 3>(psk, s); break;
    case 2: launch_glds_m<128, 64, 4>(psk, s); break;
    case 3: launch_glds_m<64, 128, 4>(psk, s); break;
    case 5: launch_glds_m<128, 64, 2>(psk, s); break;
    case 7: launch_glds_m<128, 128, 4>(psk, s); break;
    case 8: launch_glds_m<64, 128, 3>(psk, s); break;
    case 33: launch_glds_m<256, 128, 3, 1>(psk, s); break;
    case 34: launch_pipe_m<256, 128, 3, true, 0, 2, 4>(psk, s); break;   // the hand-ordered loop on 4 waves x (128x64)
    case 37: launch_pipe_m<256, 128, 3, true, 0, 4, 8, true>(psk, s); break;    // unrolled ring with s_setprio
    case 10: launch_pipe_m<256, 128, 3, false>(psk, s); break;   // rolled 8-wave pipelined kernels
    case 11: launch_pipe_m<256, 128, 3, true>(psk, s); break;
    case 12: launch_pipe_m<128, 128, 4, false>(psk, s); break;
    case 13: launch_pipe_m<128, 128, 4, true>(psk, s); break;
    case 14: launch_pipe_m<128, 128, 3, true>(psk, s); break;
    case 15: launch_pipe_m<256, 128, 3, true, 1>(psk, s); break;    // early DMA issue (after the barrier)
    case 16: launch_pipe_m<128, 128, 4, true, 1>(psk, s); break;
    case 17: launch_pipe_m<256, 128, 3, true, 2>(psk, s); break;    // measurement only: no k advance (WRONG results)
    case 18: launch_pipe_m<256, 128, 3, true, 3>(psk, s); break;    // measurement only: no DMA in the loop (WRONG results)
    case 24: launch_pipe_m<256, 128, 3, true, 4>(psk, s); break;    // lookahead-2 fragment prefetch
    case 27: launch_pipe_m<256, 128, 3, true, 5>(psk, s); break;    // measurement only: DMA-only / contiguous-source modes
    case 28: launch_pipe_m<256, 128, 3, true, 6>(psk, s); break;
    case 29: launch_pipe_m<256, 128, 3, true, 7>(psk, s); break;
    case 30: launch_pipe_m<128, 128, 4, true, 5>(psk, s); break;
    case 31: launch_pipe_m<128, 128, 4, true, 6>(psk, s); break;
    case 32: launch_pipe_m<128, 128, 4, true, 7>(psk, s); break;
    case 25: launch_pipe_m<128, 128, 4, true, 4>(psk, s); break;
    case 19:                                                    // 256x160, 8x1 waves, rolled
      if (p.N % 160 != 0) return false;
      launch_pipe_m<256, 160, 3, true, 0, 8>(psk, s); break;
    case 20: launch_ws<256, 128, 3, 2>(psk, s); break;            // 8 compute + 2 loader waves
    case 21: launch_ws<256, 128, 3, 4>(psk, s); break;            // 8 compute + 4 loader waves
    case 22: launch_ws<128, 128, 4, 2>(psk, s); break;
    case 23: launch_ws<128, 128, 4, 4>(psk, s); break;
    default: return false;
  }
  return true;
}

}  // namespace sdxl
#endif  // SDXL_MEASURE
