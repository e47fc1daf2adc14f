// Fast-path implicit GEMM for gfx950: f16 operands, direct-to-LDS staging through an NS-deep ring.
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
//   * XCD-aware block -> tile mapping; epilogue staged through LDS so loads/stores are whole row segments; folded LayerNorm.
// Kernels, in file order (selection: launch_igemm_glds at the bottom; measurements: DESIGN.md section 4.1):
//   igemm_glds_kernel  4 waves, 2..4-slot ring, several co-resident blocks per CU       (ragged multi-round grids)
//   igemm_pipe_kernel  8 waves, hand-ordered k-loop: counted lgkmcnt/vmcnt, register-double-buffered fragments, DMA pieces
//                      between the MFMAs; tiles 256x128, 128x128, 256x160 (GEGLU)        (everything else -- the default)
//   igemm_wide_kernel  256x320 tile, k-tile 32                                           (experiment, not selected)
//   igemm_ws_kernel    8 compute + 2/4 DMA-loader waves                                  (experiment, not selected)
#include "kernels.h"
#include <stdexcept>
#include <type_traits>

namespace sdxl {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// exact-erf GELU (burn nn::Gelu, unet/mod.rs:954) for the f16 fast path: 1 + erf(x/sqrt2) through the complementary form
// E = erfc(|z|) = poly(t) * exp(-z^2), t = 1/(1 + 0.3275911 |z|)  (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7, no
// cancellation for negative x); ~12 VALU instead of ocml erff's branchy ~40.  The strict fp32 kernel keeps erff.
__device__ __forceinline__ float gelu_erf2(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = poly * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  return 0.5f * x * (x >= 0.f ? 2.0f - e : e);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// LayerNorm-folded GEMM: per output row m the epilogue needs a = rstd and c = -rstd*mu from the producer's row statistics
// ln_stat[slot][m] = (mean, M2) of columns [64 slot, 64 slot + 64) of row m (equal counts).  CANONICAL summation order -- every
// path below produces the same bits, whatever tile / thread mapping evaluates it (a batch entry must not depend on the tile the
// batch size selects): four class sums over the slots k = j (mod 4), k ascending, combined as (P0 + P1) + (P2 + P3); first the
// means -> mu, then M2 and (mean_k - mu)^2 the same way:  var = (sum M2 + 64 sum (mean_k - mu)^2) / K  (Chan merge, biased,
// eps inside the sqrt: layernorm/mod.rs:42-49).
typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ln_row_coef(const IgemmParams& p, int m, float& a, float& c) {
  a = 1.f; c = 0.f;
  if (p.ln_stat && m < p.M) {
    const ln_f32x2* st = reinterpret_cast<const ln_f32x2*>(p.ln_stat) + m;
    const size_t M = (size_t)p.M;
    float P[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.ln_slots; ++k) P[k & 3] += st[(size_t)k * M][0];
    const float mu = ((P[0] + P[1]) + (P[2] + P[3])) / (float)p.ln_slots;
    float Q[4] = {0.f, 0.f, 0.f, 0.f}, D[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.ln_slots; ++k) { const ln_f32x2 v = st[(size_t)k * M]; const float d = v[0] - mu; Q[k & 3] += v[1]; D[k & 3] = fmaf(d, d, D[k & 3]); }
    const float s2 = (Q[0] + Q[1]) + (Q[2] + Q[3]), sd = (D[0] + D[1]) + (D[2] + D[3]);
    const float var = (s2 + 64.f * sd) * p.ln_invc;
    a = 1.0f / sqrtf(var + (p.ln_eps_ptr ? *p.ln_eps_ptr : p.ln_eps));
    c = -a * mu;
  }
}
// per-lane form (kernels without LDS room for the cooperative one, K > 1536): every lane evaluates its own TM rows
template <int TM>
__device__ __forceinline__ void ln_prologue(const IgemmParams& p, int mw, int fr, float (&lnA)[TM], float (&lnC)[TM]) {
#pragma unroll
  for (int i = 0; i < TM; ++i) ln_row_coef(p, mw + i * 32 + fr, lnA[i], lnC[i]);
}
// Cooperative form: the workgroup's NT threads evaluate the BM rows of the tile ONCE (TPR = NT / BM = 2 or 4 adjacent lanes per
// row, each taking 4 / TPR of the slot classes) and park (a, c) in LDS; the waves pick their rows up behind the prologue's
// barrier.  The per-lane form has every wave of a row group AND both lane halves load the same 20 x 8 bytes per row: 164 KB of
// L2 requests per workgroup for 41 KB of statistics -- +17 % on the path that bounds these GEMMs (QKV projection +4.9 us of 31;
// tools/igemm_epilogue_cost.py).  load() goes BEFORE the first DMA piece (oldest entries of the vmcnt queue), finish() after the
// pieces are issued.  Up to 24 slots (K <= 1536).
template <int BM, int NT>
struct LnCoop {
  static constexpr int TPR = NT / BM;
  static constexpr bool OK = NT % BM == 0 && (TPR == 2 || TPR == 4);
  static constexpr int NC = OK ? 4 / TPR : 1, PERC = 6;
  ln_f32x2 v[NC][PERC];
  int row, sub;
  bool live;
  __device__ __forceinline__ void load(const IgemmParams& p, int m0, int tid) {
    row = tid / TPR; sub = tid - row * TPR;
    const int m = m0 + row;
    live = p.ln_stat != nullptr && p.ln_slots <= 24;
    const ln_f32x2* st = reinterpret_cast<const ln_f32x2*>(p.ln_stat) + (m < p.M ? m : 0);
    const size_t M = (size_t)p.M;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < PERC; ++q) {
        const int k = sub + c * TPR + 4 * q;
        v[c][q] = (live && k < p.ln_slots) ? st[(size_t)k * M] : ln_f32x2{0.f, 0.f};
      }
  }
  __device__ __forceinline__ static float combine(const float (&P)[NC]) {   // (P0 + P1) + (P2 + P3), the four classes spread over TPR lanes
    if constexpr (TPR == 4) { const float t = P[0] + __shfl_xor(P[0], 1); return t + __shfl_xor(t, 2); }
    else { const float u = P[0] + __shfl_xor(P[0], 1), w = P[1] + __shfl_xor(P[1], 1); return u + w; }
  }
  __device__ __forceinline__ void finish(const IgemmParams& p, int m0, float* coef) {
    if (!live) return;
    float P[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      P[c] = 0.f;
#pragma unroll
      for (int q = 0; q < PERC; ++q) if (sub + c * TPR + 4 * q < p.ln_slots) P[c] += v[c][q][0];
    }
    const float mu = combine(P) / (float)p.ln_slots;
    float Q[NC], D[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      Q[c] = 0.f; D[c] = 0.f;
#pragma unroll
      for (int q = 0; q < PERC; ++q)
        if (sub + c * TPR + 4 * q < p.ln_slots) { const float d = v[c][q][0] - mu; Q[c] += v[c][q][1]; D[c] = fmaf(d, d, D[c]); }
    }
    const float s2 = combine(Q), sd = combine(D);
    const float var = (s2 + 64.f * sd) * p.ln_invc;
    const float a = 1.0f / sqrtf(var + (p.ln_eps_ptr ? *p.ln_eps_ptr : p.ln_eps));
    if (sub == 0) {
      const bool ok = m0 + row < p.M;
      coef[row * 2] = ok ? a : 1.f;
      coef[row * 2 + 1] = ok ? -a * mu : 0.f;
    }
  }
};

template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int fr, int fh,
                                               const float (&lnA)[TM], const float (&lnC)[TM]) {
  const bool geglu = p.act == 1;
  const int nlim = geglu ? (p.N >> 1) : p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mw + i * 32 + fr;
    if (m >= p.M) continue;
    const int bidx = m / p.rpb;
    const int key = m - bidx * p.rpb;
    const float lna = lnA[i], lnc = lnC[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nt = nw + j * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (geglu && q >= 2) continue;            // gate groups are consumed with their x group
        const int nb = nt + 8 * q + 4 * fh;       // packed column of element r = 0
        if (nb >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q * 4 + r];
        if (p.ln_stat) {
          const f32x4 cz = *reinterpret_cast<const f32x4*>(p.ln_cs + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = lna * v[r] + lnc * cz[r];
        }
        if (p.bias) {
          const f32x4 bz = *reinterpret_cast<const f32x4*>(p.bias + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bz[r];
        }
        if (p.ebias) {
          const f32x4 ez = *reinterpret_cast<const f32x4*>(p.ebias + (size_t)bidx * p.ebias_ld + nb);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += ez[r];
        }
        int nout = nb;
        if (geglu) {
          f32x4 gz = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) gz = *reinterpret_cast<const f32x4*>(p.bias + nb + 16);
          f32x4 gc = {0.f, 0.f, 0.f, 0.f};
          if (p.ln_stat) gc = *reinterpret_cast<const f32x4*>(p.ln_cs + nb + 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gelu_erf2(lna * acc[i][j][(q + 2) * 4 + r] + lnc * gc[r] + gz[r]);
          nout = (nt >> 1) + 8 * q + 4 * fh;
        }
        if (geglu || nb < p.n_split) {
          const bool vec = nout + 3 < nlim && (nout & 3) == 0;
          if (p.R) {
            if (vec && p.r_dt == DT_F16 && (p.ldr & 3) == 0) {
              const half4 rr = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(p.R) + (size_t)m * p.ldr + nout);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            } else if (vec && p.r_dt == DT_F32 && (p.ldr & 3) == 0) {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + nout);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += rr[r];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (nout + r < nlim)
                  v[r] += p.r_dt == DT_F16 ? (float)reinterpret_cast<const half_t*>(p.R)[(size_t)m * p.ldr + nout + r]
                                           : reinterpret_cast<const float*>(p.R)[(size_t)m * p.ldr + nout + r];
            }
          }
          if (vec && p.c_dt == DT_F16 && (p.ldc & 3) == 0) {
            half4 h; h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
            *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.C) + (size_t)m * p.ldc + nout) = h;
          } else if (vec && p.c_dt == DT_F32 && (p.ldc & 3) == 0) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + nout) = o;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (nout + r < nlim) {
                if (p.c_dt == DT_F16) reinterpret_cast<half_t*>(p.C)[(size_t)m * p.ldc + nout + r] = (half_t)v[r];
                else reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + nout + r] = v[r];
              }
          }
        } else {
          // transposed store Ct[b][n - n_split][key]: lanes 0..31 hold 32 consecutive keys of each row
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < p.N) {
              const size_t o = ((size_t)bidx * p.ct_rows + (nb + r - p.n_split)) * p.ct_ld + key;
              if (p.c_dt == DT_F16) reinterpret_cast<half_t*>(p.Ct)[o] = (half_t)v[r];
              else reinterpret_cast<float*>(p.Ct)[o] = v[r];
            }
        }
      }
    }
  }
}

// ---- LDS-staged epilogue (the fast one).  The direct epilogue above stores 8 bytes per lane with 32 different rows per
// instruction: every 128-byte output line is written by eight separate 16-byte partial requests, and the L2 request rate
// caps the whole GEMM at ~1.4 TB/s of output (measured: a K=64 GEMM with a 42 MB output takes 33 us).  Here each wave
// first parks its (bias / time-embedding / GEGLU applied) fp32 tile in its own LDS region -- [row][col] with 16-byte
// chunks XOR-swizzled by row&7, transposed for the V^T part -- then re-reads it row-contiguously: 8 (or 4) lanes cover
// one output row segment, add the residual with 16-byte loads and store whole 128-byte (64-byte) line segments.
// `lds` = this wave's private region of WM*WN*4 bytes (the k-loop ring, dead by now; callers barrier first).
// dynamic LDS of a pipelined kernel: ring (+ prefetch scratch) + 2 KiB for the cooperative LayerNorm coefficients when the CU's
// 160 KiB leave the room (the 5-slot 128x128 ring does not: it keeps the per-lane form)
__host__ __device__ constexpr int pipe_lds_total(int ring, int extra) { return ring + extra + 2048 <= 163840 ? ring + extra + 2048 : ring + extra; }

// block context of the GroupNorm-statistics epilogue (IgemmParams::gn_part): 4 KiB of LDS scratch past the staging regions,
// the wave's place in the 4 x 2 wave grid and the tile origin
struct GnCtx { char* scratch; int wave, wm, wn, m0, n0; };
template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ void igemm_epilogue_staged_impl(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw,
                                                           int lane, char* lds, bool transposed, const float (&lnA)[TM],
                                                           const float (&lnC)[TM], const void* zeros, const GnCtx* gc = nullptr) {
  constexpr int WM = TM * 32, WN = TN * 32;
  constexpr int ROWS = WM;                         // staged rows: m (normal) -- for the transposed part rows = n, cols = m
  constexpr int COLS = GEGLU ? WN / 2 : WN;
  // chunk swizzle (16-byte chunk index ^ row&7) needs whole groups of 8 chunks per staged row; odd widths go unswizzled
  constexpr int SWN = (COLS % 32) == 0 ? 7 : 0;    // normal image: COLS/4 chunks per row
  constexpr int SWT = (WM % 32) == 0 ? 7 : 0;      // transposed image: WM/4 chunks per row
  const int fr = lane & 31, fh = lane >> 5;
  // ---------------- stage 1: registers -> LDS (fp32)
  // Every per-column vector (bias, gate bias, folded-LayerNorm column sums, time-embedding bias) is fetched through a
  // pointer SELECT (a 16-byte zero page stands in for "absent"), never inside a branch: hipcc then issues the loads of a
  // whole 32-column group back to back and waits once.  With `if (p.bias) v += *ptr` each of the 8..32 loads became its own
  // load -> s_waitcnt vmcnt(0) -> use chain, i.e. 8..32 serial L2 round trips in every GEMM's epilogue.
  const f32x4* zv = reinterpret_cast<const f32x4*>(zeros);
  constexpr int NQ = GEGLU ? 2 : 4;
  if (!transposed) {
    constexpr int RB = COLS * 4;                   // bytes per staged row
    int bidx[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int m = mw + i * 32 + fr; bidx[i] = (p.ebias && m < p.M) ? m / p.rpb : 0; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nt = nw + j * 32;
      f32x4 bz[NQ], cz[NQ], gz[NQ], gc[NQ], ez[TM][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int nb = nt + 8 * q + 4 * fh;        // packed column of element r = 0 (bias arrays are padded to Npad)
        const bool ok = nb < p.N;                  // columns of the zero-padded weight rows: nothing to add, never stored
        bz[q] = *((p.bias && ok) ? reinterpret_cast<const f32x4*>(p.bias + nb) : zv);
        cz[q] = *((p.ln_stat && ok) ? reinterpret_cast<const f32x4*>(p.ln_cs + nb) : zv);
        if constexpr (GEGLU) {
          gz[q] = *((p.bias && ok) ? reinterpret_cast<const f32x4*>(p.bias + nb + 16) : zv);
          gc[q] = *((p.ln_stat && ok) ? reinterpret_cast<const f32x4*>(p.ln_cs + nb + 16) : zv);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
          ez[i][q] = *((p.ebias && ok) ? reinterpret_cast<const f32x4*>(p.ebias + (size_t)bidx[i] * p.ebias_ld + nb) : zv);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + fr;
        const float lna = lnA[i], lnc = lnC[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q * 4 + r];
          v = lna * v + lnc * cz[q] + bz[q] + ez[i][q];          // lna = 1, lnc = 0 without a folded LayerNorm
          int col = j * 32 + 8 * q + 4 * fh;
          if constexpr (GEGLU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= gelu_erf2(lna * acc[i][j][(q + 2) * 4 + r] + lnc * gc[q][r] + gz[q][r]);
            col = j * 16 + 8 * q + 4 * fh;
          }
          *reinterpret_cast<f32x4*>(lds + row * RB + ((((col >> 2) ^ (row & SWN))) << 4)) = v;
        }
      }
    }
  } else {
    constexpr int RB = WM * 4;                     // transposed image: row = n (WN rows), col = m (WM columns)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x4 bz[4], cz[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nb = nw + j * 32 + 8 * q + 4 * fh;
        const bool ok = nb < p.N;
        bz[q] = *((p.bias && ok) ? reinterpret_cast<const f32x4*>(p.bias + nb) : zv);
        cz[q] = *((p.ln_stat && ok) ? reinterpret_cast<const f32x4*>(p.ln_cs + nb) : zv);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mcol = i * 32 + fr;
        const float lna = lnA[i], lnc = lnC[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q * 4 + r];
          v = lna * v + lnc * cz[q] + bz[q];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nrow = j * 32 + 8 * q + 4 * fh + r;
            *reinterpret_cast<float*>(lds + nrow * RB + (((mcol >> 2) ^ (nrow & SWT)) << 4) + (mcol & 3) * 4) = v[r];
          }
        }
      }
    }
  }
  // the region is private to the wave: LDS operations of one wave complete in order, the compiler inserts the lgkmcnt wait
  // ---------------- stage 2: LDS -> (residual) -> global, row-contiguous
  if (!transposed) {
    constexpr int RB = COLS * 4;
    constexpr int LPR = COLS / 8;                  // lanes per row (8 values each)
    constexpr int ITEMS = ROWS * LPR;              // (row, 8-value piece) items, 64 per wave instruction
    const int nlim = GEGLU ? (p.N >> 1) : (p.n_split < p.N ? p.n_split : p.N);
    const int nwo = GEGLU ? (nw >> 1) : nw;
    // residual rows (f16, whole 16-byte pieces -- the case of every UNet / VAE residual): all of this lane's pieces are
    // requested up front through a pointer select, one wait for the lot instead of a load -> wait -> add -> store chain
    // per piece; anything else (fp32 residual stream, ragged or unaligned pieces) takes the per-piece path below
    constexpr int NIT = (ITEMS + 63) / 64;
    half8 rpre[NIT];
    bool rfast[NIT];
    // GroupNorm statistics of the stored tile (gn_part): a lane keeps the same 8 columns over the NIT row groups, so the column
    // sums over the wave's 64 rows are 8 per-lane accumulators + one 3-step xor reduction; shifted by the tile's first row
    constexpr bool GNP = TM == 2 && TN == 2 && !GEGLU;
    const bool gnp = GNP && gc != nullptr && p.gn_part != nullptr;
    float gpiv[8], gs1[8], gs2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gpiv[e] = 0.f; gs1[e] = 0.f; gs2[e] = 0.f; }
    float st_piv[NIT], st_s1[NIT], st_s2[NIT];    // row statistics (stat_out): per-lane partials of every row group
    size_t st_off[NIT];
    bool st_ok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { st_piv[it] = 0.f; st_s1[it] = 0.f; st_s2[it] = 0.f; st_off[it] = 0; st_ok[it] = false; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / LPR, piece = idx - row * LPR;
      const int n0 = nwo + piece * 8;
      const int m = mw + row;
      const half_t* rp = reinterpret_cast<const half_t*>(p.R) + (size_t)m * p.ldr + n0;
      rfast[it] = p.R && p.r_dt == DT_F16 && idx < ITEMS && m < p.M && n0 + 8 <= nlim && (reinterpret_cast<uintptr_t>(rp) & 15) == 0;
      rpre[it] = *(rfast[it] ? reinterpret_cast<const half8*>(rp) : reinterpret_cast<const half8*>(zeros));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / LPR, piece = idx - row * LPR;
      if (ITEMS % 64 != 0 && idx >= ITEMS) continue;
      const int n0 = nwo + piece * 8;
      const int m = mw + row;
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece) ^ (row & SWN)) << 4));
      const f32x4 b = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece + 1) ^ (row & SWN)) << 4));
      const bool valid = m < p.M && n0 < nlim;
      float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // the stored (rounded) values: what the consumer will read
      if (valid) {
      float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      const bool full = n0 + 8 <= nlim;
      if (rfast[it]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rpre[it][e];
      } else if (p.R) {
        if (p.r_dt == DT_F16) {
          const half_t* rp = reinterpret_cast<const half_t*>(p.R) + (size_t)m * p.ldr + n0;
          if (full && (reinterpret_cast<uintptr_t>(rp) & 15) == 0) {
            const half8 rr = *reinterpret_cast<const half8*>(rp);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n0 + e < nlim) v[e] += (float)rp[e];
          }
        } else {
          const float* rp = reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n0;
          if (full && (reinterpret_cast<uintptr_t>(rp) & 15) == 0) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n0 + e < nlim) v[e] += rp[e];
          }
        }
      }
      if (p.stat_out || gnp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rr[e] = p.c_dt == DT_F16 ? (float)(half_t)v[e] : v[e];
      }
      if (p.c_dt == DT_F16) {
        half_t* cp = reinterpret_cast<half_t*>(p.C) + (size_t)m * p.ldc + n0;
        if (full && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) {
          half8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
          *reinterpret_cast<half8*>(cp) = h;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (n0 + e < nlim) cp[e] = (half_t)v[e];
        }
      } else {
        float* cp = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0;
        if (full && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) {
          *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (n0 + e < nlim) cp[e] = v[e];
        }
      }
      }   // valid
      if constexpr (GNP) {
        if (gnp) {   // whole tiles only (M % 256 == 0, N % 64 == 0): every item is valid
          if (it == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gpiv[e] = __shfl(rr[e], lane & 7);     // row 0 of the wave tile, this lane's columns
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = rr[e] - gpiv[e]; gs1[e] += d; gs2[e] = fmaf(d, d, gs2[e]); }
        }
      }
      if constexpr (COLS % 64 == 0) {
        if (p.stat_out) {
          // 8 consecutive lanes hold one 64-column slot of a row.  Shifted sums around a pivot inside the data (the slot's
          // first value) -> (mean, M2) of the slot, never sum x^2 - (sum x)^2: rows with |mean| >> sigma (outlier channels of
          // the residual stream) keep their variance.  The consumer Chan-merges the K/64 slots (ln_prologue).
          // (only the per-lane partial sums here; the 8-lane reductions of ALL row groups run together behind the loop --
          // done per group they were NIT chains of three dependent cross-lane round trips, ~1.5 us per GEMM)
          const float piv = __shfl(rr[0], lane & ~7);
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = rr[e] - piv; s1 += d; s2 = fmaf(d, d, s2); }
          st_piv[it] = piv; st_s1[it] = s1; st_s2[it] = s2;
          st_ok[it] = (piece & 7) == 0 && valid;
          st_off[it] = ((size_t)(n0 >> 6) * p.M + m) * 2;
        }
      }
    }
    if constexpr (COLS % 64 == 0) {
      if (p.stat_out) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) { st_s1[it] += __shfl_xor(st_s1[it], o); st_s2[it] += __shfl_xor(st_s2[it], o); }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (st_ok[it]) {
            float* dst = p.stat_out + st_off[it];
            dst[0] = st_piv[it] + st_s1[it] * (1.0f / 64.0f);
            dst[1] = fmaxf(st_s2[it] - st_s1[it] * st_s1[it] * (1.0f / 64.0f), 0.f);
          }
      }
    }
    if constexpr (GNP) {
      if (gnp) {
        // lanes with the same piece (lane & 7) hold the same 8 columns: sum over the 8 row sub-lanes -> 64-row column sums
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 8; o < 64; o <<= 1) { gs1[e] += __shfl_xor(gs1[e], o); gs2[e] += __shfl_xor(gs2[e], o); }
        }
        float* sc = reinterpret_cast<float*>(gc->scratch);
        if (lane < 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dm = gs1[e] * (1.0f / 64.0f);
            sc[(gc->wave * 64 + lane * 8 + e) * 2] = gpiv[e] + dm;                      // mean of the wave's 64 rows
            sc[(gc->wave * 64 + lane * 8 + e) * 2 + 1] = fmaxf(gs2[e] - gs1[e] * dm, 0.f);   // M2
          }
        }
        __syncthreads();
        // the 4 row-waves of a column half merge (equal counts, Chan): wave (wm = 0, wn) writes the tile's 256-row statistics
        if (gc->wm == 0) {
          const int col = gc->n0 + gc->wn * 64 + lane;
          float mk[4], qk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { mk[k] = sc[((k * 2 + gc->wn) * 64 + lane) * 2]; qk[k] = sc[((k * 2 + gc->wn) * 64 + lane) * 2 + 1]; }
          const float mu = ((mk[0] + mk[1]) + (mk[2] + mk[3])) * 0.25f;
          float m2 = (qk[0] + qk[1]) + (qk[2] + qk[3]), sd = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float d = mk[k] - mu; sd = fmaf(d, d, sd); }
          m2 += 64.0f * sd;
          if (col < p.N) {
            float* dst = p.gn_part + ((size_t)(gc->m0 >> 8) * p.N + col) * 2;
            dst[0] = mu; dst[1] = m2;
          }
        }
      }
    }
  } else {
    // rows = n (Ct row n - n_split of batch b), 8 consecutive m = 8 consecutive keys when they sit in one batch entry
    constexpr int RB = WM * 4;
    constexpr int LPR = WM / 8;
    constexpr int ITEMS = WN * LPR;
#pragma unroll
    for (int it = 0; it < (ITEMS + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / LPR, piece = idx - row * LPR;
      if (ITEMS % 64 != 0 && idx >= ITEMS) continue;
      const int mbase = mw + piece * 8;
      const int b0 = mbase / p.rpb;
      const int key0 = mbase - b0 * p.rpb;
      const int n = nw + row;
      const f32x4 a = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece) ^ (row & SWT)) << 4));
      const f32x4 b = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece + 1) ^ (row & SWT)) << 4));
      if (n >= p.N || mbase >= p.M) continue;
      const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      const size_t o = ((size_t)b0 * p.ct_rows + (n - p.n_split)) * p.ct_ld + key0;
      const bool full = mbase + 8 <= p.M && key0 + 8 <= p.rpb;
      if (p.c_dt == DT_F16) {
        half_t* cp = reinterpret_cast<half_t*>(p.Ct) + o;
        if (full && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) {
          half8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
          *reinterpret_cast<half8*>(cp) = h;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int m = mbase + e;
            if (m < p.M) {
              const int bb = m / p.rpb;
              reinterpret_cast<half_t*>(p.Ct)[((size_t)bb * p.ct_rows + (n - p.n_split)) * p.ct_ld + (m - bb * p.rpb)] = (half_t)v[e];
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = mbase + e;
          if (m < p.M) {
            const int bb = m / p.rpb;
            reinterpret_cast<float*>(p.Ct)[((size_t)bb * p.ct_rows + (n - p.n_split)) * p.ct_ld + (m - bb * p.rpb)] = v[e];
          }
        }
      }
    }
  }
}

// dispatch: the staged path needs the wave's column range on one side of n_split; anything else takes the direct epilogue
template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue_staged(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw,
                                                      int lane, char* lds, const float (&lnA)[TM], const float (&lnC)[TM],
                                                      const void* zeros, const GnCtx* gc = nullptr) {
  constexpr int WN = TN * 32;
  if (p.act == 1) { igemm_epilogue_staged_impl<TM, TN, true>(p, acc, mw, nw, lane, lds, false, lnA, lnC, zeros); return; }
  const bool all_normal = nw + WN <= p.n_split || p.n_split >= p.N;
  const bool all_transposed = nw >= p.n_split;
  if (all_normal) igemm_epilogue_staged_impl<TM, TN, false>(p, acc, mw, nw, lane, lds, false, lnA, lnC, zeros, gc);
  else if (all_transposed) igemm_epilogue_staged_impl<TM, TN, false>(p, acc, mw, nw, lane, lds, true, lnA, lnC, zeros);
  else igemm_epilogue<TM, TN>(p, acc, mw, nw, lane & 31, lane >> 5, lnA, lnC);
}

template <int BM, int BN, int NS, int MINB = 2>
__global__ __launch_bounds__(256, MINB) void igemm_glds_kernel(const IgemmParams p, const void* zeros) {   // >= MINB blocks per CU
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
// 8-wave software-pipelined variant (one 512-thread workgroup per CU, two waves per SIMD).
//
// Why a second structure: in the 4-wave kernel above every kk-step is {4 ds_read_b128 -> lgkmcnt(0) -> 4 MFMA} on ONE
// fragment register set and every k-tile starts with vmcnt(0), so LDS latency and DMA latency are both exposed and the
// matrix pipe idles ~2/3 of the time (measured 25..35 % MFMA utilisation).  Here the k-loop is hand ordered:
//   * fragments are double buffered in registers: the ds_reads of step kk+1 are issued before the MFMAs of step kk and
//     waited for with a COUNTED lgkmcnt (hipcc only emits lgkmcnt(0) across the loop back edge, so the reads are inline
//     asm and every wait is followed by sched_barrier(0) so no MFMA is hoisted above it);
//   * the ring is NS >= 3 deep and the DMA wait is counted too: at the single barrier of k-tile kt (between its third and
//     fourth kk-step) a wave waits only for its own pieces of tile kt+1; the pieces of tile kt+NS-1 -- issued two per
//     kk-step BETWEEN the MFMAs of the first three steps, into the slot the previous barrier freed -- stay in flight
//     across the raw s_barrier.  After the barrier the first fragments of tile kt+1 are prefetched under the fourth step;
//   * 256x128 block tile (wave tile 64x64, 4x2 waves): 48 KiB of DMA per 16 MFMA per wave -> 47 B/clk/CU of the 64 B/clk
//     vector-memory path at full MFMA rate (the 128x128 tile needs all 64); 128x128 (wave tile 32x64) for small grids;
//   * optional s_setprio(1) around the MFMA clusters so the partner wave's DMA / ds_read issue yields to MFMA issue.
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int OFF, typename F = half8> __device__ __forceinline__ F lds_read128(unsigned addr) {
  F v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// one 32x32 MFMA tile step over the 16-byte fragments of a kk-step.  f16: 8 halfs per lane = one v_mfma_f32_32x32x16_f16.
// f32 (strict mode): 4 floats per lane = four v_mfma_f32_32x32x2_f32, MFMA e taking element e of every lane -- lanes 0..31
// hold k-chunk 2kk, lanes 32..63 chunk 2kk+1, so MFMA e contracts k = 8kk + e and 8kk + 4 + e: a permutation of the k order
// that A and B share (bit-for-bit an fp32 fma chain per output, at the 157 TFLOP/s f32 MFMA rate).
template <typename T> struct PipeElem;
template <> struct PipeElem<half_t> {
  typedef half8 frag;
  static __device__ __forceinline__ f32x16 mma(const half8& w, const half8& a, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0);
  }
};
template <> struct PipeElem<float> {
  typedef f32x4 frag;
  static __device__ __forceinline__ f32x16 mma(const f32x4& w, const f32x4& a, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], a[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], a[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], a[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], a[3], c, 0, 0, 0);
    return c;
  }
};
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
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

// ---- cross-attention fused into the query projection (transformer attn2: unet/mod.rs:731-763 + attention at 765-795) ----
// The context K / V^T of a trajectory are constant (projected once per prompt, UNet::set_context) and short (77 keys), and a
// wave tile of the 128- / 256-row kernels is 32 (64) queries x 64 columns = exactly ONE head.  So the wave that holds the
// finished q tile in its accumulators runs the whole attention on it, in registers, before the store:
//   S^T[key][query] = K_h q^T      A = K fragments read straight from global (12 KiB per head, pre-packed in operand order), B = q (f16)
//   P = softmax over the 77 keys   a lane owns ONE query (column lane&31) and 16 keys of each 32-key tile: max / sum are
//                                  in-lane plus one xor-32 exchange
//   O^T[d][query] = V_h^T P^T      A = V^T fragments from global, B = P (f16) -- the S accumulators re-used as operands
// The contraction index of an MFMA is free to permute as long as A and B agree, so the accumulator registers 8qq..8qq+7 of
// column tile j ARE the B fragment of "k-step (j, qq)": element e <-> d = 32j + 16qq + 8(e>>2) + 4(lane>>5) + (e&3); the K
// fragments are packed with the same map (xattn_pack_kernel, once per prompt), and likewise keys for P / V^T.  O^T comes out in the layout
// the q tile came in, so the normal staged store follows unchanged.  No LDS, no cross-wave traffic, one launch less per block.
// K / V^T fragments in MFMA operand order (launch_xattn_pack): one coalesced 1-KiB load per fragment, shared through L1/L2 by
// the waves of the same head.  Gathering them from the row-major caches cost 48 eight-byte loads with 32 different rows per
// instruction -- 11 us per projection, as much as the attention kernel this fusion removes.
__device__ __forceinline__ void xattn_load_frags(const IgemmParams& p, int mw, int nw, int lane, half8 (&kf)[3][4], half8 (&vf)[2][6]) {
  const int mclamp = mw < p.M ? mw : p.M - 1;
  const int b = __builtin_amdgcn_readfirstlane(mclamp / p.rpb);       // rpb % WM == 0: one batch entry per wave tile
  const int head = (nw < p.N ? nw : 0) >> 6;                          // zero-padded weight columns: any valid head (never used)
  const half8* fx = reinterpret_cast<const half8*>(p.xa_k) + ((size_t)b * (p.N >> 6) + head) * (24 * 64) + lane;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kf[t][s4] = fx[(t * 4 + s4) * 64];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) vf[dt][s6] = fx[(12 + dt * 6 + s6) * 64];
}
template <int TM>
__device__ __forceinline__ void xattn_inplace(const IgemmParams& p, f32x16 (&acc)[TM][2], int mw, int nw, int lane,
                                              float (&lnA)[TM], float (&lnC)[TM], const void* zeros,
                                              const half8 (&kf)[3][4], const half8 (&vf)[2][6]) {
  if (nw >= p.N) return;                                              // zero-padded weight columns: nothing is stored
  const int fr = lane & 31, fh = lane >> 5;
  const f32x4* zv = reinterpret_cast<const f32x4*>(zeros);
  const int nctx = p.xa_nctx;
  f32x4 cz[2][4], bz[2][4];                                          // folded-LayerNorm column sums, bias (beta W of the folded norm)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nb = nw + j * 32 + 8 * q + 4 * fh;
      cz[j][q] = *(p.ln_stat ? reinterpret_cast<const f32x4*>(p.ln_cs + nb) : zv);
      bz[j][q] = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nb) : zv);
    }
  const float sc = p.xa_scale * 1.44269504088896340736f;             // p = exp2(s - m)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const float lna = lnA[i], lnc = lnC[i];
    half8 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int j = s4 >> 1, qq = s4 & 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = 2 * qq + (e >> 2), r = e & 3;
        qf[s4][e] = (half_t)((lna * acc[i][j][8 * qq + e] + lnc * cz[j][q][r] + bz[j][q][r]) * sc);
      }
    }
    f32x16 sv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[t][r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) sv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[t][s4], qf[s4], sv[t], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (32 * t + 8 * (r >> 2) + 4 * fh + (r & 3) >= nctx) sv[t][r] = -INFINITY;
        mx = fmaxf(mx, sv[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
    half8 pf[6];
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      float ls = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pe = __builtin_amdgcn_exp2f(sv[s6 >> 1][8 * (s6 & 1) + e] - mx);
        ls += pe;
        pf[s6][e] = (half_t)pe;
      }
      l += ls;
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt][s6], pf[s6], o, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][dt][r] = o[r] * inv;
    }
    lnA[i] = 1.f; lnC[i] = 0.f;                                       // the staged store adds nothing more
  }
}

template <int BM, int BN, int NS, bool PRIO, int DMODE = 0, int WGM = 4, int NW = 8, bool UNR = false, typename T = half_t, int PF = 0, bool XA = false>
__global__ __launch_bounds__(64 * NW) void igemm_pipe_kernel(const IgemmParams p, const void* zeros) {
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
  int rb[AJ], ry[AJ], rx[AJ], rsw[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (j * NW + wave) * 8 + lrow;
    const int m = m0 + row;
    rsw[j] = (slot ^ ((row >> 1) & 7)) * CE;
    if (m < p.M) {
      const int b = m / HWo;
      const int rem = m - b * HWo;
      const int oy = rem / p.Wout;
      rb[j] = b; ry[j] = oy * p.stride - p.pad; rx[j] = (rem - oy * p.Wout) * p.stride - p.pad;
    } else { rb[j] = -1; ry[j] = -(1 << 28); rx[j] = 0; }
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
  auto retap = [&]() {
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
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(vo), "s"(sb) : "memory");
          } else if (q - AJ < BJ - 1 || lastb) {
            const unsigned m = lds0 + buf * STAGE + BM * 128 + wave * 1024 + (q - AJ) * (NW * 1024), vo = woff[q - AJ];
            const unsigned long long sb = wbase;
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m), "v"(vo), "s"(sb) : "memory");
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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
  // ---- prologue: tiles 0 .. NPRO-1 in flight, wait for tile 0 only
#pragma unroll
  for (int s = 0; s < NPRO; ++s)
    if (s < nk) { issue(s, IALL{}); tile_done(); }
  float lnA[TM], lnC[TM];
  const bool ln_coop = LN_COOP && p.ln_slots <= 24;
  if constexpr (LN_COOP) lnc.finish(p, m0, ln_coef);
  if (!ln_coop) ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  if (NPRO <= nk) wait_tiles(std::integral_constant<int, NPRO - 1>{}); else wait_vmcnt<0>();
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
          if (more) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
        }
        wait_lgkmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
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
  constexpr bool FITS = NW * WM * WN * 4 <= NS * STAGE;        // full-width staging regions fit the dead ring
  if (FITS || p.act == 1) {
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
template <int NS>
__global__ __launch_bounds__(512) void igemm_wide_kernel(const IgemmParams p, const void* zeros) {
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
  half8 fA[TM], fB[TN];
  auto ldfrag = [&](unsigned so, int kk) {       // chunk(kk) = (kk*2 + fh) ^ sw = (fh ^ sw) ^ (kk << 1) -> byte offset ^ (kk << 5)
    const unsigned aa = (basea ^ (kk << 5)) + so, ab = (baseb ^ (kk << 5)) + so;
    static_for<TM>([&](auto I) { fA[decltype(I)::value] = lds_read128<decltype(I)::value * 32 * ROWB>(aa); });
    static_for<TN>([&](auto J) { fB[decltype(J)::value] = lds_read128<decltype(J)::value * 32 * ROWB>(ab); });
  };
  // ten MFMAs of one kk-step; with DMA: the PER pieces of the next tile go behind MFMAs 0, 2, 4, 6, 8
  auto mma = [&](int buf, bool dma) {
    __builtin_amdgcn_s_setprio(1);
    static_for<TM * TN>([&](auto X) {
      constexpr int x = decltype(X)::value, i = x / TN, j = x % TN;
      if constexpr (i == 0) acc0[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[j], fA[0], acc0[0][j], 0, 0, 0);
      else acc1[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fB[j], fA[1], acc1[0][j], 0, 0, 0);
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
  // ---- prologue: the whole ring in flight, wait for tile 0 only
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (s < nk) static_for<PER>([&](auto Q) { issue(s, Q); });
  float lnA[TM], lnC[TM];
  const bool ln_coop = p.ln_slots <= 24;
  lnc.finish(p, m0, ln_coef);
  if (!ln_coop) ln_prologue<TM>(p, m0 + wm * WM, fr, lnA, lnC);
  if (NS <= nk) wait_tiles(std::integral_constant<int, NS - 1>{}); else wait_vmcnt<0>();
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
  // (zeroed only now: keeping 160 accumulator registers live across the statistics loads of ln_prologue spills)
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[0][j][r] = 0.f; acc1[0][j][r] = 0.f; }
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned so = cur * STAGE;
    ldfrag(so, 0);
    wait_lgkmcnt<0>();
    mma(cur, false);
    ldfrag(so, 1);
    wait_lgkmcnt<0>();                          // own reads of tile kt complete
    const bool more = kt + NS < nk;
    if (kt + 1 < nk) {
      if (kt + NS - 1 < nk) wait_tiles(std::integral_constant<int, NS - 2>{}); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();             // tile kt+1 visible; slot of tile kt free for tile kt+NS
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    mma(cur, more);
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  __builtin_amdgcn_s_barrier();                  // ring dead -> staging area
  asm volatile("" ::: "memory");
  if (p.act == 1) {
    // two passes of 32 rows: a full 64 x 80 fp32 staging region per wave would not fit next to seven others
    char* region = smem + wave * (32 * (WN / 2) * 4);
    {
      const float la1[1] = {lnA[0]}, lc1[1] = {lnC[0]};
      igemm_epilogue_staged_impl<1, TN, true>(p, acc0, m0 + wm * WM, n0 + wn * WN, lane, region, false, la1, lc1, zeros);
    }
    {
      const float la1[1] = {lnA[1]}, lc1[1] = {lnC[1]};
      igemm_epilogue_staged_impl<1, TN, true>(p, acc1, m0 + wm * WM + 32, n0 + wn * WN, lane, region, false, la1, lc1, zeros);
    }
  }   // (the launcher only admits GEGLU projections)
}

#ifdef SDXL_MEASURE   // experiment that lost its A/B: built only by `build.py --measure`
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

#endif  // SDXL_MEASURE

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

template <int BM, int BN, int NS, bool PRIO, int DMODE = 0, int WGM = 4, int NW = 8, bool UNR = false, typename T = half_t, int PF = 0, bool XA = false>
static void launch_pipe(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)pipe_lds_total(NS * (BM + BN) * 128, PF > 0 ? NW * 256 : 0);   // ring + L2-prefetch scratch + LayerNorm coefficients
  static bool attr_set[kMaxDev] = {};
  const int dev = current_device();
  set_lds_attr(&igemm_pipe_kernel<BM, BN, NS, PRIO, DMODE, WGM, NW, UNR, T, PF, XA>, lds, attr_set, dev);
  const int sk = (BN == 128 && DMODE == 0 && p.splitk > 1) ? p.splitk : 1;
  if (sk > 1 && (size_t)tilesM * tilesN * sk * BM * BN * 4 > p.splitk_ws_bytes) throw std::runtime_error("igemm: split-K workspace too small");
  IgemmParams q = p;
  q.splitk = sk;
  hipLaunchKernelGGL((igemm_pipe_kernel<BM, BN, NS, PRIO, DMODE, WGM, NW, UNR, T, PF, XA>), dim3(tilesM * tilesN * sk), dim3(64 * NW), lds, s, q, g_zero_pages[dev]);
}

#ifdef SDXL_MEASURE
template <int BM, int BN, int NS, int NL>
static void launch_ws(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  static bool attr_set[kMaxDev] = {};
  const int dev = current_device();
  set_lds_attr(&igemm_ws_kernel<BM, BN, NS, NL>, lds, attr_set, dev);
  hipLaunchKernelGGL((igemm_ws_kernel<BM, BN, NS, NL>), dim3(tilesM * tilesN), dim3(512 + 64 * NL), lds, s, p, g_zero_pages[dev]);
}

#endif  // SDXL_MEASURE

static void launch_wide(const IgemmParams& p, hipStream_t s) {
  constexpr int NS = 4;
  const int tilesM = (p.M + 255) / 256, tilesN = (p.N + 319) / 320;
  const size_t lds = (size_t)NS * (256 + 320) * 64 + 2048;   // ring + LayerNorm coefficients
  static bool attr_set[kMaxDev] = {};
  const int dev = current_device();
  set_lds_attr(&igemm_wide_kernel<NS>, lds, attr_set, dev);
  hipLaunchKernelGGL((igemm_wide_kernel<NS>), dim3(tilesM * tilesN), dim3(512), lds, s, p, g_zero_pages[dev]);
}


// variant: 0 auto; 1 = 128x128 ring 3; 2 = 128x64 ring 4; 3 = 64x128 ring 4; 4 = 128x128 ring 2; 5 = 128x64 ring 2;
// 6 = 64x128 ring 2; 7 = 128x128 ring 4; 8 = 64x128 ring 3.  Returns false when the shape needs the generic kernel.
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
  if (p.rpb <= 0 || p.rpb > 1024 || p.N > 1280 || nk < 160) return 1;
  return 3;
}
size_t igemm_splitk_ws_bytes(int batch, int rows_per_entry, int n_max) {
  const long rows = (long)batch * (rows_per_entry < 1024 ? rows_per_entry : 1024);
  if (n_max > 1280) n_max = 1280;          // wider outputs never split (igemm_splitk_slices)
  return (size_t)((rows + 255) / 256) * (size_t)((n_max + 127) / 128) * 3 * 256 * 128 * 4;
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

// GroupNorm statistics from the producing GEMM's epilogue (IgemmParams::gn_part): taken by the 256x128 kernel (plain or split-K)
// when that is the tile the selection picks anyway, whole 256-row tiles inside one batch entry, f16 operands, plain epilogue.
bool igemm_gn_part_ok(const IgemmParams& p) {
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
  return a_dt == DT_F16 && c_dt == DT_F16 && M > 0 && N % 64 == 0 && K % 64 == 0 && rpb > 0 && rpb % 64 == 0 && M % rpb == 0 &&
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
  if (p.xa_k) {
    // fused cross-attention: wave tiles of 64 columns only (256x128 / 128x128), chosen by the same cost model
    if (p.act != 0 || p.n_split < p.N || p.stat_out || p.R || p.ebias ||
        !igemm_xattn_ok(p.a_dt, p.c_dt, p.M, p.N, p.K, p.rpb, p.xa_nctx))
      throw std::runtime_error("igemm: fused cross-attention needs a plain f16 projection (no residual / split outputs)");
    const int nk = p.Kpad / 64;
    int v = variant;
    if (v != 35 && v != 36 && v != 44 && v != 45) {
      const double c256 = tile_cost(p.M, p.N, nk, 256, 128, 1.0), c128 = tile_cost(p.M, p.N, nk, 128, 128, 1.0), c96 = tile_cost(p.M, p.N, nk, 96, 128, 1.0);
      v = c256 <= c128 && c256 <= c96 ? 35 : (c96 < c128 ? 45 : (nk >= 40 ? 44 : 36));
    }
    if (v == 35) launch_pipe<256, 128, 3, false, 0, 4, 8, true, half_t, 0, true>(psk, s);
    else if (v == 36) launch_pipe<128, 128, 4, false, 0, 4, 8, true, half_t, 0, true>(psk, s);
    else if (v == 44) launch_pipe<128, 128, 5, false, 0, 4, 8, true, half_t, 0, true>(psk, s);
    else launch_pipe<96, 128, 5, false, 0, 3, 6, true, half_t, 0, true>(psk, s);
    return true;
  }
  if (p.gn_part) {
    if (!igemm_gn_part_ok(p)) throw std::runtime_error("igemm: GroupNorm statistics requested from a shape the 256x128 epilogue does not take");
    variant = 0;            // (a forced test variant must not drop the statistics)
  }
  if (variant == 0 && igemm_splitk_slices(p) > 1) {
    // long contractions over a small output (FF-out and the 32^2 convs of the CFG pair: M = 2048, N = 1280 is 80 tiles of
    // 256x128 on 256 CUs): three k-slices per tile fill the chip with the tile shape that moves the fewest bytes per flop
    psk.splitk = igemm_splitk_slices(p);
    launch_pipe<256, 128, 3, false, 0, 4, 8, true>(psk, s);
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
    case 35: launch_pipe<256, 128, 3, false, 0, 4, 8, true>(psk, s); break;   // 8 waves, hand-ordered k-loop unrolled by the ring depth
    case 36: launch_pipe<128, 128, 4, false, 0, 4, 8, true>(psk, s); break;
    case 38:                                                                // 256x160 GEGLU tile (8x1 waves)
      if (p.N % 160 != 0) return false;
      launch_pipe<256, 160, 3, false, 0, 8, 8, true>(psk, s); break;
    case 44: launch_pipe<128, 128, 5, false, 0, 4, 8, true>(psk, s); break;   // 5-slot ring = all 160 KiB of LDS: 4 tiles in flight
    case 45: launch_pipe<96, 128, 5, false, 0, 3, 6, true>(psk, s); break;    // 6 waves (3 x 2), 96-row tile: M = 2048 x N = 1280 -> 220 workgroups
    case 46: launch_pipe<96, 128, 4, false, 0, 3, 6, true>(psk, s); break;
    case 49:                                                                // 4 waves 4x1 (32x160 wave tiles): N = 640 at 64^2 -> exactly 256 tiles
      if (p.N % 160 != 0 || p.stat_out) return false;
      launch_pipe<128, 160, 3, false, 0, 4, 4, true>(psk, s); break;
    case 26:                                                                // 256x320, k-tile 32: linear GEGLU projections only
      if (p.act != 1 || p.ksize != 1 || p.stride != 1 || p.up != 0 || p.N % 320 != 0 || p.Kpad % 32 != 0) return false;
      launch_wide(psk, s); break;
#ifdef SDXL_MEASURE
    case 40: launch_pipe<256, 128, 3, false, 0, 4, 8, true, half_t, 4>(psk, s); break;   // + L2 prefetch touches 4 / 8 k-tiles ahead:
    case 41: launch_pipe<128, 128, 4, false, 0, 4, 8, true, half_t, 4>(psk, s); break;   //   measured SLOWER (profiles/r02_l2_prefetch_ab.txt)
    case 42: launch_pipe<128, 128, 4, false, 0, 4, 8, true, half_t, 8>(psk, s); break;
    case 43: launch_pipe<256, 128, 3, false, 0, 4, 8, true, half_t, 8>(psk, s); break;
    // ---- A/B partners and experiments (build.py --measure): rolled loops, other rings, loader waves, measurement modes
    case 1: launch_glds<128, 128, 3>(psk, s); break;
    case 2: launch_glds<128, 64, 4>(psk, s); break;
    case 3: launch_glds<64, 128, 4>(psk, s); break;
    case 5: launch_glds<128, 64, 2>(psk, s); break;
    case 7: launch_glds<128, 128, 4>(psk, s); break;
    case 8: launch_glds<64, 128, 3>(psk, s); break;
    case 33: launch_glds<256, 128, 3, 1>(psk, s); break;
    case 34: launch_pipe<256, 128, 3, true, 0, 2, 4>(psk, s); break;   // the hand-ordered loop on 4 waves x (128x64)
    case 37: launch_pipe<256, 128, 3, true, 0, 4, 8, true>(psk, s); break;    // unrolled ring with s_setprio
    case 10: launch_pipe<256, 128, 3, false>(psk, s); break;   // rolled 8-wave pipelined kernels
    case 11: launch_pipe<256, 128, 3, true>(psk, s); break;
    case 12: launch_pipe<128, 128, 4, false>(psk, s); break;
    case 13: launch_pipe<128, 128, 4, true>(psk, s); break;
    case 14: launch_pipe<128, 128, 3, true>(psk, s); break;
    case 15: launch_pipe<256, 128, 3, true, 1>(psk, s); break;    // early DMA issue (after the barrier)
    case 16: launch_pipe<128, 128, 4, true, 1>(psk, s); break;
    case 17: launch_pipe<256, 128, 3, true, 2>(psk, s); break;    // measurement only: no k advance (WRONG results)
    case 18: launch_pipe<256, 128, 3, true, 3>(psk, s); break;    // measurement only: no DMA in the loop (WRONG results)
    case 24: launch_pipe<256, 128, 3, true, 4>(psk, s); break;    // lookahead-2 fragment prefetch
    case 27: launch_pipe<256, 128, 3, true, 5>(psk, s); break;    // measurement only: DMA-only / contiguous-source modes
    case 28: launch_pipe<256, 128, 3, true, 6>(psk, s); break;
    case 29: launch_pipe<256, 128, 3, true, 7>(psk, s); break;
    case 30: launch_pipe<128, 128, 4, true, 5>(psk, s); break;
    case 31: launch_pipe<128, 128, 4, true, 6>(psk, s); break;
    case 32: launch_pipe<128, 128, 4, true, 7>(psk, s); break;
    case 25: launch_pipe<128, 128, 4, true, 4>(psk, s); break;
    case 19:                                                    // 256x160, 8x1 waves, rolled
      if (p.N % 160 != 0) return false;
      launch_pipe<256, 160, 3, true, 0, 8>(psk, s); break;
    case 20: launch_ws<256, 128, 3, 2>(psk, s); break;            // 8 compute + 2 loader waves
    case 21: launch_ws<256, 128, 3, 4>(psk, s); break;            // 8 compute + 4 loader waves
    case 22: launch_ws<128, 128, 4, 2>(psk, s); break;
    case 23: launch_ws<128, 128, 4, 4>(psk, s); break;
#endif
    default: return false;
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
  if (eff256 >= eff128) launch_pipe<256, 128, 3, false, 0, 4, 8, true, float>(q, s);
  else launch_pipe<128, 128, 4, false, 0, 4, 8, true, float>(q, s);
  return true;
}

}  // namespace sdxl
