// Device-side building blocks shared by the production implicit-GEMM kernels (igemm_glds.hip) and the measurement build's
// A/B partners (igemm_measure.hip, -DSDXL_MEASURE only): epilogues (direct / LDS-staged, GEGLU, folded LayerNorm, row and
// GroupNorm statistics, transposed V^T store), the cooperative LayerNorm coefficients, fragment reads, the fused cross-attention.
#pragma once
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
typedef float f32x8 __attribute__((ext_vector_type(8)));
// Wave-uniform read-only vectors (column sums, biases): eight floats through the SCALAR cache (s_load_dwordx8) instead of a
// 64-lane VMEM request whose lanes all ask for the same 16 bytes.  `ptr` must be wave-uniform and 4-byte aligned; the memory is
// never written by the running kernel (constant address space).
__device__ __forceinline__ f32x8 uniform_load8(const float* ptr) {
  const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
  const unsigned long long u = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return *(const __attribute__((address_space(4))) f32x8*)(uintptr_t)u;
}
// the four values of this lane's half (fh = lane >> 5) of the eight columns from `nbu` of a per-column vector; a null / absent
// vector reads the zero page.  Vectors are padded to Npad, so the eight-column read never leaves the array.
__device__ __forceinline__ f32x4 col_vec4(const float* arr, bool present, int nbu, int fh, const void* zeros) {
  const f32x8 c = uniform_load8(present ? arr + nbu : reinterpret_cast<const float*>(zeros));
  return fh ? f32x4{c[4], c[5], c[6], c[7]} : f32x4{c[0], c[1], c[2], c[3]};
}
typedef __attribute__((address_space(3))) void* lptr_t;

// Split-operand element (DT_HL): a logical fp32 value x travels as TWO f16 numbers, hi = f16(x) and lo = f16(x - hi), i.e. 22
// significand bits, and a product is three f16 MFMAs accumulated in fp32:  a * w ~ ah*wh + al*wh + ah*wl  (the dropped al*wl is
// 2^-22 relative).  fp32-class accuracy at 1/3 of the f16 MFMA rate instead of 1/16 (v_mfma_f32_32x32x2_f32 runs at the fp32
// VECTOR rate, 157 TFLOP/s) -- the VAE at the reference's precision (src/bin/sample/main.rs:121,271-278) without the fp32 pipe.
// Memory format "HL16": 4 bytes per logical element like fp32; every group of 16 channels is 32 halfs [16 hi | 16 lo], so a
// 128-byte tile row of the DMA ring holds 32 logical k: chunks {0,1} = hi of k 0..15, {2,3} = lo of k 0..15, {4,5} / {6,7} = the
// same for k 16..31 -- the four kk-steps of the pipelined loop read hi0, lo0, hi1, lo1 with UNCHANGED fragment addressing.
struct hl16_t { unsigned v; };
template <typename T> struct is_hl { static constexpr bool value = false; };
template <> struct is_hl<hl16_t> { static constexpr bool value = true; };
// 8 consecutive logical columns n0 .. n0+7 (n0 % 8 == 0) of an HL16 row: hi halfs at (n0/16)*32 + n0%16, lo 16 halfs further
__device__ __forceinline__ void store_hl8(void* row_base, int n0, const float (&w)[8]) {
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    // the value is pinned in its fp32 form first: otherwise the compiler may fuse the multiply / fma that produced it into the
    // f16 conversion (v_fma_mixlo_f16: ONE rounding of the exact result) for hi while lo is taken against the doubly rounded
    // value, and near a tie the two disagree by a whole f16 ulp of hi (found with one-hot operands: tools/attn_split_err.py)
    float x = w[e];
    asm("" : "+v"(x));
    hi[e] = (half_t)x; lo[e] = (half_t)(x - (float)hi[e]);
  }
  half_t* b = reinterpret_cast<half_t*>(row_base) + ((n0 >> 4) << 5) + (n0 & 15);
  *reinterpret_cast<half8*>(b) = hi;
  *reinterpret_cast<half8*>(b + 16) = lo;
}

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

// Kernel-argument prefetch.  hipcc loads kernel arguments lazily (s_load right before the first use of each field), and a struct
// as large as IgemmParams spans five 64-byte lines: the s_memtime timeline of the production kernels (tools/timeline_probe.py,
// profiles/r03_timeline_*.txt) shows one ~1.5 k-cycle scalar-cache miss to memory at kernel entry, another in the DMA geometry,
// and three to four more in the epilogue (bias / R / C / ln_* / stat_out live in different lines) -- ~3 us of a 17 us GEMM.
// Touching every line at entry overlaps all of those misses into the one wait the kernel pays anyway; the later s_loads hit
// the scalar cache.  (All loads and the wait sit in ONE asm statement: SMEM returns are asynchronous and the compiler does not
// know these are loads.)
template <int BYTES> __device__ __forceinline__ void kernarg_prefetch() {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  static_assert(BYTES > 64 && BYTES <= 64 * 8, "kernarg_prefetch: 2..8 lines");
  constexpr int L = (BYTES + 63) / 64;
  unsigned d1, d2, d3, d4, d5, d6, d7;
  if constexpr (L <= 2) asm volatile("s_load_dword %0, %1, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1) : "s"(ka) : "memory");
  else if constexpr (L == 3) asm volatile("s_load_dword %0, %2, 0x40\n\ts_load_dword %1, %2, 0x80\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1), "=s"(d2) : "s"(ka) : "memory");
  else if constexpr (L == 4) asm volatile("s_load_dword %0, %3, 0x40\n\ts_load_dword %1, %3, 0x80\n\ts_load_dword %2, %3, 0xc0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1), "=s"(d2), "=s"(d3) : "s"(ka) : "memory");
  else if constexpr (L == 5) asm volatile("s_load_dword %0, %4, 0x40\n\ts_load_dword %1, %4, 0x80\n\ts_load_dword %2, %4, 0xc0\n\ts_load_dword %3, %4, 0x100\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1), "=s"(d2), "=s"(d3), "=s"(d4) : "s"(ka) : "memory");
  else if constexpr (L == 6) asm volatile("s_load_dword %0, %5, 0x40\n\ts_load_dword %1, %5, 0x80\n\ts_load_dword %2, %5, 0xc0\n\ts_load_dword %3, %5, 0x100\n\ts_load_dword %4, %5, 0x140\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1), "=s"(d2), "=s"(d3), "=s"(d4), "=s"(d5) : "s"(ka) : "memory");
  else asm volatile("s_load_dword %0, %7, 0x40\n\ts_load_dword %1, %7, 0x80\n\ts_load_dword %2, %7, 0xc0\n\ts_load_dword %3, %7, 0x100\n\ts_load_dword %4, %7, 0x140\n\ts_load_dword %5, %7, 0x180\n\ts_load_dword %6, %7, 0x1c0\n\ts_waitcnt lgkmcnt(0)" : "=s"(d1), "=s"(d2), "=s"(d3), "=s"(d4), "=s"(d5), "=s"(d6), "=s"(d7) : "s"(ka) : "memory");
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

// body of a weight-warming workgroup (IgemmParams::warm): NT threads read `bytes` at `base` -- workgroup wi of nw takes every nw-th
// run of 8 x NT x 16 bytes, 8 loads in flight per lane -- and drop them: what matters is that the lines now sit in the memory-side cache
typedef int warm_i32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void igemm_warm_body(const void* base, unsigned bytes, int wi, int nw) {
  const char* b = reinterpret_cast<const char*>(base);
  constexpr unsigned span = NT * 16, step = span * 8;
  warm_i32x4 acc = {0, 0, 0, 0};
  for (size_t off = (size_t)wi * step; off < bytes; off += (size_t)nw * step) {
    warm_i32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = off + (size_t)u * span + (size_t)threadIdx.x * 16;
      v[u] = o + 16 <= bytes ? *reinterpret_cast<const warm_i32x4*>(b + o) : warm_i32x4{0, 0, 0, 0};      // (plain loads: nontemporal ones measured no warming at all)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  asm volatile("" ::"v"(acc));
}

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
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(lna, acc[i][j][q * 4 + r], __builtin_fmaf(lnc, cz[q][r], bz[q][r]));   // (written out: igemm_epilogue_swapped rounds alike)
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
    // lean form for whole, aligned f16 tiles without residual or statistics (every GEGLU projection of the UNet): the general loop
    // below spends most of its instructions on per-item validity / alignment selects and 64-bit pointer arithmetic -- the two staged
    // passes of the wide GEGLU kernel are 21 k of its 106 k cycles (profiles/r04_wide_geglu_timeline.txt)
    if (GEGLU && !p.R && !p.stat_out && !gc && p.c_dt == DT_F16 && mw + ROWS <= p.M && nwo + COLS <= nlim && (p.ldc & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
      half_t* cbase = reinterpret_cast<half_t*>(p.C) + (size_t)mw * p.ldc + nwo;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / LPR, piece = idx - row * LPR;
        if (ITEMS % 64 != 0 && idx >= ITEMS) continue;
        const f32x4 a = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece) ^ (row & SWN)) << 4));
        const f32x4 b = *reinterpret_cast<const f32x4*>(lds + row * RB + (((2 * piece + 1) ^ (row & SWN)) << 4));
        half8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (half_t)a[e]; h[4 + e] = (half_t)b[e]; }
        *reinterpret_cast<half8*>(cbase + row * p.ldc + piece * 8) = h;
      }
      return;
    }
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
      } else if (p.c_dt == DT_HL) {       // (whole 8-column pieces only: the launcher admits N % 8 == 0, ldc % 16 == 0 outputs)
        if (full) store_hl8(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc, n0, v);
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
      } else if (p.c_dt == DT_HL) {       // V^T rows in HL16 along the key axis (whole 8-key pieces inside one batch entry)
        if (full) store_hl8(reinterpret_cast<float*>(p.Ct) + ((size_t)b0 * p.ct_rows + (n - p.n_split)) * p.ct_ld, key0, v);
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

// ---- direct row-per-lane epilogue (whole wave tiles; the default).  Instruction-level timeline of the production kernels
// (profiles/r03_timeline_*.txt, tools/timeline_probe.py): the LDS-staged epilogue above takes 6.4 k cycles per 32x64 wave tile and
// 12.6 k per 64x64 -- 3.7 / 5.6 us of every launch, with the stores themselves draining in ~200 cycles: it is bound by its own
// instruction stream (fp32 LDS round trip, per-item address / validity arithmetic), not by memory.  Here the accumulators never
// leave the registers: in the operand-swapped 32x32 MFMA layout a lane holds, per 32-column tile, 4 groups of 4 consecutive
// columns of ONE row (columns 8q + 4(lane>>5) + r), its partner lane^32 the other 4-column halves of the same row.  One
// v_permlane32_swap per dword and group pair (q, q+1) leaves lanes 0..31 with columns 16t .. 16t+7 and lanes 32..63 with 16t+8 ..
// 16t+15 of their row (cdna_hip_programming.md T21): 8 consecutive outputs = ONE 16-byte f16 store (two for fp32) and one 16-byte
// residual load per piece.  Bias / folded-LayerNorm affine / time-embedding bias / GEGLU are applied before the swap in the
// accumulator layout (same expressions as the staged path); the residual is added, the result rounded once, and the row
// statistics of the stored values (stat_out) are taken behind it.
// Row statistics: the 64 columns of a row slot sit in pieces 0,2,4,6 (lanes 0..31) and 1,3,5,7 (lanes 32..63); per-piece shifted
// sums around the slot's first stored value, then the same pairwise tree as the staged path -- (p0+p1), (p2+p3), ... across the
// lane halves first, then in-lane -- so both paths produce the same (mean, M2) bits for the same stored values.
template <int TM, int TN, bool GEGLU>
__device__ __forceinline__ bool igemm_rows_ok(const IgemmParams& p, int nw) {
  constexpr int WN = TN * 32;
  const int nlim = GEGLU ? (p.N >> 1) : (p.n_split < p.N ? p.n_split : p.N);
  const int n_lo = GEGLU ? (nw >> 1) : nw, n_w = GEGLU ? WN / 2 : WN;
  if (n_lo + n_w > nlim || p.gn_part) return false;
  if ((p.ldc & 7) != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0) return false;
  if (p.R && ((p.ldr & 7) != 0 || (reinterpret_cast<uintptr_t>(p.R) & 15) != 0)) return false;
  if (p.stat_out && (TN != 2 || GEGLU)) return false;
  if (p.ebias && (p.ebias_ld & 3) != 0) return false;
  return true;
}
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };   // (timeline twins pass an s_memtime stamper)
template <int TM, int TN, bool GEGLU, typename ST = NoStamp>
__device__ __forceinline__ void igemm_epilogue_rows(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int lane,
                                                    const float (&lnA)[TM], const float (&lnC)[TM], const void* zeros, const ST& stamp = ST()) {
  const int fr = lane & 31, fh = lane >> 5;
  const f32x4* zv = reinterpret_cast<const f32x4*>(zeros);
  constexpr int NQ = GEGLU ? 2 : 4;            // value groups of 4 columns per 32-column MFMA tile and lane
  constexpr int NP = NQ / 2;                   // 8-column pieces per tile and lane after the half swap
  const int nwo = GEGLU ? (nw >> 1) : nw;
  int m[TM], bidx[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    m[i] = mw + i * 32 + fr;
    mok[i] = m[i] < p.M;
    bidx[i] = (p.ebias && mok[i]) ? m[i] / p.rpb : 0;
  }
  // residual pieces: requested first (oldest entries of the vmcnt queue), consumed last
  const bool r16 = p.R && p.r_dt == DT_F16, r32 = p.R && p.r_dt == DT_F32;
  half8 rh[TM][TN][NP];
  f32x4 rf[TM][TN][NP][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int n0 = nwo + (GEGLU ? j * 16 : j * 32 + t * 16) + 8 * fh;
        const size_t o = (size_t)(mok[i] ? m[i] : 0) * p.ldr + n0;
        rh[i][j][t] = *((r16 && mok[i]) ? reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(p.R) + o) : reinterpret_cast<const half8*>(zeros));
        if (r32) {   // (fp32 residual stream: wave-uniform branch, the f16 engines never take it)
          rf[i][j][t][0] = *(mok[i] ? reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + o) : zv);
          rf[i][j][t][1] = *(mok[i] ? reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + o + 4) : zv);
        }
      }
  stamp(0);      // residual requested
  float st_s1[TM][4], st_s2[TM][4], st_piv[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    st_piv[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { st_s1[i][k] = 0.f; st_s2[i][k] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nt = nw + j * 32;
    // per-column vectors in the accumulator layout (pointer selects: one wait for the lot, see the staged path)
    f32x4 bz[NQ], cz[NQ], gz[NQ], gc[NQ], ez[TM][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int nb = nt + 8 * q + 4 * fh;
      bz[q] = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nb) : zv);
      cz[q] = *(p.ln_stat ? reinterpret_cast<const f32x4*>(p.ln_cs + nb) : zv);
      if constexpr (GEGLU) {
        gz[q] = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nb + 16) : zv);
        gc[q] = *(p.ln_stat ? reinterpret_cast<const f32x4*>(p.ln_cs + nb + 16) : zv);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
        ez[i][q] = *(p.ebias ? reinterpret_cast<const f32x4*>(p.ebias + (size_t)bidx[i] * p.ebias_ld + nb) : zv);
    }
    stamp(1 + 2 * j);   // column vectors of tile j requested
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float lna = lnA[i], lnc = lnC[i];
      f32x4 v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[q][r] = acc[i][j][q * 4 + r];
        v[q] = lna * v[q] + lnc * cz[q] + bz[q] + ez[i][q];
        if constexpr (GEGLU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[q][r] *= gelu_erf2(lna * acc[i][j][(q + 2) * 4 + r] + lnc * gc[q][r] + gz[q][r]);
        }
      }
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        // half swap of the group pair (2t, 2t+1): lanes 0..31 keep group 2t and receive the partner's group 2t; lanes 32..63 receive
        // the partner's group 2t+1 and keep their own -> w[0..7] = 8 consecutive columns from n0
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * t][r]), __float_as_uint(v[2 * t + 1][r]), false, false);
          w[r] = __uint_as_float(sw[0]);
          w[4 + r] = __uint_as_float(sw[1]);
        }
        if (r16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] += (float)rh[i][j][t][e];
        } else if (r32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[e] += rf[i][j][t][0][e]; w[4 + e] += rf[i][j][t][1][e]; }
        }
        const int n0 = nwo + (GEGLU ? j * 16 : j * 32 + t * 16) + 8 * fh;
        if (p.c_dt == DT_F16) {
          half8 h;
#pragma unroll
          for (int e = 0; e < 8; ++e) h[e] = (half_t)w[e];
          if (mok[i]) *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.C) + (size_t)m[i] * p.ldc + n0) = h;
          if constexpr (TN == 2 && !GEGLU) {
            if (p.stat_out) {
#pragma unroll
              for (int e = 0; e < 8; ++e) w[e] = (float)h[e];      // the stored (rounded) values: what the consumer will read
            }
          }
        } else if (p.c_dt == DT_HL) {
          if (mok[i]) store_hl8(reinterpret_cast<float*>(p.C) + (size_t)m[i] * p.ldc, n0, w);
        } else if (mok[i]) {
          float* cp = reinterpret_cast<float*>(p.C) + (size_t)m[i] * p.ldc + n0;
          *reinterpret_cast<f32x4*>(cp) = f32x4{w[0], w[1], w[2], w[3]};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{w[4], w[5], w[6], w[7]};
        }
        if constexpr (TN == 2 && !GEGLU) {
          if (p.stat_out) {
            if (j == 0 && t == 0) st_piv[i] = __shfl(w[0], fr);      // the slot's first stored value (lanes 0..31 hold it)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = w[e] - st_piv[i]; s1 += d; s2 = fmaf(d, d, s2); }
            st_s1[i][j * 2 + t] = s1; st_s2[i][j * 2 + t] = s2;
          }
        }
      }
    }
  }
  stamp(7);      // all stores issued
  if constexpr (TN == 2 && !GEGLU) {
    if (p.stat_out) {
      // pieces 2k (lanes 0..31) and 2k+1 (lanes 32..63) pair up across the halves, then (p0+p1)+(p2+p3) and (p4+p5)+(p6+p7) in-lane
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { st_s1[i][k] += __shfl_xor(st_s1[i][k], 32); st_s2[i][k] += __shfl_xor(st_s2[i][k], 32); }
        const float s1 = (st_s1[i][0] + st_s1[i][1]) + (st_s1[i][2] + st_s1[i][3]);
        const float s2 = (st_s2[i][0] + st_s2[i][1]) + (st_s2[i][2] + st_s2[i][3]);
        if (fh == 0 && mok[i]) {
          float* dst = p.stat_out + ((size_t)(nw >> 6) * p.M + m[i]) * 2;
          dst[0] = st_piv[i] + s1 * (1.0f / 64.0f);
          dst[1] = fmaxf(s2 - s1 * s1 * (1.0f / 64.0f), 0.f);
        }
      }
    }
  }
}

// ---- direct TRANSPOSED epilogue for operand-swapped accumulators (igemm_pipe_kernel TSW: the V^T part of a fused QKV projection).
// With the activations as the MFMA A operand a lane holds, per 32x32 tile, ONE output column n = nw + 32 j + (lane & 31) and 4 groups of
// 4 consecutive ROWS m = mw + 32 i + 8 g + 4 (lane >> 5) + r -- consecutive keys of the transposed output Ct[b][n - n_split][key].  The
// half swap of igemm_epilogue_rows, on rows instead of columns, leaves a lane 8 consecutive keys: one 16-byte store per pair of groups.
// Bias and the folded-LayerNorm column sum are per lane (one column), the row coefficients (a, c) of the folded LayerNorm come from the
// workgroup's coefficient table in LDS (LnCoop; `coef` = the wave's first row, null without a folded LayerNorm).  The values are the staged
// path's: same products in the same k order, same affine expression, one rounding to f16.
template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue_swapped(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int lane,
                                                       const float* coef, const void* zeros) {
  (void)zeros;
  const int fr = lane & 31, fh = lane >> 5;
  const int b = mw / p.rpb;                          // (whole wave tiles inside one batch entry: rpb % BM == 0)
  const int key_w = mw - b * p.rpb;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 32 + fr;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const float cs = p.ln_stat ? p.ln_cs[n] : 0.f;
    half_t* crow = reinterpret_cast<half_t*>(p.Ct) + ((size_t)b * p.ct_rows + (n - p.n_split)) * p.ct_ld + key_w;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f32x4 v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a4 = {1.f, 1.f, 1.f, 1.f}, c4 = {0.f, 0.f, 0.f, 0.f};
        if (coef && p.ln_stat) {
          const f32x4 c01 = *reinterpret_cast<const f32x4*>(coef + (i * 32 + 8 * g + 4 * fh) * 2);        // (a, c) of rows r = 0, 1
          const f32x4 c23 = *reinterpret_cast<const f32x4*>(coef + (i * 32 + 8 * g + 4 * fh) * 2 + 4);    // rows r = 2, 3
          a4 = f32x4{c01[0], c01[2], c23[0], c23[2]};
          c4 = f32x4{c01[1], c01[3], c23[1], c23[3]};
        }
#pragma unroll
        // (the staged transposed path's expression, written out in both places: the two epilogues round alike, so a batch entry stays
        //  bit-identical whichever tile its batch size selects)
        for (int r = 0; r < 4; ++r) v[g][r] = __builtin_fmaf(a4[r], acc[i][j][4 * g + r], __builtin_fmaf(c4[r], cs, bias));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * t][r]), __float_as_uint(v[2 * t + 1][r]), false, false);
          w[r] = __uint_as_float(sw[0]);
          w[4 + r] = __uint_as_float(sw[1]);
        }
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (half_t)w[e];
        if (n < p.N) *reinterpret_cast<half8*>(crow + i * 32 + 16 * t + 8 * fh) = h;
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
template <> struct PipeElem<hl16_t> { typedef half8 frag; };
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

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
  const int fh = lane >> 5;
  const int nctx = p.xa_nctx;
  // folded-LayerNorm column sums and bias (beta W of the folded norm) of this wave's 64 columns: wave-uniform, so they come
  // through the scalar cache, eight columns per s_load; a lane keeps the four of its half (fh).  (The other epilogues keep the
  // 64-lane VMEM form of these vectors: scalar loads there measured +0.6 ms per UNet step -- SMEM waits are all-or-nothing.)
  f32x4 cz[2][4], bz[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nbu = nw + j * 32 + 8 * q;
#ifdef SDXL_MEASURE
      if (p.xa_vec64) {     // the ORIGINAL form: a 64-lane VMEM request whose lanes of one half all ask for the same 16 bytes (hazard experiment)
        const f32x4* zv = reinterpret_cast<const f32x4*>(zeros);
        cz[j][q] = *(p.ln_stat ? reinterpret_cast<const f32x4*>(p.ln_cs + nbu + 4 * fh) : zv);
        bz[j][q] = *(p.bias ? reinterpret_cast<const f32x4*>(p.bias + nbu + 4 * fh) : zv);
        continue;
      }
#endif
      cz[j][q] = col_vec4(p.ln_cs, p.ln_stat != nullptr, nbu, fh, zeros);
      bz[j][q] = col_vec4(p.bias, p.bias != nullptr, nbu, fh, zeros);
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

// ---- the same fusion at split precision (IgemmParams::xa_k_lo; round 6): q stays what the projection's fp32 accumulators hold -- it is split into
// (hi, lo) f16 pairs here, like the probabilities (times 2^11, so that the lo halves of small probabilities stay out of the f16 subnormals), and meets
// context keys / values that were split once per prompt: three MFMAs per product, fp32 accumulation, fp32 softmax -- the arithmetic of the stand-alone
// split-operand attention kernel (attention.hip attn_d64_hl_kernel), without its launch, without q's round trip through memory.  The 48 context
// fragments of a head (hi + lo of K and V^T, 48 KiB) do not fit next to the k-loop's registers, so nothing is requested early: K's 24 fragments are
// loaded at entry, V^T's 24 behind the S MFMAs (they land under the softmax arithmetic).  One 32-query row block per wave (TM = 1 kernels).
template <int TM>
__device__ __forceinline__ void xattn_inplace_hl(const IgemmParams& p, f32x16 (&acc)[TM][2], int mw, int nw, int lane,
                                                 float (&lnA)[TM], float (&lnC)[TM], const void* zeros) {
  static_assert(TM == 1, "split-precision fused cross-attention: one-MFMA-row wave tiles");
  if (nw >= p.N) return;                                              // zero-padded weight columns: nothing is stored
  const int fh = lane >> 5;
  const int nctx = p.xa_nctx;
  const int mclamp = mw < p.M ? mw : p.M - 1;
  const int b = __builtin_amdgcn_readfirstlane(mclamp / p.rpb);       // rpb % WM == 0: one batch entry per wave tile
  const size_t foff = ((size_t)b * (p.N >> 6) + (nw >> 6)) * (24 * 64) + lane;
  const half8* fhi = reinterpret_cast<const half8*>(p.xa_k) + foff;
  const half8* flo = reinterpret_cast<const half8*>(p.xa_k_lo) + foff;
  half8 kh[3][4], kl[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { kh[t][s4] = fhi[(t * 4 + s4) * 64]; kl[t][s4] = flo[(t * 4 + s4) * 64]; }
  f32x4 cz[2][4], bz[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nbu = nw + j * 32 + 8 * q;
      cz[j][q] = col_vec4(p.ln_cs, p.ln_stat != nullptr, nbu, fh, zeros);
      bz[j][q] = col_vec4(p.bias, p.bias != nullptr, nbu, fh, zeros);
    }
  constexpr float PSC = 2048.0f;                                      // P travels as P * 2^11 (attn_d64_hl_kernel)
  const float sc = p.xa_scale * 1.44269504088896340736f;             // p = exp2(s - m)
  const float lna = lnA[0], lnc = lnC[0];
  half8 qh[4], ql[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int j = s4 >> 1, qq = s4 & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int q = 2 * qq + (e >> 2), r = e & 3;
      float x = (lna * acc[0][j][8 * qq + e] + lnc * cz[j][q][r] + bz[j][q][r]) * sc;
      asm("" : "+v"(x));                                             // pinned in fp32: hi and lo come from the SAME rounded value (store_hl8)
      const half_t hi = (half_t)x;
      qh[s4][e] = hi; ql[s4][e] = (half_t)(x - (float)hi);
    }
  }
  f32x16 sv[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[t][r] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      sv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t][s4], qh[s4], sv[t], 0, 0, 0);
      sv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[t][s4], ql[s4], sv[t], 0, 0, 0);
      sv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[t][s4], qh[s4], sv[t], 0, 0, 0);
    }
  }
  half8 vh[2][6], vl[2][6];                                           // requested here: they land under the softmax arithmetic
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) { vh[dt][s6] = fhi[(12 + dt * 6 + s6) * 64]; vl[dt][s6] = flo[(12 + dt * 6 + s6) * 64]; }
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
  half8 ph[6], pl[6];
#pragma unroll
  for (int s6 = 0; s6 < 6; ++s6) {
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pe = __builtin_amdgcn_exp2f(sv[s6 >> 1][8 * (s6 & 1) + e] - mx);
      ls += pe;
      float ps = pe * PSC;
      asm("" : "+v"(ps));
      const half_t hi = (half_t)ps;
      ph[s6][e] = hi; pl[s6][e] = (half_t)(ps - (float)hi);
    }
    l += ls;
  }
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / (l * PSC);
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s6 = 0; s6 < 6; ++s6) {
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[dt][s6], ph[s6], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[dt][s6], pl[s6], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[dt][s6], ph[s6], o, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][dt][r] = o[r] * inv;
  }
  lnA[0] = 1.f; lnC[0] = 0.f;                                         // the store adds nothing more
}

// per-device state owned by igemm_glds.hip
const void* igemm_zero_page();          // null until igemm_glds_init() ran on the current device
int igemm_current_device();
constexpr int kIgemmMaxDev = 64;

}  // namespace sdxl
