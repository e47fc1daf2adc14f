// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4).
//
//   C[M,N] = gather(A)[M,K] * Wp[N,K]^T,  M = B*Hout*Wout (NHWC pixels / tokens), K = taps*Cin (cin fastest)
//
// One kernel covers every dense contraction of the SDXL hot path: conv3x3 (stride 1/2, optional fused
// nearest-2x upsample gather), conv1x1 and nn::Linear (a 1x1 conv in NHWC *is* a row-major GEMM).
// Structure (MI355X-first, not a warp-tiling port):
//   * 256 threads = 4 wavefronts (2x2), block tile BM x BN, k-tile = one 128-byte row segment per tile row
//     (64 f16 or 32 f32), so a conv k-tile never straddles a tap when Cin % 64 == 0 (all SDXL layers but the
//     4-/3-channel stems, which take the per-element gather path of the same kernel).
//   * global -> registers -> LDS staging (lets the gather zero-fill halo pixels and convert fp32 activations
//     to f16 on the fly), 16-byte chunks, XOR-swizzled LDS rows (chunk ^= row&7) so the ds_read_b128 fragment
//     reads of 16 different rows at one k-chunk spread over the 256-byte bank row.
//   * LDS double buffering, one barrier per k-tile; next tile's global loads are issued before the MFMAs.
//   * MFMA: v_mfma_f32_16x16x32_f16 (f16 mode) or 4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk (f32 strict
//     mode, bit-for-bit an fp32 fmaf chain); fp32 accumulation in both.
//   * fused epilogue: bias, per-batch time-embedding bias, GEGLU (x*gelu_erf(gate) on interleaved column
//     pairs), residual add, dtype conversion, and an optional transposed store (V^T for the attention kernel).
#include "kernels.h"
#include <atomic>
#include <stdexcept>
#include <hip/hip_fp16.h>

namespace sdxl {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<half_t> {
  static __device__ __forceinline__ f32x4 run(i32x4 a, i32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane group g=lane>>4 holds k = 4g..4g+3 of a 16-deep step; MFMA c consumes element c of every group.
  static __device__ __forceinline__ f32x4 run(i32x4 a, i32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// load one 16-byte compute-dtype chunk (CE elements) of A starting at element pointer `src` (dtype AT)
template <typename T, typename AT>
__device__ __forceinline__ i32x4 load_chunk(const AT* src) {
  if constexpr (sizeof(T) == sizeof(AT)) {
    return *reinterpret_cast<const i32x4*>(src);
  } else {  // T = f16, AT = f32: 8 floats -> 8 halfs
    f32x4 lo = *reinterpret_cast<const f32x4*>(src);
    f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
    half8 h;
    h[0] = (half_t)lo[0]; h[1] = (half_t)lo[1]; h[2] = (half_t)lo[2]; h[3] = (half_t)lo[3];
    h[4] = (half_t)hi[0]; h[5] = (half_t)hi[1]; h[6] = (half_t)hi[2]; h[7] = (half_t)hi[3];
    return __builtin_bit_cast(i32x4, h);
  }
}

template <typename T> __device__ __forceinline__ float load_as_float(const void* p, size_t i, int dt) {
  return dt == DT_F16 ? (float)reinterpret_cast<const half_t*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void store_from_float(void* p, size_t i, int dt, float v) {
  if (dt == DT_F16) reinterpret_cast<half_t*>(p)[i] = (half_t)v; else reinterpret_cast<float*>(p)[i] = v;
}

template <typename T, typename AT, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int CE = 16 / sizeof(T);   // elements per 16-byte chunk
  constexpr int KT = 8 * CE;           // elements per k-tile (128-byte rows)
  constexpr int AI = BM / 32;          // A chunks per thread per k-tile
  constexpr int BI = BN / 32;
  constexpr int TM = BM / 32;          // 16x16 MFMA tiles per wave along M (wave tile = BM/2 x BN/2)
  constexpr int TN = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                           // [2][BM][128 B]
  char* sB = smem + 2 * BM * 128;            // [2][BN][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> tile (XCD-aware: consecutive remapped ids sit on one XCD's L2 and share an A row panel)
  const int tilesN = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any nwg
  }
  const int tm = bid / tilesN, tn = bid - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread staging geometry: chunk column cc, rows r0 + 32*i
  const int cc = tid & 7;
  const int r0 = tid >> 3;
  const int HWo = p.Hout * p.Wout;
  const int Hup = p.Hin << p.up, Wup = p.Win << p.up;
  int rb[AI], ry[AI], rx[AI];   // batch index (or -1 if row beyond M), top-left input coords
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int b = m / HWo;
      const int rem = m - b * HWo;
      const int oy = rem / p.Wout;
      const int ox = rem - oy * p.Wout;
      rb[i] = b; ry[i] = oy * p.stride - p.pad; rx[i] = ox * p.stride - p.pad;
    } else { rb[i] = -1; ry[i] = 0; rx[i] = 0; }
  }
  const bool fastA = (p.Cin % KT) == 0 && (p.lda % CE) == 0;   // k-tiles never straddle a tap, chunks 16-byte aligned
  const AT* Ag = reinterpret_cast<const AT*>(p.A);
  const T* Wg = reinterpret_cast<const T*>(p.W);

  i32x4 ra[AI], rbv[BI];

  auto load_tile = [&](int kt) {
    const int kbase = kt * KT;
    if (fastA) {
      const int tap = kbase / p.Cin;
      const int c0 = kbase - tap * p.Cin + cc * CE;
      const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int iy = ry[i] + dy, ix = rx[i] + dx;
        const bool ok = rb[i] >= 0 && iy >= 0 && iy < Hup && ix >= 0 && ix < Wup;
        if (ok) {
          const size_t pix = ((size_t)rb[i] * p.Hin + (iy >> p.up)) * p.Win + (ix >> p.up);
          ra[i] = load_chunk<T, AT>(Ag + pix * p.lda + c0);
        } else {
          ra[i] = i32x4{0, 0, 0, 0};
        }
      }
    } else {
      // generic per-element gather (tiny Cin: the 4-/3-channel stem convs, odd K linears)
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        T tmp[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          const int k = kbase + cc * CE + e;
          float v = 0.f;
          if (rb[i] >= 0 && k < p.K) {
            const int tap = k / p.Cin;
            const int c = k - tap * p.Cin;
            const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
            const int iy = ry[i] + dy, ix = rx[i] + dx;
            if (iy >= 0 && iy < Hup && ix >= 0 && ix < Wup) {
              const size_t pix = ((size_t)rb[i] * p.Hin + (iy >> p.up)) * p.Win + (ix >> p.up);
              v = (float)Ag[pix * p.lda + c];
            }
          }
          tmp[e] = (T)v;
        }
        ra[i] = *reinterpret_cast<i32x4*>(tmp);
      }
    }
    // weights: packed + zero padded to [Npad][Kpad], no bounds checks needed
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int n = n0 + r0 + 32 * i;
      rbv[i] = *reinterpret_cast<const i32x4*>(Wg + (size_t)n * p.Kpad + kbase + cc * CE);
    }
  };
  auto store_tile = [&](int buf) {
    char* a = sA + buf * BM * 128;
    char* b = sB + buf * BN * 128;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<i32x4*>(a + row * 128 + ((cc ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<i32x4*>(b + row * 128 + ((cc ^ (row & 7)) << 4)) = rbv[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kpad / KT;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15;       // fragment row (A: m, B: n) inside a 16x16 tile
  const int fgrp = lane >> 4;       // k-chunk group
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* a = sA + cur * BM * 128;
    const char* b = sB + cur * BN * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + fgrp;
      i32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 16 + frow;
        fa[i] = *reinterpret_cast<const i32x4*>(a + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / 2) + j * 16 + frow;
        fb[j] = *reinterpret_cast<const i32x4*>(b + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 16x16 MFMA: lane holds rows (lane>>4)*4 + r, column lane&15.
  const int col_in_tile = lane & 15;
  const int row_in_tile = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mrow = m0 + wm * (BM / 2) + i * 16 + row_in_tile;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ncol = n0 + wn * (BN / 2) + j * 16 + col_in_tile;   // packed column
      if (p.act == 1 && (j & 1)) continue;                          // gate tiles are consumed with their x tile
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (ncol < p.N) {
        const float bz = p.bias ? p.bias[ncol] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bz;
        if (p.ebias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mrow + r;
            if (m < p.M) v[r] += p.ebias[(size_t)(m / p.rpb) * p.ebias_ld + ncol];
          }
        }
      }
      if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      } else if (p.act == 3) {   // QuickGELU (clip/mod.rs:309-320)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-1.702f * v[r]));
      }
      int nout = ncol;
      if (p.act == 1) {
        // this tile = x columns, tile j+1 = matching gate columns (pack_linear geglu interleave)
        const int gcol = ncol + 16;
        const float gb = p.bias ? p.bias[gcol] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float g = 0.f;
          if constexpr (TN > 1) g = acc[i][(j + 1) % TN][r];
          v[r] = v[r] * gelu_erf(g + gb);
        }
        nout = ((n0 + wn * (BN / 2) + j * 16) >> 1) + col_in_tile;
      }
      const int nlim = p.act == 1 ? (p.N >> 1) : p.N;
      if (nout >= nlim) continue;
      if (p.act == 1 || ncol < p.n_split) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mrow + r;
          if (m < p.M) {
            float o = v[r];
            if (p.R) o += load_as_float<T>(p.R, (size_t)m * p.ldr + nout, p.r_dt);
            store_from_float(p.C, (size_t)m * p.ldc + nout, p.c_dt, o);
          }
        }
      } else {
        // transposed store: Ct[b][n - n_split][key], 4 consecutive keys per lane
        const int nn = ncol - p.n_split;
        const int b = mrow / p.rpb;
        const int key = mrow - b * p.rpb;
        if (mrow + 3 < p.M && key + 3 < p.rpb && ((key | p.ct_ld) & 3) == 0 && p.c_dt == DT_F16) {
          half4 h; h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
          *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.Ct) + ((size_t)b * p.ct_rows + nn) * p.ct_ld + key) = h;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mrow + r;
            if (m < p.M) {
              const int bb = m / p.rpb, kk2 = m - bb * p.rpb;
              store_from_float(p.Ct, ((size_t)bb * p.ct_rows + nn) * p.ct_ld + kk2, p.c_dt, v[r]);
            }
          }
        }
      }
    }
  }
}

template <typename T, typename AT, int BM, int BN>
static void launch_cfg(const IgemmParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, AT, BM, BN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_kernel<T, AT, BM, BN>), dim3(tilesM * tilesN), dim3(256), lds, s, p);
}

template <typename T, typename AT>
static void launch_tiles(const IgemmParams& p, hipStream_t s) {
  // pick the block tile minimising (#waves of the grid over the chip) x (tile cost / relative efficiency)
  struct Cand { int bm, bn; float eff; };
  const Cand cands[4] = {{128, 128, 1.0f}, {128, 64, 0.85f}, {64, 128, 0.85f}, {64, 64, 0.70f}};
  int best = 0; float best_cost = 1e30f;
  for (int c = 0; c < 4; ++c) {
    if (p.act == 1 && cands[c].bn < 64) continue;
    const long tiles = (long)((p.M + cands[c].bm - 1) / cands[c].bm) * ((p.N + cands[c].bn - 1) / cands[c].bn);
    const int slots = 256 * (cands[c].bm * cands[c].bn >= 128 * 128 ? 2 : 4);
    const long waves = (tiles + slots - 1) / slots;
    const float cost = (float)waves * cands[c].bm * cands[c].bn / cands[c].eff;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  switch (best) {
    case 0: launch_cfg<T, AT, 128, 128>(p, s); break;
    case 1: launch_cfg<T, AT, 128, 64>(p, s); break;
    case 2: launch_cfg<T, AT, 64, 128>(p, s); break;
    default: launch_cfg<T, AT, 64, 64>(p, s); break;
  }
}

bool launch_igemm_glds(const IgemmParams& p, int variant, hipStream_t s);   // igemm_glds.hip
bool launch_igemm_f32_pipe(const IgemmParams& p, hipStream_t s);             // igemm_glds.hip (strict-fp32 mode)
bool launch_igemm_hl_pipe(const IgemmParams& p, hipStream_t s);              // igemm_glds.hip (split-operand fp32-class mode)
static std::atomic<int> g_igemm_variant_a{0};   // test hook (sdxl_debug_set "igemm_variant"): -1 generic kernel only, 0 auto, >0 forced tile
void igemm_set_variant(int v) { g_igemm_variant_a = v; }
static std::atomic<int> g_xa_vec64{0};
void igemm_set_xa_vec64(int v) { g_xa_vec64 = v; }
static std::atomic<int> g_igemm_epi_staged{0};
void igemm_set_epilogue_staged(int v) { g_igemm_epi_staged = v; }
static std::atomic<int> g_hl_wexact{1};
void igemm_set_hl_weights_exact(int v) { g_hl_wexact = v; }

void launch_igemm(const IgemmParams& pin, int compute_dt, hipStream_t s) {
  if (pin.M <= 0 || pin.N <= 0) return;
  IgemmParams p = pin;
  p.epi_staged = g_igemm_epi_staged.load();
  p.xa_vec64 = g_xa_vec64.load();
  p.hl_wexact_ok = g_hl_wexact.load();
  const int g_igemm_variant = g_igemm_variant_a.load();
  if (compute_dt == DT_F16 && g_igemm_variant >= 0 && launch_igemm_glds(p, g_igemm_variant, s)) return;
  if (compute_dt == DT_F16 && g_igemm_variant > 0 && launch_igemm_glds(p, 0, s)) return;   // forced tile refused the shape
  if (p.shadow && compute_dt != DT_F16) throw std::runtime_error("an f16 shadow output needs an f16 GEMM (weights-in-registers kernel)");
  if (compute_dt == DT_F32 && g_igemm_variant >= 0 && launch_igemm_f32_pipe(p, s)) return;
  if (compute_dt == DT_HL) {     // no generic twin: layers the HL pipeline cannot take are packed (and launched) as fp32 by the host
    if (launch_igemm_hl_pipe(p, s)) return;
    throw std::runtime_error("split-operand (DT_HL) GEMM: shape / alignment outside the direct-to-LDS pipeline");
  }
  if (p.xa_k) throw std::runtime_error("fused cross-attention needs the f16 direct-to-LDS kernels");
  if (p.shadow) throw std::runtime_error("an f16 shadow output needs the f16 weights-in-registers kernel");
  // the generic kernels below never write the GroupNorm statistics: a caller that was promised them (run_conv tags the output
  // Act and the consumer skips its statistics pass) must not get uninitialised memory -- forced variants, unaligned A, no zero page
  if (p.gn_part) throw std::runtime_error("GroupNorm statistics from the epilogue (gn_part) need the f16 direct-to-LDS 256x128 kernel");
  if (p.ln_stat || p.stat_out)
    throw std::runtime_error("LayerNorm-folded GEMM (ln_stat / stat_out) needs the f16 direct-to-LDS kernels");
  if (compute_dt == DT_F16) {
    if (p.a_dt == DT_F16) launch_tiles<half_t, half_t>(p, s);
    else launch_tiles<half_t, float>(p, s);
  } else {
    launch_tiles<float, float>(p, s);
  }
}

}  // namespace sdxl
