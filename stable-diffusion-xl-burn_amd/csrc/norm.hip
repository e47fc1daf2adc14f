// GroupNorm(32)+SiLU and LayerNorm for gfx950, NHWC / token-major rows.  HBM-bound kernels: 16-byte vector
// loads, fp32 statistics, one read for the statistics + one read/write for the apply.
//
// Reference semantics (groupnorm/mod.rs:75-82, layernorm/mod.rs:42-49): u = x - mean; y = u / sqrt(mean(u*u) + eps)
// (biased variance, eps inside the sqrt), then *gamma + beta (+ SiLU, silu.rs:14-16).  The reference computes the
// variance from mean-subtracted values (two passes); to keep that numerical behaviour in ONE pass over HBM the
// statistics kernel runs per-channel Welford updates and merges partial (count, mean, M2) triples with Chan's
// formula (row lanes -> channels -> groups -> row splits), never E[x^2]-E[x]^2.
#include "kernels.h"
#include <stdexcept>
#include <type_traits>

namespace sdxl {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename XT> __device__ __forceinline__ void load8(const XT* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<half_t>(const half_t* p, float (&v)[8]) {
  half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
template <typename YT> __device__ __forceinline__ void store8(YT* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<half_t>(half_t* p, const float (&v)[8]) {
  half8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
  // write-through (round 4): the rows go out to the memory side while the block is still streaming instead of staying dirty in this XCD's L2 until the
  // end-of-kernel write-back -- 7 of 8 consumers sit on other XCDs anyway (step 21.13 / 21.09 -> 21.08 / 21.02 ms, profiles/r04_write_through_outputs_ab.txt).
  // The s_nop is the VMEM-store data hazard (> 64 bits of store data, then a VALU write of those registers): the compiler cannot see into the asm.
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(h) : "memory");
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
  *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}

// output in the split-operand format HL16 (kernels.h DT_HL): 4 bytes per logical element like fp32, every 16 channels stored as
// 32 halfs [16 hi | 16 lo] with hi = f16(y), lo = f16(y - hi) -- what the 3-MFMA fp32-class GEMM stages (igemm_common.h)
struct hlout_t { float logical; };
// 8 channels starting at column `col` (multiple of 8) of the row at `row`
template <typename YT> __device__ __forceinline__ void store8r(YT* row, int col, const float (&v)[8]) { store8<YT>(row + col, v); }
template <> __device__ __forceinline__ void store8r<hlout_t>(hlout_t* row, int col, const float (&v)[8]) {
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x = v[j];
    asm("" : "+v"(x));          // one fp32 value for both halves (see store_hl8, igemm_common.h)
    hi[j] = (half_t)x; lo[j] = (half_t)(x - (float)hi[j]);
  }
  half_t* b = reinterpret_cast<half_t*>(row) + ((col >> 4) << 5) + (col & 15);
  *reinterpret_cast<half8*>(b) = hi;
  *reinterpret_cast<half8*>(b + 16) = lo;
}

// Chan merge of (nb, mb, M2b) into (na, ma, M2a)
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& m2a, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  const float n = na + nb;
  const float d = mb - ma;
  const float f = nb / n;
  ma = ma + d * f;
  m2a = m2a + m2b + d * d * na * f;
  na = n;
}

// ---------------------------------------------------------------------------------------------------------
// statistics: grid (nsplit, B), 256 threads.  thread -> (row lane rl, vector column vc); 8 channels per vector.
template <typename XT>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GroupNormParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = p.C, NV = C >> 3;
  const int VPR = NV < 256 ? NV : 256;
  const int RL = 256 / VPR;
  float* chMean = sm;                 // [C]
  float* chM2 = sm + C;               // [C]
  float* tmpMean = sm + 2 * C;        // [RL][VPR*8]
  float* tmpM2 = tmpMean + RL * VPR * 8;
  float* tmpCnt = tmpM2 + RL * VPR * 8;   // [RL]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, s = blockIdx.x;
  const int rps = (p.HW + p.nsplit - 1) / p.nsplit;
  const int row_begin = s * rps;
  const int row_end = min(p.HW, row_begin + rps);
  const int rl = tid / VPR, vl = tid - rl * VPR;
  const XT* X = reinterpret_cast<const XT*>(p.X) + (size_t)b * p.HW * p.ldx;
  const int npass = (NV + VPR - 1) / VPR;
  float total_rows = (float)max(0, row_end - row_begin);
  float amax = 0.f;                   // max |x| of this thread's samples (GroupNormParams::absmax_out)
  const bool want_amax = p.absmax_out != nullptr;      // kernel argument: uniform

  for (int pass = 0; pass < npass; ++pass) {
    const int vc = pass * VPR + vl;
    const bool active = rl < RL && vc < NV;
    // per-thread statistics of 8 channels over its rows: sums of (x - pivot) and (x - pivot)^2 with the pivot = the thread's
    // first sample of that channel (shifted-data algorithm: 3 VALU per element instead of Welford's 6 + a reciprocal, and
    // as robust -- the pivot sits within the data's own spread, so there is no E[x^2] - E[x]^2 cancellation at |mean| >> std);
    // converted to (count, mean, M2) once at the end, merged with Chan's formula from there on
    float mean[8], m2[8], cnt = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = 0.f; m2[j] = 0.f; }
    if (active) {
      float piv[8], s1[8], s2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { piv[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
      int row = row_begin + rl;
      if (row < row_end) load8<XT>(X + (size_t)row * p.ldx + vc * 8, piv);
      // four independent row loads in flight per thread (a one-load-at-a-time loop is latency bound at ~1 TB/s)
      for (; row + 3 * RL < row_end; row += 4 * RL) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<XT>(X + (size_t)(row + u * RL) * p.ldx + vc * 8, v[u]);
        cnt += 4.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = v[u][j] - piv[j];
            s1[j] += d;
            s2[j] = fmaf(d, d, s2[j]);
          }
        if (want_amax) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[u][j]));
        }
      }
      for (; row < row_end; row += RL) {
        float v[8];
        load8<XT>(X + (size_t)row * p.ldx + vc * 8, v);
        cnt += 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[j] - piv[j];
          s1[j] += d;
          s2[j] = fmaf(d, d, s2[j]);
          if (want_amax) amax = fmaxf(amax, fabsf(v[j]));
        }
      }
      if (cnt > 0.f) {
        const float inv = 1.f / cnt;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dm = s1[j] * inv;
          mean[j] = piv[j] + dm;
          m2[j] = fmaxf(s2[j] - s1[j] * dm, 0.f);
        }
      }
    }
    if (rl < RL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        tmpMean[(rl * VPR + vl) * 8 + j] = mean[j];
        tmpM2[(rl * VPR + vl) * 8 + j] = m2[j];
      }
      if (vl == 0) tmpCnt[rl] = cnt;
    }
    __syncthreads();
    if (rl == 0 && vc < NV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float na = 0.f, ma = 0.f, qa = 0.f;
        for (int r = 0; r < RL; ++r)
          chan_merge(na, ma, qa, tmpCnt[r], tmpMean[(r * VPR + vl) * 8 + j], tmpM2[(r * VPR + vl) * 8 + j]);
        chMean[vc * 8 + j] = ma;
        chM2[vc * 8 + j] = qa;
      }
    }
    __syncthreads();
  }
  if (want_amax) {     // block maximum of |x| -> this (entry, row split)'s slot; the slots no row split owns are zeroed by their residue class
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    __shared__ float wmax[4];
    if ((tid & 63) == 0) wmax[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) {
      float* slot = p.absmax_out + (size_t)b * kHlAbsBlocks;
      slot[s] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      for (int k = s + p.nsplit; k < kHlAbsBlocks; k += p.nsplit) slot[k] = 0.f;
    }
  }
  // channels -> groups (every channel of this block has total_rows samples)
  const int cpg = C / p.G;
  if (tid < p.G) {
    float na = 0.f, ma = 0.f, qa = 0.f;
    if (total_rows > 0.f)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) chan_merge(na, ma, qa, total_rows, chMean[c], chM2[c]);
    float* out = p.partial + (((size_t)b * p.G + tid) * p.nsplit + s) * 3;
    out[0] = na; out[1] = ma; out[2] = qa;
  }
}

// finalize: one wavefront per (batch, group): lane s holds row-split partial s (nsplit <= 64), butterfly Chan merge
// -> (mean, rstd) at stat[b][g][0..1], so the apply blocks do not each redo the merge.  (A single thread walking the 64
// partials was a 10 us dependent-load chain per GroupNorm.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GroupNormParams p, float* stat) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= p.B * p.G) return;
  const float* part = p.partial + (size_t)i * p.nsplit * 3;
  float na = 0.f, ma = 0.f, qa = 0.f;
  for (int s = lane; s < p.nsplit; s += 64) chan_merge(na, ma, qa, part[s * 3], part[s * 3 + 1], part[s * 3 + 2]);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float nb = __shfl_xor(na, o), mb = __shfl_xor(ma, o), qb = __shfl_xor(qa, o);
    // symmetric merge (both partners compute the same (n, mean, M2) up to rounding of the identical expression)
    const float n = na + nb;
    if (n > 0.f) {
      const float d = mb - ma;
      const float f = nb / n;
      const float mnew = (lane & o) ? mb + (ma - mb) * (na / n) : ma + d * f;   // each side updates from its own mean
      qa = qa + qb + d * d * na * f;
      ma = mnew;
    }
    na = n;
  }
  if (lane == 0) {
    stat[i * 2] = ma;
    stat[i * 2 + 1] = 1.0f / sqrtf(qa / na + (p.eps_ptr ? *p.eps_ptr : p.eps));
  }
}

// apply: grid (row chunks, B).  thread -> (row lane, fixed vector column): the 8 channels' (mean, rstd*gamma, beta)
// stay in registers while the thread walks its rows, so the inner loop is load / 8 FMA(+SiLU) / store.
// FIN: the (mean, rstd) of this batch entry's groups are merged from the row-split partials in the block's own prologue
// (256 / G threads per group: each Chan-merges its share of the nsplit partials, then a butterfly across the group's
// threads) instead of a separate finalize launch -- one kernel boundary and a 4.7 us launch less per GroupNorm; every apply
// block redoes the 12 KB merge, which the 256 CUs do in parallel.  Needs a power-of-two G <= 256; other group counts keep
// the finalize kernel.
template <typename XT, typename YT, int FIN>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GroupNormParams p, const float* stat_in, int rows_per_block) {
  const int C = p.C, NV = C >> 3;
  const int VPR = NV < 256 ? NV : 256;
  const int RL = 256 / VPR;
  const int tid = threadIdx.x, b = blockIdx.y;
  __shared__ float sstat[512];
  if constexpr (FIN != 0) {
    const int tpg = 256 / p.G;                       // threads per group (power of two)
    const int g = tid / tpg, sub = tid - g * tpg;
    float na = 0.f, ma = 0.f, qa = 0.f;
    if constexpr (FIN == 1) {
      const float* part = p.partial + ((size_t)b * p.G + g) * p.nsplit * 3;
      for (int s = sub; s < p.nsplit; s += tpg) chan_merge(na, ma, qa, part[s * 3], part[s * 3 + 1], part[s * 3 + 2]);
    } else {
      // FIN == 2: the statistics were left by the PRODUCER of x (implicit-GEMM epilogue, IgemmParams::gn_part): per 256-row tile
      // and channel a (mean, M2) pair.  A group's entries are chan_rt row tiles x C/G channels; four loads in flight per thread.
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const int cpg_ = C / p.G, total = p.chan_rt * cpg_;
      const f32x2* cp = reinterpret_cast<const f32x2*>(p.chan_part) + (size_t)b * p.chan_rt * C + g * cpg_;
      const float cnt = (float)p.chan_rows;
      for (int e0 = sub; e0 < total; e0 += 4 * tpg) {
        f32x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + u * tpg;
          const int rt = e / cpg_, cc = e - rt * cpg_;
          v[u] = e < total ? cp[(size_t)rt * C + cc] : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(na, ma, qa, e0 + u * tpg < total ? cnt : 0.f, v[u][0], v[u][1]);
      }
    }
    for (int o = 1; o < tpg; o <<= 1) {              // symmetric merge: every thread of the group ends with the same triple
      const float nb = __shfl_xor(na, o), mb = __shfl_xor(ma, o), qb = __shfl_xor(qa, o);
      const float n = na + nb;
      if (n > 0.f) {
        const float d = mb - ma;
        const float f = nb / n;
        const float mnew = (sub & o) ? mb + (ma - mb) * (na / n) : ma + d * f;
        qa = qa + qb + d * d * na * f;
        ma = mnew;
      }
      na = n;
    }
    if (sub == 0) {
      sstat[g * 2] = ma;
      sstat[g * 2 + 1] = 1.0f / sqrtf(qa / na + (p.eps_ptr ? *p.eps_ptr : p.eps));
    }
    __syncthreads();
  }
  const int rl = tid / VPR, vl = tid - rl * VPR;
  const int cpg = C / p.G;
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(p.HW, row0 + rows_per_block);
  const XT* X = reinterpret_cast<const XT*>(p.X) + (size_t)b * p.HW * p.ldx;
  YT* Y = reinterpret_cast<YT*>(p.Y) + (size_t)b * p.HW * p.ldy;
  const int npass = (NV + VPR - 1) / VPR;
  for (int pass = 0; pass < npass; ++pass) {
    const int vc = pass * VPR + vl;
    if (rl >= RL || vc >= NV) continue;
    float mean[8], scale[8], beta[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = vc * 8 + j;
      const int g = c / cpg;
      const float* st = FIN != 0 ? sstat + g * 2 : stat_in + ((size_t)b * p.G + g) * 2;
      mean[j] = st[0];
      scale[j] = st[1] * p.gamma[c];
      beta[j] = p.beta[c];
    }
    int row = row0 + rl;
    for (; row + RL < row1; row += 2 * RL) {      // two rows in flight per thread
      float v[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) load8<XT>(X + (size_t)(row + u * RL) * p.ldx + vc * 8, v[u]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = (v[u][j] - mean[j]) * scale[j] + beta[j];
          if (p.silu) y = y / (1.0f + __expf(-y));
          v[u][j] = y;
        }
        store8r<YT>(Y + (size_t)(row + u * RL) * p.ldy, vc * 8, v[u]);
      }
    }
    for (; row < row1; row += RL) {
      float v[8];
      load8<XT>(X + (size_t)row * p.ldx + vc * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = (v[j] - mean[j]) * scale[j] + beta[j];
        if (p.silu) y = y / (1.0f + __expf(-y));
        v[j] = y;
      }
      store8r<YT>(Y + (size_t)row * p.ldy, vc * 8, v);
    }
  }
}

static_assert(kGnMaxSplit <= kHlAbsBlocks, "one absmax slot per row split (GroupNormParams::absmax_out)");
int groupnorm_nsplit(int B, int HW, int C) {
  (void)B; (void)C;
  int n = HW / 32;       // >= 32 rows per block; up to kGnMaxSplit row splits x B blocks keep all 256 CUs streaming
  if (n < 1) n = 1;      // (512 splits = 4 workgroups per CU measured no faster: 29.8 vs 29.0 us per GroupNorm; nor do 8-row
                         //  blocks at 32^2, 128 x B blocks instead of 32 x B: GroupNorm class 1.215 vs 1.216 ms per step, r03 g17;
                         //  nor does a one-launch kernel that keeps a whole (entry, group) slab of the 32^2 level in the registers
                         //  of one 512-thread workgroup -- 64 workgroups, statistics + apply: 1.199 vs 1.213 ms per step,
                         //  profiles/r03_gn_onepass_ab.txt; 40..160-byte row segments per group do not coalesce)
  if (n > kGnMaxSplit) n = kGnMaxSplit;
  return n;
}

void launch_groupnorm(const GroupNormParams& pin, hipStream_t s) {
  GroupNormParams p = pin;
  p.nsplit = groupnorm_nsplit(p.B, p.HW, p.C);
  const int NV = p.C >> 3;
  const int VPR = NV < 256 ? NV : 256;
  const int RL = 256 / VPR;
  const size_t lds_stats = (size_t)(2 * p.C + 2 * RL * VPR * 8 + RL) * sizeof(float);
  const bool fin = p.G <= 256 && (p.G & (p.G - 1)) == 0 && 256 / p.G <= 64;   // merge in the apply prologue (see gn_apply_kernel)
  const bool from_producer = p.chan_part != nullptr;     // statistics left by the producing GEMM: no statistics pass at all
  if (from_producer && !(fin && p.C % p.G == 0 && p.chan_rt >= 1 && p.chan_rows >= 1))
    throw std::runtime_error("groupnorm: producer statistics need a power-of-two group count");
  dim3 g1(p.nsplit, p.B);
  if (!from_producer) {
    if (p.x_dt == DT_F16) hipLaunchKernelGGL(gn_stats_kernel<half_t>, g1, dim3(256), lds_stats, s, p);
    else hipLaunchKernelGGL(gn_stats_kernel<float>, g1, dim3(256), lds_stats, s, p);
  }
  // (mean, rstd) per (batch, group) live right after the partials: workspace is [B][G][kGnMaxSplit][3] + [B][G][2] floats
  float* stat = p.partial + (size_t)p.B * p.G * kGnMaxSplit * 3;
  if (!fin) hipLaunchKernelGGL(gn_finalize_kernel, dim3((p.B * p.G + 3) / 4), dim3(256), 0, s, p, stat);
  // apply: aim for >= ~512 blocks, each row lane walking >= 4 rows
  int rows_per_block = (int)(((long)p.B * p.HW + 511) / 512);
  if (rows_per_block < 4 * RL) rows_per_block = 4 * RL;
  if (rows_per_block > p.HW) rows_per_block = p.HW;
  dim3 g2((p.HW + rows_per_block - 1) / rows_per_block, p.B);
#define GN_APPLY(XT, YT)                                                                                              \
  do {                                                                                                                \
    if (from_producer) hipLaunchKernelGGL((gn_apply_kernel<XT, YT, 2>), g2, dim3(256), 0, s, p, stat, rows_per_block); \
    else if (fin) hipLaunchKernelGGL((gn_apply_kernel<XT, YT, 1>), g2, dim3(256), 0, s, p, stat, rows_per_block);     \
    else hipLaunchKernelGGL((gn_apply_kernel<XT, YT, 0>), g2, dim3(256), 0, s, p, stat, rows_per_block);              \
  } while (0)
  if (p.x_dt == DT_F16 && p.y_dt == DT_F16) GN_APPLY(half_t, half_t);
  else if (p.x_dt == DT_F32 && p.y_dt == DT_F16) GN_APPLY(float, half_t);
  else if (p.x_dt == DT_F32 && p.y_dt == DT_F32) GN_APPLY(float, float);
  else if (p.x_dt == DT_F32 && p.y_dt == DT_HL) {
    if ((p.C & 15) != 0 || (p.ldy & 15) != 0) throw std::runtime_error("groupnorm: HL16 output needs C % 16 == 0 rows");
    GN_APPLY(float, hlout_t);
  } else if (p.x_dt == DT_F16 && p.y_dt == DT_F32) GN_APPLY(half_t, float);
  else throw std::runtime_error("groupnorm: unsupported dtype pair");
#undef GN_APPLY
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, two-pass statistics from registers-or-L1 re-reads (rows are <= a few KB)
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const XT* x = reinterpret_cast<const XT*>(p.X) + (size_t)row * p.ldx;
  YT* y = reinterpret_cast<YT*>(p.Y) + (size_t)row * p.ldy;
  const int NV = p.C >> 3;
  float sum = 0.f;
  for (int vc = lane; vc < NV; vc += 64) {
    float v[8];
    load8<XT>(x + vc * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)p.C;
  float sq = 0.f;
  for (int vc = lane; vc < NV; vc += 64) {
    float v[8];
    load8<XT>(x + vc * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float u = v[j] - mean; sq += u * u; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)p.C + (p.eps_ptr ? *p.eps_ptr : p.eps));
  for (int vc = lane; vc < NV; vc += 64) {
    float v[8];
    load8<XT>(x + vc * 8, v);
    f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + vc * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + vc * 8 + 4);
    f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + vc * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + vc * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (v[j] - mean) * rstd * g0[j] + b0[j];
      v[4 + j] = (v[4 + j] - mean) * rstd * g1[j] + b1[j];
    }
    store8r<YT>(y, vc * 8, v);
    if constexpr (std::is_same<YT, half_t>::value) {
      if (p.dup_scale != 0.f) {      // second copy of the ROUNDED values, times a power of two (LayerNormParams::dup_scale)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float h = (float)(half_t)v[j]; v[j] = p.dup_scale > 0.f ? h * p.dup_scale : (v[j] - h) * -p.dup_scale; }
        store8r<YT>(y, p.C + vc * 8, v);
      }
    }
  }
}

// register-cached variant (C <= 8*64*MAXV): the row is read from memory exactly once
template <typename XT, typename YT, int MAXV>
__global__ __launch_bounds__(256) void layernorm_cached_kernel(const LayerNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const XT* x = reinterpret_cast<const XT*>(p.X) + (size_t)row * p.ldx;
  YT* y = reinterpret_cast<YT*>(p.Y) + (size_t)row * p.ldy;
  const int NV = p.C >> 3;
  float v[MAXV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vc = lane + 64 * i;
    if (vc < NV) {
      load8<XT>(x + vc * 8, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)p.C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < NV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float u = v[i][j] - mean; sq += u * u; }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.0f / sqrtf(sq / (float)p.C + (p.eps_ptr ? *p.eps_ptr : p.eps));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vc = lane + 64 * i;
    if (vc < NV) {
      f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + vc * 8), g1 = *reinterpret_cast<const f32x4*>(p.gamma + vc * 8 + 4);
      f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + vc * 8), b1 = *reinterpret_cast<const f32x4*>(p.beta + vc * 8 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (v[i][j] - mean) * rstd * g0[j] + b0[j];
        o[4 + j] = (v[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      store8r<YT>(y, vc * 8, o);
      if constexpr (std::is_same<YT, half_t>::value) {
        if (p.dup_scale != 0.f) {      // second copy of the ROUNDED values, times a power of two (LayerNormParams::dup_scale)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float h = (float)(half_t)o[j]; o[j] = p.dup_scale > 0.f ? h * p.dup_scale : (o[j] - h) * -p.dup_scale; }
          store8r<YT>(y, p.C + vc * 8, o);
        }
      }
    }
  }
}

void launch_layernorm(const LayerNormParams& p, hipStream_t s) {
  dim3 g((p.rows + 3) / 4);
  if (p.dup_scale != 0.f && (p.y_dt != DT_F16 || p.ldy < 2 * p.C)) throw std::runtime_error("layernorm: the duplicated output needs f16 rows of 2 C elements");
  if (p.y_dt == DT_HL) {     // split-operand output (fp32 rows in): the GEMM operand format of the fp32-class mode
    if (p.x_dt != DT_F32 || (p.C & 15) != 0 || (p.ldy & 15) != 0) throw std::runtime_error("layernorm: HL16 output needs fp32 input and C % 16 == 0 rows");
    if (p.C <= 8 * 64 * 3) hipLaunchKernelGGL((layernorm_cached_kernel<float, hlout_t, 3>), g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((layernorm_kernel<float, hlout_t>), g, dim3(256), 0, s, p);
    return;
  }
  if (p.C <= 8 * 64 * 3) {
    if (p.x_dt == DT_F16 && p.y_dt == DT_F16) hipLaunchKernelGGL((layernorm_cached_kernel<half_t, half_t, 3>), g, dim3(256), 0, s, p);
    else if (p.x_dt == DT_F32 && p.y_dt == DT_F16) hipLaunchKernelGGL((layernorm_cached_kernel<float, half_t, 3>), g, dim3(256), 0, s, p);
    else if (p.x_dt == DT_F32 && p.y_dt == DT_F32) hipLaunchKernelGGL((layernorm_cached_kernel<float, float, 3>), g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((layernorm_cached_kernel<half_t, float, 3>), g, dim3(256), 0, s, p);
    return;
  }
  if (p.x_dt == DT_F16 && p.y_dt == DT_F16) hipLaunchKernelGGL((layernorm_kernel<half_t, half_t>), g, dim3(256), 0, s, p);
  else if (p.x_dt == DT_F32 && p.y_dt == DT_F16) hipLaunchKernelGGL((layernorm_kernel<float, half_t>), g, dim3(256), 0, s, p);
  else if (p.x_dt == DT_F32 && p.y_dt == DT_F32) hipLaunchKernelGGL((layernorm_kernel<float, float>), g, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((layernorm_kernel<half_t, float>), g, dim3(256), 0, s, p);
}

}  // namespace sdxl
