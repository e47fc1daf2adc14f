// Flash-style fused attention for gfx950, head dim 64:  O = softmax(Q K^T / sqrt(d) [+ mask]) V
// Replaces Backend::qkv_attention (reference src/backend.rs:4-19, generic body :88-128) for the UNet's 70 self-
// and 70 cross-attention calls per forward; the score matrix is never materialised.
//
// CDNA4 formulation (64-lane wavefronts, 16x16 MFMA, everything between the two GEMMs stays in registers):
//   S^T = K Q^T   : A operand = K tile rows from LDS, B operand = Q fragments held in registers.  The C/D layout
//                   then gives every lane 4 consecutive keys of ONE query (query = lane&15), so
//   softmax       : row max / sum are in-lane reductions + two xor-shuffles (16, 32) across the 4 lane groups;
//   O^T = V^T P^T : the probabilities a lane holds are *already* the B-operand fragment (k index = key); the A
//                   operand is V^T (keys contiguous), which the producing GEMM writes directly (igemm transposed
//                   store), so no transpose and no P round trip through LDS.  O^T's layout keeps query = lane&15,
//                   so the online-softmax rescale is a per-lane scalar.
// Block = 4 wavefronts x 32 queries, 64-key K / V^T tiles double-buffered in LDS (XOR-swizzled 16-byte chunks),
// next tile's global loads issued before the MFMAs, one barrier per tile.  fp32 running max / sum / accumulators.
// T = _Float16 uses v_mfma_f32_16x16x32_f16, T = float uses v_mfma_f32_16x16x4_f32 (strict-parity mode).
#include "kernels.h"
#include <algorithm>
#include <atomic>
#include <stdexcept>

namespace sdxl {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct AMma;
template <> struct AMma<half_t> {
  static __device__ __forceinline__ f32x4 run(i32x4 a, i32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  }
};
template <> struct AMma<float> {
  static __device__ __forceinline__ f32x4 run(i32x4 a, i32x4 b, f32x4 c) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
    return c;
  }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_d64_kernel(const AttnParams p) {
  constexpr int D = 64, KV = 64, MQ = 2;
  constexpr int CE = 16 / sizeof(T);           // elements per 16-byte chunk
  constexpr int RB = 64 * sizeof(T);           // bytes per tile row (K: 64 d, V^T: 64 keys)
  constexpr int CPR = RB / 16;                 // chunks per row: 8 (f16) / 16 (f32)
  constexpr int NKK = D / (4 * CE);            // MFMA k-steps over d: 2 (f16) / 4 (f32)
  constexpr int LI = 64 * CPR / 256;           // chunks per thread per tile: 2 / 4
  constexpr int TILE = 64 * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                 // [2][TILE]
  char* sV = smem + 2 * TILE;      // [2][TILE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, fr = lane & 15;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const T* Qg = reinterpret_cast<const T*>(p.Q) + (size_t)b * p.Nq * p.ldq + h * D;
  const T* Kg = reinterpret_cast<const T*>(p.K) + (size_t)b * p.Nk * p.ldk + h * D;
  const T* Vg = reinterpret_cast<const T*>(p.Vt) + ((size_t)b * p.H + h) * D * p.vt_ld;

  // Q fragments (B operand of S^T): query = q0 + mq*16 + fr, d-chunk = kk*4 + g
  i32x4 qf[MQ][NKK];
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    const int q = q0 + mq * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      if (q < p.Nq) qf[mq][kk] = *reinterpret_cast<const i32x4*>(Qg + (size_t)q * p.ldq + (kk * 4 + g) * CE);
      else qf[mq][kk] = i32x4{0, 0, 0, 0};
    }
  }

  i32x4 rk[LI], rv[LI];
  auto load_tile = [&](int t) {
    const int k0 = t * KV;
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      const int c = tid + i * 256;
      const int row = c / CPR, cc = c - row * CPR;
      const int key = k0 + row;
      if (key < p.Nk) rk[i] = *reinterpret_cast<const i32x4*>(Kg + (size_t)key * p.ldk + cc * CE);
      else rk[i] = i32x4{0, 0, 0, 0};
      rv[i] = *reinterpret_cast<const i32x4*>(Vg + (size_t)row * p.vt_ld + k0 + cc * CE);   // zero padded to vt_ld
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LI; ++i) {
      const int c = tid + i * 256;
      const int row = c / CPR, cc = c - row * CPR;
      const int off = row * RB + ((cc ^ (row & 7)) << 4);
      *reinterpret_cast<i32x4*>(sK + buf * TILE + off) = rk[i];
      *reinterpret_cast<i32x4*>(sV + buf * TILE + off) = rv[i];
    }
  };

  f32x4 ot[4][MQ];
  float mrun[MQ], lrun[MQ];
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    mrun[mq] = -INFINITY; lrun[mq] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt][mq] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float sc = p.scale * 1.44269504088896340736f;   // fold log2(e): p = exp2(s*sc - m)
  const int nt = (p.Nk + KV - 1) / KV;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    const char* kb = sK + cur * TILE;
    const char* vb = sV + cur * TILE;
    // ---- S^T[key][query] = K Q^T
    f32x4 st[4][MQ];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int mq = 0; mq < MQ; ++mq) st[kt][mq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int ch = kk * 4 + g;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int row = kt * 16 + fr;
        const i32x4 kf = *reinterpret_cast<const i32x4*>(kb + row * RB + ((ch ^ (row & 7)) << 4));
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq) st[kt][mq] = AMma<T>::run(kf, qf[mq][kk], st[kt][mq]);
      }
    }
    // ---- online softmax per query (query = lane&15 of m-tile mq; this lane's keys: t*64 + kt*16 + g*4 + r)
#pragma unroll
    for (int mq = 0; mq < MQ; ++mq) {
      const int q = q0 + mq * 16 + fr;
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * KV + kt * 16 + g * 4 + r;
          float s = st[kt][mq][r] * sc;
          if (p.mask && q < p.Nq && key < p.Nk) s += p.mask[(size_t)q * p.ldmask + key] * 1.44269504088896340736f;
          if (key >= p.Nk) s = -INFINITY;
          st[kt][mq][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(mrun[mq], mx);
      const float msub = (mnew == -INFINITY) ? 0.f : mnew;
      const float alpha = exp2f(mrun[mq] - msub);      // mrun = -inf -> 0
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2f(st[kt][mq][r] - msub);
          st[kt][mq][r] = e;
          ps += e;
        }
      lrun[mq] = lrun[mq] * alpha + ps;
      mrun[mq] = mnew;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[dt][mq][r] *= alpha;
    }
    // ---- O^T[d][query] += V^T P^T
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        i32x4 pf[MQ];
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq) {
          half8 hh;
#pragma unroll
          for (int r = 0; r < 4; ++r) { hh[r] = (half_t)st[2 * u][mq][r]; hh[4 + r] = (half_t)st[2 * u + 1][mq][r]; }
          pf[mq] = __builtin_bit_cast(i32x4, hh);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int row = dt * 16 + fr;
          // keys u*32 + g*4 .. +3 (bytes 64u+8g) and u*32+16+g*4 .. +3 (bytes 64u+32+8g)
          const int c1 = 4 * u + (g >> 1), c2 = c1 + 2, sub = (g & 1) * 8;
          const i32x2 v1 = *reinterpret_cast<const i32x2*>(vb + row * RB + ((c1 ^ (row & 7)) << 4) + sub);
          const i32x2 v2 = *reinterpret_cast<const i32x2*>(vb + row * RB + ((c2 ^ (row & 7)) << 4) + sub);
          const i32x4 vf = i32x4{v1[0], v1[1], v2[0], v2[1]};
#pragma unroll
          for (int mq = 0; mq < MQ; ++mq) ot[dt][mq] = AMma<T>::run(vf, pf[mq], ot[dt][mq]);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int row = dt * 16 + fr;
          const int ch = 4 * kt + g;     // keys kt*16 + g*4 .. +3 -> bytes 64kt + 16g
          const i32x4 vf = *reinterpret_cast<const i32x4*>(vb + row * RB + ((ch ^ (row & 7)) << 4));
#pragma unroll
          for (int mq = 0; mq < MQ; ++mq)
            ot[dt][mq] = AMma<T>::run(vf, __builtin_bit_cast(i32x4, st[kt][mq]), ot[dt][mq]);
        }
      }
    }
    if (t + 1 < nt) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[query = lane&15][d = dt*16 + g*4 + r]
  T* Og = reinterpret_cast<T*>(p.O) + (size_t)b * p.Nq * p.ldo + h * D;
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    float l = lrun[mq];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int q = q0 + mq * 16 + fr;
    if (q < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        T* dst = Og + (size_t)q * p.ldo + dt * 16 + g * 4;
        if constexpr (sizeof(T) == 2) {
          half4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(ot[dt][mq][r] * inv);
          *reinterpret_cast<half4*>(dst) = o;
        } else {
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = ot[dt][mq][r] * inv;
          *reinterpret_cast<f32x4*>(dst) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// f16 production variant: same math and register layout as attn_d64_kernel<half>, with the CDNA4 staging path:
//   * K and V^T tiles go HBM/L2 -> LDS by `global_load_lds_dwordx4` (lane-linear image, swizzle on the source address,
//     keys beyond Nk read a zero page), double-buffered, one raw barrier per 64-key tile;
//   * Q fragments are pre-multiplied by scale*log2(e) once, so the per-score work is max / sub / v_exp_f32 / add;
//   * key-tail masking only runs on the last tile (wave-uniform branch); the optional additive mask likewise.
typedef const __attribute__((address_space(1))) void* agptr_t;
typedef __attribute__((address_space(3))) void* alptr_t;

__global__ __launch_bounds__(256, 2) void attn_d64_f16_kernel(const AttnParams p, const void* zeros) {
  constexpr int D = 64, KV = 64, MQ = 2, TILE = 64 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V^T tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fr = lane & 15;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const half_t* Qg = reinterpret_cast<const half_t*>(p.Q) + (size_t)b * p.Nq * p.ldq + h * D;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + (size_t)b * p.Nk * p.ldk + h * D;
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + h) * D * p.vt_ld;
  const float sc = p.scale * 1.44269504088896340736f;

  i32x4 qf[MQ][2];
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    const int q = q0 + mq * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 v;
      if (q < p.Nq) v = *reinterpret_cast<const half8*>(Qg + (size_t)q * p.ldq + (kk * 4 + g) * 8);
      else v = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sc);
      qf[mq][kk] = __builtin_bit_cast(i32x4, v);
    }
  }
  // DMA geometry: wave w stages tile rows [16w, 16w+16) of K and of V^T, two 1-KiB instructions each
  const int lrow = lane >> 3, slot = lane & 7;
  auto stage = [&](int t, int buf) {
    const int k0 = t * KV;
    char* lk = smem + buf * 2 * TILE + wave * 2048;
    char* lv = lk + TILE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave * 16 + j * 8 + lrow;
      const int ch = (slot ^ (row & 7)) * 8;
      const int key = k0 + row;
      const half_t* ks = key < p.Nk ? Kg + (size_t)key * p.ldk + ch : reinterpret_cast<const half_t*>(zeros);
      __builtin_amdgcn_global_load_lds((agptr_t)ks, (alptr_t)(lk + j * 1024), 16, 0, 0);
      const half_t* vs = Vg + (size_t)row * p.vt_ld + k0 + ch;      // V^T rows are zero padded to vt_ld
      __builtin_amdgcn_global_load_lds((agptr_t)vs, (alptr_t)(lv + j * 1024), 16, 0, 0);
    }
  };

  f32x4 ot[4][MQ];
  float mrun[MQ], lrun[MQ];
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    mrun[mq] = -INFINITY; lrun[mq] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt][mq] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nt = (p.Nk + KV - 1) / KV;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < nt) stage(t + 1, cur ^ 1);
    const char* kb = smem + cur * 2 * TILE;
    const char* vb = kb + TILE;
    f32x4 st[4][MQ];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int mq = 0; mq < MQ; ++mq) st[kt][mq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + g;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int row = kt * 16 + fr;
        const i32x4 kf = *reinterpret_cast<const i32x4*>(kb + row * 128 + ((ch ^ (row & 7)) << 4));
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq) st[kt][mq] = AMma<half_t>::run(kf, qf[mq][kk], st[kt][mq]);
      }
    }
    const bool tail = (t == nt - 1) && (p.Nk & 63) != 0;
#pragma unroll
    for (int mq = 0; mq < MQ; ++mq) {
      if (p.mask) {
        const int q = q0 + mq * 16 + fr;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = t * KV + kt * 16 + g * 4 + r;
            if (q < p.Nq && key < p.Nk) st[kt][mq][r] += p.mask[(size_t)q * p.ldmask + key] * 1.44269504088896340736f;
          }
      }
      if (tail) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KV + kt * 16 + g * 4 + r >= p.Nk) st[kt][mq][r] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(st[0][mq][0], st[0][mq][1]), fmaxf(st[0][mq][2], st[0][mq][3]));
#pragma unroll
      for (int kt = 1; kt < 4; ++kt)
        mx = fmaxf(mx, fmaxf(fmaxf(st[kt][mq][0], st[kt][mq][1]), fmaxf(st[kt][mq][2], st[kt][mq][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(mrun[mq], mx);
      const float msub = (mnew == -INFINITY) ? 0.f : mnew;
      const float alpha = __builtin_amdgcn_exp2f(mrun[mq] - msub);
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[kt][mq][r] - msub);
          st[kt][mq][r] = e;
          ps += e;
        }
      lrun[mq] = lrun[mq] * alpha + ps;
      mrun[mq] = mnew;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[dt][mq][r] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      i32x4 pf[MQ];
#pragma unroll
      for (int mq = 0; mq < MQ; ++mq) {
        half8 hh;
#pragma unroll
        for (int r = 0; r < 4; ++r) { hh[r] = (half_t)st[2 * u][mq][r]; hh[4 + r] = (half_t)st[2 * u + 1][mq][r]; }
        pf[mq] = __builtin_bit_cast(i32x4, hh);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int row = dt * 16 + fr;
        const int c1 = 4 * u + (g >> 1), c2 = c1 + 2, sub = (g & 1) * 8;
        const i32x2 v1 = *reinterpret_cast<const i32x2*>(vb + row * 128 + ((c1 ^ (row & 7)) << 4) + sub);
        const i32x2 v2 = *reinterpret_cast<const i32x2*>(vb + row * 128 + ((c2 ^ (row & 7)) << 4) + sub);
        const i32x4 vf = i32x4{v1[0], v1[1], v2[0], v2[1]};
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq) ot[dt][mq] = AMma<half_t>::run(vf, pf[mq], ot[dt][mq]);
      }
    }
  }
  half_t* Og = reinterpret_cast<half_t*>(p.O) + (size_t)b * p.Nq * p.ldo + h * D;
#pragma unroll
  for (int mq = 0; mq < MQ; ++mq) {
    float l = lrun[mq];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int q = q0 + mq * 16 + fr;
    if (q < p.Nq) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)(ot[dt][mq][r] * inv);
        *reinterpret_cast<half4*>(Og + (size_t)q * p.ldo + dt * 16 + g * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// f16 production variant 2 (no additive mask): 32x32x16 MFMA, deferred running max, 3-slot DMA ring, coalesced output.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_f16: a lane owns ONE query (lane&31) and 32 of the tile's 64 keys, its partner
//     lane^32 the other 32 -> the row max is 16 v_max3 in-lane; the cross-lane exchange and the O / l rescale only run when
//     some query's tile max exceeds the running reference by more than 2^THR (wave-uniform vote).  Between such events the
//     reference m is folded into the accumulator init (acc = -m), so the MFMA output is already s - m and the per-score
//     work is one v_exp_f32, one add and half a cvt_pk.  P <= 2^THR fits fp16 with full relative precision; l and O are fp32.
//   * P stays in registers as the B operand of O^T = V^T P^T (key order of a k-step = the two 4-key runs a lane holds).
//   * K / V^T tiles: global_load_lds into a 3-slot ring, counted vmcnt (two tiles in flight), one raw barrier per tile.
//   * O is normalised, parked in LDS per wave and written as whole 128-byte rows.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 1-D grid, XCD-aware: hardware block b runs on XCD b % 8; remap so that every XCD owns a CONTIGUOUS range of logical ids =
// whole heads (all query blocks of a head), i.e. a head's K / V^T (2 x Nk x 128 B) is fetched into ONE XCD's L2 and re-read
// there by its query blocks, instead of every XCD streaming every head from the Infinity Cache.
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// body of variant 2: the block's four waves take the 128 queries [128 qb, 128 qb + 128) of batch entry b, head hd
// PRIO = 2: the softmax / PV phase of a wave runs at raised issue priority (s_setprio), the score MFMAs at base priority -- with three
// waves per SIMD in different phases the exp-heavy phase is the one that must not wait (measured, profiles/r03_attention_block_balance.txt:
// 113.4 -> 111.1 us at 64^2, key split 22.0 -> 21.6 us at 32^2, mixed blocks 101.2 -> 99.8 us; raising the MFMA phase instead: no change)
// wave-level hand-over of an un-normalised (m, l, O) image between the two key halves of a query block that live on DIFFERENT workgroups (XH bodies
// below): nine write-through 16-byte pieces per lane, a ticket per slot; returns false in the wave that arrived first (its partner finishes the
// rows), true -- with (m, l, O) merged in the fixed order half 0, half 1 -- in the wave that arrived second.  No fence on either side: the image
// lines were never in the reader's L2 (a launch starts with the caches acquired, an image is read once), and nobody waits for anybody.
// (First form: plain stores + agent-scope release = an L2 write-back per wave: 33.6 against 21.2 us per launch of the 32^2 self-attention.)
__device__ __forceinline__ bool attn_xhalf_merge(const AttnParams& p, int slot, int kx, int lane, float& m, float& l, f32x16 (&o)[2]) {
  f32x4* img = reinterpret_cast<f32x4*>(p.xws) + ((size_t)slot * 2 + kx) * (9 * 64) + lane;
  {
    const f32x4 ml = f32x4{m, l, 0.f, 0.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(img), "v"(ml) : "memory");
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = f32x4{o[dt][4 * g4], o[dt][4 * g4 + 1], o[dt][4 * g4 + 2], o[dt][4 * g4 + 3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(img + (1 + dt * 4 + g4) * 64), "v"(v) : "memory");
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned tk = 0;
  if (lane == 0) tk = __hip_atomic_fetch_add(p.xcnt + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tk = __builtin_amdgcn_readfirstlane(tk);
  if (tk == 0) return false;
  if (lane == 0) __hip_atomic_store(p.xcnt + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
  const f32x4* oimg = reinterpret_cast<const f32x4*>(p.xws) + ((size_t)slot * 2 + (1 - kx)) * (9 * 64) + lane;
  f32x4 pi[9];
#pragma unroll
  for (int q9 = 0; q9 < 9; ++q9) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pi[q9]) : "v"(oimg + q9 * 64) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pi[0]), "+v"(pi[1]), "+v"(pi[2]), "+v"(pi[3]), "+v"(pi[4]), "+v"(pi[5]), "+v"(pi[6]), "+v"(pi[7]), "+v"(pi[8])::"memory");
  const float mo = pi[0][0], lo = pi[0][1];
  const float m0 = kx == 0 ? m : mo, m1 = kx == 0 ? mo : m;           // operands by HALF index, whoever holds them
  const float mm = fmaxf(m0, m1);
  const float e0 = __builtin_amdgcn_exp2f(m0 - mm), e1 = __builtin_amdgcn_exp2f(m1 - mm);
  const float l0 = kx == 0 ? l : lo, l1 = kx == 0 ? lo : l;
  l = fmaf(l1, e1, l0 * e0);
  m = mm;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float po = pi[1 + dt * 4 + (r >> 2)][r & 3];
      const float o0 = kx == 0 ? o[dt][r] : po, o1 = kx == 0 ? po : o[dt][r];
      o[dt][r] = fmaf(o1, e1, o0 * e0);
    }
  return true;
}

template <int NS, int PRIO = 2>
__device__ __forceinline__ void attn_d64_v2_body(const AttnParams& p, const void* zeros, char* smem, int b, int hd, int qb) {
  constexpr int KV = 64, TILE = 64 * 128;
  constexpr float THR = 8.0f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
  const int q0 = qb * 128 + wave * 32;
  const half_t* Qg = reinterpret_cast<const half_t*>(p.Q) + (size_t)b * p.Nq * p.ldq + hd * 64;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + (size_t)b * p.Nk * p.ldk + hd * 64;
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + hd) * 64 * p.vt_ld;
  const float sc = p.scale * 1.44269504088896340736f;

  // Q fragments (B operand of S^T): query = q0 + fr, d = ks*16 + h*8 .. +7, pre-multiplied by scale*log2(e)
  half8 qf[4];
  {
    const int q = q0 + fr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.Nq) v = *reinterpret_cast<const half8*>(Qg + (size_t)q * p.ldq + ks * 16 + h * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sc);
      qf[ks] = v;
    }
  }
  // DMA geometry: wave w stages tile rows [16w, 16w+16) of K and of V^T, two 1-KiB pieces each.  A lane's source chunk is a
  // loop-invariant 32-bit byte offset from the tile's (wave-uniform) base -- the per-tile address work is one 64-bit add per piece
  // (profiles/r03_attention_pmc.txt: the 64-bit multiply-adds and the per-lane tail select that used to sit here were ~20 VALU
  // instructions of every tile in a kernel whose VALU is the busiest unit)
  const int lrow = lane >> 3, slot = lane & 7;
  unsigned kofs[2], vofs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + j * 8 + lrow;
    kofs[j] = (unsigned)(row * p.ldk + (slot ^ ((row >> 1) & 7)) * 8) * 2u;
    vofs[j] = (unsigned)(row * p.vt_ld + (slot ^ ((row ^ (row >> 3)) & 7)) * 8) * 2u;      // rows zero padded to vt_ld
  }
  auto stage = [&](int t, int buf) {
    const int k0 = t * KV;
    char* lk = smem + buf * 2 * TILE + wave * 2048;
    char* lv = lk + TILE;
    const char* kt = reinterpret_cast<const char*>(Kg + (size_t)k0 * p.ldk);
    const char* vt = reinterpret_cast<const char*>(Vg + k0);
    const char* ks[2] = {kt + kofs[0], kt + kofs[1]};
    if (k0 + KV > p.Nk) {                                        // key tail (wave-uniform, last tile only): rows past Nk read zeros
      asm volatile("" ::: "memory");                             // (keeps the selects out of the full tiles' path)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (k0 + wave * 16 + j * 8 + lrow >= p.Nk) ks[j] = reinterpret_cast<const char*>(zeros);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((agptr_t)ks[j], (alptr_t)(lk + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((agptr_t)(vt + vofs[j]), (alptr_t)(lv + j * 1024), 16, 0, 0);
    }
  };
  // fragment byte offsets inside a tile
  int koff[2], voff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = u * 32 + fr;
    koff[u] = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);          // K chunk of k-step ks: ^ (ks << 5)
    voff[u] = row * 128;                                           // V^T row dt*32 + fr; chunk swizzle below
  }
  const int vsw[2] = {(fr ^ (fr >> 3)) & 7, ((32 + fr) ^ ((32 + fr) >> 3)) & 7};

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = 0.f, l = 0.f;
  // -m as a live 16-register accumulator image: the first MFMA of every score tile takes it as its C operand directly
  // (rebuilding it cost 16 v_mov per tile); it changes only in the rare rescale branch
  f32x16 minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  const int nt = (p.Nk + KV - 1) / KV;
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nt) stage(s0, s0);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    // own pieces of tile t landed; tiles t+1 .. t+NS-2 (4 pieces each) may stay in flight
    if (t + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + NS - 1 < nt) stage(t + NS - 1, cur == 0 ? NS - 1 : cur - 1);       // slot of tile t-1
    const char* kb = smem + cur * 2 * TILE;
    const char* vb = kb + TILE;
    // ---- S^T - m
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    f32x16 sv[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const half8 kf = *reinterpret_cast<const half8*>(kb + (koff[u] ^ (ks << 5)));
        sv[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? minit : sv[u], 0, 0, 0);
      }
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(2);
    if (t == nt - 1 && (p.Nk & 63) != 0) {                      // key tail (wave-uniform branch)
      asm volatile("" ::: "memory");                            // a real branch: as selects this was 32 v_cndmask in EVERY tile
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KV + u * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= p.Nk) sv[u][r] = -INFINITY;
    }
    // (four independent chains: as one chain the 16 v_max3 are each other's operands and the wave issues nothing else meanwhile)
    float lm[4] = {sv[0][0], sv[0][8], sv[1][0], sv[1][8]};
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      lm[0] = fmaxf(lm[0], sv[0][r]); lm[1] = fmaxf(lm[1], sv[0][8 + r]);
      lm[2] = fmaxf(lm[2], sv[1][r]); lm[3] = fmaxf(lm[3], sv[1][8 + r]);
    }
    const float lmax = fmaxf(fmaxf(lm[0], lm[1]), fmaxf(lm[2], lm[3]));
    if (t == 0 || __any(lmax > THR)) {                          // rare after the first tiles; wave-uniform
      const float pm = fmaxf(lmax, __shfl_xor(lmax, 32));
      const float delta = t == 0 ? pm : fmaxf(pm, 0.f);         // never lower the reference
      const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m = t == 0 ? delta : m + delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -m;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[u][r] -= delta;
    }
    // ---- P = exp2(S - m), row sums, fp16 B fragments: k-step s4 = u*2 + hf holds registers 8*hf .. 8*hf+7 of key tile u
    half8 pf[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        half8 hh;
        float ls = 0.f;                       // per-fragment partial sum: four short add chains instead of one of 32
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pe = __builtin_amdgcn_exp2f(sv[u][8 * hf + e]);
          ls += pe;
          hh[e] = (half_t)pe;
        }
        l += ls;
        pf[u * 2 + hf] = hh;
      }
    // ---- O^T += V^T P^T: k-step s4 covers keys base + {4h..4h+3, 8+4h..8+4h+3}, base = u*32 + 16*hf
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int base = (s4 >> 1) * 32 + (s4 & 1) * 16;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int b1 = (base + 4 * h) * 2, b2 = b1 + 16;           // byte offsets of the two 4-key runs in the row
        const i32x2 v1 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b1 >> 4)) ^ vsw[dt]) << 4) + (b1 & 15));
        const i32x2 v2 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b2 >> 4)) ^ vsw[dt]) << 4) + (b2 & 15));
        const i32x4 vf = i32x4{v1[0], v1[1], v2[0], v2[1]};
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, vf), pf[s4], o[dt], 0, 0, 0);
      }
    }
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  // ---- normalise, park per wave in LDS ([query][d] fp16, 16-byte chunks swizzled by query&7), store whole rows
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  __syncthreads();
  char* ob = smem + wave * 4096;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (half_t)(o[dt][g * 4 + r] * inv);
      *reinterpret_cast<half4*>(ob + fr * 128 + (((dt * 4 + g) ^ (fr & 7)) << 4) + 8 * h) = hv;
    }
  half_t* Og = reinterpret_cast<half_t*>(p.O) + (size_t)b * p.Nq * p.ldo + hd * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), piece = lane & 7;
    const i32x4 v = *reinterpret_cast<const i32x4*>(ob + row * 128 + ((piece ^ (row & 7)) << 4));
    const int q = q0 + row;
    if (q < p.Nq) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(Og + (size_t)q * p.ldo + piece * 8), "v"(v) : "memory");
  }
}

template <int NS>
// launch bound 2 waves/SIMD: with a 256-register budget hipcc keeps the MFMA accumulators in VGPRs; at the default
// bound it parks S and O in AGPRs and pays ~240 v_accvgpr_read/write per 64-key tile around the softmax (measured: VALU
// active 1370 cycles per wave-tile, 2.7x the MFMA time)
__global__ __launch_bounds__(256, 2) void attn_d64_v2_kernel(const AttnParams p, const void* zeros) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS][K tile | V^T tile]
  const int nqb = (p.Nq + 127) / 128;
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.H;
  attn_d64_v2_body<NS>(p, zeros, smem, b, bh - b * p.H, qb);
}

// ---------------------------------------------------------------------------------------------------------
// Split-operand variant (SDXL_DTYPE_F32_SPLIT): fp32-class attention on the f16 matrix pipe.  Q and O are fp32; K [B][Nk][C] and
// V^T [B][C][vt_ld] arrive in the HL16 format of the split-operand GEMMs (every 16 logical elements of a row = 16 hi halfs |
// 16 lo halfs, x = hi + lo to ~2^-22: launch_f32_to_hl), so a row of 64 d (or 64 keys) is 256 bytes = 16 chunks of 16 bytes,
// chunk 4g+{0,1} = hi of group g, 4g+{2,3} = lo.  Schedule of variant 2 (S^T = K Q^T, deferred max, P stays in registers as the
// B operand of O^T = V^T P^T), with every product as three MFMAs:  a_hi b_hi + a_hi b_lo + a_lo b_hi  (lo x lo is below fp32
// rounding).  Q is scaled by scale*log2(e) in fp32 and split once per wave; P = exp2(S - m) is split after the exponential, times 2^11:
// with thousands of keys most probabilities are ~1e-4 and their lo halves would fall into the f16 subnormals (measured: 9e-5 rel
// on 4096 keys without the scale, 5e-7 with it).
// Tiles of 64 keys: 16 KiB of K + 16 KiB of V^T per slot, TWO slots (64 KiB -> two blocks per CU), one barrier per tile.
__global__ __launch_bounds__(256, 2) void attn_d64_hl_kernel(const AttnParams p, const void* zeros) {
  constexpr int KV = 64, TILE = 64 * 256;
  constexpr float THR = 4.0f;          // P = exp2(S - m) <= 16
  constexpr float PSC = 2048.0f;       // P travels as P * 2^11 (<= 2^15): the lo halves of small probabilities stay out of the f16 subnormals
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V^T tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
  const int nqb = (p.Nq + 127) / 128;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q0 = qb * 128 + wave * 32;
  const float* Qg = reinterpret_cast<const float*>(p.Q) + (size_t)b * p.Nq * p.ldq + hd * 64;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + ((size_t)b * p.Nk * p.ldk + hd * 64) * 2;          // row stride 2 * ldk halfs
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + hd) * 64 * (size_t)(2 * p.vt_ld);   // row stride 2 * vt_ld halfs
  const float sc = p.scale * 1.44269504088896340736f;

  // Q fragments (B operand of S^T): query = q0 + fr, d = ks*16 + h*8 .. +7, scaled in fp32, then (hi, lo)
  half8 qh[4], ql[4];
  {
    const int q = q0 + fr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
      if (q < p.Nq) {
        if (p.q_dt == DT_HL) {      // the projection wrote q as HL16: x = hi + lo (22 significant bits), scaled and split again below
          const half_t* qp = reinterpret_cast<const half_t*>(p.Q) + (((size_t)b * p.Nq + q) * p.ldq + hd * 64) * 2 + ks * 32 + h * 8;
          const half8 hi = *reinterpret_cast<const half8*>(qp), lo = *reinterpret_cast<const half8*>(qp + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[e] = (float)hi[e] + (float)lo[e]; c[e] = (float)hi[4 + e] + (float)lo[4 + e]; }
        } else {
          const float* qp = Qg + (size_t)q * p.ldq + ks * 16 + h * 8;
          a = *reinterpret_cast<const f32x4*>(qp); c = *reinterpret_cast<const f32x4*>(qp + 4);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x0 = a[e] * sc, x1 = c[e] * sc;
        asm("" : "+v"(x0)); asm("" : "+v"(x1));      // pinned in fp32: hi and lo must come from the SAME rounded product (store_hl8)
        const half_t h0 = (half_t)x0, h1 = (half_t)x1;
        qh[ks][e] = h0; ql[ks][e] = (half_t)(x0 - (float)h0);
        qh[ks][4 + e] = h1; ql[ks][4 + e] = (half_t)(x1 - (float)h1);
      }
    }
  }
  const bool demote = __builtin_amdgcn_readfirstlane(p.demote) != 0;
  if (demote) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) ql[ks][e] = (half_t)0.f;
  }
  // DMA geometry: a 1-KiB piece = 4 rows x 16 chunks (lane -> row lane>>4, slot lane&15); wave w stages rows [16w, 16w+16) of K
  // and of V^T, four pieces each.  LDS slot c of row r holds source chunk c ^ (r & 7): rows 256 bytes apart spread over the banks.
  const int prow = lane >> 4, pslot = lane & 15;
  auto stage = [&](int t, int buf) {
    const int k0 = t * KV;
    char* lk = smem + buf * 2 * TILE + wave * 4096;
    char* lv = lk + TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 16 + j * 4 + prow;
      const int chunk = pslot ^ (row & 7);
      const int key = k0 + row;
      const half_t* ks = key < p.Nk ? Kg + (size_t)key * (2 * p.ldk) + chunk * 8 : reinterpret_cast<const half_t*>(zeros);
      __builtin_amdgcn_global_load_lds((agptr_t)ks, (alptr_t)(lk + j * 1024), 16, 0, 0);
      const half_t* vs = Vg + (size_t)row * (2 * p.vt_ld) + 2 * k0 + chunk * 8;      // rows zero padded to vt_ld (multiple of 64 keys)
      __builtin_amdgcn_global_load_lds((agptr_t)vs, (alptr_t)(lv + j * 1024), 16, 0, 0);
    }
  };
  int rowoff[2], rsw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) { const int row = u * 32 + fr; rowoff[u] = row * 256; rsw[u] = row & 7; }

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = 0.f, l = 0.f;
  f32x16 minit;
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  const int nt = (p.Nk + KV - 1) / KV;
  stage(0, 0);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of tile t landed
    __builtin_amdgcn_s_barrier();                              // everyone's pieces landed; everyone is done with tile t-1
    asm volatile("" ::: "memory");
    if (t + 1 < nt) stage(t + 1, cur ^ 1);
    const char* kb = smem + cur * 2 * TILE;
    const char* vb = kb + TILE;
    // ---- S^T - m: three MFMAs per 16-deep product
    f32x16 sv[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const half8 khi = *reinterpret_cast<const half8*>(kb + rowoff[u] + (((4 * ks + h) ^ rsw[u]) << 4));
        const half8 klo = *reinterpret_cast<const half8*>(kb + rowoff[u] + (((4 * ks + 2 + h) ^ rsw[u]) << 4));
        sv[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qh[ks], ks == 0 ? minit : sv[u], 0, 0, 0);
        sv[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, ql[ks], sv[u], 0, 0, 0);
        sv[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo, qh[ks], sv[u], 0, 0, 0);
      }
    if (t == nt - 1 && (p.Nk & 63) != 0) {                      // key tail (wave-uniform branch)
      asm volatile("" ::: "memory");                            // a real branch: as selects this was 32 v_cndmask in EVERY tile
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KV + u * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= p.Nk) sv[u][r] = -INFINITY;
    }
    // (four independent chains: as one chain the 16 v_max3 are each other's operands and the wave issues nothing else meanwhile)
    float lm[4] = {sv[0][0], sv[0][8], sv[1][0], sv[1][8]};
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      lm[0] = fmaxf(lm[0], sv[0][r]); lm[1] = fmaxf(lm[1], sv[0][8 + r]);
      lm[2] = fmaxf(lm[2], sv[1][r]); lm[3] = fmaxf(lm[3], sv[1][8 + r]);
    }
    const float lmax = fmaxf(fmaxf(lm[0], lm[1]), fmaxf(lm[2], lm[3]));
    if (t == 0 || __any(lmax > THR)) {                          // rare after the first tiles; wave-uniform
      const float pm = fmaxf(lmax, __shfl_xor(lmax, 32));
      const float delta = t == 0 ? pm : fmaxf(pm, 0.f);
      const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m = t == 0 ? delta : m + delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -m;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[u][r] -= delta;
    }
    // ---- P = exp2(S - m) as (hi, lo) B fragments: k-step s4 = u*2 + hf holds registers 8*hf .. 8*hf+7 of key tile u
    half8 ph[4], pl[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pe = __builtin_amdgcn_exp2f(sv[u][8 * hf + e]);
          ls += pe;
          float ps = pe * PSC;
          asm("" : "+v"(ps));
          const half_t hi = (half_t)ps;
          ph[u * 2 + hf][e] = hi;
          pl[u * 2 + hf][e] = (half_t)(ps - (float)hi);
        }
        l += ls;
      }
    if (demote) {                       // (wave-uniform; precision-frontier instrument only)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int e = 0; e < 8; ++e) pl[s4][e] = (half_t)0.f;
    }
    // ---- O^T += V^T P^T: k-step s4 = the 16 keys of HL16 group s4 in the order {4h..4h+3, 8+4h..8+4h+3}
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const char* vr = vb + rowoff[dt] + 8 * h;
        const i32x2 a0 = *reinterpret_cast<const i32x2*>(vr + (((4 * s4 + 0) ^ rsw[dt]) << 4));
        const i32x2 a1 = *reinterpret_cast<const i32x2*>(vr + (((4 * s4 + 1) ^ rsw[dt]) << 4));
        const i32x2 b0 = *reinterpret_cast<const i32x2*>(vr + (((4 * s4 + 2) ^ rsw[dt]) << 4));
        const i32x2 b1 = *reinterpret_cast<const i32x2*>(vr + (((4 * s4 + 3) ^ rsw[dt]) << 4));
        const half8 vhi = __builtin_bit_cast(half8, i32x4{a0[0], a0[1], a1[0], a1[1]});
        const half8 vlo = __builtin_bit_cast(half8, i32x4{b0[0], b0[1], b1[0], b1[1]});
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, ph[s4], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, pl[s4], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, ph[s4], o[dt], 0, 0, 0);
      }
    cur ^= 1;
  }
  // ---- normalise, park per wave in LDS ([query][d] fp32: 16 chunks of 16 bytes per row, swizzled by query & 15), store whole rows
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / (l * PSC);
  __syncthreads();
  char* ob = smem + wave * 8192;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // registers 4g..4g+3 of o[dt] = d 32dt + 8g + 4h + {0..3} of query fr
      const f32x4 v = f32x4{o[dt][g * 4] * inv, o[dt][g * 4 + 1] * inv, o[dt][g * 4 + 2] * inv, o[dt][g * 4 + 3] * inv};
      const int chunk = dt * 8 + g * 2 + h;
      *reinterpret_cast<f32x4*>(ob + fr * 256 + ((chunk ^ (fr & 15)) << 4)) = v;
    }
  if (p.o_dt != DT_F32) {
    // DT_HL: the out-projection's operand format: 8 consecutive d per lane -> hi / lo octets of an HL16 group.  DT_F16 (mixed modes: rows for an f16
    // out-projection): the hi octet alone -- the fp32 result rounded once, no fp32 round trip through memory
    const bool hl = p.o_dt == DT_HL;
    half_t* Oh = reinterpret_cast<half_t*>(p.O) + ((size_t)b * p.Nq * p.ldo + hd * 64) * (hl ? 2 : 1);
    const size_t rstride = hl ? 2 * (size_t)p.ldo : (size_t)p.ldo;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + (lane >> 3), piece = lane & 7;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ob + row * 256 + (((2 * piece) ^ (row & 15)) << 4));
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(ob + row * 256 + (((2 * piece + 1) ^ (row & 15)) << 4));
      half8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (half_t)v0[e]; lo[e] = (half_t)(v0[e] - (float)hi[e]);
        hi[4 + e] = (half_t)v1[e]; lo[4 + e] = (half_t)(v1[e] - (float)hi[4 + e]);
      }
      const int q = q0 + row;
      if (q < p.Nq) {
        half_t* dst = Oh + (size_t)q * rstride + (hl ? (piece >> 1) * 32 + (piece & 1) * 8 : piece * 8);
        *reinterpret_cast<half8*>(dst) = hi;
        if (hl) *reinterpret_cast<half8*>(dst + 16) = lo;
      }
    }
    return;
  }
  float* Og = reinterpret_cast<float*>(p.O) + (size_t)b * p.Nq * p.ldo + hd * 64;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), piece = lane & 15;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ob + row * 256 + ((piece ^ (row & 15)) << 4));
    const int q = q0 + row;
    if (q < p.Nq) *reinterpret_cast<f32x4*>(Og + (size_t)q * p.ldo + piece * 4) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// f16 variant "key split" (variant 6): variant 2 for grids that leave most SIMDs with ONE wave.
//   Self-attention at 32^2 (Nq = Nk = 1024, 40 batch-heads) is 320 blocks of 128 queries = 1280 waves on 1024 SIMDs: a wave
//   alone on its SIMD runs QK^T (MFMA), softmax (VALU, the longer part) and PV (MFMA) strictly in turn, and the SIMDs that
//   carry two waves set the kernel time while the others idle half of it.  Here a block is 64 queries and its four waves are
//   {query sub-tile} x {key half}: every wave takes 32 queries and the 32 keys of its parity out of each 64-key tile, with
//   its own (m, l, O) -- half the work per wave, twice the waves (2.5 per SIMD: softmax of one under the MFMAs of another,
//   worst SIMD 3 half-units instead of 2 whole ones).  The ring, the DMA pieces and the barriers are variant 2's.  At the end
//   the odd-half waves park (m, l, O) in the dead ring and the even-half waves merge:  O = O0 2^(m0-m) + O1 2^(m1-m).
//   Needs Nk % 64 == 0 (no key tail inside a half) -- the launcher sends everything else to variant 2.
//   KP = 4 is the same body one step finer (the small blocks of attn_d64_mix_kernel): a block is ONE 32-query sub-tile and its
//   four waves are key QUARTERS = {tile parity} x {key half}: a wave computes on every other 64-key tile only (it still stages
//   its DMA pieces and takes the barrier of every tile), and wave 0 merges three partners.
#ifdef SDXL_MEASURE
// coarse s_memtime stamps of the key-split body (tools/attn_timeline.py): [workgroup][wave][8] = entry, Q loaded + first tiles issued,
// k-loop done, merge done (key part 0 only), output stores issued; words 6 / 7 = s_memrealtime (100 MHz) at entry / exit
__device__ unsigned* g_attn_tl = nullptr;
void attention_set_timeline(void* buf) {
  unsigned* b = reinterpret_cast<unsigned*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_tl), &b, sizeof(b)) != hipSuccess) throw std::runtime_error("attention: cannot set the timeline buffer");
}
#define ATTN_STAMP(i) do { atl[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define ATTN_DUMP() do { atl[7] = (unsigned)__builtin_amdgcn_s_memrealtime(); if (g_attn_tl && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) g_attn_tl[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i_] = atl[i_]; } } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#define ATTN_DUMP() do { } while (0)
#endif
// KO (measure builds, attn_variant 11 ..): knock-outs that time one resource of the k-loop alone (results are garbage): bit 0 no DMA inside the
// loop, 1 no softmax arithmetic (scores converted as they are), 2 no per-tile barrier, 3 no MFMAs, 4 no LDS fragment reads
// XH (level 2 of attn_d64_mix_kernel): the block takes key HALF kx of its 64 queries -- tiles [kx nt / 2, (kx + 1) nt / 2) -- and its partner block
// the other half.  Each merging wave parks its (m, l, O) image in the workspace slot of (pair, half, query sub-tile), releases at agent scope and draws a
// ticket; the wave that draws the second one acquires, merges   O = O_0 2^(m_0 - m) + O_1 2^(m_1 - m)   in the fixed order half 0, half 1 -- so the
// result does not depend on which block arrived last -- and stores the output rows.  Nobody waits for anybody: no co-residency is assumed.
template <int KP, int PRIO = 2, int KO = 0, bool XH = false>
__device__ __forceinline__ void attn_d64_ks_body(const AttnParams& p, const void* zeros, char* smem, int b, int hd, int qb, int kx = 0, int pair = 0) {
  static_assert(KP == 2 || KP == 4, "key parts per query sub-tile");
  static_assert(!XH || KP == 2, "cross-workgroup key halves: 64-query blocks");
#ifdef SDXL_MEASURE
  unsigned atl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  atl[6] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
  ATTN_STAMP(0);
  constexpr int KV = 64, TILE = 64 * 128, NS = 3;
  constexpr float THR = 8.0f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qs = KP == 2 ? (wave & 1) : 0, kp = KP == 2 ? (wave >> 1) : wave;      // query sub-tile, key part
  const int kh = KP == 2 ? kp : (kp & 1), tp = kp >> 1;                             // key half inside a tile; (KP = 4) tile parity
  const int fr = lane & 31, h = lane >> 5;
  const int q0 = qb * (KP == 2 ? 64 : 32) + qs * 32;
  (void)zeros;
  const half_t* Qg = reinterpret_cast<const half_t*>(p.Q) + (size_t)b * p.Nq * p.ldq + hd * 64;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + (size_t)b * p.Nk * p.ldk + hd * 64;
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + hd) * 64 * p.vt_ld;
  const float sc = p.scale * 1.44269504088896340736f;

  half8 qf[4];
  {
    const int q = q0 + fr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.Nq) v = *reinterpret_cast<const half8*>(Qg + (size_t)q * p.ldq + ks * 16 + h * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sc);
      qf[ks] = v;
    }
  }
  // DMA geometry (as variant 2): wave w stages tile rows [16w, 16w+16) of K and of V^T, two 1-KiB pieces each
  const int lrow = lane >> 3, slot = lane & 7;
  unsigned kofs[2], vofs[2];                   // loop-invariant byte offsets of this lane's source chunks from the tile bases
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + j * 8 + lrow;
    kofs[j] = (unsigned)(row * p.ldk + (slot ^ ((row >> 1) & 7)) * 8) * 2u;
    vofs[j] = (unsigned)(row * p.vt_ld + (slot ^ ((row ^ (row >> 3)) & 7)) * 8) * 2u;
  }
  auto stage = [&](int t, int buf) {
    const int k0 = t * KV;
    char* lk = smem + buf * 2 * TILE + wave * 2048;
    char* lv = lk + TILE;
    const char* kt = reinterpret_cast<const char*>(Kg + (size_t)k0 * p.ldk);
    const char* vt = reinterpret_cast<const char*>(Vg + k0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((agptr_t)(kt + kofs[j]), (alptr_t)(lk + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((agptr_t)(vt + vofs[j]), (alptr_t)(lv + j * 1024), 16, 0, 0);
    }
  };
  // fragment byte offsets inside a tile: K rows kh*32 + fr; V^T rows dt*32 + fr, keys of this wave's half
  const int krow = kh * 32 + fr;
  const int koff = krow * 128 + ((h ^ ((krow >> 1) & 7)) << 4);   // chunk of k-step ks: ^ (ks << 5)
  const int voff[2] = {fr * 128, (32 + fr) * 128};
  const int vsw[2] = {(fr ^ (fr >> 3)) & 7, ((32 + fr) ^ ((32 + fr) >> 3)) & 7};

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = 0.f, l = 0.f;
  f32x16 minit;                                 // -m, the C operand of every score tile's first MFMA (see variant 2)
#pragma unroll
  for (int r = 0; r < 16; ++r) minit[r] = 0.f;
  const int nt = XH ? (p.Nk / KV) >> 1 : p.Nk / KV;      // tiles of this block
  const int tb = XH ? kx * nt : 0;                       // its first tile
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nt) stage(tb + s0, s0);
  ATTN_STAMP(1);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!(KO & 4)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t == 0) ATTN_STAMP(2);
    if constexpr (!(KO & 1)) { if (t + NS - 1 < nt) stage(tb + t + NS - 1, cur == 0 ? NS - 1 : cur - 1); }
    const char* kb = smem + cur * 2 * TILE;
    const char* vb = kb + TILE;
    cur = cur == NS - 1 ? 0 : cur + 1;
    if (KP == 4 && (t & 1) != tp) continue;                    // the other parity's tile (wave-uniform)
    const bool first = KP == 2 ? t == 0 : t == tp;             // this wave's first tile sets the reference
    // ---- S^T - m for this wave's 32 keys
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
    f32x16 sv;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 kf;
      if constexpr (KO & 16) kf = qf[ks]; else kf = *reinterpret_cast<const half8*>(kb + (koff ^ (ks << 5)));
      if constexpr (KO & 8) { if (ks == 0) sv = minit; asm volatile("" ::"v"(kf), "v"(qf[ks])); }
      else sv = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? minit : sv, 0, 0, 0);
    }
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(2);
    float lm[2] = {sv[0], sv[8]};                 // (two independent chains, see variant 2)
#pragma unroll
    for (int r = 1; r < 8; ++r) { lm[0] = fmaxf(lm[0], sv[r]); lm[1] = fmaxf(lm[1], sv[8 + r]); }
    const float lmax = (KO & 2) ? 0.f : fmaxf(lm[0], lm[1]);
    if (!(KO & 2) && (first || __any(lmax > THR))) {
      const float pm = fmaxf(lmax, __shfl_xor(lmax, 32));
      const float delta = first ? pm : fmaxf(pm, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m = first ? delta : m + delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) minit[r] = -m;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] -= delta;
    }
    // ---- P = exp2(S - m): k-step hf holds registers 8*hf .. 8*hf+7
    half8 pf[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      half8 hh;
      float ls = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pe = (KO & 2) ? sv[8 * hf + e] : __builtin_amdgcn_exp2f(sv[8 * hf + e]);
        if constexpr (!(KO & 2)) ls += pe;
        hh[e] = (half_t)pe;
      }
      l += ls;
      pf[hf] = hh;
    }
    // ---- O^T += V^T P^T: k-step hf covers keys kh*32 + 16*hf + {4h..4h+3, 8+4h..8+4h+3}
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int base = kh * 32 + hf * 16;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int b1 = (base + 4 * h) * 2, b2 = b1 + 16;
        i32x2 v1, v2;
        if constexpr (KO & 16) { v1 = i32x2{b1, b2}; v2 = i32x2{b2, b1}; }
        else {
          v1 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b1 >> 4)) ^ vsw[dt]) << 4) + (b1 & 15));
          v2 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b2 >> 4)) ^ vsw[dt]) << 4) + (b2 & 15));
        }
        const i32x4 vf = i32x4{v1[0], v1[1], v2[0], v2[1]};
        if constexpr (KO & 8) asm volatile("" ::"v"(vf), "v"(pf[hf]));
        else o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, vf), pf[hf], o[dt], 0, 0, 0);
      }
    }
  }
  ATTN_STAMP(3);
  l += __shfl_xor(l, 32);
  if (KP == 4 && tp >= nt) m = -INFINITY;       // (a wave that saw no tile of its parity: weight 0 in the merge; the launcher asks Nk >= 128)
  // ---- merge the key parts of each query sub-tile through the dead ring: parked images of [34][64 lanes] floats at smem + 0
  //      (KP = 2: one per query sub-tile; KP = 4: the three partners of wave 0)
  constexpr int NPARK = KP == 2 ? 2 : 3, NMERGE = KP == 2 ? 1 : 3;
  __syncthreads();
  float* mbase = reinterpret_cast<float*>(smem) + lane;
  if (kp != 0) {
    float* mb = mbase + (KP == 2 ? qs : kp - 1) * (34 * 64);
    mb[0] = m; mb[64] = l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mb[(2 + dt * 16 + r) * 64] = o[dt][r];
  }
  __syncthreads();
  if (kp != 0) { ATTN_DUMP(); return; }
  {
    const float* mb = mbase + (KP == 2 ? qs : 0) * (34 * 64);
    float mm = m;
#pragma unroll
    for (int j = 0; j < NMERGE; ++j) mm = fmaxf(mm, mb[j * (34 * 64)]);
    const float a0 = __builtin_amdgcn_exp2f(m - mm);
    m = mm;
    l *= a0;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= a0;
#pragma unroll
    for (int j = 0; j < NMERGE; ++j) {
      const float* mj = mb + j * (34 * 64);
      const float aj = __builtin_amdgcn_exp2f(mj[0] - mm);
      l += mj[64] * aj;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] += mj[(2 + dt * 16 + r) * 64] * aj;
    }
  }
  ATTN_STAMP(4);
  if constexpr (XH) {
    if (!attn_xhalf_merge(p, pair * 2 + qs, kx, lane, m, l, o)) { ATTN_DUMP(); return; }
  }
  const float inv = 1.0f / l;
  char* ob = smem + ((NPARK * 8704 + 4095) & ~4095) + qs * 4096;          // behind the parked images (8704 B each)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (half_t)(o[dt][g * 4 + r] * inv);
      *reinterpret_cast<half4*>(ob + fr * 128 + (((dt * 4 + g) ^ (fr & 7)) << 4) + 8 * h) = hv;
    }
  half_t* Og = reinterpret_cast<half_t*>(p.O) + (size_t)b * p.Nq * p.ldo + hd * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), piece = lane & 7;
    const i32x4 v = *reinterpret_cast<const i32x4*>(ob + row * 128 + ((piece ^ (row & 7)) << 4));
    const int q = q0 + row;
    if (q < p.Nq) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(Og + (size_t)q * p.ldo + piece * 8), "v"(v) : "memory");
  }
  ATTN_STAMP(5);
  ATTN_DUMP();
}

template <int PRIO = 2, int KO = 0>
__global__ __launch_bounds__(256, 2) void attn_d64_ks_kernel(const AttnParams p, const void* zeros) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][K tile | V^T tile]
  const int nqb = (p.Nq + 63) / 64;
  const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.H;
  attn_d64_ks_body<2, PRIO, KO>(p, zeros, smem, b, bh - b * p.H, qb);
}

// ---------------------------------------------------------------------------------------------------------
// f16 "mixed block sizes" (variants 7 / 8): one launch of LARGE blocks followed by SMALL blocks of half their queries.
//   Self-attention of the CFG pair is 640 equal blocks at both levels (64^2: 128-query blocks of variant 2; 32^2: 64-query
//   blocks of the key-split variant) on 256 CUs with room for three blocks each: half the CUs carry three blocks, the other half
//   two, and the launch lasts as long as the CUs with three.  Here the heads of a batch entry are divided 4 : 1 -- the first
//   4/5 of the heads run in large blocks, the rest in small ones (half the queries per block, each wave half the work) -- so
//   that the pair is 512 large + 256 small blocks = two large and one small per CU, 2.5 units of work on every SIMD.
//     LEVEL 0: large = variant 2 (128 queries), small = key split (64 queries, waves = 2 query sub-tiles x 2 key halves)
//     LEVEL 1: large = key split (64 queries),  small = key quarters (32 queries, waves = 4 key parts)
//   Large blocks come first in every XCD's dispatch order; every XCD owns contiguous ranges of both kinds (whole heads where
//   the counts allow).  Which kernel body a (head, query) runs through depends on the head index and the shape of ONE batch
//   entry only, so an entry comes out bit-identical alone, in the CFG pair or in a larger batch.
template <int LEVEL, int PRIO = 2>
__global__ __launch_bounds__(256, 2) void attn_d64_mix_kernel(const AttnParams p, const void* zeros, int big_heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][K tile | V^T tile]
  constexpr int QL = LEVEL == 0 ? 128 : 64, QS = LEVEL == 2 ? QL : QL / 2;
  // (level 2: a "small" block is one key HALF of a large block's 64 queries -- two of them per query block, consecutive ids.  The same for the
  //  128-query blocks of the 64^2 level -- large = variant 2, small = its key halves -- measured 110.0 - 110.9 against 108.8 - 109.2 us for level 0 and
  //  was not kept: profiles/r04_attention_key_halves_ab.txt)
  const int nql = (p.Nq + QL - 1) / QL, nqs = LEVEL == 2 ? 2 * nql : (p.Nq + QS - 1) / QS;
  const int NL = p.B * big_heads * nql, NSM = p.B * (p.H - big_heads) * nqs;       // large / small blocks of the launch
  int bid = blockIdx.x;
  bool large;
  if (((NL | NSM) & 7) == 0) {            // per XCD: its large blocks first, then its small ones
    const int xcd = bid & 7, idx = bid >> 3, nlx = NL >> 3;
    large = idx < nlx;
    bid = large ? xcd * nlx + idx : xcd * (NSM >> 3) + (idx - nlx);
  } else {
    large = bid < NL;
    if (!large) bid -= NL;
  }
  if (large) {
    const int per = big_heads * nql;
    const int b = bid / per, r = bid - b * per, hd = r / nql, qb = r - hd * nql;
    if constexpr (LEVEL == 0) attn_d64_v2_body<3, PRIO>(p, zeros, smem, b, hd, qb);
    else attn_d64_ks_body<2, PRIO>(p, zeros, smem, b, hd, qb);
  } else {
    const int per = (p.H - big_heads) * nqs;
    const int b = bid / per, r = bid - b * per, hs = r / nqs, qb = r - hs * nqs;
    if constexpr (LEVEL == 0) attn_d64_ks_body<2, PRIO>(p, zeros, smem, b, big_heads + hs, qb);
    else if constexpr (LEVEL == 1) attn_d64_ks_body<4, PRIO>(p, zeros, smem, b, big_heads + hs, qb);
    else attn_d64_ks_body<2, PRIO, 0, true>(p, zeros, smem, b, big_heads + hs, qb >> 1, qb & 1, (b * (p.H - big_heads) + hs) * (nqs >> 1) + (qb >> 1));
  }
}

#ifdef SDXL_MEASURE   // lost its A/B (see the MEASURED note below): built only by `build.py --measure`
// ---------------------------------------------------------------------------------------------------------
// f16 variant 3: variant 2 with the two GEMMs of consecutive tiles software-pipelined inside each wave.
//   iteration t:  [row max of S(t), rare rescale]  ->  { S(t+1) = K(t+1) Q^T  (8 MFMA)  ||  P = exp2(S(t)), row sums, fp16
//   fragments (VALU) }  ->  O^T += V^T(t) P^T (8 MFMA).  The matrix pipe and the VALU are separate: in variant 2 a wave ran
//   QK^T, softmax and PV strictly one after the other.  K runs one tile ahead of V, so K and V^T have separate 3-slot rings
//   (48 KiB together, three blocks per CU): after the barrier of iteration t the wave stages {K(t+3), V(t+2)} -- both slots
//   were last read in iteration t-1 -- and waits with vmcnt(4) for {K(t+1), V(t)} only (two iterations of prefetch).
//   Tile indices past the end are clamped (a few redundant DMA pieces keep the counts uniform).
//   MEASURED (profiles/r01_attention_bench.txt): slower than variant 2 -- the second score tile costs 49 VGPRs (189 vs 140),
//   i.e. two blocks per CU instead of three (155 vs 136 us at 64^2); bounded to 168 VGPRs it spills inside the loop (181 us).
//   Kept as forced variant 4 for the record; variant 2 is what runs.
__global__ __launch_bounds__(256, 2) void attn_d64_v3_kernel(const AttnParams p, const void* zeros) {
  constexpr int KV = 64, TILE = 64 * 128;
  constexpr float THR = 8.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3] K tiles | [3] V^T tiles
  char* sK = smem;
  char* sV = smem + 3 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, h = lane >> 5;
  const int nqb = (p.Nq + 127) / 128;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bh = bid / nqb, qb = bid - bh * nqb;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q0 = qb * 128 + wave * 32;
  const half_t* Qg = reinterpret_cast<const half_t*>(p.Q) + (size_t)b * p.Nq * p.ldq + hd * 64;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + (size_t)b * p.Nk * p.ldk + hd * 64;
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + hd) * 64 * p.vt_ld;
  const float sc = p.scale * 1.44269504088896340736f;
  const int nt = (p.Nk + KV - 1) / KV;

  half8 qf[4];
  {
    const int q = q0 + fr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (q < p.Nq) v = *reinterpret_cast<const half8*>(Qg + (size_t)q * p.ldq + ks * 16 + h * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sc);
      qf[ks] = v;
    }
  }
  const int lrow = lane >> 3, slot = lane & 7;
  auto stage_k = [&](int t) {           // K tile t (clamped) -> K slot t % 3: rows [16w, 16w+16) of this wave, two pieces
    const int tc = t < nt ? t : nt - 1;
    char* lk = sK + (t % 3) * TILE + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave * 16 + j * 8 + lrow;
      const int key = tc * KV + row;
      const half_t* ks = key < p.Nk ? Kg + (size_t)key * p.ldk + (slot ^ ((row >> 1) & 7)) * 8 : reinterpret_cast<const half_t*>(zeros);
      __builtin_amdgcn_global_load_lds((agptr_t)ks, (alptr_t)(lk + j * 1024), 16, 0, 0);
    }
  };
  auto stage_v = [&](int t) {
    const int tc = t < nt ? t : nt - 1;
    char* lv = sV + (t % 3) * TILE + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave * 16 + j * 8 + lrow;
      const half_t* vs = Vg + (size_t)row * p.vt_ld + tc * KV + (slot ^ ((row ^ (row >> 3)) & 7)) * 8;
      __builtin_amdgcn_global_load_lds((agptr_t)vs, (alptr_t)(lv + j * 1024), 16, 0, 0);
    }
  };
  int koff[2], voff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = u * 32 + fr;
    koff[u] = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    voff[u] = row * 128;
  }
  const int vsw[2] = {(fr ^ (fr >> 3)) & 7, ((32 + fr) ^ ((32 + fr) >> 3)) & 7};
  auto qk = [&](const char* kb, float init, f32x16 (&sv)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[u][r] = init;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const half8 kf = *reinterpret_cast<const half8*>(kb + (koff[u] ^ (ks << 5)));
        sv[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sv[u], 0, 0, 0);
      }
  };

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = 0.f, l = 0.f;
  // prologue: K(0) | {K(1), V(0)} | {K(2), V(1)}; wait for K(0) only
  stage_k(0);
  stage_k(1); stage_v(0);
  stage_k(2); stage_v(1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  f32x16 sc_[2];                       // S(t) - m
  qk(sK, 0.f, sc_);
  for (int t = 0; t < nt; ++t) {
    // group t = {K(t+1), V(t)} landed (group t+1 may stay in flight); everyone is done with K(t) and V(t-1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage_k(t + 3); stage_v(t + 2);
    if (t == nt - 1 && (p.Nk & 63) != 0) {                      // key tail (wave-uniform)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * KV + u * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= p.Nk) sc_[u][r] = -INFINITY;
    }
    float lmax = sc_[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) lmax = fmaxf(lmax, sc_[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) lmax = fmaxf(lmax, sc_[1][r]);
    if (t == 0 || __any(lmax > THR)) {
      const float pm = fmaxf(lmax, __shfl_xor(lmax, 32));
      const float delta = t == 0 ? pm : fmaxf(pm, 0.f);
      const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m = t == 0 ? delta : m + delta;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc_[u][r] -= delta;
    }
    // ---- S(t+1) - m on the matrix pipe while the VALU turns S(t) into P
    f32x16 sn[2];
    qk(sK + ((t + 1) % 3) * TILE, -m, sn);                     // past the last tile: a clamped copy, never used
    half8 pf[4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        half8 hh;
        float ls = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pe = __builtin_amdgcn_exp2f(sc_[u][8 * hf + e]);
          ls += pe;
          hh[e] = (half_t)pe;
        }
        l += ls;
        pf[u * 2 + hf] = hh;
      }
    // ---- O^T += V^T(t) P^T
    const char* vb = sV + (t % 3) * TILE;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int base = (s4 >> 1) * 32 + (s4 & 1) * 16;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int b1 = (base + 4 * h) * 2, b2 = b1 + 16;
        const i32x2 v1 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b1 >> 4)) ^ vsw[dt]) << 4) + (b1 & 15));
        const i32x2 v2 = *reinterpret_cast<const i32x2*>(vb + voff[dt] + ((((b2 >> 4)) ^ vsw[dt]) << 4) + (b2 & 15));
        const i32x4 vf = i32x4{v1[0], v1[1], v2[0], v2[1]};
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, vf), pf[s4], o[dt], 0, 0, 0);
      }
    }
    sc_[0] = sn[0]; sc_[1] = sn[1];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // clamped look-ahead pieces still in flight
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  __syncthreads();
  char* ob = smem + wave * 4096;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (half_t)(o[dt][g * 4 + r] * inv);
      *reinterpret_cast<half4*>(ob + fr * 128 + (((dt * 4 + g) ^ (fr & 7)) << 4) + 8 * h) = hv;
    }
  half_t* Og = reinterpret_cast<half_t*>(p.O) + (size_t)b * p.Nq * p.ldo + hd * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), piece = lane & 7;
    const i32x4 v = *reinterpret_cast<const i32x4*>(ob + row * 128 + ((piece ^ (row & 7)) << 4));
    const int q = q0 + row;
    if (q < p.Nq) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(Og + (size_t)q * p.ldo + piece * 8), "v"(v) : "memory");
  }
}

#endif  // SDXL_MEASURE

// ---------------------------------------------------------------------------------------------------------
// Flash attention for ONE wide head: head dim 512 (the VAE mid block, ConvSelfAttentionBlock::forward, reference
// autoencoder/mod.rs:550-586 -> Backend::qkv_attention backend.rs:88-128 with n_head = 1; N = 16 384 at a 128x128 latent).
// Scores are never materialised: the reference's default path would hold N^2 = 268 M of them per image.
//
//   * 64 queries per workgroup, 4 waves x 16 queries; a wave keeps its queries' whole Q^T in registers as the B operand of
//     S^T = K Q^T (v_mfma_f32_16x16x32_f16: 16 d-steps, 64 VGPRs) and its whole O^T = V^T P^T as 32 accumulator tiles
//     (128 VGPRs) -- "split-d" only in the sense that every d-tile is its own accumulator; no cross-wave reduction exists.
//   * 32-key tiles: K tile [32][512] (32 KiB) and V^T tile [512][32] (32 KiB) by global_load_lds into a 2-stage ring
//     (128 KiB), counted vmcnt, two raw barriers per tile.  LDS images are lane-linear, so the bank swizzles sit on the
//     SOURCE address: K chunk ^= row & 15 (the 16-lane service groups of ds_read_b128 then hit 16 distinct 16-byte slots),
//     V^T chunk ^= (row >> 2) & 3 (ds_read_b64 over 16 rows of 64 bytes).
//   * the S^T accumulator layout (lane = query, 4 consecutive keys of each 16-key sub-tile) IS the B-operand layout of the
//     PV product once the key order inside a 32-key step is permuted (keys {4g..4g+3, 16+4g..16+4g+3} per lane group g); the
//     V^T fragment is read with the same permutation (two 8-byte reads), so P never leaves the registers.
//   * fp32 online softmax in the exp2 domain (Q pre-scaled by d^-1/2 log2 e); O / l rescale only when some query's running
//     max moved (wave vote).
template <int D>
__global__ __launch_bounds__(256, 1) void attn_hd_kernel(const AttnParams p, const void* zeros) {
  static_assert(D == 512, "one K row = one 1-KiB DMA instruction");
  constexpr int KV = 32, NDS = D / 32, NDT = D / 16;
  constexpr int KTILE = KV * D * 2, VTILE = D * KV * 2, STAGE = KTILE + VTILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fq = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, hd = bh - b * p.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const half_t* Qg = reinterpret_cast<const half_t*>(p.Q) + (size_t)b * p.Nq * p.ldq + hd * D;
  const half_t* Kg = reinterpret_cast<const half_t*>(p.K) + (size_t)b * p.Nk * p.ldk + hd * D;
  const half_t* Vg = reinterpret_cast<const half_t*>(p.Vt) + ((size_t)b * p.H + hd) * D * p.vt_ld;
  const int ntiles = (p.Nk + KV - 1) / KV;

  // ---- DMA: per tile and wave 8 K rows (one instruction each) + 8 V^T pieces of 16 rows x 64 bytes
  auto stage = [&](int t, int buf) {
    char* ks = smem + buf * STAGE;
    char* vs = ks + KTILE;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = j * 4 + wave;                       // key row of the tile
      const int key = t * KV + row;
      const half_t* src = key < p.Nk ? Kg + (size_t)key * p.ldk + (((lane & 48) | ((lane ^ row) & 15)) << 3)
                                     : reinterpret_cast<const half_t*>(zeros) + ((lane & 15) << 3);
      __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)(ks + row * (D * 2)), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (j * 4 + wave) * 16 + (lane >> 2);    // d row of the tile; V^T rows are zero beyond Nk (vt_ld padding)
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      const half_t* src = Vg + (size_t)r * p.vt_ld + t * KV + c * 8;
      __builtin_amdgcn_global_load_lds((agptr_t)src, (alptr_t)(vs + (j * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  stage(0, 0);

  // ---- Q^T fragments (B operand): lane (query fq, k-chunk g) holds Q[q][32 s + 8 g .. +8], pre-scaled
  const float qs = p.scale * 1.44269504088896340736f;
  half8 qf[NDS];
  {
    const int q = q0 + fq;
    const half_t* qp = Qg + (size_t)(q < p.Nq ? q : 0) * p.ldq + g * 8;
#pragma unroll
    for (int s = 0; s < NDS; ++s) {
      half8 v = *reinterpret_cast<const half8*>(qp + s * 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * qs);
      qf[s] = v;
    }
  }
  f32x4 o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -1e30f, l = 0.f;

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) { stage(t + 1, buf ^ 1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave's pieces of tile t have landed
    asm volatile("" ::: "memory");
    const char* ks = smem + buf * STAGE;
    const char* vs = ks + KTILE;
    // S^T sub-tiles: keys 16 u + fq x the wave's 16 queries
    f32x4 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const char* kr = ks + (u * 16 + fq) * (D * 2);
#pragma unroll
      for (int s = 0; s < NDS; ++s) {
        const int c = s * 4 + g;
        const half8 kf = *reinterpret_cast<const half8*>(kr + (((c & 48) | ((c ^ fq) & 15)) << 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[s], acc, 0, 0, 0);
      }
      st[u] = acc;
    }
    // lane holds S^T[key = 16 u + 4 g + r][query fq]
    float mx = -1e30f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * KV + u * 16 + g * 4 + r;
        if (key >= p.Nk) st[u][r] = -1e30f;
        mx = fmaxf(mx, st[u][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
    half8 pf;
    float rs = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[u][r] - mnew);
        rs += e;
        pf[u * 4 + r] = (half_t)e;
      }
    l = l * alpha + rs;                                   // per-lane partial row sum (reduced over g at the end)
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int i = 0; i < NDT; ++i) o[i] *= alpha;
    }
    // O^T tiles: d rows 16 i + fq, keys in the permuted order {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3}
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
      const int r = i * 16 + fq;
      const int sw = (r >> 2) & 3;
      const char* vr = vs + r * 64;
      const int b0 = g * 8, b1 = 32 + g * 8;
      const i32x2 lo = *reinterpret_cast<const i32x2*>(vr + ((((b0 >> 4) ^ sw) << 4) | (b0 & 15)));
      const i32x2 hi = *reinterpret_cast<const i32x2*>(vr + ((((b1 >> 4) ^ sw) << 4) | (b1 & 15)));
      const i32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, vv), pf, o[i], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // stage `buf` is free for tile t + 2
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  const int q = q0 + fq;
  if (q < p.Nq) {
    half_t* og = reinterpret_cast<half_t*>(p.O) + ((size_t)b * p.Nq + q) * p.ldo + hd * D + g * 4;
#pragma unroll
    for (int i = 0; i < NDT; ++i) {   // lane holds O^T[d = 16 i + 4 g + r][query fq]
      half4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = (half_t)(o[i][r] * inv);
      *reinterpret_cast<half4*>(og + i * 16) = h;
    }
  }
}

// per-DEVICE zero page (key-tail / padded rows of the DMA-staged kernels) and dynamic-LDS attribute flags
constexpr int kMaxDev = 64;
static const void* g_attn_zeros[kMaxDev] = {};
static std::atomic<int> g_attn_variant{0};   // test hook (sdxl_debug_set "attn_variant"): -1 generic kernel only, 0 auto
void attention_set_variant(int v) { g_attn_variant = v; }
static int attn_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) throw std::runtime_error("attention: no current HIP device");
  return d;
}
void attention_init() {
  const int d = attn_device();
  if (g_attn_zeros[d]) return;
  void* z = nullptr;
  if (hipMalloc(&z, 4096) != hipSuccess || hipMemset(z, 0, 4096) != hipSuccess)
    throw std::runtime_error("attention: cannot allocate the zero page");
  g_attn_zeros[d] = z;
}

// cross-workgroup key halves (attn_d64_mix_kernel level 2): 1/5 of the heads run as two half-key blocks per 64 queries, so that the CFG
// pair at the 32^2 level is 512 whole + 256 half blocks = two whole and one half block on every CU (2.5 units of work each) instead of
// three blocks on one half of the CUs and two on the other
static std::atomic<int> g_attn_xsplit{1};
void attention_set_xsplit(int v) { g_attn_xsplit = v; }
static int xsplit_heads(int H) { return H - std::max(1, H * 4 / 5); }
size_t attention_xsplit_counters(int B, int H, int Nq) { return (size_t)B * xsplit_heads(H) * ((Nq + 63) / 64) * 2; }
size_t attention_xsplit_ws_bytes(int B, int H, int Nq) { return attention_xsplit_counters(B, H, Nq) * 2 * (9 * 64 * 4 * sizeof(float)); }

void launch_attention_d64(const AttnParams& p, hipStream_t s) {
  const int dev = attn_device();
  const void* g_attn_zero = g_attn_zeros[dev];
  const int g_attn_variant = sdxl::g_attn_variant.load();
  dim3 grid((p.Nq + 127) / 128, p.B * p.H);
  const bool aligned = ((p.ldq | p.ldk | p.vt_ld | p.ldo) & 7) == 0 &&
                       ((reinterpret_cast<uintptr_t>(p.Q) | reinterpret_cast<uintptr_t>(p.K) |
                         reinterpret_cast<uintptr_t>(p.Vt) | reinterpret_cast<uintptr_t>(p.O)) & 15) == 0;
  if (p.dt == DT_F16 && (g_attn_variant == 0 || (g_attn_variant >= 2 && g_attn_variant <= 17)) && g_attn_zero && aligned && !p.mask) {
    // 3-slot ring = 48 KiB per block -> three blocks per CU: the 640 blocks of the 64^2 level run as ONE round (a 4-slot
    // ring admits two per CU, a second half-empty round: 147 us vs 126 us measured); variant 3 keeps the 4-slot ring for A/B
    const dim3 g1(grid.x * grid.y);
    // key-split kernel (64-query blocks, waves = query sub-tile x key half): where 128-query blocks leave fewer than two
    // waves per SIMD (self-attention at 32^2: 160 blocks per batch entry) and the keys are whole 64-key tiles.
    // The choice must not depend on the batch size: a batch entry has to come out bit-identical whether it runs alone or
    // next to others (tests/test_gpu_fullsize.py, split-CFG chains), so it is made on one batch entry's grid (query blocks x heads).
    const bool ks_ok = (p.Nk % 64) == 0 && p.Nk >= 128;
    const bool ks_pick = ks_ok && (g_attn_variant == 6 || ((g_attn_variant == 0 || g_attn_variant == 9) && (int)grid.x * p.H < 256));
    // mixed block sizes (attn_d64_mix_kernel): the first 4/5 of the heads in 128-query blocks, the rest in 64-query key-split
    // blocks -- two large + one small block per CU for the CFG pair at 64^2 (121 -> 107 us; alone 78 -> 64 us, two pairs 210 ->
    // 218 us: profiles/r03_attention_block_balance.txt).  Like the pick above a function of one batch entry's shape only.
    // Level 1 (64-query + 32-query key-quarter blocks for the 32^2 shapes) measured no gain (23.2 -> 23.3 us) and is not picked.
    // Forced: 7 = level 0, 8 = level 1; 9 = the automatic choice without mixing (A/B).
    int mix = -1;
    if (ks_ok && g_attn_variant == 7) mix = 0;
    else if (ks_ok && g_attn_variant == 8) mix = 1;
    else if (ks_ok && g_attn_variant == 0 && !ks_pick && p.H % 5 == 0) mix = 0;
    // level 2 where the key-split kernel would run: whole 128-key pairs of tiles, a workspace from the caller, heads divisible 4 : 1
    else if (ks_pick && g_attn_variant == 0 && p.xws && p.xcnt && g_attn_xsplit.load() && p.H % 5 == 0 && p.H >= 5 && (p.Nk % 128) == 0 && (p.Nq % 64) == 0) mix = 2;
    if (mix >= 0) {
      const int big_heads = p.H >= 2 ? std::max(1, p.H * 4 / 5) : 0;
      const int ql = mix == 0 ? 128 : 64;
      const int nl = p.B * big_heads * ((p.Nq + ql - 1) / ql);
      const int nsm = mix == 2 ? p.B * (p.H - big_heads) * 2 * ((p.Nq + ql - 1) / ql) : p.B * (p.H - big_heads) * ((p.Nq + ql / 2 - 1) / (ql / 2));
      if (mix == 0) hipLaunchKernelGGL(attn_d64_mix_kernel<0>, dim3(nl + nsm), dim3(256), 3 * 2 * 64 * 128, s, p, g_attn_zero, big_heads);
      else if (mix == 1) hipLaunchKernelGGL(attn_d64_mix_kernel<1>, dim3(nl + nsm), dim3(256), 3 * 2 * 64 * 128, s, p, g_attn_zero, big_heads);
      else hipLaunchKernelGGL(attn_d64_mix_kernel<2>, dim3(nl + nsm), dim3(256), 3 * 2 * 64 * 128, s, p, g_attn_zero, big_heads);
      return;
    }
    if (ks_pick) {
      // (round 4: a software-pipelined form of this body -- QK^T of tile t issued around the softmax of tile t - 1, K and V^T on
      // separate rings -- measured 24.9 vs 22.6 us at the 32^2 level and was removed again: with three blocks per CU the other waves
      // already fill the chain's gaps, and the second live score tile costs registers; profiles/r04_attention_swp_ab.txt)
      hipLaunchKernelGGL(attn_d64_ks_kernel<2>, dim3(((p.Nq + 63) / 64) * p.B * p.H), dim3(256), 3 * 2 * 64 * 128, s, p, g_attn_zero);
      return;
    }
#ifdef SDXL_MEASURE
    if (g_attn_variant == 4) {
      hipLaunchKernelGGL(attn_d64_v3_kernel, g1, dim3(256), 6 * 64 * 128, s, p, g_attn_zero);
      return;
    }
    if (g_attn_variant >= 11 && g_attn_variant <= 17 && ks_ok) {     // knock-out timings of the key-split body (tools/attn_knockout.py)
      const dim3 gk(((p.Nq + 63) / 64) * p.B * p.H);
      constexpr int lds = 3 * 2 * 64 * 128;
      switch (g_attn_variant) {
        case 11: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 1>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        case 12: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 2>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        case 13: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 4>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        case 14: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 8>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        case 15: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 16>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        case 16: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 1 | 4>), gk, dim3(256), lds, s, p, g_attn_zero); break;
        default: hipLaunchKernelGGL((attn_d64_ks_kernel<2, 2 | 8>), gk, dim3(256), lds, s, p, g_attn_zero); break;
      }
      return;
    }
    if (g_attn_variant == 3 && p.Nk > 128) {
      constexpr int NS = 4;
      static bool set[kMaxDev] = {};
      if (!set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_d64_v2_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                NS * 2 * 64 * 128) != hipSuccess) throw std::runtime_error("attention: hipFuncSetAttribute failed");
        set[dev] = true;
      }
      hipLaunchKernelGGL(attn_d64_v2_kernel<NS>, g1, dim3(256), NS * 2 * 64 * 128, s, p, g_attn_zero);
      return;
    }
#endif
    hipLaunchKernelGGL(attn_d64_v2_kernel<3>, g1, dim3(256), 3 * 2 * 64 * 128, s, p, g_attn_zero);
    return;
  }
  if (p.dt == DT_F16 && g_attn_variant >= 0 && g_attn_zero && aligned) {
    hipLaunchKernelGGL(attn_d64_f16_kernel, grid, dim3(256), 4 * 64 * 128, s, p, g_attn_zero);
    return;
  }
  if (p.dt == DT_F16) {
    const size_t lds = 4 * 64 * 128;
    hipLaunchKernelGGL(attn_d64_kernel<half_t>, grid, dim3(256), lds, s, p);
  } else {
    const size_t lds = 4 * 64 * 256;
    static bool set[kMaxDev] = {};
    if (!set[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_d64_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess) throw std::runtime_error("attention: hipFuncSetAttribute failed");
      set[dev] = true;
    }
    hipLaunchKernelGGL(attn_d64_kernel<float>, grid, dim3(256), lds, s, p);
  }
}

// split-operand attention (attn_d64_hl_kernel): Q / O fp32, K / V^T in HL16.  Returns false when the shape / alignment needs the
// fp32 kernel (masked attention, unaligned rows).
bool launch_attention_d64_hl(const AttnParams& p, hipStream_t s) {
  const int dev = attn_device();
  const void* zeros = g_attn_zeros[dev];
  if (!zeros || p.mask) return false;
  if (p.o_dt != DT_F32 && (p.o_dt != DT_HL || (p.ldo & 15) != 0) && (p.o_dt != DT_F16 || (p.ldo & 7) != 0)) return false;
  if (p.q_dt != DT_F32 && (p.q_dt != DT_HL || (p.ldq & 15) != 0)) return false;
  if ((p.ldq & 3) != 0 || (p.ldo & 3) != 0 || (p.ldk & 15) != 0 || (p.vt_ld & 63) != 0 || p.vt_ld < (int)(((p.Nk + 63) / 64) * 64)) return false;
  if (((reinterpret_cast<uintptr_t>(p.Q) | reinterpret_cast<uintptr_t>(p.K) | reinterpret_cast<uintptr_t>(p.Vt) | reinterpret_cast<uintptr_t>(p.O)) & 15) != 0) return false;
  constexpr int lds = 2 * 2 * 64 * 256;
  static bool set[kMaxDev] = {};
  if (!set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_d64_hl_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      throw std::runtime_error("attention: hipFuncSetAttribute failed");
    set[dev] = true;
  }
  hipLaunchKernelGGL(attn_d64_hl_kernel, dim3(((p.Nq + 127) / 128) * p.B * p.H), dim3(256), lds, s, p, zeros);
  return true;
}

// one wide head (d = 512, f16, no mask): see attn_hd_kernel.  Returns false when the shape / alignment needs the unfused path.
bool launch_attention_hd512(const AttnParams& p, hipStream_t s) {
  const int dev = attn_device();
  const void* zero = g_attn_zeros[dev];
  if (!zero || p.dt != DT_F16 || p.mask) return false;
  const bool aligned = ((p.ldq | p.ldk | p.vt_ld | p.ldo) & 7) == 0 &&
                       ((reinterpret_cast<uintptr_t>(p.Q) | reinterpret_cast<uintptr_t>(p.K) |
                         reinterpret_cast<uintptr_t>(p.Vt) | reinterpret_cast<uintptr_t>(p.O)) & 15) == 0;
  if (!aligned || p.vt_ld < ((p.Nk + 31) / 32) * 32) return false;      // V^T rows must exist (zero) up to the last 32-key tile
  constexpr int LDS = 2 * (32 * 512 * 2 + 512 * 32 * 2);
  static bool set[kMaxDev] = {};
  if (!set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_hd_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      throw std::runtime_error("attention: hipFuncSetAttribute failed");
    set[dev] = true;
  }
  hipLaunchKernelGGL(attn_hd_kernel<512>, dim3((p.Nq + 63) / 64, p.B * p.H), dim3(256), LDS, s, p, zero);
  return true;
}

}  // namespace sdxl
