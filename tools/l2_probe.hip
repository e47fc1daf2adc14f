// L2 -> CU bandwidth probe for gfx950: how fast can all 256 CUs pull L2-resident data, and does the path matter?
//   mode 0: global_load_dwordx4 into VGPRs (8 independent loads in flight per lane)
//   mode 1: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction) into a per-wave LDS ring, counted vmcnt
//   pattern 0: every wave instruction reads 1 KiB contiguous;  pattern 1: 8 rows x 128 B, rows `ld` bytes apart (the GEMM's
//   k-tile staging: one 128-byte line of each of 8 tile rows)
// Every workgroup (512 threads, one per CU by LDS size) walks the SAME `ws`-byte window, so after the first touch everything
// hits in the XCD's L2 -- the situation of an implicit-GEMM grid whose tiles share operand panels.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_probe.hip -o /tmp/l2_probe && /tmp/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int PATTERN, int NW = 8, int INFL = 8, int AUX = 0>
__global__ __launch_bounds__(64 * NW) void probe(const char* buf, size_t ws, int ld, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane offset inside one 1-KiB "piece"
  const size_t lane_off = PATTERN == 0 ? (size_t)lane * 16 : (size_t)(lane >> 3) * ld + (size_t)(lane & 7) * 16;
  const unsigned piece_span = PATTERN == 0 ? 1024u : 8u * (unsigned)ld;   // bytes of address space one piece covers
  unsigned npieces = 1;                                                   // power of two: the walk is mask arithmetic only
  while ((size_t)(npieces * 2) * piece_span <= ws) npieces *= 2;          // (a 64-bit modulo per load made the probe ALU bound)
  const unsigned mask = npieces - 1;
  unsigned p = ((unsigned)(blockIdx.x * NW + wave) * 37u) & mask;         // different starting pieces per wave
  i32x4 acc = {0, 0, 0, 0};
  if constexpr (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      i32x4 v[INFL];
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const unsigned q = (p + u * 8) & mask;
        if constexpr (AUX == 2) v[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(buf + (size_t)q * piece_span + lane_off));
        else v[u] = *reinterpret_cast<const i32x4*>(buf + (size_t)q * piece_span + lane_off);
      }
#pragma unroll
      for (int u = 0; u < INFL; ++u) acc ^= v[u];
      p = (p + 8 * INFL) & mask;
    }
  } else {
    char* ring = smem + wave * (INFL * 1024);                             // INFL pieces of 1 KiB per wave
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const unsigned q = (p + u * 8) & mask;
        __builtin_amdgcn_global_load_lds((gptr_t)(buf + (size_t)q * piece_span + lane_off), (lptr_t)(ring + u * 1024), 16, 0, AUX);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");         // previous batch landed, this one in flight
      p = (p + 8 * INFL) & mask;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<const i32x4*>(ring + lane * 16);
  }
  if (acc[0] == 0x7fffffff && acc[1] == 12345) sink[0] = acc[2];        // never true: keeps the loads alive
}

// ---- the implicit GEMM's staging pattern without the math: 128x128 tiles of C[2048, 1280] = A[2048, K] W[1280, K]^T, 160
// workgroups in lockstep (tm-major ids, XCD-contiguous like the kernel), 8 DMA waves, 4 pieces per wave and k-tile into a
// 4-slot ring, counted vmcnt (2 tiles in flight) + one barrier per k-tile.  PFW = 1 adds a 9th wave that touches the lines
// of tile kt + PF (one 4-byte load per line, never waited for) -- does turning the DMA's misses into L2 hits lift the rate?
template <int PFW, int PF, int MAP = 0>
__global__ __launch_bounds__(64 * (8 + PFW)) void gemm_like(const char* A, const char* W, int K, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  { const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  int tm = bid / 10, tn = bid - tm * 10;               // MAP 0: tm-major -- an XCD owns 2 row tiles x all 10 column tiles
  if (MAP == 1) {                                      // MAP 1: 2-D -- XCD (xm, xn) of a 4 x 2 grid owns 4 row tiles x 5 column tiles
    const int xcd = bid / 20, local = bid - xcd * 20;
    tm = (xcd >> 1) * 4 + local / 5; tn = (xcd & 1) * 5 + local % 5;
  }
  if (MAP == 2) {                                      // MAP 2: tn-major -- an XCD owns all 16 row tiles x 1.25 column tiles
    tn = bid / 16; tm = bid - tn * 16;
  }
  const size_t ld = (size_t)K * 2;
  const int nk = K / 64;
  if (wave == 8) {      // prefetch wave: line L of the tile's 128 A rows + 128 W rows
    // fire and forget: the loads land in fixed high registers nothing else uses and are never waited for inside the loop
    const char* q[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int L = h * 64 + lane;
      q[h] = (L < 128 ? A + (size_t)(tm * 128 + L) * ld : W + (size_t)(tn * 128 + L - 128) * ld) + (size_t)PF * 128;
    }
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + PF < nk) {
        asm volatile("global_load_dword v120, %0, off\n\tglobal_load_dword v121, %1, off\n\t"
                     "global_load_dword v122, %2, off\n\tglobal_load_dword v123, %3, off"
                     :: "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]) : "v120", "v121", "v122", "v123", "memory");
#pragma unroll
        for (int h = 0; h < 4; ++h) q[h] += 128;
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int lrow = lane >> 3, slot = lane & 7;
  const char* ap[2]; const char* wp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (j * 8 + wave) * 8 + lrow;
    ap[j] = A + (size_t)(tm * 128 + row) * ld + slot * 16;
    wp[j] = W + (size_t)(tn * 128 + row) * ld + slot * 16;
  }
  auto issue = [&](int kt, int buf) {
    char* la = smem + buf * 32768 + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)(ap[j] + (size_t)kt * 128), (lptr_t)(la + j * 8192), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wp[j] + (size_t)kt * 128), (lptr_t)(la + 16384 + j * 8192), 16, 0, 0);
    }
  };
  for (int s = 0; s < 3; ++s) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 3 < nk) { issue(kt + 3, (kt + 3) & 3); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }   // tile kt landed, 3 in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const i32x4 v = *reinterpret_cast<const i32x4*>(smem + lane * 16);
  if (v[0] == 0x7fffffff && v[1] == 12345) sink[0] = v[2];
}

template <int PFW, int PF, int MAP = 0>
static double run_gemm_like(const char* A, const char* const* Wc, int ncopies, int K, int* sink) {
  const size_t lds = 4 * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_like<PFW, PF, MAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_like<PFW, PF, MAP>), dim3(160), dim3(64 * (8 + PFW)), lds, 0, A, Wc[i % ncopies], K, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_like<PFW, PF, MAP>), dim3(160), dim3(64 * (8 + PFW)), lds, 0, A, Wc[(3 + i) % ncopies], K, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / reps;   // us per launch
}

template <int MODE, int PATTERN, int NW = 8, int INFL = 8, int AUX = 0>
static double run(const char* buf, size_t ws, int ld, int blocks, int iters, int* sink, size_t lds = 96 * 1024) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, PATTERN, NW, INFL, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, PATTERN, NW, INFL, AUX>), dim3(blocks), dim3(64 * NW), lds, 0, buf, ws, ld, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, PATTERN, NW, INFL, AUX>), dim3(blocks), dim3(64 * NW), lds, 0, buf, ws, ld, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * NW * iters * INFL * 1024.0;
  return bytes / (ms * 1e-3) / 1e12;   // TB/s
}

int main() {
  const size_t cap = 64u << 20;
  char* buf; int* sink;
  hipMalloc(&buf, cap + (1 << 20)); hipMalloc(&sink, 64);
  hipMemset(buf, 1, cap + (1 << 20));
  const int iters = 400;
  printf("mode pattern  ws_MB  blocks   TB/s   GB/s/CU\n");
  for (size_t ws : {(size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20})
    for (int blocks : {64, 256})
      for (int mp = 0; mp < 4; ++mp) {
        const int mode = mp >> 1, pat = mp & 1, ld = 2560;
        double t = 0;
        if (mode == 0 && pat == 0) t = run<0, 0>(buf, ws, ld, blocks, iters, sink);
        if (mode == 0 && pat == 1) t = run<0, 1>(buf, ws, ld, blocks, iters, sink);
        if (mode == 1 && pat == 0) t = run<1, 0>(buf, ws, ld, blocks, iters, sink);
        if (mode == 1 && pat == 1) t = run<1, 1>(buf, ws, ld, blocks, iters, sink);
        printf("%s %s %6zu %6d  %6.2f  %7.1f\n", mode ? "dma  " : "vgpr ", pat ? "8x128B " : "1KiB   ", ws >> 20, blocks, t, t * 1e3 / blocks);
      }
  // ---- second table: what moves the per-CU rate?  (1 MB window, 8x128B pattern, 256 CUs; GB/s per CU)
  printf("\nvariant (ws 1 MB, 8x128B rows, 256 blocks)            GB/s/CU\n");
  const size_t w1 = 1 << 20;
  printf("vgpr  4 waves x  8 in flight                          %7.1f\n", run<0, 1, 4, 8>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("vgpr  8 waves x 16 in flight                          %7.1f\n", run<0, 1, 8, 16>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("vgpr 16 waves x  8 in flight                          %7.1f\n", run<0, 1, 16, 8>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("vgpr  8 waves x  8 in flight, nontemporal             %7.1f\n", run<0, 1, 8, 8, 2>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("vgpr  8 waves x  8, two workgroups per CU (512 blocks) %6.1f\n", run<0, 1, 8, 8>(buf, w1, 2560, 512, iters, sink, 64 * 1024) * 1e3 / 256);
  printf("dma   4 waves x  8 in flight                          %7.1f\n", run<1, 1, 4, 8>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("dma   8 waves x 16 in flight                          %7.1f\n", run<1, 1, 8, 16>(buf, w1, 2560, 256, iters, sink, 144 * 1024) * 1e3 / 256);
  printf("dma  16 waves x  8 in flight                          %7.1f\n", run<1, 1, 16, 8>(buf, w1, 2560, 256, iters, sink, 144 * 1024) * 1e3 / 256);
  printf("dma   8 waves x  8 in flight, nt (aux 2)              %7.1f\n", run<1, 1, 8, 8, 2>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  printf("dma   8 waves x  8 in flight, aux 1                   %7.1f\n", run<1, 1, 8, 8, 1>(buf, w1, 2560, 256, iters, sink) * 1e3 / 256);
  // every CU reads its OWN window (no sharing between CUs): 256 x 128 KB
  printf("vgpr  8 waves x  8, per-CU private 32 MB window        %6.1f\n", run<0, 0, 8, 8>(buf, (size_t)32 << 20, 2560, 256, iters, sink) * 1e3 / 256);
  // ---- third table: the GEMM's own staging pattern (FF-out: M 2048, N 1280, K 5120; 410 MB staged per launch)
  {
    const int K = 5120;
    char* A; hipMalloc(&A, (size_t)2048 * K * 2); hipMemset(A, 1, (size_t)2048 * K * 2);
    const int nc = 20;     // 20 x 13 MB of weights: every launch streams a copy that is not cache resident
    std::vector<const char*> Wc(nc);
    for (int i = 0; i < nc; ++i) { char* w; hipMalloc(&w, (size_t)1280 * K * 2); hipMemset(w, 1, (size_t)1280 * K * 2); Wc[i] = w; }
    const double bytes = 160.0 * (K / 64) * 32768.0;
    printf("\nGEMM-like staging, 160 workgroups, K = %d (%.0f MB per launch)        us    GB/s/CU\n", K, bytes / 1e6);
    double t;
    t = run_gemm_like<0, 0>(A, Wc.data(), nc, K, sink); printf("8 DMA waves, cold weights                                   %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<0, 0>(A, Wc.data(), 1, K, sink);  printf("8 DMA waves, same weights every launch (cache resident)      %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<0, 0, 1>(A, Wc.data(), nc, K, sink); printf("2-D XCD map (4 row x 5 col tiles per XCD), cold weights      %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<0, 0, 1>(A, Wc.data(), 1, K, sink);  printf("2-D XCD map, resident weights                                %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<0, 0, 2>(A, Wc.data(), nc, K, sink); printf("tn-major map (16 row x 1.25 col tiles per XCD), cold weights %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<0, 0, 2>(A, Wc.data(), 1, K, sink);  printf("tn-major map, resident weights                               %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<1, 4>(A, Wc.data(), nc, K, sink); printf("+ prefetch wave 4 k-tiles ahead, cold weights               %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<1, 8>(A, Wc.data(), nc, K, sink); printf("+ prefetch wave 8 k-tiles ahead, cold weights               %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
    t = run_gemm_like<1, 16>(A, Wc.data(), nc, K, sink); printf("+ prefetch wave 16 k-tiles ahead, cold weights              %6.1f  %7.1f\n", t, bytes / 160 / t / 1e3);
  }
  return 0;
}
