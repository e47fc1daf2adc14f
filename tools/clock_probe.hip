// Shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz constant) around an MFMA-dense and a VALU-only
// loop on every CU.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int mfma) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float v = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();     // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (mfma) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    } else {
      v = v * 1.0001f + 0.5f; v = v * 0.9999f - 0.25f; v = v * 1.0001f + 0.5f; v = v * 0.9999f - 0.25f;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = v;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)s; }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 256 * 3 * 8);
  std::vector<unsigned long long> h(256 * 3);
  for (int mfma = 1; mfma >= 0; --mfma)
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = mfma ? 4000 : 60000;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, iters, mfma);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double cyc = 0, real = 0;
      for (int i = 0; i < 256; ++i) { cyc += h[i * 3]; real += h[i * 3 + 1]; }
      cyc /= 256; real /= 256;
      const double us = real / 100.0;   // 100 MHz
      printf("%s rep %d: %.0f shader cycles in %.1f us (event %.1f us) -> %.0f MHz", mfma ? "MFMA x2 waves/SIMD" : "VALU only", rep, cyc, us, ms * 1e3, cyc / us);
      if (mfma) printf("; %.0f TFLOP/s of 2500", 256.0 * 8 * 4.0 * iters * 32768.0 / (us * 1e-6) / 1e12);
      printf("\n");
    }
  return 0;
}
