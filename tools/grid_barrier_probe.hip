// Grid barrier vs kernel boundary on gfx950: what does one "phase change" cost when NWG co-resident workgroups (one per CU) hand
// data to each other?  Groundwork for a persistent per-transformer-block kernel.
//   persistent:  every workgroup writes `bytes_per_wg` of its own chunk, agent-scope release, arrives at a global counter, spins
//                until all arrived, agent-scope acquire, reads ANOTHER workgroup's chunk (other XCD) and checks it; R rounds.
//   launches:    the same write / read-and-check phases as R dependent kernel launches replayed from one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                       // agent scope by default for global atomics
    long spins = 0;                                                          // bail out instead of hanging the GPU
    // relaxed polls (an acquire load per poll invalidates caches every time: 21 us per barrier), ONE acquire fence at the end
    while (__atomic_load_n(counter, __ATOMIC_RELAXED) < target && ++spins < 20000000L) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void persistent(unsigned* data, size_t words_per_wg, unsigned* counter, int rounds, unsigned* errors) {
  const unsigned wg = blockIdx.x, nwg = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    unsigned* mine = data + (size_t)wg * words_per_wg;
    for (size_t i = threadIdx.x; i < words_per_wg; i += blockDim.x) mine[i] = (unsigned)(r * 1000003u + wg * 131u + (unsigned)i);
    __threadfence();
    grid_barrier(counter, (unsigned)(2 * r + 1) * nwg);
    const unsigned other = (wg + 37u) % nwg;                                 // a workgroup on another XCD
    const unsigned* theirs = data + (size_t)other * words_per_wg;
    for (size_t i = threadIdx.x; i < words_per_wg; i += blockDim.x)
      if (__builtin_nontemporal_load(theirs + i) != (unsigned)(r * 1000003u + other * 131u + (unsigned)i)) ++bad;
    grid_barrier(counter, (unsigned)(2 * r + 2) * nwg);                      // nobody overwrites before everybody has read
  }
  if (bad) atomicAdd(errors, bad);
}
__global__ __launch_bounds__(512) void phase_write(unsigned* data, size_t words_per_wg, int r) {
  const unsigned wg = blockIdx.x;
  unsigned* mine = data + (size_t)wg * words_per_wg;
  for (size_t i = threadIdx.x; i < words_per_wg; i += blockDim.x) mine[i] = (unsigned)(r * 1000003u + wg * 131u + (unsigned)i);
}
__global__ __launch_bounds__(512) void phase_read(const unsigned* data, size_t words_per_wg, int r, unsigned* errors) {
  const unsigned wg = blockIdx.x, nwg = gridDim.x, other = (wg + 37u) % nwg;
  const unsigned* theirs = data + (size_t)other * words_per_wg;
  unsigned bad = 0;
  for (size_t i = threadIdx.x; i < words_per_wg; i += blockDim.x)
    if (theirs[i] != (unsigned)(r * 1000003u + other * 131u + (unsigned)i)) ++bad;
  if (bad) atomicAdd(errors, bad);
}

int main() {
  const int rounds = 200;
  unsigned *data, *counter, *errors;
  hipMalloc(&data, (size_t)256 * (1 << 20)); hipMalloc(&counter, 64); hipMalloc(&errors, 64);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("nwg  KiB/wg | persistent: us per phase pair (write+barrier+read+barrier), errors | graph of launches: us per phase pair, errors\n");
  for (int nwg : {220, 256})
    for (size_t kib : {(size_t)0, (size_t)4, (size_t)24, (size_t)96}) {
      const size_t words = kib * 256;
      float ms1 = 0, ms2 = 0; unsigned e1 = 0, e2 = 0;
      // persistent (LDS 96 KiB per workgroup forces one workgroup per CU -> all co-resident when nwg <= 256)
      hipMemsetAsync(counter, 0, 64, s); hipMemsetAsync(errors, 0, 64, s);
      hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      hipLaunchKernelGGL(persistent, dim3(nwg), dim3(512), 96 * 1024, s, data, words, counter, 3, errors);   // warm-up
      hipMemsetAsync(counter, 0, 64, s);
      hipEventRecord(a, s);
      hipLaunchKernelGGL(persistent, dim3(nwg), dim3(512), 96 * 1024, s, data, words, counter, rounds, errors);
      hipEventRecord(b, s); hipStreamSynchronize(s);
      hipEventElapsedTime(&ms1, a, b); hipMemcpy(&e1, errors, 4, hipMemcpyDeviceToHost);
      // the same phases as dependent launches in one graph
      hipMemsetAsync(errors, 0, 64, s);
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(phase_write, dim3(nwg), dim3(512), 0, s, data, words, r);
        hipLaunchKernelGGL(phase_read, dim3(nwg), dim3(512), 0, s, data, words, r, errors);
      }
      hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, s); hipStreamSynchronize(s);
      hipMemsetAsync(errors, 0, 64, s);
      hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipStreamSynchronize(s);
      hipEventElapsedTime(&ms2, a, b); hipMemcpy(&e2, errors, 4, hipMemcpyDeviceToHost);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
      printf("%3d  %5zu  | %8.2f  %u | %8.2f  %u\n", nwg, kib, ms1 * 1e3 / rounds, e1, ms2 * 1e3 / rounds, e2);
    }
  return 0;
}
