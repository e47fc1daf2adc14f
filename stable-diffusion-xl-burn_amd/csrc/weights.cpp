// Device arenas, weight sources (synthetic / flat buffer) and packing of canonical (reference-layout) parameters
// into the MFMA-friendly device layout: dense weights as [Npad][Kpad] (k contiguous, K order = (tap, cin) for convs,
// zero padded to 128 x 64), GEGLU projections column-interleaved (16 x | 16 gate) so the GEMM epilogue can apply
// x*gelu(gate) in registers, QKV / KV / time-embedding projections concatenated along N.  Also the launch helpers
// shared by the UNet and VAE drivers.
#include "engine.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

namespace sdxl {

// ------------------------------------------------------------------------------------------ arena
DeviceArena::~DeviceArena() { if (base) (void)hipFree(base); }
void DeviceArena::reserve(size_t bytes) {
  if (bytes <= cap && base) return;
  if (base) { SDXL_HIP(hipFree(base)); base = nullptr; }
  SDXL_HIP(hipMalloc((void**)&base, bytes));
  cap = bytes;
}
void* DeviceArena::alloc(size_t bytes) {
  const size_t a = round_up(off, 256);
  off = a + bytes;
  if (off > peak) peak = off;
  if (dry) return (void*)(uintptr_t)(a + 256);   // fake, never dereferenced
  if (off > cap) throw Error("device arena overflow (" + std::to_string(off) + " > " + std::to_string(cap) + ")");
  return base + a;
}

// ------------------------------------------------------------------------------------------ profiler
void Profiler::begin(int cls, double flops, hipStream_t s, int m, int n, int k, int ks, int tag) {
  Rec r; r.cls = cls; r.flops = flops; r.m = m; r.n = n; r.k = k; r.ks = ks; r.tag = tag;
  SDXL_HIP(hipEventCreate(&r.a)); SDXL_HIP(hipEventCreate(&r.b));
  SDXL_HIP(hipEventRecord(r.a, s));
  recs.push_back(r);
}
void Profiler::end(hipStream_t s) { SDXL_HIP(hipEventRecord(recs.back().b, s)); }
void Profiler::collect(float ms[NCLS], int launches[NCLS], double flops[NCLS]) {
  for (int i = 0; i < NCLS; ++i) { ms[i] = 0.f; launches[i] = 0; flops[i] = 0.0; }
  const char* dump = std::getenv("SDXL_PROFILE_DUMP");   // optional per-launch CSV (class, M, N, K, ksize, ms)
  FILE* f = dump ? std::fopen(dump, "w") : nullptr;
  if (f) std::fprintf(f, "class,M,N,K,ksize,ms,tag\n");
  for (Rec& r : recs) {
    SDXL_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    SDXL_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += t; launches[r.cls] += 1; flops[r.cls] += r.flops;
    if (f) std::fprintf(f, "%d,%d,%d,%d,%d,%.4f,%d\n", r.cls, r.m, r.n, r.k, r.ks, t, r.tag);
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  recs.clear();
  if (f) std::fclose(f);
}

// ------------------------------------------------------------------------------------------ sources
void SyntheticSource::fetch(const ParamSpec& s, size_t, float* dst, hipStream_t st) {
  const uint64_t key = fnv1a64(s.name) ^ ((seed & ~kSeedF16Weights) * 0x9E3779B97F4A7C15ull);
  launch_synth_fill(dst, s.numel(), key, s.scale, s.mean, st);
  // kSeedF16Weights: every parameter rounded to IEEE f16 and widened again -- what a burn HalfPrecisionSettings record holds
  // (src/bin/sample/main.rs:37); the per-norm eps is a module constant, not a record entry
  if ((seed & kSeedF16Weights) && s.kind != PK_EPS) launch_round_f16(dst, s.numel(), st);
}
FlatSource::FlatSource(const float* b, const std::vector<ParamSpec>& specs) : base(b) {
  size_t o = 0;
  for (const ParamSpec& p : specs) { offsets.push_back(o); o += p.numel(); }
}
void FlatSource::fetch(const ParamSpec& s, size_t index, float* dst, hipStream_t st) {
  SDXL_HIP(hipMemcpyAsync(dst, base + offsets[index], s.numel() * sizeof(float), hipMemcpyDefault, st));
  SDXL_HIP(hipStreamSynchronize(st));   // pageable host memory: keep the staging copy simple and safe
}

FlatSourceF16::FlatSourceF16(const uint16_t* b, const std::vector<ParamSpec>& specs) : base(b) {
  size_t o = 0;
  for (const ParamSpec& p : specs) { offsets.push_back(o); o += p.numel(); if (p.numel() > stage_numel) stage_numel = p.numel(); }
  SDXL_HIP(hipMalloc(&stage, stage_numel * sizeof(uint16_t)));
}
FlatSourceF16::~FlatSourceF16() { if (stage) (void)hipFree(stage); }
static float f16_bits_to_float(uint16_t h) {   // exact widening on the host (subnormals included)
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) bits = sign;
    else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); }
  } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
  else bits = sign | ((e + 112u) << 23) | (m << 13);
  float f; std::memcpy(&f, &bits, 4);
  return f;
}
void FlatSourceF16::fetch(const ParamSpec& s, size_t index, float* dst, hipStream_t st) {
  if (s.kind == PK_EPS) {
    // an f16 stream cannot carry the norm eps: 1e-5 is subnormal in f16 (-> 1.0014e-5) and anything below ~3e-8 becomes 0.  A
    // burn record never holds it (module constant: the .mpk path gets the Config default 1e-5), so the f16 image of the default
    // is taken for the default itself; any other value is widened exactly, on the host (no device denormal mode involved)
    uint16_t h = 0;
    SDXL_HIP(hipMemcpyAsync(&h, base + offsets[index], sizeof(h), hipMemcpyDefault, st));
    SDXL_HIP(hipStreamSynchronize(st));
    // (likewise 1e-6, the VAE GroupNorm eps python/save.py writes: 0x0011 is its f16 image, 1.013e-6 when widened)
    const float v = h == 0x00A8u ? 1e-5f : (h == 0x0011u ? 1e-6f : f16_bits_to_float(h));
    SDXL_HIP(hipMemcpyAsync(dst, &v, sizeof(float), hipMemcpyHostToDevice, st));
    SDXL_HIP(hipStreamSynchronize(st));
    return;
  }
  SDXL_HIP(hipMemcpyAsync(stage, base + offsets[index], s.numel() * sizeof(uint16_t), hipMemcpyDefault, st));
  SDXL_HIP(hipStreamSynchronize(st));   // pageable host memory
  launch_copy_rows(stage, DT_F16, 1, dst, DT_F32, 1, (int)s.numel(), 1, st);   // exact widening: every f16 is an fp32
}

// ------------------------------------------------------------------------------------------ builder
WeightBuilder::WeightBuilder(const std::vector<ParamSpec>& sp, WeightSource& s, DeviceArena& a, int dtype, hipStream_t stream)
    : specs(sp), src(s), arena(a), dt(dtype), st(stream) {
  for (size_t i = 0; i < specs.size(); ++i) {
    index[specs[i].name] = i;
    if (specs[i].numel() > tmp_numel) tmp_numel = specs[i].numel();
  }
  SDXL_HIP(hipMalloc((void**)&tmp, tmp_numel * sizeof(float)));
}
WeightBuilder::~WeightBuilder() { if (tmp) (void)hipFree(tmp); if (tmp2) (void)hipFree(tmp2); }

size_t WeightBuilder::arena_bound(const std::vector<ParamSpec>& specs, int dt, bool wfrag) {
  size_t total = 1 << 20;
  for (const ParamSpec& p : specs) {
    if (p.kind == PK_LINEAR_W) total += round_up(p.shape[1], 128) * round_up(p.shape[0], 64) * dt_size(dt) + 768;   // (+ the DT_HL scale scalar)
    else if (p.kind == PK_CONV_W)
      total += round_up(p.shape[0], 128) * round_up((size_t)p.shape[1] * p.shape[2] * p.shape[3], 64) * dt_size(dt) + 768;   // (+ the DT_HL scale scalar)
    else total += round_up(p.numel(), 128) * sizeof(float) + 256;
    if (p.kind == PK_LINEAR_W || p.kind == PK_CONV_W) total += round_up(p.kind == PK_LINEAR_W ? p.shape[1] : p.shape[0], 128) * 4 + 256;
    if (p.kind == PK_LINEAR_W) total += round_up(p.shape[1], 128) * 4 + 256;   // column sums of LayerNorm-folded projections
    // fragment-order image of the f16 linear / 1x1 weights (attach_wfrag)
    if (wfrag && dt == DT_F16 && p.kind == PK_LINEAR_W) total += round_up(p.shape[1], 128) * round_up(p.shape[0], 64) * 2 + 256;
    if (wfrag && dt == DT_F16 && p.kind == PK_CONV_W && p.shape[2] == 1) total += round_up(p.shape[0], 128) * round_up((size_t)p.shape[1], 64) * 2 + 256;
  }
  return total;
}
const ParamSpec& WeightBuilder::spec(const std::string& name, size_t* idx) const {
  auto it = index.find(name);
  if (it == index.end()) throw Error("unknown parameter '" + name + "'");
  if (idx) *idx = it->second;
  return specs[it->second];
}
const float* WeightBuilder::fetch(const std::string& name) {
  // (the split-operand packing asks for a weight three times in a row -- absmax, exactness flag, packer: the staging buffer still holds
  //  it, and a host-backed source would otherwise pay three H2D copies of up to 26 MB per layer)
  if (name == last_fetched) return tmp;
  size_t i; const ParamSpec& s = spec(name, &i);
  src.fetch(s, i, tmp, st);
  last_fetched = name;
  return tmp;
}
// Split-operand packing (DT_HL): hi = f16(w * 2^e), lo = f16(w * 2^e - hi).  With 2^e * max|w| in [2^13, 2^14) both halves of every
// weight down to 2^-11 of the largest stay f16-normal (22 significand bits); unscaled, |w| ~ 0.02 would push every lo into the
// subnormals (~20 bits).  The factor is exact, a function of the tensor(s) only, and the GEMM epilogue undoes it with the device
// scalar this returns through l.acc_scale (it lives in the arena, so replicas receive it with the weights).
float WeightBuilder::hl_scale(Lin& l, const std::vector<std::string>& weight_names) {
  float* sc = (float*)arena.alloc(2 * sizeof(float));     // [0] 1 / factor, [1] "every weight is one f16" (allocated on empty replicas too: identical arena layout)
  l.acc_scale = sc;
  if (src.empty()) return 1.f;
  SDXL_HIP(hipMemsetAsync(sc, 0, 2 * sizeof(float), st));
  for (const std::string& n : weight_names) launch_absmax(fetch(n), spec(n).numel(), sc, st, true);
  float h = 0.f;
  SDXL_HIP(hipMemcpyAsync(&h, sc, sizeof(float), hipMemcpyDeviceToHost, st));
  SDXL_HIP(hipStreamSynchronize(st));
  int e = 0;
  if (h > 0.f && std::isfinite(h)) { (void)std::frexp(h, &e); e = 14 - e; }       // h = m * 2^(14 - e), m in [0.5, 1) -> h * 2^e in [2^13, 2^14)
  e = e > 24 ? 24 : (e < -24 ? -24 : e);
  const float wscale = std::ldexp(1.0f, e), inv = 1.0f / wscale;
  SDXL_HIP(hipMemcpyAsync(sc, &inv, sizeof(float), hipMemcpyHostToDevice, st));
  SDXL_HIP(hipStreamSynchronize(st));
  // real SDXL records hold f16 parameters: where every scaled weight of the matrix is exactly one f16 the lo halves are zero and the
  // kernel leaves out a third of its MFMAs (igemm_glds.hip, WX).  (gamma-folded weights never qualify: the split mode does not fold.)
  bool first = true;
  for (const std::string& n : weight_names) { launch_f16_exact(fetch(n), spec(n).numel(), wscale, sc + 1, st, !first); first = false; }
  return wscale;
}

// Plain f16 linear layers / 1x1 convolutions whose width is a multiple of 128 keep a second image of the packed weights in MFMA
// fragment order: the weights-in-registers GEMM (igemm_wreg.hip) streams it straight into VGPRs.  Same bytes, same arena (so a
// replica receives it with the weight broadcast); allocated on empty replicas too (identical arena layout).
void WeightBuilder::attach_wfrag(Lin& l, bool fill) {
  const int wdt = l.dt >= 0 ? l.dt : dt;
  if (!wfrag) return;
  if (wdt != DT_F16 || l.ksize != 1 || l.N % 128 != 0 || l.K != l.Kpad || l.Kpad % 64 != 0 || l.Kpad < 128 || l.cs || l.acc_scale) return;
  void* wf = arena.alloc((size_t)l.Npad * l.Kpad * 2);
  l.wf = wf;
  if (fill) launch_repack_wfrag(l.w, wf, l.Npad, l.Kpad, st);
}
Lin WeightBuilder::linear(const std::string& name, bool geglu, int dt_override) {
  const ParamSpec& s = spec(name + ".weight");
  Lin l; l.K = s.shape[0]; l.N = s.shape[1]; l.ksize = 1; l.cin = l.K;
  // dt_override: the GEMV weights of a split-operand model stay fp32; K % 32 != 0 cannot go through the HL pipeline either
  const int wdt = dt_override >= 0 ? dt_override : ((dt == DT_HL && l.K % 32 != 0) ? DT_F32 : dt);
  if (wdt != dt) l.dt = wdt;
  const int kt = wdt == DT_F16 ? 64 : 32;
  l.Kpad = (int)round_up(l.K, kt); l.Npad = (int)round_up(l.N, 128);
  void* w = arena.alloc((size_t)l.Npad * l.Kpad * dt_size(wdt));
  float* b = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  l.w = w; l.b = b;
  const float wscale = wdt == DT_HL ? hl_scale(l, {name + ".weight"}) : 1.f;
  if (src.empty()) { if (!geglu) attach_wfrag(l, false); return l; }
  launch_pack_linear(fetch(name + ".weight"), w, wdt, l.K, l.N, l.Kpad, l.Npad, geglu ? 1 : 0, 0, st, nullptr, wscale);
  const float* bsrc = has(name + ".bias") ? fetch(name + ".bias") : nullptr;
  launch_pack_bias(bsrc, b, l.N, l.Npad, geglu ? 1 : 0, 0, st);
  if (!geglu) attach_wfrag(l, true);
  return l;
}
// f16 GEMM with UN-ROUNDED weights (round 6, SDXL_DTYPE_F32_SPLIT_MIX's GEGLU projection): every weight as (hi, lo) f16 values along a doubled K --
// dst[n] = [f16(w) | f16((w - hi) * kHiLoScale)] -- against the A operand [a | a / kHiLoScale] (run_layernorm dup_scale): two MFMAs per product,
// the activations rounded once, the weights not at all.  An f16 Lin of K = 2 K0.
Lin WeightBuilder::linear_hilo(const std::string& name, bool geglu, bool dup, bool hl_interleave) {
  const ParamSpec& s = spec(name + ".weight");
  const int K0 = s.shape[0];
  SDXL_REQUIRE(K0 % 32 == 0, "linear_hilo: K % 32 == 0");
  Lin l; l.K = 2 * K0; l.N = s.shape[1]; l.ksize = 1; l.cin = l.K; l.dt = DT_F16; l.k_form = hl_interleave ? 2 : 1;
  l.Kpad = l.K; l.Npad = (int)round_up(l.N, 128);
  void* w = arena.alloc((size_t)l.Npad * l.Kpad * 2);
  float* b = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  l.w = w; l.b = b;
  if (src.empty()) { if (hl_interleave && !geglu) attach_wfrag(l, false); return l; }
  launch_pack_linear_hilo(fetch(name + ".weight"), w, K0, l.N, l.Npad, geglu ? 1 : 0, kHiLoScale, st, hl_interleave ? 2 : dup ? 1 : 0);
  const float* bsrc = has(name + ".bias") ? fetch(name + ".bias") : nullptr;
  launch_pack_bias(bsrc, b, l.N, l.Npad, geglu ? 1 : 0, 0, st);
  if (hl_interleave && !geglu) attach_wfrag(l, true);
  return l;
}
Lin WeightBuilder::fused_linear(const std::vector<std::string>& names, int dt_override) {
  Lin l; l.ksize = 1;
  int ntot = 0;
  for (const std::string& n : names) {
    const ParamSpec& s = spec(n + ".weight");
    if (l.K == 0) l.K = s.shape[0];
    SDXL_REQUIRE(l.K == s.shape[0], "fused_linear: K mismatch");
    ntot += s.shape[1];
  }
  l.N = ntot; l.cin = l.K;
  const int wdt = dt_override >= 0 ? dt_override : ((dt == DT_HL && l.K % 32 != 0) ? DT_F32 : dt);
  if (wdt != dt) l.dt = wdt;
  const int kt = wdt == DT_F16 ? 64 : 32;
  l.Kpad = (int)round_up(l.K, kt); l.Npad = (int)round_up(l.N, 128);
  char* w = (char*)arena.alloc((size_t)l.Npad * l.Kpad * dt_size(wdt));
  float* b = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  l.w = w; l.b = b;
  float wscale = 1.f;
  if (wdt == DT_HL) {     // ONE factor for the concatenated matrix (one accumulator scale per GEMM)
    std::vector<std::string> wn;
    for (const std::string& n : names) wn.push_back(n + ".weight");
    wscale = hl_scale(l, wn);
  }
  if (src.empty()) return l;
  SDXL_HIP(hipMemsetAsync(w, 0, (size_t)l.Npad * l.Kpad * dt_size(wdt), st));
  SDXL_HIP(hipMemsetAsync(b, 0, (size_t)l.Npad * sizeof(float), st));
  int off = 0;
  for (const std::string& n : names) {
    const ParamSpec& s = spec(n + ".weight");
    const int N = s.shape[1];
    launch_pack_linear(fetch(n + ".weight"), w, wdt, l.K, N, l.Kpad, N, 0, off, st, nullptr, wscale);   // exactly N rows at row offset
    const float* bsrc = has(n + ".bias") ? fetch(n + ".bias") : nullptr;
    launch_pack_bias(bsrc, b, N, N, 0, off, st);
    off += N;
  }
  l.w = w; l.b = b;
  return l;
}
// LayerNorm(gamma, beta) followed by Linear(W, b):  LN(x) W + b = rstd (x - mu) (diag(gamma) W) + (beta W + b)
//   = rstd * (x W') - rstd * mu * colsum(W') + b'   -- W' is packed (rounded to the compute dtype) FIRST and the column sums
// are taken over the rounded values, so the identity holds exactly for what the MFMA multiplies.
Lin WeightBuilder::linear_ln(const std::string& name, bool geglu, const std::string& norm) {
  return fold_ln({name}, norm, geglu);
}
Lin WeightBuilder::fused_linear_ln(const std::vector<std::string>& names, const std::string& norm) {
  return fold_ln(names, norm, false);
}
bool WeightBuilder::all_f16_exact(const std::vector<std::string>& names) {
  if (src.empty() || names.empty()) return true;      // (replica ranks receive rank 0's arena: its decision is theirs -- see UNet::build_weights)
  float* flag = (float*)tmp2;
  if (tmp2_numel < 4) { if (tmp2) SDXL_HIP(hipFree(tmp2)); SDXL_HIP(hipMalloc((void**)&tmp2, 64 * sizeof(float))); tmp2_numel = 64; flag = tmp2; }
  bool first = true;
  for (const std::string& n : names) { launch_f16_exact(fetch(n), spec(n).numel(), 1.0f, flag, st, !first); first = false; }
  float h = 0.f;
  SDXL_HIP(hipMemcpyAsync(&h, flag, sizeof(float), hipMemcpyDeviceToHost, st));
  SDXL_HIP(hipStreamSynchronize(st));
  return h != 0.f;
}
Lin WeightBuilder::fold_ln(const std::vector<std::string>& names, const std::string& norm, bool geglu, int dt_override, bool shadow, Lin* plain, bool hilo_dup, bool hl_interleave) {
  SDXL_REQUIRE(!geglu || names.size() == 1, "GEGLU packing applies to a single projection");
  const int mdt = dt;                       // the model's dtype
  const int dt = dt_override >= 0 ? dt_override : mdt;      // (shadows the member below: the dtype THIS matrix is packed in)
  Lin l; l.ksize = 1;
  if (dt != mdt) l.dt = dt;
  int ntot = 0;
  for (const std::string& n : names) {
    const ParamSpec& s = spec(n + ".weight");
    if (l.K == 0) l.K = s.shape[0];
    SDXL_REQUIRE(l.K == s.shape[0], "fused_linear_ln: K mismatch");
    ntot += s.shape[1];
  }
  SDXL_REQUIRE(!hilo_dup || (shadow && names.size() == 1 && dt == DT_F16 && l.K % 32 == 0), "fold_ln: the (hi | lo) form is a single f16 shadow-form projection");
  const int K0 = l.K;                       // columns of the LayerNorm
  SDXL_REQUIRE(!hl_interleave || (shadow && !hilo_dup && dt == DT_F16 && l.K % 32 == 0), "fold_ln: the HL16-interleaved form is an f16 shadow-form projection");
  if (hilo_dup) { l.ln_k = K0; l.K = 2 * K0; l.k_form = 1; }
  if (hl_interleave) { l.ln_k = K0; l.K = 2 * K0; l.k_form = 2; }
  l.N = ntot; l.cin = l.K;
  const int kt = dt == DT_F16 ? 64 : 32;
  l.Kpad = (int)round_up(l.K, kt); l.Npad = (int)round_up(l.N, 128);
  char* w = (char*)arena.alloc((size_t)l.Npad * l.Kpad * dt_size(dt));
  float* b = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  float* cs = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  float* leps = (float*)arena.alloc(sizeof(float));
  l.w = w; l.b = b; l.cs = cs; l.ln_eps = leps;
  float* bplain = plain ? (float*)arena.alloc((size_t)l.Npad * sizeof(float)) : nullptr;
  if (plain) { *plain = l; plain->b = bplain; plain->cs = nullptr; plain->ln_eps = nullptr; }
  if (src.empty()) return l;
  SDXL_HIP(hipMemcpyAsync(leps, fetch(norm + ".eps"), sizeof(float), hipMemcpyDeviceToDevice, st));
  const size_t need = 3 * (size_t)l.Npad + 2 * (size_t)l.K + 64;
  if (need > tmp2_numel) {
    if (tmp2) SDXL_HIP(hipFree(tmp2));
    SDXL_HIP(hipMalloc((void**)&tmp2, need * sizeof(float)));
    tmp2_numel = need;
  }
  float* gamma = tmp2;                 // [K]
  float* beta = tmp2 + l.K;            // [K]
  float* bsrc = beta + l.K;            // [Npad] canonical bias of the current projection
  float* bfold = bsrc + l.Npad;        // [Npad] beta W + bias, canonical order
  SDXL_REQUIRE(spec(norm + ".gamma").shape[0] == K0, "fused_linear_ln: norm width mismatch");
  SDXL_HIP(hipMemcpyAsync(gamma, fetch(norm + ".gamma"), K0 * sizeof(float), hipMemcpyDeviceToDevice, st));
  SDXL_HIP(hipMemcpyAsync(beta, fetch(norm + ".beta"), K0 * sizeof(float), hipMemcpyDeviceToDevice, st));
  SDXL_HIP(hipMemsetAsync(w, 0, (size_t)l.Npad * l.Kpad * dt_size(dt), st));
  SDXL_HIP(hipMemsetAsync(b, 0, (size_t)l.Npad * sizeof(float), st));
  if (bplain) SDXL_HIP(hipMemsetAsync(bplain, 0, (size_t)l.Npad * sizeof(float), st));
  int off = 0;
  for (const std::string& n : names) {
    const ParamSpec& s = spec(n + ".weight");
    const int N = s.shape[1];
    const bool hb = has(n + ".bias");
    if (hb) SDXL_HIP(hipMemcpyAsync(bsrc, fetch(n + ".bias"), N * sizeof(float), hipMemcpyDeviceToDevice, st));
    const float* wsrc = fetch(n + ".weight");
    // (shadow form: the matrix stays W -- an f16-representable parameter stays exact -- and gamma multiplies the A operand instead)
    if (hl_interleave) launch_pack_linear_hilo(wsrc, w, K0, N, names.size() == 1 ? l.Npad : N, geglu ? 1 : 0, kHiLoScale, st, 2, off);      // the weight twice per 16-channel group
    else if (hilo_dup) launch_pack_linear_hilo(wsrc, w, K0, N, l.Npad, geglu ? 1 : 0, kHiLoScale, st, 1);      // (w | w / kHiLoScale): the (hi | lo) shadow's partner
    else if (names.size() == 1) launch_pack_linear(wsrc, w, dt, l.K, N, l.Kpad, l.Npad, geglu ? 1 : 0, 0, st, shadow ? nullptr : gamma);
    else launch_pack_linear(wsrc, w, dt, l.K, N, l.Kpad, N, 0, off, st, shadow ? nullptr : gamma);
    launch_beta_dot(wsrc, beta, hb ? bsrc : nullptr, bfold, K0, N, st);
    launch_pack_bias(bfold, b, N, names.size() == 1 ? l.Npad : N, geglu ? 1 : 0, off, st);
    if (bplain) launch_pack_bias(hb ? bsrc : nullptr, bplain, N, names.size() == 1 ? l.Npad : N, geglu ? 1 : 0, off, st);
    off += N;
  }
  if (shadow) launch_colsum_packed(w, dt, l.Kpad, l.Npad, cs, st, gamma, K0, hl_interleave ? 1 : 0);      // cs[n] = sum_k gamma[k] W[k][n] over the packed (rounded) values
  else launch_colsum_packed(w, dt, l.Kpad, l.Npad, cs, st);
  return l;
}
Lin WeightBuilder::conv(const std::string& name) {
  const ParamSpec& s = spec(name + ".weight");
  Lin l; l.N = s.shape[0]; l.cin = s.shape[1]; l.ksize = s.shape[2];
  l.K = l.cin * l.ksize * l.ksize;
  // split-operand models (DT_HL): the direct-to-LDS pipeline needs 32-channel k-tiles; the few layers without them (the 3- / 4- /
  // 8-channel ends of the VAE) are packed and run as plain fp32 on the generic kernel
  const int wdt = (dt == DT_HL && l.cin % 32 != 0) ? DT_F32 : dt;
  if (wdt != dt) l.dt = wdt;
  const int kt = wdt == DT_F16 ? 64 : 32;
  l.Kpad = (int)round_up(l.K, kt); l.Npad = (int)round_up(l.N, 128);
  void* w = arena.alloc((size_t)l.Npad * l.Kpad * dt_size(wdt));
  float* b = (float*)arena.alloc((size_t)l.Npad * sizeof(float));
  l.w = w; l.b = b;
  const float wscale = wdt == DT_HL ? hl_scale(l, {name + ".weight"}) : 1.f;
  if (src.empty()) { attach_wfrag(l, false); return l; }
  launch_pack_conv(fetch(name + ".weight"), w, wdt, l.N, l.cin, l.ksize, l.Kpad, l.Npad, st, wscale);
  launch_pack_bias(fetch(name + ".bias"), b, l.N, l.Npad, 0, 0, st);
  attach_wfrag(l, true);
  return l;
}
NormW WeightBuilder::norm(const std::string& name) {
  const ParamSpec& s = spec(name + ".gamma");
  NormW n; n.C = s.shape[0];
  float* g = (float*)arena.alloc((size_t)n.C * sizeof(float));
  float* b = (float*)arena.alloc((size_t)n.C * sizeof(float));
  float* e = (float*)arena.alloc(sizeof(float));
  n.gamma = g; n.beta = b; n.eps = e;
  if (src.empty()) return n;
  SDXL_HIP(hipMemcpyAsync(g, fetch(name + ".gamma"), n.C * sizeof(float), hipMemcpyDeviceToDevice, st));
  SDXL_HIP(hipMemcpyAsync(b, fetch(name + ".beta"), n.C * sizeof(float), hipMemcpyDeviceToDevice, st));
  SDXL_HIP(hipMemcpyAsync(e, fetch(name + ".eps"), sizeof(float), hipMemcpyDeviceToDevice, st));
  return n;
}

// ------------------------------------------------------------------------------------------ launch helpers
void WarmSeq::finish() {
  // every launch that reads 0.25 - 8 MiB of weights (14 MiB if it runs on the weights-in-registers kernel: FF-out) gets a host among the
  // four sequence entries in front of it (the nearest with room; the sequence wraps: the first launches of the next forward are warmed by the
  // last ones of this); a host carries at most three regions inside its 14 MiB budget (the packed context a fused cross-attention projection reads is a region of its own).  Measured (profiles/r04_weight_warming_ab.txt):
  // the N = 1280 projections start ~2 us earlier on warmed weights (their prologue waits for the Infinity Cache instead of HBM); the
  // fused QKV (9.8 MB) and GEGLU (26 MB) weights gain nothing, and a host that carries 26 MB outlives its own tiles.
  const size_t n = seq.size();
  for (Item& it : seq) for (int r = 0; r < 3; ++r) { it.warm[r] = nullptr; it.warm_bytes[r] = 0; }
  for (size_t j = 0; j < n && n > 4; ++j) {
    const unsigned bytes = seq[j].bytes;
    if (bytes < (1u << 18) || bytes > (seq[j].host ? 14u << 20 : 8u << 20)) continue;
    for (size_t d = 1; d <= 4; ++d) {
      Item& h = seq[(j + n - d) % n];
      if (!h.host || h.warm[2] || h.w == seq[j].w || h.budget < bytes) continue;
      const int r = !h.warm[0] ? 0 : !h.warm[1] ? 1 : 2;
      h.warm[r] = seq[j].w; h.warm_bytes[r] = bytes;
      h.budget -= bytes;
      break;
    }
  }
  recording = false; ready = true; pos = 0;
  if (std::getenv("SDXL_WARM_DUMP")) {     // schedule of the plan, one line per sequence entry: bytes, host?, regions it carries; '*' = nobody warms this entry
    std::vector<char> covered(n, 0);
    for (size_t i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) if (seq[i].warm[r]) for (size_t j = 0; j < n; ++j) if (seq[j].w == seq[i].warm[r]) covered[j] = 1;
    size_t tb = 0, cb = 0;
    for (size_t i = 0; i < n; ++i) {
      tb += seq[i].bytes; if (covered[i]) cb += seq[i].bytes;
      std::fprintf(stderr, "[warm] %3zu %9u B %s carries %u + %u + %u B %s\n", i, seq[i].bytes, seq[i].host ? "host" : "    ", seq[i].warm_bytes[0], seq[i].warm_bytes[1],
                   seq[i].warm_bytes[2], covered[i] ? "" : "*");
    }
    std::fprintf(stderr, "[warm] %zu entries, %.1f MB read per forward, %.1f MB of it warmed\n", n, tb / 1048576.0, cb / 1048576.0);
  }
}
bool run_conv(Exec& ex, const Lin& w, const Act& a, int cin, const ConvGeom& g, const Act& out, const Epi& e) {
  if (ex.dry) return false;
  SDXL_REQUIRE(cin == w.cin, "run_conv: channel mismatch");
  IgemmParams p{};
  p.A = a.p; p.W = w.w; p.Wf = w.wf; p.a_dt = a.dt;
  p.B = g.B; p.Hin = g.Hin; p.Win = g.Win; p.Cin = cin; p.lda = a.ld;
  p.Hout = g.Hout; p.Wout = g.Wout;
  p.ksize = g.ksize; p.stride = g.stride; p.pad = g.pad; p.up = g.up;
  p.M = g.B * g.Hout * g.Wout; p.N = w.N; p.K = w.K; p.Kpad = w.Kpad;
  p.bias = w.b; p.ebias = e.ebias; p.ebias_ld = e.ebias_ld;
  p.rpb = e.rpb ? e.rpb : g.Hout * g.Wout;
  p.act = e.act;
  p.R = e.R.p; p.ldr = e.R.ld; p.r_dt = e.R.dt;
  p.C = out.p; p.ldc = out.ld; p.c_dt = out.dt;
  p.n_split = e.n_split >= 0 ? e.n_split : w.N;
  p.Ct = e.Ct; p.ct_rows = e.ct_rows; p.ct_ld = e.ct_ld;
  p.ln_stat = e.ln_stat; p.ln_slots = (w.ln_k ? w.ln_k : w.K) / 64; p.ln_cs = w.cs; p.ln_invc = 1.0f / (float)(w.ln_k ? w.ln_k : w.K); p.ln_eps = 1e-5f; p.ln_eps_ptr = w.ln_eps;
  p.stat_out = e.stat_out; p.stat_slots = w.N / 64;
  p.splitk_ws = ex.splitk_ws; p.splitk_ws_bytes = ex.splitk_ws_bytes; p.splitk_cnt = ex.splitk_cnt; p.splitk = 0;
  p.xa_k = e.xa_k; p.xa_nctx = e.xa_nctx; p.xa_scale = e.xa_scale; p.xa_k_lo = e.xa_k_lo;
  // f16 shadow of an fp32 output (+ its row statistics) for the GEMM behind the next LayerNorm: only where the selection picks the
  // weights-in-registers kernel anyway; otherwise neither is written and the caller runs the LayerNorm launch
  if (e.shadow_done) *e.shadow_done = false;
  if (e.shadow) {
    p.shadow = e.shadow; p.shadow_ld = e.shadow_ld; p.shadow_gamma = e.shadow_gamma; p.shadow_lo_scale = e.shadow_lo_scale;
    const bool ok = (w.dt >= 0 ? w.dt : ex.cdt) == DT_F16 && igemm_wreg_selected(p);
    if (!ok) { p.shadow = nullptr; p.shadow_gamma = nullptr; p.stat_out = nullptr; p.shadow_lo_scale = 0.f; }
    if (e.shadow_done) *e.shadow_done = ok;
  }
  // GroupNorm statistics of the output from this GEMM's epilogue -- only when the kernel the selection picks anyway can do it
  p.gn_part = nullptr;
  {
    const int wdt_ = w.dt >= 0 ? w.dt : ex.cdt;      // (f16 engines; round 6: the split-operand convolutions that run the 256x128 split-K kernel)
    if (e.gn_part && (wdt_ == DT_F16 || wdt_ == DT_HL) && wdt_ == ex.cdt) { p.gn_part = e.gn_part; if (!igemm_gn_part_ok(p)) p.gn_part = nullptr; }
  }
  SDXL_REQUIRE(!e.xa_k || igemm_xattn_ok(a.dt, out.dt, p.M, p.N, p.K, p.rpb, e.xa_nctx), "fused cross-attention: unsupported shape");
  SDXL_REQUIRE(!e.ln_stat || (w.ln_k ? w.ln_k : w.K) % 64 == 0, "LayerNorm-folded GEMM needs K % 64 == 0");
  SDXL_REQUIRE(!e.stat_out || (w.N % 64 == 0 && (e.n_split < 0 || e.n_split >= w.N) && e.act == 0), "row statistics need a plain N % 64 == 0 output");
  SDXL_REQUIRE(!e.ln_stat || w.cs, "ln_stat given but the weight is not LayerNorm-folded");
  SDXL_REQUIRE(!w.cs || e.ln_stat, "LayerNorm-folded weight used without row statistics");
  SDXL_REQUIRE(!((w.dt >= 0 ? w.dt : ex.cdt) == DT_F32 && a.dt != DT_F32), "f32 compute needs f32 activations");
  if (ex.prof) ex.prof->begin(Profiler::IGEMM, 2.0 * p.M * (double)p.N * p.K, ex.s, p.M, p.N, p.K, p.ksize, e.cls);
  p.acc_scale = w.acc_scale;
  p.a_scale = (w.dt >= 0 ? w.dt : ex.cdt) == DT_HL ? a.a_scale : nullptr;
  p.a_scale_rpb = (p.a_scale && a.a_scale_n > 1) ? p.M / a.a_scale_n : 0;      // (M = entries x output rows per entry)
  if (ex.warm && (w.dt >= 0 ? w.dt : ex.cdt) == DT_F16) {
    // weight warming: the plan's first forward records which weights every launch reads and whether its kernel has idle CUs to host
    // warming workgroups; later forwards hand launch i the weights of a later launch (WarmSeq::finish)
    WarmSeq& ws = *ex.warm;
    // the region the launch reads cold: its weights and the per-column vectors next to them in the arena (allocation order: packed
    // weights, bias, [column sums, eps of a folded LayerNorm], [fragment-order image])
    const bool host = igemm_wreg_selected(p);
    const size_t wbytes = (size_t)w.Npad * w.Kpad * 2;
    const char* lo; const char* hi;
    if (host) {
      const char* f = reinterpret_cast<const char*>(p.Wf), *b = reinterpret_cast<const char*>(w.b);
      lo = (b && b < f && f - b <= (64 << 10)) ? b : f;
      hi = f + wbytes;
    } else {
      lo = reinterpret_cast<const char*>(p.W); hi = lo + wbytes;
      auto ext = [&](const void* q, size_t sz) { const char* c = reinterpret_cast<const char*>(q); if (q && c >= hi && c - hi <= (64 << 10)) hi = c + sz; };
      ext(w.b, (size_t)w.Npad * 4); ext(w.cs, (size_t)w.Npad * 4); ext(w.ln_eps, 4);
    }
    if (ws.recording) {
      // (a fused cross-attention projection also reads the packed context of its batch entries: a target of its own in front of it)
      if (p.xa_k) ws.seq.push_back(WarmSeq::Item{p.xa_k, (unsigned)xattn_pack_bytes(p.M / p.rpb, p.N), false, 0u, {nullptr, nullptr, nullptr}, {0u, 0u, 0u}});
      // what a host can carry without outliving its own tiles: its >= 36 warmers pull ~30 GB/s each, the shortest host runs 13 us
      const unsigned budget = host ? 14u << 20 : 0u;
      ws.seq.push_back(WarmSeq::Item{lo, (unsigned)(hi - lo), host, budget, {nullptr, nullptr, nullptr}, {0u, 0u, 0u}});
    } else if (ws.ready) {
      if (p.xa_k) ++ws.pos;
      if (ws.pos < ws.seq.size()) {
        const WarmSeq::Item& it = ws.seq[ws.pos++];
        if (it.w == lo) for (int r = 0; r < 3; ++r) { p.warm[r] = it.warm[r]; p.warm_bytes[r] = it.warm_bytes[r]; }     // (the recorded launch: anything else means a different plan)
      }
    }
  }
  launch_igemm(p, w.dt >= 0 ? w.dt : ex.cdt, ex.s);
  {   // a refused launch (bad grid / LDS attribute) must not pass silently
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess)
      throw Error(std::string("implicit-GEMM launch failed (") + hipGetErrorString(le) + "): M=" + std::to_string(p.M) + " N=" + std::to_string(p.N) +
                  " K=" + std::to_string(p.K) + " ksize=" + std::to_string(p.ksize) + " a_dt=" + std::to_string(p.a_dt) + " c_dt=" + std::to_string(p.c_dt) +
                  " compute=" + std::to_string(w.dt >= 0 ? w.dt : ex.cdt) + " act=" + std::to_string(p.act) + " n_split=" + std::to_string(p.n_split));
  }
  if (ex.prof) ex.prof->end(ex.s);
  {   // SDXL_NAN_CHECK=1 (eager forwards only: it synchronises): report the first GEMM whose output holds a non-finite value, and its input's state
    static const bool nan_check = std::getenv("SDXL_NAN_CHECK") != nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (nan_check && hipStreamIsCapturing(ex.s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      const int nout = p.act == 1 ? p.N / 2 : (p.n_split < p.N ? p.n_split : p.N);
      const unsigned bo = nout > 0 ? count_nonfinite(out.p, out.dt, out.ld, (size_t)p.M, nout, ex.s) : 0;
      if (bo) {
        const unsigned bi = p.ksize == 1 && p.stride == 1 && !p.up ? count_nonfinite(a.p, a.dt, a.ld, (size_t)p.M, cin, ex.s) : 0;
        std::fprintf(stderr, "[nan] GEMM tag %d M %d N %d K %d ks %d act %d a_dt %d c_dt %d compute %d: %u non-finite outputs, %u non-finite inputs\n", e.cls, p.M, p.N,
                     p.K, p.ksize, p.act, p.a_dt, p.c_dt, w.dt >= 0 ? w.dt : ex.cdt, bo, bi);
      }
    }
  }
  if (ex.fork_ev && ++ex.launches == ex.fork_after) SDXL_HIP(hipEventRecord(ex.fork_ev, ex.s));
  return p.gn_part != nullptr;
}
// GEMM operand view of a residual-stream tensor: the split-operand mode keeps the stream in fp32 and stages (hi, lo) f16 pairs, so
// GEMMs that read the stream directly (skip / nin_shortcut 1x1 convs, up- / downsamplers, proj_out) get an HL16 copy; other modes
// -- and layers whose weights were packed fp32 (w.dt) -- read x itself
Act hl_operand(Exec& ex, const Lin& w, const Act& x, size_t rows, int C, int nb, float* have_max) {
  if (ex.cdt != DT_HL || x.dt != DT_F32 || w.dt == DT_F32) return x;
  if (nb < 1 || rows % (size_t)nb != 0) { SDXL_REQUIRE(!have_max, "hl_operand: absmax partials need whole batch entries"); nb = 1; }
  Act o = ex.alloc(rows, C, DT_HL);
  float* sc = have_max ? have_max : (float*)ex.act->alloc(hl_scale_floats(nb) * sizeof(float));      // per entry: max|x| partials, 2^-e -- the stream's range is the model's, not ours
  if (!ex.dry) launch_f32_to_hl_scaled(x.p, x.ld, o.p, o.ld, rows, C, sc, ex.s, nb, have_max != nullptr);
  o.a_scale = hl_scale_inv(sc, nb); o.a_scale_n = nb;
  return o;
}
bool run_linear(Exec& ex, const Lin& w, const Act& a, int M, const Act& out, const Epi& e) {
  ConvGeom g{1, M, 1, M, 1, 1, 1, 0, 0};
  Epi e2 = e;
  if (e2.rpb == 0) e2.rpb = M;
  return run_conv(ex, w, a, w.cin, g, out, e2);
}
void run_groupnorm(Exec& ex, const NormW& n, const Act& x, int B, int HW, const Act& y, bool silu, int groups, float* absmax_out) {
  if (ex.dry) return;
  SDXL_REQUIRE(groups >= 1 && groups <= 256 && n.C % groups == 0, "The number of channels must be divisible by the number of groups");
  GroupNormParams p{};
  p.X = x.p; p.x_dt = x.dt; p.ldx = x.ld;
  p.Y = y.p; p.y_dt = y.dt; p.ldy = y.ld;
  p.gamma = n.gamma; p.beta = n.beta; p.partial = ex.gn_partial;
  p.B = B; p.HW = HW; p.C = n.C; p.G = groups; p.eps = 1e-5f; p.eps_ptr = n.eps; p.silu = silu ? 1 : 0;
  p.chan_part = x.gn_part; p.chan_rt = x.gn_rt; p.chan_rows = 256;          // statistics left by x's producer (Act::gn_part)
  SDXL_REQUIRE(!absmax_out || (!x.gn_part && x.dt == DT_F32), "GroupNorm absmax side output needs the statistics pass over an fp32 tensor");
  p.absmax_out = absmax_out;
  SDXL_REQUIRE(!x.gn_part || x.gn_rt * 256 == HW, "producer GroupNorm statistics do not cover the tensor");
  {   // SDXL_GN_DUMP=1 (eager forwards): one line per GroupNorm -- shape and whether its statistics came from the producer
    static const bool gn_dump = std::getenv("SDXL_GN_DUMP") != nullptr;
    if (gn_dump) std::fprintf(stderr, "[gn] B %d HW %d C %d x_dt %d %s\n", B, HW, n.C, x.dt, x.gn_part ? "producer statistics" : "statistics pass");
  }
  if (ex.prof) ex.prof->begin(Profiler::GROUPNORM, 0.0, ex.s);
  launch_groupnorm(p, ex.s);
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw Error(std::string("GroupNorm launch failed (") + hipGetErrorString(le) + "): C=" + std::to_string(p.C) + " HW=" + std::to_string(p.HW) + " y_dt=" + std::to_string(p.y_dt));
  }
  if (ex.prof) ex.prof->end(ex.s);
}
void run_layernorm(Exec& ex, const NormW& n, const Act& x, int rows, const Act& y, float dup_scale) {
  if (ex.dry) return;
  LayerNormParams p{};
  p.X = x.p; p.x_dt = x.dt; p.ldx = x.ld; p.Y = y.p; p.y_dt = y.dt; p.ldy = y.ld;
  p.gamma = n.gamma; p.beta = n.beta; p.rows = rows; p.C = n.C; p.eps = 1e-5f; p.eps_ptr = n.eps; p.dup_scale = dup_scale;
  if (ex.prof) ex.prof->begin(Profiler::LAYERNORM, 0.0, ex.s);
  launch_layernorm(p, ex.s);
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw Error(std::string("LayerNorm launch failed (") + hipGetErrorString(le) + "): C=" + std::to_string(p.C) + " rows=" + std::to_string(p.rows) + " y_dt=" + std::to_string(p.y_dt));
  }
  if (ex.prof) ex.prof->end(ex.s);
}

}  // namespace sdxl
