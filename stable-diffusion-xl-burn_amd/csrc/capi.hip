// C ABI of the engine (include/sdxl_mi355.h): opaque handles, status codes + thread-local error text, never aborts.
#include "../../include/sdxl_mi355.h"
#include "engine.h"
#include <algorithm>

#include <cmath>
#include <cstring>

using namespace sdxl;

struct sdxl_ctx { int device = 0; hipStream_t stream = nullptr; };
struct sdxl_unet { sdxl_ctx* ctx = nullptr; UNet* u = nullptr; bool owned = true; };
struct sdxl_diffuser { sdxl_ctx* ctx = nullptr; Diffuser* d = nullptr; sdxl_unet view; };
struct sdxl_vae { sdxl_ctx* ctx = nullptr; Vae* v = nullptr; };
struct sdxl_clip { sdxl_ctx* ctx = nullptr; ClipText* c = nullptr; };

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }
#define API_BEGIN try {
#define API_END                                                              \
  return SDXL_OK;                                                            \
  } catch (const sdxl::Error& e) { return fail(SDXL_ERR_RUNTIME, e.what()); } \
  catch (const std::exception& e) { return fail(SDXL_ERR_RUNTIME, e.what()); } \
  catch (...) { return fail(SDXL_ERR_RUNTIME, "unknown error"); }

hipStream_t pick(sdxl_ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }
void use(sdxl_ctx* c) { SDXL_HIP(hipSetDevice(c->device)); }

UNetCfg to_cfg(const sdxl_unet_config* c) {
  SDXL_REQUIRE(c != nullptr, "null config");
  SDXL_REQUIRE(c->n_levels >= 1 && c->n_levels <= 8, "n_levels out of range");
  UNetCfg u;
  u.adm_in_channels = c->adm_in_channels; u.in_channels = c->in_channels; u.out_channels = c->out_channels;
  u.model_channels = c->model_channels; u.n_head_channels = c->n_head_channels; u.context_dim = c->context_dim;
  u.is_refiner = c->is_refiner != 0;
  for (int i = 0; i < c->n_levels; ++i) { u.channel_mults.push_back(c->channel_mults[i]); u.transformer_depths.push_back(c->transformer_depths[i]); }
  SDXL_REQUIRE(u.model_channels > 0 && u.n_head_channels > 0 && u.context_dim > 0 && u.adm_in_channels > 0, "bad UNet config");
  return u;
}
ClipCfg to_ccfg(const sdxl_clip_config* c) {
  SDXL_REQUIRE(c != nullptr, "null config");
  ClipCfg k;
  k.n_vocab = c->n_vocab; k.n_state = c->n_state; k.embed_dim = c->embed_dim; k.n_head = c->n_head; k.n_ctx = c->n_ctx;
  k.n_layer = c->n_layer; k.quick_gelu = c->quick_gelu != 0;
  return k;
}
VaeCfg to_vcfg(const sdxl_vae_config* c) {
  SDXL_REQUIRE(c != nullptr, "null config");
  SDXL_REQUIRE(c->n_blocks >= 1 && c->n_blocks <= 8, "n_blocks out of range");
  VaeCfg v; v.enc.clear(); v.dec.clear();
  for (int i = 0; i < c->n_blocks; ++i) { v.enc.push_back({c->enc_in[i], c->enc_out[i]}); v.dec.push_back({c->dec_in[i], c->dec_out[i]}); }
  v.n_group = c->n_group; v.enc_out = c->enc_out_channels; v.scale_factor = c->scale_factor;
  SDXL_REQUIRE(v.n_group >= 1 && v.n_group <= 256, "n_group out of range (1..256)");
  for (int i = 0; i < c->n_blocks; ++i)
    for (int ch : {c->enc_in[i], c->enc_out[i], c->dec_in[i], c->dec_out[i]})
      SDXL_REQUIRE(ch > 0 && ch % v.n_group == 0 && ch % 8 == 0,
                   "The number of channels must be divisible by the number of groups (and by 8)");   // groupnorm/mod.rs:19-24
  return v;
}
void no_mix(int dtype) {     // the mixed mode is a property of the UNet driver (which classes run in f16): UNet / Diffuser handles only
  if (dtype == SDXL_DTYPE_F32_SPLIT_MIX || dtype == SDXL_DTYPE_F32_SPLIT_MIX_F16W || dtype == SDXL_DTYPE_F32_SPLIT_MIX_F16W_GEGLU2 || dtype == SDXL_DTYPE_F32_SPLIT_F16W) throw Error("SDXL_DTYPE_F32_SPLIT_MIX* are UNet / Diffuser modes (use SDXL_DTYPE_F32_SPLIT here)");
}
int mix_of(int dtype) {
  return dtype == SDXL_DTYPE_F32_SPLIT_MIX ? (MIX_ATTN_F16 | MIX_GEGLU_F16 | MIX_GEGLU_HILO)      // (round 6: GEGLU weights as (hi, lo) pairs along K -- activation rounding only on any weights)
       : dtype == SDXL_DTYPE_F32_SPLIT_MIX_F16W ? (MIX_ATTN_F16 | MIX_GEGLU_F16 | MIX_QKV_F16 | MIX_FF_F16 | MIX_OUT1_F16 | MIX_OUT2_F16 | MIX_Q2_F16 | MIX_LN_SHADOW | MIX_XATTN_SPLIT)
       : dtype == SDXL_DTYPE_F32_SPLIT_MIX_F16W_GEGLU2 ? (MIX_ATTN_F16 | MIX_GEGLU_F16 | MIX_QKV_F16 | MIX_FF_F16 | MIX_OUT1_F16 | MIX_OUT2_F16 | MIX_Q2_F16 | MIX_LN_SHADOW | MIX_XATTN_SPLIT | MIX_GEGLU_AHILO)
       : dtype == SDXL_DTYPE_F32_SPLIT_F16W ? (MIX_LINEAR_F16X2 | MIX_XATTN_SPLIT | MIX_LN_SHADOW) : 0;      // (no class on f16 OPERANDS: fp32-class arithmetic on the f16 kernels)
       // (round 6: + the cross-attention query projection on f16 with an fp32 q, the split-precision 77-key attention inside its epilogue, and the LayerNorms in
       //  front of the f16 projections folded through the f16 shadow of the stream -- DESIGN 4.1; MIX_XATTN_F16 stays a knob: DESIGN 11.2b)
}
void dtypes(int dtype, int& cdt, int& sdt) {
  switch (dtype) {
    case SDXL_DTYPE_F32: cdt = DT_F32; sdt = DT_F32; break;
    case SDXL_DTYPE_F16: cdt = DT_F16; sdt = DT_F16; break;
    case SDXL_DTYPE_F16_F32RES: cdt = DT_F16; sdt = DT_F32; break;
    case SDXL_DTYPE_F32_SPLIT: cdt = DT_HL; sdt = DT_F32; break;   // UNet / Diffuser / VAE only (no_split() guards the rest)
    case SDXL_DTYPE_F32_SPLIT_MIX: case SDXL_DTYPE_F32_SPLIT_MIX_F16W: case SDXL_DTYPE_F32_SPLIT_MIX_F16W_GEGLU2: case SDXL_DTYPE_F32_SPLIT_F16W: cdt = DT_HL; sdt = DT_F32; break;   // UNet / Diffuser only (mix_of() carries the f16 classes)
    default: throw Error("unknown dtype");
  }
}
void no_split(int cdt, const char* what) {
  if (cdt == DT_HL) throw Error(std::string("SDXL_DTYPE_F32_SPLIT is not available for ") + what);
}
void vae_dtype(int dtype, int& cdt) {     // the VAE additionally takes the split-operand fp32-class mode
  int sdt;
  if (dtype == SDXL_DTYPE_F32_SPLIT) { cdt = DT_HL; return; }
  no_mix(dtype); dtypes(dtype, cdt, sdt);
}
int spec_out(const std::vector<ParamSpec>& specs, int index, const char** name, int* ndim, int64_t shape[4], int* kind,
             float* sc, float* mean) {
  if (index < 0 || index >= (int)specs.size()) return fail(SDXL_ERR_INVALID, "parameter index out of range");
  static thread_local std::string keep;
  const ParamSpec& p = specs[index];
  keep = p.name;
  if (name) *name = keep.c_str();
  if (ndim) *ndim = (int)p.shape.size();
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = i < (int)p.shape.size() ? p.shape[i] : 1;
  if (kind) *kind = p.kind;
  if (sc) *sc = p.scale;
  if (mean) *mean = p.mean;
  return SDXL_OK;
}
struct Tmp {   // scoped device scratch for the single-op entry points
  std::vector<void*> ptrs;
  ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
  void* get(size_t bytes) { void* p = nullptr; SDXL_HIP(hipMalloc(&p, bytes ? bytes : 16)); ptrs.push_back(p); return p; }
};
// fragment-order image of a plain f16 linear / 1x1 weight for the single-op entry points (what WeightBuilder::attach_wfrag does
// for the models): the operators then run the same kernel selection as the models
static void tmp_wfrag(Lin& l, int cdt, bool geglu, Tmp& tmp, hipStream_t s) {
  if (cdt != DT_F16 || geglu || l.ksize != 1 || l.N % 128 != 0 || l.K != l.Kpad || l.Kpad % 64 != 0 || l.Kpad < 128 || l.cs || l.acc_scale) return;
  void* wf = tmp.get((size_t)l.Npad * l.Kpad * 2);
  launch_repack_wfrag(l.w, wf, l.Npad, l.Kpad, s);
  l.wf = wf;
}
// the single-op entry points run the same kernel selection as the models, split-K included (f16 compute only)
void give_splitk_ws(Exec& ex, Tmp& tmp, int batch, int rows_per_entry, int n, hipStream_t s) {
  if (ex.cdt != DT_F16) return;
  ex.splitk_ws_bytes = igemm_splitk_ws_bytes(batch, rows_per_entry, n);
  ex.splitk_ws = (float*)tmp.get(ex.splitk_ws_bytes);
  ex.splitk_cnt = (unsigned*)tmp.get(kSplitkCounters * sizeof(unsigned));
  SDXL_HIP(hipMemsetAsync(ex.splitk_cnt, 0, kSplitkCounters * sizeof(unsigned), s));
}

__global__ void transpose_pad_kernel(const float* src, int lds_, int rows, int C, void* dst, int dt, int ldd) {
  // dst[c][r] = src[r][c]   (dst rows of ldd elements, caller zero-fills the padding)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int r = i / C, c = i - (size_t)r * C;
  const float v = src[(size_t)r * lds_ + c];
  if (dt == DT_F16) reinterpret_cast<_Float16*>(dst)[(size_t)c * ldd + r] = (_Float16)v;
  else reinterpret_cast<float*>(dst)[(size_t)c * ldd + r] = v;
}
__global__ void causal_mask_kernel(float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const int r = i / n, c = i - r * n;
  out[i] = c > r ? -INFINITY : 0.f;
}
}  // namespace

extern "C" {

const char* sdxl_last_error(void) { return g_err.c_str(); }
const char* sdxl_build_info(void) {
#ifdef SDXL_MEASURE
  return "sdxl_mi355 engine, HIP kernels for gfx950 (CDNA4, wave64, MFMA 32x32x16 f16 / 32x32x2 f32) [measure build: A/B variants + debug knobs]";
#else
  return "sdxl_mi355 engine, HIP kernels for gfx950 (CDNA4, wave64, MFMA 32x32x16 f16 / 32x32x2 f32)";
#endif
}

int sdxl_ctx_create(int device_id, sdxl_ctx** out) {
  API_BEGIN
  SDXL_REQUIRE(out != nullptr, "null out");
  int n = 0;
  SDXL_HIP(hipGetDeviceCount(&n));
  SDXL_REQUIRE(n > 0, "no HIP device visible: the MI355X engine has no CPU fallback");
  SDXL_REQUIRE(device_id >= 0 && device_id < n, "device id out of range");
  SDXL_HIP(hipSetDevice(device_id));
  sdxl_ctx* c = new sdxl_ctx();
  c->device = device_id;
  SDXL_HIP(hipStreamCreate(&c->stream));
  igemm_glds_init();
  attention_init();
  *out = c;
  API_END
}
int sdxl_debug_set(const char* key, int value) {
  API_BEGIN
  SDXL_REQUIRE(key != nullptr, "null key");
  if (std::strcmp(key, "igemm_variant") == 0) igemm_set_variant(value);
  else if (std::strcmp(key, "attn_variant") == 0) attention_set_variant(value);
  else if (std::strcmp(key, "igemm_epilogue_staged") == 0) igemm_set_epilogue_staged(value);
  else if (std::strcmp(key, "hl_weights_exact") == 0) igemm_set_hl_weights_exact(value);
  else if (std::strcmp(key, "igemm_wreg") == 0) igemm_set_wreg(value);
  else if (std::strcmp(key, "igemm_tsw") == 0) igemm_set_tsw(value);
  else if (std::strcmp(key, "igemm_warm") == 0) igemm_set_warm(value);
  else if (std::strcmp(key, "attn_xsplit") == 0) attention_set_xsplit(value);
  else if (std::strcmp(key, "splitk_wt") == 0) igemm_set_splitk_wt(value);
  else if (std::strcmp(key, "hl_demote") == 0) unet_set_hl_demote(value);
  else if (std::strcmp(key, "mix_classes") == 0) unet_set_mix_classes(value);
  else if (std::strcmp(key, "hl_tile96") == 0) igemm_set_hl_tile96(value);
  else if (std::strcmp(key, "wreg_xcd2d") == 0) igemm_set_wreg_xcd2d(value);
  else if (std::strcmp(key, "wide_db") == 0) igemm_set_wide_db(value);
#ifdef SDXL_MEASURE
  else if (std::strcmp(key, "igemm_unrolled") == 0) igemm_set_unrolled(value);
  else if (std::strcmp(key, "xa_vec64") == 0) igemm_set_xa_vec64(value);
  else if (std::strcmp(key, "no_cfg") == 0) g_debug_no_cfg = value != 0;
#endif
  else throw Error(std::string("unknown debug key ") + key);
  API_END
}
// host logic of the weight-warming schedule (WarmSeq::finish) on a synthetic launch sequence -- no device needed.  bytes[j] / host[j]: what entry j
// reads and whether its kernel can carry warming workgroups; warmed_by[j] receives the index of the entry that warms j (-1: nobody)
int sdxl_debug_warm_schedule(int n, const unsigned* bytes, const unsigned char* host, int* warmed_by) {
  API_BEGIN
  WarmSeq ws;
  for (int j = 0; j < n; ++j)
    ws.seq.push_back(WarmSeq::Item{reinterpret_cast<const void*>((uintptr_t)(j + 1) << 12), bytes[j], host[j] != 0, host[j] ? 14u << 20 : 0u, {nullptr, nullptr, nullptr}, {0u, 0u, 0u}});
  ws.finish();
  for (int j = 0; j < n; ++j) warmed_by[j] = -1;
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < 3; ++r)
      if (ws.seq[i].warm[r]) warmed_by[(int)(reinterpret_cast<uintptr_t>(ws.seq[i].warm[r]) >> 12) - 1] = i;
  API_END
}
#ifdef SDXL_MEASURE
// measure builds only: device buffer [workgroups][waves][sdxl_debug_timeline_words()] unsigned that the s_memtime-stamped kernel
// variants (igemm_measure.hip, variants 135 / 136 / 145) dump their per-wave phase stamps into; null switches it off
int sdxl_debug_timeline(void* device_buf) {
  API_BEGIN
  igemm_set_timeline(device_buf);
  API_END
}
int sdxl_debug_timeline_words(void) { return igemm_timeline_words(); }
// device buffer [workgroups][8][8] unsigned for the coarse s_memtime stamps of the wide (GEGLU) kernel; null switches it off
int sdxl_debug_attn_timeline(void* device_buf) {
  API_BEGIN
  attention_set_timeline(device_buf);
  API_END
}
int sdxl_debug_wreg_timeline(void* device_buf) {
  API_BEGIN
  igemm_set_wreg_timeline(device_buf);
  API_END
}
int sdxl_debug_wide_timeline(void* device_buf) {
  API_BEGIN
  igemm_set_wide_timeline(device_buf);
  API_END
}
#endif
int sdxl_bench_igemm(sdxl_ctx* ctx, void* stream, int B, int H, int W, int Cin, int Cout, int ksize, int geglu, int iters,
                     float* avg_ms) {
  // times the implicit-GEMM kernel alone on seeded random f16 data: conv ksize x ksize (pad ksize/2) or, with ksize = 1,
  // a linear over B*H*W rows.  Epilogue: bias (+ GEGLU when geglu != 0).  Used by tools/igemm_sweep.py.
  API_BEGIN
  SDXL_REQUIRE(ctx && avg_ms && iters > 0, "bad argument");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  Tmp tmp;
  Lin l; l.N = Cout; l.cin = Cin; l.ksize = ksize; l.K = Cin * ksize * ksize;
  l.Kpad = (int)round_up(l.K, 64); l.Npad = (int)round_up(Cout, 128);
  const size_t M = (size_t)B * H * W;
  float* wsrc = (float*)tmp.get((size_t)Cout * l.K * sizeof(float));
  float* bsrc = (float*)tmp.get((size_t)Cout * sizeof(float));
  float* xsrc = (float*)tmp.get(M * Cin * sizeof(float));
  void* wp = tmp.get((size_t)l.Npad * l.Kpad * 2);
  float* bp = (float*)tmp.get((size_t)l.Npad * sizeof(float));
  void* xi = tmp.get(M * Cin * 2);
  void* yo = tmp.get(M * Cout * 2);
  launch_synth_fill(wsrc, (size_t)Cout * l.K, 0x1234, 3.4641f / std::sqrt((float)l.K), 0.f, s);
  launch_synth_fill(bsrc, Cout, 0x99, 0.1f, 0.f, s);
  launch_synth_fill(xsrc, M * Cin, 0x777, 3.4641f, 0.f, s);
  if (ksize == 1) launch_pack_linear(wsrc, wp, DT_F16, l.K, Cout, l.Kpad, l.Npad, (geglu & 1) ? 1 : 0, 0, s);   // [K][N] random == fine
  else launch_pack_conv(wsrc, wp, DT_F16, Cout, Cin, ksize, l.Kpad, l.Npad, s);
  launch_pack_bias(bsrc, bp, Cout, l.Npad, (geglu & 1) ? 1 : 0, 0, s);
  launch_copy_rows(xsrc, DT_F32, Cin, xi, DT_F16, Cin, (int)M, Cin, s);
  l.w = wp; l.b = bp;
  tmp_wfrag(l, DT_F16, (geglu & 1) != 0, tmp, s);
  Exec ex; ex.s = s; ex.cdt = DT_F16; ex.sdt = DT_F16;
  give_splitk_ws(ex, tmp, B, H * W, Cout, s);
  const bool ln_in = (geglu & 2) != 0, st_out = (geglu & 4) != 0, cold = (geglu & 8) != 0;
  geglu &= 1;
  // cold mode: rotate through enough copies of the weight (> 256 MB Infinity Cache) that every launch streams it from HBM,
  // as in the model where each of the ~500 weights is touched once per step
  std::vector<void*> wcopies(1, wp);
  std::vector<const void*> wfcopies(1, l.wf);
  if (cold) {
    const size_t wbytes = (size_t)l.Npad * l.Kpad * 2;
    const int nc = (int)std::min<size_t>(96, (size_t)(320u << 20) / wbytes + 1);
    for (int i = 1; i < nc; ++i) {
      void* c = tmp.get(wbytes);
      SDXL_HIP(hipMemcpyAsync(c, wp, wbytes, hipMemcpyDeviceToDevice, s));
      wcopies.push_back(c);
      void* cf = nullptr;
      if (l.wf) { cf = tmp.get(wbytes); SDXL_HIP(hipMemcpyAsync(cf, l.wf, wbytes, hipMemcpyDeviceToDevice, s)); }
      wfcopies.push_back(cf);
    }
  }
  Epi e; e.act = geglu ? 1 : 0;
  if (ln_in) {   // timing of the LayerNorm-folded epilogue: plausible statistics (sum 0, sum^2 = 64 per slot), unit column sums
    SDXL_REQUIRE(ksize == 1 && Cin % 64 == 0, "ln bench needs a linear with K % 64 == 0");
    float* stat = (float*)tmp.get(M * (size_t)(Cin / 64) * 2 * sizeof(float));
    float* cs = (float*)tmp.get((size_t)l.Npad * sizeof(float));
    launch_synth_fill(stat, M * (size_t)(Cin / 64) * 2, 0x31, 0.5f, 64.0f, s);
    launch_synth_fill(cs, l.Npad, 0x32, 0.1f, 0.f, s);
    l.cs = cs; e.ln_stat = stat;
  }
  if (st_out) {
    SDXL_REQUIRE(!geglu && Cout % 64 == 0, "stat bench needs a plain N % 64 == 0 output");
    e.stat_out = (float*)tmp.get(M * (size_t)(Cout / 64) * 2 * sizeof(float));
  }
  const ConvGeom g{B, H, W, H, W, ksize, 1, ksize / 2, 0};
  const Act out(yo, geglu ? Cout / 2 : Cout, DT_F16);
  for (int i = 0; i < 3; ++i) run_conv(ex, l, Act(xi, Cin, DT_F16), Cin, g, out, e);
  hipEvent_t a, b;
  SDXL_HIP(hipEventCreate(&a)); SDXL_HIP(hipEventCreate(&b));
  SDXL_HIP(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) {
    l.w = wcopies[(size_t)i % wcopies.size()];
    l.wf = wfcopies[(size_t)i % wfcopies.size()];
    run_conv(ex, l, Act(xi, Cin, DT_F16), Cin, g, out, e);
  }
  SDXL_HIP(hipEventRecord(b, s));
  SDXL_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  SDXL_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *avg_ms = ms / iters;
  API_END
}
int sdxl_bench_attention(sdxl_ctx* ctx, void* stream, int B, int H, int Nq, int Nk, int iters, float* avg_ms) {
  API_BEGIN
  SDXL_REQUIRE(ctx && avg_ms && iters > 0, "bad argument");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  Tmp tmp;
  const int C = H * 64, npad = (int)round_up(Nk, 64);
  float* src = (float*)tmp.get((size_t)B * std::max(Nq, npad) * C * sizeof(float));
  void* q = tmp.get((size_t)B * Nq * C * 2);
  void* k = tmp.get((size_t)B * Nk * C * 2);
  void* vt = tmp.get((size_t)B * C * npad * 2);
  void* o = tmp.get((size_t)B * Nq * C * 2);
  launch_synth_fill(src, (size_t)B * Nq * C, 0x51, 3.4641f, 0.f, s);
  launch_copy_rows(src, DT_F32, C, q, DT_F16, C, B * Nq, C, s);
  launch_synth_fill(src, (size_t)B * Nk * C, 0x52, 3.4641f, 0.f, s);
  launch_copy_rows(src, DT_F32, C, k, DT_F16, C, B * Nk, C, s);
  launch_synth_fill(src, (size_t)B * C * npad, 0x53, 3.4641f, 0.f, s);
  launch_copy_rows(src, DT_F32, npad, vt, DT_F16, npad, B * C, npad, s);   // random V^T (padding columns included: timing only)
  AttnParams p{};
  p.Q = q; p.ldq = C; p.K = k; p.ldk = C; p.Vt = vt; p.vt_ld = npad; p.O = o; p.ldo = C;
  p.dt = DT_F16; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
  // workspace + tickets of the cross-workgroup key split (what the UNet hands its self-attention calls)
  p.xws = (float*)tmp.get(attention_xsplit_ws_bytes(B, H, Nq) + 256);
  p.xcnt = (unsigned*)tmp.get(attention_xsplit_counters(B, H, Nq) * sizeof(unsigned) + 256);
  launch_fill_zero(p.xcnt, attention_xsplit_counters(B, H, Nq) * sizeof(unsigned), s);
  for (int i = 0; i < 3; ++i) launch_attention_d64(p, s);
  hipEvent_t a, b;
  SDXL_HIP(hipEventCreate(&a)); SDXL_HIP(hipEventCreate(&b));
  SDXL_HIP(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) launch_attention_d64(p, s);
  SDXL_HIP(hipEventRecord(b, s));
  SDXL_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  SDXL_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  *avg_ms = ms / iters;
  API_END
}
void sdxl_ctx_destroy(sdxl_ctx* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}
int sdxl_ctx_synchronize(sdxl_ctx* c) {
  API_BEGIN
  use(c);
  SDXL_HIP(hipDeviceSynchronize());
  API_END
}

void sdxl_unet_config_base(sdxl_unet_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->adm_in_channels = 2816; c->in_channels = 4; c->out_channels = 4; c->model_channels = 320; c->n_levels = 3;
  const int m[3] = {1, 2, 4}, d[3] = {0, 2, 10};
  for (int i = 0; i < 3; ++i) { c->channel_mults[i] = m[i]; c->transformer_depths[i] = d[i]; }
  c->n_head_channels = 64; c->context_dim = 2048; c->is_refiner = 0;
}
void sdxl_unet_config_refiner(sdxl_unet_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->adm_in_channels = 2560; c->in_channels = 4; c->out_channels = 4; c->model_channels = 384; c->n_levels = 4;
  const int m[4] = {1, 2, 4, 4}, d[4] = {0, 4, 4, 4};
  for (int i = 0; i < 4; ++i) { c->channel_mults[i] = m[i]; c->transformer_depths[i] = d[i]; }
  c->n_head_channels = 64; c->context_dim = 1280; c->is_refiner = 1;
}
void sdxl_vae_config_default(sdxl_vae_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->n_blocks = 4;
  const int ei[4] = {128, 128, 256, 512}, eo[4] = {128, 256, 512, 512}, di[4] = {512, 512, 512, 256}, dn[4] = {512, 512, 256, 128};
  for (int i = 0; i < 4; ++i) { c->enc_in[i] = ei[i]; c->enc_out[i] = eo[i]; c->dec_in[i] = di[i]; c->dec_out[i] = dn[i]; }
  c->n_group = 32; c->enc_out_channels = 8; c->scale_factor = 0.13025;
}

int sdxl_unet_param_count(const sdxl_unet_config* cfg) {
  try { return (int)unet_param_specs(to_cfg(cfg)).size(); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int sdxl_unet_param_spec(const sdxl_unet_config* cfg, int index, const char** name, int* ndim, int64_t shape[4], int* kind,
                         float* sc, float* mean) {
  API_BEGIN
  static thread_local std::vector<ParamSpec> cache; static thread_local sdxl_unet_config key;
  if (cache.empty() || std::memcmp(&key, cfg, sizeof(key)) != 0) { cache = unet_param_specs(to_cfg(cfg)); key = *cfg; }
  return spec_out(cache, index, name, ndim, shape, kind, sc, mean);
  API_END
}
int sdxl_vae_param_count(const sdxl_vae_config* cfg, int encoder) {
  try { return (int)(encoder ? vae_encoder_param_specs(to_vcfg(cfg)) : vae_decoder_param_specs(to_vcfg(cfg))).size(); }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int sdxl_vae_param_spec(const sdxl_vae_config* cfg, int encoder, int index, const char** name, int* ndim, int64_t shape[4],
                        int* kind, float* sc, float* mean) {
  API_BEGIN
  const std::vector<ParamSpec> specs = encoder ? vae_encoder_param_specs(to_vcfg(cfg)) : vae_decoder_param_specs(to_vcfg(cfg));
  return spec_out(specs, index, name, ndim, shape, kind, sc, mean);
  API_END
}

// ---------------------------------------------------------------------------------------------- UNet
static int unet_create_impl(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, WeightSource& src, sdxl_unet** out) {
  SDXL_REQUIRE(ctx && out, "null argument");
  use(ctx);
  int cdt, sdt; dtypes(dtype, cdt, sdt);
  sdxl_unet* h = new sdxl_unet();
  h->ctx = ctx;
  try { h->u = new UNet(to_cfg(cfg), cdt, sdt, src, ctx->stream, mix_of(dtype)); } catch (...) { delete h; throw; }
  *out = h;
  return SDXL_OK;
}
int sdxl_unet_create(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, const float* weights_flat, sdxl_unet** out) {
  API_BEGIN
  SDXL_REQUIRE(weights_flat != nullptr, "null weights");
  const std::vector<ParamSpec> specs = unet_param_specs(to_cfg(cfg));
  FlatSource src(weights_flat, specs);
  return unet_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
int sdxl_unet_create_f16(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, const uint16_t* weights_flat_f16, sdxl_unet** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && weights_flat_f16 != nullptr, "null argument");
  use(ctx);
  const std::vector<ParamSpec> specs = unet_param_specs(to_cfg(cfg));
  FlatSourceF16 src(weights_flat_f16, specs);
  return unet_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
int sdxl_unet_create_synthetic(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, uint64_t seed, sdxl_unet** out) {
  API_BEGIN
  SyntheticSource src(seed);
  return unet_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
void sdxl_unet_destroy(sdxl_unet* u) {
  if (!u) return;
  if (u->owned) delete u->u;
  delete u;
}
int sdxl_unet_forward(sdxl_unet* u, void* stream, const float* x, const int32_t* timesteps, const float* context,
                      const float* label, int B, int H, int W, int n_ctx, float* out) {
  API_BEGIN
  SDXL_REQUIRE(u && x && timesteps && context && label && out, "null argument");
  use(u->ctx);
  u->u->forward_nchw(x, timesteps, context, n_ctx, label, B, H, W, out, pick(u->ctx, stream));
  API_END
}
int sdxl_unet_set_graph(sdxl_unet* u, int enabled) {
  API_BEGIN
  SDXL_REQUIRE(u != nullptr, "null argument");
  u->u->set_use_graph(enabled != 0);
  API_END
}
int sdxl_unet_set_split_cfg(sdxl_unet* u, int enabled, int release_offset) {
  API_BEGIN
  SDXL_REQUIRE(u != nullptr && release_offset >= 0, "bad argument");
  u->u->set_split_cfg(enabled != 0, release_offset);
  API_END
}
int sdxl_unet_set_fused_cross_attention(sdxl_unet* u, int enabled) {
  API_BEGIN
  SDXL_REQUIRE(u != nullptr, "bad argument");
  u->u->set_fused_cross_attention(enabled != 0);
  API_END
}
int sdxl_unet_set_gn_from_producer(sdxl_unet* u, int enabled) {
  API_BEGIN
  SDXL_REQUIRE(u != nullptr, "bad argument");
  u->u->set_gn_from_producer(enabled != 0);
  API_END
}
int sdxl_unet_mix_classes(sdxl_unet* u, int* classes_out) {
  API_BEGIN
  SDXL_REQUIRE(u != nullptr && classes_out != nullptr, "null argument");
  *classes_out = u->u->mix_classes();
  API_END
}
int sdxl_unet_weight_arena(sdxl_unet* u, void** base, size_t* bytes) {
  API_BEGIN
  SDXL_REQUIRE(u && base && bytes, "null argument");
  *base = u->u->weight_base(); *bytes = u->u->weight_bytes();
  API_END
}

// ---------------------------------------------------------------------------------------------- attention op
int sdxl_qkv_attention(sdxl_ctx* ctx, void* stream, const float* q, const float* k, const float* v, const float* mask, int B,
                       int Nq, int Nk, int n_state, int n_head, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && q && k && v && out, "null argument");
  SDXL_REQUIRE(n_head > 0 && n_state % n_head == 0, "State size must be a multiple of head size");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt);
  const int d = n_state / n_head;
  if (cdt == DT_HL) {
    // split-operand mode: Q / O stay fp32, K and V^T go through the HL16 format of the split GEMMs (attn_d64_hl_kernel)
    SDXL_REQUIRE(d == 64 && !mask, "sdxl_qkv_attention: the split-operand mode covers unmasked head-dim-64 attention");
    const int npad = (int)round_up(Nk, 64);
    Tmp tmp;
    void* kh = tmp.get((size_t)B * Nk * n_state * 4);
    float* vt32 = (float*)tmp.get((size_t)B * n_state * npad * 4);
    void* vth = tmp.get((size_t)B * n_state * npad * 4);
    launch_f32_to_hl(k, n_state, kh, n_state, (size_t)B * Nk, n_state, s);
    launch_fill_zero(vt32, (size_t)B * n_state * npad * 4, s);
    for (int b = 0; b < B; ++b) {
      const size_t tot = (size_t)Nk * n_state;
      hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v + (size_t)b * Nk * n_state, n_state,
                         Nk, n_state, (char*)vt32 + (size_t)b * n_state * npad * 4, DT_F32, npad);
    }
    launch_f32_to_hl(vt32, npad, vth, npad, (size_t)B * n_state, npad, s);
    AttnParams p{};
    p.Q = q; p.ldq = n_state; p.K = kh; p.ldk = n_state; p.Vt = vth; p.vt_ld = npad; p.O = out; p.ldo = n_state;
    p.dt = DT_HL; p.B = B; p.H = n_head; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
    SDXL_REQUIRE(launch_attention_d64_hl(p, s), "split-operand attention kernel refused an aligned shape");
    SDXL_HIP(hipStreamSynchronize(s));
    return 0;
  }
  no_split(cdt, "sdxl_qkv_attention");
  const size_t es = dt_size(cdt);
  Tmp tmp;
  if (d == 64 || (d == 512 && cdt == DT_F16 && !mask)) {
    const int npad = (int)round_up(Nk, 64);
    void* qd = tmp.get((size_t)B * Nq * n_state * es);
    void* kd = tmp.get((size_t)B * Nk * n_state * es);
    void* vt = tmp.get((size_t)B * n_state * npad * es);
    void* od = tmp.get((size_t)B * Nq * n_state * es);
    launch_copy_rows(q, DT_F32, n_state, qd, cdt, n_state, B * Nq, n_state, s);
    launch_copy_rows(k, DT_F32, n_state, kd, cdt, n_state, B * Nk, n_state, s);
    launch_fill_zero(vt, (size_t)B * n_state * npad * es, s);
    for (int b = 0; b < B; ++b) {
      const size_t tot = (size_t)Nk * n_state;
      hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v + (size_t)b * Nk * n_state, n_state,
                         Nk, n_state, (char*)vt + (size_t)b * n_state * npad * es, cdt, npad);
    }
    AttnParams p{};
    p.Q = qd; p.ldq = n_state; p.K = kd; p.ldk = n_state; p.Vt = vt; p.vt_ld = npad; p.O = od; p.ldo = n_state;
    p.dt = cdt; p.B = B; p.H = n_head; p.Nq = Nq; p.Nk = Nk; p.scale = (float)(1.0 / std::sqrt((double)d)); p.mask = mask; p.ldmask = Nk;
    if (d == 64 && cdt == DT_F16 && !mask) {   // workspace + tickets of the cross-workgroup key split (what the UNet hands its self-attention calls)
      p.xws = (float*)tmp.get(attention_xsplit_ws_bytes(B, n_head, Nq) + 256);
      p.xcnt = (unsigned*)tmp.get(attention_xsplit_counters(B, n_head, Nq) * sizeof(unsigned) + 256);
      launch_fill_zero(p.xcnt, attention_xsplit_counters(B, n_head, Nq) * sizeof(unsigned), s);
    }
    if (d == 64) launch_attention_d64(p, s);
    else SDXL_REQUIRE(launch_attention_hd512(p, s), "wide-head attention kernel refused an aligned f16 shape");
    launch_copy_rows(od, cdt, n_state, out, DT_F32, n_state, B * Nq, n_state, s);
  } else {
    // generic head dim: QK^T GEMM -> row softmax -> PV GEMM per (batch, head)
    const int kt = cdt == DT_F16 ? 64 : 32;
    const int dpad = (int)round_up(d, kt), kpad = (int)round_up(Nk, kt);
    const int rows_k = (int)round_up(Nk, 128), rows_v = (int)round_up(d, 128);
    void* qh = tmp.get((size_t)Nq * dpad * es);
    void* kh = tmp.get((size_t)rows_k * dpad * es);
    void* vt = tmp.get((size_t)rows_v * kpad * es);
    float* S = (float*)tmp.get((size_t)Nq * Nk * sizeof(float));
    void* P = tmp.get((size_t)Nq * kpad * es);
    launch_fill_zero(qh, (size_t)Nq * dpad * es, s);
    launch_fill_zero(kh, (size_t)rows_k * dpad * es, s);
    launch_fill_zero(vt, (size_t)rows_v * kpad * es, s);
    Exec ex; ex.s = s; ex.cdt = cdt; ex.sdt = sdt;
    const float scale = (float)(1.0 / std::sqrt((double)d));
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < n_head; ++h) {
        const float* qs = q + (size_t)b * Nq * n_state + h * d;
        const float* ks = k + (size_t)b * Nk * n_state + h * d;
        const float* vs = v + (size_t)b * Nk * n_state + h * d;
        launch_copy_rows(qs, DT_F32, n_state, qh, cdt, dpad, Nq, d, s);
        launch_copy_rows(ks, DT_F32, n_state, kh, cdt, dpad, Nk, d, s);
        const size_t tot = (size_t)Nk * d;
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, vs, n_state, Nk, d, vt, cdt, kpad);
        Lin lk; lk.w = kh; lk.N = Nk; lk.K = dpad; lk.Kpad = dpad; lk.Npad = rows_k; lk.cin = dpad;
        run_linear(ex, lk, Act(qh, dpad, cdt), Nq, Act(S, Nk, DT_F32));
        launch_softmax_rows(S, Nk, P, cdt, kpad, Nq, Nk, kpad, scale, mask, Nk, Nq, s);
        Lin lv; lv.w = vt; lv.N = d; lv.K = Nk; lv.Kpad = kpad; lv.Npad = rows_v; lv.cin = Nk;
        run_linear(ex, lv, Act(P, kpad, cdt), Nq, Act(out + (size_t)b * Nq * n_state + h * d, n_state, DT_F32));
      }
  }
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}
int sdxl_attn_decoder_mask(sdxl_ctx* ctx, void* stream, int n, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && out && n > 0, "bad argument");
  use(ctx);
  hipLaunchKernelGGL(causal_mask_kernel, dim3((n * n + 255) / 256), dim3(256), 0, pick(ctx, stream), out, n);
  API_END
}

// ---------------------------------------------------------------------------------------------- Embedder (CLIP text encoders)
void sdxl_clip_config_clip_l(sdxl_clip_config* c) { if (c) *c = sdxl_clip_config{49408, 768, 768, 12, 77, 12, 1}; }
void sdxl_clip_config_open_clip_bigg(sdxl_clip_config* c) { if (c) *c = sdxl_clip_config{49408, 1280, 1280, 20, 77, 32, 0}; }
int sdxl_clip_param_count(const sdxl_clip_config* cfg) {
  try { return (int)clip_param_specs(to_ccfg(cfg)).size(); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int sdxl_clip_param_spec(const sdxl_clip_config* cfg, int index, const char** name, int* ndim, int64_t shape[4], int* kind,
                         float* sc, float* mean) {
  API_BEGIN
  static thread_local std::vector<ParamSpec> cache; static thread_local sdxl_clip_config key;
  if (cache.empty() || std::memcmp(&key, cfg, sizeof(key)) != 0) { cache = clip_param_specs(to_ccfg(cfg)); key = *cfg; }
  return spec_out(cache, index, name, ndim, shape, kind, sc, mean);
  API_END
}
static int clip_create_impl(sdxl_ctx* ctx, const sdxl_clip_config* cfg, int dtype, WeightSource& src, sdxl_clip** out) {
  SDXL_REQUIRE(ctx && out, "bad argument");
  use(ctx);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt); no_split(cdt, "the CLIP text encoders");
  sdxl_clip* h = new sdxl_clip();
  h->ctx = ctx;
  try { h->c = new ClipText(to_ccfg(cfg), cdt, sdt, src, ctx->stream); } catch (...) { delete h; throw; }
  *out = h;
  return SDXL_OK;
}
int sdxl_clip_create(sdxl_ctx* ctx, const sdxl_clip_config* cfg, int dtype, const float* weights_flat, sdxl_clip** out) {
  API_BEGIN
  SDXL_REQUIRE(weights_flat != nullptr, "null weights");
  const std::vector<ParamSpec> specs = clip_param_specs(to_ccfg(cfg));
  FlatSource src(weights_flat, specs);
  return clip_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
int sdxl_clip_create_f16(sdxl_ctx* ctx, const sdxl_clip_config* cfg, int dtype, const uint16_t* weights_flat_f16, sdxl_clip** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && weights_flat_f16 != nullptr, "null argument");
  use(ctx);
  const std::vector<ParamSpec> specs = clip_param_specs(to_ccfg(cfg));
  FlatSourceF16 src(weights_flat_f16, specs);
  return clip_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
int sdxl_clip_create_synthetic(sdxl_ctx* ctx, const sdxl_clip_config* cfg, int dtype, uint64_t seed, sdxl_clip** out) {
  API_BEGIN
  SyntheticSource src(seed);
  return clip_create_impl(ctx, cfg, dtype, src, out);
  API_END
}
void sdxl_clip_destroy(sdxl_clip* c) {
  if (!c) return;
  delete c->c;
  delete c;
}
int sdxl_clip_forward_hidden(sdxl_clip* c, void* stream, const int32_t* tokens, int n, int seq, int hidden_idx, float* out) {
  API_BEGIN
  SDXL_REQUIRE(c && tokens && out, "null argument");
  use(c->ctx);
  c->c->forward_hidden(tokens, n, seq, hidden_idx, out, pick(c->ctx, stream));
  API_END
}
int sdxl_clip_forward_hidden_pooled(sdxl_clip* c, void* stream, const int32_t* tokens, int n, int seq, int hidden_idx,
                                    float* out_hidden, float* out_pooled) {
  API_BEGIN
  SDXL_REQUIRE(c && tokens && out_hidden && out_pooled, "null argument");
  use(c->ctx);
  c->c->forward_hidden_pooled(tokens, n, seq, hidden_idx, out_hidden, out_pooled, pick(c->ctx, stream));
  API_END
}
int sdxl_conditioning_embedding(sdxl_ctx* ctx, void* stream, const float* pooled, int n, int E, const int32_t* values, int w,
                                int dim, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && pooled && values && out && n > 0 && E > 0 && w > 0 && dim > 0 && dim % 2 == 0, "bad argument");
  use(ctx);
  launch_conditioning_embedding(pooled, E, values, w, dim, out, n, pick(ctx, stream));
  API_END
}
int sdxl_clip_weight_arena(sdxl_clip* c, void** base, size_t* bytes) {
  API_BEGIN
  SDXL_REQUIRE(c && base && bytes, "null argument");
  *base = c->c->weight_base(); *bytes = c->c->weight_bytes();
  API_END
}

// ---------------------------------------------------------------------------------------------- Diffuser
static int diffuser_create_impl(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, WeightSource& src, const float* alphas,
                                int n_train, sdxl_diffuser** out) {
  SDXL_REQUIRE(ctx && out && alphas && n_train > 0, "bad argument");
  use(ctx);
  int cdt, sdt; dtypes(dtype, cdt, sdt);
  sdxl_diffuser* h = new sdxl_diffuser();
  h->ctx = ctx;
  try { h->d = new Diffuser(to_cfg(cfg), cdt, sdt, src, alphas, n_train, ctx->stream, mix_of(dtype)); } catch (...) { delete h; throw; }
  h->view.ctx = ctx; h->view.u = &h->d->unet(); h->view.owned = false;
  *out = h;
  return SDXL_OK;
}
int sdxl_diffuser_create(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, const float* weights_flat, const float* alphas,
                         int n_train, sdxl_diffuser** out) {
  API_BEGIN
  SDXL_REQUIRE(weights_flat != nullptr, "null weights");
  const std::vector<ParamSpec> specs = unet_param_specs(to_cfg(cfg));
  FlatSource src(weights_flat, specs);
  return diffuser_create_impl(ctx, cfg, dtype, src, alphas, n_train, out);
  API_END
}
int sdxl_diffuser_create_f16(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, const uint16_t* weights_flat_f16,
                             const float* alphas, int n_train, sdxl_diffuser** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && weights_flat_f16 != nullptr, "null argument");
  use(ctx);
  const std::vector<ParamSpec> specs = unet_param_specs(to_cfg(cfg));
  FlatSourceF16 src(weights_flat_f16, specs);
  return diffuser_create_impl(ctx, cfg, dtype, src, alphas, n_train, out);
  API_END
}
int sdxl_diffuser_create_synthetic(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, uint64_t seed, const float* alphas,
                                   int n_train, sdxl_diffuser** out) {
  API_BEGIN
  SyntheticSource src(seed);
  return diffuser_create_impl(ctx, cfg, dtype, src, alphas, n_train, out);
  API_END
}
int sdxl_diffuser_create_empty(sdxl_ctx* ctx, const sdxl_unet_config* cfg, int dtype, const float* alphas, int n_train,
                               sdxl_diffuser** out) {
  API_BEGIN
  NullSource src;
  return diffuser_create_impl(ctx, cfg, dtype, src, alphas, n_train, out);
  API_END
}
int sdxl_vae_create_empty(sdxl_ctx* ctx, const sdxl_vae_config* cfg, int dtype, int with_encoder, sdxl_vae** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && out, "bad argument");
  use(ctx);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt);
  NullSource src;
  sdxl_vae* h = new sdxl_vae();
  h->ctx = ctx;
  try { h->v = new Vae(to_vcfg(cfg), cdt, &src, with_encoder ? &src : nullptr, ctx->stream); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int sdxl_unet_profile(sdxl_unet* u, void* stream, int B, int H, int W, float class_ms[5], int class_launches[5],
                      double class_flops[5]) {
  API_BEGIN
  SDXL_REQUIRE(u && class_ms && class_launches && class_flops, "null argument");
  use(u->ctx);
  u->u->profile(B, H, W, class_ms, class_launches, class_flops, pick(u->ctx, stream));
  API_END
}
int sdxl_unet_eager_forward_ms(sdxl_unet* u, void* stream, int B, int H, int W, float* ms_out) {
  API_BEGIN
  SDXL_REQUIRE(u && ms_out, "null argument");
  use(u->ctx);
  *ms_out = u->u->eager_ms(B, H, W, pick(u->ctx, stream));
  API_END
}
void sdxl_diffuser_destroy(sdxl_diffuser* d) {
  if (!d) return;
  delete d->d;
  delete d;
}
sdxl_unet* sdxl_diffuser_unet(sdxl_diffuser* d) { return d ? &d->view : nullptr; }

static Conditioning to_cond(const sdxl_conditioning* c) {
  SDXL_REQUIRE(c != nullptr, "null conditioning");
  Conditioning o;
  o.unconditional_context_full = c->unconditional_context_full;
  o.unconditional_context_open_clip = c->unconditional_context_open_clip;
  o.context_full = c->context_full; o.context_open_clip = c->context_open_clip;
  o.unconditional_channel_context = c->unconditional_channel_context;
  o.unconditional_channel_context_refiner = c->unconditional_channel_context_refiner;
  o.channel_context = c->channel_context; o.channel_context_refiner = c->channel_context_refiner;
  o.n = c->n; o.n_ctx = c->n_ctx; o.height = c->height; o.width = c->width;
  SDXL_REQUIRE(o.n >= 1 && o.n_ctx >= 1 && o.height >= 8 && o.width >= 8 && o.height % 8 == 0 && o.width % 8 == 0,
               "bad conditioning shape");
  return o;
}
int sdxl_sample_latent(sdxl_diffuser* d, void* stream, const sdxl_conditioning* cond, double cfg, int n_steps,
                       const float* noise0, float* out) {
  API_BEGIN
  SDXL_REQUIRE(d && noise0 && out, "null argument");
  use(d->ctx);
  d->d->sample_latent(to_cond(cond), cfg, n_steps, noise0, out, pick(d->ctx, stream));
  API_END
}
int sdxl_sample_latent_with_inpainting(sdxl_diffuser* d, void* stream, const sdxl_conditioning* cond, double cfg, int n_steps,
                                       const float* reference, const uint8_t* mask, const float* noise0,
                                       const float* step_noise, float* out) {
  API_BEGIN
  SDXL_REQUIRE(d && noise0 && out, "null argument");
  use(d->ctx);
  d->d->sample_latent_inpaint(to_cond(cond), cfg, n_steps, reference, mask, noise0, step_noise, out, pick(d->ctx, stream));
  API_END
}
int sdxl_refine_latent(sdxl_diffuser* d, void* stream, const float* latent, const sdxl_conditioning* cond, double cfg,
                       int step_start, int n_steps, const float* noise, float* out) {
  API_BEGIN
  SDXL_REQUIRE(d && latent && noise && out, "null argument");
  use(d->ctx);
  d->d->refine_latent(latent, to_cond(cond), cfg, step_start, n_steps, noise, out, pick(d->ctx, stream));
  API_END
}
int sdxl_step_count(int n_steps, int step_start, int n_train) {
  try { return (int)Diffuser::step_schedule(n_steps, step_start, n_train).size(); }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int sdxl_diffuser_enable_step_timing(sdxl_diffuser* d, int enabled) {
  API_BEGIN
  SDXL_REQUIRE(d != nullptr, "null argument");
  d->d->time_steps = enabled != 0;
  API_END
}
int sdxl_diffuser_set_trace(sdxl_diffuser* d, float* trace_dev, int capacity_steps) {
  API_BEGIN
  SDXL_REQUIRE(d && (trace_dev || capacity_steps == 0) && capacity_steps >= 0, "bad argument");
  d->d->trace = capacity_steps > 0 ? trace_dev : nullptr;
  d->d->trace_cap = capacity_steps;
  API_END
}
int sdxl_diffuser_step_times(sdxl_diffuser* d, float* out_ms, int capacity) {
  if (!d || !out_ms) return -1;
  const int n = (int)d->d->step_ms.size() < capacity ? (int)d->d->step_ms.size() : capacity;
  for (int i = 0; i < n; ++i) out_ms[i] = d->d->step_ms[i];
  return n;
}

// ---------------------------------------------------------------------------------------------- VAE
int sdxl_vae_create_f16(sdxl_ctx* ctx, const sdxl_vae_config* cfg, int dtype, const uint16_t* dec_w, const uint16_t* enc_w, sdxl_vae** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && out && (dec_w || enc_w), "bad argument");
  use(ctx);
  int cdt; vae_dtype(dtype, cdt);
  const VaeCfg vc = to_vcfg(cfg);
  const std::vector<ParamSpec> ds = vae_decoder_param_specs(vc), es = vae_encoder_param_specs(vc);
  std::unique_ptr<FlatSourceF16> d, e;
  if (dec_w) d.reset(new FlatSourceF16(dec_w, ds));
  if (enc_w) e.reset(new FlatSourceF16(enc_w, es));
  sdxl_vae* h = new sdxl_vae();
  h->ctx = ctx;
  try { h->v = new Vae(vc, cdt, d.get(), e.get(), ctx->stream); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int sdxl_vae_create(sdxl_ctx* ctx, const sdxl_vae_config* cfg, int dtype, const float* dec_w, const float* enc_w, sdxl_vae** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && out && (dec_w || enc_w), "bad argument");
  use(ctx);
  int cdt; vae_dtype(dtype, cdt);
  const VaeCfg vc = to_vcfg(cfg);
  const std::vector<ParamSpec> ds = vae_decoder_param_specs(vc), es = vae_encoder_param_specs(vc);
  std::unique_ptr<FlatSource> d, e;
  if (dec_w) d.reset(new FlatSource(dec_w, ds));
  if (enc_w) e.reset(new FlatSource(enc_w, es));
  sdxl_vae* h = new sdxl_vae();
  h->ctx = ctx;
  try { h->v = new Vae(vc, cdt, d.get(), e.get(), ctx->stream); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int sdxl_vae_create_synthetic(sdxl_ctx* ctx, const sdxl_vae_config* cfg, int dtype, uint64_t seed, int with_encoder, sdxl_vae** out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && out, "bad argument");
  use(ctx);
  int cdt; vae_dtype(dtype, cdt);
  SyntheticSource src(seed);
  sdxl_vae* h = new sdxl_vae();
  h->ctx = ctx;
  try { h->v = new Vae(to_vcfg(cfg), cdt, &src, with_encoder ? &src : nullptr, ctx->stream); } catch (...) { delete h; throw; }
  *out = h;
  API_END
}
void sdxl_vae_destroy(sdxl_vae* v) {
  if (!v) return;
  delete v->v;
  delete v;
}
int sdxl_vae_decode_latent(sdxl_vae* v, void* stream, const float* latent, int n, int h, int w, float* out) {
  API_BEGIN
  SDXL_REQUIRE(v && latent && out, "null argument");
  use(v->ctx);
  v->v->decode_nchw(latent, n, h, w, out, pick(v->ctx, stream));
  API_END
}
int sdxl_latent_to_image(sdxl_vae* v, void* stream, const float* latent, int n, int h, int w, uint8_t* out) {
  API_BEGIN
  SDXL_REQUIRE(v && latent && out, "null argument");
  use(v->ctx);
  v->v->latent_to_image(latent, n, h, w, out, pick(v->ctx, stream));
  API_END
}
int sdxl_vae_encode_image(sdxl_vae* v, void* stream, const float* image, int n, int H, int W, float* out) {
  API_BEGIN
  SDXL_REQUIRE(v && image && out, "null argument");
  SDXL_REQUIRE(H % 8 == 0 && W % 8 == 0, "image size must be a multiple of 8");
  use(v->ctx);
  v->v->encode_nchw(image, n, H, W, out, pick(v->ctx, stream));
  API_END
}
int sdxl_image_to_latent(sdxl_vae* v, void* stream, const uint8_t* image, int n, int H, int W, float* out) {
  API_BEGIN
  SDXL_REQUIRE(v && image && out, "null argument");
  SDXL_REQUIRE(H % 8 == 0 && W % 8 == 0, "image size must be a multiple of 8");
  use(v->ctx);
  v->v->image_to_latent(image, n, H, W, out, pick(v->ctx, stream));
  API_END
}
int sdxl_vae_weight_arena(sdxl_vae* v, void** base, size_t* bytes) {
  API_BEGIN
  SDXL_REQUIRE(v && base && bytes, "null argument");
  *base = v->v->weight_base(); *bytes = v->v->weight_bytes();
  API_END
}

// ---------------------------------------------------------------------------------------------- weight broadcast (comm.cpp)
void sdxl_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }
int sdxl_unet_bcast_weights(sdxl_comm* comm, sdxl_unet* u, int root) {
  if (!comm || !u) return fail(SDXL_ERR_RUNTIME, "null argument");
  return sdxl_bcast_buffer(comm, nullptr, u->u->weight_base(), u->u->weight_bytes(), root);
}
int sdxl_vae_bcast_weights(sdxl_comm* comm, sdxl_vae* v, int root) {
  if (!comm || !v) return fail(SDXL_ERR_RUNTIME, "null argument");
  return sdxl_bcast_buffer(comm, nullptr, v->v->weight_base(), v->v->weight_bytes(), root);
}
int sdxl_clip_bcast_weights(sdxl_comm* comm, sdxl_clip* c, int root) {
  if (!comm || !c) return fail(SDXL_ERR_RUNTIME, "null argument");
  return sdxl_bcast_buffer(comm, nullptr, c->c->weight_base(), c->c->weight_bytes(), root);
}

// ---------------------------------------------------------------------------------------------- single ops
int sdxl_group_norm(sdxl_ctx* ctx, void* stream, const float* x, const float* gamma, const float* beta, int B, int C, int HW,
                    int n_group, float eps, int silu, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && x && gamma && beta && out, "null argument");
  SDXL_REQUIRE(n_group > 0 && C % n_group == 0, "The number of channels must be divisible by the number of groups");
  SDXL_REQUIRE(C % 8 == 0 && n_group <= 256, "unsupported GroupNorm shape");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt); no_split(cdt, "sdxl_group_norm");
  Tmp tmp;
  void* xi = tmp.get((size_t)B * HW * C * dt_size(sdt));
  void* yo = tmp.get((size_t)B * HW * C * dt_size(cdt));
  float* part = (float*)tmp.get(groupnorm_workspace_floats(B, n_group) * sizeof(float));
  launch_nchw_to_nhwc(x, C * HW, xi, sdt, B, C, HW, C, 1.0f, s);
  GroupNormParams p{};
  p.X = xi; p.x_dt = sdt; p.ldx = C; p.Y = yo; p.y_dt = cdt; p.ldy = C; p.gamma = gamma; p.beta = beta; p.partial = part;
  p.B = B; p.HW = HW; p.C = C; p.G = n_group; p.eps = eps; p.silu = silu;
  launch_groupnorm(p, s);
  launch_nhwc_to_nchw(yo, cdt, C, out, B, C, HW, 1.0f, s);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}
int sdxl_layer_norm(sdxl_ctx* ctx, void* stream, const float* x, const float* gamma, const float* beta, int rows, int C,
                    float eps, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && x && gamma && beta && out, "null argument");
  SDXL_REQUIRE(C % 8 == 0, "unsupported LayerNorm width");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt); no_split(cdt, "sdxl_layer_norm");
  Tmp tmp;
  void* xi = tmp.get((size_t)rows * C * dt_size(sdt));
  void* yo = tmp.get((size_t)rows * C * dt_size(cdt));
  launch_copy_rows(x, DT_F32, C, xi, sdt, C, rows, C, s);
  LayerNormParams p{};
  p.X = xi; p.x_dt = sdt; p.ldx = C; p.Y = yo; p.y_dt = cdt; p.ldy = C; p.gamma = gamma; p.beta = beta; p.rows = rows; p.C = C; p.eps = eps;
  launch_layernorm(p, s);
  launch_copy_rows(yo, cdt, C, out, DT_F32, C, rows, C, s);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}
int sdxl_conv2d(sdxl_ctx* ctx, void* stream, const float* x, const float* weight, const float* bias, int B, int Cin, int H,
                int W, int Cout, int ksize, int stride, int pad, int upsample, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && x && weight && out, "null argument");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt;
  if (dtype == SDXL_DTYPE_F32_SPLIT) { cdt = DT_HL; sdt = DT_HL; }   // split-operand GEMM as an operator (the VAE's precision): HL16 operands
  else { no_mix(dtype); dtypes(dtype, cdt, sdt); }
  SDXL_REQUIRE(cdt != DT_HL || Cin % 32 == 0, "SDXL_DTYPE_F32_SPLIT convolutions need Cin % 32 == 0");
  const int Hs = upsample ? 2 * H : H, Ws = upsample ? 2 * W : W;
  const int Ho = (Hs + 2 * pad - ksize) / stride + 1, Wo = (Ws + 2 * pad - ksize) / stride + 1;
  const int kt = cdt == DT_F16 ? 64 : 32;
  Lin l; l.N = Cout; l.cin = Cin; l.ksize = ksize; l.K = Cin * ksize * ksize;
  l.Kpad = (int)round_up(l.K, kt); l.Npad = (int)round_up(Cout, 128);
  Tmp tmp;
  void* wp = tmp.get((size_t)l.Npad * l.Kpad * dt_size(cdt));
  float* bp = (float*)tmp.get((size_t)l.Npad * sizeof(float));
  void* xi = tmp.get((size_t)B * H * W * Cin * dt_size(sdt));
  float* yo = (float*)tmp.get((size_t)B * Ho * Wo * Cout * sizeof(float));
  float wscale = 1.f;
  if (cdt == DT_HL) {     // power-of-two weight scale from max |w|, exactly as WeightBuilder::conv
    float* sc = (float*)tmp.get(256);
    launch_absmax(weight, (size_t)Cout * l.K, sc, s);
    float h = 0.f;
    SDXL_HIP(hipMemcpyAsync(&h, sc, sizeof(float), hipMemcpyDeviceToHost, s));
    SDXL_HIP(hipStreamSynchronize(s));
    int e = 0;
    if (h > 0.f && std::isfinite(h)) { (void)std::frexp(h, &e); e = 14 - e; }
    e = e > 24 ? 24 : (e < -24 ? -24 : e);
    wscale = std::ldexp(1.0f, e);
    const float inv = 1.0f / wscale;
    SDXL_HIP(hipMemcpyAsync(sc, &inv, sizeof(float), hipMemcpyHostToDevice, s));
    SDXL_HIP(hipStreamSynchronize(s));
    launch_f16_exact(weight, (size_t)Cout * l.K, wscale, sc + 1, s);      // exact-f16 weights: the kernel leaves out the w_lo MFMAs
    l.acc_scale = sc;
  }
  launch_pack_conv(weight, wp, cdt, Cout, Cin, ksize, l.Kpad, l.Npad, s, wscale);
  launch_pack_bias(bias, bp, Cout, l.Npad, 0, 0, s);
  l.w = wp; l.b = bp;
  tmp_wfrag(l, cdt, false, tmp, s);
  Act xa(xi, Cin, sdt);
  if (cdt == DT_HL && l.acc_scale) {      // range-safe conversion, as the models convert their fp32 stream tensors (hl_operand)
    float* x32 = (float*)tmp.get((size_t)B * H * W * Cin * sizeof(float));
    float* sc = (float*)tmp.get(hl_scale_floats(B) * sizeof(float));
    launch_nchw_to_nhwc(x, Cin * H * W, x32, DT_F32, B, Cin, H * W, Cin, 1.0f, s);
    launch_f32_to_hl_scaled(x32, Cin, xi, Cin, (size_t)B * H * W, Cin, sc, s, B);      // one factor per batch entry, as the models' hl_operand
    xa.a_scale = hl_scale_inv(sc, B); xa.a_scale_n = B;
  } else launch_nchw_to_nhwc(x, Cin * H * W, xi, sdt, B, Cin, H * W, Cin, 1.0f, s);     // (st_f handles the HL16 layout: Cin % 16 == 0 rows)
  Exec ex; ex.s = s; ex.cdt = cdt; ex.sdt = sdt;
  give_splitk_ws(ex, tmp, B, Ho * Wo, Cout, s);
  run_conv(ex, l, xa, Cin, ConvGeom{B, H, W, Ho, Wo, ksize, stride, pad, upsample ? 1 : 0}, Act(yo, Cout, DT_F32));
  launch_nhwc_to_nchw(yo, DT_F32, Cout, out, B, Cout, Ho * Wo, 1.0f, s);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}
int sdxl_linear(sdxl_ctx* ctx, void* stream, const float* x, const float* weight, const float* bias, int M, int K, int N,
                int geglu, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && x && weight && out, "null argument");
  SDXL_REQUIRE(!geglu || (N % 32 == 0), "GEGLU width must be a multiple of 32");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt;
  if (dtype == SDXL_DTYPE_F32_SPLIT) { cdt = DT_HL; sdt = DT_HL; }   // split-operand GEMM as an operator: HL16 operands (incl. the GEGLU epilogue)
  else { no_mix(dtype); dtypes(dtype, cdt, sdt); }
  SDXL_REQUIRE(cdt != DT_HL || K % 32 == 0, "SDXL_DTYPE_F32_SPLIT linear layers need K % 32 == 0");
  const int kt = cdt == DT_F16 ? 64 : 32;
  Lin l; l.N = N; l.K = K; l.cin = K; l.ksize = 1; l.Kpad = (int)round_up(K, kt); l.Npad = (int)round_up(N, 128);
  Tmp tmp;
  void* wp = tmp.get((size_t)l.Npad * l.Kpad * dt_size(cdt));
  float* bp = (float*)tmp.get((size_t)l.Npad * sizeof(float));
  void* xi = tmp.get((size_t)M * K * dt_size(sdt));
  float wscale = 1.f;
  if (cdt == DT_HL) {     // power-of-two weight scale from max |w|, exactly as WeightBuilder::linear
    float* sc = (float*)tmp.get(256);
    launch_absmax(weight, (size_t)K * N, sc, s);
    float h = 0.f;
    SDXL_HIP(hipMemcpyAsync(&h, sc, sizeof(float), hipMemcpyDeviceToHost, s));
    SDXL_HIP(hipStreamSynchronize(s));
    int e = 0;
    if (h > 0.f && std::isfinite(h)) { (void)std::frexp(h, &e); e = 14 - e; }
    e = e > 24 ? 24 : (e < -24 ? -24 : e);
    wscale = std::ldexp(1.0f, e);
    const float inv = 1.0f / wscale;
    SDXL_HIP(hipMemcpyAsync(sc, &inv, sizeof(float), hipMemcpyHostToDevice, s));
    SDXL_HIP(hipStreamSynchronize(s));
    launch_f16_exact(weight, (size_t)K * N, wscale, sc + 1, s);      // exact-f16 weights: the kernel leaves out the w_lo MFMAs
    l.acc_scale = sc;
  }
  launch_pack_linear(weight, wp, cdt, K, N, l.Kpad, l.Npad, geglu ? 1 : 0, 0, s, nullptr, wscale);
  launch_pack_bias(bias, bp, N, l.Npad, geglu ? 1 : 0, 0, s);
  l.w = wp; l.b = bp;
  tmp_wfrag(l, cdt, geglu != 0, tmp, s);
  Act xa(xi, K, sdt);
  if (cdt == DT_HL) {      // range-safe conversion, as the models convert their fp32 stream tensors (hl_operand)
    float* sc = (float*)tmp.get(hl_scale_floats(1) * sizeof(float));
    launch_f32_to_hl_scaled(x, K, xi, K, (size_t)M, K, sc, s, 1);
    xa.a_scale = hl_scale_inv(sc, 1);
  } else launch_copy_rows(x, DT_F32, K, xi, sdt, K, M, K, s);
  Exec ex; ex.s = s; ex.cdt = cdt; ex.sdt = sdt;
  give_splitk_ws(ex, tmp, 1, M, N, s);
  Epi e; e.act = geglu ? 1 : 0;
  run_linear(ex, l, xa, M, Act(out, geglu ? N / 2 : N, DT_F32), e);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}
int sdxl_layer_norm_linear(sdxl_ctx* ctx, void* stream, const float* x, const float* gamma, const float* beta, float eps,
                           const float* weight, const float* bias, int M, int K, int N, int geglu, int dtype, float* out) {
  API_BEGIN
  SDXL_REQUIRE(ctx && x && gamma && beta && weight && out, "null argument");
  SDXL_REQUIRE(!geglu || (N % 32 == 0), "GEGLU width must be a multiple of 32");
  SDXL_REQUIRE(K % 64 == 0, "LayerNorm width must be a multiple of 64");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  int cdt, sdt; no_mix(dtype); dtypes(dtype, cdt, sdt); no_split(cdt, "sdxl_layer_norm_linear");
  // the model's own builder does the packing / folding: a five-entry parameter list over a device-side flat buffer
  std::vector<ParamSpec> specs(5);
  specs[0].name = "lin.weight"; specs[0].shape = {K, N}; specs[0].kind = PK_LINEAR_W;
  specs[1].name = "lin.bias"; specs[1].shape = {N}; specs[1].kind = PK_BIAS;
  specs[2].name = "norm.gamma"; specs[2].shape = {K}; specs[2].kind = PK_GAMMA;
  specs[3].name = "norm.beta"; specs[3].shape = {K}; specs[3].kind = PK_BETA;
  specs[4].name = "norm.eps"; specs[4].shape = {1}; specs[4].kind = PK_EPS;
  Tmp tmp;
  const size_t nflat = (size_t)K * N + N + 2 * (size_t)K + 1;
  float* flat = (float*)tmp.get(nflat * sizeof(float));
  SDXL_HIP(hipMemcpyAsync(flat, weight, (size_t)K * N * sizeof(float), hipMemcpyDefault, s));
  if (bias) SDXL_HIP(hipMemcpyAsync(flat + (size_t)K * N, bias, (size_t)N * sizeof(float), hipMemcpyDefault, s));
  else SDXL_HIP(hipMemsetAsync(flat + (size_t)K * N, 0, (size_t)N * sizeof(float), s));
  SDXL_HIP(hipMemcpyAsync(flat + (size_t)K * N + N, gamma, (size_t)K * sizeof(float), hipMemcpyDefault, s));
  SDXL_HIP(hipMemcpyAsync(flat + (size_t)K * N + N + K, beta, (size_t)K * sizeof(float), hipMemcpyDefault, s));
  SDXL_HIP(hipMemcpyAsync(flat + (size_t)K * N + N + 2 * (size_t)K, &eps, sizeof(float), hipMemcpyHostToDevice, s));
  SDXL_HIP(hipStreamSynchronize(s));
  FlatSource src(flat, specs);
  DeviceArena arena;
  arena.reserve(WeightBuilder::arena_bound(specs, cdt) + ((size_t)round_up(K, 128) * K * 2 + (1 << 16)));
  WeightBuilder wb(specs, src, arena, cdt, s);
  Exec ex; ex.s = s; ex.cdt = cdt; ex.sdt = sdt;
  const Act o(out, geglu ? N / 2 : N, DT_F32);
  void* xi = tmp.get((size_t)M * K * dt_size(sdt));
  if (cdt == DT_F16 && sdt == DT_F16) {
    // f16 mode of the UNet: the residual stream leaves its producer GEMM with per-64-column (mean, M2) statistics and the
    // consumer applies the LayerNorm in its epilogue.  The producer here is x * I (exact), so xi = fp16(x).
    const Lin l = wb.linear_ln("lin", geglu != 0, "norm");
    Lin id; id.N = K; id.K = K; id.cin = K; id.ksize = 1; id.Kpad = K; id.Npad = (int)round_up(K, 128);
    void* ip = arena.alloc((size_t)id.Npad * id.Kpad * 2);
    float* eye = (float*)tmp.get((size_t)K * K * sizeof(float));
    {
      std::vector<float> h((size_t)K * K, 0.f);
      for (int k = 0; k < K; ++k) h[(size_t)k * K + k] = 1.0f;
      SDXL_HIP(hipMemcpyAsync(eye, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
      SDXL_HIP(hipStreamSynchronize(s));
    }
    launch_pack_linear(eye, ip, DT_F16, K, K, id.Kpad, id.Npad, 0, 0, s);
    id.w = ip; id.b = nullptr;
    tmp_wfrag(id, DT_F16, false, tmp, s);    // (C % 128 == 0: the identity runs on the weights-in-registers kernel, pair-exchanged row statistics)
    void* x16 = tmp.get((size_t)M * K * 2);
    launch_copy_rows(x, DT_F32, K, x16, DT_F16, K, M, K, s);
    float* stat = (float*)tmp.get((size_t)M * (K / 64) * 2 * sizeof(float));
    Epi ep; ep.stat_out = stat;
    run_linear(ex, id, Act(x16, K, DT_F16), M, Act(xi, K, DT_F16), ep);
    Epi e; e.act = geglu ? 1 : 0; e.ln_stat = stat;
    run_linear(ex, l, Act(xi, K, DT_F16), M, o, e);
  } else {
    const NormW n = wb.norm("norm");
    const Lin l = wb.linear("lin", geglu != 0);
    launch_copy_rows(x, DT_F32, K, xi, sdt, K, M, K, s);
    void* ln = tmp.get((size_t)M * K * dt_size(cdt));
    run_layernorm(ex, n, Act(xi, K, sdt), M, Act(ln, K, cdt));
    Epi e; e.act = geglu ? 1 : 0;
    run_linear(ex, l, Act(ln, K, cdt), M, o, e);
  }
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}

int sdxl_ln_query_cross_attention(sdxl_ctx* ctx, void* stream, const float* x, const float* gamma, const float* beta, float eps,
                                  const float* wq, const float* k, const float* v, int B, int Nq, int Nk, int C, int fused,
                                  float* out) {
  // attn2 of a transformer block up to (not including) the output projection: LayerNorm -> query projection (no bias) ->
  // qkv_attention over the projected context, 64 channels per head.  f16 engine only (the UNet's production mode); fused != 0
  // runs the attention inside the projection's epilogue, fused == 0 as projection + attention kernel; fused == 2: the epilogue at split precision
  // (context, q and P as (hi, lo) f16 pairs: the SDXL_DTYPE_F32_SPLIT_MIX_F16W form).
  API_BEGIN
  SDXL_REQUIRE(ctx && x && gamma && beta && wq && k && v && out, "null argument");
  SDXL_REQUIRE(C % 64 == 0 && B >= 1 && Nq >= 1 && Nk >= 1, "State size must be a multiple of head size");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  const int M = B * Nq, vt_ld = (int)round_up(Nk, 64);
  SDXL_REQUIRE(!fused || igemm_xattn_ok(DT_F16, DT_F16, M, C, C, Nq, Nk), "fused cross-attention: unsupported shape");
  std::vector<ParamSpec> specs(5);
  specs[0].name = "lin.weight"; specs[0].shape = {C, C}; specs[0].kind = PK_LINEAR_W;
  specs[1].name = "lin.bias"; specs[1].shape = {C}; specs[1].kind = PK_BIAS;
  specs[2].name = "norm.gamma"; specs[2].shape = {C}; specs[2].kind = PK_GAMMA;
  specs[3].name = "norm.beta"; specs[3].shape = {C}; specs[3].kind = PK_BETA;
  specs[4].name = "norm.eps"; specs[4].shape = {1}; specs[4].kind = PK_EPS;
  Tmp tmp;
  const size_t cc = (size_t)C * C;
  float* flat = (float*)tmp.get((cc + 3 * (size_t)C + 1) * sizeof(float));
  SDXL_HIP(hipMemcpyAsync(flat, wq, cc * sizeof(float), hipMemcpyDefault, s));
  SDXL_HIP(hipMemsetAsync(flat + cc, 0, (size_t)C * sizeof(float), s));
  SDXL_HIP(hipMemcpyAsync(flat + cc + C, gamma, (size_t)C * sizeof(float), hipMemcpyDefault, s));
  SDXL_HIP(hipMemcpyAsync(flat + cc + 2 * (size_t)C, beta, (size_t)C * sizeof(float), hipMemcpyDefault, s));
  SDXL_HIP(hipMemcpyAsync(flat + cc + 3 * (size_t)C, &eps, sizeof(float), hipMemcpyHostToDevice, s));
  SDXL_HIP(hipStreamSynchronize(s));
  FlatSource src(flat, specs);
  DeviceArena arena;
  arena.reserve(WeightBuilder::arena_bound(specs, DT_F16) + ((size_t)round_up(C, 128) * C * 2 + (1 << 16)));
  WeightBuilder wb(specs, src, arena, DT_F16, s);
  Exec ex; ex.s = s; ex.cdt = DT_F16; ex.sdt = DT_F16;
  // beta W is folded into the packed bias; attn2.query has none of its own, so fold with beta as given (zero beta -> no bias)
  Lin l = wb.linear_ln("lin", false, "norm");
  Lin id; id.N = C; id.K = C; id.cin = C; id.ksize = 1; id.Kpad = C; id.Npad = (int)round_up(C, 128);
  void* ip = arena.alloc((size_t)id.Npad * id.Kpad * 2);
  float* eye = (float*)tmp.get(cc * sizeof(float));
  {
    std::vector<float> h(cc, 0.f);
    for (int i = 0; i < C; ++i) h[(size_t)i * C + i] = 1.0f;
    SDXL_HIP(hipMemcpyAsync(eye, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
    SDXL_HIP(hipStreamSynchronize(s));
  }
  launch_pack_linear(eye, ip, DT_F16, C, C, id.Kpad, id.Npad, 0, 0, s);
  id.w = ip; id.b = nullptr;
  tmp_wfrag(id, DT_F16, false, tmp, s);    // (C % 128 == 0: the identity runs on the weights-in-registers kernel, pair-exchanged row statistics)
  void* x16 = tmp.get((size_t)M * C * 2);
  void* xi = tmp.get((size_t)M * C * 2);
  launch_copy_rows(x, DT_F32, C, x16, DT_F16, C, M, C, s);
  float* stat = (float*)tmp.get((size_t)M * (C / 64) * 2 * sizeof(float));
  { Epi ep; ep.stat_out = stat; run_linear(ex, id, Act(x16, C, DT_F16), M, Act(xi, C, DT_F16), ep); }
  // context keys [B][Nk][C] and V^T [B][C][vt_ld] (zero key padding), f16
  void* kd = tmp.get((size_t)B * Nk * C * 2);
  void* vt = tmp.get((size_t)B * C * vt_ld * 2);
  launch_copy_rows(k, DT_F32, C, kd, DT_F16, C, B * Nk, C, s);
  launch_fill_zero(vt, (size_t)B * C * vt_ld * 2, s);
  for (int b = 0; b < B; ++b) {
    const size_t tot = (size_t)Nk * C;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v + (size_t)b * Nk * C, C, Nk, C,
                       (char*)vt + (size_t)b * C * vt_ld * 2, DT_F16, vt_ld);
  }
  void* od = tmp.get((size_t)M * C * 2);
  Epi e; e.ln_stat = stat; e.rpb = Nq;
  if (fused == 3) {
    // the un-fused twin of fused == 2: fp32 q out of the f16 projection, HL16 context, the stand-alone split-operand attention kernel writing f16 rows
    float* q32 = (float*)tmp.get((size_t)M * C * 4);
    run_linear(ex, l, Act(xi, C, DT_F16), M, Act(q32, C, DT_F32), e);
    void* khl = tmp.get((size_t)B * Nk * C * 4);
    float* vt32 = (float*)tmp.get((size_t)B * C * vt_ld * 4);
    void* vhl = tmp.get((size_t)B * C * vt_ld * 4);
    launch_f32_to_hl(k, C, khl, C, (size_t)B * Nk, C, s);
    launch_fill_zero(vt32, (size_t)B * C * vt_ld * 4, s);
    for (int b = 0; b < B; ++b) {
      const size_t tot = (size_t)Nk * C;
      hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v + (size_t)b * Nk * C, C, Nk, C,
                         (char*)vt32 + (size_t)b * C * vt_ld * 4, DT_F32, vt_ld);
    }
    launch_f32_to_hl(vt32, vt_ld, vhl, vt_ld, (size_t)B * C, vt_ld, s);
    AttnParams p{};
    p.Q = q32; p.ldq = C; p.K = khl; p.ldk = C; p.Vt = vhl; p.vt_ld = vt_ld; p.O = od; p.ldo = C;
    p.dt = DT_HL; p.B = B; p.H = C / 64; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
    p.q_dt = DT_F32; p.o_dt = DT_F16;
    SDXL_REQUIRE(launch_attention_d64_hl(p, s), "split-operand attention: unsupported shape");
  } else if (fused == 2) {
    // split precision (IgemmParams::xa_k_lo): the fp32 context as (hi, lo) f16 pairs, q and P split inside the epilogue -- three MFMAs per product
    void* kh = tmp.get((size_t)B * Nk * C * 2); void* kl = tmp.get((size_t)B * Nk * C * 2);
    float* vt32 = (float*)tmp.get((size_t)B * C * vt_ld * 4);
    void* vh = tmp.get((size_t)B * C * vt_ld * 2); void* vl = tmp.get((size_t)B * C * vt_ld * 2);
    launch_f32_to_f16_pair(k, C, kh, kl, C, (size_t)B * Nk, C, s);
    launch_fill_zero(vt32, (size_t)B * C * vt_ld * 4, s);
    for (int b = 0; b < B; ++b) {
      const size_t tot = (size_t)Nk * C;
      hipLaunchKernelGGL(transpose_pad_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, v + (size_t)b * Nk * C, C, Nk, C,
                         (char*)vt32 + (size_t)b * C * vt_ld * 4, DT_F32, vt_ld);
    }
    launch_f32_to_f16_pair(vt32, vt_ld, vh, vl, vt_ld, (size_t)B * C, vt_ld, s);
    void* xa = tmp.get(xattn_pack_bytes(B, C)); void* xal = tmp.get(xattn_pack_bytes(B, C));
    launch_xattn_pack(kh, vh, xa, B, C, Nk, vt_ld, s);
    launch_xattn_pack(kl, vl, xal, B, C, Nk, vt_ld, s);
    e.xa_k = xa; e.xa_k_lo = xal; e.xa_nctx = Nk; e.xa_scale = 0.125f;
    run_linear(ex, l, Act(xi, C, DT_F16), M, Act(od, C, DT_F16), e);
  } else if (fused) {
    void* xa = tmp.get(xattn_pack_bytes(B, C));
    launch_xattn_pack(kd, vt, xa, B, C, Nk, vt_ld, s);
    e.xa_k = xa; e.xa_nctx = Nk; e.xa_scale = 0.125f;
    run_linear(ex, l, Act(xi, C, DT_F16), M, Act(od, C, DT_F16), e);
  } else {
    void* qd = tmp.get((size_t)M * C * 2);
    run_linear(ex, l, Act(xi, C, DT_F16), M, Act(qd, C, DT_F16), e);
    AttnParams p{};
    p.Q = qd; p.ldq = C; p.K = kd; p.ldk = C; p.Vt = vt; p.vt_ld = vt_ld; p.O = od; p.ldo = C;
    p.dt = DT_F16; p.B = B; p.H = C / 64; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
    launch_attention_d64(p, s);
  }
  launch_copy_rows(od, DT_F16, C, out, DT_F32, C, M, C, s);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}

int sdxl_conv2d_group_norm(sdxl_ctx* ctx, void* stream, const float* x, const float* weight, const float* bias, const float* residual,
                           const float* gamma, const float* beta, float eps, int B, int Cin, int H, int W, int Cout, int n_group,
                           int silu, int fused, int* fused_taken, float* out) {
  // conv3x3 (pad 1, + optional residual) followed by GroupNorm(+SiLU) -- the conv -> norm pairs of ResBlock::forward
  // (unet/mod.rs:1082-1106) and of the SpatialTransformer entry (:820-845) on the f16 engine.  fused != 0 asks the convolution's
  // epilogue for the GroupNorm statistics (no statistics pass); *fused_taken reports whether the selected kernel provided them.
  API_BEGIN
  SDXL_REQUIRE(ctx && x && weight && gamma && beta && out, "null argument");
  SDXL_REQUIRE(n_group > 0 && Cout % n_group == 0, "The number of channels must be divisible by the number of groups");
  SDXL_REQUIRE(Cout % 8 == 0 && n_group <= 256, "unsupported GroupNorm shape");
  use(ctx);
  hipStream_t s = pick(ctx, stream);
  const int cdt = DT_F16, HW = H * W;
  Lin l; l.N = Cout; l.cin = Cin; l.ksize = 3; l.K = Cin * 9;
  l.Kpad = (int)round_up(l.K, 64); l.Npad = (int)round_up(Cout, 128);
  Tmp tmp;
  void* wp = tmp.get((size_t)l.Npad * l.Kpad * 2);
  float* bp = (float*)tmp.get((size_t)l.Npad * sizeof(float));
  void* xi = tmp.get((size_t)B * HW * Cin * 2);
  void* hh = tmp.get((size_t)B * HW * Cout * 2);
  void* yo = tmp.get((size_t)B * HW * Cout * 2);
  void* ri = residual ? tmp.get((size_t)B * HW * Cout * 2) : nullptr;
  float* part = (float*)tmp.get(groupnorm_workspace_floats(B, n_group) * sizeof(float));
  float* eps_d = (float*)tmp.get(sizeof(float));
  launch_pack_conv(weight, wp, cdt, Cout, Cin, 3, l.Kpad, l.Npad, s);
  launch_pack_bias(bias, bp, Cout, l.Npad, 0, 0, s);
  l.w = wp; l.b = bp;
  launch_nchw_to_nhwc(x, Cin * HW, xi, cdt, B, Cin, HW, Cin, 1.0f, s);
  if (residual) launch_nchw_to_nhwc(residual, Cout * HW, ri, cdt, B, Cout, HW, Cout, 1.0f, s);
  SDXL_HIP(hipMemcpyAsync(eps_d, &eps, sizeof(float), hipMemcpyHostToDevice, s));
  Exec ex; ex.s = s; ex.cdt = cdt; ex.sdt = cdt; ex.gn_partial = part;
  give_splitk_ws(ex, tmp, B, HW, Cout, s);
  Act h(hh, Cout, cdt);
  Epi e;
  if (residual) e.R = Act(ri, Cout, cdt);
  if (fused && HW % 256 == 0) e.gn_part = (float*)tmp.get((size_t)B * HW / 256 * Cout * 2 * sizeof(float));
  const bool took = run_conv(ex, l, Act(xi, Cin, cdt), Cin, ConvGeom{B, H, W, H, W, 3, 1, 1, 0}, h, e);
  if (took) { h.gn_part = e.gn_part; h.gn_rt = HW / 256; }
  if (fused_taken) *fused_taken = took ? 1 : 0;
  NormW n; n.gamma = gamma; n.beta = beta; n.eps = eps_d; n.C = Cout;
  run_groupnorm(ex, n, h, B, HW, Act(yo, Cout, cdt), silu != 0, n_group);
  launch_nhwc_to_nchw(yo, cdt, Cout, out, B, Cout, HW, 1.0f, s);
  SDXL_HIP(hipStreamSynchronize(s));
  API_END
}

}  // extern "C"
