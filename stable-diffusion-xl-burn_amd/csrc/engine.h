// Host-side engine of the MI355X SDXL sampler: parameter enumeration, weight packing into a device arena,
// static per-shape execution plans (bump-allocated activation arena -> stable addresses -> hipGraph replay),
// UNet / VAE / DDIM sampler drivers.  The C ABI in capi.cpp is a thin shell over these classes.
#pragma once
#include "kernels.h"

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace sdxl {

struct Error : std::runtime_error {
  explicit Error(const std::string& m) : std::runtime_error(m) {}
};
#define SDXL_HIP(x)                                                                                   \
  do {                                                                                                \
    hipError_t e__ = (x);                                                                             \
    if (e__ != hipSuccess) throw ::sdxl::Error(std::string(#x) + " failed: " + hipGetErrorString(e__)); \
  } while (0)
#define SDXL_REQUIRE(cond, msg)                          \
  do {                                                   \
    if (!(cond)) throw ::sdxl::Error(std::string(msg));  \
  } while (0)

static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------ configs / specs
// mirrors the reference's UNetConfig (unet/mod.rs:59-69) + DiffuserConfig.is_refiner (stablediffusion/mod.rs:269-278)
struct UNetCfg {
  int adm_in_channels = 0, in_channels = 4, out_channels = 4, model_channels = 0;
  std::vector<int> channel_mults;
  int n_head_channels = 64;
  std::vector<int> transformer_depths;
  int context_dim = 0;
  bool is_refiner = false;
};
// AutoencoderConfig::init hard-codes the SDXL values (autoencoder/mod.rs:27-35); kept configurable for tiny tests
struct VaeCfg {
  std::vector<std::pair<int, int>> enc{{128, 128}, {128, 256}, {256, 512}, {512, 512}};
  std::vector<std::pair<int, int>> dec{{512, 512}, {512, 512}, {512, 256}, {256, 128}};
  int n_group = 32;
  int enc_out = 8;
  double scale_factor = 0.13025;
};

// mirrors CLIPConfig (clip/mod.rs:19-28)
struct ClipCfg { int n_vocab = 49408, n_state = 0, embed_dim = 0, n_head = 0, n_ctx = 77, n_layer = 0; bool quick_gelu = false; };

enum ParamKind { PK_LINEAR_W = 0, PK_CONV_W = 1, PK_BIAS = 2, PK_GAMMA = 3, PK_BETA = 4, PK_EPS = 5 };
struct ParamSpec {
  std::string name;
  std::vector<int> shape;
  int kind;
  float scale, mean;   // synthetic init: value = (u - 0.5)*scale + mean
  size_t numel() const { size_t n = 1; for (int d : shape) n *= (size_t)d; return n; }
};
enum BlockKind { BK_CONV, BK_RES, BK_DOWN, BK_REST, BK_RESTU, BK_RESU };
struct BlockDesc { int kind = 0, c_in = 0, c_emb = 0, c_out = 0, n_head = 0, depth = 0; };
void unet_block_plan(const UNetCfg& cfg, std::vector<BlockDesc>& inp, BlockDesc& mid, std::vector<BlockDesc>& out);
std::vector<ParamSpec> unet_param_specs(const UNetCfg& cfg);
std::vector<ParamSpec> vae_decoder_param_specs(const VaeCfg& cfg);
std::vector<ParamSpec> vae_encoder_param_specs(const VaeCfg& cfg);
std::vector<ParamSpec> clip_param_specs(const ClipCfg& cfg);
uint64_t fnv1a64(const std::string& s);

// ------------------------------------------------------------------------------------------ memory
struct DeviceArena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = false;     // dry run: hand out offsets from a null base, only track the peak
  ~DeviceArena();
  void reserve(size_t bytes);
  void* alloc(size_t bytes);
  size_t mark() const { return off; }
  void reset(size_t m) { off = m; }
};

// where canonical (reference-layout fp32) parameter tensors come from
struct WeightSource {
  virtual ~WeightSource() {}
  virtual void fetch(const ParamSpec& s, size_t index, float* dst_dev, hipStream_t st) = 0;
  virtual bool empty() const { return false; }   // true: lay the arena out but leave its contents to a later broadcast
};
struct NullSource : WeightSource {   // replica ranks: same arena layout, contents arrive by RCCL broadcast from rank 0
  void fetch(const ParamSpec&, size_t, float*, hipStream_t) override {}
  bool empty() const override { return true; }
};
constexpr float kHiLoScale = 256.0f;    // lo halves of WeightBuilder::linear_hilo travel times 2^8 (out of the f16 subnormals), the A operand's second copy times 2^-8
constexpr uint64_t kSeedF16Weights = 1ull << 63;   // seed flag (C ABI: SDXL_SEED_F16_WEIGHTS): synthetic parameters rounded to f16
struct SyntheticSource : WeightSource {
  uint64_t seed;
  explicit SyntheticSource(uint64_t s) : seed(s) {}
  void fetch(const ParamSpec& s, size_t index, float* dst_dev, hipStream_t st) override;
};
struct FlatSource : WeightSource {   // flat fp32 buffer (host or device) in spec order
  const float* base; std::vector<size_t> offsets;
  FlatSource(const float* b, const std::vector<ParamSpec>& specs);
  void fetch(const ParamSpec& s, size_t index, float* dst_dev, hipStream_t st) override;
};

struct FlatSourceF16 : WeightSource {   // flat IEEE-f16 buffer (host or device) in spec order -- how burn's HalfPrecisionSettings
  // records store the weights (src/bin/sample/main.rs:37, src/bin/convert/main.rs:65-70): half the host memory of the fp32 path
  const uint16_t* base; std::vector<size_t> offsets;
  void* stage = nullptr; size_t stage_numel = 0;
  FlatSourceF16(const uint16_t* b, const std::vector<ParamSpec>& specs);
  ~FlatSourceF16() override;
  void fetch(const ParamSpec& s, size_t index, float* dst_dev, hipStream_t st) override;
};

struct Lin {        // packed dense weight: [Npad][Kpad] compute dtype + fp32 bias [Npad]
  const void* w = nullptr; const float* b = nullptr;
  const void* wf = nullptr;        // f16 linear layers / 1x1 convs with N % 128 == 0: the same weights in MFMA fragment order (igemm_wreg.hip)
  int N = 0, K = 0, Kpad = 0, Npad = 0, ksize = 1, cin = 0;
  // LayerNorm folded in (linear_ln / fused_linear_ln): w = diag(gamma) W, b = beta W + bias, cs = column sums of the
  // packed rows; the GEMM then takes the RAW rows plus their per-64-column (mean, M2) statistics -- see IgemmParams::ln_stat
  const float* cs = nullptr;
  const float* ln_eps = nullptr;   // device scalar: eps of the folded LayerNorm
  int ln_k = 0;                    // folded LayerNorm over ln_k columns (0: K) -- the (hi | lo) shadow form packs K = 2 ln_k
  int k_form = 0;                  // 0 plain; 1: K doubled as two HALVES, weight (hi | lo s) or (w | w / s) against [a | a / s] or [a_hi | a_lo s] (linear_hilo); 2: K doubled in the HL16
                                   // interleave, the weight twice per 16-channel group: the A operand is an un-scaled HL16 tensor read as f16 (MIX_LINEAR_F16X2)
  int dt = -1;                     // dtype the weight was packed in when it differs from the model's compute dtype (-1: the model's):
                                   // a DT_HL model packs the layers its pipeline cannot take (Cin % 32 != 0) as fp32
  const float* acc_scale = nullptr;   // DT_HL packing: device scalar (weight arena) 1 / (power-of-two factor the packed weights carry)
};
// eps lives in the weight arena (device scalar, read by the kernels): replicas that receive the arena by broadcast need no
// host-side copy, and a checkpoint's per-norm eps (groupnorm/load.rs:19, layernorm/load.rs:17) travels with the weights
struct NormW { const float* gamma = nullptr; const float* beta = nullptr; const float* eps = nullptr; int C = 0; };

struct WeightBuilder {
  const std::vector<ParamSpec>& specs;
  std::map<std::string, size_t> index;
  WeightSource& src;
  DeviceArena& arena;
  int dt;
  hipStream_t st;
  float* tmp = nullptr; size_t tmp_numel = 0;
  std::string last_fetched;      // name of the tensor `tmp` holds (fetch() of the same name again is free)
  WeightBuilder(const std::vector<ParamSpec>& sp, WeightSource& s, DeviceArena& a, int dtype, hipStream_t stream);
  ~WeightBuilder();
  // wfrag: the model keeps fragment-order images of its plain f16 linear / 1x1 weights (attach_wfrag) -- only the UNet's layers can be
  // routed to the weights-in-registers kernel, so the CLIP towers and the VAE leave it off (half the arena / broadcast bytes for those layers)
  bool wfrag = true;
  static size_t arena_bound(const std::vector<ParamSpec>& specs, int dt, bool wfrag = true);
  const ParamSpec& spec(const std::string& name, size_t* idx = nullptr) const;
  bool has(const std::string& name) const { return index.count(name) != 0; }
  const float* fetch(const std::string& name);            // canonical fp32 tensor in tmp
  // dt_override >= 0: pack in that dtype whatever the model's (the GEMV weights of a split-operand model stay fp32)
  Lin linear(const std::string& name, bool geglu = false, int dt_override = -1);   // name.weight [K,N] (+ name.bias)
  Lin fused_linear(const std::vector<std::string>& names, int dt_override = -1);   // concatenated along N (same K)
  Lin linear_hilo(const std::string& name, bool geglu, bool dup = false, bool hl_interleave = false);   // hl_interleave: the weight twice in the HL16 interleave (launch_pack_linear_hilo mode 2; + fragment-order image)
  //     // dup: (w | w / kHiLoScale) for an A operand of (hi | lo * kHiLoScale) ACTIVATION halves
      // f16 GEMM on (hi | lo * kHiLoScale) weight halves along a doubled K: un-rounded weights (MIX_GEGLU_HILO)
  float hl_scale(Lin& l, const std::vector<std::string>& weight_names);     // DT_HL packing: power-of-two factor, inverse into the arena
  // the same with the preceding LayerNorm(gamma, beta) folded into weight / bias / column sums
  Lin linear_ln(const std::string& name, bool geglu, const std::string& norm);
  Lin fused_linear_ln(const std::vector<std::string>& names, const std::string& norm);
  // dt_override / shadow / plain: the SHADOW form (round 6, split-operand models): the weights are packed UN-folded in dt_override (f16) -- gamma rides on
  // the A operand, an f16 shadow  f16(x o gamma)  the producer of the fp32 stream leaves (IgemmParams::shadow) -- with cs = gamma W over the packed values
  // and b = beta W + bias; *plain receives the same packed matrix with the canonical bias (for the LayerNorm-launch path where no shadow exists)
  Lin fold_ln(const std::vector<std::string>& names, const std::string& norm, bool geglu, int dt_override = -1, bool shadow = false, Lin* plain = nullptr, bool hilo_dup = false,
              bool hl_interleave = false);   // hl_interleave: the weights twice in the HL16 interleave (K = 2 x, k_form 2) for an HL16 shadow (shadow form only; any number of fused names)   // hilo_dup: (w | w / kHiLoScale) along a doubled K for a (hi | lo) shadow
  // AND of "every value of these tensors is exactly one f16" (device flag read back): what SDXL_DTYPE_F32_SPLIT_MIX_F16W asks of the classes it moves to f16
  bool all_f16_exact(const std::vector<std::string>& names);
  float* tmp2 = nullptr; size_t tmp2_numel = 0;   // scratch for folded biases (device)
  Lin conv(const std::string& name);                                        // name.weight [Cout,Cin,k,k] + bias
  void attach_wfrag(Lin& l, bool fill);    // second image of a plain f16 linear / 1x1 weight in fragment order (arena; no-op for other layers)
  NormW norm(const std::string& name);
};

// activation view: rows of `ld` elements
struct Act {
  void* p = nullptr; int ld = 0; int dt = DT_F16;
  // GroupNorm statistics of this tensor left behind by the GEMM that produced it (Epi::gn_part -> IgemmParams::gn_part):
  // [B][gn_rt][C] (mean, M2) per 256-row tile and channel.  A GroupNorm over exactly this tensor skips its statistics pass.
  const float* gn_part = nullptr; int gn_rt = 0;
  const float* a_scale = nullptr;   // HL16 copy of an fp32 stream tensor (hl_operand): 2^-e per batch entry, multiplied back by the consuming GEMM (IgemmParams::a_scale)
  int a_scale_n = 1;                // entries of a_scale (equal row counts each)
  Act() {}
  Act(void* p_, int ld_, int dt_) : p(p_), ld(ld_), dt(dt_) {}
  Act cols(int c0) const { Act a((char*)p + (size_t)c0 * dt_size(dt), ld, dt); a.a_scale = a_scale; a.a_scale_n = a_scale_n; return a; }   // (a column slice drops the statistics)
};

// per-kernel-class hipEvent profiler (eager runs only): live measurement of the dominant kernel for bench.py's roofline
struct Profiler {
  enum { IGEMM = 0, ATTENTION = 1, GROUPNORM = 2, LAYERNORM = 3, OTHER = 4, NCLS = 5 };
  struct Rec { hipEvent_t a, b; int cls; double flops; int m, n, k, ks, tag; };
  std::vector<Rec> recs;
  void begin(int cls, double flops, hipStream_t s, int m = 0, int n = 0, int k = 0, int ks = 0, int tag = 0);   // tag: DemoteClass bit of the launch (UNet; 0 = untagged)
  void end(hipStream_t s);
  void collect(float ms[NCLS], int launches[NCLS], double flops[NCLS]);   // synchronises, then frees the events
};

// the GEMM launches of one forward in launch order (weights each one reads, whether its kernel can host warming workgroups):
// recorded on the plan's first eager forward, replayed on every later one -- launch i is handed the weights of a later launch
// (IgemmParams::warm).  Static shapes: the sequence of a plan never changes.
struct WarmSeq {
  struct Item { const void* w; unsigned bytes; bool host; unsigned budget; const void* warm[3]; unsigned warm_bytes[3]; };
  std::vector<Item> seq;
  size_t pos = 0;
  bool recording = false, ready = false;
  void finish();       // assigns targets to hosts (weights.cpp)
};
struct Exec {
  Profiler* prof = nullptr;
  hipStream_t s = nullptr;
  bool dry = false;
  int cdt = DT_F16;      // compute dtype (MFMA operands)
  int sdt = DT_F16;      // residual-stream dtype
  DeviceArena* act = nullptr;
  float* gn_partial = nullptr;
  float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0; unsigned* splitk_cnt = nullptr;   // igemm split-K workspace (optional)
  const float* ebias = nullptr;   // [nb][emb_total] per-ResBlock time-embedding biases of this run's batch entries
  int b0 = 0;                      // first batch entry of this run (split-CFG chains address the K/V caches with it)
  hipEvent_t fork_ev = nullptr;    // split-CFG: recorded on s after the fork_after-th GEMM launch of chain 0 -- the second
  int fork_after = 0, launches = 0; // chain starts there, so the two chains run out of phase (GEMMs of one under the attention of the other)
  WarmSeq* warm = nullptr;         // weight warming schedule of the plan (null: off)
  float* attn_xws = nullptr; unsigned* attn_xcnt = nullptr;   // workspace / tickets of the cross-workgroup key split (AttnParams::xws)
  int demote = 0;                  // split-operand UNet, precision-frontier instrument: GEMM classes (DemoteClass bits) whose operands lose their lo halves
  Act alloc(size_t rows, int C, int dt) {
    return Act(act->alloc(rows * (size_t)C * dt_size(dt)), C, dt);
  }
};

// Precision-frontier instrument (round 5; sdxl_debug_set "hl_demote" = bit set).  A split-operand (DT_HL) UNet runs the GEMMs of the listed
// classes on operands whose lo halves are zero -- activations by a zeroing pass behind their producer, weights through the "every weight is
// one f16" flag of the packed matrix (IgemmParams::acc_scale[1]: the kernel leaves the w_lo MFMAs out), the attention through
// AttnParams::demote -- i.e. with exactly the f16 engine's operand rounding (f16 x f16 products, fp32 accumulation) while every other class
// keeps fp32-class operands.  Measures each class's share of the f16 mode's error on the config-2 trajectory (tools/precision_frontier.py).
// Classes: self-attention QKV projection, the self-attention itself (q, k, v, p), attention out-projections, cross-attention (query / key / value
// projections + the 77-key attention), GEGLU projection, FF-out, and five kinds of convolution: the two 3x3 convs of every ResBlock, the 1x1 skip
// connections, the UNet's last conv (its first has 4 input channels and is plain fp32 in this engine), the down / up-sampling convs, proj_in / proj_out.
// SDXL_DTYPE_F32_SPLIT_MIX: what the measured frontier (profiles/r05_precision_frontier.json) lets a latents-within-bound engine run in f16
// (classes 4 ... 32: SDXL_DTYPE_F32_SPLIT_MIX_F16W, or sdxl_debug_set "mix_classes" -- outside the bound on the synthetic fp32 weights, inside it on
//  f16-representable ones)
enum MixClass {
  MIX_ATTN_F16 = 1,     // self-attention on the f16 flash kernels
  MIX_GEGLU_F16 = 2,    // GEGLU projection (f16 LayerNorm output x f16 weights)
  MIX_QKV_F16 = 4,      // QKV projection (implies the f16 self-attention)
  MIX_FF_F16 = 8,       // FF-out (reads the GEGLU kernel's f16 output)
  MIX_OUT1_F16 = 16,    // self-attention out-projection (reads the f16 attention output as it is)
  MIX_OUT2_F16 = 32,    // cross-attention out-projection (the split-operand attention rounds its fp32 result to f16 once, in its store)
  MIX_XATTN_F16 = 64,   // with OUT2: cross-attention + its query projection as the f16 engine's fused launch -- a knob, in no mode
  MIX_Q2_F16 = 128,     // cross-attention QUERY projection alone on f16 operands (HL16 output: the split-operand attention behind it keeps an fp32-class q)
  MIX_XATTN_SPLIT = 512, // with MIX_Q2_F16: the 77-key cross-attention at SPLIT precision inside the f16 query projection's epilogue (IgemmParams::xa_k_lo) -- the
                        // arithmetic of the stand-alone split-operand attention without its launch
  MIX_GEGLU_HILO = 1024, // with MIX_GEGLU_F16 (and without the shadow form): the GEGLU weights as (hi, lo) f16 pairs along a doubled K against [a | a 2^-8] -- the f16 wide-tile
                        // kernel at twice the depth, two MFMAs per product: activation rounding only on ANY weights.  SDXL_DTYPE_F32_SPLIT_MIX
  MIX_GEGLU_AHILO = 2048, // a knob, in no mode (f16-representable weights): the GEGLU projection's ACTIVATIONS as (hi, lo) f16 pairs along a doubled K against (w | w 2^-8) --
                        // the class that carries 70 % of the F16W mode's error variance at two MFMAs per product on the f16 kernel
  MIX_LINEAR_F16X2 = 4096, // (f16-representable weights) transformer linears whose A operand is an un-scaled HL16 tensor run on the F16 kernels: an HL16 row of C channels is an f16 row of 2 C
                        // columns, the weight is packed twice in the same interleave -- two MFMAs per product on the weights-in-registers / wide / pipe kernels
  MIX_LN_SHADOW = 256   // the LayerNorms in front of the f16 projections (QKV, GEGLU, the query projection with MIX_Q2_F16) folded into them: the producers of
                        // the fp32 stream leave an f16 shadow f16(x o gamma) + row statistics (IgemmParams::shadow), no LayerNorm launch
};
enum DemoteClass { DM_QKV = 1, DM_ATTN = 2, DM_OUT = 4, DM_XATTN = 8, DM_GEGLU = 16, DM_FF = 32, DM_CONV_RES = 64, DM_CONV_SKIP = 128,
                   DM_CONV_IO = 256, DM_CONV_UPDOWN = 512, DM_CONV_PROJ = 1024 };
void unet_set_mix_classes(int v);
void unet_set_hl_demote(int mask);
int unet_hl_demote();

// thin launch helpers shared by unet.cpp / vae.cpp (skip the launch on dry runs)
struct ConvGeom { int B, Hin, Win, Hout, Wout, ksize, stride, pad, up; };
struct Epi {
  const float* ln_stat = nullptr;   // [K/64][M][2] per-slot (mean, M2) of the A rows: the weight is LayerNorm-folded (Lin::cs)
  float* stat_out = nullptr;        // [N/64][M][2]: leave the per-slot (mean, M2) of the output rows for the next folded LayerNorm
  const float* ebias = nullptr; int ebias_ld = 0;
  int act = 0;
  Act R;             // residual (p == nullptr -> none)
  int n_split = -1;  // >=0: columns >= n_split go transposed into Ct
  void* Ct = nullptr; int ct_rows = 0, ct_ld = 0;
  int rpb = 0;       // rows per batch for ebias / transposed store (0 -> Hout*Wout)
  // cross-attention fused into the projection's epilogue (IgemmParams::xa_*): packed context of this run's batch entries
  const void* xa_k = nullptr; int xa_nctx = 0; float xa_scale = 0.f;   // xa_k: operand-order image (launch_xattn_pack)
  const void* xa_k_lo = nullptr;     // split precision (IgemmParams::xa_k_lo): xa_k = hi halves, this = lo halves of the context keys / values
  // room for the GroupNorm statistics of the output ([M/256][N] float pairs); run_conv reports whether the kernel it picked
  // filled it (igemm_gn_part_ok), the caller then tags the output Act
  float* gn_part = nullptr;
  int cls = 0;       // DemoteClass bit of this GEMM (UNet call sites): label of the launch in the per-launch profile dump, nothing else
  // f16 shadow of an fp32 output for the GEMM behind the next LayerNorm (IgemmParams::shadow): asked for by the caller, written only when the kernel the
  // selection picks can (weights-in-registers kernel) -- *shadow_done tells; stat_out then holds the fp32 rows' statistics
  void* shadow = nullptr; int shadow_ld = 0; const float* shadow_gamma = nullptr; bool* shadow_done = nullptr; float shadow_lo_scale = 0.f;
};
bool run_conv(Exec& ex, const Lin& w, const Act& a, int cin, const ConvGeom& g, const Act& out, const Epi& e = Epi());   // true: e.gn_part was filled
bool run_linear(Exec& ex, const Lin& w, const Act& a, int M, const Act& out, const Epi& e = Epi());
// have_max: a hl_scale_floats(nb) buffer whose max|x| partials a GroupNorm statistics pass over x already wrote (run_groupnorm absmax_out): no absmax pass
Act hl_operand(Exec& ex, const Lin& w, const Act& x, size_t rows, int C, int nb = 1, float* have_max = nullptr);    // HL16 copy of an fp32 stream tensor (nb batch entries of rows / nb rows, one power-of-two scale each) for a split-operand GEMM (else x)
// absmax_out: hl_scale_floats(B) floats; the statistics pass (fp32 x without producer statistics -- the caller checks) also leaves max|x| partials per entry there
void run_groupnorm(Exec& ex, const NormW& n, const Act& x, int B, int HW, const Act& y, bool silu, int groups = 32, float* absmax_out = nullptr);
void run_layernorm(Exec& ex, const NormW& n, const Act& x, int rows, const Act& y, float dup_scale = 0.f);   // dup_scale: LayerNormParams::dup_scale

// ------------------------------------------------------------------------------------------ UNet
struct ResBlockW { NormW norm_in, norm_out; Lin conv_in, conv_out, skip; bool has_skip = false; int emb_off = 0, cin = 0, cout = 0; };
struct TBlockW { NormW n1, n2, n3; Lin qkv, out1, q2, kv2, out2, geglu, ff;
                 Lin qkv_sh, q2_sh, geglu_sh; };   // *_sh: shadow forms (MIX_LN_SHADOW; .cs set) sharing the packed matrix of their plain twin
struct STW { NormW norm; Lin proj_in, proj_out; std::vector<TBlockW> blocks; int C = 0, heads = 0; };
struct BlockW { BlockDesc d; ResBlockW res; STW st; Lin conv; };

class UNet {
 public:
  // mix (split-operand compute only): MIX_* classes that run on plain f16 operands instead (SDXL_DTYPE_F32_SPLIT_MIX)
  UNet(const UNetCfg& cfg, int compute_dt, int stream_dt, WeightSource& src, hipStream_t st, int mix = 0);
  ~UNet();
  const UNetCfg& cfg() const { return cfg_; }
  // conditioning that is constant over a trajectory: context [B][n_ctx][ctx_dim] fp32 (device), label [B][adm] fp32.
  // Projects the cross-attention K / V^T of all transformer blocks once (reference recomputes them every step,
  // unet/mod.rs:1010-1011) and the label embedding MLP (:464-466).
  void set_context(const float* context, int n_ctx, const float* label, int B, hipStream_t s);
  // x: NHWC [B][H*W][in_ch] in compute dtype at unet_in(); t: device fp32 per batch entry (stride t_stride);
  // result: NHWC [B][H*W][out_ch] fp32 at eps_out().
  void forward(int B, int H, int W, const float* t_dev, int t_stride, hipStream_t s);
  // reference-shaped entry (unet/mod.rs:450-456): NCHW fp32 in/out, int32 device timesteps
  void forward_nchw(const float* x, const int* timesteps, const float* context, int n_ctx, const float* label,
                    int B, int H, int W, float* out, hipStream_t s);
  void* unet_in(int B, int H, int W);     // ensures the plan exists
  float* eps_out() { return eps_; }
  int compute_dt() const { return cdt_; }
  int mix_classes() const { return mix_; }     // MixClass bits in force (a SDXL_DTYPE_F32_SPLIT_MIX_F16W model on parameters that are not f16 values falls back to F32_SPLIT_MIX's)
  // dtype of the NHWC input the sampler writes (unet_in) and of the attention operands: the compute dtype, except that the
  // split-operand mode (DT_HL GEMM operands) keeps the 4-channel input and q / k / V^T in plain fp32
  int input_dt() const { return cdt_ == DT_HL ? DT_F32 : cdt_; }
  int attn_dt() const { return cdt_ == DT_HL ? DT_F32 : cdt_; }
  void set_use_graph(bool g) { use_graph_ = g; }
  // per-handle option: run the two entries of a batch-2 forward (the CFG pair) as two concurrent batch-1 chains on two
  // streams, the second released after `release_offset` GEMM launches of the first; bit-identical results
  void set_split_cfg(bool on, int release_offset) { split_cfg_ = on; split_offset_ = release_offset; }
  void set_fused_cross_attention(bool on) { fuse_xattn_ = on; }
  void set_gn_from_producer(bool on) { gn_from_producer_ = on; }   // re-plans (arena + graph) on the next forward, like the two options above
  // one eager forward of the current plan/context with hipEvents around every launch, summed per kernel class
  void profile(int B, int H, int W, float ms[Profiler::NCLS], int launches[Profiler::NCLS], double flops[Profiler::NCLS],
               hipStream_t s);
  float eager_ms(int B, int H, int W, hipStream_t s);     // the chain profile() runs, without the per-launch events (best of three)
  size_t weight_bytes() const { return warena_.off; }
  void* weight_base() const { return warena_.base; }

 private:
  void build_weights(WeightSource& src, hipStream_t st);
  void ensure_plan(int B, int H, int W);
  void run(Exec& ex, const float* t_dev, int t_stride, int b0, int nb);
  const float* res_block(Exec& ex, const ResBlockW& w, const Act& x, int B, int H, int W, const Act& out, float* out_gn_part = nullptr);
  void spatial_transformer(Exec& ex, const STW& w, int st_index, const Act& x, int B, int H, int W);

  UNetCfg cfg_;
  int cdt_, sdt_;
  int mix_ = 0;                          // MixClass bits (split-operand engine): classes on plain f16 operands
  bool mix_knob_ = false;                // the bits came from sdxl_debug_set "mix_classes" (no f16-exactness fallback)
  DeviceArena warena_;
  std::vector<BlockW> inp_, out_;
  BlockW mid_res1_, mid_res2_;   // middle_block: res1 -> transformer (in mid_res1_.st) -> res2
  Lin lin1_t_, lin2_t_, lin1_l_, lin2_l_, embcat_;
  NormW norm_out_; Lin conv_out_;
  int emb_total_ = 0;
  // cross-attention K / V^T caches (one per transformer block, in execution order)
  struct KV { void* k = nullptr; void* vt = nullptr; void* xa = nullptr; void* xa_lo = nullptr; };   // xa: operand-order image for the fused epilogue (f16; with xa_lo: hi / lo halves of the fp32-class projection)
  std::vector<std::vector<KV>> kv_;     // [spatial transformer][block]
  std::vector<const STW*> st_list_;
  DeviceArena ctx_arena_;
  int ctx_B_ = 0, n_ctx_ = 0, vt_ld_ctx_ = 0;
  float* label_emb_ = nullptr;           // [B][4mc]
  // plan
  int pB_ = 0, pH_ = 0, pW_ = 0;
  DeviceArena act_;
  void* in_ = nullptr; float* eps_ = nullptr;
  float *temb_ = nullptr, *g1_ = nullptr, *emb_ = nullptr, *ebias_ = nullptr, *gn_partial_ = nullptr, *tconv_ = nullptr;
  // split-K workspaces of the implicit GEMM (slabs + arrival counters), one per concurrent chain
  float* skws_[2] = {nullptr, nullptr}; unsigned* skcnt_[2] = {nullptr, nullptr}; size_t skws_bytes_ = 0;
  bool fuse_ln_ = false;                 // f16 compute + f16 residual stream: LayerNorms are folded into the GEMMs
  bool use_graph_ = true;
  // split-CFG mode: the two entries of a batch-2 forward run as two independent batch-1 chains on two streams (fork / join
  // by events, captured into the same graph); the second chain has its own scratch arena
  bool split_cfg_ = false; int split_offset_ = 0;
  bool plan_split_ = false; int graph_off_ = 0;
  bool fuse_xattn_ = true, plan_xattn_ = true;   // cross-attention inside the query projection's epilogue (f16 engines)
  WarmSeq warm_;                    // weight warming schedule of the current plan (recorded on its first forward)
  float* attn_xws_[2] = {nullptr, nullptr}; unsigned* attn_xcnt_[2] = {nullptr, nullptr};   // cross-workgroup key split: per chain (split-CFG runs two)
  bool graph_warm_ = false;         // the captured graph carries the warming workgroups
  size_t attn_xcnt_bytes_ = 0;
  int demote_mask_ = 0, graph_demote_ = 0;                 // hl_demote classes in force since the last set_context / captured in the graph
  std::vector<std::pair<float*, float>> demote_flags_;      // (device address of a packed matrix's exact-f16 flag, its packed value) per class bit, filled lazily
  std::vector<int> demote_flag_cls_;
  void apply_demote_weights(hipStream_t s);
  bool gn_from_producer_ = true, plan_gn_ = true;   // GroupNorm statistics from the producing convolution's epilogue where its kernel can (f16); part of the plan key
  hipStream_t s2_ = nullptr; hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  DeviceArena act2_;
  hipGraphExec_t graph_ = nullptr;
  const float* graph_t_ = nullptr; int graph_ts_ = 0; int plan_runs_ = 0;
};

// ------------------------------------------------------------------------------------------ VAE
struct VaeResW { NormW n1, n2; Lin c1, c2, nin; bool has_nin = false; int cin = 0, cout = 0; };
struct VaeMidW { VaeResW b1, b2; NormW an; Lin q, k, v, proj; int C = 0; };

class Vae {
 public:
  Vae(const VaeCfg& cfg, int compute_dt, WeightSource* dec_src, WeightSource* enc_src, hipStream_t st);
  ~Vae();
  const VaeCfg& cfg() const { return cfg_; }
  // LatentDecoder::decode_latent (stablediffusion/mod.rs:263-266): latent NCHW fp32 [n,4,h,w] -> NHWC image [n][8h*8w][3]
  // (fp32, internal buffer); returns the buffer
  const float* decode(const float* latent_nchw, int n, int h, int w, hipStream_t s);
  void decode_nchw(const float* latent_nchw, int n, int h, int w, float* out_nchw, hipStream_t s);
  void latent_to_image(const float* latent_nchw, int n, int h, int w, unsigned char* out_hwc, hipStream_t s);
  // LatentDecoder::encode_image / image_to_latent (:239-261): -> latent NCHW fp32 [n,4,H/8,W/8]
  void encode_nchw(const float* img_nchw, int n, int H, int W, float* latent_out, hipStream_t s);
  void image_to_latent(const unsigned char* img_hwc, int n, int H, int W, float* latent_out, hipStream_t s);
  size_t weight_bytes() const { return warena_.off; }
  void* weight_base() const { return warena_.base; }

 private:
  void res_block(Exec& ex, const VaeResW& w, const Act& x, int B, int H, int W, const Act& out);
  void mid(Exec& ex, const VaeMidW& w, const Act& x, int B, int H, int W);
  void run_decode(Exec& ex, const Act& in, int n, int h, int w, const Act& out);
  void run_encode(Exec& ex, const Act& in, int n, int H, int W, const Act& out);
  VaeResW load_res(WeightBuilder& wb, const std::string& p, int cin, int cout);
  VaeMidW load_mid(WeightBuilder& wb, const std::string& p, int c);

  VaeCfg cfg_;
  int cdt_;
  DeviceArena warena_;
  bool has_dec_ = false, has_enc_ = false;
  // decoder
  Lin post_quant_, d_conv_in_, d_conv_out_; NormW d_norm_out_; VaeMidW d_mid_;
  struct DecBlk { VaeResW r[3]; Lin up; bool has_up = false; };
  std::vector<DecBlk> d_blocks_;
  // encoder
  Lin quant_, e_conv_in_, e_conv_out_; NormW e_norm_out_; VaeMidW e_mid_;
  struct EncBlk { VaeResW r[2]; Lin down; bool has_down = false; };
  std::vector<EncBlk> e_blocks_;
  DeviceArena act_;
  size_t act_peak_ = 0;
  float* gn_workspace(Exec& ex, int n);
  // dtype of the residual stream and of the 3- / 4- / 8-channel ends: the compute dtype, except that the split-operand mode
  // (DT_HL operands for the GEMMs) keeps them in plain fp32
  int io_dt() const { return cdt_ == DT_HL ? DT_F32 : cdt_; }
};

// ------------------------------------------------------------------------------------------ CLIP text encoder (Embedder)
struct ClipBlockW { NormW attn_ln, mlp_ln; Lin qkv, out, fc1, fc2; };

// CLIP<B> of the reference (clip/mod.rs:62-151): token + position embedding, n_layer pre-LN causal transformer blocks,
// final LayerNorm + eot pooling + text projection.  Runs once per prompt (Embedder::text_to_conditioning,
// stablediffusion/mod.rs:661-770), so it is launch/weight-bandwidth bound: the weights are streamed once per call.
class ClipText {
 public:
  ClipText(const ClipCfg& cfg, int compute_dt, int stream_dt, WeightSource& src, hipStream_t st);
  ~ClipText();
  void set_use_graph(bool g) { use_graph_ = g; }
  const ClipCfg& cfg() const { return cfg_; }
  // CLIP::forward_hidden (:94-112): ids int32 device [B][S] -> out fp32 device [B][S][n_state], the residual stream after
  // the first hidden_idx blocks (no final LayerNorm)
  void forward_hidden(const int* ids, int B, int S, int hidden_idx, float* out, hipStream_t s);
  // CLIP::forward_hidden_pooled (:114-151): hidden = input of block hidden_idx; pooled fp32 [B][embed_dim] =
  // LayerNorm(last)[argmax ids] . text_projection
  void forward_hidden_pooled(const int* ids, int B, int S, int hidden_idx, float* hidden, float* pooled, hipStream_t s);
  size_t weight_bytes() const { return warena_.off; }
  void* weight_base() const { return warena_.base; }

 private:
  void run(const int* ids, int B, int S, int n_blocks, int tap, float* hidden, float* pooled, hipStream_t s);
  void block(Exec& ex, const ClipBlockW& w, const Act& x, int B, int S, const Act& ln, const Act& qk, void* vt, int npad,
             const Act& ao, const Act& h);
  ClipCfg cfg_;
  int cdt_, sdt_;
  DeviceArena warena_, act_;
  const void* tok_ = nullptr; const void* pos_ = nullptr;
  std::vector<ClipBlockW> blocks_;
  NormW final_ln_; Lin proj_;
  float* mask_ = nullptr;                    // causal mask [S][S] (lives in act_)
  hipGraphExec_t graph_ = nullptr; long key_[6] = {0, 0, 0, 0, 0, 0}; int runs_ = 0; bool use_graph_ = true;
};

// ------------------------------------------------------------------------------------------ sampler
// mirrors Conditioning<B> (stablediffusion/mod.rs:544-555); all device fp32
struct Conditioning {
  const float* unconditional_context_full = nullptr;        // [77][ctx_full]
  const float* unconditional_context_open_clip = nullptr;   // [77][1280]
  const float* context_full = nullptr;                      // [n][77][ctx_full]
  const float* context_open_clip = nullptr;                 // [n][77][1280]
  const float* unconditional_channel_context = nullptr;     // [adm]
  const float* unconditional_channel_context_refiner = nullptr;
  const float* channel_context = nullptr;                   // [n][adm]
  const float* channel_context_refiner = nullptr;
  int n = 1, n_ctx = 77, height = 1024, width = 1024;
};

#ifdef SDXL_MEASURE
extern bool g_debug_no_cfg;   // sdxl_debug_set("no_cfg"): base model without the unconditional branch (changes the semantics:
                              // measurement builds only)
#endif

class Diffuser {
 public:
  Diffuser(const UNetCfg& cfg, int compute_dt, int stream_dt, WeightSource& src, const float* alphas_host, int n_train,
           hipStream_t st, int mix = 0);
  ~Diffuser();
  UNet& unet() { return *unet_; }
  // Diffuser::sample_latent (stablediffusion/mod.rs:317-332); noise0 plays gen_noise(); out: NCHW fp32 [n,4,h/8,w/8]
  void sample_latent(const Conditioning& c, double cfg_scale, int n_steps, const float* noise0, float* out, hipStream_t s);
  // Diffuser::sample_latent_with_inpainting (:334-353, :434-483); mask u8 (1 = keep generated), step_noise [iters][n,4,h,w]
  void sample_latent_inpaint(const Conditioning& c, double cfg_scale, int n_steps, const float* reference,
                             const unsigned char* mask, const float* noise0, const float* step_noise, float* out,
                             hipStream_t s);
  // Diffuser::refine_latent (:355-376)
  void refine_latent(const float* latent, const Conditioning& c, double cfg_scale, int step_start, int n_steps,
                     const float* noise, float* out, hipStream_t s);
  static std::vector<int> step_schedule(int n_steps, int step_start, int n_train);
  std::vector<float> step_ms;   // per-iteration GPU time of the last trajectory (hipEvent), for "UNet step ms p50"
  bool time_steps = false;
  // parity instrumentation: after DDIM iteration i the latent [n,4,h,w] is copied to trace + i * numel (device, caller-owned)
  // for i < trace_cap -- the per-step latents the oracle's `trace` lists hold (drift reports, tests)
  float* trace = nullptr; int trace_cap = 0;

 private:
  void diffuse(float* latent, const Conditioning& c, int step_start, int n_steps, double cfg_scale, const float* reference,
               const unsigned char* mask, const float* step_noise, hipStream_t s);
  std::unique_ptr<UNet> unet_;
  std::vector<double> alphas_;
  int n_train_;
  bool is_refiner_;
  // device state
  float* latent_ = nullptr; size_t latent_cap_ = 0;
  StepCoef* table_ = nullptr; int table_cap_ = 0;
  int* step_idx_ = nullptr; float* t_dev_ = nullptr;
  float* ctx_buf_ = nullptr; size_t ctx_cap_ = 0;     // [2n][77][ctx] cond then uncond
  float* y_buf_ = nullptr; size_t y_cap_ = 0;
  const void* cached_ctx_key_[4] = {nullptr, nullptr, nullptr, nullptr}; int cached_n_ = 0;
};

}  // namespace sdxl
