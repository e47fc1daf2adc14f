// SDXL UNet forward on MI355X -- the sampling loop's hot path (reference UNet::forward, unet/mod.rs:450-492).
//
// Design (not a translation of the burn module tree):
//   * NHWC / token-major everywhere, so ResBlock convs, the SpatialTransformer's linears and attention share one
//     activation layout and the reference's NCHW<->[B,N,C] transposes (unet/mod.rs:827-829,837-841) disappear.
//   * torch.cat of the skip connections (:484) is eliminated: every input block writes its output straight into the
//     channel slice [C_x, C_x+C_skip) of the concat buffer its matching output block will read, and the previous
//     output block writes the [0, C_x) slice.
//   * nearest-2x upsample (:744-749) is an index >>1 inside the conv's gather; the time-embedding add (:1092), the
//     residual adds (:1098-1102, :887-889, :843) and GEGLU (:944-955) live in GEMM epilogues.
//   * everything that is constant over a trajectory is hoisted out of the step: cross-attention K / V^T projections of
//     the 70 transformer blocks and the label-embedding MLP are computed once per prompt (set_context).
//   * the 17+ per-ResBlock lin_embed(silu(emb)) GEMVs (:1088-1089) are one GEMV over a concatenated weight.
//   * static shapes -> bump-allocated activation arena with scoped reuse (working set stays inside the 256 MB
//     Infinity Cache) -> stable addresses -> the whole forward is captured once into a hipGraph and replayed.
#include "engine.h"

#include <atomic>
#include <cmath>
#include <cstring>

namespace sdxl {

namespace {
ResBlockW load_res(WeightBuilder& wb, const std::string& p, int cin, int cout, std::vector<std::string>& emb_names,
                   int& emb_off) {
  ResBlockW r;
  r.cin = cin; r.cout = cout;
  r.norm_in = wb.norm(p + ".norm_in");
  r.conv_in = wb.conv(p + ".conv_in");
  r.norm_out = wb.norm(p + ".norm_out");
  r.conv_out = wb.conv(p + ".conv_out");
  r.has_skip = wb.has(p + ".skip_connection.weight");
  if (r.has_skip) r.skip = wb.conv(p + ".skip_connection");
  r.emb_off = emb_off;
  emb_names.push_back(p + ".lin_embed");
  emb_off += cout;
  return r;
}
bool spec_k_ok(WeightBuilder& wb, const std::string& name) { return wb.spec(name + ".weight").shape[0] % 32 == 0; }
STW load_st(WeightBuilder& wb, const std::string& p, int C, int heads, int depth, bool fuse_ln, int mix = 0) {
  const bool geglu_f16 = (mix & MIX_GEGLU_F16) != 0, qkv_f16 = (mix & MIX_QKV_F16) != 0, ff_f16 = (mix & MIX_FF_F16) != 0, out1_f16 = (mix & MIX_OUT1_F16) != 0,
             out2_f16 = (mix & MIX_OUT2_F16) != 0, q2_fused = out2_f16 && (mix & MIX_XATTN_F16) != 0, q2_f16 = q2_fused || (mix & MIX_Q2_F16) != 0,
             ln_sh = (mix & MIX_LN_SHADOW) != 0, x2 = (mix & MIX_LINEAR_F16X2) != 0;
  STW s;
  s.C = C; s.heads = heads;
  s.norm = wb.norm(p + ".norm");
  s.proj_in = wb.linear(p + ".proj_in");
  for (int j = 0; j < depth; ++j) {
    const std::string q = p + ".blocks." + std::to_string(j);
    TBlockW t;
    if (fuse_ln) {
      // the three LayerNorms of TransformerBlock::forward (unet/mod.rs:885-891) are folded into the projections that
      // consume them; their row statistics come out of the epilogue of the GEMM that produced the residual stream
      t.qkv = wb.fused_linear_ln({q + ".attn1.query", q + ".attn1.key", q + ".attn1.value"}, q + ".norm1");
      t.out1 = wb.linear(q + ".attn1.out");
      t.q2 = wb.linear_ln(q + ".attn2.query", false, q + ".norm2");
      t.kv2 = wb.fused_linear({q + ".attn2.key", q + ".attn2.value"});
      t.out2 = wb.linear(q + ".attn2.out");
      t.geglu = wb.linear_ln(q + ".mlp.geglu.proj", true, q + ".norm3");
      t.ff = wb.linear(q + ".mlp.lin");
      s.blocks.push_back(t);
      continue;
    }
    t.n1 = wb.norm(q + ".norm1");
    // MIX_LN_SHADOW: the f16 projections behind a LayerNorm also exist in the shadow form (same packed matrix, cs = gamma W, b = beta W + bias): where the
    // producer of the stream left the f16 shadow f16(x o gamma) and the row statistics, the LayerNorm launch is skipped (spatial_transformer)
    if (ln_sh && qkv_f16) t.qkv_sh = wb.fold_ln({q + ".attn1.query", q + ".attn1.key", q + ".attn1.value"}, q + ".norm1", false, DT_F16, true, &t.qkv);
    else if (ln_sh && !qkv_f16 && x2 && wb.spec(q + ".attn1.query.weight").shape[0] % 64 == 0 && wb.spec(q + ".attn1.query.weight").shape[1] % 128 == 0)
      // MIX_LINEAR_F16X2 + MIX_LN_SHADOW: fp32-class projection on the f16 kernels over an HL16 shadow (k_form 2); the plain twin reads the LayerNorm launch's HL16 output
      t.qkv_sh = wb.fold_ln({q + ".attn1.query", q + ".attn1.key", q + ".attn1.value"}, q + ".norm1", false, DT_F16, true, &t.qkv, false, true);
    else
    t.qkv = wb.fused_linear({q + ".attn1.query", q + ".attn1.key", q + ".attn1.value"}, qkv_f16 ? (int)DT_F16 : -1);
    // MIX_LINEAR_F16X2: a projection that stays fp32-class (its A operand an HL16 tensor) runs on the f16 kernels over the weight packed twice in the HL16 interleave
    auto x2_ok = [&](const std::string& nm) { return x2 && wb.spec(nm + ".weight").shape[0] % 32 == 0 && wb.spec(nm + ".weight").shape[1] % 128 == 0; };
    t.out1 = !out1_f16 && x2_ok(q + ".attn1.out") ? wb.linear_hilo(q + ".attn1.out", false, false, true) : wb.linear(q + ".attn1.out", false, out1_f16 ? (int)DT_F16 : -1);
    t.n2 = wb.norm(q + ".norm2");
    if (ln_sh && q2_f16 && !q2_fused) t.q2_sh = wb.fold_ln({q + ".attn2.query"}, q + ".norm2", false, DT_F16, true, &t.q2);
    else if (!q2_f16 && ln_sh && x2_ok(q + ".attn2.query") && wb.spec(q + ".attn2.query.weight").shape[0] % 64 == 0)
      t.q2_sh = wb.fold_ln({q + ".attn2.query"}, q + ".norm2", false, DT_F16, true, &t.q2, false, true);
    else if (!q2_f16 && x2_ok(q + ".attn2.query")) t.q2 = wb.linear_hilo(q + ".attn2.query", false, false, true);
    else
    t.q2 = wb.linear(q + ".attn2.query", false, q2_f16 ? (int)DT_F16 : -1);
    t.kv2 = wb.fused_linear({q + ".attn2.key", q + ".attn2.value"});
    t.out2 = !out2_f16 && x2_ok(q + ".attn2.out") ? wb.linear_hilo(q + ".attn2.out", false, false, true) : wb.linear(q + ".attn2.out", false, out2_f16 ? (int)DT_F16 : -1);
    t.n3 = wb.norm(q + ".norm3");
    // (mixed mode: the GEGLU projection of a split-operand model packed as plain f16 -- it runs on the f16 wide-tile kernel)
    if (geglu_f16 && (mix & MIX_GEGLU_AHILO) && spec_k_ok(wb, q + ".mlp.geglu.proj")) {
      // activations as (hi | lo 2^8) along a doubled K against (w | w 2^-8); with MIX_LN_SHADOW the producer's shadow carries both halves
      if (ln_sh && wb.spec(q + ".mlp.geglu.proj.weight").shape[0] % 64 == 0) t.geglu_sh = wb.fold_ln({q + ".mlp.geglu.proj"}, q + ".norm3", true, DT_F16, true, &t.geglu, true);
      else t.geglu = wb.linear_hilo(q + ".mlp.geglu.proj", true, true);
    }
    else if (ln_sh && geglu_f16) t.geglu_sh = wb.fold_ln({q + ".mlp.geglu.proj"}, q + ".norm3", true, DT_F16, true, &t.geglu);
    else if (geglu_f16 && (mix & MIX_GEGLU_HILO) && spec_k_ok(wb, q + ".mlp.geglu.proj")) t.geglu = wb.linear_hilo(q + ".mlp.geglu.proj", true);
    else if (!geglu_f16 && x2 && wb.spec(q + ".mlp.geglu.proj.weight").shape[0] % 32 == 0 && wb.spec(q + ".mlp.geglu.proj.weight").shape[1] % 640 == 0) {
      // fp32-class GEGLU on the f16 wide-tile kernel (HL16 operand read as f16, K = 2 C); with MIX_LN_SHADOW also in the shadow form
      if (ln_sh && wb.spec(q + ".mlp.geglu.proj.weight").shape[0] % 64 == 0) t.geglu_sh = wb.fold_ln({q + ".mlp.geglu.proj"}, q + ".norm3", true, DT_F16, true, &t.geglu, false, true);
      else t.geglu = wb.linear_hilo(q + ".mlp.geglu.proj", true, false, true);
    }
    else
    t.geglu = wb.linear(q + ".mlp.geglu.proj", true, geglu_f16 ? (int)DT_F16 : -1);
    t.ff = !ff_f16 && x2_ok(q + ".mlp.lin") ? wb.linear_hilo(q + ".mlp.lin", false, false, true) : wb.linear(q + ".mlp.lin", false, ff_f16 ? (int)DT_F16 : -1);
    s.blocks.push_back(t);
  }
  s.proj_out = wb.linear(p + ".proj_out");
  return s;
}
void gemv(Exec& ex, const Lin& w, const float* x, int ldx, float* y, int ldy, int Bm, bool silu_in, bool silu_out,
          const float* yadd = nullptr) {
  if (ex.dry) return;
  GemvParams p{};
  p.X = x; p.ldx = ldx; p.W = w.w; p.w_dt = w.dt >= 0 ? w.dt : ex.cdt; p.Kpad = w.Kpad; p.bias = w.b;
  p.Y = y; p.ldy = ldy; p.Yadd = yadd; p.Bm = Bm; p.N = w.N; p.K = w.K;
  p.silu_in = silu_in; p.silu_out = silu_out;
  if (ex.prof) ex.prof->begin(Profiler::OTHER, 2.0 * Bm * (double)w.N * w.K, ex.s);
  launch_gemv(p, ex.s);
  if (ex.prof) ex.prof->end(ex.s);
}
void attention(Exec& ex, const Act& q, const Act& k, const void* vt, int vt_ld, const Act& o, int B, int H, int Nq,
               int Nk, int tag = 0) {
  if (ex.dry) return;
  AttnParams p{};
  p.Q = q.p; p.ldq = q.ld; p.K = k.p; p.ldk = k.ld; p.Vt = vt; p.vt_ld = vt_ld; p.O = o.p; p.ldo = o.ld;
  // (the split-operand mode runs the attention on fp32 tensors: q / k / V^T / o all carry q's dtype)
  p.dt = q.dt; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
  if (Nq == Nk) { p.xws = ex.attn_xws; p.xcnt = ex.attn_xcnt; }      // self-attention: room for the cross-workgroup key split (sized for it in ensure_plan)
  if (ex.prof) ex.prof->begin(Profiler::ATTENTION, 4.0 * B * H * (double)Nq * Nk * 64, ex.s, Nq, Nk, B * H, 0, tag ? tag : (Nq == Nk ? DM_ATTN : DM_XATTN));
  launch_attention_d64(p, ex.s);
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw Error(std::string("attention launch failed (") + hipGetErrorString(le) + "): Nq=" + std::to_string(Nq) + " Nk=" + std::to_string(Nk) + " dt=" + std::to_string(p.dt));
  }
  if (ex.prof) ex.prof->end(ex.s);
}
Epi tag_epi(int cls) { Epi e; e.cls = cls; return e; }      // plain epilogue carrying only the launch's class label
// hl_demote: lo halves of an HL16 operand := 0 when its consumer's class is demoted (no-op otherwise / for other dtypes)
void demote_lo(Exec& ex, int cls, const Act& a, size_t rows, int C) {
  if (ex.dry || !(ex.demote & cls) || a.dt != DT_HL) return;
  launch_hl_zero_lo(a.p, a.ld, rows, C, ex.s);
}
Act hl_op(Exec& ex, const Lin& w, const Act& x, size_t rows, int C, int cls, int nb, float* have_max = nullptr) {      // hl_operand + the demotion of the copy when the consumer's class asks
  Act o = hl_operand(ex, w, x, rows, C, nb, have_max);
  if (o.p != x.p) demote_lo(ex, cls, o, rows, C);
  return o;
}
// split-operand mode: q / o fp32, K [B][Nk][ldk] and V^T [B][H*64][vt_ld] in HL16 (attn_d64_hl_kernel)
void attention_hl(Exec& ex, const Act& q, const void* kh, int ldk, const void* vth, int vt_ld, const Act& o, int B, int H, int Nq, int Nk, int demote_cls = 0) {
  if (ex.dry) return;
  AttnParams p{};
  p.demote = (ex.demote & demote_cls) ? 1 : 0;
  p.Q = q.p; p.ldq = q.ld; p.K = kh; p.ldk = ldk; p.Vt = vth; p.vt_ld = vt_ld; p.O = o.p; p.ldo = o.ld;
  p.dt = DT_HL; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.mask = nullptr; p.ldmask = 0;
  p.o_dt = o.dt == DT_HL ? DT_HL : o.dt == DT_F16 ? DT_F16 : DT_F32;
  p.q_dt = q.dt == DT_HL ? DT_HL : DT_F32;
  if (ex.prof) ex.prof->begin(Profiler::ATTENTION, 4.0 * B * H * (double)Nq * Nk * 64, ex.s, Nq, Nk, B * H, 0, demote_cls);
  if (!launch_attention_d64_hl(p, ex.s)) throw Error("split-operand attention: unsupported shape / alignment (Nq=" + std::to_string(Nq) + " Nk=" + std::to_string(Nk) + ")");
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) throw Error(std::string("split-operand attention launch failed (") + hipGetErrorString(le) + ")");
  }
  if (ex.prof) ex.prof->end(ex.s);
}
}  // namespace

static std::atomic<int> g_mix_classes{-1};     // A/B / debugging knob (sdxl_debug_set "mix_classes"): overrides the MixClass bits of SDXL_DTYPE_F32_SPLIT_MIX models built afterwards (-1 = the mode's own)
void unet_set_mix_classes(int v) { g_mix_classes = v; }
static std::atomic<int> g_hl_demote{0};
void unet_set_hl_demote(int mask) { g_hl_demote = mask; }
int unet_hl_demote() { return g_hl_demote.load(); }

// the exact-f16 flag (acc_scale[1]) of every packed matrix of a demoted class reads 1 -- the kernel then leaves its w_lo MFMAs out --
// and its packed value otherwise; the packed values are read back once
void UNet::apply_demote_weights(hipStream_t s) {
  if (cdt_ != DT_HL) return;
  if (demote_flags_.empty()) {
    auto add = [&](const Lin& l, int cls) {
      if (!l.acc_scale) return;
      float v = 0.f;
      SDXL_HIP(hipMemcpy(&v, l.acc_scale + 1, sizeof(float), hipMemcpyDeviceToHost));
      demote_flags_.push_back({const_cast<float*>(l.acc_scale) + 1, v}); demote_flag_cls_.push_back(cls);
    };
    auto add_res = [&](const ResBlockW& r) { add(r.conv_in, DM_CONV_RES); add(r.conv_out, DM_CONV_RES); if (r.has_skip) add(r.skip, DM_CONV_SKIP); };
    auto add_st = [&](const STW& st) {
      add(st.proj_in, DM_CONV_PROJ); add(st.proj_out, DM_CONV_PROJ);
      for (const TBlockW& b : st.blocks) {
        add(b.qkv, DM_QKV); add(b.out1, DM_OUT); add(b.q2, DM_XATTN); add(b.kv2, DM_XATTN); add(b.out2, DM_OUT); add(b.geglu, DM_GEGLU); add(b.ff, DM_FF);
      }
    };
    auto add_block = [&](const BlockW& b) { add(b.conv, b.d.kind == BK_CONV ? DM_CONV_IO : DM_CONV_UPDOWN); add_res(b.res); add_st(b.st); };
    for (const BlockW& b : inp_) add_block(b);
    add_block(mid_res1_); add_res(mid_res2_.res);
    for (const BlockW& b : out_) add_block(b);
    add(conv_out_, DM_CONV_IO);
  }
  std::vector<float> vals(demote_flags_.size());
  for (size_t i = 0; i < demote_flags_.size(); ++i) {
    vals[i] = (demote_mask_ & demote_flag_cls_[i]) ? 1.0f : demote_flags_[i].second;
    SDXL_HIP(hipMemcpyAsync(demote_flags_[i].first, &vals[i], sizeof(float), hipMemcpyHostToDevice, s));
  }
  SDXL_HIP(hipStreamSynchronize(s));     // (vals is host memory of this frame)
}

UNet::UNet(const UNetCfg& cfg, int compute_dt, int stream_dt, WeightSource& src, hipStream_t st, int mix)
    : cfg_(cfg), cdt_(compute_dt), sdt_(stream_dt), mix_(compute_dt == DT_HL ? (mix && g_mix_classes.load() >= 0 ? g_mix_classes.load() : mix) : 0),
      mix_knob_(compute_dt == DT_HL && mix && g_mix_classes.load() >= 0) {
  SDXL_REQUIRE(cfg.n_head_channels == 64, "this engine's fused attention kernel is specialised for 64 channels per head");
  SDXL_REQUIRE(!((compute_dt == DT_F32 || compute_dt == DT_HL) && stream_dt != DT_F32), "f32 / split-operand compute implies an f32 residual stream");
  SDXL_REQUIRE(cfg.model_channels % 32 == 0, "GroupNorm(32) needs model_channels % 32 == 0");
  fuse_ln_ = compute_dt == DT_F16 && stream_dt == DT_F16;
  build_weights(src, st);
}
UNet::~UNet() {
  if (graph_) (void)hipGraphExecDestroy(graph_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (s2_) (void)hipStreamDestroy(s2_);
}

void UNet::build_weights(WeightSource& src, hipStream_t st) {
  const std::vector<ParamSpec> specs = unet_param_specs(cfg_);
  size_t bound = WeightBuilder::arena_bound(specs, cdt_);
  if (cdt_ == DT_HL && (mix_ & MIX_LINEAR_F16X2))      // f16 x 2 K images carry a fragment-order twin the split-operand bound does not count
    for (const ParamSpec& ps : specs) if (ps.kind == PK_LINEAR_W) bound += round_up(ps.shape[1], 128) * round_up((size_t)ps.shape[0], 64) * 4 + 256;
  warena_.reserve(bound);
  WeightBuilder wb(specs, src, warena_, cdt_, st);
  // SDXL_DTYPE_F32_SPLIT_MIX_F16W moves classes to plain f16 operands whose WEIGHTS must be f16 values (the reference's records are,
  // src/bin/sample/main.rs:37): on other parameters that would round the weights as well and leave the mode's error bound (0.029 against
  // 0.0212).  Checked here on the tensors themselves; a model that does not qualify falls back to F32_SPLIT_MIX's two classes (mix_classes() tells).
  // Not applied to the A/B knob "mix_classes" (the frontier tools run those maps on fp32 weights on purpose) nor on replicas built from an empty
  // source (they receive rank 0's arena: the caller compares rank 0's mix_classes() with the mode before the broadcast, bench.py does).
  constexpr int kNeedExact = MIX_QKV_F16 | MIX_FF_F16 | MIX_OUT1_F16 | MIX_OUT2_F16 | MIX_XATTN_F16 | MIX_Q2_F16 | MIX_GEGLU_AHILO | MIX_LINEAR_F16X2;
  if (cdt_ == DT_HL && (mix_ & kNeedExact) && !mix_knob_ && !src.empty()) {
    std::vector<std::string> names;
    auto ends = [](const std::string& n, const char* suf) { const size_t l = std::strlen(suf); return n.size() >= l && n.compare(n.size() - l, l, suf) == 0; };
    for (const ParamSpec& ps : specs) {
      const std::string& n = ps.name;
      if (n.find(".transformer.blocks.") == std::string::npos) continue;
      if (((mix_ & MIX_QKV_F16) && (ends(n, ".attn1.query.weight") || ends(n, ".attn1.key.weight") || ends(n, ".attn1.value.weight"))) ||
          ((mix_ & MIX_OUT1_F16) && ends(n, ".attn1.out.weight")) || ((mix_ & MIX_OUT2_F16) && ends(n, ".attn2.out.weight")) ||
          ((mix_ & (MIX_FF_F16 | MIX_LINEAR_F16X2)) && ends(n, ".mlp.lin.weight")) || ((mix_ & MIX_LINEAR_F16X2) && (ends(n, ".attn1.out.weight") || ends(n, ".attn2.out.weight"))) || ((mix_ & (MIX_GEGLU_AHILO | MIX_LINEAR_F16X2)) && ends(n, ".mlp.geglu.proj.weight")) || ((mix_ & (MIX_XATTN_F16 | MIX_Q2_F16 | MIX_LINEAR_F16X2)) && ends(n, ".attn2.query.weight")))
        names.push_back(n);
    }
    if (!wb.all_f16_exact(names))      // = SDXL_DTYPE_F32_SPLIT_MIX (capi.hip mix_of); a mode without f16-OPERAND classes (SDXL_DTYPE_F32_SPLIT_F16W) falls back to plain F32_SPLIT
      mix_ = (mix_ & (MIX_ATTN_F16 | MIX_GEGLU_F16)) ? (MIX_ATTN_F16 | MIX_GEGLU_F16 | MIX_GEGLU_HILO) : 0;
  }
  const int gv = cdt_ == DT_HL ? DT_F32 : -1;     // the M <= 8 GEMV weights of a split-operand model are packed fp32
  lin1_t_ = wb.linear("lin1_time_embed", false, gv);
  lin2_t_ = wb.linear("lin2_time_embed", false, gv);
  lin1_l_ = wb.linear("lin1_label_embed", false, gv);
  lin2_l_ = wb.linear("lin2_label_embed", false, gv);
  std::vector<BlockDesc> inp, out; BlockDesc mid;
  unet_block_plan(cfg_, inp, mid, out);
  std::vector<std::string> emb_names;
  int emb_off = 0;
  auto load_block = [&](const std::string& p, const BlockDesc& d) {
    BlockW b; b.d = d;
    switch (d.kind) {
      case BK_CONV: case BK_DOWN: b.conv = wb.conv(p); break;
      case BK_RES: b.res = load_res(wb, p, d.c_in, d.c_out, emb_names, emb_off); break;
      default:
        b.res = load_res(wb, p + ".res", d.c_in, d.c_out, emb_names, emb_off);
        if (d.kind == BK_REST || d.kind == BK_RESTU) b.st = load_st(wb, p + ".transformer", d.c_out, d.n_head, d.depth, fuse_ln_, mix_);
        if (d.kind == BK_RESTU || d.kind == BK_RESU) b.conv = wb.conv(p + ".upsample.conv");
    }
    return b;
  };
  for (size_t i = 0; i < inp.size(); ++i) inp_.push_back(load_block("input_blocks." + std::to_string(i), inp[i]));
  mid_res1_.d = mid;
  mid_res1_.res = load_res(wb, "middle_block.res1", mid.c_in, mid.c_out, emb_names, emb_off);
  mid_res1_.st = load_st(wb, "middle_block.transformer", mid.c_out, mid.n_head, mid.depth, fuse_ln_, mix_);
  mid_res2_.d = mid;
  mid_res2_.res = load_res(wb, "middle_block.res2", mid.c_in, mid.c_out, emb_names, emb_off);
  for (size_t i = 0; i < out.size(); ++i) out_.push_back(load_block("output_blocks." + std::to_string(i), out[i]));
  norm_out_ = wb.norm("norm_out");
  conv_out_ = wb.conv("conv_out");
  embcat_ = wb.fused_linear(emb_names, gv);
  emb_total_ = emb_off;
  SDXL_HIP(hipStreamSynchronize(st));
  // execution order of the spatial transformers (for the K/V caches)
  for (BlockW& b : inp_) if (!b.st.blocks.empty()) st_list_.push_back(&b.st);
  st_list_.push_back(&mid_res1_.st);
  for (BlockW& b : out_) if (!b.st.blocks.empty()) st_list_.push_back(&b.st);
}

// ------------------------------------------------------------------------------------------ conditioning
void UNet::set_context(const float* context, int n_ctx, const float* label, int B, hipStream_t s) {
  SDXL_REQUIRE(B >= 1 && B <= 8, "batch must be in 1..8");
  const int vt_ld = (int)round_up(n_ctx, 64);
  // operand-order copies for the fused cross-attention epilogue (f16 engines; the split-operand engine when that class runs on f16: MIX_XATTN_F16)
  // (MIX_XATTN_SPLIT: TWO images -- the hi and the lo halves of the fp32-class projection -- for the split-precision form of that epilogue)
  const bool pack_xs = cdt_ == DT_HL && (mix_ & MIX_XATTN_SPLIT) && (mix_ & (MIX_Q2_F16 | MIX_LINEAR_F16X2)) && !(mix_ & MIX_XATTN_F16) && n_ctx <= 96;
  const bool pack_xa = (cdt_ == DT_F16 || (cdt_ == DT_HL && (mix_ & MIX_XATTN_F16) && (mix_ & MIX_OUT2_F16)) || pack_xs) && n_ctx <= 96;
  const int kvdt = attn_dt();                           // dtype of the K / V^T caches (fp32 in the split-operand mode)
  const int emb = 4 * cfg_.model_channels;
  {   // precision-frontier instrument: the demoted classes take effect from here (weights now, activations in every forward after)
    const int dm = cdt_ == DT_HL ? unet_hl_demote() : 0;
    if (dm != demote_mask_ || (dm && demote_flags_.empty())) { demote_mask_ = dm; apply_demote_weights(s); }
  }
  if (B != ctx_B_ || n_ctx != n_ctx_) {
    // (re)allocate the caches; captured graphs hold these addresses
    if (graph_) { (void)hipGraphExecDestroy(graph_); graph_ = nullptr; plan_runs_ = 0; }
    // the recorded warming schedule holds raw pointers into ctx_arena_ (the packed context of the fused cross-attention
    // projections is a warm target): a re-laid-out / reallocated arena invalidates them -> record again on the next eager forward
    warm_ = WarmSeq();
    plan_runs_ = 0;
    size_t bytes = 1 << 16;
    for (const STW* st : st_list_)
      bytes += st->blocks.size() * (round_up((size_t)B * n_ctx * st->C * dt_size(kvdt), 256) +
                                    round_up((size_t)B * st->C * vt_ld * dt_size(kvdt), 256) +
                                    (pack_xa ? round_up(xattn_pack_bytes(B, st->C), 256) : 0) + (pack_xs ? round_up(xattn_pack_bytes(B, st->C), 256) : 0) + 768);
    bytes += 3 * round_up((size_t)B * emb * sizeof(float), 256);
    if (cdt_ == DT_HL) bytes += round_up((size_t)B * n_ctx * cfg_.context_dim * 4, 256) + 256;   // HL16 copy of the context
    if (cdt_ == DT_HL) {   // fp32 scratch of one block's K / V^T projection: the caches themselves are HL16 (what the attention kernel reads)
      size_t mx = 0;
      for (const STW* st : st_list_) mx = std::max(mx, round_up((size_t)B * n_ctx * st->C * 4, 256) + round_up((size_t)B * st->C * vt_ld * 4, 256));
      bytes += (pack_xs ? 2 * mx + 1024 : pack_xa ? mx + mx / 2 + 512 : mx) + 512;     // (+ f16 copies of both -- hi and lo with MIX_XATTN_SPLIT -- when the fused cross-attention reads them packed)
    }
    ctx_arena_.reserve(bytes);
    ctx_arena_.off = 0;
    SDXL_HIP(hipMemsetAsync(ctx_arena_.base, 0, bytes, s));   // V^T key padding must be zero
    kv_.clear();
    for (const STW* st : st_list_) {
      std::vector<KV> v;
      for (size_t j = 0; j < st->blocks.size(); ++j) {
        KV kv;
        kv.k = ctx_arena_.alloc((size_t)B * n_ctx * st->C * dt_size(kvdt));
        kv.vt = ctx_arena_.alloc((size_t)B * st->C * vt_ld * dt_size(kvdt));
        if (pack_xa) kv.xa = ctx_arena_.alloc(xattn_pack_bytes(B, st->C));
        if (pack_xs) kv.xa_lo = ctx_arena_.alloc(xattn_pack_bytes(B, st->C));
        v.push_back(kv);
      }
      kv_.push_back(v);
    }
    label_emb_ = (float*)ctx_arena_.alloc((size_t)B * emb * sizeof(float));
    ctx_B_ = B; n_ctx_ = n_ctx; vt_ld_ctx_ = vt_ld;
  }
  Exec ex; ex.s = s; ex.cdt = cdt_; ex.sdt = sdt_; ex.act = &ctx_arena_; ex.demote = demote_mask_;
  const size_t m0 = ctx_arena_.mark();
  Act ctx((void*)context, cfg_.context_dim, DT_F32);
  if (cdt_ == DT_HL && !st_list_.empty() && !st_list_[0]->blocks.empty()) {    // one HL16 copy of the context for all 70 projections
    ctx = hl_operand(ex, st_list_[0]->blocks[0].kv2, ctx, (size_t)B * n_ctx, cfg_.context_dim, B);
    demote_lo(ex, DM_XATTN, ctx, (size_t)B * n_ctx, cfg_.context_dim);
  }
  for (size_t si = 0; si < st_list_.size(); ++si) {
    const STW* st = st_list_[si];
    for (size_t j = 0; j < st->blocks.size(); ++j) {
      if (cdt_ == DT_HL) {
        const size_t ms = ctx_arena_.mark();
        void* k32 = ctx_arena_.alloc((size_t)B * n_ctx * st->C * 4);
        void* vt32 = ctx_arena_.alloc((size_t)B * st->C * vt_ld * 4);
        launch_fill_zero(vt32, (size_t)B * st->C * vt_ld * 4, s);                     // V^T key padding must be zero
        Epi e; e.n_split = st->C; e.Ct = vt32; e.ct_rows = st->C; e.ct_ld = vt_ld; e.rpb = n_ctx; e.cls = DM_XATTN;
        run_linear(ex, st->blocks[j].kv2, ctx, B * n_ctx, Act(k32, st->C, DT_F32), e);
        launch_f32_to_hl(k32, st->C, kv_[si][j].k, st->C, (size_t)B * n_ctx, st->C, s);
        launch_f32_to_hl(vt32, vt_ld, kv_[si][j].vt, vt_ld, (size_t)B * st->C, vt_ld, s);
        demote_lo(ex, DM_XATTN, Act(kv_[si][j].k, st->C, DT_HL), (size_t)B * n_ctx, st->C);
        demote_lo(ex, DM_XATTN, Act(kv_[si][j].vt, vt_ld, DT_HL), (size_t)B * st->C, vt_ld);
        if (kv_[si][j].xa_lo) {  // MIX_XATTN_SPLIT: hi = f16(x), lo = f16(x - hi) of the fp32-class projection, each in the operand order the fused launch reads
          void* kh = ctx_arena_.alloc((size_t)B * n_ctx * st->C * 2); void* kl = ctx_arena_.alloc((size_t)B * n_ctx * st->C * 2);
          void* vh = ctx_arena_.alloc((size_t)B * st->C * vt_ld * 2); void* vl = ctx_arena_.alloc((size_t)B * st->C * vt_ld * 2);
          launch_f32_to_f16_pair((const float*)k32, st->C, kh, kl, st->C, (size_t)B * n_ctx, st->C, s);
          launch_f32_to_f16_pair((const float*)vt32, vt_ld, vh, vl, vt_ld, (size_t)B * st->C, vt_ld, s);
          launch_xattn_pack(kh, vh, kv_[si][j].xa, B, st->C, n_ctx, vt_ld, s);
          launch_xattn_pack(kl, vl, kv_[si][j].xa_lo, B, st->C, n_ctx, vt_ld, s);
        } else
        if (kv_[si][j].xa) {     // MIX_XATTN_F16: the fp32-class projection rounded once to f16, in the operand order the fused launch reads
          void* k16 = ctx_arena_.alloc((size_t)B * n_ctx * st->C * 2);
          void* vt16 = ctx_arena_.alloc((size_t)B * st->C * vt_ld * 2);
          launch_copy_rows(k32, DT_F32, st->C, k16, DT_F16, st->C, B * n_ctx, st->C, s);
          launch_copy_rows(vt32, DT_F32, vt_ld, vt16, DT_F16, vt_ld, B * st->C, vt_ld, s);
          launch_xattn_pack(k16, vt16, kv_[si][j].xa, B, st->C, n_ctx, vt_ld, s);
        }
        ctx_arena_.reset(ms);
        continue;
      }
      Epi e; e.n_split = st->C; e.Ct = kv_[si][j].vt; e.ct_rows = st->C; e.ct_ld = vt_ld; e.rpb = n_ctx; e.cls = DM_XATTN;
      run_linear(ex, st->blocks[j].kv2, ctx, B * n_ctx, Act(kv_[si][j].k, st->C, kvdt), e);
      if (kv_[si][j].xa) launch_xattn_pack(kv_[si][j].k, kv_[si][j].vt, kv_[si][j].xa, B, st->C, n_ctx, vt_ld, s);
    }
  }
  ctx_arena_.reset(m0);
  // label embedding MLP (unet/mod.rs:464-466); scratch after the caches
  const size_t m = ctx_arena_.mark();
  float* l1 = (float*)ctx_arena_.alloc((size_t)B * emb * sizeof(float));
  gemv(ex, lin1_l_, label, cfg_.adm_in_channels, l1, emb, B, false, true);
  gemv(ex, lin2_l_, l1, emb, label_emb_, emb, B, false, false);
  ctx_arena_.reset(m);
}

// ------------------------------------------------------------------------------------------ blocks
const float* UNet::res_block(Exec& ex, const ResBlockW& w, const Act& x, int B, int H, int W, const Act& out, float* out_gn_part) {
  // ResBlock::forward unet/mod.rs:1082-1106.  GroupNorm statistics ride on the producing convolutions where their kernel can
  // leave them (256-row tiles: the 64^2 and 32^2 levels): the second norm never runs a statistics pass there, and the caller
  // gets the statistics of `out` (returned pointer, null if not produced) for the SpatialTransformer norm that follows.
  const size_t mk = ex.act->mark();
  const size_t M = (size_t)B * H * W;
  const int HW = H * W;
  const ConvGeom g3{B, H, W, H, W, 3, 1, 1, 0}, g1{B, H, W, H, W, 1, 1, 0, 0};
  const bool tiles256 = gn_from_producer_ && HW % 256 == 0;
  Act gn1 = ex.alloc(M, w.cin, ex.cdt);
  // split-operand engines: the 1x1 skip convolution reads x as an HL16 copy scaled per entry by max|x| -- the statistics pass of norm_in reads all
  // of x anyway and leaves the maxima, so that copy needs no absmax pass of its own (11 launches of a step)
  float* skip_max = nullptr;
  if (w.has_skip && ex.cdt == DT_HL && x.dt == DT_F32 && w.skip.dt != DT_F32 && !x.gn_part && M % (size_t)B == 0)
    skip_max = (float*)ex.act->alloc(hl_scale_floats(B) * sizeof(float));
  run_groupnorm(ex, w.norm_in, x, B, HW, gn1, true, 32, skip_max);
  demote_lo(ex, DM_CONV_RES, gn1, M, w.cin);
  Act h = ex.alloc(M, w.cout, ex.cdt == DT_HL ? ex.sdt : ex.cdt);     // (read by a GroupNorm only: fp32 in the split-operand mode)
  Epi e1; e1.ebias = ex.ebias + w.emb_off; e1.ebias_ld = emb_total_; e1.cls = DM_CONV_RES;
  if (tiles256) e1.gn_part = (float*)ex.act->alloc(M / 256 * (size_t)w.cout * 2 * sizeof(float));
  if (run_conv(ex, w.conv_in, gn1, w.cin, g3, h, e1)) { h.gn_part = e1.gn_part; h.gn_rt = HW / 256; }
  Act gn2 = ex.alloc(M, w.cout, ex.cdt);
  run_groupnorm(ex, w.norm_out, h, B, HW, gn2, true);
  demote_lo(ex, DM_CONV_RES, gn2, M, w.cout);
  Epi e2; e2.cls = DM_CONV_RES;
  if (w.has_skip) { Epi es; es.cls = DM_CONV_SKIP; run_conv(ex, w.skip, hl_op(ex, w.skip, x, M, w.cin, DM_CONV_SKIP, B, skip_max), w.cin, g1, out, es); e2.R = out; }
  else e2.R = x;
  if (tiles256) e2.gn_part = out_gn_part;
  const bool produced = run_conv(ex, w.conv_out, gn2, w.cout, g3, out, e2);
  ex.act->reset(mk);
  return produced ? out_gn_part : nullptr;
}

void UNet::spatial_transformer(Exec& ex, const STW& w, int si, const Act& x, int B, int H, int W) {
  // SpatialTransformer::forward unet/mod.rs:820-845, TransformerBlock::forward :885-891 -- in place on x
  const size_t mk = ex.act->mark();
  const int HW = H * W, C = w.C;
  const size_t M = (size_t)B * HW;
  SDXL_REQUIRE(C == w.heads * 64, "head dim must be 64");
  const int npad = (int)round_up(HW, 64);
  // cross-attention caches of this run's batch entries (dry runs carry null caches)
  const int adt = attn_dt();      // q / k / V^T / attention output: the compute dtype, fp32 in the split-operand mode
  auto kv_k = [&](int s_, size_t j) { char* k = (char*)kv_[s_][j].k; return (void*)(k ? k + (size_t)ex.b0 * n_ctx_ * C * dt_size(adt) : k); };
  auto kv_xa = [&](int s_, size_t j) { char* v = (char*)kv_[s_][j].xa; return (const void*)(v ? v + xattn_pack_bytes(ex.b0, C) : v); };
  auto kv_xa_lo = [&](int s_, size_t j) { char* v = (char*)kv_[s_][j].xa_lo; return (const void*)(v ? v + xattn_pack_bytes(ex.b0, C) : v); };
  auto kv_vt = [&](int s_, size_t j) { char* v = (char*)kv_[s_][j].vt; return (const void*)(v ? v + (size_t)ex.b0 * C * vt_ld_ctx_ * dt_size(adt) : v); };
  Act gn = ex.alloc(M, C, ex.cdt);
  run_groupnorm(ex, w.norm, x, B, HW, gn, false);
  demote_lo(ex, DM_CONV_PROJ, gn, M, C);
  Act t = ex.alloc(M, C, ex.sdt);
  // cross-attention fused into the query projection (f16 operands, <= 96 context tokens; igemm_xattn_ok)
  const bool xattn = plan_xattn_ && igemm_xattn_ok(fuse_ln_ ? ex.sdt : ex.cdt, ex.cdt, (int)M, C, C, HW, n_ctx_);
  // folded LayerNorms: two ping-pong [M][C/64][2] partial-sum buffers -- each is written by one GEMM and read by the next
  float* stbuf[2] = {nullptr, nullptr};
  if (fuse_ln_) for (int i = 0; i < 2; ++i) stbuf[i] = (float*)ex.act->alloc(M * (size_t)(C / 64) * 2 * sizeof(float));
  int stp = 0;
  { Epi ep; ep.stat_out = w.blocks.empty() ? nullptr : stbuf[stp]; ep.rpb = HW; ep.cls = DM_CONV_PROJ; run_linear(ex, w.proj_in, gn, (int)M, t, ep); }   // (rpb: kernel selection looks at ONE entry's rows)
  Act ln = ex.alloc(M, C, ex.cdt);
  // (split-operand mode: the projections write q | k and V^T as HL16 -- 4 bytes per element like fp32 -- what attention_hl reads)
  // (HL16 pieces are 8 keys wide: token counts that are not multiples of 8 -- tiny test nets -- go through fp32 + a conversion)
  const bool hl_direct = ex.cdt == DT_HL && HW % 8 == 0 && (2 * C) % 128 == 0;
  const int qdt = hl_direct ? DT_HL : adt;
  Act qk = ex.alloc(M, 2 * C, qdt);
  void* vt = ex.act->alloc((size_t)B * C * npad * dt_size(adt));
  Act ao = ex.alloc(M, C, ex.cdt);
  Act q = ex.alloc(M, C, qdt);
  Act gg = ex.alloc(M, 4 * C, ex.cdt);
  // (split-operand mode: attention_hl writes the out-projection's HL16 operand `ao` itself)
  if (npad != HW && !ex.dry) launch_fill_zero(vt, (size_t)B * C * npad * dt_size(adt), ex.s);
  // split-operand mode: the attention kernel takes K and V^T in HL16 (same bytes as fp32); q and the output stay fp32
  const bool hl_attn = ex.cdt == DT_HL;
  // mixed mode (SDXL_DTYPE_F32_SPLIT_MIX; classes chosen on the measured precision frontier, profiles/r05_precision_frontier.json):
  //   * self-attention on the f16 flash kernels: the (split-operand) QKV projection writes q | k and V^T as f16, the attention output is
  //     handed to the out-projection as HL16 with zero lo halves (an f16 value is its own hi half);
  //   * GEGLU projection on f16 operands (f16 LayerNorm output x f16-packed weights, the f16 wide-tile kernel) -- its output leaves the
  //     epilogue as HL16 (fp32-class), so FF-out's operand is not rounded a second time.
  const bool mix_qkv = hl_attn && !w.blocks.empty() && w.blocks[0].qkv.dt == DT_F16 && w.blocks[0].qkv.k_form == 0;     // (f16-packed at build: the only path those weights can take; k_form 2 = fp32-class, MIX_LINEAR_F16X2)
  const bool mix_ff = hl_attn && !w.blocks.empty() && w.blocks[0].ff.dt == DT_F16 && w.blocks[0].ff.k_form == 0;      // (K = 2 x: the MIX_LINEAR_F16X2 form, an fp32-class projection)
  const bool mix_out1 = hl_attn && !w.blocks.empty() && w.blocks[0].out1.dt == DT_F16 && w.blocks[0].out1.k_form == 0;
  const bool mix_out2 = hl_attn && !w.blocks.empty() && w.blocks[0].out2.dt == DT_F16 && w.blocks[0].out2.k_form == 0;
  Act ao2_16;      // operand of an f16 cross-attention out-projection: the split-operand attention rounds its fp32 result once, in its own store (AttnParams::o_dt)
  if (mix_out2) ao2_16 = ex.alloc(M, C, DT_F16);
  const bool mix_attn = hl_attn && ((mix_ & MIX_ATTN_F16) || mix_qkv) && C % 16 == 0;
  SDXL_REQUIRE(!mix_qkv || mix_attn, "mixed mode: an f16 QKV projection feeds the f16 self-attention");
  SDXL_REQUIRE(!mix_out1 || mix_attn, "mixed mode: an f16 out-projection reads the f16 self-attention's output");
  // (the GEGLU weights of a mixed-mode model are PACKED f16 at build: the f16 path is the only one they can take, whatever the token count)
  const bool mix_geglu = hl_attn && !w.blocks.empty() && w.blocks[0].geglu.dt == DT_F16 && w.blocks[0].geglu.k_form != 2;      // (k_form 2: an fp32-class projection on the f16 kernel, MIX_LINEAR_F16X2)
  Act qk16, ao16, ln16; void* vt16 = nullptr;
  if (mix_attn) {
    qk16 = ex.alloc(M, 2 * C, DT_F16); ao16 = ex.alloc(M, C, DT_F16);
    vt16 = ex.act->alloc((size_t)B * C * npad * 2);
    if (npad != HW && !ex.dry) launch_fill_zero(vt16, (size_t)B * C * npad * 2, ex.s);
  }
  // cross-attention of the mixed mode on f16 (MIX_XATTN_F16): the f16 engine's fused launch -- query projection and the 77-key attention in one kernel on
  // the f16 LayerNorm output and the packed f16 context; shapes that launch does not take (tiny nets) widen the f16 query for the split-operand attention
  const bool mix_q2 = hl_attn && !w.blocks.empty() && w.blocks[0].q2.dt == DT_F16 && w.blocks[0].q2.k_form == 0;
  const bool x2_q2 = hl_attn && !w.blocks.empty() && w.blocks[0].q2.k_form == 2;      // fp32-class query projection on the f16 kernels (MIX_LINEAR_F16X2)
  // the f16 engine's fused launch (MIX_XATTN_F16: the set_context of such a model packed the f16 context, kv.xa) -- otherwise an f16 query projection
  // (MIX_Q2_F16) writes an fp32 q for the split-operand attention
  const bool mix_xa = mix_q2 && mix_out2 && (mix_ & MIX_XATTN_F16) && plan_xattn_ && !kv_.empty() && kv_[si][0].xa && igemm_xattn_ok(DT_F16, DT_F16, (int)M, C, C, HW, n_ctx_);
  const bool mix_q2_widen = mix_q2 && (mix_ & MIX_XATTN_F16) && mix_out2;     // round 5's form of the knob on shapes the fused launch does not take: f16 q, widened
  if (mix_geglu || mix_qkv || mix_q2) ln16 = ex.alloc(M, C, DT_F16);
  // MIX_GEGLU_HILO: the GEGLU projection's weights are (hi | lo 2^8) halves along a doubled K (WeightBuilder::linear_hilo): its A operand is [a | a 2^-8]
  const bool gg_hilo = mix_geglu && !w.blocks.empty() && w.blocks[0].geglu.k_form == 1;      // (the LayerNorm-launch path; the shadow form of such a projection reads sh16g)
  Act ln16x2;
  if (gg_hilo) ln16x2 = ex.alloc(M, 2 * C, DT_F16);
  // MIX_LN_SHADOW: f16 shadow of the stream + the fp32 rows' statistics, left by the weights-in-registers producers (out-projections, FF-out) for the f16
  // projection behind the next LayerNorm; `have_sh` says whether the last producer wrote them (else: LayerNorm launch + the plain form of the projection)
  const bool any_sh = hl_attn && !w.blocks.empty() && (w.blocks[0].qkv_sh.cs || w.blocks[0].q2_sh.cs || w.blocks[0].geglu_sh.cs);
  Act sh16; float* shst = nullptr; bool have_sh = false;
  if (any_sh) { sh16 = ex.alloc(M, C, DT_F16); shst = (float*)ex.act->alloc(M * (size_t)((C + 63) / 64) * 2 * sizeof(float)); }
  Act sh16g;       // (hi | lo 2^8) shadow for a GEGLU projection packed (w | w 2^-8) along a doubled K (MIX_GEGLU_AHILO with MIX_LN_SHADOW)
  const bool sh_g2 = any_sh && w.blocks[0].geglu_sh.cs && w.blocks[0].geglu_sh.k_form == 1;
  if (sh_g2) sh16g = ex.alloc(M, 2 * C, DT_F16);
  // MIX_LINEAR_F16X2 with MIX_LN_SHADOW: the shadow is the HL16 image of x o gamma (what the LayerNorm launch would hand the k_form-2 projection, minus the
  // normalisation the consumer's epilogue applies from the row statistics)
  Act shhl;
  if (any_sh && (w.blocks[0].qkv_sh.k_form == 2 || w.blocks[0].q2_sh.k_form == 2 || w.blocks[0].geglu_sh.k_form == 2)) shhl = ex.alloc(M, C, DT_HL);
  Act q32;     // MIX_Q2_F16: fp32 q of the cross-attention (the f16 projection's fp32 accumulators, never rounded to f16)
  if (mix_q2 && !mix_q2_widen) q32 = ex.alloc(M, C, DT_F32);
  // MIX_XATTN_SPLIT: the split-precision attention runs inside that projection's epilogue on q's accumulators (hi / lo context images of set_context), and its
  // f16 rows are the out-projection's operand: q never reaches memory, no attention launch
  const bool mix_xs = mix_q2 && !mix_q2_widen && mix_out2 && plan_xattn_ && !kv_.empty() && kv_[si][0].xa_lo && igemm_xattn_ok(DT_F16, DT_F16, (int)M, C, C, HW, n_ctx_);
  // ... and in the MIX_LINEAR_F16X2 form of the query projection (K = 2 C, HL16 rows out: the out-projection's fp32-class operand)
  const bool x2_xs = x2_q2 && plan_xattn_ && !kv_.empty() && kv_[si][0].xa_lo && ao.dt == DT_HL && igemm_xattn_ok(DT_F16, DT_HL, (int)M, C, 2 * C, HW, n_ctx_);
  // MIX_LINEAR_F16X2: an un-scaled HL16 operand of C logical channels handed to an f16 GEMM whose weight is packed twice in the HL16 interleave (K = 2 C)
  auto x2op = [&](const Lin& l, const Act& a, int Cl) { return (l.dt == DT_F16 && l.k_form == 2 && l.K == 2 * Cl && a.dt == DT_HL && !a.a_scale) ? Act(a.p, 2 * a.ld, DT_F16) : a; };
  auto want_shadow = [&](Epi& e, const Lin& consumer_sh, const NormW& n) {     // ask producer `e` for the shadow the consumer behind LayerNorm n reads
    have_sh = false;
    if (!consumer_sh.cs || C % 64 != 0) return;
    e.shadow = sh16.p; e.shadow_ld = sh16.ld; e.shadow_gamma = n.gamma; e.stat_out = shst; e.shadow_done = &have_sh;
    if (consumer_sh.k_form == 1) { e.shadow = sh16g.p; e.shadow_ld = sh16g.ld; e.shadow_lo_scale = kHiLoScale; }      // (hi | lo) halves: the consumer's K is doubled
    if (consumer_sh.k_form == 2) { e.shadow = shhl.p; e.shadow_ld = shhl.ld; e.shadow_lo_scale = -1.f; }      // HL16 rows
  };
  auto sh_of = [&](const Lin& consumer_sh) { return consumer_sh.k_form == 2 ? Act(shhl.p, 2 * shhl.ld, DT_F16) : consumer_sh.k_form == 1 ? sh16g : sh16; };
  // producers of the shadow: the projections that run the f16 kernels (f16 class or MIX_LINEAR_F16X2), where the selection picks the weights-in-registers kernel
  auto sh_prod = [&](const Lin& l) { return hl_attn && l.dt == DT_F16; };
  // the f16 GEGLU kernels store an HL16 output through the LDS-staged epilogue of the wide / pipelined tiles -- the kernels every SDXL shape runs on
  // (M = 2048 ... 32768).  Small token counts (tiny test nets: M < 256) run on other tiles; they take the form the F16_F32RES engine
  // runs at every size -- f16 output -- and widen it.
  const bool gg_direct = M >= 256;
  Act gg16;
  if ((mix_geglu && !gg_direct) || mix_ff) gg16 = ex.alloc(M, 4 * C, DT_F16);      // (an f16 FF-out reads the GEGLU output as f16, whichever kernel wrote it)
  void* kh = hl_attn && !hl_direct ? ex.act->alloc(M * (size_t)C * 4) : nullptr;
  void* vth = hl_attn && !hl_direct ? ex.act->alloc((size_t)B * C * npad * 4) : nullptr;
  if (fuse_ln_) {
    // LayerNorms folded into the consuming GEMMs: every producer of the residual stream t also accumulates the row
    // (sum, sum^2) its consumer needs, so no LayerNorm kernel runs and t is read by the projections directly
    SDXL_REQUIRE(w.blocks.empty() || w.blocks[0].qkv.cs, "transformer weights were not LayerNorm-folded");
    for (size_t j = 0; j < w.blocks.size(); ++j) {
      const TBlockW& b = w.blocks[j];
      Epi eq; eq.n_split = 2 * C; eq.Ct = vt; eq.ct_rows = C; eq.ct_ld = npad; eq.rpb = HW; eq.ln_stat = stbuf[stp]; eq.cls = DM_QKV;
      run_linear(ex, b.qkv, t, (int)M, qk, eq);
      attention(ex, qk, qk.cols(C), vt, npad, ao, B, w.heads, HW, HW);
      stp ^= 1;
      Epi e1; e1.R = t; e1.stat_out = stbuf[stp]; e1.rpb = HW; e1.cls = DM_OUT;
      run_linear(ex, b.out1, ao, (int)M, t, e1);
      Epi e2q; e2q.ln_stat = stbuf[stp]; e2q.rpb = HW; e2q.cls = DM_XATTN;
      if (xattn) {   // the projection's waves run the 77-key attention on their own q tiles: no q round trip, no launch
        e2q.xa_k = kv_xa(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
        run_linear(ex, b.q2, t, (int)M, ao, e2q);
      } else {
        run_linear(ex, b.q2, t, (int)M, q, e2q);
        attention(ex, q, Act(kv_k(si, j), C, ex.cdt), kv_vt(si, j), vt_ld_ctx_, ao, B, w.heads, HW, n_ctx_);
      }
      stp ^= 1;
      Epi e2; e2.R = t; e2.stat_out = stbuf[stp]; e2.rpb = HW; e2.cls = DM_OUT;
      run_linear(ex, b.out2, ao, (int)M, t, e2);
      Epi eg; eg.act = 1; eg.ln_stat = stbuf[stp]; eg.rpb = HW; eg.cls = DM_GEGLU;
      run_linear(ex, b.geglu, t, (int)M, gg, eg);
      stp ^= 1;
      Epi ef; ef.R = t; ef.stat_out = j + 1 < w.blocks.size() ? stbuf[stp] : nullptr; ef.rpb = HW; ef.cls = DM_FF;
      run_linear(ex, b.ff, gg, (int)M, t, ef);
    }
  } else
  for (size_t j = 0; j < w.blocks.size(); ++j) {
    const TBlockW& b = w.blocks[j];
    Epi eq; eq.n_split = 2 * C; eq.Ct = mix_attn ? vt16 : vt; eq.ct_rows = C; eq.ct_ld = npad; eq.rpb = HW; eq.cls = DM_QKV;
    if (have_sh && b.qkv_sh.cs) {      // the previous block's FF-out left f16(t o gamma1) and the row statistics: no LayerNorm launch
      eq.ln_stat = shst;
      run_linear(ex, b.qkv_sh, sh_of(b.qkv_sh), (int)M, mix_attn ? qk16 : qk, eq);
    } else {
    run_layernorm(ex, b.n1, t, (int)M, mix_qkv ? ln16 : ln);
    demote_lo(ex, DM_QKV, ln, M, C);
    run_linear(ex, b.qkv, mix_qkv ? ln16 : x2op(b.qkv, ln, C), (int)M, mix_attn ? qk16 : qk, eq);
    }
    have_sh = false;
    if (mix_attn) {
      attention(ex, qk16, qk16.cols(C), vt16, npad, ao16, B, w.heads, HW, HW);
      if (!ex.dry && !mix_out1) launch_f16_to_hl(ao16.p, ao16.ld, ao.p, ao.ld, M, C, ex.s);
    }
    else if (hl_attn && hl_direct) {     // q | k and V^T are HL16; the attention writes the out-projection's operand
      demote_lo(ex, DM_ATTN, qk, M, 2 * C);
      demote_lo(ex, DM_ATTN, Act(vt, npad, DT_HL), (size_t)B * C, npad);
      attention_hl(ex, qk, qk.cols(C).p, qk.ld, vt, npad, ao, B, w.heads, HW, HW, DM_ATTN);
    }
    else if (hl_attn) {
      if (!ex.dry) {
        launch_f32_to_hl(qk.cols(C).p, qk.ld, kh, C, M, C, ex.s);
        launch_f32_to_hl(vt, npad, vth, npad, (size_t)B * C, npad, ex.s);
      }
      attention_hl(ex, qk, kh, C, vth, npad, ao, B, w.heads, HW, HW);
    }
    else attention(ex, qk, qk.cols(C), vt, npad, ao, B, w.heads, HW, HW);
    Epi er; er.R = t; er.rpb = HW; er.cls = DM_OUT;
    demote_lo(ex, DM_OUT, ao, M, C);
    { Epi e1 = er; if (sh_prod(b.out1)) want_shadow(e1, b.q2_sh, b.n2); run_linear(ex, b.out1, mix_out1 ? ao16 : x2op(b.out1, ao, C), (int)M, t, e1); }
    if (have_sh && b.q2_sh.cs) {       // f16 query projection on the shadow the out-projection left; fp32 q for the split-operand attention
      Epi e2q; e2q.cls = DM_XATTN; e2q.rpb = HW; e2q.ln_stat = shst;
      if (mix_xs || x2_xs) {
        e2q.xa_k = kv_xa(si, j); e2q.xa_k_lo = kv_xa_lo(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
        run_linear(ex, b.q2_sh, sh_of(b.q2_sh), (int)M, mix_xs ? ao2_16 : ao, e2q);
      } else {
      const Act& qo = x2_q2 ? q : q32;
      run_linear(ex, b.q2_sh, sh_of(b.q2_sh), (int)M, qo, e2q);
      attention_hl(ex, qo, kv_k(si, j), C, kv_vt(si, j), vt_ld_ctx_, mix_out2 ? ao2_16 : ao, B, w.heads, HW, n_ctx_, DM_XATTN);
      }
    } else {
    run_layernorm(ex, b.n2, t, (int)M, mix_q2 ? ln16 : ln);
    demote_lo(ex, DM_XATTN, ln, M, C);
    if (mix_q2 && !mix_xa && !mix_q2_widen) {
      if (mix_xs) {
        Epi e2q; e2q.cls = DM_XATTN; e2q.rpb = HW;
        e2q.xa_k = kv_xa(si, j); e2q.xa_k_lo = kv_xa_lo(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
        run_linear(ex, b.q2, ln16, (int)M, ao2_16, e2q);
      } else {
      { Epi e2q; e2q.cls = DM_XATTN; e2q.rpb = HW; run_linear(ex, b.q2, ln16, (int)M, q32, e2q); }
      attention_hl(ex, q32, kv_k(si, j), C, kv_vt(si, j), vt_ld_ctx_, mix_out2 ? ao2_16 : ao, B, w.heads, HW, n_ctx_, DM_XATTN);
      }
    } else
    if (mix_xa) {
      Epi e2q; e2q.rpb = HW; e2q.cls = DM_XATTN;
      e2q.xa_k = kv_xa(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
      run_linear(ex, b.q2, ln16, (int)M, ao2_16, e2q);
    } else if (mix_q2) {
      { Epi e2q; e2q.cls = DM_XATTN; run_linear(ex, b.q2, ln16, (int)M, ao2_16, e2q); }
      if (!ex.dry) {
        if (q.dt == DT_HL) launch_f16_to_hl(ao2_16.p, ao2_16.ld, q.p, q.ld, M, C, ex.s);
        else launch_copy_rows(ao2_16.p, DT_F16, ao2_16.ld, q.p, q.dt, q.ld, (int)M, C, ex.s);
      }
      attention_hl(ex, q, kv_k(si, j), C, kv_vt(si, j), vt_ld_ctx_, ao2_16, B, w.heads, HW, n_ctx_, DM_XATTN);
    } else if (xattn) {
      Epi e2q; e2q.rpb = HW; e2q.cls = DM_XATTN;
      e2q.xa_k = kv_xa(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
      run_linear(ex, b.q2, ln, (int)M, ao, e2q);
    } else if (x2_xs) {     // fp32-class projection on the f16 kernel (HL16 operand read as f16) with the split-precision attention in its epilogue: HL16 rows for the out-projection
      Epi e2q; e2q.rpb = HW; e2q.cls = DM_XATTN;
      e2q.xa_k = kv_xa(si, j); e2q.xa_k_lo = kv_xa_lo(si, j); e2q.xa_nctx = n_ctx_; e2q.xa_scale = 0.125f;
      run_linear(ex, b.q2, x2op(b.q2, ln, C), (int)M, ao, e2q);
    } else {
      { Epi e2q; e2q.cls = DM_XATTN; run_linear(ex, b.q2, x2op(b.q2, ln, C), (int)M, q, e2q); }
      demote_lo(ex, DM_XATTN, q, M, C);
      if (hl_attn) attention_hl(ex, q, kv_k(si, j), C, kv_vt(si, j), vt_ld_ctx_, mix_out2 ? ao2_16 : ao, B, w.heads, HW, n_ctx_, DM_XATTN);   // caches are HL16 (set_context)
      else attention(ex, q, Act(kv_k(si, j), C, adt), kv_vt(si, j), vt_ld_ctx_, ao, B, w.heads, HW, n_ctx_);
    }
    }
    have_sh = false;
    demote_lo(ex, DM_OUT, ao, M, C);
    { Epi e2 = er; if (sh_prod(b.out2)) want_shadow(e2, b.geglu_sh, b.n3); run_linear(ex, b.out2, mix_out2 ? ao2_16 : x2op(b.out2, ao, C), (int)M, t, e2); }
    Epi eg; eg.act = 1; eg.cls = DM_GEGLU;
    const bool gg_sh = have_sh && b.geglu_sh.cs;      // GEGLU projection on the shadow the cross-attention's out-projection left: no LayerNorm launch
    if (gg_sh) eg.ln_stat = shst;
    else if (gg_hilo) run_layernorm(ex, b.n3, t, (int)M, ln16x2, (mix_ & MIX_GEGLU_AHILO) ? -kHiLoScale : 1.0f / kHiLoScale);
    else {
    run_layernorm(ex, b.n3, t, (int)M, mix_geglu ? ln16 : ln);
    demote_lo(ex, DM_GEGLU, ln, M, C);
    }
    have_sh = false;
    if (gg_sh) {
      eg.rpb = HW;
      const Act shg = sh_of(b.geglu_sh);
      if (mix_ff) run_linear(ex, b.geglu_sh, shg, (int)M, gg16, eg);
      else if (!gg_direct) { run_linear(ex, b.geglu_sh, shg, (int)M, gg16, eg); if (!ex.dry) launch_f16_to_hl(gg16.p, gg16.ld, gg.p, gg.ld, M, 4 * C, ex.s); }
      else run_linear(ex, b.geglu_sh, shg, (int)M, gg, eg);
    } else
    if (mix_ff) {
      run_linear(ex, b.geglu, gg_hilo ? ln16x2 : mix_geglu ? ln16 : ln, (int)M, gg16, eg);         // f16 output for the f16 FF-out (f16 or split-operand GEGLU compute)
    } else if (mix_geglu && !gg_direct) {
      run_linear(ex, b.geglu, gg_hilo ? ln16x2 : ln16, (int)M, gg16, eg);
      if (!ex.dry) launch_f16_to_hl(gg16.p, gg16.ld, gg.p, gg.ld, M, 4 * C, ex.s);
    } else
    run_linear(ex, b.geglu, gg_hilo ? ln16x2 : mix_geglu ? ln16 : x2op(b.geglu, ln, C), (int)M, gg, eg);
    demote_lo(ex, DM_FF, gg, M, 4 * C);
    er.cls = DM_FF;
    { Epi ef = er; if (sh_prod(b.ff) && j + 1 < w.blocks.size()) want_shadow(ef, w.blocks[j + 1].qkv_sh, w.blocks[j + 1].n1); run_linear(ex, b.ff, mix_ff ? gg16 : x2op(b.ff, gg, 4 * C), (int)M, t, ef); }
  }
  Epi eo; eo.R = x; eo.rpb = HW; eo.cls = DM_CONV_PROJ;
  run_linear(ex, w.proj_out, hl_op(ex, w.proj_out, t, M, C, DM_CONV_PROJ, B), (int)M, x, eo);
  ex.act->reset(mk);
}

// ------------------------------------------------------------------------------------------ forward
void UNet::run(Exec& ex, const float* t_dev, int t_stride, int b0, int nb) {
  // batch entries [b0, b0 + nb) of the plan: every per-entry buffer is addressed through these views
  const int B = nb, H = pH_, W = pW_;
  const int mc = cfg_.model_channels, emb = 4 * mc;
  float* temb = temb_ + (size_t)b0 * mc;
  float* g1 = g1_ + (size_t)b0 * emb;
  float* embv = emb_ + (size_t)b0 * emb;
  float* ebias = ebias_ + (size_t)b0 * emb_total_;
  const float* label_emb = label_emb_ ? label_emb_ + (size_t)b0 * emb : nullptr;
  void* in = (char*)in_ + (size_t)b0 * H * W * cfg_.in_channels * dt_size(input_dt());
  float* eps = eps_ + (size_t)b0 * H * W * cfg_.out_channels;
  ex.ebias = ebias; ex.b0 = b0;
  ex.gn_partial = gn_partial_ + (size_t)b0 * groupnorm_workspace_floats(1, 32);
  // --- embeddings (unet/mod.rs:458-468)
  if (!ex.dry) launch_timestep_embedding(t_dev + (size_t)b0 * t_stride, t_stride, temb, B, mc, ex.s);
  gemv(ex, lin1_t_, temb, mc, g1, emb, B, false, true);
  gemv(ex, lin2_t_, g1, emb, embv, emb, B, false, false, label_emb);
  gemv(ex, embcat_, embv, emb, ebias, emb_total_, B, true, false);

  // --- geometry of the skip / concat buffers
  const int n_in = (int)inp_.size(), n_out = (int)out_.size();
  SDXL_REQUIRE(n_in == n_out, "input/output block count mismatch");
  std::vector<int> hs_h(n_in), hs_w(n_in), hs_c(n_in);
  {
    int h = H, w = W;
    for (int i = 0; i < n_in; ++i) {
      const BlockDesc& d = inp_[i].d;
      if (d.kind == BK_DOWN) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
      hs_h[i] = h; hs_w[i] = w; hs_c[i] = d.c_out;
    }
  }
  std::vector<Act> cat(n_out);
  std::vector<int> cx(n_out);
  for (int j = 0; j < n_out; ++j) {
    const int i = n_in - 1 - j;
    cx[j] = out_[j].d.c_in - hs_c[i];
    SDXL_REQUIRE(cx[j] > 0, "bad concat geometry");
    cat[j] = ex.alloc((size_t)B * hs_h[i] * hs_w[i], out_[j].d.c_in, ex.sdt);
  }
  // --- input blocks (:474-477)
  Act cur(in, cfg_.in_channels, input_dt());
  int h = H, w = W, cur_c = cfg_.in_channels;
  int si = 0;
  for (int i = 0; i < n_in; ++i) {
    const BlockW& b = inp_[i];
    const int j = n_in - 1 - i;
    const Act dest = cat[j].cols(cx[j]);
    switch (b.d.kind) {
      case BK_CONV: run_conv(ex, b.conv, hl_op(ex, b.conv, cur, (size_t)B * h * w, cur_c, DM_CONV_IO, B), cur_c, ConvGeom{B, h, w, h, w, 3, 1, 1, 0}, dest, tag_epi(DM_CONV_IO)); break;
      case BK_DOWN: {
        const int h2 = (h - 1) / 2 + 1, w2 = (w - 1) / 2 + 1;
        run_conv(ex, b.conv, hl_op(ex, b.conv, cur, (size_t)B * h * w, cur_c, DM_CONV_UPDOWN, B), cur_c, ConvGeom{B, h, w, h2, w2, 3, 2, 1, 0}, dest, tag_epi(DM_CONV_UPDOWN));
        h = h2; w = w2;
        break;
      }
      case BK_RES: res_block(ex, b.res, cur, B, h, w, dest); break;
      case BK_REST: {
        Act d = dest;      // statistics of the ResBlock output for the transformer's GroupNorm (scratch outlives both calls)
        const size_t mkp = ex.act->mark();
        float* part = (float*)ex.act->alloc((size_t)B * h * w / 256 * b.d.c_out * 2 * sizeof(float) + 256);
        d.gn_part = res_block(ex, b.res, cur, B, h, w, dest, part); d.gn_rt = h * w / 256;
        spatial_transformer(ex, b.st, si++, d, B, h, w);
        ex.act->reset(mkp);
        break;
      }
      default: throw Error("unexpected input block kind");
    }
    cur = dest; cur_c = b.d.c_out;
  }
  // --- middle block (:480, :713-719)
  {
    const size_t mk = ex.act->mark();
    Act m1 = ex.alloc((size_t)B * h * w, mid_res1_.d.c_out, ex.sdt);
    float* part = (float*)ex.act->alloc((size_t)B * h * w / 256 * mid_res1_.d.c_out * 2 * sizeof(float) + 256);
    m1.gn_part = res_block(ex, mid_res1_.res, cur, B, h, w, m1, part); m1.gn_rt = h * w / 256;
    spatial_transformer(ex, mid_res1_.st, si++, m1, B, h, w);
    m1.gn_part = nullptr;                 // the transformer rewrote m1 in place
    res_block(ex, mid_res2_.res, m1, B, h, w, cat[0].cols(0));
    ex.act->reset(mk);
  }
  // --- output blocks (:483-486)
  Act last = ex.alloc((size_t)B * H * W, mc, ex.sdt);
  for (int j = 0; j < n_out; ++j) {
    const BlockW& b = out_[j];
    const bool up = b.d.kind == BK_RESTU || b.d.kind == BK_RESU;
    const Act next = j + 1 < n_out ? cat[j + 1].cols(0) : last;
    const size_t mk = ex.act->mark();
    Act dest = up ? ex.alloc((size_t)B * h * w, b.d.c_out, ex.sdt) : next;
    if (b.d.kind == BK_REST || b.d.kind == BK_RESTU) {
      float* part = (float*)ex.act->alloc((size_t)B * h * w / 256 * b.d.c_out * 2 * sizeof(float) + 256);
      Act d = dest;
      d.gn_part = res_block(ex, b.res, cat[j], B, h, w, dest, part); d.gn_rt = h * w / 256;
      spatial_transformer(ex, b.st, si++, d, B, h, w);
    } else {
      res_block(ex, b.res, cat[j], B, h, w, dest);
    }
    if (up) {   // Upsample::forward :742-752 -- nearest 2x fused into the conv gather
      run_conv(ex, b.conv, hl_op(ex, b.conv, dest, (size_t)B * h * w, b.d.c_out, DM_CONV_UPDOWN, B), b.d.c_out, ConvGeom{B, h, w, 2 * h, 2 * w, 3, 1, 1, 1}, next, tag_epi(DM_CONV_UPDOWN));
      h *= 2; w *= 2;
    }
    ex.act->reset(mk);
  }
  SDXL_REQUIRE(h == H && w == W, "UNet input height/width must be divisible by 2^(levels-1)");
  // --- out (:488-490)
  {
    const size_t mk = ex.act->mark();
    Act gn = ex.alloc((size_t)B * H * W, mc, ex.cdt);
    run_groupnorm(ex, norm_out_, last, B, H * W, gn, true);
    demote_lo(ex, DM_CONV_IO, gn, (size_t)B * H * W, mc);
    run_conv(ex, conv_out_, gn, mc, ConvGeom{B, H, W, H, W, 3, 1, 1, 0}, Act(eps, cfg_.out_channels, DT_F32), tag_epi(DM_CONV_IO));
    ex.act->reset(mk);
  }
}

void UNet::ensure_plan(int B, int H, int W) {
  const bool split = split_cfg_ && B == 2;
  if (B == pB_ && H == pH_ && W == pW_ && split == plan_split_ && fuse_xattn_ == plan_xattn_ && gn_from_producer_ == plan_gn_) return;
  SDXL_REQUIRE(B >= 1 && B <= 8, "batch must be in 1..8");
  const int div = 1 << (cfg_.channel_mults.size() - 1);
  SDXL_REQUIRE(H >= div && W >= div && H % div == 0 && W % div == 0,
               "UNet input height/width must be divisible by 2^(levels-1)");
  if (graph_) { (void)hipGraphExecDestroy(graph_); graph_ = nullptr; }
  plan_runs_ = 0;
  warm_ = WarmSeq();
  pB_ = B; pH_ = H; pW_ = W;
  const int mc = cfg_.model_channels, emb = 4 * mc;
  auto persist = [&]() {
    in_ = act_.alloc((size_t)B * H * W * cfg_.in_channels * dt_size(input_dt()));
    eps_ = (float*)act_.alloc((size_t)B * H * W * cfg_.out_channels * sizeof(float));
    temb_ = (float*)act_.alloc((size_t)B * mc * sizeof(float));
    g1_ = (float*)act_.alloc((size_t)B * emb * sizeof(float));
    emb_ = (float*)act_.alloc((size_t)B * emb * sizeof(float));
    ebias_ = (float*)act_.alloc((size_t)B * emb_total_ * sizeof(float));
    gn_partial_ = (float*)act_.alloc(groupnorm_workspace_floats(B, 32) * sizeof(float));
    tconv_ = (float*)act_.alloc(8 * sizeof(float));
    if (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16)) {   // cross-workgroup key split of the (f16) self-attention: workspace + tickets per chain, sized for the largest level
      size_t wsb = 0, cnt = 0;
      { int h = H, w = W;
        for (size_t lv = 0; lv < cfg_.channel_mults.size(); ++lv) {
          const int heads = cfg_.model_channels * cfg_.channel_mults[lv] / cfg_.n_head_channels;
          if (lv < cfg_.transformer_depths.size() && cfg_.transformer_depths[lv] > 0) {      // (levels without a SpatialTransformer run no attention)
            wsb = std::max(wsb, attention_xsplit_ws_bytes(B, heads, h * w)); cnt = std::max(cnt, attention_xsplit_counters(B, heads, h * w));
          }
          h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1;
        } }
      attn_xcnt_bytes_ = cnt * sizeof(unsigned);
      attn_xws_[1] = nullptr; attn_xcnt_[1] = nullptr;
      for (int c = 0; c < (split ? 2 : 1); ++c) {
        attn_xws_[c] = (float*)act_.alloc(wsb + 256);
        attn_xcnt_[c] = (unsigned*)act_.alloc(attn_xcnt_bytes_ + 256);
      }
    }
    if (cdt_ == DT_F16 || cdt_ == DT_HL) {   // split-K slabs + counters: chain 0 (and the second split-CFG chain)
      skws_bytes_ = igemm_splitk_ws_bytes(B, 1024, 1536);
      skws_[1] = nullptr; skcnt_[1] = nullptr;
      for (int c = 0; c < (split ? 2 : 1); ++c) {   // the second set only exists for the second split-CFG chain
        skws_[c] = (float*)act_.alloc(skws_bytes_);
        skcnt_[c] = (unsigned*)act_.alloc(kSplitkCounters * sizeof(unsigned));
      }
    }
  };
  // dry run for the peak, then the real arena
  act_.dry = true; act_.off = 0; act_.peak = 0;
  persist();
  Exec ex; ex.dry = true; ex.cdt = cdt_; ex.sdt = sdt_; ex.act = &act_;
  const size_t m = act_.mark();
  const bool had_kv = !kv_.empty();
  if (!had_kv) {   // plan built before set_context: fake cache entries for the dry run
    kv_.assign(st_list_.size(), std::vector<KV>());
    for (size_t i = 0; i < st_list_.size(); ++i) kv_[i].assign(st_list_[i]->blocks.size(), KV());
  }
  run(ex, nullptr, 0, 0, B);
  if (split) {   // scratch peak of one batch-1 chain -> the second chain's own arena
    act2_.dry = true; act2_.off = 0; act2_.peak = 0;
    Exec e2; e2.dry = true; e2.cdt = cdt_; e2.sdt = sdt_; e2.act = &act2_;
    run(e2, nullptr, 0, 1, 1);
    const size_t peak2 = act2_.peak;
    act2_.dry = false;
    act2_.reserve(peak2 + 4096);
    act2_.off = 0; act2_.peak = 0;
    if (!s2_) {
      SDXL_HIP(hipStreamCreateWithFlags(&s2_, hipStreamNonBlocking));
      SDXL_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
      SDXL_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
  }
  plan_split_ = split;
  plan_xattn_ = fuse_xattn_;
  plan_gn_ = gn_from_producer_;
  if (!had_kv) kv_.clear();
  act_.reset(m);
  const size_t peak = act_.peak;
  act_.dry = false;
  act_.reserve(peak + 4096);
  act_.off = 0; act_.peak = 0;
  persist();
  for (int c = 0; c < 2; ++c) if (skcnt_[c]) SDXL_HIP(hipMemset(skcnt_[c], 0, kSplitkCounters * sizeof(unsigned)));   // armed once
  for (int c = 0; c < (split ? 2 : 1); ++c) if (attn_xcnt_[c] && (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16))) SDXL_HIP(hipMemset(attn_xcnt_[c], 0, attn_xcnt_bytes_));
}

void* UNet::unet_in(int B, int H, int W) { ensure_plan(B, H, W); return in_; }

void UNet::forward(int B, int H, int W, const float* t_dev, int t_stride, hipStream_t s) {
  ensure_plan(B, H, W);
  SDXL_REQUIRE(ctx_B_ == B && !kv_.empty(), "set_context must be called with the same batch before forward");
  Exec ex; ex.s = s; ex.cdt = cdt_; ex.sdt = sdt_; ex.act = &act_; ex.gn_partial = gn_partial_; ex.demote = demote_mask_;
  ex.splitk_ws = skws_[0]; ex.splitk_ws_bytes = skws_bytes_; ex.splitk_cnt = skcnt_[0];
  if (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16)) { ex.attn_xws = attn_xws_[0]; ex.attn_xcnt = attn_xcnt_[0]; }
  // weight warming (f16 engine, batched chain): the plan's first forward records the GEMM sequence, every later one replays it
  const bool warming = cdt_ == DT_F16 && !plan_split_ && igemm_warm_enabled();
  if (warming) { ex.warm = &warm_; if (!warm_.ready) { warm_.seq.clear(); warm_.recording = true; } }
  const size_t m = act_.mark();
  // one batched chain, or (split-CFG) entry 0 on s and entry 1 on the side stream between a fork and a join event
  auto go = [&]() {
    warm_.pos = 0;
    if (!plan_split_) { run(ex, t_dev, t_stride, 0, B); if (warm_.recording) warm_.finish(); return; }
    Exec e2; e2.s = s2_; e2.cdt = cdt_; e2.sdt = sdt_; e2.act = &act2_; e2.demote = demote_mask_;
    e2.splitk_ws = skws_[1]; e2.splitk_ws_bytes = skws_bytes_; e2.splitk_cnt = skcnt_[1];
    if (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16)) { e2.attn_xws = attn_xws_[1]; e2.attn_xcnt = attn_xcnt_[1]; }
    act2_.off = 0;
    ex.fork_ev = ev_fork_; ex.fork_after = split_offset_; ex.launches = 0;
    if (ex.fork_after <= 0) SDXL_HIP(hipEventRecord(ev_fork_, s));
    run(ex, t_dev, t_stride, 0, 1);
    if (ex.fork_after > 0 && ex.launches < ex.fork_after) SDXL_HIP(hipEventRecord(ev_fork_, s));   // offset beyond the chain
    ex.fork_ev = nullptr;
    SDXL_HIP(hipStreamWaitEvent(s2_, ev_fork_, 0));
    run(e2, t_dev, t_stride, 1, 1);
    SDXL_HIP(hipEventRecord(ev_join_, s2_));
    SDXL_HIP(hipStreamWaitEvent(s, ev_join_, 0));
  };
  if (use_graph_ && graph_ && (graph_t_ != t_dev || graph_ts_ != t_stride || graph_off_ != split_offset_ || graph_warm_ != warming || graph_demote_ != demote_mask_)) {
    (void)hipGraphExecDestroy(graph_); graph_ = nullptr;
  }
  if (use_graph_ && !graph_ && plan_runs_ >= 1) {
    // capture the whole forward (~1.3k launches) once; replay costs one launch per forward
    hipGraph_t g = nullptr;
    // the cross-workgroup tickets (attention key halves, split-K) re-arm themselves at the end of every launch; a forward that was torn down
    // half way (device error, cancelled capture) would leave them armed wrongly for good -- re-zero them whenever a graph is (re)captured
    for (int c = 0; c < 2; ++c) {
      if (attn_xcnt_[c] && attn_xcnt_bytes_) SDXL_HIP(hipMemsetAsync(attn_xcnt_[c], 0, attn_xcnt_bytes_, s));
      if (skcnt_[c]) SDXL_HIP(hipMemsetAsync(skcnt_[c], 0, kSplitkCounters * sizeof(unsigned), s));
    }
    SDXL_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    try { go(); } catch (...) { (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); act_.reset(m); throw; }
    SDXL_HIP(hipStreamEndCapture(s, &g));
    SDXL_HIP(hipGraphInstantiate(&graph_, g, nullptr, nullptr, 0));
    SDXL_HIP(hipGraphDestroy(g));
    graph_t_ = t_dev; graph_ts_ = t_stride; graph_off_ = split_offset_; graph_warm_ = warming; graph_demote_ = demote_mask_;
    act_.reset(m);
  }
  if (use_graph_ && graph_) {
    SDXL_HIP(hipGraphLaunch(graph_, s));
  } else {
    go();
    act_.reset(m);
  }
  ++plan_runs_;
}

void UNet::profile(int B, int H, int W, float ms[Profiler::NCLS], int launches[Profiler::NCLS],
                   double flops[Profiler::NCLS], hipStream_t s) {
  ensure_plan(B, H, W);
  SDXL_REQUIRE(ctx_B_ == B && !kv_.empty(), "set_context must be called with the same batch before profile");
  const float t500[8] = {500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f};
  SDXL_HIP(hipMemcpyAsync(tconv_, t500, sizeof(t500), hipMemcpyHostToDevice, s));
  SDXL_HIP(hipStreamSynchronize(s));
  Profiler prof;
  Exec ex; ex.s = s; ex.cdt = cdt_; ex.sdt = sdt_; ex.act = &act_; ex.gn_partial = gn_partial_; ex.prof = &prof; ex.demote = demote_mask_;
  ex.splitk_ws = skws_[0]; ex.splitk_ws_bytes = skws_bytes_; ex.splitk_cnt = skcnt_[0];
  if (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16)) { ex.attn_xws = attn_xws_[0]; ex.attn_xcnt = attn_xcnt_[0]; }
  const size_t m = act_.mark();
  run(ex, tconv_, 1, 0, B);   // always the batched chain: per-launch events need one stream
  act_.reset(m);
  prof.collect(ms, launches, flops);
}

// the same eager chain WITHOUT the per-launch events, bracketed by one event pair: what the launches of profile() take when nobody measures them one
// by one -- (sum of profile()'s class times - this) / launches is the event overhead a bracketed launch carries (bench.py's calibration)
float UNet::eager_ms(int B, int H, int W, hipStream_t s) {
  ensure_plan(B, H, W);
  SDXL_REQUIRE(ctx_B_ == B && !kv_.empty(), "set_context must be called with the same batch before eager_ms");
  const float t500[8] = {500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f, 500.f};
  SDXL_HIP(hipMemcpyAsync(tconv_, t500, sizeof(t500), hipMemcpyHostToDevice, s));
  SDXL_HIP(hipStreamSynchronize(s));
  hipEvent_t a, b;
  SDXL_HIP(hipEventCreate(&a)); SDXL_HIP(hipEventCreate(&b));
  float best = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    Exec ex; ex.s = s; ex.cdt = cdt_; ex.sdt = sdt_; ex.act = &act_; ex.gn_partial = gn_partial_; ex.demote = demote_mask_;
    ex.splitk_ws = skws_[0]; ex.splitk_ws_bytes = skws_bytes_; ex.splitk_cnt = skcnt_[0];
    if (cdt_ == DT_F16 || (mix_ & MIX_ATTN_F16)) { ex.attn_xws = attn_xws_[0]; ex.attn_xcnt = attn_xcnt_[0]; }
    const size_t m = act_.mark();
    SDXL_HIP(hipEventRecord(a, s));
    run(ex, tconv_, 1, 0, B);
    SDXL_HIP(hipEventRecord(b, s));
    act_.reset(m);
    SDXL_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    SDXL_HIP(hipEventElapsedTime(&ms, a, b));
    if (rep == 0 || ms < best) best = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return best;
}

void UNet::forward_nchw(const float* x, const int* timesteps, const float* context, int n_ctx, const float* label, int B,
                        int H, int W, float* out, hipStream_t s) {
  ensure_plan(B, H, W);
  set_context(context, n_ctx, label, B, s);
  launch_nchw_to_nhwc(x, cfg_.in_channels * H * W, in_, input_dt(), B, cfg_.in_channels, H * W, cfg_.in_channels, 1.0f, s);
  launch_i32_to_f32(timesteps, tconv_, B, s);
  forward(B, H, W, tconv_, 1, s);
  launch_nhwc_to_nchw(eps_, DT_F32, cfg_.out_channels, out, B, cfg_.out_channels, H * W, 1.0f, s);
}

}  // namespace sdxl
