// Canonical parameter enumeration of the SDXL UNet / VAE, named after the reference's struct fields, and the
// UNet block plan of UNetConfig::init (reference unet/mod.rs:115-173, 238-248, 250-328).  This is the host-side
// contract of the C ABI's sdxl_*_param_spec(): callers hand weights over in exactly this order, in the reference's
// own layouts (Linear [d_in,d_out], conv [out,in,kh,kw]; python/save.py:20-25,56-72).
#include "engine.h"

#include <cmath>

namespace sdxl {

uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char c : s) { h ^= c; h *= 0x100000001B3ull; }
  return h;
}

namespace {
const double kSqrt12 = std::sqrt(12.0);
const double kResGain = 0.5;   // synthetic-init gain of residual-branch output layers

struct Spec {
  std::vector<ParamSpec> items;
  static float wscale(int fan_in, double gain) { return (float)(kSqrt12 * gain / std::sqrt((double)fan_in)); }
  void add(const std::string& n, std::vector<int> shape, int kind, float scale, float mean) {
    ParamSpec p; p.name = n; p.shape = std::move(shape); p.kind = kind; p.scale = scale; p.mean = mean;
    items.push_back(std::move(p));
  }
  void linear(const std::string& n, int din, int dout, bool bias = true, double gain = 1.0) {
    add(n + ".weight", {din, dout}, PK_LINEAR_W, wscale(din, gain), 0.f);
    if (bias) add(n + ".bias", {dout}, PK_BIAS, (float)(kSqrt12 * 0.02), 0.f);
  }
  void conv(const std::string& n, int cin, int cout, int k, double gain = 1.0) {
    add(n + ".weight", {cout, cin, k, k}, PK_CONV_W, wscale(cin * k * k, gain), 0.f);
    add(n + ".bias", {cout}, PK_BIAS, (float)(kSqrt12 * 0.02), 0.f);
  }
  void norm(const std::string& n, int c) {
    add(n + ".gamma", {c}, PK_GAMMA, (float)(kSqrt12 * 0.02), 1.f);
    add(n + ".beta", {c}, PK_BETA, (float)(kSqrt12 * 0.02), 0.f);
    // per-module eps: the reference's .npy loaders read it per norm (groupnorm/load.rs:19, layernorm/load.rs:17);
    // synthetic value = the Config default 1e-5 exactly (scale 0)
    add(n + ".eps", {1}, PK_EPS, 0.f, 1e-5f);
  }
};

void res_block(Spec& s, const std::string& p, int cin, int cemb, int cout) {   // ResBlockConfig::init unet/mod.rs:1032-1067
  s.norm(p + ".norm_in", cin);
  s.conv(p + ".conv_in", cin, cout, 3);
  s.linear(p + ".lin_embed", cemb, cout);
  s.norm(p + ".norm_out", cout);
  s.conv(p + ".conv_out", cout, cout, 3, kResGain);
  if (cin != cout) s.conv(p + ".skip_connection", cin, cout, 1);
}
void mha(Spec& s, const std::string& p, int n_state, int n_ctx_state) {       // :965-994
  s.linear(p + ".query", n_state, n_state, false);
  s.linear(p + ".key", n_ctx_state, n_state, false);
  s.linear(p + ".value", n_ctx_state, n_state, false);
  s.linear(p + ".out", n_state, n_state, true, kResGain);
}
void transformer(Spec& s, const std::string& p, int c, int ctx, int depth) {  // :790-810, :854-873
  s.norm(p + ".norm", c);
  s.linear(p + ".proj_in", c, c);
  for (int j = 0; j < depth; ++j) {
    const std::string q = p + ".blocks." + std::to_string(j);
    s.norm(q + ".norm1", c);
    mha(s, q + ".attn1", c, c);
    s.norm(q + ".norm2", c);
    mha(s, q + ".attn2", c, ctx);
    s.norm(q + ".norm3", c);
    s.linear(q + ".mlp.geglu.proj", c, 8 * c);
    s.linear(q + ".mlp.lin", 4 * c, c, true, kResGain);
  }
  s.linear(p + ".proj_out", c, c, true, kResGain);
}
void block_params(Spec& s, const std::string& p, const BlockDesc& b, int ctx) {
  switch (b.kind) {
    case BK_CONV: s.conv(p, b.c_in, b.c_out, 3); break;
    case BK_DOWN: s.conv(p, b.c_in, b.c_in, 3); break;
    case BK_RES: res_block(s, p, b.c_in, b.c_emb, b.c_out); break;
    case BK_REST:
    case BK_RESTU:
      res_block(s, p + ".res", b.c_in, b.c_emb, b.c_out);
      transformer(s, p + ".transformer", b.c_out, ctx, b.depth);
      if (b.kind == BK_RESTU) s.conv(p + ".upsample.conv", b.c_out, b.c_out, 3);
      break;
    case BK_RESU:
      res_block(s, p + ".res", b.c_in, b.c_emb, b.c_out);
      s.conv(p + ".upsample.conv", b.c_out, b.c_out, 3);
      break;
  }
}
void vae_resnet(Spec& s, const std::string& p, int cin, int cout) {           // autoencoder/mod.rs:457-490
  s.norm(p + ".norm1", cin);
  s.conv(p + ".conv1", cin, cout, 3);
  s.norm(p + ".norm2", cout);
  s.conv(p + ".conv2", cout, cout, 3, kResGain);
  if (cin != cout) s.conv(p + ".nin_shortcut", cin, cout, 1);
}
void vae_mid(Spec& s, const std::string& p, int c) {                          // :420-433, :523-540
  vae_resnet(s, p + ".block_1", c, c);
  s.norm(p + ".attn.norm", c);
  s.conv(p + ".attn.q", c, c, 1);
  s.conv(p + ".attn.k", c, c, 1);
  s.conv(p + ".attn.v", c, c, 1);
  s.conv(p + ".attn.proj_out", c, c, 1, kResGain);
  vae_resnet(s, p + ".block_2", c, c);
}
}  // namespace

void unet_block_plan(const UNetCfg& cfg, std::vector<BlockDesc>& inp, BlockDesc& mid, std::vector<BlockDesc>& out) {
  const int mc = cfg.model_channels;
  const std::vector<int>& mults = cfg.channel_mults;
  const int n_levels = (int)mults.size();
  SDXL_REQUIRE(n_levels >= 1 && (int)cfg.transformer_depths.size() == n_levels, "channel_mults / transformer_depths size mismatch");
  SDXL_REQUIRE(mc % cfg.n_head_channels == 0,
               "The number of head channels must evenly divide the model channels.");   // unet/mod.rs:73-76
  const int emb = 4 * mc;
  auto nh = [&](int ch) { return ch / cfg.n_head_channels; };
  inp.clear(); out.clear();
  BlockDesc b;
  b = BlockDesc(); b.kind = BK_CONV; b.c_in = cfg.in_channels; b.c_out = mc; inp.push_back(b);
  for (int level = 0; level < n_levels; ++level) {
    const int c_in = mults[level > 0 ? level - 1 : 0] * mc;
    const int c_out = mults[level] * mc;
    const bool tr = (level == 1 || level == 2);
    for (int r = 0; r < 2; ++r) {
      b = BlockDesc();
      b.kind = tr ? BK_REST : BK_RES;
      b.c_in = r == 0 ? c_in : c_out; b.c_emb = emb; b.c_out = c_out;
      if (tr) { b.n_head = nh(c_out); b.depth = cfg.transformer_depths[level]; }
      inp.push_back(b);
    }
    if (level != n_levels - 1) { b = BlockDesc(); b.kind = BK_DOWN; b.c_in = c_out; b.c_out = c_out; inp.push_back(b); }
  }
  const int c_mid = mults.back() * mc;
  mid = BlockDesc(); mid.kind = BK_REST; mid.c_in = c_mid; mid.c_out = c_mid; mid.c_emb = c_mid;   // :240-247
  mid.n_head = nh(c_mid); mid.depth = cfg.transformer_depths.back();
  for (int level = n_levels - 1; level >= 0; --level) {
    const int nxt = level != n_levels - 1 ? level + 1 : level;
    const int c_out = mults[level] * mc;
    const int in1 = mults[nxt] * mc + c_out, in2 = 2 * c_out, in3 = c_out + mults[level > 0 ? level - 1 : 0] * mc;
    const bool tr = (level == 1 || level == 2);
    const int ins[3] = {in1, in2, in3};
    for (int r = 0; r < 3; ++r) {
      b = BlockDesc();
      b.c_in = ins[r]; b.c_emb = emb; b.c_out = c_out;
      if (tr) { b.kind = r == 2 ? BK_RESTU : BK_REST; b.n_head = nh(c_out); b.depth = cfg.transformer_depths[level]; }
      else b.kind = (r == 2 && level != 0) ? BK_RESU : BK_RES;
      out.push_back(b);
    }
  }
}

std::vector<ParamSpec> unet_param_specs(const UNetCfg& cfg) {
  Spec s;
  const int mc = cfg.model_channels, emb = 4 * mc;
  s.linear("lin1_time_embed", mc, emb);
  s.linear("lin2_time_embed", emb, emb);
  s.linear("lin1_label_embed", cfg.adm_in_channels, emb);
  s.linear("lin2_label_embed", emb, emb);
  std::vector<BlockDesc> inp, out; BlockDesc mid;
  unet_block_plan(cfg, inp, mid, out);
  for (size_t i = 0; i < inp.size(); ++i) block_params(s, "input_blocks." + std::to_string(i), inp[i], cfg.context_dim);
  res_block(s, "middle_block.res1", mid.c_in, mid.c_emb, mid.c_out);
  transformer(s, "middle_block.transformer", mid.c_out, cfg.context_dim, mid.depth);
  res_block(s, "middle_block.res2", mid.c_in, mid.c_emb, mid.c_out);
  for (size_t i = 0; i < out.size(); ++i) block_params(s, "output_blocks." + std::to_string(i), out[i], cfg.context_dim);
  s.norm("norm_out", mc);
  s.conv("conv_out", mc, cfg.out_channels, 3);
  return s.items;
}

std::vector<ParamSpec> vae_decoder_param_specs(const VaeCfg& cfg) {   // autoencoder/mod.rs:35,152-191,274-303
  Spec s;
  s.conv("post_quant_conv", 4, 4, 1);
  const int c0 = cfg.dec.front().first;
  s.conv("decoder.conv_in", 4, c0, 3);
  vae_mid(s, "decoder.mid", c0);
  for (size_t i = 0; i < cfg.dec.size(); ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i);
    const int ci = cfg.dec[i].first, co = cfg.dec[i].second;
    vae_resnet(s, p + ".res1", ci, co);
    vae_resnet(s, p + ".res2", co, co);
    vae_resnet(s, p + ".res3", co, co);
    if (i + 1 != cfg.dec.size()) s.conv(p + ".upsampler", co, co, 3);
  }
  const int cl = cfg.dec.back().second;
  s.norm("decoder.norm_out", cl);
  s.conv("decoder.conv_out", cl, 3, 3);
  return s.items;
}

std::vector<ParamSpec> vae_encoder_param_specs(const VaeCfg& cfg) {   // autoencoder/mod.rs:34,79-129,227-256
  Spec s;
  const int c0 = cfg.enc.front().second;
  s.conv("encoder.conv_in", 3, c0, 3);
  for (size_t i = 0; i < cfg.enc.size(); ++i) {
    const std::string p = "encoder.blocks." + std::to_string(i);
    const int ci = cfg.enc[i].first, co = cfg.enc[i].second;
    vae_resnet(s, p + ".res1", ci, co);
    vae_resnet(s, p + ".res2", co, co);
    if (i + 1 != cfg.enc.size()) s.conv(p + ".downsampler", co, co, 3);
  }
  const int cl = cfg.enc.back().first;
  vae_mid(s, "encoder.mid", cl);
  s.norm("encoder.norm_out", cl);
  s.conv("encoder.conv_out", cl, cfg.enc_out, 3);
  s.conv("quant_conv", cfg.enc_out, cfg.enc_out, 1);
  return s.items;
}

std::vector<ParamSpec> clip_param_specs(const ClipCfg& cfg) {   // clip/mod.rs:62-69,178-184,231-238,282-290
  SDXL_REQUIRE(cfg.n_vocab > 0 && cfg.n_state > 0 && cfg.embed_dim > 0 && cfg.n_ctx > 0 && cfg.n_layer > 0, "bad CLIP config");
  SDXL_REQUIRE(cfg.n_head > 0 && cfg.n_state == cfg.n_head * 64, "this engine's attention kernels are specialised for 64 channels per head");
  Spec s;
  const int c = cfg.n_state;
  s.add("token_embedding.weight", {cfg.n_vocab, c}, PK_LINEAR_W, (float)(kSqrt12 * 0.5), 0.f);
  s.add("position_embedding", {cfg.n_ctx, c}, PK_LINEAR_W, (float)(kSqrt12 * 0.1), 0.f);
  for (int i = 0; i < cfg.n_layer; ++i) {
    const std::string p = "blocks." + std::to_string(i);
    s.linear(p + ".attn.query", c, c);
    s.linear(p + ".attn.key", c, c);
    s.linear(p + ".attn.value", c, c);
    s.linear(p + ".attn.out", c, c, true, kResGain);
    s.norm(p + ".attn_ln", c);
    s.linear(p + ".mlp.fc1", c, 4 * c);
    s.linear(p + ".mlp.fc2", 4 * c, c, true, kResGain);
    s.norm(p + ".mlp_ln", c);
  }
  s.norm("layer_norm", c);
  s.add("text_projection", {c, cfg.embed_dim}, PK_LINEAR_W, Spec::wscale(c, 1.0), 0.f);
  return s.items;
}

}  // namespace sdxl
