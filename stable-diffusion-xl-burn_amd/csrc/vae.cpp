// SDXL KL-VAE decoder / encoder on MI355X (reference autoencoder/mod.rs; LatentDecoder stablediffusion/mod.rs:193-267).
// Same NHWC implicit-GEMM / GroupNorm kernels as the UNet; nearest-2x upsample (:313-318) is fused into the conv gather,
// the asymmetric-pad stride-2 PaddedConv2d (:326-407) is a plain stride-2 gather with zero fill on the bottom/right
// edge, and the single-head d=512 mid-block attention (:550-586) runs in the f16 mode as ONE flash kernel (attn_hd_kernel: scores
// never materialised) and in the strict-fp32 mode as QK^T GEMM -> row softmax -> PV GEMM over passes of 2048 queries (bounded).
#include "engine.h"

#include <cmath>

namespace sdxl {

VaeResW Vae::load_res(WeightBuilder& wb, const std::string& p, int cin, int cout) {
  VaeResW r; r.cin = cin; r.cout = cout;
  r.n1 = wb.norm(p + ".norm1"); r.c1 = wb.conv(p + ".conv1");
  r.n2 = wb.norm(p + ".norm2"); r.c2 = wb.conv(p + ".conv2");
  r.has_nin = wb.has(p + ".nin_shortcut.weight");
  if (r.has_nin) r.nin = wb.conv(p + ".nin_shortcut");
  return r;
}
VaeMidW Vae::load_mid(WeightBuilder& wb, const std::string& p, int c) {
  VaeMidW m; m.C = c;
  m.b1 = load_res(wb, p + ".block_1", c, c);
  m.an = wb.norm(p + ".attn.norm");
  m.q = wb.conv(p + ".attn.q"); m.k = wb.conv(p + ".attn.k"); m.v = wb.conv(p + ".attn.v");
  m.proj = wb.conv(p + ".attn.proj_out");
  m.b2 = load_res(wb, p + ".block_2", c, c);
  return m;
}

Vae::Vae(const VaeCfg& cfg, int compute_dt, WeightSource* dec_src, WeightSource* enc_src, hipStream_t st)
    : cfg_(cfg), cdt_(compute_dt) {
  const std::vector<ParamSpec> dspecs = vae_decoder_param_specs(cfg_), especs = vae_encoder_param_specs(cfg_);
  size_t bound = 0;
  // (no fragment-order weight images: the VAE's 1x1 convs run on >= 4096 rows per entry, never on the weights-in-registers kernel)
  if (dec_src) bound += WeightBuilder::arena_bound(dspecs, cdt_, false);
  if (enc_src) bound += WeightBuilder::arena_bound(especs, cdt_, false);
  warena_.reserve(bound + 4096);
  if (dec_src) {
    WeightBuilder wb(dspecs, *dec_src, warena_, cdt_, st);
    wb.wfrag = false;
    post_quant_ = wb.conv("post_quant_conv");
    d_conv_in_ = wb.conv("decoder.conv_in");
    d_mid_ = load_mid(wb, "decoder.mid", cfg_.dec.front().first);
    for (size_t i = 0; i < cfg_.dec.size(); ++i) {
      const std::string p = "decoder.blocks." + std::to_string(i);
      DecBlk b;
      b.r[0] = load_res(wb, p + ".res1", cfg_.dec[i].first, cfg_.dec[i].second);
      b.r[1] = load_res(wb, p + ".res2", cfg_.dec[i].second, cfg_.dec[i].second);
      b.r[2] = load_res(wb, p + ".res3", cfg_.dec[i].second, cfg_.dec[i].second);
      b.has_up = i + 1 != cfg_.dec.size();
      if (b.has_up) b.up = wb.conv(p + ".upsampler");
      d_blocks_.push_back(b);
    }
    d_norm_out_ = wb.norm("decoder.norm_out");
    d_conv_out_ = wb.conv("decoder.conv_out");
    SDXL_HIP(hipStreamSynchronize(st));
    has_dec_ = true;
  }
  if (enc_src) {
    WeightBuilder wb(especs, *enc_src, warena_, cdt_, st);
    wb.wfrag = false;
    e_conv_in_ = wb.conv("encoder.conv_in");
    for (size_t i = 0; i < cfg_.enc.size(); ++i) {
      const std::string p = "encoder.blocks." + std::to_string(i);
      EncBlk b;
      b.r[0] = load_res(wb, p + ".res1", cfg_.enc[i].first, cfg_.enc[i].second);
      b.r[1] = load_res(wb, p + ".res2", cfg_.enc[i].second, cfg_.enc[i].second);
      b.has_down = i + 1 != cfg_.enc.size();
      if (b.has_down) b.down = wb.conv(p + ".downsampler");
      e_blocks_.push_back(b);
    }
    e_mid_ = load_mid(wb, "encoder.mid", cfg_.enc.back().first);
    e_norm_out_ = wb.norm("encoder.norm_out");
    e_conv_out_ = wb.conv("encoder.conv_out");
    quant_ = wb.conv("quant_conv");
    SDXL_HIP(hipStreamSynchronize(st));
    has_enc_ = true;
  }
}
Vae::~Vae() {}

void Vae::res_block(Exec& ex, const VaeResW& w, const Act& x, int B, int H, int W, const Act& out) {
  // ResnetBlock::forward autoencoder/mod.rs:500-516
  const size_t mk = ex.act->mark();
  const size_t M = (size_t)B * H * W;
  const ConvGeom g3{B, H, W, H, W, 3, 1, 1, 0}, g1{B, H, W, H, W, 1, 1, 0, 0};
  Act gn1 = ex.alloc(M, w.cin, ex.cdt);
  // split-operand mode: the 1x1 nin_shortcut reads x as an HL16 copy scaled per entry by max|x| -- the statistics pass of norm1 reads all of x anyway and leaves
  // the maxima (GroupNormParams::absmax_out), so that copy needs no absmax pass of its own (0.5 / 1.1 GB tensors at the 512^2 / 1024^2 levels of a 1024^2 decode).
  // (GroupNorm statistics from the producing convolutions' epilogues, as in the UNet, were built and measured here: the staged epilogue they need costs these
  //  16-round launches more than the statistics passes it saves -- decode 39.4 -> 42.4 ms, f16 17.8 -> 21.8; not kept)
  float* nin_max = nullptr;
  if (w.has_nin && ex.cdt == DT_HL && x.dt == DT_F32 && w.nin.dt != DT_F32 && !x.gn_part && M % (size_t)B == 0)
    nin_max = (float*)ex.act->alloc(hl_scale_floats(B) * sizeof(float));
  run_groupnorm(ex, w.n1, x, B, H * W, gn1, true, cfg_.n_group, nin_max);
  Act h = ex.alloc(M, w.cout, ex.sdt);          // (read by a GroupNorm only: stays in the stream dtype)
  run_conv(ex, w.c1, gn1, w.cin, g3, h);
  Act gn2 = ex.alloc(M, w.cout, ex.cdt);
  run_groupnorm(ex, w.n2, h, B, H * W, gn2, true, cfg_.n_group);
  Epi e;
  if (w.has_nin) { run_conv(ex, w.nin, hl_operand(ex, w.nin, x, M, w.cin, B, nin_max), w.cin, g1, out); e.R = out; }
  else e.R = x;
  run_conv(ex, w.c2, gn2, w.cout, g3, out, e);
  ex.act->reset(mk);
}

void Vae::mid(Exec& ex, const VaeMidW& w, const Act& x, int B, int H, int W) {
  // Mid::forward :443-449 (in place on x); ConvSelfAttentionBlock::forward :550-586
  const size_t mk = ex.act->mark();
  const int HW = H * W, C = w.C;
  const size_t M = (size_t)B * HW;
  const ConvGeom g1{B, H, W, H, W, 1, 1, 0, 0};
  Act y = ex.alloc(M, C, ex.sdt);
  res_block(ex, w.b1, x, B, H, W, y);
  {
    const int kt = ex.cdt == DT_F16 ? 64 : 32;
    const int kpad = (int)round_up(HW, kt);        // K padding of the P V product
    const int rows_k = (int)round_up(HW, 128);     // "weight" row padding when K plays the B operand
    const int rows_v = (int)round_up(C, 128);
    SDXL_REQUIRE(C % kt == 0, "VAE attention channel count must be a multiple of the k-tile");
    Act hn = ex.alloc(M, C, ex.cdt);
    run_groupnorm(ex, w.an, y, B, HW, hn, false, cfg_.n_group);
    const float scale = (float)(1.0 / std::sqrt((double)C));   // (d^-0.25)^2, backend.rs:98
    if (ex.cdt == DT_F16 && C == 512) {
      // one head of 512 channels: the flash kernel (attn_hd_kernel) -- scores never materialised.  q | k from the 1x1 convs as
      // plain [M][C] rows, V^T straight out of the v conv's transposed epilogue (zero padded to whole key tiles)
      Act q = ex.alloc(M, C, ex.cdt), kk = ex.alloc(M, C, ex.cdt), o = ex.alloc(M, C, ex.cdt);
      void* vt = ex.act->alloc((size_t)B * C * kpad * dt_size(ex.cdt));
      if (!ex.dry && kpad != HW) launch_fill_zero(vt, (size_t)B * C * kpad * dt_size(ex.cdt), ex.s);
      run_conv(ex, w.q, hn, C, g1, q);
      run_conv(ex, w.k, hn, C, g1, kk);
      Epi ev; ev.n_split = 0; ev.Ct = vt; ev.ct_rows = C; ev.ct_ld = kpad; ev.rpb = HW;
      run_conv(ex, w.v, hn, C, g1, Act(nullptr, C, ex.cdt), ev);
      if (!ex.dry) {
        AttnParams p{};
        p.Q = q.p; p.ldq = C; p.K = kk.p; p.ldk = C; p.Vt = vt; p.vt_ld = kpad; p.O = o.p; p.ldo = C;
        p.dt = ex.cdt; p.B = B; p.H = 1; p.Nq = HW; p.Nk = HW; p.scale = scale; p.mask = nullptr; p.ldmask = 0;
        if (ex.prof) ex.prof->begin(Profiler::ATTENTION, 4.0 * B * (double)HW * HW * C, ex.s, HW, HW, B, 0);
        SDXL_REQUIRE(launch_attention_hd512(p, ex.s), "VAE mid attention: flash kernel refused the shape");
        SDXL_HIP(hipGetLastError());
        if (ex.prof) ex.prof->end(ex.s);
      }
      Epi ep; ep.R = y;
      run_conv(ex, w.proj, o, C, g1, y, ep);
      res_block(ex, w.b2, y, B, H, W, x);
      ex.act->reset(mk);
      return;
    }
    Act q = ex.alloc(M, C, ex.cdt);
    void* kbuf = ex.act->alloc((size_t)B * rows_k * C * dt_size(ex.cdt));
    void* vt = ex.act->alloc((size_t)B * rows_v * kpad * dt_size(ex.cdt));
    Act o = ex.alloc(M, C, ex.cdt);
    // scores are never materialised for the whole image: queries go through in passes of QT rows, so S / P stay bounded
    // (QT x HW: 0.54 GB + 0.27..0.54 GB at a 128x128 latent instead of 1.07 GB + 0.5..1 GB, and linear in HW beyond it).  QT = 8192 (round 6; was 2048): the
    // P V product of a pass is QT x C with K = HW -- at 2048 rows that is 88 tiles of 96x128 on 256 CUs, 281 us a pass whatever the rows; at 8192 rows 256
    // tiles of 128x128 fill the chip: 8 x 281 us -> 2 x ~330 us per decode, same k order per output (bit-identical)
    const int QT = HW < 8192 ? HW : 8192;
    float* S = (float*)ex.act->alloc((size_t)QT * HW * sizeof(float));
    void* P = ex.act->alloc((size_t)QT * kpad * dt_size(ex.cdt));
    // split-operand mode: P = softmax over up to 16 384 keys is ~6e-5 per entry -- stored times 2^12 so the lo halves of the HL16 image stay
    // f16-normal; the P V product undoes it through its accumulator scale (attention.hip measured 9e-5 vs 5e-7 rel for the same trick)
    float* pv_scale = ex.cdt == DT_HL ? (float*)ex.act->alloc(2 * sizeof(float)) : nullptr;
    if (!ex.dry) {
      launch_fill_zero(kbuf, (size_t)B * rows_k * C * dt_size(ex.cdt), ex.s);
      launch_fill_zero(vt, (size_t)B * rows_v * kpad * dt_size(ex.cdt), ex.s);
    }
    run_conv(ex, w.q, hn, C, g1, q);
    for (int b = 0; b < B; ++b) {   // K rows of batch b at kbuf + b*rows_k*C
      ConvGeom gb{1, H, W, H, W, 1, 1, 0, 0};
      run_conv(ex, w.k, Act((char*)hn.p + (size_t)b * HW * C * dt_size(ex.cdt), C, ex.cdt), C, gb,
               Act((char*)kbuf + (size_t)b * rows_k * C * dt_size(ex.cdt), C, ex.cdt));
    }
    Epi ev; ev.n_split = 0; ev.Ct = vt; ev.ct_rows = rows_v; ev.ct_ld = kpad; ev.rpb = HW;
    if (ex.cdt == DT_HL && (HW % 8 != 0 || M % 8 != 0)) {
      // HL16 rows are written as whole 8-key pieces: token counts that are not multiples of 8 (a 72 x 72 image: 9 x 9 latent) take the
      // transposed V^T out as fp32 and convert it, as the UNet does for such shapes (no generic twin of the split-operand GEMM)
      void* vt32 = ex.act->alloc((size_t)B * rows_v * kpad * sizeof(float));
      if (!ex.dry) launch_fill_zero(vt32, (size_t)B * rows_v * kpad * sizeof(float), ex.s);
      ev.Ct = vt32;
      run_conv(ex, w.v, hn, C, g1, Act(nullptr, C, DT_F32), ev);
      if (!ex.dry) launch_f32_to_hl(vt32, kpad, vt, kpad, (size_t)B * rows_v, kpad, ex.s);
    } else
    run_conv(ex, w.v, hn, C, g1, Act(nullptr, C, ex.cdt), ev);
    for (int b = 0; b < B; ++b) {
      Lin lk; lk.w = (char*)kbuf + (size_t)b * rows_k * C * dt_size(ex.cdt); lk.N = HW; lk.K = C; lk.Kpad = C; lk.Npad = rows_k; lk.cin = C;
      // (the P V contraction runs over the zero-padded key axis: P's columns and V^T's keys beyond HW are zeros, and whole k-tiles
      // keep it on the direct-to-LDS pipeline -- the split-operand mode has no generic twin)
      Lin lv; lv.w = (char*)vt + (size_t)b * rows_v * kpad * dt_size(ex.cdt); lv.N = C; lv.K = kpad; lv.Kpad = kpad; lv.Npad = rows_v; lv.cin = kpad;
      for (int q0 = 0; q0 < HW; q0 += QT) {
        const int nq = HW - q0 < QT ? HW - q0 : QT;
        const size_t row0 = ((size_t)b * HW + q0) * C * dt_size(ex.cdt);
        run_linear(ex, lk, Act((char*)q.p + row0, C, ex.cdt), nq, Act(S, HW, DT_F32));
        if (!ex.dry) launch_softmax_rows(S, HW, P, ex.cdt, kpad, nq, HW, kpad, scale, nullptr, 0, 0, ex.s, pv_scale ? 4096.0f : 1.0f, pv_scale);
        lv.acc_scale = pv_scale;
        run_linear(ex, lv, Act(P, kpad, ex.cdt), nq, Act((char*)o.p + row0, C, ex.cdt));
      }
    }
    Epi ep; ep.R = y;
    run_conv(ex, w.proj, o, C, g1, y, ep);
  }
  res_block(ex, w.b2, y, B, H, W, x);
  ex.act->reset(mk);
}

void Vae::run_decode(Exec& ex, const Act& in, int n, int h, int w, const Act& out) {
  // Autoencoder::decode_latent :67-70 -> Decoder::forward :203-216 -> DecoderBlock::forward :306-324
  const size_t M0 = (size_t)n * h * w;
  Act z = ex.alloc(M0, 4, io_dt());            // (4-channel ends run as plain fp32 in the split-operand mode)
  run_conv(ex, post_quant_, in, 4, ConvGeom{n, h, w, h, w, 1, 1, 0, 0}, z);
  const int c0 = cfg_.dec.front().first;
  Act cur = ex.alloc(M0, c0, ex.sdt);
  run_conv(ex, d_conv_in_, z, 4, ConvGeom{n, h, w, h, w, 3, 1, 1, 0}, cur);
  mid(ex, d_mid_, cur, n, h, w);
  for (size_t i = 0; i < d_blocks_.size(); ++i) {
    const DecBlk& b = d_blocks_[i];
    const int co = cfg_.dec[i].second;
    const size_t M = (size_t)n * h * w;
    Act next = b.has_up ? ex.alloc(4 * M, co, ex.sdt) : ex.alloc(M, co, ex.sdt);
    const size_t mk = ex.act->mark();
    Act a = ex.alloc(M, co, ex.sdt);
    Act bb = ex.alloc(M, co, ex.sdt);
    res_block(ex, b.r[0], cur, n, h, w, a);
    res_block(ex, b.r[1], a, n, h, w, bb);
    if (b.has_up) {
      res_block(ex, b.r[2], bb, n, h, w, a);
      run_conv(ex, b.up, hl_operand(ex, b.up, a, M, co, n), co, ConvGeom{n, h, w, 2 * h, 2 * w, 3, 1, 1, 1}, next);
      h *= 2; w *= 2;
    } else {
      res_block(ex, b.r[2], bb, n, h, w, next);
    }
    ex.act->reset(mk);
    cur = next;
  }
  const size_t M = (size_t)n * h * w;
  const int cl = cfg_.dec.back().second;
  Act gn = ex.alloc(M, cl, ex.cdt);
  run_groupnorm(ex, d_norm_out_, cur, n, h * w, gn, true, cfg_.n_group);
  run_conv(ex, d_conv_out_, gn, cl, ConvGeom{n, h, w, h, w, 3, 1, 1, 0}, out);
}

void Vae::run_encode(Exec& ex, const Act& in, int n, int H, int W, const Act& out) {
  // Autoencoder::encode_image :59-65 -> Encoder::forward :131-144 -> EncoderBlock::forward :258-268
  int h = H, w = W;
  const int c0 = cfg_.enc.front().second;
  Act cur = ex.alloc((size_t)n * h * w, c0, ex.sdt);
  run_conv(ex, e_conv_in_, in, 3, ConvGeom{n, h, w, h, w, 3, 1, 1, 0}, cur);
  for (size_t i = 0; i < e_blocks_.size(); ++i) {
    const EncBlk& b = e_blocks_[i];
    const int co = cfg_.enc[i].second;
    const size_t M = (size_t)n * h * w;
    const int h2 = b.has_down ? (h + 1 - 3) / 2 + 1 : h, w2 = b.has_down ? (w + 1 - 3) / 2 + 1 : w;
    Act next = ex.alloc((size_t)n * h2 * w2, co, ex.sdt);
    const size_t mk = ex.act->mark();
    Act a = ex.alloc(M, co, ex.sdt);
    res_block(ex, b.r[0], cur, n, h, w, a);
    if (b.has_down) {
      Act bb = ex.alloc(M, co, ex.sdt);
      res_block(ex, b.r[1], a, n, h, w, bb);
      // PaddedConv2d(pad left 0, right 1, top 0, bottom 1), stride 2 (:229-238, :384-407)
      run_conv(ex, b.down, hl_operand(ex, b.down, bb, M, co, n), co, ConvGeom{n, h, w, h2, w2, 3, 2, 0, 0}, next);
    } else {
      res_block(ex, b.r[1], a, n, h, w, next);
    }
    ex.act->reset(mk);
    cur = next; h = h2; w = w2;
  }
  mid(ex, e_mid_, cur, n, h, w);
  const size_t M = (size_t)n * h * w;
  const int cl = cfg_.enc.back().first;
  Act gn = ex.alloc(M, cl, ex.cdt);
  run_groupnorm(ex, e_norm_out_, cur, n, h * w, gn, true, cfg_.n_group);
  Act e8 = ex.alloc(M, cfg_.enc_out, io_dt());
  run_conv(ex, e_conv_out_, gn, cl, ConvGeom{n, h, w, h, w, 3, 1, 1, 0}, e8);
  run_conv(ex, quant_, e8, cfg_.enc_out, ConvGeom{n, h, w, h, w, 1, 1, 0, 0}, out);
}

// GroupNorm workspace of launch_groupnorm: [n][G][128][3] partials + [n][G][2] (mean, rstd), sized per call from the arena
// (a fixed allocation silently overflowed for n * n_group > 256)
float* Vae::gn_workspace(Exec& ex, int n) {
  return (float*)ex.act->alloc(groupnorm_workspace_floats(n, cfg_.n_group) * sizeof(float));
}

// two-pass execution: dry run sizes the arena, then the real run
#define VAE_RUN(...)                                                                   \
  do {                                                                                 \
    Exec ex; ex.s = s; ex.cdt = cdt_; ex.sdt = io_dt(); ex.act = &act_;               \
    act_.dry = true; act_.off = 0; act_.peak = 0; ex.dry = true;                       \
    __VA_ARGS__;                                                                       \
    const size_t peak = act_.peak;                                                     \
    act_.dry = false; ex.dry = false;                                                  \
    if (peak + 4096 > act_.cap) { SDXL_HIP(hipStreamSynchronize(s)); act_.reserve(peak + 4096); } \
    act_.off = 0; act_.peak = 0;                                                       \
    __VA_ARGS__;                                                                              \
  } while (0)

const float* Vae::decode(const float* latent, int n, int h, int w, hipStream_t s) {
  SDXL_REQUIRE(has_dec_, "this Vae was created without decoder weights");
  float* img = nullptr;
  VAE_RUN({
    ex.gn_partial = gn_workspace(ex, n);
    Act in = ex.alloc((size_t)n * h * w, 4, io_dt());
    Act out = ex.alloc((size_t)n * h * w * 64, 3, DT_F32);
    img = (float*)out.p;
    if (!ex.dry)   // x * (1/scale_factor), stablediffusion/mod.rs:265
      launch_nchw_to_nhwc(latent, 4 * h * w, in.p, io_dt(), n, 4, h * w, 4, (float)(1.0 / cfg_.scale_factor), s);
    run_decode(ex, in, n, h, w, out);
  });
  return img;
}
void Vae::decode_nchw(const float* latent, int n, int h, int w, float* out, hipStream_t s) {
  const float* img = decode(latent, n, h, w, s);
  launch_nhwc_to_nchw(img, DT_F32, 3, out, n, 3, 64 * h * w, 1.0f, s);
}
void Vae::latent_to_image(const float* latent, int n, int h, int w, unsigned char* out_hwc, hipStream_t s) {
  const float* img = decode(latent, n, h, w, s);
  launch_to_u8_image(img, DT_F32, 3, out_hwc, (size_t)n * 64 * h * w, s);
}
void Vae::encode_nchw(const float* img, int n, int H, int W, float* latent_out, hipStream_t s) {
  SDXL_REQUIRE(has_enc_, "this Vae was created without encoder weights");
  VAE_RUN({
    ex.gn_partial = gn_workspace(ex, n);
    Act in = ex.alloc((size_t)n * H * W, 3, io_dt());
    Act out = ex.alloc((size_t)n * (H / 8) * (W / 8), cfg_.enc_out, DT_F32);
    if (!ex.dry) launch_nchw_to_nhwc(img, 3 * H * W, in.p, io_dt(), n, 3, H * W, 3, 1.0f, s);
    run_encode(ex, in, n, H, W, out);
    if (!ex.dry)   // channels 0..4 (the mean) * scale_factor, autoencoder/mod.rs:63, stablediffusion/mod.rs:257-261
      launch_nhwc_to_nchw(out.p, DT_F32, cfg_.enc_out, latent_out, n, 4, (H / 8) * (W / 8), (float)cfg_.scale_factor, s);
  });
}
void Vae::image_to_latent(const unsigned char* img_hwc, int n, int H, int W, float* latent_out, hipStream_t s) {
  SDXL_REQUIRE(has_enc_, "this Vae was created without encoder weights");
  VAE_RUN({
    ex.gn_partial = gn_workspace(ex, n);
    Act in = ex.alloc((size_t)n * H * W, 3, io_dt());
    Act out = ex.alloc((size_t)n * (H / 8) * (W / 8), cfg_.enc_out, DT_F32);
    if (!ex.dry) launch_from_u8_image(img_hwc, in.p, io_dt(), 3, (size_t)n * H * W, s);
    run_encode(ex, in, n, H, W, out);
    if (!ex.dry)
      launch_nhwc_to_nchw(out.p, DT_F32, cfg_.enc_out, latent_out, n, 4, (H / 8) * (W / 8), (float)cfg_.scale_factor, s);
  });
}

}  // namespace sdxl
