// Kernel launch interface of the MI355X (gfx950) SDXL engine.  Everything here is device-side HIP written
// for CDNA4 (64-lane wavefronts, MFMA, LDS); the host engine (unet.cpp / vae.cpp / sampler.cpp) only calls
// these launchers.  Layout convention everywhere: activations are NHWC / token-major [rows][channels] with
// an explicit row stride ("ld", in elements), so a tensor can live inside a wider (concat) buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdxl {

// compute precision of a model instance
// DT_HL: split-operand storage "HL16" (igemm_common.h): a logical fp32 value as two f16 numbers (hi, lo), 4 bytes per element,
// every 16 channels = 32 halfs [16 hi | 16 lo] -- operands of the 3-MFMA fp32-class GEMM the VAE runs at the reference's precision
enum DType : int { DT_F32 = 0, DT_F16 = 1, DT_HL = 2 };

static inline size_t dt_size(int dt) { return dt == DT_F16 ? 2 : 4; }

// ---------------------------------------------------------------------------------------------------------
// Implicit GEMM:  C[M,N] = gather(A)[M,K] * Wp[N,K]^T   (conv3x3 / conv1x1 / linear; NHWC; K = taps*Cin)
// Replaces burn Conv2d / nn::Linear call sites (reference unet/mod.rs:116-120,1036-1053,729-731,765-770,
// 793,801,903,929,975-984,1041; autoencoder/mod.rs:161-180,281-283,461-472,526-529).
struct IgemmParams {
  const void* A;        // source activations [B][Hin][Win] rows of lda elements (dtype a_dt)
  const void* W;        // packed weights [Npad][Kpad] in compute dtype, K order = (tap, cin), zero padded
  const void* Wf;       // optional: the same f16 weights in MFMA fragment order (launch_repack_wfrag) -- linear layers / 1x1 convs the
                        // weights-in-registers kernel may take (igemm_wreg.hip); null = not packed that way
  int a_dt;             // dtype of A in memory (DT_F16 / DT_F32); converted to compute dtype while staging
  int B, Hin, Win, Cin, lda;
  int Hout, Wout;       // M = B*Hout*Wout
  int ksize, stride, pad, up;   // up=1: nearest-2x upsample of the source fused into the gather
  int M, N, K, Kpad;    // N = logical packed columns (<= Npad)
  // epilogue:  v = acc + bias[n] + ebias[b*ebias_ld + n];  act;  + R;  store
  const float* bias;    // [Npad] fp32 (packed order) or null
  const float* ebias;   // [B][ebias_ld] fp32 or null (ResBlock time-embedding add, unet/mod.rs:1092)
  int ebias_ld;
  int rpb;              // rows per batch element (Hout*Wout or tokens) for ebias / transposed store
  int act;              // 0 none, 1 GEGLU: packed column pairs (x,gate) in 16-wide groups -> x*gelu_erf(gate);
                        // 2 GELU(erf), 3 QuickGELU x*sigmoid(1.702x)  (CLIP MLP, clip/mod.rs:296-320; generic kernel only)
  const void* R; int ldr; int r_dt;     // residual added after act (same logical shape as output)
  void* C; int ldc; int c_dt;           // output for columns n < n_split (after GEGLU: n/2)
  int n_split;          // columns >= n_split go to the transposed output (set = N when unused)
  void* Ct; int ct_rows; int ct_ld;     // Ct[b][n - n_split][key]  (ct_rows rows per batch, row stride ct_ld), dtype c_dt
  // LayerNorm folded into the GEMM (reference layernorm/mod.rs:34-49 followed by nn::Linear): A holds the RAW rows x,
  // W holds diag(gamma) W, bias holds beta W + b, and the epilogue applies  v = rstd[m]*acc - rstd[m]*mu[m]*ln_cs[n] + bias[n]
  // with (mu, rstd) from the partial row sums the PRODUCER of x left behind: ln_stat[slot][m] = (mean, M2 = sum (x - mean)^2) over the
  // 64 columns of slot, ln_slots = K/64 slots per row, summed here in slot order (deterministic).  null -> plain GEMM.
  const float* ln_stat; int ln_slots; const float* ln_cs; float ln_invc; float ln_eps;
  const float* ln_eps_ptr;   // optional device scalar overriding ln_eps
  // when set, the staged epilogue stores (mean, M2) -- shifted sums, no sum x^2 - (sum x)^2 -- of every 64-column group of the stored output rows into
  // stat_out[n/64][m] (plain stores, every entry written once) -- the statistics of the LayerNorm that reads this output
  float* stat_out; int stat_slots;
  // split-K workspace (optional): fp32 partial slabs [tile][slice][256 x 128] + one arrival counter per tile (zero between
  // launches: the last-arriving slice re-arms it).  splitk is filled by the launcher (igemm_splitk_slices); 0 / 1 = off.
  float* splitk_ws; size_t splitk_ws_bytes; unsigned* splitk_cnt; int splitk;
  // cross-attention fused into this (query) projection: when xa_k is set the epilogue replaces the q tile by
  // softmax(q K_h^T * xa_scale) V_h per 64-column head before the store (igemm_glds.hip xattn_inplace).  xa_k = the context
  // keys / values of the batch entries in MFMA operand order (launch_xattn_pack); batch entry of a row = m / rpb.
  const void* xa_k; int xa_nctx; float xa_scale;
  // split-precision form (round 6): xa_k holds the HI halves f16(x) of the context keys / values and xa_k_lo the LO halves f16(x - hi), both in the
  // operand order of launch_xattn_pack; the epilogue then splits q and P the same way and runs three MFMAs per product (Kh qh + Kh ql + Kl qh, likewise
  // V^T P^T): the arithmetic of attn_d64_hl_kernel inside the projection's epilogue.  null = plain f16 attention.  One-MFMA-row wave tiles only.
  const void* xa_k_lo;
  // GroupNorm statistics of the stored output, from the epilogue (256x128 kernel; igemm_gn_part_ok): gn_part[M/256][N] (mean, M2)
  // of every column over the 256 rows of a tile -- the consumer's GroupNorm merges row tiles and channels (norm.hip, chan_part)
  float* gn_part;
  const float* acc_scale;   /* [0] = 1 / weight scale, [1] != 0: every packed weight is exactly one f16 (lo halves all zero) */  // DT_HL compute: the packed weights carry a power-of-two factor (exact); device scalar 1 / factor the epilogue multiplies the
                           // accumulators by (lives in the weight arena, so replicas that receive the arena by broadcast need no host copy); null = 1
  const float* a_scale;     // DT_HL compute: 2^-e of an A operand converted by launch_f32_to_hl_scaled (null = 1); exact.  One value per batch entry:
  int a_scale_rpb;          // output rows per entry of a_scale (row m reads a_scale[m / a_scale_rpb]); 0 = one value for all rows
  int hl_wexact_ok; // A/B knob (sdxl_debug_set "hl_weights_exact", default 1): split-operand launches may leave out the w_lo MFMAs when acc_scale[1] says every weight is one f16
  int xa_vec64;     // measure builds (sdxl_debug_set "xa_vec64"): the fused cross-attention epilogue reads its per-column vectors with the original 64-lane
  // weight warming (round 4): the launch also brings `warm_bytes` of the weights a LATER GEMM of the same stream will read into the
  // memory-side Infinity Cache -- spare workgroups behind the tile grid (kernels whose grid leaves CUs idle: igemm_wreg_selected) do
  // nothing but read them.  No effect on any result; null = off.
  const void* warm[3]; unsigned warm_bytes[3];     // up to three regions (filled from the front)
  int splitk_wt;    // split-K slabs published by write-through stores instead of plain stores + release fence (set by the launcher from the A/B knob "splitk_wt", default 1)
                    // VMEM loads instead of the scalar cache -- the hazard experiment of DESIGN 9.2 / 10.4
  int wreg_xcd2d;   // A/B knob (sdxl_debug_set "wreg_xcd2d"): weights-in-registers kernel, XCDs own 2-D patches of tiles (half the column tiles x ~ a quarter of
                    // the row tiles each) instead of whole row-tile runs -- fewer unique operand bytes per XCD's L2; set by the launcher
  int epi_staged;   // A/B knob (sdxl_debug_set "igemm_epilogue_staged"): 1 = LDS-staged epilogue everywhere, 0 = direct row-per-lane where it applies
  // f16 SHADOW of an fp32 residual stream (round 6; weights-in-registers kernel only, igemm_wreg_ok): next to the fp32 rows C the epilogue stores
  // shadow[m][n] = f16(value * shadow_gamma[n]) -- the A operand of the GEMM behind the NEXT LayerNorm (gamma = that norm's), whose weights stay
  // un-folded: LN(x) W + b = rstd (x o gamma) W - rstd mu (gamma W) + (beta W + b).  With stat_out (the fp32 rows' statistics) the LayerNorm launch
  // between two GEMMs of a split-operand transformer block disappears.  null = off.
  void* shadow; int shadow_ld; const float* shadow_gamma;
  float shadow_lo_scale;    // > 0: the shadow row is [hi (N columns) | lo (N columns)], lo = f16((value * gamma - hi) * shadow_lo_scale): the (hi, lo) A operand of a GEMM packed (w | w / scale).
                            // < 0: the shadow is an HL16 tensor (shadow_ld in LOGICAL elements, un-scaled lo halves): read as an f16 row of 2 N columns by a GEMM whose weight is
                            // packed twice in the HL16 interleave (launch_pack_linear_hilo mode 2)
};
bool igemm_gn_part_ok(const IgemmParams& p);
// shapes the fused cross-attention epilogue takes (f16 operands, head dim 64, <= 96 context tokens); otherwise run the
// projection and the attention kernel separately
bool igemm_xattn_ok(int a_dt, int c_dt, int M, int N, int K, int rpb, int n_ctx);
// K [B][n_ctx][C], V^T [B][C][vt_ld] (f16) -> operand-order image of xattn_pack_bytes(B, C) bytes (once per prompt)
size_t xattn_pack_bytes(int B, int C);
void launch_xattn_pack(const void* K, const void* Vt, void* out, int B, int C, int n_ctx, int vt_ld, hipStream_t s);
int igemm_splitk_slices(const IgemmParams& p);                       // 1 or 3: depends on one batch entry's shape only
size_t igemm_splitk_ws_bytes(int batch, int rows_per_entry, int n_max);   // slab bytes a plan must provide
constexpr int kSplitkCounters = 4096;                                // arrival counters a plan must provide (zeroed once)
void launch_igemm(const IgemmParams& p, int compute_dt, hipStream_t s);
// [Npad][Kpad] f16 row-major packed weights -> fragment order (same bytes): per 32-row block and 64-deep k-tile four 1-KiB MFMA
// A-operand fragments, contiguous over k -- what igemm_wreg_kernel streams straight into registers.  Npad % 32 == 0, Kpad % 64 == 0.
void launch_repack_wfrag(const void* w, void* wf, int Npad, int Kpad, hipStream_t s);
bool igemm_wreg_ok(const IgemmParams& p);   // shapes the weights-in-registers kernel takes (plain f16 linear / 1x1, N % 128 == 0, Wf set)
void igemm_set_tsw(int v);       // A/B knob (sdxl_debug_set "igemm_tsw"): 0 = no operand-swapped k-loop for the transposed part of a fused QKV projection
bool igemm_wreg_selected(const IgemmParams& p);   // the auto selection (variant 0) would run this launch on the weights-in-registers kernel
void igemm_set_warm(int v);      // A/B knob (sdxl_debug_set "igemm_warm"): 0 = no weight warming workgroups; read when a UNet plan records its GEMM sequence
int igemm_warm_enabled();
void igemm_set_splitk_wt(int v);
void igemm_set_hl_tile96(int v); // A/B knob (sdxl_debug_set "hl_tile96"): 0 = the split-operand GEMMs never take the 96-row tile
void igemm_set_wreg_xcd2d(int v);
void igemm_set_wide_db(int v);   // A/B knob (sdxl_debug_set "wide_db"): 0 = wide GEGLU kernel with the rolled fragment reads
void igemm_set_wreg(int v);      // A/B knob (sdxl_debug_set "igemm_wreg"): 0 = the auto selection never picks the weights-in-registers kernel
void igemm_set_variant(int v);   // debug / benchmarking knob: -1 generic kernel only, 0 auto, 1..3 forced fast-path tile
void igemm_set_hl_weights_exact(int v); // A/B: 0 keeps all three MFMAs per product even where the packed weights are exact f16 values
void igemm_set_xa_vec64(int v);          // measure builds: see IgemmParams::xa_vec64
void igemm_set_epilogue_staged(int v);   // A/B: 1 forces the LDS-staged epilogue (default 0: direct row-per-lane epilogue on whole wave tiles)
void igemm_set_unrolled(int v);   // auto selection: pipelined kernels with the k-loop unrolled by the ring depth (default on)
#ifdef SDXL_MEASURE
void igemm_set_timeline(void* device_buf);
void igemm_set_wreg_timeline(void* device_buf);   // igemm_wreg.hip: [workgroups][8 waves][16] coarse stamps
void igemm_set_wide_timeline(void* device_buf);   // igemm_glds.hip: [workgroups][8 waves][8] coarse stamps of the wide (GEGLU) kernel   // igemm_measure.hip: stamp buffer of the timeline kernel variants
int igemm_timeline_words();
#endif
void igemm_glds_init();          // allocates the zero page the DMA fast path reads halo pixels from (call once per process)

// per-entry scale workspace of launch_f32_to_hl_scaled (declared further down)
constexpr int kHlAbsBlocks = 128;
static inline size_t hl_scale_floats(int nb) { return (size_t)nb * (kHlAbsBlocks + 1); }
static inline float* hl_scale_inv(float* scale_io, int nb) { return scale_io + (size_t)nb * kHlAbsBlocks; }

// ---------------------------------------------------------------------------------------------------------
// GroupNorm (32 groups, NHWC) -- reference groupnorm/mod.rs:52-82 (+ SiLU silu.rs:14-16)
struct GroupNormParams {
  const void* X; int x_dt; int ldx;    // [B][HW] rows, C channels
  void* Y; int y_dt; int ldy;
  const float* gamma; const float* beta;
  float* partial;      // workspace: groupnorm_workspace_floats(B, G) floats
  int B, HW, C, G;
  float eps;
  const float* eps_ptr; // optional device scalar overriding eps (per-norm eps stored with the weights)
  int silu;            // fuse x*sigmoid(x) after the affine
  int nsplit;          // filled by the launcher helper
  // statistics left by the PRODUCER of X (IgemmParams::gn_part): chan_part[B][chan_rt][C] (mean, M2) over chan_rows rows each;
  // when set, no statistics kernel runs -- the apply kernel merges row tiles and channels in its prologue
  const float* chan_part; int chan_rt; int chan_rows;
  // optional side output of the statistics pass (fp32 input only): per batch entry kHlAbsBlocks block maxima of |x| -- the partials
  // launch_f32_to_hl_scaled's conversion kernel reduces (the statistics pass reads the whole tensor anyway: a split-operand skip convolution of the
  // same ResBlock input then needs no absmax pass).  Layout [B][kHlAbsBlocks]; slots beyond the row splits are written as zeros.
  float* absmax_out;
};
int  groupnorm_nsplit(int B, int HW, int C);
// workspace of launch_groupnorm for B batch entries (a run over entries [b0, b0+nb) of a larger plan may use the slice at
// b0 * groupnorm_workspace_floats(1, G)): per (entry, group) kGnMaxSplit row-split partials (count, mean, M2) + (mean, rstd)
constexpr int kGnMaxSplit = 128;
static inline size_t groupnorm_workspace_floats(int B, int G) { return (size_t)B * G * (kGnMaxSplit * 3 + 2); }
void launch_groupnorm(const GroupNormParams& p, hipStream_t s);

// LayerNorm over the last dim -- reference layernorm/mod.rs:34-49
struct LayerNormParams {
  const void* X; int x_dt; int ldx;
  void* Y; int y_dt; int ldy;
  const float* gamma; const float* beta;
  int rows, C; float eps;
  const float* eps_ptr;   // optional device scalar overriding eps
  float dup_scale;        // f16 output only, != 0: the row is written twice.  > 0: y[0, C) = f16(LN(x)) and y[C, 2C) = that f16 value * dup_scale (a power of two) --
                          // the A operand of a GEMM whose weights are packed as (hi | lo / dup_scale) halves along K (launch_pack_linear_hilo mode 0).
                          // < 0: y[C, 2C) = f16((LN(x) - hi) * |dup_scale|), the LO half of the activation -- against weights packed as (w | w / |dup_scale|) (mode 1)
};
void launch_layernorm(const LayerNormParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// Fused attention softmax(Q K^T * scale) V, head dim 64 -- reference backend.rs:88-128 (Backend::qkv_attention)
struct AttnParams {
  const void* Q; int ldq;      // [B][Nq] rows, head h at columns h*64..
  const void* K; int ldk;      // [B][Nk] rows
  const void* Vt; int vt_ld;   // V transposed: [B][H*64][vt_ld], keys contiguous, zero beyond Nk
  void* O; int ldo;
  int dt;                      // dtype of Q,K,Vt,O (compute dtype)
  int B, H, Nq, Nk;
  float scale;                 // 1/sqrt(d)
  const float* mask; int ldmask;   // optional additive [Nq][Nk] fp32 (0 / -inf), null in UNet/VAE
  int q_dt = 0;                // split-operand kernel only: DT_F32 (0) or DT_HL -- Q rows in HL16 (ldq in logical elements)
  int o_dt = 0;                // split-operand kernel only: DT_F32 (0), DT_HL -- O written as HL16 rows (ldo in logical elements), the next GEMM's operand -- or DT_F16 (rounded once: an f16 out-projection's operand)
  // key halves on DIFFERENT workgroups (attn_d64_mix_kernel level 2, round 4): workspace of attention_xsplit_ws_bytes(B, H, Nq) bytes and
  // attention_xsplit_counters(B, H, Nq) zero-initialised tickets (they re-arm themselves); null = that form is never picked
  float* xws = nullptr; unsigned* xcnt = nullptr;
  int demote = 0;              // split-operand kernel only (precision-frontier instrument): lo halves of Q and P dropped -- with K / V^T lo halves zeroed
                               // by the caller the kernel computes the f16 flash kernel's products (f16 q, k, v, p; fp32 accumulation)
};
size_t attention_xsplit_ws_bytes(int B, int H, int Nq);
size_t attention_xsplit_counters(int B, int H, int Nq);
void attention_set_xsplit(int v);      // A/B knob (sdxl_debug_set "attn_xsplit"): 0 = never pick the cross-workgroup key split
void launch_attention_d64(const AttnParams& p, hipStream_t s);
// split-operand (fp32-class) head-dim-64 attention: Q / O fp32 (ldq / ldo in floats), K [B][Nk] rows and Vt [B][H*64][vt_ld] rows in
// HL16 (ldk / vt_ld in LOGICAL elements: a row is 2 * ld halfs), no mask; vt_ld a multiple of 64 keys, zero beyond Nk.  Returns false
// when the shape / alignment needs launch_attention_d64 on fp32 tensors.
bool launch_attention_d64_hl(const AttnParams& p, hipStream_t s);
// the same contraction for ONE wide head of 512 channels (VAE mid block): flash kernel, scores never materialised; Q/K/O rows
// hold the heads at columns h*512, Vt is [B][H*512][vt_ld] with zero columns up to a multiple of 32 keys.  f16, no mask;
// returns false when the shape / alignment needs the unfused path.  p.scale = d^-1/2.
bool launch_attention_hd512(const AttnParams& p, hipStream_t s);
void attention_init();                 // zero page for the DMA-staged f16 kernel (once per process)
#ifdef SDXL_MEASURE
void attention_set_timeline(void* device_buf);   // [workgroups][4 waves][8] coarse stamps of the key-split attention body
#endif
void attention_set_variant(int v);     // -1: generic kernel only, 0: auto, 1: DMA-staged 16x16x32 kernel, 2: 32x32x16 deferred-max kernel, 6: key-split kernel

// row softmax for the unfused attention path (VAE mid block, d=512, 1 head): P[r][:] = softmax(S[r][:]*scale + mask)
// S fp32 [rows][lds] (scores from igemm), P in dtype p_dt [rows][ldp]; columns n..npad-1 of P are zero filled.
void launch_softmax_rows(const float* S, int lds, void* P, int p_dt, int ldp, int rows, int n, int npad, float scale,
                         const float* mask, int ldmask, int mask_rows, hipStream_t s, float p_scale = 1.0f, float* p_scale_out = nullptr);
// p_scale / p_scale_out: P is stored times the power of two p_scale and {1 / p_scale, 0} is left at p_scale_out -- the accumulator scale
// (IgemmParams::acc_scale layout) of the split-operand P V product that consumes it

// ---------------------------------------------------------------------------------------------------------
// Small-M linear (GEMV): Y[b][n] = act_in(X[b][:]) . Wp[n][:] + bias[n]   (time/label/emb MLPs, M<=8)
struct GemvParams {
  const float* X; int ldx;     // fp32 [Bm][K]
  const void* W; int w_dt; int Kpad;   // packed [N][Kpad]
  const float* bias;           // fp32 or null
  float* Y; int ldy;           // fp32 [Bm][N]
  const float* Yadd;           // optional fp32 [Bm][N] added to the result (t_emb + label_emb)
  int Bm, N, K;
  int silu_in;                 // apply SiLU to X while loading
  int silu_out;                // apply SiLU to the result
};
void launch_gemv(const GemvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// elementwise / layout
// sinusoidal embedding (unet/mod.rs:21-39): out[b][0:half]=cos(t*f), [half:]=sin(t*f); t read from device
void launch_timestep_embedding(const float* t_dev, int t_stride, float* out, int Bm, int dim, hipStream_t s);
// NCHW fp32 -> NHWC (dtype dt), optional batch replication (src batch stride 0)
void launch_nchw_to_nhwc(const float* src, int src_batch_stride, void* dst, int dt, int B, int C, int HW, int ldd,
                         float scale, hipStream_t s);
void launch_nhwc_to_nchw(const void* src, int dt, int lds, float* dst, int B, int C, int HW, float scale, hipStream_t s);
// generic cast copy rows: dst[r][c] = src[r][c]
void launch_copy_rows(const void* src, int sdt, int lds, void* dst, int ddt, int ldd, int rows, int C, hipStream_t s);
void launch_fill_zero(void* p, size_t bytes, hipStream_t s);
// fp32 rows -> split-operand HL16 rows (C % 16 == 0, row strides in logical elements, dst stride % 16 == 0)
// exact[0] = 1.0f if every x * wscale of src[0..n) is exactly one f16 value (the packed lo halves are all zero), else 0.0f;
// `accumulate`: AND with the value already there (several tensors of one fused matrix)
void launch_f16_exact(const float* src, size_t n, float wscale, float* exact, hipStream_t s, bool accumulate = false);
void launch_f32_to_hl(const void* src, int lds, void* dst, int ldd, size_t rows, int C, hipStream_t s);
// the same for a tensor whose range the model does not bound (residual stream, VAE hidden state): converted times the power of two
// that brings max|x| into [2^13, 2^14) -- PER BATCH ENTRY (nb entries of rows / nb rows each: an entry's bits never depend on its batch
// neighbours).  scale_io: hl_scale_floats(nb) device floats -- [nb][kHlAbsBlocks] per-block max|x| partials (no atomics, no memset: every
// launch rewrites them), then the nb factors 2^-e at hl_scale_inv(scale_io, nb): pass that as IgemmParams::a_scale (a_scale_rpb = the GEMM's output rows per entry)
void launch_f32_to_hl_scaled(const void* src, int lds, void* dst, int ldd, size_t rows, int C, float* scale_io, hipStream_t s, int nb = 1,
                             bool have_partials = false);   // have_partials: a GroupNorm statistics pass over the same tensor left the maxima (GroupNormParams::absmax_out)
unsigned count_nonfinite(const void* src, int dt, int lds, size_t rows, int C, hipStream_t s);   // debugging aid (SDXL_NAN_CHECK): synchronises
void launch_f16_to_hl(const void* src, int lds, void* dst, int ldd, size_t rows, int C, hipStream_t s);   // f16 rows -> HL16 rows (lo = 0)
// fp32 rows -> two plain f16 row sets hi = f16(x), lo = f16(x - hi) (row stride ld16 each): the context images of the split-precision fused cross-attention
void launch_f32_to_f16_pair(const float* src, int lds, void* hi, void* lo, int ld16, size_t rows, int C, hipStream_t s);
void launch_hl_zero_lo(void* dst, int ldd, size_t rows, int C, hipStream_t s);   // HL16 rows: lo halves := 0 (precision-frontier instrument, UNet hl_demote)
void launch_round_f16(float* p, size_t n, hipStream_t s);   // p[i] = float(half(p[i])): parameters as a HalfPrecisionSettings record holds them
void launch_i32_to_f32(const int* src, float* dst, int n, hipStream_t s);
// CLIP text encoder (clip/mod.rs:99-105,139-147): x[b][t][:] = tok[ids[b][t]][:] + pos[t][:] (tables in dtype w_dt);
// eot[b] = first index of max(ids[b][:]); sel[b][:] = x[b][eot[b]][:] as fp32; additive causal mask [n][n] (0 / -inf)
void launch_embed_tokens(const int* ids, const void* tok, const void* pos, int w_dt, void* x, int x_dt, int ldx, int B, int S,
                         int C, int n_vocab, hipStream_t s);
void launch_argmax_rows(const int* ids, int* out, int B, int S, hipStream_t s);
void launch_gather_rows(const void* x, int x_dt, int ldx, const int* idx, int S, float* out, int B, int C, hipStream_t s);
void launch_causal_mask(float* out, int n, hipStream_t s);
// conditioning_embedding (unet/mod.rs:41-57): out[n][E + w*dim] = [pooled | sinusoidal embedding of each of the w ints]
void launch_conditioning_embedding(const float* pooled, int E, const int* vals, int w, int dim, float* out, int n, hipStream_t s);

// DDIM step table entry (host f64 -> f32): stablediffusion/mod.rs:407-428
struct StepCoef { float t; float sqrt_a; float sqrt_1ma; float sqrt_ap; float sqrt_1map; float cfg; float pad0, pad1; };
// eps = u + (c-u)*cfg (or c when !use_cfg); x0 = (x - eps*sqrt_1ma)/sqrt_a; x = x0*sqrt_ap + eps*sqrt_1map
// eps_nhwc: UNet output [Bu][HW][4] fp32-or-T rows; latent NCHW fp32 [n][4][HW].  Also refreshes the UNet input
// (NHWC, replicated for cond/uncond) and advances *step_idx.
struct DdimParams {
  float* latent;             // [n][4][HW] fp32 NCHW (state)
  const void* eps; int eps_dt; int eps_ld;   // UNet output NHWC rows: cond batch entries [0,n), uncond [n,2n)
  const StepCoef* table; int* step_idx;
  int n, HW; int use_cfg;
  // inpainting (optional): before the *next* UNet call latent = mask ? latent : ref*sqrt_a(next)+noise*sqrt_1ma(next)
  const float* ref; const unsigned char* mask; const float* step_noise; int n_steps_total;
  void* unet_in; int in_dt; int in_ld; int in_rep;   // next UNet input NHWC [in_rep*n][HW][4]
  float* t_out;              // device scalar(s): timestep for the next UNet call
};
// do_update=0: first call of a trajectory -- sets *step_idx = 0, applies the step-0 inpaint blend, writes the UNet
// input and timestep.  do_update=1: DDIM update with table[*step_idx], blend for the next step, then advances.
void launch_ddim_step(const DdimParams& p, int do_update, hipStream_t s);
// noised = latent*sa + noise*sb   (refine_latent :363-367)
void launch_axpby(float* dst, const float* a, float sa, const float* b, float sb, size_t n, hipStream_t s);
// image post-process (stablediffusion/mod.rs:210-230): u8 = trunc(clamp(((x+1)/2)*255, 0, 255)), NHWC rows of 3
void launch_to_u8_image(const void* src, int dt, int lds, unsigned char* dst, size_t pixels, hipStream_t s);
// u8 HWC image -> NHWC activation (x/255)*2-1  (image_to_latent :239-255)
void launch_from_u8_image(const unsigned char* src, void* dst, int dt, int ldd, size_t pixels, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------
// weights: synthetic fill (bit-identical to oracle/config.py synth_values) and packing into device layouts
void launch_synth_fill(float* dst, size_t numel, uint64_t key, float scale, float mean, hipStream_t s);
// canonical Linear [K][N] fp32 (burn layout) -> packed [Npad][Kpad] (dt); optional GEGLU interleave
void launch_pack_linear(const float* src, void* dst, int dt, int K, int N, int Kpad, int Npad, int geglu,
                        int n_offset, hipStream_t s, const float* kscale = nullptr,    // kscale[k]: LayerNorm gamma fold
                        float wscale = 1.0f);                                          // power-of-two factor of the DT_HL packing
// the same packing with every weight as TWO f16 values along a doubled K: dst[n][k] = f16(w), dst[n][K + k] = f16((w - f16(w)) * lo_scale) -- against the A
// operand [a | a / lo_scale] (LayerNormParams::dup_scale) the f16 GEMM multiplies a by hi + lo: the weights are not rounded (22 significand bits), the
// activations once.  K % 32 == 0; Kpad = 2 K.
void launch_pack_linear_hilo(const float* src, void* dst, int K, int N, int Npad, int geglu, float lo_scale, hipStream_t s, int mode = 0, int n_offset = 0);   // n_offset: first packed row (fused matrices: Npad = this part's rows)
// mode 1: dst[n][K + k] = f16(f16(w) / lo_scale) -- the weight twice, for an A operand that carries (hi | lo * lo_scale) ACTIVATION halves (exact for f16-representable weights)
// mode 2: the weight twice in the HL16 INTERLEAVE -- dst[n][32 g + j] = dst[n][32 g + 16 + j] = f16(w[16 g + j]) (K % 16 == 0): an HL16 activation row of C logical channels IS an
//         f16 row of 2 C columns [16 hi | 16 lo | ...], so any f16 GEMM kernel multiplies (hi + lo) by w -- two MFMAs per product, exact for f16-representable weights
// LayerNorm fold helpers: column sums of the packed (rounded) weight rows; beta . W + bias in canonical column order
void launch_colsum_packed(const void* wp, int dt, int Kpad, int nrows, float* cs, hipStream_t s, const float* kscale = nullptr, int K = 0, int interleave = 0);   // kscale: cs[r] = sum_k kscale[k] packed[r][k]; interleave: the packed row holds the K weights twice in the HL16 interleave (first copies summed)
void launch_beta_dot(const float* w, const float* beta, const float* bias, float* out, int K, int N, hipStream_t s);
// canonical conv [Cout][Cin][kh][kw] fp32 -> packed [Npad][Kpad], k = (kh*kw_idx)*Cin + c
void launch_pack_conv(const float* src, void* dst, int dt, int Cout, int Cin, int ks, int Kpad, int Npad,
                      hipStream_t s, float wscale = 1.0f);   // wscale: power-of-two factor of the DT_HL packing (undone by IgemmParams::acc_scale)
void launch_absmax(const float* src, size_t n, float* out_dev, hipStream_t s, bool accumulate = false);   // *out_dev = max(|src[i]|[, *out_dev])
// bias vector permuted the same way as GEGLU-packed columns (fp32 -> fp32)
void launch_pack_bias(const float* src, float* dst, int N, int Npad, int geglu, int n_offset, hipStream_t s);

}  // namespace sdxl
