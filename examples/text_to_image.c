/* A host program in plain C over the C ABI -- what the reference's src/bin/sample/main.rs does (tokens -> Embedder ->
 * Diffuser::sample_latent -> LatentDecoder::latent_to_image), with the Rust model objects replaced by libsdxl_mi355 handles.
 * Synthetic seeded weights (no checkpoint on the box); the prompt arrives as token ids because tokenisation is host string
 * code that stays with the caller (src/token/{clip,open_clip}.rs).  Writes a binary PPM.
 *
 *   gcc -std=gnu99 -O2 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/text_to_image.c \
 *       -Lstable-diffusion-xl-burn_amd/lib -lsdxl_mi355 -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/stable-diffusion-xl-burn_amd/lib \
 *       -o text_to_image && ./text_to_image out.ppm 1024 30
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime under -std=c11 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sdxl_mi355.h"

#define CHECK(x) do { if ((x) != SDXL_OK) { fprintf(stderr, "%s: %s\n", #x, sdxl_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float gauss(unsigned long long* s) {   /* Box-Muller on a 64-bit LCG: gen_noise() of stablediffusion/mod.rs:378-388 */
  double u[2];
  for (int i = 0; i < 2; ++i) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; u[i] = ((*s >> 11) + 1.0) / 9007199254740993.0; }
  return (float)(sqrt(-2.0 * log(u[0])) * cos(6.283185307179586 * u[1]));
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "out.ppm";
  const int res = argc > 2 ? atoi(argv[2]) : 1024, n_steps = argc > 3 ? atoi(argv[3]) : 30;
  const int lat = res / 8, S = 77;
  sdxl_ctx* ctx; sdxl_diffuser* diff; sdxl_vae* vae; sdxl_clip *clip, *open_clip;
  sdxl_unet_config ucfg; sdxl_vae_config vcfg; sdxl_clip_config c1, c2;
  sdxl_unet_config_base(&ucfg); sdxl_vae_config_default(&vcfg); sdxl_clip_config_clip_l(&c1); sdxl_clip_config_open_clip_bigg(&c2);
  CHECK(sdxl_ctx_create(0, &ctx));
  static float alphas[1000];                 /* LegacyDDPMDiscretization: linear in sqrt(beta), python/dump.py:29-31 */
  { double a = 1.0; for (int i = 0; i < 1000; ++i) { double b = sqrt(0.00085) + (sqrt(0.012) - sqrt(0.00085)) * i / 999.0; a *= 1.0 - b * b; alphas[i] = (float)a; } }
  CHECK(sdxl_diffuser_create_synthetic(ctx, &ucfg, SDXL_DTYPE_F16, 0, alphas, 1000, &diff));
  CHECK(sdxl_vae_create_synthetic(ctx, &vcfg, SDXL_DTYPE_F16, 0, 0, &vae));
  CHECK(sdxl_clip_create_synthetic(ctx, &c1, SDXL_DTYPE_F16, 11, &clip));
  CHECK(sdxl_clip_create_synthetic(ctx, &c2, SDXL_DTYPE_F16, 12, &open_clip));

  /* tokenize_text (stablediffusion/mod.rs:785-801): row 0 = "" (unconditional), row 1 = a 6-token prompt; pads 49407 / 0 */
  int32_t ids_clip[2 * 77], ids_open[2 * 77];
  for (int b = 0; b < 2; ++b)
    for (int t = 0; t < S; ++t) { ids_clip[b * S + t] = 49407; ids_open[b * S + t] = 0; }
  const int32_t prompt[6] = {320, 1125, 539, 550, 18376, 6765};
  for (int b = 0; b < 2; ++b) {
    const int n = b ? 6 : 0;
    ids_clip[b * S] = ids_open[b * S] = 49406;
    for (int t = 0; t < n; ++t) ids_clip[b * S + 1 + t] = ids_open[b * S + 1 + t] = prompt[t];
    ids_clip[b * S + 1 + n] = ids_open[b * S + 1 + n] = 49407;
  }
  int32_t *d_ids1, *d_ids2, *d_vals;
  float *h1, *h2, *pooled, *full, *y, *yr, *noise, *latent; unsigned char* img;
  const int C1 = c1.n_state, C2 = c2.n_state, E = c2.embed_dim, ADM = E + 6 * 256;
  HIP(hipMalloc((void**)&d_ids1, sizeof ids_clip)); HIP(hipMalloc((void**)&d_ids2, sizeof ids_open));
  HIP(hipMemcpy(d_ids1, ids_clip, sizeof ids_clip, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_ids2, ids_open, sizeof ids_open, hipMemcpyHostToDevice));
  HIP(hipMalloc((void**)&h1, (size_t)2 * S * C1 * 4)); HIP(hipMalloc((void**)&h2, (size_t)2 * S * C2 * 4));
  HIP(hipMalloc((void**)&pooled, (size_t)2 * E * 4)); HIP(hipMalloc((void**)&full, (size_t)2 * S * (C1 + C2) * 4));
  HIP(hipMalloc((void**)&y, (size_t)2 * ADM * 4)); HIP(hipMalloc((void**)&yr, (size_t)2 * (E + 5 * 256) * 4));
  /* Embedder::context for both rows in one batch-2 pass per encoder (:697-770), penultimate layer */
  CHECK(sdxl_clip_forward_hidden(clip, NULL, d_ids1, 2, S, c1.n_layer - 1, h1));
  CHECK(sdxl_clip_forward_hidden_pooled(open_clip, NULL, d_ids2, 2, S, c2.n_layer - 1, h2, pooled));
  CHECK(sdxl_ctx_synchronize(ctx));          /* the engine ran on the context's stream; the copies below use the null stream */
  HIP(hipMemcpy2D(full, (size_t)(C1 + C2) * 4, h1, (size_t)C1 * 4, (size_t)C1 * 4, (size_t)2 * S, hipMemcpyDeviceToDevice));   /* Tensor::cat(.., 2) */
  HIP(hipMemcpy2D(full + C1, (size_t)(C1 + C2) * 4, h2, (size_t)C2 * 4, (size_t)C2 * 4, (size_t)2 * S, hipMemcpyDeviceToDevice));
  const int32_t vals[2 * 6] = {res, res, 0, 0, res, res, res, res, 0, 0, res, res};   /* size | crop | ar */
  HIP(hipMalloc((void**)&d_vals, sizeof vals)); HIP(hipMemcpy(d_vals, vals, sizeof vals, hipMemcpyHostToDevice));
  CHECK(sdxl_conditioning_embedding(ctx, NULL, pooled, 2, E, d_vals, 6, 256, y));
  CHECK(sdxl_ctx_synchronize(ctx));
  (void)yr;

  sdxl_conditioning cond; memset(&cond, 0, sizeof cond);
  cond.unconditional_context_full = full;                 cond.context_full = full + (size_t)S * (C1 + C2);
  cond.unconditional_channel_context = y;                 cond.channel_context = y + ADM;
  cond.n = 1; cond.n_ctx = S; cond.height = res; cond.width = res;

  const size_t nl = (size_t)4 * lat * lat;
  float* hn = (float*)malloc(nl * 4); unsigned long long seed = 42;
  for (size_t i = 0; i < nl; ++i) hn[i] = gauss(&seed);
  HIP(hipMalloc((void**)&noise, nl * 4)); HIP(hipMalloc((void**)&latent, nl * 4)); HIP(hipMalloc((void**)&img, (size_t)res * res * 3));
  HIP(hipMemcpy(noise, hn, nl * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {        /* second pass replays the captured hipGraph */
    struct timespec a, b;
    CHECK(sdxl_ctx_synchronize(ctx)); clock_gettime(CLOCK_MONOTONIC, &a);
    CHECK(sdxl_sample_latent(diff, NULL, &cond, 7.5, n_steps, noise, latent));
    CHECK(sdxl_latent_to_image(vae, NULL, latent, 1, lat, lat, img));
    CHECK(sdxl_ctx_synchronize(ctx)); clock_gettime(CLOCK_MONOTONIC, &b);
    printf("pass %d: %d x %d, %d iterations: %.1f ms\n", rep, res, res, sdxl_step_count(n_steps, 0, 1000),
           (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6);
  }
  unsigned char* himg = (unsigned char*)malloc((size_t)res * res * 3);
  HIP(hipMemcpy(himg, img, (size_t)res * res * 3, hipMemcpyDeviceToHost));
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); return 1; }
  fprintf(f, "P6\n%d %d\n255\n", res, res); fwrite(himg, 1, (size_t)res * res * 3, f); fclose(f);
  long sum = 0; for (size_t i = 0; i < (size_t)res * res * 3; ++i) sum += himg[i];
  printf("wrote %s (mean pixel %.1f)\n", path, (double)sum / ((double)res * res * 3));
  sdxl_diffuser_destroy(diff); sdxl_vae_destroy(vae); sdxl_clip_destroy(clip); sdxl_clip_destroy(open_clip); sdxl_ctx_destroy(ctx);
  return 0;
}
