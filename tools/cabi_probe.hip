// Torch-free GPU probe over the C ABI (include/ptts.h): decode-step latency of the decoder-LM engine and DAC decode time, with
// synthetic weights filled on the device. Starts in seconds on a fresh box (no `import torch`: 1-2 minutes of GPU-box time per call),
// and doubles as the plain-C++ example of the boundary: everything the product does per token / per frame is reached through the
// symbols declared in include/ptts.h with device pointers and sizes only.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cabi_probe.hip -o tools/cabi_probe \
//         -Lparler_tts_amd -lptts_hip -Wl,-rpath,'$ORIGIN/../parler_tts_amd'
//   tools/cabi_probe lm  <batch> [large] [fp32] [fp8] [kv8] [layers=<n>] [ctx=<prompt positions>] [tag=<text>] [dump=<file> [steps=<n>]] [eager=<n>]
//        eager=<n>: only a prefill + n EAGER decode forwards (ptts_push_tokens + ptts_step_forward), no hipGraph is captured or launched:
//        the target of `rocprofv3 --pmc ...` passes (which crash on graph replays in this image; tools/prof_eager.py without torch),
//        e.g. `rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/cabi_probe lm 1 ctx=431 eager=24` = the bench's timed context
//   tools/cabi_probe dac <batch> [frames=<n>] [f32] [reps=<n>] [tag=<text>] [dump=<file>]
//   tools/cabi_probe t5  <batch> [tokens=<n>] [layers=<n>] [fp32] [reps=<n>] [tag=<text>]   (round 5: the T5 description encoder, ptts_t5_*)
//   tools/cabi_probe cmp <file a> <file b>
//   (environment knobs as for tools/step_probe2.py: PTTS_NO_GEMV, PTTS_GEMV_ROWS, PTTS_GEMV_STAGE, PTTS_NO_FO, PTTS_DAC_NO_FUSE_RES ...)
//
// `dump=` writes what the run computed (lm: the fp32 logits of the prefill and of <n> teacher-forced eager steps on fixed pseudo-random
// tokens, then the ids of the free-running graph-replayed steps; dac: the waveform), `cmp` compares two dumps (max |difference|, arg-max
// flips, first differing id): the seconds-long A/B check that a kernel variant behind an environment knob computes what the default
// path computes, before the oracle-backed parity tests are spent on it.
//
// `lm` prints three readings of 250 graph replays each at growing context, like tools/step_probe2.py; `dac` prints ms per decode and
// TFLOP/s at 1.608 GFLOP per frame.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptts.h"

#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (x);                                                                                  \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)
#define PT(x)                                                                                             \
  do {                                                                                                    \
    int r_ = (x);                                                                                         \
    if (r_ != PTTS_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, ptts_last_error()); exit(3); } \
  } while (0)

// value i of stream `seed`: a hash mapped to a uniform variate of standard deviation `std` around `mean`
__global__ void fill_kernel(float* p, size_t n, unsigned seed, float std, float mean) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ (seed * 40503u + 0x9e3779b9u);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  const float u = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;  // [-1, 1)
  p[i] = mean + std * 1.7320508f * u;
}
__global__ void fill_codes_kernel(long long* p, size_t n, unsigned seed, int vocab) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ (seed * 40503u + 0x9e3779b9u);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13;
  p[i] = (long long)(h % (unsigned)vocab);
}

// weights_fp8: one workgroup per row - power-of-two scale from the row maximum, round-to-nearest-even onto the OCP e4m3 grid, the bytes to
// q, the scale to sc, and the row overwritten with its exact dequantisation (what ptts_load_weight must receive, parler_tts_amd/quant.py)
__global__ void __launch_bounds__(256) quant_e4m3_rows_kernel(float* w, unsigned char* q, float* sc, int Kd) {
  __shared__ float red[256];
  float* row = w + (size_t)blockIdx.x * Kd;
  float m = 0.f;
  for (int i = threadIdx.x; i < Kd; i += 256) m = fmaxf(m, fabsf(row[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const float amax = red[0];
  const float scale = amax > 0.f ? exp2f(ceilf(log2f(amax / 448.f))) : 1.f;
  if (threadIdx.x == 0) sc[blockIdx.x] = scale;
  for (int i = threadIdx.x; i < Kd; i += 256) {
    const float x = fminf(fmaxf(row[i] / scale, -448.f), 448.f), ax = fabsf(x);
    int e = ax > 0.f ? (int)floorf(log2f(ax)) : -6;
    e = e < -6 ? -6 : e;                                   // subnormals share the quantum of the smallest normal binade
    float v = rintf(ax / exp2f((float)(e - 3))) * exp2f((float)(e - 3));  // 3 mantissa bits
    v = fminf(v, 448.f);
    unsigned bits = 0;
    if (v > 0.f) {
      int e2 = (int)floorf(log2f(v));
      if (e2 < -6) bits = (unsigned)rintf(v * 512.f);      // subnormal: multiples of 2^-9
      else bits = ((unsigned)(e2 + 7) << 3) | (unsigned)rintf((v / exp2f((float)e2) - 1.f) * 8.f);
    }
    q[(size_t)blockIdx.x * Kd + i] = (unsigned char)(bits | (x < 0.f ? 0x80u : 0u));
    row[i] = (x < 0.f ? -v : v) * scale;
  }
}

struct Filler {
  float* buf = nullptr;
  size_t cap = 0;
  unsigned seed = 1;
  hipStream_t st = nullptr;
  float* get(size_t n, float std, float mean) {
    if (n > cap) {
      if (buf) HIPCHK(hipFree(buf));
      HIPCHK(hipMalloc(&buf, n * sizeof(float)));
      cap = n;
    }
    fill_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(buf, n, seed++, std, mean);
    HIPCHK(hipGetLastError());
    return buf;
  }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#ifdef PTTS_STAMPS
static int report_stamps(ptts_engine* e, hipStream_t st, double step_us, int B);  // measurement build (tools/build_stamps.sh)
#endif

static const char* opt(int argc, char** argv, const char* key) {  // "key=value" -> value, bare "key" -> "", absent -> nullptr
  const size_t n = strlen(key);
  for (int i = 3; i < argc; ++i) {
    if (!strncmp(argv[i], key, n) && argv[i][n] == '=') return argv[i] + n + 1;
    if (!strcmp(argv[i], key)) return "";
  }
  return nullptr;
}

static int run_lm(int argc, char** argv) {
  const int B = atoi(argv[2]);
  const bool large = opt(argc, argv, "large") != nullptr, fp32 = opt(argc, argv, "fp32") != nullptr, fp8 = opt(argc, argv, "fp8") != nullptr;
  const char* dump = opt(argc, argv, "dump");
  const int dump_steps = opt(argc, argv, "steps") ? atoi(opt(argc, argv, "steps")) : 6;
  const char* tag = opt(argc, argv, "tag") ? opt(argc, argv, "tag") : "";
  const int P = opt(argc, argv, "ctx") ? atoi(opt(argc, argv, "ctx")) : 32;
  // layers=<n>: the same widths with n layers (n = 2: every weight stays in the L2s / Infinity Cache between replays, n = 8: 240 MB, Infinity
  // Cache only): per-layer step time against the 24-layer model = what cold weights cost a dependent node
  const int H = large ? 1536 : 1024, L = opt(argc, argv, "layers") ? atoi(opt(argc, argv, "layers")) : (large ? 30 : 24), F = large ? 6144 : 4096,
            NH = large ? 24 : 16, K = 9, V = 1088, NE = 64;
  hipStream_t st;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ptts_config c;
  memset(&c, 0, sizeof c);
  c.hidden_size = H; c.num_layers = L; c.num_heads = NH; c.ffn_dim = F; c.num_codebooks = K; c.vocab_size = V; c.max_positions = 4096;
  c.rope = 0; c.rope_theta = 10000.f; c.pad_token_id = 1024; c.eos_token_id = 1024; c.bos_token_id = 1025;
  c.dtype = fp32 ? PTTS_F32 : PTTS_BF16; c.max_batch = B; c.max_ctx = P + 908; c.max_enc = NE; c.max_prompt = P + 8; c.device = 0;
  c.weights_fp8 = fp8 ? 1 : 0;
  c.kv_fp8 = opt(argc, argv, "kv8") != nullptr ? 1 : 0;  // e4m3 self-attention cache (engines of more than 8 utterances)
  ptts_engine* e = nullptr;
  const double t_create = now_s();
  PT(ptts_engine_create(&c, &e));
  Filler f;
  f.st = st;
  unsigned char* qbuf = nullptr;
  float* scbuf = nullptr;
  auto load = [&](const std::string& name, std::vector<int64_t> shape, float std, float mean) {
    size_t n = 1;
    for (int64_t s : shape) n *= (size_t)s;
    float* p = f.get(n, std, mean);
    const bool q8 = fp8 && shape.size() == 2 && name.find("embed_") == std::string::npos && name.find("encoder_attn.k_proj") == std::string::npos &&
                    name.find("encoder_attn.v_proj") == std::string::npos;  // the matrices the decode step streams (quant.py: is_fp8_matrix)
    if (q8) {
      if (!qbuf) { HIPCHK(hipMalloc(&qbuf, (size_t)F * H)); HIPCHK(hipMalloc(&scbuf, (size_t)(F > V ? F : V) * 4)); }
      quant_e4m3_rows_kernel<<<dim3((unsigned)shape[0]), dim3(256), 0, st>>>(p, qbuf, scbuf, (int)shape[1]);
    }
    PT(ptts_load_weight(e, name.c_str(), p, PTTS_F32, shape.data(), (int32_t)shape.size(), st));
    if (q8) PT(ptts_load_weight_fp8(e, name.c_str(), qbuf, scbuf, shape.data(), 2, st));
  };
  const std::string p = "model.decoder.";
  for (int k = 0; k < K; ++k) load(p + "embed_tokens." + std::to_string(k) + ".weight", {V + 1, H}, 0.02f, 0.f);
  load(p + "embed_positions.weights", {4096, H}, 0.02f, 0.f);
  for (int i = 0; i < L; ++i) {
    const std::string lp = p + "layers." + std::to_string(i) + ".";
    for (const char* att : {"self_attn", "encoder_attn"}) {
      for (const char* pr : {"q_proj", "k_proj", "v_proj", "out_proj"}) load(lp + att + "." + pr + ".weight", {H, H}, 0.02f, 0.f);
      load(lp + att + "_layer_norm.weight", {H}, 0.f, 1.f);
      load(lp + att + "_layer_norm.bias", {H}, 0.f, 0.f);
    }
    load(lp + "fc1.weight", {F, H}, 0.02f, 0.f);
    load(lp + "fc2.weight", {H, F}, 0.02f, 0.f);
    load(lp + "final_layer_norm.weight", {H}, 0.f, 1.f);
    load(lp + "final_layer_norm.bias", {H}, 0.f, 0.f);
  }
  load(p + "layer_norm.weight", {H}, 0.f, 1.f);
  load(p + "layer_norm.bias", {H}, 0.f, 0.f);
  for (int k = 0; k < K; ++k) load("lm_heads." + std::to_string(k) + ".weight", {V, H}, 0.02f, 0.f);
  PT(ptts_weights_ready(e));
  HIPCHK(hipStreamSynchronize(st));
  const double t_loaded = now_s();
  ptts_gen_params gp;
  memset(&gp, 0, sizeof gp);
  gp.max_length = 869; gp.min_new_tokens = 868; gp.do_sample = 0; gp.temperature = 1.f; gp.top_k = 0; gp.top_p = 1.f; gp.use_eos_gate = 1; gp.seed = 0;
  PT(ptts_set_gen_params(e, &gp));
  float *enc = nullptr, *prompt = nullptr;
  HIPCHK(hipMalloc(&enc, (size_t)B * NE * H * 4));
  HIPCHK(hipMalloc(&prompt, (size_t)B * P * H * 4));
  fill_kernel<<<dim3((unsigned)(((size_t)B * NE * H + 255) / 256)), dim3(256), 0, st>>>(enc, (size_t)B * NE * H, 777u, 1.f, 0.f);
  fill_kernel<<<dim3((unsigned)(((size_t)B * P * H + 255) / 256)), dim3(256), 0, st>>>(prompt, (size_t)B * P * H, 778u, 1.f, 0.f);
  HIPCHK(hipStreamSynchronize(st));
  if (opt(argc, argv, "eager")) {
    const int n = atoi(opt(argc, argv, "eager"));
    long long* tok = nullptr;
    HIPCHK(hipMalloc(&tok, (size_t)B * K * 8));
    fill_codes_kernel<<<dim3((unsigned)((B * K + 255) / 256)), dim3(256), 0, st>>>(tok, (size_t)B * K, 1000u, 1024);
    PT(ptts_prefill(e, enc, nullptr, prompt, nullptr, B, NE, P, 0, st));
    for (int s = 0; s < n; ++s) {
      PT(ptts_push_tokens(e, (const int64_t*)tok, nullptr, st));
      PT(ptts_step_forward(e, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    printf("[cabi_probe lm %s] B=%d: %d eager decode forwards after a prefill of %d positions done\n", tag, B, n, P + 1);
    ptts_engine_destroy(e);
    return 0;
  }
  FILE* df = nullptr;
  if (dump && *dump) {  // teacher-forced eager steps: the logits of every forward
    df = fopen(dump, "wb");
    if (!df) { fprintf(stderr, "cannot write %s\n", dump); return 1; }
    const int64_t hdr[4] = {0x70747473 /* 'ptts' */, (int64_t)B * K, V, dump_steps + 1};
    fwrite(hdr, sizeof hdr, 1, df);
    std::vector<float> host((size_t)B * K * V);
    long long* tok = nullptr;
    HIPCHK(hipMalloc(&tok, (size_t)B * K * 8));
    PT(ptts_prefill(e, enc, nullptr, prompt, nullptr, B, NE, P, 0, st));
    for (int s = 0; s <= dump_steps; ++s) {
      if (s > 0) {
        fill_codes_kernel<<<dim3((unsigned)((B * K + 255) / 256)), dim3(256), 0, st>>>(tok, (size_t)B * K, 1000u + s, 1024);
        PT(ptts_push_tokens(e, (const int64_t*)tok, nullptr, st));
        PT(ptts_step_forward(e, st));
      }
      float* lg = nullptr;
      PT(ptts_logits(e, &lg));
      HIPCHK(hipMemcpyAsync(host.data(), lg, host.size() * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      fwrite(host.data(), 4, host.size(), df);
    }
    HIPCHK(hipFree(tok));
  }
  double t0 = now_s();
  PT(ptts_prefill(e, enc, nullptr, prompt, nullptr, B, NE, P, 1, st));
  PT(ptts_first_token_sync(e));
  const double ttft1 = now_s() - t0;
  HIPCHK(hipStreamSynchronize(st));
  t0 = now_s();
  PT(ptts_prefill(e, enc, nullptr, prompt, nullptr, B, NE, P, 1, st));
  PT(ptts_first_token_sync(e));
  const double ttft2 = now_s() - t0;
  PT(ptts_decode_steps(e, 50, st));
  HIPCHK(hipStreamSynchronize(st));
  double us[3];
  hipEvent_t ev0, ev1;
  HIPCHK(hipEventCreate(&ev0));
  HIPCHK(hipEventCreate(&ev1));
  for (int r = 0; r < 3; ++r) {
    HIPCHK(hipEventRecord(ev0, st));
    PT(ptts_decode_steps(e, 250, st));
    HIPCHK(hipEventRecord(ev1, st));
    HIPCHK(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    us[r] = ms * 1e3 / 250;
  }
  int32_t cur = 0, fin = 0;
  PT(ptts_state(e, &cur, &fin, st));
  if (df) {  // the ids of the free-running greedy run (prefill + 800 graph-replayed steps)
    int64_t* ids = nullptr;
    int32_t ld = 0;
    PT(ptts_ids(e, &ids, &ld));
    std::vector<int64_t> host((size_t)B * K * cur);
    HIPCHK(hipMemcpy2D(host.data(), (size_t)cur * 8, ids, (size_t)ld * 8, (size_t)cur * 8, (size_t)B * K, hipMemcpyDeviceToHost));
    const int64_t hdr[2] = {(int64_t)B * K, cur};
    fwrite(hdr, sizeof hdr, 1, df);
    fwrite(host.data(), 8, host.size(), df);
    fclose(df);
  }
  const double wb = ((double)L * (6.0 * H * H + 2.0 * H * F) + (double)K * V * H) * (fp32 ? 4 : (fp8 ? 1 : 2));
  printf("[cabi_probe lm %s%s%s%s] B=%d: %.1f %.1f %.1f us/step  (weights %.0f MB/step -> %.2f TB/s at the middle reading; prefill+first token %.2f ms "
         "(first call, pre-capture) / %.2f ms; create+load %.0f ms; cur_len %d)\n",
         tag, large ? " large" : "", fp32 ? " fp32" : "", fp8 ? " fp8" : "", B, us[0], us[1], us[2], wb / 1e6, wb / 1e6 / us[1], ttft1 * 1e3, ttft2 * 1e3,
         (t_loaded - t_create) * 1e3, cur);
  fflush(stdout);
#ifdef PTTS_STAMPS
  if (B == 1 || B > 8) {  // the single-utterance GEMV step (5 nodes per layer) or the batch > 8 MFMA-path step (7 nodes per layer)
    PT(ptts_decode_steps(e, 4, st));
    HIPCHK(hipStreamSynchronize(st));
    report_stamps(e, st, us[2], B);
  }
#endif
  ptts_engine_destroy(e);
  return 0;
}

#ifdef PTTS_STAMPS
// Measurement build (tools/build_stamps.sh: the library compiled with -DPTTS_TIMING): the wall_clock64 stamps every node of the single-utterance step
// left in its last replay -> per node the phases inside the kernel, per edge the time from the producer's last store to the consumer's entry.
extern "C" int ptts_debug_stamps(ptts_engine* e, long long** stamps_dev, int32_t* layers);
__global__ void stamp_now_kernel(long long* p) { *p = (long long)wall_clock64(); }
static int report_stamps(ptts_engine* e, hipStream_t st, double step_us, int B) {
  const int NN = B > 8 ? 7 : 5;  // nodes per layer
  long long* dev = nullptr;
  int32_t nl = 0;
  PT(ptts_debug_stamps(e, &dev, &nl));
  // tick rate of s_memtime: two stamps 20 ms apart, timed by HIP events on the same stream
  long long* tk = nullptr;
  HIPCHK(hipMalloc(&tk, 16));
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  stamp_now_kernel<<<1, 1, 0, st>>>(tk); HIPCHK(hipEventRecord(a, st)); HIPCHK(hipStreamSynchronize(st));
  const double w0 = now_s();
  while (now_s() - w0 < 0.02) {}
  stamp_now_kernel<<<1, 1, 0, st>>>(tk + 1); HIPCHK(hipEventRecord(b, st)); HIPCHK(hipStreamSynchronize(st));
  long long t2[2];
  HIPCHK(hipMemcpy(t2, tk, 16, hipMemcpyDeviceToHost));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  const double tpu = (double)(t2[1] - t2[0]) / (ms * 1e3);  // ticks per microsecond
  std::vector<long long> h((size_t)(nl + 1) * NN * 48);
  HIPCHK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
  auto S = [&](int l, int k, int slot, int idx) { return h[(((size_t)l * NN + k) * 3 + slot) * 16 + idx]; };
  const char* names5[5] = {"qkv_attn (LN1 + q/k/v rows + self-attention + append)", "combine + out_proj + residual", "xfold_attn (LN2 + M rows + softmax + U columns)",
                           "partial rows + LN3 + fc1 + GELU", "fc2 + residual"};
  const char* names7[7] = {"lnproj_fused (LN1 + q|k|v)", "attn_kernel (self-attention + append)", "gemm_strip (out_proj + residual)",
                           "xattn_fused (LN2 + cross q + cross-attention)", "gemm_strip (cross out_proj + residual)", "lnproj_fused (LN3 + fc1 + GELU)",
                           "gemm_strip (fc2: split-K partials or + residual)"};
  const char** names = NN == 7 ? names7 : names5;
  int khz = 0;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  printf("[node stamps] wall_clock64 runs at %.2f ticks / us (calibrated over %.1f ms; hipDeviceAttributeWallClockRate %d kHz); step = %.1f us by HIP events; layers averaged: 2 .. %d\n",
         tpu, ms, khz, step_us, nl - 2);
  // first / last stamp of a node over the sampled workgroups
  auto first_entry = [&](int l, int k) { long long m = 0; for (int s = 0; s < 3; ++s) for (int i : {0, 2}) { const long long v = S(l, k, s, i); if (v && (!m || v < m)) m = v; } return m; };
  auto last_stamp = [&](int l, int k) { long long m = 0; for (int s = 0; s < 3; ++s) for (int i = 0; i < 16; ++i) m = std::max(m, S(l, k, s, i)); return m; };
  const int l0 = 2, l1 = nl - 2;
  double tot_in = 0, tot_gap = 0;
  for (int k = 0; k < NN; ++k) {
    double in_kernel = 0, gap = 0, ph[16] = {0};
    int cnt[16] = {0}, n = 0;
    for (int l = l0; l <= l1; ++l) {
      const long long e0 = first_entry(l, k), e1 = last_stamp(l, k);
      const int kn = (k + 1) % NN, ln = k == NN - 1 ? l + 1 : l;
      const long long en = first_entry(ln, kn);
      if (!e0 || !e1 || !en) continue;
      in_kernel += (double)(e1 - e0) / tpu; gap += (double)(en - e1) / tpu; ++n;
      for (int i = 1; i < 8; ++i) {  // phases of the FIRST sampled workgroup, relative to its own entry (weight wave: idx 2 when present)
        const long long v = S(l, k, 0, i), base = (NN == 5 && i >= 3 && S(l, k, 0, 2)) ? S(l, k, 0, 2) : S(l, k, 0, 0);
        if (v && base) { ph[i] += (double)(v - base) / tpu; ++cnt[i]; }
      }
    }
    if (NN == 7) {  // slot 14: stamped from a PRELOADED scalar parameter before the argument struct's s_load has returned (strip / lnproj nodes)
      double d = 0; int c14 = 0;
      for (int l = l0; l <= l1; ++l) { const long long v14 = S(l, k, 0, 14), v0 = S(l, k, 0, 0); if (v14 && v0) { d += (double)(v0 - v14) / tpu; ++c14; } }
      if (c14) { ph[15] = d; cnt[15] = c14; }
    }
    if (!n) { printf("[node stamps] node %d: no stamps\n", k); continue; }
    tot_in += in_kernel / n; tot_gap += gap / n;
    printf("[node stamps] node %d %-58s first entry -> last sampled stamp %.2f us | last stamp -> next node's first entry %.2f us | workgroup 0 phases (us from its entry):", k,
           names[k], in_kernel / n, gap / n);
    for (int i = 1; i < 8; ++i) if (cnt[i]) printf(" s%d=%.2f", i, ph[i] / cnt[i]);
    if (cnt[15]) printf(" | first instruction -> argument struct usable (s14 -> s0): %.2f us", ph[15] / cnt[15]);
    printf("\n");
  }
  printf("[node stamps] per layer: in-kernel %.2f us + boundaries %.2f us = %.2f us (x %d layers = %.1f us of the %.1f us step); a boundary = %.2f us on average\n", tot_in,
         tot_gap, tot_in + tot_gap, nl, (tot_in + tot_gap) * nl, step_us, tot_gap / NN);
  if (NN == 5)
    printf("[node stamps] stamp legend - qkv_attn / xfold_attn (wave 0): s1 loads issued, s2 row normalised, s3 projection rows done, s4 attention loop done, s5 combine done; "
           "GEMV nodes: s1 prologue wave done (row prepared), s3 weight wave: loads issued, s4 barrier passed, s5 dot products + reductions done, s6 store issued\n");
  else
    printf("[node stamps] stamp legend (wave 0 of the first workgroup, us from its entry) - lnproj_fused: s1 rows + first weight fragments requested, s2 its rows normalised, "
           "s3 every row of the group in LDS, s4 last weight fragment consumed, s5 cross-wave reduction complete; attn_kernel: s1 scalar state read + first K/V batch "
           "requested, s2 query chunk usable, s3 K/V loop done, s5 output stored; gemm_strip: s1 first weight fragments requested, s3 prologue done, s4 MFMA loop done, "
           "s5 epilogue stored; xattn_fused: s2 rows normalised, s3 rows in LDS, s4 last q-weight fragment consumed, s5 q rows in LDS, s6 attention done, s7 stored\n");
  return 0;
}
#endif

static int run_dac(int argc, char** argv) {
  const int B = atoi(argv[2]);
  const int T = opt(argc, argv, "frames") ? atoi(opt(argc, argv, "frames")) : 860;
  const int reps = opt(argc, argv, "reps") ? atoi(opt(argc, argv, "reps")) : 5;
  const bool f32 = opt(argc, argv, "f32") != nullptr;
  const char* tag = opt(argc, argv, "tag") ? opt(argc, argv, "tag") : "";
  const char* dump = opt(argc, argv, "dump");
  hipStream_t st;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ptts_dac_config c;
  memset(&c, 0, sizeof c);
  c.num_codebooks = 9; c.codebook_size = 1024; c.codebook_dim = 8; c.latent_dim = 1024; c.decoder_dim = 1536; c.num_rates = 4;
  const int rates[4] = {8, 8, 4, 2};
  for (int i = 0; i < 4; ++i) c.rates[i] = rates[i];
  c.compute_dtype = f32 ? PTTS_F32 : PTTS_BF16; c.max_batch = B; c.max_frames = T; c.device = 0; c.encoder_dim = 0;
  ptts_dac* d = nullptr;
  PT(ptts_dac_create(&c, &d));
  Filler f;
  f.st = st;
  auto load = [&](const std::string& name, std::vector<int64_t> shape, float std, float mean) {
    size_t n = 1;
    for (int64_t s : shape) n *= (size_t)s;
    const float* p = f.get(n, std, mean);
    PT(ptts_dac_load_weight(d, name.c_str(), p, shape.data(), (int32_t)shape.size(), st));
    HIPCHK(hipStreamSynchronize(st));  // the scratch buffer is refilled by the next call
  };
  auto conv = [&](const std::string& name, int cout, int cin, int k, bool transposed) {
    const float std = 1.0f / sqrtf((float)cin * (transposed ? 2 : k));
    if (transposed) load(name + ".weight", {cin, cout, k}, std, 0.f);
    else load(name + ".weight", {cout, cin, k}, std, 0.f);
    load(name + ".bias", {cout}, 0.01f, 0.f);
  };
  auto alpha = [&](const std::string& name, int ch) { load(name + ".alpha", {1, ch, 1}, 0.1f, 1.0f); };
  int ch = c.decoder_dim;
  conv("decoder.model.0", ch, c.latent_dim, 7, false);
  for (int bi = 0; bi < 4; ++bi) {
    const int cin = ch >> bi, cout = ch >> (bi + 1), s = rates[bi];
    const std::string b = "decoder.model." + std::to_string(bi + 1) + ".block.";
    alpha(b + "0", cin);
    conv(b + "1", cout, cin, 2 * s, true);
    for (int ri = 0; ri < 3; ++ri) {
      const std::string r = b + std::to_string(ri + 2) + ".block.";
      alpha(r + "0", cout);
      conv(r + "1", cout, cout, 7, false);
      alpha(r + "2", cout);
      conv(r + "3", cout, cout, 1, false);
    }
  }
  alpha("decoder.model.5", ch >> 4);
  load("decoder.model.6.weight", {1, ch >> 4, 7}, 1.0f / sqrtf((float)(ch >> 4) * 7), 0.f);
  load("decoder.model.6.bias", {1}, 0.f, 0.f);
  for (int q = 0; q < 9; ++q) {
    const std::string qn = "quantizer.quantizers." + std::to_string(q) + ".";
    load(qn + "codebook.weight", {1024, 8}, 1.f, 0.f);
    load(qn + "out_proj.weight", {1024, 8, 1}, 0.35f, 0.f);
    load(qn + "out_proj.bias", {1024}, 0.01f, 0.f);
  }
  PT(ptts_dac_weights_ready(d));
  long long* codes = nullptr;
  float* wave = nullptr;
  const size_t ncodes = (size_t)B * 9 * T, nwave = (size_t)B * T * 512;
  HIPCHK(hipMalloc(&codes, ncodes * 8));
  HIPCHK(hipMalloc(&wave, nwave * 4));
  fill_codes_kernel<<<dim3((unsigned)((ncodes + 255) / 256)), dim3(256), 0, st>>>(codes, ncodes, 99u, 1024);
  PT(ptts_dac_decode(d, (const int64_t*)codes, wave, B, T, st));
  HIPCHK(hipStreamSynchronize(st));
  hipEvent_t ev0, ev1;
  HIPCHK(hipEventCreate(&ev0));
  HIPCHK(hipEventCreate(&ev1));
  HIPCHK(hipEventRecord(ev0, st));
  for (int r = 0; r < reps; ++r) PT(ptts_dac_decode(d, (const int64_t*)codes, wave, B, T, st));
  HIPCHK(hipEventRecord(ev1, st));
  HIPCHK(hipEventSynchronize(ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
  ms /= reps;
  std::vector<float> head(4096);
  HIPCHK(hipMemcpy(head.data(), wave, head.size() * 4, hipMemcpyDeviceToHost));
  double ss = 0;
  bool finite = true;
  for (float v : head) { ss += (double)v * v; finite = finite && std::isfinite(v); }
  if (dump && *dump) {  // the whole waveform, in the logits container (rows = utterances)
    FILE* df = fopen(dump, "wb");
    if (!df) { fprintf(stderr, "cannot write %s\n", dump); return 1; }
    std::vector<float> host(nwave);
    HIPCHK(hipMemcpy(host.data(), wave, nwave * 4, hipMemcpyDeviceToHost));
    const int64_t hdr[4] = {0x70747473, B, (int64_t)T * 512, 1};
    fwrite(hdr, sizeof hdr, 1, df);
    fwrite(host.data(), 4, host.size(), df);
    const int64_t none[2] = {0, 0};
    fwrite(none, sizeof none, 1, df);
    fclose(df);
  }
  const double flops = 1.608e9 * (double)B * T;
  printf("[cabi_probe dac %s%s] B=%d frames=%d: %.3f ms per decode = %.1f TFLOP/s (%.1f %% of %s)  [first 4096 samples rms %.3g%s]\n", tag,
         f32 ? " f32" : " bf16", B, T, ms, flops / ms / 1e9, 100.0 * flops / ms / 1e9 / (f32 ? 157.3 : 2500.0), f32 ? "157 TF f32 MFMA" : "2.5 PF bf16 MFMA",
         sqrt(ss / head.size()), finite ? "" : ", NOT FINITE");
  fflush(stdout);
  ptts_dac_destroy(d);
  return 0;
}

struct Dump {
  int64_t rows = 0, cols = 0, blocks = 0, id_rows = 0, id_cols = 0;
  std::vector<float> v;
  std::vector<int64_t> ids;
  bool read(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    int64_t hdr[4];
    bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == 0x70747473;
    if (ok) {
      rows = hdr[1]; cols = hdr[2]; blocks = hdr[3];
      v.resize((size_t)rows * cols * blocks);
      ok = fread(v.data(), 4, v.size(), f) == v.size();
    }
    int64_t ih[2];
    if (ok && fread(ih, sizeof ih, 1, f) == 1) {
      id_rows = ih[0]; id_cols = ih[1];
      ids.resize((size_t)id_rows * id_cols);
      ok = fread(ids.data(), 8, ids.size(), f) == ids.size();
    }
    fclose(f);
    return ok;
  }
};

// 0: identical, 4: same shape but different values (printed), 1: unreadable / different shapes
static int run_cmp(const char* pa, const char* pb) {
  Dump a, b;
  if (!a.read(pa) || !b.read(pb)) { fprintf(stderr, "cannot read %s / %s\n", pa, pb); return 1; }
  if (a.rows != b.rows || a.cols != b.cols || a.blocks != b.blocks) { fprintf(stderr, "different shapes\n"); return 1; }
  double mx = 0, mxabs = 0;
  long long flips = 0, nan = 0;
  int worst_block = -1;
  for (int64_t blk = 0; blk < a.blocks; ++blk)
    for (int64_t r = 0; r < a.rows; ++r) {
      const float* x = &a.v[(size_t)(blk * a.rows + r) * a.cols];
      const float* y = &b.v[(size_t)(blk * a.rows + r) * a.cols];
      int64_t ax = 0, ay = 0;
      for (int64_t c = 0; c < a.cols; ++c) {
        if (!std::isfinite(x[c]) || !std::isfinite(y[c])) { ++nan; continue; }
        const double d = fabs((double)x[c] - (double)y[c]);
        if (d > mx) { mx = d; worst_block = (int)blk; }
        mxabs = fmax(mxabs, fabs((double)x[c]));
        if (x[c] > x[ax]) ax = c;
        if (y[c] > y[ay]) ay = c;
      }
      flips += ax != ay;
    }
  long long first_id = -1;
  if (a.id_rows == b.id_rows && a.id_cols == b.id_cols)
    for (int64_t c = 0; c < a.id_cols && first_id < 0; ++c)
      for (int64_t r = 0; r < a.id_rows; ++r)
        if (a.ids[(size_t)r * a.id_cols + c] != b.ids[(size_t)r * b.id_cols + c]) { first_id = c; break; }
  printf("[cabi_probe cmp] %lld blocks of [%lld, %lld]: max |a - b| = %.3g (block %d; max |a| = %.3g), %lld arg-max flips of %lld rows, %lld non-finite; ids [%lld, %lld]: %s",
         (long long)a.blocks, (long long)a.rows, (long long)a.cols, mx, worst_block, mxabs, flips, (long long)(a.rows * a.blocks), nan, (long long)a.id_rows,
         (long long)a.id_cols, a.id_rows != b.id_rows || a.id_cols != b.id_cols ? "different shapes" : (first_id < 0 ? "identical" : "differ"));
  if (first_id >= 0) printf(" from column %lld", first_id);
  printf("\n");
  return (mx == 0 && nan == 0 && first_id < 0 && a.id_cols == b.id_cols) ? 0 : 4;
}

// T5 description encoder (ABI v7: ptts_t5_*) at flan-t5-large widths, synthetic weights filled on the device, loaded by their transformers names
static int run_t5(int argc, char** argv) {
  const int B = atoi(argv[2]);
  const int N = opt(argc, argv, "tokens") ? atoi(opt(argc, argv, "tokens")) : 64;
  const int L = opt(argc, argv, "layers") ? atoi(opt(argc, argv, "layers")) : 24;
  const int reps = opt(argc, argv, "reps") ? atoi(opt(argc, argv, "reps")) : 50;
  const bool fp32 = opt(argc, argv, "fp32") != nullptr;
  const char* tag = opt(argc, argv, "tag") ? opt(argc, argv, "tag") : "";
  const int V = 32128, D = 1024, F = 2816, NH = 16;
  hipStream_t st;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  ptts_t5_config c;
  memset(&c, 0, sizeof c);
  c.vocab_size = V; c.d_model = D; c.d_kv = 64; c.d_ff = F; c.num_layers = L; c.num_heads = NH; c.rel_buckets = 32; c.rel_max_distance = 128;
  c.layer_norm_eps = 1e-6f; c.dtype = fp32 ? PTTS_F32 : PTTS_BF16; c.max_batch = B; c.max_len = N; c.device = 0;
  ptts_t5* e = nullptr;
  PT(ptts_t5_create(&c, &e));
  Filler f;
  f.st = st;
  auto load = [&](const std::string& name, std::vector<int64_t> shape, float std, float mean) {
    size_t n = 1;
    for (int64_t s : shape) n *= (size_t)s;
    PT(ptts_t5_load_weight(e, name.c_str(), f.get(n, std, mean), PTTS_F32, shape.data(), (int32_t)shape.size(), st));
  };
  load("shared.weight", {V, D}, 1.0f, 0.f);
  load("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", {32, NH}, 0.1f, 0.f);
  load("encoder.final_layer_norm.weight", {D}, 0.f, 1.f);
  for (int l = 0; l < L; ++l) {
    const std::string p = "encoder.block." + std::to_string(l) + ".";
    load(p + "layer.0.SelfAttention.q.weight", {D, D}, 0.004f, 0.f);
    for (const char* m : {"k", "v", "o"}) load(p + "layer.0.SelfAttention." + m + ".weight", {D, D}, 0.03f, 0.f);
    load(p + "layer.0.layer_norm.weight", {D}, 0.f, 1.f);
    load(p + "layer.1.DenseReluDense.wi_0.weight", {F, D}, 0.03f, 0.f);
    load(p + "layer.1.DenseReluDense.wi_1.weight", {F, D}, 0.03f, 0.f);
    load(p + "layer.1.DenseReluDense.wo.weight", {D, F}, 0.02f, 0.f);
    load(p + "layer.1.layer_norm.weight", {D}, 0.f, 1.f);
  }
  PT(ptts_t5_weights_ready(e));
  long long* ids = nullptr;
  float* out = nullptr;
  HIPCHK(hipMalloc(&ids, (size_t)B * N * 8));
  HIPCHK(hipMalloc(&out, (size_t)B * N * D * 4));
  fill_codes_kernel<<<dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, st>>>(ids, (size_t)B * N, 31u, V);
  for (int i = 0; i < 3; ++i) PT(ptts_t5_encode(e, (const int64_t*)ids, nullptr, B, N, out, st));  // first call captures the graph
  HIPCHK(hipStreamSynchronize(st));
  hipEvent_t ev0, ev1;
  HIPCHK(hipEventCreate(&ev0));
  HIPCHK(hipEventCreate(&ev1));
  HIPCHK(hipEventRecord(ev0, st));
  for (int i = 0; i < reps; ++i) PT(ptts_t5_encode(e, (const int64_t*)ids, nullptr, B, N, out, st));
  HIPCHK(hipEventRecord(ev1, st));
  HIPCHK(hipEventSynchronize(ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
  ms /= reps;
  std::vector<float> h((size_t)N * D);
  HIPCHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  double ss = 0;
  for (float v : h) ss += (double)v * v;
  const double flops = 2.0 * B * N * L * (4.0 * D * D + 3.0 * D * F);
  printf("[cabi_probe t5 %s%s] B=%d tokens=%d layers=%d: %.3f ms per encode = %.1f TFLOP/s (projection flops only; weights %.0f MB); rms of the first utterance's "
         "output %.4f\n", tag, fp32 ? " fp32" : "", B, N, L, ms, flops / ms / 1e9, (double)L * (4.0 * D * D + 3.0 * D * F) * (fp32 ? 4 : 2) / 1e6, std::sqrt(ss / h.size()));
  ptts_t5_destroy(e);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "cmp")) return run_cmp(argv[2], argv[3]);
  if (argc < 3 || (strcmp(argv[1], "lm") && strcmp(argv[1], "dac") && strcmp(argv[1], "t5"))) {
    fprintf(stderr, "usage: %s lm <batch> [large] [fp32] [fp8] [kv8] [ctx=<prompt positions>] [tag=<text>] [dump=<file> [steps=<n>]] [eager=<n>]\n"
                    "       %s dac <batch> [frames=<n>] [f32] [reps=<n>] [tag=<text>] [dump=<file>]\n"
                    "       %s t5 <batch> [tokens=<n>] [layers=<n>] [fp32] [reps=<n>] [tag=<text>]\n       %s cmp <file a> <file b>\n", argv[0], argv[0], argv[0], argv[0]);
    return 1;
  }
  if (ptts_abi_version() != PTTS_ABI_VERSION) { fprintf(stderr, "libptts_hip.so has ABI %d, the header %d\n", ptts_abi_version(), PTTS_ABI_VERSION); return 1; }
  if (!strcmp(argv[1], "t5")) return run_t5(argc, argv);
  return !strcmp(argv[1], "lm") ? run_lm(argc, argv) : run_dac(argc, argv);
}
