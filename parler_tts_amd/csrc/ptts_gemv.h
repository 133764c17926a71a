// Row-per-wave GEMV step (ptts_gemv_kernels.h): interface between the engine (ptts_lm.hip) and the translation units that
// instantiate the kernels (ptts_gemv_bf16.hip / ptts_gemv_f32.hip / ptts_gemv_w8.hip: compiled in parallel).
#pragma once
#include <hip/hip_runtime.h>

enum { GV_LN = 0, GV_ATTN = 1, GV_COPY = 2, GV_SOFTMAX = 3 };  // SOFTMAX: per-head softmax of folded cross-attention scores (ptts_lm_kernels.h: xfold)
enum { GV_STORE = 0, GV_RESID = 1, GV_GELU_WT = 2 };
enum { GV_F32 = 0, GV_BF16 = 1, GV_BF16_W8 = 2 };  // engine dtype of activations / weights; W8 = OCP e4m3 weights, bf16 activations

constexpr int GV_MAX_ROWS = 8;  // utterances one GEMV launch serves (instances for 1, 2..4 and 5..8); above that the MFMA strip kernels take over

struct GemvArgs {
  const void* W;        // row-major [N][K]: engine dtype, or e4m3 bytes (W8)
  const float* wscale;  // W8: per-row power-of-two scale [N]
  const float* x;       // GV_LN: residual-stream rows, fp32 [M][x_ld]
  const void* xw;       // GV_COPY: activation rows in the engine dtype [M][xw_ld]
  const float* gamma;   // GV_LN
  const float* beta;
  const float* part;    // GV_ATTN: split-KV partials [M][S][K] (unnormalised) ...
  const float* stats;   // ... and their (max, sumexp) per head [M][S][nheads][2]
  const int* mask;      // GV_SOFTMAX: description padding mask int32 [>= NE] (1 = keep) or null; x = scores fp32 [heads][NE]
  const int* n_valid;   // GV_SOFTMAX: device-resident description length N (positions >= N carry no key)
  int ne;               // GV_SOFTMAX: positions per head in the folded layout (32 or 64)
  float* out;           // GV_STORE / GV_RESID: fp32 [M][out_ld]; GV_GELU_WT: engine dtype [M][out_ld]
  int x_ld, xw_ld, out_ld;
  int M, N, K, nheads;
  float invK;
};

// 0 on success, -1: no instance for this shape (the caller falls back / refuses at create time), -2: launch error
int ptts_gemv_launch(int mode, int pro, int epi, int S, GemvArgs a, hipStream_t st);
// shapes the GEMV step is instantiated for: K * sizeof(elem) a multiple of 1 KiB with a supported chunk count
bool ptts_gemv_k_ok(int K, int mode);
