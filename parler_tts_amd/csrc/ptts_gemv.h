// Row-per-wave GEMV step (ptts_gemv_kernels.h): interface between the engine (ptts_lm.hip) and the translation units that
// instantiate the kernels (ptts_gemv_bf16.hip / ptts_gemv_f32.hip / ptts_gemv_w8.hip: compiled in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "ptts_common.h"

enum { GV_LN = 0, GV_ATTN = 1, GV_COPY = 2, GV_SOFTMAX = 3,  // SOFTMAX: per-head softmax of folded cross-attention scores (ptts_lm_kernels.h: xfold)
       GV_ATTN2 = 4,    // combine of qkv_attn_kernel's partials: S cache splits + the new position's own slot (single utterance)
       GV_LNP = 5 };    // LayerNorm of (x + the per-head partial rows of xfold_attn_kernel), single utterance, one prologue wave per 256 columns;
                        // workgroup 0 also writes the summed row to hsum (the residual operand of the fc2 node)
enum { GV_STORE = 0, GV_RESID = 1, GV_GELU_WT = 2 };
constexpr int GV_PMAX = 24;  // partial rows (= attention heads) the GV_LNP instances are built for
enum { GV_F32 = 0, GV_BF16 = 1, GV_BF16_W8 = 2 };  // engine dtype of activations / weights; W8 = OCP e4m3 weights, bf16 activations

// ---- measurement build only (-DPTTS_TIMING: tools/build_stamps.sh -> tools/stamps/, never the product library) --------------------------------
// Every node of the single-utterance step stamps the device-wide constant-rate counter (wall_clock64 = s_memrealtime, 100 MHz: 10 ns) at its phases
// into dbg[3 sampled workgroups][16] (first / middle / last workgroup of the launch; lane 0 of the wave named at the call site). s_memtime is NOT
// usable here: it counts shader cycles PER XCD with unrelated offsets (first attempt, call 4: negative intervals between workgroups of different
// XCDs). One global counter: stamps of different kernels of the same replay compare directly (entry-to-entry = kernel boundary + the phases in
// between). VERDICT r04 item 5; report: profiles/r05_node_stamps.txt.
#ifdef PTTS_TIMING
#define GV_DBG_FIELDS long long* dbg;
#define GV_STAMP(a, idx)                                                                                                   \
  do {                                                                                                                     \
    if ((a).dbg && (threadIdx.x & 63) == 0) {                                                                              \
      const unsigned nb_ = gridDim.x * gridDim.y * gridDim.z, lb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
      const int slot_ = lb_ == 0 ? 0 : (lb_ == nb_ / 2 ? 1 : (lb_ == nb_ - 1 ? 2 : -1));                                   \
      if (slot_ >= 0) (a).dbg[slot_ * 16 + (idx)] = (long long)wall_clock64();                                          \
    }                                                                                                                      \
  } while (0)
#else
#define GV_DBG_FIELDS
#define GV_STAMP(a, idx) do { } while (0)
#endif

constexpr int GV_MAX_ROWS = 8;  // utterances one GEMV launch serves (instances for 1, 2..4 and 5..8); above that the MFMA strip kernels take over

// Field order: the first 56 bytes are what a wave needs to ADDRESS its first loads - they arrive preloaded in SGPRs (ptts_common.h: kernel-argument
// preload); everything else comes by one s_load that overlaps those loads.
struct GemvArgs {
  // ---- bytes 0..55: preloaded ----
  const void* W;        // row-major [N][K]: engine dtype, or e4m3 bytes (W8)
  const float* x;       // GV_LN / GV_LNP: residual-stream rows, fp32 [M][x_ld]; GV_SOFTMAX: scores fp32 [heads][NE]
  const void* xw;       // GV_COPY: activation rows in the engine dtype [M][xw_ld]
  const float* resid;   // GV_RESID: the residual operand, fp32 [M][out_ld] (null = out: in-place accumulate)
  const float* part;    // GV_ATTN*: split-KV partials [M][S][K] (unnormalised); GV_LNP: the per-head partial rows [npart][K] of xfold_attn_kernel,
                        // added to x in row order before the LayerNorm
  const float* stats;   // GV_ATTN*: (max, sumexp) per head and split [M][S][nheads][2]
  int N;
  unsigned char M;      // utterances (<= GV_MAX_ROWS)
  unsigned char npart;  // GV_LNP: partial rows (<= GV_PMAX)
  unsigned short nheads;
  // ---- tail ----
  const float* wscale;  // W8: per-row power-of-two scale [N]
  const float* gamma;   // GV_LN / GV_LNP
  const float* beta;
  const int* mask;      // GV_SOFTMAX: description padding mask int32 [>= NE] (1 = keep) or null
  const int* n_valid;   // GV_SOFTMAX: device-resident description length N (positions >= N carry no key)
  float* out;           // GV_STORE / GV_RESID: fp32 [M][out_ld]; GV_GELU_WT: engine dtype [M][out_ld]
  float* hsum;          // GV_LNP: x + the partial rows, fp32 [K] (written by workgroup 0)
  int ne;               // GV_SOFTMAX: positions per head in the folded layout (32 or 64)
  int x_ld, xw_ld, out_ld;
  int K;                // = the instance's NCH * 64 * EPL (the kernels use the constant)
  float invK;
  GV_DBG_FIELDS
};
static_assert(sizeof(GemvArgs) % 8 == 0 && offsetof(GemvArgs, wscale) == 56, "GemvArgs: 56 preloaded bytes + tail");
#define GemvArgs_KPARAMS                                                                                                                      \
  const void *kW_, const float *kx_, const void *kxw_, const float *kresid_, const float *kpart_, const float *kstats_, int kN_, unsigned kmpn_, \
      KTail<GemvArgs> kt_
#define GemvArgs_KJOIN(a)                                                                                                          \
  GemvArgs a;                                                                                                                      \
  PTTS_KTAIL_JOIN(GemvArgs, a);                                                                                                    \
  a.W = kW_; a.x = kx_; a.xw = kxw_; a.resid = kresid_; a.part = kpart_; a.stats = kstats_; a.N = kN_;                             \
  a.M = (unsigned char)(kmpn_ & 0xffu); a.npart = (unsigned char)((kmpn_ >> 8) & 0xffu); a.nheads = (unsigned short)(kmpn_ >> 16);
template <typename K> inline void ptts_klaunch(K kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const GemvArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, shmem, st, a.W, a.x, a.xw, a.resid, a.part, a.stats, a.N,
                     (unsigned)a.M | ((unsigned)a.npart << 8) | ((unsigned)a.nheads << 16), ptts_ktail(a));
}

// Single-utterance fused node: LayerNorm + this head's q / k / v rows + split-KV self-attention + KV append (qkv_attn_kernel,
// ptts_gemv_kernels.h). Replaces the LN1+QKV node and the attention node of the GEMV step by ONE launch of nheads x (S + 3) workgroups of 8 waves.
struct QkvAttnArgs {
  // ---- bytes 0..55: preloaded (ptts_common.h) - what addresses the residual row, gamma / beta and the wave's weight rows ----
  const void* W;        // fused QKV projection, row-major [H + 2 * kv_heads * 64][H]: engine dtype or e4m3 bytes (W8)
  const float* x;       // residual-stream row of the utterance, fp32 [H]
  const float* gamma;   // self_attn_layer_norm
  const float* beta;
  int S, nheads, H, kv_heads;
  int x_ld, M;          // row pitch of x and utterances (grid.z); part / stats hold M x (S + 1) slots, the caches M x kv_heads heads
  // ---- tail ----
  const float* wscale;  // W8: per-row scale
  void* kcache;         // this layer's self K / V of utterance 0: [kv_heads][cap][64] engine dtype
  void* vcache;
  const int* cur_len;   // device-resident column count (position of the new token = *P + cur_len[0] - 1)
  const int* P;         // device-resident prompt length
  const int* mask;      // prompt padding mask int32 [mask_ld] (1 = keep) or null; applies to positions < *P
  float* part;          // [S + 1][H] unnormalised partial outputs (slot S: the new position's V row as stored in the cache)
  float* stats;         // [S + 1][nheads][2]: (max, sumexp) in log2 units per cache split; slot S: the two half dot products of q . k_new
  int cap, kv_bound, mask_ld;
  float scale, invK;
  GV_DBG_FIELDS
};
static_assert(sizeof(QkvAttnArgs) % 8 == 0 && offsetof(QkvAttnArgs, wscale) == 56, "QkvAttnArgs: 56 preloaded bytes + tail");
#define QkvAttnArgs_KPARAMS                                                                                                              \
  const void *kW_, const float *kx_, const float *kgamma_, const float *kbeta_, int kS_, int knheads_, int kH_, int kkvh_, int kxld_, int kM_, \
      KTail<QkvAttnArgs> kt_
#define QkvAttnArgs_KJOIN(a)                                                                                                       \
  QkvAttnArgs a;                                                                                                                   \
  PTTS_KTAIL_JOIN(QkvAttnArgs, a);                                                                                                 \
  a.W = kW_; a.x = kx_; a.gamma = kgamma_; a.beta = kbeta_; a.S = kS_; a.nheads = knheads_; a.H = kH_; a.kv_heads = kkvh_; a.x_ld = kxld_; a.M = kM_;
template <typename K> inline void ptts_klaunch(K kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const QkvAttnArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, shmem, st, a.W, a.x, a.gamma, a.beta, a.S, a.nheads, a.H, a.kv_heads, a.x_ld, a.M, ptts_ktail(a));
}
// Single-utterance fused cross block over the folded matrices (xfold_attn_kernel): LayerNorm + the head's 64 score rows of M + per-head
// softmax + the head's columns of U -> one partial output row per head; consumer: GV_LNP (LN3 + fc1), which also publishes the summed row.
struct XfoldAttnArgs {
  // ---- bytes 0..55: preloaded (ptts_common.h) ----
  const void* Mw;       // folded scores matrix [nheads * 64][H], engine dtype (base-2 scale folded in)
  const void* Uw;       // folded output matrix [H][nheads * 64], engine dtype
  const float* x;       // residual-stream row, fp32 [H]
  const float* gamma;   // encoder_attn_layer_norm
  const float* beta;
  const int* n_valid;   // device-resident description length (a dependent scalar load: its pointer must not wait for the tail)
  int nheads, H;
  // ---- tail ----
  const int* mask;      // description padding mask int32 [>= 64] (1 = keep) or null
  float* xpart;         // [nheads][H]
  int nur;              // rounds of output rows per workgroup (2 or 4): grid = nheads x H / (nur * rows per round)
  float invK;
  GV_DBG_FIELDS
};
static_assert(sizeof(XfoldAttnArgs) % 8 == 0 && offsetof(XfoldAttnArgs, mask) == 56, "XfoldAttnArgs: 56 preloaded bytes + tail");
#define XfoldAttnArgs_KPARAMS                                                                                                    \
  const void *kMw_, const void *kUw_, const float *kx_, const float *kgamma_, const float *kbeta_, const int *knvalid_, int knheads_, int kH_, \
      KTail<XfoldAttnArgs> kt_
#define XfoldAttnArgs_KJOIN(a)                                                                                                   \
  XfoldAttnArgs a;                                                                                                               \
  PTTS_KTAIL_JOIN(XfoldAttnArgs, a);                                                                                             \
  a.Mw = kMw_; a.Uw = kUw_; a.x = kx_; a.gamma = kgamma_; a.beta = kbeta_; a.n_valid = knvalid_; a.nheads = knheads_; a.H = kH_;
template <typename K> inline void ptts_klaunch(K kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const XfoldAttnArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, shmem, st, a.Mw, a.Uw, a.x, a.gamma, a.beta, a.n_valid, a.nheads, a.H, ptts_ktail(a));
}
// 0 on success, -1: no instance for this width / mode, -2: launch error (mode: GV_F32 / GV_BF16 - the folded matrices are never e4m3)
int ptts_xfoldattn_launch(int mode, XfoldAttnArgs a, hipStream_t st);
bool ptts_xfoldattn_ok(int H, int nheads, int mode);

// Cross block, first half, 1..8 utterances without the static fold (xq_attn_kernel): encoder_attn_layer_norm + the head's 64 cross-q rows +
// cross-attention against the utterance's description K / V as ONE launch of nheads x M workgroups (instead of the LN2 + q GEMV node and the
// attention node); output = the normalised context in the engine dtype, read by the out_proj node (GV_COPY). Sinusoidal positions only
// (RoPE rotates the cross query).
struct XqAttnArgs {
  const void* W;        // cross q projection, row-major [H][H]: engine dtype or e4m3 bytes (W8)
  const float* wscale;  // W8: per-row scale
  const float* x;       // residual rows fp32 [M][x_ld]
  const float* gamma;   // encoder_attn_layer_norm
  const float* beta;
  const void* kcache;   // this layer's cross K / V: [M][kv_heads][cap][64] engine dtype
  const void* vcache;
  const int* mask;      // description padding mask int32 [M][mask_ld] (1 = keep) or null
  const int* n_valid;   // device-resident description length of the call
  void* out;            // engine dtype [M][out_ld]
  int x_ld, out_ld, mask_ld, cap;
  int nheads, H, kv_heads, M;
  float scale, invK;
};
int ptts_xqattn_launch(int mode, XqAttnArgs a, hipStream_t st);
bool ptts_xqattn_ok(int H, int mode);

// 0 on success, -1: no instance for this width / mode, -2: launch error
int ptts_qkvattn_launch(int mode, QkvAttnArgs a, hipStream_t st);
bool ptts_qkvattn_ok(int H, int mode);
int ptts_qkvattn_rows_per_split(int mode);  // cache positions the first K/V batch of one split covers (a second batch is a dependent round trip)

// 0 on success, -1: no instance for this shape (the caller falls back / refuses at create time), -2: launch error
int ptts_gemv_launch(int mode, int pro, int epi, int S, GemvArgs a, hipStream_t st);
// shapes the GEMV step is instantiated for: K * sizeof(elem) a multiple of 1 KiB with a supported chunk count
bool ptts_gemv_k_ok(int K, int mode);
