// Decoder-LM engine behind the C ABI of include/ptts.h: packed weights, static KV arena, device-resident
// sampler state, one captured hipGraph per batch size. Replaces ParlerTTSForCausalLM.forward + the
// transformers `_sample` loop of the reference (modeling_parler_tts.py:1865, :3564).
#include <map>
#include <set>
#include <string>
#include <vector>
#include <string.h>

#include "ptts_common.h"
#include "ptts_lm_kernels.h"
#include "ptts_gemv.h"
#include "ptts_strip_w8.h"
#include "ptts_gemm_launch.h"

thread_local std::string g_ptts_err;
int ptts_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_ptts_err = buf;
  return code;
}
extern "C" const char* ptts_last_error(void) { return g_ptts_err.c_str(); }
extern "C" int ptts_abi_version(void) { return PTTS_ABI_VERSION; }

namespace {

struct LayerW {
  void *qkv = nullptr, *o = nullptr, *cq = nullptr, *ckv = nullptr, *co = nullptr, *fc1 = nullptr, *fc2 = nullptr;
  float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr, *ln3_g = nullptr, *ln3_b = nullptr;
  void *k_self = nullptr, *v_self = nullptr, *k_cross = nullptr, *v_cross = nullptr;
  float *ks_self = nullptr, *vs_self = nullptr;  // kv_fp8 engines: one power-of-two scale per (utterance, K/V head, position) of the e4m3 self cache
  // row-major [N][K] copies in the engine dtype for the single-utterance GEMV step (ptts_gemv_kernels.h); null = path off
  void *qkv_rm = nullptr, *o_rm = nullptr, *cq_rm = nullptr, *co_rm = nullptr, *fc1_rm = nullptr, *fc2_rm = nullptr;
  // weights_fp8: the row-major copies hold OCP e4m3 bytes, one power-of-two scale per output row
  float *qkv_sc = nullptr, *o_sc = nullptr, *cq_sc = nullptr, *co_sc = nullptr, *fc1_sc = nullptr, *fc2_sc = nullptr;
  // weights_fp8, engines that decode on the MFMA strips (max_batch > GV_MAX_ROWS): e4m3 strips [N/16][K/64][64][16 B] (same scales).
  // The cross q projection stays bf16: it lives inside the fused cross-block kernel.
  void *qkv_p8 = nullptr, *o_p8 = nullptr, *co_p8 = nullptr, *fc1_p8 = nullptr, *fc2_p8 = nullptr;
  void *xM = nullptr, *xU = nullptr;  // folded cross-attention of the current single utterance: M [heads*NE][H], U [H][heads*NE] (engine dtype)
};

}  // namespace

struct ptts_engine {
  ptts_config cfg;
  int max_prompt = 0;  // P + 1 rows capacity per utterance
  size_t esize = 4;
  hipStream_t own_stream = nullptr;
  hipStream_t fold_stream = nullptr;   // cross-attention folding runs here, beside the rest of the prefill
  hipEvent_t ev_kv = nullptr, ev_fold = nullptr;  // cross K/V cache written / folded matrices ready
  hipEvent_t ev_first = nullptr;                  // first token of the last prefill materialised (recorded right after the sampler tail)
  hipEvent_t ev_tail0 = nullptr;                  // recorded right before that sampler tail (ptts_first_token_times)
  hipEvent_t ev_pre0 = nullptr;                   // recorded when the stream reaches the last ptts_prefill's first piece of work
  bool first_recorded = false;
  std::vector<void*> allocs;
  std::vector<LayerW> L;
  void* embed = nullptr;       // [K][V+1][H]
  float* pos_table = nullptr;  // [max_pos][H]
  float *rope_cos = nullptr, *rope_sin = nullptr;
  float *lnf_g = nullptr, *lnf_b = nullptr;
  void* heads = nullptr;  // [K*V][H] packed
  void* heads_rm = nullptr;  // [K*V][H] row-major (GEMV step)
  void* heads_p8 = nullptr;  // [K*V][H] e4m3 strips (weights_fp8, MFMA decode)
  bool w8_strips = false;    // weights_fp8 engine whose decode step runs on the MFMA strips: e4m3 strip copies are allocated
  float* heads_sc = nullptr;
  bool use_gemv = false;     // decode step at batch <= gemv_rows on the row-per-wave GEMV kernels
  int gemv_rows = 1;         // 1 (fp32 parity engine) or GV_MAX_ROWS
  int xfold_ne = 0;          // > 0: static cross-attention folding available (positions per head in the folded layout)
  KvLayer* kv_layers = nullptr;      // [layers] operands of the batched cross K/V projection (device memory, written once at create)
  FoldLayer* fold_layers = nullptr;  // [layers] operand pointers of the fold kernels (device memory, written once at create)
  bool xfold_valid = false;  // the folded matrices of the CURRENT call are in place (single utterance)
  bool w8 = false;           // cfg.weights_fp8: e4m3 row-major weights for the GEMV step (the MFMA paths use the exact bf16 dequantisation)
  std::set<std::string> loaded_fp8, required_fp8;
  std::set<std::string> loaded, required;
  // scratch
  float *h = nullptr, *qkv = nullptr, *qc = nullptr, *part = nullptr, *stats = nullptr, *ffn = nullptr, *logits = nullptr;
  float* sort_buf = nullptr;
  void *xw = nullptr, *xw2 = nullptr;  // engine-dtype activation rows for the M > 8 path: [rows][H], [rows][F]
  int nkv = 0, nkc = 0;                // self / cross K/V heads (== num_heads unless grouped-query attention)
  long long* prefix = nullptr;         // voice-prompt codes [max_batch*K][max_ctx], valid for the next prefill when pending_T > 0
  int pending_T = 0;
  int prefill_T = 0;                   // voice-prompt columns folded into the current prefill pass (batched multi-column prefill)
  float* lnstat = nullptr;             // strip statistics of the residual rows (EPI_RESID -> PRO_LNS), [max_batch][H/16][2]
  bool use_lns = true;                 // 8 < batch <= 32 decode: LayerNorm fused into the consumer GEMM (no rows_prep node)
  bool use_fo = true;                  // batch > 8 decode: engine-dtype activations in MFMA B-fragment order (PTTS_NO_FO=1: row-major, for A/B)
  int S_self = 4, S_cross = 1;
  int attn_waves = 4;  // waves per self-attention workgroup at decode
  int cross_waves = 4; // waves per cross-attention workgroup on the GEMV step: one 8-deep batch of row groups per wave covers max_enc
  // state
  long long* ids = nullptr;
  int ids_ld = 0;
  int *cur_len = nullptr, *unfinished = nullptr, *has_eos = nullptr, *first_unf = nullptr;
  int *enc_mask = nullptr, *prompt_mask = nullptr;
  DevDims* dims = nullptr;
  DevGen* gen = nullptr;
  ptts_gen_params gp;
  // per-call
  int B = 0, N = 0, P = 0;
  bool prefilled = false;
  bool h_ready = false;  // residual-stream input of the next decode step already embedded by the last tail
  int xattn_groups_max = 256; // largest batch that runs the fused cross block in groups of 8 (PTTS_XATTN_GROUPS_MAX; above: rows_prep + q GEMM + attention)
  int lnproj = -1;            // -1 = on (3) at every batch size above 8: 8 utterances per workgroup up to 40 (round 4: -5.3 % at 32, -3.2 % at 12; neutral at 64, +12 % at 128
                              // with 8 per workgroup), 16 per workgroup above (round 5, profiles/r05_experiments.txt call 2: 48 / 64 / 96 / 128 utterances -9.7 / -6.6 / -2.3 / -5.1 % per step). Decode at batch > 8: LayerNorm + projection as ONE node tiled over 64 weight rows x lnproj_g utterances instead of rows_prep + strip GEMM
                              // (PTTS_LNPROJ: 0 off, 1 = LN1 + QKV, 2 = + LN3 + fc1 above 32 utterances, 3 = + LN3 + fc1 at 9..32 too instead of the producer-statistics prologue)
  int fuse_qa_max = 3;        // largest batch that runs it (PTTS_FUSE_QA_MAX = 1..8)
  int last_graph_nodes = 0;   // kernel nodes of the step graph captured last (ptts_debug_graph_nodes)
  bool fuse_qa = true;        // single-utterance GEMV step: LN1 + QKV rows + self-attention + append as one node (qkv_attn_kernel), PTTS_NO_FUSE_QA=1 = two nodes
  int fuse_x = -1;            // single-utterance GEMV step, folded cross block: LN2 + scores + softmax + U p as one node of per-head partial rows (xfold_attn_kernel),
                              // summed by the LN3 + fc1 node's prologue (GV_LNP). -1 = by width: on up to hidden 1024 (Mini-v1 -1 %: 566 -> 561 us per step), off
                              // above (Large-v1 +8..17 %: every fc1 workgroup pulls 24 x 6 KB of partial rows through the L2; profiles/r04_experiments.txt
                              // call 16); PTTS_FUSE_X=0 / 1 forces it
  int fuse_x_nur = 2;         // rounds of output rows per workgroup of that node (PTTS_FUSE_X_NUR = 2 / 4: nheads x 8 / nheads x 4 workgroups at Mini-v1)
  float* xpart = nullptr;     // [nheads][H] per-head partial rows of the fused cross block
  float* h2 = nullptr;        // [H] residual row after the cross block (x + partial rows), written by the LN3 + fc1 node
  bool fuse_xq = true;        // GEMV step, un-folded cross block: LN2 + cross-q rows + cross-attention as one node (xq_attn_kernel), PTTS_NO_FUSE_XQ=1 = two nodes
  int fuse_qa_s = 0;          // KV splits of that node: 0 = by context bucket (1 / 2 / 4 / 8 for <= 256 / 512 / 1024 / more positions), PTTS_FUSE_QA_S forces one
  int lnproj_g = 0;           // utterances per workgroup of that node: 0 = by batch size (8 up to 40 utterances, 16 above), PTTS_LNPROJ_G = 4 / 8 / 16 forces one
  int xattn_g = 0;            // utterances per workgroup of the fused cross block above 8 utterances: 0 = by batch size (2 up to 32, 4 up to 64, 8 above), PTTS_XATTN_G = 8 / 4 / 2 forces one
  bool xattn_g_ok = false;    // the g < 8 instances exist for this width (Mini-v1, Large-v1)
  bool xattn_groups = true;   // the fused LN2 + cross-q + cross-attention kernel also at batch 9..32, in groups of 8 utterances (PTTS_NO_XATTN_GROUPS=1: two nodes)
  int kv_ub = 0;         // host-side upper bound of the self-KV positions written so far (prefill + one per decode forward)
  int kv_bound = 0;      // attention fetch bound of the next decode forward: kv_ub + 1 rounded up to 64, <= max_ctx
  std::map<long long, hipGraphExec_t> graphs;  // key: 2 * batch size + folded-cross-block flag + context bucket + steps per launch (get_graph)
  std::map<long long, hipGraphExec_t> prefill_graphs;  // PTTS_PREFILL_GRAPH: the prefill forward of one (batch, description, prompt, voice-prompt) shape
  bool in_capture = false;
  int* host_pinned = nullptr;
#ifdef PTTS_TIMING
  long long* dbg_stamps = nullptr;  // measurement build: [layers + 1][5 or 7 nodes][3 workgroups][16] wall-clock stamps of the decode step (ptts_debug_stamps)
#endif

  template <typename T> int alloc(T** p, size_t n) {
    void* v = nullptr;
    hipError_t e = hipMalloc(&v, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    allocs.push_back(v);
    *p = reinterpret_cast<T*>(v);
    return PTTS_OK;
  }
  int alloc_bytes(void** p, size_t bytes) {
    char* c = nullptr;
    PTTS_TRY(alloc(&c, bytes));
    *p = c;
    return PTTS_OK;
  }
};

namespace {


// LayerNorm (+ fold of pending fc2 partials) + projection, tiled over 64 weight rows x G utterances (lnproj_fused_kernel): decode at batch > 8
template <typename WT, int EPI>
int launch_lnproj(ptts_engine* e, LnProjArgs p, hipStream_t st, int g) {
  constexpr int KT = Elem<WT>::KT;
  const int H = p.K;
  p.invK = 1.0f / (float)H;
  const int mg = p.M < g ? p.M : g;
  const size_t sh = (size_t)mg * (H * sizeof(WT) + 16) + 8 * 1024;
  const dim3 grid(p.N / 64, (p.M + g - 1) / g);
  const bool u16 = ((H / KT) / 2) % 16 == 0;
  // > 64 KiB of dynamic LDS needs an explicit opt-in, once per instantiation and device (the fp32 engine's 16-row instances ask for 72 KiB at
  // H = 1024 and 104 KiB at H = 1536): same pattern as launch_gemm_inst
#define PTTS_LNPROJ_ONE(UW, NF4, G)                                                                                   \
  do {                                                                                                                \
    static PttsPerDeviceOnce attr_once;                                                                               \
    const int attr_dev = PttsPerDeviceOnce::device();                                                                 \
    if (sh > 64 * 1024 && attr_once.need(attr_dev)) {                                                                 \
      hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&lnproj_fused_kernel<WT, UW, NF4, G, EPI>),  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
      if (ea != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(ea)); \
      attr_once.done(attr_dev);                                                                                       \
    }                                                                                                                 \
    ptts_klaunch(lnproj_fused_kernel<WT, UW, NF4, G, EPI>, grid, dim3(512), sh, st, p);                      \
  } while (0)
#define PTTS_LNPROJ_LAUNCH(UW, NF4)                                                                                   \
  do {                                                                                                                \
    if (g == 16) PTTS_LNPROJ_ONE(UW, NF4, 16);                                                                        \
    else if (g == 4) PTTS_LNPROJ_ONE(UW, NF4, 4);                                                                     \
    else PTTS_LNPROJ_ONE(UW, NF4, 8);                                                                                 \
  } while (0)
  if (H == 1024 && u16) PTTS_LNPROJ_LAUNCH(16, 4);
  else if (H == 1024) PTTS_LNPROJ_LAUNCH(8, 4);
  else PTTS_LNPROJ_LAUNCH(8, 6);
#undef PTTS_LNPROJ_LAUNCH
#undef PTTS_LNPROJ_ONE
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "lnproj launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}

// LN -> GEMM and split-KV-combine -> GEMM: fused prologue at M <= 8 rows, prep kernel + copy staging above (the
// redundant per-workgroup prologue is 88 % of the GEMM at M = 32: tools/phase_probe, profiles/).
template <typename WT, int PRO, int EPI>
int gemm_with_prologue(ptts_engine* e, GemmArgs g, hipStream_t st) {
  if (g.M <= 8) return launch_gemm<WT, PRO, EPI>(g, st);
  if constexpr (PRO == PRO_LN) {
    // statistics came with the residual rows and all M normalised rows fit in LDS beside the reduction buffer
    if (g.lnstat && g.M <= 32 && (size_t)g.M * ((size_t)g.K * sizeof(WT) + 16) + 16 * 1024 <= 159 * 1024 && (g.K == 1024 || g.K == 1536))
      return launch_gemm<WT, PRO_LNS, EPI>(g, st);
  }
  GemmArgs gp = g;
  gp.out_fo = g.x_fo;  // the prep kernel writes the rows in the order the consumer reads them (x_fo set by the caller: decode, batch > 8)
  PTTS_TRY((launch_prep<WT, PRO>(gp, e->xw, st)));
  g.x = reinterpret_cast<const float*>(e->xw);
  g.x_ld = g.K; g.x_row_mul = 1; g.x_row_off = 0;
  return launch_gemm<WT, PRO_COPY, EPI>(g, st);
}

// One decoder forward over Q positions per utterance (Q = P+1 at prefill, 1 at decode) up to the logits.
template <typename WT>
int forward(ptts_engine* e, bool prefill, hipStream_t st, bool with_embed = true) {
  const ptts_config& c = e->cfg;
  const int H = c.hidden_size, F = c.ffn_dim, nh = c.num_heads, B = e->B;
  const int nkv = e->nkv, nkc = e->nkc, Hkv = nkv * 64, QKV = H + 2 * Hkv;  // grouped-query attention: fewer K/V heads
  const int Q = prefill ? e->P + 1 + e->prefill_T : 1;  // prompt positions + BOS column [+ the voice-prompt columns, run in the same pass]
  const int M = B * Q;
  const bool big = M > 8;
  const int dec = prefill ? 0 : 1;  // GemmArgs::decode: rows per M pass of the strip GEMMs (msplit_rows)
  const float scale = 1.0f / sqrtf((float)(H / nh));

  if (prefill) {  // cross-attention K/V of the description, once per call (:877-878 then reused :872-875)
    // encoder states were staged into qc by ptts_prefill (fp32 row-major [B*N][H]); convert once to the engine dtype
    const size_t n = (size_t)B * e->N * H;
    hipLaunchKernelGGL((convert_kernel<WT, float>), dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, e->qc,
                       reinterpret_cast<WT*>(e->xw2), n);
    static const bool kv_batched = !(ptts_dev_env("PTTS_NO_KV_BATCHED") && atoi(ptts_dev_env("PTTS_NO_KV_BATCHED")));
    // every layer's K/V projection in ONE launch (blockIdx.z = layer): the strip kernel up to 256 rows, the bf16 engine's LDS-DMA GEMM above (round 6:
    // 24 launches of ~17 us at 32 descriptions x 64 tokens were one tenth of that prefill; a shape the GEMM declines falls back to the strips)
    const bool one_launch = kv_batched && (B * e->N <= 256 || (sizeof(WT) == 2 && H % 64 == 0 && (2 * nkc * 64) % 64 == 0));
    for (int l = 0; l < (one_launch ? 1 : c.num_layers); ++l) {
      GemmArgs g = {};
      g.W = e->L[l].ckv; g.M = B * e->N; g.N = 2 * nkc * 64; g.K = H;
      g.x = reinterpret_cast<const float*>(e->xw2); g.x_ld = H; g.x_row_mul = 1; g.x_row_off = 0;
      g.kcache = e->L[l].k_cross; g.vcache = e->L[l].v_cross; g.kv_rows_per_b = e->N; g.kv_cap = c.max_enc; g.nheads = nkc;
      if (one_launch) { g.kv_layers = e->kv_layers; g.kv_nlayers = c.num_layers; }
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_KV>(g, st)));
    }
    if (e->xfold_ne && B == 1 && !e->in_capture) hipEventRecord(e->ev_kv, st);  // the fold (ptts_prefill) may start now, beside the layer stack
  }
  if (with_embed) {
    EmbedArgs ea = {};
    ea.tables = e->embed; ea.pos_table = c.rope ? nullptr : e->pos_table; ea.prompt = prefill ? e->ffn : nullptr;
    ea.ids = e->ids; ea.ids_ld = e->ids_ld; ea.cur_len = e->cur_len; ea.dims = e->dims; ea.h = e->h;
    ea.H = H; ea.K = c.num_codebooks; ea.V1 = c.vocab_size + 1; ea.bos = c.bos_token_id; ea.pad = c.pad_token_id;
    ea.prefill = prefill ? 1 : 0;
    hipLaunchKernelGGL((embed_kernel<WT>), dim3(Q, B), dim3(256), 0, st, ea);
  }
  if (e->use_gemv && !prefill && M <= e->gemv_rows) {
    // batch 1..4: 8 row-per-wave GEMV / attention nodes per layer, every weight matrix spread over all CUs and read once
    const float* rc = c.rope ? e->rope_cos : nullptr;
    const float* rs = c.rope ? e->rope_sin : nullptr;
    const int mode = c.dtype == PTTS_F32 ? GV_F32 : (e->w8 ? GV_BF16_W8 : GV_BF16);
    // single utterance, sinusoidal positions: the two nodes of the self-attention block's first half as one (PTTS_NO_FUSE_QA=1: two nodes)
    // (round 5: 2..8 utterances too - one grid slice per utterance, the slices of a head share its weight rows in the L2; PTTS_FUSE_QA_MAX=1: one only)
    // measured, Mini-v1 us per step at contexts ~210 / ~460 / ~710, fused | two nodes (profiles/r05_experiments.txt call 2): 2 utterances 720 / 756 / 791 |
    // 769 / 777 / 791; 3: 741 / 788 / 862 | 780 / 796 / 821; 4: 777 / 864 / 923 | 803 / 832 / 862; 8: 966 / 1010 / 1068 | 907 / 965 / 1023 (the slices of
    // a head re-read its q / k / v rows from the L2 once per utterance and split): on up to fuse_qa_max utterances (3), PTTS_FUSE_QA_MAX forces a bound
    const bool fuse_qa = e->fuse_qa && (M == 1 || (M <= e->fuse_qa_max && mode != GV_F32)) && !c.rope && ptts_qkvattn_ok(H, mode);
    // KV splits of the fused node: the smallest count whose FIRST batch of row groups (8 waves x 4 groups x 8 bf16 / 4 fp32 rows per split,
    // requested before q exists) covers the context bucket this graph is captured for - fewer workgroups recompute the head's q rows, and no
    // split needs a second, dependent K/V batch (context ~210 / ~460 / ~710: 2 splits 554 / 562 / 586 us per step, 4 splits 575 / 577 / 579,
    // 8 splits 645 / 646 / 648; profiles/r04_experiments.txt call 15)
    int S_f = 1;
    const int S_cap = M == 1 ? 8 : (M <= 4 ? 4 : 2);  // the combine prologue of the out_proj node holds S + 1 slots per lane: 12-wave workgroups (5..8 utterances) fit 2 splits
    while (S_f < S_cap && S_f * ptts_qkvattn_rows_per_split(mode) < e->kv_bound) S_f *= 2;
    if (e->fuse_qa_s > 0) S_f = std::min(e->fuse_qa_s, S_cap);
    auto gv = [&](int pro, int epi, int S, GemvArgs g, const char* what) -> int {
      g.M = M;
      const int rc_ = ptts_gemv_launch(mode, pro, epi, S, g, st);
      if (rc_ == -1) return ptts_fail(PTTS_E_UNSUPPORTED, "gemv: no instance for %s (N=%d K=%d M=%d)", what, g.N, g.K, M);
      if (rc_ != 0) return ptts_fail(PTTS_E_HIP, "gemv launch failed (%s)", what);
      return PTTS_OK;
    };
#ifdef PTTS_TIMING
#define PTTS_DBG_NODE(args, l_, k_) (args).dbg = (e->dbg_stamps && M == 1) ? e->dbg_stamps + ((size_t)(l_) * 5 + (k_)) * 48 : nullptr
#else
#define PTTS_DBG_NODE(args, l_, k_) do { } while (0)
#endif
    for (int l = 0; l < c.num_layers; ++l) {
      const LayerW& w = e->L[l];
      if (fuse_qa) {  // LN1 + the head's q / k / v rows + split-KV self-attention + append in ONE node (qkv_attn_kernel), then combine + out_proj
        QkvAttnArgs q = {};
        q.W = w.qkv_rm; q.wscale = w.qkv_sc; q.x = e->h; q.gamma = w.ln1_g; q.beta = w.ln1_b;
        q.kcache = w.k_self; q.vcache = w.v_self; q.cur_len = e->cur_len; q.P = &e->dims->P; q.mask = e->prompt_mask;
        q.part = e->part; q.stats = e->stats; q.cap = c.max_ctx; q.kv_bound = e->kv_bound; q.mask_ld = e->max_prompt;
        q.S = S_f; q.nheads = nh; q.H = H; q.kv_heads = nkv; q.scale = scale; q.M = M; q.x_ld = H;
        PTTS_DBG_NODE(q, l, 0);
        if (ptts_qkvattn_launch(mode, q, st) != 0) return ptts_fail(PTTS_E_HIP, "qkv_attn launch failed");
        GemvArgs g = {};
        g.W = w.o_rm; g.wscale = w.o_sc; g.out = e->h; g.out_ld = H; g.N = H; g.K = H; g.part = e->part; g.stats = e->stats; g.nheads = nh;
        PTTS_DBG_NODE(g, l, 1);
        PTTS_TRY(gv(GV_ATTN2, GV_RESID, S_f, g, "combine+out_proj"));
      } else {
      {  // LN1 + fused QKV projection (:1020-1021, :848-850)
        GemvArgs g = {};
        g.W = w.qkv_rm; g.wscale = w.qkv_sc; g.x = e->h; g.x_ld = H; g.gamma = w.ln1_g; g.beta = w.ln1_b; g.out = e->qkv; g.out_ld = QKV; g.N = QKV; g.K = H;
        PTTS_TRY(gv(GV_LN, GV_STORE, 1, g, "LN1+QKV"));
      }
      {  // causal self-attention over the KV arena, fused RoPE + append
        AttnArgs a = {};
        a.q = e->qkv; a.q_ld = QKV; a.knew = e->qkv + H; a.vnew = e->qkv + H + Hkv; a.kv_ld = QKV;
        a.kv_heads = nkv; a.n_rep = nh / nkv;
        a.kcache = w.k_self; a.vcache = w.v_self; a.cap = c.max_ctx; a.kv_bound = e->kv_bound; a.cur_len = e->cur_len; a.dims = e->dims;
        a.mask = e->prompt_mask; a.mask_ld = e->max_prompt; a.cos = rc; a.sin = rs;
        a.part = e->part; a.stats = e->stats; a.S = e->S_self; a.Q = 1; a.nheads = nh; a.H = H; a.cross = 0;
        a.fused_append = 1; a.scale = scale;
        a.direct_out = e->S_self == 1 ? e->xw : nullptr;
        PTTS_TRY((launch_attn<WT>(a, B, st, e->attn_waves)));
      }
      {  // [combine splits] + out_proj + residual (:1034)
        GemvArgs g = {};
        g.W = w.o_rm; g.wscale = w.o_sc; g.out = e->h; g.out_ld = H; g.N = H; g.K = H;
        if (e->S_self == 1) { g.xw = e->xw; g.xw_ld = H; PTTS_TRY(gv(GV_COPY, GV_RESID, 1, g, "out_proj")); }
        else { g.part = e->part; g.stats = e->stats; g.nheads = nh; PTTS_TRY(gv(GV_ATTN, GV_RESID, e->S_self, g, "combine+out_proj")); }
      }
      }
      bool x_fused = false;  // the cross block left its output as per-head partial rows (summed by the LN3 + fc1 node)
      if (e->xfold_valid && M == 1 && (e->fuse_x < 0 ? H <= 1024 : e->fuse_x != 0) && e->xfold_ne == 64 && ptts_xfoldattn_ok(H, nh, c.dtype == PTTS_F32 ? GV_F32 : GV_BF16)) {
        // folded cross block as ONE node: LN2 + the head's rows of M + softmax + the head's columns of U -> xpart[h][:]
        XfoldAttnArgs x = {};
        x.Mw = w.xM; x.Uw = w.xU; x.x = e->h; x.gamma = w.ln2_g; x.beta = w.ln2_b; x.mask = e->enc_mask; x.n_valid = &e->dims->N;
        x.xpart = e->xpart; x.nheads = nh; x.H = H; x.nur = e->fuse_x_nur;
        PTTS_DBG_NODE(x, l, 2);
        if (ptts_xfoldattn_launch(c.dtype == PTTS_F32 ? GV_F32 : GV_BF16, x, st) != 0) return ptts_fail(PTTS_E_HIP, "xfold_attn launch failed");
        x_fused = true;
      } else if (e->xfold_valid && M == 1) {
        // folded cross block (xfold_*_kernel at prefill): LN2 + (K Wq) x -> base-2 scores; per-head softmax + (Wo V^T) p + residual
        const int K2 = nh * e->xfold_ne, gm = c.dtype == PTTS_F32 ? GV_F32 : GV_BF16;  // M / U are in the engine dtype, never e4m3
        GemvArgs g = {};
        g.W = w.xM; g.x = e->h; g.x_ld = H; g.gamma = w.ln2_g; g.beta = w.ln2_b; g.out = e->qc; g.out_ld = K2; g.N = K2; g.K = H; g.M = 1;
        if (ptts_gemv_launch(gm, GV_LN, GV_STORE, 1, g, st) != 0) return ptts_fail(PTTS_E_HIP, "gemv launch failed (LN2 + folded scores)");
        GemvArgs g2 = {};
        g2.W = w.xU; g2.x = e->qc; g2.mask = e->enc_mask; g2.n_valid = &e->dims->N; g2.ne = e->xfold_ne;
        g2.out = e->h; g2.out_ld = H; g2.N = H; g2.K = K2; g2.M = 1;
        if (ptts_gemv_launch(gm, GV_SOFTMAX, GV_RESID, 1, g2, st) != 0) return ptts_fail(PTTS_E_HIP, "gemv launch failed (softmax + folded out_proj)");
      } else {
      if (e->fuse_xq && !c.rope && ptts_xqattn_ok(H, mode)) {
        // LN2 + the head's cross-q rows + cross-attention of one (head, utterance) as ONE node (xq_attn_kernel): 1..8 utterances without the static fold
        XqAttnArgs x = {};
        x.W = w.cq_rm; x.wscale = w.cq_sc; x.x = e->h; x.x_ld = H; x.gamma = w.ln2_g; x.beta = w.ln2_b;
        x.kcache = w.k_cross; x.vcache = w.v_cross; x.mask = e->enc_mask; x.mask_ld = c.max_enc; x.cap = c.max_enc; x.n_valid = &e->dims->N;
        x.out = e->xw; x.out_ld = H; x.nheads = nh; x.H = H; x.kv_heads = nkc; x.M = M; x.scale = scale;
        if (ptts_xqattn_launch(mode, x, st) != 0) return ptts_fail(PTTS_E_HIP, "xq_attn launch failed");
      } else {
      {  // LN2 + cross q projection (:1040, :855)
        GemvArgs g = {};
        g.W = w.cq_rm; g.wscale = w.cq_sc; g.x = e->h; g.x_ld = H; g.gamma = w.ln2_g; g.beta = w.ln2_b; g.out = e->qc; g.out_ld = H; g.N = H; g.K = H;
        PTTS_TRY(gv(GV_LN, GV_STORE, 1, g, "LN2+cross q"));
      }
      {  // cross-attention against the static description K/V: one workgroup per head, softmax finished in the kernel
        AttnArgs a = {};
        a.q = e->qc; a.q_ld = H; a.kcache = w.k_cross; a.vcache = w.v_cross; a.cap = c.max_enc; a.kv_bound = c.max_enc;
        a.cur_len = e->cur_len; a.dims = e->dims; a.mask = e->enc_mask; a.mask_ld = c.max_enc;
        a.cos = rc; a.sin = rs;  // quirk: q rotated, keys not (:858 vs :880)
        a.part = e->part; a.stats = e->stats; a.S = 1; a.Q = 1; a.nheads = nh; a.H = H; a.cross = 1;
        a.kv_heads = nkc; a.n_rep = nh / nkc; a.fused_append = 0; a.scale = scale; a.direct_out = e->xw;
        PTTS_TRY((launch_attn<WT>(a, B, st, e->cross_waves)));
      }
      }
      {  // cross out_proj + residual (:1052)
        GemvArgs g = {};
        g.W = w.co_rm; g.wscale = w.co_sc; g.xw = e->xw; g.xw_ld = H; g.out = e->h; g.out_ld = H; g.N = H; g.K = H;
        PTTS_TRY(gv(GV_COPY, GV_RESID, 1, g, "cross out_proj"));
      }
      }
      {  // LN3 + fc1 + GELU (engine dtype), fc2 + residual (:1059-1064)
        GemvArgs g = {};
        g.W = w.fc1_rm; g.wscale = w.fc1_sc; g.x = e->h; g.x_ld = H; g.gamma = w.ln3_g; g.beta = w.ln3_b;
        g.out = reinterpret_cast<float*>(e->xw2); g.out_ld = F; g.N = F; g.K = H;
        PTTS_DBG_NODE(g, l, 3);
        if (x_fused) { g.part = e->xpart; g.npart = nh; g.hsum = e->h2; PTTS_TRY(gv(GV_LNP, GV_GELU_WT, 1, g, "partial rows+LN3+fc1")); }
        else PTTS_TRY(gv(GV_LN, GV_GELU_WT, 1, g, "LN3+fc1"));
        GemvArgs g2 = {};
        g2.W = w.fc2_rm; g2.wscale = w.fc2_sc; g2.xw = e->xw2; g2.xw_ld = F; g2.out = e->h; g2.out_ld = H; g2.N = H; g2.K = F;
        if (x_fused) g2.resid = e->h2;  // h = (x + partial rows) + fc2(...)
        PTTS_DBG_NODE(g2, l, 4);
        PTTS_TRY(gv(GV_COPY, GV_RESID, 1, g2, "fc2"));
      }
    }
    GemvArgs g = {};  // final LayerNorm + all K LM heads (:1632, :1917-1960)
    g.W = e->heads_rm; g.wscale = e->heads_sc; g.x = e->h; g.x_ld = H; g.gamma = e->lnf_g; g.beta = e->lnf_b; g.out = e->logits;
    g.out_ld = c.num_codebooks * c.vocab_size; g.N = c.num_codebooks * c.vocab_size; g.K = H;
    PTTS_DBG_NODE(g, c.num_layers, 0);
    PTTS_TRY(gv(GV_LN, GV_STORE, 1, g, "final LN + LM heads"));
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "forward launch failed: %s", hipGetErrorString(err));
    return PTTS_OK;
  }
  // KV splits of the self-attention: the engine's split factor at decode (long context, few rows); ONE at prefill - the context is the
  // prompt (tens of positions) and there is one workgroup per (head, position) anyway: splitting 4 ways + a combine kernel cost
  // 15 + 9 us per layer on the time-to-first-token path against ~7 us unsplit (profiles/r03_prefill_kernels.txt)
  const int S_used = prefill ? 1 : e->S_self;
  const bool lns = e->use_lns && !prefill && M > 8 && M <= 32;  // LayerNorm statistics carried by the producer GEMM (PRO_LNS)
  // decode at batch > 8: engine-dtype activations travel between kernels in MFMA B-fragment order (ptts_lm_kernels.h: fo_vec_index)
  // (and the prefill while its rows stay on the strip kernels, M <= 256: a single utterance's 33-position prefill is 240 latency-bound
  // launches on the time-to-first-token path; PTTS_NO_FO_PREFILL=1 keeps the row-major layout there for A/B)
  static const bool fo_prefill = !(ptts_dev_env("PTTS_NO_FO_PREFILL") && atoi(ptts_dev_env("PTTS_NO_FO_PREFILL")));
  const int fo = (e->use_fo && M > 8 && (!prefill || (fo_prefill && M <= 256))) ? 1 : 0;
  // LayerNorm + projection as one node (lnproj_fused_kernel): decode, batch > 8, Mini / Large widths, weights in the engine dtype (e4m3 strips keep
  // streaming bytes through the strip GEMMs)
  const int KTf = Elem<WT>::KT;
  const int lnproj = e->lnproj >= 0 ? e->lnproj : 3;
  // utterances per workgroup of that node: 8 (one row per wave) up to 40, 16 (two rows per wave, the whole MFMA tile, half the weight re-reads) above
  const int lnproj_g = e->lnproj_g > 0 ? e->lnproj_g : (M <= 40 ? 8 : 16);
  // (round 5: the prefill rows of a short prompt, 8 < M <= 40, run the same fused nodes - three rows_prep nodes less per layer on the
  //  time-to-first-token path: prefill 1.41 -> 1.29 ms, first token 2.41 -> 2.28 ms at 33 rows, profiles/r05_experiments.txt call 2; PTTS_LNPROJ_PREFILL=0: off)
  static const bool lnproj_prefill = !(ptts_dev_env("PTTS_LNPROJ_PREFILL") && !atoi(ptts_dev_env("PTTS_LNPROJ_PREFILL")));
  const bool lnproj_ok = lnproj > 0 && (!prefill || (lnproj_prefill && M <= 40)) && M > 8 && !e->w8_strips && (H == 1024 || H == 1536) && ((H / KTf) / 2) % 8 == 0 && QKV % 64 == 0 && F % 64 == 0;
  // prefill attention on the tiled kernel (8 query rows per workgroup share the K / V tile; PTTS_PREFILL_ATTN=0: one workgroup per query row, attn_kernel)
  const int prefill_attn_mode = getenv("PTTS_PREFILL_ATTN") ? atoi(getenv("PTTS_PREFILL_ATTN")) : 3;  // 0: one workgroup per query row, 1: tiled VALU kernel, 2: f32-MFMA kernel, 3: by batch
  const bool prefill_attn = prefill_attn_mode != 0;
  // prefill rows on the fused LN1 + QKV node, sinusoidal positions, engine-dtype cache: the node's epilogue writes the cache rows itself (no
  // kv_append node: 24 launches of ~4 us + their boundaries off the time-to-first-token path; PTTS_KV_IN_QKV=0: the separate node)
  static const bool kv_in_qkv_on = !(ptts_dev_env("PTTS_KV_IN_QKV") && !atoi(ptts_dev_env("PTTS_KV_IN_QKV")));
  const bool kv_in_qkv = kv_in_qkv_on && prefill && lnproj_ok && !c.rope && !e->L[0].ks_self;
  // ... and above 256 rows (a batch's prompts on the > 256-row GEMMs; round 6): the QKV projection's tile epilogue writes them (GemmArgs::kv_col0) -
  // 24 kv_append launches of ~6 us + their boundaries off the time-to-first-token path of a batch. Between 41 and 256 rows the kv_append node stays
  // (fragment-order rows_prep + strips: measured path of round 5).
  const bool kv_in_gemm = kv_in_qkv_on && prefill && !kv_in_qkv && !lnproj_ok && M > 256 && !c.rope && !e->L[0].ks_self;
#ifdef PTTS_TIMING
#define PTTS_DBG_BIG(args, l_, k_) (args).dbg = (e->dbg_stamps && !prefill && M > 8) ? e->dbg_stamps + ((size_t)(l_) * 7 + (k_)) * 48 : nullptr
#else
#define PTTS_DBG_BIG(args, l_, k_) do { } while (0)
#endif
  for (int l = 0; l < c.num_layers; ++l) {
    const LayerW& w = e->L[l];
    if (lnproj_ok) {  // LN1 + fused QKV projection in one node
      LnProjArgs p = {};
      PTTS_DBG_BIG(p, l, 0);
      p.W = w.qkv; p.x = e->h; p.x_ld = H; p.gamma = w.ln1_g; p.beta = w.ln1_b; p.K = H; p.out = e->qkv; p.out_ld = QKV; p.M = M; p.N = QKV;
      if (kv_in_qkv) { p.kcache = w.k_self; p.vcache = w.v_self; p.kv_Q = Q; p.kv_cap = c.max_ctx; p.kv_heads = nkv; p.kv_H = H; }
      PTTS_TRY((launch_lnproj<WT, EPI_STORE>(e, p, st, lnproj_g)));
    } else {  // LN1 + fused QKV projection
      GemmArgs g = {}; g.decode = dec;
      g.W = w.qkv; g.W8 = w.qkv_p8; g.wscale = w.qkv_sc; g.x = e->h; g.x_ld = H; g.x_row_mul = 1; g.gamma = w.ln1_g; g.beta = w.ln1_b;
      g.out = e->qkv; g.out_ld = QKV; g.M = M; g.N = QKV; g.K = H; g.x_fo = fo;
      if (kv_in_gemm) { g.kcache = w.k_self; g.vcache = w.v_self; g.kv_rows_per_b = Q; g.kv_cap = c.max_ctx; g.nheads = nkv; g.kv_col0 = H; }
      PTTS_TRY((gemm_with_prologue<WT, PRO_LN, EPI_STORE>(e, g, st)));
    }
    if (prefill && !kv_in_qkv && !kv_in_gemm) {
      bool done8 = false;
      if constexpr (sizeof(WT) == 2) {
        if (w.ks_self) {
          hipLaunchKernelGGL((kv_append_kernel<WT, true>), dim3(Q, nkv, B), dim3(64), 0, st, e->qkv + H, e->qkv + H + Hkv, QKV, w.k_self, w.v_self, c.max_ctx,
                             Q, nkv, c.rope ? e->rope_cos : nullptr, c.rope ? e->rope_sin : nullptr, w.ks_self, w.vs_self);
          done8 = true;
        }
      }
      if (!done8)
      hipLaunchKernelGGL((kv_append_kernel<WT>), dim3(Q, nkv, B), dim3(64), 0, st, e->qkv + H, e->qkv + H + Hkv, QKV, w.k_self,
                         w.v_self, c.max_ctx, Q, nkv, c.rope ? e->rope_cos : nullptr, c.rope ? e->rope_sin : nullptr);
    }
    {  // causal self-attention over the KV arena
      AttnArgs a = {};
      a.q = e->qkv; a.q_ld = QKV; a.knew = e->qkv + H; a.vnew = e->qkv + H + Hkv; a.kv_ld = QKV;
      a.kv_heads = nkv; a.n_rep = nh / nkv;
      a.kcache = w.k_self; a.vcache = w.v_self; a.cap = c.max_ctx; a.kv_bound = prefill ? c.max_ctx : e->kv_bound; a.cur_len = prefill ? nullptr : e->cur_len; a.dims = e->dims;
      a.mask = e->prompt_mask; a.mask_ld = e->max_prompt;
      a.cos = c.rope ? e->rope_cos : nullptr; a.sin = c.rope ? e->rope_sin : nullptr;
      a.part = e->part; a.stats = e->stats; a.S = S_used; a.Q = Q; a.nheads = nh; a.H = H; a.cross = 0;
      a.fused_append = prefill ? 0 : 1; a.scale = scale;
      a.kscale = w.ks_self; a.vscale = w.vs_self; a.hostP = e->P; a.hostN = e->N;
      a.direct_out = S_used == 1 ? e->xw : nullptr; a.out_fo = fo;
      PTTS_DBG_BIG(a, l, 1);
      if (prefill && prefill_attn) PTTS_TRY((launch_prefill_attn<WT>(a, B, st, prefill_attn_mode)));
      else PTTS_TRY((launch_attn<WT>(a, B, st, prefill ? 4 : e->attn_waves)));
    }
    {  // [combine splits] + out_proj + residual
      GemmArgs g = {}; g.decode = dec;
      g.W = w.o; g.W8 = w.o_p8; g.wscale = w.o_sc; g.part = e->part; g.stats = e->stats; g.S = S_used; g.nheads = nh;
      g.out = e->h; g.out_ld = H; g.M = M; g.N = H; g.K = H; g.x_fo = fo;
      if (lns) g.stats_out = e->lnstat;  // strip statistics of the new residual rows for LN2 (PRO_LNS)
      PTTS_DBG_BIG(g, l, 2);
      if (S_used == 1) {
        g.x = reinterpret_cast<const float*>(e->xw); g.x_ld = H; g.x_row_mul = 1;
        PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_RESID>(g, st)));
      } else {
        PTTS_TRY((gemm_with_prologue<WT, PRO_ATTN, EPI_RESID>(e, g, st)));
      }
    }
    const int KTw = Elem<WT>::KT;
    if (!prefill && (M <= 8 || (e->xattn_groups && M <= e->xattn_groups_max)) && (H / KTw) % 16 == 0 && (H == 512 || H == 1024 || H == 1536)) {
      // decode, small batch: LN2 + cross q projection + cross-attention fused, one workgroup per head
      XAttnArgs x = {};
      x.W = w.cq; x.x = e->h; x.x_ld = H; x.x_row_mul = 1; x.x_row_off = 0; x.gamma = w.ln2_g; x.beta = w.ln2_b; x.K = H;
      x.invK = 1.0f / (float)H; x.kcache = w.k_cross; x.vcache = w.v_cross; x.cap = c.max_enc; x.cur_len = e->cur_len; x.dims = e->dims;
      x.mask = e->enc_mask; x.mask_ld = c.max_enc; x.cos = c.rope ? e->rope_cos : nullptr; x.sin = c.rope ? e->rope_sin : nullptr;
      x.out = e->xw; x.B = M; x.nheads = nh; x.kv_heads = nkc; x.n_rep = nh / nkc; x.scale = scale; x.out_fo = fo;
      PTTS_DBG_BIG(x, l, 3);
      // utterances per workgroup: 8 (one per wave) up to 8 utterances; above, e->xattn_g (8 / 4 / 2: heads x ceil(M / g) workgroups, the 8 / g
      // waves of an utterance split the description's row groups)
      // measured (profiles/r04_experiments.txt, us per step at mid context, g = 8 / 4 / 2): batch 32 1432 / 1381 / 1360, batch 128 2596 / 2682 / 2860
      const int gsz = M <= 8 ? 8 : (e->xattn_g ? e->xattn_g : (e->xattn_g_ok ? (M <= 32 ? 2 : (M <= 64 ? 4 : 8)) : 8));
      const int mg = M < gsz ? M : gsz;
      const size_t sh = (size_t)mg * (H * sizeof(WT) + 16) + 8 * 1024 + (size_t)gsz * 64 * 4 + 8 * 64 * 4 + 64;
      const dim3 xg(nh, (M + gsz - 1) / gsz);
      const bool u16 = ((H / KTw) / 2) % 16 == 0;
      if (H == 1024 && u16) {  // Mini-v1
        if (gsz == 4) ptts_klaunch(xattn_fused_kernel<WT, 16, 4, 4>, xg, dim3(512), sh, st, x);
        else if (gsz == 2) ptts_klaunch(xattn_fused_kernel<WT, 16, 4, 2>, xg, dim3(512), sh, st, x);
        else ptts_klaunch(xattn_fused_kernel<WT, 16, 4, 8>, xg, dim3(512), sh, st, x);
      } else if (H == 1024) ptts_klaunch(xattn_fused_kernel<WT, 8, 4>, xg, dim3(512), sh, st, x);
      else if (H == 1536) {  // Large-v1
        if (gsz == 4) ptts_klaunch(xattn_fused_kernel<WT, 8, 6, 4>, xg, dim3(512), sh, st, x);
        else if (gsz == 2) ptts_klaunch(xattn_fused_kernel<WT, 8, 6, 2>, xg, dim3(512), sh, st, x);
        else ptts_klaunch(xattn_fused_kernel<WT, 8, 6, 8>, xg, dim3(512), sh, st, x);
      } else ptts_klaunch(xattn_fused_kernel<WT, 8, 2>, xg, dim3(512), sh, st, x);                            // hidden 512
    } else {
    if (prefill && lnproj_ok) {  // LN2 + cross q projection as one node
      LnProjArgs p = {};
      p.W = w.cq; p.x = e->h; p.x_ld = H; p.gamma = w.ln2_g; p.beta = w.ln2_b; p.K = H; p.out = e->qc; p.out_ld = H; p.M = M; p.N = H;
      PTTS_TRY((launch_lnproj<WT, EPI_STORE>(e, p, st, lnproj_g)));
    } else
    {  // LN2 + cross q projection
      GemmArgs g = {}; g.decode = dec;
      g.W = w.cq; g.x = e->h; g.x_ld = H; g.x_row_mul = 1; g.gamma = w.ln2_g; g.beta = w.ln2_b;
      g.out = e->qc; g.out_ld = H; g.M = M; g.N = H; g.K = H; g.x_fo = fo;
      if (lns) g.lnstat = e->lnstat;
      PTTS_TRY((gemm_with_prologue<WT, PRO_LN, EPI_STORE>(e, g, st)));
    }
    {  // cross-attention against the static description K/V
      AttnArgs a = {};
      a.q = e->qc; a.q_ld = H; a.kcache = w.k_cross; a.vcache = w.v_cross; a.cap = c.max_enc; a.kv_bound = c.max_enc;
      a.cur_len = prefill ? nullptr : e->cur_len; a.dims = e->dims; a.mask = e->enc_mask; a.mask_ld = c.max_enc;
      a.cos = c.rope ? e->rope_cos : nullptr; a.sin = c.rope ? e->rope_sin : nullptr;  // quirk: q rotated, keys not (:858 vs :880)
      a.part = e->part; a.stats = e->stats; a.S = 1; a.Q = Q; a.nheads = nh; a.H = H; a.cross = 1;
      a.kv_heads = nkc; a.n_rep = nh / nkc;
      a.fused_append = 0; a.scale = scale; a.hostP = e->P; a.hostN = e->N;
      a.direct_out = e->xw; a.out_fo = fo;  // the description is short: never split, softmax finished in the attention kernel
      if (prefill && prefill_attn) PTTS_TRY((launch_prefill_attn<WT>(a, B, st, prefill_attn_mode)));
      else PTTS_TRY((launch_attn<WT>(a, B, st, prefill ? 4 : e->cross_waves)));  // decode: as few waves as cover the description (no LDS combine at 1)
    }
    }
    {  // cross out_proj + residual, activations read straight from the attention output
      GemmArgs g = {}; g.decode = dec;
      g.W = w.co; g.W8 = w.co_p8; g.wscale = w.co_sc; g.x = reinterpret_cast<const float*>(e->xw); g.x_ld = H; g.x_row_mul = 1; g.x_fo = fo;
      g.out = e->h; g.out_ld = H; g.M = M; g.N = H; g.K = H;
      if (lns) g.stats_out = e->lnstat;  // for LN3
      PTTS_DBG_BIG(g, l, 4);
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_RESID>(g, st)));
    }
    {  // LN3 + fc1 + GELU, then fc2 + residual. Above 8 rows the GELU output is written in the engine dtype so fc2
       // stages it with plain copies too.
      GemmArgs g = {}; g.decode = dec;
      g.W = w.fc1; g.W8 = w.fc1_p8; g.wscale = w.fc1_sc; g.x = e->h; g.x_ld = H; g.x_row_mul = 1; g.gamma = w.ln3_g; g.beta = w.ln3_b;
      g.out_ld = F; g.M = M; g.N = F; g.K = H;
      GemmArgs g2 = {}; g2.decode = dec;
      g2.W = w.fc2; g2.W8 = w.fc2_p8; g2.wscale = w.fc2_sc; g2.x_ld = F; g2.x_row_mul = 1; g2.out = e->h; g2.out_ld = H; g2.M = M; g2.N = H; g2.K = F;
      if (big) {
        g.out = reinterpret_cast<float*>(e->xw2); g.x_fo = fo; g.out_fo = fo;
        if (lns) g.lnstat = e->lnstat;
        if (lnproj_ok && (lnproj >= 3 || (lnproj == 2 && M > 32))) {  // LN3 + fc1 + GELU in one node
          LnProjArgs p = {};
          p.W = w.fc1; p.x = e->h; p.x_ld = H; p.gamma = w.ln3_g; p.beta = w.ln3_b; p.K = H; p.out = e->xw2; p.out_ld = F; p.out_fo = fo; p.M = M; p.N = F;
          PTTS_DBG_BIG(p, l, 5);
          PTTS_TRY((launch_lnproj<WT, EPI_GELU_WT>(e, p, st, lnproj_g)));
        } else
        PTTS_TRY((gemm_with_prologue<WT, PRO_LN, EPI_GELU_WT>(e, g, st)));
        g2.x = reinterpret_cast<const float*>(e->xw2); g2.x_fo = fo;
        PTTS_DBG_BIG(g2, l, 6);
        // fc2 runs un-split at every batch size (round 6, profiles/r06_experiments.txt calls 15-16: the split-K partials of round 2 cost the next LN1 node
        // +1.0 us and the out_proj epilogue +1.2 us per layer - un-split: -2 % at 32, -3.3 % at 40, -7 % at Large-v1 x 32, equal at 12..16; only the
        // e4m3-weight Mini engine loses 1.4-3 %; the split path and its folds are gone)
        PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_RESID>(g2, st)));
      } else {
        g.out = e->ffn;
        PTTS_TRY((launch_gemm<WT, PRO_LN, EPI_GELU>(g, st)));
        g2.x = e->ffn;
        PTTS_TRY((launch_gemm<WT, PRO_PLAIN, EPI_RESID>(g2, st)));
      }
    }
  }
  // (the final LayerNorm + LM heads as one lnproj_fused_kernel node at 9..40 utterances measured neutral to +0.6 %: the 20 MB of head weights are
  //  re-read once per group of 8 utterances - profiles/r04_experiments.txt call 23; the rows_prep + strip GEMM pair stays)
  {  // final LayerNorm + all K LM heads as one [K*V, H] projection, last position of each utterance only
    GemmArgs g = {}; g.decode = dec;
    g.W = e->heads; g.W8 = e->heads_p8; g.wscale = e->heads_sc; g.x = e->h; g.x_ld = H; g.x_row_mul = Q; g.x_row_off = Q - 1; g.gamma = e->lnf_g; g.beta = e->lnf_b;
    g.out = e->logits; g.out_ld = c.num_codebooks * c.vocab_size; g.M = B; g.N = c.num_codebooks * c.vocab_size; g.K = H;
    g.x_fo = (e->use_fo && B > 8 && (!prefill || (fo_prefill && B <= 256))) ? 1 : 0;
    PTTS_TRY((gemm_with_prologue<WT, PRO_LN, EPI_STORE>(e, g, st)));
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "forward launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}

int launch_tail(ptts_engine* e, hipStream_t st, bool embed_next) {
  TailArgs t = {};
  if (embed_next) {
    t.tables = e->embed; t.pos_table = e->cfg.rope ? nullptr : e->pos_table; t.dims = e->dims; t.h = e->h;
    t.H = e->cfg.hidden_size; t.bos = e->cfg.bos_token_id; t.bf16_tables = e->cfg.dtype == PTTS_BF16;
  }
  t.logits = e->logits; t.ids = e->ids; t.ids_ld = e->ids_ld; t.cur_len = e->cur_len; t.unfinished = e->unfinished;
  t.has_eos = e->has_eos; t.first_unf = e->first_unf; t.gen = e->gen; t.sort_buf = e->sort_buf;
  t.B = e->B; t.K = e->cfg.num_codebooks; t.V = e->cfg.vocab_size; t.eos = e->cfg.eos_token_id; t.pad = e->cfg.pad_token_id;
  // one wave per codebook row (greedy arg-max or the sort-free sampler), at least 4 waves for the embedding of the next column
  const int nw = std::min(std::max(e->cfg.num_codebooks, 4), 16);
  if (t.V <= 512) hipLaunchKernelGGL(tail_kernel<8>, dim3(e->B), dim3(nw * 64), 0, st, t);
  else if (t.V <= 1152) hipLaunchKernelGGL(tail_kernel<18>, dim3(e->B), dim3(nw * 64), 0, st, t);
  else hipLaunchKernelGGL(tail_kernel<32>, dim3(e->B), dim3(nw * 64), 0, st, t);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "tail launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}

// static cross-attention folding for the utterance just prefilled (xfold_*_kernel): 2 launches covering every layer
template <typename WT, bool W8>
int fold_cross(ptts_engine* e, hipStream_t st) {
  const ptts_config& c = e->cfg;
  const int H = c.hidden_size, nh = c.num_heads, NE = e->xfold_ne, n_rep = nh / e->nkc, L = c.num_layers;
  const float qscale = 1.44269504088896340736f / sqrtf((float)(H / nh));
  // tiled kernels (round 5; bit-identical to the row kernels, PTTS_FOLD_TILED=0 selects those for A/B)
  static const bool tiled = !(ptts_dev_env("PTTS_FOLD_TILED") && !atoi(ptts_dev_env("PTTS_FOLD_TILED")));
  if (tiled && H % 64 == 0 && NE <= 64) {
    hipLaunchKernelGGL((xfold_tile_kernel<WT, W8, false>), dim3(H / 64, nh, L), dim3(256), 0, st, e->fold_layers, nh, H, NE, c.max_enc, n_rep, e->dims, qscale);
    hipLaunchKernelGGL((xfold_tile_kernel<WT, W8, true>), dim3(H / 64, nh, L), dim3(256), 0, st, e->fold_layers, nh, H, NE, c.max_enc, n_rep, e->dims, qscale);
  } else {
    hipLaunchKernelGGL((xfold_m_kernel<WT, W8>), dim3((H / 8 + 63) / 64, NE, L * nh), dim3(64), 0, st, e->fold_layers, nh, H, NE, c.max_enc, n_rep,
                       e->dims, qscale);
    hipLaunchKernelGGL((xfold_u_kernel<WT, W8>), dim3((nh * NE + 63) / 64, H, L), dim3(64), 0, st, e->fold_layers, H, NE, nh, c.max_enc, n_rep, e->dims);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "cross-attention fold launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}

int forward_dispatch(ptts_engine* e, bool prefill, hipStream_t st, bool with_embed = true) {
  return e->cfg.dtype == PTTS_BF16 ? forward<bf16_t>(e, prefill, st, with_embed) : forward<float>(e, prefill, st, with_embed);
}

// every decode forward appends one self-KV position: advance the host's bound before launching it (eagerly or as a graph)
static const bool g_no_kv_bound = ptts_dev_env("PTTS_NO_KV_BOUND") && atoi(ptts_dev_env("PTTS_NO_KV_BOUND"));
static int advance_kv(ptts_engine* e) {
  e->kv_ub += 1;
  const int cap = e->cfg.max_ctx;
  e->kv_bound = g_no_kv_bound ? cap : std::min(cap, (e->kv_ub + 1 + 63) / 64 * 64);
  return e->kv_bound;
}

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// copy/convert a plain tensor into engine storage
template <typename DT>
int convert_into(DT* dst, const void* src, int src_dtype, size_t n, hipStream_t st) {
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  if (src_dtype == PTTS_F32) hipLaunchKernelGGL((convert_kernel<DT, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, n);
  else hipLaunchKernelGGL((convert_kernel<DT, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, n);
  return PTTS_OK;
}

template <typename WT>
int pack_into(void* dst, const void* src, int src_dtype, int N, int K, int row0, hipStream_t st) {
  constexpr int KT = Elem<WT>::KT;
  if (N % 16 || K % KT || row0 % 16) return ptts_fail(PTTS_E_INVALID, "weight [%d,%d] (row offset %d) not a multiple of the 16x%d MFMA tile", N, K, row0, KT);
  const size_t total = (size_t)(N / 16) * (K / KT) * 64;
  const int blocks = (int)((total + 255) / 256);
  if (src_dtype == PTTS_F32)
    hipLaunchKernelGGL((pack_weight_kernel<WT, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (WT*)dst, N, K, row0 / 16, K / KT);
  else
    hipLaunchKernelGGL((pack_weight_kernel<WT, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (WT*)dst, N, K, row0 / 16, K / KT);
  return PTTS_OK;
}

int pack_dispatch(ptts_engine* e, void* dst, const void* src, int src_dtype, int N, int K, int row0, hipStream_t st) {
  return e->cfg.dtype == PTTS_BF16 ? pack_into<bf16_t>(dst, src, src_dtype, N, K, row0, st) : pack_into<float>(dst, src, src_dtype, N, K, row0, st);
}

// row-major copy (rows [row0, row0 + N) of a [*, K] matrix) in the engine dtype for the GEMV step
int rowmajor_dispatch(ptts_engine* e, void* dst, const void* src, int src_dtype, int N, int K, int row0, hipStream_t st) {
  const size_t off = (size_t)row0 * K, n = (size_t)N * K;
  if (e->cfg.dtype == PTTS_BF16) return convert_into<bf16_t>(reinterpret_cast<bf16_t*>(dst) + off, src, src_dtype, n, st);
  return convert_into<float>(reinterpret_cast<float*>(dst) + off, src, src_dtype, n, st);
}

// NULL is the legacy default stream (what torch.cuda.current_stream() is on ROCm unless the caller switched): pass through.
hipStream_t pick_stream(ptts_engine*, void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int ptts_engine_create(const ptts_config* cfg, ptts_engine** out) {
  PTTS_CHECK(cfg && out, PTTS_E_INVALID, "null argument");
  const ptts_config& c = *cfg;
  PTTS_CHECK(c.hidden_size > 0 && c.num_heads > 0 && c.hidden_size % c.num_heads == 0, PTTS_E_INVALID, "hidden_size %d not divisible by num_heads %d", c.hidden_size, c.num_heads);
  PTTS_CHECK(c.hidden_size / c.num_heads == 64, PTTS_E_UNSUPPORTED, "head_dim must be 64 (Mini/Large v1), got %d", c.hidden_size / c.num_heads);
  PTTS_CHECK(c.dtype == PTTS_F32 || c.dtype == PTTS_BF16, PTTS_E_INVALID, "dtype must be PTTS_F32 or PTTS_BF16");
  PTTS_CHECK(c.hidden_size % 32 == 0 && c.ffn_dim % 32 == 0 && c.vocab_size % 16 == 0, PTTS_E_UNSUPPORTED,
             "hidden_size/ffn_dim must be multiples of 32 and vocab_size of 16");
  PTTS_CHECK(c.hidden_size <= 64 * 4 * LN_MAX_F4, PTTS_E_UNSUPPORTED, "hidden_size > %d unsupported", 64 * 4 * LN_MAX_F4);
  PTTS_CHECK(c.num_codebooks >= 1 && c.num_codebooks <= 32, PTTS_E_INVALID, "num_codebooks out of range");
  PTTS_CHECK(c.vocab_size <= PTTS_SORT_N, PTTS_E_UNSUPPORTED, "vocab_size > %d unsupported by the sampler", PTTS_SORT_N);
  PTTS_CHECK(c.max_batch >= 1 && c.max_ctx >= 2 && c.max_enc >= 1 && c.max_prompt >= 1 && c.max_prompt <= c.max_ctx, PTTS_E_INVALID, "bad capacities");
  const int nkv_ = c.num_kv_heads > 0 ? c.num_kv_heads : c.num_heads, nkc_ = c.num_cross_kv_heads > 0 ? c.num_cross_kv_heads : nkv_;
  PTTS_CHECK(c.num_heads % nkv_ == 0 && c.num_heads % nkc_ == 0, PTTS_E_INVALID, "num_heads %d not divisible by the K/V head counts %d / %d", c.num_heads, nkv_, nkc_);
  PTTS_DEVICE(c.device);
  ptts_engine* e = new ptts_engine();
  e->cfg = c;
  e->nkv = nkv_; e->nkc = nkc_;
  e->esize = c.dtype == PTTS_BF16 ? 2 : 4;
  e->max_prompt = c.max_prompt;
  int rc = PTTS_OK;
  auto fail = [&](int r) { ptts_engine_destroy(e); return r; };
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(ptts_fail(PTTS_E_HIP, "hipStreamCreate failed"));
  if (hipStreamCreateWithFlags(&e->fold_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_kv, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fold, hipEventDisableTiming) != hipSuccess || hipEventCreate(&e->ev_first) != hipSuccess || hipEventCreate(&e->ev_tail0) != hipSuccess || hipEventCreate(&e->ev_pre0) != hipSuccess)
    return fail(ptts_fail(PTTS_E_HIP, "hipStreamCreate / hipEventCreate failed"));
  const int H = c.hidden_size, F = c.ffn_dim, K = c.num_codebooks, V = c.vocab_size, nh = c.num_heads;
  const size_t es = e->esize;
  e->L.resize(c.num_layers);
  // single-utterance decode step on the row-per-wave GEMV kernels: shapes whose rows are whole 1 KiB chunks
  const int gmode = c.dtype == PTTS_F32 ? GV_F32 : GV_BF16;
  // (an engine created for more than GV_MAX_ROWS utterances never takes the GEMV step: it does not hold the row-major copies either -
  // the model-level cache keeps one engine per batch-size class, each with the weight copies its own decode step streams)
  e->use_gemv = ptts_gemv_k_ok(H, gmode) && ptts_gemv_k_ok(F, gmode) && H <= 2048 && c.max_batch <= GV_MAX_ROWS &&
                !(ptts_dev_env("PTTS_NO_GEMV") && atoi(ptts_dev_env("PTTS_NO_GEMV")));
  e->gemv_rows = c.dtype == PTTS_F32 ? 1 : GV_MAX_ROWS;
  if (const char* ev = ptts_dev_env("PTTS_GEMV_ROWS")) e->gemv_rows = std::max(1, std::min(e->gemv_rows, atoi(ev)));  // A/B knob (tools/)
  e->w8 = c.weights_fp8 != 0;
  // static cross-attention folding: single utterance, sinusoidal positions (RoPE rotates the cross query by position), <= 64
  // description tokens, full cross K/V heads not required (n_rep handled), folded widths must be GEMV shapes
  if (e->use_gemv && !c.rope && c.max_enc <= 64 && ptts_gemv_k_ok(nh * 64, gmode) && !(ptts_dev_env("PTTS_NO_XFOLD") && atoi(ptts_dev_env("PTTS_NO_XFOLD"))))
    e->xfold_ne = 64;
  e->w8_strips = e->w8 && !e->use_gemv && !(ptts_dev_env("PTTS_NO_W8_STRIPS") && atoi(ptts_dev_env("PTTS_NO_W8_STRIPS")));
  if (c.kv_fp8 && (c.dtype != PTTS_BF16 || c.max_batch <= GV_MAX_ROWS)) {  // by capacity, not by path (a width without GEMV instances runs strips at 1..8 too)
    ptts_engine_destroy(e);
    return ptts_fail(PTTS_E_UNSUPPORTED, "kv_fp8 (e4m3 self-attention cache) needs the bf16 engine created for more than %d utterances (the MFMA strip step)", GV_MAX_ROWS);
  }
  if (e->w8 && (c.dtype != PTTS_BF16 || H % 512 || F % 512 || H > 2048)) {
    ptts_engine_destroy(e);
    return ptts_fail(PTTS_E_UNSUPPORTED, "weights_fp8 needs the bf16 engine and hidden / ffn sizes that are multiples of 512 (hidden <= 2048)");
  }
#define A(expr) if ((rc = (expr)) != PTTS_OK) return fail(rc)
  for (int l = 0; l < c.num_layers; ++l) {
    LayerW& w = e->L[l];
    A(e->alloc_bytes(&w.qkv, (size_t)(H + 2 * e->nkv * 64) * H * es));
    A(e->alloc_bytes(&w.o, (size_t)H * H * es));
    A(e->alloc_bytes(&w.cq, (size_t)H * H * es));
    A(e->alloc_bytes(&w.ckv, (size_t)2 * e->nkc * 64 * H * es));
    A(e->alloc_bytes(&w.co, (size_t)H * H * es));
    A(e->alloc_bytes(&w.fc1, (size_t)F * H * es));
    A(e->alloc_bytes(&w.fc2, (size_t)F * H * es));
    if (e->use_gemv) {
      const size_t res = e->w8 ? 1 : es;  // row-major element size: e4m3 bytes or the engine dtype
      const int nq = H + 2 * e->nkv * 64;
      A(e->alloc_bytes(&w.qkv_rm, (size_t)nq * H * res)); A(e->alloc_bytes(&w.o_rm, (size_t)H * H * res));
      A(e->alloc_bytes(&w.cq_rm, (size_t)H * H * res)); A(e->alloc_bytes(&w.co_rm, (size_t)H * H * res));
      A(e->alloc_bytes(&w.fc1_rm, (size_t)F * H * res)); A(e->alloc_bytes(&w.fc2_rm, (size_t)F * H * res));
      if (e->xfold_ne) {
        A(e->alloc_bytes(&w.xM, (size_t)nh * e->xfold_ne * H * es)); A(e->alloc_bytes(&w.xU, (size_t)nh * e->xfold_ne * H * es));
      }
    }
    if (e->w8) {  // row scales serve both e4m3 layouts (row-major for the GEMV step, strips for the MFMA step)
      const int nq = H + 2 * e->nkv * 64;
      A(e->alloc(&w.qkv_sc, nq)); A(e->alloc(&w.o_sc, H)); A(e->alloc(&w.cq_sc, H)); A(e->alloc(&w.co_sc, H));
      A(e->alloc(&w.fc1_sc, F)); A(e->alloc(&w.fc2_sc, H));
      const char* qm[] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "encoder_attn.q_proj",
                          "encoder_attn.out_proj", "fc1", "fc2"};
      for (const char* m : qm) { char nm[160]; snprintf(nm, sizeof nm, "model.decoder.layers.%d.%s.weight", l, m); e->required_fp8.insert(nm); }
      if (e->w8_strips) {
        A(e->alloc_bytes(&w.qkv_p8, (size_t)nq * H)); A(e->alloc_bytes(&w.o_p8, (size_t)H * H)); A(e->alloc_bytes(&w.co_p8, (size_t)H * H));
        A(e->alloc_bytes(&w.fc1_p8, (size_t)F * H)); A(e->alloc_bytes(&w.fc2_p8, (size_t)F * H));
      }
    }
    A(e->alloc(&w.ln1_g, H)); A(e->alloc(&w.ln1_b, H)); A(e->alloc(&w.ln2_g, H)); A(e->alloc(&w.ln2_b, H));
    A(e->alloc(&w.ln3_g, H)); A(e->alloc(&w.ln3_b, H));
    const size_t kvs = (size_t)c.max_batch * e->nkv * c.max_ctx * 64 * (c.kv_fp8 ? 1 : es), kvc = (size_t)c.max_batch * e->nkc * c.max_enc * 64 * es;
    A(e->alloc_bytes(&w.k_self, kvs)); A(e->alloc_bytes(&w.v_self, kvs));
    if (c.kv_fp8) { A(e->alloc(&w.ks_self, (size_t)c.max_batch * e->nkv * c.max_ctx)); A(e->alloc(&w.vs_self, (size_t)c.max_batch * e->nkv * c.max_ctx)); }
    A(e->alloc_bytes(&w.k_cross, kvc)); A(e->alloc_bytes(&w.v_cross, kvc));
    char nm[160];
    const char* mats[] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "encoder_attn.q_proj",
                          "encoder_attn.k_proj", "encoder_attn.v_proj", "encoder_attn.out_proj", "fc1", "fc2"};
    for (const char* m : mats) { snprintf(nm, sizeof nm, "model.decoder.layers.%d.%s.weight", l, m); e->required.insert(nm); }
    const char* lns[] = {"self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"};
    for (const char* m : lns) {
      snprintf(nm, sizeof nm, "model.decoder.layers.%d.%s.weight", l, m); e->required.insert(nm);
      snprintf(nm, sizeof nm, "model.decoder.layers.%d.%s.bias", l, m); e->required.insert(nm);
    }
  }
  {
    std::vector<KvLayer> kl(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) kl[l] = KvLayer{e->L[l].ckv, e->L[l].k_cross, e->L[l].v_cross};
    A(e->alloc(&e->kv_layers, (size_t)c.num_layers));
    if (hipMemcpy(e->kv_layers, kl.data(), kl.size() * sizeof(KvLayer), hipMemcpyHostToDevice) != hipSuccess)
      return fail(ptts_fail(PTTS_E_HIP, "hipMemcpy(cross K/V operand table) failed"));
  }
  if (e->xfold_ne) {
    std::vector<FoldLayer> fl(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
      const LayerW& w = e->L[l];
      fl[l] = FoldLayer{w.cq_rm, w.cq_sc, w.k_cross, w.xM, w.co_rm, w.co_sc, w.v_cross, w.xU};
    }
    A(e->alloc(&e->fold_layers, (size_t)c.num_layers));
    if (hipMemcpy(e->fold_layers, fl.data(), fl.size() * sizeof(FoldLayer), hipMemcpyHostToDevice) != hipSuccess)
      return fail(ptts_fail(PTTS_E_HIP, "hipMemcpy(fold operand table) failed"));
  }
  A(e->alloc_bytes(&e->embed, (size_t)K * (V + 1) * H * es));
  A(e->alloc_bytes(&e->heads, (size_t)K * V * H * es));
  if (e->use_gemv) A(e->alloc_bytes(&e->heads_rm, (size_t)K * V * H * (e->w8 ? 1 : es)));
  if (e->w8_strips) A(e->alloc_bytes(&e->heads_p8, (size_t)K * V * H));
  if (e->w8) {
    A(e->alloc(&e->heads_sc, (size_t)K * V));
    for (int k = 0; k < K; ++k) { char nm[96]; snprintf(nm, sizeof nm, "lm_heads.%d.weight", k); e->required_fp8.insert(nm); }
  }
  A(e->alloc(&e->lnf_g, H)); A(e->alloc(&e->lnf_b, H));
  e->required.insert("model.decoder.layer_norm.weight");
  e->required.insert("model.decoder.layer_norm.bias");
  for (int k = 0; k < K; ++k) {
    char nm[96];
    snprintf(nm, sizeof nm, "model.decoder.embed_tokens.%d.weight", k); e->required.insert(nm);
    snprintf(nm, sizeof nm, "lm_heads.%d.weight", k); e->required.insert(nm);
  }
  if (c.rope) {
    // RoPE is computed for any position in the reference (:373-406): tables cover the whole KV capacity, not max_positions
    const size_t rope_rows = (size_t)std::max(c.max_positions, c.max_ctx);
    A(e->alloc(&e->rope_cos, rope_rows * 64)); A(e->alloc(&e->rope_sin, rope_rows * 64));
    e->required.insert("rope_cos"); e->required.insert("rope_sin");
  } else {
    A(e->alloc(&e->pos_table, (size_t)c.max_positions * H));
    e->required.insert("model.decoder.embed_positions.weights");
  }
  // split-KV factor: enough workgroups to cover the chip at small batch, none needed at large batch
  {
    int s = 256 / (c.max_batch * nh);
    if (s < 1) s = 1;
    if (s > 8) s = 8;
    while (s > 1 && (s - 1) * 4 * 8 * 8 >= c.max_ctx) --s;  // do not split below one 8-deep batch of row groups per wave
    e->S_self = s;
    e->S_cross = 1;
    if (const char* ev = ptts_dev_env("PTTS_ATTN_SPLITS")) e->S_self = std::max(1, std::min(8, atoi(ev)));  // tuning knobs (tools/)
    if (const char* ev = ptts_dev_env("PTTS_ATTN_WAVES")) e->attn_waves = atoi(ev) == 16 ? 16 : (atoi(ev) == 8 ? 8 : 4);
    {
      const int rows_per_wave = (c.dtype == PTTS_BF16 ? 8 : 4) * 8;  // RPI row groups x 8 loads in flight
      const int need = (c.max_enc + rows_per_wave - 1) / rows_per_wave;
      e->cross_waves = need <= 1 ? 1 : (need <= 2 ? 2 : 4);
      if (const char* ev = ptts_dev_env("PTTS_CROSS_WAVES")) e->cross_waves = atoi(ev) == 1 ? 1 : (atoi(ev) == 2 ? 2 : 4);
    }
    if (e->use_gemv) while (e->S_self & (e->S_self - 1)) --e->S_self;  // the GEMV combine prologue is instantiated for 2 / 4 / 8 splits
  }
  const size_t rows = (size_t)c.max_batch * e->max_prompt;
  const size_t enc_rows = (size_t)c.max_batch * c.max_enc;
  A(e->alloc(&e->h, rows * H));
  A(e->alloc(&e->qkv, rows * 3 * H));
  A(e->alloc(&e->qc, std::max(rows, enc_rows) * H));
  // qkv_attn_kernel: up to 8 splits + the new position's own slot for one utterance, up to 4 + 1 for each of 2..8
  const size_t part_rows = std::max(rows * (size_t)e->S_self, (size_t)c.max_batch * 5) + 9;
  A(e->alloc(&e->part, part_rows * H));
  A(e->alloc(&e->stats, part_rows * nh * 2));
  A(e->alloc(&e->xpart, (size_t)nh * H));
  A(e->alloc(&e->h2, (size_t)H));
  if (const char* ev = getenv("PTTS_FUSE_X")) e->fuse_x = atoi(ev) ? 1 : 0;
  if (const char* ev = getenv("PTTS_FUSE_X_NUR")) { const int v = atoi(ev); if (v == 2 || v == 4) e->fuse_x_nur = v; }
  e->fuse_qa = !(getenv("PTTS_NO_FUSE_QA") && atoi(getenv("PTTS_NO_FUSE_QA")));
  if (const char* ev = getenv("PTTS_FUSE_QA_MAX")) e->fuse_qa_max = std::max(1, std::min(GV_MAX_ROWS, atoi(ev)));
  e->fuse_xq = !(getenv("PTTS_NO_FUSE_XQ") && atoi(getenv("PTTS_NO_FUSE_XQ")));
  if (const char* ev = ptts_dev_env("PTTS_FUSE_QA_S")) { const int v = atoi(ev); if (v == 1 || v == 2 || v == 4 || v == 8) e->fuse_qa_s = v; }
  A(e->alloc(&e->ffn, std::max(rows * F, rows * (size_t)H)));
  A(e->alloc(&e->logits, (size_t)c.max_batch * K * V));
  A(e->alloc(&e->sort_buf, 16));
  A(e->alloc_bytes(&e->xw, (rows + 16) * H * es));  // + 16 rows: fragment order addresses whole 16-row tiles
  A(e->alloc_bytes(&e->xw2, std::max((rows + 16) * F, enc_rows * (size_t)H) * es));
  e->use_fo = !(ptts_dev_env("PTTS_NO_FO") && atoi(ptts_dev_env("PTTS_NO_FO")));
  A(e->alloc(&e->lnstat, (size_t)c.max_batch * (H / 16) * 2 + 16));
  e->use_lns = (H == 1024 || H == 1536) && !(ptts_dev_env("PTTS_NO_LNS") && atoi(ptts_dev_env("PTTS_NO_LNS")));
  if (const char* ev = ptts_dev_env("PTTS_XATTN_GROUPS_MAX")) e->xattn_groups_max = std::max(8, atoi(ev));
  if (const char* ev = getenv("PTTS_LNPROJ")) e->lnproj = std::max(0, std::min(3, atoi(ev)));
  if (const char* ev = getenv("PTTS_LNPROJ_G")) e->lnproj_g = atoi(ev) == 4 ? 4 : (atoi(ev) == 16 ? 16 : (atoi(ev) == 8 ? 8 : 0));
  e->xattn_g_ok = (H == 1024 && ((H / (c.dtype == PTTS_BF16 ? 32 : 16)) / 2) % 16 == 0) || H == 1536;
  if (const char* ev = getenv("PTTS_XATTN_G")) { const int g = atoi(ev); if (g == 2 || g == 4 || g == 8) e->xattn_g = e->xattn_g_ok ? g : 8; }
  e->xattn_groups = !(getenv("PTTS_NO_XATTN_GROUPS") && atoi(getenv("PTTS_NO_XATTN_GROUPS")));  // measured: 1386 -> 1360 us per batch-32 step (profiles/r03_experiments.txt)
  A(e->alloc(&e->prefix, (size_t)c.max_batch * K * c.max_ctx));
  e->ids_ld = c.max_ctx + 8;
  A(e->alloc(&e->ids, (size_t)c.max_batch * K * e->ids_ld));
  A(e->alloc(&e->cur_len, c.max_batch)); A(e->alloc(&e->unfinished, (size_t)c.max_batch * K));
  A(e->alloc(&e->has_eos, (size_t)c.max_batch * K)); A(e->alloc(&e->first_unf, c.max_batch));
  A(e->alloc(&e->enc_mask, enc_rows + 64));  // + 64: the folded cross block reads the mask in int4 groups up to position 63 whatever max_enc is
  A(e->alloc(&e->prompt_mask, rows));
  A(e->alloc(&e->dims, 1)); A(e->alloc(&e->gen, 1));
#undef A
  if (hipHostMalloc((void**)&e->host_pinned, ((size_t)c.max_batch * K + 16) * 4) != hipSuccess) return fail(ptts_fail(PTTS_E_HIP, "hipHostMalloc failed"));
#ifdef PTTS_TIMING
  {
    const size_t n = (size_t)(c.num_layers + 1) * 7 * 48;  // 5 nodes per layer of the GEMV step, 7 of the batch > 8 step
    rc = e->alloc(&e->dbg_stamps, n);
    if (rc != PTTS_OK) return fail(rc);
    hipMemset(e->dbg_stamps, 0, n * sizeof(long long));
  }
#endif
  ptts_gen_params gp = {};
  gp.max_length = c.max_ctx; gp.temperature = 1.f; gp.top_p = 1.f; gp.use_eos_gate = 1;
  e->gp = gp;
  *out = e;
  return PTTS_OK;
}

extern "C" void ptts_engine_destroy(ptts_engine* e) {
  if (!e) return;
  PttsDeviceGuard _dg(e->cfg.device);
  hipDeviceSynchronize();
  for (auto& kv : e->graphs) hipGraphExecDestroy(kv.second);
  for (auto& kv : e->prefill_graphs) hipGraphExecDestroy(kv.second);
  for (void* p : e->allocs) hipFree(p);
  if (e->host_pinned) hipHostFree(e->host_pinned);
  if (e->own_stream) hipStreamDestroy(e->own_stream);
  if (e->fold_stream) hipStreamDestroy(e->fold_stream);
  if (e->ev_kv) hipEventDestroy(e->ev_kv);
  if (e->ev_fold) hipEventDestroy(e->ev_fold);
  if (e->ev_first) hipEventDestroy(e->ev_first);
  if (e->ev_tail0) hipEventDestroy(e->ev_tail0);
  if (e->ev_pre0) hipEventDestroy(e->ev_pre0);
  delete e;
}

extern "C" int ptts_load_weight(ptts_engine* e, const char* name_c, const void* dev_ptr, int32_t src_dtype, const int64_t* shape,
                                int32_t ndim, void* stream) {
  PTTS_CHECK(e && name_c && dev_ptr && shape, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(src_dtype == PTTS_F32 || src_dtype == PTTS_BF16, PTTS_E_INVALID, "src_dtype must be f32 or bf16");
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = pick_stream(e, stream);
  const ptts_config& c = e->cfg;
  const int H = c.hidden_size, F = c.ffn_dim, K = c.num_codebooks, V = c.vocab_size;
  const std::string name(name_c);
  auto want = [&](int64_t a, int64_t b) -> int {
    if (b < 0) { if (ndim != 1 || shape[0] != a) return ptts_fail(PTTS_E_INVALID, "%s: expected shape [%lld]", name_c, (long long)a); }
    else if (ndim != 2 || shape[0] != a || shape[1] != b) return ptts_fail(PTTS_E_INVALID, "%s: expected shape [%lld, %lld]", name_c, (long long)a, (long long)b);
    return PTTS_OK;
  };
  int l = -1, k = -1;
  char tail[128] = {0};
  if (sscanf(name_c, "model.decoder.layers.%d.%127s", &l, tail) == 2) {
    PTTS_CHECK(l >= 0 && l < c.num_layers, PTTS_E_INVALID, "%s: layer index out of range", name_c);
    LayerW& w = e->L[l];
    const std::string t(tail);
    struct { const char* n; void* dst; void* rm; int N, Kd, row0; } mats[] = {
        {"self_attn.q_proj.weight", w.qkv, w.qkv_rm, H, H, 0},       {"self_attn.k_proj.weight", w.qkv, w.qkv_rm, e->nkv * 64, H, H},
        {"self_attn.v_proj.weight", w.qkv, w.qkv_rm, e->nkv * 64, H, H + e->nkv * 64}, {"self_attn.out_proj.weight", w.o, w.o_rm, H, H, 0},
        {"encoder_attn.q_proj.weight", w.cq, w.cq_rm, H, H, 0},     {"encoder_attn.k_proj.weight", w.ckv, nullptr, e->nkc * 64, H, 0},
        {"encoder_attn.v_proj.weight", w.ckv, nullptr, e->nkc * 64, H, e->nkc * 64}, {"encoder_attn.out_proj.weight", w.co, w.co_rm, H, H, 0},
        {"fc1.weight", w.fc1, w.fc1_rm, F, H, 0},                    {"fc2.weight", w.fc2, w.fc2_rm, H, F, 0}};
    for (auto& m : mats)
      if (t == m.n) {
        PTTS_TRY(want(m.N, m.Kd));
        PTTS_TRY(pack_dispatch(e, m.dst, dev_ptr, src_dtype, m.N, m.Kd, m.row0, st));
        if (m.rm && !e->w8) PTTS_TRY(rowmajor_dispatch(e, m.rm, dev_ptr, src_dtype, m.N, m.Kd, m.row0, st));
        e->loaded.insert(name);
        return PTTS_OK;
      }
    struct { const char* n; float* dst; } vecs[] = {
        {"self_attn_layer_norm.weight", w.ln1_g}, {"self_attn_layer_norm.bias", w.ln1_b},
        {"encoder_attn_layer_norm.weight", w.ln2_g}, {"encoder_attn_layer_norm.bias", w.ln2_b},
        {"final_layer_norm.weight", w.ln3_g}, {"final_layer_norm.bias", w.ln3_b}};
    for (auto& v : vecs)
      if (t == v.n) {
        PTTS_TRY(want(H, -1));
        PTTS_TRY(convert_into<float>(v.dst, dev_ptr, src_dtype, H, st));
        e->loaded.insert(name);
        return PTTS_OK;
      }
    return ptts_fail(PTTS_E_INVALID, "unknown tensor name %s", name_c);
  }
  if (sscanf(name_c, "model.decoder.embed_tokens.%d.weight", &k) == 1) {
    PTTS_CHECK(k >= 0 && k < K, PTTS_E_INVALID, "%s: codebook index out of range", name_c);
    PTTS_TRY(want(V + 1, H));
    char* dst = (char*)e->embed + (size_t)k * (V + 1) * H * e->esize;
    if (c.dtype == PTTS_BF16) PTTS_TRY(convert_into<bf16_t>((bf16_t*)dst, dev_ptr, src_dtype, (size_t)(V + 1) * H, st));
    else PTTS_TRY(convert_into<float>((float*)dst, dev_ptr, src_dtype, (size_t)(V + 1) * H, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  if (sscanf(name_c, "lm_heads.%d.weight", &k) == 1) {
    PTTS_CHECK(k >= 0 && k < K, PTTS_E_INVALID, "%s: codebook index out of range", name_c);
    PTTS_TRY(want(V, H));
    PTTS_TRY(pack_dispatch(e, e->heads, dev_ptr, src_dtype, V, H, k * V, st));
    if (e->heads_rm && !e->w8) PTTS_TRY(rowmajor_dispatch(e, e->heads_rm, dev_ptr, src_dtype, V, H, k * V, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  if (name == "lm_heads.weight") {  // use_fused_lm_heads :1834-1840
    PTTS_TRY(want((int64_t)K * V, H));
    PTTS_TRY(pack_dispatch(e, e->heads, dev_ptr, src_dtype, K * V, H, 0, st));
    if (e->heads_rm && !e->w8) PTTS_TRY(rowmajor_dispatch(e, e->heads_rm, dev_ptr, src_dtype, K * V, H, 0, st));
    for (int i = 0; i < K; ++i) { char nm[64]; snprintf(nm, sizeof nm, "lm_heads.%d.weight", i); e->loaded.insert(nm); }
    return PTTS_OK;
  }
  if (name == "model.decoder.layer_norm.weight" || name == "model.decoder.layer_norm.bias") {
    PTTS_TRY(want(H, -1));
    PTTS_TRY(convert_into<float>(ends_with(name, ".weight") ? e->lnf_g : e->lnf_b, dev_ptr, src_dtype, H, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  if (name == "model.decoder.embed_positions.weights") {
    PTTS_CHECK(!c.rope, PTTS_E_INVALID, "%s given but rope_embeddings is set", name_c);
    PTTS_CHECK(ndim == 2 && shape[1] == H && shape[0] >= 1 && shape[0] <= c.max_positions, PTTS_E_INVALID, "%s: expected [<=%d, %d]", name_c, c.max_positions, H);
    PTTS_TRY(convert_into<float>(e->pos_table, dev_ptr, src_dtype, (size_t)shape[0] * H, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  if (name == "rope_cos" || name == "rope_sin") {  // fp32 tables as ParlerTTSRotaryEmbedding.forward computes them (:373-406)
    PTTS_CHECK(c.rope, PTTS_E_INVALID, "%s given but rope_embeddings is off", name_c);
    const int rope_rows = std::max(c.max_positions, c.max_ctx);
    PTTS_CHECK(ndim == 2 && shape[1] == 64 && shape[0] >= c.max_ctx && shape[0] <= rope_rows, PTTS_E_INVALID, "%s: expected [%d..%d, 64] (one row per KV position)", name_c, c.max_ctx, rope_rows);
    PTTS_TRY(convert_into<float>(name == "rope_cos" ? e->rope_cos : e->rope_sin, dev_ptr, src_dtype, (size_t)shape[0] * 64, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  return ptts_fail(PTTS_E_INVALID, "unknown tensor name %s", name_c);
}

extern "C" int ptts_weights_ready(ptts_engine* e) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  std::string missing;
  int n = 0;
  for (const auto& r : e->required)
    if (!e->loaded.count(r)) { if (n++ < 8) missing += (missing.empty() ? "" : ", ") + r; }
  if (n) return ptts_fail(PTTS_E_MISSING, "%d tensors not loaded: %s%s", n, missing.c_str(), n > 8 ? ", ..." : "");
  for (const auto& r : e->required_fp8)
    if (!e->loaded_fp8.count(r)) { if (n++ < 8) missing += (missing.empty() ? "" : ", ") + r; }
  if (n) return ptts_fail(PTTS_E_MISSING, "%d e4m3 weight copies not loaded (ptts_load_weight_fp8): %s%s", n, missing.c_str(), n > 8 ? ", ..." : "");
  return PTTS_OK;
}

// weights_fp8 engines: the OCP e4m3 bytes + per-row power-of-two scales of one projection matrix, quantised by the caller
// (parler_tts_amd/quant.py) from the SAME tensor whose exact bf16 dequantisation went through ptts_load_weight.
extern "C" int ptts_load_weight_fp8(ptts_engine* e, const char* name_c, const uint8_t* q_dev, const float* scale_dev, const int64_t* shape,
                                    int32_t ndim, void* stream) {
  PTTS_CHECK(e && name_c && q_dev && scale_dev && shape, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(e->w8, PTTS_E_INVALID, "ptts_load_weight_fp8 on an engine created without weights_fp8");
  PTTS_CHECK(ndim == 2, PTTS_E_INVALID, "%s: expected a matrix", name_c);
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = pick_stream(e, stream);
  const ptts_config& c = e->cfg;
  const int H = c.hidden_size, F = c.ffn_dim, K = c.num_codebooks, V = c.vocab_size;
  uint8_t *dst = nullptr, *p8 = nullptr;
  float* sc = nullptr;
  bool known = false;
  int N = 0, Kd = 0, row0 = 0, l = -1, k = -1;
  char tail[128] = {0};
  if (sscanf(name_c, "model.decoder.layers.%d.%127s", &l, tail) == 2) {
    PTTS_CHECK(l >= 0 && l < c.num_layers, PTTS_E_INVALID, "%s: layer index out of range", name_c);
    LayerW& w = e->L[l];
    const std::string t(tail);
    struct { const char* n; void* rm; void* p8; float* sc; int N, Kd, row0; } mats[] = {
        {"self_attn.q_proj.weight", w.qkv_rm, w.qkv_p8, w.qkv_sc, H, H, 0}, {"self_attn.k_proj.weight", w.qkv_rm, w.qkv_p8, w.qkv_sc, e->nkv * 64, H, H},
        {"self_attn.v_proj.weight", w.qkv_rm, w.qkv_p8, w.qkv_sc, e->nkv * 64, H, H + e->nkv * 64}, {"self_attn.out_proj.weight", w.o_rm, w.o_p8, w.o_sc, H, H, 0},
        {"encoder_attn.q_proj.weight", w.cq_rm, nullptr, w.cq_sc, H, H, 0}, {"encoder_attn.out_proj.weight", w.co_rm, w.co_p8, w.co_sc, H, H, 0},
        {"fc1.weight", w.fc1_rm, w.fc1_p8, w.fc1_sc, F, H, 0}, {"fc2.weight", w.fc2_rm, w.fc2_p8, w.fc2_sc, H, F, 0}};
    for (auto& m : mats)
      if (t == m.n) { dst = (uint8_t*)m.rm; p8 = (uint8_t*)m.p8; sc = m.sc; N = m.N; Kd = m.Kd; row0 = m.row0; known = true; }
  } else if (sscanf(name_c, "lm_heads.%d.weight", &k) == 1) {
    PTTS_CHECK(k >= 0 && k < K, PTTS_E_INVALID, "%s: codebook index out of range", name_c);
    dst = (uint8_t*)e->heads_rm; p8 = (uint8_t*)e->heads_p8; sc = e->heads_sc; N = V; Kd = H; row0 = k * V; known = true;
  }
  PTTS_CHECK(known && sc, PTTS_E_INVALID, "%s has no e4m3 copy (only the decode-step projection matrices do)", name_c);
  PTTS_CHECK(shape[0] == N && shape[1] == Kd, PTTS_E_INVALID, "%s: expected shape [%d, %d]", name_c, N, Kd);
  // row-major bytes for the GEMV step (engines of <= GV_MAX_ROWS utterances), strip order for the MFMA step (wider engines)
  if (dst) PTTS_HIP(hipMemcpyAsync(dst + (size_t)row0 * Kd, q_dev, (size_t)N * Kd, hipMemcpyDeviceToDevice, st));
  if (p8) {
    PTTS_CHECK(N % 16 == 0 && Kd % 64 == 0 && row0 % 16 == 0, PTTS_E_INVALID, "%s: [%d, %d] at row %d is not a whole number of 16 x 64 e4m3 strips", name_c, N, Kd, row0);
    const size_t total = (size_t)(N / 16) * (Kd / 64) * 64;
    hipLaunchKernelGGL(pack_w8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q_dev, reinterpret_cast<uint4*>(p8), N, Kd, row0 / 16, Kd / 64);
  }
  PTTS_HIP(hipMemcpyAsync(sc + row0, scale_dev, (size_t)N * 4, hipMemcpyDeviceToDevice, st));
  e->loaded_fp8.insert(name_c);
  return PTTS_OK;
}

extern "C" int ptts_set_gen_params(ptts_engine* e, const ptts_gen_params* gp) {
  PTTS_CHECK(e && gp, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(gp->max_length >= 2, PTTS_E_INVALID, "max_length must be >= 2 (BOS column + 1 token)");
  PTTS_CHECK(gp->max_length <= e->cfg.max_ctx, PTTS_E_CAPACITY, "max_length %d exceeds engine max_ctx %d", gp->max_length, e->cfg.max_ctx);
  PTTS_CHECK(!gp->do_sample || gp->temperature > 0.f, PTTS_E_INVALID, "temperature must be > 0");
  PTTS_CHECK(gp->top_p > 0.f && gp->top_p <= 1.f, PTTS_E_INVALID, "top_p must be in (0, 1]");
  PTTS_CHECK(gp->top_k >= 0, PTTS_E_INVALID, "top_k must be >= 0");
  e->gp = *gp;
  return PTTS_OK;
}

extern "C" int ptts_set_audio_prefix(ptts_engine* e, const int64_t* codes_dev, int32_t B, int32_t T, void* stream) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  const ptts_config& c = e->cfg;
  PTTS_CHECK(T >= 0 && (T == 0 || codes_dev), PTTS_E_INVALID, "bad audio prefix");
  PTTS_CHECK(B >= 1 && B <= c.max_batch, PTTS_E_CAPACITY, "batch %d exceeds engine max_batch %d", B, c.max_batch);
  PTTS_CHECK(T + 2 <= c.max_ctx, PTTS_E_CAPACITY, "voice prompt of %d frames exceeds engine max_ctx %d", T, c.max_ctx);
  PTTS_DEVICE(c.device);
  if (T > 0)
    PTTS_HIP(hipMemcpy2DAsync(e->prefix, (size_t)c.max_ctx * 8, codes_dev, (size_t)T * 8, (size_t)T * 8, (size_t)B * c.num_codebooks,
                              hipMemcpyDeviceToDevice, pick_stream(e, stream)));
  e->pending_T = T;
  return PTTS_OK;
}

static int precapture_graphs(ptts_engine* e, int max_buckets);

extern "C" int ptts_prefill(ptts_engine* e, const float* enc_dev, const int32_t* enc_mask_dev, const float* prompt_dev,
                            const int32_t* prompt_mask_dev, int32_t B, int32_t N, int32_t P, int32_t sample, void* stream) {
  PTTS_CHECK(e && enc_dev, PTTS_E_INVALID, "null argument");
  PTTS_TRY(ptts_weights_ready(e));
  const ptts_config& c = e->cfg;
  PTTS_CHECK(B >= 1 && B <= c.max_batch, PTTS_E_CAPACITY, "batch %d exceeds engine max_batch %d", B, c.max_batch);
  PTTS_CHECK(N >= 1 && N <= c.max_enc, PTTS_E_CAPACITY, "encoder length %d exceeds engine max_enc %d", N, c.max_enc);
  PTTS_CHECK(P >= 0 && P + 1 <= e->max_prompt, PTTS_E_CAPACITY, "prompt length %d exceeds engine capacity %d", P, e->max_prompt - 1);
  PTTS_CHECK(P == 0 || prompt_dev, PTTS_E_INVALID, "prompt_dev is null but P > 0");
  PTTS_CHECK(e->pending_T + 2 <= e->gp.max_length, PTTS_E_INVALID, "voice prompt of %d frames leaves no room below max_length %d", e->pending_T, e->gp.max_length);
  PTTS_CHECK(P + e->gp.max_length <= c.max_ctx, PTTS_E_CAPACITY, "P + max_length = %d exceeds engine max_ctx %d", P + e->gp.max_length, c.max_ctx);
  PTTS_CHECK(P + e->gp.max_length <= c.max_positions || c.rope, PTTS_E_CAPACITY, "P + max_length = %d exceeds max_position_embeddings %d", P + e->gp.max_length, c.max_positions);
  PTTS_DEVICE(c.device);
  hipStream_t st = pick_stream(e, stream);
  const int H = c.hidden_size, K = c.num_codebooks;
  e->B = B; e->N = N; e->P = P;
  if (sample) PTTS_HIP(hipEventRecord(e->ev_pre0, st));
  // per-call device params travel as kernel arguments (no host staging buffer to keep alive)
  {
    DevDims hd; hd.P = P; hd.N = N; hd.max_length = e->gp.max_length;
    hd.T_prefix = e->pending_T; hd.prefix = e->prefix; hd.prefix_ld = c.max_ctx;
    DevGen hg; hg.max_length = e->gp.max_length; hg.min_new_tokens = e->gp.min_new_tokens; hg.do_sample = e->gp.do_sample;
    hg.top_k = e->gp.top_k; hg.use_eos_gate = e->gp.use_eos_gate; hg.temperature = e->gp.temperature; hg.top_p = e->gp.top_p;
    hg.seed = e->gp.seed;
    hipLaunchKernelGGL(set_params_kernel, dim3(1), dim3(1), 0, st, e->dims, e->gen, hd, hg);
  }
  // masks (all-ones when absent so the captured graph never changes shape)
  if (enc_mask_dev) PTTS_HIP(hipMemcpy2DAsync(e->enc_mask, (size_t)c.max_enc * 4, enc_mask_dev, (size_t)N * 4, (size_t)N * 4, B, hipMemcpyDeviceToDevice, st));
  else hipLaunchKernelGGL(fill_int_kernel, dim3(64), dim3(256), 0, st, e->enc_mask, 1, (size_t)B * c.max_enc);
  if (prompt_mask_dev && P > 0) PTTS_HIP(hipMemcpy2DAsync(e->prompt_mask, (size_t)e->max_prompt * 4, prompt_mask_dev, (size_t)P * 4, (size_t)P * 4, B, hipMemcpyDeviceToDevice, st));
  else hipLaunchKernelGGL(fill_int_kernel, dim3(64), dim3(256), 0, st, e->prompt_mask, 1, (size_t)B * e->max_prompt);
  hipLaunchKernelGGL(reset_state_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, e->ids, e->ids_ld, e->cur_len, e->unfinished,
                     e->has_eos, e->first_unf, B, K, c.bos_token_id);
  // stage inputs: encoder states -> qc (consumed by the cross K/V projection), prompt embeddings -> ffn
  PTTS_HIP(hipMemcpyAsync(e->qc, enc_dev, (size_t)B * N * H * 4, hipMemcpyDeviceToDevice, st));
  if (P > 0) PTTS_HIP(hipMemcpyAsync(e->ffn, prompt_dev, (size_t)B * P * H * 4, hipMemcpyDeviceToDevice, st));
  // voice prompt (ptts_set_audio_prefix): the reference runs BOS + the T given code columns in ONE multi-column forward
  // (:3136-3194, causal attention); so does this prefill whenever the row capacity (max_prompt) holds P + 1 + T positions
  // per utterance. Otherwise the T columns are teacher-forced one position at a time through the decode path (same numbers:
  // causal attention over the same cache).
  const int T = e->pending_T;
  e->pending_T = 0;
  const bool batched = T > 0 && P + 1 + T <= e->max_prompt && !(ptts_dev_env("PTTS_NO_BATCHED_PREFIX") && atoi(ptts_dev_env("PTTS_NO_BATCHED_PREFIX")));
  if (batched) {
    hipLaunchKernelGGL(push_prefix_all_kernel, dim3((B * K * T + 255) / 256), dim3(256), 0, st, e->ids, e->ids_ld, e->dims, T, B, K, c.bos_token_id);
    e->prefill_T = T;
  }
  e->xfold_valid = false;  // the prefill itself (and anything before the fold below) runs the un-folded cross-attention
  // The prefill forward (~290 launches of 4-8 us kernels for one utterance) replayed from ONE captured graph per (batch, description, prompt,
  // voice-prompt) shape: every operand is an engine-owned buffer and every per-call value (lengths, masks) is device-resident, so the
  // launch list is a pure function of the shape. Measured (profiles/r05_experiments.txt call 1): time to the first token 2.420 ms with the
  // graph, 2.401 ms with eager launches - behind the description encoder's 1 ms of GPU time the host is ahead of the GPU either way, and a
  // graph costs a capture + instantiate per new (batch, description, prompt) shape. OFF by default; PTTS_PREFILL_GRAPH=1 enables it (A/B).
  static const bool prefill_graph = ptts_dev_env("PTTS_PREFILL_GRAPH") && atoi(ptts_dev_env("PTTS_PREFILL_GRAPH"));
  int rc_fwd = PTTS_OK;
  if (prefill_graph) {
    const long long key = (long long)B | ((long long)N << 12) | ((long long)P << 28) | ((long long)e->prefill_T << 44);
    auto it = e->prefill_graphs.find(key);
    hipGraphExec_t ex = nullptr;
    if (it != e->prefill_graphs.end()) {
      ex = it->second;
    } else {
      hipGraph_t g = nullptr;
      PTTS_HIP(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
      e->in_capture = true;
      rc_fwd = forward_dispatch(e, true, e->own_stream);
      e->in_capture = false;
      hipError_t ce = hipStreamEndCapture(e->own_stream, &g);
      if (rc_fwd != PTTS_OK) { if (g) hipGraphDestroy(g); e->prefill_T = 0; return rc_fwd; }
      if (ce != hipSuccess) { e->prefill_T = 0; return ptts_fail(PTTS_E_HIP, "hipStreamEndCapture(prefill) failed: %s", hipGetErrorString(ce)); }
      hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (ie != hipSuccess) { e->prefill_T = 0; return ptts_fail(PTTS_E_HIP, "hipGraphInstantiate(prefill) failed: %s", hipGetErrorString(ie)); }
      if (e->prefill_graphs.size() >= 32) {  // bounded: a server cycling through many prompt lengths re-captures instead of growing
        for (auto& kv : e->prefill_graphs) hipGraphExecDestroy(kv.second);
        e->prefill_graphs.clear();
      }
      e->prefill_graphs[key] = ex;
    }
    hipError_t le = hipGraphLaunch(ex, st);
    if (le != hipSuccess) { e->prefill_T = 0; return ptts_fail(PTTS_E_HIP, "hipGraphLaunch(prefill) failed: %s", hipGetErrorString(le)); }
    if (e->xfold_ne && B == 1) hipEventRecord(e->ev_kv, st);
  } else {
    rc_fwd = forward_dispatch(e, true, st);
  }
  e->prefill_T = 0;
  PTTS_TRY(rc_fwd);
  e->kv_ub = P + 1 + (batched ? T : 0);  // self-KV positions written by this pass
  if (batched) hipLaunchKernelGGL(set_len_kernel, dim3((B + 255) / 256), dim3(256), 0, st, e->cur_len, B, T + 1);
  for (int j = 1; j <= (batched ? 0 : T); ++j) {  // (un-folded cross block: the fold below has not run yet)
    hipLaunchKernelGGL(push_prefix_col_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, e->ids, e->ids_ld, e->dims, j, B, K, c.bos_token_id);
    hipLaunchKernelGGL(set_len_kernel, dim3((B + 255) / 256), dim3(256), 0, st, e->cur_len, B, j + 1);
    advance_kv(e);
    PTTS_TRY(forward_dispatch(e, false, st, true));
  }
  if (sample) { PTTS_HIP(hipEventRecord(e->ev_tail0, st)); PTTS_TRY(launch_tail(e, st, true)); }  // also embeds the sampled column for the first decode step
  e->first_recorded = false;
  if (sample) { PTTS_HIP(hipEventRecord(e->ev_first, st)); e->first_recorded = true; }
  if (e->xfold_ne && B == 1) {
    // The fold only needs the cross K/V cache. It is enqueued AFTER the sampler tail, so the first token is materialised before the
    // fold's two launches run (time-to-first-token does not pay for it); every decode step that follows on `st` sees the folded
    // matrices by stream order. (Round 2 ran 48 per-layer launches BEFORE the tail: +0.8 ms of time-to-first-token. Running it on a
    // second stream beside the prefill was measured and lost: 8.98 vs 8.44 ms to the first token, and the first streamed chunk 23 ms
    // later, the decode steps waiting on the second hardware queue; PTTS_FOLD_ASYNC=1 keeps that variant for A/B.)
    static const bool sync_fold = !(ptts_dev_env("PTTS_FOLD_ASYNC") && atoi(ptts_dev_env("PTTS_FOLD_ASYNC")));
    hipStream_t fs = sync_fold ? st : e->fold_stream;
    if (!sync_fold) PTTS_HIP(hipStreamWaitEvent(fs, e->ev_kv, 0));
    if (c.dtype == PTTS_F32) PTTS_TRY((fold_cross<float, false>(e, fs)));
    else if (e->w8) PTTS_TRY((fold_cross<bf16_t, true>(e, fs)));
    else PTTS_TRY((fold_cross<bf16_t, false>(e, fs)));
    if (!sync_fold) { PTTS_HIP(hipEventRecord(e->ev_fold, fs)); PTTS_HIP(hipStreamWaitEvent(st, e->ev_fold, 0)); }
    e->xfold_valid = true;
  }
  e->h_ready = sample != 0;
  e->prefilled = true;
  if (sample) PTTS_TRY(precapture_graphs(e, 3));  // device loop ahead: the step graphs of the next three 64-position buckets (the rest: ptts_decode_steps, ahead of the GPU)
  return PTTS_OK;
}

// The decode step (170 kernel nodes for Mini-v1 at batch <= 8) is captured ONCE per batch size on the engine's private stream
// (capture records, it does not execute; the legacy NULL stream cannot be captured) and replayed into the caller's.
static int get_graph(ptts_engine* e, hipGraphExec_t* out) {
  // the node set of the step depends on the batch size and on the folded cross block; the attention fetch bound (a kernel argument)
  // on the 64-position bucket of the context (forward<> reads it from e->kv_bound while capturing). (Several steps of one bucket per
  // graph launch - round 4's PTTS_GRAPH_STEPS - measured 0.3-0.7 %, profiles/r04_experiments.txt call 19: removed in round 6.)
  const long long key = e->B * 2 + (e->xfold_valid ? 1 : 0) + 4096LL * (e->kv_bound / 64);
  auto it = e->graphs.find(key);
  if (it != e->graphs.end()) { if (out) *out = it->second; return PTTS_OK; }
  hipGraph_t g = nullptr;
  hipStream_t st = e->own_stream;
  PTTS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  // decode step = layers + heads + tail; the tail embeds the column it just sampled for the NEXT replay (no embed node)
  int rc = forward_dispatch(e, false, st, false);
  if (rc == PTTS_OK) rc = launch_tail(e, st, true);
  hipError_t ce = hipStreamEndCapture(st, &g);
  if (rc != PTTS_OK) { if (g) hipGraphDestroy(g); return rc; }
  if (ce != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
  hipGraphExec_t ex = nullptr;
  {
    size_t nn = 0;
    if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) e->last_graph_nodes = (int)nn;
  }
  hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (ie != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
  e->graphs[key] = ex;
  if (out) *out = ex;
  return PTTS_OK;
}

// The step graphs of the next `max_buckets` 64-position buckets this call can still reach are captured AHEAD of the GPU: at prefill
// (host work that overlaps the prefill kernels already enqueued) and again at the end of every ptts_decode_steps (behind the launches it
// has just enqueued), never in the middle of a run of launches: a first long utterance or stream would otherwise stall for a capture +
// instantiate of ~170-270 nodes every 64 frames, on the latency-critical streaming path. Bounded: the stock generation
// config (max_length 2580) is ~41 buckets, three times what a call that stops on EOS needs, and all of them sat between the prefill and
// the first decode_steps of the first call of every batch size. Graphs are cached for the engine's life, so only the first call of a
// (batch, fold, bucket) pays. PTTS_NO_PRECAPTURE=1: capture lazily, at the launch that needs the graph (A/B).
static int precapture_graphs(ptts_engine* e, int max_buckets) {
  static const bool off = ptts_dev_env("PTTS_NO_PRECAPTURE") && atoi(ptts_dev_env("PTTS_NO_PRECAPTURE"));
  if (off) return PTTS_OK;
  const int cap = e->cfg.max_ctx, saved_ub = e->kv_ub, saved_bound = e->kv_bound;
  const int last_ub = std::min(cap - 1, e->P + e->gp.max_length);  // positions the longest run of this call writes
  int rc = PTTS_OK, done = 0;
  for (int ub = saved_ub + 1; ub <= last_ub && rc == PTTS_OK && done < max_buckets; ++done) {
    e->kv_bound = g_no_kv_bound ? cap : std::min(cap, (ub + 1 + 63) / 64 * 64);
    rc = get_graph(e, nullptr);
    ub = e->kv_bound;  // first upper bound of the next bucket: (ub + 1 + 63) / 64 * 64 > kv_bound
    if (g_no_kv_bound) break;
  }
  e->kv_ub = saved_ub; e->kv_bound = saved_bound;
  return rc;
}

extern "C" int ptts_decode_steps(ptts_engine* e, int32_t n_steps, void* stream) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  PTTS_CHECK(e->prefilled, PTTS_E_INVALID, "ptts_decode_steps called before ptts_prefill");
  PTTS_CHECK(n_steps >= 0, PTTS_E_INVALID, "n_steps < 0");
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = pick_stream(e, stream);
  if (n_steps > 0 && !e->h_ready) {  // previous column came from ptts_push_tokens / an un-sampled prefill: embed it once, eagerly
    advance_kv(e);
    PTTS_TRY(forward_dispatch(e, false, st, true));
    PTTS_TRY(launch_tail(e, st, true));
    e->h_ready = true;
    --n_steps;
  }
  const int cap = e->cfg.max_ctx;
  for (int i = 0; i < n_steps;) {
    hipGraphExec_t ex = nullptr;
    advance_kv(e);
    PTTS_TRY(get_graph(e, &ex));  // cached per (batch, fold, 64-position bucket)
    PTTS_HIP(hipGraphLaunch(ex, st));
    ++i;
  }
  if (n_steps > 0) PTTS_TRY(precapture_graphs(e, 2));  // the current bucket and the next one, while the GPU works through what was just enqueued
  return PTTS_OK;
}

extern "C" int ptts_first_token_sync(ptts_engine* e) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  PTTS_CHECK(e->prefilled && e->first_recorded, PTTS_E_INVALID, "ptts_first_token_sync: no sampling prefill to wait for");
  PTTS_DEVICE(e->cfg.device);
  PTTS_HIP(hipEventSynchronize(e->ev_first));
  return PTTS_OK;
}

extern "C" int ptts_first_token_times(ptts_engine* e, float* prefill_ms, float* tail_ms) {
  PTTS_CHECK(e && prefill_ms && tail_ms, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(e->prefilled && e->first_recorded, PTTS_E_INVALID, "ptts_first_token_times: no sampling prefill to report on");
  PTTS_DEVICE(e->cfg.device);
  PTTS_HIP(hipEventSynchronize(e->ev_first));
  PTTS_HIP(hipEventElapsedTime(prefill_ms, e->ev_pre0, e->ev_tail0));
  PTTS_HIP(hipEventElapsedTime(tail_ms, e->ev_tail0, e->ev_first));
  return PTTS_OK;
}

extern "C" int ptts_state(ptts_engine* e, int32_t* cur_len, int32_t* all_finished, void* stream) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  PTTS_CHECK(e->prefilled, PTTS_E_INVALID, "ptts_state called before ptts_prefill");
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = pick_stream(e, stream);
  const int n = e->B * e->cfg.num_codebooks;
  int* hp = e->host_pinned;
  PTTS_HIP(hipMemcpyAsync(hp, e->cur_len, 4, hipMemcpyDeviceToHost, st));
  PTTS_HIP(hipMemcpyAsync(hp + 1, e->unfinished, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  PTTS_HIP(hipStreamSynchronize(st));
  if (cur_len) *cur_len = hp[0];
  int any = 0;
  for (int i = 0; i < n; ++i) any |= hp[1 + i] > 0;  // 1 = still generating, -(t + 1) = finished at step t
  if (all_finished) *all_finished = any ? 0 : 1;
  return PTTS_OK;
}

extern "C" int ptts_ids(ptts_engine* e, int64_t** ids_dev, int32_t* row_stride) {
  PTTS_CHECK(e && ids_dev && row_stride, PTTS_E_INVALID, "null argument");
  *ids_dev = reinterpret_cast<int64_t*>(e->ids);
  *row_stride = e->ids_ld;
  return PTTS_OK;
}

extern "C" int ptts_step_forward(ptts_engine* e, void* stream) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  PTTS_CHECK(e->prefilled, PTTS_E_INVALID, "ptts_step_forward called before ptts_prefill");
  PTTS_DEVICE(e->cfg.device);
  advance_kv(e);
  return forward_dispatch(e, false, pick_stream(e, stream));
}

extern "C" int ptts_logits(ptts_engine* e, float** logits_dev) {
  PTTS_CHECK(e && logits_dev, PTTS_E_INVALID, "null argument");
  *logits_dev = e->logits;
  return PTTS_OK;
}

extern "C" int ptts_push_tokens(ptts_engine* e, const int64_t* tokens_dev, const int32_t* finished_dev, void* stream) {
  PTTS_CHECK(e && tokens_dev, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(e->prefilled, PTTS_E_INVALID, "ptts_push_tokens called before ptts_prefill");
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = pick_stream(e, stream);
  const int n = e->B * e->cfg.num_codebooks;
  hipLaunchKernelGGL(push_tokens_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const long long*)tokens_dev, finished_dev, e->ids,
                     e->ids_ld, e->cur_len, e->unfinished, e->has_eos, e->B, e->cfg.num_codebooks, e->cfg.eos_token_id);
  hipLaunchKernelGGL(bump_len_kernel, dim3((e->B + 255) / 256), dim3(256), 0, st, e->cur_len, e->B);
  e->h_ready = false;
  return PTTS_OK;
}

#ifdef PTTS_TIMING
// measurement build only: the stamp buffer [layers + 1][5][3][16] (device memory) of the last replayed single-utterance step
extern "C" int ptts_debug_stamps(ptts_engine* e, long long** stamps_dev, int32_t* layers) {
  PTTS_CHECK(e && stamps_dev && layers, PTTS_E_INVALID, "null argument");
  *stamps_dev = e->dbg_stamps;
  *layers = e->cfg.num_layers;
  return PTTS_OK;
}
#endif

extern "C" int ptts_debug_graph_nodes(ptts_engine* e, int32_t* nodes) {
  PTTS_CHECK(e && nodes, PTTS_E_INVALID, "null argument");
  *nodes = e->last_graph_nodes;
  return PTTS_OK;
}

extern "C" int ptts_debug_hidden(ptts_engine* e, float** hidden_dev, int32_t* rows) {
  PTTS_CHECK(e && hidden_dev && rows, PTTS_E_INVALID, "null argument");
  *hidden_dev = e->h;
  *rows = e->B;
  return PTTS_OK;
}
