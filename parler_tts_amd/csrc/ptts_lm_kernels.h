// Device kernels of the decoder-LM engine (gfx950 / CDNA4, wave64).
//
// Data layout in HBM:
//   packed weights  W[N][K] -> [N/16 strips][K/KT frags][64 lanes][16 B]   one coalesced 1 KiB wave-load per
//                   MFMA A-fragment (bf16: 16 rows x 32 k, mfma_f32_16x16x32_bf16; f32: 16 rows x 16 k = 4 x
//                   mfma_f32_16x16x4f32). Lane l holds row (l&15), k = KT*t + (l>>4)*EPL + e.
//   KV arena        [layer][b][head][position][64] in the engine dtype (128 B / 256 B rows, 16 B per lane)
//   activations     fp32 row-major [rows][features]; residual stream h stays fp32 in both dtypes
//   sampler state   ids int64 [B*K][ld], cur_len[B], unfinished[B*K], has_eos[B*K], first_unf[B]
#pragma once
#include "ptts_common.h"

// ------------------------------------------------------------------------------------------------------
// device-resident per-call dims and generation parameters (so one captured hipGraph serves every call)
// ------------------------------------------------------------------------------------------------------
struct DevDims {
  int P;           // prompt positions prepended to the self-attention context
  int N;           // encoder positions
  int max_length;  // delay-pattern length (GenerationConfig.max_length)
  int T_prefix;    // voice prompt: audio-code frames given as decoder_input_ids (0 = none), modeling:3136-3194
  const long long* prefix;  // [B*K][prefix_ld] un-delayed codes of the voice prompt
  int prefix_ld;
};
struct DevGen {
  int max_length, min_new_tokens, do_sample, top_k, use_eos_gate;
  float temperature, top_p;
  unsigned long long seed;
};

// ------------------------------------------------------------------------------------------------------
// weight packing / conversion
// ------------------------------------------------------------------------------------------------------
template <typename ST> __device__ __forceinline__ float load_as_f32(const ST* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename DT> __device__ __forceinline__ void store_from_f32(DT* p, float v);
template <> __device__ __forceinline__ void store_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_from_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename WT, typename ST>
__global__ void pack_weight_kernel(const ST* __restrict__ src, WT* __restrict__ dst, int N, int K, int strip0, int nfrag_total) {
  constexpr int KT = Elem<WT>::KT, EPL = Elem<WT>::EPL;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nfrag = K / KT;
  const size_t total = (size_t)(N / 16) * nfrag * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int t = (int)((idx >> 6) % nfrag);
  const int s = (int)((idx >> 6) / nfrag);
  const int row = s * 16 + (lane & 15);
  const int k = t * KT + (lane >> 4) * EPL;
  WT* d = dst + (((size_t)(strip0 + s) * nfrag_total + t) * 64 + lane) * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) store_from_f32<WT>(d + e, load_as_f32<ST>(src + (size_t)row * K + k + e));
}

template <typename DT, typename ST>
__global__ void convert_kernel(const ST* __restrict__ src, DT* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_from_f32<DT>(dst + i, load_as_f32<ST>(src + i));
}

// ------------------------------------------------------------------------------------------------------
// GEMM strip kernel: out[m][n] = sum_k W[n][k] * x[m][k]  for one 16-row strip of W per workgroup.
// HBM-bound at decode (every weight byte is read exactly once per step); MFMA is used because it
// reduces over k in-register (no cross-lane shuffle tree) and makes batch <= 32 free, not for FLOPs.
//   PRO_PLAIN : x fp32 [M][K]
//   PRO_LN    : x = LayerNorm(h) (fp32 stats per row in one shifted pass, eps 1e-5, affine)   modeling:1020,:1040,:1059,:1632
//   PRO_ATTN  : x = softmax-combine of split-KV partials written by attn_kernel (S splits)
//   PRO_COPY  : x already normalised / combined by rows_prep_kernel, in the engine dtype (M > 8: do it once, not per workgroup)
//   EPI_STORE : out = acc            EPI_GELU: out = gelu_erf(acc) (:1060)
//   EPI_RESID : out += acc (residual stream, :1034/:1052/:1064)
//   EPI_KV    : scatter into the cross-attention K/V cache [b][head][t][64] (:877-878, cached :872-875)
// ------------------------------------------------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_LN = 1, PRO_ATTN = 2, PRO_COPY = 3, PRO_LNS = 4, PRO_RMS = 5 };  // PRO_COPY: x already in the engine dtype (prep kernel)
// PRO_RMS (rows_prep_kernel only; T5 description encoder, ptts_t5.hip): T5LayerNorm = x * rsqrt(mean(x^2) + eps) * weight - no mean, no bias
// PRO_LNS (8 < M <= 32): LayerNorm whose row statistics come from the PRODUCER of the residual stream. The EPI_RESID GEMM
// that wrote h also wrote, per (row, 16-column strip), the strip mean and the strip sum of squared deviations
// (`stats_out`); the consumer combines the K/16 strip partials per row (Chan's pairwise update, exact and cancellation
// free) with 3 DPP steps for 8 rows at once, so the per-workgroup prologue is loads + one fma per element: no reduction
// over K, and the separate rows_prep_kernel node (25 % of the batch-32 step in round 1) disappears.
enum { EPI_STORE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_KV = 3, EPI_GELU_WT = 4, EPI_GATE_WT = 5 };  // _WT: output in the engine dtype
// EPI_GATE_WT (T5 gated-GELU feed-forward, ptts_t5.hip): the packed matrix interleaves the rows of wi_0 and wi_1 (row 2i = wi_0[i], row 2i + 1 =
// wi_1[i]), so a lane's 4 accumulator rows are (u, v, u', v') of two output features: out[m][n / 2 .. n / 2 + 1] = gelu_new(u) * v in the engine dtype,
// row-major [M][N / 2] or B-fragment order - the gate never leaves the registers of the GEMM that produced both halves.

struct KvLayer { const void* W; void* k; void* v; };  // one layer's operands of the batched cross K/V projection (EPI_KV over blockIdx.z)

// Field order: the first 56 bytes are what a strip-kernel wave needs to ADDRESS its first loads (weight fragments, B fragments of fragment-order
// rows, the pass's rows): gemm_strip_kernel takes them as scalar parameters that the command processor preloads into SGPRs (ptts_common.h: kernel-
// argument preload) and everything else as KTail<GemmArgs>. Round 6 measured what fetching this struct by s_load costs a strip node: 0.43 us from the
// wave's first instruction until the struct is usable, three nodes per layer (profiles/r06_node_stamps_bs32_bs128.txt, slot 14); the other kernels
// that take GemmArgs keep the by-value form.
struct GemmArgs {
  // ---- bytes 0..55: preloaded ----
  const void* W;       // packed strips
  const void* W8;      // e4m3 strips [N/16][K/64 fragment pairs][64 lanes][16 B] (weights_fp8 engines), or null: bf16 / fp32 strips in W
  const float* x;      // PLAIN/LN input; x row index = m * x_row_mul + x_row_off
  float* out;
  int K, M;
  int rows_per_pass;   // activation rows staged in LDS per pass (<= 16 * MTP)
  int frags_per_wave;  // FULL variant: K/KT/W, a multiple of 8
  int out_ld;
  int kflags;          // set by ptts_klaunch: m_split | x_fo << 1 | waves per workgroup << 8 (blockDim would be a hidden argument behind an s_load)
  // ---- tail: what the decode step's strip instances read first (one or two scalar cache lines behind the preloaded bytes: the s_load latency of
  // this struct grew with the number of lines touched - 0.14 us for lnproj's 100 bytes, 0.43 us for five scattered lines of this one) ----
  int m_split;         // PRO_COPY: blockIdx.z selects ONE pass of rows_per_pass rows (grid.z = passes) instead of looping over them
  int x_fo;            // PRO_COPY: x is in MFMA B-fragment order (fo_vec_index), written by a producer with out_fo set
  int N;
  int out_fo;          // EPI_GELU_WT / rows_prep: write the engine-dtype output in B-fragment order for the consumer GEMM
#ifdef PTTS_TIMING
  long long* dbg;      // phase stamps (PTTS_STAMP)
#endif
  float* stats_out;    // EPI_RESID: [M][N/16][2] strip statistics of the updated residual rows, or null
  const float* wscale; // W8: one power-of-two scale per weight row [N], applied to the fp32 accumulators
  // T5 RMSNorm folded into the GEMMs around it (ptts_t5.hip, <= 256 rows; strip kernel only). T5LayerNorm has no mean and no bias, so it commutes with
  // the projection: W (g o x * rstd) = rstd * (W (g o x)). The PRODUCER of the residual rows (EPI_RESID with nx_out) also writes the next GEMM's
  // operand g o h in the engine dtype and the per-strip sums of h^2; the CONSUMER (EPI_STORE / EPI_GATE_WT with rs_part) sums a row's rs_n partials
  // once per workgroup, in a fixed order, while its weights stream, and scales its accumulators by rstd[m]: no rows_prep node, no extra pass over h.
  void* nx_out;          // producer (EPI_RESID): engine-dtype [M][N] (row-major or fragment order by out_fo) = updated residual row * nx_gamma, or null
  const float* nx_gamma; // [N]
  float* ss_out;         // producer: [M][N/16] sum of squares of the updated residual row over this strip's 16 columns
  const float* rs_part;  // consumer: [M][rs_n] per-strip sums of squares of the un-normalised rows, or null
  int rs_n;
  float rs_invD;         // 1 / d_model
  float rms_eps;       // PRO_RMS / rs_part: T5Config.layer_norm_epsilon
  // ---- everything else ----
  int x_ld, x_row_mul, x_row_off;
  const float* gamma;
  const float* beta;
  const float* part;   // ATTN: [rows][S][K]
  const float* stats;  // ATTN: [rows][S][heads][2] = (max, sumexp)
  int S, nheads;
  void* kcache;        // EPI_KV
  void* vcache;
  int kv_rows_per_b;   // EPI_KV: N (rows of x per batch element)
  int kv_cap;          // EPI_KV: capacity (positions) of the cache
  int kv_col0;         // EPI_STORE with kcache set (round 6: the prefill's QKV projection above 256 rows): output columns [kv_col0, kv_col0 + 64 nheads) are K,
                       // the next 64 nheads V - also written into the self-attention cache rows (utterance m / kv_rows_per_b, position m % kv_rows_per_b) in
                       // the engine dtype, as kv_append_kernel would: no kv_append node (sinusoidal positions, engine-dtype cache); 0 = off
  int kv_pad_;
  float invK;          // 1 / K
  const float* lnstat; // PRO_LNS: [M][K/16][2] = (strip mean, strip M2) of x, written by the producer GEMM
  const KvLayer* kv_layers;  // PRO_COPY + EPI_KV: blockIdx.z selects the layer (W, kcache, vcache from this table): the description's K/V
  int kv_nlayers;            // of EVERY layer in one launch (24 launches of ~8 us sat on the time-to-first-token path); null = W / kcache / vcache
  int xcd_swz;         // gemm_block_kernel / gemm_tile_kernel: XCD-aware tile order (xcd_tile_order), set by launch_gemm (PTTS_GEMM_XCD=0: launch order)
  int decode;          // host-side launch policy only: 1 = a decode-step GEMM (light M passes, msplit_rows), 0 = prefill-sized rows
  const int* row_keep; // PRO_RMS: [M] int32 or null; rows with 0 are written as zeros (masked description positions, modeling_parler_tts.py:3093-3097)
};
static_assert(sizeof(GemmArgs) % 8 == 0 && offsetof(GemmArgs, m_split) == 56, "GemmArgs: 56 preloaded bytes + tail");
#define GemmArgs_KPARAMS                                                                                                                          \
  const void *kW_, const void *kW8_, const float *kx_, float *kout_, int kK_, int kM_, int krpp_, int kfpw_, int kold_, int kfl_, KTail<GemmArgs> kt_
#define GemmArgs_KJOIN(a)                                                                                          \
  GemmArgs a;                                                                                                      \
  PTTS_KTAIL_JOIN(GemmArgs, a);                                                                                    \
  a.W = kW_; a.W8 = kW8_; a.x = kx_; a.out = kout_; a.K = kK_; a.M = kM_; a.rows_per_pass = krpp_; a.frags_per_wave = kfpw_; a.out_ld = kold_;  \
  a.kflags = kfl_; a.m_split = kfl_ & 1; a.x_fo = (kfl_ >> 1) & 1;

// ---- activations in MFMA B-fragment order ("FO") -----------------------------------------------------------------------------
// Decode at batch > 8: the engine-dtype activation rows that a PRO_COPY GEMM consumes are written by their producers (rows_prep,
// attention, fused cross block, fc1's GELU epilogue) in the order the consumer's MFMA loop reads them:
//   X_fo[M/16 tiles][K/KT fragments][64 lanes][16 B], lane l of fragment (mt, t) = row mt*16 + (l & 15), k = t*KT + (l >> 4)*EPL + e
// (the weight packing with rows <-> utterances), so one B-fragment load of a wave is ONE contiguous 1 KiB instead of 16 rows x 64 B
// (half-used 128-B lines; measured: M = 128 GEMMs of 2 MB weights took 11-12 us, the K = 4096 one 34 us, profiles/r03_step_bf16_bs128_v0.txt).
// Index of the 16-byte vector that holds elements (m, k .. k + EPL - 1), k % EPL == 0:
template <typename WT> __device__ __forceinline__ size_t fo_vec_index(int m, int k, int nfrag) {
  constexpr int KT = Elem<WT>::KT, EPL = Elem<WT>::EPL;
  return ((size_t)(m >> 4) * nfrag + k / KT) * 64 + ((k % KT) / EPL) * 16 + (m & 15);
}
// store 4 consecutive elements (k % 4 == 0) of row m: row-major [M][K] or fragment order
template <typename WT> __device__ __forceinline__ void act_store4(WT* base, int m, int k, int K, int fo, float a, float b, float c, float d);
template <> __device__ __forceinline__ void act_store4<float>(float* base, int m, int k, int K, int fo, float a, float b, float c, float d) {
  float4* p = fo ? reinterpret_cast<float4*>(base) + fo_vec_index<float>(m, k, K / Elem<float>::KT) : reinterpret_cast<float4*>(base + (size_t)m * K + k);
  *p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void act_store4<bf16_t>(bf16_t* base, int m, int k, int K, int fo, float a, float b, float c, float d) {
  uint2* p = fo ? reinterpret_cast<uint2*>(reinterpret_cast<uint4*>(base) + fo_vec_index<bf16_t>(m, k & ~7, K / Elem<bf16_t>::KT)) + ((k >> 2) & 1)
                : reinterpret_cast<uint2*>(base + (size_t)m * K + k);
  *p = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}

template <typename WT> struct MfmaStep;
template <> struct MfmaStep<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MfmaStep<float> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

// ---- e4m3 weight strips (weights_fp8 engines, decode at batch >= 5) -------------------------------------------------------------
// A lane's 16 bytes hold its 8 weights of TWO consecutive k fragments (bytes 0-7: k = 64p + (l >> 4)*8 + e, bytes 8-15: + 32): one
// 1 KiB wave-load per fragment PAIR, half the bytes of the bf16 strips. e4m3 -> bf16 is exact (3 mantissa bits into 7) and done in
// registers by v_cvt_scalef32_pk_bf16_fp8 (scale 1.0): 4 VALU ops per A fragment beside an HBM-bound stream; the MFMA is the same
// mfma_f32_16x16x32_bf16 with bf16 activations, i.e. the arithmetic of the bf16 engine on the exact dequantisation, and the per-row
// power-of-two scale multiplies the fp32 accumulators (exact, commutes with the sum).
// row-major e4m3 bytes [N][K] (rows of one projection, landing at strip `strip0`) -> strip order; one thread = one lane's 16 bytes
static __global__ void pack_w8_kernel(const uint8_t* __restrict__ src, uint4* __restrict__ dst, int N, int K, int strip0, int npair_total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int npair = K / 64;
  if (idx >= (size_t)(N / 16) * npair * 64) return;
  const int lane = idx & 63, p = (int)((idx >> 6) % npair), s = (int)((idx >> 6) / npair);
  const uint8_t* r = src + (size_t)(s * 16 + (lane & 15)) * K + 64 * p + (lane >> 4) * 8;
  const uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 32);
  dst[((size_t)(strip0 + s) * npair_total + p) * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// Measurement build only (-DPTTS_TIMING, tools/build_stamps.sh; never the product library): phase stamps of the device-wide 100 MHz counter
// (wall_clock64 = s_memrealtime; s_memtime counts per XCD with unrelated offsets) by lane 0 of wave 0 of the first / middle / last workgroup of
// a launch into dbg[3][16] - round 5 stamped the single-utterance GEMV step this way (ptts_gemv.h: GV_STAMP), round 6 the seven nodes per layer of
// the batch > 8 step (lnproj_fused / attn / gemm_strip / xattn_fused): profiles/r06_node_stamps_bs32.txt
#ifdef PTTS_TIMING
#define PTTS_DBG_FIELD long long* dbg;
#define PTTS_STAMP(ptr, i)                                                                                                  \
  do {                                                                                                                      \
    if ((ptr) && threadIdx.x == 0) {                                                                                        \
      const unsigned nb_ = gridDim.x * gridDim.y * gridDim.z, lb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
      const int slot_ = lb_ == 0 ? 0 : (lb_ == nb_ / 2 ? 1 : (lb_ == nb_ - 1 ? 2 : -1));                                    \
      if (slot_ >= 0) (ptr)[slot_ * 16 + (i)] = (long long)wall_clock64();                                                  \
    }                                                                                                                       \
  } while (0)
#define PTTS_WSTAMP(a, i) PTTS_STAMP((a).dbg, i)
// gemm_strip_kernel / lnproj_fused_kernel take the stamp buffer a second time as a leading SCALAR parameter: the command processor preloads it into
// SGPRs (-amdgpu-kernarg-preload-count), so slot 14 is stamped at the wave's very first instructions - before the s_load of the by-value argument
// struct has returned. s0 - s14 = what fetching the struct costs a node = the most a kernarg-preload conversion of these kernels could win.
#define PTTS_DBG0_PARAM long long* dbg0_,
#define PTTS_DBG0_ARG(a) (a).dbg,
#define PTTS_STAMP0() PTTS_STAMP(dbg0_, 14)
#else
#define PTTS_DBG0_PARAM
#define PTTS_DBG0_ARG(a)
#define PTTS_STAMP0() do { } while (0)
#define PTTS_DBG_FIELD
#define PTTS_STAMP(ptr, i) do { } while (0)
#define PTTS_WSTAMP(a, i) do { } while (0)
#endif
#ifdef PTTS_TIMING
template <typename A> __device__ __forceinline__ long long* ptts_dbg_of(const A&) { return nullptr; }
#define PTTS_LN_STAMP(a, i) PTTS_STAMP(ptts_dbg_of(a), i)
#else
#define PTTS_LN_STAMP(a, i) do { } while (0)
#endif
#ifndef PTTS_TIMING
#define PTTS_DBG(a) ((long long*)nullptr)
#else
#define PTTS_DBG(a) ((a).dbg)
#endif

#ifdef PTTS_TIMING
template <> __device__ __forceinline__ long long* ptts_dbg_of<GemmArgs>(const GemmArgs& a) { return a.dbg; }
#endif
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// transformers NewGELUActivation ("gelu_new", T5 v1.1 / flan-t5 gated-gelu): 0.5 x (1 + tanh(sqrt(2 / pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_new(float x) { return 0.5f * x * (1.0f + tanhf(0.79788456080286535588f * (x + 0.044715f * (x * x * x)))); }
// two consecutive engine-dtype elements (k % 2 == 0) of row m: row-major [M][K] or fragment order
template <typename WT> __device__ __forceinline__ void act_store2(WT* base, int m, int k, int K, int fo, float a, float b);
template <> __device__ __forceinline__ void act_store2<float>(float* base, int m, int k, int K, int fo, float a, float b) {
  float2* p = fo ? reinterpret_cast<float2*>(reinterpret_cast<float4*>(base) + fo_vec_index<float>(m, k & ~3, K / Elem<float>::KT)) + ((k >> 1) & 1)
                 : reinterpret_cast<float2*>(base + (size_t)m * K + k);
  *p = make_float2(a, b);
}
template <> __device__ __forceinline__ void act_store2<bf16_t>(bf16_t* base, int m, int k, int K, int fo, float a, float b) {
  uint32_t* p = fo ? reinterpret_cast<uint32_t*>(reinterpret_cast<uint4*>(base) + fo_vec_index<bf16_t>(m, k & ~7, K / Elem<bf16_t>::KT)) + ((k >> 1) & 3)
                   : reinterpret_cast<uint32_t*>(base + (size_t)m * K + k);
  *p = pack_bf16x2(a, b);
}

// ---- activation staging ---------------------------------------------------------------------------------
// The rows of one pass are brought into LDS ONCE per workgroup, already in their final form (LayerNorm applied /
// split-KV partials combined) and already in the engine dtype, with a 16-byte row pad (conflict-free b128 reads of
// 16 different rows). The MFMA loop then only touches registers (weights) and LDS (activations): no global
// round trip sits between two MFMAs. Profiled motivation: with per-fragment global loads each decode GEMM took
// 8-16 us for <= 8 MB of weights (profiles/r01_step_bf16_bs1_v0.txt).
template <typename WT> __device__ __forceinline__ void lds_store4(char* row, int k, float a, float b, float c, float d);
template <> __device__ __forceinline__ void lds_store4<float>(char* row, int k, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(row + (size_t)k * 4) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void lds_store4<bf16_t>(char* row, int k, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(row + (size_t)k * 2) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}

constexpr int LN_MAX_F4 = 8;  // LayerNorm rows up to 64 lanes * 8 float4 = 2048 wide live in registers

// One wave normalises R rows at a time (R = 1 at bs=1, 2 when a wave owns several rows: two independent latency
// chains in flight). NF4 float4 per lane, statically indexed (registers, never scratch). gamma/beta are loaded once
// per wave; every row load is issued before the first dependent instruction; mean and variance come from ONE fused
// pass of sum(x-c) and sum((x-c)^2) with the shift c = x[0] (shifted-data variance: no catastrophic cancellation,
// error ~ eps*(1 + (mean-c)^2/var)); eps 1e-5 as nn.LayerNorm (modeling:961). EXACT: K == NF4*256, no lane masks.
template <typename WT, int NF4, bool EXACT, int R, typename Args, typename Hook>
__device__ __forceinline__ void ln_rows(const Args& a, const float* const (&xr)[R], char* const (&row)[R], int lane, Hook&& hook) {
  // Issue order = need order: the CU returns loads in issue order, so the 4 KB row (critical path) goes first, the
  // caller's bulk weight loads (hook) next, gamma/beta (needed only after the reductions) last. With the weights
  // first the row queued behind 32-128 KB per CU: +0.7 us per LayerNorm kernel (tools/chain_probe.hip).
  float4 v[R][NF4];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int k = (lane + 64 * i) * 4;
      v[r][i] = *reinterpret_cast<const float4*>(xr[r] + ((EXACT || k < a.K) ? k : 0));
    }
  float4 g[NF4], bt[NF4];
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int k = (lane + 64 * i) * 4;
    const int kk = (EXACT || k < a.K) ? k : 0;
    g[i] = *reinterpret_cast<const float4*>(a.gamma + kk);
    bt[i] = *reinterpret_cast<const float4*>(a.beta + kk);
  }
  PTTS_LN_STAMP(a, 6);
  hook(0);  // workgroup rendezvous: every wave's row loads are queued before any wave's weight loads
  PTTS_LN_STAMP(a, 7);
  float c[R], s1[R], s2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    c[r] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[r][0].x)));
    s1[r] = 0.f; s2[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const float msk = (EXACT || (lane + 64 * i) * 4 < a.K) ? 1.f : 0.f;
      const float d0 = (v[r][i].x - c[r]) * msk, d1 = (v[r][i].y - c[r]) * msk, d2 = (v[r][i].z - c[r]) * msk, d3 = (v[r][i].w - c[r]) * msk;
      s1[r] += (d0 + d1) + (d2 + d3);
      s2[r] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
#pragma unroll
  PTTS_LN_STAMP(a, 8);
  for (int r = 0; r < R; ++r) { s1[r] = wave_sum(s1[r]); s2[r] = wave_sum(s2[r]); }
  PTTS_LN_STAMP(a, 9);
  // normalise in registers (consumes gamma/beta: every load of this wave has landed), THEN start this wave's weight
  // stream, then write LDS: no wait on a row/gamma/beta load can end up behind the weight loads
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float dm = s1[r] * a.invK;
    const float mean = c[r] + dm;
    const float rstd = rsqrtf(fmaxf(s2[r] * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      v[r][i].x = (v[r][i].x - mean) * rstd * g[i].x + bt[i].x;
      v[r][i].y = (v[r][i].y - mean) * rstd * g[i].y + bt[i].y;
      v[r][i].z = (v[r][i].z - mean) * rstd * g[i].z + bt[i].z;
      v[r][i].w = (v[r][i].w - mean) * rstd * g[i].w + bt[i].w;
    }
  }
  PTTS_LN_STAMP(a, 10);
  hook(1);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int k = (lane + 64 * i) * 4;
      if (EXACT || k < a.K) lds_store4<WT>(row[r], k, v[r][i].x, v[r][i].y, v[r][i].z, v[r][i].w);
    }
}

// `first` runs exactly once per wave: right after the wave's first row loads are in flight, or at once if it owns no row.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <typename WT, int NF4, bool EXACT, typename Args, bool PAIRS = true, typename Hook>
__device__ __forceinline__ void ln_stage(const Args& a, int m0, int nrows, char* s_x, int row_bytes, int lane, int wave, int W, Hook&& first) {
  auto xrow = [&](int r) { return a.x + (size_t)((m0 + r) * a.x_row_mul + a.x_row_off) * a.x_ld; };
  int r = wave;
  if (PAIRS && r + W < nrows) {
    const float* const xr[2] = {xrow(r), xrow(r + W)};
    char* const row[2] = {s_x + (size_t)r * row_bytes, s_x + (size_t)(r + W) * row_bytes};
    ln_rows<WT, NF4, EXACT, 2>(a, xr, row, lane, first);
    r += 2 * W;
  } else if (r < nrows) {
    const float* const xr[1] = {xrow(r)};
    char* const row[1] = {s_x + (size_t)r * row_bytes};
    ln_rows<WT, NF4, EXACT, 1>(a, xr, row, lane, first);
    r += W;
  } else {
    first(2);
  }
  for (; PAIRS && r + W < nrows; r += 2 * W) {  // two rows of this wave in flight
    const float* const xr[2] = {xrow(r), xrow(r + W)};
    char* const row[2] = {s_x + (size_t)r * row_bytes, s_x + (size_t)(r + W) * row_bytes};
    ln_rows<WT, NF4, EXACT, 2>(a, xr, row, lane, NoHook());
  }
  for (; r < nrows; r += W) {
    const float* const xr[1] = {xrow(r)};
    char* const row[1] = {s_x + (size_t)r * row_bytes};
    ln_rows<WT, NF4, EXACT, 1>(a, xr, row, lane, NoHook());
  }
}

// PRO_LNS staging. Wave w owns rows w, w + W, ... (<= 8 of them). Statistics: lane (ri, g) = (lane >> 3, lane & 7) reads a
// contiguous chunk of K/128 strip partials of row w + W*ri, folds them relative to the first strip's mean, and an 8-lane DPP
// reduction finishes all 8 rows together. Rows are then normalised from registers into LDS exactly as PRO_LN does.
template <typename WT, int NF4, typename Hook>
__device__ __forceinline__ void lns_stage(const GemmArgs& a, int nrows, char* s_x, int row_bytes, int lane, int wave, int W, Hook&& first) {
  constexpr int NS = NF4 * 16;        // producer strips per row (K == NF4 * 256)
  constexpr int CH = NF4 * 2;         // strip partials per lane
  const int ri = lane >> 3, gq = lane & 7;
  const int srow = min(wave + W * ri, nrows - 1);
  const float2* sp = reinterpret_cast<const float2*>(a.lnstat) + (size_t)srow * NS;
  const float2 first_strip = sp[0];
  float2 part[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) part[i] = sp[gq * CH + i];
  // first batch of rows (RB * NF4 float4 per lane) in flight together with gamma / beta
  constexpr int RB = 2;  // rows in flight per wave (4 spilled 10-25 registers at the 256 budget of a 512-thread workgroup)
  float4 g[NF4], bt[NF4];
  auto load_rows = [&](int i0, float4 (&v)[RB][NF4]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = min(wave + W * (i0 + i), nrows - 1);
      const float* xr = a.x + (size_t)(r * a.x_row_mul + a.x_row_off) * a.x_ld;
#pragma unroll
      for (int f = 0; f < NF4; ++f) v[i][f] = *reinterpret_cast<const float4*>(xr + (lane + 64 * f) * 4);
    }
  };
  float4 v[RB][NF4];
  load_rows(0, v);
#pragma unroll
  for (int f = 0; f < NF4; ++f) {
    g[f] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * f) * 4);
    bt[f] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * f) * 4);
  }
  first(2);  // rendezvous + this wave's first weight fragments, queued behind the activations
  // ---- row statistics for the 8 row slots of this wave
  const float c = first_strip.x;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const float d = part[i].x - c;
    s1 += d;
    s2 += part[i].y + 16.f * d * d;
  }
  s1 = group_reduce<OpSum, 8>(s1);
  s2 = group_reduce<OpSum, 8>(s2);
  const float dm = s1 / (float)NS;
  const float mean_l = c + dm;
  const float rstd_l = rsqrtf(fmaxf((s2 - 16.f * (float)NS * dm * dm) * a.invK, 0.f) + 1e-5f);
  const int nslots = (nrows - wave + W - 1) / W;  // rows this wave really owns
#pragma unroll 1
  for (int i0 = 0; i0 < nslots; i0 += RB) {
    if (i0 != 0) load_rows(i0, v);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if (i0 + i < nslots) {
        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean_l), (i0 + i) * 8));
        const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstd_l), (i0 + i) * 8));
        char* row = s_x + (size_t)(wave + W * (i0 + i)) * row_bytes;
#pragma unroll
        for (int f = 0; f < NF4; ++f)
          lds_store4<WT>(row, (lane + 64 * f) * 4, (v[i][f].x - mean) * rstd * g[f].x + bt[f].x, (v[i][f].y - mean) * rstd * g[f].y + bt[f].y,
                         (v[i][f].z - mean) * rstd * g[f].z + bt[f].z, (v[i][f].w - mean) * rstd * g[f].w + bt[f].w);
      }
    }
  }
}

// one (row, 4-column) element of the PLAIN / ATTN prologue
template <int PRO>
__device__ __forceinline__ float4 stage_elem(const GemmArgs& a, int m, int k) {
  if (PRO == PRO_ATTN) {
    const int head = k >> 6;
    const float* st = a.stats + ((size_t)m * a.S * a.nheads + head) * 2;
    float mx = -INFINITY;
    for (int sp = 0; sp < a.S; ++sp) mx = fmaxf(mx, st[(size_t)sp * a.nheads * 2]);
    float den = 0.f;
    float4 o = make_float4(0, 0, 0, 0);
    for (int sp = 0; sp < a.S; ++sp) {
      const float ms = st[(size_t)sp * a.nheads * 2], ls = st[(size_t)sp * a.nheads * 2 + 1];
      const float w = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ms - mx);  // statistics are in log2 units (attn_kernel)
      den += w * ls;
      const float4 t = *reinterpret_cast<const float4*>(a.part + ((size_t)m * a.S + sp) * a.K + k);
      o.x += w * t.x; o.y += w * t.y; o.z += w * t.z; o.w += w * t.w;
    }
    const float inv = den > 0.f ? __frcp_rn(den) : 0.f;
    return make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
  return *reinterpret_cast<const float4*>(a.x + (size_t)(m * a.x_row_mul + a.x_row_off) * a.x_ld + k);
}

// `first` (the caller's bulk weight loads) runs exactly once per thread, right after the thread's first activation
// loads are in flight: the CU returns loads in issue order and the activations are the critical path.
template <typename WT, int PRO, bool FULL, typename Hook>
__device__ __forceinline__ void stage_rows(const GemmArgs& a, int m0, int nrows, char* s_x, int row_bytes, int lane, int wave, int W, Hook&& first) {
  if (PRO == PRO_LN) {
    const int nf4 = (a.K + 255) >> 8;  // workgroup-uniform
    if (FULL) {  // host guarantees K in {256, 512, 1024, 1536} for the FULL LayerNorm variant
      if (nf4 == 4) ln_stage<WT, 4, true>(a, m0, nrows, s_x, row_bytes, lane, wave, W, first);        // hidden 1024 (Mini-v1)
      else if (nf4 == 6) ln_stage<WT, 6, true>(a, m0, nrows, s_x, row_bytes, lane, wave, W, first);   // hidden 1536 (Large-v1)
      else ln_stage<WT, 2, false>(a, m0, nrows, s_x, row_bytes, lane, wave, W, first);
    } else {
      if (nf4 <= 1) ln_stage<WT, 1, false>(a, m0, nrows, s_x, row_bytes, lane, wave, W, first);
      else ln_stage<WT, LN_MAX_F4, false>(a, m0, nrows, s_x, row_bytes, lane, wave, W, first);
    }
  } else if (PRO == PRO_LNS) {  // host guarantees K in {1024, 1536}, 8 < M <= 32, rows_per_pass == M, W >= 4
    if (a.K == 1024) lns_stage<WT, 4>(a, nrows, s_x, row_bytes, lane, wave, W, first);
    else lns_stage<WT, 6>(a, nrows, s_x, row_bytes, lane, wave, W, first);
  } else if (PRO == PRO_COPY) {
    first(2);
    // bulk copy of engine-dtype rows, 16 B per lane, 8 independent loads in flight per thread
    constexpr int EPV = 16 / (int)sizeof(WT);
    const int vpr = a.K / EPV;  // 16-byte vectors per row
    const int total = nrows * vpr;
    const WT* xb = reinterpret_cast<const WT*>(a.x);
    for (int i0 = wave * 64 + lane; i0 < total; i0 += 8 * W * 64) {
      uint4 v[8];
      int rr[8], cc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u * W * 64, total - 1);
        rr[u] = i / vpr;
        cc[u] = i - rr[u] * vpr;
        v[u] = *reinterpret_cast<const uint4*>(xb + (size_t)((m0 + rr[u]) * a.x_row_mul + a.x_row_off) * a.x_ld + (size_t)cc[u] * EPV);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * W * 64 < total) *reinterpret_cast<uint4*>(s_x + (size_t)rr[u] * row_bytes + (size_t)cc[u] * 16) = v[u];
    }
  } else {
    // element-parallel: rows in the outer loop (no integer division), two rows in flight per thread. The thread's first
    // element(s) are peeled so that `first` sits in straight-line code between their loads and their LDS stores (the
    // compiler then waits with an exact vmcnt for the activation loads only, not for the weights issued behind them).
    const int k4n = a.K >> 2, k4s = wave * 64 + lane, step = W * 64;
    const bool two = nrows >= 2;
    {
      float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
      if (k4s < k4n) {
        o0 = stage_elem<PRO>(a, m0, k4s * 4);
        if (two) o1 = stage_elem<PRO>(a, m0 + 1, k4s * 4);
      }
      first(2);
      if (k4s < k4n) {
        lds_store4<WT>(s_x, k4s * 4, o0.x, o0.y, o0.z, o0.w);
        if (two) lds_store4<WT>(s_x + (size_t)row_bytes, k4s * 4, o1.x, o1.y, o1.z, o1.w);
      }
    }
    for (int k4 = k4s + step; k4 < k4n; k4 += step) {
      const float4 o0 = stage_elem<PRO>(a, m0, k4 * 4);
      lds_store4<WT>(s_x, k4 * 4, o0.x, o0.y, o0.z, o0.w);
      if (two) {
        const float4 o1 = stage_elem<PRO>(a, m0 + 1, k4 * 4);
        lds_store4<WT>(s_x + (size_t)row_bytes, k4 * 4, o1.x, o1.y, o1.z, o1.w);
      }
    }
    int r = 2;
    for (; r + 1 < nrows; r += 2) {
      for (int k4 = k4s; k4 < k4n; k4 += step) {
        const float4 o0 = stage_elem<PRO>(a, m0 + r, k4 * 4);
        const float4 o1 = stage_elem<PRO>(a, m0 + r + 1, k4 * 4);
        lds_store4<WT>(s_x + (size_t)r * row_bytes, k4 * 4, o0.x, o0.y, o0.z, o0.w);
        lds_store4<WT>(s_x + (size_t)(r + 1) * row_bytes, k4 * 4, o1.x, o1.y, o1.z, o1.w);
      }
    }
    if (r < nrows) {
      for (int k4 = k4s; k4 < k4n; k4 += step) {
        const float4 o0 = stage_elem<PRO>(a, m0 + r, k4 * 4);
        lds_store4<WT>(s_x + (size_t)r * row_bytes, k4 * 4, o0.x, o0.y, o0.z, o0.w);
      }
    }
  }
}

// EPI_STORE with GemmArgs::kv_col0: the K / V columns of a QKV projection row go straight into the self-attention cache (kv_append_kernel's store)
template <typename WT>
__device__ __forceinline__ void gemm_store_kv_cols(const GemmArgs& a, int m, int n, const f32x4& r) {
  const int Hkv = a.nheads * 64, nn = n - a.kv_col0, isv = nn >= Hkv, n2 = isv ? nn - Hkv : nn;
  const int head = n2 >> 6, d = n2 & 63, bu = m / a.kv_rows_per_b, pos = m - bu * a.kv_rows_per_b;
  WT* dst = reinterpret_cast<WT*>(isv ? a.vcache : a.kcache) + (((size_t)bu * a.nheads + head) * a.kv_cap + pos) * 64 + d;
#pragma unroll
  for (int e = 0; e < 4; ++e) store_from_f32<WT>(dst + e, r[e]);
}
// host side of the split argument list of gemm_strip_kernel (GemmArgs_KPARAMS)
template <typename Kn> inline void ptts_klaunch(Kn kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const GemmArgs& a) {
  const int kflags = (a.m_split ? 1 : 0) | (a.x_fo ? 2 : 0) | ((int)(block.x >> 6) << 8);
  hipLaunchKernelGGL(kern, grid, block, shmem, st, PTTS_DBG0_ARG(a) a.W, a.W8, a.x, a.out, a.K, a.M, a.rows_per_pass, a.frags_per_wave, a.out_ld, kflags,
                     ptts_ktail(a));
}

// LN / ATTN prologues always reduce over K = hidden_size (<= 8 waves of >= 8 fragments); only the plain prologue
// (fc2, K = ffn_dim) at batch <= 16 wants 16 waves, so only it pays the 128-VGPR cap of a 1024-thread workgroup.
template <int PRO, int MTP> struct GemmMaxThreads { static constexpr int value = (PRO == PRO_PLAIN && MTP == 1) ? 1024 : 512; };

// a.rows_per_pass rows (<= 16*MTP) of activations are staged per pass; LDS = staging + cross-wave reduction.
// FULL: every wave owns a whole number of 8-fragment groups and K % 256 == 0 -> straight-line code, no predicates.
// W8 (bf16 engine, FULL only): the strip's weights are e4m3 fragment pairs (a.W8) + row scales (a.wscale) instead of a.W.
template <typename WT, int PRO, int EPI, int MTP, bool FULL, bool W8 = false>
__device__ __forceinline__ void gemm_strip_body(GemmArgs& a) {
  static_assert(!W8 || (FULL && sizeof(WT) == 2), "e4m3 strips: bf16 engine, FULL variant");
  constexpr int KT = Elem<WT>::KT, U = 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = a.kflags >> 8;  // waves per workgroup, from the preloaded flags (not blockDim)
  const int row_bytes = a.K * (int)sizeof(WT) + 16;
  char* s_x = smem_raw;                                                                   // [rows_per_pass][row_bytes]
  float* s_red = reinterpret_cast<float*>(smem_raw + (PRO == PRO_COPY ? 0 : (size_t)a.rows_per_pass * row_bytes));  // [W][MTP][64][4]
  if (PRO == PRO_COPY && EPI == EPI_KV && a.kv_layers) {  // batched over the layers
    const KvLayer t = a.kv_layers[blockIdx.z];
    a.W = t.W; a.kcache = t.k; a.vcache = t.v;
  }
  const int strip = blockIdx.x;
  const int nfrag = a.K / KT;
  const int per = FULL ? a.frags_per_wave : (nfrag + W - 1) / W;
  const int t0 = wave * per, t1 = FULL ? t0 + per : min(nfrag, t0 + per);
  // W8: one uint4 per fragment PAIR (t0 and every 8-fragment group are even)
  const uint4* Wp = W8 ? reinterpret_cast<const uint4*>(a.W8) + (size_t)strip * (nfrag >> 1) * 64 + lane
                       : reinterpret_cast<const uint4*>(a.W) + (size_t)strip * nfrag * 64 + lane;
  const int q = lane >> 4, j = lane & 15;

  const int m_first = (PRO == PRO_COPY && a.m_split) ? (int)blockIdx.z * a.rows_per_pass : 0;
  const int m_last = (PRO == PRO_COPY && a.m_split) ? min(a.M, m_first + a.rows_per_pass) : a.M;
  float* s_rstd = s_red + (size_t)W * MTP * 256;  // [rows_per_pass <= 64] (rs_part consumers; the launch reserves 256 bytes behind the reduction buffer)
  for (int m0 = m_first; m0 < m_last; m0 += a.rows_per_pass) {
    const int nrows = min(a.rows_per_pass, a.M - m0);
    PTTS_STAMP(PTTS_DBG(a), 0);
    // 0. EPI_RESID: the residual values this wave will update are fetched now, not after the reduction (one cold
    //    round trip off the tail of the kernel)
    float4 resid_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_RESID && wave < MTP && wave * 16 + j < nrows)
      resid_pre = *reinterpret_cast<const float4*>(a.out + (size_t)(m0 + wave * 16 + j) * a.out_ld + strip * 16 + q * 4);
    // 1. the first group of weight fragments goes in flight as early as possible - but AFTER this thread's first
    //    activation loads (issue order = return order; the activations are the critical path, the weights are bulk).
    uint4 afr[W8 ? U / 2 : U];
    float4 wsc4 = make_float4(1.f, 1.f, 1.f, 1.f);  // W8: scales of this lane's 4 output rows (D[row = q*4 + e]), fetched with the first loads
    if (W8) wsc4 = *reinterpret_cast<const float4*>(a.wscale + strip * 16 + q * 4);
    // stage 0 = rendezvous only, 1 = loads only, 2 = both (see ln_rows for why LayerNorm waves split the two)
    auto issue_w = [&](int stage) __attribute__((always_inline)) {
      if (stage != 1) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // no fence: every wave's activation loads are queued before anybody's weights
      }
      if (stage != 0) {
        if constexpr (W8) {
#pragma unroll
          for (int u = 0; u < U / 2; ++u) afr[u] = ld_nt16(Wp + (size_t)((t0 >> 1) + u) * 64);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (FULL || t0 + u < t1) afr[u] = ld_nt16(Wp + (size_t)(t0 + u) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the issue order: nothing that waits on a load moves above these
      }
    };
    // 2. activations of this pass -> LDS (final form, engine dtype). PRO_COPY rows are already final and
    //    L2-resident: their B fragments (16 B per lane) are read straight from global, no staging, no barrier.
    PTTS_STAMP(PTTS_DBG(a), 1);
    if (PRO != PRO_COPY) {
      stage_rows<WT, PRO, FULL>(a, m0, nrows, s_x, row_bytes, lane, wave, W, issue_w);
      PTTS_STAMP(PTTS_DBG(a), 2);
      __syncthreads();
    } else {
      issue_w(1);
    }
    if (PRO == PRO_COPY && (EPI == EPI_STORE || EPI == EPI_GATE_WT) && a.rs_part) {
      // rstd of this pass's rows: 4 lanes per row, each a quarter of the rs_n strip partials (requested right behind the first weight fragments: call 7 - requested BEFORE them the round trip sat in front of the weight stream and ate what the removed rows_prep nodes saved), fixed summation order
      for (int r0 = 0; r0 < nrows; r0 += (int)(blockDim.x >> 2)) {
        const int r = r0 + (int)(threadIdx.x >> 2), part = threadIdx.x & 3, per = a.rs_n >> 2;
        float s = 0.f;
        if (r < nrows) {
          const float* pp = a.rs_part + (size_t)(m0 + r) * a.rs_n + part * per;
          for (int i = 0; i < per; ++i) s += pp[i];
        }
        s += dpp_mov<0xB1>(s);  // lanes 4k .. 4k + 3: quad_perm [1,0,3,2], then [2,3,0,1]
        s += dpp_mov<0x4E>(s);
        if (r < nrows && part == 0) s_rstd[r] = rsqrtf(s * a.rs_invD + a.rms_eps);
      }
    }
    PTTS_STAMP(PTTS_DBG(a), 3);
    // 3. MFMA over this wave's K slice; B fragments come from LDS (rows beyond nrows are clamped: their output
    //    columns are never stored). All LDS reads of a group are issued before its first MFMA.
    const char* brow[MTP];
    const bool xfo = PRO == PRO_COPY && a.x_fo;  // B fragments in fragment order: fragment t of row tile mt is 1 KiB at ((tile * nfrag + t) * 64 + lane) * 16
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) {
      const int rloc = min(mt * 16 + j, nrows - 1);
      if (xfo)  // rows past M inside the last tile hold stale data: their output columns are never stored
        brow[mt] = reinterpret_cast<const char*>(a.x) + (((size_t)((m0 >> 4) + min(mt, (nrows - 1) >> 4)) * nfrag) * 64 + lane) * 16;
      else
        brow[mt] = PRO == PRO_COPY
                       ? reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.x) + (size_t)((m0 + rloc) * a.x_row_mul + a.x_row_off) * a.x_ld) + (size_t)q * 16
                       : s_x + (size_t)rloc * row_bytes + (size_t)q * 16;
    }
    const size_t bstep = xfo ? (size_t)1024 : (size_t)(KT * sizeof(WT));  // bytes between consecutive fragments of one row tile
    f32x4 acc[MTP], acc2[MTP];  // two independent accumulator chains per tile (MFMA dependent latency)
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) { acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int tb = t0; tb < t1; tb += U) {
      if (tb != t0) {
        if constexpr (W8) {
#pragma unroll
          for (int u = 0; u < U / 2; ++u) afr[u] = ld_nt16(Wp + (size_t)((tb >> 1) + u) * 64);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (FULL || tb + u < t1) afr[u] = ld_nt16(Wp + (size_t)(tb + u) * 64);
        }
      }
      constexpr int UB = MTP == 1 ? 8 : (MTP == 2 ? 4 : 2);  // B-fragment reads hoisted per sub-group (VGPR budget: MTP*UB uint4)
#pragma unroll
      for (int uh = 0; uh < U; uh += UB) {
        uint4 bfr[MTP][UB];
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
          for (int mt = 0; mt < MTP; ++mt)
            if (FULL || tb + uh + u < t1)
              bfr[mt][u] = *reinterpret_cast<const uint4*>(brow[mt] + (size_t)(tb + uh + u) * bstep);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if (FULL || tb + uh + u < t1) {
            uint4 af;
            if constexpr (W8) {  // fragment uh + u = half ((uh + u) & 1) of pair (uh + u) >> 1, converted in registers
              const uint4 pr = afr[(uh + u) >> 1];
              af = ((uh + u) & 1) ? e4m3x8_to_bf16x8(pr.z, pr.w) : e4m3x8_to_bf16x8(pr.x, pr.y);
            } else {
              af = afr[uh + u];
            }
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
              if (u & 1) acc2[mt] = MfmaStep<WT>::run(af, bfr[mt][u], acc2[mt]);
              else acc[mt] = MfmaStep<WT>::run(af, bfr[mt][u], acc[mt]);
            }
          }
        }
      }
    }
    // 4. deterministic cross-wave reduction through LDS (fixed wave order)
    PTTS_STAMP(PTTS_DBG(a), 4);
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt)
      *reinterpret_cast<f32x4*>(s_red + (((size_t)wave * MTP + mt) * 64 + lane) * 4) = acc[mt] + acc2[mt];
    __syncthreads();
    for (int mt = wave; mt < MTP; mt += W) {
      f32x4 r = *reinterpret_cast<const f32x4*>(s_red + ((size_t)mt * 64 + lane) * 4);
      for (int w = 1; w < W; ++w) r += *reinterpret_cast<const f32x4*>(s_red + (((size_t)w * MTP + mt) * 64 + lane) * 4);
      if (W8) { r[0] *= wsc4.x; r[1] *= wsc4.y; r[2] *= wsc4.z; r[3] *= wsc4.w; }
      const int mloc = mt * 16 + j;
      const int m = m0 + mloc;
      const int n = strip * 16 + q * 4;  // D[row = (l>>4)*4 + r][col = l&15]
      if (PRO == PRO_COPY && (EPI == EPI_STORE || EPI == EPI_GATE_WT) && a.rs_part && mloc < nrows) {
        const float rs = s_rstd[mloc];
        r[0] *= rs; r[1] *= rs; r[2] *= rs; r[3] *= rs;
      }
      if (mloc < nrows) {
        if (EPI == EPI_STORE) {
          *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n) = make_float4(r[0], r[1], r[2], r[3]);
          if (a.kv_col0 && n >= a.kv_col0) gemm_store_kv_cols<WT>(a, m, n, r);  // (a shape the > 256-row kernels decline)
        } else if (EPI == EPI_GELU) {
          *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n) =
              make_float4(gelu_erf(r[0]), gelu_erf(r[1]), gelu_erf(r[2]), gelu_erf(r[3]));
        } else if (EPI == EPI_GELU_WT) {
          act_store4<WT>(reinterpret_cast<WT*>(a.out), m, n, a.out_ld, a.out_fo, gelu_erf(r[0]), gelu_erf(r[1]), gelu_erf(r[2]), gelu_erf(r[3]));
        } else if (EPI == EPI_GATE_WT) {  // rows n .. n + 3 of the interleaved matrix = (wi_0, wi_1) of features n / 2, n / 2 + 1
          act_store2<WT>(reinterpret_cast<WT*>(a.out), m, n >> 1, a.out_ld, a.out_fo, gelu_new(r[0]) * r[1], gelu_new(r[2]) * r[3]);
        } else if (EPI == EPI_RESID) {
          float4* p = reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n);
          float4 o = mt == wave ? resid_pre : *p;
          o.x += r[0]; o.y += r[1]; o.z += r[2]; o.w += r[3];
          *p = o;
          r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = o.w;  // the updated residual values, for the strip statistics below
        } else {  // EPI_KV: n in [0, 2H): first half K, second half V
          const int H = a.N >> 1;
          const int which = n >= H;
          const int nn = n - which * H;
          const int head = nn >> 6, d = nn & 63;
          const int b = m / a.kv_rows_per_b, t = m - b * a.kv_rows_per_b;
          WT* base = reinterpret_cast<WT*>(which ? a.vcache : a.kcache) +
                     (((size_t)b * a.nheads + head) * a.kv_cap + t) * 64 + d;
#pragma unroll
          for (int e = 0; e < 4; ++e) store_from_f32<WT>(base + e, r[e]);
        }
      }
      if (EPI == EPI_RESID && a.stats_out) {
        // strip statistics of the updated residual rows for the consumer's PRO_LNS prologue: lane (q, j) holds 4 of the strip's
        // 16 columns of row j; the other 12 sit in lanes j + 16 / 32 / 48 (two permlane-swap steps, every lane takes part)
        float sm = (r[0] + r[1]) + (r[2] + r[3]);
        sm = swap32_reduce<OpSum>(swap16_reduce<OpSum>(sm));
        const float mean = sm * 0.0625f;
        const float d0 = r[0] - mean, d1 = r[1] - mean, d2 = r[2] - mean, d3 = r[3] - mean;
        float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        m2 = swap32_reduce<OpSum>(swap16_reduce<OpSum>(m2));
        if (q == 0 && mloc < nrows) reinterpret_cast<float2*>(a.stats_out)[(size_t)m * (a.N >> 4) + strip] = make_float2(mean, m2);
      }
      if (EPI == EPI_RESID && a.nx_out) {  // T5: the next GEMM's operand g o h and this strip's share of sum(h^2) (every lane takes part in the swaps)
        if (mloc < nrows) {
          const float4 g = *reinterpret_cast<const float4*>(a.nx_gamma + n);
          act_store4<WT>(reinterpret_cast<WT*>(a.nx_out), m, n, a.N, a.out_fo, r[0] * g.x, r[1] * g.y, r[2] * g.z, r[3] * g.w);
        }
        float ss = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        ss = swap32_reduce<OpSum>(swap16_reduce<OpSum>(ss));
        if (q == 0 && mloc < nrows) a.ss_out[(size_t)m * (a.N >> 4) + strip] = ss;
      }
    }
    PTTS_STAMP(PTTS_DBG(a), 5);
    __syncthreads();
  }
}

// The two entry points of the strip GEMM. gemm_strip_kernel takes the first 56 bytes of GemmArgs as scalar parameters, which gfx950's command processor
// preloads into SGPRs before the first wave starts (ptts_common.h), and the rest as KTail<GemmArgs>; gemm_strip_kernel_bv takes the struct by value
// behind s_loads, as in rounds 1-5. Same body, same results. Measured on one box, preloaded | by value, us per step at mid context (profiles/
// r06_experiments.txt call 23): 16 utterances 1100.9 | 1142.1, 32: 1200.1 | 1244.6, 64: 1540.2 | 1582.4, 128: 2384.6 | 2431.8, 32 x e4m3 weights
// 1204.6 | 1263.5, Large-v1 x 32 2406.1 | 2457.4 (-1.9 .. -4.7 %). PTTS_STRIP_PRELOAD(PRO, EPI, FULL) picks the entry point per instance at compile
// time: the FULL instances (every decoder width of the released checkpoints) are preloaded, and so are the non-FULL instances on prepared rows
// (PRO_COPY: flan-t5-large's wo projection, K = 2816 = 88 fragments, is not a whole number of 8-fragment groups per wave for any wave count up to 8 - 24
// nodes on the time-to-first-token path). The other non-FULL instances (fused LayerNorm / split-KV / plain prologues at widths no released checkpoint
// has: the small golden specs of the tests) keep the by-value form: one of them - <bf16, PRO_LN, EPI_GELU, 1 tile>, 256 VGPRs and 186 spills - returns
// NaN on the preloaded entry point and is correct by value (bisected over five builds, profiles/r06_strip_preload_bisect.txt: the PRO_COPY, PRO_PLAIN,
// PRO_ATTN and PRO_LN + EPI_STORE families are correct on both; its prologue and s_loads read correct in the ISA; not root-caused).
#ifndef PTTS_STRIP_PRELOAD
#define PTTS_STRIP_PRELOAD(PRO, EPI, FULL) ((FULL) || (PRO) == PRO_COPY)
#endif
template <typename WT, int PRO, int EPI, int MTP, bool FULL, bool W8 = false>
__global__ void __launch_bounds__((GemmMaxThreads<PRO, MTP>::value)) gemm_strip_kernel(PTTS_DBG0_PARAM GemmArgs_KPARAMS) {
  PTTS_STAMP0();
  GemmArgs_KJOIN(a)
  gemm_strip_body<WT, PRO, EPI, MTP, FULL, W8>(a);
}
template <typename WT, int PRO, int EPI, int MTP, bool FULL, bool W8 = false>
__global__ void __launch_bounds__((GemmMaxThreads<PRO, MTP>::value)) gemm_strip_kernel_bv(PTTS_DBG0_PARAM GemmArgs a) {
  PTTS_STAMP0();
  a.kflags = (int)(blockDim.x >> 6) << 8;
  gemm_strip_body<WT, PRO, EPI, MTP, FULL, W8>(a);
}
template <typename WT, int PRO, int EPI, int MTP, bool FULL, bool W8 = false>
inline void launch_strip(dim3 grid, dim3 block, size_t shmem, hipStream_t st, const GemmArgs& a) {
  if constexpr (PTTS_STRIP_PRELOAD(PRO, EPI, FULL)) ptts_klaunch(gemm_strip_kernel<WT, PRO, EPI, MTP, FULL, W8>, grid, block, shmem, st, a);
  else hipLaunchKernelGGL((gemm_strip_kernel_bv<WT, PRO, EPI, MTP, FULL, W8>), grid, block, shmem, st, PTTS_DBG0_ARG(a) a);
}
template <typename WT, int PRO, int EPI, int MTP, bool FULL, bool W8 = false>
inline const void* strip_entry() {
  if constexpr (PTTS_STRIP_PRELOAD(PRO, EPI, FULL)) return reinterpret_cast<const void*>(&gemm_strip_kernel<WT, PRO, EPI, MTP, FULL, W8>);
  else return reinterpret_cast<const void*>(&gemm_strip_kernel_bv<WT, PRO, EPI, MTP, FULL, W8>);
}

// ------------------------------------------------------------------------------------------------------
// gemm_block_kernel: prefill-sized M (> 128 rows, e.g. batch 32 x 33 prompt positions): compute-bound, so every
// fragment has to feed many MFMAs. One wave owns NS weight strips x 4 row tiles (NS*16 x 64 outputs) over the whole K:
// per 32/16-wide k step it loads NS A fragments (weights, already in fragment order) and 4 B fragments (prepared
// engine-dtype rows, PRO_COPY layout) straight from L1/L2 and issues 4*NS MFMAs (0.5 KB of operand loads per MFMA
// at NS = 4, vs 1.1 KB in the strip kernel's 128-row passes); no K split, so no cross-wave reduction. The 4 waves
// of a workgroup take 4 consecutive row groups of the same strips (weight fragments shared through L1).
// Measured motivation: the strip kernel ran the batch-32 prefill GEMMs at 33-96 TFLOP/s (profiles/r01_step_bf16_bs32_v2.txt).
// ------------------------------------------------------------------------------------------------------
template <typename WT, int EPI>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& a, int m, int n, const f32x4& r) {
  if (EPI == EPI_STORE) {
    *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n) = make_float4(r[0], r[1], r[2], r[3]);
    if (a.kv_col0 && n >= a.kv_col0) gemm_store_kv_cols<WT>(a, m, n, r);
  } else if (EPI == EPI_GELU) {
    *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n) = make_float4(gelu_erf(r[0]), gelu_erf(r[1]), gelu_erf(r[2]), gelu_erf(r[3]));
  } else if (EPI == EPI_GELU_WT) {
    WT* o = reinterpret_cast<WT*>(a.out) + (size_t)m * a.out_ld + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) store_from_f32<WT>(o + e, gelu_erf(r[e]));
  } else if (EPI == EPI_GATE_WT) {
    act_store2<WT>(reinterpret_cast<WT*>(a.out), m, n >> 1, a.out_ld, 0, gelu_new(r[0]) * r[1], gelu_new(r[2]) * r[3]);
  } else if (EPI == EPI_RESID) {
    float4* p = reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + n);
    float4 o = *p;
    o.x += r[0]; o.y += r[1]; o.z += r[2]; o.w += r[3];
    *p = o;
  } else {  // EPI_KV: n in [0, 2H): first half K, second half V
    const int H = a.N >> 1;
    const int which = n >= H;
    const int nn = n - which * H;
    const int head = nn >> 6, d = nn & 63;
    const int b = m / a.kv_rows_per_b, t = m - b * a.kv_rows_per_b;
    WT* base = reinterpret_cast<WT*>(which ? a.vcache : a.kcache) + (((size_t)b * a.nheads + head) * a.kv_cap + t) * 64 + d;
#pragma unroll
    for (int e = 0; e < 4; ++e) store_from_f32<WT>(base + e, r[e]);
  }
}

// XCD-aware tile order for the 2-D grids of the prefill-sized GEMMs (round 5). The dispatcher places workgroup b (x fastest) on XCD b % 8, each
// XCD with its own 4 MiB L2: in launch order the 8 neighbours that share a weight panel sit on 8 different L2s and every XCD streams EVERY panel
// of both operands (measured: the batch-32 T5 GEMMs moved ~200 MB per launch at ~3 TB/s - L2-miss bound). Remapped, XCD x owns one contiguous
// (RN x RM) region of the tile grid (RN * RM = 8), so a panel is fetched by ONE XCD. Bijective; grids it cannot split evenly keep launch order
// (a speed choice only: the placement itself is not guaranteed by the runtime).
// (NB, MB given: the caller has the grid extents in preloaded SGPRs - gridDim itself is a hidden argument behind an s_load)
__device__ __forceinline__ void xcd_tile_order(int& bx, int& by, int swz, int NB, int MB) {
  const int total = NB * MB;
  bx = blockIdx.x; by = blockIdx.y;
  if (!swz || total % 8) return;
  int RM = 2, RN = 4;
  if (MB % 2 || NB % 4) {
    if (NB % 8 == 0) { RM = 1; RN = 8; }
    else if (MB % 4 == 0 && NB % 2 == 0) { RM = 4; RN = 2; }
    else if (MB % 8 == 0) { RM = 8; RN = 1; }
    else return;
  }
  const int lin = by * NB + bx, xcd = lin & 7, slot = lin >> 3;
  const int nbr = NB / RN, mbr = MB / RM;  // tiles per region along N / M: slot runs over nbr * mbr = total / 8 of them
  bx = (xcd % RN) * nbr + slot % nbr;
  by = (xcd / RN) * mbr + slot / nbr;
  (void)mbr;
}
__device__ __forceinline__ void xcd_tile_order(int& bx, int& by, int swz) { xcd_tile_order(bx, by, swz, (int)gridDim.x, (int)gridDim.y); }

template <typename WT, int EPI, int NS>
__global__ void __launch_bounds__(256) gemm_block_kernel(GemmArgs a) {
  constexpr int KT = Elem<WT>::KT, MT = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  int bx, by;
  xcd_tile_order(bx, by, a.xcd_swz);
  const int strip0 = bx * NS;
  const int m0 = (by * 4 + wave) * (16 * MT);
  if (m0 >= a.M) return;
  const int nfrag = a.K / KT;
  const uint4* Wp = reinterpret_cast<const uint4*>(a.W) + (size_t)strip0 * nfrag * 64 + lane;
  const char* brow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = min(m0 + mt * 16 + j, a.M - 1);  // clamped rows are computed and dropped
    brow[mt] = reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.x) + (size_t)(m * a.x_row_mul + a.x_row_off) * a.x_ld) + (size_t)q * 16;
  }
  f32x4 acc[NS][MT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 af[NS], bf[MT];
#pragma unroll
  for (int s = 0; s < NS; ++s) af[s] = Wp[(size_t)s * nfrag * 64];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bf[mt] = *reinterpret_cast<const uint4*>(brow[mt]);
  for (int t = 0; t < nfrag; ++t) {
    uint4 an[NS], bn[MT];
    if (t + 1 < nfrag) {  // next step's fragments in flight while this step's MFMAs issue
#pragma unroll
      for (int s = 0; s < NS; ++s) an[s] = Wp[((size_t)s * nfrag + t + 1) * 64];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bn[mt] = *reinterpret_cast<const uint4*>(brow[mt] + (size_t)(t + 1) * (KT * sizeof(WT)));
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[s][mt] = MfmaStep<WT>::run(af[s], bf[mt], acc[s][mt]);
    if (t + 1 < nfrag) {
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = an[s];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = bn[mt];
    }
  }
  if constexpr (EPI == EPI_RESID) {
    // every residual piece is requested BEFORE the first store: interleaved (load, add, store per piece) each load's s_waitcnt vmcnt() also
    // waited for the store in front of it - loads and stores retire in order on one counter (found in the DAC epilogues, round 5)
    float4 res[NS][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = min(m0 + mt * 16 + j, a.M - 1);
        res[s][mt] = *reinterpret_cast<const float4*>(a.out + (size_t)m * a.out_ld + (strip0 + s) * 16 + q * 4);
      }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + mt * 16 + j;
        const f32x4 r = acc[s][mt];
        if (m < a.M)
          *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + (strip0 + s) * 16 + q * 4) =
              make_float4(res[s][mt].x + r[0], res[s][mt].y + r[1], res[s][mt].z + r[2], res[s][mt].w + r[3]);
      }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + j;
      if (m < a.M) gemm_store_tile<WT, EPI>(a, m, (strip0 + s) * 16 + q * 4, acc[s][mt]);  // D[row = q*4 + r][col = j]
    }
}

// ------------------------------------------------------------------------------------------------------
// gemm_tile_kernel (round 5): prefill-sized M (> 256 rows: 32 utterances x 33 prompt positions, 32 descriptions x 64 tokens in the T5 encoder) is
// COMPUTE-bound, and gemm_block_kernel above feeds its MFMAs straight from L1 / L2: 0.5 KB of operand loads per MFMA against a 64 B/clk L1 - it ran
// the batch-32 prefill at ~70 and the T5 encoder at ~215 TFLOP/s (profiles/r05_experiments.txt call 4). This one is the classic LDS-tiled form:
// a workgroup owns BNS weight strips x BMT row tiles (128 x 128 outputs at 8 x 8) over the whole K; per stage of two k fragments every thread copies
// its share of the A tile (weights, already in fragment order: contiguous 1 KiB pieces) and of the B tile (engine-dtype rows, 128 contiguous bytes
// per row, re-ordered to fragment order on the way into LDS) global -> registers -> LDS, double-buffered with ONE barrier per stage; the 2 x 2 waves
// each hold (BNS / 2) x (BMT / 2) accumulator tiles and read every fragment as one conflict-free ds_read_b128 per lane. No K split: deterministic.
// Epilogues: gemm_store_tile (store / residual / GELU / gated GELU / cross K-V scatter).
// ------------------------------------------------------------------------------------------------------
// (staging registers and fragments are ext_vector values, not HIP's uint4 struct: arrays of the struct type assigned under the loop's conditions were
//  kept in scratch memory by the compiler - every global load followed by a scratch store, i.e. no prefetch at all: the 131 TFLOP/s of call 5)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename WT> __device__ __forceinline__ f32x4 mfma_step_v(const u32x4_t& a, const u32x4_t& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma_step_v<bf16_t>(const u32x4_t& a, const u32x4_t& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma_step_v<float>(const u32x4_t& a, const u32x4_t& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  return c;
}

template <typename WT, int EPI, int BNS, int BMT>
__global__ void __launch_bounds__(256) gemm_tile_kernel(GemmArgs a) {
  constexpr int KT = Elem<WT>::KT;
  constexpr int NS = BNS / 2, MT = BMT / 2;              // strips / row tiles per wave
  constexpr int ACH = BNS / 2, BCH = BMT / 2;            // 16-byte pieces per thread and stage (BNS * 128 / 256, BMT * 128 / 256)
  constexpr int STAGE = (BNS + BMT) * 2 * 64;            // 16-byte slots per LDS stage: [BNS strips][2 frags][64] + [BMT tiles][2 frags][64]
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  u32x4_t* sm = reinterpret_cast<u32x4_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 1, wm = wave >> 1;
  const int q = lane >> 4, j = lane & 15;
  int bx, by;
  xcd_tile_order(bx, by, a.xcd_swz);
  const int strip0 = bx * BNS, m0 = by * BMT * 16;
  const int nfrag = a.K / KT, nstage = nfrag >> 1;       // host guarantees an even fragment count
  const u32x4_t* gA[ACH];
  const u32x4_t* gB[BCH];
  int sA[ACH], sB[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int ca = tid + 256 * i, strip = ca >> 7, frag = (ca >> 6) & 1, ln = ca & 63;
    gA[i] = reinterpret_cast<const u32x4_t*>(a.W) + ((size_t)(strip0 + strip) * nfrag + frag) * 64 + ln;
    sA[i] = (strip * 2 + frag) * 64 + ln;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int cb = tid + 256 * i, rl = cb >> 3, kc = cb & 7;  // row of the tile, 16-byte piece of the stage's 2 * KT elements
    const int row = min(m0 + rl, a.M - 1);                    // clamped rows are computed and dropped
    gB[i] = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const WT*>(a.x) + (size_t)(row * a.x_row_mul + a.x_row_off) * a.x_ld) + kc;
    sB[i] = BNS * 128 + ((rl >> 4) * 2 + (kc >> 2)) * 64 + (kc & 3) * 16 + (rl & 15);
  }
  f32x4 acc[NS][MT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // register staging TWO stages ahead (two register sets): a stage's pieces have two stages of MFMA work (and the other resident workgroups') to arrive
  u32x4_t ra0[ACH], rb0[BCH], ra1[ACH], rb1[BCH];
#define PTTS_TILE_FETCH(stg, ra, rb)                                                                                          \
  do {                                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < ACH; ++i) ra[i] = gA[i][(size_t)(stg) * 128]; /* two fragments = 128 slots along the strip */ \
    _Pragma("unroll") for (int i = 0; i < BCH; ++i) rb[i] = gB[i][(size_t)(stg) * 8];   /* two fragments = 8 pieces along the row */   \
  } while (0)
#define PTTS_TILE_COMMIT(stg, ra, rb)                                                \
  do {                                                                               \
    u32x4_t* dst_ = sm + ((stg) & 1) * STAGE;                                        \
    _Pragma("unroll") for (int i = 0; i < ACH; ++i) dst_[sA[i]] = ra[i];             \
    _Pragma("unroll") for (int i = 0; i < BCH; ++i) dst_[sB[i]] = rb[i];             \
  } while (0)
#define PTTS_TILE_COMPUTE(stg)                                                                                                  \
  do {                                                                                                                          \
    const u32x4_t* cur_ = sm + ((stg) & 1) * STAGE;                                                                             \
    _Pragma("unroll") for (int frag = 0; frag < 2; ++frag) {                                                                    \
      u32x4_t af[NS], bf[MT];                                                                                                   \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) af[s] = cur_[((wn * NS + s) * 2 + frag) * 64 + lane];                      \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) bf[mt] = cur_[BNS * 128 + ((wm * MT + mt) * 2 + frag) * 64 + lane];     \
      _Pragma("unroll") for (int s = 0; s < NS; ++s)                                                                            \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mfma_step_v<WT>(af[s], bf[mt], acc[s][mt]);              \
    }                                                                                                                           \
  } while (0)
  PTTS_TILE_FETCH(0, ra0, rb0);
  if (nstage > 1) PTTS_TILE_FETCH(1, ra1, rb1);
  PTTS_TILE_COMMIT(0, ra0, rb0);
  __syncthreads();
  // stage st computes from LDS buffer st & 1; set (st & 1) is free again (its stage-st pieces were committed at the end of stage st - 1): it takes
  // stage st + 2; the other set holds stage st + 1, committed into the other buffer (last read in stage st - 1, behind that stage's barrier)
  for (int st = 0; st < nstage; st += 2) {
    if (st + 2 < nstage) PTTS_TILE_FETCH(st + 2, ra0, rb0);
    PTTS_TILE_COMPUTE(st);
    if (st + 1 < nstage) PTTS_TILE_COMMIT(st + 1, ra1, rb1);
    __syncthreads();
    if (st + 1 >= nstage) break;
    if (st + 3 < nstage) PTTS_TILE_FETCH(st + 3, ra1, rb1);
    PTTS_TILE_COMPUTE(st + 1);
    if (st + 2 < nstage) PTTS_TILE_COMMIT(st + 2, ra0, rb0);
    __syncthreads();
  }
#undef PTTS_TILE_FETCH
#undef PTTS_TILE_COMMIT
#undef PTTS_TILE_COMPUTE
  if constexpr (EPI == EPI_RESID) {  // residual pieces requested before the first store (gemm_block_kernel)
    float4 res[NS][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = min(m0 + (wm * MT + mt) * 16 + j, a.M - 1);
        res[s][mt] = *reinterpret_cast<const float4*>(a.out + (size_t)m * a.out_ld + (strip0 + wn * NS + s) * 16 + q * 4);
      }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + (wm * MT + mt) * 16 + j;
        const f32x4 r = acc[s][mt];
        if (m < a.M)
          *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + (strip0 + wn * NS + s) * 16 + q * 4) =
              make_float4(res[s][mt].x + r[0], res[s][mt].y + r[1], res[s][mt].z + r[2], res[s][mt].w + r[3]);
      }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + (wm * MT + mt) * 16 + j;
      if (m < a.M) gemm_store_tile<WT, EPI>(a, m, (strip0 + wn * NS + s) * 16 + q * 4, acc[s][mt]);  // D[row = q*4 + r][col = j]
    }
}

// ------------------------------------------------------------------------------------------------------
// rows_prep_kernel: for M > 8 rows (batch 32, prefill) the LayerNorm / split-KV combine is computed ONCE here
// (one wave per row) into an engine-dtype [M][K] buffer that the GEMM then stages with plain 16-byte copies
// (PRO_COPY). At M <= 8 the fused prologues win (one graph node less: 1.58 us + a latency chain).
// ------------------------------------------------------------------------------------------------------
// one pass, row held in registers (K == NF4 * 256): every load of the row (+ gamma/beta) is in
// flight at once - one dependent round trip instead of the three of the generic two-pass loop below
template <typename WT, int NF4>
__device__ __forceinline__ void prep_ln_row_regs(const GemmArgs& a, int m, WT* dst, int lane) {
  const float* xr = a.x + (size_t)(m * a.x_row_mul + a.x_row_off) * a.x_ld;
  float4 v[NF4], g[NF4], bt[NF4];
#pragma unroll
  for (int i = 0; i < NF4; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
    bt[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
  }
  const float c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[0].x)));
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const float d0 = v[i].x - c, d1 = v[i].y - c, d2 = v[i].z - c, d3 = v[i].w - c;
    s1 += (d0 + d1) + (d2 + d3);
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float dm = s1 * a.invK, mean = c + dm;
  const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
  for (int i = 0; i < NF4; ++i)
    act_store4<WT>(dst, m, (lane + 64 * i) * 4, a.K, a.out_fo, (v[i].x - mean) * rstd * g[i].x + bt[i].x, (v[i].y - mean) * rstd * g[i].y + bt[i].y,
                   (v[i].z - mean) * rstd * g[i].z + bt[i].z, (v[i].w - mean) * rstd * g[i].w + bt[i].w);
}

// T5LayerNorm (transformers modeling_t5.py T5LayerNorm.forward): variance = mean(x^2) in fp32, x * rsqrt(variance + eps), then * weight.
// NF4 > 0: K == NF4 * 256, the row lives in registers (one round trip); NF4 == 0: any K % 4 == 0, two passes over the (L2-resident) row.
template <typename WT, int NF4>
__device__ __forceinline__ void prep_rms_row(const GemmArgs& a, int m, WT* dst, int lane) {
  const float* xr = a.x + (size_t)(m * a.x_row_mul + a.x_row_off) * a.x_ld;
  const bool keep = !a.row_keep || a.row_keep[m] != 0;
  if constexpr (NF4 > 0) {
    float4 v[NF4], g[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NF4; ++i) g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    ss = wave_sum(ss);
    const float rstd = keep ? rsqrtf(ss * a.invK + a.rms_eps) : 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i)
      act_store4<WT>(dst, m, (lane + 64 * i) * 4, a.K, a.out_fo, (v[i].x * rstd) * g[i].x, (v[i].y * rstd) * g[i].y, (v[i].z * rstd) * g[i].z, (v[i].w * rstd) * g[i].w);
  } else {
    float ss = 0.f;
    for (int k = lane * 4; k < a.K; k += 256) {
      const float4 t = *reinterpret_cast<const float4*>(xr + k);
      ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    ss = wave_sum(ss);
    const float rstd = keep ? rsqrtf(ss * a.invK + a.rms_eps) : 0.f;
    for (int k = lane * 4; k < a.K; k += 256) {
      const float4 t = *reinterpret_cast<const float4*>(xr + k);
      const float4 g = *reinterpret_cast<const float4*>(a.gamma + k);
      act_store4<WT>(dst, m, k, a.K, a.out_fo, (t.x * rstd) * g.x, (t.y * rstd) * g.y, (t.z * rstd) * g.z, (t.w * rstd) * g.w);
    }
  }
}

// Kernel-argument preload (ptts_common.h; call 54): the node takes the first 56 bytes of GemmArgs as scalars like the GEMMs it feeds. It has no weights, so
// launch_prep puts what addresses a wave's first loads into the preloaded slots it does not use - gamma / beta in W / W8, the destination in `out`, the row
// stride in rows_per_pass, x_row_mul | x_row_off << 16 in frags_per_wave, out_fo in out_ld - and the kernel writes them back into the re-assembled struct (in
// registers: the tail's own copies of those fields are never loaded).
template <typename WT, int PRO>
__global__ void __launch_bounds__(256) rows_prep_kernel(PTTS_DBG0_PARAM GemmArgs_KPARAMS) {
  GemmArgs_KJOIN(a)
  a.gamma = reinterpret_cast<const float*>(a.W); a.beta = reinterpret_cast<const float*>(a.W8);
  a.x_ld = a.rows_per_pass; a.x_row_mul = a.frags_per_wave & 0xffff; a.x_row_off = (int)((unsigned)a.frags_per_wave >> 16); a.out_fo = a.out_ld;
  WT* __restrict__ dst = reinterpret_cast<WT*>(a.out);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  if constexpr (PRO == PRO_RMS) {
    if (a.K == 1024) prep_rms_row<WT, 4>(a, m, dst, lane);
    else if (a.K == 512) prep_rms_row<WT, 2>(a, m, dst, lane);
    else if (a.K == 768) prep_rms_row<WT, 3>(a, m, dst, lane);
    else prep_rms_row<WT, 0>(a, m, dst, lane);
    return;
  }
  if (PRO == PRO_LN && a.K == 1024) { prep_ln_row_regs<WT, 4>(a, m, dst, lane); return; }  // Mini-v1
  if (PRO == PRO_LN && a.K == 1536) { prep_ln_row_regs<WT, 6>(a, m, dst, lane); return; }  // Large-v1
  if (PRO == PRO_LN) {
    const float* xr = a.x + (size_t)(m * a.x_row_mul + a.x_row_off) * a.x_ld;
    float s1 = 0.f, s2 = 0.f;
    const float c = xr[0];
    for (int k = lane * 4; k < a.K; k += 256) {
      const float4 t = *reinterpret_cast<const float4*>(xr + k);
      const float d0 = t.x - c, d1 = t.y - c, d2 = t.z - c, d3 = t.w - c;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float dm = s1 * a.invK, mean = c + dm;
    const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
    for (int k = lane * 4; k < a.K; k += 256) {
      const float4 t = *reinterpret_cast<const float4*>(xr + k);
      const float4 g = *reinterpret_cast<const float4*>(a.gamma + k);
      const float4 bt = *reinterpret_cast<const float4*>(a.beta + k);
      act_store4<WT>(dst, m, k, a.K, a.out_fo, (t.x - mean) * rstd * g.x + bt.x, (t.y - mean) * rstd * g.y + bt.y, (t.z - mean) * rstd * g.z + bt.z,
                     (t.w - mean) * rstd * g.w + bt.w);
    }
  } else {
    for (int k = lane * 4; k < a.K; k += 256) {
      const float4 o = stage_elem<PRO>(a, m, k);
      act_store4<WT>(dst, m, k, a.K, a.out_fo, o.x, o.y, o.z, o.w);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Attention over the KV cache, one query row per (b, qi): split over S workgroups x NW waves, online
// softmax per wave, LDS combine per workgroup, unnormalised partial + (max, sumexp) written for the
// consumer GEMM's PRO_ATTN prologue. HBM-bound: K/V rows are read once, 16 B per lane, a whole
// wave-instruction covers RPI consecutive rows = 1 KiB contiguous.
//   self, decode : fused RoPE + KV append of the new position (modeling:858-859, :880-889), causal by length
//   self, prefill: causal (row qi sees positions <= qi); cache filled beforehand by kv_append_kernel
//   cross        : static K/V (:872-875), additive padding mask (:1553-1562) as -inf, q rotated if RoPE (quirk)
// ------------------------------------------------------------------------------------------------------
struct AttnArgs {
  // ---- bytes 0..55: what the PREFILL attention kernels take as scalar kernel parameters (preloaded into SGPRs by the command processor; call 54) ----
  const float* q;      // [rows][q_ld], head h at column h*64
  int q_ld;
  int pre0;            // prefill kernels: P | mask_ld << 16 (launch_prefill_attn)
  const float* knew;   // fused append sources (same row indexing), or null            | prefill kernels: the mask pointer
  const float* vnew;   //                                                              | prefill kernels: Q | kv_heads << 32 | n_rep << 48
  int kv_ld;           //                                                              | prefill kernels: cap
  int pre1;            // prefill kernels: the description length N (cross block) or -1 (self block)
  void* kcache;
  void* vcache;
  // ---- tail ----
  int cap;             // cache capacity in positions
  int kv_bound;        // host-known upper bound of the valid length (<= cap): first-batch rows at or beyond it are not fetched
  const int* cur_len;  // decode: per-batch column count (position = P + cur_len[b] - 1); null in prefill
  const DevDims* dims;
  const int* mask;     // [B][mask_ld] int32 (1 = keep) or null
  int mask_ld;
  const float* cos;    // RoPE tables [max_pos][64] or null
  const float* sin;
  float* part;         // [rows][S][H]
  void* direct_out;    // S == 1: normalised output [rows][H] in the engine dtype (consumer GEMM uses PRO_COPY), or null
  float* stats;        // [rows][S][heads][2]
  int S, Q, nheads, H;
  int kv_heads, n_rep; // grouped-query attention (repeat_kv :280-289): query head h reads K/V head h / n_rep
  int cross;           // 1: length = dims->N, mask over all positions; 0: causal self-attention, mask over positions < P
  int fused_append;
  float scale;
  int out_fo;          // direct_out in MFMA B-fragment order (fo_vec_index) instead of row-major [rows][H]
  float* kscale;       // (KV8 instances) e4m3 self-attention cache: kcache / vcache hold 64 bytes per row and these one power-of-two scale per
  float* vscale;       // (utterance, K/V head, position): [B][kv_heads][cap] fp32, written at append like the rows; null = engine-dtype cache
  int hostP, hostN;    // prefill: the prompt / description lengths of this call as the HOST knows them (dims holds the same numbers on the device)
  PTTS_DBG_FIELD
};
static_assert(sizeof(AttnArgs) % 8 == 0 && offsetof(AttnArgs, cap) == 56, "AttnArgs: 56 preloaded bytes + tail");
// The prefill attention kernels (prefill_attn_kernel, prefill_attn_mfma_kernel) take the first 56 bytes as scalars; they never use knew / vnew / kv_ld, so
// launch_prefill_attn carries in those slots what addresses a wave's first loads - and what the kernels used to fetch through `dims` on the device (a
// second, dependent scalar round trip): P, N, Q, the cache capacity, the mask pointer / row stride and the K/V head geometry. PREFILL_ATTN_JOIN writes them
// back into the re-assembled struct (in registers: the tail's own copies are never loaded) and defines P and NK (N of the cross block, -1 for self).
#define AttnArgs_KPARAMS \
  const float *kq_, int kqld_, int kpre0_, const float *kknew_, const float *kvnew_, int kkvld_, int kpre1_, void *kkc_, void *kvc_, KTail<AttnArgs> kt_
#define PREFILL_ATTN_JOIN(a)                                                                                                          \
  AttnArgs a;                                                                                                                         \
  PTTS_KTAIL_JOIN(AttnArgs, a);                                                                                                       \
  a.q = kq_; a.q_ld = kqld_; a.kcache = kkc_; a.vcache = kvc_; a.cap = kkvld_;                                                        \
  a.mask = reinterpret_cast<const int*>(kknew_); a.mask_ld = (int)((unsigned)kpre0_ >> 16);                                          \
  const int P = kpre0_ & 0xffff, NK = kpre1_;                                                                                         \
  a.cross = NK >= 0;                                                                                                                  \
  {                                                                                                                                   \
    const unsigned long long u_ = reinterpret_cast<unsigned long long>(kvnew_);                                                       \
    a.Q = (int)(u_ & 0xffffffffull); a.kv_heads = (int)((u_ >> 32) & 0xffff); a.n_rep = (int)(u_ >> 48);                              \
  }
template <typename Kn> inline int ptts_launch_prefill_attn_kernel(Kn kern, dim3 grid, dim3 block, hipStream_t st, const AttnArgs& a) {
  if ((unsigned)a.hostP > 0xffffu || (unsigned)a.mask_ld > 0xffffu || (unsigned)a.kv_heads > 0xffffu || (unsigned)a.n_rep > 0xffffu || a.Q < 0)
    return ptts_fail(PTTS_E_UNSUPPORTED, "prefill attention: P %d / mask_ld %d / heads %d x %d do not fit the packed slots", a.hostP, a.mask_ld, a.kv_heads, a.n_rep);
  AttnArgs b = a;
  b.pre0 = (int)((unsigned)a.hostP | ((unsigned)a.mask_ld << 16));
  b.pre1 = a.cross ? a.hostN : -1;
  b.knew = reinterpret_cast<const float*>(a.mask);
  b.vnew = reinterpret_cast<const float*>((unsigned long long)(unsigned)a.Q | ((unsigned long long)a.kv_heads << 32) | ((unsigned long long)a.n_rep << 48));
  b.kv_ld = a.cap;
  hipLaunchKernelGGL(kern, grid, block, 0, st, b.q, b.q_ld, b.pre0, b.knew, b.vnew, b.kv_ld, b.pre1, b.kcache, b.vcache, ptts_ktail(b));
  return PTTS_OK;
}

// load EPL consecutive floats of a row chunk, optionally RoPE-rotated (x*cos + rotate_half(x)*sin, modeling:409-436)
template <int EPL>
__device__ __forceinline__ void load_chunk_rope(const float* row, int d0, const float* cos, const float* sin, size_t pos, float (&out)[EPL]) {
  float x[EPL], y[EPL], cs[EPL], sn[EPL];
  const int dp = d0 < 32 ? d0 + 32 : d0 - 32;  // rotate_half partner chunk (EPL divides 32: a chunk never straddles)
#pragma unroll
  for (int e4 = 0; e4 < EPL / 4; ++e4) {
    const float4 t = reinterpret_cast<const float4*>(row + d0)[e4];
    x[e4 * 4] = t.x; x[e4 * 4 + 1] = t.y; x[e4 * 4 + 2] = t.z; x[e4 * 4 + 3] = t.w;
    if (cos) {
      const float4 u = reinterpret_cast<const float4*>(row + dp)[e4];
      y[e4 * 4] = u.x; y[e4 * 4 + 1] = u.y; y[e4 * 4 + 2] = u.z; y[e4 * 4 + 3] = u.w;
      const float4 c4 = reinterpret_cast<const float4*>(cos + pos * 64 + d0)[e4];
      const float4 s4 = reinterpret_cast<const float4*>(sin + pos * 64 + d0)[e4];
      cs[e4 * 4] = c4.x; cs[e4 * 4 + 1] = c4.y; cs[e4 * 4 + 2] = c4.z; cs[e4 * 4 + 3] = c4.w;
      sn[e4 * 4] = s4.x; sn[e4 * 4 + 1] = s4.y; sn[e4 * 4 + 2] = s4.z; sn[e4 * 4 + 3] = s4.w;
    }
  }
  const float sign = d0 < 32 ? -1.f : 1.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) out[e] = cos ? x[e] * cs[e] + sign * y[e] * sn[e] : x[e];
}

// raw loads of a row chunk and (for RoPE) of its rotate_half partner chunk; the rotation is applied later, once the
// position is known, so that these loads do not wait for the device-resident lengths
template <int EPL>
__device__ __forceinline__ void load_chunk_raw(const float* row, int d0, bool partner, float (&x)[EPL], float (&y)[EPL]) {
  const int dp = d0 < 32 ? d0 + 32 : d0 - 32;
#pragma unroll
  for (int e4 = 0; e4 < EPL / 4; ++e4) {
    const float4 t = reinterpret_cast<const float4*>(row + d0)[e4];
    x[e4 * 4] = t.x; x[e4 * 4 + 1] = t.y; x[e4 * 4 + 2] = t.z; x[e4 * 4 + 3] = t.w;
    if (partner) {
      const float4 u = reinterpret_cast<const float4*>(row + dp)[e4];
      y[e4 * 4] = u.x; y[e4 * 4 + 1] = u.y; y[e4 * 4 + 2] = u.z; y[e4 * 4 + 3] = u.w;
    }
  }
}
template <int EPL>
__device__ __forceinline__ void rope_apply(float (&x)[EPL], const float (&y)[EPL], int d0, const float* cos, const float* sin, size_t pos) {
  if (!cos) return;
  const float sign = d0 < 32 ? -1.f : 1.f;
#pragma unroll
  for (int e4 = 0; e4 < EPL / 4; ++e4) {
    const float4 c4 = reinterpret_cast<const float4*>(cos + pos * 64 + d0)[e4];
    const float4 s4 = reinterpret_cast<const float4*>(sin + pos * 64 + d0)[e4];
    x[e4 * 4 + 0] = x[e4 * 4 + 0] * c4.x + sign * y[e4 * 4 + 0] * s4.x;
    x[e4 * 4 + 1] = x[e4 * 4 + 1] * c4.y + sign * y[e4 * 4 + 1] * s4.y;
    x[e4 * 4 + 2] = x[e4 * 4 + 2] * c4.z + sign * y[e4 * 4 + 2] * s4.z;
    x[e4 * 4 + 3] = x[e4 * 4 + 3] * c4.w + sign * y[e4 * 4 + 3] * s4.w;
  }
}

// ONE dependent round trip per kernel: every global load of a wave's first batch (q chunk, new K/V row, 8 K + 8 V cache
// rows, masks) is issued together with the loads of the device-resident lengths. Addresses are clamped by kv_bound, a kernel
// argument: the cache capacity, or - decode steps - the host's upper bound of the context rounded up to 64 (the step graph is
// captured once per 64-position bucket), so the speculative batch does not fetch rows no utterance can have yet (PMC: 34 MB per
// launch at batch 32 and context 57 against 7.5 MB of live cache); validity against the lengths is applied when the data is used.
// ---- opt-in e4m3 self-attention cache (ptts_config::kv_fp8, engines of more than GV_MAX_ROWS utterances; round 5) -----------------------------
// At 64+ utterances the K/V stream is the bandwidth-bound third of the step and already runs at 6+ TB/s: only fewer bytes
// move it. A cache row is stored as 64 OCP e4m3 bytes + ONE power-of-two fp32 scale per (utterance, K/V head, position), quantised when the row
// is appended (from the fp32 projection, after RoPE): scale = 2^ceil(log2(max|x| / 448)), q = rne_e4m3(x / scale). Attention converts the bytes
// in registers (v_cvt_pk_f32_fp8) and applies the scales to the score / the probability, so every position - the new one included - is seen
// exactly as the cache holds it. NOT a reference mode (modeling_parler_tts.py:3497-3501 raises on quantised caches): its own oracle leg
// (oracle/fp8_oracle.py: quantize_kv_rows), its own tolerance, never the default.
__device__ __forceinline__ float kv8_row_scale(float amax) {  // 2^ceil(log2(amax / 448)), 1 for an all-zero row
  if (!(amax > 0.f)) return 1.f;
  int e;
  const float m = frexpf(amax / 448.0f, &e);  // amax / 448 = m * 2^e, m in [0.5, 1)
  return ldexpf(1.0f, m == 0.5f ? e - 1 : e);
}
__device__ __forceinline__ uint2 kv8_pack8(const float (&x)[8], float inv_scale) {
  int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * inv_scale, x[1] * inv_scale, 0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * inv_scale, x[3] * inv_scale, w0, true);
  int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * inv_scale, x[5] * inv_scale, 0, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * inv_scale, x[7] * inv_scale, w1, true);
  return make_uint2((unsigned)w0, (unsigned)w1);
}
__device__ __forceinline__ void kv8_unpack8(const uint2& v, float (&o)[8]) {
  const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
  const auto c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
  o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1]; o[6] = d[0]; o[7] = d[1];
}
template <bool KV8> struct KvVec { typedef uint4 T; };
template <> struct KvVec<true> { typedef uint2 T; };

template <typename WT, int NW, bool KV8 = false>
__global__ void __launch_bounds__(NW * 64) attn_kernel(AttnArgs a) {
  static_assert(!KV8 || sizeof(WT) == 2, "e4m3 cache: bf16 engine (8 elements per lane)");
  typedef typename KvVec<KV8>::T KVV;
  constexpr int EPL = Elem<WT>::EPL, LPR = 64 / EPL, RPI = 64 / LPR, U = 8;
  __shared__ float s_o[NW][64];
  __shared__ float s_ml[NW][2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z / a.Q, qi = blockIdx.z % a.Q;
  const int row = b * a.Q + qi;
  const int r = lane / LPR, c = lane % LPR;
  const int TW = a.S * NW, wv = s * NW + w;
  const int kvh = h / a.n_rep;

  PTTS_WSTAMP(a, 0);
  // ---- t = 0: all loads ---------------------------------------------------------------------------------------------
  const int P = a.dims->P;
  const int Nenc = a.cross ? a.dims->N : 0;
  const int cl = a.cur_len ? a.cur_len[b] : 0;
  float qv[EPL], qy[EPL], kk[EPL], ky[EPL], vv[EPL], vy[EPL];
  load_chunk_raw<EPL>(a.q + (size_t)row * a.q_ld + h * 64, c * EPL, a.cos != nullptr, qv, qy);
  if (a.fused_append) {
    load_chunk_raw<EPL>(a.knew + (size_t)row * a.kv_ld + kvh * 64, c * EPL, a.cos != nullptr, kk, ky);
    load_chunk_raw<EPL>(a.vnew + (size_t)row * a.kv_ld + kvh * 64, c * EPL, false, vv, vy);
  }
  // row pitch of the cache: 64 elements of WT, or 64 bytes (KV8)
  char* Kc = reinterpret_cast<char*>(a.kcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64 * (KV8 ? 1 : sizeof(WT));
  char* Vc = reinterpret_cast<char*>(a.vcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64 * (KV8 ? 1 : sizeof(WT));
  const KVV* Kb = reinterpret_cast<const KVV*>(Kc);
  const KVV* Vb = reinterpret_cast<const KVV*>(Vc);
  float* Ks = KV8 ? a.kscale + ((size_t)b * a.kv_heads + kvh) * a.cap : nullptr;
  float* Vs = KV8 ? a.vscale + ((size_t)b * a.kv_heads + kvh) * a.cap : nullptr;
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  KVV kf[U], vf[U];
  float ksc[KV8 ? U : 1], vsc[KV8 ? U : 1];
  int mk[U];
  // The first batch is addressed from kernel arguments only (the host's bound kv_bound: rows below it exist in the arena whatever they hold; validity
  // is applied from L below, and every later batch is clamped by L itself). Clamping it by the device-resident length instead (round 5's
  // `exact_len`) measured no difference at 16 / 32 / 128 utterances (profiles/r06_experiments.txt call 14: 1248.4 vs 1248.7 us per step at 32) -
  // the 2.9 us from entry to "first batch requested" in profiles/r06_node_stamps_bs32_bs128.txt is the memory system taking the requests
  // of every workgroup at once, not the two scalar round trips; the flag is gone.
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int t = (wv + u * TW) * RPI + r;
    const int tc = t < a.kv_bound ? t : 0;
    kf[u] = Kb[(size_t)tc * LPR + c];
    vf[u] = Vb[(size_t)tc * LPR + c];
    if constexpr (KV8) { ksc[u] = Ks[tc]; vsc[u] = Vs[tc]; }
    mk[u] = (mrow && t < a.mask_ld) ? mrow[t] : 1;
  }
  PTTS_WSTAMP(a, 1);  // scalar state (dims, cur_len) read and the first batch of K / V rows + the q / k / v chunks requested
  // ---- lengths known from here on -------------------------------------------------------------------------------------
  const int pos = (a.cur_len ? P + cl - 1 : 0) + qi;
  const int L = a.cross ? Nenc : pos + 1;
  const int mask_len = a.cross ? L : P;
  rope_apply<EPL>(qv, qy, c * EPL, a.cos, a.sin, (size_t)pos);
  KVV knew_p = KVV(), vnew_p = KVV();
  float ks_new = 1.f, vs_new = 1.f;
  if (a.fused_append) {
    rope_apply<EPL>(kk, ky, c * EPL, a.cos, a.sin, (size_t)pos);
    const bool writer = s == 0 && w == 0 && r == 0 && h == kvh * a.n_rep;  // single writer of the new cache row (first query head of the group)
    if constexpr (KV8) {
      float ka = 0.f, va = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) { ka = fmaxf(ka, fabsf(kk[e])); va = fmaxf(va, fabsf(vv[e])); }
      ks_new = kv8_row_scale(group_reduce<OpMax, LPR>(ka));  // the row's 64 values sit in the LPR lanes of this row group
      vs_new = kv8_row_scale(group_reduce<OpMax, LPR>(va));
      knew_p = kv8_pack8(kk, 1.0f / ks_new);  // reciprocal of a power of two: exact
      vnew_p = kv8_pack8(vv, 1.0f / vs_new);
      if (writer) {
        reinterpret_cast<uint2*>(Kc + (size_t)pos * 64)[c] = knew_p;
        reinterpret_cast<uint2*>(Vc + (size_t)pos * 64)[c] = vnew_p;
        if (c == 0) { Ks[pos] = ks_new; Vs[pos] = vs_new; }
      }
    } else {
      knew_p = pack16(kk, WT());
      vnew_p = pack16(vv, WT());
      if (writer) {
        reinterpret_cast<uint4*>(Kc + (size_t)pos * 64 * sizeof(WT))[c] = knew_p;
        reinterpret_cast<uint4*>(Vc + (size_t)pos * 64 * sizeof(WT))[c] = vnew_p;
      }
    }
  }
  // softmax in base 2: log2(e) is folded into the query scale, every exponential is ONE v_exp_f32 (expf is a ~15-instruction
  // sequence, 11 of them per lane sat on this kernel's dependent chain); the (max, sumexp) statistics handed to the split-KV
  // combine are therefore in log2 units too (stage_elem<PRO_ATTN>, gv_attn_wave)
  const float qscale = a.scale * 1.44269504088896340736f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) qv[e] *= qscale;

  const int G = (L + RPI - 1) / RPI;
  float m_run = -INFINITY, l_run = 0.f, o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  PTTS_WSTAMP(a, 2);  // the query chunk (the dependent operand: the previous node wrote it) is usable: rotated, scaled

  for (int g0 = wv; g0 < G; g0 += TW * U) {
    bool ok[U];
    if (g0 != wv) {
#pragma unroll
      for (int u = 0; u < U; ++u) {  // loads only: clamped addresses, validity applied afterwards
        const int t = (g0 + u * TW) * RPI + r;
        const int tc = t < L ? t : 0;
        kf[u] = Kb[(size_t)tc * LPR + c];
        vf[u] = Vb[(size_t)tc * LPR + c];
        if constexpr (KV8) { ksc[u] = Ks[tc]; vsc[u] = Vs[tc]; }
        mk[u] = (mrow && tc < mask_len) ? mrow[tc] : 1;
      }
    }
    float sc[U], bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (g0 + u * TW) * RPI + r;
      ok[u] = t < L && (t >= mask_len || mk[u] != 0);
      if (a.fused_append && t == pos) {
        kf[u] = knew_p; vf[u] = vnew_p;
        if constexpr (KV8) { ksc[u] = ks_new; vsc[u] = vs_new; }
      }
      float kx[EPL];
      if constexpr (KV8) kv8_unpack8(kf[u], kx);
      else unpack16(kf[u], kx, WT());
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) d = fmaf(qv[e], kx[e], d);
      d = group_reduce<OpSum, LPR>(d);
      if constexpr (KV8) d *= ksc[u];
      sc[u] = ok[u] ? d : -INFINITY;
      bm = fmaxf(bm, sc[u]);
    }
    bm = across_groups_reduce<OpMax, LPR>(bm);
    const float m_new = fmaxf(m_run, bm);
    if (m_new == -INFINITY) continue;  // wave-uniform: nothing visible yet
    const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = ok[u] ? __builtin_amdgcn_exp2f(sc[u] - m_new) : 0.f;
      float vx[EPL];
      if constexpr (KV8) kv8_unpack8(vf[u], vx);
      else unpack16(vf[u], vx, WT());
      l_run += p;
      const float pv = KV8 ? p * vsc[KV8 ? u : 0] : p;  // the value row's scale rides on the probability
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = ok[u] ? fmaf(pv, vx[e], o[e]) : o[e];  // masked rows may hold NaN/garbage V
    }
    m_run = m_new;
  }
  PTTS_WSTAMP(a, 3);  // K / V loop done
  // reduce over the RPI row slots of the wave (lanes sharing chunk c)
  l_run = across_groups_reduce<OpSum, LPR>(l_run);
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = across_groups_reduce<OpSum, LPR>(o[e]);
  if (r == 0) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_o[w][c * EPL + e] = o[e];
    if (c == 0) { s_ml[w][0] = m_run; s_ml[w][1] = l_run; }
  }
  __syncthreads();
  if (tid < 64) {
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NW; ++i) M = fmaxf(M, s_ml[i][0]);
    float ov = 0.f, lv = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float wgt = (s_ml[i][0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s_ml[i][0] - M);
      ov += wgt * s_o[i][tid];
      lv += wgt * s_ml[i][1];
    }
    if (a.direct_out) {  // unsplit launch: this workgroup saw every key, finish the softmax here
      const int kcol = h * 64 + tid;
      WT* dst = reinterpret_cast<WT*>(a.direct_out);
      if (a.out_fo) dst += fo_vec_index<WT>(row, kcol & ~(EPL - 1), a.H / Elem<WT>::KT) * EPL + (kcol & (EPL - 1));
      else dst += (size_t)row * a.H + kcol;
      store_from_f32<WT>(dst, lv > 0.f ? ov / lv : 0.f);
      PTTS_WSTAMP(a, 5);
      return;
    }
    a.part[((size_t)row * a.S + s) * a.H + h * 64 + tid] = ov;
    if (tid == 0) {
      float* st = a.stats + (((size_t)row * a.S + s) * a.nheads + h) * 2;
      st[0] = M;
      st[1] = lv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// prefill_attn_kernel (round 5): the attention of the PREFILL (Q = P + 1 [+ voice-prompt] positions per utterance, K/V of at most a few tiles),
// self (causal, prompt padding mask) and cross (description mask), as a tiled kernel: one workgroup = 8 consecutive query rows of one
// (utterance, head), 4 waves x 2 queries; the K / V rows of a 64-position tile are read from the cache ONCE per workgroup into LDS (fp32, as the
// cache holds them: engine dtype or e4m3 x row scale) and shared by the 8 queries; lane = key for the scores, lane = head dimension for P V;
// online softmax in base 2 across tiles. attn_kernel ran one workgroup per QUERY ROW here, every one re-reading the utterance's K / V rows:
// 16 896 workgroups and 89 us per launch at 32 utterances x 33 positions, 48 launches = 4.3 of the 10.6 ms of that prefill
// (profiles/r04_step_bf16_bs32_v2.txt). Same semantics as attn_kernel with S = 1 (masked keys carry no weight; a row without a visible key
// yields 0; RoPE on q, also in the cross block - quirk :858 vs :880); output = the normalised context in the engine dtype (direct_out).
// Reference: modeling_parler_tts.py:906-914 (SDPA), :1474-1501 / :1553-1562 (masks).
// ------------------------------------------------------------------------------------------------------
template <typename WT, bool KV8 = false>
__global__ void __launch_bounds__(256) prefill_attn_kernel(AttnArgs_KPARAMS) {
  PREFILL_ATTN_JOIN(a)
  static_assert(!KV8 || sizeof(WT) == 2, "e4m3 cache: bf16 engine");
  constexpr int QW = 2, QB = 4 * QW, EPL = Elem<WT>::EPL;
  __shared__ float sK[64 * 65];
  __shared__ __attribute__((aligned(16))) float sV[64 * 64];
  __shared__ float sQ[QB][64];
  __shared__ float sP[QB][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * QB;
  const int kvh = h / a.n_rep;
  const int Lmax = a.cross ? NK : min(i0 + QB, a.Q);  // keys any query of this workgroup can see (self: causal, position = row index)
  const int mask_len = a.cross ? Lmax : P;
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  const float qscale = a.scale * 1.44269504088896340736f;  // softmax in base 2 (attn_kernel)
  for (int e = tid; e < QB * 64; e += 256) {  // the workgroup's query rows, RoPE-rotated at their own position, pre-scaled
    const int qi = e >> 6, d = e & 63, i = min(i0 + qi, a.Q - 1);
    const float* qr = a.q + (size_t)(b * a.Q + i) * a.q_ld + h * 64;
    float v = qr[d];
    if (a.cos) {
      const float other = d < 32 ? -qr[d + 32] : qr[d - 32];
      v = v * a.cos[(size_t)i * 64 + d] + other * a.sin[(size_t)i * 64 + d];
    }
    sQ[qi][d] = v * qscale;
  }
  const char* Kc = reinterpret_cast<const char*>(a.kcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64 * (KV8 ? 1 : sizeof(WT));
  const char* Vc = reinterpret_cast<const char*>(a.vcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64 * (KV8 ? 1 : sizeof(WT));
  const float* Ks = KV8 ? a.kscale + ((size_t)b * a.kv_heads + kvh) * a.cap : nullptr;
  const float* Vs = KV8 ? a.vscale + ((size_t)b * a.kv_heads + kvh) * a.cap : nullptr;
  float m_run[QW], l_run[QW], o[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) { m_run[q] = -INFINITY; l_run[q] = 0.f; o[q] = 0.f; }
  for (int j0 = 0; j0 < Lmax; j0 += 64) {
    __syncthreads();  // the previous tile is consumed (first pass: sQ is visible)
    // stage the tile: 64 rows x 8 pieces of 8 elements per matrix, one piece per (thread, pass)
    for (int e = tid; e < 64 * 8; e += 256) {
      const int r = e >> 3, c8 = e & 7, j = j0 + r;
      float kx[8], vx[8];
      if (j < Lmax) {
        if constexpr (KV8) {
          const float ks = Ks[j], vs = Vs[j];
          kv8_unpack8(reinterpret_cast<const uint2*>(Kc + (size_t)j * 64)[c8], kx);
          kv8_unpack8(reinterpret_cast<const uint2*>(Vc + (size_t)j * 64)[c8], vx);
#pragma unroll
          for (int i = 0; i < 8; ++i) { kx[i] *= ks; vx[i] *= vs; }
        } else {
          const WT* kr = reinterpret_cast<const WT*>(Kc) + (size_t)j * 64 + c8 * 8;
          const WT* vr = reinterpret_cast<const WT*>(Vc) + (size_t)j * 64 + c8 * 8;
#pragma unroll
          for (int i = 0; i < 8; ++i) { kx[i] = Elem<WT>::ld(kr + i); vx[i] = Elem<WT>::ld(vr + i); }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { kx[i] = 0.f; vx[i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { sK[r * 65 + c8 * 8 + i] = kx[i]; sV[r * 64 + c8 * 8 + i] = vx[i]; }
    }
    __syncthreads();
    const int j = j0 + lane;
    const int mk = (mrow && j < mask_len && j < a.mask_ld) ? mrow[j] : 1;
    float s[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) s[q] = 0.f;
    for (int d = 0; d < 64; ++d) {
      const float kd = sK[lane * 65 + d];
#pragma unroll
      for (int q = 0; q < QW; ++q) s[q] = fmaf(sQ[w * QW + q][d], kd, s[q]);
    }
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      const int i = i0 + w * QW + q;                 // query row = position (prefill)
      const int L = a.cross ? Lmax : min(i, a.Q - 1) + 1;
      const bool ok = j < L && (j >= mask_len || mk != 0);
      const float sc = ok ? s[q] : -INFINITY;
      const float m_new = fmaxf(m_run[q], wave_max(sc));
      float p = 0.f, alpha = 1.f;
      if (m_new != -INFINITY) {  // wave-uniform
        alpha = m_run[q] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run[q] - m_new);
        p = ok ? __builtin_amdgcn_exp2f(sc - m_new) : 0.f;
      }
      l_run[q] = l_run[q] * alpha + wave_sum(p);
      o[q] *= alpha;
      m_run[q] = m_new;
      sP[w * QW + q][lane] = p;
    }
    __syncthreads();
    for (int jj = 0; jj < 64; ++jj) {
      const float v = sV[jj * 64 + lane];
#pragma unroll
      for (int q = 0; q < QW; ++q) o[q] = fmaf(sP[w * QW + q][jj], v, o[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    const int i = i0 + w * QW + q;
    if (i >= a.Q) continue;
    const int row = b * a.Q + i, kcol = h * 64 + lane;
    WT* dst = reinterpret_cast<WT*>(a.direct_out);
    if (a.out_fo) dst += fo_vec_index<WT>(row, kcol & ~(EPL - 1), a.H / Elem<WT>::KT) * EPL + (kcol & (EPL - 1));
    else dst += (size_t)row * a.H + kcol;
    store_from_f32<WT>(dst, l_run[q] > 0.f ? o[q] / l_run[q] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------
// prefill_attn_mfma_kernel (round 6): prefill_attn_kernel's attention on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, an fmaf chain per
// output - the same arithmetic in another summation order) for batches whose (utterance, head) pairs fill the chip: the VALU kernel took 21-22 us
// per launch at 32 utterances x 33 positions, 48 launches = 1.0 of the 4.3 ms of that prefill (profiles/r06_prefill_kernels_bs32_v1.txt).
// One workgroup = up to 64 consecutive query rows of one (utterance, head), one wave = 16 queries x all keys the workgroup can see; key blocks of
// 64 with the K / V tiles in LDS as fp32 (the cache rows converted once per workgroup; 16-byte slots XOR-swizzled by row & 15: conflict-free b128
// reads of K and b32 reads of V without padding). S^T = K Q^T (A = K rows, B = the wave's RoPE-rotated, pre-scaled Q fragments, held in registers):
// lane (i = l & 15, g = l >> 4) holds the scores of query i against keys 16 kt + 4 g + r; base-2 softmax per query across the 4 lanes with the same
// i, online across key blocks; O = P V takes the lane's own probability registers as the A operand (the MFMA sums over g) - no transpose.
// Semantics of prefill_attn_kernel (masked keys carry no weight, a row without a visible key yields 0, q rotated also in the cross block).
// ------------------------------------------------------------------------------------------------------
// eight consecutive cache elements of a row as they are loaded (16 or 32 bytes), converted to fp32 when they are used
template <typename WT> struct Row8;
template <> struct Row8<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float (&o)[8]) const { unpack16(v, o, bf16_t()); }
  __device__ __forceinline__ void fill(int x) { v = make_uint4(0x3c003c00u + x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); }  // probe only
};
template <> struct Row8<float> {
  float4 lo, hi;
  __device__ __forceinline__ void load(const float* p) { lo = *reinterpret_cast<const float4*>(p); hi = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float (&o)[8]) const { o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w; }
  __device__ __forceinline__ void fill(int x) { lo = make_float4(0.01f * x, 0.02f, 0.03f, 0.04f); hi = lo; }  // probe only
};
// NW (round 6, call 38): waves per workgroup at compile time - with blockDim read at run time the K / V staging loop stayed rolled: three rounds of
// load -> wait -> LDS store in a row in front of every key block (profiles/r06_prefill_kernels_bs32_v4.txt: 15.4 us per launch for 33 x 33 / 33 x 64 scores
// per head). Unrolled, all of a thread's K / V pieces are requested before the first is stored. Same arithmetic: bit-identical.
// ABL (tools/attn_probe.hip only; 0 on the product path): ablation bits - 1: no MFMAs, 2: no K / V / mask loads, 4: no LDS staging, 8: no store, 16: exit at
// entry, 32: no query loads
template <typename WT, int NW, int ABL = 0>
__global__ void __launch_bounds__(NW * 64) prefill_attn_mfma_kernel(AttnArgs_KPARAMS) {
  PREFILL_ATTN_JOIN(a)
  if (ABL & 16) return;
  __shared__ __attribute__((aligned(16))) float sK[64 * 64];
  __shared__ __attribute__((aligned(16))) float sV[64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, g = lane >> 4;
  constexpr int nthreads = NW * 64, qwg = NW * 16;  // 1..4 waves of 16 queries
  const int h = blockIdx.y, b = blockIdx.z, i0w = blockIdx.x * qwg, i0 = i0w + w * 16;
  const int kvh = h / a.n_rep;
  const int Lmax = a.cross ? NK : min(i0w + qwg, a.Q);  // keys any query of this workgroup can see (self: causal, position = row index)
  const int mask_len = a.cross ? Lmax : P;
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  const int mlim = mrow ? min(mask_len, a.mask_ld) : 0;  // keys below it have a mask entry
  const float qscale = a.scale * 1.44269504088896340736f;  // softmax in base 2
  const int iq = min(i0 + j, a.Q - 1);                     // clamped queries are computed and dropped
  const int Lq = a.cross ? Lmax : iq + 1;
  const WT* Kc = reinterpret_cast<const WT*>(a.kcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64;
  const WT* Vc = reinterpret_cast<const WT*>(a.vcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64;
  const int* msrc = mrow ? mrow : reinterpret_cast<const int*>(Kc);
  // Every global load below is UNCONDITIONAL on a clamped address and selected afterwards (call 38): with the loads inside per-lane conditions the compiler
  // wrapped each one in a branch with its own s_waitcnt vmcnt(0) - 4 + 6 + 16 dependent round trips per workgroup (q chunks, K / V pieces, mask flags) in
  // front of 128 MFMAs; a wave-uniform branch (rope?, mask?) around a whole group of loads keeps them in flight together.
  // Call 46: the first key block's K / V pieces and mask flags are requested BEFORE the query rows (whose RoPE arithmetic waits for them): one memory
  // round trip in front of the first MFMA instead of two.
  constexpr int NIT = (64 * 8 + nthreads - 1) / nthreads;  // 64 rows x 8 pieces of 8 elements per matrix
  Row8<WT> kraw[NIT], vraw[NIT];  // as loaded: converted on the way into LDS, so that nothing waits for them before the query rows are requested
  int mk[4][4];  // mask flags of this lane's 16 keys, raw (selected against the mask length where they are used)
  auto request = [&](int j0) {
    if (ABL & 2) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) { kraw[it].fill(tid + it); vraw[it].fill(tid - it); }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mk[kt][r] = 1;
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * nthreads, r = e >> 3, c8 = e & 7, key = max(min(j0 + r, Lmax - 1), 0);  // a row of the block that exists
      kraw[it].load(Kc + (size_t)key * 64 + c8 * 8);
      vraw[it].load(Vc + (size_t)key * 64 + c8 * 8);
    }
    // no branch around these (inside `if (mask)` the compiler compared the flags in the branch: 16 waits in front of the query loads): without a mask
    // the same 16 loads read the first bytes of the K row block - a valid address, values ignored
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mk[kt][r] = msrc[max(min(j0 + 16 * kt + 4 * g + r, mlim - 1), 0)];
  };
  request(0);  // unconditional (row 0 exists even for an empty key range): a branch here pulls the flag compares - and their waits - in front of the query loads
  float4 qr[4];
  {
    const float* qrow = a.q + (size_t)(b * a.Q + iq) * a.q_ld + h * 64;
    float4 qv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qv[c] = (ABL & 32) ? make_float4(0.01f * lane, 0.02f, 0.03f * c, 0.04f) : *reinterpret_cast<const float4*>(qrow + 16 * c + 4 * g);
    if (a.cos && !(ABL & 32)) {  // x * cos + rotate_half(x) * sin at the query's own position (modeling:409-436); a 4-element chunk never straddles the halves
      float4 qt[4], cs[4], sn[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d0 = 16 * c + 4 * g;
        qt[c] = *reinterpret_cast<const float4*>(qrow + (d0 < 32 ? d0 + 32 : d0 - 32));
        cs[c] = *reinterpret_cast<const float4*>(a.cos + (size_t)iq * 64 + d0);
        sn[c] = *reinterpret_cast<const float4*>(a.sin + (size_t)iq * 64 + d0);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float sg = 16 * c + 4 * g < 32 ? -1.f : 1.f;
        const float4 v = qv[c], t = qt[c];
        qv[c] = make_float4(v.x * cs[c].x + sg * t.x * sn[c].x, v.y * cs[c].y + sg * t.y * sn[c].y, v.z * cs[c].z + sg * t.z * sn[c].z, v.w * cs[c].w + sg * t.w * sn[c].w);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) qr[c] = make_float4(qv[c].x * qscale, qv[c].y * qscale, qv[c].z * qscale, qv[c].w * qscale);
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float4* sK4 = reinterpret_cast<const float4*>(sK);
  for (int j0 = 0; j0 < Lmax; j0 += 64) {
    if (j0) {
      __syncthreads();  // the previous tiles are consumed
      request(j0);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * nthreads, r = e >> 3, c8 = e & 7;
      const bool live = j0 + r < Lmax;  // rows beyond the visible keys are zeros
      if (e < 64 * 8 && !((ABL & 4) && c8 != 7)) {
        const int s0 = (2 * c8) ^ (r & 15), s1 = (2 * c8 + 1) ^ (r & 15);
        float kx[8], vx[8];
        kraw[it].get(kx);
        vraw[it].get(vx);
        reinterpret_cast<float4*>(sK)[r * 16 + s0] = live ? make_float4(kx[0], kx[1], kx[2], kx[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(sK)[r * 16 + s1] = live ? make_float4(kx[4], kx[5], kx[6], kx[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(sV)[r * 16 + s0] = live ? make_float4(vx[0], vx[1], vx[2], vx[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(sV)[r * 16 + s1] = live ? make_float4(vx[4], vx[5], vx[6], vx[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    bool ok[4][4];  // key visible to this lane's query: inside its causal / description length and not padding (no mask entry: kept)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = j0 + 16 * kt + 4 * g + r;
        ok[kt][r] = (key < Lq) & ((key >= mlim) | (mk[kt][r] != 0));  // bitwise: && / || became 16 branches with a wait each; beyond mlim there is no mask entry
      }
    if (!(ABL & 4)) __syncthreads();
    f32x4 st[4];
    float4 vb[4][4];
    if (ABL & 1) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const float4 kr = sK4[(16 * kt + j) * 16 + (g ^ j)];
        st[kt] = f32x4{kr.x * qr[0].x, kr.y * qr[1].y, kr.z * qr[2].z, kr.w * qr[3].w};
      }
    } else {
      attn_block_scores(sK4, j, g, qr, st);
    }
    attn_block_v_request(reinterpret_cast<const float4*>(sV), j, g, vb);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, ok[kt][r] ? st[kt][r] : -INFINITY);
    const float m_new = fmaxf(m_run, across_groups_reduce<OpMax, 16>(mx));
    const float alpha = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);  // m_run finite implies m_new finite
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = ok[kt][r] ? __builtin_amdgcn_exp2f(st[kt][r] - m_new) : 0.f;  // ok implies a finite score <= m_new
        st[kt][r] = pv;
        sum += pv;
      }
    l_run = l_run * alpha + across_groups_reduce<OpSum, 16>(sum);
    m_run = m_new;
    if (j0) {  // the accumulators hold queries 4 g + r; their factors live in the lanes whose l & 15 is that query
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ar = __shfl(alpha, 4 * g + r);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt][r] *= ar;
      }
    }
    if (ABL & 1) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[0][r] += st[kt][r] * vb[kt][r].x; o[1][r] += st[kt][r] * vb[kt][r].y; o[2][r] += st[kt][r] * vb[kt][r].z; o[3][r] += st[kt][r] * vb[kt][r].w; }
    } else {
      attn_block_pv(st, vb, o);
    }
  }
  WT* dst0 = reinterpret_cast<WT*>(a.direct_out);
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // o[dt][r]: query 4 g + r, column 4 j + dt of the head - four consecutive elements per lane
    const float lr = __shfl(l_run, 4 * g + r);
    const int i = i0 + 4 * g + r;
    if (i >= a.Q) continue;
    if ((ABL & 8) && o[0][r] != 12345.f) continue;
    const bool any = lr > 0.f;  // a row without a visible key yields 0
    act_store4<WT>(dst0, b * a.Q + i, h * 64 + 4 * j, a.H, a.out_fo, any ? o[0][r] / lr : 0.f, any ? o[1][r] / lr : 0.f, any ? o[2][r] / lr : 0.f,
                   any ? o[3][r] / lr : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------
// xattn_fused_kernel (decode, batch <= 8): encoder_attn_layer_norm + cross q projection + cross-attention for ONE head
// per workgroup (modeling:1040-1052, :855-859, :872-875, :906-914). The description K/V is static and short, so the
// whole chain is head-parallel: 8 waves = 4 weight strips (the head's 64 q rows) x 2 K halves for the projection,
// then wave b runs the single-query attention of utterance b. Replaces two graph nodes (LN+GEMM, attention) and one
// cross-XCD round trip by one; output is the normalised context in the engine dtype, read by the out_proj GEMM
// (PRO_COPY). RoPE quirk kept: q rotated, keys not (:858-859 vs :880).
// ------------------------------------------------------------------------------------------------------
struct XAttnArgs {
  // ---- bytes 0..55: scalar kernel parameters, preloaded into SGPRs by the command processor (ptts_common.h; round 6) ----
  const void* W;       // packed cross q_proj [H/16][K/KT][64][16 B]
  const float* x;      // residual stream h [B][x_ld]
  const float* gamma;
  void* kcache;        // cross K/V [B][heads][cap][64]
  void* vcache;
  int x_ld;
  int K;               // hidden size
  int B;
  int cap;
  // ---- tail (KTail<XAttnArgs>) ----
  int x_row_mul, x_row_off;
  const float* beta;
  float invK;
  int mask_ld;
  const int* cur_len;
  const DevDims* dims;
  const int* mask;     // [B][mask_ld] or null
  const float* cos;
  const float* sin;
  void* out;           // [B][K] engine dtype
  int nheads;
  int kv_heads, n_rep; // cross K/V heads (grouped-query attention)
  float scale;
  int out_fo;          // out in MFMA B-fragment order (fo_vec_index) for the out_proj GEMM at batch > 8
  int xa_pad_;
  PTTS_DBG_FIELD
};
static_assert(sizeof(XAttnArgs) % 8 == 0 && offsetof(XAttnArgs, x_row_mul) == 56, "XAttnArgs: 56 preloaded bytes + tail");
#define XAttnArgs_KPARAMS \
  const void *kW_, const float *kx_, const float *kg_, void *kkc_, void *kvc_, int kxld_, int kK_, int kB_, int kcap_, KTail<XAttnArgs> kt_
#define XAttnArgs_KJOIN(a)                                                                                   \
  XAttnArgs a;                                                                                               \
  PTTS_KTAIL_JOIN(XAttnArgs, a);                                                                             \
  a.W = kW_; a.x = kx_; a.gamma = kg_; a.kcache = kkc_; a.vcache = kvc_; a.x_ld = kxld_; a.K = kK_; a.B = kB_; a.cap = kcap_;
template <typename Kn> inline void ptts_klaunch(Kn kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const XAttnArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, shmem, st, a.W, a.x, a.gamma, a.kcache, a.vcache, a.x_ld, a.K, a.B, a.cap, ptts_ktail(a));
}

// G = utterances per workgroup (8: one per wave; 4 / 2: at batch > 8 the launch covers heads x ceil(B / G) workgroups - 128 / 256 at 32
// utterances instead of 64 - and the 8 / G waves of an utterance split the description's row groups and merge through LDS).
template <typename WT, int UW, int NF4, int G = 8>
__global__ void __launch_bounds__(512) xattn_fused_kernel(XAttnArgs_KPARAMS) {
  XAttnArgs_KJOIN(a)
  constexpr int KT = Elem<WT>::KT, EPL = Elem<WT>::EPL, LPR = 64 / EPL, RPI = 64 / LPR, U = 8, NWV = 8;
  constexpr int WPU = NWV / G, UA = U / WPU;  // waves per utterance in the attention phase; row groups per wave and batch
  static_assert(G == 8 || G == 4 || G == 2, "utterances per workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.x;
  // utterances b0 .. b0 + nb - 1 of this workgroup: one group of <= G per blockIdx.y
  const int b0 = blockIdx.y * G, nb = min(G, a.B - b0), nbmax = min(G, a.B);
  const int row_bytes = a.K * (int)sizeof(WT) + 16;
  char* s_x = smem_raw;                                                        // [G][row_bytes]
  float* s_red = reinterpret_cast<float*>(smem_raw + (size_t)nbmax * row_bytes);  // [8 waves][64][4]
  float* s_q = s_red + NWV * 256;                                              // [G][64]
  float* s_o = s_q + G * 64;                                                   // [8 waves][64]  (WPU > 1: partial contexts)
  float* s_ml = s_o + NWV * 64;                                                // [8 waves][2]
  const int nfrag = a.K / KT;
  const int strip = h * 4 + (wave >> 1);
  const int per = nfrag >> 1;  // host guarantees per % UW == 0
  const int t0 = (wave & 1) * per, t1 = t0 + per;
  const uint4* Wp = reinterpret_cast<const uint4*>(a.W) + (size_t)strip * nfrag * 64 + lane;
  const int q4 = lane >> 4, j = lane & 15;
  const int r = lane / LPR, c = lane % LPR;
  const int ul = wave / WPU, sub = wave % WPU;  // attention: utterance slot and row-group phase of this wave
  const int b = b0 + min(ul, nb - 1);
  PTTS_WSTAMP(a, 0);
  const int N = a.dims->N;

  // ---- t = 0: every independent global load of the kernel goes in flight; the residual rows first (critical path:
  // LayerNorm -> projection), then the bulk loads nobody waits for yet. Addresses are clamped by the cache capacity
  // (a kernel argument), NOT by dims->N (device memory: would put a dependent round trip in front of every K/V load).
  uint4 afr[UW];
  const uint4* Kb = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.kcache) + ((size_t)b * a.kv_heads + h / a.n_rep) * a.cap * 64);
  const uint4* Vb = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.vcache) + ((size_t)b * a.kv_heads + h / a.n_rep) * a.cap * 64);
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  uint4 kf[UA], vf[UA];
  int mk[UA];
  auto issue_bulk = [&](int stage) __attribute__((always_inline)) {
    if (stage != 1) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();  // no fence: the residual-row loads of every wave are queued before the bulk loads
    }
    if (stage != 0) {
#pragma unroll
      for (int u = 0; u < UW; ++u) afr[u] = ld_nt16(Wp + (size_t)(t0 + u) * 64);
#pragma unroll
      for (int u = 0; u < UA; ++u) {  // the utterance's first 8 row groups over its WPU waves (covers N <= 64 bf16 / 32 fp32 in one batch)
        const int t = (u * WPU + sub) * RPI + r;
        const int tc = t < a.cap ? t : 0;
        kf[u] = Kb[(size_t)tc * LPR + c];
        vf[u] = Vb[(size_t)tc * LPR + c];
        mk[u] = (mrow && t < a.mask_ld) ? mrow[t] : 1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- LayerNorm of the rows -> LDS, then the head's 64 q rows ------------------------------------------------------
  ln_stage<WT, NF4, true, XAttnArgs, false>(a, b0, nb, s_x, row_bytes, lane, wave, NWV, issue_bulk);  // K == NF4 * 256
  PTTS_WSTAMP(a, 2);  // this wave's rows normalised
  __syncthreads();
  PTTS_WSTAMP(a, 3);  // every row of the group is in LDS
  const char* brow = s_x + (size_t)min(j, nb - 1) * row_bytes + (size_t)q4 * 16;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tb = t0; tb < t1; tb += UW) {
    if (tb != t0) {
#pragma unroll
      for (int u = 0; u < UW; ++u) afr[u] = ld_nt16(Wp + (size_t)(tb + u) * 64);
    }
#pragma unroll
    for (int uh = 0; uh < UW; uh += 8) {
      uint4 bfr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) bfr[u] = *reinterpret_cast<const uint4*>(brow + (size_t)(tb + uh + u) * (KT * sizeof(WT)));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u & 1) acc2 = MfmaStep<WT>::run(afr[uh + u], bfr[u], acc2);
        else acc = MfmaStep<WT>::run(afr[uh + u], bfr[u], acc);
      }
    }
  }
  PTTS_WSTAMP(a, 4);  // last weight fragment consumed
  *reinterpret_cast<f32x4*>(s_red + ((size_t)wave * 64 + lane) * 4) = acc + acc2;
  __syncthreads();
  if (wave < 4) {  // wave s combines the two K halves of strip s: D[row = q4*4 + e][col = utterance j]
    const f32x4 rr = *reinterpret_cast<const f32x4*>(s_red + ((size_t)(2 * wave) * 64 + lane) * 4) +
                     *reinterpret_cast<const f32x4*>(s_red + ((size_t)(2 * wave + 1) * 64 + lane) * 4);
    if (j < nb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s_q[j * 64 + wave * 16 + q4 * 4 + e] = rr[e];
    }
  }
  __syncthreads();
  PTTS_WSTAMP(a, 5);  // the head's 64 q rows of every utterance of the group are in LDS
  if (WPU == 1 && ul >= nb) return;
  // ---- wave (b, sub): single-query attention of utterance b over its share of the N description positions ---------------
  const bool live = ul < nb;
  float qv[EPL];
  {
    const float* qs = s_q + (b - b0) * 64;
    const int d0 = c * EPL;
    if (a.cos) {
      const size_t pos = (size_t)(a.dims->P + a.cur_len[b] - 1);
      const int dp = d0 < 32 ? d0 + 32 : d0 - 32;
      const float sign = d0 < 32 ? -1.f : 1.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        qv[e] = (qs[d0 + e] * a.cos[pos * 64 + d0 + e] + sign * qs[dp + e] * a.sin[pos * 64 + d0 + e]) * a.scale;
    } else {
#pragma unroll
      for (int e = 0; e < EPL; ++e) qv[e] = qs[d0 + e] * a.scale;
    }
  }
  const int NG = (N + RPI - 1) / RPI;
  float m_run = -INFINITY, l_run = 0.f, o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  for (int g0 = 0; g0 < NG; g0 += U) {
    bool ok[UA];
    if (g0 != 0) {
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const int t = (g0 + u * WPU + sub) * RPI + r;
        const int tc = t < N ? t : 0;
        kf[u] = Kb[(size_t)tc * LPR + c];
        vf[u] = Vb[(size_t)tc * LPR + c];
        mk[u] = mrow ? mrow[tc] : 1;
      }
    }
    float sc[UA], bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      ok[u] = ((g0 + u * WPU + sub) * RPI + r) < N && mk[u] != 0;
      float kk[EPL];
      unpack16(kf[u], kk, WT());
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) d = fmaf(qv[e], kk[e], d);
      d = group_reduce<OpSum, LPR>(d);
      sc[u] = ok[u] ? d : -INFINITY;
      bm = fmaxf(bm, sc[u]);
    }
    bm = across_groups_reduce<OpMax, LPR>(bm);
    const float m_new = fmaxf(m_run, bm);
    if (m_new == -INFINITY) continue;
    const float alpha = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const float p = ok[u] ? expf(sc[u] - m_new) : 0.f;
      float vv[EPL];
      unpack16(vf[u], vv, WT());
      l_run += p;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = ok[u] ? fmaf(p, vv[e], o[e]) : o[e];
    }
    m_run = m_new;
  }
  l_run = across_groups_reduce<OpSum, LPR>(l_run);
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = across_groups_reduce<OpSum, LPR>(o[e]);
  if constexpr (WPU > 1) {  // merge the WPU partial softmaxes of an utterance (fixed wave order: deterministic)
    if (r == 0) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) s_o[wave * 64 + c * EPL + e] = o[e];
      if (c == 0) { s_ml[wave * 2] = m_run; s_ml[wave * 2 + 1] = l_run; }
    }
    PTTS_WSTAMP(a, 6);  // this wave's share of the description attended
    __syncthreads();
    if (sub != 0 || !live) return;
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < WPU; ++i) M = fmaxf(M, s_ml[(wave + i) * 2]);
    l_run = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
    for (int i = 0; i < WPU; ++i) {
      const float mi = s_ml[(wave + i) * 2];
      const float wgt = (mi == -INFINITY) ? 0.f : expf(mi - M);
      l_run += wgt * s_ml[(wave + i) * 2 + 1];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] += wgt * s_o[(wave + i) * 64 + c * EPL + e];
    }
  }
  if (r == 0) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    float res[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) res[e] = o[e] * inv;
    if (a.out_fo) reinterpret_cast<uint4*>(a.out)[fo_vec_index<WT>(b, h * 64 + c * EPL, a.K / KT)] = pack16(res, WT());
    else reinterpret_cast<uint4*>(reinterpret_cast<WT*>(a.out) + (size_t)b * a.K + h * 64)[c] = pack16(res, WT());
  }
  PTTS_WSTAMP(a, 7);
}

// ------------------------------------------------------------------------------------------------------
// lnproj_fused_kernel (decode, batch > 8): LayerNorm + a 64-row block of a projection for a
// GROUP of G <= 8 utterances per workgroup - the front half of xattn_fused_kernel as a node of its own, for LN1 + QKV (:1020-1021, :848-850) and
// LN3 + fc1 + GELU (:1059-1060). The strip GEMMs tile N only (16 weight rows x ALL M rows per workgroup): a LayerNorm prologue there would
// normalise every row in every one of the N / 16 workgroups (88 % of the GEMM at M = 32, round 1), so LN1 / LN3 ran as a rows_prep node in
// front of the GEMM: 4.9 + ~1.5 (boundary) us per layer at batch 32, 2 x 5.0 + gaps at 128. Tiled over N AND M (64 rows x G utterances, full K)
// a workgroup normalises only its own G rows (redundant over the N / 64 row blocks: G x 4 KB each, from the L2) and re-streams its 128 KB of
// weights once per utterance group, also from the L2: the prep node and its boundary are gone and the arithmetic per row is prep_ln_row_regs'
// (same fold order, same one-pass shifted statistics), so the normalised rows are bit-identical to the two-node path.
// 8 waves = 4 weight strips x 2 K halves; wave w < nb normalises row b0 + w first.
// EPI_STORE: fp32 [M][out_ld] (QKV); EPI_GELU_WT: gelu_erf in the engine dtype, row-major or MFMA B-fragment order (fc1 -> fc2).
// ------------------------------------------------------------------------------------------------------
struct LnProjArgs {
  // ---- bytes 0..55: scalar kernel parameters, preloaded into SGPRs by the command processor (ptts_common.h; round 6: the node is on the path of every
  // decode step above 8 utterances twice per layer and of a short prompt's prefill three times per layer) ----
  const void* W;        // packed strips [N/16][K/KT][64][16 B]
  const float* x;       // residual stream h [M][x_ld]
  const float* gamma;
  const float* beta;
  void* out;
  int x_ld;
  int K;                // hidden size (= NF4 * 256)
  int M, N;
  // ---- tail (KTail<LnProjArgs>) ----
  float invK;
  int out_ld;
  int out_fo;           // EPI_GELU_WT: B-fragment order for the consumer GEMM
  // EPI_STORE at prefill (round 5): the QKV node also writes the K / V columns of its rows into the self-attention cache in the engine dtype
  // (kv_append_kernel's store for sinusoidal positions and an engine-dtype cache: one node and one kernel boundary less per layer on the
  // time-to-first-token path); null = no append
  int kv_Q;             // rows per utterance (row m = utterance m / kv_Q, position m % kv_Q)
  void* kcache;         // [B][kv_heads][kv_cap][64]
  void* vcache;
  int kv_cap, kv_heads, kv_H, kv_pad_;  // columns [kv_H, kv_H + 64 kv_heads) = K, then V
  PTTS_DBG_FIELD
};
static_assert(sizeof(LnProjArgs) % 8 == 0 && offsetof(LnProjArgs, invK) == 56, "LnProjArgs: 56 preloaded bytes + tail");
#define LnProjArgs_KPARAMS \
  const void *kW_, const float *kx_, const float *kg_, const float *kb_, void *kout_, int kxld_, int kK_, int kM_, int kN_, KTail<LnProjArgs> kt_
#define LnProjArgs_KJOIN(a)                                                                                  \
  LnProjArgs a;                                                                                              \
  PTTS_KTAIL_JOIN(LnProjArgs, a);                                                                            \
  a.W = kW_; a.x = kx_; a.gamma = kg_; a.beta = kb_; a.out = kout_; a.x_ld = kxld_; a.K = kK_; a.M = kM_; a.N = kN_;
template <typename Kn> inline void ptts_klaunch(Kn kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const LnProjArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, shmem, st, PTTS_DBG0_ARG(a) a.W, a.x, a.gamma, a.beta, a.out, a.x_ld, a.K, a.M, a.N, ptts_ktail(a));
}

// G = 16 (round 5): two rows per wave (rows w and w + 8, both in flight), all 16 columns of the MFMA tile in use - half the weight re-reads of
// G = 8 per utterance (the L2 traffic of the strip GEMM it replaces at 64..128 utterances).
template <typename WT, int UW, int NF4, int G, int EPI>
__global__ void __launch_bounds__(512) lnproj_fused_kernel(PTTS_DBG0_PARAM LnProjArgs_KPARAMS) {
  PTTS_STAMP0();
  LnProjArgs_KJOIN(a)
  constexpr int KT = Elem<WT>::KT, NWV = 8;
  static_assert(G <= 2 * NWV, "at most two rows per wave");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b0 = blockIdx.y * G, nb = min(G, a.M - b0), nbmax = min(G, a.M);
  const int row_bytes = a.K * (int)sizeof(WT) + 16;
  char* s_x = smem_raw;                                                           // [G][row_bytes]
  float* s_red = reinterpret_cast<float*>(smem_raw + (size_t)nbmax * row_bytes);  // [8 waves][64][4]
  const int nfrag = a.K / KT;
  const int strip = blockIdx.x * 4 + (wave >> 1);
  const int per = nfrag >> 1;  // host guarantees per % UW == 0
  const int t0 = (wave & 1) * per, t1 = t0 + per;
  const uint4* Wp = reinterpret_cast<const uint4*>(a.W) + (size_t)strip * nfrag * 64 + lane;
  const int q4 = lane >> 4, j = lane & 15;
  uint4 afr[UW];
  PTTS_WSTAMP(a, 0);
  if constexpr (G > NWV) {  // two rows per wave
    if (wave < nb) {
      const bool two = wave + NWV < nb;
      const float* xr0 = a.x + (size_t)(b0 + wave) * a.x_ld;
      const float* xr1 = a.x + (size_t)(b0 + (two ? wave + NWV : wave)) * a.x_ld;
      float4 v0[NF4], v1[NF4], g[NF4], bt[NF4];
#pragma unroll
      for (int i = 0; i < NF4; ++i) { v0[i] = *reinterpret_cast<const float4*>(xr0 + (lane + 64 * i) * 4); v1[i] = *reinterpret_cast<const float4*>(xr1 + (lane + 64 * i) * 4); }
#pragma unroll
      for (int i = 0; i < NF4; ++i) {
        g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
        bt[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int uu = 0; uu < UW; ++uu) afr[uu] = ld_nt16(Wp + (size_t)(t0 + uu) * 64);
      __builtin_amdgcn_sched_barrier(0);
      auto norm_row = [&](float4 (&v)[NF4], char* row) __attribute__((always_inline)) {  // prep_ln_row_regs' arithmetic
        const float c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[0].x)));
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
          const float d0 = v[i].x - c, d1 = v[i].y - c, d2 = v[i].z - c, d3 = v[i].w - c;
          s1 += (d0 + d1) + (d2 + d3);
          s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float dm = s1 * a.invK, mean = c + dm;
        const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
        for (int i = 0; i < NF4; ++i)
          lds_store4<WT>(row, (lane + 64 * i) * 4, (v[i].x - mean) * rstd * g[i].x + bt[i].x, (v[i].y - mean) * rstd * g[i].y + bt[i].y,
                         (v[i].z - mean) * rstd * g[i].z + bt[i].z, (v[i].w - mean) * rstd * g[i].w + bt[i].w);
      };
      PTTS_WSTAMP(a, 1);  // rows + gamma / beta + the first weight fragments requested
      norm_row(v0, s_x + (size_t)wave * row_bytes);
      if (two) norm_row(v1, s_x + (size_t)(wave + NWV) * row_bytes);
      PTTS_WSTAMP(a, 2);  // this wave's rows normalised (the dependent operand has arrived and been reduced)
    } else {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int uu = 0; uu < UW; ++uu) afr[uu] = ld_nt16(Wp + (size_t)(t0 + uu) * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  // ---- t = 0: the residual row (+ gamma / beta) of this wave first, then - behind a rendezvous, so that no wave's row
  // queues behind another wave's weights - the first UW weight fragments
  if (wave < nb) {
    const int m = b0 + wave;
    const float* xr = a.x + (size_t)m * a.x_ld;
    float4 v[NF4], g[NF4], bt[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
      bt[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int uu = 0; uu < UW; ++uu) afr[uu] = ld_nt16(Wp + (size_t)(t0 + uu) * 64);
    __builtin_amdgcn_sched_barrier(0);
    PTTS_WSTAMP(a, 1);  // row + gamma / beta + the first weight fragments requested
    const float c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[0].x)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const float d0 = v[i].x - c, d1 = v[i].y - c, d2 = v[i].z - c, d3 = v[i].w - c;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float dm = s1 * a.invK, mean = c + dm;
    const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
    char* row = s_x + (size_t)wave * row_bytes;
#pragma unroll
    for (int i = 0; i < NF4; ++i)
      lds_store4<WT>(row, (lane + 64 * i) * 4, (v[i].x - mean) * rstd * g[i].x + bt[i].x, (v[i].y - mean) * rstd * g[i].y + bt[i].y,
                     (v[i].z - mean) * rstd * g[i].z + bt[i].z, (v[i].w - mean) * rstd * g[i].w + bt[i].w);
    PTTS_WSTAMP(a, 2);  // this wave's row normalised
  } else {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int uu = 0; uu < UW; ++uu) afr[uu] = ld_nt16(Wp + (size_t)(t0 + uu) * 64);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  PTTS_WSTAMP(a, 3);  // every row of the group is in LDS
  // ---- the block's 64 projection rows for the group's utterances (columns j < nb of the MFMA tile)
  const char* brow = s_x + (size_t)min(j, nb - 1) * row_bytes + (size_t)q4 * 16;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tb = t0; tb < t1; tb += UW) {
    if (tb != t0) {
#pragma unroll
      for (int uu = 0; uu < UW; ++uu) afr[uu] = ld_nt16(Wp + (size_t)(tb + uu) * 64);
    }
#pragma unroll
    for (int uh = 0; uh < UW; uh += 8) {
      uint4 bfr[8];
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) bfr[uu] = *reinterpret_cast<const uint4*>(brow + (size_t)(tb + uh + uu) * (KT * sizeof(WT)));
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) {
        if (uu & 1) acc2 = MfmaStep<WT>::run(afr[uh + uu], bfr[uu], acc2);
        else acc = MfmaStep<WT>::run(afr[uh + uu], bfr[uu], acc);
      }
    }
  }
  PTTS_WSTAMP(a, 4);  // last weight fragment consumed
  *reinterpret_cast<f32x4*>(s_red + ((size_t)wave * 64 + lane) * 4) = acc + acc2;
  __syncthreads();
  PTTS_WSTAMP(a, 5);  // cross-wave reduction buffer complete
  if (wave < 4 && j < nb) {  // wave s combines the two K halves of strip s: D[row = q4*4 + e][col = utterance j]
    const f32x4 rr = *reinterpret_cast<const f32x4*>(s_red + ((size_t)(2 * wave) * 64 + lane) * 4) +
                     *reinterpret_cast<const f32x4*>(s_red + ((size_t)(2 * wave + 1) * 64 + lane) * 4);
    const int m = b0 + j, n = (blockIdx.x * 4 + wave) * 16 + q4 * 4;
    if (EPI == EPI_STORE) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.out_ld + n) = make_float4(rr[0], rr[1], rr[2], rr[3]);
      if (a.kcache && n >= a.kv_H) {  // K / V columns: the cache row of (utterance, head, position), rounded as kv_append_kernel rounds it
        const int Hkv = a.kv_heads * 64, nn = n - a.kv_H, isv = nn >= Hkv, n2 = isv ? nn - Hkv : nn;
        const int head = n2 >> 6, d = n2 & 63, bu = m / a.kv_Q, pos = m - bu * a.kv_Q;
        WT* dst = reinterpret_cast<WT*>(isv ? a.vcache : a.kcache) + (((size_t)bu * a.kv_heads + head) * a.kv_cap + pos) * 64 + d;
#pragma unroll
        for (int e = 0; e < 4; ++e) store_from_f32<WT>(dst + e, rr[e]);
      }
    } else {
      act_store4<WT>(reinterpret_cast<WT*>(a.out), m, n, a.out_ld, a.out_fo, gelu_erf(rr[0]), gelu_erf(rr[1]), gelu_erf(rr[2]), gelu_erf(rr[3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Static cross-attention folding (single utterance, sinusoidal positions, description <= 64 tokens).
// The description K/V never change during a call (:872-875), so for one utterance both cross projections can absorb them once
// at prefill:   scores_h = K_h (Wq_h x) = (K_h Wq_h) x = M_h x        out = Wo concat_h(V_h^T p_h) = sum_h (Wo_h V_h^T) p_h = U p
// M [heads*NE][H] has exactly the size of Wq (NE = 64 = head_dim) and U [H][heads*NE] the size of Wo, so the decode step streams
// the same bytes but the cross block becomes TWO GEMV nodes (LN2 + M x; per-head softmax + U p + residual) instead of three
// (LN2 + q projection, attention kernel, out projection): one dependent ~3.6 us node less per layer. log2(e) / sqrt(d) is folded
// into M (base-2 softmax). Exact algebra; in bf16 M and U are rounded once to bf16 (the reference rounds q and the context).
// ------------------------------------------------------------------------------------------------------
template <typename WT, bool W8> struct RmRow {  // one row-major weight row as fp32: engine dtype, or e4m3 bytes * row scale
  static __device__ __forceinline__ void ld8(const void* W, const float* sc, int row, int K, int k, float (&o)[8]) {
    if constexpr (W8) {
      const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(W) + (size_t)row * K + k);
      const float s = sc[row];
      const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.x, true);
      const auto c = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)v.y, true);
      o[0] = a[0] * s; o[1] = a[1] * s; o[2] = b[0] * s; o[3] = b[1] * s; o[4] = c[0] * s; o[5] = c[1] * s; o[6] = d[0] * s; o[7] = d[1] * s;
    } else {
      const WT* p = reinterpret_cast<const WT*>(W) + (size_t)row * K + k;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = Elem<WT>::ld(p + e);
    }
  }
};

// per-layer operands of the fold: one launch covers every layer (48 launches of ~16 us each were 0.8 ms on the first-token path)
struct FoldLayer {
  const void* wq; const float* wq_sc; const void* kcache; void* M;
  const void* wo; const float* wo_sc; const void* vcache; void* U;
};

// M[(h*NE + n)][k] = qscale * sum_d K[h/n_rep][n][d] * Wq[h*64 + d][k]; one thread = 8 consecutive k. grid (H/8/64, NE, layers*heads), 64 threads
template <typename WT, bool W8>
__global__ void __launch_bounds__(64) xfold_m_kernel(const FoldLayer* __restrict__ layers, int nheads, int H, int NE, int cap, int n_rep,
                                                     const DevDims* dims, float qscale) {
  const int k = (blockIdx.x * 64 + threadIdx.x) * 8, n = blockIdx.y, l = blockIdx.z / nheads, h = blockIdx.z - l * nheads;
  if (k >= H) return;
  const FoldLayer fl = layers[l];
  WT* out = reinterpret_cast<WT*>(fl.M) + ((size_t)h * NE + n) * H + k;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (n < dims->N) {
    const WT* kr = reinterpret_cast<const WT*>(fl.kcache) + ((size_t)(h / n_rep) * cap + n) * 64;
    for (int d = 0; d < 64; ++d) {
      const float kd = Elem<WT>::ld(kr + d);
      float w[8];
      RmRow<WT, W8>::ld8(fl.wq, fl.wq_sc, h * 64 + d, H, k, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(kd, w[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) store_from_f32<WT>(out + e, acc[e] * qscale);
}

// U[o][h*NE + n] = sum_d Wo[o][h*64 + d] * V[h/n_rep][n][d]; one thread = one (o, h, n). grid (NE*heads/64, H, layers), 64 threads
template <typename WT, bool W8>
__global__ void __launch_bounds__(64) xfold_u_kernel(const FoldLayer* __restrict__ layers, int H, int NE, int heads, int cap, int n_rep,
                                                     const DevDims* dims) {
  const int col = blockIdx.x * 64 + threadIdx.x, o = blockIdx.y;
  if (col >= heads * NE) return;
  const FoldLayer fl = layers[blockIdx.z];
  const int h = col / NE, n = col - h * NE;
  float acc = 0.f;
  if (n < dims->N) {
    const WT* vr = reinterpret_cast<const WT*>(fl.vcache) + ((size_t)(h / n_rep) * cap + n) * 64;
    for (int d0 = 0; d0 < 64; d0 += 8) {
      float w[8];
      RmRow<WT, W8>::ld8(fl.wo, fl.wo_sc, o, H, h * 64 + d0, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(w[e], Elem<WT>::ld(vr + d0 + e), acc);
    }
  }
  store_from_f32<WT>(reinterpret_cast<WT*>(fl.U) + (size_t)o * heads * NE + col, acc);
}

// The same two products, tiled (round 5): the kernels above re-read their second operand once per output row - xfold_u_kernel the head's V rows
// for each of the H output rows (438 us for the 24 layers of Mini-v1), xfold_m_kernel the head's 64 rows of Wq for each of the NE positions
// (141 us). Here a workgroup owns a 64 x 64 output tile of one (layer, head): both operands ([64][64 head dimensions]) go through LDS once,
// transposed to [d][i] / [d][j], and a thread accumulates a 4 x 4 patch with ONE fmaf chain per output in ascending d - the order (and therefore
// every bit) of the kernels above. IS_U: out = U, i = output row o (tile index blockIdx.x), j = position n; else out = M, i = position n,
// j = input feature k (tile index blockIdx.x). grid (H / 64, heads, layers), 256 threads; H % 64 == 0.
template <typename WT, bool W8, bool IS_U>
__global__ void __launch_bounds__(256) xfold_tile_kernel(const FoldLayer* __restrict__ layers, int nheads, int H, int NE, int cap, int n_rep,
                                                         const DevDims* dims, float qscale) {
  constexpr int LD = 68;  // row pitch in floats: 16-byte aligned rows for the float4 reads
  __shared__ __attribute__((aligned(16))) float At[64 * LD];
  __shared__ __attribute__((aligned(16))) float Bt[64 * LD];
  const int tid = threadIdx.x, t0 = blockIdx.x * 64, h = blockIdx.y;
  const FoldLayer fl = layers[blockIdx.z];
  const int N = dims->N;
  const WT* kv = reinterpret_cast<const WT*>(IS_U ? fl.vcache : fl.kcache) + (size_t)(h / n_rep) * cap * 64;
  float* const kvt = IS_U ? Bt : At;  // the cache rows are the j operand of U and the i operand of M
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = tid & 63, d0 = (r * 4 + (tid >> 6)) * 8;  // consecutive lanes = consecutive rows: conflict-free transposed stores
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < NE && row < N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = Elem<WT>::ld(kv + (size_t)row * 64 + d0 + e);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) kvt[(d0 + e) * LD + row] = v[e];
    float w[8];
    if constexpr (IS_U) {  // Wo row t0 + row, head dimensions d0 .. d0 + 7 of head h
      RmRow<WT, W8>::ld8(fl.wo, fl.wo_sc, t0 + row, H, h * 64 + d0, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) At[(d0 + e) * LD + row] = w[e];
    } else {               // Wq row h * 64 + row (= head dimension d), input features t0 + d0 .. + 7
      RmRow<WT, W8>::ld8(fl.wq, fl.wq_sc, h * 64 + row, H, t0 + d0, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) Bt[row * LD + d0 + e] = w[e];
    }
  }
  __syncthreads();
  const int i0 = (tid >> 4) * 4, j0 = (tid & 15) * 4;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
#pragma unroll 8
  for (int d = 0; d < 64; ++d) {
    const float4 av = *reinterpret_cast<const float4*>(At + d * LD + i0), bv = *reinterpret_cast<const float4*>(Bt + d * LD + j0);
    const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(aa[a], bb[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (IS_U) {
        const int o = t0 + i0 + a, n = j0 + b;
        if (n < NE) store_from_f32<WT>(reinterpret_cast<WT*>(fl.U) + (size_t)o * nheads * NE + h * NE + n, n < N ? acc[a][b] : 0.f);
      } else {
        const int n = i0 + a, k = t0 + j0 + b;
        if (n < NE) store_from_f32<WT>(reinterpret_cast<WT*>(fl.M) + ((size_t)h * NE + n) * H + k, (n < N ? acc[a][b] : 0.f) * qscale);
      }
    }
}

// prefill: write all Q new K/V rows (RoPE on k) into the self cache.  grid (Q, heads, B), 64 threads
template <typename WT, bool KV8 = false>
__global__ void kv_append_kernel(const float* __restrict__ knew, const float* __restrict__ vnew, int kv_ld, void* kcache,
                                 void* vcache, int cap, int Q, int nheads, const float* cos, const float* sin, float* kscale = nullptr,
                                 float* vscale = nullptr) {
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
  const int row = b * Q + qi, pos = qi;
  const float* kr = knew + (size_t)row * kv_ld + h * 64;
  float kx = kr[d];
  if (cos) {
    const float other = d < 32 ? -kr[d + 32] : kr[d - 32];
    kx = kx * cos[(size_t)pos * 64 + d] + other * sin[(size_t)pos * 64 + d];
  }
  const float vx = vnew[(size_t)row * kv_ld + h * 64 + d];
  const size_t off = (((size_t)b * nheads + h) * cap + pos) * 64 + d;
  if constexpr (KV8) {  // e4m3 cache rows: one wave = one row, scale from the row's maximum (attn_kernel's quantiser)
    const float ks = kv8_row_scale(wave_max(fabsf(kx))), vs = kv8_row_scale(wave_max(fabsf(vx)));
    reinterpret_cast<uint8_t*>(kcache)[off] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(kx * (1.0f / ks), 0.f, 0, false) & 0xff);
    reinterpret_cast<uint8_t*>(vcache)[off] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(vx * (1.0f / vs), 0.f, 0, false) & 0xff);
    if (d == 0) {
      kscale[((size_t)b * nheads + h) * cap + pos] = ks;
      vscale[((size_t)b * nheads + h) * cap + pos] = vs;
    }
  } else {
    store_from_f32<WT>(reinterpret_cast<WT*>(kcache) + off, kx);
    store_from_f32<WT>(reinterpret_cast<WT*>(vcache) + off, vx);
  }
}

// ------------------------------------------------------------------------------------------------------
// Embedding: h = sum_k E_k[token_k] (+ sinusoidal position)   modeling:1433, :1506-1511; delay mask :205-276
// ------------------------------------------------------------------------------------------------------
// value the model sees at (codebook k, column j): BOS / PAD of the delay pattern, the shifted voice-prompt code where the
// pattern holds one (input_ids_shifted[:, k, k : seq_len + k] = prompt, :250-252; BOS / PAD win, :263), else the raw id
__device__ __forceinline__ long long delayed_token(const long long* ids, int ld, int rowk, int k, int j, int K, const DevDims& dd, int bos,
                                                   int pad) {
  if (dd.max_length >= 2 * K - 1) {  // :246-247: short max_length disables the pattern
    if (j <= k) return bos;                             // tril :260
    if (j - k >= dd.max_length - K + 1) return pad;     // triu(diagonal = max_length - K + 1) :256-258
    if (j - k - 1 < dd.T_prefix) return dd.prefix[(size_t)rowk * dd.prefix_ld + (j - k - 1)];
  }
  return ids[(size_t)rowk * ld + j];
}

struct EmbedArgs {
  const void* tables;      // [K][V+1][H] engine dtype
  const float* pos_table;  // [max_pos][H] or null (RoPE)
  const float* prompt;     // [B][P][H] or null
  const long long* ids;
  int ids_ld;
  const int* cur_len;
  const DevDims* dims;
  float* h;
  int H, K, V1, bos, pad;
  int prefill;             // 1: grid (P+1, B) rows; 0: grid (1, B), column cur_len-1 at position P + cur_len - 1
};

template <typename WT>
__global__ void embed_kernel(EmbedArgs a) {
  const int b = blockIdx.y;
  const int P = a.dims->P;
  const int qi = blockIdx.x;
  const int Q = a.prefill ? (int)gridDim.x : 1;        // prefill: P prompt positions + the given decoder columns (BOS [+ voice-prompt prefix])
  const int j = a.prefill ? max(qi - P, 0) : a.cur_len[b] - 1;  // token column
  const int pos = a.prefill ? qi : P + j;              // absolute position (padded prompt ids still count, :1470)
  float* out = a.h + ((size_t)b * Q + qi) * a.H;
  const WT* tab = reinterpret_cast<const WT*>(a.tables);
  if (a.prefill && qi < P) {
    const float* pr = a.prompt + ((size_t)b * P + qi) * a.H;
    for (int d = threadIdx.x; d < a.H; d += blockDim.x) out[d] = pr[d] + (a.pos_table ? a.pos_table[(size_t)pos * a.H + d] : 0.f);
    return;
  }
  __shared__ int s_tok[32];
  if (threadIdx.x < a.K)
    s_tok[threadIdx.x] = (int)delayed_token(a.ids, a.ids_ld, b * a.K + threadIdx.x, threadIdx.x, j, a.K, *a.dims, a.bos, a.pad);
  __syncthreads();
  for (int d = threadIdx.x; d < a.H; d += blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < a.K; ++k) acc += Elem<WT>::ld(tab + ((size_t)k * a.V1 + s_tok[k]) * a.H + d);  // sum([...]) order :1433
    if (a.pos_table) acc += a.pos_table[(size_t)pos * a.H + d];
    out[d] = acc;
  }
}

// ------------------------------------------------------------------------------------------------------
// Sampler tail: one workgroup per utterance (K rows). Restates one _sample iteration on device:
//   fp32 logits -> MinNewTokens -> ParlerTTSLogitsProcessor (logits_processors.py:44-53) -> [temperature, top-k,
//   top-p] -> argmax | multinomial -> pad finished rows -> append -> EOS / max_length stopping.
// ------------------------------------------------------------------------------------------------------
struct TailArgs {
  const float* logits;  // [B][K][V]
  long long* ids;
  int ids_ld;
  int* cur_len;        // [B]
  int* unfinished;     // [B*K]: 1 = still generating; -(t + 1) = finished at step (column) t
  int* has_eos;        // [B*K]
  int* first_unf;      // [B] local codebook index
  const DevGen* gen;
  float* sort_buf;     // unused (sampling sorts in LDS)
  int B, K, V, eos, pad;
  // fused embedding of the NEXT step (decode graph): h[b] = sum_k E_k[delayed token of column t] + pos[P + t]
  const void* tables;      // [K][V+1][H] engine dtype, or null: no fused embedding (manual path / prefill-only probes)
  const float* pos_table;  // or null (RoPE)
  const DevDims* dims;
  float* h;                // [B][H]
  int H, bos, bf16_tables;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

#define PTTS_SORT_N 2048

// ---- sort-free sampling, one wave per codebook row ------------------------------------------------------------------
// monotone key of a float (larger value <=> larger unsigned key); -inf is the smallest key of any valid entry
__device__ __forceinline__ unsigned f32_key(float x) {
  const unsigned b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

// TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial for ONE row held by one wave
// (lane l owns entries l, l + 64, ...; NV per lane). No sort: both warpers keep a PREFIX of the descending order, i.e. a
// threshold on the value, found by a 32-step radix search whose predicate is one wave reduction:
//   top-k: largest key T with count(key >= T) >= k      (keeps ties at the k-th value, as `scores < topk[..., -1]` does)
//   top-p: smallest key L with sum_{key > L} p < top_p  (sorted ascending, HF removes cumsum <= 1 - top_p: a token stays iff
//          the mass strictly above it is < top_p; the arg-max always stays)
// The draw walks the kept entries in lane-major order (any fixed order gives the same distribution; torch.multinomial's
// own stream cannot be matched, parity is statistical: tests/test_lm_gpu.py chi-square + support checks).
template <int NV>
__device__ __forceinline__ int wave_sample_row(const float (&lg)[NV], int V, int lane, const DevGen& g, bool eos_blocked, int eos, float u01) {
  const int nv = (V + 63) >> 6;
  float x[NV];
  unsigned key[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = lane + 64 * i;
    float xv = -INFINITY;
    if (i < nv && v < V) {
      xv = lg[i];
      xv = (eos_blocked && v == eos) ? -INFINITY : xv / g.temperature;  // processors first (-inf stays -inf), then scores / T
    }
    x[i] = xv;
    key[i] = (i < nv && v < V) ? f32_key(xv) : 0u;  // key 0 < key(-inf): padding never counts
  }
  if (g.top_k > 0 && g.top_k < V) {
    unsigned T = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = T | (1u << bit);
      float c = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) c += key[i] >= cand ? 1.f : 0.f;
      if (wave_sum(c) >= (float)g.top_k) T = cand;  // counts <= 2048 are exact in fp32
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (i < nv && key[i] < T) x[i] = -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) mx = fmaxf(mx, x[i]);
  mx = wave_max(mx);
  float (&e)[NV] = x;  // from here on x holds the softmax numerators (one array less live: the kernel must not spill)
  float ls = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    e[i] = (i < nv && x[i] != -INFINITY) ? expf(x[i] - mx) : 0.f;
    ls += e[i];
  }
  float tot = wave_sum(ls);
  if (g.top_p < 1.0f) {
    const float thr = g.top_p * tot;
    unsigned L = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned test = L | ((1u << bit) - 1u);
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (i < nv) m += key[i] > test ? e[i] : 0.f;
      if (!(wave_sum(m) < thr)) L |= 1u << bit;
    }
    ls = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < nv && key[i] < L) e[i] = 0.f;
      ls += e[i];
    }
    tot = wave_sum(ls);
  }
  // inverse CDF over the kept entries in lane-major order
  const float target = u01 * tot;
  float incl = ls;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const unsigned long long has = __ballot(ls > 0.f);
  const unsigned long long hit = __ballot(ls > 0.f && incl >= target);
  const int sel = hit ? (int)__builtin_ctzll(hit) : (has ? 63 - (int)__builtin_clzll(has) : 0);
  float run = incl - ls;
  int pick = -1, last = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i < nv && e[i] > 0.f) {
      run += e[i];
      last = lane + 64 * i;
      if (pick < 0 && run >= target) pick = last;
    }
  }
  if (pick < 0) pick = last;
  return __shfl(pick, sel);
}

// Embedding of the token column just produced (column t, fed at position P + t) for the next decode step:
// the delay pattern is applied exactly as embed_kernel / apply_delay_pattern_mask do (modeling:205-276, :1433).
// One thread owns 4 consecutive features; the K table rows (+ the position row) are ONE batch of independent loads
// (the rolled k loop of round 1 was K sequential cold round trips: most of the 14 us the tail took).
template <int KB, bool BF16>  // codebooks per batch of loads; table dtype
__device__ __forceinline__ void tail_embed_rows(const TailArgs& a, const DevDims& dd, int b, int t, const int* s_tok, int d, float (&acc)[4]) {
  const bf16_t* tb16 = reinterpret_cast<const bf16_t*>(a.tables);
  const float* tb32 = reinterpret_cast<const float*>(a.tables);
#pragma unroll 1
  for (int k0 = 0; k0 < a.K; k0 += KB) {
    uint2 r16[BF16 ? KB : 1];
    float4 r32[BF16 ? 1 : KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int k = min(k0 + u, a.K - 1);
      int tok = s_tok[k];
      if (dd.max_length >= 2 * a.K - 1) {
        if (t <= k) tok = a.bos;
        else if (t - k >= dd.max_length - a.K + 1) tok = a.pad;
        else if (t - k - 1 < dd.T_prefix) tok = (int)dd.prefix[(size_t)(b * a.K + k) * dd.prefix_ld + (t - k - 1)];  // voice prompt
      }
      const size_t off = ((size_t)k * (a.V + 1) + tok) * a.H + d;
      if constexpr (BF16) r16[u] = *reinterpret_cast<const uint2*>(tb16 + off);
      else r32[u] = *reinterpret_cast<const float4*>(tb32 + off);
    }
#pragma unroll
    for (int u = 0; u < KB; ++u) {  // sum([...]) order :1433
      if (k0 + u < a.K) {
        if constexpr (BF16) {
          acc[0] += __uint_as_float(r16[u].x << 16); acc[1] += __uint_as_float(r16[u].x & 0xffff0000u);
          acc[2] += __uint_as_float(r16[u].y << 16); acc[3] += __uint_as_float(r16[u].y & 0xffff0000u);
        } else {
          acc[0] += r32[u].x; acc[1] += r32[u].y; acc[2] += r32[u].z; acc[3] += r32[u].w;
        }
      }
    }
  }
}
__device__ __forceinline__ void tail_embed_next(const TailArgs& a, const DevDims& dd, int b, int t, const int* s_tok, int tid) {
  if (!a.tables) return;
  float* out = a.h + (size_t)b * a.H;
  for (int d = tid * 4; d < a.H; d += blockDim.x * 4) {  // H % 4 == 0 (hidden_size % 32 == 0 is checked at create)
    float4 pos = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.pos_table) pos = *reinterpret_cast<const float4*>(a.pos_table + (size_t)(dd.P + t) * a.H + d);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bf16_tables) tail_embed_rows<9, true>(a, dd, b, t, s_tok, d, acc);  // K = 9 (Mini / Large v1): one batch; more codebooks loop
    else tail_embed_rows<9, false>(a, dd, b, t, s_tok, d, acc);
    *reinterpret_cast<float4*>(out + d) = make_float4(acc[0] + pos.x, acc[1] + pos.y, acc[2] + pos.z, acc[3] + pos.w);
  }
}

// NV = logits per lane: 8 (vocab <= 512), 18 (<= 1152: Mini / Large v1, vocab 1088) or 32 (<= 2048)
template <int NV>
__global__ void __launch_bounds__(1024) tail_kernel(TailArgs a) {
  __shared__ int s_tok[32];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // ---- t = 0: every load that does not depend on another load goes in flight together (the step's critical path ends
  // here: lengths / flags / parameters / this wave's logits row are ONE round trip, the embedding rows a second one)
  const DevGen g = *a.gen;
  const int t = a.cur_len[b];  // column the new token is written to; t-1 new tokens generated so far
  // "has the reference loop already exited?" = every row finished BEFORE this step. Rows that finish during this very launch
  // (other workgroups, stamp -(t + 1)) still count as active here, so the answer does not depend on workgroup timing.
  int any_local = 0;
  for (int i = tid; i < a.B * a.K; i += blockDim.x) {
    const int v = a.unfinished[i];
    any_local |= (v > 0) | (v <= -(t + 1));
  }
  const int fu0 = a.first_unf[b];
  const DevDims dd = *a.dims;  // the embedding of the next column needs P / max_length / the voice-prompt prefix: fetched now, not after the argmax
  const int t_prefix = dd.T_prefix;
  const int he = lane < a.K ? a.has_eos[b * a.K + lane] : 0;  // every wave: the K EOS flags of this utterance
  float lg[NV];
  const int k0 = w;  // greedy: wave w starts with codebook row w
  if (k0 < a.K) {
    const float* sc0 = a.logits + (size_t)(b * a.K + k0) * a.V;
#pragma unroll
    for (int i = 0; i < NV; ++i) lg[i] = (lane + 64 * i < a.V) ? sc0[lane + 64 * i] : -INFINITY;
  }
  const int unf0 = k0 < a.K ? (a.unfinished[b * a.K + k0] > 0) : 0;
  if (!__syncthreads_or(any_local)) return;  // every row of every utterance finished: the reference loop has exited (no-op step)

  int fu = fu0;
  if (__shfl(he, fu0) > 0 && fu0 < a.K - 1) fu += 1;  // logits_processors.py:48 (advance <= 1 per step)
  if (tid == 0) a.first_unf[b] = fu;
  const bool block_eos_all = (t - 1 - t_prefix) < g.min_new_tokens;  // new tokens = columns after the (1 + T_prefix) given ones

  // one wave per codebook row, no workgroup barrier until every row has its token. greedy: torch.argmax semantics (first
  // index on ties); do_sample: the warpers + multinomial on the same wave (wave_sample_row)
  for (int k = w; k < a.K; k += (int)(blockDim.x >> 6)) {
    const int row = b * a.K + k;
    if (k != k0) {
      const float* sc = a.logits + (size_t)row * a.V;
#pragma unroll
      for (int i = 0; i < NV; ++i) lg[i] = (lane + 64 * i < a.V) ? sc[lane + 64 * i] : -INFINITY;
    }
    const bool eos_blocked = block_eos_all || (g.use_eos_gate && k > fu);
    int widx;
    if (!g.do_sample) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        float x = lg[i];
        if (eos_blocked && v == a.eos) x = -INFINITY;
        if (v < a.V && (x > best || (x == best && v < bi))) { best = x; bi = v; }
      }
      const float wbest = wave_max(best);
      const float cand = (best == wbest) ? (float)bi : 3.0e9f;  // vocabulary indices are exact in fp32
      widx = (int)(-wave_max(-cand));
    } else {
      const unsigned long long hsh = splitmix64(g.seed ^ splitmix64(((unsigned long long)t << 32) ^ (unsigned long long)row));
      const float u = (float)((hsh >> 40) + 0.5) * (1.0f / 16777216.0f);  // (0,1)
      widx = wave_sample_row<NV>(lg, a.V, lane, g, eos_blocked, a.eos, u);
    }
    if (lane == 0) {
      const int unf = k == k0 ? unf0 : (a.unfinished[row] > 0);
      const int nxt = unf ? widx : a.pad;  // next_tokens * unfinished + pad * (1 - unfinished)
      a.ids[(size_t)row * a.ids_ld + t] = nxt;
      s_tok[k] = nxt;
      if (nxt == a.eos) a.has_eos[row] = 1;
      if (unf && ((nxt == a.eos) || (t + 1 >= g.max_length))) a.unfinished[row] = -(t + 1);  // EosTokenCriteria | MaxLengthCriteria
    }
  }
  __syncthreads();
  if (tid == 0) a.cur_len[b] = t + 1;
  tail_embed_next(a, dd, b, t, s_tok, tid);
}

// manual path: append caller-chosen tokens (user LogitsProcessorList / StoppingCriteria ran on the host side)
static __global__ void push_tokens_kernel(const long long* tokens, const int* finished, long long* ids, int ids_ld, int* cur_len,
                                   int* unfinished, int* has_eos, int B, int K, int eos) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B * K) return;
  const int b = row / K;
  const int t = cur_len[b];
  const long long tk = tokens[row];
  ids[(size_t)row * ids_ld + t] = tk;
  if (tk == eos) has_eos[row] = 1;
  if (finished && finished[row] && unfinished[row] > 0) unfinished[row] = -(t + 1);
}
static __global__ void bump_len_kernel(int* cur_len, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) cur_len[b] += 1;
}

// voice prompt, teacher forcing: column j (1 <= j <= T_prefix) of the raw ids = the delay pattern's value there
static __global__ void push_prefix_col_kernel(long long* ids, int ids_ld, const DevDims* dims, int j, int B, int K, int bos) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B * K) return;
  const int k = row % K;
  const DevDims dd = *dims;
  ids[(size_t)row * ids_ld + j] = j <= k ? (long long)bos : dd.prefix[(size_t)row * dd.prefix_ld + (j - k - 1)];
}
// all T voice-prompt columns of the raw ids at once (batched multi-column prefill)
static __global__ void push_prefix_all_kernel(long long* ids, int ids_ld, const DevDims* dims, int T, int B, int K, int bos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K * T) return;
  const int row = i / T, j = 1 + i % T, k = row % K;
  const DevDims dd = *dims;
  ids[(size_t)row * ids_ld + j] = j <= k ? (long long)bos : dd.prefix[(size_t)row * dd.prefix_ld + (j - k - 1)];
}
static __global__ void set_len_kernel(int* cur_len, int B, int v) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) cur_len[b] = v;
}
static __global__ void reset_state_kernel(long long* ids, int ids_ld, int* cur_len, int* unfinished, int* has_eos, int* first_unf,
                                   int B, int K, int bos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * K) {
    ids[(size_t)i * ids_ld] = bos;
    unfinished[i] = 1;
    has_eos[i] = 0;
  }
  if (i < B) {
    cur_len[i] = 1;
    first_unf[i] = 0;
  }
}

static __global__ void set_params_kernel(DevDims* dd, DevGen* dg, DevDims d, DevGen g) {
  *dd = d;
  *dg = g;
}

static __global__ void fill_int_kernel(int* p, int v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
