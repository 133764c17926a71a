// T5 description encoder behind the C ABI of include/ptts.h (ptts_t5_*): the text_encoder forward that generate() runs once per call on the
// time-to-first-token path (modeling_parler_tts.py:3048-3097 -> transformers T5EncoderModel / T5Stack, the reference's third-party dependency).
// Stock PyTorch-ROCm spends ~50 launches per T5 block there (~5 of 7.3 ms to the first token, profiles/r03_prefill_kernels.txt); here a block
// is 7 kernel nodes on the decoder engine's MFMA strip / block GEMMs, and the whole encoder is one captured hipGraph per (batch, length):
//   RMSNorm -> [q | k | v] projection -> bidirectional attention with the relative-position bias -> o projection + residual ->
//   RMSNorm -> [wi_0 / wi_1 interleaved] projection with the gelu_new gate in the epilogue -> wo projection + residual.
// Numerics: weights / GEMM operands in the engine dtype (bf16 or fp32), fp32 accumulation, fp32 residual stream, fp32 attention
// (transformers' bf16 run rounds the stream and the probabilities to bf16 after every op; the fp32 engine is the parity mode).
#include <map>
#include <set>
#include <string>
#include <vector>
#include <math.h>
#include <string.h>

#include "ptts_common.h"
#include "ptts_lm_kernels.h"
#include "ptts_gemm_launch.h"

namespace {

struct T5Layer {
  void *qkv = nullptr, *o = nullptr, *wi = nullptr, *wo = nullptr;  // packed strips; wi = rows of wi_0 / wi_1 interleaved (EPI_GATE_WT)
  float *ln1 = nullptr, *ln2 = nullptr;
};

// ---- kernels ---------------------------------------------------------------------------------------------------------------------
// inputs_embeds = shared(input_ids) (T5Stack.forward): one workgroup per token row, fp32 residual stream
template <typename WT>
__global__ void __launch_bounds__(256) t5_embed_kernel(const WT* __restrict__ table, const long long* __restrict__ ids, float* __restrict__ h, int D, int vocab) {
  const int m = blockIdx.x;
  long long id = ids[m];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // (the stock embedding asserts on the device; never read outside the table)
  const WT* row = table + (size_t)id * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    float4 v;
    v.x = Elem<WT>::ld(row + d); v.y = Elem<WT>::ld(row + d + 1); v.z = Elem<WT>::ld(row + d + 2); v.w = Elem<WT>::ld(row + d + 3);
    *reinterpret_cast<float4*>(h + (size_t)m * D + d) = v;
  }
}

// bias[h][delta + L - 1] = relative_attention_bias[bucket(delta)][h], delta = key position - query position (T5Attention.compute_bias)
static __global__ void t5_bias_table_kernel(const float* __restrict__ rel, const int* __restrict__ bucket_of, float* __restrict__ bias, int nheads, int ndelta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nheads * ndelta) return;
  const int h = i / ndelta, d = i - h * ndelta;
  bias[i] = rel[(size_t)bucket_of[d] * nheads + h];
}

// rows of wi_0 (which = 0) / wi_1 (which = 1) into the interleaved strip matrix: packed row 2 i + which = source row i
template <typename WT, typename ST>
__global__ void pack_weight_ilv_kernel(const ST* __restrict__ src, WT* __restrict__ dst, int n_src, int K, int which) {
  constexpr int KT = Elem<WT>::KT, EPL = Elem<WT>::EPL;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nfrag = K / KT;
  const size_t total = (size_t)(2 * n_src / 16) * nfrag * 64;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int t = (int)((idx >> 6) % nfrag);
  const int s = (int)((idx >> 6) / nfrag);
  const int R = s * 16 + (lane & 15);
  if ((R & 1) != which) return;
  const int row = R >> 1;
  const int k = t * KT + (lane >> 4) * EPL;
  WT* d = dst + (((size_t)s * nfrag + t) * 64 + lane) * EPL;
#pragma unroll
  for (int e = 0; e < EPL; ++e) store_from_f32<WT>(d + e, load_as_f32<ST>(src + (size_t)row * K + k + e));
}

struct T5AttnArgs {
  const float* qkv;   // [B*N][ld] fp32: q at column h*64, k at inner + h*64, v at 2*inner + h*64
  int ld, inner;
  const float* bias;  // [heads][bias_ld], entry (key - query) + bias_zero
  int bias_ld, bias_zero;
  const int* mask;    // [B][N] int32 (1 = keep) or null
  void* out;          // [B*N][inner] engine dtype, row-major or MFMA B-fragment order
  int N;
  int out_fo;
};
// The attention kernels take the struct's fields as SCALAR parameters (14 dwords = exactly what the command processor preloads into SGPRs before the first wave
// starts, -mllvm -amdgpu-kernarg-preload-count=14, ptts_common.h): no s_load in front of the wave's first loads (call 54).
static_assert(sizeof(T5AttnArgs) == 56, "T5AttnArgs: the 14 preloaded dwords");
#define T5AttnArgs_KPARAMS const float *kqkv_, int kld_, int kinner_, const float *kbias_, int kbld_, int kbz_, const int *kmask_, void *kout_, int kN_, int kfo_
#define T5AttnArgs_KJOIN(a) \
  T5AttnArgs a;             \
  a.qkv = kqkv_; a.ld = kld_; a.inner = kinner_; a.bias = kbias_; a.bias_ld = kbld_; a.bias_zero = kbz_; a.mask = kmask_; a.out = kout_; a.N = kN_; a.out_fo = kfo_;
template <typename Kn> inline void t5_attn_launch(Kn kern, dim3 grid, dim3 block, hipStream_t st, const T5AttnArgs& a) {
  hipLaunchKernelGGL(kern, grid, block, 0, st, a.qkv, a.ld, a.inner, a.bias, a.bias_ld, a.bias_zero, a.mask, a.out, a.N, a.out_fo);
}

// T5Attention.forward, encoder self-attention: scores = q k^T (NO 1/sqrt(d) scale) + position_bias (+ (1 - mask) * finfo.min), softmax in fp32,
// context = p v. One workgroup = 8 queries of one (utterance, head): 4 waves x 2 queries; keys in tiles of 64 (lane = key), K / V tiles staged in
// LDS once per workgroup, online softmax across tiles. A masked key keeps the score -FLT_MAX exactly as the additive mask leaves it (a fully
// masked row is therefore uniform over all N keys, like the reference); keys beyond N do not exist.
template <typename WT>
__global__ void __launch_bounds__(256) t5_attn_kernel(T5AttnArgs_KPARAMS) {
  T5AttnArgs_KJOIN(a)
  constexpr int QW = 2, QB = 4 * QW, EPL = Elem<WT>::EPL;
  __shared__ float sK[64 * 65];
  __shared__ __attribute__((aligned(16))) float sV[64 * 64];
  __shared__ float sQ[QB][64];
  __shared__ float sP[QB][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * QB;
  const float* base = a.qkv + (size_t)b * a.N * a.ld;
  for (int e = tid; e < QB * 64; e += 256) {
    const int qi = e >> 6, d = e & 63, i = min(i0 + qi, a.N - 1);
    sQ[qi][d] = base[(size_t)i * a.ld + h * 64 + d];
  }
  float m_run[QW], l_run[QW], o[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) { m_run[q] = -INFINITY; l_run[q] = 0.f; o[q] = 0.f; }
  for (int j0 = 0; j0 < a.N; j0 += 64) {
    __syncthreads();  // the previous tile is consumed (first pass: sQ is visible)
    for (int e = tid; e < 64 * 16; e += 256) {
      const int r = e >> 4, c4 = e & 15, j = j0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j < a.N) {
        kv = *reinterpret_cast<const float4*>(base + (size_t)j * a.ld + a.inner + h * 64 + c4 * 4);
        vv = *reinterpret_cast<const float4*>(base + (size_t)j * a.ld + 2 * a.inner + h * 64 + c4 * 4);
      }
      float* kd = sK + r * 65 + c4 * 4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      *reinterpret_cast<float4*>(sV + r * 64 + c4 * 4) = vv;
    }
    __syncthreads();
    const int j = j0 + lane;
    const bool exists = j < a.N;
    const bool kept = exists && (!a.mask || a.mask[(size_t)b * a.N + j] != 0);
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
      const int i = min(i0 + w * QW + q, a.N - 1);
      const float bias = a.bias[(size_t)h * a.bias_ld + (min(j, a.N - 1) - i) + a.bias_zero];
      const float sc = !exists ? -INFINITY : (kept ? s[q] + bias : -3.402823466e38f);
      const float m_new = fmaxf(m_run[q], wave_max(sc));
      const float alpha = m_run[q] == -INFINITY ? 0.f : expf(m_run[q] - m_new);
      const float p = exists ? expf(sc - m_new) : 0.f;
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
    if (i >= a.N) continue;
    const int m = b * a.N + i, kcol = h * 64 + lane;
    WT* dst = reinterpret_cast<WT*>(a.out);
    if (a.out_fo) dst += fo_vec_index<WT>(m, kcol & ~(EPL - 1), a.inner / Elem<WT>::KT) * EPL + (kcol & (EPL - 1));
    else dst += (size_t)m * a.inner + kcol;
    store_from_f32<WT>(dst, o[q] / l_run[q]);
  }
}

// t5_attn_mfma_kernel (round 6): the same attention on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, an fmaf chain per output - the
// arithmetic of the kernel above in another summation order), for batches that fill the chip: t5_attn_kernel spends 34 us per block at 32 x 64 tokens
// (16 TFLOP/s of VALU fmaf behind one LDS read per fmaf pair; profiles/r06_prefill_kernels_bs32_v1.txt). One workgroup = 64 queries of one
// (utterance, head), one wave = 16 queries x ALL keys, key blocks of 64 with the K / V tiles in LDS (fp32, 16-byte slots XOR-swizzled by row & 15:
// the b128 fragment reads of K and the b32 reads of V are both conflict-free without padding):
//   S^T = K Q^T   A = K[key][d], B = Q[query][d] (Q fragments live in registers): lane (i = l & 15, g = l >> 4) ends up holding the scores of
//                 query i against keys 16 kt + 4 g + r - 16 of the block's 64 keys, the other 48 in the three lanes with the same i
//   softmax       per query across those 4 lanes (permlane swaps), online across key blocks
//   O = P V       A = P: step (kt, r) takes the lane's OWN register P[i][16 kt + 4 g + r] (the MFMA sums over g: no transpose, no LDS round trip
//                 for the probabilities), B = V[16 kt + 4 g + r][4 j + dt] (one b128 read per key: ptts_common.h, attn_block_*)
// k order of the q.k sums: d = 16 c + e + 4 g over (c, e) then g (fixed, deterministic); of the p.v sums: keys 16 kt + r + 4 g over (kt, r) then g.
template <typename WT>
__global__ void __launch_bounds__(256) t5_attn_mfma_kernel(T5AttnArgs_KPARAMS) {
  T5AttnArgs_KJOIN(a)
  __shared__ __attribute__((aligned(16))) float sK[64 * 64];
  __shared__ __attribute__((aligned(16))) float sV[64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * 64 + w * 16;
  const float* base = a.qkv + (size_t)b * a.N * a.ld;
  const int iq = min(i0 + j, a.N - 1);  // clamped queries are computed and dropped
  float4 qr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qr[c] = *reinterpret_cast<const float4*>(base + (size_t)iq * a.ld + h * 64 + 16 * c + 4 * g);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float4* sK4 = reinterpret_cast<const float4*>(sK);
  for (int j0 = 0; j0 < a.N; j0 += 64) {
    if (j0) __syncthreads();  // the previous tiles are consumed
    // (call 38) every global load is unconditional on a clamped address and selected afterwards: inside per-lane conditions each one had been compiled into
    // its own branch + s_waitcnt vmcnt(0) - 8 + 32 dependent round trips per key block (tools/isa_load_chains.py)
    float4 kq[4], vq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u, r = e >> 4, sl = e & 15, key = min(j0 + r, a.N - 1);
      kq[u] = *reinterpret_cast<const float4*>(base + (size_t)key * a.ld + a.inner + h * 64 + sl * 4);
      vq[u] = *reinterpret_cast<const float4*>(base + (size_t)key * a.ld + 2 * a.inner + h * 64 + sl * 4);
    }
    // bias and key flags of this lane's 16 (query, key) pairs: independent of the tiles, in flight across the barrier
    float bias[4][4];
    int flag[4][4];  // 0: no such key, 1: masked, 2: kept
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kc = min(j0 + 16 * kt + 4 * g + r, a.N - 1);
        bias[kt][r] = a.bias[(size_t)h * a.bias_ld + (kc - iq) + a.bias_zero];
        flag[kt][r] = 2;
      }
    if (a.mask) {  // wave-uniform
      int mv[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[kt][r] = a.mask[(size_t)b * a.N + min(j0 + 16 * kt + 4 * g + r, a.N - 1)];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) flag[kt][r] = mv[kt][r] != 0 ? 2 : 1;
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + 16 * kt + 4 * g + r >= a.N) flag[kt][r] = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u, r = e >> 4, sl = e & 15;
      const bool live = j0 + r < a.N;
      reinterpret_cast<float4*>(sK)[r * 16 + (sl ^ (r & 15))] = live ? kq[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(sV)[r * 16 + (sl ^ (r & 15))] = live ? vq[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    f32x4 st[4];
    attn_block_scores(sK4, j, g, qr, st);
    float4 vb[4][4];
    attn_block_v_request(reinterpret_cast<const float4*>(sV), j, g, vb);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sc = flag[kt][r] == 0 ? -INFINITY : (flag[kt][r] == 2 ? st[kt][r] + bias[kt][r] : -3.402823466e38f);
        st[kt][r] = sc;
        mx = fmaxf(mx, sc);
      }
    const float m_new = fmaxf(m_run, across_groups_reduce<OpMax, 16>(mx));
    const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = flag[kt][r] == 0 ? 0.f : expf(st[kt][r] - m_new);
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
    attn_block_pv(st, vb, o);
  }
  WT* dst0 = reinterpret_cast<WT*>(a.out);
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // o[dt][r]: query 4 g + r, column 4 j + dt of the head - four consecutive elements per lane
    const float lr = __shfl(l_run, 4 * g + r);
    const int i = i0 + 4 * g + r;
    if (i >= a.N) continue;
    act_store4<WT>(dst0, b * a.N + i, h * 64 + 4 * j, a.inner, a.out_fo, o[0][r] / lr, o[1][r] / lr, o[2][r] / lr, o[3][r] / lr);
  }
}

template <typename DT>
int t5_convert_into(DT* dst, const void* src, int src_dtype, size_t n, hipStream_t st) {
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  if (src_dtype == PTTS_F32) hipLaunchKernelGGL((convert_kernel<DT, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, n);
  else hipLaunchKernelGGL((convert_kernel<DT, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, n);
  return PTTS_OK;
}

}  // namespace

struct ptts_t5 {
  ptts_t5_config cfg;
  size_t esize = 4;
  int inner = 0;           // num_heads * d_kv
  int rows = 0;            // max_batch * max_len
  hipStream_t own_stream = nullptr;
  std::vector<void*> allocs;
  std::vector<T5Layer> L;
  void* embed = nullptr;   // [vocab][d_model] engine dtype
  float* final_ln = nullptr;
  float* rel = nullptr;    // relative_attention_bias [buckets][heads] fp32
  float* bias = nullptr;   // [heads][2 * max_len - 1]
  int* bucket_of = nullptr;
  bool bias_built = false;
  // scratch + static inputs of the captured graphs
  float *h = nullptr, *qkv = nullptr;
  float* ss = nullptr;     // [rows][d_model / 16] per-strip sums of squares of the residual rows (RMSNorm folded into the GEMMs, <= 256 rows)
  bool use_fold = true;
  void *xw = nullptr, *ctx = nullptr, *ff = nullptr;
  long long* ids = nullptr;
  int* mask = nullptr;
  bool use_fo = true, use_graph = true;
  std::set<std::string> loaded, required;
  std::map<long long, hipGraphExec_t> graphs;  // key: batch, length, masked?
  int last_graph_nodes = 0;                    // kernel nodes of the graph captured last (ptts_t5_debug_graph_nodes)

  int alloc_bytes(void** p, size_t bytes) {
    void* v = nullptr;
    hipError_t e = hipMalloc(&v, bytes > 0 ? bytes : 16);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    allocs.push_back(v);
    *p = v;
    return PTTS_OK;
  }
  template <typename T> int alloc(T** p, size_t n) {
    void* v = nullptr;
    PTTS_TRY(alloc_bytes(&v, n * sizeof(T)));
    *p = reinterpret_cast<T*>(v);
    return PTTS_OK;
  }
};

// T5Attention._relative_position_bucket, bidirectional (modeling_t5.py): the host-side restatement the bias table is built from. Pure host code
// (no device needed): tests pin it against the installed transformers function for every relative position an engine can see.
extern "C" int32_t ptts_t5_relative_bucket(int32_t relative_position, int32_t num_buckets, int32_t max_distance) {
  int nb = num_buckets / 2;                   // bidirectional: half of the buckets per sign
  int bucket = relative_position > 0 ? nb : 0;
  const int rp = relative_position < 0 ? -relative_position : relative_position;
  const int max_exact = nb / 2;
  if (rp < max_exact) return bucket + rp;
  // max_exact + log(rp / max_exact) / log(max_distance / max_exact) * (nb - max_exact), truncated, capped at nb - 1. Exact ratios (rp = max_exact *
  // 2^k with max_distance / max_exact a power of two) land on integers in double precision; the reference evaluates the same expression in fp32.
  const double v = log((double)rp / (double)max_exact) / log((double)max_distance / (double)max_exact) * (double)(nb - max_exact);
  int large = max_exact + (int)v;
  if (large > nb - 1) large = nb - 1;
  return bucket + large;
}

extern "C" int ptts_t5_create(const ptts_t5_config* cfg, ptts_t5** out) {
  PTTS_CHECK(cfg && out, PTTS_E_INVALID, "null argument");
  const ptts_t5_config& c = *cfg;
  PTTS_CHECK(c.dtype == PTTS_F32 || c.dtype == PTTS_BF16, PTTS_E_INVALID, "dtype must be PTTS_F32 or PTTS_BF16");
  PTTS_CHECK(c.d_kv == 64, PTTS_E_UNSUPPORTED, "d_kv must be 64 (t5-small/base/large, flan-t5-*), got %d", c.d_kv);
  PTTS_CHECK(c.num_heads >= 1 && c.num_layers >= 1 && c.vocab_size >= 1, PTTS_E_INVALID, "bad T5 config");
  PTTS_CHECK(c.d_model % 32 == 0 && c.d_ff % 32 == 0 && c.d_model >= 32 && c.d_ff >= 32, PTTS_E_UNSUPPORTED, "d_model / d_ff must be multiples of 32");
  PTTS_CHECK(c.rel_buckets >= 4 && c.rel_buckets % 4 == 0 && c.rel_max_distance > c.rel_buckets / 4, PTTS_E_INVALID, "bad relative-attention buckets %d / max distance %d", c.rel_buckets, c.rel_max_distance);
  PTTS_CHECK(c.max_batch >= 1 && c.max_len >= 1, PTTS_E_INVALID, "bad capacities");
  PTTS_DEVICE(c.device);
  ptts_t5* e = new ptts_t5();
  e->cfg = c;
  e->esize = c.dtype == PTTS_BF16 ? 2 : 4;
  e->inner = c.num_heads * c.d_kv;
  e->rows = c.max_batch * c.max_len;
  int rc = PTTS_OK;
  auto fail = [&](int r) { ptts_t5_destroy(e); return r; };
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(ptts_fail(PTTS_E_HIP, "hipStreamCreate failed"));
  const int D = c.d_model, F = c.d_ff, I = e->inner;
  const size_t es = e->esize;
#define A(expr) if ((rc = (expr)) != PTTS_OK) return fail(rc)
  e->L.resize(c.num_layers);
  for (int l = 0; l < c.num_layers; ++l) {
    T5Layer& w = e->L[l];
    A(e->alloc_bytes(&w.qkv, (size_t)3 * I * D * es));
    A(e->alloc_bytes(&w.o, (size_t)D * I * es));
    A(e->alloc_bytes(&w.wi, (size_t)2 * F * D * es));
    A(e->alloc_bytes(&w.wo, (size_t)D * F * es));
    A(e->alloc(&w.ln1, D)); A(e->alloc(&w.ln2, D));
    char nm[160];
    const char* mats[] = {"layer.0.SelfAttention.q.weight", "layer.0.SelfAttention.k.weight", "layer.0.SelfAttention.v.weight", "layer.0.SelfAttention.o.weight",
                          "layer.0.layer_norm.weight", "layer.1.DenseReluDense.wi_0.weight", "layer.1.DenseReluDense.wi_1.weight",
                          "layer.1.DenseReluDense.wo.weight", "layer.1.layer_norm.weight"};
    for (const char* m : mats) { snprintf(nm, sizeof nm, "encoder.block.%d.%s", l, m); e->required.insert(nm); }
  }
  e->required.insert("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight");
  e->required.insert("encoder.final_layer_norm.weight");
  e->required.insert("shared.weight");
  A(e->alloc_bytes(&e->embed, (size_t)c.vocab_size * D * es));
  A(e->alloc(&e->final_ln, D));
  A(e->alloc(&e->rel, (size_t)c.rel_buckets * c.num_heads));
  const int ndelta = 2 * c.max_len - 1;
  A(e->alloc(&e->bias, (size_t)c.num_heads * ndelta));
  A(e->alloc(&e->bucket_of, (size_t)ndelta));
  {
    std::vector<int> lut(ndelta);
    for (int d = 0; d < ndelta; ++d) lut[d] = ptts_t5_relative_bucket(d - (c.max_len - 1), c.rel_buckets, c.rel_max_distance);
    if (hipMemcpy(e->bucket_of, lut.data(), lut.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
      return fail(ptts_fail(PTTS_E_HIP, "hipMemcpy(relative-position bucket table) failed"));
  }
  const size_t rows = (size_t)e->rows;
  A(e->alloc(&e->h, rows * D));
  A(e->alloc(&e->qkv, rows * 3 * I));
  A(e->alloc(&e->ss, rows * (size_t)(D / 16) + 64));
  A(e->alloc_bytes(&e->xw, (rows + 16) * D * es));   // + 16 rows: fragment order addresses whole 16-row tiles
  A(e->alloc_bytes(&e->ctx, (rows + 16) * I * es));
  A(e->alloc_bytes(&e->ff, (rows + 16) * F * es));
  A(e->alloc(&e->ids, rows));
  A(e->alloc(&e->mask, rows));
#undef A
  e->use_fo = !(ptts_dev_env("PTTS_T5_NO_FO") && atoi(ptts_dev_env("PTTS_T5_NO_FO")));
  e->use_graph = !(getenv("PTTS_T5_NO_GRAPH") && atoi(getenv("PTTS_T5_NO_GRAPH")));
  e->use_fold = D % 64 == 0 && !(getenv("PTTS_T5_NO_FOLD") && atoi(getenv("PTTS_T5_NO_FOLD")));  // (a row's d_model / 16 partials are summed by 4 lanes)
  *out = e;
  return PTTS_OK;
}

extern "C" void ptts_t5_destroy(ptts_t5* e) {
  if (!e) return;
  PttsDeviceGuard _dg(e->cfg.device);
  hipDeviceSynchronize();
  for (auto& kv : e->graphs) hipGraphExecDestroy(kv.second);
  for (void* p : e->allocs) hipFree(p);
  if (e->own_stream) hipStreamDestroy(e->own_stream);
  delete e;
}

namespace {

template <typename WT>
int t5_pack(void* dst, const void* src, int src_dtype, int N, int K, int row0, hipStream_t st) {
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
template <typename WT>
int t5_pack_ilv(void* dst, const void* src, int src_dtype, int n_src, int K, int which, hipStream_t st) {
  constexpr int KT = Elem<WT>::KT;
  if (n_src % 8 || K % KT) return ptts_fail(PTTS_E_INVALID, "gated feed-forward weight [%d,%d] not a multiple of the 8x%d half tile", n_src, K, KT);
  const size_t total = (size_t)(2 * n_src / 16) * (K / KT) * 64;
  const int blocks = (int)((total + 255) / 256);
  if (src_dtype == PTTS_F32)
    hipLaunchKernelGGL((pack_weight_ilv_kernel<WT, float>), dim3(blocks), dim3(256), 0, st, (const float*)src, (WT*)dst, n_src, K, which);
  else
    hipLaunchKernelGGL((pack_weight_ilv_kernel<WT, bf16_t>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, (WT*)dst, n_src, K, which);
  return PTTS_OK;
}

template <typename WT>
int t5_forward(ptts_t5* e, int B, int N, bool has_mask, hipStream_t st) {
  const ptts_t5_config& c = e->cfg;
  const int D = c.d_model, F = c.d_ff, I = e->inner, M = B * N;
  const int fo = (e->use_fo && M <= 256) ? 1 : 0;  // the rows stay on the strip kernels: engine-dtype activations in MFMA B-fragment order
  hipLaunchKernelGGL((t5_embed_kernel<WT>), dim3(M), dim3(256), 0, st, reinterpret_cast<const WT*>(e->embed), e->ids, e->h, D, c.vocab_size);
  auto prep = [&](const float* gamma) -> int {
    GemmArgs p = {};
    p.x = e->h; p.x_ld = D; p.x_row_mul = 1; p.gamma = gamma; p.M = M; p.K = D; p.out_fo = fo; p.rms_eps = c.layer_norm_eps;
    return launch_prep<WT, PRO_RMS>(p, e->xw, st);
  };
  // <= 256 rows (the strip kernels): T5LayerNorm folded into the GEMMs around it - it has no mean and no bias, so W (g o x * rstd) = rstd * W (g o x):
  // the o / wo GEMMs' residual epilogues write the next GEMM's operand g o h and per-strip sums of h^2, the q|k|v and wi GEMMs scale their
  // accumulators by rstd (GemmArgs::rs_part / nx_out). 7 -> 5 nodes per block; block 0's first norm keeps its rows_prep node (PTTS_T5_NO_FOLD=1: all do)
  // (round 6, measured and closed: the same producer / consumer epilogues on the LDS-DMA GEMM above 256 rows - T5 at 32 x 64 tokens 2.89 -> 2.97 ms:
  //  the consumer's dependent round trip for its rows' partial sums at the end of every workgroup costs what the two removed 5 us nodes saved;
  //  profiles/r06_experiments.txt call 11)
  const bool fold = e->use_fold && M <= 256;
  const int mfma_mode = getenv("PTTS_T5_ATTN_MFMA") ? atoi(getenv("PTTS_T5_ATTN_MFMA")) : 2;  // read per forward (a test switches it inside one process)
  auto consume = [&](GemmArgs& g) { g.rs_part = e->ss; g.rs_n = D / 16; g.rs_invD = 1.0f / (float)D; g.rms_eps = c.layer_norm_eps; };
  auto produce = [&](GemmArgs& g, const float* gamma) { g.nx_out = e->xw; g.nx_gamma = gamma; g.ss_out = e->ss; g.out_fo = fo; };
  for (int l = 0; l < c.num_layers; ++l) {
    const T5Layer& w = e->L[l];
    if (!fold || l == 0) PTTS_TRY(prep(w.ln1));  // T5LayerSelfAttention: normed = layer_norm(hidden)
    {
      GemmArgs g = {};
      g.W = w.qkv; g.x = reinterpret_cast<const float*>(e->xw); g.x_ld = D; g.x_row_mul = 1; g.x_fo = fo;
      g.out = e->qkv; g.out_ld = 3 * I; g.M = M; g.N = 3 * I; g.K = D;
      if (fold && l > 0) consume(g);
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_STORE>(g, st)));
    }
    {
      T5AttnArgs a = {};
      a.qkv = e->qkv; a.ld = 3 * I; a.inner = I; a.bias = e->bias; a.bias_ld = 2 * c.max_len - 1; a.bias_zero = c.max_len - 1;
      a.mask = has_mask ? e->mask : nullptr; a.out = e->ctx; a.out_fo = fo; a.N = N;
      // batches that fill the chip with 64-query workgroups: the f32-MFMA kernel (PTTS_T5_ATTN_MFMA=0: the VALU kernel everywhere; =1: the MFMA kernel everywhere)
      if (mfma_mode == 1 || (mfma_mode == 2 && B * c.num_heads * ((N + 63) / 64) >= 128))
        t5_attn_launch(t5_attn_mfma_kernel<WT>, dim3((N + 63) / 64, c.num_heads, B), dim3(256), st, a);
      else
        t5_attn_launch(t5_attn_kernel<WT>, dim3((N + 7) / 8, c.num_heads, B), dim3(256), st, a);
    }
    {  // hidden = hidden + o(context)
      GemmArgs g = {};
      g.W = w.o; g.x = reinterpret_cast<const float*>(e->ctx); g.x_ld = I; g.x_row_mul = 1; g.x_fo = fo;
      g.out = e->h; g.out_ld = D; g.M = M; g.N = D; g.K = I;
      if (fold) produce(g, w.ln2);
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_RESID>(g, st)));
    }
    if (!fold) PTTS_TRY(prep(w.ln2));  // T5LayerFF: hidden + wo(gelu_new(wi_0 x) * wi_1 x)
    {
      GemmArgs g = {};
      g.W = w.wi; g.x = reinterpret_cast<const float*>(e->xw); g.x_ld = D; g.x_row_mul = 1; g.x_fo = fo;
      g.out = reinterpret_cast<float*>(e->ff); g.out_ld = F; g.out_fo = fo; g.M = M; g.N = 2 * F; g.K = D;
      if (fold) consume(g);
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_GATE_WT>(g, st)));
    }
    {
      GemmArgs g = {};
      g.W = w.wo; g.x = reinterpret_cast<const float*>(e->ff); g.x_ld = F; g.x_row_mul = 1; g.x_fo = fo;
      g.out = e->h; g.out_ld = D; g.M = M; g.N = D; g.K = F;
      if (fold && l + 1 < c.num_layers) produce(g, e->L[l + 1].ln1);
      PTTS_TRY((launch_gemm<WT, PRO_COPY, EPI_RESID>(g, st)));
    }
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "T5 forward launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}

int t5_forward_dispatch(ptts_t5* e, int B, int N, bool has_mask, hipStream_t st) {
  return e->cfg.dtype == PTTS_BF16 ? t5_forward<bf16_t>(e, B, N, has_mask, st) : t5_forward<float>(e, B, N, has_mask, st);
}

}  // namespace

extern "C" int ptts_t5_load_weight(ptts_t5* e, const char* name_c, const void* dev_ptr, int32_t src_dtype, const int64_t* shape, int32_t ndim,
                                   void* stream) {
  PTTS_CHECK(e && name_c && dev_ptr && shape, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(src_dtype == PTTS_F32 || src_dtype == PTTS_BF16, PTTS_E_INVALID, "src_dtype must be f32 or bf16");
  PTTS_DEVICE(e->cfg.device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const ptts_t5_config& c = e->cfg;
  const int D = c.d_model, F = c.d_ff, I = e->inner;
  const bool bf = c.dtype == PTTS_BF16;
  const std::string name(name_c);
  auto want = [&](int64_t a, int64_t b) -> int {
    if (b < 0) { if (ndim != 1 || shape[0] != a) return ptts_fail(PTTS_E_INVALID, "%s: expected shape [%lld]", name_c, (long long)a); }
    else if (ndim != 2 || shape[0] != a || shape[1] != b) return ptts_fail(PTTS_E_INVALID, "%s: expected shape [%lld, %lld]", name_c, (long long)a, (long long)b);
    return PTTS_OK;
  };
  if (name == "shared.weight" || name == "encoder.embed_tokens.weight") {  // one tensor under two names (tied)
    PTTS_TRY(want(c.vocab_size, D));
    if (bf) PTTS_TRY(t5_convert_into<bf16_t>((bf16_t*)e->embed, dev_ptr, src_dtype, (size_t)c.vocab_size * D, st));
    else PTTS_TRY(t5_convert_into<float>((float*)e->embed, dev_ptr, src_dtype, (size_t)c.vocab_size * D, st));
    e->loaded.insert("shared.weight");
    return PTTS_OK;
  }
  if (name == "encoder.final_layer_norm.weight") {
    PTTS_TRY(want(D, -1));
    PTTS_TRY(t5_convert_into<float>(e->final_ln, dev_ptr, src_dtype, D, st));
    e->loaded.insert(name);
    return PTTS_OK;
  }
  int l = -1;
  char tail[128] = {0};
  if (sscanf(name_c, "encoder.block.%d.%127s", &l, tail) == 2) {
    PTTS_CHECK(l >= 0 && l < c.num_layers, PTTS_E_INVALID, "%s: block index out of range", name_c);
    T5Layer& w = e->L[l];
    const std::string t(tail);
    if (t == "layer.0.SelfAttention.relative_attention_bias.weight") {
      PTTS_CHECK(l == 0, PTTS_E_INVALID, "%s: only block 0 holds the relative attention bias", name_c);
      PTTS_TRY(want(c.rel_buckets, c.num_heads));
      PTTS_TRY(t5_convert_into<float>(e->rel, dev_ptr, src_dtype, (size_t)c.rel_buckets * c.num_heads, st));
      const int ndelta = 2 * c.max_len - 1, n = c.num_heads * ndelta;
      hipLaunchKernelGGL(t5_bias_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, e->rel, e->bucket_of, e->bias, c.num_heads, ndelta);
      e->loaded.insert(name);
      return PTTS_OK;
    }
    struct { const char* n; void* dst; int N, Kd, row0; } mats[] = {
        {"layer.0.SelfAttention.q.weight", w.qkv, I, D, 0}, {"layer.0.SelfAttention.k.weight", w.qkv, I, D, I}, {"layer.0.SelfAttention.v.weight", w.qkv, I, D, 2 * I},
        {"layer.0.SelfAttention.o.weight", w.o, D, I, 0}, {"layer.1.DenseReluDense.wo.weight", w.wo, D, F, 0}};
    for (auto& m : mats)
      if (t == m.n) {
        PTTS_TRY(want(m.N, m.Kd));
        PTTS_TRY(bf ? t5_pack<bf16_t>(m.dst, dev_ptr, src_dtype, m.N, m.Kd, m.row0, st) : t5_pack<float>(m.dst, dev_ptr, src_dtype, m.N, m.Kd, m.row0, st));
        e->loaded.insert(name);
        return PTTS_OK;
      }
    for (int which = 0; which < 2; ++which)
      if (t == (which ? "layer.1.DenseReluDense.wi_1.weight" : "layer.1.DenseReluDense.wi_0.weight")) {
        PTTS_TRY(want(F, D));
        PTTS_TRY(bf ? t5_pack_ilv<bf16_t>(w.wi, dev_ptr, src_dtype, F, D, which, st) : t5_pack_ilv<float>(w.wi, dev_ptr, src_dtype, F, D, which, st));
        e->loaded.insert(name);
        return PTTS_OK;
      }
    for (int which = 0; which < 2; ++which)
      if (t == (which ? "layer.1.layer_norm.weight" : "layer.0.layer_norm.weight")) {
        PTTS_TRY(want(D, -1));
        PTTS_TRY(t5_convert_into<float>(which ? w.ln2 : w.ln1, dev_ptr, src_dtype, D, st));
        e->loaded.insert(name);
        return PTTS_OK;
      }
  }
  return ptts_fail(PTTS_E_INVALID, "unknown tensor name %s", name_c);
}

extern "C" int ptts_t5_debug_graph_nodes(ptts_t5* e, int32_t* nodes) {
  PTTS_CHECK(e && nodes, PTTS_E_INVALID, "null argument");
  *nodes = e->last_graph_nodes;
  return PTTS_OK;
}

extern "C" int ptts_t5_weights_ready(ptts_t5* e) {
  PTTS_CHECK(e, PTTS_E_INVALID, "null engine");
  std::string missing;
  int n = 0;
  for (const auto& r : e->required)
    if (!e->loaded.count(r)) { if (n++ < 8) missing += (missing.empty() ? "" : ", ") + r; }
  if (n) return ptts_fail(PTTS_E_MISSING, "%d tensors not loaded: %s%s", n, missing.c_str(), n > 8 ? ", ..." : "");
  return PTTS_OK;
}

extern "C" int ptts_t5_encode(ptts_t5* e, const int64_t* ids_dev, const int32_t* mask_dev, int32_t B, int32_t N, float* out_dev, void* stream) {
  PTTS_CHECK(e && ids_dev && out_dev, PTTS_E_INVALID, "null argument");
  PTTS_TRY(ptts_t5_weights_ready(e));
  const ptts_t5_config& c = e->cfg;
  PTTS_CHECK(B >= 1 && B <= c.max_batch, PTTS_E_CAPACITY, "batch %d exceeds engine max_batch %d", B, c.max_batch);
  PTTS_CHECK(N >= 1 && N <= c.max_len, PTTS_E_CAPACITY, "description length %d exceeds engine max_len %d", N, c.max_len);
  PTTS_DEVICE(c.device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int M = B * N;
  // the captured graph reads its inputs from the engine's own buffers
  PTTS_HIP(hipMemcpyAsync(e->ids, ids_dev, (size_t)M * 8, hipMemcpyDeviceToDevice, st));
  if (mask_dev) PTTS_HIP(hipMemcpyAsync(e->mask, mask_dev, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
  if (!e->use_graph) {
    PTTS_TRY(t5_forward_dispatch(e, B, N, mask_dev != nullptr, st));
  } else {
    const long long key = ((long long)B << 32) | ((long long)N << 1) | (mask_dev ? 1 : 0);
    auto it = e->graphs.find(key);
    hipGraphExec_t ex = nullptr;
    if (it != e->graphs.end()) {
      ex = it->second;
    } else {
      hipGraph_t g = nullptr;
      PTTS_HIP(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
      const int rc = t5_forward_dispatch(e, B, N, mask_dev != nullptr, e->own_stream);
      hipError_t ce = hipStreamEndCapture(e->own_stream, &g);
      if (rc != PTTS_OK) { if (g) hipGraphDestroy(g); return rc; }
      if (ce != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
      size_t nn = 0;
      if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) e->last_graph_nodes = (int)nn;
      hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (ie != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
      if (e->graphs.size() > 64) {  // bounded: a server with many (batch, length) shapes re-captures instead of growing without limit
        PTTS_HIP(hipStreamSynchronize(st));  // an evicted graph may still be in flight on the caller's stream
        for (auto& kv : e->graphs) hipGraphExecDestroy(kv.second);
        e->graphs.clear();
      }
      e->graphs[key] = ex;
    }
    PTTS_HIP(hipGraphLaunch(ex, st));
  }
  // encoder.final_layer_norm straight into the caller's buffer (fp32), masked positions zeroed (modeling_parler_tts.py:3093-3097)
  GemmArgs p = {};
  p.x = e->h; p.x_ld = c.d_model; p.x_row_mul = 1; p.gamma = e->final_ln; p.M = M; p.K = c.d_model; p.rms_eps = c.layer_norm_eps;
  p.row_keep = mask_dev ? e->mask : nullptr;
  PTTS_TRY((launch_prep<float, PRO_RMS>(p, out_dev, st)));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return ptts_fail(PTTS_E_HIP, "T5 final norm launch failed: %s", hipGetErrorString(err));
  return PTTS_OK;
}
