// Single-utterance decode step (M == 1): row-per-wave GEMV kernels (gfx950 / CDNA4, wave64).
//
// Why not the MFMA strip kernel at M == 1 (profiles/r01_bench_bs1_rocprof_summary.txt, r01_sync_and_chain_probes.txt):
// a 16-row MFMA strip fixes the unit of work at 16 x K weights, so the N = 1024 projections ran on 64 workgroups
// (32 KB per CU), fc2 on 64 x 128 KB and the fused cross block on 16 x 128 KB, while one CU pulls only ~25-50 GB/s:
// the step was bound by the per-CU stream of the least parallel kernels plus two workgroup barriers and an LDS
// cross-wave reduction per node. Here the unit of work is ONE weight row per wave:
//   * weights row-major [N][K] in the engine dtype (a second copy beside the MFMA-packed one, made at load time);
//     lane l of a wave reads 16 B chunks c*1 KiB + 16 l of each of its R rows: every wave-load is 1 KiB contiguous,
//     all R x NCH loads of a wave are in flight before the first wait, N / R waves cover all 256 CUs evenly
//     (1024 waves: 8 KB per CU for the N = 1024 projections, 32 KB for fc1 / fc2);
//   * the dot product runs on the VALU (v_dot2c_f32_bf16 / v_fma_f32: an M = 1 product has no reuse for the matrix
//     core to exploit), the k reduction is one DPP / permlane wave reduction per row: no LDS, no cross-wave step;
//   * LayerNorm / split-KV combine is done ONCE per workgroup by a dedicated prologue wave that owns no weights
//     (its VMEM queue holds only the 4 KB row + gamma/beta, so nothing it waits for sits behind a weight burst)
//     and hands the normalised row to the 4 GEMV waves through LDS in the engine dtype: one barrier per kernel;
//   * COPY prologue (activations already final, in the engine dtype): no LDS and no barrier at all.
// Reference semantics: modeling_parler_tts.py:983-1074 (layer), :1917-1960 (heads); LayerNorm eps 1e-5 (:961).
#pragma once
#include "ptts_common.h"

enum { GV_LN = 0, GV_ATTN = 1, GV_COPY = 2 };
enum { GV_STORE = 0, GV_RESID = 1, GV_GELU_WT = 2 };

struct GemvArgs {
  const void* W;       // row-major [N][K], engine dtype
  const float* x;      // GV_LN: residual-stream row, fp32 [K]
  const void* xw;      // GV_COPY: activation row in the engine dtype [K]
  const float* gamma;  // GV_LN
  const float* beta;
  const float* part;   // GV_ATTN: split-KV partials [S][K] (unnormalised) ...
  const float* stats;  // ... and their (max, sumexp) per head [S][nheads][2]
  float* out;          // GV_STORE / GV_RESID: fp32 [N]; GV_GELU_WT: engine dtype [N]
  int N, K, nheads;
  float invK;
};

__device__ __forceinline__ float gv_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename WT> struct GvDot;
template <> struct GvDot<bf16_t> {  // 8 bf16 x 8 bf16 -> fp32 (products of two bf16 are exact in fp32)
  typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
  static __device__ __forceinline__ float run(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.x), __builtin_bit_cast(v2bf, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.y), __builtin_bit_cast(v2bf, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.z), __builtin_bit_cast(v2bf, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.w), __builtin_bit_cast(v2bf, x.w), acc, false);
    return acc;
  }
};
template <> struct GvDot<float> {
  static __device__ __forceinline__ float run(const uint4& w, const uint4& x, float acc) {
    acc = fmaf(__uint_as_float(w.x), __uint_as_float(x.x), acc);
    acc = fmaf(__uint_as_float(w.y), __uint_as_float(x.y), acc);
    acc = fmaf(__uint_as_float(w.z), __uint_as_float(x.z), acc);
    acc = fmaf(__uint_as_float(w.w), __uint_as_float(x.w), acc);
    return acc;
  }
};

template <typename WT> __device__ __forceinline__ void gv_lds_store4(char* base, int k, float a, float b, float c, float d);
template <> __device__ __forceinline__ void gv_lds_store4<float>(char* base, int k, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(base + (size_t)k * 4) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void gv_lds_store4<bf16_t>(char* base, int k, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(base + (size_t)k * 2) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}

// Both LayerNorm sums in ONE reduction tree: a permlane32 swap puts sum-halves of s1 in lanes 0-31 and of s2 in lanes
// 32-63, five more steps finish both (6 dependent cross-lane ops instead of 12); read back with v_readlane.
__device__ __forceinline__ void gv_pair_sum(float& s1, float& s2) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
  float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lanes 0-31: s1[l] + s1[l+32]; lanes 32-63: s2[l-32] + s2[l]
  t = swap16_reduce<OpSum>(t);
  t = group_reduce<OpSum, 16>(t);
  s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
  s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32));
}

// prologue wave, LayerNorm: row held in registers (K == NF4 * 256), shifted one-pass mean / variance
template <typename WT, int NF4>
__device__ __forceinline__ void gv_ln_wave(const GemvArgs& a, char* s_x, int lane) {
  float4 v[NF4], g[NF4], bt[NF4];
#pragma unroll
  for (int i = 0; i < NF4; ++i) v[i] = *reinterpret_cast<const float4*>(a.x + (lane + 64 * i) * 4);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    g[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
    bt[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
  }
  __builtin_amdgcn_sched_barrier(0);  // all 3 * NF4 loads are issued before the first wait (gamma / beta were sunk otherwise)
  const float c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[0].x)));
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const float d0 = v[i].x - c, d1 = v[i].y - c, d2 = v[i].z - c, d3 = v[i].w - c;
    s1 += (d0 + d1) + (d2 + d3);
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  gv_pair_sum(s1, s2);
  const float dm = s1 * a.invK, mean = c + dm;
  const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
  for (int i = 0; i < NF4; ++i)
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, (v[i].x - mean) * rstd * g[i].x + bt[i].x, (v[i].y - mean) * rstd * g[i].y + bt[i].y,
                      (v[i].z - mean) * rstd * g[i].z + bt[i].z, (v[i].w - mean) * rstd * g[i].w + bt[i].w);
}

// prologue wave, split-KV combine of the attention partials (attn_kernel wrote unnormalised sums + (max, sumexp) per
// split and head): every load of the wave is issued before the first exp
template <typename WT, int NF4, int S>
__device__ __forceinline__ void gv_attn_wave(const GemvArgs& a, char* s_x, int lane) {
  float4 p[NF4][S];
  float2 st[NF4][S];
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int k = (lane + 64 * i) * 4, head = k >> 6;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) {
      st[i][sp] = *reinterpret_cast<const float2*>(a.stats + ((size_t)sp * a.nheads + head) * 2);
      p[i][sp] = *reinterpret_cast<const float4*>(a.part + (size_t)sp * a.K + k);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // every load in flight before the first exp
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    float mx = -INFINITY;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) mx = fmaxf(mx, st[i][sp].x);
    float den = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int sp = 0; sp < S; ++sp) {
      const float w = (st[i][sp].x == -INFINITY) ? 0.f : __expf(st[i][sp].x - mx);
      den += w * st[i][sp].y;
      o.x += w * p[i][sp].x; o.y += w * p[i][sp].y; o.z += w * p[i][sp].z; o.w += w * p[i][sp].w;
    }
    const float inv = den > 0.f ? __frcp_rn(den) : 0.f;
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
}

// NCH = 16-byte chunks per lane per weight row (K * sizeof(WT) == NCH * 1024), R = rows per wave, S = KV splits (GV_ATTN).
template <typename WT, int NCH, int R, int PRO, int EPI, int S>
__global__ void __launch_bounds__(PRO == GV_COPY ? 256 : 320) gemv_kernel(GemvArgs a) {
  constexpr bool HASPRO = PRO != GV_COPY;
  constexpr int EPL = Elem<WT>::EPL;
  constexpr int NF4 = NCH * EPL / 4;  // float4 per lane of the fp32 row (K / 256)
  extern __shared__ __attribute__((aligned(16))) char s_x[];  // HASPRO: the prepared row, engine dtype [K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (HASPRO && wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    if (PRO == GV_LN) gv_ln_wave<WT, NF4>(a, s_x, lane);
    else gv_attn_wave<WT, NF4, S>(a, s_x, lane);
    __syncthreads();
    return;
  }
  const int gw = blockIdx.x * 4 + wave - (HASPRO ? 1 : 0);
  const int r0 = gw * R;
  // everything this wave will ever load goes in flight now: residual values, (COPY) its activation chunks, its weights
  float res_pre = 0.f;
  if (EPI == GV_RESID && lane < R && r0 + lane < a.N) res_pre = a.out[r0 + lane];
  uint4 xv[NCH];
  if (!HASPRO) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) xv[c] = reinterpret_cast<const uint4*>(a.xw)[c * 64 + lane];
  }
  uint4 wv[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = min(r0 + r, a.N - 1);  // clamped rows are computed and dropped
    const uint4* wp = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.W) + (size_t)row * a.K) + lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c) wv[r][c] = ld_nt16(wp + c * 64);
  }
  if (HASPRO) {
    __builtin_amdgcn_sched_barrier(0);  // the weight loads stay above the barrier
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) xv[c] = *reinterpret_cast<const uint4*>(s_x + (size_t)(c * 64 + lane) * 16);
  }
  float acc[R], acc2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (c & 1) acc2[r] = GvDot<WT>::run(wv[r][c], xv[c], acc2[r]);
      else acc[r] = GvDot<WT>::run(wv[r][c], xv[c], acc[r]);
    }
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float t = wave_sum(acc[r] + acc2[r]);
    v = lane == r ? t : v;
  }
  if (lane < R && r0 + lane < a.N) {
    if (EPI == GV_STORE) a.out[r0 + lane] = v;
    else if (EPI == GV_RESID) a.out[r0 + lane] = res_pre + v;
    else store_from_f32<WT>(reinterpret_cast<WT*>(a.out) + r0 + lane, gv_gelu_erf(v));
  }
}
