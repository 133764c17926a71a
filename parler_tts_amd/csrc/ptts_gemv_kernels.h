// Decode step at batch 1..4: row-per-wave GEMV kernels (gfx950 / CDNA4, wave64).
//
// Why not the MFMA strip kernel at these batch sizes (profiles/r01_bench_bs1_rocprof_summary.txt, r01_sync_and_chain_probes.txt):
// a 16-row MFMA strip fixes the unit of work at 16 x K weights, so the N = 1024 projections ran on 64 workgroups
// (32 KB per CU), fc2 on 64 x 128 KB and the fused cross block on 16 x 128 KB, while one CU pulls only ~25-50 GB/s:
// the step was bound by the per-CU stream of the least parallel kernels plus two workgroup barriers and an LDS
// cross-wave reduction per node. Here the unit of work is ONE weight row per wave:
//   * weights row-major [N][K] (a second copy beside the MFMA-packed one, made at load time) in the engine dtype or,
//     in W8 mode, as OCP e4m3 bytes with one power-of-two scale per row; lane l of a wave reads chunk c*64 + l of each
//     of its R rows (16 B = 8 bf16 / 4 fp32, or 8 B = 8 e4m3): every wave-load is contiguous, all R x NCH loads of a
//     wave are in flight before the first wait, N / R waves cover all 256 CUs evenly;
//   * the dot products run on the VALU (v_dot2c_f32_bf16 / v_fma_f32; W8: v_cvt_pk_f32_fp8 + v_fma_f32) for each of
//     the M <= 4 utterances against the SAME weight registers, the k reduction is one DPP / permlane wave reduction
//     per (utterance, row): no LDS, no cross-wave step;
//   * LayerNorm / split-KV combine is done ONCE per workgroup by dedicated prologue waves (one per utterance) that own
//     no weights (their VMEM queue holds only the row + gamma/beta, so nothing they wait for sits behind a weight
//     burst) and hand the normalised rows to the 4 GEMV waves through LDS in the engine dtype: one barrier per kernel;
//   * COPY prologue (activations already final, in the engine dtype): no LDS and no barrier at all.
// Reference semantics: modeling_parler_tts.py:983-1074 (layer), :1917-1960 (heads); LayerNorm eps 1e-5 (:961).
#pragma once
#include "ptts_common.h"
#include "ptts_gemv.h"

__device__ __forceinline__ float gv_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename DT> __device__ __forceinline__ void gv_store(DT* p, float v);
template <> __device__ __forceinline__ void gv_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void gv_store<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// one activation chunk (16 B: 8 bf16 / 4 fp32) against one weight chunk
template <typename WT, bool W8> struct GvDot;
template <> struct GvDot<bf16_t, false> {  // 8 bf16 x 8 bf16 -> fp32 (products of two bf16 are exact in fp32)
  typedef uint4 WV;
  typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
  static __device__ __forceinline__ float run(const uint4& w, const uint4& x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.x), __builtin_bit_cast(v2bf, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.y), __builtin_bit_cast(v2bf, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.z), __builtin_bit_cast(v2bf, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, w.w), __builtin_bit_cast(v2bf, x.w), acc, false);
    return acc;
  }
};
template <> struct GvDot<float, false> {
  typedef uint4 WV;
  static __device__ __forceinline__ float run(const uint4& w, const uint4& x, float acc) {
    acc = fmaf(__uint_as_float(w.x), __uint_as_float(x.x), acc);
    acc = fmaf(__uint_as_float(w.y), __uint_as_float(x.y), acc);
    acc = fmaf(__uint_as_float(w.z), __uint_as_float(x.z), acc);
    acc = fmaf(__uint_as_float(w.w), __uint_as_float(x.w), acc);
    return acc;
  }
};
template <> struct GvDot<bf16_t, true> {  // 8 e4m3 weights (exact in fp32) x 8 bf16 activations (exact in fp32), fp32 fma chain
  typedef uint2 WV;
  static __device__ __forceinline__ float run(const uint2& w, const uint4& x, float acc) {
    const auto w01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.x, false), w23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.x, true);
    const auto w45 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.y, false), w67 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w.y, true);
    acc = fmaf(w01[0], __uint_as_float(x.x << 16), acc); acc = fmaf(w01[1], __uint_as_float(x.x & 0xffff0000u), acc);
    acc = fmaf(w23[0], __uint_as_float(x.y << 16), acc); acc = fmaf(w23[1], __uint_as_float(x.y & 0xffff0000u), acc);
    acc = fmaf(w45[0], __uint_as_float(x.z << 16), acc); acc = fmaf(w45[1], __uint_as_float(x.z & 0xffff0000u), acc);
    acc = fmaf(w67[0], __uint_as_float(x.w << 16), acc); acc = fmaf(w67[1], __uint_as_float(x.w & 0xffff0000u), acc);
    return acc;
  }
};
template <typename WV> __device__ __forceinline__ WV gv_ld_nt(const WV* p);
template <> __device__ __forceinline__ uint4 gv_ld_nt<uint4>(const uint4* p) { return ld_nt16(p); }
template <> __device__ __forceinline__ uint2 gv_ld_nt<uint2>(const uint2* p) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
  return make_uint2(v.x, v.y);
}

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

// prologue wave, LayerNorm of one row: row held in registers (K == NF4 * 256), shifted one-pass mean / variance
template <typename WT, int NF4>
__device__ __forceinline__ void gv_ln_row(const float* xr, const float* gamma, const float* beta, float invK, char* s_x, int lane) {
  float4 v[NF4], g[NF4], bt[NF4];
#pragma unroll
  for (int i = 0; i < NF4; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    g[i] = *reinterpret_cast<const float4*>(gamma + (lane + 64 * i) * 4);
    bt[i] = *reinterpret_cast<const float4*>(beta + (lane + 64 * i) * 4);
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
  const float dm = s1 * invK, mean = c + dm;
  const float rstd = rsqrtf(fmaxf(s2 * invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
  for (int i = 0; i < NF4; ++i)
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, (v[i].x - mean) * rstd * g[i].x + bt[i].x, (v[i].y - mean) * rstd * g[i].y + bt[i].y,
                      (v[i].z - mean) * rstd * g[i].z + bt[i].z, (v[i].w - mean) * rstd * g[i].w + bt[i].w);
}

template <typename WT, int NF4>
__device__ __forceinline__ void gv_ln_wave(const GemvArgs& a, const float* xr, char* s_x, int lane) {
  gv_ln_row<WT, NF4>(xr, a.gamma, a.beta, a.invK, s_x, lane);
}

// prologue wave, split-KV combine of one row's attention partials (attn_kernel wrote unnormalised sums + (max, sumexp) per
// split and head): every load of the wave is issued before the first exp
template <typename WT, int NF4, int S>
__device__ __forceinline__ void gv_attn_wave(const GemvArgs& a, const float* part, const float* stats, char* s_x, int lane) {
  float4 p[NF4][S];
  float2 st[NF4][S];
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int k = (lane + 64 * i) * 4, head = k >> 6;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) {
      st[i][sp] = *reinterpret_cast<const float2*>(stats + ((size_t)sp * a.nheads + head) * 2);
      p[i][sp] = *reinterpret_cast<const float4*>(part + (size_t)sp * (NF4 * 256) + k);
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
      const float w = (st[i][sp].x == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[i][sp].x - mx);  // statistics are in log2 units (attn_kernel)
      den += w * st[i][sp].y;
      o.x += w * p[i][sp].x; o.y += w * p[i][sp].y; o.z += w * p[i][sp].z; o.w += w * p[i][sp].w;
    }
    const float inv = den > 0.f ? __frcp_rn(den) : 0.f;
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
}

// prologue wave, combine of qkv_attn_kernel's partials (single utterance): S cache splits as above + slot S = the new position, whose
// base-2 score arrives as two half dot products (q . k over head dimensions 0-31 / 32-63, from two workgroups) and whose value row is
// the partial itself (weight 1 before normalisation)
template <typename WT, int NF4, int S>
__device__ __forceinline__ void gv_attn2_wave(const GemvArgs& a, const float* part, const float* stats, char* s_x, int lane) {
  float4 p[NF4][S + 1];
  float2 st[NF4][S + 1];
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int k = (lane + 64 * i) * 4, head = k >> 6;
#pragma unroll
    for (int sp = 0; sp <= S; ++sp) {
      st[i][sp] = *reinterpret_cast<const float2*>(stats + ((size_t)sp * a.nheads + head) * 2);
      p[i][sp] = *reinterpret_cast<const float4*>(part + (size_t)sp * (NF4 * 256) + k);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const float m_new = st[i][S].x + st[i][S].y;
    float mx = m_new;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) mx = fmaxf(mx, st[i][sp].x);
    float den = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int sp = 0; sp < S; ++sp) {
      const float w = (st[i][sp].x == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(st[i][sp].x - mx);
      den += w * st[i][sp].y;
      o.x += w * p[i][sp].x; o.y += w * p[i][sp].y; o.z += w * p[i][sp].z; o.w += w * p[i][sp].w;
    }
    const float wn = __builtin_amdgcn_exp2f(m_new - mx);
    den += wn;
    o.x += wn * p[i][S].x; o.y += wn * p[i][S].y; o.z += wn * p[i][S].z; o.w += wn * p[i][S].w;
    const float inv = __frcp_rn(den);  // den >= the weight of the larger of (cache maximum, new position) = 1
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
}

// prologue waves of GV_LNP (single utterance): LayerNorm of x + the npart per-head partial rows of xfold_attn_kernel. One wave per 256
// columns (17+ rows x K fp32 do not fit one wave's registers): every wave adds the rows in order, takes its slab's shifted mean / M2,
// the NF4 slabs are merged by the parallel-variance formula through LDS (one extra workgroup barrier, shared by the weight waves).
// Workgroup 0 also publishes the summed row (hsum): the fc2 node's residual operand, bit-identical to what the LayerNorm saw.
template <typename WT, int NF4>
__device__ __forceinline__ void gv_lnp_waves(const GemvArgs& a, char* s_x, float* s_st, int wave, int lane) {
  const int k = (wave * 64 + lane) * 4;
  float4 v = *reinterpret_cast<const float4*>(a.x + k);
  float4 p[GV_PMAX];
#pragma unroll
  for (int i = 0; i < GV_PMAX; ++i)
    if (i < a.npart) p[i] = *reinterpret_cast<const float4*>(a.part + (size_t)i * (NF4 * 256) + k);
  const float4 g = *reinterpret_cast<const float4*>(a.gamma + k), bt = *reinterpret_cast<const float4*>(a.beta + k);
  __builtin_amdgcn_sched_barrier(0);  // every load of the wave is issued before the first wait
#pragma unroll
  for (int i = 0; i < GV_PMAX; ++i)
    if (i < a.npart) { v.x += p[i].x; v.y += p[i].y; v.z += p[i].z; v.w += p[i].w; }
  if (blockIdx.x == 0 && a.hsum) *reinterpret_cast<float4*>(a.hsum + k) = v;
  const float c = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.x)));
  const float d0 = v.x - c, d1 = v.y - c, d2 = v.z - c, d3 = v.w - c;
  float s1 = (d0 + d1) + (d2 + d3), s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  gv_pair_sum(s1, s2);
  const float dm = s1 * (1.0f / 256.0f);
  if (lane == 0) { s_st[wave * 2] = c + dm; s_st[wave * 2 + 1] = fmaxf(s2 - s1 * dm, 0.f); }
  __syncthreads();
  float mean = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < NF4; ++i) mean += s_st[i * 2];
  mean *= 1.0f / (float)NF4;
#pragma unroll
  for (int i = 0; i < NF4; ++i) { const float dd = s_st[i * 2] - mean; m2 += s_st[i * 2 + 1] + 256.0f * dd * dd; }
  const float rstd = rsqrtf(m2 * a.invK + 1e-5f);
  gv_lds_store4<WT>(s_x, k, (v.x - mean) * rstd * g.x + bt.x, (v.y - mean) * rstd * g.y + bt.y, (v.z - mean) * rstd * g.z + bt.z,
                    (v.w - mean) * rstd * g.w + bt.w);
}

// prologue wave, folded cross-attention (static description K/V folded into the projections at prefill: scores = M x, out = U p):
// per-head softmax of the K = heads * NE base-2 scores. A head's NE scores sit in NE/4 consecutive lanes of one float4 slot
// (16 lanes for NE = 64, 8 for NE = 32): max and sum are DPP reductions inside that lane group. Masked / absent positions get 0.
template <typename WT, int NF4>
__device__ __forceinline__ void gv_softmax_wave(const GemvArgs& a, char* s_x, int lane) {
  float4 sc[NF4];
  int4 mk[NF4];
  const int nv = *a.n_valid;
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int k = (lane + 64 * i) * 4;
    sc[i] = *reinterpret_cast<const float4*>(a.x + k);
    mk[i] = a.mask ? *reinterpret_cast<const int4*>(a.mask + (k & (a.ne - 1))) : make_int4(1, 1, 1, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NF4; ++i) {
    const int n = ((lane + 64 * i) * 4) & (a.ne - 1);
    const bool v0 = n < nv && mk[i].x != 0, v1 = n + 1 < nv && mk[i].y != 0, v2 = n + 2 < nv && mk[i].z != 0, v3 = n + 3 < nv && mk[i].w != 0;
    const float s0 = v0 ? sc[i].x : -INFINITY, s1 = v1 ? sc[i].y : -INFINITY, s2 = v2 ? sc[i].z : -INFINITY, s3 = v3 ? sc[i].w : -INFINITY;
    float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
    mx = a.ne == 64 ? group_reduce<OpMax, 16>(mx) : group_reduce<OpMax, 8>(mx);
    const float e0 = v0 ? __builtin_amdgcn_exp2f(s0 - mx) : 0.f, e1 = v1 ? __builtin_amdgcn_exp2f(s1 - mx) : 0.f;
    const float e2 = v2 ? __builtin_amdgcn_exp2f(s2 - mx) : 0.f, e3 = v3 ? __builtin_amdgcn_exp2f(s3 - mx) : 0.f;
    float sm = (e0 + e1) + (e2 + e3);
    sm = a.ne == 64 ? group_reduce<OpSum, 16>(sm) : group_reduce<OpSum, 8>(sm);
    const float inv = sm > 0.f ? 1.0f / sm : 0.f;
    gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, e0 * inv, e1 * inv, e2 * inv, e3 * inv);
  }
}

// NCH = chunks per lane per row (K = NCH * 64 * EPL), R = weight rows per wave, S = KV splits (GV_ATTN), MB = utterances the
// instance is built for (1, 4 or GV_MAX_ROWS = 8; a.M <= MB of them are live), W8 = e4m3 weights + per-row scale.
// MB = 8 (batch 5..8): the activation chunks of MG utterances sit in registers at a time (MG * NCH <= 40 vectors), the wave's weight
// registers are reused for every group; groups past the live batch are skipped (workgroup-uniform).
template <typename WT, int NCH, int R, int PRO, int EPI, int S, int MB, bool W8, bool STG = false>
__global__ void __launch_bounds__(((PRO == GV_COPY ? 0 : (PRO == GV_LNP ? NCH * Elem<WT>::EPL / 4 : MB)) + 4) * 64) gemv_kernel(GemvArgs_KPARAMS) {
  GemvArgs_KJOIN(a)
  static_assert(!STG || (PRO == GV_COPY && MB == 8), "staged activation rows: GV_COPY nodes of the 5..8-utterance instances only");
  constexpr bool HASPRO = PRO != GV_COPY;
  constexpr int EPL = Elem<WT>::EPL;
  constexpr int NF4 = NCH * EPL / 4;  // float4 per lane of one fp32 row (K / 256)
  constexpr int NPW = PRO == GV_LNP ? NF4 : (HASPRO ? MB : 0);  // prologue waves (GV_LNP: one per 256 columns of the single row)
  static_assert(PRO != GV_LNP || MB == 1, "GV_LNP: single utterance");
  __shared__ float s_st[PRO == GV_LNP ? 2 * NF4 : 1];
  constexpr int MG = (MB <= 4 || MB * NCH <= 40) ? MB : ((MB / 2) * NCH <= 40 ? MB / 2 : MB / 4);  // utterances per register group
  constexpr int NG = MB / MG;
  typedef typename GvDot<WT, W8>::WV WV;
  extern __shared__ __attribute__((aligned(16))) char s_x[];  // HASPRO: the prepared rows, engine dtype [MB][K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROW_BYTES = NCH * 64 * 16;  // K * sizeof(WT), from the template: no kernel argument is read before the branch below,
  if (wave == 0 || wave == NPW) GV_STAMP(a, wave == 0 ? 0 : 2);  // entry: wave 0 (a prologue wave when the node has one), first weight wave
  if (HASPRO && wave < NPW) {             // so each kind of wave fetches its arguments in ONE scalar round trip (two cost ~0.12 us per node)
    __builtin_amdgcn_s_setprio(3);
    if constexpr (PRO == GV_LNP) {
      gv_lnp_waves<WT, NF4>(a, s_x, s_st, wave, lane);
    } else if (MB == 1 || wave < a.M) {
      constexpr int KK = NF4 * 256;              // = a.K
      const size_t pw = MB == 1 ? 0 : wave;      // single utterance: no tail argument (x_ld) in front of the row's loads
      if (PRO == GV_LN) gv_ln_wave<WT, NF4>(a, a.x + (MB == 1 ? 0 : pw * a.x_ld), s_x + pw * ROW_BYTES, lane);
      else if (PRO == GV_SOFTMAX) gv_softmax_wave<WT, NF4>(a, s_x, lane);  // single utterance only
      else if (PRO == GV_ATTN2) gv_attn2_wave<WT, NF4, S>(a, a.part + pw * (S + 1) * KK, a.stats + pw * (S + 1) * a.nheads * 2, s_x + pw * ROW_BYTES, lane);
      else gv_attn_wave<WT, NF4, S>(a, a.part + pw * S * KK, a.stats + pw * S * a.nheads * 2, s_x + pw * ROW_BYTES, lane);
    }
    if (wave == 0) GV_STAMP(a, 1);  // prologue: operands landed, row prepared, LDS stores issued
    __syncthreads();
    return;
  }
  const int gw = blockIdx.x * 4 + wave - NPW;
  const int r0 = gw * R;
  // everything this wave will ever load goes in flight now: residual values, row scales, (COPY) its activation chunks, its weights
  const int em = lane / R, er = lane - em * R;            // epilogue role of this lane: (utterance, row) = (em, er)
  const bool elive = lane < MB * R && (MB == 1 || em < a.M) && r0 + er < a.N;
  float res_pre = 0.f, wsc = 1.f;
  if (EPI == GV_RESID && elive) res_pre = a.resid[(MB == 1 ? 0 : (size_t)em * a.out_ld) + r0 + er];  // the launcher points resid at out when the caller left it null
  uint4 xv[MG][NCH];
  auto load_x = [&](int g) __attribute__((always_inline)) {  // activation chunks of utterances g*MG .. g*MG + MG - 1 (absent ones clamped: computed, dropped)
#pragma unroll
    for (int m = 0; m < MG; ++m) {
      const int mm = MB == 1 ? 0 : min(g * MG + m, a.M - 1);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (HASPRO || STG) xv[m][c] = *reinterpret_cast<const uint4*>(s_x + (size_t)mm * ROW_BYTES + (size_t)(c * 64 + lane) * 16);
        else xv[m][c] = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.xw) + (size_t)mm * a.xw_ld)[c * 64 + lane];
      }
    }
  };
  // GV_COPY with 5..8 utterances: each wave of the plain path pulls all M activation rows out of the L2 for R weight rows (8 : 1 bytes
  // at R = 1). Staged, the four waves of the workgroup copy the rows into LDS once and read their chunks from there.
  if (!HASPRO && !STG) load_x(0);
  WV wv[R][NCH];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = min(r0 + r, a.N - 1);  // clamped rows are computed and dropped
    const WV* wp = reinterpret_cast<const WV*>(reinterpret_cast<const char*>(a.W) + (size_t)((unsigned)row * (unsigned)(ROW_BYTES / (W8 ? (int)sizeof(WT) : 1)))) + lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c) wv[r][c] = gv_ld_nt<WV>(wp + c * 64);
  }
  // the tail arguments (one s_load) are first needed HERE, behind the burst: the row scales now, the output pointer in the epilogue
  __builtin_amdgcn_sched_barrier(0);
  if (W8 && elive) wsc = a.wscale[r0 + er];
  PTTS_KTOUCH(a.out);
  PTTS_KTOUCH(a.out_ld);
  if constexpr (STG) {
    {
      constexpr int PER_ROW = NCH * 64, TOT = MB * PER_ROW / 256;  // 16-byte vectors per row / per thread over the MB rows
      uint4 t[TOT];
#pragma unroll
      for (int i = 0; i < TOT; ++i) {
        const int idx = (int)threadIdx.x + i * 256, m = min(idx / PER_ROW, a.M - 1), k = idx % PER_ROW;
        t[i] = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.xw) + (size_t)m * a.xw_ld)[k];
      }
#pragma unroll
      for (int i = 0; i < TOT; ++i) {
        const int idx = (int)threadIdx.x + i * 256;
        *reinterpret_cast<uint4*>(s_x + (size_t)idx * 16) = t[i];  // row m of the tile starts at m * ROW_BYTES = m * PER_ROW * 16
      }
      __syncthreads();
      load_x(0);
    }
  }
  if (wave == NPW) GV_STAMP(a, 3);  // every load of this weight wave issued
  if (HASPRO) {
    __builtin_amdgcn_sched_barrier(0);  // the weight loads stay above the barrier
    if (PRO == GV_LNP) __syncthreads();  // the prologue waves' slab statistics
    __syncthreads();
    if (wave == NPW) GV_STAMP(a, 4);  // prepared row available (barrier passed)
    load_x(0);
  }
  // e4m3 weights serving several utterances: convert each weight chunk to packed bf16 ONCE and run the bf16 dot products on it
  // (the per-utterance v_cvt_pk_f32_fp8 + fma path converted every chunk MB times: at 8 utterances the e4m3 step was slower than bf16)
  constexpr bool W8B = W8 && MB > 1;
  uint4 wb[W8B ? R : 1][W8B ? NCH : 1];
  if constexpr (W8B) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) wb[r][c] = e4m3x8_to_bf16x8(wv[r][c].x, wv[r][c].y);
  }
  float v = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g > 0) {
      if (g * MG >= a.M) break;  // no live utterance in this group or the following ones
      load_x(g);
    }
#pragma unroll
    for (int m = 0; m < MG; ++m) {
      float acc[R], acc2[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if constexpr (W8B) {
            if (c & 1) acc2[r] = GvDot<bf16_t, false>::run(wb[r][c], xv[m][c], acc2[r]);
            else acc[r] = GvDot<bf16_t, false>::run(wb[r][c], xv[m][c], acc[r]);
          } else {
            if (c & 1) acc2[r] = GvDot<WT, W8>::run(wv[r][c], xv[m][c], acc2[r]);
            else acc[r] = GvDot<WT, W8>::run(wv[r][c], xv[m][c], acc[r]);
          }
        }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float t = wave_sum(acc[r] + acc2[r]);
        v = lane == (g * MG + m) * R + r ? t : v;
      }
    }
  }
  if (wave == NPW) GV_STAMP(a, 5);  // weights landed, dot products and wave reductions done
  if (elive) {
    if (W8) v *= wsc;
    float* o = a.out + (size_t)em * a.out_ld + r0 + er;
    if (EPI == GV_STORE) *o = v;
    else if (EPI == GV_RESID) *o = res_pre + v;
    else gv_store<WT>(reinterpret_cast<WT*>(a.out) + (size_t)em * a.out_ld + r0 + er, gv_gelu_erf(v));
  }
  if (wave == NPW) GV_STAMP(a, 6);  // store issued
}

// ------------------------------------------------------------------------------------------------------
// qkv_attn_kernel (single utterance, sinusoidal positions): self_attn_layer_norm + the head's q / k / v projection rows + causal
// self-attention over the KV arena + append of the new position, ONE launch per layer instead of two (LN1+QKV GEMV, attn_kernel).
// Reference: modeling_parler_tts.py:1020-1021 (LayerNorm), :848-850 (projections), :880-889 (cache update), :906-914 (attention).
//
// The step at one utterance is a chain of dependent nodes that cost ~3.6 us each whatever they stream: what a
// node saves is its kernel boundary and one global round trip. The attention of head h needs only the head's 64 q rows, so the
// projection is recomputed inside every workgroup that needs it instead of being published by a node of its own:
//   grid = nheads x (S + 3) workgroups of 8 weight waves (8 rows each, all loads of a wave in flight at once; wave 0 requests the
//   residual row + gamma / beta BEFORE its weights and normalises it while they fly: 8 waves = 2 per SIMD, 256 VGPRs each - the separate
//   LayerNorm wave of the first version made 9, i.e. 3 on one SIMD and a 168-register ceiling that excluded fp32 at H = 1024);
//   role s < S : q rows of the head (128 KB bf16 at H = 1024; plain loads: the S + 2 workgroups of a head sit on one XCD when
//                nheads % 8 == 0 and share them in its L2) -> q in LDS -> single-query attention over cache rows [0, pos) of split s
//                (first batch of K/V rows requested at kernel start, before q exists) -> unnormalised partial + (max, sumexp);
//   role S, S+1: q rows and k rows of head dimensions [0,32) / [32,64) -> half of q . k_new, K cache row of the new position;
//   role S + 2 : the 64 v rows -> V cache row, and the row as the new position's partial (weight exp2(score - max) in the combine).
// The combine (gv_attn2_wave, prologue of the out_proj node) merges the S + 1 slots. Dot products run in gemv_kernel's order (chunk
// parity accumulators, one wave reduction per row): q / k / v are bit-identical to the two-node path; only the association of the
// softmax sums differs (the new position is a slot of its own instead of a row inside a split).
// ------------------------------------------------------------------------------------------------------
template <typename WT, int NCH, bool W8, int U>
__global__ void __launch_bounds__(512) qkv_attn_kernel(QkvAttnArgs_KPARAMS) {
  QkvAttnArgs_KJOIN(a)
  constexpr int EPL = Elem<WT>::EPL, LPR = 64 / EPL, RPI = 64 / LPR, NW = 8, RW = 8;
  constexpr int NF4 = NCH * EPL / 4;
  constexpr int ROW_BYTES = NCH * 64 * 16;                          // H * sizeof(WT)
  constexpr int WROW_BYTES = ROW_BYTES / (W8 ? (int)sizeof(WT) : 1);  // one weight row
  typedef typename GvDot<WT, W8>::WV WV;
  __shared__ __attribute__((aligned(16))) char s_x[ROW_BYTES];
  __shared__ float s_r[64];
  __shared__ float s_o[NW][64];
  __shared__ float s_ml[NW][2];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave 0: the LayerNorm operands first - loads return in order, so nothing it normalises waits behind its weight burst
  float4 lv[NF4], lg[NF4], lb[NF4];
  const int b = blockIdx.z;  // utterance (round 5: 2..8 utterances run the same node, one grid slice each)
  if (w == 0) GV_STAMP(a, 0);  // entry
  if (w == 0) {
    const float* xr = a.x + (size_t)b * a.x_ld;
#pragma unroll
    for (int i = 0; i < NF4; ++i) lv[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      lg[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
      lb[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const int h = blockIdx.x, role = blockIdx.y, S = a.S;
  const int n_rep = a.nheads / a.kv_heads, kvh = h / n_rep, Hkv = a.kv_heads * 64;
  int row0;
  if (role < S) row0 = h * 64 + w * RW;
  else if (role < S + 2) row0 = (w < 4 ? h * 64 : a.H + kvh * 64) + (role - S) * 32 + (w & 3) * RW;
  else row0 = a.H + Hkv + kvh * 64 + w * RW;
  // ---- t = 0: everything this wave will ever need from memory (bar later K/V batches of a long context) goes in flight ---------
  WV wv[RW][NCH];
  {
    const char* wbase = reinterpret_cast<const char*>(a.W) + (size_t)row0 * WROW_BYTES;
    if (row0 < a.H || a.M > 1) {  // q rows: read by the other workgroups of this head as well (k / v rows too when several utterances share them)
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < NCH; ++c) wv[r][c] = (reinterpret_cast<const WV*>(wbase + (size_t)r * WROW_BYTES) + lane)[c * 64];
    } else {           // k / v rows: read once per step
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < NCH; ++c) wv[r][c] = gv_ld_nt<WV>(reinterpret_cast<const WV*>(wbase + (size_t)r * WROW_BYTES) + lane + c * 64);
    }
  }
  float wsc = 1.f;
  if (W8 && lane < RW) wsc = a.wscale[row0 + lane];
  const int P = *a.P, cl = a.cur_len[b];
  WT* Kc = reinterpret_cast<WT*>(a.kcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64;
  WT* Vc = reinterpret_cast<WT*>(a.vcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64;
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  float* part = a.part + (size_t)b * (S + 1) * a.H;              // this utterance's S + 1 slots
  float* stats = a.stats + (size_t)b * (S + 1) * a.nheads * 2;
  const uint4* Kb = reinterpret_cast<const uint4*>(Kc);
  const uint4* Vb = reinterpret_cast<const uint4*>(Vc);
  const int r = lane / LPR, c = lane % LPR;
  const int TW = S * NW, wvid = role * NW + w;
  uint4 kf[U], vf[U];
  int mk[U];
  if (role < S) {
#pragma unroll
    for (int u = 0; u < U; ++u) {  // rows at or beyond the host-known context bucket are not fetched; validity is applied when the data is used
      const int t = (wvid + u * TW) * RPI + r;
      const int tc = t < a.kv_bound ? t : 0;
      kf[u] = Kb[(size_t)tc * LPR + c];
      vf[u] = Vb[(size_t)tc * LPR + c];
      mk[u] = (mrow && t < a.mask_ld) ? mrow[t] : 1;
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the loads stay above the barrier
  if (w == 0) GV_STAMP(a, 1);  // every load issued
  if (w == 0) {  // LayerNorm of the row (gv_ln_row's arithmetic: shifted one-pass mean / variance), engine dtype into LDS
    const float c0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lv[0].x)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const float d0 = lv[i].x - c0, d1 = lv[i].y - c0, d2 = lv[i].z - c0, d3 = lv[i].w - c0;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    gv_pair_sum(s1, s2);
    const float dm = s1 * a.invK, mean = c0 + dm;
    const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NF4; ++i)
      gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, (lv[i].x - mean) * rstd * lg[i].x + lb[i].x, (lv[i].y - mean) * rstd * lg[i].y + lb[i].y,
                        (lv[i].z - mean) * rstd * lg[i].z + lb[i].z, (lv[i].w - mean) * rstd * lg[i].w + lb[i].w);
  }
  if (w == 0) GV_STAMP(a, 2);         // residual row landed, normalised, in LDS
  __syncthreads();                    // normalised row in LDS
  // ---- the wave's 8 projection rows ---------------------------------------------------------------------------------------------
  {
    uint4 xv[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) xv[cc] = *reinterpret_cast<const uint4*>(s_x + (size_t)(cc * 64 + lane) * 16);
    float acc[RW], acc2[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) { acc[rr] = 0.f; acc2[rr] = 0.f; }
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (cc & 1) acc2[rr] = GvDot<WT, W8>::run(wv[rr][cc], xv[cc], acc2[rr]);
        else acc[rr] = GvDot<WT, W8>::run(wv[rr][cc], xv[cc], acc[rr]);
      }
    float v = 0.f;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const float t = wave_sum(acc[rr] + acc2[rr]);
      v = lane == rr ? t : v;
    }
    if (lane < RW) s_r[w * RW + lane] = W8 ? v * wsc : v;
  }
  if (w == 0) GV_STAMP(a, 3);  // weight rows landed, 8 dot products + reductions done
  __syncthreads();  // the 64 projected values of this workgroup in LDS
  const int pos = P + cl - 1;  // position of the new token; cache rows [0, pos) are in place
  const float qscale = a.scale * 1.44269504088896340736f;  // softmax in base 2 (attn_kernel)
  if (role >= S) {
    if (w != 0) return;
    const bool writer = h == kvh * n_rep;  // grouped-query attention: the first query head of a group writes the shared K/V row
    if (role < S + 2) {
      const int half = role - S;
      const float kd = Elem<WT>::rnd(s_r[32 + (lane & 31)]);  // the key as the cache holds it
      const float pr = lane < 32 ? (s_r[lane] * qscale) * kd : 0.f;
      const float dot = wave_sum(pr);
      if (lane == 0) stats[((size_t)S * a.nheads + h) * 2 + half] = dot;
      if (writer && lane < 32) gv_store<WT>(Kc + (size_t)pos * 64 + half * 32 + lane, s_r[32 + lane]);
    } else {
      const float vd = s_r[lane];
      part[(size_t)S * a.H + h * 64 + lane] = Elem<WT>::rnd(vd);
      if (writer) gv_store<WT>(Vc + (size_t)pos * 64 + lane, vd);
    }
    return;
  }
  // ---- role < S: single-query attention over the cache rows of this split -----------------------------------------------------------
  float qv[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qv[e] = s_r[c * EPL + e] * qscale;
  const int L = pos;  // rows of the cache proper; the new position is the combine's slot S
  const int G = (L + RPI - 1) / RPI;
  float m_run = -INFINITY, l_run = 0.f, o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  for (int g0 = wvid; g0 < G; g0 += TW * U) {
    bool ok[U];
    if (g0 != wvid) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = (g0 + u * TW) * RPI + r;
        const int tc = t < L ? t : 0;
        kf[u] = Kb[(size_t)tc * LPR + c];
        vf[u] = Vb[(size_t)tc * LPR + c];
        mk[u] = (mrow && tc < P) ? mrow[tc] : 1;
      }
    }
    float sc[U], bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (g0 + u * TW) * RPI + r;
      ok[u] = t < L && (t >= P || mk[u] != 0);
      float kx[EPL];
      unpack16(kf[u], kx, WT());
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) d = fmaf(qv[e], kx[e], d);
      d = group_reduce<OpSum, LPR>(d);
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
      unpack16(vf[u], vx, WT());
      l_run += p;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = ok[u] ? fmaf(p, vx[e], o[e]) : o[e];  // masked rows may hold NaN/garbage V
    }
    m_run = m_new;
  }
  l_run = across_groups_reduce<OpSum, LPR>(l_run);
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = across_groups_reduce<OpSum, LPR>(o[e]);
  if (w == 0) GV_STAMP(a, 4);  // K / V rows landed, attention loop of this wave done
  if (r == 0) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_o[w][c * EPL + e] = o[e];
    if (c == 0) { s_ml[w][0] = m_run; s_ml[w][1] = l_run; }
  }
  __syncthreads();
  if (w == 0) {
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NW; ++i) M = fmaxf(M, s_ml[i][0]);
    float ov = 0.f, lv = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float wgt = (s_ml[i][0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s_ml[i][0] - M);
      ov += wgt * s_o[i][lane];
      lv += wgt * s_ml[i][1];
    }
    GV_STAMP(a, 5);  // cross-wave combine done, stores next
    part[(size_t)role * a.H + h * 64 + lane] = ov;
    if (lane == 0) {
      float* st = stats + ((size_t)role * a.nheads + h) * 2;
      st[0] = M;
      st[1] = lv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// xfold_attn_kernel (single utterance, folded cross block): encoder_attn_layer_norm + the head's 64 rows of M (base-2 scores against the
// static description) + per-head softmax + the head's 64 columns of U for a slice of the output rows, ONE launch per layer instead of
// two (LN2 + M x GEMV, softmax + U p + residual GEMV). Reference: modeling_parler_tts.py:1038-1052 (cross block), :906-914 (attention);
// M = qscale K Wq and U = Wo V^T are built at prefill (ptts_lm_kernels.h: xfold_m_kernel / xfold_u_kernel).
//
// Same reasoning as qkv_attn_kernel: the node costs its kernel boundary + global round trip, not its bytes. Head h's softmax needs only
// the head's 64 scores, so grid = nheads x J workgroups of 8 weight waves (8 rows of M each; wave 0 also normalises the row; the J workgroups of a head
// sit on one XCD when nheads % 8 == 0 and share the rows in its L2); every wave finishes the head's softmax redundantly (64 values) and
// multiplies its NUR x (64 / LPR) rows of U_h (128 / 256 contiguous bytes per row, requested at kernel start) by it. The output of the
// block is a SUM over heads: each head's contribution goes to its own partial row xpart[h][:] and the next node's LayerNorm prologue
// (GV_LNP) adds the rows in a fixed order - no atomics, bit-reproducible.
// ------------------------------------------------------------------------------------------------------
template <typename WT, int NCH, int NUR>
__global__ void __launch_bounds__(512) xfold_attn_kernel(XfoldAttnArgs_KPARAMS) {
  XfoldAttnArgs_KJOIN(a)
  constexpr int EPL = Elem<WT>::EPL, LPR = 64 / EPL, RPI = 64 / LPR, NW = 8, RW = 8;
  constexpr int NF4 = NCH * EPL / 4;
  constexpr int ROW_BYTES = NCH * 64 * 16;  // H * sizeof(WT)
  typedef typename GvDot<WT, false>::WV WV;
  __shared__ __attribute__((aligned(16))) char s_x[ROW_BYTES];
  __shared__ float s_r[64];
  __shared__ __attribute__((aligned(16))) WT s_p[NW][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // 8 waves, all of them weight waves; wave 0 requests the LayerNorm operands first and normalises while its weights fly (qkv_attn_kernel)
  float4 lv[NF4], lg[NF4], lb[NF4];
  if (w == 0) GV_STAMP(a, 0);  // entry
  if (w == 0) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) lv[i] = *reinterpret_cast<const float4*>(a.x + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      lg[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
      lb[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const int h = blockIdx.x, j = blockIdx.y;
  const int r = lane / LPR, c = lane % LPR;
  // ---- t = 0: the wave's 8 rows of M_h, its NUR x RPI row segments of U_h, the mask ---------------------------------------------
  WV wv[RW][NCH];
  {
    const char* wbase = reinterpret_cast<const char*>(a.Mw) + (size_t)(h * 64 + w * RW) * ROW_BYTES;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) wv[rr][cc] = (reinterpret_cast<const WV*>(wbase + (size_t)rr * ROW_BYTES) + lane)[cc * 64];
  }
  const int n0 = (j * NUR * NW + w) * RPI + r;  // output row of round 0; round u: + u * NW * RPI
  const size_t KU = (size_t)a.nheads * 64;      // row pitch of U in elements
  uint4 uf[NUR];
#pragma unroll
  for (int u = 0; u < NUR; ++u)
    uf[u] = ld_nt16(reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.Uw) + (size_t)(n0 + u * NW * RPI) * KU + h * 64) + c);
  const int nv = *a.n_valid;
  const int mk = a.mask ? a.mask[lane] : 1;
  __builtin_amdgcn_sched_barrier(0);  // the loads stay above the barrier
  PTTS_KTOUCH(nv);       // the dependent scalar load and the epilogue's tail arguments are fetched here, in the shadow of the burst
  PTTS_KTOUCH(a.xpart);
  if (w == 0) GV_STAMP(a, 1);  // every load issued
  if (w == 0) {  // gv_ln_row's arithmetic
    const float c0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lv[0].x)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const float d0 = lv[i].x - c0, d1 = lv[i].y - c0, d2 = lv[i].z - c0, d3 = lv[i].w - c0;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    gv_pair_sum(s1, s2);
    const float dm = s1 * a.invK, mean = c0 + dm;
    const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NF4; ++i)
      gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, (lv[i].x - mean) * rstd * lg[i].x + lb[i].x, (lv[i].y - mean) * rstd * lg[i].y + lb[i].y,
                        (lv[i].z - mean) * rstd * lg[i].z + lb[i].z, (lv[i].w - mean) * rstd * lg[i].w + lb[i].w);
  }
  if (w == 0) GV_STAMP(a, 2);         // residual row landed, normalised, in LDS
  __syncthreads();                    // normalised row in LDS
  {
    uint4 xv[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) xv[cc] = *reinterpret_cast<const uint4*>(s_x + (size_t)(cc * 64 + lane) * 16);
    float acc[RW], acc2[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) { acc[rr] = 0.f; acc2[rr] = 0.f; }
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (cc & 1) acc2[rr] = GvDot<WT, false>::run(wv[rr][cc], xv[cc], acc2[rr]);
        else acc[rr] = GvDot<WT, false>::run(wv[rr][cc], xv[cc], acc[rr]);
      }
    float v = 0.f;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const float t = wave_sum(acc[rr] + acc2[rr]);
      v = lane == rr ? t : v;
    }
    if (lane < RW) s_r[w * RW + lane] = v;
  }
  if (w == 0) GV_STAMP(a, 3);  // rows of M landed, 8 dot products + reductions done
  __syncthreads();  // the head's 64 base-2 scores in LDS
  // ---- softmax over the description positions (lane = position), masked / absent positions weigh 0 (gv_softmax_wave's rules) ------
  {
    const bool valid = lane < nv && mk != 0;
    const float sc = valid ? s_r[lane] : -INFINITY;
    const float mx = wave_max(sc);
    const float e = valid ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
    const float sm = wave_sum(e);
    const float inv = sm > 0.f ? 1.0f / sm : 0.f;
    gv_store<WT>(&s_p[w][lane], e * inv);  // the weights enter the U product in the engine dtype (the two-node path's GV_SOFTMAX prologue)
  }
  __builtin_amdgcn_wave_barrier();
  const uint4 pv = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(&s_p[w][0]) + c * 16);  // the wave's own copy: no workgroup barrier
#pragma unroll
  for (int u = 0; u < NUR; ++u) {
    float d = GvDot<WT, false>::run(uf[u], pv, 0.f);
    d = group_reduce<OpSum, LPR>(d);
    if (c == 0) a.xpart[(size_t)h * a.H + n0 + u * NW * RPI] = d;
  }
  if (w == 0) GV_STAMP(a, 5);  // softmax, U rows landed, partial row stores issued
}

// ------------------------------------------------------------------------------------------------------
// xq_attn_kernel (1..8 utterances, GEMV engines, no static fold, sinusoidal positions): encoder_attn_layer_norm + the head's 64 cross-q
// projection rows + cross-attention of ONE (head, utterance) against the static description K / V - the two nodes "LN2 + q GEMV" and
// "cross attention" of the un-folded cross block as one launch of nheads x M workgroups. Reference: modeling_parler_tts.py:1038-1047
// (LayerNorm + attention call), :855 (q projection), :906-914 (attention). Same structure and arithmetic as qkv_attn_kernel's roles s < S
// with one split: 8 weight waves of 8 rows (wave 0 normalises the row while its weights fly), the first batch of K / V row groups is
// requested at kernel start, softmax in base 2 finished in the kernel (the description is short: never split). The M workgroups of a head
// sit on one XCD (nheads % 8 == 0) and share the head's q rows in its L2.
// ------------------------------------------------------------------------------------------------------
template <typename WT, int NCH, bool W8>
__global__ void __launch_bounds__(512) xq_attn_kernel(XqAttnArgs a) {
  constexpr int EPL = Elem<WT>::EPL, LPR = 64 / EPL, RPI = 64 / LPR, U = 4, NW = 8, RW = 8;
  constexpr int NF4 = NCH * EPL / 4;
  constexpr int ROW_BYTES = NCH * 64 * 16;                          // H * sizeof(WT)
  constexpr int WROW_BYTES = ROW_BYTES / (W8 ? (int)sizeof(WT) : 1);  // one weight row
  typedef typename GvDot<WT, W8>::WV WV;
  __shared__ __attribute__((aligned(16))) char s_x[ROW_BYTES];
  __shared__ float s_r[64];
  __shared__ float s_o[NW][64];
  __shared__ float s_ml[NW][2];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = blockIdx.x, b = blockIdx.y;
  float4 lv[NF4], lg[NF4], lb[NF4];
  if (w == 0) {
    const float* xr = a.x + (size_t)b * a.x_ld;
#pragma unroll
    for (int i = 0; i < NF4; ++i) lv[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      lg[i] = *reinterpret_cast<const float4*>(a.gamma + (lane + 64 * i) * 4);
      lb[i] = *reinterpret_cast<const float4*>(a.beta + (lane + 64 * i) * 4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const int row0 = h * 64 + w * RW;
  WV wv[RW][NCH];
  {
    const char* wbase = reinterpret_cast<const char*>(a.W) + (size_t)row0 * WROW_BYTES;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) wv[r][c] = (reinterpret_cast<const WV*>(wbase + (size_t)r * WROW_BYTES) + lane)[c * 64];
  }
  float wsc = 1.f;
  if (W8 && lane < RW) wsc = a.wscale[row0 + lane];
  const int L = *a.n_valid;
  const int n_rep = a.nheads / a.kv_heads, kvh = h / n_rep;
  const uint4* Kb = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.kcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64);
  const uint4* Vb = reinterpret_cast<const uint4*>(reinterpret_cast<const WT*>(a.vcache) + ((size_t)b * a.kv_heads + kvh) * a.cap * 64);
  const int* mrow = a.mask ? a.mask + (size_t)b * a.mask_ld : nullptr;
  const int r = lane / LPR, c = lane % LPR;
  uint4 kf[U], vf[U];
  int mk[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // positions beyond the arena are not fetched; validity against the length is applied when the data is used
    const int t = (w + u * NW) * RPI + r;
    const int tc = t < a.cap ? t : 0;
    kf[u] = Kb[(size_t)tc * LPR + c];
    vf[u] = Vb[(size_t)tc * LPR + c];
    mk[u] = (mrow && t < a.mask_ld) ? mrow[t] : 1;
  }
  __builtin_amdgcn_sched_barrier(0);  // the loads stay above the barrier
  if (w == 0) {  // gv_ln_row's arithmetic
    const float c0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lv[0].x)));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const float d0 = lv[i].x - c0, d1 = lv[i].y - c0, d2 = lv[i].z - c0, d3 = lv[i].w - c0;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    gv_pair_sum(s1, s2);
    const float dm = s1 * a.invK, mean = c0 + dm;
    const float rstd = rsqrtf(fmaxf(s2 * a.invK - dm * dm, 0.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NF4; ++i)
      gv_lds_store4<WT>(s_x, (lane + 64 * i) * 4, (lv[i].x - mean) * rstd * lg[i].x + lb[i].x, (lv[i].y - mean) * rstd * lg[i].y + lb[i].y,
                        (lv[i].z - mean) * rstd * lg[i].z + lb[i].z, (lv[i].w - mean) * rstd * lg[i].w + lb[i].w);
  }
  __syncthreads();  // normalised row in LDS
  {
    uint4 xv[NCH];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) xv[cc] = *reinterpret_cast<const uint4*>(s_x + (size_t)(cc * 64 + lane) * 16);
    float acc[RW], acc2[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) { acc[rr] = 0.f; acc2[rr] = 0.f; }
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (cc & 1) acc2[rr] = GvDot<WT, W8>::run(wv[rr][cc], xv[cc], acc2[rr]);
        else acc[rr] = GvDot<WT, W8>::run(wv[rr][cc], xv[cc], acc[rr]);
      }
    float v = 0.f;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const float t = wave_sum(acc[rr] + acc2[rr]);
      v = lane == rr ? t : v;
    }
    if (lane < RW) s_r[w * RW + lane] = W8 ? v * wsc : v;
  }
  __syncthreads();  // the head's 64 q values in LDS
  const float qscale = a.scale * 1.44269504088896340736f;  // softmax in base 2 (attn_kernel)
  float qv[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) qv[e] = s_r[c * EPL + e] * qscale;
  const int G = (L + RPI - 1) / RPI;
  float m_run = -INFINITY, l_run = 0.f, o[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = 0.f;
  for (int g0 = w; g0 < G; g0 += NW * U) {
    bool ok[U];
    if (g0 != w) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = (g0 + u * NW) * RPI + r;
        const int tc = t < L ? t : 0;
        kf[u] = Kb[(size_t)tc * LPR + c];
        vf[u] = Vb[(size_t)tc * LPR + c];
        mk[u] = mrow ? mrow[tc] : 1;
      }
    }
    float sc[U], bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (g0 + u * NW) * RPI + r;
      ok[u] = t < L && mk[u] != 0;  // cross-attention: the padding mask covers every description position
      float kx[EPL];
      unpack16(kf[u], kx, WT());
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) d = fmaf(qv[e], kx[e], d);
      d = group_reduce<OpSum, LPR>(d);
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
      unpack16(vf[u], vx, WT());
      l_run += p;
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = ok[u] ? fmaf(p, vx[e], o[e]) : o[e];  // masked rows may hold NaN/garbage V
    }
    m_run = m_new;
  }
  l_run = across_groups_reduce<OpSum, LPR>(l_run);
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = across_groups_reduce<OpSum, LPR>(o[e]);
  if (r == 0) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_o[w][c * EPL + e] = o[e];
    if (c == 0) { s_ml[w][0] = m_run; s_ml[w][1] = l_run; }
  }
  __syncthreads();
  if (w == 0) {
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NW; ++i) M = fmaxf(M, s_ml[i][0]);
    float ov = 0.f, lsum = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float wgt = (s_ml[i][0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s_ml[i][0] - M);
      ov += wgt * s_o[i][lane];
      lsum += wgt * s_ml[i][1];
    }
    gv_store<WT>(reinterpret_cast<WT*>(a.out) + (size_t)b * a.out_ld + h * 64 + lane, lsum > 0.f ? ov / lsum : 0.f);  // every key masked: 0 (attn_kernel)
  }
}
