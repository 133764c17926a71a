// DAC engine behind include/ptts.h. Replaces DACModel.decode (dac_wrapper/modeling_dac.py:106-142): quantizer.from_codes
// (:138) + the descript-audio-codec decoder (:139); and DACModel.encode (:33-104, voice prompts): model.preprocess (:64,
// done by the caller: right-pad to a hop multiple) + model.encode (:95) = encoder stack + residual VQ search.
//
// Design:
//   * activations are channels-last fp32 [b][t][c]: a B-operand fragment (16 consecutive channels of one frame and
//     tap) is 4 x 16 B per lane group, an output fragment (4 consecutive channels of one frame) is one float4 store;
//   * every Conv1d / ConvTranspose1d is an implicit GEMM  out[co][t] = sum_{tap,ci} W[co][tap][ci] * x[t + off(tap)][ci]
//     on v_mfma_f32_16x16x4_f32 (exact f32 = an fmaf chain, so the fp32 parity bar RMS <= 1e-4 holds by construction);
//     a stride-s transposed conv is s interleaved 2-tap phase convolutions;
//   * Snake is evaluated ONCE per element in the producer's epilogue (x + sin^2(ax)/(a+1e-9)), never in a consumer
//     prologue (7 taps x Cout/16 strips would recompute each sin dozens of times); bias, residual skip and the
//     next layer's Snake are fused into the same epilogue; weight-norm is folded by the caller at load;
//   * RVQ from_codes is a gather-sum over K precomputed [codebook_size][latent] tables (out_proj(codebook_i)+bias_i);
//   * encode: the encoder's stride-s Conv1d(k = 2s) is the same implicit GEMM with input frame t*s + tap - pad; the
//     residual VQ runs one workgroup per frame through all stages (in_proj, cosine nearest neighbour over the
//     L2-normalised codebook, straight-through expression p + (c - p), out_proj, residual update).
#include <map>
#include <set>
#include <string>
#include <vector>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "ptts_common.h"

namespace {

constexpr int MAXTAPS = 8;

struct ConvArgs {
  const void* x;       // activated input, channels-last [B][Tin][Cin]: fp32, or bf16 in the bf16-operand mode
  const void* Wp;      // packed [phase][Cout/16][ntaps*Cin/16][64][4] fp32, or [..][ntaps*Cin/32][64][8] bf16
  const float* bias;   // [Cout]
  const float* skip;   // optional residual [B][Tout][Cout] (may alias out_raw)
  float* out_raw;      // optional
  void* out_act;       // optional: snake(alpha) of the result (bf16 in the bf16-operand mode unless act_f32)
  int act_f32;         // bf16 mode: write out_act as fp32 (the layer feeding the final Conv1d(C -> 1) + tanh)
  const float* alpha;  // [Cout] for out_act
  int dil, pad, transposed;  // tap offset: conv  tap*dil - pad ; transposed (stride = nphase)  (ph + pad)/nphase - tap
  int B, Tin, Cin, Cout, ntaps, nphase;
  int stride, Tn;      // input stride of a down-sampling conv (else 1); output frames per phase (Tin unless strided)
  const int* lens;     // ragged decode (ptts_dac_decode_ragged): latent frames per utterance [B] on the device, or null. Utterance b then has
  int len_mul;         // lens[b] * len_mul valid input rows (= output rows per phase): rows beyond read as the zero padding, tiles beyond exit
  int epi_direct;      // conv_lds_kernel A/B (PTTS_DAC_EPI_DIRECT=1): the round-3 epilogue (a lane stores 4 channels of one frame)
};

// valid input rows of utterance b (buffers keep the full stride a.Tin)
__device__ __forceinline__ int valid_rows(const ConvArgs& a, int b) { return a.lens ? min(a.Tin, max(a.lens[b], 0) * a.len_mul) : a.Tin; }  // lengths clamped to [0, T]

// FAST (bf16-operand mode, whose activations are rounded to bf16 anyway): v_sin_f32 on a*x / 2pi instead of the ~100-instruction
// exact sinf - the Snake epilogue of the 42 M-element layers was ~100 us of VALU per layer (profiles/r02_dac_layers.txt).
// inv = 1.0f / (al + 1e-9f), computed ONCE per channel at load time (inv_alpha_kernel: the same correctly rounded fp32 division the
// epilogues evaluated per element - ~10 VALU instructions beside a sin; the Snake of the fused units was 7.3 ms of a 69 ms batch-32 decode,
// profiles/r04_experiments.txt call 8). Stored right behind the alphas: alpha[C + c].
template <bool FAST>
__device__ __forceinline__ float snake_f(float x, float al, float inv) {
  float s;
#ifdef PTTS_DAC_FAST_SIN
  s = __sinf(al * x);
#else
  if constexpr (FAST) s = __sinf(al * x);
  else s = sinf(al * x);
#endif
  return x + inv * (s * s);
}
__device__ __forceinline__ float4 ld_inv4(const float* alpha, int C, int c) { return *reinterpret_cast<const float4*>(alpha + C + c); }
__global__ void inv_alpha_kernel(float* __restrict__ alpha, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) alpha[C + i] = 1.0f / (alpha[i] + 1e-9f);
}

// workgroup: 4 waves = 4 consecutive 32-frame tiles; every wave owns the SAME CS 16-channel output strips (CS = 6 or 8
// divides every decoder width / 16: 96, 48, 24, 12, 6 strips), so each activation fragment is reused CS times from
// registers and the 4 waves read identical weight fragments (L1 broadcast). First mapping (waves over strips, one
// shared tile) left 25 % of the waves idle on the 6/12/24-strip layers and ran at 43 % of the f32-MFMA peak.
// BF: bf16 MFMA operands (v_mfma_f32_16x16x32_bf16: 32 channels per step, activations and weights stored in bf16), fp32
// accumulate, bias / skip / Snake in fp32 - at least the precision of the reference run in bf16, where every conv output
// is rounded to bf16. BF = false is the exact-f32 parity mode.
template <int CS, bool BF>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int fpb = 32 * (int)(blockDim.x >> 6);  // frames per workgroup: 32 per wave, 1/2/4 waves (host picks for balance)
  const int ntile = (a.Tn + fpb - 1) / fpb;
  const int tile = blockIdx.x % ntile, ph = (blockIdx.x / ntile) % a.nphase, b = blockIdx.x / (ntile * a.nphase);
  const int strip0 = blockIdx.y * CS;
  const int nstrips = a.Cout / 16;
  const int Tv = valid_rows(a, b), Tnv = a.lens ? Tv : a.Tn;  // ragged decode: this utterance's rows
  if (tile * fpb + wave * 32 >= Tnv) return;
  constexpr int KC = BF ? 32 : 16;      // channels per k-step
  const int cpt = a.Cin / KC;           // k-steps per tap
  const int nk = a.ntaps * cpt;
  const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)b * a.Tin * a.Cin * (BF ? 2 : 4);
  const float4* Wp = reinterpret_cast<const float4*>(a.Wp) + ((size_t)ph * nstrips * nk) * 64 + lane;  // 16 B per lane either way
  const int j0 = tile * fpb + wave * 32;

  f32x4 acc[CS][2];
#pragma unroll
  for (int s = 0; s < CS; ++s) { acc[s][0] = f32x4{0, 0, 0, 0}; acc[s][1] = f32x4{0, 0, 0, 0}; }

  // software-pipelined k loop over (tap, 16-channel chunk): the fragments of step ks+1 are in flight while the
  // 8*CS MFMAs of step ks issue (measured: neutral, 65.5 -> 66.9 TFLOP/s -- L2 latency was already hidden by occupancy)
  const int nks = nk;
  // (a macro, not a lambda: array-by-reference parameters kept the CS = 8 fragment arrays in scratch)
#define PTTS_DAC_LOAD_STEP(KS, WF, B0, B1)                                                                              \
  do {                                                                                                                  \
    const int tap_ = (KS) / cpt, cc_ = (KS) - tap_ * cpt;                                                                \
    const int off_ = a.transposed ? (ph + a.pad) / a.nphase - tap_ : tap_ * a.dil - a.pad;                               \
    const int ti0_ = (j0 + j) * a.stride + off_, ti1_ = ti0_ + 16 * a.stride;                                            \
    B0 = make_float4(0, 0, 0, 0);                                                                                        \
    B1 = make_float4(0, 0, 0, 0);                                                                                        \
    if (ti0_ >= 0 && ti0_ < Tv) B0 = *reinterpret_cast<const float4*>(xb + ((size_t)ti0_ * a.Cin + cc_ * KC) * (BF ? 2 : 4) + q * 16);  \
    if (ti1_ >= 0 && ti1_ < Tv) B1 = *reinterpret_cast<const float4*>(xb + ((size_t)ti1_ * a.Cin + cc_ * KC) * (BF ? 2 : 4) + q * 16);  \
    _Pragma("unroll") for (int s_ = 0; s_ < CS; ++s_) WF[s_] = Wp[((size_t)(strip0 + s_) * nk + (KS)) * 64];             \
  } while (0)
  float4 wf[CS], b0, b1;
  PTTS_DAC_LOAD_STEP(0, wf, b0, b1);
  for (int ks = 0; ks < nks; ++ks) {
    float4 wn[CS], n0, n1;
    if (ks + 1 < nks) PTTS_DAC_LOAD_STEP(ks + 1, wn, n0, n1);
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      if constexpr (BF) {
        acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, b0), acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, b1), acc[s][1], 0, 0, 0);
      } else {
        acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].x, b0.x, acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].x, b1.x, acc[s][1], 0, 0, 0);
        acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].y, b0.y, acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].y, b1.y, acc[s][1], 0, 0, 0);
        acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].z, b0.z, acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].z, b1.z, acc[s][1], 0, 0, 0);
        acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].w, b0.w, acc[s][0], 0, 0, 0);
        acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s].w, b1.w, acc[s][1], 0, 0, 0);
      }
    }
    if (ks + 1 < nks) {
#pragma unroll
      for (int s = 0; s < CS; ++s) wf[s] = wn[s];
      b0 = n0;
      b1 = n1;
    }
  }
  // epilogue: D[row = co_local = q*4 + r][col = frame j]. A helper called with compile-time (s, half) keeps `acc`
  // statically indexed: the big inlined snake bodies otherwise stop the unroller at CS = 8 and push acc to scratch.
  const int Tout = a.Tn * a.nphase;
  auto emit = [&](const f32x4& av, int s, int half) {
    const int jj = j0 + half * 16 + j;
    if (jj >= Tnv) return;
    const int co = (strip0 + s) * 16 + q * 4;
    const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
    const size_t o = ((size_t)b * Tout + (size_t)jj * a.nphase + ph) * a.Cout + co;
    float4 v = make_float4(av[0] + bs.x, av[1] + bs.y, av[2] + bs.z, av[3] + bs.w);
    if (a.skip) {
      const float4 sk = *reinterpret_cast<const float4*>(a.skip + o);
      v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
    }
    if (a.out_raw) *reinterpret_cast<float4*>(a.out_raw + o) = v;
    if (a.out_act) {
      const float4 al = *reinterpret_cast<const float4*>(a.alpha + co), ia = ld_inv4(a.alpha, a.Cout, co);
      const float4 sv = make_float4(snake_f<BF>(v.x, al.x, ia.x), snake_f<BF>(v.y, al.y, ia.y), snake_f<BF>(v.z, al.z, ia.z), snake_f<BF>(v.w, al.w, ia.w));
      if (BF && !a.act_f32) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out_act) + o) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out_act) + o) = sv;
    }
  };
  if (CS >= 1) { emit(acc[0][0], 0, 0); emit(acc[0][1], 0, 1); }
  if (CS >= 2) { emit(acc[1 % CS][0], 1, 0); emit(acc[1 % CS][1], 1, 1); }
  if (CS >= 4) { emit(acc[2 % CS][0], 2, 0); emit(acc[2 % CS][1], 2, 1); emit(acc[3 % CS][0], 3, 0); emit(acc[3 % CS][1], 3, 1); }
  if (CS >= 6) { emit(acc[4 % CS][0], 4, 0); emit(acc[4 % CS][1], 4, 1); emit(acc[5 % CS][0], 5, 0); emit(acc[5 % CS][1], 5, 1); }
  if (CS >= 8) { emit(acc[6 % CS][0], 6, 0); emit(acc[6 % CS][1], 6, 1); emit(acc[7 % CS][0], 7, 0); emit(acc[7 % CS][1], 7, 1); }
}

// LDS-tiled variant of the bf16-operand kernel for the stride-1 layers of the 44.1 kHz stack (k7 dilated convs, k1 convs,
// the 2-tap phases of the transposed convs): the kernel above fetches every activation fragment from L1/L2 once per TAP and
// every weight fragment once per WAVE (10 KB per wave per 256 MFMA cycles = 160 B/clk/CU asked of a 64 B/clk L1), and ran at
// 10 % of the bf16 MFMA peak. Here a workgroup owns 128 consecutive frames x NW*CSW 16-channel strips:
//   * the activation slab of one KCH-channel chunk ([128 + halo] rows x KCH bf16) is staged ONCE into LDS and read by all
//     taps (7x fewer global reads) and all waves; rows are RS = KCH*2 + 32 bytes apart: RS/32 is odd, so the 16 lanes of each
//     ds_read_b128 service group land on 16 distinct 16-byte bank slots for ANY row base (the tap offset tap*dil is arbitrary);
//   * waves split the output strips (CSW each) and share the frames, so every wave's weight fragments are its own: CSW KB
//     per wave per k-step from L2/L1 (24 B/clk/CU at CSW = 3), prefetched one k-step ahead in registers, also across chunks;
//   * two LDS buffers: chunk c+1 is fetched into registers while chunk c computes and committed before the ONE barrier of
//     the chunk; B fragments are read in two halves of 4 frame tiles, each half in flight under the other half's 12 MFMAs.
// Per k-step and wave: 24 MFMAs 16x16x32 (384 cycles) against 8 ds_read_b128 + 3 global 16-B loads.
// FT (round 5): 16-frame tiles per workgroup - 8 (128 frames), or 4 where 128-frame tiles leave the launch with fewer than two workgroups per CU (the
// first block of a single utterance: 56-224 workgroups on 256 CUs; the short windows of the streamer). Same k order per output: bit-identical.
template <int CSW, int NW, int KS, int MAXHALO, int FT = 8>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_lds_kernel(ConvArgs a) {
  constexpr int TF = FT * 16, HF = FT / 2;
  static_assert(FT == 8 || FT == 4, "frame tiles per workgroup");
  constexpr int KCH = 32 * KS;
  constexpr int RS = KS * 64 + 32;
  constexpr int SL = KS * 4;  // 16-byte slots per slab row
  constexpr int NT = NW * 64;
  constexpr int MAXROWS = TF + MAXHALO;
  constexpr int NST = (MAXROWS * SL + NT - 1) / NT;  // staged 16-byte pieces per thread and chunk
  __shared__ __attribute__((aligned(16))) unsigned char slab[2][MAXROWS * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int ntile = (a.Tn + TF - 1) / TF;
  // (round 6, profiles/r06_dac_up_order_ab.txt: the phases of a transposed conv as neighbours in ONE XCD's queue - sharing its L2 copy of the input tile -
  //  or the phase as the slowest index - one phase's weights L2-resident at a time - measured equal to this order, 54.2-54.7 vs 53.8-54.2 ms per batch-32
  //  decode, bit-identical: the re-fetched bytes the PMC table shows for these layers do not cost time. Not kept.)
  const int tile = blockIdx.x % ntile, ph = (blockIdx.x / ntile) % a.nphase, b = blockIdx.x / (ntile * a.nphase);
  const int nstrips = a.Cout / 16;
  const int strip0 = (blockIdx.y * NW + wave) * CSW;
  const int cpt = a.Cin / 32, nk = a.ntaps * cpt;
  const int nchunk = a.Cin / KCH;
  const int NS = a.ntaps * KS;  // k-steps per chunk
  const int t0 = tile * TF;
  const int Tv = valid_rows(a, b), Tnv = a.lens ? Tv : a.Tn;  // ragged decode: this utterance's rows (workgroup-uniform)
  if (t0 >= Tnv) return;
  // slab row r holds input frame t0 + o0 + r; tap `tap` of output frame t0 + f reads row f + rel(tap)
  const int o0 = a.transposed ? (ph + a.pad) / a.nphase - (a.ntaps - 1) : -a.pad;
  const int halo = a.transposed ? a.ntaps - 1 : (a.ntaps - 1) * a.dil;
  const int nslot = (TF + halo) * SL;
  const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)b * a.Tin * a.Cin * 2;
  const float4* Wp = reinterpret_cast<const float4*>(a.Wp) + ((size_t)ph * nstrips + strip0) * nk * 64 + lane;
  const int lrow = j * RS + q * 16;

  f32x4 acc[CSW][FT];
#pragma unroll
  for (int s = 0; s < CSW; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f) acc[s][f] = f32x4{0, 0, 0, 0};

  uint4 stg[NST];
#define PTTS_SLAB_FETCH(C)                                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < NST; ++i_) {                                                                    \
    const int idx_ = tid + i_ * NT, r_ = idx_ / SL, sl_ = idx_ - r_ * SL, ti_ = t0 + o0 + r_;                              \
    stg[i_] = make_uint4(0, 0, 0, 0);                                                                                    \
    if (idx_ < nslot && ti_ >= 0 && ti_ < Tv)                                                                             \
      stg[i_] = *reinterpret_cast<const uint4*>(xb + ((size_t)ti_ * a.Cin + (size_t)(C) * KCH) * 2 + sl_ * 16);            \
  }
#define PTTS_SLAB_COMMIT(BUF)                                                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < NST; ++i_) {                                                                    \
    const int idx_ = tid + i_ * NT, r_ = idx_ / SL, sl_ = idx_ - r_ * SL;                                                  \
    if (idx_ < nslot) *reinterpret_cast<uint4*>(&slab[BUF][r_ * RS + sl_ * 16]) = stg[i_];                                 \
  }
  // weight fragments of k-step (chunk C, step S): ks = tap * cpt + C * KS + kk
#define PTTS_W_FETCH(WF, C, S)                                                                                          \
  do {                                                                                                                  \
    const int tap_ = (S) / KS, kk_ = (S) - tap_ * KS;                                                                     \
    const float4* wp_ = Wp + (size_t)(tap_ * cpt + (C) * KS + kk_) * 64;                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < CSW; ++s_) WF[s_] = wp_[(size_t)s_ * nk * 64];                               \
  } while (0)
  // B fragments (4 frame tiles from F0) of step S out of buffer SB
#define PTTS_B_FETCH(BV, SB, S, F0)                                                                                     \
  do {                                                                                                                  \
    const int tap_ = (S) / KS, kk_ = (S) - tap_ * KS;                                                                     \
    const int rel_ = a.transposed ? (a.ntaps - 1 - tap_) : tap_ * a.dil;                                                  \
    const unsigned char* sp_ = (SB) + (rel_ + (F0) * 16) * RS + kk_ * 64 + lrow;                                          \
    _Pragma("unroll") for (int f_ = 0; f_ < HF; ++f_) BV[f_] = *reinterpret_cast<const uint4*>(sp_ + f_ * 16 * RS);       \
  } while (0)

  // One k-step: MFMAs of (chunk c, step st) with the weight fragments WC while WN receives those of the NEXT k-step (possibly the
  // first of chunk c + 1). The register sets rotate (never copied: a copy makes the compiler wait for the loads it has just issued). The last step of a chunk commits the staged slab and holds the chunk's barrier.
#define PTTS_MFMA_HALF(WC, BV, F0)                                                                                      \
  _Pragma("unroll") for (int f_ = 0; f_ < HF; ++f_) _Pragma("unroll") for (int s_ = 0; s_ < CSW; ++s_)                     \
    acc[s_][(F0) + f_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WC[s_]), __builtin_bit_cast(bf16x8, BV[f_]), acc[s_][(F0) + f_], 0, 0, 0);
#define PTTS_STEP(WC, WN)                                                                                               \
  {                                                                                                                     \
    const unsigned char* sb_ = slab[c & 1];                                                                               \
    const bool more_ = c + 1 < nchunk, lastst_ = st + 1 == NS;                                                            \
    const int gn_ = min(gi + 3, total - 1); /* weights of k-step + 3 (resunit_lds_kernel: WD). Unconditional (a valid */    \
    const int gc_ = gn_ / NS;               /* re-fetch at the very end): a branch here would merge into a conservative */ \
    if (st == 0 && more_) { PTTS_SLAB_FETCH(c + 1); } /* vmcnt on the MFMAs below */                                      \
    PTTS_W_FETCH(WN, gc_, gn_ - gc_ * NS);                                                                                \
    PTTS_B_FETCH(bB, sb_, st, HF);                                                                                        \
    PTTS_MFMA_HALF(WC, bA, 0)                                                                                             \
    if (!lastst_) PTTS_B_FETCH(bA, sb_, st + 1, 0);                                                                       \
    PTTS_MFMA_HALF(WC, bB, HF)                                                                                            \
    if (lastst_) {                                                                                                        \
      if (more_) { PTTS_SLAB_COMMIT((c + 1) & 1); }                                                                       \
      __syncthreads();                                                                                                    \
      if (more_) PTTS_B_FETCH(bA, slab[(c + 1) & 1], 0, 0);                                                               \
      ++c;                                                                                                                \
      st = 0;                                                                                                             \
    } else {                                                                                                              \
      ++st;                                                                                                               \
    }                                                                                                                     \
    ++gi;                                                                                                                 \
  }
  const int total = nchunk * NS;
  float4 w0[CSW], w1[CSW], w2[CSW], w3[CSW];  // four register sets: k-step g computes out of w[g % 4] while the fragments of k-step g + 3 land
  uint4 bA[HF], bB[HF];
  PTTS_W_FETCH(w0, 0, 0);
  { const int g1_ = min(1, total - 1), c1_ = g1_ / NS; PTTS_W_FETCH(w1, c1_, g1_ - c1_ * NS); }
  { const int g2_ = min(2, total - 1), c2_ = g2_ / NS; PTTS_W_FETCH(w2, c2_, g2_ - c2_ * NS); }
  PTTS_SLAB_FETCH(0);
  PTTS_SLAB_COMMIT(0);
  __syncthreads();
  PTTS_B_FETCH(bA, slab[0], 0, 0);
  int c = 0, st = 0, gi = 0;
  for (int g = 0; g < total; g += 4) {
    PTTS_STEP(w0, w3)
    if (g + 1 < total) PTTS_STEP(w1, w0)
    if (g + 2 < total) PTTS_STEP(w2, w1)
    if (g + 3 < total) PTTS_STEP(w3, w2)
  }
#undef PTTS_STEP
#undef PTTS_MFMA_HALF
#undef PTTS_SLAB_FETCH
#undef PTTS_SLAB_COMMIT
#undef PTTS_W_FETCH
#undef PTTS_B_FETCH
  const int Tout = a.Tn * a.nphase;
  // epilogue through LDS (default; as resunit_lds_kernel's): the workgroup's output tile [128 frames][NW * 48 channels] goes through the slab
  // memory in passes of EF frames as fp32 rows, and every lane then moves 16 CONSECUTIVE bytes of a row (a row of the tile = NW * 192 bytes of
  // an output row; the rows of a transposed conv's phase are `nphase` rows apart). The direct form below stores 64-byte pieces per lane group:
  // the transposed convs ran 45-55 % parked at 25 % MFMA busy (profiles/r04_pmc_dac_sq.txt). Same fp32 operations in the same order.
  if (!a.epi_direct) {
    constexpr int CW = NW * CSW * 16, RSE = CW * 4 + 16, SLB = 2 * MAXROWS * RS, VPR = CW / 4;
    constexpr int EF0 = SLB / RSE;
    constexpr int EF1 = EF0 >= 128 ? 128 : (EF0 >= 64 ? 64 : (EF0 >= 32 ? 32 : 16)), EF = EF1 > TF ? TF : EF1, TPP = EF / 16;
    static_assert(EF0 >= 16, "one 16-frame pass of the output tile fits the slab memory");
    unsigned char* et = &slab[0][0];  // the last chunk's barrier has retired every slab read
    const int c0 = blockIdx.y * CW;
    // Round 5 (as resunit_lds_kernel's epilogue): the Snake parameters of this workgroup's CW channels go through LDS, and a pass over EF whole
    // rows is ONE straight-line block per (residual?, stream?, activation format) - no load from global memory and no branch between the stores,
    // so no s_waitcnt vmcnt(0) (= "until my last stores are acknowledged") in front of every row piece.
    constexpr int NPT = EF * VPR / NT;
    static_assert(EF * VPR % NT == 0, "row pieces per thread and pass");
    int tid_e = tid;  // an opaque copy for the epilogue's address arithmetic: computed from `tid` it is hoisted above the MFMA loop and spilled there
    asm volatile("" : "+v"(tid_e));
    static_assert(EF * RSE + 2 * CW * 4 <= SLB, "LDS: one pass of the output tile + the Snake parameters");
    float* s_al = reinterpret_cast<float*>(et + EF * RSE);  // [2][CW]: alpha | 1 / (alpha + 1e-9)
    if (a.out_act)
      for (int i = tid; i < 2 * VPR; i += NT) {
        const int hf = i / VPR, c4 = i - hf * VPR;
        reinterpret_cast<float4*>(s_al)[i] = *reinterpret_cast<const float4*>(a.alpha + (size_t)hf * a.Cout + c0 + c4 * 4);
      }
    typedef std::integral_constant<int, 0> I0_;
    typedef std::integral_constant<int, 1> I1_;
    typedef std::integral_constant<int, 2> I2_;
    auto pass = [&](auto skip_c, auto raw_c, auto act_c, const int r0, const int rows) __attribute__((always_inline)) {
      constexpr bool SK = decltype(skip_c)::value != 0, RW = decltype(raw_c)::value != 0;
      constexpr int AC = decltype(act_c)::value;  // 0: no activation output, 1: bf16, 2: fp32
      auto offs = [&](const int i) { const int rr = i / VPR, cv = i - rr * VPR; return ((size_t)b * Tout + (size_t)(r0 + rr) * a.nphase + ph) * a.Cout + c0 + cv * 4; };
      auto piece = [&](const int i, const size_t o, const float4 sk) __attribute__((always_inline)) {
        const int rr = i / VPR, cv = i - rr * VPR;
        float4 v = *reinterpret_cast<const float4*>(et + rr * RSE + cv * 16);
        if constexpr (SK) { v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w; }
        if constexpr (RW) *reinterpret_cast<float4*>(a.out_raw + o) = v;
        if constexpr (AC != 0) {
          const float4 al = *reinterpret_cast<const float4*>(s_al + cv * 4), ia = *reinterpret_cast<const float4*>(s_al + CW + cv * 4);
          const float4 sv = make_float4(snake_f<true>(v.x, al.x, ia.x), snake_f<true>(v.y, al.y, ia.y), snake_f<true>(v.z, al.z, ia.z), snake_f<true>(v.w, al.w, ia.w));
          if constexpr (AC == 1) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out_act) + o) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
          else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out_act) + o) = sv;
        }
      };
      if (rows == EF) {
        constexpr int GP = NPT > 6 ? 6 : NPT;  // residual values in flight per group (registers: acc is still live for the next pass)
        static_assert(NPT % GP == 0, "row pieces per group");
#pragma unroll
        for (int k0 = 0; k0 < NPT; k0 += GP) {
          float4 skp[SK ? GP : 1];
          if constexpr (SK) {
#pragma unroll
            for (int k = 0; k < GP; ++k) skp[k] = *reinterpret_cast<const float4*>(a.skip + offs(tid_e + (k0 + k) * NT));
          }
#pragma unroll
          for (int k = 0; k < GP; ++k) piece(tid_e + (k0 + k) * NT, offs(tid_e + (k0 + k) * NT), SK ? skp[SK ? k : 0] : make_float4(0.f, 0.f, 0.f, 0.f));
        }
      } else {
#pragma unroll 2
        for (int k = 0; k < NPT; ++k) {
          const int i = tid_e + k * NT;
          if (i / VPR < rows) {
            const size_t o = offs(i);
            float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (SK) sk = *reinterpret_cast<const float4*>(a.skip + o);
            piece(i, o, sk);
          }
        }
      }
    };
    auto by_act = [&](auto skip_c, auto raw_c, const int r0, const int rows) __attribute__((always_inline)) {
      if (!a.out_act) pass(skip_c, raw_c, I0_(), r0, rows);
      else if (a.act_f32) pass(skip_c, raw_c, I2_(), r0, rows);
      else pass(skip_c, raw_c, I1_(), r0, rows);
    };
    // (explicitly instantiated per pass: inside a `#pragma unroll` loop the dispatch below stopped the unroller and `acc` went to scratch)
    auto do_pass = [&](auto p0_c) __attribute__((always_inline)) {
      constexpr int p0 = decltype(p0_c)::value;
      if constexpr (p0 < FT) {
#pragma unroll
      for (int s = 0; s < CSW; ++s) {
        const float4 bs = *reinterpret_cast<const float4*>(a.bias + (strip0 + s) * 16 + q * 4);
#pragma unroll
        for (int f = 0; f < TPP; ++f) {
          const f32x4 av = acc[s][p0 + f];
          *reinterpret_cast<float4*>(et + (f * 16 + j) * RSE + ((wave * CSW + s) * 16 + q * 4) * 4) = make_float4(av[0] + bs.x, av[1] + bs.y, av[2] + bs.z, av[3] + bs.w);
        }
      }
      __syncthreads();
      const int r0 = t0 + p0 * 16, rows = min(EF, Tnv - r0);
      if (a.skip) { if (a.out_raw) by_act(I1_(), I1_(), r0, rows); else by_act(I1_(), I0_(), r0, rows); }
      else { if (a.out_raw) by_act(I0_(), I1_(), r0, rows); else by_act(I0_(), I0_(), r0, rows); }
      if (p0 + TPP < FT) __syncthreads();  // the tile memory is rewritten by the next pass
      }
    };
    do_pass(std::integral_constant<int, 0>()); do_pass(std::integral_constant<int, TPP>()); do_pass(std::integral_constant<int, 2 * TPP>());
    do_pass(std::integral_constant<int, 3 * TPP>()); do_pass(std::integral_constant<int, 4 * TPP>()); do_pass(std::integral_constant<int, 5 * TPP>());
    do_pass(std::integral_constant<int, 6 * TPP>()); do_pass(std::integral_constant<int, 7 * TPP>());
    return;
  }
  // direct epilogue: as conv_mfma_kernel (D[row = co_local = q*4 + r][col = frame j]); explicit (s, f) calls keep `acc` statically indexed
  auto emit = [&](const f32x4 av, const int s, const int f) {
    const int jj = t0 + f * 16 + j;
    if (jj >= Tnv) return;
    const int co = (strip0 + s) * 16 + q * 4;
    const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
    const size_t o = ((size_t)b * Tout + (size_t)jj * a.nphase + ph) * a.Cout + co;
    float4 v = make_float4(av[0] + bs.x, av[1] + bs.y, av[2] + bs.z, av[3] + bs.w);
    if (a.skip) {
      const float4 sk = *reinterpret_cast<const float4*>(a.skip + o);
      v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;
    }
    if (a.out_raw) *reinterpret_cast<float4*>(a.out_raw + o) = v;
    if (a.out_act) {
      const float4 al = *reinterpret_cast<const float4*>(a.alpha + co), ia = ld_inv4(a.alpha, a.Cout, co);
      const float4 sv = make_float4(snake_f<true>(v.x, al.x, ia.x), snake_f<true>(v.y, al.y, ia.y), snake_f<true>(v.z, al.z, ia.z), snake_f<true>(v.w, al.w, ia.w));
      if (!a.act_f32) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.out_act) + o) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out_act) + o) = sv;
    }
  };
#define PTTS_EMIT_ROW(S)                                                                                                  \
  if (CSW > (S)) {                                                                                                        \
    emit(acc[(S) % CSW][0], S, 0); emit(acc[(S) % CSW][1], S, 1); emit(acc[(S) % CSW][2], S, 2); emit(acc[(S) % CSW][3], S, 3); \
    if (FT > 4) { emit(acc[(S) % CSW][4 % FT], S, 4); emit(acc[(S) % CSW][5 % FT], S, 5); emit(acc[(S) % CSW][6 % FT], S, 6); emit(acc[(S) % CSW][7 % FT], S, 7); } \
  }
  PTTS_EMIT_ROW(0)
  PTTS_EMIT_ROW(1)
  PTTS_EMIT_ROW(2)
  PTTS_EMIT_ROW(3)
#undef PTTS_EMIT_ROW
}

// Fused residual unit (default; PTTS_DAC_NO_FUSE_RES=1 restores the two-launch path for A/B): one residual unit of the three narrower blocks
// (C = 384 / 192 / 96) in ONE launch (measured on MI355X, profiles/r03_experiments.txt: 860 frames 3.62 -> 3.25 ms, batch 32 109.4 -> 97.5 ms):
//   y = Snake_a(conv_k7_dil(x) + b7)  ->  bf16 tile in LDS  ->  out = skip + conv_k1(y) + b1 ; act = Snake_a1(out)
// The workgroup owns all C channels of 128 frames (NW * 3 strips = C / 16), so the k1 GEMM's operand never leaves the CU: the unit's
// HBM traffic drops from 16 to 12 bytes per element (no bf16 round trip of y) and one launch per unit goes away (per-layer table,
// profiles/r02_dac_layers.txt: k7 + k1 = 339 us at C = 192, 260 us at C = 96, the k1 half bound by its epilogue traffic).
// Phase A is conv_lds_kernel<3, NW, 1, 54>'s loop; the y tile overlays the two slab buffers once the last chunk's barrier has passed.
// weight prefetch depth of the XIN instances per width (registers: an XIN unit stages twice the bytes per slab slot)
template <int NW> struct ResunitXinWD { static constexpr int value = NW == 2 ? 1 : 3; };
struct ResArgs {
  ConvArgs a;           // the k7 conv (x, Wp, bias, alpha = Snake between the two convs, dil, pad, B, Tin, Cin = Cout = C)
  const void* Wp1;      // k1 weights, packed [C/16][C/32][64][8 bf16]
  const float* bias1;
  const float* alpha1;  // Snake of the unit's output (the next layer's input activation)
  const float* skip;    // fp32 residual stream [B][T][C]
  float* out_raw;       // fp32 residual stream after the unit (may alias skip), or null
  void* out_act;        // activated output, bf16 (fp32 if act_f32): NOT the buffer x lives in (neighbouring tiles read x's halo rows)
  int act_f32;
  int epi_direct;       // A/B (PTTS_DAC_EPI_DIRECT=1): the round-3 epilogue (a lane owns 4 channels of one frame: 64-byte pieces of the stream)
  const float* alpha_in;  // XIN instances: [alpha | 1 / (alpha + 1e-9)] of the Snake that turns the fp32 stream `a.x` into this unit's input activation
};

// dynamic LDS of resunit_lds_kernel<NW>: the two slab buffers of phase A, overlaid by the y tile [128 frames][C bf16 + pad] of phase B
// (NW = 8, C = 384: 100 KB - above the 64 KB a static __shared__ array may declare, hence dynamic for every instance)
template <int NW, int KS = 1> struct ResunitLds {
  static constexpr int C = NW * 3 * 16, slabs = 2 * (128 + 54) * (KS * 64 + 32), ytile = 128 * (C * 2 + 32);
  static constexpr int etile = 64 * (C * 4 + 16);  // epilogue: half of the output tile as fp32 rows (16 bytes of padding: the 16 frames of one store hit 16 different bank groups)
  static constexpr int ain = slabs + 2 * C * 4;  // XIN instances: the input Snake's [alpha | 1 / alpha] behind the two slabs
  static constexpr int bytes0 = ain > ytile ? ain : ytile;
  static constexpr int bytes = bytes0 > etile ? bytes0 : etile;
};
// KS: 32-channel k-steps per staged chunk (1: 64-byte slab rows, the round-3 form; 2 for C >= 192: twice the bytes in flight per staging
// round and twice the MFMA work to hide them behind - the units ran latency x concurrency-bound at ~2.7 TB/s with ~12-16 KB in flight per workgroup)
// WD: how many k-steps ahead a wave requests its weight fragments (1: two register sets, the round-3 form; 3: four sets). A k-step is 24 MFMAs
// = 384 cycles per wave, ~770 with the SIMD's second wave interleaved: one step ahead is ~0.3 us, less than an L2 round trip - and the
// fragments DO come from the L2 every time (a wave re-streams its 3 strips x 7 taps x C channels = 129 KB at C = 192 per tile through a
// 32 KB L1 shared by 8 waves), so every k-step waited for its weights: MFMA pipe 23-28 % busy (profiles/r03_pmc_dac_mfma.txt).
// RAW / F32 (round 5): does the launch write the fp32 stream, and is the activation written as fp32 (the unit feeding the final conv)? Compile-time,
// so that the epilogue's pass over a whole half tile is ONE straight-line block (see there).
// XIN / ACT (round 6): the codec is HBM-bound at batch 32 (profiles/r06_pmc_dac_bs32.txt: 195.7 GB per decode, the C = 96 units at 4.5-4.6 TB/s), so bytes
// are what is left to cut. An XIN unit takes its input straight from the fp32 residual stream (`a.x` = `skip`): the Snake of the PREVIOUS layer's
// output is evaluated on the way into the LDS slab (same function on the same fp32 values, rounded to bf16 once: bit-identical to reading the bf16
// activation the producer would have written), and a producer whose consumer is an XIN unit does not write that activation at all (ACT = false):
// per element and unit 12 C -> ~9.7 C bytes (halo rows of the stream are fp32 now, the bf16 copy is neither written nor read).
template <int NW, int KS = 1, int WD = 3, bool RAW = true, bool F32 = false, bool XIN = false, bool ACT = true>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) resunit_lds_kernel(ResArgs ra) {
  constexpr int CSW = 3, FT = 8, TF = FT * 16, MAXHALO = 54;
  constexpr int C = NW * CSW * 16;
  constexpr int KCH = 32 * KS, RS = KS * 64 + 32, SL = KS * 4, NT = NW * 64;
  constexpr int MAXROWS = TF + MAXHALO;
  constexpr int NST = (MAXROWS * SL + NT - 1) / NT;
  constexpr int RS2 = C * 2 + 32;  // y tile row stride: an odd multiple of 32 bytes, conflict-free ds_read_b128 like the slab
  constexpr int NK1 = C / 32;      // k-steps of the k1 GEMM
  static_assert(ResunitLds<NW, KS>::bytes >= 2 * MAXROWS * RS && ResunitLds<NW, KS>::bytes >= TF * RS2, "LDS size");
  static_assert(C % KCH == 0, "chunk width");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const ConvArgs& a = ra.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & 15;
  const int ntile = (a.Tn + TF - 1) / TF;
  const int tile = blockIdx.x % ntile, b = blockIdx.x / ntile;
  const int strip0 = wave * CSW;
  const int cpt = C / 32, nk = a.ntaps * cpt;
  constexpr int nchunk = C / KCH;
  constexpr int NS = 7 * KS;  // k-steps per chunk (the unit's first conv is k7: run_resunit)
  const int t0 = tile * TF;
  const int Tv = valid_rows(a, b);  // ragged decode: this utterance's rows (workgroup-uniform)
  if (t0 >= Tv) return;
  const int o0 = -a.pad;
  const int halo = (a.ntaps - 1) * a.dil;
  const int nslot = (TF + halo) * SL;
  const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)b * a.Tin * C * (XIN ? 4 : 2);
  float* s_ain = reinterpret_cast<float*>(lds + 2 * MAXROWS * RS);  // XIN: [2][C]
  if constexpr (XIN) {
    for (int i = tid; i < 2 * C / 4; i += NT) reinterpret_cast<float4*>(s_ain)[i] = reinterpret_cast<const float4*>(ra.alpha_in)[i];
    __syncthreads();
  }
  const float4* Wp = reinterpret_cast<const float4*>(a.Wp) + (size_t)strip0 * nk * 64 + lane;
  const int lrow = j * RS + q * 16;
  unsigned char* slab0 = lds;
  unsigned char* slab1 = lds + MAXROWS * RS;

  f32x4 acc[CSW][FT];
#pragma unroll
  for (int s = 0; s < CSW; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f) acc[s][f] = f32x4{0, 0, 0, 0};

  // a staged 16-byte slot = 8 bf16 channels of one frame: 16 bytes of the bf16 activation, or (XIN) 32 bytes of the fp32 stream that become those 16
  // bytes at commit time (Snake, rounded once); rows outside the utterance read as zeros either way (Snake(0) = 0: the conv's zero padding)
  uint4 stg[XIN ? 2 * NST : NST];
#define RU_SLAB_FETCH(CC)                                                                                               \
  _Pragma("unroll") for (int i_ = 0; i_ < NST; ++i_) {                                                                    \
    const int idx_ = tid + i_ * NT, r_ = idx_ / SL, sl_ = idx_ - r_ * SL, ti_ = t0 + o0 + r_;                              \
    const bool ok_ = idx_ < nslot && ti_ >= 0 && ti_ < Tv;                                                                \
    if constexpr (XIN) {                                                                                                  \
      stg[2 * i_] = make_uint4(0, 0, 0, 0); stg[2 * i_ + 1] = make_uint4(0, 0, 0, 0);                                     \
      if (ok_) {                                                                                                          \
        const uint4* p_ = reinterpret_cast<const uint4*>(xb + ((size_t)ti_ * C + (size_t)(CC) * KCH) * 4 + sl_ * 32);     \
        stg[2 * i_] = p_[0]; stg[2 * i_ + 1] = p_[1];                                                                     \
      }                                                                                                                   \
    } else {                                                                                                              \
      stg[i_] = make_uint4(0, 0, 0, 0);                                                                                   \
      if (ok_) stg[i_] = *reinterpret_cast<const uint4*>(xb + ((size_t)ti_ * C + (size_t)(CC) * KCH) * 2 + sl_ * 16);     \
    }                                                                                                                     \
  }
#define RU_SLAB_COMMIT(BUFP, CC)                                                                                        \
  {                                                                                                                     \
    float4 al0_, al1_, ia0_, ia1_;                                                                                        \
    if constexpr (XIN) {  /* the thread's slot column is the same for every i_ (NT % SL == 0): 8 channels of chunk CC */   \
      const float* ap_ = s_ain + (CC) * KCH + (tid % SL) * 8;                                                             \
      al0_ = *reinterpret_cast<const float4*>(ap_); al1_ = *reinterpret_cast<const float4*>(ap_ + 4);                      \
      ia0_ = *reinterpret_cast<const float4*>(ap_ + C); ia1_ = *reinterpret_cast<const float4*>(ap_ + C + 4);              \
    }                                                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < NST; ++i_) {                                                                  \
      const int idx_ = tid + i_ * NT, r_ = idx_ / SL, sl_ = idx_ - r_ * SL;                                                \
      uint4 v_;                                                                                                           \
      if constexpr (XIN) {                                                                                                \
        const uint4 lo_ = stg[2 * i_], hi_ = stg[2 * i_ + 1];                                                             \
        v_ = make_uint4(pack_bf16x2(snake_f<true>(__uint_as_float(lo_.x), al0_.x, ia0_.x), snake_f<true>(__uint_as_float(lo_.y), al0_.y, ia0_.y)), \
                        pack_bf16x2(snake_f<true>(__uint_as_float(lo_.z), al0_.z, ia0_.z), snake_f<true>(__uint_as_float(lo_.w), al0_.w, ia0_.w)), \
                        pack_bf16x2(snake_f<true>(__uint_as_float(hi_.x), al1_.x, ia1_.x), snake_f<true>(__uint_as_float(hi_.y), al1_.y, ia1_.y)), \
                        pack_bf16x2(snake_f<true>(__uint_as_float(hi_.z), al1_.z, ia1_.z), snake_f<true>(__uint_as_float(hi_.w), al1_.w, ia1_.w))); \
      } else {                                                                                                            \
        v_ = stg[i_];                                                                                                     \
      }                                                                                                                   \
      if (idx_ < nslot) *reinterpret_cast<uint4*>((BUFP) + r_ * RS + sl_ * 16) = v_;                                       \
    }                                                                                                                     \
  }
#define RU_W_FETCH(WF, CC, S)                                                                                           \
  do {                                                                                                                  \
    const int tap_ = (S) / KS, kk_ = (S) - tap_ * KS;                                                                     \
    const float4* wp_ = Wp + (size_t)(tap_ * cpt + (CC) * KS + kk_) * 64;                                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < CSW; ++s_) WF[s_] = wp_[(size_t)s_ * nk * 64];                               \
  } while (0)
#define RU_B_FETCH(BV, SB, S, F0)                                                                                       \
  do {                                                                                                                  \
    const int tap_ = (S) / KS, kk_ = (S) - tap_ * KS;                                                                     \
    const unsigned char* sp_ = (SB) + (tap_ * a.dil + (F0) * 16) * RS + kk_ * 64 + lrow;                                  \
    _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) BV[f_] = *reinterpret_cast<const uint4*>(sp_ + f_ * 16 * RS);       \
  } while (0)
#define RU_MFMA_HALF(WC, BV, F0)                                                                                        \
  _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_) _Pragma("unroll") for (int s_ = 0; s_ < CSW; ++s_)                      \
    acc[s_][(F0) + f_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WC[s_]), __builtin_bit_cast(bf16x8, BV[f_]), acc[s_][(F0) + f_], 0, 0, 0);
#define RU_STEP(WC, WN)                                                                                                 \
  {                                                                                                                     \
    const unsigned char* sb_ = (c & 1) ? slab1 : slab0;                                                                   \
    unsigned char* nb_ = (c & 1) ? slab0 : slab1;                                                                         \
    const bool more_ = c + 1 < nchunk, lastst_ = st + 1 == NS;                                                            \
    const int gn_ = min(gi + WD, total - 1); /* unconditional (a valid re-fetch at the very end): a branch here would */   \
    if (st == 0 && more_) { RU_SLAB_FETCH(c + 1); } /* merge into a conservative vmcnt on the MFMAs below */             \
    RU_W_FETCH(WN, gn_ / NS, gn_ % NS);                                                                                   \
    RU_B_FETCH(bB, sb_, st, 4);                                                                                           \
    RU_MFMA_HALF(WC, bA, 0)                                                                                               \
    if (!lastst_) RU_B_FETCH(bA, sb_, st + 1, 0);                                                                         \
    RU_MFMA_HALF(WC, bB, 4)                                                                                               \
    if (lastst_) {                                                                                                        \
      if (more_) { RU_SLAB_COMMIT(nb_, c + 1); }                                                                          \
      __syncthreads();                                                                                                    \
      if (more_) RU_B_FETCH(bA, nb_, 0, 0);                                                                               \
      ++c;                                                                                                                \
      st = 0;                                                                                                             \
    } else {                                                                                                              \
      ++st;                                                                                                               \
    }                                                                                                                     \
    ++gi;                                                                                                                 \
  }
  constexpr int total = nchunk * NS;
  float4 w0[CSW], w1[CSW], w2[CSW], w3[CSW];
  uint4 bA[4], bB[4];
  RU_W_FETCH(w0, 0, 0);
  if constexpr (WD == 3) {
    RU_W_FETCH(w1, (1 < total ? 1 : 0) / NS, (1 < total ? 1 : 0) % NS);
    RU_W_FETCH(w2, (2 < total ? 2 : 0) / NS, (2 < total ? 2 : 0) % NS);
  }
  RU_SLAB_FETCH(0);
  RU_SLAB_COMMIT(slab0, 0);
  __syncthreads();
  RU_B_FETCH(bA, slab0, 0, 0);
  int c = 0, st = 0, gi = 0;
  if constexpr (WD == 3) {  // four register sets: k-step g computes out of w[g % 4] while the fragments of k-step g + 3 land in w[(g + 3) % 4]
    for (int g = 0; g < total; g += 4) {
      RU_STEP(w0, w3)
      if (g + 1 < total) RU_STEP(w1, w0)
      if (g + 2 < total) RU_STEP(w2, w1)
      if (g + 3 < total) RU_STEP(w3, w2)
    }
  } else {
    for (int g = 0; g < total; g += 2) {
      RU_STEP(w0, w1)
      if (g + 1 < total) RU_STEP(w1, w0)
    }
  }
#undef RU_STEP
#undef RU_MFMA_HALF
#undef RU_B_FETCH
#undef RU_W_FETCH
#undef RU_SLAB_COMMIT
#undef RU_SLAB_FETCH
  // ---- phase A epilogue: y = Snake(acc + b7) as bf16 into the LDS tile [frame][C] (the loop's last barrier has retired every slab read)
  unsigned char* ytile = lds;
  auto put_y = [&](const f32x4 av, const int s, const int f) {
    const int co = (strip0 + s) * 16 + q * 4;
    const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
    const float4 al = *reinterpret_cast<const float4*>(a.alpha + co), ia = ld_inv4(a.alpha, C, co);
    const float4 sv = make_float4(snake_f<true>(av[0] + bs.x, al.x, ia.x), snake_f<true>(av[1] + bs.y, al.y, ia.y), snake_f<true>(av[2] + bs.z, al.z, ia.z),
                                  snake_f<true>(av[3] + bs.w, al.w, ia.w));
    *reinterpret_cast<uint2*>(ytile + (f * 16 + j) * RS2 + co * 2) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
  };
#define RU_PUT_ROW(S)                                                                                                   \
  put_y(acc[S][0], S, 0); put_y(acc[S][1], S, 1); put_y(acc[S][2], S, 2); put_y(acc[S][3], S, 3);                         \
  put_y(acc[S][4], S, 4); put_y(acc[S][5], S, 5); put_y(acc[S][6], S, 6); put_y(acc[S][7], S, 7);
  RU_PUT_ROW(0)
  RU_PUT_ROW(1)
  RU_PUT_ROW(2)
#undef RU_PUT_ROW
  __syncthreads();
  // the epilogue's residual rows: thread i owns float4 i, i + NT, ... of the [64][C] half tile, all 12 loads of a half in flight at once.
  // (Requesting them BEFORE the k1 GEMM, and 64-channel staging chunks, were measured and dropped: 72.3 vs 70.2 ms per batch-32 decode,
  // profiles/r04_experiments.txt - the units are not short of bytes in flight.)
  constexpr int VPRH = C / 4, NPT = 64 * VPRH / NT;  // float4 per row; per thread and half tile (= 12 for every NW)
  float4 skp[NPT];
  auto load_skip = [&](const int hh) __attribute__((always_inline)) {
    const int r0_ = t0 + hh * 64, rows_ = min(64, Tv - r0_);
    const float* sb_ = ra.skip + ((size_t)b * a.Tn + r0_) * C;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i_ = tid + k * NT;
      skp[k] = (i_ / VPRH) < rows_ ? *reinterpret_cast<const float4*>(sb_ + (size_t)i_ * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // ---- phase B: the k1 conv as a [C x C] GEMM over the tile; A = this wave's 3 strips of W1, B = y rows out of LDS
  const float4* W1 = reinterpret_cast<const float4*>(ra.Wp1) + (size_t)strip0 * NK1 * 64 + lane;
#pragma unroll
  for (int s = 0; s < CSW; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f) acc[s][f] = f32x4{0, 0, 0, 0};
  const unsigned char* yl = ytile + j * RS2 + q * 16;
#pragma unroll
  for (int kk = 0; kk < NK1; ++kk) {
    float4 wk[CSW];
#pragma unroll
    for (int s = 0; s < CSW; ++s) wk[s] = W1[((size_t)s * NK1 + kk) * 64];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const uint4 bv = *reinterpret_cast<const uint4*>(yl + f * 16 * RS2 + kk * 64);
#pragma unroll
      for (int s = 0; s < CSW; ++s)
        acc[s][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wk[s]), __builtin_bit_cast(bf16x8, bv), acc[s][f], 0, 0, 0);
    }
  }
  // ---- phase B epilogue: + bias + residual, fp32 stream out, Snake of the unit's output.
  // Through LDS (default): a lane's accumulators are 4 channels of ONE frame, so the direct form below moves the fp32 stream in 64-byte pieces
  // (rows C * 4 bytes apart) and ran the three fused units at 2.3-2.9 TB/s, 68 % of the batch-32 decode (profiles/r03_dac_kernels_bs32.txt).
  // Here the output tile goes through LDS in two halves of 64 frames ([64][C fp32], rows padded by 16 bytes) and every lane then owns 16
  // CONSECUTIVE bytes of a row: the residual read, the stream write and the activation write are whole rows (tools/epilogue_probe.hip on
  // MI355X: 2.5-2.8 -> 3.9-5.1 TB/s with 64-frame tiles; 128-frame tiles do not pay). Same fp32 operations in the same order: bit-identical.
  if (!ra.epi_direct) {
    constexpr int RSE = C * 4 + 16, VPR = C / 4;
    unsigned char* et = lds;
    // The output Snake's [alpha | 1 / (alpha + 1e-9)] go through LDS (round 5): fetched from global memory inside the loop below, every iteration's
    // s_waitcnt vmcnt() for them also waited for the STORES of the iteration before (loads and stores retire in order on one counter) - 24 store
    // round trips in a row per workgroup, ~2/3 of the time the three fused units spent outside their MFMA loops (found in the ISA, not in a counter).
    static_assert(ResunitLds<NW, KS>::bytes >= 64 * RSE + 2 * C * 4, "LDS: half output tile + the output Snake's parameters");
    float* s_al = reinterpret_cast<float*>(lds + 64 * RSE);  // [2][C]
    __syncthreads();  // every wave has finished reading the y tile
    if constexpr (ACT)
      for (int i = tid; i < 2 * C / 4; i += NT) reinterpret_cast<float4*>(s_al)[i] = reinterpret_cast<const float4*>(ra.alpha1)[i];
    // (requesting the residual rows before the transposition, and the second half's as the first half's registers free up, measured SLOWER:
    //  C = 96 unit 4228 -> 4735 us, profiles/r04_experiments.txt call 7; they are requested after the tile's barrier, 12 loads at once)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int s = 0; s < CSW; ++s) {
        const int co = (strip0 + s) * 16 + q * 4;
        const float4 bs = *reinterpret_cast<const float4*>(ra.bias1 + co);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const f32x4 av = acc[s][hh * 4 + f];
          *reinterpret_cast<float4*>(et + (f * 16 + j) * RSE + co * 4) = make_float4(av[0] + bs.x, av[1] + bs.y, av[2] + bs.z, av[3] + bs.w);
        }
      }
      __syncthreads();
      const int r0 = t0 + hh * 64;
      const int rows = min(64, Tv - r0);
      const size_t base = ((size_t)b * a.Tn + r0) * C;
      // A whole half tile (every tile but an utterance's last) is ONE straight-line block: 12 unconditional residual loads, then per row piece
      // LDS reads -> add -> store(s) -> Snake -> store, no load from global memory and no branch in between. With the stream / activation format
      // decided by branches inside the loop (and the Snake parameters loaded from global memory there), the compiler's wait-count bookkeeping
      // put an s_waitcnt vmcnt(0) - "until my last STORES are acknowledged", loads and stores retire in order on one counter - in front of every
      // row piece: 24 store round trips in a row per workgroup.
      auto piece = [&](const int k, const float4 sk) __attribute__((always_inline)) {
        const int i = tid + k * NT;
        const int rr = i / VPR, cv = i - rr * VPR;
        const float4 av = *reinterpret_cast<const float4*>(et + rr * RSE + cv * 16);
        const size_t o = base + (size_t)i * 4;  // the rows of a tile are contiguous in memory: i * 4 == rr * C + cv * 4
        const float4 v = make_float4(av.x + sk.x, av.y + sk.y, av.z + sk.z, av.w + sk.w);
        if constexpr (RAW) *reinterpret_cast<float4*>(ra.out_raw + o) = v;
        if constexpr (ACT) {  // ACT = false: the consumer is an XIN unit and evaluates this Snake itself, out of the stream
          const float4 al = *reinterpret_cast<const float4*>(s_al + cv * 4), ia = *reinterpret_cast<const float4*>(s_al + C + cv * 4);
          const float4 sv = make_float4(snake_f<true>(v.x, al.x, ia.x), snake_f<true>(v.y, al.y, ia.y), snake_f<true>(v.z, al.z, ia.z), snake_f<true>(v.w, al.w, ia.w));
          if constexpr (!F32) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(ra.out_act) + o) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
          else *reinterpret_cast<float4*>(reinterpret_cast<float*>(ra.out_act) + o) = sv;
        }
      };
      if (rows == 64) {
        const float* sb_ = ra.skip + base;
#pragma unroll
        for (int k = 0; k < NPT; ++k) skp[k] = *reinterpret_cast<const float4*>(sb_ + (size_t)(tid + k * NT) * 4);
#pragma unroll
        for (int k = 0; k < NPT; ++k) piece(k, skp[k]);
      } else {
        load_skip(hh);
#pragma unroll
        for (int k = 0; k < NPT; ++k)
          if ((tid + k * NT) / VPR < rows) piece(k, skp[k]);
      }
      if (hh == 0) __syncthreads();  // the tile is rewritten by the second half
    }
    return;
  }
  auto emit = [&](const f32x4 av, const int s, const int f) {
    const int jj = t0 + f * 16 + j;
    if (jj >= Tv) return;
    const int co = (strip0 + s) * 16 + q * 4;
    const float4 bs = *reinterpret_cast<const float4*>(ra.bias1 + co);
    const size_t o = ((size_t)b * a.Tn + (size_t)jj) * C + co;
    const float4 sk = *reinterpret_cast<const float4*>(ra.skip + o);
    const float4 v = make_float4(av[0] + bs.x + sk.x, av[1] + bs.y + sk.y, av[2] + bs.z + sk.z, av[3] + bs.w + sk.w);
    if (ra.out_raw) *reinterpret_cast<float4*>(ra.out_raw + o) = v;
    if (ra.out_act) {
      const float4 al = *reinterpret_cast<const float4*>(ra.alpha1 + co), ia = ld_inv4(ra.alpha1, C, co);
      const float4 sv = make_float4(snake_f<true>(v.x, al.x, ia.x), snake_f<true>(v.y, al.y, ia.y), snake_f<true>(v.z, al.z, ia.z), snake_f<true>(v.w, al.w, ia.w));
      if (!ra.act_f32) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(ra.out_act) + o) = make_uint2(pack_bf16x2(sv.x, sv.y), pack_bf16x2(sv.z, sv.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(ra.out_act) + o) = sv;
    }
  };
#define RU_EMIT_ROW(S)                                                                                                  \
  emit(acc[S][0], S, 0); emit(acc[S][1], S, 1); emit(acc[S][2], S, 2); emit(acc[S][3], S, 3);                             \
  emit(acc[S][4], S, 4); emit(acc[S][5], S, 5); emit(acc[S][6], S, 6); emit(acc[S][7], S, 7);
  RU_EMIT_ROW(0)
  RU_EMIT_ROW(1)
  RU_EMIT_ROW(2)
#undef RU_EMIT_ROW
}

// final Conv1d(C -> 1, k7, pad 3) + tanh; weights [7][C] in LDS. One thread per OS = 4 consecutive output samples: the 10 input rows
// they touch are read once (60 float4 loads per sample instead of 168; the kernel was 187 us of the 860-frame decode, L1-bound on
// the 7x re-read). Per sample the fma order is unchanged (bias, then tap 0..6 x channel 0..C-1), so the exact-f32 mode is bit-identical.
// samples [skip, t_end) of every utterance are written to out[b * out_ld + (t - skip)] (skip > 0: the halo frames of a chunk)
constexpr int OUT_OS = 4;
__global__ void conv_out_tanh_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                     float* __restrict__ out, int B, int T, int C, int ktaps, int skip, long long out_ld, int t_end,
                                     const int* __restrict__ lens, int len_mul) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < ktaps * C; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const int TG = (T + OUT_OS - 1) / OUT_OS;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * TG) return;
  const int b = (int)(idx / TG), t0 = (int)(idx % TG) * OUT_OS;
  const int Tv = lens ? min(T, lens[b] * len_mul) : T;  // ragged decode: samples of this utterance (rows beyond are the zero padding)
  t_end = min(t_end, Tv);
  if (t0 + OUT_OS <= skip || t0 >= t_end) return;
  float acc[OUT_OS];
#pragma unroll
  for (int s = 0; s < OUT_OS; ++s) acc[s] = bias[0];
  const int half = ktaps / 2;
  for (int r = 0; r < OUT_OS + ktaps - 1; ++r) {  // input row t0 - half + r feeds sample s with tap r - s
    const int ti = t0 - half + r;
    if (ti < 0 || ti >= Tv) continue;
    const float4* xr = reinterpret_cast<const float4*>(x + ((size_t)b * T + ti) * C);
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const float4 xv = xr[c4];
#pragma unroll
      for (int s = 0; s < OUT_OS; ++s) {
        const int tap = r - s;
        if (tap >= 0 && tap < ktaps) {
          const float4 wv = reinterpret_cast<const float4*>(sw + tap * C)[c4];
          acc[s] = fmaf(xv.x, wv.x, acc[s]); acc[s] = fmaf(xv.y, wv.y, acc[s]); acc[s] = fmaf(xv.z, wv.z, acc[s]); acc[s] = fmaf(xv.w, wv.w, acc[s]);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < OUT_OS; ++s) {
    const int t = t0 + s;
    if (t < Tv && t >= skip && t < t_end) out[(size_t)b * out_ld + (t - skip)] = tanhf(acc[s]);
  }
}

// The same final convolution, LDS-tiled (C == 96, k7: the 44.1 kHz decoder). conv_out_tanh_kernel's threads each walk their own
// input rows (a wave-load touches 64 different 384-byte rows: 169 MB at 1.4 TB/s, 121 us of the 860-frame decode). Here a workgroup
// owns OUT_TILE consecutive samples: the (OUT_TILE + 6) x 96 input tile is ONE contiguous 27 KB block, staged into LDS with coalesced
// float4 loads; rows are 100 floats apart (400 B = 36 dwords mod 64: the 16 lanes of a ds_read_b128 service group hit 16 distinct
// 4-bank slots), the 7 x 96 weights are read as wave-uniform broadcasts. Thread t then computes sample t with EXACTLY the fma order
// of the reference restatement (bias, tap 0..6 x channel 0..95): the exact-f32 mode stays bit-identical to the direct kernel.
constexpr int OUT_TILE = 64, OUT_C = 96, OUT_ROW = 100;  // 64 samples per (one-wave) workgroup: 28 KB of LDS, 5 workgroups per CU
__global__ void __launch_bounds__(OUT_TILE) conv_out_tanh_lds_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                     float* __restrict__ out, int T, int skip, long long out_ld, int t_end,
                                                                     const int* __restrict__ lens, int len_mul) {
  __shared__ __attribute__((aligned(16))) float sx[(OUT_TILE + 6) * OUT_ROW];
  __shared__ __attribute__((aligned(16))) float sw[7 * OUT_C];
  const int b = blockIdx.y, t0 = blockIdx.x * OUT_TILE, tid = threadIdx.x;
  const int Tv = lens ? min(T, lens[b] * len_mul) : T;  // ragged decode: samples of this utterance (rows beyond are the zero padding)
  t_end = min(t_end, Tv);
  if (t0 + OUT_TILE <= skip || t0 >= t_end) return;  // workgroup-uniform: nothing of this tile is emitted
  for (int i = tid; i < 7 * OUT_C / 4; i += OUT_TILE) reinterpret_cast<float4*>(sw)[i] = reinterpret_cast<const float4*>(w)[i];
  const float4* xb = reinterpret_cast<const float4*>(x + (size_t)b * T * OUT_C);
  constexpr int NV = (OUT_TILE + 6) * (OUT_C / 4), UL = 9;  // 1680 float4 of the tile (rows t0 - 3 .. t0 + OUT_TILE + 2), 9 loads in flight per thread
  for (int i0 = tid; i0 < NV; i0 += UL * OUT_TILE) {
    float4 v[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int i = min(i0 + u * OUT_TILE, NV - 1), r = i / (OUT_C / 4), ti = t0 - 3 + r;
      v[u] = (ti >= 0 && ti < Tv) ? xb[(size_t)ti * (OUT_C / 4) + (i - r * (OUT_C / 4))] : make_float4(0.f, 0.f, 0.f, 0.f);  // zero outside [0, Tv)
    }
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int i = i0 + u * OUT_TILE, r = i / (OUT_C / 4);
      if (i < NV) *reinterpret_cast<float4*>(sx + r * OUT_ROW + (i - r * (OUT_C / 4)) * 4) = v[u];
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t >= Tv || t < skip || t >= t_end) return;
  float acc = bias[0];
#pragma unroll 1
  for (int tap = 0; tap < 7; ++tap) {
    const int ti = t - 3 + tap;
    if (ti < 0 || ti >= Tv) continue;  // the direct kernel skips out-of-range rows too (no fma with the zero padding)
    const float* xr = sx + (tid + tap) * OUT_ROW;
    const float* wr = sw + tap * OUT_C;
#pragma unroll
    for (int c4 = 0; c4 < OUT_C / 4; ++c4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + c4 * 4), wv = *reinterpret_cast<const float4*>(wr + c4 * 4);
      acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
    }
  }
  out[(size_t)b * out_ld + (t - skip)] = tanhf(acc);
}

// first encoder Conv1d(1 -> C, k7, pad 3) on the raw waveform: one thread per (sample, 4 channels); writes the raw
// result (residual skip of the first unit) and its Snake. HBM-bound (C floats out per sample), trivially parallel.
__global__ void conv_in_kernel(const float* __restrict__ wave, const float* __restrict__ w /*[C][7]*/, const float* __restrict__ bias,
                               const float* __restrict__ alpha, float* __restrict__ out_raw, float* __restrict__ out_act, int B, int L, int C) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  if (idx >= (size_t)B * L * c4n) return;
  const int c4 = (int)(idx % c4n);
  const size_t bt = idx / c4n;
  const int t = (int)(bt % L), b = (int)(bt / L);
  float xs[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { const int ti = t + k - 3; xs[k] = (ti >= 0 && ti < L) ? wave[(size_t)b * L + ti] : 0.f; }
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c4 * 4 + e;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc = fmaf(w[c * 7 + k], xs[k], acc);
    o[e] = acc + bias[c];
  }
  const size_t off = bt * C + c4 * 4;
  *reinterpret_cast<float4*>(out_raw + off) = make_float4(o[0], o[1], o[2], o[3]);
  const float4 al = *reinterpret_cast<const float4*>(alpha + c4 * 4), ia = ld_inv4(alpha, C, c4 * 4);
  *reinterpret_cast<float4*>(out_act + off) = make_float4(snake_f<false>(o[0], al.x, ia.x), snake_f<false>(o[1], al.y, ia.y), snake_f<false>(o[2], al.z, ia.z), snake_f<false>(o[3], al.w, ia.w));
}

// F.normalize(codebook) rows: c / max(||c||_2, 1e-12)
__global__ void cb_normalize_kernel(const float* __restrict__ cb, float* __restrict__ cbn, float* __restrict__ cbn_sq, int ncodes, int cdim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncodes) return;
  float ss = 0.f;
  for (int d = 0; d < cdim; ++d) ss += cb[(size_t)i * cdim + d] * cb[(size_t)i * cdim + d];
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  float s2 = 0.f;
  for (int d = 0; d < cdim; ++d) { const float v = cb[(size_t)i * cdim + d] * inv; cbn[(size_t)i * cdim + d] = v; s2 += v * v; }
  cbn_sq[i] = s2;  // the ||c||^2 term of the reference's distance (1 up to rounding, kept so that near-ties order the same way)
}

// Residual VQ encode (ResidualVectorQuantize.forward, eval mode), one workgroup per latent frame, all stages in sequence:
//   p = in_proj_i(residual); idx = argmax_c <normalize(p), normalize(codebook_i[c])> (first index on ties);
//   z_q = out_proj_i(p + (codebook_i[idx] - p));  residual -= z_q.
// in_w [K][cdim][latent], in_b [K][cdim], cb/cbn [K][codes][cdim], out_w [K][latent][cdim], out_b [K][latent]. cdim <= 16.
constexpr int RVQ_MAXD = 16;
__global__ void __launch_bounds__(256) rvq_encode_kernel(const float* __restrict__ z /*[B][T][latent]*/, const float* __restrict__ in_w,
                                                         const float* __restrict__ in_b, const float* __restrict__ cb, const float* __restrict__ cbn,
                                                         const float* __restrict__ cbn_sq, const float* __restrict__ out_w, const float* __restrict__ out_b,
                                                         long long* __restrict__ codes /*[B][nq][T]*/, int T, int latent, int cdim, int ncodes, int nq) {
  extern __shared__ float s_res[];  // [latent] residual, then [4 waves][RVQ_MAXD] partials
  __shared__ float s_part[4][RVQ_MAXD];
  __shared__ float s_p[RVQ_MAXD], s_e[RVQ_MAXD], s_q[RVQ_MAXD];
  __shared__ float s_ee;
  __shared__ float s_best[4];
  __shared__ int s_bidx[4];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* zr = z + ((size_t)b * T + t) * latent;
  for (int c = tid; c < latent; c += 256) s_res[c] = zr[c];
  __syncthreads();
  for (int i = 0; i < nq; ++i) {
    const float* W = in_w + (size_t)i * cdim * latent;
    // 1. p[d] = b[d] + sum_c W[d][c] * r[c]
    float part[RVQ_MAXD];
#pragma unroll
    for (int d = 0; d < RVQ_MAXD; ++d) part[d] = 0.f;
    for (int c = tid; c < latent; c += 256) {
      const float r = s_res[c];
#pragma unroll
      for (int d = 0; d < RVQ_MAXD; ++d)
        if (d < cdim) part[d] = fmaf(W[(size_t)d * latent + c], r, part[d]);
    }
#pragma unroll
    for (int d = 0; d < RVQ_MAXD; ++d) {
      if (d < cdim) {
        const float v = wave_sum(part[d]);
        if (lane == 0) s_part[wave][d] = v;
      }
    }
    __syncthreads();
    if (tid < cdim) s_p[tid] = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + in_b[(size_t)i * cdim + tid];
    __syncthreads();
    if (tid == 0) {
      float ss = 0.f;
      for (int d = 0; d < cdim; ++d) ss += s_p[d] * s_p[d];
      const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
      float ee = 0.f;
      for (int d = 0; d < cdim; ++d) { s_e[d] = s_p[d] * inv; ee += s_e[d] * s_e[d]; }
      s_ee = ee;
    }
    __syncthreads();
    // 2. nearest code: argmax of -(||e||^2 - 2 e.c + ||c||^2) over the normalised codebook, first index on ties
    float best = -INFINITY;
    int bidx = 0;
    const float ee = s_ee;
    for (int c = tid; c < ncodes; c += 256) {
      const float* cr = cbn + ((size_t)i * ncodes + c) * cdim;
      float dot = 0.f;
      for (int d = 0; d < cdim; ++d) dot = fmaf(s_e[d], cr[d], dot);
      const float score = -((ee - 2.0f * dot) + cbn_sq[(size_t)i * ncodes + c]);
      if (score > best) { best = score; bidx = c; }
    }
    const float wb = wave_max(best);
    // lowest code index among the lanes holding the wave maximum
    int cand = best == wb ? bidx : 0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
    if (lane == 0) { s_best[wave] = wb; s_bidx[wave] = cand; }
    __syncthreads();
    if (tid == 0) {
      float bb = s_best[0]; int bi = s_bidx[0];
      for (int w2 = 1; w2 < 4; ++w2)
        if (s_best[w2] > bb || (s_best[w2] == bb && s_bidx[w2] < bi)) { bb = s_best[w2]; bi = s_bidx[w2]; }
      codes[((size_t)b * nq + i) * T + t] = bi;
      const float* cr = cb + ((size_t)i * ncodes + bi) * cdim;
      for (int d = 0; d < cdim; ++d) { const float diff = cr[d] - s_p[d]; s_q[d] = s_p[d] + diff; }  // straight-through expression
    }
    __syncthreads();
    // 3. residual -= out_proj(q)
    const float* OW = out_w + (size_t)i * latent * cdim;
    for (int c = tid; c < latent; c += 256) {
      float acc = 0.f;
      for (int d = 0; d < cdim; ++d) acc = fmaf(OW[(size_t)c * cdim + d], s_q[d], acc);
      s_res[c] -= acc + out_b[(size_t)i * latent + c];
    }
    __syncthreads();
  }
}

// RVQ table: table[i][code][c] = bias_i[c] + sum_d W_i[c][d] * codebook_i[code][d]
__global__ void rvq_table_kernel(const float* __restrict__ cb, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ table, int ncodes, int cdim, int latent) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)ncodes * latent) return;
  const int code = (int)(idx / latent), c = (int)(idx % latent);
  float acc = 0.f;
  for (int d = 0; d < cdim; ++d) acc = fmaf(w[(size_t)c * cdim + d], cb[(size_t)code * cdim + d], acc);
  table[idx] = acc + bias[c];
}

// z[b][t][c] = sum_i table[i][codes[b][i][t]][c]   (sequential over i, like from_codes)
// codes rows have stride `ld` frames and the window starts at frame `t0` (chunked / streaming decode reads a slice in place)
__global__ void rvq_gather_kernel(const long long* __restrict__ codes, const float* __restrict__ table, void* __restrict__ z,
                                  int K, int T, int ncodes, int latent, int z_bf16, long long ld, int t0, const int* __restrict__ lens) {
  const int t = blockIdx.x, b = blockIdx.y;
  if (lens && t >= lens[b]) return;  // ragged decode: frames beyond the utterance's length are never read
  __shared__ int s_code[32];
  if (threadIdx.x < K) {
    long long cde = codes[((size_t)b * K + threadIdx.x) * ld + t0 + t];
    if (cde < 0) cde = 0;
    if (cde >= ncodes) cde = ncodes - 1;
    s_code[threadIdx.x] = (int)cde;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < latent; c += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < K; ++i) acc += table[((size_t)i * ncodes + s_code[i]) * latent + c];
    if (z_bf16) reinterpret_cast<bf16_t*>(z)[((size_t)b * T + t) * latent + c] = f32_to_bf16(acc);
    else reinterpret_cast<float*>(z)[((size_t)b * T + t) * latent + c] = acc;
  }
}

// pack conv weights into MFMA A-fragment order. src element (co, ci, k) at src[co*s_co + ci*s_ci + k*s_k].
__global__ void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int ntaps, int nphase,
                                 const int* __restrict__ ktap /*[phase][tap] -> kernel index*/, long long s_co, long long s_ci, long long s_k) {
  const int cpt = Cin / 16, nk = ntaps * cpt, nstrips = Cout / 16;
  const size_t total = (size_t)nphase * nstrips * nk * 64;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  size_t r = idx >> 6;
  const int ks = (int)(r % nk); r /= nk;
  const int strip = (int)(r % nstrips);
  const int ph = (int)(r / nstrips);
  const int tap = ks / cpt, cc = ks % cpt;
  const int co = strip * 16 + (lane & 15);
  const int kidx = ktap[ph * MAXTAPS + tap];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ci = cc * 16 + (lane >> 4) * 4 + e;
    dst[idx * 4 + e] = src[co * s_co + ci * s_ci + kidx * s_k];
  }
}

// bf16-operand mode: [phase][Cout/16][ntaps*Cin/32][64 lanes][8 bf16]; lane: row = lane & 15, channels (lane >> 4) * 8 .. + 8
__global__ void pack_conv_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int Cout, int Cin, int ntaps, int nphase,
                                      const int* __restrict__ ktap, long long s_co, long long s_ci, long long s_k) {
  const int cpt = Cin / 32, nk = ntaps * cpt, nstrips = Cout / 16;
  const size_t total = (size_t)nphase * nstrips * nk * 64;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  size_t r = idx >> 6;
  const int ks = (int)(r % nk); r /= nk;
  const int strip = (int)(r % nstrips);
  const int ph = (int)(r / nstrips);
  const int tap = ks / cpt, cc = ks % cpt;
  const int co = strip * 16 + (lane & 15);
  const int kidx = ktap[ph * MAXTAPS + tap];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = cc * 32 + (lane >> 4) * 8 + e;
    dst[idx * 8 + e] = f32_to_bf16(src[co * s_co + ci * s_ci + kidx * s_k]);
  }
}

struct ConvLayer {
  std::string name;        // descript module name, e.g. "decoder.model.1.block.1"
  std::string alpha_name;  // Snake applied to this conv's OUTPUT (the next layer's input activation), or ""
  int Cin, Cout, ksize, dil, stride;  // stride > 1 => transposed
  bool transposed;
  float *Wp = nullptr, *bias = nullptr, *alpha = nullptr;
  bool has_skip = false, write_raw = false;
  bool bf16 = false;       // bf16-operand mode (decoder layers of a PTTS_BF16 engine)
  int pad = -1;            // explicit padding (down-sampling convs, k3 final conv); -1: "same" padding (k-1)*dil/2
};

}  // namespace

struct ptts_dac {
  ptts_dac_config cfg;
  hipStream_t own_stream = nullptr;
  std::vector<void*> allocs;
  std::vector<ConvLayer> convs;  // all MFMA convs in execution order
  float *out_w = nullptr, *out_b = nullptr;  // final Conv1d(C->1): [7][C], [1]
  int out_C = 0;
  float* table = nullptr;  // [K][codes][latent]
  std::vector<float*> cb, opw, opb;
  float *bufA0 = nullptr, *bufA1 = nullptr, *bufY = nullptr, *bufS = nullptr, *bufZ = nullptr;
  // encode side (cfg.encoder_dim > 0)
  std::vector<ConvLayer> enc_convs;             // MFMA convs of the encoder in execution order
  float *in0_w = nullptr, *in0_b = nullptr, *in0_alpha = nullptr;  // encoder.block.0: Conv1d(1 -> encoder_dim, k7) + Snake of the first unit
  float *ipw = nullptr, *ipb = nullptr;         // in_proj [K][cdim][latent], [K][cdim]
  float *cb_all = nullptr, *cbn = nullptr, *cbn_sq = nullptr, *opw_all = nullptr, *opb_all = nullptr;  // contiguous copies for the VQ search
  bool vq_ready = false;
  int* d_ktap = nullptr;
  std::set<std::string> loaded, required;
  bool table_ready = false;
  int hop = 1;

  template <typename T> int alloc(T** p, size_t n) {
    void* v = nullptr;
    hipError_t e = hipMalloc(&v, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    allocs.push_back(v);
    *p = reinterpret_cast<T*>(v);
    return PTTS_OK;
  }
};

extern "C" void ptts_dac_destroy(ptts_dac* d) {
  if (!d) return;
  PttsDeviceGuard _dg(d->cfg.device);
  hipDeviceSynchronize();
  for (void* p : d->allocs) hipFree(p);
  if (d->own_stream) hipStreamDestroy(d->own_stream);
  delete d;
}

extern "C" int ptts_dac_create(const ptts_dac_config* cfg, ptts_dac** out) {
  PTTS_CHECK(cfg && out, PTTS_E_INVALID, "null argument");
  const ptts_dac_config& c = *cfg;
  PTTS_CHECK(c.num_rates >= 1 && c.num_rates <= 8, PTTS_E_INVALID, "num_rates out of range");
  PTTS_CHECK(c.num_codebooks >= 1 && c.num_codebooks <= 32, PTTS_E_INVALID, "num_codebooks out of range");
  PTTS_CHECK(c.latent_dim % 16 == 0 && c.decoder_dim % (16 << c.num_rates) == 0, PTTS_E_UNSUPPORTED,
             "latent_dim and every decoder width must be multiples of 16");
  PTTS_CHECK(c.encoder_dim >= 0 && c.encoder_dim % 16 == 0 && c.codebook_dim <= RVQ_MAXD, PTTS_E_UNSUPPORTED,
             "encoder_dim must be 0 (decode only) or a multiple of 16, codebook_dim <= %d", RVQ_MAXD);
  PTTS_CHECK(c.compute_dtype == PTTS_F32 || c.compute_dtype == PTTS_BF16, PTTS_E_INVALID, "compute_dtype must be PTTS_F32 or PTTS_BF16");
  if (c.compute_dtype == PTTS_BF16)
    PTTS_CHECK(c.latent_dim % 32 == 0 && c.decoder_dim % (32 << c.num_rates) == 0, PTTS_E_UNSUPPORTED,
               "bf16-operand mode needs latent_dim and every decoder width to be multiples of 32");
  PTTS_CHECK(c.max_batch >= 1 && c.max_frames >= 1, PTTS_E_INVALID, "bad capacities");
  for (int i = 0; i < c.num_rates; ++i) PTTS_CHECK(c.rates[i] >= 2 && c.rates[i] <= 8 && c.rates[i] % 2 == 0, PTTS_E_UNSUPPORTED, "decoder rate %d unsupported (need an even stride in [2, 8])", c.rates[i]);
  PTTS_DEVICE(c.device);
  ptts_dac* d = new ptts_dac();
  d->cfg = c;
  int rc = PTTS_OK;
  auto fail = [&](int r) { ptts_dac_destroy(d); return r; };
  if (hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(ptts_fail(PTTS_E_HIP, "hipStreamCreate failed"));
#define A(expr) if ((rc = (expr)) != PTTS_OK) return fail(rc)
  std::vector<ConvLayer>* target = &d->convs;
  auto add_conv = [&](const std::string& name, const std::string& alpha_name, int Cin, int Cout, int k, int dil, int stride, bool tr,
                      bool has_skip, bool write_raw) -> int {
    ConvLayer L;
    L.name = name; L.alpha_name = alpha_name; L.Cin = Cin; L.Cout = Cout; L.ksize = k; L.dil = dil; L.stride = stride; L.transposed = tr;
    L.has_skip = has_skip; L.write_raw = write_raw;
    L.bf16 = target == &d->convs && c.compute_dtype == PTTS_BF16;  // the encoder always runs the exact-f32 kernels
    const int ntaps = tr ? 2 : k, nphase = tr ? stride : 1;
    if (ntaps > 16) return ptts_fail(PTTS_E_UNSUPPORTED, "%s: kernel size %d unsupported", name.c_str(), k);
    PTTS_TRY(d->alloc(&L.Wp, (size_t)nphase * Cout * ntaps * Cin));
    PTTS_TRY(d->alloc(&L.bias, Cout));
    if (!alpha_name.empty()) PTTS_TRY(d->alloc(&L.alpha, 2 * (size_t)Cout));  // [alpha | 1 / (alpha + 1e-9)]
    d->required.insert(name + ".weight");
    d->required.insert(name + ".bias");
    if (!alpha_name.empty()) d->required.insert(alpha_name + ".alpha");
    target->push_back(L);
    return PTTS_OK;
  };
  char nm[128], an[128];
  int ch = c.decoder_dim;
  snprintf(an, sizeof an, "decoder.model.1.block.0");
  A(add_conv("decoder.model.0", an, c.latent_dim, ch, 7, 1, 1, false, false, false));
  int hop = 1;
  for (int bi = 0; bi < c.num_rates; ++bi) {
    const int cin = ch >> bi, cout = ch >> (bi + 1), s = c.rates[bi];
    hop *= s;
    snprintf(nm, sizeof nm, "decoder.model.%d.block.1", bi + 1);
    snprintf(an, sizeof an, "decoder.model.%d.block.2.block.0", bi + 1);
    A(add_conv(nm, an, cin, cout, 2 * s, 1, s, true, false, true));
    const int dils[3] = {1, 3, 9};
    for (int ri = 0; ri < 3; ++ri) {
      snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.1", bi + 1, ri + 2);
      snprintf(an, sizeof an, "decoder.model.%d.block.%d.block.2", bi + 1, ri + 2);
      A(add_conv(nm, an, cout, cout, 7, dils[ri], 1, false, false, false));
      snprintf(nm, sizeof nm, "decoder.model.%d.block.%d.block.3", bi + 1, ri + 2);
      if (ri < 2) snprintf(an, sizeof an, "decoder.model.%d.block.%d.block.0", bi + 1, ri + 3);
      else if (bi + 1 < c.num_rates) snprintf(an, sizeof an, "decoder.model.%d.block.0", bi + 2);
      else snprintf(an, sizeof an, "decoder.model.%d", c.num_rates + 1);
      A(add_conv(nm, an, cout, cout, 1, 1, 1, false, true, ri < 2));
    }
  }
  d->hop = hop;
  d->out_C = ch >> c.num_rates;
  A(d->alloc(&d->out_w, (size_t)7 * d->out_C));
  A(d->alloc(&d->out_b, 1));
  snprintf(nm, sizeof nm, "decoder.model.%d", c.num_rates + 2);
  d->required.insert(std::string(nm) + ".weight");
  d->required.insert(std::string(nm) + ".bias");
  A(d->alloc(&d->table, (size_t)c.num_codebooks * c.codebook_size * c.latent_dim));
  d->cb.resize(c.num_codebooks); d->opw.resize(c.num_codebooks); d->opb.resize(c.num_codebooks);
  for (int i = 0; i < c.num_codebooks; ++i) {
    A(d->alloc(&d->cb[i], (size_t)c.codebook_size * c.codebook_dim));
    A(d->alloc(&d->opw[i], (size_t)c.latent_dim * c.codebook_dim));
    A(d->alloc(&d->opb[i], c.latent_dim));
    snprintf(nm, sizeof nm, "quantizer.quantizers.%d.", i);
    d->required.insert(std::string(nm) + "codebook.weight");
    d->required.insert(std::string(nm) + "out_proj.weight");
    d->required.insert(std::string(nm) + "out_proj.bias");
  }
  if (c.encoder_dim > 0) {  // encode side: descript Encoder, strides = the decoder's reversed
    target = &d->enc_convs;
    const int n = c.num_rates;
    int dim = c.encoder_dim;
    A(d->alloc(&d->in0_w, (size_t)dim * 7)); A(d->alloc(&d->in0_b, dim)); A(d->alloc(&d->in0_alpha, 2 * (size_t)dim));
    d->required.insert("encoder.block.0.weight"); d->required.insert("encoder.block.0.bias");
    d->required.insert("encoder.block.1.block.0.block.0.alpha");
    for (int bi = 0; bi < n; ++bi) {
      const int st = c.rates[n - 1 - bi];
      const int dils[3] = {1, 3, 9};
      for (int ri = 0; ri < 3; ++ri) {
        snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.1", bi + 1, ri);
        snprintf(an, sizeof an, "encoder.block.%d.block.%d.block.2", bi + 1, ri);
        A(add_conv(nm, an, dim, dim, 7, dils[ri], 1, false, false, false));
        snprintf(nm, sizeof nm, "encoder.block.%d.block.%d.block.3", bi + 1, ri);
        if (ri < 2) snprintf(an, sizeof an, "encoder.block.%d.block.%d.block.0", bi + 1, ri + 1);
        else snprintf(an, sizeof an, "encoder.block.%d.block.3", bi + 1);
        A(add_conv(nm, an, dim, dim, 1, 1, 1, false, true, ri < 2));
      }
      snprintf(nm, sizeof nm, "encoder.block.%d.block.4", bi + 1);
      if (bi + 1 < n) snprintf(an, sizeof an, "encoder.block.%d.block.0.block.0", bi + 2);
      else snprintf(an, sizeof an, "encoder.block.%d", n + 1);
      A(add_conv(nm, an, dim, 2 * dim, 2 * st, 1, st, false, false, true));
      d->enc_convs.back().pad = (st + 1) / 2;
      dim *= 2;
    }
    snprintf(nm, sizeof nm, "encoder.block.%d", n + 2);
    A(add_conv(nm, "", dim, c.latent_dim, 3, 1, 1, false, false, true));
    const size_t K = c.num_codebooks;
    A(d->alloc(&d->ipw, K * c.codebook_dim * c.latent_dim)); A(d->alloc(&d->ipb, K * c.codebook_dim));
    A(d->alloc(&d->cb_all, K * c.codebook_size * c.codebook_dim)); A(d->alloc(&d->cbn, K * c.codebook_size * c.codebook_dim));
    A(d->alloc(&d->cbn_sq, K * c.codebook_size));
    A(d->alloc(&d->opw_all, K * c.latent_dim * c.codebook_dim)); A(d->alloc(&d->opb_all, K * c.latent_dim));
    for (int i = 0; i < c.num_codebooks; ++i) {
      snprintf(nm, sizeof nm, "quantizer.quantizers.%d.in_proj.", i);
      d->required.insert(std::string(nm) + "weight");
      d->required.insert(std::string(nm) + "bias");
    }
    target = &d->convs;
  }
  // activation buffers: max over layers of T_l * C_l
  size_t mx = (size_t)c.max_frames * std::max(c.latent_dim, c.decoder_dim);
  if (c.encoder_dim > 0) mx = std::max(mx, (size_t)c.max_frames * hop * c.encoder_dim);
  {
    size_t T = c.max_frames;
    for (int bi = 0; bi < c.num_rates; ++bi) { T *= c.rates[bi]; mx = std::max(mx, T * (size_t)(ch >> (bi + 1))); }
  }
  mx *= c.max_batch;
  A(d->alloc(&d->bufA0, mx)); A(d->alloc(&d->bufA1, mx)); A(d->alloc(&d->bufY, mx)); A(d->alloc(&d->bufS, mx));
  A(d->alloc(&d->bufZ, (size_t)c.max_batch * c.max_frames * c.latent_dim));
  A(d->alloc(&d->d_ktap, MAXTAPS * 8));
#undef A
  *out = d;
  return PTTS_OK;
}

extern "C" int ptts_dac_load_weight(ptts_dac* d, const char* name_c, const float* dev_ptr, const int64_t* shape, int32_t ndim, void* stream) {
  PTTS_CHECK(d && name_c && dev_ptr && shape, PTTS_E_INVALID, "null argument");
  PTTS_DEVICE(d->cfg.device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // NULL = legacy default stream
  const ptts_dac_config& c = d->cfg;
  const std::string name(name_c);
  auto numel = [&]() { size_t n = 1; for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i]; return n; };
  auto copy = [&](float* dst, size_t n) -> int {
    if (numel() != n) return ptts_fail(PTTS_E_INVALID, "%s: expected %zu elements, got %zu", name_c, n, numel());
    PTTS_HIP(hipMemcpyAsync(dst, dev_ptr, n * 4, hipMemcpyDeviceToDevice, st));
    d->loaded.insert(name);
    return PTTS_OK;
  };
  int qi = -1;
  char tail[64] = {0};
  if (sscanf(name_c, "quantizer.quantizers.%d.%63s", &qi, tail) == 2) {
    PTTS_CHECK(qi >= 0 && qi < c.num_codebooks, PTTS_E_INVALID, "%s: quantizer index out of range", name_c);
    d->table_ready = false;
    d->vq_ready = false;
    const std::string t(tail);
    if (t == "codebook.weight") return copy(d->cb[qi], (size_t)c.codebook_size * c.codebook_dim);
    if (t == "out_proj.weight") return copy(d->opw[qi], (size_t)c.latent_dim * c.codebook_dim);
    if (t == "out_proj.bias") return copy(d->opb[qi], c.latent_dim);
    if (t == "in_proj.weight") { d->vq_ready = false; return c.encoder_dim > 0 ? copy(d->ipw + (size_t)qi * c.codebook_dim * c.latent_dim, (size_t)c.codebook_dim * c.latent_dim) : PTTS_OK; }
    if (t == "in_proj.bias") return c.encoder_dim > 0 ? copy(d->ipb + (size_t)qi * c.codebook_dim, c.codebook_dim) : PTTS_OK;
    return ptts_fail(PTTS_E_INVALID, "unknown tensor name %s", name_c);
  }
  char fin[64];
  snprintf(fin, sizeof fin, "decoder.model.%d", c.num_rates + 2);
  if (name == std::string(fin) + ".weight") {  // [1][C][7] -> [7][C]
    PTTS_CHECK(ndim == 3 && shape[0] == 1 && shape[1] == d->out_C && shape[2] == 7, PTTS_E_INVALID, "%s: expected [1,%d,7]", name_c, d->out_C);
    // transpose on device via strided copies: 7 rows
    for (int k = 0; k < 7; ++k)
      PTTS_HIP(hipMemcpy2DAsync(d->out_w + (size_t)k * d->out_C, 4, dev_ptr + k, 7 * 4, 4, d->out_C, hipMemcpyDeviceToDevice, st));
    d->loaded.insert(name);
    return PTTS_OK;
  }
  if (name == std::string(fin) + ".bias") return copy(d->out_b, 1);
  if (c.encoder_dim > 0) {
    if (name == "encoder.block.0.weight") return copy(d->in0_w, (size_t)c.encoder_dim * 7);
    if (name == "encoder.block.0.bias") return copy(d->in0_b, c.encoder_dim);
    if (name == "encoder.block.1.block.0.block.0.alpha") {
      PTTS_TRY(copy(d->in0_alpha, c.encoder_dim));
      hipLaunchKernelGGL(inv_alpha_kernel, dim3((c.encoder_dim + 255) / 256), dim3(256), 0, st, d->in0_alpha, c.encoder_dim);
      return PTTS_OK;
    }
  }
  std::vector<ConvLayer*> all;
  for (ConvLayer& L : d->convs) all.push_back(&L);
  for (ConvLayer& L : d->enc_convs) all.push_back(&L);
  for (ConvLayer* Lp : all) {
    ConvLayer& L = *Lp;
    if (!L.alpha_name.empty() && name == L.alpha_name + ".alpha") {
      PTTS_TRY(copy(L.alpha, L.Cout));
      hipLaunchKernelGGL(inv_alpha_kernel, dim3((L.Cout + 255) / 256), dim3(256), 0, st, L.alpha, L.Cout);
      return PTTS_OK;
    }
    if (name == L.name + ".bias") return copy(L.bias, L.Cout);
    if (name == L.name + ".weight") {
      const int k = L.ksize;
      int ktap[MAXTAPS * 8];
      memset(ktap, 0, sizeof ktap);  // a plain conv (one phase) may use up to 16 entries of row 0/1
      long long s_co, s_ci, s_k = 1;
      int ntaps, nphase;
      if (!L.transposed) {  // torch Conv1d weight [Cout][Cin][k]
        PTTS_CHECK(ndim == 3 && shape[0] == L.Cout && shape[1] == L.Cin && shape[2] == k, PTTS_E_INVALID, "%s: expected [%d,%d,%d]", name_c, L.Cout, L.Cin, k);
        s_co = (long long)L.Cin * k; s_ci = k; ntaps = k; nphase = 1;
        for (int t = 0; t < k; ++t) ktap[t] = t;
      } else {  // torch ConvTranspose1d weight [Cin][Cout][2s]
        PTTS_CHECK(ndim == 3 && shape[0] == L.Cin && shape[1] == L.Cout && shape[2] == k, PTTS_E_INVALID, "%s: expected [%d,%d,%d]", name_c, L.Cin, L.Cout, k);
        s_ci = (long long)L.Cout * k; s_co = k; ntaps = 2; nphase = L.stride;
        const int s = L.stride, pad = (s + 1) / 2;
        for (int ph = 0; ph < s; ++ph) { const int r = (ph + pad) % s; ktap[ph * MAXTAPS + 0] = r; ktap[ph * MAXTAPS + 1] = r + s; }
      }
      PTTS_HIP(hipMemcpyAsync(d->d_ktap, ktap, sizeof ktap, hipMemcpyHostToDevice, st));
      const size_t total = (size_t)nphase * (L.Cout / 16) * ntaps * (L.Cin / 16) * 64;
      if (L.bf16) {
        const size_t tb = (size_t)nphase * (L.Cout / 16) * ntaps * (L.Cin / 32) * 64;
        hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, st, dev_ptr, reinterpret_cast<bf16_t*>(L.Wp), L.Cout,
                           L.Cin, ntaps, nphase, d->d_ktap, s_co, s_ci, s_k);
      } else {
        hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dev_ptr, L.Wp, L.Cout, L.Cin, ntaps, nphase,
                           d->d_ktap, s_co, s_ci, s_k);
      }
      PTTS_HIP(hipStreamSynchronize(st));  // d_ktap is reused by the next call
      d->loaded.insert(name);
      return PTTS_OK;
    }
  }
  if (name.rfind("encoder.", 0) == 0) return PTTS_OK;  // DAC encoder weights are not on the decode path
  return ptts_fail(PTTS_E_INVALID, "unknown tensor name %s", name_c);
}

extern "C" int ptts_dac_weights_ready(ptts_dac* d) {
  PTTS_CHECK(d, PTTS_E_INVALID, "null dac");
  std::string missing;
  int n = 0;
  for (const auto& r : d->required)
    if (!d->loaded.count(r)) { if (n++ < 8) missing += (missing.empty() ? "" : ", ") + r; }
  if (n) return ptts_fail(PTTS_E_MISSING, "%d tensors not loaded: %s%s", n, missing.c_str(), n > 8 ? ", ..." : "");
  return PTTS_OK;
}

static int run_conv(ptts_dac* d, const ConvLayer& L, const void* x, const float* skip, float* out_raw, void* out_act, int B, int Tin, hipStream_t st,
                    bool act_f32 = false, const int* lens = nullptr, int len_mul = 1) {
  ConvArgs a = {};
  a.act_f32 = act_f32 ? 1 : 0;
  a.lens = lens; a.len_mul = len_mul;
  a.x = x; a.Wp = L.Wp; a.bias = L.bias; a.skip = skip; a.out_raw = out_raw; a.out_act = out_act; a.alpha = L.alpha;
  a.B = B; a.Tin = Tin; a.Cin = L.Cin; a.Cout = L.Cout;
  if (!L.transposed) {
    a.ntaps = L.ksize; a.nphase = 1; a.dil = L.dil; a.pad = L.pad >= 0 ? L.pad : (L.ksize - 1) * L.dil / 2; a.transposed = 0;
    a.stride = L.stride; a.Tn = L.stride > 1 ? (Tin + 2 * a.pad - L.ksize) / L.stride + 1 : Tin;
  } else {  // to = j*s + ph = ti*s - pad + k  =>  tap 0 (k = r): ti = j + c0 ; tap 1 (k = r + s): ti = j + c0 - 1, c0 = (ph + pad)/s
    a.ntaps = 2; a.nphase = L.stride; a.dil = 1; a.pad = (L.stride + 1) / 2; a.transposed = 1;
    a.stride = 1; a.Tn = Tin;
  }
  const int nstrips = L.Cout / 16;
  // LDS-tiled kernel where it wins (rocprof per layer, profiles/r02_dac_layers.txt): every k7 conv (2x the direct kernel) and the
  // transposed convs into >= 192 channels. The last transposed conv is bound by its epilogue traffic (two writes per element) and runs as fast
  // on the direct kernel's 4-5 waves per SIMD as on this one's 2 (round 2 said the same of the k1 convs; see lds_k1 below).
  static const bool no_lds = ptts_dev_env("PTTS_DAC_NO_LDS") != nullptr;
  static const int lds_min_c = ptts_dev_env("PTTS_DAC_LDS_MIN_C") ? atoi(ptts_dev_env("PTTS_DAC_LDS_MIN_C")) : 96;
  static const bool lds_small_taps = ptts_dev_env("PTTS_DAC_LDS_K1") != nullptr;
  // k1 convs (the un-fused units of the C = 768 block): on the LDS-tiled kernel from 32 K rows per launch (round 5) - with the whole-row epilogue
  // through LDS it beats the direct kernel at batch 32 (63.1 -> 61.3 ms per decode) and loses 1 % on a single utterance's 6880 rows (2.69 vs 2.71 ms;
  // profiles/r05_experiments.txt calls 13 / 15). PTTS_DAC_NO_LDS_K1=1: always direct; PTTS_DAC_LDS_K1=1: every k1 / transposed conv on the LDS kernel.
  static const bool lds_k1_on = !(ptts_dev_env("PTTS_DAC_NO_LDS_K1") && atoi(ptts_dev_env("PTTS_DAC_NO_LDS_K1")));
  const bool lds_k1 = lds_k1_on && (long long)B * Tin >= 32768;
  // the last transposed conv (-> 96 channels): on the LDS-tiled kernel since round 5. Round 4 measured it equal to the direct kernel (5.19 ms per
  // batch-32 launch) - with the 96-channel staging instance, which needs 324 VGPRs and ran one wave per SIMD; with 32-channel staging chunks
  // (conv_lds_kernel<3, 2, 1, 2>, 228 VGPRs, two waves per SIMD) the batch-32 decode drops 61.3 -> 59.3 ms, 860 frames 2.71 -> 2.67 ms
  // (profiles/r05_experiments.txt call 15). PTTS_DAC_LAST_UP_LDS=0: the direct kernel.
  static const bool last_up_lds = !(ptts_dev_env("PTTS_DAC_LAST_UP_LDS") && !atoi(ptts_dev_env("PTTS_DAC_LAST_UP_LDS")));
  const bool lds_ok = a.ntaps > 2 ? L.Cout >= lds_min_c
                                  : (lds_small_taps || (a.transposed && L.Cout >= (last_up_lds ? 96 : 192)) || (lds_k1 && !a.transposed && a.ntaps == 1));
  {
    const char* ced = ptts_dev_env("PTTS_DAC_EPI_DIRECT");  // read per call (A/B inside one process): the same switch as the fused residual unit's epilogue
    a.epi_direct = (ced && atoi(ced)) ? 1 : 0;
  }
  if (L.bf16 && !no_lds && lds_ok && a.stride == 1 && nstrips % 6 == 0) {
    const int nw = nstrips % 12 == 0 ? 4 : 2;
    const int halo = a.transposed ? a.ntaps - 1 : (a.ntaps - 1) * a.dil;
    // 64-frame tiles (FT = 4) where 128-frame tiles would leave the launch with about one workgroup per CU or fewer (round 5: the first block of a
    // single utterance ran 56-224 workgroups on 256 CUs; the streamer's short windows even fewer): 860 frames 2.34 -> 2.27 ms, a 56-frame window
    // 0.91 -> 0.82 ms; at 432-448 workgroups (two utterances) the smaller tiles LOSE 3.5 % (profiles/r05_experiments.txt call 18), hence the bound.
    // Four-wave instances only; PTTS_DAC_NO_FT4=1: always 128 frames.
    static const bool ft4_on = !(ptts_dev_env("PTTS_DAC_NO_FT4") && atoi(ptts_dev_env("PTTS_DAC_NO_FT4")));
    const bool ft4 = ft4_on && nw == 4 && (long long)((a.Tn + 127) / 128) * a.nphase * B * (nstrips / (3 * nw)) < 320;
    const int tfr = ft4 ? 64 : 128;
    const dim3 grid((unsigned)(((a.Tn + tfr - 1) / tfr) * a.nphase * B), (unsigned)(nstrips / (3 * nw)));
    bool done = true;
    if (a.ntaps > 2 && halo <= 54 && a.Cin % 32 == 0) {
      if (nw == 4 && ft4) hipLaunchKernelGGL((conv_lds_kernel<3, 4, 1, 54, 4>), grid, dim3(256), 0, st, a);
      else if (nw == 4) hipLaunchKernelGGL((conv_lds_kernel<3, 4, 1, 54>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv_lds_kernel<3, 2, 1, 54>), grid, dim3(128), 0, st, a);
    } else if (a.ntaps <= 2 && a.Cin % 96 == 0 && nw == 4 && ft4) {
      hipLaunchKernelGGL((conv_lds_kernel<3, 4, 3, 2, 4>), grid, dim3(256), 0, st, a);
    } else if (a.ntaps <= 2 && a.Cin % 96 == 0) {
      // two-wave workgroups (Cout = 96: the last transposed conv): 32-channel staging chunks - a 96-channel instance staged 13 16-byte pieces per
      // thread, needed 324 VGPRs and ran ONE wave per SIMD (round 5: 5.59 vs 3.35 ms per batch-32 launch; the instance is gone since round 6)
      if (nw == 4) hipLaunchKernelGGL((conv_lds_kernel<3, 4, 3, 2>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((conv_lds_kernel<3, 2, 1, 2>), grid, dim3(128), 0, st, a);
    } else {
      done = false;
    }
    if (done) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "conv launch failed: %s", hipGetErrorString(e));
      return PTTS_OK;
    }
  }
  // waves per workgroup: fewer (finer tiles) when the launch would otherwise put < ~6 workgroups on each CU, so the
  // 256 CUs finish together (324 four-wave workgroups on 256 CUs = 63 % balance; 1296 one-wave ones = 84 %+)
  static int forced_nw = ptts_dev_env("PTTS_DAC_WAVES") ? atoi(ptts_dev_env("PTTS_DAC_WAVES")) : 0;
  int nwb = 4;
  {
    const int CS0 = nstrips % 8 == 0 ? 8 : (nstrips % 6 == 0 ? 6 : (nstrips % 4 == 0 ? 4 : (nstrips % 2 == 0 ? 2 : 1)));
    auto blocks = [&](int nw) { return (long long)((a.Tn + 32 * nw - 1) / (32 * nw)) * a.nphase * B * (nstrips / CS0); };
    while (nwb > 1 && blocks(nwb) < 256LL * 6) nwb >>= 1;
    if (forced_nw == 1 || forced_nw == 2 || forced_nw == 4) nwb = forced_nw;
  }
  const int ntile = (a.Tn + 32 * nwb - 1) / (32 * nwb);
  // strips per wave: the largest of {8, 6, 4, 2, 1} that divides the layer (real DAC widths: 96/48/24/12/6 strips)
  const int CS = nstrips % 8 == 0 ? 8 : (nstrips % 6 == 0 ? 6 : (nstrips % 4 == 0 ? 4 : (nstrips % 2 == 0 ? 2 : 1)));
  const dim3 grid((unsigned)(ntile * a.nphase * B), (unsigned)(nstrips / CS));
  const dim3 blk(64 * nwb);
  if (L.bf16) {
    if (CS == 8) hipLaunchKernelGGL((conv_mfma_kernel<8, true>), grid, blk, 0, st, a);
    else if (CS == 6) hipLaunchKernelGGL((conv_mfma_kernel<6, true>), grid, blk, 0, st, a);
    else if (CS == 4) hipLaunchKernelGGL((conv_mfma_kernel<4, true>), grid, blk, 0, st, a);
    else if (CS == 2) hipLaunchKernelGGL((conv_mfma_kernel<2, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_mfma_kernel<1, true>), grid, blk, 0, st, a);
  } else {
    if (CS == 8) hipLaunchKernelGGL((conv_mfma_kernel<8, false>), grid, blk, 0, st, a);
    else if (CS == 6) hipLaunchKernelGGL((conv_mfma_kernel<6, false>), grid, blk, 0, st, a);
    else if (CS == 4) hipLaunchKernelGGL((conv_mfma_kernel<4, false>), grid, blk, 0, st, a);
    else if (CS == 2) hipLaunchKernelGGL((conv_mfma_kernel<2, false>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_mfma_kernel<1, false>), grid, blk, 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "conv launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// one residual unit (k7 dilated conv -> Snake -> k1 conv -> + skip -> Snake) in one launch; out_act must not be x's buffer
static bool resunit_fusable(const ConvLayer& c7, const ConvLayer& c1) {
  const char* ev = getenv("PTTS_DAC_NO_FUSE_RES");  // read per call: a test can switch it inside one process
  const int off = ev ? atoi(ev) : 0;  // 1: every unit as two launches; 2: only the C = 384 block's units (round 3's baseline)
  const bool on = off != 1;
  // the C = 384 block's units too (8 waves, 100 KB of LDS, one workgroup per CU): batch 32 83.3 -> 77.9 ms, 860 frames 3.39 -> 3.28 ms
  // (profiles/r03_pmc_dac_mfma.txt)
  const bool fuse384 = off != 2;
  return on && c7.bf16 && c1.bf16 && !c7.transposed && !c1.transposed && c7.stride == 1 && c1.stride == 1 && c7.ksize == 7 && c1.ksize == 1 &&
         c7.Cin == c7.Cout && c1.Cin == c7.Cout && c1.Cout == c7.Cout && (c7.Cout == 192 || c7.Cout == 96 || (c7.Cout == 384 && fuse384)) && 6 * c7.dil <= 54 &&
         c7.alpha && c1.alpha;
}
// Which widths run their residual units as XIN units (input = the fp32 stream, see resunit_lds_kernel): bit 0: C = 96, bit 1: C = 192, bit 2: C = 384.
// Measured on MI355X (profiles/r06_dac_xin_ab.txt, r06_dac_up_order_ab.txt; ms per decode of 860 frames, mask 0 | 1 | 3 | 7): 32 utterances 56.16 | 54.86 |
// 54.35 | 53.99, 8: - | 14.30 | - | 14.13, 4: - | 7.69 | - | 7.55, one utterance 2.409 | 2.316 | 2.335 | 2.349 - the wider units gain only where the launch
// is bandwidth-bound, hence by the latent frames of the call. PTTS_DAC_XIN=<mask> is read per call (a test switches it inside one process: every mask
// gives the same bits); the round-3 epilogue (PTTS_DAC_EPI_DIRECT=1, an A/B path) exists for the units on the bf16 activation only.
static bool resunit_xin_width(int C, long long frames) {
  const char* ed = ptts_dev_env("PTTS_DAC_EPI_DIRECT");
  if (ed && atoi(ed)) return false;
  const char* ev = getenv("PTTS_DAC_XIN");
  const int m = ev ? atoi(ev) : (frames >= 2 * 860 ? 7 : 1);
  return (C == 96 && (m & 1)) || (C == 192 && (m & 2)) || (C == 384 && (m & 4));
}
// `alpha_in` != null: an XIN unit - x is the fp32 stream (= skip) and alpha_in the [alpha | 1 / alpha] of the Snake in front of the unit; out_raw must
// then be a DIFFERENT buffer (neighbouring workgroups read x's halo rows). out_act may be null for a unit whose consumer is an XIN unit.
static int run_resunit(const ConvLayer& c7, const ConvLayer& c1, const void* x, const float* skip, float* out_raw, void* out_act, int B, int T,
                       hipStream_t st, bool act_f32, const int* lens = nullptr, int len_mul = 1, const float* alpha_in = nullptr) {
  ResArgs r = {};
  r.a.lens = lens; r.a.len_mul = len_mul;
  r.a.x = x; r.a.Wp = c7.Wp; r.a.bias = c7.bias; r.a.alpha = c7.alpha; r.a.dil = c7.dil; r.a.pad = (c7.ksize - 1) * c7.dil / 2;
  r.a.B = B; r.a.Tin = T; r.a.Tn = T; r.a.Cin = c7.Cin; r.a.Cout = c7.Cout; r.a.ntaps = 7; r.a.nphase = 1; r.a.stride = 1;
  r.Wp1 = c1.Wp; r.bias1 = c1.bias; r.alpha1 = c1.alpha; r.skip = skip; r.out_raw = out_raw; r.out_act = out_act; r.act_f32 = act_f32 ? 1 : 0;
  r.alpha_in = alpha_in;
  {
    const char* ed = ptts_dev_env("PTTS_DAC_EPI_DIRECT");  // read per call (A/B inside one process, like PTTS_DAC_NO_FUSE_RES)
    r.epi_direct = (ed && atoi(ed)) ? 1 : 0;
  }
  const dim3 grid((unsigned)(((T + 127) / 128) * B));
  // instances by (width, stream written?, fp32 activation?, input from the stream?, activation written?): compile-time in the kernel (its epilogue is
  // straight-line code). The two-register-set weight prefetch (WD = 1, round 3) serves the XIN instances at C = 96 (registers: see the kernel).
  const bool raw = out_raw != nullptr, xin = alpha_in != nullptr, act = out_act != nullptr;
  if (!act && !(xin && raw)) return ptts_fail(PTTS_E_INVALID, "residual unit: no activation output");
  if (xin && ((const void*)skip != x || (const void*)out_raw == x || r.epi_direct))
    return ptts_fail(PTTS_E_INVALID, "residual unit on the stream: x must be the skip buffer, out_raw another one, whole-row epilogue");
  if (act_f32 && c7.Cout != 96) return ptts_fail(PTTS_E_UNSUPPORTED, "residual unit: fp32 activations only at the last block's width");
#define PTTS_RU_LAUNCH(NWV, RAWV, F32V) \
  hipLaunchKernelGGL((resunit_lds_kernel<NWV, 1, 3, RAWV, F32V>), grid, dim3(NWV * 64), (ResunitLds<NWV, 1>::bytes), st, r)
#define PTTS_RU_LAUNCH_XIN(NWV, RAWV, F32V, ACTV) \
  hipLaunchKernelGGL((resunit_lds_kernel<NWV, 1, ResunitXinWD<NWV>::value, RAWV, F32V, true, ACTV>), grid, dim3(NWV * 64), (ResunitLds<NWV, 1>::bytes), st, r)
  // (stream written, activation written) of an XIN unit: (1, 0) units 1 and 2 of a block, (0, 1) unit 3, (1, 1) the parity probe's stop stage
#define PTTS_RU_XIN_BY_OUT(NWV, F32V)                                   \
  do {                                                                  \
    if (raw && !act) PTTS_RU_LAUNCH_XIN(NWV, true, false, false);       \
    else if (raw) PTTS_RU_LAUNCH_XIN(NWV, true, F32V, true);            \
    else PTTS_RU_LAUNCH_XIN(NWV, false, F32V, true);                    \
  } while (0)
  if (c7.Cout == 384) {
    static PttsPerDeviceOnce attr_once;  // 100 KB of dynamic LDS needs the opt-in
    const int attr_dev = PttsPerDeviceOnce::device();
    if (attr_once.need(attr_dev)) {
      constexpr int WDX = ResunitXinWD<8>::value;
      const void* fns[] = {reinterpret_cast<const void*>(&resunit_lds_kernel<8, 1, 3, true, false>), reinterpret_cast<const void*>(&resunit_lds_kernel<8, 1, 3, false, false>),
                           reinterpret_cast<const void*>(&resunit_lds_kernel<8, 1, WDX, true, false, true, false>),
                           reinterpret_cast<const void*>(&resunit_lds_kernel<8, 1, WDX, true, false, true, true>),
                           reinterpret_cast<const void*>(&resunit_lds_kernel<8, 1, WDX, false, false, true, true>)};
      for (const void* fn : fns) {
        const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ResunitLds<8, 1>::bytes);
        if (ea != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(ea));
      }
      attr_once.done(attr_dev);
    }
    if (xin) PTTS_RU_XIN_BY_OUT(8, false);
    else if (raw) PTTS_RU_LAUNCH(8, true, false); else PTTS_RU_LAUNCH(8, false, false);
  } else if (c7.Cout == 192) {
    if (xin) PTTS_RU_XIN_BY_OUT(4, false);
    else if (raw) PTTS_RU_LAUNCH(4, true, false); else PTTS_RU_LAUNCH(4, false, false);
  } else if (act_f32) {
    if (xin) PTTS_RU_XIN_BY_OUT(2, true);
    else if (raw) PTTS_RU_LAUNCH(2, true, true); else PTTS_RU_LAUNCH(2, false, true);
  } else {
    if (xin) PTTS_RU_XIN_BY_OUT(2, false);
    else if (raw) PTTS_RU_LAUNCH(2, true, false); else PTTS_RU_LAUNCH(2, false, false);
  }
#undef PTTS_RU_XIN_BY_OUT
#undef PTTS_RU_LAUNCH_XIN
#undef PTTS_RU_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "residual-unit launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// parity probe (ptts_dac_debug_decode_upto): the buffers holding a stage's outputs when the decode stops there
struct DacDebugTap { void* act; int act_is_bf16; float* raw; int rows, channels; };

// decode the window [t0, t0 + T) of codes rows with stride `ld`; samples [skip, hop*T) of the window go to wave_dev rows of out_ld
static int dac_decode_window(ptts_dac* d, const int64_t* codes_dev, long long ld, int t0, float* wave_dev, int skip, long long out_ld,
                             int32_t B, int32_t T, void* stream, int emit = -1, const int* lens = nullptr, int stop_stage = -1,
                             DacDebugTap* tap = nullptr) {
  PTTS_TRY(ptts_dac_weights_ready(d));
  const ptts_dac_config& c = d->cfg;
  PTTS_CHECK(B >= 1 && B <= c.max_batch, PTTS_E_CAPACITY, "batch %d exceeds dac max_batch %d", B, c.max_batch);
  PTTS_CHECK(T >= 1 && T <= c.max_frames, PTTS_E_CAPACITY, "frames %d exceed dac max_frames %d", T, c.max_frames);
  PTTS_DEVICE(c.device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // NULL = legacy default stream
  if (!d->table_ready) {
    for (int i = 0; i < c.num_codebooks; ++i) {
      const size_t n = (size_t)c.codebook_size * c.latent_dim;
      hipLaunchKernelGGL(rvq_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d->cb[i], d->opw[i], d->opb[i],
                         d->table + (size_t)i * n, c.codebook_size, c.codebook_dim, c.latent_dim);
    }
    d->table_ready = true;
  }
  const bool bf = c.compute_dtype == PTTS_BF16;
  hipLaunchKernelGGL(rvq_gather_kernel, dim3(T, B), dim3(256), 0, st, (const long long*)codes_dev, d->table, (void*)d->bufZ, c.num_codebooks, T,
                     c.codebook_size, c.latent_dim, bf ? 1 : 0, ld, t0, lens);
  float *cur = d->bufA0, *other = d->bufA1;
  int Tcur = T, mul = 1;  // mul: rows per latent frame at the current layer (ragged decode: utterance b has lens[b] * mul valid rows)
  size_t li = 0;
  PTTS_TRY(run_conv(d, d->convs[li++], d->bufZ, nullptr, nullptr, cur, B, Tcur, st, false, lens, mul));
  int stage = 0;  // 0 = decoder.model.0; per block: the transposed conv, then its three residual units
  const bool dbg = stop_stage >= 0;
  auto stop_here = [&](float* raw, int ch, bool act_f32) {
    if (stage++ != stop_stage) return false;
    if (tap) { tap->act = cur; tap->act_is_bf16 = (bf && !act_f32) ? 1 : 0; tap->raw = raw; tap->rows = Tcur; tap->channels = ch; }
    return true;
  };
  if (stop_here(nullptr, d->convs[0].Cout, false)) return PTTS_OK;
  for (int bi = 0; bi < c.num_rates; ++bi) {
    const ConvLayer& up = d->convs[li++];
    // XIN block (round 6): all three units fused and this width enabled - the units read the fp32 stream (bufY / bufS in turns; bufS is otherwise only
    // the un-fused units' y) and evaluate the Snake in front of them on the way into LDS; nobody writes a bf16 activation but the block's last unit
    // (and the parity probe's stop stage, whose outputs the test reads). Same values, bit for bit.
    bool xin = up.alpha != nullptr && resunit_xin_width(up.Cout, (long long)B * T);
    for (int ri = 0; ri < 3 && xin; ++ri) xin = resunit_fusable(d->convs[li + 2 * ri], d->convs[li + 2 * ri + 1]);
    const bool up_act = !xin || (dbg && stage == stop_stage);
    PTTS_TRY(run_conv(d, up, cur, nullptr, d->bufY, up_act ? other : nullptr, B, Tcur, st, false, lens, mul));
    std::swap(cur, other);
    Tcur *= up.stride;
    mul *= up.stride;
    if (stop_here(d->bufY, up.Cout, false)) return PTTS_OK;
    float *s_in = d->bufY, *s_out = d->bufS;  // XIN: the stream before / after the next unit
    const float* alpha_in = up.alpha;
    for (int ri = 0; ri < 3; ++ri) {
      const ConvLayer& c7 = d->convs[li++];
      const ConvLayer& c1 = d->convs[li++];
      const bool last = bi + 1 == c.num_rates && ri == 2;  // feeds the final Conv1d(C -> 1): fp32 activations
      if (xin) {
        const bool want_raw = ri < 2 || dbg, want_act = ri == 2 || (dbg && stage == stop_stage);
        PTTS_TRY(run_resunit(c7, c1, s_in, s_in, want_raw ? s_out : nullptr, want_act ? other : nullptr, B, Tcur, st, last, lens, mul, alpha_in));
        if (want_act) std::swap(cur, other);
        if (want_raw) std::swap(s_in, s_out);
        alpha_in = c1.alpha;
        if (stop_here(s_in, c1.Cout, last)) return PTTS_OK;
        continue;
      }
      float* raw_out = (c1.write_raw || dbg) ? d->bufY : nullptr;  // the parity probe also wants the stream after a block's last unit (same arithmetic, one more store)
      if (resunit_fusable(c7, c1)) {  // both convs in one launch; the output goes to the OTHER activation buffer
        PTTS_TRY(run_resunit(c7, c1, cur, d->bufY, raw_out, other, B, Tcur, st, last, lens, mul));
        std::swap(cur, other);
      } else {
        PTTS_TRY(run_conv(d, c7, cur, nullptr, nullptr, d->bufS, B, Tcur, st, false, lens, mul));
        PTTS_TRY(run_conv(d, c1, d->bufS, d->bufY, raw_out, cur, B, Tcur, st, last, lens, mul));
      }
      if (stop_here(d->bufY, c1.Cout, last)) return PTTS_OK;
    }
  }
  const int t_end = emit < 0 ? Tcur : std::min(Tcur, skip + emit);
  static const bool out_direct = ptts_dev_env("PTTS_DAC_OUT_DIRECT") && atoi(ptts_dev_env("PTTS_DAC_OUT_DIRECT"));  // A/B: the per-thread kernel
  if (d->out_C == OUT_C && !out_direct && B <= 65535) {
    hipLaunchKernelGGL(conv_out_tanh_lds_kernel, dim3((unsigned)((Tcur + OUT_TILE - 1) / OUT_TILE), (unsigned)B), dim3(OUT_TILE), 0, st, (const float*)cur,
                       d->out_w, d->out_b, wave_dev, Tcur, skip, out_ld, t_end, lens, mul);
  } else {
    const size_t n = (size_t)B * ((Tcur + OUT_OS - 1) / OUT_OS);
    hipLaunchKernelGGL(conv_out_tanh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)7 * d->out_C * 4, st, cur, d->out_w, d->out_b,
                       wave_dev, B, Tcur, d->out_C, 7, skip, out_ld, t_end, lens, mul);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "dac launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

extern "C" int ptts_dac_decode(ptts_dac* d, const int64_t* codes_dev, float* wave_dev, int32_t B, int32_t T, void* stream) {
  PTTS_CHECK(d && codes_dev && wave_dev, PTTS_E_INVALID, "null argument");
  return dac_decode_window(d, codes_dev, T, 0, wave_dev, 0, (long long)d->hop * T, B, T, stream);
}

// Ragged batch (generate()'s per-sample branch, modeling_parler_tts.py:3615-3647: every utterance keeps its own number of frames after
// the special-id filter; the reference decodes them one by one and zero-pads): ONE pass over codes [B][K][T] in which utterance b is
// decoded as exactly frames_dev[b] frames - rows beyond its length read as the convolutions' zero padding, tiles beyond it exit at once,
// its samples beyond hop * frames_dev[b] are zero (the buffer is cleared first). frames_dev: int32 [B] ON THE DEVICE (no host round trip
// between the filter and the codec), values clamped to [0, T].
extern "C" int ptts_dac_decode_ragged(ptts_dac* d, const int64_t* codes_dev, const int32_t* frames_dev, float* wave_dev, int32_t B, int32_t T,
                                      void* stream) {
  PTTS_CHECK(d && codes_dev && frames_dev && wave_dev, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(B >= 1 && T >= 1, PTTS_E_INVALID, "bad batch / frames");
  {
    PTTS_DEVICE(d->cfg.device);
    PTTS_HIP(hipMemsetAsync(wave_dev, 0, (size_t)B * d->hop * T * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
  }
  return dac_decode_window(d, codes_dev, T, 0, wave_dev, 0, (long long)d->hop * T, B, T, stream, -1, frames_dev);
}

// The filter in front of it (modeling_parler_tts.py:3627-3636): per utterance, drop every frame (column) in which ANY codebook holds an
// id >= codebook_size (or < 0), keep the others in order. codes_in / codes_out int64 [B][K][T] (may NOT alias), frames_out int32 [B].
// One workgroup per utterance: keep flags -> block-wide exclusive scan over the T columns -> scatter; columns past the kept count are
// filled with 0 (a valid id: the ragged decode never reads them).
__global__ void __launch_bounds__(256) compact_codes_kernel(const long long* __restrict__ in, long long* __restrict__ out, int* __restrict__ frames,
                                                            int K, int T, int ncodes) {
  __shared__ int s_wsum[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long* src = in + (size_t)b * K * T;
  long long* dst = out + (size_t)b * K * T;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < T; c0 += 256) {
    const int t = c0 + tid;
    int keep = 0;
    if (t < T) {
      keep = 1;
      for (int k = 0; k < K; ++k) { const long long v = src[(size_t)k * T + t]; if (v < 0 || v >= ncodes) keep = 0; }
    }
    // exclusive scan of `keep` over the 256 columns of this chunk: wave ballot + 4 wave totals
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wsum[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w2 = 0; w2 < wave; ++w2) off += s_wsum[w2];
    if (keep) for (int k = 0; k < K; ++k) dst[(size_t)k * T + off + before] = src[(size_t)k * T + t];
    __syncthreads();
    if (tid == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
  }
  const int n = s_base;
  for (int i = tid; i < (T - n) * K; i += 256) dst[(size_t)(i / (T - n)) * T + n + i % (T - n)] = 0;
  if (tid == 0) frames[b] = n;
}

extern "C" int ptts_dac_compact_codes(ptts_dac* d, const int64_t* codes_in_dev, int64_t* codes_out_dev, int32_t* frames_out_dev, int32_t B, int32_t T,
                                      void* stream) {
  PTTS_CHECK(d && codes_in_dev && codes_out_dev && frames_out_dev, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(B >= 1 && T >= 1 && codes_in_dev != codes_out_dev, PTTS_E_INVALID, "bad batch / frames, or codes_out aliases codes_in");
  PTTS_DEVICE(d->cfg.device);
  hipLaunchKernelGGL(compact_codes_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const long long*)codes_in_dev,
                     (long long*)codes_out_dev, (int*)frames_out_dev, d->cfg.num_codebooks, T, d->cfg.codebook_size);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "compact launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// Streaming / chunked decode (parler_tts/streamer.py:66-131 re-decodes the WHOLE token cache every `play_steps`): samples of
// frames [first_frame, first_frame + n_frames) of codes [B][K][codes_ld], computed from the window that starts `halo` frames
// earlier (clamped at 0) and ends at first_frame + n_frames - exactly the samples a decode of frames [0, first_frame + n_frames)
// yields there whenever `halo` covers the decoder's one-sided receptive field (13 frames for strides 8, 8, 4, 2).
extern "C" int ptts_dac_decode_chunk(ptts_dac* d, const int64_t* codes_dev, int64_t codes_ld, int32_t first_frame, int32_t n_frames,
                                     int32_t halo, float* wave_dev, int64_t wave_ld, int32_t n_emit, int32_t B, void* stream) {
  PTTS_CHECK(d && codes_dev && wave_dev, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(first_frame >= 0 && n_frames >= 1 && halo >= 0 && (int64_t)first_frame + n_frames <= codes_ld, PTTS_E_INVALID,
             "bad chunk [%d, %d) of %lld frames (halo %d)", first_frame, first_frame + n_frames, (long long)codes_ld, halo);
  if (n_emit <= 0 || n_emit > n_frames) n_emit = n_frames;
  if (wave_ld <= 0) wave_ld = (int64_t)d->hop * n_emit;
  PTTS_CHECK(wave_ld >= (int64_t)d->hop * n_emit, PTTS_E_INVALID, "wave_ld %lld < %d emitted frames", (long long)wave_ld, n_emit);
  const int w0 = first_frame > halo ? first_frame - halo : 0;
  return dac_decode_window(d, codes_dev, codes_ld, w0, wave_dev, (first_frame - w0) * d->hop, wave_ld, B, first_frame + n_frames - w0, stream,
                           n_emit * d->hop);
}

// DACModel.encode (dac_wrapper/modeling_dac.py:33-104) for one chunk: wave_dev float32 [B][L] with L a multiple of the hop
// (the caller applies model.preprocess's right padding, :64) -> codes_dev int64 [B][nq][L / hop].
extern "C" int ptts_dac_encode(ptts_dac* d, const float* wave_dev, int64_t* codes_dev, int32_t B, int32_t L, int32_t n_quantizers, void* stream) {
  PTTS_CHECK(d && wave_dev && codes_dev, PTTS_E_INVALID, "null argument");
  const ptts_dac_config& c = d->cfg;
  PTTS_CHECK(c.encoder_dim > 0, PTTS_E_UNSUPPORTED, "this dac engine was created without the encoder (encoder_dim = 0)");
  PTTS_TRY(ptts_dac_weights_ready(d));
  PTTS_CHECK(B >= 1 && B <= c.max_batch, PTTS_E_CAPACITY, "batch %d exceeds dac max_batch %d", B, c.max_batch);
  PTTS_CHECK(L >= d->hop && L % d->hop == 0, PTTS_E_INVALID, "waveform length %d is not a positive multiple of the hop %d (apply preprocess padding)", L, d->hop);
  const int T = L / d->hop;
  PTTS_CHECK(T <= c.max_frames, PTTS_E_CAPACITY, "frames %d exceed dac max_frames %d", T, c.max_frames);
  const int nq = n_quantizers <= 0 || n_quantizers > c.num_codebooks ? c.num_codebooks : n_quantizers;
  PTTS_DEVICE(c.device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!d->vq_ready) {  // contiguous [K][...] copies for the search kernel + the normalised codebooks
    for (int i = 0; i < c.num_codebooks; ++i) {
      const size_t ncb = (size_t)c.codebook_size * c.codebook_dim, nw = (size_t)c.latent_dim * c.codebook_dim;
      PTTS_HIP(hipMemcpyAsync(d->cb_all + i * ncb, d->cb[i], ncb * 4, hipMemcpyDeviceToDevice, st));
      PTTS_HIP(hipMemcpyAsync(d->opw_all + i * nw, d->opw[i], nw * 4, hipMemcpyDeviceToDevice, st));
      PTTS_HIP(hipMemcpyAsync(d->opb_all + (size_t)i * c.latent_dim, d->opb[i], (size_t)c.latent_dim * 4, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(cb_normalize_kernel, dim3((c.codebook_size + 255) / 256), dim3(256), 0, st, d->cb_all + i * ncb, d->cbn + i * ncb,
                         d->cbn_sq + (size_t)i * c.codebook_size, c.codebook_size, c.codebook_dim);
    }
    d->vq_ready = true;
  }
  float *cur = d->bufA0, *other = d->bufA1;
  int Tcur = L;
  {
    const size_t n = (size_t)B * L * (c.encoder_dim / 4);
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wave_dev, d->in0_w, d->in0_b, d->in0_alpha, d->bufY, cur, B,
                       L, c.encoder_dim);
  }
  size_t li = 0;
  for (int bi = 0; bi < c.num_rates; ++bi) {
    for (int ri = 0; ri < 3; ++ri) {
      const ConvLayer& c7 = d->enc_convs[li++];
      PTTS_TRY(run_conv(d, c7, cur, nullptr, nullptr, d->bufS, B, Tcur, st));
      const ConvLayer& c1 = d->enc_convs[li++];
      PTTS_TRY(run_conv(d, c1, d->bufS, d->bufY, c1.write_raw ? d->bufY : nullptr, cur, B, Tcur, st));
    }
    const ConvLayer& down = d->enc_convs[li++];
    PTTS_TRY(run_conv(d, down, cur, nullptr, d->bufY, other, B, Tcur, st));
    std::swap(cur, other);
    Tcur = (Tcur + 2 * down.pad - down.ksize) / down.stride + 1;
  }
  PTTS_CHECK(Tcur == T, PTTS_E_INVALID, "encoder produced %d frames for %d expected", Tcur, T);
  PTTS_TRY(run_conv(d, d->enc_convs[li++], cur, nullptr, d->bufZ, nullptr, B, Tcur, st));
  hipLaunchKernelGGL(rvq_encode_kernel, dim3(T, B), dim3(256), (size_t)c.latent_dim * 4, st, d->bufZ, d->ipw, d->ipb, d->cb_all, d->cbn, d->cbn_sq,
                     d->opw_all, d->opb_all, (long long*)codes_dev, T, c.latent_dim, c.codebook_dim, c.codebook_size, nq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "dac encode launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// Debug / parity probe: ptts_dac_decode stopped after `stage` (0 = decoder.model.0; then per up-sampling block: its transposed conv, residual
// unit 1, 2, 3: 1 + 4 * num_rates stages in all). Hands out the engine's own buffers holding that stage's outputs (valid until the next call):
// act [B][rows][channels] = the Snake'd activation the next conv reads (bf16 bits if *act_is_bf16, else fp32), raw [B][rows][channels] fp32 =
// the residual stream (null for stage 0). tests/test_dac_stage_parity_gpu.py feeds stage s - 1's outputs to the oracle's restatement of stage s:
// every kernel is pinned on IDENTICAL inputs, free of the end-to-end amplification of bf16 rounding flips.
extern "C" int ptts_dac_debug_decode_upto(ptts_dac* d, const int64_t* codes_dev, int32_t B, int32_t T, int32_t stage, void* stream, void** act_dev,
                                          int32_t* act_is_bf16, float** raw_dev, int32_t* rows, int32_t* channels) {
  PTTS_CHECK(d && codes_dev && act_dev && act_is_bf16 && raw_dev && rows && channels, PTTS_E_INVALID, "null argument");
  PTTS_CHECK(stage >= 0 && stage < 1 + 4 * d->cfg.num_rates, PTTS_E_INVALID, "stage %d out of range [0, %d)", stage, 1 + 4 * d->cfg.num_rates);
  DacDebugTap tap = {};
  PTTS_TRY(dac_decode_window(d, codes_dev, T, 0, d->bufS /* unused: the decode stops before the final conv */, 0, (long long)d->hop * T, B, T, stream, -1, nullptr,
                             stage, &tap));
  *act_dev = tap.act; *act_is_bf16 = tap.act_is_bf16; *raw_dev = tap.raw; *rows = tap.rows; *channels = tap.channels;
  return PTTS_OK;
}

// Debug / parity probe: the latents z of the last ptts_dac_encode, channels-last fp32 [B][T][latent].
extern "C" int ptts_dac_debug_latents(ptts_dac* d, float** latents_dev) {
  PTTS_CHECK(d && latents_dev, PTTS_E_INVALID, "null argument");
  *latents_dev = d->bufZ;
  return PTTS_OK;
}
