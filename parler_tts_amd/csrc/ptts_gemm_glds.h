// gemm_glds_kernel (round 6): the prefill-sized GEMM (> 256 activation rows: T5 on 32 descriptions x 64 tokens = 2048 rows, the prefill of 32
// utterances x 33 positions = 1056 rows) as an LDS ring filled by LDS-DMA (`global_load_lds_dwordx4`: global -> LDS without destination registers),
// NST stages deep, ONE raw s_barrier per stage and COUNTED vmcnt waits - loads stay in flight across the barriers. Round 5's gemm_tile_kernel staged
// global -> registers -> LDS two stages ahead and ran at 6-15 % of the bf16 MFMA peak (144-364 TFLOP/s, profiles/r05_prefill_kernels_bs32.txt): with
// K = 1024 a workgroup's whole MFMA work is shorter than one memory latency and the register ring cannot be deepened (VGPRs).
//   out[m][n] = sum_k W[n][k] x[m][k]     (modeling_parler_tts.py:3048-3097 description encoder, :1437-1439 prefill rows, :877-878 cross K/V)
// Operands (bf16 engine only; the fp32 parity engine keeps gemm_tile_kernel):
//   A = weights, packed at load in MFMA A-fragment order [N/16 strips][K/32 fragments][64 lanes][16 B]: one fragment = one contiguous 1 KiB = ONE
//       LDS-DMA wave-instruction, and the lane-linear LDS image (base + lane * 16) is exactly what the fragment's ds_read_b128 wants (conflict-free)
//   B = activations, row-major bf16 [M][K]: one wave-instruction brings 8 rows x 128 B (full cache lines: lanes 8 r .. 8 r + 7 read one line) into an
//       [8][8 x 16 B] image; the 16-byte piece a lane FETCHES is permuted within its line (piece = slot ^ row & 7: the swizzle sits on the SOURCE
//       address, the LDS-DMA destination is always lane-linear) and the fragment read applies the same XOR: the 16 lanes of every ds_read_b128 lane
//       group hit 16 different 16-byte slots of the 256-byte bank row (conflict-free; linear rows would be 4-way)
// Tile: BNS strips (16 weight rows each) x BMT row tiles (16 activation rows each) per workgroup, WN x WM waves of (BNS / WN) x (BMT / WM) MFMA tiles,
// BK = 64 per stage (2 fragments). Accumulation order over k is ascending whole fragments, no K split: bit-identical to gemm_tile_kernel /
// gemm_block_kernel (tests compare them bitwise). Epilogues: gemm_store_tile (store / residual / GELU / gated GELU / cross K/V scatter).
#pragma once
#include "ptts_lm_kernels.h"

namespace {

template <int N> __device__ __forceinline__ void ptts_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL (tools/gemm_probe only): 0 = the kernel; 1 = no fragment reads / MFMAs (load stream + barriers only); 2 = no LDS-DMA (compute on whatever the LDS holds)
template <int EPI, int BNS, int BMT, int WN, int WM, int NST, int ABL = 0, int RP = 0, int KF = 2>
__global__ void __launch_bounds__(WN * WM * 64) gemm_glds_kernel(PTTS_DBG0_PARAM GemmArgs_KPARAMS) {
  // Kernel-argument preload (ptts_common.h; call 52): everything that addresses the wave's first LDS-DMA pieces - W, x, K, M and, in the two preloaded slots
  // the strip kernel's pass geometry occupies, the activation row stride and the tile-order flag + grid extents (launch_gemm_glds_inst; gridDim is a hidden argument behind an s_load) - arrives in SGPRs written by the
  // command processor; the tail (epilogue operands) comes by s_load in the shadow of the first stage.
  GemmArgs_KJOIN(a)
  const int x_ld = a.rows_per_pass, xcd_swz = a.frags_per_wave & 1, grid_n = (a.frags_per_wave >> 1) & 0x7ff, grid_m = (int)((unsigned)a.frags_per_wave >> 12);
  typedef bf16_t WT;
  static_assert(KF == 2 || KF == 4, "BK = 64 or 128 per stage");
  constexpr int RB = KF * 64, SPR = RB / 16, RPP = 1024 / RB;  // activation image: bytes per row, 16-byte slots per row, rows per 1 KiB piece
  constexpr int NW = WN * WM, NS = BNS / WN, MT = BMT / WM;
  constexpr int APC = BNS * KF, BPC = BMT * KF;           // 1 KiB pieces per stage: weights, activations
  // pieces per wave and stage (the first APW of them weight pieces). EVEN: every wave moves the same pieces per stage, weights first; otherwise
  // (tiles such as 176 weight rows: 22 + 32 pieces on 8 waves) piece p = wave + NW * i is a weight piece iff p < APC, decided per wave at run time, and a
  // wave without an i-th piece fetches its (i - 1)-th again (same source, same destination: the per-wave load count stays a compile-time constant)
  constexpr bool EVEN = APC % NW == 0 && BPC % NW == 0;
  constexpr int PPW = (APC + BPC + NW - 1) / NW, APW = APC / NW;
  static_assert(BNS % WN == 0 && BMT % WM == 0, "tile / wave split");
  static_assert((NST - 2) * PPW <= 63 && NST >= 2, "vmcnt is a 6-bit counter");
  static_assert(RP != 2 || NST >= 3, "the half-stage pipeline reads stage t + 1 while stage t computes");
  constexpr int STAGE_B = (APC + BPC) * 1024;
  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave % WN, wm = wave / WN;
  const int q = lane >> 4, j = lane & 15;
  if constexpr (EPI == EPI_KV) {  // the description's K/V of EVERY layer in one launch: blockIdx.z selects the layer's weights and caches
    if (a.kv_layers) { const KvLayer kl = a.kv_layers[blockIdx.z]; a.W = kl.W; a.kcache = kl.k; a.vcache = kl.v; }
  }
  int bx, by;
  xcd_tile_order(bx, by, xcd_swz, grid_n, grid_m);
  const int strip0 = bx * BNS, m0 = by * BMT * 16;
  const int nfrag = a.K >> 5, nstage = nfrag / KF;        // host guarantees K % 64 == 0
  // this wave's pieces: piece p = wave + NW * i; p < APC: fragment p % KF of strip p / KF, else rows 8 (p - APC) .. + 7 of the row tile
  const char* src[PPW];
  int adv[PPW], pdst[PPW];  // !EVEN: bytes per stage the piece's source advances by, and its LDS slot
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    int p = wave + NW * i;
    if constexpr (!EVEN) { if (p >= APC + BPC) p -= NW; }
    const bool wpiece = EVEN ? i < APW : p < APC;
    adv[i] = wpiece ? KF * 1024 : KF * 64;
    pdst[i] = p * 1024;
    if (wpiece) {
      src[i] = reinterpret_cast<const char*>(a.W) + ((size_t)(strip0 + p / KF) * nfrag + p % KF) * 1024 + lane * 16;
    } else {
      const int rl = (p - APC) * RPP + lane / SPR;            // row of the tile; its image slot lane % SPR holds piece slot ^ (row & (SPR - 1))
      const int row = min(m0 + rl, a.M - 1);                  // clamped rows are computed and dropped
      src[i] = reinterpret_cast<const char*>(a.x) + (size_t)row * x_ld * sizeof(WT) + (((lane % SPR) ^ (rl & (SPR - 1))) << 4);  // x_row_mul 1, x_row_off 0 (launcher)
    }
  }
  auto issue = [&](int t, int buf) {
    if constexpr (ABL == 2) return;
    char* dst = smem_raw + buf * STAGE_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if constexpr (EVEN)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src[i] + (size_t)t * (i < APW ? KF * 1024 : KF * 64)),
                                         (__attribute__((address_space(3))) void*)(dst + NW * i * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src[i] + (size_t)t * adv[i]),
                                         (__attribute__((address_space(3))) void*)(smem_raw + buf * STAGE_B + pdst[i]), 16, 0, 0);
    }
  };
  f32x4 acc[NS][MT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane read offsets: weights lane-linear; activations row j of the tile, piece (f * 4 + q) ^ (j & 7)
  const int a_off = wn * NS * KF * 1024 + lane * 16;
  int b_off[KF];
#pragma unroll
  for (int f = 0; f < KF; ++f) b_off[f] = APC * 1024 + (wm * MT * 16 + j) * RB + (((f * 4 + q) ^ (j & (SPR - 1))) << 4);

#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nstage) issue(t, t);
  int buf = 0;
  if constexpr (RP == 2 && ABL == 0) {
    // Half-stage software pipeline for ONE resident workgroup per CU (tiles of ~1 / 256 of the output: nobody else fills the MFMA pipe while this
    // workgroup's waves wait for their fragments): the reads of a stage's second half are issued before the MFMAs of its first half, and the reads of
    // the NEXT stage's first half before the MFMAs of the second - so the top-of-iteration barrier is for stage t + 1 (landed for every wave), and at most
    // NST - 3 later stages stay in flight across it. Same MFMA order per accumulator (fragments ascending): bit-identical to the other variants.
    static_assert(KF == 2, "two halves per stage");
    u32x4_t a0[NS], b0[MT], a1[NS], b1[MT];
    auto reads = [&](const char* cur, const int f, u32x4_t (&af)[NS], u32x4_t (&bf)[MT]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const u32x4_t*>(cur + a_off + (s * KF + f) * 1024);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = *reinterpret_cast<const u32x4_t*>(cur + b_off[f] + mt * 16 * RB);
    };
    auto mfmas = [&](const u32x4_t (&af)[NS], const u32x4_t (&bf)[MT]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mfma_step_v<WT>(af[s], bf[mt], acc[s][mt]);
    };
    // stage 0 landed (for every wave) before its first half is read
    if (nstage - 1 >= NST - 2) ptts_wait_vmcnt<(NST - 2) * PPW>(); else ptts_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    reads(smem_raw, 0, a0, b0);
    for (int t = 0; t < nstage; ++t) {
      // stage t + 1 has landed for this wave: loads of at most min(NST - 3, nstage - 2 - t) later stages may still be in flight
      const int later = nstage - 2 - t;
      if (later >= NST - 3) ptts_wait_vmcnt<(NST - 3) * PPW>();
      else if (NST > 4 && later == 1) ptts_wait_vmcnt<PPW>();
      else if (NST > 5 && later == 2) ptts_wait_vmcnt<2 * PPW>();
      else ptts_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (t, first half) is in registers; every read of stage t - 1 retired long ago
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + NST - 1 < nstage) issue(t + NST - 1, buf == 0 ? NST - 1 : buf - 1);
      const char* cur = smem_raw + buf * STAGE_B;
      const int nbuf = buf + 1 == NST ? 0 : buf + 1;
      reads(cur, 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (t + 1 < nstage) reads(smem_raw + nbuf * STAGE_B, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      buf = nbuf;
    }
  } else
  for (int t = 0; t < nstage; ++t) {
    // stage t has landed for this wave: loads of at most min(NST - 2, nstage - 1 - t) later stages may still be in flight
    const int later = nstage - 1 - t;
    if (later >= NST - 2) ptts_wait_vmcnt<(NST - 2) * PPW>();
    else if (NST > 3 && later == 1) ptts_wait_vmcnt<PPW>();
    else if (NST > 4 && later == 2) ptts_wait_vmcnt<2 * PPW>();
    else ptts_wait_vmcnt<0>();
    // ... and for every wave, and every wave is done reading buffer (t - 1) % NST (its fragments were consumed by MFMAs issued before this point)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (t + NST - 1 < nstage) issue(t + NST - 1, buf == 0 ? NST - 1 : buf - 1);
    const char* cur = smem_raw + buf * STAGE_B;
    if constexpr (RP == 1 && ABL != 1) {
      // every fragment read of the stage is issued before its first MFMA: one exposed LDS latency per stage instead of one per group of reads
      u32x4_t af[KF][NS], bf[KF][MT];
#pragma unroll
      for (int f = 0; f < KF; ++f) {
#pragma unroll
        for (int s = 0; s < NS; ++s) af[f][s] = *reinterpret_cast<const u32x4_t*>(cur + a_off + (s * KF + f) * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) bf[f][mt] = *reinterpret_cast<const u32x4_t*>(cur + b_off[f] + mt * 16 * RB);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mfma_step_v<WT>(af[f][s], bf[f][mt], acc[s][mt]);
    } else {
#pragma unroll
    for (int f = 0; f < (ABL == 1 ? 0 : KF); ++f) {
      u32x4_t af[NS], bf[MT];
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const u32x4_t*>(cur + a_off + (s * KF + f) * 1024);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bf[mt] = *reinterpret_cast<const u32x4_t*>(cur + b_off[f] + mt * 16 * RB);
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mfma_step_v<WT>(af[s], bf[mt], acc[s][mt]);
    }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  if constexpr (EPI == EPI_RESID) {  // residual pieces requested before the first store (loads and stores retire in order on one counter)
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

template <int EPI, int BNS, int BMT, int WN, int WM, int NST, int ABL = 0, int RP = 0, int KF = 2>
int launch_gemm_glds_inst(const GemmArgs& a, hipStream_t st) {
  constexpr size_t sh = (size_t)NST * (BNS + BMT) * KF * 1024;
  static PttsPerDeviceOnce attr_once;
  const int attr_dev = PttsPerDeviceOnce::device();
  if (sh > 64 * 1024 && attr_once.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<EPI, BNS, BMT, WN, WM, NST, ABL, RP, KF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
    attr_once.done(attr_dev);
  }
  const dim3 grid(a.N / (16 * BNS), (a.M + BMT * 16 - 1) / (BMT * 16), (EPI == EPI_KV && a.kv_layers) ? a.kv_nlayers : 1);
  if (a.x_row_mul != 1 || a.x_row_off != 0) return ptts_fail(PTTS_E_UNSUPPORTED, "gemm_glds: activation rows must be consecutive (x_row_mul %d, x_row_off %d)", a.x_row_mul, a.x_row_off);
  GemmArgs b = a;  // the two preloaded slots of the strip kernel's pass geometry carry what this kernel needs to address its first loads
  if (grid.x > 0x7ffu || grid.y > 0xfffffu) return ptts_fail(PTTS_E_UNSUPPORTED, "gemm_glds: %u x %u tiles do not fit the packed grid extents", grid.x, grid.y);
  b.rows_per_pass = a.x_ld; b.frags_per_wave = (int)((a.xcd_swz ? 1u : 0u) | (grid.x << 1) | (grid.y << 12));  // tile-order flag | tiles along N | tiles along M
  ptts_klaunch(gemm_glds_kernel<EPI, BNS, BMT, WN, WM, NST, ABL, RP, KF>, grid, dim3(WN * WM * 64), sh, st, b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "gemm launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// Tile policy, measured per shape on MI355X (tools/gemm_probe, profiles/r06_gemm_probe.txt; us per launch, fp32 outputs, weights from HBM):
//   narrow projections (N <= 1024: o, cross q / o, wo, fc2): 64 x 64 tiles, 3-stage ring - 272..512 workgroups, two per CU
//       2048 x 1024 x 1024 9.4 (r05 17.6)   2048 x 1024 x 2816 19.7 (41.0)   1056 x 1024 x 1024 8.3 (15.7)   1056 x 1024 x 4096 23.0 (53.0)
//   wide projections with >= 400 tiles of 128 x 128 (T5 wi, Large fc1): 128 x 128, 8 waves, 2-stage ring (64 KiB: two workgroups per CU)
//       2048 x 5632 x 1024 34.4 (73.0)   1056 x 6144 x 1536 28.7 (64.5)
//   the rest (q|k|v, fc1, cross K|V): 128 weight rows x 64 activation rows, 4 waves, 2-stage ring (48 KiB: three workgroups per CU)
//       2048 x 3072 x 1024 19.7 (41.6)   1056 x 3072 x 1024 14.5 (29.0)   1056 x 4096 x 1024 18.8 (34.5)   2048 x 2048 x 1024 15.2 (30.1)
// Deeper rings lose above one stage in flight wherever they cost a resident workgroup: every variant is bound by what ONE workgroup's
// dependent chain (barrier -> fragment reads -> MFMAs per stage) and its LDS-DMA issue rate sustain, so resident workgroups per CU beat
// pipeline depth; BK = 128 stages (half the barriers) lose for the same reason (twice the LDS per stage). -1 = shape not served.
// Round 6, calls 32-34 (profiles/r06_gemm_probe_call32.txt, _call34.txt): where tiles of ~1 / 256 of the output (or whole rounds of them) exist, fewer and
// larger workgroups beat two or three smaller resident ones - fewer L2 -> LDS bytes per flop: 2048 | 4096 | 8192 x 3072 x 1024 on 192 x 128 tiles (2-stage
// ring: two workgroups per CU) 19.2 | 31.3 | 61.0 us against 21.4 | 34.8 | 65.5, x 5632 on 176 x 256 tiles (one workgroup per CU; 22 + 32 pieces per stage
// on 8 waves: the uneven split) 29.5 | 60.9 | 111.2 against 34.8 | 62.5 | 118.9 (the vendor library at 2048 rows, graph-captured torch.matmul: 16.9 / 32.6,
// profiles/r06_vendor_gemm.txt). Only where the tile count fills whole rounds (>= 85 %): 1056 x 3072 on 192 x 128 tiles (144 of them) loses, 17.0 against 14.2.
// Deeper rings and the half-stage pipeline (RP = 2) lose on every shape and stay probe-only.
static inline bool glds_fills_rounds(int tiles, int slots) {
  const int rounds = (tiles + slots - 1) / slots;
  return tiles > 224 && (tiles <= 256 || tiles * 100 >= rounds * slots * 85);
}
template <int EPI>
int launch_gemm_glds(const GemmArgs& a, hipStream_t st) {
  if (a.K % 64 || a.N % 64 || a.x_ld % 8 || a.x_row_mul != 1 || a.x_row_off != 0) return -1;
  static const bool big_tiles = !(ptts_dev_env("PTTS_GLDS_BIG_TILES") && !atoi(ptts_dev_env("PTTS_GLDS_BIG_TILES")));  // A/B (dev-knob build): 0 = calls 2-6 policy
  if constexpr (EPI == EPI_STORE) {
    if (big_tiles && a.N % 192 == 0 && glds_fills_rounds((a.N / 192) * ((a.M + 127) / 128), 512)) return launch_gemm_glds_inst<EPI, 12, 8, 4, 2, 2, 0, 1>(a, st);
  }
  if constexpr (EPI == EPI_GATE_WT) {
    if (big_tiles && a.N % 176 == 0 && glds_fills_rounds((a.N / 176) * ((a.M + 255) / 256), 256)) return launch_gemm_glds_inst<EPI, 11, 16, 1, 8, 2, 0, 1>(a, st);
  }
  if (a.N <= 1024 || a.N % 128) return launch_gemm_glds_inst<EPI, 4, 4, 2, 2, 3>(a, st);
  // (the description's K / V of every layer in ONE launch is kv_nlayers problems' worth of tiles: it fills the chip for many rounds like a large GEMM)
  static const bool kv_all = !(ptts_dev_env("PTTS_GLDS_KV_TILES") && !atoi(ptts_dev_env("PTTS_GLDS_KV_TILES")));  // A/B (dev-knob build): 0 = per-layer tile count
  const int tiles128 = (a.N / 128) * ((a.M + 127) / 128) * ((EPI == EPI_KV && a.kv_layers && kv_all) ? a.kv_nlayers : 1);
  if (tiles128 >= 400) return launch_gemm_glds_inst<EPI, 8, 8, 4, 2, 2, 0, 1>(a, st);
  return launch_gemm_glds_inst<EPI, 8, 4, 2, 2, 2, 0, 1>(a, st);
}

}  // namespace
