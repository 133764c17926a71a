// Host-side launch helpers of the MFMA strip / block GEMMs and the rows_prep kernel (ptts_lm_kernels.h), shared by the decoder-LM engine
// (ptts_lm.hip) and the T5 description encoder (ptts_t5.hip). Every function is internal to its translation unit (anonymous namespace).
#pragma once
#include <algorithm>
#include <stdlib.h>

#include "ptts_common.h"
#include "ptts_lm_kernels.h"
#include "ptts_gemm_glds.h"
#include "ptts_strip_w8.h"

namespace {

// rows of one M pass (one workgroup over blockIdx.z) of the strip GEMMs on fragment-order activations: 16, 32 or 64 (1 / 2 / 4 MFMA column tiles per
// weight fragment). Fewer rows = more, lighter workgroups (the N = 1024 projections run on N / 16 = 64 workgroups per pass) at the price of one L2 re-read
// of every strip per pass. Measured, us per Mini-v1 step at mid context, rows 64 / 32 / 16 (profiles/r04_experiments.txt call 25):
//   24: 1235 / 1236 / 1191   32: 1291 / 1292 / 1241   48: 1766 / 1594 / 1554   64: 1958 / 1762 / 1828   96: 2293 / 2183 / 2272   128: 2580 / 2546 / 2713
// -> 2..3 passes of the smallest tile: 16 rows up to 48 utterances, 32 above (56 utterances: 1764 us on 16-row passes, 60 on 32-row passes 1725; PTTS_MSPLIT_ROWS forces one). Round 3 ran one 32-row pass up to 32
// utterances and 64-row passes above.
// `decode` (GemmArgs::decode, set by the caller): the measured policy applies to decode steps; prefill-sized rows keep the 64-row passes
inline int msplit_rows(int M, int N, bool decode) {
  static const int forced = [] {
    const char* ev = ptts_dev_env("PTTS_MSPLIT_ROWS");
    const int x = ev ? atoi(ev) : 0;
    return (x == 16 || x == 32 || x == 64) ? x : 0;
  }();
  if (forced) return forced;
  // prefill rows (time-to-first-token path): the decode policy only where 64-row passes leave the projection with fewer workgroups than CUs
  // (strips x passes < 256: the N = 1024 .. 3072 projections of a short prompt); wide projections keep the 64-row passes. Measured, prefill + first
  // token in ms (profiles/r04_experiments.txt calls 27-28), 64-row passes | lighter everywhere | lighter where strips x passes < 128:
  //   Mini-v1 33 rows 1.89 | 1.57 | 1.63   66 rows 2.16 | 1.95 | 2.16   101 rows 2.34 | 2.13 | 2.31   132 rows 2.75 | 2.69 | 2.73   fp32 33 rows 3.47 | 2.32 | 2.32
  //   Large-v1 33 rows 3.17 | 3.36 | 3.17 (its 288- / 384-strip projections lose on light passes)
  // PTTS_MSPLIT_PREFILL = 0: never, 1: everywhere, 2 (default): by workgroup count
  static const int prefill_mode = ptts_dev_env("PTTS_MSPLIT_PREFILL") ? atoi(ptts_dev_env("PTTS_MSPLIT_PREFILL")) : 2;
  if (!decode) {
    const bool lighter = prefill_mode == 1 || (prefill_mode == 2 && N > 0 && (N / 16) * ((M + 63) / 64) < 256);
    if (!lighter) return M > 32 ? 64 : 32;
  }
  if (N >= 8192) return M > 32 ? 64 : 32;  // the LM heads (612 strips): plenty of workgroups already - light passes cost 7.4 -> 10.6 us at 32 utterances (call 29)
  // (the wide projections - QKV, fc1, N >= 3072 - on their own pass size measured no better: 64-row passes for them cost +4 % at 48 / 64 utterances and
  //  are within 0.5 % at 96 / 128, profiles/r04_experiments.txt call 31: one policy for every projection below 8192 rows)
  // round 5 (the LayerNorm + projection nodes took QKV / fc1 off the strips above 40 utterances, fc2 runs un-split there): rows 16 / 32 / 64 at
  //   64: 1597 / 1671 / 1839   96: 2227 / 2157 / 2322   128: 2516 / 2444 / 2608 us per step (profiles/r05_experiments.txt call 3) -> 16 rows up to 64 utterances
  return M <= 64 ? 16 : 32;
}

template <typename WT, int PRO, int EPI, int MTP, bool FULL>
int launch_gemm_inst(GemmArgs& a, dim3 grid, dim3 block, size_t sh, hipStream_t st) {
  static PttsPerDeviceOnce attr_once;  // > 64 KiB of dynamic LDS needs an explicit opt-in, once per instantiation
  const int attr_dev = PttsPerDeviceOnce::device();
  if (attr_once.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute(strip_entry<WT, PRO, EPI, MTP, FULL>(), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
    attr_once.done(attr_dev);
  }
  launch_strip<WT, PRO, EPI, MTP, FULL>(grid, block, sh, st, a);
  return PTTS_OK;
}

// prefill-sized rows on the LDS-tiled kernel (gemm_tile_kernel): the largest tile that still gives every CU a workgroup
template <typename WT, int EPI, int BNS, int BMT>
int launch_gemm_tile_inst(const GemmArgs& a, hipStream_t st) {
  constexpr size_t sh = (size_t)2 * (BNS + BMT) * 2 * 64 * 16;
  static PttsPerDeviceOnce attr_once;
  const int attr_dev = PttsPerDeviceOnce::device();
  if (attr_once.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<WT, EPI, BNS, BMT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
    attr_once.done(attr_dev);
  }
  const dim3 grid(a.N / (16 * BNS), (a.M + BMT * 16 - 1) / (BMT * 16));
  hipLaunchKernelGGL((gemm_tile_kernel<WT, EPI, BNS, BMT>), grid, dim3(256), sh, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "gemm launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}
// -1 = shape not served (the caller keeps the register-blocked kernel)
template <typename WT, int EPI>
int launch_gemm_tile(const GemmArgs& a, hipStream_t st) {
  static const int mode = ptts_dev_env("PTTS_GEMM_TILE") ? atoi(ptts_dev_env("PTTS_GEMM_TILE")) : 1;  // 0: off (gemm_block_kernel), 1: by workgroup count, 88 / 48 / 84 / 44: force a tile
  const int nstrips = a.N / 16, nfrag = a.K / Elem<WT>::KT;
  if (!mode || nstrips % 4 || nfrag % 2 || a.K % Elem<WT>::KT) return -1;
  auto wgs = [&](int bns, int bmt) { return (nstrips / bns) * ((a.M + bmt * 16 - 1) / (bmt * 16)); };
  // measured (profiles/r05_experiments.txt call 6, time to the first token at 32 utterances = T5 on 2048 rows + prefill on 1056 rows, ms):
  //   register-blocked kernel 13.41 | tiles 8x8 14.38, 4x8 12.94, 8x4 11.57, 4x4 11.43 -> the SMALLEST tile: three workgroups per CU hide the
  //   staging latency that two stages of register prefetch do not (the loop is still latency-bound: ~250 TFLOP/s); larger tiles need an
  //   asynchronous global->LDS ring (next step)
  int pick = 44;
  (void)wgs;
  if (mode != 1) {
    pick = mode;
    if ((pick == 88 || pick == 84) && nstrips % 8) pick = pick == 88 ? 48 : 44;
  }
  switch (pick) {
    case 88: return launch_gemm_tile_inst<WT, EPI, 8, 8>(a, st);
    case 48: return launch_gemm_tile_inst<WT, EPI, 4, 8>(a, st);
    case 84: return launch_gemm_tile_inst<WT, EPI, 8, 4>(a, st);
    default: return launch_gemm_tile_inst<WT, EPI, 4, 4>(a, st);
  }
}

template <typename WT, int PRO, int EPI>
int launch_gemm(GemmArgs a, hipStream_t st) {
  constexpr int KT = Elem<WT>::KT;
  if (a.N % 16 != 0 || a.K % KT != 0) return ptts_fail(PTTS_E_INVALID, "gemm N=%d K=%d not multiples of 16/%d", a.N, a.K, KT);
  if (PRO == PRO_LN && a.K > 64 * 4 * LN_MAX_F4) return ptts_fail(PTTS_E_UNSUPPORTED, "LayerNorm width %d > %d", a.K, 64 * 4 * LN_MAX_F4);
  const int nfrag = a.K / KT;
  const int wmax = (a.M > 16 ? GemmMaxThreads<PRO, 2>::value : GemmMaxThreads<PRO, 1>::value) / 64;  // MTP 2 and 8 share the 512-thread bound
  // FULL variant: every wave owns whole 8-fragment groups (and K % 256 == 0): straight-line kernel
  int W = 0;
  const bool ln_ok = (PRO != PRO_LN && PRO != PRO_LNS) || a.K == 256 || a.K == 512 || a.K == 1024 || a.K == 1536;  // ln_row<> instances
  if (a.K % 256 == 0 && ln_ok)
    for (int w = wmax; w >= 2; --w)
      if (nfrag % (8 * w) == 0) { W = w; break; }
  const bool full = W > 0;
  if (!full) {
    W = (nfrag + 7) / 8;
    if (W < 2) W = 2;
    if (W > wmax) W = wmax;
  }
  a.frags_per_wave = nfrag / W;
  a.invK = 1.0f / (float)a.K;
  // activation rows staged in LDS per pass: as many as fit beside the cross-wave reduction buffer (<= 32)
  const size_t row_bytes = PRO == PRO_COPY ? 0 : (size_t)a.K * sizeof(WT) + 16;  // PRO_COPY reads B fragments from global
  const size_t lds_cap = 160 * 1024 - 1024;
  // prefill-sized M with prepared (PRO_COPY) activations: 128-row passes (8 MFMA tiles per weight fragment) so the
  // strip's weights are re-streamed from L2 M/128 times instead of M/32
  // (fragment-order activations, decode at batch > 32: 64-row passes, ONE pass per workgroup via blockIdx.z - twice the workgroups and
  // half the B fragments per workgroup of a 128-row pass)
  const bool msplit = PRO == PRO_COPY && a.x_fo && a.M > msplit_rows(a.M, a.N, a.decode != 0);
  const int max_rows = msplit ? msplit_rows(a.M, a.N, a.decode != 0) : ((PRO == PRO_COPY && a.M > 32) ? 128 : 32);
  int rpp = a.M < max_rows ? a.M : max_rows;
  auto tiles = [](int r) { return r > 64 ? 8 : (r > 32 ? 4 : (r > 16 ? 2 : 1)); };
  while (rpp > 1 && rpp * row_bytes + (size_t)W * tiles(rpp) * 1024 > lds_cap) --rpp;
  if (rpp * row_bytes + (size_t)W * 1024 > lds_cap) return ptts_fail(PTTS_E_UNSUPPORTED, "gemm K=%d does not fit in LDS", a.K);
  a.rows_per_pass = rpp;
  const int mtp = tiles(rpp);
  const size_t sh = rpp * row_bytes + (size_t)W * mtp * 1024 + 256;  // + rstd of the pass's rows (rs_part consumers)
  a.m_split = msplit ? 1 : 0;
  const dim3 grid(a.N / 16, 1, msplit ? (a.M + rpp - 1) / rpp : ((EPI == EPI_KV && a.kv_layers) ? a.kv_nlayers : 1)), block(W * 64);
  int rc;
  if constexpr (sizeof(WT) == 2) {
    if (a.W8 && full && a.K % 64 == 0) {  // e4m3 strips (weights_fp8): same grid / LDS, half the weight bytes; -1 = no instance, bf16 strips below
      rc = ptts_strip_w8_launch(PRO, EPI, mtp, a, grid, block, sh, st);
      if (rc == 0) return PTTS_OK;
      if (rc != -1) return PTTS_E_HIP;
    }
  }
  if constexpr (PRO == PRO_COPY) {
    static const int block_min_m = ptts_dev_env("PTTS_BLOCK_MIN_M") ? atoi(ptts_dev_env("PTTS_BLOCK_MIN_M")) : 256;
    // measured (tools/ttft_bs32_probe.py, Mini-v1 prefill ms, strip / block): M=132 5.7 / 8.2, 264 8.3 / 8.1, 528 13.5 / 8.7, 1056 25.7 / 11.2
    static const int xcd_swz = !(ptts_dev_env("PTTS_GEMM_XCD") && !atoi(ptts_dev_env("PTTS_GEMM_XCD")));
    a.xcd_swz = xcd_swz;
    if constexpr (sizeof(WT) == 2) {  // bf16 engine: the LDS-DMA ring (round 6, ptts_gemm_glds.h); PTTS_GEMM_GLDS=0: round 5's register-staged tiles
      static const bool glds_on = !(ptts_dev_env("PTTS_GEMM_GLDS") && !atoi(ptts_dev_env("PTTS_GEMM_GLDS")));
      if (glds_on && a.M > block_min_m && EPI != EPI_GELU && !a.x_fo && (!a.kv_layers || EPI == EPI_KV) && !a.stats_out && !a.W8 && !a.rs_part && !a.nx_out) {
        const int rg = launch_gemm_glds<EPI>(a, st);
        if (rg != -1) return rg;
      }
    }
    if (a.M > block_min_m && EPI != EPI_GELU && !a.x_fo && !a.kv_layers && !a.stats_out && !a.W8) {  // LDS-tiled kernel (round 5)
      const int rt = launch_gemm_tile<WT, EPI>(a, st);
      if (rt != -1) return rt;
    }
    if (a.M > block_min_m && EPI != EPI_GELU && !a.x_fo && !a.kv_layers) {  // prefill-sized: register-blocked kernel, no K split
      const int nstrips = a.N / 16;
      const int ns = (nstrips % 4 == 0 && nstrips >= 128) ? 4 : (nstrips % 2 == 0 ? 2 : 0);  // N = 1024: 2 strips per wave keeps > 500 waves in flight
      if (ns) {
        const dim3 g2(nstrips / ns, (a.M + 255) / 256), b2(256);
        if (ns == 4) hipLaunchKernelGGL((gemm_block_kernel<WT, EPI, 4>), g2, b2, 0, st, a);
        else hipLaunchKernelGGL((gemm_block_kernel<WT, EPI, 2>), g2, b2, 0, st, a);
        hipError_t eb = hipGetLastError();
        if (eb != hipSuccess) return ptts_fail(PTTS_E_HIP, "gemm launch failed: %s", hipGetErrorString(eb));
        return PTTS_OK;
      }
    }
    if (mtp >= 4) {
      if (mtp == 8) rc = full ? launch_gemm_inst<WT, PRO, EPI, 8, true>(a, grid, block, sh, st) : launch_gemm_inst<WT, PRO, EPI, 8, false>(a, grid, block, sh, st);
      else rc = full ? launch_gemm_inst<WT, PRO, EPI, 4, true>(a, grid, block, sh, st) : launch_gemm_inst<WT, PRO, EPI, 4, false>(a, grid, block, sh, st);
      PTTS_TRY(rc);
      hipError_t e8 = hipGetLastError();
      if (e8 != hipSuccess) return ptts_fail(PTTS_E_HIP, "gemm launch failed: %s", hipGetErrorString(e8));
      return PTTS_OK;
    }
  }
  if (mtp > 2) return ptts_fail(PTTS_E_UNSUPPORTED, "gemm: %d activation rows per pass need the PRO_COPY path", rpp);
  if (full) rc = mtp == 1 ? launch_gemm_inst<WT, PRO, EPI, 1, true>(a, grid, block, sh, st) : launch_gemm_inst<WT, PRO, EPI, 2, true>(a, grid, block, sh, st);
  else rc = mtp == 1 ? launch_gemm_inst<WT, PRO, EPI, 1, false>(a, grid, block, sh, st) : launch_gemm_inst<WT, PRO, EPI, 2, false>(a, grid, block, sh, st);
  PTTS_TRY(rc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "gemm launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

template <typename WT>
int launch_attn(const AttnArgs& a, int B, hipStream_t st, int waves = 4) {
  const dim3 grid(a.S, a.nheads, B * a.Q);
  if (a.kscale) {  // e4m3 self-attention cache (ptts_config::kv_fp8): bf16 engine, 4 waves per workgroup
    if constexpr (sizeof(WT) == 2) {
      if (waves != 4) return ptts_fail(PTTS_E_UNSUPPORTED, "kv_fp8: the attention kernel is built for 4 waves per workgroup (PTTS_ATTN_WAVES=%d)", waves);
      hipLaunchKernelGGL((attn_kernel<WT, 4, true>), grid, dim3(256), 0, st, a);
      hipError_t e8 = hipGetLastError();
      if (e8 != hipSuccess) return ptts_fail(PTTS_E_HIP, "attn launch failed: %s", hipGetErrorString(e8));
      return PTTS_OK;
    } else {
      return ptts_fail(PTTS_E_UNSUPPORTED, "kv_fp8 needs the bf16 engine");
    }
  }
  if (waves == 1) hipLaunchKernelGGL((attn_kernel<WT, 1>), grid, dim3(64), 0, st, a);
  else if (waves == 2) hipLaunchKernelGGL((attn_kernel<WT, 2>), grid, dim3(128), 0, st, a);
  else if (waves == 8) hipLaunchKernelGGL((attn_kernel<WT, 8>), grid, dim3(512), 0, st, a);
  else if (waves == 16) hipLaunchKernelGGL((attn_kernel<WT, 16>), grid, dim3(1024), 0, st, a);
  else hipLaunchKernelGGL((attn_kernel<WT, 4>), grid, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "attn launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

// prefill attention, tiled (prefill_attn_kernel): 8 query rows per workgroup; Q = a.Q rows per utterance, output in a.direct_out
// mode (PTTS_PREFILL_ATTN): 1 = the VALU kernel, 2 = the f32-MFMA kernel (engine-dtype caches), anything else = by batch: the MFMA kernel's 16-query
// waves fill the chip from 128 (utterance, head) pairs up
template <typename WT>
int launch_prefill_attn(const AttnArgs& a, int B, hipStream_t st, int mode = 3) {
  if (!a.kscale && (mode == 2 || (mode != 1 && B * a.nheads >= 128))) {
    const int waves = std::min(4, (a.Q + 15) / 16);
    const dim3 gm((a.Q + 16 * waves - 1) / (16 * waves), a.nheads, B);
    int rcm;
    if (waves == 4) rcm = ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<WT, 4>, gm, dim3(256), st, a);
    else if (waves == 3) rcm = ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<WT, 3>, gm, dim3(192), st, a);
    else if (waves == 2) rcm = ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<WT, 2>, gm, dim3(128), st, a);
    else rcm = ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<WT, 1>, gm, dim3(64), st, a);
    if (rcm != PTTS_OK) return rcm;
    hipError_t em = hipGetLastError();
    if (em != hipSuccess) return ptts_fail(PTTS_E_HIP, "prefill attention launch failed: %s", hipGetErrorString(em));
    return PTTS_OK;
  }
  const dim3 grid((a.Q + 7) / 8, a.nheads, B);
  int rcv;
  if (a.kscale) {
    if constexpr (sizeof(WT) == 2) rcv = ptts_launch_prefill_attn_kernel(prefill_attn_kernel<WT, true>, grid, dim3(256), st, a);
    else return ptts_fail(PTTS_E_UNSUPPORTED, "kv_fp8 needs the bf16 engine");
  } else {
    rcv = ptts_launch_prefill_attn_kernel(prefill_attn_kernel<WT, false>, grid, dim3(256), st, a);
  }
  if (rcv != PTTS_OK) return rcv;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ptts_fail(PTTS_E_HIP, "prefill attention launch failed: %s", hipGetErrorString(e));
  return PTTS_OK;
}

template <typename WT, int PRO>
int launch_prep(GemmArgs a, void* dst, hipStream_t st) {
  a.invK = 1.0f / (float)a.K;
  if ((unsigned)a.x_row_mul > 0xffffu || (unsigned)a.x_row_off > 0xffffu) return ptts_fail(PTTS_E_UNSUPPORTED, "rows_prep: row selection %d * m + %d does not fit the packed slot", a.x_row_mul, a.x_row_off);
  // the preloaded slots this node does not use carry what addresses its first loads (rows_prep_kernel)
  a.W = a.gamma; a.W8 = a.beta; a.out = reinterpret_cast<float*>(dst); a.rows_per_pass = a.x_ld; a.frags_per_wave = (int)((unsigned)a.x_row_mul | ((unsigned)a.x_row_off << 16));
  a.out_ld = a.out_fo;
  ptts_klaunch(rows_prep_kernel<WT, PRO>, dim3((a.M + 3) / 4), dim3(256), 0, st, a);
  return PTTS_OK;
}

}  // namespace
