// gemm_strip_kernel<bf16, PRO, EPI, MTP, FULL = true, W8 = true>: the (prologue, epilogue, row tiles) combinations the decode step
// launches at batch 5..256 (ptts_lm.hip forward<>): fused prologues at batch <= 8, prepared / producer-statistics rows above.
#include "ptts_common.h"
#include "ptts_lm_kernels.h"
#include "ptts_strip_w8.h"

namespace {

template <int PRO, int EPI, int MTP>
int launch(const GemmArgs& a, dim3 grid, dim3 block, size_t sh, hipStream_t st) {
  static PttsPerDeviceOnce attr_once;  // > 64 KiB of dynamic LDS needs an explicit opt-in, once per instantiation
  const int attr_dev = PttsPerDeviceOnce::device();
  if (attr_once.need(attr_dev)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_strip_kernel<bf16_t, PRO, EPI, MTP, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { ptts_fail(PTTS_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e)); return -2; }
    attr_once.done(attr_dev);
  }
  ptts_klaunch(gemm_strip_kernel<bf16_t, PRO, EPI, MTP, true, true>, grid, block, sh, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ptts_fail(PTTS_E_HIP, "e4m3 strip gemm launch failed: %s", hipGetErrorString(e)); return -2; }
  return 0;
}

}  // namespace

int ptts_strip_w8_launch(int pro, int epi, int mtp, const GemmArgs& a, dim3 grid, dim3 block, size_t sh, hipStream_t st) {
#define PTTS_W8_CASE(P, E, M) if (pro == P && epi == E && mtp == M) return launch<P, E, M>(a, grid, block, sh, st);
  // batch 5..8: fused prologues, one row tile
  PTTS_W8_CASE(PRO_LN, EPI_STORE, 1) PTTS_W8_CASE(PRO_ATTN, EPI_RESID, 1) PTTS_W8_CASE(PRO_LN, EPI_GELU, 1) PTTS_W8_CASE(PRO_PLAIN, EPI_RESID, 1)
  // prepared rows (rows_prep / attention / fused cross block / GELU epilogue), 1 / 2 / 4 row tiles = batch <= 16 / <= 32 / 64-row passes
  PTTS_W8_CASE(PRO_COPY, EPI_STORE, 1) PTTS_W8_CASE(PRO_COPY, EPI_STORE, 2) PTTS_W8_CASE(PRO_COPY, EPI_STORE, 4)
  PTTS_W8_CASE(PRO_COPY, EPI_RESID, 1) PTTS_W8_CASE(PRO_COPY, EPI_RESID, 2) PTTS_W8_CASE(PRO_COPY, EPI_RESID, 4)
  PTTS_W8_CASE(PRO_COPY, EPI_GELU_WT, 4)
  // producer-statistics LayerNorm (batch 9..32)
  PTTS_W8_CASE(PRO_LNS, EPI_GELU_WT, 1) PTTS_W8_CASE(PRO_LNS, EPI_GELU_WT, 2)
#undef PTTS_W8_CASE
  return -1;
}
