// e4m3-weight instances of gemm_strip_kernel (ptts_lm_kernels.h, W8 = true), compiled in their own translation unit
// (ptts_lm_w8.hip) beside ptts_lm.hip. Decode step of weights_fp8 engines at batch >= 5 (BASELINE configs[4]).
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs;
// 0 = launched, -1 = no instance for (pro, epi, mtp) - the caller runs the bf16 strips on the exact dequantisation instead,
// -2 = HIP error (ptts_last_error set)
int ptts_strip_w8_launch(int pro, int epi, int mtp, const GemmArgs& a, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st);
