// gemm_probe: the prefill-sized GEMMs of the time-to-first-token path (T5 on 2048 rows, prefill on 1056 rows) on gemm_glds_kernel (round 6, LDS-DMA
// ring) against gemm_tile_kernel (round 5, register staging): bitwise comparison of the outputs (same k order) and time per launch / TFLOP/s per
// shape and tile variant. Weights rotate over enough copies to exceed the 256 MB Infinity Cache (in the encoder every layer has its own weights).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 -I include -I parler_tts_amd/csrc tools/gemm_probe.hip -o tools/gemm_probe
//   tools/gemm_probe [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ptts_gemm_launch.h"

// ptts_fail lives in the library's translation units; the probe links none of them
int ptts_fail(int code, const char* fmt, ...) { fprintf(stderr, "ptts_fail(%d): %s\n", code, fmt); return code; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static __global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float v = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale;  // uniform [-scale, scale): full-range signs (guide rule 25)
    p[i] = f32_to_bf16(v);
  }
}

struct Variant { const char* name; int (*fn)(const GemmArgs&, hipStream_t); int bn, bm; };
template <int BNS, int BMT, int WN, int WM, int NST, int ABL = 0, int RP = 0, int KF = 2> int run_glds(const GemmArgs& a, hipStream_t st) { return launch_gemm_glds_inst<EPI_STORE, BNS, BMT, WN, WM, NST, ABL, RP, KF>(a, st); }
static int run_tile44(const GemmArgs& a, hipStream_t st) { return launch_gemm_tile_inst<bf16_t, EPI_STORE, 4, 4>(a, st); }
static int run_tile88(const GemmArgs& a, hipStream_t st) { return launch_gemm_tile_inst<bf16_t, EPI_STORE, 8, 8>(a, st); }

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  struct Shape { const char* what; int M, N, K; };
  const Shape shapes[] = {
      {"T5 q|k|v      ", 2048, 3072, 1024}, {"T5 o          ", 2048, 1024, 1024}, {"T5 wi (gated) ", 2048, 5632, 1024}, {"T5 wo         ", 2048, 1024, 2816},
      {"prefill qkv   ", 1056, 3072, 1024}, {"prefill o / cq ", 1056, 1024, 1024}, {"prefill fc1   ", 1056, 4096, 1024}, {"prefill fc2   ", 1056, 1024, 4096},
      {"cross K|V     ", 2048, 2048, 1024}, {"Large fc1     ", 1056, 6144, 1536}, {"square 4096   ", 4096, 4096, 4096},
      {"T5 q|k|v x64  ", 4096, 3072, 1024}, {"T5 wi x64     ", 4096, 5632, 1024}, {"T5 q|k|v x128 ", 8192, 3072, 1024}, {"T5 wi x128    ", 8192, 5632, 1024}};
  const char* only = argc > 2 ? argv[2] : nullptr;  // run only the shapes whose name contains this text
  const Variant vars[] = {
      {"tile 64x64 (r05)      ", run_tile44, 64, 64},
      {"glds 128x128 8w s2 rp ", run_glds<8, 8, 4, 2, 2, 0, 1>, 128, 128},
      {"glds 128x64  4w s2 rp ", run_glds<8, 4, 2, 2, 2, 0, 1>, 128, 64},
      {"glds 128x64  4w s3 rp ", run_glds<8, 4, 2, 2, 3, 0, 1>, 128, 64},
      {"glds 64x64   4w s3    ", run_glds<4, 4, 2, 2, 3>, 64, 64},
      // call 32: ONE workgroup per CU on ~1 / 256 of the output (the L2 -> LDS bytes per flop of a tile bound every variant above: 128 x 128 tiles reach
      // 42 % even on 4096^3), deep rings, and the half-stage pipeline (p2: next half's fragment reads under this half's MFMAs)
      {"glds 128x64  8w s3 p2 ", run_glds<8, 4, 4, 2, 3, 0, 2>, 128, 64},
      {"glds 128x64  8w s4 p2 ", run_glds<8, 4, 4, 2, 4, 0, 2>, 128, 64},
      {"glds 128x64  8w s6 p2 ", run_glds<8, 4, 4, 2, 6, 0, 2>, 128, 64},
      {"glds 128x64  4w s4 p2 ", run_glds<8, 4, 2, 2, 4, 0, 2>, 128, 64},
      {"glds 128x64  4w s6 rp ", run_glds<8, 4, 2, 2, 6, 0, 1>, 128, 64},
      {"glds 64x64   4w s3 p2 ", run_glds<4, 4, 2, 2, 3, 0, 2>, 64, 64},
      {"glds 128x128 8w s3 p2 ", run_glds<8, 8, 4, 2, 3, 0, 2>, 128, 128},
      {"glds 128x128 8w s4 p2 ", run_glds<8, 8, 4, 2, 4, 0, 2>, 128, 128},
      {"glds 192x128 8w s2 rp ", run_glds<12, 8, 4, 2, 2, 0, 1>, 192, 128},
      {"glds 192x128 8w s3 rp ", run_glds<12, 8, 4, 2, 3, 0, 1>, 192, 128},
      {"glds 192x128 8w s3 p2 ", run_glds<12, 8, 4, 2, 3, 0, 2>, 192, 128},
      {"glds 192x128 8w s4 p2 ", run_glds<12, 8, 4, 2, 4, 0, 2>, 192, 128},
      {"glds 256x128 8w s3 p2 ", run_glds<16, 8, 4, 2, 3, 0, 2>, 256, 128},
      {"glds 352x128 8w s2 rp ", run_glds<22, 8, 2, 4, 2, 0, 1>, 352, 128},
      {"glds 176x256 8w s2 rp ", run_glds<11, 16, 1, 8, 2, 0, 1>, 176, 256},
      {"glds 256x256 8w s2 rp ", run_glds<16, 16, 4, 2, 2, 0, 1>, 256, 256},
      // call 35: 1056 rows = 66 row tiles of 16 = 2 x 3 x 11: 48-row tiles give an EVEN number of row tiles (22), which xcd_tile_order can split 2 x 4
      // (17 or 11 row tiles: every XCD streams the whole activation matrix)
      {"glds 64x48   4w s3    ", run_glds<4, 3, 4, 1, 3>, 64, 48},
      {"glds 64x48   4w s4    ", run_glds<4, 3, 4, 1, 4>, 64, 48},
      {"glds 128x48  4w s3    ", run_glds<8, 3, 4, 1, 3>, 128, 48},
      {"glds 128x48  8w s3    ", run_glds<8, 3, 8, 1, 3>, 128, 48},
      {"glds 64x96   4w s3 rp ", run_glds<4, 6, 2, 2, 3, 0, 1>, 64, 96},
      {"glds 128x96  4w s2 rp ", run_glds<8, 6, 2, 2, 2, 0, 1>, 128, 96},
      {"glds 128x96  8w s3 rp ", run_glds<8, 6, 4, 2, 3, 0, 1>, 128, 96},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    if (only && !strstr(s.what, only)) continue;
    const size_t wbytes = (size_t)s.N * s.K * 2, xbytes = (size_t)s.M * s.K * 2, obytes = (size_t)s.M * s.N * 4;
    const int ncopy = (int)std::min<size_t>(24, (300u << 20) / wbytes + 1);
    uint16_t *Wsrc, *W, *X; float *O, *Oref;
    CK(hipMalloc(&Wsrc, wbytes)); CK(hipMalloc(&W, wbytes * ncopy)); CK(hipMalloc(&X, xbytes)); CK(hipMalloc(&O, obytes)); CK(hipMalloc(&Oref, obytes));
    for (int c = 0; c < ncopy; ++c) {
      hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, st, Wsrc, (size_t)s.N * s.K, 17u + c, 0.05f);
      const size_t total = (size_t)(s.N / 16) * (s.K / 32) * 64;
      hipLaunchKernelGGL((pack_weight_kernel<bf16_t, bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Wsrc, W + (size_t)c * s.N * s.K, s.N, s.K, 0, s.K / 32);
    }
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, st, X, (size_t)s.M * s.K, 99u, 1.0f);
    CK(hipStreamSynchronize(st));
    GemmArgs a = {};
    a.x = reinterpret_cast<const float*>(X); a.x_ld = s.K; a.x_row_mul = 1; a.M = s.M; a.N = s.N; a.K = s.K; a.out_ld = s.N; a.xcd_swz = 1;
    const double flop = 2.0 * s.M * s.N * s.K;
    printf("%s M=%d N=%d K=%d (%.1f GFLOP, %d weight copies)\n", s.what, s.M, s.N, s.K, flop * 1e-9, ncopy);
    std::vector<float> href((size_t)s.M * s.N), hout((size_t)s.M * s.N);
    bool have_ref = false;
    for (const Variant& v : vars) {
      if (s.N % v.bn) { printf("  %s  n/a\n", v.name); continue; }
      a.W = W; a.out = O;
      CK(hipMemsetAsync(O, 0xff, obytes, st));
      if (v.fn(a, st) != 0) { printf("  %s  launch failed\n", v.name); continue; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hout.data(), O, obytes, hipMemcpyDeviceToHost));
      const char* verdict = "reference";
      if (!have_ref) { href = hout; have_ref = true; }
      else verdict = memcmp(href.data(), hout.data(), obytes) == 0 ? "bit-identical" : "DIFFERENT";
      if (strstr(v.name, "compute only")) verdict = "(ablation)";
      if (!strcmp(verdict, "DIFFERENT")) {
        double worst = 0; size_t nbad = 0;
        for (size_t i = 0; i < hout.size(); ++i) { const double d = fabs((double)hout[i] - href[i]); if (!(d == 0)) ++nbad; if (d > worst || d != d) worst = d; }
        printf("    (%zu of %zu values differ, max |d| %.3e)\n", nbad, hout.size(), worst);
      }
      for (int w = 0; w < 5; ++w) { a.W = W + (size_t)(w % ncopy) * s.N * s.K; v.fn(a, st); }
      float best = 1e30f, sum = 0;
      for (int rnd = 0; rnd < 3; ++rnd) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) { a.W = W + (size_t)(r % ncopy) * s.N * s.K; v.fn(a, st); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
      }
      const double us = best * 1e3 / reps;
      printf("  %s  %8.2f us (mean %.2f)  %7.1f TFLOP/s  %5.1f %% of 2.5 PF   %s\n", v.name, us, sum * 1e3 / 3 / reps, flop / us * 1e-6, flop / us * 1e-6 / 2500 * 100, verdict);
    }
    CK(hipFree(Wsrc)); CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(O)); CK(hipFree(Oref));
  }
  return 0;
}
