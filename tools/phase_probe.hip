// Standalone latency probe (not part of the product): phase breakdown of the decode GEMM kernel in shader cycles
// (s_memtime) + wall time per dependent launch (hipEvents), to separate launch floor / HBM latency / LDS staging.
#define PTTS_TIMING 1
#include "../parler_tts_amd/csrc/ptts_lm_kernels.h"
#include <vector>
#include <stdio.h>
thread_local std::string g_ptts_err;
int ptts_fail(int code, const char*, ...) { return code; }

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void spin_kernel(long long cycles, long long* out) {
  long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memtime() - t0;
}
__global__ void chase_kernel(const int* next, int n, int* out, long long* cyc) {  // dependent global loads: latency per hop
  int i = 0; long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < n; ++k) i = next[i];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[0] = i; cyc[0] = t1 - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename F> float time_launches(F f, int n, hipStream_t st) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / n;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int H = 1024, F = 4096;
  void *W; float *x, *out, *gamma, *beta; long long* dbg;
  CK(hipMalloc(&W, (size_t)F * H * 2 * 4)); CK(hipMalloc(&x, F * 4 * 32)); CK(hipMalloc(&out, F * 4 * 32));
  CK(hipMalloc(&gamma, H * 4)); CK(hipMalloc(&beta, H * 4)); CK(hipMalloc(&dbg, 64 * 8));
  CK(hipMemset(W, 0, (size_t)F * H * 2 * 4)); CK(hipMemset(x, 0, F * 4 * 32)); CK(hipMemset(gamma, 0, H * 4)); CK(hipMemset(beta, 0, H * 4));
  // 1. clock: spin for 2.1e6 shader cycles, compare with wall time
  { long long* o; CK(hipMalloc(&o, 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, st); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 2100000LL, o); hipEventRecord(b, st); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); long long c; hipMemcpy(&c, o, 8, hipMemcpyDeviceToHost);
      printf("spin: %lld s_memtime ticks in %.3f ms -> %.1f MHz tick rate\n", c, ms, c / ms / 1e3);
    } }
  // 2. launch floor
  printf("empty kernel, dependent launches: %.2f us each\n", time_launches([&] { hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st, (int*)nullptr); }, 2000, st));
  // 2b. hipGraph replay floor: 195 empty dependent kernel nodes per replay (the decode step has 195 nodes)
  for (int wg : {64, 256}) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 195; ++i) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(256), 0, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    float us = time_launches([&] { hipGraphLaunch(ex, st); }, 200, st);
    printf("graph of 195 empty kernels (%d WGs each): %.1f us per replay = %.2f us per node\n", wg, us, us / 195);
  }
  // 3. dependent-load latency: pointer chase over 64 MB (HBM) and 256 KB (L2)
  for (size_t bytes : {(size_t)256 << 10, (size_t)64 << 20}) {
    int n = (int)(bytes / 4); std::vector<int> h(n);
    const int stride = 4099 * 16;  // co-prime-ish hop in ints (64 KB-ish jumps)
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i + stride) % n);
    int* d; int* o; long long* c; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&o, 4)); CK(hipMalloc(&c, 8));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, st, d, 2000, o, c); hipStreamSynchronize(st);
    hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, st, d, 2000, o, c); hipStreamSynchronize(st);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("pointer chase over %zu KB: %.0f ticks per dependent load\n", bytes >> 10, cy / 2000.0);
    hipFree(d);
  }
  // 4. the decode GEMMs (bf16) at M = 1 and M = 32
  struct Case { const char* name; int N, K, pro; } cases[] = {{"LN+QKV   N=3072 K=1024", 3072, 1024, 1}, {"LN+crossQ N=1024 K=1024", 1024, 1024, 1},
                                                               {"fc2      N=1024 K=4096", 1024, 4096, 0}, {"LN+fc1   N=4096 K=1024", 4096, 1024, 1}};
  for (int M : {1, 32}) {
    for (auto& cs : cases) {
      GemmArgs a = {}; a.W = W; a.x = x; a.x_ld = cs.K; a.x_row_mul = 1; a.gamma = gamma; a.beta = beta; a.out = out; a.out_ld = cs.N;
      a.M = M; a.N = cs.N; a.K = cs.K; a.dbg = dbg; a.invK = 1.0f / cs.K;
      const int nfrag = cs.K / 32;
      const int wmax = (cs.pro == 0 && M <= 16) ? 16 : 8;
      int Wv = 0;
      for (int w = wmax; w >= 2; --w) if (nfrag % (8 * w) == 0) { Wv = w; break; }
      a.frags_per_wave = nfrag / Wv;
      const size_t row_bytes = (size_t)cs.K * 2 + 16;
      int rpp = M < 32 ? M : 32;
      while (rpp > 1 && rpp * row_bytes + (size_t)Wv * (rpp > 16 ? 2 : 1) * 1024 > 160 * 1024 - 1024) --rpp;
      a.rows_per_pass = rpp;
      const int mtp = rpp > 16 ? 2 : 1;
      const size_t sh = rpp * row_bytes + (size_t)Wv * mtp * 1024;
      auto launch = [&] {
        if (cs.pro && mtp == 1) { hipFuncSetAttribute((const void*)&gemm_strip_kernel_bv<bf16_t, PRO_LN, EPI_STORE, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipLaunchKernelGGL((gemm_strip_kernel_bv<bf16_t, PRO_LN, EPI_STORE, 1, true>), dim3(cs.N / 16), dim3(Wv * 64), sh, st, a); }
        else if (cs.pro) { hipFuncSetAttribute((const void*)&gemm_strip_kernel_bv<bf16_t, PRO_LN, EPI_STORE, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipLaunchKernelGGL((gemm_strip_kernel_bv<bf16_t, PRO_LN, EPI_STORE, 2, true>), dim3(cs.N / 16), dim3(Wv * 64), sh, st, a); }
        else if (mtp == 1) { hipFuncSetAttribute((const void*)&gemm_strip_kernel_bv<bf16_t, PRO_PLAIN, EPI_STORE, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipLaunchKernelGGL((gemm_strip_kernel_bv<bf16_t, PRO_PLAIN, EPI_STORE, 1, true>), dim3(cs.N / 16), dim3(Wv * 64), sh, st, a); }
        else { hipFuncSetAttribute((const void*)&gemm_strip_kernel_bv<bf16_t, PRO_PLAIN, EPI_STORE, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          hipLaunchKernelGGL((gemm_strip_kernel_bv<bf16_t, PRO_PLAIN, EPI_STORE, 2, true>), dim3(cs.N / 16), dim3(Wv * 64), sh, st, a); }
      };
      float us = time_launches(launch, 1000, st);
      long long t[8]; hipMemcpy(t, dbg, 64, hipMemcpyDeviceToHost);
      printf("M=%2d %s W=%d rpp=%d: %.2f us/launch | ticks (last pass): issue-W %lld, stage %lld, barrier %lld, mfma %lld, reduce+store %lld, total %lld\n", M, cs.name, Wv, rpp, us,
             t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
    }
  }
  return 0;
}
