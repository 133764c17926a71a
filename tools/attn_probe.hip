// GPU probe (not a test, not the product): where does a launch of prefill_attn_mfma_kernel spend its time? The kernel at the 32-utterance prefill shape
// (33 query rows x 16 heads x 32 utterances, 3 waves per workgroup, 512 workgroups), self (33 keys) and cross (64 keys) flavours, launched back to back
// on one stream with phases compiled out (template parameter ABL of the kernel: 1 no MFMAs, 2 no K / V / mask loads, 4 no LDS staging, 8 no store,
// 16 exit at entry, 32 no query loads). us per launch by HIP events over 400 launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 -Iinclude tools/attn_probe.hip -o tools/attn_probe
#include "../parler_tts_amd/csrc/ptts_common.h"
#include "../parler_tts_amd/csrc/ptts_lm_kernels.h"
#include <vector>
int ptts_fail(int code, const char* fmt, ...) { fprintf(stderr, "ptts_fail(%d): %s\n", code, fmt); return code; }  // the library's error sink, stubbed

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ABL> float run(const AttnArgs& a, dim3 grid, hipStream_t st, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<bf16_t, 3, ABL>, grid, dim3(192), st, a);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) ptts_launch_prefill_attn_kernel(prefill_attn_mfma_kernel<bf16_t, 3, ABL>, grid, dim3(192), st, a);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const int B = 32, Q = 33, nh = 16, H = 1024, QKV = 3072, cap = 940, N = 64;
  float* q; bf16_t *kc, *vc, *out; DevDims* dims;
  CK(hipMalloc(&q, (size_t)B * Q * QKV * 4)); CK(hipMalloc(&kc, (size_t)B * nh * cap * 64 * 2)); CK(hipMalloc(&vc, (size_t)B * nh * cap * 64 * 2));
  CK(hipMalloc(&out, (size_t)B * Q * H * 2)); CK(hipMalloc(&dims, sizeof(DevDims)));
  std::vector<float> hq((size_t)B * Q * QKV);
  for (size_t i = 0; i < hq.size(); ++i) hq[i] = 0.01f * (float)((i * 2654435761u >> 20) & 255) - 1.28f;
  CK(hipMemcpy(q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
  std::vector<bf16_t> hk((size_t)B * nh * cap * 64);
  for (size_t i = 0; i < hk.size(); ++i) hk[i] = (bf16_t)(0x3c00 + ((i * 40503u >> 8) & 0x1ff));
  CK(hipMemcpy(kc, hk.data(), hk.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
  DevDims hd = {}; hd.P = 32; hd.N = N;
  CK(hipMemcpy(dims, &hd, sizeof(hd), hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int cross = 0; cross < 2; ++cross) {
    AttnArgs a = {};
    a.q = q; a.q_ld = QKV; a.kcache = kc; a.vcache = vc; a.cap = cross ? N : cap; a.kv_bound = a.cap; a.dims = dims; a.mask = nullptr; a.mask_ld = 40;
    a.hostP = hd.P; a.hostN = hd.N; a.S = 1; a.Q = Q; a.nheads = nh; a.H = H; a.kv_heads = nh; a.n_rep = 1; a.cross = cross; a.scale = 0.125f; a.direct_out = out; a.out_fo = 1;
    const dim3 grid((Q + 47) / 48, nh, B);
    const int reps = 400;
    printf("[attn_probe %s] full %.2f | no MFMA %.2f | no K/V/mask loads %.2f | no q loads %.2f | no loads at all %.2f | no LDS staging %.2f | no store %.2f | "
           "no loads + no MFMA %.2f | loads only (no MFMA, no staging, no store) %.2f | empty %.2f us per launch\n", cross ? "cross 64 keys" : "self 33 keys",
           run<0>(a, grid, st, reps), run<1>(a, grid, st, reps), run<2>(a, grid, st, reps), run<32>(a, grid, st, reps), run<34>(a, grid, st, reps),
           run<4>(a, grid, st, reps), run<8>(a, grid, st, reps), run<35>(a, grid, st, reps), run<13>(a, grid, st, reps), run<16>(a, grid, st, reps));
  }
  return 0;
}
