// GPU probe (not part of the product): can a node of the single-utterance decode step have its weights waiting in the XCD's L2?
// The step is a chain of dependent kernel nodes; HBM is idle ~80 % of it (0.78 GB in 560 us), in particular during every kernel boundary.
// A chain of GEMV-like nodes (tools/kernarg_probe.hip's node: reads the row the previous node wrote, streams its own weight rows):
//   cold      weights rotate over 680 MB: HBM-cold, MALL-cold (the real step: 0.73 GB of weights per pass)
//   rot=1     every node reads the SAME matrix: resident in the L2 of the XCD that read it last time IF lines survive a kernel boundary
//   rot=4/16  32 MB in rotation: fits the 256 MB MALL, not the 8 x 4 MB L2s -> separates MALL residency from L2 residency
//   prefetch  cold, and G extra workgroups of node i touch the rows node i + 1 will read - workgroup G + j touches what workgroup j of the next
//             node reads (observed placement: block b runs on XCD b % 8, G % 8 == 0 -> the same XCD's L2), plain loads, results discarded
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/prefetch_probe tools/prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct NArgs {
  const u32x4* W;      // [N][1024] bf16: 128 x 16 B per row
  const u32x4* Wnext;  // the next node's matrix (prefetch workgroups), or null
  const float* x;      // [1024] fp32, written by the previous node
  float* out;          // [N]
  float* sink;         // never written (the prefetch loads must not be optimised away)
  int N, G, nt;
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int R>
__global__ void __launch_bounds__(256) node_kernel(NArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= a.G) {  // prefetch workgroup: touch the rows of workgroup (blockIdx.x - G) of the next node
    const int row0 = ((blockIdx.x - a.G) * 4 + wave) * R;
    if (row0 >= a.N) return;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc |= a.Wnext[(size_t)(row0 + r) * 128 + c * 64 + lane];
    if ((acc.x & acc.y & acc.z & acc.w) == 0x12345678u) a.sink[lane] = 1.f;  // never true (the matrices are zero-filled)
    return;
  }
  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= a.N) return;
  u32x4 w[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32x4* p = a.W + ((size_t)(row0 + r) * 128 + c * 64 + lane);
      w[r][c] = a.nt ? __builtin_nontemporal_load(p) : *p;
    }
  f32x4 xv[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    xv[c][0] = *reinterpret_cast<const f32x4*>(a.x + (c * 64 + lane) * 8);
    xv[c][1] = *reinterpret_cast<const f32x4*>(a.x + (c * 64 + lane) * 8 + 4);
  }
  float keep = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32x4 v = w[r][c];
      acc += bf_lo(v.x) * xv[c][0].x + bf_hi(v.x) * xv[c][0].y + bf_lo(v.y) * xv[c][0].z + bf_hi(v.y) * xv[c][0].w;
      acc += bf_lo(v.z) * xv[c][1].x + bf_hi(v.z) * xv[c][1].y + bf_lo(v.w) * xv[c][1].z + bf_hi(v.w) * xv[c][1].w;
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    keep = lane == r ? acc : keep;
  }
  if (lane < R) a.out[row0 + lane] = keep + 1.0f;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NODES = 170, REPS = 100;
  const size_t slot = (size_t)4096 * 1024 * 2;  // 8 MB
  char* W; float *xa, *xb, *sink;
  CK(hipMalloc(&W, slot * 85)); CK(hipMemset(W, 0, slot * 85));
  CK(hipMalloc(&xa, 4096 * 4)); CK(hipMalloc(&xb, 4096 * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(xa, 0, 4096 * 4)); CK(hipMemset(xb, 0, 4096 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; int N, R, rot, prefetch, nt; } cases[] = {
      {"2 MB cold (rot 85)            nt", 1024, 1, 85, 0, 1}, {"2 MB rot 1 (L2-resident?)     nt", 1024, 1, 1, 0, 1},
      {"2 MB rot 1                 plain", 1024, 1, 1, 0, 0},  {"2 MB rot 16 (MALL-resident)   nt", 1024, 1, 16, 0, 1},
      {"2 MB cold + prefetch next     nt", 1024, 1, 85, 1, 1}, {"2 MB cold + prefetch next  plain", 1024, 1, 85, 1, 0},
      {"8 MB cold (rot 85)            nt", 4096, 4, 85, 0, 1}, {"8 MB rot 1 (L2-resident?)     nt", 4096, 4, 1, 0, 1},
      {"8 MB rot 1                 plain", 4096, 4, 1, 0, 0},  {"8 MB rot 4 (MALL-resident)    nt", 4096, 4, 4, 0, 1},
      {"8 MB cold + prefetch next     nt", 4096, 4, 85, 1, 1}, {"8 MB cold + prefetch next  plain", 4096, 4, 85, 1, 0},
      {"2 MB cold (rot 85)            nt", 1024, 1, 85, 0, 1},
      {"8 MB rot 2  ( 16 MB in rotation) nt", 4096, 4, 2, 0, 1},  {"8 MB rot 8  ( 64 MB: MALL only) nt", 4096, 4, 8, 0, 1},
      {"8 MB rot 16 (128 MB: MALL only) nt", 4096, 4, 16, 0, 1}, {"8 MB rot 24 (192 MB: MALL only) nt", 4096, 4, 24, 0, 1},
      {"8 MB rot 40 (320 MB: > MALL)    nt", 4096, 4, 40, 0, 1}, {"2 MB rot 64 (128 MB: MALL only) nt", 1024, 1, 64, 0, 1},
      {"8 MB rot 16 (128 MB)         plain", 4096, 4, 16, 0, 0},
  };
  for (auto& cs : cases) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NODES; ++i) {
      NArgs a = {};
      a.W = reinterpret_cast<const u32x4*>(W + (size_t)(i % cs.rot) * slot);
      a.Wnext = reinterpret_cast<const u32x4*>(W + (size_t)((i + 1) % cs.rot) * slot);
      a.x = (i & 1) ? xb : xa; a.out = (i & 1) ? xa : xb; a.N = cs.N; a.sink = sink; a.nt = cs.nt;
      a.G = cs.N / (4 * cs.R);
      const dim3 grid(cs.prefetch ? 2 * a.G : a.G), blk(256);
      if (cs.R == 1) hipLaunchKernelGGL(node_kernel<1>, grid, blk, 0, st, a);
      else hipLaunchKernelGGL(node_kernel<4>, grid, blk, 0, st, a);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ex, st);
    CK(hipStreamSynchronize(st));
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < REPS; ++i) hipGraphLaunch(ex, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best; sum += ms;
    }
    printf("[prefetch_probe] %-34s: %.3f us per node (best of 3; mean %.3f)\n", cs.name, best * 1e3f / REPS / NODES, sum / 3 * 1e3f / REPS / NODES);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  return 0;
}
