// GPU probe (not part of the product): what does the kernel-argument fetch cost a node of the single-utterance decode step, and does
// gfx950's KERNARG PRELOAD (the CP writes the first <= 14 argument dwords into user SGPRs before the first wave starts; LLVM:
// -mllvm -amdgpu-kernarg-preload-count=N, only for arguments passed as scalars / pointers - a by-value struct stays behind s_load) remove it?
// A chain of dependent graph nodes, each a GEMV-like kernel as the step's nodes are (reads the row the previous node wrote: cold, another
// XCD wrote it; streams its own cold weight rows; one value per row out), in two argument forms:
//   S  one by-value struct (the product's form): the wave's first instructions are s_load of the struct, then the dependent loads;
//   F  the same fields as scalar arguments: preloaded when this file is built with the flag, s_load otherwise.
// plus a trivial relay node (one workgroup, one value) in both forms = the dependent-node floor (bench.py NODE_FLOOR_US).
// Build twice:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kernarg_probe_base tools/kernarg_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 -o tools/kernarg_probe_pre tools/kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct NArgs {
  const u32x4* W;   // [N][K] bf16, K = 1024: 128 x 16 B per row
  const float* x;   // [1024] fp32, written by the previous node
  float* out;       // [N] (N >= 1024: the next node reads the first 1024)
  int N;
  int rot;          // a few more fields, like the product's 100+ byte structs
  const float* gamma;
  const float* beta;
  const float* resid;
  float invK;
  int pad[7];
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int R>
__device__ __forceinline__ void node_body(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ out, int N, const float* resid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= N) return;
  u32x4 w[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) w[r][c] = __builtin_nontemporal_load(W + ((size_t)(row0 + r) * 128 + c * 64 + lane));
  f32x4 xv[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    xv[c][0] = *reinterpret_cast<const f32x4*>(x + (c * 64 + lane) * 8);
    xv[c][1] = *reinterpret_cast<const f32x4*>(x + (c * 64 + lane) * 8 + 4);
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
  if (lane < R) out[row0 + lane] = keep + (resid ? resid[row0 + lane] : 0.f) + 1.0f;
}

template <int R> __global__ void __launch_bounds__(256) node_struct(NArgs a) { node_body<R>(a.W, a.x, a.out, a.N, a.resid); }
template <int R>
__global__ void __launch_bounds__(256) node_flat(const u32x4* W, const float* x, float* out, int N, const float* resid, int rot, const float* gamma, const float* beta,
                                                 float invK, int p0, int p1, int p2, int p3) {
  node_body<R>(W, x, out, N, resid);
}
__global__ void relay_struct(NArgs a) { if (threadIdx.x == 0) a.out[0] = a.x[0] + 1.0f; }
__global__ void relay_flat(const u32x4* W, const float* x, float* out, int N, const float* resid) { if (threadIdx.x == 0) out[0] = x[0] + 1.0f; }

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int NODES = 170, REPS = 100;
  const size_t slot = (size_t)4096 * 1024 * 2;  // 8 MB
  char* W; float *xa, *xb;
  CK(hipMalloc(&W, slot * 85)); CK(hipMemset(W, 0, slot * 85));
  CK(hipMalloc(&xa, 4096 * 4)); CK(hipMalloc(&xb, 4096 * 4)); CK(hipMemset(xa, 0, 4096 * 4)); CK(hipMemset(xb, 0, 4096 * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; int N, R, flat, relay; } cases[] = {
      {"relay  struct", 0, 0, 0, 1}, {"relay  flat  ", 0, 0, 1, 1},
      {"gemv 1024 rows (2 MB) struct", 1024, 1, 0, 0}, {"gemv 1024 rows (2 MB) flat  ", 1024, 1, 1, 0},
      {"gemv 4096 rows (8 MB) struct", 4096, 4, 0, 0}, {"gemv 4096 rows (8 MB) flat  ", 4096, 4, 1, 0},
      {"gemv 1024 rows (2 MB) struct", 1024, 1, 0, 0}, {"gemv 1024 rows (2 MB) flat  ", 1024, 1, 1, 0},
  };
  for (auto& cs : cases) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NODES; ++i) {
      NArgs a = {};
      a.W = reinterpret_cast<const u32x4*>(W + (size_t)(i % 85) * slot); a.x = (i & 1) ? xb : xa; a.out = (i & 1) ? xa : xb; a.N = cs.N; a.resid = nullptr;
      const dim3 grid(cs.relay ? 1 : cs.N / (4 * cs.R)), blk(cs.relay ? 64 : 256);
      if (cs.relay) {
        if (cs.flat) hipLaunchKernelGGL(relay_flat, grid, blk, 0, st, a.W, a.x, a.out, a.N, a.resid);
        else hipLaunchKernelGGL(relay_struct, grid, blk, 0, st, a);
      } else if (cs.R == 1) {
        if (cs.flat) hipLaunchKernelGGL(node_flat<1>, grid, blk, 0, st, a.W, a.x, a.out, a.N, a.resid, 0, a.gamma, a.beta, 0.f, 0, 0, 0, 0);
        else hipLaunchKernelGGL(node_struct<1>, grid, blk, 0, st, a);
      } else {
        if (cs.flat) hipLaunchKernelGGL(node_flat<4>, grid, blk, 0, st, a.W, a.x, a.out, a.N, a.resid, 0, a.gamma, a.beta, 0.f, 0, 0, 0, 0);
        else hipLaunchKernelGGL(node_struct<4>, grid, blk, 0, st, a);
      }
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
    printf("[kernarg_probe %s] %-30s: %.3f us per node (best of 3; mean %.3f)\n",
#ifdef PROBE_TAG
           PROBE_TAG,
#else
           "?",
#endif
           cs.name, best * 1e3f / REPS / NODES, sum / 3 * 1e3f / REPS / NODES);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  return 0;
}
