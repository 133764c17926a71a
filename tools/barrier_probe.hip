// Standalone probe (not part of the product): what would a persistent decode-step kernel pay per phase?
//   1. grid barrier latency (agent-scope atomic counter, one arrival per workgroup, bounded spin: can not hang)
//   2. barrier + a read of data another workgroup (another XCD) wrote just before the barrier (producer -> consumer)
//   3. the same with a weight tile prefetched into registers BEFORE the barrier (what persistence buys)
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/barrier_probe tools/barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) { ok = false; *fail = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// Flag-array barrier: every WG publishes its epoch in its own word (no same-address RMW serialisation); wave 0 of every WG
// polls all flags with one relaxed agent-scope load per lane group. flags: [nwg] words, nwg <= 512.
__device__ __forceinline__ bool flag_barrier(unsigned* flags, int nwg, unsigned epoch, int* fail) {
  __syncthreads();  // all stores of this WG issued
  bool ok = true;
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int spins = 0;
    while (true) {
      bool all = true;
      for (int i = threadIdx.x; i < nwg; i += 64)
        all = all && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
      if (__all(all)) break;
      if (++spins > (1 << 20)) { ok = false; *fail = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// mode 0: barrier only; mode 1: + producer/consumer 4 KB activation vector (every WG reads all of it, like LN);
// mode 2: + 32 KB weight tile per WG streamed AFTER the barrier (cold); mode 3: weight tile loaded BEFORE the barrier
__global__ void __launch_bounds__(256) phases_kernel(unsigned* ctr, int* fail, float* act, const uint4* w, size_t w_stride_phase,
                                                     float* sink, int nphase, int mode, unsigned* flags) {
  const int nwg = gridDim.x, wg = blockIdx.x, t = threadIdx.x;
  const bool use_flags = mode >= 4; mode &= 3;
  float acc = 0.f;
  __shared__ float red[256];
  for (int p = 0; p < nphase; ++p) {
    uint4 wt[8];
    const uint4* wp = w + (size_t)(p & 7) * w_stride_phase + (size_t)wg * 2048;  // 32 KB per WG per phase, 8 phase slots
    if (mode == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) wt[u] = ldnt(wp + u * 256 + t);
    }
    // produce: WG writes its 4 floats of the activation vector (1024 floats over 256 WGs)
    if (mode >= 1 && t < 4) act[(p & 1) * 1024 + (wg * 4 + t) % 1024] = acc + (float)p;
    if (use_flags) { if (!flag_barrier(flags, nwg, (unsigned)(p + 1), fail)) return; }
    else if (!grid_barrier(ctr, (unsigned)(p + 1) * nwg, fail)) return;
    if (mode >= 1) {
      const float4 v = reinterpret_cast<const float4*>(act + (p & 1) * 1024)[t];  // consume the whole vector
      float s = v.x + v.y + v.z + v.w;
      if (mode == 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wt[u] = ldnt(wp + u * 256 + t);
      }
      if (mode >= 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s += __uint_as_float(wt[u].x ^ wt[u].y ^ wt[u].z ^ wt[u].w) * 1e-30f;
      }
      red[t] = s; __syncthreads();
      for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
      acc = red[0] * 1e-3f;
      __syncthreads();
    }
  }
  if (t == 0) sink[wg] = acc;
}

// 4. floor of a dependent kernel chain with a real hand-off: every WG reads the whole 4 KB vector the previous kernel wrote
// (cold: other XCDs wrote it), reduces it, writes its own slice of the next vector. No weights, no MFMA.
__global__ void relay_kernel(const float* __restrict__ in, float* __restrict__ out, int slice) {
  const int t = threadIdx.x;
  __shared__ float red[16];
  float s = 0.f;
  if (t < 256) { const float4 v = reinterpret_cast<const float4*>(in)[t]; s = v.x + v.y + v.z + v.w; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (int)blockDim.x / 64; ++w) tot += red[w];
  if (t < slice) out[(blockIdx.x * slice + t) & 1023] = tot * 1e-3f + 1.f;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  {
    float* bufs; CK(hipMalloc(&bufs, 2 * 4096)); CK(hipMemset(bufs, 0, 8192));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wg : {16, 64, 256}) for (int th : {256, 512, 1024}) {
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 170; ++i) hipLaunchKernelGGL(relay_kernel, dim3(wg), dim3(th), 0, st, bufs + (i & 1) * 1024, bufs + ((i + 1) & 1) * 1024, 1024 / wg > 0 ? (1024 / wg > th ? th : 1024 / wg) : 1);
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      for (int i = 0; i < 20; ++i) hipGraphLaunch(ex, st);
      hipStreamSynchronize(st);
      hipEventRecord(a, st);
      for (int i = 0; i < 200; ++i) hipGraphLaunch(ex, st);
      hipEventRecord(b, st); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("relay chain, %3d WGs x %4d threads: %.2f us per dependent node\n", wg, th, ms * 1e3f / 200 / 170);
      hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
  }
  unsigned* ctr; int* fail; float *act, *sink; uint4* w; unsigned* flags;
  CK(hipMalloc(&flags, 4096));
  const size_t phase_bytes = (size_t)512 * 32768;  // up to 512 WGs x 32 KB
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&act, 2 * 1024 * 4)); CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&w, phase_bytes * 8)); CK(hipMemset(w, 0, phase_bytes * 8)); CK(hipMemset(act, 0, 8192));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("CUs %d\n", prop.multiProcessorCount);
  const int nphase = 2000;
  for (int nwg : {64, 128, 256, 512}) {
    for (int mode = 0; mode < 8; ++mode) {
      float best = 1e9f; int failed = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(fail, 0, 4, st)); CK(hipMemsetAsync(flags, 0, 4096, st));
        size_t stride16 = phase_bytes / 16;
        void* args[] = {&ctr, &fail, &act, &w, &stride16, &sink, (void*)&nphase, &mode, &flags};
        CK(hipEventRecord(a, st));
        CK(hipLaunchCooperativeKernel((void*)phases_kernel, dim3(nwg), dim3(256), args, 0, st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        CK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
        if (failed) break;
      }
      printf("WGs %3d %s mode %d (%s): %.2f us per phase%s\n", nwg, mode >= 4 ? "FLAGS  " : "COUNTER", mode & 3,
             (mode & 3) == 0 ? "barrier only" : (mode & 3) == 1 ? "barrier + act produce/consume" : (mode & 3) == 2 ? "+ 32KB/WG weights after barrier"
                                                                                                   : "+ 32KB/WG weights prefetched before barrier",
             best * 1e3f / nphase, failed ? "  [SPIN LIMIT HIT]" : "");
    }
  }
  return 0;
}
