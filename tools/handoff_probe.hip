// GPU probe (not part of the product): is an intra-launch per-head hand-off cheaper than a kernel boundary for the batch-1 decode
// step's QKV -> self-attention edge ("fewer nodes")?
//   A  two launches per layer, as today: [QKV-like GEMV, 256 workgroups, 3 weight rows per wave] -> [attention-like, 16 heads x 4
//      splits, K/V slice loads issued first, then q/k/v of its head]
//   B  ONE launch: workgroups 0..255 are the producers, 256..319 the consumers. A producer publishes its 12 outputs with write-through
//      (sc1) stores, drains them, and one lane adds the row count to its head's arrival counter (relaxed, agent scope); a consumer issues
//      its K/V loads first, then ONE lane polls the head's counter (sc1 loads + s_sleep, bounded spin), the workgroup rendezvous, sc1
//      loads of q/k/v. Counters are monotonic: the target of launch i is 192 * i (a kernel argument baked into the graph node).
// Both as hipGraphs of N dependent layers; reported: microseconds per layer. The weights / KV arrays are large enough (and distinct per
// layer) that every load is cold, as in the real step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/handoff_probe tools/handoff_probe.hip && tools/handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../parler_tts_amd/csrc/ptts_gemv_kernels.h"   // the product's GEMV kernel, for variants E / F (same chain, real node)
int ptts_fail(int code, const char*, ...) { return code; }
thread_local std::string g_ptts_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int H = 1024, NROWS = 3072, HEADS = 16, SPLITS = 4, LAYERS = 24;

__device__ __forceinline__ float wave_sum_shfl(float v) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// producer body: wave gw computes rows 3*gw .. 3*gw+2 of W[3072][1024] bf16 x x[1024] (x = the previous layer's output, L2-resident)
__device__ __forceinline__ void producer(int wg, const uint4* __restrict__ W, const float* __restrict__ x, float* __restrict__ q, unsigned* counters,
                                         bool publish) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = wg * 4 + wave;
  uint4 w[3][2];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 2; ++c) {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + ((size_t)(3 * gw + r) * 128 + c * 64 + lane)));
      w[r][c] = make_uint4(v.x, v.y, v.z, v.w);
    }
  float4 xv[4];
  for (int c = 0; c < 4; ++c) xv[c] = reinterpret_cast<const float4*>(x)[c * 64 + lane];
  float out = 0.f;
  for (int r = 0; r < 3; ++r) {
    float a = 0.f;
    for (int c = 0; c < 2; ++c) {
      a += __uint_as_float(w[r][c].x << 16) * xv[2 * c].x + __uint_as_float(w[r][c].y << 16) * xv[2 * c].y + __uint_as_float(w[r][c].z << 16) * xv[2 * c + 1].x +
           __uint_as_float(w[r][c].w << 16) * xv[2 * c + 1].y;
    }
    a = wave_sum_shfl(a);
    if (lane == r) out = a * 1e-3f + 1.0f;
  }
  if (lane < 3) {
    if (publish) __hip_atomic_store(q + 3 * gw + lane, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
    else q[3 * gw + lane] = out;
  }
  if (publish) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {  // rows 12*wg .. 12*wg+11: one or two heads of one section
      const int r0 = 12 * wg, r1 = r0 + 11, h0 = (r0 % 1024) / 64, h1 = (r1 % 1024) / 64;
      int n0 = 0, n1 = 0;
      for (int r = r0; r <= r1; ++r) { if ((r % 1024) / 64 == h0) ++n0; else ++n1; }
      __hip_atomic_fetch_add(counters + h0, (unsigned)n0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n1) __hip_atomic_fetch_add(counters + h1, (unsigned)n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// consumer body: (head, split) streams its 24 KB K/V slice first, then needs q_h / k_h / v_h (3 x 64 floats)
__device__ __forceinline__ void consumer(int cw, const uint4* __restrict__ KV, const float* __restrict__ q, float* __restrict__ out, unsigned* counters,
                                         unsigned target, bool wait, int* err) {
  const int h = cw / SPLITS, s = cw % SPLITS, tid = threadIdx.x;
  uint4 kv[6];
  for (int u = 0; u < 6; ++u) kv[u] = KV[((size_t)(h * SPLITS + s) * 6 + u) * 256 + tid];
  if (wait) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(counters + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *err = 1; break; }  // bounded: never hangs the GPU
      }
    }
    __syncthreads();
  }
  float qv = 0.f;
  if (tid < 192) {
    const float* p = q + (tid / 64) * 1024 + h * 64 + (tid & 63);
    qv = wait ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  }
  float acc = qv;
  for (int u = 0; u < 6; ++u) acc += __uint_as_float(kv[u].x << 16) * 1e-6f + __uint_as_float(kv[u].w << 16) * 1e-6f;
  acc = wave_sum_shfl(acc);
  __shared__ float red[4];
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid < 64) out[(h * SPLITS + s) * 64 + tid] = red[0] + red[1] + red[2] + red[3] + qv;
}

__global__ void __launch_bounds__(256) k_producer(const uint4* W, const float* x, float* q) { producer(blockIdx.x, W, x, q, nullptr, false); }
__global__ void __launch_bounds__(256) k_consumer(const uint4* KV, const float* q, float* out) { consumer(blockIdx.x, KV, q, out, nullptr, 0, false, nullptr); }
__global__ void __launch_bounds__(256) k_merged(const uint4* W, const float* x, float* q, const uint4* KV, float* out, unsigned* counters, unsigned target, int* err) {
  if (blockIdx.x < 256) producer(blockIdx.x, W, x, q, counters, true);
  else consumer(blockIdx.x - 256, KV, q, out, counters, target, true, err);
}
// the rest of the layer as one stand-in node: reads the attention output, writes the next layer's x (keeps the chain dependent)
__global__ void k_rest(const float* in, float* x) { const int i = blockIdx.x * 256 + threadIdx.x; x[i] = in[i % (HEADS * SPLITS * 64)] * 0.5f + 0.1f; }

int main() {
  uint4 *W, *KV; float *x, *q, *out; unsigned* counters; int* err;
  const size_t wbytes = (size_t)LAYERS * NROWS * H * 2, kvbytes = (size_t)LAYERS * HEADS * SPLITS * 6 * 256 * 16;
  CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&KV, kvbytes)); CK(hipMalloc(&x, H * 4)); CK(hipMalloc(&q, NROWS * 4));
  CK(hipMalloc(&out, HEADS * SPLITS * 64 * 4)); CK(hipMalloc(&counters, 64 * 4)); CK(hipMalloc(&err, 4));
  CK(hipMemset(W, 0x3c, wbytes)); CK(hipMemset(KV, 0x3c, kvbytes)); CK(hipMemset(x, 0, H * 4)); CK(hipMemset(err, 0, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int REPS = 40;
  float *gamma, *beta; CK(hipMalloc(&gamma, H * 4)); CK(hipMalloc(&beta, H * 4)); CK(hipMemset(gamma, 0, H * 4)); CK(hipMemset(beta, 0, H * 4));
  for (int variant = 0; variant < 6; ++variant) {
    CK(hipMemset(counters, 0, 64 * 4));
    hipGraph_t g; hipGraphExec_t ex;
    // the graph is replayed REPS + 3 times: counters are monotonic over the whole run, so each replay gets its own graph of targets? No:
    // kernel arguments are baked per node. One graph = LAYERS layers with targets base + 192 * (l + 1); the counters are reset by a
    // memset node at the head of the graph (outside the timed chain's critical path it is one more node: same for every variant).
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(counters, 0, 64 * 4, st));
    for (int l = 0; l < LAYERS; ++l) {
      const uint4* Wl = W + (size_t)l * NROWS * 128; const uint4* KVl = KV + (size_t)l * HEADS * SPLITS * 6 * 256;
      if (variant == 0) { hipLaunchKernelGGL(k_producer, dim3(256), dim3(256), 0, st, Wl, x, q); hipLaunchKernelGGL(k_consumer, dim3(64), dim3(256), 0, st, KVl, q, out); }
      else if (variant == 1) hipLaunchKernelGGL(k_merged, dim3(320), dim3(256), 0, st, Wl, x, q, KVl, out, counters, 192u * (l + 1), err);
      else if (variant == 2) hipLaunchKernelGGL(k_producer, dim3(256), dim3(256), 0, st, Wl, x, q);
      else if (variant == 3) hipLaunchKernelGGL(k_consumer, dim3(64), dim3(256), 0, st, KVl, q, out);
      else {  // E: the product's LN1 + QKV node (LayerNorm prologue wave + 4 GEMV waves, 3 rows per wave) in the producer's place; F: + consumer
        GemvArgs ga = {};
        ga.W = Wl; ga.x = x; ga.x_ld = H; ga.gamma = gamma; ga.beta = beta; ga.out = q; ga.out_ld = NROWS; ga.N = NROWS; ga.K = H; ga.M = 1; ga.invK = 1.0f / H;
        ptts_klaunch(gemv_kernel<bf16_t, 2, 3, GV_LN, GV_STORE, 1, 1, false>, dim3(256), dim3(320), H * 2, st, ga);
        if (variant == 5) hipLaunchKernelGGL(k_consumer, dim3(64), dim3(256), 0, st, KVl, q, out);
      }
      hipLaunchKernelGGL(k_rest, dim3(4), dim3(256), 0, st, out, x);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ex, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const char* names[] = {"A two launches (QKV GEMV -> attention) + rest", "B one launch with per-head hand-off + rest", "C producer only + rest", "D consumer only + rest",
                           "E product gemv_kernel<LN,QKV> only + rest", "F product gemv_kernel<LN,QKV> -> attention-like + rest"};
    printf("[handoff_probe] %-52s %.2f us per layer%s\n", names[variant], ms * 1e3 / REPS / LAYERS, herr ? "  (SPIN LIMIT HIT: hand-off did not complete)" : "");
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  float hq[4]; CK(hipMemcpy(hq, q, 16, hipMemcpyDeviceToHost));
  printf("[handoff_probe] q[0..3] = %g %g %g %g\n", hq[0], hq[1], hq[2], hq[3]);
  return 0;
}
