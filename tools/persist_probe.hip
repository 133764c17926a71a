// GPU probe (standalone, no torch, no libptts): ONE Mini-v1 decoder layer at batch 1 as a single PERSISTENT launch against the same work
// as seven dependent launches in a hipGraph (recipe rows `allgather`, `prefetch-credit`, `engine-vs-launches` of
// /opt/skills/guides/MI355X_MICROARCH.md's price list). It answers one question with real byte counts: does replacing the 7 kernel boundaries of a
// layer (LN1+QKV, self-attention, combine+out_proj, LN2+Mx, softmax+Up, LN3+fc1, fc2 - DESIGN.md section 4.1) by 7 in-launch all-to-all
// hand-offs shorten the 25.4 us per layer the product's step takes?
//
// Both variants run the SAME phase bodies (row-per-wave GEMV over cold bf16 weights: a 24-layer weight set of 725 MB is streamed once per
// step; the attention phase streams the layer's K/V slice, 1.9 MB at context 460, and emits 4 split partials per head):
//   launches   : 7 kernels per layer x 24 layers captured in one hipGraph (168 dependent nodes); a phase reads its input vector as plain
//                fp32 and writes its outputs plainly.
//   persistent : one launch of 256 workgroups x 256 threads (one per CU). An edge = the producing phase's output vector, published element
//                by element as naturally aligned 8-byte {fp32 data, u32 tag} granules with ONE agent-scope (sc1, write-through) 64-bit store
//                each, and swept by every workgroup with agent-scope 64-bit loads until each granule carries the phase's tag - no separate
//                flag, no fence (the guide's `allgather` row). The weights of phase p + 1 are loaded into registers BEFORE the sweep of
//                edge p starts (<= 32 KB per workgroup and phase: the run-ahead the guide gets from its LDS-DMA ring fits in VGPRs at this
//                model size), so a hand-off overlaps the next phase's cold HBM round trip (`prefetch-credit`).
// Spins are bounded: a sweep that does not complete within ~0.2 s sets an abort word every workgroup checks, the kernel drains and the
// probe reports the failure - a protocol bug cannot hang the box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/persist_probe.hip -o tools/persist_probe && tools/persist_probe [layers=24] [reps=20]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (x);                                                                                  \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

constexpr int NWG = 256, NT = 256, NPH = 7;
constexpr int H = 1024, F = 4096, QKV = 3072, CTX = 460, HEADS = 16, SPLITS = 4;
// phase p: N outputs from a K-wide input; weights [N][K] bf16 (the attention phase streams K/V instead)
struct Phase { int N, K; };
static const Phase h_ph[NPH] = {{QKV, H}, {H * SPLITS, QKV}, {H, H * SPLITS}, {H, H}, {H, H}, {F, H}, {H, F}};
constexpr int MAXK = 4096, MAXN = 4096;

typedef unsigned short bf16_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt16(const void* p) {  // 16-byte non-temporal load (streamed once)
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- one phase's arithmetic: this workgroup's rows [r0, r0 + R) of a [N][K] bf16 matrix against x (LDS, fp32); one wave = R / 4 rows,
// lanes split K in 16-byte chunks. The weights are already in registers (wreg: up to 8 uint4 per thread = 32 KB per workgroup).
// Effective K of the matrix: phase 1 (attention) has no matrix - it streams `kvbytes` of K/V per workgroup and reduces them against x;
// phase 2 reads the 4 split partials (K = 4096 inputs) but its matrix is [1024][1024]: the combine is an add of 4 LDS values.
constexpr int WREGS = 8;
struct WRegs { uint4 v[WREGS]; };
template <int P> struct Ph {  // compile-time shapes: every register array below is statically indexed
  static constexpr int N = P == 0 ? QKV : (P == 1 ? H * SPLITS : (P == 5 ? F : H));
  static constexpr int K = P == 0 ? H : (P == 1 ? QKV : (P == 2 ? H * SPLITS : (P == 6 ? F : H)));   // input vector width (granules of the edge)
  static constexpr int KM = P == 2 ? H : K;                                                             // matrix width
  static constexpr int R = N / NWG, RPW = R / 4, CPR = KM / 8 / 64;
};

template <int P>
__device__ __forceinline__ void load_weights(const bf16_t* __restrict__ W, const unsigned char* __restrict__ kv, int wg, int tid, WRegs& w) {
  const int lane = tid & 63, wave = tid >> 6;
  if constexpr (P == 1) {  // K/V slice of this workgroup: 8 KB = 512 uint4, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) w.v[i] = ld_nt16(kv + ((size_t)wg * 512 + (size_t)(tid + i * NT)) * 16);
  } else {
    static_assert(Ph<P>::RPW * Ph<P>::CPR <= WREGS, "weights of a phase fit the register set");
#pragma unroll
    for (int r = 0; r < Ph<P>::RPW; ++r)
#pragma unroll
      for (int c = 0; c < Ph<P>::CPR; ++c)
        w.v[r * Ph<P>::CPR + c] = ld_nt16(W + ((size_t)(wg * Ph<P>::R + wave * Ph<P>::RPW + r) * Ph<P>::KM) + (size_t)(c * 64 + lane) * 8);
  }
}

// the wave's row results in out[0 .. RPW)  (phase 1: 4)
template <int P>
__device__ __forceinline__ void compute_phase(int tid, const float* __restrict__ x, const WRegs& w, float (&out)[4]) {
  const int lane = tid & 63;
  if constexpr (P == 1) {  // stand-in for split-KV attention: the streamed bytes reduced against q, 4 outputs per wave = 16 per workgroup
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 v = w.v[i];
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += bf2f((bf16_t)(u[e] & 0xffff)) * x[(tid * 8 + e * 2) & 2047] + bf2f((bf16_t)(u[e] >> 16)) * x[(tid * 8 + e * 2 + 1) & 2047];
    }
    acc = wave_sum(acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = acc * (0.001f * (r + 1));
  } else {
#pragma unroll
    for (int r = 0; r < Ph<P>::RPW; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < Ph<P>::CPR; ++c) {
        const uint4 v = w.v[r * Ph<P>::CPR + c];
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        const int k0 = (c * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x0 = x[k0 + e * 2], x1 = x[k0 + e * 2 + 1];
          if constexpr (P == 2) {  // combine of the 4 split partials
            x0 += x[k0 + e * 2 + H] + x[k0 + e * 2 + 2 * H] + x[k0 + e * 2 + 3 * H];
            x1 += x[k0 + e * 2 + 1 + H] + x[k0 + e * 2 + 1 + 2 * H] + x[k0 + e * 2 + 1 + 3 * H];
          }
          acc += bf2f((bf16_t)(u[e] & 0xffff)) * x0 + bf2f((bf16_t)(u[e] >> 16)) * x1;
        }
      }
      out[r] = wave_sum(acc) * 0.03f;  // keep magnitudes bounded over 168 phases
    }
  }
}

// ---- variant A: one kernel per phase -----------------------------------------------------------------------------------------------
template <int P>
__global__ void __launch_bounds__(NT) phase_kernel(const bf16_t* __restrict__ W, const unsigned char* __restrict__ kv, const float* __restrict__ xin,
                                                   float* __restrict__ xout) {
  __shared__ float sx[MAXK];
  const int tid = threadIdx.x, wg = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  WRegs w;
  load_weights<P>(W, kv, wg, tid, w);  // weights first: they do not depend on the previous kernel
  for (int i = tid; i < Ph<P>::K / 4; i += NT) reinterpret_cast<float4*>(sx)[i] = reinterpret_cast<const float4*>(xin)[i];
  __syncthreads();
  float out[4];
  compute_phase<P>(tid, sx, w, out);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < Ph<P>::RPW; ++r) xout[wg * Ph<P>::R + wave * Ph<P>::RPW + r] = out[r];
  }
}

// ---- variant B: the whole layer stack in one launch ---------------------------------------------------------------------------------
struct PersistArgs {
  const bf16_t* W;            // [layers][7 phases] matrices, contiguous
  const size_t* w_off;        // element offset of (layer, phase)
  const unsigned char* kv;    // [layers][NWG][512 x 16 B]
  unsigned long long* gran;   // [NPH][MAXN] granules {fp32 data (low), u32 tag (high)}
  int layers;
  unsigned tag0;              // tags of this launch start at tag0 + 1 (a launch uses layers * 7 tags)
  int* abort_flag;
  int prefetch;               // 1: weights of phase p + 1 requested before the sweep of edge p; 0: after it (what the edge costs alone)
};

// one phase of the persistent kernel: sweep edge (P - 1) -> compute -> request phase P + 1's weights -> publish edge P. Returns false on abort.
template <int P>
__device__ __forceinline__ bool persist_phase(const PersistArgs& a, int l, unsigned& tag, WRegs& w, float* sx, int tid, int wg) {
  constexpr int K = Ph<P>::K, PP = (P + NPH - 1) % NPH, NP = (P + 1) % NPH;
  const int wave = tid >> 6, lane = tid & 63;
  const unsigned long long* g = a.gran + (size_t)PP * MAXN;
  int aborted = 0;
  // every thread sweeps its granules (4 loads in flight per thread = 16 per wave-pass... x 4 waves), re-reading only what was not ready
  for (int i0 = tid; i0 < K && !aborted; i0 += NT * 4) {
    unsigned long long v[4] = {0, 0, 0, 0};
    int ready = 0, spins = 0;
    while (ready != 15) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (!((ready >> u) & 1) && i0 + u * NT < K) v[u] = __hip_atomic_load(g + i0 + u * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * NT >= K || (unsigned)(v[u] >> 32) == tag) ready |= 1 << u;
      if (ready != 15 && ++spins > 4096) {  // bounded: ~0.3 s at most, then every workgroup drains
        if ((spins & 1023) == 0 && (spins > (1 << 18) || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(a.abort_flag, 1 + l * NPH + P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          aborted = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NT < K) sx[i0 + u * NT] = __uint_as_float((unsigned)v[u]);
  }
  if (__syncthreads_or(aborted)) return false;
  float out[4];
  compute_phase<P>(tid, sx, w, out);
  ++tag;
  const int nl = l + (P + 1 == NPH);
  const bool more = nl < a.layers;
  if (a.prefetch && more) load_weights<NP>(a.W + a.w_off[nl * NPH + NP], a.kv + (size_t)nl * NWG * 512 * 16, wg, tid, w);
  if (lane == 0) {
    unsigned long long* o = a.gran + (size_t)P * MAXN + wg * Ph<P>::R + wave * Ph<P>::RPW;
#pragma unroll
    for (int r = 0; r < Ph<P>::RPW; ++r)
      __hip_atomic_store(o + r, ((unsigned long long)tag << 32) | __float_as_uint(out[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!a.prefetch && more) load_weights<NP>(a.W + a.w_off[nl * NPH + NP], a.kv + (size_t)nl * NWG * 512 * 16, wg, tid, w);
  __syncthreads();  // sx is rewritten by the next sweep
  return true;
}

__global__ void __launch_bounds__(NT) persist_kernel(PersistArgs a) {
  __shared__ float sx[MAXK];
  const int tid = threadIdx.x, wg = blockIdx.x;
  unsigned tag = a.tag0;
  WRegs w;
  // the input of the very first phase was published by the init kernel with tag0
  load_weights<0>(a.W + a.w_off[0], a.kv, wg, tid, w);
  for (int l = 0; l < a.layers; ++l) {
    if (!persist_phase<0>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<1>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<2>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<3>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<4>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<5>(a, l, tag, w, sx, tid, wg)) return;
    if (!persist_phase<6>(a, l, tag, w, sx, tid, wg)) return;
  }
}

__global__ void init_gran(unsigned long long* g, unsigned tag) {  // the edge into phase 0: MAXN granules of phase 6's slot
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < MAXN) g[(size_t)(NPH - 1) * MAXN + i] = ((unsigned long long)tag << 32) | __float_as_uint(0.01f * (float)(i & 63));
}
__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  const float v = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.05f;
  p[i] = (bf16_t)(__float_as_uint(v) >> 16);
}

int main(int argc, char** argv) {
  int layers = 24, reps = 20;
  for (int i = 1; i < argc; ++i) {
    if (!strncmp(argv[i], "layers=", 7)) layers = atoi(argv[i] + 7);
    if (!strncmp(argv[i], "reps=", 5)) reps = atoi(argv[i] + 5);
  }
  std::vector<size_t> off(layers * NPH);
  size_t total = 0;
  for (int l = 0; l < layers; ++l)
    for (int p = 0; p < NPH; ++p) {
      off[l * NPH + p] = total;
      if (p != 1) total += (size_t)h_ph[p].N * (p == 2 ? H : h_ph[p].K);
    }
  bf16_t* W;
  unsigned char* kv;
  size_t* d_off;
  unsigned long long* gran;
  float *xa, *xb;
  int* abort_flag;
  HIPCHK(hipMalloc(&W, total * 2));
  HIPCHK(hipMalloc(&kv, (size_t)layers * NWG * 512 * 16));
  HIPCHK(hipMalloc(&d_off, off.size() * sizeof(size_t)));
  HIPCHK(hipMalloc(&gran, (size_t)NPH * MAXN * 8));
  HIPCHK(hipMalloc(&xa, MAXN * 4 * 2)); xb = xa + MAXN;
  HIPCHK(hipMalloc(&abort_flag, 4));
  HIPCHK(hipMemcpy(d_off, off.data(), off.size() * sizeof(size_t), hipMemcpyHostToDevice));
  fill_bf16<<<dim3((unsigned)((total + 255) / 256)), dim3(256)>>>(W, total, 1u);
  fill_bf16<<<dim3((unsigned)(((size_t)layers * NWG * 512 * 8 + 255) / 256)), dim3(256)>>>(reinterpret_cast<bf16_t*>(kv), (size_t)layers * NWG * 512 * 8, 2u);
  HIPCHK(hipMemset(gran, 0, (size_t)NPH * MAXN * 8));
  HIPCHK(hipMemset(xa, 0, MAXN * 4 * 2));
  HIPCHK(hipMemset(abort_flag, 0, 4));
  HIPCHK(hipDeviceSynchronize());
  const double mb = (double)total * 2 / 1e6 + (double)layers * NWG * 512 * 16 / 1e6;
  hipStream_t st;
  HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));

  // ---- A: 7 launches per layer in one graph -----------------------------------------------------------------------------------------
  hipGraph_t g;
  hipGraphExec_t ge;
  HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < layers; ++l)
    for (int p = 0; p < NPH; ++p) {
      const int idx = l * NPH + p;
      const bf16_t* wp = W + off[idx];
      const unsigned char* kp = kv + (size_t)l * NWG * 512 * 16;
      const float* xi = (idx & 1) ? xb : xa;
      float* xo = (idx & 1) ? xa : xb;
      switch (p) {
        case 0: phase_kernel<0><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        case 1: phase_kernel<1><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        case 2: phase_kernel<2><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        case 3: phase_kernel<3><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        case 4: phase_kernel<4><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        case 5: phase_kernel<5><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
        default: phase_kernel<6><<<dim3(NWG), dim3(NT), 0, st>>>(wp, kp, xi, xo); break;
      }
    }
  HIPCHK(hipStreamEndCapture(st, &g));
  HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) HIPCHK(hipGraphLaunch(ge, st));
  HIPCHK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) HIPCHK(hipGraphLaunch(ge, st));
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  float msA = 0;
  HIPCHK(hipEventElapsedTime(&msA, e0, e1));
  printf("[persist_probe] %d layers x 7 phases, %.0f MB of weights + K/V per pass (cold: > the 256 MB Infinity Cache at 24 layers)\n", layers, mb);
  printf("[persist_probe] launches   : %8.1f us per pass = %6.2f us per layer = %5.2f us per phase (168-node hipGraph)\n", msA * 1e3 / reps, msA * 1e3 / reps / layers,
         msA * 1e3 / reps / layers / NPH);
  fflush(stdout);

  // ---- B: persistent, with and without the run-ahead weight requests --------------------------------------------------------------------
  unsigned tag = 100;
  for (int prefetch = 1; prefetch >= 0; --prefetch) {
    PersistArgs a = {W, d_off, kv, gran, layers, 0, abort_flag, prefetch};
    float ms = 0;
    bool failed = false;
    for (int i = 0; i < reps + 3 && !failed; ++i) {
      if (i == 3) HIPCHK(hipEventRecord(e0, st));
      init_gran<<<dim3(MAXN / 256), dim3(256), 0, st>>>(gran, tag);
      a.tag0 = tag;
      persist_kernel<<<dim3(NWG), dim3(NT), 0, st>>>(a);
      tag += layers * NPH + 1;
      if (i < 3) {  // check the protocol on the warm-up passes before timing
        HIPCHK(hipStreamSynchronize(st));
        int ab = 0;
        HIPCHK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        if (ab) { printf("[persist_probe] persistent (prefetch=%d): ABORTED at phase index %d (a sweep did not complete)\n", prefetch, ab - 1); failed = true; }
      }
    }
    if (failed) { HIPCHK(hipMemset(abort_flag, 0, 4)); continue; }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    int ab = 0;
    HIPCHK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
    printf("[persist_probe] persistent (weights of phase p+1 requested %s the sweep of edge p): %8.1f us per pass = %6.2f us per layer = %5.2f us per edge%s  -> %.2fx the launches\n",
           prefetch ? "BEFORE" : "AFTER ", ms * 1e3 / reps, ms * 1e3 / reps / layers, ms * 1e3 / reps / layers / NPH, ab ? "  (ABORTED during timing!)" : "", ms / msA);
    fflush(stdout);
  }
  return 0;
}
