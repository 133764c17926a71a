// Standalone probe (not part of the product): can a persistent decode-step kernel synchronise its phases through
// UNCACHED device memory (MTYPE_UC: no L2 write-back / invalidate fences needed) cheaper than a kernel boundary
// (2.13 us per dependent graph node, tools/barrier_probe.hip)?  Bounded spins: can not hang.
//   variant 0: every WG publishes flag[wg] = epoch; wave 0 of every WG polls all flags
//   variant 1: WG 0 polls all flags, then publishes go = epoch; the others poll go only
//   mode bit 0: + 4 KB activation vector produced before / consumed after the sync (UC memory)
//   mode bit 1: + 32 KB of weights per WG, loads issued right after the flag store (in flight across the sync)
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/sync_probe tools/sync_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned ld_uc(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_uc(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int VARIANT>
__global__ void __launch_bounds__(256) uc_phases(unsigned* flags, unsigned* go, unsigned* act, const uint4* w, size_t w_stride_phase,
                                                 float* sink, int nphase, int mode, int* fail) {
  const int nwg = gridDim.x, wg = blockIdx.x, t = threadIdx.x;
  float acc = 0.f;
  __shared__ float red[256];
  __shared__ int s_ok;
  for (int p = 0; p < nphase; ++p) {
    const unsigned epoch = (unsigned)p + 1;
    // produce this WG's slice of the activation vector, make it globally visible, publish the flag
    if ((mode & 1) && t < 4) st_uc(act + (p & 1) * 1024 + (wg * 4 + t) % 1024, __float_as_uint(acc + (float)p));
    __builtin_amdgcn_s_waitcnt(0);  // stores acknowledged by memory (UC: written through)
    __syncthreads();
    if (t == 0) st_uc(flags + wg, epoch);
    // weights of the NEXT phase: issued now, in flight while waiting
    uint4 wt[8];
    const uint4* wp = w + (size_t)(p & 7) * w_stride_phase + (size_t)wg * 2048;
    if (mode & 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) wt[u] = ldnt(wp + u * 256 + t);
    }
    if (t < 64) {
      bool ok = true;
      int spins = 0;
      if (VARIANT == 0 || wg == 0) {
        while (true) {
          bool all = true;
          for (int i = t; i < nwg; i += 64) all = all && (ld_uc(flags + i) >= epoch);
          if (__all(all)) break;
          if (++spins > (1 << 20)) { ok = false; break; }
        }
        if (VARIANT == 1 && t == 0) st_uc(go, epoch);
      } else {
        while (ld_uc(go) < epoch) { if (++spins > (1 << 20)) { ok = false; break; } __builtin_amdgcn_s_sleep(1); }
      }
      if (t == 0) { s_ok = ok; if (!ok) *fail = 1; }
    }
    __syncthreads();
    if (!s_ok) return;
    if (mode) {
      float s = 0.f;
      if (mode & 1) {
        const unsigned* a = act + (p & 1) * 1024 + t * 4;
        s = __uint_as_float(ld_uc(a)) + __uint_as_float(ld_uc(a + 1)) + __uint_as_float(ld_uc(a + 2)) + __uint_as_float(ld_uc(a + 3));
      }
      if (mode & 2) {
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

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned *flags, *go, *act; int* fail; float* sink; uint4* w;
  const size_t phase_bytes = (size_t)512 * 32768;
  for (int memkind = 0; memkind < 3; ++memkind) {
    const char* mk = memkind == 0 ? "hipMalloc (coarse, cached)" : memkind == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained";
    if (memkind == 0) { CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&go, 4096)); CK(hipMalloc(&act, 8192)); }
    else {
      const unsigned fl = memkind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
      CK(hipExtMallocWithFlags((void**)&flags, 4096, fl)); CK(hipExtMallocWithFlags((void**)&go, 4096, fl)); CK(hipExtMallocWithFlags((void**)&act, 8192, fl));
    }
    CK(hipMalloc(&fail, 4)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&w, phase_bytes * 8)); CK(hipMemset(w, 0, phase_bytes * 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int nphase = 2000;
    for (int nwg : {64, 128, 256}) for (int variant = 0; variant < 2; ++variant) for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f; int failed = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(flags, 0, 4096, st)); CK(hipMemsetAsync(go, 0, 4096, st)); CK(hipMemsetAsync(act, 0, 8192, st)); CK(hipMemsetAsync(fail, 0, 4, st));
        size_t stride16 = phase_bytes / 16;
        void* args[] = {&flags, &go, &act, &w, &stride16, &sink, (void*)&nphase, &mode, &fail};
        CK(hipEventRecord(a, st));
        CK(hipLaunchCooperativeKernel(variant == 0 ? (void*)uc_phases<0> : (void*)uc_phases<1>, dim3(nwg), dim3(256), args, 0, st));
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        CK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
        if (failed) break;
      }
      printf("[%s] WGs %3d %s mode %d (%s%s): %.2f us per phase%s\n", mk, nwg, variant == 0 ? "all-poll-all" : "master/go   ", mode,
             (mode & 1) ? "act " : "", (mode & 2) ? "weights" : "", best * 1e3f / nphase, failed ? "  [SPIN LIMIT HIT / stale data]" : "");
    }
    hipFree(flags); hipFree(go); hipFree(act); hipFree(fail); hipFree(sink); hipFree(w);
  }
  return 0;
}
