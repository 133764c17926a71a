// GPU probe (not part of the product): weights of the single-utterance decode step waiting in the Infinity Cache (MALL, 256 MB, memory-side,
// shared by the 8 XCDs) when their node starts. tools/prefetch_probe.hip showed: an 8 MB GEMV node costs 3.70 us with HBM-cold weights and
// 2.72 us when they come from the MALL (2 MB: 2.35 -> 2.25), and that prefetching INSIDE the previous node buys nothing (the bytes only move
// from one node of the chain to the other). HBM is idle ~80 % of the step - in particular during every kernel boundary - so here the bytes are
// moved by a CONCURRENT kernel on a second stream:
//   main stream : hipGraph of dependent GEMV nodes over 680 MB of weights in rotation (cold without help); workgroup 0 of each node bumps a
//                 progress counter (one relaxed agent-scope atomic, fire and forget);
//   side stream : ONE persistent launch of P single-wave workgroups; wave p touches slice p of node u's matrix as soon as u < progress + LEAD
//                 (relaxed sc1 poll + s_sleep by lane 0), plain 16-byte loads, 32 in flight per lane (one round trip per node at P = 256), results discarded. It ends after the last node,
//                 or when the counter has not moved for ~20 ms.
// Reported: microseconds per node without / with the prefetcher for several (P, LEAD), for 8 MB nodes, 2 MB nodes and a layer-like mix.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mall_prefetch_probe tools/mall_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct NArgs {
  const u32x4* W;      // [N][1024] bf16: 128 x 16 B per row
  const float* x;      // [1024] fp32, written by the previous node
  float* out;          // [N]
  unsigned* progress;  // bumped once per node; progress[64 + j] = XCC_ID workgroup j < 8 ran on (the prefetcher's XCD map)
  int N, write_map;  // write_map: the first node of a pass only
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <int R>
__global__ void __launch_bounds__(256) node_kernel(NArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (a.progress && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(a.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.write_map && blockIdx.x < 8 && threadIdx.x == 0) {
    unsigned xc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
    __hip_atomic_store(a.progress + 64 + blockIdx.x, xc & 15u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= a.N) return;
  u32x4 w[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) w[r][c] = __builtin_nontemporal_load(a.W + ((size_t)(row0 + r) * 128 + c * 64 + lane));
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

struct Unit { const u32x4* p; unsigned n16, gran16; };  // one node's matrix: pointer, 16-byte vectors, vectors one workgroup of the node reads (workgroup j: granule j, XCD j % 8)

// persistent prefetcher: P single-wave workgroups. units[] lists ONE pass of the chain (npass passes are walked); unit index u (global) is
// touched once progress + lead > u. The loads are LDS-DMA (global_load_lds_dwordx4: no destination registers, up to the hardware's 63
// wave-instructions = 63 KB in flight per wave, nothing ever waits for them; all land on the same 1 KB of LDS, never read). The progress word
// is polled (one vector load: drains the queue, harmless at that point) only when the wave has reached the limit it last saw.
// mode 0: prefetch (any XCD: Infinity Cache only); mode 1: poll only, no data loads (what does a concurrent launch cost the chain by itself?);
// mode 2: L2-targeted - a wave touches only the granules the node's workgroups of ITS OWN XCD will read (lead of 1-3 nodes: 4 MB of L2 per XCD)
__global__ void __launch_bounds__(64) prefetch_kernel(const Unit* __restrict__ units, int nunits, int npass, const unsigned* progress, int lead, int mode,
                                                      unsigned* stats) {
  __shared__ __attribute__((aligned(16))) char s_dst[1024];
  const int lane = threadIdx.x, P = gridDim.x, p = blockIdx.x;
  const long long total = (long long)nunits * npass;
  unsigned seen = 0, idle = 0, skipped = 0, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15;
  const unsigned hwxcc = xcc;
  int slot = -1;  // k such that the chain's workgroups j = k (mod 8) run on this wave's XCD (read from the map the chain's nodes publish)
  const unsigned wx = p / 8, nwx = P / 8;  // wave index within its XCD, assuming the launch spreads its workgroups round-robin too
  unsigned long long waits = 0;
  for (long long u = 0; u < total; ++u) {
    while ((long long)seen + lead <= u) {  // wave-uniform: every lane reads the same word
      seen = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((long long)seen + lead > u) break;
      __builtin_amdgcn_s_sleep(4);
      ++waits;
      if (++idle > 300000u) { if (lane == 0 && p == 0) stats[1] = 1; return; }  // the chain has stopped: give up
    }
    idle = 0;
    if ((long long)seen > u + 1) { ++skipped; continue; }  // the chain has passed this node already: nothing to win
    if (mode == 1) continue;
    const int ui = __builtin_amdgcn_readfirstlane((int)(u % nunits));
    const Unit un = units[ui];
    if (mode == 2 && slot < 0) {
      for (int k = 0; k < 8; ++k)
        if (__hip_atomic_load(progress + 64 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hwxcc) slot = k;
      if (slot < 0) continue;  // the chain has not published its placement yet
      xcc = (unsigned)slot;
      if (lane == 0) { atomicAdd(&stats[3], 1u); atomicAdd(&stats[4 + (p & 7)], 1u << (4 * slot)); }
    }
    if (mode == 2) {  // L2-targeted: this wave touches only what workgroups of ITS XCD will read (granule g is read on XCD g % 8)
      const unsigned ng = un.n16 / un.gran16, ngx = (ng + 7 - xcc) / 8, nvx = ngx * un.gran16;  // granules / vectors of this XCD
      const unsigned per = (nvx + nwx - 1) / nwx, b0 = wx * per, b1 = min(nvx, b0 + per);
      for (unsigned i = b0; i < b1; i += 64) {
        const unsigned li = min(i + lane, b1 - 1), gi = li / un.gran16, j = (xcc + 8 * gi) * un.gran16 + (li - gi * un.gran16);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(un.p + j), (__attribute__((address_space(3))) void*)(s_dst), 16, 0, 0);
      }
      continue;
    }
    const unsigned per = (un.n16 + P - 1) / P, b0 = p * per, b1 = min(un.n16, b0 + per);
    for (unsigned i = b0; i < b1; i += 64) {
      const unsigned j = min(i + lane, b1 - 1);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(un.p + j), (__attribute__((address_space(3))) void*)(s_dst), 16, 0, 0);
    }
  }
  if (lane == 0 && p == 0) { stats[0] = (unsigned)(waits > 0xffffffffull ? 0xffffffffull : waits); stats[2] = skipped; }
}

int main() {
  hipStream_t st, st2; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  const int LAYERS = 24, REPS = 60, WARM = 5;
  const size_t slot = (size_t)4096 * 1024 * 2;  // 8 MB
  const int NSLOT = 85;
  char* W; float *xa, *xb, *sink; unsigned *progress, *stats; Unit* dunits;
  CK(hipMalloc(&W, slot * NSLOT)); CK(hipMemset(W, 0, slot * NSLOT));
  CK(hipMalloc(&xa, 4096 * 4)); CK(hipMalloc(&xb, 4096 * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(xa, 0, 4096 * 4)); CK(hipMemset(xb, 0, 4096 * 4));
  CK(hipMalloc(&progress, 1024)); CK(hipMalloc(&stats, 64)); CK(hipMalloc(&dunits, sizeof(Unit) * 256));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // chains: rows per node of one "layer" (x LAYERS)
  struct Chain { const char* name; int nn; int rows[5]; } chains[] = {
      {"layer-like mix: 3072 / 1024 / 2048 / 4096 / 4096 rows (6 / 2 / 4 / 8 / 8 MB)", 5, {3072, 1024, 2048, 4096, 4096}},
      {"8 MB nodes", 5, {4096, 4096, 4096, 4096, 4096}},
      {"2 MB nodes", 5, {1024, 1024, 1024, 1024, 1024}},
  };
  struct Cfg { int P, lead, mode; } cfgs[] = {{0, 0, 0}, {64, 10, 1}, {128, 10, 0}, {128, 2, 2}, {256, 1, 2}, {256, 2, 2}, {256, 3, 2}, {512, 2, 2}, {256, 5, 2}, {0, 0, 0}};
  for (auto& ch : chains) {
    // build the graph: LAYERS x nn nodes; node matrices walk the 680 MB arena in order (cold at every pass)
    std::vector<Unit> units;
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    size_t off = 0;
    int i = 0;
    for (int l = 0; l < LAYERS; ++l)
      for (int k = 0; k < ch.nn; ++k, ++i) {
        const size_t bytes = (size_t)ch.rows[k] * 2048;
        if (off + bytes > slot * NSLOT) off = 0;
        NArgs a = {};
        a.W = reinterpret_cast<const u32x4*>(W + off); off += bytes;
        a.x = (i & 1) ? xb : xa; a.out = (i & 1) ? xa : xb; a.N = ch.rows[k]; a.progress = progress; a.write_map = i == 0;
        units.push_back({a.W, (unsigned)(bytes / 16), (unsigned)((ch.rows[k] >= 2048 ? 4 : 1) * 4 * 128)});
        const int R = ch.rows[k] >= 2048 ? 4 : 1;
        const dim3 grid(ch.rows[k] / (4 * R)), blk(256);
        if (R == 1) hipLaunchKernelGGL(node_kernel<1>, grid, blk, 0, st, a);
        else hipLaunchKernelGGL(node_kernel<4>, grid, blk, 0, st, a);
      }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    const int nunits = (int)units.size();
    CK(hipMemcpy(dunits, units.data(), sizeof(Unit) * nunits, hipMemcpyHostToDevice));
    double mb = 0; for (auto& u : units) mb += u.n16 * 16.0 / 1e6;
    printf("[mall_prefetch_probe] chain: %s; %d nodes, %.0f MB per pass\n", ch.name, nunits, mb);
    for (auto& cf : cfgs) {
      CK(hipMemset(progress, 0, 4)); CK(hipMemset(progress + 64, 0xff, 32)); CK(hipMemset(stats, 0, 64));
      CK(hipDeviceSynchronize());
      if (cf.P) hipLaunchKernelGGL(prefetch_kernel, dim3(cf.P), dim3(64), 0, st2, dunits, nunits, WARM + REPS, progress, cf.lead, cf.mode, stats);
      for (int r = 0; r < WARM; ++r) hipGraphLaunch(ex, st);
      hipEventRecord(e0, st);
      for (int r = 0; r < REPS; ++r) hipGraphLaunch(ex, st);
      hipEventRecord(e1, st); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      CK(hipDeviceSynchronize());
      unsigned hs[12]; CK(hipMemcpy(hs, stats, 48, hipMemcpyDeviceToHost));
      printf("[mall_prefetch_probe]   P=%3d lead=%2d: %.3f us per node, %.1f us per pass (%.2f TB/s)%s  [prefetcher%s: %u polls, %u nodes skipped by wave 0, %u waves mapped, slots by launch residue %x %x %x %x%s]\n", cf.P, cf.lead,
             ms * 1e3f / REPS / nunits, ms * 1e3f / REPS, mb / (ms * 1e3 / REPS) , cf.P ? "" : "  <- no prefetcher", cf.mode == 1 ? " (poll only)" : (cf.mode == 2 ? " (L2-targeted)" : ""), hs[0], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7], hs[1] ? ", GAVE UP" : "");
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  return 0;
}
