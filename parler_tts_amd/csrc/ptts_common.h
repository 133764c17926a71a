// Shared host/device helpers for libptts_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <atomic>

#include "../../include/ptts.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint16_t bf16_t;  // raw bf16 bits

// ---- error plumbing: no exception crosses the C ABI ------------------------------------------------
extern thread_local std::string g_ptts_err;
int ptts_fail(int code, const char* fmt, ...);
#define PTTS_HIP(call)                                                                       \
  do {                                                                                       \
    hipError_t _e = (call);                                                                  \
    if (_e != hipSuccess) return ptts_fail(PTTS_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define PTTS_CHECK(cond, code, ...) \
  do {                              \
    if (!(cond)) return ptts_fail(code, __VA_ARGS__); \
  } while (0)
#define PTTS_TRY(expr)       \
  do {                       \
    int _r = (expr);         \
    if (_r != PTTS_OK) return _r; \
  } while (0)

// Every C-ABI entry point runs on the engine's device and restores the caller's current device on return (the engine
// may live on cuda:N while torch's current device is another one).
struct PttsDeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit PttsDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = prev == dev || hipSetDevice(dev) == hipSuccess;
    if (prev == dev) prev = -1;  // nothing to restore
  }
  ~PttsDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define PTTS_DEVICE(dev)            \
  PttsDeviceGuard _ptts_dg(dev);    \
  if (!_ptts_dg.ok) return ptts_fail(PTTS_E_HIP, "hipSetDevice(%d) failed (%s:%d)", (int)(dev), __FILE__, __LINE__)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's function object: the opt-in is tracked per
// (kernel instantiation, device), so a second engine on another GPU of the same process gets it too. Two threads racing
// on the first launch both set the attribute: idempotent.
struct PttsPerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  static int device() { int d = 0; (void)hipGetDevice(&d); return d < 0 ? 0 : (d > 63 ? 63 : d); }
  bool need(int dev) const { return !((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
  void done(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};

// ---- environment switches ------------------------------------------------------------------------------------------------------
// The product library reads FIFTEEN PTTS_* variables with getenv: each selects the un-fused / alternative side of a node that a parity test
// compares with the default path (table in DESIGN.md section 6; tests/test_host_logic.py counts them). Every other switch that rounds 1-6 used
// for an A/B measurement is a development knob: ptts_dev_env() answers nullptr in the product build, so the library takes the measured default
// and the compiler folds the other side away; `tools/build_variant.sh <name> -DPTTS_DEV_KNOBS` builds a probe library that reads them.
inline const char* ptts_dev_env(const char* name) {
#ifdef PTTS_DEV_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// ---- kernel-argument preload (gfx950) ---------------------------------------------------------------------------------------------
// A wave's first instructions are s_load of its kernel arguments: one scalar round trip before the first global load can be addressed.
// gfx950's command processor can instead write the first 14 argument dwords of a dispatch into user SGPRs before the first wave starts
// (LLVM: -mllvm -amdgpu-kernarg-preload-count=14, set in __graft_entry__.build), but only for arguments the kernel takes as SCALARS - a
// by-value struct stays behind s_load. Measured on a chain of dependent GEMV nodes: -0.075 us per node (profiles/r05_experiments.txt call 9).
// The nodes of the decode step therefore take the fields in the first 56 bytes of their argument struct A as scalar parameters of their own
// types (pointers stay pointers: re-assembled from integers they would lose their address space and every load would become a flat_load)
// and the rest of A as KTail<A>; A's fields are ordered so that everything a wave needs to address its FIRST loads sits in those 56 bytes.
// Per struct (ptts_gemv.h): <A>_KPARAMS = the parameter list, <A>_KJOIN(a) re-assembles `A a` in registers (SROA: preloaded SGPRs + the
// tail's s_loads, no memory), ptts_klaunch(kernel, ..., a) splits it on the host.
typedef unsigned long long ptts_u64;
template <typename A> struct KTail {
  static_assert(sizeof(A) > 56 && sizeof(A) % 8 == 0, "argument struct: more than the 56 preloaded bytes, a multiple of 8");
  ptts_u64 w[(sizeof(A) - 56) / 8];
};
// A tail field first used by the EPILOGUE (the output pointer) would be fetched by an s_load right in front of the store - a scalar round trip at
// the end of the node's critical path. PTTS_KTOUCH(v), placed behind the wave's load burst, makes the compiler fetch it there, in the burst's shadow.
#define PTTS_KTOUCH(v) asm volatile("" ::"s"(v))
#define PTTS_KTAIL_JOIN(A, a) __builtin_memcpy(reinterpret_cast<char*>(&a) + 56, &kt_, sizeof(kt_))
template <typename A> inline KTail<A> ptts_ktail(const A& a) {
  KTail<A> t;
  memcpy(&t, reinterpret_cast<const char*>(&a) + 56, sizeof(t));
  return t;
}

// ---- bf16 <-> f32 (round-to-nearest-even, identical to torch's .to(bfloat16)) -------------------------
__host__ __device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)0x7fc0;  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two fp32 -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (gfx950; round-to-nearest-even like f32_to_bf16 above)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// element traits of the engine dtype
template <typename WT> struct Elem;
template <> struct Elem<float> {
  static constexpr int KT = 16;   // k per packed weight fragment (4 x mfma_f32_16x16x4f32)
  static constexpr int EPL = 4;   // elements per 16-byte lane load
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ float rnd(float v) { return v; }  // value as stored in this dtype
};
template <> struct Elem<bf16_t> {
  static constexpr int KT = 32;   // 1 x mfma_f32_16x16x32_bf16
  static constexpr int EPL = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ float rnd(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// unpack a 16-byte lane load into EPL floats
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[4], float) {
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ void unpack16(const uint4& v, float (&o)[8], bf16_t) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack16(const float (&o)[4], float) {
  return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
}
__device__ __forceinline__ uint4 pack16(const float (&o)[8], bf16_t) {
  return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}

// 16-byte non-temporal (streamed-once) global load: weights are read exactly once per decode step
__device__ __forceinline__ uint4 ld_nt16(const uint4* p) {
#ifdef PTTS_WEIGHT_CACHED  // A/B experiment only (tools/l2_probe.py): let weight lines allocate in L2
  return *p;
#endif
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// 8 OCP e4m3 bytes -> 8 packed bf16 (exact: 3 mantissa bits into 7), 4 x v_cvt_scalef32_pk_bf16_fp8 with scale 1.0
__device__ __forceinline__ uint4 e4m3x8_to_bf16x8(uint32_t lo, uint32_t hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  return make_uint4(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c), __builtin_bit_cast(uint32_t, d));
}

// ---- cross-lane reductions without LDS ---------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute (an LDS round trip, ~100 cycles each; 12 of them were most of the LayerNorm
// chain in the first profile). These use DPP row operations inside a 16-lane row and v_permlane16/32_swap (gfx950)
// across rows: 6 dependent VALU ops for a full wave64 reduction, result in EVERY lane.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
struct OpSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };

template <typename Op> __device__ __forceinline__ float swap16_reduce(float v) {  // combine adjacent 16-lane rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <typename Op> __device__ __forceinline__ float swap32_reduce(float v) {  // combine the two 32-lane halves
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return Op::f(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// reduce within aligned groups of G lanes (G = 8 or 16), result in every lane of the group
template <typename Op, int G> __device__ __forceinline__ float group_reduce(float v) {
  v = Op::f(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = Op::f(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = Op::f(v, dpp_mov<0x141>(v));  // row_half_mirror: other quad of the 8-lane half
  if (G == 16) v = Op::f(v, dpp_mov<0x140>(v));  // row_mirror: other half of the row
  return v;
}
// lane-wise reduction across the 64/G aligned groups of G lanes (lane c of every group combines with lane c of the
// others), result in every lane
template <typename Op, int G> __device__ __forceinline__ float across_groups_reduce(float v) {
  if (G == 8) v = Op::f(v, dpp_mov<0x128>(v));  // row_ror:8 == xor 8 inside a 16-lane row
  v = swap16_reduce<Op>(v);
  return swap32_reduce<Op>(v);
}
template <typename Op> __device__ __forceinline__ float wave_reduce(float v) {
  return across_groups_reduce<Op, 16>(group_reduce<Op, 16>(v));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<OpMax>(v); }

// ---- 64-key block of the f32-MFMA attention kernels (prefill_attn_mfma_kernel, t5_attn_mfma_kernel) ----------------------------------
// K / V tiles in LDS as fp32 [64 keys][16 slots of 16 B], slot XOR-swizzled by key & 15; lane (j = l & 15, g = l >> 4).
// Round 6, call 46: the compiler had scheduled each fragment read directly in front of its MFMAs - ds_read -> s_waitcnt lgkmcnt(0) -> MFMA, 16 + 64 LDS
// round trips in a row per key block (~5 us of a 14 us launch for 1.7 us of MFMA issue). Now every fragment of a phase is requested before the first
// MFMA of that phase (a scheduling barrier keeps the order; the waits become counted), and V is read as ONE b128 per (key, lane) instead of four b32:
// output tile dt of lane j is column d = 4 j + dt (was 16 dt + j) - which lane holds which column changes, the sum of each output does not.
// S^T = K Q^T: lane ends up with the scores of query j against keys 16 kt + 4 g + r; k order d = 16 c + e + 4 g over (c, e) then g.
__device__ __forceinline__ void attn_block_scores(const float4* sK4, int j, int g, const float4 (&qr)[4], f32x4 (&st)[4]) {
  float4 kr[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) kr[c][kt] = sK4[(16 * kt + j) * 16 + ((4 * c + g) ^ j)];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // four independent accumulator chains interleaved; each chain's own order is (c, e) as before
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c][kt].x, qr[c].x, st[kt], 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c][kt].y, qr[c].y, st[kt], 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c][kt].z, qr[c].z, st[kt], 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[c][kt].w, qr[c].w, st[kt], 0, 0, 0);
  }
}
// V rows of this lane's 16 keys, columns 4 j .. 4 j + 3: requested behind the score MFMAs, in flight under the softmax
__device__ __forceinline__ void attn_block_v_request(const float4* sV4, int j, int g, float4 (&vb)[4][4]) {
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * kt + 4 * g + r;
      vb[kt][r] = sV4[row * 16 + (j ^ (row & 15))];
    }
  __builtin_amdgcn_sched_barrier(0);
}
// O += P V with the lane's own probabilities as the A operand (the MFMA sums over g): o[dt][r] = query 4 g + r, column 4 j + dt;
// k order keys 16 kt + r + 4 g over (kt, r) then g
__device__ __forceinline__ void attn_block_pv(const f32x4 (&p)[4], const float4 (&vb)[4][4], f32x4 (&o)[4]) {
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[kt][r], vb[kt][r].x, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[kt][r], vb[kt][r].y, o[1], 0, 0, 0);
      o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[kt][r], vb[kt][r].z, o[2], 0, 0, 0);
      o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[kt][r], vb[kt][r].w, o[3], 0, 0, 0);
    }
}
