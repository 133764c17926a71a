// GPU probe (standalone, no torch, no libptts): is the fused DAC residual unit's epilogue access pattern what holds the C = 96 / 192
// units at ~2.9 TB/s (profiles/r03_dac_kernels_bs32.txt)? resunit_lds_kernel's emit() gives every lane 4 channels of ONE frame: a wave
// instruction touches 16 frames x 64 contiguous bytes (rows C * 4 bytes apart) for the fp32 residual read, the fp32 stream write and
// the bf16 activation write (16 x 32 bytes). The alternative transposes the output tile through LDS and moves whole rows.
//   direct : the product's pattern (accumulators synthetic: a function of (frame, channel), same for both variants)
//   lds    : accumulators + bias -> LDS tile [128 frames][C fp32], barrier, then row-contiguous float4 per lane: + residual,
//            stream write, Snake, bf16 write
// Both kernels produce the same bytes (compared on the host). Traffic = 10 bytes per element (4 read + 4 + 2 written).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/epilogue_probe.hip -o tools/epilogue_probe && tools/epilogue_probe [batch]
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

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ float snake(float x, float al) {
  const float s = __sinf(al * x);
  return x + s * s / (al + 1e-9f);
}
__device__ __forceinline__ float synth_acc(int frame, int ch) { return 0.001f * (float)((frame * 31 + ch * 7) & 1023) - 0.5f; }


// NW waves, each owning 3 strips of 16 channels x 8 frame tiles, as resunit_lds_kernel<NW>
template <int NW, int FT>
__global__ void __launch_bounds__(NW * 64) epi_direct(const float* __restrict__ skip, const float* __restrict__ bias, const float* __restrict__ alpha,
                                                      float* __restrict__ out_raw, unsigned short* __restrict__ out_act, int T) {
  constexpr int C = NW * 48, TF = FT * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  const int ntile = (T + TF - 1) / TF, tile = blockIdx.x % ntile, b = blockIdx.x / ntile, t0 = tile * TF;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int jj = t0 + f * 16 + j;
      if (jj >= T) continue;
      const int co = (wave * 3 + s) * 16 + q * 4;
      const float4 bs = *reinterpret_cast<const float4*>(bias + co);
      const size_t o = ((size_t)b * T + jj) * C + co;
      const float4 sk = *reinterpret_cast<const float4*>(skip + o);
      const float4 v = make_float4(synth_acc(jj, co) + bs.x + sk.x, synth_acc(jj, co + 1) + bs.y + sk.y, synth_acc(jj, co + 2) + bs.z + sk.z,
                                   synth_acc(jj, co + 3) + bs.w + sk.w);
      *reinterpret_cast<float4*>(out_raw + o) = v;
      const float4 al = *reinterpret_cast<const float4*>(alpha + co);
      *reinterpret_cast<uint2*>(out_act + o) = make_uint2(pack_bf16x2(snake(v.x, al.x), snake(v.y, al.y)), pack_bf16x2(snake(v.z, al.z), snake(v.w, al.w)));
    }
}

template <int NW, int FT>
__global__ void __launch_bounds__(NW * 64) epi_lds(const float* __restrict__ skip, const float* __restrict__ bias, const float* __restrict__ alpha,
                                                   float* __restrict__ out_raw, unsigned short* __restrict__ out_act, int T) {
  constexpr int C = NW * 48, TF = FT * 16, RS = C * 4 + 16;  // row stride: 16 bytes of padding (lanes of one store hit 16 different rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  const int ntile = (T + TF - 1) / TF, tile = blockIdx.x % ntile, b = blockIdx.x / ntile, t0 = tile * TF;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int jj = t0 + f * 16 + j;
      const int co = (wave * 3 + s) * 16 + q * 4;
      const float4 bs = *reinterpret_cast<const float4*>(bias + co);
      *reinterpret_cast<float4*>(tile_lds + (f * 16 + j) * RS + co * 4) =
          make_float4(synth_acc(jj, co) + bs.x, synth_acc(jj, co + 1) + bs.y, synth_acc(jj, co + 2) + bs.z, synth_acc(jj, co + 3) + bs.w);
    }
  __syncthreads();
  constexpr int VPR = C / 4, NV = TF * VPR;  // float4 per row, per tile
  const int rows = min(TF, T - t0);
  const size_t base = ((size_t)b * T + t0) * C;
#pragma unroll 4
  for (int i = threadIdx.x; i < NV; i += NW * 64) {
    const int r = i / VPR, cv = i - r * VPR;
    if (r >= rows) break;
    const float4 a = *reinterpret_cast<const float4*>(tile_lds + r * RS + cv * 16);
    const size_t o = base + (size_t)i * 4;  // rows of a tile are contiguous in memory: i * 4 == r * C + cv * 4
    const float4 sk = *reinterpret_cast<const float4*>(skip + o);
    const float4 v = make_float4(a.x + sk.x, a.y + sk.y, a.z + sk.z, a.w + sk.w);
    *reinterpret_cast<float4*>(out_raw + o) = v;
    const float4 al = *reinterpret_cast<const float4*>(alpha + cv * 4);
    *reinterpret_cast<uint2*>(out_act + o) = make_uint2(pack_bf16x2(snake(v.x, al.x), snake(v.y, al.y)), pack_bf16x2(snake(v.z, al.z), snake(v.w, al.w)));
  }
}

__global__ void fill(float* p, size_t n, unsigned seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  p[i] = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

// FT: frame tiles (of 16) per workgroup in the through-LDS variant (the direct variant always runs the product's 128-frame tiles)
template <int NW, int FT>
static void run(int B, int T) {
  constexpr int C = NW * 48, TF = FT * 16;
  const size_t n = (size_t)B * T * C;
  float *skip, *bias, *alpha, *raw0, *raw1;
  unsigned short *act0, *act1;
  HIPCHK(hipMalloc(&skip, n * 4)); HIPCHK(hipMalloc(&raw0, n * 4)); HIPCHK(hipMalloc(&raw1, n * 4));
  HIPCHK(hipMalloc(&act0, n * 2)); HIPCHK(hipMalloc(&act1, n * 2));
  HIPCHK(hipMalloc(&bias, C * 4)); HIPCHK(hipMalloc(&alpha, C * 4));
  fill<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(skip, n, 1u);
  fill<<<dim3(1), dim3(256)>>>(bias, C, 2u);
  std::vector<float> al(C, 1.0f);
  HIPCHK(hipMemcpy(alpha, al.data(), C * 4, hipMemcpyHostToDevice));
  const dim3 grid_d((unsigned)((T + 127) / 128 * B)), grid_l((unsigned)((T + TF - 1) / TF * B)), block(NW * 64);
  const size_t lds = (size_t)TF * (C * 4 + 16);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&epi_lds<NW, FT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int variant = 0; variant < 2; ++variant) {
    for (int rep = 0; rep < 4; ++rep) {  // 1 warm-up + 3 timed
      if (rep == 1) HIPCHK(hipEventRecord(e0));
      if (variant == 0) epi_direct<NW, 8><<<grid_d, block>>>(skip, bias, alpha, raw0, act0, T);
      else epi_lds<NW, FT><<<grid_l, block, lds>>>(skip, bias, alpha, raw1, act1, T);
    }
    HIPCHK(hipEventRecord(e1));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventElapsedTime(&ms[variant], e0, e1));
    ms[variant] /= 3;
  }
  // same bytes?
  const size_t chk = n < (size_t)(1 << 24) ? n : (size_t)(1 << 24);
  std::vector<float> h0(chk), h1(chk);
  std::vector<unsigned short> a0(chk), a1(chk);
  HIPCHK(hipMemcpy(h0.data(), raw0 + (n - chk), chk * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h1.data(), raw1 + (n - chk), chk * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(a0.data(), act0 + (n - chk), chk * 2, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(a1.data(), act1 + (n - chk), chk * 2, hipMemcpyDeviceToHost));
  const bool same = !memcmp(h0.data(), h1.data(), chk * 4) && !memcmp(a0.data(), a1.data(), chk * 2);
  const double gb = (double)n * 10 / 1e9;
  printf("[epilogue_probe] C=%3d B=%d T=%d (%.2f GB): direct %.3f ms = %.2f TB/s | through LDS (%d-frame tiles, %zu KB) %.3f ms = %.2f TB/s | outputs %s\n", C, B, T, gb, ms[0],
         gb / ms[0], TF, lds >> 10, ms[1], gb / ms[1], same ? "identical" : "DIFFER");
  fflush(stdout);
  HIPCHK(hipFree(skip)); HIPCHK(hipFree(raw0)); HIPCHK(hipFree(raw1)); HIPCHK(hipFree(act0)); HIPCHK(hipFree(act1)); HIPCHK(hipFree(bias)); HIPCHK(hipFree(alpha));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  run<2, 8>(B, 860 * 512);  // C = 96: the last decoder block
  run<2, 4>(B, 860 * 512);
  run<4, 8>(B, 860 * 256);  // C = 192 (128-frame fp32 tile: 98 KB, one workgroup per CU)
  run<4, 4>(B, 860 * 256);
  run<8, 4>(B, 860 * 64);   // C = 384 (a 128-frame fp32 tile does not fit the LDS)
  return 0;
}
