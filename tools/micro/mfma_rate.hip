// Micro-benchmark (diagnostic): issue rate and dependent latency of the bf16 MFMA shapes the mask-head / attention kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void k(const int reps, float *sink, unsigned long long *cyc)
{
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{(float)threadIdx.x, 1.f, 2.f, 3.f};
  s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i); }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (KIND == 0) {        // 8 independent 16x16x16
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    } else if constexpr (KIND == 1) { // 8 independent 16x16x32
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    } else if constexpr (KIND == 2) { // dependent chain of 16x16x16 (one accumulator)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[0], 0, 0, 0);
    } else if constexpr (KIND == 3) { // MFMA -> VALU (cvt) -> MFMA chain, as the mask head's layers
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[0], 0, 0, 0);
        b4[0] = (short)__builtin_bit_cast(int, acc[0][0]);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e30f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int KIND>
static void run(const char *name, int block, float *sink, unsigned long long *cyc, std::vector<unsigned long long> &h)
{
  const int reps = 2000;
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(block), 0, 0, reps, sink, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  const int nw = block / 64;
  double sum = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) sum += (double)h[b * 16 + w];
  const double per = sum / (256.0 * nw) / (reps * 8.0);
  printf("%-44s %d waves/SIMD: %7.2f wave-cycles per MFMA, %6.2f SIMD-cycles per MFMA\n", name, nw / 4, per, per / (nw / 4.0));
}
int main()
{
  float *sink; unsigned long long *cyc;
  hipMalloc(&sink, 64); hipMalloc(&cyc, 256 * 16 * 8);
  std::vector<unsigned long long> h(256 * 16);
  for (int block : {256, 512, 768}) {
    run<0>("16x16x16 bf16_1k, 8 independent", block, sink, cyc, h);
    run<1>("16x16x32 bf16, 8 independent", block, sink, cyc, h);
    run<2>("16x16x16 bf16_1k, one accumulator chain", block, sink, cyc, h);
    run<3>("16x16x16 -> VALU -> 16x16x16 chain", block, sink, cyc, h);
  }
  return 0;
}
