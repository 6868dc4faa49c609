// Micro-benchmark: throughput of LDS float / integer atomics on gfx950 (development tool, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride)
{
  __shared__ __attribute__((aligned(16))) float lds[16384];
  unsigned long long *lds64 = reinterpret_cast<unsigned long long *>(lds);
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x;
  float *p = lds + (lane * stride) % 8192;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float *q = p + u * 264;
      if constexpr (MODE == 0)
        __hip_atomic_fetch_add((__attribute__((address_space(3))) float *)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if constexpr (MODE == 1)
        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)q, (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if constexpr (MODE == 2) {
        *(volatile float *)q = *(volatile float *)q + v;
      } else if constexpr (MODE == 3) {
        v += __hip_atomic_fetch_add((__attribute__((address_space(3))) float *)q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if constexpr (MODE == 5) {      // 64-bit integer add: two 32-bit fields per instruction
        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long *)(lds64 + ((lane * stride) % 4096) + (u * 264) % 4096),
                               (unsigned long long)lane * 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if constexpr (MODE == 6) {      // random 32-bit: lane -> pseudo-random dword
        unsigned h = (unsigned)(lane * 2654435761u + u * 40503u + it * 97u);
        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)(lds + ((h >> 7) % 8192)), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if constexpr (MODE == 7) {      // random 64-bit
        unsigned h = (unsigned)(lane * 2654435761u + u * 40503u + it * 97u);
        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long *)(lds64 + ((h >> 7) % 4096)), (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if constexpr (MODE == 4) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 hv = {(_Float16)1.0f, (_Float16)2.0f};
        __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2 *)q, hv);
      }
    }
  }
  __syncthreads();
  if (out) out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + v;
}

template <int MODE>
void run(const char *name, int stride, int wg_per_cu)
{
  float *out;
  hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 10, stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, stride);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)iters * 16 * 4 * wg_per_cu;   // wave-instructions per CU
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s stride %2d  wg/cu %d : %.3f ms  -> %.1f cycles per wave-instruction per CU\n", name, stride, wg_per_cu, ms,
         cyc / instr_per_cu);
  hipFree(out);
}

int main()
{
  for (int wg : {2, 4}) {
    run<0>("ds_add_f32", 1, wg);
    run<0>("ds_add_f32", 16, wg);
    run<0>("ds_add_f32 (same addr)", 0, wg);
    run<1>("ds_add_u32", 1, wg);
    run<1>("ds_add_u32", 16, wg);
    run<2>("read+add+write", 1, wg);
    run<3>("ds_add_rtn_f32", 1, wg);
    run<4>("ds_pk_add_f16", 1, wg);
    run<5>("ds_add_u64", 1, wg);
    run<5>("ds_add_u64", 16, wg);
    run<6>("ds_add_u32 random", 1, wg);
    run<7>("ds_add_u64 random", 1, wg);
  }
  return 0;
}
