// Micro-benchmark (diagnostic, not part of the library): issue rate of the vector instructions the MSDeformAttn gather is
// made of, by waves per SIMD.  One workgroup per CU; BLOCK = 256 / 512 / 768 / 1024 threads = 1 / 2 / 3 / 4 waves per SIMD.
// Every wave runs REPS iterations of a 32-instruction body of ONE kind on 8 (packed: 8 x 2) independent accumulators and
// stamps s_memtime around the loop; the host prints cycles per instruction per SIMD (= wave cycles / instructions / waves
// per SIMD, the number that compares with the guide's "2 cycles per v_fma_f32 at full rate, 4 from one wave alone").
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_FMA = 0, K_PKFMA, K_PKMUL, K_CNDMASK, K_DPPMOV, K_LDS128, K_MIX_PK, K_MIX_FMA, K_CVT, K_CND_S, K_CND_IND, K_BFI, K_MOV, K_ADDU, K_MAD24, K_FRACT, K_EXP, K_LDS128_CF, K_MIX_PK_CF, K_MIX_PK_RND, K_LDS128_RND, K_PIPE_CF, K_PIPE_RND, K_GATE_VCC, K_GATE_SGPR, K_NKINDS };
static const char *kind_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cndmask_b32", "v_mov_b32 dpp",
                                  "ds_read_b128", "8 ds_read_b128 + 32 v_pk_fma_f32", "8 ds_read_b128 + 64 v_fma_f32",
                                  "v_cvt_flr_i32_f32", "v_cndmask_b32_e64 (sgpr mask)", "v_cndmask_b32 vcc, independent",
                                  "v_bfi_b32", "v_mov_b32", "v_add_u32", "v_mad_u32_u24", "v_fract_f32", "v_exp_f32",
                                  "ds_read_b128 conflict-free", "8 ds_read_b128 (c-free) + 32 v_pk_fma_f32",
                                  "8 ds_read_b128 (random px) + 32 v_pk_fma", "ds_read_b128 random pixels",
                                  "pipelined 4 reads || 16 pk_fma (c-free), per 40", "pipelined 4 reads || 16 pk_fma (random), per 40",
                                  "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (pair)", "v_cmp_lt_f32_e64 s + v_cndmask_b32_e64 s (pair)"};

template <int KIND>
__global__ void rate_kernel(const int reps, float *sink, unsigned long long *cyc)
{
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i) * 1e-3f;
  float x = 1.0001f, y = 0.9999f;
  const unsigned addr = (threadIdx.x & 63) * 64 + ((threadIdx.x >> 6) & 3) * 4096;   // conflict-free: lanes 64 B apart, 4 reads
  // conflict-free: lane L at pixel L (64 B apart), piece rotated per 8-lane block as the kernel does: (lane / 8) % 4
  const unsigned ln = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  const unsigned addr_cf = ln * 64 + (((ln >> 3) & 3) << 4) + wv * 4096;
  // "random pixels": a hashed pixel index per lane (M-like: the 4 lanes of a quad meet on p mod 4 at random)
  const unsigned hp = (ln * 2654435761u + wv * 40503u + blockIdx.x * 97u) >> 7;
  const unsigned addr_rnd = (hp % 60u) * 64 + (((ln >> 3) & 3) << 4) + wv * 4096;
  f32x4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_barrier();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < reps; ++r) {
    if constexpr (KIND == K_FMA) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
    } else if constexpr (KIND == K_PKFMA) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x2 p = {a[2 * i], a[2 * i + 1]};
          const f32x2 xx = {x, x}, yy = {y, y};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(xx), "v"(yy));
          a[2 * i] = p[0];
          a[2 * i + 1] = p[1];
        }
    } else if constexpr (KIND == K_PKMUL) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x2 p = {a[2 * i], a[2 * i + 1]};
          const f32x2 xx = {x, x};
          asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p) : "v"(xx));
          a[2 * i] = p[0];
          a[2 * i + 1] = p[1];
        }
    } else if constexpr (KIND == K_CNDMASK) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
    } else if constexpr (KIND == K_DPPMOV) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i + 8]));
    } else if constexpr (KIND == K_CVT) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(a[i]) : "v"(a[i + 8]));
    } else if constexpr (KIND == K_LDS128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"((i & 3) * 16 + (i >> 2) * 128));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                     "+v"(v[7]));
      }
    } else if constexpr (KIND == K_CND_S) {
      const unsigned long long msk = 0x0F0F33335555AAAAull + (unsigned long long)reps;   // uniform: an SGPR pair
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i + 8]), "v"(x), "s"(msk));
    } else if constexpr (KIND == K_CND_IND) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i + 8]), "v"(x));
    } else if constexpr (KIND == K_BFI) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(y), "v"(a[i + 8]), "v"(x));
    } else if constexpr (KIND == K_MOV) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[i + 8]));
    } else if constexpr (KIND == K_ADDU) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
    } else if constexpr (KIND == K_MAD24) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
    } else if constexpr (KIND == K_FRACT) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fract_f32 %0, %1" : "=v"(a[i]) : "v"(a[i + 8]));
    } else if constexpr (KIND == K_EXP) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i + 8]));
    } else if constexpr (KIND == K_LDS128_CF || KIND == K_LDS128_RND) {
      const unsigned ad = KIND == K_LDS128_CF ? addr_cf : addr_rnd;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(i & 1 ? ad ^ 16u : ad), "i"((i >> 1) * 256));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                     "+v"(v[7]));
      }
    } else if constexpr (KIND == K_MIX_PK_CF || KIND == K_MIX_PK_RND) {
      const unsigned ad = KIND == K_MIX_PK_CF ? addr_cf : addr_rnd;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(i & 1 ? ad ^ 16u : ad), "i"((i >> 1) * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                   "+v"(v[7]));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          f32x2 p = {a[(i & 3) * 4 + 2 * e], a[(i & 3) * 4 + 2 * e + 1]};
          const f32x2 xx = {x, x}, d = {v[i][2 * e], v[i][2 * e + 1]};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(xx), "v"(d));
          a[(i & 3) * 4 + 2 * e] = p[0];
          a[(i & 3) * 4 + 2 * e + 1] = p[1];
        }
    } else if constexpr (KIND == K_PIPE_CF || KIND == K_PIPE_RND) {
      // the same 8 reads + 32 packed FMAs per iteration, as a rolling pipeline at CORNER granularity: the next corner's
      // four reads are issued before the previous corner's 16 FMAs, which wait for all but the four youngest reads
      const unsigned ad = KIND == K_PIPE_CF ? addr_cf : addr_rnd;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[4 * h + i]) : "v"(i & 1 ? ad ^ 16u : ad), "i"((i >> 1) * 256 + h * 512));
        // the OTHER half's reads (issued one step ago) are complete once at most 4 are outstanding
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v[4 * (1 - h)]), "+v"(v[4 * (1 - h) + 1]), "+v"(v[4 * (1 - h) + 2]), "+v"(v[4 * (1 - h) + 3]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 p = {a[i * 4 + 2 * e], a[i * 4 + 2 * e + 1]};
            const f32x2 xx = {x, x}, d = {v[4 * (1 - h) + i][2 * e], v[4 * (1 - h) + i][2 * e + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(xx), "v"(d));
            a[i * 4 + 2 * e] = p[0];
            a[i * 4 + 2 * e + 1] = p[1];
          }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 p = {a[i * 4 + 2 * e], a[i * 4 + 2 * e + 1]};
            const f32x2 xx = {y, y}, d = {v[4 * (1 - h) + i][2 * e], v[4 * (1 - h) + i][2 * e + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(xx), "v"(d));
            a[i * 4 + 2 * e] = p[0];
            a[i * 4 + 2 * e + 1] = p[1];
          }
      }
    } else if constexpr (KIND == K_GATE_VCC) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_cmp_lt_f32_e32 vcc, -1.0, %1\n\tv_cndmask_b32_e32 %0, -2.0, %1, vcc" : "=v"(a[i]) : "v"(a[i + 8]) : "vcc");
    } else if constexpr (KIND == K_GATE_SGPR) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          unsigned long long mk;
          asm volatile("v_cmp_lt_f32_e64 %1, -1.0, %2\n\tv_cndmask_b32_e64 %0, -2.0, %2, %1" : "=v"(a[i]), "=&s"(mk) : "v"(a[i + 8]));
        }
    } else if constexpr (KIND == K_MIX_PK || KIND == K_MIX_FMA) {
      // the gather's own shape: 8 reads (one pixel row of a sample, two corners), then their FMAs
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "i"((i & 3) * 16 + (i >> 2) * 128));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                   "+v"(v[7]));
      if constexpr (KIND == K_MIX_PK) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            f32x2 p = {a[(i & 3) * 4 + 2 * e], a[(i & 3) * 4 + 2 * e + 1]};
            const f32x2 xx = {x, x}, d = {v[i][2 * e], v[i][2 * e + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(xx), "v"(d));
            a[(i & 3) * 4 + 2 * e] = p[0];
            a[(i & 3) * 4 + 2 * e + 1] = p[1];
          }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[(i & 3) * 4 + e]) : "v"(x), "v"(v[i][e]));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  if (s == 1.2345e30f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const int block, const int reps, float *sink, unsigned long long *cyc, std::vector<unsigned long long> &h)
{
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(block), 0, 0, reps, sink, cyc);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(block), 0, 0, reps, sink, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  const int nw = block / 64;
  double sum = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < nw; ++w) sum += (double)h[b * 16 + w];
  const double per_wave = sum / (256.0 * nw);
  const int insts = (KIND == K_MIX_PK || KIND == K_MIX_PK_CF || KIND == K_MIX_PK_RND || KIND == K_PIPE_CF || KIND == K_PIPE_RND) ? 40 : (KIND == K_MIX_FMA ? 72 : 32);
  const double wps = nw / 4.0;
  printf("%-34s %4d threads (%d waves/SIMD): %8.2f wave-cycles per instruction, %6.2f SIMD-cycles per instruction\n",
         kind_name[KIND], block, nw / 4, per_wave / ((double)reps * insts), per_wave / ((double)reps * insts) / wps);
}

int main()
{
  float *sink;
  unsigned long long *cyc;
  hipMalloc(&sink, 64);
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(256 * 16);
  const int reps = 2000;
  for (int block : {256, 512, 768, 1024}) {
    run<K_FMA>(block, reps, sink, cyc, h);
    run<K_PKFMA>(block, reps, sink, cyc, h);
    run<K_PKMUL>(block, reps, sink, cyc, h);
    run<K_CNDMASK>(block, reps, sink, cyc, h);
    run<K_DPPMOV>(block, reps, sink, cyc, h);
    run<K_CVT>(block, reps, sink, cyc, h);
    run<K_LDS128>(block, reps, sink, cyc, h);
    run<K_MIX_PK>(block, reps, sink, cyc, h);
    run<K_MIX_FMA>(block, reps, sink, cyc, h);
    run<K_CND_S>(block, reps, sink, cyc, h);
    run<K_CND_IND>(block, reps, sink, cyc, h);
    run<K_BFI>(block, reps, sink, cyc, h);
    run<K_MOV>(block, reps, sink, cyc, h);
    run<K_ADDU>(block, reps, sink, cyc, h);
    run<K_MAD24>(block, reps, sink, cyc, h);
    run<K_FRACT>(block, reps, sink, cyc, h);
    run<K_EXP>(block, reps, sink, cyc, h);
    run<K_LDS128_CF>(block, reps, sink, cyc, h);
    run<K_LDS128_RND>(block, reps, sink, cyc, h);
    run<K_MIX_PK_CF>(block, reps, sink, cyc, h);
    run<K_MIX_PK_RND>(block, reps, sink, cyc, h);
    run<K_PIPE_CF>(block, reps, sink, cyc, h);
    run<K_PIPE_RND>(block, reps, sink, cyc, h);
    run<K_GATE_VCC>(block, reps, sink, cyc, h);
    run<K_GATE_SGPR>(block, reps, sink, cyc, h);
  }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
