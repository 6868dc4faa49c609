// probe: what v_permlane16_swap / v_permlane32_swap do to a lane-id pattern (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o)
{
  const unsigned l = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main()
{
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
