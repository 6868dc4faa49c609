// Reproducer (development tool, not part of the library): does a hipMemsetAsync recorded into a HIP graph replay?
//   hipcc -O2 --offload-arch=gfx950 graph_memset_replay.hip -o graph_memset_replay && ./graph_memset_replay
// Graph = { memset(buf, 0, bytes); add_one(buf) }; before every replay the buffer is filled with 7.0 by a plain kernel.
// Expected after every replay: 1.0 everywhere.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void fill(float *p, size_t n, float v)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void add_one(float *p, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
}

// a kernel with as many arguments as the MSDeformAttn backward (17: ~130 bytes of kernel arguments)
__global__ void add_one_many_args(float *p, const long *a1, const long *a2, const float *a3, const float *a4, const float *a5, int i1,
                                  int i2, int i3, int i4, int i5, int i6, int i7, float *o1, float *o2, float *o3, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.0f;
  if (i1 == 12345 && o1) o1[0] = (float)(i2 + i3 + i4 + i5 + i6 + i7) + (a1 ? 1.f : 0.f) + (a2 ? 1.f : 0.f) + (a3 ? 1.f : 0.f) +
                                 (a4 ? 1.f : 0.f) + (a5 ? 1.f : 0.f) + (o2 ? 1.f : 0.f) + (o3 ? 1.f : 0.f);
}

int main()
{
  const size_t sizes[] = {1u << 16, 5570560, 89128960};            // floats: 256 KB, 22 MB (grad_value at batch 2), 356 MB (batch 32)
  for (size_t n : sizes) {
    float *d;
    hipMalloc(&d, n * 4);
    float *h = (float *)malloc(n * 4);
    hipStream_t s;
    hipStreamCreate(&s);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    hipMemsetAsync(d, 0, n * 4, s);
    if (getenv("MANY_ARGS"))
      hipLaunchKernelGGL(add_one_many_args, dim3(1024), dim3(256), 0, s, d, (const long *)nullptr, (const long *)nullptr,
                         (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 1, 2, 3, 4, 5, 6, 7, (float *)nullptr,
                         (float *)nullptr, (float *)nullptr, n);
    else
      hipLaunchKernelGGL(add_one, dim3(1024), dim3(256), 0, s, d, n);
    hipStreamEndCapture(s, &g);
    if (getenv("AUTO_FREE_FLAG"))                  // what torch.cuda.CUDAGraph instantiates with
      hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch);
    else
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (getenv("DESTROY_GRAPH_FIRST")) {          // torch.cuda.CUDAGraph keeps only the executable graph
      hipGraphDestroy(g);
      g = nullptr;
    }
    if (getenv("CHURN_HOST_HEAP")) {              // overwrite whatever host memory the capture may have freed
      for (int it = 0; it < 4000; ++it) {
        const size_t sz = 16 + (size_t)(rand() % (1 << 20));
        unsigned char *q = (unsigned char *)malloc(sz);
        memset(q, 0xAB, sz);
        free(q);
      }
    }
    for (int r = 0; r < 4; ++r) {
      hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, d, n, 7.0f);
      hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < n; ++i)
        if (h[i] != 1.0f) { if (!bad) first = i; ++bad; }
      printf("n = %9zu floats, replay %d: %zu wrong", n, r, bad);
      if (bad) printf(" (first at %zu: %g, next %g %g %g)", first, h[first], h[first + 1], h[first + 2], h[first + 3]);
      printf("\n");
    }
    hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    hipStreamDestroy(s);
    hipFree(d);
    free(h);
  }
  return 0;
}
