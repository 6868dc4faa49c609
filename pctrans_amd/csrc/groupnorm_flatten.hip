// GroupNorm on an NCHW map, written straight into the encoder's token-major buffer, for MI355X (gfx950, wave64):
//   out[n, row_offset + p, c] = (x[n, c, p] - mean[n, g(c)]) * rstd[n, g(c)] * gamma[c] + beta[c]
//
// Replaces `nn.GroupNorm(32, conv_dim)` of the pixel decoder's input projections followed by
// `src.flatten(2).transpose(1, 2)` and the `torch.cat(src_flatten, 1)` over the levels
// (pixel_decoder/msdeformattn.py:220-224, 75-83 of the reference): torch normalises in NCHW (2 passes), then the
// concat re-reads every level with a transposing access pattern (~1 TB/s).  Here:
//   * statistics: one workgroup per (image, group); a group's channels are contiguous in NCHW (cpg*HW floats), summed
//     two-pass (mean, then centred squares; the second pass hits L2);
//   * apply + transpose: one workgroup per (image, 64-pixel strip): 256-B coalesced reads along the pixels of every
//     channel, transpose through a padded LDS tile, 512-B coalesced writes along the channels of every pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pct {

constexpr int GN_BLOCK = 256;

__device__ __forceinline__ float gn_block_sum(float v, float *scratch)
{
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6;
  __syncthreads();                                   // scratch free
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// stats[(n * groups + g) * 2 + {0, 1}] = mean, rstd
__global__ __launch_bounds__(GN_BLOCK) void gn_stats_kernel(const float *__restrict__ x, const long long count,
                                                            const float eps, float *__restrict__ stats)
{
  __shared__ float scratch[4];
  const float *p = x + (long long)blockIdx.x * count;           // this (image, group): `count` contiguous floats
  const long long n4 = count / 4;
  float s = 0.f;
  for (long long i = threadIdx.x; i < n4; i += GN_BLOCK) {
    const float4 v = reinterpret_cast<const float4 *>(p)[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (long long i = n4 * 4 + threadIdx.x; i < count; i += GN_BLOCK) s += p[i];
  const float mean = gn_block_sum(s, scratch) / (float)count;
  float q = 0.f;
  for (long long i = threadIdx.x; i < n4; i += GN_BLOCK) {
    const float4 v = reinterpret_cast<const float4 *>(p)[i];
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  for (long long i = n4 * 4 + threadIdx.x; i < count; i += GN_BLOCK) {
    const float a = p[i] - mean;
    q += a * a;
  }
  const float var = gn_block_sum(q, scratch) / (float)count;
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rsqrtf(var + eps);
  }
}

// C = 128 channels; strip of 64 pixels per workgroup
__global__ __launch_bounds__(GN_BLOCK) void gn_apply_flatten_kernel(
    const float *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ gamma,
    const float *__restrict__ beta, const int HW, const int groups, float *__restrict__ out,
    const long long out_batch_stride, const long long out_offset)
{
  constexpr int C = 128, TP = 64;
  __shared__ float tile[TP][C + 1];
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * TP;
  const int cpg = C / groups;
  const float *xn = x + (long long)n * C * HW;
  // read: thread -> pixel (tid % 64), channels tid/64 + 4*i
  {
    const int px = threadIdx.x & 63, c0 = threadIdx.x >> 6;
    const bool ok = p0 + px < HW;
#pragma unroll 4
    for (int i = 0; i < C / 4; ++i) {
      const int c = c0 + 4 * i;
      const int g = c / cpg;
      const float mean = stats[2 * (n * groups + g)], rstd = stats[2 * (n * groups + g) + 1];
      const float v = ok ? xn[(long long)c * HW + p0 + px] : 0.f;
      tile[px][c] = (v - mean) * rstd * gamma[c] + beta[c];
    }
  }
  __syncthreads();
  // write: thread -> channel (tid % 128), pixels tid/128 + 2*i
  {
    const int c = threadIdx.x & 127, q0 = threadIdx.x >> 7;
    float *on = out + (long long)n * out_batch_stride + out_offset;
#pragma unroll 4
    for (int i = 0; i < TP / 2; ++i) {
      const int px = q0 + 2 * i;
      if (p0 + px < HW) on[(long long)(p0 + px) * C + c] = tile[px][c];
    }
  }
}

int launch_groupnorm_flatten(const float *x, const float *gamma, const float *beta, int N, int C, int HW, int groups,
                             float eps, float *stats, float *out, long long out_batch_stride, long long out_offset,
                             hipStream_t stream)
{
  if (N == 0 || HW == 0) return 0;
  if (C != 128 || groups <= 0 || C % groups) return -4;
  const long long count = (long long)(C / groups) * HW;
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(N * groups)), dim3(GN_BLOCK), 0, stream, x, count, eps, stats);
  hipLaunchKernelGGL(gn_apply_flatten_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)N), dim3(GN_BLOCK), 0, stream, x,
                     stats, gamma, beta, HW, groups, out, out_batch_stride, out_offset);
  return (int)hipGetLastError();
}

}  // namespace pct
