// Shared device helpers for the MSDeformAttn kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pct {

template <typename T, int N>
using vec_t = T __attribute__((ext_vector_type(N)));

// storage type tags for the 16-bit paths (raw bit patterns cross the C ABI)
struct half_bits { uint16_t u; };
struct bf16_bits { uint16_t u; };

template <typename T> struct Traits;
template <> struct Traits<float> {
  using acc_t = float;
  using store_t = float;
  static __device__ __forceinline__ float to_acc(float v) { return v; }
  static __device__ __forceinline__ float from_acc(float v) { return v; }
};
template <> struct Traits<double> {
  using acc_t = double;
  using store_t = double;
  static __device__ __forceinline__ double to_acc(double v) { return v; }
  static __device__ __forceinline__ double from_acc(double v) { return v; }
};
template <> struct Traits<half_bits> {
  using acc_t = float;
  using store_t = _Float16;
  static __device__ __forceinline__ float to_acc(_Float16 v) { return (float)v; }
  static __device__ __forceinline__ _Float16 from_acc(float v) { return (_Float16)v; }
};
template <> struct Traits<bf16_bits> {
  using acc_t = float;
  using store_t = __bf16;
  static __device__ __forceinline__ float to_acc(__bf16 v) { return (float)v; }
  static __device__ __forceinline__ __bf16 from_acc(float v) { return (__bf16)v; }  // RNE, NaN-preserving
};

// Smallest stride >= n (in elements of `elem_dwords` dwords each) such that stride*elem_dwords == 2 (mod 4)
// dwords: consecutive (query, head) records then land on distinct 2-bank LDS slots for ds_read_b64 (64 banks)
// and on distinct banks for ds_read_b32 (32 banks) within a 32-lane group.
__host__ __device__ inline int padded_record_stride(int n_dwords)
{
  int s = n_dwords;
  while ((s & 3) != 2) ++s;
  return s;
}

// XCD-aware block remap: the dispatcher deals consecutive block ids round-robin over the 8 XCDs, so ids b and
// b+8 share an L2.  Give each XCD one contiguous chunk of the logical block range so that blocks that gather
// neighbouring pixels hit the same L2.  Bijective for any grid size (the ragged tail keeps its identity).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk)
{
  const unsigned per = nblk >> 3;
  return (bid < (per << 3)) ? (bid & 7u) * per + (bid >> 3) : bid;
}

}  // namespace pct
