// Helpers shared by the windowed-LDS MSDeformAttn kernels (msda_forward_win.hip, msda_backward_win.hip):
// DPP moves, packed-u16 min/max wave reductions and the per-level window record.
#pragma once
#include <mutex>

#include "msda_common.hpp"

namespace pct {

constexpr int WIN_BLOCK = 256;
constexpr int WIN_TH = 8;              // tile height in pyramid mode (tile width = TQ / 8)
constexpr int WIN_MAXL = 8;

static __device__ const float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // source of the zero apron texels

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u(unsigned v)
{
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true);
}
// quad_perm control that makes every lane of a QL-lane group read lane `src` of its own group
template <int QL, int SRC>
struct BcastCtrl {
  static constexpr int value = QL == 4 ? (SRC | (SRC << 2) | (SRC << 4) | (SRC << 6))
                                       : (SRC | (SRC << 2) | ((2 + SRC) << 4) | ((2 + SRC) << 6));
};

// full-wave reduction of packed (u16, u16) values; result in every lane (uniform)
template <bool IS_MIN>
__device__ __forceinline__ unsigned wave_reduce_pk(unsigned v)
{
  auto op = [](unsigned a, unsigned b) { return IS_MIN ? pk_min(a, b) : pk_max(a, b); };
  v = op(v, dpp_u<0xB1>(v));    // quad_perm [1,0,3,2]
  v = op(v, dpp_u<0x4E>(v));    // quad_perm [2,3,0,1]
  v = op(v, dpp_u<0x141>(v));   // row_half_mirror
  v = op(v, dpp_u<0x140>(v));   // row_mirror
  const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return op(op(r0, r1), op(r2, r3));
}

// LDS integer atomic add without return (ds_add_u32).  Measured on MI355X (tools/micro/lds_atomic_rate.hip): 4.3
// cycles per conflict-free wave instruction per CU, against 193 for ds_add_f32 / ds_pk_add_f16 whatever the layout --
// LDS float atomics run at ~1 lane per 3 clocks and are not usable on a hot path.
__device__ __forceinline__ void lds_add(int *p, int v)
{
  __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)p, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// full-wave unsigned max; result in every lane (uniform)
__device__ __forceinline__ unsigned wave_reduce_umax(unsigned v)
{
  v = max(v, dpp_u<0xB1>(v));
  v = max(v, dpp_u<0x4E>(v));
  v = max(v, dpp_u<0x141>(v));
  v = max(v, dpp_u<0x140>(v));
  const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return max(max(r0, r1), max(r2, r3));
}

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KB for a kernel, once per (kernel, device): a function-local
// `static const hipError_t rc = hipFuncSetAttribute(...)` ran once per PROCESS, i.e. only for the device that happened to be
// current at the first launch (fine for one process per GPU, wrong for a process that drives several).
inline hipError_t func_attr_per_device(const void *fn, const int bytes = 160 * 1024)   // (bytes: dynamic LDS only; static LDS counts against the 160 KB too)
{
  constexpr int MAXD = 64, MAXF = 64;
  static std::mutex mu;
  static const void *fns[MAXF] = {};
  static unsigned long long done[MAXF] = {};                     // bit per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXD) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  int slot = -1;
  for (int i = 0; i < MAXF; ++i) {
    if (fns[i] == fn) { slot = i; break; }
    if (!fns[i]) { fns[i] = fn; slot = i; break; }
  }
  if (slot >= 0 && (done[slot] >> dev) & 1ull) return hipSuccess;
  const hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (rc == hipSuccess && slot >= 0) done[slot] |= 1ull << dev;
  return rc;
}

// A set of 8 zeroed per-XCD work-item counters for one launch of a persistent windowed kernel (defined in
// msda_forward_win.hip; nullptr = use the static item stride).  Protocol inside the kernels: the first round of items is
// static (workgroup slot), every further item of XCD x is nslots + atomicAdd(queue + x, 1); every processed item costs
// exactly one fetch, so the fetch that returns n_x - 1 is the launch's last one and zeroes the counter again.
// The 8 counters of a set are WIN_QUEUE_STRIDE unsigned apart (counter of XCD x: queue + x * WIN_QUEUE_STRIDE): 256 bytes, each
// on a memory line of its own.  Packed into one 32-byte run (rounds 1-3) the eight counters shared a line, their returning
// atomics were served one after the other wherever that line lives, and the whole chip could draw at most ~85 items per
// microsecond: a launch with every memory operation, staging and gather knocked out took 1.03 ms for the north-star's
// 87 040 items whether 1, 2 or 3 workgroups per CU ran it -- the figure rounds 2 and 3 read as "the record stream's
// skeleton" -- and 0.34 ms with the counters apart (profiles/r04_forward_structural_ab.txt).  The product kernels draw
// ~40 items per microsecond and hide the fetch two items ahead, so they did not get faster; the cap is gone.
#ifndef PCT_QUEUE_STRIDE
#define PCT_QUEUE_STRIDE 64
#endif
constexpr int WIN_QUEUE_STRIDE = PCT_QUEUE_STRIDE;
unsigned *win_queue_slot(hipStream_t stream);
int prepare_win_queue_device();     // allocate the ring now (outside any capture)

// One level's window, derived identically by every lane from the 4 per-wave boxes in LDS.
struct LevelWindow {
  int x0, y0, wid, size;   // origin, width (pixels), pixel count (0 = no gated sample on this level)
};
__device__ __forceinline__ LevelWindow read_window(const unsigned *bb, const int L, const int l)
{
  unsigned lo = bb[l * 2], hi = bb[l * 2 + 1];
#pragma unroll
  for (int w = 1; w < WIN_BLOCK / 64; ++w) {
    lo = pk_min(lo, bb[(w * L + l) * 2]);
    hi = pk_max(hi, bb[(w * L + l) * 2 + 1]);
  }
  lo = __builtin_amdgcn_readfirstlane(lo);
  hi = __builtin_amdgcn_readfirstlane(hi);
  const int x0 = (int)(lo & 0xFFFFu) - 1, y0 = (int)(lo >> 16) - 1;      // un-bias: origin may be -1 (apron)
  const int x1 = (int)(hi & 0xFFFFu) - 1, y1 = (int)(hi >> 16) - 1;
  LevelWindow w;
  const bool empty = x0 > x1 || y0 > y1;
  w.x0 = x0;
  w.y0 = y0;
  w.wid = empty ? 1 : x1 - x0 + 1;
  w.size = empty ? 0 : w.wid * (y1 - y0 + 1);
  return w;
}


}  // namespace pct
