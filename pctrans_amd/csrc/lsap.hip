// Linear sum assignment (Hungarian matching) on the device for MI355X (gfx950, wave64).
//
// Replaces `scipy.optimize.linear_sum_assignment(C.cpu())` of the reference's matcher
// (connectomics/model/loss/matcher.py:154-165, called once per image for each of the decoder's 10 prediction heads:
// 10 x N host synchronisations per training step).  C is the [num_queries, num_targets] cost matrix
// with num_targets <= num_queries; every target gets exactly one query and the total cost is minimal.
//
// Algorithm: the shortest-augmenting-path method scipy itself implements (Crouse, "On implementing 2D rectangular
// assignment algorithms", 2016), run on the transposed problem like scipy does when there are more rows than columns:
// one augmentation per target, each a Dijkstra sweep over the queries.  The sweeps are data-parallel (relax all
// remaining queries, arg-min with the "prefer an unassigned query on ties" rule) and everything else is short serial
// bookkeeping.  One problem = one WORKGROUP of 256 threads (round 4; it was one wavefront: 11.8 ms per training step of
// BASELINE configs[3], 20 problems of 300 x 60): a sweep relaxes every query at once (300 queries = two per lane at
// most), the cost matrix is read ONCE into LDS, transposed to [target][query] so a sweep's reads are consecutive (it
// was a 4-byte load per query at a stride of a row, from L2, in every sweep), the arg-min is a shuffle butterfly per wave
// and one LDS hop across the four.  Arithmetic is fp64 as in scipy (the fp32 costs are widened), the tie rules are
// scipy's, so the dual updates take the same decisions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "msda_win_common.hpp"

namespace pct {

constexpr int LSAP_MAXQ = 1024;                   // queries per problem (LDS-resident state)
constexpr int LSAP_MAXG = 512;                    // targets per problem
constexpr int LSAP_BLOCK = 256;
constexpr int LSAP_COST_LDS = 96 * 1024;          // the transposed cost matrix is kept in LDS up to this size

struct LsapBest {
  double s;
  int j, fr;
};
// scipy's choice between two candidates: the smaller path cost; on ties a free query; then the lower index
__device__ __forceinline__ bool lsap_takes(const LsapBest &cur, const double os, const int oj, const int of)
{
  return oj >= 0 && (cur.j < 0 || os < cur.s || (os == cur.s && (of > cur.fr || (of == cur.fr && oj < cur.j))));
}

// cost [B, Q, ldg] fp32 (row = query, column = target; only the first G[b] columns are used),
// row_for_target [B, ldg] int32 (query assigned to target j, -1 for j >= G[b])
__global__ __launch_bounds__(LSAP_BLOCK) void lsap_kernel(const float *__restrict__ cost, const int Q, const int ldg,
                                                          const int *__restrict__ G, int *__restrict__ row_for_target,
                                                          int *__restrict__ status, const int cost_in_lds)
{
  __shared__ double v[LSAP_MAXQ];                 // dual of query j
  __shared__ double shortest[LSAP_MAXQ];
  __shared__ double u[LSAP_MAXG];                 // dual of target i
  __shared__ int path[LSAP_MAXQ];                 // predecessor target on the shortest path to query j
  __shared__ int tgt_of_query[LSAP_MAXQ];         // "row4col": target currently holding query j, -1 = free
  __shared__ int query_of_tgt[LSAP_MAXG];         // "col4row"
  __shared__ unsigned char scanned_q[LSAP_MAXQ];  // SC
  __shared__ unsigned char scanned_t[LSAP_MAXG];  // SR
  __shared__ double red_s[LSAP_BLOCK / 64];
  __shared__ int red_j[LSAP_BLOCK / 64], red_f[LSAP_BLOCK / 64];
  __shared__ int flag;
  extern __shared__ __attribute__((aligned(16))) float ct[];   // [g][Q] transposed costs (when cost_in_lds)

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = G[b];                             // device data: validated here, not trusted
  const float *C = cost + (size_t)b * Q * ldg;
  int *out = row_for_target + (size_t)b * ldg;
  for (int j = tid; j < ldg; j += LSAP_BLOCK) out[j] = -1;
  // (the kernel clears its own status word: a hipMemsetAsync recorded into a HIP graph did not replay reliably under
  // torch.cuda.graph on ROCm 7.2, msda_backward.hip; one thread writes it, later writes of 1 come behind a barrier)
  if (tid == 0) {
    status[b] = 0;
    flag = 0;
  }
  if (g <= 0) return;
  if (g > ldg || g > Q || g > LSAP_MAXG) {        // more targets than columns / queries: no assignment exists
    if (tid == 0) status[b] = 1;
    return;
  }
  __syncthreads();
  // scipy raises on any NaN or -inf entry ("matrix contains invalid numeric entries"): report those the same way
  // (+inf entries are allowed: an infeasible problem is detected by the search below).  The same pass transposes the
  // matrix into LDS: consecutive threads read consecutive targets of a query (the global layout) ...
  {
    int bad = 0;
    const int n = Q * g;
    for (int e = tid; e < n; e += LSAP_BLOCK) {
      const int j = e / g, i = e - j * g;
      const float c = C[(size_t)j * ldg + i];
      bad |= (c != c) || (c == -INFINITY);
      if (cost_in_lds) ct[i * Q + j] = c;
    }
    if (bad) flag = 1;                            // (benign race: every writer stores 1)
  }
  for (int j = tid; j < Q; j += LSAP_BLOCK) {
    v[j] = 0.0;
    tgt_of_query[j] = -1;
  }
  for (int i = tid; i < g; i += LSAP_BLOCK) {
    u[i] = 0.0;
    query_of_tgt[i] = -1;
  }
  __syncthreads();
  if (flag) {
    if (tid == 0) status[b] = 1;
    return;
  }

  for (int cur = 0; cur < g; ++cur) {
    for (int j = tid; j < Q; j += LSAP_BLOCK) {
      shortest[j] = INFINITY;
      scanned_q[j] = 0;
    }
    for (int i = tid; i < g; i += LSAP_BLOCK) scanned_t[i] = 0;
    __syncthreads();

    double min_val = 0.0;
    int i = cur, sink = -1;
    while (sink < 0) {
      if (tid == 0) scanned_t[i] = 1;
      const double ui = u[i];
      // relax every unscanned query through target i, then arg-min over the unscanned queries; ties prefer a free
      // query (scipy: `shortestPathCosts[j] == lowest && row4col[j] == -1`), then the lower index
      LsapBest best{INFINITY, -1, 0};
      for (int j = tid; j < Q; j += LSAP_BLOCK) {
        if (scanned_q[j]) continue;
        const float c = cost_in_lds ? ct[i * Q + j] : C[(size_t)j * ldg + i];
        const double r = min_val + (double)c - ui - v[j];
        double s = shortest[j];
        if (r < s) {
          path[j] = i;
          shortest[j] = s = r;
        }
        const int fr = tgt_of_query[j] < 0;
        if (s < best.s || (s == best.s && fr > best.fr)) best = LsapBest{s, j, fr};
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const double os = __shfl_xor(best.s, o);
        const int oj = __shfl_xor(best.j, o), of = __shfl_xor(best.fr, o);
        if (lsap_takes(best, os, oj, of)) best = LsapBest{os, oj, of};
      }
      if (lane == 0) {
        red_s[wave] = best.s;
        red_j[wave] = best.j;
        red_f[wave] = best.fr;
      }
      __syncthreads();
      best = LsapBest{red_s[0], red_j[0], red_f[0]};
#pragma unroll
      for (int w = 1; w < LSAP_BLOCK / 64; ++w)
        if (lsap_takes(best, red_s[w], red_j[w], red_f[w])) best = LsapBest{red_s[w], red_j[w], red_f[w]};
      if (best.j < 0 || best.s == INFINITY) {       // infeasible (a row of +inf costs): report, leave -1 (uniform exit)
        if (tid == 0) status[b] = 1;
        return;
      }
      min_val = best.s;
      const int j = best.j;
      const int t = tgt_of_query[j];
      __syncthreads();                              // every thread has read red_* before the next sweep writes them
      if ((j % LSAP_BLOCK) == tid) scanned_q[j] = 1;  // (by the thread that owns query j: the only one that reads its flag)
      if (t < 0) sink = j;
      else i = t;
    }

    // dual update (scipy rectangular_lsap.cpp: u[cur] += minVal; other scanned rows / columns shifted)
    for (int t = tid; t < g; t += LSAP_BLOCK)
      if (scanned_t[t]) u[t] += (t == cur) ? min_val : min_val - shortest[query_of_tgt[t]];
    for (int j = tid; j < Q; j += LSAP_BLOCK)
      if (scanned_q[j]) v[j] -= min_val - shortest[j];
    __syncthreads();
    // augment along the path back from the sink (serial, short)
    if (tid == 0) {
      int j = sink;
      while (true) {
        const int t = path[j];
        tgt_of_query[j] = t;
        const int prev = query_of_tgt[t];
        query_of_tgt[t] = j;
        j = prev;
        if (t == cur) break;
      }
    }
    __syncthreads();
  }
  for (int t = tid; t < g; t += LSAP_BLOCK) out[t] = query_of_tgt[t];
}

int launch_lsap(const float *cost, int batch, int num_query, int ld_target, const int *num_target, int *row_for_target,
                int *status, hipStream_t stream)
{
  if (batch == 0) return 0;
  if (num_query > LSAP_MAXQ || ld_target > LSAP_MAXG) return -4;
  // (the transposed costs use ld_target columns at most: sized for the worst case the caller declares)
  const size_t ct_bytes = (size_t)num_query * (size_t)ld_target * sizeof(float);
  const int in_lds = ct_bytes <= (size_t)LSAP_COST_LDS;
  if (in_lds) {
    const hipError_t rc = func_attr_per_device(reinterpret_cast<const void *>(&lsap_kernel), LSAP_COST_LDS);   // + ~34 KB of static state
    if (rc != hipSuccess) return (int)rc;
  }
  hipLaunchKernelGGL(lsap_kernel, dim3((unsigned)batch), dim3(LSAP_BLOCK), in_lds ? ct_bytes : 0, stream, cost, num_query,
                     ld_target, num_target, row_for_target, status, in_lds);
  return (int)hipGetLastError();
}

}  // namespace pct
