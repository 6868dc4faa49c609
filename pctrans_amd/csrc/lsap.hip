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
// bookkeeping, so one problem = one wavefront: lanes stride over the queries, reductions are shuffle butterflies, the
// state lives in LDS; the workgroup is that single wave, so its barriers cost nothing.  Arithmetic is fp64 as in scipy
// (the fp32 costs are widened), so the dual updates take the same decisions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace pct {

constexpr int LSAP_MAXQ = 1024;                   // queries per problem (LDS-resident state)
constexpr int LSAP_MAXG = 512;                    // targets per problem

// cost [B, Q, ldg] fp32 (row = query, column = target; only the first G[b] columns are used),
// row_for_target [B, ldg] int32 (query assigned to target j, -1 for j >= G[b])
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ cost, const int Q, const int ldg,
                                                  const int *__restrict__ G, int *__restrict__ row_for_target,
                                                  int *__restrict__ status)
{
  __shared__ double v[LSAP_MAXQ];                 // dual of query j
  __shared__ double shortest[LSAP_MAXQ];
  __shared__ double u[LSAP_MAXG];                 // dual of target i
  __shared__ int path[LSAP_MAXQ];                 // predecessor target on the shortest path to query j
  __shared__ int tgt_of_query[LSAP_MAXQ];         // "row4col": target currently holding query j, -1 = free
  __shared__ int query_of_tgt[LSAP_MAXG];         // "col4row"
  __shared__ unsigned char scanned_q[LSAP_MAXQ];  // SC
  __shared__ unsigned char scanned_t[LSAP_MAXG];  // SR

  const int b = blockIdx.x, lane = threadIdx.x;
  const int g = G[b];                             // device data: validated here, not trusted
  const float *C = cost + (size_t)b * Q * ldg;
  int *out = row_for_target + (size_t)b * ldg;
  for (int j = lane; j < ldg; j += 64) out[j] = -1;
  // (the kernel clears its own status word: a hipMemsetAsync recorded into a HIP graph does not replay reliably on ROCm
  // 7.2, msda_backward.hip; one lane, one wave per problem -- the later `status[b] = 1` of the same lane is ordered behind it)
  if (lane == 0) status[b] = 0;
  if (g <= 0) return;
  if (g > ldg || g > Q || g > LSAP_MAXG) {        // more targets than columns / queries: no assignment exists
    if (lane == 0) status[b] = 1;
    return;
  }
  // scipy raises on any NaN or -inf entry ("matrix contains invalid numeric entries"): report those the same way
  // (+inf entries are allowed: an infeasible problem is detected by the search below)
  {
    int bad = 0;
    for (int j = lane; j < Q; j += 64)
      for (int i = 0; i < g; ++i) {
        const float c = C[(size_t)j * ldg + i];
        bad |= (c != c) || (c == -INFINITY);
      }
    if (__any(bad)) {
      if (lane == 0) status[b] = 1;
      return;
    }
  }
  for (int j = lane; j < Q; j += 64) {
    v[j] = 0.0;
    tgt_of_query[j] = -1;
  }
  for (int i = lane; i < g; i += 64) {
    u[i] = 0.0;
    query_of_tgt[i] = -1;
  }
  __syncthreads();

  for (int cur = 0; cur < g; ++cur) {
    for (int j = lane; j < Q; j += 64) {
      shortest[j] = INFINITY;
      scanned_q[j] = 0;
    }
    for (int i = lane; i < g; i += 64) scanned_t[i] = 0;
    __syncthreads();

    double min_val = 0.0;
    int i = cur, sink = -1;
    while (sink < 0) {
      if (lane == 0) scanned_t[i] = 1;
      const double ui = u[i];
      // relax every unscanned query through target i, then arg-min over the unscanned queries; ties prefer a free
      // query (scipy: `shortestPathCosts[j] == lowest && row4col[j] == -1`), then the lower index
      double best = INFINITY;
      int best_j = -1, best_free = 0;
      for (int j = lane; j < Q; j += 64) {
        if (scanned_q[j]) continue;
        const double r = min_val + (double)C[(size_t)j * ldg + i] - ui - v[j];
        double s = shortest[j];
        if (r < s) {
          path[j] = i;
          shortest[j] = s = r;
        }
        const int fr = tgt_of_query[j] < 0;
        if (s < best || (s == best && fr > best_free)) {
          best = s;
          best_j = j;
          best_free = fr;
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oj = __shfl_xor(best_j, o), of = __shfl_xor(best_free, o);
        const bool take = oj >= 0 && (best_j < 0 || ob < best || (ob == best && (of > best_free || (of == best_free && oj < best_j))));
        if (take) {
          best = ob;
          best_j = oj;
          best_free = of;
        }
      }
      if (best_j < 0 || best == INFINITY) {         // infeasible (a row of +inf costs): report, leave -1
        if (lane == 0) status[b] = 1;
        return;
      }
      min_val = best;
      const int j = best_j;
      if (lane == 0) scanned_q[j] = 1;
      const int t = tgt_of_query[j];
      if (t < 0) sink = j;
      else i = t;
      __syncthreads();
    }

    // dual update (scipy rectangular_lsap.cpp: u[cur] += minVal; other scanned rows / columns shifted)
    for (int t = lane; t < g; t += 64)
      if (scanned_t[t]) u[t] += (t == cur) ? min_val : min_val - shortest[query_of_tgt[t]];
    for (int j = lane; j < Q; j += 64)
      if (scanned_q[j]) v[j] -= min_val - shortest[j];
    __syncthreads();
    // augment along the path back from the sink (serial, short)
    if (lane == 0) {
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
  for (int t = lane; t < g; t += 64) out[t] = query_of_tgt[t];
}

int launch_lsap(const float *cost, int batch, int num_query, int ld_target, const int *num_target, int *row_for_target,
                int *status, hipStream_t stream)
{
  if (batch == 0) return 0;
  if (num_query > LSAP_MAXQ || ld_target > LSAP_MAXG) return -4;
  hipLaunchKernelGGL(lsap_kernel, dim3((unsigned)batch), dim3(64), 0, stream, cost, num_query, ld_target, num_target,
                     row_for_target, status);
  return (int)hipGetLastError();
}

}  // namespace pct
