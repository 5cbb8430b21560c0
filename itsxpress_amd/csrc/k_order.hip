// k_order.hip -- the processing orders of the shared schedules (k_share.hip) by ONE stable radix sort each.
//
// Round 5 ordered the chains by (batch, depth, sorted position) with a counting pass per depth.  With two-sided sharing (round 6) a
// Forward chain ends where it joins a saved Backward state -- after one block or after fifteen -- and a wave takes its 64 chains to
// the longest one's last row: the order is (batch, depth, LAST ROW, sorted position), so that a wave's chains end together, and the
// Backward chains (the uniques that save a state for somebody) are ordered likewise by (batch, blocks from the end they start at,
// rows they walk).  key = [batch | depth | minor]; rocPRIM's radix sort is stable, so equal keys keep their sorted position (ascending
// length).  Work per search: 6 M keys of 64 bits, ~2 ms.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "engine.h"
#include "k_api.h"

namespace itsx {

static constexpr int KEY_DSHIFT = 32, KEY_BSHIFT = 40;       // minor: 32 bits, depth: 8 bits, batch: 24 bits

__global__ void __launch_bounds__(256) k_order_keys(const uint8_t *__restrict__ depth, const int32_t *__restrict__ minor, const int32_t *__restrict__ runs, int32_t U,
                                                    const int32_t *__restrict__ bstart, int nb, unsigned long long *__restrict__ keys, int32_t *__restrict__ vals)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= U) return;
  unsigned long long k = ~0ull;                               // (a chain that does not run goes to the end)
  if (!runs || runs[s] > 0) {
    int lo = 0, hi = nb;                                      // bstart[lo] <= s < bstart[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bstart[mid] <= s) lo = mid; else hi = mid; }
    k = ((unsigned long long)lo << KEY_BSHIFT) | ((unsigned long long)depth[s] << KEY_DSHIFT) | (unsigned long long)(uint32_t)minor[s];
  }
  keys[s] = k; vals[s] = s;
}
void launch_order_keys(const uint8_t *depth, const int32_t *minor, const int32_t *runs, int32_t U, const int32_t *bstart, int nb, unsigned long long *keys, int32_t *vals, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_order_keys, dim3((U + 255) / 256), dim3(256), 0, st, depth, minor, runs, U, bstart, nb, keys, vals);
}

size_t order_sort_bytes(int64_t n)
{
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr,
                                  (size_t)n, 0, 64, (hipStream_t)0);
  return bytes + 256;
}
int order_sort(void *tmp, size_t bytes, const unsigned long long *kin, unsigned long long *kout, const int32_t *vin, int32_t *vout, int64_t n, hipStream_t st)
{
  if (n <= 0) return 0;
  return (int)rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0, 64, st);
}

// segk[b][d] = first position whose (batch, depth) is at least (b, d); inv[s] = position of s (chains that run); *nvalid = chains that run
__global__ void __launch_bounds__(256) k_order_segk(const unsigned long long *__restrict__ keys, int32_t n, int nb, int32_t *__restrict__ segk)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * SHARE_SEGS) return;
  const unsigned long long want = ((unsigned long long)(i / SHARE_SEGS) << KEY_BSHIFT) | ((unsigned long long)(i % SHARE_SEGS) << KEY_DSHIFT);
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
  segk[i] = lo;
}
__global__ void __launch_bounds__(256) k_order_inv(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ vals, int32_t n, int32_t *__restrict__ inv,
                                                   unsigned long long *__restrict__ nvalid)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  int v = 0;
  if (k < n) { v = keys[k] != ~0ull; if (inv) inv[vals[k]] = v ? k : -1; }
  const unsigned long long m = __ballot(v);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(nvalid, (unsigned long long)__builtin_popcountll(m));
}
void launch_order_segk(const unsigned long long *sorted_keys, int32_t n, int nb, int32_t *segk, int32_t *inv, const int32_t *vals, int32_t U, unsigned long long *nvalid, hipStream_t st)
{
  (void)U;
  if (nb > 0) hipLaunchKernelGGL(k_order_segk, dim3((nb * SHARE_SEGS + 255) / 256), dim3(256), 0, st, sorted_keys, n, nb, segk);
  if (n > 0) hipLaunchKernelGGL(k_order_inv, dim3((n + 255) / 256), dim3(256), 0, st, sorted_keys, vals, n, inv, nvalid);
}

}  // namespace itsx
