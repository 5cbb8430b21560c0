// k_lazy.hip -- the lazy domain stage: only (representative, profile) pairs that can still win ItsPosition's argmax go
// through Backward, decoding, envelope re-scoring and the ensemble stage.
//
// What the consumer reads.  ItsPosition.parse/_score (itsxpress/SeqSample.py:400-461) keeps, per target sequence and side
// (profile NAME prefix), the FIRST domain row with the strictly greatest %.1f score; rows below the winner are thrown away.
// hmmsearch computes every row all the same (itsxpress/SeqSample.py:191-209); on amplicon data ~130 profiles of a side model
// the same flank, so 98.6 % of the rows lose.
//
// The bound.  For a pair whose multihit Forward score over the whole target is fwdsc (nats), every domain's bit score obeys
//     bits <= (fwdsc - nullsc) / ln 2 + C(n),   C(n) = 1 + (2 ln(2 (n + 3) / (3 (n + 2))) + n ln((n + 3) / (n + 2))) / ln 2  (~1.27)
// (scripts/lazy_bound.py derives it: every path of the envelope's unihit Forward sum is a path of the multihit sum with its
// flanks in N and C, and dombias >= 0; measured slack 1.9-2.6 bits, 0 violations).  LenTables::lazy_c holds C(n) plus a margin
// of 0.02 bits for float rounding (both sides are sums of positive products: relative error <= ~3 (n + M) 2^-24).
//
// The schedule (engine.hip: lazy_rounds), per chunk of representatives:
//   pass A   Forward score of EVERY pair past the MSV filter (k_filters_fwd<., 1>: nothing stored but the score);
//   round 1  per (representative, class) the pair with the best bound goes through the whole domain pipeline;
//   round 2  every pair whose bound, as %.1f tenths, reaches the group's best CERTAIN row so far (k_compact_best: reported for
//            every domZ the data set can have) -- ties included, they go to the EARLIER row -- and every pair of a group that
//            has no certain row.  The winner of the full table is among them, and so is every row that ties with it.
// Exactness of the thresholds: hmmsearch's domZ (reported targets per profile) is not known when pairs are skipped, only
// bounds on it -- the reported targets among the evaluated pairs below, the pairs past the MSV filter above.  k_finalize_lazy
// decides every row that both bounds decide alike; a row that neither decides and that could change a winner or the "sequence
// has a row" flag is counted (k_lazy_pending), and the search is then repeated in full (engine.hip: itsx_search_finalize).
#include "engine.h"
#include "k_api.h"
#include "detmath.h"

namespace itsx {

// largest %.1f tenths (biased) a domain of each pair can print, and the best-bound pair of every (representative, class)
__global__ void __launch_bounds__(256) k_lazy_bound(LazyArgs a)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.NP) return;
  const PairRec pr = a.pairs[i];
  if (pr.prof < 0) { a.b10[i] = 0u; return; }
  const float f = a.fb[i];
  const LenTables lt = a.lt[pr.L];
  uint32_t b;
  if (f != f || f == __builtin_inff()) b = (1u << 24) - 1;                  // no usable bound: always evaluated
  else if (f == -__builtin_inff()) b = 1u;
  else {
    const double bits = ((double)f - (double)lt.nullsc) / 0.69314718055994529 + (double)lt.lazy_c;
    double t = __builtin_floor(bits * 10.0 + 0.5) + (double)LAZY_TENTHS_BIAS;      // round half UP: never below rint()
    if (t < 1.0) t = 1.0;
    if (t > (double)((1 << 24) - 1)) t = (double)((1 << 24) - 1);
    b = (uint32_t)t;
  }
  a.b10[i] = b;
  const size_t g = (size_t)pr.useq * a.ncls + a.cls[pr.prof];
  atomicMax(&a.gtop[g], ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
}

// round 0: the best-bound pair of each group.  round 1: everything not yet evaluated that could still beat (or tie with) the
// group's best certain row; all of a group that has none.
__global__ void __launch_bounds__(256) k_lazy_mark(LazyArgs a, int round)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > a.NP) return;
  int f = 0;
  if (i < a.NP) {
    const PairRec pr = a.pairs[i];
    if (pr.prof >= 0 && !a.done[i]) {
      const int c = a.cls[pr.prof];
      if (round == 0) f = (uint32_t)(a.gtop[(size_t)pr.useq * a.ncls + c] & 0xFFFFFFFFull) == 0xFFFFFFFFu - (uint32_t)i;
      else {
        const unsigned long long bc = a.bestc[(size_t)a.sorted_uniq[pr.useq] * a.ncls + c];
        f = bc == 0ull || a.b10[i] >= (uint32_t)(bc >> 40);
      }
      if (f) a.done[i] = 1;
    }
  }
  a.flag[i] = f;
}

// selected pairs keep their profile segment and their order inside it (ascending length): out[seg_new[p] + rank inside p]
__global__ void __launch_bounds__(256) k_lazy_scatter(const PairRec *__restrict__ pairs, int64_t NP, const int32_t *__restrict__ flag,
                                                      const int32_t *__restrict__ pos, const int64_t *__restrict__ seg_old,
                                                      const int64_t *__restrict__ seg_new, PairRec *__restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP || !flag[i]) return;
  const PairRec pr = pairs[i];
  out[seg_new[pr.prof] + (int64_t)(pos[i] - pos[seg_old[pr.prof]])] = pr;
}

__global__ void __launch_bounds__(256) k_finalize_lazy(itsx_domain *__restrict__ dom, int64_t n, const int64_t *__restrict__ zlb,
                                                       const int64_t *__restrict__ zub, double domE, const int32_t *__restrict__ usample, int P)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0) return;
  int rep = 0;
  if (d.seq_reported) {
    const size_t z = (size_t)(usample ? usample[d.rep] * P : 0) + d.prof;
    const double p = det_exp(d.lnP);
    // this target is reported itself, so the true count is at least 1 and at least the lower bound
    const double lo = (double)(zlb[z] > 1 ? zlb[z] : 1), hi = (double)(zub[z] > 1 ? zub[z] : 1);
    rep = (p * hi <= domE) ? 1 : (p * lo <= domE) ? 2 : 0;
  }
  dom[i].dom_reported = rep;
}

__global__ void __launch_bounds__(256) k_lazy_sure(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                   unsigned long long *__restrict__ sure, int32_t *__restrict__ has)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || d.dom_reported != 1) return;
  has[d.rep] = 1;
  atomicMax(&sure[(size_t)d.rep * ncls + cls[d.prof]], rank_key(d));
}
__global__ void __launch_bounds__(256) k_lazy_pending(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                      const unsigned long long *__restrict__ sure, const int32_t *__restrict__ has,
                                                      unsigned long long *__restrict__ count, int32_t *__restrict__ prof_flag)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  if (i < n) {
    const itsx_domain d = dom[i];
    // an undecided row matters when it would beat its group's best sure row, or when the sequence has no sure row at all
    if (d.dom_idx >= 0 && d.dom_reported == 2) c = !has[d.rep] || rank_key(d) > sure[(size_t)d.rep * ncls + cls[d.prof]];
    if (c) prof_flag[d.prof] = 1;
  }
  const unsigned long long m = __ballot(c);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
}

void launch_lazy_bound(const LazyArgs &a, hipStream_t st)
{
  if (a.NP > 0) hipLaunchKernelGGL(k_lazy_bound, dim3((unsigned)((a.NP + 255) / 256)), dim3(256), 0, st, a);
}
void launch_lazy_mark(const LazyArgs &a, int round, hipStream_t st)
{
  hipLaunchKernelGGL(k_lazy_mark, dim3((unsigned)((a.NP + 1 + 255) / 256)), dim3(256), 0, st, a, round);
}
void launch_lazy_scatter(const PairRec *pairs, int64_t NP, const int32_t *flag, const int32_t *pos, const int64_t *seg_old, const int64_t *seg_new,
                         PairRec *out, hipStream_t st)
{
  if (NP > 0) hipLaunchKernelGGL(k_lazy_scatter, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, pairs, NP, flag, pos, seg_old, seg_new, out);
}
void launch_finalize_lazy(itsx_domain *dom, int64_t n, const int64_t *zlb, const int64_t *zub, double domE, const int32_t *usample, int P, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_finalize_lazy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, zlb, zub, domE, usample, P);
}
void launch_lazy_sure(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, unsigned long long *sure, int32_t *has, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_lazy_sure, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, sure, has);
}
void launch_lazy_pending(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, const unsigned long long *sure, const int32_t *has,
                         unsigned long long *count, int32_t *prof_flag, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_lazy_pending, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, sure, has, count, prof_flag);
}

}  // namespace itsx
