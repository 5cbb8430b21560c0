// k_cluster.hip -- row a2 of the path: greedy centroid clustering on the device.
//
// Replaces `vsearch --cluster_size IN --centroids rep.fa --uc uc.txt --strand both --id X`
// (reference call site itsxpress/SeqSample.py:147-162).  The procedure is the one the CPU test oracle
// restates (PARITY UNPINNED against a real vsearch: no fixture, no binary): queries in label order; per
// strand the distinct unambiguous 8-mers of the query are counted against every centroid; candidates with
// >= min(12, #words) shared words are tried in (shared words desc, length asc, position asc) order by a
// global alignment (+2/-4, gaps 20+2k interior, 2+k terminal) until identity >= X accepts one or 32 are
// rejected; a query without an accepted hit becomes a centroid.
//
// How a sequential greedy runs on 256 CUs and still gives the sequential answer:
//  * speculate: a WINDOW of queries is searched against the centroids that exist at the window start, all
//    queries in parallel;
//  * validate: the queries that found nothing would become centroids; a later query of the same window is
//    AFFECTED if such a new centroid would have entered its candidate walk (enough shared words and a rank
//    above the point where the walk stopped).  The window is cut at the first affected query: everything
//    before it is exactly what the sequential procedure produces, the rest is searched again in the next
//    window against the enlarged centroid set.  The first query of a window is never affected, so the loop
//    always advances; on amplicon data cuts are rare once a few hundred centroids exist.
//
// Data layout: the centroid index is a COLUMN BIT MATRIX bits[65536 words][stride]: row = 8-mer, bit c = centroid
// c holds that word.  Counting the shared words of one query against 2048 centroids is then 64 lanes x one
// 32-bit column word per query word, accumulated with carry-save adders on bit-sliced counters (Harley-Seal):
// about 6 integer ops per query word per 32 centroids, coalesced 256-B row segments, no atomics.  Amplicons
// share their conserved flanks, so nearly every centroid shares words with every query: the dense form is the
// right one (an inverted index would touch the same cells one atomic at a time).
// The alignment is one wave per (query, candidate): lane l owns S consecutive DP rows, columns advance as an
// anti-diagonal wavefront, the row above arrives by a one-lane shift; every cell carries (score, matches,
// counted columns) packed into one int64 so that integer max is the lexicographic max and no traceback exists.
#include <algorithm>
#include "engine.h"
#include "k_api.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "this engine ships gfx950 code only (k_orient alone keeps 73 KB of LDS per workgroup: over the 64 KB of older targets)"
#endif

namespace itsx {

typedef long long i64;
typedef unsigned long long u64;

static constexpr int SH_S = 40, SH_M = 20;
static constexpr i64 NEGV = -(1LL << 60);
static constexpr i64 ONE_S = 1LL << SH_S, ONE_M = 1LL << SH_M;
static constexpr i64 GOI = -22 * ONE_S - 1, GEI = -2 * ONE_S - 1, GOT = -3 * ONE_S, GET = -1 * ONE_S;
// diagonal increments: match 2*ONE_S + ONE_M - 1, mismatch -4*ONE_S - 1, compatible ambiguity ONE_M - 1, incompatible -1 (built in align_pair)
static constexpr u64 MASK4_LUT = 0xFD7EB96C3A508421ULL;      // IUPAC set of each digital code, 4 bits each

__device__ __forceinline__ uint32_t mask4(uint32_t code) { return (uint32_t)(MASK4_LUT >> (4 * code)) & 15u; }
__device__ __forceinline__ uint32_t revmask4(uint32_t m) { return ((m & 1u) << 3) | ((m & 2u) << 1) | ((m & 4u) >> 1) | ((m & 8u) >> 3); }
__device__ __forceinline__ bool unamb4(uint32_t m) { return __popc(m) == 1; }
__device__ __forceinline__ i64 max64(i64 a, i64 b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t rc16(uint32_t k)
{
  uint32_t x = ~k & 0xffffu;
  x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
  x = ((x & 0x0f0fu) << 4) | ((x >> 4) & 0x0f0fu);
  return ((x & 0x00ffu) << 8) | (x >> 8);
}
__device__ __forceinline__ u64 cand_key(uint32_t cnt, int32_t len, int32_t pos)
{
  return ((u64)cnt << 48) | ((u64)(65535 - len) << 32) | (u64)(0xffffffffu - (uint32_t)pos);
}

// ------------------------------------------------------------------ identical reads inside a window share one search
// Exact duplicates have the same words, the same counts and the same walk against the same centroids, so only the
// first copy in the window (its CANONICAL query) is searched; the copies read its state.  What differs is the
// position: validation (entrants, replay) stays per query.  canon[qi] = window index of the first identical read.
static constexpr int CTAB = 16384;
__device__ __forceinline__ bool same_read(const ReadsDev &rd, int64_t x, int64_t y)
{
  const int L = rd.len[x];
  if (L != rd.len[y]) return false;
  const int64_t ex = rd.excoff[x], ey = rd.excoff[y];
  const int ne = (int)(rd.excoff[x + 1] - ex);
  if (ne != (int)(rd.excoff[y + 1] - ey)) return false;
  const uint32_t *wx = rd.words + rd.woff[x], *wy = rd.words + rd.woff[y];
  const int nw = (L + 15) >> 4;
  for (int i = 0; i < nw; i++) if (wx[i] != wy[i]) return false;
  for (int i = 0; i < ne; i++) if (rd.exc[ex + i] != rd.exc[ey + i]) return false;
  return true;
}
__global__ void k_cl_canon_insert(ClusterArgs a)
{
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= a.nq) return;
  const u64 h = a.rhash[a.order[a.f + qi]] | 1ULL;
  for (uint32_t slot = (uint32_t)(h >> 17) & (CTAB - 1);; slot = (slot + 1) & (CTAB - 1)) {
    const u64 prev = atomicCAS(&a.ctab_key[slot], 0ULL, h);
    if (prev == 0ULL || prev == h) { atomicMin(&a.ctab_val[slot], qi); break; }
  }
}
__global__ void k_cl_canon_lookup(ClusterArgs a)
{
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= a.nq) return;
  const int64_t r = a.order[a.f + qi];
  const u64 h = a.rhash[r] | 1ULL;
  uint32_t slot = (uint32_t)(h >> 17) & (CTAB - 1);
  while (a.ctab_key[slot] != h) slot = (slot + 1) & (CTAB - 1);
  const int c = a.ctab_val[slot];
  a.canon[qi] = (c < qi && same_read(a.rd, r, a.order[a.f + c])) ? c : qi;
}

// ------------------------------------------------------------------ distinct 8-mers of each (query, strand)
__global__ __launch_bounds__(256) void k_cl_kmers(ClusterArgs a)
{
  __shared__ uint32_t bm[2048];
  __shared__ uint32_t bad[2048];
  __shared__ int32_t part[256];
  const int tid = threadIdx.x;
  const int qs = blockIdx.x, qi = qs >> 1, s = qs & 1;
  if (a.canon[qi] != qi) return;
  if (s && !a.strand_both) { if (tid == 0) a.nk[qs] = 0; return; }
  const int64_t r = a.order[a.f + qi];
  const int L = a.rd.len[r];
  const uint32_t *w = a.rd.words + a.rd.woff[r];
  const int64_t eo = a.rd.excoff[r];
  const int nexc = (int)(a.rd.excoff[r + 1] - eo);
  for (int i = tid; i < 2048; i += 256) { bm[i] = 0u; bad[i] = 0u; }
  __syncthreads();
  for (int e = tid; e < nexc; e += 256) {
    const int pos = (int)(a.rd.exc[eo + e] >> 4);
    for (int d = 0; d < 8; d++) { const int p = pos - d; if (p >= 0) atomicOr(&bad[p >> 5], 1u << (p & 31)); }
  }
  __syncthreads();
  for (int i = tid; i + 8 <= L; i += 256) {
    if ((bad[i >> 5] >> (i & 31)) & 1u) continue;
    const int wi = i >> 4, sh = (i & 15) * 2;
    uint32_t k = w[wi] >> sh;
    if (sh > 16) k |= w[wi + 1] << (32 - sh);
    k &= 0xffffu;
    if (s) k = rc16(k);
    atomicOr(&bm[k >> 5], 1u << (k & 31));
  }
  __syncthreads();
  int c = 0;
  for (int j = 0; j < 8; j++) c += __popc(bm[tid * 8 + j]);
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - c;
  uint16_t *out = a.klist + (size_t)qs * a.kcap;
  for (int j = 0; j < 8; j++) {
    uint32_t x = bm[tid * 8 + j];
    while (x) { const int b = __ffs(x) - 1; out[pos++] = (uint16_t)((tid * 8 + j) * 32 + b); x &= x - 1; }
  }
  if (tid == 255) a.nk[qs] = part[255];
}

// ------------------------------------------------------------------ shared-word counts: bit-sliced carry-save counting
#define CSA(h, l, x, y, z) { const uint32_t u_ = (x) ^ (y); h = ((x) & (y)) | (u_ & (z)); l = u_ ^ (z); }
static constexpr int HCL = 13;          // counts/8 < 8192

// with_best: the launch over the old centroids also leaves each query strand's BEST candidate key in best0[] (round 0 of the
// walk then needs no pass over the count row); keys are ordered by the count first, so length and position are fetched
// only for columns that reach the lane's running maximum
__global__ __launch_bounds__(256) void k_cl_count(ClusterArgs a, int tile0, int ntiles, int with_best)
{
  // query strands vary fastest over the grid: the blocks in flight share one group of 4 tiles, whose touched rows
  // (1 KB each) then stay in L2 / MALL while every query of the window streams over them
  const int qs = blockIdx.x;
  if (a.canon[qs >> 1] != (qs >> 1)) return;
  const int lane = threadIdx.x & 63;
  const int tile = tile0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (tile >= tile0 + ntiles) return;
  const int n = a.nk[qs];
  const uint16_t *kl = a.klist + (size_t)qs * a.kcap;
  const uint32_t *col = a.bits + (size_t)tile * 64 + lane;
  const size_t stride = (size_t)a.stride;
  uint32_t ones = 0, twos = 0, fours = 0;
  uint32_t hc[HCL];
#pragma unroll
  for (int b = 0; b < HCL; b++) hc[b] = 0;
  int nlev = 1;
  while ((n >> 3) >> nlev) nlev++;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    uint32_t x[8];
#pragma unroll
    for (int t = 0; t < 8; t++) x[t] = col[(size_t)kl[i + t] * stride];
    uint32_t ta, tb, fa, fb, eights;
    CSA(ta, ones, ones, x[0], x[1])
    CSA(tb, ones, ones, x[2], x[3])
    CSA(fa, twos, twos, ta, tb)
    CSA(ta, ones, ones, x[4], x[5])
    CSA(tb, ones, ones, x[6], x[7])
    CSA(fb, twos, twos, ta, tb)
    CSA(eights, fours, fours, fa, fb)
    uint32_t carry = eights;
#pragma unroll
    for (int b = 0; b < HCL; b++) if (b < nlev) { const uint32_t t_ = hc[b] & carry; hc[b] ^= carry; carry = t_; }
  }
  for (; i < n; i++) {
    uint32_t carry = col[(size_t)kl[i] * stride], t_;
    t_ = ones & carry; ones ^= carry; carry = t_;
    t_ = twos & carry; twos ^= carry; carry = t_;
    t_ = fours & carry; fours ^= carry; carry = t_;
#pragma unroll
    for (int b = 0; b < HCL; b++) if (b < nlev) { t_ = hc[b] & carry; hc[b] ^= carry; carry = t_; }
  }
  uint16_t *out = a.cnt + (size_t)qs * a.cpitch + (size_t)tile * 2048 + lane * 32;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    uint32_t pk[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
      uint32_t v2[2];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int bit = g * 8 + h * 2 + e;
        uint32_t v = ((ones >> bit) & 1u) | (((twos >> bit) & 1u) << 1) | (((fours >> bit) & 1u) << 2);
#pragma unroll
        for (int b = 0; b < HCL; b++) if (b < nlev) v |= ((hc[b] >> bit) & 1u) << (3 + b);
        v2[e] = v;
      }
      pk[h] = v2[0] | (v2[1] << 16);
    }
    *reinterpret_cast<uint4 *>(out + g * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
  if (with_best) {
    const uint32_t minm = n < 12 ? n : 12;
    u64 best = 0;
    uint32_t bestv = minm > 0 ? minm : 1;
    const int c0 = tile * 2048 + lane * 32;
#pragma unroll
    for (int bit = 0; bit < 32; bit++) {
      uint32_t v = ((ones >> bit) & 1u) | (((twos >> bit) & 1u) << 1) | (((fours >> bit) & 1u) << 2);
#pragma unroll
      for (int b = 0; b < HCL; b++) if (b < nlev) v |= ((hc[b] >> bit) & 1u) << (3 + b);
      const int c = c0 + bit;
      if (v >= bestv && c < a.C) {
        const u64 key = cand_key(v, a.cent_len[c], a.cent_pos[c]);
        if (key > best) { best = key; bestv = v; }
      }
    }
    for (int off = 32; off; off >>= 1) { const u64 o = __shfl_xor(best, off); best = o > best ? o : best; }
    if (lane == 0 && best) atomicMax(&a.best0[qs], best);
  }
}

// ------------------------------------------------------------------ the candidate walk
__global__ void k_cl_init(ClusterArgs a)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs == 0) { for (int r = 0; r < 4; r++) a.dbg[r] = 0; a.work_n[0] = 0; a.work_n[1] = 0; }
  if (qs < a.nq) { a.replay[qs] = 0; a.skipm[qs] = 0; }
  if (qs >= 2 * a.nq) return;
  a.state[qs] = a.canon[qs >> 1] != (qs >> 1) ? 4 : (a.nk[qs] == 0 || a.C == 0) ? 3 : 0;      // 4 = a copy: reads its canonical query's state
  a.rejects[qs] = 0; a.acc_col[qs] = -1; a.wn[qs] = 0; a.selm[qs] = 0; a.sel_short[qs] = 0;
  a.prev[qs] = ~0ULL; a.bound[qs] = 0ULL; a.acc_id[qs] = -1.0; a.xn[qs] = 0; a.hard[qs] = 0;
}

// round 0 of the walk: the best key left by the counting pass is the only selected candidate
__global__ void k_cl_pick0(ClusterArgs a)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs >= 2 * a.nq || a.state[qs] != 0) return;
  const u64 key = a.best0[qs];
  if (key == 0) { a.selm[qs] = 0; a.sel_short[qs] = 1; return; }
  const int32_t pos = (int32_t)(0xffffffffu - (uint32_t)(key & 0xffffffffu));
  int lo = 0, hi = a.C - 1;                                 // cent_pos grows with the column index
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.cent_pos[mid] < pos) lo = mid + 1; else hi = mid; }
  a.sel[qs * 32] = lo; a.selkey[qs * 32] = key; a.selm[qs] = 1; a.sel_short[qs] = 0;
  a.work[atomicAdd(&a.work_n[0], 1)] = qs * 32;
}

// the next (up to kmax, within the reject budget) candidates strictly below the last one tried, in rank order.
// Every thread keeps the best key of its own strided subset of the count row; the block maximum is consumed and only
// its owner rescans (its subset, below the consumed key): one pass over the row plus need short rescans instead of
// need passes.  The (query strand, slot) items are appended to a work list for the alignment kernel.
__global__ __launch_bounds__(256) void k_cl_select(ClusterArgs a, int kmax)
{
  __shared__ u64 red[4];
  __shared__ u64 bestk;
  __shared__ int owner, ownc;
  const int qs = blockIdx.x, tid = threadIdx.x;
  if (a.state[qs] != 0) return;
  if ((qs & 1) && kmax > 1 && a.state[qs - 1] == 1 && a.acc_id[qs - 1] == 100.0) {
    // a minus-strand hit wins only with a HIGHER identity than the plus-strand hit: nothing beats 100 %, so the
    // remaining 31 candidates of this strand cannot change the outcome (k_cl_resolve cuts the window if the plus hit is lost)
    if (tid == 0) { a.state[qs] = 3; a.skipm[qs >> 1] = 1; }
    return;
  }
  const int n = a.nk[qs];
  const uint32_t minm = n < 12 ? n : 12;
  const uint16_t *cn = a.cnt + (size_t)qs * a.cpitch;
  int need = 32 - a.rejects[qs];
  need = need < kmax ? need : kmax;
  u64 lim = a.prev[qs];
  u64 best = 0; int bestc = -1;
  auto rescan = [&]() {
    best = 0; bestc = -1;
    for (int c = tid; c < a.C; c += 256) {
      const uint32_t v = cn[c];
      if (v >= minm) {
        const u64 key = cand_key(v, a.cent_len[c], a.cent_pos[c]);
        if (key < lim && key > best) { best = key; bestc = c; }
      }
    }
  };
  rescan();
  int m = 0;
  for (; m < need; m++) {
    u64 mx = best;
    for (int off = 32; off; off >>= 1) { const u64 o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      u64 b = red[0];
      for (int i = 1; i < 4; i++) b = red[i] > b ? red[i] : b;
      bestk = b;
      if (b) a.selkey[qs * 32 + m] = b;
    }
    __syncthreads();
    const u64 bk = bestk;
    if (bk == 0) break;
    if (best == bk) { a.sel[qs * 32 + m] = bestc; owner = tid; }
    __syncthreads();
    if (m + 1 >= need) { m++; break; }
    // the owner's subset (columns = owner mod 256) is rescanned by the whole block, below the consumed key
    const int ow = owner;
    u64 nb = 0; int nbc = -1;
    for (int c = ow + tid * 256; c < a.C; c += 256 * 256) {
      const uint32_t v = cn[c];
      if (v >= minm) {
        const u64 key = cand_key(v, a.cent_len[c], a.cent_pos[c]);
        if (key < bk && key > nb) { nb = key; nbc = c; }
      }
    }
    u64 mx2 = nb;
    for (int off = 32; off; off >>= 1) { const u64 o = __shfl_xor(mx2, off); mx2 = o > mx2 ? o : mx2; }
    if ((tid & 63) == 0) red[tid >> 6] = mx2;
    __syncthreads();
    u64 b2 = red[0];
    for (int i = 1; i < 4; i++) b2 = red[i] > b2 ? red[i] : b2;
    if (nb == b2 && b2 != 0) ownc = nbc;
    __syncthreads();
    if (tid == ow) { best = b2; bestc = b2 ? ownc : -1; }
    __syncthreads();
  }
  if (tid == 0) {
    a.selm[qs] = m; a.sel_short[qs] = m < need ? 1 : 0;
    if (m > 0) { const int w0 = atomicAdd(&a.work_n[0], m); for (int k = 0; k < m; k++) a.work[w0 + k] = qs * 32 + k; }
  }
}

__device__ __forceinline__ i64 shfl_up64(i64 v)
{
  int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
  lo = __shfl_up(lo, 1); hi = __shfl_up(hi, 1);
  return ((i64)hi << 32) | (i64)(uint32_t)lo;
}

// one wave = one (query strand, candidate centroid) alignment
// (lane l owns S consecutive DP rows; returns true on the lane that holds cell (Lq, Lt), with its packed value)
template <int S> __device__ __forceinline__ bool align_pair(const ClusterArgs &a, int qs, int col, int scr_slot, uint8_t *tmask, i64 &res)
{
  const int lane = threadIdx.x;
  const int qi = qs >> 1, s = qs & 1;
  const int64_t rq = a.order[a.f + qi], rt = a.cent_read[col];
  const int Lq = a.rd.len[rq], Lt = a.rd.len[rt];
  const uint32_t *wq = a.rd.words + a.rd.woff[rq];
  const uint32_t *wt = a.rd.words + a.rd.woff[rt];
  const int64_t eoq = a.rd.excoff[rq], eot = a.rd.excoff[rt];
  const int nexq = (int)(a.rd.excoff[rq + 1] - eoq), next_ = (int)(a.rd.excoff[rt + 1] - eot);
  i64 *scr = reinterpret_cast<i64 *>(a.scratch) + (size_t)scr_slot * a.scratch_pitch * 2;
  const int RB = 64 * S;
  const int npass = (Lq + 1 + RB - 1) / RB;
  res = NEGV; bool have = false;

  // the target's IUPAC sets go to LDS once; lane l then reads the symbol of its own column every step
  __syncthreads();
  for (int o = lane; o < Lt; o += 64) tmask[o] = (uint8_t)(1u << ((wt[o >> 4] >> ((o & 15) * 2)) & 3u));
  __syncthreads();
  for (int e = lane; e < next_; e += 64) { const uint32_t ex = a.rd.exc[eot + e]; tmask[ex >> 4] = (uint8_t)mask4(ex & 15u); }
  __syncthreads();

  for (int pass = 0; pass < npass; pass++) {
    const int i0 = pass * RB + lane * S;
    uint32_t qm[S]; bool qu[S]; i64 goE[S], geE[S];
#pragma unroll
    for (int r = 0; r < S; r++) {
      const int i = i0 + r;
      uint32_t m = 0;
      if (i >= 1 && i <= Lq) {
        const int x = i - 1, o = s ? Lq - 1 - x : x;
        const uint32_t c2 = (wq[o >> 4] >> ((o & 15) * 2)) & 3u;
        m = 1u << (s ? 3u - c2 : c2);
      }
      qm[r] = m; goE[r] = (i == 0 || i == Lq) ? GOT : GOI; geE[r] = (i == 0 || i == Lq) ? GET : GEI;
    }
    for (int e = 0; e < nexq; e++) {
      const uint32_t ex = a.rd.exc[eoq + e];
      const int pos = (int)(ex >> 4);
      const int i = (s ? Lq - 1 - pos : pos) + 1;
      const uint32_t m = s ? revmask4(mask4(ex & 15u)) : mask4(ex & 15u);
#pragma unroll
      for (int r = 0; r < S; r++) if (i == i0 + r) qm[r] = m;
    }
#pragma unroll
    for (int r = 0; r < S; r++) qu[r] = unamb4(qm[r]);

    i64 Hl[S], El[S];
#pragma unroll
    for (int r = 0; r < S; r++) { Hl[r] = NEGV; El[r] = NEGV; }
    // cell (0,0) is 0: its "diagonal" enters as +1 and meets the -1 of a column without symbol
    i64 diag_carry = (pass == 0 && lane == 0) ? 1 : NEGV, pubH = NEGV, pubF = NEGV;
    const int nsteps = Lt + 1 + 63;
    for (int t = 0; t < nsteps; t++) {
      i64 upH = shfl_up64(pubH), upF = shfl_up64(pubF);
      const int j = t - lane;
      if (lane == 0) {
        if (pass > 0 && j <= Lt) {
          upH = __hip_atomic_load(&scr[2 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          upF = __hip_atomic_load(&scr[2 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else { upH = NEGV; upF = NEGV; }
      }
      if (j >= 0 && j <= Lt) {
        const uint32_t tm = j >= 1 ? (uint32_t)tmask[j - 1] : 0u;
        const bool tF = (j == 0 || j == Lt);
        const i64 goF = tF ? GOT : GOI, geF = tF ? GET : GEI;
        const bool tu = unamb4(tm);
        i64 aboveH = upH, aboveF = upF, dg = diag_carry;
#pragma unroll
        for (int r = 0; r < S; r++) {
          const i64 E = max64(Hl[r] + goE[r], El[r] + geE[r]);
          const i64 F = max64(aboveH + goF, aboveF + geF);
          // D = score*2^40 + match*2^20 - 1 built from its 32-bit halves (no branches): compatible symbols are a match;
          // the score is +2 / -4 only when both symbols are unambiguous
          const bool compat = (qm[r] & tm) != 0u;
          const bool bu = qu[r] && tu;
          const uint32_t dlo = compat ? 0x000FFFFFu : 0xFFFFFFFFu;
          const int32_t dhiA = compat ? 0 : -1, dhiU = compat ? 512 : -1025;
          const i64 D = (i64)(((u64)(uint32_t)(bu ? dhiU : dhiA) << 32) | (u64)dlo);
          const i64 Hn = max64(max64(dg + D, E), F);
          dg = Hl[r];
          Hl[r] = Hn; El[r] = E;
          aboveH = Hn; aboveF = F;
        }
        pubH = aboveH; pubF = aboveF;
        diag_carry = upH;
        if (lane == 63 && pass + 1 < npass) {
          __hip_atomic_store(&scr[2 * j], aboveH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&scr[2 * j + 1], aboveF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (pass + 1 < npass) __threadfence();
    else {                                               // every row now holds its value in column Lt
#pragma unroll
      for (int r = 0; r < S; r++) if (i0 + r == Lq) { res = Hl[r]; have = true; }
    }
  }
  return have;
}
__device__ __forceinline__ double identity_of(i64 res)
{
  const i64 score = (res + (1LL << (SH_S - 1))) >> SH_S;
  const i64 low = res - score * ONE_S;
  const i64 matches = (low + (1LL << (SH_M - 1))) >> SH_M;
  const i64 cols = matches * ONE_M - low;
  return cols > 0 ? 100.0 * (double)matches / (double)cols : 0.0;
}

// ------------------------------------------------------------------ a rejection certificate before the full alignment
// 96 % of the alignments end in a rejection.  Most of them can be PROVEN rejections for a fraction of the work:
//  (1) an accepted alignment A (identity = M / (M + E) >= X, M matching columns, E mismatching pairs + interior gap
//      columns) has E <= K, the largest E with 100 min(Lq, Lt) / (min(Lq, Lt) + E) >= 100 X (evaluated as the identity is);
//  (2) every alignment scores at most 4 P - (Lq + Lt), P its aligned pairs (a pair is worth at most 2, every unpaired
//      symbol costs at least 1), and the chosen alignment is score-optimal, so with the score LB of ANY alignment in
//      hand -- here the better of "no pairs at all" and the ungapped alignment along the diagonal with most shared 8-mers --
//      an accepted alignment has P >= (LB + Lq + Lt) / 4 =: Pmin;
//  (3) so an accepted alignment is a path from the top/left border to the bottom/right border (whatever lies outside is
//      terminal gap) with at most K unit-cost edits (compatible symbols match) spanning at least Pmin rows and columns.
// Whether such a path exists is a k-differences question: furthest-reaching diagonals (Landau-Vishkin) from every border
// cell that leaves enough room, one lane per start cell.  No such path => the candidate is rejected without the O(Lq Lt)
// dynamic program; otherwise (all accepts, and the rare unlucky reject) the full alignment decides as before.
// For unrelated candidates the walk dies after one or two symbols per start; for near misses (a family member with one
// difference too many) Pmin is nearly the whole read and only a handful of start cells qualify.
// the certificate is used for edit budgets K <= 16 (identity thresholds >= ~0.95 at 300 bases; engine.hip sets pre_k)
static constexpr int PRE_LMAX = 2040;            // and reads up to this length (LDS); longer ones go straight to the full alignment

__device__ __forceinline__ uint32_t read_mask(const ReadsDev &rd, const uint32_t *w, int pos) { return 1u << ((w[pos >> 4] >> ((pos & 15) * 2)) & 3u); }

__device__ bool precheck_pair(const ClusterArgs &a, int qs, int col, uint8_t *lds)
{
  const int lane = threadIdx.x;
  const int qi = qs >> 1, s = qs & 1;
  const int64_t rq = a.order[a.f + qi], rt = a.cent_read[col];
  const int Lq = a.rd.len[rq], Lt = a.rd.len[rt];
  const int minL = Lq < Lt ? Lq : Lt;
  // K = the largest E for which ANY alignment with M <= minL matches could pass: M / (M + E) grows with M, and the test is
  // made with the very expression identity_of() evaluates, so rounding cannot open a gap between the two
  int K = 0;
  while (K <= a.pre_k && 100.0 * (double)minL / (double)(minL + K + 1) >= a.thr) K++;
  if (K > a.pre_k || Lq > PRE_LMAX || Lt > PRE_LMAX || Lq < 8 || Lt < 8) { if (lane == 0) atomicAdd(&a.pre_stats[0], 1ULL); return true; }   // no certificate: align
  uint8_t *qm = lds, *tm = qm + ((Lq + 3) & ~3);
  int32_t *head = reinterpret_cast<int32_t *>(tm + ((Lt + 3) & ~3));      // [256] chain heads of q's 8-mers (hashed)
  int32_t *nextp = head + 256;                                              // [Lq]
  uint16_t *qk = reinterpret_cast<uint16_t *>(nextp + Lq);                  // [Lq] 8-mer at each q position (0xFFFF: none)
  int32_t *votes = reinterpret_cast<int32_t *>(qk + ((Lq + 1) & ~1));       // [Lq + Lt]
  int32_t *fr = votes + Lq + Lt;                                            // [64 lanes][2][2 pre_k + 3]
  __shared__ int s_hit;
  const uint32_t *wq = a.rd.words + a.rd.woff[rq];
  const uint32_t *wt = a.rd.words + a.rd.woff[rt];
  const int64_t eoq = a.rd.excoff[rq], eot = a.rd.excoff[rt];
  const int nexq = (int)(a.rd.excoff[rq + 1] - eoq), next_ = (int)(a.rd.excoff[rt + 1] - eot);
  __syncthreads();
  for (int x = lane; x < Lq; x += 64) { const int o = s ? Lq - 1 - x : x; const uint32_t c2 = (wq[o >> 4] >> ((o & 15) * 2)) & 3u; qm[x] = (uint8_t)(1u << (s ? 3u - c2 : c2)); }
  for (int o = lane; o < Lt; o += 64) tm[o] = (uint8_t)(1u << ((wt[o >> 4] >> ((o & 15) * 2)) & 3u));
  for (int i = lane; i < 256; i += 64) head[i] = -1;
  for (int i = lane; i < Lq + Lt; i += 64) votes[i] = 0;
  if (lane == 0) s_hit = 0;
  __syncthreads();
  for (int e = lane; e < nexq; e += 64) { const uint32_t ex = a.rd.exc[eoq + e]; const int pos = (int)(ex >> 4); const uint32_t m = mask4(ex & 15u); qm[s ? Lq - 1 - pos : pos] = (uint8_t)(s ? revmask4(m) : m); }
  for (int e = lane; e < next_; e += 64) { const uint32_t ex = a.rd.exc[eot + e]; tm[ex >> 4] = (uint8_t)mask4(ex & 15u); }
  __syncthreads();
  // ---- the diagonal with most shared 8-mers (unambiguous symbols only)
  auto kmer_at = [&](const uint8_t *m, int L, int p, uint32_t &k) {
    if (p + 8 > L) return false;
    k = 0;
    for (int t = 0; t < 8; t++) { const uint32_t b = m[p + t]; if (__popc(b) != 1) return false; k |= (uint32_t)(__ffs(b) - 1) << (2 * t); }
    return true;
  };
  for (int p = lane; p < Lq; p += 64) {
    uint32_t k;
    if (kmer_at(qm, Lq, p, k)) { qk[p] = (uint16_t)k; nextp[p] = atomicExch(&head[(k * 40503u >> 8) & 255u], p); } else qk[p] = 0xFFFF;
  }
  __syncthreads();
  for (int j = lane; j < Lt; j += 64) {
    uint32_t k;
    if (!kmer_at(tm, Lt, j, k)) continue;
    for (int p = head[(k * 40503u >> 8) & 255u]; p >= 0; p = nextp[p]) if (qk[p] == (uint16_t)k) atomicAdd(&votes[p - j + Lt - 1], 1);
  }
  __syncthreads();
  int bv = -1, bd = 0;
  for (int i = lane; i < Lq + Lt - 1; i += 64) if (votes[i] > bv) { bv = votes[i]; bd = i; }
  for (int off = 32; off; off >>= 1) { const int ov = __shfl_xor(bv, off), od = __shfl_xor(bd, off); if (ov > bv || (ov == bv && od < bd)) { bv = ov; bd = od; } }
  const int d = bd - (Lt - 1);                              // q index i pairs with t index i - d
  // ---- LB: the ungapped alignment along that diagonal, its overhangs as terminal gaps
  const int i_lo = d > 0 ? d : 0, i_hi = (Lt + d < Lq) ? Lt + d : Lq;
  long long part = 0;
  for (int i = i_lo + lane; i < i_hi; i += 64) {
    const uint32_t x = qm[i], y = tm[i - d];
    if (__popc(x) == 1 && __popc(y) == 1) part += (x == y) ? 2 : -4;
  }
  for (int off = 32; off; off >>= 1) part += __shfl_xor(part, off);
  long long lb = -(long long)(Lq + Lt + 4);                 // no pairs at all: two terminal runs
  if (bv > 0 && i_hi > i_lo) {
    const int left = d > 0 ? d : -d;                        // one of the two reads overhangs on the left, one on the right
    const int right = (Lq - i_hi) + (Lt - (i_hi - d));
    const long long sc = part - (left ? 2 + left : 0) - (right ? 2 + right : 0);
    if (sc > lb) lb = sc;
  }
  const long long num = lb + Lq + Lt;
  const int Pmin = num <= 0 ? 0 : (int)((num + 3) / 4);
  if (Pmin <= K + 4) { if (lane == 0) atomicAdd(&a.pre_stats[1], 1ULL); return true; }           // too weak to exclude chance overlaps: align
  // ---- k-differences reachability from every border cell with room for Pmin rows and columns
  const int W = 2 * K + 3, Wmax = 2 * a.pre_k + 3;
  int32_t *cur = fr + lane * 2 * Wmax, *prv = cur + Wmax;
  const int nstart = Lq + Lt + 1;                           // 0..Lq: (i0, 0); Lq+1..: (0, j0 = idx - Lq)
  bool hit = false;
  for (int st = lane; st < nstart && !hit; st += 64) {
    const int i0 = st <= Lq ? st : 0, j0 = st <= Lq ? 0 : st - Lq;
    if (Lq - i0 < Pmin || Lt - j0 < Pmin) continue;
    for (int z = 0; z < W; z++) prv[z] = -1;
    for (int e = 0; e <= K && !hit; e++) {
      for (int z = 0; z < W; z++) cur[z] = -1;
      for (int dl = -e; dl <= e; dl++) {                    // diagonal (i - j) - (i0 - j0) = dl, slot dl + K + 1
        const int z = dl + K + 1;
        int i;
        if (e == 0) i = i0;
        else {
          i = -1;
          if (prv[z] >= 0) i = prv[z] + 1;                                        // mismatching pair
          if (prv[z - 1] >= 0 && prv[z - 1] + 1 > i) i = prv[z - 1] + 1;          // a q symbol against a gap
          if (prv[z + 1] >= 0 && prv[z + 1] > i) i = prv[z + 1];                  // a t symbol against a gap
          if (i < 0) continue;
        }
        int j = i - (i0 - j0) - dl;
        if (i > Lq || j > Lt || j < j0 || i < i0) continue;
        while (i < Lq && j < Lt && (qm[i] & tm[j])) { i++; j++; }
        cur[z] = i;
        if ((i == Lq || j == Lt) && i - i0 >= Pmin && j - j0 >= Pmin) { hit = true; break; }
      }
      int32_t *t_ = cur; cur = prv; prv = t_;
    }
  }
  if (hit) s_hit = 1;
  __syncthreads();
  if (lane == 0) atomicAdd(&a.pre_stats[s_hit ? 2 : 3], 1ULL);
  return s_hit != 0;
}

__global__ __launch_bounds__(64) void k_cl_precheck(ClusterArgs a, int which)
{
  extern __shared__ uint8_t pre_lds[];
  const int nw = a.work_n[which];
  const int32_t *work = which ? a.xwork : a.work;
  const int32_t *cols = which ? a.xlist : a.sel;
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {
    const int item = work[w];
    const bool need = precheck_pair(a, item >> 5, cols[item], pre_lds);
    if (threadIdx.x == 0) a.need[which * a.need_pitch + item] = need ? 1 : 0;
    __syncthreads();
  }
}

// one wave = one (query strand, selected candidate) alignment; the walk consumes the identities in rank order
template <int S> __global__ __launch_bounds__(64) void k_cl_align(ClusterArgs a)
{
  extern __shared__ uint8_t tmask_lds[];
  const int nw = a.work_n[0];
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {      // every wave drains its share of the list and exits
    const int item = a.work[w];
    if (a.need && !a.need[item]) { if (threadIdx.x == 0) { a.selpid[item] = -1.0; atomicAdd(a.n_skipped, 1ULL); } continue; }   // proven reject
    i64 res;
    if (align_pair<S>(a, item >> 5, a.sel[item], item, tmask_lds, res)) {
      a.selpid[item] = identity_of(res);
      atomicAdd(a.n_align, 1ULL);
    }
  }
}

__global__ void k_cl_walk(ClusterArgs a)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs >= 2 * a.nq || a.state[qs] != 0) return;
  const int m = a.selm[qs];
  int w = a.wn[qs], rej = a.rejects[qs], st = 0;
  for (int k = 0; k < m; k++) {
    const u64 key = a.selkey[qs * 32 + k];
    const double pid = a.selpid[qs * 32 + k];
    a.wkey[qs * 32 + w] = key; a.wpid[qs * 32 + w] = pid; a.wcol[qs * 32 + w] = a.sel[qs * 32 + k];
    w++;
    if (pid >= a.thr) { st = 1; a.acc_col[qs] = a.sel[qs * 32 + k]; a.acc_id[qs] = pid; a.bound[qs] = key; break; }
    rej++;
    if (rej >= 32) { st = 2; a.bound[qs] = key; break; }
  }
  if (st == 0) { if (a.sel_short[qs]) st = 3; else if (m > 0) a.prev[qs] = a.selkey[qs * 32 + m - 1]; }
  a.state[qs] = st; a.wn[qs] = w; a.rejects[qs] = rej; a.selm[qs] = 0;
}
__global__ void k_cl_reset_work(ClusterArgs a, int which) { a.work_n[which] = 0; }

// ------------------------------------------------------------------ outcomes, new centroids, validation
__global__ void k_cl_outcome(ClusterArgs a)
{
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= a.nq) return;
  const int c2 = 2 * a.canon[qi];
  const bool p = a.state[c2] == 1, m = a.state[c2 + 1] == 1;
  const bool hit = p || m;
  const bool minus = m && (!p || a.acc_id[c2 + 1] > a.acc_id[c2]);
  const int k = c2 + (minus ? 1 : 0);
  const int pos = a.f + qi;
  a.res_col[pos] = hit ? a.acc_col[k] : -1;
  a.res_strand[pos] = (int8_t)(hit && minus ? -1 : 1);
  a.res_id[pos] = hit ? a.acc_id[k] : -1.0;
  a.is_new[qi] = hit ? 0 : 1;
  if (qi == 0) a.is_new[a.nq] = 0;
}

// set the column of each speculative centroid (mode 0) or clear the columns flagged in rm[] (mode 1); one wave per query
__global__ __launch_bounds__(64) void k_cl_columns(ClusterArgs a, int clear)
{
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (clear ? !a.rm[qi] : !a.is_new[qi]) return;
  const int col = a.C + a.new_rank[qi];
  const int qs = 2 * a.canon[qi];
  const int n = a.nk[qs];
  const uint16_t *kl = a.klist + (size_t)qs * a.kcap;
  const uint32_t bit = 1u << (col & 31);
  uint32_t *base = a.bits + (col >> 5);
  for (int i = lane; i < n; i += 64) {
    uint32_t *p = base + (size_t)kl[i] * (size_t)a.stride;
    if (clear) atomicAnd(p, ~bit); else atomicOr(p, bit);
  }
  if (lane == 0 && !clear) {
    const int64_t r = a.order[a.f + qi];
    a.cent_len[col] = a.rd.len[r]; a.cent_pos[col] = a.f + qi; a.cent_read[col] = (int32_t)r;
    a.res_col[a.f + qi] = col;
    a.newq[a.new_rank[qi]] = qi;
  }
}

// Which speculative centroids (new in this window, before the query) would enter the query's walk?  Those with enough
// shared words and a rank above the point where the walk stopped.  They are aligned (k_cl_align_x) and k_cl_resolve
// replays the walk with them merged in; more than 32 of them cut the window at this query.
__global__ __launch_bounds__(256) void k_cl_affected(ClusterArgs a)
{
  __shared__ int xcount;
  const int qs = blockIdx.x, tid = threadIdx.x;
  const int cqi = a.canon[qs >> 1], cs = 2 * cqi + (qs & 1);      // search state lives with the canonical copy
  const int n = a.nk[cs];
  if (n == 0) return;
  const int n_new = a.new_rank[a.nq];
  if (n_new == 0) return;
  if ((qs & 1) && a.skipm[cqi]) return;                  // this strand's walk was cut short: it has no say (see k_cl_select)
  if (tid == 0) xcount = 0;
  __syncthreads();
  const int pos = a.f + (qs >> 1);
  const uint32_t minm = n < 12 ? n : 12;
  const int state = a.state[cs];
  const u64 bound = state == 3 ? 0ULL : a.bound[cs];
  const uint16_t *cn = a.cnt + (size_t)cs * a.cpitch;
  for (int c = a.C + tid; c < a.C + n_new; c += 256) {
    if (a.cent_pos[c] >= pos) break;
    const uint32_t v = cn[c];
    if (v >= minm) {
      const u64 key = cand_key(v, a.cent_len[c], a.cent_pos[c]);
      if (key > bound) {
        const int slot = atomicAdd(&xcount, 1);
        if (slot < 32) { a.xlist[qs * 32 + slot] = c; a.xkey[qs * 32 + slot] = key; }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int cnt = xcount;
    a.hard[qs] = cnt > 32; a.xn[qs] = cnt > 32 ? 0 : cnt;
    if (cnt > 32) a.replay[qs >> 1] = 1;
    else if (cnt > 0) {
      if (state == 1 && a.rejects[cs] + cnt >= 32) a.replay[qs >> 1] = 1;     // the accepted hit could fall out of the reject budget
      const int w0 = atomicAdd(&a.work_n[1], cnt);
      for (int k = 0; k < cnt; k++) a.xwork[w0 + k] = qs * 32 + k;
    }
  }
}

template <int S> __global__ __launch_bounds__(64) void k_cl_align_x(ClusterArgs a)
{
  extern __shared__ uint8_t tmask_lds[];
  const int nw = a.work_n[1];
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {
    const int item = a.xwork[w];
    if (a.need && !a.need[a.need_pitch + item]) { if (threadIdx.x == 0) { a.xpid[item] = -1.0; atomicAdd(a.n_skipped, 1ULL); } continue; }
    i64 res;
    if (align_pair<S>(a, item >> 5, a.xlist[item], item, tmask_lds, res)) {
      const double pid = identity_of(res);
      a.xpid[item] = pid;
      if (pid >= a.thr) a.replay[item >> 6] = 1;           // an entrant accepts this query: its walk must be replayed
      atomicAdd(a.n_align, 1ULL);
      atomicAdd(&a.dbg[3], 1);
    }
  }
}

// Replay, in processing order, the walk of every query that a speculative centroid ACCEPTS (or whose reject budget
// could run out; a query whose entrants all reject keeps its outcome whatever becomes of them): the recorded walk
// (rank keys and identities of the old candidates tried) is merged with the entrants that are STILL centroids;
// the first accepting element wins if fewer than 32 rejects precede it.  A query that turns from centroid into
// member is dropped from the entrants of the queries after it; a query that would turn from member into centroid
// (its accepted hit fell out of the reject budget) has no column in this window, so the window is cut there.
// One wave: lanes 0-31 hold the walk, lanes 32-63 the entrants; the merge is a rank count over 64 keys.
static constexpr int CL_MAXB = 4096;
__global__ __launch_bounds__(64) void k_cl_resolve(ClusterArgs a)
{
  __shared__ uint8_t tnew[CL_MAXB];
  __shared__ int32_t list[CL_MAXB];
  __shared__ int nlist;
  const int lane = threadIdx.x;
  const int nq = a.nq;
  for (int qi = lane; qi < nq; qi += 64) tnew[qi] = (uint8_t)a.is_new[qi];
  if (lane == 0) nlist = 0;
  __syncthreads();
  for (int base = 0; base < nq; base += 64) {
    const int qi = base + lane;
    const bool flag = qi < nq && a.replay[qi] != 0;
    const u64 mask = __ballot(flag);
    if (flag) list[nlist + __popcll(mask & ((1ULL << lane) - 1ULL))] = qi;
    __syncthreads();
    if (lane == 0) nlist += __popcll(mask);
    __syncthreads();
  }
  int cut = nq;
  const int nl = nlist;
  for (int t = 0; t < nl; t++) {
    const int qi = list[t];
    if (a.hard[2 * qi] || a.hard[2 * qi + 1]) { cut = qi; if (lane == 0) atomicAdd(&a.dbg[0], 1); break; }
    int hcol[2]; double hid[2];
    const int cqi = a.canon[qi];
    for (int s = 0; s < 2; s++) {
      const int qs = 2 * qi + s, cs = 2 * cqi + s;       // the walk was recorded for the canonical copy, the entrants are this query's
      const int wn = a.wn[cs], xn = a.xn[qs];
      u64 key = 0; double pid = -1.0; int col = -1;
      if (lane < 32) { if (lane < wn) { key = a.wkey[cs * 32 + lane]; pid = a.wpid[cs * 32 + lane]; col = a.wcol[cs * 32 + lane]; } }
      else if (lane - 32 < xn) {
        const int c = a.xlist[qs * 32 + lane - 32];
        if (tnew[a.newq[c - a.C]]) { key = a.xkey[qs * 32 + lane - 32]; pid = a.xpid[qs * 32 + lane - 32]; col = c; }
      }
      int rank = 0;
      for (int j = 0; j < 64; j++) { const u64 kj = __shfl(key, j); rank += kj > key ? 1 : 0; }
      const bool acc = key != 0 && pid >= a.thr;
      int r = acc ? rank : 1 << 20;
      for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(r, off); r = o < r ? o : r; }
      hcol[s] = -1; hid[s] = -1.0;
      if (r < 32) {
        const u64 who = __ballot(acc && rank == r);
        const int src = __ffsll((unsigned long long)who) - 1;
        hcol[s] = __shfl(col, src); hid[s] = __shfl(pid, src);
      }
    }
    if (a.skipm[cqi]) {
      if (!(hcol[0] >= 0 && hid[0] == 100.0)) { cut = qi; if (lane == 0) atomicAdd(&a.dbg[1], 1); break; }   // the 100 % plus hit is gone: search again
      hcol[1] = -1;
    }
    const bool p = hcol[0] >= 0, m = hcol[1] >= 0;
    const bool hit = p || m;
    const bool minus = m && (!p || hid[1] > hid[0]);
    if (!hit && !a.is_new[qi]) { cut = qi; if (lane == 0) atomicAdd(&a.dbg[1], 1); break; }
    if (lane == 0) {
      if (hit) {
        const int pos = a.f + qi;
        a.res_col[pos] = hcol[minus ? 1 : 0]; a.res_strand[pos] = (int8_t)(minus ? -1 : 1); a.res_id[pos] = hid[minus ? 1 : 0];
        if (tnew[qi]) atomicAdd(&a.dbg[2], 1);
      }
      tnew[qi] = hit ? 0 : 1;
    }
    __syncthreads();
  }
  int ntrue = 0;
  for (int qi = lane; qi < nq; qi += 64) {
    const int isn = a.is_new[qi];
    a.rm[qi] = (isn && (qi >= cut || !tnew[qi])) ? 1 : 0;
    ntrue += (qi < cut && tnew[qi]) ? 1 : 0;
  }
  for (int off = 32; off; off >>= 1) ntrue += __shfl_xor(ntrue, off);
  if (lane == 0) { a.wout[0] = cut; a.wout[1] = a.new_rank[cut]; a.wout[2] = ntrue; }
}

__global__ void k_cl_finalize(int32_t nk, const int32_t *order, const int32_t *res_col, const int8_t *res_strand, const double *res_id,
                              const int32_t *cent_read, int32_t *rep_of, int8_t *strand, double *pct, int32_t *is_seed)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nk) return;
  const int32_t r = order[p];
  const int32_t rep = cent_read[res_col[p]];
  rep_of[r] = rep; strand[r] = res_strand[p]; pct[r] = res_id[p]; is_seed[r] = rep == r ? 1 : 0;
}

__global__ void k_cl_relayout(const uint32_t *src, int64_t sstride, uint32_t *dst, int64_t dstride)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 65536 * sstride) return;
  dst[(i / sstride) * dstride + (i % sstride)] = src[i];
}

// ------------------------------------------------------------------ launchers
void launch_cl_kmers(const ClusterArgs &a, hipStream_t st)
{
  (void)hipMemsetAsync(a.ctab_key, 0, CTAB * sizeof(unsigned long long), st);
  (void)hipMemsetAsync(a.ctab_val, 0x7f, CTAB * sizeof(int32_t), st);
  hipLaunchKernelGGL(k_cl_canon_insert, dim3((a.nq + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_cl_canon_lookup, dim3((a.nq + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_cl_kmers, dim3(2 * a.nq), dim3(256), 0, st, a);
}
void launch_cl_count(const ClusterArgs &a, int tile0, int ntiles, int with_best, hipStream_t st)
{
  if (ntiles <= 0) return;
  hipLaunchKernelGGL(k_cl_count, dim3(2 * a.nq, (ntiles + 3) / 4), dim3(256), 0, st, a, tile0, ntiles, with_best);
}
void launch_cl_init(const ClusterArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_cl_init, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a); }
static size_t precheck_lds(const ClusterArgs &a)
{
  const size_t L = (size_t)std::min(a.scratch_pitch, PRE_LMAX + 1);      // masks, 8-mer index of the query, votes, per-lane LV rows
  return 2 * (L + 4) + 1024 + 4 * L + 2 * (L + 2) + 8 * L + 64 * 2 * (2 * (size_t)a.pre_k + 3) * 4 + 64;
}
void launch_cl_walk(const ClusterArgs &a, int rows_per_lane, hipStream_t st)
{
  // round 0: the best candidate of every query (most reads accept it); round 1: the whole remaining reject budget at once
  for (int round = 0; round < 2; round++) {
    const int kmax = round == 0 ? 1 : 31;
    if (round == 0 && a.use_best0) hipLaunchKernelGGL(k_cl_pick0, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_cl_select, dim3(2 * a.nq), dim3(256), 0, st, a, kmax);
    const int grid = std::min(2 * a.nq * kmax, 16384);
    const size_t lds = ((size_t)a.scratch_pitch + 63) & ~(size_t)63;
    if (a.need) hipLaunchKernelGGL(k_cl_precheck, dim3(grid), dim3(64), precheck_lds(a), st, a, 0);
    if (rows_per_lane <= 5) hipLaunchKernelGGL(k_cl_align<5>, dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL(k_cl_align<10>, dim3(grid), dim3(64), lds, st, a);
    hipLaunchKernelGGL(k_cl_walk, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_cl_reset_work, dim3(1), dim3(1), 0, st, a, 0);
  }
}
void launch_cl_outcome(const ClusterArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_cl_outcome, dim3((a.nq + 255) / 256), dim3(256), 0, st, a); }
void launch_cl_columns(const ClusterArgs &a, int clear, hipStream_t st) { hipLaunchKernelGGL(k_cl_columns, dim3(a.nq), dim3(64), 0, st, a, clear); }
void launch_cl_validate(const ClusterArgs &a, int rows_per_lane, hipStream_t st)
{
  hipLaunchKernelGGL(k_cl_affected, dim3(2 * a.nq), dim3(256), 0, st, a);
  const int grid = std::min(2 * a.nq * 32, 16384);
  const size_t lds = ((size_t)a.scratch_pitch + 63) & ~(size_t)63;
  if (a.need) hipLaunchKernelGGL(k_cl_precheck, dim3(grid), dim3(64), precheck_lds(a), st, a, 1);
  if (rows_per_lane <= 5) hipLaunchKernelGGL(k_cl_align_x<5>, dim3(grid), dim3(64), lds, st, a);
  else hipLaunchKernelGGL(k_cl_align_x<10>, dim3(grid), dim3(64), lds, st, a);
  hipLaunchKernelGGL(k_cl_resolve, dim3(1), dim3(64), 0, st, a);
}
void launch_cl_finalize(int32_t nk, const int32_t *order, const int32_t *res_col, const int8_t *res_strand, const double *res_id,
                        const int32_t *cent_read, int32_t *rep_of, int8_t *strand, double *pct, int32_t *is_seed, hipStream_t st)
{
  if (nk > 0) hipLaunchKernelGGL(k_cl_finalize, dim3((nk + 255) / 256), dim3(256), 0, st, nk, order, res_col, res_strand, res_id, cent_read, rep_of, strand, pct, is_seed);
}
void launch_cl_relayout(const uint32_t *src, int64_t sstride, uint32_t *dst, int64_t dstride, hipStream_t st)
{
  const int64_t n = 65536 * sstride;
  hipLaunchKernelGGL(k_cl_relayout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, sstride, dst, dstride);
}

}  // namespace itsx

// ------------------------------------------------------------------ f4: read orientation (vsearch --orient restated)
// Reference call site itsxpress/SeqSample.py:48-91.  One block per read: its distinct unambiguous 12-mers (an LDS hash
// set removes repeats) and their reverse complements are looked up in the database's 12-mer bitmap (2 MB, L2-resident);
// forward when count_fwd >= 1 and >= 4 x count_rev, reverse when the mirror holds, otherwise undetermined.
namespace itsx {
static constexpr int OTAB = 16384;
__device__ __forceinline__ uint32_t rc24(uint32_t k)
{
  uint32_t r = __brev(~k & 0xffffffu) >> 8;
  return ((r >> 1) & 0x555555u) | ((r & 0x555555u) << 1);
}
__global__ __launch_bounds__(256) void k_orient(ReadsDev rd, const uint32_t *dbbits, int8_t *strand, int32_t *cfwd, int32_t *crev)
{
  __shared__ uint32_t tab[OTAB];
  __shared__ uint32_t bad[2048];
  __shared__ int cf, cr;
  const int tid = threadIdx.x;
  for (int64_t r = blockIdx.x; r < rd.n; r += gridDim.x) {
    const int L = rd.len[r];
    const uint32_t *w = rd.words + rd.woff[r];
    const int nw = (int)(rd.woff[r + 1] - rd.woff[r]);
    const int64_t eo = rd.excoff[r];
    const int nexc = (int)(rd.excoff[r + 1] - eo);
    __syncthreads();
    for (int i = tid; i < OTAB; i += 256) tab[i] = 0u;
    for (int i = tid; i < 2048; i += 256) bad[i] = 0u;
    if (tid == 0) { cf = 0; cr = 0; }
    __syncthreads();
    for (int e = tid; e < nexc; e += 256) {
      const int pos = (int)(rd.exc[eo + e] >> 4);
      for (int d = 0; d < 12; d++) { const int p = pos - d; if (p >= 0) atomicOr(&bad[p >> 5], 1u << (p & 31)); }
    }
    __syncthreads();
    int f = 0, v = 0;
    if (L - 11 <= 12000) {
      for (int i = tid; i + 12 <= L; i += 256) {
        if ((bad[i >> 5] >> (i & 31)) & 1u) continue;
        const int wi = i >> 4, sh = (i & 15) * 2;
        const unsigned long long lo = w[wi], hi = wi + 1 < nw ? w[wi + 1] : 0u;
        const uint32_t k = (uint32_t)(((hi << 32) | lo) >> sh) & 0xffffffu;
        for (uint32_t slot = (k * 2654435761u) >> 18;; slot = (slot + 1) & (OTAB - 1)) {
          const uint32_t old = atomicCAS(&tab[slot], 0u, k + 1u);
          if (old == 0u) {                                   // first occurrence of this word in the read
            const uint32_t kr = rc24(k);
            f += (dbbits[k >> 5] >> (k & 31)) & 1u;
            v += (dbbits[kr >> 5] >> (kr & 31)) & 1u;
            break;
          }
          if (old == k + 1u) break;
        }
      }
    } else { f = 0; v = 0; }
    if (f) atomicAdd(&cf, f);
    if (v) atomicAdd(&cr, v);
    __syncthreads();
    if (tid == 0) {
      const int a = cf, b = cr;
      cfwd[r] = a; crev[r] = b;
      strand[r] = (int8_t)((a >= 1 && a >= 4 * b) ? 1 : (b >= 1 && b >= 4 * a) ? -1 : 0);
    }
  }
}
void launch_orient(const ReadsDev &rd, const uint32_t *dbbits, int8_t *strand, int32_t *cfwd, int32_t *crev, hipStream_t st)
{
  if (rd.n <= 0) return;
  hipLaunchKernelGGL(k_orient, dim3((unsigned)std::min<int64_t>(rd.n, 65536)), dim3(256), 0, st, rd, dbbits, strand, cfwd, crev);
}
}  // namespace itsx
