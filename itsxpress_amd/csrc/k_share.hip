// k_share.hip -- prefix sharing for the two kernels that scan every row of every (representative, profile) pair.
//
// hmmsearch (itsxpress/SeqSample.py:191-209) scores every target from its first residue.  The row state of the MSV filter and of
// Forward after row i depends on the profile, the target's LENGTH (the N / C / J loop cost) and the first i residues only; the
// representatives of amplicon data are error variants of far fewer templates, so 40 % of all rows repeat a row that another
// representative of the same length has computed already (scripts/prefix_sim.py).  Here the chunk's uniques of equal length are put
// into a prefix tree over blocks of B rows:
//   * node (d, prefix) = the first d * B residues some unique of that length starts with; its OWNER is the first unique (smallest
//     sorted position) that starts with them;
//   * unique s starts its CHAIN at depth[s] = the deepest node it does not own, from the state parent[s] (that node's owner) saved
//     there, and walks its own rows depth * B + 1 .. L; where a later chain starts from one of its nodes it saves the state (mask).
// The tree is found with a hash table over (length, chunk, depth, prefix) keys -- rolling XXH-style hash of the packed words, one
// insertion per (unique, depth), atomicMin keeps the owner -- and every link is then verified word by word, so a key collision can
// only cost sharing, never a wrong state.  Residues outside ACGT end a unique's tree at their block (the 2-bit plane holds A there).
// The result is bitwise the unshared kernels' (same operations on the same operands in the same order; tests/test_gpu_share.py).
#include "engine.h"
#include "k_api.h"

namespace itsx {

#define DEVI __device__ __forceinline__
static constexpr uint64_t SP1 = 11400714785074694791ULL, SP2 = 14029467366897019727ULL, SP3 = 1609587929392839161ULL,
                          SP4 = 9650029242287828579ULL, SP5 = 2870177450012600261ULL;
static constexpr unsigned long long T_EMPTY = ~0ull;
static constexpr int T_SBITS = 26;
DEVI uint64_t t_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
DEVI uint64_t t_mix(uint64_t h, uint32_t w) { h += (uint64_t)w * SP2; h = t_rotl(h, 31); return h * SP1; }
DEVI uint64_t t_fin(uint64_t h) { h ^= h >> 33; h *= SP2; h ^= h >> 29; h *= SP3; h ^= h >> 32; return h; }

struct URead { const uint32_t *w; int L, dlim; uint64_t h0; };
DEVI URead open_u(const TrieArgs &a, int s)
{
  const int r = a.seed_read[a.sorted_uniq[s]];
  URead u;
  u.w = a.rd.words + a.rd.woff[r]; u.L = a.rd.len[r];
  const int64_t eo = a.rd.excoff[r];
  const int fe = (a.rd.excoff[r + 1] > eo) ? (int)(a.rd.exc[eo] >> 4) : 0x7fffffff;      // exceptions ascend by position
  int dl = u.L > 0 ? (u.L - 1) / a.B : 0;            // a chain has at least one row of its own
  if (dl > 63) dl = 63;
  const int de = fe / a.B;                           // the first d * B residues hold no exception <=> d <= fe / B
  if (de < dl) dl = de;
  u.dlim = dl;
  u.h0 = ((uint64_t)(uint32_t)u.L * SP3) ^ ((uint64_t)(uint32_t)(s / a.Uc) * SP5) ^ SP4;
  return u;
}
DEVI void key_parts(uint64_t h, int d, uint64_t tmask, uint64_t &tag, uint64_t &slot)
{
  const uint64_t k = t_fin(h ^ ((uint64_t)d * SP5));
  tag = k >> T_SBITS;
  if (tag == (1ull << (64 - T_SBITS)) - 1) tag--;            // (tag of the empty marker)
  slot = (k * 0x9E3779B97F4A7C15ULL >> 17) & tmask;
}

__global__ void __launch_bounds__(256) k_trie_insert(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.U) return;
  const URead u = open_u(a, s);
  const int wpb = a.B >> 4;
  uint64_t h = u.h0;
  for (int d = 1; d <= u.dlim; d++) {
    for (int j = 0; j < wpb; j++) h = t_mix(h, u.w[(d - 1) * wpb + j]);
    uint64_t tag, slot;
    key_parts(h, d, a.tmask, tag, slot);
    const unsigned long long ent = (tag << T_SBITS) | (unsigned long long)(uint32_t)s;
    for (;;) {
      // (a plain load may be stale: an empty slot is then taken with a CAS that returns the truth; a slot once filled keeps its tag)
      unsigned long long cur = a.tab[slot];
      if (cur == T_EMPTY) { cur = atomicCAS(&a.tab[slot], T_EMPTY, ent); if (cur == T_EMPTY) break; }
      if ((cur >> T_SBITS) == tag) { if (ent < cur) atomicMin(&a.tab[slot], ent); break; }
      slot = (slot + 1) & a.tmask;
    }
  }
}

// deepest node of s that an earlier unique owns; the link is kept only if the two really start with the same depth * B residues
__global__ void __launch_bounds__(256) k_trie_resolve(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.U) return;
  const URead u = open_u(a, s);
  const int wpb = a.B >> 4;
  uint64_t h = u.h0;
  int dep = 0, par = -1;
  for (int d = 1; d <= u.dlim; d++) {
    for (int j = 0; j < wpb; j++) h = t_mix(h, u.w[(d - 1) * wpb + j]);
    uint64_t tag, slot;
    key_parts(h, d, a.tmask, tag, slot);
    int own = s;
    for (;;) {
      const unsigned long long cur = a.tab[slot];
      if (cur == T_EMPTY) break;                             // (cannot happen: s put this key in itself)
      if ((cur >> T_SBITS) == tag) { own = (int)(cur & ((1ull << T_SBITS) - 1)); break; }
      slot = (slot + 1) & a.tmask;
    }
    if (own != s) { dep = d; par = own; }
  }
  if (dep > 0) {
    bool ok = par >= 0 && par < s && (par / a.Uc) == (s / a.Uc);
    if (ok) {
      const URead q = open_u(a, par);
      ok = q.L == u.L && q.dlim >= dep;
      const int nw = dep * wpb;
      for (int j = 0; ok && j < nw; j++) ok = u.w[j] == q.w[j];
    }
    if (!ok) { dep = 0; par = -1; }
  }
  a.depth[s] = (uint8_t)dep; a.parent[s] = par;
}

// a parent must start above its child's branch point (it does, unless keys collided); then the parent learns where to save
__global__ void __launch_bounds__(256) k_trie_link(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.U) return;
  const int d = a.depth[s];
  if (d == 0) return;
  const int p = a.parent[s];
  // (depth[p] may be reset by its own thread meanwhile: either value it can hold is below d, or the link goes)
  if ((int)a.depth[p] >= d) { a.depth[s] = 0; a.parent[s] = -1; return; }
  atomicOr(&a.mask[p], 1ull << d);
}

__global__ void __launch_bounds__(256) k_trie_count(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  int d = 0; unsigned long long skip = 0, rows = 0, ch = 0;
  if (s < a.U) {
    a.nn[s] = __popcll(a.mask[s]);
    d = a.depth[s];
    const int r = a.seed_read[a.sorted_uniq[s]];
    rows = (unsigned long long)a.rd.len[r]; skip = (unsigned long long)d * (unsigned long long)a.B; ch = d > 0;
  }
  if (s == a.U) a.nn[s] = 0;
  for (int o = 32; o >= 1; o >>= 1) {
    const int od = __shfl_xor(d, o, 64); d = od > d ? od : d;
    skip += __shfl_xor(skip, o, 64); rows += __shfl_xor(rows, o, 64); ch += __shfl_xor(ch, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && rows) {
    atomicMax(&a.counters[0], (unsigned long long)d); atomicAdd(&a.counters[1], skip); atomicAdd(&a.counters[2], rows); atomicAdd(&a.counters[3], ch);
  }
}

// keys the table will hold at most (one per unique and depth), before it is sized
__global__ void __launch_bounds__(256) k_trie_keycount(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  if (s < a.U) n = (unsigned long long)open_u(a, s).dlim;
  for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&a.counters[4], n);
}
void launch_trie_keycount(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_keycount, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
// where a batch may start: at a new length, at a new chunk (node0_s = saved states before s); unordered, the host sorts
__global__ void __launch_bounds__(256) k_share_cuts(const int32_t *__restrict__ ulen, const int32_t *__restrict__ node0_s, int32_t U, int32_t Uc, int32_t cap,
                                                    int32_t *__restrict__ cuts, unsigned long long *__restrict__ n)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= U) return;
  if (s == 0 || ulen[s] != ulen[s - 1] || s % Uc == 0) {
    const unsigned long long i = atomicAdd(n, 1ull);
    if ((int64_t)i < cap) { cuts[2 * i] = s; cuts[2 * i + 1] = node0_s[s]; }
  }
}
void launch_share_cuts(const int32_t *ulen, const int32_t *node0_s, int32_t U, int32_t Uc, int32_t cap, int32_t *cuts, unsigned long long *n, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_share_cuts, dim3((U + 255) / 256), dim3(256), 0, st, ulen, node0_s, U, Uc, cap, cuts, n);
}
// test hooks (ITSX_SHARE_CHECK=1): cells / scores of the shared kernels that differ from the unshared kernels'
__global__ void __launch_bounds__(256) k_diff_u16(const uint16_t *__restrict__ x, const uint16_t *__restrict__ y, int64_t n, unsigned long long *__restrict__ c)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int d = (i < n) && x[i] != y[i];
  const unsigned long long m = __ballot(d);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(c, (unsigned long long)__builtin_popcountll(m));
}
__global__ void __launch_bounds__(256) k_diff_scores(const float *__restrict__ x, const float *__restrict__ y, const PairRec *__restrict__ pairs, int64_t n,
                                                     unsigned long long *__restrict__ c)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int d = 0;
  if (i < n && pairs[i].prof >= 0) d = __builtin_bit_cast(uint32_t, x[i]) != __builtin_bit_cast(uint32_t, y[i]);
  const unsigned long long m = __ballot(d);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(c, (unsigned long long)__builtin_popcountll(m));
}
void launch_diff_u16(const uint16_t *x, const uint16_t *y, int64_t n, unsigned long long *c, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_diff_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, c);
}
void launch_diff_scores(const float *x, const float *y, const PairRec *pairs, int64_t n, unsigned long long *c, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_diff_scores, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, pairs, n, c);
}

void launch_trie_insert(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_insert, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_resolve(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_resolve, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_link(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_link, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_count(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_count, dim3((a.U + 1 + 255) / 256), dim3(256), 0, st, a); }

// ---- the processing order: (batch, depth, sorted position).  A stable counting sort, one pass per depth that occurs: the flags of a
// depth are scanned (k_util.hip), a batch's chains of that depth follow its chains of the depths before (cursor).
__global__ void __launch_bounds__(256) k_share_flag(const uint8_t *__restrict__ depth, int32_t U, int d, int32_t *__restrict__ flag)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s <= U) flag[s] = (s < U && depth[s] == d) ? 1 : 0;
}
DEVI int batch_of(const int32_t *bstart, int nb, int s)
{
  int lo = 0, hi = nb;                                       // bstart[lo] <= s < bstart[hi]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bstart[mid] <= s) lo = mid; else hi = mid; }
  return lo;
}
__global__ void __launch_bounds__(256) k_share_scatter(const uint8_t *__restrict__ depth, int32_t U, int d, const int32_t *__restrict__ pos,
                                                       const int32_t *__restrict__ bstart, int nb, const int32_t *__restrict__ cursor,
                                                       int32_t *__restrict__ uorder, int32_t *__restrict__ inv)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= U || depth[s] != d) return;
  const int b = batch_of(bstart, nb, s);
  const int k = cursor[b] + pos[s] - pos[bstart[b]];
  uorder[k] = s; inv[s] = k;
}
__global__ void __launch_bounds__(256) k_share_advance(int d, const int32_t *__restrict__ pos, const int32_t *__restrict__ bstart, int nb,
                                                       int32_t *__restrict__ cursor, int32_t *__restrict__ segk)
{
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  const int c = cursor[b];
  segk[(size_t)b * SHARE_SEGS + d] = c;
  cursor[b] = c + pos[bstart[b + 1]] - pos[bstart[b]];
}
void launch_share_flag(const uint8_t *depth, int32_t U, int d, int32_t *flag, hipStream_t st)
{
  hipLaunchKernelGGL(k_share_flag, dim3((U + 1 + 255) / 256), dim3(256), 0, st, depth, U, d, flag);
}
void launch_share_scatter(const uint8_t *depth, int32_t U, int d, const int32_t *pos, const int32_t *bstart, int nb, const int32_t *cursor,
                          int32_t *uorder, int32_t *inv, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_share_scatter, dim3((U + 255) / 256), dim3(256), 0, st, depth, U, d, pos, bstart, nb, cursor, uorder, inv);
}
void launch_share_advance(int d, const int32_t *pos, const int32_t *bstart, int nb, int32_t *cursor, int32_t *segk, hipStream_t st)
{
  if (nb > 0) hipLaunchKernelGGL(k_share_advance, dim3((nb + 255) / 256), dim3(256), 0, st, d, pos, bstart, nb, cursor, segk);
}

__global__ void __launch_bounds__(256) k_share_permute(TrieArgs a, const int32_t *__restrict__ ulen_s, const int32_t *__restrict__ uorder,
                                                       const int32_t *__restrict__ inv, ShareDev o)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k > a.U) return;
  if (k == a.U) { o.nn[k] = 0; return; }
  const int s = uorder[k];
  const int d = a.depth[s];
  o.depth[k] = (uint8_t)d;
  o.parent[k] = d ? inv[a.parent[s]] - (k / a.Uc) * a.Uc : -1;       // (a batch never crosses a chunk: k and s lie in the same one)
  const unsigned long long m = a.mask[s];
  o.mask[k] = m; o.nn[k] = __popcll(m);
  o.order[k] = a.sorted_uniq[s]; o.ulen[k] = ulen_s[s];
}
__global__ void __launch_bounds__(256) k_share_src(ShareDev o, int32_t U, int32_t Uc)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= U) return;
  const int d = o.depth[k];
  int v = -1;
  if (d > 0) {
    const int par = (k / Uc) * Uc + o.parent[k];
    v = o.node0[par] + __popcll(o.mask[par] & ((1ull << d) - 1ull));
  }
  o.src[k] = v;
}
void launch_share_src(const ShareDev &o, int32_t U, int32_t Uc, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_share_src, dim3((U + 255) / 256), dim3(256), 0, st, o, U, Uc);
}
void launch_share_permute(const TrieArgs &a, const int32_t *ulen_s, const int32_t *uorder, const int32_t *inv, const ShareDev &o, hipStream_t st)
{
  hipLaunchKernelGGL(k_share_permute, dim3((a.U + 1 + 255) / 256), dim3(256), 0, st, a, ulen_s, uorder, inv, o);
}

// ---- which chains run for which profile (lazy searches: pass A only takes pairs past the MSV filter).  pass / need: one bit per
// profile, W words per chain.  A chain that failed the filter itself is still NEEDED for a profile when a chain below it passed.
__global__ void __launch_bounds__(256) k_need_bits(const uint16_t *__restrict__ res, int32_t U, int32_t P, int32_t W, uint32_t *__restrict__ pass,
                                                   uint32_t *__restrict__ need)
{
  const int k = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
  if (k >= U) return;
  uint32_t v = 0;
  for (int b = 0; b < 32; b++) { const int p = 32 * w + b; if (p < P) v |= (uint32_t)((res[(size_t)p * U + k] >> 8) & 1) << b; }
  pass[(size_t)k * W + w] = v; need[(size_t)k * W + w] = v;
}
__global__ void __launch_bounds__(256) k_need_up(int d, const uint8_t *__restrict__ depth, const int32_t *__restrict__ parent, int32_t U, int32_t W,
                                                 uint32_t *__restrict__ need)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= U || depth[k] != d) return;
  const int p = parent[k];
  for (int w = 0; w < W; w++) {
    const uint32_t v = need[(size_t)k * W + w];
    if (v & ~need[(size_t)p * W + w]) atomicOr(&need[(size_t)p * W + w], v);
  }
}
__global__ void __launch_bounds__(256) k_need_mark(uint16_t *__restrict__ res, int32_t U, int32_t P, int32_t W, const uint32_t *__restrict__ pass,
                                                   const uint32_t *__restrict__ need, unsigned long long *__restrict__ n_helpers)
{
  const int k = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
  int n = 0;
  if (k < U) {
    uint32_t h = need[(size_t)k * W + w] & ~pass[(size_t)k * W + w];
    while (h) { const int b = __builtin_ctz(h); h &= h - 1; res[(size_t)(32 * w + b) * U + k] = 0x200; n++; }
  }
  for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(n_helpers, (unsigned long long)n);
}
void launch_need_bits(const uint16_t *res, int32_t U, int32_t P, int32_t W, uint32_t *pass, uint32_t *need, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_need_bits, dim3((U + 255) / 256, W), dim3(256), 0, st, res, U, P, W, pass, need);
}
void launch_need_up(int d, const uint8_t *depth, const int32_t *parent, int32_t U, int32_t W, uint32_t *need, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_need_up, dim3((U + 255) / 256), dim3(256), 0, st, d, depth, parent, U, W, need);
}
void launch_need_mark(uint16_t *res, int32_t U, int32_t P, int32_t W, const uint32_t *pass, const uint32_t *need, unsigned long long *n_helpers, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_need_mark, dim3((U + 255) / 256, W), dim3(256), 0, st, res, U, P, W, pass, need, n_helpers);
}

// ---- the wave list of a chunk's pairs in share order.  The pairs of a profile ascend by processing position (= by batch, depth,
// length); segment t = (batch, depth) covers the positions [segk[t], segk[t + 1]).
__global__ void __launch_bounds__(256) k_share_bounds(const PairRec *__restrict__ pairs, const int64_t *__restrict__ seg_start, const int32_t *__restrict__ total,
                                                      const int32_t *__restrict__ segk, int32_t nseg, int32_t P, int64_t *__restrict__ bnd)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (nseg + 1) * P) return;
  const int t = i / P, p = i % P;
  const int32_t want = segk[t];
  const int64_t base = seg_start[p];
  int64_t lo = 0, hi = total[p];                             // first pair of the segment with useq >= want
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (pairs[base + mid].useq < want) lo = mid + 1; else hi = mid; }
  bnd[i] = base + lo;
}
__global__ void __launch_bounds__(256) k_share_wcount(const int64_t *__restrict__ bnd, int32_t nseg, int32_t P, int32_t *__restrict__ wc)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > nseg * P) return;
  wc[i] = (i < nseg * P) ? (int32_t)((bnd[i + P] - bnd[i] + 63) / 64) : 0;
}
__global__ void __launch_bounds__(256) k_share_waves(int32_t nw, int32_t nseg, int32_t P, const int32_t *__restrict__ woff, const int64_t *__restrict__ bnd,
                                                     const int32_t *__restrict__ seg_depth, int32_t B, const PairRec *__restrict__ pairs,
                                                     WaveDesc *__restrict__ w, unsigned long long *__restrict__ lane_rows)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned long long mine = 0, full = 0;
  if (i < nw) {
    int lo = 0, hi = nseg * P;                               // woff[lo] <= i < woff[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (woff[mid] <= i) lo = mid; else hi = mid; }
    const int t = lo / P, p = lo % P;
    const int64_t first = bnd[lo] + (int64_t)(i - woff[lo]) * 64;
    WaveDesc d;
    d.prof = p; d.first = first; d.count = (int32_t)min((int64_t)64, bnd[lo + P] - first); d.slab = 0; d.pad = 0;
    d.rows = pairs[first + d.count - 1].L + 1;               // ascending length inside a (profile, batch, depth) run
    w[i] = d;
    mine = (unsigned long long)(d.rows - 1 - seg_depth[t] * B) * (unsigned long long)d.count;
    full = (unsigned long long)(d.rows - 1) * (unsigned long long)d.count;
  }
  for (int o = 32; o >= 1; o >>= 1) { mine += __shfl_xor(mine, o, 64); full += __shfl_xor(full, o, 64); }
  if ((threadIdx.x & 63) == 0 && full) { atomicAdd(&lane_rows[0], mine); atomicAdd(&lane_rows[1], full); }
}
void launch_share_bounds(const PairRec *pairs, const int64_t *seg_start, const int32_t *total, const int32_t *segk, int32_t nseg, int32_t P, int64_t *bnd, hipStream_t st)
{
  hipLaunchKernelGGL(k_share_bounds, dim3(((nseg + 1) * P + 255) / 256), dim3(256), 0, st, pairs, seg_start, total, segk, nseg, P, bnd);
}
void launch_share_wcount(const int64_t *bnd, int32_t nseg, int32_t P, int32_t *wc, hipStream_t st)
{
  hipLaunchKernelGGL(k_share_wcount, dim3((nseg * P + 1 + 255) / 256), dim3(256), 0, st, bnd, nseg, P, wc);
}
void launch_share_waves(int32_t nw, int32_t nseg, int32_t P, const int32_t *woff, const int64_t *bnd, const int32_t *seg_depth, int32_t B, const PairRec *pairs,
                        WaveDesc *w, unsigned long long *lane_rows, hipStream_t st)
{
  if (nw > 0) hipLaunchKernelGGL(k_share_waves, dim3((nw + 255) / 256), dim3(256), 0, st, nw, nseg, P, woff, bnd, seg_depth, B, pairs, w, lane_rows);
}

}  // namespace itsx
