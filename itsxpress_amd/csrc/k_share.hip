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

struct URead { const uint32_t *w; int L, dlim, nw, A; uint64_t h0; };
// rev (round 6): the SUFFIX tree -- depth r = the read's last r blocks, rows (A - r) B + 1 .. L (A = ceil(L / B): the boundaries are
// the prefix tree's, the last block is the partial one); a block with a residue outside ACGT ends the tree on that side too
DEVI URead open_u(const TrieArgs &a, int s)
{
  const int r = a.seed_read[a.sorted_uniq[s]];
  URead u;
  u.w = a.rd.words + a.rd.woff[r]; u.L = a.rd.len[r];
  u.nw = (u.L + 15) >> 4; u.A = (u.L + a.B - 1) / a.B;
  const int64_t eo = a.rd.excoff[r], e1 = a.rd.excoff[r + 1];
  int dl = u.L > 0 ? (u.L - 1) / a.B : 0;            // a chain has at least one row of its own
  if (dl > 63) dl = 63;
  if (!a.rev) {
    const int fe = (e1 > eo) ? (int)(a.rd.exc[eo] >> 4) : 0x7fffffff;      // exceptions ascend by position
    const int de = fe / a.B;                         // the first d * B residues hold no exception <=> d <= fe / B
    if (de < dl) dl = de;
  } else if (e1 > eo) {
    const int le = (int)(a.rd.exc[e1 - 1] >> 4);     // rows (A - r) B + 1 .. L hold no exception <=> (A - r) B > le
    const int de = u.A - 1 - le / a.B;
    if (de < dl) dl = de;
    if (dl < 0) dl = 0;
  }
  u.dlim = dl;
  u.h0 = ((uint64_t)(uint32_t)u.L * SP3) ^ ((uint64_t)(uint32_t)(s / a.Uc) * SP5) ^ SP4 ^ (a.rev ? SP1 : 0ull);
  return u;
}
// the packed words of block d (prefix tree: rows (d - 1) B + 1 .. d B; suffix tree: the d-th block from the end) into the rolling hash
DEVI uint64_t hash_block(const TrieArgs &a, const URead &u, uint64_t h, int d)
{
  const int wpb = a.B >> 4;
  int w0 = (d - 1) * wpb, w1 = d * wpb;
  if (a.rev) { w0 = (u.A - d) * wpb; w1 = w0 + wpb; if (w1 > u.nw) w1 = u.nw; }
  for (int j = w0; j < w1; j++) h = t_mix(h, u.w[j]);
  return h;
}
// do u and q agree on the first (suffix tree: last) dep blocks?
DEVI bool same_blocks(const TrieArgs &a, const URead &u, const URead &q, int dep)
{
  const int wpb = a.B >> 4;
  int w0 = 0, w1 = dep * wpb;
  if (a.rev) { w0 = (u.A - dep) * wpb; w1 = u.nw; }
  bool ok = true;
  for (int j = w0; ok && j < w1; j++) ok = u.w[j] == q.w[j];
  return ok;
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
  uint64_t h = u.h0;
  for (int d = 1; d <= u.dlim; d++) {
    h = hash_block(a, u, h, d);
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
  uint64_t h = u.h0;
  int dep = 0, par = -1;
  for (int d = 1; d <= u.dlim; d++) {
    h = hash_block(a, u, h, d);
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
      ok = q.L == u.L && q.dlim >= dep && same_blocks(a, u, q, dep);
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
  if (!a.rev) atomicOr(&a.mask[p], 1ull << d);       // (a Backward chain saves only where somebody joins: k_join_*)
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
  __shared__ unsigned long long red[4][4];
  if ((threadIdx.x & 63) == 0) { unsigned long long *q = red[threadIdx.x >> 6]; q[0] = (unsigned long long)d; q[1] = skip; q[2] = rows; q[3] = ch; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { red[0][0] = red[w][0] > red[0][0] ? red[w][0] : red[0][0]; red[0][1] += red[w][1]; red[0][2] += red[w][2]; red[0][3] += red[w][3]; }
    if (red[0][2]) { atomicMax(&a.counters[0], red[0][0]); atomicAdd(&a.counters[1], red[0][1]); atomicAdd(&a.counters[2], red[0][2]); atomicAdd(&a.counters[3], red[0][3]); }
  }
}

// keys the table will hold at most (one per unique and depth), before it is sized
__global__ void __launch_bounds__(256) k_trie_keycount(TrieArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  if (s < a.U) n = (unsigned long long)open_u(a, s).dlim;
  for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&a.counters[a.rev ? 6 : 4], n);
}
void launch_trie_keycount(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_keycount, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
// where a batch may start: at a new length, at a new chunk (node0_s = saved states before s); unordered, the host sorts
__global__ void __launch_bounds__(256) k_share_cuts(const int32_t *__restrict__ ulen, const int32_t *__restrict__ node0_s, const int32_t *__restrict__ rnode0_s,
                                                    int32_t U, int32_t Uc, int32_t cap, int32_t *__restrict__ cuts, unsigned long long *__restrict__ n)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= U) return;
  if (s == 0 || ulen[s] != ulen[s - 1] || s % Uc == 0) {
    const unsigned long long i = atomicAdd(n, 1ull);
    if ((int64_t)i < cap) { cuts[3 * i] = s; cuts[3 * i + 1] = node0_s[s]; cuts[3 * i + 2] = rnode0_s ? rnode0_s[s] : 0; }
  }
}
void launch_share_cuts(const int32_t *ulen, const int32_t *node0_s, const int32_t *rnode0_s, int32_t U, int32_t Uc, int32_t cap, int32_t *cuts, unsigned long long *n, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_share_cuts, dim3((U + 255) / 256), dim3(256), 0, st, ulen, node0_s, rnode0_s, U, Uc, cap, cuts, n);
}

// ---- two-sided sharing (round 6).  A unique whose last r blocks some EARLIER unique of its length ends with too (suffix tree: rdepth,
// rparent) need not walk its Forward chain to L: after row j B, j = max(its Forward start, A - rdepth), the rest of the sum over paths
// is the inner product of its Forward state with the BACKWARD state gamma_j of the suffix -- the transposed recurrences run from row L
// down to row j B + 1 (k_lazy.hip: k_bwd_bound) -- and that state belongs to the suffix, not to the read: the owner of suffix node
// (A - j) computes it once for everybody.  k_join_resolve finds the level and the owner (the suffix tree's hash table is still alive:
// when j lies above the deepest shared node the node's owner is looked up and verified word by word like every link), and marks the
// owner's save mask; k_join_up makes every Backward chain that runs a saved state of ITS parent (deepest start first); k_join_ends
// turns the masks into the chains' extents: a Forward chain runs on to the deepest level a prefix child starts from, a Backward
// chain from its start down to the lowest level it saves at.
__global__ void __launch_bounds__(256) k_join_resolve(JoinArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.t.U) return;
  int jl = -1, jo = -1;
  const int rdp = a.t.depth[s];
  if (rdp > 0) {
    const URead u = open_u(a.t, s);
    const int fd = a.fdepth[s];
    const int j = max(fd, u.A - rdp), rj = u.A - j;
    int own = -1;
    if (rj == rdp) own = a.t.parent[s];                       // (verified by k_trie_resolve / k_trie_link)
    else if (rj >= 1) {
      // the Forward chain starts above the deepest shared suffix node: the owner of the (shorter) suffix at the chain's start level
      uint64_t h = u.h0;
      for (int d = 1; d <= rj; d++) h = hash_block(a.t, u, h, d);
      uint64_t tag, slot;
      key_parts(h, rj, a.t.tmask, tag, slot);
      for (;;) {
        const unsigned long long cur = a.t.tab[slot];
        if (cur == T_EMPTY) break;
        if ((cur >> T_SBITS) == tag) { own = (int)(cur & ((1ull << T_SBITS) - 1)); break; }
        slot = (slot + 1) & a.t.tmask;
      }
      bool ok = own >= 0 && own < s && (own / a.t.Uc) == (s / a.t.Uc);
      if (ok) {
        const URead q = open_u(a.t, own);
        ok = q.L == u.L && q.dlim >= rj && same_blocks(a.t, u, q, rj) && (int)a.t.depth[own] < rj;
      }
      if (!ok) own = -1;
    }
    if (own >= 0 && rj >= 1 && rj < 64) { jl = j; jo = own; atomicOr(&a.rmask[own], 1ull << rj); }
  }
  a.jlev[s] = jl; a.jown[s] = jo;
}
__global__ void __launch_bounds__(256) k_join_up(JoinArgs a, int r)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.t.U || (int)a.t.depth[s] != r) return;
  if (a.rmask[s]) atomicOr(&a.rmask[a.t.parent[s]], 1ull << r);
}
__global__ void __launch_bounds__(256) k_join_ends(JoinArgs a)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  unsigned long long nj = 0, fr = 0, br = 0, rdm = 0, nb = 0;
  if (s < a.t.U) {
    const int L = a.t.rd.len[a.t.seed_read[a.t.sorted_uniq[s]]];
    const int A = (L + a.t.B - 1) / a.t.B;
    const int fd = a.fdepth[s], jl = a.jlev[s];
    int er = L;
    if (jl >= 0) {
      const unsigned long long fm = a.fmask[s];
      const int top = fm ? 63 - __clzll((long long)fm) : 0;
      const int fe = max(jl, top);
      er = min(L, fe * a.t.B);
      nj = 1;
    }
    a.endrow[s] = er;
    fr = (unsigned long long)(er - min(er, fd * a.t.B));
    const unsigned long long rm = a.rmask[s];
    int steps = 0;
    if (rm) {
      const int rd = a.t.depth[s], rend = 63 - __clzll((long long)rm);
      steps = (rend - rd) * a.t.B;
      br = (unsigned long long)(min(L, (A - rd) * a.t.B) - (A - rend) * a.t.B);
      rdm = (unsigned long long)rd; nb = 1;
    }
    a.rsteps[s] = steps;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    nj += __shfl_xor(nj, o, 64); fr += __shfl_xor(fr, o, 64); br += __shfl_xor(br, o, 64); nb += __shfl_xor(nb, o, 64);
    const unsigned long long t = __shfl_xor(rdm, o, 64); rdm = t > rdm ? t : rdm;
  }
  // (one set of atomics per block: 23 k blocks hammering five addresses wave by wave took 5.6 ms of the 38-ms build)
  __shared__ unsigned long long red[4][5];
  if ((threadIdx.x & 63) == 0) { unsigned long long *q = red[threadIdx.x >> 6]; q[0] = nj; q[1] = fr; q[2] = br; q[3] = rdm; q[4] = nb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { red[0][0] += red[w][0]; red[0][1] += red[w][1]; red[0][2] += red[w][2]; red[0][3] = red[w][3] > red[0][3] ? red[w][3] : red[0][3]; red[0][4] += red[w][4]; }
    if (red[0][0] | red[0][1] | red[0][2]) {
      atomicAdd(&a.counters[10], red[0][0]); atomicAdd(&a.counters[11], red[0][1]); atomicAdd(&a.counters[12], red[0][2]); atomicMax(&a.counters[13], red[0][3]);
      atomicAdd(&a.counters[14], red[0][4]);
    }
  }
}
void launch_join_resolve(const JoinArgs &a, hipStream_t st) { if (a.t.U > 0) hipLaunchKernelGGL(k_join_resolve, dim3((a.t.U + 255) / 256), dim3(256), 0, st, a); }
void launch_join_up(const JoinArgs &a, int r, hipStream_t st) { if (a.t.U > 0) hipLaunchKernelGGL(k_join_up, dim3((a.t.U + 255) / 256), dim3(256), 0, st, a, r); }
void launch_join_ends(const JoinArgs &a, hipStream_t st) { if (a.t.U > 0) hipLaunchKernelGGL(k_join_ends, dim3((a.t.U + 255) / 256), dim3(256), 0, st, a); }
// test hooks (ITSX_SHARE_CHECK=1): cells / scores of the shared kernels that differ from the unshared kernels'
__global__ void __launch_bounds__(256) k_diff_u16(const uint16_t *__restrict__ x, const uint16_t *__restrict__ y, int64_t n, unsigned long long *__restrict__ c)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int d = (i < n) && x[i] != y[i];
  const unsigned long long m = __ballot(d);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(c, (unsigned long long)__builtin_popcountll(m));
}
// jlev (two-sided sharing; by the pair's useq): a JOINED pair's score is the same sum over paths in another order of operations -- it may differ
// from the unshared kernel's by rounding (more than 2e-3 nats counts as a mismatch; c[1] = the largest difference's bit pattern); every
// other pair must agree bit for bit.  Pairs that ran for their row states only (xj < 0) have no score.
__global__ void __launch_bounds__(256) k_diff_scores(const float *__restrict__ x, const float *__restrict__ y, const PairRec *__restrict__ pairs, int64_t n,
                                                     const int32_t *__restrict__ jlev, unsigned long long *__restrict__ c)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int d = 0;
  if (i < n && pairs[i].prof >= 0 && !(jlev && pairs[i].xj < 0)) {
    const float a = x[i], b = y[i];
    if (jlev && jlev[pairs[i].useq] >= 0) {
      if (a != a || b != b) d = (a != a) != (b != b);
      else {
        const float df = a > b ? a - b : b - a;
        d = !(df <= 2e-3f);
        if (df == df && df < 1e30f) atomicMax(&c[1], (unsigned long long)__builtin_bit_cast(uint32_t, df));
      }
    } else d = __builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b);
  }
  const unsigned long long m = __ballot(d);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(c, (unsigned long long)__builtin_popcountll(m));
}
__global__ void __launch_bounds__(256) k_popc64(const unsigned long long *__restrict__ m, int32_t *__restrict__ out, int64_t n, int64_t nvalid)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i < nvalid ? __popcll(m[i]) : 0;
}
void launch_popc64(const unsigned long long *m, int32_t *out, int64_t n, int64_t nvalid, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_popc64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m, out, n, nvalid);
}
void launch_diff_u16(const uint16_t *x, const uint16_t *y, int64_t n, unsigned long long *c, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_diff_u16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, n, c);
}
void launch_diff_scores(const float *x, const float *y, const PairRec *pairs, int64_t n, const int32_t *jlev, unsigned long long *c, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_diff_scores, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, pairs, n, jlev, c);
}

void launch_trie_insert(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_insert, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_resolve(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_resolve, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_link(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_link, dim3((a.U + 255) / 256), dim3(256), 0, st, a); }
void launch_trie_count(const TrieArgs &a, hipStream_t st) { if (a.U > 0) hipLaunchKernelGGL(k_trie_count, dim3((a.U + 1 + 255) / 256), dim3(256), 0, st, a); }

// ---- the processing order (batch, depth, last row, sorted position): k_order.hip
__global__ void __launch_bounds__(256) k_share_permute(TrieArgs a, const int32_t *__restrict__ ulen_s, const int32_t *__restrict__ uorder,
                                                       const int32_t *__restrict__ inv, const int32_t *__restrict__ endrow_s, const int32_t *__restrict__ jlev_s, ShareDev o)
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
  o.endrow[k] = endrow_s ? endrow_s[s] : ulen_s[s]; o.jlev[k] = jlev_s ? jlev_s[s] : -1; o.jsrc[k] = -1;
}
// the Backward chains by backward position kb (round 6): the suffix tree's links, the save masks, the chains' rows
__global__ void __launch_bounds__(256) k_bshare_permute(TrieArgs a /* rev */, int32_t Ub, const int32_t *__restrict__ ulen_s, const int32_t *__restrict__ border_s,
                                                        const int32_t *__restrict__ invb, const unsigned long long *__restrict__ rmask_s,
                                                        const int32_t *__restrict__ rsteps_s, BShareDev o)
{
  const int kb = blockIdx.x * 256 + threadIdx.x;
  if (kb > Ub) return;
  if (kb == Ub) { o.nn[kb] = 0; return; }
  const int s = border_s[kb];
  const int d = a.depth[s];
  o.depth[kb] = (uint8_t)d;
  o.parent[kb] = d ? invb[a.parent[s]] : -1;                  // (a parent of a chain that runs runs: k_join_up)
  const unsigned long long m = rmask_s[s];
  o.mask[kb] = m; o.nn[kb] = __popcll(m);
  o.order[kb] = a.sorted_uniq[s]; o.ulen[kb] = ulen_s[s]; o.steps[kb] = rsteps_s[s];
}
__global__ void __launch_bounds__(256) k_bshare_src(BShareDev o, int32_t Ub)
{
  const int kb = blockIdx.x * 256 + threadIdx.x;
  if (kb >= Ub) return;
  const int d = o.depth[kb];
  int v = -1;
  if (d > 0) { const int par = o.parent[kb]; v = par >= 0 ? o.node0[par] + __popcll(o.mask[par] & ((1ull << d) - 1ull)) : -1; }
  o.src[kb] = v;
}
// the Backward state a Forward chain joins: number (in backward order) and owner's backward position
__global__ void __launch_bounds__(256) k_join_src(int32_t U, int32_t B, const int32_t *__restrict__ uorder, const int32_t *__restrict__ jown_s, const int32_t *__restrict__ invb,
                                                  BShareDev ob, ShareDev o, int32_t *__restrict__ jownb)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= U) return;
  const int jl = o.jlev[k];
  int src = -1, ob_k = -1;
  if (jl >= 0) {
    const int own = jown_s[uorder[k]];
    ob_k = invb[own];
    if (ob_k >= 0) {
      const int A = (o.ulen[k] + B - 1) / B, rj = A - jl;
      src = ob.node0[ob_k] + __popcll(ob.mask[ob_k] & ((1ull << rj) - 1ull));
    } else o.jlev[k] = -1;                                     // (cannot happen: the owner's mask has the bit, so it runs)
  }
  o.jsrc[k] = src; jownb[k] = ob_k;
}
void launch_bshare_permute(const TrieArgs &a, int32_t Ub, const int32_t *ulen_s, const int32_t *border_s, const int32_t *invb, const unsigned long long *rmask_s,
                           const int32_t *rsteps_s, const BShareDev &o, hipStream_t st)
{
  hipLaunchKernelGGL(k_bshare_permute, dim3((Ub + 1 + 255) / 256), dim3(256), 0, st, a, Ub, ulen_s, border_s, invb, rmask_s, rsteps_s, o);
}
__global__ void __launch_bounds__(256) k_chain_recs(int32_t n, ReadsDev rd, const int32_t *__restrict__ order, const int32_t *__restrict__ seed_read, const int32_t *__restrict__ src,
                                                    const int32_t *__restrict__ node0, const unsigned long long *__restrict__ mask, const int32_t *__restrict__ endrow,
                                                    const int32_t *__restrict__ jlev, const int32_t *__restrict__ jsrc, ChainRec *__restrict__ out)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int r = seed_read[order[k]];
  ChainRec c;
  c.woff = rd.woff[r]; c.excoff = rd.excoff[r]; c.nexc = (int32_t)(rd.excoff[r + 1] - c.excoff); c.L = rd.len[r];
  c.src = src[k]; c.node0 = node0[k]; c.mask = mask[k]; c.endrow = endrow[k]; c.jlev = jlev ? jlev[k] : -1; c.jsrc = jsrc ? jsrc[k] : -1; c.pad = 0;
  out[k] = c;
}
void launch_chain_recs(int32_t n, const ReadsDev &rd, const int32_t *order, const int32_t *seed_read, const int32_t *src, const int32_t *node0, const unsigned long long *mask,
                       const int32_t *endrow, const int32_t *jlev, const int32_t *jsrc, ChainRec *out, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_chain_recs, dim3((n + 255) / 256), dim3(256), 0, st, n, rd, order, seed_read, src, node0, mask, endrow, jlev, jsrc, out);
}
void launch_bshare_src(const BShareDev &o, int32_t Ub, hipStream_t st) { if (Ub > 0) hipLaunchKernelGGL(k_bshare_src, dim3((Ub + 255) / 256), dim3(256), 0, st, o, Ub); }
void launch_join_src(int32_t U, int32_t B, const int32_t *uorder, const int32_t *jown_s, const int32_t *invb, const BShareDev &ob, const ShareDev &o, int32_t *jownb, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_join_src, dim3((U + 255) / 256), dim3(256), 0, st, U, B, uorder, jown_s, invb, ob, o, jownb);
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
void launch_share_permute(const TrieArgs &a, const int32_t *ulen_s, const int32_t *uorder, const int32_t *inv, const int32_t *endrow_s, const int32_t *jlev_s,
                          const ShareDev &o, hipStream_t st)
{
  hipLaunchKernelGGL(k_share_permute, dim3((a.U + 1 + 255) / 256), dim3(256), 0, st, a, ulen_s, uorder, inv, endrow_s, jlev_s, o);
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
// one thread per (wave, lane): a wave runs to the last row of its longest chain (forward: endrow[useq], nullptr = the pair's L;
// backward: the chain's rows), lane_rows[2 * (block & 63) + {0, 1}] += rows computed / rows of the pairs
__global__ void __launch_bounds__(256) k_share_waves(int32_t nw, int32_t nseg, int32_t P, const int32_t *__restrict__ woff, const int64_t *__restrict__ bnd,
                                                     const int32_t *__restrict__ seg_depth, int32_t B, const PairRec *__restrict__ pairs,
                                                     const int32_t *__restrict__ endrow, int backward, WaveDesc *__restrict__ w, unsigned long long *__restrict__ lane_rows)
{
  __shared__ unsigned long long sm[4][2];
  const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(gi >> 6), lane = threadIdx.x & 63;
  unsigned long long mine = 0, full = 0;
  if (i < nw) {
    int lo = 0, hi = nseg * P;                               // woff[lo] <= i < woff[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (woff[mid] <= i) lo = mid; else hi = mid; }
    const int t = lo / P, p = lo % P;
    const int64_t first = bnd[lo] + (int64_t)(i - woff[lo]) * 64;
    const int count = (int32_t)min((int64_t)64, bnd[lo + P] - first);
    int rows = 0;
    if (lane < count) {
      const PairRec pr = pairs[first + lane];
      if (backward) { rows = endrow[pr.useq]; mine = (unsigned long long)rows; }
      else {
        rows = endrow ? endrow[pr.useq] : pr.L;
        mine = (unsigned long long)max(0, rows - seg_depth[t] * B); full = (unsigned long long)pr.L;
      }
    }
    for (int o = 32; o >= 1; o >>= 1) { const int r2 = __shfl_xor(rows, o, 64); rows = r2 > rows ? r2 : rows; }
    if (lane == 0) {
      WaveDesc d;
      d.prof = p; d.first = first; d.count = count; d.slab = 0; d.pad = 0; d.rows = rows + 1;
      w[i] = d;
    }
  }
  for (int o = 32; o >= 1; o >>= 1) { mine += __shfl_xor(mine, o, 64); full += __shfl_xor(full, o, 64); }
  if (lane == 0) { sm[threadIdx.x >> 6][0] = mine; sm[threadIdx.x >> 6][1] = full; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long m = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0], f = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
    if (m) atomicAdd(&lane_rows[2 * (blockIdx.x & 63)], m);
    if (f) atomicAdd(&lane_rows[2 * (blockIdx.x & 63) + 1], f);
  }
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
                        const int32_t *endrow, int backward, WaveDesc *w, unsigned long long *lane_rows, hipStream_t st)
{
  if (nw > 0) hipLaunchKernelGGL(k_share_waves, dim3((unsigned)(((int64_t)nw * 64 + 255) / 256)), dim3(256), 0, st, nw, nseg, P, woff, bnd, seg_depth, B, pairs, endrow, backward, w, lane_rows);
}

// ---- which Backward chains run for which profile (round 6): the owner of the state a pair past the filter joins, and every chain above
// it in the suffix tree (deepest start first).  Their work list is made like the Forward pass's: res[p][kb] bit 9 -> PairRec with xj = -1
__global__ void __launch_bounds__(256) k_join_need(const int32_t *__restrict__ jlev, const int32_t *__restrict__ jownb, int32_t U, int32_t W, int32_t cb0,
                                                   const uint32_t *__restrict__ pass, uint32_t *__restrict__ need_b)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= U || jlev[k] < 0) return;
  const int o = jownb[k] - cb0;
  for (int w = 0; w < W; w++) {
    const uint32_t v = pass[(size_t)k * W + w];
    if (v & ~need_b[(size_t)o * W + w]) atomicOr(&need_b[(size_t)o * W + w], v);
  }
}
__global__ void __launch_bounds__(256) k_need_up_b(int r, const uint8_t *__restrict__ rdepth, const int32_t *__restrict__ rparent, int32_t cb0, int32_t Ub, int32_t W,
                                                   uint32_t *__restrict__ need_b)
{
  const int kb = blockIdx.x * 256 + threadIdx.x;
  if (kb >= Ub || rdepth[kb] != r) return;
  const int p = rparent[kb] - cb0;
  for (int w = 0; w < W; w++) {
    const uint32_t v = need_b[(size_t)kb * W + w];
    if (v & ~need_b[(size_t)p * W + w]) atomicOr(&need_b[(size_t)p * W + w], v);
  }
}
__global__ void __launch_bounds__(256) k_need_res(const uint32_t *__restrict__ need_b, int32_t Ub, int32_t P, int32_t W, uint16_t *__restrict__ res)
{
  const int kb = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
  if (kb >= Ub) return;
  const uint32_t v = need_b[(size_t)kb * W + w];
  for (int b = 0; b < 32; b++) { const int p = 32 * w + b; if (p < P) res[(size_t)p * Ub + kb] = ((v >> b) & 1u) ? 0x200 : 0; }
}
void launch_join_need(const int32_t *jlev, const int32_t *jownb, int32_t U, int32_t W, int32_t cb0, const uint32_t *pass, uint32_t *need_b, hipStream_t st)
{
  if (U > 0) hipLaunchKernelGGL(k_join_need, dim3((U + 255) / 256), dim3(256), 0, st, jlev, jownb, U, W, cb0, pass, need_b);
}
void launch_need_up_b(int r, const uint8_t *rdepth, const int32_t *rparent, int32_t cb0, int32_t Ub, int32_t W, uint32_t *need_b, hipStream_t st)
{
  if (Ub > 0) hipLaunchKernelGGL(k_need_up_b, dim3((Ub + 255) / 256), dim3(256), 0, st, r, rdepth, rparent, cb0, Ub, W, need_b);
}
void launch_need_res(const uint32_t *need_b, int32_t Ub, int32_t P, int32_t W, uint16_t *res, hipStream_t st)
{
  if (Ub > 0) hipLaunchKernelGGL(k_need_res, dim3((Ub + 255) / 256, W), dim3(256), 0, st, need_b, Ub, P, W, res);
}

}  // namespace itsx
