// k_derep.hip -- stage A of the path: exact full-length dereplication on the device.
//
// Replaces `vsearch --fastx_uniques IN --fastaout rep.fa --uc uc.txt --strand both`
// (reference call site itsxpress/SeqSample.py:106-116) and the uc re-parse in Dedup.parse
// (itsxpress/SeqSample.py:542-562).  Semantics (pinned by the reference fixture
// tests/test_data/ex_tmpdir/uc.txt): reads are compared full length; a read joins the
// cluster whose seed (first occurrence in input order) equals it (+) or equals its reverse
// complement (-); otherwise it seeds a new cluster.
//
// Layout in HBM: each read is a run of 32-bit words holding 16 bases each at 2 bits/base
// (A0 C1 G2 T3, base i in bits 2*(i%16)), unused high fields zero; non-ACGT symbols are 0 in
// the 2-bit plane and listed as (pos<<4 | code) exceptions.  Kernels are HBM/latency-bound
// integer work: one lane owns one read; keys are XXH64 of the packed forward strand and of
// the packed reverse complement; grouping is an open-addressing table keyed on
// min(fwd,rc) holding the smallest read index (= vsearch's "first occurrence"), followed
// by an exact word-by-word verification so that a 64-bit collision can never merge reads.
#include "engine.h"
#include "k_api.h"

namespace itsx {

static constexpr uint64_t XP1 = 11400714785074694791ULL, XP2 = 14029467366897019727ULL,
                          XP3 = 1609587929392839161ULL, XP4 = 9650029242287828579ULL,
                          XP5 = 2870177450012600261ULL;
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xround(uint64_t acc, uint64_t in) { acc += in * XP2; acc = rotl64(acc, 31); return acc * XP1; }
__device__ __forceinline__ uint64_t xmerge(uint64_t acc, uint64_t v) { acc ^= xround(0, v); return acc * XP1 + XP4; }

// complement of HMMER DNA digital codes (A C G T - R Y M K S W H B V D N)
__device__ __forceinline__ uint32_t comp_code(uint32_t c)
{
  // 0->3 1->2 2->1 3->0 4->4 5->6 6->5 7->8 8->7 9->9 10->10 11->14 12->13 13->12 14->11 15->15
  const uint64_t lut = 0xFBCDEA9785640123ULL;
  return (uint32_t)(lut >> (4 * c)) & 15u;
}

__device__ __forceinline__ uint32_t rev_fields(uint32_t w)
{
  w = __brev(w);
  return ((w >> 1) & 0x55555555u) | ((w & 0x55555555u) << 1);
}

// word j of the reverse complement of a read of L bases packed in w[0..nw)
__device__ __forceinline__ uint32_t rc_word(const uint32_t *__restrict__ w, int L, int nw, int j)
{
  const int hi = L - 1 - 16 * j;
  const int start = hi - 15;
  uint32_t W;
  if (start >= 0) {
    const int a = start >> 4, s = (start & 15) * 2;
    const uint32_t lo = w[a];
    const uint32_t up = (a + 1 < nw) ? w[a + 1] : 0u;
    W = s ? ((lo >> s) | (up << (32 - s))) : lo;
  } else {
    W = w[0] << (2 * (-start));
  }
  W = ~rev_fields(W);
  const int cnt = L - 16 * j;
  if (cnt < 16) W &= (1u << (2 * cnt)) - 1u;
  return W;
}

struct FwdKey {   // key words of the forward strand: packed words, exceptions, length
  const uint32_t *w; const uint32_t *exc; int nw, nexc, L;
  __device__ __forceinline__ uint32_t operator()(int j) const
  {
    if (j < nw) return w[j];
    j -= nw;
    if (j < nexc) return exc[j];
    return (uint32_t)L;
  }
};
struct RcKey {    // key words of the reverse complement, built on the fly
  const uint32_t *w; const uint32_t *exc; int nw, nexc, L;
  __device__ __forceinline__ uint32_t operator()(int j) const
  {
    if (j < nw) {
      uint32_t W = rc_word(w, L, nw, j);
      for (int e = 0; e < nexc; e++) {           // ambiguous positions are 0 in the 2-bit plane
        const int p = L - 1 - (int)(exc[e] >> 4);
        if ((p >> 4) == j) W &= ~(3u << (2 * (p & 15)));
      }
      return W;
    }
    j -= nw;
    if (j < nexc) {
      const uint32_t e = exc[nexc - 1 - j];
      return ((uint32_t)(L - 1 - (int)(e >> 4)) << 4) | comp_code(e & 15u);
    }
    return (uint32_t)L;
  }
};

template <class F>
__device__ __forceinline__ uint64_t xxh64_words(const F &get, int n, uint64_t seed)
{
  uint64_t h;
  int j = 0;
  auto get64 = [&](int i) { return (uint64_t)get(i) | ((uint64_t)get(i + 1) << 32); };
  if (n >= 8) {
    uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
    do {
      v1 = xround(v1, get64(j)); v2 = xround(v2, get64(j + 2));
      v3 = xround(v3, get64(j + 4)); v4 = xround(v4, get64(j + 6));
      j += 8;
    } while (j + 8 <= n);
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
  } else {
    h = seed + XP5;
  }
  h += (uint64_t)n * 4ull;
  while (j + 2 <= n) { h ^= xround(0, get64(j)); h = rotl64(h, 27) * XP1 + XP4; j += 2; }
  if (j < n) { h ^= (uint64_t)get(j) * XP1; h = rotl64(h, 23) * XP2 + XP3; }
  h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
  return h;
}

__global__ void __launch_bounds__(256) k_hash_reads(ReadsDev rd, uint64_t seed0, int strand_both,
                                                    uint64_t *__restrict__ hf, uint64_t *__restrict__ hr, const int32_t *__restrict__ sample)
{
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rd.n; r += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t seed = sample ? seed0 + (uint64_t)(uint32_t)sample[r] * XP3 : seed0;   // per-sample batching: samples never share a key
    const int L = rd.len[r];
    const int64_t wo = rd.woff[r];
    const int nw = (int)(rd.woff[r + 1] - wo);
    const int64_t eo = rd.excoff[r];
    const int nexc = (int)(rd.excoff[r + 1] - eo);
    FwdKey fk{rd.words + wo, rd.exc + eo, nw, nexc, L};
    const uint64_t f = xxh64_words(fk, nw + nexc + 1, seed);
    uint64_t c = f;
    if (strand_both) {
      RcKey rk{rd.words + wo, rd.exc + eo, nw, nexc, L};
      c = xxh64_words(rk, nw + nexc + 1, seed);
    }
    hf[r] = f; hr[r] = c;
  }
}

__global__ void __launch_bounds__(256) k_table_insert(int64_t n, const int32_t *__restrict__ len, int minlen,
                                                      const uint64_t *__restrict__ hf, const uint64_t *__restrict__ hr,
                                                      unsigned long long *__restrict__ keys, int32_t *__restrict__ vals,
                                                      uint64_t mask, uint32_t *__restrict__ slot_of)
{
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    if (len[r] < minlen) { slot_of[r] = 0xFFFFFFFFu; continue; }
    uint64_t key = hf[r] < hr[r] ? hf[r] : hr[r];
    if (key == 0) key = 1;
    uint64_t slot = (key * 0x9E3779B97F4A7C15ULL >> 20) & mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&keys[slot], 0ull, (unsigned long long)key);
      if (old == 0ull || old == key) { atomicMin(&vals[slot], (int32_t)r); slot_of[r] = (uint32_t)slot; break; }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ bool same_forward(const ReadsDev &rd, int64_t a, int64_t b)
{
  const int64_t wa = rd.woff[a], wb = rd.woff[b];
  const int nw = (int)(rd.woff[a + 1] - wa);
  const int64_t ea = rd.excoff[a], eb = rd.excoff[b];
  const int ne = (int)(rd.excoff[a + 1] - ea);
  if (rd.len[a] != rd.len[b] || ne != (int)(rd.excoff[b + 1] - eb)) return false;
  for (int j = 0; j < nw; j++) if (rd.words[wa + j] != rd.words[wb + j]) return false;
  for (int j = 0; j < ne; j++) if (rd.exc[ea + j] != rd.exc[eb + j]) return false;
  return true;
}
__device__ bool rc_equals_forward(const ReadsDev &rd, int64_t a, int64_t b)   // revcomp(a) == b ?
{
  const int64_t wa = rd.woff[a], wb = rd.woff[b];
  const int nw = (int)(rd.woff[a + 1] - wa);
  const int64_t ea = rd.excoff[a], eb = rd.excoff[b];
  const int ne = (int)(rd.excoff[a + 1] - ea);
  const int L = rd.len[a];
  if (L != rd.len[b] || ne != (int)(rd.excoff[b + 1] - eb)) return false;
  RcKey rk{rd.words + wa, rd.exc + ea, nw, ne, L};
  for (int j = 0; j < nw; j++) if (rk(j) != rd.words[wb + j]) return false;
  for (int j = 0; j < ne; j++) if (rk(nw + j) != rd.exc[eb + j]) return false;
  return true;
}

__global__ void __launch_bounds__(256) k_table_resolve(ReadsDev rd, const uint64_t *__restrict__ hf,
                                                       const int32_t *__restrict__ vals, const uint32_t *__restrict__ slot_of,
                                                       int32_t *__restrict__ rep_of, int8_t *__restrict__ strand,
                                                       int32_t *__restrict__ is_seed, unsigned int *__restrict__ n_collisions,
                                                       const int32_t *__restrict__ sample)
{
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rd.n; r += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t slot = slot_of[r];
    if (slot == 0xFFFFFFFFu) { rep_of[r] = -1; strand[r] = 0; is_seed[r] = 0; continue; }
    const int32_t s = vals[slot];
    int8_t st = 0;
    if (s == (int32_t)r) st = 1;
    else if (sample && sample[r] != sample[s]) st = 0;     // keys of two samples met: a collision like any other
    else if (hf[r] == hf[s] && same_forward(rd, r, s)) st = 1;
    else if (rc_equals_forward(rd, r, s)) st = -1;
    else if (same_forward(rd, r, s)) st = 1;
    if (st == 0) { atomicAdd(n_collisions, 1u); rep_of[r] = (int32_t)r; strand[r] = 1; is_seed[r] = 1; continue; }
    rep_of[r] = s; strand[r] = st; is_seed[r] = (s == (int32_t)r);
  }
}

// uniq_of[r] = index of r's cluster in the unique list; abundance by cluster; seed read per unique
__global__ void __launch_bounds__(256) k_uniques(int64_t n, const int32_t *__restrict__ rep_of, const int32_t *__restrict__ seed_rank,
                                                 int32_t *__restrict__ uniq_of, int32_t *__restrict__ seed_read,
                                                 int32_t *__restrict__ abundance)
{
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = rep_of[r];
    if (s < 0) { uniq_of[r] = -1; continue; }
    const int32_t u = seed_rank[s];
    uniq_of[r] = u;
    if (s == (int32_t)r) seed_read[u] = (int32_t)r;
    atomicAdd(&abundance[u], 1);
  }
}

// ---- ordering the uniques by length for the HMM stages (counting sort on length) ----
__global__ void __launch_bounds__(256) k_len_hist(int32_t U, const int32_t *__restrict__ seed_read, const int32_t *__restrict__ len,
                                                  int32_t *__restrict__ hist, int32_t lcap)
{
  // amplicon lengths cluster on a few values: one atomic per distinct length per wave, not per lane
  const int32_t stride = gridDim.x * blockDim.x;
  for (int32_t u0 = blockIdx.x * blockDim.x; u0 < U; u0 += stride) {
    const int32_t u = u0 + threadIdx.x;
    const bool ok = u < U;
    int L = ok ? len[seed_read[u]] : -1; if (L >= lcap) L = lcap - 1;
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int L0 = __shfl(L, leader, 64);
      const unsigned long long same = __ballot(ok && L == L0) & todo;
      if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[L0], __popcll(same));
      todo &= ~same;
    }
  }
}
// The same two steps for read sets whose longest read is short enough for a block to keep its own counters in LDS (every amplicon set:
// the bench's lengths spread over 281 values, so a wave met ~57 distinct ones and the kernels above fell back to nearly one global atomic
// per lane on a few hundred hot addresses: 5.4 ms each per 6 M representatives).  A block takes LEN_TILE consecutive representatives:
// counts them by length in LDS, then touches global memory once per (block, distinct length).  SCATTER: the block reserves its range of
// every length with that one atomic and ranks its members in LDS (the order inside a length is as arbitrary as above).
constexpr int LEN_TILE = 4096;
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_len_bins(int32_t U, const int32_t *__restrict__ seed_read, const int32_t *__restrict__ len,
                                                  int32_t *__restrict__ hist_or_cursor, int32_t nb, int32_t *__restrict__ sorted_uniq)
{
  extern __shared__ int32_t lds_bins[];                    // cnt[nb] (+ base[nb] when scattering)
  int32_t *cnt = lds_bins, *base = lds_bins + nb;
  for (int b = threadIdx.x; b < nb; b += 256) cnt[b] = 0;
  __syncthreads();
  const int32_t u0 = blockIdx.x * LEN_TILE, u1 = min(U, u0 + LEN_TILE);
  int32_t myL[LEN_TILE / 256];
#pragma unroll
  for (int k = 0; k < LEN_TILE / 256; k++) {
    const int32_t u = u0 + k * 256 + (int32_t)threadIdx.x;
    int L = -1;
    if (u < u1) { L = len[seed_read[u]]; if (L >= nb) L = nb - 1; atomicAdd(&cnt[L], 1); }
    myL[k] = L;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 256) {
    const int32_t c = cnt[b];
    if (c) {
      const int32_t g = atomicAdd(&hist_or_cursor[b], c);
      if (SCATTER) { base[b] = g; cnt[b] = 0; }
    }
  }
  if (!SCATTER) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < LEN_TILE / 256; k++) {
    const int L = myL[k];
    if (L >= 0) sorted_uniq[base[L] + atomicAdd(&cnt[L], 1)] = u0 + k * 256 + (int32_t)threadIdx.x;
  }
}
__global__ void __launch_bounds__(256) k_len_scatter(int32_t U, const int32_t *__restrict__ seed_read, const int32_t *__restrict__ len,
                                                     int32_t *__restrict__ cursor, int32_t lcap, int32_t *__restrict__ sorted_uniq)
{
  const int32_t stride = gridDim.x * blockDim.x;
  for (int32_t u0 = blockIdx.x * blockDim.x; u0 < U; u0 += stride) {
    const int32_t u = u0 + threadIdx.x;
    const bool ok = u < U;
    int L = ok ? len[seed_read[u]] : -1; if (L >= lcap) L = lcap - 1;
    unsigned long long todo = __ballot(ok);
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int L0 = __shfl(L, leader, 64);
      const unsigned long long same = __ballot(ok && L == L0) & todo;
      int32_t base = 0;
      if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&cursor[L0], __popcll(same));
      base = __shfl(base, leader, 64);
      if (ok && L == L0 && (same >> (threadIdx.x & 63) & 1)) sorted_uniq[base + __popcll(same & below)] = u;
      todo &= ~same;
    }
  }
}

// ---- envelope memoisation ---------------------------------------------------------------------
// Re-scoring an envelope (unihit Forward/Backward/decoding/null2) depends only on the profile, the
// target length L (length model) and the envelope's residues.  Amplicon reads share their conserved
// flanks, so many (representative, profile) pairs ask for the SAME computation: regions are keyed by
// XXH64(profile, L, Ld, residues), grouped with the same table kernels as the reads, verified exactly,
// and only the distinct ones are re-scored.  Results are bit-identical by construction.
struct RegKey {   // key words of one region: 2-bit residues re-packed from an arbitrary offset, exceptions, ids
  const uint32_t *w; const uint32_t *exc; int nexc, nw_read, off, Ld, prof, L;
  int nwd, nex_in;
  __device__ __forceinline__ uint32_t word(int j) const
  {
    const int start = off + 16 * j;
    const int a = start >> 4, s = (start & 15) * 2;
    const uint32_t lo = w[a];
    const uint32_t up = (a + 1 < nw_read) ? w[a + 1] : 0u;
    uint32_t W = s ? ((lo >> s) | (up << (32 - s))) : lo;
    const int cnt = Ld - 16 * j;
    if (cnt < 16) W &= (1u << (2 * cnt)) - 1u;
    return W;
  }
  __device__ __forceinline__ uint32_t operator()(int j) const
  {
    if (j < nwd) return word(j);
    j -= nwd;
    if (j < nex_in) {                       // the j-th exception inside [off, off+Ld)
      int seen = 0;
      for (int e = 0; e < nexc; e++) {
        const int p = (int)(exc[e] >> 4);
        if (p >= off && p < off + Ld) { if (seen == j) return ((uint32_t)(p - off) << 4) | (exc[e] & 15u); seen++; }
      }
      return 0u;
    }
    j -= nex_in;
    return j == 0 ? (uint32_t)prof : j == 1 ? (uint32_t)L : (uint32_t)Ld;
  }
};
__device__ __forceinline__ RegKey make_regkey(const ReadsDev &rd, int read, int ienv, int jenv, int prof, int L)
{
  RegKey k;
  const int64_t wo = rd.woff[read], eo = rd.excoff[read];
  k.w = rd.words + wo; k.exc = rd.exc + eo; k.nexc = (int)(rd.excoff[read + 1] - eo); k.nw_read = (int)(rd.woff[read + 1] - wo);
  k.off = ienv - 1; k.Ld = jenv - ienv + 1; k.prof = prof; k.L = L;
  k.nwd = (k.Ld + 15) >> 4;
  int n = 0;
  for (int e = 0; e < k.nexc; e++) { const int p = (int)(k.exc[e] >> 4); n += (p >= k.off && p < k.off + k.Ld); }
  k.nex_in = n;
  return k;
}

__global__ void __launch_bounds__(256) k_region_keys(ReadsDev rd, const RegionRec *__restrict__ regions, int64_t nr, const PairRec *__restrict__ pairs,
                                                     const int32_t *__restrict__ sorted_uniq, const int32_t *__restrict__ seed_read,
                                                     unsigned long long *__restrict__ keys, int32_t *__restrict__ vals, uint64_t mask,
                                                     uint32_t *__restrict__ slot_of)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nr; g += (int64_t)gridDim.x * blockDim.x) {
    const RegionRec rg = regions[g];
    if (rg.pair < 0) { slot_of[g] = 0xFFFFFFFFu; continue; }
    const PairRec pr = pairs[rg.pair];
    const RegKey rk = make_regkey(rd, seed_read[sorted_uniq[pr.useq]], rg.ienv, rg.jenv, pr.prof, pr.L);
    uint64_t key = xxh64_words(rk, rk.nwd + rk.nex_in + 3, 0x9E3779B97F4A7C15ULL);
    if (key == 0) key = 1;
    uint64_t slot = (key * 0x9E3779B97F4A7C15ULL >> 20) & mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&keys[slot], 0ull, (unsigned long long)key);
      if (old == 0ull || old == key) { atomicMin(&vals[slot], (int32_t)g); slot_of[g] = (uint32_t)slot; break; }
      slot = (slot + 1) & mask;
    }
  }
}
// rep_region[g] = smallest region index with the same key AND the same content; is_uniq[g] = (rep == g).
// A key collision between different contents just leaves the later region un-shared.
__global__ void __launch_bounds__(256) k_region_resolve(ReadsDev rd, const RegionRec *__restrict__ regions, int64_t nr, const PairRec *__restrict__ pairs,
                                                        const int32_t *__restrict__ sorted_uniq, const int32_t *__restrict__ seed_read,
                                                        const int32_t *__restrict__ vals, const uint32_t *__restrict__ slot_of,
                                                        int32_t *__restrict__ rep_region, int32_t *__restrict__ is_uniq)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nr; g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t slot = slot_of[g];
    if (slot == 0xFFFFFFFFu) { rep_region[g] = -1; is_uniq[g] = 0; continue; }
    int32_t s = vals[slot];
    if (s != (int32_t)g) {
      const RegionRec a = regions[g], b = regions[s];
      const PairRec pa = pairs[a.pair], pb = pairs[b.pair];
      bool same = pa.prof == pb.prof && pa.L == pb.L && (a.jenv - a.ienv) == (b.jenv - b.ienv);
      if (same) {
        const RegKey ka = make_regkey(rd, seed_read[sorted_uniq[pa.useq]], a.ienv, a.jenv, pa.prof, pa.L);
        const RegKey kb = make_regkey(rd, seed_read[sorted_uniq[pb.useq]], b.ienv, b.jenv, pb.prof, pb.L);
        same = ka.nex_in == kb.nex_in;
        for (int j = 0; same && j < ka.nwd + ka.nex_in; j++) same = ka(j) == kb(j);
      }
      if (!same) s = (int32_t)g;
    }
    rep_region[g] = s; is_uniq[g] = (s == (int32_t)g);
  }
}
// position of every region's result in the list of distinct regions
__global__ void __launch_bounds__(256) k_region_upos(int64_t nr, const RegionRec *__restrict__ regions, const PairRec *__restrict__ pairs,
                                                     const int32_t *__restrict__ is_uniq, const int32_t *__restrict__ urank,
                                                     const int64_t *__restrict__ rseg, const int64_t *__restrict__ useg,
                                                     int32_t *__restrict__ upos, RegionRec *__restrict__ ulist)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nr; g += (int64_t)gridDim.x * blockDim.x) {
    if (!is_uniq[g]) continue;
    const RegionRec rg = regions[g];
    const int p = pairs[rg.pair].prof;
    const int64_t pos = useg[p] + (int64_t)(urank[g] - urank[rseg[p]]);
    upos[g] = (int32_t)pos;
    ulist[pos] = rg;
  }
}
__global__ void __launch_bounds__(256) k_region_upos_follow(int64_t nr, const int32_t *__restrict__ rep_region, const int32_t *__restrict__ is_uniq,
                                                            int32_t *__restrict__ upos)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nr; g += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = rep_region[g];
    if (s >= 0 && !is_uniq[g]) upos[g] = upos[s];
  }
}

// ---- host launchers ----
static inline int grid_for(int64_t n, int block = 256, int cap = 8192)
{
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

void launch_hash_reads(const ReadsDev &rd, uint64_t seed, int strand_both, uint64_t *hf, uint64_t *hr, hipStream_t st, const int32_t *sample)
{
  hipLaunchKernelGGL(k_hash_reads, dim3(grid_for(rd.n)), dim3(256), 0, st, rd, seed, strand_both, hf, hr, sample);
}
void launch_table_insert(int64_t n, const int32_t *len, int minlen, const uint64_t *hf, const uint64_t *hr,
                         unsigned long long *keys, int32_t *vals, uint64_t mask, uint32_t *slot_of, hipStream_t st)
{
  hipLaunchKernelGGL(k_table_insert, dim3(grid_for(n)), dim3(256), 0, st, n, len, minlen, hf, hr, keys, vals, mask, slot_of);
}
void launch_table_resolve(const ReadsDev &rd, const uint64_t *hf, const int32_t *vals, const uint32_t *slot_of,
                          int32_t *rep_of, int8_t *strand, int32_t *is_seed, unsigned int *n_collisions, hipStream_t st,
                          const int32_t *sample)
{
  hipLaunchKernelGGL(k_table_resolve, dim3(grid_for(rd.n)), dim3(256), 0, st, rd, hf, vals, slot_of, rep_of, strand, is_seed, n_collisions, sample);
}
void launch_uniques(int64_t n, const int32_t *rep_of, const int32_t *seed_rank, int32_t *uniq_of, int32_t *seed_read,
                    int32_t *abundance, hipStream_t st)
{
  hipLaunchKernelGGL(k_uniques, dim3(grid_for(n)), dim3(256), 0, st, n, rep_of, seed_rank, uniq_of, seed_read, abundance);
}
// nb = longest read + 1 when the caller knows it (0: unknown): up to LEN_LDS_BINS lengths the blocks count in LDS
constexpr int LEN_LDS_BINS = 8192;
void launch_len_hist(int32_t U, const int32_t *seed_read, const int32_t *len, int32_t *hist, int32_t lcap, hipStream_t st, int32_t nb)
{
  if (nb > 0 && nb <= LEN_LDS_BINS && nb <= lcap)
    hipLaunchKernelGGL(k_len_bins<false>, dim3((unsigned)((U + LEN_TILE - 1) / LEN_TILE)), dim3(256), (size_t)nb * 4, st, U, seed_read, len, hist, nb, nullptr);
  else
    hipLaunchKernelGGL(k_len_hist, dim3(grid_for(U)), dim3(256), 0, st, U, seed_read, len, hist, lcap);
}
void launch_region_keys(const ReadsDev &rd, const RegionRec *regions, int64_t nr, const PairRec *pairs, const int32_t *sorted_uniq,
                        const int32_t *seed_read, unsigned long long *keys, int32_t *vals, uint64_t mask, uint32_t *slot_of, hipStream_t st)
{
  hipLaunchKernelGGL(k_region_keys, dim3(grid_for(nr)), dim3(256), 0, st, rd, regions, nr, pairs, sorted_uniq, seed_read, keys, vals, mask, slot_of);
}
void launch_region_resolve(const ReadsDev &rd, const RegionRec *regions, int64_t nr, const PairRec *pairs, const int32_t *sorted_uniq,
                           const int32_t *seed_read, const int32_t *vals, const uint32_t *slot_of, int32_t *rep_region, int32_t *is_uniq, hipStream_t st)
{
  hipLaunchKernelGGL(k_region_resolve, dim3(grid_for(nr)), dim3(256), 0, st, rd, regions, nr, pairs, sorted_uniq, seed_read, vals, slot_of, rep_region, is_uniq);
}
void launch_region_upos(int64_t nr, const RegionRec *regions, const PairRec *pairs, const int32_t *is_uniq, const int32_t *urank,
                        const int64_t *rseg, const int64_t *useg, int32_t *upos, RegionRec *ulist, hipStream_t st)
{
  hipLaunchKernelGGL(k_region_upos, dim3(grid_for(nr)), dim3(256), 0, st, nr, regions, pairs, is_uniq, urank, rseg, useg, upos, ulist);
}
void launch_region_upos_follow(int64_t nr, const int32_t *rep_region, const int32_t *is_uniq, int32_t *upos, hipStream_t st)
{
  hipLaunchKernelGGL(k_region_upos_follow, dim3(grid_for(nr)), dim3(256), 0, st, nr, rep_region, is_uniq, upos);
}
void launch_len_scatter(int32_t U, const int32_t *seed_read, const int32_t *len, int32_t *cursor, int32_t lcap,
                        int32_t *sorted_uniq, hipStream_t st, int32_t nb)
{
  if (nb > 0 && nb <= LEN_LDS_BINS && nb <= lcap)
    hipLaunchKernelGGL(k_len_bins<true>, dim3((unsigned)((U + LEN_TILE - 1) / LEN_TILE)), dim3(256), (size_t)nb * 8, st, U, seed_read, len, cursor, nb, sorted_uniq);
  else
    hipLaunchKernelGGL(k_len_scatter, dim3(grid_for(U)), dim3(256), 0, st, U, seed_read, len, cursor, lcap, sorted_uniq);
}

}  // namespace itsx
