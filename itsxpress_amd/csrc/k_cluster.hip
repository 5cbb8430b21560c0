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
// Shared-word counts (round 3: "query-stationary streaming"; rounds 1-2 kept a [65536 words][centroids] column bit matrix and
// a [query strands][centroids] count matrix: both passes over it were O(queries x centroids) at 44 bytes and 66 lane operations
// per (query strand, centroid), 2.4 s of a 6.1-s run at 1 M reads and quadratic from there).  Now NOTHING of size
// queries x centroids is ever stored or swept.  Per window:
//  * the window's query strands are indexed by word: word -> list of the strands that hold it (k_cl_qi_*: 65536 counters, one
//    scan, one scatter; ~3 M entries for 4096 reads, L2-resident);
//  * the centroids stream past: one workgroup takes one centroid at a time, walks the lists of the centroid's own distinct
//    words (kept per centroid, append-only) and adds into a histogram over the window's query strands in LDS (32 KB): the
//    work per (query strand, centroid) is the number of words they actually SHARE (2-3 on amplicons: the chance shares of
//    random spacer words plus the conserved motifs) instead of the query's ~380 words x 6 bit-sliced operations;
//  * a strand's count is looked at only where it can still matter: every strand carries a threshold (the rank key of its
//    32nd best candidate so far, refreshed between chunks of centroids, and that key's count for the quick test); a
//    centroid whose key beats it is appended to the strand's candidate list, everything else is dropped;
//  * k_cl_topk cuts each list back to its 32 best keys between chunks, and after the last chunk that list is the walk's
//    whole candidate list (maxaccepts 1 + maxrejects 32 never tries more), in exactly the order the count matrix gave.
// The speculative centroids of the window are streamed the same way against the window's strands (mode 2) into a small
// [strands][window] count array for the validation step.  The result is the same candidate order as before, bit for bit.
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

// ------------------------------------------------------------------ DUST soft masking of the seeds (vsearch --qmask dust / --dbmask dust)
// vsearch's default for --cluster_size and --orient: words that touch a soft-masked symbol are left out of the k-mer sets (the
// alignment sees every symbol).  mask.cc's dust() / wo() (after Tatusov & Lipman) restated (the CPU test checker states it a second time, Python a third):
// windows of 64 symbols advancing by 32; per window, for every start i (one LANE each) the running 3-mer repeat score
// 10 * sum / j over the ends j; the first best (i, j) of the window is masked when its score exceeds 20; a masked window that
// ends in its first half pulls the next one forward.  One wave per read (the windows of a read depend on each other), the
// lane's 64 3-mer counters as bytes in LDS [3-mer][lane]; anything but A C G T counts as A (it is 0 in the 2-bit plane already).
// dmask: one bit per base, the read's words at woff[r] (half of them used).
__global__ __launch_bounds__(64) void k_dust(ReadsDev rd, uint32_t *dmask)
{
  __shared__ uint8_t cnt[64 * 64];
  __shared__ uint8_t wd[64];
  __shared__ uint32_t mbits[2048];                           // reads up to 65 535 bases
  const int lane = threadIdx.x;
  for (int64_t r = blockIdx.x; r < rd.n; r += gridDim.x) {
    const int L = rd.len[r];
    const uint32_t *w = rd.words + rd.woff[r];
    const int nmw = (L + 31) >> 5;
    for (int i = lane; i < nmw; i += 64) mbits[i] = 0u;
    __syncthreads();
    for (int i = 0; i < L; i += 32) {
      const int l = (L > i + 64) ? 64 : L - i;
      // 3-mer codes of the window; the register that builds them starts empty at the window's first symbol
      if (lane < l) {
        uint32_t code = 0;
        for (int d = 2; d >= 0; d--) { const int p = lane - d; const uint32_t c2 = p >= 0 ? (w[(i + p) >> 4] >> (((i + p) & 15) * 2)) & 3u : 0u; code = (code << 2) | c2; }
        wd[lane] = (uint8_t)code;
      }
      for (int k = 0; k < 64; k++) cnt[k * 64 + lane] = 0;
      __syncthreads();
      const int l1 = l - 7;
      int bv = 0, bj = 0;
      if (lane < l1) {
        int sum = 0;
        for (int j = 2; j < l - lane; j++) {
          const int word = wd[lane + j];
          const int c = cnt[word * 64 + lane];
          if (c) { sum += c; const int v = 10 * sum / j; if (v > bv) { bv = v; bj = j; } }
          cnt[word * 64 + lane] = (uint8_t)(c + 1);
        }
      }
      // the first best in (i, j) order: the highest score, then the lowest start
      int key = (bv << 8) | (63 - lane);
      for (int off = 32; off; off >>= 1) { const int o = __shfl_xor(key, off); key = o > key ? o : key; }
      const int v = key >> 8, a = 63 - (key & 255);
      const int b = a + __shfl(bj, a);
      __syncthreads();
      if (v > 20) {
        for (int j = a + i + lane; j <= b + i; j += 64) atomicOr(&mbits[j >> 5], 1u << (j & 31));
        if (b < 32) i += 32 - b;
      }
      __syncthreads();
    }
    uint32_t *out = dmask + rd.woff[r];
    for (int i = lane; i < nmw; i += 64) out[i] = mbits[i];
    __syncthreads();
  }
}
void launch_dust(const ReadsDev &rd, uint32_t *dmask, hipStream_t st)
{
  if (rd.n > 0) hipLaunchKernelGGL(k_dust, dim3((unsigned)std::min<int64_t>(rd.n, 1 << 20)), dim3(64), 0, st, rd, dmask);
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
  if (a.dmask) {                                             // soft-masked symbols (k_dust) spoil the words that touch them, like ambiguity symbols
    const uint32_t *dm = a.dmask + a.rd.woff[r];
    for (int mw = tid; mw < ((L + 31) >> 5); mw += 256) {
      uint32_t x = dm[mw];
      while (x) {
        const int pos = mw * 32 + __ffs(x) - 1; x &= x - 1;
        for (int d = 0; d < 8; d++) { const int p = pos - d; if (p >= 0) atomicOr(&bad[p >> 5], 1u << (p & 31)); }
      }
    }
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

// ------------------------------------------------------------------ the window's query index: word -> query strands
// Where strand qs sits in k_cl_stream's LDS histogram (and in the threshold arrays tq / minm): bit-major, [qs & 31][qs >> 5].  The
// conserved-word phase has lane g of a workgroup own strands 32 g .. 32 g + 31 (one dword of every bitmap); with the strands in
// their own order its 64 lanes would write slot 32 g + bit -- two LDS banks for the whole wave, a 32-way conflict per access.
__device__ __forceinline__ int cl_phys(int qs) { return ((qs & 31) << 8) | (qs >> 5); }
__device__ __forceinline__ int cl_strand(int p) { return ((p & 255) << 5) | (p >> 8); }
static_assert(CL_QS_MAX == 8192, "cl_phys / cl_strand are written for 256 groups of 32 strands");
__global__ __launch_bounds__(256) void k_cl_qi_count(ClusterArgs a, int fill)
{
  const int qs = blockIdx.x;
  if (a.canon[qs >> 1] != (qs >> 1)) return;                // copies read their canonical query's state
  const int n = a.nk[qs];
  const uint16_t *kl = a.klist + (size_t)qs * a.kcap;
  for (int i = threadIdx.x; i < n; i += 256) {
    const uint32_t w = kl[i];
    if (!fill) atomicAdd(&a.qi_cnt[w], 1);
    else {
      const int hid = a.qi_hid[w];
      if (hid >= 0) atomicOr(&a.qi_bm[(size_t)hid * (CL_QS_MAX / 32) + (qs >> 5)], 1u << (qs & 31));      // a conserved word: its strand bitmap
      else a.qi_ent[a.qi_off[w] + atomicAdd(&a.qi_cur[w], 1)] = (uint16_t)(cl_phys(qs) * 4);     // the strand's byte offset in the LDS histogram
    }
  }
}
__global__ void k_cl_qi_pad(ClusterArgs a)
{
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < 65536) {
    int c = a.qi_cnt[w], hid = -1;
    if (c >= a.heavy_min && a.hcap > 0) { hid = atomicAdd(a.qi_nheavy, 1); if (hid >= a.hcap) hid = -1; }   // (no bitmap left: the word keeps its list)
    a.qi_hid[w] = hid;
    if (hid >= 0) c = 0;                                     // no list for a word with a bitmap
    a.qi_cnt[w] = (c + 7) & ~7; a.qi_cur[w] = 0;             // lists are read 8 entries (16 bytes) at a time
  }
  if (w == 65536) a.qi_cnt[w] = 0;
}
__global__ void k_cl_qi_fill_dummy(ClusterArgs a)
{
  const int n = a.qi_off[65536];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a.qi_ent[i] = (uint16_t)(CL_QS_MAX * 4);
  const int nh = a.qi_nheavy[0] < a.hcap ? a.qi_nheavy[0] : a.hcap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nh * (CL_QS_MAX / 32); i += gridDim.x * blockDim.x) a.qi_bm[i] = 0u;
}
// per strand: min(12, words) = the starting count threshold; the candidate lists start empty
__global__ void k_cl_tq_init(ClusterArgs a)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs == 0) a.ovf[0] = 0;
  if (qs >= 2 * a.nq) return;
  const bool own = a.canon[qs >> 1] == (qs >> 1);
  const int n = own ? a.nk[qs] : 0;
  const uint32_t m = n > 0 ? ((uint32_t)(n < 12 ? n : 12) << 16) - 1u : 0xFFFFFFFFu;      // count >= min(12, words), any length
  a.minm[cl_phys(qs)] = m; a.tq[cl_phys(qs)] = m; a.tkey[qs] = 0ULL; a.ncand[qs] = 0; a.ntop[qs] = 0;
}

// ------------------------------------------------------------------ the centroids stream past the window
// One workgroup, one centroid at a time.  items[] = pieces (<= 8 x 8 entries) of the word lists the centroid touches, so that
// a conserved word held by thousands of strands is spread over the threads instead of serialising one of them.
static constexpr int CL_ITEMS = 1024;                          // (a centroid makes ~300 pieces once the conserved words go through bitmaps)
// Eight entries (one 16-byte load) of a word's list go into the histogram.  The lists of a centroid's words are nearly the
// SAME list when the window holds reads of the centroid's own family (every one of those strands has every one of those
// words), and they are filled in nearly the same order, so lanes that walk 64 of them in step would hit ONE address per
// instruction: LDS atomics to one address are serialised, 64 deep.  Every lane therefore starts its piece at its own chunk
// (chunk order rotated by `rot & 7` in the caller) and rotates the eight entries of a chunk by `rot >> 3`.
__device__ __forceinline__ void cl_add8(uint32_t *hist, uint4 v, uint32_t r)
{
  char *h = reinterpret_cast<char *>(hist);
  if (r & 4u) { const uint32_t t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
  if (r & 2u) { const uint32_t t0 = v.x; v.x = v.y; v.y = v.z; v.z = v.w; v.w = t0; }
  const uint32_t sh = (r & 1u) * 16u;
  const uint32_t x = __builtin_amdgcn_alignbit(v.y, v.x, sh), y = __builtin_amdgcn_alignbit(v.z, v.y, sh),
                 z = __builtin_amdgcn_alignbit(v.w, v.z, sh), w = __builtin_amdgcn_alignbit(v.x, v.w, sh);
  atomicAdd(reinterpret_cast<uint32_t *>(h + (x & 0xffffu)), 1u); atomicAdd(reinterpret_cast<uint32_t *>(h + (x >> 16)), 1u);
  atomicAdd(reinterpret_cast<uint32_t *>(h + (y & 0xffffu)), 1u); atomicAdd(reinterpret_cast<uint32_t *>(h + (y >> 16)), 1u);
  atomicAdd(reinterpret_cast<uint32_t *>(h + (z & 0xffffu)), 1u); atomicAdd(reinterpret_cast<uint32_t *>(h + (z >> 16)), 1u);
  atomicAdd(reinterpret_cast<uint32_t *>(h + (w & 0xffffu)), 1u); atomicAdd(reinterpret_cast<uint32_t *>(h + (w >> 16)), 1u);
}
// a full adder over 32 strands at once is TWO instructions on gfx950: v_bitop3_b32 with the majority (0xE8) and the parity (0x96) tables
// (written with the bit operators the compiler shared x ^ y between the two and spent three or more)
#define CSA(h, l, x, y, z) { const uint32_t x_ = (x), y_ = (y), z_ = (z); h = __builtin_amdgcn_bitop3_b32(x_, y_, z_, 0xE8); l = __builtin_amdgcn_bitop3_b32(x_, y_, z_, 0x96); }
static constexpr int CL_HVL = 8;                               // bit-sliced levels above 4: counts of up to 1024 conserved words per batch
static constexpr int CL_WBATCH = 1024;                         // words of a centroid taken at a time (a read has rarely more)
__global__ __launch_bounds__(256, 4) void k_cl_stream(ClusterArgs a, int c0, int c1, int mode)
{
  __shared__ uint32_t hist[CL_QS_MAX + 8];
  __shared__ uint32_t items[CL_ITEMS];
  __shared__ __attribute__((aligned(16))) uint16_t hv[CL_WBATCH];
  __shared__ int n_items, n_hv;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nqs = 2 * a.nq;
  const int nrounds = nqs > 0 ? 8 : 0;                       // a thread scans histogram slots (j * 256 + tid) * 4 .. + 3 (cl_phys order: every round holds strands of every window size)
  // count thresholds, four strands (one 16-byte load) at a time; they only change between launches.  The arrays are
  // allocated for the largest window and the launcher sets every entry past the window's end to 0xFFFF (never reached),
  // so the scan needs no bound check.
  const uint4 *thr = reinterpret_cast<const uint4 *>(mode == 2 ? a.minm : a.tq);
  for (int i = tid; i < CL_QS_MAX + 8; i += 256) hist[i] = 0u;
  if (mode == 2) { const int lim = a.C + a.new_rank[a.nq]; c1 = c1 < lim ? c1 : lim; }
  const uint4 *ent = reinterpret_cast<const uint4 *>(a.qi_ent);
#ifdef ITSX_CL_PROF
  long long tp[6] = {0, 0, 0, 0, 0, 0}, tn = 0, tadd = 0;
#define CLK(i) { const long long now_ = (long long)__builtin_readcyclecounter(); tp[i] += now_ - tlast; tlast = now_; }
  long long tlast = (long long)__builtin_readcyclecounter();
#else
#define CLK(i)
#endif
  __syncthreads();
  for (int c = c0 + blockIdx.x; c < c1; c += gridDim.x) {
    const int n = a.cw_n[c];
    if (n == 0) continue;                                     // a rolled-back column
    const uint16_t *cw = a.cw_pool + a.cw_off[c];
    uint4 T4[8];                                              // the thread's 32 thresholds of this turn (requested below, used by the scan)
    // one batch, unless the read has more than 1024 distinct words; the LAST batch is straight-line code of its own, so that the
    // thresholds it requests are not live around a loop (they would sit beside the bitmap phase's 32 loads: spills)
    auto batch = [&](const int w0, const int nb, auto last_batch) {
      if (tid == 0) { n_items = 0; n_hv = 0; }
      __syncthreads();
      // ---- the batch's words: conserved ones to the bitmap list, the others' lists cut into pieces of <= 64 entries
      for (int i0 = 0; i0 < nb; i0 += 256) {                  // (wave-uniform trip count: the shuffles below need every lane)
        const int i = i0 + tid;
        int s = 0, nch = 0, hid = -1;
        if (i < nb) { const uint32_t w = cw[w0 + i]; hid = a.qi_hid[w]; s = a.qi_off[w] >> 3; nch = (a.qi_off[w + 1] >> 3) - s; }
        if (hid >= 0) hv[atomicAdd(&n_hv, 1)] = (uint16_t)hid;
        // slots for this word's pieces: a prefix sum over the wave and ONE atomic per wave (one per piece on one LDS address
        // serialised ~2 300 returning atomics per centroid: 55 k clock ticks, as long as the additions themselves)
        const int np = (nch + 7) >> 3;
        int incl = np;
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        int base = 0;
        if (lane == 63) base = atomicAdd(&n_items, incl);
        base = __shfl(base, 63);
        int slot = base + incl - np;
        while (nch > 0) {
          const int take = nch < 8 ? nch : 8;
          if (slot < CL_ITEMS) items[slot] = ((uint32_t)s << 3) | (uint32_t)(take - 1);
          else for (int k = 0; k < take; k++) cl_add8(hist, ent[s + k], 0u);              // list full (very long reads): the piece is added here
          s += take; nch -= take; slot++;
        }
      }
      __syncthreads();
      CLK(0)
      const int ni = n_items < CL_ITEMS ? n_items : CL_ITEMS;
#ifdef ITSX_CL_PROF
      tn++; tadd += ni;
#endif
      for (int it = tid; it < ni; it += 256) {
        const uint32_t x = items[it];
        const int s = (int)(x >> 3), take = (int)(x & 7u) + 1;
        // all (up to 8) loads of the piece are in flight before the first is used: the lists come from L2 / MALL at ~1 us a
        // round trip, and a load -> wait -> add loop pays that once per 16 bytes (measured: 75 us per centroid and workgroup)
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < take) v[k] = ent[s + k];
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < take) cl_add8(hist, v[k], 0u);
      }
      __syncthreads();
      CLK(1)
      // ---- conserved words: wave wv owns strands 2048 wv .. + 2047, a lane one dword (32 strands) of every bitmap; bit-sliced
      // carry-save counting over the batch's bitmaps, then the counts go into the lane's own 32 histogram slots (no atomics)
      const int nh = n_hv;
      const bool bitmaps = nh > 0 && wv * 2048 < nqs;
      uint32_t ones = 0, twos = 0, fours = 0, hc[CL_HVL];
#pragma unroll
      for (int b = 0; b < CL_HVL; b++) hc[b] = 0;
      if (bitmaps) {
        // (a bitmap's address = a wave-uniform base + this lane's dword: the load takes the base from scalar registers and the
        // lane's offset as it is, no address arithmetic per load)
        const uint32_t loff = (uint32_t)(wv * 64 + lane);
        const uint32_t *bmb = a.qi_bm;
        int i = 0;
        auto group8 = [&](const uint32_t *x) {                 // eight bitmaps into ones / twos / fours; what overflows is worth 8
          uint32_t ta, tb, fa, fb, eights;
          CSA(ta, ones, ones, x[0], x[1])
          CSA(tb, ones, ones, x[2], x[3])
          CSA(fa, twos, twos, ta, tb)
          CSA(ta, ones, ones, x[4], x[5])
          CSA(tb, ones, ones, x[6], x[7])
          CSA(fb, twos, twos, ta, tb)
          CSA(eights, fours, fours, fa, fb)
          return eights;
        };
        auto add8 = [&](const uint32_t *x) {
          uint32_t carry = group8(x);
#pragma unroll
          for (int b = 0; b < CL_HVL; b++) { const uint32_t t_ = hc[b] & carry; hc[b] ^= carry; carry = t_; }
        };
        // 32 bitmaps: the four groups' eights meet in two more adder levels (hc[0] holds 8s, hc[1] 16s), and only ONE carry, worth 32,
        // ripples through the upper planes: 31 adders + 12 operations instead of 28 adders + 64
        auto add32 = [&](const uint32_t *x) {
          const uint32_t e0 = group8(x), e1 = group8(x + 8), e2 = group8(x + 16), e3 = group8(x + 24);
          uint32_t sa, sb, carry;
          CSA(sa, hc[0], hc[0], e0, e1)
          CSA(sb, hc[0], hc[0], e2, e3)
          CSA(carry, hc[1], hc[1], sa, sb)
#pragma unroll
          for (int b = 2; b < CL_HVL; b++) { const uint32_t t_ = hc[b] & carry; hc[b] ^= carry; carry = t_; }
        };
        // the bitmaps come from L2 at about a microsecond a round trip: 32 loads are in flight before the first is used.  (This
        // phase WAITS: with the adders at two instructions each and, as an experiment, the loads written out with scalar bases
        // -- 207 -> 94 vector instructions per 32 words -- a 6 M-read run took as long as before, 39.3 vs 38.8 s.)
        for (; i + 32 <= nh; i += 32) {
          uint32_t x[32];
#pragma unroll
          for (int t = 0; t < 32; t++) x[t] = (bmb + (size_t)hv[i + t] * (CL_QS_MAX / 32))[loff];
          add32(x);
        }
        for (; i + 8 <= nh; i += 8) {
          uint32_t x[8];
#pragma unroll
          for (int t = 0; t < 8; t++) x[t] = (bmb + (size_t)hv[i + t] * (CL_QS_MAX / 32))[loff];
          add8(x);
        }
        for (; i < nh; i++) {
          uint32_t carry = (bmb + (size_t)hv[i] * (CL_QS_MAX / 32))[loff], t_;
          t_ = ones & carry; ones ^= carry; carry = t_;
          t_ = twos & carry; twos ^= carry; carry = t_;
          t_ = fours & carry; fours ^= carry; carry = t_;
#pragma unroll
          for (int b = 0; b < CL_HVL; b++) { t_ = hc[b] & carry; hc[b] ^= carry; carry = t_; }
        }
#ifdef ITSX_CL_PROF
        if (wv == 0) CLK(4)
#endif
      }
      // the scan's thresholds are requested HERE, eight loads together: the bitmap loads' registers are free again, and the round trip
      // runs while the counts are unpacked (requested before the bitmap phase they cost more registers than four waves have)
      if constexpr (decltype(last_batch)::value) {
        if (nrounds) {
#pragma unroll
          for (int j = 0; j < 8; j++) T4[j] = thr[j * 256 + tid];
        }
      }
      if (bitmaps) {
        // ---- the counts leave their bit planes.  Plane by plane that is 3 operations x 11 planes for every strand with a count;
        // instead the planes are first gathered four at a time into NIBBLES: for the strands whose bit index is k mod 4, nibble j of
        // n[g][k] holds planes 4g .. 4g+3 of strand bit 4j + k, so a strand's count is three bit-field extracts.  The upper planes
        // (counts of 16 and more, of 256 and more) are skipped where no lane of the wave has any.
        uint32_t any = ones | twos | fours;
        uint32_t anyB = hc[1] | hc[2] | hc[3] | hc[4], anyC = hc[5] | hc[6] | hc[7];
#pragma unroll
        for (int b = 0; b < CL_HVL; b++) any |= hc[b];
        const bool useB = __ballot(anyB != 0u) != 0ull, useC = __ballot(anyC != 0u) != 0ull;
        uint32_t *hl = hist + (wv * 64 + lane);                // slot of (strand 32 g + bit) = 256 bit + g: the wave's lanes are neighbours
        constexpr uint32_t M = 0x11111111u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t left = any & (M << k);
          if (__ballot(left != 0u) == 0ull) continue;
          const uint32_t nA = ((ones >> k) & M) | (((twos >> k) & M) << 1) | (((fours >> k) & M) << 2) | (((hc[0] >> k) & M) << 3);
          uint32_t nB = 0u, nC = 0u;
          if (useB) nB = ((hc[1] >> k) & M) | (((hc[2] >> k) & M) << 1) | (((hc[3] >> k) & M) << 2) | (((hc[4] >> k) & M) << 3);
          if (useC) nC = ((hc[5] >> k) & M) | (((hc[6] >> k) & M) << 1) | (((hc[7] >> k) & M) << 2);
          while (left) {
            const int bit = __ffs(left) - 1; left &= left - 1;
            const int sh = bit - k;                            // a multiple of 4
            uint32_t v = (nA >> sh) & 15u;
            if (useB) v |= ((nB >> sh) & 15u) << 4;
            if (useC) v |= ((nC >> sh) & 15u) << 8;
            atomicAdd(&hl[bit << 8], v);                       // (no value comes back: nothing to wait for; the slot is this lane's own)
          }
        }
      }
      __syncthreads();
      CLK(3)
    };
    {
      int w0 = 0;
      for (; w0 + CL_WBATCH < n; w0 += CL_WBATCH) batch(w0, CL_WBATCH, std::false_type{});
      batch(w0, n - w0, std::true_type{});
    }
    const int32_t clen = a.cent_len[c], cpos = a.cent_pos[c];
    const uint32_t lenpart = (uint32_t)(65535 - clen) & 0xffffu;
    // The thread's 32 thresholds come from global memory in every turn, eight loads issued together: kept in registers for the
    // launch they do not fit beside the bitmap phase's 32 loads in flight, and from scratch memory -- where the compiler put them --
    // the rolled scan paid eight DEPENDENT round trips per turn (13.6 k of its 15.7 k clock ticks).  Pass 1 only compares; the rare
    // slot that beats its threshold is handled in pass 2, slot by slot; then the thread's slots are cleared.
    if (nrounds) {
      uint32_t hm = 0u;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&hist[(j * 256 + tid) * 4]);
        hm |= ((((v.x << 16) | lenpart) > T4[j].x ? 1u : 0u) | (((v.y << 16) | lenpart) > T4[j].y ? 2u : 0u) |
               (((v.z << 16) | lenpart) > T4[j].z ? 4u : 0u) | (((v.w << 16) | lenpart) > T4[j].w ? 8u : 0u)) << (4 * j);
      }
      if (__ballot(hm != 0u) != 0ull) {
        for (uint32_t m = hm; m; m &= m - 1u) {                // the high half of the rank key beats the strand's 32nd (see k_api.h: exact)
          const int b = __ffs((int)m) - 1;
          const int slot_h = ((b >> 2) * 256 + tid) * 4 + (b & 3);
          const uint32_t cnt = hist[slot_h];
          const int qs = cl_strand(slot_h);
          if (mode == 2) a.cntx[(size_t)qs * a.xpitch + (c - a.C)] = (uint16_t)cnt;
          else {
            const int slot = atomicAdd(&a.ncand[qs], 1);
            if (slot < a.ccap) a.cand[(size_t)qs * a.ccap + slot] = cand_key(cnt, clen, cpos);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) *reinterpret_cast<uint4 *>(&hist[(j * 256 + tid) * 4]) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    CLK(2)
  }
#ifdef ITSX_CL_PROF
  if (tid == 0 && mode == 1) { for (int i = 0; i < 4; i++) atomicAdd(&a.pre_stats[8 + i], (unsigned long long)tp[i]); atomicAdd(&a.pre_stats[12], (unsigned long long)tn); atomicAdd(&a.pre_stats[13], (unsigned long long)tadd); atomicAdd(&a.pre_stats[14], (unsigned long long)tp[4]); }
#endif
}

// Between two chunks of centroids (final = 0) and after the last one (final = 1): the 32 best keys of the strand's list, in
// rank order, replace the list; the 32nd becomes the strand's threshold (key, and its count for the quick test).  After the
// last chunk that list IS the candidate list of the whole walk (maxaccepts 1 + maxrejects 32 never tries more): its columns
// are looked up by position.  One wave per strand.
__global__ __launch_bounds__(64) void k_cl_topk(ClusterArgs a, int final)
{
  __shared__ unsigned long long top[32];
  const int qs = blockIdx.x, lane = threadIdx.x;
  int n = a.ncand[qs];
  if (n > a.ccap) { if (lane == 0) a.ovf[0] = 1; n = a.ccap; }
  if (!final && n <= 32) return;                              // nothing to cut yet
  unsigned long long *keys = a.cand + (size_t)qs * a.ccap;
  unsigned long long lim = ~0ULL;
  int m = 0;
  for (; m < 32 && m < n; m++) {
    unsigned long long best = 0;
    for (int i = lane; i < n; i += 64) { const unsigned long long k = keys[i]; if (k < lim && k > best) best = k; }
    for (int off = 32; off; off >>= 1) { const unsigned long long o = __shfl_xor(best, off); best = o > best ? o : best; }
    if (best == 0ULL) break;
    if (lane == 0) top[m] = best;
    lim = best;
  }
  __syncthreads();
  if (lane < m) keys[lane] = top[lane];
  if (lane == 0) {
    a.ncand[qs] = m;
    if (m == 32) { a.tkey[qs] = top[31]; a.tq[cl_phys(qs)] = (uint32_t)(top[31] >> 32); }
    if (final) a.ntop[qs] = m;
  }
  if (final && lane < m) {
    const unsigned long long b = top[lane];
    const int32_t pos = (int32_t)(0xffffffffu - (uint32_t)(b & 0xffffffffu));
    int lo = 0, hi = a.C - 1;                                 // cent_pos grows with the column index
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.cent_pos[mid] < pos) lo = mid + 1; else hi = mid; }
    a.sel[qs * 32 + lane] = lo; a.selkey[qs * 32 + lane] = b;
  }
}

// ------------------------------------------------------------------ the candidate walk
__global__ void k_cl_init(ClusterArgs a)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs == 0) { for (int r = 0; r < 4; r++) a.dbg[r] = 0; for (int r = 0; r < 6; r++) a.work_n[r] = 0; }
  if (qs < a.nq) { a.replay[qs] = 0; a.skipm[qs] = 0; }
  if (qs >= 2 * a.nq) return;
  a.state[qs] = a.canon[qs >> 1] != (qs >> 1) ? 4 : (a.nk[qs] == 0 || a.C == 0) ? 3 : 0;      // 4 = a copy: reads its canonical query's state
  a.rejects[qs] = 0; a.acc_col[qs] = -1; a.wn[qs] = 0; a.selm[qs] = 0; a.sel_short[qs] = 0;
  a.prev[qs] = ~0ULL; a.bound[qs] = 0ULL; a.acc_id[qs] = -1.0; a.xn[qs] = 0; a.hard[qs] = 0;
}

// The walk takes its candidates from the strand's list (k_cl_topk) in rank order: round 0 the best one (most reads accept
// it), round 1 the whole remaining reject budget at once.  The (query strand, slot) items go to the alignment kernels' work list.
__global__ void k_cl_take(ClusterArgs a, int round)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs >= 2 * a.nq || a.state[qs] != 0) return;
  if (round == 1 && (qs & 1) && a.state[qs - 1] == 1 && a.acc_id[qs - 1] == 100.0) {
    // a minus-strand hit wins only with a HIGHER identity than the plus-strand hit: nothing beats 100 %, so the
    // remaining 31 candidates of this strand cannot change the outcome (k_cl_resolve cuts the window if the plus hit is lost)
    a.state[qs] = 3; a.skipm[qs >> 1] = 1;
    return;
  }
  const int first = round == 0 ? 0 : 1;
  const int need = round == 0 ? 1 : 32 - a.rejects[qs];
  int avail = a.ntop[qs] - first;
  avail = avail < 0 ? 0 : avail;
  const int m = avail < need ? avail : need;
  a.selm[qs] = m; a.sel_short[qs] = m < need ? 1 : 0;
  if (m > 0) { const int w0 = atomicAdd(&a.work_n[0], m); for (int k = 0; k < m; k++) a.work[w0 + k] = qs * 32 + first + k; }
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
// Whether such a path exists is a k-differences question: furthest-reaching diagonals (Landau-Vishkin) with every border
// cell as a source, one lane per diagonal, K + 1 synchronous levels.  No such path => the candidate is rejected without the
// O(Lq Lt) dynamic program; otherwise (all accepts, and the rare unlucky reject) the full alignment decides as before.
// For unrelated candidates the front dies after one or two symbols per diagonal; for near misses (a family member with one
// difference too many) Pmin is nearly the whole read and no front spans it.
// the certificate is used for edit budgets K <= 16 (identity thresholds >= ~0.95 at 300 bases; engine.hip sets pre_k)
static constexpr int PRE_LMAX = 2040;            // and reads up to this length (LDS); longer ones go straight to the full alignment

__device__ __forceinline__ uint32_t read_mask(const ReadsDev &rd, const uint32_t *w, int pos) { return 1u << ((w[pos >> 4] >> ((pos & 15) * 2)) & 3u); }

// returns 0 = proven reject, 1 (-1) = the full alignment decides, 2 = the bound from the best diagonal is too weak (the score pass,
// k_cl_score, computes the optimal score itself and asks again with it: have_lb)
// same_q: the wave's previous call had the same query strand and built its 8-mer index (not have_lb): q's masks and index are still in LDS
__device__ int precheck_pair(const ClusterArgs &a, int qs, int col, uint8_t *lds, bool have_lb = false, long long lb_given = 0, bool same_q = false)
{
  const int lane = threadIdx.x;
  const int qi = qs >> 1, s = qs & 1;
  const int64_t rq = a.order[a.f + qi], rt = a.cent_read[col];
  const int Lq = a.rd.len[rq], Lt = a.rd.len[rt];
  const int minL = Lq < Lt ? Lq : Lt;
  // K = the largest E for which ANY alignment with M <= minL matches could pass: M / (M + E) grows with M, and the test is
  // made with the very expression identity_of() evaluates, so rounding cannot open a gap between the two
  // (lane e tests budget e + 1: the quotient falls with e, so the first lane that fails is K)
  const unsigned long long kfail = __ballot(lane > a.pre_k || !(100.0 * (double)minL / (double)(minL + lane + 1) >= a.thr));
  const int K = __ffsll((long long)kfail) - 1;
  if (K > a.pre_k || Lq > PRE_LMAX || Lt > PRE_LMAX || Lq < 8 || Lt < 8) { if (lane == 0 && !have_lb) atomicAdd(&a.pre_stats[0], 1ULL); return have_lb ? 1 : -1; }   // no certificate: align (-1: and nothing was built in LDS)
  uint8_t *qm = lds;                                                        // the query's side first: it survives from one candidate to the next
  int32_t *head = reinterpret_cast<int32_t *>(qm + ((Lq + 3) & ~3));      // [256] chain heads of q's 8-mers (hashed)
  // (16-bit chain links, votes and front levels: positions stay below 2 048, and 8 KB of LDS per wave instead of 13 lets five waves
  // share a SIMD where three did -- the kernel waits on LDS round trips half of its time)
  int16_t *nextp = reinterpret_cast<int16_t *>(head + 256);                 // [Lq]
  uint16_t *qk = reinterpret_cast<uint16_t *>(nextp + ((Lq + 1) & ~1));     // [Lq] 8-mer at each q position (0xFFFF: none)
  uint8_t *tm = reinterpret_cast<uint8_t *>(qk + ((Lq + 1) & ~1));
  uint32_t *votes = reinterpret_cast<uint32_t *>(tm + ((Lt + 3) & ~3));     // [(Lq + Lt) / 2 + 2]: diagonal votes, two to a word; then a level of the front
  int16_t *fr = reinterpret_cast<int16_t *>(votes + ((Lq + Lt) >> 1) + 2);  // [Lq + Lt + 4]: the other level
  __shared__ int s_hit;
  const uint32_t *wq = a.rd.words + a.rd.woff[rq];
  const uint32_t *wt = a.rd.words + a.rd.woff[rt];
  const int64_t eoq = a.rd.excoff[rq], eot = a.rd.excoff[rt];
  const int nexq = (int)(a.rd.excoff[rq + 1] - eoq), next_ = (int)(a.rd.excoff[rt + 1] - eot);
  __syncthreads();
  if (!same_q) {
    for (int x = lane; x < Lq; x += 64) { const int o = s ? Lq - 1 - x : x; const uint32_t c2 = (wq[o >> 4] >> ((o & 15) * 2)) & 3u; qm[x] = (uint8_t)(1u << (s ? 3u - c2 : c2)); }
    for (int i = lane; i < 256; i += 64) head[i] = -1;
  }
  for (int o = lane; o < Lt; o += 64) tm[o] = (uint8_t)(1u << ((wt[o >> 4] >> ((o & 15) * 2)) & 3u));
  for (int i = lane; i < ((Lq + Lt) >> 1) + 1; i += 64) votes[i] = 0u;
  if (lane == 0) s_hit = 0;
  __syncthreads();
  if (!same_q) for (int e = lane; e < nexq; e += 64) { const uint32_t ex = a.rd.exc[eoq + e]; const int pos = (int)(ex >> 4); const uint32_t m = mask4(ex & 15u); qm[s ? Lq - 1 - pos : pos] = (uint8_t)(s ? revmask4(m) : m); }
  for (int e = lane; e < next_; e += 64) { const uint32_t ex = a.rd.exc[eot + e]; tm[ex >> 4] = (uint8_t)mask4(ex & 15u); }
  __syncthreads();
  long long lb = -(long long)(Lq + Lt + 4);                 // no pairs at all: two terminal runs
  if (have_lb) lb = lb_given;                               // the optimal score itself: no alignment scores more, so the bound below is the tightest
  else {
  // ---- the diagonal with most shared 8-mers (unambiguous symbols only)
  auto kmer_at = [&](const uint8_t *m, int L, int p, uint32_t &k) {
    if (p + 8 > L) return false;
    k = 0;
    for (int t = 0; t < 8; t++) { const uint32_t b = m[p + t]; if (__popc(b) != 1) return false; k |= (uint32_t)(__ffs(b) - 1) << (2 * t); }
    return true;
  };
  if (!same_q) {
    for (int p = lane; p < Lq; p += 64) {
      uint32_t k;
      if (kmer_at(qm, Lq, p, k)) { qk[p] = (uint16_t)k; nextp[p] = (int16_t)atomicExch(&head[(k * 40503u >> 8) & 255u], p); } else qk[p] = 0xFFFF;
    }
    __syncthreads();
  }
  for (int j = lane; j < Lt; j += 64) {
    uint32_t k;
    if (!kmer_at(tm, Lt, j, k)) continue;
    for (int p = head[(k * 40503u >> 8) & 255u]; p >= 0; p = nextp[p]) if (qk[p] == (uint16_t)k) { const int z = p - j + Lt - 1; atomicAdd(&votes[z >> 1], 1u << ((z & 1) * 16)); }
  }
  __syncthreads();
  int bv = -1, bd = 0;
  for (int i = lane; 2 * i < Lq + Lt - 1; i += 64) {          // (a word's low half is the lower diagonal: ties go to the lowest index as before)
    const uint32_t w = votes[i];
    const int lo = (int)(w & 0xffffu), hi = (int)(w >> 16);
    if (lo > bv) { bv = lo; bd = 2 * i; }
    if (2 * i + 1 < Lq + Lt - 1 && hi > bv) { bv = hi; bd = 2 * i + 1; }
  }
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
  if (bv > 0 && i_hi > i_lo) {
    const int left = d > 0 ? d : -d;                        // one of the two reads overhangs on the left, one on the right
    const int right = (Lq - i_hi) + (Lt - (i_hi - d));
    const long long sc = part - (left ? 2 + left : 0) - (right ? 2 + right : 0);
    if (sc > lb) lb = sc;
  }
  }
  const long long num = lb + Lq + Lt;
  const int Pmin = num <= 0 ? 0 : (int)((num + 3) / 4);
  if (Pmin <= K + 4) { if (lane == 0 && !have_lb) atomicAdd(&a.pre_stats[1], 1ULL); return have_lb ? 1 : 2; }   // too weak to exclude chance overlaps
  // ---- k-differences reachability from ALL border cells at once.  Diagonal d = i - j (slot z = d + Lt) has exactly one border cell
  // to start from, (max(d, 0), max(-d, 0)), and exactly one to end in; F[z] = the furthest row reached on it within e edits from ANY
  // start (furthest-reaching diagonals with every start as a source of level 0: whatever one start reaches, the merged front reaches).
  // The front no longer knows its start, so the span is measured from the most favourable start within K diagonals of the end
  // diagonal -- a path of <= K edits cannot have come from further away -- which gives away at most K symbols of Pmin.
  // Lanes own diagonals, levels are synchronous: (K + 1) (Lq + Lt + 1) independent extensions instead of one walk per start.
  int16_t *P = reinterpret_cast<int16_t *>(votes) + 1, *N = fr + 1;      // [-1 .. nd]: a sentinel either side
  const int nd = Lq + Lt + 1;
  auto span_ok = [&](int i, int j, int d) {
    int rows = i, cols = j;
    if (d > K) rows = i - (d - K); else if (d < -K) cols = j + d + K;
    return rows >= Pmin && cols >= Pmin;
  };
  bool hit = false;
  __syncthreads();                                           // votes[] is free now
  for (int z = lane; z < nd; z += 64) {
    const int d = z - Lt;
    int i = d > 0 ? d : 0, j = i - d;
    while (i < Lq && j < Lt && (qm[i] & tm[j])) { i++; j++; }
    P[z] = i;
    if ((i == Lq || j == Lt) && span_ok(i, j, d)) hit = true;
  }
  if (lane == 0) { P[-1] = P[nd] = -1; N[-1] = N[nd] = -1; }
  for (int e = 1; e <= K; e++) {
    __syncthreads();
    if (__ballot(hit) != 0ull) break;
    for (int z = lane; z < nd; z += 64) {
      const int d = z - Lt;
      const int pa = P[z], pb = P[z - 1], pc = P[z + 1];
      int i = pa;
      if (pa < Lq && pa - d < Lt) i = pa + 1;                                   // a mismatching pair
      if (pb >= 0 && pb < Lq && pb + 1 > i) i = pb + 1;                         // a q symbol against a gap (from diagonal d - 1)
      if (pc >= 0 && pc - d <= Lt && pc > i) i = pc;                            // a t symbol against a gap (from diagonal d + 1)
      int j = i - d;
      while (i < Lq && j < Lt && (qm[i] & tm[j])) { i++; j++; }
      N[z] = i;
      if ((i == Lq || j == Lt) && span_ok(i, j, d)) hit = true;
    }
    int16_t *t_ = P; P = N; N = t_;
  }
  const bool any_hit = __ballot(hit) != 0ull;
  if (lane == 0 && any_hit) s_hit = 1;
  __syncthreads();
  if (lane == 0) atomicAdd(&a.pre_stats[have_lb ? (s_hit ? 5 : 4) : (s_hit ? 2 : 3)], 1ULL);
  return s_hit != 0 ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_cl_precheck(ClusterArgs a, int which, int per)
{
  extern __shared__ uint8_t pre_lds[];
  const int nw = a.work_n[which];
  const int32_t *work = which ? a.xwork : a.work;
  const int32_t *cols = which ? a.xlist : a.sel;
  // the items of a query strand are consecutive in the list (k_cl_take): a wave takes `per` of them in a row (small: a related strand's items are the expensive ones) and builds a
  // strand's masks and 8-mer index once (indexed = the strand whose index is in LDS: an early return of precheck_pair builds nothing)
  int indexed = -1;
  for (int w0 = blockIdx.x * per; w0 < nw; w0 += gridDim.x * per)
  for (int w = w0; w < w0 + per && w < nw; w++) {
    const int item = work[w];
    const int qs = item >> 5;
    int need = precheck_pair(a, qs, cols[item], pre_lds, false, 0, qs == indexed);
    indexed = need == -1 ? -1 : qs;
    if (need < 0) need = 1;
    if (need == 2 && !a.use_score) need = 1;                  // no score pass: the full alignment decides
    if (threadIdx.x == 0) {
      a.need[which * a.need_pitch + item] = need;
      if (need == 1) a.awork[which * a.need_pitch + atomicAdd(&a.work_n[2 + which], 1)] = item;      // the alignment kernels take these, one each
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ the score pass: the optimal score decides most rejections
// 88 % of the full alignments were candidates whose best diagonal says nothing (unrelated reads that share a conserved motif):
// the certificate's bound P >= (LB + Lq + Lt) / 4 needs a better LB, and the best there is is the optimal score itself.  The same
// wavefront as align_pair, on 32-bit scores alone (no matches / columns, a third of the instructions); with it the certificate's
// k-differences test runs again (unrelated pairs score about -650: Pmin ~ 37, no 2-edit path of that length) and only pairs that
// pass it get the 64-bit dynamic program.  Single-pass queries only (Lq + 1 <= 64 S); the others keep need = 1.
static constexpr int NEG32 = -(1 << 28);
template <int S> __device__ __forceinline__ bool score_pair(const ClusterArgs &a, int qs, int col, uint8_t *tmask, int &res)
{
  const int lane = threadIdx.x;
  const int qi = qs >> 1, s = qs & 1;
  const int64_t rq = a.order[a.f + qi], rt = a.cent_read[col];
  const int Lq = a.rd.len[rq], Lt = a.rd.len[rt];
  const uint32_t *wq = a.rd.words + a.rd.woff[rq];
  const uint32_t *wt = a.rd.words + a.rd.woff[rt];
  const int64_t eoq = a.rd.excoff[rq], eot = a.rd.excoff[rt];
  const int nexq = (int)(a.rd.excoff[rq + 1] - eoq), next_ = (int)(a.rd.excoff[rt + 1] - eot);
  res = NEG32; bool have = false;
  __syncthreads();
  for (int o = lane; o < Lt; o += 64) tmask[o] = (uint8_t)(1u << ((wt[o >> 4] >> ((o & 15) * 2)) & 3u));
  __syncthreads();
  for (int e = lane; e < next_; e += 64) { const uint32_t ex = a.rd.exc[eot + e]; tmask[ex >> 4] = (uint8_t)mask4(ex & 15u); }
  __syncthreads();
  const int i0 = lane * S;
  uint32_t qm[S]; bool qu[S]; int goE[S], geE[S];
#pragma unroll
  for (int r = 0; r < S; r++) {
    const int i = i0 + r;
    uint32_t m = 0;
    if (i >= 1 && i <= Lq) {
      const int x = i - 1, o = s ? Lq - 1 - x : x;
      const uint32_t c2 = (wq[o >> 4] >> ((o & 15) * 2)) & 3u;
      m = 1u << (s ? 3u - c2 : c2);
    }
    qm[r] = m; goE[r] = (i == 0 || i == Lq) ? -3 : -22; geE[r] = (i == 0 || i == Lq) ? -1 : -2;
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
  int Hl[S], El[S];
#pragma unroll
  for (int r = 0; r < S; r++) { Hl[r] = NEG32; El[r] = NEG32; }
  int diag_carry = lane == 0 ? 0 : NEG32, pubH = NEG32, pubF = NEG32;      // cell (0, 0) is 0
  const int nsteps = Lt + 1 + 63;
  for (int t = 0; t < nsteps; t++) {
    int upH = __shfl_up(pubH, 1), upF = __shfl_up(pubF, 1);
    const int j = t - lane;
    if (lane == 0) { upH = NEG32; upF = NEG32; }
    if (j >= 0 && j <= Lt) {
      const uint32_t tm = j >= 1 ? (uint32_t)tmask[j - 1] : 0u;
      const bool tF = (j == 0 || j == Lt);
      const int goF = tF ? -3 : -22, geF = tF ? -1 : -2;
      const bool tu = unamb4(tm);
      int aboveH = upH, aboveF = upF, dg = diag_carry;
#pragma unroll
      for (int r = 0; r < S; r++) {
        const int E = max(Hl[r] + goE[r], El[r] + geE[r]);
        const int F = max(aboveH + goF, aboveF + geF);
        const int D = (qu[r] && tu) ? ((qm[r] & tm) ? 2 : -4) : 0;       // +2 / -4 between unambiguous symbols, 0 otherwise
        const int Hn = max(max(dg + D, E), F);
        dg = Hl[r];
        Hl[r] = Hn; El[r] = E;
        aboveH = Hn; aboveF = F;
      }
      pubH = aboveH; pubF = aboveF;
      diag_carry = upH;
    }
  }
#pragma unroll
  for (int r = 0; r < S; r++) if (i0 + r == Lq) { res = Hl[r]; have = true; }
  return have;
}
// Two candidates of ONE query strand at a time: the scores of a cell of both dynamic programs as the 16-bit halves of one register
// (packed add / max: the scores stay within +-2 (Lq + Lt) + 44, far inside 16 bits; NEG16 plays NEG32's part and nothing derived
// from it ever beats a real score or wraps), the query side -- masks, gap flags -- shared.  The shorter target's half runs on past
// its last column with whatever it finds there; a target's score is taken when ITS last column passes the lane that holds row Lq.
typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
static constexpr int NEG16 = -20000;
__device__ __forceinline__ pk16 pk_of(int lo, int hi) { return (pk16){(short)lo, (short)hi}; }
__device__ __forceinline__ pk16 pk_max(pk16 x, pk16 y) { return __builtin_elementwise_max(x, y); }
__device__ __forceinline__ int pk_bits(pk16 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ pk16 pk_from(int x) { return __builtin_bit_cast(pk16, x); }
template <int S> __device__ __forceinline__ void score_pair2(const ClusterArgs &a, int qs, int col1, int col2, uint8_t *tmask1, uint8_t *tmask2, int &res1, int &res2)
{
  const int lane = threadIdx.x;
  const int qi = qs >> 1, s = qs & 1;
  const int64_t rq = a.order[a.f + qi], rt1 = a.cent_read[col1], rt2 = a.cent_read[col2];
  const int Lq = a.rd.len[rq], Lt1 = a.rd.len[rt1], Lt2 = a.rd.len[rt2];
  const uint32_t *wq = a.rd.words + a.rd.woff[rq];
  const int64_t eoq = a.rd.excoff[rq];
  const int nexq = (int)(a.rd.excoff[rq + 1] - eoq);
  __syncthreads();
  for (int k = 0; k < 2; k++) {
    const int64_t rt = k ? rt2 : rt1;
    uint8_t *tmask = k ? tmask2 : tmask1;
    const uint32_t *wt = a.rd.words + a.rd.woff[rt];
    const int Lt = k ? Lt2 : Lt1;
    for (int o = lane; o < Lt; o += 64) tmask[o] = (uint8_t)(1u << ((wt[o >> 4] >> ((o & 15) * 2)) & 3u));
  }
  __syncthreads();
  for (int k = 0; k < 2; k++) {
    const int64_t rt = k ? rt2 : rt1;
    uint8_t *tmask = k ? tmask2 : tmask1;
    const int64_t eot = a.rd.excoff[rt];
    const int next_ = (int)(a.rd.excoff[rt + 1] - eot);
    for (int e = lane; e < next_; e += 64) { const uint32_t ex = a.rd.exc[eot + e]; tmask[ex >> 4] = (uint8_t)mask4(ex & 15u); }
  }
  __syncthreads();
  const int i0 = lane * S;
  uint32_t qpk[S], qall[S]; pk16 goE[S], geE[S];         // the row's mask in both halves (0 when ambiguous), all ones when unambiguous
#pragma unroll
  for (int r = 0; r < S; r++) {
    const int i = i0 + r;
    uint32_t m = 0;
    if (i >= 1 && i <= Lq) {
      const int x = i - 1, o = s ? Lq - 1 - x : x;
      const uint32_t c2 = (wq[o >> 4] >> ((o & 15) * 2)) & 3u;
      m = 1u << (s ? 3u - c2 : c2);
    }
    qpk[r] = m;
    const bool term = (i == 0 || i == Lq);
    goE[r] = pk_of(term ? -3 : -22, term ? -3 : -22); geE[r] = pk_of(term ? -1 : -2, term ? -1 : -2);
  }
  for (int e = 0; e < nexq; e++) {
    const uint32_t ex = a.rd.exc[eoq + e];
    const int pos = (int)(ex >> 4);
    const int i = (s ? Lq - 1 - pos : pos) + 1;
    const uint32_t m = s ? revmask4(mask4(ex & 15u)) : mask4(ex & 15u);
#pragma unroll
    for (int r = 0; r < S; r++) if (i == i0 + r) qpk[r] = m;
  }
  // an unambiguous row symbol c becomes the value 6 in a 3-bit field of its own, 6 << 3 c, in both halves: shifted right by
  // 3 x (the target's symbol) and cut to 3 bits it leaves 6 where the two match and 0 where they do not
  bool isrow[S];
#pragma unroll
  for (int r = 0; r < S; r++) {
    const bool u = unamb4(qpk[r]);
    const uint32_t six = u ? 6u << (3u * ((uint32_t)__ffs((int)qpk[r]) - 1u)) : 0u;
    qall[r] = u ? 0xFFFFFFFFu : 0u; qpk[r] = six | (six << 16);
    isrow[r] = (i0 + r == Lq);
  }
  const pk16 neg = pk_of(NEG16, NEG16);
  pk16 Hl[S], El[S];
#pragma unroll
  for (int r = 0; r < S; r++) { Hl[r] = neg; El[r] = neg; }
  pk16 diag_carry = lane == 0 ? pk_of(0, 0) : neg, pubH = neg, pubF = neg;      // cell (0, 0) is 0
  const int Ltmax = Lt1 > Lt2 ? Lt1 : Lt2;
  const int nsteps = Ltmax + 1 + 63;
  const bool mine = (Lq >= i0 && Lq < i0 + S);            // this lane holds row Lq
  int cap1 = 0, cap2 = 0;
  // No lane is switched off while the wavefront passes: a lane that has not reached column 0 yet (j < 0) works on columns that do
  // not exist, from NEG16 state and with no symbol, so everything it produces stays near NEG16 -- exactly what column 0 wants to its
  // left -- and a lane past its target's last column produces values nobody reads (the result was taken when that column passed).
  for (int t = 0; t < nsteps; t++) {
    pk16 upH = pk_from(__shfl_up(pk_bits(pubH), 1)), upF = pk_from(__shfl_up(pk_bits(pubF), 1));
    const int j = t - lane;
    if (lane == 0) { upH = neg; upF = neg; }
    const int jc = j < 1 ? 0 : (j > Ltmax ? Ltmax - 1 : j - 1);
    uint32_t tm1 = (uint32_t)tmask1[jc], tm2 = (uint32_t)tmask2[jc];
    if (j < 1) { tm1 = 0u; tm2 = 0u; }
    const bool tu1 = unamb4(tm1), tu2 = unamb4(tm2);
    const uint32_t ct1 = tu1 ? 3u * ((uint32_t)__ffs((int)tm1) - 1u) : 12u, ct2 = tu2 ? 3u * ((uint32_t)__ffs((int)tm2) - 1u) : 12u;
    const pku16 ctpk = __builtin_bit_cast(pku16, ct1 | (ct2 << 16));              // (12: an ambiguous target symbol matches nothing)
    const pku16 seven = (pku16){7, 7};
    const uint32_t tbase = (tu1 ? 0x0000FFFCu : 0u) | (tu2 ? 0xFFFC0000u : 0u);            // -4 where the target symbol is unambiguous
    const bool tF1 = (j == 0 || j == Lt1), tF2 = (j == 0 || j == Lt2);
    const pk16 goF = pk_from((int)((tF1 ? 0xFFFDu : 0xFFEAu) | (tF2 ? 0xFFFD0000u : 0xFFEA0000u)));       // -3 : -22
    const pk16 geF = pk_from((int)((tF1 ? 0xFFFFu : 0xFFFEu) | (tF2 ? 0xFFFF0000u : 0xFFFE0000u)));       // -1 : -2
    pk16 aboveH = upH, aboveF = upF, dg = diag_carry;
#pragma unroll
    for (int r = 0; r < S; r++) {
      const pk16 E = pk_max(Hl[r] + goE[r], El[r] + geE[r]);
      const pk16 F = pk_max(aboveH + goF, aboveF + geF);
      // +2 / -4 between unambiguous symbols, 0 otherwise: -4 where both are, + 6 where they match
      const pku16 hit6 = (__builtin_bit_cast(pku16, qpk[r]) >> ctpk) & seven;
      const pk16 D = __builtin_bit_cast(pk16, hit6) + pk_from((int)(qall[r] & tbase));
      const pk16 Hn = pk_max(pk_max(dg + D, E), F);
      dg = Hl[r];
      Hl[r] = Hn; El[r] = E;
      aboveH = Hn; aboveF = F;
    }
    pubH = aboveH; pubF = aboveF;
    diag_carry = upH;
    int hq = 0;                                                                    // row Lq of this column, where this lane holds it
#pragma unroll
    for (int r = 0; r < S; r++) hq = isrow[r] ? pk_bits(Hl[r]) : hq;
    cap1 = j == Lt1 ? hq : cap1; cap2 = j == Lt2 ? hq : cap2;
  }
  if (mine) { res1 = (int)pk_from(cap1).x; res2 = (int)pk_from(cap2).y; }
}
// the score pass's list: the items the certificate could not judge, two of one strand per entry where a strand has two (a strand's
// items are consecutive in the work list; a thread looks at 32 entries)
__global__ void k_cl_pairs(ClusterArgs a, int which)
{
  const int nw = a.work_n[which];
  const int32_t *work = which ? a.xwork : a.work;
  const int32_t *need = a.need + which * a.need_pitch;
  int32_t *out = a.spairs + (size_t)which * a.need_pitch * 2;
  const int w0 = (blockIdx.x * blockDim.x + threadIdx.x) * 32;
  int pend = -1;
  for (int w = w0; w < w0 + 32 && w < nw; w++) {
    const int item = work[w];
    if (need[item] != 2) continue;
    if (pend >= 0 && (pend >> 5) == (item >> 5)) { const int o = atomicAdd(&a.work_n[4 + which], 1); out[2 * o] = pend; out[2 * o + 1] = item; pend = -1; }
    else { if (pend >= 0) { const int o = atomicAdd(&a.work_n[4 + which], 1); out[2 * o] = pend; out[2 * o + 1] = -1; } pend = item; }
  }
  if (pend >= 0) { const int o = atomicAdd(&a.work_n[4 + which], 1); out[2 * o] = pend; out[2 * o + 1] = -1; }
}
template <int S> __global__ __launch_bounds__(64) void k_cl_score(ClusterArgs a, int which, int lds_pre, int lds_t2)
{
  extern __shared__ uint8_t sc_lds[];
  const int np = a.work_n[4 + which];
  const int32_t *pairs = a.spairs + (size_t)which * a.need_pitch * 2;
  const int32_t *cols = which ? a.xlist : a.sel;
  int32_t *need = a.need + which * a.need_pitch;
  for (int w = blockIdx.x; w < np; w += gridDim.x) {
    const int c[2] = {pairs[2 * w], pairs[2 * w + 1]};
    const int qs = c[0] >> 5;
    const int Lq = a.rd.len[a.order[a.f + (qs >> 1)]];
    const bool pair = c[1] >= 0;
    int verdict[2] = {1, 1};
    if (Lq + 1 <= 64 * S) {
      if (pair) {
        int r1 = NEG32, r2 = NEG32;
        score_pair2<S>(a, qs, cols[c[0]], cols[c[1]], sc_lds + lds_pre, sc_lds + lds_t2, r1, r2);
        const unsigned long long who = __ballot(r1 != NEG32 || r2 != NEG32);        // the lane that holds row Lq
        const int src = __ffsll((long long)who) - 1;
        const int s1 = __shfl(r1, src), s2 = __shfl(r2, src);
        verdict[0] = precheck_pair(a, qs, cols[c[0]], sc_lds, true, (long long)s1);
        verdict[1] = precheck_pair(a, qs, cols[c[1]], sc_lds, true, (long long)s2);
      } else {
        int res;
        const bool have = score_pair<S>(a, qs, cols[c[0]], sc_lds + lds_pre, res);
        const unsigned long long who = __ballot(have);
        const int sstar = __shfl(res, __ffsll((long long)who) - 1);
        verdict[0] = precheck_pair(a, qs, cols[c[0]], sc_lds, true, (long long)sstar);
      }
    }
    if (threadIdx.x == 0) {
      for (int y = 0; y < (pair ? 2 : 1); y++) {
        need[c[y]] = verdict[y];
        if (verdict[y] == 1) a.awork[which * a.need_pitch + atomicAdd(&a.work_n[2 + which], 1)] = c[y];
      }
    }
    __syncthreads();
  }
}

// proven rejections: identity -1 (below every threshold), counted
__global__ void k_cl_skipped(ClusterArgs a, int which)
{
  const int nw = a.work_n[which];
  const int32_t *work = which ? a.xwork : a.work;
  double *pid = which ? a.xpid : a.selpid;
  unsigned long long n = 0;
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += gridDim.x * blockDim.x) {
    const int item = work[w];
    if (a.need[which * a.need_pitch + item] == 0) { pid[item] = -1.0; n++; }
  }
  for (int off = 32; off; off >>= 1) n += __shfl_xor(n, off);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(a.n_skipped, n);
}

// one wave = one (query strand, selected candidate) alignment; the walk consumes the identities in rank order
template <int S> __global__ __launch_bounds__(64) void k_cl_align(ClusterArgs a)
{
  extern __shared__ uint8_t tmask_lds[];
  // with the certificate on, the items that need the dynamic program come as a compact list of their own (the proven rejections
  // got their -1 from k_cl_skipped): consecutive waves, one alignment each, instead of a few waves meeting several
  const int nw = a.need ? a.work_n[2] : a.work_n[0];
  const int32_t *list = a.need ? a.awork : a.work;
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {      // every wave drains its share of the list and exits
    const int item = list[w];
    i64 res;
    if (align_pair<S>(a, item >> 5, a.sel[item], item, tmask_lds, res)) {
      a.selpid[item] = identity_of(res);
      atomicAdd(a.n_align, 1ULL);
    }
  }
}

__global__ void k_cl_walk(ClusterArgs a, int first)
{
  const int qs = blockIdx.x * blockDim.x + threadIdx.x;
  if (qs >= 2 * a.nq || a.state[qs] != 0) return;
  const int m = a.selm[qs];
  int w = a.wn[qs], rej = a.rejects[qs], st = 0;
  for (int k = first; k < first + m; k++) {
    const u64 key = a.selkey[qs * 32 + k];
    const double pid = a.selpid[qs * 32 + k];
    a.wkey[qs * 32 + w] = key; a.wpid[qs * 32 + w] = pid; a.wcol[qs * 32 + w] = a.sel[qs * 32 + k];
    w++;
    if (pid >= a.thr) { st = 1; a.acc_col[qs] = a.sel[qs * 32 + k]; a.acc_id[qs] = pid; a.bound[qs] = key; break; }
    rej++;
    if (rej >= 32) { st = 2; a.bound[qs] = key; break; }
  }
  if (st == 0) { if (a.sel_short[qs]) st = 3; else if (m > 0) a.prev[qs] = a.selkey[qs * 32 + first + m - 1]; }
  a.state[qs] = st; a.wn[qs] = w; a.rejects[qs] = rej; a.selm[qs] = 0;
}
__global__ void k_cl_reset_work(ClusterArgs a, int which) { a.work_n[which] = 0; a.work_n[2 + which] = 0; a.work_n[4 + which] = 0; }

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

// words of every would-be centroid of the window (their exclusive scan places the word lists in the pool)
__global__ void k_cl_wsum(ClusterArgs a)
{
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi == 0) a.cw_base[0] = a.cw_off[a.C];
  if (qi > a.nq) return;
  a.wsum[qi] = (qi < a.nq && a.is_new[qi]) ? a.nk[2 * a.canon[qi]] : 0;
}
// register each speculative centroid as a column: its distinct forward words go to the pool (mode 0); or roll back the
// columns flagged in rm[] (mode 1: the column stays, without words, and never counts again); one wave per query
__global__ __launch_bounds__(64) void k_cl_columns(ClusterArgs a, int clear)
{
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (qi == a.nq) {                                          // one extra block: where the next window's first column starts
    if (lane == 0 && !clear) a.cw_off[a.C + a.new_rank[a.nq]] = a.cw_base[0] + a.wscan[a.nq];
    return;
  }
  if (clear ? !a.rm[qi] : !a.is_new[qi]) return;
  const int col = a.C + a.new_rank[qi];
  if (clear) { if (lane == 0) a.cw_n[col] = 0; return; }
  const int qs = 2 * a.canon[qi];
  const int n = a.nk[qs];
  const uint16_t *kl = a.klist + (size_t)qs * a.kcap;
  const int64_t off = a.cw_base[0] + a.wscan[qi];
  for (int i = lane; i < n; i += 64) a.cw_pool[off + i] = kl[i];
  if (lane == 0) {
    const int64_t r = a.order[a.f + qi];
    a.cw_off[col] = off; a.cw_n[col] = n;
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
  if ((qs & 1) && a.skipm[cqi]) return;                  // this strand's walk was cut short: it has no say (see k_cl_take)
  if (tid == 0) xcount = 0;
  __syncthreads();
  const int pos = a.f + (qs >> 1);
  const uint32_t minm = n < 12 ? n : 12;
  const int state = a.state[cs];
  const u64 bound = state == 3 ? 0ULL : a.bound[cs];
  const uint16_t *cn = a.cntx + (size_t)cs * a.xpitch;      // k_cl_stream, mode 2
  for (int c = a.C + tid; c < a.C + n_new; c += 256) {
    if (a.cent_pos[c] >= pos) break;
    const uint32_t v = cn[c - a.C];
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
  const int nw = a.need ? a.work_n[3] : a.work_n[1];
  const int32_t *list = a.need ? a.awork + a.need_pitch : a.xwork;
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {
    const int item = list[w];
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
  if (lane == 0) { a.wout[0] = cut; a.wout[1] = a.new_rank[cut]; a.wout[2] = ntrue; a.cw_base[1] = a.cw_off[a.C + a.new_rank[cut]]; }
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

// ------------------------------------------------------------------ launchers
void launch_cl_kmers(const ClusterArgs &a, hipStream_t st)
{
  (void)hipMemsetAsync(a.ctab_key, 0, CTAB * sizeof(unsigned long long), st);
  (void)hipMemsetAsync(a.ctab_val, 0x7f, CTAB * sizeof(int32_t), st);
  hipLaunchKernelGGL(k_cl_canon_insert, dim3((a.nq + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_cl_canon_lookup, dim3((a.nq + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_cl_kmers, dim3(2 * a.nq), dim3(256), 0, st, a);
}
void launch_cl_qindex(const ClusterArgs &a, int32_t *scan_tmp, hipStream_t st)
{
  (void)hipMemsetAsync(a.qi_cnt, 0, 65537 * sizeof(int32_t), st);
  (void)hipMemsetAsync(a.qi_nheavy, 0, sizeof(int32_t), st);
  hipLaunchKernelGGL(k_cl_qi_count, dim3(2 * a.nq), dim3(256), 0, st, a, 0);
  hipLaunchKernelGGL(k_cl_qi_pad, dim3(65537 / 256 + 1), dim3(256), 0, st, a);
  launch_exclusive_scan(a.qi_cnt, a.qi_off, 65537, scan_tmp, st);
  hipLaunchKernelGGL(k_cl_qi_fill_dummy, dim3(1024), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_cl_qi_count, dim3(2 * a.nq), dim3(256), 0, st, a, 1);
  // strands past the window's end never reach a threshold (k_cl_stream scans four strands per load without a bound check)
  (void)hipMemsetAsync(a.tq, 0xff, (size_t)(CL_QS_MAX + 8) * sizeof(uint32_t), st);
  (void)hipMemsetAsync(a.minm, 0xff, (size_t)(CL_QS_MAX + 8) * sizeof(uint32_t), st);
  hipLaunchKernelGGL(k_cl_tq_init, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a);
}
void launch_cl_stream(const ClusterArgs &a, int c0, int c1, int mode, hipStream_t st)
{
  if (c1 <= c0) return;
  hipLaunchKernelGGL(k_cl_stream, dim3(std::min(c1 - c0, 1024)), dim3(256), 0, st, a, c0, c1, mode);      // four workgroups per CU (LDS 39 KB, <= 128 registers)
}
void launch_cl_topk(const ClusterArgs &a, int final, hipStream_t st) { hipLaunchKernelGGL(k_cl_topk, dim3(2 * a.nq), dim3(64), 0, st, a, final); }
void launch_cl_init(const ClusterArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_cl_init, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a); }
static size_t precheck_lds(const ClusterArgs &a)
{
  const size_t L = (size_t)std::min(a.scratch_pitch, PRE_LMAX + 1);      // masks, 8-mer index of the query, votes, two levels of the front
  return 2 * (L + 4) + 1024 + 2 * (L + 2) + 2 * (L + 2) + 2 * (4 * L + 16) + 64;
}
static void launch_cl_score(const ClusterArgs &a, int which, int grid, int maxitems, int rows_per_lane, hipStream_t st)
{
  if (!a.use_score) return;                                   // (ITSX_CL_NOSCORE=1: the too-weak candidates go straight to the full alignment)
  const int pre = (int)((precheck_lds(a) + 63) & ~(size_t)63);
  hipLaunchKernelGGL(k_cl_pairs, dim3((maxitems + 32 * 256 - 1) / (32 * 256)), dim3(256), 0, st, a, which);
  const int tpitch = (int)(((size_t)a.scratch_pitch + 63) & ~(size_t)63);      // one target's masks; two targets at a time
  const size_t lds = (size_t)pre + 2 * (size_t)tpitch;
  if (rows_per_lane <= 5) hipLaunchKernelGGL(k_cl_score<5>, dim3(grid), dim3(64), lds, st, a, which, pre, pre + tpitch);
  else if (rows_per_lane <= 8) hipLaunchKernelGGL(k_cl_score<8>, dim3(grid), dim3(64), lds, st, a, which, pre, pre + tpitch);
  else hipLaunchKernelGGL(k_cl_score<10>, dim3(grid), dim3(64), lds, st, a, which, pre, pre + tpitch);
}
void launch_cl_walk(const ClusterArgs &a, int rows_per_lane, hipStream_t st)
{
  // round 0: the best candidate of every query (most reads accept it); round 1: the whole remaining reject budget at once
  for (int round = 0; round < 2; round++) {
    const int kmax = round == 0 ? 1 : 31;
    hipLaunchKernelGGL(k_cl_take, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a, round);
    const int grid = std::min(2 * a.nq * kmax, 16384);
    const size_t lds = ((size_t)a.scratch_pitch + 63) & ~(size_t)63;
    if (a.need) {
      const int per = round == 0 ? 1 : 4;
      hipLaunchKernelGGL(k_cl_precheck, dim3(std::min((2 * a.nq * kmax + per - 1) / per, 65536)), dim3(64), precheck_lds(a), st, a, 0, per);
      launch_cl_score(a, 0, grid, 2 * a.nq * kmax, rows_per_lane, st);
      hipLaunchKernelGGL(k_cl_skipped, dim3(256), dim3(256), 0, st, a, 0);
    }
    if (rows_per_lane <= 5) hipLaunchKernelGGL(k_cl_align<5>, dim3(grid), dim3(64), lds, st, a);
    else if (rows_per_lane <= 8) hipLaunchKernelGGL(k_cl_align<8>, dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL(k_cl_align<10>, dim3(grid), dim3(64), lds, st, a);
    hipLaunchKernelGGL(k_cl_walk, dim3((2 * a.nq + 255) / 256), dim3(256), 0, st, a, round);
    hipLaunchKernelGGL(k_cl_reset_work, dim3(1), dim3(1), 0, st, a, 0);
  }
}
void launch_cl_outcome(const ClusterArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_cl_outcome, dim3((a.nq + 255) / 256), dim3(256), 0, st, a); }
void launch_cl_wsum(const ClusterArgs &a, hipStream_t st) { hipLaunchKernelGGL(k_cl_wsum, dim3((a.nq + 256) / 256), dim3(256), 0, st, a); }
void launch_cl_columns(const ClusterArgs &a, int clear, hipStream_t st) { hipLaunchKernelGGL(k_cl_columns, dim3(a.nq + 1), dim3(64), 0, st, a, clear); }
void launch_cl_validate(const ClusterArgs &a, int rows_per_lane, hipStream_t st)
{
  hipLaunchKernelGGL(k_cl_affected, dim3(2 * a.nq), dim3(256), 0, st, a);
  const int grid = std::min(2 * a.nq * 32, 16384);
  const size_t lds = ((size_t)a.scratch_pitch + 63) & ~(size_t)63;
  if (a.need) {
    hipLaunchKernelGGL(k_cl_precheck, dim3(grid), dim3(64), precheck_lds(a), st, a, 1, 1);
    launch_cl_score(a, 1, grid, 2 * a.nq * 32, rows_per_lane, st);
    hipLaunchKernelGGL(k_cl_skipped, dim3(256), dim3(256), 0, st, a, 1);
  }
  if (rows_per_lane <= 5) hipLaunchKernelGGL(k_cl_align_x<5>, dim3(grid), dim3(64), lds, st, a);
  else if (rows_per_lane <= 8) hipLaunchKernelGGL(k_cl_align_x<8>, dim3(grid), dim3(64), lds, st, a);
  else hipLaunchKernelGGL(k_cl_align_x<10>, dim3(grid), dim3(64), lds, st, a);
  hipLaunchKernelGGL(k_cl_resolve, dim3(1), dim3(64), 0, st, a);
}
void launch_cl_finalize(int32_t nk, const int32_t *order, const int32_t *res_col, const int8_t *res_strand, const double *res_id,
                        const int32_t *cent_read, int32_t *rep_of, int8_t *strand, double *pct, int32_t *is_seed, hipStream_t st)
{
  if (nk > 0) hipLaunchKernelGGL(k_cl_finalize, dim3((nk + 255) / 256), dim3(256), 0, st, nk, order, res_col, res_strand, res_id, cent_read, rep_of, strand, pct, is_seed);
}
}  // namespace itsx

// ------------------------------------------------------------------ f4: read orientation (vsearch --orient restated)
// Reference call site itsxpress/SeqSample.py:48-91.  One block per read: its distinct unambiguous 12-mers (an LDS hash
// set removes repeats) and their reverse complements are looked up in the database's 12-mer bitmap (2 MB, L2-resident);
// forward when count_fwd >= 1 and >= 4 x count_rev, reverse when the mirror holds, otherwise undetermined.  Reads and database are
// DUST-masked first (vsearch's defaults; k_dust above / dust_host in engine.hip) unless ITSX_QMASK=none.
namespace itsx {
static constexpr int OTAB = 16384;
__device__ __forceinline__ uint32_t rc24(uint32_t k)
{
  uint32_t r = __brev(~k & 0xffffffu) >> 8;
  return ((r >> 1) & 0x555555u) | ((r & 0x555555u) << 1);
}
__global__ __launch_bounds__(256) void k_orient(ReadsDev rd, const uint32_t *dbbits, const uint32_t *dmask, int8_t *strand, int32_t *cfwd, int32_t *crev)
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
    if (dmask) {                                               // vsearch --qmask dust: soft-masked symbols spoil the words that touch them
      const uint32_t *dm = dmask + rd.woff[r];
      for (int mw = tid; mw < ((L + 31) >> 5); mw += 256) {
        uint32_t x = dm[mw];
        while (x) {
          const int pos = mw * 32 + __ffs(x) - 1; x &= x - 1;
          for (int d = 0; d < 12; d++) { const int p = pos - d; if (p >= 0) atomicOr(&bad[p >> 5], 1u << (p & 31)); }
        }
      }
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
void launch_orient(const ReadsDev &rd, const uint32_t *dbbits, const uint32_t *dmask, int8_t *strand, int32_t *cfwd, int32_t *crev, hipStream_t st)
{
  if (rd.n <= 0) return;
  hipLaunchKernelGGL(k_orient, dim3((unsigned)std::min<int64_t>(rd.n, 65536)), dim3(256), 0, st, rd, dbbits, dmask, strand, cfwd, crev);
}
}  // namespace itsx
