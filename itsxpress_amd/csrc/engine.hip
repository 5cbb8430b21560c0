// engine.hip -- host orchestration and the C ABI (include/itsx_hip.h) of the gfx950 engine.
//
// The path it drives (reference: itsxpress/SeqSample.py:93-225 + 368-562, main.py:176-231):
//   reads (packed, resident in HBM) -> derep (k_derep) -> uniques ordered by length ->
//   MSV for every (unique, profile) (k_msv) -> survivor list grouped by profile ->
//   bias filter + Forward -> Backward + posterior decoding + regions -> envelope re-scoring
//   (k_float) -> scores, per-profile reported counts (domZ) -> domain thresholds ->
//   ItsPosition argmax -> per-read (start, stop, tlen).
// Everything numeric runs on the device; the host parses text, packs reads, sizes buffers,
// builds per-length constant tables with libm (as hmmsearch does on its host), and sorts rows
// for the file-compatible writers.  There is no CPU fallback for any stage.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "detmath.h"
#include "engine.h"
#include "k_api.h"
#include "fastq_io.h"

namespace itsx {
void launch_region_offsets(int64_t npairs, const PairRec *pairs, const int32_t *pref, const int64_t *seg_pair_start,
                           const int64_t *seg_region_start, int64_t *pair_region0, hipStream_t st);
void launch_finalize(itsx_domain *dom, int64_t n, const int64_t *domz, double domE, const int32_t *usample, int P, hipStream_t st);
void launch_positions(const itsx_domain *dom, int64_t n, const int8_t *side, unsigned long long *bl, unsigned long long *br,
                      int32_t *in_ddict, hipStream_t st);
void launch_position_coords(const itsx_domain *dom, int64_t n, const int8_t *side, const unsigned long long *bl, const unsigned long long *br,
                            int32_t *cl, int32_t *cr, hipStream_t st);
void launch_position_flags(const itsx_domain *dom, int64_t n, const int8_t *side, const unsigned long long *bl, const unsigned long long *br,
                           int32_t *uflag, hipStream_t st);
void launch_count_flags(const int32_t *uflag, int32_t U, const int32_t *uniq_of, int64_t n, unsigned long long *c, hipStream_t st);
void launch_compact_best(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, double lnp_certain, unsigned long long *bestc, hipStream_t st);
void launch_compact_mark(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, double lnp_certain, const unsigned long long *bestc, int32_t *keep, hipStream_t st);
void launch_compact_scatter(const itsx_domain *dom, int64_t n, const int32_t *keep, const int32_t *pos, itsx_domain *out, hipStream_t st);

// ---- a few tiny kernels that only the orchestration needs --------------------------------
__global__ void k_wave_rows_pairs(const WaveDesc *w, int nw, const PairRec *pairs, int32_t *rows)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nw) return;
  // (lengths ascend inside a segment -- piecewise, with prefix sharing: by (batch, depth) run -- so the longest need not be the last)
  int mx = 0;
  for (int k = 0; k < w[i].count; k++) mx = max(mx, pairs[w[i].first + k].L);
  rows[i] = mx + 1;
}
// The wave list of a whole chunk's pairs, made where it is used: wave i covers 64 consecutive pairs of one profile's segment, the
// profiles in launch order (`order`, waves before them in `woff`).  A million-read shard has 1.3 M waves: built on the host they were
// 43 MB up, 5 MB of row counts down and 43 MB up again -- 20 ms in which the GPU did nothing.  Also sums the waves' lane-rows.
__global__ void k_waves_build(int nw, int P, const int32_t *__restrict__ order, const int64_t *__restrict__ woff, const int64_t *__restrict__ seg_start,
                              const int32_t *__restrict__ total, const PairRec *__restrict__ pairs, WaveDesc *__restrict__ w, unsigned long long *__restrict__ lane_rows)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  if (i < nw) {
    int lo = 0, hi = P;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (woff[mid] <= (int64_t)i) lo = mid; else hi = mid; }
    const int p = order[lo];
    const int64_t k = ((int64_t)i - woff[lo]) * 64;
    WaveDesc d;
    d.prof = p; d.first = seg_start[p] + k; d.count = (int32_t)min((int64_t)64, (int64_t)total[p] - k); d.slab = 0;
    // (called on length-ordered segments today -- the unshared schedule -- but a share-ordered list must not under-size a wave silently:
    // the longest of the wave's pairs, not the last)
    int mx = 0;
    for (int q = 0; q < d.count; q++) { const int L = pairs[d.first + q].L; mx = L > mx ? L : mx; mine += (unsigned long long)L; }
    d.rows = mx + 1;
    d.pad = 0;
    w[i] = d;
  }
  for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(lane_rows, mine);
}
// merged read j out of the merge kernel's output (pair i's bases sit at foff[i] + roff[i]) into a gap-free text: one wave per read
__global__ void __launch_bounds__(256) k_gather_reads(const uint8_t *__restrict__ src, const int64_t *__restrict__ srcoff, const int64_t *__restrict__ dstoff, int64_t m,
                                                      uint8_t *__restrict__ dst)
{
  const int lane = threadIdx.x & 63;
  for (int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); j < m; j += (int64_t)gridDim.x * 4) {
    const int64_t so = srcoff[j], d0 = dstoff[j], len = dstoff[j + 1] - d0;
    for (int64_t k = lane; k < len; k += 64) dst[d0 + k] = src[so + k];
  }
}
__global__ void k_wave_rows_regions(const WaveDesc *w, int nw, const RegionRec *rg, int32_t *rows)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nw) return;
  int mx = 0;
  for (int k = 0; k < w[i].count; k++) { const RegionRec r = rg[w[i].first + k]; mx = max(mx, r.jenv - r.ienv + 1); }
  rows[i] = mx + 1;
}
__global__ void k_pair_counters(const PairOut *po, int64_t n, unsigned long long *c)
{
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const PairOut o = po[i];
    a0 += o.pass_bias; a1 += o.pass_fwd; a2 += o.pass_fwd ? o.nregions : 0; a3 += (o.flags & 2) ? 1 : 0; a4 += o.pass_fwd ? ((o.flags >> 8) & 0xff) : 0;
  }
  for (int d = 32; d >= 1; d >>= 1) { a0 += __shfl_down(a0, d, 64); a1 += __shfl_down(a1, d, 64); a2 += __shfl_down(a2, d, 64); a3 += __shfl_down(a3, d, 64); a4 += __shfl_down(a4, d, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&c[0], a0); atomicAdd(&c[1], a1); atomicAdd(&c[2], a2); atomicAdd(&c[3], a3); atomicAdd(&c[4], a4); }
}
__global__ void k_gather_i32(const int32_t *src, const int64_t *idx, int n, int32_t *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}
__global__ void k_rep_coords(int32_t U, const unsigned long long *bl, const unsigned long long *br, const int32_t *cl, const int32_t *cr,
                             const int32_t *seed_read, const int32_t *len, int32_t *start, int32_t *stop, int32_t *tlen)
{
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const unsigned long long l = bl[u], r = br[u];
  start[u] = l ? cl[u] : -1;                      // left.to_pos  (itsxpress/SeqSample.py:480)
  stop[u] = r ? cr[u] - 1 : -1;                   // right.from_pos - 1  (:484)
  tlen[u] = (l || r) ? len[seed_read[u]] : -1;
}
__global__ void k_read_coords(int64_t n, const int32_t *uniq_of, const int32_t *us, const int32_t *ue, const int32_t *ut, const int32_t *uind,
                              int32_t *start, int32_t *stop, int32_t *tlen, int32_t *ind)
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int u = uniq_of[r];
  if (u < 0) { start[r] = stop[r] = tlen[r] = -1; ind[r] = 0; return; }
  start[r] = us[u]; stop[r] = ue[u]; tlen[r] = ut[u]; ind[r] = uind[u];
}
__global__ void k_pack_coords4(int64_t n, const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d, int32_t *out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[i * 4] = a[i]; out[i * 4 + 1] = b[i]; out[i * 4 + 2] = c[i]; out[i * 4 + 3] = d[i]; }
}
// per unique: the orientation-free 128-bit key (two XXH64 seeds over the packed read, length folded in; the smaller of
// forward / reverse complement), the global index of its first occurrence, and whether the forward strand is the canonical one
__global__ void k_unique_keys128(int32_t U, const int32_t *seed_read, const int32_t *len, const uint64_t *f0, const uint64_t *r0, const uint64_t *f1,
                                 const uint64_t *r1, int64_t base, int64_t *out)
{
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const int32_t r = seed_read[u];
  const uint64_t lm = (uint64_t)len[r] * 0x9E3779B97F4A7C15ULL;
  const uint64_t kf0 = f0[r] ^ lm, kr0 = r0[r] ^ lm, kf1 = f1[r] ^ lm, kr1 = r1[r] ^ lm;
  const bool fwd_le = (kf0 < kr0) || (kf0 == kr0 && kf1 <= kr1);
  out[(int64_t)u * 4 + 0] = (int64_t)(fwd_le ? kf0 : kr0);
  out[(int64_t)u * 4 + 1] = (int64_t)(fwd_le ? kf1 : kr1);
  out[(int64_t)u * 4 + 2] = base + r;
  out[(int64_t)u * 4 + 3] = fwd_le ? 1 : 0;
}
__global__ void k_widen_i32(int64_t n, const int32_t *in, int64_t *out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
}  // namespace itsx

using namespace itsx;

static std::string g_create_error;

// device buffer that only ever grows: hipMalloc/hipFree of multi-GB buffers costs far more than the
// kernels that use them, so every work buffer is kept in the context between calls
// hipFree waits for the DEVICE to go idle.  With several contexts at work on one GPU (a streaming run: two chunks searched, a third
// being loaded, a fourth finalized) a buffer that grows in one context stalled its thread until every other context's kernels had
// finished -- 435 frees, 4.4 s of such waits in a 7.6-s pipeline.  Device memory that a context gives up is therefore kept on a
// list and returned in one go: when a context is destroyed, or when the list holds more than ITSX_DEFER_FREE_GB (16; 0 = free at
// once, as before).  Nothing is reused from the list, so a kernel still reading an old buffer reads valid memory.
static std::mutex g_grave_mu;
static std::vector<void *> g_grave;
static size_t g_grave_bytes = 0;
static void grave_flush()
{
  std::vector<void *> v;
  { std::lock_guard<std::mutex> g(g_grave_mu); v.swap(g_grave); g_grave_bytes = 0; }
  for (void *q : v) (void)hipFree(q);
}
static void defer_free(void *q, size_t bytes)
{
  static const size_t budget = (size_t)(sw_get("ITSX_DEFER_FREE_GB") ? std::max(0.0, atof(sw_get("ITSX_DEFER_FREE_GB"))) : 16.0) << 30;
  if (budget == 0) { (void)hipFree(q); return; }
  bool flush = false;
  { std::lock_guard<std::mutex> g(g_grave_mu); g_grave.push_back(q); g_grave_bytes += bytes; flush = g_grave_bytes > budget; }
  if (flush) grave_flush();
}

// The DP slab (pass A's saved states, then the rounds' rows) is the one buffer of tens of GB, and FRESH device memory costs 20-40 ms per GB
// (it is cleared): a streamed file's chunk contexts each brought their own (round 6: 9 x ~1 s).  A context that is done with its slab
// (itsx_release_scratch: its stream is idle) puts it here, and the next context's slab request takes a block of at least its size from
// here before it asks the driver.  Blocks leave the pool when the last context is destroyed.
struct PoolBlock { void *p; size_t bytes; int device; };
static std::mutex g_pool_mu;
static std::vector<PoolBlock> g_pool;
static int g_live_contexts = 0;
static void *pool_take(size_t bytes, int device, size_t *got)
{
  std::lock_guard<std::mutex> g(g_pool_mu);
  int best = -1;
  for (size_t i = 0; i < g_pool.size(); i++)
    if (g_pool[i].device == device && g_pool[i].bytes >= bytes && (best < 0 || g_pool[i].bytes < g_pool[(size_t)best].bytes)) best = (int)i;
  if (best < 0) return nullptr;
  void *q = g_pool[(size_t)best].p; *got = g_pool[(size_t)best].bytes;
  g_pool.erase(g_pool.begin() + best);
  return q;
}
static void pool_put(void *q, size_t bytes, int device)
{
  std::lock_guard<std::mutex> g(g_pool_mu);
  g_pool.push_back(PoolBlock{q, bytes, device});
}
static void pool_flush()
{
  std::vector<PoolBlock> v;
  { std::lock_guard<std::mutex> g(g_pool_mu); v.swap(g_pool); }
  for (auto &b : v) (void)hipFree(b.p);
}

// The labels of a read set: one blob + offsets instead of one std::string per record (10-20 M allocations per 10 M-read run, and
// as many moves when the parser's pieces are joined).  The part of std::vector<std::string>'s interface the engine uses.
struct NameList {
  std::vector<char> blob; std::vector<int64_t> off{0};
  size_t size() const { return off.size() - 1; }
  bool empty() const { return off.size() <= 1; }
  void clear() { blob.clear(); off.assign(1, 0); }
  void swap(NameList &o) { blob.swap(o.blob); off.swap(o.off); }
  const char *ptr(size_t i) const { return blob.data() + off[i]; }
  size_t len(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
  void emplace_back(const char *b, const char *e) { blob.insert(blob.end(), b, e); off.push_back((int64_t)blob.size()); }
  std::string operator[](size_t i) const { return std::string(ptr(i), len(i)); }
  int cmp(size_t a, size_t b) const          // strcmp's order (labels hold no NUL)
  {
    const size_t la = len(a), lb = len(b), m = la < lb ? la : lb;
    const int c = m ? memcmp(ptr(a), ptr(b), m) : 0;
    return c ? c : (la < lb ? -1 : la > lb ? 1 : 0);
  }
  void assign(const char *names, const int64_t *name_offsets, int64_t n)      // names[name_offsets[i] .. name_offsets[i + 1])
  {
    const int64_t o0 = name_offsets[0];
    blob.assign(names + o0, names + name_offsets[n]);
    off.resize((size_t)n + 1);
    for (int64_t i = 0; i <= n; i++) off[(size_t)i] = name_offsets[i] - o0;
  }
  NameList slice(size_t lo, size_t hi) const
  {
    NameList r;
    r.blob.assign(blob.begin() + off[lo], blob.begin() + off[hi]);
    r.off.resize(hi - lo + 1);
    for (size_t i = lo; i <= hi; i++) r.off[i - lo] = off[i] - off[lo];
    return r;
  }
};

template <class T> struct DBuf {
  T *p = nullptr; size_t n = 0, cap = 0;
  int pool_device = -1;       // >= 0: a request that must allocate looks in the slab pool first (the DP slab only)
  // exact: no growth headroom (the slabs: their size is a budget, not a data-dependent count)
  hipError_t alloc(size_t count, bool exact = false)
  {
    n = count;
    if (count <= cap && p) return hipSuccess;
    static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
    if (p) {
      const auto f0 = std::chrono::steady_clock::now();
      defer_free(p, cap * sizeof(T));
      if (trace) fprintf(stderr, "[itsx] free %.3f GB: %.1f ms\n", cap * sizeof(T) / 1073741824.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count());
    }
    p = nullptr; cap = 0;
    if (!count) return hipSuccess;
    if (pool_device >= 0) {
      size_t got = 0;
      if (void *q = pool_take(count * sizeof(T), pool_device, &got)) { p = (T *)q; cap = got / sizeof(T); return hipSuccess; }
    }
    const size_t want = exact ? count : count + count / 8 + 64;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (trace) fprintf(stderr, "[itsx] hipMalloc %.3f GB: %.1f ms\n", want * sizeof(T) / 1073741824.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) {               // no room for the growth headroom: the exact size, and the failed attempt's sticky error cleared
      (void)hipGetLastError();
      grave_flush();                     // (what was given up but not returned yet)
      e = hipMalloc((void **)&p, count * sizeof(T));
      if (e != hipSuccess) { p = nullptr; return e; }
      cap = count; return e;
    }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) defer_free(p, cap * sizeof(T)); p = nullptr; n = 0; cap = 0; }
  ~DBuf() { release(); }
};

struct itsx_ctx {
  int device = 0;
  hipStream_t st = nullptr, st2 = nullptr;   // st2: the bias filter of the next batch beside the decoder of this one
  // st_hi: a stream of the highest priority that the LOAD stage's entry points swap in for `st` (LoadPriority below).  Several contexts
  // share a GPU in a streaming run: a chunk's packing / hashing / grouping kernels are milliseconds of work that otherwise wait in line
  // behind the 20-ms launches of the chunks being searched (round 5: a chunk resident 0.9 s after its text was there, 0.3 s of it a
  // profile upload's synchronisation, 0.26 s a dereplication that takes 0.06 on an idle device)
  hipStream_t st_hi = nullptr;
  hipStream_t st3 = nullptr; hipEvent_t ev_s3a = nullptr, ev_s3b = nullptr;   // st3: every other batch of a shared MSV filter (their launch tails overlap)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  // the MSV filter of chunk c + 1 runs on st2 beside the domain stage of chunk c (latency-bound kernels that leave the vector
  // ALUs idle): ev_msv0/1 bracket it; msv_pre_u0 = the chunk it was launched for (-1: none)
  hipEvent_t ev_msv0 = nullptr, ev_msv1 = nullptr, ev_c = nullptr;
  int64_t msv_pre_u0 = -1, next_u0 = -1; int32_t next_U = 0;
  mutable std::string err;
  void set_error(const std::string &m) const { err = m; }
  itsx_stats stats{};

  // ---- profiles
  std::vector<HostProfile> profs;
  int P = 0, G = 0;
  DBuf<DevProfile> d_prof;
  DBuf<uint32_t> d_etab;
  DBuf<int32_t> d_pbias, d_ptec, d_ptbm;
  DBuf<float> d_rtab, d_entab;             // k_bwd_bound's table; the folded emission odds by node (round 6: two-sided sharing)
  DBuf<ChainRec> sh_chain, sh_bchain;      // the chains' records by processing position (Forward / Backward)
  DBuf<float> d_flogsum; DBuf<LogTab> d_logtab; DBuf<float> d_btab; bool bound_fold = false;   // (the bound kernel's table, and whether it is the folded one)
  std::vector<char> generic_q;          // per profile: needs the runtime-Q kernels

  // ---- reads
  int64_t N = 0;
  itsx_io::Text h_bases;                 // original text (rep.fa keeps the input's case); not zero-filled when it grows
  const char *bases_view = "";           // = h_bases.data(), or the caller's buffer after itsx_set_reads_view
  const uint8_t *dev_bases = nullptr;    // after itsx_set_reads_device: the caller's device buffer (bases_view is fetched on demand)
  static constexpr int NSTAGE = 3;       // pinned / device staging of the hand-over (pack_and_upload)
  void *stage_pin[NSTAGE] = {nullptr, nullptr, nullptr}; DBuf<uint8_t> stage_dev[NSTAGE]; hipEvent_t stage_ev[NSTAGE] = {nullptr, nullptr, nullptr}; size_t stage_cap = 0;
  DBuf<int64_t> w_pk_off; DBuf<int8_t> w_pk_lut; DBuf<int32_t> w_pk_excnt, w_pk_exstart, w_pk_tmp; DBuf<long long> w_pk_bad;
  std::vector<int64_t> h_off;
  NameList h_names;
  std::vector<int64_t> h_woff;           // word offset of each read (the words themselves and the exceptions live on the device only)
  std::vector<int32_t> h_len;
  DBuf<uint32_t> d_words, d_exc;
  DBuf<int64_t> d_woff, d_excoff;
  DBuf<int32_t> d_len;
  ReadsDev rd{};
  int Lmax = 0;
  // per-sample batching: S samples share the context; reads group only within their sample and
  // hmmsearch's domZ is kept per (sample, profile)
  int32_t S = 1, sel_sample = -1;
  std::vector<int32_t> h_sample, h_usample;          // per read / per unique; empty when S == 1
  DBuf<int32_t> d_sample, d_usample;
  const int32_t *dev_sample() const { return S > 1 ? d_sample.p : nullptr; }
  const int32_t *dev_usample() const { return S > 1 ? d_usample.p : nullptr; }
  int32_t usample(int64_t u) const { return S > 1 ? h_usample[(size_t)u] : 0; }
  double slab_gb = 0.0;                  // HBM budget for per-batch DP slabs

  // ---- derep
  bool have_derep = false;
  std::vector<int32_t> order_cache; bool order_cache_ok = false;      // cluster_order() of the current dereplication (uc.txt and rep.fa both ask)
  int derep_strand_both = 1;             // how the last itsx_derep / itsx_cluster matched (the cross-shard keys follow it)
  int32_t U = 0;
  int32_t U_active = 0;                  // uniques this context scores (all of them unless itsx_set_active_uniques narrowed it)
  DBuf<int32_t> d_rep_of, d_uniq_of, d_seed_read, d_abund, d_sorted_uniq, d_ulen;
  DBuf<int8_t> d_strand;
  std::vector<int32_t> h_rep_of, h_uniq_of, h_seed_read, h_abund, h_sorted_uniq;
  std::vector<int8_t> h_strand;
  DBuf<uint32_t> d_orient_db; bool have_orient_db = false;   // f4: the orientation database's 12-mer bitmap
  bool clustered = false;                // last grouping came from itsx_cluster at id < 1 (uc rows carry identities)
  std::vector<double> h_pct;             // per read: identity of its H row, -1 for centroids / dropped reads
  std::vector<int32_t> h_order;          // kept reads in processing (label) order

  // ---- search
  bool have_search = false, have_final = false;
  double T = 10.0;
  int64_t npairs_padded = 0;
  DBuf<PairRec> d_pairs;
  DBuf<PairOut> d_pout;
  DBuf<RegionRec> d_regions;
  DBuf<RegionOut> d_rout;
  DBuf<int64_t> d_pair_region0;
  DBuf<int32_t> d_upos;                  // region -> index of its (shared) result
  DBuf<RegionRec> d_ulist;               // distinct envelopes, grouped by profile
  std::vector<std::unique_ptr<DBuf<itsx_domain>>> dom_bufs;   // one segment per chunk of uniques
  std::vector<int64_t> dom_n;            // padded rows in each segment
  int64_t pair_budget = 0; int32_t trace_u0 = 0; int n_chunks = 0; bool keep_trace = false, trace_sorted = false;
  DBuf<int16_t> d_vtab; DBuf<VitOut> d_vit; bool have_vit = false; double F2 = 1e-6;     // Viterbi filter (F2 < F1 only)
  DBuf<int32_t> d_domz32;
  DBuf<int64_t> d_domz64; bool domz_on_device = false;     // itsx_domz_device: the caller reduces the counters where they are
  // ITSX_COMPACT_ROWS=1: a chunk's domain rows live in w_domscratch (reused by every chunk) and only the rows that can still
  // win the argmax move to dom_bufs[chunk] (k_compact_*): 80 B x ~2 rows per representative instead of x ~130
  bool compact_rows = false; double compact_zmax = 1e9, compact_dome_min = 1e-2, compact_lnp = 0; int compact_ncls = 0;
  DBuf<itsx_domain> w_domscratch; DBuf<int8_t> w_cls; DBuf<unsigned long long> w_bestc; DBuf<int32_t> w_keep, w_keeppos, w_keeptmp;
  int64_t rows_before_compaction = 0;
  // rows mode (itsx_set_rows_mode / ITSX_ROWS): 0 every domain row stays resident (domtbl.txt), 1 compact, 2 lazy (k_lazy.hip)
  int rows_mode = -1;                    // -1: from the environment at every search
  bool lazy = false;                     // this search runs the lazy domain stage
  bool domz_exchanged = false;           // the caller moved domZ between search and finalize (a multi-rank driver): finalize never re-runs on its own
  int64_t lazy_pending = 0;              // undecided rows that could change a result, after the last lazy finalize
  double sF1 = 1e-6, sF3 = 1e-6;         // the last search's thresholds (finalize may have to repeat it in full)
  std::vector<int64_t> domz_ub;          // [S][P] pairs past the MSV filter: an upper bound of hmmsearch's domZ
  std::vector<int64_t> domz_loc, domz_ub_loc;   // this context's own counters (domz / domz_ub hold what finalize uses: the same, or the ranks' sums)
  bool completing = false;               // itsx_lazy_complete: every pair of the profiles in plist goes through the domain pipeline
  std::vector<int32_t> h_plist; DBuf<int32_t> d_plist;
  int64_t s_Uc = 0; int s_Lcap = 0;       // the last search's chunk size and length cap (the completion walks the same chunks)
  DBuf<float> l_fb; DBuf<uint32_t> l_b10; DBuf<uint8_t> l_done; DBuf<int32_t> l_flag, l_pos, l_scan, l_has; DBuf<unsigned long long> l_gtop, l_sure;
  DBuf<PairRec> l_pairs; DBuf<int64_t> l_seg, l_zub; DBuf<int32_t> l_pflag; DBuf<uint8_t> l_uflag; bool partial_coords = false; bool kept_rows = false; std::vector<int32_t> lazy_pending_prof;
  std::vector<uint16_t> thr_memo; std::vector<char> thr_memo_have; double thr_memo_F1 = -1.0; int thr_memo_Ppad = 0;   // MSV thresholds by length, kept between searches
  DBuf<int32_t> w_coords4; DBuf<int64_t> w_keys128; DBuf<uint64_t> w_hf1, w_hr1;
  std::vector<int32_t> h_sorted_active;  // the length-sorted list the HMM stages walk (= h_sorted_uniq unless itsx_set_active_uniques narrowed it)
  DBuf<LenTables> d_lt;
  std::vector<int64_t> domz;
  std::vector<itsx_domain> h_dom;        // valid rows, domtblout order
  std::vector<itsx_pairtrace> h_trace;

  // ---- persistent work buffers (see DBuf)
  DBuf<uint64_t> w_hf, w_hr; DBuf<unsigned long long> w_keys; DBuf<int32_t> w_vals, w_is_seed, w_seed_rank, w_scan_tmp, w_hist, w_cursor, w_tmp2;
  DBuf<uint32_t> w_slot_of; DBuf<unsigned int> w_ncoll;
  DBuf<uint16_t> w_thr, w_res; DBuf<int32_t> w_tjb, w_cnt, w_total, w_rows, w_rcnt, w_rpref, w_scan2, w_b, w_rrows;
  DBuf<int64_t> w_seg_start, w_idx, w_rseg, w_dz, w_counters, w_useg;
  DBuf<int32_t> w_rrep, w_ruq, w_rurank;
  DBuf<WaveDesc> w_waves, w_rw; DBuf<RegionRec> w_raw; DBuf<float> w_slab, w_eslab;
  DBuf<int32_t> w_mrcnt, w_mroff, w_mrlen, w_mrloff, w_mrrows, w_mrulist, w_mrulist2, w_mru; DBuf<int64_t> w_mrrowoff; DBuf<MrRec> w_mr; DBuf<MrOut> w_mrout; DBuf<int64_t> w_n2off; DBuf<float> w_n2sc, w_mrslab;
  DBuf<uint8_t> w_mrscratch; DBuf<WaveDesc> w_mrwaves;
  // regions past a pair's MAXDOM slots (k_decode appends; ordered by (pair, k) afterwards) and the ensemble stage's overflow path
  DBuf<RegionRec> w_pool, w_pool_sorted; DBuf<int32_t> w_pool_k; DBuf<unsigned long long> w_pool_n, w_pool_key; int64_t pool_cap = 0;
  DBuf<int32_t> w_mrsel; DBuf<int64_t> w_mrselrow; DBuf<MrBig> w_mrbig; DBuf<uint8_t> w_mrarena; DBuf<int32_t> w_mrenv; DBuf<WaveDesc> w_mrbigwaves;
  DBuf<int32_t> w_cl, w_cr;
  DBuf<int8_t> w_side; DBuf<unsigned long long> w_bl, w_br; DBuf<int32_t> w_uind, w_us, w_ue, w_ut, w_rs, w_re, w_rt, w_ri, w_uflag;

  // ---- prefix sharing (k_share.hip), rebuilt by every itsx_search: the prefix tree of the active uniques by sorted position s (sh_*_s)
  // and by processing position k = (batch, depth, s) (sh_dev); a batch is a range of s whose saved row states fit the slot budget
  struct ShareBatch { int32_t k0, k1; int64_t node0, nnodes; int32_t nsplit; int64_t gnode0, gnnodes; };   // nsplit > 1: the profiles in that many ranges, one after the other; g*: the batch's saved Backward states
  bool share_on = false; int share_B = 32, share_logB = 5, share_maxd = 0;
  DBuf<unsigned long long> sh_tab, sh_mask_s, sh_mask, sh_counters; DBuf<uint8_t> sh_depth_s, sh_depth;
  DBuf<int32_t> sh_src, sh_parent_s, sh_parent, sh_nn_s, sh_nn, sh_node0_s, sh_node0, sh_order, sh_ulen, sh_uorder, sh_inv, sh_flag, sh_pos, sh_bstart, sh_cursor, sh_segk, sh_scan, sh_cuts;
  std::vector<ShareBatch> sh_batches; std::vector<int32_t> sh_segk_h;            // [batch][SHARE_SEGS]
  DBuf<uint4> sh_mgslots; size_t sh_mgslots_half = 0;        // the MSV filter's saved Backward states (two-sided sharing)
  DBuf<uint4> sh_mslots; size_t sh_mslots_half = 0, sh_fslots_half = 0; DBuf<uint32_t> sh_pass, sh_need; DBuf<int32_t> sh_real, sh_segflat, sh_segdepth, sh_wc, sh_woff; DBuf<int64_t> sh_bnd;
  DBuf<uint16_t> sh_res_chk; DBuf<float> sh_fb_chk;
  ShareDev sh_dev{};
  // two-sided sharing (round 6): the suffix tree by s (sh_r*_s), the joins, the Backward chains by backward position kb (sh_bdev), their
  // batches' first positions (sh_bsegk_h), the slots of the saved Backward states behind the Forward ones in the DP slab
  std::string switches_at_search;
  // the top-up round of itsx_search_finalize (round 6): the last search's pair list, while it is still in the buffers (one chunk, one sample)
  struct TopUp { bool valid = false; int64_t NP = 0; std::vector<int64_t> seg_start; std::vector<int32_t> total; int32_t u0 = 0, Uc = 0; } topup;
  DBuf<unsigned long long> l_zneed, l_zsplit, l_tcnt; DBuf<int32_t> l_slot, l_pslot; DBuf<uint32_t> l_cut;
  std::vector<unsigned long long> h_zneed;
  bool prev_lazy = false; int prev_P = 0; int64_t prev_U = 0, prev_Uc = 0;      // the last search's chunking (itsx_search)
  std::vector<int32_t> h_merge_index;      // per pair of the last merge-and-load: its merged read, -1 = not merged
  bool two_on = false; int share_maxrd = 0; int32_t Ub = 0; size_t sh_gslots_off = 0;
  DBuf<uint8_t> sh_rdepth_s, sh_rdepth; DBuf<unsigned long long> sh_rmask_s, sh_rmask, sh_keys, sh_keys2;
  DBuf<int32_t> sh_rparent_s, sh_rparent, sh_jlev_s, sh_jown_s, sh_endrow_s, sh_rsteps_s, sh_rnn_s, sh_rnode0_s, sh_endrow, sh_jlev, sh_jsrc, sh_jownb;
  DBuf<int32_t> sh_border_s, sh_invb, sh_bsegk, sh_rnn, sh_rnode0, sh_border, sh_bulen, sh_rsrc, sh_bsteps, sh_vals;
  DBuf<uint8_t> sh_sorttmp;
  std::vector<int32_t> sh_bsegk_h;
  BShareDev sh_bdev{};
  DBuf<uint32_t> sh_needb; DBuf<uint16_t> sh_resb; DBuf<int32_t> sh_cntb, sh_totalb, sh_realb, sh_bsegflat, sh_bsegdepth, sh_bwc, sh_bwoff; DBuf<int64_t> sh_bbnd, sh_bseg_start;
  DBuf<PairRec> sh_bpairs; DBuf<WaveDesc> sh_bwaves;
};

#define CTXCHK(c)                                   \
  if (!(c)) return ITSX_E_ARG;
#define SET_ERR(ctx, code, msg) do { (ctx)->set_error(msg); return (code); } while (0)
#undef HIPCHK
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) { ctx->set_error(std::string(#expr) + ": " + hipGetErrorString(e__)); return ITSX_E_DEVICE; } \
  } while (0)

template <class T> static hipError_t upload(DBuf<T> &d, const std::vector<T> &h, hipStream_t st, size_t min_elems = 1)
{
  hipError_t e = d.alloc(std::max(h.size(), min_elems));
  if (e != hipSuccess) return e;
  if (h.empty()) return hipSuccess;
  return hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
}

// kernel timers that do not stall the stream: event pairs are recorded around launches and read back once,
// after the whole stage has been enqueued
struct LazyTimers {
  struct Rec { hipEvent_t a, b; float *acc; };
  std::vector<Rec> recs; hipStream_t st;
  explicit LazyTimers(hipStream_t s) : st(s) {}
  size_t begin(float *acc) { Rec r; (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); r.acc = acc; (void)hipEventRecord(r.a, st); recs.push_back(r); return recs.size() - 1; }
  void end(size_t i) { (void)hipEventRecord(recs[i].b, st); }
  void collect()
  {
    for (auto &r : recs) { (void)hipEventSynchronize(r.b); float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b); *r.acc += ms; (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    recs.clear();
  }
  ~LazyTimers() { collect(); }
};

struct StageTimer {
  hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
  explicit StageTimer(hipStream_t s) : st(s) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
  float stop() { (void)hipEventRecord(b, st); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
  ~StageTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

// run fn(t) on T host threads (the domain table of a large sample has millions of rows: ordering and formatting it on one
// thread took longer than the whole GPU path)
template <class F> static void on_threads(int T, F fn)
{
  if (T <= 1) { fn(0); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)T);
  for (int t = 0; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
  for (auto &x : th) x.join();
}

extern "C" {

int itsx_abi_version(void) { return ITSX_ABI_VERSION; }

const char *itsx_last_error(const itsx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// the load stage's work on the high-priority stream: swapped in for the call, waited for and swapped back at its end (everything after
// the call -- the search on `st` -- sees the load stage's results complete)
struct LoadPriority {
  itsx_ctx *c; bool on;
  explicit LoadPriority(itsx_ctx *ctx) : c(ctx), on(ctx && ctx->st_hi != nullptr) { if (on) std::swap(c->st, c->st_hi); }
  ~LoadPriority() { if (on) { (void)hipStreamSynchronize(c->st); std::swap(c->st, c->st_hi); } }
  LoadPriority(const LoadPriority &) = delete;
  LoadPriority &operator=(const LoadPriority &) = delete;
};

itsx_ctx *itsx_create(int device_id, int flags)
{
  (void)flags;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) { g_create_error = "no HIP device is visible (the engine has no CPU fallback)"; return nullptr; }
  if (device_id < 0 || device_id >= ndev) { g_create_error = "device ordinal out of range"; return nullptr; }
  e = hipSetDevice(device_id);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return nullptr; }
  if (std::string(prop.gcnArchName).compare(0, 6, "gfx950") != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + "; this engine ships gfx950 code only";
    return nullptr;
  }
  itsx_ctx *ctx = new itsx_ctx();
  ctx->device = device_id;
  ctx->w_slab.pool_device = device_id;
  { std::lock_guard<std::mutex> g(g_pool_mu); g_live_contexts++; }
  if (hipStreamCreate(&ctx->st) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete ctx; return nullptr; }
  if (!(sw_get("ITSX_LOAD_PRIORITY") && atoi(sw_get("ITSX_LOAD_PRIORITY")) == 0)) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least ||
        hipStreamCreateWithPriority(&ctx->st_hi, hipStreamDefault, greatest) != hipSuccess) { (void)hipGetLastError(); ctx->st_hi = nullptr; }
  }
  // p7_FLogsum's table, built with libm exactly as hmmsearch builds it at start-up
  std::vector<float> tbl(16000);
  for (int i = 0; i < 16000; i++) tbl[i] = (float)log(1. + exp((double)-i / 1000.f));
  if (upload(ctx->d_flogsum, tbl, ctx->st) != hipSuccess) { g_create_error = "device allocation failed"; delete ctx; return nullptr; }
  std::vector<LogTab> lt(LOGTAB_N);
  build_logtab(lt.data());
  if (upload(ctx->d_logtab, lt, ctx->st) != hipSuccess) { g_create_error = "device allocation failed"; delete ctx; return nullptr; }
  (void)hipStreamSynchronize(ctx->st);
  return ctx;
}

void itsx_destroy(itsx_ctx *ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->st);
  if (ctx->st3) { (void)hipStreamSynchronize(ctx->st3); (void)hipStreamDestroy(ctx->st3); (void)hipEventDestroy(ctx->ev_s3a); (void)hipEventDestroy(ctx->ev_s3b); }
  if (ctx->st2) { (void)hipStreamSynchronize(ctx->st2); (void)hipStreamDestroy(ctx->st2); (void)hipEventDestroy(ctx->ev_a); (void)hipEventDestroy(ctx->ev_b);
                  if (ctx->ev_msv0) { (void)hipEventDestroy(ctx->ev_msv0); (void)hipEventDestroy(ctx->ev_msv1); (void)hipEventDestroy(ctx->ev_c); } }
  (void)hipStreamDestroy(ctx->st);
  if (ctx->st_hi) { (void)hipStreamSynchronize(ctx->st_hi); (void)hipStreamDestroy(ctx->st_hi); }
  for (int k = 0; k < itsx_ctx::NSTAGE; k++) { if (ctx->stage_pin[k]) (void)hipHostFree(ctx->stage_pin[k]); if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]); }
  delete ctx;
  grave_flush();                         // this context's buffers, and whatever the others have given up meanwhile
  bool last = false;
  { std::lock_guard<std::mutex> g(g_pool_mu); last = --g_live_contexts <= 0; }
  if (last) pool_flush();
}

// A context that will not search again soon (a streamed file's chunk after its search; results, row lists and everything a finalize or
// a completion needs stay) hands its DP slab to the next context on the device.  Safe: its streams are drained first.
int itsx_release_scratch(itsx_ctx *ctx)
{
  CTXCHK(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (ctx->st2) HIPCHK(hipStreamSynchronize(ctx->st2));
  if (ctx->st3) HIPCHK(hipStreamSynchronize(ctx->st3));
  if (ctx->w_slab.p) { pool_put(ctx->w_slab.p, ctx->w_slab.cap * sizeof(float), ctx->device); ctx->w_slab.p = nullptr; ctx->w_slab.n = ctx->w_slab.cap = 0; }
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ profiles
static int install_profiles(itsx_ctx *ctx, std::vector<HostProfile> &pv, int *n_profiles)
{
  for (auto &p : pv) {
    if (p.M > MMAX || p.Q > QMAX) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "model '" + p.name + "' has more than 46 nodes; the device kernels hold at most 46");
    if (p.base_b + p.bias_b >= 255) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "model '" + p.name + "': MSV bias too large");
  }
  ctx->thr_memo.clear(); ctx->thr_memo_have.clear(); ctx->thr_memo_F1 = -1.0;      // (the MSV thresholds kept between searches belong to the old profiles)
  HIPCHK(hipSetDevice(ctx->device));
  LoadPriority load_priority(ctx);
  ctx->profs = std::move(pv);
  const int P = (int)ctx->profs.size();
  ctx->P = P; ctx->G = (P + 63) / 64;
  const int Ppad = ctx->G * 64;
  std::vector<DevProfile> dp(std::max(P, 1));
  ctx->generic_q.assign(P, 0);
  std::vector<uint32_t> etab((size_t)std::max(P, 1) * 16 * MSV_TW, 0);
  std::vector<int32_t> pb(std::max(Ppad, 64), 0), pt(std::max(Ppad, 64), 0), pm(std::max(Ppad, 64), 0);
  for (size_t i = 0; i < etab.size(); i++) etab[i] = 0xFF01FF01u;   // (0 - 255) as int16, twice: cells past M never rise above 0
  for (int i = 0; i < P; i++) {
    const HostProfile &h = ctx->profs[i];
    DevProfile &d = dp[i];
    memset(&d, 0, sizeof(d));
    for (int q = 0; q < h.Q; q++) {
      for (int t = 0; t < 7; t++) for (int z = 0; z < 4; z++) d.tf[(q * 8 + t) * 4 + z] = h.tfv[((size_t)q * 7 + t) * 4 + z];
      for (int z = 0; z < 4; z++) d.tf[(q * 8 + 7) * 4 + z] = h.tfv[((size_t)7 * h.Q + q) * 4 + z];
    }
    for (int k = 1; k <= 4 * h.Q; k++)            // the same transitions by node (slot k = z Q + q + 1 of the striped vectors)
      for (int t = 0; t < 8; t++) d.tfn[k * 8 + t] = d.tf[(((k - 1) % h.Q) * 8 + t) * 4 + (k - 1) / h.Q];
    for (int q = 0; q < h.Q; q++) {
      const int order[3] = {6, 5, 0};      // II MI BM of this group
      for (int k = 0; k < 3; k++) for (int z = 0; z < 4; z++) d.tb[(q * 6 + k) * 4 + z] = d.tf[(q * 8 + order[k]) * 4 + z];
      for (int k = 0; k < 3; k++)          // MM IM DM of the next group; group Q-1 wraps to group 0 shifted left by one lane
        for (int z = 0; z < 4; z++) {
          float v;
          if (q + 1 < h.Q) v = d.tf[((q + 1) * 8 + 1 + k) * 4 + z];
          else v = (z < 3) ? d.tf[(0 * 8 + 1 + k) * 4 + z + 1] : 0.0f;
          d.tb[(q * 6 + 3 + k) * 4 + z] = v;
        }
    }
    for (int x = 0; x < NCODE; x++)
      for (int q = 0; q < h.Q; q++) for (int z = 0; z < 4; z++) d.rf[(x * QMAX + q) * 4 + z] = h.rfv[((size_t)x * h.Q + q) * 4 + z];
    for (int x = 0; x < NCODE; x++) { d.feo[x * 2] = h.feo[x][0]; d.feo[x * 2 + 1] = h.feo[x][1]; }
    d.ft10 = h.ft10; d.ft11 = h.ft11; d.fpi0 = h.fpi0; d.fpi1 = h.fpi1;
    for (int k = 0; k < 6; k++) d.ev[k] = h.evparam[k];
    d.M = h.M; d.Q = h.Q;
    ctx->generic_q[i] = (h.Q != QMAX);
    for (int x = 0; x < NCODE; x++)
      for (int r = 0; r < MSV_REGS; r++) {
        uint32_t packed = 0;
        for (int half = 0; half < 2; half++) {
          const int k = half * MSV_REGS + r + 1;          // striped: register r = cells r and r + 23 (k_msv.hip)
          const int cost = (k <= h.M) ? h.rbv[(size_t)x * (h.M + 1) + k] : 255;
          const int16_t e = (int16_t)(h.bias_b - cost);
          packed |= (uint32_t)(uint16_t)e << (16 * half);
        }
        etab[((size_t)i * 16 + x) * MSV_TW + r] = packed;
      }
    pb[i] = h.bias_b; pt[i] = h.tec_b; pm[i] = h.tbm_b;
  }
  {   // Viterbi-filter word tables: per profile [8][47] transitions, [16][47] emissions
    std::vector<int16_t> vt((size_t)std::max(P, 1) * VIT_TAB, (int16_t)-32768);
    for (int i = 0; i < P; i++) {
      const HostProfile &h = ctx->profs[i];
      int16_t *t = vt.data() + (size_t)i * VIT_TAB;
      for (int k8 = 0; k8 < 8; k8++) for (int k = 0; k <= h.M; k++) t[k8 * (MMAX + 1) + k] = h.tww[(size_t)k8 * (h.M + 1) + k];
      for (int x = 0; x < NCODE; x++) for (int k = 0; k <= h.M; k++) t[(8 + x) * (MMAX + 1) + k] = h.rww[(size_t)x * (h.M + 1) + k];
    }
    HIPCHK(upload(ctx->d_vtab, vt, ctx->st));
  }
  {   // the bound kernel's transitions by pair of nodes (k_lazy.hip: k_fwd_bound): record j = pair j's early half, pair j - 1's late half
    // FOLD (k_lazy.hip): with H_k = 1 + t(D_k -> D_k+1) H_k+1 (what one unit of D_k adds to the row's E through itself and the deletes
    // after it), g_k = 1 + t(M_k -> D_k+1) H_k+1, s_1 = 1, s_k+1 = t(M_k -> D_k+1) / g_k, a_k = s_k t(D_k -> D_k+1) / s_k+1 the kernel
    // keeps M~_k = M_k g_k and D^_k = D_k / s_k:  D^_k+1 = D^_k a_k + M~_k,  E = sum of M~_k;  M -> M / I transitions divided by g_k, the D -> M
    // ones multiplied by s_k, the emission odds multiplied by g_k (in the kernel, from the table's last 48 floats); the insert cells likewise
    // divided by r_k = t(M_k -> I_k) / g_k (I^_k' = I^_k t(I_k -> I_k) + M~_k; the I -> M transitions multiplied by r_k).  Needs
    // t(M_k -> D_k+1) > 0 wherever the delete path goes on (every node hmmbuild writes; after the last node both are 0 and a_k = 0 ends the
    // chain); one profile without that and the whole set runs the plain recurrences (ITSX_BOUND_FOLD=0 does too).
    const int KK = 2 * BOUND_PAIRS;
    std::vector<std::vector<double>> G((size_t)P), SC((size_t)P), AA((size_t)P), RI((size_t)P);
    bool fold = !(sw_get("ITSX_BOUND_FOLD") && atoi(sw_get("ITSX_BOUND_FOLD")) == 0);
    for (int i = 0; i < P && fold; i++) {
      const DevProfile &d = dp[i];
      const int K = 4 * ctx->profs[i].Q;
      auto tn = [&](int k, int t) { return (k >= 1 && k <= K) ? (double)d.tfn[k * 8 + t] : 0.0; };
      std::vector<double> H((size_t)KK + 3, 1.0), &g = G[(size_t)i], &sc = SC[(size_t)i], &aa = AA[(size_t)i], &ri = RI[(size_t)i];
      g.assign((size_t)KK + 2, 1.0); sc.assign((size_t)KK + 3, 1.0); aa.assign((size_t)KK + 2, 0.0); ri.assign((size_t)KK + 2, 1.0);
      for (int k = KK; k >= 1; k--) H[(size_t)k] = 1.0 + tn(k, 7) * H[(size_t)k + 1];
      for (int k = 1; k <= KK; k++) g[(size_t)k] = 1.0 + tn(k, 4) * H[(size_t)k + 1];
      for (int k = 1; k <= KK; k++) {
        const double md = tn(k, 4), dd = tn(k, 7);
        if (md > 0.0) { sc[(size_t)k + 1] = md / g[(size_t)k]; aa[(size_t)k] = sc[(size_t)k] * dd / sc[(size_t)k + 1]; }
        else if (dd > 0.0 && k > 1) fold = false;          // a delete path that goes on past a node no match cell feeds: not foldable
        else { sc[(size_t)k + 1] = 1.0; aa[(size_t)k] = 0.0; }
        // (rescaling lets a row's cells reach ~1e21 -- 1e29 under ITSX_BOUND_RESCALE_EXP=28 -- before they are scaled back: the scaled cells
        // must stay far below FLT_MAX, 3.4e38)
        if (!(sc[(size_t)k + 1] > 1e-6 && sc[(size_t)k + 1] < 1e6 && aa[(size_t)k] < 1e6)) fold = false;
        // the insert cells by r_k = t(M_k -> I_k) / g_k: I^_k' = I^_k t(I_k -> I_k) + M~_k.  A node without M -> I must not use its insert cell at all
        const double mi = tn(k, 5);
        if (mi > 0.0) ri[(size_t)k] = mi / g[(size_t)k];
        else if (tn(k, 6) > 0.0 || tn(k + 1, 2) > 0.0) fold = false;
        else ri[(size_t)k] = 1.0;
        if (!(ri[(size_t)k] > 1e-6)) fold = false;
      }
    }
    ctx->bound_fold = fold;
    std::vector<float> bt((size_t)std::max(P, 1) * BOUND_TAB, 0.0f);
    for (int i = 0; i < P; i++) {
      const DevProfile &d = dp[i];
      const int K = 4 * ctx->profs[i].Q;                                      // slots of the striped layout (transitions beyond them: 0)
      auto tn = [&](int k, int t) { return (k >= 1 && k <= K) ? d.tfn[k * 8 + t] : 0.0f; };
      auto gk = [&](int k) { return fold && k >= 1 && k <= KK ? G[(size_t)i][(size_t)k] : 1.0; };
      auto sk = [&](int k) { return fold && k >= 1 && k <= KK + 1 ? SC[(size_t)i][(size_t)k] : 1.0; };
      auto rk = [&](int k) { return fold && k >= 1 && k <= KK ? RI[(size_t)i][(size_t)k] : 1.0; };
      for (int j = 0; j <= BOUND_PAIRS; j++) {
        float *o = bt.data() + (size_t)i * BOUND_TAB + (size_t)j * 16;
        if (j < BOUND_PAIRS) {
          const int k1 = 2 * j + 1, k2 = 2 * j + 2;
          o[0] = (float)(tn(k1 + 1, 1) / gk(k1)); o[1] = (float)(tn(k2 + 1, 1) / gk(k2));      // M_k -> M_k+1 (tf stores it at the target node)
          o[2] = (float)(tn(k1 + 1, 2) * rk(k1)); o[3] = (float)(tn(k2 + 1, 2) * rk(k2));      // I_k -> M_k+1
          o[4] = (float)(tn(k1 + 1, 3) * sk(k1)); o[5] = (float)(tn(k2 + 1, 3) * sk(k2));      // D_k -> M_k+1
          o[6] = tn(k1, 6); o[7] = tn(k2, 6);                                                    // I_k -> I_k
          o[12] = (float)(tn(k1, 5) / gk(k1)); o[13] = (float)(tn(k2, 5) / gk(k2));            // M_k -> I_k (FOLD: not read)
        }
        if (j > 0) {
          const int k1 = 2 * (j - 1) + 1, k2 = 2 * (j - 1) + 2;
          o[8] = tn(k1, 0); o[9] = tn(k2, 0);              // B -> M_k
          if (fold) { o[10] = k1 > 1 ? (float)AA[(size_t)i][(size_t)k1 - 1] : 0.0f; o[11] = (float)AA[(size_t)i][(size_t)k1]; }
          else { o[10] = tn(k1 - 1, 7); o[11] = tn(k1, 7); }   // D_k1-1 -> D_k1 (0 before node 1), D_k1 -> D_k2
          o[14] = tn(k1, 4); o[15] = tn(k2, 4);            // M_k -> D_k+1 (FOLD: not read)
        }
      }
      float *gq = bt.data() + (size_t)i * BOUND_TAB + (size_t)(BOUND_PAIRS + 1) * 16;
      for (int k0 = 0; k0 < KK; k0++) gq[k0] = (float)gk(k0 + 1);
    }
    HIPCHK(upload(ctx->d_btab, bt, ctx->st));
    // k_bwd_bound (round 6) walks the same recurrences transposed: per pair of nodes (k1, k2) the folded M_k -> M_k+1, I_k -> M_k+1,
    // D_k -> M_k+1, I_k -> I_k, B -> M_k and D_k -> D_k+1 constants of the table above, by SOURCE node
    std::vector<float> rt((size_t)std::max(P, 1) * BOUND_RTAB, 0.0f);
    for (int i = 0; i < P && fold; i++) {
      const float *b = bt.data() + (size_t)i * BOUND_TAB;
      // record j = what the kernel's step j reads: mm im dm bm of pair j (the late half of pair j: k_lazy.hip) and ii, dd of pair j - 1
      // (the early half of the pair after it in walking order); the last pair's ii, dd stand behind the records
      for (int j = 0; j <= BOUND_PAIRS; j++) {
        float *o = rt.data() + (size_t)i * BOUND_RTAB + (size_t)j * 12;
        if (j < BOUND_PAIRS) {
          for (int q = 0; q < 6; q++) o[q] = b[(size_t)j * 16 + q];                        // mm im dm of the pair's nodes
          o[6] = b[(size_t)(j + 1) * 16 + 8]; o[7] = b[(size_t)(j + 1) * 16 + 9];          // B -> M_k1, B -> M_k2
        }
        if (j >= 1) {
          const int jj = j - 1;
          float *q = j < BOUND_PAIRS ? o + 8 : o;
          q[0] = b[(size_t)jj * 16 + 6]; q[1] = b[(size_t)jj * 16 + 7];                    // I_k -> I_k of pair j - 1
          q[2] = b[(size_t)(jj + 1) * 16 + 11];                                            // D_k1 -> D_k2
          q[3] = (jj + 2 <= BOUND_PAIRS) ? b[(size_t)(jj + 2) * 16 + 10] : 0.0f;           // D_k2 -> D_k2+1 (none after the last node)
        }
      }
    }
    HIPCHK(upload(ctx->d_rtab, rt, ctx->st));
    // the folded emission odds by node, [code][k - 1]: what both kernels keep in LDS
    std::vector<float> et((size_t)std::max(P, 1) * NCODE * KK, 0.0f);
    for (int i = 0; i < P && fold; i++) {
      const DevProfile &d = dp[i];
      const int Q = ctx->profs[i].Q;
      const float *gq = bt.data() + (size_t)i * BOUND_TAB + (size_t)(BOUND_PAIRS + 1) * 16;
      for (int x = 0; x < NCODE; x++)
        for (int k0 = 0; k0 < KK; k0++) {
          const float e = (k0 < 4 * Q) ? d.rf[(x * QMAX + (k0 % Q)) * 4 + k0 / Q] : 0.0f;
          et[((size_t)i * NCODE + x) * KK + k0] = e * gq[k0];
        }
    }
    HIPCHK(upload(ctx->d_entab, et, ctx->st));
  }
  HIPCHK(upload(ctx->d_prof, dp, ctx->st));
  HIPCHK(upload(ctx->d_etab, etab, ctx->st));
  HIPCHK(upload(ctx->d_pbias, pb, ctx->st));
  HIPCHK(upload(ctx->d_ptec, pt, ctx->st));
  HIPCHK(upload(ctx->d_ptbm, pm, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->stats.n_profiles = P;
  ctx->have_search = ctx->have_final = false;
  if (n_profiles) *n_profiles = P;
  return ITSX_OK;
}

int itsx_load_profiles_mem(itsx_ctx *ctx, const char *text, int64_t len, int *n_profiles)
{
  CTXCHK(ctx && text && len >= 0);
  std::vector<HostProfile> pv;
  const std::string e = parse_hmm_text(text, len, pv);
  if (!e.empty()) SET_ERR(ctx, ITSX_E_FORMAT, e);
  return install_profiles(ctx, pv, n_profiles);
}

// whole file, decompressed by its magic bytes (fastq_io.h); read files stay in the text cache for the writers
static std::shared_ptr<const itsx_io::Text> slurp(const char *path, bool cacheable, std::string &err)
{
  return itsx_io::read_text(path, err, cacheable);
}

int itsx_load_profiles_file(itsx_ctx *ctx, const char *hmm_path, int *n_profiles)
{
  CTXCHK(ctx && hmm_path);
  std::string err;
  const auto tp = slurp(hmm_path, false, err);
  if (!tp) SET_ERR(ctx, ITSX_E_IO, err);
  return itsx_load_profiles_mem(ctx, tp->data(), (int64_t)tp->size(), n_profiles);
}

int itsx_profile_name(const itsx_ctx *ctx, int i, char *buf, int buflen)
{
  CTXCHK(ctx && buf && buflen > 0 && i >= 0 && i < ctx->P);
  snprintf(buf, (size_t)buflen, "%s", ctx->profs[i].name.c_str());
  return ITSX_OK;
}

int itsx_profile_tables(const itsx_ctx *ctx, int i, uint8_t *rbv, float *rfv, float *tfv, int32_t *params6)
{
  CTXCHK(ctx && i >= 0 && i < ctx->P);
  const HostProfile &h = ctx->profs[i];
  if (rbv) memcpy(rbv, h.rbv.data(), h.rbv.size());
  if (rfv) memcpy(rfv, h.rfv.data(), h.rfv.size() * sizeof(float));
  if (tfv) memcpy(tfv, h.tfv.data(), h.tfv.size() * sizeof(float));
  if (params6) { params6[0] = h.M; params6[1] = h.Q; params6[2] = h.base_b; params6[3] = h.bias_b; params6[4] = h.tbm_b; params6[5] = h.tec_b; }
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ reads
static int8_t g_code[256];
static bool g_code_init = false;
static void init_codes()
{
  if (g_code_init) return;
  memset(g_code, -1, sizeof(g_code));
  const char *sym = "ACGT-RYMKSWHBVDN";
  for (int i = 0; i < 16; i++) { g_code[(unsigned char)sym[i]] = (int8_t)i; g_code[(unsigned char)(sym[i] | 0x20)] = (int8_t)i; }
  g_code[(unsigned char)'-'] = -1;
  g_code[(unsigned char)'U'] = g_code[(unsigned char)'u'] = 3;
  g_code[(unsigned char)'X'] = g_code[(unsigned char)'x'] = 15;
  g_code_init = true;
}

// Hand-over of the reads: the ASCII bases travel in chunks of whole reads through three pinned staging buffers (a pool of
// host threads copies chunk c+1 into its buffer while chunk c is on the bus and chunk c-1 is being packed), and are packed on
// the device chunk by chunk (k_util.hip).  A pageable hipMemcpy of the whole text moved 4.4 GB (10 M merged reads) at ~5 GB/s.
static int pack_and_upload(itsx_ctx *ctx, const char *view = nullptr, const uint8_t *dev_raw = nullptr)
{
  init_codes();
  const auto tp0 = std::chrono::steady_clock::now();
  const int64_t n = ctx->N;
  ctx->dev_bases = dev_raw;
  ctx->bases_view = dev_raw ? nullptr : (view ? view : ctx->h_bases.data());
  const char *bases = ctx->bases_view;
  ctx->h_len.resize((size_t)n); ctx->h_woff.resize((size_t)n + 1);
  ctx->h_woff[0] = 0;
  int Lmax = 0;
  {   // lengths, word offsets (a running sum: per thread over its share of the reads, then shifted by the shares before it), longest read
    const int T = n >= (1 << 20) ? std::max(1, std::min(8, itsx_io::io_threads())) : 1;
    std::vector<int64_t> wsum((size_t)T + 1, 0); std::vector<int> lmx((size_t)T, 0), bad((size_t)T, 0);
    on_threads(T, [&](int t) {
      const int64_t lo = n * t / T, hi = n * (t + 1) / T;
      int64_t w = 0; int mx = 0, b = 0;
      for (int64_t r = lo; r < hi; r++) {
        const int64_t L = ctx->h_off[r + 1] - ctx->h_off[r];
        if (L < 0) b |= 1;
        if (L > 65535) b |= 2;
        ctx->h_len[r] = (int32_t)L;
        mx = std::max(mx, (int)std::min<int64_t>(L, 1 << 30));
        w += std::max<int64_t>(1, (L + 15) / 16);
        ctx->h_woff[r + 1] = w;
      }
      wsum[(size_t)t + 1] = w; lmx[(size_t)t] = mx; bad[(size_t)t] = b;
    });
    int b = 0;
    for (int t = 0; t < T; t++) { wsum[(size_t)t + 1] += wsum[(size_t)t]; Lmax = std::max(Lmax, lmx[(size_t)t]); b |= bad[(size_t)t]; }
    if (b & 1) SET_ERR(ctx, ITSX_E_ARG, "read offsets must be non-decreasing");
    if (b & 2) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "reads longer than 65535 bases are not supported");
    if (T > 1)
      on_threads(T, [&](int t) {
        if (t == 0) return;
        const int64_t lo = n * t / T, hi = n * (t + 1) / T, add = wsum[(size_t)t];
        for (int64_t r = lo; r < hi; r++) ctx->h_woff[r + 1] += add;
      });
  }
  ctx->Lmax = Lmax;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->st;
  const int64_t nb = n > 0 ? ctx->h_off[n] : 0;
  int64_t CH = 64ll << 20;
  if (const char *e = sw_get("ITSX_PACK_CHUNK")) CH = std::max<int64_t>(70000, atoll(e));       // bytes; a read has at most 65535
  CH = std::min<int64_t>(CH, std::max<int64_t>(nb, 70000));
  constexpr int K = itsx_ctx::NSTAGE;
  if (!dev_raw && (int64_t)ctx->stage_cap < CH) {
    for (int k = 0; k < K; k++) {
      if (ctx->stage_pin[k]) { (void)hipHostFree(ctx->stage_pin[k]); ctx->stage_pin[k] = nullptr; }
      HIPCHK(hipHostMalloc(&ctx->stage_pin[k], (size_t)CH, hipHostMallocDefault));
      HIPCHK(ctx->stage_dev[k].alloc((size_t)CH + 16, true));
      if (!ctx->stage_ev[k]) HIPCHK(hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    ctx->stage_cap = (size_t)CH;
  }
  DBuf<int64_t> &d_off = ctx->w_pk_off; DBuf<int8_t> &d_lut = ctx->w_pk_lut; DBuf<int32_t> &d_excnt = ctx->w_pk_excnt, &d_exstart = ctx->w_pk_exstart, &d_tmp = ctx->w_pk_tmp;
  DBuf<long long> &d_bad = ctx->w_pk_bad;
  HIPCHK(d_off.alloc((size_t)n + 1)); HIPCHK(d_lut.alloc(256)); HIPCHK(d_excnt.alloc((size_t)n + 2)); HIPCHK(d_exstart.alloc((size_t)n + 2));
  HIPCHK(d_tmp.alloc((size_t)scan_tmp_elems(n + 2))); HIPCHK(d_bad.alloc(4));
  HIPCHK(ctx->d_words.alloc((size_t)ctx->h_woff[n] + 1)); HIPCHK(ctx->d_woff.alloc((size_t)n + 1)); HIPCHK(ctx->d_excoff.alloc((size_t)n + 1)); HIPCHK(ctx->d_len.alloc((size_t)n + 1));
  HIPCHK(hipMemcpyAsync(d_off.p, ctx->h_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->d_woff.p, ctx->h_woff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(ctx->d_len.p, ctx->h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_lut.p, g_code, 256, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_excnt.p, 0, ((size_t)n + 2) * 4, st));
  // the exception list is sized by a guess; a read set with more non-ACGT symbols than that repeats the exception pass
  int64_t ecap = std::max<int64_t>((int64_t)ctx->d_exc.cap, nb / 256 + 4096);
  if (const char *e = sw_get("ITSX_PACK_ECAP")) ecap = std::max<int64_t>(1, atoll(e));
  HIPCHK(ctx->d_exc.alloc((size_t)ecap));
  int copy_threads = std::max(1, std::min(8, itsx_io::io_threads()));
  long long hb[4] = {0, 0, 0, 0};                        // exceptions so far, list overflow, first illegal read, -
  for (int pass = 0; pass < 2; pass++) {                 // pass 1 only when the exception list was too small
    const long long init[4] = {0, 0, 0x7f7f7f7f7f7f7f7fll, 0};
    HIPCHK(hipMemcpyAsync(d_bad.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    int64_t r0 = 0; int c = 0;
    if (dev_raw && n > 0) {                              // the text is in device memory already: one chunk, no staging
      if (pass == 0) launch_pack(dev_raw, 0, d_off.p, ctx->d_woff.p, 0, n, d_lut.p, ctx->d_words.p, d_excnt.p, d_bad.p + 2, st);
      launch_exclusive_scan(d_excnt.p, d_exstart.p, n + 1, d_tmp.p, st);
      launch_pack_exc(dev_raw, 0, d_off.p, 0, n, d_lut.p, d_excnt.p, d_exstart.p, d_bad.p, ecap, ctx->d_excoff.p, ctx->d_exc.p, st);
      r0 = n;
    }
    while (r0 < n) {
      // reads [r0, r1): as many whole reads as fit the staging buffer
      const int64_t b0 = ctx->h_off[r0];
      int64_t r1 = (int64_t)(std::upper_bound(ctx->h_off.begin() + r0, ctx->h_off.begin() + n + 1, b0 + CH) - ctx->h_off.begin()) - 1;
      r1 = std::min<int64_t>(n, std::max<int64_t>(r1, r0 + 1));
      const int64_t bytes = ctx->h_off[r1] - b0;
      const int k = c % K;
      if (c >= K) HIPCHK(hipEventSynchronize(ctx->stage_ev[k]));
      if (bytes > 0) {
        char *dst = (char *)ctx->stage_pin[k];
        const int T = bytes >= (4 << 20) ? copy_threads : 1;
        on_threads(T, [&](int t) { const int64_t lo = bytes * t / T, hi = bytes * (t + 1) / T; memcpy(dst + lo, bases + b0 + lo, (size_t)(hi - lo)); });
        HIPCHK(hipMemcpyAsync(ctx->stage_dev[k].p, dst, (size_t)bytes, hipMemcpyHostToDevice, st));
      }
      if (pass == 0) launch_pack(ctx->stage_dev[k].p, b0, d_off.p, ctx->d_woff.p, r0, r1, d_lut.p, ctx->d_words.p, d_excnt.p, d_bad.p + 2, st);
      launch_exclusive_scan(d_excnt.p + r0, d_exstart.p + r0, r1 - r0 + 1, d_tmp.p, st);
      launch_pack_exc(ctx->stage_dev[k].p, b0, d_off.p, r0, r1, d_lut.p, d_excnt.p, d_exstart.p, d_bad.p, ecap, ctx->d_excoff.p, ctx->d_exc.p, st);
      HIPCHK(hipEventRecord(ctx->stage_ev[k], st));
      r0 = r1; c++;
    }
    if (n == 0) HIPCHK(hipMemsetAsync(ctx->d_excoff.p, 0, 8, st));
    HIPCHK(hipMemcpyAsync(hb, d_bad.p, sizeof(hb), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (hb[2] >= 0 && hb[2] < n) SET_ERR(ctx, ITSX_E_FORMAT, "read " + std::to_string(hb[2]) + " contains a symbol outside the IUPAC DNA alphabet");
    if (!hb[1]) break;
    if (pass == 1) SET_ERR(ctx, ITSX_E_DEVICE, "exception list overflow after resizing");
    ecap = hb[0] + 1;
    HIPCHK(ctx->d_exc.alloc((size_t)ecap));
  }
  ctx->rd.words = ctx->d_words.p; ctx->rd.woff = ctx->d_woff.p; ctx->rd.len = ctx->d_len.p;
  ctx->rd.excoff = ctx->d_excoff.p; ctx->rd.exc = ctx->d_exc.p; ctx->rd.n = n;
  ctx->have_derep = ctx->have_search = ctx->have_final = false;
  ctx->stats.n_reads = n;
  ctx->order_cache_ok = false;
  ctx->S = 1; ctx->sel_sample = -1; ctx->h_sample.clear(); ctx->h_usample.clear();      // a new read set is one sample until told otherwise
  ctx->stats.ms_pack = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
  return ITSX_OK;
}

// the caller's offsets, rebased to 0 (10 M reads: 80 MB -- copied and shifted in one pass by a few threads)
static int64_t take_offsets(itsx_ctx *ctx, const int64_t *offsets, int64_t n)
{
  const int64_t base0 = offsets[0];
  ctx->h_off.resize((size_t)n + 1);
  int64_t *dst = ctx->h_off.data();
  const int T = n >= (1 << 20) ? std::max(1, std::min(8, itsx_io::io_threads())) : 1;
  on_threads(T, [&](int t) { for (int64_t r = (n + 1) * t / T, hi = (n + 1) * (t + 1) / T; r < hi; r++) dst[r] = offsets[r] - base0; });
  return base0;
}
static int set_reads_impl(itsx_ctx *ctx, const char *bases, const int64_t *offsets, int64_t n, const char *names, const int64_t *name_offsets, bool borrow)
{
  CTXCHK(ctx && offsets && n >= 0 && (bases || offsets[n] == offsets[0]));
  if (n >= (1ll << 31) - 64) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 reads in one context");
  ctx->N = n;
  const int64_t base0 = take_offsets(ctx, offsets, n);
  const char *view = nullptr;
  if (borrow) { itsx_io::Text().swap(ctx->h_bases); view = bases ? bases + base0 : ""; }
  else if (bases) ctx->h_bases.assign(bases + base0, (size_t)(offsets[n] - base0));
  else ctx->h_bases.clear();
  ctx->h_names.clear();
  if (names && name_offsets) {
    ctx->h_names.assign(names, name_offsets, n);
  }
  return pack_and_upload(ctx, view);
}
int itsx_set_reads(itsx_ctx *ctx, const char *bases, const int64_t *offsets, int64_t n, const char *names, const int64_t *name_offsets)
{ return set_reads_impl(ctx, bases, offsets, n, names, name_offsets, false); }
int itsx_set_reads_view(itsx_ctx *ctx, const char *bases, const int64_t *offsets, int64_t n, const char *names, const int64_t *name_offsets)
{ return set_reads_impl(ctx, bases, offsets, n, names, name_offsets, true); }
// the text already lives in device memory (the output of a device-side parser or of the merge kernel): packed where it is
int itsx_set_reads_device(itsx_ctx *ctx, const void *d_bases, const int64_t *offsets, int64_t n, const char *names, const int64_t *name_offsets)
{
  CTXCHK(ctx && offsets && n >= 0 && (d_bases || offsets[n] == offsets[0]));
  if (n >= (1ll << 31) - 64) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 reads in one context");
  ctx->N = n;
  const int64_t base0 = take_offsets(ctx, offsets, n);
  itsx_io::Text().swap(ctx->h_bases);
  ctx->h_names.clear();
  if (names && name_offsets) {
    ctx->h_names.assign(names, name_offsets, n);
  }
  static const uint8_t none = 0;
  return pack_and_upload(ctx, nullptr, d_bases ? (const uint8_t *)d_bases + base0 : &none);
}

// ---- FASTA / FASTQ text -> records (labels up to the first blank, as vsearch labels them).  Large texts are cut at record
// starts and parsed by the I/O pool; the pieces are joined in order, so the result is the serial parser's.
struct FastxPart {
  itsx_io::Text seq, qual; std::vector<int64_t> off{0}; NameList ids;
  int rc = ITSX_OK; std::string err;
};

static void parse_fastx_range(const char *s, const char *end, bool want_qual, bool upper, FastxPart &out)
{
  auto next_line = [&](const char *&b, const char *&e) -> bool {
    if (s >= end) return false;
    b = s; const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
    e = nl ? nl : end; s = nl ? nl + 1 : end;
    if (e > b && e[-1] == '\r') e--;
    return true;
  };
  auto put_seq = [&](const char *b, const char *e) {
    const size_t o = out.seq.size();
    out.seq.append(b, e);
    if (upper) { char *q = out.seq.data(); for (size_t i = o; i < out.seq.size(); i++) q[i] = (char)toupper((unsigned char)q[i]); }
  };
  // (a FASTQ record is about half sequence: room for it up front instead of a dozen doublings with their copies)
  if (end > s && (size_t)(end - s) > ((size_t)1 << 20)) { out.seq.reserve(out.seq.size() + (size_t)(end - s) / 2 + 4096); if (want_qual) out.qual.reserve(out.qual.size() + (size_t)(end - s) / 2 + 4096); out.ids.blob.reserve(out.ids.blob.size() + (size_t)(end - s) / 8); }
  const char *b, *e;
  bool pending = false;                  // a header line already read into b,e
  while (pending || next_line(b, e)) {
    pending = false;
    if (b == e) continue;
    if (*b == '@') {                     // FASTQ record: 4 lines
      const char *ne = b + 1; while (ne < e && *ne != ' ' && *ne != '\t') ne++;
      out.ids.emplace_back(b + 1, ne);
      const char *sb, *se, *pb, *pe, *qb, *qe;
      if (!next_line(sb, se) || !next_line(pb, pe) || !next_line(qb, qe) || pb == pe || *pb != '+' || (qe - qb) != (se - sb)) {
        out.rc = ITSX_E_FORMAT; out.err = "malformed FASTQ record"; return;
      }
      put_seq(sb, se);
      if (want_qual) out.qual.append(qb, qe);
      out.off.push_back((int64_t)out.seq.size());
    } else if (*b == '>') {              // FASTA record: header + sequence lines
      const char *ne = b + 1; while (ne < e && *ne != ' ' && *ne != '\t') ne++;
      out.ids.emplace_back(b + 1, ne);
      while (next_line(b, e)) {
        if (b < e && *b == '>') { pending = true; break; }
        put_seq(b, e);
      }
      if (want_qual) { const size_t q0 = out.qual.size(); out.qual.resize(out.seq.size()); if (out.qual.size() > q0) memset(out.qual.data() + q0, 'I', out.qual.size() - q0); }
      out.off.push_back((int64_t)out.seq.size());
    } else { out.rc = ITSX_E_FORMAT; out.err = "input is neither FASTA nor FASTQ"; return; }
  }
}

// first record start at or after `from`: FASTQ -- a line that starts with '@' whose second line below starts with '+' (a
// quality line may start with '@', but then that line is a sequence line, which never starts with '+'); FASTA -- a '>' line
static size_t next_record_start(const char *t, size_t n, size_t from, bool fastq)
{
  size_t q = from;
  if (q > 0) { const char *nl = (const char *)memchr(t + q - 1, '\n', n - (q - 1)); if (!nl) return n; q = (size_t)(nl - t) + 1; }
  while (q < n) {
    if (!fastq) { if (t[q] == '>') return q; }
    else if (t[q] == '@') {
      const char *l1 = (const char *)memchr(t + q, '\n', n - q);
      const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', n - (size_t)(l1 + 1 - t)) : nullptr;
      if (l2 && (size_t)(l2 + 1 - t) < n && l2[1] == '+') return q;
    }
    const char *nl = (const char *)memchr(t + q, '\n', n - q);
    if (!nl) return n;
    q = (size_t)(nl - t) + 1;
  }
  return n;
}

static int parse_fastx(const itsx_io::Text &text, bool want_qual, bool upper, FastxPart &out, std::string &err)
{
  const size_t n = text.size();
  int T = itsx_io::io_threads();
  if (const char *e = sw_get("ITSX_PARSE_MIN_MB")) { if (n < (size_t)atol(e) << 20) T = 1; }
  else if (n < ((size_t)16 << 20)) T = 1;
  size_t first = 0;
  while (first < n && (text[first] == '\n' || text[first] == '\r')) first++;
  if (first < n && text[first] != '@' && text[first] != '>') T = 1;          // the serial parser reports it
  std::vector<size_t> cut(1, 0);
  if (T > 1) {
    const bool fastq = text[first] == '@';
    for (int k = 1; k < T; k++) { const size_t c = next_record_start(text.data(), n, n / (size_t)T * (size_t)k, fastq); if (c > cut.back() && c < n) cut.push_back(c); }
  }
  cut.push_back(n);
  const size_t np = cut.size() - 1;
  if (np == 1) {
    parse_fastx_range(text.data(), text.data() + n, want_qual, upper, out);
    if (out.rc != ITSX_OK) err = out.err + (out.rc == ITSX_E_FORMAT && out.err[0] == 'm' ? " near read " + std::to_string(out.ids.size()) : "");
    return out.rc;
  }
  std::vector<FastxPart> parts(np);
  on_threads((int)np, [&](int k) { parse_fastx_range(text.data() + cut[(size_t)k], text.data() + cut[(size_t)k + 1], want_qual, upper, parts[(size_t)k]); });
  size_t nrec = 0, nseq = 0;
  for (size_t k = 0; k < np; k++) {
    if (parts[k].rc != ITSX_OK) { err = parts[k].err + " near read " + std::to_string(nrec + parts[k].ids.size()); return parts[k].rc; }
    nrec += parts[k].ids.size(); nseq += parts[k].seq.size();
  }
  const size_t rec0 = out.ids.size(), seq0 = out.seq.size(), nam0 = out.ids.blob.size();
  size_t nnam = 0;
  for (size_t k = 0; k < np; k++) nnam += parts[k].ids.blob.size();
  out.ids.off.resize(rec0 + nrec + 1); out.ids.blob.resize(nam0 + nnam); out.off.resize(rec0 + nrec + 1);
  if (!out.seq.resize(seq0 + nseq) || (want_qual && !out.qual.resize(seq0 + nseq))) { err = "out of memory parsing the reads"; return ITSX_E_NOMEM; }
  std::vector<size_t> rbase(np), sbase(np), nbase(np);
  { size_t r = rec0, q = seq0, m = nam0; for (size_t k = 0; k < np; k++) { rbase[k] = r; sbase[k] = q; nbase[k] = m; r += parts[k].ids.size(); q += parts[k].seq.size(); m += parts[k].ids.blob.size(); } }
  on_threads((int)np, [&](int k) {
    FastxPart &p = parts[(size_t)k];
    if (p.seq.size()) memcpy(out.seq.data() + sbase[(size_t)k], p.seq.data(), p.seq.size());
    if (want_qual && p.qual.size()) memcpy(out.qual.data() + sbase[(size_t)k], p.qual.data(), p.qual.size());
    if (p.ids.blob.size()) memcpy(out.ids.blob.data() + nbase[(size_t)k], p.ids.blob.data(), p.ids.blob.size());
    for (size_t i = 0; i < p.ids.size(); i++) {
      out.ids.off[rbase[(size_t)k] + i + 1] = (int64_t)nbase[(size_t)k] + p.ids.off[i + 1];
      out.off[rbase[(size_t)k] + i + 1] = (int64_t)sbase[(size_t)k] + p.off[i + 1];
    }
  });
  return ITSX_OK;
}

// appended to the context's host-side read set
static int parse_fastx_append(itsx_ctx *ctx, const itsx_io::Text &text)
{
  FastxPart part;
  part.seq.swap(ctx->h_bases); part.off.swap(ctx->h_off); part.ids.swap(ctx->h_names);
  if (part.off.empty()) part.off.assign(1, 0);
  std::string err;
  const int rc = parse_fastx(text, false, false, part, err);
  ctx->h_bases.swap(part.seq); ctx->h_off.swap(part.off); ctx->h_names.swap(part.ids);
  if (rc != ITSX_OK) SET_ERR(ctx, rc, err);
  return ITSX_OK;
}

int itsx_load_reads_file(itsx_ctx *ctx, const char *path, int64_t *n_reads)
{
  CTXCHK(ctx && path);
  std::string rerr;
  static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
  const auto tt0 = std::chrono::steady_clock::now();
  const auto tp = slurp(path, true, rerr);
  if (!tp) SET_ERR(ctx, ITSX_E_IO, rerr);
  const auto tt1 = std::chrono::steady_clock::now();
  ctx->h_bases.clear(); ctx->h_off.assign(1, 0); ctx->h_names.clear();
  { const int prc = parse_fastx_append(ctx, *tp); if (prc != ITSX_OK) return prc; }
  ctx->N = (int64_t)ctx->h_names.size();
  if (n_reads) *n_reads = ctx->N;
  const auto tt2 = std::chrono::steady_clock::now();
  const int rc = pack_and_upload(ctx);
  if (trace) {
    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[itsx] load %s: read+inflate %.0f ms, parse %.0f ms, upload+pack %.0f ms\n", path, ms(tt0, tt1), ms(tt1, tt2), ms(tt2, std::chrono::steady_clock::now()));
  }
  return rc;
}

int itsx_load_reads_text(itsx_ctx *ctx, const char *text, int64_t nbytes, int64_t *n_reads)
{
  CTXCHK(ctx && (text || nbytes == 0) && nbytes >= 0);
  LoadPriority load_priority(ctx);
  itsx_io::Text view;
  view.borrow(text ? text : "", (size_t)nbytes);
  ctx->h_bases.clear(); ctx->h_off.assign(1, 0); ctx->h_names.clear();
  { const int prc = parse_fastx_append(ctx, view); if (prc != ITSX_OK) return prc; }
  ctx->N = (int64_t)ctx->h_names.size();
  if (n_reads) *n_reads = ctx->N;
  return pack_and_upload(ctx);
}

// One shard of a file's reads: records [n shard / n_shards, n (shard + 1) / n_shards) in file order (input order is kept inside a
// shard: first-occurrence representatives depend on it).  Every worker of a multi-GPU run parses the file itself (the decompressed
// text is cached per process) and keeps its own slice.
int itsx_load_reads_file_shard(itsx_ctx *ctx, const char *path, int32_t shard, int32_t n_shards, int64_t *n_total, int64_t *first, int64_t *n_reads)
{
  CTXCHK(ctx && path);
  if (n_shards < 1 || shard < 0 || shard >= n_shards) SET_ERR(ctx, ITSX_E_ARG, "itsx_load_reads_file_shard: shard out of range");
  std::string rerr;
  const auto tp = slurp(path, true, rerr);
  if (!tp) SET_ERR(ctx, ITSX_E_IO, rerr);
  ctx->h_bases.clear(); ctx->h_off.assign(1, 0); ctx->h_names.clear();
  { const int prc = parse_fastx_append(ctx, *tp); if (prc != ITSX_OK) return prc; }
  const int64_t tot = (int64_t)ctx->h_names.size();
  const int64_t lo = tot * shard / n_shards, hi = tot * (shard + 1) / n_shards;
  if (lo > 0 || hi < tot) {
    const int64_t b0 = ctx->h_off[(size_t)lo], b1 = ctx->h_off[(size_t)hi];
    itsx_io::Text bases;
    bases.assign(ctx->h_bases.data() + b0, (size_t)(b1 - b0));
    std::vector<int64_t> off((size_t)(hi - lo) + 1);
    for (int64_t r = lo; r <= hi; r++) off[(size_t)(r - lo)] = ctx->h_off[(size_t)r] - b0;
    NameList names = ctx->h_names.slice((size_t)lo, (size_t)hi);
    ctx->h_bases.swap(bases); ctx->h_off.swap(off); ctx->h_names.swap(names);
  }
  ctx->N = hi - lo;
  if (n_total) *n_total = tot;
  if (first) *first = lo;
  if (n_reads) *n_reads = ctx->N;
  return pack_and_upload(ctx);
}

// ---- per-sample batching (q2_itsxpress.py:273-333 runs the whole path once per sample)
int itsx_set_samples(itsx_ctx *ctx, const int32_t *sample_of_read, int32_t n_samples)
{
  CTXCHK(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  ctx->have_derep = ctx->have_search = ctx->have_final = false;
  ctx->sel_sample = -1; ctx->h_usample.clear();
  ctx->order_cache_ok = false;
  if (!sample_of_read || n_samples <= 1) { ctx->S = 1; ctx->h_sample.clear(); return ITSX_OK; }
  if ((int64_t)n_samples * std::max(ctx->P, 1) > (1ll << 28)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "too many samples x profiles for one batch");
  for (int64_t r = 0; r < ctx->N; r++)
    if (sample_of_read[r] < 0 || sample_of_read[r] >= n_samples) SET_ERR(ctx, ITSX_E_ARG, "sample index of read " + std::to_string(r) + " is out of range");
  ctx->h_sample.assign(sample_of_read, sample_of_read + ctx->N);
  ctx->S = n_samples;
  HIPCHK(upload(ctx->d_sample, ctx->h_sample, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}
int itsx_num_samples(const itsx_ctx *ctx) { return ctx ? ctx->S : ITSX_E_ARG; }
int itsx_select_sample(itsx_ctx *ctx, int32_t sample)
{
  CTXCHK(ctx);
  if (sample < -1 || sample >= ctx->S) SET_ERR(ctx, ITSX_E_ARG, "sample index out of range");
  ctx->sel_sample = sample;
  ctx->order_cache_ok = false;
  return ITSX_OK;
}
int itsx_load_reads_files(itsx_ctx *ctx, const char *const *paths, int32_t n_paths, int64_t *n_reads_per_file)
{
  CTXCHK(ctx && paths && n_paths >= 1);
  ctx->h_bases.clear(); ctx->h_off.assign(1, 0); ctx->h_names.clear();
  std::vector<int32_t> smp;
  for (int32_t f = 0; f < n_paths; f++) {
    CTXCHK(paths[f]);
    std::string rerr;
    const auto tp = slurp(paths[f], true, rerr);
    if (!tp) SET_ERR(ctx, ITSX_E_IO, rerr);
    const size_t before = ctx->h_names.size();
    { const int prc = parse_fastx_append(ctx, *tp); if (prc != ITSX_OK) { ctx->set_error(ctx->err + " (" + paths[f] + ")"); return prc; } }
    if (n_reads_per_file) n_reads_per_file[f] = (int64_t)(ctx->h_names.size() - before);
    smp.resize(ctx->h_names.size(), f);
  }
  ctx->N = (int64_t)ctx->h_names.size();
  if (ctx->N >= (1ll << 31) - 64) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 reads in one context");
  const int rc = pack_and_upload(ctx);
  if (rc != ITSX_OK) return rc;
  return itsx_set_samples(ctx, smp.data(), n_paths);
}

// ------------------------------------------------------------------------------ derep
// uniques = seeds in input order, then ordered by length for the HMM stages (shared by derep and cluster)
static int build_unique_lists(itsx_ctx *ctx)
{
  const int64_t n = ctx->N;
  DBuf<int32_t> &is_seed = ctx->w_is_seed, &seed_rank = ctx->w_seed_rank, &scan_tmp = ctx->w_scan_tmp;
  int32_t U = 0;
  if (n > 0) {
    launch_exclusive_scan(is_seed.p, seed_rank.p, n + 1, scan_tmp.p, ctx->st);   // element n = total (is_seed[n] unused but allocated)
    HIPCHK(hipMemcpyAsync(&U, seed_rank.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
  }
  ctx->U = U; ctx->U_active = U;
  HIPCHK(ctx->d_seed_read.alloc((size_t)U + 1)); HIPCHK(ctx->d_abund.alloc((size_t)U + 1)); HIPCHK(ctx->d_sorted_uniq.alloc((size_t)U + 1));
  HIPCHK(ctx->d_ulen.alloc((size_t)U + 1));
  HIPCHK(hipMemsetAsync(ctx->d_abund.p, 0, ((size_t)U + 1) * sizeof(int32_t), ctx->st));
  if (n > 0) launch_uniques(n, ctx->d_rep_of.p, seed_rank.p, ctx->d_uniq_of.p, ctx->d_seed_read.p, ctx->d_abund.p, ctx->st);
  // order the uniques by length (counting sort) for the HMM stages
  if (U > 0) {
    const int32_t lcap = 65536;
    DBuf<int32_t> &hist = ctx->w_hist, &cursor = ctx->w_cursor, &tmp2 = ctx->w_tmp2;
    HIPCHK(hist.alloc(lcap)); HIPCHK(cursor.alloc(lcap)); HIPCHK(tmp2.alloc((size_t)scan_tmp_elems(lcap)));
    HIPCHK(hipMemsetAsync(hist.p, 0, lcap * sizeof(int32_t), ctx->st));
    launch_len_hist(U, ctx->d_seed_read.p, ctx->rd.len, hist.p, lcap, ctx->st, ctx->Lmax + 1);
    launch_exclusive_scan(hist.p, cursor.p, lcap, tmp2.p, ctx->st);
    launch_len_scatter(U, ctx->d_seed_read.p, ctx->rd.len, cursor.p, lcap, ctx->d_sorted_uniq.p, ctx->st, ctx->Lmax + 1);
    launch_fill_ulen(U, ctx->d_sorted_uniq.p, ctx->d_seed_read.p, ctx->rd.len, ctx->d_ulen.p, ctx->st);
    HIPCHK(hipStreamSynchronize(ctx->st));
  }
  return ITSX_OK;
}
static int mirror_derep(itsx_ctx *ctx)
{
  const int64_t n = ctx->N; const int32_t U = ctx->U;
  // host mirrors (writers, getters)
  ctx->h_rep_of.resize((size_t)n); ctx->h_strand.resize((size_t)n); ctx->h_uniq_of.resize((size_t)n);
  ctx->h_seed_read.resize((size_t)U); ctx->h_abund.resize((size_t)U); ctx->h_sorted_uniq.resize((size_t)U);
  if (n > 0) {
    HIPCHK(hipMemcpy(ctx->h_rep_of.data(), ctx->d_rep_of.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ctx->h_strand.data(), ctx->d_strand.p, (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ctx->h_uniq_of.data(), ctx->d_uniq_of.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  if (U > 0) {
    HIPCHK(hipMemcpy(ctx->h_seed_read.data(), ctx->d_seed_read.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ctx->h_abund.data(), ctx->d_abund.p, (size_t)U * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ctx->h_sorted_uniq.data(), ctx->d_sorted_uniq.p, (size_t)U * 4, hipMemcpyDeviceToHost));
  }
  ctx->h_sorted_active = ctx->h_sorted_uniq;
  if (ctx->S > 1) {
    ctx->h_usample.resize((size_t)U);
    for (int32_t u = 0; u < U; u++) ctx->h_usample[(size_t)u] = ctx->h_sample[(size_t)ctx->h_seed_read[(size_t)u]];
    HIPCHK(upload(ctx->d_usample, ctx->h_usample, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
  }
  return ITSX_OK;
}

int itsx_derep(itsx_ctx *ctx, int strand_both, int minseqlength, int64_t *n_unique)
{
  CTXCHK(ctx);
  HIPCHK(hipSetDevice(ctx->device));
  LoadPriority load_priority(ctx);
  const int64_t n = ctx->N;
  StageTimer tm(ctx->st);
  DBuf<uint64_t> &hf = ctx->w_hf, &hr = ctx->w_hr; DBuf<unsigned long long> &keys = ctx->w_keys;
  DBuf<int32_t> &vals = ctx->w_vals, &is_seed = ctx->w_is_seed, &seed_rank = ctx->w_seed_rank, &scan_tmp = ctx->w_scan_tmp;
  DBuf<uint32_t> &slot_of = ctx->w_slot_of; DBuf<unsigned int> &ncoll = ctx->w_ncoll;
  HIPCHK(hf.alloc((size_t)n + 1)); HIPCHK(hr.alloc((size_t)n + 1));
  uint64_t tsize = 1024; while (tsize < (uint64_t)n * 2 + 16) tsize <<= 1;
  HIPCHK(keys.alloc(tsize)); HIPCHK(vals.alloc(tsize)); HIPCHK(slot_of.alloc((size_t)n + 1)); HIPCHK(ncoll.alloc(1));
  HIPCHK(is_seed.alloc((size_t)n + 1)); HIPCHK(seed_rank.alloc((size_t)n + 1)); HIPCHK(scan_tmp.alloc((size_t)scan_tmp_elems(n + 1)));
  HIPCHK(hipMemsetAsync(is_seed.p, 0, ((size_t)n + 1) * sizeof(int32_t), ctx->st));
  HIPCHK(ctx->d_rep_of.alloc((size_t)n + 1)); HIPCHK(ctx->d_strand.alloc((size_t)n + 1)); HIPCHK(ctx->d_uniq_of.alloc((size_t)n + 1));
  unsigned int hcoll = 0;
  uint64_t seed = 0;
  ctx->stats.hash_reseeds = 0;
  for (int attempt = 0; attempt < 4; attempt++) {
    HIPCHK(hipMemsetAsync(keys.p, 0, tsize * sizeof(unsigned long long), ctx->st));
    HIPCHK(hipMemsetAsync(vals.p, 0x7f, tsize * sizeof(int32_t), ctx->st));
    HIPCHK(hipMemsetAsync(ncoll.p, 0, sizeof(unsigned int), ctx->st));
    if (n > 0) {
      launch_hash_reads(ctx->rd, seed, strand_both, hf.p, hr.p, ctx->st, ctx->dev_sample());
      launch_table_insert(n, ctx->rd.len, minseqlength, hf.p, hr.p, keys.p, vals.p, tsize - 1, slot_of.p, ctx->st);
      launch_table_resolve(ctx->rd, hf.p, vals.p, slot_of.p, ctx->d_rep_of.p, ctx->d_strand.p, is_seed.p, ncoll.p, ctx->st, ctx->dev_sample());
    }
    HIPCHK(hipMemcpyAsync(&hcoll, ncoll.p, sizeof(hcoll), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    if (hcoll == 0) break;
    seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    ctx->stats.hash_reseeds++;
  }
  if (hcoll != 0) SET_ERR(ctx, ITSX_E_COLLISION, "64-bit key collisions survived 4 reseeds");
  { const int rc_ = build_unique_lists(ctx); if (rc_ != ITSX_OK) return rc_; }
  ctx->stats.ms_derep = tm.stop();
  HIPCHK(hipGetLastError());
  { const int rc_ = mirror_derep(ctx); if (rc_ != ITSX_OK) return rc_; }
  const int32_t U = ctx->U;
  int64_t dropped = 0;
  for (int64_t r = 0; r < n; r++) dropped += ctx->h_rep_of[r] < 0;
  ctx->stats.n_unique = U; ctx->stats.n_dropped_short = dropped;
  ctx->have_derep = true; ctx->have_search = ctx->have_final = false; ctx->clustered = false;
  ctx->order_cache_ok = false;
  ctx->derep_strand_both = strand_both != 0;
  if (n_unique) *n_unique = U;
  return ITSX_OK;
}

// vsearch's DUST soft mask (mask.cc dust() / wo()) on the host, for the orientation database (the reads are masked on the device by
// k_dust, same procedure): windows of 64 advancing by 32, 3-mer repeat score 10 * sum / j, masked above 20
static bool qmask_dust() { const char *e = sw_get("ITSX_QMASK"); return !(e && strcmp(e, "none") == 0); }
static void dust_host(const uint8_t *codes, int64_t L, std::vector<uint8_t> &masked)
{
  masked.assign((size_t)L, 0);
  for (int64_t i = 0; i < L; i += 32) {
    const int len = (L > i + 64) ? 64 : (int)(L - i);
    const int l1 = len - 7;
    int bestv = 0, besti = 0, bestj = 0;
    if (l1 >= 0) {
      int words[64], counts[64], word = 0;
      for (int j = 0; j < len; j++) { const int c = codes[i + j]; word = (word << 2) | (c < 4 ? c : 0); words[j] = word & 63; }
      for (int a = 0; a < l1; a++) {
        memset(counts, 0, sizeof(counts));
        int sum = 0;
        for (int j = 2; j < len - a; j++) {
          const int w = words[a + j], c = counts[w];
          if (c) { sum += c; const int v = 10 * sum / j; if (v > bestv) { bestv = v; besti = a; bestj = j; } }
          counts[w]++;
        }
      }
    }
    if (bestv > 20) {
      const int b = besti + bestj;
      for (int64_t j = besti + i; j <= b + i; j++) masked[(size_t)j] = 1;
      if (b < 32) i += 32 - b;
    }
  }
}

// a2: greedy centroid clustering (k_cluster.hip explains the speculative windows)
int itsx_cluster(itsx_ctx *ctx, double id, int strand_both, int64_t *n_unique)
{
  CTXCHK(ctx);
  if (!(id > 0.0 && id <= 1.0)) SET_ERR(ctx, ITSX_E_ARG, "cluster id must be in (0, 1]");
  if (id == 1.0) return itsx_derep(ctx, strand_both, 32, n_unique);      // main.py:534-537 never clusters at 1.0
  if (ctx->S > 1) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "greedy clustering (id < 1) is sequential per sample: run it one sample per call, not on a sample batch");
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->N;
  const int minlen = 32;
  StageTimer tm(ctx->st);
  // processing order: abundance (all 1) descending, then label, then input position
  std::vector<int32_t> ord; ord.reserve((size_t)n);
  int Lmax = 1;
  for (int64_t r = 0; r < n; r++)                          // vsearch defaults: --minseqlength 32, --maxseqlength 50000
    if (ctx->h_len[r] >= minlen && ctx->h_len[r] <= 50000) { ord.push_back((int32_t)r); Lmax = std::max(Lmax, (int)ctx->h_len[r]); }
  if (!ctx->h_names.empty())
    std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
      const int c = ctx->h_names.cmp((size_t)a, (size_t)b);
      return c < 0 || (c == 0 && a < b);
    });
  const int32_t nk = (int32_t)ord.size();
  int Bmax = 4096;                                          // k_cl_resolve keeps the window's flags in LDS
  if (const char *e = sw_get("ITSX_CL_WINDOW")) Bmax = std::min(4096, std::max(1, atoi(e)));
  // DP rows per lane of the alignment wave: 64 lanes x S rows hold the whole query when Lmax + 1 <= 64 S; the lanes past the
  // query's end idle, so S is the smallest that fits (2x250-merged reads of <= 480 bases: S = 8, not 10: 78 % instead of 62 % busy)
  int rows_per_lane = (Lmax + 1 <= 320) ? 5 : (Lmax + 1 <= 512) ? 8 : 10;
  if (const char *e = sw_get("ITSX_CL_ROWS")) rows_per_lane = atoi(e) <= 5 ? 5 : atoi(e) <= 8 ? 8 : 10;
  const bool multipass = Lmax + 1 > 64 * rows_per_lane;
  const int32_t scratch_pitch = Lmax + 1;
  if (multipass) while (Bmax > 16 && 2LL * Bmax * 32 * scratch_pitch * 16 > (4LL << 30)) Bmax /= 2;
  const int32_t kcap = std::max(8, ((Lmax - 7 + 7) / 8) * 8);

  DBuf<int32_t> d_order, cent_len, cent_pos, cent_read, res_col, knk, state, rejects, acc_col, is_new, new_rank, scan_tmp, xlist, xn, hard, dbg;
  DBuf<int32_t> sel, selm, sel_short, wn, wcol, newq, rm, wout, work, xwork, work_n, replay, skipm, canon, ctab_val, need, awork, spairs;
  DBuf<int32_t> cw_n, wsum, wscan, qi_cnt, qi_cur, qi_off, ncand, ntop, ovf, qi_hid, qi_nheavy;
  DBuf<uint32_t> qi_bm;
  DBuf<int64_t> cw_off, cw_base;
  DBuf<uint16_t> klist, cw_poolA, cw_poolB, qi_ent, cntx; DBuf<uint32_t> tq, minm;
  DBuf<unsigned long long> ctab_key, n_skipped, pre_stats, cand, tkey;
  DBuf<int8_t> res_strand; DBuf<double> res_id, acc_id, d_pct, selpid, wpid, xpid;
  DBuf<unsigned long long> prev, bound, selkey, wkey, xkey, scratch, n_align;
  DBuf<uint16_t> *cw_pool = &cw_poolA, *cw_other = &cw_poolB;
  HIPCHK(upload(d_order, ord, ctx->st));
  const size_t ccap = (size_t)nk + 2048;
  HIPCHK(cent_len.alloc(ccap)); HIPCHK(cent_pos.alloc(ccap)); HIPCHK(cent_read.alloc(ccap));
  HIPCHK(cw_n.alloc(ccap)); HIPCHK(cw_off.alloc(ccap + 1)); HIPCHK(cw_base.alloc(2));
  HIPCHK(hipMemsetAsync(cw_off.p, 0, sizeof(int64_t), ctx->st));                       // the first column starts at 0
  HIPCHK(res_col.alloc((size_t)nk + 1)); HIPCHK(res_strand.alloc((size_t)nk + 1)); HIPCHK(res_id.alloc((size_t)nk + 1));
  const size_t nqs = 2 * (size_t)Bmax;
  HIPCHK(klist.alloc(nqs * kcap)); HIPCHK(knk.alloc(nqs)); HIPCHK(state.alloc(nqs)); HIPCHK(rejects.alloc(nqs));
  HIPCHK(acc_col.alloc(nqs)); HIPCHK(prev.alloc(nqs)); HIPCHK(bound.alloc(nqs)); HIPCHK(acc_id.alloc(nqs));
  HIPCHK(sel.alloc(nqs * 32)); HIPCHK(selm.alloc(nqs)); HIPCHK(sel_short.alloc(nqs)); HIPCHK(selkey.alloc(nqs * 32)); HIPCHK(selpid.alloc(nqs * 32));
  HIPCHK(wn.alloc(nqs)); HIPCHK(wcol.alloc(nqs * 32)); HIPCHK(wkey.alloc(nqs * 32)); HIPCHK(wpid.alloc(nqs * 32));
  HIPCHK(xlist.alloc(nqs * 32)); HIPCHK(xn.alloc(nqs)); HIPCHK(hard.alloc(nqs)); HIPCHK(xkey.alloc(nqs * 32)); HIPCHK(xpid.alloc(nqs * 32));
  HIPCHK(is_new.alloc((size_t)Bmax + 1)); HIPCHK(new_rank.alloc((size_t)Bmax + 1)); HIPCHK(newq.alloc((size_t)Bmax + 1)); HIPCHK(rm.alloc((size_t)Bmax + 1));
  HIPCHK(wsum.alloc((size_t)Bmax + 1)); HIPCHK(wscan.alloc((size_t)Bmax + 1));
  HIPCHK(wout.alloc(4)); HIPCHK(dbg.alloc(4)); HIPCHK(work.alloc(nqs * 32)); HIPCHK(xwork.alloc(nqs * 32)); HIPCHK(work_n.alloc(8)); HIPCHK(awork.alloc(2 * nqs * 32)); HIPCHK(spairs.alloc(4 * nqs * 32)); HIPCHK(replay.alloc((size_t)Bmax + 1)); HIPCHK(skipm.alloc((size_t)Bmax + 1)); HIPCHK(canon.alloc((size_t)Bmax + 1)); HIPCHK(need.alloc(2 * nqs * 32)); HIPCHK(n_skipped.alloc(1)); HIPCHK(pre_stats.alloc(16)); HIPCHK(hipMemsetAsync(pre_stats.p, 0, 16 * sizeof(unsigned long long), ctx->st));
  HIPCHK(hipMemsetAsync(n_skipped.p, 0, sizeof(unsigned long long), ctx->st)); HIPCHK(ctab_key.alloc(16384)); HIPCHK(ctab_val.alloc(16384));
  HIPCHK(ctx->w_hf.alloc((size_t)n + 1)); HIPCHK(ctx->w_hr.alloc((size_t)n + 1));
  if (n > 0) launch_hash_reads(ctx->rd, 0, 0, ctx->w_hf.p, ctx->w_hr.p, ctx->st);      // identical reads of a window share one search
  // vsearch's default --qmask dust / --dbmask dust: the DUST soft mask of every read (queries and centroids alike), once
  DBuf<uint32_t> dmask;
  const bool use_dust = qmask_dust();
  if (use_dust && n > 0) { HIPCHK(dmask.alloc((size_t)ctx->h_woff[n] + 1)); launch_dust(ctx->rd, dmask.p, ctx->st); }
  HIPCHK(scan_tmp.alloc((size_t)scan_tmp_elems(65537 + Bmax + 1))); HIPCHK(n_align.alloc(1));
  HIPCHK(scratch.alloc(multipass ? nqs * 32 * (size_t)scratch_pitch * 2 : 2));
  HIPCHK(hipMemsetAsync(n_align.p, 0, sizeof(unsigned long long), ctx->st));
  // the window's query index, the strands' thresholds and candidate lists, the counts against the window's own centroids
  HIPCHK(qi_cnt.alloc(65537)); HIPCHK(qi_cur.alloc(65536)); HIPCHK(qi_off.alloc(65537));
  HIPCHK(qi_ent.alloc(nqs * (size_t)kcap + 65536 * 8 + 64));
  int32_t hcap = 16384;                                     // strand bitmaps of conserved words (1 KB each); ITSX_CL_HEAVY=0: every word keeps its list
  if (const char *e = sw_get("ITSX_CL_HEAVY")) hcap = std::max(0, std::min(65536, atoi(e)));
  HIPCHK(qi_hid.alloc(65536)); HIPCHK(qi_nheavy.alloc(1)); HIPCHK(qi_bm.alloc((size_t)std::max(hcap, 1) * (CL_QS_MAX / 32)));
  HIPCHK(tq.alloc(CL_QS_MAX + 8)); HIPCHK(minm.alloc(CL_QS_MAX + 8)); HIPCHK(tkey.alloc(nqs)); HIPCHK(ncand.alloc(nqs)); HIPCHK(ntop.alloc(nqs)); HIPCHK(ovf.alloc(1));
  int32_t cand_cap = 4096;                                  // per strand; grows (and the window is searched again) when a list overflows
  if (const char *e = sw_get("ITSX_CL_CCAP")) cand_cap = std::max(32, atoi(e));
  HIPCHK(cand.alloc(nqs * (size_t)cand_cap));
  HIPCHK(cntx.alloc(nqs * (size_t)Bmax));
  int64_t pool_cap = 1LL << 26;                             // words of all centroids; doubles (with a copy) when a window could overrun it
  if (const char *e = sw_get("ITSX_CL_CAPACITY")) pool_cap = std::max<int64_t>(2048, atoll(e));
  pool_cap = std::max<int64_t>(pool_cap, (int64_t)Bmax * kcap);
  HIPCHK(cw_pool->alloc((size_t)pool_cap, true));
  int64_t pool_used = 0;

  ClusterArgs a{};
  a.rd = ctx->rd; a.order = d_order.p; a.strand_both = strand_both ? 1 : 0; a.dmask = (use_dust && n > 0) ? dmask.p : nullptr;
  a.cent_len = cent_len.p; a.cent_pos = cent_pos.p; a.cent_read = cent_read.p;
  a.klist = klist.p; a.kcap = kcap; a.nk = knk.p;
  a.cw_off = cw_off.p; a.cw_n = cw_n.p; a.cw_base = cw_base.p; a.wsum = wsum.p; a.wscan = wscan.p;
  a.qi_cnt = qi_cnt.p; a.qi_cur = qi_cur.p; a.qi_off = qi_off.p; a.qi_ent = qi_ent.p;
  a.qi_hid = qi_hid.p; a.qi_nheavy = qi_nheavy.p; a.qi_bm = qi_bm.p; a.hcap = hcap;
  a.heavy_min = CL_HEAVY;                                    // ITSX_CL_HEAVY_MIN: strands that must hold a word before it gets a bitmap (tuning; results do not depend on it)
  if (const char *e = sw_get("ITSX_CL_HEAVY_MIN")) a.heavy_min = std::max(8, atoi(e));
  a.tq = tq.p; a.minm = minm.p; a.tkey = tkey.p; a.ncand = ncand.p; a.ntop = ntop.p; a.ovf = ovf.p; a.cntx = cntx.p;
  a.state = state.p; a.rejects = rejects.p; a.acc_col = acc_col.p; a.prev = prev.p; a.bound = bound.p; a.acc_id = acc_id.p;
  a.sel = sel.p; a.selm = selm.p; a.sel_short = sel_short.p; a.selkey = selkey.p; a.selpid = selpid.p;
  a.wn = wn.p; a.wcol = wcol.p; a.wkey = wkey.p; a.wpid = wpid.p;
  a.res_col = res_col.p; a.res_strand = res_strand.p; a.res_id = res_id.p;
  a.is_new = is_new.p; a.new_rank = new_rank.p; a.newq = newq.p; a.rm = rm.p;
  a.xlist = xlist.p; a.xn = xn.p; a.hard = hard.p; a.xkey = xkey.p; a.xpid = xpid.p; a.wout = wout.p; a.dbg = dbg.p; a.work = work.p; a.xwork = xwork.p; a.work_n = work_n.p; a.awork = awork.p; a.spairs = spairs.p; a.replay = replay.p; a.skipm = skipm.p; a.canon = canon.p; a.need = sw_get("ITSX_CL_NOPRECHECK") ? nullptr : need.p; a.need_pitch = (int32_t)(nqs * 32); a.n_skipped = n_skipped.p; a.pre_stats = pre_stats.p;
  a.use_score = sw_get("ITSX_CL_NOSCORE") ? 0 : 1;
  a.pre_k = std::min(16, (int)((double)Lmax * (1.0 - id) / id) + 1); a.ctab_key = ctab_key.p; a.ctab_val = ctab_val.p; a.rhash = ctx->w_hf.p;
  a.scratch = scratch.p; a.scratch_pitch = scratch_pitch;
  a.thr = 100.0 * id; a.n_align = n_align.p;

  int32_t f = 0, C = 0, ncent = 0;
  int B = std::min(Bmax, 256);
  int64_t windows = 0, cuts = 0, regrown = 0, stream_launches = 0;
  const bool debug = sw_get("ITSX_CL_DEBUG") != nullptr;
  while (f < nk) {
    const int nq = std::min<int32_t>(B, nk - f);
    if (pool_used + (int64_t)nq * kcap > pool_cap) {          // the window's would-be centroids could overrun the word pool: grow it
      int64_t ncap = pool_cap;
      while (pool_used + (int64_t)nq * kcap > ncap) ncap *= 2;
      HIPCHK(cw_other->alloc((size_t)ncap, true));
      HIPCHK(hipMemcpyAsync(cw_other->p, cw_pool->p, (size_t)pool_used * sizeof(uint16_t), hipMemcpyDeviceToDevice, ctx->st));
      HIPCHK(hipStreamSynchronize(ctx->st));
      std::swap(cw_pool, cw_other); cw_other->release();
      pool_cap = ncap;
    }
    a.f = f; a.nq = nq; a.C = C; a.cw_pool = cw_pool->p; a.cand = cand.p; a.ccap = cand_cap; a.xpitch = nq;
    launch_cl_kmers(a, ctx->st);
    // every centroid streams past the window's query index; a strand keeps the candidates that can still be among its 32 best
    launch_cl_qindex(a, scan_tmp.p, ctx->st);
    for (int c0 = 0, step = 2048; c0 < C; step = std::min(step * 2, 1 << 20)) {
      const int c1 = std::min(C, c0 + step);
      launch_cl_stream(a, c0, c1, 1, ctx->st);
      stream_launches++;
      if (c1 < C) launch_cl_topk(a, 0, ctx->st);              // cut the lists back to their 32 best, raise the thresholds
      c0 = c1;
    }
    launch_cl_topk(a, 1, ctx->st);
    {   // a candidate list that overflowed is known HERE: the window is searched again before its walk, its validation and their
        // counters have run (they ran first and were counted twice until round 3: advisor)
      int32_t early = 0;
      HIPCHK(hipMemcpyAsync(&early, ovf.p, sizeof(early), hipMemcpyDeviceToHost, ctx->st));
      HIPCHK(hipStreamSynchronize(ctx->st));
      if (early) {
        if ((int64_t)cand_cap >= (int64_t)C) SET_ERR(ctx, ITSX_E_DEVICE, "clustering candidate lists overflow although they hold every centroid");
        cand_cap *= 4;
        HIPCHK(cand.alloc(nqs * (size_t)cand_cap));
        regrown++;
        continue;
      }
    }
    launch_cl_init(a, ctx->st);
    if (C > 0) launch_cl_walk(a, rows_per_lane, ctx->st);
    launch_cl_outcome(a, ctx->st);
    launch_exclusive_scan(is_new.p, new_rank.p, nq + 1, scan_tmp.p, ctx->st);
    launch_cl_wsum(a, ctx->st);
    launch_exclusive_scan(wsum.p, wscan.p, nq + 1, scan_tmp.p, ctx->st);
    launch_cl_columns(a, 0, ctx->st);
    HIPCHK(hipMemsetAsync(cntx.p, 0, (size_t)2 * nq * nq * sizeof(uint16_t), ctx->st));
    launch_cl_stream(a, C, C + nq, 2, ctx->st);               // the window's speculative centroids against the window's strands
    launch_cl_validate(a, rows_per_lane, ctx->st);
    launch_cl_columns(a, 1, ctx->st);                         // roll back the speculative centroids that did not survive
    int32_t wo[3] = {0, 0, 0}, overflow = 0;
    int64_t cwb[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(wo, wout.p, sizeof(wo), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(cwb, cw_base.p, sizeof(cwb), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(&overflow, ovf.p, sizeof(overflow), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());                              // a kernel that failed to launch must not pass silently
    if (overflow) {
      // a strand met more candidates at (or above) its threshold than its list holds -- long runs of equal counts: nothing of
      // this window is kept, the lists grow and the window is searched again (the centroid set has not changed yet)
      if ((int64_t)cand_cap >= (int64_t)C) SET_ERR(ctx, ITSX_E_DEVICE, "clustering candidate lists overflow although they hold every centroid");
      cand_cap *= 4;
      HIPCHK(cand.alloc(nqs * (size_t)cand_cap));
      regrown++;
      continue;
    }
    const int32_t cut = wo[0];
    if (debug) {
      int32_t d[4] = {0, 0, 0, 0};
      (void)hipMemcpy(d, dbg.p, sizeof(d), hipMemcpyDeviceToHost);
      std::vector<int32_t> hs(2 * (size_t)nq), hw(2 * (size_t)nq), hc(2 * (size_t)nq);
      (void)hipMemcpy(hs.data(), state.p, hs.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hw.data(), wn.p, hw.size() * 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hc.data(), ncand.p, hc.size() * 4, hipMemcpyDeviceToHost);
      long sc[4] = {0, 0, 0, 0}, sw[4] = {0, 0, 0, 0}, acc1 = 0, ncs = 0, ncm = 0;
      for (size_t q = 0; q < hs.size(); q++) { sc[hs[q] & 3]++; sw[hs[q] & 3] += hw[q]; acc1 += hs[q] == 1 && hw[q] == 1; ncs += hc[q]; ncm = std::max<long>(ncm, hc[q]); }
      fprintf(stderr, "[cluster] f=%d nq=%d C=%d cut=%d cols=%d new=%d | hard=%d budget=%d joined_new=%d xaligns=%d | accept %ld (first try %ld, walk %ld) rej32 %ld exhausted %ld (walk %ld) | appended %ld (max %ld per strand)\n",
              f, nq, C, cut, wo[1], wo[2], d[0], d[1], d[2], d[3], sc[1], acc1, sw[1], sc[2], sc[3], sw[3], ncs, ncm);
    }
    if (cut < 1 || cut > nq) SET_ERR(ctx, ITSX_E_DEVICE, "clustering window validation returned an impossible cut");
    C += wo[1]; ncent += wo[2]; f += cut; windows++; cuts += cut < nq;
    pool_used = cwb[1];
    B = cut == nq ? std::min(Bmax, std::max(B, nq) * 2) : std::min(Bmax, std::max(64, 2 * cut));
  }
  if (debug) fprintf(stderr, "[cluster] %lld windows, %lld stream launches, %lld regrown candidate lists, word pool %.1f MB\n", (long long)windows, (long long)stream_launches, (long long)regrown, pool_used * 2.0 / 1e6);

  HIPCHK(ctx->d_rep_of.alloc((size_t)n + 1)); HIPCHK(ctx->d_strand.alloc((size_t)n + 1)); HIPCHK(ctx->d_uniq_of.alloc((size_t)n + 1));
  HIPCHK(ctx->w_is_seed.alloc((size_t)n + 1)); HIPCHK(ctx->w_seed_rank.alloc((size_t)n + 1)); HIPCHK(ctx->w_scan_tmp.alloc((size_t)scan_tmp_elems(n + 1)));
  HIPCHK(d_pct.alloc((size_t)n + 1));
  HIPCHK(hipMemsetAsync(ctx->w_is_seed.p, 0, ((size_t)n + 1) * sizeof(int32_t), ctx->st));
  HIPCHK(hipMemsetAsync(ctx->d_rep_of.p, 0xff, ((size_t)n + 1) * sizeof(int32_t), ctx->st));
  HIPCHK(hipMemsetAsync(ctx->d_strand.p, 1, (size_t)n + 1, ctx->st));
  HIPCHK(hipMemsetAsync(d_pct.p, 0, ((size_t)n + 1) * sizeof(double), ctx->st));
  launch_cl_finalize(nk, d_order.p, res_col.p, res_strand.p, res_id.p, cent_read.p, ctx->d_rep_of.p, ctx->d_strand.p, d_pct.p, ctx->w_is_seed.p, ctx->st);
  { const int rc_ = build_unique_lists(ctx); if (rc_ != ITSX_OK) return rc_; }
  unsigned long long naln = 0, nskip = 0;
  HIPCHK(hipMemcpy(&naln, n_align.p, sizeof(naln), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&nskip, n_skipped.p, sizeof(nskip), hipMemcpyDeviceToHost));
  if (debug) {
    unsigned long long ps[16] = {0};
    (void)hipMemcpy(ps, pre_stats.p, sizeof(ps), hipMemcpyDeviceToHost);
    if (ps[12]) fprintf(stderr, "[cluster] stream phases (clock ticks per centroid and workgroup): pieces %.0f, list additions %.0f, scan %.0f, bitmaps %.0f (of which loads + adders %.0f); pieces per centroid %.0f\n",
                        (double)ps[8] / ps[12], (double)ps[9] / ps[12], (double)ps[10] / ps[12], (double)(ps[11] + ps[14]) / ps[12], (double)ps[14] / ps[12], (double)ps[13] / ps[12]);
    fprintf(stderr, "[cluster] certificate: not applicable %llu, bound too weak %llu (score pass: proven reject %llu, path exists %llu), path exists %llu, proven reject %llu; full alignments %llu\n", ps[0], ps[1], ps[4], ps[5], ps[2], ps[3], naln);
  }
  ctx->stats.ms_cluster = tm.stop();
  { const int rc_ = mirror_derep(ctx); if (rc_ != ITSX_OK) return rc_; }
  ctx->h_pct.assign((size_t)n, -1.0);
  if (n > 0) HIPCHK(hipMemcpy(ctx->h_pct.data(), d_pct.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < n; r++) if (ctx->h_rep_of[r] < 0 || ctx->h_rep_of[r] == r) ctx->h_pct[r] = -1.0;
  ctx->h_order = ord;
  ctx->stats.n_unique = ctx->U; ctx->stats.n_dropped_short = n - nk;
  ctx->stats.cl_windows = windows; ctx->stats.cl_cuts = cuts; ctx->stats.cl_alignments = (int64_t)naln; ctx->stats.cl_certified = (int64_t)nskip;
  ctx->have_derep = true; ctx->have_search = ctx->have_final = false; ctx->clustered = true;
  ctx->order_cache_ok = false;
  if (ctx->U != ncent) SET_ERR(ctx, ITSX_E_DEVICE, "clustering bookkeeping mismatch (centroids " + std::to_string(ncent) + " vs uniques " + std::to_string(ctx->U) + ")");
  if (n_unique) *n_unique = ctx->U;
  return ITSX_OK;
}

int itsx_get_cluster(const itsx_ctx *ctx, double *pct_id, int64_t *order, int64_t *n_order)
{
  CTXCHK(ctx && ctx->have_derep && ctx->clustered);
  if (pct_id) for (int64_t r = 0; r < ctx->N; r++) pct_id[r] = ctx->h_pct[r];
  if (order) for (size_t i = 0; i < ctx->h_order.size(); i++) order[i] = ctx->h_order[i];
  if (n_order) *n_order = (int64_t)ctx->h_order.size();
  return ITSX_OK;
}

int itsx_get_derep(const itsx_ctx *ctx, int64_t *rep_of, int8_t *strand, int64_t *uniq_of)
{
  CTXCHK(ctx && ctx->have_derep);
  for (int64_t r = 0; r < ctx->N; r++) {
    if (rep_of) rep_of[r] = ctx->h_rep_of[r];
    if (strand) strand[r] = ctx->h_strand[r];
    if (uniq_of) uniq_of[r] = ctx->h_uniq_of[r];
  }
  return ITSX_OK;
}
int itsx_get_uniques(const itsx_ctx *ctx, int64_t *seed_read, int64_t *abundance)
{
  CTXCHK(ctx && ctx->have_derep);
  for (int32_t u = 0; u < ctx->U; u++) { if (seed_read) seed_read[u] = ctx->h_seed_read[u]; if (abundance) abundance[u] = ctx->h_abund[u]; }
  return ITSX_OK;
}


// ---- multi-GPU exact dereplication (SURVEY 8e option 2): keys out, active set in
int itsx_unique_keys(itsx_ctx *ctx, uint64_t seed, uint64_t *fwd, uint64_t *rc)
{
  CTXCHK(ctx && ctx->have_derep && fwd && rc);
  if (ctx->S > 1) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "cross-rank dereplication of a sample batch is not supported: shard whole samples across ranks");
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->N;
  if (n == 0 || ctx->U == 0) return ITSX_OK;
  HIPCHK(ctx->w_hf.alloc((size_t)n + 1)); HIPCHK(ctx->w_hr.alloc((size_t)n + 1));
  launch_hash_reads(ctx->rd, seed, 1, ctx->w_hf.p, ctx->w_hr.p, ctx->st);
  std::vector<uint64_t> hf((size_t)n), hr((size_t)n);
  HIPCHK(hipMemcpyAsync(hf.data(), ctx->w_hf.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(hr.data(), ctx->w_hr.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  // trailing A's are zero bits in the packed words: the length is folded in so that a read and its A-extended copy differ
  for (int32_t u = 0; u < ctx->U; u++) {
    const size_t r = (size_t)ctx->h_seed_read[u];
    const uint64_t lm = (uint64_t)ctx->h_len[r] * 0x9E3779B97F4A7C15ULL;
    fwd[u] = hf[r] ^ lm; rc[u] = hr[r] ^ lm;
  }
  return ITSX_OK;
}

int itsx_set_active_uniques(itsx_ctx *ctx, const uint8_t *active)
{
  CTXCHK(ctx && ctx->have_derep && (active || ctx->U == 0));
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<int32_t> keep; keep.reserve((size_t)ctx->U);
  for (int32_t s = 0; s < ctx->U; s++) { const int32_t u = ctx->h_sorted_uniq[(size_t)s]; if (active[u]) keep.push_back(u); }   // stays length-sorted
  ctx->U_active = (int32_t)keep.size();
  ctx->h_sorted_active = keep;           // the traces and the searched-target count follow the active list
  if (!keep.empty()) {
    HIPCHK(hipMemcpyAsync(ctx->d_sorted_uniq.p, keep.data(), keep.size() * 4, hipMemcpyHostToDevice, ctx->st));
    launch_fill_ulen(ctx->U_active, ctx->d_sorted_uniq.p, ctx->d_seed_read.p, ctx->rd.len, ctx->d_ulen.p, ctx->st);
    HIPCHK(hipStreamSynchronize(ctx->st));
  }
  ctx->have_search = ctx->have_final = false;
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ search
static float msv_score_from_byte(int xj, int tjb)
{
  float sc = ((float)(xj - tjb) - 190.0f);
  sc /= (float)(3.0 / 0.69314718055994529);
  sc -= 3.0f;
  return sc;
}

static int search_chunk(itsx_ctx *ctx, int ci, int32_t u0, int32_t U, int Lcap, double T, double F1, double F3);
static int append_traces(itsx_ctx *ctx);

// Prefix sharing (k_share.hip) for the search that is about to run: the prefix tree of the active uniques (per length and per chunk of
// ctx->s_Uc uniques), the batches whose saved row states fit the slot budget, the processing order (batch, depth, length) and the tree
// in that order.  ctx->share_on = false when it is switched off (ITSX_SHARE=0), cannot apply, or would share less than ITSX_SHARE_MIN
// (default 0.10) of the rows: the search then runs today's schedule.
static int build_share(itsx_ctx *ctx)
{
  hipStream_t st = ctx->st;
  itsx_stats &S = ctx->stats;
  ctx->share_on = false; ctx->sh_batches.clear(); ctx->sh_segk_h.clear();
  S.share_B = 0; S.share_batches = 0; S.share_nodes = S.share_chains = 0; S.ms_share_build = 0; S.share_frac = 0;
  ctx->two_on = false; ctx->Ub = 0; ctx->share_maxrd = 0; ctx->sh_bsegk_h.clear();
  S.two_sided = 0; S.n_joined = 0; S.bwd_chains = 0; S.gamma_nodes = 0; S.bwd_rows = 0; S.join_maxdiff = 0; S.ms_bwd_bound = 0; S.n_bwd_launches = 0; S.two_fwd_rows = S.two_bwd_rows = S.two_rows_full = 0;
  const int32_t U = ctx->U_active, Uc = (int32_t)std::min<int64_t>(ctx->s_Uc, 0x7fffffff), P = ctx->P;
  if (const char *e = sw_get("ITSX_SHARE")) if (atoi(e) == 0) return ITSX_OK;
  // (the A/B switch of the bound pass uses the classic wave list)
  if (sw_get("ITSX_LAZY_EXACT_BOUND")) return ITSX_OK;
  if (U < 2 || U >= (1 << 26) || P <= 0) return ITSX_OK;
  int B = 32;
  if (const char *e = sw_get("ITSX_SHARE_B")) B = atoi(e);
  int logB = 0; while ((1 << logB) < B) logB++;
  if (B < 16 || B > 1024 || (1 << logB) != B) SET_ERR(ctx, ITSX_E_ARG, "ITSX_SHARE_B must be a power of two between 16 and 1024");
  StageTimer tm(st);
  TrieArgs a{};
  a.rd = ctx->rd; a.sorted_uniq = ctx->d_sorted_uniq.p; a.seed_read = ctx->d_seed_read.p; a.U = U; a.Uc = std::max(1, Uc); a.B = B;
  HIPCHK(ctx->sh_counters.alloc(16));
  HIPCHK(hipMemsetAsync(ctx->sh_counters.p, 0, 16 * sizeof(unsigned long long), st));
  a.counters = ctx->sh_counters.p;
  launch_trie_keycount(a, st);
  if (ctx->lazy && ctx->bound_fold) { TrieArgs r0 = a; r0.rev = 1; launch_trie_keycount(r0, st); }      // (the suffix tree uses the same table after the prefix tree)
  unsigned long long hc0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(hc0, ctx->sh_counters.p, sizeof(hc0), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const unsigned long long nkeys = std::max(hc0[4], hc0[6]);
  if (nkeys == 0) return ITSX_OK;
  uint64_t slots = 1024; while (slots < 2 * nkeys) slots <<= 1;
  HIPCHK(ctx->sh_tab.alloc((size_t)slots, true));
  HIPCHK(hipMemsetAsync(ctx->sh_tab.p, 0xFF, (size_t)slots * sizeof(unsigned long long), st));
  HIPCHK(ctx->sh_depth_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_parent_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_mask_s.alloc((size_t)U + 1));
  HIPCHK(ctx->sh_nn_s.alloc((size_t)U + 2)); HIPCHK(ctx->sh_node0_s.alloc((size_t)U + 2)); HIPCHK(ctx->sh_scan.alloc((size_t)scan_tmp_elems((int64_t)U + 2)));
  HIPCHK(hipMemsetAsync(ctx->sh_mask_s.p, 0, ((size_t)U + 1) * sizeof(unsigned long long), st));
  a.tab = ctx->sh_tab.p; a.tmask = slots - 1; a.depth = ctx->sh_depth_s.p; a.parent = ctx->sh_parent_s.p; a.mask = ctx->sh_mask_s.p; a.nn = ctx->sh_nn_s.p;
  launch_trie_insert(a, st); launch_trie_resolve(a, st); launch_trie_link(a, st); launch_trie_count(a, st);
  launch_exclusive_scan(ctx->sh_nn_s.p, ctx->sh_node0_s.p, (int64_t)U + 1, ctx->sh_scan.p, st);
  // ---- two-sided sharing (round 6; lazy searches with the folded bound kernel): the SUFFIX tree in the same table, the joins, the
  // chains' extents (k_share.hip: k_join_*)
  bool two = ctx->lazy && ctx->bound_fold && !(sw_get("ITSX_SHARE_TWO") && atoi(sw_get("ITSX_SHARE_TWO")) == 0);
  TrieArgs ar = a;
  JoinArgs ja{};
  if (two) {
    HIPCHK(ctx->sh_rdepth_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_rparent_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_rmask_s.alloc((size_t)U + 1));
    HIPCHK(ctx->sh_jlev_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_jown_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_endrow_s.alloc((size_t)U + 1));
    HIPCHK(ctx->sh_rsteps_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_rnn_s.alloc((size_t)U + 2)); HIPCHK(ctx->sh_rnode0_s.alloc((size_t)U + 2));
    HIPCHK(hipMemsetAsync(ctx->sh_tab.p, 0xFF, (size_t)slots * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(ctx->sh_rmask_s.p, 0, ((size_t)U + 1) * sizeof(unsigned long long), st));
    ar.rev = 1; ar.depth = ctx->sh_rdepth_s.p; ar.parent = ctx->sh_rparent_s.p; ar.mask = ctx->sh_rmask_s.p; ar.nn = ctx->sh_rnn_s.p;
    launch_trie_insert(ar, st); launch_trie_resolve(ar, st); launch_trie_link(ar, st);
    ja.t = ar; ja.fdepth = ctx->sh_depth_s.p; ja.fmask = ctx->sh_mask_s.p; ja.rmask = ctx->sh_rmask_s.p; ja.jlev = ctx->sh_jlev_s.p; ja.jown = ctx->sh_jown_s.p;
    ja.endrow = ctx->sh_endrow_s.p; ja.rsteps = ctx->sh_rsteps_s.p; ja.counters = ctx->sh_counters.p;
    launch_join_resolve(ja, st);
    for (int r = 63; r >= 1; r--) launch_join_up(ja, r, st);
    launch_join_ends(ja, st);
    launch_popc64(ctx->sh_rmask_s.p, ctx->sh_rnn_s.p, (int64_t)U + 1, U, st);
    launch_exclusive_scan(ctx->sh_rnn_s.p, ctx->sh_rnode0_s.p, (int64_t)U + 1, ctx->sh_scan.p, st);
  }
  // where a batch may start: a new length or a new chunk
  const int32_t ncap = 65536 + U / std::max(1, Uc) + 8;
  HIPCHK(ctx->sh_cuts.alloc((size_t)ncap * 3));
  launch_share_cuts(ctx->d_ulen.p, ctx->sh_node0_s.p, two ? ctx->sh_rnode0_s.p : nullptr, U, a.Uc, ncap, ctx->sh_cuts.p, ctx->sh_counters.p + 5, st);
  int32_t NN = 0, NG = 0;
  unsigned long long hc[16];
  HIPCHK(hipMemcpyAsync(hc, ctx->sh_counters.p, sizeof(hc), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&NN, ctx->sh_node0_s.p + U, sizeof(NN), hipMemcpyDeviceToHost, st));
  if (two) HIPCHK(hipMemcpyAsync(&NG, ctx->sh_rnode0_s.p + U, sizeof(NG), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int maxd = (int)hc[0];
  const double frac = hc[2] ? (double)hc[1] / (double)hc[2] : 0.0;
  S.share_frac = (float)frac;
  double min_frac = 0.10;
  if (const char *e = sw_get("ITSX_SHARE_MIN")) min_frac = atof(e);
  const int64_t ncuts = (int64_t)hc[5];
  if (frac < min_frac || maxd <= 0 || NN <= 0 || ncuts > ncap) { S.ms_share_build = tm.stop(); return ITSX_OK; }
  if (two && (hc[10] == 0 || NG <= 0)) two = false;            // nobody joins: the one-sided schedule
  const int maxrd = two ? (int)hc[13] : 0;
  std::vector<int32_t> cuts((size_t)ncuts * 3);
  HIPCHK(hipMemcpyAsync(cuts.data(), ctx->sh_cuts.p, cuts.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  struct Cut { int32_t s, n, g; };
  std::vector<Cut> cv((size_t)ncuts);
  for (int64_t i = 0; i < ncuts; i++) cv[(size_t)i] = {cuts[(size_t)3 * i], cuts[(size_t)3 * i + 1], two ? cuts[(size_t)3 * i + 2] : 0};
  std::sort(cv.begin(), cv.end(), [](const Cut &x, const Cut &y) { return x.s < y.s; });
  cv.push_back({U, NN, two ? NG : 0});
  // ---- batches: consecutive groups of one chunk while their saved states (for every profile, at the Forward pass's size) fit the budget
  // (a lazy search: the Forward pass's states, up to 48 GB (in the DP slab's memory: below) and a third of what is free; otherwise only the MSV filter shares, its
  // states are a fifth the size, and the full table's rows and slabs want the memory: up to 8 GB and an eighth of what is free)
  const bool fwd_too = ctx->lazy;
  double gb = fwd_too ? (two ? 76.0 : 48.0) : 8.0;
  {
    size_t fr = 0, tot = 0;
    const double held = (double)ctx->w_slab.cap * sizeof(float) + (double)ctx->sh_mslots.cap * sizeof(uint4);
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) gb = std::min(gb, std::max(0.25, ((double)fr + held) / (double)(1ull << 30) / (fwd_too ? 3.0 : 8.0)));
  }
  if (const char *e = sw_get("ITSX_SHARE_GB")) gb = std::max(0.0001, atof(e));
  // Two batches of the MSV filter always run side by side, each on its own half of its (small) buffer.  The Forward pass does so only
  // with ITSX_SHARE_FWD_STREAMS=2: measured, its launches' tails are too short for that to gain anything (2.029 vs 2.024 s per 10 M
  // reads), and overlapping launches make a kernel trace's durations (each stretched by the other) disagree with the wall time
  // ... except in a SMALL search (under 2 M representatives: a GPU's share of a sharded job), where a (batch, depth) launch is a few
  // milliseconds and its tail a tenth of that: 1.25 M reads 244 -> 232 ms, the step 496 -> 476 (ITSX_SHARE_FWD_STREAMS=1 / 2 force one way)
  // (not under an explicit slot budget: two batches side by side need twice the slots)
  const bool two_fwd = sw_get("ITSX_SHARE_FWD_STREAMS") ? atoi(sw_get("ITSX_SHARE_FWD_STREAMS")) >= 2 : (U < 2000000 && !sw_get("ITSX_SHARE_GB"));
  const double state_b = fwd_too ? (two_fwd ? 2.0 : 1.0) * (double)FWD_STATE_Q * sizeof(float4) : 2.0 * (double)MSV_STATE_Q * sizeof(uint4);
  // ... and no more than a job of this size needs: a quarter of all its states at a time still gives every (batch, depth) launch
  // thousands of waves, and device memory is not free to get (20-40 ms per GB in a fresh context: a streamed file's chunks each bring
  // their own -- round 5's first streamed run spent 5 s in hipMalloc for slots its 1.3 M-read chunks filled to a tenth)
  // (a quarter and a twentieth: the groups a batch is made of do not fill four exact quarters, and a fifth batch with the 2 % that are left
  // over is thirty launches of a few thousand waves -- 39 ms of pass A and 20 ms of the filter at 10 M reads for 15 ms' worth of rows)
  if (!sw_get("ITSX_SHARE_GB")) gb = std::min(gb, std::max(1.0, (double)((int64_t)NN + NG) * state_b * (double)P / (double)(1ull << 30) / 3.8));
  // (in steps of half a GB: a budget that follows the free memory byte by byte made every search of a warm context ask for a slab a few
  // KB larger than the one it held -- 20 GB freed and allocated again per search)
  if (gb > 1.0) gb = floor(gb * 2.0) / 2.0;
  const int64_t nodes_max = std::max<int64_t>(1, (int64_t)(gb * (double)(1ull << 30) / (state_b * (double)P)));
  std::vector<int32_t> bstart; std::vector<itsx_ctx::ShareBatch> &bt = ctx->sh_batches;
  {
    size_t i = 0;
    while (i + 1 < cv.size()) {
      size_t j = i + 1;                       // the batch takes groups i .. j - 1
      while (j + 1 < cv.size() && cv[j].s % a.Uc != 0 && ((int64_t)cv[j + 1].n - cv[i].n) + ((int64_t)cv[j + 1].g - cv[i].g) <= nodes_max) j++;
      itsx_ctx::ShareBatch b{};
      b.k0 = cv[i].s; b.k1 = cv[j].s; b.node0 = cv[i].n; b.nnodes = (int64_t)cv[j].n - cv[i].n; b.gnode0 = cv[i].g; b.gnnodes = (int64_t)cv[j].g - cv[i].g;
      b.nsplit = (int32_t)std::min<int64_t>(P, (b.nnodes + b.gnnodes + nodes_max - 1) / nodes_max); if (b.nsplit < 1) b.nsplit = 1;
      bstart.push_back(b.k0); bt.push_back(b);
      i = j;
    }
    bstart.push_back(U);
  }
  const int nb = (int)bt.size();
  if (nb >= (1 << 24)) { bt.clear(); S.ms_share_build = tm.stop(); return ITSX_OK; }
  // a batch that one group overfills takes its profiles in nsplit ranges; if even one profile's states do not fit, nothing is shared
  for (auto &b : bt) if ((double)(b.nnodes + b.gnnodes) * state_b * (double)((P + b.nsplit - 1) / b.nsplit) > 1.5 * gb * (double)(1ull << 30)) { bt.clear(); S.ms_share_build = tm.stop(); return ITSX_OK; }
  // ---- the processing order: stable by (batch, depth, last row) (k_order.hip)
  HIPCHK(upload(ctx->sh_bstart, bstart, st)); HIPCHK(ctx->sh_segk.alloc((size_t)nb * SHARE_SEGS));
  HIPCHK(ctx->sh_uorder.alloc((size_t)U + 1)); HIPCHK(ctx->sh_inv.alloc((size_t)U + 1));
  HIPCHK(ctx->sh_keys.alloc((size_t)U + 1)); HIPCHK(ctx->sh_keys2.alloc((size_t)U + 1)); HIPCHK(ctx->sh_vals.alloc((size_t)U + 1));
  const size_t sort_bytes = order_sort_bytes(U);
  HIPCHK(ctx->sh_sorttmp.alloc(sort_bytes));
  launch_order_keys(ctx->sh_depth_s.p, two ? ctx->sh_endrow_s.p : ctx->d_ulen.p, nullptr, U, ctx->sh_bstart.p, nb, ctx->sh_keys.p, ctx->sh_vals.p, st);
  if (order_sort(ctx->sh_sorttmp.p, sort_bytes, ctx->sh_keys.p, ctx->sh_keys2.p, ctx->sh_vals.p, ctx->sh_uorder.p, U, st) != 0) SET_ERR(ctx, ITSX_E_DEVICE, "prefix sharing: the order's radix sort failed");
  HIPCHK(hipMemsetAsync(ctx->sh_counters.p + 15, 0, sizeof(unsigned long long), st));
  launch_order_segk(ctx->sh_keys2.p, U, nb, ctx->sh_segk.p, ctx->sh_inv.p, ctx->sh_uorder.p, U, ctx->sh_counters.p + 15, st);
  ctx->sh_segk_h.assign((size_t)nb * SHARE_SEGS, 0);
  HIPCHK(hipMemcpyAsync(ctx->sh_segk_h.data(), ctx->sh_segk.p, (size_t)nb * SHARE_SEGS * 4, hipMemcpyDeviceToHost, st));
  // ---- the tree by processing position
  HIPCHK(ctx->sh_depth.alloc((size_t)U + 1)); HIPCHK(ctx->sh_parent.alloc((size_t)U + 1)); HIPCHK(ctx->sh_mask.alloc((size_t)U + 1));
  HIPCHK(ctx->sh_nn.alloc((size_t)U + 2)); HIPCHK(ctx->sh_node0.alloc((size_t)U + 2)); HIPCHK(ctx->sh_order.alloc((size_t)U + 1)); HIPCHK(ctx->sh_ulen.alloc((size_t)U + 1));
  HIPCHK(ctx->sh_endrow.alloc((size_t)U + 1)); HIPCHK(ctx->sh_jlev.alloc((size_t)U + 1)); HIPCHK(ctx->sh_jsrc.alloc((size_t)U + 1)); HIPCHK(ctx->sh_jownb.alloc((size_t)U + 1));
  ShareDev &o = ctx->sh_dev;
  HIPCHK(ctx->sh_src.alloc((size_t)U + 1));
  o.src = ctx->sh_src.p;
  o.depth = ctx->sh_depth.p; o.parent = ctx->sh_parent.p; o.mask = ctx->sh_mask.p; o.nn = ctx->sh_nn.p; o.node0 = ctx->sh_node0.p; o.order = ctx->sh_order.p; o.ulen = ctx->sh_ulen.p;
  o.endrow = ctx->sh_endrow.p; o.jlev = ctx->sh_jlev.p; o.jsrc = ctx->sh_jsrc.p;
  launch_share_permute(a, ctx->d_ulen.p, ctx->sh_uorder.p, ctx->sh_inv.p, two ? ctx->sh_endrow_s.p : nullptr, two ? ctx->sh_jlev_s.p : nullptr, o, st);
  launch_exclusive_scan(ctx->sh_nn.p, ctx->sh_node0.p, (int64_t)U + 1, ctx->sh_scan.p, st);
  launch_share_src(o, U, a.Uc, st);
  int32_t Ub = 0;
  if (two) {
    // ---- the Backward chains: the uniques that save a state for somebody, by (batch, blocks from the end they start at, rows they walk)
    HIPCHK(ctx->sh_border_s.alloc((size_t)U + 1)); HIPCHK(ctx->sh_invb.alloc((size_t)U + 1)); HIPCHK(ctx->sh_bsegk.alloc((size_t)nb * SHARE_SEGS));
    launch_order_keys(ctx->sh_rdepth_s.p, ctx->sh_rsteps_s.p, ctx->sh_rsteps_s.p, U, ctx->sh_bstart.p, nb, ctx->sh_keys.p, ctx->sh_vals.p, st);
    if (order_sort(ctx->sh_sorttmp.p, sort_bytes, ctx->sh_keys.p, ctx->sh_keys2.p, ctx->sh_vals.p, ctx->sh_border_s.p, U, st) != 0) SET_ERR(ctx, ITSX_E_DEVICE, "two-sided sharing: the order's radix sort failed");
    HIPCHK(hipMemsetAsync(ctx->sh_counters.p + 15, 0, sizeof(unsigned long long), st));
    launch_order_segk(ctx->sh_keys2.p, U, nb, ctx->sh_bsegk.p, ctx->sh_invb.p, ctx->sh_border_s.p, U, ctx->sh_counters.p + 15, st);
    unsigned long long nv = 0;
    ctx->sh_bsegk_h.assign((size_t)nb * SHARE_SEGS, 0);
    HIPCHK(hipMemcpyAsync(ctx->sh_bsegk_h.data(), ctx->sh_bsegk.p, (size_t)nb * SHARE_SEGS * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&nv, ctx->sh_counters.p + 15, sizeof(nv), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    Ub = (int32_t)nv;
    HIPCHK(ctx->sh_rdepth.alloc((size_t)Ub + 1)); HIPCHK(ctx->sh_rparent.alloc((size_t)Ub + 1)); HIPCHK(ctx->sh_rmask.alloc((size_t)Ub + 1));
    HIPCHK(ctx->sh_rnn.alloc((size_t)Ub + 2)); HIPCHK(ctx->sh_rnode0.alloc((size_t)Ub + 2)); HIPCHK(ctx->sh_border.alloc((size_t)Ub + 1)); HIPCHK(ctx->sh_bulen.alloc((size_t)Ub + 1));
    HIPCHK(ctx->sh_rsrc.alloc((size_t)Ub + 1)); HIPCHK(ctx->sh_bsteps.alloc((size_t)Ub + 1));
    BShareDev &ob = ctx->sh_bdev;
    ob.depth = ctx->sh_rdepth.p; ob.parent = ctx->sh_rparent.p; ob.mask = ctx->sh_rmask.p; ob.nn = ctx->sh_rnn.p; ob.node0 = ctx->sh_rnode0.p; ob.order = ctx->sh_border.p;
    ob.ulen = ctx->sh_bulen.p; ob.src = ctx->sh_rsrc.p; ob.steps = ctx->sh_bsteps.p;
    launch_bshare_permute(ar, Ub, ctx->d_ulen.p, ctx->sh_border_s.p, ctx->sh_invb.p, ctx->sh_rmask_s.p, ctx->sh_rsteps_s.p, ob, st);
    launch_exclusive_scan(ctx->sh_rnn.p, ctx->sh_rnode0.p, (int64_t)Ub + 1, ctx->sh_scan.p, st);
    launch_bshare_src(ob, Ub, st);
    launch_join_src(U, B, ctx->sh_uorder.p, ctx->sh_jown_s.p, ctx->sh_invb.p, ob, o, ctx->sh_jownb.p, st);
    HIPCHK(ctx->sh_chain.alloc((size_t)U + 1)); HIPCHK(ctx->sh_bchain.alloc((size_t)Ub + 1));
    launch_chain_recs(U, ctx->rd, o.order, ctx->d_seed_read.p, o.src, o.node0, o.mask, o.endrow, o.jlev, o.jsrc, ctx->sh_chain.p, st);
    launch_chain_recs(Ub, ctx->rd, ob.order, ctx->d_seed_read.p, ob.src, ob.node0, ob.mask, ob.steps, nullptr, nullptr, ctx->sh_bchain.p, st);
  }
  HIPCHK(hipStreamSynchronize(st));
  // ---- slots of the saved states: the largest batch's, for the MSV filter and for the Forward pass
  int64_t need_slots = 0, need_gslots = 0;
  for (auto &b : bt) {
    const int64_t pr = (int64_t)((P + b.nsplit - 1) / b.nsplit);
    need_slots = std::max<int64_t>(need_slots, b.nnodes * pr); need_gslots = std::max<int64_t>(need_gslots, b.gnnodes * pr);
  }
  // (no room for them: the search runs unshared)
  ctx->sh_mslots_half = (size_t)need_slots * MSV_STATE_Q;      // two batches of the MSV filter run side by side (search_chunk)
  // The Forward pass's states live in the DP slab (ctx->w_slab): pass A is over before the rounds' Forward / Backward kernels write their
  // rows there, so the two never need the memory at the same time -- 24-48 GB that round 5's first version held twice (and that cost the
  // stages behind it their batch sizes).  Two-sided: a batch's Backward states follow its Forward states.
  const size_t half = ((size_t)need_slots + (size_t)(two ? need_gslots : 0)) * FWD_STATE_Q;
  ctx->sh_gslots_off = (size_t)need_slots * FWD_STATE_Q;
  ctx->sh_fslots_half = (fwd_too && two_fwd && nb > 1) ? half : 0;       // (one batch: nothing to run beside it)
  ctx->sh_mgslots_half = (size_t)need_gslots * MSV_STATE_Q;
  if (ctx->sh_mslots.alloc(2 * (size_t)need_slots * MSV_STATE_Q + 1) != hipSuccess || (two && ctx->sh_mgslots.alloc(2 * (size_t)need_gslots * MSV_STATE_Q + 1) != hipSuccess) ||
      (fwd_too && ctx->w_slab.alloc(((4 * (half * (ctx->sh_fslots_half ? 2 : 1) + 1) + ((size_t)64 << 20) - 1) >> 26) << 26, true) != hipSuccess)) {
    (void)hipGetLastError();
    ctx->sh_mslots.release(); bt.clear();
    S.ms_share_build = tm.stop();
    return ITSX_OK;
  }
  ctx->share_on = true; ctx->share_B = B; ctx->share_logB = logB; ctx->share_maxd = maxd;
  ctx->two_on = two; ctx->Ub = Ub; ctx->share_maxrd = maxrd;
  S.share_B = B; S.share_batches = nb; S.share_nodes = NN; S.share_chains = (int64_t)hc[3];
  S.msv_rows_full = (int64_t)hc[2] * P; S.msv_rows = (int64_t)(hc[2] - hc[1]) * P;
  if (two && !(sw_get("ITSX_MSV_TWO") && atoi(sw_get("ITSX_MSV_TWO")) == 0)) S.msv_rows = (int64_t)(hc[11] + hc[12]) * P;
  if (two) { S.two_sided = 1; S.n_joined = (int64_t)hc[10]; S.bwd_chains = (int64_t)hc[14]; S.gamma_nodes = NG; S.two_fwd_rows = (int64_t)hc[11]; S.two_bwd_rows = (int64_t)hc[12]; S.two_rows_full = (int64_t)hc[2]; }
  S.ms_share_build = tm.stop();
  return ITSX_OK;
}

// the active uniques, chunk by chunk (s_Uc uniques at a time), through search_chunk
static int run_chunks(itsx_ctx *ctx, double T, double F1, double F3)
{
  const int64_t U = ctx->U_active, Uc = std::max<int64_t>(1, ctx->s_Uc);
  int ci = 0;
  if (ctx->st2) HIPCHK(hipStreamSynchronize(ctx->st2));      // a search that failed half-way may have left an MSV launch behind
  ctx->msv_pre_u0 = -1;
  for (int64_t u0 = 0; u0 < U; u0 += Uc, ci++) {
    ctx->next_u0 = (u0 + Uc < U) ? u0 + Uc : -1;
    ctx->next_U = (int32_t)std::min<int64_t>(Uc, U - (u0 + Uc));
    const int rc = search_chunk(ctx, ci, (int32_t)u0, (int32_t)std::min<int64_t>(Uc, U - u0), ctx->s_Lcap, T, F1, F3);
    if (rc != ITSX_OK) return rc;
  }
  ctx->n_chunks = ci;
  return ITSX_OK;
}

int itsx_search(itsx_ctx *ctx, double T, double F1, double F2, double F3)
{
  CTXCHK(ctx);
  if (!ctx->have_derep) SET_ERR(ctx, ITSX_E_ARG, "itsx_search called before itsx_derep / itsx_cluster");
  if (ctx->P <= 0) SET_ERR(ctx, ITSX_E_ARG, "no profiles loaded");
  ctx->F2 = F2; ctx->have_vit = F2 < F1;              // hmmsearch enters the Viterbi filter only for P > F2: never when F1 <= F2
  ctx->switches_at_search = sw_report();              // what the environment asked of this search (itsx_switches)
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->st;
  const int P = ctx->P, G = ctx->G, Ppad = G * 64, U = ctx->U_active;
  ctx->T = T;
  ctx->h_dom.clear(); ctx->h_trace.clear(); ctx->trace_sorted = false;
  ctx->domz.assign((size_t)P * ctx->S, 0);
  itsx_stats &S = ctx->stats;
  S.n_pairs = (int64_t)U * P; S.n_past_msv = S.n_past_bias = S.n_past_fwd = S.n_regions = S.n_multidomain = S.n_domains = S.n_domain_overflow = 0;
  S.ms_msv = S.ms_filters = S.ms_domains = S.ms_msv_kernel = S.ms_fwd_kernel = S.ms_bwd_kernel = S.ms_env_kernel = S.ms_bias_kernel = S.ms_decode_kernel = 0; S.n_batches = 0; S.ms_ensemble = 0; S.n_mr_clustered = S.n_mr_distinct = S.n_mr_failed = S.n_mr_envelopes = 0; S.n_slab_shrinks = 0; S.n_mr_overflow = 0; S.ms_vit_kernel = 0; for (int k = 0; k < 8; k++) S.n_mr_fail_kind[k] = 0; S.n_rows_resident = 0; S.msv_cells = 0; S.msv_launches = 0; S.fwd_rows = 0; S.env_rows = 0; S.n_env_unique = 0;
  ctx->npairs_padded = 0; ctx->dom_n.clear(); ctx->trace_u0 = 0; ctx->n_chunks = 0;
  ctx->have_search = true; ctx->have_final = false; ctx->domz_on_device = false; ctx->topup.valid = false;
  // rows mode: what the caller selected (itsx_set_rows_mode), else the environment (ITSX_ROWS=full|compact|lazy; ITSX_COMPACT_ROWS=1)
  int mode = ctx->rows_mode;
  if (mode < 0) {
    mode = ITSX_ROWS_FULL;
    if (sw_get("ITSX_COMPACT_ROWS") && atoi(sw_get("ITSX_COMPACT_ROWS")) != 0) mode = ITSX_ROWS_COMPACT;
    if (const char *e = sw_get("ITSX_ROWS")) mode = !strcmp(e, "lazy") ? ITSX_ROWS_LAZY : !strcmp(e, "compact") ? ITSX_ROWS_COMPACT : ITSX_ROWS_FULL;
  }
  if (mode == ITSX_ROWS_LAZY && sw_get("ITSX_KEEP_TRACE")) mode = ITSX_ROWS_COMPACT;      // pair traces describe the full pipeline
  ctx->compact_rows = mode != ITSX_ROWS_FULL;
  ctx->lazy = mode == ITSX_ROWS_LAZY;
  ctx->lazy_pending = 0; ctx->domz_exchanged = false; ctx->sF1 = F1; ctx->sF3 = F3;
  ctx->domz_ub.assign((size_t)P * ctx->S, 0);
  if (ctx->compact_rows)                    // a full table of an earlier search (80 B x ~130 per representative) goes back to the device
    for (auto &b : ctx->dom_bufs) if (b && b->cap * sizeof(itsx_domain) > ((size_t)256 << 20)) b->release();
  S.lazy = ctx->lazy; S.n_lazy_evaluated = S.n_lazy_round1 = S.n_lazy_pending = S.n_lazy_reruns = 0; S.n_lazy_completed = S.n_lazy_completed_profiles = S.n_lazy_pending_profiles = 0; S.ms_lazy_complete = 0; S.lazy_bound_maxdiff = 0; S.ms_bound_kernel = S.ms_lazy_select = 0; S.bound_rows = 0; S.n_bound_launches = 0; S.n_lazy_topup = 0; S.ms_lazy_topup = 0; S.ms_lazy_topup_stages = 0;
  if (ctx->compact_rows && U > 0) {
    ctx->compact_zmax = 1e9; ctx->compact_dome_min = 1e-2;          // hmmsearch's --domE is 10 unless given; 1e9 reported targets per profile is a lot of data
    if (const char *e = sw_get("ITSX_COMPACT_ZMAX")) ctx->compact_zmax = std::max(1.0, atof(e));
    if (const char *e = sw_get("ITSX_COMPACT_DOME_MIN")) ctx->compact_dome_min = std::max(1e-300, atof(e));
    ctx->compact_lnp = log(ctx->compact_dome_min / ctx->compact_zmax) - 1e-6;       // certain: exp(lnP) * Zmax <= domE_min, with room for exp's last bit
    // classes = the distinct 2-character NAME prefixes (what create_runtime_hmm and ItsPosition select by: 1_ 2_ 3_ 4_)
    std::vector<int8_t> cls((size_t)P, 0); std::vector<std::string> seen;
    for (int p = 0; p < P; p++) {
      const std::string pre = ctx->profs[p].name.substr(0, 2);
      size_t k = 0; while (k < seen.size() && seen[k] != pre) k++;
      if (k == seen.size()) seen.push_back(pre);
      if (k > 126) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "ITSX_COMPACT_ROWS: more than 127 distinct 2-character profile name prefixes");
      cls[p] = (int8_t)k;
    }
    ctx->compact_ncls = (int)seen.size();
    HIPCHK(upload(ctx->w_cls, cls, st));
    HIPCHK(ctx->w_bestc.alloc((size_t)ctx->U * ctx->compact_ncls + 1));
    HIPCHK(hipMemsetAsync(ctx->w_bestc.p, 0, ((size_t)ctx->U * ctx->compact_ncls + 1) * sizeof(unsigned long long), st));
  }
  if (U == 0) return ITSX_OK;

  // ---- per-length constants (host libm, as hmmsearch computes them per target)
  const int Lcap = ctx->Lmax + 1;
  std::vector<LenTables> lt((size_t)Lcap);
  std::vector<int32_t> tjb((size_t)Lcap, 0);
  std::vector<char> present((size_t)Lcap, 0);
  int64_t cells_active = 0;                                 // residues of the active representatives (the filter's cell count, below)
  {   // two gathers per representative from arrays the size of the read set: a few threads (10 M reads: 25 ms on one, the GPU idle)
    const int32_t Ua = ctx->U, Uact = ctx->U_active;
    const int Th = Ua >= (1 << 19) ? std::max(1, std::min(8, itsx_io::io_threads())) : 1;
    std::vector<std::vector<char>> pres((size_t)Th); std::vector<int64_t> csum((size_t)Th, 0);
    on_threads(Th, [&](int t) {
      std::vector<char> &pr = pres[(size_t)t]; pr.assign((size_t)Lcap, 0);
      int64_t c = 0;
      for (int64_t u = (int64_t)Ua * t / Th, hi = (int64_t)Ua * (t + 1) / Th; u < hi; u++) {
        const int32_t L = ctx->h_len[ctx->h_seed_read[(size_t)u]];
        pr[(size_t)L] = 1;
        if (u < Uact) c += L;
      }
      csum[(size_t)t] = c;
    });
    for (int t = 0; t < Th; t++) { cells_active += csum[(size_t)t]; for (int L = 0; L < Lcap; L++) present[(size_t)L] |= pres[(size_t)t][(size_t)L]; }
  }
  for (int L = 0; L < Lcap; L++) {
    LenTables &t = lt[L]; memset(&t, 0, sizeof(t));
    if (L == 0) continue;
    const float p1 = (float)L / (float)(L + 1);
    t.p1 = p1;
    t.nullsc = (float)((double)(float)L * log((double)p1) + log(1. - (double)p1));
    t.bias_a = (float)L * logf(p1);
    t.bias_b = logf((float)(1. - (double)p1));
    t.lognn3 = log((double)((float)L / (float)(L + 3)));
    t.tjb = host_tjb_b(L);
    tjb[L] = t.tjb;
    {   // lazy domain stage (k_lazy.hip): bits <= (fwdsc - nullsc) / ln 2 + C(L); + 0.02 bits for float rounding, rounded up to float
      const double n = (double)L;
      const double c = 1.0 + (2.0 * log(2.0 * (n + 3.0) / (3.0 * (n + 2.0))) + n * log((n + 3.0) / (n + 2.0))) / 0.69314718055994529;
      // the margin covers the rounding of both float sums (the bound kernel's and HMMER's own: sums of positive products, relative
      // error <= ~3 (n + M) 2^-24 each): 0.02 bits up to ~38 000 residues, growing with n beyond (0.034 bits at 65 535)
      const double margin = std::max(0.02, 6.0 * (n + (double)MMAX) * 5.9604644775390625e-8 / 0.69314718055994529);
      t.lazy_c = nextafterf((float)(c + margin), 1e30f);
    }
    { const float w = roundf((float)(500.0 / 0.69314718055994529) * logf((2.0f + 1.0f) / ((float)L + 2.0f + 1.0f)));
      t.vmove = (w >= 32767.0f) ? 32767 : (w <= -32768.0f) ? -32768 : (int)w; }
  }
  // MSV pass threshold on the final xJ byte, per (length, profile): P(score) <= F1
  // (a row depends on the length, the profiles and F1 only: rows made by an earlier search of this context are kept -- 200 k
  // evaluations of the Gumbel tail, 13 ms per search of the bench's 280 lengths x 88 profiles, while the GPU waited)
  if (ctx->thr_memo_F1 != F1 || ctx->thr_memo_Ppad != Ppad) { ctx->thr_memo.clear(); ctx->thr_memo_have.clear(); ctx->thr_memo_F1 = F1; ctx->thr_memo_Ppad = Ppad; }
  if (ctx->thr_memo_have.size() < (size_t)Lcap) { ctx->thr_memo_have.resize((size_t)Lcap, 0); ctx->thr_memo.resize((size_t)Lcap * Ppad, 257); }
  std::vector<uint16_t> thr((size_t)Lcap * Ppad, 257);
  for (int L = 1; L < Lcap; L++) {
    if (!present[L]) continue;
    if (ctx->thr_memo_have[(size_t)L]) { memcpy(&thr[(size_t)L * Ppad], &ctx->thr_memo[(size_t)L * Ppad], (size_t)Ppad * sizeof(uint16_t)); continue; }
    ctx->thr_memo_have[(size_t)L] = 2;                 // (filled below)
    for (int p = 0; p < P; p++) {
      const HostProfile &h = ctx->profs[p];
      auto passes = [&](int xj) {
        const float usc = msv_score_from_byte(xj, lt[L].tjb);
        const double Pv = gumbel_surv((double)(usc - lt[L].nullsc) / 0.69314718055994529, (double)h.evparam[0], (double)h.evparam[1]);
        return !(Pv > F1);
      };
      int lo = 0, hi = 255;             // smallest xj in [0,254] that passes; 256 if none
      if (!passes(254)) { thr[(size_t)L * Ppad + p] = 256; continue; }
      while (lo < hi) { const int mid = (lo + hi) / 2; if (passes(mid)) hi = mid; else lo = mid + 1; }
      thr[(size_t)L * Ppad + p] = (uint16_t)lo;
    }
  }
  for (int L = 1; L < Lcap; L++)
    if (ctx->thr_memo_have[(size_t)L] == 2) { memcpy(&ctx->thr_memo[(size_t)L * Ppad], &thr[(size_t)L * Ppad], (size_t)Ppad * sizeof(uint16_t)); ctx->thr_memo_have[(size_t)L] = 1; }
  DBuf<uint16_t> &d_thr = ctx->w_thr; DBuf<int32_t> &d_tjb = ctx->w_tjb;
  HIPCHK(upload(ctx->d_lt, lt, st)); HIPCHK(upload(d_thr, thr, st)); HIPCHK(upload(d_tjb, tjb, st));

  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(ctx->d_domz32.alloc((size_t)P * ctx->S));
  HIPCHK(hipMemsetAsync(ctx->d_domz32.p, 0, (size_t)P * ctx->S * 4, st));
  {
    const int64_t cells = cells_active;
    int64_t msum = 0; for (auto &h : ctx->profs) msum += h.M;
    S.msv_cells = cells * msum;
    S.msv_rows = S.msv_rows_full = cells * P; S.bound_rows_full = 0; S.n_share_helpers = 0; S.share_mismatch = 0;
  }
  // ---- the unique reads are searched in chunks so that every work list of a chunk fits a fixed share of HBM
  // (about 400 B per potential (unique, profile) pair); one chunk covers the 1 M-read bench
  if (ctx->pair_budget <= 0) {
    size_t fr = 0, tot = 0;
    double gb = 32.0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) gb = std::max(1.0, (double)fr / (double)(1ull << 30) / 4.0);
    ctx->pair_budget = (int64_t)(gb * (double)(1ull << 30) / 400.0);
  }
  // (the lazy stage keeps ~40 B per pair -- the record, its Forward score, its bound, the selection scan -- and the 400 B only for
  // the pairs it evaluates: chunks can be several times larger)
  int64_t lazy_budget = ctx->pair_budget * 5;
  if (ctx->lazy) {
    // ... but never more than the device can hold NOW: a context that ran the full pipeline before keeps its slabs and work lists
    // (buffers only grow).  ~64 B per pair, a third of what is free plus what the lazy lists hold already.
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      const double held = (double)ctx->d_pairs.cap * sizeof(PairRec) + (double)ctx->l_fb.cap * 4 + (double)ctx->l_b10.cap * 4 + (double)ctx->l_flag.cap * 4 +
                          (double)ctx->l_pos.cap * 4 + (double)ctx->l_done.cap + (double)ctx->w_res.cap * 2;
      lazy_budget = std::min<int64_t>(lazy_budget, (int64_t)(((double)fr + held) / 3.0 / 64.0));
      lazy_budget = std::max<int64_t>(lazy_budget, (int64_t)1 << 20);
    }
  }
  int64_t Uc = std::max<int64_t>(1, (ctx->lazy ? lazy_budget : ctx->pair_budget) / std::max(P, 1));
  Uc = std::min<int64_t>(Uc, ((1ll << 31) - 4096) / std::max(Ppad, 1));
  // (a context that has searched a job of this size already holds the buffers for it -- they only grow --: what is FREE now says nothing
  // about whether the chunk fits, and a second, tiny chunk costs a round of every launch and the top-up round its list)
  if (ctx->lazy && ctx->prev_lazy && ctx->prev_P == P && (int64_t)U <= ctx->prev_U && ctx->prev_Uc > Uc) Uc = ctx->prev_Uc;
  if (const char *e = sw_get("ITSX_CHUNK_UNIQUES")) Uc = std::max<int64_t>(1, atoll(e));
  ctx->keep_trace = sw_get("ITSX_KEEP_TRACE") != nullptr;
  ctx->s_Uc = Uc; ctx->s_Lcap = Lcap;
  ctx->prev_lazy = ctx->lazy; ctx->prev_P = P; ctx->prev_U = U; ctx->prev_Uc = Uc;
  ctx->domz_ub_loc.assign((size_t)P * ctx->S, 0);
  { const int rc = build_share(ctx); if (rc != ITSX_OK) return rc; }
  { const int rc = run_chunks(ctx, T, F1, F3); if (rc != ITSX_OK) return rc; }
  std::vector<int32_t> dz32((size_t)P * ctx->S, 0);
  HIPCHK(hipMemcpyAsync(dz32.data(), ctx->d_domz32.p, dz32.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  ctx->domz_loc.assign(dz32.begin(), dz32.end());
  ctx->domz = ctx->domz_loc; ctx->domz_ub = ctx->domz_ub_loc;
  S.n_rows_resident = 0;
  for (int64_t r : ctx->dom_n) S.n_rows_resident += r;
  // Nothing is capped: a pair's regions beyond its slots go to an overflow list, an ensemble beyond the fast kernel's bookkeeping to the
  // overflow path (k_ensemble.hip).  What remains is what makes hmmsearch itself throw: a region whose Forward matrix cannot be sampled
  // (p7_StochasticTrace's "probabilities were not normalised"): reported, not papered over.
  if (S.n_mr_failed > 0) {
    ctx->have_search = false;
    SET_ERR(ctx, ITSX_E_UNSUPPORTED, std::to_string(S.n_mr_failed) + " multidomain region(s) with a Forward matrix that cannot be sampled (hmmsearch's stochastic traceback throws on them)");
  }
  return ITSX_OK;
}

// a work list of (representative, profile) pairs grouped by profile: 64-aligned segments, ascending length inside a segment
struct PairList {
  PairRec *pairs = nullptr; PairOut *pout = nullptr;
  int64_t NP = 0;
  std::vector<int64_t> seg_start;        // [P + 1]
  std::vector<int32_t> total;            // [P] pairs of each profile
  const int64_t *d_seg_start = nullptr;  // seg_start on the device
};

// wave descriptors of a pair list (fast Q == 12 waves first, then the runtime-Q ones) and each wave's longest target + 1
static int build_waves(itsx_ctx *ctx, const PairList &pl, std::vector<WaveDesc> &waves, std::vector<char> &wgeneric, std::vector<int32_t> &rows)
{
  hipStream_t st = ctx->st;
  const int P = ctx->P;
  waves.clear(); wgeneric.clear();
  for (int pass = 0; pass < 2; pass++)           // fast (Q == 12) waves first, then runtime-Q waves
    for (int p = 0; p < P; p++) {
      if ((int)ctx->generic_q[p] != pass) continue;
      for (int64_t k = 0; k < pl.total[(size_t)p]; k += 64) {
        WaveDesc w{}; w.prof = p; w.first = pl.seg_start[(size_t)p] + k; w.count = (int32_t)std::min<int64_t>(64, pl.total[(size_t)p] - k);
        waves.push_back(w); wgeneric.push_back((char)pass);
      }
    }
  const int NW = (int)waves.size();
  DBuf<WaveDesc> &d_waves = ctx->w_waves; DBuf<int32_t> &d_rows = ctx->w_rows;
  HIPCHK(upload(d_waves, waves, st)); HIPCHK(d_rows.alloc((size_t)std::max(NW, 1)));
  rows.assign((size_t)NW, 0);
  if (NW == 0) return ITSX_OK;
  hipLaunchKernelGGL(k_wave_rows_pairs, dim3((NW + 255) / 256), dim3(256), 0, st, d_waves.p, NW, pl.pairs, d_rows.p);
  HIPCHK(hipMemcpyAsync(rows.data(), d_rows.p, (size_t)NW * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return ITSX_OK;
}

// every stage after the survivor list, for the pairs of `pl`: bias filter + Forward, Backward + decoding + regions, the ensemble
// stage, envelope re-scoring, scores and domZ counts, domain rows (compacted when the context keeps only what can still win)
static int domain_pipeline(itsx_ctx *ctx, const PairList &pl, const int32_t *d_sorted, double T, double F1, double F3, const std::function<int()> *next_msv)
{
  hipStream_t st = ctx->st;
  const int P = ctx->P;
  itsx_stats &S = ctx->stats;
  const int64_t NP = pl.NP;
  const size_t di = ctx->dom_n.size();           // this call's segment of domain rows
  ctx->dom_n.push_back(0);
  while (ctx->dom_bufs.size() <= di) ctx->dom_bufs.emplace_back(new DBuf<itsx_domain>());
  DBuf<itsx_domain> &d_dom = ctx->compact_rows ? ctx->w_domscratch : *ctx->dom_bufs[di];
  if (NP == 0) return ITSX_OK;
  StageTimer tm_list(st);
  std::vector<WaveDesc> waves; std::vector<char> wgeneric; std::vector<int32_t> rows;
  { const int rc = build_waves(ctx, pl, waves, wgeneric, rows); if (rc != ITSX_OK) return rc; }
  const int NW = (int)waves.size();
  DBuf<WaveDesc> &d_waves = ctx->w_waves;
  S.ms_msv += tm_list.stop();

  // ---- batches of waves sized to the slab budget
  if (ctx->slab_gb <= 0.0) {            // the default is decided once per context: a quarter of the free HBM, at most 64 GB
    ctx->slab_gb = 16.0;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) ctx->slab_gb = std::min(64.0, std::max(1.0, (double)fr / (double)(1ull << 30) / 4.0));
  }
  double slab_gb = ctx->slab_gb;
  if (const char *e = sw_get("ITSX_SLAB_GB")) slab_gb = std::max(0.25, atof(e));     // read at every call
  {   // never more than a quarter of what the device has left right now (plus what the context's slab already holds): the stages
      // after the DP kernels need room too.  A budget above that is cut here; one the allocator still refuses is halved below.
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      const double room = ((double)fr / 4.0 + (double)ctx->w_slab.cap * 4.0) / (double)(1ull << 30);
      if (slab_gb > room) { slab_gb = std::max(0.0625, room); S.n_slab_shrinks++; }
    }
  }
  const int64_t row_bytes = 12 * 64 * 4;                   // XF fields of k_float.hip's parser slab
  int64_t budget_rows = (int64_t)(slab_gb * (1 << 30)) / row_bytes;
  // A job that fits a few batches does not need the whole budget: eight batches already keep the launch tails small, and
  // device memory is not free to get (20-40 ms per GB in a fresh process: a 64-GB slab costs more than the search of a
  // 100 k-read sample).  Memory the context holds already is used in full.
  static const bool adapt = !(sw_get("ITSX_SLAB_ADAPT") && atoi(sw_get("ITSX_SLAB_ADAPT")) == 0);
  if (adapt && !sw_get("ITSX_SLAB_GB")) {
    int64_t total_rows = 0;
    for (int w = 0; w < NW; w++) total_rows += rows[(size_t)w];
    const int64_t floor_rows = ((int64_t)4 << 30) / row_bytes, have_rows = (int64_t)(ctx->w_slab.cap / (12 * 64));
    // (memory the context holds is kept as it is while it is at least half of that: growing a buffer is a free and an allocation, and
    // on this driver both are paid at ~30 GB/s -- freed memory is wiped, fresh memory cleared: scripts/alloc_probe.hip)
    const int64_t want_rows = std::max(total_rows / 8 + 1, floor_rows);
    budget_rows = std::min(budget_rows, 2 * have_rows >= want_rows ? have_rows : want_rows);
  }
  DBuf<RegionRec> &d_raw = ctx->w_raw;
  HIPCHK(d_raw.alloc((size_t)NP * MAXDOM));
  // regions past a pair's MAXDOM slots: hmmsearch keeps them all (a concatemer, a CCS read with tandem copies), so they go to an
  // overflow list that k_decode appends to; should even that list run full it is enlarged and the pipeline call repeated (pool_retry)
  if (ctx->pool_cap <= 0) ctx->pool_cap = 1 << 16;
  RegionPoolView pv{nullptr, nullptr, 0};
 pool_retry:
  HIPCHK(ctx->w_pool.alloc((size_t)ctx->pool_cap)); HIPCHK(ctx->w_pool_k.alloc((size_t)ctx->pool_cap)); HIPCHK(ctx->w_pool_n.alloc(2));
  HIPCHK(hipMemsetAsync(ctx->w_pool_n.p, 0, 2 * sizeof(unsigned long long), st));
  const itsx_stats stats_before = S;
  {
    StageTimer tm(st);
    // ---- batches of waves that fit the slab
    struct Batch { int w0, w1; int64_t r; };
    std::vector<Batch> batches;
    DBuf<float> &d_slab = ctx->w_slab;
    for (;;) {
      batches.clear();
      int64_t rmax = 0;
      for (int w0 = 0; w0 < NW;) {
        int w1 = w0; int64_t r = 0;
        while (w1 < NW && wgeneric[w1] == wgeneric[w0] && (w1 == w0 || r + rows[w1] <= budget_rows)) { waves[w1].slab = r; waves[w1].rows = rows[w1]; r += rows[w1]; w1++; }
        batches.push_back(Batch{w0, w1, r});
        rmax = std::max(rmax, r);
        w0 = w1;
      }
      // several batches: take the whole budget, so that the next job (whose fullest batch may be a few waves larger) fits too
      const int64_t slab_rows = batches.size() > 1 ? std::max(rmax, budget_rows) : rmax;
      if ((size_t)slab_rows * 12 * 64 <= d_slab.cap) break;
      if (d_slab.alloc((size_t)slab_rows * 12 * 64, true) == hipSuccess) break;
      // the budget asked for more memory than the device has left (ITSX_SLAB_GB above the free HBM, another tenant):
      // halve it and batch again -- more, smaller launches -- rather than fail the search
      (void)hipGetLastError();
      if (budget_rows <= ((int64_t)64 << 20) / row_bytes) SET_ERR(ctx, ITSX_E_NOMEM, "not even 64 MB of device memory are left for the DP slab");
      budget_rows = std::max<int64_t>(budget_rows / 2, ((int64_t)64 << 20) / row_bytes);
      ctx->slab_gb = std::min(ctx->slab_gb, (double)(budget_rows * row_bytes) / (double)(1 << 30));
      S.n_slab_shrinks++;
    }
    HIPCHK(hipMemcpyAsync(d_waves.p, waves.data(), (size_t)NW * sizeof(WaveDesc), hipMemcpyHostToDevice, st));
    FloatArgs a{};
    a.rd = ctx->rd; a.sorted_uniq = d_sorted; a.seed_read = ctx->d_seed_read.p; a.prof = ctx->d_prof.p; a.lt = ctx->d_lt.p;
    a.flogsum = ctx->d_flogsum.p; a.logtab = ctx->d_logtab.p; a.pairs = pl.pairs; a.pout = pl.pout; a.waves = d_waves.p; a.slab = d_slab.p;
    a.regions = d_raw.p; a.F1 = F1; a.F3 = F3;
    a.pool.rec = ctx->w_pool.p; a.pool.k = ctx->w_pool_k.p; a.pool.n = ctx->w_pool_n.p; a.pool.cap = ctx->pool_cap;
    VitArgs va{};
    if (ctx->have_vit) {
      HIPCHK(ctx->d_vit.alloc((size_t)NP));
      va.rd = ctx->rd; va.sorted_uniq = d_sorted; va.seed_read = ctx->d_seed_read.p; va.prof = ctx->d_prof.p; va.lt = ctx->d_lt.p;
      va.pairs = pl.pairs; va.pout = pl.pout; va.waves = d_waves.p; va.vtab = ctx->d_vtab.p; va.vit = ctx->d_vit.p; va.F2 = ctx->F2;
      va.eloop = (int32_t)roundf((float)(500.0 / 0.69314718055994529) * logf(0.5f));
      a.vit = ctx->d_vit.p;
    }
    // The bias-composition filter of a batch (its own full-occupancy kernel, VALU-bound, ~48 registers) runs on a second
    // stream beside the decoder of the batch before it (latency-bound, ~50 registers): the two share the SIMDs, which
    // the 256-register DP kernels never do with anything.
    static const bool overlap = !(sw_get("ITSX_BIAS_OVERLAP") && atoi(sw_get("ITSX_BIAS_OVERLAP")) == 0);
    if (overlap && !ctx->st2) {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (sw_get("ITSX_ST2_PRIO") && atoi(sw_get("ITSX_ST2_PRIO")) == 0) least = 0;
      HIPCHK(hipStreamCreateWithPriority(&ctx->st2, hipStreamNonBlocking, least));     // what runs there fills gaps, it does not take turns
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
    }
    LazyTimers lazy(st), lazy2(overlap ? ctx->st2 : st);
    auto bias_batch = [&](const Batch &bt, hipStream_t s, LazyTimers &lz) {
      // contiguous runs of pairs (the waves of one pass skip the profiles of the other pass)
      int w = bt.w0;
      while (w < bt.w1) {
        int e = w + 1;
        while (e < bt.w1 && waves[(size_t)e].first == waves[(size_t)e - 1].first + 64) e++;
        FloatArgs ab = a;
        ab.pairs = pl.pairs + waves[(size_t)w].first; ab.pout = pl.pout + waves[(size_t)w].first;
        const size_t t = lz.begin(&S.ms_bias_kernel); launch_bias(ab, (int64_t)(e - w) * 64, s); lz.end(t);
        w = e;
      }
    };
    if (!batches.empty()) bias_batch(batches[0], st, lazy);
    for (size_t bi = 0; bi < batches.size(); bi++) {
      const Batch &bt = batches[bi];
      const int w0 = bt.w0, w1 = bt.w1;
      a.slab_plane = bt.r * 6 * 64;
      const bool more = bi + 1 < batches.size();
      if (ctx->have_vit) { const size_t t = lazy.begin(&S.ms_vit_kernel); launch_vit(va, w1 - w0, w0, st); lazy.end(t); }
      { const size_t t = lazy.begin(&S.ms_fwd_kernel); launch_filters_fwd(a, w1 - w0, w0, wgeneric[w0], st); lazy.end(t); }
      { const size_t t = lazy.begin(&S.ms_bwd_kernel); launch_bwd_decode(a, w1 - w0, w0, wgeneric[w0], st); lazy.end(t); }
      if (more && overlap) {
        HIPCHK(hipEventRecord(ctx->ev_a, st)); HIPCHK(hipStreamWaitEvent(ctx->st2, ctx->ev_a, 0));
        bias_batch(batches[bi + 1], ctx->st2, lazy2);
        HIPCHK(hipEventRecord(ctx->ev_b, ctx->st2));
      }
      { const size_t t = lazy.begin(&S.ms_decode_kernel); launch_decode(a, w1 - w0, w0, st); lazy.end(t); }
      if (more && overlap) HIPCHK(hipStreamWaitEvent(st, ctx->ev_b, 0));
      else if (more) bias_batch(batches[bi + 1], st, lazy);
      for (int w = w0; w < w1; w++) S.fwd_rows += (int64_t)(rows[w] - 1) * waves[w].count;
      S.n_batches++;
    }
    lazy2.collect();
    lazy.collect();
    S.ms_filters += tm.stop();
  }
  {   // the overflow list: ordered by (pair, region number) so that the kernels below find region k >= MAXDOM of a pair by bisection
    unsigned long long pn = 0;
    HIPCHK(hipMemcpyAsync(&pn, ctx->w_pool_n.p, sizeof(pn), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if ((int64_t)pn > ctx->pool_cap) {               // (never seen; a sample made of concatemers would do it)
      while (ctx->pool_cap < (int64_t)pn) ctx->pool_cap *= 2;
      const float keep_ms = S.ms_filters;
      S = stats_before; S.ms_filters = keep_ms; S.n_slab_shrinks++;
      goto pool_retry;
    }
    if (pn > 0) {
      std::vector<RegionRec> hr((size_t)pn); std::vector<int32_t> hk((size_t)pn);
      HIPCHK(hipMemcpyAsync(hr.data(), ctx->w_pool.p, (size_t)pn * sizeof(RegionRec), hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(hk.data(), ctx->w_pool_k.p, (size_t)pn * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      std::vector<size_t> ord((size_t)pn);
      std::iota(ord.begin(), ord.end(), (size_t)0);
      auto key = [&](size_t i) { return ((unsigned long long)(uint32_t)hr[i].pair << 24) | (unsigned long long)(uint32_t)hk[i]; };
      std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return key(x) < key(y); });
      std::vector<RegionRec> sr((size_t)pn); std::vector<unsigned long long> sk((size_t)pn);
      for (size_t i = 0; i < (size_t)pn; i++) { sr[i] = hr[ord[i]]; sk[i] = key(ord[i]); }
      HIPCHK(upload(ctx->w_pool_sorted, sr, st)); HIPCHK(upload(ctx->w_pool_key, sk, st));
      HIPCHK(hipStreamSynchronize(st));
      pv.rec = ctx->w_pool_sorted.p; pv.key = ctx->w_pool_key.p; pv.n = (int64_t)pn;
    }
  }
  // ---- the next chunk's MSV filter, on the second stream: the kernels of the domain stage below wait on memory most of the
  // time (tracebacks, envelope slabs, hashing), the MSV kernel is pure VALU work at 75 registers -- the two share the SIMDs
  if (next_msv) { const int rc = (*next_msv)(); if (rc != ITSX_OK) return rc; }
  StageTimer tm_dom(st);
  // ---- multidomain regions: resolved into envelopes by stochastic traceback clustering (k_ensemble.hip)
  int64_t NMR = 0;
  const bool ensemble = !(sw_get("ITSX_NO_ENSEMBLE") && atoi(sw_get("ITSX_NO_ENSEMBLE")) != 0);
  if (ensemble) {
    DBuf<int32_t> &mrcnt = ctx->w_mrcnt, &mroff = ctx->w_mroff, &stmp = ctx->w_scan2;
    HIPCHK(mrcnt.alloc((size_t)NP + 2)); HIPCHK(mroff.alloc((size_t)NP + 2)); HIPCHK(stmp.alloc((size_t)scan_tmp_elems(NP + 2)));
    launch_mr_count(pl.pout, d_raw.p, pv, NP, mrcnt.p, st);
    launch_exclusive_scan(mrcnt.p, mroff.p, NP + 1, stmp.p, st);
    int32_t tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, mroff.p + NP, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    NMR = tot;
    if (NMR > 0) {
      StageTimer tm_e(st);
      static_assert(sizeof(MrRec) == sizeof(RegionRec), "MrRec is handed to the region memoisation kernels as a RegionRec");
      HIPCHK(ctx->w_mr.alloc((size_t)NMR));
      launch_mr_fill(pl.pout, d_raw.p, pv, NP, mroff.p, ctx->w_mr.p, st);
      // ---- memoisation: distinct (profile, target length, residues) regions
      DBuf<unsigned long long> &rkeys = ctx->w_keys; DBuf<int32_t> &rvals = ctx->w_vals; DBuf<uint32_t> &rslot = ctx->w_slot_of;
      DBuf<int32_t> &rrep = ctx->w_rrep, &ruq = ctx->w_ruq, &rurank = ctx->w_rurank, &rscan = ctx->w_scan_tmp;
      uint64_t tsize = 1024; while (tsize < (uint64_t)NMR * 2 + 16) tsize <<= 1;
      HIPCHK(rkeys.alloc(tsize)); HIPCHK(rvals.alloc(tsize)); HIPCHK(rslot.alloc((size_t)NMR + 1));
      HIPCHK(rrep.alloc((size_t)NMR + 1)); HIPCHK(ruq.alloc((size_t)NMR + 1)); HIPCHK(rurank.alloc((size_t)NMR + 1)); HIPCHK(rscan.alloc((size_t)scan_tmp_elems(NMR + 1)));
      HIPCHK(hipMemsetAsync(rkeys.p, 0, tsize * sizeof(unsigned long long), st));
      HIPCHK(hipMemsetAsync(rvals.p, 0x7f, tsize * sizeof(int32_t), st));
      HIPCHK(hipMemsetAsync(ruq.p, 0, ((size_t)NMR + 1) * sizeof(int32_t), st));
      const RegionRec *mr_as_regions = (const RegionRec *)ctx->w_mr.p;
      launch_region_keys(ctx->rd, mr_as_regions, NMR, pl.pairs, d_sorted, ctx->d_seed_read.p, rkeys.p, rvals.p, tsize - 1, rslot.p, st);
      launch_region_resolve(ctx->rd, mr_as_regions, NMR, pl.pairs, d_sorted, ctx->d_seed_read.p, rvals.p, rslot.p, rrep.p, ruq.p, st);
      launch_exclusive_scan(ruq.p, rurank.p, NMR + 1, rscan.p, st);
      int32_t NU = 0;
      HIPCHK(hipMemcpyAsync(&NU, rurank.p + NMR, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      DBuf<int32_t> &ulist0 = ctx->w_mrulist, &ulist = ctx->w_mrulist2, &ulen = ctx->w_mrlen, &newpos = ctx->w_mrloff, &mru = ctx->w_mru;
      HIPCHK(ulist0.alloc((size_t)NU + 1)); HIPCHK(ulist.alloc((size_t)NU + 1)); HIPCHK(ulen.alloc((size_t)NU + 2)); HIPCHK(newpos.alloc((size_t)NU + 2)); HIPCHK(mru.alloc((size_t)NMR + 1));
      launch_mr_ulist(NMR, ctx->w_mr.p, rrep.p, ruq.p, rurank.p, ulist0.p, ulen.p, mru.p, st);
      // the distinct regions are walked in order of length: the lanes of a wave then have paths and residue loops of like length
      std::vector<int32_t> hlen((size_t)NU), ord((size_t)NU), hpos((size_t)NU);
      HIPCHK(hipMemcpyAsync(hlen.data(), ulen.p, (size_t)NU * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      std::iota(ord.begin(), ord.end(), 0);
      std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return hlen[(size_t)x] < hlen[(size_t)y]; });
      std::vector<int64_t> hn2((size_t)NU + 1, 0), hrow((size_t)NU + 1, 0);
      for (int32_t x = 0; x < NU; x++) {
        const int32_t o = ord[(size_t)x];
        hpos[(size_t)o] = x;
        hn2[(size_t)x + 1] = hn2[(size_t)x] + hlen[(size_t)o];
        hrow[(size_t)x + 1] = hrow[(size_t)x] + hlen[(size_t)o] + 1;
      }
      HIPCHK(hipMemcpyAsync(newpos.p, hpos.data(), (size_t)NU * 4, hipMemcpyHostToDevice, st));
      launch_mr_reorder(NU, NMR, newpos.p, ulist0.p, ulist.p, mru.p, st);
      HIPCHK(upload(ctx->w_n2off, hn2, st)); HIPCHK(upload(ctx->w_mrrowoff, hrow, st));
      HIPCHK(ctx->w_mrout.alloc((size_t)NU)); HIPCHK(ctx->w_n2sc.alloc((size_t)hn2[(size_t)NU] + 1));
      // waves over the regions in ascending length.  The longest regions decide when a batch ends (200 sequential paths each, and a
      // wave walks its lanes' paths in phases, so it takes as long as its slowest lane in EVERY phase): they get waves of
      // MR_LANES_LONG lanes, and the kernels take the waves from the back (longest first).
      static const int lanes_long = sw_get("ITSX_MR_LONG_LANES") ? std::max(1, std::min(MR_LANES, atoi(sw_get("ITSX_MR_LONG_LANES")))) : MR_LANES_LONG;
      static const double frac_long = sw_get("ITSX_MR_LONG_FRAC") ? atof(sw_get("ITSX_MR_LONG_FRAC")) : MR_LONG_FRAC;
      const int64_t n_long = std::min<int64_t>(NU, (int64_t)(frac_long * (double)NU));
      // A lazy search sends few regions here (10 M reads: 1 000 - 8 000 distinct ones per round, against 600 000 of a full table): in waves of
      // 64 they are 16 - 120 waves on a chip of 1 024 SIMDs, each walking 200 x ~125 dependent steps for its lanes' 200 x ~60.  The kernel holds
      // four waves per SIMD: as few lanes per wave as keep every wave resident at once (ITSX_MR_WAVES, 4 096; never under two lanes) --
      // less of the phase divergence, the same latency per step (10 M reads: 215 -> 168 ms per step; a full table's waves stay at 64 lanes)
      const int64_t waves_target = sw_get("ITSX_MR_WAVES") ? atoll(sw_get("ITSX_MR_WAVES")) : 4096;       // (read at every search: the tests walk every layout)
      const int lanes_norm = waves_target <= 0 ? MR_LANES : (int)std::max<int64_t>(2, std::min<int64_t>(MR_LANES, (NU + waves_target - 1) / waves_target));
      std::vector<WaveDesc> mw;
      std::vector<int64_t> wfirst;
      for (int64_t x = 0; x < NU;) {
        const int lanes = x >= NU - n_long ? std::min(lanes_long, lanes_norm) : (int)std::min<int64_t>(lanes_norm, NU - n_long - x);
        WaveDesc d{};
        d.prof = -1; d.first = x; d.count = (int32_t)std::min<int64_t>(lanes, (int64_t)NU - x);
        d.rows = hlen[(size_t)ord[(size_t)(d.first + d.count - 1)]] + 1;                  // ascending length: the last lane's
        wfirst.push_back(x);
        mw.push_back(d);
        x += d.count;
      }
      const int NMW = (int)mw.size();
      wfirst.push_back(NU);
      HIPCHK(upload(ctx->w_mrwaves, mw, st));
      const int64_t mrow_bytes = (int64_t)MRV * 16;
      const int64_t mbudget = std::max<int64_t>(1, (int64_t)(std::min(slab_gb, 16.0) * (1 << 30)) / mrow_bytes);
      const int64_t wave_cap = std::max<int64_t>(1, ((int64_t)6 << 30) / ((int64_t)MR_SCRATCH * MR_LANES));      // 6 GB of bookkeeping blocks per batch
      int w0 = 0;
      int64_t mr_wave_cap = wave_cap;
      while (w0 < NMW) {
        int w1 = w0;
        const int64_t row0 = hrow[(size_t)wfirst[(size_t)w0]];
        auto rows_to = [&](int w) { return hrow[(size_t)wfirst[(size_t)std::min(w, NMW)]] - row0; };
        while (w1 < NMW && w1 - w0 < std::min(wave_cap, mr_wave_cap) && (w1 == w0 || rows_to(w1 + 1) <= mbudget)) w1++;
        const int64_t r = rows_to(w1);
        if (((size_t)r * MRV * 4 > ctx->w_mrslab.cap && ctx->w_mrslab.alloc((size_t)r * MRV * 4) != hipSuccess) ||
            ctx->w_mrscratch.alloc((size_t)(wfirst[(size_t)w1] - wfirst[(size_t)w0]) * MR_SCRATCH) != hipSuccess) {
          (void)hipGetLastError();
          if (w1 - w0 <= 1) SET_ERR(ctx, ITSX_E_NOMEM, "no device memory left for the matrices of one wave of multidomain regions");
          mr_wave_cap = std::max<int64_t>(1, (int64_t)(w1 - w0) / 2);     // half as many regions at a time
          S.n_slab_shrinks++;
          continue;
        }
        MrArgs ma{};
        ma.rd = ctx->rd; ma.sorted_uniq = d_sorted; ma.seed_read = ctx->d_seed_read.p; ma.prof = ctx->d_prof.p; ma.pairs = pl.pairs;
        ma.mr = ctx->w_mr.p; ma.ulist = ulist.p; ma.u0 = wfirst[(size_t)w0]; ma.waves = ctx->w_mrwaves.p; ma.slab = (float4 *)ctx->w_mrslab.p;
        ma.rowoff = ctx->w_mrrowoff.p; ma.rowoff0 = row0;
        ma.n2off = ctx->w_n2off.p; ma.n2sc = ctx->w_n2sc.p; ma.out = ctx->w_mrout.p; ma.scratch = ctx->w_mrscratch.p;
        static const bool mrdbg = sw_get("ITSX_MR_DEBUG") != nullptr;
        if (mrdbg) { HIPCHK(ctx->w_counters.alloc(8)); HIPCHK(hipMemsetAsync(ctx->w_counters.p, 0, 64, st)); ma.dbg = (unsigned long long *)ctx->w_counters.p; }
        // a small batch (a shard of a million reads, a chunk of a streaming run) cannot hide a path's chain of dependent matrix reads
        // behind other waves: its regions are walked one per wave with the matrix in LDS (k_ensemble.hip: k_mr_trace<., true>)
        const int64_t one_max = sw_get("ITSX_MR_ONE_MAX") ? atoll(sw_get("ITSX_MR_ONE_MAX")) : 2048;      // (10 M reads: 6 400 and 21 500 regions in the two rounds took 252 and 531 ms this way, 212 in waves of 64)
        const int64_t nreg = wfirst[(size_t)w1] - wfirst[(size_t)w0];
        const bool one = nreg <= one_max;
        if (one) ma.lds_bytes = (int32_t)std::min<int64_t>(40 << 10, (int64_t)mw[(size_t)w1 - 1].rows * MRV * 16);
        launch_mr_ensemble(ma, w1 - w0, w0, st, one ? wfirst[(size_t)w0] : 0, one ? nreg : 0);
        if (mrdbg) {
          unsigned long long d[5];
          HIPCHK(hipMemcpyAsync(d, ctx->w_counters.p, 40, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
          fprintf(stderr, "[itsx] ensemble batch: %d waves, %lld matrix rows; per wave (100 MHz ticks): walk %.0f, close %.0f (samples %.0f), cluster %.0f\n", w1 - w0, (long long)r,
                  (double)d[0] / std::max<double>(1.0, (double)d[3]), (double)d[1] / std::max<double>(1.0, (double)d[3]), (double)d[4] / std::max<double>(1.0, (double)d[3]),
                  (double)d[2] / std::max<double>(1.0, (double)d[3]));
        }
        w0 = w1;
      }
      // ---- the overflow path: regions whose ensemble did not fit the fast kernel's fixed bookkeeping (more than 8 domains in a
      // sampled path, more than 512 distinct tuples, more than 4 envelopes) run again with per-region arrays sized by their length
      {
        HIPCHK(ctx->w_mrsel.alloc((size_t)NU + 1)); HIPCHK(ctx->w_pool_n.alloc(2));
        HIPCHK(hipMemsetAsync(ctx->w_pool_n.p + 1, 0, sizeof(unsigned long long), st));
        launch_mr_overflowed(ctx->w_mrout.p, NU, ctx->w_mrsel.p, ctx->w_pool_n.p + 1, st);
        unsigned long long nov = 0;
        HIPCHK(hipMemcpyAsync(&nov, ctx->w_pool_n.p + 1, sizeof(nov), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (nov > 0) {
          std::vector<int32_t> sel((size_t)nov);
          HIPCHK(hipMemcpyAsync(sel.data(), ctx->w_mrsel.p, (size_t)nov * 4, hipMemcpyDeviceToHost, st));
          HIPCHK(hipStreamSynchronize(st));
          std::sort(sel.begin(), sel.end());              // ascending position = ascending length (the fast pass's order)
          S.n_mr_overflow += (int64_t)nov;
          int64_t envtot = 0;                             // the envelope lists of all of them: 2 x 4 (Lr + 1) ints each, kept until the chunk is scored
          std::vector<int64_t> envoff(sel.size());
          for (size_t x = 0; x < sel.size(); x++) { envoff[x] = envtot; envtot += 8 * ((int64_t)hlen[(size_t)ord[(size_t)sel[x]]] + 1); }
          HIPCHK(ctx->w_mrenv.alloc((size_t)envtot + 8));
          size_t x0 = 0;
          while (x0 < sel.size()) {                       // batches of regions whose blocks fit 8 GB together (one region at least)
            std::vector<MrBig> big; std::vector<int64_t> srow; std::vector<WaveDesc> bw;
            int64_t arena = 0, rows = 0; size_t x1 = x0;
            while (x1 < sel.size()) {
              const int Lr = hlen[(size_t)ord[(size_t)sel[x1]]];
              MrBig g{}; g.capD = Lr + 1; g.capT = 200 * g.capD;
              uint32_t hs = 1024; while (hs < 2u * (uint32_t)g.capT) hs <<= 1;
              g.hmask = hs - 1;
              auto al = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
              const int64_t capSig = 4 * (int64_t)g.capD;
              const int64_t bytes = al(8ll * g.capT) + 6 * al(4ll * g.capT) + al(4ll * hs) + al(4ll * g.capD) + al(16ll * g.capD) + 3 * al(4 * capSig) +
                                    al(g.capT) + al(2ll * g.capD) + al(capSig) + al(2 * MR_EPC) + 64;
              g.envoff = envoff[x1];
              if (x1 > x0 && arena + bytes > ((int64_t)8 << 30)) break;
              g.off = arena; arena += bytes;
              big.push_back(g); srow.push_back(rows); rows += Lr + 1;
              x1++;
            }
            for (size_t x = x0; x < x1; x += MR_LANES_LONG) {      // few lanes per wave: these are the long, slow regions
              WaveDesc d{}; d.prof = -1; d.first = (int64_t)(x - x0); d.count = (int32_t)std::min<size_t>(MR_LANES_LONG, x1 - x);
              d.rows = hlen[(size_t)ord[(size_t)sel[x + d.count - 1]]] + 1;
              bw.push_back(d);
            }
            std::vector<int32_t> selb(sel.begin() + x0, sel.begin() + x1);
            HIPCHK(upload(ctx->w_mrsel, selb, st)); HIPCHK(upload(ctx->w_mrselrow, srow, st)); HIPCHK(upload(ctx->w_mrbig, big, st)); HIPCHK(upload(ctx->w_mrbigwaves, bw, st));
            if ((size_t)rows * MRV * 4 > ctx->w_mrslab.cap) HIPCHK(ctx->w_mrslab.alloc((size_t)rows * MRV * 4));
            HIPCHK(ctx->w_mrarena.alloc((size_t)arena));
            MrArgs mb{};
            mb.rd = ctx->rd; mb.sorted_uniq = d_sorted; mb.seed_read = ctx->d_seed_read.p; mb.prof = ctx->d_prof.p; mb.pairs = pl.pairs;
            mb.mr = ctx->w_mr.p; mb.ulist = ulist.p; mb.u0 = 0; mb.waves = ctx->w_mrbigwaves.p; mb.slab = (float4 *)ctx->w_mrslab.p;
            mb.rowoff = ctx->w_mrrowoff.p; mb.rowoff0 = 0; mb.n2off = ctx->w_n2off.p; mb.n2sc = ctx->w_n2sc.p; mb.out = ctx->w_mrout.p;
            mb.scratch = nullptr; mb.sel = ctx->w_mrsel.p; mb.sel_rowoff = ctx->w_mrselrow.p; mb.big = ctx->w_mrbig.p; mb.arena = ctx->w_mrarena.p; mb.envpool = ctx->w_mrenv.p;
            launch_mr_ensemble_big(mb, (int)bw.size(), st);
            x0 = x1;
          }
        }
      }
      S.ms_ensemble += tm_e.stop();
      HIPCHK(hipGetLastError());
      S.n_mr_clustered += NMR; S.n_mr_distinct += NU;
    }
  }
  // ---- compact regions into a profile-grouped list
  DBuf<int32_t> &d_rcnt = ctx->w_rcnt, &d_rpref = ctx->w_rpref, &d_scan_tmp = ctx->w_scan2;
  HIPCHK(d_rcnt.alloc((size_t)NP + 1)); HIPCHK(d_rpref.alloc((size_t)NP + 1)); HIPCHK(d_scan_tmp.alloc((size_t)scan_tmp_elems(NP + 1)));
  HIPCHK(hipMemsetAsync(d_rcnt.p, 0, ((size_t)NP + 1) * 4, st));
  RegionListArgs rl{};
  rl.pout = pl.pout; rl.raw = d_raw.p; rl.pv = pv; rl.npairs = NP;
  if (NMR > 0) { rl.mr_off = ctx->w_mroff.p; rl.mr_u = ctx->w_mru.p; rl.mrout = ctx->w_mrout.p; rl.envpool = ctx->w_mrenv.p; }
  {
    DBuf<int64_t> &d_c = ctx->w_counters;
    HIPCHK(d_c.alloc(16));
    HIPCHK(hipMemsetAsync(d_c.p, 0, 16 * sizeof(int64_t), st));
    launch_region_list_count(rl, d_rcnt.p, (unsigned long long *)d_c.p, st);
    int64_t hc[10] = {0};
    HIPCHK(hipMemcpyAsync(hc, d_c.p, sizeof(hc), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    S.n_mr_failed += hc[0]; S.n_mr_envelopes += hc[1];
    for (int k = 0; k < 8; k++) S.n_mr_fail_kind[k] += hc[2 + k];
  }
  launch_exclusive_scan(d_rcnt.p, d_rpref.p, NP + 1, d_scan_tmp.p, st);
  std::vector<int32_t> bound((size_t)P + 1);
  {
    DBuf<int64_t> &d_idx = ctx->w_idx; DBuf<int32_t> &d_b = ctx->w_b;
    HIPCHK(upload(d_idx, pl.seg_start, st)); HIPCHK(d_b.alloc((size_t)P + 1));
    hipLaunchKernelGGL(k_gather_i32, dim3((P + 1 + 255) / 256), dim3(256), 0, st, d_rpref.p, d_idx.p, P + 1, d_b.p);
    HIPCHK(hipMemcpyAsync(bound.data(), d_b.p, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  std::vector<int64_t> rseg((size_t)P + 1, 0);
  std::vector<int32_t> rtotal((size_t)P);
  for (int p = 0; p < P; p++) { rtotal[p] = bound[p + 1] - bound[p]; rseg[p + 1] = rseg[p] + ((int64_t)rtotal[p] + 63) / 64 * 64; S.n_domains += rtotal[p]; }
  const int64_t NR = rseg[P];
  ctx->dom_n[di] = NR;
  HIPCHK(ctx->d_pair_region0.alloc((size_t)NP));
  HIPCHK(ctx->d_regions.alloc((size_t)std::max<int64_t>(NR, 1)));
  HIPCHK(d_dom.alloc((size_t)std::max<int64_t>(NR, 1)));
  HIPCHK(hipMemsetAsync(d_dom.p, 0xFF, (size_t)std::max<int64_t>(NR, 1) * sizeof(itsx_domain), st));
  if (NR > 0) {
    DBuf<int64_t> &d_rseg = ctx->w_rseg;
    HIPCHK(upload(d_rseg, rseg, st));
    launch_region_offsets(NP, pl.pairs, d_rpref.p, pl.d_seg_start, d_rseg.p, ctx->d_pair_region0.p, st);
    HIPCHK(hipMemsetAsync(ctx->d_regions.p, 0xFF, (size_t)NR * sizeof(RegionRec), st));     // padding slots: pair = -1
    launch_region_list_fill(rl, ctx->d_pair_region0.p, ctx->d_regions.p, st);
    // ---- envelope memoisation: only distinct (profile, L, residues) envelopes are re-scored
    DBuf<unsigned long long> &rkeys = ctx->w_keys; DBuf<int32_t> &rvals = ctx->w_vals; DBuf<uint32_t> &rslot = ctx->w_slot_of;
    DBuf<int32_t> &rrep = ctx->w_rrep, &ruq = ctx->w_ruq, &rurank = ctx->w_rurank, &rscan = ctx->w_scan_tmp;
    uint64_t tsize = 1024; while (tsize < (uint64_t)NR * 2 + 16) tsize <<= 1;
    HIPCHK(rkeys.alloc(tsize)); HIPCHK(rvals.alloc(tsize)); HIPCHK(rslot.alloc((size_t)NR + 1));
    HIPCHK(rrep.alloc((size_t)NR + 1)); HIPCHK(ruq.alloc((size_t)NR + 1)); HIPCHK(rurank.alloc((size_t)NR + 1));
    HIPCHK(rscan.alloc((size_t)scan_tmp_elems(NR + 1))); HIPCHK(ctx->d_upos.alloc((size_t)NR + 1));
    HIPCHK(hipMemsetAsync(rkeys.p, 0, tsize * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(rvals.p, 0x7f, tsize * sizeof(int32_t), st));
    HIPCHK(hipMemsetAsync(ruq.p, 0, ((size_t)NR + 1) * sizeof(int32_t), st));
    launch_region_keys(ctx->rd, ctx->d_regions.p, NR, pl.pairs, d_sorted, ctx->d_seed_read.p, rkeys.p, rvals.p, tsize - 1, rslot.p, st);
    launch_region_resolve(ctx->rd, ctx->d_regions.p, NR, pl.pairs, d_sorted, ctx->d_seed_read.p, rvals.p, rslot.p, rrep.p, ruq.p, st);
    launch_exclusive_scan(ruq.p, rurank.p, NR + 1, rscan.p, st);
    std::vector<int32_t> ub((size_t)P + 1);
    {
      DBuf<int64_t> &d_idx = ctx->w_idx; DBuf<int32_t> &d_b = ctx->w_b;
      HIPCHK(upload(d_idx, rseg, st)); HIPCHK(d_b.alloc((size_t)P + 1));
      hipLaunchKernelGGL(k_gather_i32, dim3((P + 1 + 255) / 256), dim3(256), 0, st, rurank.p, d_idx.p, P + 1, d_b.p);
      HIPCHK(hipMemcpyAsync(ub.data(), d_b.p, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    std::vector<int64_t> useg((size_t)P + 1, 0);
    std::vector<int32_t> utotal((size_t)P);
    for (int p = 0; p < P; p++) { utotal[p] = ub[p + 1] - ub[p]; useg[p + 1] = useg[p] + ((int64_t)utotal[p] + 63) / 64 * 64; S.n_env_unique += utotal[p]; }
    const int64_t NU = useg[P];
    DBuf<int64_t> &d_useg = ctx->w_useg;
    HIPCHK(upload(d_useg, useg, st));
    HIPCHK(ctx->d_ulist.alloc((size_t)std::max<int64_t>(NU, 1))); HIPCHK(ctx->d_rout.alloc((size_t)std::max<int64_t>(NU, 1)));
    HIPCHK(hipMemsetAsync(ctx->d_rout.p, 0, (size_t)std::max<int64_t>(NU, 1) * sizeof(RegionOut), st));
    launch_region_upos(NR, ctx->d_regions.p, pl.pairs, ruq.p, rurank.p, d_rseg.p, d_useg.p, ctx->d_upos.p, ctx->d_ulist.p, st);
    launch_region_upos_follow(NR, rrep.p, ruq.p, ctx->d_upos.p, st);
    std::vector<WaveDesc> rw; std::vector<char> rgen;
    for (int pass = 0; pass < 2; pass++)
      for (int p = 0; p < P; p++) {
        if ((int)ctx->generic_q[p] != pass) continue;
        for (int64_t k = 0; k < utotal[p]; k += 64) {
          WaveDesc w{}; w.prof = p; w.first = useg[p] + k; w.count = (int32_t)std::min<int64_t>(64, utotal[p] - k);
          rw.push_back(w); rgen.push_back((char)pass);
        }
      }
    const int NRW = (int)rw.size();
    DBuf<WaveDesc> &d_rw = ctx->w_rw; DBuf<int32_t> &d_rrows = ctx->w_rrows;
    HIPCHK(upload(d_rw, rw, st)); HIPCHK(d_rrows.alloc((size_t)NRW));
    hipLaunchKernelGGL(k_wave_rows_regions, dim3((NRW + 255) / 256), dim3(256), 0, st, d_rw.p, NRW, ctx->d_ulist.p, d_rrows.p);
    std::vector<int32_t> rrows((size_t)NRW);
    HIPCHK(hipMemcpyAsync(rrows.data(), d_rrows.p, (size_t)NRW * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t erow_bytes = 104 * 64 * 4;
    int64_t ebudget = (int64_t)(slab_gb * (1 << 30)) / erow_bytes;
    LazyTimers elazy(st);
    DBuf<float> &d_eslab = ctx->w_eslab; int64_t ealloc = (int64_t)(d_eslab.cap / (104 * 64));
    if (adapt && !sw_get("ITSX_SLAB_GB")) {                  // as for the parser slab: four batches are enough here
      int64_t total_rows = 0;
      for (int w = 0; w < NRW; w++) total_rows += rrows[(size_t)w];
      ebudget = std::min(ebudget, std::max(std::max(total_rows / 4 + 1, ((int64_t)2 << 30) / erow_bytes), ealloc));
    }
    int w0 = 0;
    while (w0 < NRW) {
      int w1 = w0; int64_t r = 0;
      while (w1 < NRW && rgen[w1] == rgen[w0] && (w1 == w0 || r + rrows[w1] <= ebudget)) { rw[w1].slab = r; rw[w1].rows = rrows[w1]; r += rrows[w1]; w1++; }
      if (r > ealloc) {
        if (d_eslab.alloc((size_t)r * 104 * 64) != hipSuccess) {      // as for the parser slab: a budget the device cannot supply is halved
          (void)hipGetLastError();
          ealloc = 0;
          if (w1 - w0 <= 1) SET_ERR(ctx, ITSX_E_NOMEM, "no device memory left for the envelope slab of a single wave");
          ebudget = std::max<int64_t>(r / 2, rrows[(size_t)w0]);
          S.n_slab_shrinks++;
          continue;
        }
        ealloc = r;
      }
      HIPCHK(hipMemcpyAsync(d_rw.p + w0, rw.data() + w0, (size_t)(w1 - w0) * sizeof(WaveDesc), hipMemcpyHostToDevice, st));
      EnvArgs a{};
      a.rd = ctx->rd; a.sorted_uniq = d_sorted; a.seed_read = ctx->d_seed_read.p; a.prof = ctx->d_prof.p; a.lt = ctx->d_lt.p;
      a.pairs = pl.pairs; a.regions = ctx->d_ulist.p; a.rout = ctx->d_rout.p; a.waves = d_rw.p; a.slab = d_eslab.p;
      { const size_t t = elazy.begin(&S.ms_env_kernel); launch_envelopes(a, w1 - w0, w0, rgen[w0], st); elazy.end(t); }
      for (int w = w0; w < w1; w++) S.env_rows += (int64_t)(rrows[w] - 1) * rw[w].count;
      w0 = w1;
    }
    ScoreArgs sa{};
    sa.rd = ctx->rd; sa.sorted_uniq = d_sorted; sa.seed_read = ctx->d_seed_read.p; sa.prof = ctx->d_prof.p; sa.lt = ctx->d_lt.p;
    sa.flogsum = ctx->d_flogsum.p; sa.pairs = pl.pairs; sa.pout = pl.pout; sa.regions = ctx->d_regions.p; sa.rout = ctx->d_rout.p;
    sa.pair_region0 = ctx->d_pair_region0.p; sa.upos = ctx->d_upos.p; sa.dom = d_dom.p; sa.npairs = NP; sa.T = T;
    if (NMR > 0) { sa.mr = ctx->w_mr.p; sa.mr_u = ctx->w_mru.p; sa.mrout = ctx->w_mrout.p; sa.n2off = ctx->w_n2off.p; sa.n2sc = ctx->w_n2sc.p; sa.mr_off = ctx->w_mroff.p; }
    sa.domz = ctx->d_domz32.p; sa.usample = ctx->dev_usample(); sa.P = ctx->P;
    launch_score(sa, st);
    if (ctx->compact_rows) {
      // keep what can still win once domZ is known (k_compact_*), in row order; the chunk's full rows are scratch
      launch_compact_best(d_dom.p, NR, ctx->w_cls.p, ctx->compact_ncls, ctx->compact_lnp, ctx->w_bestc.p, st);
      HIPCHK(ctx->w_keep.alloc((size_t)NR + 1)); HIPCHK(ctx->w_keeppos.alloc((size_t)NR + 1)); HIPCHK(ctx->w_keeptmp.alloc((size_t)scan_tmp_elems(NR + 1)));
      launch_compact_mark(d_dom.p, NR, ctx->w_cls.p, ctx->compact_ncls, ctx->compact_lnp, ctx->w_bestc.p, ctx->w_keep.p, st);
      launch_exclusive_scan(ctx->w_keep.p, ctx->w_keeppos.p, NR + 1, ctx->w_keeptmp.p, st);
      int32_t kept = 0;
      HIPCHK(hipMemcpyAsync(&kept, ctx->w_keeppos.p + NR, sizeof(kept), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      DBuf<itsx_domain> &small = *ctx->dom_bufs[di];
      HIPCHK(small.alloc((size_t)std::max<int32_t>(kept, 1), true));
      launch_compact_scatter(d_dom.p, NR, ctx->w_keep.p, ctx->w_keeppos.p, small.p, st);
      ctx->rows_before_compaction += NR;
      ctx->dom_n[di] = kept;
    }
  }
  S.ms_domains += tm_dom.stop();
  // filter counters
  {
    DBuf<int64_t> &d_c = ctx->w_counters;
    HIPCHK(d_c.alloc(8));
    HIPCHK(hipMemsetAsync(d_c.p, 0, 8 * sizeof(int64_t), st));
    hipLaunchKernelGGL(k_pair_counters, dim3((unsigned)std::min<int64_t>(4096, (NP + 255) / 256)), dim3(256), 0, st, pl.pout, NP, (unsigned long long *)d_c.p);
    int64_t hc[8];
    HIPCHK(hipMemcpyAsync(hc, d_c.p, sizeof(hc), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    S.n_past_bias += hc[0]; S.n_past_fwd += hc[1]; S.n_regions += hc[2]; S.n_domain_overflow += hc[3]; S.n_multidomain += hc[4];
  }
  HIPCHK(hipGetLastError());                                // a kernel of this chunk that failed to launch must not pass silently
  return ITSX_OK;
}


namespace itsx {
__global__ void k_dbg_group_count(const PairRec *__restrict__ pairs, int64_t NP, const int32_t *__restrict__ flag, const int8_t *__restrict__ cls, int ncls, unsigned long long *__restrict__ cnt)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP || !flag[i]) return;
  const PairRec pr = pairs[i];
  if (pr.prof < 0) return;
  atomicAdd(&cnt[(size_t)pr.useq * ncls + cls[pr.prof]], 1ull);
}
__global__ void k_dbg_group_hist(const unsigned long long *__restrict__ cnt, int64_t ng, unsigned long long *__restrict__ hist)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ng) return;
  const unsigned long long c = cnt[i];
  const int b = c <= 4 ? (int)c : c <= 8 ? 5 : c <= 16 ? 6 : 7;
  atomicAdd(&hist[b], 1ull);
  if (c > 1) atomicAdd(&hist[8], c - 1);
  if (c > 2) atomicAdd(&hist[9], c - 2);
}
}  // namespace itsx

// The lazy domain stage (k_lazy.hip): Forward scores for every pair of the chunk, then two rounds of the domain pipeline over
// the pairs that can still win ItsPosition's argmax.
static int lazy_rounds(itsx_ctx *ctx, const PairList &pl, const int32_t *d_sorted, int32_t u0, int32_t Uc, double T, double F1, double F3, const std::function<int()> *next_msv)
{
  hipStream_t st = ctx->st;
  const int P = ctx->P, ncls = ctx->compact_ncls;
  itsx_stats &S = ctx->stats;
  const int64_t NP = pl.NP;
  // ---- pass A: the Forward score of every pair (nothing else is kept)
  HIPCHK(ctx->l_fb.alloc((size_t)NP + 1)); HIPCHK(ctx->l_b10.alloc((size_t)NP + 1)); HIPCHK(ctx->l_done.alloc((size_t)NP + 1));
  HIPCHK(ctx->l_flag.alloc((size_t)NP + 2)); HIPCHK(ctx->l_pos.alloc((size_t)NP + 2)); HIPCHK(ctx->l_scan.alloc((size_t)scan_tmp_elems(NP + 2)));
  HIPCHK(ctx->l_gtop.alloc((size_t)Uc * ncls + 1));
  HIPCHK(hipMemsetAsync(ctx->l_done.p, 0, (size_t)NP + 1, st));
  HIPCHK(hipMemsetAsync(ctx->l_gtop.p, 0, ((size_t)Uc * ncls + 1) * sizeof(unsigned long long), st));
  if (ctx->share_on) {
    // prefix sharing (k_share.hip): the pairs of a profile ascend by processing position = (batch, depth, length); one launch takes the
    // chains of one (batch, depth) for every profile, the depths of a batch in ascending order -- a chain starts from the state that a
    // chain of a lower depth saved for the same profile
    const int maxd = ctx->share_maxd, DS = maxd + 1;
    const bool two = ctx->two_on;
    std::vector<int32_t> segflat, segdepth, segbatch;
    for (size_t bi = 0; bi < ctx->sh_batches.size(); bi++) {
      const auto &b = ctx->sh_batches[bi];
      if (b.k0 < u0 || b.k0 >= u0 + Uc) continue;
      for (int d = 0; d < DS; d++) { segflat.push_back(ctx->sh_segk_h[bi * SHARE_SEGS + d] - u0); segdepth.push_back(d); segbatch.push_back((int32_t)bi); }
    }
    const int nseg = (int)segdepth.size();
    segflat.push_back(Uc);
    if ((int64_t)nseg * P + 1 >= (1ll << 31)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "prefix sharing: too many (batch, depth, profile) segments in one chunk");
    HIPCHK(upload(ctx->sh_segflat, segflat, st)); HIPCHK(upload(ctx->sh_segdepth, segdepth, st, 1)); HIPCHK(upload(ctx->w_rcnt, pl.total, st));
    HIPCHK(ctx->sh_bnd.alloc((size_t)(nseg + 1) * P)); HIPCHK(ctx->sh_wc.alloc((size_t)nseg * P + 2)); HIPCHK(ctx->sh_woff.alloc((size_t)nseg * P + 2));
    HIPCHK(ctx->sh_scan.alloc((size_t)scan_tmp_elems((int64_t)nseg * P + 2)));
    launch_share_bounds(pl.pairs, pl.d_seg_start, ctx->w_rcnt.p, ctx->sh_segflat.p, nseg, P, ctx->sh_bnd.p, st);
    launch_share_wcount(ctx->sh_bnd.p, nseg, P, ctx->sh_wc.p, st);
    launch_exclusive_scan(ctx->sh_wc.p, ctx->sh_woff.p, (int64_t)nseg * P + 1, ctx->sh_scan.p, st);
    std::vector<int32_t> woff((size_t)nseg * P + 1);
    HIPCHK(hipMemcpyAsync(woff.data(), ctx->sh_woff.p, woff.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int NW = woff.back();
    HIPCHK(ctx->w_waves.alloc((size_t)std::max(NW, 1))); HIPCHK(ctx->w_counters.alloc(512));
    HIPCHK(hipMemsetAsync(ctx->w_counters.p, 0, 512 * sizeof(int64_t), st));
    launch_share_waves(NW, nseg, P, ctx->sh_woff.p, ctx->sh_bnd.p, ctx->sh_segdepth.p, ctx->share_B, pl.pairs, two ? ctx->sh_endrow.p + u0 : nullptr, 0, ctx->w_waves.p,
                       (unsigned long long *)ctx->w_counters.p, st);
    std::vector<int64_t> lr((size_t)384, 0);
    FloatArgs a{};
    a.rd = ctx->rd; a.sorted_uniq = d_sorted; a.seed_read = ctx->d_seed_read.p; a.prof = ctx->d_prof.p; a.lt = ctx->d_lt.p;
    a.pairs = pl.pairs; a.waves = ctx->w_waves.p; a.F1 = F1; a.F3 = F3;
    // ---- two-sided sharing (round 6): the Backward chains' work list.  A chain runs for a profile when a pair past the filter joins one of
    // its states, or a chain below it in the suffix tree runs (k_join_need / k_need_up_b); the list is made like the Forward pass's own
    const int maxrd = ctx->share_maxrd, DSB = maxrd + 1;
    int32_t cb0 = 0, Ubc = 0; int NWB = 0;
    std::vector<int32_t> bwoff;
    FloatArgs ab = a;
    if (two) {
      int32_t cb1 = 0; bool any = false;
      std::vector<int32_t> bsegflat, bsegdepth;
      for (size_t bi = 0; bi < ctx->sh_batches.size(); bi++) {
        const auto &b = ctx->sh_batches[bi];
        if (b.k0 < u0 || b.k0 >= u0 + Uc) continue;
        const int32_t lo = ctx->sh_bsegk_h[bi * SHARE_SEGS], hi = ctx->sh_bsegk_h[bi * SHARE_SEGS + SHARE_SEGS - 1];
        if (!any) { cb0 = lo; any = true; }
        cb1 = hi;
      }
      Ubc = any ? cb1 - cb0 : 0;
      for (size_t bi = 0; bi < ctx->sh_batches.size(); bi++) {
        const auto &b = ctx->sh_batches[bi];
        if (b.k0 < u0 || b.k0 >= u0 + Uc) continue;
        for (int d = 0; d < DSB; d++) { bsegflat.push_back(ctx->sh_bsegk_h[bi * SHARE_SEGS + d] - cb0); bsegdepth.push_back(d); }
      }
      const int nbseg = (int)bsegdepth.size();
      bsegflat.push_back(Ubc);
      bwoff.assign((size_t)nbseg * P + 1, 0);
      if (Ubc > 0) {
        const int W = (P + 31) / 32;
        HIPCHK(ctx->sh_needb.alloc((size_t)Ubc * W + 1));
        HIPCHK(hipMemsetAsync(ctx->sh_needb.p, 0, ((size_t)Ubc * W + 1) * 4, st));
        launch_join_need(ctx->sh_jlev.p + u0, ctx->sh_jownb.p + u0, Uc, W, cb0, ctx->sh_pass.p, ctx->sh_needb.p, st);
        for (int r = maxrd; r >= 1; r--) launch_need_up_b(r, ctx->sh_rdepth.p + cb0, ctx->sh_rparent.p + cb0, cb0, Ubc, W, ctx->sh_needb.p, st);
        HIPCHK(ctx->sh_resb.alloc((size_t)P * Ubc + 1));
        launch_need_res(ctx->sh_needb.p, Ubc, P, W, ctx->sh_resb.p, st);
        const int nchb = (Ubc + CHUNK - 1) / CHUNK;
        HIPCHK(ctx->sh_cntb.alloc((size_t)P * nchb)); HIPCHK(ctx->sh_totalb.alloc((size_t)P)); HIPCHK(ctx->sh_realb.alloc((size_t)P));
        HIPCHK(hipMemsetAsync(ctx->sh_realb.p, 0, (size_t)P * 4, st));
        launch_pair_count(ctx->sh_resb.p, P, Ubc, nchb, ctx->sh_cntb.p, ctx->sh_realb.p, st);
        launch_chunk_scan(ctx->sh_cntb.p, P, nchb, ctx->sh_totalb.p, st);
        std::vector<int32_t> totalb((size_t)P);
        HIPCHK(hipMemcpyAsync(totalb.data(), ctx->sh_totalb.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<int64_t> bseg((size_t)P + 1, 0);
        for (int p = 0; p < P; p++) bseg[(size_t)p + 1] = bseg[(size_t)p] + ((int64_t)totalb[(size_t)p] + 63) / 64 * 64;
        const int64_t NPB = bseg[(size_t)P];
        if (NPB >= (1ll << 31) || (int64_t)nbseg * P + 1 >= (1ll << 31)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "two-sided sharing: too many Backward chains in one chunk");
        if (NPB > 0) {
          HIPCHK(upload(ctx->sh_bseg_start, bseg, st));
          HIPCHK(ctx->sh_bpairs.alloc((size_t)NPB));
          HIPCHK(hipMemsetAsync(ctx->sh_bpairs.p, 0xFF, (size_t)NPB * sizeof(PairRec), st));
          launch_pair_fill(ctx->sh_resb.p, P, Ubc, nchb, ctx->sh_cntb.p, ctx->sh_bseg_start.p, ctx->sh_bulen.p + cb0, ctx->sh_bpairs.p, st);
          HIPCHK(upload(ctx->sh_bsegflat, bsegflat, st)); HIPCHK(upload(ctx->sh_bsegdepth, bsegdepth, st, 1));
          HIPCHK(ctx->sh_bbnd.alloc((size_t)(nbseg + 1) * P)); HIPCHK(ctx->sh_bwc.alloc((size_t)nbseg * P + 2)); HIPCHK(ctx->sh_bwoff.alloc((size_t)nbseg * P + 2));
          HIPCHK(ctx->sh_scan.alloc((size_t)scan_tmp_elems((int64_t)nbseg * P + 2)));
          launch_share_bounds(ctx->sh_bpairs.p, ctx->sh_bseg_start.p, ctx->sh_totalb.p, ctx->sh_bsegflat.p, nbseg, P, ctx->sh_bbnd.p, st);
          launch_share_wcount(ctx->sh_bbnd.p, nbseg, P, ctx->sh_bwc.p, st);
          launch_exclusive_scan(ctx->sh_bwc.p, ctx->sh_bwoff.p, (int64_t)nbseg * P + 1, ctx->sh_scan.p, st);
          HIPCHK(hipMemcpyAsync(bwoff.data(), ctx->sh_bwoff.p, bwoff.size() * 4, hipMemcpyDeviceToHost, st));
          HIPCHK(hipStreamSynchronize(st));
          NWB = bwoff.back();
          HIPCHK(ctx->sh_bwaves.alloc((size_t)std::max(NWB, 1)));
          launch_share_waves(NWB, nbseg, P, ctx->sh_bwoff.p, ctx->sh_bbnd.p, ctx->sh_bsegdepth.p, ctx->share_B, ctx->sh_bpairs.p, ctx->sh_bsteps.p + cb0, 1, ctx->sh_bwaves.p,
                             (unsigned long long *)ctx->w_counters.p + 128, st);
          ab.sorted_uniq = ctx->sh_border.p + cb0; ab.pairs = ctx->sh_bpairs.p; ab.waves = ctx->sh_bwaves.p;
        }
      }
    }
    HIPCHK(hipMemcpyAsync(lr.data(), ctx->w_counters.p, 256 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    // diagnostic (scripts/passa_launches.py): every launch's waves and wave-rows, in launch order, to set against a kernel trace
    std::vector<WaveDesc> dumpf, dumpb;
    const bool dump = sw_get("ITSX_PASSA_DUMP") != nullptr;
    if (dump) {
      dumpf.resize((size_t)std::max(NW, 1)); dumpb.resize((size_t)std::max(NWB, 1));
      HIPCHK(hipMemcpyAsync(dumpf.data(), ctx->w_waves.p, (size_t)NW * sizeof(WaveDesc), hipMemcpyDeviceToHost, st));
      if (NWB > 0) HIPCHK(hipMemcpyAsync(dumpb.data(), ctx->sh_bwaves.p, (size_t)NWB * sizeof(WaveDesc), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    auto dump_launch = [&](const char *what, int batch, int d, int w0, int w1, const std::vector<WaveDesc> &wv, int row0) {
      int64_t rows = 0, mx = 0, lanes = 0;
      for (int w = w0; w < w1; w++) { const int64_t r = wv[(size_t)w].rows - 1 - row0; rows += r; mx = std::max(mx, r); lanes += wv[(size_t)w].count; }
      fprintf(stderr, "[passa] %s batch %d depth %d waves %d wave_rows %lld max_rows %lld lanes %lld\n", what, batch, d, w1 - w0, (long long)rows, (long long)mx, (long long)lanes);
    };
    StageTimer tm(st);
    LazyTimers tb(st);                                      // the Backward chains' share of the pass (batches on the main stream)
    // (batches are independent: every other one on a second stream with its own half of the slot buffer, so that a launch's tail --
    // the next depth waits for its last waves -- is filled by the other batch's waves)
    hipStream_t alt = st;
    if (ctx->sh_fslots_half > 0 && nseg > DS) {
      if (!ctx->st3) {
        if (hipStreamCreateWithFlags(&ctx->st3, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->st3 = nullptr; }
        else { (void)hipEventCreateWithFlags(&ctx->ev_s3a, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ctx->ev_s3b, hipEventDisableTiming); }
      }
      if (ctx->st3) { alt = ctx->st3; HIPCHK(hipEventRecord(ctx->ev_s3a, st)); HIPCHK(hipStreamWaitEvent(alt, ctx->ev_s3a, 0)); }
    }
    for (int t0 = 0; t0 < nseg; t0 += DS) {                 // one batch: its profile ranges one after the other, depths ascending
      const auto &b = ctx->sh_batches[(size_t)segbatch[(size_t)t0]];
      const bool on_alt = alt != st && ((t0 / DS) & 1);
      hipStream_t bs = on_alt ? alt : st;
      float4 *fsl = (float4 *)ctx->w_slab.p + (on_alt ? ctx->sh_fslots_half : 0);
      float4 *gsl = fsl + ctx->sh_gslots_off;
      const int tb0 = (t0 / DS) * DSB;
      for (int r = 0; r < b.nsplit; r++) {
        const int pa = (int)((int64_t)P * r / b.nsplit), pb = (int)((int64_t)P * (r + 1) / b.nsplit);
        if (pb <= pa) continue;
        // the batch's Backward chains first, the deepest suffixes last (a chain starts from the state a shorter suffix's chain saved):
        // every state a Forward chain of this batch joins is there before the first of them runs
        const size_t tbi = (two && NWB > 0 && bs == st) ? tb.begin(&S.ms_bwd_bound) : (size_t)-1;
        if (two && NWB > 0)
          for (int d = 0; d < DSB; d++) {
            const int t = tb0 + d;
            const int w0 = bwoff[(size_t)t * P + pa], w1 = bwoff[(size_t)t * P + pb];
            if (w1 <= w0) continue;
            ShareLaunch sl{};
            sl.src = ctx->sh_rsrc.p + cb0; sl.mask = ctx->sh_rmask.p + cb0; sl.node0 = ctx->sh_rnode0.p + cb0; sl.endrow = ctx->sh_bsteps.p + cb0;
            sl.slots = gsl; sl.node_base = b.gnode0; sl.p0 = pa; sl.Pb = pb - pa; sl.depth = d; sl.logB = ctx->share_logB;
            sl.chain = ctx->sh_bchain.p + cb0; sl.entab = ctx->d_entab.p;
            if (sw_get("ITSX_TEST_HOOKS") && sw_get("ITSX_PASSA_DBG")) sl.dbg = atoi(sw_get("ITSX_PASSA_DBG"));
            if (dump) dump_launch("bwd", t0 / DS, d, w0, w1, dumpb, 0);
            for (int w = w0; w < w1; w += 1 << 20) { launch_bwd_bound_share(ab, ctx->d_btab.p, ctx->d_rtab.p, std::min(1 << 20, w1 - w), w, sl, bs); S.n_bwd_launches++; }
          }
        if (tbi != (size_t)-1) tb.end(tbi);
        for (int d = 0; d < DS; d++) {
          const int t = t0 + d;
          const int w0 = woff[(size_t)t * P + pa], w1 = woff[(size_t)t * P + pb];
          if (w1 <= w0) continue;
          ShareLaunch sl{};
          sl.src = ctx->sh_src.p + u0; sl.mask = ctx->sh_mask.p + u0; sl.node0 = ctx->sh_node0.p + u0;
          sl.slots = fsl; sl.node_base = b.node0; sl.p0 = pa; sl.Pb = pb - pa; sl.depth = d; sl.logB = ctx->share_logB;
          if (two) {
            sl.endrow = ctx->sh_endrow.p + u0; sl.jlev = ctx->sh_jlev.p + u0; sl.jsrc = ctx->sh_jsrc.p + u0; sl.gslots = gsl; sl.gnode_base = b.gnode0;
            if (!sw_get("ITSX_NO_CHAINREC")) { sl.chain = ctx->sh_chain.p + u0; sl.entab = ctx->d_entab.p; }
            if (sw_get("ITSX_TEST_HOOKS") && sw_get("ITSX_PASSA_DBG")) sl.dbg = atoi(sw_get("ITSX_PASSA_DBG"));
          }
          if (dump) dump_launch("fwd", t0 / DS, d, w0, w1, dumpf, d << ctx->share_logB);
          for (int w = w0; w < w1; w += 1 << 20) { launch_fwd_bound_share(a, ctx->d_btab.p, ctx->bound_fold, ctx->l_fb.p, std::min(1 << 20, w1 - w), w, sl, bs); S.n_bound_launches++; }
        }
      }
    }
    if (alt != st) { HIPCHK(hipEventRecord(ctx->ev_s3b, alt)); HIPCHK(hipStreamWaitEvent(st, ctx->ev_s3b, 0)); }
    const float ms = tm.stop();
    tb.collect();
    S.ms_bound_kernel += ms; S.ms_filters += ms;
    for (int q = 0; q < 64; q++) { S.bound_rows += lr[(size_t)2 * q]; S.bound_rows_full += lr[(size_t)2 * q + 1]; S.bwd_rows += lr[(size_t)128 + 2 * q]; }
    // the check hooks below run every pair from row 1: their waves reach the pairs' last rows
    const bool want_check = sw_get("ITSX_LAZY_CHECK_BOUND") || (sw_get("ITSX_SHARE_CHECK") && atoi(sw_get("ITSX_SHARE_CHECK")) != 0);
    if (two && want_check)
      launch_share_waves(NW, nseg, P, ctx->sh_woff.p, ctx->sh_bnd.p, ctx->sh_segdepth.p, ctx->share_B, pl.pairs, nullptr, 0, ctx->w_waves.p, (unsigned long long *)ctx->w_counters.p + 256, st);
    if (sw_get("ITSX_LAZY_CHECK_BOUND")) {       // test hook: the shared chains' scores against HMMER's own arithmetic from row 1
      DBuf<float> ref;
      HIPCHK(ref.alloc((size_t)NP + 1));
      for (int t = 0; t < nseg; t++)
        for (int p = 0; p < P; p++) {
          const int w0 = woff[(size_t)t * P + p], w1 = woff[(size_t)t * P + p + 1];
          if (w1 > w0) launch_fwd_bound(a, ref.p, w1 - w0, w0, (int)ctx->generic_q[(size_t)p], st);
        }
      std::vector<float> h0((size_t)NP), h1((size_t)NP); std::vector<PairRec> hp((size_t)NP);
      HIPCHK(hipMemcpyAsync(h0.data(), ref.p, (size_t)NP * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(h1.data(), ctx->l_fb.p, (size_t)NP * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(hp.data(), pl.pairs, (size_t)NP * sizeof(PairRec), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      float mx = S.lazy_bound_maxdiff;
      for (int64_t i = 0; i < NP; i++) {
        if (hp[(size_t)i].prof < 0 || hp[(size_t)i].xj < 0) continue;          // (a pair that ran for its row states only has no score)
        const float x = h0[(size_t)i], y = h1[(size_t)i];
        if (x != x || y != y) { if ((x != x) != (y != y)) mx = 1e30f; continue; }
        mx = std::max(mx, fabsf(x - y));
      }
      S.lazy_bound_maxdiff = mx;
    }
    if (sw_get("ITSX_SHARE_CHECK") && atoi(sw_get("ITSX_SHARE_CHECK")) != 0) {
      // test hook: every pair again from row 1 (the same wave list serves: a wave needs its profile, its pairs and its longest target).  A
      // chain that ran on its own to L has the unshared score bit for bit; a JOINED pair's score is the same sum over paths in another
      // order of operations: it must agree within 2e-3 nats (join_maxdiff reports the largest difference)
      HIPCHK(ctx->sh_fb_chk.alloc((size_t)NP + 1));
      for (int w0 = 0; w0 < NW; w0 += 1 << 20) launch_fwd_bound_seq(a, ctx->d_btab.p, ctx->bound_fold, ctx->sh_fb_chk.p, std::min(1 << 20, NW - w0), w0, st);
      HIPCHK(hipMemsetAsync(ctx->sh_counters.p + 8, 0, 2 * sizeof(unsigned long long), st));
      launch_diff_scores(ctx->l_fb.p, ctx->sh_fb_chk.p, pl.pairs, NP, two ? ctx->sh_jlev.p + u0 : nullptr, ctx->sh_counters.p + 8, st);
      unsigned long long nd[2] = {0, 0};
      HIPCHK(hipMemcpyAsync(nd, ctx->sh_counters.p + 8, sizeof(nd), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      S.share_mismatch += (int64_t)nd[0];
      S.join_maxdiff = std::max(S.join_maxdiff, __builtin_bit_cast(float, (uint32_t)nd[1]));
    }
  } else {
    // the waves over every pair, fast (Q == 12) profiles first, then runtime-Q ones: built on the device (k_waves_build)
    std::vector<int32_t> order; std::vector<int64_t> woff(1, 0);
    int64_t nfast64 = 0;
    for (int pass = 0; pass < 2; pass++)
      for (int p = 0; p < P; p++) {
        if ((int)ctx->generic_q[p] != pass) continue;
        order.push_back(p);
        woff.push_back(woff.back() + ((int64_t)pl.total[(size_t)p] + 63) / 64);
        if (pass == 0) nfast64 = woff.back();
      }
    if (woff.back() >= (1ll << 31)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 waves of pairs in one chunk");
    const int NW = (int)woff.back(), nfast = (int)nfast64;
    DBuf<int32_t> &d_order = ctx->w_b; DBuf<int64_t> &d_woff = ctx->w_idx; DBuf<int32_t> &d_tot = ctx->w_rcnt;
    HIPCHK(upload(d_order, order, st)); HIPCHK(upload(d_woff, woff, st)); HIPCHK(upload(d_tot, pl.total, st));
    HIPCHK(ctx->w_waves.alloc((size_t)std::max(NW, 1))); HIPCHK(ctx->w_counters.alloc(16));
    HIPCHK(hipMemsetAsync(ctx->w_counters.p, 0, 16 * sizeof(int64_t), st));
    if (NW > 0) hipLaunchKernelGGL(k_waves_build, dim3((NW + 255) / 256), dim3(256), 0, st, NW, (int)order.size(), d_order.p, d_woff.p, pl.d_seg_start, d_tot.p,
                                   pl.pairs, ctx->w_waves.p, (unsigned long long *)ctx->w_counters.p);
    int64_t lane_rows = 0;
    HIPCHK(hipMemcpyAsync(&lane_rows, ctx->w_counters.p, sizeof(lane_rows), hipMemcpyDeviceToHost, st));
    FloatArgs a{};
    a.rd = ctx->rd; a.sorted_uniq = d_sorted; a.seed_read = ctx->d_seed_read.p; a.prof = ctx->d_prof.p; a.lt = ctx->d_lt.p;
    a.pairs = pl.pairs; a.waves = ctx->w_waves.p; a.F1 = F1; a.F3 = F3;
    StageTimer tm(st);
    // launches of at most 2^20 waves: the timers of bench.py's roofline block want more than one sample
    static const bool exact_bound = sw_get("ITSX_LAZY_EXACT_BOUND") && atoi(sw_get("ITSX_LAZY_EXACT_BOUND")) != 0;     // A/B: HMMER's own Forward as the bound pass
    if (!exact_bound) {
      for (int w0 = 0; w0 < NW; w0 += 1 << 20) { launch_fwd_bound_seq(a, ctx->d_btab.p, ctx->bound_fold, ctx->l_fb.p, std::min(1 << 20, NW - w0), w0, st); S.n_bound_launches++; }
    } else {
      for (int w0 = 0; w0 < nfast; w0 += 1 << 20) { launch_fwd_bound(a, ctx->l_fb.p, std::min(1 << 20, nfast - w0), w0, 0, st); S.n_bound_launches++; }
      if (NW > nfast) { launch_fwd_bound(a, ctx->l_fb.p, NW - nfast, nfast, 1, st); S.n_bound_launches++; }
    }
    const float ms = tm.stop();
    S.ms_bound_kernel += ms; S.ms_filters += ms;
    if (sw_get("ITSX_LAZY_CHECK_BOUND") && !exact_bound) {      // test hook: the fast kernel's scores against HMMER's own arithmetic
      DBuf<float> ref;
      HIPCHK(ref.alloc((size_t)NP + 1));
      for (int w0 = 0; w0 < nfast; w0 += 1 << 20) launch_fwd_bound(a, ref.p, std::min(1 << 20, nfast - w0), w0, 0, st);
      if (NW > nfast) launch_fwd_bound(a, ref.p, NW - nfast, nfast, 1, st);
      std::vector<float> h0((size_t)NP), h1((size_t)NP); std::vector<PairRec> hp((size_t)NP);
      HIPCHK(hipMemcpyAsync(h0.data(), ref.p, (size_t)NP * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(h1.data(), ctx->l_fb.p, (size_t)NP * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(hp.data(), pl.pairs, (size_t)NP * sizeof(PairRec), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      float mx = S.lazy_bound_maxdiff;
      for (int64_t i = 0; i < NP; i++) {
        if (hp[(size_t)i].prof < 0) continue;
        const float x = h0[(size_t)i], y = h1[(size_t)i];
        if (x != x || y != y) { if ((x != x) != (y != y)) mx = 1e30f; continue; }
        mx = std::max(mx, fabsf(x - y));
      }
      S.lazy_bound_maxdiff = mx;
    }
    S.bound_rows += lane_rows; S.bound_rows_full += lane_rows;     // (its copy was waited for with the kernel's timer)
  }
  StageTimer tm_sel(st);
  LazyArgs la{};
  la.pairs = pl.pairs; la.NP = NP; la.fb = ctx->l_fb.p; la.lt = ctx->d_lt.p; la.cls = ctx->w_cls.p; la.ncls = ncls; la.sorted_uniq = d_sorted;
  la.b10 = ctx->l_b10.p; la.gtop = ctx->l_gtop.p; la.bestc = ctx->w_bestc.p; la.done = ctx->l_done.p; la.flag = ctx->l_flag.p;
  launch_lazy_bound(la, st);
  float ms_sel = tm_sel.stop();
  for (int round = 0; round < 2; round++) {
    StageTimer tm_r(st);
    launch_lazy_mark(la, round, st);
    if (round == 1 && sw_get("ITSX_LAZY_HIST")) {      // experiment: how many candidates a group sends into round 2
      const int64_t ng = (int64_t)Uc * ncls;
      HIPCHK(hipMemsetAsync(ctx->l_gtop.p, 0, (size_t)ng * 8, st));
      hipLaunchKernelGGL(k_dbg_group_count, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, pl.pairs, NP, ctx->l_flag.p, ctx->w_cls.p, ncls, ctx->l_gtop.p);
      DBuf<unsigned long long> hist; HIPCHK(hist.alloc(16)); HIPCHK(hipMemsetAsync(hist.p, 0, 16 * 8, st));
      hipLaunchKernelGGL(k_dbg_group_hist, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, ctx->l_gtop.p, ng, hist.p);
      unsigned long long h[16];
      HIPCHK(hipMemcpyAsync(h, hist.p, sizeof(h), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
      fprintf(stderr, "[itsx] round 2 candidates per group: 0:%llu 1:%llu 2:%llu 3:%llu 4:%llu 5-8:%llu 9-16:%llu 17+:%llu; pairs beyond a group's first %llu, beyond its second %llu\n",
              h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
    }
    launch_exclusive_scan(ctx->l_flag.p, ctx->l_pos.p, NP + 1, ctx->l_scan.p, st);
    std::vector<int32_t> bound((size_t)P + 1);
    {
      DBuf<int64_t> &d_idx = ctx->w_idx; DBuf<int32_t> &d_b = ctx->w_b;
      HIPCHK(upload(d_idx, pl.seg_start, st)); HIPCHK(d_b.alloc((size_t)P + 1));
      hipLaunchKernelGGL(k_gather_i32, dim3((P + 1 + 255) / 256), dim3(256), 0, st, ctx->l_pos.p, d_idx.p, P + 1, d_b.p);
      HIPCHK(hipMemcpyAsync(bound.data(), d_b.p, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    PairList sub;
    sub.seg_start.assign((size_t)P + 1, 0); sub.total.assign((size_t)P, 0);
    for (int p = 0; p < P; p++) { sub.total[(size_t)p] = bound[(size_t)p + 1] - bound[(size_t)p]; sub.seg_start[(size_t)p + 1] = sub.seg_start[(size_t)p] + ((int64_t)sub.total[(size_t)p] + 63) / 64 * 64; }
    sub.NP = sub.seg_start[(size_t)P];
    const int64_t nsel = (int64_t)bound[(size_t)P] - bound[0];
    const bool last = round == 1;
    if (sub.NP == 0) {
      ms_sel += tm_r.stop();
      if (last && next_msv) { const int rc = (*next_msv)(); if (rc != ITSX_OK) return rc; }
      continue;
    }
    HIPCHK(ctx->l_pairs.alloc((size_t)sub.NP)); HIPCHK(ctx->d_pout.alloc((size_t)sub.NP));
    HIPCHK(hipMemsetAsync(ctx->l_pairs.p, 0xFF, (size_t)sub.NP * sizeof(PairRec), st));
    HIPCHK(hipMemsetAsync(ctx->d_pout.p, 0, (size_t)sub.NP * sizeof(PairOut), st));
    HIPCHK(upload(ctx->l_seg, sub.seg_start, st));
    launch_lazy_scatter(pl.pairs, NP, ctx->l_flag.p, ctx->l_pos.p, pl.d_seg_start, ctx->l_seg.p, ctx->l_pairs.p, st);
    sub.pairs = ctx->l_pairs.p; sub.pout = ctx->d_pout.p; sub.d_seg_start = ctx->l_seg.p;
    ms_sel += tm_r.stop();
    S.n_lazy_evaluated += nsel;
    if (round == 0) S.n_lazy_round1 += nsel;
    { const int rc = domain_pipeline(ctx, sub, d_sorted, T, F1, F3, last ? next_msv : nullptr); if (rc != ITSX_OK) return rc; }
  }
  S.ms_lazy_select += ms_sel; S.ms_filters += ms_sel;
  // (the list, the bounds and the done flags stay where they are: the top-up round of itsx_search_finalize reads them when this was the
  // search's only chunk)
  ctx->topup.valid = !ctx->completing; ctx->topup.NP = NP; ctx->topup.seg_start = pl.seg_start; ctx->topup.total = pl.total; ctx->topup.u0 = u0; ctx->topup.Uc = Uc;
  return ITSX_OK;
}

// one chunk [u0, u0+U) of the length-sorted unique list through every stage up to per-sequence reporting
static int search_chunk(itsx_ctx *ctx, int ci, int32_t u0, int32_t U, int Lcap, double T, double F1, double F3)
{
  (void)ci;
  hipStream_t st = ctx->st;
  const int P = ctx->P, G = ctx->G, Ppad = G * 64;
  itsx_stats &S = ctx->stats;
  // PairRec::useq is relative to the chunk; with prefix sharing it is the PROCESSING position: (batch, depth, length) order
  const bool sh = ctx->share_on;
  const int32_t *d_sorted = (sh ? ctx->sh_order.p : ctx->d_sorted_uniq.p) + u0;
  const int32_t *d_ulen = (sh ? ctx->sh_ulen.p : ctx->d_ulen.p) + u0;
  DBuf<uint16_t> &d_thr = ctx->w_thr; DBuf<int32_t> &d_tjb = ctx->w_tjb;
  ctx->trace_u0 = u0;
  // ---- MSV for every (unique, profile)
  DBuf<uint16_t> &d_res = ctx->w_res;
  HIPCHK(d_res.alloc((size_t)Ppad * std::max<int64_t>(U, ctx->next_u0 >= 0 ? ctx->next_U : 0)));
  auto msv_for = [&](int64_t cu0, int32_t cU, hipStream_t s, int lds_pad = 0, uint16_t *res_out = nullptr, bool shared = true) {
    MsvArgs a{};
    a.rd = ctx->rd; a.sorted_uniq = (sh ? ctx->sh_order.p : ctx->d_sorted_uniq.p) + cu0; a.seed_read = ctx->d_seed_read.p; a.U = cU; a.G = G;
    a.etab = ctx->d_etab.p; a.pbias = ctx->d_pbias.p; a.ptec = ctx->d_ptec.p; a.ptbm = ctx->d_ptbm.p;
    a.thr = d_thr.p; a.tjb = d_tjb.p; a.Lcap = Lcap; a.res = res_out ? res_out : d_res.p;
    a.P = P;
    // blocks of 256 sequences x PB profiles: a few thousand blocks at least, and up to 32 profiles per block so that a
    // large job re-reads its sequences' packed words (from L2) 32 times less often than it has profiles
    const int64_t tiles = ((int64_t)cU + 255) / 256;
    if (ctx->completing) {           // only the listed profiles are filtered; every other row of res must read "not passed"
      a.plist = ctx->d_plist.p; a.nlist = (int32_t)ctx->h_plist.size();
      (void)hipMemsetAsync(a.res, 0, (size_t)Ppad * (size_t)cU * sizeof(uint16_t), s);
    }
    const int np = a.plist ? a.nlist : P;
    a.PB = (int)std::max<int64_t>(1, std::min<int64_t>(32, tiles * np / 4096));
    // (the completion filters a handful of profiles: its launches would be a hundred tiny ones per chunk -- it runs unshared, on the same
    // processing order)
    if (!sh || !shared || ctx->completing) { launch_msv(a, s, lds_pad); return; }
    // prefix sharing: batch by batch (the saved states of one batch fit the slot buffer), depth by depth (a chain starts from a state
    // that a chain of a lower depth saved: launches of one stream run in order).  Batches are independent of each other: every other
    // one runs on a second stream with a slot buffer of its own, so that the tail of one launch -- the next depth waits for its last
    // blocks -- is filled by the other batch's blocks (and blocks may take their sequences through more profiles: fewer re-reads)
    static const bool two = !(sw_get("ITSX_SHARE_MSV_STREAMS") && atoi(sw_get("ITSX_SHARE_MSV_STREAMS")) < 2);
    int nbatch_here = 0;
    for (const auto &b : ctx->sh_batches) nbatch_here += (b.k0 >= cu0 && b.k0 < cu0 + cU);
    hipStream_t alt = s;
    if (two && nbatch_here > 1) {
      if (!ctx->st3) {
        if (hipStreamCreateWithFlags(&ctx->st3, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->st3 = nullptr; }
        else { (void)hipEventCreateWithFlags(&ctx->ev_s3a, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ctx->ev_s3b, hipEventDisableTiming); }
      }
      if (ctx->st3) { alt = ctx->st3; (void)hipEventRecord(ctx->ev_s3a, s); (void)hipStreamWaitEvent(alt, ctx->ev_s3a, 0); }
    }
    // two-sided sharing (round 6): the filter is max-plus arithmetic on integers, so a chain that stops where its suffix is shared and
    // takes the rest from the saved Backward state (k_msv_bwd) gets the unshared kernel's xJ exactly (ITSX_SHARE_CHECK counts cells)
    static const bool msv_two = !(sw_get("ITSX_MSV_TWO") && atoi(sw_get("ITSX_MSV_TWO")) == 0);
    const bool bi2 = ctx->two_on && msv_two;
    int32_t cb0 = 0;
    if (bi2) { bool any = false; for (const auto &b : ctx->sh_batches) if (b.k0 >= cu0 && b.k0 < cu0 + cU && !any) { cb0 = ctx->sh_bsegk_h[(size_t)(&b - ctx->sh_batches.data()) * SHARE_SEGS]; any = true; } }
    int seen = 0;
    for (const auto &b : ctx->sh_batches) {
      if (b.k0 < cu0 || b.k0 >= cu0 + cU) continue;
      const size_t bi = (size_t)(&b - ctx->sh_batches.data());
      const bool on_alt = alt != s && (seen++ & 1);
      hipStream_t bs = on_alt ? alt : s;
      uint4 *slots = ctx->sh_mslots.p + (on_alt ? ctx->sh_mslots_half : 0);
      uint4 *gslots = bi2 ? ctx->sh_mgslots.p + (on_alt ? ctx->sh_mgslots_half : 0) : nullptr;
      for (int r = 0; r < b.nsplit; r++) {
        const int pa = (int)((int64_t)np * r / b.nsplit), pb = (int)((int64_t)np * (r + 1) / b.nsplit);
        if (pb <= pa) continue;
        if (bi2)
          for (int d = 0; d <= ctx->share_maxrd; d++) {       // the batch's Backward chains, shortest suffixes first
            const int32_t k0 = ctx->sh_bsegk_h[bi * SHARE_SEGS + d], k1 = ctx->sh_bsegk_h[bi * SHARE_SEGS + d + 1];
            if (k1 <= k0) continue;
            MsvArgs c = a;
            c.sorted_uniq = ctx->sh_border.p + cb0; c.U = 0; c.res = nullptr;
            c.k0 = (int32_t)(k0 - cb0); c.k1 = (int32_t)(k1 - cb0); c.pfirst = pa; c.plast = pb; c.share = 2;
            if (sw_get("ITSX_TEST_HOOKS") && sw_get("ITSX_PASSA_DBG")) c.sl.dbg = atoi(sw_get("ITSX_PASSA_DBG"));
            c.sl.src = ctx->sh_rsrc.p + cb0; c.sl.mask = ctx->sh_rmask.p + cb0; c.sl.node0 = ctx->sh_rnode0.p + cb0; c.sl.endrow = ctx->sh_bsteps.p + cb0;
            c.sl.slots = gslots; c.sl.node_base = b.gnode0; c.sl.p0 = pa; c.sl.Pb = pb - pa; c.sl.depth = d; c.sl.logB = ctx->share_logB;
            const int64_t t2 = ((int64_t)(k1 - k0) + 255) / 256;
            c.PB = (int)std::max<int64_t>(1, std::min<int64_t>(32, t2 * (pb - pa) / (alt != s ? 4096 : 16384)));
            launch_msv(c, bs, lds_pad);
          }
        for (int d = 0; d <= ctx->share_maxd; d++) {
          const int32_t k0 = ctx->sh_segk_h[bi * SHARE_SEGS + d], k1 = ctx->sh_segk_h[bi * SHARE_SEGS + d + 1];
          if (k1 <= k0) continue;
          MsvArgs c = a;
          c.k0 = (int32_t)(k0 - cu0); c.k1 = (int32_t)(k1 - cu0); c.pfirst = pa; c.plast = pb; c.share = 1;
          c.sl.src = ctx->sh_src.p + cu0; c.sl.mask = ctx->sh_mask.p + cu0; c.sl.node0 = ctx->sh_node0.p + cu0;
          c.sl.slots = slots; c.sl.node_base = b.node0; c.sl.p0 = pa; c.sl.Pb = pb - pa; c.sl.depth = d; c.sl.logB = ctx->share_logB;
          if (bi2) { c.sl.endrow = ctx->sh_endrow.p + cu0; c.sl.jlev = ctx->sh_jlev.p + cu0; c.sl.jsrc = ctx->sh_jsrc.p + cu0; c.sl.gslots = gslots; c.sl.gnode_base = b.gnode0; }
          // (a launch ends when its last blocks end, and the next depth waits for it: many short blocks -- rounds of the ~1 000-1 500 a
          // chip holds -- rather than a few long ones that take their sequences through 32 profiles; with the other batch's launches
          // beside it a block may be four times as long)
          const int64_t t2 = ((int64_t)(k1 - k0) + 255) / 256;
          c.PB = (int)std::max<int64_t>(1, std::min<int64_t>(32, t2 * (pb - pa) / (alt != s ? 4096 : 16384)));
          launch_msv(c, bs, lds_pad);
        }
      }
    }
    if (alt != s) { (void)hipEventRecord(ctx->ev_s3b, alt); (void)hipStreamWaitEvent(s, ctx->ev_s3b, 0); }
  };
  if (ctx->msv_pre_u0 == (int64_t)u0) {
    // launched on st2 while the chunk before this one was in its domain stage
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_msv1, 0));
    HIPCHK(hipEventSynchronize(ctx->ev_msv1));
    float ms = 0; (void)hipEventElapsedTime(&ms, ctx->ev_msv0, ctx->ev_msv1);
    S.ms_msv_kernel += ms;                       // its stretched wall time beside other kernels: not extra step time
    S.msv_launches += 1;
    ctx->msv_pre_u0 = -1;
  } else {
    StageTimer tm(st);
    msv_for(u0, U, st);
    const float ms = tm.stop();
    S.ms_msv_kernel += ms; S.ms_msv += ms;
    S.msv_launches += 1;
  }
  StageTimer tm_list(st);
  // ---- survivor list grouped by profile (64-aligned segments, ascending length)
  const int nchunks = (U + CHUNK - 1) / CHUNK;
  DBuf<int32_t> &d_cnt = ctx->w_cnt, &d_total = ctx->w_total;
  HIPCHK(d_cnt.alloc((size_t)P * nchunks)); HIPCHK(d_total.alloc((size_t)P));
  const bool lazy_now = ctx->lazy && !ctx->completing;
  if (sh && sw_get("ITSX_SHARE_CHECK") && atoi(sw_get("ITSX_SHARE_CHECK")) != 0) {
    // test hook: the same chunk through the unshared kernel; every cell of the result must be the same
    HIPCHK(ctx->sh_res_chk.alloc((size_t)Ppad * (size_t)U));
    msv_for(u0, U, st, 0, ctx->sh_res_chk.p, false);
    HIPCHK(hipMemsetAsync(ctx->sh_counters.p + 8, 0, sizeof(unsigned long long), st));
    launch_diff_u16(d_res.p, ctx->sh_res_chk.p, (int64_t)P * U, ctx->sh_counters.p + 8, st);
    unsigned long long nd = 0;
    HIPCHK(hipMemcpyAsync(&nd, ctx->sh_counters.p + 8, sizeof(nd), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    S.share_mismatch += (int64_t)nd;
  }
  if (sh && lazy_now) {
    // pass A runs a pair that failed the filter when a chain below it passed: it needs the row states (PairRec::xj = -1)
    const int W = (P + 31) / 32;
    HIPCHK(ctx->sh_pass.alloc((size_t)U * W + 1)); HIPCHK(ctx->sh_need.alloc((size_t)U * W + 1));
    launch_need_bits(d_res.p, U, P, W, ctx->sh_pass.p, ctx->sh_need.p, st);
    for (int d = ctx->share_maxd; d >= 1; d--) launch_need_up(d, ctx->sh_depth.p + u0, ctx->sh_parent.p + u0, U, W, ctx->sh_need.p, st);
    HIPCHK(hipMemsetAsync(ctx->sh_counters.p + 9, 0, sizeof(unsigned long long), st));
    launch_need_mark(d_res.p, U, P, W, ctx->sh_pass.p, ctx->sh_need.p, ctx->sh_counters.p + 9, st);
  }
  HIPCHK(ctx->sh_real.alloc((size_t)P));
  HIPCHK(hipMemsetAsync(ctx->sh_real.p, 0, (size_t)P * 4, st));
  launch_pair_count(d_res.p, P, U, nchunks, d_cnt.p, ctx->sh_real.p, st);
  launch_chunk_scan(d_cnt.p, P, nchunks, d_total.p, st);
  std::vector<int32_t> total((size_t)P), real((size_t)P);       // pairs on the list / pairs past the filter (the same without sharing)
  HIPCHK(hipMemcpyAsync(total.data(), d_total.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(real.data(), ctx->sh_real.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<int64_t> seg_start((size_t)P + 1, 0);
  for (int p = 0; p < P; p++) { seg_start[p + 1] = seg_start[p] + ((int64_t)total[p] + 63) / 64 * 64; if (ctx->completing) S.n_lazy_completed += real[p]; else S.n_past_msv += real[p]; S.n_share_helpers += total[p] - real[p]; }
  const int64_t NP = seg_start[P];
  ctx->npairs_padded = ctx->lazy ? 0 : NP;
  if (NP == 0) { S.ms_msv += tm_list.stop(); ctx->dom_n.push_back(0); while (ctx->dom_bufs.size() < ctx->dom_n.size()) ctx->dom_bufs.emplace_back(new DBuf<itsx_domain>()); return ITSX_OK; }
  if (NP >= (1ll << 31)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 surviving (representative, profile) pairs");
  DBuf<int64_t> &d_seg_start = ctx->w_seg_start;
  HIPCHK(upload(d_seg_start, seg_start, st));
  HIPCHK(ctx->d_pairs.alloc((size_t)NP));
  HIPCHK(hipMemsetAsync(ctx->d_pairs.p, 0xFF, (size_t)NP * sizeof(PairRec), st));
  if (!lazy_now) {                                 // (the lazy stage keeps a PairOut only for the pairs it evaluates)
    HIPCHK(ctx->d_pout.alloc((size_t)NP));
    HIPCHK(hipMemsetAsync(ctx->d_pout.p, 0, (size_t)NP * sizeof(PairOut), st));
  }
  launch_pair_fill(d_res.p, P, U, nchunks, d_cnt.p, d_seg_start.p, d_ulen, ctx->d_pairs.p, st);
  S.ms_msv += tm_list.stop();
  const int64_t zub_scale = sw_get("ITSX_LAZY_ZUB_SCALE") ? std::max<int64_t>(1, atoll(sw_get("ITSX_LAZY_ZUB_SCALE"))) : 1;   // test hook: looser bounds, more undecided rows
  if (!ctx->completing)
    for (int p = 0; p < P; p++)                   // an upper bound of hmmsearch's domZ: every reported target is a pair past the MSV filter
      for (int32_t sm = 0; sm < ctx->S; sm++) ctx->domz_ub_loc[(size_t)sm * P + p] += (int64_t)real[p] * zub_scale;
  PairList pl;
  pl.pairs = ctx->d_pairs.p; pl.pout = lazy_now ? nullptr : ctx->d_pout.p; pl.NP = NP; pl.seg_start = seg_start; pl.total = total; pl.d_seg_start = d_seg_start.p;
  // the next chunk's MSV filter on the second stream (its result buffer is free: this chunk's survivor list is built)
  const std::function<int()> next_msv = [&]() -> int {
    const bool msv_overlap = !(sw_get("ITSX_MSV_OVERLAP") && atoi(sw_get("ITSX_MSV_OVERLAP")) == 0);     // (read at every search: bench.py times one step without the overlap)
    if (msv_overlap && ctx->next_u0 >= 0 && ctx->st2 && !ctx->keep_trace) {
      if (!ctx->ev_msv0) {
        HIPCHK(hipEventCreate(&ctx->ev_msv0)); HIPCHK(hipEventCreate(&ctx->ev_msv1));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_c, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(ctx->ev_c, st)); HIPCHK(hipStreamWaitEvent(ctx->st2, ctx->ev_c, 0));
      HIPCHK(hipEventRecord(ctx->ev_msv0, ctx->st2));
      msv_for(ctx->next_u0, ctx->next_U, ctx->st2, sw_get("ITSX_MSV_PAD") ? atoi(sw_get("ITSX_MSV_PAD")) : 40000);
      HIPCHK(hipEventRecord(ctx->ev_msv1, ctx->st2));
      ctx->msv_pre_u0 = ctx->next_u0;
    }
    return ITSX_OK;
  };
  if (lazy_now) return lazy_rounds(ctx, pl, d_sorted, u0, U, T, F1, F3, &next_msv);
  { const int rc = domain_pipeline(ctx, pl, d_sorted, T, F1, F3, &next_msv); if (rc != ITSX_OK) return rc; }
  if (ctx->keep_trace) { const int rc = append_traces(ctx); if (rc != ITSX_OK) return rc; }
  return ITSX_OK;
}

int itsx_set_rows_mode(itsx_ctx *ctx, int mode)
{
  CTXCHK(ctx);
  if (mode < -1 || mode > ITSX_ROWS_LAZY) SET_ERR(ctx, ITSX_E_ARG, "itsx_set_rows_mode: mode must be -1 (environment), 0 (full), 1 (compact) or 2 (lazy)");
  ctx->rows_mode = mode;
  return ITSX_OK;
}
int64_t itsx_lazy_pending(const itsx_ctx *ctx) { return ctx ? ctx->lazy_pending : -1; }
int itsx_lazy_pending_profiles(const itsx_ctx *ctx, int32_t *flags)
{
  CTXCHK(ctx && flags && ctx->have_search);
  for (int p = 0; p < ctx->P; p++) flags[p] = (ctx->lazy && (size_t)p < ctx->lazy_pending_prof.size()) ? (ctx->lazy_pending_prof[(size_t)p] != 0) : 0;
  return ITSX_OK;
}
int itsx_set_partial_coords(itsx_ctx *ctx, int on) { CTXCHK(ctx); ctx->partial_coords = on != 0; return ITSX_OK; }
int itsx_set_kept_rows(itsx_ctx *ctx, int on) { CTXCHK(ctx); if (ctx->kept_rows != (on != 0)) ctx->h_dom.clear(); ctx->kept_rows = on != 0; return ITSX_OK; }
// flags[n_unique]: 1 where an undecided row could still change the representative's coordinates (the rows itsx_lazy_pending counts)
int itsx_lazy_pending_uniques(itsx_ctx *ctx, uint8_t *flags)
{
  CTXCHK(ctx && flags && ctx->have_search);
  if (ctx->U <= 0) return ITSX_OK;
  if (!ctx->lazy || !ctx->have_final || ctx->l_uflag.n < (size_t)ctx->U) { memset(flags, 0, (size_t)ctx->U); return ITSX_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(flags, ctx->l_uflag.p, (size_t)ctx->U, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}
// Exact counters for the flagged profiles: EVERY pair of theirs past the MSV filter goes through the domain pipeline (the ones the
// lazy rounds evaluated already come again -- their rows are duplicates with the same rank key, which changes no argmax), so that the
// reported targets of these profiles are counted, not bounded.  Afterwards this context's lower and upper counters of the flagged
// profiles are equal; a multi-rank driver exchanges the counters again before it finalizes.
int itsx_lazy_complete(itsx_ctx *ctx, const int32_t *flags)
{
  CTXCHK(ctx && flags && ctx->have_search);
  if (!ctx->lazy) SET_ERR(ctx, ITSX_E_ARG, "itsx_lazy_complete: the last search was not a lazy one");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->st;
  const int P = ctx->P;
  ctx->h_plist.clear();
  for (int p = 0; p < P; p++) if (flags[p]) ctx->h_plist.push_back(p);
  if (ctx->h_plist.empty() || ctx->U_active == 0) return ITSX_OK;
  StageTimer tm(st);
  HIPCHK(upload(ctx->d_plist, ctx->h_plist, st));
  // the flagged profiles are counted afresh (k_score adds one per reported target)
  std::vector<int32_t> dz32((size_t)P * ctx->S, 0);
  HIPCHK(hipMemcpyAsync(dz32.data(), ctx->d_domz32.p, dz32.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int32_t sm = 0; sm < ctx->S; sm++) for (int p : ctx->h_plist) dz32[(size_t)sm * P + p] = 0;
  HIPCHK(hipMemcpyAsync(ctx->d_domz32.p, dz32.data(), dz32.size() * 4, hipMemcpyHostToDevice, st));
  ctx->completing = true;
  const int rc = run_chunks(ctx, ctx->T, ctx->sF1, ctx->sF3);
  ctx->completing = false;
  if (rc != ITSX_OK) { ctx->have_search = false; return rc; }
  HIPCHK(hipMemcpyAsync(dz32.data(), ctx->d_domz32.p, dz32.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int32_t sm = 0; sm < ctx->S; sm++)
    for (int p : ctx->h_plist) {
      const size_t z = (size_t)sm * P + p;
      ctx->domz_loc[z] = ctx->domz_ub_loc[z] = dz32[z];
      if (!ctx->domz_exchanged) ctx->domz[z] = ctx->domz_ub[z] = dz32[z];
    }
  ctx->have_final = false;
  ctx->stats.n_lazy_completed_profiles += (int64_t)ctx->h_plist.size();
  ctx->stats.n_rows_resident = 0;
  for (int64_t r : ctx->dom_n) ctx->stats.n_rows_resident += r;
  ctx->stats.ms_lazy_complete += tm.stop();
  return ITSX_OK;
}

// After a lazy search the counters are BOUNDS on hmmsearch's domZ: [S][P] reported targets among the evaluated pairs (below),
// then [S][P] pairs past the MSV filter (above).  itsx_get_domz / itsx_set_domz move the lower half only when the search was
// not lazy; after a lazy search they move both halves (2 x S x P values), like itsx_domz_device.
int itsx_get_domz(const itsx_ctx *ctx, int64_t *domZ)
{
  CTXCHK(ctx && domZ && ctx->have_search);
  // this context's OWN counters (what a multi-rank driver sums), whatever itsx_set_domz / an in-place reduction put in their place
  for (size_t p = 0; p < ctx->domz_loc.size(); p++) domZ[p] = ctx->domz_loc[p];
  if (ctx->lazy) for (size_t p = 0; p < ctx->domz_ub_loc.size(); p++) domZ[ctx->domz_loc.size() + p] = ctx->domz_ub_loc[p];
  return ITSX_OK;
}
int itsx_set_domz(itsx_ctx *ctx, const int64_t *domZ)
{
  CTXCHK(ctx && domZ && ctx->have_search);
  for (size_t p = 0; p < ctx->domz.size(); p++) ctx->domz[p] = domZ[p];
  if (ctx->lazy) for (size_t p = 0; p < ctx->domz_ub.size(); p++) ctx->domz_ub[p] = domZ[ctx->domz.size() + p];
  ctx->domz_on_device = false; ctx->domz_exchanged = true;
  return ITSX_OK;
}
int64_t itsx_domz_count(const itsx_ctx *ctx) { return ctx ? (int64_t)ctx->domz.size() * (ctx->lazy ? 2 : 1) : -1; }

// ---- device-resident exchange (multi-GPU): the caller reduces / gathers these buffers where they are (RCCL), nothing bounces
// through host arrays.  Pointers stay valid until the next call that recomputes the same quantity.
int itsx_domz_device(itsx_ctx *ctx, int64_t **d_domz, int64_t *n)
{
  CTXCHK(ctx && d_domz && ctx->have_search);
  HIPCHK(hipSetDevice(ctx->device));
  const size_t m = ctx->domz.size(), mm = m * (ctx->lazy ? 2 : 1);
  HIPCHK(ctx->d_domz64.alloc(std::max<size_t>(mm, 1)));
  // (this context's OWN counters: after an exchange ctx->domz holds the ranks' sums, and a second exchange -- after
  // itsx_lazy_complete -- must not add them up again)
  if (m) HIPCHK(hipMemcpyAsync(ctx->d_domz64.p, ctx->domz_loc.data(), m * 8, hipMemcpyHostToDevice, ctx->st));
  if (m && ctx->lazy) HIPCHK(hipMemcpyAsync(ctx->d_domz64.p + m, ctx->domz_ub_loc.data(), m * 8, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->domz_on_device = true; ctx->domz_exchanged = true;
  *d_domz = ctx->d_domz64.p;
  if (n) *n = (int64_t)mm;
  return ITSX_OK;
}

// thresholds of a lazy search: rows both bounds decide alike are final; ctx->lazy_pending = undecided rows that could matter
static int finalize_lazy(itsx_ctx *ctx, double domE)
{
  hipStream_t st = ctx->st;
  const size_t m = ctx->domz.size();
  DBuf<int64_t> &d_dz = ctx->domz_on_device ? ctx->d_domz64 : ctx->w_dz;
  if (ctx->domz_on_device) {             // reduced in place by the caller: lower bounds, then upper bounds
    if (m) { HIPCHK(hipMemcpyAsync(ctx->domz.data(), d_dz.p, m * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(ctx->domz_ub.data(), d_dz.p + m, m * 8, hipMemcpyDeviceToHost, st)); }
  } else {
    std::vector<int64_t> both(ctx->domz); both.insert(both.end(), ctx->domz_ub.begin(), ctx->domz_ub.end());
    HIPCHK(upload(d_dz, both, st, 2));
  }
  const int32_t U = ctx->U; const int ncls = std::max(ctx->compact_ncls, 1);
  HIPCHK(ctx->l_sure.alloc((size_t)U * ncls + 1)); HIPCHK(ctx->l_has.alloc((size_t)U + 1)); HIPCHK(ctx->w_counters.alloc(16));
  HIPCHK(hipMemsetAsync(ctx->l_sure.p, 0, ((size_t)U * ncls + 1) * sizeof(unsigned long long), st));
  HIPCHK(hipMemsetAsync(ctx->l_has.p, 0, ((size_t)U + 1) * sizeof(int32_t), st));
  HIPCHK(hipMemsetAsync(ctx->w_counters.p, 0, 16 * sizeof(int64_t), st));
  HIPCHK(ctx->l_pflag.alloc((size_t)std::max(ctx->P, 1)));
  HIPCHK(hipMemsetAsync(ctx->l_pflag.p, 0, (size_t)std::max(ctx->P, 1) * 4, st));
  HIPCHK(ctx->l_uflag.alloc((size_t)U + 1));
  HIPCHK(hipMemsetAsync(ctx->l_uflag.p, 0, (size_t)U + 1, st));
  {   // per profile: what would settle its undecided rows from below / from above (lazy_topup); the split between the two lies 60 % of the way up
    const size_t Pn = (size_t)std::max(ctx->P, 1);
    std::vector<unsigned long long> init(2 * Pn), split(Pn, 0);
    for (size_t p = 0; p < Pn; p++) {
      init[2 * p] = 0; init[2 * p + 1] = ~0ull;
      if (p < ctx->domz.size() && ctx->S == 1) {
        double lo = (double)ctx->domz[p], hi = (double)std::max(ctx->domz_ub[p], ctx->domz[p]);
        // (no more targets than the profile has pairs on the last search's list, whatever the upper bound says: a multi-rank sum, a test's inflated bound)
        if (ctx->topup.valid && ctx->n_chunks == 1 && p < ctx->topup.total.size()) hi = std::min(hi, std::max(lo, (double)ctx->topup.total[p]));
        split[p] = (unsigned long long)(lo + 0.6 * (hi - lo));
      }
    }
    HIPCHK(upload(ctx->l_zneed, init, st)); HIPCHK(upload(ctx->l_zsplit, split, st));
  }
  for (size_t c = 0; c < ctx->dom_n.size(); c++)
    if (ctx->dom_n[c] > 0) launch_finalize_lazy(ctx->dom_bufs[c]->p, ctx->dom_n[c], d_dz.p, d_dz.p + m, domE, ctx->dev_usample(), ctx->P, st);
  for (size_t c = 0; c < ctx->dom_n.size(); c++)
    if (ctx->dom_n[c] > 0) launch_lazy_sure(ctx->dom_bufs[c]->p, ctx->dom_n[c], ctx->w_cls.p, ncls, ctx->l_sure.p, ctx->l_has.p, st);
  for (size_t c = 0; c < ctx->dom_n.size(); c++)
    if (ctx->dom_n[c] > 0) launch_lazy_pending(ctx->dom_bufs[c]->p, ctx->dom_n[c], ctx->w_cls.p, ncls, ctx->l_sure.p, ctx->l_has.p, (unsigned long long *)ctx->w_counters.p, ctx->l_pflag.p, ctx->l_uflag.p, ctx->S == 1 ? ctx->l_zneed.p : nullptr, ctx->l_zsplit.p, domE, st);
  int64_t pend = 0;
  ctx->h_zneed.assign((size_t)std::max(ctx->P, 1) * 2, 0);
  HIPCHK(hipMemcpyAsync(ctx->h_zneed.data(), ctx->l_zneed.p, (size_t)std::max(ctx->P, 1) * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&pend, ctx->w_counters.p, sizeof(pend), hipMemcpyDeviceToHost, st));
  ctx->lazy_pending_prof.assign((size_t)std::max(ctx->P, 1), 0);
  HIPCHK(hipMemcpyAsync(ctx->lazy_pending_prof.data(), ctx->l_pflag.p, (size_t)std::max(ctx->P, 1) * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  { int np = 0; for (int32_t f : ctx->lazy_pending_prof) np += f != 0; ctx->stats.n_lazy_pending_profiles = np; }
  if (sw_get("ITSX_LAZY_FORCE_PENDING")) pend += atoll(sw_get("ITSX_LAZY_FORCE_PENDING"));      // test hook: exercises the full re-run
  ctx->lazy_pending = pend; ctx->stats.n_lazy_pending = pend;
  return ITSX_OK;
}

// The top-up round (round 6).  A row is undecided when domE / P-value lies between the bounds on its profile's domZ -- the reported
// targets among the EVALUATED pairs below, the pairs past the MSV filter above.  On amplicon data nearly every pair past the filter is a
// reported target, so the truth is almost always "domZ is far larger, the row is not reported", and proving it does not take the exact
// count (itsx_lazy_complete: every pair of the profile through the whole pipeline, 13.7 M pairs for 4 rows at 10 M reads): it takes
// enough MORE reported targets for the lower bound to pass domE / P-value.  So the profile's unevaluated pairs are ordered by their
// bound (one radix sort), as many of the best as the rows need (+ 2 %) go through the domain pipeline like a third lazy round, and the
// thresholds are applied again; what is still undecided then (a row that IS reported, or one whose count lies in the top third of the
// profile's pairs) is counted in full as before.  The upper bound is tightened on the way: reported among the evaluated + pairs not evaluated.  Exact for the same reason the bounds are: a lower bound that passes domE / P-value decides the row whatever the
// exact count is.  Only for a search of one chunk and one sample whose lists are still in the buffers; everything else goes straight on.
static int lazy_topup(itsx_ctx *ctx, double domE, bool *ran)
{
  (void)domE;
  *ran = false;
  const auto &tu = ctx->topup;
  if (sw_get("ITSX_LAZY_HIST")) fprintf(stderr, "[itsx] top-up: valid %d chunks %d samples %d pairs %lld\n", (int)tu.valid, ctx->n_chunks, ctx->S, (long long)tu.NP);
  if (!tu.valid || ctx->n_chunks != 1 || ctx->S != 1 || tu.NP <= 0) return ITSX_OK;
  if (sw_get("ITSX_LAZY_TOPUP") && atoi(sw_get("ITSX_LAZY_TOPUP")) == 0) return ITSX_OK;      // (read at every call)
  hipStream_t st = ctx->st;
  const int P = ctx->P;
  itsx_stats &S = ctx->stats;
  // the profiles with undecided rows
  std::vector<int32_t> slot((size_t)P, -1), prof_of;
  for (int p = 0; p < P; p++) {
    if (!ctx->lazy_pending_prof[(size_t)p]) continue;
    if (ctx->h_zneed[(size_t)2 * p] == 0 && ctx->h_zneed[(size_t)2 * p + 1] == ~0ull) continue;
    slot[(size_t)p] = (int32_t)prof_of.size(); prof_of.push_back(p);
  }
  const int ns = (int)prof_of.size();
  if (ns == 0) return ITSX_OK;
  StageTimer tm(st);
  const int64_t NP = tu.NP;
  HIPCHK(upload(ctx->l_slot, slot, st)); HIPCHK(upload(ctx->l_pslot, prof_of, st));
  HIPCHK(upload(ctx->w_idx, tu.seg_start, st)); HIPCHK(upload(ctx->w_rcnt, tu.total, st));
  // evaluated pairs per profile so far: the upper bound on its domZ is (reported among them) + (pairs not evaluated)
  auto count_done = [&](std::vector<unsigned long long> &out) -> int {
    HIPCHK(ctx->l_tcnt.alloc((size_t)ns));
    HIPCHK(hipMemsetAsync(ctx->l_tcnt.p, 0, (size_t)ns * sizeof(unsigned long long), st));
    launch_topup_count(ctx->l_done.p, ctx->w_idx.p, ctx->w_rcnt.p, ctx->l_pslot.p, ns, ctx->l_tcnt.p, st);
    out.assign((size_t)ns, 0);
    HIPCHK(hipMemcpyAsync(out.data(), ctx->l_tcnt.p, (size_t)ns * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return ITSX_OK;
  };
  std::vector<unsigned long long> ndone;
  { const int rc = count_done(ndone); if (rc != ITSX_OK) return rc; }
  // the unevaluated pairs of those profiles, compact (a few per cent of the list), then ordered by (profile, bound descending)
  launch_topup_keys(ctx->d_pairs.p, NP, ctx->l_done.p, ctx->l_b10.p, ctx->l_slot.p, ctx->l_flag.p, nullptr, nullptr, nullptr, st);
  launch_exclusive_scan(ctx->l_flag.p, ctx->l_pos.p, NP + 1, ctx->l_scan.p, st);
  int32_t M32 = 0;
  HIPCHK(hipMemcpyAsync(&M32, ctx->l_pos.p + NP, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const int64_t M = M32;
  if (M <= 0) return ITSX_OK;
  HIPCHK(ctx->sh_keys.alloc((size_t)M + 1)); HIPCHK(ctx->sh_keys2.alloc((size_t)M + 1)); HIPCHK(ctx->sh_vals.alloc((size_t)M + 1)); HIPCHK(ctx->sh_uorder.alloc((size_t)M + 1));
  const size_t sort_bytes = order_sort_bytes(M);
  HIPCHK(ctx->sh_sorttmp.alloc(sort_bytes));
  launch_topup_keys(ctx->d_pairs.p, NP, ctx->l_done.p, ctx->l_b10.p, ctx->l_slot.p, ctx->l_flag.p, ctx->l_pos.p, ctx->sh_keys.p, ctx->sh_vals.p, st);
  if (order_sort(ctx->sh_sorttmp.p, sort_bytes, ctx->sh_keys.p, ctx->sh_keys2.p, ctx->sh_vals.p, ctx->sh_uorder.p, M, st) != 0) SET_ERR(ctx, ITSX_E_DEVICE, "top-up round: the radix sort failed");
  HIPCHK(hipStreamSynchronize(st));
  auto lower = [&](unsigned long long key, int64_t &pos) -> int {       // binary search over device memory: ~24 eight-byte copies per probe
    int64_t lo = 0, hi = M;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1; unsigned long long v = 0;
      HIPCHK(hipMemcpy(&v, ctx->sh_keys2.p + mid, 8, hipMemcpyDeviceToHost));
      if (v < key) lo = mid + 1; else hi = mid;
    }
    pos = lo; return ITSX_OK;
  };
  std::vector<uint32_t> cut((size_t)2 * ns, 0);
  bool any = false;
  for (int k = 0; k < ns; k++) {
    const int p = prof_of[(size_t)k];
    int64_t a = 0, b = 0;
    { const int rc = lower((unsigned long long)k << 32, a); if (rc != ITSX_OK) return rc; }
    { const int rc = lower((unsigned long long)(k + 1) << 32, b); if (rc != ITSX_OK) return rc; }
    const int64_t have = b - a;                          // its unevaluated pairs, best bound first in [a, b)
    if (have <= 0) continue;
    const int64_t lo = ctx->domz[(size_t)p], hi = lo + have;       // (the tight upper bound: every unevaluated pair reported)
    int64_t from_top = 0, from_bottom = 0;
    const unsigned long long zl = ctx->h_zneed[(size_t)2 * p], zh = ctx->h_zneed[(size_t)2 * p + 1];
    // rows settled from BELOW: the lower bound must reach zl -- that many more reported targets, taken from the best pairs (+ 2 %: a few are not)
    if (zl > 0 && (int64_t)zl > lo) { const int64_t need = (int64_t)zl - lo; from_top = need + need / 50 + 64; }
    // rows settled from ABOVE: the upper bound must fall to zh -- that many pairs shown unreported, looked for among the weakest (+ 25 %: some are reported)
    // (only with ITSX_LAZY_TOPUP=2: on amplicon data nearly every pair past the filter IS a reported target -- at 10 M reads 387 k of a profile's
    // 394 k weakest pairs were --, so the upper bound hardly moves and such a row goes to the full count anyway)
    const bool both = sw_get("ITSX_LAZY_TOPUP") && atoi(sw_get("ITSX_LAZY_TOPUP")) == 2;
    if (both && zh != ~0ull && hi > (int64_t)zh) { const int64_t need = hi - (int64_t)zh; from_bottom = need + need / 4 + 64; }
    if (sw_get("ITSX_LAZY_HIST")) fprintf(stderr, "[itsx] top-up: profile %d bounds %lld .. %lld (%lld unevaluated), rows need >= %llu / <= %llu: best %lld, weakest %lld\n", p, (long long)lo, (long long)hi,
                                          (long long)have, zl, zh == ~0ull ? 0ull : zh, (long long)from_top, (long long)from_bottom);
    // A profile whose rows need most of its pairs anyway gets ALL of its unevaluated pairs in this round: its bounds meet afterwards (every
    // pair evaluated once, the count exact), and what itsx_lazy_complete would do for it -- the filter again on the profile, every pair of
    // it through the pipeline again, the evaluated ones a second time, in an invocation of its own with its own ensemble batch -- is not
    // needed (10 M reads: top-up 89 + full count 165 ms -> one round of 2xx ms).  ITSX_LAZY_TOPUP_ALL=0: left to the full count as before.
    static const bool topup_all = !(sw_get("ITSX_LAZY_TOPUP_ALL") && atoi(sw_get("ITSX_LAZY_TOPUP_ALL")) == 0);
    const bool above = zh != ~0ull && hi > (int64_t)zh;      // a row that only the upper bound settles: its count lies in the top third of the profile's pairs
    if ((above && !both) || from_top + from_bottom > have - have / 3) {
      if (!topup_all) continue;
      cut[(size_t)2 * k] = 1u; cut[(size_t)2 * k + 1] = 0u;      // (every bound is at least 1: k_lazy_bound)
      any = true;
      continue;
    }
    unsigned long long v = 0;
    if (from_top > 0) {
      HIPCHK(hipMemcpy(&v, ctx->sh_keys2.p + a + from_top - 1, 8, hipMemcpyDeviceToHost));
      cut[(size_t)2 * k] = std::max<uint32_t>(1u, 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
    }
    if (from_bottom > 0) {
      HIPCHK(hipMemcpy(&v, ctx->sh_keys2.p + b - from_bottom, 8, hipMemcpyDeviceToHost));
      cut[(size_t)2 * k + 1] = std::max<uint32_t>(1u, 0xFFFFFFFFu - (uint32_t)(v & 0xFFFFFFFFull));
      if (cut[(size_t)2 * k] > 0 && cut[(size_t)2 * k + 1] >= cut[(size_t)2 * k]) { cut[(size_t)2 * k] = cut[(size_t)2 * k + 1] = 0; continue; }   // (the two ends meet: the full count)
    }
    any = any || from_top > 0 || from_bottom > 0;
  }
  if (!any) return ITSX_OK;
  HIPCHK(upload(ctx->l_cut, cut, st));
  launch_topup_mark(ctx->d_pairs.p, NP, ctx->l_done.p, ctx->l_b10.p, ctx->l_slot.p, ctx->l_cut.p, ctx->l_flag.p, st);
  launch_exclusive_scan(ctx->l_flag.p, ctx->l_pos.p, NP + 1, ctx->l_scan.p, st);
  std::vector<int32_t> bound((size_t)P + 1);
  {
    DBuf<int32_t> &d_b = ctx->w_b;
    HIPCHK(d_b.alloc((size_t)P + 1));
    hipLaunchKernelGGL(k_gather_i32, dim3((P + 1 + 255) / 256), dim3(256), 0, st, ctx->l_pos.p, ctx->w_idx.p, P + 1, d_b.p);
    HIPCHK(hipMemcpyAsync(bound.data(), d_b.p, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  PairList sub;
  sub.seg_start.assign((size_t)P + 1, 0); sub.total.assign((size_t)P, 0);
  for (int p = 0; p < P; p++) { sub.total[(size_t)p] = bound[(size_t)p + 1] - bound[(size_t)p]; sub.seg_start[(size_t)p + 1] = sub.seg_start[(size_t)p] + ((int64_t)sub.total[(size_t)p] + 63) / 64 * 64; }
  sub.NP = sub.seg_start[(size_t)P];
  const int64_t nsel = (int64_t)bound[(size_t)P] - bound[0];
  if (sub.NP == 0) return ITSX_OK;
  HIPCHK(ctx->l_pairs.alloc((size_t)sub.NP)); HIPCHK(ctx->d_pout.alloc((size_t)sub.NP));
  HIPCHK(hipMemsetAsync(ctx->l_pairs.p, 0xFF, (size_t)sub.NP * sizeof(PairRec), st));
  HIPCHK(hipMemsetAsync(ctx->d_pout.p, 0, (size_t)sub.NP * sizeof(PairOut), st));
  HIPCHK(upload(ctx->l_seg, sub.seg_start, st));
  launch_lazy_scatter(ctx->d_pairs.p, NP, ctx->l_flag.p, ctx->l_pos.p, ctx->w_seg_start.p, ctx->l_seg.p, ctx->l_pairs.p, st);
  sub.pairs = ctx->l_pairs.p; sub.pout = ctx->d_pout.p; sub.d_seg_start = ctx->l_seg.p;
  const int32_t *d_sorted = (ctx->share_on ? ctx->sh_order.p : ctx->d_sorted_uniq.p) + tu.u0;
  ctx->trace_u0 = tu.u0;
  { StageTimer tp(st); const int rc = domain_pipeline(ctx, sub, d_sorted, ctx->T, ctx->sF1, ctx->sF3, nullptr); if (rc != ITSX_OK) return rc; S.ms_lazy_topup_stages += tp.stop(); }
  // the bounds after the round: reported among the evaluated pairs below, those + the pairs still not evaluated above
  std::vector<int32_t> dz32((size_t)P * ctx->S, 0);
  HIPCHK(hipMemcpyAsync(dz32.data(), ctx->d_domz32.p, dz32.size() * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(upload(ctx->w_idx, tu.seg_start, st)); HIPCHK(upload(ctx->w_rcnt, tu.total, st));      // (the pipeline used the buffers)
  { const int rc = count_done(ndone); if (rc != ITSX_OK) return rc; }
  for (size_t z = 0; z < dz32.size(); z++) { ctx->domz_loc[z] = dz32[z]; ctx->domz[z] = dz32[z]; }
  for (int k = 0; k < ns; k++) {
    const int p = prof_of[(size_t)k];
    const int64_t left = (int64_t)tu.total[(size_t)p] - (int64_t)ndone[(size_t)k];       // (helper pairs, xj < 0, are marked done by k_lazy_bound)
    const int64_t hi = (int64_t)dz32[(size_t)p] + std::max<int64_t>(0, left);
    if (hi < ctx->domz_ub[(size_t)p]) { ctx->domz_ub[(size_t)p] = hi; ctx->domz_ub_loc[(size_t)p] = hi; }
    if (sw_get("ITSX_LAZY_HIST")) fprintf(stderr, "[itsx] top-up: profile %d now %d .. %lld\n", p, dz32[(size_t)p], (long long)ctx->domz_ub[(size_t)p]);
  }
  S.n_lazy_evaluated += nsel; S.n_lazy_topup += nsel;
  S.n_rows_resident = 0;
  for (int64_t r : ctx->dom_n) S.n_rows_resident += r;
  S.ms_lazy_topup += tm.stop();
  *ran = true;
  return ITSX_OK;
}

int itsx_search_finalize(itsx_ctx *ctx, double domE)
{
  CTXCHK(ctx && ctx->have_search);
  HIPCHK(hipSetDevice(ctx->device));
  StageTimer tm(ctx->st);
  ctx->h_dom.clear();
  auto check_compaction = [&](const std::vector<int64_t> &z) -> int {       // the rows were thinned out under two assumptions: check them against what finalize was given
    int64_t zmax = 0;
    for (int64_t v : z) zmax = std::max(zmax, v);
    if ((double)zmax > ctx->compact_zmax || domE < ctx->compact_dome_min)
      SET_ERR(ctx, ITSX_E_UNSUPPORTED, "compacted domain rows: only the rows that are reported for every domZ <= " + std::to_string(ctx->compact_zmax) +
              " and every domE >= " + std::to_string(ctx->compact_dome_min) + " were kept, but finalize got domZ up to " + std::to_string(zmax) + " and domE " + std::to_string(domE) +
              " (raise ITSX_COMPACT_ZMAX / lower ITSX_COMPACT_DOME_MIN and search again)");
    return ITSX_OK;
  };
  if (ctx->lazy) {
    { const int rc = finalize_lazy(ctx, domE); if (rc != ITSX_OK) return rc; }
    { const int rc = check_compaction(ctx->domz_ub); if (rc != ITSX_OK) return rc; }
    if (ctx->lazy_pending > 0 && !ctx->domz_exchanged && !sw_get("ITSX_LAZY_NO_RERUN")) {
      // Rows whose reporting depends on the exact domZ could change a result.  This context is on its own (nobody exchanged
      // counters): the profiles of those rows are counted exactly (itsx_lazy_complete) and the thresholds applied again; a
      // multi-rank driver does the same across ranks (itsxpress_amd/dist.py: exchange_and_finalize).
      const int64_t pend = ctx->lazy_pending;
      {   // first the cheap way: enough more of the profiles' best pairs for the lower bounds to decide the rows (lazy_topup)
        bool ran = false;
        { const int rc = lazy_topup(ctx, domE, &ran); if (rc != ITSX_OK) return rc; }
        if (ran) {
          { const int rc = finalize_lazy(ctx, domE); if (rc != ITSX_OK) return rc; }
          ctx->stats.n_lazy_pending = pend;
          if (ctx->lazy_pending == 0) {
            ctx->stats.ms_finalize = tm.stop();
            ctx->have_final = true;
            return ITSX_OK;
          }
        }
      }
      if (!sw_get("ITSX_LAZY_NO_COMPLETE")) {
        std::vector<int32_t> flags(ctx->lazy_pending_prof);
        flags.resize((size_t)std::max(ctx->P, 1), 0);
        { const int rc = itsx_lazy_complete(ctx, flags.data()); if (rc != ITSX_OK) return rc; }
        { const int rc = finalize_lazy(ctx, domE); if (rc != ITSX_OK) return rc; }
        ctx->stats.n_lazy_pending = pend;
      }
      if (ctx->lazy_pending == 0) {
        ctx->stats.ms_finalize = tm.stop();
        ctx->have_final = true;
        return ITSX_OK;
      }
      // (cannot happen after a completion -- every row of a counted profile is decided --; kept as the safety net: everything in full)
      const int keep = ctx->rows_mode;
      ctx->rows_mode = ITSX_ROWS_COMPACT;
      const int rc = itsx_search(ctx, ctx->T, ctx->sF1, ctx->F2, ctx->sF3);
      ctx->rows_mode = keep;
      if (rc != ITSX_OK) return rc;
      ctx->stats.n_lazy_reruns = 1; ctx->stats.n_lazy_pending = pend; ctx->stats.n_lazy_pending_profiles = 0; for (int32_t f : ctx->lazy_pending_prof) ctx->stats.n_lazy_pending_profiles += f != 0;
    } else {
      ctx->stats.ms_finalize = tm.stop();
      ctx->have_final = true;
      return ITSX_OK;
    }
  }
  {
    DBuf<int64_t> &d_dz = ctx->domz_on_device ? ctx->d_domz64 : ctx->w_dz;
    if (ctx->domz_on_device) {             // reduced in place by the caller (RCCL): the host copy follows the device
      if (!ctx->domz.empty()) HIPCHK(hipMemcpyAsync(ctx->domz.data(), d_dz.p, ctx->domz.size() * 8, hipMemcpyDeviceToHost, ctx->st));
    } else HIPCHK(upload(d_dz, ctx->domz, ctx->st));
    for (size_t c = 0; c < ctx->dom_n.size(); c++)
      if (ctx->dom_n[c] > 0) launch_finalize(ctx->dom_bufs[c]->p, ctx->dom_n[c], d_dz.p, domE, ctx->dev_usample(), ctx->P, ctx->st);
    HIPCHK(hipStreamSynchronize(ctx->st));
    if (ctx->compact_rows) { const int rc = check_compaction(ctx->domz); if (rc != ITSX_OK) return rc; }
  }
  ctx->stats.ms_finalize = tm.stop();
  ctx->have_final = true;
  return ITSX_OK;
}

static int fetch_domains(const itsx_ctx *cctx)
{
  itsx_ctx *ctx = const_cast<itsx_ctx *>(cctx);
  if (ctx->compact_rows && !ctx->kept_rows) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "the domain rows of this search were compacted (ITSX_COMPACT_ROWS=1): only coordinates are available, not the row table / domtbl.txt (itsx_set_kept_rows(ctx, 1) serves the rows that were kept)");
  if (ctx->compact_rows && !ctx->have_final) SET_ERR(ctx, ITSX_E_ARG, "the kept rows of a compact / lazy search are served after itsx_search_finalize");
  if (ctx->lazy && ctx->lazy_pending > 0) SET_ERR(ctx, ITSX_E_ARG, "the kept rows of a lazy search are served once no row is undecided (itsx_lazy_pending)");
  if (!ctx->h_dom.empty()) return ITSX_OK;
  // domtblout order: profile, then target, then domain.  The device rows are grouped by profile already (not contiguously
  // across chunks), so: counting sort by profile, then every profile's rows are ordered on their own by a pool of threads.
  const int P = std::max(ctx->P, 1);
  std::vector<itsx_domain> all;
  std::vector<int64_t> start((size_t)P + 1, 0);
  for (size_t c = 0; c < ctx->dom_n.size(); c++) {
    if (ctx->dom_n[c] <= 0) continue;
    const size_t o = all.size();
    all.resize(o + (size_t)ctx->dom_n[c]);
    HIPCHK(hipMemcpy(all.data() + o, ctx->dom_bufs[c]->p, (size_t)ctx->dom_n[c] * sizeof(itsx_domain), hipMemcpyDeviceToHost));
  }
  for (const auto &d : all) if (d.dom_idx >= 0 && d.prof >= 0 && d.prof < P) start[(size_t)d.prof + 1]++;
  for (int p = 0; p < P; p++) start[(size_t)p + 1] += start[(size_t)p];
  ctx->h_dom.resize((size_t)start[(size_t)P]);
  {
    std::vector<int64_t> cur(start.begin(), start.end() - 1);
    for (const auto &d : all) if (d.dom_idx >= 0 && d.prof >= 0 && d.prof < P) ctx->h_dom[(size_t)cur[(size_t)d.prof]++] = d;
  }
  std::vector<itsx_domain>().swap(all);
  std::atomic<int> next{0};
  on_threads(std::min(itsx_io::io_threads(), P), [&](int) {
    for (int p = next.fetch_add(1); p < P; p = next.fetch_add(1))
      std::sort(ctx->h_dom.begin() + start[(size_t)p], ctx->h_dom.begin() + start[(size_t)p + 1], [](const itsx_domain &a, const itsx_domain &b) {
        if (a.rep != b.rep) return a.rep < b.rep;
        return a.dom_idx < b.dom_idx;
      });
  });
  if (ctx->compact_rows) {
    // kept rows (itsx_set_kept_rows): a pair the lazy stage evaluated twice -- a top-up or completion round after round 1 / 2 -- left
    // its rows twice, bit for bit the same; the table names every (profile, target, domain) once.  A row that is still undecided
    // (dom_reported == 2: reported under the lower bound on domZ only) is one that cannot win -- it ranks below its group's best sure
    // row, or it would have been settled (k_lazy_pending) -- and is left out like every other row that cannot
    // ... and of the decided rows the table lists the WINNERS: per target and 2-character prefix the reported row with the largest rank
    // key, the one ItsPosition.parse ends up with.  (What else is resident -- a round-1 row that a round-2 row outranked, a row that
    // was best until a later batch -- depends on the order the batches ran in; the winners do not.)
    std::vector<int8_t> cls((size_t)P, 0); std::vector<std::string> seen;
    for (int p = 0; p < ctx->P; p++) {
      const std::string pre = ctx->profs[(size_t)p].name.substr(0, 2);
      size_t k = 0; while (k < seen.size() && seen[k] != pre) k++;
      if (k == seen.size()) seen.push_back(pre);
      cls[(size_t)p] = (int8_t)k;
    }
    const size_t ncls = std::max<size_t>(seen.size(), 1);
    std::vector<unsigned long long> best((size_t)std::max(ctx->U, 1) * ncls, 0ull);
    for (const auto &d : ctx->h_dom)
      if (d.dom_reported == 1) { unsigned long long &b = best[(size_t)d.rep * ncls + (size_t)cls[(size_t)d.prof]]; b = std::max(b, rank_key(d)); }
    size_t w = 0;
    for (size_t r = 0; r < ctx->h_dom.size(); r++) {
      const itsx_domain &d = ctx->h_dom[r];
      if (d.dom_reported != 1 || rank_key(d) != best[(size_t)d.rep * ncls + (size_t)cls[(size_t)d.prof]]) continue;
      if (w > 0 && ctx->h_dom[w - 1].prof == d.prof && ctx->h_dom[w - 1].rep == d.rep && ctx->h_dom[w - 1].dom_idx == d.dom_idx) continue;
      if (w != r) ctx->h_dom[w] = d;
      w++;
    }
    ctx->h_dom.resize(w);
  }
  return ITSX_OK;
}

int64_t itsx_num_domains(const itsx_ctx *ctx)
{
  if (!ctx || !ctx->have_search) return -1;
  if (fetch_domains(ctx) != ITSX_OK) return -1;
  return (int64_t)ctx->h_dom.size();
}
int itsx_get_domains(const itsx_ctx *ctx, itsx_domain *rows)
{
  CTXCHK(ctx && rows && ctx->have_search);
  const int rc = fetch_domains(ctx);
  if (rc != ITSX_OK) return rc;
  if (!ctx->h_dom.empty()) memcpy(rows, ctx->h_dom.data(), ctx->h_dom.size() * sizeof(itsx_domain));
  return ITSX_OK;
}

// copy the current chunk's pair records into the host trace list
static int append_traces(itsx_ctx *ctx)
{
  const int64_t NP = ctx->npairs_padded;
  if (NP == 0) return ITSX_OK;
  std::vector<PairRec> pr((size_t)NP); std::vector<PairOut> po((size_t)NP); std::vector<VitOut> vo;
  HIPCHK(hipMemcpy(pr.data(), ctx->d_pairs.p, (size_t)NP * sizeof(PairRec), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(po.data(), ctx->d_pout.p, (size_t)NP * sizeof(PairOut), hipMemcpyDeviceToHost));
  if (ctx->have_vit) { vo.resize((size_t)NP); HIPCHK(hipMemcpy(vo.data(), ctx->d_vit.p, (size_t)NP * sizeof(VitOut), hipMemcpyDeviceToHost)); }
  // PairRec::useq is a position in the list the search walked: with prefix sharing the processing order (k_share.hip), else by length
  std::vector<int32_t> order;
  if (ctx->share_on) {
    order.resize((size_t)ctx->U_active);
    HIPCHK(hipMemcpy(order.data(), ctx->sh_order.p, order.size() * 4, hipMemcpyDeviceToHost));
  }
  const int32_t *walked = ctx->share_on ? order.data() : ctx->h_sorted_active.data();
  for (int64_t i = 0; i < NP; i++) {
    if (pr[i].prof < 0 || pr[i].xj < 0) continue;          // (padding; a pair that ran only for a chain below it)
    itsx_pairtrace t{};
    t.rep = walked[(size_t)ctx->trace_u0 + pr[i].useq]; t.prof = pr[i].prof; t.msv_xj = pr[i].xj; t.pass_msv = 1;
    t.pass_bias = po[i].pass_bias; t.pass_fwd = po[i].pass_fwd; t.msv_sc = po[i].msv_sc; t.filtersc = po[i].filtersc;
    t.fwdsc = po[i].fwdsc; t.bcksc = po[i].bcksc; t.nullsc = po[i].nullsc; t.nregions = po[i].nregions; t.ndom = po[i].ndom;
    const bool ran = ctx->have_vit && po[i].pass_bias && vo[(size_t)i].ran;
    t.ran_vit = ran; t.vitsc = ran ? vo[(size_t)i].vitsc : 0.0f; t.pass_vit = po[i].pass_bias && (!ran || vo[(size_t)i].pass);
    ctx->h_trace.push_back(t);
  }
  return ITSX_OK;
}
// traces are fetched lazily when the search ran as one chunk; multi-chunk runs keep them only with ITSX_KEEP_TRACE=1
static int fetch_traces(const itsx_ctx *cctx)
{
  itsx_ctx *ctx = const_cast<itsx_ctx *>(cctx);
  if (ctx->h_trace.empty() && !ctx->keep_trace) {
    if (ctx->n_chunks > 1) SET_ERR(ctx, ITSX_E_ARG, "pair traces of a multi-chunk search are kept only when ITSX_KEEP_TRACE=1");
    const int rc = append_traces(ctx);
    if (rc != ITSX_OK) return rc;
  }
  if (!ctx->trace_sorted) {
    std::sort(ctx->h_trace.begin(), ctx->h_trace.end(), [](const itsx_pairtrace &a, const itsx_pairtrace &b) {
      if (a.prof != b.prof) return a.prof < b.prof;
      return a.rep < b.rep;
    });
    ctx->trace_sorted = true;
  }
  return ITSX_OK;
}
int64_t itsx_num_pairtraces(const itsx_ctx *ctx)
{
  if (!ctx || !ctx->have_search) return -1;
  if (fetch_traces(ctx) != ITSX_OK) return -1;
  return (int64_t)ctx->h_trace.size();
}
int itsx_get_pairtraces(const itsx_ctx *ctx, itsx_pairtrace *rows, int64_t row_size)
{
  if (ctx && row_size != (int64_t)sizeof(itsx_pairtrace)) SET_ERR(ctx, ITSX_E_ARG, "itsx_get_pairtraces: the caller's rows have " + std::to_string(row_size) + " bytes, this library's " + std::to_string(sizeof(itsx_pairtrace)));
  CTXCHK(ctx && rows && ctx->have_search);
  const int rc = fetch_traces(ctx);
  if (rc != ITSX_OK) return rc;
  if (!ctx->h_trace.empty()) memcpy(rows, ctx->h_trace.data(), ctx->h_trace.size() * sizeof(itsx_pairtrace));
  return ITSX_OK;
}



// ------------------------------------------------------------------------------ f4: read orientation
int itsx_orient_load_db(itsx_ctx *ctx, const char *fasta_path, int64_t *n_sequences)
{
  CTXCHK(ctx && fasta_path);
  HIPCHK(hipSetDevice(ctx->device));
  init_codes();
  std::string rerr;
  const auto tp = slurp(fasta_path, false, rerr);
  if (!tp) SET_ERR(ctx, ITSX_E_IO, rerr);
  const itsx_io::Text &text = *tp;
  std::vector<uint32_t> bits(1u << 19, 0u);                 // 4^12 bits
  int64_t nseq = 0;
  const bool use_dust = qmask_dust();                        // vsearch --dbmask dust (its default): masked words are not indexed
  std::vector<uint8_t> seq, msk;
  auto add_seq = [&]() {
    if (seq.empty()) return;
    if (use_dust) dust_host(seq.data(), (int64_t)seq.size(), msk);
    uint32_t w = 0; int good = 0;
    for (size_t i = 0; i < seq.size(); i++) {
      const int code = seq[i];
      if (code <= 3 && !(use_dust && msk[i])) { w = (w >> 2) | ((uint32_t)code << 22); good++; } else { w = 0; good = 0; }
      if (good >= 12) bits[w >> 5] |= 1u << (w & 31);
    }
    seq.clear();
  };
  bool header = false;
  for (size_t i = 0; i < text.size(); i++) {
    const char c = text[i];
    if (c == '>') { add_seq(); header = true; nseq++; continue; }
    if (c == '\n') { header = false; continue; }
    if (header || c == '\r') continue;
    const int code = g_code[(unsigned char)c];
    seq.push_back((uint8_t)((code >= 0 && code <= 15) ? code : 15));
  }
  add_seq();
  if (nseq == 0) SET_ERR(ctx, ITSX_E_FORMAT, std::string("no FASTA records in ") + fasta_path);
  HIPCHK(upload(ctx->d_orient_db, bits, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->have_orient_db = true;
  if (n_sequences) *n_sequences = nseq;
  return ITSX_OK;
}

int itsx_orient(itsx_ctx *ctx, int8_t *strand, int32_t *count_fwd, int32_t *count_rev)
{
  CTXCHK(ctx && strand);
  if (!ctx->have_orient_db) SET_ERR(ctx, ITSX_E_ARG, "itsx_orient called before itsx_orient_load_db");
  if (ctx->Lmax - 11 > 12000) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "reads longer than 12011 bases are not supported by the orientation kernel");
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->N;
  if (n == 0) return ITSX_OK;
  DBuf<int8_t> d_s; DBuf<int32_t> d_f, d_r;
  HIPCHK(d_s.alloc((size_t)n)); HIPCHK(d_f.alloc((size_t)n)); HIPCHK(d_r.alloc((size_t)n));
  DBuf<uint32_t> dmask;
  const bool use_dust = qmask_dust();                        // vsearch --qmask dust (its default)
  if (use_dust) { HIPCHK(dmask.alloc((size_t)ctx->h_woff[n] + 1)); launch_dust(ctx->rd, dmask.p, ctx->st); }
  launch_orient(ctx->rd, ctx->d_orient_db.p, use_dust ? dmask.p : nullptr, d_s.p, d_f.p, d_r.p, ctx->st);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(strand, d_s.p, (size_t)n, hipMemcpyDeviceToHost, ctx->st));
  if (count_fwd) HIPCHK(hipMemcpyAsync(count_fwd, d_f.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st));
  if (count_rev) HIPCHK(hipMemcpyAsync(count_rev, d_r.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ f2: paired-end merge (k_merge.hip)
extern "C++" {
namespace {
struct MergeTables {
  std::vector<double> q2p, match, mism; std::vector<uint8_t> qsame, qdiff;
  MergeTables() : q2p(128, 0.0), match(128 * 128, 0.0), mism(128 * 128, 0.0), qsame(128 * 128, 0), qdiff(128 * 128, 0)
  {
    // quality-aware scores and merged qualities of vsearch --fastq_mergepairs (Edgar & Flyvbjerg 2015), host libm
    auto q_to_p = [](int c) { const int x = c - 33; return x < 2 ? 0.75 : pow(10.0, -(double)x / 10.0); };
    auto qual_of = [](double p) { double q = rint(-10.0 * log10(p)); if (q > 41.0) q = 41.0; if (q < 0.0) q = 0.0; return (uint8_t)(33 + (int)q); };
    for (int x = 33; x < 127; x++) {
      const double px = q_to_p(x);
      q2p[x] = px;
      for (int y = 33; y < 127; y++) {
        const double py = q_to_p(y);
        qsame[x * 128 + y] = qual_of(px * py / 3.0 / (1.0 - px - py + 4.0 * px * py / 3.0));
        qdiff[x * 128 + y] = qual_of(px * (1.0 - py / 3.0) / (px + py - 4.0 * px * py / 3.0));
        match[x * 128 + y] = log2((1.0 - px - py + px * py * 4.0 / 3.0) / 0.25);
        mism[x * 128 + y] = log2(((px + py) / 3.0 - px * py * 4.0 / 9.0) / 0.25);
      }
    }
  }
};
const MergeTables &merge_tables() { static MergeTables t; return t; }
}  // namespace
}  // extern "C++"

int itsx_merge_tables(double *q2p, double *match, double *mism, uint8_t *qsame, uint8_t *qdiff)
{
  const MergeTables &t = merge_tables();
  if (q2p) memcpy(q2p, t.q2p.data(), 128 * sizeof(double));
  if (match) memcpy(match, t.match.data(), 128 * 128 * sizeof(double));
  if (mism) memcpy(mism, t.mism.data(), 128 * 128 * sizeof(double));
  if (qsame) memcpy(qsame, t.qsame.data(), 128 * 128);
  if (qdiff) memcpy(qdiff, t.qdiff.data(), 128 * 128);
  return ITSX_OK;
}

// keep_os (may be null): the merged bases stay in device memory (taken over by *keep_os; pair i's at foff[i] + roff[i]) and are not copied
// back -- itsx_merge_pairs_load packs them where they are
static int merge_core(itsx_ctx *ctx, const char *fseq, const char *fqual, const int64_t *foff, const char *rseq, const char *rqual,
                      const int64_t *roff, int64_t n, int maxdiffs, double maxee, int allow_stagger,
                      char *out_seq, char *out_qual, int32_t *out_len, int32_t *reason, double *score, int32_t *shift, DBuf<uint8_t> *keep_os)
{
  CTXCHK(ctx && foff && roff && n >= 0 && out_len && reason && (n == 0 || (fseq && fqual && rseq && rqual && (keep_os || (out_seq && out_qual)))));
  HIPCHK(hipSetDevice(ctx->device));
  if (n == 0) return ITSX_OK;
  const int64_t fb = foff[n], rb = roff[n];
  int64_t max_total = 2;
  for (int64_t i = 0; i < n; i++) {
    const int64_t fl = foff[i + 1] - foff[i], rl = roff[i + 1] - roff[i];
    if (fl < 0 || rl < 0) SET_ERR(ctx, ITSX_E_ARG, "offsets must be non-decreasing");
    max_total = std::max(max_total, fl + rl);
  }
  if (max_total > 12000) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "read pairs longer than 12000 bases in total are not supported by the merge kernel");
  {   // every quality byte is looked at once (a gigabyte per million pairs: a pool of threads, not one)
    std::atomic<int> badf{0}, badr{0};
    const int T = (fb + rb) >= ((int64_t)8 << 20) ? std::max(1, std::min(itsx_io::io_threads(), 16)) : 1;
    on_threads(T, [&](int t) {
      int bf = 0, br = 0;
      for (int64_t i = fb * t / T, e = fb * (t + 1) / T; i < e; i++) bf |= ((unsigned char)fqual[i] < 33) | ((unsigned char)fqual[i] > 126);
      for (int64_t i = rb * t / T, e = rb * (t + 1) / T; i < e; i++) br |= ((unsigned char)rqual[i] < 33) | ((unsigned char)rqual[i] > 126);
      if (bf) badf = 1;
      if (br) badr = 1;
    });
    if (badf) SET_ERR(ctx, ITSX_E_FORMAT, "forward quality outside ASCII 33..126 (--fastq_qmax 93)");
    if (badr) SET_ERR(ctx, ITSX_E_FORMAT, "reverse quality outside ASCII 33..126 (--fastq_qmax 93)");
  }
  const MergeTables &t = merge_tables();
  DBuf<uint8_t> d_fs, d_fq, d_rs, d_rq, d_os, d_oq, d_qs, d_qd; DBuf<int64_t> d_fo, d_ro; DBuf<int32_t> d_len, d_reason, d_shift; DBuf<double> d_q2p, d_m, d_x, d_score;
  HIPCHK(d_fs.alloc((size_t)fb + 1)); HIPCHK(d_fq.alloc((size_t)fb + 1)); HIPCHK(d_rs.alloc((size_t)rb + 1)); HIPCHK(d_rq.alloc((size_t)rb + 1));
  HIPCHK(d_os.alloc((size_t)(fb + rb) + 1)); HIPCHK(d_oq.alloc((size_t)(fb + rb) + 1));
  HIPCHK(d_fo.alloc((size_t)n + 1)); HIPCHK(d_ro.alloc((size_t)n + 1)); HIPCHK(d_len.alloc((size_t)n)); HIPCHK(d_reason.alloc((size_t)n)); HIPCHK(d_shift.alloc((size_t)n)); HIPCHK(d_score.alloc((size_t)n));
  HIPCHK(upload(d_q2p, t.q2p, ctx->st)); HIPCHK(upload(d_m, t.match, ctx->st)); HIPCHK(upload(d_x, t.mism, ctx->st));
  HIPCHK(upload(d_qs, t.qsame, ctx->st)); HIPCHK(upload(d_qd, t.qdiff, ctx->st));
  HIPCHK(hipMemcpyAsync(d_fs.p, fseq, (size_t)fb, hipMemcpyHostToDevice, ctx->st)); HIPCHK(hipMemcpyAsync(d_fq.p, fqual, (size_t)fb, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(d_rs.p, rseq, (size_t)rb, hipMemcpyHostToDevice, ctx->st)); HIPCHK(hipMemcpyAsync(d_rq.p, rqual, (size_t)rb, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipMemcpyAsync(d_fo.p, foff, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->st)); HIPCHK(hipMemcpyAsync(d_ro.p, roff, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, ctx->st));
  MergeArgs a{};
  a.fseq = d_fs.p; a.fqual = d_fq.p; a.rseq = d_rs.p; a.rqual = d_rq.p; a.foff = d_fo.p; a.roff = d_ro.p; a.n = n; a.max_total = (int32_t)max_total;
  a.maxdiffs = maxdiffs; a.allow_stagger = allow_stagger ? 1 : 0; a.maxee = maxee;
  a.q2p = d_q2p.p; a.match = d_m.p; a.mism = d_x.p; a.qsame = d_qs.p; a.qdiff = d_qd.p;
  a.out_seq = d_os.p; a.out_qual = d_oq.p; a.out_len = d_len.p; a.reason = d_reason.p; a.shift = d_shift.p; a.score = d_score.p;
  StageTimer tm(ctx->st);
  launch_merge(a, ctx->st);
  ctx->stats.ms_merge = tm.stop();
  HIPCHK(hipGetLastError());
  if (out_seq) HIPCHK(hipMemcpyAsync(out_seq, d_os.p, (size_t)(fb + rb), hipMemcpyDeviceToHost, ctx->st));
  if (out_qual) HIPCHK(hipMemcpyAsync(out_qual, d_oq.p, (size_t)(fb + rb), hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipMemcpyAsync(out_len, d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st)); HIPCHK(hipMemcpyAsync(reason, d_reason.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st));
  if (score) HIPCHK(hipMemcpyAsync(score, d_score.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->st));
  if (shift) HIPCHK(hipMemcpyAsync(shift, d_shift.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (keep_os) { std::swap(keep_os->p, d_os.p); std::swap(keep_os->n, d_os.n); std::swap(keep_os->cap, d_os.cap); }
  return ITSX_OK;
}
int itsx_merge_buffers(itsx_ctx *ctx, const char *fseq, const char *fqual, const int64_t *foff, const char *rseq, const char *rqual,
                       const int64_t *roff, int64_t n, int maxdiffs, double maxee, int allow_stagger,
                       char *out_seq, char *out_qual, int32_t *out_len, int32_t *reason, double *score, int32_t *shift)
{
  CTXCHK(ctx && (n == 0 || (out_seq && out_qual)));
  return merge_core(ctx, fseq, fqual, foff, rseq, rqual, roff, n, maxdiffs, maxee, allow_stagger, out_seq, out_qual, out_len, reason, score, shift, nullptr);
}

// FASTQ in, FASTQ out: R1/R2 (plain or gzip) -> merged reads, labels = forward read's identifier up to the first blank
int itsx_merge_pairs_files(itsx_ctx *ctx, const char *r1_path, const char *r2_path, const char *out_path, int maxdiffs, double maxee,
                           int allow_stagger, int64_t *n_pairs, int64_t *n_merged)
{
  CTXCHK(ctx && r1_path && r2_path && out_path);
  typedef FastxPart Side;
  auto parse = [](const char *path, Side &sd, std::string &perr) -> int {      // no shared state: the two files are read side by side
    const auto tp = slurp(path, true, perr);
    if (!tp) return ITSX_E_IO;
    if (!tp->empty() && (*tp)[0] != '@') { perr = std::string("malformed FASTQ record 1 in ") + path; return ITSX_E_FORMAT; }
    const int prc = parse_fastx(*tp, true, true, sd, perr);
    if (prc != ITSX_OK) perr += std::string(" in ") + path;
    return prc;
  };
  Side f, r;
  std::string ferr, rerr2;
  int rc2 = ITSX_OK;
  static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  std::thread other([&] { rc2 = parse(r2_path, r, rerr2); });
  int rc = parse(r1_path, f, ferr);
  other.join();
  const auto tm1 = std::chrono::steady_clock::now();
  if (rc != ITSX_OK) { ctx->set_error(ferr); return rc; }
  if (rc2 != ITSX_OK) { ctx->set_error(rerr2); return rc2; }
  if (f.ids.size() != r.ids.size()) SET_ERR(ctx, ITSX_E_FORMAT, "R1 and R2 hold different numbers of records");
  const int64_t n = (int64_t)f.ids.size();
  std::string oseq((size_t)(f.seq.size() + r.seq.size()) + 1, '\0'), oqual = oseq;
  std::vector<int32_t> olen((size_t)n + 1), reason((size_t)n + 1);
  rc = itsx_merge_buffers(ctx, f.seq.data(), f.qual.data(), f.off.data(), r.seq.data(), r.qual.data(), r.off.data(), n, maxdiffs, maxee, allow_stagger,
                          &oseq[0], &oqual[0], olen.data(), reason.data(), nullptr, nullptr);
  if (rc != ITSX_OK) return rc;
  const auto tm2 = std::chrono::steady_clock::now();
  // seq.fq is read back by the loader and by the paired trimmer: written through the block writer, which also leaves
  // its text in the reader's cache
  itsx_io::BlockWriter bw;
  std::string werr, buf;
  if (!bw.open(out_path, itsx_io::PLAIN, werr, true)) SET_ERR(ctx, ITSX_E_IO, werr);
  int64_t merged = 0;
  for (int64_t i = 0; i < n; i++) {
    if (reason[i] != 0) continue;
    const size_t o = (size_t)(f.off[i] + r.off[i]);
    buf += '@'; buf.append(f.ids.ptr((size_t)i), f.ids.len((size_t)i)); buf += '\n';
    buf.append(oseq.data() + o, (size_t)olen[i]); buf += "\n+\n";
    buf.append(oqual.data() + o, (size_t)olen[i]); buf += '\n';
    if (buf.size() >= (1u << 20)) { bw.put(buf); buf.clear(); }
    merged++;
  }
  bw.put(buf);
  if (!bw.close(werr)) SET_ERR(ctx, ITSX_E_IO, werr);
  if (trace) {
    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[itsx] merge files: read+inflate+parse %.0f ms, upload+kernel+download %.0f ms, write %.0f ms\n", ms(tm0, tm1), ms(tm1, tm2), ms(tm2, std::chrono::steady_clock::now()));
  }
  if (n_pairs) *n_pairs = n;
  if (n_merged) *n_merged = merged;
  return ITSX_OK;
}

// R1/R2 -> merged reads as THIS CONTEXT'S READ SET (arrays mode: nothing written, nothing parsed again).  The reference writes the merged
// reads to tempdir/seq.fq and vsearch reads them back (SeqSample.py:266-365, 93-131); here the merge kernel's output is gathered into
// a gap-free text on the device and packed where it is -- the merged bases never visit the host, the merged qualities are not needed at
// all (the paired writer slices the ORIGINAL R1 / R2 records with the merged reads' coordinates).  Labels = R1 identifiers of the merged
// pairs, in input order, as itsx_merge_pairs_files writes them.
// the two sides of a paired sample from files (text1 == nullptr) or from record-aligned pieces of their text already in memory (a
// streaming driver's slices: itsx_merge_pairs_load_text)
static int merge_pairs_load_impl(itsx_ctx *ctx, const char *r1_path, const char *r2_path, const char *text1, int64_t nb1, const char *text2, int64_t nb2,
                                 int maxdiffs, double maxee, int allow_stagger, int64_t *n_pairs, int64_t *n_merged)
{
  typedef FastxPart Side;
  auto parse = [](const char *path, const char *text, int64_t nb, Side &sd, std::string &perr) -> int {
    std::shared_ptr<const itsx_io::Text> tp;
    itsx_io::Text view;
    const itsx_io::Text *t = &view;
    if (text) view.borrow(text, (size_t)nb);
    else { tp = slurp(path, true, perr); if (!tp) return ITSX_E_IO; t = tp.get(); }
    if (!t->empty() && (*t)[0] != '@') { perr = std::string("malformed FASTQ record 1 in ") + path; return ITSX_E_FORMAT; }
    const int prc = parse_fastx(*t, true, true, sd, perr);
    if (prc != ITSX_OK) perr += std::string(" in ") + path;
    return prc;
  };
  Side f, r;
  std::string ferr, rerr2;
  int rc2 = ITSX_OK;
  static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
  const auto tm0 = std::chrono::steady_clock::now();
  std::thread other([&] { rc2 = parse(r2_path, text2, nb2, r, rerr2); });
  int rc = parse(r1_path, text1, nb1, f, ferr);
  other.join();
  const auto tm1 = std::chrono::steady_clock::now();
  if (rc != ITSX_OK) { ctx->set_error(ferr); return rc; }
  if (rc2 != ITSX_OK) { ctx->set_error(rerr2); return rc2; }
  if (f.ids.size() != r.ids.size()) SET_ERR(ctx, ITSX_E_FORMAT, "R1 and R2 hold different numbers of records");
  const int64_t n = (int64_t)f.ids.size();
  std::vector<int32_t> olen((size_t)n + 1), reason((size_t)n + 1);
  DBuf<uint8_t> d_os;
  rc = merge_core(ctx, f.seq.data(), f.qual.data(), f.off.data(), r.seq.data(), r.qual.data(), r.off.data(), n, maxdiffs, maxee, allow_stagger,
                  nullptr, nullptr, olen.data(), reason.data(), nullptr, nullptr, &d_os);
  if (rc != ITSX_OK) return rc;
  const auto tm2 = std::chrono::steady_clock::now();
  hipStream_t st = ctx->st;
  std::vector<int64_t> srcoff, dstoff(1, 0);
  ctx->h_names.clear();
  ctx->h_merge_index.assign((size_t)n, -1);              // per pair: its merged read (itsx_merge_pair_index)
  for (int64_t i = 0; i < n; i++) {
    if (reason[i] != 0) continue;
    ctx->h_merge_index[(size_t)i] = (int32_t)srcoff.size();
    srcoff.push_back(f.off[i] + r.off[i]);
    dstoff.push_back(dstoff.back() + olen[i]);
    ctx->h_names.emplace_back(f.ids.ptr((size_t)i), f.ids.ptr((size_t)i) + f.ids.len((size_t)i));
  }
  const int64_t m = (int64_t)srcoff.size();
  if (m >= (1ll << 31) - 64) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "more than 2^31 reads in one context");
  DBuf<int64_t> d_src, d_dst; DBuf<uint8_t> d_cmp;
  HIPCHK(upload(d_src, srcoff, st)); HIPCHK(upload(d_dst, dstoff, st)); HIPCHK(d_cmp.alloc((size_t)dstoff.back() + 64));
  if (m > 0) hipLaunchKernelGGL(k_gather_reads, dim3((unsigned)std::min<int64_t>((m + 3) / 4, 65535)), dim3(256), 0, st, d_os.p, d_src.p, d_dst.p, m, d_cmp.p);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  ctx->N = m;
  ctx->h_off.swap(dstoff);
  itsx_io::Text().swap(ctx->h_bases);
  static const uint8_t none = 0;
  rc = pack_and_upload(ctx, nullptr, m > 0 ? d_cmp.p : &none);
  if (trace) {
    const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[itsx] merge + load: read+inflate+parse %.0f ms, upload+kernel %.0f ms, gather+pack %.0f ms\n", ms(tm0, tm1), ms(tm1, tm2), ms(tm2, std::chrono::steady_clock::now()));
  }
  if (n_pairs) *n_pairs = n;
  if (n_merged) *n_merged = m;
  return rc;
}
int itsx_merge_pairs_load(itsx_ctx *ctx, const char *r1_path, const char *r2_path, int maxdiffs, double maxee, int allow_stagger, int64_t *n_pairs, int64_t *n_merged)
{
  CTXCHK(ctx && r1_path && r2_path);
  return merge_pairs_load_impl(ctx, r1_path, r2_path, nullptr, 0, nullptr, 0, maxdiffs, maxee, allow_stagger, n_pairs, n_merged);
}
// Round 6 (a streamed paired sample, itsxpress_amd/stream.py): the same, from record-aligned pieces of R1's and R2's text that hold the
// same number of records (itsx_stream_next / itsx_stream_next_records)
int itsx_merge_pairs_load_text(itsx_ctx *ctx, const char *text1, int64_t nbytes1, const char *text2, int64_t nbytes2, int maxdiffs, double maxee, int allow_stagger,
                               int64_t *n_pairs, int64_t *n_merged)
{
  CTXCHK(ctx && text1 && text2 && nbytes1 >= 0 && nbytes2 >= 0);
  return merge_pairs_load_impl(ctx, "(R1 slice)", "(R2 slice)", text1, nbytes1, text2, nbytes2, maxdiffs, maxee, allow_stagger, n_pairs, n_merged);
}
// per pair of the last itsx_merge_pairs_load / _load_text: the index of its merged read in the context's read set, -1 = not merged
int itsx_merge_pair_index(const itsx_ctx *ctx, int32_t *index, int64_t n_pairs)
{
  CTXCHK(ctx && index);
  if ((size_t)n_pairs != ctx->h_merge_index.size()) SET_ERR(ctx, ITSX_E_ARG, "itsx_merge_pair_index: the last merge held " + std::to_string(ctx->h_merge_index.size()) + " pairs");
  memcpy(index, ctx->h_merge_index.data(), (size_t)n_pairs * sizeof(int32_t));
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ coordinates
static int coords_common(itsx_ctx *ctx, const char *lp, const char *rp, bool per_read, int32_t *start, int32_t *stop, int32_t *tlen, int32_t *ind)
{
  CTXCHK(ctx && lp && rp);
  const bool to_host = start && stop && tlen && ind;        // all four, or none (the results stay on the device)
  if (!to_host && (start || stop || tlen || ind)) return ITSX_E_ARG;
  if (!ctx->have_final) SET_ERR(ctx, ITSX_E_ARG, "coordinates requested before itsx_search_finalize");
  if (ctx->lazy && ctx->lazy_pending > 0 && !ctx->partial_coords)
    SET_ERR(ctx, ITSX_E_ARG, std::to_string(ctx->lazy_pending) + " domain row(s) of the lazy search depend on the exact domZ and could change a result: "
            "search again with itsx_set_rows_mode(ctx, 1) (itsx_lazy_pending reports this after itsx_search_finalize)");
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->st;
  const int32_t U = ctx->U; const int64_t n = ctx->N;
  std::vector<int8_t> side((size_t)std::max(ctx->P, 1), 0);
  const size_t ll = strlen(lp), rl = strlen(rp);
  if (ctx->compact_rows && (ll > 2 || rl > 2)) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "ITSX_COMPACT_ROWS keeps the best rows per 2-character profile prefix: a longer prefix cannot be served");
  for (int p = 0; p < ctx->P; p++) {
    const std::string &nm = ctx->profs[p].name;
    if (nm.compare(0, ll, lp) == 0) side[p] = 1;
    else if (nm.compare(0, rl, rp) == 0) side[p] = 2;
  }
  DBuf<int8_t> &d_side = ctx->w_side; DBuf<unsigned long long> &bl = ctx->w_bl, &br = ctx->w_br; DBuf<int32_t> &uind = ctx->w_uind, &us = ctx->w_us, &ue = ctx->w_ue, &ut = ctx->w_ut;
  HIPCHK(upload(d_side, side, st));
  HIPCHK(bl.alloc((size_t)U + 1)); HIPCHK(br.alloc((size_t)U + 1)); HIPCHK(uind.alloc((size_t)U + 1));
  HIPCHK(us.alloc((size_t)U + 1)); HIPCHK(ue.alloc((size_t)U + 1)); HIPCHK(ut.alloc((size_t)U + 1));
  HIPCHK(hipMemsetAsync(bl.p, 0, ((size_t)U + 1) * 8, st)); HIPCHK(hipMemsetAsync(br.p, 0, ((size_t)U + 1) * 8, st));
  HIPCHK(hipMemsetAsync(uind.p, 0, ((size_t)U + 1) * 4, st));
  HIPCHK(ctx->w_cl.alloc((size_t)U + 1)); HIPCHK(ctx->w_cr.alloc((size_t)U + 1));
  for (size_t c = 0; c < ctx->dom_n.size(); c++)
    if (ctx->dom_n[c] > 0) launch_positions(ctx->dom_bufs[c]->p, ctx->dom_n[c], d_side.p, bl.p, br.p, uind.p, st);
  for (size_t c = 0; c < ctx->dom_n.size(); c++)          // the winners' coordinates (exactly one row carries each winning key)
    if (ctx->dom_n[c] > 0) launch_position_coords(ctx->dom_bufs[c]->p, ctx->dom_n[c], d_side.p, bl.p, br.p, ctx->w_cl.p, ctx->w_cr.p, st);
  if (U > 0) hipLaunchKernelGGL(k_rep_coords, dim3((U + 255) / 256), dim3(256), 0, st, U, bl.p, br.p, ctx->w_cl.p, ctx->w_cr.p, ctx->d_seed_read.p, ctx->rd.len, us.p, ue.p, ut.p);
  {   // parity-risk counters (itsx_stats): winners out of clustered regions, pairs at the region cap
    DBuf<int32_t> &uflag = ctx->w_uflag; DBuf<int64_t> &d_c = ctx->w_counters;
    HIPCHK(uflag.alloc((size_t)U + 1)); HIPCHK(d_c.alloc(8));
    HIPCHK(hipMemsetAsync(uflag.p, 0, ((size_t)U + 1) * 4, st)); HIPCHK(hipMemsetAsync(d_c.p, 0, 8 * sizeof(int64_t), st));
    for (size_t c = 0; c < ctx->dom_n.size(); c++)
      if (ctx->dom_n[c] > 0) launch_position_flags(ctx->dom_bufs[c]->p, ctx->dom_n[c], d_side.p, bl.p, br.p, uflag.p, st);
    launch_count_flags(uflag.p, U, ctx->d_uniq_of.p, n, (unsigned long long *)d_c.p, st);
    int64_t hc[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(hc, d_c.p, sizeof(hc), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ctx->stats.n_uniq_multi_winner = hc[0]; ctx->stats.n_reads_multi_winner = hc[1]; ctx->stats.n_uniq_region_cap = hc[2]; ctx->stats.n_reads_region_cap = hc[3];
  }
  if (!per_read) {
    if (U > 0 && to_host) {
      HIPCHK(hipMemcpyAsync(start, us.p, (size_t)U * 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(stop, ue.p, (size_t)U * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(tlen, ut.p, (size_t)U * 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(ind, uind.p, (size_t)U * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return ITSX_OK;
  }
  DBuf<int32_t> &rs = ctx->w_rs, &re = ctx->w_re, &rt = ctx->w_rt, &ri = ctx->w_ri;
  HIPCHK(rs.alloc((size_t)n + 1)); HIPCHK(re.alloc((size_t)n + 1)); HIPCHK(rt.alloc((size_t)n + 1)); HIPCHK(ri.alloc((size_t)n + 1));
  if (n > 0) {
    hipLaunchKernelGGL(k_read_coords, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, ctx->d_uniq_of.p, us.p, ue.p, ut.p, uind.p, rs.p, re.p, rt.p, ri.p);
    if (to_host) {
    HIPCHK(hipMemcpyAsync(start, rs.p, (size_t)n * 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(stop, re.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(tlen, rt.p, (size_t)n * 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipMemcpyAsync(ind, ri.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
  }
  HIPCHK(hipStreamSynchronize(st));
  return ITSX_OK;
}
int itsx_trim_coords(itsx_ctx *ctx, const char *lp, const char *rp, int32_t *start, int32_t *stop, int32_t *tlen, int32_t *ind)
{ return coords_common(ctx, lp, rp, true, start, stop, tlen, ind); }
// the same results left on the device as [n][4] int32 rows (start, stop, tlen, in_ddict): per read / per representative
static int coords_device(itsx_ctx *ctx, const char *lp, const char *rp, bool per_read, int32_t **d_rows, int64_t *n_rows)
{
  CTXCHK(ctx && d_rows);
  const int rc = coords_common(ctx, lp, rp, per_read, nullptr, nullptr, nullptr, nullptr);
  if (rc != ITSX_OK) return rc;
  const int64_t n = per_read ? ctx->N : ctx->U;
  HIPCHK(ctx->w_coords4.alloc((size_t)n * 4 + 4));
  if (n > 0) {
    if (per_read) hipLaunchKernelGGL(k_pack_coords4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->st, n, ctx->w_rs.p, ctx->w_re.p, ctx->w_rt.p, ctx->w_ri.p, ctx->w_coords4.p);
    else hipLaunchKernelGGL(k_pack_coords4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->st, n, ctx->w_us.p, ctx->w_ue.p, ctx->w_ut.p, ctx->w_uind.p, ctx->w_coords4.p);
  }
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipGetLastError());
  *d_rows = ctx->w_coords4.p;
  if (n_rows) *n_rows = n;
  return ITSX_OK;
}
int itsx_trim_coords_device(itsx_ctx *ctx, const char *lp, const char *rp, int32_t **d_rows, int64_t *n_rows) { return coords_device(ctx, lp, rp, true, d_rows, n_rows); }
int itsx_rep_coords_device(itsx_ctx *ctx, const char *lp, const char *rp, int32_t **d_rows, int64_t *n_rows) { return coords_device(ctx, lp, rp, false, d_rows, n_rows); }
int itsx_derep_device(itsx_ctx *ctx, const int32_t **d_rep_of, const int32_t **d_uniq_of, const int8_t **d_strand, const int32_t **d_seed_read)
{
  CTXCHK(ctx && ctx->have_derep);
  if (d_rep_of) *d_rep_of = ctx->d_rep_of.p;
  if (d_uniq_of) *d_uniq_of = ctx->d_uniq_of.p;
  if (d_strand) *d_strand = ctx->d_strand.p;
  if (d_seed_read) *d_seed_read = ctx->d_seed_read.p;
  return ITSX_OK;
}
int itsx_unique_keys128_device(itsx_ctx *ctx, uint64_t seed_a, uint64_t seed_b, int64_t gidx_base, int64_t **d_tuples, int64_t *n_unique)
{
  CTXCHK(ctx && ctx->have_derep && d_tuples);
  if (ctx->S > 1) SET_ERR(ctx, ITSX_E_UNSUPPORTED, "cross-rank dereplication of a sample batch is not supported: shard whole samples across ranks");
  HIPCHK(hipSetDevice(ctx->device));
  LoadPriority load_priority(ctx);
  const int64_t n = ctx->N; const int32_t U = ctx->U;
  HIPCHK(ctx->w_keys128.alloc((size_t)U * 4 + 4));
  if (n > 0 && U > 0) {
    HIPCHK(ctx->w_hf.alloc((size_t)n + 1)); HIPCHK(ctx->w_hr.alloc((size_t)n + 1)); HIPCHK(ctx->w_hf1.alloc((size_t)n + 1)); HIPCHK(ctx->w_hr1.alloc((size_t)n + 1));
    // (a plus-strand-only dereplication keeps a sequence and its reverse complement apart across shards too: the key is the forward
    // strand's then -- k_hash_reads returns it for both strands -- and every flag says "forward")
    launch_hash_reads(ctx->rd, seed_a, ctx->derep_strand_both, ctx->w_hf.p, ctx->w_hr.p, ctx->st);
    launch_hash_reads(ctx->rd, seed_b, ctx->derep_strand_both, ctx->w_hf1.p, ctx->w_hr1.p, ctx->st);
    hipLaunchKernelGGL(k_unique_keys128, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, ctx->st, U, ctx->d_seed_read.p, ctx->rd.len, ctx->w_hf.p, ctx->w_hr.p,
                       ctx->w_hf1.p, ctx->w_hr1.p, gidx_base, ctx->w_keys128.p);
  }
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipGetLastError());
  *d_tuples = ctx->w_keys128.p;
  if (n_unique) *n_unique = U;
  return ITSX_OK;
}
int itsx_rep_coords(itsx_ctx *ctx, const char *lp, const char *rp, int32_t *start, int32_t *stop, int32_t *tlen, int32_t *ind)
{ return coords_common(ctx, lp, rp, false, start, stop, tlen, ind); }
// the same tuples in host memory (a driver without a collective library: itsxpress_amd/multi.py moves them through pipes)
int itsx_unique_keys128(itsx_ctx *ctx, uint64_t seed_a, uint64_t seed_b, int64_t gidx_base, int64_t *tuples)
{
  CTXCHK(ctx && (tuples || ctx->U == 0));
  int64_t *d = nullptr; int64_t nu = 0;
  const int rc = itsx_unique_keys128_device(ctx, seed_a, seed_b, gidx_base, &d, &nu);
  if (rc != ITSX_OK) return rc;
  if (nu > 0) HIPCHK(hipMemcpy(tuples, d, (size_t)nu * 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return ITSX_OK;
}
// E-value parameters and length of profile i: {MSV mu, lambda, Viterbi mu, lambda, Forward tau, lambda} (the array writers' columns)
int itsx_profile_params(const itsx_ctx *ctx, int i, int32_t *M, float *evparam6)
{
  CTXCHK(ctx && i >= 0 && i < ctx->P);
  if (M) *M = ctx->profs[(size_t)i].M;
  if (evparam6) for (int k = 0; k < 6; k++) evparam6[k] = ctx->profs[(size_t)i].evparam[k];
  return ITSX_OK;
}
// sequences of the unique representatives in input order of the seeds (what rep.fa holds), concatenated; offsets[n_unique + 1];
// bases == NULL fills the offsets only
int itsx_get_unique_seqs(itsx_ctx *ctx, char *bases, int64_t cap, int64_t *offsets)
{
  CTXCHK(ctx && offsets && ctx->have_derep);
  if (!ctx->bases_view) {                                   // reads handed over in device memory: the text comes back once
    HIPCHK(hipSetDevice(ctx->device));
    ctx->h_bases.resize((size_t)ctx->h_off[(size_t)ctx->N]);
    if (!ctx->h_bases.empty()) HIPCHK(hipMemcpy(&ctx->h_bases[0], ctx->dev_bases, ctx->h_bases.size(), hipMemcpyDeviceToHost));
    ctx->bases_view = ctx->h_bases.data();
  }
  int64_t o = 0;
  for (int32_t u = 0; u < ctx->U; u++) {
    const int64_t s = ctx->h_seed_read[(size_t)u], L = ctx->h_len[(size_t)s];
    offsets[u] = o;
    if (bases) { if (o + L > cap) SET_ERR(ctx, ITSX_E_ARG, "sequence buffer too small"); memcpy(bases + o, ctx->bases_view + ctx->h_off[(size_t)s], (size_t)L); }
    o += L;
  }
  offsets[ctx->U] = o;
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ writers
// blocks of a text file formatted by a pool of threads while this thread writes the finished ones in order (a few blocks ahead at most);
// false on a short write
static bool write_blocks(FILE *f, size_t nb, const std::function<void(size_t, std::string &)> &format_block)
{
  const int T = (int)std::min<size_t>((size_t)itsx_io::io_threads(), std::max<size_t>(nb, 1));
  bool io_ok = true;
  if (T <= 1 || nb <= 1) {
    std::string out;
    for (size_t b = 0; b < nb; b++) { out.clear(); format_block(b, out); if (!out.empty() && fwrite(out.data(), 1, out.size(), f) != out.size()) io_ok = false; }
  } else {
    std::vector<std::string> blocks(nb);
    std::vector<char> ready(nb, 0);
    std::mutex mu; std::condition_variable cv;
    size_t next = 0, written = 0;
    const size_t ahead = (size_t)T * 4;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&] {
        for (;;) {
          size_t b;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return next >= nb || next < written + ahead; });
            if (next >= nb) return;
            b = next++;
          }
          std::string out;
          out.reserve(32768 * 200);
          format_block(b, out);
          { std::lock_guard<std::mutex> lk(mu); blocks[b].swap(out); ready[b] = 1; }
          cv.notify_all();
        }
      });
    for (size_t b = 0; b < nb; b++) {
      std::string out;
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[b] != 0; }); out.swap(blocks[b]); }
      if (!out.empty() && fwrite(out.data(), 1, out.size(), f) != out.size()) io_ok = false;
      { std::lock_guard<std::mutex> lk(mu); written = b + 1; }
      cv.notify_all();
    }
    for (auto &x : th) x.join();
  }
  return io_ok;
}

static std::string read_name(const itsx_ctx *ctx, int64_t r)
{
  if (!ctx->h_names.empty()) return ctx->h_names[(size_t)r];
  char b[32]; snprintf(b, sizeof(b), "r%09lld", (long long)r); return b;
}
// clusters in vsearch's output order: abundance descending, ties by label
static const std::vector<int32_t> &cluster_order(const itsx_ctx *cctx)
{
  itsx_ctx *ctx = const_cast<itsx_ctx *>(cctx);
  if (ctx->order_cache_ok) return ctx->order_cache;
  std::vector<int32_t> &ord = ctx->order_cache;
  ord.clear();
  ord.reserve((size_t)ctx->U);
  const int32_t sel = ctx->S > 1 ? ctx->sel_sample : -1;      // writers restricted to one sample of a batch
  if (ctx->clustered) {                  // --cluster_size: clusters are numbered as their centroids were created
    for (int32_t r : ctx->h_order) if (ctx->h_rep_of[r] == r) ord.push_back(ctx->h_uniq_of[r]);
    ctx->order_cache_ok = true;
    return ord;
  }
  for (int32_t u = 0; u < ctx->U; u++) if (sel < 0 || ctx->usample(u) == sel) ord.push_back(u);
  std::vector<std::string> lab((size_t)ctx->U);
  const int T = ctx->U >= (1 << 17) ? std::max(1, std::min(16, itsx_io::io_threads())) : 1;
  on_threads(T, [&](int t) { for (int64_t u = (int64_t)ctx->U * t / T, hi = (int64_t)ctx->U * (t + 1) / T; u < hi; u++) lab[(size_t)u] = read_name(ctx, ctx->h_seed_read[(size_t)u]); });
  auto before = [&](int32_t a, int32_t b) {
    if (ctx->h_abund[a] != ctx->h_abund[b]) return ctx->h_abund[a] > ctx->h_abund[b];
    return strcmp(lab[a].c_str(), lab[b].c_str()) < 0;
  };
  // a stable sort in pieces (6 M labels of a 10 M-read sample: seconds on one thread): every thread orders its share, neighbouring
  // shares are merged pairwise -- std::inplace_merge keeps equal elements in order, so the result is std::stable_sort's
  const size_t n = ord.size();
  std::vector<size_t> cut((size_t)T + 1);
  for (int t = 0; t <= T; t++) cut[(size_t)t] = n * (size_t)t / (size_t)T;
  on_threads(T, [&](int t) { std::stable_sort(ord.begin() + (ptrdiff_t)cut[(size_t)t], ord.begin() + (ptrdiff_t)cut[(size_t)t + 1], before); });
  for (int w = 1; w < T; w *= 2) {
    std::vector<int> lefts;
    for (int t = 0; t + w < T; t += 2 * w) lefts.push_back(t);
    on_threads((int)lefts.size(), [&](int k) {
      const int t = lefts[(size_t)k];
      std::inplace_merge(ord.begin() + (ptrdiff_t)cut[(size_t)t], ord.begin() + (ptrdiff_t)cut[(size_t)(t + w)], ord.begin() + (ptrdiff_t)cut[(size_t)std::min(T, t + 2 * w)], before);
    });
  }
  ctx->order_cache_ok = true;
  return ord;
}

int itsx_write_uc(const itsx_ctx *ctx, const char *path)
{
  CTXCHK(ctx && path && ctx->have_derep);
  FILE *f = fopen(path, "w");
  if (!f) SET_ERR(ctx, ITSX_E_IO, std::string("cannot write ") + path);
  const std::vector<int32_t> &ord = cluster_order(ctx);
  if (ctx->clustered) {
    // vsearch --cluster_size writes the S and H rows as the queries are processed, then one C row per cluster
    std::vector<int32_t> cno((size_t)ctx->U);
    for (size_t c = 0; c < ord.size(); c++) cno[ord[c]] = (int32_t)c;
    for (int32_t r : ctx->h_order) {
      const int32_t u = ctx->h_uniq_of[r];
      if (ctx->h_rep_of[r] == r) fprintf(f, "S\t%d\t%d\t*\t*\t*\t*\t*\t%s\t*\n", cno[u], ctx->h_len[r], read_name(ctx, r).c_str());
      else fprintf(f, "H\t%d\t%d\t%.1f\t%c\t0\t0\t*\t%s\t%s\n", cno[u], ctx->h_len[r], ctx->h_pct[r], ctx->h_strand[r] < 0 ? '-' : '+',
                   read_name(ctx, r).c_str(), read_name(ctx, ctx->h_rep_of[r]).c_str());
    }
    for (size_t c = 0; c < ord.size(); c++)
      fprintf(f, "C\t%zu\t%d\t*\t*\t*\t*\t*\t%s\t*\n", c, ctx->h_abund[ord[c]], read_name(ctx, ctx->h_seed_read[ord[c]]).c_str());
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) SET_ERR(ctx, ITSX_E_IO, std::string("short write to ") + path);
    return ITSX_OK;
  }
  // the members of every cluster in input order (a counting sort by representative), then blocks of clusters formatted by the I/O pool:
  // the S row and the H rows of a cluster, cluster after cluster, then the C rows (10 M reads: 16 M lines)
  std::vector<int64_t> mstart((size_t)ctx->U + 1, 0);
  for (int64_t r = 0; r < ctx->N; r++) { const int32_t u = ctx->h_uniq_of[r]; if (u >= 0 && ctx->h_rep_of[r] != r) mstart[(size_t)u + 1]++; }
  for (int32_t u = 0; u < ctx->U; u++) mstart[(size_t)u + 1] += mstart[(size_t)u];
  std::vector<int64_t> member((size_t)mstart[(size_t)ctx->U]);
  {
    std::vector<int64_t> cur(mstart.begin(), mstart.end() - 1);
    for (int64_t r = 0; r < ctx->N; r++) { const int32_t u = ctx->h_uniq_of[r]; if (u >= 0 && ctx->h_rep_of[r] != r) member[(size_t)cur[(size_t)u]++] = r; }
  }
  const size_t BLK = 16384, nbc = (ord.size() + BLK - 1) / BLK;
  auto append = [](std::string &out, const char *fmt, auto... args) {
    char line[512];
    const int len = snprintf(line, sizeof(line), fmt, args...);
    if (len < 0) return;
    if ((size_t)len < sizeof(line)) { out.append(line, (size_t)len); return; }
    std::string big((size_t)len + 1, '\0');                  // (labels longer than the line buffer)
    snprintf(&big[0], big.size(), fmt, args...);
    out.append(big.data(), (size_t)len);
  };
  auto format_block = [&](size_t b, std::string &out) {
    if (b < nbc) {
      for (size_t c = b * BLK; c < std::min(ord.size(), (b + 1) * BLK); c++) {
        const int32_t u = ord[c]; const int64_t s = ctx->h_seed_read[u];
        const std::string sl = read_name(ctx, s);
        append(out, "S\t%zu\t%d\t*\t*\t*\t*\t*\t%s\t*\n", c, ctx->h_len[s], sl.c_str());
        for (int64_t k = mstart[(size_t)u]; k < mstart[(size_t)u + 1]; k++) {
          const int64_t r = member[(size_t)k];
          append(out, "H\t%zu\t%d\t100.0\t%c\t0\t0\t*\t%s\t%s\n", c, ctx->h_len[r], ctx->h_strand[r] < 0 ? '-' : '+', read_name(ctx, r).c_str(), sl.c_str());
        }
      }
    } else {
      for (size_t c = (b - nbc) * BLK; c < std::min(ord.size(), (b - nbc + 1) * BLK); c++) {
        const int32_t u = ord[c];
        append(out, "C\t%zu\t%d\t*\t*\t*\t*\t*\t%s\t*\n", c, ctx->h_abund[u], read_name(ctx, ctx->h_seed_read[u]).c_str());
      }
    }
  };
  // a truncated uc.txt would silently shorten Dedup.parse's matchdict (vsearch exits non-zero on a full disk)
  const bool ok = write_blocks(f, 2 * nbc, format_block);
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad || !ok) SET_ERR(ctx, ITSX_E_IO, std::string("short write to ") + path);
  return ITSX_OK;
}

int itsx_write_rep_fasta(const itsx_ctx *cctx, const char *path)
{
  CTXCHK(cctx && path && cctx->have_derep);
  itsx_ctx *ctx = const_cast<itsx_ctx *>(cctx);
  if (!ctx->bases_view) {                                   // reads handed over in device memory: the text comes back once
    ctx->h_bases.resize((size_t)ctx->h_off[(size_t)ctx->N]);
    if (!ctx->h_bases.empty()) HIPCHK(hipMemcpy(&ctx->h_bases[0], ctx->dev_bases, ctx->h_bases.size(), hipMemcpyDeviceToHost));
    ctx->bases_view = ctx->h_bases.data();
  }
  FILE *f = fopen(path, "w");
  if (!f) SET_ERR(ctx, ITSX_E_IO, std::string("cannot write ") + path);
  const std::vector<int32_t> &ord = cluster_order(ctx);
  const size_t BLK = 8192, nb = (ord.size() + BLK - 1) / BLK;
  auto format_block = [&](size_t bk, std::string &out) {
    for (size_t c = bk * BLK; c < std::min(ord.size(), (bk + 1) * BLK); c++) {
      const int64_t s = ctx->h_seed_read[ord[c]];
      out.push_back('>'); out.append(read_name(ctx, s)); out.push_back('\n');
      const char *b = ctx->bases_view + ctx->h_off[s]; const int64_t L = ctx->h_len[s];
      for (int64_t i = 0; i < L; i += 80) { out.append(b + i, (size_t)std::min<int64_t>(80, L - i)); out.push_back('\n'); }
    }
  };
  const bool ok = write_blocks(f, nb, format_block);
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad || !ok) SET_ERR(ctx, ITSX_E_IO, std::string("short write to ") + path);
  return ITSX_OK;
}

int itsx_write_domtbl(const itsx_ctx *ctx, const char *path)
{
  CTXCHK(ctx && path && ctx->have_final);
  const int rc = fetch_domains(ctx);
  if (rc != ITSX_OK) return rc;
  FILE *f = fopen(path, "w");
  if (!f) SET_ERR(ctx, ITSX_E_IO, std::string("cannot write ") + path);
  fprintf(f, "#                                                                            --- full sequence --- -------------- this domain -------------   hmm coord   ali coord   env coord\n");
  fprintf(f, "# target name        accession   tlen query name           accession   qlen   E-value  score  bias   #  of  c-Evalue  i-Evalue  score  bias  from    to  from    to  from    to  acc description of target\n");
  // rows: profile order; within a profile, targets; within a target, reported domains renumbered
  const std::vector<itsx_domain> &D = ctx->h_dom;
  std::vector<int64_t> Zs((size_t)ctx->S, 0);          // hmmsearch's Z: targets searched, per sample
  // (a context narrowed by itsx_set_active_uniques searched only its active targets: the E-value columns of a sharded run are
  // this shard's own, the reported / not-reported decisions are global through the all-reduced domZ)
  if (ctx->S == 1) Zs[0] = ctx->U_active; else for (int32_t u = 0; u < ctx->U; u++) Zs[(size_t)ctx->usample(u)]++;
  // one (profile, target) group is formatted on its own, so the table is cut into blocks of whole groups that a pool of
  // threads formats while this thread writes the finished blocks in order (a few blocks ahead at most)
  std::vector<size_t> cut(1, 0);
  {
    const size_t target = 32768;
    size_t i = 0, last = 0;
    while (i < D.size()) {
      size_t j = i;
      while (j < D.size() && D[j].prof == D[i].prof && D[j].rep == D[i].rep) j++;
      if (j - last >= target) { cut.push_back(j); last = j; }
      i = j;
    }
    if (cut.back() != D.size()) cut.push_back(D.size());
  }
  const size_t nb = cut.size() - 1;
  auto format_block = [&](size_t b, std::string &out) {
    char line[1024];
    size_t i = cut[b];
    const size_t end = cut[b + 1];
    while (i < end) {
      size_t j = i; int nrep = 0;
      while (j < end && D[j].prof == D[i].prof && D[j].rep == D[i].rep) { nrep += D[j].dom_reported; j++; }
      int k = 0;
      const int32_t smp = ctx->usample(D[i].rep);
      if (ctx->S > 1 && ctx->sel_sample >= 0 && smp != ctx->sel_sample) { i = j; continue; }
      const std::string tname = nrep ? read_name(ctx, ctx->h_seed_read[D[i].rep]) : std::string();
      for (size_t d = i; d < j; d++) {
        if (!D[d].dom_reported) continue;
        k++;
        const HostProfile &h = ctx->profs[D[d].prof];
        const double Z = (double)Zs[(size_t)smp], dz = (double)ctx->domz[(size_t)smp * ctx->P + D[d].prof];
        const double seqE = Z * det_exp(exp_logsurv((double)D[d].seq_score, (double)h.evparam[4], (double)h.evparam[5]));
        const double P = det_exp(D[d].lnP);
        // hmm/ali coordinates and acc need the optimal-accuracy alignment, which the engine does not compute:
        // envelope coordinates are written in their place (the reference reads only env coords and the score).
        const int len = snprintf(line, sizeof(line), "%-20s %-10s %5d %-20s %-10s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5d %5d %5d %5d %4.2f %s\n",
              tname.c_str(), "-", D[d].tlen, h.name.c_str(), "-", h.M, seqE, D[d].seq_score, D[d].seq_bias,
              k, nrep, P * dz, P * Z, D[d].bitscore, D[d].dombias / 0.69314718055994529, 1, h.M, D[d].ienv, D[d].jenv, D[d].ienv, D[d].jenv, 0.0, "-");
        if (len < 0) continue;
        if ((size_t)len < sizeof(line)) out.append(line, (size_t)len);
        else {                                        // a label longer than the line buffer
          std::string big((size_t)len + 1, '\0');
          snprintf(&big[0], big.size(), "%-20s %-10s %5d %-20s %-10s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5d %5d %5d %5d %4.2f %s\n",
              tname.c_str(), "-", D[d].tlen, h.name.c_str(), "-", h.M, seqE, D[d].seq_score, D[d].seq_bias,
              k, nrep, P * dz, P * Z, D[d].bitscore, D[d].dombias / 0.69314718055994529, 1, h.M, D[d].ienv, D[d].jenv, D[d].ienv, D[d].jenv, 0.0, "-");
          out.append(big.data(), (size_t)len);
        }
      }
      i = j;
    }
  };
  bool io_ok = write_blocks(f, nb, format_block);
  if (fclose(f) != 0) io_ok = false;
  if (!io_ok) SET_ERR(ctx, ITSX_E_IO, std::string("short write to ") + path);
  return ITSX_OK;
}

// labels of the loaded reads (identifier up to the first blank), concatenated; offsets[n+1].  names == NULL: offsets only
// (offsets[n] = bytes needed).  Reads handed over without names are labelled r%09d, as the writers label them.
int itsx_get_read_names(const itsx_ctx *ctx, char *names, int64_t cap, int64_t *offsets)
{
  CTXCHK(ctx && offsets);
  if (!ctx->h_names.empty() && (int64_t)ctx->h_names.size() == ctx->N) {      // the labels as they are kept: one copy
    const int64_t tot = ctx->h_names.off[(size_t)ctx->N];
    if (names) { if (tot > cap) SET_ERR(ctx, ITSX_E_ARG, "name buffer too small"); if (tot) memcpy(names, ctx->h_names.blob.data(), (size_t)tot); }
    memcpy(offsets, ctx->h_names.off.data(), ((size_t)ctx->N + 1) * sizeof(int64_t));
    return ITSX_OK;
  }
  int64_t o = 0;
  for (int64_t r = 0; r < ctx->N; r++) {
    offsets[r] = o;
    const std::string nm = read_name(ctx, r);
    if (names) { if (o + (int64_t)nm.size() > cap) SET_ERR(ctx, ITSX_E_ARG, "name buffer too small"); memcpy(names + o, nm.data(), nm.size()); }
    o += (int64_t)nm.size();
  }
  offsets[ctx->N] = o;
  return ITSX_OK;
}

// the library's environment switches (csrc/switches.cpp) that are set: of the last search of ctx (snapshot taken when it started), or, with
// ctx == NULL, of the environment right now.  "NAME=value\n" lines; returns the length needed (incl. the terminator)
int64_t itsx_switches(const itsx_ctx *ctx, char *buf, int64_t cap)
{
  const std::string s = ctx ? ctx->switches_at_search : sw_report();
  if (buf && cap > 0) { const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1); memcpy(buf, s.data(), n); buf[n] = 0; }
  return (int64_t)s.size() + 1;
}

// the registry itself: "NAME\tclass\tmeaning\n" lines (class: tuning | mode | diagnostic | hook)
int64_t itsx_switch_registry(char *buf, int64_t cap)
{
  static const char *kinds[] = {"tuning", "mode", "diagnostic", "hook"};
  int n = 0; const Switch *r = sw_registry(&n);
  std::string s;
  for (int i = 0; i < n; i++) { s += r[i].name; s += "\t"; s += kinds[r[i].kind]; s += "\t"; s += r[i].what; s += "\n"; }
  if (buf && cap > 0) { const size_t m = std::min<size_t>(s.size(), (size_t)cap - 1); memcpy(buf, s.data(), m); buf[m] = 0; }
  return (int64_t)s.size() + 1;
}

int itsx_get_stats(const itsx_ctx *ctx, itsx_stats *out, int64_t out_size)
{
  CTXCHK(ctx && out);
  if (out_size != (int64_t)sizeof(itsx_stats)) SET_ERR(ctx, ITSX_E_ARG, "itsx_get_stats: the caller's itsx_stats has " + std::to_string(out_size) + " bytes, this library's " + std::to_string(sizeof(itsx_stats)) + " (header and library from different sources)");
  *out = ctx->stats;
  return ITSX_OK;
}

// ------------------------------------------------------------------------------ test hooks
int itsx_debug_read_hashes(itsx_ctx *ctx, uint64_t *fwd, uint64_t *rc)
{
  CTXCHK(ctx && fwd && rc);
  HIPCHK(hipSetDevice(ctx->device));
  DBuf<uint64_t> hf, hr;
  HIPCHK(hf.alloc((size_t)ctx->N + 1)); HIPCHK(hr.alloc((size_t)ctx->N + 1));
  if (ctx->N > 0) {
    launch_hash_reads(ctx->rd, 0, 1, hf.p, hr.p, ctx->st);
    HIPCHK(hipMemcpyAsync(fwd, hf.p, (size_t)ctx->N * 8, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(rc, hr.p, (size_t)ctx->N * 8, hipMemcpyDeviceToHost, ctx->st));
  }
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}

int itsx_debug_packed_read(const itsx_ctx *ctx, int64_t i, uint32_t *words, int32_t *nwords, uint32_t *exc, int32_t *nexc)
{
  CTXCHK(ctx && i >= 0 && i < ctx->N && nwords && nexc);
  HIPCHK(hipSetDevice(ctx->device));
  int64_t eo[2] = {0, 0};
  HIPCHK(hipMemcpy(eo, ctx->d_excoff.p + i, sizeof(eo), hipMemcpyDeviceToHost));
  const int32_t nw = (int32_t)(ctx->h_woff[i + 1] - ctx->h_woff[i]), ne = (int32_t)(eo[1] - eo[0]);
  if (words) HIPCHK(hipMemcpy(words, ctx->d_words.p + ctx->h_woff[i], (size_t)nw * 4, hipMemcpyDeviceToHost));
  if (exc && ne) HIPCHK(hipMemcpy(exc, ctx->d_exc.p + eo[0], (size_t)ne * 4, hipMemcpyDeviceToHost));
  *nwords = nw; *nexc = ne;
  return ITSX_OK;
}

// streams `gbytes` GB in slab pattern `pattern` (k_util.hip: launch_calib) `iters` times; returns the bytes one launch touches
int itsx_debug_calibrate(itsx_ctx *ctx, int pattern, double gbytes, int iters, int64_t *bytes_per_launch, double *ms_per_launch)
{
  CTXCHK(ctx && pattern >= 0 && pattern <= 3 && gbytes > 0 && iters >= 1);
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t nwaves = 256 * 32;                          // 8 waves per SIMD, as k_decode runs
  const int64_t row_floats = pattern == 3 ? 4 * 64 : 6 * 64;
  const int64_t R = std::max<int64_t>(1, (int64_t)(gbytes * 1e9) / (nwaves * row_floats * 4));
  DBuf<float> slab, out;
  HIPCHK(slab.alloc((size_t)(nwaves * R * row_floats), true)); HIPCHK(out.alloc((size_t)nwaves * 64));
  HIPCHK(hipMemsetAsync(slab.p, 0, (size_t)(nwaves * R * row_floats) * 4, ctx->st));
  StageTimer tm(ctx->st);
  for (int i = 0; i < iters; i++) launch_calib(pattern, slab.p, nwaves, R, out.p, ctx->st);
  const float ms = tm.stop();
  HIPCHK(hipGetLastError());
  const int64_t touched = pattern == 1 ? 5 * 64 * 4 : row_floats * 4;
  if (bytes_per_launch) *bytes_per_launch = nwaves * R * touched;
  if (ms_per_launch) *ms_per_launch = ms / iters;
  return ITSX_OK;
}

// VALU issue rates (profiles/round5_valu_issue.md): one instruction class, no dependence between consecutive instructions,
// waves_per_simd waves on every SIMD of the chip.  cycles_per_instr = median over waves of s_memtime ticks per instruction as ONE wave
// sees them (divide by waves_per_simd for the SIMD's issue interval); ms = the launch.
int itsx_debug_issue(itsx_ctx *ctx, int op, int waves_per_simd, int iters, double *cycles_per_instr, double *ms)
{
  CTXCHK(ctx && op >= 0 && op <= 11 && waves_per_simd >= 1 && waves_per_simd <= 8 && (waves_per_simd <= 4 || waves_per_simd % 2 == 0) && iters >= 1);
  HIPCHK(hipSetDevice(ctx->device));
  int ncu = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, ctx->device) == hipSuccess) ncu = pr.multiProcessorCount; }
  const int nw = ncu * 4 * waves_per_simd;
  DBuf<unsigned long long> ticks; DBuf<float> sink;
  HIPCHK(ticks.alloc((size_t)nw)); HIPCHK(sink.alloc(1024));          // (the scalar probes read 1.1 KB of it)
  HIPCHK(hipMemsetAsync(sink.p, 0, 1024 * sizeof(float), ctx->st));
  launch_issue(op, waves_per_simd, 16, ncu, ticks.p, sink.p, ctx->st);          // warm-up (clocks, code)
  StageTimer tm(ctx->st);
  launch_issue(op, waves_per_simd, iters, ncu, ticks.p, sink.p, ctx->st);
  const float t = tm.stop();
  HIPCHK(hipGetLastError());
  std::vector<unsigned long long> h((size_t)nw);
  HIPCHK(hipMemcpy(h.data(), ticks.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  if (cycles_per_instr) *cycles_per_instr = (double)h[h.size() / 2] / (64.0 * (double)iters);
  if (ms) *ms = t;
  return ITSX_OK;
}

int itsx_debug_detmath(itsx_ctx *ctx, const double *x, int64_t n, double *out_log, double *out_exp)
{
  CTXCHK(ctx && x && out_log && out_exp && n >= 0);
  HIPCHK(hipSetDevice(ctx->device));
  DBuf<double> dx, dl, de;
  HIPCHK(dx.alloc((size_t)n + 1)); HIPCHK(dl.alloc((size_t)n + 1)); HIPCHK(de.alloc((size_t)n + 1));
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->st));
    launch_detmath(dx.p, n, dl.p, de.p, ctx->st);
    HIPCHK(hipMemcpyAsync(out_log, dl.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(out_exp, de.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->st));
  }
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}

int itsx_debug_dust(itsx_ctx *ctx, uint8_t *masked)
{
  CTXCHK(ctx && masked);
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n = ctx->N;
  if (n == 0) return ITSX_OK;
  DBuf<uint32_t> dm;
  HIPCHK(dm.alloc((size_t)ctx->h_woff[n] + 1));
  launch_dust(ctx->rd, dm.p, ctx->st);
  std::vector<uint32_t> h((size_t)ctx->h_woff[n] + 1);
  HIPCHK(hipMemcpyAsync(h.data(), dm.p, h.size() * 4, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipGetLastError());
  int64_t o = 0;
  for (int64_t r = 0; r < n; r++)
    for (int p = 0; p < ctx->h_len[r]; p++) masked[o++] = (uint8_t)((h[(size_t)ctx->h_woff[r] + (p >> 5)] >> (p & 31)) & 1u);
  return ITSX_OK;
}

int itsx_debug_logf(itsx_ctx *ctx, const float *x, int64_t n, float *out)
{
  CTXCHK(ctx && x && out && n >= 0);
  HIPCHK(hipSetDevice(ctx->device));
  DBuf<float> dx, dy;
  HIPCHK(dx.alloc((size_t)n + 1)); HIPCHK(dy.alloc((size_t)n + 1));
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, ctx->st));
    launch_logf_fast(dx.p, n, ctx->d_logtab.p, dy.p, ctx->st);
    HIPCHK(hipMemcpyAsync(out, dy.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->st));
  }
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ITSX_OK;
}

}  // extern "C"
