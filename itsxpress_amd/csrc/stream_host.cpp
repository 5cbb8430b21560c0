// stream_host.cpp -- the two host-side pieces of a STREAMING file-to-file run (itsxpress_amd/stream.py): the file's text handed out
// in record-aligned slices while it is still being inflated, and the set of sequences seen so far.
//
// The reference reads the whole FASTQ before vsearch sees it and the whole uc.txt / domtbl.txt before it writes (main.py:534-624,
// SeqSample.py:93-131,178-225,886-949); one GPU outruns the inflater, so a large .fastq.gz is cut into file-order chunks, each in a
// context of its own: chunk k is dereplicated and its NEW sequences scored while chunk k + 1 is inflated and parsed.  Exactness is
// the multi-GPU scheme's: a sequence is scored once, where it first occurs -- which in file order IS vsearch's
// representative -- and hmmsearch's domZ is summed over the chunks before any threshold is applied.
// Host-only, no GPU call, no arithmetic of the path.
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "../../include/itsx_hip.h"
#include "fastq_io.h"

struct itsx_stream {
  itsx_io::TextStream ts;
  std::string err;
};

// open addressing over the 128-bit orientation-free keys of itsx_unique_keys128; value = the first holder.  PARTS sub-tables by the
// key's top bits, each filled by ONE thread of itsx_keyset_assign's pool: a chunk's 0.75-1.5 M lookups into a table of hundreds of MB are
// cache misses one after the other on one thread (1.9 s of a 10 M-read streamed run's loader, round 5), independent across partitions
struct itsx_keyset {
  struct Slot { uint64_t k0, k1; int64_t gidx, fwd, chunk, lu, id; };
  static constexpr int PARTS = 16;
  struct Part {
    std::vector<Slot> tab;
    std::vector<uint8_t> used;
    size_t n = 0, mask = 0;
    void grow(size_t cap)
    {
      std::vector<Slot> ot; ot.swap(tab);
      std::vector<uint8_t> ou; ou.swap(used);
      tab.assign(cap, Slot{}); used.assign(cap, 0); mask = cap - 1;
      for (size_t i = 0; i < ot.size(); i++) if (ou[i]) { size_t h = (size_t)(ot[i].k0 ^ (ot[i].k1 * 0x9E3779B97F4A7C15ull)) & mask; while (used[h]) h = (h + 1) & mask; tab[h] = ot[i]; used[h] = 1; }
    }
  };
  Part part[PARTS];
  size_t n = 0;                          // distinct keys so far = the next id
  static int part_of(uint64_t k0, uint64_t k1) { return (int)(((k0 * 0x9E3779B97F4A7C15ull) ^ k1) >> 60) & (PARTS - 1); }
};

namespace { std::string g_stream_error; }

extern "C" {

const char *itsx_stream_last_error(void) { return g_stream_error.c_str(); }

int itsx_stream_open(const char *path, itsx_stream **out)
{
  if (!path || !out) { g_stream_error = "itsx_stream_open: missing argument"; return ITSX_E_ARG; }
  itsx_stream *s = new itsx_stream;
  if (!s->ts.open(path, s->err)) { g_stream_error = s->err; delete s; return ITSX_E_IO; }
  *out = s;
  return ITSX_OK;
}

// the text goes into a shared mapping of `backing` (a new, sparse file the caller unlinks): other processes map the slices they are told
// about -- offset = slice address - itsx_stream_base(s).  *plain_input = 1: the input is uncompressed and was NOT copied; the offsets
// are offsets into the input file itself.
int itsx_stream_open_shared(const char *path, const char *backing, itsx_stream **out, int32_t *plain_input)
{
  if (!path || !backing || !out) { g_stream_error = "itsx_stream_open_shared: missing argument"; return ITSX_E_ARG; }
  itsx_stream *s = new itsx_stream;
  bool plain = false;
  if (!s->ts.open(path, s->err, backing, &plain)) { g_stream_error = s->err; delete s; return ITSX_E_IO; }
  if (plain_input) *plain_input = plain ? 1 : 0;
  *out = s;
  return ITSX_OK;
}
// the same as itsx_stream_open with a pool of `threads` inflating threads (a paired run's two streams share the CPUs)
int itsx_stream_open_threads(const char *path, int32_t threads, itsx_stream **out)
{
  if (!path || !out) { g_stream_error = "itsx_stream_open_threads: missing argument"; return ITSX_E_ARG; }
  itsx_stream *s = new itsx_stream;
  if (!s->ts.open(path, s->err, nullptr, nullptr, threads)) { g_stream_error = s->err; delete s; return ITSX_E_IO; }
  *out = s;
  return ITSX_OK;
}
const char *itsx_stream_base(itsx_stream *s) { return s ? s->ts.base() : nullptr; }
int itsx_stream_progress(itsx_stream *s, int64_t *avail, int64_t *consumed, int64_t *raw_size)
{
  if (!s) return ITSX_E_ARG;
  size_t a = 0, c = 0, r = 0;
  s->ts.progress(&a, &c, &r);
  if (avail) *avail = (int64_t)a;
  if (consumed) *consumed = (int64_t)c;
  if (raw_size) *raw_size = (int64_t)r;
  return ITSX_OK;
}

int itsx_stream_next(itsx_stream *s, int64_t min_bytes, const char **text, int64_t *nbytes, int32_t *last)
{
  if (!s || !text || !nbytes || !last) { g_stream_error = "itsx_stream_next: missing argument"; return ITSX_E_ARG; }
  size_t nb = 0; bool l = false;
  if (!s->ts.next((size_t)(min_bytes > 0 ? min_bytes : 1), text, &nb, &l, s->err)) { g_stream_error = s->err; return ITSX_E_IO; }
  *nbytes = (int64_t)nb; *last = l ? 1 : 0;
  return ITSX_OK;
}

// the mate file's slice: exactly n_records records (fewer only at the end of the file)
int itsx_stream_next_records(itsx_stream *s, int64_t n_records, const char **text, int64_t *nbytes, int64_t *got, int32_t *last)
{
  if (!s || !text || !nbytes || !got || !last || n_records < 0) { g_stream_error = "itsx_stream_next_records: missing argument"; return ITSX_E_ARG; }
  size_t nb = 0, g = 0; bool l = false;
  if (!s->ts.next_records((size_t)n_records, text, &nb, &g, &l, s->err)) { g_stream_error = s->err; return ITSX_E_IO; }
  *nbytes = (int64_t)nb; *got = (int64_t)g; *last = l ? 1 : 0;
  return ITSX_OK;
}
// FASTQ records in a record-aligned piece of text (lines / 4; a last line without its newline counts), on the I/O pool
int64_t itsx_count_records(const char *text, int64_t nbytes)
{
  if (!text || nbytes <= 0) return 0;
  const size_t n = (size_t)nbytes;
  const int T = std::max(1, std::min(itsx_io::io_threads(), (int)(n >> 24) + 1));
  std::vector<size_t> cnt((size_t)T, 0);
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++)
    th.emplace_back([&, k] {
      const char *p = text + n / (size_t)T * (size_t)k, *e = (k + 1 == T) ? text + n : text + n / (size_t)T * (size_t)(k + 1);
      size_t c = 0;
      while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; c++; p = q + 1; }
      cnt[(size_t)k] = c;
    });
  for (auto &x : th) x.join();
  size_t lines = text[n - 1] != '\n' ? 1 : 0;
  for (size_t c : cnt) lines += c;
  return (int64_t)(lines / 4);
}

int64_t itsx_stream_records_bound(itsx_stream *s) { return s ? (int64_t)s->ts.records_bound() : -1; }

int itsx_stream_close(itsx_stream *s, int32_t keep_text)
{
  if (!s) return ITSX_OK;
  const bool ok = s->ts.finish(keep_text != 0, s->err);
  if (!ok) g_stream_error = s->err;
  delete s;
  return ok ? ITSX_OK : ITSX_E_IO;
}

itsx_keyset *itsx_keyset_create(void)
{
  itsx_keyset *k = new itsx_keyset;
  for (auto &p : k->part) p.grow((size_t)1 << 14);
  return k;
}
void itsx_keyset_destroy(itsx_keyset *k) { delete k; }
int64_t itsx_keyset_size(const itsx_keyset *k) { return k ? (int64_t)k->n : 0; }

int itsx_keyset_assign(itsx_keyset *k, const int64_t *tuples, int64_t n_unique, int32_t chunk, int64_t *verdict, int64_t *gid)
{
  if (!k || n_unique < 0 || (n_unique > 0 && (!tuples || !verdict))) { g_stream_error = "itsx_keyset_assign: missing argument"; return ITSX_E_ARG; }
  constexpr int PARTS = itsx_keyset::PARTS;
  // where every tuple's slot is (partition, index) and whether it is new: by partition, in parallel; a partition's thread walks the
  // tuples in order, so the FIRST holder of a key inside this call is the earlier unique, as on one thread
  std::vector<uint32_t> where((size_t)n_unique);
  std::vector<uint8_t> fresh((size_t)n_unique, 0);
  auto work = [&](int p) {
    itsx_keyset::Part &pt = k->part[p];
    for (int64_t u = 0; u < n_unique; u++) {
      const uint64_t k0 = (uint64_t)tuples[4 * u], k1 = (uint64_t)tuples[4 * u + 1];
      if (itsx_keyset::part_of(k0, k1) != p) continue;
      if ((pt.n + 1) * 2 > pt.tab.size()) pt.grow(pt.tab.size() * 2);
      size_t h = (size_t)(k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) & pt.mask;
      while (pt.used[h] && !(pt.tab[h].k0 == k0 && pt.tab[h].k1 == k1)) h = (h + 1) & pt.mask;
      if (!pt.used[h]) { pt.tab[h] = itsx_keyset::Slot{k0, k1, tuples[4 * u + 2], tuples[4 * u + 3], (int64_t)chunk, u, -1}; pt.used[h] = 1; pt.n++; fresh[(size_t)u] = 1; }
      where[(size_t)u] = (uint32_t)h;
    }
  };
  const int T = n_unique >= 4096 ? std::min(PARTS, std::max(1, itsx_io::io_threads())) : 1;
  if (T <= 1) { for (int p = 0; p < PARTS; p++) work(p); }
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] { for (int p = t; p < PARTS; p += T) work(p); });
    for (auto &x : th) x.join();
  }
  // (a table that grew after a tuple's slot was noted moved that slot: the slots of a partition that grew in this call are looked up again)
  // ids in first-seen order: the new keys of this call are numbered by their unique number
  int64_t next = (int64_t)k->n;
  for (int64_t u = 0; u < n_unique; u++) {
    const uint64_t k0 = (uint64_t)tuples[4 * u], k1 = (uint64_t)tuples[4 * u + 1];
    itsx_keyset::Part &pt = k->part[itsx_keyset::part_of(k0, k1)];
    size_t h = where[(size_t)u];
    if (h > pt.mask || !pt.used[h] || pt.tab[h].k0 != k0 || pt.tab[h].k1 != k1) {      // the partition was rehashed since
      h = (size_t)(k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) & pt.mask;
      while (!(pt.used[h] && pt.tab[h].k0 == k0 && pt.tab[h].k1 == k1)) h = (h + 1) & pt.mask;
    }
    itsx_keyset::Slot &s = pt.tab[h];
    if (fresh[(size_t)u]) s.id = next++;
    verdict[4 * u] = s.gidx; verdict[4 * u + 1] = s.fwd; verdict[4 * u + 2] = s.chunk; verdict[4 * u + 3] = s.lu;
    if (gid) gid[u] = s.id;
  }
  k->n = (size_t)next;
  return ITSX_OK;
}

}  // extern "C"
