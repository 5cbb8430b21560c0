// trim_host.cpp -- SURVEY section 8(f) rank 1: native FASTQ parse -> slice -> write for the coordinates the
// engine produces.  Host-only (this is I/O, not arithmetic); context-free so that it is testable without a GPU.
//
// Replaces the Biopython loops of the reference:
//   Dedup.create_trimmed_seqs          itsxpress/SeqSample.py:886-949 (+ _get_trimmed_seq_generator 792-884)
//   Dedup.create_paired_trimmed_seqs   itsxpress/SeqSample.py:713-790 (+ _get_paired_seq_generator 564-711)
// Semantics kept: a record is written iff both boundaries exist and start < stop; single-end slice
// record[start:stop]; paired r2start = tlen - stop, r2end = tlen - start, R1[start:] if stop > tlen else
// R1[start:stop], R2[r2start:] if r2end > tlen else R2[r2start:r2end] (Python slice clamping, negative indices
// included); --trim-ccs primer stitching with quality 93; the title line is written back verbatim and the
// '+' line is bare, as Biopython's FASTQ writer does.  Pinned byte-for-byte by the reference's t2_r1.fq /
// t2_r2.fq goldens (tests/test_trim_cpu.py).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include "../../include/itsx_hip.h"
#include "fastq_io.h"

namespace {

std::string g_trim_error;

// The whole (decompressed) input is in memory (fastq_io.h: read_text, shared with the engine's loader through the
// text cache); records are views into it.
struct View {
  const char *p = nullptr; size_t n = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};
struct Rec { View title, seq, qual; };
struct Records {
  std::shared_ptr<const itsx_io::Text> text;
  const char *s = nullptr, *end = nullptr;
  bool open(const char *path)
  {
    std::string err;
    text = itsx_io::read_text(path, err, true);
    if (!text) { g_trim_error = err; return false; }
    s = text->data(); end = s + text->size();
    return true;
  }
  bool line(View &v)
  {
    if (s >= end) return false;
    const char *nl = (const char *)memchr(s, '\n', (size_t)(end - s));
    const char *e = nl ? nl : end;
    v.p = s; s = nl ? nl + 1 : end;
    if (e > v.p && e[-1] == '\r') e--;
    v.n = (size_t)(e - v.p);
    return true;
  }
  // returns 1 on a record, 0 at end of file, -1 on malformed input
  int next(Rec &rec)
  {
    View plus;
    do { if (!line(rec.title)) return 0; } while (rec.title.empty());
    if (rec.title.p[0] != '@') return -1;
    if (!line(rec.seq) || !line(plus) || !line(rec.qual)) return -1;
    if (plus.empty() || plus.p[0] != '+' || rec.qual.size() != rec.seq.size()) return -1;
    return 1;
  }
};
struct Writer {
  itsx_io::BlockWriter w;
  std::string buf;
  bool open(const char *path, int kind, bool keep_text = false)
  {
    std::string err;
    if (!w.open(path, kind, err, keep_text)) { g_trim_error = err; return false; }
    buf.reserve(1 << 20);
    return true;
  }
  void flush() { if (!buf.empty()) { w.put(buf); buf.clear(); } }
  bool close()
  {
    flush();
    std::string err;
    if (!w.close(err)) { g_trim_error = err; return false; }
    return true;
  }
};

std::string id_of(const View &title)
{
  size_t e = 1; while (e < title.n && title.p[e] != ' ' && title.p[e] != '\t') e++;
  return std::string(title.p + 1, e - 1);
}
// Python's seq[a:b] for a sequence of length n (b_open: no upper bound)
void py_slice(int64_t n, int64_t a, int64_t b, bool b_open, int64_t &lo, int64_t &hi)
{
  if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
  if (b_open) b = n;
  else if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
  lo = a; hi = b < a ? a : b;
}
void emit(Writer &w, const View &title, const char *seq, const char *qual, int64_t lo, int64_t hi, bool ccs, int64_t *total)
{
  static const char *fwd = "GACAGGTACAAGAAGGA", *rev = "TTAACCCAGTCTCCAGT";
  std::string &out = w.buf;
  out.append(title.p, title.n); out += '\n';
  if (ccs) out += fwd;
  out.append(seq + lo, (size_t)(hi - lo));
  if (ccs) out += rev;
  out += "\n+\n";
  if (ccs) out.append(17, '~');
  out.append(qual + lo, (size_t)(hi - lo));
  if (ccs) out.append(17, '~');
  out += '\n';
  if (out.size() >= (1u << 20) - 4096) w.flush();
  if (total) *total += (hi - lo) + (ccs ? 34 : 0);
}

}  // namespace

extern "C" {

const char *itsx_trim_last_error(void) { return g_trim_error.c_str(); }

// a cursor over a byte range of a FASTQ text that starts at a record start (the record parser of Records, on a slice)
static bool fastq_record_start(const char *t, const char *end, const char *p)
{
  // a line that starts with '@' and whose second line below starts with '+' (a quality line may start with '@' too, but then that
  // second line is a sequence line): the rule the engine's loader cuts large texts by
  if (p >= end || *p != '@') return false;
  const char *l1 = (const char *)memchr(p, '\n', (size_t)(end - p));
  if (!l1) return false;
  const char *l2 = (const char *)memchr(l1 + 1, '\n', (size_t)(end - (l1 + 1)));
  if (!l2 || l2 + 1 >= end) return false;
  (void)t;
  return l2[1] == '+';
}

int itsx_write_trimmed_fastq(const char *seq_path, const char *out_path, int compression, int trim_ccs,
                             const int32_t *start, const int32_t *stop, int64_t n_records,
                             int64_t *n_written, int64_t *total_len)
{
  if (!seq_path || !out_path || !start || !stop) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  if (compression < 0 || compression > 2) { g_trim_error = "compression must be 0 (plain), 1 (gzip) or 2 (zstd)"; return ITSX_E_ARG; }
  Records in; Writer out;
  if (!in.open(seq_path)) return ITSX_E_IO;
  // The writer proper is itsx_twriter (below): units of the text sliced and deflated by a pool of threads, written in order --
  // fed here with the whole text and every coordinate at once, by a streaming run piece by piece: the same bytes either way.
  // (One I/O thread: the serial walk below; a malformed record sends the file there too, and the walk names it.)
  const size_t size = (size_t)(in.end - in.s);
  const size_t min_par = sw_get("ITSX_WRITE_MIN_MB") ? (size_t)atoll(sw_get("ITSX_WRITE_MIN_MB")) << 20 : 0;
  if (itsx_io::io_threads() > 1 && size >= min_par) {
    { itsx_io::PieceCompressor probe(compression); if (!probe.ok()) { g_trim_error = "zstd output requested but libzstd.so.1 could not be loaded"; return ITSX_E_IO; } }
    itsx_twriter *tw = nullptr;
    int rc = itsx_twriter_open(out_path, compression, trim_ccs, &tw);
    if (rc != ITSX_OK) return rc;
    rc = itsx_twriter_text(tw, in.s, (int64_t)size, 1);
    if (rc == ITSX_OK) rc = itsx_twriter_coords(tw, 0, n_records, start, stop, nullptr);
    int64_t nw = 0, tot = 0;
    const int crc = itsx_twriter_close(tw, &nw, &tot);
    if (rc == ITSX_OK) rc = crc;
    if (rc == ITSX_OK) {
      if (n_written) *n_written = nw;
      if (total_len) *total_len = tot;
      return ITSX_OK;
    }
    if (rc != ITSX_E_FORMAT) return rc;
    // the pool's writer object met a malformed record: a read-only walk names it (the output is NOT opened a second time -- what the
    // object wrote before the record stays as it is, as after the single-threaded walk below; round 5's advisor)
    Rec bad; int64_t k = 0; int rr;
    while ((rr = in.next(bad)) == 1) k++;
    g_trim_error = rr < 0 ? "malformed FASTQ record " + std::to_string(k) : std::string("malformed FASTQ text");
    return ITSX_E_FORMAT;
  }
  if (!out.open(out_path, compression)) return ITSX_E_IO;        // (the single-threaded way through: one I/O thread, or a text below ITSX_WRITE_MIN_MB)
  Rec rec; int64_t i = 0, nw = 0, tot = 0; int rc;
  while ((rc = in.next(rec)) == 1) {
    if (i >= n_records) { g_trim_error = "more records in the file than coordinates"; return ITSX_E_ARG; }
    const int64_t a = start[i], b = stop[i];
    i++;
    if (a < 0 || b < 0 || !(a < b)) continue;
    int64_t lo, hi; py_slice((int64_t)rec.seq.size(), a, b, false, lo, hi);
    emit(out, rec.title, rec.seq.p, rec.qual.p, lo, hi, trim_ccs != 0, &tot);
    nw++;
  }
  if (rc < 0) { g_trim_error = "malformed FASTQ record " + std::to_string(i); return ITSX_E_FORMAT; }
  if (!out.close()) return ITSX_E_IO;
  if (n_written) *n_written = nw;
  if (total_len) *total_len = tot;
  return ITSX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// itsx_twriter_*: the trimmed-FASTQ writer as an object that takes the input's text and the coordinates PIECE BY PIECE and works
// while more arrive (a streaming file-to-file run deflates the first chunks' reads while the GPU scores the later chunks).
// The text is cut into UNITS at the first record start at or after every multiple of 8 MB of the text: a unit's output is one
// gzip member / zstd frame of its own (a unit without a surviving record adds nothing), so a unit depends on its own records only,
// the units are written in order, and the bytes of the file do not depend on how text and coordinates arrived.
// itsx_write_trimmed_fastq is the same object fed once.  A record whose coordinates are not decided yet (`decided[i] == 0`) holds
// back its unit alone, until itsx_twriter_update names it.
}  // extern "C" (the writer object's types are C++)

// grows at the end without ever moving what it holds: the pool's threads read rows while the caller appends more
template <class T> struct StableVec {
  static constexpr size_t LOG = 20, BLK = (size_t)1 << LOG;
  std::vector<std::unique_ptr<T[]>> blocks; size_t n = 0;
  StableVec() { blocks.reserve(1 << 16); }            // 6.9e10 rows before the table itself would move
  size_t size() const { return n; }
  T &operator[](size_t i) { return blocks[i >> LOG][i & (BLK - 1)]; }
  const T &operator[](size_t i) const { return blocks[i >> LOG][i & (BLK - 1)]; }
  void append(const T *src, size_t m, T fill)
  {
    for (size_t k = 0; k < m;) {
      if ((n >> LOG) >= blocks.size()) blocks.emplace_back(new T[BLK]);
      const size_t at = n & (BLK - 1), take = std::min(m - k, BLK - at);
      T *dst = blocks[n >> LOG].get() + at;
      if (src) memcpy(dst, src + k, take * sizeof(T)); else for (size_t q = 0; q < take; q++) dst[q] = fill;
      n += take; k += take;
    }
  }
};

struct itsx_twriter {
  struct Unit {
    size_t lo = 0, hi = 0;                 // text range (record-aligned)
    int64_t count = -1, first = -1;        // records inside (counted by the pool), index of the first
    int64_t undecided = -1;                // -1: coordinates not complete yet
    int state = 0;                         // 0 cut, 1 being sliced, 2 done
    std::string comp; int64_t nw = 0, tot = 0;
  };
  std::string path; int kind = 0; bool ccs = false; size_t unit_bytes = (size_t)8 << 20;
  int mode = 0;                            // 1: (start, stop) are Python slice bounds as they come (a paired run's mates: start may be negative,
                                           // stop == INT32_MAX = open end, stop == INT32_MIN = the record is not written)
  int fd = -1; uint64_t file_off = 0;      // the output, written at explicit offsets (a burst of finished units goes out on several threads)
  bool seekable = true;                    // (a pipe -- /dev/stdout, a process substitution -- takes the pieces one after the other)
  const char *base = nullptr; size_t avail = 0; bool text_done = false;
  size_t next_cut_at = 0;                  // the next multiple of unit_bytes to cut behind
  std::deque<Unit> units;
  size_t counted_prefix = 0;               // units [0, counted_prefix) know their `first`
  int64_t records_cut = 0;                 // records in those units
  StableVec<int32_t> start, stop; StableVec<uint8_t> decided;
  size_t next_write = 0; bool flushing = false;
  bool wrote_any = false, failed = false, malformed = false, closing = false;
  std::string err;
  int64_t nw = 0, tot = 0;
  // pool
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_work, cv_idle;
  std::deque<std::pair<int, size_t>> jobs;  // (0 count | 1 slice, unit)
  int busy = 0; bool quit = false;

  // ---- all of these with `mu` held
  void cut_units()
  {
    // a cut needs the record-start test's two following lines: only where enough text is final (or the text is complete)
    for (;;) {
      const size_t lo = units.empty() ? 0 : units.back().hi;
      if (lo >= avail) break;
      size_t hi;
      const size_t want = next_cut_at + unit_bytes;
      if (want >= avail) {
        if (!text_done) break;
        hi = avail;
      } else {
        const char *t0 = base, *tend = base + avail;
        const char *p = (const char *)memchr(t0 + want, '\n', avail - want);
        while (p && p + 1 < tend && !fastq_record_start_at(t0, tend, p + 1)) p = (const char *)memchr(p + 1, '\n', (size_t)(tend - (p + 1)));
        if (!p || p + 1 >= tend) { if (!text_done) break; hi = avail; }
        else hi = (size_t)(p + 1 - t0);
      }
      next_cut_at = want;
      if (hi <= lo) continue;               // (a record longer than a unit)
      Unit u; u.lo = lo; u.hi = hi;
      units.push_back(std::move(u));
      jobs.emplace_back(0, units.size() - 1);
      if (hi >= avail) break;
    }
    cv_work.notify_all();
  }
  void advance()
  {
    while (counted_prefix < units.size() && units[counted_prefix].count >= 0) {
      units[counted_prefix].first = records_cut;
      records_cut += units[counted_prefix].count;
      counted_prefix++;
    }
    for (size_t k = next_write; k < counted_prefix; k++) {
      Unit &u = units[k];
      if (u.state != 0) continue;
      if (u.first + u.count > (int64_t)start.size()) break;      // its coordinates have not arrived
      if (u.undecided < 0) { int64_t c = 0; for (int64_t i = u.first; i < u.first + u.count; i++) c += decided[(size_t)i] == 0; u.undecided = c; }
      if (u.undecided == 0) { u.state = 1; jobs.emplace_back(1, k); }
    }
    cv_work.notify_all();
  }
  static bool fastq_record_start_at(const char *t, const char *end, const char *p);
  void work();
  // writes every finished unit that is next in order (called without the lock)
  void flush_ready()
  {
    std::unique_lock<std::mutex> lk(mu);
    if (flushing) return;                  // somebody is at it already and will see this unit too: one writer keeps the order
    flushing = true;
    while (next_write < units.size() && units[next_write].state == 2) {
      // every finished unit that is next, up to 256 MB at a time: usually one; after a unit that waited for its last coordinates
      // (a streaming run's exact thresholds) everything behind it -- most of the file -- at once
      std::vector<std::string> run; std::vector<uint64_t> at; uint64_t bytes = 0;
      while (next_write < units.size() && units[next_write].state == 2 && bytes < ((uint64_t)256 << 20)) {
        Unit &u = units[next_write];
        nw += u.nw; tot += u.tot;
        next_write++;
        if (u.comp.empty()) continue;
        run.emplace_back(); run.back().swap(u.comp);
        at.push_back(file_off + bytes); bytes += run.back().size();
      }
      if (run.empty()) continue;
      wrote_any = true;
      file_off += bytes;
      lk.unlock();
      const bool ok = write_run(run, at);
      lk.lock();
      if (!ok) failed = true;
    }
    flushing = false;
  }
  static bool pwrite_all(int fd, const char *p, size_t n, uint64_t off, bool seekable = true)
  {
    while (n > 0) {
      const ssize_t w = seekable ? pwrite(fd, p, n, (off_t)off) : write(fd, p, n);
      if (w <= 0) return false;
      p += w; n -= (size_t)w; off += (uint64_t)w;
    }
    return true;
  }
  bool write_run(const std::vector<std::string> &run, const std::vector<uint64_t> &at) const
  {
    uint64_t bytes = 0;
    for (const auto &c : run) bytes += c.size();
    const int T = (seekable && bytes >= ((uint64_t)32 << 20) && run.size() >= 4) ? (int)std::min<size_t>(8, run.size()) : 1;
    if (T == 1) { for (size_t k = 0; k < run.size(); k++) if (!pwrite_all(fd, run[k].data(), run[k].size(), at[k], seekable)) return false; return true; }
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&] { for (size_t k = next.fetch_add(1); k < run.size(); k = next.fetch_add(1)) if (!pwrite_all(fd, run[k].data(), run[k].size(), at[k])) bad = 1; });
    for (auto &x : th) x.join();
    return !bad;
  }
};

bool itsx_twriter::fastq_record_start_at(const char *t, const char *end, const char *p) { return fastq_record_start(t, end, p); }

void itsx_twriter::work()
{
  itsx_io::PieceCompressor pc(kind);
  std::string out;
  for (;;) {
    std::pair<int, size_t> j;
    size_t lo, hi; int64_t first = 0;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_work.wait(lk, [&] { return quit || !jobs.empty(); });
      if (jobs.empty()) return;
      j = jobs.front(); jobs.pop_front();
      busy++;
      lo = units[j.second].lo; hi = units[j.second].hi; first = units[j.second].first;
    }
    Records r; r.s = base + lo; r.end = base + hi;
    Rec rec; int rc;
    if (j.first == 0) {
      int64_t n = 0;
      while ((rc = r.next(rec)) == 1) n++;
      std::lock_guard<std::mutex> lk(mu);
      if (rc < 0) malformed = true;
      units[j.second].count = n;
      advance();
      busy--;
      cv_idle.notify_all();
      continue;
    }
    out.clear();
    out.reserve((hi - lo) / 2 + 4096);
    static const char *fwd = "GACAGGTACAAGAAGGA", *rev = "TTAACCCAGTCTCCAGT";
    int64_t i = first, n_out = 0, t_out = 0;
    while ((rc = r.next(rec)) == 1) {
      const int64_t a = start[(size_t)i], b = stop[(size_t)i];      // (rows of decided records are not written to any more)
      i++;
      int64_t l, h;
      if (mode == 1) {
        if (b == INT32_MIN) continue;
        py_slice((int64_t)rec.seq.size(), a, b, b == INT32_MAX, l, h);
      } else {
        if (a < 0 || b < 0 || !(a < b)) continue;
        py_slice((int64_t)rec.seq.size(), a, b, false, l, h);
      }
      out.append(rec.title.p, rec.title.n); out += '\n';
      if (ccs) out += fwd;
      out.append(rec.seq.p + l, (size_t)(h - l));
      if (ccs) out += rev;
      out += "\n+\n";
      if (ccs) out.append(17, '~');
      out.append(rec.qual.p + l, (size_t)(h - l));
      if (ccs) out.append(17, '~');
      out += '\n';
      n_out++; t_out += (h - l) + (ccs ? 34 : 0);
    }
    std::string comp;
    bool ok = rc == 0;
    if (ok && !out.empty()) ok = pc.run(out, comp);
    {
      std::lock_guard<std::mutex> lk(mu);
      Unit &u = units[j.second];
      if (!ok) { if (rc < 0) malformed = true; else failed = true; }
      u.comp.swap(comp); u.nw = n_out; u.tot = t_out; u.state = 2;
    }
    flush_ready();
    { std::lock_guard<std::mutex> lk(mu); busy--; }
    cv_idle.notify_all();
  }
}

extern "C" {

int itsx_twriter_open(const char *out_path, int compression, int trim_ccs, itsx_twriter **out)
{
  if (!out_path || !out) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  if (compression < 0 || compression > 2) { g_trim_error = "compression must be 0 (plain), 1 (gzip) or 2 (zstd)"; return ITSX_E_ARG; }
  { itsx_io::PieceCompressor probe(compression); if (!probe.ok()) { g_trim_error = "zstd output requested but libzstd.so.1 could not be loaded"; return ITSX_E_IO; } }
  itsx_twriter *w = new itsx_twriter;
  w->path = out_path; w->kind = compression; w->ccs = trim_ccs != 0;
  if (const char *e = sw_get("ITSX_WRITE_UNIT_KB")) w->unit_bytes = std::max<size_t>(1, (size_t)atoll(e)) << 10;
  w->fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
  if (w->fd < 0) { g_trim_error = std::string("cannot write ") + out_path; delete w; return ITSX_E_IO; }
  w->seekable = lseek(w->fd, 0, SEEK_CUR) != (off_t)-1;
  const int T = itsx_io::io_threads();
  for (int t = 0; t < T; t++) w->workers.emplace_back([w] { w->work(); });
  *out = w;
  return ITSX_OK;
}

// mode 1: the coordinates are Python slice bounds as they come -- a paired run's mates, itsxpress/SeqSample.py:587-670: R1[start:stop]
// (or [start:]), R2[tlen - stop : tlen - start]; stop == INT32_MAX = open end, stop == INT32_MIN = the record is not written
int itsx_twriter_set_mode(itsx_twriter *w, int32_t mode)
{
  if (!w || mode < 0 || mode > 1) { g_trim_error = "itsx_twriter_set_mode: mode must be 0 or 1"; return ITSX_E_ARG; }
  std::lock_guard<std::mutex> lk(w->mu);
  w->mode = mode;
  return ITSX_OK;
}
int itsx_twriter_text(itsx_twriter *w, const char *base, int64_t avail, int32_t last)
{
  if (!w || avail < 0 || (!base && avail > 0)) { g_trim_error = "itsx_twriter_text: bad argument"; return ITSX_E_ARG; }
  std::lock_guard<std::mutex> lk(w->mu);
  if (w->text_done || (w->base && base != w->base) || (size_t)avail < w->avail) { g_trim_error = "itsx_twriter_text: the text grows at one address, front to back"; return ITSX_E_ARG; }
  w->base = base; w->avail = (size_t)avail; w->text_done = last != 0;
  w->cut_units();
  return ITSX_OK;
}

int itsx_twriter_coords(itsx_twriter *w, int64_t first_record, int64_t n, const int32_t *start, const int32_t *stop, const uint8_t *decided)
{
  if (!w || n < 0 || (n > 0 && (!start || !stop))) { g_trim_error = "itsx_twriter_coords: bad argument"; return ITSX_E_ARG; }
  std::lock_guard<std::mutex> lk(w->mu);
  if (first_record != (int64_t)w->start.size()) { g_trim_error = "itsx_twriter_coords: coordinates arrive in record order, without gaps"; return ITSX_E_ARG; }
  w->start.append(start, (size_t)n, 0);
  w->stop.append(stop, (size_t)n, 0);
  w->decided.append(decided, (size_t)n, 1);
  w->advance();
  return ITSX_OK;
}

int itsx_twriter_update(itsx_twriter *w, const int64_t *records, int64_t m, const int32_t *start, const int32_t *stop)
{
  if (!w || m < 0 || (m > 0 && (!records || !start || !stop))) { g_trim_error = "itsx_twriter_update: bad argument"; return ITSX_E_ARG; }
  std::lock_guard<std::mutex> lk(w->mu);
  for (int64_t k = 0; k < m; k++) {
    const int64_t r = records[k];
    if (r < 0 || r >= (int64_t)w->start.size()) { g_trim_error = "itsx_twriter_update: record out of range"; return ITSX_E_ARG; }
    if (w->decided[(size_t)r]) continue;
    w->start[(size_t)r] = start[k]; w->stop[(size_t)r] = stop[k]; w->decided[(size_t)r] = 1;
    // its unit: the one whose record range holds r (units know their ranges once counted)
    size_t a = 0, b = w->counted_prefix;
    while (a + 1 < b) { const size_t mid = (a + b) / 2; if (w->units[mid].first <= r) a = mid; else b = mid; }
    // (a unit that has not been counted yet, or whose rows were not complete, will count its undecided rows itself, later)
    if (a < w->counted_prefix && r >= w->units[a].first && r < w->units[a].first + w->units[a].count && w->units[a].undecided > 0) w->units[a].undecided--;
  }
  w->advance();
  return ITSX_OK;
}

int itsx_twriter_close(itsx_twriter *w, int64_t *n_written, int64_t *total_len)
{
  if (!w) return ITSX_OK;
  int rc = ITSX_OK;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    if (!w->text_done) { g_trim_error = "itsx_twriter_close: the text is not complete"; rc = ITSX_E_ARG; }
    else {
      // everything that can run does; then every unit must be done
      w->cv_idle.wait(lk, [&] { return w->jobs.empty() && w->busy == 0; });
      if (w->malformed) { g_trim_error = "malformed FASTQ record"; rc = ITSX_E_FORMAT; }
      else if (w->records_cut > (int64_t)w->start.size() || w->counted_prefix < w->units.size()) { g_trim_error = "more records in the file than coordinates"; rc = ITSX_E_ARG; }
      else if (w->next_write < w->units.size()) { g_trim_error = "itsx_twriter_close: records whose coordinates were never decided"; rc = ITSX_E_ARG; }
    }
    w->quit = true;
  }
  w->cv_work.notify_all();
  for (auto &t : w->workers) t.join();
  if (rc == ITSX_OK && !w->wrote_any && w->kind != 0) {          // an empty file is still one valid member / frame
    itsx_io::PieceCompressor pc(w->kind);
    std::string c;
    if (!pc.run(std::string(), c) || !itsx_twriter::pwrite_all(w->fd, c.data(), c.size(), w->file_off, w->seekable)) w->failed = true;
  }
  if (close(w->fd) != 0) w->failed = true;
  if (rc == ITSX_OK && w->failed) { g_trim_error = "compressing or writing the output failed"; rc = ITSX_E_IO; }
  if (n_written) *n_written = w->nw;
  if (total_len) *total_len = w->tot;
  delete w;
  return rc;
}

// the merged reads' labels -> their index, without a string per label: open addressing over a 64-bit hash, the label bytes compared
// on a hit (a std::unordered_map<std::string, ...> of 10 M labels was a second of allocations on one thread).  Built by a pool of
// threads: a slot is claimed with a compare-and-swap, a slot's LABEL never changes once it is set, and of equal labels the smallest
// index stays (the first of them wins, as emplace() kept it) -- lookups do not depend on the order of the insertions.
struct NameIndex {
  const char *names = nullptr; const int64_t *offs = nullptr; int64_t n = 0; std::unique_ptr<std::atomic<int64_t>[]> slot; uint64_t mask = 0;
  static uint64_t hash(const char *p, size_t len)
  {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < len; i++) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
    return h ^ (h >> 29);
  }
  bool same(int64_t j, const char *p, size_t len) const { return (size_t)(offs[j + 1] - offs[j]) == len && memcmp(names + offs[j], p, len) == 0; }
  void insert(int64_t i)
  {
    const char *p = names + offs[i]; const size_t len = (size_t)(offs[i + 1] - offs[i]);
    uint64_t h = hash(p, len) & mask;
    for (;;) {
      int64_t j = slot[h].load(std::memory_order_relaxed);
      if (j < 0 && slot[h].compare_exchange_strong(j, i, std::memory_order_relaxed)) return;
      // j = who holds the slot
      if (same(j, p, len)) { while (i < j && !slot[h].compare_exchange_weak(j, i, std::memory_order_relaxed)) {} return; }
      h = (h + 1) & mask;
    }
  }
  void build(const char *nm, const int64_t *of, int64_t count, int threads)
  {
    names = nm; offs = of; n = count;
    size_t cap = 16; while (cap < (size_t)count * 2) cap <<= 1;
    slot.reset(new std::atomic<int64_t>[cap]); mask = cap - 1;
    const int T = (count >= (1 << 16)) ? std::max(1, threads) : 1;
    auto part = [&](int t) {
      for (size_t k = cap * (size_t)t / (size_t)T, e = cap * (size_t)(t + 1) / (size_t)T; k < e; k++) slot[k].store(-1, std::memory_order_relaxed);
    };
    auto fill = [&](int t) { for (int64_t i = count * t / T, e = count * (t + 1) / T; i < e; i++) insert(i); };
    if (T == 1) { part(0); fill(0); return; }
    for (int phase = 0; phase < 2; phase++) {
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([&, t] { if (phase == 0) part(t); else fill(t); });
      for (auto &x : th) x.join();
    }
  }
  int64_t find(const char *p, size_t len) const
  {
    uint64_t h = hash(p, len) & mask;
    for (;;) {
      const int64_t j = slot[h].load(std::memory_order_relaxed);
      if (j < 0) return -1;
      if (same(j, p, len)) return j;
      h = (h + 1) & mask;
    }
  }
};

int itsx_write_trimmed_paired(const char *r1_path, const char *r2_path, const char *out1_path, const char *out2_path,
                              int compression, int trim_ccs, const char *names, const int64_t *name_offsets, int64_t n_names,
                              const int32_t *start, const int32_t *stop, const int32_t *tlen, int64_t *n_written)
{
  if (!r1_path || !r2_path || !out1_path || !out2_path || !names || !name_offsets || !start || !stop || !tlen) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  if (compression < 0 || compression > 2) { g_trim_error = "compression must be 0 (plain), 1 (gzip) or 2 (zstd)"; return ITSX_E_ARG; }
  const int T = itsx_io::io_threads();
  NameIndex idx;
  Records in1, in2;
  {   // the two inputs are inflated side by side (each by its own pool of block decoders) while the labels are indexed
    std::string e1, e2;
    auto open = [](Records &r, const char *path, std::string &err) {
      r.text = itsx_io::read_text(path, err, true);
      if (r.text) { r.s = r.text->data(); r.end = r.s + r.text->size(); }
    };
    std::thread t1([&] { open(in1, r1_path, e1); });
    std::thread t2([&] { open(in2, r2_path, e2); });
    idx.build(names, name_offsets, n_names, T);
    t1.join(); t2.join();
    if (!in1.text || !in2.text) { g_trim_error = !in1.text ? e1 : e2; return ITSX_E_IO; }
  }
  { itsx_io::PieceCompressor probe(compression); if (!probe.ok()) { g_trim_error = "zstd output requested but libzstd.so.1 could not be loaded"; return ITSX_E_IO; } }
  const bool ccs = trim_ccs != 0;
  // one pair of records -> its two trimmed records (appended to the two buffers), as the serial walk below emits them
  auto one_pair = [&](const Rec &a, const Rec &b, std::string &b1, std::string &b2) -> bool {
    size_t e0 = 1; while (e0 < a.title.n && a.title.p[e0] != ' ' && a.title.p[e0] != '\t') e0++;
    const int64_t k = idx.find(a.title.p + 1, e0 - 1);
    if (k < 0) return false;
    const int64_t s = start[k], e = stop[k], t = tlen[k];
    if (s < 0 || e < 0 || !(s < e)) return false;
    const int64_t r2start = t - e, r2end = t - s;
    static const char *fwd = "GACAGGTACAAGAAGGA", *rev = "TTAACCCAGTCTCCAGT";
    auto put = [&](std::string &out, const Rec &r, int64_t lo, int64_t hi) {
      out.append(r.title.p, r.title.n); out += '\n';
      if (ccs) out += fwd;
      out.append(r.seq.p + lo, (size_t)(hi - lo));
      if (ccs) out += rev;
      out += "\n+\n";
      if (ccs) out.append(17, '~');
      out.append(r.qual.p + lo, (size_t)(hi - lo));
      if (ccs) out.append(17, '~');
      out += '\n';
    };
    int64_t lo, hi;
    py_slice((int64_t)a.seq.size(), s, e, e > t, lo, hi);
    put(b1, a, lo, hi);
    py_slice((int64_t)b.seq.size(), r2start, r2end, r2end > t, lo, hi);
    put(b2, b, lo, hi);
    return true;
  };
  // Large inputs: R1 is cut into ranges at record starts, a counting pass gives every range its first record's number, R2 is cut at
  // the SAME record numbers (its ranges' own counts say in which range a number lies; the worker walks to it).  A pool of threads
  // takes the ranges in order: a thread slices its range AND compresses the two pieces (each an independent gzip member / zstd
  // frame), the calling thread writes the pieces in range order -- the records of the serial walk below, in its order.  (Round 4
  // sliced T ranges at a time and then fed the two block writers from ONE thread: 8 GB of text copied block by block between the
  // slicers and the compressors' 64 threads, and 10 M labels indexed by one thread before anything started: 3 s of a 6-s writer.)
  const size_t size1 = (size_t)(in1.end - in1.s), size2 = (size_t)(in2.end - in2.s);
  const size_t min_par = sw_get("ITSX_WRITE_MIN_MB") ? (size_t)atoll(sw_get("ITSX_WRITE_MIN_MB")) << 20 : (size_t)32 << 20;
  if (T > 1 && size1 >= min_par && size2 > 0) {
    const size_t range = sw_get("ITSX_WRITE_UNIT_KB") ? (size_t)atoll(sw_get("ITSX_WRITE_UNIT_KB")) << 10 : (size_t)8 << 20;
    auto cuts_of = [&](const char *t0, const char *tend, std::vector<const char *> &cut) {
      const size_t size = (size_t)(tend - t0);
      cut.assign(1, t0);
      for (size_t at = range; at < size; at += range) {
        const char *p = (const char *)memchr(t0 + at, '\n', size - at);
        while (p && p + 1 < tend && !fastq_record_start(t0, tend, p + 1)) p = (const char *)memchr(p + 1, '\n', (size_t)(tend - (p + 1)));
        if (!p || p + 1 >= tend) break;
        if (p + 1 > cut.back()) cut.push_back(p + 1);
      }
      cut.push_back(tend);
    };
    std::vector<const char *> c1, c2;
    cuts_of(in1.s, in1.end, c1); cuts_of(in2.s, in2.end, c2);
    const size_t K1 = c1.size() - 1, K2 = c2.size() - 1;
    std::vector<int64_t> f1(K1 + 1, 0), f2(K2 + 1, 0);
    std::atomic<int> bad{0};
    {
      std::atomic<size_t> next{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&] {
          for (size_t k = next.fetch_add(1); k < K1 + K2; k = next.fetch_add(1)) {
            const bool first = k < K1; const size_t q = first ? k : k - K1;
            Records r; r.s = (first ? c1 : c2)[q]; r.end = (first ? c1 : c2)[q + 1];
            Rec rec; int64_t n = 0; int rc;
            while ((rc = r.next(rec)) == 1) n++;
            if (rc < 0) bad = 1;
            (first ? f1 : f2)[q + 1] = n;
          }
        });
      for (auto &x : th) x.join();
    }
    FILE *fo1 = nullptr, *fo2 = nullptr;
    if (!bad) {
      fo1 = fopen(out1_path, "wb"); fo2 = fo1 ? fopen(out2_path, "wb") : nullptr;
      if (!fo1 || !fo2) { if (fo1) fclose(fo1); g_trim_error = std::string("cannot write ") + (fo1 ? out2_path : out1_path); return ITSX_E_IO; }
      setvbuf(fo1, nullptr, _IOFBF, 1 << 20); setvbuf(fo2, nullptr, _IOFBF, 1 << 20);
      for (size_t k = 0; k < K1; k++) f1[k + 1] += f1[k];
      for (size_t k = 0; k < K2; k++) f2[k + 1] += f2[k];
      const int64_t npairs = std::min(f1[K1], f2[K2]);            // zip(): stops at the shorter file
      struct Piece { std::string z1, z2; int64_t n = 0; bool ready = false; };
      std::vector<Piece> pieces(K1);
      std::mutex mu; std::condition_variable cv_ready, cv_room;
      size_t written = 0;                                          // ranges the calling thread has written
      const size_t window = (size_t)T * 3;                         // a thread never runs further ahead of the writer than this
      std::atomic<size_t> next{0};
      std::atomic<int> io_failed{0};
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&] {
          itsx_io::PieceCompressor pc(compression);
          std::string b1, b2;
          for (size_t k = next.fetch_add(1); k < K1; k = next.fetch_add(1)) {
            { std::unique_lock<std::mutex> lk(mu); cv_room.wait(lk, [&] { return k < written + window; }); }
            Piece &pz = pieces[k];
            const int64_t first = f1[k], last = std::min(f1[k + 1], npairs);
            if (!bad && first < last) {
              Records ra; ra.s = c1[k]; ra.end = c1[k + 1];
              // R2 at record `first`: the range that holds it, then a walk
              const size_t q = (size_t)(std::upper_bound(f2.begin(), f2.end(), first) - f2.begin()) - 1;
              Records rb; rb.s = c2[std::min(q, K2 - 1)]; rb.end = in2.end;
              Rec a, b;
              bool ok = q < K2;
              for (int64_t skip = ok ? first - f2[q] : 0; ok && skip > 0; skip--) ok = rb.next(b) == 1;
              b1.clear(); b2.clear();
              b1.reserve((size_t)(c1[k + 1] - c1[k]) + 4096); b2.reserve(b1.capacity());
              for (int64_t i = first; ok && i < last; i++) {
                ok = ra.next(a) == 1 && rb.next(b) == 1;
                if (ok && one_pair(a, b, b1, b2)) pz.n++;
              }
              if (!ok) bad = 1;
              else if (pz.n > 0 && (!pc.run(b1, pz.z1) || !pc.run(b2, pz.z2))) io_failed = 1;
            }
            { std::lock_guard<std::mutex> lk(mu); pz.ready = true; }
            cv_ready.notify_all();
          }
        });
      int64_t nw = 0; bool any = false;
      for (size_t k = 0; k < K1; k++) {
        Piece &pz = pieces[k];
        { std::unique_lock<std::mutex> lk(mu); cv_ready.wait(lk, [&] { return pz.ready; }); }
        if (!bad && !io_failed && pz.n > 0) {
          if (fwrite(pz.z1.data(), 1, pz.z1.size(), fo1) != pz.z1.size() || fwrite(pz.z2.data(), 1, pz.z2.size(), fo2) != pz.z2.size()) io_failed = 1;
          nw += pz.n; any = true;
        }
        std::string().swap(pz.z1); std::string().swap(pz.z2);
        { std::lock_guard<std::mutex> lk(mu); written = k + 1; }
        cv_room.notify_all();
      }
      for (auto &x : th) x.join();
      if (!bad && !io_failed && !any && compression != 0) {       // an empty file is still one valid member / frame
        itsx_io::PieceCompressor pc(compression); std::string z;
        if (!pc.run(std::string(), z) || fwrite(z.data(), 1, z.size(), fo1) != z.size() || fwrite(z.data(), 1, z.size(), fo2) != z.size()) io_failed = 1;
      }
      if (fflush(fo1) != 0 || ferror(fo1) || fflush(fo2) != 0 || ferror(fo2)) io_failed = 1;
      if (fclose(fo1) != 0) io_failed = 1;
      if (fclose(fo2) != 0) io_failed = 1;
      if (!bad) {
        if (io_failed) { g_trim_error = "compressing or writing the output failed"; return ITSX_E_IO; }
        if (n_written) *n_written = nw;
        return ITSX_OK;
      }
    }
    // a malformed record somewhere: the serial walk below names it (the output files are started again)
  }
  Writer o1, o2;
  if (!o1.open(out1_path, compression) || !o2.open(out2_path, compression)) return ITSX_E_IO;
  Rec a, b; int64_t nw = 0, k = 0; int ra, rb;
  for (;;) {
    ra = in1.next(a); rb = in2.next(b);
    if (ra != 1 || rb != 1) break;            // zip(): stops at the shorter file
    k++;
    if (one_pair(a, b, o1.buf, o2.buf)) nw++;
    if (o1.buf.size() >= (1u << 20) - 4096) o1.flush();
    if (o2.buf.size() >= (1u << 20) - 4096) o2.flush();
  }
  if (ra < 0 || rb < 0) { g_trim_error = "malformed FASTQ record near pair " + std::to_string(k); return ITSX_E_FORMAT; }
  if (!o1.close() || !o2.close()) return ITSX_E_IO;
  if (n_written) *n_written = nw;
  return ITSX_OK;
}

// f4: write the reads vsearch --orient keeps (SeqSample.py:48-91): forward ones as they are, reverse ones
// reverse-complemented (IUPAC complement, qualities reversed), undetermined ones dropped
int itsx_write_oriented_fastq(const char *seq_path, const char *out_path, const int8_t *strand, int64_t n_records, int64_t *n_written)
{
  if (!seq_path || !out_path || !strand) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  Records in; Writer out;
  if (!in.open(seq_path)) return ITSX_E_IO;
  if (!out.open(out_path, itsx_io::PLAIN, true)) return ITSX_E_IO;      // oriented.fq is read back by the loader and the trimmer
  static char comp[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) comp[i] = (char)i;
    const char *a = "ACGTURYMKSWHBVDNacgturymkswhbvdn", *b = "TGCAAYRKMSWDVBHNtgcaayrkmswdvbhn";
    for (int i = 0; a[i]; i++) comp[(unsigned char)a[i]] = b[i];
    init = true;
  }
  Rec rec; int64_t i = 0, nw = 0; int rc;
  std::string t, q;
  while ((rc = in.next(rec)) == 1) {
    if (i >= n_records) { g_trim_error = "more records in the file than orientations"; return ITSX_E_ARG; }
    const int s = strand[i++];
    if (s == 0) continue;
    const size_t L = rec.seq.size();
    if (s < 0) {
      t.resize(L); q.resize(L);
      for (size_t k = 0; k < L; k++) { t[k] = comp[(unsigned char)rec.seq.p[L - 1 - k]]; q[k] = rec.qual.p[L - 1 - k]; }
      emit(out, rec.title, t.data(), q.data(), 0, (int64_t)L, false, nullptr);
    } else emit(out, rec.title, rec.seq.p, rec.qual.p, 0, (int64_t)L, false, nullptr);
    nw++;
  }
  if (rc < 0) { g_trim_error = "malformed FASTQ record " + std::to_string(i); return ITSX_E_FORMAT; }
  if (!out.close()) return ITSX_E_IO;
  if (n_written) *n_written = nw;
  return ITSX_OK;
}

int itsx_io_read(const char *path, char **text, int64_t *len)
{
  if (!path || !text || !len) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  std::string err;
  const auto tp = itsx_io::read_text(path, err, false);
  if (!tp) { g_trim_error = err; return ITSX_E_IO; }
  char *buf = (char *)malloc(tp->size() + 1);
  if (!buf) { g_trim_error = "out of memory"; return ITSX_E_IO; }
  memcpy(buf, tp->data(), tp->size()); buf[tp->size()] = 0;
  *text = buf; *len = (int64_t)tp->size();
  return ITSX_OK;
}
void itsx_io_free(char *text) { free(text); }

// identifiers of a FASTQ file's records (title up to the first blank, without '@'), through the SAME record parser the
// trimming writers use, so that coordinates looked up by these names line up with the records that are written
int itsx_fastq_ids(const char *path, char **names, int64_t **offsets, int64_t *n_records)
{
  if (!path || !names || !offsets || !n_records) return ITSX_E_ARG;
  Records in;
  if (!in.open(path)) { g_trim_error = std::string("cannot read ") + path; return ITSX_E_IO; }
  std::string blob; std::vector<int64_t> off(1, 0);
  Rec rec;
  for (;;) {
    const int r = in.next(rec);
    if (r == 0) break;
    if (r < 0) { g_trim_error = std::string("malformed FASTQ record ") + std::to_string(off.size()) + " in " + path; return ITSX_E_FORMAT; }
    const char *b = rec.title.p + 1, *e = rec.title.p + rec.title.n, *q = b;
    while (q < e && *q != ' ' && *q != '\t') q++;
    blob.append(b, q);
    off.push_back((int64_t)blob.size());
  }
  char *nb = (char *)malloc(blob.size() + 1);
  int64_t *ob = (int64_t *)malloc(off.size() * sizeof(int64_t));
  if (!nb || !ob) { free(nb); free(ob); g_trim_error = "out of memory"; return ITSX_E_NOMEM; }
  memcpy(nb, blob.data(), blob.size()); nb[blob.size()] = 0;
  memcpy(ob, off.data(), off.size() * sizeof(int64_t));
  *names = nb; *offsets = ob; *n_records = (int64_t)off.size() - 1;
  return ITSX_OK;
}
int itsx_io_codecs(void) { return itsx_io::codec_flags(); }
void itsx_io_cache_clear(void) { itsx_io::cache_clear(); }
int64_t itsx_io_parallel_inflates(void) { return itsx_io::parallel_inflates(); }

}  // extern "C"
