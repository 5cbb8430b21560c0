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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>
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
  if (!out.open(out_path, compression)) return ITSX_E_IO;
  // Large inputs: the record walk itself (10 M records = 9 GB of text in, 4 GB out) took longer on one thread than the whole GPU
  // path.  The text is cut at record starts into ranges that a pool of threads slices independently -- a counting pass gives every
  // range the index of its first record --, and the ranges' outputs go to the block writer in order: the same bytes as the serial walk.
  const int T = itsx_io::io_threads();
  const size_t size = (size_t)(in.end - in.s);
  const size_t min_par = getenv("ITSX_WRITE_MIN_MB") ? (size_t)atoll(getenv("ITSX_WRITE_MIN_MB")) << 20 : (size_t)64 << 20;
  if (T > 1 && size >= min_par) {
    const size_t range = getenv("ITSX_WRITE_RANGE_KB") ? (size_t)atoll(getenv("ITSX_WRITE_RANGE_KB")) << 10 : (size_t)16 << 20;
    const char *t0 = in.s, *tend = in.end;
    std::vector<const char *> cut(1, t0);
    for (size_t at = range; at < size; at += range) {
      const char *p = (const char *)memchr(t0 + at, '\n', size - at);
      while (p && p + 1 < tend && !fastq_record_start(t0, tend, p + 1)) p = (const char *)memchr(p + 1, '\n', (size_t)(tend - (p + 1)));
      if (!p || p + 1 >= tend) break;
      if (p + 1 > cut.back()) cut.push_back(p + 1);
    }
    cut.push_back(tend);
    const size_t K = cut.size() - 1;
    std::vector<int64_t> first(K + 1, 0);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto pool = [&](auto fn) {
      std::vector<std::thread> th;
      next = 0;
      for (int t = 0; t < T; t++) th.emplace_back([&] { for (size_t k = next.fetch_add(1); k < K; k = next.fetch_add(1)) fn(k); });
      for (auto &x : th) x.join();
    };
    pool([&](size_t k) {                                    // pass 1: records per range
      Records r; r.s = cut[k]; r.end = cut[k + 1];
      Rec rec; int64_t n = 0; int rc;
      while ((rc = r.next(rec)) == 1) n++;
      if (rc < 0) bad = 1;
      first[k + 1] = n;
    });
    if (!bad) {
      for (size_t k = 0; k < K; k++) first[k + 1] += first[k];
      if (first[K] > n_records) { g_trim_error = "more records in the file than coordinates"; return ITSX_E_ARG; }
      int64_t nw = 0, tot = 0;
      const bool ccs = trim_ccs != 0;
      for (size_t k0 = 0; k0 < K && !bad; k0 += (size_t)T) {             // rounds of T ranges, written in order
        const size_t k1 = std::min(K, k0 + (size_t)T);
        std::vector<Writer> part(k1 - k0);
        std::vector<int64_t> pn(k1 - k0, 0), pt(k1 - k0, 0);
        std::vector<std::thread> th;
        for (size_t k = k0; k < k1; k++)
          th.emplace_back([&, k] {
            Records r; r.s = cut[k]; r.end = cut[k + 1];
            Writer &w = part[k - k0];                         // (never opened: emit() only appends to its buffer, which is taken below)
            w.buf.reserve((size_t)(cut[k + 1] - cut[k]) / 2 + 4096);
            Rec rec; int64_t i = first[k]; int rc;
            std::string &o = w.buf;
            static const char *fwd = "GACAGGTACAAGAAGGA", *rev = "TTAACCCAGTCTCCAGT";
            while ((rc = r.next(rec)) == 1) {
              const int64_t a = start[i], b = stop[i];
              i++;
              if (a < 0 || b < 0 || !(a < b)) continue;
              int64_t lo, hi; py_slice((int64_t)rec.seq.size(), a, b, false, lo, hi);
              o.append(rec.title.p, rec.title.n); o += '\n';
              if (ccs) o += fwd;
              o.append(rec.seq.p + lo, (size_t)(hi - lo));
              if (ccs) o += rev;
              o += "\n+\n";
              if (ccs) o.append(17, '~');
              o.append(rec.qual.p + lo, (size_t)(hi - lo));
              if (ccs) o.append(17, '~');
              o += '\n';
              pn[k - k0]++; pt[k - k0] += (hi - lo) + (ccs ? 34 : 0);
            }
            if (rc < 0) bad = 1;
          });
        for (auto &x : th) x.join();
        for (size_t k = k0; k < k1 && !bad; k++) { out.w.put(part[k - k0].buf); nw += pn[k - k0]; tot += pt[k - k0]; }
      }
      if (!bad) {
        if (!out.close()) return ITSX_E_IO;
        if (n_written) *n_written = nw;
        if (total_len) *total_len = tot;
        return ITSX_OK;
      }
    }
    // a malformed record somewhere: the serial walk below names it (the output file is started again)
    { std::string e; out.w.close(e); }
    if (!out.open(out_path, compression)) return ITSX_E_IO;
  }
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

int itsx_write_trimmed_paired(const char *r1_path, const char *r2_path, const char *out1_path, const char *out2_path,
                              int compression, int trim_ccs, const char *names, const int64_t *name_offsets, int64_t n_names,
                              const int32_t *start, const int32_t *stop, const int32_t *tlen, int64_t *n_written)
{
  if (!r1_path || !r2_path || !out1_path || !out2_path || !names || !name_offsets || !start || !stop || !tlen) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  std::unordered_map<std::string, int64_t> idx;
  idx.reserve((size_t)n_names * 2);
  for (int64_t i = 0; i < n_names; i++) idx.emplace(std::string(names + name_offsets[i], (size_t)(name_offsets[i + 1] - name_offsets[i])), i);
  if (compression < 0 || compression > 2) { g_trim_error = "compression must be 0 (plain), 1 (gzip) or 2 (zstd)"; return ITSX_E_ARG; }
  Records in1, in2; Writer o1, o2;
  if (!in1.open(r1_path) || !in2.open(r2_path)) return ITSX_E_IO;
  if (!o1.open(out1_path, compression) || !o2.open(out2_path, compression)) return ITSX_E_IO;
  Rec a, b; int64_t nw = 0, k = 0; int ra, rb;
  for (;;) {
    ra = in1.next(a); rb = in2.next(b);
    if (ra != 1 || rb != 1) break;            // zip(): stops at the shorter file
    k++;
    auto it = idx.find(id_of(a.title));
    if (it == idx.end()) continue;
    const int64_t s = start[it->second], e = stop[it->second], t = tlen[it->second];
    if (s < 0 || e < 0 || !(s < e)) continue;
    const int64_t r2start = t - e, r2end = t - s;
    int64_t lo, hi;
    py_slice((int64_t)a.seq.size(), s, e, e > t, lo, hi);
    emit(o1, a.title, a.seq.p, a.qual.p, lo, hi, trim_ccs != 0, nullptr);
    py_slice((int64_t)b.seq.size(), r2start, r2end, r2end > t, lo, hi);
    emit(o2, b.title, b.seq.p, b.qual.p, lo, hi, trim_ccs != 0, nullptr);
    nw++;
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
