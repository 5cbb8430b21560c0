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
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/itsx_hip.h"

namespace {

std::string g_trim_error;

struct LineReader {                       // plain or gzip, line by line
  gzFile f = nullptr;
  std::vector<char> buf;
  bool open(const char *path) { f = gzopen(path, "rb"); if (f) gzbuffer(f, 1 << 20); buf.resize(1 << 16); return f != nullptr; }
  bool line(std::string &out)
  {
    out.clear();
    for (;;) {
      if (!gzgets(f, buf.data(), (int)buf.size())) return !out.empty();
      out.append(buf.data());
      if (!out.empty() && out.back() == '\n') { out.pop_back(); if (!out.empty() && out.back() == '\r') out.pop_back(); return true; }
      if (gzeof(f)) return true;
    }
  }
  ~LineReader() { if (f) gzclose(f); }
};

struct Writer {
  FILE *fp = nullptr; gzFile gz = nullptr;
  bool open(const char *path, bool gzipped)
  {
    if (gzipped) { gz = gzopen(path, "wb"); return gz != nullptr; }
    fp = fopen(path, "w"); return fp != nullptr;
  }
  void put(const std::string &s) { if (gz) gzwrite(gz, s.data(), (unsigned)s.size()); else fwrite(s.data(), 1, s.size(), fp); }
  ~Writer() { if (gz) gzclose(gz); if (fp) fclose(fp); }
};

struct Rec { std::string title, seq, qual; };
// returns 1 on a record, 0 at end of file, -1 on malformed input
int next_record(LineReader &r, Rec &rec)
{
  std::string plus;
  do { if (!r.line(rec.title)) return 0; } while (rec.title.empty());
  if (rec.title[0] != '@') return -1;
  if (!r.line(rec.seq) || !r.line(plus) || !r.line(rec.qual)) return -1;
  if (plus.empty() || plus[0] != '+' || rec.qual.size() != rec.seq.size()) return -1;
  return 1;
}
std::string id_of(const std::string &title)
{
  size_t e = 1; while (e < title.size() && title[e] != ' ' && title[e] != '\t') e++;
  return title.substr(1, e - 1);
}
// Python's seq[a:b] for a sequence of length n (b_open: no upper bound)
void py_slice(int64_t n, int64_t a, int64_t b, bool b_open, int64_t &lo, int64_t &hi)
{
  if (a < 0) { a += n; if (a < 0) a = 0; } else if (a > n) a = n;
  if (b_open) b = n;
  else if (b < 0) { b += n; if (b < 0) b = 0; } else if (b > n) b = n;
  lo = a; hi = b < a ? a : b;
}
void emit(Writer &w, const Rec &r, int64_t lo, int64_t hi, bool ccs, int64_t *total)
{
  static const char *fwd = "GACAGGTACAAGAAGGA", *rev = "TTAACCCAGTCTCCAGT";
  std::string out;
  out.reserve(r.title.size() + 2 * (size_t)(hi - lo) + 80);
  out += r.title; out += '\n';
  if (ccs) out += fwd;
  out.append(r.seq, (size_t)lo, (size_t)(hi - lo));
  if (ccs) out += rev;
  out += "\n+\n";
  if (ccs) out.append(17, '~');
  out.append(r.qual, (size_t)lo, (size_t)(hi - lo));
  if (ccs) out.append(17, '~');
  out += '\n';
  w.put(out);
  if (total) *total += (hi - lo) + (ccs ? 34 : 0);
}

}  // namespace

extern "C" {

const char *itsx_trim_last_error(void) { return g_trim_error.c_str(); }

int itsx_write_trimmed_fastq(const char *seq_path, const char *out_path, int gzipped, int trim_ccs,
                             const int32_t *start, const int32_t *stop, int64_t n_records,
                             int64_t *n_written, int64_t *total_len)
{
  if (!seq_path || !out_path || !start || !stop) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  LineReader in; Writer out;
  if (!in.open(seq_path)) { g_trim_error = std::string("cannot read ") + seq_path; return ITSX_E_IO; }
  if (!out.open(out_path, gzipped != 0)) { g_trim_error = std::string("cannot write ") + out_path; return ITSX_E_IO; }
  Rec rec; int64_t i = 0, nw = 0, tot = 0; int rc;
  while ((rc = next_record(in, rec)) == 1) {
    if (i >= n_records) { g_trim_error = "more records in the file than coordinates"; return ITSX_E_ARG; }
    const int64_t a = start[i], b = stop[i];
    i++;
    if (a < 0 || b < 0 || !(a < b)) continue;
    int64_t lo, hi; py_slice((int64_t)rec.seq.size(), a, b, false, lo, hi);
    emit(out, rec, lo, hi, trim_ccs != 0, &tot);
    nw++;
  }
  if (rc < 0) { g_trim_error = "malformed FASTQ record " + std::to_string(i); return ITSX_E_FORMAT; }
  if (n_written) *n_written = nw;
  if (total_len) *total_len = tot;
  return ITSX_OK;
}

int itsx_write_trimmed_paired(const char *r1_path, const char *r2_path, const char *out1_path, const char *out2_path,
                              int gzipped, int trim_ccs, const char *names, const int64_t *name_offsets, int64_t n_names,
                              const int32_t *start, const int32_t *stop, const int32_t *tlen, int64_t *n_written)
{
  if (!r1_path || !r2_path || !out1_path || !out2_path || !names || !name_offsets || !start || !stop || !tlen) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  std::unordered_map<std::string, int64_t> idx;
  idx.reserve((size_t)n_names * 2);
  for (int64_t i = 0; i < n_names; i++) idx.emplace(std::string(names + name_offsets[i], (size_t)(name_offsets[i + 1] - name_offsets[i])), i);
  LineReader in1, in2; Writer o1, o2;
  if (!in1.open(r1_path)) { g_trim_error = std::string("cannot read ") + r1_path; return ITSX_E_IO; }
  if (!in2.open(r2_path)) { g_trim_error = std::string("cannot read ") + r2_path; return ITSX_E_IO; }
  if (!o1.open(out1_path, gzipped != 0) || !o2.open(out2_path, gzipped != 0)) { g_trim_error = "cannot write the output files"; return ITSX_E_IO; }
  Rec a, b; int64_t nw = 0, k = 0; int ra, rb;
  for (;;) {
    ra = next_record(in1, a); rb = next_record(in2, b);
    if (ra != 1 || rb != 1) break;            // zip(): stops at the shorter file
    k++;
    auto it = idx.find(id_of(a.title));
    if (it == idx.end()) continue;
    const int64_t s = start[it->second], e = stop[it->second], t = tlen[it->second];
    if (s < 0 || e < 0 || !(s < e)) continue;
    const int64_t r2start = t - e, r2end = t - s;
    int64_t lo, hi;
    py_slice((int64_t)a.seq.size(), s, e, e > t, lo, hi);
    emit(o1, a, lo, hi, trim_ccs != 0, nullptr);
    py_slice((int64_t)b.seq.size(), r2start, r2end, r2end > t, lo, hi);
    emit(o2, b, lo, hi, trim_ccs != 0, nullptr);
    nw++;
  }
  if (ra < 0 || rb < 0) { g_trim_error = "malformed FASTQ record near pair " + std::to_string(k); return ITSX_E_FORMAT; }
  if (n_written) *n_written = nw;
  return ITSX_OK;
}

// f4: write the reads vsearch --orient keeps (SeqSample.py:48-91): forward ones as they are, reverse ones
// reverse-complemented (IUPAC complement, qualities reversed), undetermined ones dropped
int itsx_write_oriented_fastq(const char *seq_path, const char *out_path, const int8_t *strand, int64_t n_records, int64_t *n_written)
{
  if (!seq_path || !out_path || !strand) { g_trim_error = "null argument"; return ITSX_E_ARG; }
  LineReader in; Writer out;
  if (!in.open(seq_path)) { g_trim_error = std::string("cannot read ") + seq_path; return ITSX_E_IO; }
  if (!out.open(out_path, false)) { g_trim_error = std::string("cannot write ") + out_path; return ITSX_E_IO; }
  static char comp[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) comp[i] = (char)i;
    const char *a = "ACGTURYMKSWHBVDNacgturymkswhbvdn", *b = "TGCAAYRKMSWDVBHNtgcaayrkmswdvbhn";
    for (int i = 0; a[i]; i++) comp[(unsigned char)a[i]] = b[i];
    init = true;
  }
  Rec rec; int64_t i = 0, nw = 0; int rc;
  while ((rc = next_record(in, rec)) == 1) {
    if (i >= n_records) { g_trim_error = "more records in the file than orientations"; return ITSX_E_ARG; }
    const int s = strand[i++];
    if (s == 0) continue;
    if (s < 0) {
      std::string q(rec.qual.rbegin(), rec.qual.rend()), t(rec.seq.size(), 'N');
      for (size_t k = 0; k < rec.seq.size(); k++) t[k] = comp[(unsigned char)rec.seq[rec.seq.size() - 1 - k]];
      rec.seq.swap(t); rec.qual.swap(q);
    }
    emit(out, rec, 0, (int64_t)rec.seq.size(), false, nullptr);
    nw++;
  }
  if (rc < 0) { g_trim_error = "malformed FASTQ record " + std::to_string(i); return ITSX_E_FORMAT; }
  if (n_written) *n_written = nw;
  return ITSX_OK;
}

}  // extern "C"
