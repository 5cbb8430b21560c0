// shard_host.cpp -- one file's records cut into pieces for the workers of a multi-GPU run (itsxpress_amd/multi.py).
//
// The reference hands the whole FASTQ to ONE vsearch process (itsxpress/SeqSample.py:93-131, 266-365).  With ITSXPRESS_GPUS=N every
// worker used to inflate and parse the whole file to keep its Nth (advisor, round 4: N times the host memory and CPU of one load);
// now the driver inflates it ONCE (the block-parallel inflater, the process-wide text cache), cuts the text at record starts near
// equal byte counts -- or, for a mate file, at the same record counts as its partner -- and writes the pieces as plain files
// (under /dev/shm: memory) that the workers load like any file.  Host-only, no GPU call, no arithmetic of the path.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include "../../include/itsx_hip.h"
#include "fastq_io.h"

namespace {
std::string g_shard_error;

size_t count_nl(const char *t, size_t a, size_t b)
{
  size_t c = 0;
  const char *p = t + a, *e = t + b;
  while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; c++; p = q + 1; }
  return c;
}
// FASTA: a record starts at a line that begins with '>'
size_t fasta_start_from(const char *t, size_t n, size_t from)
{
  size_t q = from;
  if (q > 0) { const char *nl = (const char *)memchr(t + q - 1, '\n', n - (q - 1)); if (!nl) return n; q = (size_t)(nl - t) + 1; }
  while (q < n) {
    if (t[q] == '>') return q;
    const char *nl = (const char *)memchr(t + q, '\n', n - q);
    if (!nl) return n;
    q = (size_t)(nl - t) + 1;
  }
  return n;
}
size_t count_gt(const char *t, size_t a, size_t b)      // lines of t[a, b) that start with '>' (a is a line start)
{
  size_t c = 0, q = a;
  while (q < b) {
    if (t[q] == '>') c++;
    const char *nl = (const char *)memchr(t + q, '\n', b - q);
    if (!nl) break;
    q = (size_t)(nl - t) + 1;
  }
  return c;
}
template <class F> void on_pool(int T, int jobs, F fn)
{
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++) th.emplace_back([&, k] { for (int j = k; j < jobs; j += T) fn(j); });
  for (auto &x : th) x.join();
}
}  // namespace

extern "C" {

const char *itsx_shard_last_error(void) { return g_shard_error.c_str(); }

// Round 6: one piece of a text that is still being inflated (a slice of itsx_stream_next) into a new file, by the I/O pool -- the multi-GPU
// driver hands a worker its piece while the parent's inflater is busy with the rest (itsxpress_amd/multi.py: _load_streamed).  write()
// into a tmpfs file allocates its pages in the kernel: 9 GB through a SHARED MAPPING of such a file took the inflater 4.7 s instead of 1.5
// (2.3 M page faults of 4 KB), which is why the pieces are files and not a mapping.
int itsx_write_range(const char *path, const char *text, int64_t nbytes)
{
  if (!path || (!text && nbytes > 0) || nbytes < 0) { g_shard_error = "itsx_write_range: missing argument"; return ITSX_E_ARG; }
  const int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0600);
  if (fd < 0) { g_shard_error = std::string("cannot write ") + path; return ITSX_E_IO; }
  const size_t n = (size_t)nbytes;
  if (n > 0 && ftruncate(fd, (off_t)n) != 0) { close(fd); unlink(path); g_shard_error = std::string("cannot write ") + path; return ITSX_E_IO; }
  const size_t BS = (size_t)16 << 20;
  const int nb = (int)((n + BS - 1) / BS);
  const int T = std::max(1, std::min(itsx_io::io_threads(), 8));
  std::atomic<int> bad{0};
  on_pool(T, nb, [&](int b) {
    size_t at = (size_t)b * BS; const size_t e = std::min(n, at + BS);
    while (at < e && !bad) {
      const ssize_t w = pwrite(fd, text + at, e - at, (off_t)at);
      if (w <= 0) { bad = 1; break; }
      at += (size_t)w;
    }
  });
  if (close(fd) != 0) bad = 1;
  if (bad) { unlink(path); g_shard_error = std::string("cannot write ") + path + " (no space left?)"; return ITSX_E_IO; }
  return ITSX_OK;
}

int itsx_shard_text(const char *path, int32_t n_parts, const int64_t *match_records, const char *out_prefix, int64_t *records, int64_t *bytes)
{
  if (!path || !out_prefix || n_parts < 1 || !records) { g_shard_error = "itsx_shard_text: missing argument"; return ITSX_E_ARG; }
  std::string err;
  const auto tp = itsx_io::read_text(path, err, true);
  if (!tp) { g_shard_error = err; return ITSX_E_IO; }
  const char *t = tp->data();
  const size_t n = tp->size();
  const bool fastq = n > 0 && t[0] == '@';
  if (n > 0 && !fastq && t[0] != '>') { g_shard_error = std::string("neither FASTA nor FASTQ: ") + path; return ITSX_E_FORMAT; }
  const int T = std::max(1, itsx_io::io_threads());
  std::vector<size_t> cut((size_t)n_parts + 1, n);
  cut[0] = 0;
  if (!match_records) {
    for (int p = 1; p < n_parts; p++) {
      const size_t target = (size_t)((double)n * (double)p / (double)n_parts);
      size_t c = fastq ? itsx_io::fastq_record_start(t, n, target) : fasta_start_from(t, n, target);
      cut[(size_t)p] = std::max(c, cut[(size_t)p - 1]);
    }
  } else {
    // the mate file: piece p ends after sum(match_records[0..p]) records.  Lines are counted block by block on the pool, then each
    // boundary is located inside its block (FASTQ: a record is four lines -- the parser's own assumption; FASTA: title lines)
    const size_t BS = (size_t)8 << 20;
    const int nb = (int)((n + BS - 1) / BS);
    std::vector<size_t> cnt((size_t)nb + 1, 0);
    // (block b counts the units that START in it: FASTQ lines by their terminating '\n' -- line k starts after the k-th newline)
    on_pool(T, nb, [&](int b) {
      const size_t a = (size_t)b * BS, e = std::min(n, a + BS);
      if (fastq) cnt[(size_t)b + 1] = count_nl(t, a, e);
      else { size_t q = a; if (q > 0) { q = fasta_start_from(t, n, a); } cnt[(size_t)b + 1] = q < e ? count_gt(t, q, e) : 0; }
    });
    for (int b = 0; b < nb; b++) cnt[(size_t)b + 1] += cnt[(size_t)b];
    int64_t acc = 0;
    for (int p = 1; p < n_parts; p++) {
      acc += match_records[p - 1];
      if (fastq) {
        // the record with index acc starts after newline number 4 * acc (1-based); 0 newlines: the start of the text
        const size_t want = (size_t)acc * 4;
        if (want == 0) { cut[(size_t)p] = 0; continue; }
        if (want > cnt[(size_t)nb]) { cut[(size_t)p] = n; continue; }
        int b = (int)(std::lower_bound(cnt.begin(), cnt.end(), want) - cnt.begin()) - 1;      // cnt[b] < want <= cnt[b + 1]
        size_t seen = cnt[(size_t)b], q = (size_t)b * BS;
        const size_t e = std::min(n, q + BS);
        while (seen < want && q < e) { const char *nl = (const char *)memchr(t + q, '\n', e - q); if (!nl) { q = e; break; } seen++; q = (size_t)(nl - t) + 1; }
        cut[(size_t)p] = q;
      } else {
        const size_t want = (size_t)acc;                    // title lines before the cut
        if (want == 0) { cut[(size_t)p] = 0; continue; }
        if (want >= cnt[(size_t)nb]) { cut[(size_t)p] = n; continue; }
        int b = (int)(std::upper_bound(cnt.begin(), cnt.end(), want) - cnt.begin()) - 1;      // cnt[b] <= want < cnt[b + 1]
        size_t seen = cnt[(size_t)b], q = (size_t)b * BS;
        if (q > 0) q = fasta_start_from(t, n, q);
        while (q < n) {                                     // title number `want` (0-based) starts the piece
          if (t[q] == '>') { if (seen == want) break; seen++; }
          const char *nl = (const char *)memchr(t + q, '\n', n - q);
          if (!nl) { q = n; break; }
          q = (size_t)(nl - t) + 1;
        }
        cut[(size_t)p] = q;
      }
    }
  }
  for (int p = 1; p <= n_parts; p++) cut[(size_t)p] = std::max(cut[(size_t)p], cut[(size_t)p - 1]);
  // records per piece, and the pieces themselves
  std::vector<int> okv((size_t)n_parts, 1);
  on_pool(std::min(T, n_parts), n_parts, [&](int p) {
    const size_t a = cut[(size_t)p], e = cut[(size_t)p + 1];
    size_t rec;
    if (fastq) { size_t nl = count_nl(t, a, e); if (e > a && t[e - 1] != '\n') nl++; rec = nl / 4; }
    else rec = count_gt(t, a, e);
    records[p] = (int64_t)rec;
    if (bytes) bytes[p] = (int64_t)(e - a);
    const std::string out = std::string(out_prefix) + "." + std::to_string(p);
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) { okv[(size_t)p] = 0; return; }
    if (e > a && fwrite(t + a, 1, e - a, f) != e - a) okv[(size_t)p] = 0;
    if (fclose(f) != 0) okv[(size_t)p] = 0;
  });
  for (int p = 0; p < n_parts; p++) if (!okv[(size_t)p]) { g_shard_error = std::string("cannot write ") + out_prefix + "." + std::to_string(p); return ITSX_E_IO; }
  if (match_records) for (int p = 0; p < n_parts; p++) if (records[p] != match_records[p]) { g_shard_error = std::string(path) + " does not hold the records of its mate file (piece " + std::to_string(p) + ")"; return ITSX_E_FORMAT; }
  return ITSX_OK;
}

}  // extern "C"

// The owner's step of the cross-shard dereplication (itsxpress_amd/multi.py: owner_verdicts is its numpy statement and the test's
// reference): rows (key0, key1, global index of the first occurrence, forward-is-canonical flag, local unique number) from the workers
// `src`; per row out: global index and flag of its group's first occurrence -- the row with the smallest global index among those with
// the same (key0, key1) --, and worker / local unique number of the holder that scores the sequence: the (key1 mod count)-th, in the
// order of their global indices, of the holders whose flag equals the first occurrence's.  numpy's three-key sort of 3 M rows took a
// second of every worker's exchange; here the rows go into 256 buckets by the key's top byte and a pool sorts the buckets.
extern "C" int itsx_owner_verdicts(const int64_t *recv, const int64_t *src, int64_t m, int64_t *out)
{
  if (m < 0 || (m > 0 && (!recv || !src || !out))) { g_shard_error = "itsx_owner_verdicts: bad argument"; return ITSX_E_ARG; }
  if (m == 0) return ITSX_OK;
  constexpr int B = 256;
  const int T = m >= (1 << 16) ? std::max(1, std::min(itsx_io::io_threads(), 32)) : 1;
  auto bucket = [&](int64_t i) { return (int)((uint64_t)recv[i * 5] >> 56); };
  std::vector<int64_t> idx((size_t)m);
  std::vector<std::vector<int64_t>> cnt((size_t)T, std::vector<int64_t>(B, 0));
  auto on_pool = [&](auto fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] { fn(t); });
    for (auto &x : th) x.join();
  };
  on_pool([&](int t) { for (int64_t i = m * t / T, e = m * (t + 1) / T; i < e; i++) cnt[(size_t)t][(size_t)bucket(i)]++; });
  std::vector<int64_t> start(B + 1, 0);
  for (int b = 0; b < B; b++) { int64_t c = 0; for (int t = 0; t < T; t++) c += cnt[(size_t)t][(size_t)b]; start[(size_t)b + 1] = start[(size_t)b] + c; }
  {   // every thread's own write positions inside a bucket, in thread (= row) order
    std::vector<std::vector<int64_t>> at((size_t)T, std::vector<int64_t>(B, 0));
    for (int b = 0; b < B; b++) { int64_t p = start[(size_t)b]; for (int t = 0; t < T; t++) { at[(size_t)t][(size_t)b] = p; p += cnt[(size_t)t][(size_t)b]; } }
    on_pool([&](int t) { for (int64_t i = m * t / T, e = m * (t + 1) / T; i < e; i++) idx[(size_t)at[(size_t)t][(size_t)bucket(i)]++] = i; });
  }
  std::atomic<int> next{0};
  on_pool([&](int) {
    for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
      int64_t *lo = idx.data() + start[(size_t)b], *hi = idx.data() + start[(size_t)b + 1];
      std::sort(lo, hi, [&](int64_t x, int64_t y) {
        const int64_t *a = recv + x * 5, *c = recv + y * 5;
        if (a[0] != c[0]) return a[0] < c[0];
        if (a[1] != c[1]) return a[1] < c[1];
        return a[2] < c[2];
      });
      for (int64_t *g = lo; g < hi;) {
        const int64_t *s = recv + *g * 5;
        int64_t *e = g + 1;
        while (e < hi && recv[*e * 5] == s[0] && recv[*e * 5 + 1] == s[1]) e++;
        const int64_t seed_gidx = s[2], seed_fwd = s[3];
        int64_t ncand = 0;
        for (int64_t *q = g; q < e; q++) ncand += recv[*q * 5 + 3] == seed_fwd;
        int64_t pick = s[1] % ncand;                      // (the first row is a candidate: ncand >= 1)
        if (pick < 0) pick += ncand;                      // numpy's remainder: the sign of the divisor
        int64_t sr = 0, su = 0, k = 0;
        for (int64_t *q = g; q < e; q++) if (recv[*q * 5 + 3] == seed_fwd) { if (k == pick) { sr = src[*q]; su = recv[*q * 5 + 4]; break; } k++; }
        for (int64_t *q = g; q < e; q++) { int64_t *o = out + *q * 4; o[0] = seed_gidx; o[1] = seed_fwd; o[2] = sr; o[3] = su; }
        g = e;
      }
    }
  });
  return ITSX_OK;
}
