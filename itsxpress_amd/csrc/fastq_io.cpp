// fastq_io.cpp -- see fastq_io.h.  Host-only.
#include "fastq_io.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace itsx_io {

// ------------------------------------------------------------------ Text
namespace {
constexpr size_t TEXT_MAP_MIN = (size_t)1 << 20;        // mappings from here up
inline size_t page_up(size_t n) { return (n + 4095) & ~(size_t)4095; }
}
void Text::release()
{
  if (kind_ == 1) free(p_);
  else if (kind_ == 2 || kind_ == 5) munmap(p_, cap_);
  std::string().swap(own_);
  p_ = nullptr; n_ = cap_ = 0; kind_ = 0; pinned_ = false;
}
bool Text::reserve_file(const char *path, size_t cap)
{
  release();
  const size_t want = page_up(std::max<size_t>(cap, 4096));
  const int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
  if (fd < 0) return false;
  if (ftruncate(fd, (off_t)want) != 0) { close(fd); unlink(path); return false; }
  void *q = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0);
  close(fd);
  if (q == MAP_FAILED) { unlink(path); return false; }
  p_ = (char *)q; cap_ = want; n_ = 0; kind_ = 5;
  return true;
}
bool Text::reserve(size_t cap)
{
  if (cap <= cap_) return true;
  if (pinned_ || kind_ == 4 || kind_ == 5) return false;
  if (kind_ == 3) {                                     // an adopted string grows as a string does
    const size_t keep = n_;
    try { own_.resize(cap); } catch (...) { return false; }
    p_ = &own_[0]; cap_ = own_.size(); n_ = keep;
    return true;
  }
  if (cap < TEXT_MAP_MIN) {
    if (kind_ == 2) return true;                        // (cannot happen: a mapping is never below the threshold)
    char *q = (char *)realloc(kind_ == 1 ? p_ : nullptr, cap);
    if (!q) return false;
    p_ = q; cap_ = cap; kind_ = 1;
    return true;
  }
  const size_t want = page_up(cap);
  if (kind_ == 2) {
    void *q = mremap(p_, cap_, want, MREMAP_MAYMOVE);
    if (q != MAP_FAILED) { p_ = (char *)q; cap_ = want; return true; }
    // (a mapping that something has split into several areas cannot be remapped in one call: copy)
  }
  void *q = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (q == MAP_FAILED) return false;
  // huge pages under a text that is written once, front to back, by many threads: asked for over the WHOLE mapping, so that it
  // stays one area (advice on a part splits it, and a split mapping cannot grow by mremap)
  static const bool huge = !(sw_get("ITSX_HUGEPAGES") && atoi(sw_get("ITSX_HUGEPAGES")) == 0);
  if (huge) (void)madvise(q, want, MADV_HUGEPAGE);
  if (n_) memcpy(q, p_, n_);
  if (kind_ == 1) free(p_);
  else if (kind_ == 2) munmap(p_, cap_);
  p_ = (char *)q; cap_ = want; kind_ = 2;
  return true;
}
bool Text::resize(size_t n)
{
  if (n > cap_) {
    // geometric growth, so that a decoder that doubles on demand does not remap for every step
    size_t cap = std::max(n, cap_ + cap_ / 2);
    if (!reserve(cap) && !reserve(n)) return false;
  }
  n_ = n;
  return true;
}
void Text::shrink_to_fit()
{
  if (kind_ == 2) {
    const size_t want = std::max(page_up(n_), (size_t)4096);
    if (want < cap_) { munmap(p_ + want, cap_ - want); cap_ = want; }      // the front stays where it is
  } else if (kind_ == 1 && n_ && n_ < cap_) {
    char *q = (char *)realloc(p_, n_);
    if (q) { p_ = q; cap_ = n_; }
  }
}
void Text::swap(Text &o)
{
  std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); std::swap(kind_, o.kind_); std::swap(pinned_, o.pinned_);
  own_.swap(o.own_);
  if (kind_ == 3) p_ = own_.empty() ? nullptr : &own_[0];                  // (a short string lives inside the object)
  if (o.kind_ == 3) o.p_ = o.own_.empty() ? nullptr : &o.own_[0];
}
void Text::adopt(std::string &&s)
{
  release();
  own_ = std::move(s);
  p_ = own_.empty() ? nullptr : &own_[0]; n_ = cap_ = own_.size(); kind_ = 3;
}

namespace {

// ------------------------------------------------------------------ codecs bound at run time
struct LibDeflate {
  void *h = nullptr;
  void *(*alloc_c)(int) = nullptr;
  size_t (*gz_bound)(void *, size_t) = nullptr;
  size_t (*gz_c)(void *, const void *, size_t, void *, size_t) = nullptr;
  void (*free_c)(void *) = nullptr;
  void *(*alloc_d)() = nullptr;
  int (*gz_d)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
  void (*free_d)(void *) = nullptr;
  uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;       // optional (carry-less multiply: ~10x zlib's table walk)
  bool ok = false;
};
struct ZBuf { void *p; size_t size, pos; };          // ZSTD_inBuffer / ZSTD_outBuffer (same shape, src is const in the former)
struct LibZstd {
  void *h = nullptr;
  size_t (*bound)(size_t) = nullptr;
  size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  void *(*create_d)() = nullptr;
  size_t (*free_d)(void *) = nullptr;
  size_t (*decompress_stream)(void *, ZBuf *, ZBuf *) = nullptr;
  bool ok = false;
};
LibDeflate g_ld;
LibZstd g_zs;
std::once_flag g_codec_once;

template <class F> bool sym(void *h, const char *name, F &f) { f = reinterpret_cast<F>(dlsym(h, name)); return f != nullptr; }

void load_codecs()
{
  const char *off = sw_get("ITSX_IO_LIBDEFLATE");
  if (!(off && atoi(off) == 0)) {
    for (const char *n : {"libdeflate.so.0", "libdeflate.so"}) { g_ld.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_ld.h) break; }
    if (g_ld.h) {
      g_ld.ok = sym(g_ld.h, "libdeflate_alloc_compressor", g_ld.alloc_c) && sym(g_ld.h, "libdeflate_gzip_compress_bound", g_ld.gz_bound) &&
                sym(g_ld.h, "libdeflate_gzip_compress", g_ld.gz_c) && sym(g_ld.h, "libdeflate_free_compressor", g_ld.free_c) &&
                sym(g_ld.h, "libdeflate_alloc_decompressor", g_ld.alloc_d) && sym(g_ld.h, "libdeflate_gzip_decompress_ex", g_ld.gz_d) &&
                sym(g_ld.h, "libdeflate_free_decompressor", g_ld.free_d);
      (void)sym(g_ld.h, "libdeflate_crc32", g_ld.crc);
    }
  }
  for (const char *n : {"libzstd.so.1", "libzstd.so"}) { g_zs.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_zs.h) break; }
  if (g_zs.h) {
    g_zs.ok = sym(g_zs.h, "ZSTD_compressBound", g_zs.bound) && sym(g_zs.h, "ZSTD_compress", g_zs.compress) && sym(g_zs.h, "ZSTD_isError", g_zs.is_error) &&
              sym(g_zs.h, "ZSTD_createDStream", g_zs.create_d) && sym(g_zs.h, "ZSTD_freeDStream", g_zs.free_d) &&
              sym(g_zs.h, "ZSTD_decompressStream", g_zs.decompress_stream);
  }
}
void codecs() { std::call_once(g_codec_once, load_codecs); }

int env_int(const char *name, int dflt) { const char *e = itsx::sw_get(name); return e ? atoi(e) : dflt; }

// ------------------------------------------------------------------ decompression of a whole buffer
bool is_gzip(const Text &r, size_t pos = 0) { return r.size() >= pos + 2 && (unsigned char)r[pos] == 0x1f && (unsigned char)r[pos + 1] == 0x8b; }
bool is_zstd(const Text &r) { return r.size() >= 4 && (unsigned char)r[0] == 0x28 && (unsigned char)r[1] == 0xb5 && (unsigned char)r[2] == 0x2f && (unsigned char)r[3] == 0xfd; }

// The trailer's ISIZE (length mod 2^32 of the LAST member) is the whole length for the usual single-member file below
// 4 GB; anything implausible falls back to 4x the compressed size.  Only a first guess: both inflaters grow on demand.
size_t gzip_size_hint(const Text &raw)
{
  const size_t guess = std::max<size_t>(raw.size() * 4, 1 << 20);
  if (raw.size() < 18) return guess;
  const unsigned char *t = (const unsigned char *)raw.data() + raw.size() - 4;
  const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
  if (isize >= raw.size() / 2 && isize <= raw.size() * 40) return isize + 64;
  return guess;
}

bool gunzip_libdeflate(const Text &raw, Text &out, std::string &err)
{
  void *d = g_ld.alloc_d();
  if (!d) { err = "libdeflate: out of memory"; return false; }
  size_t pos = 0, opos = 0;
  bool ok = out.resize(gzip_size_hint(raw));
  if (!ok) err = "out of memory inflating";
  while (ok && pos < raw.size() && is_gzip(raw, pos)) {
    size_t ain = 0, aout = 0;
    const int rc = g_ld.gz_d(d, raw.data() + pos, raw.size() - pos, &out[0] + opos, out.size() - opos, &ain, &aout);
    if (rc == 3) { if (!out.resize(out.size() * 2)) { err = "out of memory inflating"; ok = false; break; } continue; }              // LIBDEFLATE_INSUFFICIENT_SPACE: retry this member
    if (rc != 0) { err = "corrupt gzip data"; ok = false; break; }
    pos += ain; opos += aout;
  }
  g_ld.free_d(d);
  out.resize(ok ? opos : 0);
  return ok;
}
bool gunzip_zlib(const Text &raw, Text &out, std::string &err)
{
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return false; }
  size_t pos = 0, opos = 0;
  bool ok = out.resize(gzip_size_hint(raw));
  if (!ok) err = "out of memory inflating";
  while (ok && pos < raw.size() && is_gzip(raw, pos)) {
    for (;;) {
      if (opos == out.size() && !out.resize(out.size() * 2)) { ok = false; break; }
      const size_t in_chunk = std::min<size_t>(raw.size() - pos, 1u << 30), out_chunk = std::min<size_t>(out.size() - opos, 1u << 30);
      zs.next_in = (Bytef *)(raw.data() + pos); zs.avail_in = (uInt)in_chunk;
      zs.next_out = (Bytef *)(&out[0] + opos); zs.avail_out = (uInt)out_chunk;
      const int rc = inflate(&zs, Z_NO_FLUSH);
      pos += in_chunk - zs.avail_in; opos += out_chunk - zs.avail_out;
      if (rc == Z_STREAM_END) break;
      if (rc == Z_OK) { if (pos >= raw.size() && zs.avail_out != 0) { ok = false; break; } continue; }   // input ended inside a member
      if (rc == Z_BUF_ERROR && zs.avail_out == 0) continue;                                              // needs room
      ok = false; break;
    }
    if (!ok) { err = "corrupt or truncated gzip data"; break; }
    inflateReset(&zs);
  }
  inflateEnd(&zs);
  out.resize(ok ? opos : 0);
  return ok;
}
bool unzstd(const Text &raw, Text &out, std::string &err)
{
  if (!g_zs.ok) { err = "zstd-compressed file but libzstd.so.1 could not be loaded"; return false; }
  void *ds = g_zs.create_d();
  if (!ds) { err = "zstd: out of memory"; return false; }
  if (!out.resize(std::max<size_t>(raw.size() * 4, 1 << 20))) { g_zs.free_d(ds); err = "out of memory"; return false; }
  ZBuf in{(void *)raw.data(), raw.size(), 0}, ob{&out[0], out.size(), 0};
  bool ok = true; size_t last = 0;
  while (in.pos < in.size) {
    if (ob.pos == ob.size) { if (!out.resize(out.size() * 2)) { err = "out of memory"; ok = false; break; } ob.p = &out[0]; ob.size = out.size(); }
    last = g_zs.decompress_stream(ds, &ob, &in);
    if (g_zs.is_error(last)) { err = "corrupt zstd data"; ok = false; break; }
  }
  // the input is consumed; a frame that still wants output has only been waiting for room
  while (ok && last != 0) {
    if (ob.pos == ob.size) { if (!out.resize(out.size() * 2)) { err = "out of memory"; ok = false; break; } ob.p = &out[0]; ob.size = out.size(); }
    const size_t before = ob.pos;
    last = g_zs.decompress_stream(ds, &ob, &in);
    if (g_zs.is_error(last)) { err = "corrupt zstd data"; ok = false; break; }
    if (last != 0 && ob.pos == before && ob.pos < ob.size) { err = "truncated zstd data"; ok = false; break; }
  }
  g_zs.free_d(ds);
  out.resize(ok ? ob.pos : 0);
  return ok;
}

// ------------------------------------------------------------------ the text cache
struct CacheEntry { std::string path; int64_t size, mtime_ns; std::shared_ptr<const Text> text; };
std::mutex g_cache_mu;
std::deque<CacheEntry> g_cache;          // most recent at the back

// ITSX_TEXT_CACHE_GB, or by default a quarter of the memory this process may still take (MemAvailable, cut by the cgroup's limit), at
// least 4 GB and at most 32: the writer of a 10 M-read sample inflated its 9 GB of text a second time because the cache held 4 GB
double cache_budget_bytes()
{
  if (const char *e = sw_get("ITSX_TEXT_CACHE_GB")) return atof(e) * (double)(1ull << 30);
  static const double deflt = [] {
    double avail = 0.0;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
      char line[256];
      while (fgets(line, sizeof(line), f)) { long long kb; if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) { avail = (double)kb * 1024.0; break; } }
      fclose(f);
    }
    for (const char *p : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"})
      if (FILE *f = fopen(p, "r")) {
        char buf[64] = {0};
        if (fgets(buf, sizeof(buf), f) && buf[0] >= '0' && buf[0] <= '9') { const double lim = atof(buf); if (lim > 0 && (avail <= 0 || lim < avail)) avail = lim; }
        fclose(f);
        break;
      }
    double gb = avail > 0 ? avail / 4.0 / (double)(1ull << 30) : 4.0;
    if (gb < 4.0) gb = 4.0;
    if (gb > 32.0) gb = 32.0;
    return gb;
  }();
  return deflt * (double)(1ull << 30);
}
bool stat_of(const char *path, int64_t &size, int64_t &mtime_ns)
{
  struct stat st;
  if (stat(path, &st) != 0) return false;
  size = (int64_t)st.st_size; mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  return true;
}

}  // namespace

void cache_clear() { std::lock_guard<std::mutex> g(g_cache_mu); g_cache.clear(); }

uint32_t crc32_fast(uint32_t crc, const void *p, size_t n)
{
  codecs();
  if (g_ld.crc) return g_ld.crc(crc, p, n);
  uLong c = (uLong)crc;
  const unsigned char *q = (const unsigned char *)p;
  while (n > 0) { const size_t m = std::min<size_t>(n, (size_t)1 << 30); c = crc32(c, q, (uInt)m); q += m; n -= m; }
  return (uint32_t)c;
}

static void cache_insert(const char *path, int64_t fsize, int64_t mtime, std::shared_ptr<const Text> text, double budget)
{
  std::lock_guard<std::mutex> g(g_cache_mu);
  for (auto it = g_cache.begin(); it != g_cache.end(); ++it) if (it->path == path) { g_cache.erase(it); break; }
  g_cache.push_back(CacheEntry{path, fsize, mtime, text});
  double total = 0;
  for (auto &e : g_cache) total += (double)e.text->size();
  while (total > budget && g_cache.size() > 1) { total -= (double)g_cache.front().text->size(); g_cache.pop_front(); }
}

void cache_put(const char *path, std::shared_ptr<const Text> text)
{
  const double budget = cache_budget_bytes();
  int64_t fsize = 0, mtime = 0;
  if (!text || budget <= 0 || (double)text->size() > budget || !stat_of(path, fsize, mtime)) return;
  if (fsize != (int64_t)text->size()) return;          // only a plain file whose bytes are exactly `text`
  cache_insert(path, fsize, mtime, text, budget);
}

static std::atomic<int64_t> g_parallel_inflates{0};
int64_t parallel_inflates() { return g_parallel_inflates.load(); }

int codec_flags() { codecs(); return (g_ld.ok ? 1 : 0) | (g_zs.ok ? 2 : 0); }

int io_threads()
{
  int n = env_int("ITSX_IO_THREADS", 0);
  if (n <= 0) { n = (int)std::thread::hardware_concurrency(); if (n <= 0) n = 4; n = std::min(n, 32); }
  return std::max(1, n);
}

// the file's bytes (+ 64 zero bytes behind them that are not part of size(): the inflaters may look a little past the end)
static bool read_raw(const char *path, int64_t fsize, Text &raw, std::string &err)
{
  // the file's bytes: large regular files are read by a pool of threads into an untouched mapping (one thread's fread into a
  // zero-filled string cost 0.9 s for the 3.6 GB of a 10 M-read sample)
  {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { err = std::string("cannot read ") + path; return false; }
    bool rerr = false;
    size_t got = 0;
    if (!raw.resize((size_t)fsize + 64)) { close(fd); err = std::string("out of memory reading ") + path; return false; }
    const int T = (fsize >= (int64_t)(64 << 20)) ? std::min(io_threads(), 16) : 1;
    if (T > 1) {
      std::vector<std::thread> th;
      std::atomic<bool> bad{false};
      const size_t per = (((size_t)fsize + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
          size_t at = std::min((size_t)fsize, per * (size_t)t);
          const size_t end = std::min((size_t)fsize, at + per);
          while (at < end) { const ssize_t n = pread(fd, raw.data() + at, end - at, (off_t)at); if (n <= 0) { if (n < 0) bad = true; break; } at += (size_t)n; }
          if (at < end) bad = true;                      // shorter than its size said: the serial loop below decides
        });
      for (auto &x : th) x.join();
      if (!bad) got = (size_t)fsize;
    }
    if (got == 0 || got < (size_t)fsize) {
      got = 0;
      while (got < (size_t)fsize) { const ssize_t n = pread(fd, raw.data() + got, (size_t)fsize - got, (off_t)got); if (n < 0) { rerr = true; break; } if (n == 0) break; got += (size_t)n; }
    }
    // a file that is still growing, or something that has no size and no positions (a FIFO, a process substitution): read the rest
    bool seekable = true;
    while (!rerr) {
      if (!raw.resize(got + (1 << 20) + 64)) { rerr = true; break; }
      ssize_t n = seekable ? pread(fd, raw.data() + got, 1 << 20, (off_t)got) : read(fd, raw.data() + got, 1 << 20);
      if (n < 0 && seekable && errno == ESPIPE) { seekable = false; continue; }
      if (n < 0 && errno == EINTR) continue;
      if (n < 0) rerr = true;
      if (n <= 0) break;
      got += (size_t)n;
    }
    close(fd);
    if (rerr) { err = std::string("read error on ") + path; return false; }
    raw.resize(got + 64);
    memset(raw.data() + got, 0, 64);                     // the inflaters may look a few bytes past the end
    raw.resize(got);
  }
  return true;
}

std::shared_ptr<const Text> read_text(const char *path, std::string &err, bool cacheable)
{
  codecs();
  int64_t fsize = 0, mtime = 0;
  if (!stat_of(path, fsize, mtime)) { err = std::string("cannot read ") + path; return nullptr; }
  const double budget = cache_budget_bytes();
  if (cacheable && budget > 0) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
      if (it->path == path) {
        if (it->size == fsize && it->mtime_ns == mtime) { CacheEntry e = *it; g_cache.erase(it); g_cache.push_back(e); return e.text; }
        g_cache.erase(it);                 // the file changed
        break;
      }
  }
  static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
  const auto c0 = std::chrono::steady_clock::now();
  Text raw;
  if (!read_raw(path, fsize, raw, err)) return nullptr;
  const auto c1 = std::chrono::steady_clock::now();
  auto text = std::make_shared<Text>();
  bool par = false;
  if (is_gzip(raw)) {
    // large single-member files: block-parallel inflate, accepted only on a CRC-32 and length match (pinflate.cpp)
    if (env_int("ITSX_PARALLEL_INFLATE", 1) != 0 && io_threads() > 1) {
      par = gunzip_parallel(raw.data(), raw.size(), *text, io_threads());
      if (par) g_parallel_inflates.fetch_add(1);
    }
    const bool ok = par || (g_ld.ok ? gunzip_libdeflate(raw, *text, err) : gunzip_zlib(raw, *text, err));
    if (!ok) { err += std::string(" in ") + path; return nullptr; }
  } else if (is_zstd(raw)) {
    if (!unzstd(raw, *text, err)) { err += std::string(" in ") + path; return nullptr; }
  } else text->swap(raw);
  if (text->capacity() > text->size() + text->size() / 4 + (1 << 20)) text->shrink_to_fit();      // a copy: only when it frees a lot
  if (trace) fprintf(stderr, "[itsx] read %s: file %.0f ms, decode%s %.0f ms (%.1f MB -> %.1f MB)\n", path, std::chrono::duration<double, std::milli>(c1 - c0).count(), par ? " (block-parallel)" : "",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c1).count(), raw.size() / 1e6, text->size() / 1e6);
  if (cacheable && budget > 0 && (double)text->size() <= budget) cache_insert(path, fsize, mtime, text, budget);
  return text;
}


// ------------------------------------------------------------------ TextStream
struct StreamImpl {
  std::string path;
  int64_t fsize = 0, mtime = 0;
  Text raw;
  std::shared_ptr<Text> text = std::make_shared<Text>();
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  size_t avail = 0;            // bytes of *text that are final
  size_t consumed = 0;         // compressed bytes behind them
  bool done = false, failed = false, joined = true;
  std::string err;
  size_t handed = 0;
  bool fastq = true;
  long long bound = -1;
  // next() looks for a record start with the lock released: the inflater's serial fallback (which may move the text) waits while a
  // scan is under way, and a scan that ran across a fallback is thrown away (advisor, round 4)
  bool scanning = false;
  int gen = 0;
};

namespace {
// first FASTQ record start at or after `from` inside t[0, n): a line that starts with '@' whose second line below starts with '+'
// (a quality line may start with '@'; then the second line below it is a sequence line, which never starts with '+')
size_t fastq_start_from(const char *t, size_t n, size_t from)
{
  size_t q = from;
  if (q > 0) { const char *nl = (const char *)memchr(t + q - 1, '\n', n - (q - 1)); if (!nl) return n; q = (size_t)(nl - t) + 1; }
  while (q < n) {
    if (t[q] == '@') {
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
}  // namespace

size_t fastq_record_start(const char *t, size_t n, size_t from) { return fastq_start_from(t, n, from); }

TextStream::TextStream() : s(new StreamImpl) {}
TextStream::~TextStream()
{
  if (!s->joined) { s->th.join(); s->joined = true; }
  delete s;
}

void TextStream::progress(size_t *avail, size_t *consumed, size_t *raw_size)
{
  std::lock_guard<std::mutex> g(s->mu);
  const size_t raw = (size_t)s->fsize;
  if (avail) *avail = s->avail;
  if (consumed) *consumed = s->done ? raw : s->consumed;
  if (raw_size) *raw_size = raw;
}
const char *TextStream::base() const { return s->text->data(); }

bool TextStream::open(const char *path, std::string &err, const char *shared_backing, bool *plain_input, int threads)
{
  codecs();
  s->path = path;
  if (plain_input) *plain_input = false;
  if (!stat_of(path, s->fsize, s->mtime)) { err = std::string("cannot read ") + path; return false; }
  const double budget = cache_budget_bytes();
  if (budget > 0 && !shared_backing) {                // already inflated in this process: one piece
    std::lock_guard<std::mutex> g(g_cache_mu);
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
      if (it->path == s->path && it->size == s->fsize && it->mtime_ns == s->mtime) {
        s->text = std::const_pointer_cast<Text>(it->text);      // read only from here on
        s->avail = s->text->size(); s->done = true;
        return true;
      }
  }
  if (!read_raw(path, s->fsize, s->raw, err)) return false;
  const int T = threads > 0 ? threads : io_threads();
  const bool par = is_gzip(s->raw) && env_int("ITSX_PARALLEL_INFLATE", 1) != 0 && T > 1;
  if (!par) {
    bool ok = true;
    const bool packed = is_gzip(s->raw) || is_zstd(s->raw);
    if (is_gzip(s->raw)) ok = g_ld.ok ? gunzip_libdeflate(s->raw, *s->text, err) : gunzip_zlib(s->raw, *s->text, err);
    else if (is_zstd(s->raw)) ok = unzstd(s->raw, *s->text, err);
    else { s->text->swap(s->raw); if (plain_input) *plain_input = true; }
    if (!ok) { err += std::string(" in ") + path; return false; }
    if (shared_backing && packed) {                   // (one thread's inflate: the text goes to the shared file afterwards)
      auto sh = std::make_shared<Text>();
      if (!sh->reserve_file(shared_backing, s->text->size() + 1)) { err = std::string("cannot create ") + shared_backing; return false; }
      memcpy(sh->data(), s->text->data(), s->text->size());
      const size_t m = s->text->size();
      sh->pin(false);
      s->text = sh;
      s->text->resize(m);
    }
    s->avail = s->text->size(); s->done = true;
    return true;
  }
  // the text must not move while slices of it are out: address space for any plausible ratio up front (pages arrive on first touch)
  const size_t n = s->raw.size();
  // (ITSX_STREAM_RESERVE_X / _MB: the factor and the constant of that reservation -- tests make it too small on purpose)
  const size_t rx = (size_t)std::max(1, env_int("ITSX_STREAM_RESERVE_X", 64)), rmb = (size_t)std::max(0, env_int("ITSX_STREAM_RESERVE_MB", 1024));
  size_t want = n * rx + (rmb << 20);
  const size_t least = std::min(want, n * 4 + ((size_t)1 << 20));
  if (shared_backing) {
    // (a sparse file: only the pages the text touches exist; 16 x is room for any FASTQ -- past it the inflater gives the file back)
    want = std::min(want, n * 16 + (rmb << 20));
    if (!s->text->reserve_file(shared_backing, std::max(want, least))) { err = std::string("cannot create the shared text ") + shared_backing; return false; }
  } else {
    while (want > least && !s->text->reserve(want)) want /= 2;
    if (s->text->capacity() < least && !s->text->reserve(least)) { err = std::string("out of memory inflating ") + path; return false; }
  }
  s->text->pin(true);
  s->joined = false;
  s->th = std::thread([this, T] {
    StreamImpl *z = s;
    const std::function<void(size_t, size_t)> progress = [z](size_t total, size_t consumed) {
      { std::lock_guard<std::mutex> g(z->mu); z->avail = total; z->consumed = consumed; }
      z->cv.notify_all();
    };
    bool ok = gunzip_parallel(z->raw.data(), z->raw.size(), *z->text, T, &progress);
    std::string e;
    if (ok) g_parallel_inflates.fetch_add(1);
    else {
      // not a file the block-parallel inflater takes (or it ran out of pinned room).  Whatever it reported so far was NOT yet
      // vouched for by a CRC: if a slice is out already the run is void, else the serial inflater delivers the file in one piece
      std::unique_lock<std::mutex> g(z->mu);
      if (z->handed == 0) {
        z->avail = 0;
        z->gen++;
        z->cv.wait(g, [z] { return !z->scanning; });       // nobody reads the text while it is unpinned and filled again
        if (z->handed != 0) e = "the block-parallel inflater gave up after slices had been handed out";
      }
      if (z->handed == 0) {
        g.unlock();
        z->text->pin(false);
        ok = g_ld.ok ? gunzip_libdeflate(z->raw, *z->text, e) : gunzip_zlib(z->raw, *z->text, e);
        g.lock();
        if (ok) z->avail = z->text->size();
      } else e = "the block-parallel inflater gave up after slices had been handed out";
    }
    { std::lock_guard<std::mutex> g(z->mu); z->done = true; z->failed = !ok; if (!ok) z->err = e + " in " + z->path; }
    z->cv.notify_all();
  });
  return true;
}

bool TextStream::next(size_t min_bytes, const char **ptr, size_t *nbytes, bool *last, std::string &err)
{
  std::unique_lock<std::mutex> g(s->mu);
  min_bytes = std::max<size_t>(min_bytes, 1);
  // how far before the target a record start is looked for (longer records: the loop widens it)
  size_t window = std::min<size_t>((size_t)256 << 10, std::max<size_t>(min_bytes / 4, (size_t)4 << 10));
  for (;;) {
    s->cv.wait(g, [&] { return s->done || s->avail >= s->handed + min_bytes + window; });
    if (s->failed) { err = s->err; return false; }
    const char *t = s->text->data();
    const size_t avail = s->avail;
    const bool done = s->done;
    if (s->handed == 0) {                              // what kind of file is it?  only FASTQ is cut; anything else arrives whole
      size_t f = 0;
      while (f < avail && (t[f] == '\n' || t[f] == '\r')) f++;
      s->fastq = f < avail && t[f] == '@';
    }
    if (!s->fastq && !done) { s->cv.wait(g, [&] { return s->done; }); continue; }
    // a slice is min_bytes .. 1.5 min_bytes long whatever has piled up meanwhile: chunks of even size keep the consumer's pipeline
    // even (and a plain file, all of it final at once, is cut like an inflating one)
    size_t target = avail;
    if (s->fastq && avail - s->handed > min_bytes + min_bytes / 2 + window) target = s->handed + min_bytes + min_bytes / 2;
    if (done && target == avail) {
      *ptr = t + s->handed; *nbytes = avail - s->handed; *last = true;
      s->handed = avail;
      return true;
    }
    // a record start about `window` before the target (the record-start test needs the two lines that follow the title)
    const size_t from = target > s->handed + window ? target - window : s->handed + 1;
    const int gen = s->gen;
    s->scanning = true;
    g.unlock();
    const size_t c = fastq_start_from(t, avail, from);
    g.lock();
    s->scanning = false;
    s->cv.notify_all();
    if (gen != s->gen) continue;                        // the text was void (the inflater fell back): wait for the real one
    if (c < avail && c > s->handed) {
      *ptr = t + s->handed; *nbytes = c - s->handed; *last = false;
      s->handed = c;
      return true;
    }
    if (done) {                                         // no record start behind `from`: the rest is one slice
      *ptr = t + s->handed; *nbytes = avail - s->handed; *last = true;
      s->handed = avail;
      return true;
    }
    window *= 8;                                        // records longer than the window: look further back, wait for more
    if (window > min_bytes) min_bytes = window;
  }
}

bool TextStream::next_records(size_t n_records, const char **ptr, size_t *nbytes, size_t *got, bool *last, std::string &err)
{
  std::unique_lock<std::mutex> g(s->mu);
  const size_t want_lines = n_records * 4;
  size_t pos = s->handed, lines = 0;
  for (;;) {
    if (s->failed) { err = s->err; return false; }
    const int gen = s->gen;
    const char *t = s->text->data();
    const size_t avail = s->avail;
    const bool done = s->done;
    // count newlines in what is final and not yet looked at (with the lock released: the serial fallback waits for scans)
    s->scanning = true;
    g.unlock();
    while (lines < want_lines && pos < avail) {
      const char *q = (const char *)memchr(t + pos, '\n', avail - pos);
      if (!q) { pos = avail; break; }
      pos = (size_t)(q - t) + 1; lines++;
    }
    g.lock();
    s->scanning = false;
    s->cv.notify_all();
    if (gen != s->gen) { pos = s->handed; lines = 0; continue; }      // the text was void (the inflater fell back): start over
    if (lines >= want_lines || done) {
      size_t end = pos;
      if (lines < want_lines && done) end = avail;                    // (a last line without its newline belongs to the last record)
      *ptr = t + s->handed; *nbytes = end - s->handed;
      if (lines >= want_lines) *got = n_records;
      else { const size_t all = lines + ((end > s->handed && t[end - 1] != '\n') ? 1 : 0); *got = all / 4; }
      s->handed = end;
      *last = done && end >= avail;
      return true;
    }
    s->cv.wait(g, [&] { return s->done || s->failed || s->avail > avail; });
  }
}

long long TextStream::records_bound()
{
  size_t n;
  {
    std::lock_guard<std::mutex> g(s->mu);
    if (!s->done || s->failed) return -1;
    if (s->bound >= 0) return s->bound;
    n = s->avail;
  }
  const char *t = s->text->data();
  const int T = std::max(1, std::min(io_threads(), (int)(n >> 24) + 1));
  std::vector<size_t> cnt((size_t)T, 0);
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++)
    th.emplace_back([&, k] {
      const char *p = t + n / (size_t)T * (size_t)k, *e = (k + 1 == T) ? t + n : t + n / (size_t)T * (size_t)(k + 1);
      size_t c = 0;
      while (p < e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); if (!q) break; c++; p = q + 1; }
      cnt[(size_t)k] = c;
    });
  for (auto &x : th) x.join();
  size_t lines = 1;                                   // (a last line without its newline)
  for (size_t c : cnt) lines += c;
  size_t f = 0;
  while (f < n && (t[f] == '\n' || t[f] == '\r')) f++;
  const bool fastq = f < n && t[f] == '@';
  const long long b = (long long)(lines / (fastq ? 4 : 2)) + 1;
  std::lock_guard<std::mutex> g(s->mu);
  s->bound = b;
  return b;
}

bool TextStream::finish(bool keep, std::string &err)
{
  if (!s->joined) { s->th.join(); s->joined = true; }
  if (s->failed) { err = s->err; return false; }
  s->text->pin(false);
  Text().swap(s->raw);
  if (keep) {
    const double budget = cache_budget_bytes();
    // (the reserved address space behind the text costs nothing, and trimming it would need the slices to be gone)
    if (budget > 0 && (double)s->text->size() <= budget) cache_insert(s->path.c_str(), s->fsize, s->mtime, s->text, budget);
  }
  return true;
}

// ------------------------------------------------------------------ ordered block writer
struct Job { uint64_t seq; std::string in, out; bool ok; };

struct WriterImpl {
  FILE *fp = nullptr;
  int kind = PLAIN, level = 6;
  size_t block = 4u << 20;
  std::string cur, path;
  std::shared_ptr<std::string> kept;       // PLAIN + keep_text: everything written
  bool failed = false;
  // pool
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_todo, cv_done;
  std::deque<Job *> todo;
  std::map<uint64_t, Job *> done;
  uint64_t next_submit = 0, next_write = 0;
  bool stop = false;

  static bool compress_block(int kind, int level, void *ldc, const std::string &in, std::string &out)
  {
    if (kind == GZIP) {
      if (ldc) {
        out.resize(g_ld.gz_bound(ldc, in.size()));
        const size_t n = g_ld.gz_c(ldc, in.data(), in.size(), &out[0], out.size());
        if (n == 0) return false;
        out.resize(n);
        return true;
      }
      z_stream zs; memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
      out.resize(deflateBound(&zs, (uLong)in.size()) + 64);
      zs.next_in = (Bytef *)in.data(); zs.avail_in = (uInt)in.size();
      zs.next_out = (Bytef *)&out[0]; zs.avail_out = (uInt)out.size();
      const int rc = deflate(&zs, Z_FINISH);
      const size_t n = out.size() - zs.avail_out;
      deflateEnd(&zs);
      if (rc != Z_STREAM_END) return false;
      out.resize(n);
      return true;
    }
    out.resize(g_zs.bound(in.size()));
    const size_t n = g_zs.compress(&out[0], out.size(), in.data(), in.size(), 3);
    if (g_zs.is_error(n)) return false;
    out.resize(n);
    return true;
  }
  void work()
  {
    void *ldc = (kind == GZIP && g_ld.ok) ? g_ld.alloc_c(level) : nullptr;
    for (;;) {
      Job *j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_todo.wait(lk, [&] { return stop || !todo.empty(); });
        if (todo.empty()) break;
        j = todo.front(); todo.pop_front();
      }
      j->ok = compress_block(kind, level, ldc, j->in, j->out);
      std::string().swap(j->in);
      { std::lock_guard<std::mutex> lk(mu); done[j->seq] = j; }
      cv_done.notify_all();
    }
    if (ldc) g_ld.free_c(ldc);
  }
  // caller holds no lock; writes every finished block that is next in order, waits while `pending` exceeds the bound
  void drain(size_t max_pending)
  {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      auto it = done.find(next_write);
      if (it != done.end()) {
        Job *j = it->second; done.erase(it); next_write++;
        lk.unlock();
        if (!j->ok || fwrite(j->out.data(), 1, j->out.size(), fp) != j->out.size()) failed = true;
        delete j;
        lk.lock();
        continue;
      }
      if (next_submit - next_write <= max_pending) break;
      cv_done.wait(lk);
    }
  }
  void submit(std::string &&data)
  {
    Job *j = new Job{next_submit++, std::move(data), std::string(), false};
    { std::lock_guard<std::mutex> lk(mu); todo.push_back(j); }
    cv_todo.notify_one();
    drain(workers.size() * 3);
  }
};

PieceCompressor::PieceCompressor(int kind) : kind_(kind), level_(std::min(9, std::max(1, env_int("ITSX_GZIP_LEVEL", 6))))
{
  codecs();
  if (kind_ == ZSTD && !g_zs.ok) ok_ = false;
  if (kind_ == GZIP && g_ld.ok) ldc_ = g_ld.alloc_c(level_);
}
PieceCompressor::~PieceCompressor() { if (ldc_) g_ld.free_c(ldc_); }
bool PieceCompressor::run(const std::string &in, std::string &out)
{
  if (kind_ == PLAIN) { out = in; return true; }
  if (!ok_) return false;
  return WriterImpl::compress_block(kind_, level_, ldc_, in, out);
}

BlockWriter::BlockWriter() : w(new WriterImpl) {}
BlockWriter::~BlockWriter() { std::string e; if (w->fp) close(e); delete w; }

bool BlockWriter::open(const char *path, int kind, std::string &err, bool keep_text)
{
  codecs();
  if (kind != PLAIN && kind != GZIP && kind != ZSTD) { err = "unknown compression kind"; return false; }
  if (kind == ZSTD && !g_zs.ok) { err = "zstd output requested but libzstd.so.1 could not be loaded"; return false; }
  w->fp = fopen(path, "wb");
  if (!w->fp) { err = std::string("cannot write ") + path; return false; }
  setvbuf(w->fp, nullptr, _IOFBF, 1 << 20);
  w->kind = kind;
  w->path = path;
  if (keep_text && kind == PLAIN && cache_budget_bytes() > 0) w->kept = std::make_shared<std::string>();
  w->level = std::min(9, std::max(1, env_int("ITSX_GZIP_LEVEL", 6)));
  w->block = (size_t)std::max(1, env_int("ITSX_IO_BLOCK_KB", 4096)) << 10;
  if (kind != PLAIN) {
    const int nt = io_threads();
    for (int i = 0; i < nt; i++) w->workers.emplace_back([this] { w->work(); });
  }
  return true;
}

void BlockWriter::put(const char *p, size_t n)
{
  if (!w->fp || n == 0) return;
  if (w->kind == PLAIN) {
    if (fwrite(p, 1, n, w->fp) != n) w->failed = true;
    if (w->kept) { if ((double)(w->kept->size() + n) <= cache_budget_bytes()) w->kept->append(p, n); else w->kept.reset(); }
    return;
  }
  while (n > 0) {
    const size_t take = std::min(n, w->block - w->cur.size());
    w->cur.append(p, take); p += take; n -= take;
    if (w->cur.size() >= w->block) { std::string b; b.swap(w->cur); w->cur.reserve(w->block); w->submit(std::move(b)); }
  }
}

bool BlockWriter::close(std::string &err)
{
  if (!w->fp) return true;
  if (w->kind != PLAIN) {
    if (!w->cur.empty() || w->next_submit == 0) { std::string b; b.swap(w->cur); w->submit(std::move(b)); }   // an empty file is still one valid member
    w->drain(0);
    { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
    w->cv_todo.notify_all();
    for (auto &t : w->workers) t.join();
    w->workers.clear();
  }
  if (fflush(w->fp) != 0 || ferror(w->fp)) w->failed = true;
  if (fclose(w->fp) != 0) w->failed = true;
  w->fp = nullptr;
  if (w->failed) { err = "compressing or writing the output failed"; return false; }
  if (w->kept) { auto t = std::make_shared<Text>(); t->adopt(std::move(*w->kept)); cache_put(w->path.c_str(), t); w->kept.reset(); }
  return true;
}

}  // namespace itsx_io
