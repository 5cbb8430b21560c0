// fastq_io.cpp -- see fastq_io.h.  Host-only.
#include "fastq_io.h"
#include <dlfcn.h>
#include <sys/stat.h>
#include <zlib.h>
#include <algorithm>
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
  const char *off = getenv("ITSX_IO_LIBDEFLATE");
  if (!(off && atoi(off) == 0)) {
    for (const char *n : {"libdeflate.so.0", "libdeflate.so"}) { g_ld.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_ld.h) break; }
    if (g_ld.h) {
      g_ld.ok = sym(g_ld.h, "libdeflate_alloc_compressor", g_ld.alloc_c) && sym(g_ld.h, "libdeflate_gzip_compress_bound", g_ld.gz_bound) &&
                sym(g_ld.h, "libdeflate_gzip_compress", g_ld.gz_c) && sym(g_ld.h, "libdeflate_free_compressor", g_ld.free_c) &&
                sym(g_ld.h, "libdeflate_alloc_decompressor", g_ld.alloc_d) && sym(g_ld.h, "libdeflate_gzip_decompress_ex", g_ld.gz_d) &&
                sym(g_ld.h, "libdeflate_free_decompressor", g_ld.free_d);
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

int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

// ------------------------------------------------------------------ decompression of a whole buffer
bool is_gzip(const std::string &r, size_t pos = 0) { return r.size() >= pos + 2 && (unsigned char)r[pos] == 0x1f && (unsigned char)r[pos + 1] == 0x8b; }
bool is_zstd(const std::string &r) { return r.size() >= 4 && (unsigned char)r[0] == 0x28 && (unsigned char)r[1] == 0xb5 && (unsigned char)r[2] == 0x2f && (unsigned char)r[3] == 0xfd; }

// The trailer's ISIZE (length mod 2^32 of the LAST member) is the whole length for the usual single-member file below
// 4 GB; anything implausible falls back to 4x the compressed size.  Only a first guess: both inflaters grow on demand.
size_t gzip_size_hint(const std::string &raw)
{
  const size_t guess = std::max<size_t>(raw.size() * 4, 1 << 20);
  if (raw.size() < 18) return guess;
  const unsigned char *t = (const unsigned char *)raw.data() + raw.size() - 4;
  const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
  if (isize >= raw.size() / 2 && isize <= raw.size() * 40) return isize + 64;
  return guess;
}

bool gunzip_libdeflate(const std::string &raw, std::string &out, std::string &err)
{
  void *d = g_ld.alloc_d();
  if (!d) { err = "libdeflate: out of memory"; return false; }
  size_t pos = 0, opos = 0;
  out.resize(gzip_size_hint(raw));
  bool ok = true;
  while (pos < raw.size() && is_gzip(raw, pos)) {
    size_t ain = 0, aout = 0;
    const int rc = g_ld.gz_d(d, raw.data() + pos, raw.size() - pos, &out[0] + opos, out.size() - opos, &ain, &aout);
    if (rc == 3) { out.resize(out.size() * 2); continue; }              // LIBDEFLATE_INSUFFICIENT_SPACE: retry this member
    if (rc != 0) { err = "corrupt gzip data"; ok = false; break; }
    pos += ain; opos += aout;
  }
  g_ld.free_d(d);
  out.resize(ok ? opos : 0);
  return ok;
}
bool gunzip_zlib(const std::string &raw, std::string &out, std::string &err)
{
  z_stream zs; memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) { err = "zlib: inflateInit2 failed"; return false; }
  out.resize(gzip_size_hint(raw));
  size_t pos = 0, opos = 0;
  bool ok = true;
  while (ok && pos < raw.size() && is_gzip(raw, pos)) {
    for (;;) {
      if (opos == out.size()) out.resize(out.size() * 2);
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
bool unzstd(const std::string &raw, std::string &out, std::string &err)
{
  if (!g_zs.ok) { err = "zstd-compressed file but libzstd.so.1 could not be loaded"; return false; }
  void *ds = g_zs.create_d();
  if (!ds) { err = "zstd: out of memory"; return false; }
  out.resize(std::max<size_t>(raw.size() * 4, 1 << 20));
  ZBuf in{(void *)raw.data(), raw.size(), 0}, ob{&out[0], out.size(), 0};
  bool ok = true; size_t last = 0;
  while (in.pos < in.size) {
    if (ob.pos == ob.size) { out.resize(out.size() * 2); ob.p = &out[0]; ob.size = out.size(); }
    last = g_zs.decompress_stream(ds, &ob, &in);
    if (g_zs.is_error(last)) { err = "corrupt zstd data"; ok = false; break; }
  }
  // the input is consumed; a frame that still wants output has only been waiting for room
  while (ok && last != 0) {
    if (ob.pos == ob.size) { out.resize(out.size() * 2); ob.p = &out[0]; ob.size = out.size(); }
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
struct CacheEntry { std::string path; int64_t size, mtime_ns; std::shared_ptr<const std::string> text; };
std::mutex g_cache_mu;
std::deque<CacheEntry> g_cache;          // most recent at the back

// ITSX_TEXT_CACHE_GB, or by default a quarter of the memory this process may still take (MemAvailable, cut by the cgroup's limit), at
// least 4 GB and at most 32: the writer of a 10 M-read sample inflated its 9 GB of text a second time because the cache held 4 GB
double cache_budget_bytes()
{
  if (const char *e = getenv("ITSX_TEXT_CACHE_GB")) return atof(e) * (double)(1ull << 30);
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

static void cache_insert(const char *path, int64_t fsize, int64_t mtime, std::shared_ptr<const std::string> text, double budget)
{
  std::lock_guard<std::mutex> g(g_cache_mu);
  for (auto it = g_cache.begin(); it != g_cache.end(); ++it) if (it->path == path) { g_cache.erase(it); break; }
  g_cache.push_back(CacheEntry{path, fsize, mtime, text});
  double total = 0;
  for (auto &e : g_cache) total += (double)e.text->size();
  while (total > budget && g_cache.size() > 1) { total -= (double)g_cache.front().text->size(); g_cache.pop_front(); }
}

void cache_put(const char *path, std::shared_ptr<const std::string> text)
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

std::shared_ptr<const std::string> read_text(const char *path, std::string &err, bool cacheable)
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
  static const bool trace = getenv("ITSX_TRACE_ALLOC") != nullptr;
  const auto c0 = std::chrono::steady_clock::now();
  FILE *f = fopen(path, "rb");
  if (!f) { err = std::string("cannot read ") + path; return nullptr; }
  std::string raw;
  raw.reserve((size_t)fsize + 64);
  raw.resize((size_t)fsize);
  size_t got = 0;
  while (got < raw.size()) { const size_t n = fread(&raw[got], 1, raw.size() - got, f); if (n == 0) break; got += n; }
  raw.resize(got);
  // a file that is still growing, or a pipe: read the rest
  char tail[1 << 16]; size_t n;
  while ((n = fread(tail, 1, sizeof(tail), f)) > 0) raw.append(tail, n);
  const bool rerr = ferror(f) != 0;
  fclose(f);
  if (rerr) { err = std::string("read error on ") + path; return nullptr; }
  const auto c1 = std::chrono::steady_clock::now();
  auto text = std::make_shared<std::string>();
  bool par = false;
  if (is_gzip(raw)) {
    // large single-member files: block-parallel inflate, accepted only on a CRC-32 and length match (pinflate.cpp)
    if (env_int("ITSX_PARALLEL_INFLATE", 1) != 0 && io_threads() > 1) {
      const size_t n0 = raw.size();
      raw.append(16, '\0');
      par = gunzip_parallel(raw.data(), n0, *text, io_threads());
      raw.resize(n0);
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
  if (w->kept) { cache_put(w->path.c_str(), w->kept); w->kept.reset(); }
  return true;
}

}  // namespace itsx_io
