// pinflate.cpp -- block-parallel inflate of an ordinary gzip file (one member or many).  Host-only.
//
// Why: once the path itself runs at ~1 s per million reads, a file-to-file run of the reference's CLI spends most of
// its time inflating the input FASTQ on one thread (libdeflate: 0.55 GB/s of text).  A deflate stream has no index,
// but it can still be decoded from the middle (the pugz / rapidgzip idea, restated here for text input):
//   1. the compressed bytes are cut into chunks; every chunk but the first looks for the start of a dynamic-Huffman
//      block near its beginning by trying bit offsets: a candidate must carry a valid, complete code description,
//      decode to printable text only, and be followed by another valid block header;
//   2. each chunk is inflated from its block start up to the next chunk's block start into 16-bit symbols: a byte, or
//      -- for a back-reference that reaches into the 32 KB BEFORE the chunk, which the thread does not have -- a
//      marker naming the position in that unknown window;
//   3. the windows are then known front to back (32 KB per chunk, sequential and cheap) and every chunk replaces its
//      markers while it narrows its symbols into the output, in parallel.
// Large files go through in rounds of a bounded number of chunks, so the 16-bit staging stays a few GB at most.
// Members that end inside a chunk are walked through (trailer, next header, on with the next member's blocks).
// Safety net: the result is accepted only if EVERY member's length and CRC-32 equal its trailer and the last member ends
// where the file ends; anything else (a false block start, a chunk that inflates beyond its staging buffer, bytes behind
// the last member) returns false and the caller inflates serially as before.
#include "fastq_io.h"
#include <sys/mman.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace itsx_io {
namespace {

constexpr int PRIMARY = 11;                 // bits resolved by one table look-up
constexpr int WSIZE = 32768;

struct Huff {
  uint16_t count[16];                       // codes of each length
  uint16_t symbol[288];                     // symbols in canonical order (slow path, codes longer than PRIMARY)
  uint32_t fast[1 << PRIMARY];              // (symbol << 4) | length, 0 = longer code / invalid
};

// canonical Huffman code from lengths; false if over-subscribed, or incomplete (a single 1-bit code is allowed, as in zlib)
bool build(Huff &h, const uint8_t *len, int n, bool allow_empty = false)
{
  memset(h.count, 0, sizeof(h.count));
  for (int i = 0; i < n; i++) h.count[len[i]]++;
  if (h.count[0] == n) { if (allow_empty) memset(h.fast, 0, sizeof(h.fast)); return allow_empty; }   // zlib: a block of literals only may carry no distance code
  int left = 1;
  for (int l = 1; l <= 15; l++) { left <<= 1; left -= h.count[l]; if (left < 0) return false; }
  if (left > 0 && !(n - h.count[0] == 1 && h.count[1] == 1)) return false;
  uint16_t offs[16]; offs[1] = 0;
  for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
  for (int i = 0; i < n; i++) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
  memset(h.fast, 0, sizeof(h.fast));
  uint32_t code = 0; int idx = 0;
  for (int l = 1; l <= 15; l++) {
    for (int k = 0; k < h.count[l]; k++, idx++, code++) {
      if (l > PRIMARY) continue;
      uint32_t rev = 0;
      for (int b = 0; b < l; b++) rev |= ((code >> b) & 1u) << (l - 1 - b);
      for (uint32_t e = rev; e < (1u << PRIMARY); e += (1u << l)) h.fast[e] = ((uint32_t)h.symbol[idx] << 4) | (uint32_t)l;
    }
    code <<= 1;
  }
  return true;
}

// decode one symbol from the low bits of `buf`; consumed bits are added to c.  -1 = invalid code
inline int decode(const Huff &h, uint64_t buf, int &c)
{
  const uint32_t e = h.fast[(buf >> c) & ((1u << PRIMARY) - 1)];
  if (e) { c += (int)(e & 15u); return (int)(e >> 4); }
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; l++) {
    code |= (int)((buf >> (c + l - 1)) & 1u);
    const int cnt = h.count[l];
    if (code - cnt < first) { c += l; return h.symbol[index + (code - first)]; }
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -1;
}

const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct In {                                 // the compressed bytes; in[n .. n+8) is readable padding
  const uint8_t *in; size_t n;
  inline uint64_t peek(uint64_t bitpos) const { uint64_t w; memcpy(&w, in + (bitpos >> 3), 8); return w >> (bitpos & 7); }   // >= 57 valid bits
  inline bool room(uint64_t bitpos) const { return (bitpos >> 3) <= n; }
};

// the code description of a dynamic block at `pos` (just past the 3 header bits); false = not a valid description
bool read_dynamic(const In &s, uint64_t &pos, Huff &ll, Huff &dd)
{
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  if (!s.room(pos + 14 + 19 * 3)) return false;
  uint64_t b = s.peek(pos);
  const int nlen = (int)(b & 31) + 257, ndist = (int)((b >> 5) & 31) + 1, ncode = (int)((b >> 10) & 15) + 4;
  pos += 14;
  if (nlen > 286 || ndist > 30) return false;
  uint8_t cl[19] = {0};
  b = s.peek(pos);
  for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)((b >> (3 * i)) & 7);
  pos += 3 * (uint64_t)ncode;
  Huff lc;
  if (!build(lc, cl, 19)) return false;
  uint8_t lens[320];
  int i = 0;
  while (i < nlen + ndist) {
    if (!s.room(pos + 16)) return false;
    b = s.peek(pos);
    int c = 0;
    const int sym = decode(lc, b, c);
    if (sym < 0) return false;
    if (sym < 16) { lens[i++] = (uint8_t)sym; }
    else {
      int prev = 0, rep;
      if (sym == 16) { if (i == 0) return false; prev = lens[i - 1]; rep = 3 + (int)((b >> c) & 3); c += 2; }
      else if (sym == 17) { rep = 3 + (int)((b >> c) & 7); c += 3; }
      else { rep = 11 + (int)((b >> c) & 127); c += 7; }
      if (i + rep > nlen + ndist) return false;
      while (rep--) lens[i++] = (uint8_t)prev;
    }
    pos += (uint64_t)c;
  }
  if (lens[256] == 0) return false;
  return build(ll, lens, nlen) && build(dd, lens + nlen, ndist, true);
}

void fixed_tables(Huff &ll, Huff &dd)
{
  uint8_t l[288];
  for (int i = 0; i < 144; i++) l[i] = 8;
  for (int i = 144; i < 256; i++) l[i] = 9;
  for (int i = 256; i < 280; i++) l[i] = 7;
  for (int i = 280; i < 288; i++) l[i] = 8;
  build(ll, l, 288);
  uint8_t d[32];                            // 30 and 31 complete the code; using them is an error (checked at the use)
  for (int i = 0; i < 32; i++) d[i] = 5;
  build(dd, d, 32);
}

inline bool texty(int c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// Anonymous memory with transparent huge pages asked for: the staging buffers and the output are written once, front to
// back, and with 4-KB pages the page faults cost more than the decoding (measured: 269 -> 40 ms for 37 MB of text).
bool huge_pages() { static const bool on = !(sw_get("ITSX_HUGEPAGES") && atoi(sw_get("ITSX_HUGEPAGES")) == 0); return on; }
struct HugeBuf {
  void *p = nullptr; size_t bytes = 0;
  bool alloc(size_t want)
  {
    release();
    const size_t two_mb = (size_t)2 << 20;
    bytes = (want + two_mb - 1) / two_mb * two_mb;
    p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { p = nullptr; bytes = 0; return false; }
    if (huge_pages()) (void)madvise(p, bytes, MADV_HUGEPAGE);
    return true;
  }
  void release() { if (p) munmap(p, bytes); p = nullptr; bytes = 0; }
  ~HugeBuf() { release(); }
  HugeBuf() = default;
  HugeBuf(const HugeBuf &) = delete;
  HugeBuf &operator=(const HugeBuf &) = delete;
};

struct Sink {                               // 16-bit symbols of one chunk: < 256 a byte, else 256 + position in the unknown window
  uint16_t *v = nullptr; size_t n = 0, cap = 0;
  bool text_only = false;                   // probing: literals must be text
};

enum { BLK_OK = 0, BLK_FINAL = 1, BLK_BAD = 2, BLK_LIMIT = 3 };

// inflate ONE block starting at pos (its 3 header bits included); pos is left at the next block
int inflate_block(const In &s, uint64_t &pos, Sink &out, Huff &ll, Huff &dd)
{
  if (!s.room(pos + 3)) return BLK_BAD;
  const uint64_t hb = s.peek(pos);
  const int final = (int)(hb & 1), type = (int)((hb >> 1) & 3);
  pos += 3;
  if (type == 3) return BLK_BAD;
  uint16_t *const v = out.v;
  if (type == 0) {
    pos = (pos + 7) & ~(uint64_t)7;
    const size_t p = (size_t)(pos >> 3);
    if (p + 4 > s.n) return BLK_BAD;
    const unsigned len = s.in[p] | (s.in[p + 1] << 8), nlen = s.in[p + 2] | (s.in[p + 3] << 8);
    if ((len ^ 0xffffu) != nlen || p + 4 + len > s.n) return BLK_BAD;
    if (out.n + len > out.cap) return BLK_LIMIT;
    for (unsigned i = 0; i < len; i++) { const int c = s.in[p + 4 + i]; if (out.text_only && !texty(c)) return BLK_BAD; v[out.n++] = (uint16_t)c; }
    pos = (uint64_t)(p + 4 + len) * 8;
    return final ? BLK_FINAL : BLK_OK;
  }
  if (type == 1) fixed_tables(ll, dd);
  else if (!read_dynamic(s, pos, ll, dd)) return BLK_BAD;
  for (;;) {
    if (!s.room(pos + 48)) return BLK_BAD;
    const uint64_t b = s.peek(pos);
    int c = 0;
    int sym = decode(ll, b, c);
    if (sym < 0) return BLK_BAD;
    if (sym < 256) {
      if (out.text_only && !texty(sym)) return BLK_BAD;
      if (out.n >= out.cap) return BLK_LIMIT;
      v[out.n++] = (uint16_t)sym;
      pos += (uint64_t)c;
      continue;
    }
    if (sym == 256) { pos += (uint64_t)c; return final ? BLK_FINAL : BLK_OK; }
    sym -= 257;
    if (sym >= 29) return BLK_BAD;
    int len = LBASE[sym] + (int)((b >> c) & ((1u << LEXT[sym]) - 1)); c += LEXT[sym];
    const int ds = decode(dd, b, c);
    if (ds < 0 || ds >= 30) return BLK_BAD;
    const int dist = DBASE[ds] + (int)((b >> c) & ((1u << DEXT[ds]) - 1)); c += DEXT[ds];
    pos += (uint64_t)c;
    const size_t have = out.n;
    if (have + (size_t)len > out.cap) return BLK_LIMIT;
    uint16_t *o = v + have;
    const int64_t src0 = (int64_t)have - dist;
    if (src0 >= 0) { const uint16_t *sp = v + src0; for (int i = 0; i < len; i++) o[i] = sp[i]; }
    else
      for (int i = 0; i < len; i++) {
        const int64_t src = src0 + i;
        o[i] = src >= 0 ? v[(size_t)src] : (uint16_t)(256 + WSIZE + src);   // src in [-32768, -1]: the window before the chunk
      }
    out.n = have + (size_t)len;
  }
}

// first bit position >= from (and < to) at which a non-final dynamic block starts that survives a trial decode
bool find_block(const In &s, uint64_t from, uint64_t to, uint64_t &found)
{
  Huff ll, dd;
  // (a candidate that decodes 32 K symbols of printable text from one code description is a block: every symbol of a false start would
  // have to be a valid code AND text; the chunk is decoded again from there anyway, and every member's CRC is the safety net)
  std::vector<uint16_t> pbuf((size_t)1 << 15);
  Sink probe; probe.text_only = true; probe.v = pbuf.data();
  for (uint64_t p = from; p < to; p++) {
    if (!s.room(p + 64)) return false;
    const uint64_t b = s.peek(p);
    if ((b & 7) != 4) continue;                                   // BFINAL 0, BTYPE 2
    if ((int)((b >> 3) & 31) > 29 || (int)((b >> 8) & 31) > 29) continue;
    uint64_t q = p + 3;
    if (!read_dynamic(s, q, ll, dd)) continue;
    probe.n = 0; probe.cap = pbuf.size();
    q = p;
    int rc = inflate_block(s, q, probe, ll, dd);
    if (rc == BLK_LIMIT) { found = p; return true; }             // 32 K symbols of text from one code description: it is a block
    if (rc != BLK_OK || probe.n < 64) continue;
    // the next block must look like a block too: header type valid, and some text
    probe.n = 0; probe.cap = 4096;
    uint64_t q2 = q;
    rc = inflate_block(s, q2, probe, ll, dd);
    if (rc == BLK_BAD) continue;
    found = p;
    return true;
  }
  return false;
}

size_t gzip_header_len(const uint8_t *p, size_t n)
{
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return 0;
  const int flg = p[3];
  size_t o = 10;
  if (flg & 4) { if (o + 2 > n) return 0; o += 2 + (size_t)(p[o] | (p[o + 1] << 8)); }
  if (flg & 8) { while (o < n && p[o]) o++; o++; }
  if (flg & 16) { while (o < n && p[o]) o++; o++; }
  if (flg & 2) o += 2;
  return o < n ? o : 0;
}

inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct MemberEnd { size_t at; uint32_t crc, isize; };      // a gzip member ended after `at` symbols of the chunk; its trailer

struct Chunk {
  uint64_t start = 0, want_end = 0, end = 0;   // block start; the boundary it must reach (0: first boundary >= min_end); where it stopped
  uint64_t min_end = 0;
  bool final = false, ok = false;
  Sink out;
  std::vector<MemberEnd> ends;
};

void run_chunk(const In &s, Chunk &c)
{
  Huff ll, dd;
  uint64_t pos = c.start;
  for (;;) {
    const int rc = inflate_block(s, pos, c.out, ll, dd);
    if (rc == BLK_BAD || rc == BLK_LIMIT) return;
    if (rc == BLK_FINAL) {
      // end of a gzip member: its trailer, then the end of the file or the next member's header (a member boundary is a
      // block boundary like any other, so the stop rules below apply to it too)
      size_t b = (size_t)((pos + 7) >> 3);
      if (b + 8 > s.n) return;
      c.ends.push_back(MemberEnd{c.out.n, le32(s.in + b), le32(s.in + b + 4)});
      b += 8;
      if (b == s.n) { c.final = true; c.end = (uint64_t)b * 8; c.ok = (c.want_end == 0); return; }
      const size_t hl = gzip_header_len(s.in + b, s.n - b);
      if (!hl) return;                       // bytes after the last member that are not a member: the serial inflater decides
      pos = (uint64_t)(b + hl) * 8;
    }
    if (c.want_end) { if (pos == c.want_end) { c.end = pos; c.ok = true; return; } if (pos > c.want_end) return; }
    else if (pos >= c.min_end) { c.end = pos; c.ok = true; return; }
  }
}

template <class F> void parallel(int T, F fn)
{
  std::vector<std::thread> th;
  for (int t = 1; t < T; t++) th.emplace_back([&fn, t] { fn(t); });
  fn(0);
  for (auto &x : th) x.join();
}

}  // namespace

bool gunzip_parallel(const char *data, size_t n, Text &out, int threads, const std::function<void(size_t, size_t)> *progress)
{
  const uint8_t *p = (const uint8_t *)data;
  const size_t hdr = gzip_header_len(p, n);
  size_t chunk_bytes = 1u << 20;
  if (const char *e = sw_get("ITSX_PINFLATE_CHUNK_KB")) chunk_bytes = std::max<size_t>(16, (size_t)atol(e)) << 10;      // tests use small chunks
  if (!hdr || threads < 2 || n < 4 * chunk_bytes) return false;
  const uint32_t last_len = le32(p + n - 4);
  const In s{p, n};
  const uint64_t stream_end_bits = (uint64_t)n * 8;
  out.clear();
  size_t guess = (size_t)last_len;           // right for the usual single-member file; otherwise only a first guess
  if (guess < n || guess > n * 40) guess = n * 4;
  out.reserve(guess + 64);
  // (the output is written once, front to back: Text asks for huge pages under its mappings itself)
  std::vector<uint8_t> window(WSIZE, 0);
  // staging: one buffer of 16-bit symbols per chunk slot, reused round after round (12 symbols per compressed byte of a
  // chunk; a chunk that would need more gives the file back to the serial inflater)
  const size_t stage_syms = chunk_bytes * 12 + ((size_t)1 << 20);
  std::vector<HugeBuf> stage((size_t)threads);
  for (auto &hb : stage) if (!hb.alloc(stage_syms * 2)) return false;
  uint64_t pos = (uint64_t)hdr * 8;
  uLong crc = crc32(0L, Z_NULL, 0);          // of the member being assembled
  uint64_t member_len = 0;
  bool finished = false;
  static const bool trace = sw_get("ITSX_TRACE_ALLOC") != nullptr;
  double t_find = 0, t_dec = 0, t_res = 0; int rounds = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  while (!finished) {
    const auto r0 = now();
    // ---- one round: up to `threads` chunks of chunk_bytes each, starting at the known block boundary `pos`
    const size_t b0 = (size_t)(pos >> 3);
    int nc = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, (n - b0 + chunk_bytes - 1) / chunk_bytes));
    std::vector<Chunk> ch((size_t)nc);
    ch[0].start = pos;
    std::vector<char> found((size_t)nc, 1);
    parallel(nc, [&](int k) {                                    // block starts of chunks 1..
      if (k == 0) return;
      const uint64_t from = (uint64_t)(b0 + (size_t)k * chunk_bytes) * 8;
      const uint64_t to = std::min<uint64_t>(from + (uint64_t)chunk_bytes * 8, stream_end_bits);
      uint64_t f = 0;
      if (from < to && find_block(s, from, to, f)) ch[(size_t)k].start = f; else found[(size_t)k] = 0;
    });
    const auto r1 = now();
    // chunks whose start was not found (or that lie past the end) are merged into their predecessor
    std::vector<Chunk> live;
    for (int k = 0; k < nc; k++)
      if (k == 0 || found[(size_t)k]) {
        live.emplace_back();
        live.back().start = ch[(size_t)k].start;
        live.back().out.v = (uint16_t *)stage[live.size() - 1].p; live.back().out.cap = stage_syms; live.back().out.n = 0;
      }
    const uint64_t round_end = std::min<uint64_t>((uint64_t)(b0 + (size_t)nc * chunk_bytes) * 8, stream_end_bits);
    for (size_t k = 0; k < live.size(); k++) {
      if (k + 1 < live.size()) live[k].want_end = live[k + 1].start;
      else { live[k].want_end = 0; live[k].min_end = round_end; }
    }
    parallel((int)live.size(), [&](int k) { run_chunk(s, live[(size_t)k]); });
    const auto r2 = now();
    for (size_t k = 0; k < live.size(); k++) {
      if (!live[k].ok) return false;
      if (live[k].final && k + 1 != live.size()) return false;
    }
    // ---- windows front to back, then every chunk narrows its symbols into the output
    std::vector<std::vector<uint8_t>> win(live.size());
    std::vector<size_t> off(live.size());
    size_t total = out.size();
    for (size_t k = 0; k < live.size(); k++) {
      win[k] = window;
      off[k] = total;
      const uint16_t *v = live[k].out.v;
      const size_t vn = live[k].out.n;
      total += vn;
      const size_t take = std::min<size_t>(vn, WSIZE);
      std::vector<uint8_t> nw((size_t)WSIZE);
      if (take < (size_t)WSIZE) memcpy(nw.data(), window.data() + take, WSIZE - take);
      for (size_t i = 0; i < take; i++) {
        const uint16_t x = v[vn - take + i];
        nw[WSIZE - take + i] = x < 256 ? (uint8_t)x : window[(size_t)(x - 256)];
      }
      window.swap(nw);
    }
    if (!out.resize(total)) return false;
    std::vector<std::vector<uLong>> crcs(live.size());          // per chunk: CRC-32 of every piece between member ends
    parallel((int)live.size(), [&](int k) {
      const uint16_t *v = live[(size_t)k].out.v;
      const size_t vn = live[(size_t)k].out.n;
      const uint8_t *w = win[(size_t)k].data();
      uint8_t *o = (uint8_t *)&out[0] + off[(size_t)k];
      size_t i = 0;
#if defined(__SSE2__)
      // sixteen symbols at a time: all of them bytes (no marker among them, the rule past a chunk's first window) -> one pack and store
      const __m128i zero = _mm_setzero_si128();
      for (; i + 16 <= vn; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i *)(v + i)), b = _mm_loadu_si128((const __m128i *)(v + i + 8));
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_srli_epi16(_mm_or_si128(a, b), 8), zero)) == 0xFFFF)
          _mm_storeu_si128((__m128i *)(o + i), _mm_packus_epi16(a, b));
        else
          for (size_t q = i; q < i + 16; q++) { const uint16_t x = v[q]; o[q] = x < 256 ? (uint8_t)x : w[x - 256]; }
      }
#endif
      for (; i < vn; i++) { const uint16_t x = v[i]; o[i] = x < 256 ? (uint8_t)x : w[x - 256]; }
      const std::vector<MemberEnd> &ends = live[(size_t)k].ends;
      size_t from = 0;
      for (size_t e = 0; e <= ends.size(); e++) {
        const size_t to = e < ends.size() ? ends[e].at : vn;
        crcs[(size_t)k].push_back((uLong)crc32_fast(0u, o + from, to - from));
        from = to;
      }
    });
    // every member must agree with its own trailer (length mod 2^32 and CRC-32), or the file goes to the serial inflater
    for (size_t k = 0; k < live.size(); k++) {
      const std::vector<MemberEnd> &ends = live[k].ends;
      size_t from = 0;
      for (size_t e = 0; e <= ends.size(); e++) {
        const size_t to = e < ends.size() ? ends[e].at : live[k].out.n;
        crc = crc32_combine(crc, crcs[k][e], (z_off_t)(to - from));
        member_len += to - from;
        if (e < ends.size()) {
          if ((uint32_t)crc != ends[e].crc || (uint32_t)member_len != ends[e].isize) { out.clear(); return false; }
          crc = crc32(0L, Z_NULL, 0); member_len = 0;
        }
        from = to;
      }
    }
    pos = live.back().end;
    finished = live.back().final;
    if (progress && *progress) (*progress)(total, (size_t)(pos >> 3));
    t_find += ms(r0, r1); t_dec += ms(r1, r2); t_res += ms(r2, now()); rounds++;
    if (!finished && pos >= stream_end_bits) return false;
  }
  if (trace) fprintf(stderr, "[itsx] parallel inflate: %d rounds, find %.0f ms, decode %.0f ms, resolve+crc %.0f ms\n", rounds, t_find, t_dec, t_res);
  // the last member ended at the end of the file and every member agreed with its trailer
  if ((pos >> 3) != n || member_len != 0) { out.clear(); return false; }
  return true;
}

}  // namespace itsx_io
