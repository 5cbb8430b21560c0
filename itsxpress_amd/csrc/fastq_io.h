// fastq_io.h -- host-side file plumbing shared by the engine's loaders and the FASTQ writers.
//
// Replaces, for the files either side of the hot path, what the reference does with gzip.open / pyzstd.open + Biopython
// (itsxpress/SeqSample.py:886-949 writes through a temporary plain file and one compression stream; main.py:296-330
// sniffs .gz / .zst by extension).  Host-only; no GPU call in here.
//
//  * read_text: a whole file, decompressed by its magic bytes (gzip incl. concatenated members, zstd incl. concatenated
//    frames, or plain).  The decompressed text of the most recently read files is kept in a small process-wide cache
//    (bounded, see ITSX_TEXT_CACHE_GB), so the trimmed-FASTQ writer does not inflate the file the loader has just
//    inflated: inflating is the largest serial cost of a file-to-file run.
//  * BlockWriter: plain, gzip or zstd output.  Compressed output is cut into blocks that a pool of threads compresses
//    independently and that are written in order as concatenated gzip members / zstd frames: standard files any
//    gzip / zstd reader accepts, the decompressed bytes are exactly what one stream would give.
//  * gzip work goes through libdeflate when the shared object is present (2-4x zlib on FASTQ), else zlib; zstd is used
//    through libzstd.so.1's stable one-shot / streaming entry points (no headers for either in this image, so both are
//    bound at run time; absent libzstd = a loud error for .zst files only).
#pragma once
#include "switches.h"
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>

namespace itsx_io {

enum Kind { PLAIN = 0, GZIP = 1, ZSTD = 2 };

// A file's bytes.  The part of std::string's interface the loaders and writers use, over memory that is NOT value-initialised when
// it grows: a 9-GB text filled front to back by a pool of inflating threads must not be zeroed first by one of them (1.5 s of a
// 5.5-s inflate).  Large buffers are anonymous mappings (pages arrive zeroed from the kernel on the first touch, by whichever
// thread touches them; growth is an mremap, not a copy); small ones are malloc'd.  adopt() takes over a std::string without a copy.
class Text {
 public:
  Text() = default;
  ~Text() { release(); }
  Text(const Text &) = delete;
  Text &operator=(const Text &) = delete;
  const char *data() const { return p_; }
  char *data() { return p_; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool empty() const { return n_ == 0; }
  char operator[](size_t i) const { return p_[i]; }
  char &operator[](size_t i) { return p_[i]; }
  bool reserve(size_t cap);           // false: out of memory (content kept)
  // a SHARED mapping of a (new) file of `cap` bytes instead of anonymous memory: other processes map the same file and read the text
  // where the inflater puts it (the multi-GPU driver's workers: no copy of their piece).  The file is sparse; the caller unlinks it.
  bool reserve_file(const char *path, size_t cap);
  bool resize(size_t n);              // new bytes are unspecified (zero only where the kernel has just supplied the page)
  void clear() { n_ = 0; }
  void shrink_to_fit();
  void swap(Text &o);
  void adopt(std::string &&s);
  void borrow(const char *q, size_t m) { release(); p_ = const_cast<char *>(q); n_ = cap_ = m; kind_ = 4; }   // a view: never written, never freed
  // pin: the bytes stay where they are -- growth past the capacity FAILS instead of moving them (readers hold pointers into a
  // buffer that is still being filled: TextStream)
  void pin(bool on) { pinned_ = on; }
  bool append(const char *q, size_t m) { const size_t at = n_; if (!resize(n_ + m)) return false; if (m) __builtin_memcpy(p_ + at, q, m); return true; }
  bool append(const char *b, const char *e) { return append(b, (size_t)(e - b)); }
  bool assign(const char *q, size_t m) { n_ = 0; return append(q, m); }
  std::string substr(size_t o, size_t m) const { return std::string(p_ + o, m); }
 private:
  void release();
  char *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
  int kind_ = 0;                      // 0 nothing, 1 malloc, 2 mmap, 3 the adopted string, 4 borrowed, 5 a shared file mapping (never grows)
  bool pinned_ = false;
  std::string own_;
};

// Returns the decompressed content or nullptr (err set).  cacheable: keep / look up the text in the process-wide cache.
std::shared_ptr<const Text> read_text(const char *path, std::string &err, bool cacheable);
void cache_clear();
// CRC-32 (the gzip polynomial), continuing from `crc` (0 to start): libdeflate's when the library is there, else zlib's
uint32_t crc32_fast(uint32_t crc, const void *p, size_t n);
// Adopt `text` as the content of the file just written at `path` (temporary files the next stage reads back).
void cache_put(const char *path, std::shared_ptr<const Text> text);

int io_threads();               // ITSX_IO_THREADS or min(hardware threads, 32)

// pinflate.cpp: block-parallel inflate of a single-member gzip buffer (data[n .. n+16) must be readable).  true = `out`
// holds the content and its length and CRC-32 matched the trailer; false = not applicable or any doubt: inflate serially.
// progress (may be null): called after every round with the number of bytes of `out` that are final (front to back).
// (second argument: compressed bytes consumed so far -- a driver that deals the text out in N even pieces estimates its final size from the two)
bool gunzip_parallel(const char *data, size_t n, Text &out, int threads, const std::function<void(size_t, size_t)> *progress = nullptr);
int64_t parallel_inflates();    // files the block-parallel inflater has delivered since the library was loaded

// A file's text delivered front to back WHILE it is being inflated: the loader of a large .fastq.gz hands record-aligned slices
// to the GPU stages as the block-parallel inflater finishes its rounds, instead of after the last byte (itsx_stream_* in
// include/itsx_hip.h).  Plain, zstd and small files arrive in one piece.  The text ends up in the cache like read_text's.
struct StreamImpl;
class TextStream {
 public:
  TextStream();
  ~TextStream();
  // shared_backing (may be null): the text is inflated into a shared mapping of that (new, sparse) file, so that other processes can
  // map the slices they are told about; a plain (uncompressed) input is not copied there -- *plain_input says so and the caller maps the
  // input file itself (slices are offsets into it all the same)
  // threads (0: io_threads()): the inflater's pool -- a paired run's two streams share the CPUs
  bool open(const char *path, std::string &err, const char *shared_backing = nullptr, bool *plain_input = nullptr, int threads = 0);
  // text bytes that are final, compressed bytes consumed, compressed size (equal pairs once the file is done)
  void progress(size_t *avail, size_t *consumed, size_t *raw_size);
  const char *base() const;
  // Blocks until at least min_bytes past the last slice are final (or the file ends); the slice is cut at a FASTQ record start.
  // false: the file could not be delivered (err); *last: nothing follows this slice.
  bool next(size_t min_bytes, const char **ptr, size_t *nbytes, bool *last, std::string &err);
  // The mate file of a paired run: the next slice holds exactly n_records FASTQ records (4 n lines) -- blocks until that many are
  // final; fewer only at the end of the file (*last).  *got = records in the slice.
  bool next_records(size_t n_records, const char **ptr, size_t *nbytes, size_t *got, bool *last, std::string &err);
  // joins the inflater; keep: the text goes to the cache under the file's path.  Slices stay valid until the TextStream dies.
  bool finish(bool keep, std::string &err);
  // an upper bound of the number of records in the WHOLE text (lines / 4 for FASTQ, / 2 otherwise, + 1), or -1 while the text is
  // still being inflated: what a streaming driver needs to bound the counts of the chunks it has not seen yet
  long long records_bound();
 private:
  StreamImpl *s;
};

// first FASTQ record start at or after `from` in t[0, n) (a line that starts with '@' whose second line below starts with '+'); n: none
size_t fastq_record_start(const char *t, size_t n, size_t from);

struct WriterImpl;
class BlockWriter {
 public:
  BlockWriter();
  ~BlockWriter();
  // keep_text (PLAIN only): the bytes written are also kept and handed to the text cache when the file is closed
  bool open(const char *path, int kind, std::string &err, bool keep_text = false);
  void put(const char *p, size_t n);
  void put(const std::string &s) { put(s.data(), s.size()); }
  bool close(std::string &err);   // flushes; false when any block failed to compress or write
 private:
  WriterImpl *w;
};

// One piece of output as an independent gzip member / zstd frame (PLAIN: the bytes themselves), with the BlockWriter's codecs and
// level: what a writer that orders its pieces itself (trim_host.cpp: itsx_twriter_*) runs on its own threads.  One per thread.
class PieceCompressor {
 public:
  explicit PieceCompressor(int kind);
  ~PieceCompressor();
  bool ok() const { return ok_; }
  bool run(const std::string &in, std::string &out);
 private:
  int kind_, level_; void *ldc_ = nullptr; bool ok_ = true;
};

// which codecs are live in this process (for tests / logs): bit 0 libdeflate, bit 1 libzstd
int codec_flags();

}  // namespace itsx_io
