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
#include <cstdint>
#include <memory>
#include <string>

namespace itsx_io {

enum Kind { PLAIN = 0, GZIP = 1, ZSTD = 2 };

// Returns the decompressed content or nullptr (err set).  cacheable: keep / look up the text in the process-wide cache.
std::shared_ptr<const std::string> read_text(const char *path, std::string &err, bool cacheable);
void cache_clear();
// Adopt `text` as the content of the file just written at `path` (temporary files the next stage reads back).
void cache_put(const char *path, std::shared_ptr<const std::string> text);

int io_threads();               // ITSX_IO_THREADS or min(hardware threads, 32)

// pinflate.cpp: block-parallel inflate of a single-member gzip buffer (data[n .. n+16) must be readable).  true = `out`
// holds the content and its length and CRC-32 matched the trailer; false = not applicable or any doubt: inflate serially.
bool gunzip_parallel(const char *data, size_t n, std::string &out, int threads);
int64_t parallel_inflates();    // files the block-parallel inflater has delivered since the library was loaded

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

// which codecs are live in this process (for tests / logs): bit 0 libdeflate, bit 1 libzstd
int codec_flags();

}  // namespace itsx_io
