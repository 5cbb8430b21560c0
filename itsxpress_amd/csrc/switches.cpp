// switches.cpp -- the registry behind switches.h: name, class, one line of meaning.  INTEGRATION.md section 7 is generated from this
// table (scripts/switch_table.py) and tests/test_oracle_cpu.py checks that every sw_get("...") in csrc/ names an entry.
#include "switches.h"
#include <cstdlib>
#include <cstring>

namespace itsx {

static const Switch g_sw[] = {
  // ---- HOOKs: ignored unless ITSX_TEST_HOOKS=1
  {"ITSX_NO_ENSEMBLE", SW_HOOK, "multidomain regions keep their region as one envelope instead of hmmsearch's stochastic-traceback clustering (changes envelopes)"},
  {"ITSX_LAZY_ZUB_SCALE", SW_HOOK, "multiplies the lazy stage's upper bound on domZ: more undecided rows for the completion tests"},
  {"ITSX_LAZY_FORCE_PENDING", SW_HOOK, "treats rows as undecided so that the completion path runs"},
  {"ITSX_LAZY_NO_RERUN", SW_HOOK, "skips the full re-search safety net (leaves undecided rows undecided)"},
  {"ITSX_LAZY_NO_COMPLETE", SW_HOOK, "skips itsx_lazy_complete inside itsx_search_finalize (leaves undecided rows undecided)"},
  {"ITSX_COMPACT_ZMAX", SW_HOOK, "largest domZ the compact / lazy modes call a row 'certain' for (default 1e9); a value below the data's domZ can drop a winner"},
  {"ITSX_COMPACT_DOME_MIN", SW_HOOK, "smallest --domE the compact / lazy modes call a row 'certain' for (default 1e-2)"},
  {"ITSX_PASSA_DBG", SW_HOOK, "pass A diagnostic bits (no rows / no restore / no join): timings only, scores are garbage"},
  // ---- MODEs
  {"ITSX_ROWS", SW_MODE, "rows mode when the caller did not call itsx_set_rows_mode: full | compact | lazy (coordinates identical in all three)"},
  {"ITSX_COMPACT_ROWS", SW_MODE, "=1: rows mode compact (older spelling of ITSX_ROWS=compact)"},
  {"ITSX_QMASK", SW_MODE, "=none: vsearch's --qmask none for cluster_size / orient seeds (default dust, as vsearch)"},
  {"ITSX_KEEP_TRACE", SW_MODE, "keeps per-pair filter traces for itsx_get_pairtraces (forces the compact path of a lazy search)"},
  {"ITSX_GZIP_LEVEL", SW_MODE, "deflate level of the trimmed-FASTQ writers (default 6; output bytes differ, records do not)"},
  // ---- DIAGnostics
  {"ITSX_TEST_HOOKS", SW_DIAG, "=1: honour the HOOK switches above"},
  {"ITSX_SHARE_CHECK", SW_DIAG, "runs the unshared kernels beside the shared ones and counts differences (itsx_stats.share_mismatch, join_maxdiff)"},
  {"ITSX_LAZY_CHECK_BOUND", SW_DIAG, "compares pass A's scores with HMMER's own Forward arithmetic (itsx_stats.lazy_bound_maxdiff)"},
  {"ITSX_LAZY_HIST", SW_DIAG, "prints the histogram of round-2 candidates per group"},
  {"ITSX_PASSA_DUMP", SW_DIAG, "prints every pass-A launch's waves and wave-rows (scripts/passa_launches.py)"},
  {"ITSX_TRACE_ALLOC", SW_DIAG, "prints every hipMalloc / free with its time"},
  {"ITSX_CL_DEBUG", SW_DIAG, "clustering: prints per-window counters"},
  {"ITSX_MR_DEBUG", SW_DIAG, "ensemble stage: prints per-batch counters"},
  // ---- TUNING (result-neutral)
  {"ITSX_SHARE", SW_TUNING, "=0: no prefix sharing (round 4's schedule)"},
  {"ITSX_SHARE_TWO", SW_TUNING, "=0: one-sided (prefix-only) sharing for pass A"},
  {"ITSX_MSV_TWO", SW_TUNING, "=0: the MSV filter keeps the one-sided (prefix-only) schedule when pass A shares two-sidedly"},
  {"ITSX_SHARE_B", SW_TUNING, "rows per block of the prefix / suffix trees (default 32)"},
  {"ITSX_SHARE_GB", SW_TUNING, "budget of the saved row states, GB"},
  {"ITSX_SHARE_MIN", SW_TUNING, "smallest shared-row fraction for which the shared schedule is used (default 0.10)"},
  {"ITSX_SHARE_FWD_STREAMS", SW_TUNING, "1 | 2 streams for pass A's batches"},
  {"ITSX_SHARE_MSV_STREAMS", SW_TUNING, "1 | 2 streams for the MSV filter's batches"},
  {"ITSX_NO_CHAINREC", SW_TUNING, "pass A reads a chain's data through the round-5 arrays instead of its 64-byte record (A/B)"},
  {"ITSX_BOUND_FOLD", SW_TUNING, "=0: pass A's plain recurrences instead of the folded ones (also switches two-sided sharing off)"},
  {"ITSX_BOUND_RESCALE_EXP", SW_TUNING, "power of ten at which pass A rescales a row (default 20)"},
  {"ITSX_LAZY_TOPUP", SW_TUNING, "=0: undecided rows go straight to the full count of their profiles (no top-up round); =2: the round also looks for unreported pairs among a profile's weakest (rows settled from above)"},
  {"ITSX_LAZY_EXACT_BOUND", SW_TUNING, "pass A through the HMMER-order Forward kernel (A/B)"},
  {"ITSX_CHUNK_UNIQUES", SW_TUNING, "representatives per search chunk"},
  {"ITSX_MSV_OVERLAP", SW_TUNING, "=0: the next chunk's MSV filter does not run beside the domain stage"},
  {"ITSX_MSV_PAD", SW_TUNING, "dynamic LDS of the overlapped MSV launch (an occupancy cap)"},
  {"ITSX_MSV_WHOLE", SW_TUNING, "=0: MSV stages 16 words at a time even when the whole read fits LDS"},
  {"ITSX_BIAS_OVERLAP", SW_TUNING, "=0: the bias filter does not run on the second stream"},
  {"ITSX_ST2_PRIO", SW_TUNING, "priority of the second stream"},
  {"ITSX_LOAD_PRIORITY", SW_TUNING, "=0: loads do not use the high-priority stream"},
  {"ITSX_SLAB_GB", SW_TUNING, "DP slab budget, GB"},
  {"ITSX_SLAB_ADAPT", SW_TUNING, "=0: the slab budget does not adapt to the job"},
  {"ITSX_DEFER_FREE_GB", SW_TUNING, "device memory kept on the deferred-free list, GB"},
  {"ITSX_MR_LONG_FRAC", SW_TUNING, "ensemble stage: share of regions taken as 'long'"},
  {"ITSX_MR_LONG_LANES", SW_TUNING, "ensemble stage: lanes per long region"},
  {"ITSX_MSV_BWD_WHOLE", SW_TUNING, "=1: the MSV filter's Backward chains copy their whole read to LDS first (A/B arm: same cells)"},
  {"ITSX_LAZY_TOPUP_ALL", SW_TUNING, "=0: a profile whose undecided rows need most of its pairs is left to the full count (itsx_lazy_complete) instead of taking all its unevaluated pairs in the top-up round (same rows)"},
  {"ITSX_MR_WAVES", SW_TUNING, "ensemble stage: waves a batch is spread over (lanes per wave = regions / this, 2 .. 64); 0 = waves of 64 lanes"},
  {"ITSX_MR_ONE_MAX", SW_TUNING, "ensemble stage: a batch of at most this many regions (default 2048) is walked one region per wave with the matrix in LDS; 0 = never"},
  {"ITSX_CL_NOSCORE", SW_TUNING, "clustering: skips the score-only pre-pass (A/B arm; the walk then aligns every candidate: same outcomes)"},
  {"ITSX_CL_NOPRECHECK", SW_TUNING, "clustering: skips the certificate pre-check (A/B arm: same outcomes)"},
  {"ITSX_CL_CAPACITY", SW_TUNING, "clustering: candidate-list capacity"},
  {"ITSX_CL_CCAP", SW_TUNING, "clustering: centroid capacity step"},
  {"ITSX_CL_HEAVY", SW_TUNING, "clustering: heavy-word split on / off"},
  {"ITSX_CL_HEAVY_MIN", SW_TUNING, "clustering: centroids a word needs to count as conserved"},
  {"ITSX_CL_ROWS", SW_TUNING, "clustering: alignment rows per lane"},
  {"ITSX_CL_WINDOW", SW_TUNING, "clustering: queries per speculative window"},
  {"ITSX_PACK_CHUNK", SW_TUNING, "reads per packing launch"},
  {"ITSX_PACK_ECAP", SW_TUNING, "initial capacity of the exception list"},
  {"ITSX_HUGEPAGES", SW_TUNING, "=0: host buffers without MADV_HUGEPAGE"},
  {"ITSX_IO_LIBDEFLATE", SW_TUNING, "=0: zlib instead of libdeflate when both are present"},
  {"ITSX_IO_THREADS", SW_TUNING, "host threads of the FASTQ readers / writers"},
  {"ITSX_IO_BLOCK_KB", SW_TUNING, "deflate unit of the block-parallel writer"},
  {"ITSX_PARALLEL_INFLATE", SW_TUNING, "=0: single-stream inflate"},
  {"ITSX_PINFLATE_CHUNK_KB", SW_TUNING, "chunk of the block-parallel inflater"},
  {"ITSX_PARSE_MIN_MB", SW_TUNING, "smallest text parsed on several threads"},
  {"ITSX_TEXT_CACHE_GB", SW_TUNING, "inflated-text cache, GB"},
  {"ITSX_WRITE_MIN_MB", SW_TUNING, "smallest output written by several threads"},
  {"ITSX_WRITE_UNIT_KB", SW_TUNING, "unit of the streamed writer"},
  {"ITSX_STREAM_RESERVE_MB", SW_TUNING, "streamed loads: first reservation"},
  {"ITSX_STREAM_RESERVE_X", SW_TUNING, "streamed loads: growth factor"},
};
static const int g_nsw = (int)(sizeof(g_sw) / sizeof(g_sw[0]));

const Switch *sw_registry(int *n) { if (n) *n = g_nsw; return g_sw; }

static bool hooks_on() { const char *e = sw_get("ITSX_TEST_HOOKS"); return e && atoi(e) == 1; }

const char *sw_get(const char *name)
{
  const char *v = getenv(name);
  if (!v) return nullptr;
  for (int i = 0; i < g_nsw; i++)
    if (strcmp(g_sw[i].name, name) == 0) return (g_sw[i].kind == SW_HOOK && !hooks_on()) ? nullptr : v;
  return nullptr;            // not in the registry: not a switch of this library
}

std::string sw_report()
{
  std::string out;
  const bool on = hooks_on();
  for (int i = 0; i < g_nsw; i++) {
    const char *v = getenv(g_sw[i].name);
    if (!v) continue;
    out += g_sw[i].name; out += "="; out += v;
    if (g_sw[i].kind == SW_HOOK && !on) out += " (ignored: ITSX_TEST_HOOKS is not 1)";
    out += "\n";
  }
  return out;
}

}  // namespace itsx
