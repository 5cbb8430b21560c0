/*
 * itsx_hip.h -- C ABI of the MI355X (gfx950) ITS-trimming engine.
 *
 * Drop-in boundary for ONE path of USDA-ARS-GBRU/itsxpress: dereplicate reads ->
 * score the representatives against the ITSx profile HMMs -> per-read trim
 * coordinates.  In the reference this path is three Python methods that shell
 * out to vsearch / hmmsearch and two parsers that re-read their output files.
 * Every entry point below names the reference interface it replaces.
 *
 * Conventions: plain C types only; every call returns 0 on success or a negative
 * ITSX_E_* code, with text available from itsx_last_error(); the caller owns all
 * output arrays; nothing throws across the boundary.  One context drives one GPU
 * (one process per GPU); a context is not thread-safe.  There is no CPU fallback:
 * itsx_create fails if no gfx950 device is usable.
 */
#ifndef ITSX_HIP_H
#define ITSX_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITSX_ABI_VERSION 6      /* 6: two-sided sharing (itsx_stats grew); 5: prefix sharing (itsx_stats grew); 4: streaming loads (itsx_stream_*, itsx_keyset_*, itsx_load_reads_text), itsx_io_cache_clear; 3: rows modes (itsx_set_rows_mode, the lazy domain stage), itsx_stats grew; 2: itsx_get_stats / itsx_get_pairtraces take the caller's struct size */

enum {
  ITSX_OK            =  0,
  ITSX_E_ARG         = -1,   /* bad argument / call order */
  ITSX_E_IO          = -2,   /* file could not be read or written */
  ITSX_E_FORMAT      = -3,   /* malformed HMMER3/f or FASTA/FASTQ text, illegal residue */
  ITSX_E_DEVICE      = -4,   /* HIP error or no usable device */
  ITSX_E_UNSUPPORTED = -5,   /* e.g. model longer than 46 nodes, read longer than 65535 bases */
  ITSX_E_COLLISION   = -6,   /* 64-bit hash collision survived every reseed (never observed) */
  ITSX_E_NOMEM       = -7
};

typedef struct itsx_ctx itsx_ctx;
typedef struct itsx_stream itsx_stream;   /* a file's text, handed out while it is being inflated (itsx_stream_*) */
typedef struct itsx_keyset itsx_keyset;   /* the sequences seen so far in a streaming run (itsx_keyset_*) */
typedef struct itsx_twriter itsx_twriter; /* a trimmed-FASTQ output that takes text and coordinates piece by piece (itsx_twriter_*) */

/* One reported-or-not domain, in hmmsearch --domtblout row order (profile file order,
 * then target index, then domain index).  Replaces one text row of domtbl.txt as
 * consumed by ItsPosition.parse (itsxpress/SeqSample.py:431-461): ll[0]=target name
 * -> rep, ll[2]=tlen, ll[3]=query name -> prof, ll[13]=domain score -> bitscore,
 * ll[19]/ll[20] = env from/to -> ienv/jenv. */
typedef struct {
  int64_t rep;            /* index of the representative in the unique list (see itsx_get_uniques) */
  int32_t prof;           /* profile index in file order */
  int32_t tlen;
  int32_t ienv, jenv;     /* 1-based envelope coordinates */
  int32_t dom_idx, ndom;
  int32_t flags;          /* bit0: the envelope was defined by stochastic traceback clustering of a multidomain region
                           * (hmmsearch's region_trace_ensemble); with ITSX_NO_ENSEMBLE=1: such a region kept as ONE envelope */
  float   envsc;          /* nats */
  float   domcorrection;  /* nats */
  float   dombias;        /* nats */
  float   bitscore;       /* bits (column 14 of domtblout, before %.1f) */
  double  lnP;
  float   seq_score;      /* bits (column 8) */
  float   seq_bias;       /* bits */
  int32_t seq_reported;   /* per-sequence score >= T */
  int32_t dom_reported;   /* exp(lnP) * domZ <= domE, set by itsx_search_finalize */
} itsx_domain;

/* per (representative, profile) filter trace for parity tests; only pairs past the MSV filter */
typedef struct {
  int64_t rep; int32_t prof;
  int32_t msv_xj;         /* final xJ byte of the MSV filter; 255 = overflow (score +inf) */
  int32_t pass_msv, pass_bias, pass_fwd;
  float   msv_sc, filtersc, fwdsc, bcksc, nullsc;
  int32_t nregions, ndom;
  int32_t ran_vit, pass_vit;   /* Viterbi filter: runs only for pairs whose bias-corrected MSV P-value exceeds F2 (none when F1 == F2) */
  float   vitsc;
  int32_t pad;
} itsx_pairtrace;

typedef struct {
  int64_t n_reads, n_unique, n_dropped_short;
  int64_t n_pairs, n_past_msv, n_past_bias, n_past_fwd, n_regions, n_multidomain, n_domains;
  int64_t n_domain_overflow;      /* (rep,profile) pairs with more than 8 regions: the ones past the pair's slots live in an overflow list (none is dropped) */
  int32_t n_profiles, hash_reseeds;
  /* device time of the last call of each stage, milliseconds (HIP events on the engine's stream) */
  float   ms_derep, ms_msv, ms_filters, ms_domains, ms_finalize;
  /* dominant-kernel accounting for bench.py's roofline block */
  float   ms_msv_kernel;  int64_t msv_cells;  int64_t msv_launches;
  float   ms_fwd_kernel, ms_bwd_kernel;  int64_t fwd_rows;   /* lane-rows: sum of target lengths over surviving pairs */
  float   ms_env_kernel;  float ms_bias_kernel;  int64_t env_rows;    /* the three envelope sweeps together; lane-rows per sweep */
  int64_t n_env_unique;           /* distinct (profile, length, envelope subsequence) actually re-scored */
  float   ms_decode_kernel;  int32_t n_batches;          /* launches of each DP kernel in the last search */
  float   ms_cluster;        int32_t pad0;               /* itsx_cluster at id < 1: whole call */
  int64_t cl_windows, cl_cuts, cl_alignments;            /* speculative windows, windows cut by validation, alignments */
  float   ms_merge;          int32_t pad1;               /* k_merge of the last itsx_merge_* call */
  int64_t cl_certified;                                  /* candidate alignments proven rejections without the full DP */
  float   ms_pack;           int32_t pad2;               /* last hand-over of reads: staging + upload + device packing, host wall time */
  /* parity-risk counters of the last itsx_trim_coords / itsx_rep_coords call (the prefixes decide which domains count):
   * uniques / reads whose winning left or right domain came from a region hmmsearch resolves by stochastic clustering
   * (itsx_domain.flags bit 0), and uniques / reads with a (representative, profile) pair that had more regions than
   * the engine keeps (bit 1) among the domains of the two sides */
  int64_t n_uniq_multi_winner, n_reads_multi_winner, n_uniq_region_cap, n_reads_region_cap;
  /* multidomain regions of the last search: resolved by stochastic traceback clustering (k_ensemble.hip), of which
   * n_mr_failed hit a bookkeeping limit or an unsampleable matrix and yielded nothing; envelopes the clustering defined */
  int64_t n_mr_clustered, n_mr_failed, n_mr_envelopes;
  float   ms_ensemble;       int32_t pad3;
  int64_t n_mr_distinct;     /* distinct (profile, target length, residues) multidomain regions actually sampled */
  int64_t n_slab_shrinks;    /* times the DP slab budget was halved because the device could not supply it */
  float   ms_vit_kernel;     int32_t pad4;               /* Viterbi filter (F2 < F1 only) */
  /* multidomain regions of the last search whose Forward matrix could not be sampled, by kind: [1] probabilities not normalised,
   * [5] a path left the region -- the cases in which hmmsearch's own stochastic traceback throws; itsx_search then returns
   * ITSX_E_UNSUPPORTED.  The former bookkeeping limits ([2] more than 8 domains in one sampled path, [4] more than 512 distinct
   * sampled tuples, [7] more than 4 envelopes in one region) no longer fail anything: such regions take the ensemble stage's overflow
   * path (n_mr_overflow), and a pair's regions past its 8 slots an overflow list (n_domain_overflow counts those pairs). */
  int64_t n_mr_fail_kind[8];
  /* domain rows (80 B each) resident on the device after the last search: all of them (= n_domains plus segment padding), or, with
   * ITSX_COMPACT_ROWS=1, only the rows that can still win ItsPosition's argmax whatever the dataset-wide domZ turns out to be */
  int64_t n_rows_resident;
  /* the lazy domain stage (itsx_set_rows_mode(ctx, ITSX_ROWS_LAZY)) of the last search: pairs that went through Backward / decoding /
   * envelopes (of n_past_msv), of which round 1 took the best-bound pair of each (representative, class); rows whose reporting
   * depended on the exact domZ and could have changed a result (then the search was repeated in full: n_lazy_reruns);
   * lane-rows and launches of the score-only Forward pass, its time and the selection's */
  int32_t lazy;              int32_t n_bound_launches;
  int64_t n_lazy_pending_profiles;                      /* distinct profiles among the undecided rows that mattered */
  int64_t n_lazy_completed, n_lazy_completed_profiles;  /* itsx_lazy_complete: pairs of the profiles counted exactly, and those profiles */
  int64_t n_mr_overflow;     /* distinct multidomain regions that went through the ensemble stage's overflow path (per-region arrays sized by the region) */
  float   ms_lazy_complete;  float lazy_bound_maxdiff;  /* ITSX_LAZY_CHECK_BOUND=1 (tests): largest |bound kernel - HMMER-order Forward| of the last search, nats */
  int64_t n_lazy_evaluated, n_lazy_round1, n_lazy_pending, n_lazy_reruns, bound_rows;
  float   ms_bound_kernel, ms_lazy_select;
  /* prefix sharing (csrc/k_share.hip) of the last search: rows per block of the prefix tree (0: off); saved row states and chains that
   * start from one; lane-rows the MSV filter and the lazy stage's score-only Forward pass computed (msv_rows, bound_rows) against the
   * rows of their pairs (msv_rows_full, bound_rows_full); pairs that ran in the Forward pass only for a chain below them;
   * ITSX_SHARE_CHECK=1 (tests): results that differ from the unshared kernels' (MSV cells, Forward scores; must be 0) */
  int32_t share_B;           int32_t share_batches;
  int64_t share_nodes, share_chains, msv_rows, msv_rows_full, bound_rows_full, n_share_helpers, share_mismatch;
  float   ms_share_build;    float share_frac;          /* building the tree and the order, ms; rows the chains skip / rows of all uniques */
  /* two-sided sharing (round 6; lazy searches): a Forward chain stops where its SUFFIX is one an earlier representative of its length
   * ends with too, and takes the rest of the sum over paths from the Backward state that representative's chain saved there.
   * n_joined: representatives that join; bwd_chains / gamma_nodes: Backward chains and their saved states; per representative (one
   * profile) two_fwd_rows + two_bwd_rows rows are walked of two_rows_full; bwd_rows: lane-rows of the Backward chains of the last
   * search (bound_rows: the Forward chains'); ITSX_SHARE_CHECK=1 (tests): join_maxdiff = largest |joined score - unshared score|, nats
   * (more than 2e-3 counts into share_mismatch) */
  int32_t two_sided;         int32_t n_bwd_launches;
  int64_t n_joined, bwd_chains, gamma_nodes, bwd_rows, two_fwd_rows, two_bwd_rows, two_rows_full;
  float   join_maxdiff;      float ms_bwd_bound;        /* the Backward chains' kernel time (part of ms_bound_kernel) */
  /* the top-up round of itsx_search_finalize (round 6): pairs it evaluated -- the best-bound unevaluated pairs of the profiles with
   * undecided rows, as many as those rows need for the lower bound on domZ to decide them -- and its time (part of ms_finalize);
   * n_lazy_completed stays 0 when it settles everything.  ms_lazy_topup_stages = the part of ms_lazy_topup spent in the domain pipeline:
   * that part is ALSO accumulated into ms_filters / ms_domains, like the stages itsx_lazy_complete re-runs (it took the place of a
   * padding word: the structure's size and every other offset are unchanged) */
  int64_t n_lazy_topup;      float ms_lazy_topup;       float ms_lazy_topup_stages;
} itsx_stats;

int         itsx_abi_version(void);
/* itsx_last_error(NULL) returns the message of the last failed itsx_create. */
const char *itsx_last_error(const itsx_ctx *ctx);

/* Replaces: process start-up of the vsearch / hmmsearch subprocesses
 * (itsxpress/SeqSample.py:117,162,210).  device_id = HIP ordinal; flags = 0. */
itsx_ctx   *itsx_create(int device_id, int flags);
void        itsx_destroy(itsx_ctx *ctx);

/* ---- profiles: replaces hmmsearch reading `hmmfile` (itsxpress/SeqSample.py:207), the
 * HMMER3/f text written by create_runtime_hmm (itsxpress/main.py:176-231). */
int itsx_load_profiles_file(itsx_ctx *ctx, const char *hmm_path, int *n_profiles);
int itsx_load_profiles_mem(itsx_ctx *ctx, const char *text, int64_t len, int *n_profiles);
int itsx_profile_name(const itsx_ctx *ctx, int i, char *buf, int buflen);
/* table parity: copy out the configured MSV byte costs [18][M+1], striped odds ratios
 * rfv [18][Q][4] / tfv [8Q][4], and {M, Q, base, bias, tbm, tec}. */
int itsx_profile_tables(const itsx_ctx *ctx, int i, uint8_t *rbv, float *rfv, float *tfv, int32_t *params6);

/* ---- reads: replaces vsearch/hmmsearch reading self.seq_file / rep.fa
 * (itsxpress/SeqSample.py:109,208).  ASCII bases, IUPAC allowed, case-insensitive, U == T.
 * offsets has n+1 entries.  names may be NULL (then the writers emit r%09d).
 * The reads are packed 2 bits/base (+ an exception list for non-ACGT) and are resident in
 * HBM when the call returns. */
int itsx_set_reads(itsx_ctx *ctx, const char *bases, const int64_t *offsets, int64_t n,
                   const char *names, const int64_t *name_offsets);
/* The same without the host-side copy of the bases: the engine keeps the caller's `bases` pointer (itsx_write_rep_fasta
 * reads the representatives from it), so the caller keeps that buffer valid and unchanged until the next reads call on
 * this context or itsx_destroy.  offsets and names are copied as before. */
int itsx_set_reads_view(itsx_ctx *ctx, const char *bases, const int64_t *offsets, int64_t n,
                        const char *names, const int64_t *name_offsets);
/* The same for text that already lives in DEVICE memory (d_bases: a hipMalloc'ed buffer of ASCII bases on this
 * context's GPU, e.g. the output of a device-side parser or of the merge kernel); offsets and names are host arrays.
 * Nothing crosses PCIe but the offsets; the bases are packed where they are.  The caller keeps d_bases valid until the
 * next reads call or itsx_destroy (itsx_write_rep_fasta copies the text back on demand). */
int itsx_set_reads_device(itsx_ctx *ctx, const void *d_bases, const int64_t *offsets, int64_t n,
                          const char *names, const int64_t *name_offsets);
/* FASTA or FASTQ file, plain or gzip (.gz): native parser for the same input. */
int itsx_load_reads_file(itsx_ctx *ctx, const char *path, int64_t *n_reads);
/* One contiguous shard of the same file: records [n shard / n_shards, n (shard + 1) / n_shards) in file order, for a driver that
 * spreads one sample over several GPUs (itsxpress_amd/multi.py); n_total = records in the file, first = index of the shard's first. */
int itsx_load_reads_file_shard(itsx_ctx *ctx, const char *path, int32_t shard, int32_t n_shards, int64_t *n_total, int64_t *first, int64_t *n_reads);
/* the records of a piece of FASTA / FASTQ TEXT already in memory (a slice of itsx_stream_next); replaces the context's reads */
int itsx_load_reads_text(itsx_ctx *ctx, const char *text, int64_t nbytes, int64_t *n_reads);

/* ---- streaming file-to-file runs (host-only, context-free: csrc/stream_host.cpp; driver: itsxpress_amd/stream.py).
 * The reference inflates and parses the whole FASTQ before vsearch starts (itsxpress/main.py:534-554, SeqSample.py:93-131); on one
 * MI355X the inflater is the slower party, so a large .fastq.gz is cut into file-order chunks that are dereplicated and scored while
 * the rest is still being inflated.
 * itsx_stream_open: reads the file and starts inflating in the background (block-parallel for large gzip files; plain, zstd and
 *   small files are ready at once).  itsx_stream_next: BLOCKS until >= min_bytes of new text are final (or the file ends) and returns
 *   the next slice, cut at a FASTQ record start (non-FASTQ text arrives in one slice); *last = 1 on the final slice.  Slices stay
 *   valid until itsx_stream_close; keep_text != 0 leaves the whole text in the process's text cache under the file's path, where
 *   itsx_write_trimmed_fastq finds it.  A corrupt file is an error of the LAST call at the latest (CRC-32 and length of every gzip
 *   member are checked as in itsx_load_reads_file): a driver discards what it computed from earlier slices.
 * itsx_keyset_assign: tuples[n_unique][4] as itsx_unique_keys128 returns them for chunk `chunk`; verdict[n_unique][4] = per local
 *   unique (global index, orientation flag) of the sequence's FIRST occurrence so far and (chunk, local unique) of the holder that
 *   scores it -- the first holder, which in file order is vsearch's representative.  Same rows as the multi-GPU owner verdicts.
 *   gid[n_unique] (may be NULL): the sequence's number in first-seen order = its index in the global unique list.
 * Errors: negative code, text from itsx_stream_last_error(). */
int itsx_stream_open(const char *path, itsx_stream **out);
/* Round 6, for a multi-GPU driver (itsxpress_amd/multi.py; replaces the reference's whole-file read of itsxpress/main.py:295-330 for N
 * worker processes): the text is inflated into a SHARED mapping of `backing` (a new, sparse file the caller unlinks), so that a worker
 * maps the slice it is told about -- offset = slice address - itsx_stream_base(s) -- while later slices are still being inflated; no
 * copy of the pieces.  *plain_input = 1: the input is uncompressed and was not copied: the offsets are offsets into the input file.
 * itsx_stream_progress: text bytes that are final, compressed bytes behind them, the file's size (for an estimate of the whole
 * text's size before it is there: even pieces). */
int itsx_stream_open_threads(const char *path, int32_t threads, itsx_stream **out);     /* itsx_stream_open with a pool of `threads` inflating threads */
int itsx_stream_open_shared(const char *path, const char *backing, itsx_stream **out, int32_t *plain_input);
const char *itsx_stream_base(itsx_stream *s);
int itsx_stream_progress(itsx_stream *s, int64_t *avail, int64_t *consumed, int64_t *raw_size);
int itsx_stream_next(itsx_stream *s, int64_t min_bytes, const char **text, int64_t *nbytes, int32_t *last);
int itsx_stream_close(itsx_stream *s, int32_t keep_text);
/* Round 6, a streamed PAIRED sample (itsxpress/SeqSample.py:266-365, main.py:513-519: the reference merges the whole files first):
 * R1's slice comes from itsx_stream_next, itsx_count_records says how many records it holds, and the mate file's stream hands out
 * exactly that many with itsx_stream_next_records (fewer only at the end of its file: *got). */
int itsx_stream_next_records(itsx_stream *s, int64_t n_records, const char **text, int64_t *nbytes, int64_t *got, int32_t *last);
int64_t itsx_count_records(const char *text, int64_t nbytes);
/* an upper bound of the number of records in the whole file (lines / 4 for FASTQ), or -1 while it is still being inflated */
int64_t itsx_stream_records_bound(itsx_stream *s);
const char *itsx_stream_last_error(void);
/* ---- one file's records in pieces, for the workers of a multi-GPU driver (host-only, context-free: csrc/shard_host.cpp; driver:
 * itsxpress_amd/multi.py).  The reference hands the whole file to one vsearch process (itsxpress/SeqSample.py:93-131, 266-365).
 * itsx_shard_text: the file's text (plain / gzip / zstd, inflated once, kept in the process's text cache) cut into n_parts contiguous
 *   pieces at record starts near equal byte counts; piece p is written as the plain file out_prefix.<p> (use a /dev/shm prefix) and
 *   holds records[p] records (bytes[p] bytes; bytes may be NULL).  match_records != NULL (a mate file, R2 after R1): piece p is cut
 *   to hold exactly match_records[p] records instead; a file that does not hold them is ITSX_E_FORMAT.
 * Errors: negative code, text from itsx_shard_last_error(). */
/* one piece of a text in memory (a slice of itsx_stream_next, while the rest of the file is still being inflated) into a new file, by the
 * I/O pool: a multi-GPU driver's worker loads it with itsx_load_reads_file */
int itsx_write_range(const char *path, const char *text, int64_t nbytes);
int itsx_shard_text(const char *path, int32_t n_parts, const int64_t *match_records, const char *out_prefix, int64_t *records, int64_t *bytes);
/* The owner's step of the cross-shard dereplication for a multi-worker run (itsxpress_amd/multi.py: owner_verdicts states it in numpy):
 * recv[m][5] = (key0, key1, global index of the first occurrence, forward-is-canonical flag, local unique number) of the uniques whose
 * keys this worker owns, src[m] = the worker each row came from; out[m][4] = per row the global index and flag of its group's first
 * occurrence and the worker / local unique number of the holder that scores the sequence.  Host-only. */
int itsx_owner_verdicts(const int64_t *recv, const int64_t *src, int64_t m, int64_t *out);
const char *itsx_shard_last_error(void);
itsx_keyset *itsx_keyset_create(void);
void itsx_keyset_destroy(itsx_keyset *k);
int64_t itsx_keyset_size(const itsx_keyset *k);
int itsx_keyset_assign(itsx_keyset *k, const int64_t *tuples, int64_t n_unique, int32_t chunk, int64_t *verdict, int64_t *gid);

/* ---- f4, per-sample batching: the QIIME 2 plugin runs the whole path once per sample
 * (itsxpress/q2_itsxpress.py:273-333: one SeqSample, one vsearch and one hmmsearch process per manifest row), which
 * starves a GPU when samples are small.  Here many samples share ONE read set and one pass of every kernel while each
 * keeps the results of its own run: reads are dereplicated only within their sample (the first occurrence INSIDE the
 * sample is the representative, so orientation and labels are the sample's own), and hmmsearch's domZ -- the number
 * of reported targets the domain E-value threshold is scaled by -- is counted per (sample, profile).
 * itsx_load_reads_files: the samples' sequence files (FASTA/FASTQ, plain/gzip/zstd) in order; sample index = file index.
 * itsx_set_samples: the same for reads handed over with itsx_set_reads (sample_of_read[n_reads], values in
 * [0, n_samples)); NULL or n_samples <= 1 returns to one sample.  A new read set resets the batch to one sample.
 * With S samples itsx_get_domz / itsx_set_domz move S * n_profiles counters ([sample][profile]).
 * itsx_select_sample: the file writers (itsx_write_uc / _rep_fasta / _domtbl) emit that sample only (cluster numbers
 * and the E-value columns as in a run of that sample alone); -1 = every sample.
 * Not batched: itsx_cluster at id < 1 (sequential per sample) and itsx_unique_keys (shard whole samples instead). */
int itsx_load_reads_files(itsx_ctx *ctx, const char *const *paths, int32_t n_paths, int64_t *n_reads_per_file);
int itsx_set_samples(itsx_ctx *ctx, const int32_t *sample_of_read, int32_t n_samples);
int itsx_num_samples(const itsx_ctx *ctx);
int itsx_select_sample(itsx_ctx *ctx, int32_t sample);

/* ---- f4: SeqSample.orient_reads (itsxpress/SeqSample.py:48-91) = vsearch --orient IN --db REF --fastqout OUT
 * (12-mer presence counts on both strands, 4x rule; restated in oracle/orc_cluster.c, parity unpinned).
 * itsx_orient_load_db: FASTA (plain/gzip) of reference sequences -> 12-mer bitmap on the device.
 * itsx_orient: per loaded read strand = +1 forward, -1 reverse (to be reverse-complemented), 0 undetermined; counts may be NULL.
 * itsx_write_oriented_fastq (host, context-free): the FASTQ vsearch would write for those orientations. */
int itsx_orient_load_db(itsx_ctx *ctx, const char *fasta_path, int64_t *n_sequences);
int itsx_orient(itsx_ctx *ctx, int8_t *strand, int32_t *count_fwd, int32_t *count_rev);
int itsx_write_oriented_fastq(const char *seq_path, const char *out_path, const int8_t *strand, int64_t n_records, int64_t *n_written);

/* ---- f2: SeqSample._merge_reads (itsxpress/SeqSample.py:266-365) = vsearch --fastq_mergepairs R1 --reverse R2
 * --fastqout seq.fq --fastq_maxdiffs 40 --fastq_maxee 2 --fastq_qmax 93 [--fastq_allowmergestagger]; restated in
 * oracle/orc_merge.c (parity unpinned: the reference's merged fixture was made by another tool).
 * itsx_merge_buffers: pair i = forward read fseq/fqual[foff[i]..foff[i+1]) and reverse read (as in the file)
 * rseq/rqual[roff[i]..roff[i+1]); the merged read of pair i is written at out_seq/out_qual[foff[i] + roff[i]] with
 * length out_len[i]; reason[i]: 0 merged, 1 no shared 5-mers, 2 several candidate alignments, 3 score < 16,
 * 4 > maxdiffs, 5 overlap < 10, 6 staggered, 7 expected errors > maxee, 8 empty/too long.  score/shift may be NULL.
 * itsx_merge_pairs_files: FASTQ (plain/gzip) in, plain FASTQ out, labels = forward identifier up to the first blank.
 * itsx_merge_tables: the score / quality tables (context-free; tests compare them with the oracle's). */
int itsx_merge_buffers(itsx_ctx *ctx, const char *fseq, const char *fqual, const int64_t *foff, const char *rseq, const char *rqual,
                       const int64_t *roff, int64_t n, int maxdiffs, double maxee, int allow_stagger,
                       char *out_seq, char *out_qual, int32_t *out_len, int32_t *reason, double *score, int32_t *shift);
int itsx_merge_pairs_files(itsx_ctx *ctx, const char *r1_path, const char *r2_path, const char *out_path, int maxdiffs, double maxee,
                           int allow_stagger, int64_t *n_pairs, int64_t *n_merged);
/* the same merge with the merged reads left as the context's read set (labels: R1 identifiers of the merged pairs, input order): no
 * seq.fq is written and nothing is parsed again; the merged bases stay on the device.  What an arrays-mode caller uses in place of
 * itsx_merge_pairs_files + itsx_load_reads_file. */
int itsx_merge_pairs_load(itsx_ctx *ctx, const char *r1_path, const char *r2_path, int maxdiffs, double maxee, int allow_stagger,
                          int64_t *n_pairs, int64_t *n_merged);
/* the same from record-aligned pieces of the two files' text that hold the same number of records (a streaming driver's slices), and,
 * per pair of the last merge-and-load, the index of its merged read in the context's read set (-1: not merged) */
int itsx_merge_pairs_load_text(itsx_ctx *ctx, const char *text1, int64_t nbytes1, const char *text2, int64_t nbytes2, int maxdiffs, double maxee, int allow_stagger,
                               int64_t *n_pairs, int64_t *n_merged);
int itsx_merge_pair_index(const itsx_ctx *ctx, int32_t *index, int64_t n_pairs);
int itsx_merge_tables(double *q2p, double *match, double *mism, uint8_t *qsame, uint8_t *qdiff);

/* ---- a1: SeqSample.deduplicate (itsxpress/SeqSample.py:93-131)
 * = vsearch --fastx_uniques --strand both; vsearch's --minseqlength default is 1 for this command (32 for the clustering
 * commands): SeqSample.deduplicate passes 1, shorter reads are dropped (rep_of -1). */
int itsx_derep(itsx_ctx *ctx, int strand_both, int minseqlength, int64_t *n_unique);
/* ---- a2: SeqSample.cluster (itsxpress/SeqSample.py:133-176) = vsearch --cluster_size --id X --strand both:
 * greedy centroid clustering in label order (8-mer candidate ranking, global alignment, --iddef 2 identity,
 * maxaccepts 1 / maxrejects 32), restated in oracle/orc_cluster.c (parity unpinned: the reference holds no
 * fixture for it).  id == 1.0 is exact dereplication (itsx_derep), which is what the reference runs at 1.0
 * (main.py:534-537).  Fills the same arrays as itsx_derep. */
int itsx_cluster(itsx_ctx *ctx, double id, int strand_both, int64_t *n_unique);
/* after itsx_cluster at id < 1: pct_id[n_reads] = identity of each member with its centroid (uc column 4; -1 for
 * centroids and dropped reads), order[n_order] = kept reads in processing order (the order of uc's S/H rows). */
int itsx_get_cluster(const itsx_ctx *ctx, double *pct_id, int64_t *order, int64_t *n_order);
/* Dedup.parse's matchdict (itsxpress/SeqSample.py:542-562) as arrays over reads:
 * rep_of[i] = read index of the cluster seed (first occurrence), -1 if the read was dropped;
 * strand[i] = +1 / -1 (uc column 5); uniq_of[i] = index into the unique list, -1 if dropped. */
int itsx_get_derep(const itsx_ctx *ctx, int64_t *rep_of, int8_t *strand, int64_t *uniq_of);
/* ---- multi-GPU exact dereplication (SURVEY 8e option 2; itsxpress_amd/dist.py:global_derep).
 * itsx_unique_keys: XXH64 (given seed) of each local unique's packed forward strand and of its reverse complement,
 * to be exchanged between ranks.  itsx_set_active_uniques: active[U] != 0 keeps a unique in the set the HMM stages
 * score (a unique whose global first occurrence lives on another rank is scored THERE); inactive uniques get no
 * domains, their coordinates arrive through the exchange.  A later itsx_derep / itsx_cluster resets the set. */
int itsx_unique_keys(itsx_ctx *ctx, uint64_t seed, uint64_t *fwd, uint64_t *rc);
int itsx_set_active_uniques(itsx_ctx *ctx, const uint8_t *active);
/* read index of each unique (rep.fa order = input order of seeds), abundance of its cluster */
int itsx_get_uniques(const itsx_ctx *ctx, int64_t *seed_read, int64_t *abundance);

/* ---- a4: SeqSample._search (itsxpress/SeqSample.py:178-225)
 * = hmmsearch --domtblout -T <T> --F1 --F2 --F3 (domE = 10).  With the reference's flags (F1 == F2) the Viterbi filter never
 * runs; with F2 < F1 (hmmsearch's own defaults: 0.02, 1e-3, 1e-5) it runs between the bias filter and Forward (k_vit).
 * itsx_search runs every stage up to per-sequence reporting and counts, per profile, the
 * reported representatives (hmmsearch's domZ).  itsx_search_finalize applies the
 * domain threshold.  Between the two a multi-GPU driver all-reduces domZ. */
int itsx_search(itsx_ctx *ctx, double T, double F1, double F2, double F3);
/* What itsx_search keeps of hmmsearch's per-domain table (--domtblout, itsxpress/SeqSample.py:190-209).  The reference's only
 * consumer, ItsPosition.parse/_score (SeqSample.py:400-461), keeps per target and side the first row with the strictly greatest
 * %.1f score and drops the rest:
 *   ITSX_ROWS_FULL     every row stays resident (itsx_get_domains / itsx_write_domtbl work): what _search needs with --keeptemp;
 *   ITSX_ROWS_COMPACT  every pair is evaluated, only the rows that can still win that argmax once domZ is known are kept;
 *   ITSX_ROWS_LAZY     pairs that cannot win it are not evaluated past their Forward score (csrc/k_lazy.hip: a rigorous bound of
 *                      a pair's best domain score from its Forward score); coordinates are exactly those of the other modes.
 * In the last two the row table / domtbl.txt are refused -- unless the caller asks for the KEPT rows: itsx_set_kept_rows(ctx, 1)
 * makes itsx_num_domains / itsx_get_domains / itsx_write_domtbl serve, after itsx_search_finalize, the WINNERS among the rows the context
 * holds: per target and 2-character profile prefix the reported row ItsPosition's argmax ends up with -- a row of the full table, field
 * for field, in --domtblout order.  ItsPosition.parse (SeqSample.py:400-461) reads the same dictionary out of that domtbl.txt as out of
 * the full one (itsxpress_amd/SeqSample.py: ITSXPRESS_DOMTBL=winners); the '#' / 'of' columns count the listed rows of the target.
 * mode -1 (the default) reads the environment at every search: ITSX_ROWS=full|compact|lazy, or ITSX_COMPACT_ROWS=1.
 * After a LAZY search hmmsearch's domZ is known by bounds only: itsx_get_domz / itsx_set_domz / itsx_domz_device move
 * itsx_domz_count() = 2 x n_samples x n_profiles counters (lower bounds, then upper bounds; a multi-rank driver sums both).
 * itsx_search_finalize decides every row both bounds decide alike.  itsx_lazy_pending() = rows left undecided that could change
 * a coordinate (itsx_trim_coords refuses while it is positive): see itsx_lazy_complete. */
enum { ITSX_ROWS_FULL = 0, ITSX_ROWS_COMPACT = 1, ITSX_ROWS_LAZY = 2 };
int itsx_set_rows_mode(itsx_ctx *ctx, int mode);
int itsx_set_kept_rows(itsx_ctx *ctx, int on);
int64_t itsx_lazy_pending(const itsx_ctx *ctx);
/* The profiles of the undecided rows (flags[n_profiles], 1 = some row of this profile is pending), and the cure: itsx_lazy_complete
 * sends EVERY pair of the flagged profiles through the domain pipeline, after which their counters are exact (lower == upper) and
 * every row of theirs is decided.  Alone, itsx_search_finalize does this by itself; a multi-rank driver ORs the flags over the
 * ranks, completes on every rank, exchanges the counters again and finalizes again. */
int itsx_lazy_pending_profiles(const itsx_ctx *ctx, int32_t *flags);
/* The representatives of the undecided rows (flags[n_unique]); itsx_set_partial_coords(ctx, 1): itsx_rep_coords / itsx_trim_coords
 * answer although rows are undecided -- the caller promises not to use the flagged representatives' coordinates (a streaming driver
 * finalizes a chunk with provisional bounds, writes what is decided and comes back for the rest: itsxpress_amd/stream.py). */
int itsx_lazy_pending_uniques(itsx_ctx *ctx, uint8_t *flags);
int itsx_set_partial_coords(itsx_ctx *ctx, int on);
int itsx_lazy_complete(itsx_ctx *ctx, const int32_t *flags);
int64_t itsx_domz_count(const itsx_ctx *ctx);
int itsx_get_domz(const itsx_ctx *ctx, int64_t *domZ /* [itsx_domz_count]: [n_samples][n_profiles] (x 2 after a lazy search) */);
int itsx_set_domz(itsx_ctx *ctx, const int64_t *domZ /* same */);
int itsx_search_finalize(itsx_ctx *ctx, double domE);
int64_t itsx_num_domains(const itsx_ctx *ctx);
int itsx_get_domains(const itsx_ctx *ctx, itsx_domain *rows /* [itsx_num_domains] */);
int64_t itsx_num_pairtraces(const itsx_ctx *ctx);
/* row_size = sizeof(itsx_pairtrace) as the CALLER was compiled: a library built from other sources is refused (ITSX_E_ARG) instead of
 * writing rows of another stride */
int itsx_get_pairtraces(const itsx_ctx *ctx, itsx_pairtrace *rows, int64_t row_size);

/* ---- a5/a6/a7: ItsPosition.parse/_score/get_position (itsxpress/SeqSample.py:400-498)
 * composed with Dedup.matchdict: per READ, start = left.env_to, stop = right.env_from - 1,
 * tlen = length of the representative; -1 where the reference returns None; in_ddict[i] = 0
 * where ItsPosition.get_position would raise KeyError (or the read was dropped). */
int itsx_trim_coords(itsx_ctx *ctx, const char *left_prefix, const char *right_prefix,
                     int32_t *start, int32_t *stop, int32_t *tlen, int32_t *in_ddict);
/* same, per unique representative */
int itsx_rep_coords(itsx_ctx *ctx, const char *left_prefix, const char *right_prefix,
                    int32_t *start, int32_t *stop, int32_t *tlen, int32_t *in_ddict);

/* ---- device-resident exchange for multi-GPU drivers (itsxpress_amd/dist.py; RCCL over xGMI): the quantities the ranks
 * exchange stay in the context's device memory, so a collective library reduces / gathers / scatters them where they are.
 * Every pointer is valid until the next call that recomputes the same quantity on this context.
 * itsx_domz_device: after itsx_search, int64[n_samples * n_profiles] reported-target counters; all-reduce them in place, then
 *   call itsx_search_finalize (it uses the device values; itsx_set_domz returns to host values).
 * itsx_trim_coords_device / itsx_rep_coords_device: int32 rows [n][4] = (start, stop, tlen, in_ddict) per read / representative.
 * itsx_derep_device: rep_of / uniq_of (int32[n_reads]), strand (int8[n_reads]), seed_read (int32[n_unique]).
 * itsx_unique_keys128_device: int64 rows [n_unique][4] = (key0, key1, gidx_base + first occurrence, 1 if the forward strand is
 *   the canonical orientation): the orientation-free 128-bit key of each local unique (two XXH64 seeds). */
int itsx_domz_device(itsx_ctx *ctx, int64_t **d_domz, int64_t *n);
int itsx_trim_coords_device(itsx_ctx *ctx, const char *left_prefix, const char *right_prefix, int32_t **d_rows, int64_t *n_rows);
int itsx_rep_coords_device(itsx_ctx *ctx, const char *left_prefix, const char *right_prefix, int32_t **d_rows, int64_t *n_rows);
int itsx_derep_device(itsx_ctx *ctx, const int32_t **d_rep_of, const int32_t **d_uniq_of, const int8_t **d_strand, const int32_t **d_seed_read);
int itsx_unique_keys128_device(itsx_ctx *ctx, uint64_t seed_a, uint64_t seed_b, int64_t gidx_base, int64_t **d_tuples, int64_t *n_unique);
/* the same tuples copied to host memory (tuples[n_unique][4]): for a driver that has no collective library and moves them itself */
int itsx_unique_keys128(itsx_ctx *ctx, uint64_t seed_a, uint64_t seed_b, int64_t gidx_base, int64_t *tuples);

/* ---- file-compatible outputs (users pass --keeptemp; itsxpress/SeqSample.py:104-105,190) */
int itsx_write_uc(const itsx_ctx *ctx, const char *path);
int itsx_write_rep_fasta(const itsx_ctx *ctx, const char *path);
int itsx_write_domtbl(const itsx_ctx *ctx, const char *path);
/* The same three files written from ARRAYS (host-only, context-free: csrc/writers_host.cpp), for a driver that spreads ONE sample
 * over several GPUs and must hand the reference's parsers one uc.txt / rep.fa / domtbl.txt, byte for byte what one GPU writes.
 * itsx_write_derep_arrays: rep_of[i] = read index of the cluster's seed (-1 dropped), strand[i] = +1 / -1 relative to the seed,
 *   len[i], names (NULL: r%09d), the seeds' sequences concatenated in input order of the seeds; either path may be NULL.
 * itsx_write_domtbl_arrays: domain rows from any number of contexts with rep = index into target_names (the global unique list),
 *   Z = targets searched, domz[n_profiles] = the data set's reported targets per profile, per profile NAME / M / Forward tau, lambda.
 * itsx_profile_params / itsx_get_unique_seqs: what a driver needs from its contexts to fill those arrays.
 * Errors: negative code, text from itsx_writers_last_error(). */
int itsx_write_derep_arrays(const char *uc_path, const char *rep_path, int64_t n, const int64_t *rep_of, const int8_t *strand,
                            const int32_t *len, const char *names, const int64_t *name_offsets, const char *seed_bases,
                            const int64_t *seed_offs, int64_t n_seeds);
int itsx_write_domtbl_arrays(const char *path, const itsx_domain *rows, int64_t n_rows, int64_t Z, const int64_t *domz, int32_t n_profiles,
                             const char *prof_names, const int64_t *prof_name_offsets, const int32_t *prof_M, const float *prof_tau,
                             const float *prof_lambda, const char *target_names, const int64_t *target_name_offsets);
const char *itsx_writers_last_error(void);
int itsx_profile_params(const itsx_ctx *ctx, int i, int32_t *M, float *evparam6);
int itsx_get_unique_seqs(itsx_ctx *ctx, char *bases, int64_t cap, int64_t *offsets);

/* labels of the loaded reads in input order (what Dedup.matchdict is keyed by, itsxpress/SeqSample.py:542-562; the paired
 * writer looks R1/R2 records up by them): concatenated into names[cap], offsets[n_reads + 1]; names == NULL fills the
 * offsets only (offsets[n_reads] = bytes needed). */
int itsx_get_read_names(const itsx_ctx *ctx, char *names, int64_t cap, int64_t *offsets);

/* out_size = sizeof(itsx_stats) as the caller was compiled (checked, like itsx_get_pairtraces' row_size) */
int itsx_get_stats(const itsx_ctx *ctx, itsx_stats *out, int64_t out_size);
/* The library's environment switches that are set ("NAME=value" lines; the registry with every switch's class -- tuning, mode,
 * diagnostic, test hook -- is csrc/switches.cpp, the table INTEGRATION.md section 7): of ctx's last search (snapshot taken when it
 * started) or, with ctx == NULL, of the environment now.  A test hook (result-changing) is honoured only under ITSX_TEST_HOOKS=1 and
 * says so here otherwise.  The reference has no such surface: it passes fixed flags (itsxpress/SeqSample.py:191-209).
 * Returns the length needed, terminator included. */
/* A context that will not search again soon (a streamed file's chunk, itsxpress_amd/stream.py) hands its largest scratch buffer -- the DP
 * slab, tens of GB -- to the next context of the process on that device; results and everything itsx_search_finalize /
 * itsx_lazy_complete need stay.  The reference has nothing of the kind (one process per tool run, itsxpress/SeqSample.py:117,210). */
int itsx_release_scratch(itsx_ctx *ctx);
int64_t itsx_switches(const itsx_ctx *ctx, char *buf, int64_t cap);
/* the registry: "NAME<tab>class<tab>meaning" lines, class = tuning | mode | diagnostic | hook */
int64_t itsx_switch_registry(char *buf, int64_t cap);

/* ---- f1 ("next"): native FASTQ parse -> slice -> write.  Host-only and context-free.
 * itsx_write_trimmed_fastq replaces Dedup.create_trimmed_seqs (itsxpress/SeqSample.py:886-949): record i of
 * seq_path (plain or .gz) is written as record[start[i]:stop[i]] iff both are >= 0 and start < stop.
 * itsx_write_trimmed_paired replaces Dedup.create_paired_trimmed_seqs (SeqSample.py:713-790): R1/R2 pair k is
 * looked up by the id of its R1 title among `names` (the merged reads the coordinates belong to) and sliced
 * with the reference's r2start = tlen - stop / r2end = tlen - start arithmetic.  trim_ccs stitches the CCS
 * primers with quality 93.  Errors: negative code, text from itsx_trim_last_error().
 * Inputs are plain, gzip or zstd (told apart by their magic bytes; the reference goes by .gz / .zst, main.py:296-330).
 * `compression` of the output: 0 plain, 1 gzip, 2 zstd (the reference's gzipped / zstd_file flags, SeqSample.py:909-925).
 * Compressed output is written as concatenated gzip members / zstd frames produced by a pool of threads
 * (ITSX_IO_THREADS, default min(hardware threads, 32)); any gzip / zstd reader sees one stream. */
int itsx_write_trimmed_fastq(const char *seq_path, const char *out_path, int compression, int trim_ccs,
                             const int32_t *start, const int32_t *stop, int64_t n_records,
                             int64_t *n_written, int64_t *total_len);
/* The same writer as an object that takes the input's TEXT and the COORDINATES piece by piece and slices / deflates on its own
 * threads while more arrive (a streaming run writes the first chunks' reads while the GPU scores the later ones).  The text is cut
 * into units at the first record start at or after every multiple of 8 MB; a unit's output is one gzip member / zstd frame, units are
 * written in order, so the file's bytes do not depend on how text and coordinates arrived: itsx_write_trimmed_fastq IS this object
 * fed once.  itsx_twriter_text: text[0, avail) is final (one address, growing; last = 1 with the final piece; the text must stay
 * valid until close).  itsx_twriter_coords: rows of records first_record .. first_record + n - 1 (record order, no gaps);
 * decided[i] == 0 (NULL: all decided) marks a record whose coordinates may still change -- it holds back its own unit only, until
 * itsx_twriter_update names it.  itsx_twriter_close waits for the pool, fails if a record was left undecided, frees the object.
 * Errors: negative code, text from itsx_trim_last_error(). */
int itsx_twriter_open(const char *out_path, int compression, int trim_ccs, itsx_twriter **w);
/* mode 1 (a paired run's mates, itsxpress/SeqSample.py:587-670): (start, stop) are Python slice bounds as they come -- start may be
 * negative, stop == INT32_MAX = open end, stop == INT32_MIN = the record is not written */
int itsx_twriter_set_mode(itsx_twriter *w, int32_t mode);
int itsx_twriter_text(itsx_twriter *w, const char *text, int64_t avail, int32_t last);
int itsx_twriter_coords(itsx_twriter *w, int64_t first_record, int64_t n, const int32_t *start, const int32_t *stop, const uint8_t *decided);
int itsx_twriter_update(itsx_twriter *w, const int64_t *records, int64_t m, const int32_t *start, const int32_t *stop);
int itsx_twriter_close(itsx_twriter *w, int64_t *n_written, int64_t *total_len);

int itsx_write_trimmed_paired(const char *r1_path, const char *r2_path, const char *out1_path, const char *out2_path,
                              int compression, int trim_ccs, const char *names, const int64_t *name_offsets, int64_t n_names,
                              const int32_t *start, const int32_t *stop, const int32_t *tlen, int64_t *n_written);
const char *itsx_trim_last_error(void);
/* The readers behind every *_file entry point, exposed for the host side (replaces gzip.open / pyzstd.open of
 * main.py:296-330 and SeqSample.py:929-949): the decompressed content of a plain / gzip / zstd file in a buffer the
 * caller releases with itsx_io_free.  itsx_io_codecs: bit 0 = libdeflate in use for gzip, bit 1 = libzstd available.
 * A single-member gzip file of more than a few MB is inflated by a pool of threads (csrc/pinflate.cpp: block starts are
 * found inside the stream, every chunk is decoded with its unknown 32-KB history as markers, the markers are resolved
 * front to back) and accepted only when length and CRC-32 match the trailer; everything else takes the serial inflater.
 * itsx_io_parallel_inflates: files delivered that way since the library was loaded (ITSX_PARALLEL_INFLATE=0 turns it off). */
int  itsx_io_read(const char *path, char **text, int64_t *len);
/* identifiers of a FASTQ file's records (title up to the first blank) through the record parser the trimming writers use:
 * names[offsets[i] .. offsets[i + 1]) is record i's; both buffers are released with itsx_io_free */
int  itsx_fastq_ids(const char *path, char **names, int64_t **offsets, int64_t *n_records);
void itsx_io_free(char *text);
int  itsx_io_codecs(void);
int64_t itsx_io_parallel_inflates(void);
/* drops the process-wide cache of decompressed texts (ITSX_TEXT_CACHE_GB; the loaders leave a file's text there for the writers) */
void itsx_io_cache_clear(void);

/* ---- test hooks (parity tests only) */
/* XXH64 of each read's packed forward / reverse-complement key, as computed on the device */
int itsx_debug_read_hashes(itsx_ctx *ctx, uint64_t *fwd, uint64_t *rc);
/* packed representation of read i: words [ceil(len/16)], exception list (pos<<4|code) */
int itsx_debug_packed_read(const itsx_ctx *ctx, int64_t i, uint32_t *words, int32_t *nwords,
                           uint32_t *exc, int32_t *nexc);
/* PMC calibration (scripts/fetch_calib.py): stream `gbytes` GB `iters` times in one of the DP slab's access patterns --
 * 0: read all 6 fields of a [row][6][64] float plane at 4 B per lane, 1: read 5 of the 6 fields (k_decode), 2: write all 6,
 * 3: read 16 B per lane -- and report the bytes one launch touches, so rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled */
int itsx_debug_calibrate(itsx_ctx *ctx, int pattern, double gbytes, int iters, int64_t *bytes_per_launch, double *ms_per_launch);
/* VALU issue-rate probe behind bench.py's valu_issue_frac (profiles/round5_valu_issue.md): op 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_max_i16,
 * 3 v_pk_add_u16, 4 v_pk_mul_f32, 5 v_pk_add_f32, 6 s_nop 0, 7 v_mul_f32, 8 v_pk_mov_b32, 9 v_max_i16; 10 the scalar-cache probe: two
 * s_load_dwordx8 of a 1-KB table per step, waited for one step later (what k_fwd_bound asks per pair of nodes), 11 the same under 12
 * v_pk_fma_f32 per step (scripts/scalar_probe.py); 64 x iters independent instructions (steps) per
 * wave, waves_per_simd (1-4, 6, 8) waves on every SIMD; cycles_per_instr as one wave sees them (shader clock ticks, median over the waves) */
int itsx_debug_issue(itsx_ctx *ctx, int op, int waves_per_simd, int iters, double *cycles_per_instr, double *ms);
/* deterministic log/exp evaluated ON THE DEVICE for n inputs */
int itsx_debug_detmath(itsx_ctx *ctx, const double *x, int64_t n, double *out_log, double *out_exp);
/* the bias filter's table-driven float logarithm (detmath.h: det_logf_fast) ON THE DEVICE: out[i] must equal (float)det_log((double)x[i]) */
int itsx_debug_logf(itsx_ctx *ctx, const float *x, int64_t n, float *out);
/* the DUST soft mask (vsearch --qmask dust) the device computes for the current reads: masked[sum of lengths], 1 = masked */
int itsx_debug_dust(itsx_ctx *ctx, uint8_t *masked);

#ifdef __cplusplus
}
#endif
#endif
