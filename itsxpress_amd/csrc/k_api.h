// k_api.h -- device-side argument blocks and host launchers shared by the engine's kernels.
#pragma once
#include "engine.h"

#include "detmath.h"
namespace itsx {

struct ReadsDev {
  const uint32_t *words;    // 2-bit packed bases, 16 per word
  const int64_t  *woff;     // [n+1] word offset of each read
  const int32_t  *len;      // [n]
  const int64_t  *excoff;   // [n+1]
  const uint32_t *exc;      // (pos<<4 | code) for non-ACGT symbols
  int64_t         n;
};

// ---- k_derep.hip
// sample (may be null): per-read sample index folded into the hash seed, so equal sequences of different samples get different keys
void launch_hash_reads(const ReadsDev &rd, uint64_t seed, int strand_both, uint64_t *hf, uint64_t *hr, hipStream_t st, const int32_t *sample = nullptr);
void launch_table_insert(int64_t n, const int32_t *len, int minlen, const uint64_t *hf, const uint64_t *hr,
                         unsigned long long *keys, int32_t *vals, uint64_t mask, uint32_t *slot_of, hipStream_t st);
void launch_table_resolve(const ReadsDev &rd, const uint64_t *hf, const int32_t *vals, const uint32_t *slot_of,
                          int32_t *rep_of, int8_t *strand, int32_t *is_seed, unsigned int *n_collisions, hipStream_t st,
                          const int32_t *sample = nullptr);
void launch_uniques(int64_t n, const int32_t *rep_of, const int32_t *seed_rank, int32_t *uniq_of, int32_t *seed_read,
                    int32_t *abundance, hipStream_t st);
void launch_len_hist(int32_t U, const int32_t *seed_read, const int32_t *len, int32_t *hist, int32_t lcap, hipStream_t st, int32_t nb = 0 /*longest read + 1, if known*/);
void launch_len_scatter(int32_t U, const int32_t *seed_read, const int32_t *len, int32_t *cursor, int32_t lcap,
                        int32_t *sorted_uniq, hipStream_t st, int32_t nb = 0);

void launch_region_keys(const ReadsDev &rd, const RegionRec *regions, int64_t nr, const PairRec *pairs, const int32_t *sorted_uniq,
                        const int32_t *seed_read, unsigned long long *keys, int32_t *vals, uint64_t mask, uint32_t *slot_of, hipStream_t st);
void launch_region_resolve(const ReadsDev &rd, const RegionRec *regions, int64_t nr, const PairRec *pairs, const int32_t *sorted_uniq,
                           const int32_t *seed_read, const int32_t *vals, const uint32_t *slot_of, int32_t *rep_region, int32_t *is_uniq, hipStream_t st);
void launch_region_upos(int64_t nr, const RegionRec *regions, const PairRec *pairs, const int32_t *is_uniq, const int32_t *urank,
                        const int64_t *rseg, const int64_t *useg, int32_t *upos, RegionRec *ulist, hipStream_t st);
void launch_region_upos_follow(int64_t nr, const int32_t *rep_region, const int32_t *is_uniq, int32_t *upos, hipStream_t st);

// ---- k_cluster.hip (row a2: greedy centroid clustering, speculative windows)
struct ClusterArgs {
  ReadsDev rd;
  const int32_t *order;         // [nk] read index by processing position (label order)
  int32_t f, nq;                // window = positions [f, f+nq)
  int32_t strand_both;
  int32_t *cent_len, *cent_pos, *cent_read;   // per centroid column
  int32_t C;                    // centroids at the window start
  uint16_t *klist; int32_t kcap; int32_t *nk;  // distinct words of each (query, strand): [2 nq][kcap], [2 nq]
  // the centroids' distinct forward words, column after column (append-only; a rolled-back column has cw_n = 0)
  uint16_t *cw_pool; int64_t *cw_off; int32_t *cw_n; int64_t *cw_base;   // cw_base[0] = cw_off[C] of this window; cw_base[1] = end after validation
  int32_t *wsum, *wscan;        // [nq+1] words of each would-be centroid, and their exclusive scan
  // the window's query index: word -> the (query strand) byte offsets 4 * qs that hold it, each list padded to 8 entries with the
  // dummy offset 4 * QS_MAX
  int32_t *qi_cnt, *qi_cur, *qi_off; uint16_t *qi_ent;   // [65537], [65536], [65537], [qi_off[65536]]
  // conserved words (held by >= CL_HEAVY strands of the window) have no list: a bitmap over the window's strands instead, added
  // by bit-sliced carry-save adders (one dword = 32 strands per lane) -- an LDS atomic per (strand, word) is what the lists cost
  int32_t *qi_hid; int32_t *qi_nheavy; uint32_t *qi_bm; int32_t hcap; int32_t heavy_min;   // [65536] bitmap number or -1; [1]; [hcap][CL_QS_MAX / 32]
  uint32_t *tq, *minm;          // [CL_QS_MAX + 8] thresholds on the HIGH HALF of a rank key, (count << 16 | 65535 - length): a centroid is a
                                //   candidate when its high half is greater.  tq: high half of tkey, or (min(12, words) << 16) - 1 while the strand has
                                //   fewer than 32 candidates; minm: (min(12, words) << 16) - 1 (mode 2); 0xFFFFFFFF = never.  Exact, not a pre-test:
                                //   centroids stream in position order, so one that ties with the 32nd key on count and length always loses on position
  unsigned long long *tkey;     // [2 nq] rank key of the strand's 32nd best candidate so far (0: fewer than 32 yet)
  unsigned long long *cand; int32_t *ncand; int32_t ccap; int32_t *ovf;   // appended candidate keys [2 nq][ccap]; ovf[0] = a list overflowed
  int32_t *ntop;                // [2 nq] candidates in sel/selkey (<= 32, rank order): the whole walk's candidate list
  uint16_t *cntx; int32_t xpitch;   // shared-word counts against the window's speculative centroids [2 nq][xpitch]
  int32_t *state, *rejects, *acc_col;          // walk state per (query, strand): 0 active 1 accepted 2 32 rejects 3 exhausted
  unsigned long long *prev, *bound;            // rank key of the last candidate tried / of the point where the walk stopped
  double *acc_id;
  int32_t *sel, *selm, *sel_short; unsigned long long *selkey; double *selpid;   // the candidate list [2 nq][32]; selm = taken this round
  int32_t *wn, *wcol; unsigned long long *wkey; double *wpid;                    // the recorded walk [2 nq][32]
  int32_t *res_col; int8_t *res_strand; double *res_id;   // outcome by processing position
  int32_t *is_new, *new_rank;   // [nq+1]
  int32_t *newq, *rm;           // [nq] window index of each speculative centroid; columns to clear after validation
  int32_t *xlist, *xn, *hard; unsigned long long *xkey; double *xpid;            // speculative centroids entering a walk [2 nq][32]
  int32_t *work, *xwork, *work_n;   // (query strand * 32 + slot) items for the two alignment kernels; work_n[6]: items of work / xwork, of awork's two halves, of spairs' two halves
  int32_t *spairs;                  // [2][need_pitch][2]: the score pass's list: two items of one strand (or one and -1) per entry
  int32_t *awork;                   // [2][need_pitch] the items the certificate leaves to the dynamic program (walk / validation)
  const uint32_t *dmask;        // DUST soft mask of every read, one bit per base at woff[r] (k_dust); nullptr: no masking (ITSX_QMASK=none)
  const uint64_t *rhash;        // [n reads] XXH64 of the packed forward strand
  unsigned long long *ctab_key; int32_t *ctab_val; int32_t *canon;   // window-local table of identical reads; canon[nq]
  int32_t *replay;              // [nq] queries whose walk must be replayed by k_cl_resolve
  int32_t *skipm;               // [nq] minus-strand walk cut short because the plus strand holds a 100 % hit
  int32_t *wout;                // [3] cut, columns consumed, true new centroids
  int32_t *dbg;                 // [4] hard cuts, member->centroid cuts, centroid->member resolutions, validation alignments
  unsigned long long *scratch; int32_t scratch_pitch;     // alignment boundary rows [2 nq * 32][pitch][2]
  double thr;                   // 100 * id
  unsigned long long *n_align;
  int32_t *need; int32_t need_pitch;             // rejection certificate: need[which * pitch + item] = 0 when the full alignment is not needed (nullptr: off)
  unsigned long long *n_skipped;
  unsigned long long *pre_stats;                 // [4] certificate outcomes: not applicable, bound too weak, a path exists, proven reject
  int32_t pre_k;                                 // largest edit budget K of this run (sizes the certificate's LDS rows)
  int32_t use_score;                             // the score pass (k_cl_score) settles the candidates whose best diagonal says nothing
};
constexpr int CL_QS_MAX = 8192;  // query strands of one window (2 x the largest window)
constexpr int CL_HEAVY = 128;    // a word held by at least this many strands of the window goes through its strand bitmap
void launch_cl_kmers(const ClusterArgs &a, hipStream_t st);
void launch_cl_qindex(const ClusterArgs &a, int32_t *scan_tmp, hipStream_t st);
void launch_cl_stream(const ClusterArgs &a, int c0, int c1, int mode, hipStream_t st);      // mode 1: the old centroids [c0, c1); mode 2: the window's speculative ones
void launch_cl_topk(const ClusterArgs &a, int final, hipStream_t st);
void launch_cl_init(const ClusterArgs &a, hipStream_t st);
void launch_cl_walk(const ClusterArgs &a, int rows_per_lane, hipStream_t st);
void launch_cl_outcome(const ClusterArgs &a, hipStream_t st);
void launch_cl_wsum(const ClusterArgs &a, hipStream_t st);
void launch_cl_columns(const ClusterArgs &a, int clear, hipStream_t st);
void launch_cl_validate(const ClusterArgs &a, int rows_per_lane, hipStream_t st);
void launch_cl_finalize(int32_t nk, const int32_t *order, const int32_t *res_col, const int8_t *res_strand, const double *res_id,
                        const int32_t *cent_read, int32_t *rep_of, int8_t *strand, double *pct, int32_t *is_seed, hipStream_t st);

// ---- f4: read orientation (k_cluster.hip)
void launch_orient(const ReadsDev &rd, const uint32_t *dbbits /*4^12 bits*/, const uint32_t *dmask /*k_dust, or nullptr*/, int8_t *strand, int32_t *cfwd, int32_t *crev, hipStream_t st);
void launch_dust(const ReadsDev &rd, uint32_t *dmask, hipStream_t st);

// ---- k_merge.hip (paired-end merge)
struct MergeArgs {
  const uint8_t *fseq, *fqual, *rseq, *rqual;   // ASCII bases / qualities of the forward and the reverse reads, concatenated
  const int64_t *foff, *roff;                   // [n+1]
  int64_t n;
  int32_t max_total;                            // largest fl + rl (sizes the LDS)
  int32_t maxdiffs, allow_stagger; double maxee;
  const double *q2p, *match, *mism;             // [128], [128*128], [128*128]
  const uint8_t *qsame, *qdiff;                 // [128*128]
  uint8_t *out_seq, *out_qual;                  // merged read of pair i at foff[i] + roff[i]
  int32_t *out_len, *reason, *shift; double *score;
};
void launch_merge(const MergeArgs &a, hipStream_t st);

// ---- k_util.hip
// exclusive prefix sum of n int32 values (n < 2^31); tmp must hold scan_tmp_elems(n) int32
int64_t scan_tmp_elems(int64_t n);
void    launch_exclusive_scan(const int32_t *in, int32_t *out, int64_t n, int32_t *tmp, hipStream_t st);
void    launch_pack(const uint8_t *raw, int64_t raw_base, const int64_t *off, const int64_t *woff, int64_t r0, int64_t r1, const int8_t *lut,
                    uint32_t *words, int32_t *excnt, long long *first_bad, hipStream_t st);
// exstart: exclusive scan of excnt over reads [r0, r1] (exstart[r1] = the chunk's total); ebase[0] advances by it, excoff[r1] is set
void    launch_pack_exc(const uint8_t *raw, int64_t raw_base, const int64_t *off, int64_t r0, int64_t r1, const int8_t *lut, const int32_t *excnt,
                        const int32_t *exstart, long long *ebase, int64_t ecap, int64_t *excoff, uint32_t *exc, hipStream_t st);
// PMC calibration streams: pattern 0 read 6 of 6 fields (4 B/lane), 1 read 5 of 6, 2 write 6 of 6, 3 read 16 B/lane
void    launch_calib(int pattern, float *slab, int64_t nwaves, int64_t R, float *out, hipStream_t st);
// VALU issue-rate probe: op 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_max_i16, 3 v_pk_add_u16, 4 v_pk_mul_f32, 5 v_pk_add_f32, 6 s_nop 0, 7 v_mul_f32,
// 8 v_pk_mov_b32, 9 v_max_i16; 64 x iters instructions per wave, ticks[block * waves + wave] = s_memtime ticks
void    launch_issue(int op, int waves_per_simd, int iters, int blocks, unsigned long long *ticks, float *sink, hipStream_t st);
void    launch_detmath(const double *x, int64_t n, double *ol, double *oe, hipStream_t st);
void    launch_logf_fast(const float *x, int64_t n, const LogTab *tab, float *out, hipStream_t st);

// ---- k_share.hip: prefix sharing (round 5).  k_msv and k_fwd_bound scan every row of every (representative, profile) pair from row 1,
// and their row state after row i depends on the profile, the target's LENGTH and its first i residues only -- while the representatives
// of amplicon data are error variants of far fewer templates.  The chunk's uniques of equal length form a prefix tree over blocks of B
// rows; a representative's CHAIN is the part of its path that no earlier representative walks: it starts from the row state its parent
// chain saved at the branch point and saves its own state wherever a later chain branches off.
struct TrieArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;   // [U] the length-sorted uniques the search walks
  int32_t U, Uc, B;                          // Uc: uniques per search chunk (nothing is shared across chunks); B: rows per block (16 x 2^n)
  int32_t rev;                               // 0: the PREFIX tree (depth d = the first d blocks).  1: the SUFFIX tree (round 6: depth r = the LAST r blocks, rows
                                             // (A - r) B + 1 .. L with A = ceil(L / B) -- the boundaries are the prefix tree's): mask is left alone (k_join_*)
  unsigned long long *tab; uint64_t tmask;   // open addressing over (length, chunk, depth, prefix) keys: [38-bit tag | 26-bit s], ~0 = empty
  uint8_t *depth;                            // [U] by sorted position s: first block the chain computes itself (0: from row 1)
  int32_t *parent;                           // [U] s of the chain whose saved state it starts from (-1: none)
  unsigned long long *mask;                  // [U] bit d: a later chain starts from this chain's state after row d * B
  int32_t *nn;                               // [U + 1] saved states of the chain (popcount of mask)
  unsigned long long *counters;              // [0] deepest start, [1] rows the chains skip, [2] rows of all uniques, [3] chains with a parent, [4] keys
};
void launch_trie_keycount(const TrieArgs &a, hipStream_t st);  // counters[4] (suffix tree: [6]) += keys (one per unique and depth)
void launch_trie_insert(const TrieArgs &a, hipStream_t st);
void launch_trie_resolve(const TrieArgs &a, hipStream_t st);
void launch_trie_link(const TrieArgs &a, hipStream_t st);      // depth / parent final, masks set
void launch_trie_count(const TrieArgs &a, hipStream_t st);     // nn, counters
void launch_share_cuts(const int32_t *ulen, const int32_t *node0_s, const int32_t *rnode0_s /*nullptr: none*/, int32_t U, int32_t Uc, int32_t cap,
                       int32_t *cuts /*[cap][3]: s, saved Forward / MSV states before s, saved Backward states before s*/, unsigned long long *n, hipStream_t st);
struct PairRec;
void launch_diff_u16(const uint16_t *x, const uint16_t *y, int64_t n, unsigned long long *c, hipStream_t st);
void launch_diff_scores(const float *x, const float *y, const PairRec *pairs, int64_t n, const int32_t *jlev /*by useq, or nullptr*/, unsigned long long *c /*[2]*/, hipStream_t st);
void launch_popc64(const unsigned long long *m, int32_t *out, int64_t n, int64_t nvalid, hipStream_t st);
constexpr int SHARE_SEGS = 66;             // per batch: first k of depth 0 .. 64, and the batch's end
struct ShareDev {                          // the tree by processing position k
  uint8_t *depth; int32_t *parent /* k relative to the chunk */; unsigned long long *mask; int32_t *nn, *node0; int32_t *order /* global unique */, *ulen;
  int32_t *src;                            // number of the saved state the chain starts from (-1: from row 1): one coalesced load instead of two gathers
  // two-sided sharing (round 6), Forward chains: last row the chain computes, the level it joins a saved Backward state at (-1: none; it
  // then runs to its L) and that state's number
  int32_t *endrow, *jlev, *jsrc;
};
void launch_share_src(const ShareDev &o, int32_t U, int32_t Uc, hipStream_t st);     // after node0 is scanned
// ---- two-sided sharing (round 6): the suffix tree's links are in rdepth / rparent (TrieArgs::rev = 1, its hash table still alive)
struct JoinArgs {
  TrieArgs t;                                // the SUFFIX tree's arguments (rev = 1; depth / parent = rdepth / rparent by sorted position s)
  const uint8_t *fdepth; const unsigned long long *fmask;     // the prefix tree by s
  unsigned long long *rmask;                 // [U] bit r: somebody takes this chain's Backward state r blocks from the end
  int32_t *jlev, *jown;                      // [U] join level (-1: none) and the owner (s) of the Backward state there
  int32_t *endrow;                           // [U] last row of the Forward chain
  int32_t *rsteps;                           // [U] rows of the Backward chain (0: it does not run)
  unsigned long long *counters;              // [10] joins, [11] Forward rows, [12] Backward rows, [13] deepest suffix start among the runners, [14] Backward chains
};
void launch_join_resolve(const JoinArgs &a, hipStream_t st);     // jlev / jown, rmask bits of the owners
void launch_join_up(const JoinArgs &a, int r, hipStream_t st);   // chains that start r blocks from the end and run make their parents save there (r descending)
void launch_join_ends(const JoinArgs &a, hipStream_t st);        // endrow, rsteps, counters
// processing orders by a stable radix sort (k_order.hip): key[s] -> order; segk[b][d] = first position with (batch, depth) >= (b, d)
void launch_order_keys(const uint8_t *depth, const int32_t *minor, const int32_t *runs /*nullptr: all*/, int32_t U, const int32_t *bstart, int nb, unsigned long long *keys, int32_t *vals, hipStream_t st);
size_t order_sort_bytes(int64_t n);
int  order_sort(void *tmp, size_t bytes, const unsigned long long *kin, unsigned long long *kout, const int32_t *vin, int32_t *vout, int64_t n, hipStream_t st);
void launch_order_segk(const unsigned long long *sorted_keys, int32_t n, int nb, int32_t *segk /*[nb][SHARE_SEGS]*/, int32_t *inv /*[U] or nullptr*/, const int32_t *vals, int32_t U, unsigned long long *nvalid, hipStream_t st);
void launch_share_permute(const TrieArgs &a, const int32_t *ulen_s, const int32_t *uorder, const int32_t *inv, const int32_t *endrow_s, const int32_t *jlev_s,
                          const ShareDev &o, hipStream_t st);
struct BShareDev {                         // the Backward chains by backward position kb (round 6): only the uniques that save a state for somebody
  uint8_t *depth; int32_t *parent /* kb, global */; unsigned long long *mask; int32_t *nn, *node0; int32_t *order /* global unique */, *ulen;
  int32_t *src;                            // number of the saved Backward state the chain starts from (-1: from row L)
  int32_t *steps;                          // rows the chain walks (those past L of a chain that starts at the end included)
};
void launch_bshare_permute(const TrieArgs &a, int32_t Ub, const int32_t *ulen_s, const int32_t *border_s, const int32_t *invb, const unsigned long long *rmask_s,
                           const int32_t *rsteps_s, const BShareDev &o, hipStream_t st);
void launch_bshare_src(const BShareDev &o, int32_t Ub, hipStream_t st);
void launch_join_src(int32_t U, int32_t B, const int32_t *uorder, const int32_t *jown_s, const int32_t *invb, const BShareDev &ob, const ShareDev &o, int32_t *jownb, hipStream_t st);
void launch_join_need(const int32_t *jlev, const int32_t *jownb, int32_t U, int32_t W, int32_t cb0, const uint32_t *pass, uint32_t *need_b, hipStream_t st);
void launch_need_up_b(int r, const uint8_t *rdepth, const int32_t *rparent, int32_t cb0, int32_t Ub, int32_t W, uint32_t *need_b, hipStream_t st);
void launch_need_res(const uint32_t *need_b, int32_t Ub, int32_t P, int32_t W, uint16_t *res, hipStream_t st);
// one launch of k_msv / k_fwd_bound over chains that start at the same depth (pointers relative to the chunk)
struct ShareLaunch {
  const int32_t *src; const unsigned long long *mask; const int32_t *node0;
  void *slots;                 // saved row states: [(node - node_base) * Pb + profile - p0][MSV_STATE_Q uint4 | FWD_STATE_Q float4]
  int64_t node_base; int32_t p0, Pb;
  int32_t depth, logB;
  float rescale;               // k_fwd_bound: a row's cells are scaled back when its E passes this
  // two-sided sharing (round 6).  k_fwd_bound: endrow[k] = last row chain k computes (nullptr: every chain runs to its L), jlev[k] = the level
  // (row jlev * B) at which it takes the rest of the sum over paths from Backward state jsrc[k] of gslots (-1: no join).
  // k_bwd_bound: src / mask / node0 / slots / node_base describe the BACKWARD chains (by backward position), depth = blocks from the end the
  // launch's chains start at, endrow[kb] = rows (virtual ones past L included) chain kb walks
  const int32_t *endrow, *jlev, *jsrc;
  const void *gslots; int64_t gnode_base;
  // everything a lane needs of its chain in ONE 64-byte record by processing position (k_fwd_bound: by k; k_bwd_bound: by kb) instead of a
  // chain of dependent loads (pair -> sorted position -> read -> offsets): a wave's start-up was 30 rows' worth of latency
  const struct ChainRec *chain;
  const float *entab;          // [P][NCODE][2 * BOUND_PAIRS] the folded emission odds by node (engine.hip: install_profiles)
  int32_t dbg;                 // diagnostic (ITSX_TEST_HOOKS=1 ITSX_PASSA_DBG=bits; results are garbage): 1 no rows, 2 no restore, 4 no join; k_bwd_bound: 8 no saves, 16 no restore, 32 no rows
};
struct alignas(16) ChainRec {
  int64_t woff, excoff;        // the read's packed words / exceptions
  int32_t nexc, L, src, node0; // src: saved state the chain starts from; node0: first of its own saved states
  unsigned long long mask;     // levels it saves at
  int32_t endrow;              // Forward: last row; Backward: rows it walks
  int32_t jlev, jsrc, pad;     // Forward: join level (-1: none) and the Backward state joined
};
static_assert(sizeof(ChainRec) == 64, "k_fwd_bound / k_bwd_bound read a chain's record as four 16-byte loads");
void launch_chain_recs(int32_t n, const ReadsDev &rd, const int32_t *order /*global unique by position*/, const int32_t *seed_read, const int32_t *src, const int32_t *node0,
                       const unsigned long long *mask, const int32_t *endrow, const int32_t *jlev, const int32_t *jsrc, ChainRec *out, hipStream_t st);
constexpr int MSV_STATE_Q = 8;             // 23 packed registers + xJ, xB, xEmax, padded to one 128-byte line
constexpr int FWD_STATE_Q = 36;            // M, I, D of 46 nodes + xN xJ xC xB + the scale's logarithm (double)
// lazy searches only: a chain whose own pair failed the MSV filter still has to run for a profile when a chain below it needs its state
void launch_need_bits(const uint16_t *res, int32_t U, int32_t P, int32_t W, uint32_t *pass, uint32_t *need, hipStream_t st);
void launch_need_up(int d, const uint8_t *depth, const int32_t *parent, int32_t U, int32_t W, uint32_t *need, hipStream_t st);
void launch_need_mark(uint16_t *res, int32_t U, int32_t P, int32_t W, const uint32_t *pass, const uint32_t *need, unsigned long long *n_helpers, hipStream_t st);
// wave list of one chunk's pairs in share order: bnd[t][p] = first pair of profile p with useq >= segk[t]
void launch_share_bounds(const PairRec *pairs, const int64_t *seg_start, const int32_t *total, const int32_t *segk, int32_t nseg, int32_t P, int64_t *bnd, hipStream_t st);
void launch_share_wcount(const int64_t *bnd, int32_t nseg, int32_t P, int32_t *wc, hipStream_t st);            // wc[t * P + p] = waves; wc[nseg * P] = 0
struct WaveDesc;
void launch_share_waves(int32_t nw, int32_t nseg, int32_t P, const int32_t *woff, const int64_t *bnd, const int32_t *seg_depth, int32_t B, const PairRec *pairs,
                        const int32_t *endrow /*by useq; nullptr: the pair's L*/, int backward, WaveDesc *w,
                        unsigned long long *lane_rows /*[64][2], zeroed: rows computed, rows of the pairs (summed by the host)*/, hipStream_t st);

// ---- k_msv.hip
struct MsvArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq;   // [U] unique index, ascending length
  const int32_t *seed_read;     // [U] read index of each unique
  int32_t U;
  int32_t G;                    // profile groups of 64 (the stride of thr)
  int32_t P;                    // profiles
  int32_t PB;                   // profiles one block takes its 256 sequences through (grid.y = ceil(P / PB))
  const uint32_t *etab;         // [P][16 codes][MSV_TW] packed int16x2 of (bias - cost): dword r = cells r, r + 23
  const int32_t *pbias, *ptec, *ptbm;   // [G*64]
  const uint16_t *thr;          // [Lcap][G*64] smallest passing xJ, 256 = only overflow passes
  const int32_t *tjb;           // [Lcap]
  int32_t Lcap;
  uint16_t *res;                // [G*64][U]  bit8 = passed, low byte = xJ (255 = overflow)
  const int32_t *plist;         // optional: only these profiles (the others' rows of res are left as they are); nullptr = all P
  int32_t nlist;
  int32_t k0, k1;               // sorted positions [k0, k1) of this launch (k1 = 0: all U)
  int32_t pfirst, plast;        // profiles (list positions with plist) [pfirst, plast) (plast = 0: all)
  int32_t wtl;                  // packed words a lane keeps in LDS (set by launch_msv: the whole read, or 16 at a time)
  int32_t share;                // 1: prefix sharing, every sequence of [k0, k1) is a chain that starts at sl.depth; 2: the Backward chains of the
                                // two-sided schedule (k_msv_bwd: [k0, k1) are backward positions, sl.depth blocks from the end)
  ShareLaunch sl;
};
void launch_msv(const MsvArgs &a, hipStream_t st, int lds_pad = 0);

// survivor list: pairs grouped by profile, 64-aligned segments, ascending length inside a segment
constexpr int CHUNK = 2048;
void launch_pair_count(const uint16_t *res, int32_t P, int32_t U, int32_t nchunks, int32_t *cnt /*[P][nchunks]*/, int32_t *real /*[P], zeroed: pairs past the filter*/, hipStream_t st);
void launch_chunk_scan(int32_t *cnt, int32_t P, int32_t nchunks, int32_t *total /*[P]*/, hipStream_t st);
void launch_pair_fill(const uint16_t *res, int32_t P, int32_t U, int32_t nchunks, const int32_t *cnt,
                      const int64_t *seg_start /*[P]*/, const int32_t *ulen /*[U] length by sorted position*/,
                      PairRec *pairs, hipStream_t st);
void launch_fill_ulen(int32_t U, const int32_t *sorted_uniq, const int32_t *seed_read, const int32_t *len, int32_t *ulen, hipStream_t st);

// ---- k_float.hip
struct WaveDesc {               // one wave = up to 64 items that share a profile
  int32_t prof;
  int32_t count;
  int64_t first;                // first item index
  int64_t slab;                 // slab offset of this wave, in rows of 64 lanes
  int32_t rows;                 // slab rows reserved per field block
  int32_t pad;
};
struct VitOut { float vitsc; int32_t ran, pass; };      // k_vit.hip: verdict of the Viterbi filter for one pair
// hmmsearch has no limit on the regions of a target (itsxpress/SeqSample.py:191-209 runs it on concatemers like on anything
// else): a pair's first MAXDOM regions sit in its raw slots, the rest -- rare by construction -- go to this list, (pair, k) tagged
struct RegionPool { RegionRec *rec; int32_t *k; unsigned long long *n; int64_t cap; };
__device__ __forceinline__ void pool_put(const RegionPool &p, const RegionRec &r, int k)
{
  const unsigned long long i = atomicAdd(p.n, 1ULL);
  if ((int64_t)i < p.cap) { p.rec[i] = r; p.k[i] = k; }
}
// after the list has been ordered by (pair, k): region k >= MAXDOM of pair pi (binary search; only pairs past their slots come here)
struct RegionPoolView { const RegionRec *rec; const unsigned long long *key; int64_t n; };
__device__ __forceinline__ RegionRec raw_region(const RegionRec *raw, const RegionPoolView &pv, int64_t pi, int k)
{
  if (k < MAXDOM) return raw[pi * MAXDOM + k];
  const unsigned long long want = ((unsigned long long)pi << 24) | (unsigned long long)k;
  int64_t lo = 0, hi = pv.n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (pv.key[mid] < want) lo = mid + 1; else hi = mid; }
  return pv.rec[lo];
}
struct FloatArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;
  const DevProfile *prof;
  const LenTables *lt;
  const float *flogsum;         // [16000]
  const LogTab *logtab;         // [LOGTAB_N] detmath.h: the bias filter's table-driven logarithm
  const PairRec *pairs;
  PairOut *pout;
  const WaveDesc *waves;
  float *slab;                  // xmx rows: two planes of [row][6][64] (k_float.hip: SLAB)
  int64_t slab_plane;           // floats between the planes = rows of this batch x 6 x 64
  RegionRec *regions;           // [npairs][MAXDOM] raw, before compaction
  RegionPool pool;              // regions past a pair's MAXDOM slots
  const VitOut *vit;            // Viterbi filter verdicts (null when the filter did not run: F2 >= F1)
  double F1, F3;
};
// ---- k_vit.hip: the Viterbi filter (runs only when F2 < F1; never under the reference's flags)
struct VitArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;
  const DevProfile *prof;
  const LenTables *lt;
  const PairRec *pairs;
  const PairOut *pout;
  const WaveDesc *waves;
  const int16_t *vtab;          // [P][VIT_TAB]: per profile 8 x 47 transition words then 16 x 47 emission words
  VitOut *vit;                  // [npairs]
  int32_t eloop;                // wordify(logf(0.5))
  double F2;
};
void launch_vit(const VitArgs &a, int nwaves, int wave0, hipStream_t st);
void launch_bias(const FloatArgs &a, int64_t npairs, hipStream_t st, int lds_pad = 0);
void launch_filters_fwd(const FloatArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st);
// the lazy domain stage's first pass (k_lazy.hip): Forward scores only, nothing goes to the slab or to PairOut; fb[pair] = fwdsc
void launch_fwd_bound(const FloatArgs &a, float *fb, int nwaves, int wave0, int generic_q, hipStream_t st);
// the same score by the node-sequential fused-multiply-add kernel (k_lazy.hip: k_fwd_bound); btab = [P][BOUND_PAIRS][16] floats per
// pair of nodes (2j + 1, 2j + 2): {MM IM DM out of the node (into the next one's match cell)} {MI II BM MD of the node} as pairs, DD x 2
void launch_fwd_bound_seq(const FloatArgs &a, const float *btab, bool fold, float *fb, int nwaves, int wave0, hipStream_t st);
// ... over waves of chains that start at sl.depth (k_share.hip): the rows before come from the parent chain's saved state
void launch_fwd_bound_share(const FloatArgs &a, const float *btab, bool fold, float *fb, int nwaves, int wave0, const ShareLaunch &sl, hipStream_t st);
// round 6: the Backward chains of the two-sided schedule (folded units only; rtab: engine.hip, BOUND_RTAB floats per profile)
void launch_bwd_bound_share(const FloatArgs &a, const float *btab, const float *rtab, int nwaves, int wave0, const ShareLaunch &sl, hipStream_t st);
void launch_bwd_decode(const FloatArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st);
void launch_decode(const FloatArgs &a, int nwaves, int wave0, hipStream_t st);

struct EnvArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;
  const DevProfile *prof;
  const LenTables *lt;
  const PairRec *pairs;
  const RegionRec *regions;     // compacted, grouped by profile
  RegionOut *rout;
  const WaveDesc *waves;
  float *slab;                  // per wave: [row][2 passes][96][64]
};
void launch_envelopes(const EnvArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st);

// ---- k_ensemble.hip: multidomain regions (stochastic traceback clustering)
constexpr int MRV = 52;            // float4 vectors per matrix row: {M D I B} of node 0 (zeros) .. 48 | (E N J B) | (E J C SCALE) | -
constexpr int MRENV = 4;           // envelopes kept per clustered region
#ifndef ITSX_MR_LANES
#define ITSX_MR_LANES 64
#endif
constexpr int MR_LANES = ITSX_MR_LANES;       // regions per wave (64 since the matrix went node-major: same time as 32, half the scratch)
constexpr int MR_LANES_LONG = 4;   // ... and per wave of the longest regions (engine.hip: they decide when a batch ends)
constexpr double MR_LONG_FRAC = 0.02;   // the share of a batch's regions that counts as longest (measured: 0.01-0.05 alike, 2 or 4 lanes alike)
constexpr int MR_MAXD = 8;         // domains in one sampled path
constexpr int MR_TCAP = 512;       // distinct sampled (i, j, k, m) tuples per region
constexpr int MR_SCAP = 200 * MR_MAXD;
constexpr int MR_NSIG = 32;        // clusters that reach the posterior threshold: each holds >= 50 of the <= 200 x MR_MAXD sampled domains, so 32 always suffice
static_assert(MR_NSIG * 50 >= 200 * MR_MAXD, "MR_NSIG must hold every cluster that can reach 25 % of 200 paths");
constexpr int MR_HASH = 1024;      // slots of the per-region tuple index (a power of two >= 2 MR_TCAP)
constexpr int MR_EPC = 256;        // widest endpoint histogram kept as an array (wider ones are counted pairwise)
constexpr int MR_SCRATCH = 19456;  // bytes of per-region bookkeeping (k_ensemble.hip: MrScratch)
struct MrRec { int32_t pair, ireg, jreg, slot; };                         // field for field a RegionRec (the memoisation kernels take it as one)
// status 0 = resolved; 2 / 4 / 7 = a bookkeeping limit of the fast kernel was hit (the region then goes through the overflow path, whose
// arrays are sized by the region's length and cannot overrun); 1 / 5 = the matrix could not be sampled (hmmsearch itself throws there).
// envelopes relative to the region (1-based): the first MRENV here, all nenv of them at envpool[big ...] when the overflow path ran (big >= 0)
struct MrOut { int32_t status, nenv; int32_t ei[MRENV], ej[MRENV]; int64_t big; };
struct MrBig { int64_t off, envoff; int32_t capD, capT; uint32_t hmask; int32_t pad; };    // one region's block of the overflow arena (engine.hip sizes it)
struct MrArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;
  const DevProfile *prof;
  const PairRec *pairs;
  const MrRec *mr;              // all multidomain regions of the chunk, in pair order
  const int32_t *ulist;         // distinct regions (same profile, target length and residues): index into mr of each one's first copy
  int64_t u0;                   // first distinct region of this batch (scratch blocks are per batch)
  const WaveDesc *waves;        // first = index into ulist, count, rows = longest region of the wave + 1
  int32_t lds_bytes = 0;        // k_mr_trace<., true>: LDS per wave for the region's matrix
  float4 *slab;                 // a region's matrix is contiguous: [row 0..Lr][MRV]; region u starts at row rowoff[u] - rowoff0
  const int64_t *rowoff;        // [distinct] first slab row of each distinct region (all batches; rowoff0 = the batch's first)
  int64_t rowoff0;
  const int64_t *n2off;         // [distinct] offset of each distinct region's per-residue null2 scores
  float *n2sc;
  MrOut *out;                   // [distinct]
  uint8_t *scratch;
  // the overflow path (regions the fast kernel could not hold): lanes take the regions sel[first + lane], whose matrices start at slab
  // row sel_rowoff[first + lane], with per-region arrays in `arena`
  const int32_t *sel; const int64_t *sel_rowoff; const MrBig *big; uint8_t *arena;
  int32_t *envpool;             // the overflow path's envelope lists (they outlive the arena: read when the chunk's region list is built)
  unsigned long long *dbg;      // optional [4]: wave-clock ticks spent walking paths, closing them, clustering (ITSX_MR_DEBUG)
};
void launch_mr_count(const PairOut *pout, const RegionRec *raw, RegionPoolView pv, int64_t npairs, int32_t *cnt, hipStream_t st);
void launch_mr_fill(const PairOut *pout, const RegionRec *raw, RegionPoolView pv, int64_t npairs, const int32_t *off, MrRec *mr, hipStream_t st);
// distinct regions: ulist[urank[m]] = m and ulen = region length for the first copies; mr_u[m] = index of m's distinct region
void launch_mr_ulist(int64_t nmr, const MrRec *mr, const int32_t *rep, const int32_t *is_uniq, const int32_t *urank, int32_t *ulist, int32_t *ulen,
                     int32_t *mr_u, hipStream_t st);
// regions ordered by length: ulist_out[newpos[u]] = ulist_in[u], mr_u[m] = newpos[mr_u[m]]
void launch_mr_reorder(int64_t nu, int64_t nmr, const int32_t *newpos, const int32_t *ulist_in, int32_t *ulist_out, int32_t *mr_u, hipStream_t st);
void launch_mr_ensemble(const MrArgs &a, int nwaves, int wave0, hipStream_t st, int64_t one_first = 0, int64_t one_count = 0);
void launch_mr_ensemble_big(const MrArgs &a, int nwaves, hipStream_t st);
// distinct regions the fast kernel gave up on (status 2, 4, 7 -- and 3, 6, which cannot occur): appended to list, counted in n[0]
void launch_mr_overflowed(const MrOut *out, int64_t nu, int32_t *list, unsigned long long *n, hipStream_t st);
// A pair's final envelope list: its regions in order, every clustered region replaced by its envelopes (k_mr_apply's job until round 3,
// now without a fixed-size buffer: counted, then written straight into the profile-grouped list).  cnt[pi] = envelopes (pout.ndom is set
// to it); counters: [0] regions whose matrix could not be sampled, [1] cluster envelopes, [2 + status] such regions by kind
struct RegionListArgs {
  PairOut *pout; const RegionRec *raw; RegionPoolView pv; int64_t npairs;
  const int32_t *mr_off; const int32_t *mr_u; const MrOut *mrout; const int32_t *envpool;     // null mr_off: no clustered regions in this chunk
};
void launch_region_list_count(const RegionListArgs &a, int32_t *cnt, unsigned long long *counters, hipStream_t st);
void launch_region_list_fill(const RegionListArgs &a, const int64_t *pair_region0, RegionRec *out, hipStream_t st);

struct ScoreArgs {
  ReadsDev rd;
  const int32_t *sorted_uniq, *seed_read;
  const DevProfile *prof;
  const LenTables *lt;
  const float *flogsum;
  const PairRec *pairs;
  const PairOut *pout;
  const RegionRec *regions;     // compacted
  const RegionOut *rout;
  const int64_t *pair_region0;  // [npairs] index of the pair's first compacted region
  const int32_t *upos;          // [nregions] index into rout (distinct envelopes) of each region's result
  itsx_domain *dom;             // one per region
  int64_t npairs;
  double T;
  // clustered regions of the chunk (null when it has none): a pair's regions are mr[mr_off[pi] .. mr_off[pi + 1])
  // (mr_u[m] = the distinct region whose results m shares: mrout and n2off are indexed by it)
  const MrRec *mr; const int32_t *mr_u; const MrOut *mrout; const int64_t *n2off; const float *n2sc; const int32_t *mr_off;
  int32_t *domz;                // [S][P] reported targets per (sample, profile); S = 1 without per-sample batching
  const int32_t *usample;       // [U] sample of each unique (null: one sample)
  int32_t P;
};
void launch_score(const ScoreArgs &a, hipStream_t st);
void launch_region_count_fill(const PairOut *pout, const RegionRec *raw, int64_t npairs, const int32_t *pref,
                              const int64_t *seg_pair_start, const int64_t *seg_region_start, const int32_t *seg_of_pair_base,
                              RegionRec *out, hipStream_t st);

// ItsPosition's order on domain rows (itsxpress/SeqSample.py:400-429: the FIRST row with a strictly greater %.1f score wins; rows come
// in profile order, then domain order): [24b tenths + 2^23][20b ~profile][20b ~domain index].  A larger key wins.
__host__ __device__ __forceinline__ unsigned long long rank_key(const itsx_domain &d)
{
  long long tenths = (long long)__builtin_rint((double)d.bitscore * 10.0) + (1ll << 23);
  if (tenths < 0) tenths = 0;
  if (tenths > (1ll << 24) - 1) tenths = (1ll << 24) - 1;
  const unsigned long long di = (unsigned long long)(d.dom_idx < 0 ? 0 : (d.dom_idx > 0xFFFFF ? 0xFFFFF : d.dom_idx));
  return ((unsigned long long)tenths << 40) | ((unsigned long long)(0xFFFFF - (d.prof & 0xFFFFF)) << 20) | (0xFFFFFull - di);
}

// ---- k_lazy.hip: the lazy domain stage (pairs that cannot win ItsPosition's argmax never reach Backward)
constexpr int LAZY_TENTHS_BIAS = 1 << 23;      // the bias of the tenths field in rank_key
struct LazyArgs {
  const PairRec *pairs; int64_t NP;
  const float *fb;                 // [NP] Forward score of the bound pass (nats; need not be HMMER's bits: a margin covers it)
  const LenTables *lt;
  const int8_t *cls; int32_t ncls; // 2-character prefix class of each profile
  const int32_t *sorted_uniq;      // chunk-relative useq -> global unique index
  uint32_t *b10;                   // [NP] largest %.1f tenths (+ LAZY_TENTHS_BIAS) a domain of the pair can print; 0 = no pair here
  unsigned long long *gtop;        // [chunk uniques x ncls] (b10 << 32 | ~pair) of the group's best-bound pair
  const unsigned long long *bestc; // [uniques x ncls] rank key of the best CERTAIN row so far (k_compact_best)
  uint8_t *done;                   // [NP] the pair went through the domain stage already
  int32_t *flag;                   // [NP + 1] selected for this round
};
void launch_lazy_bound(const LazyArgs &a, hipStream_t st);
void launch_lazy_mark(const LazyArgs &a, int round, hipStream_t st);
void launch_lazy_scatter(const PairRec *pairs, int64_t NP, const int32_t *flag, const int32_t *pos, const int64_t *seg_old, const int64_t *seg_new,
                         PairRec *out, hipStream_t st);
// three-valued thresholds with bounds on hmmsearch's domZ: dom_reported = 1 reported for every domZ in [zlb, zub], 0 for none, 2 = depends
void launch_finalize_lazy(itsx_domain *dom, int64_t n, const int64_t *zlb, const int64_t *zub, double domE, const int32_t *usample, int P, hipStream_t st);
// best sure row per (representative, class) and "has a sure row" per representative; then the rows whose status matters
void launch_lazy_sure(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, unsigned long long *sure, int32_t *has, hipStream_t st);
void launch_lazy_pending(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, const unsigned long long *sure, const int32_t *has,
                         unsigned long long *count, int32_t *prof_flag, uint8_t *uniq_flag, unsigned long long *zneed /*[2 P] or nullptr*/, const unsigned long long *zsplit /*[P]*/, double domE, hipStream_t st);
void launch_topup_count(const uint8_t *done, const int64_t *seg_start, const int32_t *total, const int32_t *prof_of_slot, int nslots, unsigned long long *cnt, hipStream_t st);
void launch_topup_keys(const PairRec *pairs, int64_t NP, const uint8_t *done, const uint32_t *b10, const int32_t *slot_of_prof, int32_t *flag, const int32_t *pos /*nullptr: flags only*/,
                       unsigned long long *keys, int32_t *vals, hipStream_t st);
void launch_topup_mark(const PairRec *pairs, int64_t NP, uint8_t *done, const uint32_t *b10, const int32_t *slot_of_prof, const uint32_t *cutoff, int32_t *flag, hipStream_t st);

}  // namespace itsx
