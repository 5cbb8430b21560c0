/*
 * orc_cluster.c -- ORACLE (test infrastructure only): greedy centroid clustering, restating
 *   vsearch --cluster_size IN --centroids rep.fa --uc uc.txt --strand both --id X
 * (reference call site itsxpress/SeqSample.py:147-162; vsearch >= 2.21.1 is an un-vendored
 * dependency, recipes/itsxpress/meta.yaml:37; neither its source nor a binary is available here).
 *
 * PARITY UNPINNED: the reference's tests hold no fixture for this path (only a no-assert smoke test,
 * tests/test_main_pytest.py:174-180).  What follows restates vsearch's documented procedure and
 * defaults; the engine (itsxpress_amd/csrc/k_cluster.hip) must equal THIS bit for bit.
 *
 * The procedure
 *  1. reads shorter than --minseqlength (32) or longer than --maxseqlength (50000) are dropped; the rest are processed in order of
 *     decreasing abundance (all 1: FASTQ input carries no ;size=), ties by label (strcmp), then by
 *     input position;
 *  2. a query is compared with the existing centroids on both strands.  Per strand: the DISTINCT
 *     8-mers (--wordlength 8) of the query that contain no ambiguity symbol are looked up in the
 *     centroid index (each centroid contributes its distinct forward 8-mers); a centroid is a
 *     candidate when it shares at least min(12, #query words) words (--minwordmatches 12).
 *     Candidates are tried in the order (shared words descending, centroid length ascending,
 *     centroid processing position ascending) until one is accepted (--maxaccepts 1) or 32 have been
 *     rejected (--maxrejects 32);
 *  3. a candidate is tried by a global alignment: match +2, mismatch -4, any pair with an ambiguity
 *     symbol 0; a gap of k columns costs 20 + 2k inside the alignment and 2 + k at either end
 *     (--gapopen 20I/2E --gapext 2I/1E); identity = 100 * matches / (alignment columns - terminal
 *     gap columns) (--iddef 2), a column being a match when the two IUPAC symbols are compatible;
 *     accepted iff identity >= 100 * X;
 *  4. of the two strands' accepted hits the one with the higher identity wins (plus on a tie); a
 *     query with no accepted hit becomes a new centroid.
 *
 * Documented choices where the published description leaves freedom (identical in the engine):
 *  - among the optimal-score alignments the one with the most matches, then the fewest counted
 *    columns, defines the identity (vsearch takes whatever its traceback yields); this makes the
 *    identity a function of the DP alone: every cell carries (score, matches, columns) packed in one
 *    64-bit integer so that integer max is the lexicographic max;
 *  - DUST soft-masking of the seeds (vsearch's default --qmask dust / --dbmask dust: orc_dust below) is ON unless the
 *    environment holds ORC_QMASK=none; it only removes words from the k-mer sets, the alignment sees every symbol;
 *  - labels are the identifiers up to the first blank;
 *  - the uc CIGAR column is not produced (the consumer, Dedup.parse SeqSample.py:542-562, reads columns
 *    0, 8 and 9 only).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* 4-bit IUPAC sets for the digital codes A C G T - R Y M K S W H B V D N */
static const uint8_t MASK4[16] = { 1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15 };
static inline int revmask(int m) { return ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3); }
static inline int unamb(int m) { return m == 1 || m == 2 || m == 4 || m == 8; }
static inline int code2(int m) { return m == 1 ? 0 : m == 2 ? 1 : m == 4 ? 2 : 3; }

#define SH_S 40
#define SH_M 20
#define NEGV (-(1LL << 60))
#define ONE_S (1LL << SH_S)
#define ONE_M (1LL << SH_M)
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

/* global alignment of query masks q[0..Lq) with target masks t[0..Lt): returns matches and counted columns */
void orc_align_identity(const uint8_t *q, int Lq, const uint8_t *t, int Lt, int64_t *ret_score, int64_t *ret_matches, int64_t *ret_cols)
{
  const int64_t GOI = -22 * ONE_S - 1, GEI = -2 * ONE_S - 1;      /* interior: open+first extension, extension; counted */
  const int64_t GOT = -3 * ONE_S, GET = -1 * ONE_S;               /* terminal: not counted */
  int64_t *H = (int64_t *)malloc(sizeof(int64_t) * (size_t)(Lt + 1));
  int64_t *F = (int64_t *)malloc(sizeof(int64_t) * (size_t)(Lt + 1));
  for (int i = 0; i <= Lq; i++) {
    const int64_t goE = (i == 0 || i == Lq) ? GOT : GOI, geE = (i == 0 || i == Lq) ? GET : GEI;
    int64_t Hleft = NEGV, Eleft = NEGV, diag = NEGV;
    for (int j = 0; j <= Lt; j++) {
      const int64_t goF = (j == 0 || j == Lt) ? GOT : GOI, geF = (j == 0 || j == Lt) ? GET : GEI;
      const int64_t upH = i > 0 ? H[j] : NEGV, upF = i > 0 ? F[j] : NEGV;
      const int64_t E = max64(Hleft + goE, Eleft + geE);
      const int64_t Fv = max64(upH + goF, upF + geF);
      int64_t D = 0;
      if (i > 0 && j > 0) {
        const int a = q[i - 1], b = t[j - 1];
        if (unamb(a) && unamb(b)) D = (a == b) ? (2 * ONE_S + ONE_M - 1) : (-4 * ONE_S - 1);
        else D = (a & b) ? (ONE_M - 1) : -1;
      }
      int64_t Hn = max64(max64(diag + D, E), Fv);
      if (i == 0 && j == 0) Hn = 0;
      diag = upH;
      H[j] = Hn; F[j] = Fv;
      Hleft = Hn; Eleft = E;
    }
  }
  const int64_t v = H[Lt];
  const int64_t score = (v + (1LL << (SH_S - 1))) >> SH_S;
  const int64_t low = v - score * ONE_S;
  const int64_t matches = (low + (1LL << (SH_M - 1))) >> SH_M;
  *ret_score = score; *ret_matches = matches; *ret_cols = matches * ONE_M - low;
  free(H); free(F);
}

/*
 * orc_dust -- the DUST low-complexity filter as vsearch applies it (mask.cc: dust() / wo(), after Tatusov & Lipman's `dust`;
 * restated from the published algorithm -- PARITY UNPINNED like the rest of this file): windows of 64 symbols that advance
 * by 32; inside a window, for every start i and every end j the score 10 * sum / j, where sum adds, for each 3-mer met
 * again, the number of times it was met before (3-mers as 2-bit codes, anything but A C G T counts as A); the best-scoring
 * interval of a window is masked when its score exceeds 20 (the first best one in (i, j) order); after a masked window that
 * ends in its first half the next window starts right behind it (i += 32 - b).  masked[pos] = 1 for soft-masked symbols.
 * vsearch removes from the k-mer sets every word that touches a masked symbol, exactly like words with ambiguity symbols.
 */
static int dust_wo(int len, const uint8_t *s, int *beg, int *end)
{
  const int l1 = len - 3 + 1 - 5;                 /* the smallest possible region is 8 symbols */
  *beg = 0; *end = 0;
  if (l1 < 0) return 0;
  int bestv = 0, besti = 0, bestj = 0;
  int counts[64], words[64];
  int word = 0;
  for (int j = 0; j < len; j++) { word = (word << 2) | (s[j] < 4 ? s[j] : 0); words[j] = word & 63; }
  for (int i = 0; i < l1; i++) {
    memset(counts, 0, sizeof(counts));
    int sum = 0;
    for (int j = 2; j < len - i; j++) {
      const int w = words[i + j];
      const int c = counts[w];
      if (c) {
        sum += c;
        const int v = 10 * sum / j;
        if (v > bestv) { bestv = v; besti = i; bestj = j; }
      }
      counts[w]++;
    }
  }
  *beg = besti; *end = besti + bestj;
  return bestv;
}
void orc_dust(const uint8_t *codes, int64_t L, uint8_t *masked)
{
  memset(masked, 0, (size_t)L);
  for (int64_t i = 0; i < L; i += 32) {
    const int l = (L > i + 64) ? 64 : (int)(L - i);
    int a, b;
    const int v = dust_wo(l, codes + i, &a, &b);
    if (v > 20) {
      for (int64_t j = a + i; j <= b + i; j++) masked[j] = 1;
      if (b < 32) i += 32 - b;
    }
  }
}
static int qmask_dust(void) { const char *e = getenv("ORC_QMASK"); return !(e && strcmp(e, "none") == 0); }

/* distinct unambiguous, unmasked 8-mers of a mask sequence, first base in the low bits; returns count, fills bitmap[65536/8];
 * msk (may be NULL): soft-masked positions, in the order of m */
static int kmer_set(const uint8_t *m, const uint8_t *msk, int L, uint8_t *bitmap, uint16_t *list)
{
  memset(bitmap, 0, 8192);
  int n = 0, good = 0;
  uint32_t w = 0;
  for (int i = 0; i < L; i++) {
    if (unamb(m[i]) && !(msk && msk[i])) { w = (w >> 2) | ((uint32_t)code2(m[i]) << 14); good++; }
    else { good = 0; w = 0; }
    if (good >= 8) {
      const uint32_t k = w & 0xffffu;
      if (!(bitmap[k >> 3] & (1u << (k & 7)))) { bitmap[k >> 3] |= (uint8_t)(1u << (k & 7)); list[n++] = (uint16_t)k; }
    }
  }
  return n;
}

typedef struct { int32_t *v; int32_t n, cap; } post_t;
typedef struct { uint64_t key; int32_t c; } cand_t;
static int cand_cmp(const void *a, const void *b)
{
  const uint64_t x = ((const cand_t *)a)->key, y = ((const cand_t *)b)->key;
  return x > y ? -1 : x < y ? 1 : 0;
}

typedef struct { const char *lab; int64_t lablen; int64_t idx; } ord_t;
static int ord_cmp(const void *a, const void *b)
{
  const ord_t *x = (const ord_t *)a, *y = (const ord_t *)b;
  const int64_t m = x->lablen < y->lablen ? x->lablen : y->lablen;
  const int r = m ? memcmp(x->lab, y->lab, (size_t)m) : 0;   /* strcmp on NUL-free labels */
  if (r) return r;
  if (x->lablen != y->lablen) return x->lablen < y->lablen ? -1 : 1;
  return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;
}

/*
 * codes: digital codes 0..15; offsets[n+1]; labels/label_offsets may be NULL (then input order).
 * Out (all [n]): rep_of = read index of the centroid (itself for a centroid, -1 if dropped), strand (+1/-1),
 * pct_id (identity of an H row; -1 for centroids and dropped reads), order = kept reads in processing order.
 * stats[0] = alignments computed, stats[1] = centroids.  Returns the number of kept reads.
 */
int64_t orc_cluster(const uint8_t *codes, const int64_t *offsets, int64_t n, const char *labels, const int64_t *label_offsets,
                    double id, int strand_both, int minlen, int64_t *rep_of, int8_t *strand, double *pct_id, int64_t *order,
                    int64_t *stats)
{
  ord_t *ord = (ord_t *)malloc(sizeof(ord_t) * (size_t)(n + 1));
  int64_t nk = 0, Lmax = 1;
  for (int64_t r = 0; r < n; r++) {
    const int64_t L = offsets[r + 1] - offsets[r];
    rep_of[r] = -1; strand[r] = 1; pct_id[r] = -1.0;
    if (L < minlen || L > 50000) continue;                 /* --minseqlength 32, --maxseqlength 50000 (vsearch defaults) */
    if (L > Lmax) Lmax = L;
    ord[nk].idx = r;
    ord[nk].lab = labels ? labels + label_offsets[r] : "";
    ord[nk].lablen = labels ? label_offsets[r + 1] - label_offsets[r] : 0;
    nk++;
  }
  qsort(ord, (size_t)nk, sizeof(ord_t), ord_cmp);

  post_t *post = (post_t *)calloc(65536, sizeof(post_t));
  int32_t *cent_pos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nk + 1));   /* centroid -> processing position */
  int32_t *cnt = (int32_t *)calloc((size_t)nk + 1, sizeof(int32_t));
  int32_t *touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nk + 1));
  cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(nk + 1));
  uint8_t *qm = (uint8_t *)malloc((size_t)Lmax), *tm = (uint8_t *)malloc((size_t)Lmax);
  /* soft masks: of the forward strand of every read (a query's reverse strand carries the reversed mask: vsearch masks the
   * query once and reverse-complements the masked copy; a centroid is indexed with its forward mask) */
  const int use_dust = qmask_dust();
  uint8_t *dmask = use_dust ? (uint8_t *)malloc((size_t)(offsets[n] + 1)) : NULL, *qmsk = (uint8_t *)malloc((size_t)Lmax);
  if (use_dust) for (int64_t r = 0; r < n; r++) orc_dust(codes + offsets[r], offsets[r + 1] - offsets[r], dmask + offsets[r]);
  uint8_t *bitmap = (uint8_t *)malloc(8192);
  uint16_t *klist = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)Lmax);
  int32_t C = 0;
  int64_t naln = 0;
  const double thr = 100.0 * id;

  for (int64_t p = 0; p < nk; p++) {
    const int64_t r = ord[p].idx;
    const int Lq = (int)(offsets[r + 1] - offsets[r]);
    order[p] = r;
    int best_c = -1, best_strand = 1; double best_id = -1.0;
    for (int s = 0; s < (strand_both ? 2 : 1); s++) {
      if (s == 0) for (int i = 0; i < Lq; i++) qm[i] = MASK4[codes[offsets[r] + i]];
      else for (int i = 0; i < Lq; i++) qm[i] = (uint8_t)revmask(MASK4[codes[offsets[r] + Lq - 1 - i]]);
      if (use_dust) for (int i = 0; i < Lq; i++) qmsk[i] = dmask[offsets[r] + (s == 0 ? i : Lq - 1 - i)];
      const int nq = kmer_set(qm, use_dust ? qmsk : NULL, Lq, bitmap, klist);
      if (nq == 0) continue;
      const int minm = nq < 12 ? nq : 12;
      int nt = 0;
      for (int a = 0; a < nq; a++) {
        const post_t *pl = &post[klist[a]];
        for (int32_t b = 0; b < pl->n; b++) { const int32_t c = pl->v[b]; if (cnt[c]++ == 0) touched[nt++] = c; }
      }
      int nc = 0;
      for (int a = 0; a < nt; a++) {
        const int32_t c = touched[a];
        if (cnt[c] >= minm) {
          const int64_t cr = ord[cent_pos[c]].idx;
          const uint64_t len = (uint64_t)(offsets[cr + 1] - offsets[cr]);
          cand[nc].key = ((uint64_t)cnt[c] << 48) | ((uint64_t)(65535 - len) << 32) | (uint64_t)(0xffffffffu - (uint32_t)cent_pos[c]);
          cand[nc].c = c; nc++;
        }
        cnt[c] = 0;
      }
      qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
      int rejects = 0;
      for (int a = 0; a < nc && rejects < 32; a++) {
        const int64_t cr = ord[cent_pos[cand[a].c]].idx;
        const int Lt = (int)(offsets[cr + 1] - offsets[cr]);
        for (int i = 0; i < Lt; i++) tm[i] = MASK4[codes[offsets[cr] + i]];
        int64_t sc, m, cols;
        orc_align_identity(qm, Lq, tm, Lt, &sc, &m, &cols);
        naln++;
        const double pid = cols > 0 ? 100.0 * (double)m / (double)cols : 0.0;
        if (pid >= thr) {
          if (best_c < 0 || pid > best_id) { best_c = cand[a].c; best_id = pid; best_strand = s == 0 ? 1 : -1; }
          break;
        }
        rejects++;
      }
    }
    if (best_c >= 0) {
      rep_of[r] = ord[cent_pos[best_c]].idx; strand[r] = (int8_t)best_strand; pct_id[r] = best_id;
    } else {
      rep_of[r] = r;
      for (int i = 0; i < Lq; i++) qm[i] = MASK4[codes[offsets[r] + i]];
      const int nq = kmer_set(qm, use_dust ? dmask + offsets[r] : NULL, Lq, bitmap, klist);
      for (int a = 0; a < nq; a++) {
        post_t *pl = &post[klist[a]];
        if (pl->n == pl->cap) { pl->cap = pl->cap ? pl->cap * 2 : 4; pl->v = (int32_t *)realloc(pl->v, sizeof(int32_t) * (size_t)pl->cap); }
        pl->v[pl->n++] = C;
      }
      cent_pos[C++] = (int32_t)p;
    }
  }
  if (stats) { stats[0] = naln; stats[1] = C; }
  for (int k = 0; k < 65536; k++) free(post[k].v);
  free(post); free(cent_pos); free(cnt); free(touched); free(cand); free(qm); free(tm); free(bitmap); free(klist); free(ord); free(dmask); free(qmsk);
  return nk;
}

/*
 * orc_orient -- ORACLE (test infrastructure only): read orientation, restating
 *   vsearch --orient IN --db universal_orient_ref_clean.fasta.gz --fastqout oriented.fq
 * (reference call site itsxpress/SeqSample.py:48-91).  PARITY UNPINNED (the reference holds no
 * test or fixture for it).  The procedure: the distinct unambiguous 12-mers (--wordlength 12) of the query and of its
 * reverse complement are looked up in the set of 12-mers of the database sequences; with count_fwd / count_rev hits the
 * read is forward when count_fwd >= 1 and count_fwd >= 4 * count_rev, reverse (to be reverse-complemented) when
 * count_rev >= 1 and count_rev >= 4 * count_fwd, otherwise undetermined (not written).  Query and database sequences are
 * DUST-masked first (vsearch's defaults --qmask dust --dbmask dust; orc_dust above): words that touch a masked symbol are
 * left out on both sides; ORC_QMASK=none switches the masking off.
 *   dbbits: 4^12 bits (2 MB), bit k set when 12-mer k (first base in the low bits) occurs in the database.
 */
static inline uint32_t rc24(uint32_t k)
{
  uint32_t x = ~k & 0xffffffu, y = 0;
  for (int t = 0; t < 12; t++) { y = (y << 2) | (x & 3u); x >>= 2; }
  return y;
}
void orc_orient_db_add(uint8_t *dbbits, const uint8_t *codes, int64_t L)
{
  uint32_t w = 0; int good = 0;
  uint8_t *msk = qmask_dust() ? (uint8_t *)malloc((size_t)L + 1) : NULL;
  if (msk) orc_dust(codes, L, msk);
  for (int64_t i = 0; i < L; i++) {
    if (codes[i] < 4 && !(msk && msk[i])) { w = (w >> 2) | ((uint32_t)codes[i] << 22); good++; } else { good = 0; w = 0; }
    if (good >= 12) dbbits[w >> 3] |= (uint8_t)(1u << (w & 7));
  }
  free(msk);
}
void orc_orient(const uint8_t *dbbits, const uint8_t *codes, const int64_t *offsets, int64_t n, int8_t *strand, int32_t *cfwd, int32_t *crev)
{
  uint8_t *seen = (uint8_t *)calloc(1u << 21, 1);
  uint32_t *list = NULL; int64_t cap = 0, mcap = 0;
  uint8_t *msk = NULL; const int use_dust = qmask_dust();
  for (int64_t r = 0; r < n; r++) {
    const int64_t L = offsets[r + 1] - offsets[r];
    if (L > cap) { cap = L; list = (uint32_t *)realloc(list, sizeof(uint32_t) * (size_t)cap); }
    int64_t nk = 0; uint32_t w = 0; int good = 0;
    if (use_dust) { if (L > mcap) { mcap = L; msk = (uint8_t *)realloc(msk, (size_t)mcap + 1); } orc_dust(codes + offsets[r], L, msk); }
    for (int64_t i = 0; i < L; i++) {
      const uint8_t c = codes[offsets[r] + i];
      if (c < 4 && !(use_dust && msk[i])) { w = (w >> 2) | ((uint32_t)c << 22); good++; } else { good = 0; w = 0; }
      if (good >= 12 && !(seen[w >> 3] & (1u << (w & 7)))) { seen[w >> 3] |= (uint8_t)(1u << (w & 7)); list[nk++] = w; }
    }
    int32_t f = 0, v = 0;
    for (int64_t a = 0; a < nk; a++) {
      const uint32_t k = list[a], kr = rc24(k);
      f += (dbbits[k >> 3] >> (k & 7)) & 1;
      v += (dbbits[kr >> 3] >> (kr & 7)) & 1;
      seen[k >> 3] &= (uint8_t)~(1u << (k & 7));
    }
    cfwd[r] = f; crev[r] = v;
    strand[r] = (f >= 1 && f >= 4 * v) ? 1 : (v >= 1 && v >= 4 * f) ? -1 : 0;
  }
  free(seen); free(list); free(msk);
}
