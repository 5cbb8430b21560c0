/*
 * orc_merge.c -- ORACLE (test infrastructure only): paired-end read merging, restating
 *   vsearch --fastq_mergepairs R1 --reverse R2 --fastqout seq.fq --fastq_maxdiffs 40 --fastq_maxee 2
 *           --fastq_qmax 93 [--fastq_allowmergestagger]
 * (reference call site itsxpress/SeqSample.py:266-365, constants itsxpress/definitions.py:79-82; vsearch >= 2.21.1
 * is an un-vendored dependency, recipes/itsxpress/meta.yaml:37).  SURVEY section 8f row f2.
 *
 * PARITY UNPINNED.  The reference's merged-reads fixture (tests/test_data/4774-1-MSITS3_merged.fastq, 227 reads) was
 * made by BBMerge in the reference's 1.x days (its overlap qualities follow max + min/4 and max - min), not by
 * vsearch; the reference's end-to-end test expects 235 trimmed reads from the vsearch merge
 * (tests/test_main_pytest.py:252,346).  Anchors used by tests/test_merge_cpu.py: of the 250 fixture pairs this
 * restatement merges 236; on the 226 pairs both tools merge the merged SEQUENCES are identical in 225.
 *
 * The procedure (vsearch's, with its defaults: ascii 33, qmin 0, qmaxout 41, minovlen 10, maxdiffpct 100):
 *  - candidate alignments are the ungapped diagonals on which the forward read and the reverse-complemented
 *    reverse read share at least 4 5-mers (5-mers holding an N do not count);
 *  - a diagonal's score is the sum over the overlap, walked from the forward read's 3' end, of
 *    log2(P(observed pair | same base) / 0.25): match  p = 1 - px - py + 4 px py / 3,
 *    mismatch p = (px + py)/3 - 4 px py / 9, px = 10^(-q/10) (0.75 below q 2); a diagonal whose running score
 *    ever falls 16 or more below its running maximum is discarded; the best diagonal must score >= 16 and be
 *    the only one scoring >= 16; at most maxdiffs mismatches; overlap >= 10; no staggered pairs unless allowed;
 *  - merged base/quality in the overlap (Edgar & Flyvbjerg 2015): equal bases keep the base with
 *    p = px py / 3 / (1 - px - py + 4 px py / 3); different bases keep the higher-quality one (the reverse read's
 *    on a tie) with p = px (1 - py/3) / (px + py - 4 px py / 3), px the smaller error; an N yields to the other
 *    read; q = round(-10 log10 p) clamped to [0, 41];
 *  - the merged read is kept when its expected errors (sum of 10^(-q/10) over the merged qualities) <= maxee.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static int    g_init = 0;
static double g_q2p[128], g_match[128][128], g_mism[128][128];
static unsigned char g_qsame[128][128], g_qdiff[128][128];

static double q_to_p(int c) { const int x = c - 33; return x < 2 ? 0.75 : pow(10.0, -(double)x / 10.0); }
static unsigned char qual_of(double p)
{
  double q = rint(-10.0 * log10(p));
  if (q > 41.0) q = 41.0;
  if (q < 0.0) q = 0.0;
  return (unsigned char)(33 + (int)q);
}
static void init_tables(void)
{
  if (g_init) return;
  for (int x = 33; x < 127; x++) {
    const double px = q_to_p(x);
    g_q2p[x] = px;
    for (int y = 33; y < 127; y++) {
      const double py = q_to_p(y);
      g_qsame[x][y] = qual_of(px * py / 3.0 / (1.0 - px - py + 4.0 * px * py / 3.0));
      g_qdiff[x][y] = qual_of(px * (1.0 - py / 3.0) / (px + py - 4.0 * px * py / 3.0));
      g_match[x][y] = log2((1.0 - px - py + px * py * 4.0 / 3.0) / 0.25);
      g_mism[x][y] = log2(((px + py) / 3.0 - px * py * 4.0 / 9.0) / 0.25);
    }
  }
  g_init = 1;
}
/* the engine's host code builds the same tables with the same libm calls; tests compare them byte for byte */
void orc_merge_tables(double *q2p, double *match, double *mism, unsigned char *qsame, unsigned char *qdiff)
{
  init_tables();
  memcpy(q2p, g_q2p, sizeof(g_q2p)); memcpy(match, g_match, sizeof(g_match)); memcpy(mism, g_mism, sizeof(g_mism));
  memcpy(qsame, g_qsame, sizeof(g_qsame)); memcpy(qdiff, g_qdiff, sizeof(g_qdiff));
}

static int code_of(char c)
{
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
               case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}
static char comp_of(char c)
{
  switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'U': return 'A';
               case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; case 'u': return 'a';
               default: return 'N'; }
}

/* reasons */
enum { MRG_OK = 0, MRG_NOKMERS = 1, MRG_REPEAT = 2, MRG_MINSCORE = 3, MRG_MAXDIFFS = 4, MRG_MINOVLEN = 5, MRG_STAGGERED = 6, MRG_MAXEE = 7,
       MRG_EMPTY = 8 };

/*
 * One pair.  f/fq: forward read and its qualities (ASCII), r/rq: reverse read as it is in the file.
 * out_seq/out_qual need flen + rlen bytes.  Returns the reason (0 = merged, *out_len set).
 */
int orc_merge_pair(const char *f, const char *fq, int fl, const char *r, const char *rq, int rl, int maxdiffs, double maxee,
                   int allow_stagger, char *out_seq, char *out_qual, int *out_len, double *ret_score, int *ret_shift)
{
  init_tables();
  *out_len = 0;
  if (ret_score) *ret_score = 0.0;
  if (ret_shift) *ret_shift = 0;
  if (fl < 1 || rl < 1) return MRG_EMPTY;
  char *rc = (char *)malloc((size_t)rl), *rcq = (char *)malloc((size_t)rl);
  int *f5 = (int *)malloc(sizeof(int) * (size_t)fl), *r5 = (int *)malloc(sizeof(int) * (size_t)rl);
  for (int j = 0; j < rl; j++) { rc[j] = comp_of(r[rl - 1 - j]); rcq[j] = rq[rl - 1 - j]; }
  for (int p = 0; p < fl; p++) {                    /* 5-mer starting at p, -1 when it holds an N or runs off the end */
    f5[p] = -1;
    if (p + 5 <= fl) { int v = 0, ok = 1; for (int t = 0; t < 5; t++) { const int c = code_of(f[p + t]); if (c > 3) ok = 0; v = v * 4 + c; } if (ok) f5[p] = v; }
  }
  for (int p = 0; p < rl; p++) {
    r5[p] = -1;
    if (p + 5 <= rl) { int v = 0, ok = 1; for (int t = 0; t < 5; t++) { const int c = code_of(rc[p + t]); if (c > 3) ok = 0; v = v * 4 + c; } if (ok) r5[p] = v; }
  }
  /* shift = position in the forward read that rc[0] is aligned with; larger shifts first (vsearch walks overlaps upward) */
  int kmers = 0, hits = 0, best_shift = 0, best_diffs = 0, have = 0;
  double best = 0.0;
  for (int shift = fl - 1; shift >= -(rl - 1); shift--) {
    const int a = shift > 0 ? shift : 0, b = (shift + rl < fl) ? shift + rl : fl;
    if (b <= a) continue;
    int cnt = 0;
    for (int p = a; p < b; p++) if (f5[p] >= 0 && f5[p] == r5[p - shift]) cnt++;
    if (cnt < 4) continue;
    kmers = 1;
    double score = 0.0, high = 0.0, drop = 0.0;
    int diffs = 0;
    for (int p = b - 1; p >= a; p--) {
      const unsigned char qa = (unsigned char)fq[p], qb = (unsigned char)rcq[p - shift];
      if (f[p] == rc[p - shift]) score += g_match[qa][qb];
      else { score += g_mism[qa][qb]; diffs++; }
      if (score > high) high = score;
      if (high - score > drop) drop = high - score;
    }
    if (drop >= 16.0) score = -1000.0;
    if (score >= 16.0) hits++;
    if (!have || score > best) { best = score; best_shift = shift; best_diffs = diffs; have = 1; }
  }
  int reason = MRG_OK;
  const int a = best_shift > 0 ? best_shift : 0, b = (best_shift + rl < fl) ? best_shift + rl : fl;
  if (!kmers) reason = MRG_NOKMERS;
  else if (hits > 1) reason = MRG_REPEAT;
  else if (best < 16.0) reason = MRG_MINSCORE;
  else if (best_diffs > maxdiffs) reason = MRG_MAXDIFFS;
  else if (b - a < 10) reason = MRG_MINOVLEN;
  else if (!allow_stagger && best_shift < 0) reason = MRG_STAGGERED;          /* the reverse read's 3' end overhangs the forward read's 5' end */
  if (ret_score) *ret_score = have ? best : 0.0;
  if (ret_shift) *ret_shift = best_shift;
  if (reason == MRG_OK) {
    int n = 0;
    double ee = 0.0;
    for (int p = 0; p < a; p++) { out_seq[n] = f[p]; out_qual[n] = fq[p]; ee += g_q2p[(unsigned char)fq[p]]; n++; }   /* forward only */
    for (int p = a; p < b; p++) {
      const char fs = f[p], rs = rc[p - best_shift];
      const unsigned char qa = (unsigned char)fq[p], qb = (unsigned char)rcq[p - best_shift];
      char s; unsigned char q;
      if (rs == 'N') { s = fs; q = qa; }
      else if (fs == 'N') { s = rs; q = qb; }
      else if (fs == rs) { s = fs; q = g_qsame[qa][qb]; }
      else if (qa > qb) { s = fs; q = g_qdiff[qa][qb]; }
      else { s = rs; q = g_qdiff[qb][qa]; }
      out_seq[n] = s; out_qual[n] = (char)q; ee += g_q2p[q]; n++;
    }
    /* then the rest of the reverse read; a reverse read that ends inside the forward read adds nothing and the forward
       read's 3' remainder is dropped, as is a staggered pair's overhang on the other side */
    for (int j = b - best_shift; j < rl && best_shift + rl >= fl; j++) { out_seq[n] = rc[j]; out_qual[n] = rcq[j]; ee += g_q2p[(unsigned char)rcq[j]]; n++; }
    if (ee > maxee) reason = MRG_MAXEE;
    else *out_len = n;
  }
  free(rc); free(rcq); free(f5); free(r5);
  return reason;
}
