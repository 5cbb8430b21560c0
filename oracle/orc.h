/*
 * orc.h -- CPU ORACLE for the ITSxpress hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped engine (itsxpress_amd/) never links, imports or
 * executes anything under oracle/.
 *
 * What it restates (reference file:line = the call sites whose results it must
 * reproduce; the arithmetic itself lives in un-vendored third-party binaries):
 *   - vsearch --fastx_uniques ... --strand both   itsxpress/SeqSample.py:106-116
 *     (vsearch >= 2.21.1, recipes/itsxpress/meta.yaml:37)
 *   - hmmsearch --domtblout -T 10 --F1 1e-6 --F2 1e-6 --F3 1e-6
 *                                                  itsxpress/SeqSample.py:191-209
 *     (hmmer >= 3.1b2, recipes/itsxpress/meta.yaml:36): MSV -> bias filter ->
 *     Forward -> Backward -> posterior domain definition -> per-envelope
 *     Forward + null2 -> thresholds, following HMMER's published algorithm and
 *     its SSE implementation's operation order (4-lane striping).
 *   - vsearch --cluster_size ... --id X --strand both itsxpress/SeqSample.py:147-162
 *     (orc_cluster.c; parity unpinned)
 *   - ItsPosition.parse/_score/get_position        itsxpress/SeqSample.py:400-498
 *   - Dedup.parse                                  itsxpress/SeqSample.py:542-562
 *
 * PARITY STATUS: derep is pinned by the reference fixture tests/test_data/
 * ex_tmpdir/{uc.txt,rep.fa}.  The HMM stages are "parity unpinned" against a
 * real hmmsearch binary (none is available, and the Fungi model file F.hmm the
 * reference's golden outputs were made with is absent from the mount); they are
 * anchored only by the 226 golden trim coordinates (tests/golden/) as a
 * plausibility check with the other taxa's profiles, and cross-checked by an independent float64 log-space
 * statement of the model (tests/hmm_generic.py: Forward scores, envelopes, domain and sequence bit scores, MSV).
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_K   4          /* canonical residues */
#define ORC_KP  18         /* A C G T - R Y M K S W H B V D N * ~ */

typedef struct {
  char   name[160];
  int    M;
  int    Q;                /* max(2, ceil(M/4)) : SSE float striping segment length */
  float  compo[4];
  float  evparam[6];       /* MSV mu, lambda ; VITERBI mu, lambda ; FORWARD tau, lambda */
  /* core probabilities, as read: p = expf(-x) */
  float *t;                /* [(M+1)][7]  MM MI MD IM II DM DD */
  float *mat;              /* [(M+1)][4] */
  /* generic profile, log-odds (nats) */
  float *tsc;              /* [(M+1)][8]  MM MI MD IM II DM DD BM(entry into k+1) */
  float *msc;              /* [(M+1)][ORC_KP] */
  /* MSV byte model */
  float   scale_b;
  uint8_t base_b, bias_b, tbm_b, tec_b;
  uint8_t *rbv;            /* [ORC_KP][M+1] cost bytes (k=1..M) */
  /* Forward/Backward odds-ratio model, striped: slot (q,z) holds node k=z*Q+q+1 */
  float *rfv;              /* [ORC_KP][Q][4] */
  float *tfv;              /* [8*Q][4]  per q: BM MM IM DM MD MI II ; then Q x DD */
  /* bias filter 2-state HMM */
  float  ft[2][3];         /* t[m][k], k=2 is the end transition */
  float  fpi[2];
  float  feo[ORC_KP][2];   /* emission odds */
  /* Viterbi filter word model (p7_oprofile.c: vf_conversion), unstriped: scale 500/ln2, base 12000 */
  int16_t *rww;            /* [ORC_KP][M+1] match emission words */
  int16_t *tww;            /* [8][M+1]: BM MM IM DM (into node k) | MD MI II DD (out of node k); -32768 = impossible */
} orc_profile;

typedef struct {
  int          n;
  orc_profile *p;
} orc_hmmset;

/* a reported-or-not domain, before thresholding */
typedef struct {
  int64_t seq;             /* index into the searched sequence set */
  int32_t prof;            /* index into the hmm set (file order) */
  int32_t tlen;
  int32_t ienv, jenv;      /* 1-based envelope */
  int32_t dom_idx, ndom;
  int32_t flags;           /* bit0: region was multidomain (stochastic clustering NOT done) */
  float   envsc;           /* nats */
  float   domcorrection;   /* nats */
  float   dombias;         /* nats */
  float   bitscore;        /* bits */
  double  lnP;
  float   seq_score;       /* bits, per-sequence */
  float   seq_bias;        /* bits */
  int32_t seq_reported;    /* per-sequence score >= T */
  int32_t dom_reported;    /* filled by orc_threshold */
} orc_domain;

typedef struct {
  /* per (seq,profile) filter trace, for stage-by-stage parity tests */
  int64_t seq; int32_t prof;
  int32_t msv_xj;          /* final xJ byte, 255 = overflow (+inf) */
  int32_t pass_msv, pass_bias, pass_fwd;
  float   msv_sc, filtersc, fwdsc, bcksc, nullsc;
  int32_t nregions, ndom;
  int32_t ran_vit, pass_vit;   /* Viterbi filter: runs only when the bias-corrected MSV P-value exceeds F2 (never at F1 == F2) */
  float   vitsc;
  int32_t pad;
} orc_pairtrace;

typedef struct {
  int64_t     n_dom, cap_dom;
  orc_domain *dom;
  int64_t     n_trace, cap_trace;
  orc_pairtrace *trace;    /* only pairs that passed MSV are traced (keep_trace=1), or all (keep_trace=2) */
  int64_t     n_pairs, n_past_msv, n_past_bias, n_past_fwd, n_multidomain;
} orc_results;

/* ---- HMM set ---- */
orc_hmmset *orc_hmmset_read(const char *path, char *err, int errlen);
orc_hmmset *orc_hmmset_parse(const char *text, int64_t len, char *err, int errlen);
void        orc_hmmset_free(orc_hmmset *hs);
int         orc_hmmset_count(const orc_hmmset *hs);
const char *orc_hmmset_name(const orc_hmmset *hs, int i);
int         orc_hmmset_M(const orc_hmmset *hs, int i);
/* copy-out helpers for table parity tests; return number of elements written */
int         orc_profile_rbv(const orc_hmmset *hs, int i, uint8_t *out /*[KP*(M+1)]*/);
int         orc_profile_rfv(const orc_hmmset *hs, int i, float *out /*[KP*Q*4]*/);
int         orc_profile_tfv(const orc_hmmset *hs, int i, float *out /*[8*Q*4]*/);
int         orc_profile_msvparams(const orc_hmmset *hs, int i, int *out /*base,bias,tbm,tec*/);

/* ---- sequences: digital codes 0..15 per residue (HMMER DNA digital alphabet) ---- */
int orc_digitize(const char *ascii, int64_t len, uint8_t *out); /* returns 0, or -1 on an illegal character */

/* ---- per-stage kernels (1-based dsq[1..L]; dsq[0] unused) ---- */
int   orc_msv(const orc_profile *p, const uint8_t *dsq, int L, int *ret_xJ, float *ret_sc);
float orc_nullsc(int L);
float orc_bias_filtersc(const orc_profile *p, const uint8_t *dsq, int L);
/* p7_ViterbiFilter restated without striping (max and saturating adds do not depend on it): returns 1 on overflow (+inf) */
int   orc_vitfilter(const orc_profile *p, const uint8_t *dsq, int L, float *ret_sc);

/* ---- the search (hmmsearch restated) ---- */
orc_results *orc_search(const orc_hmmset *hs, const uint8_t *codes, const int64_t *offsets, int64_t nseq,
                        double T, double F1, double F2, double F3, int keep_trace, int nthreads);
void         orc_threshold(orc_results *r, const orc_hmmset *hs, const int64_t *domZ_override, double domE);
void         orc_results_free(orc_results *r);
int64_t      orc_results_ndom(const orc_results *r);
const orc_domain *orc_results_dom(const orc_results *r);
int64_t      orc_results_ntrace(const orc_results *r);
const orc_pairtrace *orc_results_trace(const orc_results *r);
void         orc_results_counts(const orc_results *r, int64_t *out5);

/* ItsPosition semantics on thresholded results: -1 = None. left/right given by name prefixes. */
void orc_positions(const orc_results *r, const orc_hmmset *hs, int64_t nseq,
                   const char *leftprefix, const char *rightprefix,
                   int32_t *start, int32_t *stop, int32_t *tlen, int32_t *in_ddict);

/* ---- dereplication (vsearch --fastx_uniques --strand both restated) ---- */
/* codes: 4-bit-per-base digital codes as bytes; rep_of[i] = index of first occurrence (cluster seed),
 * -1 for reads shorter than minlen; strand[i] = +1/-1. returns number of clusters. */
int64_t orc_derep(const uint8_t *codes, const int64_t *offsets, int64_t n, int strand_both, int minlen,
                  int64_t *rep_of, int8_t *strand);

/* ---- xxHash64 (known-answer tests pin it to the xxhash library) ---- */
uint64_t orc_xxh64(const void *data, int64_t len, uint64_t seed);

/* ---- deterministic math, exported for tests ---- */
double orc_det_log(double x);
double orc_det_exp(double x);

/* ---- greedy centroid clustering (vsearch --cluster_size restated; orc_cluster.c; PARITY UNPINNED) ---- */
void    orc_align_identity(const uint8_t *qmask, int Lq, const uint8_t *tmask, int Lt,
                           int64_t *ret_score, int64_t *ret_matches, int64_t *ret_cols);
int64_t orc_cluster(const uint8_t *codes, const int64_t *offsets, int64_t n, const char *labels, const int64_t *label_offsets,
                    double id, int strand_both, int minlen, int64_t *rep_of, int8_t *strand, double *pct_id, int64_t *order,
                    int64_t *stats);

/* ---- paired-end merging (vsearch --fastq_mergepairs restated; orc_merge.c; PARITY UNPINNED) ---- */
int  orc_merge_pair(const char *f, const char *fq, int fl, const char *r, const char *rq, int rl, int maxdiffs, double maxee,
                    int allow_stagger, char *out_seq, char *out_qual, int *out_len, double *ret_score, int *ret_shift);
void orc_merge_tables(double *q2p, double *match, double *mism, unsigned char *qsame, unsigned char *qdiff);

/* DUST soft mask as vsearch applies it to seeds (--qmask dust / --dbmask dust): masked[pos] = 1 */
void orc_dust(const uint8_t *codes, int64_t L, uint8_t *masked);

/* ---- read orientation (vsearch --orient restated; orc_cluster.c; PARITY UNPINNED) ---- */
void orc_orient_db_add(uint8_t *dbbits /*2 MB, zeroed*/, const uint8_t *codes, int64_t L);
void orc_orient(const uint8_t *dbbits, const uint8_t *codes, const int64_t *offsets, int64_t n, int8_t *strand, int32_t *cfwd, int32_t *crev);

#ifdef __cplusplus
}
#endif
/* multidomain regions that hit a bookkeeping limit, by kind (index = the engine's MrOut.status code); reset != 0 clears them */
void orc_mr_fail_counts(long long *out, int reset);

#endif
