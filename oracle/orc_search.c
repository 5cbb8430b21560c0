/*
 * orc_search.c -- ORACLE (test infrastructure only): hmmsearch's per-target
 * pipeline restated on the CPU, for the flags the reference uses
 * (itsxpress/SeqSample.py:191-209: -T 10 --F1 1e-6 --F2 1e-6 --F3 1e-6).
 *
 * Stages (HMMER >= 3.1b2, un-vendored; restated from its published algorithm and
 * its SSE implementation's operation order):
 *   null1 -> MSV (uint8, saturating) -> bias filter (2-state HMM forward) ->
 *   [Viterbi filter: guarded by P > F2, never runs when F1 == F2] ->
 *   Forward parser (odds-ratio floats, 4-lane striped, sparse rescaling) ->
 *   Backward parser -> posterior decoding of B/E/occupancy -> region scan
 *   (rt1 .25, rt2 .10, rt3 .20) -> per-envelope unihit Forward/Backward,
 *   posterior decoding, null2 by expectation -> bit scores, thresholds.
 * Consumers that define "correct": ItsPosition.parse/_score/get_position
 * (itsxpress/SeqSample.py:400-498).
 *
 * Deviations, all documented in DESIGN.md:
 *   - libm log/exp inside the per-target pipeline are replaced by orc_log/orc_exp
 *     (orc_math.h) so that CPU and GPU agree bit-for-bit.
 *   - multidomain regions ARE resolved the way p7_domaindef.c resolves them (region_trace_ensemble below: 200
 *     stochastic tracebacks, null2 by trace, single-linkage clustering of the sampled domains); ORC_NO_ENSEMBLE=1
 *     restores the older behaviour (the region kept as one envelope, flags bit0) for A/B tests only.
 *   - optimal-accuracy alignment (ali/hmm coordinates, acc) is not computed; the
 *     reference consumes only envelope coordinates and the domain bit score.
 * PARITY UNPINNED against a real hmmsearch build (none available here).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "orc.h"
#include "orc_math.h"

#define LOG2 0.69314718055994529
#define QMAX 64

extern uint8_t orc_tjb_b(int L);
extern const uint8_t *orc_degen_table(void);

double orc_det_log(double x) { return orc_log(x); }
double orc_det_exp(double x) { return orc_exp(x); }

/* The 4-lane vectors of HMMER's SSE implementation.  The CHECKER (liborc.so) spells them out in scalar C, one float at a
 * time; the timed CPU BASELINE (libbase_sse.so: make -C oracle libbase_sse.so, -DORC_SSE) maps the same eight operations to
 * real SSE2 registers -- same operations, same order, IEEE single precision both ways, so every result is bit-identical
 * (tests/test_oracle_cpu.py::test_sse_baseline_is_bit_identical_to_the_checker) and only the speed differs. */
#ifdef ORC_SSE
#include <emmintrin.h>
typedef __m128 v4;
static inline v4 v4_set1(float a) { return _mm_set1_ps(a); }
static inline v4 v4_zero(void) { return _mm_setzero_ps(); }
static inline v4 v4_add(v4 a, v4 b) { return _mm_add_ps(a, b); }
static inline v4 v4_mul(v4 a, v4 b) { return _mm_mul_ps(a, b); }
static inline v4 v4_rshift(v4 a) { return _mm_castsi128_ps(_mm_slli_si128(_mm_castps_si128(a), 4)); }   /* lane z takes lane z-1, lane 0 takes +0 */
static inline v4 v4_lshift(v4 a) { return _mm_castsi128_ps(_mm_srli_si128(_mm_castps_si128(a), 4)); }
static inline float v4_hsum(v4 a)
{
  const v4 s = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(2, 3, 0, 1)));     /* (v0+v1, v1+v0, v2+v3, v3+v2) */
  return _mm_cvtss_f32(_mm_add_ss(s, _mm_movehl_ps(s, s)));                      /* (v0+v1) + (v2+v3) */
}
static inline v4 v4_ld(const float *p) { return _mm_loadu_ps(p); }
static inline float v4_get(v4 a, int r) { float t[4]; _mm_storeu_ps(t, a); return t[r]; }
#else
typedef struct { float v[4]; } v4;
static inline v4 v4_set1(float a) { v4 r = {{a, a, a, a}}; return r; }
static inline v4 v4_zero(void) { v4 r = {{0.f, 0.f, 0.f, 0.f}}; return r; }
static inline v4 v4_add(v4 a, v4 b) { v4 r; for (int z = 0; z < 4; z++) r.v[z] = a.v[z] + b.v[z]; return r; }
static inline v4 v4_mul(v4 a, v4 b) { v4 r; for (int z = 0; z < 4; z++) r.v[z] = a.v[z] * b.v[z]; return r; }
static inline v4 v4_rshift(v4 a) { v4 r = {{0.f, a.v[0], a.v[1], a.v[2]}}; return r; }  /* node k-1 wraps to next lane */
static inline v4 v4_lshift(v4 a) { v4 r = {{a.v[1], a.v[2], a.v[3], 0.f}}; return r; }
static inline float v4_hsum(v4 a) { return (a.v[0] + a.v[1]) + (a.v[2] + a.v[3]); }
static inline v4 v4_ld(const float *p) { v4 r; memcpy(r.v, p, 16); return r; }
static inline float v4_get(v4 a, int r) { return a.v[r]; }
#endif

typedef struct { float E, N, J, B, C, SCALE; } xrow;

/* flogsum table: built with libm at start-up exactly as HMMER's p7_FLogsumInit does */
static float flogsum_tbl[16000];
static int   flogsum_ready = 0;
static void flogsum_init(void)
{
  if (flogsum_ready) return;
  for (int i = 0; i < 16000; i++) flogsum_tbl[i] = (float)log(1. + exp((double)-i / 1000.f));
  flogsum_ready = 1;
}
static float flogsum(float a, float b)
{
  const float max = (a > b) ? a : b;
  const float min = (a > b) ? b : a;
  return (min == -INFINITY || (max - min) >= 15.7f) ? max : max + flogsum_tbl[(int)((max - min) * 1000.f)];
}
const float *orc_flogsum_table(void) { flogsum_init(); return flogsum_tbl; }

static double gumbel_surv(double x, double mu, double lambda)
{
  double y = lambda * (x - mu);
  double ey = -orc_exp(-y);
  if (fabs(ey) < 5e-9) return -ey;
  return 1 - orc_exp(ey);
}
static double exp_surv(double x, double mu, double lambda)
{
  if (x < mu) return 1.0;
  return orc_exp(-lambda * (x - mu));
}
static double exp_logsurv(double x, double mu, double lambda)
{
  if (x < mu) return 0.0;
  return -lambda * (x - mu);
}

float orc_nullsc(int L)
{
  float p1 = (float)L / (float)(L + 1);
  return (float)((double)(float)L * log((double)p1) + log(1. - (double)p1));
}
/* exported so tests can compare the engine's host-side per-length tables */
double orc_len_lognn3(int L) { return log((double)((float)L / (float)(L + 3))); }

/* ------------------------------------------------------------------ MSV */
#ifdef ORC_SSE
/* the timed baseline's MSV filter: 16 unsigned bytes per SSE2 register, striped like p7_MSVFilter (vector q, lane z = node
 * z Q + q + 1), saturating adds / subtracts and maxima -- operations whose result does not depend on the order, so xJ and
 * the overflow verdict equal the scalar loop's of the checker below (compared in tests).  The striped cost table is built
 * per call (3 vectors x 16 codes for a 45-node model: noise beside L rows). */
int orc_msv(const orc_profile *p, const uint8_t *dsq, int L, int *ret_xJ, float *ret_sc)
{
  const int M = p->M, Q = (M + 15) / 16 < 2 ? 2 : (M + 15) / 16;
  __m128i rbv[ORC_KP][4 * QMAX / 16 + 2], dp[4 * QMAX / 16 + 2];
  for (int x = 0; x < ORC_KP; x++)
    for (int q = 0; q < Q; q++) {
      uint8_t b[16];
      for (int z = 0; z < 16; z++) { const int k = z * Q + q + 1; b[z] = k <= M ? p->rbv[(size_t)x * (M + 1) + k] : 255; }
      rbv[x][q] = _mm_loadu_si128((const __m128i *)b);
    }
  const int tjb = orc_tjb_b(L);
  const int bias = p->bias_b, base = p->base_b, tec = p->tec_b;
  const int tjbm = tjb + p->tbm_b;
  const __m128i biasv = _mm_set1_epi8((char)bias);
  for (int q = 0; q < Q; q++) dp[q] = _mm_setzero_si128();
  int xJ = 0;
  int xB = base - tjbm; if (xB < 0) xB = 0;
  for (int i = 1; i <= L; i++) {
    const __m128i *rsc = rbv[dsq[i]];
    const __m128i xBv = _mm_set1_epi8((char)xB);
    __m128i xEv = _mm_setzero_si128();
    __m128i mpv = _mm_slli_si128(dp[Q - 1], 1);           /* node k-1 of the previous row: the last vector shifted by one lane */
    for (int q = 0; q < Q; q++) {
      __m128i sv = _mm_max_epu8(mpv, xBv);
      sv = _mm_adds_epu8(sv, biasv);
      sv = _mm_subs_epu8(sv, rsc[q]);
      xEv = _mm_max_epu8(xEv, sv);
      mpv = dp[q];
      dp[q] = sv;
    }
    xEv = _mm_max_epu8(xEv, _mm_srli_si128(xEv, 8));
    xEv = _mm_max_epu8(xEv, _mm_srli_si128(xEv, 4));
    xEv = _mm_max_epu8(xEv, _mm_srli_si128(xEv, 2));
    xEv = _mm_max_epu8(xEv, _mm_srli_si128(xEv, 1));
    int xE = _mm_cvtsi128_si32(xEv) & 0xff;
    if (xE + bias >= 255) { *ret_xJ = 255; *ret_sc = INFINITY; return 1; }
    xE -= tec; if (xE < 0) xE = 0;
    if (xE > xJ) xJ = xE;
    xB = (base > xJ ? base : xJ) - tjbm; if (xB < 0) xB = 0;
  }
  *ret_xJ = xJ;
  float sc = ((float)(xJ - tjb) - (float)base);
  sc /= p->scale_b;
  sc -= 3.0f;
  *ret_sc = sc;
  return 0;
}
#else
int orc_msv(const orc_profile *p, const uint8_t *dsq, int L, int *ret_xJ, float *ret_sc)
{
  const int M = p->M;
  uint8_t dp[4 * QMAX + 2], nd[4 * QMAX + 2];
  memset(dp, 0, sizeof(dp));
  const int tjb = orc_tjb_b(L);
  const int bias = p->bias_b, base = p->base_b, tec = p->tec_b;
  const int tjbm = tjb + p->tbm_b;
  int xJ = 0;
  int xB = base - tjbm; if (xB < 0) xB = 0;
  for (int i = 1; i <= L; i++) {
    const uint8_t *rsc = p->rbv + (size_t)dsq[i] * (M + 1);
    int xE = 0;
    nd[0] = 0;
    for (int k = 1; k <= M; k++) {
      int sv = dp[k - 1] > xB ? dp[k - 1] : xB;
      sv += bias; if (sv > 255) sv = 255;
      sv -= rsc[k]; if (sv < 0) sv = 0;
      if (sv > xE) xE = sv;
      nd[k] = (uint8_t)sv;
    }
    memcpy(dp, nd, M + 1);
    if (xE + bias >= 255) { *ret_xJ = 255; *ret_sc = INFINITY; return 1; }
    xE -= tec; if (xE < 0) xE = 0;
    if (xE > xJ) xJ = xE;
    xB = (base > xJ ? base : xJ) - tjbm; if (xB < 0) xB = 0;
  }
  *ret_xJ = xJ;
  float sc = ((float)(xJ - tjb) - (float)base);
  sc /= p->scale_b;
  sc -= 3.0f;
  *ret_sc = sc;
  return 0;
}

#endif

/* ------------------------------------------------------------------ bias filter */
float orc_bias_filtersc(const orc_profile *p, const uint8_t *dsq, int L)
{
  float d0, d1, max, logsc;
  float scsum;
  /* p7_bg_SetLength() resets the filter HMM's state-0 transitions for every target length */
  const float p1 = (float)L / (float)(L + 1);
  const float t00 = p1, t01 = 1.0f - p1;
  const float t10 = p->ft[1][0], t11 = p->ft[1][1];
  /* row 1 */
  d0 = p->feo[dsq[1]][0] * p->fpi[0];
  d1 = p->feo[dsq[1]][1] * p->fpi[1];
  max = 0.0f; if (d0 > max) max = d0; if (d1 > max) max = d1;
  d0 /= max; d1 /= max;
  /* HMMER keeps per-row log scale factors in a float array and sums them at the end (rows 1..L+1) */
  float *sc = (float *)malloc(sizeof(float) * (L + 2));
  sc[1] = (float)orc_log((double)max);
  for (int i = 2; i <= L; i++) {
    float n0, n1;
    n0 = 0.0f; n0 += d0 * t00; n0 += d1 * t10; n0 *= p->feo[dsq[i]][0];
    n1 = 0.0f; n1 += d0 * t01; n1 += d1 * t11; n1 *= p->feo[dsq[i]][1];
    max = 0.0f; if (n0 > max) max = n0; if (n1 > max) max = n1;
    d0 = n0 / max; d1 = n1 / max;
    sc[i] = (float)orc_log((double)max);
  }
  scsum = 0.0f;
  scsum += d0 * p->ft[0][2];
  scsum += d1 * p->ft[1][2];
  sc[L + 1] = (float)orc_log((double)scsum);
  logsc = 0.0f;
  for (int i = 1; i <= L + 1; i++) logsc += sc[i];
  free(sc);
  /* nullsc + (float) L * logf(p1) + logf(1.-p1): per-length terms use libm (host-side tables in the engine) */
  return logsc + (float)L * logf(p1) + logf((float)(1. - (double)p1));
}

/* ------------------------------------------------------------------ Viterbi filter
 * p7_ViterbiFilter (impl_sse/vitfilter.c): 16-bit saturating Viterbi, multihit local, NN/CC/JJ costs 0 with a flat -3 nat
 * correction at the end.  HMMER evaluates it striped and skips the D->D paths of a row when they provably cannot matter
 * ("lazy F"); maxima and saturating additions do not depend on the evaluation order and a skipped D->D path never raises a
 * match cell of the next row, so the plain recurrence below gives the same xC.  Under the reference's flags (--F1 == --F2)
 * the pipeline never calls it. */
static inline int sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
int orc_vitfilter(const orc_profile *p, const uint8_t *dsq, int L, float *ret_sc)
{
  const int M = p->M;
  const int16_t *tbm = p->tww, *tmm = tbm + (M + 1), *tim = tmm + (M + 1), *tdm = tim + (M + 1);
  const int16_t *tmd = tdm + (M + 1), *tmi = tmd + (M + 1), *tii = tmi + (M + 1), *tdd = tii + (M + 1);
  const float scale_w = (float)(500.0 / LOG2);
  const int base = 12000;
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
  int16_t xmove, eloop;
  { float w = roundf(scale_w * logf(pmove)); xmove = (w >= 32767.0f) ? 32767 : (w <= -32768.0f) ? -32768 : (int16_t)w; }
  { float w = roundf(scale_w * logf(0.5f)); eloop = (int16_t)w; }
  int mm[4 * QMAX + 2], im[4 * QMAX + 2], dm[4 * QMAX + 2];
  for (int k = 0; k <= M; k++) mm[k] = im[k] = dm[k] = -32768;
  int16_t xN = (int16_t)base, xB = (int16_t)(xN + xmove), xJ = -32768, xC = -32768, xE;
  for (int i = 1; i <= L; i++) {
    const int16_t *rsc = p->rww + (size_t)dsq[i] * (M + 1);
    int xe = -32768;
    for (int k = M; k >= 1; k--) {                 /* descending: cell k-1 still holds the previous row */
      const int ni = sat16(mm[k] + tmi[k]) > sat16(im[k] + tii[k]) ? sat16(mm[k] + tmi[k]) : sat16(im[k] + tii[k]);
      int sv = sat16(xB + tbm[k]);
      int v = sat16(mm[k - 1] + tmm[k]); if (v > sv) sv = v;
      v = sat16(im[k - 1] + tim[k]); if (v > sv) sv = v;
      v = sat16(dm[k - 1] + tdm[k]); if (v > sv) sv = v;
      sv = sat16(sv + rsc[k]);
      if (sv > xe) xe = sv;
      mm[k] = sv; im[k] = ni;
    }
    dm[1] = -32768;
    for (int k = 2; k <= M; k++) {
      const int a = sat16(mm[k - 1] + tmd[k - 1]), b = sat16(dm[k - 1] + tdd[k - 1]);
      dm[k] = a > b ? a : b;
    }
    xE = (int16_t)xe;
    if (xE >= 32767) { *ret_sc = INFINITY; return 1; }
    xN = (int16_t)(xN + 0);
    { const int a = xC + 0, b = xE + eloop; xC = (int16_t)(a > b ? a : b); }           /* E->C and E->J both cost log(1/2) */
    { const int a = xJ + 0, b = xE + eloop; xJ = (int16_t)(a > b ? a : b); }
    { const int a = xJ + xmove, b = xN + xmove; xB = (int16_t)(a > b ? a : b); }
  }
  if (xC > -32768) {
    float sc = (float)xC + (float)xmove - (float)base;
    sc /= scale_w;
    sc -= 3.0f;
    *ret_sc = sc;
  } else *ret_sc = -INFINITY;
  return 0;
}

/* ------------------------------------------------------------------ Forward */
/* rows: if non-NULL, receives M and I of every row i=1..L as [i][q][2] v4 (row 0 untouched) */
/* full: if non-NULL, receives M, D and I of every row i=0..L as [i][q][3] v4 (p7_Forward's whole matrix, for the
 * stochastic tracebacks of a multidomain region) */
static int fwd_engine_x(const orc_profile *p, const uint8_t *dsq, int L,
                        float pmove, float ploop, float eloop, float emove,
                        xrow *xmx, v4 *rows, v4 *full, float *ret_sc);
static int fwd_engine(const orc_profile *p, const uint8_t *dsq, int L,
                      float pmove, float ploop, float eloop, float emove,
                      xrow *xmx, v4 *rows, float *ret_sc)
{ return fwd_engine_x(p, dsq, L, pmove, ploop, eloop, emove, xmx, rows, NULL, ret_sc); }
static int fwd_engine_x(const orc_profile *p, const uint8_t *dsq, int L,
                        float pmove, float ploop, float eloop, float emove,
                        xrow *xmx, v4 *rows, v4 *full, float *ret_sc)
{
  const int Q = p->Q;
  v4 mmx[QMAX], dmx[QMAX], imx[QMAX];
  dmx[0] = v4_zero();
  float xN, xE, xB, xC, xJ, totscale = 0.0f;
  for (int q = 0; q < Q; q++) mmx[q] = dmx[q] = imx[q] = v4_zero();
  xE = 0.f; xN = 1.f; xJ = 0.f; xB = pmove; xC = 0.f;
  xmx[0].E = xE; xmx[0].N = xN; xmx[0].J = xJ; xmx[0].B = xB; xmx[0].C = xC; xmx[0].SCALE = 1.0f;
  if (full) for (int q = 0; q < Q * 3; q++) full[q] = v4_zero();
  for (int i = 1; i <= L; i++) {
    const float *rp = p->rfv + (size_t)dsq[i] * Q * 4;
    const float *tp = p->tfv;
    v4 dcv = v4_zero(), xEv = v4_zero(), xBv = v4_set1(xB);
    v4 mpv = v4_rshift(mmx[Q - 1]), dpv = v4_rshift(dmx[Q - 1]), ipv = v4_rshift(imx[Q - 1]);
    v4 sv;
    for (int q = 0; q < Q; q++) {
      sv = v4_mul(xBv, v4_ld(tp)); tp += 4;
      sv = v4_add(sv, v4_mul(mpv, v4_ld(tp))); tp += 4;
      sv = v4_add(sv, v4_mul(ipv, v4_ld(tp))); tp += 4;
      sv = v4_add(sv, v4_mul(dpv, v4_ld(tp))); tp += 4;
      sv = v4_mul(sv, v4_ld(rp)); rp += 4;
      xEv = v4_add(xEv, sv);
      mpv = mmx[q]; dpv = dmx[q]; ipv = imx[q];
      mmx[q] = sv; dmx[q] = dcv;
      dcv = v4_mul(sv, v4_ld(tp)); tp += 4;
      sv = v4_mul(mpv, v4_ld(tp)); tp += 4;
      imx[q] = v4_add(sv, v4_mul(ipv, v4_ld(tp))); tp += 4;
    }
    /* DD paths: one pass that adds M->D and D->D, then three that extend D->D across lanes */
    dcv = v4_rshift(dcv);
    dmx[0] = v4_zero();
    tp = p->tfv + 7 * Q * 4;
    for (int q = 0; q < Q; q++) {
      dmx[q] = v4_add(dcv, dmx[q]);
      dcv = v4_mul(dmx[q], v4_ld(tp)); tp += 4;
    }
    for (int j = 1; j < 4; j++) {
      dcv = v4_rshift(dcv);
      tp = p->tfv + 7 * Q * 4;
      for (int q = 0; q < Q; q++) {
        dmx[q] = v4_add(dcv, dmx[q]);
        dcv = v4_mul(dcv, v4_ld(tp)); tp += 4;
      }
    }
    for (int q = 0; q < Q; q++) xEv = v4_add(dmx[q], xEv);
    xE = v4_hsum(xEv);
    xN = xN * ploop;
    xC = (xC * ploop) + (xE * emove);
    xJ = (xJ * ploop) + (xE * eloop);
    xB = (xJ * pmove) + (xN * pmove);
    if (xE > 1.0e4f) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      v4 sc = v4_set1((float)(1.0 / (double)xE));
      for (int q = 0; q < Q; q++) { mmx[q] = v4_mul(mmx[q], sc); dmx[q] = v4_mul(dmx[q], sc); imx[q] = v4_mul(imx[q], sc); }
      xmx[i].SCALE = xE;
      totscale = (float)((double)totscale + orc_log((double)xE));
      xE = 1.0f;
    } else xmx[i].SCALE = 1.0f;
    xmx[i].E = xE; xmx[i].N = xN; xmx[i].J = xJ; xmx[i].B = xB; xmx[i].C = xC;
    if (rows) for (int q = 0; q < Q; q++) { rows[((size_t)i * Q + q) * 2] = mmx[q]; rows[((size_t)i * Q + q) * 2 + 1] = imx[q]; }
    if (full) for (int q = 0; q < Q; q++) { full[((size_t)i * Q + q) * 3] = mmx[q]; full[((size_t)i * Q + q) * 3 + 1] = dmx[q]; full[((size_t)i * Q + q) * 3 + 2] = imx[q]; }
  }
  if (isnan(xC)) return -1;
  if (L > 0 && xC == 0.0f) return -2;      /* underflow */
  if (isinf(xC)) return -3;
  *ret_sc = (float)((double)totscale + orc_log((double)(xC * pmove)));
  return 0;
}

/* ------------------------------------------------------------------ Backward */
static int bwd_engine(const orc_profile *p, const uint8_t *dsq, int L,
                      float pmove, float ploop, float eloop, float emove,
                      const xrow *fwd, xrow *bck, int *ret_own_scales, v4 *rows, float *ret_sc)
{
  const int Q = p->Q;
  const float *tfv = p->tfv;
  v4 mmx[QMAX], dmx[QMAX], imx[QMAX];
  dmx[0] = v4_zero();
  v4 mpv, ipv, dpv, mcv, dcv, tmmv, timv, tdmv, xBv, xEv;
  float xN, xE, xB, xC, xJ, totscale;
  int own = 0;
#define TFV(idx) v4_ld(tfv + (size_t)(idx) * 4)
  xJ = 0.f; xB = 0.f; xN = 0.f;
  xC = pmove;
  xE = xC * emove;
  xEv = v4_set1(xE);
  dcv = v4_zero();
  for (int q = 0; q < Q; q++) { mmx[q] = dmx[q] = xEv; imx[q] = v4_zero(); }
  /* row L: D->D paths (first segment carries the E contribution), then three extension passes */
  {
    int tp = 8 * Q - 1;
    dpv = v4_lshift(dmx[Q - 1]);             /* HMMER shifts D(Q-1) here; all D are equal on this row */
    for (int q = Q - 1; q >= 0; q--) {
      dcv = v4_mul(dpv, TFV(tp)); tp--;
      dmx[q] = v4_add(dmx[q], dcv);
      dpv = dmx[q];
    }
    for (int j = 1; j < 4; j++) {
      tp = 8 * Q - 1;
      dcv = v4_lshift(dcv);
      for (int q = Q - 1; q >= 0; q--) {
        dcv = v4_mul(dcv, TFV(tp)); tp--;
        dmx[q] = v4_add(dmx[q], dcv);
      }
    }
    tp = 7 * Q - 3;
    dcv = v4_lshift(dmx[0]);
    for (int q = Q - 1; q >= 0; q--) {
      mmx[q] = v4_add(mmx[q], v4_mul(dcv, TFV(tp))); tp -= 7;
      dcv = dmx[q];
    }
  }
  if (fwd[L].SCALE > 1.0f) {
    float s = fwd[L].SCALE;
    xE = xE / s; xN = xN / s; xC = xC / s; xJ = xJ / s; xB = xB / s;
    v4 sc = v4_set1((float)(1.0 / (double)s));
    for (int q = 0; q < Q; q++) { mmx[q] = v4_mul(mmx[q], sc); dmx[q] = v4_mul(dmx[q], sc); imx[q] = v4_mul(imx[q], sc); }
  }
  bck[L].SCALE = fwd[L].SCALE;
  totscale = (float)orc_log((double)bck[L].SCALE);
  bck[L].E = xE; bck[L].N = xN; bck[L].J = xJ; bck[L].B = xB; bck[L].C = xC;
  if (rows) for (int q = 0; q < Q; q++) { rows[((size_t)L * Q + q) * 2] = mmx[q]; rows[((size_t)L * Q + q) * 2 + 1] = imx[q]; }

  for (int i = L - 1; i >= 1; i--) {
    const float *rfx = p->rfv + (size_t)dsq[i + 1] * Q * 4;
    int rp = Q - 1;
    int tp = 7 * Q - 1;
    tmmv = v4_lshift(TFV(1)); timv = v4_lshift(TFV(2)); tdmv = v4_lshift(TFV(3));
    mpv = v4_mul(mmx[0], v4_ld(rfx));
    mpv = v4_lshift(mpv);
    xBv = v4_zero();
    for (int q = Q - 1; q >= 0; q--) {
      ipv = imx[q];
      imx[q] = v4_add(v4_mul(ipv, TFV(tp)), v4_mul(mpv, timv)); tp--;
      dmx[q] = v4_mul(mpv, tdmv);
      mcv = v4_add(v4_mul(ipv, TFV(tp)), v4_mul(mpv, tmmv)); tp -= 2;
      mpv = v4_mul(mmx[q], v4_ld(rfx + (size_t)rp * 4)); rp--;
      mmx[q] = mcv;
      tdmv = TFV(tp); tp--;
      timv = TFV(tp); tp--;
      tmmv = TFV(tp); tp--;
      xBv = v4_add(xBv, v4_mul(mpv, TFV(tp))); tp--;
    }
    xB = v4_hsum(xBv);
    xC = xC * ploop;
    xJ = (xB * pmove) + (xJ * ploop);
    xN = (xB * pmove) + (xN * ploop);
    xE = (xC * emove) + (xJ * eloop);
    xEv = v4_set1(xE);
    tp = 8 * Q - 1;
    dpv = v4_add(dmx[0], xEv);
    dpv = v4_lshift(dpv);
    for (int q = Q - 1; q >= 0; q--) {
      dcv = v4_mul(dpv, TFV(tp)); tp--;
      dmx[q] = v4_add(dmx[q], v4_add(dcv, xEv));
      dpv = dmx[q];
      mmx[q] = v4_add(mmx[q], xEv);
    }
    for (int j = 1; j < 4; j++) {
      dcv = v4_lshift(dcv);
      tp = 8 * Q - 1;
      for (int q = Q - 1; q >= 0; q--) {
        dcv = v4_mul(dcv, TFV(tp)); tp--;
        dmx[q] = v4_add(dmx[q], dcv);
      }
    }
    dcv = v4_lshift(dmx[0]);
    tp = 7 * Q - 3;
    for (int q = Q - 1; q >= 0; q--) {
      mmx[q] = v4_add(mmx[q], v4_mul(dcv, TFV(tp))); tp -= 7;
      dcv = dmx[q];
    }
    if (xB > 1.0e16f) own = 1;
    if (own) bck[i].SCALE = (xB > 1.0e4f) ? xB : 1.0f;
    else     bck[i].SCALE = fwd[i].SCALE;
    if (bck[i].SCALE > 1.0f) {
      float s = bck[i].SCALE;
      xE /= s; xN /= s; xJ /= s; xB /= s; xC /= s;
      v4 sc = v4_set1((float)(1.0 / (double)s));
      for (int q = 0; q < Q; q++) { mmx[q] = v4_mul(mmx[q], sc); dmx[q] = v4_mul(dmx[q], sc); imx[q] = v4_mul(imx[q], sc); }
      totscale = (float)((double)totscale + orc_log((double)s));
    }
    bck[i].E = xE; bck[i].N = xN; bck[i].J = xJ; bck[i].B = xB; bck[i].C = xC;
    if (rows) for (int q = 0; q < Q; q++) { rows[((size_t)i * Q + q) * 2] = mmx[q]; rows[((size_t)i * Q + q) * 2 + 1] = imx[q]; }
  }
  /* row 0: no residue; only B and N are reachable */
  {
    const float *rfx = p->rfv + (size_t)dsq[1] * Q * 4;
    int tp = 7 * Q - 7;                    /* B->M of the last q */
    xBv = v4_zero();
    for (int q = Q - 1; q >= 0; q--) {
      mpv = v4_mul(mmx[q], v4_ld(rfx + (size_t)q * 4));
      xBv = v4_add(xBv, v4_mul(mpv, TFV(tp))); tp -= 7;
    }
    xB = v4_hsum(xBv);
    xN = (xB * pmove) + (xN * ploop);
    bck[0].B = xB; bck[0].C = 0.f; bck[0].J = 0.f; bck[0].N = xN; bck[0].E = 0.f; bck[0].SCALE = 1.0f;
  }
#undef TFV
  *ret_own_scales = own;
  if (isnan(xN)) return -1;
  if (L > 0 && xN == 0.0f) return -2;
  if (isinf(xN)) return -3;
  if (ret_sc) *ret_sc = (float)((double)totscale + orc_log((double)xN));
  return 0;
}

/* ------------------------------------------------------------------ results buffer */
static void res_push_dom(orc_results *r, const orc_domain *d)
{
  if (r->n_dom == r->cap_dom) { r->cap_dom = r->cap_dom ? r->cap_dom * 2 : 1024; r->dom = (orc_domain *)realloc(r->dom, sizeof(orc_domain) * r->cap_dom); }
  r->dom[r->n_dom++] = *d;
}
static void res_push_trace(orc_results *r, const orc_pairtrace *t)
{
  if (r->n_trace == r->cap_trace) { r->cap_trace = r->cap_trace ? r->cap_trace * 2 : 1024; r->trace = (orc_pairtrace *)realloc(r->trace, sizeof(orc_pairtrace) * r->cap_trace); }
  r->trace[r->n_trace++] = *t;
}

typedef struct {
  xrow *xf, *xb, *ef, *eb;      /* parser specials; envelope specials */
  float *btot, *etot, *mocc, *n2sc;
  v4 *frows, *brows;            /* envelope M/I rows */
  v4 *full;                     /* region Forward matrix, [row][q][M D I] */
  int cap;
} workspace;

static void ws_grow(workspace *w, int L, int Q)
{
  if (L + 2 <= w->cap) return;
  int cap = L + 64;
  w->xf = (xrow *)realloc(w->xf, sizeof(xrow) * cap); w->xb = (xrow *)realloc(w->xb, sizeof(xrow) * cap);
  w->ef = (xrow *)realloc(w->ef, sizeof(xrow) * cap); w->eb = (xrow *)realloc(w->eb, sizeof(xrow) * cap);
  w->btot = (float *)realloc(w->btot, sizeof(float) * cap); w->etot = (float *)realloc(w->etot, sizeof(float) * cap);
  w->mocc = (float *)realloc(w->mocc, sizeof(float) * cap); w->n2sc = (float *)realloc(w->n2sc, sizeof(float) * cap);
  w->frows = (v4 *)realloc(w->frows, sizeof(v4) * (size_t)cap * QMAX * 2);
  w->brows = (v4 *)realloc(w->brows, sizeof(v4) * (size_t)cap * QMAX * 2);
  w->full = (v4 *)realloc(w->full, sizeof(v4) * (size_t)cap * QMAX * 3);
  w->cap = cap; (void)Q;
}
static void ws_free(workspace *w)
{
  free(w->xf); free(w->xb); free(w->ef); free(w->eb); free(w->btot); free(w->etot); free(w->mocc); free(w->n2sc); free(w->frows); free(w->brows); free(w->full);
}

typedef struct { int ienv, jenv, flags; float envsc, domcorrection; } domrec;

/* Re-score one envelope i..j in unihit mode; fills n2sc[i..j]; returns 0 and a domain record, or <0 if skipped */
static int rescore_domain(const orc_profile *p, const uint8_t *dsq, int L, int i, int j, int null2_is_done, workspace *w, domrec *out)
{
  const int Q = p->Q;
  const int Ld = j - i + 1;
  const uint8_t *sub = dsq + i - 1;
  const float pmove = 2.0f / ((float)L + 2.0f);          /* unihit, nj = 0; length model stays at the full L */
  const float ploop = 1.0f - pmove;
  float envsc;
  int own = 0;
  if (fwd_engine(p, sub, Ld, pmove, ploop, 0.0f, 1.0f, w->ef, w->frows, &envsc) != 0) return -1;
  if (bwd_engine(p, sub, Ld, pmove, ploop, 0.0f, 1.0f, w->ef, w->eb, &own, w->brows, NULL) != 0) return -1;
  /* posterior decoding + expected state usage (rows summed in increasing order) */
  float scaleproduct = (float)(1.0 / (double)w->eb[0].N);
  v4 accM[QMAX], accI[QMAX];
  float accN = 0.f, accC = 0.f, accJ = 0.f;
  for (int r = 1; r <= Ld; r++) {
    v4 totrv = v4_set1(scaleproduct * w->ef[r].SCALE);
    for (int q = 0; q < Q; q++) {
      v4 pm = v4_mul(v4_mul(w->frows[((size_t)r * Q + q) * 2], w->brows[((size_t)r * Q + q) * 2]), totrv);
      v4 pi = v4_mul(v4_mul(w->frows[((size_t)r * Q + q) * 2 + 1], w->brows[((size_t)r * Q + q) * 2 + 1]), totrv);
      if (r == 1) { accM[q] = pm; accI[q] = pi; }
      else { accM[q] = v4_add(pm, accM[q]); accI[q] = v4_add(pi, accI[q]); }
    }
    float pN = w->ef[r - 1].N * w->eb[r].N * ploop * scaleproduct;
    float pJ = w->ef[r - 1].J * w->eb[r].J * ploop * scaleproduct;
    float pC = w->ef[r - 1].C * w->eb[r].C * ploop * scaleproduct;
    if (r == 1) { accN = pN; accC = pC; accJ = pJ; } else { accN += pN; accC += pC; accJ += pJ; }
    if (own) scaleproduct *= w->ef[r].SCALE / w->eb[r].SCALE;
  }
  if (isinf(scaleproduct)) return -2;
  if (null2_is_done) {          /* an envelope out of a clustered region: n2sc[i..j] was set from the traceback ensemble */
    float dc = 0.0f;
    for (int pos = i; pos <= j; pos++) dc += w->n2sc[pos];
    out->ienv = i; out->jenv = j; out->envsc = envsc; out->domcorrection = dc; out->flags = 1;
    return 0;
  }
  float norm = (float)(1.0 / (double)(float)Ld);
  for (int q = 0; q < Q; q++) { accM[q] = v4_mul(accM[q], v4_set1(norm)); accI[q] = v4_mul(accI[q], v4_set1(norm)); }
  accN *= norm; accC *= norm; accJ *= norm;
  float xfactor = accN + accC + accJ;
  float null2[ORC_KP];
  for (int x = 0; x < 4; x++) {
    v4 sv = v4_zero();
    const float *rp = p->rfv + (size_t)x * Q * 4;
    for (int q = 0; q < Q; q++) {
      sv = v4_add(sv, v4_mul(accM[q], v4_ld(rp))); rp += 4;
      sv = v4_add(sv, accI[q]);
    }
    null2[x] = v4_hsum(sv);
    null2[x] += xfactor;
  }
  const uint8_t *degen = orc_degen_table();
  for (int x = 5; x <= 15; x++) {
    float result = 0.f; int nd = 0;
    for (int y = 0; y < 4; y++) if (degen[x] & (1 << y)) { result += null2[y]; nd++; }
    null2[x] = result / (float)nd;
  }
  null2[4] = null2[16] = null2[17] = 1.0f;
  float n2log[ORC_KP];
  for (int x = 0; x < ORC_KP; x++) n2log[x] = orc_logf(null2[x]);
  float domcorrection = 0.0f;
  for (int pos = i; pos <= j; pos++) { w->n2sc[pos] = n2log[dsq[pos]]; }
  for (int pos = i; pos <= j; pos++) domcorrection += w->n2sc[pos];
  out->ienv = i; out->jenv = j; out->envsc = envsc; out->domcorrection = domcorrection; out->flags = 0;
  return 0;
}

/* ------------------------------------------------------------------ multidomain regions
 * hmmsearch resolves a region whose posterior suggests more than one domain (is_multidomain_region, rt3 = 0.20) by
 * p7_domaindef.c:region_trace_ensemble(): 200 stochastic tracebacks over the region's (multihit) Forward matrix, a null2
 * score per residue from the traces, single-linkage clustering of the sampled domain coordinates (p7_spensemble.c) and
 * removal of dominated clusters.  Restated from the published procedure (impl_sse/stotrace.c, impl_sse/null2.c:
 * p7_Null2_ByTrace, p7_spensemble.c, esl_random.c); PARITY UNPINNED like the rest of the HMM half.  Choices that follow
 * HMMER 3.1b2 and are version-sensitive: the pipeline's generator is esl_randomness_CreateFast(42) -- Knuth's linear
 * congruential x <- 69069 x + 1 on 32 bits, state = esl_rnd_mix3(seed, 87654321, 12345678), re-initialised for every
 * region (do_reseeding) -- NOT the Mersenne Twister of esl_randomness_Create(); esl_vec_FNorm sums plainly (no
 * compensated summation); esl_rnd_FChoose accumulates in double. */
typedef struct { uint32_t x; } orc_rng;
static uint32_t rnd_mix3(uint32_t a, uint32_t b, uint32_t c)
{
  a -= b; a -= c; a ^= (c >> 13);
  b -= c; b -= a; b ^= (a << 8);
  c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);
  b -= c; b -= a; b ^= (a << 16);
  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);
  b -= c; b -= a; b ^= (a << 10);
  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
static void rng_init(orc_rng *r, uint32_t seed) { r->x = rnd_mix3(seed, 87654321u, 12345678u); if (r->x == 0) r->x = 42; }
static double rng_next(orc_rng *r) { r->x *= 69069u; r->x += 1u; return (double)r->x / 4294967296.0; }
static void fnorm(float *v, int n)
{
  float sum = 0.f;
  for (int x = 0; x < n; x++) sum += v[x];
  if (sum != 0.0f) for (int x = 0; x < n; x++) v[x] /= sum;
  else for (int x = 0; x < n; x++) v[x] = (float)(1. / (double)(float)n);
}
static int fchoose(orc_rng *r, const float *pv, int n)
{
  const double roll = rng_next(r);
  double sum = 0.0;
  for (int i = 0; i < n; i++) { sum += pv[i]; if (roll < sum) return i; }
  int i;
  do { i = (int)(rng_next(r) * n); } while (pv[i] == 0.f);
  return i;
}
enum { ST_M = 1, ST_D, ST_I, ST_S, ST_N, ST_B, ST_E, ST_C, ST_T, ST_J };
typedef struct { int idx, i, j, k, m; float prob; } spcoord;

#define FULLV(row, q, s) full[((size_t)(row) * Q + (q)) * 3 + (s)]      /* s: 0 M, 1 D, 2 I */
#define TFVQ(idx) v4_ld(p->tfv + (size_t)(idx) * 4)
/* sample one path through the region's Forward matrix; reports its domains last-first through cb arrays */
static int stochastic_trace(orc_rng *rng, const orc_profile *p, const v4 *full, const xrow *xf, int Lr, float pmove, float ploop,
                            int *dfrom, int *dto, int *dk, int *dm, float *cntM /*[ndom][Q*4]*/, float *cntI, int maxdom)
{
  const int Q = p->Q;
  int i = Lr, k = 0, s0 = ST_C, s1, nd = 0, cur = -1;
  while (s0 != ST_S) {
    float path[4];
    switch (s0) {
    case ST_M: {
      const int q = (k - 1) % Q, r = (k - 1) / Q;
      v4 mpv, dpv, ipv;
      if (q > 0) { mpv = FULLV(i - 1, q - 1, 0); dpv = FULLV(i - 1, q - 1, 1); ipv = FULLV(i - 1, q - 1, 2); }
      else { mpv = v4_rshift(FULLV(i - 1, Q - 1, 0)); dpv = v4_rshift(FULLV(i - 1, Q - 1, 1)); ipv = v4_rshift(FULLV(i - 1, Q - 1, 2)); }
      const v4 xBv = v4_set1(xf[i - 1].B);
      path[0] = v4_get(v4_mul(xBv, TFVQ(7 * q + 0)), r);
      path[1] = v4_get(v4_mul(mpv, TFVQ(7 * q + 1)), r);
      path[2] = v4_get(v4_mul(ipv, TFVQ(7 * q + 2)), r);
      path[3] = v4_get(v4_mul(dpv, TFVQ(7 * q + 3)), r);
      fnorm(path, 4);
      static const int st4[4] = { ST_B, ST_M, ST_I, ST_D };
      s1 = st4[fchoose(rng, path, 4)];
      k--; i--;
      break; }
    case ST_D: {
      const int q = (k - 1) % Q, r = (k - 1) / Q;
      v4 mpv, dpv, tmdv, tddv;
      if (q > 0) { mpv = FULLV(i, q - 1, 0); dpv = FULLV(i, q - 1, 1); tmdv = TFVQ(7 * (q - 1) + 4); tddv = TFVQ(7 * Q + (q - 1)); }
      else { mpv = v4_rshift(FULLV(i, Q - 1, 0)); dpv = v4_rshift(FULLV(i, Q - 1, 1)); tmdv = v4_rshift(TFVQ(7 * (Q - 1) + 4)); tddv = v4_rshift(TFVQ(8 * Q - 1)); }
      path[0] = v4_get(mpv, r) * v4_get(tmdv, r);
      path[1] = v4_get(dpv, r) * v4_get(tddv, r);
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? ST_M : ST_D;
      k--;
      break; }
    case ST_I: {
      const int q = (k - 1) % Q, r = (k - 1) / Q;
      path[0] = v4_get(v4_mul(FULLV(i - 1, q, 0), TFVQ(7 * q + 5)), r);
      path[1] = v4_get(v4_mul(FULLV(i - 1, q, 2), TFVQ(7 * q + 6)), r);
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? ST_M : ST_I;
      i--;
      break; }
    case ST_N: s1 = (i == 0) ? ST_S : ST_N; break;
    case ST_C:
      if (i < 1) return -3;
      path[0] = xf[i - 1].C * ploop;
      path[1] = xf[i].E * 0.5f * xf[i].SCALE;
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? ST_C : ST_E;
      break;
    case ST_J:
      if (i < 1) return -3;
      path[0] = xf[i - 1].J * ploop;
      path[1] = xf[i].E * 0.5f * xf[i].SCALE;
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? ST_J : ST_E;
      break;
    case ST_E: {
      double sum = 0.0;
      const double roll = rng_next(rng);
      const double norm = 1.0 / xf[i].E;
      const v4 xEv = v4_set1((float)norm);
      s1 = -1;
      while (s1 < 0) {
        for (int q = 0; q < Q && s1 < 0; q++) {
          v4 u = v4_mul(FULLV(i, q, 0), xEv);
          for (int r = 0; r < 4 && s1 < 0; r++) { sum += v4_get(u, r); if (roll < sum) { k = r * Q + q + 1; s1 = ST_M; } }
          if (s1 >= 0) break;
          u = v4_mul(FULLV(i, q, 1), xEv);
          for (int r = 0; r < 4 && s1 < 0; r++) { sum += v4_get(u, r); if (roll < sum) { k = r * Q + q + 1; s1 = ST_D; } }
        }
        if (s1 < 0 && sum < 0.99) return -1;        /* HMMER throws: probabilities were not normalised */
      }
      /* a new domain opens (read backwards: this is its end) */
      if (nd >= maxdom) return -2;
      cur = nd++;
      dfrom[cur] = dto[cur] = dk[cur] = dm[cur] = 0;
      for (int z = 0; z < Q * 4; z++) cntM[(size_t)cur * QMAX * 4 + z] = cntI[(size_t)cur * QMAX * 4 + z] = 0.f;
      break; }
    case ST_B:
      path[0] = xf[i].N * pmove;
      path[1] = xf[i].J * pmove;
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? ST_N : ST_J;
      break;
    default: return -3;
    }
    /* the appended state (s1, k, i): match and insert states carry the coordinates p7_trace_Index / p7_Null2_ByTrace read */
    if (s1 == ST_M) {
      if (dto[cur] == 0) { dto[cur] = i; dm[cur] = k; }     /* first seen = last in the path */
      dfrom[cur] = i; dk[cur] = k;
      cntM[(size_t)cur * QMAX * 4 + ((k - 1) % Q) * 4 + (k - 1) / Q] += 1.0f;
    } else if (s1 == ST_I) {
      cntI[(size_t)cur * QMAX * 4 + ((k - 1) % Q) * 4 + (k - 1) / Q] += 1.0f;
    }
    if ((s1 == ST_N || s1 == ST_J || s1 == ST_C) && s1 == s0) i--;
    s0 = s1;
  }
  return nd;
}

static int link_samples(const spcoord *h1, const spcoord *h2)
{
  const float min_overlap = 0.8f; const int max_diagdiff = 4;     /* of_smaller = TRUE */
  int nov = (h1->j < h2->j ? h1->j : h2->j) - (h1->i > h2->i ? h1->i : h2->i) + 1;
  int a = h1->j - h1->i + 1, b = h2->j - h2->i + 1;
  int n = a < b ? a : b;
  if ((float)nov / (float)n < min_overlap) return 0;
  nov = (h1->m < h2->m ? h1->m : h2->m) - (h1->k > h2->k ? h1->k : h2->k);       /* as published: no "+ 1" on the model side */
  a = h1->m - h1->k + 1; b = h2->m - h2->k + 1;
  n = a < b ? a : b;
  if ((float)nov / (float)n < min_overlap) return 0;
  int d1 = h1->i - h1->k, d2 = h2->i - h2->k;
  if (abs(d1 - d2) <= max_diagdiff) return 1;
  d1 = h1->j - h1->m; d2 = h2->j - h2->m;
  if (abs(d1 - d2) <= max_diagdiff) return 1;
  return 0;
}

/* region ireg..jreg of dsq[1..L]: fills n2sc[ireg..jreg] and returns the cluster envelopes (absolute coordinates),
 * ordered by start; ret < 0 if the ensemble could not be sampled (the region then yields no envelope) */
/* regions that ran into one of the bookkeeping limits below, by kind (the engine's MrOut.status codes): 1 unsampleable matrix,
 * 2 more than 8 domains in one path, 4 more than 512 distinct tuples, 6 more than 32 clusters, 7 more than 4 envelopes */
static long long g_mr_fail_kind[8];
void orc_mr_fail_counts(long long *out, int reset)
{
  for (int k = 0; k < 8; k++) { out[k] = g_mr_fail_kind[k]; if (reset) g_mr_fail_kind[k] = 0; }
}
static void mr_fail(int kind)
{
#pragma omp atomic
  g_mr_fail_kind[kind]++;
}
static int region_trace_ensemble(const orc_profile *p, const uint8_t *dsq, int L, int ireg, int jreg, workspace *w, int **env_i, int **env_j, int *envcap)
{
  const int Q = p->Q, Lr = jreg - ireg + 1, nsamples = 200;
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f), ploop = 1.0f - pmove;   /* multihit, length model of the whole target */
  float fsc;
  fwd_engine_x(p, dsq + ireg - 1, Lr, pmove, ploop, 0.5f, 0.5f, w->ef, NULL, w->full, &fsc);
  for (int pos = ireg; pos <= jreg; pos++) w->n2sc[pos] = 0.0f;
  orc_rng rng; rng_init(&rng, 42);
  /* no bookkeeping limit, as in hmmsearch (p7_domaindef.c / p7_spensemble.c grow their lists): a path of a region of Lr residues
   * has at most Lr domains; every list below is sized by that or grows.  (Until round 3 the device kernel's fixed sizes -- 8 domains
   * per path, 512 distinct tuples, 4 envelopes -- were mirrored here; its overflow path has none.)  What remains is what makes
   * hmmsearch itself throw: a matrix that cannot be sampled (kinds 1 and 5 below). */
  const int MAXD = Lr + 1;
  int *dfrom = (int *)malloc(sizeof(int) * 4 * (size_t)MAXD), *dto = dfrom + MAXD, *dk = dto + MAXD, *dm = dk + MAXD;
  float *cntM = (float *)malloc(sizeof(float) * (size_t)MAXD * QMAX * 4 * 2), *cntI = cntM + (size_t)MAXD * QMAX * 4;
  int cap = 1024, n = 0;
  spcoord *sp = (spcoord *)malloc(sizeof(spcoord) * cap);
  const uint8_t *degen = orc_degen_table();
  int bad = 0;
  for (int t = 0; t < nsamples && !bad; t++) {
    const int nd = stochastic_trace(&rng, p, w->full, w->ef, Lr, pmove, ploop, dfrom, dto, dk, dm, cntM, cntI, MAXD);
    if (nd < 0) { bad = 1; mr_fail(nd == -2 ? 2 : nd == -3 ? 5 : 1); break; }
    int hi = Lr;                                   /* positions above hi have had their contribution of this trace */
    for (int d = 0; d < nd; d++) {                 /* domains last-first */
      if (n == cap) { cap *= 2; sp = (spcoord *)realloc(sp, sizeof(spcoord) * cap); }
      sp[n].idx = t; sp[n].i = dfrom[d] + ireg - 1; sp[n].j = dto[d] + ireg - 1; sp[n].k = dk[d]; sp[n].m = dm[d]; sp[n].prob = 0.f; n++;
      /* p7_Null2_ByTrace over the domain's B..E segment: only match and insert states emit there */
      const float *cm = cntM + (size_t)d * QMAX * 4, *ci = cntI + (size_t)d * QMAX * 4;
      int Ld = 0;
      for (int z = 0; z < Q * 4; z++) Ld += (int)cm[z] + (int)ci[z];
      const float norm = (float)(1.0 / (double)(float)Ld);
      float null2[ORC_KP];
      const float xfactor = (0.0f * norm + 0.0f * norm) + 0.0f * norm;
      for (int x = 0; x < 4; x++) {
        v4 sv = v4_zero();
        const float *rp = p->rfv + (size_t)x * Q * 4;
        for (int q = 0; q < Q; q++) {
          v4 mv = v4_mul(v4_ld(cm + q * 4), v4_set1(norm)), iv = v4_mul(v4_ld(ci + q * 4), v4_set1(norm));
          sv = v4_add(sv, v4_mul(mv, v4_ld(rp))); rp += 4;
          sv = v4_add(sv, iv);
        }
        null2[x] = v4_hsum(sv);
        null2[x] += xfactor;
      }
      for (int x = 5; x <= 15; x++) {
        float result = 0.f; int ndg = 0;
        for (int y = 0; y < 4; y++) if (degen[x] & (1 << y)) { result += null2[y]; ndg++; }
        null2[x] = result / (float)ndg;
      }
      null2[4] = null2[16] = null2[17] = 1.0f;
      /* as published: residues up to AND INCLUDING the domain's first one count as outside (+1), the rest of it by null2 */
      for (int pos = hi; pos > dto[d]; pos--) w->n2sc[ireg + pos - 1] += 1.0f;
      for (int pos = dto[d]; pos > dfrom[d]; pos--) w->n2sc[ireg + pos - 1] += null2[dsq[ireg + pos - 1]];
      hi = dfrom[d];
    }
    for (int pos = hi; pos >= 1; pos--) w->n2sc[ireg + pos - 1] += 1.0f;
  }
  if (bad) {          /* HMMER would have thrown; keep the region without envelopes and without a null2 correction */
    for (int pos = ireg; pos <= jreg; pos++) w->n2sc[pos] = 0.0f;
    free(cntM); free(sp); free(dfrom);
    return -1;
  }
  for (int pos = ireg; pos <= jreg; pos++) w->n2sc[pos] = orc_logf(w->n2sc[pos] / (float)nsamples);

  /* single-linkage clustering of the sampled (i, j, k, m); components numbered by their smallest member */
  int *comp = (int *)malloc(sizeof(int) * (n + 1)), *stack = (int *)malloc(sizeof(int) * (n + 1));
  for (int h = 0; h < n; h++) comp[h] = -1;
  int nc = 0;
  for (int h0 = 0; h0 < n; h0++) {
    if (comp[h0] >= 0) continue;
    int ns = 0; stack[ns++] = h0; comp[h0] = nc;
    while (ns > 0) {
      const int v = stack[--ns];
      for (int u = 0; u < n; u++) if (comp[u] < 0 && link_samples(&sp[v], &sp[u])) { comp[u] = nc; stack[ns++] = u; }
    }
    nc++;
  }
  spcoord *sig = (spcoord *)malloc(sizeof(spcoord) * (nc + 1));
  int nsig = 0;
  int *epc = (int *)malloc(sizeof(int) * (L + p->M + 8));
  for (int c = 0; c < nc; c++) {
    int ninc = 0, last = -1;
    for (int h = 0; h < n; h++) if (comp[h] == c) { if (sp[h].idx != last) ninc++; last = sp[h].idx; }
    if ((float)ninc / (float)nsamples < 0.25f) continue;
    int imin = 0, imax = 0, jmin = 0, jmax = 0, kmin = 0, kmax = 0, mmin = 0, mmax = 0;
    for (int h = 0; h < n; h++) if (comp[h] == c) {
      if (imin == 0) { imin = imax = sp[h].i; jmin = jmax = sp[h].j; kmin = kmax = sp[h].k; mmin = mmax = sp[h].m; }
      else {
        if (sp[h].i < imin) imin = sp[h].i;
        if (sp[h].i > imax) imax = sp[h].i;
        if (sp[h].j < jmin) jmin = sp[h].j;
        if (sp[h].j > jmax) jmax = sp[h].j;
        if (sp[h].k < kmin) kmin = sp[h].k;
        if (sp[h].k > kmax) kmax = sp[h].k;
        if (sp[h].m < mmin) mmin = sp[h].m;
        if (sp[h].m > mmax) mmax = sp[h].m;
      }
    }
    const int thr = (int)ceilf((float)ninc * 0.02f);
    int best_i, best_j, best_k, best_m, am;
#define HIST(field, lo, hi) do { for (int z = 0; z <= (hi) - (lo); z++) epc[z] = 0; \
      for (int h = 0; h < n; h++) if (comp[h] == c) epc[sp[h].field - (lo)]++; \
      am = 0; for (int z = 1; z <= (hi) - (lo); z++) if (epc[z] > epc[am]) am = z; } while (0)
    HIST(i, imin, imax);
    for (best_i = imin; best_i <= imax; best_i++) if (epc[best_i - imin] >= thr) break;
    if (best_i > imax) best_i = imin + am;
    HIST(k, kmin, kmax);
    for (best_k = kmin; best_k <= kmax; best_k++) if (epc[best_k - kmin] >= thr) break;
    if (best_k > kmax) best_k = kmin + am;
    HIST(j, jmin, jmax);
    for (best_j = jmax; best_j >= jmin; best_j--) if (epc[best_j - jmin] >= thr) break;
    if (best_j < jmin) best_j = jmin + am;
    HIST(m, mmin, mmax);
    for (best_m = mmax; best_m >= mmin; best_m--) if (epc[best_m - mmin] >= thr) break;
    if (best_m < mmin) best_m = mmin + am;
#undef HIST
    if (best_i > best_j || best_k > best_m) continue;
    sig[nsig].i = best_i; sig[nsig].j = best_j; sig[nsig].k = best_k; sig[nsig].m = best_m; sig[nsig].idx = c;
    sig[nsig].prob = (float)ninc / (float)nsamples;
    nsig++;
  }
  /* order by start (stable) */
  for (int a = 1; a < nsig; a++) { spcoord t = sig[a]; int b = a - 1; while (b >= 0 && sig[b].i > t.i) { sig[b + 1] = sig[b]; b--; } sig[b + 1] = t; }
  /* dominated clusters (>= 80 % overlap of the shorter one): the less probable goes */
  int *dominated = (int *)calloc(nsig + 1, sizeof(int));
  for (int d = 0; d < nsig; d++)
    for (int d2 = d + 1; d2 < nsig; d2++) {
      const int nov = (sig[d].j < sig[d2].j ? sig[d].j : sig[d2].j) - (sig[d].i > sig[d2].i ? sig[d].i : sig[d2].i) + 1;
      if (nov == 0) break;
      const int a = sig[d].j - sig[d].i + 1, b = sig[d2].j - sig[d2].i + 1;
      const int nn = a < b ? a : b;
      if ((float)nov / (float)nn >= 0.8f) { if (sig[d].prob > sig[d2].prob) dominated[d2] = 1; else dominated[d] = 1; }
    }
  int ne = 0;
  for (int d = 0; d < nsig && !bad; d++) if (!dominated[d]) {
    if (ne >= *envcap) { *envcap = 2 * *envcap + 8; *env_i = (int *)realloc(*env_i, sizeof(int) * *envcap); *env_j = (int *)realloc(*env_j, sizeof(int) * *envcap); }
    (*env_i)[ne] = sig[d].i; (*env_j)[ne] = sig[d].j; ne++;
  }
  free(dominated); free(epc); free(sig); free(comp); free(stack); free(cntM); free(sp); free(dfrom);
  return ne;
}
#undef FULLV
#undef TFVQ

/* one (sequence, profile) comparison: everything p7_Pipeline does for it */
static void pipeline_pair(const orc_profile *p, int prof, int64_t seq, const uint8_t *dsq, int L,
                          double T, double F1, double F2, double F3, workspace *w,
                          orc_results *r, int keep_trace, double log_omega)
{
  orc_pairtrace tr; memset(&tr, 0, sizeof(tr));
  tr.seq = seq; tr.prof = prof;
  r->n_pairs++;
  if (L == 0) return;
  float nullsc = orc_nullsc(L);
  float usc, filtersc, fwdsc;
  int xJ;
  double P, seq_score_d;
  tr.nullsc = nullsc;
  orc_msv(p, dsq, L, &xJ, &usc);
  tr.msv_xj = xJ; tr.msv_sc = usc;
  seq_score_d = (double)(usc - nullsc) / LOG2;
  P = gumbel_surv(seq_score_d, p->evparam[0], p->evparam[1]);
  if (P > F1) { if (keep_trace >= 2) res_push_trace(r, &tr); return; }
  tr.pass_msv = 1; r->n_past_msv++;
  filtersc = orc_bias_filtersc(p, dsq, L);
  tr.filtersc = filtersc;
  seq_score_d = (double)(usc - filtersc) / LOG2;
  P = gumbel_surv(seq_score_d, p->evparam[0], p->evparam[1]);
  if (P > F1) { if (keep_trace) res_push_trace(r, &tr); return; }
  tr.pass_bias = 1; r->n_past_bias++;
  if (P > F2) {                 /* second filter; never entered when F1 == F2 (the reference's flags) */
    float vfsc;
    orc_vitfilter(p, dsq, L, &vfsc);
    tr.ran_vit = 1; tr.vitsc = vfsc;
    seq_score_d = (double)(vfsc - filtersc) / LOG2;
    P = gumbel_surv(seq_score_d, p->evparam[2], p->evparam[3]);
    if (P > F2) { if (keep_trace) res_push_trace(r, &tr); return; }
  }
  tr.pass_vit = 1;
  ws_grow(w, L, p->Q);
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
  const float ploop = 1.0f - pmove;
  int st = fwd_engine(p, dsq, L, pmove, ploop, 0.5f, 0.5f, w->xf, NULL, &fwdsc);
  if (st != 0) { if (keep_trace) res_push_trace(r, &tr); return; }
  tr.fwdsc = fwdsc;
  seq_score_d = (double)(fwdsc - filtersc) / LOG2;
  P = exp_surv(seq_score_d, p->evparam[4], p->evparam[5]);
  if (P > F3) { if (keep_trace) res_push_trace(r, &tr); return; }
  tr.pass_fwd = 1; r->n_past_fwd++;
  int own = 0; float bcksc = 0.f;
  st = bwd_engine(p, dsq, L, pmove, ploop, 0.5f, 0.5f, w->xf, w->xb, &own, NULL, &bcksc);
  tr.bcksc = bcksc;
  if (st != 0) { if (keep_trace) res_push_trace(r, &tr); return; }

  /* posterior decoding of domain starts/ends/occupancy */
  {
    float scaleproduct = (float)(1.0 / (double)w->xb[0].N);
    w->btot[0] = 0.f; w->etot[0] = 0.f; w->mocc[0] = 0.f;
    for (int i = 1; i <= L; i++) {
      w->btot[i] = w->btot[i - 1] + (w->xf[i - 1].B * w->xb[i - 1].B * w->xf[i - 1].SCALE * scaleproduct);
      if (own) scaleproduct *= w->xf[i - 1].SCALE / w->xb[i - 1].SCALE;
      w->etot[i] = w->etot[i - 1] + (w->xf[i].E * w->xb[i].E * w->xf[i].SCALE * scaleproduct);
      float njcp;
      njcp  = w->xf[i - 1].N * w->xb[i].N * ploop * scaleproduct;
      njcp += w->xf[i - 1].J * w->xb[i].J * ploop * scaleproduct;
      njcp += w->xf[i - 1].C * w->xb[i].C * ploop * scaleproduct;
      w->mocc[i] = (float)(1. - (double)njcp);
    }
    if (isinf(scaleproduct)) { if (keep_trace) res_push_trace(r, &tr); return; }
  }
  for (int i = 0; i <= L; i++) w->n2sc[i] = 0.0f;

  /* region scan */
  domrec *doms = NULL; int ndom = 0, domcap = 0, nregions = 0;        /* as many as the target has (hmmsearch keeps them all) */
#define PUSH_DOM(d_) do { if (ndom == domcap) { domcap = 2 * domcap + 16; doms = (domrec *)realloc(doms, sizeof(domrec) * domcap); } doms[ndom++] = (d_); } while (0)
  int *ei = NULL, *ej = NULL, envcap = 0;
  {
    const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
    int i = -1, triggered = 0;
    for (int j = 1; j <= L; j++) {
      if (!triggered) {
        if (w->mocc[j] - (w->btot[j] - w->btot[j - 1]) < rt2) i = j;
        else if (i == -1) i = j;
        if (w->mocc[j] >= rt1) triggered = 1;
      } else if (w->mocc[j] - (w->etot[j] - w->etot[j - 1]) < rt2) {
        nregions++;
        int multi = 0;
        { float max = -1.0f;
          for (int z = i; z <= j; z++) {
            float a = w->etot[z] - w->etot[i - 1], b = w->btot[j] - w->btot[z - 1];
            float e = a < b ? a : b;
            if (e > max) max = e;
          }
          multi = (max >= rt3); }
        if (multi && !getenv("ORC_NO_ENSEMBLE")) {
          /* the region is resolved into envelopes by stochastic traceback clustering; its null2 scores come from the traces */
          r->n_multidomain++;
          const int ne = region_trace_ensemble(p, dsq, L, i, j, w, &ei, &ej, &envcap);
          for (int e = 0; e < ne; e++) {
            domrec d;
            if (rescore_domain(p, dsq, L, ei[e], ej[e], 1, w, &d) == 0) PUSH_DOM(d);
          }
          if (ne < 0) {
            /* the matrix could not be sampled (hmmsearch throws there; the engine reports such a search): the region is kept the
             * way it was before the ensemble stage existed, as ONE envelope with null2 by expectation, flagged */
            domrec d;
            if (rescore_domain(p, dsq, L, i, j, 0, w, &d) == 0) { d.flags |= 1; PUSH_DOM(d); }
          }
        } else {
          domrec d;
          if (rescore_domain(p, dsq, L, i, j, 0, w, &d) == 0) {
            if (multi) { d.flags |= 1; r->n_multidomain++; }
            PUSH_DOM(d);
          }
        }
        i = -1; triggered = 0;
      }
    }
  }
  tr.nregions = nregions; tr.ndom = ndom;
  if (keep_trace) res_push_trace(r, &tr);
  if (nregions == 0 || ndom == 0) { free(doms); free(ei); free(ej); return; }

  /* per-sequence score with null2 correction, and the reconstruction score */
  float seqbias = 0.0f;
  for (int i = 0; i <= L; i++) seqbias += w->n2sc[i];
  seqbias = flogsum(0.0f, (float)(log_omega + (double)seqbias));
  float seq_score = (float)((double)(fwdsc - (nullsc + seqbias)) / LOG2);
  const double lnn3 = log((double)((float)L / (float)(L + 3)));
  float sum_score = 0.0f; float sbias = 0.0f; int Ldsum = 0;
  for (int d = 0; d < ndom; d++)
    if (doms[d].envsc - doms[d].domcorrection > 0.0f) {
      sum_score += doms[d].envsc;
      Ldsum += doms[d].jenv - doms[d].ienv + 1;
      sbias += doms[d].domcorrection;
    }
  sbias = flogsum(0.0f, (float)(log_omega + (double)sbias));
  sum_score = (float)((double)sum_score + (double)(L - Ldsum) * lnn3);
  sum_score = (float)((double)(sum_score - (nullsc + sbias)) / LOG2);
  float final_bias = seqbias;
  if (Ldsum > 0 && sum_score > seq_score) { seq_score = sum_score; final_bias = sbias; }
  int seq_rep = ((double)seq_score >= T);
  for (int d = 0; d < ndom; d++) {
    orc_domain o; memset(&o, 0, sizeof(o));
    int Ld = doms[d].jenv - doms[d].ienv + 1;
    float bits = (float)((double)doms[d].envsc + (double)(L - Ld) * lnn3);
    float dombias = flogsum(0.0f, (float)(log_omega + (double)doms[d].domcorrection));
    bits = (float)((double)(bits - (nullsc + dombias)) / LOG2);
    o.seq = seq; o.prof = prof; o.tlen = L; o.ienv = doms[d].ienv; o.jenv = doms[d].jenv;
    o.dom_idx = d; o.ndom = ndom; o.flags = doms[d].flags; o.envsc = doms[d].envsc;
    o.domcorrection = doms[d].domcorrection; o.dombias = dombias; o.bitscore = bits;
    o.lnP = exp_logsurv((double)bits, p->evparam[4], p->evparam[5]);
    o.seq_score = seq_score; o.seq_bias = (float)((double)final_bias / LOG2);
    o.seq_reported = seq_rep; o.dom_reported = 0;
    res_push_dom(r, &o);
  }
  free(doms); free(ei); free(ej);
#undef PUSH_DOM
}

static int cmp_dom(const void *a, const void *b)
{
  const orc_domain *x = (const orc_domain *)a, *y = (const orc_domain *)b;
  if (x->prof != y->prof) return x->prof < y->prof ? -1 : 1;
  if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
  return x->dom_idx - y->dom_idx;
}
static int cmp_trace(const void *a, const void *b)
{
  const orc_pairtrace *x = (const orc_pairtrace *)a, *y = (const orc_pairtrace *)b;
  if (x->prof != y->prof) return x->prof < y->prof ? -1 : 1;
  if (x->seq != y->seq) return x->seq < y->seq ? -1 : 1;
  return 0;
}

orc_results *orc_search(const orc_hmmset *hs, const uint8_t *codes, const int64_t *offsets, int64_t nseq,
                        double T, double F1, double F2, double F3, int keep_trace, int nthreads)
{
  flogsum_init();
  const double log_omega = log((double)(1.0f / 256.0f));
  orc_results *R = (orc_results *)calloc(1, sizeof(*R));
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
  omp_set_num_threads(nthreads);
#else
  nthreads = 1;
#endif
  orc_results *part = (orc_results *)calloc(nthreads, sizeof(orc_results));
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    unsigned int csr = _mm_getcsr();
    _mm_setcsr(csr | 0x8040);           /* FTZ | DAZ, as HMMER's SSE implementation sets */
    workspace w; memset(&w, 0, sizeof(w));
    uint8_t *dsq = NULL; int64_t dcap = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int64_t s = 0; s < nseq; s++) {
      int L = (int)(offsets[s + 1] - offsets[s]);
      if (L + 2 > dcap) { dcap = L + 64; dsq = (uint8_t *)realloc(dsq, dcap); }
      dsq[0] = 17;
      memcpy(dsq + 1, codes + offsets[s], L);
      dsq[L + 1] = 17;
      for (int pi = 0; pi < hs->n; pi++)
        pipeline_pair(&hs->p[pi], pi, s, dsq, L, T, F1, F2, F3, &w, &part[tid], keep_trace, log_omega);
    }
    free(dsq); ws_free(&w);
    _mm_setcsr(csr);
  }
  /* the domain rows of all threads, in (profile, sequence, domain) order.  Every (sequence, profile) pair was scored by exactly one
   * thread and its rows carry ndom, so the place of every row follows from a count per pair and a prefix sum: no comparison sort,
   * and the scatter runs on all threads (a serial merge + qsort of millions of 80-byte rows was most of the timed CPU-baseline leg
   * on a 256-thread host) */
  int64_t total_dom = 0;
  for (int t = 0; t < nthreads; t++) total_dom += part[t].n_dom;
  if (total_dom > 0) {
    const int64_t np = (int64_t)hs->n * nseq;
    int64_t *off = (int64_t *)calloc((size_t)np + 1, sizeof(int64_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1)
#endif
    for (int t = 0; t < nthreads; t++)
      for (int64_t i = 0; i < part[t].n_dom; i++)
        if (part[t].dom[i].dom_idx == 0) off[(int64_t)part[t].dom[i].prof * nseq + part[t].dom[i].seq] = part[t].dom[i].ndom;
    int64_t run = 0;
    for (int64_t k = 0; k < np; k++) { const int64_t c = off[k]; off[k] = run; run += c; }
    R->dom = (orc_domain *)malloc(sizeof(orc_domain) * (size_t)total_dom);
    R->n_dom = total_dom; R->cap_dom = total_dom;
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1)
#endif
    for (int t = 0; t < nthreads; t++)
      for (int64_t i = 0; i < part[t].n_dom; i++) {
        const orc_domain *d = &part[t].dom[i];
        R->dom[off[(int64_t)d->prof * nseq + d->seq] + d->dom_idx] = *d;
      }
    free(off);
  }
  for (int t = 0; t < nthreads; t++) {
    for (int64_t i = 0; i < part[t].n_trace; i++) res_push_trace(R, &part[t].trace[i]);
    R->n_pairs += part[t].n_pairs; R->n_past_msv += part[t].n_past_msv; R->n_past_bias += part[t].n_past_bias;
    R->n_past_fwd += part[t].n_past_fwd; R->n_multidomain += part[t].n_multidomain;
    free(part[t].dom); free(part[t].trace);
  }
  free(part);
  if (R->n_trace) qsort(R->trace, R->n_trace, sizeof(orc_pairtrace), cmp_trace);
  return R;
}

void orc_threshold(orc_results *r, const orc_hmmset *hs, const int64_t *domZ_override, double domE)
{
  int64_t *domZ = (int64_t *)calloc(hs->n, sizeof(int64_t));
  for (int64_t i = 0; i < r->n_dom; i++)
    if (r->dom[i].dom_idx == 0 && r->dom[i].seq_reported) domZ[r->dom[i].prof]++;
  if (domZ_override) memcpy(domZ, domZ_override, sizeof(int64_t) * hs->n);
  for (int64_t i = 0; i < r->n_dom; i++) {
    orc_domain *d = &r->dom[i];
    d->dom_reported = d->seq_reported && (orc_exp(d->lnP) * (double)domZ[d->prof] <= domE);
  }
  free(domZ);
}

void orc_results_free(orc_results *r) { if (!r) return; free(r->dom); free(r->trace); free(r); }
int64_t orc_results_ndom(const orc_results *r) { return r->n_dom; }
const orc_domain *orc_results_dom(const orc_results *r) { return r->dom; }
int64_t orc_results_ntrace(const orc_results *r) { return r->n_trace; }
const orc_pairtrace *orc_results_trace(const orc_results *r) { return r->trace; }
void orc_results_counts(const orc_results *r, int64_t *o)
{ o[0] = r->n_pairs; o[1] = r->n_past_msv; o[2] = r->n_past_bias; o[3] = r->n_past_fwd; o[4] = r->n_multidomain; }

/* printf("%.1f") of a float, as tenths: exact because float*10 is exact in double */
static int64_t tenths(float bits) { return (int64_t)rint((double)bits * 10.0); }

void orc_positions(const orc_results *r, const orc_hmmset *hs, int64_t nseq,
                   const char *leftprefix, const char *rightprefix,
                   int32_t *start, int32_t *stop, int32_t *tlen, int32_t *in_ddict)
{
  int64_t *lsc = (int64_t *)malloc(sizeof(int64_t) * nseq), *rsc = (int64_t *)malloc(sizeof(int64_t) * nseq);
  for (int64_t s = 0; s < nseq; s++) { start[s] = stop[s] = tlen[s] = -1; in_ddict[s] = 0; lsc[s] = rsc[s] = INT64_MIN; }
  size_t ll = strlen(leftprefix), rl = strlen(rightprefix);
  /* rows are in domtblout order: profile (file) order, then target, then domain index */
  for (int64_t i = 0; i < r->n_dom; i++) {
    const orc_domain *d = &r->dom[i];
    if (!d->dom_reported) continue;
    const char *name = hs->p[d->prof].name;
    int64_t s = d->seq;
    in_ddict[s] = 1;
    int64_t sc = tenths(d->bitscore);
    if (strncmp(name, leftprefix, ll) == 0) {
      if (lsc[s] == INT64_MIN) { lsc[s] = sc; start[s] = d->jenv; tlen[s] = d->tlen; }
      else if (sc > lsc[s]) { lsc[s] = sc; start[s] = d->jenv; }
    } else if (strncmp(name, rightprefix, rl) == 0) {
      if (rsc[s] == INT64_MIN) { rsc[s] = sc; stop[s] = d->ienv - 1; tlen[s] = d->tlen; }
      else if (sc > rsc[s]) { rsc[s] = sc; stop[s] = d->ienv - 1; }
    }
  }
  free(lsc); free(rsc);
}
