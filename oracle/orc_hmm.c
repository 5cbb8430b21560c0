/*
 * orc_hmm.c -- ORACLE (test infrastructure only): HMMER3/f text parser and profile
 * configuration, restating what `hmmsearch` does to each query model before it
 * scores anything (reference call site: itsxpress/SeqSample.py:191-209; model
 * files: itsxpress/ITSx_db/HMMs/ (one .hmm per taxon); selection: itsxpress/main.py:176-231).
 *
 * HMMER (>=3.1b2, recipes/itsxpress/meta.yaml:36) is not vendored in the
 * reference; this restates its published algorithm: p7_hmmfile (ASCII 3/f),
 * p7_ProfileConfig (multihit local), p7_oprofile_Convert (MSV byte costs,
 * striped odds-ratio floats via the Cephes-style vector expf), p7_bg_SetFilter.
 * PARITY UNPINNED against a real hmmsearch build (none available here).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc.h"

#define LOG2  0.69314718055994529
#define LOG2R 1.44269504088896341

enum { T_MM = 0, T_MI, T_MD, T_IM, T_II, T_DM, T_DD, T_BM };

/* degeneracy sets over ACGT for codes 0..17 (A C G T - R Y M K S W H B V D N * ~) */
static const uint8_t DEGEN[ORC_KP] = {
  0x1, 0x2, 0x4, 0x8, 0x0,
  0x1 | 0x4,        /* R = AG */
  0x2 | 0x8,        /* Y = CT */
  0x1 | 0x2,        /* M = AC */
  0x4 | 0x8,        /* K = GT */
  0x2 | 0x4,        /* S = CG */
  0x1 | 0x8,        /* W = AT */
  0x1 | 0x2 | 0x8,  /* H = ACT */
  0x2 | 0x4 | 0x8,  /* B = CGT */
  0x1 | 0x2 | 0x4,  /* V = ACG */
  0x1 | 0x4 | 0x8,  /* D = AGT */
  0xF,              /* N */
  0x0, 0x0 };

const uint8_t *orc_degen_table(void) { return DEGEN; }

int orc_digitize(const char *a, int64_t len, uint8_t *out)
{
  static int8_t map[256]; static int init = 0;
  if (!init) {
    memset(map, -1, sizeof(map));
    const char *sym = "ACGT-RYMKSWHBVDN";
    for (int i = 0; i < 16; i++) { map[(unsigned char)sym[i]] = (int8_t)i; map[(unsigned char)(sym[i] | 0x20)] = (int8_t)i; }
    map['-'] = -1;                               /* gaps are not legal in unaligned input */
    map['U'] = map['u'] = 3; map['X'] = map['x'] = 15;
    init = 1;
  }
  for (int64_t i = 0; i < len; i++) {
    int c = map[(unsigned char)a[i]];
    if (c < 0) return -1;
    out[i] = (uint8_t)c;
  }
  return 0;
}

/* Scalar form of the 4-lane vector expf HMMER uses when it converts log-odds to
 * odds ratios (Cephes polynomial, float arithmetic, no FMA). */
static float vec_expf(float x)
{
  static const float p0 = 1.9875691500E-4f, p1 = 1.3981999507E-3f, p2 = 8.3334519073E-3f,
                     p3 = 4.1665795894E-2f, p4 = 1.6666665459E-1f, p5 = 5.0000001201E-1f;
  static const float c0 = 0.693359375f, c1 = -2.12194440e-4f;
  static const float maxlogf = 88.72283905206835f, minlogf = -103.27892990343185f;
  if (x > maxlogf) return INFINITY;
  if (x <= minlogf) return 0.0f;
  volatile float fx = x * (float)LOG2R;
  fx = fx + 0.5f;
  int k = (int)fx;                 /* truncation */
  float tmp = (float)k;
  if (tmp > fx) tmp = tmp - 1.0f;  /* floor */
  fx = tmp;
  k = (int)fx;
  volatile float t = fx * c0;
  volatile float z = fx * c1;
  volatile float xr = x - t;
  xr = xr - z;
  z = xr * xr;
  volatile float y = p0;
  y = y * xr; y = y + p1;
  y = y * xr; y = y + p2;
  y = y * xr; y = y + p3;
  y = y * xr; y = y + p4;
  y = y * xr; y = y + p5;
  y = y * z;
  y = y + xr;
  y = y + 1.0f;
  union { uint32_t u; float f; } pw; pw.u = (uint32_t)(k + 127) << 23;
  y = y * pw.f;
  return y;
}

static void profile_free_members(orc_profile *p)
{
  free(p->t); free(p->mat); free(p->tsc); free(p->msc); free(p->rbv); free(p->rfv); free(p->tfv); free(p->rww); free(p->tww);
}

static float prob_from_tok(const char *tok)
{
  if (tok[0] == '*') return 0.0f;
  return expf((float)(-1.0 * atof(tok)));
}

static uint8_t unbiased_byteify(float scale_b, float sc)
{
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255.0f) ? 255 : (uint8_t)(int)sc;
}
static uint8_t biased_byteify(float scale_b, uint8_t bias_b, float sc)
{
  sc = -1.0f * roundf(scale_b * sc);
  if (sc > (float)(255 - bias_b)) return 255;
  return (uint8_t)((int)sc + (int)bias_b);
}

uint8_t orc_tjb_b(int L)
{
  float scale_b = (float)(3.0 / LOG2);
  return unbiased_byteify(scale_b, logf(3.0f / (float)(L + 3)));
}

static int configure(orc_profile *p)
{
  const int M = p->M;
  const float f[4] = { 0.25f, 0.25f, 0.25f, 0.25f };
  p->tsc = (float *)malloc(sizeof(float) * (M + 1) * 8);
  p->msc = (float *)malloc(sizeof(float) * (M + 1) * ORC_KP);
  for (int i = 0; i < (M + 1) * 8; i++) p->tsc[i] = -INFINITY;
  for (int i = 0; i < (M + 1) * ORC_KP; i++) p->msc[i] = -INFINITY;

  /* match occupancy -> local entry distribution */
  float *occ = (float *)calloc(M + 1, sizeof(float));
  occ[0] = 0.0f;
  occ[1] = p->t[0 * 7 + T_MI] + p->t[0 * 7 + T_MM];
  for (int k = 2; k <= M; k++)
    occ[k] = (float)((double)(occ[k - 1] * (p->t[(k - 1) * 7 + T_MM] + p->t[(k - 1) * 7 + T_MI])) +
                     (1.0 - (double)occ[k - 1]) * (double)p->t[(k - 1) * 7 + T_DM]);
  float Z = 0.0f;
  for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M - k + 1);
  for (int k = 1; k <= M; k++) p->tsc[(k - 1) * 8 + T_BM] = (float)log((double)(occ[k] / Z));
  free(occ);

  for (int k = 1; k < M; k++)
    for (int x = 0; x < 7; x++) p->tsc[k * 8 + x] = (float)log((double)p->t[k * 7 + x]);

  for (int k = 1; k <= M; k++) {
    float sc[ORC_KP];
    for (int x = 0; x < ORC_KP; x++) sc[x] = -INFINITY;
    for (int x = 0; x < 4; x++) sc[x] = (float)log((double)p->mat[k * 4 + x] / (double)f[x]);
    for (int x = 5; x <= 15; x++) {        /* degenerate residues: expected score */
      float result = 0.0f, denom = 0.0f;
      for (int y = 0; y < 4; y++) if (DEGEN[x] & (1 << y)) { result += sc[y] * f[y]; denom += f[y]; }
      sc[x] = result / denom;
    }
    for (int x = 0; x < ORC_KP; x++) p->msc[k * ORC_KP + x] = sc[x];
  }

  /* --- MSV byte model --- */
  float max = 0.0f;
  for (int x = 0; x < 4; x++) for (int k = 1; k <= M; k++) if (p->msc[k * ORC_KP + x] > max) max = p->msc[k * ORC_KP + x];
  p->scale_b = (float)(3.0 / LOG2);
  p->base_b = 190;
  p->bias_b = unbiased_byteify(p->scale_b, (float)(-1.0 * (double)max));
  p->rbv = (uint8_t *)malloc((size_t)ORC_KP * (M + 1));
  for (int x = 0; x < ORC_KP; x++) {
    p->rbv[x * (M + 1)] = 255;
    for (int k = 1; k <= M; k++) p->rbv[x * (M + 1) + k] = biased_byteify(p->scale_b, p->bias_b, p->msc[k * ORC_KP + x]);
  }
  p->tbm_b = unbiased_byteify(p->scale_b, logf(2.0f / ((float)M * (float)(M + 1))));
  p->tec_b = unbiased_byteify(p->scale_b, logf(0.5f));

  /* --- Forward/Backward striped odds ratios --- */
  int Q = (M + 3) / 4; if (Q < 2) Q = 2;
  p->Q = Q;
  p->rfv = (float *)malloc(sizeof(float) * ORC_KP * Q * 4);
  p->tfv = (float *)malloc(sizeof(float) * 8 * Q * 4);
  for (int x = 0; x < ORC_KP; x++)
    for (int q = 0; q < Q; q++)
      for (int z = 0; z < 4; z++) {
        int k = q + 1 + z * Q;
        float v = (k <= M) ? p->msc[k * ORC_KP + x] : -INFINITY;
        p->rfv[(x * Q + q) * 4 + z] = vec_expf(v);
      }
  for (int q = 0; q < Q; q++) {
    int k = q + 1;
    static const int tg[7] = { T_BM, T_MM, T_IM, T_DM, T_MD, T_MI, T_II };
    static const int off[7] = { -1, -1, -1, -1, 0, 0, 0 };
    for (int t = 0; t < 7; t++)
      for (int z = 0; z < 4; z++) {
        int kb = k + off[t] + z * Q;
        float v = (kb < M) ? p->tsc[kb * 8 + tg[t]] : -INFINITY;
        p->tfv[(q * 7 + t) * 4 + z] = vec_expf(v);
      }
  }
  for (int q = 0; q < Q; q++)
    for (int z = 0; z < 4; z++) {
      int k = q + 1 + z * Q;
      float v = (k < M) ? p->tsc[k * 8 + T_DD] : -INFINITY;
      p->tfv[(7 * Q + q) * 4 + z] = vec_expf(v);
    }

  /* --- Viterbi filter word model (vf_conversion): wordify() every score, no transition above 0, no II above -1 --- */
  {
    const float scale_w = (float)(500.0 / LOG2);
    p->rww = (int16_t *)malloc(sizeof(int16_t) * ORC_KP * (M + 1));
    p->tww = (int16_t *)malloc(sizeof(int16_t) * 8 * (M + 1));
#define WORDIFY(dst, scv) do { float w_ = roundf(scale_w * (scv)); (dst) = (w_ >= 32767.0f) ? 32767 : (w_ <= -32768.0f) ? -32768 : (int16_t)w_; } while (0)
    for (int x = 0; x < ORC_KP; x++) {
      p->rww[x * (M + 1)] = -32768;
      for (int k = 1; k <= M; k++) WORDIFY(p->rww[x * (M + 1) + k], p->msc[k * ORC_KP + x]);
    }
    static const int into[4] = { T_BM, T_MM, T_IM, T_DM }, outof[4] = { T_MD, T_MI, T_II, T_DD };
    for (int t = 0; t < 8; t++) p->tww[t * (M + 1)] = -32768;
    for (int k = 1; k <= M; k++) {
      for (int t = 0; t < 4; t++) {                      /* from node k-1 (B for BM) into node k */
        int16_t v; WORDIFY(v, p->tsc[(k - 1) * 8 + into[t]]);
        p->tww[t * (M + 1) + k] = v > 0 ? 0 : v;
      }
      for (int t = 0; t < 4; t++) {                      /* out of node k; the last node has none */
        int16_t v = -32768;
        if (k < M) WORDIFY(v, p->tsc[k * 8 + outof[t]]);
        const int16_t maxval = (outof[t] == T_II) ? -1 : 0;
        p->tww[(4 + t) * (M + 1) + k] = (outof[t] == T_DD) ? v : (v > maxval ? maxval : v);
      }
    }
#undef WORDIFY
  }

  /* --- bias-composition filter HMM --- */
  {
    float L0 = 400.0f;
    float L1 = (float)((double)(float)M / 8.0);
    p->ft[0][0] = L0 / (L0 + 1.0f);
    p->ft[0][1] = 1.0f / (L0 + 1.0f);
    p->ft[0][2] = 1.0f;
    p->ft[1][0] = 1.0f / (L1 + 1.0f);
    p->ft[1][1] = L1 / (L1 + 1.0f);
    p->ft[1][2] = 1.0f;
    p->fpi[0] = (float)0.999; p->fpi[1] = (float)0.001;
    float e[2][4];
    for (int x = 0; x < 4; x++) { e[0][x] = f[x]; e[1][x] = p->compo[x]; }
    for (int x = 0; x < ORC_KP; x++) for (int k = 0; k < 2; k++) p->feo[x][k] = 1.0f;
    for (int x = 0; x < 4; x++) for (int k = 0; k < 2; k++) p->feo[x][k] = e[k][x] / f[x];
    for (int x = 5; x <= 15; x++)
      for (int k = 0; k < 2; k++) {
        float num = 0.0f, denom = 0.0f;
        for (int y = 0; y < 4; y++) if (DEGEN[x] & (1 << y)) { num += e[k][y]; denom += f[y]; }
        p->feo[x][k] = (denom > 0.0f) ? num / denom : 0.0f;
      }
  }
  return 0;
}

/* tokenizer over one line */
static int split_ws(char *line, char **tok, int maxtok)
{
  int n = 0; char *s = line;
  while (*s && n < maxtok) {
    while (*s == ' ' || *s == '\t' || *s == '\r') s++;
    if (!*s || *s == '\n') break;
    tok[n++] = s;
    while (*s && *s != ' ' && *s != '\t' && *s != '\n' && *s != '\r') s++;
    if (*s) { *s = 0; s++; }
  }
  return n;
}

orc_hmmset *orc_hmmset_parse(const char *text, int64_t len, char *err, int errlen)
{
  char *buf = (char *)malloc((size_t)len + 1);
  memcpy(buf, text, (size_t)len); buf[len] = 0;
  /* index lines */
  int64_t nl = 0, cap = 1024; char **lines = (char **)malloc(sizeof(char *) * cap);
  for (char *s = buf; *s;) {
    if (nl == cap) { cap *= 2; lines = (char **)realloc(lines, sizeof(char *) * cap); }
    lines[nl++] = s;
    char *e = strchr(s, '\n');
    if (!e) break;
    *e = 0; s = e + 1;
  }
  orc_hmmset *hs = (orc_hmmset *)calloc(1, sizeof(*hs));
  int pcap = 64; hs->p = (orc_profile *)calloc(pcap, sizeof(orc_profile));
  char *tok[64];
  int64_t i = 0;
  orc_profile *cur = NULL;
#define FAIL(msg) do { snprintf(err, errlen, "%s (line %lld)", msg, (long long)i + 1); goto fail; } while (0)
  while (i < nl) {
    while (i < nl && strncmp(lines[i], "HMMER3", 6) != 0) {
      char *s = lines[i]; while (*s == ' ' || *s == '\t' || *s == '\r') s++;
      if (*s) FAIL("expected HMMER3 header");
      i++;
    }
    if (i >= nl) break;
    i++;
    if (hs->n == pcap) { pcap *= 2; hs->p = (orc_profile *)realloc(hs->p, sizeof(orc_profile) * pcap); memset(hs->p + hs->n, 0, sizeof(orc_profile) * (pcap - hs->n)); }
    orc_profile *p = &hs->p[hs->n];
    memset(p, 0, sizeof(*p));
    cur = p;
    int have_compo = 0, have_stats = 0;
    /* header */
    for (; i < nl; i++) {
      char *ln = lines[i];
      if (strncmp(ln, "HMM ", 4) == 0 || strcmp(ln, "HMM") == 0) break;
      if (strncmp(ln, "NAME", 4) == 0) { char *s = ln + 4; while (*s == ' ') s++; snprintf(p->name, sizeof(p->name), "%s", s); char *e = p->name + strlen(p->name); while (e > p->name && (e[-1] == ' ' || e[-1] == '\r')) *--e = 0; }
      else if (strncmp(ln, "LENG", 4) == 0) p->M = atoi(ln + 4);
      else if (strncmp(ln, "ALPH", 4) == 0) { if (!strstr(ln, "DNA") && !strstr(ln, "dna")) FAIL("only ALPH DNA is supported"); }
      else if (strncmp(ln, "STATS", 5) == 0) {
        char tmp[256]; snprintf(tmp, sizeof(tmp), "%s", ln);
        int n = split_ws(tmp, tok, 8);
        if (n >= 5) {
          int b = -1;
          if (!strcmp(tok[2], "MSV")) b = 0; else if (!strcmp(tok[2], "VITERBI")) b = 2; else if (!strcmp(tok[2], "FORWARD")) b = 4;
          if (b >= 0) { p->evparam[b] = (float)atof(tok[3]); p->evparam[b + 1] = (float)atof(tok[4]); have_stats |= 1 << (b / 2); }
        }
      }
    }
    if (i >= nl) FAIL("truncated model: no HMM line");
    if (p->M <= 0) FAIL("missing LENG");
    if (have_stats != 7) FAIL("missing STATS LOCAL lines (model not calibrated)");
    i += 2;                                       /* HMM line + transition header line */
    const int M = p->M;
    p->t = (float *)calloc((size_t)(M + 1) * 7, sizeof(float));
    p->mat = (float *)calloc((size_t)(M + 1) * 4, sizeof(float));
    if (i < nl) {
      char tmp[512]; snprintf(tmp, sizeof(tmp), "%s", lines[i]);
      int n = split_ws(tmp, tok, 16);
      if (n >= 5 && !strcmp(tok[0], "COMPO")) {
        for (int x = 0; x < 4; x++) p->compo[x] = prob_from_tok(tok[1 + x]);
        have_compo = 1; i++;
      }
    }
    if (!have_compo) FAIL("model has no COMPO line");
    i++;                                          /* node-0 insert emissions (inserts are scored 0) */
    if (i >= nl) FAIL("truncated model");
    { int n = split_ws(lines[i], tok, 16); if (n < 7) FAIL("bad node-0 transition line");
      for (int x = 0; x < 7; x++) { p->t[x] = prob_from_tok(tok[x]); } i++; }
    for (int k = 1; k <= M; k++) {
      if (i + 2 >= nl) FAIL("truncated model");
      int n = split_ws(lines[i], tok, 16);
      if (n < 5 || atoi(tok[0]) != k) FAIL("bad match emission line");
      for (int x = 0; x < 4; x++) p->mat[k * 4 + x] = prob_from_tok(tok[1 + x]);
      i++;                                        /* insert emission line: ignored */
      i++;
      n = split_ws(lines[i], tok, 16);
      if (n < 7) FAIL("bad transition line");
      for (int x = 0; x < 7; x++) p->t[k * 7 + x] = prob_from_tok(tok[x]);
      i++;
    }
    if (i >= nl || strncmp(lines[i], "//", 2) != 0) FAIL("expected // at end of model");
    i++;
    configure(p);
    hs->n++;
    cur = NULL;
  }
  free(lines); free(buf);
  return hs;
fail:
  free(lines); free(buf);
  if (cur) { free(cur->t); free(cur->mat); }
  orc_hmmset_free(hs);
  return NULL;
#undef FAIL
}

orc_hmmset *orc_hmmset_read(const char *path, char *err, int errlen)
{
  FILE *fp = fopen(path, "rb");
  if (!fp) { snprintf(err, errlen, "cannot open %s", path); return NULL; }
  fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
  char *b = (char *)malloc((size_t)sz + 1);
  if (fread(b, 1, (size_t)sz, fp) != (size_t)sz) { fclose(fp); free(b); snprintf(err, errlen, "short read"); return NULL; }
  fclose(fp);
  orc_hmmset *hs = orc_hmmset_parse(b, sz, err, errlen);
  free(b);
  return hs;
}

void orc_hmmset_free(orc_hmmset *hs)
{
  if (!hs) return;
  for (int i = 0; i < hs->n && hs->p; i++) profile_free_members(&hs->p[i]);
  free(hs->p); free(hs);
}
int orc_hmmset_count(const orc_hmmset *hs) { return hs->n; }
const char *orc_hmmset_name(const orc_hmmset *hs, int i) { return hs->p[i].name; }
int orc_hmmset_M(const orc_hmmset *hs, int i) { return hs->p[i].M; }
int orc_profile_rbv(const orc_hmmset *hs, int i, uint8_t *out)
{ const orc_profile *p = &hs->p[i]; int n = ORC_KP * (p->M + 1); memcpy(out, p->rbv, n); return n; }
int orc_profile_rfv(const orc_hmmset *hs, int i, float *out)
{ const orc_profile *p = &hs->p[i]; int n = ORC_KP * p->Q * 4; memcpy(out, p->rfv, sizeof(float) * n); return n; }
int orc_profile_tfv(const orc_hmmset *hs, int i, float *out)
{ const orc_profile *p = &hs->p[i]; int n = 8 * p->Q * 4; memcpy(out, p->tfv, sizeof(float) * n); return n; }
int orc_profile_msvparams(const orc_hmmset *hs, int i, int *out)
{ const orc_profile *p = &hs->p[i]; out[0] = p->base_b; out[1] = p->bias_b; out[2] = p->tbm_b; out[3] = p->tec_b; return 4; }
