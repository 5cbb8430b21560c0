// k_ensemble.hip -- multidomain regions: hmmsearch's stochastic-traceback clustering on the device.
//
// Replaces, for the regions the posterior region scan flags (is_multidomain_region, rt3 = 0.20), the branch of
// p7_domaindef_ByPosteriorHeuristics that calls region_trace_ensemble(): a full (multihit) Forward matrix of the region,
// 200 stochastic tracebacks over it (impl_sse/stotrace.c), a null2 score per residue from the traces (p7_Null2_ByTrace),
// single-linkage clustering of the sampled domain coordinates (p7_spensemble.c) and removal of dominated clusters.  The
// reference reads the resulting envelope coordinates (itsxpress/SeqSample.py:445-450).  The tests hold this file to their
// CPU restatement of the same procedure bit for bit; parity with a real hmmsearch is unpinned like the rest of the HMM half.
//
// Mapping.  Such regions are rare (0.4 % of all regions on the bench workload) and their 200 tracebacks are inherently
// sequential -- one random stream, and the number of draws a trace consumes depends on its path -- so one LANE owns one
// region: k_mr_fwd fills the region's Forward matrix (rows of M, D, I and the special states in a [row][vector][lane]
// slab), k_mr_trace walks the 200 paths, accumulates the per-residue null2 odds, dedupes the sampled (i, j, k, m) tuples,
// clusters them and writes the surviving envelopes.  Lanes of a wave work on different profiles, so the transition and
// emission tables come through ordinary vector loads here (the wave-uniform scalar-operand pipeline of k_float.hip does
// not apply); these kernels are latency-bound integer/float bookkeeping, not throughput kernels.
#include "engine.h"
#include "k_api.h"
#include "detmath.h"
#include "k_vec.h"

namespace itsx {

enum { tBM = 0, tMM, tIM, tDM, tMD, tMI, tII, tDD };
enum { ST_M = 1, ST_D, ST_I, ST_S, ST_N, ST_B, ST_E, ST_C, ST_T, ST_J };

DEV f4 *mrslab(const MrArgs &a, int64_t r0, int row, int v, int lane) { return (f4 *)a.slab + (((r0 + row) * MRV + v) * 64 + lane); }
DEV float comp4(const f4 &t, int r) { return r == 0 ? t.x : r == 1 ? t.y : r == 2 ? t.z : t.w; }
// component r of the vector right-shifted by one lane ([0 a b c])
DEV float comp4_rsh(const f4 &t, int r) { return r == 0 ? 0.0f : r == 1 ? t.x : r == 2 ? t.y : t.z; }
DEV V4 tov4(const f4 &t) { V4 r; r.a = (f2){t.x, t.y}; r.b = (f2){t.z, t.w}; return r; }
DEV f4 tof4(const V4 &v) { return (f4){v.a.x, v.a.y, v.b.x, v.b.y}; }

struct MrLane {
  bool active; int64_t mi; MrRec m; int L, Lr, off, Q; const DevProfile *pp; Seq sq; float pmove, ploop; int64_t r0;
};
DEV MrLane mr_lane(const MrArgs &a, const WaveDesc &wd, int lane)
{
  MrLane e;
  e.active = lane < wd.count;
  e.mi = wd.first + (e.active ? lane : 0);
  e.m = a.mr[e.mi];
  const PairRec pr = a.pairs[e.m.pair];
  e.L = pr.L; e.Lr = e.m.jreg - e.m.ireg + 1; e.off = e.m.ireg - 1;
  e.pp = a.prof + pr.prof; e.Q = e.pp->Q;
  e.sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  e.pmove = (2.0f + 1.0f) / ((float)e.L + 2.0f + 1.0f);       // multihit, length model of the whole target
  e.ploop = 1.0f - e.pmove;
  e.r0 = wd.slab;
  return e;
}

// =========================================================================================
// the region's Forward matrix: p7_Forward(dsq + i - 1, j - i + 1) in multihit mode, every row kept
__global__ void __launch_bounds__(64) k_mr_fwd(MrArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const MrLane e = mr_lane(a, wd, lane);
  const int Q = e.Q, Lw = wd.rows - 1;
  const float *tf = e.pp->tf;
  V4 M[QMAX], D[QMAX], I[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; q++) { M[q] = vzero(); D[q] = vzero(); I[q] = vzero(); }
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = e.pmove, xC = 0.f;
  if (e.active) {
    for (int q = 0; q < Q; q++) for (int s = 0; s < 3; s++) *mrslab(a, e.r0, 0, q * 3 + s, lane) = (f4){0.f, 0.f, 0.f, 0.f};
    *mrslab(a, e.r0, 0, 36, lane) = (f4){xE, xN, xJ, xB};
    *mrslab(a, e.r0, 0, 37, lane) = (f4){xC, 1.0f, 0.f, 0.f};
  }
  for (int i = 1; i <= Lw; i++) {
    if (!(e.active && i <= e.Lr)) continue;
    const int x = e.sq.code(e.off + i - 1);
    const float *rfx = e.pp->rf + x * QMAX * 4;
#define T(q, t) vld(tf + ((q) * 8 + (t)) * 4)
    V4 dcv = vzero(), xEv = vzero();
    const V4 xBv = vset(xB);
    V4 mpv = vzero(), dpv = vzero(), ipv = vzero(), sv;
#pragma unroll
    for (int q = 0; q < QMAX; q++) if (q == Q - 1) { mpv = vrsh(M[q]); dpv = vrsh(D[q]); ipv = vrsh(I[q]); }
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      if (q >= Q) break;
      sv = vmul(xBv, T(q, tBM));
      sv = vadd(sv, vmul(mpv, T(q, tMM)));
      sv = vadd(sv, vmul(ipv, T(q, tIM)));
      sv = vadd(sv, vmul(dpv, T(q, tDM)));
      sv = vmul(sv, vld(rfx + q * 4));
      xEv = vadd(xEv, sv);
      mpv = M[q]; dpv = D[q]; ipv = I[q];
      M[q] = sv; D[q] = dcv;
      dcv = vmul(sv, T(q, tMD));
      sv = vmul(mpv, T(q, tMI));
      I[q] = vadd(sv, vmul(ipv, T(q, tII)));
    }
    dcv = vrsh(dcv);
    D[0] = vzero();
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      if (q >= Q) break;
      D[q] = vadd(dcv, D[q]);
      dcv = vmul(D[q], T(q, tDD));
    }
    for (int j = 1; j < 4; j++) {
      dcv = vrsh(dcv);
#pragma unroll
      for (int q = 0; q < QMAX; q++) {
        if (q >= Q) break;
        D[q] = vadd(dcv, D[q]);
        dcv = vmul(dcv, T(q, tDD));
      }
    }
#pragma unroll
    for (int q = 0; q < QMAX; q++) { if (q >= Q) break; xEv = vadd(D[q], xEv); }
#undef T
    xE = vhsum(xEv);
    xN = xN * e.ploop;
    xC = (xC * e.ploop) + (xE * 0.5f);
    xJ = (xJ * e.ploop) + (xE * 0.5f);
    xB = (xJ * e.pmove) + (xN * e.pmove);
    float sc = 1.0f;
    if (xE > 1.0e4f) {
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const V4 s = vset((float)(1.0 / (double)xE));
#pragma unroll
      for (int q = 0; q < QMAX; q++) { if (q >= Q) break; M[q] = vmul(M[q], s); D[q] = vmul(D[q], s); I[q] = vmul(I[q], s); }
      sc = xE;
      xE = 1.0f;
    }
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      if (q >= Q) break;
      *mrslab(a, e.r0, i, q * 3 + 0, lane) = tof4(M[q]);
      *mrslab(a, e.r0, i, q * 3 + 1, lane) = tof4(D[q]);
      *mrslab(a, e.r0, i, q * 3 + 2, lane) = tof4(I[q]);
    }
    *mrslab(a, e.r0, i, 36, lane) = (f4){xE, xN, xJ, xB};
    *mrslab(a, e.r0, i, 37, lane) = (f4){xC, sc, 0.f, 0.f};
  }
}

// =========================================================================================
// Knuth's linear congruential generator as esl_randomness_CreateFast() runs it
DEV uint32_t rnd_mix3(uint32_t a, uint32_t b, uint32_t c)
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
DEV double rng_next(uint32_t &x) { x *= 69069u; x += 1u; return (double)x / 4294967296.0; }
// esl_vec_FNorm + esl_rnd_FChoose over two or four path weights
DEV int choose2(uint32_t &rng, float p0, float p1)
{
  float sum = 0.f; sum += p0; sum += p1;
  if (sum != 0.0f) { p0 /= sum; p1 /= sum; } else { p0 = p1 = (float)(1. / (double)2.0f); }
  const double roll = rng_next(rng);
  double s = 0.0;
  s += p0; if (roll < s) return 0;
  s += p1; if (roll < s) return 1;
  int i;
  do { i = (int)(rng_next(rng) * 2); } while ((i == 0 ? p0 : p1) == 0.f);
  return i;
}
DEV int choose4(uint32_t &rng, float p0, float p1, float p2, float p3)
{
  float sum = 0.f; sum += p0; sum += p1; sum += p2; sum += p3;
  if (sum != 0.0f) { p0 /= sum; p1 /= sum; p2 /= sum; p3 /= sum; } else { p0 = p1 = p2 = p3 = (float)(1. / (double)4.0f); }
  const double roll = rng_next(rng);
  double s = 0.0;
  s += p0; if (roll < s) return 0;
  s += p1; if (roll < s) return 1;
  s += p2; if (roll < s) return 2;
  s += p3; if (roll < s) return 3;
  int i;
  do { i = (int)(rng_next(rng) * 4); } while ((i == 0 ? p0 : i == 1 ? p1 : i == 2 ? p2 : p3) == 0.f);
  return i;
}

struct Tup { int i, j, k, m; };
DEV Tup unpack_tup(unsigned long long v) { Tup t; t.i = (int)(v & 0xffff); t.j = (int)((v >> 16) & 0xffff); t.k = (int)((v >> 32) & 0xff); t.m = (int)((v >> 40) & 0xff); return t; }
DEV unsigned long long pack_tup(int i, int j, int k, int m) { return (unsigned long long)i | ((unsigned long long)j << 16) | ((unsigned long long)k << 32) | ((unsigned long long)m << 40); }
// p7_spensemble.c: link_spsamples with min_overlap 0.8 of the smaller segment, max_diagdiff 4
DEV bool link_tup(const Tup &h1, const Tup &h2)
{
  int nov = min(h1.j, h2.j) - max(h1.i, h2.i) + 1;
  int n = min(h1.j - h1.i + 1, h2.j - h2.i + 1);
  if ((float)nov / (float)n < 0.8f) return false;
  nov = min(h1.m, h2.m) - max(h1.k, h2.k);                  // as published: no "+ 1" on the model side
  n = min(h1.m - h1.k + 1, h2.m - h2.k + 1);
  if ((float)nov / (float)n < 0.8f) return false;
  int d1 = h1.i - h1.k, d2 = h2.i - h2.k;
  if (abs(d1 - d2) <= 4) return true;
  d1 = h1.j - h1.m; d2 = h2.j - h2.m;
  return abs(d1 - d2) <= 4;
}

// per-lane scratch block (MR_SCRATCH bytes): the sampled tuples and the clustering's bookkeeping
struct MrScratch {
  unsigned long long key[MR_TCAP];
  uint16_t tcount[MR_TCAP], comp[MR_TCAP], stack[MR_TCAP], ninc[MR_TCAP]; int16_t last[MR_TCAP];
  uint16_t tid[MR_SCAP]; uint8_t tidx[MR_SCAP];
  int32_t sig_i[MR_NSIG], sig_j[MR_NSIG]; float sig_p[MR_NSIG]; uint8_t dominated[MR_NSIG];
};
static_assert(sizeof(MrScratch) <= MR_SCRATCH, "scratch block too small");

__global__ void __launch_bounds__(64) k_mr_trace(MrArgs a, int wave0)
{
  __shared__ float cnt_s[2][QMAX * 4][64];                  // match / insert usage of the domain being walked
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const MrLane e = mr_lane(a, wd, lane);
  if (!e.active) return;
  const int Q = e.Q, Lr = e.Lr;
  const float *tf = e.pp->tf;
  const float pmove = e.pmove, ploop = e.ploop;
  float *n2 = a.n2sc + a.n2off[e.mi];                       // n2[pos - 1], pos = 1..Lr relative to the region
  MrScratch &S = *(MrScratch *)(a.scratch + (int64_t)(e.mi - a.mr0) * MR_SCRATCH);
  for (int pos = 0; pos < Lr; pos++) n2[pos] = 0.0f;
  uint32_t rng = rnd_mix3(42u, 87654321u, 12345678u);
  if (rng == 0) rng = 42;
  int ntup = 0, nsamp = 0, status = 0;
  auto MV = [&](int row, int q) { return *mrslab(a, e.r0, row, q * 3 + 0, lane); };
  auto DV = [&](int row, int q) { return *mrslab(a, e.r0, row, q * 3 + 1, lane); };
  auto IV = [&](int row, int q) { return *mrslab(a, e.r0, row, q * 3 + 2, lane); };
  auto TFC = [&](int q, int t, int r) { return tf[((q) * 8 + (t)) * 4 + r]; };
  const unsigned short dgm[16] = {1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15};

  for (int t = 0; t < 200 && status == 0; t++) {
    int i = Lr, k = 0, s0 = ST_C, s1 = 0, nd = 0, hi = Lr;
    int dfrom = 0, dto = 0, dk = 0, dm = 0;
    while (s0 != ST_S) {
      switch (s0) {
      case ST_M: {
        const int q = (k - 1) % Q, r = (k - 1) / Q;
        float mp, dp, ip;
        if (q > 0) { mp = comp4(MV(i - 1, q - 1), r); dp = comp4(DV(i - 1, q - 1), r); ip = comp4(IV(i - 1, q - 1), r); }
        else { mp = comp4_rsh(MV(i - 1, Q - 1), r); dp = comp4_rsh(DV(i - 1, Q - 1), r); ip = comp4_rsh(IV(i - 1, Q - 1), r); }
        const float xB = mrslab(a, e.r0, i - 1, 36, lane)->w;
        const int c = choose4(rng, xB * TFC(q, tBM, r), mp * TFC(q, tMM, r), ip * TFC(q, tIM, r), dp * TFC(q, tDM, r));
        s1 = c == 0 ? ST_B : c == 1 ? ST_M : c == 2 ? ST_I : ST_D;
        k--; i--;
        break; }
      case ST_D: {
        const int q = (k - 1) % Q, r = (k - 1) / Q;
        float mp, dp, tmd, tdd;
        if (q > 0) { mp = comp4(MV(i, q - 1), r); dp = comp4(DV(i, q - 1), r); tmd = TFC(q - 1, tMD, r); tdd = TFC(q - 1, tDD, r); }
        else {
          mp = comp4_rsh(MV(i, Q - 1), r); dp = comp4_rsh(DV(i, Q - 1), r);
          tmd = r == 0 ? 0.0f : TFC(Q - 1, tMD, r - 1); tdd = r == 0 ? 0.0f : TFC(Q - 1, tDD, r - 1);
        }
        s1 = choose2(rng, mp * tmd, dp * tdd) == 0 ? ST_M : ST_D;
        k--;
        break; }
      case ST_I: {
        const int q = (k - 1) % Q, r = (k - 1) / Q;
        s1 = choose2(rng, comp4(MV(i - 1, q), r) * TFC(q, tMI, r), comp4(IV(i - 1, q), r) * TFC(q, tII, r)) == 0 ? ST_M : ST_I;
        i--;
        break; }
      case ST_N: s1 = (i == 0) ? ST_S : ST_N; break;
      case ST_C: {
        if (i < 1) { status = 5; s1 = ST_S; break; }
        const float cprev = mrslab(a, e.r0, i - 1, 37, lane)->x;
        const f4 x0 = *mrslab(a, e.r0, i, 36, lane); const float scl = mrslab(a, e.r0, i, 37, lane)->y;
        s1 = choose2(rng, cprev * ploop, x0.x * 0.5f * scl) == 0 ? ST_C : ST_E;
        break; }
      case ST_J: {
        if (i < 1) { status = 5; s1 = ST_S; break; }
        const float jprev = mrslab(a, e.r0, i - 1, 36, lane)->z;
        const f4 x0 = *mrslab(a, e.r0, i, 36, lane); const float scl = mrslab(a, e.r0, i, 37, lane)->y;
        s1 = choose2(rng, jprev * ploop, x0.x * 0.5f * scl) == 0 ? ST_J : ST_E;
        break; }
      case ST_E: {
        double sum = 0.0;
        const double roll = rng_next(rng);
        const double norm = 1.0 / (double)mrslab(a, e.r0, i, 36, lane)->x;
        const float xEv = (float)norm;
        s1 = -1;
        while (s1 < 0) {
          for (int q = 0; q < Q && s1 < 0; q++) {
            f4 u = MV(i, q);
            for (int r = 0; r < 4 && s1 < 0; r++) { sum += (double)(comp4(u, r) * xEv); if (roll < sum) { k = r * Q + q + 1; s1 = ST_M; } }
            if (s1 >= 0) break;
            u = DV(i, q);
            for (int r = 0; r < 4 && s1 < 0; r++) { sum += (double)(comp4(u, r) * xEv); if (roll < sum) { k = r * Q + q + 1; s1 = ST_D; } }
          }
          if (s1 < 0 && sum < 0.99) { status = 1; s1 = ST_S; }          // HMMER throws here
        }
        if (status) break;
        if (nd >= MR_MAXD) { status = 2; s1 = ST_S; break; }
        nd++;
        dfrom = dto = dk = dm = 0;
        for (int z = 0; z < Q * 4; z++) { cnt_s[0][z][lane] = 0.f; cnt_s[1][z][lane] = 0.f; }
        break; }
      case ST_B: {
        const f4 x0 = *mrslab(a, e.r0, i, 36, lane);
        s1 = choose2(rng, x0.y * pmove, x0.z * pmove) == 0 ? ST_N : ST_J;
        // the domain that was being walked is complete: its sample, its null2 odds, its residues
        if (nsamp >= MR_SCAP) { status = 3; s1 = ST_S; break; }
        const unsigned long long key = pack_tup(dfrom + e.m.ireg - 1, dto + e.m.ireg - 1, dk, dm);
        int tix = 0;
        while (tix < ntup && S.key[tix] != key) tix++;
        if (tix == ntup) {
          if (ntup >= MR_TCAP) { status = 4; s1 = ST_S; break; }
          S.key[ntup] = key; S.tcount[ntup] = 0; ntup++;
        }
        S.tcount[tix]++;
        S.tid[nsamp] = (uint16_t)tix; S.tidx[nsamp] = (uint8_t)t; nsamp++;
        int Ld = 0;
        for (int z = 0; z < Q * 4; z++) Ld += (int)cnt_s[0][z][lane] + (int)cnt_s[1][z][lane];
        const float norm = (float)(1.0 / (double)(float)Ld);
        const float xfactor = (0.0f * norm + 0.0f * norm) + 0.0f * norm;
        float null2[NCODE];
        for (int x = 0; x < 4; x++) {
          V4 sv = vzero();
          const float *rp = e.pp->rf + x * QMAX * 4;
          for (int q = 0; q < Q; q++) {
            V4 mv, iv;
            mv.a = (f2){cnt_s[0][q * 4 + 0][lane], cnt_s[0][q * 4 + 1][lane]}; mv.b = (f2){cnt_s[0][q * 4 + 2][lane], cnt_s[0][q * 4 + 3][lane]};
            iv.a = (f2){cnt_s[1][q * 4 + 0][lane], cnt_s[1][q * 4 + 1][lane]}; iv.b = (f2){cnt_s[1][q * 4 + 2][lane], cnt_s[1][q * 4 + 3][lane]};
            mv = vmul(mv, vset(norm)); iv = vmul(iv, vset(norm));
            sv = vadd(sv, vmul(mv, vld(rp + q * 4)));
            sv = vadd(sv, iv);
          }
          null2[x] = vhsum(sv);
          null2[x] += xfactor;
        }
        null2[4] = 1.0f;
        for (int x = 5; x < 16; x++) {
          float acc = 0.f; int ndg = 0;
          for (int y = 0; y < 4; y++) if (dgm[x] >> y & 1) { acc += null2[y]; ndg++; }
          null2[x] = acc / (float)ndg;
        }
        // as published: residues up to AND INCLUDING the domain's first one count as outside (+1), the rest of it by null2
        for (int pos = hi; pos > dto; pos--) n2[pos - 1] += 1.0f;
        for (int pos = dto; pos > dfrom; pos--) {
          const int x = e.sq.code(e.off + pos - 1);
          float v = null2[0];
#pragma unroll
          for (int c = 1; c < NCODE; c++) v = (x == c) ? null2[c] : v;
          n2[pos - 1] += v;
        }
        hi = dfrom;
        break; }
      default: status = 5; s1 = ST_S; break;
      }
      if (status) break;
      if (s1 == ST_M) {
        if (dto == 0) { dto = i; dm = k; }
        dfrom = i; dk = k;
        cnt_s[0][((k - 1) % Q) * 4 + (k - 1) / Q][lane] += 1.0f;
      } else if (s1 == ST_I) {
        cnt_s[1][((k - 1) % Q) * 4 + (k - 1) / Q][lane] += 1.0f;
      }
      if ((s1 == ST_N || s1 == ST_J || s1 == ST_C) && s1 == s0) i--;
      s0 = s1;
    }
    if (status) break;
    for (int pos = hi; pos >= 1; pos--) n2[pos - 1] += 1.0f;
  }

  MrOut out;
  out.status = status; out.nenv = 0;
#pragma unroll
  for (int z = 0; z < MRENV; z++) { out.ei[z] = 0; out.ej[z] = 0; }
  if (status != 0) {
    for (int pos = 0; pos < Lr; pos++) n2[pos] = 0.0f;
    a.out[e.mi] = out;
    return;
  }
  for (int pos = 0; pos < Lr; pos++) n2[pos] = det_logf(n2[pos] / (float)200);

  // ---- single-linkage clustering over the DISTINCT tuples.  Copies of one tuple always share their neighbours, so they
  // fall into one component -- except an isolated tuple that does not link to itself (model span under 5 nodes): there
  // every copy is a cluster of its own with posterior 1/200, which never reaches 0.25.
  for (int h = 0; h < ntup; h++) S.comp[h] = 0xffff;
  int nc = 0;
  for (int h0 = 0; h0 < ntup; h0++) {
    if (S.comp[h0] != 0xffff) continue;
    int ns = 0; S.stack[ns++] = (uint16_t)h0; S.comp[h0] = (uint16_t)nc;
    int members = 0;
    while (ns > 0) {
      const int v = S.stack[--ns];
      members++;
      const Tup tv = unpack_tup(S.key[v]);
      for (int u = 0; u < ntup; u++)
        if (S.comp[u] == 0xffff && link_tup(tv, unpack_tup(S.key[u]))) { S.comp[u] = (uint16_t)nc; S.stack[ns++] = (uint16_t)u; }
    }
    const Tup t0 = unpack_tup(S.key[h0]);
    S.ninc[nc] = (members == 1 && !link_tup(t0, t0)) ? 0xffff : 0;      // 0xffff: a set of singletons, never reported
    S.last[nc] = -1;
    nc++;
  }
  // posterior of each cluster: traces with at least one member (samples are in trace order)
  for (int h = 0; h < nsamp; h++) {
    const int c = S.comp[S.tid[h]];
    if (S.ninc[c] == 0xffff) continue;
    if ((int)S.tidx[h] != (int)S.last[c]) S.ninc[c]++;
    S.last[c] = (int16_t)S.tidx[h];
  }
  int nsig = 0;
  for (int c = 0; c < nc; c++) {
    if (S.ninc[c] == 0xffff) continue;
    const int ninc = S.ninc[c];
    if ((float)ninc / (float)200 < 0.25f) continue;
    const int thr = (int)ceilf((float)ninc * 0.02f);
    // endpoint histograms without the arrays: weight of value v = copies of the cluster's tuples that carry it
    int best[4];
    for (int f = 0; f < 4; f++) {
      const bool leftmost = f < 2;                    // i and k: leftmost value with enough endpoints; j and m: rightmost
      int pick = -1, am = -1, amw = -1;
      for (int h = 0; h < ntup; h++) {
        if (S.comp[h] != c) continue;
        const Tup th = unpack_tup(S.key[h]);
        const int v = f == 0 ? th.i : f == 1 ? th.k : f == 2 ? th.j : th.m;
        int w = 0;
        for (int u = 0; u < ntup; u++) {
          if (S.comp[u] != c) continue;
          const Tup tu = unpack_tup(S.key[u]);
          const int vu = f == 0 ? tu.i : f == 1 ? tu.k : f == 2 ? tu.j : tu.m;
          if (vu == v) w += S.tcount[u];
        }
        if (w >= thr && (pick < 0 || (leftmost ? v < pick : v > pick))) pick = v;
        if (w > amw || (w == amw && v < am)) { amw = w; am = v; }        // esl_vec_IArgMax: the first (smallest) maximum
      }
      best[f] = pick >= 0 ? pick : am;
    }
    if (best[0] > best[2] || best[1] > best[3]) continue;
    if (nsig >= MR_NSIG) { status = 6; break; }
    S.sig_i[nsig] = best[0]; S.sig_j[nsig] = best[2]; S.sig_p[nsig] = (float)ninc / (float)200;
    nsig++;
  }
  if (status == 0) {
    // order by start (stable insertion sort), then drop dominated clusters
    for (int x = 1; x < nsig; x++) {
      const int ti = S.sig_i[x], tj = S.sig_j[x]; const float tp = S.sig_p[x];
      int y = x - 1;
      while (y >= 0 && S.sig_i[y] > ti) { S.sig_i[y + 1] = S.sig_i[y]; S.sig_j[y + 1] = S.sig_j[y]; S.sig_p[y + 1] = S.sig_p[y]; y--; }
      S.sig_i[y + 1] = ti; S.sig_j[y + 1] = tj; S.sig_p[y + 1] = tp;
    }
    for (int d = 0; d < nsig; d++) S.dominated[d] = 0;
    for (int d = 0; d < nsig; d++)
      for (int d2 = d + 1; d2 < nsig; d2++) {
        const int nov = min(S.sig_j[d], S.sig_j[d2]) - max(S.sig_i[d], S.sig_i[d2]) + 1;
        if (nov == 0) break;
        const int n = min(S.sig_j[d] - S.sig_i[d] + 1, S.sig_j[d2] - S.sig_i[d2] + 1);
        if ((float)nov / (float)n >= 0.8f) { if (S.sig_p[d] > S.sig_p[d2]) S.dominated[d2] = 1; else S.dominated[d] = 1; }
      }
    for (int d = 0; d < nsig; d++) {
      if (S.dominated[d]) continue;
      if (out.nenv >= MRENV) { status = 7; break; }
      out.ei[out.nenv] = S.sig_i[d]; out.ej[out.nenv] = S.sig_j[d]; out.nenv++;
    }
  }
  if (status != 0) {              // a bookkeeping limit was hit: the region yields nothing, and says so
    out.nenv = 0;
    for (int pos = 0; pos < Lr; pos++) n2[pos] = 0.0f;
  }
  out.status = status;
  a.out[e.mi] = out;
}

// =========================================================================================
// list building: multidomain regions of every pair, in pair order
__global__ void __launch_bounds__(256) k_mr_count(const PairOut *__restrict__ pout, const RegionRec *__restrict__ raw, int64_t npairs,
                                                  int32_t *__restrict__ cnt, int32_t *__restrict__ len)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi > npairs) return;
  int c = 0, l = 0;
  if (pi < npairs && pout[pi].pass_fwd)
    for (int k = 0; k < pout[pi].ndom; k++) { const RegionRec r = raw[pi * MAXDOM + k]; if (r.multi) { c++; l += r.jenv - r.ienv + 1; } }
  cnt[pi] = c; len[pi] = l;
}
__global__ void __launch_bounds__(256) k_mr_fill(const PairOut *__restrict__ pout, const RegionRec *__restrict__ raw, int64_t npairs,
                                                 const int32_t *__restrict__ off, const int32_t *__restrict__ loff, MrRec *__restrict__ mr,
                                                 int64_t *__restrict__ n2off)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs || !pout[pi].pass_fwd) return;
  int o = off[pi]; int64_t lo = loff[pi];
  for (int k = 0; k < pout[pi].ndom; k++) {
    const RegionRec r = raw[pi * MAXDOM + k];
    if (!r.multi) continue;
    MrRec m; m.pair = (int32_t)pi; m.slot = k; m.ireg = r.ienv; m.jreg = r.jenv;
    mr[o] = m; n2off[o] = lo;
    o++; lo += r.jenv - r.ienv + 1;
  }
}
__global__ void k_mr_wave_rows(const WaveDesc *w, int nw, const MrRec *mr, int32_t *rows)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nw) return;
  int mx = 0;
  for (int k = 0; k < w[i].count; k++) { const MrRec r = mr[w[i].first + k]; mx = max(mx, r.jreg - r.ireg + 1); }
  rows[i] = mx + 1;
}
// every pair with multidomain regions gets its envelope list rebuilt: a clustered region is replaced by its envelopes
__global__ void __launch_bounds__(256) k_mr_apply(PairOut *__restrict__ pout, RegionRec *__restrict__ raw, int64_t npairs,
                                                  const int32_t *__restrict__ off, const MrOut *__restrict__ out, unsigned long long *__restrict__ counters)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs || !pout[pi].pass_fwd) return;
  const int nm = off[pi + 1] - off[pi];
  if (nm == 0) return;
  RegionRec tmp[MAXDOM];
  int n = 0, over = 0, mk = off[pi];
  unsigned long long nfail = 0, nenv = 0;
  for (int k = 0; k < pout[pi].ndom; k++) {
    const RegionRec r = raw[pi * MAXDOM + k];
    if (!r.multi) { if (n < MAXDOM) tmp[n++] = r; else over = 1; continue; }
    const MrOut o = out[mk];
    nfail += o.status != 0; nenv += o.nenv;
    for (int z = 0; z < o.nenv; z++) {
      RegionRec c; c.pair = (int32_t)pi; c.ienv = o.ei[z]; c.jenv = o.ej[z]; c.multi = mk + 1;
      if (n < MAXDOM) tmp[n++] = c; else over = 1;
    }
    mk++;
  }
  for (int k = 0; k < n; k++) raw[pi * MAXDOM + k] = tmp[k];
  pout[pi].ndom = n;
  if (over) pout[pi].flags |= 2;
  if (nfail) atomicAdd(&counters[0], nfail);
  if (nenv) atomicAdd(&counters[1], nenv);
}

void launch_mr_count(const PairOut *pout, const RegionRec *raw, int64_t npairs, int32_t *cnt, int32_t *len, hipStream_t st)
{
  hipLaunchKernelGGL(k_mr_count, dim3((unsigned)((npairs + 1 + 255) / 256)), dim3(256), 0, st, pout, raw, npairs, cnt, len);
}
void launch_mr_fill(const PairOut *pout, const RegionRec *raw, int64_t npairs, const int32_t *off, const int32_t *loff, MrRec *mr, int64_t *n2off, hipStream_t st)
{
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_mr_fill, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, pout, raw, npairs, off, loff, mr, n2off);
}
void launch_mr_wave_rows(const WaveDesc *w, int nw, const MrRec *mr, int32_t *rows, hipStream_t st)
{
  if (nw <= 0) return;
  hipLaunchKernelGGL(k_mr_wave_rows, dim3((nw + 255) / 256), dim3(256), 0, st, w, nw, mr, rows);
}
void launch_mr_ensemble(const MrArgs &a, int nwaves, int wave0, hipStream_t st)
{
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_mr_fwd, dim3(nwaves), dim3(64), 0, st, a, wave0);
  hipLaunchKernelGGL(k_mr_trace, dim3(nwaves), dim3(64), 0, st, a, wave0);
}
void launch_mr_apply(PairOut *pout, RegionRec *raw, int64_t npairs, const int32_t *off, const MrOut *out, unsigned long long *counters, hipStream_t st)
{
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_mr_apply, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, pout, raw, npairs, off, out, counters);
}

}  // namespace itsx
