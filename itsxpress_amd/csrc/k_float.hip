// k_float.hip -- stages C-F of the path: bias filter, Forward, Backward, posterior domain
// definition, per-envelope re-scoring with null2, and bit scores.
//
// Replaces the float half of hmmsearch (reference call site itsxpress/SeqSample.py:191-209):
// p7_bg_FilterScore, p7_ForwardParser, p7_BackwardParser, p7_domaindef_ByPosteriorHeuristics,
// rescore_isolated_domain + p7_Null2_ByExpectation, and the score bookkeeping of p7_Pipeline.
//
// CDNA4 mapping.  These are serial recurrences over (row, node); the float results must not
// depend on how the work is scheduled, because the consumer compares scores after %.1f
// rounding (ItsPosition._score, itsxpress/SeqSample.py:400-429).  So:
//   * one LANE owns one (representative, profile) comparison and evaluates HMMER's own
//     4-lane x Q striped recurrence literally, in registers (3 x 4Q floats of DP row), in the
//     same operation order (separate mul/add, no FMA: built with -ffp-contract=off; f32
//     denormals flushed like HMMER's FTZ/DAZ: -fgpu-flush-denormals-to-zero).  No cross-lane
//     traffic, no wavefront shuffles: a shuffle-carried scan would change the association
//     order of the D->D path and of the E-state sum.
//   * one WAVE holds 64 comparisons against the SAME profile (work lists are grouped by
//     profile, ascending length), so transition odds are wave-uniform and stream through
//     scalar loads into SGPR operands; match-emission odds (16 codes x 4Q floats, 3 KB) sit
//     in LDS and are read with one ds_read_b128 per vector, conflict-free across A/C/G/T.
//   * packed f32 math (float2 -> v_pk_mul_f32 / v_pk_add_f32) carries two of the four
//     emulated SSE lanes per instruction.
//   * per-row special-state values go to an HBM slab laid out [row][field][lane] so that
//     every store/load is one coalesced 256-byte line per wave.
// No MFMA: nothing here is a contraction.
#include "engine.h"
#include "k_api.h"
#include "detmath.h"
#include "k_vec.h"

namespace itsx {

// transition vector `idx` of the wave's profile: [q*8 + t] -> 4 floats (uniform address)
#define TF(q, t) vldc(tf + ((q) * 8 + (t)) * 4)
enum { tBM = 0, tMM, tIM, tDM, tMD, tMI, tII, tDD };

struct Specials { float E, N, J, B, C, SCALE; };

template <int QT> struct Row { V4 m[QMAX], d[QMAX], i[QMAX]; };

// ---- transition operands ------------------------------------------------------------------
// The 96 transition vectors of a profile are wave-uniform.  They are fetched with scalar loads and
// used as SGPR operands of the packed multiplies.  Left alone, the scheduler hoists all of a row's
// loads to the top of the row and spills hundreds of SGPRs to VGPR lanes; so the row loops are
// software-pipelined by hand: the vectors of node-group q+1 are requested while group q is being
// computed (two buffers of 28 SGPRs), and a sched_barrier at the end of every group keeps the
// compiler from undoing that.
struct T7 { V4 bm, mm, im, dm, md, mi, ii; };
DEV T7 ld7(const float *tf, int q)
{
  T7 t;
  t.bm = TF(q, tBM); t.mm = TF(q, tMM); t.im = TF(q, tIM); t.dm = TF(q, tDM);
  t.md = TF(q, tMD); t.mi = TF(q, tMI); t.ii = TF(q, tII);
  return t;
}
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// wait for everything requested during the previous node-group (scalar operands + the LDS emission vector),
// BEFORE the next group's requests are issued: otherwise the wait at the first use of the current operands
// (scalar loads return out of order, so it can only be lgkmcnt(0)) would also wait for the prefetch
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// ---- one Forward row: HMMER's striped forward_engine inner body, literally --------------
template <int QT>
DEV void fwd_row(Row<QT> &R, const int Q, const float *__restrict__ tf, const float *rfx /* LDS: &rf[x][0][0] */,
                 float &xN, float &xB, float &xJ, float &xC, float &xE,
                 const float pmove, const float ploop, const float eloop, const float emove)
{
  tf += opaque_zero();
  V4 dcv = vzero(), xEv = vzero();
  const V4 xBv = vset(xB);
  V4 mpv = vrsh(R.m[Q - 1]), dpv = vrsh(R.d[Q - 1]), ipv = vrsh(R.i[Q - 1]);
  V4 sv;
  if constexpr (QT != 0) {
    // The node groups are independent of one another within a row (group q reads the OLD row's group q-1), so they
    // are evaluated from the last group down: every new value then overwrites an old one that nobody needs any more
    // and the row state stays in the same registers from row to row (evaluated upwards, the new M/I vectors are born
    // while the old ones are still live, and the loop back-edge pays ~47 register copies per row).  Every value is
    // computed by the same operations on the same operands; xE's sum over the groups is taken afterwards, ascending.
    T7 cur = ld7(tf, QT - 1);
    V4 ecur = vld(rfx + (QT - 1) * 4);
#pragma unroll
    for (int q = QT - 1; q >= 0; q--) {
      WAIT_LGKM0();
      T7 nxt = cur;
      V4 enxt = ecur;
      if (q > 0) { nxt = ld7(tf, q - 1); enxt = vld(rfx + (q - 1) * 4); }   // next group's operands fly during this group
      const V4 pm = q > 0 ? R.m[q - 1] : mpv, pi = q > 0 ? R.i[q - 1] : ipv, pd = q > 0 ? R.d[q - 1] : dpv;
      // the insert state first: it is the last reader of the old M[q] and I[q], whose registers the new values then take
      R.i[q] = vadd(vmul(R.m[q], cur.mi), vmul(R.i[q], cur.ii));
      asm volatile("" : "+v"(R.i[q].a), "+v"(R.i[q].b));
      sv = vmul(xBv, cur.bm);
      sv = vadd(sv, vmul(pm, cur.mm));
      sv = vadd(sv, vmul(pi, cur.im));
      sv = vadd(sv, vmul(pd, cur.dm));
      sv = vmul(sv, ecur);
      R.m[q] = sv;
      if (q + 1 < QT) R.d[q + 1] = vmul(sv, cur.md); else dcv = vmul(sv, cur.md);
      asm volatile("" : "+v"(R.m[q].a), "+v"(R.m[q].b));
      SCHED_FENCE();
      cur = nxt; ecur = enxt;
    }
#pragma unroll
    for (int q = 0; q < QT; q++) xEv = vadd(xEv, R.m[q]);
    V4 dd[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) dd[q] = TF(q, tDD);
    dcv = vrsh(dcv);
    R.d[0] = vzero();
#pragma unroll
    for (int q = 0; q < QT; q++) {
      R.d[q] = vadd(dcv, R.d[q]);
      dcv = vmul(R.d[q], dd[q]);
    }
    // passes 2-4 only extend D->D across lanes: after j shifts the j lowest lanes of dcv are +0 and stay +0,
    // and x + (+0) == x bit-for-bit for the non-negative DP values, so passes 3 and 4 skip the low lane pair
    dcv = vrsh(dcv);
#pragma unroll
    for (int q = 0; q < QT; q++) {
      R.d[q] = vadd(dcv, R.d[q]);
      dcv = vmul(dcv, dd[q]);
    }
#pragma unroll
    for (int j = 2; j < 4; j++) {
      dcv = vrsh(dcv);
#pragma unroll
      for (int q = 0; q < QT; q++) {
        R.d[q].b = dcv.b + R.d[q].b;
        dcv.b = dcv.b * dd[q].b;
      }
    }
#pragma unroll
    for (int q = 0; q < QT; q++) xEv = vadd(R.d[q], xEv);
    SCHED_FENCE();
  } else {
    for (int q = 0; q < Q; q++) {
      sv = vmul(xBv, TF(q, tBM));
      sv = vadd(sv, vmul(mpv, TF(q, tMM)));
      sv = vadd(sv, vmul(ipv, TF(q, tIM)));
      sv = vadd(sv, vmul(dpv, TF(q, tDM)));
      sv = vmul(sv, vld(rfx + q * 4));
      xEv = vadd(xEv, sv);
      mpv = R.m[q]; dpv = R.d[q]; ipv = R.i[q];
      R.m[q] = sv; R.d[q] = dcv;
      dcv = vmul(sv, TF(q, tMD));
      sv = vmul(mpv, TF(q, tMI));
      R.i[q] = vadd(sv, vmul(ipv, TF(q, tII)));
    }
    dcv = vrsh(dcv);
    R.d[0] = vzero();
    for (int q = 0; q < Q; q++) {
      R.d[q] = vadd(dcv, R.d[q]);
      dcv = vmul(R.d[q], TF(q, tDD));
    }
    for (int j = 1; j < 4; j++) {
      dcv = vrsh(dcv);
      for (int q = 0; q < Q; q++) {
        R.d[q] = vadd(dcv, R.d[q]);
        dcv = vmul(dcv, TF(q, tDD));
      }
    }
    for (int q = 0; q < Q; q++) xEv = vadd(R.d[q], xEv);
  }
  xE = vhsum(xEv);
  xN = xN * ploop;
  xC = (xC * ploop) + (xE * emove);
  xJ = (xJ * ploop) + (xE * eloop);
  xB = (xJ * pmove) + (xN * pmove);
}

template <int QT>
DEV void scale_row(Row<QT> &R, const int Q, const float s)
{
  const V4 sc = vset((float)(1.0 / (double)s));
#pragma unroll
  for (int q = 0; q < (QT ? QT : QMAX); q++) {
    if (QT == 0 && q >= Q) break;
    R.m[q] = vmul(R.m[q], sc); R.d[q] = vmul(R.d[q], sc); R.i[q] = vmul(R.i[q], sc);
  }
}

// D->D / M->D completion shared by Backward's row L and rows L-1..1 (phases 3-5 of HMMER's engine)
template <int QT>
DEV void bwd_dd_md(Row<QT> &R, const int Q, const float *__restrict__ tf, const V4 xEv, const bool rowL)
{
  tf += opaque_zero();
  V4 dpv, dcv = vzero();
  if (rowL) dpv = vlsh(R.d[Q - 1]);
  else      dpv = vlsh(vadd(R.d[0], xEv));
  if constexpr (QT != 0) {
    {
      V4 dd[QT];
#pragma unroll
      for (int q = 0; q < QT; q++) dd[q] = TF(q, tDD);
#pragma unroll
      for (int q = QT - 1; q >= 0; q--) {
        dcv = vmul(dpv, dd[q]);
        if (rowL) R.d[q] = vadd(R.d[q], dcv);
        else { R.d[q] = vadd(R.d[q], vadd(dcv, xEv)); R.m[q] = vadd(R.m[q], xEv); }
        dpv = R.d[q];
      }
      // as in Forward: after j left shifts the j highest lanes of dcv are +0, so passes 3 and 4 skip the high pair
      dcv = vlsh(dcv);
#pragma unroll
      for (int q = QT - 1; q >= 0; q--) {
        dcv = vmul(dcv, dd[q]);
        R.d[q] = vadd(R.d[q], dcv);
      }
#pragma unroll
      for (int j = 2; j < 4; j++) {
        dcv = vlsh(dcv);
#pragma unroll
        for (int q = QT - 1; q >= 0; q--) {
          dcv.a = dcv.a * dd[q].a;
          R.d[q].a = R.d[q].a + dcv.a;
        }
      }
    }
    SCHED_FENCE();
    V4 md[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) md[q] = TF(q, tMD);
    dcv = vlsh(R.d[0]);
#pragma unroll
    for (int q = QT - 1; q >= 0; q--) {
      R.m[q] = vadd(R.m[q], vmul(dcv, md[q]));
      dcv = R.d[q];
    }
    SCHED_FENCE();
  } else {
    for (int q = Q - 1; q >= 0; q--) {
      dcv = vmul(dpv, TF(q, tDD));
      if (rowL) R.d[q] = vadd(R.d[q], dcv);
      else { R.d[q] = vadd(R.d[q], vadd(dcv, xEv)); R.m[q] = vadd(R.m[q], xEv); }
      dpv = R.d[q];
    }
    for (int j = 1; j < 4; j++) {
      dcv = vlsh(dcv);
      for (int q = Q - 1; q >= 0; q--) {
        dcv = vmul(dcv, TF(q, tDD));
        R.d[q] = vadd(R.d[q], dcv);
      }
    }
    dcv = vlsh(R.d[0]);
    for (int q = Q - 1; q >= 0; q--) {
      R.m[q] = vadd(R.m[q], vmul(dcv, TF(q, tMD)));
      dcv = R.d[q];
    }
  }
}

// one Backward row i (L-1 >= i >= 1): consumes residue x_{i+1}
struct T6 { V4 ii, mi, bm, mmn, imn, dmn; };
DEV T6 ld6(const float *tb, int q)
{
  T6 t;
  t.ii = vldc(tb + (q * 6 + 0) * 4); t.mi = vldc(tb + (q * 6 + 1) * 4); t.bm = vldc(tb + (q * 6 + 2) * 4);
  t.mmn = vldc(tb + (q * 6 + 3) * 4); t.imn = vldc(tb + (q * 6 + 4) * 4); t.dmn = vldc(tb + (q * 6 + 5) * 4);
  return t;
}
template <int QT>
DEV void bwd_row(Row<QT> &R, const int Q, const float *__restrict__ tf, const float *rfx,
                 float &xN, float &xB, float &xJ, float &xC, float &xE,
                 const float pmove, const float ploop, const float eloop, const float emove)
{
  V4 mpv = vlsh(vmul(R.m[0], vld(rfx)));
  V4 xBv = vzero(), ipv, mcv;
  if constexpr (QT != 0) {
    // tb = the Backward re-ordering of the same transition table (DevProfile::tb follows tf)
    const float *tb = tf + QMAX * 8 * 4 + opaque_zero();
    T6 cur = ld6(tb, QT - 1);
    V4 ecur = vld(rfx + (QT - 1) * 4);
#pragma unroll
    for (int q = QT - 1; q >= 0; q--) {
      WAIT_LGKM0();
      T6 nxt = cur;
      V4 enxt = ecur;
      if (q > 0) { nxt = ld6(tb, q - 1); enxt = vld(rfx + (q - 1) * 4); }
      ipv = R.i[q];
      // the old M of this group is consumed first, so that the new one can take its registers (a value born while the one
      // it replaces is still live makes its whole array move one slot per row: 24 register copies at the loop's end)
      V4 mnext = vmul(R.m[q], ecur);
      V4 imi = vmul(ipv, cur.mi);             // likewise: I*tMI before the new I, whose first product is then the old I's last use
      asm volatile("" : "+v"(mnext.a), "+v"(mnext.b), "+v"(imi.a), "+v"(imi.b));
      R.i[q] = vadd(vmul(ipv, cur.ii), vmul(mpv, cur.imn));
      R.d[q] = vmul(mpv, cur.dmn);
      mcv = vadd(imi, vmul(mpv, cur.mmn));
      R.m[q] = mcv;
      mpv = mnext;
      xBv = vadd(xBv, vmul(mpv, cur.bm));
      // pinned here: left free, instruction selection moves this accumulation chain behind the last group and keeps
      // all twelve B->M operand vectors alive for it (86 SGPR spill moves through VGPR lanes per row)
      asm volatile("" : "+v"(xBv.a), "+v"(xBv.b));
      SCHED_FENCE();
      cur = nxt; ecur = enxt;
    }
  } else {
    const float *tfo = tf + opaque_zero();
    V4 tmmv = vlsh(vldc(tfo + (0 * 8 + tMM) * 4)), timv = vlsh(vldc(tfo + (0 * 8 + tIM) * 4)), tdmv = vlsh(vldc(tfo + (0 * 8 + tDM) * 4));
    for (int q = Q - 1; q >= 0; q--) {
      ipv = R.i[q];
      R.i[q] = vadd(vmul(ipv, vldc(tfo + (q * 8 + tII) * 4)), vmul(mpv, timv));
      R.d[q] = vmul(mpv, tdmv);
      mcv = vadd(vmul(ipv, vldc(tfo + (q * 8 + tMI) * 4)), vmul(mpv, tmmv));
      mpv = vmul(R.m[q], vld(rfx + q * 4));
      R.m[q] = mcv;
      tdmv = vldc(tfo + (q * 8 + tDM) * 4); timv = vldc(tfo + (q * 8 + tIM) * 4); tmmv = vldc(tfo + (q * 8 + tMM) * 4);
      xBv = vadd(xBv, vmul(mpv, vldc(tfo + (q * 8 + tBM) * 4)));
    }
  }
  xB = vhsum(xBv);
  xC = xC * ploop;
  xJ = (xB * pmove) + (xJ * ploop);
  xN = (xB * pmove) + (xN * ploop);
  xE = (xC * emove) + (xJ * eloop);
  bwd_dd_md<QT>(R, Q, tf, vset(xE), false);
}

DEV void fill_lds_rf(float *rf_s, const DevProfile *pp)
{
  for (int i = threadIdx.x; i < NCODE * QMAX * 4; i += 64) rf_s[i] = pp->rf[i];
  __syncthreads();
}

// ---- slab addressing: [row][field][lane] ---------------------------------------------------
// The parser slab of a batch is two planes of [row][6 fields][64 lanes]: Forward's E N J B C S (fields 0-5), then --
// slab_plane floats further -- the six decoding terms Backward writes (fields 6-11).  Every kernel then streams whole
// 1.5-KB rows of ONE plane (Forward writes plane 0, Backward reads it and writes plane 1, the decoder reads plane 1)
// instead of one half of interleaved 3-KB rows.
#define SLAB(a, row0, row, field, lane) \
  ((a).slab + ((field) >= 6 ? (a).slab_plane : (int64_t)0) + ((((row0) + (row)) * 6 + ((field) % 6)) * 64 + (lane)))
// The slab rows are streamed once in each direction and never re-used from a cache: non-temporal accesses keep them from
// evicting the profiles' transition tables (1.5 KB per wave-row through scalar loads) out of L2.  Same-box A/B at 1 M reads:
// Forward 288.8 -> 284.1 ms, Backward 330.8 -> 323.5 ms, 1.041 M -> 1.060 M reads/s.
#define SLAB_ST(p, v) __builtin_nontemporal_store((float)(v), (p))
#define SLAB_LD(p) __builtin_nontemporal_load(p)


// =========================================================================================
// K0: bias-composition filter for every MSV survivor (HMMER p7_bg_FilterScore): a 2-state HMM forward
// over the whole target, rescaled by the row maximum with one deterministic log per residue.  Its own
// kernel: it needs ~40 registers, so it runs at full occupancy instead of inside the 2-wave Forward kernel.
__global__ void __launch_bounds__(256) k_bias(FloatArgs a, int64_t npairs)
{
  __shared__ LogTab ltab[LOGTAB_N];
  if (threadIdx.x < LOGTAB_N) ltab[threadIdx.x] = a.logtab[threadIdx.x];
  __syncthreads();
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs) return;
  const PairRec pr = a.pairs[pi];
  if (pr.prof < 0) return;
  const DevProfile *pp = a.prof + pr.prof;
  const int L = pr.L;
  const Seq sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  const LenTables lt = a.lt[L];
  PairOut po;
  po.nullsc = lt.nullsc; po.pass_bias = 0; po.pass_fwd = 0; po.nregions = 0; po.ndom = 0; po.flags = 0;
  po.filtersc = 0.f; po.fwdsc = 0.f; po.bcksc = 0.f;
  float usc;
  if (pr.xj == 255) usc = __builtin_inff();
  else { usc = ((float)(pr.xj - lt.tjb) - 190.0f); usc /= (float)(3.0 / kLn2); usc -= 3.0f; }
  po.msv_sc = usc;
  {
    const float t00 = lt.p1, t01 = 1.0f - lt.p1, t10 = pp->ft10, t11 = pp->ft11;
    float d0 = 0.f, d1 = 0.f, logsc = 0.0f;
    SeqStream ss; ss.open(sq, 0, +1);
    for (int i = 1; i <= L; i++) {
      const int x = ss.get(i - 1);
      float n0, n1;
      if (i == 1) { n0 = pp->feo[x * 2] * pp->fpi0; n1 = pp->feo[x * 2 + 1] * pp->fpi1; }
      else {
        n0 = 0.0f; n0 += d0 * t00; n0 += d1 * t10; n0 *= pp->feo[x * 2];
        n1 = 0.0f; n1 += d0 * t01; n1 += d1 * t11; n1 *= pp->feo[x * 2 + 1];
      }
      float mx = 0.0f; if (n0 > mx) mx = n0; if (n1 > mx) mx = n1;
      // x / x == 1 exactly: only the smaller state needs the division
      const float q01 = ((n0 > n1) ? n1 : n0) / mx;
      d0 = (n0 == mx) ? 1.0f : q01; d1 = (n1 == mx) ? 1.0f : q01;
      logsc += det_logf_fast(mx, ltab);             // == (float)det_log((double)mx): detmath.h
    }
    float e = 0.0f; e += d0 * 1.0f; e += d1 * 1.0f;
    logsc += (float)det_log((double)e);
    po.filtersc = logsc + lt.bias_a + lt.bias_b;
  }
  const double P = gumbel_surv((double)(usc - po.filtersc) / kLn2, (double)pp->ev[0], (double)pp->ev[1]);
  po.pass_bias = !(P > a.F1);
  a.pout[pi] = po;
}

// =========================================================================================
// K1: bias filter + Forward parser + F3 test, for one wave of survivors of the MSV filter
// BOUND = 1: the lazy domain stage's first pass -- the same recurrence, but nothing is kept except the score (fb[pair]): no slab
// rows, no PairOut (the pairs that matter come back through the BOUND = 0 kernel, so this pass only has to be an upper bound)
template <int QT, int BOUND>
__global__ void __launch_bounds__(64, 2) k_filters_fwd(FloatArgs a, int wave0, float *__restrict__ fb)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const DevProfile *pp = a.prof + uni(wd.prof);
  __shared__ __attribute__((aligned(16))) float rf_s[NCODE * QMAX * 4];
  fill_lds_rf(rf_s, pp);
  const float *tf = pp->tf;
  const int Q = QT ? QT : uni(pp->Q);
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  const int L = pr.L;
  const Seq sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  PairOut po;
  if constexpr (!BOUND) po = a.pout[pi];            // filtersc / pass_bias come from k_bias
  const int Lw = wd.rows - 1;       // longest sequence in this wave

  // ---- Forward parser
  {
    Row<QT> R;
#pragma unroll
    for (int q = 0; q < QMAX; q++) { R.m[q] = vzero(); R.d[q] = vzero(); R.i[q] = vzero(); }
    const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
    const float ploop = 1.0f - pmove;
    float xE = 0.f, xN = 1.f, xJ = 0.f, xB = pmove, xC = 0.f, totscale = 0.0f;
    const int64_t r0 = wd.slab;
    if constexpr (!BOUND) {
      *SLAB(a, r0, 0, 0, lane) = xE; *SLAB(a, r0, 0, 1, lane) = xN;
      *SLAB(a, r0, 0, 2, lane) = xJ; *SLAB(a, r0, 0, 3, lane) = xB;
      *SLAB(a, r0, 0, 4, lane) = xC; *SLAB(a, r0, 0, 5, lane) = 1.0f;
    }
    SeqStream ss; ss.open(sq, 0, +1);
    int xnext = ss.get(0);                   // residue of the NEXT row
    for (int i = 1; i <= Lw; i++) {
      if (i <= L) {
        const int x = xnext;
        if (i < L) xnext = ss.get(i);
        fwd_row<QT>(R, Q, tf, rf_s + x * QMAX * 4, xN, xB, xJ, xC, xE, pmove, ploop, 0.5f, 0.5f);
        float sc = 1.0f;
        if (xE > 1.0e4f) {
          xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
          scale_row<QT>(R, Q, xE);
          sc = xE;
          totscale = (float)((double)totscale + det_log((double)xE));
          xE = 1.0f;
        }
        if constexpr (!BOUND) {
          SLAB_ST(SLAB(a, r0, i, 0, lane), xE); SLAB_ST(SLAB(a, r0, i, 1, lane), xN);
          SLAB_ST(SLAB(a, r0, i, 2, lane), xJ); SLAB_ST(SLAB(a, r0, i, 3, lane), xB);
          SLAB_ST(SLAB(a, r0, i, 4, lane), xC); SLAB_ST(SLAB(a, r0, i, 5, lane), sc);
        }
        (void)sc;
      }
    }
    const bool bad = (xC != xC) || (xC == 0.0f) || (xC == __builtin_inff());
    po.fwdsc = (float)((double)totscale + det_log((double)(xC * pmove)));
    if constexpr (BOUND) {
      // a score the pipeline would call unusable gets no bound at all (NaN): such a pair is always evaluated
      if (active) fb[pi] = bad ? __builtin_nanf("") : po.fwdsc;
      return;
    }
    const double P = exp_surv((double)(po.fwdsc - po.filtersc) / kLn2, (double)pp->ev[4], (double)pp->ev[5]);
    po.pass_fwd = po.pass_bias && (!a.vit || a.vit[pi].pass) && !bad && !(P > a.F3);
  }
  if (active) a.pout[pi] = po;
}

// =========================================================================================
// K2: Backward parser + posterior decoding of B/E/occupancy + region scan
template <int QT>
__global__ void __launch_bounds__(64, 2) k_bwd_decode(FloatArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const DevProfile *pp = a.prof + uni(wd.prof);
  __shared__ __attribute__((aligned(16))) float rf_s[NCODE * QMAX * 4];
  fill_lds_rf(rf_s, pp);
  const float *tf = pp->tf;
  const int Q = QT ? QT : uni(pp->Q);
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  PairOut po = a.pout[pi];
  const bool alive = active && po.pass_fwd;
  if (__ballot(alive) == 0ull) return;
  const int L = pr.L;
  const Seq sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  const int Lw = wd.rows - 1;
  const int64_t r0 = wd.slab;
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
  const float ploop = 1.0f - pmove;
  int own = 0;
  bool bad = false;
  {
    Row<QT> R;
    float xJ = 0.f, xB = 0.f, xN = 0.f, xC = pmove, xE = xC * 0.5f, totscale;
    {
      const V4 xEv = vset(xE);
#pragma unroll
      for (int q = 0; q < QMAX; q++) { R.m[q] = xEv; R.d[q] = xEv; R.i[q] = vzero(); }
      bwd_dd_md<QT>(R, Q, tf, xEv, true);
    }
    // rows are walked from the wave's longest length down; a lane joins at its own L
    float sL = 1.0f;
    if (alive) sL = *SLAB(a, r0, L, 5, lane);
    if (sL > 1.0f) {
      xE = xE / sL; xN = xN / sL; xC = xC / sL; xJ = xJ / sL; xB = xB / sL;
      scale_row<QT>(R, Q, sL);
    }
    totscale = (float)det_log((double)sL);
    // The posterior-decoding pass needs, per row, five products of Forward and Backward special-state values.
    // They are formed here (the Forward row is prefetched while the Backward row is computed) in exactly the
    // association order of the decoding formulas -- ((f*b)*c), later times the running scale product -- so the
    // decoding kernel streams 5 floats per row instead of 12:
    //   field 9 : T1[i] = (fB[i]*bB[i])*fS[i]         -> btot[i+1]        field 6 : T2[i] = (fE[i]*bE[i])*fS[i] -> etot[i]
    //   field 7/8/10 : (fN|fJ|fC)[i-1]*b(N|J|C)[i]*ploop -> occupancy       field 11: fS[i]/bS[i] (used only with own scales)
    struct FRow { float E, N, J, B, C, S; };
    auto load_f = [&](int i) {
      FRow f;
      f.E = SLAB_LD(SLAB(a, r0, i, 0, lane)); f.N = SLAB_LD(SLAB(a, r0, i, 1, lane)); f.J = SLAB_LD(SLAB(a, r0, i, 2, lane));
      f.B = SLAB_LD(SLAB(a, r0, i, 3, lane)); f.C = SLAB_LD(SLAB(a, r0, i, 4, lane)); f.S = SLAB_LD(SLAB(a, r0, i, 5, lane));
      return f;
    };
    auto store_terms = [&](int i, const FRow &fc, const FRow &fp, float s) {
      SLAB_ST(SLAB(a, r0, i, 6, lane), fc.E * xE * fc.S);
      SLAB_ST(SLAB(a, r0, i, 7, lane), fp.N * xN * ploop);
      SLAB_ST(SLAB(a, r0, i, 8, lane), fp.J * xJ * ploop);
      SLAB_ST(SLAB(a, r0, i, 9, lane), fc.B * xB * fc.S);
      SLAB_ST(SLAB(a, r0, i, 10, lane), fp.C * xC * ploop);
      // Forward's scale over Backward's: 1 exactly (x / x) unless the lane rescaled on its own -- the division is taken only then
      float ratio = 1.0f;
      if (fc.S != s) { ratio = fc.S / s; asm volatile("" : "+v"(ratio)); }
      SLAB_ST(SLAB(a, r0, i, 11, lane), ratio);
    };
    FRow fcur, fprv;
    fcur.E = fcur.N = fcur.J = fcur.B = fcur.C = 0.f; fcur.S = sL; fprv = fcur;
    if (alive) { fcur = load_f(L); fprv = load_f(L >= 1 ? L - 1 : 0); store_terms(L, fcur, fprv, sL); }
    SeqStream ss; ss.open(sq, L >= 2 ? L - 1 : 0, -1);
    int xnext = (alive && L >= 2) ? ss.get(L - 1) : 0;
    for (int i = Lw - 1; i >= 1; i--) {
      if (alive && i <= L - 1) {
        fcur = fprv;                       // Forward's row i, requested one row ago
        // taken here, before anything else of this row may issue a load: the wait is then "all but the six stores of the row
        // before"; behind the stream's occasional load the compiler can only wait for everything, stores included
        asm volatile("" : "+v"(fcur.E), "+v"(fcur.N), "+v"(fcur.J), "+v"(fcur.B), "+v"(fcur.C), "+v"(fcur.S));
        const int x = xnext;               // residue i+1, 0-based index i
        xnext = ss.get(i - 1);             // the next row down needs residue i (0-based i-1)
        fprv = load_f(i - 1);
        const float sfw = fcur.S;
        bwd_row<QT>(R, Q, tf, rf_s + x * QMAX * 4, xN, xB, xJ, xC, xE, pmove, ploop, 0.5f, 0.5f);
        if (xB > 1.0e16f) own = 1;
        // branch-free on purpose: a branch here lets the optimizer sink the whole row update below it,
        // stretching the live ranges of every transition operand (hundreds of SGPR spills)
        const float sown = (xB > 1.0e4f) ? xB : 1.0f;
        const float s = own ? sown : sfw;
        if (s > 1.0f) {
          xE /= s; xN /= s; xJ /= s; xB /= s; xC /= s;
          scale_row<QT>(R, Q, s);
          totscale = (float)((double)totscale + det_log((double)s));
        }
        store_terms(i, fcur, fprv, s);
      }
    }
    // row 0
    if (alive) {
      const int x = sq.code(0);
      const float *rfx = rf_s + x * QMAX * 4;
      V4 xBv = vzero();
#pragma unroll
      for (int q = (QT ? QT : QMAX) - 1; q >= 0; q--) {
        if (QT == 0 && q >= Q) continue;
        const V4 mpv = vmul(R.m[q], vld(rfx + q * 4));
        xBv = vadd(xBv, vmul(mpv, TF(q, tBM)));
      }
      xB = vhsum(xBv);
      xN = (xB * pmove) + (xN * ploop);
      const FRow f0 = (L >= 2) ? fprv : load_f(0);     // L == 1: the row loop never ran
      *SLAB(a, r0, 0, 7, lane) = xN;       // Backward's N at row 0 = total probability (the decoder's 1/x)
      *SLAB(a, r0, 0, 9, lane) = f0.B * xB * f0.S;
      *SLAB(a, r0, 0, 11, lane) = f0.S / 1.0f;
      bad = (xN != xN) || (xN == 0.0f) || (xN == __builtin_inff());
      po.bcksc = (float)((double)totscale + det_log((double)xN));
    }
  }
  if (active) { po.nregions = 0; po.ndom = 0; po.flags = (own ? 4 : 0) | (bad ? 8 : 0); a.pout[pi] = po; }
}

// =========================================================================================
// K2b: posterior decoding of domain starts / ends / occupancy and the region scan, rows ascending (the order
// the sums are defined in).  ~25 flops per row on 12 values streamed from the slab: latency-bound, so it is
// its own low-register kernel and runs at full occupancy instead of at the DP kernels' 2 waves per SIMD.
__global__ void __launch_bounds__(64) k_decode(FloatArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  PairOut po = a.pout[pi];
  const bool alive = active && po.pass_fwd;
  const int own = (po.flags & 4) ? 1 : 0;
  const bool bad = (po.flags & 8) != 0;
  const int L = pr.L;
  const int64_t r0 = wd.slab;
  // ---- posterior decoding + region scan, rows ascending (the order the sums are defined in)
  int nreg = 0, nkept = 0, flags = 0, nmulti = 0;
  if (alive && !bad) {
    const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
    float scaleproduct = (float)(1.0 / (double)*SLAB(a, r0, 0, 7, lane));
    float btot = 0.f, etot = 0.f;
    // one row of decoding terms (see k_bwd_decode); the next row is requested while the current one is consumed,
    // so the serial chain of sums never waits on HBM latency
    struct DRow { float t2, t3, t4, t1, t5, rs; };
    auto load_row = [&](int j) {
      DRow d;
      d.t2 = SLAB_LD(SLAB(a, r0, j, 6, lane)); d.t3 = SLAB_LD(SLAB(a, r0, j, 7, lane));
      d.t4 = SLAB_LD(SLAB(a, r0, j, 8, lane)); d.t1 = SLAB_LD(SLAB(a, r0, j, 9, lane));
      d.t5 = SLAB_LD(SLAB(a, r0, j, 10, lane));
      d.rs = own ? *SLAB(a, r0, j, 11, lane) : 1.0f;
      return d;
    };
    DRow prv = load_row(0);
    DRow nxt = load_row(L >= 1 ? 1 : 0);
    int ri = -1; bool triggered = false;
    // The cumulative sums are NOT written back (8 B per row saved): a region's inner scan needs btot/etot of its own rows
    // only, and those are rebuilt from a checkpoint taken where the region starts -- same operations, same order, same bits.
    float ck_btot = 0.f, ck_etot = 0.f, ck_sp = scaleproduct, ck_t1 = prv.t1, ck_rs = prv.rs;
    struct Pend { int ri, j, slot; float b, e, sp, t1, rs, btot; };
    Pend p0, p1; int npend = 0;
    p0.ri = p0.j = p0.slot = 0; p0.b = p0.e = p0.sp = p0.t1 = p0.rs = p0.btot = 0.f; p1 = p0;
    // max over z in [ri, j] of min(etot[z] - etot[ri-1], btot[j] - btot[z-1]) >= rt3 ?  The sums are rebuilt from the
    // checkpoint taken where the region starts, with the same operations in the same order as the row loop's
    auto close_region = [&](const Pend &p) {
      float mx = -1.0f;
      float b = p.b, e = p.e, sp = p.sp, t1p = p.t1, rsp = p.rs;
      const float et0 = p.e;
      // four rows are requested before the first is used: the sums are a serial chain, the loads are not
      for (int z0 = p.ri; z0 <= p.j; z0 += 4) {
        float t2v[4], t1v[4], rsv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int z = z0 + k <= p.j ? z0 + k : p.j;
          t2v[k] = *SLAB(a, r0, z, 6, lane); t1v[k] = *SLAB(a, r0, z, 9, lane);
          rsv[k] = own ? *SLAB(a, r0, z, 11, lane) : 1.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (z0 + k > p.j) break;
          const float bprev = b;                         // btot[z-1]
          b = b + (t1p * sp);
          if (own) sp *= rsp;
          e = e + (t2v[k] * sp);                         // etot[z]
          t1p = t1v[k]; rsp = rsv[k];
          const float ea = e - et0;
          const float bb = p.btot - bprev;
          const float m2 = ea < bb ? ea : bb;
          mx = m2 > mx ? m2 : mx;
        }
      }
      const int multi = (mx >= rt3);
      nmulti += multi;
      RegionRec rr; rr.pair = (int32_t)pi; rr.ienv = p.ri; rr.jenv = p.j; rr.multi = multi;
      if (p.slot < MAXDOM) a.regions[pi * MAXDOM + p.slot] = rr;
      else pool_put(a.pool, rr, p.slot);          // past the pair's slots: the overflow list (engine.hip orders it afterwards)
    };
    for (int j = 1; j <= L; j++) {
      const DRow cur = nxt;
      if (j < L) nxt = load_row(j + 1);
      const float btot_prev = btot, etot_prev = etot, sp_prev = scaleproduct;
      btot = btot + (prv.t1 * scaleproduct);
      if (own) scaleproduct *= prv.rs;
      etot = etot + (cur.t2 * scaleproduct);
      float njcp;
      njcp = cur.t3 * scaleproduct;
      njcp += cur.t4 * scaleproduct;
      njcp += cur.t5 * scaleproduct;
      const float mocc = (float)(1. - (double)njcp);
      const float pt1 = prv.t1, prs = prv.rs;
      prv = cur;
      if (!triggered) {
        bool set = false;
        if (mocc - (btot - btot_prev) < rt2) { ri = j; set = true; }
        else if (ri == -1) { ri = j; set = true; }
        if (set) { ck_btot = btot_prev; ck_etot = etot_prev; ck_sp = sp_prev; ck_t1 = pt1; ck_rs = prs; }   // state just before row ri
        if (mocc >= rt1) triggered = true;
      } else if (mocc - (etot - etot_prev) < rt2) {
        nreg++;
        // The multidomain test of a region re-reads the region's rows.  Done here it would run for ONE lane while the
        // other 63 wait (lanes close their regions at different rows) and it is a chain of dependent single-lane loads;
        // so the first two regions of a lane are only recorded, and tested after the row loop by all lanes together.
        const int slot = nkept++;                  // hmmsearch keeps every region of a target: so does this (flag 2 = past the slots)
        if (slot >= MAXDOM) flags |= 2;
        if (npend < 2) {
          Pend &p = npend == 0 ? p0 : p1;
          p.ri = ri; p.j = j; p.slot = slot; p.b = ck_btot; p.e = ck_etot; p.sp = ck_sp; p.t1 = ck_t1; p.rs = ck_rs; p.btot = btot;
          npend++;
        } else {
          Pend p; p.ri = ri; p.j = j; p.slot = slot; p.b = ck_btot; p.e = ck_etot; p.sp = ck_sp; p.t1 = ck_t1; p.rs = ck_rs; p.btot = btot;
          close_region(p);
        }
        ri = -1; triggered = false;
      }
    }
    if (npend > 0) close_region(p0);
    if (npend > 1) close_region(p1);
    if (scaleproduct == __builtin_inff()) { nreg = 0; nkept = 0; }
  }
  if (active && alive) { po.nregions = nreg; po.ndom = nkept; po.flags = flags | ((nmulti > 255 ? 255 : nmulti) << 8); a.pout[pi] = po; }   // bits 8-15: multidomain regions (statistics)
}

// =========================================================================================
// K3: re-score one envelope per lane in unihit mode (Forward, Backward, decoding, null2).
// Three sweeps over the envelope's rows, one kernel each (separate kernels keep every sweep
// inside 256 VGPRs without spills); only the Backward rows go through HBM:
//   A. k_env_fwd : Forward, keeping nothing but each row's scale factor     (write 1 float / row)
//   B. k_env_bwd : Backward with those scale factors, M and I of every row  (write 100 floats / row)
//   C. k_env_post: Forward again (same arithmetic, same values), each freshly computed row combined
//      with the stored Backward row into the posterior sums, rows ascending as HMMER sums them
//      (read 100 floats / row); then null2 and the envelope's null2 correction.
// Recomputing Forward costs one more sweep of ALU work and halves the slab traffic that bounded
// the first version of this stage (fwd rows + bck rows written, both read back).
constexpr int EV = 26;     // float4 vectors per row: bck M,I of each q (24) | bck (N J C S) | (fwd S, -, -, -)
// 16 bytes per lane per access: one global_store_dwordx4 / global_load_dwordx4 moves 1 KiB per wave
DEV f4 *eslab_at(float *slab, int64_t row0, int row, int v, int lane)
{
  return (f4 *)slab + (((row0 + row) * EV + v) * 64 + lane);
}

struct EnvLane {           // what every sweep needs to know about its lane's envelope
  Seq sq; int L, Ld, off, Lw; int64_t r0, ri; float pmove, ploop; bool active;
};
DEV EnvLane env_lane(const EnvArgs &a, const WaveDesc &wd, int lane)
{
  EnvLane e;
  e.active = lane < wd.count;
  e.ri = wd.first + (e.active ? lane : 0);
  const RegionRec rg = a.regions[e.ri];
  const PairRec pr = a.pairs[rg.pair];
  e.L = pr.L;
  e.sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  e.Ld = rg.jenv - rg.ienv + 1;
  e.off = rg.ienv - 1;
  e.Lw = wd.rows - 1;
  e.r0 = wd.slab;
  e.pmove = 2.0f / ((float)e.L + 2.0f);
  e.ploop = 1.0f - e.pmove;
  return e;
}

template <int QT>
__global__ void __launch_bounds__(64, 2) k_env_fwd(EnvArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const DevProfile *pp = a.prof + uni(wd.prof);
  __shared__ __attribute__((aligned(16))) float rf_s[NCODE * QMAX * 4];
  fill_lds_rf(rf_s, pp);
  const float *tf = pp->tf;
  const int Q = QT ? QT : uni(pp->Q);
  const EnvLane e = env_lane(a, wd, lane);
  Row<QT> R;
#pragma unroll
  for (int q = 0; q < QMAX; q++) { R.m[q] = vzero(); R.d[q] = vzero(); R.i[q] = vzero(); }
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = e.pmove, xC = 0.f, totscale = 0.0f;
  for (int i = 1; i <= e.Lw; i++) {
    if (e.active && i <= e.Ld) {
      const int x = e.sq.code(e.off + i - 1);
      fwd_row<QT>(R, Q, tf, rf_s + x * QMAX * 4, xN, xB, xJ, xC, xE, e.pmove, e.ploop, 0.0f, 1.0f);
      float sc = 1.0f;
      if (xE > 1.0e4f) {
        xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
        scale_row<QT>(R, Q, xE);
        sc = xE;
        totscale = (float)((double)totscale + det_log((double)xE));
        xE = 1.0f;
      }
      eslab_at(a.slab, e.r0, i, 25, lane)->x = sc;
    }
  }
  if (e.active) {
    RegionOut ro;
#pragma unroll
    for (int x = 0; x < NCODE; x++) ro.n2log[x] = 0.f;
    ro.domcorrection = 0.f;
    const bool bad = (xC != xC) || (xC == 0.0f) || (xC == __builtin_inff());
    ro.envsc = (float)((double)totscale + det_log((double)(xC * e.pmove)));
    ro.ok = bad ? -1 : 0;                 // -1: unusable; 0: not finished yet; 1: complete
    ro.own = 0; ro.bN0 = 0.f;
    a.rout[e.ri] = ro;
  }
}

template <int QT>
__global__ void __launch_bounds__(64, 2) k_env_bwd(EnvArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const DevProfile *pp = a.prof + uni(wd.prof);
  __shared__ __attribute__((aligned(16))) float rf_s[NCODE * QMAX * 4];
  fill_lds_rf(rf_s, pp);
  const float *tf = pp->tf;
  const int Q = QT ? QT : uni(pp->Q);
  const EnvLane e = env_lane(a, wd, lane);
  const int Ld = e.Ld; const int64_t r0 = e.r0; const bool active = e.active;
  const float pmove = e.pmove, ploop = e.ploop;
  int own = 0;
  Row<QT> R;
  float xJ = 0.f, xB = 0.f, xN = 0.f, xC = pmove, xE = xC * 1.0f;
  {
    const V4 xEv = vset(xE);
#pragma unroll
    for (int q = 0; q < QMAX; q++) { R.m[q] = xEv; R.d[q] = xEv; R.i[q] = vzero(); }
    bwd_dd_md<QT>(R, Q, tf, xEv, true);
  }
  float sL = 1.0f;
  if (active) sL = eslab_at(a.slab, r0, Ld, 25, lane)->x;
  if (sL > 1.0f) {
    xE = xE / sL; xN = xN / sL; xC = xC / sL; xJ = xJ / sL; xB = xB / sL;
    scale_row<QT>(R, Q, sL);
  }
  auto store_row = [&](int i, float s) {
    __builtin_nontemporal_store((f4){xN, xJ, xC, s}, eslab_at(a.slab, r0, i, 24, lane));
#pragma unroll
    for (int q = 0; q < (QT ? QT : QMAX); q++) {
      if (QT == 0 && q >= Q) break;
      __builtin_nontemporal_store((f4){R.m[q].a.x, R.m[q].a.y, R.m[q].b.x, R.m[q].b.y}, eslab_at(a.slab, r0, i, q * 2, lane));
      __builtin_nontemporal_store((f4){R.i[q].a.x, R.i[q].a.y, R.i[q].b.x, R.i[q].b.y}, eslab_at(a.slab, r0, i, q * 2 + 1, lane));
    }
  };
  if (active) store_row(Ld, sL);
  for (int i = e.Lw - 1; i >= 1; i--) {
    if (active && i <= Ld - 1) {
      const int x = e.sq.code(e.off + i);
      bwd_row<QT>(R, Q, tf, rf_s + x * QMAX * 4, xN, xB, xJ, xC, xE, pmove, ploop, 0.0f, 1.0f);
      if (xB > 1.0e16f) own = 1;
      const float sfw = eslab_at(a.slab, r0, i, 25, lane)->x;
      const float sown = (xB > 1.0e4f) ? xB : 1.0f;
      const float s = own ? sown : sfw;
      if (s > 1.0f) {
        xE /= s; xN /= s; xJ /= s; xB /= s; xC /= s;
        scale_row<QT>(R, Q, s);
      }
      store_row(i, s);
    }
  }
  if (active) {
    const int x = e.sq.code(e.off);
    const float *rfx = rf_s + x * QMAX * 4;
    V4 xBv = vzero();
#pragma unroll
    for (int q = (QT ? QT : QMAX) - 1; q >= 0; q--) {
      if (QT == 0 && q >= Q) continue;
      const V4 mpv = vmul(R.m[q], vld(rfx + q * 4));
      xBv = vadd(xBv, vmul(mpv, TF(q, tBM)));
    }
    xB = vhsum(xBv);
    xN = (xB * pmove) + (xN * ploop);
    const bool bad = (xN != xN) || (xN == 0.0f) || (xN == __builtin_inff());
    if (bad) a.rout[e.ri].ok = -1;
    a.rout[e.ri].own = own; a.rout[e.ri].bN0 = xN;
  }
}

template <int QT>
__global__ void __launch_bounds__(64, 2) k_env_post(EnvArgs a, int wave0)
{
  // this sweep needs the DP row (144 registers) plus 96 posterior sums per lane; the 48 insert-state
  // sums live in LDS ([q][lane] x float4: conflict-free b128 accesses) so the rest fits in 256 VGPRs
  __shared__ f4 accI_s[QMAX * 64];
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const DevProfile *pp = a.prof + uni(wd.prof);
  __shared__ __attribute__((aligned(16))) float rf_s[NCODE * QMAX * 4];
  fill_lds_rf(rf_s, pp);
  const float *tf = pp->tf;
  const int Q = QT ? QT : uni(pp->Q);
  const EnvLane e = env_lane(a, wd, lane);
  const int Ld = e.Ld; const int64_t r0 = e.r0;
  const float pmove = e.pmove, ploop = e.ploop;
  RegionOut ro = a.rout[e.ri];
  const int own = ro.own;
  const bool go = e.active && ro.ok == 0;
  Row<QT> R;
#pragma unroll
  for (int q = 0; q < QMAX; q++) { R.m[q] = vzero(); R.d[q] = vzero(); R.i[q] = vzero(); }
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = pmove, xC = 0.f;
  float scaleproduct = (float)(1.0 / (double)ro.bN0);
  V4 accM[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; q++) { accM[q] = vzero(); accI_s[q * 64 + lane] = (f4){0.f, 0.f, 0.f, 0.f}; }
  float accN = 0.f, accC = 0.f, accJ = 0.f;
  for (int r = 1; r <= e.Lw; r++) {
    if (go && r <= Ld) {
      const float fNp = xN, fJp = xJ, fCp = xC;
      const int x = e.sq.code(e.off + r - 1);
      fwd_row<QT>(R, Q, tf, rf_s + x * QMAX * 4, xN, xB, xJ, xC, xE, pmove, ploop, 0.0f, 1.0f);
      float fS = 1.0f;
      if (xE > 1.0e4f) {
        xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
        scale_row<QT>(R, Q, xE);
        fS = xE;
        xE = 1.0f;
      }
      const V4 totrv = vset(scaleproduct * fS);
#pragma unroll
      for (int q = 0; q < (QT ? QT : QMAX); q++) {
        if (QT == 0 && q >= Q) break;
        const f4 bm4 = __builtin_nontemporal_load(eslab_at(a.slab, r0, r, q * 2, lane)), bi4 = __builtin_nontemporal_load(eslab_at(a.slab, r0, r, q * 2 + 1, lane));
        V4 bm, bi;
        bm.a = (f2){bm4.x, bm4.y}; bm.b = (f2){bm4.z, bm4.w}; bi.a = (f2){bi4.x, bi4.y}; bi.b = (f2){bi4.z, bi4.w};
        const V4 pm = vmul(vmul(R.m[q], bm), totrv);
        const V4 pi = vmul(vmul(R.i[q], bi), totrv);
        if (r == 1) { accM[q] = pm; accI_s[q * 64 + lane] = (f4){pi.a.x, pi.a.y, pi.b.x, pi.b.y}; }
        else {
          accM[q] = vadd(pm, accM[q]);
          const f4 o = accI_s[q * 64 + lane];
          V4 ai; ai.a = (f2){o.x, o.y}; ai.b = (f2){o.z, o.w};
          ai = vadd(pi, ai);
          accI_s[q * 64 + lane] = (f4){ai.a.x, ai.a.y, ai.b.x, ai.b.y};
        }
      }
      const f4 bsp = __builtin_nontemporal_load(eslab_at(a.slab, r0, r, 24, lane));
      const float bN = bsp.x, bJ = bsp.y, bC = bsp.z, bS = bsp.w;
      const float pN = fNp * bN * ploop * scaleproduct;
      const float pJ = fJp * bJ * ploop * scaleproduct;
      const float pC = fCp * bC * ploop * scaleproduct;
      if (r == 1) { accN = pN; accC = pC; accJ = pJ; } else { accN += pN; accC += pC; accJ += pJ; }
      if (own) scaleproduct *= fS / bS;
    }
  }
  int ok = ro.ok;
  if (go && scaleproduct != __builtin_inff()) {
    const float norm = (float)(1.0 / (double)(float)Ld);
    const V4 nv = vset(norm);
    V4 accI[QMAX];
#pragma unroll
    for (int q = 0; q < (QT ? QT : QMAX); q++) {
      if (QT == 0 && q >= Q) break;
      const f4 o = accI_s[q * 64 + lane];
      accI[q].a = (f2){o.x, o.y}; accI[q].b = (f2){o.z, o.w};
      accM[q] = vmul(accM[q], nv); accI[q] = vmul(accI[q], nv);
    }
    accN *= norm; accC *= norm; accJ *= norm;
    const float xfactor = accN + accC + accJ;
    float null2[NCODE];
#pragma unroll
    for (int x = 0; x < 4; x++) {
      V4 sv = vzero();
#pragma unroll
      for (int q = 0; q < (QT ? QT : QMAX); q++) {
        if (QT == 0 && q >= Q) break;
        sv = vadd(sv, vmul(accM[q], vld(rf_s + x * QMAX * 4 + q * 4)));
        sv = vadd(sv, accI[q]);
      }
      null2[x] = vhsum(sv);
      null2[x] += xfactor;
    }
    null2[4] = 1.0f;
    // degenerate codes: mean of the member odds (order A,C,G,T)
    const unsigned short dm[16] = {1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15};
#pragma unroll
    for (int x = 5; x < 16; x++) {
      float acc = 0.f; int nd = 0;
#pragma unroll
      for (int y = 0; y < 4; y++) if (dm[x] >> y & 1) { acc += null2[y]; nd++; }
      null2[x] = acc / (float)nd;
    }
#pragma unroll
    for (int x = 0; x < NCODE; x++) ro.n2log[x] = det_logf(null2[x]);
    // null2 correction of this envelope: sum over its residues, in order
    float dc = 0.0f;
    for (int pos = 0; pos < Ld; pos++) {
      const int x = e.sq.code(e.off + pos);
      float v = ro.n2log[0];
#pragma unroll
      for (int c = 1; c < NCODE; c++) v = (x == c) ? ro.n2log[c] : v;
      dc += v;
    }
    ro.domcorrection = dc;
    ok = 1;
  }
  if (e.active) { ro.ok = ok > 0 ? 1 : 0; a.rout[e.ri] = ro; }
}

// =========================================================================================
// K4: per pair, chain its envelopes in order: null2 corrections, bit scores, reporting
// count one "reported target" for profile `prof` per flagged lane, with one atomic per distinct profile per
// wave: the work lists are grouped by profile, so per-lane atomics would all hit the same address
// (`prof` is the counter's index: sample * P + profile when several samples share the batch)
DEV void count_reported(int32_t *domz, int prof, bool flag)
{
  unsigned long long todo = __ballot(flag);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int p0 = __shfl(prof, leader, 64);
    const unsigned long long same = __ballot(flag && prof == p0) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&domz[p0], __popcll(same));
    todo &= ~same;
  }
}

DEV float flogsum_dev(const float *tbl, float x, float y)
{
  const float mx = x > y ? x : y, mn = x > y ? y : x;
  return (mn == -__builtin_inff() || (mx - mn) >= 15.7f) ? mx : mx + tbl[(int)((mx - mn) * 1000.f)];
}

DEV void emit_domain(const ScoreArgs &a, int64_t slot, const PairRec &pr, const DevProfile *pp, const RegionRec &rg, const RegionOut &ro,
                     float domcorr, int dom_idx, int ndom, int poflags, float nullsc, float seq_score, float final_bias, int seq_rep,
                     int L, double lognn3, double log_omega)
{
  itsx_domain o;
  const int Ld = rg.jenv - rg.ienv + 1;
  float bits = (float)((double)ro.envsc + (double)(L - Ld) * lognn3);
  const float dombias = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + (double)domcorr));
  bits = (float)((double)(bits - (nullsc + dombias)) / kLn2);
  o.rep = a.sorted_uniq[pr.useq]; o.prof = pr.prof; o.tlen = L; o.ienv = rg.ienv; o.jenv = rg.jenv;
  o.dom_idx = dom_idx; o.ndom = ndom; o.flags = (rg.multi ? 1 : 0) | (poflags & 0xff);
  o.envsc = ro.envsc; o.domcorrection = domcorr; o.dombias = dombias; o.bitscore = bits;
  o.lnP = exp_logsurv((double)bits, (double)pp->ev[4], (double)pp->ev[5]);
  o.seq_score = seq_score; o.seq_bias = (float)((double)final_bias / kLn2);
  o.seq_reported = dom_idx >= 0 ? seq_rep : 0; o.dom_reported = 0;
  a.dom[slot] = o;
}

__global__ void __launch_bounds__(256) k_score(ScoreArgs a)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= a.npairs) return;
  const PairOut po = a.pout[pi];
  if (!po.pass_fwd || po.ndom <= 0) return;
  const PairRec pr = a.pairs[pi];
  const DevProfile *pp = a.prof + pr.prof;
  const int L = pr.L;
  const LenTables lt = a.lt[L];
  const double log_omega = -5.545177444479562;   // log(1/256), the null2 prior
  const int64_t g0 = a.pair_region0[pi];
  const int nd_all = po.ndom;
  const float nullsc = po.nullsc;
  const int mr0 = a.mr_off ? a.mr_off[pi] : 0, mr1 = a.mr_off ? a.mr_off[pi + 1] : 0;
  if (nd_all == 1 && mr1 == mr0) {
    // the common case, one envelope: every running sum of the general path collapses to one term
    const RegionOut ro = a.rout[a.upos[g0]];
    if (!ro.ok) return;
    const RegionRec rg = a.regions[g0];
    const float dc = ro.domcorrection;
    const float seqbias = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + (double)dc));
    float seq_score = (float)((double)(po.fwdsc - (nullsc + seqbias)) / kLn2);
    float final_bias = seqbias;
    if (ro.envsc - dc > 0.0f) {
      const int Ld = rg.jenv - rg.ienv + 1;
      float sum_score = 0.0f; sum_score += ro.envsc;
      float sb = 0.0f; sb += dc;
      sb = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + (double)sb));
      sum_score = (float)((double)sum_score + (double)(L - Ld) * lt.lognn3);
      sum_score = (float)((double)(sum_score - (nullsc + sb)) / kLn2);
      if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; final_bias = sb; }
    } else {
      float sb = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + 0.0));
      float sum_score = (float)(0.0 + (double)L * lt.lognn3);
      sum_score = (float)((double)(sum_score - (nullsc + sb)) / kLn2);
      (void)sum_score;                      // Ld == 0: the reconstruction score cannot override
    }
    const int seq_rep = ((double)seq_score >= a.T);
    emit_domain(a, g0, pr, pp, rg, ro, dc, 0, 1, po.flags, nullsc, seq_score, final_bias, seq_rep, L, lt.lognn3, log_omega);
    count_reported(a.domz, (a.usample ? a.usample[a.sorted_uniq[pr.useq]] * a.P : 0) + pr.prof, seq_rep != 0);
    return;
  }
  const Seq sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  // hmmsearch keeps one null2 score per residue (ddef->n2sc): a simple region's residues get log null2[x] of its envelope
  // (k_env_post), a clustered region's residues the scores of its traceback ensemble (k_mr_trace).  The per-sequence
  // correction is ONE running float sum over all residues in order; an envelope's own correction the sum over its residues.
  auto add_region = [&](int m, float acc) {
    const int u = a.mr_u[m];
    if (a.mrout[u].status != 0) return acc;
    const MrRec mr = a.mr[m];
    const float *n2 = a.n2sc + a.n2off[u];
    for (int pos = mr.ireg; pos <= mr.jreg; pos++) acc += n2[pos - mr.ireg];
    return acc;
  };
  // with the ensemble stage switched off (ITSX_NO_ENSEMBLE) there are no clustered regions: `multi` is then only the flag
  // of a region kept as one envelope, and every envelope is a simple one
  const bool have_mr = a.mr != nullptr;
  auto env_dc = [&](const RegionRec &rg, const RegionOut &ro) {
    if (!have_mr || rg.multi <= 0) return ro.domcorrection;      // (< 0: a region whose ensemble failed, kept as a simple envelope)
    const int m = rg.multi - 1;
    const float *n2 = a.n2sc + a.n2off[a.mr_u[m]];
    const int ireg = a.mr[m].ireg;
    float dc = 0.0f;
    for (int pos = rg.ienv; pos <= rg.jenv; pos++) dc += n2[pos - ireg];
    return dc;
  };
  float seqbias = 0.0f;
  bool any = false;
  int ndom = 0, mk = mr0;
  for (int d = 0; d < nd_all; d++) {
    const RegionRec rg = a.regions[g0 + d];
    const RegionOut ro = a.rout[a.upos[g0 + d]];
    if (!have_mr || rg.multi <= 0) {
      while (mk < mr1 && a.mr[mk].ireg < rg.ienv) { seqbias = add_region(mk, seqbias); any = true; mk++; }
      if (ro.ok) {
        if (!any) seqbias = ro.domcorrection;              // first term: same sum, same order
        else
          for (int pos = rg.ienv; pos <= rg.jenv; pos++) {
            const int x = sq.code(pos - 1);
            float v = ro.n2log[0];
#pragma unroll
            for (int c = 1; c < NCODE; c++) v = (x == c) ? ro.n2log[c] : v;
            seqbias += v;
          }
        any = true;
        ndom++;
      }
    } else {
      while (mk <= rg.multi - 1 && mk < mr1) { seqbias = add_region(mk, seqbias); any = true; mk++; }
      if (ro.ok) ndom++;
    }
  }
  while (mk < mr1) { seqbias = add_region(mk, seqbias); mk++; }
  if (ndom == 0) return;
  seqbias = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + (double)seqbias));
  float seq_score = (float)((double)(po.fwdsc - (nullsc + seqbias)) / kLn2);
  float sum_score = 0.0f, sbias = 0.0f; int Ldsum = 0;
  for (int d = 0; d < nd_all; d++) {
    const RegionOut ro = a.rout[a.upos[g0 + d]];
    if (!ro.ok) continue;
    const RegionRec rg = a.regions[g0 + d];
    const float dc = env_dc(rg, ro);
    if (ro.envsc - dc > 0.0f) { sum_score += ro.envsc; Ldsum += rg.jenv - rg.ienv + 1; sbias += dc; }
  }
  sbias = flogsum_dev(a.flogsum, 0.0f, (float)(log_omega + (double)sbias));
  sum_score = (float)((double)sum_score + (double)(L - Ldsum) * lt.lognn3);
  sum_score = (float)((double)(sum_score - (nullsc + sbias)) / kLn2);
  float final_bias = seqbias;
  if (Ldsum > 0 && sum_score > seq_score) { seq_score = sum_score; final_bias = sbias; }
  const int seq_rep = ((double)seq_score >= a.T);
  int k = 0;
  for (int d = 0; d < nd_all; d++) {
    const RegionOut ro = a.rout[a.upos[g0 + d]];
    const RegionRec rg = a.regions[g0 + d];
    emit_domain(a, g0 + d, pr, pp, rg, ro, ro.ok ? env_dc(rg, ro) : 0.0f, ro.ok ? k : -1, ndom, po.flags, nullsc, seq_score, final_bias,
                seq_rep, L, lt.lognn3, log_omega);
    if (ro.ok) k++;
  }
  count_reported(a.domz, (a.usample ? a.usample[a.sorted_uniq[pr.useq]] * a.P : 0) + pr.prof, seq_rep != 0);
}

// pair_region0[pi] = seg_region_start[seg] + (pref[pi] - pref[seg_pair_start[seg]])
__global__ void __launch_bounds__(256) k_region_offsets(int64_t npairs, const PairRec *__restrict__ pairs, const int32_t *__restrict__ pref,
                                                        const int64_t *__restrict__ seg_pair_start, const int64_t *__restrict__ seg_region_start,
                                                        int64_t *__restrict__ pair_region0)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs) return;
  const int seg = pairs[pi].prof;
  if (seg < 0) { pair_region0[pi] = 0; return; }
  pair_region0[pi] = seg_region_start[seg] + (int64_t)(pref[pi] - pref[seg_pair_start[seg]]);
}

// final thresholds + the ItsPosition argmax
__global__ void __launch_bounds__(256) k_finalize(itsx_domain *__restrict__ dom, int64_t n, const int64_t *__restrict__ domz, double domE,
                                                  const int32_t *__restrict__ usample, int P)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  itsx_domain d = dom[i];
  if (d.dom_idx < 0) return;
  const int rep = d.seq_reported && (det_exp(d.lnP) * (double)domz[(usample ? usample[d.rep] * P : 0) + d.prof] <= domE);
  dom[i].dom_reported = rep;
}
// rank_key (k_api.h) = [24b tenths+bias][20b ~prof][20b ~dom]; atomicMax picks the highest %.1f score, then the earliest profile,
// then the earliest domain -- ItsPosition._score's "first strictly greater".  The winner's coordinate is fetched by a second
// pass (k_position_coords): (profile, domain index) is unique per representative, so exactly one row carries the winning key.
__global__ void __launch_bounds__(256) k_positions(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ side /*[P] 1 left 2 right*/,
                                                   unsigned long long *__restrict__ best_l, unsigned long long *__restrict__ best_r,
                                                   int32_t *__restrict__ in_ddict)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || d.dom_reported != 1) return;
  in_ddict[d.rep] = 1;
  const int sd = side[d.prof];
  if (sd == 0) return;
  atomicMax(sd == 1 ? &best_l[d.rep] : &best_r[d.rep], rank_key(d));
}
__global__ void __launch_bounds__(256) k_position_coords(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ side,
                                                         const unsigned long long *__restrict__ best_l, const unsigned long long *__restrict__ best_r,
                                                         int32_t *__restrict__ cl, int32_t *__restrict__ cr)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || d.dom_reported != 1) return;
  const int sd = side[d.prof];
  if (sd == 0) return;
  const unsigned long long k = rank_key(d);
  if (sd == 1) { if (k == best_l[d.rep]) cl[d.rep] = d.jenv; }
  else if (k == best_r[d.rep]) cr[d.rep] = d.ienv;
}
// ---- row compaction (ITSX_COMPACT_ROWS=1): per (representative, 2-character profile prefix) only the rows that can still
// win ItsPosition's argmax once the dataset-wide domZ is known are kept.  A row is CERTAIN to be reported when its target is
// reported and exp(lnP) * Zmax <= domE_min (lnP <= lnp_certain); below the best certain row of its class nothing can win any
// more, and nothing below it changes the "sequence has a row" flag either.
DEV unsigned long long compact_key(const itsx_domain &d) { return rank_key(d); }       // [tenths][~profile][~domain]
__global__ void __launch_bounds__(256) k_compact_best(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                      double lnp_certain, unsigned long long *__restrict__ bestc)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || !d.seq_reported || !(d.lnP <= lnp_certain)) return;
  atomicMax(&bestc[(size_t)d.rep * ncls + cls[d.prof]], compact_key(d));
}
__global__ void __launch_bounds__(256) k_compact_mark(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                      double lnp_certain, const unsigned long long *__restrict__ bestc, int32_t *__restrict__ keep)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int k = 0;
  if (i < n) {
    const itsx_domain d = dom[i];
    if (d.dom_idx >= 0 && d.seq_reported) {                 // a row of an unreported target is never reported
      const unsigned long long b = bestc[(size_t)d.rep * ncls + cls[d.prof]], key = compact_key(d);
      const bool certain = d.lnP <= lnp_certain;
      k = (certain ? key == b : key > b) || (d.flags & 2);  // (rows at the region cap stay: the parity-risk counters read them)
    }
  }
  keep[i] = k;
}
__global__ void __launch_bounds__(256) k_compact_scatter(const itsx_domain *__restrict__ dom, int64_t n, const int32_t *__restrict__ keep,
                                                         const int32_t *__restrict__ pos, itsx_domain *__restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) out[pos[i]] = dom[i];
}
void launch_compact_best(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, double lnp_certain, unsigned long long *bestc, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_compact_best, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, lnp_certain, bestc);
}
void launch_compact_mark(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, double lnp_certain, const unsigned long long *bestc, int32_t *keep, hipStream_t st)
{
  hipLaunchKernelGGL(k_compact_mark, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, lnp_certain, bestc, keep);
}
void launch_compact_scatter(const itsx_domain *dom, int64_t n, const int32_t *keep, const int32_t *pos, itsx_domain *out, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_compact_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, keep, pos, out);
}

// parity-risk bookkeeping, after k_positions: uflag[rep] bit 0 = the winning left/right domain carries flags bit 0 (its
// region is one hmmsearch resolves by stochastic clustering), bit 1 = a pair of either side hit the region cap
__global__ void __launch_bounds__(256) k_position_flags(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ side,
                                                        const unsigned long long *__restrict__ best_l, const unsigned long long *__restrict__ best_r,
                                                        int32_t *__restrict__ uflag)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || d.dom_reported != 1) return;
  const int sd = side[d.prof];
  if (sd == 0) return;
  int f = 0;
  if ((d.flags & 1) && rank_key(d) == (sd == 1 ? best_l[d.rep] : best_r[d.rep])) f |= 1;
  if (d.flags & 2) f |= 2;
  if (f) atomicOr(&uflag[d.rep], f);
}
// c[0] uniques with bit 0, c[1] reads with bit 0, c[2] uniques with bit 1, c[3] reads with bit 1
__global__ void __launch_bounds__(256) k_count_flags(const int32_t *__restrict__ uflag, int32_t U, const int32_t *__restrict__ uniq_of, int64_t n,
                                                     unsigned long long *__restrict__ c)
{
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t u = t0; u < U; u += stride) { const int f = uflag[u]; a0 += f & 1; a2 += (f >> 1) & 1; }
  for (int64_t r = t0; r < n; r += stride) { const int u = uniq_of[r]; const int f = u >= 0 ? uflag[u] : 0; a1 += f & 1; a3 += (f >> 1) & 1; }
  for (int d = 32; d >= 1; d >>= 1) { a0 += __shfl_down(a0, d, 64); a1 += __shfl_down(a1, d, 64); a2 += __shfl_down(a2, d, 64); a3 += __shfl_down(a3, d, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&c[0], a0); atomicAdd(&c[1], a1); atomicAdd(&c[2], a2); atomicAdd(&c[3], a3); }
}

// ---- host launchers -----------------------------------------------------------------------
// lds_pad: untouched dynamic LDS that caps the blocks per CU when the kernel runs beside another stream's (see launch_msv)
void launch_bias(const FloatArgs &a, int64_t npairs, hipStream_t st, int lds_pad)
{
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_bias, dim3((unsigned)((npairs + 255) / 256)), dim3(256), (size_t)lds_pad, st, a, npairs);
}
void launch_filters_fwd(const FloatArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st)
{
  if (nwaves <= 0) return;
  if (generic_q) hipLaunchKernelGGL((k_filters_fwd<0, 0>), dim3(nwaves), dim3(64), 0, st, a, wave0, (float *)nullptr);
  else           hipLaunchKernelGGL((k_filters_fwd<QMAX, 0>), dim3(nwaves), dim3(64), 0, st, a, wave0, (float *)nullptr);
}
void launch_fwd_bound(const FloatArgs &a, float *fb, int nwaves, int wave0, int generic_q, hipStream_t st)
{
  if (nwaves <= 0) return;
  if (generic_q) hipLaunchKernelGGL((k_filters_fwd<0, 1>), dim3(nwaves), dim3(64), 0, st, a, wave0, fb);
  else           hipLaunchKernelGGL((k_filters_fwd<QMAX, 1>), dim3(nwaves), dim3(64), 0, st, a, wave0, fb);
}
void launch_bwd_decode(const FloatArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st)
{
  if (nwaves <= 0) return;
  if (generic_q) hipLaunchKernelGGL(k_bwd_decode<0>, dim3(nwaves), dim3(64), 0, st, a, wave0);
  else           hipLaunchKernelGGL(k_bwd_decode<QMAX>, dim3(nwaves), dim3(64), 0, st, a, wave0);
}
void launch_decode(const FloatArgs &a, int nwaves, int wave0, hipStream_t st)
{
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_decode, dim3(nwaves), dim3(64), 0, st, a, wave0);
}
void launch_envelopes(const EnvArgs &a, int nwaves, int wave0, int generic_q, hipStream_t st)
{
  if (nwaves <= 0) return;
  if (generic_q) {
    hipLaunchKernelGGL(k_env_fwd<0>, dim3(nwaves), dim3(64), 0, st, a, wave0);
    hipLaunchKernelGGL(k_env_bwd<0>, dim3(nwaves), dim3(64), 0, st, a, wave0);
    hipLaunchKernelGGL(k_env_post<0>, dim3(nwaves), dim3(64), 0, st, a, wave0);
  } else {
    hipLaunchKernelGGL(k_env_fwd<QMAX>, dim3(nwaves), dim3(64), 0, st, a, wave0);
    hipLaunchKernelGGL(k_env_bwd<QMAX>, dim3(nwaves), dim3(64), 0, st, a, wave0);
    hipLaunchKernelGGL(k_env_post<QMAX>, dim3(nwaves), dim3(64), 0, st, a, wave0);
  }
}
void launch_score(const ScoreArgs &a, hipStream_t st)
{
  if (a.npairs <= 0) return;
  hipLaunchKernelGGL(k_score, dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, st, a);
}
void launch_region_offsets(int64_t npairs, const PairRec *pairs, const int32_t *pref, const int64_t *seg_pair_start,
                           const int64_t *seg_region_start, int64_t *pair_region0, hipStream_t st)
{
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_region_offsets, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, npairs, pairs, pref, seg_pair_start, seg_region_start, pair_region0);
}
void launch_finalize(itsx_domain *dom, int64_t n, const int64_t *domz, double domE, const int32_t *usample, int P, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, domz, domE, usample, P);
}
void launch_positions(const itsx_domain *dom, int64_t n, const int8_t *side, unsigned long long *bl, unsigned long long *br,
                      int32_t *in_ddict, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_positions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, side, bl, br, in_ddict);
}

void launch_position_coords(const itsx_domain *dom, int64_t n, const int8_t *side, const unsigned long long *bl, const unsigned long long *br,
                            int32_t *cl, int32_t *cr, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_position_coords, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, side, bl, br, cl, cr);
}
void launch_position_flags(const itsx_domain *dom, int64_t n, const int8_t *side, const unsigned long long *bl, const unsigned long long *br,
                           int32_t *uflag, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_position_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, side, bl, br, uflag);
}
void launch_count_flags(const int32_t *uflag, int32_t U, const int32_t *uniq_of, int64_t n, unsigned long long *c, hipStream_t st)
{
  hipLaunchKernelGGL(k_count_flags, dim3(1024), dim3(256), 0, st, uflag, U, uniq_of, n, c);
}

}  // namespace itsx
