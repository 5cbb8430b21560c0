// k_lazy.hip -- the lazy domain stage: only (representative, profile) pairs that can still win ItsPosition's argmax go
// through Backward, decoding, envelope re-scoring and the ensemble stage.
//
// What the consumer reads.  ItsPosition.parse/_score (itsxpress/SeqSample.py:400-461) keeps, per target sequence and side
// (profile NAME prefix), the FIRST domain row with the strictly greatest %.1f score; rows below the winner are thrown away.
// hmmsearch computes every row all the same (itsxpress/SeqSample.py:191-209); on amplicon data ~130 profiles of a side model
// the same flank, so 98.6 % of the rows lose.
//
// The bound.  For a pair whose multihit Forward score over the whole target is fwdsc (nats), every domain's bit score obeys
//     bits <= (fwdsc - nullsc) / ln 2 + C(n),   C(n) = 1 + (2 ln(2 (n + 3) / (3 (n + 2))) + n ln((n + 3) / (n + 2))) / ln 2  (~1.27)
// (scripts/lazy_bound.py derives it: every path of the envelope's unihit Forward sum is a path of the multihit sum with its
// flanks in N and C, and dombias >= 0; measured slack 1.9-2.6 bits, 0 violations).  LenTables::lazy_c holds C(n) plus a margin
// of 0.02 bits for float rounding (both sides are sums of positive products: relative error <= ~3 (n + M) 2^-24).
//
// The schedule (engine.hip: lazy_rounds), per chunk of representatives:
//   pass A   Forward score of EVERY pair past the MSV filter (k_filters_fwd<., 1>: nothing stored but the score);
//   round 1  per (representative, class) the pair with the best bound goes through the whole domain pipeline;
//   round 2  every pair whose bound, as %.1f tenths, reaches the group's best CERTAIN row so far (k_compact_best: reported for
//            every domZ the data set can have) -- ties included, they go to the EARLIER row -- and every pair of a group that
//            has no certain row.  The winner of the full table is among them, and so is every row that ties with it.
// Exactness of the thresholds: hmmsearch's domZ (reported targets per profile) is not known when pairs are skipped, only
// bounds on it -- the reported targets among the evaluated pairs below, the pairs past the MSV filter above.  k_finalize_lazy
// decides every row that both bounds decide alike; a row that neither decides and that could change a winner or the "sequence
// has a row" flag is counted (k_lazy_pending), and the search is then repeated in full (engine.hip: itsx_search_finalize).
#include "engine.h"
#include "k_api.h"
#include "detmath.h"
#include "k_vec.h"

namespace itsx {

// =========================================================================================
// Pass A's kernel: the Forward score of one (representative, profile) pair per lane, as a BOUND -- the same sum over paths as
// p7_ForwardParser (k_filters_fwd), but not HMMER's arithmetic: nodes in their natural order 1..46 instead of the SSE striping
// (whose four D->D passes per row exist only because of the striping), fused multiply-adds, rescaling at 1e20 instead of 1e4.
// The result differs from HMMER's float by rounding only (relative ~1e-5, far inside LenTables::lazy_c's margin) and is used
// for nothing but the selection; the pairs that matter get HMMER's own arithmetic in the domain pipeline.
//   * lane = pair, wave = 64 pairs of one profile (the work list's grouping), state M, I, D of 46 nodes in 69 register pairs,
//     pair j = nodes (2j + 1, 2j + 2): every operation on M and I is a packed one over two adjacent nodes;
//   * a node's contribution to the NEXT node's match cell, S[k] = M[k] t(M_k->M_k+1) + I[k] t(I_k->M_k+1) + D[k] t(D_k->M_k+1), is
//     formed on the aligned pairs and shifted by one node with a single register move per pair (the transitions are stored by
//     their source node for that: BoundTab);
//   * the D->D chain is 2 scalar FMAs per pair, sequential, as the recurrence is;
//   * the profile's transitions are wave-uniform: 64 B per pair of nodes through scalar loads, double-buffered by hand (as in
//     k_float.hip: left alone, the scheduler hoists a row's loads and spills SGPRs); emission odds by node in LDS (3 KB).
// ~13 packed instructions per pair of nodes, ~320 per row against the striped kernel's 533.
constexpr int BP = BOUND_PAIRS;                     // 23 pairs of nodes: models of up to 46 nodes (engine.h: MMAX)
typedef const f4 __attribute__((address_space(4))) *cf4q;
// record j of a profile's table (engine.hip: install_profiles), 16 floats: what pair j needs EARLY -- its transitions out of the
// nodes (mm im dm: into the next node's match cell; mi ii) -- and what pair j - 1 needs LATE (bm md dd dd): one record per step
struct BT { f2 mm, im, dm, mi, ii, bm, md; float dd1, dd2; };
DEV BT ldbt(const float *tab, int j)
{
  const f4 a = *(cf4q)(uintptr_t)(tab + j * 16), b = *(cf4q)(uintptr_t)(tab + j * 16 + 4), c = *(cf4q)(uintptr_t)(tab + j * 16 + 8),
           d = *(cf4q)(uintptr_t)(tab + j * 16 + 12);
  BT t;
  // (what the folded recurrences read comes first: 12 floats, one dwordx8 + one dwordx4; mi and md are the plain kernel's alone)
  t.mm = (f2){a.x, a.y}; t.im = (f2){a.z, a.w}; t.dm = (f2){b.x, b.y}; t.ii = (f2){b.z, b.w};
  t.bm = (f2){c.x, c.y}; t.dd1 = c.z; t.dd2 = c.w; t.mi = (f2){d.x, d.y}; t.md = (f2){d.z, d.w};
  return t;
}
DEV f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// SHARE (k_share.hip): the wave's pairs are chains that start at the same depth of their length's prefix tree -- the rows before
// come from the state the parent chain saved for this profile (FWD_STATE_Q float4: M, I, D of the 46 nodes, xN xJ xC xB, the scale's
// logarithm), and a chain saves its own state after row d * B where a later chain branches off.  The same operations on the same
// operands in the same order as the chain's unshared run: the scores are bitwise equal (tests/test_gpu_share.py).
// A block is FWD_WPB consecutive waves of the work list -- the same profile but at a run's end -- so that the waves a CU holds at one time
// read one or two profiles' transition tables through its scalar cache, not eight (SQC_DCACHE_MISSES: 5e)
#ifndef ITSX_FWD_WPB
#define ITSX_FWD_WPB 1
#endif
constexpr int FWD_WPB = ITSX_FWD_WPB;
// FOLD (engine.hip: install_profiles builds the table for it; every profile whose interior M->D transitions are all > 0, i.e. every
// profile hmmbuild writes): the delete cells are kept DIVIDED by s_k = t(M_k-1 -> D_k) / g_k-1 and the match cells MULTIPLIED by
// g_k = 1 + the share of M_k that the deletes after it carry into E -- constants of the profile, folded into its transitions and
// emission odds on the host.  The same sum over paths term by term, with two operations per node fewer: a new delete cell is ONE
// multiply-add (D^_k+1 = D^_k a_k + M~_k: no product M_k t(M_k -> D_k+1) first), and the row's E is the sum of the match cells alone
// (no second accumulation of the deletes).  The insert cells are kept divided by r_k = t(M_k -> I_k) / g_k in the same way: I^_k' = I^_k t(I_k -> I_k)
// + M~_k, one multiply-add instead of a product and a multiply-add.  8 packed + 2 plain instructions per pair of nodes instead of 11 + 2.
template <bool SHARE, bool FOLD>
__global__ void __launch_bounds__(64 * FWD_WPB, 2) k_fwd_bound(FloatArgs a, int wave0, int nwaves, const float *__restrict__ btab, float *__restrict__ fb, ShareLaunch sl)
{
  const int wv = threadIdx.x >> 6, widx = blockIdx.x * FWD_WPB + wv;
  WaveDesc wd = a.waves[wave0 + min(widx, nwaves - 1)];
  if (widx >= nwaves) { wd.count = 0; wd.rows = 1; }          // (a block's spare waves: no lane, no row)
  const int lane = threadIdx.x & 63;
  const int prof = uni(wd.prof);
  const DevProfile *pp = a.prof + prof;
  // emission odds by node: en[code][k - 1], k = z Q + q + 1 in the striped table (one copy per wave: a run may end inside the block)
  __shared__ __attribute__((aligned(16))) float en_all[FWD_WPB][NCODE * 2 * BP];
  float *en = en_all[wv];
  const float *tab = btab + (size_t)prof * BOUND_TAB;
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  const int L = pr.L;
  // (two-sided schedule: the chain's record -- one 64-byte load instead of four dependent ones -- and the folded emission table as it
  // stands in memory; each wave fills and reads its OWN copy, so the wave's own order of LDS operations is all the fence it needs)
  const bool recs = SHARE && sl.chain != nullptr;
  ChainRec cr{};
  if (recs) { const f4 *q = (const f4 *)(sl.chain + pr.useq); const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3]; f4 *o = (f4 *)&cr; o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3; }
  if (FOLD && recs && sl.entab) {
    const f4 *e4 = (const f4 *)(sl.entab + (size_t)prof * (NCODE * 2 * BP));
    for (int i = lane; i < NCODE * 2 * BP / 4; i += 64) ((f4 *)en)[i] = e4[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    const int Q = uni(pp->Q);
    const float *g = tab + (BP + 1) * 16;                      // FOLD: the match cells' scale by node
    for (int i = lane; i < NCODE * 2 * BP; i += 64) {
      const int x = i / (2 * BP), k0 = i % (2 * BP);           // node k0 + 1
      const float e = (k0 < 4 * Q) ? pp->rf[(x * QMAX + (k0 % Q)) * 4 + k0 / Q] : 0.0f;
      en[i] = FOLD ? e * g[k0] : e;
    }
    __syncthreads();
  }
  Seq sq;
  if (recs) { sq.w = a.rd.words + cr.woff; sq.exc = a.rd.exc + cr.excoff; sq.nexc = cr.nexc; sq.L = cr.L; }
  else sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  int Lw = wd.rows - 1;
  f2 M[BP], I[BP], D[BP];
#pragma unroll
  for (int j = 0; j < BP; j++) { M[j] = (f2){0.f, 0.f}; I[j] = (f2){0.f, 0.f}; D[j] = (f2){0.f, 0.f}; }
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
  const float ploop = 1.0f - pmove;
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = pmove, xC = 0.f;
  double totscale = 0.0;
  const int row0 = SHARE ? (sl.depth << sl.logB) : 0;       // rows done before this launch (wave-uniform)
  unsigned long long smask = 0ull; int64_t snode = 0;
  // two-sided sharing (round 6): the chain's last row, and the level at which it takes the rest of the sum over paths from a saved
  // Backward state (k_bwd_bound) instead of walking on to L
  int endrow = L, jlev = -1; int64_t jnode = 0;
  if constexpr (SHARE) {
    const int k = pr.useq;
    if (active) {
      if (recs) {
        smask = cr.mask; snode = (int64_t)cr.node0 - sl.node_base;
        endrow = cr.endrow; jlev = pr.xj >= 0 ? cr.jlev : -1; jnode = (int64_t)cr.jsrc - sl.gnode_base;
      } else {
        smask = sl.mask[k]; snode = (int64_t)sl.node0[k] - sl.node_base;
        if (sl.endrow) { endrow = sl.endrow[k]; jlev = pr.xj >= 0 ? sl.jlev[k] : -1; jnode = (int64_t)sl.jsrc[k] - sl.gnode_base; }
      }
    }
    if (sl.dbg & 1) Lw = row0;
    if (sl.dbg & 4) jlev = -1;
    if (sl.depth > 0 && !(sl.dbg & 2)) {
      const int64_t node = (int64_t)(recs ? cr.src : sl.src[k]) - sl.node_base;
      const f4 *src = (const f4 *)sl.slots + (node * sl.Pb + (prof - sl.p0)) * FWD_STATE_Q;
#pragma unroll
      for (int j = 0; j < BP; j++) { const f4 v = src[j]; M[j] = (f2){v.x, v.y}; I[j] = (f2){v.z, v.w}; }
#pragma unroll
      for (int j = 0; j < BP / 2; j++) { const f4 v = src[BP + j]; D[2 * j] = (f2){v.x, v.y}; D[2 * j + 1] = (f2){v.z, v.w}; }
      const f4 u = src[BP + BP / 2], t = src[BP + BP / 2 + 1];
      D[BP - 1] = (f2){u.x, u.y}; xN = u.z; xJ = u.w; xC = t.x; xB = t.y;
      totscale = __builtin_bit_cast(double, (f2){t.z, t.w});
    }
  }
  SeqStream ss; ss.open(sq, row0, +1);
  int xnext = ss.get(row0);
  for (int i = row0 + 1; ; i++) {
    if constexpr (SHARE) {
      // a block boundary: the state after row i - 1 = d * B is what a chain that branches off there starts from
      if (((i - 1) & ((1 << sl.logB) - 1)) == 0) {
        const int d = (i - 1) >> sl.logB;
        if (d < 64 && ((smask >> d) & 1ull) && i - 1 > row0) {           // (a mask has no bit above the chain's last level)
          const int64_t node = snode + __popcll(smask & ((1ull << d) - 1ull));
          f4 *dst = (f4 *)sl.slots + (node * sl.Pb + (prof - sl.p0)) * FWD_STATE_Q;
#pragma unroll
          for (int j = 0; j < BP; j++) dst[j] = (f4){M[j].x, M[j].y, I[j].x, I[j].y};
#pragma unroll
          for (int j = 0; j < BP / 2; j++) dst[BP + j] = (f4){D[2 * j].x, D[2 * j].y, D[2 * j + 1].x, D[2 * j + 1].y};
          const f2 ts = __builtin_bit_cast(f2, totscale);
          dst[BP + BP / 2] = (f4){D[BP - 1].x, D[BP - 1].y, xN, xJ};
          dst[BP + BP / 2 + 1] = (f4){xC, xB, ts.x, ts.y};
        }
        if (d == jlev) {
          // the join: every path crosses the cut after row d * B in exactly one of M_k, I_k, D_k, N, J, C, B -- the score is the inner
          // product of the Forward state with the suffix's Backward state (same layout, dual units: k_bwd_bound), plus both scales
          const f4 *g = (const f4 *)sl.gslots + (jnode * sl.Pb + (prof - sl.p0)) * FWD_STATE_Q;
          f2 sa = (f2){0.f, 0.f}, sb = (f2){0.f, 0.f};
#pragma unroll
          for (int j = 0; j < BP; j++) { const f4 v = g[j]; sa = pfma(M[j], (f2){v.x, v.y}, sa); sb = pfma(I[j], (f2){v.z, v.w}, sb); }
#pragma unroll
          for (int j = 0; j < BP / 2; j++) { const f4 v = g[BP + j]; sa = pfma(D[2 * j], (f2){v.x, v.y}, sa); sb = pfma(D[2 * j + 1], (f2){v.z, v.w}, sb); }
          const f4 u = g[BP + BP / 2], t = g[BP + BP / 2 + 1];
          sa = pfma(D[BP - 1], (f2){u.x, u.y}, sa);
          sb = pfma((f2){xN, xJ}, (f2){u.z, u.w}, sb);
          sa = pfma((f2){xC, xB}, (f2){t.x, t.y}, sa);
          const float dot = (sa.x + sa.y) + (sb.x + sb.y);
          const double gscale = __builtin_bit_cast(double, (f2){t.z, t.w});
          const bool jbad = (dot != dot) || (dot <= 0.0f) || (dot == __builtin_inff());
          fb[pi] = jbad ? __builtin_nanf("") : (float)(totscale + gscale + det_log((double)dot));
        }
      }
    }
    if (i > Lw) break;
    if (i <= endrow) {
      const int x = xnext;
      if (i < L) xnext = ss.get(i);
      const float *ex = en + x * (2 * BP);
      const float *tb = tab + opaque_zero();
      const f2 xBv = (f2){xB, xB};
      f2 acc = (f2){0.f, 0.f};
      float sprev = 0.0f;                       // S of the node before the pair (S[0] = 0)
      // pair 0's early half: S = what its nodes send to the next node's match cell (from the OLD row), and its new insert cells
      BT cur = ldbt(tb, 0);
      f2 ecur = *(const f2 *)(ex);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      f2 Scur = pfma(M[0], cur.mm, pfma(I[0], cur.im, D[0] * cur.dm));
      if constexpr (FOLD) I[0] = pfma(I[0], cur.ii, M[0]); else I[0] = pfma(M[0], cur.mi, I[0] * cur.ii);
      float dprev = 0.0f, mdprev = 0.0f;         // D'[2j] and M'[2j] t(M->D) (FOLD: M'[2j] itself) of the node before the pair (none before node 1)
      f2 Sprev = (f2){0.f, 0.f};                 // S of the pair before (its second node feeds this pair's first; S[0] = 0)
      f2 dlast = (f2){0.f, 0.f};                 // the pair before's new delete cells, not yet in the row's sum
      cur = ldbt(tb, 1);
      // One step = the late half of pair j (8 instructions, a chain: sh -> w -> mn -> md -> dn.y) and the early half of pair j + 1 (5,
      // independent of it).  At two waves per SIMD an instruction that reads the result of the one before it costs a wait state that the
      // other wave only half fills (profiles/round5_valu_issue.md: an s_nop holds the SIMD 0.9 ns at this occupancy; the compiler's own
      // order had 4 per step, 11 % of the row).  The order is therefore fixed by hand -- every statement fenced -- so that no instruction
      // follows its producer directly: the early half's instructions sit between the links of the chain, and the new delete cells go
      // into the row's sum one step later.
#define SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
      for (int j = 0; j < BP; j++) {
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): record j + 1 and pair j's emissions, requested one step ago
        const bool more = j + 1 < BP;
        BT nxt = cur; f2 enxt = ecur;
        if (more) { nxt = ldbt(tb, j + 2); enxt = *(const f2 *)(ex + 2 * (j + 1)); }
        SB;
        const f2 sh = (f2){Sprev.y, Scur.x}; SB;
        f2 t = (f2){0.f, 0.f}, u = t, Sn = Scur;
        if (more) { t = D[j + 1] * cur.dm; SB; }
        // both new delete cells of the pair are born in this step (its old ones went into Scur one step ago): dd1 = D_2j -> D_2j+1 (0
        // for j = 0), dd2 = D_2j+1 -> D_2j+2
        // (as inline assembly: left to itself the compiler accumulates into the dying product's register -- v_fmac -- and then
        // moves the result into the pair, twice per pair of nodes and row)
        f2 dn;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(dn.x) : "v"(dprev), "s"(cur.dd1), "v"(mdprev)); SB;
        const f2 w = pfma(xBv, cur.bm, sh); SB;
        if (more) { t = pfma(I[j + 1], cur.im, t); SB; if constexpr (!FOLD) { u = I[j + 1] * cur.ii; SB; } }
        const f2 mn = w * ecur; SB;
        if (more) { Sn = pfma(M[j + 1], cur.mm, t); SB; }
        if constexpr (FOLD) {
          acc = acc + mn; SB;
          if (more) { I[j + 1] = pfma(I[j + 1], cur.ii, M[j + 1]); SB; }      // (the insert cells divided by t(M_k -> I_k) / g_k)
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(dn.y) : "v"(dn.x), "s"(cur.dd2), "v"(mn.x)); SB;
          mdprev = mn.y;
        } else {
          const f2 md = mn * cur.md; SB;
          acc = acc + mn; SB;
          if (more) { I[j + 1] = pfma(M[j + 1], cur.mi, u); SB; }
          acc = acc + dlast; SB;
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(dn.y) : "v"(dn.x), "s"(cur.dd2), "v"(md.x)); SB;
          mdprev = md.y; dlast = dn;
        }
        M[j] = mn; D[j] = dn;
        dprev = dn.y;
        Sprev = Scur; Scur = Sn; cur = nxt; ecur = enxt;
      }
#undef SB
      if constexpr (!FOLD) acc = acc + dlast;
      xE = acc.x + acc.y;
      xN = xN * ploop;
      xC = __builtin_fmaf(xC, ploop, xE * 0.5f);
      xJ = __builtin_fmaf(xJ, ploop, xE * 0.5f);
      xB = (xJ + xN) * pmove;
      // (1e20; at 1e28 -- most targets never get here, a 45-node domain multiplies E by up to ~1e25 -- pass A takes the same 1.600 s per
      // 10 M reads: ITSX_BOUND_RESCALE_EXP)
      if (xE > sl.rescale) {
        const float r = 1.0f / xE;
        xN *= r; xC *= r; xJ *= r; xB *= r;
        const f2 rv = (f2){r, r};
#pragma unroll
        for (int j = 0; j < BP; j++) { M[j] = M[j] * rv; I[j] = I[j] * rv; D[j] = D[j] * rv; }
        totscale += det_log((double)xE);
        xE = 1.0f;
      }
    }
  }
  const bool bad = (xC != xC) || (xC == 0.0f) || (xC == __builtin_inff());
  if (active && jlev < 0) fb[pi] = bad ? __builtin_nanf("") : (float)(totscale + det_log((double)(xC * pmove)));
}

// =========================================================================================
// Round 6, two-sided sharing: the BACKWARD half of pass A's sum over paths.  One Forward row is a linear map of the row state
// (M~, I^, D^ of 46 nodes, N, J, C, B: k_fwd_bound's folded units); the score is a linear functional of the last row's state (C times
// the move out).  Pulled back through the rows L, L - 1, ..., j B + 1 -- the TRANSPOSED maps, applied in that order -- it becomes a
// vector gamma_j with   score = < Forward state after row j B, gamma_j >   for EVERY prefix, and gamma_j depends on the profile, the
// target's length and the residues after row j B only: the uniques of one length that end alike share it.  This kernel walks a
// Backward chain (lane = chain x profile, wave = 64 chains of one (batch, blocks-from-the-end) segment) and saves gamma at the block
// boundaries where somebody joins (k_share.hip: k_join_*), in k_fwd_bound's state layout, so that the join is 72 packed multiply-adds.
// The transposed row, with a' = adjoint of the new row's cell and x the row's residue:
//     nJ = gJ + move gB,  nN = gN + move gB,  aE = (nJ + gC) / 2
//     aD_k = gD_k + dd_k aD_k+1,   aM_k = gM_k + aE + aD_k+1,   w_k = e_k(x) aM_k                  (k = 46 .. 1, aD_47 = w_47 = 0)
//     gM_k <- mm_k w_k+1 + gI_k,   gI_k <- im_k w_k+1 + ii_k gI_k,   gD_k <- dm_k w_k+1,   gB <- sum_k bm_k w_k
//     gN <- loop nN,  gJ <- loop nJ,  gC <- loop gC
// (the same table entries as the Forward kernel's, by pair of nodes: engine.hip builds rtab).  The cells are kept below ~1e6 by a sum test
// every 16 rows (block boundaries are multiples of 16), the scale's logarithm travels with the state like Forward's.
// record j of the table (engine.hip: install_profiles): mm im dm bm of pair j, then ii and dd (aa) of pair j - 1 -- what ONE step of the
// row loop below reads: the late half of pair j and the early half of pair j - 1
struct RT { f2 mm, im, dm, bm, ii, aa; };
DEV RT ldrt(const float *tab, int j)
{
  const f4 a = *(cf4q)(uintptr_t)(tab + j * 12), b = *(cf4q)(uintptr_t)(tab + j * 12 + 4), c = *(cf4q)(uintptr_t)(tab + j * 12 + 8);
  RT t;
  t.mm = (f2){a.x, a.y}; t.im = (f2){a.z, a.w}; t.dm = (f2){b.x, b.y}; t.bm = (f2){b.z, b.w}; t.ii = (f2){c.x, c.y}; t.aa = (f2){c.z, c.w};
  return t;
}
__global__ void __launch_bounds__(64 * FWD_WPB, 2) k_bwd_bound(FloatArgs a, int wave0, int nwaves, const float *__restrict__ btab, const float *__restrict__ rtab, ShareLaunch sl)
{
  const int wv = threadIdx.x >> 6, widx = blockIdx.x * FWD_WPB + wv;
  WaveDesc wd = a.waves[wave0 + min(widx, nwaves - 1)];
  if (widx >= nwaves) { wd.count = 0; wd.rows = 1; }
  const int lane = threadIdx.x & 63;
  const int prof = uni(wd.prof);
  const DevProfile *pp = a.prof + prof;
  __shared__ __attribute__((aligned(16))) float en_all[FWD_WPB][NCODE * 2 * BP];
  float *en = en_all[wv];
  const float *tab = rtab + (size_t)prof * BOUND_RTAB;
  (void)pp; (void)btab;
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  const int L = pr.L;
  ChainRec cr;
  { const f4 *q = (const f4 *)(sl.chain + pr.useq); const f4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3]; f4 *o = (f4 *)&cr; o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3; }
  {
    const f4 *e4 = (const f4 *)(sl.entab + (size_t)prof * (NCODE * 2 * BP));
    for (int i = lane; i < NCODE * 2 * BP / 4; i += 64) ((f4 *)en)[i] = e4[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  Seq sq; sq.w = a.rd.words + cr.woff; sq.exc = a.rd.exc + cr.excoff; sq.nexc = cr.nexc; sq.L = cr.L;
  const int nsteps = (sl.dbg & 32) ? 0 : wd.rows - 1;         // the wave's longest chain
  f2 M[BP], I[BP], D[BP];
#pragma unroll
  for (int j = 0; j < BP; j++) { M[j] = (f2){0.f, 0.f}; I[j] = (f2){0.f, 0.f}; D[j] = (f2){0.f, 0.f}; }
  const float pmove = (2.0f + 1.0f) / ((float)L + 2.0f + 1.0f);
  const float ploop = 1.0f - pmove;
  float xN = 0.f, xJ = 0.f, xB = 0.f, xC = pmove;             // gamma after the last row: the score is C times the move out
  double totscale = 0.0;
  const int kb = pr.useq;
  const int rd = sl.depth;                                    // blocks from the end the launch's chains start at (wave-uniform)
  unsigned long long smask = 0ull; int64_t snode = 0; int mysteps = 0;
  (void)kb;
  if (active) { smask = cr.mask; snode = (int64_t)cr.node0 - sl.node_base; mysteps = cr.endrow; }
  if (rd > 0 && !(sl.dbg & 16)) {
    const int64_t node = (int64_t)cr.src - sl.node_base;
    const f4 *src = (const f4 *)sl.slots + (node * sl.Pb + (prof - sl.p0)) * FWD_STATE_Q;
#pragma unroll
    for (int j = 0; j < BP; j++) { const f4 v = src[j]; M[j] = (f2){v.x, v.y}; I[j] = (f2){v.z, v.w}; }
#pragma unroll
    for (int j = 0; j < BP / 2; j++) { const f4 v = src[BP + j]; D[2 * j] = (f2){v.x, v.y}; D[2 * j + 1] = (f2){v.z, v.w}; }
    const f4 u = src[BP + BP / 2], t = src[BP + BP / 2 + 1];
    D[BP - 1] = (f2){u.x, u.y}; xN = u.z; xJ = u.w; xC = t.x; xB = t.y;
    totscale = __builtin_bit_cast(double, (f2){t.z, t.w});
  }
  const int A = (L + (1 << sl.logB) - 1) >> sl.logB;
  const int top = (A - rd) << sl.logB;                        // the row the chain's state stands after (virtual when rd = 0 and L is no multiple of B)
  const int first = top < L ? top : L;
  SeqStream ss; ss.open(sq, first > 0 ? first - 1 : 0, -1);
  int xnext = first > 0 ? ss.get(first - 1) : 0;
  for (int step = 0; ; step++) {
    if ((step & ((1 << sl.logB) - 1)) == 0 && step > 0) {
      // a block boundary: gamma after row top - step = (A - rdl) B is what the Forward chains that end there take
      const int rdl = rd + (step >> sl.logB);
      if (active && rdl < 64 && ((smask >> rdl) & 1ull) && !(sl.dbg & 8)) {
        const int64_t node = snode + __popcll(smask & ((1ull << rdl) - 1ull));
        f4 *dst = (f4 *)sl.slots + (node * sl.Pb + (prof - sl.p0)) * FWD_STATE_Q;
#pragma unroll
        for (int j = 0; j < BP; j++) dst[j] = (f4){M[j].x, M[j].y, I[j].x, I[j].y};
#pragma unroll
        for (int j = 0; j < BP / 2; j++) dst[BP + j] = (f4){D[2 * j].x, D[2 * j].y, D[2 * j + 1].x, D[2 * j + 1].y};
        const f2 ts = __builtin_bit_cast(f2, totscale);
        dst[BP + BP / 2] = (f4){D[BP - 1].x, D[BP - 1].y, xN, xJ};
        dst[BP + BP / 2 + 1] = (f4){xC, xB, ts.x, ts.y};
      }
    }
    if (step >= nsteps) break;
    const int row = top - step;                               // this step pulls gamma back through row `row`
    if (row <= L && step < mysteps) {
      const int x = xnext;
      if (row > 1) xnext = ss.get(row - 2);
      const float *ex = en + x * (2 * BP);
      const float *tb = tab + opaque_zero();
      const float gb0 = xB * pmove;
      const float nJ = xJ + gb0, nN = xN + gb0;
      const float aE = 0.5f * (nJ + xC);
      const f2 aEv = (f2){aE, aE};
      float aDn = 0.0f, wn = 0.0f;                            // aD and w of the node after the pair
      f2 accB = (f2){0.f, 0.f};
      // One step = the LATE half of pair j (w = e aM, its share of B, the new M I D of the pair: a chain w -> wsh -> cells) and the EARLY half
      // of pair j - 1 (its aD chain -- two dependent multiply-adds --, aM, ii I), statement by statement in a fixed order so that no
      // instruction reads the result of the one before it (at two waves per SIMD that costs a wait state the other wave only half fills:
      // the compiler's own order ran at 0.87 ns per wave-row against the Forward step's 0.56); the step's table record (engine.hip: what
      // both halves read, 12 floats) and emission odds are requested one step ahead, as in the Forward step.
#define SB __builtin_amdgcn_sched_barrier(0)
      const f4 tl = *(cf4q)(uintptr_t)(tb + BP * 12);         // ii, dd of the last pair
      RT cur = ldrt(tb, BP - 1);
      f2 ecur = *(const f2 *)(ex + 2 * (BP - 1));
      // the early half of the last pair (no node after it: aDn = 0)
      __builtin_amdgcn_s_waitcnt(0xC07F);
      float aD2 = __builtin_fmaf(tl.w, aDn, D[BP - 1].y);
      f2 pre = M[BP - 1] + aEv;
      float aD1 = __builtin_fmaf(tl.z, aD2, D[BP - 1].x);
      f2 iig = (f2){tl.x, tl.y} * I[BP - 1];
      f2 aM = pre + (f2){aD2, aDn};
      aDn = aD1;
#pragma unroll
      for (int j = BP - 1; j >= 0; j--) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): step j's record and pair j's emissions, requested one step ago
        const RT t = cur;
        const f2 e = ecur;
        const bool more = j > 0;
        if (more) { cur = ldrt(tb, j - 1); ecur = *(const f2 *)(ex + 2 * (j - 1)); }
        SB;
        const f2 w = aM * e; SB;
        float aD2n = 0.f, aD1n = 0.f; f2 pren = (f2){0.f, 0.f}, iign = pren, aMn = pren;
        if (more) { aD2n = __builtin_fmaf(t.aa.y, aDn, D[j - 1].y); SB; }
        accB = pfma(t.bm, w, accB); SB;
        if (more) { pren = M[j - 1] + aEv; SB; }
        const f2 wsh = (f2){w.y, wn}; SB;                     // w of the node after each of the pair's two
        if (more) { aD1n = __builtin_fmaf(t.aa.x, aD2n, D[j - 1].x); SB; }
        const f2 gi = I[j];
        M[j] = pfma(t.mm, wsh, gi); SB;
        if (more) { iign = t.ii * I[j - 1]; SB; }
        I[j] = pfma(t.im, wsh, iig); SB;
        if (more) { aMn = pren + (f2){aD2n, aDn}; SB; }
        D[j] = t.dm * wsh; SB;
        wn = w.x; aDn = aD1n; aM = aMn; iig = iign;
      }
#undef SB
      xB = accB.x + accB.y;
      xN = ploop * nN; xJ = ploop * nJ; xC = ploop * xC;
    }
    if ((step & 15) == 15) {
      f2 sm = (f2){0.f, 0.f};
#pragma unroll
      for (int j = 0; j < BP; j++) sm = sm + ((M[j] + I[j]) + D[j]);
      const float tot = (sm.x + sm.y) + ((xN + xJ) + (xC + xB));
      if (tot > 1e6f || (tot < 1e-6f && tot > 0.0f)) {
        const float r = 1.0f / tot;
        const f2 rv = (f2){r, r};
#pragma unroll
        for (int j = 0; j < BP; j++) { M[j] = M[j] * rv; I[j] = I[j] * rv; D[j] = D[j] * rv; }
        xN *= r; xJ *= r; xC *= r; xB *= r;
        totscale += det_log((double)tot);
      }
    }
  }
}
void launch_bwd_bound_share(const FloatArgs &a, const float *btab, const float *rtab, int nwaves, int wave0, const ShareLaunch &sl, hipStream_t st)
{
  if (nwaves <= 0) return;
  const dim3 g((nwaves + FWD_WPB - 1) / FWD_WPB), b(64 * FWD_WPB);
  hipLaunchKernelGGL(k_bwd_bound, g, b, 0, st, a, wave0, nwaves, btab, rtab, sl);
}

// 1e20 by default (ITSX_BOUND_RESCALE_EXP: another power of ten, for A/B; at most 28 -- the folded cells must stay inside float)
static float bound_rescale()
{
  static const float v = [] { const char *e = sw_get("ITSX_BOUND_RESCALE_EXP"); const double x = e ? atof(e) : 20.0; return (float)pow(10.0, x < 4.0 ? 4.0 : x > 28.0 ? 28.0 : x); }();
  return v;
}
void launch_fwd_bound_seq(const FloatArgs &a, const float *btab, bool fold, float *fb, int nwaves, int wave0, hipStream_t st)
{
  if (nwaves <= 0) return;
  ShareLaunch none{}; none.rescale = bound_rescale();
  const dim3 g((nwaves + FWD_WPB - 1) / FWD_WPB), b(64 * FWD_WPB);
  if (fold) hipLaunchKernelGGL((k_fwd_bound<false, true>), g, b, 0, st, a, wave0, nwaves, btab, fb, none);
  else hipLaunchKernelGGL((k_fwd_bound<false, false>), g, b, 0, st, a, wave0, nwaves, btab, fb, none);
}
void launch_fwd_bound_share(const FloatArgs &a, const float *btab, bool fold, float *fb, int nwaves, int wave0, const ShareLaunch &sl0, hipStream_t st)
{
  if (nwaves <= 0) return;
  ShareLaunch sl = sl0; sl.rescale = bound_rescale();
  const dim3 g((nwaves + FWD_WPB - 1) / FWD_WPB), b(64 * FWD_WPB);
  if (fold) hipLaunchKernelGGL((k_fwd_bound<true, true>), g, b, 0, st, a, wave0, nwaves, btab, fb, sl);
  else hipLaunchKernelGGL((k_fwd_bound<true, false>), g, b, 0, st, a, wave0, nwaves, btab, fb, sl);
}

// largest %.1f tenths (biased) a domain of each pair can print, and the best-bound pair of every (representative, class)
__global__ void __launch_bounds__(256) k_lazy_bound(LazyArgs a)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.NP) return;
  const PairRec pr = a.pairs[i];
  if (pr.prof < 0) { a.b10[i] = 0u; return; }
  if (pr.xj < 0) { a.b10[i] = 0u; a.done[i] = 1; return; }      // ran for its row states only (k_share.hip): never selected
  const float f = a.fb[i];
  const LenTables lt = a.lt[pr.L];
  uint32_t b;
  if (f != f || f == __builtin_inff()) b = (1u << 24) - 1;                  // no usable bound: always evaluated
  else if (f == -__builtin_inff()) b = 1u;
  else {
    const double bits = ((double)f - (double)lt.nullsc) / 0.69314718055994529 + (double)lt.lazy_c;
    double t = __builtin_floor(bits * 10.0 + 0.5) + (double)LAZY_TENTHS_BIAS;      // round half UP: never below rint()
    if (t < 1.0) t = 1.0;
    if (t > (double)((1 << 24) - 1)) t = (double)((1 << 24) - 1);
    b = (uint32_t)t;
  }
  a.b10[i] = b;
  const size_t g = (size_t)pr.useq * a.ncls + a.cls[pr.prof];
  atomicMax(&a.gtop[g], ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
}

// round 0: the best-bound pair of each group.  round 1: everything not yet evaluated that could still beat (or tie with) the
// group's best certain row; all of a group that has none.
__global__ void __launch_bounds__(256) k_lazy_mark(LazyArgs a, int round)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > a.NP) return;
  int f = 0;
  if (i < a.NP) {
    const PairRec pr = a.pairs[i];
    if (pr.prof >= 0 && !a.done[i]) {
      const int c = a.cls[pr.prof];
      if (round == 0) f = (uint32_t)(a.gtop[(size_t)pr.useq * a.ncls + c] & 0xFFFFFFFFull) == 0xFFFFFFFFu - (uint32_t)i;
      else {
        const unsigned long long bc = a.bestc[(size_t)a.sorted_uniq[pr.useq] * a.ncls + c];
        f = bc == 0ull || a.b10[i] >= (uint32_t)(bc >> 40);
      }
      if (f) a.done[i] = 1;
    }
  }
  a.flag[i] = f;
}

// selected pairs keep their profile segment and their order inside it (ascending length): out[seg_new[p] + rank inside p]
__global__ void __launch_bounds__(256) k_lazy_scatter(const PairRec *__restrict__ pairs, int64_t NP, const int32_t *__restrict__ flag,
                                                      const int32_t *__restrict__ pos, const int64_t *__restrict__ seg_old,
                                                      const int64_t *__restrict__ seg_new, PairRec *__restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NP || !flag[i]) return;
  const PairRec pr = pairs[i];
  out[seg_new[pr.prof] + (int64_t)(pos[i] - pos[seg_old[pr.prof]])] = pr;
}

__global__ void __launch_bounds__(256) k_finalize_lazy(itsx_domain *__restrict__ dom, int64_t n, const int64_t *__restrict__ zlb,
                                                       const int64_t *__restrict__ zub, double domE, const int32_t *__restrict__ usample, int P)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0) return;
  int rep = 0;
  if (d.seq_reported) {
    const size_t z = (size_t)(usample ? usample[d.rep] * P : 0) + d.prof;
    const double p = det_exp(d.lnP);
    // this target is reported itself, so the true count is at least 1 and at least the lower bound
    const double lo = (double)(zlb[z] > 1 ? zlb[z] : 1), hi = (double)(zub[z] > 1 ? zub[z] : 1);
    rep = (p * hi <= domE) ? 1 : (p * lo <= domE) ? 2 : 0;
  }
  dom[i].dom_reported = rep;
}

__global__ void __launch_bounds__(256) k_lazy_sure(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                   unsigned long long *__restrict__ sure, int32_t *__restrict__ has)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const itsx_domain d = dom[i];
  if (d.dom_idx < 0 || d.dom_reported != 1) return;
  has[d.rep] = 1;
  atomicMax(&sure[(size_t)d.rep * ncls + cls[d.prof]], rank_key(d));
}
__global__ void __launch_bounds__(256) k_lazy_pending(const itsx_domain *__restrict__ dom, int64_t n, const int8_t *__restrict__ cls, int ncls,
                                                      const unsigned long long *__restrict__ sure, const int32_t *__restrict__ has,
                                                      unsigned long long *__restrict__ count, int32_t *__restrict__ prof_flag,
                                                      uint8_t *__restrict__ uniq_flag, unsigned long long *__restrict__ zneed, const unsigned long long *__restrict__ zsplit,
                                                      double domE)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  if (i < n) {
    const itsx_domain d = dom[i];
    // an undecided row matters when it would beat its group's best sure row, or when the sequence has no sure row at all
    if (d.dom_idx >= 0 && d.dom_reported == 2) c = !has[d.rep] || rank_key(d) > sure[(size_t)d.rep * ncls + cls[d.prof]];
    if (c) {
      prof_flag[d.prof] = 1; uniq_flag[d.rep] = 1;
      // the row is NOT reported as soon as the profile's reported targets exceed domE / P-value: the count that settles it (the top-up
      // round of itsx_search_finalize evaluates that many more of the profile's best pairs before anything is counted in full)
      // ... and it IS reported as soon as the UPPER bound falls to domE / P-value.  zneed[2 p] = the largest such count at or below the
      // profile's split (settled from below: more of its best pairs), zneed[2 p + 1] = the smallest one above it (settled from above:
      // its weakest pairs shown unreported); the split lies between the profile's two bounds (finalize_lazy)
      if (zneed) {
        const double z = domE / det_exp(d.lnP);
        if (z < 9.0e18) {
          const unsigned long long zf = (unsigned long long)z;
          if (zf + 2ull <= zsplit[d.prof]) atomicMax(&zneed[2 * d.prof], zf + 2ull);
          else atomicMin(&zneed[2 * d.prof + 1], zf > 2ull ? zf - 2ull : 0ull);
        } else atomicMin(&zneed[2 * d.prof + 1], (unsigned long long)9.0e18);
      }
    }
  }
  const unsigned long long m = __ballot(c);
  if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(count, (unsigned long long)__builtin_popcountll(m));
}

void launch_lazy_bound(const LazyArgs &a, hipStream_t st)
{
  if (a.NP > 0) hipLaunchKernelGGL(k_lazy_bound, dim3((unsigned)((a.NP + 255) / 256)), dim3(256), 0, st, a);
}
void launch_lazy_mark(const LazyArgs &a, int round, hipStream_t st)
{
  hipLaunchKernelGGL(k_lazy_mark, dim3((unsigned)((a.NP + 1 + 255) / 256)), dim3(256), 0, st, a, round);
}
void launch_lazy_scatter(const PairRec *pairs, int64_t NP, const int32_t *flag, const int32_t *pos, const int64_t *seg_old, const int64_t *seg_new,
                         PairRec *out, hipStream_t st)
{
  if (NP > 0) hipLaunchKernelGGL(k_lazy_scatter, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, st, pairs, NP, flag, pos, seg_old, seg_new, out);
}
void launch_finalize_lazy(itsx_domain *dom, int64_t n, const int64_t *zlb, const int64_t *zub, double domE, const int32_t *usample, int P, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_finalize_lazy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, zlb, zub, domE, usample, P);
}
void launch_lazy_sure(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, unsigned long long *sure, int32_t *has, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_lazy_sure, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, sure, has);
}
void launch_lazy_pending(const itsx_domain *dom, int64_t n, const int8_t *cls, int ncls, const unsigned long long *sure, const int32_t *has,
                         unsigned long long *count, int32_t *prof_flag, uint8_t *uniq_flag, unsigned long long *zneed, const unsigned long long *zsplit, double domE, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_lazy_pending, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, n, cls, ncls, sure, has, count, prof_flag, uniq_flag, zneed, zsplit, domE);
}
// evaluated pairs (done flags) of the listed profiles' segments: cnt[slot] (zeroed); grid (blocks per profile, slots)
__global__ void __launch_bounds__(256) k_topup_count(const uint8_t *__restrict__ done, const int64_t *__restrict__ seg_start, const int32_t *__restrict__ total,
                                                     const int32_t *__restrict__ prof_of_slot, unsigned long long *__restrict__ cnt)
{
  const int sl = blockIdx.y, p = prof_of_slot[sl];
  const int64_t base = seg_start[p], n = total[p];
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += done[base + i] != 0;
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cnt[sl], c);
}
void launch_topup_count(const uint8_t *done, const int64_t *seg_start, const int32_t *total, const int32_t *prof_of_slot, int nslots, unsigned long long *cnt, hipStream_t st)
{
  if (nslots > 0) hipLaunchKernelGGL(k_topup_count, dim3(512, (unsigned)nslots), dim3(256), 0, st, done, seg_start, total, prof_of_slot, cnt);
}

// ---- the top-up round (itsx_search_finalize): the unevaluated pairs of the profiles that have undecided rows, best bound first
// pass 0 (pos == nullptr): flag[i] = the pair belongs to a profile with undecided rows and has not been evaluated; pass 1: its key
// [slot | ~bound] into the compact list at pos[i]
__global__ void __launch_bounds__(256) k_topup_keys(const PairRec *__restrict__ pairs, int64_t NP, const uint8_t *__restrict__ done, const uint32_t *__restrict__ b10,
                                                    const int32_t *__restrict__ slot_of_prof, int32_t *__restrict__ flag, const int32_t *__restrict__ pos,
                                                    unsigned long long *__restrict__ keys, int32_t *__restrict__ vals)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > NP) return;
  int f = 0, sl = -1;
  if (i < NP) {
    const PairRec pr = pairs[i];
    if (pr.prof >= 0 && pr.xj >= 0 && !done[i]) { sl = slot_of_prof[pr.prof]; f = sl >= 0; }
  }
  if (!pos) { flag[i] = f; return; }
  if (f) { keys[pos[i]] = ((unsigned long long)sl << 32) | (unsigned long long)(0xFFFFFFFFu - b10[i]); vals[pos[i]] = (int32_t)i; }
}
__global__ void __launch_bounds__(256) k_topup_mark(const PairRec *__restrict__ pairs, int64_t NP, uint8_t *__restrict__ done, const uint32_t *__restrict__ b10,
                                                    const int32_t *__restrict__ slot_of_prof, const uint32_t *__restrict__ cutoff, int32_t *__restrict__ flag)
{
  // cutoff[2 slot] > 0: the pairs whose bound is at least that (the best ones); cutoff[2 slot + 1] > 0: those whose bound is at most that
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > NP) return;
  int f = 0;
  if (i < NP) {
    const PairRec pr = pairs[i];
    if (pr.prof >= 0 && pr.xj >= 0 && !done[i]) {
      const int sl = slot_of_prof[pr.prof];
      if (sl >= 0) { const uint32_t hi = cutoff[2 * sl], lo = cutoff[2 * sl + 1], b = b10[i]; f = (hi > 0 && b >= hi) || (lo > 0 && b <= lo); }
    }
    if (f) done[i] = 1;
  }
  flag[i] = f;
}
void launch_topup_keys(const PairRec *pairs, int64_t NP, const uint8_t *done, const uint32_t *b10, const int32_t *slot_of_prof, int32_t *flag, const int32_t *pos,
                       unsigned long long *keys, int32_t *vals, hipStream_t st)
{
  hipLaunchKernelGGL(k_topup_keys, dim3((unsigned)((NP + 1 + 255) / 256)), dim3(256), 0, st, pairs, NP, done, b10, slot_of_prof, flag, pos, keys, vals);
}
void launch_topup_mark(const PairRec *pairs, int64_t NP, uint8_t *done, const uint32_t *b10, const int32_t *slot_of_prof, const uint32_t *cutoff, int32_t *flag, hipStream_t st)
{
  hipLaunchKernelGGL(k_topup_mark, dim3((unsigned)((NP + 1 + 255) / 256)), dim3(256), 0, st, pairs, NP, done, b10, slot_of_prof, cutoff, flag);
}

}  // namespace itsx
