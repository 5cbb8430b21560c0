// k_vit.hip -- the Viterbi filter (HMMER p7_ViterbiFilter), the second of hmmsearch's three filters.
//
// Reference call site: itsxpress/SeqSample.py:191-209 runs hmmsearch with --F1 1e-6 --F2 1e-6 --F3 1e-6; the pipeline enters
// this filter only for targets whose bias-corrected MSV P-value is ABOVE F2, which cannot happen when F1 == F2 -- so under the
// reference's flags the kernel is never launched.  It exists so that the engine follows hmmsearch for any flags (hmmsearch's
// own defaults are 0.02 / 1e-3 / 1e-5).
//
// Arithmetic: 16-bit saturating Viterbi (scale 500/ln2, base 12000), multihit local, N/C/J loops at cost 0 with a flat -3 nat
// correction at the end.  HMMER evaluates it 8-way striped and skips a row's D->D paths when they provably cannot matter
// ("lazy F"); maxima and saturating additions do not depend on the order of evaluation and a skipped D->D path never raises a
// match cell of the next row, so the plain recurrence gives the same xC (the tests hold it to the CPU restatement bit for bit).
// Mapping: as the float DP kernels -- one lane per (representative, profile) pair, a wave = 64 pairs of ONE profile, the profile's
// word tables (2.2 KB) in LDS, the three DP rows (3 x 47 cells) in registers through full unrolling.  Not tuned: off the
// reference's path.
#include "engine.h"
#include "k_api.h"
#include "detmath.h"
#include "k_vec.h"

namespace itsx {

DEV int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

__global__ void __launch_bounds__(64) k_vit(VitArgs a, int wave0)
{
  __shared__ int16_t tab[VIT_TAB];
  const WaveDesc wd = a.waves[wave0 + blockIdx.x];
  const int lane = threadIdx.x;
  const int prof = uni(wd.prof);
  const DevProfile *pp = a.prof + prof;
  const int M = uni(pp->M);
  for (int i = lane; i < VIT_TAB; i += 64) tab[i] = a.vtab[(size_t)prof * VIT_TAB + i];
  __syncthreads();
  const bool active = lane < wd.count;
  const int64_t pi = wd.first + (active ? lane : 0);
  const PairRec pr = a.pairs[pi];
  const PairOut po = a.pout[pi];
  bool needs = false;
  if (active && po.pass_bias) {
    const double P = gumbel_surv((double)(po.msv_sc - po.filtersc) / kLn2, (double)pp->ev[0], (double)pp->ev[1]);
    needs = P > a.F2;
  }
  VitOut vo; vo.vitsc = 0.0f; vo.ran = needs ? 1 : 0; vo.pass = 1;
  if (__ballot(needs) == 0ull) { if (active) a.vit[pi] = vo; return; }
  const int L = pr.L;
  const Seq sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  const int Lw = wd.rows - 1;
  const int xmove = a.lt[L].vmove, eloop = a.eloop, base = 12000;
  constexpr int S = MMAX + 1;
  const int16_t *tbm = tab, *tmm = tab + S, *tim = tab + 2 * S, *tdm = tab + 3 * S, *tmd = tab + 4 * S, *tmi = tab + 5 * S, *tii = tab + 6 * S, *tdd = tab + 7 * S;
  const int16_t *emis = tab + 8 * S;
  int mm[MMAX + 1], im[MMAX + 1], dm[MMAX + 1];
#pragma unroll
  for (int k = 0; k <= MMAX; k++) { mm[k] = -32768; im[k] = -32768; dm[k] = -32768; }
  int xN = base, xB = (int)(int16_t)(xN + xmove), xJ = -32768, xC = -32768;
  bool ovf = false;
  for (int i = 1; i <= Lw; i++) {
    if (!(needs && i <= L && !ovf)) continue;
    const int x = sq.code(i - 1);
    const int16_t *rsc = emis + x * S;
    int xe = -32768;
#pragma unroll
    for (int k = MMAX; k >= 1; k--) {               // descending: cell k-1 still holds the previous row
      if (k > M) continue;
      const int a0 = sat16(mm[k] + tmi[k]), a1 = sat16(im[k] + tii[k]);
      const int ni = a0 > a1 ? a0 : a1;
      int sv = sat16(xB + tbm[k]);
      int v = sat16(mm[k - 1] + tmm[k]); sv = v > sv ? v : sv;
      v = sat16(im[k - 1] + tim[k]); sv = v > sv ? v : sv;
      v = sat16(dm[k - 1] + tdm[k]); sv = v > sv ? v : sv;
      sv = sat16(sv + rsc[k]);
      xe = sv > xe ? sv : xe;
      mm[k] = sv; im[k] = ni;
    }
    dm[1] = -32768;
#pragma unroll
    for (int k = 2; k <= MMAX; k++) {
      if (k > M) continue;
      const int a0 = sat16(mm[k - 1] + tmd[k - 1]), a1 = sat16(dm[k - 1] + tdd[k - 1]);
      dm[k] = a0 > a1 ? a0 : a1;
    }
    if (xe >= 32767) { ovf = true; continue; }
    { const int c0 = xC, c1 = xe + eloop; xC = (int)(int16_t)(c0 > c1 ? c0 : c1); }
    { const int c0 = xJ, c1 = xe + eloop; xJ = (int)(int16_t)(c0 > c1 ? c0 : c1); }
    { const int c0 = xJ + xmove, c1 = xN + xmove; xB = (int)(int16_t)(c0 > c1 ? c0 : c1); }
  }
  if (needs) {
    float sc;
    if (ovf) sc = __builtin_inff();
    else if (xC > -32768) { sc = (float)xC + (float)xmove - (float)base; sc /= (float)(500.0 / kLn2); sc -= 3.0f; }
    else sc = -__builtin_inff();
    vo.vitsc = sc;
    const double P = gumbel_surv((double)(sc - po.filtersc) / kLn2, (double)pp->ev[2], (double)pp->ev[3]);
    vo.pass = !(P > a.F2);
  }
  if (active) a.vit[pi] = vo;
}

void launch_vit(const VitArgs &a, int nwaves, int wave0, hipStream_t st)
{
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_vit, dim3(nwaves), dim3(64), 0, st, a, wave0);
}

}  // namespace itsx
