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
// region: k_mr_fwd fills the region's Forward matrix (contiguous per region, node-major rows: see mrslab below),
// k_mr_trace walks the 200 paths, accumulates the per-residue null2 odds, dedupes the sampled (i, j, k, m) tuples,
// clusters them and writes the surviving envelopes.  Lanes of a wave work on different profiles, so the transition and
// emission tables come through ordinary vector loads here (the wave-uniform scalar-operand pipeline of k_float.hip does
// not apply); these kernels are latency-bound integer/float bookkeeping, not throughput kernels.
#include "engine.h"
#include "k_api.h"
#include "detmath.h"
#include "k_vec.h"
#include <type_traits>

namespace itsx {

enum { tBM = 0, tMM, tIM, tDM, tMD, tMI, tII, tDD };
enum { ST_M = 1, ST_D, ST_I, ST_S, ST_N, ST_B, ST_E, ST_C, ST_T, ST_J };

// A region's matrix is contiguous (rows of MRV float4; interleaved by lane, every float4 of a step was a line of its own).  Row
// layout (node-major): vector k = node k's {M, D, I, the row's B} for k = 0 (no node: zeros) .. 4 Q, then the row's specials
// {E N J B} and {E J C scale}.  A step of a traceback needs M, D, I of ONE node and B of that row: one vector load instead of four
// from a row striped like the DP's, and one more for the node's transitions (DevProfile::tfn).
DEV f4 *mrslab(const MrArgs &a, int64_t r0, int row, int v, int lane) { (void)lane; return (f4 *)a.slab + ((r0 + row) * MRV + v); }
constexpr int MR_SP = QMAX * 4 + 1, MR_CJ = QMAX * 4 + 2;
static_assert(MR_CJ < MRV, "matrix row too short");
DEV float comp4(const f4 &t, int r) { return r == 0 ? t.x : r == 1 ? t.y : r == 2 ? t.z : t.w; }
DEV f4 tof4(const V4 &v) { return (f4){v.a.x, v.a.y, v.b.x, v.b.y}; }

struct MrLane {
  bool active; int64_t mi, slot; MrRec m; int L, Lr, off, Q; const DevProfile *pp; Seq sq; float pmove, ploop; int64_t r0;
};
DEV MrLane mr_lane(const MrArgs &a, const WaveDesc &wd, int lane)
{
  MrLane e;
  e.active = lane < wd.count;
  const int64_t slot = wd.first + (e.active ? lane : 0);
  e.mi = a.sel ? (int64_t)a.sel[slot] : slot;          // index of the distinct region (the overflow path walks a selection)
  e.m = a.mr[a.ulist[e.mi]];
  const PairRec pr = a.pairs[e.m.pair];
  e.L = pr.L; e.Lr = e.m.jreg - e.m.ireg + 1; e.off = e.m.ireg - 1;
  e.pp = a.prof + pr.prof; e.Q = e.pp->Q;
  e.sq = open_seq(a.rd, a.seed_read[a.sorted_uniq[pr.useq]]);
  e.pmove = (2.0f + 1.0f) / ((float)e.L + 2.0f + 1.0f);       // multihit, length model of the whole target
  e.ploop = 1.0f - e.pmove;
  e.r0 = a.sel ? a.sel_rowoff[slot] : a.rowoff[e.mi] - a.rowoff0;
  e.slot = slot;
  return e;
}

// =========================================================================================
// the region's Forward matrix: p7_Forward(dsq + i - 1, j - i + 1) in multihit mode, every row kept
__global__ void __launch_bounds__(64) k_mr_fwd(MrArgs a, int wave0)
{
  const WaveDesc wd = a.waves[wave0 + (gridDim.x - 1 - blockIdx.x)];       // longest regions first
  const int lane = threadIdx.x;
  const MrLane e = mr_lane(a, wd, lane);
  const int Q = e.Q, Lw = wd.rows - 1;
  const float *tf = e.pp->tf;
  V4 M[QMAX], D[QMAX], I[QMAX];
#pragma unroll
  for (int q = 0; q < QMAX; q++) { M[q] = vzero(); D[q] = vzero(); I[q] = vzero(); }
  float xE = 0.f, xN = 1.f, xJ = 0.f, xB = e.pmove, xC = 0.f;
  if (e.active) {
    for (int k = 0; k <= Q * 4; k++) *mrslab(a, e.r0, 0, k, lane) = (f4){0.f, 0.f, 0.f, xB};
    *mrslab(a, e.r0, 0, MR_SP, lane) = (f4){xE, xN, xJ, xB};
    *mrslab(a, e.r0, 0, MR_CJ, lane) = (f4){xE, xJ, xC, 1.0f};
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
    *mrslab(a, e.r0, i, 0, lane) = (f4){0.f, 0.f, 0.f, xB};
#pragma unroll
    for (int q = 0; q < QMAX; q++) {
      if (q >= Q) break;
      const f4 m4 = tof4(M[q]), d4 = tof4(D[q]), i4 = tof4(I[q]);          // striped: component r is node r Q + q + 1
      *mrslab(a, e.r0, i, 0 * Q + q + 1, lane) = (f4){m4.x, d4.x, i4.x, xB};
      *mrslab(a, e.r0, i, 1 * Q + q + 1, lane) = (f4){m4.y, d4.y, i4.y, xB};
      *mrslab(a, e.r0, i, 2 * Q + q + 1, lane) = (f4){m4.z, d4.z, i4.z, xB};
      *mrslab(a, e.r0, i, 3 * Q + q + 1, lane) = (f4){m4.w, d4.w, i4.w, xB};
    }
    *mrslab(a, e.r0, i, MR_SP, lane) = (f4){xE, xN, xJ, xB};
    *mrslab(a, e.r0, i, MR_CJ, lane) = (f4){xE, xJ, xC, sc};
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
// esl_vec_FNorm + esl_rnd_FChoose over n = 2 or 4 path weights (w2 = w3 = 0 when n == 2: adding +0 leaves the sums as they are)
DEV int choose_n(uint32_t &rng, int n, float p0, float p1, float p2, float p3)
{
  float sum = 0.f; sum += p0; sum += p1; sum += p2; sum += p3;
  if (sum != 0.0f) { p0 /= sum; p1 /= sum; p2 /= sum; p3 /= sum; }
  else { p0 = p1 = (float)(1. / (double)(float)n); p2 = p3 = n == 4 ? p0 : 0.0f; }
  const double roll = rng_next(rng);
  double s = 0.0;
  s += p0; if (roll < s) return 0;
  s += p1; if (roll < s) return 1;
  s += p2; if (roll < s) return 2;
  s += p3; if (roll < s) return 3;
  int i;
  do { i = (int)(rng_next(rng) * n); } while ((i == 0 ? p0 : i == 1 ? p1 : i == 2 ? p2 : p3) == 0.f);
  return i;
}

// the same over two weights (choose_n with w2 = w3 = 0: the sums, the quotients and the draws are the same)
DEV int choose2(uint32_t &rng, float p0, float p1)
{
  float sum = 0.f; sum += p0; sum += p1;
  if (sum != 0.0f) { p0 /= sum; p1 /= sum; }
  else { p0 = p1 = (float)(1. / (double)(float)2); }
  const double roll = rng_next(rng);
  double s = 0.0;
  s += p0; if (roll < s) return 0;
  s += p1; if (roll < s) return 1;
  int i;
  do { i = (int)(rng_next(rng) * 2); } while ((i == 0 ? p0 : p1) == 0.f);
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
  uint16_t hslot[MR_HASH];                  // open-addressing index over key[]: tuple number + 1, 0 = empty
  uint16_t epc[MR_EPC];                     // endpoint histogram of one coordinate of one cluster
  int32_t sig_i[MR_NSIG], sig_j[MR_NSIG]; float sig_p[MR_NSIG]; uint8_t dominated[MR_NSIG];
  f4 dn2[MR_MAXD];                          // null2 odds (A C G T) of the domains of the path being sampled
};
static_assert(sizeof(MrScratch) <= MR_SCRATCH, "scratch block too small");

// BIG = the overflow path: the same procedure for the regions whose ensemble does not fit the fast kernel's fixed bookkeeping (more
// than 8 domains in one sampled path, more than 512 distinct tuples, more than 4 envelopes: concatemers, tandem partial copies).
// Every per-region array then lives in a block of global memory sized by the region's length -- a path of a region of Lr residues
// has at most Lr domains, 200 paths at most 200 Lr tuples, at most 4 Lr clusters reach a quarter of the paths -- with 32-bit
// indices: nothing can overrun, as in hmmsearch (p7_domaindef.c grows its lists).  Same random stream, same sums, same order.
// ONE = the small-batch form: a wave walks ONE region (lane 0) and keeps the region's Forward matrix in LDS.  A sampled path is a
// chain of dependent reads of that matrix -- 200 paths x ~150 steps, each a round trip to L2 when the matrix is in global memory:
// 20 ms of a wave's 32 whatever the batch holds -- and a batch of a thousand regions (a 1 M-read shard, a chunk of a streaming run)
// cannot hide it behind other waves.  From LDS the same reads take a fifth of the time; the wave's other lanes only help copying.
// wave0 is then the first REGION of the launch; a matrix that does not fit the launch's LDS is walked where it is.
template <bool BIG, bool ONE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_mr_trace(MrArgs a, int wave0)
{
  extern __shared__ f4 mr_mat[];
  using idx_t = typename std::conditional<BIG, uint32_t, uint16_t>::type;
  using sidx_t = typename std::conditional<BIG, int32_t, int16_t>::type;
  constexpr idx_t NONE = (idx_t)~(idx_t)0;
  // usage of the domain being walked: a match state is visited at most once per node (a bit mask), insert states are counted
  constexpr int NL = ONE ? 1 : MR_LANES;                   // (one region per wave: the per-lane arrays are 150 bytes, the LDS goes to the matrix)
  __shared__ uint16_t cntI_s[QMAX * 4][NL];
  // the domains of the path being sampled, last first: first / last residue, first / last node, null2 odds of A C G T
  __shared__ uint32_t dom_ij[MR_MAXD][NL];
  __shared__ uint16_t dom_km[MR_MAXD][NL];                 // (the domains' null2 odds live in the region's scratch block: S.dn2 ...
  __shared__ f4 dn2_one[ONE ? MR_MAXD : 1];                // ... ONE: here, where the lanes that did not write them read them)
  WaveDesc wd;
  if constexpr (ONE) { wd.prof = -1; wd.first = wave0 + (int64_t)(gridDim.x - 1 - blockIdx.x); wd.count = 1; wd.rows = 0; }
  else wd = a.waves[wave0 + (gridDim.x - 1 - blockIdx.x)];                 // longest regions first
  // COOP (a wave of a lazy round: 2 - 8 regions, the other lanes used to leave at once): lane t serves region t mod P2 of the wave as
  // helper t / P2 of its G = 64 / P2; helper 0 is the region's walker.  Helpers take part in nothing but the loops over the region's
  // residues and arrays (below); the per-lane LDS columns and the scratch block they index are their walker's.
  const bool coop = !ONE && !BIG && wd.count <= 8;
  int P2 = 1; while (P2 < wd.count) P2 <<= 1;
  const int G = coop ? 64 / P2 : 1, hlp = coop ? (int)threadIdx.x / P2 : 0;
  const int lane = ONE ? 0 : (coop ? (int)threadIdx.x % P2 : (int)threadIdx.x);
  const MrLane e = mr_lane(a, wd, lane);
  const f4 *mat = (const f4 *)a.slab + e.r0 * MRV;                          // the region's matrix: [row 0..Lr][MRV]
  bool in_lds = false;
  if constexpr (ONE) {
    const int64_t nvec = (int64_t)(e.Lr + 1) * MRV;
    if (nvec * 16 <= (int64_t)a.lds_bytes) {
      for (int64_t z = threadIdx.x; z < nvec; z += 64) mr_mat[z] = mat[z];
      in_lds = true;
    }
    __syncthreads();
  } else {
    if (!e.active) return;
  }
  // ONE: lane 0 walks the paths (a chain of dependent draws: nothing to share); what a finished path needs done for its residues, and the
  // loops over the region's arrays before and after, are spread over the wave's 64 lanes
  const bool walker = ONE ? threadIdx.x == 0 : hlp == 0;
  const int tid = (int)threadIdx.x;
  // (two typed reads under a uniform branch rather than one generic pointer: an LDS read, not a flat load that happens to land there)
#define MATV(idx) ((ONE && in_lds) ? mr_mat[(idx)] : mat[(idx)])
  const int Q = e.Q, Lr = e.Lr, Lw = ONE ? e.Lr : wd.rows - 1;
  const float *tfn = e.pp->tfn;
  const float pmove = e.pmove, ploop = e.ploop;
  float *n2 = a.n2sc + a.n2off[e.mi];                       // n2[pos - 1], pos = 1..Lr relative to the region
  // the region's bookkeeping: a fixed block per lane (fast path) or a block of the arena sized by the region (overflow path)
  MrBig bg; bg.off = 0; bg.envoff = 0; bg.capD = MR_MAXD; bg.capT = MR_TCAP; bg.hmask = MR_HASH - 1; bg.pad = 0;
  if constexpr (BIG) bg = a.big[e.slot];
  const int capD = BIG ? bg.capD : MR_MAXD, capT = BIG ? bg.capT : MR_TCAP, capS = BIG ? bg.capT : MR_SCAP;
  const int capSig = BIG ? 4 * bg.capD : MR_NSIG, capEnv = BIG ? 4 * bg.capD : MRENV;
  const uint32_t hmask = BIG ? bg.hmask : (uint32_t)(MR_HASH - 1);
  unsigned long long *Skey; idx_t *Stcount, *Scomp, *Sstack, *Sninc, *Stid, *Shslot; sidx_t *Slast; uint8_t *Stidx, *Sdominated;
  uint16_t *Sepc; int32_t *Ssig_i, *Ssig_j, *Senv = nullptr; float *Ssig_p; f4 *Sdn2; uint32_t *Gdom_ij = nullptr; uint16_t *Gdom_km = nullptr;
  if constexpr (BIG) {
    uint8_t *blk = a.arena + bg.off;
    size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t *q = blk + o; o += (bytes + 15) & ~(size_t)15; return q; };
    Skey = (unsigned long long *)take(8 * (size_t)capT);
    Stcount = (idx_t *)take(4 * (size_t)capT); Scomp = (idx_t *)take(4 * (size_t)capT); Sstack = (idx_t *)take(4 * (size_t)capT);
    Sninc = (idx_t *)take(4 * (size_t)capT); Slast = (sidx_t *)take(4 * (size_t)capT); Stid = (idx_t *)take(4 * (size_t)capT);
    Shslot = (idx_t *)take(4 * ((size_t)hmask + 1));
    Gdom_ij = (uint32_t *)take(4 * (size_t)capD); Sdn2 = (f4 *)take(16 * (size_t)capD);
    Ssig_i = (int32_t *)take(4 * (size_t)capSig); Ssig_j = (int32_t *)take(4 * (size_t)capSig); Ssig_p = (float *)take(4 * (size_t)capSig);
    Senv = a.envpool + bg.envoff;
    Stidx = (uint8_t *)take((size_t)capT); Gdom_km = (uint16_t *)take(2 * (size_t)capD); Sdominated = (uint8_t *)take((size_t)capSig);
    Sepc = (uint16_t *)take(2 * MR_EPC);
  } else {
    MrScratch &S = *(MrScratch *)(a.scratch + (int64_t)(e.mi - a.u0) * MR_SCRATCH);
    Skey = S.key; Stcount = (idx_t *)S.tcount; Scomp = (idx_t *)S.comp; Sstack = (idx_t *)S.stack; Sninc = (idx_t *)S.ninc; Slast = (sidx_t *)S.last;
    Stid = (idx_t *)S.tid; Stidx = S.tidx; Shslot = (idx_t *)S.hslot; Sepc = S.epc; Ssig_i = S.sig_i; Ssig_j = S.sig_j; Ssig_p = S.sig_p;
    Sdominated = S.dominated; Sdn2 = S.dn2;
  }
#define DOM_IJ(d) (*(BIG ? &Gdom_ij[(d)] : &dom_ij[(d) < MR_MAXD ? (d) : 0][lane]))
#define DOM_KM(d) (*(BIG ? &Gdom_km[(d)] : &dom_km[(d) < MR_MAXD ? (d) : 0][lane]))
  if constexpr (ONE) {
    for (int pos = tid; pos < Lr; pos += 64) n2[pos] = 0.0f;
    for (uint32_t z = (uint32_t)tid; z <= hmask; z += 64) Shslot[z] = 0;
    __threadfence_block();
    __syncthreads();
  } else if (coop) {
    for (int pos = hlp; pos < Lr; pos += G) n2[pos] = 0.0f;
    for (uint32_t z = (uint32_t)hlp; z <= hmask; z += (uint32_t)G) Shslot[z] = 0;
    __threadfence_block();
    __syncthreads();
  } else {
    for (int pos = 0; pos < Lr; pos++) n2[pos] = 0.0f;
    for (uint32_t z = 0; z <= hmask; z++) Shslot[z] = 0;
  }
  uint32_t rng = rnd_mix3(42u, 87654321u, 12345678u);
  if (rng == 0) rng = 42;
  int ntup = 0, nsamp = 0, status = walker ? 0 : 9;         // (9: a lane that does not walk)
  const unsigned short dgm[16] = {1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15};
  // Seq::code() walks the read's exception list through global loads on every call: the residue loop below would pay that for
  // every residue of every path of a read that has a single N (15 % of the bench's reads have one)
  int ex0p = -1, ex0c = 0, ex1p = -1, ex1c = 0;
  if (e.sq.nexc >= 1 && e.sq.nexc <= 2) { const uint32_t v = e.sq.exc[0]; ex0p = (int)(v >> 4); ex0c = (int)(v & 15u); }
  if (e.sq.nexc == 2) { const uint32_t v = e.sq.exc[1]; ex1p = (int)(v >> 4); ex1c = (int)(v & 15u); }

  // The lanes of a wave sample different regions.  Left to themselves they would be in different states at any moment and
  // the wave would run every state's code in every step (measured: 9 us per step, most of it the rare but long E and B
  // states that SOME lane is always in).  Every path has the same shape -- C* E (M|D|I)* B [J* E (M|D|I)* B]* N* -- so the
  // wave walks it in phases, all lanes in the same kind of state: the special-state loops, then E for every lane that reached
  // it, then the core states, then B.  A lane's own sequence of random draws is untouched.  Inside the core phase the loads
  // of a step are issued for all three states at once and the choice runs through one routine.  N -> N steps draw nothing
  // and record nothing, so a path that enters N is finished.  What a finished domain needs done for its residues waits until
  // the path is complete and is then done by all lanes together.
  unsigned long long tk_walk = 0, tk_close = 0, tk_dedupe = 0;
  for (int t = 0; t < 200; t++) {
    if (__ballot(status == 0) == 0ull) break;
    const unsigned long long tk0 = wall_clock64();
    int i = Lr, k = 0, s0 = status == 0 ? ST_C : ST_S, nd = 0;
    int cj_row = -1; f4 cj_cur = (f4){0.f, 0.f, 0.f, 0.f};
    int dfrom = 0, dto = 0, dk = 0, dm = 0;
    unsigned long long maskM = 0ull;
    while (__ballot(s0 != ST_S) != 0ull) {
      // ---- C* / J*: one draw per step
      while (__ballot(s0 == ST_C || s0 == ST_J) != 0ull) {
        if (!(s0 == ST_C || s0 == ST_J)) continue;
        if (i < 1) { status = 5; s0 = ST_S; continue; }
        // {E J C scale} of rows i - 1 and i: row i's vector is the one the previous step loaded as row i - 1
        if (cj_row != i) { cj_cur = MATV((int64_t)i * MRV + MR_CJ); cj_row = i; }
        const f4 c0 = MATV((int64_t)(i - 1) * MRV + MR_CJ);
        const float w0 = (s0 == ST_C ? c0.z : c0.y) * ploop, w1 = cj_cur.x * 0.5f * cj_cur.w;
        if (choose2(rng, w0, w1) == 0) { i--; cj_cur = c0; cj_row = i; } else s0 = ST_E;
      }
      // ---- E: which match or delete state the domain ends in
      if (__ballot(s0 == ST_E) != 0ull) {
        if (s0 == ST_E) {
          double sum = 0.0;
          const double roll = rng_next(rng);
          const double norm = 1.0 / (double)cj_cur.x;                       // E of row i: the C* / J* loop ends holding it
          const float xEv = (float)norm;
          int s1 = -1;
          while (s1 < 0) {
            for (int q0 = 0; q0 < Q && s1 < 0; q0 += 3) {               // three node groups' (M, D) pairs requested at once;
              f2 md[3][4];                                               // the sum runs in the striped order: group by group
#pragma unroll
              for (int z = 0; z < 3; z++) {
                const int qq = q0 + z < Q ? q0 + z : Q - 1;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) { const f4 q4 = MATV((int64_t)i * MRV + (rr * Q + qq + 1)); md[z][rr] = (f2){q4.x, q4.y}; }
              }
#pragma unroll
              for (int z = 0; z < 3; z++) {
                if (q0 + z >= Q || s1 >= 0) continue;
                const int qq = q0 + z;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) { if (s1 < 0) { sum += (double)(md[z][rr].x * xEv); if (roll < sum) { k = rr * Q + qq + 1; s1 = ST_M; } } }
#pragma unroll
                for (int rr = 0; rr < 4; rr++) { if (s1 < 0) { sum += (double)(md[z][rr].y * xEv); if (roll < sum) { k = rr * Q + qq + 1; s1 = ST_D; } } }
              }
            }
            if (s1 < 0 && sum < 0.99) { status = 1; s1 = ST_S; }          // HMMER throws here
          }
          if (status == 0 && nd >= capD) { status = 2; s1 = ST_S; }
          if (status == 0) {
            nd++;
            dfrom = dto = dk = dm = 0;
            maskM = 0ull;
            for (int z = 0; z < Q * 4; z++) cntI_s[z][lane] = 0;
            if (s1 == ST_M) { dto = i; dm = k; dfrom = i; dk = k; maskM |= 1ull << k; }
          }
          s0 = s1;
        }
      }
      // ---- (M | D | I)*: the loads of a step for whichever of the three the lane is in, then one choice
      while (__ballot(s0 == ST_M || s0 == ST_D || s0 == ST_I) != 0ull) {
        if (!(s0 == ST_M || s0 == ST_D || s0 == ST_I)) continue;
        // one vector = M, D, I of the node the step comes from and B of its row; one vector = that step's transitions
        // (M: B M I D -> M of node k; D: M D -> D out of node k - 1; I: M I -> I of node k)
        const int rowA = (s0 == ST_D) ? i : i - 1;
        const int kA = (s0 == ST_I) ? k : k - 1;
        const f4 nv = MATV((int64_t)rowA * MRV + kA);
        const f4 tv = *(const f4 *)(tfn + (s0 == ST_D ? kA : k) * 8 + (s0 == ST_M ? 0 : 4));
        float w0, w1, w2 = 0.0f, w3 = 0.0f;
        if (s0 == ST_M) { w0 = nv.w * tv.x; w1 = nv.x * tv.y; w2 = nv.z * tv.z; w3 = nv.y * tv.w; }
        else if (s0 == ST_D) { w0 = nv.x * tv.x; w1 = nv.y * tv.w; }              // (node 0 and its transitions are zeros)
        else { w0 = nv.x * tv.y; w1 = nv.z * tv.z; }
        const int c = choose_n(rng, s0 == ST_M ? 4 : 2, w0, w1, w2, w3);
        int s1;
        if (s0 == ST_M) { s1 = c == 0 ? ST_B : c == 1 ? ST_M : c == 2 ? ST_I : ST_D; k--; i--; }
        else if (s0 == ST_D) { s1 = c == 0 ? ST_M : ST_D; k--; }
        else { s1 = c == 0 ? ST_M : ST_I; i--; }
        if (s1 == ST_M) {
          if (dto == 0) { dto = i; dm = k; }
          dfrom = i; dk = k;
          maskM |= 1ull << k;
        } else if (s1 == ST_I) {
          cntI_s[k - 1][lane] += 1;
        }
        s0 = s1;
      }
      // ---- B: where the domain was entered from, and p7_Null2_ByTrace over its states (only match and insert states emit)
      if (__ballot(s0 == ST_B) != 0ull) {
        if (s0 == ST_B) {
          const f4 sp1 = MATV((int64_t)i * MRV + MR_SP);
          s0 = choose2(rng, sp1.y * pmove, sp1.z * pmove) == 0 ? ST_S : ST_J;        // N: the path is finished
          int Ld = __popcll(maskM);
          for (int z = 0; z < Q * 4; z++) Ld += (int)cntI_s[z][lane];
          const float norm = (float)(1.0 / (double)(float)Ld);
          const float xfactor = (0.0f * norm + 0.0f * norm) + 0.0f * norm;
          V4 sv[4];
#pragma unroll
          for (int x = 0; x < 4; x++) sv[x] = vzero();
          for (int qq = 0; qq < Q; qq++) {
            V4 mv, iv;
            // node of (group qq, SSE lane z) = z Q + qq + 1; its match count is 1 or 0
            mv.a = (f2){(float)((maskM >> (0 * Q + qq + 1)) & 1ull), (float)((maskM >> (1 * Q + qq + 1)) & 1ull)};
            mv.b = (f2){(float)((maskM >> (2 * Q + qq + 1)) & 1ull), (float)((maskM >> (3 * Q + qq + 1)) & 1ull)};
            iv.a = (f2){(float)cntI_s[0 * Q + qq][lane], (float)cntI_s[1 * Q + qq][lane]}; iv.b = (f2){(float)cntI_s[2 * Q + qq][lane], (float)cntI_s[3 * Q + qq][lane]};
            mv = vmul(mv, vset(norm)); iv = vmul(iv, vset(norm));
#pragma unroll
            for (int x = 0; x < 4; x++) {
              sv[x] = vadd(sv[x], vmul(mv, vld(e.pp->rf + x * QMAX * 4 + qq * 4)));
              sv[x] = vadd(sv[x], iv);
            }
          }
          DOM_IJ(nd - 1) = (uint32_t)dfrom | ((uint32_t)dto << 16);
          DOM_KM(nd - 1) = (uint16_t)((uint32_t)dk | ((uint32_t)dm << 8));
          { float v4_[4];
#pragma unroll
            for (int x = 0; x < 4; x++) { float v = vhsum(sv[x]); v += xfactor; v4_[x] = v; }
            if constexpr (ONE) dn2_one[nd - 1 < MR_MAXD ? nd - 1 : 0] = (f4){v4_[0], v4_[1], v4_[2], v4_[3]};
            else Sdn2[nd - 1] = (f4){v4_[0], v4_[1], v4_[2], v4_[3]}; }
        }
      }
    }
    const unsigned long long tk1 = wall_clock64();
    tk_walk += tk1 - tk0;
    if constexpr (ONE) { if (__shfl(status, 0, 64) != 0) continue; }      // (the walker's status decides for the wave)
    else if (!coop && status) continue;                                    // (COOP: no lane leaves the path early -- predicates below)
    // ---- after the path, all lanes together: its samples ...
    for (int d = 0; d < ((walker && status == 0) ? nd : 0); d++) {
      if (nsamp >= capS) { status = 3; break; }
      const uint32_t ij = DOM_IJ(d), km = (uint32_t)DOM_KM(d);
      const unsigned long long key = pack_tup((int)(ij & 0xffff), (int)(ij >> 16), (int)(km & 0xff), (int)(km >> 8));   // relative to the region
      uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (BIG ? 32 : 40)) & hmask;
      int tix = -1;
      for (;;) {
        const int v = Shslot[h];
        if (v == 0) break;
        if (Skey[v - 1] == key) { tix = v - 1; break; }
        h = (h + 1) & hmask;
      }
      if (tix < 0) {
        if (ntup >= capT) { status = 4; break; }
        tix = ntup; Skey[ntup] = key; Stcount[ntup] = 0; ntup++;
        Shslot[h] = (idx_t)(tix + 1);
      }
      Stcount[tix]++;
      Stid[nsamp] = (idx_t)tix; Stidx[nsamp] = (uint8_t)t; nsamp++;
    }
    tk_dedupe += wall_clock64() - tk1;
    if constexpr (ONE) { if (__shfl(status, 0, 64) != 0) continue; }
    else if (!coop && status) continue;
    if (!ONE && !BIG && coop) {
      // the region's helpers share its residues: helper h takes positions Lr - h, Lr - h - G, ...  The path's domains are in the walker's
      // LDS column; a domain's null2 odds are the walker's to read (its own stores) and reach the helpers through the wave.  A residue's
      // terms still come from ONE lane, one per path, in path order: every sum is the one the walker alone would have formed.
      const int wok = __shfl((int)(walker && status == 0), lane, 64);
      const int ndw = __shfl(nd, lane, 64);
      __syncthreads();                                       // (the walkers' dom_ij columns are written)
      for (int dd = 0; dd <= MR_MAXD; dd++) {                // domain dd of the path; dd == its number of domains: the residues outside all
        if (__ballot(wok && dd <= ndw) == 0ull) break;
        f4 q = (f4){1.0f, 1.0f, 1.0f, 1.0f};
        if (walker && wok && dd < ndw) q = Sdn2[dd];
        const float qx = __shfl(q.x, lane, 64), qy = __shfl(q.y, lane, 64), qz = __shfl(q.z, lane, 64), qw = __shfl(q.w, lane, 64);
        if (!wok || dd > ndw) continue;
        const f4 n2d = (f4){qx, qy, qz, qw};
        for (int pos = Lr - hlp; pos >= 1; pos -= G) {
          int d = 0;
          while (d < ndw && pos <= (int)(DOM_IJ(d) & 0xffff)) d++;
          const bool inside = d < ndw && pos <= (int)(DOM_IJ(d) >> 16);
          if (inside ? d != dd : dd != ndw) continue;        // (this round is another domain's, or the outside residues')
          float v = 1.0f;
          if (inside) {
            const int p0 = e.off + pos - 1;
            int x;
            if (e.sq.nexc <= 2) {
              const uint32_t cw = e.sq.w[p0 >> 4];
              x = (int)((cw >> (2 * (p0 & 15))) & 3u);
              x = p0 == ex0p ? ex0c : x; x = p0 == ex1p ? ex1c : x;
            } else x = e.sq.code(p0);
            if (x < 4) v = comp4(n2d, x);
            else {
              float acc = 0.f; int ndg = 0;
#pragma unroll
              for (int y = 0; y < 4; y++) if (dgm[x] >> y & 1) { acc += comp4(n2d, y); ndg++; }
              v = acc / (float)ndg;
            }
          }
          __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(n2 + pos - 1), v);
        }
      }
      __syncthreads();                                       // (before the walkers' next path overwrites the columns)
      tk_close += wall_clock64() - tk1;
      continue;
    }
    // ... and its residues' null2 terms.  As published: residues up to AND INCLUDING a domain's first one count as outside
    // (+1), the rest of the domain by its null2 odds; a residue takes exactly one term per path, so their order is free.
    if constexpr (ONE) {
      // one residue per lane and round: the path's domains come from the walker (LDS), a residue's terms still arrive in path order --
      // the same lane sends them, one per path -- so every sum is the one lane 0 alone would have formed
      const int ndw = __shfl(nd, 0, 64);
      __syncthreads();                                       // (the walker's dom_ij / dn2_one are written)
      for (int pos = Lw - tid; pos >= 1; pos -= 64) {
        if (pos > Lr) continue;
        float v = 1.0f;
        int d = 0;
        while (d < ndw && pos <= (int)(DOM_IJ(d) & 0xffff)) d++;
        if (d < ndw && pos <= (int)(DOM_IJ(d) >> 16)) {
          const int p0 = e.off + pos - 1;
          int x;
          if (e.sq.nexc <= 2) {
            const uint32_t cw = e.sq.w[p0 >> 4];
            x = (int)((cw >> (2 * (p0 & 15))) & 3u);
            x = p0 == ex0p ? ex0c : x; x = p0 == ex1p ? ex1c : x;
          } else x = e.sq.code(p0);
          const f4 n2d = dn2_one[d < MR_MAXD ? d : 0];
          if (x < 4) v = comp4(n2d, x);
          else {
            float acc = 0.f; int ndg = 0;
#pragma unroll
            for (int y = 0; y < 4; y++) if (dgm[x] >> y & 1) { acc += comp4(n2d, y); ndg++; }
            v = acc / (float)ndg;
          }
        }
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(n2 + pos - 1), v);
      }
      __syncthreads();                                       // (before the walker's next path overwrites the domains)
    } else {
      int d = 0, dl = -1;
      f4 n2d = (f4){0.f, 0.f, 0.f, 0.f};                       // the null2 odds of domain dl
      uint32_t cw = 0; int cwi = -1;
      // four residues per round: their terms first, then four independent read-modify-writes in flight together
      for (int p4 = Lw; p4 >= 1; p4 -= 4) {
        float v[4];
#pragma unroll
        for (int z = 0; z < 4; z++) {
          const int pos = p4 - z;
          v[z] = 1.0f;
          if (pos < 1 || pos > Lr) continue;
          while (d < nd && pos <= (int)(DOM_IJ(d) & 0xffff)) d++;
          if (d < nd && pos <= (int)(DOM_IJ(d) >> 16)) {
            const int p0 = e.off + pos - 1;
            int x;
            if (e.sq.nexc <= 2) {                       // the read's (at most two) non-ACGT symbols sit in registers
              if ((p0 >> 4) != cwi) { cwi = p0 >> 4; cw = e.sq.w[cwi]; }
              x = (int)((cw >> (2 * (p0 & 15))) & 3u);
              x = p0 == ex0p ? ex0c : x; x = p0 == ex1p ? ex1c : x;
            } else x = e.sq.code(p0);
            if (d != dl) { dl = d; n2d = Sdn2[d]; }
            if (x < 4) v[z] = comp4(n2d, x);
            else {
              float acc = 0.f; int ndg = 0;
#pragma unroll
              for (int y = 0; y < 4; y++) if (dgm[x] >> y & 1) { acc += comp4(n2d, y); ndg++; }
              v[z] = acc / (float)ndg;
            }
          }
        }
        // fire-and-forget float adds at L2 (global_atomic_add_f32, round to nearest even like v_add_f32): a residue's terms
        // arrive in path order -- one wave, one address, one channel -- and nothing waits for them until the scores are read
#pragma unroll
        for (int z = 0; z < 4; z++) {
          const int pos = p4 - z;
          if (pos >= 1 && pos <= Lr) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(n2 + pos - 1), v[z]);
        }
      }
    }
    tk_close += wall_clock64() - tk1;
  }
  const unsigned long long tk2 = wall_clock64();

  if constexpr (ONE) status = __shfl(status, 0, 64);        // (from here on every lane carries the walker's)
  else if (coop) status = __shfl(status, lane, 64);
  if (ONE || coop) __threadfence();                         // (the residues' sums are read below by other lanes than sent their terms)
  MrOut out;
  out.status = status; out.nenv = 0; out.big = -1;
  if constexpr (BIG) out.big = bg.envoff;
#pragma unroll
  for (int z = 0; z < MRENV; z++) { out.ei[z] = 0; out.ej[z] = 0; }
  if (status != 0) {
    for (int pos = ONE ? tid : hlp; pos < Lr; pos += ONE ? 64 : G) n2[pos] = 0.0f;
    if (walker) a.out[e.mi] = out;
    return;
  }
  for (int pos = ONE ? tid : hlp; pos < Lr; pos += ONE ? 64 : G) n2[pos] = det_logf(n2[pos] / (float)200);
  if (!walker) return;                                      // (the clustering below is the walker's alone)

  // ---- single-linkage clustering over the DISTINCT tuples.  Copies of one tuple always share their neighbours, so they
  // fall into one component -- except an isolated tuple that does not link to itself (model span under 5 nodes): there
  // every copy is a cluster of its own with posterior 1/200, which never reaches 0.25.
  for (int h = 0; h < ntup; h++) Scomp[h] = NONE;
  int nc = 0;
  for (int h0 = 0; h0 < ntup; h0++) {
    if (Scomp[h0] != NONE) continue;
    int ns = 0; Sstack[ns++] = (idx_t)h0; Scomp[h0] = (idx_t)nc;
    int members = 0;
    while (ns > 0) {
      const int v = Sstack[--ns];
      members++;
      const Tup tv = unpack_tup(Skey[v]);
      for (int u0 = 0; u0 < ntup; u0 += 4) {                      // four candidates requested together
        unsigned long long ku[4]; idx_t cu[4];
#pragma unroll
        for (int z = 0; z < 4; z++) { const int u = u0 + z < ntup ? u0 + z : ntup - 1; ku[z] = Skey[u]; cu[z] = Scomp[u]; }
#pragma unroll
        for (int z = 0; z < 4; z++) {
          const int u = u0 + z;
          if (u < ntup && cu[z] == NONE && link_tup(tv, unpack_tup(ku[z]))) { Scomp[u] = (idx_t)nc; Sstack[ns++] = (idx_t)u; }
        }
      }
    }
    const Tup t0 = unpack_tup(Skey[h0]);
    Sninc[nc] = (members == 1 && !link_tup(t0, t0)) ? NONE : (idx_t)0;      // 0xffff: a set of singletons, never reported
    Slast[nc] = -1;
    nc++;
  }
  // posterior of each cluster: traces with at least one member (samples are in trace order)
  for (int h = 0; h < nsamp; h++) {
    const int c = Scomp[Stid[h]];
    if (Sninc[c] == NONE) continue;
    if ((int)Stidx[h] != (int)Slast[c]) Sninc[c]++;
    Slast[c] = (sidx_t)Stidx[h];
  }
  int nsig = 0;
  for (int c = 0; c < nc; c++) {
    if (Sninc[c] == NONE) continue;
    const int ninc = (int)Sninc[c];
    if ((float)ninc / (float)200 < 0.25f) continue;
    const int thr = (int)ceilf((float)ninc * 0.02f);
    // the cluster's members, then per coordinate the endpoint histogram (weight of a value = copies of the tuples carrying it)
    int nm = 0;
    int lo[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30}, hi[4] = {-1, -1, -1, -1};
    for (int h = 0; h < ntup; h++) {
      if (Scomp[h] != c) continue;
      Sstack[nm++] = (idx_t)h;
      const Tup th = unpack_tup(Skey[h]);
      const int v[4] = {th.i, th.k, th.j, th.m};
#pragma unroll
      for (int f = 0; f < 4; f++) { lo[f] = min(lo[f], v[f]); hi[f] = max(hi[f], v[f]); }
    }
    int best[4];
    for (int f = 0; f < 4; f++) {
      const bool leftmost = f < 2;                    // i and k: leftmost value with enough endpoints; j and m: rightmost
      const int W = hi[f] - lo[f] + 1;
      int pick = -1, am = -1, amw = -1;
      if (W <= MR_EPC && (!BIG || nsamp < 65536)) {
        for (int z = 0; z < W; z++) Sepc[z] = 0;
        for (int x = 0; x < nm; x++) {
          const int h = Sstack[x];
          const Tup th = unpack_tup(Skey[h]);
          const int v = f == 0 ? th.i : f == 1 ? th.k : f == 2 ? th.j : th.m;
          Sepc[v - lo[f]] += Stcount[h];
        }
        for (int z = 0; z < W; z++) {
          const int w = Sepc[z];
          if (w >= thr && (pick < 0 || !leftmost)) pick = lo[f] + z;           // first hit from the left / last hit from the right
          if (w > amw) { amw = w; am = lo[f] + z; }                              // esl_vec_IArgMax: the first maximum
        }
      } else {
        for (int x = 0; x < nm; x++) {
          const Tup th = unpack_tup(Skey[Sstack[x]]);
          const int v = f == 0 ? th.i : f == 1 ? th.k : f == 2 ? th.j : th.m;
          int w = 0;
          for (int y = 0; y < nm; y++) {
            const Tup tu = unpack_tup(Skey[Sstack[y]]);
            const int vu = f == 0 ? tu.i : f == 1 ? tu.k : f == 2 ? tu.j : tu.m;
            if (vu == v) w += Stcount[Sstack[y]];
          }
          if (w >= thr && (pick < 0 || (leftmost ? v < pick : v > pick))) pick = v;
          if (w > amw || (w == amw && v < am)) { amw = w; am = v; }
        }
      }
      best[f] = pick >= 0 ? pick : am;
    }
    if (best[0] > best[2] || best[1] > best[3]) continue;
    if (nsig >= capSig) { status = 6; break; }
    Ssig_i[nsig] = best[0]; Ssig_j[nsig] = best[2]; Ssig_p[nsig] = (float)ninc / (float)200;
    nsig++;
  }
  if (status == 0) {
    // order by start (stable insertion sort), then drop dominated clusters
    for (int x = 1; x < nsig; x++) {
      const int ti = Ssig_i[x], tj = Ssig_j[x]; const float tp = Ssig_p[x];
      int y = x - 1;
      while (y >= 0 && Ssig_i[y] > ti) { Ssig_i[y + 1] = Ssig_i[y]; Ssig_j[y + 1] = Ssig_j[y]; Ssig_p[y + 1] = Ssig_p[y]; y--; }
      Ssig_i[y + 1] = ti; Ssig_j[y + 1] = tj; Ssig_p[y + 1] = tp;
    }
    for (int d = 0; d < nsig; d++) Sdominated[d] = 0;
    for (int d = 0; d < nsig; d++)
      for (int d2 = d + 1; d2 < nsig; d2++) {
        const int nov = min(Ssig_j[d], Ssig_j[d2]) - max(Ssig_i[d], Ssig_i[d2]) + 1;
        if (nov == 0) break;
        const int n = min(Ssig_j[d] - Ssig_i[d] + 1, Ssig_j[d2] - Ssig_i[d2] + 1);
        if ((float)nov / (float)n >= 0.8f) { if (Ssig_p[d] > Ssig_p[d2]) Sdominated[d2] = 1; else Sdominated[d] = 1; }
      }
    for (int d = 0; d < nsig; d++) {
      if (Sdominated[d]) continue;
      if (out.nenv >= capEnv) { status = 7; break; }
      if (out.nenv < MRENV) { out.ei[out.nenv] = Ssig_i[d]; out.ej[out.nenv] = Ssig_j[d]; }
      if constexpr (BIG) { Senv[2 * out.nenv] = Ssig_i[d]; Senv[2 * out.nenv + 1] = Ssig_j[d]; }
      out.nenv++;
    }
  }
  if (status != 0) {              // a bookkeeping limit was hit: no cluster envelopes (k_mr_apply keeps the region whole), and it says so
    out.nenv = 0;
    for (int pos = 0; pos < Lr; pos++) n2[pos] = 0.0f;
  }
  out.status = status;
  a.out[e.mi] = out;
  if (a.dbg && lane == 0) { atomicAdd(&a.dbg[0], tk_walk); atomicAdd(&a.dbg[1], tk_close); atomicAdd(&a.dbg[2], wall_clock64() - tk2); atomicAdd(&a.dbg[3], 1ull); atomicAdd(&a.dbg[4], tk_dedupe); }
}
#undef DOM_IJ
#undef DOM_KM

// =========================================================================================
// list building: multidomain regions of every pair, in pair order
__global__ void __launch_bounds__(256) k_mr_count(const PairOut *__restrict__ pout, const RegionRec *__restrict__ raw, RegionPoolView pv, int64_t npairs,
                                                  int32_t *__restrict__ cnt)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi > npairs) return;
  int c = 0;
  if (pi < npairs && pout[pi].pass_fwd)
    for (int k = 0; k < pout[pi].ndom; k++) c += raw_region(raw, pv, pi, k).multi != 0;
  cnt[pi] = c;
}
__global__ void __launch_bounds__(256) k_mr_fill(const PairOut *__restrict__ pout, const RegionRec *__restrict__ raw, RegionPoolView pv, int64_t npairs,
                                                 const int32_t *__restrict__ off, MrRec *__restrict__ mr)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= npairs || !pout[pi].pass_fwd) return;
  int o = off[pi];
  for (int k = 0; k < pout[pi].ndom; k++) {
    const RegionRec r = raw_region(raw, pv, pi, k);
    if (!r.multi) continue;
    MrRec m; m.pair = (int32_t)pi; m.slot = k; m.ireg = r.ienv; m.jreg = r.jenv;
    mr[o++] = m;
  }
}
// The ensemble of a region depends on nothing but (profile, target length, the region's residues) -- the random stream is
// re-seeded for every region -- and amplicons share their conserved flanks, so the regions are memoised exactly like the
// envelopes (k_derep.hip: k_region_keys / k_region_resolve): only the first copy of each distinct region is sampled.
__global__ void __launch_bounds__(256) k_mr_ulist(int64_t nmr, const MrRec *__restrict__ mr, const int32_t *__restrict__ rep, const int32_t *__restrict__ is_uniq,
                                                  const int32_t *__restrict__ urank, int32_t *__restrict__ ulist, int32_t *__restrict__ ulen,
                                                  int32_t *__restrict__ mr_u)
{
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nmr) return;
  const int32_t r = rep[m] >= 0 ? rep[m] : (int32_t)m;
  mr_u[m] = urank[r];
  if (is_uniq[m]) { ulist[urank[m]] = (int32_t)m; ulen[urank[m]] = mr[m].jreg - mr[m].ireg + 1; }
}
__global__ void __launch_bounds__(256) k_mr_reorder(int64_t nu, int64_t nmr, const int32_t *__restrict__ newpos, const int32_t *__restrict__ ulist_in,
                                                    int32_t *__restrict__ ulist_out, int32_t *__restrict__ mr_u)
{
  const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (x < nu) ulist_out[newpos[x]] = ulist_in[x];
  if (x < nmr) mr_u[x] = newpos[mr_u[x]];
}
// A pair's final envelope list: its regions in order, every clustered region replaced by its envelopes.  Walked twice with the same
// code -- once to count, once to write straight into the profile-grouped list -- so that nothing needs a per-pair buffer of fixed size.
template <class Emit>
DEV int walk_region_list(const RegionListArgs &a, int64_t pi, int nraw, unsigned long long *nfail, unsigned long long *nenv, unsigned long long *kind, Emit emit)
{
  int n = 0, mk = a.mr_off ? a.mr_off[pi] : 0;
  for (int k = 0; k < nraw; k++) {
    const RegionRec r = raw_region(a.raw, a.pv, pi, k);
    if (!r.multi || !a.mr_off) { emit(n++, r); continue; }
    const MrOut o = a.mrout[a.mr_u[mk]];
    if (o.status != 0) {
      // the region's matrix could not be sampled (hmmsearch itself throws there; itsx_search reports it): the region stays ONE
      // envelope with null2 by expectation, multi < 0 = "a simple envelope that carries the multidomain flag"
      if (nfail) { (*nfail)++; kind[o.status & 7]++; }
      RegionRec c = r; c.multi = -1;
      emit(n++, c);
    } else {
      if (nenv) *nenv += (unsigned long long)o.nenv;
      const int32_t *env = o.big >= 0 ? a.envpool + o.big : nullptr;
      for (int z = 0; z < o.nenv; z++) {
        RegionRec c; c.pair = (int32_t)pi; c.multi = mk + 1;
        c.ienv = (env ? env[2 * z] : o.ei[z]) + r.ienv - 1; c.jenv = (env ? env[2 * z + 1] : o.ej[z]) + r.ienv - 1;
        emit(n++, c);
      }
    }
    mk++;
  }
  return n;
}
__global__ void __launch_bounds__(256) k_region_list_count(RegionListArgs a, int32_t *__restrict__ cnt, unsigned long long *__restrict__ counters)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi > a.npairs) return;
  int n = 0;
  if (pi < a.npairs && a.pout[pi].pass_fwd) {
    unsigned long long nfail = 0, nenv = 0, kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    n = walk_region_list(a, pi, a.pout[pi].ndom, &nfail, &nenv, kind, [](int, const RegionRec &) {});
    a.pout[pi].nregions_raw = a.pout[pi].ndom;      // the fill pass walks the raw regions again
    a.pout[pi].ndom = n;
    if (nfail) { atomicAdd(&counters[0], nfail); for (int z = 1; z < 8; z++) if (kind[z]) atomicAdd(&counters[2 + z], kind[z]); }
    if (nenv) atomicAdd(&counters[1], nenv);
  }
  cnt[pi] = n;
}
__global__ void __launch_bounds__(256) k_region_list_fill(RegionListArgs a, const int64_t *__restrict__ pair_region0, RegionRec *__restrict__ out)
{
  const int64_t pi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= a.npairs || !a.pout[pi].pass_fwd) return;
  const int64_t o = pair_region0[pi];
  walk_region_list(a, pi, a.pout[pi].nregions_raw, nullptr, nullptr, nullptr, [&](int n, const RegionRec &r) { out[o + n] = r; });
}
__global__ void __launch_bounds__(256) k_mr_overflowed(const MrOut *__restrict__ out, int64_t nu, int32_t *__restrict__ list, unsigned long long *__restrict__ n)
{
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nu) return;
  const int st = out[u].status;
  if (st == 2 || st == 3 || st == 4 || st == 6 || st == 7) list[atomicAdd(n, 1ULL)] = (int32_t)u;
}

void launch_mr_count(const PairOut *pout, const RegionRec *raw, RegionPoolView pv, int64_t npairs, int32_t *cnt, hipStream_t st)
{
  hipLaunchKernelGGL(k_mr_count, dim3((unsigned)((npairs + 1 + 255) / 256)), dim3(256), 0, st, pout, raw, pv, npairs, cnt);
}
void launch_mr_fill(const PairOut *pout, const RegionRec *raw, RegionPoolView pv, int64_t npairs, const int32_t *off, MrRec *mr, hipStream_t st)
{
  if (npairs <= 0) return;
  hipLaunchKernelGGL(k_mr_fill, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, pout, raw, pv, npairs, off, mr);
}
void launch_region_list_count(const RegionListArgs &a, int32_t *cnt, unsigned long long *counters, hipStream_t st)
{
  hipLaunchKernelGGL(k_region_list_count, dim3((unsigned)((a.npairs + 1 + 255) / 256)), dim3(256), 0, st, a, cnt, counters);
}
void launch_region_list_fill(const RegionListArgs &a, const int64_t *pair_region0, RegionRec *out, hipStream_t st)
{
  if (a.npairs <= 0) return;
  hipLaunchKernelGGL(k_region_list_fill, dim3((unsigned)((a.npairs + 255) / 256)), dim3(256), 0, st, a, pair_region0, out);
}
void launch_mr_overflowed(const MrOut *out, int64_t nu, int32_t *list, unsigned long long *n, hipStream_t st)
{
  if (nu > 0) hipLaunchKernelGGL(k_mr_overflowed, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, st, out, nu, list, n);
}

void launch_mr_ulist(int64_t nmr, const MrRec *mr, const int32_t *rep, const int32_t *is_uniq, const int32_t *urank, int32_t *ulist, int32_t *ulen,
                     int32_t *mr_u, hipStream_t st)
{
  if (nmr <= 0) return;
  hipLaunchKernelGGL(k_mr_ulist, dim3((unsigned)((nmr + 255) / 256)), dim3(256), 0, st, nmr, mr, rep, is_uniq, urank, ulist, ulen, mr_u);
}
void launch_mr_reorder(int64_t nu, int64_t nmr, const int32_t *newpos, const int32_t *ulist_in, int32_t *ulist_out, int32_t *mr_u, hipStream_t st)
{
  const int64_t n = nu > nmr ? nu : nmr;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_mr_reorder, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nu, nmr, newpos, ulist_in, ulist_out, mr_u);
}
// one_count > 0: the launch's regions [one_first, one_first + one_count) are walked one per wave with their matrices in LDS
// (a.lds_bytes of it per wave); otherwise the waves of 64 (4) regions walk them where k_mr_fwd left them
void launch_mr_ensemble(const MrArgs &a, int nwaves, int wave0, hipStream_t st, int64_t one_first, int64_t one_count)
{
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_mr_fwd, dim3(nwaves), dim3(64), 0, st, a, wave0);
  if (one_count > 0) hipLaunchKernelGGL((k_mr_trace<false, true>), dim3((unsigned)one_count), dim3(64), (size_t)a.lds_bytes, st, a, (int)one_first);
  else hipLaunchKernelGGL(k_mr_trace<false>, dim3(nwaves), dim3(64), 0, st, a, wave0);
}
void launch_mr_ensemble_big(const MrArgs &a, int nwaves, hipStream_t st)
{
  if (nwaves <= 0) return;
  hipLaunchKernelGGL(k_mr_fwd, dim3(nwaves), dim3(64), 0, st, a, 0);
  hipLaunchKernelGGL(k_mr_trace<true>, dim3(nwaves), dim3(64), 0, st, a, 0);
}
}  // namespace itsx
