// k_msv.hip -- stage B of the path: the MSV filter for every (representative, profile) pair.
//
// Replaces hmmsearch's first filter (HMMER p7_MSVFilter: 8-bit saturating multi-hit
// ungapped local score; reference call site itsxpress/SeqSample.py:191-209, threshold
// --F1 1e-6).  This is U x P x L x M byte-cell updates -- the largest cell count on the path.
//
// CDNA4 mapping (no MFMA: this is a max/add scan, not a contraction):
//   * lane = SEQUENCE, profile = uniform over the block.  A block of 256 lanes takes 256 representatives (neighbours in
//     ascending length, so a wave's lanes finish together) through a.PB profiles (up to 32), one after the other; the
//     profile's emission table (16 residue codes x 23 registers' worth, 1.5 KB) sits in LDS, double-buffered, and every
//     lane fetches the 96 bytes of ITS residue's row with six ds_read_b128 -- lanes with the same residue read the same
//     banks (broadcast), the four nucleotides' rows start 8 banks apart.  There is no branch on the residue: the kernel
//     this one replaced (lane = profile, emission rows in 92 registers, one copy of the row code per nucleotide) spent a
//     sixth of its loop moving the row's registers back after the four-way branch, and could not be talked out of it.
//   * each lane keeps the DP row in 23 VGPRs, two cells per register as packed int16 (v_pk_max_i16 / v_pk_add_u16),
//     STRIPED: register r holds cells r (low half) and r + 23 (high half), so the cell before both halves of register r is
//     register r - 1 as it stands; only register 0 takes its predecessors from a shifted copy of the old register 22.
//     Three packed instructions per two cells (max with xB, add the emission, max into xE); the row is updated in place.
//   * degenerate residues come from the read's exception list (position, code): one compare per row against the lane's
//     next exception; their emission rows are in the same LDS table.
//   * HMMER's unsigned arithmetic floors cells at 0; cells here are signed and unfloored, which is equivalent because
//     every cell is max'ed with xB >= 0 before it is used and xE starts at 0.  HMMER returns at the first row whose
//     xE + bias reaches 255; here the row maximum of xE is kept and tested once at the end -- after that row nothing
//     else of the lane's state is used.
//   * the P-value test is folded into a per-(length, profile) threshold on the final xJ byte, computed on the host with
//     the same double arithmetic hmmsearch uses.
//   * results go to res[profile][sorted position]: consecutive lanes, consecutive halfwords.
#include <algorithm>
#include <cstdlib>
#include "engine.h"
#include "k_api.h"

namespace itsx {

constexpr int MSV_WT = 16;      // packed words a lane stages in LDS at a time (256 rows)
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// SHARE (k_share.hip): the block's sequences are chains that start at the same depth -- the rows before come from the state the parent
// chain saved (23 packed registers, xJ, xB, xEmax: MSV_STATE_Q uint4 per (node, profile)), and a chain saves its own state where a
// later chain branches off.  Integer arithmetic on the same operands: the xJ bytes are those of the unshared kernel.
template <bool SHARE>
__global__ void __launch_bounds__(256, 6) k_msv(MsvArgs a)
{
  __shared__ __attribute__((aligned(16))) uint32_t tab[2][16 * MSV_TW];
  // each lane's packed words in LDS (its own column: no barrier): the WHOLE read when the launch's longest fits 37 words (592 bases:
  // 37 KB a block, four blocks a CU) -- loaded once for all the block's profiles --, else 16 words (256 rows) at a time
  extern __shared__ uint32_t wst[];
  const bool whole = a.wtl > MSV_WT;
  const int s = a.k0 + blockIdx.x * 256 + threadIdx.x;
  const bool valid = s < a.k1;
  int L = 0, nexc = 0, tjb = 0, Lt = 0;
  const uint32_t *wp = a.rd.words;
  const uint32_t *ep = a.rd.exc;
  const int row0 = SHARE ? (a.sl.depth << a.sl.logB) : 0;
  if (valid) {
    const int r = a.seed_read[a.sorted_uniq[s]];
    L = a.rd.len[r];
    wp = a.rd.words + a.rd.woff[r];
    const int64_t eo = a.rd.excoff[r];
    nexc = (int)(a.rd.excoff[r + 1] - eo);
    ep = a.rd.exc + eo;
    Lt = L < a.Lcap ? L : a.Lcap - 1;
    tjb = a.tjb[Lt];
  }
  if (whole) {
    const int nw = (L + 15) >> 4;
    for (int j = 0; j < a.wtl; j++) wst[j * 256 + threadIdx.x] = (j < nw) ? wp[j] : 0u;
  }
  // two-sided sharing (round 6, k_share.hip: k_join_*): a chain whose suffix another representative of its length ends with too stops
  // after row `myend` = jlev * B and takes the rest of the maximum over paths from that representative's saved BACKWARD state
  // (k_msv_bwd below).  The filter is max-plus arithmetic on integers -- every cell is a maximum over paths of sums -- so
  //     final xJ = max over the state's components (cell_k + g_k, xJ + gJ, xB + gB, g0)
  // is the unshared kernel's xJ exactly, and xEmax follows from it (xJ = max(0, max_i xE_i - tec) by the recurrence).
  int myend = L, jlev = -1; int64_t jnode = 0;
  if constexpr (SHARE) {
    if (valid && a.sl.endrow) { myend = a.sl.endrow[s]; jlev = a.sl.jlev[s]; jnode = (int64_t)a.sl.jsrc[s] - a.sl.gnode_base; }
  }
  int Lw = valid ? myend : 0;                     // the wave runs to its longest chain (chains of one depth ascend by their last row)
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(Lw, d, 64); Lw = o > Lw ? o : Lw; }
  Lw = uni(Lw);
  const int Ppad = a.G * 64;
  const int p0 = a.pfirst + blockIdx.y * a.PB;
  const int np = a.plast;
  int p1 = p0 + a.PB; if (p1 > np) p1 = np;
  const int base = 190;
  for (int pj = p0; pj < p1; pj++) {
    const int p = a.plist ? uni(a.plist[pj]) : pj;
    uint32_t *tb = tab[(pj - p0) & 1];
    // every wave is past the barrier of profile p - 1, so none still reads the table of profile p - 2 in this buffer
    for (int i = threadIdx.x; i < 16 * MSV_TW; i += 256) tb[i] = a.etab[(size_t)p * 16 * MSV_TW + i];
    __syncthreads();
    const int bias = uni(a.pbias[p]), tec = uni(a.ptec[p]), tbm = uni(a.ptbm[p]);
    const int tjbm = tjb + tbm;
    int bm0 = base - tjbm; bm0 = bm0 < 0 ? 0 : bm0;          // xB while xJ is below the base
    int xJ = 0, xB = bm0, xEmax = 0;
    uint32_t dp[MSV_REGS];
#pragma unroll
    for (int i = 0; i < MSV_REGS; i++) dp[i] = 0;
    if constexpr (SHARE) {
      if (valid && a.sl.depth > 0) {
        // the parent chain's saved state this chain starts from (one 128-byte line; its number: one coalesced load per profile)
        const int64_t src_node = (int64_t)a.sl.src[s] - a.sl.node_base;
        const uint4 *src = (const uint4 *)a.sl.slots + (src_node * a.sl.Pb + (pj - a.pfirst)) * MSV_STATE_Q;
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const uint4 v = src[q];
          dp[4 * q] = v.x; dp[4 * q + 1] = v.y; dp[4 * q + 2] = v.z; if (4 * q + 3 < MSV_REGS) dp[4 * q + 3] = v.w;
        }
        const uint4 v = src[6];
        xJ = (int)v.x; xB = (int)v.y; xEmax = (int)v.z;
      }
    }
    int ei = 0;
    int next_exc = nexc > 0 ? (int)(ep[0] >> 4) : 0x7fffffff;        // (a chain has no exception above its start: k_share.hip)
    uint32_t w = 0;
    const int lend = myend;
    int jxJ = -1, jem = 0;                           // the joined result, if the chain joins
    for (int pos = row0; pos <= Lw; pos++) {
      const int sh = (pos & 15) * 2;
      if (sh == 0) {
        if constexpr (SHARE) {
          // a block boundary: the state after row `pos` (1-based) is what a chain that branches off here starts from
          if ((pos & ((1 << a.sl.logB) - 1)) == 0 && valid) {
            const int d = pos >> a.sl.logB;
            const unsigned long long m = a.sl.mask[s];
            if (pos > row0 && pos < L && d < 64 && ((m >> d) & 1ull)) {       // (a mask has no bit above the chain's last level)
              const int64_t node = (int64_t)a.sl.node0[s] + __popcll(m & ((1ull << d) - 1ull)) - a.sl.node_base;
              uint4 *dst = (uint4 *)a.sl.slots + (node * a.sl.Pb + (pj - a.pfirst)) * MSV_STATE_Q;
#pragma unroll
              for (int q = 0; q < 6; q++) dst[q] = make_uint4(dp[4 * q], dp[4 * q + 1], dp[4 * q + 2], (4 * q + 3 < MSV_REGS) ? dp[(4 * q + 3) % MSV_REGS] : 0u);
              dst[6] = make_uint4((uint32_t)xJ, (uint32_t)xB, (uint32_t)xEmax, 0u);
            }
            if (d == jlev) {
              // the join (saturating adds: a suffix that overflows stays an overflow)
              const uint4 *g = (const uint4 *)a.sl.gslots + (jnode * a.sl.Pb + (pj - a.pfirst)) * MSV_STATE_Q;
              s2 mx = as_s2(0x80008000u);
#pragma unroll
              for (int q = 0; q < 6; q++) {
                const uint4 v = g[q];
                mx = __builtin_elementwise_max(mx, __builtin_elementwise_add_sat(as_s2(dp[4 * q]), as_s2(v.x)));
                mx = __builtin_elementwise_max(mx, __builtin_elementwise_add_sat(as_s2(dp[4 * q + 1]), as_s2(v.y)));
                mx = __builtin_elementwise_max(mx, __builtin_elementwise_add_sat(as_s2(dp[4 * q + 2]), as_s2(v.z)));
                if (4 * q + 3 < MSV_REGS) mx = __builtin_elementwise_max(mx, __builtin_elementwise_add_sat(as_s2(dp[4 * q + 3]), as_s2(v.w)));
              }
              const uint4 t = g[6];
              int xf = (int)mx.x > (int)mx.y ? (int)mx.x : (int)mx.y;
              const int c1 = xJ + (int)t.x, c2 = xB + (int)t.y, c3 = (int)t.z;
              xf = xf > c1 ? xf : c1; xf = xf > c2 ? xf : c2; xf = xf > c3 ? xf : c3;
              jxJ = xf; jem = xf + tec;               // (xf + tec = the largest row maximum of the whole read when xf > 0); the chain may walk
                                                      // on for the states its prefix children start from
            }
          }
        }
        if (pos >= Lw) break;
        // the lane's packed words, 16 at a time (256 rows) through LDS: one contiguous 64-byte piece of the read per load instead of a
        // dword every 16 rows -- every one of those dwords cost a whole line once 49 k lanes' lines no longer fit the XCD's L2
        // (75.6 GB fetched per 1 M reads in round 4 against 0.1-0.4 GB of packed reads; profiles/round5_pmc_hbm_traffic_1M.md)
        if (!whole && ((pos & 255) == 0 || pos == row0)) {
          const int w0 = (pos >> 8) << 4, nw = (L + 15) >> 4;
#pragma unroll
          for (int j = 0; j < MSV_WT; j++) wst[j * 256 + threadIdx.x] = (w0 + j < nw) ? wp[w0 + j] : 0u;
        }
        w = wst[(whole ? (pos >> 4) : ((pos >> 4) & (MSV_WT - 1))) * 256 + threadIdx.x];
      }
      if (pos < lend) {
        int code = (int)((w >> sh) & 3u);
        if (pos == next_exc) {
          code = (int)(ep[ei] & 15u);
          ei++;
          next_exc = ei < nexc ? (int)(ep[ei] >> 4) : 0x7fffffff;
        }
        const uint4 *e4 = (const uint4 *)(tb + code * MSV_TW);
        uint32_t e[MSV_TW];
#pragma unroll
        for (int q = 0; q < MSV_TW / 4; q++) { const uint4 v = e4[q]; e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
        const s2 xBv = as_s2((uint32_t)xB * 0x10001u);
        const uint32_t wrap = dp[MSV_REGS - 1] << 16;
        s2 xEa = as_s2(0u), xEb = as_s2(0u);
#pragma unroll
        for (int r = MSV_REGS - 1; r >= 0; r--) {
          const uint32_t prev = (r > 0) ? dp[r > 0 ? r - 1 : 0] : wrap;
          const s2 sv = __builtin_elementwise_max(as_s2(prev), xBv) + as_s2(e[r]);
          if (r & 1) xEb = __builtin_elementwise_max(xEb, sv); else xEa = __builtin_elementwise_max(xEa, sv);
          dp[r] = as_u(sv);
        }
        const uint32_t xe2 = as_u(__builtin_elementwise_max(xEa, xEb));
        int xE = (int)(xe2 & 0xffffu);
        const int xEh = (int)(xe2 >> 16);
        xE = xE > xEh ? xE : xEh;
        xEmax = xEmax > xE ? xEmax : xE;
        xE -= tec;
        xJ = xJ > xE ? xJ : xE;
        xB = xJ - tjbm; xB = xB > bm0 ? xB : bm0;            // = max(max(base, xJ) - tjbm, 0)
      }
    }
    if (valid) {
      if (jxJ >= 0) { xJ = jxJ; xEmax = xEmax > jem ? xEmax : jem; }
      const int ovf = (xEmax + bias >= 255);
      const int thr = a.thr[(size_t)Lt * Ppad + p];
      const int pass = ovf | (xJ >= thr);
      const int xj = ovf ? 255 : xJ;
      a.res[(size_t)p * a.U + s] = (uint16_t)(pass ? (0x100 | xj) : 0);
    }
  }
}

// Round 6, two-sided sharing: the BACKWARD chains of the filter.  A row of the filter is a max-plus linear map of the state (46 cells, xJ,
// xB and the constant 0: xB = max(xJ - tjbm, bm0)); the final xJ is a max-plus linear functional of the last state.  Pulled back through
// the rows L, L - 1, ..., j B + 1 it becomes a vector g_j with  final xJ = max over components (state after row j B + g_j)  for EVERY
// prefix: it belongs to the suffix, and the uniques of one length that end alike share it.  Lane = Backward chain (k_share.hip: the
// uniques that save a state for somebody), the block's 256 chains start the same number of blocks from the end, a.PB profiles one after
// the other.  With a' the adjoint of a new row's value and x the row's residue:
//     aJ = max(gJ, gB - tjbm),  g0 <- max(g0, gB + bm0),  aE = aJ - tec
//     w_k = max(g_k, aE) + e_k(x)   (k = 46 .. 1),   g_k-1 <- w_k  (g_46 <- none),   gB <- max_k w_k,   gJ <- aJ
// in the forward kernel's register striping (register r = cells r, r + 23: the shift by one cell is a renaming but for the wrap register).
// Adds saturate: a suffix whose score overflows stays above every threshold instead of wrapping.  The rows a Backward chain walks lie
// above its read's last residue outside ACGT (k_share.hip), so it never meets an exception.
__global__ void __launch_bounds__(256, 6) k_msv_bwd(MsvArgs a)
{
  __shared__ __attribute__((aligned(16))) uint32_t tab[2][16 * MSV_TW];
  extern __shared__ uint32_t wst[];
  const bool whole = a.wtl > MSV_WT;
  const int s = a.k0 + blockIdx.x * 256 + threadIdx.x;      // backward position
  const bool valid = s < a.k1;
  int L = 0, tjb = 0, mysteps = 0;
  const uint32_t *wp = a.rd.words;
  if (valid) {
    const int r = a.seed_read[a.sorted_uniq[s]];
    L = a.rd.len[r];
    wp = a.rd.words + a.rd.woff[r];
    const int Lt = L < a.Lcap ? L : a.Lcap - 1;
    tjb = a.tjb[Lt];
    mysteps = a.sl.endrow[s];
  }
  const int nw = (L + 15) >> 4;
  if (whole) for (int j = 0; j < a.wtl; j++) wst[j * 256 + threadIdx.x] = (j < nw) ? wp[j] : 0u;
  int nsteps = mysteps;
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(nsteps, d, 64); nsteps = o > nsteps ? o : nsteps; }
  nsteps = (a.sl.dbg & 32) ? 0 : uni(nsteps);                // (diagnostic bit, ITSX_TEST_HOOKS=1 ITSX_PASSA_DBG=32: no rows)
  const int rd = a.sl.depth, logB = a.sl.logB;
  const int A = (L + (1 << logB) - 1) >> logB;
  const int top = (A - rd) << logB;                          // the row the chain's state stands after (virtual past L when rd = 0)
  const int p0 = a.pfirst + blockIdx.y * a.PB;
  int p1 = p0 + a.PB; if (p1 > a.plast) p1 = a.plast;
  const int base = 190;
  constexpr int NEG = -16384;
  const unsigned long long m = valid ? a.sl.mask[s] : 0ull;
  for (int pj = p0; pj < p1; pj++) {
    const int p = pj;
    uint32_t *tb = tab[(pj - p0) & 1];
    for (int i = threadIdx.x; i < 16 * MSV_TW; i += 256) tb[i] = a.etab[(size_t)p * 16 * MSV_TW + i];
    __syncthreads();
    const int tec = uni(a.ptec[p]), tbm = uni(a.ptbm[p]);
    const int tjbm = tjb + tbm;
    int bm0 = base - tjbm; bm0 = bm0 < 0 ? 0 : bm0;
    int gJ = 0, gB = NEG, g0 = NEG;
    uint32_t g[MSV_REGS];
#pragma unroll
    for (int i = 0; i < MSV_REGS; i++) g[i] = 0xC000C000u;   // NEG in both halves
    if (valid && rd > 0) {
      const int64_t src_node = (int64_t)a.sl.src[s] - a.sl.node_base;
      const uint4 *src = (const uint4 *)a.sl.slots + (src_node * a.sl.Pb + (pj - a.pfirst)) * MSV_STATE_Q;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const uint4 v = src[q];
        g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; if (4 * q + 3 < MSV_REGS) g[4 * q + 3] = v.w;
      }
      const uint4 v = src[6];
      gJ = (int)v.x; gB = (int)v.y; g0 = (int)v.z;
    }
    uint32_t w = 0; int wi = -1;
    for (int step = 0; ; step++) {
      if ((step & ((1 << logB) - 1)) == 0 && step > 0 && valid) {
        // a block boundary: the state after row top - step is what the Forward chains that end there take
        const int rdl = rd + (step >> logB);
        if (step <= mysteps && rdl < 64 && ((m >> rdl) & 1ull)) {
          const int64_t node = (int64_t)a.sl.node0[s] + __popcll(m & ((1ull << rdl) - 1ull)) - a.sl.node_base;
          uint4 *dst = (uint4 *)a.sl.slots + (node * a.sl.Pb + (pj - a.pfirst)) * MSV_STATE_Q;
#pragma unroll
          for (int q = 0; q < 6; q++) dst[q] = make_uint4(g[4 * q], g[4 * q + 1], g[4 * q + 2], (4 * q + 3 < MSV_REGS) ? g[(4 * q + 3) % MSV_REGS] : 0u);
          dst[6] = make_uint4((uint32_t)gJ, (uint32_t)gB, (uint32_t)g0, 0u);
        }
      }
      if (step >= nsteps) break;
      const int row = top - step;                             // this step pulls the state back through row `row` (1-based)
      if (row <= L && step < mysteps) {
        const int pos = row - 1;
        if ((pos >> 4) != wi) {
          wi = pos >> 4;
          if (whole) w = wst[wi * 256 + threadIdx.x];
          else w = wp[wi];                                    // (reads past 592 bases: a dword every 16 rows)
        }
        const int code = (int)((w >> ((pos & 15) * 2)) & 3u);
        const uint4 *e4 = (const uint4 *)(tb + code * MSV_TW);
        uint32_t e[MSV_TW];
#pragma unroll
        for (int q = 0; q < MSV_TW / 4; q++) { const uint4 v = e4[q]; e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
        int aJ = gB - tjbm; aJ = aJ > gJ ? aJ : gJ;
        const int n0 = gB + bm0; g0 = g0 > n0 ? g0 : n0;
        int aE = aJ - tec; aE = aE > 30000 ? 30000 : aE;
        const s2 aEv = as_s2((uint32_t)(aE & 0xffff) * 0x10001u);
        s2 mxa = as_s2(0xC000C000u), mxb = as_s2(0xC000C000u);
        // W[r] = (w of cells r, r + 23); the new g[r] is W[r + 1], the new g[22] = (W[0]'s high half, nothing)
        const s2 W0 = __builtin_elementwise_add_sat(__builtin_elementwise_max(as_s2(g[0]), aEv), as_s2(e[0]));
        mxa = __builtin_elementwise_max(mxa, W0);
#pragma unroll
        for (int r = 1; r < MSV_REGS; r++) {
          const s2 Wr = __builtin_elementwise_add_sat(__builtin_elementwise_max(as_s2(g[r]), aEv), as_s2(e[r]));
          if (r & 1) mxb = __builtin_elementwise_max(mxb, Wr); else mxa = __builtin_elementwise_max(mxa, Wr);
          g[r - 1] = as_u(Wr);
        }
        g[MSV_REGS - 1] = (as_u(W0) >> 16) | 0xC0000000u;
        const uint32_t m2 = as_u(__builtin_elementwise_max(mxa, mxb));
        const int lo = (int)(int16_t)(m2 & 0xffffu), hi = (int)(int16_t)(m2 >> 16);
        gB = lo > hi ? lo : hi;
        gJ = aJ;
      }
    }
  }
}

// lds_pad: dynamic LDS the launch asks for and never touches -- a cap on the blocks a CU holds (160 KB per CU), for the launch that
// runs beside another stream's kernels and must leave them their registers
void launch_msv(const MsvArgs &a0, hipStream_t st, int lds_pad)
{
  MsvArgs a = a0;
  if (a.k1 <= 0) { a.k0 = 0; a.k1 = a.U; }
  if (a.plast <= 0) { a.pfirst = 0; a.plast = a.plist ? a.nlist : a.P; }
  const int np = a.plast - a.pfirst, nk = a.k1 - a.k0;
  if (nk <= 0 || np <= 0) return;
  const dim3 grid((unsigned)((nk + 255) / 256), (unsigned)((np + a.PB - 1) / a.PB));
  // words a lane keeps in LDS: every word of the launch's longest read when that is at most 37 (four blocks of 37 + 3 KB fill a CU's
  // 160 KB exactly), else 16 at a time; ITSX_MSV_WHOLE=0: always 16 (A/B)
  static const bool allow_whole = !(sw_get("ITSX_MSV_WHOLE") && atoi(sw_get("ITSX_MSV_WHOLE")) == 0);
  const int need = (a.Lcap - 1 + 15) / 16;
  a.wtl = (allow_whole && need > MSV_WT && need <= 37) ? need : MSV_WT;
  // (a Backward chain walks a few blocks of its read through one to eight profiles: it takes its words from memory as it goes, a dword
  // every 16 rows -- the whole read in LDS first was 37 gathers per chain for ~6 words used, and 37 KB per block held a CU to 4 blocks)
  static const bool bwd_whole = sw_get("ITSX_MSV_BWD_WHOLE") && atoi(sw_get("ITSX_MSV_BWD_WHOLE")) != 0;
  if (a.share == 2 && !bwd_whole) a.wtl = MSV_WT;
  const size_t lds = std::max<size_t>((size_t)lds_pad, (a.share == 2 && !bwd_whole) ? 0 : (size_t)a.wtl * 256 * sizeof(uint32_t));
  if (a.share == 2) hipLaunchKernelGGL(k_msv_bwd, grid, dim3(256), lds, st, a);
  else if (a.share) hipLaunchKernelGGL(k_msv<true>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(k_msv<false>, grid, dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------
// survivor list: (profile, sorted position) cells with the pass bit -> PairRec list grouped by
// profile.  Three small passes: per-chunk counts, per-profile scan of chunk counts, fill.
// a cell is on the list when it passed (bit 8) or when only a chain below it did (bit 9, k_share.hip: the pair runs in pass A for its
// row states and is nobody's result: PairRec::xj = -1); real[p] counts the pairs that passed
__global__ void __launch_bounds__(256) k_pair_count(const uint16_t *__restrict__ res, int32_t U, int32_t nchunks, int32_t *__restrict__ cnt, int32_t *__restrict__ real)
{
  __shared__ int32_t ws[4], wr[4];
  const int p = blockIdx.y, c = blockIdx.x;
  const int64_t base = (int64_t)p * U;
  int32_t n = 0, nr = 0;
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const int s = c * CHUNK + i;
    if (s < U) { const int v = res[base + s]; n += (v & 0x300) != 0; nr += (v >> 8) & 1; }
  }
  for (int d = 32; d >= 1; d >>= 1) { n += __shfl_down(n, d, 64); nr += __shfl_down(nr, d, 64); }
  if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = n; wr[threadIdx.x >> 6] = nr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt[(int64_t)p * nchunks + c] = ws[0] + ws[1] + ws[2] + ws[3];
    const int32_t r = wr[0] + wr[1] + wr[2] + wr[3];
    if (r) atomicAdd(&real[p], r);
  }
}
// one block per profile: exclusive scan of its chunk counts (in place), total out
__global__ void __launch_bounds__(256) k_chunk_scan(int32_t *__restrict__ cnt, int32_t nchunks, int32_t *__restrict__ total)
{
  __shared__ int32_t ws[4];
  __shared__ int32_t carry;
  int32_t *row = cnt + (int64_t)blockIdx.x * nchunks;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b = 0; b < nchunks; b += 256) {
    const int i = b + threadIdx.x;
    const int32_t v = i < nchunks ? row[i] : 0;
    int32_t inc = v;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) ws[wid] = inc;
    __syncthreads();
    int32_t off = carry;
    for (int w = 0; w < wid; w++) off += ws[w];
    if (i < nchunks) row[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[blockIdx.x] = carry;
}
__global__ void __launch_bounds__(256) k_pair_fill(const uint16_t *__restrict__ res, int32_t U, int32_t nchunks, const int32_t *__restrict__ cnt,
                                                   const int64_t *__restrict__ seg_start, const int32_t *__restrict__ ulen,
                                                   PairRec *__restrict__ pairs)
{
  __shared__ int32_t ws[4];
  const int p = blockIdx.y, c = blockIdx.x;
  const int64_t base = (int64_t)p * U;
  const int64_t out0 = seg_start[p] + cnt[(int64_t)p * nchunks + c];
  // each thread owns 8 consecutive cells
  const int s0 = c * CHUNK + threadIdx.x * 8;
  uint16_t v[8]; int32_t n = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { v[i] = (s0 + i < U) ? res[base + s0 + i] : 0; n += (v[i] & 0x300) != 0; }
  int32_t inc = n;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) ws[wid] = inc;
  __syncthreads();
  int32_t off = 0;
  for (int w = 0; w < wid; w++) off += ws[w];
  int64_t o = out0 + off + inc - n;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (v[i] & 0x300) {
      PairRec pr; pr.useq = s0 + i; pr.prof = p; pr.xj = (v[i] & 0x100) ? (v[i] & 0xff) : -1; pr.L = ulen[s0 + i];
      pairs[o++] = pr;
    }
}
__global__ void __launch_bounds__(256) k_fill_ulen(int32_t U, const int32_t *__restrict__ sorted_uniq, const int32_t *__restrict__ seed_read,
                                                   const int32_t *__restrict__ len, int32_t *__restrict__ ulen)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < U) ulen[s] = len[seed_read[sorted_uniq[s]]];
}

void launch_pair_count(const uint16_t *res, int32_t P, int32_t U, int32_t nchunks, int32_t *cnt, int32_t *real, hipStream_t st)
{
  hipLaunchKernelGGL(k_pair_count, dim3((unsigned)nchunks, (unsigned)P), dim3(256), 0, st, res, U, nchunks, cnt, real);
}
void launch_chunk_scan(int32_t *cnt, int32_t P, int32_t nchunks, int32_t *total, hipStream_t st)
{
  hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)P), dim3(256), 0, st, cnt, nchunks, total);
}
void launch_pair_fill(const uint16_t *res, int32_t P, int32_t U, int32_t nchunks, const int32_t *cnt, const int64_t *seg_start,
                      const int32_t *ulen, PairRec *pairs, hipStream_t st)
{
  hipLaunchKernelGGL(k_pair_fill, dim3((unsigned)nchunks, (unsigned)P), dim3(256), 0, st, res, U, nchunks, cnt, seg_start, ulen, pairs);
}
void launch_fill_ulen(int32_t U, const int32_t *sorted_uniq, const int32_t *seed_read, const int32_t *len, int32_t *ulen, hipStream_t st)
{
  if (U <= 0) return;
  hipLaunchKernelGGL(k_fill_ulen, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, U, sorted_uniq, seed_read, len, ulen);
}

}  // namespace itsx
