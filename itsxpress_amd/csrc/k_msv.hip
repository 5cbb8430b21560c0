// k_msv.hip -- stage B of the path: the MSV filter for every (representative, profile) pair.
//
// Replaces hmmsearch's first filter (HMMER p7_MSVFilter: 8-bit saturating multi-hit
// ungapped local score; reference call site itsxpress/SeqSample.py:191-209, threshold
// --F1 1e-6).  This is U x P x L x M byte-cell updates -- the largest cell count on the path.
//
// CDNA4 mapping (no MFMA: this is a max/add scan, not a contraction):
//   * one wave = ONE sequence x 64 profiles; lane = profile.  The residue stream is then
//     wave-uniform (scalar loads + scalar control flow), every lane runs the same row
//     update, and there is no length divergence inside a wave.
//   * each lane keeps its own profile's emission costs for A/C/G/T in 4 x 23 VGPRs and the DP
//     row in 23 VGPRs, two cells per register as packed int16 (v_pk_max_i16 / v_pk_add_i16):
//     the inner loop touches no LDS and no memory.  Degenerate residues (rare) take a path
//     that loads that code's costs from a table in HBM (coalesced, lane = profile).
//   * HMMER's unsigned arithmetic floors cells at 0; cells here are signed and unfloored,
//     which is equivalent because every cell is max'ed with xB >= 0 before it is used and
//     xE starts at 0.  The upper saturation cannot trigger before the overflow test fires.
//   * the P-value test is folded into a per-(length, profile) threshold on the final xJ
//     byte, computed on the host with the same double arithmetic hmmsearch uses.
#include "engine.h"
#include "k_api.h"

namespace itsx {

typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t u) { return __builtin_bit_cast(s2, u); }
__device__ __forceinline__ uint32_t as_u(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

#define MSV_ROW(EXPR_E, TAG)                                                                   \
  {                                                                                            \
    _Pragma("unroll") for (int r = MSV_REGS - 1; r >= 0; r--) {                                \
      const uint32_t prev = (r > 0) ? __builtin_amdgcn_alignbit(dp[r], dp[r - 1], 16) : (dp[0] << 16); \
      s2 sv = __builtin_elementwise_max(as_s2(prev), xBv);                                     \
      sv = sv + as_s2(EXPR_E);                                                                 \
      xEv = __builtin_elementwise_max(xEv, sv);                                                \
      dp[r] = as_u(sv);                                                                        \
    }                                                                                          \
    asm volatile("; msv row " TAG ::: "memory"); /* distinct tails: keeps the 4 variants from being merged behind 23 v_mov */ \
  }

__global__ void __launch_bounds__(64) k_msv(MsvArgs a)
{
  const int lane = threadIdx.x;
  const int g = blockIdx.x / a.nchunks;
  const int chunk = blockIdx.x - g * a.nchunks;
  const int p = g * 64 + lane;
  const int Ppad = a.G * 64;
  uint32_t eA[MSV_REGS], eC[MSV_REGS], eG[MSV_REGS], eT[MSV_REGS];
  const uint32_t *tb = a.etab + (size_t)g * 16 * MSV_REGS * 64 + lane;
#pragma unroll
  for (int r = 0; r < MSV_REGS; r++) {
    eA[r] = tb[(0 * MSV_REGS + r) * 64];
    eC[r] = tb[(1 * MSV_REGS + r) * 64];
    eG[r] = tb[(2 * MSV_REGS + r) * 64];
    eT[r] = tb[(3 * MSV_REGS + r) * 64];
  }
  const int bias = a.pbias[p], tec = a.ptec[p], tbm = a.ptbm[p];
  const int base = 190;
  const int s0 = chunk * a.seqs_per_wave;
  int s1 = s0 + a.seqs_per_wave; if (s1 > a.U) s1 = a.U;
  // per-sequence metadata is a chain of dependent scalar loads (sorted position -> unique -> read -> offsets);
  // the next sequence's chain is started before the current sequence's rows so its latency is hidden
  struct Meta { int L, nexc, tjb; int64_t wo, eo; uint32_t w0; };
  auto fetch = [&](int s) {
    Meta m;
    const int u = uni(a.sorted_uniq[s]);
    const int r = uni(a.seed_read[u]);
    m.L = uni(a.rd.len[r]);
    m.wo = a.rd.woff[r];
    m.eo = a.rd.excoff[r];
    m.nexc = uni((int)(a.rd.excoff[r + 1] - m.eo));
    const int Lt = m.L < a.Lcap ? m.L : a.Lcap - 1;
    m.tjb = uni(a.tjb[Lt]);
    m.w0 = (uint32_t)uni((int)a.rd.words[m.wo]);
    return m;
  };
  Meta nx = fetch(s0 < s1 ? s0 : 0);
  for (int s = s0; s < s1; s++) {
    const Meta cur = nx;
    if (s + 1 < s1) nx = fetch(s + 1);
    const int L = cur.L;
    const int64_t eo = cur.eo;
    const int nexc = cur.nexc;
    const int Lt = L < a.Lcap ? L : a.Lcap - 1;
    const int tjbm = cur.tjb + tbm;
    int xJ = 0;
    int xB = base - tjbm; xB = xB < 0 ? 0 : xB;
    uint32_t dp[MSV_REGS];
#pragma unroll
    for (int i = 0; i < MSV_REGS; i++) dp[i] = 0;
    int ovf = 0;
    int ei = 0;
    int next_exc = nexc > 0 ? uni((int)(a.rd.exc[eo] >> 4)) : 0x7fffffff;
    const uint32_t *wp = a.rd.words + cur.wo;
    uint32_t wnext = cur.w0;
    for (int i0 = 0; i0 < L; i0 += 16) {
      uint32_t w = wnext;
      if (i0 + 16 < L) wnext = (uint32_t)uni((int)wp[(i0 >> 4) + 1]);     // the next 16 bases fly during these 16 rows
      int cnt = L - i0; cnt = cnt > 16 ? 16 : cnt;
      for (int t = 0; t < cnt; t++) {
        const s2 xBv = as_s2((uint32_t)xB * 0x10001u);
        s2 xEv = as_s2(0u);
        if (i0 + t == next_exc) {
          const int code = uni((int)(a.rd.exc[eo + ei] & 15u));
          ei++;
          next_exc = ei < nexc ? uni((int)(a.rd.exc[eo + ei] >> 4)) : 0x7fffffff;
          const uint32_t *tc = tb + (size_t)code * MSV_REGS * 64;
          uint32_t ex[MSV_REGS];
#pragma unroll
          for (int q = 0; q < MSV_REGS; q++) ex[q] = tc[q * 64];
          MSV_ROW(ex[r], "degenerate")
        } else {
          const int x = (int)(w & 3u);
          if (x == 0) MSV_ROW(eA[r], "A")
          else if (x == 1) MSV_ROW(eC[r], "C")
          else if (x == 2) MSV_ROW(eG[r], "G")
          else MSV_ROW(eT[r], "T")
        }
        w >>= 2;
        const uint32_t xe2 = as_u(xEv);
        int xE = (int)(xe2 & 0xffffu);
        const int xEh = (int)(xe2 >> 16);
        xE = xE > xEh ? xE : xEh;
        ovf |= (xE + bias >= 255);
        xE = xE > 255 ? 255 : xE;
        xE -= tec; xE = xE < 0 ? 0 : xE;
        xJ = xJ > xE ? xJ : xE;
        xB = (base > xJ ? base : xJ) - tjbm; xB = xB < 0 ? 0 : xB;
      }
    }
    const int thr = a.thr[(size_t)Lt * Ppad + p];
    const int pass = ovf | (xJ >= thr);
    const int xj = ovf ? 255 : xJ;
    a.res[(size_t)p * a.U + s] = (uint16_t)(pass ? (0x100 | xj) : 0);
  }
}

void launch_msv(const MsvArgs &a, hipStream_t st)
{
  if (a.U <= 0 || a.G <= 0) return;
  hipLaunchKernelGGL(k_msv, dim3((unsigned)(a.G * a.nchunks)), dim3(64), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// survivor list: (profile, sorted position) cells with the pass bit -> PairRec list grouped by
// profile.  Three small passes: per-chunk counts, per-profile scan of chunk counts, fill.
__global__ void __launch_bounds__(256) k_pair_count(const uint16_t *__restrict__ res, int32_t U, int32_t nchunks, int32_t *__restrict__ cnt)
{
  __shared__ int32_t ws[4];
  const int p = blockIdx.y, c = blockIdx.x;
  const int64_t base = (int64_t)p * U;
  int32_t n = 0;
  for (int i = threadIdx.x; i < CHUNK; i += 256) {
    const int s = c * CHUNK + i;
    if (s < U) n += (res[base + s] >> 8) & 1;
  }
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[(int64_t)p * nchunks + c] = ws[0] + ws[1] + ws[2] + ws[3];
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
  for (int i = 0; i < 8; i++) { v[i] = (s0 + i < U) ? res[base + s0 + i] : 0; n += (v[i] >> 8) & 1; }
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
    if (v[i] & 0x100) {
      PairRec pr; pr.useq = s0 + i; pr.prof = p; pr.xj = v[i] & 0xff; pr.L = ulen[s0 + i];
      pairs[o++] = pr;
    }
}
__global__ void __launch_bounds__(256) k_fill_ulen(int32_t U, const int32_t *__restrict__ sorted_uniq, const int32_t *__restrict__ seed_read,
                                                   const int32_t *__restrict__ len, int32_t *__restrict__ ulen)
{
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < U) ulen[s] = len[seed_read[sorted_uniq[s]]];
}

void launch_pair_count(const uint16_t *res, int32_t P, int32_t U, int32_t nchunks, int32_t *cnt, hipStream_t st)
{
  hipLaunchKernelGGL(k_pair_count, dim3((unsigned)nchunks, (unsigned)P), dim3(256), 0, st, res, U, nchunks, cnt);
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
