// k_util.hip -- small device utilities: multi-level exclusive scan, device detmath probe.
#include <algorithm>
#include "engine.h"
#include "k_api.h"
#include "detmath.h"

namespace itsx {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;   // 2048 elements per block

// per-block exclusive scan; block totals to sums[blockIdx.x]
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                           int64_t n, int32_t *__restrict__ sums)
{
  __shared__ int32_t wsum[SCAN_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t tsum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n) ? in[base + i] : 0; tsum += v[i]; }
  // inclusive scan of tsum across the wave
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int32_t woff = 0;
  for (int w = 0; w < wid; w++) woff += wsum[w];
  int32_t run = woff + inc - tsum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
  if (threadIdx.x == SCAN_BLOCK - 1 && sums) sums[blockIdx.x] = run;
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(int32_t *__restrict__ out, int64_t n, const int32_t *__restrict__ offs)
{
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  const int32_t o = offs[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) out[base + i] += o;
}

int64_t scan_tmp_elems(int64_t n)
{
  int64_t tot = 0;
  while (n > SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; tot += n + 8; }
  return tot + 8;
}

void launch_exclusive_scan(const int32_t *in, int32_t *out, int64_t n, int32_t *tmp, hipStream_t st)
{
  if (n <= 0) return;
  const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, n, (int32_t *)nullptr);
    return;
  }
  int32_t *sums = tmp;
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, n, sums);
  launch_exclusive_scan(sums, sums, nb, tmp + nb + 8, st);   // in-place is safe: each block reads its tile before writing it
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, out, n, sums);
}

__global__ void k_detmath(const double *__restrict__ x, int64_t n, double *__restrict__ ol, double *__restrict__ oe)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ol[i] = det_log(x[i]); oe[i] = det_exp(x[i]); }
}
__global__ void k_logf_fast(const float *__restrict__ x, int64_t n, const LogTab *__restrict__ tab, float *__restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = det_logf_fast(x[i], tab);
}
void launch_logf_fast(const float *x, int64_t n, const LogTab *tab, float *out, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_logf_fast, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, tab, out);
}
void launch_detmath(const double *x, int64_t n, double *ol, double *oe, hipStream_t st)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_detmath, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, ol, oe);
}

}  // namespace itsx

// ------------------------------------------------------------------ PMC calibration streams (test hook)
// Streams of known size in the slab access patterns of k_float.hip, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be
// calibrated for them (MI355X_MICROARCH.md, HBM: only 16-B-per-lane streams are calibrated there).  One wave per block;
// wave w walks rows [w R, (w + 1) R) of a [row][6 fields][64 lanes] float plane, one row after the other, like k_decode.
namespace itsx {
__global__ void __launch_bounds__(64) k_calib_read4(const float *__restrict__ slab, int64_t R, int nf, float *__restrict__ out)
{
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  float acc = 0.f;
  for (int64_t r = 0; r < R; r++)
    for (int f = 0; f < nf; f++) acc += slab[((r0 + r) * 6 + f) * 64 + lane];
  out[(int64_t)blockIdx.x * 64 + lane] = acc;
}
__global__ void __launch_bounds__(64) k_calib_write4(float *__restrict__ slab, int64_t R, int nf)
{
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  for (int64_t r = 0; r < R; r++)
    for (int f = 0; f < nf; f++) slab[((r0 + r) * 6 + f) * 64 + lane] = (float)(r + f);
}
__global__ void __launch_bounds__(64) k_calib_read16(const float4 *__restrict__ slab, int64_t R, float *__restrict__ out)
{
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  float acc = 0.f;
  for (int64_t r = 0; r < R; r++) { const float4 v = slab[(r0 + r) * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
  out[(int64_t)blockIdx.x * 64 + lane] = acc;
}
// ---- VALU issue-rate probe (profiles/round5_valu_issue.md): chains of ONE instruction class with no dependence between
// consecutive instructions (16 accumulators in rotation), `waves` waves per SIMD on every CU, timed per wave with s_memtime.
// What bench.py's valu_issue_frac prices a kernel's instruction mix with.
template <int OP>
__device__ __forceinline__ void issue16(float (&x)[32], float a, float b)
{
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(float2 *)&x[2 * i]) : "v"(*(float2 *)&x[0]));   // (reads pair 0: written once per 16)
    else if (OP == 2) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    else if (OP == 3) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    else if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(float2 *)&x[2 * i]) : "v"(*(float2 *)&x[0]));
    else if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(float2 *)&x[2 * i]) : "v"(*(float2 *)&x[0]));
    else if (OP == 6) asm volatile("s_nop 0");
    else if (OP == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    else if (OP == 8) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(*(float2 *)&x[2 * i]) : "v"(*(float2 *)&x[(2 * i + 2) & 31]));
    else if (OP == 9) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
  }
}
// ---- scalar-cache probe: what k_fwd_bound asks of it per pair of nodes -- 64 B of wave-uniform transitions through two
// s_load_dwordx8, waited for one step later -- alone (OP 10) and under the kernel's own ~12 packed instructions per step (OP 11)
template <int OP>
__device__ __forceinline__ void sload16(float (&x)[32], const float *tab)
{
#pragma unroll
  for (int i = 0; i < 16; i++) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 s[36:43], %0, %1\n\ts_load_dwordx8 s[44:51], %0, %2"
                 :: "s"(tab), "n"(i * 64), "n"(i * 64 + 32)
                 : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory");
    if (OP == 11) {
#pragma unroll
      for (int k = 0; k < 12; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(float2 *)&x[2 * ((i + k) & 15)]) : "v"(*(float2 *)&x[0]));
    }
  }
}
template <int OP>
__global__ void __launch_bounds__(1024) k_issue(int iters, unsigned long long *__restrict__ ticks, float *__restrict__ sink)
{
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; i++) x[i] = 1.0f + 1e-6f * (float)(threadIdx.x + i);
  const float a = 0.999999f, b = 1e-7f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (OP >= 10) { for (int it = 0; it < iters; it++) { sload16<OP>(x, sink); sload16<OP>(x, sink); sload16<OP>(x, sink); sload16<OP>(x, sink); } }
  else for (int it = 0; it < iters; it++) { issue16<OP>(x, a, b); issue16<OP>(x, a, b); issue16<OP>(x, a, b); issue16<OP>(x, a, b); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i++) s += x[i];
  if ((threadIdx.x & 63) == 0) ticks[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 12345.678f) sink[0] = s;
}
void launch_issue(int op, int waves_per_simd, int iters, int blocks, unsigned long long *ticks, float *sink, hipStream_t st)
{
  // one block per CU with 4 x waves_per_simd waves and dynamic LDS no second block fits beside; past 4 waves per SIMD two blocks
  // per CU (a block holds at most 16 waves)
  const int per_cu = waves_per_simd > 4 ? 2 : 1;
  const dim3 g((unsigned)(blocks * per_cu)), b((unsigned)(256 * waves_per_simd / per_cu));
  const size_t lds = per_cu == 1 ? 96 * 1024 : 0;      // (two blocks of 12 / 16 waves per CU fill the chip exactly: the grid itself forces two on every CU)
  switch (op) {
    case 0: hipLaunchKernelGGL(k_issue<0>, g, b, lds, st, iters, ticks, sink); break;
    case 1: hipLaunchKernelGGL(k_issue<1>, g, b, lds, st, iters, ticks, sink); break;
    case 2: hipLaunchKernelGGL(k_issue<2>, g, b, lds, st, iters, ticks, sink); break;
    case 3: hipLaunchKernelGGL(k_issue<3>, g, b, lds, st, iters, ticks, sink); break;
    case 4: hipLaunchKernelGGL(k_issue<4>, g, b, lds, st, iters, ticks, sink); break;
    case 5: hipLaunchKernelGGL(k_issue<5>, g, b, lds, st, iters, ticks, sink); break;
    case 6: hipLaunchKernelGGL(k_issue<6>, g, b, lds, st, iters, ticks, sink); break;
    case 7: hipLaunchKernelGGL(k_issue<7>, g, b, lds, st, iters, ticks, sink); break;
    case 8: hipLaunchKernelGGL(k_issue<8>, g, b, lds, st, iters, ticks, sink); break;
    case 9: hipLaunchKernelGGL(k_issue<9>, g, b, lds, st, iters, ticks, sink); break;
    case 10: hipLaunchKernelGGL(k_issue<10>, g, b, lds, st, iters, ticks, sink); break;
    default: hipLaunchKernelGGL(k_issue<11>, g, b, lds, st, iters, ticks, sink); break;
  }
}

void launch_calib(int pattern, float *slab, int64_t nwaves, int64_t R, float *out, hipStream_t st)
{
  if (pattern == 0) hipLaunchKernelGGL(k_calib_read4, dim3((unsigned)nwaves), dim3(64), 0, st, slab, R, 6, out);
  else if (pattern == 1) hipLaunchKernelGGL(k_calib_read4, dim3((unsigned)nwaves), dim3(64), 0, st, slab, R, 5, out);
  else if (pattern == 2) hipLaunchKernelGGL(k_calib_write4, dim3((unsigned)nwaves), dim3(64), 0, st, slab, R, 6);
  else hipLaunchKernelGGL(k_calib_read16, dim3((unsigned)nwaves), dim3(64), 0, st, (const float4 *)slab, R, out);
}
}  // namespace itsx

// ------------------------------------------------------------------ packing reads on the device
// The boundary hands over ASCII bases; they are uploaded as they are -- in chunks of whole reads through pinned staging
// buffers (engine.hip: pack_and_upload) -- and packed here (2 bits per base, 16 bases per word, every read on a word
// boundary; non-ACGT symbols are 0 in the 2-bit plane and listed as (pos << 4 | code) exceptions in position order).
// One lane per 16-base word, a block per 64 consecutive reads (k_pack_words).  `raw` holds the bases of reads [r0, r1) only: byte
// raw_base of the whole read set is raw[0].
namespace itsx {
struct __attribute__((packed, aligned(1))) Raw16 { uint32_t a, b, c, d; };
// 16 bases in ONE load (the text is not aligned to anything: gfx950 takes the unaligned 16 bytes as they are), and -- when all of them are
// A / C / G / T in either case, the rule -- their codes by arithmetic: (c >> 1) & 3 is 0 1 3 2 for A C G T, x ^ (x >> 1) makes that 0 1 2 3.
// A 0x41, C 0x43, G 0x47, T 0x54: bits 7..5 of the upper-cased byte are 010 and its low five bits are one of 1, 3, 7, 20.
__device__ __forceinline__ bool pack16_plain(const uint8_t *p, uint32_t &w)
{
  const Raw16 v = *(const Raw16 *)p;
  const uint32_t d[4] = {v.a, v.b, v.c, v.d};
  bool plain = true;
  w = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const uint32_t c = (d[q] >> (8 * t)) & 0xffu, u = c & 0xdfu;
      plain = plain && ((u & 0xe0u) == 0x40u) && (((0x0010008au >> (u & 31u)) & 1u) != 0u);
      const uint32_t x = (c >> 1) & 3u;
      w |= (x ^ (x >> 1)) << (2 * (4 * q + t));
    }
  }
  return plain;
}
// A block takes PACK_RB consecutive reads: their text is one contiguous stretch (the reads are concatenated), and so are their words, so
// thread t of the block takes word t, t + 256, ... of that stretch whichever read it belongs to (a 6-step search over the group's 65 word
// offsets in LDS) -- every lane busy and consecutive lanes on consecutive 16-byte chunks, where a wave per read (rounds 1-5) kept 19-37 of
// its 64 lanes busy on merged amplicons and paid a read's bookkeeping per wave.
constexpr int PACK_RB = 64;
__global__ void __launch_bounds__(256) k_pack_words(const uint8_t *__restrict__ raw, int64_t raw_base, const int64_t *__restrict__ off,
                                                    const int64_t *__restrict__ woff, int64_t r0, int64_t r1, const int8_t *__restrict__ lut,
                                                    uint32_t *__restrict__ words, int32_t *__restrict__ excnt, long long *__restrict__ first_bad)
{
  __shared__ int8_t code[256];
  __shared__ int32_t s_orel[PACK_RB + 1], s_wrel[PACK_RB + 1];        // byte / word offsets of the group's reads relative to its first
  __shared__ int32_t s_ne[PACK_RB];
  __shared__ int32_t s_bad;
  code[threadIdx.x] = lut[threadIdx.x];
  const int tid = (int)threadIdx.x;
  const int64_t ngroups = (r1 - r0 + PACK_RB - 1) / PACK_RB;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t first = r0 + g * PACK_RB;
    const int nr = (int)(r1 - first < PACK_RB ? r1 - first : PACK_RB);
    const int64_t o0 = off[first], w0 = woff[first];      // (64 reads of at most 65 535 bases: the relative offsets fit 32 bits)
    __syncthreads();                                       // (the group before this one is done with the arrays; `code` is there)
    if (tid <= nr) { s_orel[tid] = (int32_t)(off[first + tid] - o0); s_wrel[tid] = (int32_t)(woff[first + tid] - w0); }
    if (tid < PACK_RB) s_ne[tid] = 0;
    if (tid == 0) s_bad = PACK_RB;
    __syncthreads();
    const uint8_t *text = raw + (o0 - raw_base);
    const int Wb = s_wrel[nr];
    for (int w = tid; w < Wb; w += 256) {
      int j = 0;                                           // the read of word w: s_wrel[j] <= w < s_wrel[j + 1]
#pragma unroll
      for (int step = PACK_RB / 2; step >= 1; step >>= 1) if (j + step < nr && s_wrel[j + step] <= w) j += step;
      const int k = w - s_wrel[j];
      const int o = s_orel[j], L = s_orel[j + 1] - o;
      const int base = k * 16, m = L - base < 16 ? L - base : 16;
      uint32_t wd = 0;
      // a full word: its 16 bases in ONE load and, when all of them are A / C / G / T in either case, their codes by arithmetic
      const bool plain = m == 16 && pack16_plain(text + o + base, wd);
      if (!plain) {
        wd = 0;
        int ne = 0; bool bad = false;
        for (int t = 0; t < m; t++) {
          const int c = code[text[o + base + t]];
          if (c < 0) bad = true;
          else if (c <= 3) wd |= (uint32_t)c << (2 * t);
          else ne++;
        }
        if (ne) atomicAdd(&s_ne[j], ne);
        if (bad) atomicMin(&s_bad, j);
      }
      words[w0 + w] = wd;
    }
    __syncthreads();
    if (tid < nr) excnt[first + tid] = s_ne[tid];
    if (tid == 0 && s_bad < PACK_RB) atomicMin(first_bad, (long long)(first + s_bad));
  }
}
// exceptions are rare: one thread per read that has any, positions ascending.  exstart = exclusive scan of excnt over THIS
// chunk of reads; ebase[0] = exceptions of all chunks before it, ebase[1] = set when the list would outgrow `ecap`
__global__ void __launch_bounds__(256) k_pack_exc(const uint8_t *__restrict__ raw, int64_t raw_base, const int64_t *__restrict__ off, int64_t r0, int64_t r1,
                                                  const int8_t *__restrict__ lut, const int32_t *__restrict__ excnt, const int32_t *__restrict__ exstart,
                                                  long long *__restrict__ ebase, int64_t ecap, int64_t *__restrict__ excoff, uint32_t *__restrict__ exc)
{
  const int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= r1) return;
  const int64_t g = (int64_t)ebase[0] + exstart[r];
  excoff[r] = g;
  const int ne = excnt[r];
  if (ne == 0) return;
  if (g + ne > ecap) { ebase[1] = 1; return; }
  const int64_t o = off[r] - raw_base;
  const int L = (int)(off[r + 1] - off[r]);
  uint32_t *e = exc + g;
  // (sixteen bases at a time, and only a chunk that is not all A / C / G / T is taken apart: a read with one N in 440 bases cost its
  // thread -- and the 63 beside it -- 440 dependent byte loads)
  for (int base = 0; base < L; base += 16) {
    const int m = L - base < 16 ? L - base : 16;
    uint32_t w;
    if (m == 16 && pack16_plain(raw + o + base, w)) continue;
    for (int t = 0; t < m; t++) { const int c = lut[raw[o + base + t]]; if (c > 3) *e++ = ((uint32_t)(base + t) << 4) | (uint32_t)c; }
  }
}
__global__ void k_pack_advance(long long *__restrict__ ebase, const int32_t *__restrict__ chunk_total, int64_t *__restrict__ excoff_end)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) { ebase[0] += chunk_total[0]; excoff_end[0] = (int64_t)ebase[0]; }
}
void launch_pack(const uint8_t *raw, int64_t raw_base, const int64_t *off, const int64_t *woff, int64_t r0, int64_t r1, const int8_t *lut,
                 uint32_t *words, int32_t *excnt, long long *first_bad, hipStream_t st)
{
  if (r1 <= r0) return;
  hipLaunchKernelGGL(k_pack_words, dim3((unsigned)std::min<int64_t>((r1 - r0 + PACK_RB - 1) / PACK_RB, 1 << 20)), dim3(256), 0, st, raw, raw_base, off, woff, r0, r1, lut, words, excnt, first_bad);
}
void launch_pack_exc(const uint8_t *raw, int64_t raw_base, const int64_t *off, int64_t r0, int64_t r1, const int8_t *lut, const int32_t *excnt,
                     const int32_t *exstart, long long *ebase, int64_t ecap, int64_t *excoff, uint32_t *exc, hipStream_t st)
{
  if (r1 > r0)
    hipLaunchKernelGGL(k_pack_exc, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, st, raw, raw_base, off, r0, r1, lut, excnt, exstart, ebase, ecap, excoff, exc);
  hipLaunchKernelGGL(k_pack_advance, dim3(1), dim3(64), 0, st, ebase, exstart + r1, excoff + r1);
}
}  // namespace itsx
