// k_vec.h -- the 4-lane vector helpers (HMMER's SSE lanes, emulated per GPU lane) and the per-lane view of a packed read,
// shared by k_float.hip and k_ensemble.hip.
#pragma once
#include "engine.h"
#include "k_api.h"

namespace itsx {

#define DEV __device__ __forceinline__
static constexpr double kLn2 = 0.69314718055994529;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct V4 { f2 a, b; };
DEV V4 vset(float x) { V4 r; r.a = (f2){x, x}; r.b = (f2){x, x}; return r; }
DEV V4 vzero() { return vset(0.0f); }
DEV V4 vadd(V4 x, V4 y) { V4 r; r.a = x.a + y.a; r.b = x.b + y.b; return r; }
DEV V4 vmul(V4 x, V4 y) { V4 r; r.a = x.a * y.a; r.b = x.b * y.b; return r; }
DEV V4 vrsh(V4 v) { V4 r; r.a = (f2){0.0f, v.a.x}; r.b = (f2){v.a.y, v.b.x}; return r; }   // [0 a b c]
DEV V4 vlsh(V4 v) { V4 r; r.a = (f2){v.a.y, v.b.x}; r.b = (f2){v.b.y, 0.0f}; return r; }   // [b c d 0]
DEV float vhsum(V4 v) { return (v.a.x + v.a.y) + (v.b.x + v.b.y); }
DEV V4 vld(const float *p) { const f4 t = *(const f4 *)p; V4 r; r.a = (f2){t.x, t.y}; r.b = (f2){t.z, t.w}; return r; }
// The transition table is read through the constant address space: with a wave-uniform address the
// backend then selects scalar loads (s_load_dwordx4..x16 into SGPRs) even though the kernel also
// stores to global memory (a plain global pointer would get vector loads).
typedef const f4 __attribute__((address_space(4))) *cf4p;
DEV V4 vldc(const float *p) { const f4 t = *(cf4p)(uintptr_t)p; V4 r; r.a = (f2){t.x, t.y}; r.b = (f2){t.z, t.w}; return r; }
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A scalar zero the optimizer cannot see through.  Added to the (wave-uniform) transition-table
// pointer once per DP row, it keeps the 96 transition vectors as per-row scalar loads into SGPR
// operands; without it loop-invariant code motion hoists all 384 floats into VGPR/AGPRs.
DEV int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

// ---- per-lane view of one packed read ---------------------------------------------------
struct Seq {
  const uint32_t *w; const uint32_t *exc; int nexc; int L;
  DEV int code(int pos0) const   // digital code of base pos0 (0-based)
  {
    int x = (int)((w[pos0 >> 4] >> (2 * (pos0 & 15))) & 3u);
    for (int e = 0; e < nexc; e++) { const uint32_t v = exc[e]; if ((int)(v >> 4) == pos0) x = (int)(v & 15u); }
    return x;
  }
};
DEV Seq open_seq(const ReadsDev &rd, int read)
{
  Seq s; const int64_t wo = rd.woff[read], eo = rd.excoff[read];
  s.w = rd.words + wo; s.exc = rd.exc + eo; s.nexc = (int)(rd.excoff[read + 1] - eo); s.L = rd.len[read];
  return s;
}

// Residues of one read taken in order (ascending for Forward, descending for Backward), one per DP row.  Seq::code() costs a
// global load per row and per exception, each followed by a wait that -- vmcnt being in order -- also drains the row's six
// slab stores.  The stream keeps the current 16-base word and the next one in registers (the next word is requested a whole
// word ahead), and knows the position of the next exception, so a row costs a shift, a mask and one compare.
struct SeqStream {
  const uint32_t *w; const uint32_t *exc;
  uint32_t cur, nxt; int cur_i, nw, dir;
  int e, nexc, epos;                      // next exception in walking order: index, count, its position (-1: none left)
  DEV void open(const Seq &s, int first_pos, int direction)
  {
    w = s.w; exc = s.exc; nexc = s.nexc; dir = direction; nw = (s.L + 15) >> 4; if (nw < 1) nw = 1;
    cur_i = first_pos >> 4; if (cur_i >= nw) cur_i = nw - 1;
    cur = w[cur_i];
    const int ni = cur_i + dir;
    nxt = (ni >= 0 && ni < nw) ? w[ni] : 0u;
    if (dir > 0) { e = 0; while (e < nexc && (int)(exc[e] >> 4) < first_pos) e++; }
    else { e = nexc - 1; while (e >= 0 && (int)(exc[e] >> 4) > first_pos) e--; }
    epos = (e >= 0 && e < nexc) ? (int)(exc[e] >> 4) : -1;
    // nothing may still be in flight when the row loop is entered: a load pending at the loop's entry makes the compiler wait
    // for ALL memory operations (the previous row's stores included) at the first use in every iteration
    asm volatile("" : "+v"(cur), "+v"(nxt), "+v"(epos));
  }
  // positions must be taken in walking order, one step at a time, starting at first_pos
  DEV int get(int pos)
  {
    const int wi = pos >> 4;
    if (wi != cur_i) {                    // every 16th row: the word requested 16 rows ago moves up, the one after it is requested
      cur = nxt; cur_i = wi;
      asm volatile("" : "+v"(cur));        // stays a branch: as a select it would read nxt (and wait for its load) on every row
      const int ni = wi + dir;
      if (ni >= 0 && ni < nw) nxt = w[ni];
    }
    int x = (int)((cur >> (2 * (pos & 15))) & 3u);
    if (pos == epos) {                    // rare; everything loaded here is also consumed here, so no wait leaks into the common path
      x = (int)(exc[e] & 15u);
      e += dir;
      int np = -1;
      if (e >= 0 && e < nexc) np = (int)(exc[e] >> 4);
      epos = np;
      asm volatile("" : "+v"(epos), "+v"(x));
    }
    return x;
  }
};

}  // namespace itsx
