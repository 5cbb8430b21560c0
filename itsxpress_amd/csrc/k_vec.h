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

}  // namespace itsx
