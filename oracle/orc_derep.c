/*
 * orc_derep.c -- ORACLE (test infrastructure only): full-length exact
 * dereplication with reverse-complement matching, restating
 *   vsearch --fastx_uniques IN --fastaout rep.fa --uc uc.txt --strand both
 * (reference call site itsxpress/SeqSample.py:106-116; vsearch >= 2.21.1 is an
 * un-vendored dependency, recipes/itsxpress/meta.yaml:37).
 *
 * Semantics pinned by the reference fixture tests/test_data/ex_tmpdir/uc.txt
 * (see tests/test_oracle_derep.py): reads are compared full-length,
 * case-insensitively, U == T, ambiguity symbols literal; in input order each read
 * joins the cluster whose seed equals it (+) or equals its reverse complement (-),
 * else it seeds a new cluster; reads shorter than --minseqlength (32) are dropped.
 * Consumer that defines "correct": Dedup.parse (itsxpress/SeqSample.py:542-562).
 *
 * Also holds a plain XXH64 (the engine hashes packed reads with XXH64 on the
 * device; known-answer tests pin both against the xxhash library).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* complement on HMMER DNA digital codes: A C G T - R Y M K S W H B V D N */
static const uint8_t COMP[16] = { 3, 2, 1, 0, 4, 6, 5, 8, 7, 9, 10, 14, 13, 12, 11, 15 };

#define P1 11400714785074694791ULL
#define P2 14029467366897019727ULL
#define P3 1609587929392839161ULL
#define P4 9650029242287828579ULL
#define P5 2870177450012600261ULL
static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xround(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); acc *= P1; return acc; }
static inline uint64_t xmerge(uint64_t acc, uint64_t v) { v = xround(0, v); acc ^= v; acc = acc * P1 + P4; return acc; }

uint64_t orc_xxh64(const void *data, int64_t len, uint64_t seed)
{
  const uint8_t *p = (const uint8_t *)data, *end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    do {
      v1 = xround(v1, rd64(p)); p += 8; v2 = xround(v2, rd64(p)); p += 8;
      v3 = xround(v3, rd64(p)); p += 8; v4 = xround(v4, rd64(p)); p += 8;
    } while (p + 32 <= end);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
  } else h = seed + P5;
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= xround(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; p++; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

int64_t orc_derep(const uint8_t *codes, const int64_t *offsets, int64_t n, int strand_both, int minlen,
                  int64_t *rep_of, int8_t *strand)
{
  /* canonical form of each read: the lexicographically smaller of (read, revcomp(read)) */
  int64_t total = offsets[n];
  uint8_t *canon = (uint8_t *)malloc((size_t)total + 1);
  uint8_t *rc = NULL; int64_t rccap = 0;
  uint64_t tsize = 16; while (tsize < (uint64_t)n * 2 + 16) tsize <<= 1;
  int64_t *table = (int64_t *)malloc(sizeof(int64_t) * tsize);
  for (uint64_t i = 0; i < tsize; i++) table[i] = -1;
  int64_t nclusters = 0;
  for (int64_t r = 0; r < n; r++) {
    int64_t L = offsets[r + 1] - offsets[r];
    const uint8_t *s = codes + offsets[r];
    uint8_t *c = canon + offsets[r];
    if (L < minlen) { rep_of[r] = -1; strand[r] = 0; memcpy(c, s, (size_t)L); continue; }
    memcpy(c, s, (size_t)L);
    if (strand_both) {
      if (L > rccap) { rccap = L + 64; rc = (uint8_t *)realloc(rc, (size_t)rccap); }
      for (int64_t i = 0; i < L; i++) rc[i] = COMP[s[L - 1 - i]];
      if (memcmp(rc, s, (size_t)L) < 0) memcpy(c, rc, (size_t)L);
    }
    uint64_t h = orc_xxh64(c, L, 0) & (tsize - 1);
    for (;;) {
      int64_t e = table[h];
      if (e < 0) { table[h] = r; rep_of[r] = r; strand[r] = 1; nclusters++; break; }
      int64_t Le = offsets[e + 1] - offsets[e];
      if (Le == L && memcmp(canon + offsets[e], c, (size_t)L) == 0) {
        rep_of[r] = e;
        strand[r] = (memcmp(codes + offsets[e], s, (size_t)L) == 0) ? 1 : -1;
        break;
      }
      h = (h + 1) & (tsize - 1);
    }
  }
  free(canon); free(rc); free(table);
  return nclusters;
}
