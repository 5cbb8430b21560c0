/* synthc.c -- the two byte-moving loops of tests/synth.py's ragged read generator (10 M reads of 300-580 bases are
 * 4.4 GB of text: numpy's gather + mask-select takes minutes for that).  Test/bench data preparation only. */
#include <stdint.h>
#include <string.h>

/* blob[offs[i] .. offs[i] + rlen[i]) = tmpl[ids[i]][0 .. rlen[i]) */
void synth_ragged_gather(const uint8_t *tmpl, int64_t lmax, const int64_t *ids, const int64_t *rlen, const int64_t *offs, int64_t n,
                         uint8_t *blob)
{
#pragma omp parallel for schedule(static, 4096)
  for (int64_t i = 0; i < n; i++) memcpy(blob + offs[i], tmpl + ids[i] * lmax, (size_t)rlen[i]);
}

/* reverse-complement read i in place where flag[i] != 0 (comp = 256-entry table) */
void synth_revcomp(uint8_t *blob, const int64_t *offs, const uint8_t *flag, int64_t n, const uint8_t *comp)
{
#pragma omp parallel for schedule(static, 4096)
  for (int64_t i = 0; i < n; i++) {
    if (!flag[i]) continue;
    uint8_t *a = blob + offs[i], *b = blob + offs[i + 1] - 1;
    while (a < b) { const uint8_t x = comp[*a], y = comp[*b]; *a++ = y; *b-- = x; }
    if (a == b) *a = comp[*a];
  }
}
