// writers_host.cpp -- the file-compatible outputs written from ARRAYS instead of from one context's state: what a driver that
// spreads one sample over several GPUs (itsxpress_amd/multi.py: ITSXPRESS_GPUS=N) assembles from its workers and must hand to the
// reference's parsers as ONE uc.txt / rep.fa / domtbl.txt, byte for byte the files one GPU writes.
//
// Replaces, like the context's own writers (engine.hip): the --uc / --fastaout files of `vsearch --fastx_uniques`
// (itsxpress/SeqSample.py:104-116, read back by Dedup.parse :542-562) and hmmsearch's --domtblout (SeqSample.py:190-209, read back
// by ItsPosition.parse :431-461).  Host-only text formatting, context-free, no arithmetic of the path.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/itsx_hip.h"
#include "detmath.h"

namespace {
std::string g_wr_error;
int fail(int code, const std::string &m) { g_wr_error = m; return code; }

struct Names {
  const char *blob; const int64_t *offs;
  std::string get(int64_t r) const
  {
    if (blob && offs) return std::string(blob + offs[r], (size_t)(offs[r + 1] - offs[r]));
    char b[32]; snprintf(b, sizeof(b), "r%09lld", (long long)r); return b;
  }
};
}  // namespace

extern "C" {

const char *itsx_writers_last_error(void) { return g_wr_error.c_str(); }

// uc.txt and rep.fa of exact dereplication from per-read arrays: rep_of[i] = read index of the cluster's seed (its first
// occurrence; -1 = the read was dropped), strand[i] = +1 / -1 relative to the seed, len[i]; names (NULL: r%09d); the seeds'
// sequences concatenated in INPUT order of the seeds (seed_offs[n_seeds + 1]).  Either path may be NULL.
// vsearch's order (confirmed on the reference's fixture): clusters by abundance descending, ties by label
// (strcmp); each S row followed by its H rows in input order; all C rows last; rep.fa in S order, 80 columns.
int itsx_write_derep_arrays(const char *uc_path, const char *rep_path, int64_t n, const int64_t *rep_of, const int8_t *strand,
                            const int32_t *len, const char *names, const int64_t *name_offsets, const char *seed_bases,
                            const int64_t *seed_offs, int64_t n_seeds)
{
  if (n < 0 || (n > 0 && (!rep_of || !strand || !len))) return fail(ITSX_E_ARG, "itsx_write_derep_arrays: missing arrays");
  const Names nm{names, name_offsets};
  std::vector<int32_t> uniq_of((size_t)n, -1);
  std::vector<int64_t> seed_read;
  for (int64_t r = 0; r < n; r++) if (rep_of[r] == r) { uniq_of[(size_t)r] = (int32_t)seed_read.size(); seed_read.push_back(r); }
  const int64_t U = (int64_t)seed_read.size();
  if (rep_path && U != n_seeds) return fail(ITSX_E_ARG, "itsx_write_derep_arrays: " + std::to_string(U) + " seeds in rep_of, " + std::to_string(n_seeds) + " sequences given");
  std::vector<int32_t> abund((size_t)U, 0);
  std::vector<std::vector<int64_t>> members((size_t)U);
  for (int64_t r = 0; r < n; r++) {
    const int64_t s = rep_of[r];
    if (s < 0) continue;
    if (s >= n || rep_of[s] != s) return fail(ITSX_E_ARG, "itsx_write_derep_arrays: rep_of[" + std::to_string(r) + "] is not a seed");
    const int32_t u = uniq_of[(size_t)s];
    abund[(size_t)u]++;
    if (s != r) members[(size_t)u].push_back(r);
  }
  std::vector<int32_t> ord((size_t)U);
  for (int64_t u = 0; u < U; u++) ord[(size_t)u] = (int32_t)u;
  std::vector<std::string> lab((size_t)U);
  for (int64_t u = 0; u < U; u++) lab[(size_t)u] = nm.get(seed_read[(size_t)u]);
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
    if (abund[(size_t)a] != abund[(size_t)b]) return abund[(size_t)a] > abund[(size_t)b];
    return strcmp(lab[(size_t)a].c_str(), lab[(size_t)b].c_str()) < 0;
  });
  if (uc_path) {
    FILE *f = fopen(uc_path, "w");
    if (!f) return fail(ITSX_E_IO, std::string("cannot write ") + uc_path);
    for (size_t c = 0; c < ord.size(); c++) {
      const int32_t u = ord[c]; const int64_t s = seed_read[(size_t)u];
      const std::string &sl = lab[(size_t)u];
      fprintf(f, "S\t%zu\t%d\t*\t*\t*\t*\t*\t%s\t*\n", c, len[s], sl.c_str());
      for (int64_t r : members[(size_t)u])
        fprintf(f, "H\t%zu\t%d\t100.0\t%c\t0\t0\t*\t%s\t%s\n", c, len[r], strand[r] < 0 ? '-' : '+', nm.get(r).c_str(), sl.c_str());
    }
    for (size_t c = 0; c < ord.size(); c++) fprintf(f, "C\t%zu\t%d\t*\t*\t*\t*\t*\t%s\t*\n", c, abund[(size_t)ord[c]], lab[(size_t)ord[c]].c_str());
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return fail(ITSX_E_IO, std::string("short write to ") + uc_path);
  }
  if (rep_path) {
    if (U > 0 && (!seed_bases || !seed_offs)) return fail(ITSX_E_ARG, "itsx_write_derep_arrays: rep.fa needs the seeds' sequences");
    FILE *f = fopen(rep_path, "w");
    if (!f) return fail(ITSX_E_IO, std::string("cannot write ") + rep_path);
    for (int32_t u : ord) {
      fprintf(f, ">%s\n", lab[(size_t)u].c_str());
      const char *b = seed_bases + seed_offs[u]; const int64_t L = seed_offs[u + 1] - seed_offs[u];
      for (int64_t i = 0; i < L; i += 80) { fwrite(b + i, 1, (size_t)std::min<int64_t>(80, L - i), f); fputc('\n', f); }
    }
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) return fail(ITSX_E_IO, std::string("short write to ") + rep_path);
  }
  return ITSX_OK;
}

// domtbl.txt from domain rows gathered from several contexts: rows[n_rows] with rep = index into target_names (the GLOBAL unique
// list, input order of the seeds), any order (sorted here: profile, target, domain -- the context writer's order); dom_reported
// as itsx_search_finalize left it.  Z = targets searched (hmmsearch's Z: the global number of uniques), domz[n_profiles] = the
// data set's reported targets per profile; per profile its NAME, length M and the Forward tail (tau, lambda) for the E-value columns.
int itsx_write_domtbl_arrays(const char *path, const itsx_domain *rows, int64_t n_rows, int64_t Z, const int64_t *domz, int32_t n_profiles,
                             const char *prof_names, const int64_t *prof_name_offsets, const int32_t *prof_M, const float *prof_tau,
                             const float *prof_lambda, const char *target_names, const int64_t *target_name_offsets)
{
  if (!path || n_rows < 0 || (n_rows > 0 && !rows) || n_profiles < 0 || (n_profiles > 0 && (!domz || !prof_names || !prof_name_offsets || !prof_M || !prof_tau || !prof_lambda)))
    return fail(ITSX_E_ARG, "itsx_write_domtbl_arrays: missing arrays");
  const Names tn{target_names, target_name_offsets};
  std::vector<itsx_domain> D;
  D.reserve((size_t)n_rows);
  for (int64_t i = 0; i < n_rows; i++) if (rows[i].dom_idx >= 0 && rows[i].prof >= 0 && rows[i].prof < n_profiles) D.push_back(rows[i]);
  std::stable_sort(D.begin(), D.end(), [](const itsx_domain &a, const itsx_domain &b) {
    if (a.prof != b.prof) return a.prof < b.prof;
    if (a.rep != b.rep) return a.rep < b.rep;
    return a.dom_idx < b.dom_idx;
  });
  FILE *f = fopen(path, "w");
  if (!f) return fail(ITSX_E_IO, std::string("cannot write ") + path);
  fprintf(f, "#                                                                            --- full sequence --- -------------- this domain -------------   hmm coord   ali coord   env coord\n");
  fprintf(f, "# target name        accession   tlen query name           accession   qlen   E-value  score  bias   #  of  c-Evalue  i-Evalue  score  bias  from    to  from    to  from    to  acc description of target\n");
  size_t i = 0;
  while (i < D.size()) {
    size_t j = i; int nrep = 0;
    while (j < D.size() && D[j].prof == D[i].prof && D[j].rep == D[i].rep) { nrep += D[j].dom_reported == 1; j++; }
    const std::string tname = nrep ? tn.get(D[i].rep) : std::string();
    const int p = D[i].prof;
    const std::string pname(prof_names + prof_name_offsets[p], (size_t)(prof_name_offsets[p + 1] - prof_name_offsets[p]));
    int k = 0;
    for (size_t d = i; d < j; d++) {
      if (D[d].dom_reported != 1) continue;
      k++;
      const double dz = (double)domz[p];
      const double seqE = (double)Z * itsx::det_exp(itsx::exp_logsurv((double)D[d].seq_score, (double)prof_tau[p], (double)prof_lambda[p]));
      const double P = itsx::det_exp(D[d].lnP);
      fprintf(f, "%-20s %-10s %5d %-20s %-10s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5d %5d %5d %5d %4.2f %s\n",
              tname.c_str(), "-", D[d].tlen, pname.c_str(), "-", prof_M[p], seqE, D[d].seq_score, D[d].seq_bias,
              k, nrep, P * dz, P * (double)Z, D[d].bitscore, D[d].dombias / 0.69314718055994529, 1, prof_M[p], D[d].ienv, D[d].jenv, D[d].ienv, D[d].jenv, 0.0, "-");
    }
    i = j;
  }
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad) return fail(ITSX_E_IO, std::string("short write to ") + path);
  return ITSX_OK;
}

}  // extern "C"
