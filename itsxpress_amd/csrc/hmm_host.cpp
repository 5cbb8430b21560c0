// hmm_host.cpp -- HMMER3/f text -> configured search profiles (host side of the engine).
//
// Replaces what hmmsearch does with `hmmfile` before scoring (reference call site
// itsxpress/SeqSample.py:191-209; the file is written by create_runtime_hmm,
// itsxpress/main.py:176-231): read each model, configure it for multihit local
// alignment against a uniform DNA background (p7_ProfileConfig), and convert it to the
// two scoring systems the filters use: unsigned-byte MSV costs (scale 3/ln2, base 190)
// and striped odds-ratio floats for Forward/Backward (4 lanes x Q, node k = z*Q+q+1).
// This happens once per model and uses the host libm, exactly as hmmsearch does.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include "engine.h"

namespace itsx {

static const double kLn2 = 0.69314718055994529;
static const double kLn2R = 1.44269504088896341;

// IUPAC degeneracy masks over {A,C,G,T} for digital codes 0..17
static const unsigned char kDegen[KP] = {1, 2, 4, 8, 0, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13, 15, 0, 0};

// 4-lane Cephes-style expf as used for log-odds -> odds conversion; float ops only, no FMA
static float lane_expf(float x)
{
  const float c[6] = {1.9875691500E-4f, 1.3981999507E-3f, 8.3334519073E-3f,
                      4.1665795894E-2f, 1.6666665459E-1f, 5.0000001201E-1f};
  if (x > 88.72283905206835f) return INFINITY;
  if (x <= -103.27892990343185f) return 0.0f;
  volatile float fx = x * (float)kLn2R;
  fx = fx + 0.5f;
  float fl = (float)(int)fx;
  if (fl > fx) fl = fl - 1.0f;
  const int k = (int)fl;
  volatile float a = fl * 0.693359375f;
  volatile float b = fl * -2.12194440e-4f;
  volatile float r = x - a;
  r = r - b;
  volatile float rr = r * r;
  volatile float y = c[0];
  for (int i = 1; i < 6; i++) { y = y * r; y = y + c[i]; }
  y = y * rr;
  y = y + r;
  y = y + 1.0f;
  uint32_t bits = (uint32_t)(k + 127) << 23;
  float p2; memcpy(&p2, &bits, 4);
  y = y * p2;
  return y;
}

static float tok_prob(const std::string &t) { return t[0] == '*' ? 0.0f : expf((float)(-1.0 * atof(t.c_str()))); }

static int cost_byte(float scale, float sc)             // unbiased
{
  float v = -1.0f * roundf(scale * sc);
  return v > 255.0f ? 255 : (int)(uint8_t)(int)v;
}
static int cost_byte_biased(float scale, int bias, float sc)
{
  float v = -1.0f * roundf(scale * sc);
  if (v > (float)(255 - bias)) return 255;
  return (int)(uint8_t)((int)v + bias);
}
uint8_t host_tjb_b(int L) { return (uint8_t)cost_byte((float)(3.0 / kLn2), logf(3.0f / (float)(L + 3))); }

static void configure(HostProfile &p)
{
  const int M = p.M;
  const float bgf = 0.25f;
  enum { MM, MI, MD, IM, II, DM, DD, BM };
  p.tsc.assign((size_t)(M + 1) * 8, -INFINITY);
  p.msc.assign((size_t)(M + 1) * KP, -INFINITY);
  // local entry: occupancy-weighted
  std::vector<float> occ(M + 1, 0.0f);
  occ[1] = p.t[MI] + p.t[MM];
  for (int k = 2; k <= M; k++) {
    const float *tp = &p.t[(size_t)(k - 1) * 7];
    occ[k] = (float)((double)(occ[k - 1] * (tp[MM] + tp[MI])) + (1.0 - (double)occ[k - 1]) * (double)tp[DM]);
  }
  float Z = 0.0f;
  for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M - k + 1);
  for (int k = 1; k <= M; k++) p.tsc[(size_t)(k - 1) * 8 + BM] = (float)log((double)(occ[k] / Z));
  for (int k = 1; k < M; k++)
    for (int x = 0; x < 7; x++) p.tsc[(size_t)k * 8 + x] = (float)log((double)p.t[(size_t)k * 7 + x]);
  for (int k = 1; k <= M; k++) {
    float sc[KP];
    for (int x = 0; x < KP; x++) sc[x] = -INFINITY;
    for (int x = 0; x < 4; x++) sc[x] = (float)log((double)p.mat[(size_t)k * 4 + x] / (double)bgf);
    for (int x = 5; x <= 15; x++) {
      float num = 0.0f, den = 0.0f;
      for (int y = 0; y < 4; y++) if (kDegen[x] >> y & 1) { num += sc[y] * bgf; den += bgf; }
      sc[x] = num / den;
    }
    for (int x = 0; x < KP; x++) p.msc[(size_t)k * KP + x] = sc[x];
  }
  // MSV byte costs
  float mx = 0.0f;
  for (int x = 0; x < 4; x++) for (int k = 1; k <= M; k++) mx = std::max(mx, p.msc[(size_t)k * KP + x]);
  p.scale_b = (float)(3.0 / kLn2);
  p.base_b = 190;
  p.bias_b = cost_byte(p.scale_b, (float)(-1.0 * (double)mx));
  p.rbv.assign((size_t)KP * (M + 1), 255);
  for (int x = 0; x < KP; x++)
    for (int k = 1; k <= M; k++) p.rbv[(size_t)x * (M + 1) + k] = (uint8_t)cost_byte_biased(p.scale_b, p.bias_b, p.msc[(size_t)k * KP + x]);
  p.tbm_b = cost_byte(p.scale_b, logf(2.0f / ((float)M * (float)(M + 1))));
  p.tec_b = cost_byte(p.scale_b, logf(0.5f));
  // striped odds ratios
  const int Q = std::max(2, (M + 3) / 4);
  p.Q = Q;
  p.rfv.assign((size_t)KP * Q * 4, 0.0f);
  p.tfv.assign((size_t)8 * Q * 4, 0.0f);
  for (int x = 0; x < KP; x++)
    for (int q = 0; q < Q; q++)
      for (int z = 0; z < 4; z++) {
        const int k = q + 1 + z * Q;
        p.rfv[((size_t)x * Q + q) * 4 + z] = lane_expf(k <= M ? p.msc[(size_t)k * KP + x] : -INFINITY);
      }
  const int src[7] = {BM, MM, IM, DM, MD, MI, II};
  const int rot[7] = {-1, -1, -1, -1, 0, 0, 0};
  for (int q = 0; q < Q; q++)
    for (int t = 0; t < 7; t++)
      for (int z = 0; z < 4; z++) {
        const int kb = q + 1 + rot[t] + z * Q;
        p.tfv[((size_t)q * 7 + t) * 4 + z] = lane_expf(kb < M ? p.tsc[(size_t)kb * 8 + src[t]] : -INFINITY);
      }
  for (int q = 0; q < Q; q++)
    for (int z = 0; z < 4; z++) {
      const int k = q + 1 + z * Q;
      p.tfv[((size_t)7 * Q + q) * 4 + z] = lane_expf(k < M ? p.tsc[(size_t)k * 8 + DD] : -INFINITY);
    }
  // ---- Viterbi filter word model (p7_oprofile.c: vf_conversion), unstriped: scale 500/ln2; no transition above 0, no II above -1
  {
    const float scale_w = (float)(500.0 / 0.69314718055994529);
    auto wordify = [&](float sc) { const float w = roundf(scale_w * sc); return (int16_t)((w >= 32767.0f) ? 32767 : (w <= -32768.0f) ? -32768 : (int)w); };
    p.rww.assign((size_t)KP * (M + 1), (int16_t)-32768);
    for (int x = 0; x < KP; x++) for (int kk = 1; kk <= M; kk++) p.rww[(size_t)x * (M + 1) + kk] = wordify(p.msc[(size_t)kk * KP + x]);
    p.tww.assign((size_t)8 * (M + 1), (int16_t)-32768);
    const int into[4] = {BM, MM, IM, DM}, outof[4] = {MD, MI, II, DD};
    for (int kk = 1; kk <= M; kk++) {
      for (int t = 0; t < 4; t++) { const int16_t v = wordify(p.tsc[(size_t)(kk - 1) * 8 + into[t]]); p.tww[(size_t)t * (M + 1) + kk] = v > 0 ? (int16_t)0 : v; }
      for (int t = 0; t < 4; t++) {
        int16_t v = -32768;
        if (kk < M) v = wordify(p.tsc[(size_t)kk * 8 + outof[t]]);
        const int16_t maxval = outof[t] == II ? -1 : 0;
        p.tww[(size_t)(4 + t) * (M + 1) + kk] = outof[t] == DD ? v : (v > maxval ? maxval : v);
      }
    }
  }
  // bias-composition filter: state 0 = background (its transitions follow the target length),
  // state 1 = the model's composition
  const float L1 = (float)((double)(float)M / 8.0);
  p.ft10 = 1.0f / (L1 + 1.0f);
  p.ft11 = L1 / (L1 + 1.0f);
  p.fpi0 = (float)0.999;
  p.fpi1 = (float)0.001;
  float e[2][4];
  for (int x = 0; x < 4; x++) { e[0][x] = bgf; e[1][x] = p.compo[x]; }
  for (int x = 0; x < KP; x++) p.feo[x][0] = p.feo[x][1] = 1.0f;
  for (int x = 0; x < 4; x++) for (int s = 0; s < 2; s++) p.feo[x][s] = e[s][x] / bgf;
  for (int x = 5; x <= 15; x++)
    for (int s = 0; s < 2; s++) {
      float num = 0.0f, den = 0.0f;
      for (int y = 0; y < 4; y++) if (kDegen[x] >> y & 1) { num += e[s][y]; den += bgf; }
      p.feo[x][s] = den > 0.0f ? num / den : 0.0f;
    }
}

static std::vector<std::string> words_of(const std::string &line)
{
  std::vector<std::string> w;
  std::istringstream is(line);
  std::string t;
  while (is >> t) w.push_back(t);
  return w;
}

std::string parse_hmm_text(const char *text, int64_t len, std::vector<HostProfile> &out)
{
  std::vector<std::string> lines;
  {
    const char *s = text, *end = text + len;
    while (s < end) {
      const char *e = (const char *)memchr(s, '\n', (size_t)(end - s));
      if (!e) e = end;
      std::string ln(s, e);
      if (!ln.empty() && ln.back() == '\r') ln.pop_back();
      lines.push_back(ln);
      s = e + 1;
    }
  }
  size_t i = 0;
  auto fail = [&](const std::string &m) { return m + " (line " + std::to_string(i + 1) + ")"; };
  while (i < lines.size()) {
    if (lines[i].find_first_not_of(" \t") == std::string::npos) { i++; continue; }
    if (lines[i].compare(0, 6, "HMMER3") != 0) return fail("expected a HMMER3 header");
    i++;
    HostProfile p;
    int stats = 0;
    for (; i < lines.size(); i++) {
      const std::string &ln = lines[i];
      if (ln == "HMM" || ln.compare(0, 4, "HMM ") == 0) break;
      if (ln.compare(0, 4, "NAME") == 0) { auto w = words_of(ln.substr(4)); if (!w.empty()) p.name = w[0]; }
      else if (ln.compare(0, 4, "LENG") == 0) p.M = atoi(ln.c_str() + 4);
      else if (ln.compare(0, 4, "ALPH") == 0) { auto w = words_of(ln.substr(4)); if (w.empty() || (w[0] != "DNA" && w[0] != "dna")) return fail("only ALPH DNA models are supported"); }
      else if (ln.compare(0, 5, "STATS") == 0) {
        auto w = words_of(ln);
        if (w.size() >= 5) {
          int b = w[2] == "MSV" ? 0 : w[2] == "VITERBI" ? 2 : w[2] == "FORWARD" ? 4 : -1;
          if (b >= 0) { p.evparam[b] = (float)atof(w[3].c_str()); p.evparam[b + 1] = (float)atof(w[4].c_str()); stats |= 1 << (b / 2); }
        }
      }
    }
    if (i >= lines.size()) return fail("truncated model (no HMM line)");
    if (p.M <= 0) return fail("model without LENG");
    if (stats != 7) return fail("model is not calibrated (STATS LOCAL lines missing)");
    i += 2;
    if (i >= lines.size()) return fail("truncated model");
    {
      auto w = words_of(lines[i]);
      if (w.size() < 5 || w[0] != "COMPO") return fail("model has no COMPO line");
      for (int x = 0; x < 4; x++) p.compo[x] = tok_prob(w[1 + x]);
      i++;
    }
    i++;  // node-0 insert emissions: inserts score 0 in search profiles
    const int M = p.M;
    p.t.assign((size_t)(M + 1) * 7, 0.0f);
    p.mat.assign((size_t)(M + 1) * 4, 0.0f);
    if (i >= lines.size()) return fail("truncated model");
    {
      auto w = words_of(lines[i]);
      if (w.size() < 7) return fail("bad begin-transition line");
      for (int x = 0; x < 7; x++) p.t[x] = tok_prob(w[x]);
      i++;
    }
    for (int k = 1; k <= M; k++) {
      if (i + 2 >= lines.size()) return fail("truncated model");
      auto w = words_of(lines[i]);
      if (w.size() < 5 || atoi(w[0].c_str()) != k) return fail("bad match line");
      for (int x = 0; x < 4; x++) p.mat[(size_t)k * 4 + x] = tok_prob(w[1 + x]);
      i += 2;
      w = words_of(lines[i]);
      if (w.size() < 7) return fail("bad transition line");
      for (int x = 0; x < 7; x++) p.t[(size_t)k * 7 + x] = tok_prob(w[x]);
      i++;
    }
    if (i >= lines.size() || lines[i].compare(0, 2, "//") != 0) return fail("expected // after the last node");
    i++;
    configure(p);
    out.push_back(std::move(p));
  }
  return "";
}

}  // namespace itsx
