// engine.h -- internal declarations of the gfx950 ITS-trimming engine (not part of the ABI).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/itsx_hip.h"
#include "switches.h"

namespace itsx {

constexpr int KP = 18;          // A C G T - R Y M K S W H B V D N * ~  (HMMER's DNA digital alphabet)
constexpr int NCODE = 16;       // codes that can occur in a read: 0..15 (4 = gap never does)
constexpr int QMAX = 12;        // float striping segment length the device kernels are unrolled for
constexpr int MMAX = 46;        // longest model the MSV kernel holds in registers (23 packed pairs)
constexpr int MSV_REGS = 23;
constexpr int MSV_TW = 24;            // dwords per residue code in a profile's MSV emission table (MSV_REGS used)
constexpr int MAXDOM = 8;       // regions kept per (rep, profile)
constexpr int BOUND_PAIRS = 23; // k_lazy.hip: pairs of adjacent nodes the bound kernel keeps (2 x 23 = MMAX)
constexpr int BOUND_RTAB = BOUND_PAIRS * 12 + 4;   // k_bwd_bound's table: per pair of nodes mm im dm ii bm dd, two floats each (round 6)
constexpr int BOUND_TAB = (BOUND_PAIRS + 1) * 16 + 2 * BOUND_PAIRS + 2;   // floats per profile of its table: 24 records of 16, then the match cells' scale by node (FOLD)

// ---- host-side model (profile configuration happens once per model, on the host, with libm) ----
struct HostProfile {
  std::string name;
  int M = 0, Q = 0;
  float compo[4];
  float evparam[6];
  std::vector<float> t, mat, tsc, msc;
  float scale_b; int base_b, bias_b, tbm_b, tec_b;
  std::vector<uint8_t> rbv;   // [KP][M+1]
  std::vector<float> rfv;     // [KP][Q][4]
  std::vector<float> tfv;     // [8Q][4]
  float ft10, ft11, fpi0, fpi1;
  float feo[KP][2];
  std::vector<int16_t> rww;   // Viterbi filter words [KP][M+1]
  std::vector<int16_t> tww;   // [8][M+1]: BM MM IM DM into node k | MD MI II DD out of node k
};
// returns "" on success, else an error message; appends to out
std::string parse_hmm_text(const char *text, int64_t len, std::vector<HostProfile> &out);
uint8_t host_tjb_b(int L);

// ---- device-side model block: one per profile, read with scalar loads (wave-uniform) ----
struct alignas(16) DevProfile {
  float tf[QMAX * 8 * 4];       // [q][t][z], t: BM MM IM DM MD MI II DD
  float tb[QMAX * 6 * 4];       // Backward's main loop, per q: II(q) MI(q) BM(q) MM(q+1) IM(q+1) DM(q+1), the
                                // wrap-around at q = Q-1 (left shift of group 0) already applied
  float rf[NCODE * QMAX * 4];   // [code][q][z] match emission odds ratios
  float feo[NCODE * 2];         // bias-filter emission odds [code][state]
  float ft10, ft11, fpi0, fpi1; // bias-filter HMM
  float ev[6];                  // MSV mu,lambda  VIT mu,lambda  FWD tau,lambda
  int   M, Q;
  int   pad[4];                 // (tfn on a 16-byte boundary)
  float tfn[(QMAX * 4 + 1) * 8]; // tf by node: [k][t], k = 0 (zeros) .. 4 Q (k_ensemble.hip: a traceback step loads one node's transitions)
};

static_assert(offsetof(DevProfile, tb) == sizeof(float) * QMAX * 8 * 4, "k_float.hip reaches tb as tf + 384 floats");
static_assert(offsetof(DevProfile, tfn) % 16 == 0 && sizeof(DevProfile) % 16 == 0, "tfn is read with 16-byte loads");

constexpr int VIT_TAB = 8 * (MMAX + 1) + NCODE * (MMAX + 1);   // int16 words of one profile's Viterbi-filter tables on the device
struct LenTables {              // per target length L, built on the host with libm
  float   nullsc;               // L*log(p1) + log(1-p1)
  float   bias_a;               // (float)L * logf(p1)
  float   bias_b;               // logf(1-p1)
  float   p1;
  double  lognn3;               // log((float)L/(float)(L+3))
  int     tjb;                  // MSV J->B / N->B cost byte
  int     vmove;                // Viterbi filter: wordify(logf(3 / (L + 3)))
  float   lazy_c;               // lazy domain stage: C(L) + margin, bits (engine.hip: lazy_bound_c)
  int     pad;
};

// a surviving (representative, profile) comparison
struct PairRec {
  int32_t  useq;                // index in the length-sorted unique list
  int32_t  prof;
  int32_t  xj;                  // MSV xJ byte (255 = overflow)
  int32_t  L;
};
struct PairOut {                // filled by the filter/DP kernels
  float filtersc, fwdsc, bcksc, nullsc, msv_sc;
  int32_t pass_bias, pass_fwd, nregions, ndom, flags;
  int32_t nregions_raw;         // regions the scan kept before clustered ones were replaced by their envelopes (ndom counts those)
};
struct RegionRec {              // one envelope to re-score
  int32_t pair;                 // index into the pair list
  int32_t ienv, jenv;
  int32_t multi;
};
struct RegionOut {
  float envsc;
  float domcorrection;          // sum of n2log over the envelope's residues
  float n2log[NCODE];           // log null2 odds per residue code
  int32_t ok;                   // -1 unusable, 0 sweeps pending, 1 complete
  int32_t own;                  // Backward had to switch to its own scale factors
  float bN0;                    // Backward's N at row 0 (total probability)
};

#define HIPCHK(expr)                                                                            \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e__));                            \
      return ITSX_E_DEVICE;                                                                     \
    }                                                                                           \
  } while (0)

}  // namespace itsx
