// Ecosystem source/sink terms of BGC_SourceSink (and, for obgc_fused_step, the cell terms of DMS_SourceSink and
// MACROS_SourceSink), in two forms that share their device functions statement for statement and give the same bits:
//
//   native (column-fastest) layout -- ONE launch:
//   bgc_column_kernel  64 columns x 4 levels / workgroup, marching down the columns; lane = column, wavefront = level.
//                      Light and the particle recurrence are carried through LDS (see the comment above the kernel).
//
//   reference (level-fastest) layout, or OBGC_COLUMN_KERNEL=0 -- four kernels with hand-off planes in global memory:
//   bgc_att_kernel     1 lane / cell     attenuation exp(-KPARdz) of every level
//   bgc_parscan_kernel 1 lane / column   PAR entering every level: one multiply per level
//   bgc_cell_kernel    1 lane / cell     every cell-local term: quotas, uptake, photoadaptation, grazing, mortality, DOM,
//                                        the PRODUCTION terms of the sinking particles, 19 final tendencies and the local
//                                        part of 10 more
//   bgc_sweep_kernel   1 lane / column   the Armstrong ballast recurrence top to bottom, iron scavenging, sea-floor burial,
//                                        and the 11 tendencies and the conservation integrals that depend on it
//   (fused_att_kernel / fused_parscan_kernel / fused_cell_kernel: the same with the DMS / MACROS terms)
//
//   bgc_zint_kernel    1 lane / column   column integrals of the production profiles, O2 minimum, Jint_* (both forms)
//
// Behavioural contract = BGC_mod.F90:733-931 (clamps, zero-mask, quotas, light), 1041-1529 (kinetics), 1531-1536 ->
// 2116-2699 (particles), 1545-1968 (tendencies, diagnostics, column integrals, O2 minimum), 2006-2109 (particle
// initialisation).  The carbonate block 935-1032 is obgc_co2.h: a kernel of its own (obgc_carbonate.hip) or solved by the
// lanes of the kernels here (FUSE).
//
// The reference walks each column sequentially, but only PAR, the particle fluxes, the QA-dust deficit and the column
// integrals are really carried from level to level -- a few dozen flops per cell; the ~2 000 flops of kinetics per cell
// are level-independent once PAR is known.  With one lane per column and level after level in registers the 100k-column
// configuration offers 1 563 wavefronts for 1 024 SIMDs (measured: 5.2 ms, latency bound at one 512-register wave per
// SIMD); both forms here give every cell a lane of its own for the kinetics.
//
// Arithmetic order is the reference's everywhere except three tendency sums that the hand-off re-associates (Fe, SiO3,
// PO4: the per-autotroph terms are folded before the particle terms are added) and P_iron production (Fe scavenging added
// last); the change is a few ulp of the largest term, far inside the parity bar.
#include <cstdio>
#include <cstring>
#include <mutex>

#include "obgc_co2.h"
#include "obgc_dm.h"

namespace obgc {

namespace {

constexpr int kBlockCol = 64;
constexpr int kBlockCell = 128;

// ---- compile-time constants of BGC_parms.F90 (values, not names, are the contract) ----
constexpr double DPS = 1.0 / 86400.0;                   // 37-40
constexpr double YPS = 1.0 / (365.0 * 86400.0);
constexpr double SPD = 86400.0;
constexpr double RED_FE_C = 3.0e-6;                     // parm_Red_Fe_C, 338
constexpr double RED_D_C_O2 = 117.0 / 170.0;            // 335
constexpr double REMIN_D_C_O2 = 117.0 / 138.0;          // 336
constexpr double RED_D_C_O2_DIAZ = 117.0 / 150.0;       // 339
constexpr double FE_SCAV_THRES1 = 0.8e-3, DUST_FESCAV_SCALE = 1.0e9, FE_MAX_SCALE2 = 1200.0;  // 371-374
constexpr double DUST_TO_FE = 0.035 / 55.847 * 1.0e9;   // 385-386
constexpr double CACO3_POC_MIN = 0.4, SPC_POC_FAC = 0.11, F_GRAZE_SP_POC_LIM = 0.3, F_PHOTOSP_CACO3 = 0.4,
                 F_GRAZE_CACO3_REMIN = 0.33, F_GRAZE_SI_REMIN = 0.35;   // 394-405
constexpr double R_NFIX_PHOTO = 1.25;                   // 411-412
constexpr double QN = 0.137, QP_ZOO_POM = 0.00855, QFE_ZOO = 3.0e-6, GQSI_0 = 0.137, GQSI_MAX = 0.685,
                 GQSI_MIN = 0.0457, QCACO3_MAX = 0.4, DENITRIF_C_N = 117.0 / 136.0;   // 420-429
constexpr double THRES_Z1 = 100.0e2, THRES_Z2 = 150.0e2, LOSS_THRES_ZOO = 0.005, CACO3_TEMP_THRES1 = 6.0,
                 CACO3_TEMP_THRES2 = -2.0, CACO3_SP_THRES = 4.0;        // 435-441
constexpr int TFNC_Q10 = 1, TFNC_QUASI_MMRT = 2;        // 447-449
constexpr double F_QSW_PAR = 0.45, TREF = 30.0;   // 454-463 (Q_10 = 1.5 enters through LN_Q_10 below)
// Q_10**x and 0.99**x are evaluated as exp(x ln b), x**1.5 as x sqrt(x): the device pow() costs ~200 fp64 instructions
// (it cannot know its base is a constant), exp ~40, sqrt ~20.  Same function, <= 2 ulp from pow.
constexpr double LN_Q_10 = 0.4054651081081644, LN_0_99 = -0.01005033585350145;
constexpr double DOC_REMINR = (1.0 / 250.0) * DPS, DON_REMINR = (1.0 / 160.0) * DPS, DOFE_REMINR = (1.0 / 160.0) * DPS,
                 DOP_REMINR = (1.0 / 160.0) * DPS, DONR_REMINR = (1.0 / (365.0 * 2.5)) * DPS,
                 DOPR_REMINR = (1.0 / (365.0 * 2.5)) * DPS, DONREFRACT = 0.08, DOPREFRACT = 0.03;   // 469-477
// 479-482: default-real literals assigned to r8 parameters keep their float32 value
constexpr double EPSC = (double)1.00e-8f, EPSTINV = (double)3.17e-8f;
constexpr double CKS = 9.0, CKSI = 5.0;                 // 484-486
// particle classes (BGC_mod.F90:2046-2069): mass, hard fraction, QA mass ratio to POC
constexpr double POC_MASS = 12.01;
constexpr double CACO3_GAMMA = 0.30, CACO3_MASS = 100.09, CACO3_RHO = 0.05 * CACO3_MASS / POC_MASS;
constexpr double SIO2_GAMMA = 0.030, SIO2_MASS = 60.08, SIO2_RHO = 0.05 * SIO2_MASS / POC_MASS;
constexpr double DUST_DISS = 20000.0, DUST_GAMMA = 0.97, DUST_MASS = 1.0e9, DUST_RHO = 0.05 * DUST_MASS / POC_MASS;

__device__ __forceinline__ double sum4(const double (&v)[4]) { return ((v[0] + v[1]) + v[2]) + v[3]; }

// Garcia & Gordon (1992) O2 saturation in mmol/m^3 (BGC_mod.F90:3012-3083)
__device__ __forceinline__ double o2sat(double SST, double SSS, double T0K) {
  const double TS = log(((T0K + 25.0) - SST) / (T0K + SST));
  const double v = exp(2.00907 + TS * (3.22014 + TS * (4.05010 + TS * (4.94457 + TS * (-2.56847E-1 + TS * 3.88767)))) +
                       SSS * ((-6.24523E-3 + TS * (-7.37614E-3 + TS * (-1.03410E-2 + TS * -8.17083E-3))) + SSS * -4.88682E-7));
  return v / 0.0223916;
}

// Pins a wave-uniform parameter in an SGPR pair.  Without it the compiler turns  lane_cond ? param_a : param_b
// into a per-lane VECTOR load from a selected address; inside the autotroph loop that load queues behind every
// diagnostic store already issued (loads and stores retire in order) and stalls the wave for a memory round trip.
__device__ __forceinline__ double uniform(double v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Masked, clamped autotroph pools of one cell (BGC_mod.F90:764-785, 826-844)
struct Pools { double Chl[4], C[4], Fe[4], Si[4], Ca[4]; };

// chl0: the clamped chlorophyll before the zero-mask (what DMS_SourceSink sums, DMS_mod.F90:478-481, 539)
__device__ __forceinline__ void load_pools(const BgcKernelArgs& A, const obgc_autotroph* at, long long ix, Pools& p, double (&chl0)[4]) {
  const long long pl = A.g.plane;
  // all loads first and unconditional (an autotroph without Si reads its C plane again and discards it): one
  // wait for the whole batch instead of one memory round trip per uniform branch
  double si[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    p.Chl[a] = A.tracers[(long long)(at[a].Chl_ind - 1) * pl + ix];
    p.C[a] = A.tracers[(long long)(at[a].C_ind - 1) * pl + ix];
    p.Fe[a] = A.tracers[(long long)(at[a].Fe_ind - 1) * pl + ix];
    si[a] = A.tracers[(long long)((at[a].Si_ind > 0 ? at[a].Si_ind : at[a].C_ind) - 1) * pl + ix];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    p.Chl[a] = dmax(0.0, p.Chl[a]);
    chl0[a] = p.Chl[a];
    p.C[a] = dmax(0.0, p.C[a]);
    p.Fe[a] = dmax(0.0, p.Fe[a]);
    p.Si[a] = (at[a].Si_ind > 0) ? dmax(0.0, si[a]) : 0.0;
    p.Ca[a] = 0.0;
    bool empty = p.Chl[a] == 0.0 || p.C[a] == 0.0 || p.Fe[a] == 0.0;
    if (at[a].Si_ind > 0) empty = empty || p.Si[a] == 0.0;
    if (empty) { p.Chl[a] = 0.0; p.C[a] = 0.0; p.Fe[a] = 0.0; p.Si[a] = 0.0; }
  }
}

// Morel & Maritorena attenuation across one level (BGC_mod.F90:914-921)
__device__ __forceinline__ double kpar_dz(const double (&Chl)[4], double dz) { return kpar_dz_of(sum4(Chl), dz); }

// uniform (scalar) index into a 4-element register array without making it addressable
__device__ __forceinline__ double sel4(const double (&v)[4], int a) {
  return a == 0 ? v[0] : (a == 1 ? v[1] : (a == 2 ? v[2] : v[3]));
}

// total chlorophyll as DMS_SourceSink adds it up: sp + diat + diaz + phaeo (DMS_mod.F90:539-540)
__device__ __forceinline__ double dms_total_chl(const double (&chl)[4], const obgc_bgc_params& P) {
  return ((sel4(chl, P.sp_ind - 1) + sel4(chl, P.diat_ind - 1)) + sel4(chl, P.diaz_ind - 1)) + sel4(chl, P.phaeo_ind - 1);
}

// ------------------------------------------------------------------------- light: attenuation + scan
// Attenuation of PAR across each level is cell-local (one pow + one exp, BGC_mod.F90:914-923); only the
// product down the column is sequential.  One lane per cell computes exp(-KPARdz); the scan kernel then
// needs a single multiply per level.
// dms_flag (fused step only, else NULL): set for a column as soon as one of its cells has a DMS chlorophyll total that is
// not, bit for bit, the total BGC attenuates with; such a column gets a DMS light pass of its own from the scan kernel.
__device__ __forceinline__ void att_body(const BgcKernelArgs& A, const obgc_bgc_params& P, int* dms_flag) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  if (k >= A.kmax[c]) return;
  const long long ix = g.at(k, c);
  Pools p;
  double chl0[4];
  load_pools(A, P.autotrophs, ix, p, chl0);
  const double KPARdz = kpar_dz(p.Chl, A.dz[ix]);
  A.scratch[(long long)S_KPARDZ * A.splane + ix] = KPARdz;
  A.scratch[(long long)S_ATT * A.splane + ix] = exp(-KPARdz);
  if (dms_flag && dms_total_chl(chl0, P) != sum4(p.Chl)) dms_flag[c] = 1;   // every writer stores the same value
}

__global__ __launch_bounds__(256) void bgc_att_kernel(const BgcKernelArgs A, const obgc_bgc_params P) { att_body(A, P, nullptr); }
__global__ __launch_bounds__(256) void fused_att_kernel(const BgcKernelArgs A, const obgc_bgc_params P, int* dms_flag) {
  att_body(A, P, dms_flag);
}

// the scan for one column; returns false for a column without active levels
__device__ __forceinline__ bool parscan_body(const BgcKernelArgs& A, int c, int& kmax_out) {
  const GridView& g = A.g;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;
  kmax_out = kmax;
  if (kmax < 1) return false;
  const double* att = A.scratch + (long long)S_ATT * A.splane;
  double* par_in = A.scratch + (long long)S_PARIN * A.splane;
  double PAR = dmax(0.0, A.sw[c]) * F_QSW_PAR;                     // BGC_mod.F90:813-814
  int k = 0;
  for (; k + 8 <= kmax; k += 8) {                                  // 8 independent loads in flight per lane
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = att[g.at(k + j, c)];
#pragma unroll
    for (int j = 0; j < 8; ++j) { par_in[g.at(k + j, c)] = PAR; PAR = PAR * a[j]; }   // PAR_out = PAR_in * exp(-KPARdz), 923
  }
  for (; k < kmax; ++k) { par_in[g.at(k, c)] = PAR; PAR = PAR * att[g.at(k, c)]; }
  return true;
}

__global__ __launch_bounds__(kBlockCol) void bgc_parscan_kernel(const BgcKernelArgs A) {
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= A.g.ncol) return;
  int kmax;
  (void)parscan_body(A, c, kmax);
}

// Fused step: the same scan, and for flagged columns the whole light pass of DMS_SourceSink (DMS_mod.F90:506-507, 538-551)
// from the clamped chlorophyll of the BGC tracer array.  Rare (a pool the zero-mask removes while it still holds
// chlorophyll), so one lane per column with a pow and an exp per level is good enough.
__global__ __launch_bounds__(kBlockCol) void fused_parscan_kernel(const BgcKernelArgs A, const obgc_bgc_params P, const FusedDmArgs F) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= g.ncol) return;
  int kmax;
  if (!parscan_body(A, c, kmax)) return;
  if (!F.dms || !(F.all_flagged || F.flag[c])) return;
  const long long pl = g.plane;
  double PAR = dmax(0.0, A.sw[c]) * F.f_qsw_par_dms;
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    double chl[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) chl[a] = dmax(0.0, A.tracers[(long long)(P.autotrophs[a].Chl_ind - 1) * pl + ix]);
    const double KPARdz = kpar_dz_of(dms_total_chl(chl, P), A.dz[ix]);
    const double att = exp(-KPARdz);
    F.d_kpardz[ix] = KPARdz;
    F.d_att[ix] = att;
    F.d_parin[ix] = PAR;
    PAR = PAR * att;
  }
}

// ------------------------------------------------------------------------------- cell kernel
// Register discipline: the zooplankton terms only need the grazing rates, which need nothing but the
// carbon pools -- so a cheap pre-pass produces f_zoo_detr / zoo_loss first.  Every summation chain of the
// reference that starts from a zooplankton term (DOP_prod, DOFe_prod, P_iron production ...) can then be
// accumulated in its original order inside ONE rolled loop over the autotrophs, and only scalar
// accumulators stay live across iterations (the fully unrolled form needed 293 registers).
struct Pool1 { double Chl, C, Fe, Si, Ca; };

// raw pools of one autotroph; every load is unconditional (missing Si / CaCO3 slots re-read the C plane)
__device__ __forceinline__ Pool1 load_pool_raw(const BgcKernelArgs& A, const obgc_autotroph& t, long long ix) {
  const long long pl = A.g.plane;
  Pool1 p;
  p.Chl = A.tracers[(long long)(t.Chl_ind - 1) * pl + ix];
  p.C = A.tracers[(long long)(t.C_ind - 1) * pl + ix];
  p.Fe = A.tracers[(long long)(t.Fe_ind - 1) * pl + ix];
  p.Si = A.tracers[(long long)((t.Si_ind > 0 ? t.Si_ind : t.C_ind) - 1) * pl + ix];
  p.Ca = A.tracers[(long long)((t.CaCO3_ind > 0 ? t.CaCO3_ind : t.C_ind) - 1) * pl + ix];
  return p;
}

// clamps and zero-mask (BGC_mod.F90:764-785, 826-844)
__device__ __forceinline__ Pool1 mask_pool(const obgc_autotroph& t, Pool1 p) {
  p.Chl = dmax(0.0, p.Chl);
  p.C = dmax(0.0, p.C);
  p.Fe = dmax(0.0, p.Fe);
  p.Si = (t.Si_ind > 0) ? dmax(0.0, p.Si) : 0.0;
  p.Ca = (t.CaCO3_ind > 0) ? dmax(0.0, p.Ca) : 0.0;
  bool empty = p.Chl == 0.0 || p.C == 0.0 || p.Fe == 0.0;
  if (t.Si_ind > 0) empty = empty || p.Si == 0.0;
  if (empty) { p.Chl = 0.0; p.C = 0.0; p.Fe = 0.0; p.Si = 0.0; p.Ca = 0.0; }
  return p;
}

__device__ __forceinline__ double pick_C(const Pool1 (&r)[4], int a) { return a == 0 ? r[0].C : (a == 1 ? r[1].C : (a == 2 ? r[2].C : r[3].C)); }
__device__ __forceinline__ double pick_Chl(const Pool1 (&r)[4], int a) {
  return a == 0 ? r[0].Chl : (a == 1 ? r[1].Chl : (a == 2 ? r[2].Chl : r[3].Chl));
}

struct Flux { double s, h; };   // soft / hard sub-class flux leaving a level

// Everything one level of the sweep reads.  The loads do not depend on the recurrence, so the next level's
// set is requested before the current level is evaluated (one wave per SIMD cannot hide HBM latency any
// other way: 100k columns are only 1 563 wavefronts).
template <bool DIAG>
struct LevelIn {
  double dz, zb, fesed, T, NO3, Fe, O2;
  double prodPOC, prodCa, prodSi, PfeLoc;
  double L_no3, L_nh4, L_fe, L_sio3, L_po4, L_donr, L_dopr, L_dic, L_alk, L_o2;
  double oc0, d0, jc, jn, jp, jsi;   // DIAG only
  double chl100;                     // DIAG, column-block kernel only: chlorophyll x thickness above 100 m
  double S, zmid, PO4c, SiO3c;       // column-block kernel that solves the carbonate system itself: what that block shares with
                                     // the cell terms stays in registers instead of being read again (clamped PO4, SiO3)
};

// Dissolution length scales and decay factors across one level (2273-2339): cell-local functions of zbot, T, O2 and dz.
// TfuncS = Q_10 ** (((T + T0) - (Tref + T0)) / 10), the same number the ecosystem terms use.
struct LevelDerived {
  double poc_diss, sio2_diss, caco3_diss;
  double decay_POC_E, decay_SiO2, decay_CaCO3, decay_dust, DECAY_Hard, DECAY_HardDust;
};

__device__ __forceinline__ LevelDerived derive_level(const obgc_bgc_params& P, const double* rc, double zb, double TfuncS, double O2, double dz) {
  double scalelength;
  {
    const double* sz = P.parm_scalelen_z;
    const double* sv = P.parm_scalelen_vals;
    if (zb < sz[0]) scalelength = sv[0];
    else if (zb >= sz[3]) scalelength = sv[3];
    else if (zb < sz[1]) scalelength = sv[0] + div_by((sv[1] - sv[0]) * (zb - sz[0]), sz[1] - sz[0], rc[R_SCALELEN_DZ]);
    else if (zb < sz[2]) scalelength = sv[1] + div_by((sv[2] - sv[1]) * (zb - sz[1]), sz[2] - sz[1], rc[R_SCALELEN_DZ + 1]);
    else scalelength = sv[2] + div_by((sv[3] - sv[2]) * (zb - sz[2]), sz[3] - sz[2], rc[R_SCALELEN_DZ + 2]);
  }
  double poc_diss = P.parm_POC_diss;
  if ((O2 >= 5.0) && (O2 < 40.0)) poc_diss = P.parm_POC_diss * (1.0 + div_by((3.3 - 1.0) * (40.0 - O2), 35.0, rc[R_35]));
  else if (O2 < 5.0) poc_diss = P.parm_POC_diss * 3.3;
  poc_diss = scalelength * poc_diss;
  double sio2_diss = scalelength * P.parm_SiO2_diss;
  const double caco3_diss = scalelength * P.parm_CaCO3_diss;
  const double dust_diss = scalelength * DUST_DISS;
  sio2_diss = sio2_diss / TfuncS;
  LevelDerived d;
  d.poc_diss = poc_diss; d.sio2_diss = sio2_diss; d.caco3_diss = caco3_diss;
  d.decay_POC_E = exp(-dz / poc_diss); d.decay_SiO2 = exp(-dz / sio2_diss); d.decay_CaCO3 = exp(-dz / caco3_diss);
  d.decay_dust = exp(-dz / dust_diss); d.DECAY_Hard = exp(div_by(-dz, 4.0e6, rc[R_4E6])); d.DECAY_HardDust = exp(div_by(-dz, 1.2e7, rc[R_12E7]));
  return d;
}

// ---- the column-block kernel's view of a cell (bgc_column_kernel below) ----
// A workgroup of kColLevels wavefronts owns 64 columns; wavefront w evaluates level 4 j + w of chunk j, lane = column.
constexpr int kColLevels = 4;
constexpr int kColThreads = 64 * kColLevels;

// Hand-off of a cell to the wavefront that walks the column and back (LDS, one fp64 slot per lane and quantity).  The
// same 29 slots serve, one after the other, as the cell terms' own scratch (20 masked pools + 9 prologue results), as the
// inputs of the recurrence (H_IN_*), and as its results for the cell's finishing arithmetic (H_OUT_*: the walking lane
// reads a cell's inputs into registers before it writes that cell's results).
enum HandoffSlot {
  H_DZ, H_ZB, H_FESED, H_NO3, H_FE, H_O2,
  H_POC_DISS, H_SIO2_DISS, H_CACO3_DISS, H_DEC_POC, H_DEC_SI, H_DEC_CA, H_DEC_DUST, H_DEC_HARD, H_DEC_HARDDUST,
  H_PRODPOC, H_PRODCA, H_PRODSI, H_PFELOC,
  H_NIN,
  // the local parts of the ten particle-dependent tendencies wait here for the cell's own finishing arithmetic (slots the
  // recurrence does not touch) instead of in 20 registers across the barriers
  H_L_NO3 = H_NIN, H_L_NH4, H_L_FE, H_L_SIO3, H_L_PO4, H_L_DONR, H_L_DOPR, H_L_DIC, H_L_ALK, H_L_O2,
  H_NSLOTS
};
static_assert(H_NSLOTS == 29, "the slots also serve as the cell terms' 20 pool + 9 prologue scratch values");
enum HandoffOutSlot {
  H_OUT_REMINPOC, H_OUT_REMINCA, H_OUT_REMINSI, H_OUT_REMINFE, H_OUT_FESCAV, H_OUT_SEDDEN, H_OUT_OTHER,
  H_OUT_NBASE,
  H_OUT_INPOC = H_OUT_NBASE, H_OUT_INCA, H_OUT_INSI, H_OUT_INDUST, H_OUT_INFE, H_OUT_REMINDUST, H_OUT_FESCAVRATE, H_OUT_SEDPOC,
  H_OUT_SEDCA, H_OUT_SEDSI, H_OUT_SEDDUST, H_OUT_SEDFE,
  H_OUT_NDIAG
};
static_assert((int)H_NIN <= (int)H_NSLOTS && (int)H_OUT_NDIAG <= (int)H_NSLOTS, "hand-off values must fit the slots");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store the wavefront has in
// flight (s_waitcnt vmcnt(0)) -- here that would drain ~150 diagnostic stores per lane three times per chunk, although
// nothing another wavefront reads travels through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ColCtx {
  int k, c, w, lane;      // level, column, wavefront (= level within the chunk), lane (= column within the block)
  bool valid;             // c < ncol && k < nlev; invalid lanes load from clamped addresses and only keep the barriers company
  int parity;             // chunk parity: which PAR carry is read, which is written
  double* ho;             // [H_N*][kColThreads]
  double* att;            // [kColLevels][64]  exp(-KPARdz) of the chunk's levels
  double* par;            // [2][64]           PAR entering the chunk's first level
  double* attd;           // the same two for DMS's own light (fused step)
  double* pard;
};

// DM: DMS_SourceSink and MACROS_SourceSink of the same cell are evaluated too (obgc_fused_step): the tracers they share
// with BGC are in registers already, only their own five tracers are extra loads.
// COLK: false = cell-parallel kernel (flat cell index; light from the attenuation / scan kernels' planes; hand-off to the
// sweep kernel through global scratch planes and parked tendencies); true = column-block kernel (cell given by X;
// attenuation computed here and scanned down the chunk through LDS; hand-off to the sweep wavefront through LDS).
// Everything one cell's terms read from global memory, as values.
struct CellIn {
  int kmax;
  double lat, T, dz, zc, zb, fesed, salt, ztop;
  double PO4, NO3, SiO3, NH4, Fe, O2, DOC, DON, DOFe, DOP, DOPr, DONr, zooC;   // raw tracer values
  Pool1 raw[4];
  double o_dms, o_dmsp, o_sst, o_spca, o_prot, o_poly, o_lip;                  // fused step: own tracers of DMS / MACROS
  double PAR_in, KPARdz, att;                                                   // cell-parallel kernel: the light planes
  double d_parin, d_att, d_kpardz;
  bool dflag;                                                                   // the column has a DMS light pass of its own
};

// Every global load of a cell is issued here, back to back, before anything waits: ~45 independent requests per lane and a
// single memory round trip per wavefront.  Cells below the sea floor load too (valid addresses, values discarded) so that
// the kmax test does not cost a round trip of its own.  (k, c) must be inside the grid.
template <bool DIAG, bool DM, bool COLK, bool CARB = false>
__device__ __forceinline__ void cell_load(const BgcKernelArgs& A, const obgc_bgc_params& P, const FusedDmArgs* F, int k, int c,
                                          CellIn& in) {
  const GridView& g = A.g;
  const long long ix = g.at(k, c), pl = g.plane;
  const obgc_autotroph* at = P.autotrophs;
#define RAW(slot) A.tracers[(long long)((slot)-1) * pl + ix]
#define SCR(s) A.scratch[(long long)(s) * A.splane + ix]
  in.kmax = A.kmax[c];
  in.dflag = false;
  if (!COLK && DM && F->dms) in.dflag = F->all_flagged || F->flag[c] != 0;
  in.lat = A.lat[c];
  in.T = A.temp[ix]; in.dz = A.dz[ix]; in.zc = A.zmid[ix];
  in.PO4 = RAW(P.ind[0]); in.NO3 = RAW(P.ind[1]); in.SiO3 = RAW(P.ind[2]); in.NH4 = RAW(P.ind[3]); in.Fe = RAW(P.ind[4]);
  in.O2 = RAW(P.ind[5]); in.DOC = RAW(P.ind[9]); in.DON = RAW(P.ind[10]); in.DOFe = RAW(P.ind[11]); in.DOP = RAW(P.ind[12]);
  in.DOPr = RAW(P.ind[13]); in.DONr = RAW(P.ind[14]); in.zooC = RAW(P.ind[15]);
  in.PAR_in = 0.0; in.KPARdz = 1.0; in.att = 0.0;
  if (!COLK) { in.PAR_in = SCR(S_PARIN); in.KPARdz = SCR(S_KPARDZ); in.att = SCR(S_ATT); }
  in.zb = 0.0; in.fesed = 0.0;
  if (COLK) { in.zb = A.zbot[ix]; in.fesed = A.fesed[ix]; }
  in.salt = 0.0; in.ztop = 0.0;
  if (CARB && !DIAG) in.salt = A.salt[ix];
  if (DIAG) {
    in.salt = A.salt[ix];
    in.ztop = A.zbot[g.at(k > 0 ? k - 1 : 0, c)];
    if (k == 0) in.ztop = 0.0;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) in.raw[a] = load_pool_raw(A, at[a], ix);
  in.o_dms = in.o_dmsp = in.o_sst = in.o_spca = in.o_prot = in.o_poly = in.o_lip = 0.0;
  in.d_parin = 0.0; in.d_att = 0.0; in.d_kpardz = 1.0;
  if (DM) {
    if (F->dms) { in.o_dms = F->dms[ix]; in.o_dmsp = F->dmsp[ix]; in.o_sst = F->SST[c]; in.o_spca = RAW(P.ind[19]); }
    if (F->prot) { in.o_prot = F->prot[ix]; in.o_poly = F->poly[ix]; in.o_lip = F->lip[ix]; }
    if (!COLK && in.dflag) { in.d_parin = F->d_parin[ix]; in.d_att = F->d_att[ix]; in.d_kpardz = F->d_kpardz[ix]; }   // last: the only dependent loads
  }
#undef RAW
#undef SCR
}

// COLK only: *lo receives what the cell's finishing arithmetic needs once the recurrence has passed (the local parts of
// the ten particle-dependent tendencies, the local parts of its diagnostics); the return value says whether this lane
// evaluated an active cell.  `in` = the cell's loaded values (cell_load) for the cell (k, c) given here.
// FD: the caller has checked that every diagnostic the routine writes was passed: no per-member test (two scalar
// instructions a store, ~300 per cell).
template <bool DIAG, bool DM, bool COLK, bool FD = false>
__device__ __forceinline__ bool cell_body(const BgcKernelArgs& A, const obgc_bgc_params& P, const FusedDmArgs* F, const FusedParams* FP,
                                          const ColCtx* X, LevelIn<DIAG>* lo, int k, int c, const CellIn& in) {
  const GridView& g = A.g;
  const long long ix = g.at(k, c), pl = g.plane;
  const obgc_bgc_diag& D = A.diag;
  const obgc_autotroph* at = P.autotrophs;
#define RAW(slot) A.tracers[(long long)((slot)-1) * pl + ix]
#define TEND(slot) A.tend[(long long)((slot)-1) * pl + ix]
#define SCR(s) A.scratch[(long long)(s) * A.splane + ix]
// A lane whose cell lies below the sea floor (or on land) runs the same statements on whatever its arrays hold there and
// stores the reference's zero fill of tendencies and diagnostics (BGC_mod.F90:570, 625-727; DMS_mod.F90:413 and
// MACROS_mod.F90:267 for the tendencies of the other two modules, whose diagnostics the reference leaves alone) through the
// SAME store instructions: Z(v) = 0 for it.  A wavefront that straddles the floor -- one in five at 1 M x 80 -- then issues
// every store once, over full rows, instead of once per side of the floor.
#define Z(v) (live ? (v) : 0.0)
#define ST_KC(name, v) do { if (DIAG && (FD || D.name)) OBGC_STORE(&D.name[ix], Z(v)); } while (0)
#define ST_KCA(name, a, v) do { if (DIAG && (FD || D.name)) OBGC_STORE(&D.name[(long long)(a) * pl + ix], Z(v)); } while (0)
#define ST_TEND(slot, v) OBGC_STORE(&TEND(slot), Z(v))
  bool live = true;
  int kmax = in.kmax;
  bool dflag = in.dflag;
  const double lat = in.lat, T = in.T, dz = in.dz, zc = in.zc;
  const double r_PO4 = in.PO4, r_NO3 = in.NO3, r_SiO3 = in.SiO3, r_NH4 = in.NH4, r_Fe = in.Fe, r_O2 = in.O2, r_DOC = in.DOC,
               r_DON = in.DON, r_DOFe = in.DOFe, r_DOP = in.DOP, r_DOPr = in.DOPr, r_DONr = in.DONr, r_zooC = in.zooC;
  double PAR_in = in.PAR_in, KPARdz = in.KPARdz, att = in.att;
  const double zb = in.zb, r_fesed = in.fesed, r_salt = in.salt, ztop = in.ztop;
  const Pool1 raw[4] = {in.raw[0], in.raw[1], in.raw[2], in.raw[3]};
  const double o_dms = in.o_dms, o_dmsp = in.o_dmsp, o_sst = in.o_sst, o_spca = in.o_spca, o_prot = in.o_prot, o_poly = in.o_poly,
               o_lip = in.o_lip;
  double d_parin = in.d_parin, d_att = in.d_att, d_kpardz = in.d_kpardz;
  if (kmax > g.nlev) kmax = g.nlev;

  if (COLK) {
    // Light (BGC_mod.F90:907-924): the attenuation of this level from the masked chlorophyll, then the product down the
    // chunk's levels in the reference's order: PAR entering level w = ((carry * att_0) * att_1) ... * att_(w-1).
    const int w = X->w, lane = X->lane;
    double chl[4], chl0[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { chl[a] = mask_pool(at[a], raw[a]).Chl; chl0[a] = dmax(0.0, raw[a].Chl); }
    KPARdz = kpar_dz(chl, dz);
    att = exp(-KPARdz);
    X->att[w * 64 + lane] = att;
    if (DM && F->dms) {
      // DMS attenuates with the clamped, unmasked chlorophyll and its own PAR fraction (DMS_mod.F90:506-507, 539-550): the
      // same numbers as BGC's unless the mask removed a pool with chlorophyll left in it
      d_kpardz = KPARdz; d_att = att;
      if (dms_total_chl(chl0, P) != sum4(chl)) { d_kpardz = kpar_dz_of(dms_total_chl(chl0, P), dz); d_att = exp(-d_kpardz); }
      X->attd[w * 64 + lane] = d_att;
    }
    lds_barrier();
    double PAR = X->par[X->parity * 64 + lane];
    for (int j = 0; j < w; ++j) PAR = PAR * X->att[j * 64 + lane];
    PAR_in = PAR;
    if (w == kColLevels - 1) X->par[(X->parity ^ 1) * 64 + lane] = PAR_in * att;
    if (DM && F->dms) {
      double PARd = X->pard[X->parity * 64 + lane];
      for (int j = 0; j < w; ++j) PARd = PARd * X->attd[j * 64 + lane];
      d_parin = PARd;
      if (w == kColLevels - 1) X->pard[(X->parity ^ 1) * 64 + lane] = PARd * d_att;
      dflag = true;   // DMS always takes its own (usually identical) numbers here
    }
    if (!X->valid) return false;
  }
  live = k < kmax;
  if (!COLK && !live) {
    // separate kernels: the sweep only visits cells above the floor, so what only IT writes gets the zero fill here
    TEND(P.ind[7]) = 0.0;
    if (DIAG) {
#define X(n) if (D.n) D.n[ix] = 0.0;
      OBGC_SWEEP_DIAG_KC(X)
#undef X
    }
  }


  const double T0K = P.T0_Kelvin_BGC;
  const bool north = lat >= 0.0, south_or_eq = lat <= 0.0;
  // clamped local copies (747-762)
  const double PO4 = dmax(0.0, r_PO4), NO3 = dmax(0.0, r_NO3), SiO3 = dmax(0.0, r_SiO3), NH4 = dmax(0.0, r_NH4), Fe = dmax(0.0, r_Fe),
               DOP = dmax(0.0, r_DOP), zooC = dmax(0.0, r_zooC);

  // light (907-924): attenuation and PAR entering the level come from the att / scan kernels
  const double PAR_out = PAR_in * att;
  const double PAR_avg = PAR_in * (1.0 - att) / KPARdz;

  if (DM) {
    // DMS_SourceSink (DMS_mod.F90:516-761) and MACROS_SourceSink (MACROS_mod.F90:331-406) of this cell, first: every value
    // they need is loaded, nothing of theirs stays live afterwards.  They see the clamped tracers, not the zero-masked pools.
    const double c_zoo = dmax(0.0, r_zooC), c_sp = dmax(0.0, pick_C(raw, P.sp_ind - 1)), c_diat = dmax(0.0, pick_C(raw, P.diat_ind - 1)),
                 c_diaz = dmax(0.0, pick_C(raw, P.diaz_ind - 1)), c_phaeo = dmax(0.0, pick_C(raw, P.phaeo_ind - 1));
    if (F->dms) {
      // the level mean of PAR as DMS computes it (551): BGC's own value unless the column has its own light pass
      const double PAR_avg_dms = dflag ? d_parin * (1.0 - d_att) / d_kpardz : PAR_avg;
      const DmsColumn col = dms_column_terms(FP->dms, o_sst);
      double t_dms, t_dmsp;
      dms_cell_terms<true, FD>(FP->dms, F->ddiag, ix, col, dmax(0.0, o_dms), dmax(0.0, o_dmsp), c_zoo, c_sp, dmax(0.0, o_spca), c_diat,
                           c_diaz, c_phaeo, dmax(0.0, pick_Chl(raw, P.sp_ind - 1)), PAR_avg_dms, t_dms, t_dmsp, live);
      F->dms_tend[ix] = Z(t_dms);
      F->dmsp_tend[ix] = Z(t_dmsp);
    }
    if (F->prot) {
      double t_prot, t_poly, t_lip;
      macros_cell_terms<true, FD>(FP->macros, F->mdiag, ix, c_zoo, c_sp, c_diat, c_diaz, c_phaeo, dmax(0.0, o_prot), dmax(0.0, o_poly),
                              dmax(0.0, o_lip), t_prot, t_poly, t_lip, live);
      F->prot_tend[ix] = Z(t_prot);
      F->poly_tend[ix] = Z(t_poly);
      F->lip_tend[ix] = Z(t_lip);
    }
  }

  // Everything below the autotroph loop that needs a global LOAD is evaluated now and parked in LDS: loads and
  // stores retire through one in-order counter, so a load issued after the loop's ~100 diagnostic stores would
  // wait for every one of them to be acknowledged.  From here on the kernel only stores.
  //   DOM remineralisation (1436-1461), nitrification (1554-1561), the O2 weights of denitrification and
  //   O2 consumption (1569-1577, 1784-1790), AOU, the 100 m weight of the Chl integral
  constexpr int NT = COLK ? kColThreads : kBlockCell;
  __shared__ double s_tail_own[COLK ? 1 : 9][COLK ? 1 : kBlockCell];
  const int tid = threadIdx.x;
  // the column-block kernel lends its hand-off slots: pools in slots 0..19, prologue results in 20..28
  double (*s_tail)[NT] = COLK ? (double (*)[NT])(X->ho + 20 * NT) : (double (*)[NT])s_tail_own;
  double aou = 0.0;
  {
    const double O2 = dmax(0.0, r_O2), DOC = dmax(0.0, r_DOC), DON = dmax(0.0, r_DON), DOFe = dmax(0.0, r_DOFe),
                 DOPr = dmax(0.0, r_DOPr), DONr = dmax(0.0, r_DONr);
    double DOC_remin = DOC * DOC_REMINR, DON_remin = DON * DON_REMINR, DOFe_remin = DOFe * DOFE_REMINR,
           DOP_remin = DOP * DOP_REMINR, DONr_remin, DOPr_remin;
    if (PAR_avg > 1.0) {
      DONr_remin = DONr * DONR_REMINR;
      DOPr_remin = DOPr * DOPR_REMINR;
    } else {   // below the euphotic zone (1451-1461)
      DONr_remin = DONr * (1.0 / (365.0 * 670.0)) * DPS;
      DOPr_remin = DOPr * (1.0 / (365.0 * 460.0)) * DPS;
      DOC_remin = DOC_remin * 0.0685;
      DON_remin = DON_remin * 0.1;
      DOFe_remin = DOFe_remin * 0.05;
      DOP_remin = DOP_remin * 0.05;
    }
    double NITRIF;
    if (PAR_out < P.parm_nitrif_par_lim) {
      NITRIF = P.parm_kappa_nitrif * NH4;
      if (PAR_in > P.parm_nitrif_par_lim) NITRIF = NITRIF * log(PAR_out / P.parm_nitrif_par_lim) / (-KPARdz);
    } else NITRIF = 0.0;
    double wd = div_by((P.parm_o2_min + P.parm_o2_min_delta) - O2, P.parm_o2_min_delta, A.recip[R_O2DELTA]);
    wd = dmin(dmax(wd, 0.0), 1.0);
    if (NO3 == 0.0) wd = 0.0;
    double wo = div_by(O2 - P.parm_o2_min, P.parm_o2_min_delta, A.recip[R_O2DELTA]);
    wo = dmin(dmax(wo, 0.0), 1.0);
    s_tail[0][tid] = DOC_remin; s_tail[1][tid] = DON_remin; s_tail[2][tid] = DOFe_remin; s_tail[3][tid] = DOP_remin;
    s_tail[4][tid] = DONr_remin; s_tail[5][tid] = DOPr_remin; s_tail[6][tid] = NITRIF;
    s_tail[7][tid] = wd * div_by(DOC_remin, DENITRIF_C_N, A.recip[R_DENITRIF]);
    s_tail[8][tid] = wo;
    if (DIAG && D.AOU) aou = o2sat(T, r_salt, T0K) - O2;
  }
  double pt100 = 0.0;   // thickness of this level above 100 m (1940-1945)
  if (DIAG) {
    const double w = dmin(100.0e2 - ztop, dz);
    pt100 = (w > 0.0) ? w : 0.0;
  }

  // temperature response and loss thresholds (1041-1055)
  const double Tfunc = exp(div_by((T + T0K) - (TREF + T0K), 10.0, A.recip[R_TEN]) * LN_Q_10);   // Q_10 ** ((T + T0) - (Tref + T0)) / 10)
  double f_loss_thres;
  if (zc > THRES_Z1) f_loss_thres = (zc < THRES_Z2) ? (THRES_Z2 - zc) / (THRES_Z2 - THRES_Z1) : 0.0;
  else f_loss_thres = 1.0;

  // ---- pre-pass: Pprime (1072-1094), grazing (1297-1324), zooplankton losses (1395-1415) ----
  // The masked pools are read from HBM once and parked in LDS for the rolled loop below (each lane only
  // ever touches its own column of the array, so no barrier is needed).
  __shared__ double s_pool_own[COLK ? 1 : 20][COLK ? 1 : kBlockCell];
  double (*s_pool)[NT] = COLK ? (double (*)[NT])X->ho : (double (*)[NT])s_pool_own;
  double Pprime[4], graze[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const Pool1 p = mask_pool(at[a], raw[a]);
    s_pool[a][tid] = p.Chl; s_pool[4 + a][tid] = p.C; s_pool[8 + a][tid] = p.Fe; s_pool[12 + a][tid] = p.Si; s_pool[16 + a][tid] = p.Ca;
    double C_loss_thres = f_loss_thres * at[a].loss_thres;
    if (at[a].temp_function == TFNC_Q10) {
      if (T < at[a].temp_thres) C_loss_thres = f_loss_thres * at[a].loss_thres2;
    } else if (at[a].temp_function == TFNC_QUASI_MMRT) {
      const double tmpTmax = north ? uniform(at[a].temp_thresN) : uniform(at[a].temp_thresS);
      if (T > tmpTmax) C_loss_thres = f_loss_thres * at[a].loss_thres2;
    }
    Pprime[a] = dmax(p.C - C_loss_thres, 0.0);
  }
  double w1 = 0.0, w2 = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double cls = 0.0;   // carbon of the grazee class this autotroph belongs to
#pragma unroll
    for (int a2 = 0; a2 < 4; ++a2)
      if (at[a2].grazee_ind == at[a].grazee_ind) cls = cls + Pprime[a2];
    double z_umax = at[a].z_umax_0 * Tfunc;
    if (a + 1 == P.diat_ind) {
      if (north && (T > at[a].temp_optN))
        z_umax = z_umax * dmax((at[a].temp_thresN - T) / (at[a].temp_thresN - at[a].temp_optN), 0.95);
      else if (south_or_eq && (T > at[a].temp_optS))
        z_umax = z_umax * dmax((at[a].temp_thresS - T) / (at[a].temp_thresS - at[a].temp_optS), 0.95);
    }
    graze[a] = (cls > 0.0) ? (Pprime[a] / cls) * z_umax * zooC * (cls / (cls + at[a].z_grz)) : 0.0;
    w1 = w1 + at[a].f_zoo_detr * (graze[a] + EPSC * EPSTINV);
    w2 = w2 + (graze[a] + EPSC * EPSTINV);
  }
  const double f_zoo_detr = w1 / w2;
  if (DIAG) { ST_KC(AOU, aou); ST_KC(PAR_avg, PAR_avg); }   // first stores: every load has been issued by now
  const double Zprime = dmax(zooC - f_loss_thres * LOSS_THRES_ZOO, 0.0);
  const double zoo_loss = (P.parm_z_mort2_0 * (Zprime * sqrt(Zprime)) + P.parm_z_mort_0 * Zprime) * Tfunc;   // Zprime ** 1.5
  const double zoo_loss_doc = (1.0 - P.parm_labile_ratio) * (1.0 - f_zoo_detr) * zoo_loss;
  const double zoo_loss_dic = P.parm_labile_ratio * (1.0 - f_zoo_detr) * zoo_loss;

  // ---- autotroph kinetics (1107-1388): one rolled loop, scalar accumulators only ----
  double DOP_prod = QP_ZOO_POM * zoo_loss_doc;                 // 1423-1430
  double DOFe_prod = QFE_ZOO * zoo_loss_doc;                   // 1431-1434
  double PfeLoc = (zoo_loss * f_zoo_detr * QFE_ZOO);           // 1524-1529 without the scavenging term
  double sNO3V = 0, sNH4V = 0, sPO4V = 0, sDOPV = 0, sPhotoFe = 0, sPhotoC = 0, s_gr_zoo = 0, s_ls_doc = 0, s_gr_doc = 0,
         s_ls_dic = 0, s_gr_dic = 0, s_gr_poc = 0, s_agg = 0, s_ls_poc = 0;
  double L_fe_a = 0.0, L_po4_a = 0.0, L_sio3_a = 0.0, nexc = 0.0, ca_x = 0.0, ca_prod = 0.0, O2_PRODUCTION = 0.0;
  double d_totCa = 0.0, d_totNfix = 0.0, d_pCNO3 = 0.0, prodCa = 0.0, prodSi = 0.0;
  double sC = 0.0, sCa = 0.0, sSi = 0.0, sNfix = 0.0, sQpC = 0.0, chl = 0.0;   // diagnostics: integrand pieces
#pragma unroll 1
  for (int a = 0; a < 4; ++a) {
    const obgc_autotroph& t = at[a];
    const Pool1 p{s_pool[a][tid], s_pool[4 + a][tid], s_pool[8 + a][tid], s_pool[12 + a][tid], s_pool[16 + a][tid]};
    const double Pp = sel4(Pprime, a), gz = sel4(graze, a);
    // quotas for growth (850-898)
    const double Cden = p.C + EPSC;
    const double thetaC = p.Chl / Cden;
    const double Qfe = p.Fe / Cden;
    const double Qsi = (t.Si_ind > 0) ? dmin(p.Si / Cden, GQSI_MAX) : 0.0;
    double gQfe = t.gQfe_0;
    if (Fe < CKS * t.kFe) gQfe = dmax(div_by(gQfe * Fe, CKS * t.kFe, A.recip[R_CKS_KFE + a]), t.gQfe_min);
    double gQsi = 0.0;
    if (t.Si_ind > 0) {
      gQsi = GQSI_0;
      if ((Fe < CKSI * t.kFe) && (Fe > 0.0) && (SiO3 > (CKSI * t.kSiO3))) gQsi = dmin((gQsi * CKSI * t.kFe / Fe), GQSI_MAX);
      if (Fe == 0.0) gQsi = GQSI_MAX;
      if (SiO3 < (CKSI * t.kSiO3)) gQsi = dmax(div_by(gQsi * SiO3, CKSI * t.kSiO3, A.recip[R_CKSI_KSIO3 + a]), GQSI_MIN);
    }
    double QCa = 0.0;
    if (t.CaCO3_ind > 0) {
      QCa = p.Ca / Cden;
      if (QCa > QCACO3_MAX) QCa = QCACO3_MAX;
    }
    // nutrient limitation (1108-1139)
    const double rNO3 = div_by(NO3, t.kNO3, A.recip[R_KNO3 + a]), rNH4 = div_by(NH4, t.kNH4, A.recip[R_KNH4 + a]);
    const double VNO3 = rNO3 / (1.0 + rNO3 + rNH4);
    const double VNH4 = rNH4 / (1.0 + rNO3 + rNH4);
    double VNtot = VNO3 + VNH4;
    if (t.Nfixer) VNtot = 1.0;
    ST_KCA(N_lim, a, VNtot);
    const double VFe = Fe / (Fe + t.kFe);
    ST_KCA(Fe_lim, a, VFe);
    double f_nut = dmin(VNtot, VFe);
    const double rPO4 = div_by(PO4, t.kPO4, A.recip[R_KPO4 + a]), rDOP = div_by(DOP, t.kDOP, A.recip[R_KDOP + a]);
    const double VPO4 = rPO4 / (1.0 + rPO4 + rDOP);
    const double VDOP = rDOP / (1.0 + rPO4 + rDOP);
    const double VPtot = VPO4 + VDOP;
    ST_KCA(P_lim, a, VPtot);
    f_nut = dmin(f_nut, VPtot);
    double VSiO3 = 0.0;
    if (t.kSiO3 > 0.0) {
      VSiO3 = SiO3 / (SiO3 + t.kSiO3);
      f_nut = dmin(f_nut, VSiO3);
    }
    ST_KCA(SiO3_lim, a, VSiO3);
    // photosynthesis (1146-1181)
    double PCmax = t.PCref * f_nut * Tfunc;
    if (T < t.temp_thres) PCmax = 0.0;
    if (t.temp_function == TFNC_QUASI_MMRT) {
      const double tmpTopt = north ? uniform(t.temp_optN) : uniform(t.temp_optS);
      const double tmpTmax = north ? uniform(t.temp_thresN) : uniform(t.temp_thresS);
      PCmax = PCmax * dmin(1.0, ((tmpTmax - T) / (tmpTmax - tmpTopt)));
      if (T > tmpTmax) PCmax = 0.0;
    }
    const double light_lim = (1.0 - exp((-1.0 * t.alphaPI * thetaC * PAR_avg) / (PCmax + EPSTINV)));
    const double PCphoto = PCmax * light_lim;
    ST_KCA(light_lim, a, light_lim);
    const double photoC = PCphoto * p.C;
    double NO3_V, NH4_V, VNC, pC_NO3;
    if (VNtot > 0.0) {
      NO3_V = (VNO3 / VNtot) * photoC * QN;
      NH4_V = (VNH4 / VNtot) * photoC * QN;
      VNC = PCphoto * QN;
      pC_NO3 = (VNO3 / VNtot) * photoC;            // 1828-1834
    } else { NO3_V = 0.0; NH4_V = 0.0; VNC = 0.0; pC_NO3 = 0.0; }
    ST_KCA(photoNO3, a, NO3_V);
    ST_KCA(photoNH4, a, NH4_V);
    double PO4_V, DOP_V;
    if (VPtot > 0.0) {
      PO4_V = (VPO4 / VPtot) * photoC * t.Qp;
      DOP_V = (VDOP / VPtot) * photoC * t.Qp;
    } else { PO4_V = 0.0; DOP_V = 0.0; }
    ST_KCA(PO4_uptake, a, PO4_V);
    ST_KCA(DOP_uptake, a, DOP_V);
    const double photoFe = photoC * gQfe;
    ST_KCA(photoFe, a, photoFe);
    const double photoSi = (t.Si_ind > 0) ? photoC * gQsi : 0.0;
    double wk = t.alphaPI * thetaC * PAR_avg;
    double photoacc;
    if (wk > 0.0) {
      const double pChl = t.thetaN_max * PCphoto / wk;
      photoacc = (pChl * VNC / thetaC) * p.Chl;
    } else photoacc = 0.0;
    double CaProd = 0.0;
    if (t.imp_calcifier) {   // 1255-1278
      double cp = P.parm_f_prod_sp_CaCO3 * photoC;
      cp = cp * f_nut;
      if (T < CACO3_TEMP_THRES1) cp = cp * dmax((T - CACO3_TEMP_THRES2), 0.0) / (CACO3_TEMP_THRES1 - CACO3_TEMP_THRES2);
      if (p.C > CACO3_SP_THRES) cp = dmin((cp * p.C / CACO3_SP_THRES), (F_PHOTOSP_CACO3 * photoC));
      CaProd = cp;
      d_totCa = d_totCa + cp;
    }
    if (DIAG) {   // the sweep integrates these four profiles; caller's arrays or scratch
      A.z_photoC[(long long)a * pl + ix] = Z(photoC);
      A.z_photoC_NO3[(long long)a * pl + ix] = Z(pC_NO3);
      A.z_CaCO3_form[(long long)a * pl + ix] = Z(CaProd);
      A.z_bSi_form[(long long)a * pl + ix] = Z(photoSi);
    }
    // losses (1285-1290)
    const double loss = t.mort * Pp * Tfunc;
    double ag = dmin((t.agg_rate_max * DPS) * Pp, t.mort2 * Pp * Pp);
    ag = dmax((t.agg_rate_min * DPS) * Pp, ag);
    double Nfix = 0.0;
    if (t.Nfixer) {   // 1331-1338
      wk = photoC * QN;
      Nfix = (wk * R_NFIX_PHOTO) - NO3_V - NH4_V;
      nexc = nexc + (Nfix + NO3_V + NH4_V - wk);      // Nexcrete, added to NH4 in the sweep (1589-1592)
      d_totNfix = d_totNfix + Nfix;
    }
    ST_KCA(Nfix, a, Nfix);
    // routing (1354-1386)
    const double gr_zoo = t.graze_zoo * gz;
    double gr_poc;
    if (t.imp_calcifier) gr_poc = gz * dmax((CACO3_POC_MIN * QCa), dmin(SPC_POC_FAC * dmax(1.0, Pp), F_GRAZE_SP_POC_LIM));
    else gr_poc = t.graze_poc * gz;
    const double gr_doc = t.graze_doc * gz;
    const double gr_dic = gz - (gr_zoo + gr_poc + gr_doc);
    const double ls_poc = t.imp_calcifier ? QCa * loss : t.loss_poc * loss;
    const double ls_doc = (1.0 - P.parm_labile_ratio) * (loss - ls_poc);
    const double ls_dic = P.parm_labile_ratio * (loss - ls_poc);
    // chains in the reference's order
    if (t.Qp == QP_ZOO_POM) {
      DOP_prod = DOP_prod + t.Qp * (ls_doc + gr_doc);
      L_po4_a = L_po4_a + t.Qp * (ls_dic + gr_dic);
    } else {
      const double remaining_P = ((gz + loss + ag) * t.Qp) - ((gr_zoo) * QP_ZOO_POM) - ((gr_poc + ls_poc + ag) * QP_ZOO_POM);
      DOP_prod = DOP_prod + (1.0 - P.parm_labile_ratio) * remaining_P;
      L_po4_a = L_po4_a + P.parm_labile_ratio * remaining_P;
    }
    DOFe_prod = DOFe_prod + Qfe * (ls_doc + gr_doc);
    PfeLoc = PfeLoc + Qfe * (ag + gr_poc + ls_poc);
    sNO3V += NO3_V; sNH4V += NH4_V; sPO4V += PO4_V; sDOPV += DOP_V; sPhotoFe += photoFe; sPhotoC += photoC;
    s_gr_zoo += gr_zoo; s_ls_doc += ls_doc; s_gr_doc += gr_doc; s_ls_dic += ls_dic; s_gr_dic += gr_dic;
    s_gr_poc += gr_poc; s_agg += ag; s_ls_poc += ls_poc;
    d_pCNO3 = d_pCNO3 + pC_NO3;
    L_fe_a = L_fe_a + (Qfe * (ls_dic + gr_dic)) + gr_zoo * (Qfe - QFE_ZOO);                      // 1601-1605
    if (t.Si_ind > 0) {
      L_sio3_a = L_sio3_a - photoSi + Qsi * (F_GRAZE_SI_REMIN * gz + (1.0 - t.loss_poc) * loss);   // 1622-1628
      prodSi = Qsi * ((1.0 - F_GRAZE_SI_REMIN) * gz + ag + t.loss_poc * loss);                    // 1492-1498
    }
    if (t.CaCO3_ind > 0) {
      prodCa = ((1.0 - F_GRAZE_CACO3_REMIN) * gz + loss + ag) * QCa;                              // 1480-1485
      ca_x = ca_x + F_GRAZE_CACO3_REMIN * gz * QCa;                                               // 1734-1739, 1754-1759
      ca_prod = ca_prod + CaProd;
    }
    if (photoC > 0.0) {   // O2 production (1767-1782)
      if (!t.Nfixer) {
        const double den = NO3_V + NH4_V;
        O2_PRODUCTION = O2_PRODUCTION + photoC * (div_by(NO3_V / den, RED_D_C_O2, A.recip[R_RED]) +
                                                  div_by(NH4_V / den, REMIN_D_C_O2, A.recip[R_REMIN]));
      } else {
        const double den = NO3_V + NH4_V + Nfix;
        O2_PRODUCTION = O2_PRODUCTION + photoC * (div_by(NO3_V / den, RED_D_C_O2, A.recip[R_RED]) +
                                                  div_by(NH4_V / den, REMIN_D_C_O2, A.recip[R_REMIN]) +
                                                  div_by(Nfix / den, RED_D_C_O2_DIAZ, A.recip[R_RED_DIAZ]));
      }
    }
    // autotroph tendencies (1676-1697)
    wk = gz + loss + ag;
    const double t_C = photoC - wk;
    ST_TEND(t.C_ind, t_C);
    ST_TEND(t.Chl_ind, photoacc - thetaC * wk);
    ST_TEND(t.Fe_ind, photoFe - Qfe * wk);
    if (t.Si_ind > 0) { const double v = photoSi - Qsi * wk; ST_TEND(t.Si_ind, v); sSi = sSi + v; }
    if (t.CaCO3_ind > 0) { const double v = CaProd - QCa * wk; ST_TEND(t.CaCO3_ind, v); sCa = sCa + v; }
    if (DIAG) {
      ST_KCA(auto_graze, a, gz); ST_KCA(auto_loss, a, loss); ST_KCA(auto_agg, a, ag);
      sC = sC + t_C; sQpC = sQpC + t.Qp * t_C; sNfix = sNfix + Nfix; chl = chl + p.Chl;
    }
  }
  ST_KC(tot_CaCO3_form, d_totCa);
  ST_KC(tot_Nfix, d_totNfix);
  ST_KC(photoC_TOT, sPhotoC);
  ST_KC(photoC_NO3_TOT, d_pCNO3);
  ST_KC(auto_graze_TOT, sum4(graze));

  // ---- DOM (1421-1461): remineralisation terms come from the prologue ----
  const double DOC_prod = zoo_loss_doc + s_ls_doc + s_gr_doc;
  const double DON_prod = QN * DOC_prod;
  const double DOC_remin = s_tail[0][tid], DON_remin = s_tail[1][tid], DOFe_remin = s_tail[2][tid], DOP_remin = s_tail[3][tid],
               DONr_remin = s_tail[4][tid], DOPr_remin = s_tail[5][tid], NITRIF = s_tail[6][tid], D0 = s_tail[7][tid],
               wo = s_tail[8][tid];
  const double prodPOC = f_zoo_detr * zoo_loss + s_gr_poc + s_agg + s_ls_poc;        // 1467-1468

  double (*ho)[NT] = COLK ? (double (*)[NT])X->ho : nullptr;   // written below: every read of s_tail / s_pool is done
  if (COLK) { ho[H_PRODPOC][tid] = prodPOC; ho[H_PRODCA][tid] = prodCa; ho[H_PRODSI][tid] = prodSi; ho[H_PFELOC][tid] = PfeLoc; }
  else { SCR(S_PRODPOC) = prodPOC; SCR(S_PRODCA) = prodCa; SCR(S_PRODSI) = prodSi; SCR(S_PFELOC) = PfeLoc; }

  // ---- local parts of denitrification (1569-1577) and O2 consumption (1784-1790): the terms that do not
  //      involve the particle remineralisation, weighted in the prologue; the sweep adds  w * (remin terms)  ----
  const double S3 = zoo_loss_dic + s_ls_dic + s_gr_dic;
  const double CAD = ca_x - ca_prod;
  const double OC0 = wo * (div_by(DOC_remin + S3, REMIN_D_C_O2, A.recip[R_REMIN]) + (2.0 * NITRIF));

  // ---- tendencies that do not see the particles (1703-1723) ----
  const double t_zooC = s_gr_zoo - zoo_loss;
  const double t_doc = DOC_prod - DOC_remin;
  const double t_don = (DON_prod * (1.0 - DONREFRACT)) - DON_remin;
  const double t_dop = (DOP_prod * (1.0 - DOPREFRACT)) - DOP_remin - sDOPV;
  ST_TEND(P.ind[15], t_zooC);
  ST_TEND(P.ind[9], t_doc);
  ST_TEND(P.ind[10], t_don);
  ST_TEND(P.ind[11], DOFe_prod - DOFe_remin);
  ST_TEND(P.ind[12], t_dop);
  // ---- particle-dependent tendencies: everything local is folded here and waits in the caller's
  //      tendency plane; the sweep adds the remineralisation terms (listed with each line) ----
  const double L_no3 = (NITRIF - sNO3V) - D0;                                             // 1583: - wd*((reminPOC-OTHER)/C - SED) - SED
  const double L_nh4 = ((-sNH4V - NITRIF + DON_remin + DONr_remin) + QN * S3) + nexc;     // 1585-1592: + QN*(1-DONrefract)*reminPOC
  const double L_fe = ((QFE_ZOO * zoo_loss_dic) + DOFe_remin - sPhotoFe) + L_fe_a;        // 1598-1605: + reminFe - Fe_scavenge
  const double L_sio3 = L_sio3_a;                                                         // 1620-1628: + reminSi
  const double L_po4 = ((DOP_remin + DOPr_remin - sPO4V) + L_po4_a) + QP_ZOO_POM * zoo_loss_dic;   // 1643-1661: + Qp*(1-DOPrefract)*reminPOC
  const double L_donr = (DON_prod * DONREFRACT) - DONr_remin;                             // 1714: + reminPOC*DONrefract*Q
  const double L_dopr = (DOP_prod * DOPREFRACT) - DOPr_remin;                             // 1720: + reminPOC*DOPrefract*Qp
  const double L_dic = ((s_ls_dic + s_gr_dic - sPhotoC + DOC_remin) + zoo_loss_dic) + CAD;   // 1729-1739: + reminPOC + reminCaCO3
  const double L_alk = 2.0 * CAD;                                                         // 1749-1759: - t_no3 + t_nh4 + 2 reminCaCO3
  const double L_o2 = O2_PRODUCTION - OC0;                                                // 1790: - wo*(reminPOC - SED*C - OTHER)/R
  if (COLK) {
    ho[H_L_NO3][tid] = L_no3; ho[H_L_NH4][tid] = L_nh4; ho[H_L_FE][tid] = L_fe; ho[H_L_SIO3][tid] = L_sio3; ho[H_L_PO4][tid] = L_po4;
    ho[H_L_DONR][tid] = L_donr; ho[H_L_DOPR][tid] = L_dopr; ho[H_L_DIC][tid] = L_dic; ho[H_L_ALK][tid] = L_alk; ho[H_L_O2][tid] = L_o2;
    lo->dz = dz; lo->zb = zb; lo->NO3 = NO3; lo->O2 = dmax(0.0, r_O2); lo->Fe = Fe; lo->fesed = r_fesed; lo->T = T;
    lo->S = r_salt; lo->zmid = zc; lo->PO4c = PO4; lo->SiO3c = SiO3;
    lo->prodPOC = prodPOC; lo->prodCa = prodCa; lo->prodSi = prodSi; lo->PfeLoc = PfeLoc;
    // what the walking wavefront would otherwise read from HBM again or compute at one lane per column
    const LevelDerived dv = derive_level(P, A.recip, zb, Tfunc, dmax(0.0, r_O2), dz);
    ho[H_DZ][tid] = dz; ho[H_ZB][tid] = zb; ho[H_FESED][tid] = r_fesed; ho[H_NO3][tid] = NO3; ho[H_FE][tid] = Fe; ho[H_O2][tid] = dmax(0.0, r_O2);
    ho[H_POC_DISS][tid] = dv.poc_diss; ho[H_SIO2_DISS][tid] = dv.sio2_diss; ho[H_CACO3_DISS][tid] = dv.caco3_diss;
    ho[H_DEC_POC][tid] = dv.decay_POC_E; ho[H_DEC_SI][tid] = dv.decay_SiO2; ho[H_DEC_CA][tid] = dv.decay_CaCO3;
    ho[H_DEC_DUST][tid] = dv.decay_dust; ho[H_DEC_HARD][tid] = dv.DECAY_Hard; ho[H_DEC_HARDDUST][tid] = dv.DECAY_HardDust;
  } else {
    ST_TEND(P.ind[1], L_no3); ST_TEND(P.ind[3], L_nh4); ST_TEND(P.ind[4], L_fe); ST_TEND(P.ind[2], L_sio3); ST_TEND(P.ind[0], L_po4);
    ST_TEND(P.ind[14], L_donr); ST_TEND(P.ind[13], L_dopr); ST_TEND(P.ind[6], L_dic); ST_TEND(P.ind[8], L_alk); ST_TEND(P.ind[5], L_o2);
  }

  if (DIAG) {
    ST_KC(NO3_RESTORE, 0.0); ST_KC(SiO3_RESTORE, 0.0); ST_KC(PO4_RESTORE, 0.0);       // lrest_* never set (131-134)
    ST_KC(NITRIF, NITRIF); ST_KC(O2_PRODUCTION, O2_PRODUCTION);
    ST_KC(zoo_loss, zoo_loss);
    ST_KC(DOC_prod, DOC_prod); ST_KC(DOC_remin, DOC_remin); ST_KC(DON_prod, DON_prod); ST_KC(DON_remin, DON_remin);
    ST_KC(DOP_prod, DOP_prod); ST_KC(DOP_remin, DOP_remin); ST_KC(DOFe_prod, DOFe_prod); ST_KC(DOFe_remin, DOFe_remin);
    // local parts of the conservation integrals (1875-1938); the sweep adds the particle-dependent tendencies
    const double j_c = t_doc + t_zooC + sC + sCa, j_n = t_don + QN * t_zooC + QN * sC - sNfix, j_p = t_dop + QP_ZOO_POM * t_zooC + sQpC;
    if (COLK) { lo->oc0 = OC0; lo->d0 = D0; lo->jc = j_c; lo->jn = j_n; lo->jp = j_p; lo->jsi = sSi; }
    else { SCR(S_OC0) = OC0; SCR(S_D0) = D0; SCR(S_JC) = j_c; SCR(S_JN) = j_n; SCR(S_JP) = j_p; SCR(S_JSI) = sSi; }
    if (COLK) lo->chl100 = chl * pt100;
    else A.chl100[ix] = chl * pt100;
  }
  return true;
#undef RAW
#undef TEND
#undef SCR
#undef ST_KC
#undef ST_KCA
#undef ST_TEND
#undef Z
}

// FUSE = 1: the wavefront goes on to solve the carbonate system of its own cells (obgc_co2.h) after the
// ecosystem terms.  The two halves want different things from the SIMD -- ~150 stores per lane drained by the
// memory pipeline against ~6 600 fp64 VALU instructions -- and as separate kernels their waves compete for
// the register file instead of overlapping (the cell kernel fills a SIMD with two waves).  In one wave the
// stores drain under the Newton iterations and the co-resident wave's loads hide under them.
__device__ __forceinline__ void carbonate_of_this_lane(const BgcKernelArgs& A, const obgc_bgc_params& P) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  carbonate_cell(A, P.ind[6], P.ind[8], P.ind[0], P.ind[2], idx < ncell, k, c);
}

// the cell-parallel kernels' lane -> cell map: lanes run along the contiguous axis of the caller's layout
template <bool DIAG, bool DM>
__device__ __forceinline__ void cell_of_this_lane(const BgcKernelArgs& A, const obgc_bgc_params& P, const FusedDmArgs* F, const FusedParams* FP) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  CellIn in;
  cell_load<DIAG, DM, false>(A, P, F, k, c, in);
  (void)cell_body<DIAG, DM, false>(A, P, F, FP, nullptr, nullptr, k, c, in);
}

template <bool DIAG, int FUSE>
__global__ __launch_bounds__(kBlockCell, 2) void bgc_cell_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  cell_of_this_lane<DIAG, false>(A, P, nullptr, nullptr);
  if (FUSE) carbonate_of_this_lane(A, P);
}

static_assert(sizeof(BgcKernelArgs) + sizeof(obgc_bgc_params) + sizeof(FusedDmArgs) + sizeof(FusedParams) <= 4096,
              "kernel arguments of the fused cell kernel must fit the 4 KB kernarg segment");

// the cell kernel of obgc_fused_step: BGC + DMS + MACROS terms of a cell from one set of loads
template <bool DIAG, int FUSE>
__global__ __launch_bounds__(kBlockCell, 2) void fused_cell_kernel(const BgcKernelArgs A, const obgc_bgc_params P, const FusedDmArgs F,
                                                                   const FusedParams FP) {
  cell_of_this_lane<DIAG, true>(A, P, &F, &FP);
  if (FUSE) carbonate_of_this_lane(A, P);
}

// ------------------------------------------------------------------------------ particle sweep
template <bool DIAG>
__device__ __forceinline__ void load_level(const BgcKernelArgs& A, const obgc_bgc_params& P, long long ix, LevelIn<DIAG>& L) {
  const long long pl = A.g.plane, sp = A.splane;
#define RAWT(slot) A.tracers[(long long)((slot)-1) * pl + ix]
#define TEND(slot) A.tend[(long long)((slot)-1) * pl + ix]
#define SCR(s) A.scratch[(long long)(s) * sp + ix]
  L.dz = A.dz[ix]; L.zb = A.zbot[ix]; L.fesed = A.fesed[ix]; L.T = A.temp[ix];
  // raw values: the clamps (747-752) are applied where the level is evaluated -- a dmax() here would make the wave wait
  // for these loads right after issuing them and put a memory round trip back into every level of the chain
  L.NO3 = RAWT(P.ind[1]); L.Fe = RAWT(P.ind[4]); L.O2 = RAWT(P.ind[5]);
  L.prodPOC = SCR(S_PRODPOC); L.prodCa = SCR(S_PRODCA); L.prodSi = SCR(S_PRODSI); L.PfeLoc = SCR(S_PFELOC);
  L.L_no3 = TEND(P.ind[1]); L.L_nh4 = TEND(P.ind[3]); L.L_fe = TEND(P.ind[4]); L.L_sio3 = TEND(P.ind[2]); L.L_po4 = TEND(P.ind[0]);
  L.L_donr = TEND(P.ind[14]); L.L_dopr = TEND(P.ind[13]); L.L_dic = TEND(P.ind[6]); L.L_alk = TEND(P.ind[8]); L.L_o2 = TEND(P.ind[5]);
  if (DIAG) { L.oc0 = SCR(S_OC0); L.d0 = SCR(S_D0); L.jc = SCR(S_JC); L.jn = SCR(S_JN); L.jp = SCR(S_JP); L.jsi = SCR(S_JSI); }
#undef RAWT
#undef TEND
#undef SCR
}

// Makes the compiler retire every load of a LevelIn here.  Used once, on the first level, before the loop: otherwise
// the wait-count state of the loop header is the merge of "first level still in flight" and "steady state", and the
// conservative s_waitcnt it emits at the top of EVERY iteration waits for two of the just-issued prefetch loads and
// for all stores of the previous level -- a memory round trip per level of the sequential chain.
template <bool DIAG>
__device__ __forceinline__ void settle(const LevelIn<DIAG>& L) {
  asm volatile("" ::"v"(L.dz), "v"(L.zb), "v"(L.fesed), "v"(L.T), "v"(L.NO3), "v"(L.Fe), "v"(L.O2), "v"(L.prodPOC), "v"(L.prodCa),
               "v"(L.prodSi), "v"(L.PfeLoc));
  asm volatile("" ::"v"(L.L_no3), "v"(L.L_nh4), "v"(L.L_fe), "v"(L.L_sio3), "v"(L.L_po4), "v"(L.L_donr), "v"(L.L_dopr), "v"(L.L_dic),
               "v"(L.L_alk), "v"(L.L_o2));
  if (DIAG) asm volatile("" ::"v"(L.oc0), "v"(L.d0), "v"(L.jc), "v"(L.jn), "v"(L.jp), "v"(L.jsi));
}

// State the sweep carries from level to level of one column: the particle fluxes leaving the level above, the dust
// deficit of the ballast model and (diagnostics) the conservation integrals.
struct SweepState {
  Flux fPOC, fCa, fSi, fFe, fDust;
  double QA_dust_def;
  double J[8];   // C, C above 100 m, N, N 100 m, P, P 100 m, Si, Si 100 m
  double zbot_above;
};

// per-column initialisation (BGC_mod.F90:808-811, 2075-2104)
__device__ __forceinline__ void sweep_init(SweepState& S, double dust_flux_in, bool has_levels) {
  const double dust_in = has_levels ? dmax(0.0, dust_flux_in) : 0.0;
  S.fPOC = Flux{0.0, 0.0}; S.fCa = Flux{0.0, 0.0}; S.fSi = Flux{0.0, 0.0}; S.fFe = Flux{0.0, 0.0}; S.fDust = Flux{0.0, 0.0};
  if (dust_in != 0.0) { S.fDust.s = (1.0 - DUST_GAMMA) * dust_in; S.fDust.h = DUST_GAMMA * dust_in; }
  S.QA_dust_def = DUST_RHO * (S.fDust.s + S.fDust.h);
#pragma unroll
  for (int i = 0; i < 8; ++i) S.J[i] = 0.0;
  S.zbot_above = 0.0;
}

// What one step of the recurrence leaves for the rest of the level's arithmetic.
struct SweepCoreOut {
  double reminPOC, reminCa, reminSi, reminFe, Fe_scavenge, SED_DENITRIF, OTHER_REMIN;
  // diagnostics only
  double inPOC, inCa, inSi, inDust, inFe, reminDust, Fe_scav_rate, sedPOC, sedCa, sedSi, sedDust, sedFe;
};

// The part of a level that really is sequential down the column: iron scavenging onto the particles entering, one step
// of the ballast recurrence, the sea floor (BGC_mod.F90:1510-1529, 2242-2631).  ~100 flops.
__device__ __forceinline__ void sweep_core(const obgc_bgc_params& P, bool bottom, double dz, double zb, double fesed, double NO3, double Fe,
                                           double O2, double prodPOC, double prodCa, double prodSi, double PfeLoc, const LevelDerived& dv,
                                           SweepState& S, SweepCoreOut& out) {
  Flux &fPOC = S.fPOC, &fCa = S.fCa, &fSi = S.fSi, &fFe = S.fFe, &fDust = S.fDust;
  double& QA_dust_def = S.QA_dust_def;
  const double poc_diss = dv.poc_diss, sio2_diss = dv.sio2_diss, caco3_diss = dv.caco3_diss;
  const double decay_POC_E = dv.decay_POC_E, decay_SiO2 = dv.decay_SiO2, decay_CaCO3 = dv.decay_CaCO3, decay_dust = dv.decay_dust,
               DECAY_Hard = dv.DECAY_Hard, DECAY_HardDust = dv.DECAY_HardDust;
  struct { double PfeLoc, fesed; } cur{PfeLoc, fesed};

  // iron scavenging onto the particles LEAVING the level above (1510-1529)
  double Fe_scav_rate = P.parm_fe_scavenge_rate0;
  Fe_scav_rate = Fe_scav_rate * ((fPOC.s + fPOC.h) * 120.1 + (fCa.s + fCa.h) * CACO3_MASS + (fSi.s + fSi.h) * SIO2_MASS +
                                 (fDust.s + fDust.h) * DUST_FESCAV_SCALE);
  if (Fe > FE_SCAV_THRES1) Fe_scav_rate = Fe_scav_rate + (Fe - FE_SCAV_THRES1) * FE_MAX_SCALE2;
  const double Fe_scavenge = YPS * Fe * Fe_scav_rate;
  const double prodFe = cur.PfeLoc + Fe_scavenge;

  // one step of the Armstrong et al. ballast recurrence (2242-2501)
  const Flux inPOC = fPOC, inCa = fCa, inSi = fSi, inFe = fFe, inDust = fDust;
  double sedPOC = 0.0, sedCa = 0.0, sedSi = 0.0, sedDust = 0.0, sedFe = 0.0, SED_DENITRIF = 0.0, OTHER_REMIN = 0.0;
  const double dzr = 1.0 / dz;
  fCa.s = inCa.s * decay_CaCO3 + prodCa * ((1.0 - CACO3_GAMMA) * (1.0 - decay_CaCO3) * caco3_diss);
  fCa.h = inCa.h * DECAY_Hard + prodCa * (CACO3_GAMMA * dz);
  fSi.s = inSi.s * decay_SiO2 + prodSi * ((1.0 - SIO2_GAMMA) * (1.0 - decay_SiO2) * sio2_diss);
  fSi.h = inSi.h * DECAY_Hard + prodSi * (SIO2_GAMMA * dz);
  fDust.s = inDust.s * decay_dust;
  fDust.h = inDust.h * DECAY_HardDust;
  double POC_PROD_avail = prodPOC - CACO3_RHO * prodCa - SIO2_RHO * prodSi;
  double new_def = (QA_dust_def > 0) ? QA_dust_def * (fDust.s + fDust.h) / (inDust.s + inDust.h) : 0.0;
  if (new_def > 0.0) {
    new_def = new_def - POC_PROD_avail * dz;
    if (new_def < 0.0) { POC_PROD_avail = -new_def * dzr; new_def = 0.0; }
    else POC_PROD_avail = 0.0;
  }
  QA_dust_def = new_def;
  if (inPOC.h == 0.0 && prodPOC == 0.0) {
    fPOC.h = 0.0;
  } else {
    fPOC.h = CACO3_RHO * (fCa.s + fCa.h) + SIO2_RHO * (fSi.s + fSi.h) + DUST_RHO * (fDust.s + fDust.h) - new_def;
    fPOC.h = dmax(fPOC.h, 0.0);
  }
  fPOC.s = inPOC.s * decay_POC_E + POC_PROD_avail * ((1.0 - decay_POC_E) * poc_diss);
  double reminCa = prodCa + ((inCa.s - fCa.s) + (inCa.h - fCa.h)) * dzr;
  double reminSi = prodSi + ((inSi.s - fSi.s) + (inSi.h - fSi.h)) * dzr;
  double reminPOC = prodPOC + ((inPOC.s - fPOC.s) + (inPOC.h - fPOC.h)) * dzr;
  const double reminDust = ((inDust.s - fDust.s) + (inDust.h - fDust.h)) * dzr;
  double reminFe;
  if (inPOC.s + inPOC.h == 0.0) reminFe = (reminPOC * RED_FE_C);
  else reminFe = (reminPOC * (inFe.s + inFe.h) / (inPOC.s + inPOC.h));
  reminFe = reminFe + (inFe.s * 1.5e-5);
  fFe.s = inFe.s + dz * ((1.0 - 0.0) * prodFe - reminFe);
  if (fFe.s < 0.0) {
    fFe.s = 0.0;
    reminFe = inFe.s * dzr + (1.0 - 0.0) * prodFe;
  }
  reminFe = reminFe + reminDust * DUST_TO_FE + (cur.fesed * dzr);
  fFe.h = inFe.h;

  if (bottom) {   // sea floor: burial, sediment denitrification, instantaneous remin (2522-2631)
    double flux = fPOC.s + fPOC.h, flux_alt;
    if (flux > 0.0) {
      flux_alt = flux * 0.01 * SPD;
      sedPOC = flux * dmin(0.8, P.parm_POMbury * (0.013 + 0.53 * flux_alt * flux_alt / ((7.0 + flux_alt) * (7.0 + flux_alt))));
      SED_DENITRIF = dzr * flux * (0.06 + 0.19 * exp((O2 - NO3) * LN_0_99));   // 0.99 ** (O2 - NO3)
      if (NO3 < 5.0) SED_DENITRIF = 0.0;
      flux_alt = flux * 1.0e-6 * SPD * 365.0;
      OTHER_REMIN = dzr * dmin(dmin(0.1 + flux_alt, 0.5) * (flux - sedPOC), (flux - sedPOC - (SED_DENITRIF * dz * DENITRIF_C_N)));
      if (O2 < 1.0) OTHER_REMIN = dzr * (flux - sedPOC - (SED_DENITRIF * dz * DENITRIF_C_N));
    }
    flux = fSi.s + fSi.h;
    flux_alt = flux * 0.01 * SPD;
    sedSi = (flux_alt > 2.0) ? 0.2 : 0.04;
    sedSi = flux * P.parm_BSIbury * sedSi;
    if (zb < 3300.0e2) sedCa = fCa.s + fCa.h;
    flux = fCa.s + fCa.h;
    if (flux > 0.0) reminCa = reminCa + ((flux - sedCa) * dzr);
    flux = fSi.s + fSi.h;
    if (flux > 0.0) reminSi = reminSi + ((flux - sedSi) * dzr);
    flux = fPOC.s + fPOC.h;
    if (flux > 0.0) reminPOC = reminPOC + ((flux - sedPOC) * dzr);
    flux = (fFe.s + fFe.h);
    if (flux > 0.0) sedFe = flux;
    sedDust = fDust.s + fDust.h;
    fCa = Flux{0.0, 0.0}; fSi = Flux{0.0, 0.0}; fDust = Flux{0.0, 0.0}; fPOC = Flux{0.0, 0.0}; fFe = Flux{0.0, 0.0};
  }

  out.reminPOC = reminPOC; out.reminCa = reminCa; out.reminSi = reminSi; out.reminFe = reminFe; out.Fe_scavenge = Fe_scavenge;
  out.SED_DENITRIF = SED_DENITRIF; out.OTHER_REMIN = OTHER_REMIN;
  out.inPOC = inPOC.s + inPOC.h; out.inCa = inCa.s + inCa.h; out.inSi = inSi.s + inSi.h; out.inDust = inDust.s + inDust.h;
  out.inFe = inFe.s + inFe.h; out.reminDust = reminDust; out.Fe_scav_rate = Fe_scav_rate;
  out.sedPOC = sedPOC; out.sedCa = sedCa; out.sedSi = sedSi; out.sedDust = sedDust; out.sedFe = sedFe;
}

// The integrands of the conservation integrals of one level (BGC_mod.F90:1870-1938): the local parts from the cell terms plus
// the tendencies finished with the particle terms.
struct JTerms { double C, N, P, Si; };

// The rest of the level, cell-local once the recurrence has passed: denitrification, the particle-dependent part of 11
// tendencies and 27 diagnostics (1569-1577, 1583-1790, 2637-2694).  Stores go to the caller's arrays at index ix.
template <bool DIAG, bool FD = false>
__device__ __forceinline__ JTerms sweep_finish(const BgcKernelArgs& A, const obgc_bgc_params& P, long long ix, const LevelIn<DIAG>& cur,
                                               const SweepCoreOut& o, bool live = true) {
  const obgc_bgc_diag& D = A.diag;
  const long long pl = A.g.plane;
#define TEND(slot, ix) A.tend[(long long)((slot)-1) * pl + (ix)]
#define Z(v) (live ? (v) : 0.0)      // a cell below the sea floor (column-block kernel only): the reference's zero fill
#define ST_KC(name, v) do { if (DIAG && (FD || D.name)) OBGC_STORE(&D.name[ix], Z(v)); } while (0)
  const double dz = cur.dz, NO3 = dmax(0.0, cur.NO3), O2 = dmax(0.0, cur.O2);
  const double prodPOC = cur.prodPOC, prodCa = cur.prodCa, prodSi = cur.prodSi;
  const double reminPOC = o.reminPOC, reminCa = o.reminCa, reminSi = o.reminSi, reminFe = o.reminFe, Fe_scavenge = o.Fe_scavenge,
               SED_DENITRIF = o.SED_DENITRIF, OTHER_REMIN = o.OTHER_REMIN;
  // ---- denitrification (1569-1577) and the particle-dependent tendencies (1583-1790): the cell kernel
  //      left every local term in the tendency planes; only the remineralisation terms are added here ----
  double wd = div_by((P.parm_o2_min + P.parm_o2_min_delta) - O2, P.parm_o2_min_delta, A.recip[R_O2DELTA]);
  wd = dmin(dmax(wd, 0.0), 1.0);
  if (NO3 == 0.0) wd = 0.0;
  const double den_p = wd * (div_by(reminPOC - OTHER_REMIN, DENITRIF_C_N, A.recip[R_DENITRIF]) - SED_DENITRIF);
  const double t_no3 = (cur.L_no3 - den_p) - SED_DENITRIF;
  const double t_nh4 = cur.L_nh4 + QN * (reminPOC * (1.0 - DONREFRACT));
  const double t_fe = (reminFe + cur.L_fe) - Fe_scavenge;
  const double t_sio3 = reminSi + cur.L_sio3;
  const double t_po4 = cur.L_po4 + QP_ZOO_POM * ((1.0 - DOPREFRACT) * reminPOC);
  const double t_donr = cur.L_donr + (reminPOC * DONREFRACT * QN);
  const double t_dopr = cur.L_dopr + (reminPOC * DOPREFRACT * QP_ZOO_POM);
  const double t_dic = (cur.L_dic + reminPOC) + reminCa;
  const double t_alk = (-t_no3 + t_nh4 + 2.0 * reminCa) + cur.L_alk;
  double wo = div_by(O2 - P.parm_o2_min, P.parm_o2_min_delta, A.recip[R_O2DELTA]);
  wo = dmin(dmax(wo, 0.0), 1.0);
  const double oc_p = wo * div_by(reminPOC - (SED_DENITRIF * DENITRIF_C_N) - OTHER_REMIN, REMIN_D_C_O2, A.recip[R_REMIN]);
  OBGC_STORE(&TEND(P.ind[1], ix), Z(t_no3));
  OBGC_STORE(&TEND(P.ind[3], ix), Z(t_nh4));
  OBGC_STORE(&TEND(P.ind[4], ix), Z(t_fe));
  OBGC_STORE(&TEND(P.ind[2], ix), Z(t_sio3));
  OBGC_STORE(&TEND(P.ind[0], ix), Z(t_po4));
  OBGC_STORE(&TEND(P.ind[14], ix), Z(t_donr));
  OBGC_STORE(&TEND(P.ind[13], ix), Z(t_dopr));
  OBGC_STORE(&TEND(P.ind[6], ix), Z(t_dic));
  OBGC_STORE(&TEND(P.ind[7], ix), Z(A.alt_co2_use_eco ? t_dic : 0.0));
  OBGC_STORE(&TEND(P.ind[8], ix), Z(t_alk));
  OBGC_STORE(&TEND(P.ind[5], ix), Z(cur.L_o2 - oc_p));

  JTerms jt{0.0, 0.0, 0.0, 0.0};
  if (DIAG) {   // 2637-2694, 1563-1577, 1866-1868
    ST_KC(POC_FLUX_IN, o.inPOC); ST_KC(POC_PROD, prodPOC); ST_KC(POC_REMIN, reminPOC);
    ST_KC(CaCO3_FLUX_IN, o.inCa); ST_KC(CaCO3_PROD, prodCa); ST_KC(CaCO3_REMIN, reminCa);
    ST_KC(SiO2_FLUX_IN, o.inSi); ST_KC(SiO2_PROD, prodSi); ST_KC(SiO2_REMIN, reminSi);
    ST_KC(dust_FLUX_IN, o.inDust); ST_KC(dust_REMIN, o.reminDust);
    ST_KC(P_iron_FLUX_IN, o.inFe); ST_KC(P_iron_PROD, cur.PfeLoc + Fe_scavenge); ST_KC(P_iron_REMIN, reminFe);
    ST_KC(calcToSed, o.sedCa); ST_KC(bsiToSed, o.sedSi); ST_KC(pocToSed, o.sedPOC);
    ST_KC(SedDenitrif, SED_DENITRIF * dz); ST_KC(OtherRemin, OTHER_REMIN * dz);
    ST_KC(ponToSed, (o.sedPOC * QN)); ST_KC(popToSed, (o.sedPOC * QP_ZOO_POM));
    ST_KC(dustToSed, o.sedDust); ST_KC(pfeToSed, o.sedFe);
    const double DENITRIF = cur.d0 + den_p;
    ST_KC(DENITRIF, DENITRIF); ST_KC(O2_CONSUMPTION, cur.oc0 + oc_p);
    ST_KC(Fe_scavenge, Fe_scavenge); ST_KC(Fe_scavenge_rate, o.Fe_scav_rate);
    jt.C = t_dic + cur.jc;
    double w1 = t_no3 + t_nh4 + t_donr + cur.jn;
    jt.N = w1 + DENITRIF + SED_DENITRIF;
    jt.P = t_po4 + t_dopr + cur.jp;
    jt.Si = t_sio3 + cur.jsi;
  }
  return jt;
#undef TEND
#undef ST_KC
#undef Z
}

// conservation integrals (1870-1938), top to bottom; J[0..7] = C, C 100 m, N, N 100 m, P, P 100 m, Si, Si 100 m
__device__ __forceinline__ void accumulate_J(double (&J)[8], const JTerms& jt, double dz, double zb, double zbot_above, double sedPOC,
                                             double sedCa, double sedSi) {
  const double w2 = dmin(100.0e2 - zbot_above, dz);
  const double pt100 = (w2 > 0.0) ? w2 : 0.0;
  const bool shallow = zb <= 100.0e2;
  J[0] = J[0] + jt.C * dz + sedPOC + sedCa;
  J[1] = J[1] + jt.C * pt100 + (shallow ? sedPOC + sedCa : 0.0);
  J[2] = J[2] + jt.N * dz + sedPOC * QN;
  J[3] = J[3] + jt.N * pt100 + (shallow ? sedPOC * QN : 0.0);
  J[4] = J[4] + jt.P * dz + sedPOC * QP_ZOO_POM;
  J[5] = J[5] + jt.P * pt100 + (shallow ? sedPOC * QP_ZOO_POM : 0.0);
  J[6] = J[6] + jt.Si * dz + sedSi;
  J[7] = J[7] + jt.Si * pt100 + (shallow ? sedSi : 0.0);
}

// LANES = active lanes per wavefront (experiment knob, see launch_bgc_ecosys_sweep; 64 is the default).
template <bool DIAG, int LANES>
__global__ __launch_bounds__(kBlockCol) void bgc_sweep_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  if ((int)threadIdx.x >= LANES) return;
  const int c = blockIdx.x * LANES + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;
#define ST_C(name, v) do { if (DIAG && D.name) D.name[c] = (v); } while (0)
  SweepState S;
  sweep_init(S, A.dust[c], kmax >= 1);

  LevelIn<DIAG> cur, nxt;
  if (kmax >= 1) { load_level<DIAG>(A, P, g.at(0, c), cur); settle<DIAG>(cur); }
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    if (k + 1 < kmax) load_level<DIAG>(A, P, g.at(k + 1, c), nxt);   // in flight while this level is evaluated
    // length scales and decay factors are functions of inputs this kernel reads anyway, so they are evaluated here
    // instead of travelling through HBM
    const double TfuncS = exp(div_by((cur.T + P.T0_Kelvin_BGC) - (TREF + P.T0_Kelvin_BGC), 10.0, A.recip[R_TEN]) * LN_Q_10);   // 1.5 ** (...)
    const LevelDerived dv = derive_level(P, A.recip, cur.zb, TfuncS, dmax(0.0, cur.O2), cur.dz);
    SweepCoreOut o;
    sweep_core(P, k == kmax - 1, cur.dz, cur.zb, cur.fesed, dmax(0.0, cur.NO3), dmax(0.0, cur.Fe), dmax(0.0, cur.O2), cur.prodPOC,
               cur.prodCa, cur.prodSi, cur.PfeLoc, dv, S, o);
    const JTerms jt = sweep_finish<DIAG>(A, P, ix, cur, o);
    if (DIAG) accumulate_J(S.J, jt, cur.dz, cur.zb, S.zbot_above, o.sedPOC, o.sedCa, o.sedSi);
    S.zbot_above = cur.zb;
    cur = nxt;
  }   // k

  if (DIAG) {
    ST_C(Jint_Ctot, S.J[0]); ST_C(Jint_100m_Ctot, S.J[1]); ST_C(Jint_Ntot, S.J[2]); ST_C(Jint_100m_Ntot, S.J[3]);
    ST_C(Jint_Ptot, S.J[4]); ST_C(Jint_100m_Ptot, S.J[5]); ST_C(Jint_Sitot, S.J[6]); ST_C(Jint_100m_Sitot, S.J[7]);
  }
#undef ST_C
}

// ---- the sweep of a small batch as three kernels ----
// A few thousand columns are a few dozen sweep wavefronts, each alone on its SIMD, walking its levels through ~800 dependent
// instructions apiece (six exp for the decay factors, the recurrence, eleven tendencies, 27 diagnostics): 2.6 us per level,
// 60 % of the step at 2 048 columns x 60.  Only sweep_core is carried from level to level.  So: the decay factors with a lane per
// cell (bgc_derive_kernel), the recurrence with a lane per column and nothing else in the chain (bgc_walk_kernel), the rest
// of the level with a lane per cell again (bgc_finish_kernel), and the conservation integrals as a column sum of what that
// left (bgc_jint_kernel).  Same functions on the same values as bgc_sweep_kernel, hence the same bits.  The planes in
// between (SplitSlot) cost traffic a large batch would feel; small ones keep them in the caches.
#define WPL(slot) A.split[(long long)(slot) * A.splane + ix]

__global__ __launch_bounds__(256) void bgc_derive_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.nlev * g.ncol) return;
  const int c = (int)(idx % g.ncol), k = (int)(idx / g.ncol);
  if (k >= A.kmax[c]) return;
  const long long ix = g.at(k, c);
  const double T = A.temp[ix], O2 = A.tracers[(long long)(P.ind[5] - 1) * g.plane + ix];
  const double TfuncS = exp(div_by((T + P.T0_Kelvin_BGC) - (TREF + P.T0_Kelvin_BGC), 10.0, A.recip[R_TEN]) * LN_Q_10);
  const LevelDerived dv = derive_level(P, A.recip, A.zbot[ix], TfuncS, dmax(0.0, O2), A.dz[ix]);
  WPL(W_POC_DISS) = dv.poc_diss; WPL(W_SIO2_DISS) = dv.sio2_diss; WPL(W_CACO3_DISS) = dv.caco3_diss;
  WPL(W_DECAY_POC_E) = dv.decay_POC_E; WPL(W_DECAY_SIO2) = dv.decay_SiO2; WPL(W_DECAY_CACO3) = dv.decay_CaCO3;
  WPL(W_DECAY_DUST) = dv.decay_dust; WPL(W_DECAY_HARD) = dv.DECAY_Hard; WPL(W_DECAY_HARDDUST) = dv.DECAY_HardDust;
}

// what one level of the walk reads: nothing of it depends on the recurrence, so the next level's set is in flight while this
// one is evaluated
struct WalkIn {
  double dz, zb, fesed, NO3, Fe, O2, prodPOC, prodCa, prodSi, PfeLoc;
  LevelDerived dv;
};
__device__ __forceinline__ void walk_load(const BgcKernelArgs& A, const obgc_bgc_params& P, long long ix, WalkIn& L) {
  const long long pl = A.g.plane, sp = A.splane;
  L.dz = A.dz[ix]; L.zb = A.zbot[ix]; L.fesed = A.fesed[ix];
  L.NO3 = A.tracers[(long long)(P.ind[1] - 1) * pl + ix]; L.Fe = A.tracers[(long long)(P.ind[4] - 1) * pl + ix];
  L.O2 = A.tracers[(long long)(P.ind[5] - 1) * pl + ix];
  L.prodPOC = A.scratch[(long long)S_PRODPOC * sp + ix]; L.prodCa = A.scratch[(long long)S_PRODCA * sp + ix];
  L.prodSi = A.scratch[(long long)S_PRODSI * sp + ix]; L.PfeLoc = A.scratch[(long long)S_PFELOC * sp + ix];
  L.dv.poc_diss = WPL(W_POC_DISS); L.dv.sio2_diss = WPL(W_SIO2_DISS); L.dv.caco3_diss = WPL(W_CACO3_DISS);
  L.dv.decay_POC_E = WPL(W_DECAY_POC_E); L.dv.decay_SiO2 = WPL(W_DECAY_SIO2); L.dv.decay_CaCO3 = WPL(W_DECAY_CACO3);
  L.dv.decay_dust = WPL(W_DECAY_DUST); L.dv.DECAY_Hard = WPL(W_DECAY_HARD); L.dv.DECAY_HardDust = WPL(W_DECAY_HARDDUST);
}
// (see settle() above: retires the first level's loads before the loop so that the loop header's wait counts are the steady state's)
__device__ __forceinline__ void walk_settle(const WalkIn& L) {
  asm volatile("" ::"v"(L.dz), "v"(L.zb), "v"(L.fesed), "v"(L.NO3), "v"(L.Fe), "v"(L.O2), "v"(L.prodPOC), "v"(L.prodCa), "v"(L.prodSi),
               "v"(L.PfeLoc));
  asm volatile("" ::"v"(L.dv.poc_diss), "v"(L.dv.sio2_diss), "v"(L.dv.caco3_diss), "v"(L.dv.decay_POC_E), "v"(L.dv.decay_SiO2),
               "v"(L.dv.decay_CaCO3), "v"(L.dv.decay_dust), "v"(L.dv.DECAY_Hard), "v"(L.dv.DECAY_HardDust));
}

template <bool DIAG>
__global__ __launch_bounds__(kBlockCol) void bgc_walk_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= g.ncol) return;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;
  SweepState S;
  sweep_init(S, A.dust[c], kmax >= 1);
  // one level ahead, as in bgc_sweep_kernel (a ring of three levels in flight measured slower: 63 against 52 us at 2 048 x 60)
  WalkIn cur, nxt;
  if (kmax >= 1) { walk_load(A, P, g.at(0, c), cur); walk_settle(cur); }
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    if (k + 1 < kmax) walk_load(A, P, g.at(k + 1, c), nxt);
    SweepCoreOut o;
    sweep_core(P, k == kmax - 1, cur.dz, cur.zb, cur.fesed, dmax(0.0, cur.NO3), dmax(0.0, cur.Fe), dmax(0.0, cur.O2), cur.prodPOC,
               cur.prodCa, cur.prodSi, cur.PfeLoc, cur.dv, S, o);
    WPL(W_REMIN_POC) = o.reminPOC; WPL(W_REMIN_CA) = o.reminCa; WPL(W_REMIN_SI) = o.reminSi; WPL(W_REMIN_FE) = o.reminFe;
    WPL(W_FE_SCAVENGE) = o.Fe_scavenge; WPL(W_SED_DENITRIF) = o.SED_DENITRIF; WPL(W_OTHER_REMIN) = o.OTHER_REMIN;
    if (DIAG) {
      WPL(W_IN_POC) = o.inPOC; WPL(W_IN_CA) = o.inCa; WPL(W_IN_SI) = o.inSi; WPL(W_IN_DUST) = o.inDust; WPL(W_IN_FE) = o.inFe;
      WPL(W_REMIN_DUST) = o.reminDust; WPL(W_FE_SCAV_RATE) = o.Fe_scav_rate;
      WPL(W_SED_POC) = o.sedPOC; WPL(W_SED_CA) = o.sedCa; WPL(W_SED_SI) = o.sedSi; WPL(W_SED_DUST) = o.sedDust; WPL(W_SED_FE) = o.sedFe;
    }
    cur = nxt;
  }
}

// the rest of the level (sweep_finish) for every active cell; with diagnostics the integrands of the conservation integrals
// replace their local parts in the hand-off planes (same cell, read before written)
template <bool DIAG>
__global__ __launch_bounds__(256) void bgc_finish_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)g.nlev * g.ncol) return;
  const int c = (int)(idx % g.ncol), k = (int)(idx / g.ncol);
  if (k >= A.kmax[c]) return;
  const long long ix = g.at(k, c);
  LevelIn<DIAG> cur;
  load_level<DIAG>(A, P, ix, cur);
  SweepCoreOut o;
  o.reminPOC = WPL(W_REMIN_POC); o.reminCa = WPL(W_REMIN_CA); o.reminSi = WPL(W_REMIN_SI); o.reminFe = WPL(W_REMIN_FE);
  o.Fe_scavenge = WPL(W_FE_SCAVENGE); o.SED_DENITRIF = WPL(W_SED_DENITRIF); o.OTHER_REMIN = WPL(W_OTHER_REMIN);
  if (DIAG) {
    o.inPOC = WPL(W_IN_POC); o.inCa = WPL(W_IN_CA); o.inSi = WPL(W_IN_SI); o.inDust = WPL(W_IN_DUST); o.inFe = WPL(W_IN_FE);
    o.reminDust = WPL(W_REMIN_DUST); o.Fe_scav_rate = WPL(W_FE_SCAV_RATE);
    o.sedPOC = WPL(W_SED_POC); o.sedCa = WPL(W_SED_CA); o.sedSi = WPL(W_SED_SI); o.sedDust = WPL(W_SED_DUST); o.sedFe = WPL(W_SED_FE);
  }
  const JTerms jt = sweep_finish<DIAG>(A, P, ix, cur, o);
  if (DIAG) {
    const long long sp = A.splane;
    A.scratch[(long long)S_JC * sp + ix] = jt.C; A.scratch[(long long)S_JN * sp + ix] = jt.N;
    A.scratch[(long long)S_JP * sp + ix] = jt.P; A.scratch[(long long)S_JSI * sp + ix] = jt.Si;
  }
}

// conservation integrals of the columns from the integrands bgc_finish_kernel left, top to bottom (accumulate_J)
__global__ __launch_bounds__(kBlockCol) void bgc_jint_kernel(const BgcKernelArgs A) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;
  const long long sp = A.splane;
  double J[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double zbot_above = 0.0;
  // (eight levels of loads fetched ahead of their sums measured slower than this: 47 against 29 us at 2 048 x 60)
#pragma unroll 4
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    const JTerms jt{A.scratch[(long long)S_JC * sp + ix], A.scratch[(long long)S_JN * sp + ix], A.scratch[(long long)S_JP * sp + ix],
                    A.scratch[(long long)S_JSI * sp + ix]};
    const double zb = A.zbot[ix];
    accumulate_J(J, jt, A.dz[ix], zb, zbot_above, WPL(W_SED_POC), WPL(W_SED_CA), WPL(W_SED_SI));
    zbot_above = zb;
  }
#define ST_C(name, v) do { if (D.name) D.name[c] = (v); } while (0)
  ST_C(Jint_Ctot, J[0]); ST_C(Jint_100m_Ctot, J[1]); ST_C(Jint_Ntot, J[2]); ST_C(Jint_100m_Ntot, J[3]);
  ST_C(Jint_Ptot, J[4]); ST_C(Jint_100m_Ptot, J[5]); ST_C(Jint_Sitot, J[6]); ST_C(Jint_100m_Sitot, J[7]);
#undef ST_C
}
#undef WPL

// the recurrence state of a column (five fluxes, the dust deficit) to / from its LDS row
constexpr int kStateVals = 11;
__device__ __forceinline__ void state_store(const SweepState& S, double (*st)[64], int lane) {
  st[0][lane] = S.fPOC.s; st[1][lane] = S.fPOC.h; st[2][lane] = S.fCa.s; st[3][lane] = S.fCa.h; st[4][lane] = S.fSi.s;
  st[5][lane] = S.fSi.h; st[6][lane] = S.fFe.s; st[7][lane] = S.fFe.h; st[8][lane] = S.fDust.s; st[9][lane] = S.fDust.h;
  st[10][lane] = S.QA_dust_def;
}
__device__ __forceinline__ void state_load(SweepState& S, double (*st)[64], int lane) {
  S.fPOC.s = st[0][lane]; S.fPOC.h = st[1][lane]; S.fCa.s = st[2][lane]; S.fCa.h = st[3][lane]; S.fSi.s = st[4][lane];
  S.fSi.h = st[5][lane]; S.fFe.s = st[6][lane]; S.fFe.h = st[7][lane]; S.fDust.s = st[8][lane]; S.fDust.h = st[9][lane];
  S.QA_dust_def = st[10][lane];
  S.zbot_above = 0.0;
}

// ---- column integrals inside the column-block kernel (diagnostics only) ----
// The conservation integrals (BGC_mod.F90:1870-1938), the O2 minimum (1954-1968), chlorophyll over the top 100 m
// (1940-1945) and -- when the carbonate system is solved inside the kernel -- the saturation horizons (1003-1032) are
// top-to-bottom sums and searches over quantities that are final only when a cell's finishing arithmetic is.  Once a lane
// has read its own entries of the hand-off slots in the finish phase, those entries are dead until the next chunk's cell
// terms: the lane leaves its integrands there, and after ONE more LDS barrier per chunk four wavefronts each take one
// kind of integral for the 64 columns over the chunk's levels, in the reference's order (same bits as the column-integral
// kernel).  Running values rest in LDS between chunks.  The barrier stands behind the next chunk's loads, which are in
// flight while the integrals advance.
enum IntegrandSlot {   // own entries of the hand-off slots; the burial fluxes are read where the walk left them (H_OUT_SED*)
  I_SATC = 0, I_SATA = 1,
  I_JC = H_NIN, I_JN, I_JP, I_JSI, I_DZ, I_ZB, I_O2, I_ZMID, I_CHL, I_CO3,
  I_END
};
static_assert((int)I_END <= (int)H_NSLOTS && (int)I_SATA < (int)H_OUT_SEDPOC, "integrands must not touch the burial fluxes of the walk");
enum IntegralState {
  N_J0, N_J1, N_J2, N_J3, N_J4, N_J5, N_J6, N_J7,   // C, C 100 m, N, N 100 m, P, P 100 m, Si, Si 100 m
  N_ZABOVE0, N_ZABOVE1,                              // bottom depth of the level above, one copy per integrating wavefront
  N_ZC, N_ZA, N_ANOMC, N_ANOMA, N_ZPREV,             // saturation horizons: results, anomalies and depth of the level above
  N_O2MIN, N_O2Z, N_CHL,
  N_INTEGRALS
};

// one kind of integral (role = wavefront index) for column `lane` over the levels of chunk j
template <int FUSE>
__device__ __forceinline__ void integrate_chunk(double (*ho)[kColThreads], double (*st)[64], int role, int lane, int kmax, int j) {
  const int k0 = j * kColLevels;
  if (k0 >= kmax) return;
  if (role < 2) {
    // role 0: carbon and nitrogen, role 1: phosphorus and silicon (accumulate_J: the other half is dead code)
    double J[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (role == 0) { J[0] = st[N_J0][lane]; J[1] = st[N_J1][lane]; J[2] = st[N_J2][lane]; J[3] = st[N_J3][lane]; }
    else { J[4] = st[N_J4][lane]; J[5] = st[N_J5][lane]; J[6] = st[N_J6][lane]; J[7] = st[N_J7][lane]; }
    double zabove = role == 0 ? st[N_ZABOVE0][lane] : st[N_ZABOVE1][lane];
#pragma unroll 1
    for (int l = 0; l < kColLevels; ++l) {
      if (k0 + l >= kmax) break;
      const int t = l * 64 + lane;
      JTerms jt{0.0, 0.0, 0.0, 0.0};
      double sedCa = 0.0, sedSi = 0.0;
      if (role == 0) { jt.C = ho[I_JC][t]; jt.N = ho[I_JN][t]; sedCa = ho[H_OUT_SEDCA][t]; }
      else { jt.P = ho[I_JP][t]; jt.Si = ho[I_JSI][t]; sedSi = ho[H_OUT_SEDSI][t]; }
      const double zb = ho[I_ZB][t];
      accumulate_J(J, jt, ho[I_DZ][t], zb, zabove, ho[H_OUT_SEDPOC][t], sedCa, sedSi);
      zabove = zb;
    }
    if (role == 0) { st[N_J0][lane] = J[0]; st[N_J1][lane] = J[1]; st[N_J2][lane] = J[2]; st[N_J3][lane] = J[3]; st[N_ZABOVE0][lane] = zabove; }
    else { st[N_J4][lane] = J[4]; st[N_J5][lane] = J[5]; st[N_J6][lane] = J[6]; st[N_J7][lane] = J[7]; st[N_ZABOVE1][lane] = zabove; }
  } else if (role == 2) {
    if (FUSE) {   // saturation horizons: first crossing down the column (BGC_mod.F90:1003-1032)
      double zc = st[N_ZC][lane], za = st[N_ZA][lane], anom_c = st[N_ANOMC][lane], anom_a = st[N_ANOMA][lane], zprev = st[N_ZPREV][lane];
#pragma unroll 1
      for (int l = 0; l < kColLevels; ++l) {
        const int k = k0 + l;
        if (k >= kmax) break;
        const int t = l * 64 + lane;
        const double CO3 = ho[I_CO3][t], sc = ho[I_SATC][t], sa = ho[I_SATA][t], zmid = ho[I_ZMID][t];
        if (k == 0) {
          zc = (CO3 > sc) ? -1.0 : 0.0;
          za = (CO3 > sa) ? -1.0 : 0.0;
        } else {
          const double w4 = zprev + (zmid - zprev);                       // BGC_mod.F90:1009
          if (zc == -1.0 && CO3 <= sc) zc = w4 * anom_c / (anom_c - (CO3 - sc));
          if (za == -1.0 && CO3 <= sa) za = w4 * anom_a / (anom_a - (CO3 - sa));
          if (k == kmax - 1) {
            const double zb = ho[I_ZB][t];
            if (zc == -1.0) zc = zb;
            if (za == -1.0) za = zb;
          }
        }
        anom_c = CO3 - sc;
        anom_a = CO3 - sa;
        zprev = zmid;
      }
      st[N_ZC][lane] = zc; st[N_ZA][lane] = za; st[N_ANOMC][lane] = anom_c; st[N_ANOMA][lane] = anom_a; st[N_ZPREV][lane] = zprev;
    }
  } else {
    double o2min = st[N_O2MIN][lane], o2min_z = st[N_O2Z][lane], chl = st[N_CHL][lane];
#pragma unroll 1
    for (int l = 0; l < kColLevels; ++l) {
      const int k = k0 + l;
      if (k >= kmax) break;
      const int t = l * 64 + lane;
      const double O2 = ho[I_O2][t];
      if (k == 0 || O2 < o2min) { o2min = O2; o2min_z = ho[I_ZMID][t]; }   // 1954-1968
      chl = chl + ho[I_CHL][t];                                             // 1940-1945
    }
    st[N_O2MIN][lane] = o2min; st[N_O2Z][lane] = o2min_z; st[N_CHL][lane] = chl;
  }
}

// ------------------------------------------------------------------------- column-block kernel
// The whole ecosystem part of BGC_SourceSink (and, for obgc_fused_step, DMS_SourceSink and MACROS_SourceSink) in ONE
// launch for callers in the native layout.  A workgroup owns 64 adjacent columns and marches down them in chunks of
// kColLevels levels; wavefront w evaluates the cell terms of level 4 j + w (lane = column, every load and store a
// contiguous 512-byte row).  What the separate kernels pass through HBM stays on the chip:
//   * light: each wavefront computes its level's attenuation; the product down the chunk runs through 2 KB of LDS and
//     the PAR entering the next chunk is carried in LDS (no attenuation / scan kernels, no light planes);
//   * particles: the cell terms leave the inputs of the recurrence (production terms, length scales, decay factors: 19
//     values) in LDS; one wavefront runs the ~100 sequential flops per level of the ballast recurrence over the chunk's
//     levels (state of the 64 columns in LDS between chunks) and leaves its results in the same slots; every lane then
//     finishes its own cell: the particle-dependent part of 11 tendencies and 27 diagnostics (no hand-off planes, no
//     tendency written twice, no second read of dz, zbot, T, O2, NO3, Fe by a sweep kernel).
// Three barriers per chunk: attenuations in LDS, recurrence inputs in LDS, recurrence results in LDS (a fourth with
// diagnostics, for the column integrals: integrate_chunk above).  Same statements in the same order as the separate
// kernels, hence the same bits (tests/test_gpu_column_kernel.py).  Only the vertical integrals of the production
// profiles are left to the column-integral kernel: their ~12 integrands per cell are known before the walk and there is
// neither LDS nor register room to keep them until a wavefront is free to sum them.
// FUSE = 1: every lane goes on to solve the carbonate system of its cell (obgc_co2.h), as bgc_cell_kernel<.., 1> does.
#ifndef OBGC_WALK_PRIO
#define OBGC_WALK_PRIO 1
#endif
#ifdef OBGC_COLTIMING
// Development build only (make BUILD=build_timing CXXEXTRA=-DOBGC_COLTIMING): shader-clock cycles per phase of the chunk
// loop, summed over the wavefronts of a launch by wavefront index; printed by launch_bgc_column.
__device__ unsigned long long g_coltiming[8][kColLevels];
#define OBGC_T(i) do { __builtin_amdgcn_s_waitcnt(0xc07f | 0); tnow = __builtin_readcyclecounter(); tacc[i] += tnow - tlast; tlast = tnow; } while (0)
#else
#define OBGC_T(i) do { } while (0)
#endif

template <bool DIAG, bool DM, int FUSE, bool FD = false>
__global__ __launch_bounds__(kColThreads, 2) void bgc_column_kernel(const BgcKernelArgs A, const obgc_bgc_params P, const FusedDmArgs F,
                                                                    const FusedParams FP) {
  __shared__ double s_ho[H_NSLOTS][kColThreads];
  __shared__ double s_att[kColLevels][64], s_par[2][64];
  __shared__ double s_attd[DM ? kColLevels : 1][64], s_pard[DM ? 2 : 1][64];
  // The recurrence state of the 64 columns rests in LDS between chunks: whichever wavefront walks a chunk holds it in
  // registers only while it does, so the cell terms do not share the register file with it.
  __shared__ double s_state[kStateVals][64];
  __shared__ double s_int[DIAG ? (int)N_INTEGRALS : 1][64];   // running column integrals (integrate_chunk)
  const GridView& g = A.g;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const bool colvalid = c < g.ncol;
  const int cc = colvalid ? c : g.ncol - 1;
  int kmax = A.kmax[cc];
  if (kmax > g.nlev) kmax = g.nlev;
  if (!colvalid) kmax = 0;
  if (w == 0) {   // read behind the first barrier
    SweepState S;
    sweep_init(S, A.dust[cc], kmax >= 1);
    state_store(S, s_state, lane);
    const double sw = dmax(0.0, A.sw[cc]);   // PAR entering the first level (BGC_mod.F90:813-814; DMS_mod.F90:506-507)
    s_par[0][lane] = sw * F_QSW_PAR;
    if (DM) s_pard[0][lane] = sw * F.f_qsw_par_dms;
  }
  if (DIAG) {   // first read behind the integrals' barrier of the second chunk
    for (int i = w; i < (int)N_INTEGRALS; i += kColLevels) s_int[i][lane] = 0.0;
  }
  const int role = __builtin_amdgcn_readfirstlane(w);   // which integrals this wavefront advances
  const int nchunk = (g.nlev + kColLevels - 1) / kColLevels;
#ifdef OBGC_COLTIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tnow, tlast = __builtin_readcyclecounter();
#endif
  for (int j = 0; j < nchunk; ++j) {
    ColCtx X;
    X.k = j * kColLevels + w; X.c = c; X.w = w; X.lane = lane;
    X.valid = colvalid && X.k < g.nlev;
    X.parity = j & 1;
    X.ho = &s_ho[0][0]; X.att = &s_att[0][0]; X.par = &s_par[0][0]; X.attd = &s_attd[0][0]; X.pard = &s_pard[0][0];
    // The kernel arguments are read through pointers the optimiser must take as new in every chunk: otherwise it hoists
    // the ~250 scalar loads of array pointers and parameters out of the loop and then spills them (1 600 SGPRs into 26
    // VGPRs) -- inside one chunk they are loaded where they are used, as in the kernels without a loop.
    // (An offset of zero the optimiser cannot see through keeps the pointers in the kernel-argument address space.)
    long long zero = 0;
    asm volatile("" : "+s"(zero));
    const BgcKernelArgs* Ap = (const BgcKernelArgs*)((const char*)&A + zero);
    const obgc_bgc_params* Pp = (const obgc_bgc_params*)((const char*)&P + zero);
    const FusedDmArgs* Fp = (const FusedDmArgs*)((const char*)&F + zero);
    const FusedParams* FPp = (const FusedParams*)((const char*)&FP + zero);
    LevelIn<DIAG> lo;
    const int kc = X.k < g.nlev ? X.k : g.nlev - 1;      // invalid lanes work on clamped addresses and store nothing
    // (Requesting the NEXT chunk's inputs ahead of this chunk's last stores -- vector memory operations complete in issue
    // order, so these loads wait for every store before them -- was measured and lost: the ~50 values live across a chunk
    // boundary cost more in spills than the wait, 44.9 ms against 41.3 ms at 1 M columns x 80 levels.)
    CellIn in;
    cell_load<DIAG, DM, true, FUSE != 0>(*Ap, *Pp, DM ? Fp : nullptr, kc, cc, in);
    if (DIAG && j > 0) {   // the chunk above has left its integrands in the slots; this chunk's loads are in flight
      lds_barrier();
      integrate_chunk<FUSE>(s_ho, s_int, role, lane, kmax, j - 1);
    }
#ifdef OBGC_COLTIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OBGC_T(0);   // the chunk's loads (behind the stores of the chunk before)
#endif
    const bool active = cell_body<DIAG, DM, true, FD>(*Ap, *Pp, DM ? Fp : nullptr, DM ? FPp : nullptr, &X, &lo, kc, cc, in);
    OBGC_T(1);   // cell terms (with the light barrier inside)
    lds_barrier();   // the chunk's recurrence inputs are in LDS
    OBGC_T(2);
    // One wavefront walks the chunk's levels for the 64 columns -- a different one in every chunk, so that the four SIMDs
    // share the walk -- reading a cell's inputs from LDS and leaving its results in the same slots.
    if (w == (j & (kColLevels - 1)) && j * kColLevels < kmax) {
#if OBGC_WALK_PRIO
      __builtin_amdgcn_s_setprio(3);   // three wavefronts of this workgroup wait for this one
#endif
      SweepState S;
      state_load(S, s_state, lane);
#pragma unroll 1
      for (int l = 0; l < kColLevels; ++l) {
        const int k = j * kColLevels + l;
        if (k < kmax) {
          const int t = l * 64 + lane;
          LevelDerived dv;
          dv.poc_diss = s_ho[H_POC_DISS][t]; dv.sio2_diss = s_ho[H_SIO2_DISS][t]; dv.caco3_diss = s_ho[H_CACO3_DISS][t];
          dv.decay_POC_E = s_ho[H_DEC_POC][t]; dv.decay_SiO2 = s_ho[H_DEC_SI][t]; dv.decay_CaCO3 = s_ho[H_DEC_CA][t];
          dv.decay_dust = s_ho[H_DEC_DUST][t]; dv.DECAY_Hard = s_ho[H_DEC_HARD][t]; dv.DECAY_HardDust = s_ho[H_DEC_HARDDUST][t];
          SweepCoreOut o;
          sweep_core(*Pp, k == kmax - 1, s_ho[H_DZ][t], s_ho[H_ZB][t], s_ho[H_FESED][t], s_ho[H_NO3][t], s_ho[H_FE][t], s_ho[H_O2][t],
                     s_ho[H_PRODPOC][t], s_ho[H_PRODCA][t], s_ho[H_PRODSI][t], s_ho[H_PFELOC][t], dv, S, o);
          s_ho[H_OUT_REMINPOC][t] = o.reminPOC; s_ho[H_OUT_REMINCA][t] = o.reminCa; s_ho[H_OUT_REMINSI][t] = o.reminSi;
          s_ho[H_OUT_REMINFE][t] = o.reminFe; s_ho[H_OUT_FESCAV][t] = o.Fe_scavenge; s_ho[H_OUT_SEDDEN][t] = o.SED_DENITRIF;
          s_ho[H_OUT_OTHER][t] = o.OTHER_REMIN;
          if (DIAG) {
            s_ho[H_OUT_INPOC][t] = o.inPOC; s_ho[H_OUT_INCA][t] = o.inCa; s_ho[H_OUT_INSI][t] = o.inSi; s_ho[H_OUT_INDUST][t] = o.inDust;
            s_ho[H_OUT_INFE][t] = o.inFe; s_ho[H_OUT_REMINDUST][t] = o.reminDust; s_ho[H_OUT_FESCAVRATE][t] = o.Fe_scav_rate;
            s_ho[H_OUT_SEDPOC][t] = o.sedPOC; s_ho[H_OUT_SEDCA][t] = o.sedCa; s_ho[H_OUT_SEDSI][t] = o.sedSi;
            s_ho[H_OUT_SEDDUST][t] = o.sedDust; s_ho[H_OUT_SEDFE][t] = o.sedFe;
          }
        }
      }
      state_store(S, s_state, lane);
#if OBGC_WALK_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    OBGC_T(3);   // the walk (one wavefront per chunk)
    lds_barrier();   // the recurrence has passed the chunk
    OBGC_T(4);
    // The carbonate block's nine inputs are requested HERE, ahead of the stores of the finish phase: vector memory
    // operations complete in issue order, so requested behind them they would wait for every one of them.
    CarbIn cin;
    if (FUSE) {
      carbonate_load<true>(*Ap, Pp->ind[6], Pp->ind[8], Pp->ind[0], Pp->ind[2], X.valid, X.k, X.valid ? c : 0, cin);
      if (cin.live) { cin.T = lo.T; cin.S = lo.S; cin.zmid = lo.zmid; cin.po4 = lo.PO4c; cin.sio3 = lo.SiO3c; }
    }
    double zmid_i = 0.0;   // depth of the cell centre, for the O2 minimum and the saturation horizons
    if (DIAG && !FUSE && active) zmid_i = Ap->zmid[g.at(X.k, c)];
    JTerms jt{0.0, 0.0, 0.0, 0.0};
    if (active) {
      const int t = threadIdx.x;
      const long long ix = g.at(X.k, c);
      SweepCoreOut o;
      o.reminPOC = s_ho[H_OUT_REMINPOC][t]; o.reminCa = s_ho[H_OUT_REMINCA][t]; o.reminSi = s_ho[H_OUT_REMINSI][t];
      o.reminFe = s_ho[H_OUT_REMINFE][t]; o.Fe_scavenge = s_ho[H_OUT_FESCAV][t]; o.SED_DENITRIF = s_ho[H_OUT_SEDDEN][t];
      o.OTHER_REMIN = s_ho[H_OUT_OTHER][t];
      if (DIAG) {
        o.inPOC = s_ho[H_OUT_INPOC][t]; o.inCa = s_ho[H_OUT_INCA][t]; o.inSi = s_ho[H_OUT_INSI][t]; o.inDust = s_ho[H_OUT_INDUST][t];
        o.inFe = s_ho[H_OUT_INFE][t]; o.reminDust = s_ho[H_OUT_REMINDUST][t]; o.Fe_scav_rate = s_ho[H_OUT_FESCAVRATE][t];
        o.sedPOC = s_ho[H_OUT_SEDPOC][t]; o.sedCa = s_ho[H_OUT_SEDCA][t]; o.sedSi = s_ho[H_OUT_SEDSI][t];
        o.sedDust = s_ho[H_OUT_SEDDUST][t]; o.sedFe = s_ho[H_OUT_SEDFE][t];
      }
      lo.L_no3 = s_ho[H_L_NO3][t]; lo.L_nh4 = s_ho[H_L_NH4][t]; lo.L_fe = s_ho[H_L_FE][t]; lo.L_sio3 = s_ho[H_L_SIO3][t];
      lo.L_po4 = s_ho[H_L_PO4][t]; lo.L_donr = s_ho[H_L_DONR][t]; lo.L_dopr = s_ho[H_L_DOPR][t]; lo.L_dic = s_ho[H_L_DIC][t];
      lo.L_alk = s_ho[H_L_ALK][t]; lo.L_o2 = s_ho[H_L_O2][t];
      jt = sweep_finish<DIAG, FD>(*Ap, *Pp, ix, lo, o, X.k < kmax);
    }
    OBGC_T(5);   // finish
    // no barrier here: the next chunk's first write into the slots sits behind its light barrier
    CarbOut co{0.0, 0.0, 0.0};
    if (FUSE) carbonate_solve(*Ap, X.valid, X.k, cin, DIAG ? &co : nullptr);
    if (DIAG && active && X.k < kmax) {
      // integrands of the column integrals, into this lane's own (now dead) entries of the hand-off slots
      const int t = threadIdx.x;
      s_ho[I_JC][t] = jt.C; s_ho[I_JN][t] = jt.N; s_ho[I_JP][t] = jt.P; s_ho[I_JSI][t] = jt.Si;
      s_ho[I_DZ][t] = lo.dz; s_ho[I_ZB][t] = lo.zb; s_ho[I_O2][t] = lo.O2; s_ho[I_CHL][t] = lo.chl100;
      s_ho[I_ZMID][t] = FUSE ? cin.zmid : zmid_i;
      if (FUSE) { s_ho[I_CO3][t] = co.CO3; s_ho[I_SATC][t] = co.sat_calc; s_ho[I_SATA][t] = co.sat_arag; }
    }
    OBGC_T(6);
  }
  if (DIAG) {
    lds_barrier();
    integrate_chunk<FUSE>(s_ho, s_int, role, lane, kmax, nchunk - 1);
    if (colvalid) {   // each wavefront stores the integrals it advanced (its own LDS entries: no barrier)
      const obgc_bgc_diag& D = A.diag;
#define ST_C(name, n) do { if (D.name) D.name[c] = s_int[n][lane]; } while (0)
      if (role == 0) { ST_C(Jint_Ctot, N_J0); ST_C(Jint_100m_Ctot, N_J1); ST_C(Jint_Ntot, N_J2); ST_C(Jint_100m_Ntot, N_J3); }
      else if (role == 1) { ST_C(Jint_Ptot, N_J4); ST_C(Jint_100m_Ptot, N_J5); ST_C(Jint_Sitot, N_J6); ST_C(Jint_100m_Sitot, N_J7); }
      else if (role == 2) { if (FUSE) { ST_C(zsatcalc, N_ZC); ST_C(zsatarag, N_ZA); } }
      else { ST_C(O2_ZMIN, N_O2MIN); ST_C(O2_ZMIN_DEPTH, N_O2Z); ST_C(Chl_TOT_zint_100m, N_CHL); }
#undef ST_C
    }
  }
#ifdef OBGC_COLTIMING
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_coltiming[i][w], tacc[i]);
#endif
}

// ---------------------------------------------------------------- column integrals (diagnostics only)
// Vertical integrals of the production profiles (BGC_mod.F90:1230, 1273-1277, 1816-1846) and, for the separate kernels,
// chlorophyll over the top 100 m and the O2 minimum (1940-1945, 1954-1968), which the column-block kernel advances itself
// (A.colint).  None of them feeds or needs the particle recurrence.  Sums keep the reference's top-down order.
template <bool COLINT>
__global__ __launch_bounds__(kBlockCol) void bgc_zint_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  const obgc_autotroph* at = P.autotrophs;
  const long long pl = g.plane;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;
  double z_bSi = 0.0, z_CaTot = 0.0, z_pCTot = 0.0, z_pCNO3Tot = 0.0, z_Chl100 = 0.0;
  double z_Ca[4] = {0, 0, 0, 0}, z_pC[4] = {0, 0, 0, 0}, z_pCNO3[4] = {0, 0, 0, 0};
  double o2min = 0.0, o2min_z = 0.0;
  const double* chl100 = A.chl100;
  const double* o2p = A.tracers + (long long)(P.ind[5] - 1) * pl;
  bool has_si[4], calcifies[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) { has_si[a] = at[a].Si_ind > 0; calcifies[a] = at[a].imp_calcifier != 0; }
  // not unrolled: three wavefronts per SIMD instead of two (2.68 against 2.80 ms at 1 M x 80); four levels at a time gain a
  // small batch nothing either
#pragma unroll 1
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    // every load of the level first, the sums behind them: one memory round trip per level.  With each load next to its sum
    // the wavefront waited once per autotroph and conditional -- up to a dozen round trips a level, which is all a batch of a
    // few hundred wavefronts spends its time on.
    const double dz = A.dz[ix];
    double O2raw = 0.0, zm = 0.0, chl = 0.0;
    if (!COLINT) { O2raw = o2p[ix]; zm = A.zmid[ix]; chl = chl100[ix]; }
    double pC[4], pN[4], bSi[4], Ca[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      pC[a] = A.z_photoC[(long long)a * pl + ix];
      pN[a] = A.z_photoC_NO3[(long long)a * pl + ix];
      bSi[a] = has_si[a] ? A.z_bSi_form[(long long)a * pl + ix] : 0.0;
      Ca[a] = calcifies[a] ? A.z_CaCO3_form[(long long)a * pl + ix] : 0.0;
    }
    if (!COLINT) {
      const double O2 = dmax(0.0, O2raw);
      if (k == 0 || O2 < o2min) { o2min = O2; o2min_z = zm; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      z_pC[a] = z_pC[a] + dz * pC[a];
      if (has_si[a]) z_bSi = z_bSi + bSi[a];
      if (calcifies[a]) {
        const double w = dz * Ca[a];
        z_Ca[a] = z_Ca[a] + w;
        z_CaTot = z_CaTot + w;
      }
    }
    z_pCTot = z_pCTot + sum4(pC) * dz;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      z_pCNO3[a] = z_pCNO3[a] + pN[a] * dz;
      z_pCNO3Tot = z_pCNO3Tot + z_pCNO3[a];   // adds the RUNNING integral (1844-1846)
    }
    if (!COLINT) z_Chl100 = z_Chl100 + chl;
  }
  if (!COLINT) {
    if (D.O2_ZMIN) D.O2_ZMIN[c] = o2min;
    if (D.O2_ZMIN_DEPTH) D.O2_ZMIN_DEPTH[c] = o2min_z;
    if (D.Chl_TOT_zint_100m) D.Chl_TOT_zint_100m[c] = z_Chl100;
  }
  if (D.tot_bSi_form) D.tot_bSi_form[c] = z_bSi;
  if (D.tot_CaCO3_form_zint) D.tot_CaCO3_form_zint[c] = z_CaTot;
  if (D.photoC_TOT_zint) D.photoC_TOT_zint[c] = z_pCTot;
  if (D.photoC_NO3_TOT_zint) D.photoC_NO3_TOT_zint[c] = z_pCNO3Tot;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (D.CaCO3_form_zint) D.CaCO3_form_zint[(long long)a * g.ldc + c] = z_Ca[a];
    if (D.photoC_zint) D.photoC_zint[(long long)a * g.ldc + c] = z_pC[a];
    if (D.photoC_NO3_zint) D.photoC_NO3_zint[(long long)a * g.ldc + c] = z_pCNO3[a];
  }
}

// The divisor's share of a division, exactly as the compiler expands it in the kernels (Makefile, FASTDIV): hardware
// reciprocal and two Newton steps.
struct RecipIn { double y[R_N]; };
__device__ double g_recip_out[R_N];
__global__ void bgc_recip_kernel(const RecipIn in) {
  const int i = threadIdx.x;
  if (i >= (int)R_N) return;
  const double y = in.y[i];
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  g_recip_out[i] = r;
}

}  // namespace

hipError_t bgc_recips_compute(const obgc_bgc_params& P, double* recip_host, hipStream_t s) {
  static std::mutex mu;   // one result array per device: handles that share a device take turns
  std::lock_guard<std::mutex> lk(mu);
  RecipIn in;
  for (int i = 0; i < (int)R_N; ++i) in.y[i] = 1.0;
  for (int a = 0; a < 4; ++a) {
    const obgc_autotroph& t = P.autotrophs[a];
    in.y[R_KNO3 + a] = t.kNO3; in.y[R_KNH4 + a] = t.kNH4; in.y[R_KPO4 + a] = t.kPO4; in.y[R_KDOP + a] = t.kDOP;
    in.y[R_CKS_KFE + a] = CKS * t.kFe; in.y[R_CKSI_KSIO3 + a] = CKSI * t.kSiO3;   // the products as the kernels form them
  }
  in.y[R_RED] = RED_D_C_O2; in.y[R_REMIN] = REMIN_D_C_O2; in.y[R_RED_DIAZ] = RED_D_C_O2_DIAZ; in.y[R_DENITRIF] = DENITRIF_C_N;
  in.y[R_O2DELTA] = P.parm_o2_min_delta; in.y[R_TEN] = 10.0; in.y[R_35] = 35.0; in.y[R_4E6] = 4.0e6; in.y[R_12E7] = 1.2e7;
  for (int i = 0; i < 3; ++i) in.y[R_SCALELEN_DZ + i] = P.parm_scalelen_z[i + 1] - P.parm_scalelen_z[i];
  hipLaunchKernelGGL(bgc_recip_kernel, dim3(1), dim3(64), 0, s, in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  return hipMemcpyFromSymbol(recip_host, HIP_SYMBOL(g_recip_out), sizeof(double) * R_N);
}

void launch_bgc_att(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  hipLaunchKernelGGL(bgc_att_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, a, p);
}

void launch_bgc_parscan(const BgcKernelArgs& a, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  hipLaunchKernelGGL(bgc_parscan_kernel, dim3((unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol)), dim3(kBlockCol), 0, s, a);
}

void launch_bgc_cell(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, int fuse_carbonate) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  const unsigned gcell = (unsigned)((ncell + kBlockCell - 1) / kBlockCell);
  if (fuse_carbonate) {
    if (a.have_diag) hipLaunchKernelGGL((bgc_cell_kernel<true, 1>), dim3(gcell), dim3(kBlockCell), 0, s, a, p);
    else hipLaunchKernelGGL((bgc_cell_kernel<false, 1>), dim3(gcell), dim3(kBlockCell), 0, s, a, p);
  } else {
    if (a.have_diag) hipLaunchKernelGGL((bgc_cell_kernel<true, 0>), dim3(gcell), dim3(kBlockCell), 0, s, a, p);
    else hipLaunchKernelGGL((bgc_cell_kernel<false, 0>), dim3(gcell), dim3(kBlockCell), 0, s, a, p);
  }
}

void launch_bgc_ecosys_cells(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, hipEvent_t ev_cell0, hipEvent_t ev_cell1,
                             int fuse_carbonate) {
  if (a.g.ncol <= 0) return;
  launch_bgc_att(a, p, s);
  launch_bgc_parscan(a, s);
  if (ev_cell0) (void)hipEventRecord(ev_cell0, s);
  launch_bgc_cell(a, p, s, fuse_carbonate);
  if (ev_cell1) (void)hipEventRecord(ev_cell1, s);
}

// clears the per-column flags of the fused light pass (a kernel, not a memset node: the call may be under graph capture)
__global__ __launch_bounds__(256) void clear_flags_kernel(int* __restrict__ flag, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = 0;
}

void launch_fused_ecosys_cells(const BgcKernelArgs& a, const obgc_bgc_params& p, const FusedDmArgs& f, const FusedParams& fp, hipStream_t s,
                               hipEvent_t ev_cell0, hipEvent_t ev_cell1, int fuse_carbonate) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  if (f.dms && f.flag) hipLaunchKernelGGL(clear_flags_kernel, dim3((unsigned)((a.g.ncol + 255) / 256)), dim3(256), 0, s, f.flag, a.g.ncol);
  hipLaunchKernelGGL(fused_att_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, a, p, f.dms ? f.flag : nullptr);
  hipLaunchKernelGGL(fused_parscan_kernel, dim3((unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol)), dim3(kBlockCol), 0, s, a, p, f);
  if (ev_cell0) (void)hipEventRecord(ev_cell0, s);
  const unsigned gcell = (unsigned)((ncell + kBlockCell - 1) / kBlockCell);
  if (fuse_carbonate) {
    if (a.have_diag) hipLaunchKernelGGL((fused_cell_kernel<true, 1>), dim3(gcell), dim3(kBlockCell), 0, s, a, p, f, fp);
    else hipLaunchKernelGGL((fused_cell_kernel<false, 1>), dim3(gcell), dim3(kBlockCell), 0, s, a, p, f, fp);
  } else {
    if (a.have_diag) hipLaunchKernelGGL((fused_cell_kernel<true, 0>), dim3(gcell), dim3(kBlockCell), 0, s, a, p, f, fp);
    else hipLaunchKernelGGL((fused_cell_kernel<false, 0>), dim3(gcell), dim3(kBlockCell), 0, s, a, p, f, fp);
  }
  if (ev_cell1) (void)hipEventRecord(ev_cell1, s);
}

void launch_bgc_column(const BgcKernelArgs& a, const obgc_bgc_params& p, const FusedDmArgs* f, const FusedParams* fp, hipStream_t s,
                       int fuse_carbonate) {
  if (a.g.ncol <= 0) return;
  const unsigned grid = (unsigned)((a.g.ncol + 63) / 64);
  FusedDmArgs f0;
  FusedParams fp0;
  if (!f) { std::memset(&f0, 0, sizeof f0); std::memset(&fp0, 0, sizeof fp0); }
  const FusedDmArgs& F = f ? *f : f0;
  const FusedParams& FP = fp ? *fp : fp0;
#define OBGC_LAUNCH_COL(DIAG, DM, FUSE) hipLaunchKernelGGL((bgc_column_kernel<DIAG, DM, FUSE>), dim3(grid), dim3(kColThreads), 0, s, a, p, F, FP)
#define OBGC_LAUNCH_COL_FD(DM, FUSE) hipLaunchKernelGGL((bgc_column_kernel<true, DM, FUSE, true>), dim3(grid), dim3(kColThreads), 0, s, a, p, F, FP)
  const bool dm = f != nullptr;
  // every diagnostic the cell terms and the finishing arithmetic store (and, in the fused step, every one of DMS and MACROS)
  bool full = a.have_diag != 0;
  if (full) {
    // (three members of the reference's type are never written: BGC_mod.F90 has no statement for them)
#define X(n) full = full && (a.diag.n != nullptr || !std::strcmp(#n, "POC_ACCUM") || !std::strcmp(#n, "DONr_remin") || !std::strcmp(#n, "DOPr_remin"));
    OBGC_BGC_DIAG_KC(X)
#undef X
#define X(n) full = full && a.diag.n != nullptr;
    OBGC_BGC_DIAG_KCA(X)
#undef X
    if (dm && f->dms) {
      full = full && f->have_ddiag;
#define X(n) full = full && f->ddiag.n != nullptr;
      OBGC_DMS_DIAG_KC(X)
#undef X
    }
    if (dm && f->prot) {
      full = full && f->have_mdiag;
#define X(n) full = full && f->mdiag.n != nullptr;
      OBGC_MACROS_DIAG_KC(X)
#undef X
    }
  }
  if (full) {
    if (dm) { if (fuse_carbonate) OBGC_LAUNCH_COL_FD(true, 1); else OBGC_LAUNCH_COL_FD(true, 0); }
    else { if (fuse_carbonate) OBGC_LAUNCH_COL_FD(false, 1); else OBGC_LAUNCH_COL_FD(false, 0); }
  } else if (a.have_diag) {
    if (dm) { if (fuse_carbonate) OBGC_LAUNCH_COL(true, true, 1); else OBGC_LAUNCH_COL(true, true, 0); }
    else { if (fuse_carbonate) OBGC_LAUNCH_COL(true, false, 1); else OBGC_LAUNCH_COL(true, false, 0); }
  } else {
    if (dm) { if (fuse_carbonate) OBGC_LAUNCH_COL(false, true, 1); else OBGC_LAUNCH_COL(false, true, 0); }
    else { if (fuse_carbonate) OBGC_LAUNCH_COL(false, false, 1); else OBGC_LAUNCH_COL(false, false, 0); }
  }
#undef OBGC_LAUNCH_COL
#undef OBGC_LAUNCH_COL_FD
#ifdef OBGC_COLTIMING
  {
    (void)hipDeviceSynchronize();
    unsigned long long t[8][kColLevels];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_coltiming), sizeof t);
    const char* nm[8] = {"loads", "cell terms", "barrier 1", "walk", "barrier 2", "finish", "carbonate", "-"};
    const double nwg = (double)grid, nch = (double)((a.g.nlev + kColLevels - 1) / kColLevels);
    std::fprintf(stderr, "column kernel: cycles per chunk and wavefront (mean over %u workgroups x %.0f chunks)\n", grid, nch);
    for (int i = 0; i < 7; ++i) {
      std::fprintf(stderr, "  %-11s", nm[i]);
      for (int ww = 0; ww < kColLevels; ++ww) std::fprintf(stderr, " %9.0f", (double)t[i][ww] / (nwg * nch));
      std::fprintf(stderr, "\n");
    }
    std::memset(t, 0, sizeof t);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_coltiming), t, sizeof t);
  }
#endif
}

void launch_bgc_zint(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  if (a.g.ncol <= 0 || !a.have_diag) return;
  const unsigned gcol = (unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol);
  if (a.colint) hipLaunchKernelGGL(bgc_zint_kernel<true>, dim3(gcol), dim3(kBlockCol), 0, s, a, p);
  else hipLaunchKernelGGL(bgc_zint_kernel<false>, dim3(gcol), dim3(kBlockCol), 0, s, a, p);
}

template <int LANES>
static void launch_sweep_lanes(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  const unsigned grid = (unsigned)((a.g.ncol + LANES - 1) / LANES);
  if (a.have_diag) hipLaunchKernelGGL((bgc_sweep_kernel<true, LANES>), dim3(grid), dim3(kBlockCol), 0, s, a, p);
  else hipLaunchKernelGGL((bgc_sweep_kernel<false, LANES>), dim3(grid), dim3(kBlockCol), 0, s, a, p);
}

void launch_bgc_split_sweep(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  const unsigned gcell = (unsigned)(((long long)a.g.nlev * a.g.ncol + 255) / 256), gcol = (unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol);
  hipLaunchKernelGGL(bgc_derive_kernel, dim3(gcell), dim3(256), 0, s, a, p);
  if (a.have_diag) {
    hipLaunchKernelGGL(bgc_walk_kernel<true>, dim3(gcol), dim3(kBlockCol), 0, s, a, p);
    hipLaunchKernelGGL(bgc_finish_kernel<true>, dim3(gcell), dim3(256), 0, s, a, p);
    hipLaunchKernelGGL(bgc_jint_kernel, dim3(gcol), dim3(kBlockCol), 0, s, a);
  } else {
    hipLaunchKernelGGL(bgc_walk_kernel<false>, dim3(gcol), dim3(kBlockCol), 0, s, a, p);
    hipLaunchKernelGGL(bgc_finish_kernel<false>, dim3(gcell), dim3(256), 0, s, a, p);
  }
}

void launch_bgc_ecosys_sweep(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, int lanes) {
  if (a.g.ncol <= 0) return;
  // Measured on MI355X at 100k columns (OBGC_SWEEP_LANES): 64 lanes 4.87 ms/step, 32 lanes 5.04, 16 lanes 5.85 --
  // partially filled wavefronts multiply the independent chains but waste memory-instruction issue, which is
  // what binds the sweep; full wavefronts stay the default.
  if (lanes <= 0) lanes = 64;
  if (lanes >= 64) launch_sweep_lanes<64>(a, p, s);
  else if (lanes >= 32) launch_sweep_lanes<32>(a, p, s);
  else launch_sweep_lanes<16>(a, p, s);
}

}  // namespace obgc
