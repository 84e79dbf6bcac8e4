// Ecosystem source/sink terms of BGC_SourceSink as four kernels:
//
//   bgc_att_kernel     1 lane / cell     attenuation exp(-KPARdz) of every level (pow + exp)
//   bgc_parscan_kernel 1 lane / column   PAR entering every level: one multiply per level
//   bgc_cell_kernel    1 lane / cell     every cell-local term: quotas, uptake, photoadaptation,
//                                        grazing, mortality, DOM, the PRODUCTION terms and decay
//                                        factors of the sinking particles, 19 final tendencies
//   bgc_sweep_kernel   1 lane / column   the Armstrong ballast recurrence top to bottom, iron
//                                        scavenging, sea-floor burial, and the 11 tendencies and
//                                        the column integrals that depend on it
//
// Behavioural contract = BGC_mod.F90:733-931 (clamps, zero-mask, quotas, light), 1041-1529
// (kinetics), 1531-1536 -> 2116-2699 (particles), 1545-1968 (tendencies, diagnostics, column
// integrals, O2 minimum), 2006-2109 (particle initialisation).  The carbonate block 935-1032
// is a separate cell-parallel kernel (obgc_carbonate.hip).
//
// Why three: the reference walks each column sequentially, but only the particle fluxes, the
// QA-dust deficit, PAR and the column integrals are really carried from level to level -- a few
// dozen flops per cell.  The ~2 000 flops of kinetics per cell are level-independent once PAR
// is known.  With one lane per column the 100k-column config offers 1 563 wavefronts for 1 024
// SIMDs (measured round 1: 5.2 ms, latency bound at one 512-register wave per SIMD); with one
// lane per cell it offers 94 000.  The price is a scratch hand-off between the cell kernel and
// the sweep (22 doubles per cell, + 5 with diagnostics) in the same layout as the caller's
// arrays, so a single index addresses everything.
//
// Arithmetic order is the reference's everywhere except three tendency sums that the hand-off
// re-associates (Fe, SiO3, PO4: the per-autotroph terms are folded before the particle terms
// are added) and P_iron production (Fe scavenging added last); the change is a few ulp of the
// largest term, far inside the parity bar, and is listed in DESIGN.md.
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

// DM: DMS_SourceSink and MACROS_SourceSink of the same cell are evaluated too (obgc_fused_step): the tracers they share
// with BGC are in registers already, only their own five tracers are extra loads.
template <bool DIAG, bool DM>
__device__ __forceinline__ void cell_body(const BgcKernelArgs& A, const obgc_bgc_params& P, const FusedDmArgs* F, const FusedParams* FP) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  const long long ix = g.at(k, c), pl = g.plane;
  const obgc_bgc_diag& D = A.diag;
  const obgc_autotroph* at = P.autotrophs;
#define RAW(slot) A.tracers[(long long)((slot)-1) * pl + ix]
#define TEND(slot) A.tend[(long long)((slot)-1) * pl + ix]
#define SCR(s) A.scratch[(long long)(s) * A.splane + ix]
#define ST_KC(name, v) do { if (DIAG && D.name) D.name[ix] = (v); } while (0)
#define ST_KCA(name, a, v) do { if (DIAG && D.name) D.name[(long long)(a) * pl + ix] = (v); } while (0)

  // Every global load of the kernel is issued here, back to back, before anything waits: ~45 independent
  // requests per lane and a single memory round trip per wavefront.  Cells below the sea floor load too (valid
  // addresses, values discarded) so that the kmax test does not cost a round trip of its own.
  int kmax = A.kmax[c];
  bool dflag = false;   // this column has a DMS light pass of its own
  if (DM && F->dms) dflag = F->all_flagged || F->flag[c] != 0;
  const double lat = A.lat[c];
  const double T = A.temp[ix], dz = A.dz[ix], zc = A.zmid[ix];
  const double r_PO4 = RAW(P.ind[0]), r_NO3 = RAW(P.ind[1]), r_SiO3 = RAW(P.ind[2]), r_NH4 = RAW(P.ind[3]), r_Fe = RAW(P.ind[4]),
               r_O2 = RAW(P.ind[5]), r_DOC = RAW(P.ind[9]), r_DON = RAW(P.ind[10]), r_DOFe = RAW(P.ind[11]), r_DOP = RAW(P.ind[12]),
               r_DOPr = RAW(P.ind[13]), r_DONr = RAW(P.ind[14]), r_zooC = RAW(P.ind[15]);
  const double PAR_in = SCR(S_PARIN), KPARdz = SCR(S_KPARDZ), att = SCR(S_ATT);
  double r_salt = 0.0, ztop = 0.0;
  if (DIAG) {
    r_salt = A.salt[ix];
    ztop = A.zbot[g.at(k > 0 ? k - 1 : 0, c)];
    if (k == 0) ztop = 0.0;
  }
  Pool1 raw[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) raw[a] = load_pool_raw(A, at[a], ix);
  double o_dms = 0.0, o_dmsp = 0.0, o_sst = 0.0, o_spca = 0.0, o_prot = 0.0, o_poly = 0.0, o_lip = 0.0;
  double d_parin = 0.0, d_att = 0.0, d_kpardz = 1.0;
  if (DM) {
    if (F->dms) { o_dms = F->dms[ix]; o_dmsp = F->dmsp[ix]; o_sst = F->SST[c]; o_spca = RAW(P.ind[19]); }
    if (F->prot) { o_prot = F->prot[ix]; o_poly = F->poly[ix]; o_lip = F->lip[ix]; }
    if (dflag) { d_parin = F->d_parin[ix]; d_att = F->d_att[ix]; d_kpardz = F->d_kpardz[ix]; }   // last: the only dependent loads
  }
  if (kmax > g.nlev) kmax = g.nlev;

  if (k >= kmax) {
    // below the sea floor / land: the reference zero-fills tendencies and diagnostics (570, 625-727; DMS_mod.F90:413,
    // MACROS_mod.F90:267 for the tendencies of the other two modules, whose diagnostics it leaves alone)
    if (DM) {
      if (F->dms) { F->dms_tend[ix] = 0.0; F->dmsp_tend[ix] = 0.0; }
      if (F->prot) { F->prot_tend[ix] = 0.0; F->poly_tend[ix] = 0.0; F->lip_tend[ix] = 0.0; }
    }
#pragma unroll 6
    for (int n = 0; n < OBGC_BGC_NTRACER; ++n) A.tend[(long long)n * pl + ix] = 0.0;
    if (DIAG) {
#define ZKC(n) if (D.n) D.n[ix] = 0.0;
      ZKC(tot_Nfix) ZKC(O2_PRODUCTION) ZKC(O2_CONSUMPTION) ZKC(AOU) ZKC(PO4_RESTORE) ZKC(NO3_RESTORE) ZKC(SiO3_RESTORE)
      ZKC(PAR_avg) ZKC(POC_FLUX_IN) ZKC(POC_PROD) ZKC(POC_REMIN) ZKC(CaCO3_FLUX_IN) ZKC(CaCO3_PROD) ZKC(CaCO3_REMIN)
      ZKC(SiO2_FLUX_IN) ZKC(SiO2_PROD) ZKC(SiO2_REMIN) ZKC(dust_FLUX_IN) ZKC(dust_REMIN) ZKC(P_iron_FLUX_IN)
      ZKC(P_iron_PROD) ZKC(P_iron_REMIN) ZKC(auto_graze_TOT) ZKC(zoo_loss) ZKC(photoC_TOT) ZKC(photoC_NO3_TOT)
      ZKC(DOC_prod) ZKC(DOC_remin) ZKC(DON_prod) ZKC(DON_remin) ZKC(DOFe_prod) ZKC(DOFe_remin) ZKC(DOP_prod)
      ZKC(DOP_remin) ZKC(Fe_scavenge) ZKC(Fe_scavenge_rate) ZKC(NITRIF) ZKC(DENITRIF) ZKC(calcToSed) ZKC(pocToSed)
      ZKC(ponToSed) ZKC(popToSed) ZKC(bsiToSed) ZKC(dustToSed) ZKC(pfeToSed) ZKC(SedDenitrif) ZKC(OtherRemin)
      ZKC(tot_CaCO3_form)
#undef ZKC
#define ZKCA(n) if (D.n) { D.n[ix] = 0.0; D.n[pl + ix] = 0.0; D.n[2 * pl + ix] = 0.0; D.n[3 * pl + ix] = 0.0; }
      ZKCA(N_lim) ZKCA(P_lim) ZKCA(Fe_lim) ZKCA(SiO3_lim) ZKCA(light_lim) ZKCA(photoC) ZKCA(photoC_NO3) ZKCA(photoFe)
      ZKCA(photoNO3) ZKCA(photoNH4) ZKCA(DOP_uptake) ZKCA(PO4_uptake) ZKCA(auto_graze) ZKCA(auto_loss) ZKCA(auto_agg)
      ZKCA(bSi_form) ZKCA(CaCO3_form) ZKCA(Nfix)
#undef ZKCA
    }
    return;
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
      dms_cell_terms<true>(FP->dms, F->ddiag, ix, col, dmax(0.0, o_dms), dmax(0.0, o_dmsp), c_zoo, c_sp, dmax(0.0, o_spca), c_diat,
                           c_diaz, c_phaeo, dmax(0.0, pick_Chl(raw, P.sp_ind - 1)), PAR_avg_dms, t_dms, t_dmsp);
      F->dms_tend[ix] = t_dms;
      F->dmsp_tend[ix] = t_dmsp;
    }
    if (F->prot) {
      double t_prot, t_poly, t_lip;
      macros_cell_terms<true>(FP->macros, F->mdiag, ix, c_zoo, c_sp, c_diat, c_diaz, c_phaeo, dmax(0.0, o_prot), dmax(0.0, o_poly),
                              dmax(0.0, o_lip), t_prot, t_poly, t_lip);
      F->prot_tend[ix] = t_prot;
      F->poly_tend[ix] = t_poly;
      F->lip_tend[ix] = t_lip;
    }
  }

  // Everything below the autotroph loop that needs a global LOAD is evaluated now and parked in LDS: loads and
  // stores retire through one in-order counter, so a load issued after the loop's ~100 diagnostic stores would
  // wait for every one of them to be acknowledged.  From here on the kernel only stores.
  //   DOM remineralisation (1436-1461), nitrification (1554-1561), the O2 weights of denitrification and
  //   O2 consumption (1569-1577, 1784-1790), AOU, the 100 m weight of the Chl integral
  __shared__ double s_tail[9][kBlockCell];
  const int tid = threadIdx.x;
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
    double wd = ((P.parm_o2_min + P.parm_o2_min_delta) - O2) / P.parm_o2_min_delta;
    wd = dmin(dmax(wd, 0.0), 1.0);
    if (NO3 == 0.0) wd = 0.0;
    double wo = (O2 - P.parm_o2_min) / P.parm_o2_min_delta;
    wo = dmin(dmax(wo, 0.0), 1.0);
    s_tail[0][tid] = DOC_remin; s_tail[1][tid] = DON_remin; s_tail[2][tid] = DOFe_remin; s_tail[3][tid] = DOP_remin;
    s_tail[4][tid] = DONr_remin; s_tail[5][tid] = DOPr_remin; s_tail[6][tid] = NITRIF; s_tail[7][tid] = wd * (DOC_remin / DENITRIF_C_N);
    s_tail[8][tid] = wo;
    if (DIAG && D.AOU) aou = o2sat(T, r_salt, T0K) - O2;
  }
  double pt100 = 0.0;   // thickness of this level above 100 m (1940-1945)
  if (DIAG) {
    const double w = dmin(100.0e2 - ztop, dz);
    pt100 = (w > 0.0) ? w : 0.0;
  }

  // temperature response and loss thresholds (1041-1055)
  const double Tfunc = exp((((T + T0K) - (TREF + T0K)) / 10.0) * LN_Q_10);   // Q_10 ** ((T + T0) - (Tref + T0)) / 10)
  double f_loss_thres;
  if (zc > THRES_Z1) f_loss_thres = (zc < THRES_Z2) ? (THRES_Z2 - zc) / (THRES_Z2 - THRES_Z1) : 0.0;
  else f_loss_thres = 1.0;

  // ---- pre-pass: Pprime (1072-1094), grazing (1297-1324), zooplankton losses (1395-1415) ----
  // The masked pools are read from HBM once and parked in LDS for the rolled loop below (each lane only
  // ever touches its own column of the array, so no barrier is needed).
  __shared__ double s_pool[20][kBlockCell];
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
    if (Fe < CKS * t.kFe) gQfe = dmax((gQfe * Fe / (CKS * t.kFe)), t.gQfe_min);
    double gQsi = 0.0;
    if (t.Si_ind > 0) {
      gQsi = GQSI_0;
      if ((Fe < CKSI * t.kFe) && (Fe > 0.0) && (SiO3 > (CKSI * t.kSiO3))) gQsi = dmin((gQsi * CKSI * t.kFe / Fe), GQSI_MAX);
      if (Fe == 0.0) gQsi = GQSI_MAX;
      if (SiO3 < (CKSI * t.kSiO3)) gQsi = dmax((gQsi * SiO3 / (CKSI * t.kSiO3)), GQSI_MIN);
    }
    double QCa = 0.0;
    if (t.CaCO3_ind > 0) {
      QCa = p.Ca / Cden;
      if (QCa > QCACO3_MAX) QCa = QCACO3_MAX;
    }
    // nutrient limitation (1108-1139)
    const double rNO3 = NO3 / t.kNO3, rNH4 = NH4 / t.kNH4;
    const double VNO3 = rNO3 / (1.0 + rNO3 + rNH4);
    const double VNH4 = rNH4 / (1.0 + rNO3 + rNH4);
    double VNtot = VNO3 + VNH4;
    if (t.Nfixer) VNtot = 1.0;
    ST_KCA(N_lim, a, VNtot);
    const double VFe = Fe / (Fe + t.kFe);
    ST_KCA(Fe_lim, a, VFe);
    double f_nut = dmin(VNtot, VFe);
    const double rPO4 = PO4 / t.kPO4, rDOP = DOP / t.kDOP;
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
      A.z_photoC[(long long)a * pl + ix] = photoC;
      A.z_photoC_NO3[(long long)a * pl + ix] = pC_NO3;
      A.z_CaCO3_form[(long long)a * pl + ix] = CaProd;
      A.z_bSi_form[(long long)a * pl + ix] = photoSi;
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
        O2_PRODUCTION = O2_PRODUCTION + photoC * ((NO3_V / den) / RED_D_C_O2 + (NH4_V / den) / REMIN_D_C_O2);
      } else {
        const double den = NO3_V + NH4_V + Nfix;
        O2_PRODUCTION = O2_PRODUCTION + photoC * ((NO3_V / den) / RED_D_C_O2 + (NH4_V / den) / REMIN_D_C_O2 + (Nfix / den) / RED_D_C_O2_DIAZ);
      }
    }
    // autotroph tendencies (1676-1697)
    wk = gz + loss + ag;
    const double t_C = photoC - wk;
    TEND(t.C_ind) = t_C;
    TEND(t.Chl_ind) = photoacc - thetaC * wk;
    TEND(t.Fe_ind) = photoFe - Qfe * wk;
    if (t.Si_ind > 0) { const double v = photoSi - Qsi * wk; TEND(t.Si_ind) = v; sSi = sSi + v; }
    if (t.CaCO3_ind > 0) { const double v = CaProd - QCa * wk; TEND(t.CaCO3_ind) = v; sCa = sCa + v; }
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

  SCR(S_PRODPOC) = prodPOC; SCR(S_PRODCA) = prodCa; SCR(S_PRODSI) = prodSi; SCR(S_PFELOC) = PfeLoc;

  // ---- local parts of denitrification (1569-1577) and O2 consumption (1784-1790): the terms that do not
  //      involve the particle remineralisation, weighted in the prologue; the sweep adds  w * (remin terms)  ----
  const double S3 = zoo_loss_dic + s_ls_dic + s_gr_dic;
  const double CAD = ca_x - ca_prod;
  const double OC0 = wo * ((DOC_remin + S3) / REMIN_D_C_O2 + (2.0 * NITRIF));

  // ---- tendencies that do not see the particles (1703-1723) ----
  const double t_zooC = s_gr_zoo - zoo_loss;
  const double t_doc = DOC_prod - DOC_remin;
  const double t_don = (DON_prod * (1.0 - DONREFRACT)) - DON_remin;
  const double t_dop = (DOP_prod * (1.0 - DOPREFRACT)) - DOP_remin - sDOPV;
  TEND(P.ind[15]) = t_zooC;
  TEND(P.ind[9]) = t_doc;
  TEND(P.ind[10]) = t_don;
  TEND(P.ind[11]) = DOFe_prod - DOFe_remin;
  TEND(P.ind[12]) = t_dop;
  // ---- particle-dependent tendencies: everything local is folded here and waits in the caller's
  //      tendency plane; the sweep adds the remineralisation terms (listed with each line) ----
  TEND(P.ind[1]) = (NITRIF - sNO3V) - D0;                                             // 1583: - wd*((reminPOC-OTHER)/C - SED) - SED
  TEND(P.ind[3]) = ((-sNH4V - NITRIF + DON_remin + DONr_remin) + QN * S3) + nexc;     // 1585-1592: + QN*(1-DONrefract)*reminPOC
  TEND(P.ind[4]) = ((QFE_ZOO * zoo_loss_dic) + DOFe_remin - sPhotoFe) + L_fe_a;       // 1598-1605: + reminFe - Fe_scavenge
  TEND(P.ind[2]) = L_sio3_a;                                                          // 1620-1628: + reminSi
  TEND(P.ind[0]) = ((DOP_remin + DOPr_remin - sPO4V) + L_po4_a) + QP_ZOO_POM * zoo_loss_dic;   // 1643-1661: + Qp*(1-DOPrefract)*reminPOC
  TEND(P.ind[14]) = (DON_prod * DONREFRACT) - DONr_remin;                             // 1714: + reminPOC*DONrefract*Q
  TEND(P.ind[13]) = (DOP_prod * DOPREFRACT) - DOPr_remin;                             // 1720: + reminPOC*DOPrefract*Qp
  TEND(P.ind[6]) = ((s_ls_dic + s_gr_dic - sPhotoC + DOC_remin) + zoo_loss_dic) + CAD;   // 1729-1739: + reminPOC + reminCaCO3
  TEND(P.ind[8]) = 2.0 * CAD;                                                         // 1749-1759: - t_no3 + t_nh4 + 2 reminCaCO3
  TEND(P.ind[5]) = O2_PRODUCTION - OC0;                                               // 1790: - wo*(reminPOC - SED*C - OTHER)/R

  if (DIAG) {
    ST_KC(NO3_RESTORE, 0.0); ST_KC(SiO3_RESTORE, 0.0); ST_KC(PO4_RESTORE, 0.0);       // lrest_* never set (131-134)
    ST_KC(NITRIF, NITRIF); ST_KC(O2_PRODUCTION, O2_PRODUCTION);
    ST_KC(zoo_loss, zoo_loss);
    ST_KC(DOC_prod, DOC_prod); ST_KC(DOC_remin, DOC_remin); ST_KC(DON_prod, DON_prod); ST_KC(DON_remin, DON_remin);
    ST_KC(DOP_prod, DOP_prod); ST_KC(DOP_remin, DOP_remin); ST_KC(DOFe_prod, DOFe_prod); ST_KC(DOFe_remin, DOFe_remin);
    // local parts of the conservation integrals (1875-1938); the sweep adds the particle-dependent tendencies
    SCR(S_OC0) = OC0; SCR(S_D0) = D0;
    SCR(S_JC) = t_doc + t_zooC + sC + sCa;
    SCR(S_JN) = t_don + QN * t_zooC + QN * sC - sNfix;
    SCR(S_JP) = t_dop + QP_ZOO_POM * t_zooC + sQpC;
    SCR(S_JSI) = sSi;
    SCR(S_CHL100) = chl * pt100;
  }
#undef RAW
#undef TEND
#undef SCR
#undef ST_KC
#undef ST_KCA
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

template <bool DIAG, int FUSE>
__global__ __launch_bounds__(kBlockCell, 2) void bgc_cell_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  cell_body<DIAG, false>(A, P, nullptr, nullptr);
  if (FUSE) carbonate_of_this_lane(A, P);
}

static_assert(sizeof(BgcKernelArgs) + sizeof(obgc_bgc_params) + sizeof(FusedDmArgs) + sizeof(FusedParams) <= 4096,
              "kernel arguments of the fused cell kernel must fit the 4 KB kernarg segment");

// the cell kernel of obgc_fused_step: BGC + DMS + MACROS terms of a cell from one set of loads
template <bool DIAG, int FUSE>
__global__ __launch_bounds__(kBlockCell, 2) void fused_cell_kernel(const BgcKernelArgs A, const obgc_bgc_params P, const FusedDmArgs F,
                                                                   const FusedParams FP) {
  cell_body<DIAG, true>(A, P, &F, &FP);
  if (FUSE) carbonate_of_this_lane(A, P);
}

// ------------------------------------------------------------------------------ particle sweep
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
};

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

// LANES = active lanes per wavefront (experiment knob, see launch_bgc_ecosys_sweep; 64 is the default).
template <bool DIAG, int LANES>
__global__ __launch_bounds__(kBlockCol) void bgc_sweep_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  if ((int)threadIdx.x >= LANES) return;
  const int c = blockIdx.x * LANES + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  const long long pl = g.plane;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;

#define TEND(slot, ix) A.tend[(long long)((slot)-1) * pl + (ix)]
#define ST_KC(name, v) do { if (DIAG && D.name) D.name[ix] = (v); } while (0)
#define ST_C(name, v) do { if (DIAG && D.name) D.name[c] = (v); } while (0)

  // ---- per-column initialisation (BGC_mod.F90:808-811, 2075-2104) ----
  const double dust_in = (kmax >= 1) ? dmax(0.0, A.dust[c]) : 0.0;
  Flux fPOC{0.0, 0.0}, fCa{0.0, 0.0}, fSi{0.0, 0.0}, fFe{0.0, 0.0}, fDust{0.0, 0.0};
  if (dust_in != 0.0) { fDust.s = (1.0 - DUST_GAMMA) * dust_in; fDust.h = DUST_GAMMA * dust_in; }
  double QA_dust_def = DUST_RHO * (fDust.s + fDust.h);
  double J_C = 0, J_C100 = 0, J_N = 0, J_N100 = 0, J_P = 0, J_P100 = 0, J_Si = 0, J_Si100 = 0;
  double zbot_above = 0.0;

  LevelIn<DIAG> cur, nxt;
  if (kmax >= 1) { load_level<DIAG>(A, P, g.at(0, c), cur); settle<DIAG>(cur); }
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    if (k + 1 < kmax) load_level<DIAG>(A, P, g.at(k + 1, c), nxt);   // in flight while this level is evaluated
    const double dz = cur.dz, zb = cur.zb, NO3 = dmax(0.0, cur.NO3), Fe = dmax(0.0, cur.Fe), O2 = dmax(0.0, cur.O2);
    const double prodPOC = cur.prodPOC, prodCa = cur.prodCa, prodSi = cur.prodSi;

    // dissolution length scales and decay factors across this level (2273-2339): functions of the inputs
    // this kernel reads anyway, so they are evaluated here instead of travelling through HBM
    double scalelength;
    {
      const double* sz = P.parm_scalelen_z;
      const double* sv = P.parm_scalelen_vals;
      if (zb < sz[0]) scalelength = sv[0];
      else if (zb >= sz[3]) scalelength = sv[3];
      else if (zb < sz[1]) scalelength = sv[0] + (sv[1] - sv[0]) * (zb - sz[0]) / (sz[1] - sz[0]);
      else if (zb < sz[2]) scalelength = sv[1] + (sv[2] - sv[1]) * (zb - sz[1]) / (sz[2] - sz[1]);
      else scalelength = sv[2] + (sv[3] - sv[2]) * (zb - sz[2]) / (sz[3] - sz[2]);
    }
    const double TfuncS = exp((((cur.T + P.T0_Kelvin_BGC) - (TREF + P.T0_Kelvin_BGC)) / 10.0) * LN_Q_10);   // 1.5 ** (...)
    double poc_diss = P.parm_POC_diss;
    if ((O2 >= 5.0) && (O2 < 40.0)) poc_diss = P.parm_POC_diss * (1.0 + (3.3 - 1.0) * (40.0 - O2) / 35.0);
    else if (O2 < 5.0) poc_diss = P.parm_POC_diss * 3.3;
    poc_diss = scalelength * poc_diss;
    double sio2_diss = scalelength * P.parm_SiO2_diss;
    const double caco3_diss = scalelength * P.parm_CaCO3_diss;
    const double dust_diss = scalelength * DUST_DISS;
    sio2_diss = sio2_diss / TfuncS;
    const double decay_POC_E = exp(-dz / poc_diss), decay_SiO2 = exp(-dz / sio2_diss), decay_CaCO3 = exp(-dz / caco3_diss),
                 decay_dust = exp(-dz / dust_diss), DECAY_Hard = exp(-dz / 4.0e6), DECAY_HardDust = exp(-dz / 1.2e7);

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

    if (k == kmax - 1) {   // sea floor: burial, sediment denitrification, instantaneous remin (2522-2631)
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

    // ---- denitrification (1569-1577) and the particle-dependent tendencies (1583-1790): the cell kernel
    //      left every local term in the tendency planes; only the remineralisation terms are added here ----
    double wd = ((P.parm_o2_min + P.parm_o2_min_delta) - O2) / P.parm_o2_min_delta;
    wd = dmin(dmax(wd, 0.0), 1.0);
    if (NO3 == 0.0) wd = 0.0;
    const double den_p = wd * ((reminPOC - OTHER_REMIN) / DENITRIF_C_N - SED_DENITRIF);
    const double t_no3 = (cur.L_no3 - den_p) - SED_DENITRIF;
    const double t_nh4 = cur.L_nh4 + QN * (reminPOC * (1.0 - DONREFRACT));
    const double t_fe = (reminFe + cur.L_fe) - Fe_scavenge;
    const double t_sio3 = reminSi + cur.L_sio3;
    const double t_po4 = cur.L_po4 + QP_ZOO_POM * ((1.0 - DOPREFRACT) * reminPOC);
    const double t_donr = cur.L_donr + (reminPOC * DONREFRACT * QN);
    const double t_dopr = cur.L_dopr + (reminPOC * DOPREFRACT * QP_ZOO_POM);
    const double t_dic = (cur.L_dic + reminPOC) + reminCa;
    const double t_alk = (-t_no3 + t_nh4 + 2.0 * reminCa) + cur.L_alk;
    double wo = (O2 - P.parm_o2_min) / P.parm_o2_min_delta;
    wo = dmin(dmax(wo, 0.0), 1.0);
    const double oc_p = wo * ((reminPOC - (SED_DENITRIF * DENITRIF_C_N) - OTHER_REMIN) / REMIN_D_C_O2);
    double w1;
    TEND(P.ind[1], ix) = t_no3;
    TEND(P.ind[3], ix) = t_nh4;
    TEND(P.ind[4], ix) = t_fe;
    TEND(P.ind[2], ix) = t_sio3;
    TEND(P.ind[0], ix) = t_po4;
    TEND(P.ind[14], ix) = t_donr;
    TEND(P.ind[13], ix) = t_dopr;
    TEND(P.ind[6], ix) = t_dic;
    TEND(P.ind[7], ix) = A.alt_co2_use_eco ? t_dic : 0.0;
    TEND(P.ind[8], ix) = t_alk;
    TEND(P.ind[5], ix) = cur.L_o2 - oc_p;

    if (DIAG) {   // 2637-2694, 1563-1577, 1866-1868
      ST_KC(POC_FLUX_IN, inPOC.s + inPOC.h); ST_KC(POC_PROD, prodPOC); ST_KC(POC_REMIN, reminPOC);
      ST_KC(CaCO3_FLUX_IN, inCa.s + inCa.h); ST_KC(CaCO3_PROD, prodCa); ST_KC(CaCO3_REMIN, reminCa);
      ST_KC(SiO2_FLUX_IN, inSi.s + inSi.h); ST_KC(SiO2_PROD, prodSi); ST_KC(SiO2_REMIN, reminSi);
      ST_KC(dust_FLUX_IN, inDust.s + inDust.h); ST_KC(dust_REMIN, reminDust);
      ST_KC(P_iron_FLUX_IN, inFe.s + inFe.h); ST_KC(P_iron_PROD, prodFe); ST_KC(P_iron_REMIN, reminFe);
      ST_KC(calcToSed, sedCa); ST_KC(bsiToSed, sedSi); ST_KC(pocToSed, sedPOC);
      ST_KC(SedDenitrif, SED_DENITRIF * dz); ST_KC(OtherRemin, OTHER_REMIN * dz);
      ST_KC(ponToSed, (sedPOC * QN)); ST_KC(popToSed, (sedPOC * QP_ZOO_POM));
      ST_KC(dustToSed, sedDust); ST_KC(pfeToSed, sedFe);
      const double DENITRIF = cur.d0 + den_p;
      ST_KC(DENITRIF, DENITRIF); ST_KC(O2_CONSUMPTION, cur.oc0 + oc_p);
      ST_KC(Fe_scavenge, Fe_scavenge); ST_KC(Fe_scavenge_rate, Fe_scav_rate);
      // conservation integrals (1870-1938): local parts from the cell kernel + the tendencies finished here
      const double w2 = dmin(100.0e2 - zbot_above, dz);
      const double pt100 = (w2 > 0.0) ? w2 : 0.0;
      const bool shallow = zb <= 100.0e2;
      w1 = t_dic + cur.jc;
      J_C = J_C + w1 * dz + sedPOC + sedCa;
      J_C100 = J_C100 + w1 * pt100 + (shallow ? sedPOC + sedCa : 0.0);
      w1 = t_no3 + t_nh4 + t_donr + cur.jn;
      w1 = w1 + DENITRIF + SED_DENITRIF;
      J_N = J_N + w1 * dz + sedPOC * QN;
      J_N100 = J_N100 + w1 * pt100 + (shallow ? sedPOC * QN : 0.0);
      w1 = t_po4 + t_dopr + cur.jp;
      J_P = J_P + w1 * dz + sedPOC * QP_ZOO_POM;
      J_P100 = J_P100 + w1 * pt100 + (shallow ? sedPOC * QP_ZOO_POM : 0.0);
      w1 = t_sio3 + cur.jsi;
      J_Si = J_Si + w1 * dz + sedSi;
      J_Si100 = J_Si100 + w1 * pt100 + (shallow ? sedSi : 0.0);
    }
    zbot_above = zb;
    cur = nxt;
  }   // k

  if (DIAG) {
    ST_C(Jint_Ctot, J_C); ST_C(Jint_100m_Ctot, J_C100); ST_C(Jint_Ntot, J_N); ST_C(Jint_100m_Ntot, J_N100);
    ST_C(Jint_Ptot, J_P); ST_C(Jint_100m_Ptot, J_P100); ST_C(Jint_Sitot, J_Si); ST_C(Jint_100m_Sitot, J_Si100);
  }
#undef TEND
#undef ST_KC
#undef ST_C
}

// ---------------------------------------------------------------- column integrals (diagnostics only)
// Vertical integrals of the production profiles, chlorophyll over the top 100 m and the O2 minimum
// (BGC_mod.F90:1230, 1273-1277, 1816-1846, 1940-1945, 1954-1968).  None of them feeds or needs the particle
// recurrence, so they run beside the sweep on another stream.  Sums keep the reference's top-down order.
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
  const double* chl100 = A.scratch + (long long)S_CHL100 * A.splane;
  const double* o2p = A.tracers + (long long)(P.ind[5] - 1) * pl;
#pragma unroll 2
  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    const double dz = A.dz[ix];
    const double O2 = dmax(0.0, o2p[ix]);
    if (k == 0 || O2 < o2min) { o2min = O2; o2min_z = A.zmid[ix]; }
    double pC[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      pC[a] = A.z_photoC[(long long)a * pl + ix];
      z_pC[a] = z_pC[a] + dz * pC[a];
      if (at[a].Si_ind > 0) z_bSi = z_bSi + A.z_bSi_form[(long long)a * pl + ix];
      if (at[a].imp_calcifier) {
        const double w = dz * A.z_CaCO3_form[(long long)a * pl + ix];
        z_Ca[a] = z_Ca[a] + w;
        z_CaTot = z_CaTot + w;
      }
    }
    z_pCTot = z_pCTot + sum4(pC) * dz;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      z_pCNO3[a] = z_pCNO3[a] + A.z_photoC_NO3[(long long)a * pl + ix] * dz;
      z_pCNO3Tot = z_pCNO3Tot + z_pCNO3[a];   // adds the RUNNING integral (1844-1846)
    }
    z_Chl100 = z_Chl100 + chl100[ix];
  }
  if (D.O2_ZMIN) D.O2_ZMIN[c] = o2min;
  if (D.O2_ZMIN_DEPTH) D.O2_ZMIN_DEPTH[c] = o2min_z;
  if (D.tot_bSi_form) D.tot_bSi_form[c] = z_bSi;
  if (D.tot_CaCO3_form_zint) D.tot_CaCO3_form_zint[c] = z_CaTot;
  if (D.photoC_TOT_zint) D.photoC_TOT_zint[c] = z_pCTot;
  if (D.photoC_NO3_TOT_zint) D.photoC_NO3_TOT_zint[c] = z_pCNO3Tot;
  if (D.Chl_TOT_zint_100m) D.Chl_TOT_zint_100m[c] = z_Chl100;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    if (D.CaCO3_form_zint) D.CaCO3_form_zint[(long long)a * g.ldc + c] = z_Ca[a];
    if (D.photoC_zint) D.photoC_zint[(long long)a * g.ldc + c] = z_pC[a];
    if (D.photoC_NO3_zint) D.photoC_NO3_zint[(long long)a * g.ldc + c] = z_pCNO3[a];
  }
}

}  // namespace

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

void launch_fused_ecosys_cells(const BgcKernelArgs& a, const obgc_bgc_params& p, const FusedDmArgs& f, const FusedParams& fp, hipStream_t s,
                               hipEvent_t ev_cell0, hipEvent_t ev_cell1, int fuse_carbonate) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
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

void launch_bgc_zint(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  if (a.g.ncol <= 0 || !a.have_diag) return;
  const unsigned gcol = (unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol);
  hipLaunchKernelGGL(bgc_zint_kernel, dim3(gcol), dim3(kBlockCol), 0, s, a, p);
}

template <int LANES>
static void launch_sweep_lanes(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  const unsigned grid = (unsigned)((a.g.ncol + LANES - 1) / LANES);
  if (a.have_diag) hipLaunchKernelGGL((bgc_sweep_kernel<true, LANES>), dim3(grid), dim3(kBlockCol), 0, s, a, p);
  else hipLaunchKernelGGL((bgc_sweep_kernel<false, LANES>), dim3(grid), dim3(kBlockCol), 0, s, a, p);
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
