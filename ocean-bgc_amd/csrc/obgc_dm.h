// Cell-local arithmetic of DMS_SourceSink (DMS_mod.F90:516-761) and MACROS_SourceSink (MACROS_mod.F90:331-406) as device
// functions over VALUES, so that the stand-alone kernels (obgc_dms_macros.hip) and the fused BGC + DMS + MACROS step
// (obgc_ecosys.hip, obgc_column.hip) execute the same statements in the same order and give the same bits.
#pragma once

#include "obgc_internal.h"

namespace obgc {

// Morel & Maritorena attenuation across one level from the total chlorophyll (BGC_mod.F90:914-921 and, with the same
// constants, DMS_mod.F90:541-548): KPARdz
// The power is taken as exp(e ln w): w >= 0.02, |e ln w| < 3, so the result is within 2-3 ulp of pow()'s -- and ONE
// logarithm and ONE exponential (~100 fp64 instructions) serve both branches, where two pow() calls with different
// exponents cost ~200 each and a wavefront that straddles the 0.13224 mg/m3 break executes both.
__device__ __forceinline__ double kpar_dz_of(double total_chl, double dz) {
  const double w1 = dmax(total_chl, 0.02);
  const bool low = w1 < 0.13224;
  const double K = (low ? 0.000919 : 0.001131) * exp((low ? 0.3536 : 0.4562) * log(w1));
  return K * dz;
}

// Column-constant fuzzy-logic terms of DMS: they use the SURFACE temperature at every depth (DMS_mod.F90:504, 584-592,
// 637-640)
struct DmsColumn {
  double cyano_T, yield;
};
__device__ __forceinline__ DmsColumn dms_column_terms(const obgc_dms_params& P, double SST) {
  double T_ind = (SST - P.T_lo) / (P.T_hi - P.T_lo);
  if (T_ind <= 0.0) T_ind = 0.0;
  if (T_ind >= 1.0) T_ind = 1.0;
  DmsColumn c;
  c.cyano_T = (T_ind * (P.Max_cyano_frac - P.Min_cyano_frac)) + P.Min_cyano_frac;
  c.yield = (T_ind * (P.Max_yld - P.Min_yld)) + P.Min_yld;
  if (SST < P.T_cryo_hi && SST > P.T_cryo_lo) c.yield = 0.5;
  if (SST < -1.0) c.yield = 0.25;
  return c;
}

// One active cell of DMS_SourceSink.  Every tracer value is already clamped at zero (DMS_mod.F90:471-485); PAR_avg is the
// level mean of PAR (551).  Writes the 27 diagnostics at index ix when DIAG; returns the two tendencies (718-719).
template <bool DIAG, bool FD = false>   // FD: every diagnostic was passed, no per-member test
__device__ __forceinline__ void dms_cell_terms(const obgc_dms_params& P, const obgc_dms_diag& D, long long ix, const DmsColumn& col,
                                               double DMS, double DMSP, double zooC, double spC, double spCaCO3, double diatC,
                                               double diazC, double phaeoC, double spChl, double PAR_avg, double& t_dms,
                                               double& t_dmsp, bool live = true) {
// live = false: a lane below the sea floor that runs along (column-block kernel); the reference leaves these diagnostics alone there
#define ST(name, v) do { if (DIAG && live && (FD || D.name)) OBGC_STORE(&D.name[ix], (v)); } while (0)
  const double k_S_p = P.k_S_p_base * (P.mort + (zooC / 0.3));          // 529: hard-wired 0.3, not zooC_avg
  const double j_dms = P.j_dms_perI * PAR_avg;
  // community partition (570-612)
  double Cocco_frac = spCaCO3 / (spC + 1.00e-8);
  if (Cocco_frac > 0.4) Cocco_frac = 0.4;
  const double Cyano_frac = (1.0 - Cocco_frac) * col.cyano_T;
  const double Eukar_frac = 1.0 - Cocco_frac - Cyano_frac;
  const double diatN = P.R * diatC, phaeoN = P.R * phaeoC, coccoN = Cocco_frac * P.R * spC, cyanoN = Cyano_frac * P.R * spC,
               eukarN = Eukar_frac * P.R * spC, diazN = P.R * diazC, zooN = P.R * zooC;
  const double phytoN = diatN + coccoN + cyanoN + eukarN + diazN + phaeoN;
  // oxidant-stress up-regulation (621-628)
  double Sp_dec = (P.Sp_ref - spChl) / P.Sp_ref;
  if (Sp_dec <= 0.0) Sp_dec = 0.0;
  if (Sp_dec >= 1.0) Sp_dec = 1.0;
  double Stress_fac = 1.0 + P.Stress_mult * Sp_dec * Sp_dec;
  if (Stress_fac >= 10.0) Stress_fac = 10.0;
  // sulfur pools (647-684)
  const double diatS = P.Rs2n_diat * diatN, phaeoS = P.Rs2n_phaeo * phaeoN, coccoS = P.Rs2n_cocco * coccoN,
               cyanoS = P.Rs2n_cyano * cyanoN, eukarS = P.Rs2n_eukar * eukarN * Stress_fac, diazS = P.Rs2n_diaz * diazN;
  const double phytoS = diatS + coccoS + cyanoS + eukarS + diazS + P.G_phaeo_S * phaeoS;
  double Rs2n_zoo;
  if (phytoN > 0.0)
    Rs2n_zoo = (P.Rs2n_diat * diatN + P.G_phaeo_S * P.Rs2n_phaeo * phaeoN + P.Rs2n_cocco * coccoN + P.Rs2n_cyano * cyanoN +
                P.Rs2n_eukar * eukarN * Stress_fac + P.Rs2n_diaz * diazN) / phytoN;
  else
    Rs2n_zoo = (P.Rs2n_diat + P.Rs2n_cocco + P.Rs2n_cyano + P.Rs2n_eukar + P.Rs2n_diaz + P.Rs2n_phaeo) / 6.0;
  const double zooS = Rs2n_zoo * zooN;
  // 695: phytoN ** B_exp as exp(B_exp ln phytoN) (half of pow()'s instructions); an empty community keeps pow()'s
  // conventions for 0 ** x
  const double B_diagnosed = P.B_preexp * (phytoN > 0.0 ? exp(P.B_exp * log(phytoN)) : pow(phytoN, P.B_exp));
  // kinetics (701-719)
  const double dms_s_dmsp = col.yield * P.k_conv * DMSP;
  const double dms_r_B = P.k_S_B * B_diagnosed * DMS, dms_r_phot = j_dms * DMS, dms_r_bkgnd = P.k_bkgnd * DMS;
  const double dms_r = dms_r_B + dms_r_phot + dms_r_bkgnd;
  const double dmsp_s_phaeo = P.inject_scale * P.k_S_p_base * phaeoS, dmsp_s_nonphaeo = P.inject_scale * k_S_p * phytoS,
               dmsp_s_zoo = P.inject_scale * P.k_S_z * zooS;
  const double dmsp_s = dmsp_s_phaeo + dmsp_s_nonphaeo + dmsp_s_zoo;
  const double dmsp_r_B = P.k_conv * DMSP, dmsp_r_bkgnd = P.k_bkgnd * DMSP;
  const double dmsp_r = dmsp_r_B + dmsp_r_bkgnd;
  t_dms = dms_s_dmsp - dms_r;
  t_dmsp = dmsp_s - dmsp_r;
  if (DIAG) {   // 723-761
    ST(DMS_S_DMSP, dms_s_dmsp); ST(DMS_S_TOTAL, dms_s_dmsp); ST(DMS_R_B, dms_r_B); ST(DMS_R_PHOT, dms_r_phot);
    ST(DMS_R_BKGND, dms_r_bkgnd); ST(DMS_R_TOTAL, dms_r); ST(DMSP_S_PHAEO, dmsp_s_phaeo);
    ST(DMSP_S_NONPHAEO, dmsp_s_nonphaeo); ST(DMSP_S_ZOO, dmsp_s_zoo); ST(DMSP_S_TOTAL, dmsp_s); ST(DMSP_R_B, dmsp_r_B);
    ST(DMSP_R_BKGND, dmsp_r_bkgnd); ST(DMSP_R_TOTAL, dmsp_r); ST(Cyano_frac, Cyano_frac); ST(Cocco_frac, Cocco_frac);
    ST(Eukar_frac, Eukar_frac); ST(diatS, diatS); ST(diatN, diatN); ST(phytoN, phytoN); ST(coccoS, coccoS);
    ST(cyanoS, cyanoS); ST(eukarS, eukarS); ST(diazS, diazS); ST(phaeoS, phaeoS); ST(zooS, zooS); ST(zooCC, zooC);
    ST(RSNzoo, Rs2n_zoo);
  }
#undef ST
}

// One active cell of MACROS_SourceSink (MACROS_mod.F90:349-402); tracer values already clamped at zero (301-325).
template <bool DIAG, bool FD = false>
__device__ __forceinline__ void macros_cell_terms(const obgc_macros_params& P, const obgc_macros_diag& D, long long ix, double zooC,
                                                  double spC, double diatC, double diazC, double phaeoC, double PROT, double POLY,
                                                  double LIP, double& t_prot, double& t_poly, double& t_lip, bool live = true) {
  const double k_C_p = P.k_C_p_base * (P.mort + (zooC / P.zooC_avg));                  // 349
  const double phytoC = diatC + phaeoC + spC + diazC;                                   // 366
  const double prot_s = P.inject_scale * P.f_prot * k_C_p * phytoC;                     // 372-378
  const double poly_s = P.inject_scale * P.f_poly * k_C_p * phytoC;
  const double lip_s = P.inject_scale * P.f_lip * k_C_p * phytoC;
  const double prot_r = P.k_prot_bac * PROT, poly_r = P.k_poly_bac * POLY, lip_r = P.k_lip_bac * LIP;
  t_prot = prot_s - prot_r;
  t_poly = poly_s - poly_r;
  t_lip = lip_s - lip_r;
  if (DIAG && live) {
    if (FD || D.PROT_S_TOTAL) D.PROT_S_TOTAL[ix] = prot_s;
    if (FD || D.POLY_S_TOTAL) D.POLY_S_TOTAL[ix] = poly_s;
    if (FD || D.LIP_S_TOTAL) D.LIP_S_TOTAL[ix] = lip_s;
    if (FD || D.PROT_R_TOTAL) D.PROT_R_TOTAL[ix] = prot_r;
    if (FD || D.POLY_R_TOTAL) D.POLY_R_TOTAL[ix] = poly_r;
    if (FD || D.LIP_R_TOTAL) D.LIP_R_TOTAL[ix] = lip_r;
  }
}

}  // namespace obgc
