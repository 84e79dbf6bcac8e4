// DMS_SourceSink (DMS_mod.F90:156-770) and MACROS_SourceSink (MACROS_mod.F90:137-411).
//
// DMS: one lane per column because PAR is attenuated level by level (538-551); only the
// DMS and DMSP planes of the 14-plane tendency array are ever non-zero, the other twelve
// are zero-filled like the reference's whole-array assignment (413).
// MACROS: no vertical coupling at all -> one lane per cell, pure streaming
// (88 algorithmic bytes per active cell).
#include "obgc_internal.h"

namespace obgc {

namespace {

constexpr int kBlockCol = 64;
constexpr int kBlockCell = 256;

// DMS_SourceSink as three kernels, like the BGC light: only PAR is carried down the column (DMS_mod.F90:538-551),
// everything else is cell-local, and cell-parallel kernels are indifferent to the layout (their lanes follow the
// contiguous axis), which is what makes the reference layout as fast as the native one.
//   dms_att_kernel   1 lane / cell     exp(-KPARdz), KPARdz
//   dms_scan_kernel  1 lane / column   PAR entering each level (one multiply per level)
//   dms_cell_kernel  1 lane / cell     the kinetics, 14 tendencies (12 of them zero), 27 diagnostics
#define TRD(j, ix) dmax(0.0, A.tracers[(long long)(P.ind[j] - 1) * pl + (ix)])

__global__ __launch_bounds__(kBlockCell) void dms_att_kernel(const DmsKernelArgs A, const obgc_dms_params P) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  if (k >= A.kmax[c]) return;
  const long long ix = g.at(k, c), pl = g.plane;
  const double spChl = TRD(10, ix), diatChl = TRD(11, ix), diazChl = TRD(12, ix), phaeoChl = TRD(13, ix);
  // light (538-551); the UV branch (531-536) produces nothing that is used
  const double totalChl = spChl + diatChl + diazChl + phaeoChl;
  const double work = dmax(totalChl, 0.02);
  double KPARdz = (work < 0.13224) ? 0.000919 * pow(work, 0.3536) : 0.001131 * pow(work, 0.4562);
  KPARdz = KPARdz * A.dz[ix];
  A.s_kpardz[ix] = KPARdz;
  A.s_att[ix] = exp(-KPARdz);
}

// `att` / `par_in` are column-fastest planes (dense, leading dimension ncol) when cf != 0, else in the view's own index space
__global__ __launch_bounds__(kBlockCol) void dms_scan_kernel(GridView g, const int* __restrict__ kmaxv, const double* __restrict__ sw,
                                                            const double* __restrict__ att, double* __restrict__ par_in, double f_qsw_par,
                                                            int cf) {
  const int c = blockIdx.x * kBlockCol + threadIdx.x;
  if (c >= g.ncol) return;
  int kmax = kmaxv[c];
  if (kmax > g.nlev) kmax = g.nlev;
  double PAR = dmax(0.0, sw[c]) * f_qsw_par;   // 506-507
  for (int k = 0; k < kmax; ++k) {
    const long long ix = cf ? (long long)k * g.ncol + c : g.at(k, c);
    par_in[ix] = PAR;
    PAR = PAR * att[ix];
  }
}

template <bool DIAG>
__global__ __launch_bounds__(kBlockCell) void dms_cell_kernel(const DmsKernelArgs A, const obgc_dms_params P) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  const long long ix = g.at(k, c), pl = g.plane;
  const obgc_dms_diag& D = A.diag;
#define ST(name, v) do { if (DIAG && D.name) D.name[ix] = (v); } while (0)
  if (k >= A.kmax[c]) {   // the reference zero-fills the tendencies of the whole array first
#pragma unroll
    for (int n = 0; n < OBGC_DMS_NTRACER; ++n) A.tend[(long long)n * pl + ix] = 0.0;
    return;
  }
  const double SST = A.SST[c];
  // column-constant fuzzy-logic terms: they use the SURFACE temperature at every depth (504, 584-592, 637-640)
  double T_ind = (SST - P.T_lo) / (P.T_hi - P.T_lo);
  if (T_ind <= 0.0) T_ind = 0.0;
  if (T_ind >= 1.0) T_ind = 1.0;
  const double cyano_T = (T_ind * (P.Max_cyano_frac - P.Min_cyano_frac)) + P.Min_cyano_frac;
  double yield = (T_ind * (P.Max_yld - P.Min_yld)) + P.Min_yld;
  if (SST < P.T_cryo_hi && SST > P.T_cryo_lo) yield = 0.5;
  if (SST < -1.0) yield = 0.25;
  {
    const double DMS = TRD(0, ix), DMSP = TRD(1, ix), zooC = TRD(4, ix), spC = TRD(5, ix), spCaCO3 = TRD(6, ix),
                 diatC = TRD(7, ix), diazC = TRD(8, ix), phaeoC = TRD(9, ix), spChl = TRD(10, ix);
    const double PAR_in = A.s_parin[ix], att = A.s_att[ix], KPARdz = A.s_kpardz[ix];
    const double PAR_avg = PAR_in * (1.0 - att) / KPARdz;
    const double k_S_p = P.k_S_p_base * (P.mort + (zooC / 0.3));          // 529: hard-wired 0.3, not zooC_avg
    const double j_dms = P.j_dms_perI * PAR_avg;
    // community partition (570-612)
    double Cocco_frac = spCaCO3 / (spC + 1.00e-8);
    if (Cocco_frac > 0.4) Cocco_frac = 0.4;
    const double Cyano_frac = (1.0 - Cocco_frac) * cyano_T;
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
    const double B_diagnosed = P.B_preexp * pow(phytoN, P.B_exp);          // 695
    // kinetics (701-719)
    const double dms_s_dmsp = yield * P.k_conv * DMSP;
    const double dms_r_B = P.k_S_B * B_diagnosed * DMS, dms_r_phot = j_dms * DMS, dms_r_bkgnd = P.k_bkgnd * DMS;
    const double dms_r = dms_r_B + dms_r_phot + dms_r_bkgnd;
    const double dmsp_s_phaeo = P.inject_scale * P.k_S_p_base * phaeoS, dmsp_s_nonphaeo = P.inject_scale * k_S_p * phytoS,
                 dmsp_s_zoo = P.inject_scale * P.k_S_z * zooS;
    const double dmsp_s = dmsp_s_phaeo + dmsp_s_nonphaeo + dmsp_s_zoo;
    const double dmsp_r_B = P.k_conv * DMSP, dmsp_r_bkgnd = P.k_bkgnd * DMSP;
    const double dmsp_r = dmsp_r_B + dmsp_r_bkgnd;
#pragma unroll
    for (int n = 0; n < OBGC_DMS_NTRACER; ++n) A.tend[(long long)n * pl + ix] = 0.0;
    A.tend[(long long)(P.ind[0] - 1) * pl + ix] = dms_s_dmsp - dms_r;
    A.tend[(long long)(P.ind[1] - 1) * pl + ix] = dmsp_s - dmsp_r;
    if (DIAG) {   // 723-761
      ST(DMS_S_DMSP, dms_s_dmsp); ST(DMS_S_TOTAL, dms_s_dmsp); ST(DMS_R_B, dms_r_B); ST(DMS_R_PHOT, dms_r_phot);
      ST(DMS_R_BKGND, dms_r_bkgnd); ST(DMS_R_TOTAL, dms_r); ST(DMSP_S_PHAEO, dmsp_s_phaeo);
      ST(DMSP_S_NONPHAEO, dmsp_s_nonphaeo); ST(DMSP_S_ZOO, dmsp_s_zoo); ST(DMSP_S_TOTAL, dmsp_s); ST(DMSP_R_B, dmsp_r_B);
      ST(DMSP_R_BKGND, dmsp_r_bkgnd); ST(DMSP_R_TOTAL, dmsp_r); ST(Cyano_frac, Cyano_frac); ST(Cocco_frac, Cocco_frac);
      ST(Eukar_frac, Eukar_frac); ST(diatS, diatS); ST(diatN, diatN); ST(phytoN, phytoN); ST(coccoS, coccoS);
      ST(cyanoS, cyanoS); ST(eukarS, eukarS); ST(diazS, diazS); ST(phaeoS, phaeoS); ST(zooS, zooS); ST(zooCC, zooC);
      ST(RSNzoo, Rs2n_zoo);
    }
  }
#undef ST
}

#undef TRD

template <bool DIAG>
__global__ __launch_bounds__(kBlockCell) void macros_kernel(const MacrosKernelArgs A, const obgc_macros_params P) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlockCell + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  const long long ix = g.at(k, c), pl = g.plane;
  const obgc_macros_diag& D = A.diag;
#pragma unroll
  for (int n = 0; n < OBGC_MACROS_NTRACER; ++n) A.tend[(long long)n * pl + ix] = 0.0;   // MACROS_mod.F90:267
  if (k >= A.kmax[c]) return;
#define TRM(j) dmax(0.0, A.tracers[(long long)(P.ind[j] - 1) * pl + ix])
  const double zooC = TRM(3);
  const double k_C_p = P.k_C_p_base * (P.mort + (zooC / P.zooC_avg));                  // 349
  const double phytoC = TRM(5) + TRM(7) + TRM(4) + TRM(6);                             // diat + phaeo + sp + diaz, 366
  const double prot_s = P.inject_scale * P.f_prot * k_C_p * phytoC;                     // 372-378
  const double poly_s = P.inject_scale * P.f_poly * k_C_p * phytoC;
  const double lip_s = P.inject_scale * P.f_lip * k_C_p * phytoC;
  const double prot_r = P.k_prot_bac * TRM(0), poly_r = P.k_poly_bac * TRM(1), lip_r = P.k_lip_bac * TRM(2);
  A.tend[(long long)(P.ind[0] - 1) * pl + ix] = prot_s - prot_r;
  A.tend[(long long)(P.ind[1] - 1) * pl + ix] = poly_s - poly_r;
  A.tend[(long long)(P.ind[2] - 1) * pl + ix] = lip_s - lip_r;
  if (DIAG) {
    if (D.PROT_S_TOTAL) D.PROT_S_TOTAL[ix] = prot_s;
    if (D.POLY_S_TOTAL) D.POLY_S_TOTAL[ix] = poly_s;
    if (D.LIP_S_TOTAL) D.LIP_S_TOTAL[ix] = lip_s;
    if (D.PROT_R_TOTAL) D.PROT_R_TOTAL[ix] = prot_r;
    if (D.POLY_R_TOTAL) D.POLY_R_TOTAL[ix] = poly_r;
    if (D.LIP_R_TOTAL) D.LIP_R_TOTAL[ix] = lip_r;
  }
#undef TRM
}

}  // namespace

void launch_dms(const DmsKernelArgs& a, const obgc_dms_params& p, hipStream_t s) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  const unsigned gcell = (unsigned)((ncell + kBlockCell - 1) / kBlockCell), gcol = (unsigned)((a.g.ncol + kBlockCol - 1) / kBlockCol);
  hipLaunchKernelGGL(dms_att_kernel, dim3(gcell), dim3(kBlockCell), 0, s, a, p);
  if (a.g.kfast && a.w_att) {   // reference layout: the scan runs on column-fastest twins of its two planes
    const long long ld = a.g.sc;
    launch_transpose_k2c(a.s_att, a.w_att, a.g.nlev, a.g.ncol, ld, 0, a.g.ncol, 0, 1, s);
    hipLaunchKernelGGL(dms_scan_kernel, dim3(gcol), dim3(kBlockCol), 0, s, a.g, a.kmax, a.sw, a.w_att, a.w_parin, p.f_qsw_par_DMS, 1);
    launch_transpose_c2k(a.w_parin, a.s_parin, a.g.nlev, a.g.ncol, a.g.ncol, 0, ld, 0, 1, s);
  } else {
    hipLaunchKernelGGL(dms_scan_kernel, dim3(gcol), dim3(kBlockCol), 0, s, a.g, a.kmax, a.sw, a.s_att, a.s_parin, p.f_qsw_par_DMS, 0);
  }
  if (a.have_diag) hipLaunchKernelGGL(dms_cell_kernel<true>, dim3(gcell), dim3(kBlockCell), 0, s, a, p);
  else hipLaunchKernelGGL(dms_cell_kernel<false>, dim3(gcell), dim3(kBlockCell), 0, s, a, p);
}

void launch_macros(const MacrosKernelArgs& a, const obgc_macros_params& p, hipStream_t s) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  const unsigned grid = (unsigned)((ncell + kBlockCell - 1) / kBlockCell);
  if (a.have_diag) hipLaunchKernelGGL(macros_kernel<true>, dim3(grid), dim3(kBlockCell), 0, s, a, p);
  else hipLaunchKernelGGL(macros_kernel<false>, dim3(grid), dim3(kBlockCell), 0, s, a, p);
}

}  // namespace obgc
