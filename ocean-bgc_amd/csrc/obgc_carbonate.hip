// Interior carbonate chemistry of BGC_SourceSink as a cell-parallel kernel.
//
// In the reference this work sits inside the sequential level loop of every column
// (BGC_mod.F90:935-1032 -> co2calc.F90:214-316, 1096-1238) although no level depends
// on another: two Newton solves and ~35 transcendentals per cell, about 70 % of the
// flops of the whole routine.  Here every active (level, column) cell
// is one lane, so even the 100k-column config exposes 5.7 M independent lanes, and
// the vertical recurrences are left to the much lighter column-sweep kernel.
//
// What is shared per cell instead of recomputed like the reference does:
//  * the equilibrium constants are evaluated once and serve both solves (the "ALT_CO2"
//    solve is fed the same T, S, depth, DIC, ALK, PO4, SiO3: BGC_mod.F90:974-977 passes
//    DIC_loc, not DIC_ALT_CO2_loc);
//  * when both brackets are bit-identical the second solve must return the first
//    solve's bits, so it is skipped wave-wide unless some lane's brackets differ.
#include "obgc_co2.h"

namespace obgc {

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void bgc_carbonate_kernel(BgcKernelArgs A, int i_dic, int i_alk, int i_po4,
                                                              int i_sio3) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  // lanes run along the contiguous axis of the caller's layout
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  const bool inside = idx < ncell;
  const int kmax = inside ? A.kmax[c] : 0;
  const bool live = inside && k < kmax;
  const long long ix = inside ? g.at(k, c) : 0;

  double T = 10.0, S = 35.0, depth = 0.0, dic = 2000.0, alk = 2300.0, po4 = 0.0, sio3 = 0.0;
  double php = 0.0, pha = 0.0;
  if (live) {
    T = A.temp[ix];
    S = A.salt[ix];
    depth = A.zmid[ix] * 0.01;                                   // cm -> m, BGC_mod.F90:950
    dic = dmax(0.0, A.tracers[(long long)(i_dic - 1) * g.plane + ix]);   // clamps of 747-752
    alk = dmax(0.0, A.tracers[(long long)(i_alk - 1) * g.plane + ix]);
    po4 = dmax(0.0, A.tracers[(long long)(i_po4 - 1) * g.plane + ix]);
    sio3 = dmax(0.0, A.tracers[(long long)(i_sio3 - 1) * g.plane + ix]);
    php = A.ph_prev[ix];
    pha = A.ph_prev_alt[ix];
  }
  const bool deep = k > 0;   // reference level index k > 1
  if (__ballot(live) == 0ull) {
    // whole wavefront has nothing to solve (land / below the sea floor): only the zero fill
    if (inside && A.have_diag) {
      const obgc_bgc_diag& D = A.diag;
      if (D.CO3) D.CO3[ix] = 0.0;
      if (D.HCO3) D.HCO3[ix] = 0.0;
      if (D.H2CO3) D.H2CO3[ix] = 0.0;
      if (D.pH_3D) D.pH_3D[ix] = 0.0;
      if (D.CO3_ALT_CO2) D.CO3_ALT_CO2[ix] = 0.0;
      if (D.HCO3_ALT_CO2) D.HCO3_ALT_CO2[ix] = 0.0;
      if (D.H2CO3_ALT_CO2) D.H2CO3_ALT_CO2[ix] = 0.0;
      if (D.pH_3D_ALT_CO2) D.pH_3D_ALT_CO2[ix] = 0.0;
      if (D.co3_sat_calc) D.co3_sat_calc[ix] = 0.0;
      if (D.co3_sat_arag) D.co3_sat_arag[ix] = 0.0;
    }
    return;
  }

  Co3Consts cc;
  double ff_unused;
  comp_co3_coeffs<false>(deep, depth, T, S, cc, ff_unused);

  // comp_htotal's clamps and conversion to mol/kg (co2calc.F90:843-846)
  const double vol_to_mass = 1.0 / OBGC_MASS_TO_VOL;
  const double dic_m = dmax(dic, OBGC_DIC_MIN) * vol_to_mass;
  const double ta_m = dmax(alk, OBGC_ALK_MIN) * vol_to_mass;
  const double pt_m = dmax(po4, 0.0) * vol_to_mass;
  const double sit_m = dmax(sio3, 0.0) * vol_to_mass;
  TalkInv tv;
  talk_setup(cc, dic_m, ta_m, pt_m, sit_m, tv);

  // brackets: previous pH +- del_ph, or [6, 9] when there is none (BGC_mod.F90:943-949, 965-971)
  const double lo1 = (php != 0.0) ? php - 0.20 : 6.0, hi1 = (php != 0.0) ? php + 0.20 : 9.0;
  const double lo2 = (pha != 0.0) ? pha - 0.20 : 6.0, hi2 = (pha != 0.0) ? pha + 0.20 : 9.0;

  const double h1 = solve_htotal(live, tv, cc.kw, lo1, hi1);
  double h2 = h1;
  const bool differs = live && (lo2 != lo1 || hi2 != hi1);
  if (__ballot(differs) != 0ull) {
    const double h2s = solve_htotal(differs, tv, cc.kw, lo2, hi2);
    if (differs) h2 = h2s;
  }

  double sat_calc, sat_arag;
  comp_co3_sat_vals(deep, depth, T, S, sat_calc, sat_arag);

  if (!inside) return;
  if (live) {
    // species from [H+] (co2calc.F90:301-314)
    const double ph1 = -log10(h1);
    const double ph2 = (h2 == h1) ? ph1 : -log10(h2);
    A.ph_prev[ix] = ph1;
    A.ph_prev_alt[ix] = ph2;
    const double k12 = cc.k1 * cc.k2;
    double hh = h1 * h1;
    double denom = 1.0 / (hh + cc.k1 * h1 + k12);
    const double H2CO3 = (dic_m * hh * denom) * OBGC_MASS_TO_VOL;
    const double HCO3 = (dic_m * cc.k1 * h1 * denom) * OBGC_MASS_TO_VOL;
    const double CO3 = (dic_m * cc.k1 * cc.k2 * denom) * OBGC_MASS_TO_VOL;
    if (A.co3_out) A.co3_out[ix] = CO3;
    if (A.sat_calc_out) A.sat_calc_out[ix] = sat_calc;
    if (A.sat_arag_out) A.sat_arag_out[ix] = sat_arag;
    if (A.have_diag) {
      const obgc_bgc_diag& D = A.diag;
      if (D.CO3 && D.CO3 != A.co3_out) D.CO3[ix] = CO3;   // may double as the zsat scan's input
      if (D.HCO3) D.HCO3[ix] = HCO3;
      if (D.H2CO3) D.H2CO3[ix] = H2CO3;
      if (D.pH_3D) D.pH_3D[ix] = ph1;
      if (D.co3_sat_calc && D.co3_sat_calc != A.sat_calc_out) D.co3_sat_calc[ix] = sat_calc;
      if (D.co3_sat_arag && D.co3_sat_arag != A.sat_arag_out) D.co3_sat_arag[ix] = sat_arag;
      if (D.CO3_ALT_CO2 || D.HCO3_ALT_CO2 || D.H2CO3_ALT_CO2) {
        hh = h2 * h2;
        denom = 1.0 / (hh + cc.k1 * h2 + k12);
        if (D.H2CO3_ALT_CO2) D.H2CO3_ALT_CO2[ix] = (dic_m * hh * denom) * OBGC_MASS_TO_VOL;
        if (D.HCO3_ALT_CO2) D.HCO3_ALT_CO2[ix] = (dic_m * cc.k1 * h2 * denom) * OBGC_MASS_TO_VOL;
        if (D.CO3_ALT_CO2) D.CO3_ALT_CO2[ix] = (dic_m * cc.k1 * cc.k2 * denom) * OBGC_MASS_TO_VOL;
      }
      if (D.pH_3D_ALT_CO2) D.pH_3D_ALT_CO2[ix] = ph2;
    }
  } else if (A.have_diag) {
    // the reference zero-fills every diagnostic first (BGC_mod.F90:649-658); PH_PREV_* of
    // inactive cells are left untouched
    const obgc_bgc_diag& D = A.diag;
    if (D.CO3) D.CO3[ix] = 0.0;
    if (D.HCO3) D.HCO3[ix] = 0.0;
    if (D.H2CO3) D.H2CO3[ix] = 0.0;
    if (D.pH_3D) D.pH_3D[ix] = 0.0;
    if (D.CO3_ALT_CO2) D.CO3_ALT_CO2[ix] = 0.0;
    if (D.HCO3_ALT_CO2) D.HCO3_ALT_CO2[ix] = 0.0;
    if (D.H2CO3_ALT_CO2) D.H2CO3_ALT_CO2[ix] = 0.0;
    if (D.pH_3D_ALT_CO2) D.pH_3D_ALT_CO2[ix] = 0.0;
    if (D.co3_sat_calc) D.co3_sat_calc[ix] = 0.0;
    if (D.co3_sat_arag) D.co3_sat_arag[ix] = 0.0;
  }
}

// Saturation horizons ZSATCALC / ZSATARAG: a first-crossing search down each column
// (BGC_mod.F90:1003-1032) over the CO3 and saturation profiles the cell kernel left in
// memory.  One lane per column; three reads per cell.
__global__ __launch_bounds__(kBlock) void bgc_zsat_kernel(BgcKernelArgs A) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  const int kmax = min(A.kmax[c], g.nlev);
  double zc = 0.0, za = 0.0;
  if (kmax >= 1) {
    double anom_c = 0.0, anom_a = 0.0, zprev = 0.0;
    for (int k = 0; k < kmax; ++k) {
      const long long ix = g.at(k, c);
      const double CO3 = A.co3_out[ix], sc = A.sat_calc_out[ix], sa = A.sat_arag_out[ix];
      const double zmid = A.zmid[ix];
      if (k == 0) {
        zc = (CO3 > sc) ? -1.0 : 0.0;
        za = (CO3 > sa) ? -1.0 : 0.0;
      } else {
        const double w4 = zprev + (zmid - zprev);                       // BGC_mod.F90:1009
        if (zc == -1.0 && CO3 <= sc) zc = w4 * anom_c / (anom_c - (CO3 - sc));
        if (za == -1.0 && CO3 <= sa) za = w4 * anom_a / (anom_a - (CO3 - sa));
        if (k == kmax - 1) {
          const double zb = A.zbot[ix];
          if (zc == -1.0) zc = zb;
          if (za == -1.0) za = zb;
        }
      }
      anom_c = CO3 - sc;
      anom_a = CO3 - sa;
      zprev = zmid;
    }
  }
  if (D.zsatcalc) D.zsatcalc[c] = zc;
  if (D.zsatarag) D.zsatarag[c] = za;
}

}  // namespace

void launch_bgc_carbonate(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  const long long ncell = (long long)a.g.nlev * a.g.ncol;
  if (ncell <= 0) return;
  const unsigned grid = (unsigned)((ncell + kBlock - 1) / kBlock);
  // slots in OBGC_BGC_TRACERS order: po4=0 sio3=2 dic=6 alk=8
  hipLaunchKernelGGL(bgc_carbonate_kernel, dim3(grid), dim3(kBlock), 0, s, a, p.ind[6], p.ind[8], p.ind[0], p.ind[2]);
}

void launch_bgc_zsat(const BgcKernelArgs& a, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  const unsigned grid = (unsigned)((a.g.ncol + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(bgc_zsat_kernel, dim3(grid), dim3(kBlock), 0, s, a);
}

}  // namespace obgc
