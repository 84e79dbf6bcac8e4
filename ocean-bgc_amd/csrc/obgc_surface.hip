// Air-sea exchange: BGC_SurfaceFluxes (BGC_mod.F90:2706-2957) and DMS_SurfaceFluxes
// (DMS_mod.F90:778-908).  One lane per column.  The two surface carbonate solves of a
// column (DIC and DIC_ALT_CO2 against atmCO2 / atmCO2_ALT_CO2) share one set of
// equilibrium constants (same SST, SSS; level 1 has no pressure factor) and use the
// same wavefront-ballot Newton loop as the interior kernel.
#include "obgc_co2.h"

namespace obgc {

namespace {

constexpr int kBlock = 128;

__device__ __forceinline__ double cubic(double a, double b, double c, double d, double t) {
  return a + t * (-b + t * (c + t * (-d)));   // Horner form of the Schmidt-number fits
}

__device__ __forceinline__ double o2sat_1atm(double SST, double SSS, double T0K) {   // BGC_mod.F90:3012-3083
  const double TS = log(((T0K + 25.0) - SST) / (T0K + SST));
  const double v = exp(2.00907 + TS * (3.22014 + TS * (4.05010 + TS * (4.94457 + TS * (-2.56847E-1 + TS * 3.88767)))) +
                       SSS * ((-6.24523E-3 + TS * (-7.37614E-3 + TS * (-1.03410E-2 + TS * -8.17083E-3))) + SSS * -4.88682E-7));
  return v / 0.0223916;
}

__global__ __launch_bounds__(kBlock) void bgc_surface_kernel(const SurfKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  const bool live = c < g.ncol;
  const int cc_ = live ? c : 0;
  const long long ix = g.at(0, cc_);
  const long long ldc = g.ldc;
#define SRF(slot) dmax(0.0, A.tracers[(long long)((slot)-1) * g.plane + ix])
#define ROW(arr, slot) arr[(long long)((slot)-1) * ldc + cc_]
  const double DIC = SRF(P.ind[6]), DIC_ALT = SRF(P.ind[7]), ALK = SRF(P.ind[8]), PO4 = SRF(P.ind[0]),
               SiO3 = SRF(P.ind[2]), O2 = SRF(P.ind[5]);
  const obgc_bgc_flux_diag& D = A.diag;
  const double SST = A.SST[cc_], SSS = A.SSS[cc_], patm = A.surfacePressure[cc_];

  // in-place updates of the forcing (2828-2838)
  const int fe = P.ind[4];
  if (live) {
    ROW(A.depo, fe) = ROW(A.depo, fe) * P.parm_Fe_bioavail;
    ROW(A.river, fe) = ROW(A.river, fe) * P.parm_Fe_bioavail;
    ROW(A.gas, fe) = ROW(A.gas, fe) * P.parm_Fe_bioavail;
    ROW(A.seaice, fe) = ROW(A.seaice, fe) * P.parm_Fe_bioavail;
  }
  double ice = A.iceFraction[cc_];
  if (ice < 0.0) ice = 0.0;
  if (ice > 1.0) ice = 1.0;
  if (live) A.iceFraction[cc_] = ice;
  const double xkw = 8.6e-9 * A.wind2[cc_];          // xkw_coeff, BGC_parms.F90:488-489
  const double xkw_ice = (1.0 - ice) * xkw;

  double d_pv_o2 = 0.0, d_sc_o2 = 0.0, d_o2sat = 0.0, d_xkw = 0.0;
  if (A.lcalc_o2) {                                  // 2847-2860
    d_sc_o2 = cubic(1638.0, 81.83, 1.483, 0.008004, SST);
    const double O2SAT_1atm = o2sat_1atm(SST, SSS, P.T0_Kelvin_BGC);
    d_pv_o2 = xkw_ice * sqrt(660.0 / d_sc_o2);
    d_o2sat = patm * O2SAT_1atm;
    d_xkw = xkw_ice;
    if (live) ROW(A.gas, P.ind[5]) = d_pv_o2 * (d_o2sat - O2);
  }
  if (live && A.have_diag) {
    if (D.pistonVel_O2) D.pistonVel_O2[c] = d_pv_o2;
    if (D.SCHMIDT_O2) D.SCHMIDT_O2[c] = d_sc_o2;
    if (D.O2SAT) D.O2SAT[c] = d_o2sat;
    if (D.xkw) D.xkw[c] = d_xkw;
  }

  double r_co2star[2] = {0, 0}, r_dco2star[2] = {0, 0}, r_pco2[2] = {0, 0}, r_dpco2[2] = {0, 0};
  double d_pv_co2 = 0.0, d_sc_co2 = 0.0;
  if (A.lcalc_co2) {                                 // 2866-2923
    d_sc_co2 = cubic(2073.1, 125.62, 3.6276, 0.043219, SST);
    d_pv_co2 = xkw_ice * sqrt(660.0 / d_sc_co2);
    Co3Consts cc;
    double ff = 1.0;
    comp_co3_coeffs<true>(false, 0.0, SST, SSS, cc, ff);   // level 1: no pressure terms (co2calc.F90:149-162)
    const double vol_to_mass = 1.0 / OBGC_MASS_TO_VOL;
    const double ta_m = dmax(ALK, OBGC_ALK_MIN) * vol_to_mass;
    const double pt_m = dmax(PO4, 0.0) * vol_to_mass;
    const double sit_m = dmax(SiO3, 0.0) * vol_to_mass;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      double* ph_arr = pass == 0 ? A.surface_pH : A.surface_pH_alt;
      const double dic_in = pass == 0 ? DIC : DIC_ALT;
      const double xco2_in = pass == 0 ? A.atmCO2[cc_] : A.atmCO2_alt[cc_];
      const double php = ph_arr[cc_];
      const double phlo = (php != 0.0) ? php - 0.20 : 7.0, phhi = (php != 0.0) ? php + 0.20 : 9.0;   // 2872-2878
      const double dic_m = dmax(dic_in, OBGC_DIC_MIN) * vol_to_mass;
      TalkInv tv;
      talk_setup(cc, dic_m, ta_m, pt_m, sit_m, tv);
      const double htotal = solve_htotal(live, tv, cc.kw, phlo, phhi);
      // co2calc_1point, co2calc.F90:175-208
      const double xco2 = xco2_in * 1e-6;
      const double htotal2 = htotal * htotal;
      double co2star = dic_m * htotal2 / (htotal2 + cc.k1 * htotal + cc.k1 * cc.k2);
      const double co2starair = xco2 * ff * patm;
      double dco2star = co2starair - co2star;
      const double ph = -log10(htotal);
      double pco2surf = co2star / ff;
      double dpco2 = pco2surf - xco2 * patm;
      co2star = co2star * OBGC_MASS_TO_VOL;
      dco2star = dco2star * OBGC_MASS_TO_VOL;
      pco2surf = pco2surf * 1e6;
      dpco2 = dpco2 * 1e6;
      if (live) {
        ph_arr[cc_] = ph;
        ROW(A.gas, pass == 0 ? P.ind[6] : P.ind[7]) = d_pv_co2 * dco2star;
      }
      r_co2star[pass] = co2star; r_dco2star[pass] = dco2star; r_pco2[pass] = pco2surf; r_dpco2[pass] = dpco2;
    }
  }
  if (!live) return;
  if (A.have_diag) {
    if (D.co2star) D.co2star[c] = r_co2star[0];
    if (D.dco2star) D.dco2star[c] = r_dco2star[0];
    if (D.pco2surf) D.pco2surf[c] = r_pco2[0];
    if (D.dpco2) D.dpco2[c] = r_dpco2[0];
    if (D.pistonVel_CO2) D.pistonVel_CO2[c] = d_pv_co2;
    if (D.SCHMIDT_CO2) D.SCHMIDT_CO2[c] = d_sc_co2;
    if (D.co2star_alt_co2) D.co2star_alt_co2[c] = r_co2star[1];
    if (D.dco2star_alt_co2) D.dco2star_alt_co2[c] = r_dco2star[1];
    if (D.pco2surf_alt_co2) D.pco2surf_alt_co2[c] = r_pco2[1];
    if (D.dpco2_alt_co2) D.dpco2_alt_co2[c] = r_dpco2[1];
  }
  // net flux of every tracer, then the NO3 / NH4 contribution to alkalinity (2929-2942)
  double net_alk = 0.0, net_nh4 = 0.0, net_no3 = 0.0;
  for (int n = 1; n <= OBGC_BGC_NTRACER; ++n) {
    const double v = ROW(A.depo, n) + ROW(A.gas, n) + ROW(A.river, n) + ROW(A.seaice, n);
    if (n == P.ind[8]) net_alk = v;
    else ROW(A.net, n) = v;
    if (n == P.ind[3]) net_nh4 = v;
    if (n == P.ind[1]) net_no3 = v;
  }
  ROW(A.net, P.ind[8]) = net_alk + net_nh4 - net_no3;
#undef SRF
#undef ROW
}

__global__ __launch_bounds__(kBlock) void dms_surface_kernel(const DmsSurfKernelArgs A, int i_dms, int i_dmsp) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_dms_flux_diag& D = A.diag;
  const double seaSurfaceDMS = dmax(0.0, A.tracers[(long long)(i_dms - 1) * g.plane + g.at(0, c)]);
  const double SST = A.SST[c], patm = A.surfacePressure[c];
  double ice = A.iceFraction[c];
  if (ice < 0.0) ice = 0.0;
  if (ice > 1.0) ice = 1.0;
  A.iceFraction[c] = ice;
  const double SCHMIDT_DMS = cubic(2674.0, 147.12, 3.726, 0.038, SST);          // DMS_mod.F90:915-959
  const double WIND_SPEED = sqrt(fabs(A.wind2[c])) * 0.01;
  const double XKW_W92 = 0.31 * (pow((660.0 / SCHMIDT_DMS), 0.500)) * WIND_SPEED * WIND_SPEED;
  const double XKW_LM86 = 2.85 * (pow((600.0 / SCHMIDT_DMS), 0.500)) * (WIND_SPEED - 3.6) +
                          0.612 * (pow((600.0 / SCHMIDT_DMS), 0.667));
  double xkw = 0.0;
  if (WIND_SPEED < 3.6) xkw = XKW_W92;
  if ((WIND_SPEED >= 3.6) && (WIND_SPEED < 5.6)) {   // blend of Wanninkhof 92 and Liss-Merlivat 86
    const double FLM86 = 0.5 * (WIND_SPEED - 3.6);
    const double FW92 = 1.0 - FLM86;
    xkw = FW92 * XKW_W92 + FLM86 * XKW_LM86;
  }
  if (WIND_SPEED >= 5.6) xkw = XKW_LM86;
  xkw = xkw / 3600.0;
  const double xkw_ice = (1.0 - ice) * xkw;
  const double pistonVel_DMS = xkw_ice * sqrt(660.0 / SCHMIDT_DMS);
  const double DMSSAT = patm * 0.0;                                            // DMSSAT_singleValue == 0 (1003)
  A.net[(long long)(i_dms - 1) * g.ldc + c] = pistonVel_DMS * (DMSSAT - seaSurfaceDMS);
  A.net[(long long)(i_dmsp - 1) * g.ldc + c] = 0.0;
  if (A.have_diag) {
    if (D.DMS_IFRAC) D.DMS_IFRAC[c] = ice;
    if (D.DMS_XKW) D.DMS_XKW[c] = xkw_ice;
    if (D.DMS_ATM_PRESS) D.DMS_ATM_PRESS[c] = patm;
    if (D.DMS_PV) D.DMS_PV[c] = pistonVel_DMS;
    if (D.DMS_SCHMIDT) D.DMS_SCHMIDT[c] = SCHMIDT_DMS;
    if (D.DMS_SAT) D.DMS_SAT[c] = DMSSAT;
    if (D.DMS_SURF) D.DMS_SURF[c] = seaSurfaceDMS;
    if (D.DMS_WS) D.DMS_WS[c] = WIND_SPEED;
  }
}

}  // namespace

void launch_bgc_surface(const SurfKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  const unsigned grid = (unsigned)((a.g.ncol + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(bgc_surface_kernel, dim3(grid), dim3(kBlock), 0, s, a, p);
}

void launch_dms_surface(const DmsSurfKernelArgs& a, const obgc_dms_params& p, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  const unsigned grid = (unsigned)((a.g.ncol + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(dms_surface_kernel, dim3(grid), dim3(kBlock), 0, s, a, p.ind[0], p.ind[1]);
}

}  // namespace obgc
