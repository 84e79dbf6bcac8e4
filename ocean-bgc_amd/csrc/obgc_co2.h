// Device-side carbonate chemistry: equilibrium constants, alkalinity residual and the
// safeguarded Newton / bisection solve for [H+].
//
// Behavioural contract = co2calc.F90 of the reference (comp_co3_coeffs 320-777,
// comp_htotal 781-868, drtsafe_row 872-997, talk_row 1001-1092, comp_co3_sat_vals
// 1096-1238).  Unlike the reference, which parks the eleven pressure-corrected
// constants and the four converted totals in module-level SAVE arrays
// (co2calc.F90:65-67, hence not thread-safe), everything here lives in the registers
// of one lane, and the per-cell loop invariants of the residual (products of
// constants) are hoisted out of the iteration.
#pragma once

#include "obgc_internal.h"

namespace obgc {

// correctly rounded values of the reference's LOG(c10) and LOG(1e-2) (co2calc.F90:424, 472)
#define OBGC_LN10 0x1.26bb1bbb55516p+1
#define OBGC_LN1EM2 (-0x1.26bb1bbb55515p+2)

struct Co3Consts {
  double k1, k2;                                   // NOT pressure corrected (co2calc.F90:478, 507)
  double kw, kb, ks, kf, k1p, k2p, k3p, ksi, bt, st, ft;
};

// POP reference pressure in bar at depth [m] (co2calc.F90:371-372)
__device__ __forceinline__ double press_bar_of(double depth) {
  return 0.059808 * (exp(-0.025 * depth) - 1.0) + 0.100766 * depth + 2.28405e-7 * (depth * depth);
}

__device__ __forceinline__ double millero_kfac(double deltaV, double Kappa, double press_bar, double invRtk) {
  double lnKfac = (-deltaV + 0.5 * Kappa * press_bar) * press_bar * invRtk;
  return exp(lnKfac);
}

// deep = (level index > 1): apply the pressure factors.  WANT_GAS also returns the
// Weiss-Price fugacity factor ff (surface CO2 flux only; interior cells never use k0/ff).
template <bool WANT_GAS>
__device__ __forceinline__ void comp_co3_coeffs(bool deep, double depth, double temp, double salt, Co3Consts& c,
                                                double& ff) {
  const double p001 = 0.001;
  const double salt_lim = dmax(salt, 0.1);
  const double tk = 273.15 + temp;
  const double tk100 = tk * 1e-2;
  const double tk1002 = tk100 * tk100;
  const double invtk = 1.0 / tk;
  const double dlogtk = log(tk);
  const double invRtk = (1.0 / 83.1451) * invtk;
  const double is = 19.924 * salt_lim / (1000.0 - 1.005 * salt_lim);
  const double is2 = is * is;
  const double sqrtis = sqrt(is);
  const double sqrts = sqrt(salt_lim);
  const double s2 = salt_lim * salt_lim;
  const double scl = salt_lim / 1.80655;
  const double log_1_m_1p005em3_s = log(1.0 - 0.001005 * salt_lim);
  double arg;

  if (WANT_GAS) {  // co2calc.F90:423-431
    arg = -162.8301 + 218.2968 / tk100 + 90.9241 * (dlogtk + OBGC_LN1EM2) - 1.47696 * tk1002 +
          salt_lim * (.025695 - .025225 * tk100 + 0.0049867 * tk1002);
    ff = exp(arg);
  }
  // k1, k2 on the total scale (461-478, 492-507); the reference hands the solver the
  // values taken BEFORE its pressure factor, so no factor is applied here
  arg = 3633.86 * invtk - 61.2172 + 9.67770 * dlogtk - 0.011555 * salt_lim + 0.0001152 * s2;
  c.k1 = exp(-OBGC_LN10 * arg);
  arg = 471.78 * invtk + 25.9290 - 3.16967 * dlogtk - 0.01781 * salt_lim + 0.0001122 * s2;
  c.k2 = exp(-OBGC_LN10 * arg);

  const double press_bar = deep ? press_bar_of(depth) : 0.0;
  const double bdV = -29.48 + (0.1622 - 0.002608 * temp) * temp;  // boric-acid deltaV, shared by kb and ksi

  // kb (529-551)
  arg = (-8966.90 - 2890.53 * sqrts - 77.942 * salt_lim + 1.728 * salt_lim * sqrts - 0.0996 * s2) * invtk +
        (148.0248 + 137.1942 * sqrts + 1.62142 * salt_lim) + (-24.4344 - 25.085 * sqrts - 0.2474 * salt_lim) * dlogtk +
        0.053105 * sqrts * tk;
  c.kb = exp(arg);
  // k1p (560-580)
  arg = -4576.752 * invtk + 115.525 - 18.453 * dlogtk + (-106.736 * invtk + 0.69171) * sqrts +
        (-0.65643 * invtk - 0.01844) * salt_lim;
  c.k1p = exp(arg);
  // k2p (589-609)
  arg = -8814.715 * invtk + 172.0883 - 27.927 * dlogtk + (-160.340 * invtk + 1.3566) * sqrts +
        (0.37335 * invtk - 0.05778) * salt_lim;
  c.k2p = exp(arg);
  // k3p (618-637)
  arg = -3070.75 * invtk - 18.141 + (17.27039 * invtk + 2.81197) * sqrts + (-44.99486 * invtk - 0.09984) * salt_lim;
  c.k3p = exp(arg);
  // ksi (647-669)
  arg = -8904.2 * invtk + 117.385 - 19.334 * dlogtk + (-458.79 * invtk + 3.5913) * sqrtis +
        (188.74 * invtk - 1.5998) * is + (-12.1652 * invtk + 0.07871) * is2 + log_1_m_1p005em3_s;
  c.ksi = exp(arg);
  // kw (681-700)
  arg = -13847.26 * invtk + 148.9652 - 23.6521 * dlogtk + (118.67 * invtk - 5.977 + 1.0495 * dlogtk) * sqrts -
        0.01615 * salt_lim;
  c.kw = exp(arg);
  // ks (709-731)
  arg = -4276.1 * invtk + 141.328 - 23.093 * dlogtk + (-13856.0 * invtk + 324.57 - 47.986 * dlogtk) * sqrtis +
        (35474.0 * invtk - 771.54 + 114.723 * dlogtk) * is - 2698.0 * invtk * is * sqrtis + 1776.0 * invtk * is2 +
        log_1_m_1p005em3_s;
  c.ks = exp(arg);
  if (deep) {
    c.kb = c.kb * millero_kfac(bdV, -2.84 * p001, press_bar, invRtk);
    c.k1p = c.k1p * millero_kfac(-14.51 + (0.1211 - 0.000321 * temp) * temp, (-2.67 + 0.0427 * temp) * p001, press_bar, invRtk);
    c.k2p = c.k2p * millero_kfac(-23.12 + (0.1758 - 0.002647 * temp) * temp, (-5.15 + 0.09 * temp) * p001, press_bar, invRtk);
    c.k3p = c.k3p * millero_kfac(-26.57 + (0.202 - 0.003042 * temp) * temp, (-4.08 + 0.0714 * temp) * p001, press_bar, invRtk);
    c.ksi = c.ksi * millero_kfac(bdV, -2.84 * p001, press_bar, invRtk);
    c.kw = c.kw * millero_kfac(-20.02 + (0.1119 - 0.001409 * temp) * temp, (-5.13 + 0.0794 * temp) * p001, press_bar, invRtk);
    c.ks = c.ks * millero_kfac(-18.03 + (0.0466 + 0.000316 * temp) * temp, (-4.53 + 0.09 * temp) * p001, press_bar, invRtk);
  }
  // kf needs the (corrected) ks (740-764)
  const double log_1_p_tot_sulfate_div_ks = log(1.0 + (0.1400 / 96.062) * (scl) / c.ks);
  arg = 1590.2 * invtk - 12.641 + 1.525 * sqrtis + log_1_m_1p005em3_s + log_1_p_tot_sulfate_div_ks;
  c.kf = exp(arg);
  if (deep)
    c.kf = c.kf * millero_kfac(-9.78 - (0.009 + 0.000942 * temp) * temp, (-3.91 + 0.054 * temp) * p001, press_bar, invRtk);
  // total borate, sulfate, fluoride (773-775)
  c.bt = 0.000232 / 10.811 * scl;
  c.st = 0.14 / 96.062 * scl;
  c.ft = 0.000067 / 18.9984 * scl;
}

// Loop invariants of the alkalinity residual for one cell: products of constants and
// totals in the association order the reference evaluates them in (talk_row,
// co2calc.F90:1036-1090), so hoisting changes no rounding.
struct TalkInv {
  double k1, k1p, kb, ksi, kf, st, ft, ta;
  double k12, k12p, k123p, two_k1p;       // k1*k2, k1p*k2p, k12p*k3p, 2*k1p
  double k1_dic, two_dic_k12;             // k1*dic, (2*dic)*k12
  double bt_kb, sit_ksi;                  // bt*kb, sit*ksi
  double pt, pt_k12p, two_pt_k123p;       // pt, pt*k12p, (2*pt)*k123p
  double c_r, c_ks;                       // 1/(1+st/ks), (1+st/ks)*ks
};

__device__ __forceinline__ void talk_setup(const Co3Consts& c, double dic, double ta, double pt, double sit, TalkInv& v) {
  v.k1 = c.k1; v.k1p = c.k1p; v.kb = c.kb; v.ksi = c.ksi; v.kf = c.kf; v.st = c.st; v.ft = c.ft; v.ta = ta;
  v.k12 = c.k1 * c.k2;
  v.k12p = c.k1p * c.k2p;
  v.k123p = v.k12p * c.k3p;
  v.two_k1p = 2.0 * c.k1p;
  v.k1_dic = c.k1 * dic;
  v.two_dic_k12 = 2.0 * dic * v.k12;
  v.bt_kb = c.bt * c.kb;
  v.sit_ksi = sit * c.ksi;
  v.pt = pt;
  v.pt_k12p = pt * v.k12p;
  v.two_pt_k123p = 2.0 * pt * v.k123p;
  const double cc = 1.0 + c.st / c.ks;
  v.c_r = 1.0 / cc;
  v.c_ks = cc * c.ks;
  (void)c.kw;
}

// alkalinity residual fn(x) and d fn / dx  (talk_row)
__device__ __forceinline__ void talk_row(const TalkInv& v, double kw, double x1, double& fn, double& df) {
  const double x1_r = 1.0 / x1;
  const double x2 = x1 * x1;
  const double x2_r = x1_r * x1_r;
  const double x3 = x2 * x1;
  const double a = x3 + v.k1p * x2 + v.k12p * x1 + v.k123p;
  const double a_r = 1.0 / a;
  const double a2_r = a_r * a_r;
  const double da = 3.0 * x2 + v.two_k1p * x1 + v.k12p;
  const double b = x2 + v.k1 * x1 + v.k12;
  const double b_r = 1.0 / b;
  const double b2_r = b_r * b_r;
  const double db = 2.0 * x1 + v.k1;
  const double kb_p_x1_r = 1.0 / (v.kb + x1);
  const double ksi_p_x1_r = 1.0 / (v.ksi + x1);
  const double c1_p_c_ks_x1_r_r = 1.0 / (1.0 + v.c_ks * x1_r);
  const double c1_p_kf_x1_r_r = 1.0 / (1.0 + v.kf * x1_r);

  fn = v.k1_dic * x1 * b_r + v.two_dic_k12 * b_r + v.bt_kb * kb_p_x1_r + kw * x1_r + v.pt_k12p * x1 * a_r +
       v.two_pt_k123p * a_r + v.sit_ksi * ksi_p_x1_r - x1 * v.c_r - v.st * c1_p_c_ks_x1_r_r - v.ft * c1_p_kf_x1_r_r -
       v.pt * x3 * a_r - v.ta;

  df = v.k1_dic * (b - x1 * db) * b2_r - v.two_dic_k12 * db * b2_r - v.bt_kb * kb_p_x1_r * kb_p_x1_r - kw * x2_r +
       (v.pt_k12p * (a - x1 * da)) * a2_r - v.two_pt_k123p * da * a2_r - v.sit_ksi * ksi_p_x1_r * ksi_p_x1_r -
       1.0 * v.c_r - v.st * c1_p_c_ks_x1_r_r * c1_p_c_ks_x1_r_r * (v.c_ks * x2_r) -
       v.ft * c1_p_kf_x1_r_r * c1_p_kf_x1_r_r * v.kf * x2_r - v.pt * x2 * (3.0 * a - x1 * da) * a2_r;
}

// Unit conversion + clamps of comp_htotal (co2calc.F90:836-846)
#define OBGC_DIC_MIN 0x1.63796ac9dfd13p+2 /* salt_min/35*1944 */
#define OBGC_ALK_MIN 0x1.96db6db6db6dbp+2 /* salt_min/35*2225 */
#define OBGC_MASS_TO_VOL (1e6 * 1.026)

// Safeguarded Newton / bisection for [H+] (drtsafe_row).  `live` = this lane has a cell
// to solve.  Lanes of a wavefront iterate in lock-step; each sweep ends with a wavefront
// ballot and the whole wave leaves as soon as no lane is still iterating, so a wave pays
// for its slowest lane only (2..14 sweeps) and never for maxit = 100.
__device__ __forceinline__ double solve_htotal(bool live, const TalkInv& v, double kw, double phlo, double phhi) {
  double x1 = live ? pow(10.0, -phhi) : 1.0;
  double x2 = live ? pow(10.0, -phlo) : 2.0;
  double flo = 0.0, fhi = 0.0, f = 0.0, df = 1.0, dx, dxold, xlo, xhi, soln;

  // bracket: grow by sqrt(x2/x1) while the residual has one sign on both ends (920-938)
  bool seeking = live;
  while (__ballot(seeking) != 0ull) {
    if (seeking) {
      double d0;
      talk_row(v, kw, x1, flo, d0);
      talk_row(v, kw, x2, fhi, d0);
      const bool same = (flo > 0.0 && fhi > 0.0) || (flo < 0.0 && fhi < 0.0);
      if (same) {
        const double g = sqrt(x2 / x1);
        x2 = x2 * g;
        x1 = x1 / g;
      } else {
        seeking = false;
      }
    }
  }
  if (flo < 0.0) { xlo = x1; xhi = x2; }
  else { xlo = x2; xhi = x1; }
  soln = 0.5 * (xlo + xhi);
  dxold = fabs(xlo - xhi);
  dx = dxold;
  if (live) talk_row(v, kw, soln, f, df);

  bool iterating = live;
  for (int it = 1; it <= 100; ++it) {   // maxit (co2calc.F90:55)
    if (iterating) {
      const bool leave_bracket = ((soln - xhi) * df - f) * ((soln - xlo) * df - f) >= 0.0;
      const bool dx_decrease = fabs(2.0 * f) <= fabs(dxold * df);
      bool done;
      if (leave_bracket || !dx_decrease) {
        dxold = dx;
        dx = 0.5 * (xhi - xlo);
        soln = xlo + dx;
        done = (xlo == soln);
      } else {
        dxold = dx;
        dx = -f / df;
        const double prev = soln;
        soln = soln + dx;
        done = (prev == soln);
      }
      if (fabs(dx) < 1e-10) done = true;   // xacc (co2calc.F90:53)
      if (done) {
        iterating = false;
      } else {
        talk_row(v, kw, soln, f, df);
        if (f < 0.0) xlo = soln;
        else xhi = soln;
      }
    }
    if (__ballot(iterating) == 0ull) break;
  }
  return soln;
}

// CO3 concentration at calcite / aragonite saturation (comp_co3_sat_vals)
__device__ __forceinline__ void comp_co3_sat_vals(bool deep, double depth, double temp, double salt, double& sat_calc,
                                                  double& sat_arag) {
  const double p001 = 0.001;
  const double salt_lim = dmax(salt, 0.1);
  const double tk = 273.15 + temp;
  double log10tk = log(tk);
  log10tk = log10tk / OBGC_LN10;
  const double invtk = 1.0 / tk;
  const double invRtk = (1.0 / 83.1451) * invtk;
  const double sqrts = sqrt(salt_lim);
  const double s15 = sqrts * salt_lim;
  double arg = -171.9065 - 0.077993 * tk + 2839.319 * invtk + 71.595 * log10tk +
               (-0.77712 + 0.0028426 * tk + 178.34 * invtk) * sqrts - 0.07711 * salt_lim + 0.0041249 * s15;
  double K_calc = exp(OBGC_LN10 * arg);
  arg = -171.945 - 0.077993 * tk + 2903.293 * invtk + 71.595 * log10tk +
        (-0.068393 + 0.0017276 * tk + 88.135 * invtk) * sqrts - 0.10018 * salt_lim + 0.0059415 * s15;
  double K_arag = exp(OBGC_LN10 * arg);
  if (deep) {
    const double press_bar = press_bar_of(depth);
    double deltaV = -48.76 + 0.5304 * temp;
    const double Kappa = (-11.76 + 0.3692 * temp) * p001;
    K_calc = K_calc * millero_kfac(deltaV, Kappa, press_bar, invRtk);
    deltaV = deltaV + 2.8;
    K_arag = K_arag * millero_kfac(deltaV, Kappa, press_bar, invRtk);
  }
  const double inv_Ca = (35.0 / 0.01028) / salt_lim;
  sat_calc = (K_calc * inv_Ca) * OBGC_MASS_TO_VOL;
  sat_arag = (K_arag * inv_Ca) * OBGC_MASS_TO_VOL;
}

}  // namespace obgc
