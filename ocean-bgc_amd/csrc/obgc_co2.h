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

__device__ __forceinline__ double millero_lnfac(double deltaV, double Kappa, double press_bar, double invRtk) {
  return (-deltaV + 0.5 * Kappa * press_bar) * press_bar * invRtk;
}
__device__ __forceinline__ double millero_kfac(double deltaV, double Kappa, double press_bar, double invRtk) {
  return exp(millero_lnfac(deltaV, Kappa, press_bar, invRtk));
}

// K(T,S) exp(lnKfac) -- the reference's two exponentials and a product (co2calc.F90:480-490 and alike) -- is evaluated as
// ONE exponential of the summed exponents below the surface level: nine exp (~40 fp64 instructions each) less per
// interior cell of a kernel whose instruction count is its run time.  |ln K| < 60 and |lnKfac| < 1, so the sum carries
// an absolute rounding error of <= 4e-15, i.e. K moves by <= 4e-15 relative (pH by 2e-15).  At the surface level the
// added exponent is an exact 0 and nothing changes.  -DOBGC_PRESSURE_FACTOR_SEPARATE restores the two-exponential form.
#ifdef OBGC_PRESSURE_FACTOR_SEPARATE
#define OBGC_K_AT_PRESSURE(arg, deep, dV, kap, pb, irt) (exp(arg) * ((deep) ? millero_kfac(dV, kap, pb, irt) : 1.0))
#else
#define OBGC_K_AT_PRESSURE(arg, deep, dV, kap, pb, irt) exp((arg) + ((deep) ? millero_lnfac(dV, kap, pb, irt) : 0.0))
#endif

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
  c.kb = OBGC_K_AT_PRESSURE(arg, deep, bdV, -2.84 * p001, press_bar, invRtk);
  // k1p (560-580)
  arg = -4576.752 * invtk + 115.525 - 18.453 * dlogtk + (-106.736 * invtk + 0.69171) * sqrts +
        (-0.65643 * invtk - 0.01844) * salt_lim;
  c.k1p = OBGC_K_AT_PRESSURE(arg, deep, -14.51 + (0.1211 - 0.000321 * temp) * temp, (-2.67 + 0.0427 * temp) * p001, press_bar, invRtk);
  // k2p (589-609)
  arg = -8814.715 * invtk + 172.0883 - 27.927 * dlogtk + (-160.340 * invtk + 1.3566) * sqrts +
        (0.37335 * invtk - 0.05778) * salt_lim;
  c.k2p = OBGC_K_AT_PRESSURE(arg, deep, -23.12 + (0.1758 - 0.002647 * temp) * temp, (-5.15 + 0.09 * temp) * p001, press_bar, invRtk);
  // k3p (618-637)
  arg = -3070.75 * invtk - 18.141 + (17.27039 * invtk + 2.81197) * sqrts + (-44.99486 * invtk - 0.09984) * salt_lim;
  c.k3p = OBGC_K_AT_PRESSURE(arg, deep, -26.57 + (0.202 - 0.003042 * temp) * temp, (-4.08 + 0.0714 * temp) * p001, press_bar, invRtk);
  // ksi (647-669)
  arg = -8904.2 * invtk + 117.385 - 19.334 * dlogtk + (-458.79 * invtk + 3.5913) * sqrtis +
        (188.74 * invtk - 1.5998) * is + (-12.1652 * invtk + 0.07871) * is2 + log_1_m_1p005em3_s;
  c.ksi = OBGC_K_AT_PRESSURE(arg, deep, bdV, -2.84 * p001, press_bar, invRtk);
  // kw (681-700)
  arg = -13847.26 * invtk + 148.9652 - 23.6521 * dlogtk + (118.67 * invtk - 5.977 + 1.0495 * dlogtk) * sqrts -
        0.01615 * salt_lim;
  c.kw = OBGC_K_AT_PRESSURE(arg, deep, -20.02 + (0.1119 - 0.001409 * temp) * temp, (-5.13 + 0.0794 * temp) * p001, press_bar, invRtk);
  // ks (709-731)
  arg = -4276.1 * invtk + 141.328 - 23.093 * dlogtk + (-13856.0 * invtk + 324.57 - 47.986 * dlogtk) * sqrtis +
        (35474.0 * invtk - 771.54 + 114.723 * dlogtk) * is - 2698.0 * invtk * is * sqrtis + 1776.0 * invtk * is2 +
        log_1_m_1p005em3_s;
  c.ks = OBGC_K_AT_PRESSURE(arg, deep, -18.03 + (0.0466 + 0.000316 * temp) * temp, (-4.53 + 0.09 * temp) * p001, press_bar, invRtk);
  // kf needs the (corrected) ks (740-764)
  const double log_1_p_tot_sulfate_div_ks = log(1.0 + (0.1400 / 96.062) * (scl) / c.ks);
  arg = 1590.2 * invtk - 12.641 + 1.525 * sqrtis + log_1_m_1p005em3_s + log_1_p_tot_sulfate_div_ks;
  c.kf = OBGC_K_AT_PRESSURE(arg, deep, -9.78 - (0.009 + 0.000942 * temp) * temp, (-3.91 + 0.054 * temp) * p001, press_bar, invRtk);
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
  double st_cks, ft_kf;                   // st*c_ks, ft*kf (single-division form only)
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
  v.st_cks = c.st * v.c_ks;
  v.ft_kf = c.ft * c.kf;
  (void)c.kw;
}

// alkalinity residual fn(x) and d fn / dx, term by term in the reference's order (talk_row,
// co2calc.F90:1036-1090): seven divisions per evaluation.  Kept as the documented contract and as a build
// knob (-DOBGC_TALK_REFORDER) for A/B runs; the shipped solver uses talk_row below.
__device__ __forceinline__ void talk_row_reforder(const TalkInv& v, double kw, double x1, double& fn, double& df) {
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

// The same residual and derivative with ONE reciprocal per evaluation.  talk_row is 80 % of the carbonate
// kernel's instructions (2 solves x ~13 evaluations per cell) and the kernel runs at 96 % of the fp64 issue
// rate, so its instruction count IS its run time.  Identities used:
//   1/(1 + c_ks/x) = x/(x + c_ks),  1/(1 + kf/x) = x/(x + kf)      (the two "free-scale" terms)
//   the seven reciprocals 1/x, 1/a, 1/b, 1/(kb+x), 1/(ksi+x), 1/(x+c_ks), 1/(x+kf) come from one reciprocal
//   of their product (|product| stays within 1e-90..1e-30 for 1e-13 < x < 1e-3) and 18 multiplications;
//   sums are written with fma.
// This leaves the reference's operation order: each term moves by a few ulp, the root by ~1e-15 relative
// (Newton is self-correcting), far inside the pH bar of 1e-12; the measured parity did not move.
__device__ __forceinline__ double rcp_nr(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(r, e, r);
}

__device__ __forceinline__ void talk_row(const TalkInv& v, double kw, double x, double& fn, double& df) {
#ifdef OBGC_TALK_REFORDER
  talk_row_reforder(v, kw, x, fn, df);
#else
  const double x2 = x * x;
  const double x3 = x2 * x;
  const double a = __builtin_fma(__builtin_fma(x + v.k1p, x, v.k12p), x, v.k123p);
  const double da = __builtin_fma(__builtin_fma(3.0, x, v.two_k1p), x, v.k12p);
  const double b = __builtin_fma(x + v.k1, x, v.k12);
  const double db = __builtin_fma(2.0, x, v.k1);
  const double u = v.kb + x, w = v.ksi + x, s = x + v.c_ks, q = x + v.kf;
  // batch inversion
  const double p1 = x * a, p2 = p1 * b, p3 = p2 * u, p4 = p3 * w, p5 = p4 * s, p6 = p5 * q;
  double r = rcp_nr(p6);
  const double q_r = r * p5; r = r * q;
  const double s_r = r * p4; r = r * s;
  const double w_r = r * p3; r = r * w;
  const double u_r = r * p2; r = r * u;
  const double b_r = r * p1; r = r * b;
  const double a_r = r * x;
  const double x_r = r * a;

  const double n_b = __builtin_fma(v.k1_dic, x, v.two_dic_k12);
  const double n_a = __builtin_fma(-v.pt, x3, __builtin_fma(v.pt_k12p, x, v.two_pt_k123p));
  double t = n_b * b_r;
  t = __builtin_fma(v.bt_kb, u_r, t);
  t = __builtin_fma(kw, x_r, t);
  t = __builtin_fma(n_a, a_r, t);
  t = __builtin_fma(v.sit_ksi, w_r, t);
  t = __builtin_fma(-x, v.c_r, t);
  t = __builtin_fma(-v.st, x * s_r, t);
  t = __builtin_fma(-v.ft, x * q_r, t);
  fn = t - v.ta;

  const double xda = x * da;
  double m_b = v.k1_dic * __builtin_fma(-x, db, b);
  m_b = __builtin_fma(-v.two_dic_k12, db, m_b);
  double m_a = v.pt_k12p * (a - xda);
  m_a = __builtin_fma(-v.two_pt_k123p, da, m_a);
  m_a = __builtin_fma(-(v.pt * x2), __builtin_fma(3.0, a, -xda), m_a);
  double d = m_b * (b_r * b_r);
  d = __builtin_fma(-v.bt_kb, u_r * u_r, d);
  d = __builtin_fma(-kw, x_r * x_r, d);
  d = __builtin_fma(m_a, a_r * a_r, d);
  d = __builtin_fma(-v.sit_ksi, w_r * w_r, d);
  d = d - v.c_r;
  d = __builtin_fma(-v.st_cks, s_r * s_r, d);
  df = __builtin_fma(-v.ft_kf, q_r * q_r, d);
#endif
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
  // 10**(-pH) as exp(-pH ln 10): a fifth of the instructions of the generic pow; the bracket ends move by ~1e-15
  // relative, which the bracketing and the Newton iteration do not see at the pH bar
  double x1 = live ? exp(-phhi * OBGC_LN10) : 1.0;
  double x2 = live ? exp(-phlo * OBGC_LN10) : 2.0;
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
  const double press_bar = deep ? press_bar_of(depth) : 0.0;
  const double deltaV = -48.76 + 0.5304 * temp;
  const double Kappa = (-11.76 + 0.3692 * temp) * p001;
  const double K_calc = OBGC_K_AT_PRESSURE(OBGC_LN10 * arg, deep, deltaV, Kappa, press_bar, invRtk);
  arg = -171.945 - 0.077993 * tk + 2903.293 * invtk + 71.595 * log10tk +
        (-0.068393 + 0.0017276 * tk + 88.135 * invtk) * sqrts - 0.10018 * salt_lim + 0.0059415 * s15;
  const double K_arag = OBGC_K_AT_PRESSURE(OBGC_LN10 * arg, deep, deltaV + 2.8, Kappa, press_bar, invRtk);
  const double inv_Ca = (35.0 / 0.01028) / salt_lim;
  sat_calc = (K_calc * inv_Ca) * OBGC_MASS_TO_VOL;
  sat_arag = (K_arag * inv_Ca) * OBGC_MASS_TO_VOL;
}

// One cell of the interior carbonate block (BGC_mod.F90:935-1001): constants, the two [H+] solves, species,
// saturation values, PH_PREV_* update.  Every lane of the wavefront must call it (the solver votes with
// wavefront ballots); `inside` = the lane maps to a cell of the grid at all.
// the inputs of one cell's carbonate block, as values
struct CarbIn {
  bool live;
  long long ix;
  double T, S, zmid, dic, alk, po4, sio3, php, pha;   // raw
};

// SHARED: the caller already holds T, S, zmid and the clamped PO4 and SiO3 of the cell and fills them in itself
template <bool SHARED = false>
__device__ __forceinline__ void carbonate_load(const BgcKernelArgs& A, int i_dic, int i_alk, int i_po4, int i_sio3, bool inside, int k,
                                               int c, CarbIn& in) {
  const GridView& g = A.g;
  const int kmax = inside ? A.kmax[c] : 0;
  in.live = inside && k < kmax;
  in.ix = inside ? g.at(k, c) : 0;
  const long long ix = in.ix;
  in.T = 10.0; in.S = 35.0; in.zmid = 0.0; in.dic = 2000.0; in.alk = 2300.0; in.po4 = 0.0; in.sio3 = 0.0; in.php = 0.0; in.pha = 0.0;
  if (in.live) {
    if (!SHARED) {
      in.T = A.temp[ix];
      in.S = A.salt[ix];
      in.zmid = A.zmid[ix];
      in.po4 = A.tracers[(long long)(i_po4 - 1) * g.plane + ix];
      in.sio3 = A.tracers[(long long)(i_sio3 - 1) * g.plane + ix];
    }
    in.dic = A.tracers[(long long)(i_dic - 1) * g.plane + ix];
    in.alk = A.tracers[(long long)(i_alk - 1) * g.plane + ix];
    in.php = A.ph_prev[ix];
    in.pha = A.ph_prev_alt[ix];
  }
}

// what the saturation-horizon search (BGC_mod.F90:1003-1032) needs from a cell's carbonate block
struct CarbOut { double CO3, sat_calc, sat_arag; };

__device__ __forceinline__ void carbonate_solve(const BgcKernelArgs& A, bool inside, int k, const CarbIn& in, CarbOut* out = nullptr);

__device__ __forceinline__ void carbonate_cell(const BgcKernelArgs& A, int i_dic, int i_alk, int i_po4, int i_sio3, bool inside,
                                               int k, int c) {
  CarbIn in;
  carbonate_load(A, i_dic, i_alk, i_po4, i_sio3, inside, k, c, in);
  carbonate_solve(A, inside, k, in);
}

__device__ __forceinline__ void carbonate_solve(const BgcKernelArgs& A, bool inside, int k, const CarbIn& in, CarbOut* out) {
  const bool live = in.live;
  const long long ix = in.ix;
  double T = 10.0, S = 35.0, depth = 0.0, dic = 2000.0, alk = 2300.0, po4 = 0.0, sio3 = 0.0;
  double php = 0.0, pha = 0.0;
  if (live) {
    T = in.T;
    S = in.S;
    depth = in.zmid * 0.01;                                      // cm -> m, BGC_mod.F90:950
    dic = dmax(0.0, in.dic);                                     // clamps of 747-752
    alk = dmax(0.0, in.alk);
    po4 = dmax(0.0, in.po4);
    sio3 = dmax(0.0, in.sio3);
    php = in.php;
    pha = in.pha;
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

  // the saturation concentrations feed diagnostics and the saturation-horizon search only (BGC_mod.F90:995-1032): a call
  // that writes neither skips their ~300 instructions per cell (wave-uniform test)
  double sat_calc = 0.0, sat_arag = 0.0;
  if (out || A.sat_calc_out || A.sat_arag_out || (A.have_diag && (A.diag.co3_sat_calc || A.diag.co3_sat_arag)))
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
    if (out) { out->CO3 = CO3; out->sat_calc = sat_calc; out->sat_arag = sat_arag; }
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
    // inactive cells are left untouched (the reference's loop stops at the sea floor, BGC_mod.F90:935-1001)
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

}  // namespace obgc
