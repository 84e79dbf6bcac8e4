/* ORACLE -- TEST INFRASTRUCTURE ONLY (see obgc_oracle.h).
 *
 * CPU restatement of the reference hot path.  Every function cites the reference
 * lines it follows; operation order inside each expression is the reference's
 * (left to right, explicit parentheses kept) because the reference is compiled
 * without reassociation or FMA contraction (oracle/Makefile).
 */
#include "obgc_oracle.h"

#include <math.h>
#include <string.h>

/* Branch counters (the arms of every data-dependent branch the workload must reach): every BR() marks one arm of a data-dependent
 * branch of the reference; tests assert that the synthetic workload reaches all of them.  Thread-local, so the
 * restatement stays re-entrant. */
enum { BR_zero_mask, BR_gQfe_limited, BR_gQsi_fe_limited, BR_gQsi_fe_zero, BR_gQsi_si_limited, BR_kpar_low, BR_kpar_high, BR_ph_cold, BR_ph_warm, BR_zsat_found, BR_zsat_never, BR_q10_below_thres, BR_mmrt_north, BR_mmrt_south, BR_caco3_cold, BR_caco3_bloom, BR_zumax_taper_north, BR_zumax_taper_south, BR_par_avg_dark, BR_fe_scav_high, BR_nitrif_active, BR_nitrif_taper, BR_o2_lt5, BR_o2_5_40, BR_scalelen_top, BR_scalelen_mid, BR_scalelen_deep, BR_qa_def_consumed, BR_qa_def_remains, BR_poc_hflux_zero, BR_iron_clamp, BR_bottom_flux, BR_bottom_no3_lt5, BR_bottom_o2_lt1, BR_bottom_sio2_fast, BR_bottom_above_lysocline, BR_newton_bracket_grow, BR_newton_bisect, BR_newton_step, BR_newton_exit_same, BR_newton_exit_xacc, BR_COUNT };
static _Thread_local unsigned long long ora_branch_hits[BR_COUNT];
static const char *const ora_branch_names[BR_COUNT] = {"zero_mask", "gQfe_limited", "gQsi_fe_limited", "gQsi_fe_zero", "gQsi_si_limited", "kpar_low", "kpar_high", "ph_cold", "ph_warm", "zsat_found", "zsat_never", "q10_below_thres", "mmrt_north", "mmrt_south", "caco3_cold", "caco3_bloom", "zumax_taper_north", "zumax_taper_south", "par_avg_dark", "fe_scav_high", "nitrif_active", "nitrif_taper", "o2_lt5", "o2_5_40", "scalelen_top", "scalelen_mid", "scalelen_deep", "qa_def_consumed", "qa_def_remains", "poc_hflux_zero", "iron_clamp", "bottom_flux", "bottom_no3_lt5", "bottom_o2_lt1", "bottom_sio2_fast", "bottom_above_lysocline", "newton_bracket_grow", "newton_bisect", "newton_step", "newton_exit_same", "newton_exit_xacc"};
#define BR(n) (ora_branch_hits[BR_##n]++)
int ora_branch_count(void) { return BR_COUNT; }
const char *ora_branch_name(int i) { return (i >= 0 && i < BR_COUNT) ? ora_branch_names[i] : ""; }
unsigned long long ora_branch_hit(int i) { return (i >= 0 && i < BR_COUNT) ? ora_branch_hits[i] : 0; }
void ora_branch_reset(void) { memset(ora_branch_hits, 0, sizeof ora_branch_hits); }


/* ------------------------------------------------------------------ helpers */
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

/* BGC_parms.F90:37-40 */
#define SPD 86400.0
static const double DPS = 1.0 / 86400.0;
static const double YPS = 1.0 / (365.0 * 86400.0);

/* BGC_parms.F90:327-339 */
static const double parm_Red_D_C_P = 117.0;
static const double parm_Red_Fe_C = 3.0e-6;
#define parm_Red_D_C_O2 (117.0 / 170.0)
#define parm_Remin_D_C_O2 (117.0 / 138.0)
#define parm_Red_D_C_O2_diaz (117.0 / 150.0)

/* BGC_parms.F90:371-374 */
static const double fe_scavenge_thres1 = 0.8e-3;
static const double dust_fescav_scale = 1.0e9;
static const double fe_max_scale2 = 1200.0;
/* BGC_parms.F90:385-386 */
#define dust_to_Fe (0.035 / 55.847 * 1.0e9)
/* BGC_parms.F90:394-405 */
static const double caco3_poc_min = 0.4, spc_poc_fac = 0.11, f_graze_sp_poc_lim = 0.3,
                    f_photosp_CaCO3 = 0.4, f_graze_CaCO3_remin = 0.33, f_graze_si_remin = 0.35;
/* BGC_parms.F90:411-429 */
static const double r_Nfix_photo = 1.25;
static const double Q = 0.137, Qp_zoo_pom = 0.00855, Qfe_zoo = 3.0e-6, gQsi_0 = 0.137,
                    gQsi_max = 0.685, gQsi_min = 0.0457, QCaCO3_max = 0.4;
#define denitrif_C_N (117.0 / 136.0)
/* BGC_parms.F90:435-441 */
static const double thres_z1 = 100.0e2, thres_z2 = 150.0e2, loss_thres_zoo = 0.005,
                    CaCO3_temp_thres1 = 6.0, CaCO3_temp_thres2 = -2.0, CaCO3_sp_thres = 4.0;
/* BGC_parms.F90:447-463 */
enum { tfnc_q10 = 1, tfnc_quasi_mmrt = 2 };
static const double f_qsw_par = 0.45, Tref = 30.0, Q_10 = 1.5;
/* BGC_parms.F90:469-477 */
#define DOC_reminR ((1.0 / 250.0) * DPS)
#define DON_reminR ((1.0 / 160.0) * DPS)
#define DOFe_reminR ((1.0 / 160.0) * DPS)
#define DOP_reminR ((1.0 / 160.0) * DPS)
#define DONr_reminR ((1.0 / (365.0 * 2.5)) * DPS)
#define DOPr_reminR ((1.0 / (365.0 * 2.5)) * DPS)
static const double DONrefract = 0.08, DOPrefract = 0.03;
/* BGC_parms.F90:479-486: default-real (single precision) literals widened to r8 */
static const double epsC = (double)1.00e-8f, epsTinv = (double)3.17e-8f;
static const double cks = 9.0, cksi = 5.0;
/* BGC_parms.F90:488-489 */
static const double xkw_coeff = 8.6e-9;
/* BGC_mod.F90:144-149 */
static const double phlo_surf_init = 7.0, phhi_surf_init = 9.0, phlo_3d_init = 6.0,
                    phhi_3d_init = 9.0, del_ph = 0.20;

/* ============================================================ parameters == */

/* BGC_parms.F90:497-699 */
void ora_bgc_params_init(ora_bgc_params *p, const int *ind, double T0K) {
  memset(p, 0, sizeof(*p));
  for (int i = 0; i < ORA_BGC_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->T0_Kelvin_BGC = T0K;
  p->sp_ind = 1; p->diat_ind = 2; p->diaz_ind = 3; p->phaeo_ind = 4;       /* 515-518 */
  p->parm_Fe_bioavail = 1.0; p->parm_o2_min = 4.0; p->parm_o2_min_delta = 2.0;
  p->parm_kappa_nitrif = 0.06 * DPS; p->parm_nitrif_par_lim = 1.0;
  p->parm_z_mort_0 = 0.1 * DPS; p->parm_z_mort2_0 = 0.4 * DPS; p->parm_labile_ratio = 0.85;
  p->parm_POMbury = 1.4; p->parm_BSIbury = 0.65; p->parm_fe_scavenge_rate0 = 3.0;
  p->parm_f_prod_sp_CaCO3 = 0.055; p->parm_POC_diss = 88.0e2; p->parm_SiO2_diss = 250.0e2;
  p->parm_CaCO3_diss = 150.0e2;
  const double sz[4] = {130.0e2, 290.0e2, 670.0e2, 1700.0e2}, sv[4] = {1.0, 3.0, 5.0, 9.0};
  memcpy(p->parm_scalelen_z, sz, sizeof sz);
  memcpy(p->parm_scalelen_vals, sv, sizeof sv);

  ora_autotroph *a = &p->autotrophs[0];                                        /* sp 543-580 */
  a->Nfixer = 0; a->imp_calcifier = 1; a->exp_calcifier = 0; a->grazee_ind = 1;
  a->kFe = 0.04e-3; a->kPO4 = 0.01; a->kDOP = 0.26; a->kNO3 = 0.1; a->kNH4 = 0.01; a->kSiO3 = 0.0;
  a->Qp = 0.00855; a->gQfe_0 = 20.0e-6; a->gQfe_min = 3.0e-6; a->alphaPI = 0.6 * DPS;
  a->PCref = 5.5 * DPS; a->thetaN_max = 2.5; a->loss_thres = 0.04; a->loss_thres2 = 0.0;
  a->temp_thres = -20.0; a->temp_thresN = -20.0; a->temp_thresS = -20.0; a->temp_function = tfnc_q10;
  a->temp_optN = 50.0; a->temp_optS = 50.0; a->mort = 0.12 * DPS; a->mort2 = 0.001 * DPS;
  a->agg_rate_max = 0.9; a->agg_rate_min = 0.01; a->z_umax_0 = 3.3 * DPS; a->z_grz = 1.05;
  a->graze_zoo = 0.3; a->graze_poc = 0.0; a->graze_doc = 0.15; a->loss_poc = 0.0; a->f_zoo_detr = 0.15;

  a = &p->autotrophs[1];                                                       /* diat 582-619 */
  a->Nfixer = 0; a->imp_calcifier = 0; a->exp_calcifier = 0; a->grazee_ind = 2;
  a->kFe = 0.06e-3; a->kPO4 = 0.05; a->kDOP = 0.9; a->kNO3 = 0.5; a->kNH4 = 0.05; a->kSiO3 = 0.8;
  a->Qp = 0.00855; a->gQfe_0 = 20.0e-6; a->gQfe_min = 3.0e-6; a->alphaPI = 0.465 * DPS;
  a->PCref = 5.5 * DPS; a->thetaN_max = 4.0; a->loss_thres = 0.04; a->loss_thres2 = 0.0;
  a->temp_thres = -20.0; a->temp_thresN = 35.0; a->temp_thresS = 10.0; a->temp_function = tfnc_q10;
  a->temp_optN = 16.3; a->temp_optS = 5.0; a->mort = 0.12 * DPS; a->mort2 = 0.001 * DPS;
  a->agg_rate_max = 0.9; a->agg_rate_min = 0.02; a->z_umax_0 = 3.23 * DPS; a->z_grz = 1.0;
  a->graze_zoo = 0.3; a->graze_poc = 0.42; a->graze_doc = 0.15; a->loss_poc = 0.0; a->f_zoo_detr = 0.2;

  a = &p->autotrophs[2];                                                       /* diaz 621-658 */
  a->Nfixer = 1; a->imp_calcifier = 0; a->exp_calcifier = 0; a->grazee_ind = 3;
  a->kFe = 0.04e-3; a->kPO4 = 0.02; a->kDOP = 0.09; a->kNO3 = 1.0; a->kNH4 = 0.15; a->kSiO3 = 0.0;
  a->Qp = 0.002735; a->gQfe_0 = 60.0e-6; a->gQfe_min = 12.0e-6; a->alphaPI = 0.4 * DPS;
  a->PCref = 0.7 * DPS; a->thetaN_max = 2.5; a->loss_thres = 0.022; a->loss_thres2 = 0.001;
  a->temp_thres = 14.0; a->temp_thresN = -20.0; a->temp_thresS = -20.0; a->temp_function = tfnc_q10;
  a->temp_optN = 50.0; a->temp_optS = 50.0; a->mort = 0.15 * DPS; a->mort2 = 0.0;
  a->agg_rate_max = 0.0; a->agg_rate_min = 0.0; a->z_umax_0 = 0.6 * DPS; a->z_grz = 1.2;
  a->graze_zoo = 0.3; a->graze_poc = 0.05; a->graze_doc = 0.15; a->loss_poc = 0.0; a->f_zoo_detr = 0.15;

  a = &p->autotrophs[3];                                                       /* phaeo 660-697 */
  a->Nfixer = 0; a->imp_calcifier = 0; a->exp_calcifier = 0; a->grazee_ind = 2;
  a->kFe = 0.075e-3; a->kPO4 = 0.05; a->kDOP = 0.9; a->kNO3 = 0.7; a->kNH4 = 0.05; a->kSiO3 = 0.0;
  a->Qp = 0.00855; a->gQfe_0 = 20.0e-6; a->gQfe_min = 3.0e-6; a->alphaPI = 0.77 * DPS;
  a->PCref = 5.5 * DPS; a->thetaN_max = 2.5; a->loss_thres = 0.04; a->loss_thres2 = 0.0;
  a->temp_thres = -20.0; a->temp_thresN = 35.0; a->temp_thresS = 10.0; a->temp_function = tfnc_quasi_mmrt;
  a->temp_optN = 16.3; a->temp_optS = 5.0; a->mort = 0.12 * DPS; a->mort2 = 0.001 * DPS;
  a->agg_rate_max = 0.9; a->agg_rate_min = 0.02; a->z_umax_0 = 3.23 * DPS; a->z_grz = 1.0;
  a->graze_zoo = 0.3; a->graze_poc = 0.42; a->graze_doc = 0.15; a->loss_poc = 0.0; a->f_zoo_detr = 0.2;

  ora_bgc_derive_autotroph_indices(p);
}

/* BGC_mod.F90:271-321 */
void ora_bgc_derive_autotroph_indices(ora_bgc_params *p) {
  for (int a = 0; a < ORA_AUTOTROPH_CNT; ++a) {
    ora_autotroph *at = &p->autotrophs[a];
    int ai = a + 1;
    if (ai == p->sp_ind) { at->Chl_ind = p->ind[ORA_spChl]; at->C_ind = p->ind[ORA_spC]; at->Fe_ind = p->ind[ORA_spFe]; }
    else if (ai == p->diat_ind) { at->Chl_ind = p->ind[ORA_diatChl]; at->C_ind = p->ind[ORA_diatC]; at->Fe_ind = p->ind[ORA_diatFe]; }
    else if (ai == p->diaz_ind) { at->Chl_ind = p->ind[ORA_diazChl]; at->C_ind = p->ind[ORA_diazC]; at->Fe_ind = p->ind[ORA_diazFe]; }
    else if (ai == p->phaeo_ind) { at->Chl_ind = p->ind[ORA_phaeoChl]; at->C_ind = p->ind[ORA_phaeoC]; at->Fe_ind = p->ind[ORA_phaeoFe]; }
    at->Si_ind = (at->kSiO3 > 0.0) ? p->ind[ORA_diatSi] : 0;
    at->CaCO3_ind = (at->imp_calcifier || at->exp_calcifier) ? p->ind[ORA_spCaCO3] : 0;
  }
}

/* DMS_parms.F90:203-241 */
void ora_dms_params_init(ora_dms_params *p, const int *ind) {
  memset(p, 0, sizeof(*p));
  for (int i = 0; i < ORA_DMS_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->k_S_p_base = 0.1 * DPS; p->zooC_avg = 0.3; p->mort = 0.0; p->k_conv = 1.0 * DPS;
  p->k_S_z = 0.1 * DPS; p->B_preexp = 0.1; p->B_exp = 0.5; p->k_S_B = 30.0 * DPS;
  p->k_bkgnd = 0.01 * DPS; p->j_dms_perI = 0.005 * DPS; p->inject_scale = 1.0;
  p->T_cryo_hi = 1.0; p->T_cryo_lo = -1.0; p->T_lo = 15.0; p->T_hi = 20.0;
  p->Min_cyano_frac = 0.0; p->Max_cyano_frac = 0.5; p->Min_yld = 0.2; p->Max_yld = 0.7;
  p->G_phaeo_S = 0.4; p->Sp_ref = 0.1; p->Stress_mult = 10.0; p->R = 0.137;
  p->Rs2n_diat = 0.01; p->Rs2n_phaeo = 0.3; p->Rs2n_cocco = 0.1; p->Rs2n_cyano = 0.0;
  p->Rs2n_eukar = 0.1; p->Rs2n_diaz = 0.0;
  p->f_qsw_par_DMS = 0.45;                                                    /* 191-192 */
}

/* MACROS_parms.F90:143-162 */
void ora_macros_params_init(ora_macros_params *p, const int *ind) {
  memset(p, 0, sizeof(*p));
  for (int i = 0; i < ORA_MACROS_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->f_prot = 0.6; p->f_poly = 0.2; p->f_lip = 0.2; p->k_C_p_base = DPS * 0.1;
  p->zooC_avg = 0.3; p->mort = 0.0; p->k_prot_bac = DPS * 0.1; p->k_poly_bac = DPS * 0.01;
  p->k_lip_bac = DPS * 1.0; p->inject_scale = 1.0;
}

int ora_sizeof_bgc_params(void) { return (int)sizeof(ora_bgc_params); }
int ora_sizeof_dms_params(void) { return (int)sizeof(ora_dms_params); }
int ora_sizeof_macros_params(void) { return (int)sizeof(ora_macros_params); }

/* =============================================================== co2calc == */

/* co2calc.F90:30-59 */
static const double rho_sw = 1.026, T0_Kelvin = 273.15;
static const double xacc = 1e-10;
static const int maxit = 100;
static const double salt_min = 0.1;
#define dic_min (0.1 / 35.0 * 1944.0)
#define alk_min (0.1 / 35.0 * 2225.0)

/* the reference keeps these in module state (co2calc.F90:65-67) */
typedef struct {
  double kw, kb, ks, kf, k1p, k2p, k3p, ksi, bt, st, ft; /* set by comp_co3_coeffs */
  double dic, ta, pt, sit;                               /* set by comp_htotal */
} co3_state;

static long long g_talk_calls = 0;
long long ora_talk_row_calls(void) { return g_talk_calls; }

/* POP ref_pressure, co2calc.F90:371-372 (also 156-157, 1153-1154) */
static inline double press_bar_of(double depth) {
  return 0.059808 * (exp(-0.025 * depth) - 1.0) + 0.100766 * depth + 2.28405e-7 * (depth * depth);
}

/* Millero pressure factor, e.g. co2calc.F90:481-489 */
static inline double kfac(double deltaV, double Kappa, double press_bar, double invRtk) {
  double lnKfac = (-deltaV + 0.5 * Kappa * press_bar) * press_bar * invRtk;
  return exp(lnKfac);
}

/* comp_co3_coeffs, co2calc.F90:320-777.  sk1/sk2 are the values BEFORE the
 * pressure factor (478 vs 480-490; 507 vs 509-519). */
static void comp_co3_coeffs(int k, double depth, double temp, double salt, double *sk0, double *sk1,
                            double *sk2, double *sff, co3_state *s) {
  const double p001 = 0.001;
  double press_bar = press_bar_of(depth);
  double salt_lim = dmax(salt, salt_min);
  double tk = T0_Kelvin + temp;
  double tk100 = tk * 1e-2;
  double tk1002 = tk100 * tk100;
  double invtk = 1.0 / tk;
  double dlogtk = log(tk);
  double invRtk = (1.0 / 83.1451) * invtk;
  double is = 19.924 * salt_lim / (1000.0 - 1.005 * salt_lim);
  double is2 = is * is;
  double sqrtis = sqrt(is);
  double sqrts = sqrt(salt_lim);
  double s2 = salt_lim * salt_lim;
  double scl = salt_lim / 1.80655;
  double arg = 1.0 - 0.001005 * salt_lim;
  double log_1_m_1p005em3_s = log(arg);
  const double log1em2 = log(1e-2), log10v = log(10.0);
  double deltaV, Kappa;

  /* ff, Weiss & Price 1980: 423-431 */
  arg = -162.8301 + 218.2968 / tk100 + 90.9241 * (dlogtk + log1em2) - 1.47696 * tk1002 +
        salt_lim * (.025695 - .025225 * tk100 + 0.0049867 * tk1002);
  *sff = exp(arg);
  /* K0, Weiss 1974: 437-444 */
  arg = 93.4517 / tk100 - 60.2409 + 23.3585 * (dlogtk + log1em2) +
        salt_lim * (.023517 - 0.023656 * tk100 + 0.0047036 * tk1002);
  *sk0 = exp(arg);
  /* k1 total scale (Lueker 2000): 461-478; callers always pass k1_k2_pH_tot=.true. */
  arg = 3633.86 * invtk - 61.2172 + 9.67770 * dlogtk - 0.011555 * salt_lim + 0.0001152 * s2;
  arg = -log10v * arg;
  *sk1 = exp(arg);
  /* 480-490: factor applied to the local k1 only -- unobservable, skipped */
  /* k2: 492-507 */
  arg = 471.78 * invtk + 25.9290 - 3.16967 * dlogtk - 0.01781 * salt_lim + 0.0001122 * s2;
  arg = -log10v * arg;
  *sk2 = exp(arg);

  /* kb: 529-551 */
  arg = (-8966.90 - 2890.53 * sqrts - 77.942 * salt_lim + 1.728 * salt_lim * sqrts - 0.0996 * s2) * invtk +
        (148.0248 + 137.1942 * sqrts + 1.62142 * salt_lim) +
        (-24.4344 - 25.085 * sqrts - 0.2474 * salt_lim) * dlogtk + 0.053105 * sqrts * tk;
  s->kb = exp(arg);
  if (k > 1) {
    deltaV = -29.48 + (0.1622 - 0.002608 * temp) * temp;
    Kappa = -2.84 * p001;
    s->kb = s->kb * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* k1p: 560-580 */
  arg = -4576.752 * invtk + 115.525 - 18.453 * dlogtk + (-106.736 * invtk + 0.69171) * sqrts +
        (-0.65643 * invtk - 0.01844) * salt_lim;
  s->k1p = exp(arg);
  if (k > 1) {
    deltaV = -14.51 + (0.1211 - 0.000321 * temp) * temp;
    Kappa = (-2.67 + 0.0427 * temp) * p001;
    s->k1p = s->k1p * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* k2p: 589-609 */
  arg = -8814.715 * invtk + 172.0883 - 27.927 * dlogtk + (-160.340 * invtk + 1.3566) * sqrts +
        (0.37335 * invtk - 0.05778) * salt_lim;
  s->k2p = exp(arg);
  if (k > 1) {
    deltaV = -23.12 + (0.1758 - 0.002647 * temp) * temp;
    Kappa = (-5.15 + 0.09 * temp) * p001;
    s->k2p = s->k2p * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* k3p: 618-637 */
  arg = -3070.75 * invtk - 18.141 + (17.27039 * invtk + 2.81197) * sqrts +
        (-44.99486 * invtk - 0.09984) * salt_lim;
  s->k3p = exp(arg);
  if (k > 1) {
    deltaV = -26.57 + (0.202 - 0.003042 * temp) * temp;
    Kappa = (-4.08 + 0.0714 * temp) * p001;
    s->k3p = s->k3p * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* ksi: 647-669 */
  arg = -8904.2 * invtk + 117.385 - 19.334 * dlogtk + (-458.79 * invtk + 3.5913) * sqrtis +
        (188.74 * invtk - 1.5998) * is + (-12.1652 * invtk + 0.07871) * is2 + log_1_m_1p005em3_s;
  s->ksi = exp(arg);
  if (k > 1) {
    deltaV = -29.48 + (0.1622 - 0.002608 * temp) * temp;
    Kappa = -2.84 * p001;
    s->ksi = s->ksi * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* kw: 681-700 */
  arg = -13847.26 * invtk + 148.9652 - 23.6521 * dlogtk +
        (118.67 * invtk - 5.977 + 1.0495 * dlogtk) * sqrts - 0.01615 * salt_lim;
  s->kw = exp(arg);
  if (k > 1) {
    deltaV = -20.02 + (0.1119 - 0.001409 * temp) * temp;
    Kappa = (-5.13 + 0.0794 * temp) * p001;
    s->kw = s->kw * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* ks: 709-731 */
  arg = -4276.1 * invtk + 141.328 - 23.093 * dlogtk +
        (-13856.0 * invtk + 324.57 - 47.986 * dlogtk) * sqrtis +
        (35474.0 * invtk - 771.54 + 114.723 * dlogtk) * is - 2698.0 * invtk * is * sqrtis +
        1776.0 * invtk * is2 + log_1_m_1p005em3_s;
  s->ks = exp(arg);
  if (k > 1) {
    deltaV = -18.03 + (0.0466 + 0.000316 * temp) * temp;
    Kappa = (-4.53 + 0.09 * temp) * p001;
    s->ks = s->ks * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* kf: 740-764 */
  arg = 1.0 + (0.1400 / 96.062) * (scl) / s->ks;
  double log_1_p_tot_sulfate_div_ks = log(arg);
  arg = 1590.2 * invtk - 12.641 + 1.525 * sqrtis + log_1_m_1p005em3_s + log_1_p_tot_sulfate_div_ks;
  s->kf = exp(arg);
  if (k > 1) {
    deltaV = -9.78 - (0.009 + 0.000942 * temp) * temp;
    Kappa = (-3.91 + 0.054 * temp) * p001;
    s->kf = s->kf * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  /* bt, st, ft: 773-775 */
  s->bt = 0.000232 / 10.811 * scl;
  s->st = 0.14 / 96.062 * scl;
  s->ft = 0.000067 / 18.9984 * scl;
}

/* talk_row, co2calc.F90:1001-1092 */
static void talk_row(const co3_state *s, double k1, double k2, double x, double *fn, double *df) {
  ++g_talk_calls;
  double x1 = x;
  double x1_r = 1.0 / x1;
  double x2 = x1 * x1;
  double x2_r = x1_r * x1_r;
  double x3 = x2 * x1;
  double k12 = k1 * k2;
  double k12p = s->k1p * s->k2p;
  double k123p = k12p * s->k3p;
  double a = x3 + s->k1p * x2 + k12p * x1 + k123p;
  double a_r = 1.0 / a;
  double a2_r = a_r * a_r;
  double da = 3.0 * x2 + 2.0 * s->k1p * x1 + k12p;
  double b = x2 + k1 * x1 + k12;
  double b_r = 1.0 / b;
  double b2_r = b_r * b_r;
  double db = 2.0 * x1 + k1;
  double c = 1.0 + s->st / s->ks;
  double c_r = 1.0 / c;
  double kb_p_x1_r = 1.0 / (s->kb + x1);
  double ksi_p_x1_r = 1.0 / (s->ksi + x1);
  double c1_p_c_ks_x1_r_r = 1.0 / (1.0 + c * s->ks * x1_r);
  double c1_p_kf_x1_r_r = 1.0 / (1.0 + s->kf * x1_r);

  *fn = k1 * s->dic * x1 * b_r + 2.0 * s->dic * k12 * b_r + s->bt * s->kb * kb_p_x1_r + s->kw * x1_r +
        s->pt * k12p * x1 * a_r + 2.0 * s->pt * k123p * a_r + s->sit * s->ksi * ksi_p_x1_r - x1 * c_r -
        s->st * c1_p_c_ks_x1_r_r - s->ft * c1_p_kf_x1_r_r - s->pt * x3 * a_r - s->ta;

  *df = k1 * s->dic * (b - x1 * db) * b2_r - 2.0 * s->dic * k12 * db * b2_r -
        s->bt * s->kb * kb_p_x1_r * kb_p_x1_r - s->kw * x2_r + (s->pt * k12p * (a - x1 * da)) * a2_r -
        2.0 * s->pt * k123p * da * a2_r - s->sit * s->ksi * ksi_p_x1_r * ksi_p_x1_r - 1.0 * c_r -
        s->st * c1_p_c_ks_x1_r_r * c1_p_c_ks_x1_r_r * (c * s->ks * x2_r) -
        s->ft * c1_p_kf_x1_r_r * c1_p_kf_x1_r_r * s->kf * x2_r - s->pt * x2 * (3.0 * a - x1 * da) * a2_r;
}

/* drtsafe_row, co2calc.F90:872-997 */
static double drtsafe_row(const co3_state *s, double k1, double k2, double x1, double x2, double xacc_) {
  double flo, fhi, f, df, dx, dxold, xlo, xhi, soln, temp;
  for (;;) {                                                        /* 920-938 */
    talk_row(s, k1, k2, x1, &flo, &df);
    talk_row(s, k1, k2, x2, &fhi, &df);
    int mask = (flo > 0.0 && fhi > 0.0) || (flo < 0.0 && fhi < 0.0);
    if (!mask) break;
    BR(newton_bracket_grow);
    dx = sqrt(x2 / x1);
    x2 = x2 * dx;
    x1 = x1 / dx;
  }
  if (flo < 0.0) { xlo = x1; xhi = x2; }                           /* 940-949 */
  else { xlo = x2; xhi = x1; temp = flo; flo = fhi; fhi = temp; }
  soln = 0.5 * (xlo + xhi);
  dxold = fabs(xlo - xhi);
  dx = dxold;
  talk_row(s, k1, k2, soln, &f, &df);
  for (int it = 1; it <= maxit; ++it) {                             /* 961-991 */
    int mask = 1;
    int leave_bracket = ((soln - xhi) * df - f) * ((soln - xlo) * df - f) >= 0;
    int dx_decrease = fabs(2.0 * f) <= fabs(dxold * df);
    if (leave_bracket || !dx_decrease) {
      dxold = dx;
      dx = 0.5 * (xhi - xlo);
      soln = xlo + dx;
      BR(newton_bisect);
      if (xlo == soln) mask = 0;
    } else {
      dxold = dx;
      dx = -f / df;
      temp = soln;
      soln = soln + dx;
      BR(newton_step);
      if (temp == soln) mask = 0;
    }
    if (!mask) BR(newton_exit_same);
    else if (fabs(dx) < xacc_) BR(newton_exit_xacc);
    if (fabs(dx) < xacc_) mask = 0;
    if (!mask) return soln;
    talk_row(s, k1, k2, soln, &f, &df);
    if (f < 0.0) { xlo = soln; flo = f; }
    else { xhi = soln; fhi = f; }
  }
  (void)flo; (void)fhi;
  return soln;
}

/* comp_htotal, co2calc.F90:781-868 */
static double comp_htotal(co3_state *s, double dic_in, double ta_in, double pt_in, double sit_in, double k1,
                          double k2, double phlo, double phhi) {
  double mass_to_vol = 1e6 * rho_sw;
  double vol_to_mass = 1.0 / mass_to_vol;
  s->dic = dmax(dic_in, dic_min) * vol_to_mass;
  s->ta = dmax(ta_in, alk_min) * vol_to_mass;
  s->pt = dmax(pt_in, 0.0) * vol_to_mass;
  s->sit = dmax(sit_in, 0.0) * vol_to_mass;
  double x1 = pow(10.0, -phhi);
  double x2 = pow(10.0, -phlo);
  return drtsafe_row(s, k1, k2, x1, x2, xacc);
}

/* comp_CO3terms, co2calc.F90:214-316 */
static void comp_CO3terms(int k, double depth, double temp, double salt, double dic_in, double ta_in,
                          double pt_in, double sit_in, double phlo, double phhi, double *ph, double *H2CO3,
                          double *HCO3, double *CO3) {
  co3_state s;
  double k0, k1, k2, ff;
  double mass_to_vol = 1e6 * rho_sw;
  comp_co3_coeffs(k, depth, temp, salt, &k0, &k1, &k2, &ff, &s);
  double htotal = comp_htotal(&s, dic_in, ta_in, pt_in, sit_in, k1, k2, phlo, phhi);
  double htotal2 = htotal * htotal;
  double denom = 1.0 / (htotal2 + k1 * htotal + k1 * k2);
  *H2CO3 = s.dic * htotal2 * denom;
  *HCO3 = s.dic * k1 * htotal * denom;
  *CO3 = s.dic * k1 * k2 * denom;
  *ph = -log10(htotal);
  *H2CO3 = *H2CO3 * mass_to_vol;
  *HCO3 = *HCO3 * mass_to_vol;
  *CO3 = *CO3 * mass_to_vol;
}

/* co2calc_1point, co2calc.F90:75-210 (k = 1; press_bar handed on as "depth", 156-161) */
static void co2calc_1point(double depth, double temp, double salt, double dic_in, double ta_in, double pt_in,
                           double sit_in, double phlo, double phhi, double *ph, double xco2_in, double atmpres,
                           double *co2star, double *dco2star, double *pCO2surf, double *dpco2) {
  co3_state s;
  double k0, k1, k2, ff;
  double mass_to_vol = 1e6 * rho_sw;
  double press_bar = press_bar_of(depth);
  comp_co3_coeffs(1, press_bar, temp, salt, &k0, &k1, &k2, &ff, &s);
  double htotal = comp_htotal(&s, dic_in, ta_in, pt_in, sit_in, k1, k2, phlo, phhi);
  double xco2 = xco2_in * 1e-6;
  double htotal2 = htotal * htotal;
  *co2star = s.dic * htotal2 / (htotal2 + k1 * htotal + k1 * k2);
  double co2starair = xco2 * ff * atmpres;
  *dco2star = co2starair - *co2star;
  *ph = -log10(htotal);
  *pCO2surf = *co2star / ff;
  *dpco2 = *pCO2surf - xco2 * atmpres;
  *co2star = *co2star * mass_to_vol;
  *dco2star = *dco2star * mass_to_vol;
  *pCO2surf = *pCO2surf * 1e6;
  *dpco2 = *dpco2 * 1e6;
}

/* comp_co3_sat_vals, co2calc.F90:1096-1238 */
static void comp_co3_sat_vals(int k, double depth, double temp, double salt, double *co3_sat_calc,
                              double *co3_sat_arag) {
  const double p001 = 0.001;
  double mass_to_vol = 1e6 * rho_sw;
  double press_bar = press_bar_of(depth);
  double salt_lim = dmax(salt, salt_min);
  double tk = T0_Kelvin + temp;
  double log10tk = log(tk);
  const double log10v = log(10.0);
  log10tk = log10tk / log10v;
  double invtk = 1.0 / tk;
  double invRtk = (1.0 / 83.1451) * invtk;
  double sqrts = sqrt(salt_lim);
  double s15 = sqrts * salt_lim;
  double deltaV = 0.0, Kappa = 0.0;
  double arg = -171.9065 - 0.077993 * tk + 2839.319 * invtk + 71.595 * log10tk +
               (-0.77712 + 0.0028426 * tk + 178.34 * invtk) * sqrts - 0.07711 * salt_lim + 0.0041249 * s15;
  arg = log10v * arg;
  double K_calc = exp(arg);
  if (k > 1) {
    deltaV = -48.76 + 0.5304 * temp;
    Kappa = (-11.76 + 0.3692 * temp) * p001;
    K_calc = K_calc * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  arg = -171.945 - 0.077993 * tk + 2903.293 * invtk + 71.595 * log10tk +
        (-0.068393 + 0.0017276 * tk + 88.135 * invtk) * sqrts - 0.10018 * salt_lim + 0.0059415 * s15;
  arg = log10v * arg;
  double K_arag = exp(arg);
  if (k > 1) {
    deltaV = deltaV + 2.8;
    K_arag = K_arag * kfac(deltaV, Kappa, press_bar, invRtk);
  }
  double inv_Ca = (35.0 / 0.01028) / salt_lim;
  *co3_sat_calc = K_calc * inv_Ca;
  *co3_sat_arag = K_arag * inv_Ca;
  *co3_sat_calc = *co3_sat_calc * mass_to_vol;
  *co3_sat_arag = *co3_sat_arag * mass_to_vol;
}

void ora_comp_co3terms(int n, const int *k, const double *depth, const double *temp, const double *salt,
                       const double *dic, const double *ta, const double *pt, const double *sit,
                       const double *phlo, const double *phhi, double *ph, double *h2co3, double *hco3,
                       double *co3) {
  g_talk_calls = 0;
  for (int i = 0; i < n; ++i)
    comp_CO3terms(k[i], depth[i], temp[i], salt[i], dic[i], ta[i], pt[i], sit[i], phlo[i], phhi[i], &ph[i],
                  &h2co3[i], &hco3[i], &co3[i]);
}
void ora_comp_co3_sat_vals(int n, const int *k, const double *depth, const double *temp, const double *salt,
                           double *calc, double *arag) {
  for (int i = 0; i < n; ++i) comp_co3_sat_vals(k[i], depth[i], temp[i], salt[i], &calc[i], &arag[i]);
}
void ora_co2calc_1point(int n, const double *depth, const double *temp, const double *salt, const double *dic,
                        const double *ta, const double *pt, const double *sit, const double *phlo,
                        const double *phhi, const double *xco2, const double *atmpres, double *ph,
                        double *co2star, double *dco2star, double *pco2surf, double *dpco2) {
  for (int i = 0; i < n; ++i)
    co2calc_1point(depth[i], temp[i], salt[i], dic[i], ta[i], pt[i], sit[i], phlo[i], phhi[i], &ph[i], xco2[i],
                   atmpres[i], &co2star[i], &dco2star[i], &pco2surf[i], &dpco2[i]);
}

/* ======================================================= gas-exchange bits == */

/* SCHMIDT_O2_singleValue, BGC_mod.F90:2965-3005 */
static double schmidt_o2(double SST) {
  const double a = 1638.0, b = 81.83, c = 1.483, d = 0.008004;
  return a + SST * (-b + SST * (c + SST * (-d)));
}
/* SCHMIDT_CO2_singleValue, BGC_mod.F90:3091-3128 */
static double schmidt_co2(double SST) {
  const double a = 2073.1, b = 125.62, c = 3.6276, d = 0.043219;
  return a + SST * (-b + SST * (c + SST * (-d)));
}
/* O2SAT_singleValue, BGC_mod.F90:3012-3083 */
static double o2sat(double SST, double SSS, double T0K) {
  const double a_0 = 2.00907, a_1 = 3.22014, a_2 = 4.05010, a_3 = 4.94457, a_4 = -2.56847E-1, a_5 = 3.88767,
               b_0 = -6.24523E-3, b_1 = -7.37614E-3, b_2 = -1.03410E-2, b_3 = -8.17083E-3, c_0 = -4.88682E-7;
  double TS = log(((T0K + 25.0) - SST) / (T0K + SST));
  double v = exp(a_0 + TS * (a_1 + TS * (a_2 + TS * (a_3 + TS * (a_4 + TS * a_5)))) +
                 SSS * ((b_0 + TS * (b_1 + TS * (b_2 + TS * b_3))) + SSS * c_0));
  return v / 0.0223916;
}

/* ============================================================= particles == */

/* sinking_particle, BGC_mod.F90:155-170 */
typedef struct {
  double diss, gamma, mass, rho;
  double sflux_in, hflux_in, prod, sflux_out, hflux_out, sed_loss, remin;
} sinking_particle;

/* init_particulate_terms, BGC_mod.F90:2006-2109 */
static void init_particulate_terms(const ora_bgc_params *p, sinking_particle *POC, sinking_particle *P_CaCO3,
                                   sinking_particle *P_SiO2, sinking_particle *dust, sinking_particle *P_iron,
                                   double *QA_dust_def, double NET_DUST_IN) {
  POC->diss = p->parm_POC_diss; POC->gamma = 0.0; POC->mass = 12.01; POC->rho = 0.0;
  P_CaCO3->diss = p->parm_CaCO3_diss; P_CaCO3->gamma = 0.30; P_CaCO3->mass = 100.09;
  P_CaCO3->rho = 0.05 * P_CaCO3->mass / POC->mass;
  P_SiO2->diss = p->parm_SiO2_diss; P_SiO2->gamma = 0.030; P_SiO2->mass = 60.08;
  P_SiO2->rho = 0.05 * P_SiO2->mass / POC->mass;
  dust->diss = 20000.0; dust->gamma = 0.97; dust->mass = 1.0e9;
  dust->rho = 0.05 * dust->mass / POC->mass;
  P_iron->diss = 60000.0; P_iron->gamma = 0.0; P_iron->mass = 0.0; P_iron->rho = 0.0;
  P_CaCO3->sflux_out = 0.0; P_CaCO3->hflux_out = 0.0;
  P_SiO2->sflux_out = 0.0; P_SiO2->hflux_out = 0.0;
  if (NET_DUST_IN != 0.0) {
    dust->sflux_out = (1.0 - dust->gamma) * NET_DUST_IN;
    dust->hflux_out = dust->gamma * NET_DUST_IN;
  } else {
    dust->sflux_out = 0.0; dust->hflux_out = 0.0;
  }
  P_iron->sflux_out = 0.0; P_iron->hflux_out = 0.0;
  POC->sflux_out = 0.0; POC->hflux_out = 0.0;
  *QA_dust_def = dust->rho * (dust->sflux_out + dust->hflux_out);
}

/* compute_particulate_terms, BGC_mod.F90:2116-2699.  `isbottom` = (k == kmax). */
static void compute_particulate_terms(const ora_bgc_params *p, int isbottom, sinking_particle *POC,
                                      sinking_particle *P_CaCO3, sinking_particle *P_SiO2,
                                      sinking_particle *dust, sinking_particle *P_iron, double *QA_dust_def,
                                      double TEMP, double O2_loc, double NO3_loc, double *SED_DENITRIF,
                                      double *OTHER_REMIN, double cell_thickness, double cell_bottom_depth,
                                      double FESEDFLUX_loc) {
  const double mpercm = 0.01;
  double scalelength = 0.0;
  /* 2242-2255 */
  P_CaCO3->sflux_in = P_CaCO3->sflux_out; P_CaCO3->hflux_in = P_CaCO3->hflux_out;
  P_SiO2->sflux_in = P_SiO2->sflux_out; P_SiO2->hflux_in = P_SiO2->hflux_out;
  dust->sflux_in = dust->sflux_out; dust->hflux_in = dust->hflux_out;
  POC->sflux_in = POC->sflux_out; POC->hflux_in = POC->hflux_out;
  P_iron->sflux_in = P_iron->sflux_out; P_iron->hflux_in = P_iron->hflux_out;
  /* 2261-2267 */
  P_iron->sed_loss = 0.0; POC->sed_loss = 0.0; P_CaCO3->sed_loss = 0.0; P_SiO2->sed_loss = 0.0;
  dust->sed_loss = 0.0; *SED_DENITRIF = 0.0; *OTHER_REMIN = 0.0;
  /* 2273-2286 */
  const double *sz = p->parm_scalelen_z, *sv = p->parm_scalelen_vals;
  if (cell_bottom_depth < sz[0]) BR(scalelen_top); else if (cell_bottom_depth >= sz[3]) BR(scalelen_deep); else BR(scalelen_mid);
  if (cell_bottom_depth < sz[0]) scalelength = sv[0];
  else if (cell_bottom_depth >= sz[3]) scalelength = sv[3];
  else {
    for (int n = 1; n < 4; ++n) {
      if (cell_bottom_depth < sz[n]) {
        scalelength = sv[n - 1] + (sv[n] - sv[n - 1]) * (cell_bottom_depth - sz[n - 1]) / (sz[n] - sz[n - 1]);
        break;
      }
    }
  }
  double DECAY_Hard = exp(-cell_thickness / 4.0e6);
  double DECAY_HardDust = exp(-cell_thickness / 1.2e7);
  /* 2295 */
  double TfuncS = pow(1.5, ((TEMP + p->T0_Kelvin_BGC) - (Tref + p->T0_Kelvin_BGC)) / 10.0);
  double dz_loc = cell_thickness;
  double dzr_loc = 1.0 / dz_loc;
  double poc_diss = POC->diss, sio2_diss = P_SiO2->diss, caco3_diss = P_CaCO3->diss, dust_diss = dust->diss;
  /* 2311-2315 */
  if ((O2_loc >= 5.0) && (O2_loc < 40.0)) BR(o2_5_40); else if (O2_loc < 5.0) BR(o2_lt5);
  if ((O2_loc >= 5.0) && (O2_loc < 40.0)) poc_diss = POC->diss * (1.0 + (3.3 - 1.0) * (40.0 - O2_loc) / 35.0);
  else if (O2_loc < 5.0) poc_diss = POC->diss * 3.3;
  /* 2321-2330 */
  poc_diss = scalelength * poc_diss;
  sio2_diss = scalelength * sio2_diss;
  caco3_diss = scalelength * caco3_diss;
  dust_diss = scalelength * dust_diss;
  sio2_diss = sio2_diss / TfuncS;
  /* 2336-2339 */
  double decay_POC_E = exp(-dz_loc / poc_diss);
  double decay_SiO2 = exp(-dz_loc / sio2_diss);
  double decay_CaCO3 = exp(-dz_loc / caco3_diss);
  double decay_dust = exp(-dz_loc / dust_diss);
  /* 2349-2365 */
  P_CaCO3->sflux_out = P_CaCO3->sflux_in * decay_CaCO3 +
                       P_CaCO3->prod * ((1.0 - P_CaCO3->gamma) * (1.0 - decay_CaCO3) * caco3_diss);
  P_CaCO3->hflux_out = P_CaCO3->hflux_in * DECAY_Hard + P_CaCO3->prod * (P_CaCO3->gamma * dz_loc);
  P_SiO2->sflux_out = P_SiO2->sflux_in * decay_SiO2 +
                      P_SiO2->prod * ((1.0 - P_SiO2->gamma) * (1.0 - decay_SiO2) * sio2_diss);
  P_SiO2->hflux_out = P_SiO2->hflux_in * DECAY_Hard + P_SiO2->prod * (P_SiO2->gamma * dz_loc);
  dust->sflux_out = dust->sflux_in * decay_dust;
  dust->hflux_out = dust->hflux_in * DECAY_HardDust;
  /* 2373-2375 */
  double POC_PROD_avail = POC->prod - P_CaCO3->rho * P_CaCO3->prod - P_SiO2->rho * P_SiO2->prod;
  /* 2390-2412 */
  double new_QA_dust_def;
  if (*QA_dust_def > 0)
    new_QA_dust_def = *QA_dust_def * (dust->sflux_out + dust->hflux_out) / (dust->sflux_in + dust->hflux_in);
  else
    new_QA_dust_def = 0.0;
  if (new_QA_dust_def > 0.0) {
    new_QA_dust_def = new_QA_dust_def - POC_PROD_avail * dz_loc;
    if (new_QA_dust_def < 0.0) BR(qa_def_consumed); else BR(qa_def_remains);
    if (new_QA_dust_def < 0.0) {
      POC_PROD_avail = -new_QA_dust_def * dzr_loc;
      new_QA_dust_def = 0.0;
    } else {
      POC_PROD_avail = 0.0;
    }
  }
  *QA_dust_def = new_QA_dust_def;
  /* 2423-2438 */
  if (POC->hflux_in == 0.0 && POC->prod == 0.0) BR(poc_hflux_zero);
  if (POC->hflux_in == 0.0 && POC->prod == 0.0) {
    POC->hflux_out = 0.0;
  } else {
    POC->hflux_out = P_CaCO3->rho * (P_CaCO3->sflux_out + P_CaCO3->hflux_out) +
                     P_SiO2->rho * (P_SiO2->sflux_out + P_SiO2->hflux_out) +
                     dust->rho * (dust->sflux_out + dust->hflux_out) - new_QA_dust_def;
    POC->hflux_out = dmax(POC->hflux_out, 0.0);
  }
  POC->sflux_out = POC->sflux_in * decay_POC_E + POC_PROD_avail * ((1.0 - decay_POC_E) * poc_diss);
  /* 2445-2463 */
  P_CaCO3->remin = P_CaCO3->prod +
                   ((P_CaCO3->sflux_in - P_CaCO3->sflux_out) + (P_CaCO3->hflux_in - P_CaCO3->hflux_out)) * dzr_loc;
  P_SiO2->remin = P_SiO2->prod +
                  ((P_SiO2->sflux_in - P_SiO2->sflux_out) + (P_SiO2->hflux_in - P_SiO2->hflux_out)) * dzr_loc;
  POC->remin = POC->prod + ((POC->sflux_in - POC->sflux_out) + (POC->hflux_in - POC->hflux_out)) * dzr_loc;
  dust->remin = ((dust->sflux_in - dust->sflux_out) + (dust->hflux_in - dust->hflux_out)) * dzr_loc;
  /* 2469-2486 */
  if (POC->sflux_in + POC->hflux_in == 0.0)
    P_iron->remin = (POC->remin * parm_Red_Fe_C);
  else
    P_iron->remin = (POC->remin * (P_iron->sflux_in + P_iron->hflux_in) / (POC->sflux_in + POC->hflux_in));
  P_iron->remin = P_iron->remin + (P_iron->sflux_in * 1.5e-5);
  P_iron->sflux_out = P_iron->sflux_in + dz_loc * ((1.0 - P_iron->gamma) * P_iron->prod - P_iron->remin);
  if (P_iron->sflux_out < 0.0) BR(iron_clamp);
  if (P_iron->sflux_out < 0.0) {
    P_iron->sflux_out = 0.0;
    P_iron->remin = P_iron->sflux_in * dzr_loc + (1.0 - P_iron->gamma) * P_iron->prod;
  }
  /* 2497-2501 */
  P_iron->remin = P_iron->remin + dust->remin * dust_to_Fe + (FESEDFLUX_loc * dzr_loc);
  P_iron->hflux_out = P_iron->hflux_in;

  if (isbottom) {                                                     /* 2522-2631 */
    double flux = POC->sflux_out + POC->hflux_out, flux_alt;
    if (flux > 0.0) {
      BR(bottom_flux);
      flux_alt = flux * mpercm * SPD;
      POC->sed_loss = flux * dmin(0.8, p->parm_POMbury * (0.013 + 0.53 * flux_alt * flux_alt /
                                                            ((7.0 + flux_alt) * (7.0 + flux_alt))));
      *SED_DENITRIF = dzr_loc * flux * (0.06 + 0.19 * pow(0.99, (O2_loc - NO3_loc)));
      if (NO3_loc < 5.0) BR(bottom_no3_lt5);
      if (NO3_loc < 5.0) *SED_DENITRIF = 0.;
      flux_alt = flux * 1.0e-6 * SPD * 365.0;
      *OTHER_REMIN = dzr_loc * dmin(dmin(0.1 + flux_alt, 0.5) * (flux - POC->sed_loss),
                                    (flux - POC->sed_loss - (*SED_DENITRIF * dz_loc * denitrif_C_N)));
      if (O2_loc < 1.0) BR(bottom_o2_lt1);
      if (O2_loc < 1.0)
        *OTHER_REMIN = dzr_loc * (flux - POC->sed_loss - (*SED_DENITRIF * dz_loc * denitrif_C_N));
    }
    flux = P_SiO2->sflux_out + P_SiO2->hflux_out;
    flux_alt = flux * mpercm * SPD;
    if (flux_alt > 2.0) BR(bottom_sio2_fast);
    if (flux_alt > 2.0) P_SiO2->sed_loss = 0.2;
    else P_SiO2->sed_loss = 0.04;
    P_SiO2->sed_loss = flux * p->parm_BSIbury * P_SiO2->sed_loss;
    if (cell_bottom_depth < 3300.0e2) BR(bottom_above_lysocline);
    if (cell_bottom_depth < 3300.0e2) {
      flux = P_CaCO3->sflux_out + P_CaCO3->hflux_out;
      P_CaCO3->sed_loss = flux;
    }
    flux = P_CaCO3->sflux_out + P_CaCO3->hflux_out;
    if (flux > 0.0) P_CaCO3->remin = P_CaCO3->remin + ((flux - P_CaCO3->sed_loss) * dzr_loc);
    flux = P_SiO2->sflux_out + P_SiO2->hflux_out;
    if (flux > 0.0) P_SiO2->remin = P_SiO2->remin + ((flux - P_SiO2->sed_loss) * dzr_loc);
    flux = POC->sflux_out + POC->hflux_out;
    if (flux > 0.0) POC->remin = POC->remin + ((flux - POC->sed_loss) * dzr_loc);
    flux = (P_iron->sflux_out + P_iron->hflux_out);
    if (flux > 0.0) P_iron->sed_loss = flux;
    dust->sed_loss = dust->sflux_out + dust->hflux_out;
    P_CaCO3->sflux_out = 0.0; P_CaCO3->hflux_out = 0.0;
    P_SiO2->sflux_out = 0.0; P_SiO2->hflux_out = 0.0;
    dust->sflux_out = 0.0; dust->hflux_out = 0.0;
    POC->sflux_out = 0.0; POC->hflux_out = 0.0;
    P_iron->sflux_out = 0.0; P_iron->hflux_out = 0.0;
  }
}

/* ======================================================== BGC_SourceSink == */

void ora_bgc_source_sink(const ora_bgc_params *p, int nz, int nc, const double *tracers, const double *temp,
                         const double *salt, const double *zmid, const double *dzv, const double *zbot,
                         const double *lat, const int *kmaxv, const double *fesed, const double *dustv,
                         const double *sw, double *tend, double *ph_prev, double *ph_prev_alt, double *dkc,
                         double *dkca, double *dca, double *dc, int alt_co2_use_eco) {
  const size_t plane = (size_t)nz * nc;
  const ora_autotroph *at = p->autotrophs;
  const double T0K = p->T0_Kelvin_BGC;
  double dummy;
#define TR(slot, k, c) tracers[(size_t)((slot)-1) * plane + (size_t)(c) * nz + (k)]
#define TEND(slot, k, c) tend[(size_t)((slot)-1) * plane + (size_t)(c) * nz + (k)]
#define KC(name, k, c) (*(dkc ? &dkc[(size_t)DKC_##name * plane + (size_t)(c) * nz + (k)] : &dummy))
#define KCA(name, k, c, a) (*(dkca ? &dkca[((size_t)DKCA_##name * 4 + (a)) * plane + (size_t)(c) * nz + (k)] : &dummy))
#define CA(name, c, a) (*(dca ? &dca[((size_t)DCA_##name * 4 + (a)) * nc + (c)] : &dummy))
#define CC(name, c) (*(dc ? &dc[(size_t)DC_##name * nc + (c)] : &dummy))

  /* 570: all tendencies zero */
  memset(tend, 0, sizeof(double) * plane * ORA_BGC_NTRACER);
  /* 625-727: zero every diagnostic except the three the reference never touches */
  if (dkc)
    for (int i = 0; i < ORA_NDIAG_KC; ++i) {
      if (i == DKC_POC_ACCUM || i == DKC_DONr_remin || i == DKC_DOPr_remin) continue;
      memset(dkc + (size_t)i * plane, 0, sizeof(double) * plane);
    }
  if (dkca) memset(dkca, 0, sizeof(double) * plane * 4 * ORA_NDIAG_KCA);
  if (dca) memset(dca, 0, sizeof(double) * nc * 4 * ORA_NDIAG_CA);
  if (dc) memset(dc, 0, sizeof(double) * nc * ORA_NDIAG_C);

  for (int c = 0; c < nc; ++c) {                                     /* column_loop 799 */
    int kmax = kmaxv[c];
    if (kmax < 1) continue;
    sinking_particle POC, P_CaCO3, P_SiO2, dust, P_iron;
    memset(&POC, 0, sizeof POC); memset(&P_CaCO3, 0, sizeof POC); memset(&P_SiO2, 0, sizeof POC);
    memset(&dust, 0, sizeof POC); memset(&P_iron, 0, sizeof POC);
    double QA_dust_def, SED_DENITRIF = 0.0, OTHER_REMIN = 0.0;
    double ZSATCALC = 0.0, ZSATARAG = 0.0, CO3_CALC_ANOM_km1 = 0.0, CO3_ARAG_ANOM_km1 = 0.0;
    double dust_flux_in_loc = dmax(0.0, dustv[c]);                   /* 808 */
    init_particulate_terms(p, &POC, &P_CaCO3, &P_SiO2, &dust, &P_iron, &QA_dust_def, dust_flux_in_loc);
    double PAR_out = dmax(0.0, sw[c]);                               /* 813-814 */
    PAR_out = PAR_out * f_qsw_par;
    /* running column diagnostics */
    double a_tot_bSi_form = 0.0, a_tot_CaCO3_form_zint = 0.0, a_photoC_TOT_zint = 0.0,
           a_photoC_NO3_TOT_zint = 0.0, a_Chl100 = 0.0;
    double a_CaCO3_form_zint[4] = {0, 0, 0, 0}, a_photoC_zint[4] = {0, 0, 0, 0}, a_photoC_NO3_zint[4] = {0, 0, 0, 0};
    double J_C = 0, J_C100 = 0, J_N = 0, J_N100 = 0, J_P = 0, J_P100 = 0, J_Si = 0, J_Si100 = 0;

    for (int k = 0; k < kmax; ++k) {                                 /* 820 */
      const size_t ix = (size_t)c * nz + k;
      /* 747-785: clamped local copies */
      double DIC_loc = dmax(0.0, TR(p->ind[ORA_dic], k, c));
      double ALK_loc = dmax(0.0, TR(p->ind[ORA_alk], k, c));
      double PO4_loc = dmax(0.0, TR(p->ind[ORA_po4], k, c));
      double NO3_loc = dmax(0.0, TR(p->ind[ORA_no3], k, c));
      double SiO3_loc = dmax(0.0, TR(p->ind[ORA_sio3], k, c));
      double NH4_loc = dmax(0.0, TR(p->ind[ORA_nh4], k, c));
      double Fe_loc = dmax(0.0, TR(p->ind[ORA_fe], k, c));
      double O2_loc = dmax(0.0, TR(p->ind[ORA_o2], k, c));
      double DOC_loc = dmax(0.0, TR(p->ind[ORA_doc], k, c));
      double zooC_loc = dmax(0.0, TR(p->ind[ORA_zooC], k, c));
      double DON_loc = dmax(0.0, TR(p->ind[ORA_don], k, c));
      double DOFe_loc = dmax(0.0, TR(p->ind[ORA_dofe], k, c));
      double DOP_loc = dmax(0.0, TR(p->ind[ORA_dop], k, c));
      double DOPr_loc = dmax(0.0, TR(p->ind[ORA_dopr], k, c));
      double DONr_loc = dmax(0.0, TR(p->ind[ORA_donr], k, c));
      double aChl[4], aC[4], aFe[4], aSi[4] = {0, 0, 0, 0}, aCaCO3[4] = {0, 0, 0, 0};
      for (int a = 0; a < 4; ++a) {
        aChl[a] = dmax(0.0, TR(at[a].Chl_ind, k, c));
        aC[a] = dmax(0.0, TR(at[a].C_ind, k, c));
        aFe[a] = dmax(0.0, TR(at[a].Fe_ind, k, c));
        if (at[a].Si_ind > 0) aSi[a] = dmax(0.0, TR(at[a].Si_ind, k, c));
        if (at[a].CaCO3_ind > 0) aCaCO3[a] = dmax(0.0, TR(at[a].CaCO3_ind, k, c));
      }
      const double T = temp[ix], S = salt[ix], dz = dzv[ix], zc = zmid[ix], zb = zbot[ix];

      /* 826-844: zero_mask */
      for (int a = 0; a < 4; ++a) {
        int zero_mask = aChl[a] == 0.0 || aC[a] == 0.0 || aFe[a] == 0.0;
        if (at[a].Si_ind > 0) zero_mask = zero_mask || aSi[a] == 0.0;
        if (zero_mask) BR(zero_mask);
        if (zero_mask) { aChl[a] = 0.0; aC[a] = 0.0; aFe[a] = 0.0; }
        if (at[a].Si_ind > 0) { if (zero_mask) aSi[a] = 0.0; }
        if (at[a].CaCO3_ind > 0) { if (zero_mask) aCaCO3[a] = 0.0; }
      }
      /* 850-856 */
      double thetaC[4], Qfe[4], Qsi[4] = {0, 0, 0, 0}, gQfe[4], gQsi[4] = {0, 0, 0, 0}, QCaCO3[4] = {0, 0, 0, 0};
      for (int a = 0; a < 4; ++a) {
        thetaC[a] = aChl[a] / (aC[a] + epsC);
        Qfe[a] = aFe[a] / (aC[a] + epsC);
        if (at[a].Si_ind > 0) Qsi[a] = dmin(aSi[a] / (aC[a] + epsC), gQsi_max);
      }
      /* 864-898 */
      for (int a = 0; a < 4; ++a) {
        gQfe[a] = at[a].gQfe_0;
        if (Fe_loc < cks * at[a].kFe)
          { BR(gQfe_limited); gQfe[a] = dmax((gQfe[a] * Fe_loc / (cks * at[a].kFe)), at[a].gQfe_min); }
        if (at[a].Si_ind > 0) {
          gQsi[a] = gQsi_0;
          if ((Fe_loc < cksi * at[a].kFe) && (Fe_loc > 0.0) && (SiO3_loc > (cksi * at[a].kSiO3)))
            { BR(gQsi_fe_limited); gQsi[a] = dmin((gQsi[a] * cksi * at[a].kFe / Fe_loc), gQsi_max); }
          if (Fe_loc == 0.0) { BR(gQsi_fe_zero); gQsi[a] = gQsi_max; }
          if (SiO3_loc < (cksi * at[a].kSiO3))
            { BR(gQsi_si_limited); gQsi[a] = dmax((gQsi[a] * SiO3_loc / (cksi * at[a].kSiO3)), gQsi_min); }
        }
        if (at[a].CaCO3_ind > 0) {
          QCaCO3[a] = aCaCO3[a] / (aC[a] + epsC);
          if (QCaCO3[a] > QCaCO3_max) QCaCO3[a] = QCaCO3_max;
        }
      }
      /* 907-924: PAR */
      double PAR_in = PAR_out;
      double work1 = dmax(((aChl[0] + aChl[1]) + aChl[2]) + aChl[3], 0.02);
      double KPARdz;
      if (work1 < 0.13224) BR(kpar_low); else BR(kpar_high);
      if (work1 < 0.13224) KPARdz = 0.000919 * pow(work1, 0.3536);
      else KPARdz = 0.001131 * pow(work1, 0.4562);
      KPARdz = KPARdz * dz;
      PAR_out = PAR_in * exp(-KPARdz);
      double PAR_avg = PAR_in * (1.0 - exp(-KPARdz)) / KPARdz;

      /* 940-1001: carbonate chemistry (NB 975: the ALT solve is fed DIC_loc) */
      double work2, work3 = 0.0, work4 = 0.0, work5, H2CO3, HCO3, CO3, H2CO3_ALT, HCO3_ALT, CO3_ALT;
      if (ph_prev[ix] != 0.0) BR(ph_warm); else BR(ph_cold);
      if (ph_prev[ix] != 0.0) { work1 = ph_prev[ix] - del_ph; work2 = ph_prev[ix] + del_ph; }
      else { work1 = phlo_3d_init; work2 = phhi_3d_init; }
      work5 = zc * 0.01;
      comp_CO3terms(k + 1, work5, T, S, DIC_loc, ALK_loc, PO4_loc, SiO3_loc, work1, work2, &work3, &H2CO3, &HCO3, &CO3);
      ph_prev[ix] = work3;
      if (ph_prev_alt[ix] != 0.0) { work1 = ph_prev_alt[ix] - del_ph; work2 = ph_prev_alt[ix] + del_ph; }
      else { work1 = phlo_3d_init; work2 = phhi_3d_init; }
      comp_CO3terms(k + 1, work5, T, S, DIC_loc, ALK_loc, PO4_loc, SiO3_loc, work1, work2, &work4, &H2CO3_ALT,
                    &HCO3_ALT, &CO3_ALT);
      ph_prev_alt[ix] = work4;
      KC(CO3, k, c) = CO3; KC(HCO3, k, c) = HCO3; KC(H2CO3, k, c) = H2CO3; KC(pH_3D, k, c) = work3;
      KC(CO3_ALT_CO2, k, c) = CO3_ALT; KC(HCO3_ALT_CO2, k, c) = HCO3_ALT; KC(H2CO3_ALT_CO2, k, c) = H2CO3_ALT;
      KC(pH_3D_ALT_CO2, k, c) = work4;
      comp_co3_sat_vals(k + 1, work5, T, S, &work1, &work2);
      KC(co3_sat_calc, k, c) = work1; KC(co3_sat_arag, k, c) = work2;
      /* 1003-1032: saturation horizons */
      if (k == 0) {
        ZSATCALC = (CO3 > work1) ? -1.0 : 0.0;
        ZSATARAG = (CO3 > work2) ? -1.0 : 0.0;
      } else {
        double zprev = zmid[ix - 1];
        work4 = zprev + (zc - zprev);
        if (ZSATCALC == -1.0 && CO3 <= work1)
          ZSATCALC = work4 * CO3_CALC_ANOM_km1 / (CO3_CALC_ANOM_km1 - (CO3 - work1));
        if (ZSATARAG == -1.0 && CO3 <= work2)
          ZSATARAG = work4 * CO3_ARAG_ANOM_km1 / (CO3_ARAG_ANOM_km1 - (CO3 - work2));
        if (ZSATCALC == -1.0 && k == kmax - 1) ZSATCALC = zb;
        if (ZSATARAG == -1.0 && k == kmax - 1) ZSATARAG = zb;
      }
      CO3_CALC_ANOM_km1 = CO3 - work1;
      CO3_ARAG_ANOM_km1 = CO3 - work2;
      if (k == kmax - 1 && k > 0) { if (ZSATCALC == zb) BR(zsat_never); else if (ZSATCALC > 0.0) BR(zsat_found); }
      if (k == kmax - 1) { CC(zsatcalc, c) = ZSATCALC; CC(zsatarag, c) = ZSATARAG; }

      /* 1041 */
      double Tfunc = pow(Q_10, ((T + T0K) - (Tref + T0K)) / 10.0);
      /* 1047-1055 */
      double f_loss_thres;
      if (zc > thres_z1) {
        if (zc < thres_z2) f_loss_thres = (thres_z2 - zc) / (thres_z2 - thres_z1);
        else f_loss_thres = 0.0;
      } else f_loss_thres = 1.0;
      /* 1072-1094 */
      double Pprime[4], C_loss_thres, tmpTopt, tmpTmax;
      for (int a = 0; a < 4; ++a) {
        C_loss_thres = f_loss_thres * at[a].loss_thres;
        if (at[a].temp_function == tfnc_q10) {
          if (T < at[a].temp_thres) BR(q10_below_thres);
          if (T < at[a].temp_thres) C_loss_thres = f_loss_thres * at[a].loss_thres2;
        } else if (at[a].temp_function == tfnc_quasi_mmrt) {
          tmpTmax = (lat[c] >= 0.0) ? at[a].temp_thresN : at[a].temp_thresS;
          if (T > tmpTmax) C_loss_thres = f_loss_thres * at[a].loss_thres2;
        }
        Pprime[a] = dmax(aC[a] - C_loss_thres, 0.0);
      }

      /* 1107-1388: autotroph loop */
      double VNO3[4], VNH4[4], VNtot[4], NO3_V[4], NH4_V[4], PO4_V[4], DOP_V[4], photoC[4], photoFe[4],
          photoSi[4] = {0, 0, 0, 0}, CaCO3_PROD[4] = {0, 0, 0, 0}, photoacc[4], Nfix[4] = {0, 0, 0, 0},
          Nexcrete[4] = {0, 0, 0, 0}, auto_loss[4], auto_agg[4], auto_graze[4], auto_graze_zoo[4],
          auto_graze_poc[4], auto_graze_doc[4], auto_graze_dic[4], auto_loss_poc[4], auto_loss_doc[4],
          auto_loss_dic[4], remaining_P_dop[4] = {0, 0, 0, 0}, remaining_P_dip[4] = {0, 0, 0, 0};
      double d_tot_CaCO3_form = 0.0, d_tot_Nfix = 0.0;
      for (int a = 0; a < 4; ++a) {
        VNO3[a] = (NO3_loc / at[a].kNO3) / (1.0 + (NO3_loc / at[a].kNO3) + (NH4_loc / at[a].kNH4));
        VNH4[a] = (NH4_loc / at[a].kNH4) / (1.0 + (NO3_loc / at[a].kNO3) + (NH4_loc / at[a].kNH4));
        VNtot[a] = VNO3[a] + VNH4[a];
        if (at[a].Nfixer) VNtot[a] = 1.0;
        KCA(N_lim, k, c, a) = VNtot[a];
        double VFe = Fe_loc / (Fe_loc + at[a].kFe);
        KCA(Fe_lim, k, c, a) = VFe;
        double f_nut = dmin(VNtot[a], VFe);
        double VPO4 = (PO4_loc / at[a].kPO4) / (1.0 + (PO4_loc / at[a].kPO4) + (DOP_loc / at[a].kDOP));
        double VDOP = (DOP_loc / at[a].kDOP) / (1.0 + (PO4_loc / at[a].kPO4) + (DOP_loc / at[a].kDOP));
        double VPtot = VPO4 + VDOP;
        KCA(P_lim, k, c, a) = VPtot;
        f_nut = dmin(f_nut, VPtot);
        if (at[a].kSiO3 > 0.0) {
          double VSiO3 = SiO3_loc / (SiO3_loc + at[a].kSiO3);
          KCA(SiO3_lim, k, c, a) = VSiO3;
          f_nut = dmin(f_nut, VSiO3);
        }
        /* 1146-1173 */
        double PCmax = at[a].PCref * f_nut * Tfunc;
        if (T < at[a].temp_thres) PCmax = 0.0;
        if (at[a].temp_function == tfnc_quasi_mmrt) {
          if (lat[c] >= 0.0) BR(mmrt_north); else BR(mmrt_south);
          if (lat[c] >= 0.0) { tmpTopt = at[a].temp_optN; tmpTmax = at[a].temp_thresN; }
          else { tmpTopt = at[a].temp_optS; tmpTmax = at[a].temp_thresS; }
          PCmax = PCmax * dmin(1.0, ((tmpTmax - T) / (tmpTmax - tmpTopt)));
          if (T > tmpTmax) PCmax = 0.0;
        }
        /* 1175-1181 */
        double light_lim = (1.0 - exp((-1.0 * at[a].alphaPI * thetaC[a] * PAR_avg) / (PCmax + epsTinv)));
        double PCphoto = PCmax * light_lim;
        KCA(light_lim, k, c, a) = light_lim;
        photoC[a] = PCphoto * aC[a];
        /* 1193-1221 */
        double VNC;
        if (VNtot[a] > 0.0) {
          NO3_V[a] = (VNO3[a] / VNtot[a]) * photoC[a] * Q;
          NH4_V[a] = (VNH4[a] / VNtot[a]) * photoC[a] * Q;
          VNC = PCphoto * Q;
        } else { NO3_V[a] = 0.0; NH4_V[a] = 0.0; VNC = 0.0; }
        KCA(photoNO3, k, c, a) = NO3_V[a];
        KCA(photoNH4, k, c, a) = NH4_V[a];
        if (VPtot > 0.0) {
          PO4_V[a] = (VPO4 / VPtot) * photoC[a] * at[a].Qp;
          DOP_V[a] = (VDOP / VPtot) * photoC[a] * at[a].Qp;
        } else { PO4_V[a] = 0.0; DOP_V[a] = 0.0; }
        KCA(PO4_uptake, k, c, a) = PO4_V[a];
        KCA(DOP_uptake, k, c, a) = DOP_V[a];
        photoFe[a] = photoC[a] * gQfe[a];
        KCA(photoFe, k, c, a) = photoFe[a];
        /* 1227-1232 */
        if (at[a].Si_ind > 0) {
          photoSi[a] = photoC[a] * gQsi[a];
          KCA(bSi_form, k, c, a) = photoSi[a];
          a_tot_bSi_form = a_tot_bSi_form + photoSi[a];
        }
        /* 1240-1246 */
        work1 = at[a].alphaPI * thetaC[a] * PAR_avg;
        if (work1 > 0.0) {
          double pChl = at[a].thetaN_max * PCphoto / work1;
          photoacc[a] = (pChl * VNC / thetaC[a]) * aChl[a];
        } else photoacc[a] = 0.0;
        /* 1255-1278 */
        if (at[a].imp_calcifier) {
          CaCO3_PROD[a] = p->parm_f_prod_sp_CaCO3 * photoC[a];
          CaCO3_PROD[a] = CaCO3_PROD[a] * f_nut;
          if (T < CaCO3_temp_thres1) BR(caco3_cold);
          if (aC[a] > CaCO3_sp_thres) BR(caco3_bloom);
          if (T < CaCO3_temp_thres1)
            CaCO3_PROD[a] = CaCO3_PROD[a] * dmax((T - CaCO3_temp_thres2), 0.0) / (CaCO3_temp_thres1 - CaCO3_temp_thres2);
          if (aC[a] > CaCO3_sp_thres)
            CaCO3_PROD[a] = dmin((CaCO3_PROD[a] * aC[a] / CaCO3_sp_thres), (f_photosp_CaCO3 * photoC[a]));
          KCA(CaCO3_form, k, c, a) = CaCO3_PROD[a];
          d_tot_CaCO3_form = d_tot_CaCO3_form + CaCO3_PROD[a];
          work1 = dz * CaCO3_PROD[a];
          a_CaCO3_form_zint[a] = a_CaCO3_form_zint[a] + work1;
          a_tot_CaCO3_form_zint = a_tot_CaCO3_form_zint + work1;
        }
        /* 1285-1290 */
        auto_loss[a] = at[a].mort * Pprime[a] * Tfunc;
        auto_agg[a] = dmin((at[a].agg_rate_max * DPS) * Pprime[a], at[a].mort2 * Pprime[a] * Pprime[a]);
        auto_agg[a] = dmax((at[a].agg_rate_min * DPS) * Pprime[a], auto_agg[a]);
        /* 1297-1324 */
        work1 = 0.0;
        for (int a2 = 0; a2 < 4; ++a2)
          if (at[a2].grazee_ind == at[a].grazee_ind) work1 = work1 + Pprime[a2];
        double z_umax = at[a].z_umax_0 * Tfunc;
        if (a + 1 == p->diat_ind) {
          if ((lat[c] >= 0.0) && (T > at[a].temp_optN)) BR(zumax_taper_north);
          else if ((lat[c] <= 0.0) && (T > at[a].temp_optS)) BR(zumax_taper_south);
          if ((lat[c] >= 0.0) && (T > at[a].temp_optN))
            z_umax = z_umax * dmax((at[a].temp_thresN - T) / (at[a].temp_thresN - at[a].temp_optN), 0.95);
          else if ((lat[c] <= 0.0) && (T > at[a].temp_optS))
            z_umax = z_umax * dmax((at[a].temp_thresS - T) / (at[a].temp_thresS - at[a].temp_optS), 0.95);
        }
        if (work1 > 0.0)
          auto_graze[a] = (Pprime[a] / work1) * z_umax * zooC_loc * (work1 / (work1 + at[a].z_grz));
        else auto_graze[a] = 0.0;
        /* 1331-1338 */
        if (at[a].Nfixer) {
          work1 = photoC[a] * Q;
          Nfix[a] = (work1 * r_Nfix_photo) - NO3_V[a] - NH4_V[a];
          Nexcrete[a] = Nfix[a] + NO3_V[a] + NH4_V[a] - work1;
          KCA(Nfix, k, c, a) = Nfix[a];
          d_tot_Nfix = d_tot_Nfix + Nfix[a];
        }
        /* 1354-1372 */
        auto_graze_zoo[a] = at[a].graze_zoo * auto_graze[a];
        if (at[a].imp_calcifier)
          auto_graze_poc[a] = auto_graze[a] * dmax((caco3_poc_min * QCaCO3[a]),
                                                   dmin(spc_poc_fac * dmax(1.0, Pprime[a]), f_graze_sp_poc_lim));
        else auto_graze_poc[a] = at[a].graze_poc * auto_graze[a];
        auto_graze_doc[a] = at[a].graze_doc * auto_graze[a];
        auto_graze_dic[a] = auto_graze[a] - (auto_graze_zoo[a] + auto_graze_poc[a] + auto_graze_doc[a]);
        if (at[a].imp_calcifier) auto_loss_poc[a] = QCaCO3[a] * auto_loss[a];
        else auto_loss_poc[a] = at[a].loss_poc * auto_loss[a];
        auto_loss_doc[a] = (1.0 - p->parm_labile_ratio) * (auto_loss[a] - auto_loss_poc[a]);
        auto_loss_dic[a] = p->parm_labile_ratio * (auto_loss[a] - auto_loss_poc[a]);
        /* 1380-1386 */
        if (at[a].Qp != Qp_zoo_pom) {
          double remaining_P = ((auto_graze[a] + auto_loss[a] + auto_agg[a]) * at[a].Qp) -
                               ((auto_graze_zoo[a]) * Qp_zoo_pom) -
                               ((auto_graze_poc[a] + auto_loss_poc[a] + auto_agg[a]) * Qp_zoo_pom);
          remaining_P_dop[a] = (1.0 - p->parm_labile_ratio) * remaining_P;
          remaining_P_dip[a] = p->parm_labile_ratio * remaining_P;
        }
      }
      KC(tot_CaCO3_form, k, c) = d_tot_CaCO3_form;
      KC(tot_Nfix, k, c) = d_tot_Nfix;

      /* 1395-1401 */
      work1 = 0.0; work2 = 0.0;
      for (int a = 0; a < 4; ++a) {
        work1 = work1 + at[a].f_zoo_detr * (auto_graze[a] + epsC * epsTinv);
        work2 = work2 + (auto_graze[a] + epsC * epsTinv);
      }
      double f_zoo_detr = work1 / work2;
      /* 1408-1415 */
      C_loss_thres = f_loss_thres * loss_thres_zoo;
      double Zprime = dmax(zooC_loc - C_loss_thres, 0.0);
      double zoo_loss = (p->parm_z_mort2_0 * pow(Zprime, 1.5) + p->parm_z_mort_0 * Zprime) * Tfunc;
      double zoo_loss_doc = (1.0 - p->parm_labile_ratio) * (1.0 - f_zoo_detr) * zoo_loss;
      double zoo_loss_dic = p->parm_labile_ratio * (1.0 - f_zoo_detr) * zoo_loss;
#define SUM4(v) ((((v)[0] + (v)[1]) + (v)[2]) + (v)[3])
      /* 1421-1434 */
      double DOC_prod = zoo_loss_doc + SUM4(auto_loss_doc) + SUM4(auto_graze_doc);
      double DON_prod = Q * DOC_prod;
      double DOP_prod = Qp_zoo_pom * zoo_loss_doc;
      for (int a = 0; a < 4; ++a) {
        if (at[a].Qp == Qp_zoo_pom) DOP_prod = DOP_prod + at[a].Qp * (auto_loss_doc[a] + auto_graze_doc[a]);
        else DOP_prod = DOP_prod + remaining_P_dop[a];
      }
      double DOFe_prod = Qfe_zoo * zoo_loss_doc;
      for (int a = 0; a < 4; ++a) DOFe_prod = DOFe_prod + Qfe[a] * (auto_loss_doc[a] + auto_graze_doc[a]);
      /* 1436-1461 */
      double DOC_remin = DOC_loc * DOC_reminR;
      double DON_remin = DON_loc * DON_reminR;
      double DOFe_remin = DOFe_loc * DOFe_reminR;
      double DOP_remin = DOP_loc * DOP_reminR;
      double DONr_remin, DOPr_remin;
      if (!(PAR_avg > 1.0)) BR(par_avg_dark);
      if (PAR_avg > 1.0) {
        DONr_remin = DONr_loc * DONr_reminR;
        DOPr_remin = DOPr_loc * DOPr_reminR;
      } else {
        DONr_remin = DONr_loc * (1.0 / (365.0 * 670.0)) * DPS;
        DOPr_remin = DOPr_loc * (1.0 / (365.0 * 460.0)) * DPS;
        DOC_remin = DOC_remin * 0.0685;
        DON_remin = DON_remin * 0.1;
        DOFe_remin = DOFe_remin * 0.05;
        DOP_remin = DOP_remin * 0.05;
      }
      /* 1467-1500 */
      POC.prod = f_zoo_detr * zoo_loss + SUM4(auto_graze_poc) + SUM4(auto_agg) + SUM4(auto_loss_poc);
      for (int a = 0; a < 4; ++a)
        if (at[a].CaCO3_ind > 0)
          P_CaCO3.prod = ((1.0 - f_graze_CaCO3_remin) * auto_graze[a] + auto_loss[a] + auto_agg[a]) * QCaCO3[a];
      for (int a = 0; a < 4; ++a)
        if (at[a].Si_ind > 0)
          P_SiO2.prod = Qsi[a] * ((1.0 - f_graze_si_remin) * auto_graze[a] + auto_agg[a] +
                                  at[a].loss_poc * auto_loss[a]);
      dust.prod = 0.0;
      /* 1510-1529: iron scavenging from the fluxes LEAVING the level above */
      double Fe_scavenge_rate = p->parm_fe_scavenge_rate0;
      Fe_scavenge_rate = Fe_scavenge_rate * ((POC.sflux_out + POC.hflux_out) * 120.1 +
                                             (P_CaCO3.sflux_out + P_CaCO3.hflux_out) * P_CaCO3.mass +
                                             (P_SiO2.sflux_out + P_SiO2.hflux_out) * P_SiO2.mass +
                                             (dust.sflux_out + dust.hflux_out) * dust_fescav_scale);
      if (Fe_loc > fe_scavenge_thres1) BR(fe_scav_high);
      if (Fe_loc > fe_scavenge_thres1)
        Fe_scavenge_rate = Fe_scavenge_rate + (Fe_loc - fe_scavenge_thres1) * fe_max_scale2;
      double Fe_scavenge = YPS * Fe_loc * Fe_scavenge_rate;
      P_iron.prod = (zoo_loss * f_zoo_detr * Qfe_zoo) + Fe_scavenge;
      for (int a = 0; a < 4; ++a)
        P_iron.prod = P_iron.prod + Qfe[a] * (auto_agg[a] + auto_graze_poc[a] + auto_loss_poc[a]);

      /* 1531-1536 */
      compute_particulate_terms(p, k == kmax - 1, &POC, &P_CaCO3, &P_SiO2, &dust, &P_iron, &QA_dust_def, T, O2_loc,
                                NO3_loc, &SED_DENITRIF, &OTHER_REMIN, dz, zb, fesed[ix]);
      /* 2637-2694: its diagnostics */
      KC(POC_FLUX_IN, k, c) = POC.sflux_in + POC.hflux_in;
      KC(POC_PROD, k, c) = POC.prod;
      KC(POC_REMIN, k, c) = POC.remin;
      KC(CaCO3_FLUX_IN, k, c) = P_CaCO3.sflux_in + P_CaCO3.hflux_in;
      KC(CaCO3_PROD, k, c) = P_CaCO3.prod;
      KC(CaCO3_REMIN, k, c) = P_CaCO3.remin;
      KC(SiO2_FLUX_IN, k, c) = P_SiO2.sflux_in + P_SiO2.hflux_in;
      KC(SiO2_PROD, k, c) = P_SiO2.prod;
      KC(SiO2_REMIN, k, c) = P_SiO2.remin;
      KC(dust_FLUX_IN, k, c) = dust.sflux_in + dust.hflux_in;
      KC(dust_REMIN, k, c) = dust.remin;
      KC(P_iron_FLUX_IN, k, c) = P_iron.sflux_in + P_iron.hflux_in;
      KC(P_iron_PROD, k, c) = P_iron.prod;
      KC(P_iron_REMIN, k, c) = P_iron.remin;
      KC(calcToSed, k, c) = P_CaCO3.sed_loss;
      KC(bsiToSed, k, c) = P_SiO2.sed_loss;
      KC(pocToSed, k, c) = POC.sed_loss;
      KC(SedDenitrif, k, c) = SED_DENITRIF * dz;
      KC(OtherRemin, k, c) = OTHER_REMIN * dz;
      KC(ponToSed, k, c) = (POC.sed_loss * Q);
      KC(popToSed, k, c) = (POC.sed_loss * Qp_zoo_pom);
      KC(dustToSed, k, c) = dust.sed_loss;
      KC(pfeToSed, k, c) = P_iron.sed_loss;

      /* 1545-1577 */
      double RESTORE = 0.0; /* lrest_* never set in the reference (131-134) */
      KC(NO3_RESTORE, k, c) = RESTORE;
      double NITRIF;
      if (PAR_out < p->parm_nitrif_par_lim) {
        NITRIF = p->parm_kappa_nitrif * NH4_loc;
        BR(nitrif_active);
        if (PAR_in > p->parm_nitrif_par_lim) BR(nitrif_taper);
        if (PAR_in > p->parm_nitrif_par_lim) NITRIF = NITRIF * log(PAR_out / p->parm_nitrif_par_lim) / (-KPARdz);
      } else NITRIF = 0.0;
      KC(NITRIF, k, c) = NITRIF;
      work1 = ((p->parm_o2_min + p->parm_o2_min_delta) - O2_loc) / p->parm_o2_min_delta;
      work1 = dmin(dmax(work1, 0.0), 1.0);
      work1 = (NO3_loc == 0.0) ? 0.0 : work1;
      double DENITRIF = work1 * ((DOC_remin + POC.remin - OTHER_REMIN) / denitrif_C_N - SED_DENITRIF);
      KC(DENITRIF, k, c) = DENITRIF;

      /* 1583-1592 */
      double t_no3 = RESTORE + NITRIF - DENITRIF - SED_DENITRIF - SUM4(NO3_V);
      double t_nh4 = -SUM4(NH4_V) - NITRIF + DON_remin + DONr_remin +
                     Q * (zoo_loss_dic + SUM4(auto_loss_dic) + SUM4(auto_graze_dic) + POC.remin * (1.0 - DONrefract));
      for (int a = 0; a < 4; ++a)
        if (at[a].Nfixer) t_nh4 = t_nh4 + Nexcrete[a];
      /* 1598-1605 */
      double t_fe = P_iron.remin + (Qfe_zoo * zoo_loss_dic) + DOFe_remin - SUM4(photoFe) - Fe_scavenge;
      for (int a = 0; a < 4; ++a)
        t_fe = t_fe + (Qfe[a] * (auto_loss_dic[a] + auto_graze_dic[a])) + auto_graze_zoo[a] * (Qfe[a] - Qfe_zoo);
      /* 1611-1628 */
      KC(SiO3_RESTORE, k, c) = RESTORE;
      double t_sio3 = RESTORE + P_SiO2.remin;
      for (int a = 0; a < 4; ++a)
        if (at[a].Si_ind > 0)
          t_sio3 = t_sio3 - photoSi[a] +
                   Qsi[a] * (f_graze_si_remin * auto_graze[a] + (1.0 - at[a].loss_poc) * auto_loss[a]);
      /* 1634-1661 */
      KC(PO4_RESTORE, k, c) = RESTORE;
      double t_po4 = RESTORE + DOP_remin + DOPr_remin - SUM4(PO4_V) +
                     Qp_zoo_pom * ((1.0 - DOPrefract) * POC.remin + zoo_loss_dic);
      for (int a = 0; a < 4; ++a) {
        if (at[a].Qp == Qp_zoo_pom) t_po4 = t_po4 + at[a].Qp * (auto_loss_dic[a] + auto_graze_dic[a]);
        else t_po4 = t_po4 + remaining_P_dip[a];
      }
      TEND(p->ind[ORA_no3], k, c) = t_no3;
      TEND(p->ind[ORA_nh4], k, c) = t_nh4;
      TEND(p->ind[ORA_fe], k, c) = t_fe;
      TEND(p->ind[ORA_sio3], k, c) = t_sio3;
      TEND(p->ind[ORA_po4], k, c) = t_po4;
      /* 1676-1697 */
      double t_C[4], t_Si[4] = {0, 0, 0, 0}, t_Ca[4] = {0, 0, 0, 0};
      for (int a = 0; a < 4; ++a) {
        work1 = auto_graze[a] + auto_loss[a] + auto_agg[a];
        t_C[a] = photoC[a] - work1;
        TEND(at[a].C_ind, k, c) = t_C[a];
        TEND(at[a].Chl_ind, k, c) = photoacc[a] - thetaC[a] * work1;
        TEND(at[a].Fe_ind, k, c) = photoFe[a] - Qfe[a] * work1;
        if (at[a].Si_ind > 0) { t_Si[a] = photoSi[a] - Qsi[a] * work1; TEND(at[a].Si_ind, k, c) = t_Si[a]; }
        if (at[a].CaCO3_ind > 0) { t_Ca[a] = CaCO3_PROD[a] - QCaCO3[a] * work1; TEND(at[a].CaCO3_ind, k, c) = t_Ca[a]; }
      }
      /* 1703-1723 */
      double t_zooC = SUM4(auto_graze_zoo) - zoo_loss;
      double t_doc = DOC_prod - DOC_remin;
      double t_don = (DON_prod * (1.0 - DONrefract)) - DON_remin;
      double t_donr = (DON_prod * DONrefract) - DONr_remin + (POC.remin * DONrefract * Q);
      double t_dop = (DOP_prod * (1.0 - DOPrefract)) - DOP_remin - SUM4(DOP_V);
      double t_dopr = (DOP_prod * DOPrefract) - DOPr_remin + (POC.remin * DOPrefract * Qp_zoo_pom);
      double t_dofe = DOFe_prod - DOFe_remin;
      TEND(p->ind[ORA_zooC], k, c) = t_zooC;
      TEND(p->ind[ORA_doc], k, c) = t_doc;
      TEND(p->ind[ORA_don], k, c) = t_don;
      TEND(p->ind[ORA_donr], k, c) = t_donr;
      TEND(p->ind[ORA_dop], k, c) = t_dop;
      TEND(p->ind[ORA_dopr], k, c) = t_dopr;
      TEND(p->ind[ORA_dofe], k, c) = t_dofe;
      /* 1729-1745 */
      double t_dic = SUM4(auto_loss_dic) + SUM4(auto_graze_dic) - SUM4(photoC) + DOC_remin + POC.remin +
                     zoo_loss_dic + P_CaCO3.remin;
      for (int a = 0; a < 4; ++a)
        if (at[a].CaCO3_ind > 0) t_dic = t_dic + f_graze_CaCO3_remin * auto_graze[a] * QCaCO3[a] - CaCO3_PROD[a];
      TEND(p->ind[ORA_dic], k, c) = t_dic;
      TEND(p->ind[ORA_dic_alt_co2], k, c) = alt_co2_use_eco ? t_dic : 0.0;
      /* 1751-1759 */
      double t_alk = -t_no3 + t_nh4 + 2.0 * P_CaCO3.remin;
      for (int a = 0; a < 4; ++a)
        if (at[a].CaCO3_ind > 0)
          t_alk = t_alk + 2.0 * (f_graze_CaCO3_remin * auto_graze[a] * QCaCO3[a] - CaCO3_PROD[a]);
      TEND(p->ind[ORA_alk], k, c) = t_alk;
      /* 1765-1790 */
      double O2_PRODUCTION = 0.0;
      for (int a = 0; a < 4; ++a) {
        if (!at[a].Nfixer) {
          if (photoC[a] > 0.0)
            O2_PRODUCTION = O2_PRODUCTION + photoC[a] * ((NO3_V[a] / (NO3_V[a] + NH4_V[a])) / parm_Red_D_C_O2 +
                                                         (NH4_V[a] / (NO3_V[a] + NH4_V[a])) / parm_Remin_D_C_O2);
        } else {
          if (photoC[a] > 0.0)
            O2_PRODUCTION = O2_PRODUCTION +
                            photoC[a] * ((NO3_V[a] / (NO3_V[a] + NH4_V[a] + Nfix[a])) / parm_Red_D_C_O2 +
                                         (NH4_V[a] / (NO3_V[a] + NH4_V[a] + Nfix[a])) / parm_Remin_D_C_O2 +
                                         (Nfix[a] / (NO3_V[a] + NH4_V[a] + Nfix[a])) / parm_Red_D_C_O2_diaz);
        }
      }
      work1 = (O2_loc - p->parm_o2_min) / p->parm_o2_min_delta;
      work1 = dmin(dmax(work1, 0.0), 1.0);
      double O2_CONSUMPTION =
          work1 * ((POC.remin + DOC_remin - (SED_DENITRIF * denitrif_C_N) - OTHER_REMIN + zoo_loss_dic +
                    SUM4(auto_loss_dic) + SUM4(auto_graze_dic)) / parm_Remin_D_C_O2 + (2.0 * NITRIF));
      TEND(p->ind[ORA_o2], k, c) = O2_PRODUCTION - O2_CONSUMPTION;

      /* 1796-1868 */
      KC(O2_PRODUCTION, k, c) = O2_PRODUCTION;
      KC(O2_CONSUMPTION, k, c) = O2_CONSUMPTION;
      work1 = o2sat(T, S, T0K);
      work1 = work1 - O2_loc;
      KC(AOU, k, c) = work1;
      KC(PAR_avg, k, c) = PAR_avg;
      KC(zoo_loss, k, c) = zoo_loss;
      KC(auto_graze_TOT, k, c) = SUM4(auto_graze);
      for (int a = 0; a < 4; ++a) {
        KCA(auto_graze, k, c, a) = auto_graze[a];
        KCA(auto_loss, k, c, a) = auto_loss[a];
        KCA(auto_agg, k, c, a) = auto_agg[a];
        KCA(photoC, k, c, a) = photoC[a];
        work1 = dz * photoC[a];
        a_photoC_zint[a] = a_photoC_zint[a] + work1;
      }
      work1 = SUM4(photoC);
      KC(photoC_TOT, k, c) = work1;
      work1 = work1 * dz;
      a_photoC_TOT_zint = a_photoC_TOT_zint + work1;
      double d_photoC_NO3_TOT = 0.0;
      for (int a = 0; a < 4; ++a) {
        if (VNtot[a] > 0.0) work1 = (VNO3[a] / VNtot[a]) * photoC[a];
        else work1 = 0.0;
        KCA(photoC_NO3, k, c, a) = work1;
        d_photoC_NO3_TOT = d_photoC_NO3_TOT + work1;
        work1 = work1 * dz;
        a_photoC_NO3_zint[a] = a_photoC_NO3_zint[a] + work1;
        /* 1844-1846: adds the RUNNING zint (quirk 13) */
        a_photoC_NO3_TOT_zint = a_photoC_NO3_TOT_zint + a_photoC_NO3_zint[a];
      }
      KC(photoC_NO3_TOT, k, c) = d_photoC_NO3_TOT;
      KC(DOC_prod, k, c) = DOC_prod; KC(DOC_remin, k, c) = DOC_remin;
      KC(DON_prod, k, c) = DON_prod; KC(DON_remin, k, c) = DON_remin;
      KC(DOP_prod, k, c) = DOP_prod; KC(DOP_remin, k, c) = DOP_remin;
      KC(DOFe_prod, k, c) = DOFe_prod; KC(DOFe_remin, k, c) = DOFe_remin;
      KC(Fe_scavenge, k, c) = Fe_scavenge; KC(Fe_scavenge_rate, k, c) = Fe_scavenge_rate;

      /* 1870-1945: conservation integrals */
      double ztop = 0.0;
      if (k > 0) ztop = zbot[ix - 1];
      work2 = dmin(100.0e2 - ztop, dz);
      double partial_thickness_100m = (work2 > 0.0) ? work2 : 0.0;
      int shallow = zb <= 100.0e2;
      work1 = t_dic + t_doc + t_zooC + (((t_C[0] + t_C[1]) + t_C[2]) + t_C[3]);
      for (int a = 0; a < 4; ++a)
        if (at[a].CaCO3_ind > 0) work1 = work1 + t_Ca[a];
      J_C = J_C + work1 * dz + POC.sed_loss + P_CaCO3.sed_loss;
      J_C100 = J_C100 + work1 * partial_thickness_100m + (shallow ? POC.sed_loss + P_CaCO3.sed_loss : 0.0);
      work1 = t_no3 + t_nh4 + t_don + t_donr + Q * t_zooC + Q * (((t_C[0] + t_C[1]) + t_C[2]) + t_C[3]);
      work1 = work1 + DENITRIF + SED_DENITRIF;
      for (int a = 0; a < 4; ++a)
        if (at[a].Nfixer) work1 = work1 - Nfix[a];
      J_N = J_N + work1 * dz + POC.sed_loss * Q;
      J_N100 = J_N100 + work1 * partial_thickness_100m + (shallow ? POC.sed_loss * Q : 0.0);
      work1 = t_po4 + t_dop + t_dopr + Qp_zoo_pom * t_zooC;
      for (int a = 0; a < 4; ++a) work1 = work1 + at[a].Qp * t_C[a];
      J_P = J_P + work1 * dz + POC.sed_loss * Qp_zoo_pom;
      J_P100 = J_P100 + work1 * partial_thickness_100m + (shallow ? POC.sed_loss * Qp_zoo_pom : 0.0);
      work1 = t_sio3;
      for (int a = 0; a < 4; ++a)
        if (at[a].Si_ind > 0) work1 = work1 + t_Si[a];
      J_Si = J_Si + work1 * dz + P_SiO2.sed_loss;
      J_Si100 = J_Si100 + work1 * partial_thickness_100m + (shallow ? P_SiO2.sed_loss : 0.0);
      for (int a = 0; a < 4; ++a) a_Chl100 = a_Chl100 + aChl[a] * partial_thickness_100m;
    } /* k */

    /* 1954-1968: O2 minimum */
    {
      double w2 = dmax(0.0, TR(p->ind[ORA_o2], 0, c)), w3 = zmid[(size_t)c * nz];
      for (int k = 1; k < kmax; ++k) {
        double w1 = dmax(0.0, TR(p->ind[ORA_o2], k, c));
        if (w1 < w2) { w2 = w1; w3 = zmid[(size_t)c * nz + k]; }
      }
      CC(O2_ZMIN, c) = w2;
      CC(O2_ZMIN_DEPTH, c) = w3;
    }
    CC(tot_bSi_form, c) = a_tot_bSi_form;
    CC(tot_CaCO3_form_zint, c) = a_tot_CaCO3_form_zint;
    CC(photoC_TOT_zint, c) = a_photoC_TOT_zint;
    CC(photoC_NO3_TOT_zint, c) = a_photoC_NO3_TOT_zint;
    CC(Chl_TOT_zint_100m, c) = a_Chl100;
    CC(Jint_Ctot, c) = J_C; CC(Jint_100m_Ctot, c) = J_C100;
    CC(Jint_Ntot, c) = J_N; CC(Jint_100m_Ntot, c) = J_N100;
    CC(Jint_Ptot, c) = J_P; CC(Jint_100m_Ptot, c) = J_P100;
    CC(Jint_Sitot, c) = J_Si; CC(Jint_100m_Sitot, c) = J_Si100;
    for (int a = 0; a < 4; ++a) {
      CA(CaCO3_form_zint, c, a) = a_CaCO3_form_zint[a];
      CA(photoC_zint, c, a) = a_photoC_zint[a];
      CA(photoC_NO3_zint, c, a) = a_photoC_NO3_zint[a];
    }
  } /* column */
#undef TR
#undef TEND
#undef KC
#undef KCA
#undef CA
#undef CC
}

/* ===================================================== BGC_SurfaceFluxes == */

void ora_bgc_surface_fluxes(const ora_bgc_params *p, int nz, int nc, const double *tracers, double *forcing,
                            double *depo, double *river, double *gas, double *seaice, double *net,
                            double *fdiag, int lcalc_o2, int lcalc_co2) {
  const size_t plane = (size_t)nz * nc;
#define TR1(slot, c) tracers[(size_t)((slot)-1) * plane + (size_t)(c) * nz]
#define FRC(name, c) forcing[(size_t)SF_##name * nc + (c)]
#define FD(name, c) fdiag[(size_t)FD_##name * nc + (c)]
#define FL(arr, slot, c) arr[(size_t)((slot)-1) * nc + (c)]
  memset(fdiag, 0, sizeof(double) * nc * ORA_NFLUX_DIAG);             /* 2789-2802 */
  for (int c = 0; c < nc; ++c) {
    /* 2816-2822 */
    double DIC_loc = dmax(0.0, TR1(p->ind[ORA_dic], c));
    double DIC_ALT_CO2_loc = dmax(0.0, TR1(p->ind[ORA_dic_alt_co2], c));
    double ALK_loc = dmax(0.0, TR1(p->ind[ORA_alk], c));
    double PO4_loc = dmax(0.0, TR1(p->ind[ORA_po4], c));
    double SiO3_loc = dmax(0.0, TR1(p->ind[ORA_sio3], c));
    double O2_loc = dmax(0.0, TR1(p->ind[ORA_o2], c));
    /* 2828-2838: in-place updates of the forcing */
    int fe = p->ind[ORA_fe];
    FL(depo, fe, c) = FL(depo, fe, c) * p->parm_Fe_bioavail;
    FL(river, fe, c) = FL(river, fe, c) * p->parm_Fe_bioavail;
    FL(gas, fe, c) = FL(gas, fe, c) * p->parm_Fe_bioavail;
    FL(seaice, fe, c) = FL(seaice, fe, c) * p->parm_Fe_bioavail;
    if (FRC(iceFraction, c) < 0.0) FRC(iceFraction, c) = 0.0;
    if (FRC(iceFraction, c) > 1.0) FRC(iceFraction, c) = 1.0;
    double xkw = xkw_coeff * FRC(windSpeedSquared10m, c);
    double xkw_ice = (1.0 - FRC(iceFraction, c)) * xkw;
    double SST = FRC(SST, c), SSS = FRC(SSS, c);
    if (lcalc_o2) {                                                  /* 2847-2860 */
      double SCHMIDT_O2 = schmidt_o2(SST);
      double O2SAT_1atm = o2sat(SST, SSS, p->T0_Kelvin_BGC);
      double pistonVel_O2 = xkw_ice * sqrt(660.0 / SCHMIDT_O2);
      double O2SAT = FRC(surfacePressure, c) * O2SAT_1atm;
      FL(gas, p->ind[ORA_o2], c) = pistonVel_O2 * (O2SAT - O2_loc);
      FD(pistonVel_O2, c) = pistonVel_O2; FD(SCHMIDT_O2, c) = SCHMIDT_O2; FD(O2SAT, c) = O2SAT;
      FD(xkw, c) = xkw_ice;
    }
    if (lcalc_co2) {                                                 /* 2866-2923 */
      double SCHMIDT_CO2 = schmidt_co2(SST);
      double pistonVel_CO2 = xkw_ice * sqrt(660.0 / SCHMIDT_CO2);
      double phlo, phhi, ph_new, co2star, dco2star, pco2surf, dpco2;
      if (FRC(surface_pH, c) != 0.0) { phlo = FRC(surface_pH, c) - del_ph; phhi = FRC(surface_pH, c) + del_ph; }
      else { phlo = phlo_surf_init; phhi = phhi_surf_init; }
      double depth = FRC(surfaceDepth, c);
      co2calc_1point(depth, SST, SSS, DIC_loc, ALK_loc, PO4_loc, SiO3_loc, phlo, phhi, &ph_new, FRC(atmCO2, c),
                     FRC(surfacePressure, c), &co2star, &dco2star, &pco2surf, &dpco2);
      FRC(surface_pH, c) = ph_new;
      FL(gas, p->ind[ORA_dic], c) = pistonVel_CO2 * dco2star;
      FD(co2star, c) = co2star; FD(dco2star, c) = dco2star; FD(pco2surf, c) = pco2surf; FD(dpco2, c) = dpco2;
      FD(pistonVel_CO2, c) = pistonVel_CO2; FD(SCHMIDT_CO2, c) = SCHMIDT_CO2;
      if (FRC(surface_pH_alt_co2, c) != 0.0) {
        phlo = FRC(surface_pH_alt_co2, c) - del_ph; phhi = FRC(surface_pH_alt_co2, c) + del_ph;
      } else { phlo = phlo_surf_init; phhi = phhi_surf_init; }
      co2calc_1point(depth, SST, SSS, DIC_ALT_CO2_loc, ALK_loc, PO4_loc, SiO3_loc, phlo, phhi, &ph_new,
                     FRC(atmCO2_ALT_CO2, c), FRC(surfacePressure, c), &co2star, &dco2star, &pco2surf, &dpco2);
      FRC(surface_pH_alt_co2, c) = ph_new;
      FL(gas, p->ind[ORA_dic_alt_co2], c) = pistonVel_CO2 * dco2star;
      FD(co2star_alt_co2, c) = co2star; FD(dco2star_alt_co2, c) = dco2star;
      FD(pco2surf_alt_co2, c) = pco2surf; FD(dpco2_alt_co2, c) = dpco2;
    }
    /* 2929-2942 */
    for (int n = 1; n <= ORA_BGC_NTRACER; ++n)
      FL(net, n, c) = FL(depo, n, c) + FL(gas, n, c) + FL(river, n, c) + FL(seaice, n, c);
    FL(net, p->ind[ORA_alk], c) = FL(net, p->ind[ORA_alk], c) + FL(net, p->ind[ORA_nh4], c) - FL(net, p->ind[ORA_no3], c);
  }
#undef TR1
#undef FRC
#undef FD
}

/* ================================================================== DMS == */

enum { D_dms, D_dmsp, D_no3, D_doc, D_zooC, D_spC, D_spCaCO3, D_diatC, D_diazC, D_phaeoC, D_spChl, D_diatChl,
       D_diazChl, D_phaeoChl };

void ora_dms_source_sink(const ora_dms_params *p, int nz, int nc, const double *tracers, const double *dzv,
                         const int *kmaxv, const double *sst, const double *sw, double *tend, double *dkc) {
  const size_t plane = (size_t)nz * nc;
  const double epsC_d = 1.00e-8;                                       /* DMS_parms.F90:194-195 */
  double dummy;
#define TRD(j, k, c) dmax(0.0, tracers[(size_t)(p->ind[j] - 1) * plane + (size_t)(c) * nz + (k)])
#define DK(name, k, c) (*(dkc ? &dkc[(size_t)DDKC_##name * plane + (size_t)(c) * nz + (k)] : &dummy))
  memset(tend, 0, sizeof(double) * plane * ORA_DMS_NTRACER);           /* 413 */
  for (int c = 0; c < nc; ++c) {
    int kmax = kmaxv[c];
    if (kmax < 1) continue;
    double SST_loc = sst[c];                                          /* 504-510 */
    double PAR_out = dmax(0.0, sw[c]);
    PAR_out = PAR_out * p->f_qsw_par_DMS;
    double UV_out = PAR_out * 0.01;
    for (int k = 0; k < kmax; ++k) {
      const size_t ix = (size_t)c * nz + k;
      double dz = dzv[ix];
      double zooC_loc = TRD(D_zooC, k, c), DOC_loc = TRD(D_doc, k, c), spC_loc = TRD(D_spC, k, c),
             diatC_loc = TRD(D_diatC, k, c), diazC_loc = TRD(D_diazC, k, c), phaeoC_loc = TRD(D_phaeoC, k, c),
             spChl_loc = TRD(D_spChl, k, c), diatChl_loc = TRD(D_diatChl, k, c), diazChl_loc = TRD(D_diazChl, k, c),
             phaeoChl_loc = TRD(D_phaeoChl, k, c), spCaCO3_loc = TRD(D_spCaCO3, k, c), DMS_loc = TRD(D_dms, k, c),
             DMSP_loc = TRD(D_dmsp, k, c);
      double k_S_p = p->k_S_p_base * (p->mort + (zooC_loc / 0.3));     /* 529 */
      double UV_in = UV_out;                                          /* 531-536 (UV_avg unused) */
      double KUVdz = (0.01e-2 * DOC_loc + 0.04e-4) * dz;
      UV_out = UV_in * exp(-KUVdz);
      double PAR_in = PAR_out;                                        /* 538-551 */
      double totalChl = spChl_loc + diatChl_loc + diazChl_loc + phaeoChl_loc;
      double work = dmax(totalChl, 0.02), KPARdz;
      if (work < 0.13224) KPARdz = 0.000919 * pow(work, 0.3536);
      else KPARdz = 0.001131 * pow(work, 0.4562);
      KPARdz = KPARdz * dz;
      PAR_out = PAR_in * exp(-KPARdz);
      double PAR_avg = PAR_in * (1.0 - exp(-KPARdz)) / KPARdz;
      double j_dms = p->j_dms_perI * PAR_avg;                          /* 562 */
      double Fcocco = spCaCO3_loc / (spC_loc + epsC_d);                /* 570-573 */
      if (Fcocco > 0.4) Fcocco = 0.4;
      double Cocco_frac = Fcocco;
      double T_ind = (SST_loc - p->T_lo) / (p->T_hi - p->T_lo);        /* 584-592 */
      if (T_ind <= 0.0) T_ind = 0.0;
      if (T_ind >= 1.0) T_ind = 1.0;
      double Cyano_frac = (T_ind * (p->Max_cyano_frac - p->Min_cyano_frac)) + p->Min_cyano_frac;
      Cyano_frac = (1.0 - Cocco_frac) * Cyano_frac;
      double Eukar_frac = 1.0 - Cocco_frac - Cyano_frac;
      double diatN_loc = p->R * diatC_loc;                             /* 598-612 */
      double phaeoN_loc = p->R * phaeoC_loc;
      double coccoN_loc = Cocco_frac * p->R * spC_loc;
      double cyanoN_loc = Cyano_frac * p->R * spC_loc;
      double eukarN_loc = Eukar_frac * p->R * spC_loc;
      double diazN_loc = p->R * diazC_loc;
      double zooN_loc = p->R * zooC_loc;
      double phytoN_loc = diatN_loc + coccoN_loc + cyanoN_loc + eukarN_loc + diazN_loc + phaeoN_loc;
      double Sp_dec = (p->Sp_ref - spChl_loc) / p->Sp_ref;             /* 621-628 */
      if (Sp_dec <= 0.0) Sp_dec = 0.0;
      if (Sp_dec >= 1.0) Sp_dec = 1.0;
      double Stress_fac = 1.0 + p->Stress_mult * Sp_dec * Sp_dec;
      if (Stress_fac >= 10.0) Stress_fac = 10.0;
      double yield = (T_ind * (p->Max_yld - p->Min_yld)) + p->Min_yld; /* 637-640 */
      if (SST_loc < p->T_cryo_hi && SST_loc > p->T_cryo_lo) yield = 0.5;
      if (SST_loc < -1.0) yield = 0.25;
      double diatS_loc = p->Rs2n_diat * diatN_loc;                     /* 647-660 */
      double phaeoS_loc = p->Rs2n_phaeo * phaeoN_loc;
      double coccoS_loc = p->Rs2n_cocco * coccoN_loc;
      double cyanoS_loc = p->Rs2n_cyano * cyanoN_loc;
      double eukarS_loc = p->Rs2n_eukar * eukarN_loc * Stress_fac;
      double diazS_loc = p->Rs2n_diaz * diazN_loc;
      double phytoS_loc = diatS_loc + coccoS_loc + cyanoS_loc + eukarS_loc + diazS_loc + p->G_phaeo_S * phaeoS_loc;
      double Rs2n_zoo;                                                /* 671-684 */
      if (phytoN_loc > 0.0)
        Rs2n_zoo = (p->Rs2n_diat * diatN_loc + p->G_phaeo_S * p->Rs2n_phaeo * phaeoN_loc +
                    p->Rs2n_cocco * coccoN_loc + p->Rs2n_cyano * cyanoN_loc +
                    p->Rs2n_eukar * eukarN_loc * Stress_fac + p->Rs2n_diaz * diazN_loc) / phytoN_loc;
      else
        Rs2n_zoo = (p->Rs2n_diat + p->Rs2n_cocco + p->Rs2n_cyano + p->Rs2n_eukar + p->Rs2n_diaz + p->Rs2n_phaeo) /
                   (double)6.0f;
      double zooS_loc = Rs2n_zoo * zooN_loc;
      double B_diagnosed = p->B_preexp * pow(phytoN_loc, p->B_exp);    /* 695 */
      double dms_s_dmsp = yield * p->k_conv * DMSP_loc;                /* 701-719 */
      double dms_s = dms_s_dmsp;
      double dms_r_B = p->k_S_B * B_diagnosed * DMS_loc;
      double dms_r_phot = j_dms * DMS_loc;
      double dms_r_bkgnd = p->k_bkgnd * DMS_loc;
      double dms_r = dms_r_B + dms_r_phot + dms_r_bkgnd;
      double dmsp_s_phaeo = p->inject_scale * p->k_S_p_base * phaeoS_loc;
      double dmsp_s_nonphaeo = p->inject_scale * k_S_p * phytoS_loc;
      double dmsp_s_zoo = p->inject_scale * p->k_S_z * zooS_loc;
      double dmsp_s = dmsp_s_phaeo + dmsp_s_nonphaeo + dmsp_s_zoo;
      double dmsp_r_B = p->k_conv * DMSP_loc;
      double dmsp_r_bkgnd = p->k_bkgnd * DMSP_loc;
      double dmsp_r = dmsp_r_B + dmsp_r_bkgnd;
      tend[(size_t)(p->ind[D_dms] - 1) * plane + ix] = dms_s - dms_r;
      tend[(size_t)(p->ind[D_dmsp] - 1) * plane + ix] = dmsp_s - dmsp_r;
      DK(DMS_S_DMSP, k, c) = dms_s_dmsp; DK(DMS_S_TOTAL, k, c) = dms_s;      /* 723-761 */
      DK(DMS_R_B, k, c) = dms_r_B; DK(DMS_R_PHOT, k, c) = dms_r_phot; DK(DMS_R_BKGND, k, c) = dms_r_bkgnd;
      DK(DMS_R_TOTAL, k, c) = dms_r; DK(DMSP_S_PHAEO, k, c) = dmsp_s_phaeo;
      DK(DMSP_S_NONPHAEO, k, c) = dmsp_s_nonphaeo; DK(DMSP_S_ZOO, k, c) = dmsp_s_zoo;
      DK(DMSP_S_TOTAL, k, c) = dmsp_s; DK(DMSP_R_B, k, c) = dmsp_r_B; DK(DMSP_R_BKGND, k, c) = dmsp_r_bkgnd;
      DK(DMSP_R_TOTAL, k, c) = dmsp_r; DK(Cyano_frac, k, c) = Cyano_frac; DK(Cocco_frac, k, c) = Cocco_frac;
      DK(Eukar_frac, k, c) = Eukar_frac; DK(diatS, k, c) = diatS_loc; DK(diatN, k, c) = diatN_loc;
      DK(phytoN, k, c) = phytoN_loc; DK(coccoS, k, c) = coccoS_loc; DK(cyanoS, k, c) = cyanoS_loc;
      DK(eukarS, k, c) = eukarS_loc; DK(diazS, k, c) = diazS_loc; DK(phaeoS, k, c) = phaeoS_loc;
      DK(zooS, k, c) = zooS_loc; DK(zooCC, k, c) = zooC_loc; DK(RSNzoo, k, c) = Rs2n_zoo;
    }
  }
#undef TRD
#undef DK
}

/* SCHMIDT_DMS_singleValue, DMS_mod.F90:915-959 */
static double schmidt_dms(double SST) {
  const double a = 2674.0, b = 147.12, c = 3.726, d = 0.038;
  return a + SST * (-b + SST * (c + SST * (-d)));
}

/* DMS_SurfaceFluxes, DMS_mod.F90:778-908 (DMSSAT == 0, 1003) */
void ora_dms_surface_fluxes(const ora_dms_params *p, int nz, int nc, const double *tracers, double *forcing,
                            double *net, double *fdiag, int lcalc) {
  const size_t plane = (size_t)nz * nc;
  const double a = 0.31, e2 = 2.85, e3 = 0.612;
  if (!lcalc) return;
#define FRC(name, c) forcing[(size_t)DSF_##name * nc + (c)]
#define FD(name, c) fdiag[(size_t)DFD_##name * nc + (c)]
  for (int c = 0; c < nc; ++c) {
    double seaSurfaceDMS = dmax(0.0, tracers[(size_t)(p->ind[D_dms] - 1) * plane + (size_t)c * nz]);
    double seaSurfaceTemp = FRC(SST, c);
    if (FRC(iceFraction, c) < 0.0) FRC(iceFraction, c) = 0.0;
    if (FRC(iceFraction, c) > 1.0) FRC(iceFraction, c) = 1.0;
    double SCHMIDT_DMS = schmidt_dms(seaSurfaceTemp);
    double WIND_SPEED = sqrt(fabs(FRC(windSpeedSquared10m, c))) * 0.01;
    double XKW_W92 = a * (pow((660.0 / SCHMIDT_DMS), 0.500)) * WIND_SPEED * WIND_SPEED;
    double XKW_LM86 = e2 * (pow((600.0 / SCHMIDT_DMS), 0.500)) * (WIND_SPEED - 3.6) +
                      e3 * (pow((600.0 / SCHMIDT_DMS), 0.667));
    double xkw = 0.0;
    if (WIND_SPEED < 3.6) xkw = XKW_W92;
    if ((WIND_SPEED >= 3.6) && (WIND_SPEED < 5.6)) {
      double FLM86 = 0.5 * (WIND_SPEED - 3.6);
      double FW92 = 1.0 - FLM86;
      xkw = FW92 * XKW_W92 + FLM86 * XKW_LM86;
    }
    if (WIND_SPEED >= 5.6) xkw = XKW_LM86;
    xkw = xkw / 3600.0;
    double xkw_ice = (1.0 - FRC(iceFraction, c)) * xkw;
    double DMSSAT_1atm = 0.0;
    double pistonVel_DMS = xkw_ice * sqrt(660.0 / SCHMIDT_DMS);
    double DMSSAT = FRC(surfacePressure, c) * DMSSAT_1atm;
    net[(size_t)(p->ind[D_dms] - 1) * nc + c] = pistonVel_DMS * (DMSSAT - seaSurfaceDMS);
    net[(size_t)(p->ind[D_dmsp] - 1) * nc + c] = 0.0;
    FD(DMS_IFRAC, c) = FRC(iceFraction, c); FD(DMS_XKW, c) = xkw_ice;
    FD(DMS_ATM_PRESS, c) = FRC(surfacePressure, c); FD(DMS_PV, c) = pistonVel_DMS;
    FD(DMS_SCHMIDT, c) = SCHMIDT_DMS; FD(DMS_SAT, c) = DMSSAT; FD(DMS_SURF, c) = seaSurfaceDMS;
    FD(DMS_WS, c) = WIND_SPEED;
  }
#undef FRC
#undef FD
}

/* =============================================================== MACROS == */

enum { M_prot, M_poly, M_lip, M_zooC, M_spC, M_diatC, M_diazC, M_phaeoC };

/* MACROS_SourceSink, MACROS_mod.F90:137-411 */
void ora_macros_source_sink(const ora_macros_params *p, int nz, int nc, const double *tracers, const int *kmaxv,
                            double *tend, double *dkc) {
  const size_t plane = (size_t)nz * nc;
  double dummy;
#define TRM(j, k, c) dmax(0.0, tracers[(size_t)(p->ind[j] - 1) * plane + (size_t)(c) * nz + (k)])
#define DK(name, k, c) (*(dkc ? &dkc[(size_t)MDKC_##name * plane + (size_t)(c) * nz + (k)] : &dummy))
  memset(tend, 0, sizeof(double) * plane * ORA_MACROS_NTRACER);        /* 267 */
  for (int c = 0; c < nc; ++c) {
    int kmax = kmaxv[c];
    if (kmax < 1) continue;
    for (int k = 0; k < kmax; ++k) {
      const size_t ix = (size_t)c * nz + k;
      double zooC_loc = TRM(M_zooC, k, c);
      double k_C_p = p->k_C_p_base * (p->mort + (zooC_loc / p->zooC_avg));   /* 349 */
      double spCk = TRM(M_spC, k, c), diatCk = TRM(M_diatC, k, c), phaeoCk = TRM(M_phaeoC, k, c),
             diazCk = TRM(M_diazC, k, c);
      double phytoC_loc = diatCk + phaeoCk + spCk + diazCk;                   /* 366 */
      double prot_s = p->inject_scale * p->f_prot * k_C_p * phytoC_loc;       /* 372-378 */
      double poly_s = p->inject_scale * p->f_poly * k_C_p * phytoC_loc;
      double lip_s = p->inject_scale * p->f_lip * k_C_p * phytoC_loc;
      double prot_r = p->k_prot_bac * TRM(M_prot, k, c);
      double poly_r = p->k_poly_bac * TRM(M_poly, k, c);
      double lip_r = p->k_lip_bac * TRM(M_lip, k, c);
      tend[(size_t)(p->ind[M_prot] - 1) * plane + ix] = prot_s - prot_r;      /* 388-390 */
      tend[(size_t)(p->ind[M_poly] - 1) * plane + ix] = poly_s - poly_r;
      tend[(size_t)(p->ind[M_lip] - 1) * plane + ix] = lip_s - lip_r;
      DK(PROT_S_TOTAL, k, c) = prot_s; DK(POLY_S_TOTAL, k, c) = poly_s; DK(LIP_S_TOTAL, k, c) = lip_s;
      DK(PROT_R_TOTAL, k, c) = prot_r; DK(POLY_R_TOTAL, k, c) = poly_r; DK(LIP_R_TOTAL, k, c) = lip_r;
    }
  }
#undef TRM
#undef DK
}

/* explicit update of resident tracers (no reference counterpart; see obgc_oracle.h) */
void ora_apply_tendencies(int nz, int nc, int ntracer, double *tracers, const double *tend, const double *net_flux,
                          const double *dz, const int *kmax, double dt, int clamp_nonneg) {
  const size_t plane = (size_t)nz * nc;
  for (int n = 0; n < ntracer; ++n)
    for (int c = 0; c < nc; ++c) {
      int km = kmax[c] < nz ? kmax[c] : nz;
      for (int k = 0; k < km; ++k) {
        const size_t ix = (size_t)n * plane + (size_t)c * nz + k;
        double s = tend[ix];
        if (k == 0 && net_flux) s = s + net_flux[(size_t)n * nc + c] / dz[(size_t)c * nz];
        double v = tracers[ix] + dt * s;
        if (clamp_nonneg && !(v > 0.0)) v = 0.0;
        tracers[ix] = v;
      }
    }
}
