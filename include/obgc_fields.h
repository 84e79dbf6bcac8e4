/* Field lists of the ocean-BGC column engine C ABI (X-macros).
 *
 * Orders follow the declaration order of the reference's derived types so that a
 * Fortran shim can fill the pointer structs component by component:
 *   BGC_indices_type       /root/reference/BGC_parms.F90:81-112
 *   BGC_diagnostics_type   /root/reference/BGC_parms.F90:192-321
 *   BGC_flux_diagnostics   /root/reference/BGC_parms.F90:174-190
 *   DMS_indices_type       /root/reference/DMS_parms.F90:62-77
 *   DMS_diagnostics_type   /root/reference/DMS_parms.F90:125-154
 *   DMS_flux_diagnostics   /root/reference/DMS_parms.F90:113-123
 *   MACROS_indices_type    /root/reference/MACROS_parms.F90:62-71
 *   MACROS_diagnostics     /root/reference/MACROS_parms.F90:105-113
 */
#ifndef OBGC_FIELDS_H
#define OBGC_FIELDS_H

/* tracer slots of BGC_indices_type (member <name>_ind) */
#define OBGC_BGC_TRACERS(X) \
  X(po4) X(no3) X(sio3) X(nh4) X(fe) X(o2) X(dic) X(dic_alt_co2) X(alk) X(doc) X(don) X(dofe) \
  X(dop) X(dopr) X(donr) X(zooC) X(spC) X(spChl) X(spFe) X(spCaCO3) X(diatC) X(diatChl) \
  X(diatFe) X(diatSi) X(phaeoC) X(phaeoChl) X(phaeoFe) X(diazC) X(diazChl) X(diazFe)

/* BGC diagnostics dimensioned (k,column); reference member = diag_<name> */
#define OBGC_BGC_DIAG_KC(X) \
  X(tot_Nfix) X(O2_PRODUCTION) X(O2_CONSUMPTION) X(AOU) X(PO4_RESTORE) X(NO3_RESTORE) \
  X(SiO3_RESTORE) X(PAR_avg) X(POC_FLUX_IN) X(POC_PROD) X(POC_REMIN) X(POC_ACCUM) \
  X(CaCO3_FLUX_IN) X(CaCO3_PROD) X(CaCO3_REMIN) X(SiO2_FLUX_IN) X(SiO2_PROD) X(SiO2_REMIN) \
  X(dust_FLUX_IN) X(dust_REMIN) X(P_iron_FLUX_IN) X(P_iron_PROD) X(P_iron_REMIN) \
  X(auto_graze_TOT) X(zoo_loss) X(photoC_TOT) X(photoC_NO3_TOT) X(DOC_prod) X(DOC_remin) \
  X(DON_prod) X(DON_remin) X(DOFe_prod) X(DOFe_remin) X(DOP_prod) X(DOP_remin) X(Fe_scavenge) \
  X(Fe_scavenge_rate) X(NITRIF) X(DENITRIF) X(DONr_remin) X(DOPr_remin) X(CO3) X(HCO3) X(H2CO3) \
  X(pH_3D) X(CO3_ALT_CO2) X(HCO3_ALT_CO2) X(H2CO3_ALT_CO2) X(pH_3D_ALT_CO2) X(co3_sat_calc) \
  X(co3_sat_arag) X(calcToSed) X(pocToSed) X(ponToSed) X(popToSed) X(bsiToSed) X(dustToSed) \
  X(pfeToSed) X(SedDenitrif) X(OtherRemin) X(tot_CaCO3_form)
/* (k,column,autotroph) */
#define OBGC_BGC_DIAG_KCA(X) \
  X(N_lim) X(P_lim) X(Fe_lim) X(SiO3_lim) X(light_lim) X(photoC) X(photoC_NO3) X(photoFe) \
  X(photoNO3) X(photoNH4) X(DOP_uptake) X(PO4_uptake) X(auto_graze) X(auto_loss) X(auto_agg) \
  X(bSi_form) X(CaCO3_form) X(Nfix)
/* (column,autotroph) */
#define OBGC_BGC_DIAG_CA(X) X(photoC_zint) X(photoC_NO3_zint) X(CaCO3_form_zint)
/* (column) */
#define OBGC_BGC_DIAG_C(X) \
  X(photoC_TOT_zint) X(photoC_NO3_TOT_zint) X(Jint_Ctot) X(Jint_100m_Ctot) X(Jint_Ntot) \
  X(Jint_100m_Ntot) X(Jint_Ptot) X(Jint_100m_Ptot) X(Jint_Sitot) X(Jint_100m_Sitot) \
  X(Chl_TOT_zint_100m) X(tot_CaCO3_form_zint) X(tot_bSi_form) X(zsatcalc) X(zsatarag) \
  X(O2_ZMIN) X(O2_ZMIN_DEPTH)
/* BGC_flux_diagnostics_type (column); reference members carry no prefix */
#define OBGC_BGC_FLUX_DIAG(X) \
  X(pistonVel_O2) X(SCHMIDT_O2) X(O2SAT) X(xkw) X(co2star) X(dco2star) X(pco2surf) X(dpco2) \
  X(pistonVel_CO2) X(SCHMIDT_CO2) X(co2star_alt_co2) X(dco2star_alt_co2) X(pco2surf_alt_co2) \
  X(dpco2_alt_co2)

#define OBGC_DMS_TRACERS(X) \
  X(dms) X(dmsp) X(no3) X(doc) X(zooC) X(spC) X(spCaCO3) X(diatC) X(diazC) X(phaeoC) X(spChl) \
  X(diatChl) X(diazChl) X(phaeoChl)
#define OBGC_DMS_DIAG_KC(X) \
  X(DMS_S_DMSP) X(DMS_S_TOTAL) X(DMS_R_B) X(DMS_R_PHOT) X(DMS_R_BKGND) X(DMS_R_TOTAL) \
  X(DMSP_S_PHAEO) X(DMSP_S_NONPHAEO) X(DMSP_S_ZOO) X(DMSP_S_TOTAL) X(DMSP_R_B) X(DMSP_R_BKGND) \
  X(DMSP_R_TOTAL) X(Cyano_frac) X(Cocco_frac) X(Eukar_frac) X(diatS) X(diatN) X(phytoN) \
  X(coccoS) X(cyanoS) X(eukarS) X(diazS) X(phaeoS) X(zooS) X(zooCC) X(RSNzoo)
#define OBGC_DMS_FLUX_DIAG(X) \
  X(DMS_IFRAC) X(DMS_XKW) X(DMS_ATM_PRESS) X(DMS_PV) X(DMS_SCHMIDT) X(DMS_SAT) X(DMS_SURF) X(DMS_WS)
/* DMS_parms.F90:160-192 */
#define OBGC_DMS_PARMS(X) \
  X(k_S_p_base) X(zooC_avg) X(mort) X(k_conv) X(k_S_z) X(B_preexp) X(B_exp) X(k_S_B) X(k_bkgnd) \
  X(j_dms_perI) X(inject_scale) X(T_cryo_hi) X(T_cryo_lo) X(T_lo) X(T_hi) X(Min_cyano_frac) \
  X(Max_cyano_frac) X(Min_yld) X(Max_yld) X(G_phaeo_S) X(Sp_ref) X(Stress_mult) X(R) X(Rs2n_diat) \
  X(Rs2n_phaeo) X(Rs2n_cocco) X(Rs2n_cyano) X(Rs2n_eukar) X(Rs2n_diaz) X(f_qsw_par_DMS)

#define OBGC_MACROS_TRACERS(X) X(prot) X(poly) X(lip) X(zooC) X(spC) X(diatC) X(diazC) X(phaeoC)
#define OBGC_MACROS_DIAG_KC(X) \
  X(PROT_S_TOTAL) X(POLY_S_TOTAL) X(LIP_S_TOTAL) X(PROT_R_TOTAL) X(POLY_R_TOTAL) X(LIP_R_TOTAL)
/* MACROS_parms.F90:122-132 */
#define OBGC_MACROS_PARMS(X) \
  X(f_prot) X(f_poly) X(f_lip) X(k_C_p_base) X(zooC_avg) X(mort) X(k_prot_bac) X(k_poly_bac) \
  X(k_lip_bac) X(inject_scale)

/* BGC_parms.F90:346-361 */
#define OBGC_BGC_PARMS(X) \
  X(parm_Fe_bioavail) X(parm_o2_min) X(parm_o2_min_delta) X(parm_kappa_nitrif) \
  X(parm_nitrif_par_lim) X(parm_z_mort_0) X(parm_z_mort2_0) X(parm_labile_ratio) X(parm_POMbury) \
  X(parm_BSIbury) X(parm_fe_scavenge_rate0) X(parm_f_prod_sp_CaCO3) X(parm_POC_diss) \
  X(parm_SiO2_diss) X(parm_CaCO3_diss)
/* autotroph_type, BGC_parms.F90:51-78 */
#define OBGC_AUTOTROPH_INTS(X) \
  X(Nfixer) X(imp_calcifier) X(exp_calcifier) X(grazee_ind) X(temp_function) X(Chl_ind) X(C_ind) \
  X(Fe_ind) X(Si_ind) X(CaCO3_ind)
#define OBGC_AUTOTROPH_REALS(X) \
  X(kFe) X(kPO4) X(kDOP) X(kNO3) X(kNH4) X(kSiO3) X(Qp) X(gQfe_0) X(gQfe_min) X(alphaPI) X(PCref) \
  X(thetaN_max) X(loss_thres) X(loss_thres2) X(temp_thres) X(temp_thresS) X(temp_thresN) \
  X(temp_optN) X(temp_optS) X(mort) X(mort2) X(agg_rate_max) X(agg_rate_min) X(z_umax_0) X(z_grz) \
  X(graze_zoo) X(graze_poc) X(graze_doc) X(loss_poc) X(f_zoo_detr)

#endif
