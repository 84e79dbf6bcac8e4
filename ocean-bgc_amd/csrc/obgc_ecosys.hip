// Ecosystem source/sink terms + sinking-particle sweep of BGC_SourceSink: one lane per
// column, levels visited top to bottom.
//
// Behavioural contract = BGC_mod.F90:733-931 (clamps, zero-mask, quotas, light),
// 1041-1529 (kinetics), 1531-1536 -> 2116-2699 (Armstrong ballast model, bottom cell),
// 1545-1968 (tendencies, diagnostics, column integrals, O2 minimum) and 2006-2109
// (particle initialisation).  The carbonate block 935-1032 is NOT here: it has no
// vertical coupling and runs as its own cell-parallel kernel (obgc_carbonate.hip).
//
// What is carried down a column in registers: PAR leaving the level, ten particle
// fluxes, the QA-dust POC deficit and the running column integrals.  The reference keeps
// five 11-member structs and 21 (k,column[,4]) scratch arrays for the same purpose.
// Tracer planes are addressed through the caller's strides, so with the native COLFAST
// layout every load and store of a wavefront is one contiguous 512-byte row.
#include "obgc_internal.h"

namespace obgc {

namespace {

constexpr int kBlock = 64;   // one wavefront per workgroup: 100k columns -> 1563 workgroups over 256 CUs

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
constexpr double F_QSW_PAR = 0.45, TREF = 30.0, Q_10 = 1.5;             // 454-463
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

struct Flux { double s, h; };   // soft / hard sub-class flux leaving a level

__device__ __forceinline__ double sum4(const double (&v)[4]) { return ((v[0] + v[1]) + v[2]) + v[3]; }

// Garcia & Gordon (1992) O2 saturation in mmol/m^3 (BGC_mod.F90:3012-3083)
__device__ __forceinline__ double o2sat(double SST, double SSS, double T0K) {
  const double TS = log(((T0K + 25.0) - SST) / (T0K + SST));
  const double v = exp(2.00907 + TS * (3.22014 + TS * (4.05010 + TS * (4.94457 + TS * (-2.56847E-1 + TS * 3.88767)))) +
                       SSS * ((-6.24523E-3 + TS * (-7.37614E-3 + TS * (-1.03410E-2 + TS * -8.17083E-3))) + SSS * -4.88682E-7));
  return v / 0.0223916;
}

template <bool DIAG>
__global__ __launch_bounds__(kBlock) void bgc_ecosys_kernel(const BgcKernelArgs A, const obgc_bgc_params P) {
  const GridView& g = A.g;
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= g.ncol) return;
  const obgc_bgc_diag& D = A.diag;
  const obgc_autotroph* at = P.autotrophs;
  const double T0K = P.T0_Kelvin_BGC;
  const long long pl = g.plane;
  int kmax = A.kmax[c];
  if (kmax > g.nlev) kmax = g.nlev;

#define TRC(slot, ix) dmax(0.0, A.tracers[(long long)((slot)-1) * pl + (ix)])
#define TEND(slot, ix) A.tend[(long long)((slot)-1) * pl + (ix)]
#define ST_KC(name, v) do { if (DIAG && D.name) D.name[ix] = (v); } while (0)
#define ST_KCA(name, a, v) do { if (DIAG && D.name) D.name[(long long)(a) * pl + ix] = (v); } while (0)
#define ST_C(name, v) do { if (DIAG && D.name) D.name[c] = (v); } while (0)
#define ST_CA(name, a, v) do { if (DIAG && D.name) D.name[(long long)(a) * g.ldc + c] = (v); } while (0)

  // ---- per-column initialisation (BGC_mod.F90:808-814, 2075-2104) ----
  const double lat = A.lat[c];
  const bool north = lat >= 0.0, south_or_eq = lat <= 0.0;
  const double dust_in = (kmax >= 1) ? dmax(0.0, A.dust[c]) : 0.0;
  Flux fPOC{0.0, 0.0}, fCa{0.0, 0.0}, fSi{0.0, 0.0}, fFe{0.0, 0.0}, fDust{0.0, 0.0};
  if (dust_in != 0.0) { fDust.s = (1.0 - DUST_GAMMA) * dust_in; fDust.h = DUST_GAMMA * dust_in; }
  double QA_dust_def = DUST_RHO * (fDust.s + fDust.h);
  double PAR_out = (kmax >= 1) ? dmax(0.0, A.sw[c]) * F_QSW_PAR : 0.0;
  double prodCa = 0.0, prodSi = 0.0;   // keep their value when no autotroph calcifies / silicifies (1480-1498)

  // running column diagnostics
  double z_bSi = 0.0, z_CaTot = 0.0, z_pCTot = 0.0, z_pCNO3Tot = 0.0, z_Chl100 = 0.0;
  double z_Ca[4] = {0, 0, 0, 0}, z_pC[4] = {0, 0, 0, 0}, z_pCNO3[4] = {0, 0, 0, 0};
  double J_C = 0, J_C100 = 0, J_N = 0, J_N100 = 0, J_P = 0, J_P100 = 0, J_Si = 0, J_Si100 = 0;
  double o2min = 0.0, o2min_z = 0.0, zbot_above = 0.0;

  for (int k = 0; k < kmax; ++k) {
    const long long ix = g.at(k, c);
    const double T = A.temp[ix], S = A.salt[ix], dz = A.dz[ix], zc = A.zmid[ix], zb = A.zbot[ix];
    // clamped local copies (747-762)
    const double PO4 = TRC(P.ind[0], ix), NO3 = TRC(P.ind[1], ix), SiO3 = TRC(P.ind[2], ix), NH4 = TRC(P.ind[3], ix),
                 Fe = TRC(P.ind[4], ix), O2 = TRC(P.ind[5], ix), DOC = TRC(P.ind[9], ix), DON = TRC(P.ind[10], ix),
                 DOFe = TRC(P.ind[11], ix), DOP = TRC(P.ind[12], ix), DOPr = TRC(P.ind[13], ix),
                 DONr = TRC(P.ind[14], ix), zooC = TRC(P.ind[15], ix);
    double aChl[4], aC[4], aFe[4], aSi[4], aCa[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      aChl[a] = TRC(at[a].Chl_ind, ix);
      aC[a] = TRC(at[a].C_ind, ix);
      aFe[a] = TRC(at[a].Fe_ind, ix);
      aSi[a] = (at[a].Si_ind > 0) ? TRC(at[a].Si_ind, ix) : 0.0;
      aCa[a] = (at[a].CaCO3_ind > 0) ? TRC(at[a].CaCO3_ind, ix) : 0.0;
    }
    // O2 minimum (1954-1968) folded into the sweep
    if (k == 0 || O2 < o2min) { o2min = O2; o2min_z = zc; }

    // zero-mask: an autotroph with any empty pool is empty (826-844), then quotas (850-898)
    double thetaC[4], Qfe[4], Qsi[4], gQfe[4], gQsi[4], QCa[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bool empty = aChl[a] == 0.0 || aC[a] == 0.0 || aFe[a] == 0.0;
      if (at[a].Si_ind > 0) empty = empty || aSi[a] == 0.0;
      if (empty) { aChl[a] = 0.0; aC[a] = 0.0; aFe[a] = 0.0; aSi[a] = 0.0; aCa[a] = 0.0; }
      const double Cden = aC[a] + EPSC;
      thetaC[a] = aChl[a] / Cden;
      Qfe[a] = aFe[a] / Cden;
      Qsi[a] = (at[a].Si_ind > 0) ? dmin(aSi[a] / Cden, GQSI_MAX) : 0.0;
      gQfe[a] = at[a].gQfe_0;
      if (Fe < CKS * at[a].kFe) gQfe[a] = dmax((gQfe[a] * Fe / (CKS * at[a].kFe)), at[a].gQfe_min);
      gQsi[a] = 0.0;
      if (at[a].Si_ind > 0) {
        gQsi[a] = GQSI_0;
        if ((Fe < CKSI * at[a].kFe) && (Fe > 0.0) && (SiO3 > (CKSI * at[a].kSiO3)))
          gQsi[a] = dmin((gQsi[a] * CKSI * at[a].kFe / Fe), GQSI_MAX);
        if (Fe == 0.0) gQsi[a] = GQSI_MAX;
        if (SiO3 < (CKSI * at[a].kSiO3)) gQsi[a] = dmax((gQsi[a] * SiO3 / (CKSI * at[a].kSiO3)), GQSI_MIN);
      }
      QCa[a] = 0.0;
      if (at[a].CaCO3_ind > 0) {
        QCa[a] = aCa[a] / Cden;
        if (QCa[a] > QCACO3_MAX) QCa[a] = QCACO3_MAX;
      }
    }

    // light (907-924)
    const double PAR_in = PAR_out;
    double w1 = dmax(sum4(aChl), 0.02);
    double KPARdz = (w1 < 0.13224) ? 0.000919 * pow(w1, 0.3536) : 0.001131 * pow(w1, 0.4562);
    KPARdz = KPARdz * dz;
    const double att = exp(-KPARdz);
    PAR_out = PAR_in * att;
    const double PAR_avg = PAR_in * (1.0 - att) / KPARdz;

    // temperature response and loss thresholds (1041-1094)
    const double Tfunc = pow(Q_10, ((T + T0K) - (TREF + T0K)) / 10.0);
    double f_loss_thres;
    if (zc > THRES_Z1) f_loss_thres = (zc < THRES_Z2) ? (THRES_Z2 - zc) / (THRES_Z2 - THRES_Z1) : 0.0;
    else f_loss_thres = 1.0;
    double Pprime[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double C_loss_thres = f_loss_thres * at[a].loss_thres;
      if (at[a].temp_function == TFNC_Q10) {
        if (T < at[a].temp_thres) C_loss_thres = f_loss_thres * at[a].loss_thres2;
      } else if (at[a].temp_function == TFNC_QUASI_MMRT) {
        const double tmpTmax = north ? at[a].temp_thresN : at[a].temp_thresS;
        if (T > tmpTmax) C_loss_thres = f_loss_thres * at[a].loss_thres2;
      }
      Pprime[a] = dmax(aC[a] - C_loss_thres, 0.0);
    }

    // ---- autotroph kinetics (1107-1388) ----
    double NO3_V[4], NH4_V[4], PO4_V[4], DOP_V[4], photoC[4], photoFe[4], photoSi[4], CaProd[4], photoacc[4], Nfix[4],
        Nexc[4], loss[4], agg[4], graze[4], gr_zoo[4], gr_poc[4], gr_doc[4], gr_dic[4], ls_poc[4], ls_doc[4], ls_dic[4],
        remP_dop[4], remP_dip[4], pC_NO3[4];
    double d_totCa = 0.0, d_totNfix = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double rNO3 = NO3 / at[a].kNO3, rNH4 = NH4 / at[a].kNH4;
      const double VNO3 = rNO3 / (1.0 + rNO3 + rNH4);
      const double VNH4 = rNH4 / (1.0 + rNO3 + rNH4);
      double VNtot = VNO3 + VNH4;
      if (at[a].Nfixer) VNtot = 1.0;
      ST_KCA(N_lim, a, VNtot);
      const double VFe = Fe / (Fe + at[a].kFe);
      ST_KCA(Fe_lim, a, VFe);
      double f_nut = dmin(VNtot, VFe);
      const double rPO4 = PO4 / at[a].kPO4, rDOP = DOP / at[a].kDOP;
      const double VPO4 = rPO4 / (1.0 + rPO4 + rDOP);
      const double VDOP = rDOP / (1.0 + rPO4 + rDOP);
      const double VPtot = VPO4 + VDOP;
      ST_KCA(P_lim, a, VPtot);
      f_nut = dmin(f_nut, VPtot);
      if (at[a].kSiO3 > 0.0) {
        const double VSiO3 = SiO3 / (SiO3 + at[a].kSiO3);
        ST_KCA(SiO3_lim, a, VSiO3);
        f_nut = dmin(f_nut, VSiO3);
      } else {
        ST_KCA(SiO3_lim, a, 0.0);
      }
      double PCmax = at[a].PCref * f_nut * Tfunc;
      if (T < at[a].temp_thres) PCmax = 0.0;
      if (at[a].temp_function == TFNC_QUASI_MMRT) {
        const double tmpTopt = north ? at[a].temp_optN : at[a].temp_optS;
        const double tmpTmax = north ? at[a].temp_thresN : at[a].temp_thresS;
        PCmax = PCmax * dmin(1.0, ((tmpTmax - T) / (tmpTmax - tmpTopt)));
        if (T > tmpTmax) PCmax = 0.0;
      }
      const double light_lim = (1.0 - exp((-1.0 * at[a].alphaPI * thetaC[a] * PAR_avg) / (PCmax + EPSTINV)));
      const double PCphoto = PCmax * light_lim;
      ST_KCA(light_lim, a, light_lim);
      photoC[a] = PCphoto * aC[a];
      double VNC;
      if (VNtot > 0.0) {
        NO3_V[a] = (VNO3 / VNtot) * photoC[a] * QN;
        NH4_V[a] = (VNH4 / VNtot) * photoC[a] * QN;
        VNC = PCphoto * QN;
        pC_NO3[a] = (VNO3 / VNtot) * photoC[a];            // 1828-1834
      } else { NO3_V[a] = 0.0; NH4_V[a] = 0.0; VNC = 0.0; pC_NO3[a] = 0.0; }
      ST_KCA(photoNO3, a, NO3_V[a]);
      ST_KCA(photoNH4, a, NH4_V[a]);
      if (VPtot > 0.0) {
        PO4_V[a] = (VPO4 / VPtot) * photoC[a] * at[a].Qp;
        DOP_V[a] = (VDOP / VPtot) * photoC[a] * at[a].Qp;
      } else { PO4_V[a] = 0.0; DOP_V[a] = 0.0; }
      ST_KCA(PO4_uptake, a, PO4_V[a]);
      ST_KCA(DOP_uptake, a, DOP_V[a]);
      photoFe[a] = photoC[a] * gQfe[a];
      ST_KCA(photoFe, a, photoFe[a]);
      photoSi[a] = 0.0;
      if (at[a].Si_ind > 0) {
        photoSi[a] = photoC[a] * gQsi[a];
        z_bSi = z_bSi + photoSi[a];
      }
      ST_KCA(bSi_form, a, photoSi[a]);
      w1 = at[a].alphaPI * thetaC[a] * PAR_avg;
      if (w1 > 0.0) {
        const double pChl = at[a].thetaN_max * PCphoto / w1;
        photoacc[a] = (pChl * VNC / thetaC[a]) * aChl[a];
      } else photoacc[a] = 0.0;
      CaProd[a] = 0.0;
      if (at[a].imp_calcifier) {
        double cp = P.parm_f_prod_sp_CaCO3 * photoC[a];
        cp = cp * f_nut;
        if (T < CACO3_TEMP_THRES1) cp = cp * dmax((T - CACO3_TEMP_THRES2), 0.0) / (CACO3_TEMP_THRES1 - CACO3_TEMP_THRES2);
        if (aC[a] > CACO3_SP_THRES) cp = dmin((cp * aC[a] / CACO3_SP_THRES), (F_PHOTOSP_CACO3 * photoC[a]));
        CaProd[a] = cp;
        d_totCa = d_totCa + cp;
        w1 = dz * cp;
        z_Ca[a] = z_Ca[a] + w1;
        z_CaTot = z_CaTot + w1;
      }
      ST_KCA(CaCO3_form, a, CaProd[a]);
      loss[a] = at[a].mort * Pprime[a] * Tfunc;
      agg[a] = dmin((at[a].agg_rate_max * DPS) * Pprime[a], at[a].mort2 * Pprime[a] * Pprime[a]);
      agg[a] = dmax((at[a].agg_rate_min * DPS) * Pprime[a], agg[a]);
      // grazing on the grazee class this autotroph belongs to (1297-1324)
      w1 = 0.0;
#pragma unroll
      for (int a2 = 0; a2 < 4; ++a2)
        if (at[a2].grazee_ind == at[a].grazee_ind) w1 = w1 + Pprime[a2];
      double z_umax = at[a].z_umax_0 * Tfunc;
      if (a + 1 == P.diat_ind) {
        if (north && (T > at[a].temp_optN))
          z_umax = z_umax * dmax((at[a].temp_thresN - T) / (at[a].temp_thresN - at[a].temp_optN), 0.95);
        else if (south_or_eq && (T > at[a].temp_optS))
          z_umax = z_umax * dmax((at[a].temp_thresS - T) / (at[a].temp_thresS - at[a].temp_optS), 0.95);
      }
      graze[a] = (w1 > 0.0) ? (Pprime[a] / w1) * z_umax * zooC * (w1 / (w1 + at[a].z_grz)) : 0.0;
      Nfix[a] = 0.0; Nexc[a] = 0.0;
      if (at[a].Nfixer) {
        w1 = photoC[a] * QN;
        Nfix[a] = (w1 * R_NFIX_PHOTO) - NO3_V[a] - NH4_V[a];
        Nexc[a] = Nfix[a] + NO3_V[a] + NH4_V[a] - w1;
        d_totNfix = d_totNfix + Nfix[a];
      }
      ST_KCA(Nfix, a, Nfix[a]);
      // routing (1354-1386)
      gr_zoo[a] = at[a].graze_zoo * graze[a];
      if (at[a].imp_calcifier)
        gr_poc[a] = graze[a] * dmax((CACO3_POC_MIN * QCa[a]), dmin(SPC_POC_FAC * dmax(1.0, Pprime[a]), F_GRAZE_SP_POC_LIM));
      else gr_poc[a] = at[a].graze_poc * graze[a];
      gr_doc[a] = at[a].graze_doc * graze[a];
      gr_dic[a] = graze[a] - (gr_zoo[a] + gr_poc[a] + gr_doc[a]);
      ls_poc[a] = at[a].imp_calcifier ? QCa[a] * loss[a] : at[a].loss_poc * loss[a];
      ls_doc[a] = (1.0 - P.parm_labile_ratio) * (loss[a] - ls_poc[a]);
      ls_dic[a] = P.parm_labile_ratio * (loss[a] - ls_poc[a]);
      remP_dop[a] = 0.0; remP_dip[a] = 0.0;
      if (at[a].Qp != QP_ZOO_POM) {
        const double remaining_P = ((graze[a] + loss[a] + agg[a]) * at[a].Qp) - ((gr_zoo[a]) * QP_ZOO_POM) -
                                   ((gr_poc[a] + ls_poc[a] + agg[a]) * QP_ZOO_POM);
        remP_dop[a] = (1.0 - P.parm_labile_ratio) * remaining_P;
        remP_dip[a] = P.parm_labile_ratio * remaining_P;
      }
    }
    ST_KC(tot_CaCO3_form, d_totCa);
    ST_KC(tot_Nfix, d_totNfix);

    // ---- zooplankton, DOM, detritus production (1395-1500) ----
    w1 = 0.0;
    double w2 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      w1 = w1 + at[a].f_zoo_detr * (graze[a] + EPSC * EPSTINV);
      w2 = w2 + (graze[a] + EPSC * EPSTINV);
    }
    const double f_zoo_detr = w1 / w2;
    const double Zprime = dmax(zooC - f_loss_thres * LOSS_THRES_ZOO, 0.0);
    const double zoo_loss = (P.parm_z_mort2_0 * pow(Zprime, 1.5) + P.parm_z_mort_0 * Zprime) * Tfunc;
    const double zoo_loss_doc = (1.0 - P.parm_labile_ratio) * (1.0 - f_zoo_detr) * zoo_loss;
    const double zoo_loss_dic = P.parm_labile_ratio * (1.0 - f_zoo_detr) * zoo_loss;
    const double DOC_prod = zoo_loss_doc + sum4(ls_doc) + sum4(gr_doc);
    const double DON_prod = QN * DOC_prod;
    double DOP_prod = QP_ZOO_POM * zoo_loss_doc;
    double DOFe_prod = QFE_ZOO * zoo_loss_doc;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (at[a].Qp == QP_ZOO_POM) DOP_prod = DOP_prod + at[a].Qp * (ls_doc[a] + gr_doc[a]);
      else DOP_prod = DOP_prod + remP_dop[a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) DOFe_prod = DOFe_prod + Qfe[a] * (ls_doc[a] + gr_doc[a]);
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
    const double prodPOC = f_zoo_detr * zoo_loss + sum4(gr_poc) + sum4(agg) + sum4(ls_poc);
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].CaCO3_ind > 0) prodCa = ((1.0 - F_GRAZE_CACO3_REMIN) * graze[a] + loss[a] + agg[a]) * QCa[a];
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].Si_ind > 0) prodSi = Qsi[a] * ((1.0 - F_GRAZE_SI_REMIN) * graze[a] + agg[a] + at[a].loss_poc * loss[a]);

    // iron scavenging onto the particles LEAVING the level above (1510-1529)
    double Fe_scav_rate = P.parm_fe_scavenge_rate0;
    Fe_scav_rate = Fe_scav_rate * ((fPOC.s + fPOC.h) * 120.1 + (fCa.s + fCa.h) * CACO3_MASS + (fSi.s + fSi.h) * SIO2_MASS +
                                   (fDust.s + fDust.h) * DUST_FESCAV_SCALE);
    if (Fe > FE_SCAV_THRES1) Fe_scav_rate = Fe_scav_rate + (Fe - FE_SCAV_THRES1) * FE_MAX_SCALE2;
    const double Fe_scavenge = YPS * Fe * Fe_scav_rate;
    double prodFe = (zoo_loss * f_zoo_detr * QFE_ZOO) + Fe_scavenge;
#pragma unroll
    for (int a = 0; a < 4; ++a) prodFe = prodFe + Qfe[a] * (agg[a] + gr_poc[a] + ls_poc[a]);

    // ---- sinking particles: one step of the Armstrong et al. ballast recurrence (2242-2501) ----
    const Flux inPOC = fPOC, inCa = fCa, inSi = fSi, inFe = fFe, inDust = fDust;
    double sedPOC = 0.0, sedCa = 0.0, sedSi = 0.0, sedDust = 0.0, sedFe = 0.0, SED_DENITRIF = 0.0, OTHER_REMIN = 0.0;
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
    const double DECAY_Hard = exp(-dz / 4.0e6), DECAY_HardDust = exp(-dz / 1.2e7);
    const double TfuncS = pow(1.5, ((T + T0K) - (TREF + T0K)) / 10.0);
    const double dzr = 1.0 / dz;
    double poc_diss = P.parm_POC_diss;
    if ((O2 >= 5.0) && (O2 < 40.0)) poc_diss = P.parm_POC_diss * (1.0 + (3.3 - 1.0) * (40.0 - O2) / 35.0);
    else if (O2 < 5.0) poc_diss = P.parm_POC_diss * 3.3;
    poc_diss = scalelength * poc_diss;
    double sio2_diss = scalelength * P.parm_SiO2_diss;
    const double caco3_diss = scalelength * P.parm_CaCO3_diss;
    const double dust_diss = scalelength * DUST_DISS;
    sio2_diss = sio2_diss / TfuncS;
    const double decay_POC_E = exp(-dz / poc_diss), decay_SiO2 = exp(-dz / sio2_diss), decay_CaCO3 = exp(-dz / caco3_diss),
                 decay_dust = exp(-dz / dust_diss);
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
    reminFe = reminFe + reminDust * DUST_TO_FE + (A.fesed[ix] * dzr);
    fFe.h = inFe.h;

    if (k == kmax - 1) {   // sea floor: burial, sediment denitrification, instantaneous remin (2522-2631)
      double flux = fPOC.s + fPOC.h, flux_alt;
      if (flux > 0.0) {
        flux_alt = flux * 0.01 * SPD;
        sedPOC = flux * dmin(0.8, P.parm_POMbury * (0.013 + 0.53 * flux_alt * flux_alt / ((7.0 + flux_alt) * (7.0 + flux_alt))));
        SED_DENITRIF = dzr * flux * (0.06 + 0.19 * pow(0.99, (O2 - NO3)));
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
    if (DIAG) {   // 2637-2694
      ST_KC(POC_FLUX_IN, inPOC.s + inPOC.h); ST_KC(POC_PROD, prodPOC); ST_KC(POC_REMIN, reminPOC);
      ST_KC(CaCO3_FLUX_IN, inCa.s + inCa.h); ST_KC(CaCO3_PROD, prodCa); ST_KC(CaCO3_REMIN, reminCa);
      ST_KC(SiO2_FLUX_IN, inSi.s + inSi.h); ST_KC(SiO2_PROD, prodSi); ST_KC(SiO2_REMIN, reminSi);
      ST_KC(dust_FLUX_IN, inDust.s + inDust.h); ST_KC(dust_REMIN, reminDust);
      ST_KC(P_iron_FLUX_IN, inFe.s + inFe.h); ST_KC(P_iron_PROD, prodFe); ST_KC(P_iron_REMIN, reminFe);
      ST_KC(calcToSed, sedCa); ST_KC(bsiToSed, sedSi); ST_KC(pocToSed, sedPOC);
      ST_KC(SedDenitrif, SED_DENITRIF * dz); ST_KC(OtherRemin, OTHER_REMIN * dz);
      ST_KC(ponToSed, (sedPOC * QN)); ST_KC(popToSed, (sedPOC * QP_ZOO_POM));
      ST_KC(dustToSed, sedDust); ST_KC(pfeToSed, sedFe);
    }

    // ---- nitrification / denitrification (1545-1577) ----
    const double RESTORE = 0.0;   // lrest_* are never set by the reference (BGC_mod.F90:131-134)
    double NITRIF;
    if (PAR_out < P.parm_nitrif_par_lim) {
      NITRIF = P.parm_kappa_nitrif * NH4;
      if (PAR_in > P.parm_nitrif_par_lim) NITRIF = NITRIF * log(PAR_out / P.parm_nitrif_par_lim) / (-KPARdz);
    } else NITRIF = 0.0;
    w1 = ((P.parm_o2_min + P.parm_o2_min_delta) - O2) / P.parm_o2_min_delta;
    w1 = dmin(dmax(w1, 0.0), 1.0);
    if (NO3 == 0.0) w1 = 0.0;
    const double DENITRIF = w1 * ((DOC_remin + reminPOC - OTHER_REMIN) / DENITRIF_C_N - SED_DENITRIF);

    // ---- tendencies (1583-1790) ----
    const double s_ls_dic = sum4(ls_dic), s_gr_dic = sum4(gr_dic);
    const double t_no3 = RESTORE + NITRIF - DENITRIF - SED_DENITRIF - sum4(NO3_V);
    double t_nh4 = -sum4(NH4_V) - NITRIF + DON_remin + DONr_remin +
                   QN * (zoo_loss_dic + s_ls_dic + s_gr_dic + reminPOC * (1.0 - DONREFRACT));
    double t_fe = reminFe + (QFE_ZOO * zoo_loss_dic) + DOFe_remin - sum4(photoFe) - Fe_scavenge;
    double t_sio3 = RESTORE + reminSi;
    double t_po4 = RESTORE + DOP_remin + DOPr_remin - sum4(PO4_V) + QP_ZOO_POM * ((1.0 - DOPREFRACT) * reminPOC + zoo_loss_dic);
    double t_dic = s_ls_dic + s_gr_dic - sum4(photoC) + DOC_remin + reminPOC + zoo_loss_dic + reminCa;
    double t_C[4], t_Si[4], t_Ca[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].Nfixer) t_nh4 = t_nh4 + Nexc[a];
#pragma unroll
    for (int a = 0; a < 4; ++a) t_fe = t_fe + (Qfe[a] * (ls_dic[a] + gr_dic[a])) + gr_zoo[a] * (Qfe[a] - QFE_ZOO);
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].Si_ind > 0)
        t_sio3 = t_sio3 - photoSi[a] + Qsi[a] * (F_GRAZE_SI_REMIN * graze[a] + (1.0 - at[a].loss_poc) * loss[a]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (at[a].Qp == QP_ZOO_POM) t_po4 = t_po4 + at[a].Qp * (ls_dic[a] + gr_dic[a]);
      else t_po4 = t_po4 + remP_dip[a];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      w1 = graze[a] + loss[a] + agg[a];
      t_C[a] = photoC[a] - w1;
      TEND(at[a].C_ind, ix) = t_C[a];
      TEND(at[a].Chl_ind, ix) = photoacc[a] - thetaC[a] * w1;
      TEND(at[a].Fe_ind, ix) = photoFe[a] - Qfe[a] * w1;
      t_Si[a] = 0.0; t_Ca[a] = 0.0;
      if (at[a].Si_ind > 0) { t_Si[a] = photoSi[a] - Qsi[a] * w1; TEND(at[a].Si_ind, ix) = t_Si[a]; }
      if (at[a].CaCO3_ind > 0) { t_Ca[a] = CaProd[a] - QCa[a] * w1; TEND(at[a].CaCO3_ind, ix) = t_Ca[a]; }
    }
    const double t_zooC = sum4(gr_zoo) - zoo_loss;
    const double t_doc = DOC_prod - DOC_remin;
    const double t_don = (DON_prod * (1.0 - DONREFRACT)) - DON_remin;
    const double t_donr = (DON_prod * DONREFRACT) - DONr_remin + (reminPOC * DONREFRACT * QN);
    const double t_dop = (DOP_prod * (1.0 - DOPREFRACT)) - DOP_remin - sum4(DOP_V);
    const double t_dopr = (DOP_prod * DOPREFRACT) - DOPr_remin + (reminPOC * DOPREFRACT * QP_ZOO_POM);
    const double t_dofe = DOFe_prod - DOFe_remin;
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].CaCO3_ind > 0) t_dic = t_dic + F_GRAZE_CACO3_REMIN * graze[a] * QCa[a] - CaProd[a];
    double t_alk = -t_no3 + t_nh4 + 2.0 * reminCa;
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (at[a].CaCO3_ind > 0) t_alk = t_alk + 2.0 * (F_GRAZE_CACO3_REMIN * graze[a] * QCa[a] - CaProd[a]);
    double O2_PRODUCTION = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (photoC[a] > 0.0) {
        if (!at[a].Nfixer) {
          const double den = NO3_V[a] + NH4_V[a];
          O2_PRODUCTION = O2_PRODUCTION + photoC[a] * ((NO3_V[a] / den) / RED_D_C_O2 + (NH4_V[a] / den) / REMIN_D_C_O2);
        } else {
          const double den = NO3_V[a] + NH4_V[a] + Nfix[a];
          O2_PRODUCTION = O2_PRODUCTION + photoC[a] * ((NO3_V[a] / den) / RED_D_C_O2 + (NH4_V[a] / den) / REMIN_D_C_O2 +
                                                       (Nfix[a] / den) / RED_D_C_O2_DIAZ);
        }
      }
    }
    w1 = (O2 - P.parm_o2_min) / P.parm_o2_min_delta;
    w1 = dmin(dmax(w1, 0.0), 1.0);
    const double O2_CONSUMPTION =
        w1 * ((reminPOC + DOC_remin - (SED_DENITRIF * DENITRIF_C_N) - OTHER_REMIN + zoo_loss_dic + s_ls_dic + s_gr_dic) / REMIN_D_C_O2 +
              (2.0 * NITRIF));
    TEND(P.ind[0], ix) = t_po4;
    TEND(P.ind[1], ix) = t_no3;
    TEND(P.ind[2], ix) = t_sio3;
    TEND(P.ind[3], ix) = t_nh4;
    TEND(P.ind[4], ix) = t_fe;
    TEND(P.ind[5], ix) = O2_PRODUCTION - O2_CONSUMPTION;
    TEND(P.ind[6], ix) = t_dic;
    TEND(P.ind[7], ix) = A.alt_co2_use_eco ? t_dic : 0.0;
    TEND(P.ind[8], ix) = t_alk;
    TEND(P.ind[9], ix) = t_doc;
    TEND(P.ind[10], ix) = t_don;
    TEND(P.ind[11], ix) = t_dofe;
    TEND(P.ind[12], ix) = t_dop;
    TEND(P.ind[13], ix) = t_dopr;
    TEND(P.ind[14], ix) = t_donr;
    TEND(P.ind[15], ix) = t_zooC;

    // ---- diagnostics and column integrals (1796-1945) ----
    if (DIAG) {
      ST_KC(NO3_RESTORE, RESTORE); ST_KC(SiO3_RESTORE, RESTORE); ST_KC(PO4_RESTORE, RESTORE);
      ST_KC(NITRIF, NITRIF); ST_KC(DENITRIF, DENITRIF);
      ST_KC(O2_PRODUCTION, O2_PRODUCTION); ST_KC(O2_CONSUMPTION, O2_CONSUMPTION);
      if (D.AOU) D.AOU[ix] = o2sat(T, S, T0K) - O2;
      ST_KC(PAR_avg, PAR_avg); ST_KC(zoo_loss, zoo_loss); ST_KC(auto_graze_TOT, sum4(graze));
      const double pCtot = sum4(photoC);
      ST_KC(photoC_TOT, pCtot);
      z_pCTot = z_pCTot + pCtot * dz;
      double d_pCNO3 = 0.0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        ST_KCA(auto_graze, a, graze[a]); ST_KCA(auto_loss, a, loss[a]); ST_KCA(auto_agg, a, agg[a]);
        ST_KCA(photoC, a, photoC[a]); ST_KCA(photoC_NO3, a, pC_NO3[a]);
        z_pC[a] = z_pC[a] + dz * photoC[a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        d_pCNO3 = d_pCNO3 + pC_NO3[a];
        z_pCNO3[a] = z_pCNO3[a] + pC_NO3[a] * dz;
        z_pCNO3Tot = z_pCNO3Tot + z_pCNO3[a];   // adds the RUNNING integral (1844-1846)
      }
      ST_KC(photoC_NO3_TOT, d_pCNO3);
      ST_KC(DOC_prod, DOC_prod); ST_KC(DOC_remin, DOC_remin); ST_KC(DON_prod, DON_prod); ST_KC(DON_remin, DON_remin);
      ST_KC(DOP_prod, DOP_prod); ST_KC(DOP_remin, DOP_remin); ST_KC(DOFe_prod, DOFe_prod); ST_KC(DOFe_remin, DOFe_remin);
      ST_KC(Fe_scavenge, Fe_scavenge); ST_KC(Fe_scavenge_rate, Fe_scav_rate);
      // conservation integrals
      w2 = dmin(100.0e2 - zbot_above, dz);
      const double pt100 = (w2 > 0.0) ? w2 : 0.0;
      const bool shallow = zb <= 100.0e2;
      const double sC = ((t_C[0] + t_C[1]) + t_C[2]) + t_C[3];
      w1 = t_dic + t_doc + t_zooC + sC;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (at[a].CaCO3_ind > 0) w1 = w1 + t_Ca[a];
      J_C = J_C + w1 * dz + sedPOC + sedCa;
      J_C100 = J_C100 + w1 * pt100 + (shallow ? sedPOC + sedCa : 0.0);
      w1 = t_no3 + t_nh4 + t_don + t_donr + QN * t_zooC + QN * sC;
      w1 = w1 + DENITRIF + SED_DENITRIF;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (at[a].Nfixer) w1 = w1 - Nfix[a];
      J_N = J_N + w1 * dz + sedPOC * QN;
      J_N100 = J_N100 + w1 * pt100 + (shallow ? sedPOC * QN : 0.0);
      w1 = t_po4 + t_dop + t_dopr + QP_ZOO_POM * t_zooC;
#pragma unroll
      for (int a = 0; a < 4; ++a) w1 = w1 + at[a].Qp * t_C[a];
      J_P = J_P + w1 * dz + sedPOC * QP_ZOO_POM;
      J_P100 = J_P100 + w1 * pt100 + (shallow ? sedPOC * QP_ZOO_POM : 0.0);
      w1 = t_sio3;
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (at[a].Si_ind > 0) w1 = w1 + t_Si[a];
      J_Si = J_Si + w1 * dz + sedSi;
      J_Si100 = J_Si100 + w1 * pt100 + (shallow ? sedSi : 0.0);
#pragma unroll
      for (int a = 0; a < 4; ++a) z_Chl100 = z_Chl100 + aChl[a] * pt100;
    }
    zbot_above = zb;
  }   // k

  // ---- below the sea floor: the reference zero-fills tendencies and diagnostics (570, 625-727) ----
  for (int k = (kmax > 0 ? kmax : 0); k < g.nlev; ++k) {
    const long long ix = g.at(k, c);
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
  }

  if (DIAG) {
    ST_C(O2_ZMIN, o2min); ST_C(O2_ZMIN_DEPTH, o2min_z);
    ST_C(tot_bSi_form, z_bSi); ST_C(tot_CaCO3_form_zint, z_CaTot); ST_C(photoC_TOT_zint, z_pCTot);
    ST_C(photoC_NO3_TOT_zint, z_pCNO3Tot); ST_C(Chl_TOT_zint_100m, z_Chl100);
    ST_C(Jint_Ctot, J_C); ST_C(Jint_100m_Ctot, J_C100); ST_C(Jint_Ntot, J_N); ST_C(Jint_100m_Ntot, J_N100);
    ST_C(Jint_Ptot, J_P); ST_C(Jint_100m_Ptot, J_P100); ST_C(Jint_Sitot, J_Si); ST_C(Jint_100m_Sitot, J_Si100);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      ST_CA(CaCO3_form_zint, a, z_Ca[a]); ST_CA(photoC_zint, a, z_pC[a]); ST_CA(photoC_NO3_zint, a, z_pCNO3[a]);
    }
  }
#undef TRC
#undef TEND
#undef ST_KC
#undef ST_KCA
#undef ST_C
#undef ST_CA
}

}  // namespace

void launch_bgc_ecosys(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s) {
  if (a.g.ncol <= 0) return;
  const unsigned grid = (unsigned)((a.g.ncol + kBlock - 1) / kBlock);
  if (a.have_diag) hipLaunchKernelGGL(bgc_ecosys_kernel<true>, dim3(grid), dim3(kBlock), 0, s, a, p);
  else hipLaunchKernelGGL(bgc_ecosys_kernel<false>, dim3(grid), dim3(kBlock), 0, s, a, p);
}

}  // namespace obgc
