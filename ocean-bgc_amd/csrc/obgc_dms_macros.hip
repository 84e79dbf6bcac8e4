// DMS_SourceSink (DMS_mod.F90:156-770) and MACROS_SourceSink (MACROS_mod.F90:137-411).
//
// DMS: one lane per column because PAR is attenuated level by level (538-551); only the
// DMS and DMSP planes of the 14-plane tendency array are ever non-zero, the other twelve
// are zero-filled like the reference's whole-array assignment (413).
// MACROS: no vertical coupling at all -> one lane per cell, pure streaming
// (88 algorithmic bytes per active cell).
#include "obgc_dm.h"

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
  const double KPARdz = kpar_dz_of(spChl + diatChl + diazChl + phaeoChl, A.dz[ix]);
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
  // A cell below the sea floor runs along on whatever its arrays hold and stores zeros: one store pass per wavefront, also
  // where it straddles the floor (the reference zero-fills the tendencies of the whole array first, DMS_mod.F90:413, and
  // leaves the diagnostics below the floor alone)
  const bool live = k < A.kmax[c];
  const DmsColumn col = dms_column_terms(P, A.SST[c]);
  const double DMS = TRD(0, ix), DMSP = TRD(1, ix), zooC = TRD(4, ix), spC = TRD(5, ix), spCaCO3 = TRD(6, ix), diatC = TRD(7, ix),
               diazC = TRD(8, ix), phaeoC = TRD(9, ix), spChl = TRD(10, ix);
  const double PAR_in = A.s_parin[ix], att = A.s_att[ix], KPARdz = A.s_kpardz[ix];
  const double PAR_avg = PAR_in * (1.0 - att) / KPARdz;
  double t_dms, t_dmsp;
  dms_cell_terms<DIAG>(P, D, ix, col, DMS, DMSP, zooC, spC, spCaCO3, diatC, diazC, phaeoC, spChl, PAR_avg, t_dms, t_dmsp, live);
  const int n_dms = P.ind[0] - 1, n_dmsp = P.ind[1] - 1;
#pragma unroll
  for (int n = 0; n < OBGC_DMS_NTRACER; ++n)
    A.tend[(long long)n * pl + ix] = (live && n == n_dms) ? t_dms : (live && n == n_dmsp) ? t_dmsp : 0.0;
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
  const bool live = k < A.kmax[c];   // below the floor: zeros through the same stores (MACROS_mod.F90:267), diagnostics left alone
#define TRM(j) dmax(0.0, A.tracers[(long long)(P.ind[j] - 1) * pl + ix])
  double t_prot, t_poly, t_lip;
  macros_cell_terms<DIAG>(P, D, ix, TRM(3), TRM(4), TRM(5), TRM(6), TRM(7), TRM(0), TRM(1), TRM(2), t_prot, t_poly, t_lip, live);
  const int n_prot = P.ind[0] - 1, n_poly = P.ind[1] - 1, n_lip = P.ind[2] - 1;
#pragma unroll
  for (int n = 0; n < OBGC_MACROS_NTRACER; ++n)
    A.tend[(long long)n * pl + ix] = !live ? 0.0 : (n == n_prot) ? t_prot : (n == n_poly) ? t_poly : (n == n_lip) ? t_lip : 0.0;
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
