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

constexpr int kBlock = 64;

__global__ __launch_bounds__(kBlock) void bgc_carbonate_kernel(BgcKernelArgs A, int i_dic, int i_alk, int i_po4,
                                                              int i_sio3) {
  const GridView& g = A.g;
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  // lanes run along the contiguous axis of the caller's layout
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  carbonate_cell(A, i_dic, i_alk, i_po4, i_sio3, idx < ncell, k, c);
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
