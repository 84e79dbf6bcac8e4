// Internal declarations shared by the HIP translation units of libobgc.
// gfx950 (MI355X) only; fp64 throughout; no fast-math.
#pragma once

#include <hip/hip_runtime.h>

#include "obgc.h"

// Diagnostic planes are written once and not read again by the kernel that writes them: streamed past the caches
// (-3.5 % on the column-block kernel at 1 M columns x 80 levels with every diagnostic; the same bits)
#define OBGC_STORE(p, v) __builtin_nontemporal_store((v), (p))

// (k,column) diagnostics written by the particle part of the path (bgc_sweep_kernel / the finish phase of the column-block
// kernel); the others come from the cell terms and the carbonate block
#define OBGC_SWEEP_DIAG_KC(X) \
  X(POC_FLUX_IN) X(POC_PROD) X(POC_REMIN) X(CaCO3_FLUX_IN) X(CaCO3_PROD) X(CaCO3_REMIN) X(SiO2_FLUX_IN) X(SiO2_PROD) \
  X(SiO2_REMIN) X(dust_FLUX_IN) X(dust_REMIN) X(P_iron_FLUX_IN) X(P_iron_PROD) X(P_iron_REMIN) X(calcToSed) X(bsiToSed) \
  X(pocToSed) X(SedDenitrif) X(OtherRemin) X(ponToSed) X(popToSed) X(dustToSed) X(pfeToSed) X(DENITRIF) X(O2_CONSUMPTION) \
  X(Fe_scavenge) X(Fe_scavenge_rate)

namespace obgc {

// Strided view of the level-dependent arrays of one call.  Element (k, c) of plane n
// lives at  n*plane + k*sk + c*sc + (c / 64)*sbx.   KFAST: sk = 1, sc = ld.   COLFAST: sk = ld, sc = 1.
// TILED (blocks of 64 columns, a block's levels ld apart): sk = ld, sc = 1, sbx = ld*nlev - 64; sbx = 0 otherwise.
struct GridView {
  int nlev, ncol;
  long long sk, sc, plane, ldc, sbx;
  int kfast, tiled;
  __host__ __device__ inline long long at(int k, int c) const { return (long long)k * sk + (long long)c * sc + (long long)(c >> 6) * sbx; }
};

inline GridView make_view(const obgc_grid& g) {
  GridView v;
  v.nlev = g.nlev;
  v.ncol = g.ncol;
  v.kfast = (g.layout == OBGC_LAYOUT_KFAST);
  v.tiled = (g.layout == OBGC_LAYOUT_TILED);
  v.ldc = g.ldc > 0 ? g.ldc : g.ncol;
  if (v.tiled) {
    v.sk = g.ld;
    v.sc = 1;
    v.sbx = g.ld * (long long)g.nlev - 64;
    v.plane = g.plane > 0 ? g.plane : 64;
    return v;
  }
  long long ld = g.ld > 0 ? g.ld : (v.kfast ? g.nlev : g.ncol);
  v.sk = v.kfast ? 1 : ld;
  v.sc = v.kfast ? ld : 1;
  v.sbx = 0;
  v.plane = g.plane > 0 ? g.plane : (v.kfast ? ld * (long long)g.ncol : ld * (long long)g.nlev);
  return v;
}

// Hand-off between the cell-parallel ecosystem kernel and the column sweep: one fp64 plane per
// slot, laid out exactly like the caller's (k,column) arrays (same index, plane stride splane).
// Everything else the sweep needs is either recomputed from the inputs it reads anyway (scale lengths,
// decay factors) or parked by the cell kernel in the caller's own tendency planes (ten coupled
// tendencies whose local part is final and whose particle part the sweep adds).
enum ScratchSlot {
  S_PARIN, S_ATT, S_KPARDZ,                    // PAR entering the level, exp(-KPARdz), KPARdz (att / scan kernels -> cell kernel)
  S_PRODPOC, S_PRODCA, S_PRODSI, S_PFELOC,     // particle production terms (P_iron without scavenging)
  S_NBASE,
  S_OC0 = S_NBASE, S_D0,                       // diagnostics only: local parts of O2_CONSUMPTION and DENITRIF
  S_JC, S_JN, S_JP, S_JSI, S_CHL100,           //                   local parts of the column integrals
  S_NDIAG
};

// Planes of the particle sweep when it runs as three kernels (small batches, bgc_walk_kernel): the level's dissolution
// lengths and decay factors in front of the recurrence, what one step of the recurrence leaves behind it.
enum SplitSlot {
  W_POC_DISS, W_SIO2_DISS, W_CACO3_DISS, W_DECAY_POC_E, W_DECAY_SIO2, W_DECAY_CACO3, W_DECAY_DUST, W_DECAY_HARD, W_DECAY_HARDDUST,
  W_REMIN_POC, W_REMIN_CA, W_REMIN_SI, W_REMIN_FE, W_FE_SCAVENGE, W_SED_DENITRIF, W_OTHER_REMIN,
  W_NBASE,
  W_IN_POC = W_NBASE, W_IN_CA, W_IN_SI, W_IN_DUST, W_IN_FE, W_REMIN_DUST, W_FE_SCAV_RATE,   // diagnostics only
  W_SED_POC, W_SED_CA, W_SED_SI, W_SED_DUST, W_SED_FE,
  W_NDIAG
};

// ---------------------------------------------------------------- kernel arguments
struct BgcKernelArgs {
  GridView g;
  const double* tracers;
  const double *temp, *salt, *zmid, *dz, *zbot, *fesed;
  const double *lat, *dust, *sw;
  const int* kmax;
  double* tend;
  double *ph_prev, *ph_prev_alt;
  // carbonate results needed by the saturation-horizon scan when the caller did not ask
  // for the corresponding diagnostics (scratch owned by the handle, same shape as (k,c))
  double *co3_out, *sat_calc_out, *sat_arag_out;
  // ecosystem hand-off scratch (ScratchSlot planes) and the four production profiles the sweep
  // integrates vertically: the caller's diagnostic arrays when given, else scratch (4 planes each)
  double* scratch;
  long long splane;
  // separate kernels with diagnostics: chlorophyll x thickness above 100 m of every cell, summed by the column-integral kernel
  double* chl100;
  // small batches: the sweep as three kernels, SplitSlot planes (NULL = the one-kernel sweep)
  double* split;
  // the column-block kernel advances the conservation integrals, the O2 minimum and the 100 m chlorophyll integral itself
  // (and the saturation horizons when it solves the carbonate system): the column-integral kernel skips them
  int colint;
  double *z_photoC, *z_photoC_NO3, *z_CaCO3_form, *z_bSi_form;
  obgc_bgc_diag diag;
  int have_diag;
  int alt_co2_use_eco;
  double recip[40];   // refined reciprocals of the divisors every cell shares (RecipSlot, div_by)
};

// A cell's terms divide ~45 times by something every cell shares -- half-saturation constants, Redfield ratios, the width of
// the O2 window.  The division as this library is compiled (reciprocal, two Newton steps, quotient, one correction) spends
// five of its eight instructions on the divisor alone; for these divisors that part is done once per parameter set, on the
// device, by the same instructions (bgc_recips_compute), and travels to the kernels in their arguments.  div_by finishes the
// division with the remaining three: the same operations on the same values, hence the same bits.
enum RecipSlot {
  R_KNO3 = 0, R_KNH4 = 4, R_KPO4 = 8, R_KDOP = 12, R_CKS_KFE = 16, R_CKSI_KSIO3 = 20,   // + autotroph index
  R_RED = 24, R_REMIN, R_RED_DIAZ, R_DENITRIF, R_O2DELTA, R_TEN, R_35, R_4E6, R_12E7,
  R_SCALELEN_DZ,   // + 0..2: spacing of the length-scale table's depths
  R_N = R_SCALELEN_DZ + 3
};
static_assert(R_N <= 40, "BgcKernelArgs::recip");
__device__ __forceinline__ double div_by(double x, double y, double r) {
#ifdef OBGC_IEEE_DIV
  (void)r;
  return x / y;
#else
  const double q = x * r;
  return __builtin_fma(__builtin_fma(-y, q, x), r, q);
#endif
}
// host: the table for a parameter set, computed on the current device (synchronises `s`)
hipError_t bgc_recips_compute(const obgc_bgc_params& P, double* recip_host, hipStream_t s);

// DMS_SourceSink + MACROS_SourceSink folded into the BGC kernels (obgc_fused_step).  The tracers both modules share with
// BGC are read from the BGC tracer array; only their own tracers, tendencies and diagnostics are separate planes in the
// same (k,column) index space.
struct FusedDmArgs {
  const double *dms, *dmsp, *SST;       // dms == NULL: no DMS part
  double *dms_tend, *dmsp_tend;
  const double *prot, *poly, *lip;      // prot == NULL: no MACROS part
  double *prot_tend, *poly_tend, *lip_tend;
  obgc_dms_diag ddiag;
  obgc_macros_diag mdiag;
  int have_ddiag, have_mdiag;
  // DMS attenuates PAR with the clamped chlorophyll (DMS_mod.F90:539-548), BGC with the clamped AND zero-masked one
  // (BGC_mod.F90:826-844, 914-921).  Where the two sums are the same bits -- everywhere unless the mask removes a pool with
  // chlorophyll left in it -- DMS shares BGC's light pass; a column with a cell where they differ is flagged by the
  // attenuation kernel and gets a DMS light pass of its own from the scan kernel (three planes, flagged columns only).
  int* flag;                            // (column), zeroed before the attenuation kernel
  int all_flagged;                      // f_qsw_par_DMS differs from BGC's 0.45: nothing can be shared
  double f_qsw_par_dms;
  double *d_att, *d_kpardz, *d_parin;
};

struct SurfKernelArgs {
  GridView g;
  const double* tracers;
  const double *surfacePressure, *wind2, *atmCO2, *atmCO2_alt, *surfaceDepth, *SST, *SSS;
  double *iceFraction, *surface_pH, *surface_pH_alt;
  double *depo, *river, *gas, *seaice, *net;
  obgc_bgc_flux_diag diag;
  int have_diag;
  int lcalc_o2, lcalc_co2;
};

struct DmsKernelArgs {
  GridView g;
  const double *tracers, *dz, *SST, *sw;
  const int* kmax;
  double* tend;
  obgc_dms_diag diag;
  int have_diag;
  // engine scratch in the caller's (k,column) index space: exp(-KPARdz), KPARdz, PAR entering the level
  double *s_att, *s_kpardz, *s_parin;
  // for the reference layout: column-fastest twins of s_att / s_parin (the PAR scan is the only column kernel)
  double *w_att, *w_parin;
};

struct DmsSurfKernelArgs {
  GridView g;
  const double *tracers, *surfacePressure, *wind2, *SST;
  double *iceFraction, *net;
  obgc_dms_flux_diag diag;
  int have_diag;
};

struct MacrosKernelArgs {
  GridView g;
  const double* tracers;
  const int* kmax;
  double* tend;
  obgc_macros_diag diag;
  int have_diag;
};

// ---------------------------------------------------------------- launchers (one per .hip file)
void launch_bgc_carbonate(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s);
void launch_bgc_zsat(const BgcKernelArgs& a, hipStream_t s);
void launch_bgc_ecosys_cells(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, hipEvent_t ev_cell0, hipEvent_t ev_cell1,
                             int fuse_carbonate);
void launch_bgc_ecosys_sweep(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, int lanes);
// the same results from three launches (a.split != NULL): decay factors per cell, the recurrence per column, the rest per cell
void launch_bgc_split_sweep(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s);
void launch_bgc_zint(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s);
void launch_bgc_surface(const SurfKernelArgs& a, const obgc_bgc_params& p, hipStream_t s);
void launch_dms(const DmsKernelArgs& a, const obgc_dms_params& p, hipStream_t s);   // att -> PAR scan -> cell kernel
void launch_dms_surface(const DmsSurfKernelArgs& a, const obgc_dms_params& p, hipStream_t s);
void launch_macros(const MacrosKernelArgs& a, const obgc_macros_params& p, hipStream_t s);
void launch_synth(const obgc_synth_args& a, const int* ind30, hipStream_t s);
void launch_synth_extra(const obgc_synth_extra_args& a, hipStream_t s);
// (source, destination) pairs of obgc_bgc_diag_accumulate: first `nplane` (k,column) planes, then `nvec` (column) vectors
struct AccumTable {
  enum { kMax = 176 };
  const double* src[kMax];
  double* dst[kMax];
  int nplane, nvec;
};
void launch_accumulate(const GridView& g, const AccumTable& t, double w, int reset, hipStream_t s);
void launch_apply_tendencies(const GridView& g, int ntracer, double* x, const double* t, const double* nf, const double* dz,
                             const int* kmax, double dt, int clamp, hipStream_t s);
// LDS-tiled transposes between the reference's KFAST order and the native COLFAST order
// for `nplanes` planes: src/dst element strides as in GridView.
void launch_transpose_k2c(const double* src, double* dst, int nlev, int ncol, long long src_ld,
                          long long src_plane, long long dst_ld, long long dst_plane, int nplanes, hipStream_t s);
void launch_transpose_c2k(const double* src, double* dst, int nlev, int ncol, long long src_ld,
                          long long src_plane, long long dst_ld, long long dst_plane, int nplanes, hipStream_t s);

// table of unrelated planes moved by one transpose launch
struct PlaneTable {
  enum { kMax = 64 };
  const double* src[kMax];
  double* dst[kMax];
  int n;
  void add(const double* s, double* d) { src[n] = s; dst[n] = d; ++n; }
};
void launch_transpose_table_k2c(const PlaneTable& t, int nlev, int ncol, long long src_ld, long long dst_ld, hipStream_t s);
void launch_transpose_table_c2k(const PlaneTable& t, int nlev, int ncol, long long src_ld, long long dst_ld, const int* kmax,
                                hipStream_t s);
void launch_bgc_att(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s);
void launch_bgc_parscan(const BgcKernelArgs& a, hipStream_t s);
void launch_bgc_cell(const BgcKernelArgs& a, const obgc_bgc_params& p, hipStream_t s, int fuse_carbonate);
// the same three with DMS / MACROS folded in (obgc_fused_step)
struct FusedParams {
  obgc_dms_params dms;
  obgc_macros_params macros;
};
// the column-block kernel: attenuation, scan, cell terms and sweep of the native layout in one launch; f / fp non-NULL adds
// the DMS / MACROS terms of obgc_fused_step
void launch_bgc_column(const BgcKernelArgs& a, const obgc_bgc_params& p, const FusedDmArgs* f, const FusedParams* fp, hipStream_t s,
                       int fuse_carbonate);
void launch_fused_ecosys_cells(const BgcKernelArgs& a, const obgc_bgc_params& p, const FusedDmArgs& f, const FusedParams& fp, hipStream_t s,
                               hipEvent_t ev_cell0, hipEvent_t ev_cell1, int fuse_carbonate);

// device ordinal a handle was created on (obgc_api.hip owns the handle's definition)
int handle_device(const obgc_handle* h);

// ---------------------------------------------------------------- small device helpers
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }

}  // namespace obgc
