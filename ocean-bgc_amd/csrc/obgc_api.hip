// Host side of libobgc: handle, parameter tables, argument marshalling, stream layout.
// There is no CPU compute path in this library: every entry point either launches the
// HIP kernels or returns a negative status.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "obgc_handle.h"

using namespace obgc;



namespace {

constexpr double kDps = 1.0 / 86400.0;


// ---- scratch arena ---------------------------------------------------------------------
int arena_reserve(obgc_handle* h, size_t bytes) {
  if (bytes <= h->arena_bytes) return OBGC_OK;
  HIP_TRY(h, hipStreamSynchronize(h->s_main));
  HIP_TRY(h, hipStreamSynchronize(h->s_aux));
  HIP_TRY(h, hipStreamSynchronize(h->s_aux2));
  if (h->arena) (void)hipFree(h->arena);
  h->arena = nullptr;
  h->arena_bytes = 0;
  size_t want = bytes + bytes / 8;
  if (hipMalloc((void**)&h->arena, want) != hipSuccess) {
    if (hipMalloc((void**)&h->arena, bytes) != hipSuccess)
      return fail(h, OBGC_ERR_ALLOC, "device scratch allocation of " + std::to_string(bytes) + " bytes failed");
    want = bytes;
  }
  h->arena_bytes = want;
  return OBGC_OK;
}
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// A staged copy of one caller array (host memspace): device twin + copy directions.
struct Stage {
  void* dev;
  void* host;
  size_t bytes;
  bool in, out;
  int nplanes;   // > 0: a (k,column[,n]) array of that many planes, copied per column chunk by the pipelined host path
};

struct Stager {
  obgc_handle* h;
  bool host;            // caller arrays are host memory
  std::vector<Stage> items;
  size_t need = 0;
  explicit Stager(obgc_handle* hh, bool is_host) : h(hh), host(is_host) {}
  // pass 1: register; returns an index
  int add(const void* p, size_t bytes, bool in, bool out) {
    if (!p) return -1;
    items.push_back(Stage{nullptr, const_cast<void*>(p), bytes, in, out, 0});
    need += align_up(bytes);
    return (int)items.size() - 1;
  }
  int add_kc(const void* p, const GridView& v, int nplanes, bool in, bool out);
  template <class T>
  T* dev(int i) const { return i < 0 ? nullptr : (T*)items[(size_t)i].dev; }
};

// ---- parameter tables ----------------------------------------------------------------------
void set_autotroph(obgc_autotroph& a, int nfix, int impc, int grazee, int tfn, const double (&v)[30]) {
  a.Nfixer = nfix; a.imp_calcifier = impc; a.exp_calcifier = 0; a.grazee_ind = grazee; a.temp_function = tfn;
  a.Chl_ind = a.C_ind = a.Fe_ind = a.Si_ind = a.CaCO3_ind = 0;
  double* d = &a.kFe;   // the 30 reals are contiguous in OBGC_AUTOTROPH_REALS order
  for (int i = 0; i < 30; ++i) d[i] = v[i];
}

const char* kBgcShort[30] = {"PO4", "NO3", "SiO3", "NH4", "Fe", "O2", "DIC", "DIC_ALT_CO2", "ALK", "DOC", "DON", "DOFe",
                             "DOP", "DOPr", "DONr", "zooC", "spC", "spChl", "spFe", "spCaCO3", "diatC", "diatChl",
                             "diatFe", "diatSi", "phaeoC", "phaeoChl", "phaeoFe", "diazC", "diazChl", "diazFe"};
const char* kBgcLong[30] = {"Dissolved Inorganic Phosphate", "Dissolved Inorganic Nitrate", "Dissolved Inorganic Silicate",
                            "Dissolved Ammonia", "Dissolved Inorganic Iron", "Dissolved Oxygen", "Dissolved Inorganic Carbon",
                            "Dissolved Inorganic Carbon, Alternative CO2", "Alkalinity", "Dissolved Organic Carbon",
                            "Dissolved Organic Nitrogen", "Dissolved Organic Iron", "Dissolved Organic Phosphorus",
                            "Refractory DOP", "Refractory DON", "Zooplankton Carbon", "Small Phyto Carbon",
                            "Small Phyto Chlorophyll", "Small Phyto Iron", "Small Phyto CaCO3", "Diatom Carbon",
                            "Diatom Chlorophyll", "Diatom Iron", "Diatom Silicon", "Phaeocystis Carbon",
                            "Phaeocystis Chlorophyll", "Phaeocystis Iron", "Diazotroph Carbon", "Diazotroph Chlorophyll",
                            "Diazotroph Iron"};
const char* kDmsShort[14] = {"DMS", "DMSP", "NO3", "DOC", "zooC", "spC", "spCaCO3", "diatC", "diazC", "phaeoC", "spChl",
                             "diatChl", "diazChl", "phaeoChl"};
const char* kDmsLong[14] = {"DiMethyl Sulfide", "Dimethylsulfoniopropionate", "Dissolved Inorganic Nitrate",
                            "Dissolved Organic Carbon", "Zooplankton Carbon", " Small Phytoplankton Carbon",
                            " Small Phytoplankton Calcium Carbonate", " Diatom Carbon", " Diazotroph Carbon",
                            "Phaeocystis Carbon", " Small Phytoplankton Chlorophyll", " Diatom Chlorophyll",
                            " Diazotroph Chlorophyll", "Phaeocystis Chlorophyll"};
const char* kMacShort[8] = {"PROT", "POLY", "LIP", "zooC", "spC", "diatC", "diazC", "phaeoC"};
const char* kMacLong[8] = {"Proteins", "Polysaccharides", "Lipids", "Zooplankton Carbon", " Small Phytoplankton Carbon",
                           " Diatom Carbon", " Diazotroph Carbon", "Phaeocystis Carbon"};

bool grid_ok(const obgc_grid& g) {
  if (g.nlev < 1 || g.ncol < 0) return false;
  if (g.layout != OBGC_LAYOUT_KFAST && g.layout != OBGC_LAYOUT_COLFAST && g.layout != OBGC_LAYOUT_TILED) return false;
  if (g.memspace != OBGC_MEM_HOST && g.memspace != OBGC_MEM_DEVICE) return false;
  if (g.layout == OBGC_LAYOUT_TILED) {   // column blocks: device memory, an explicit level stride, planes inside it
    if (g.memspace != OBGC_MEM_DEVICE || g.ld < 64) return false;
    if (g.plane != 0 && (g.plane < 64 || g.plane > g.ld)) return false;
    return g.ldc == 0 || g.ldc >= g.ncol;
  }
  const long long minld = (g.layout == OBGC_LAYOUT_KFAST) ? g.nlev : g.ncol;
  if (g.ld != 0 && g.ld < minld) return false;
  if (g.ldc != 0 && g.ldc < g.ncol) return false;
  return true;
}

// Which kernel form a native-layout call takes.  A workgroup of the column-block kernel marches its 64 columns level by
// level: ~17 us per chunk of four levels whatever the batch, 0.25-0.3 ms for 60 levels, with 32 of 256 CUs busy at 2 048
// columns.  The separate kernels spread the cells of a small batch over the whole device and pay for it with hand-off
// planes, which cost nothing at that size.  Measured at 60 levels with every diagnostic (without): 2 048 columns 0.25
// against 0.44 ms (0.18 / 0.29), 8 192 columns 0.38 against 0.46 (0.24 / 0.30), 16 384 columns 0.59 against 0.55
// (0.35 / 0.33): the column-block kernel from 700 000 cells.  The column-block layout always takes it.
inline bool use_column_kernel(const obgc_handle* h, const GridView& v);
// Whether the separate kernels run their sweep as three launches (bgc_walk_kernel): native layout, batches small enough that
// the sweep's few wavefronts are what the step waits for and the planes in between stay in the caches.
inline bool use_split_sweep(const obgc_handle* h, const GridView& v, bool column) {
  if (column || v.kfast || v.tiled || h->split_sweep == 0) return false;
  return h->split_sweep >= 2 || (long long)v.nlev * v.ncol <= h->split_max_cells;
}
inline bool use_column_kernel(const obgc_handle* h, const GridView& v) {
  if (v.tiled) return true;
  if (h->column == 0 || v.kfast) return false;
  return h->column >= 2 || (long long)v.nlev * v.ncol >= h->column_min_cells;
}

// The column-block layout is served by the column-block kernel (the carbonate solve inside it or in the carbonate kernel
// beside it, as for the native layout) and the column integrals of the production profiles, which read the caller's own
// photoC / photoC_NO3 / CaCO3_form / bSi_form.
int tiled_ok(obgc_handle* h, const obgc_bgc_diag* d, const char* who) {
  if (!h->column)
    return fail(h, OBGC_ERR_UNSUPPORTED, std::string(who) + ": the TILED layout needs the column-block kernel (OBGC_COLUMN_KERNEL=0 is set)");
  if (d && !(d->photoC && d->photoC_NO3 && d->CaCO3_form && d->bSi_form))
    return fail(h, OBGC_ERR_UNSUPPORTED, std::string(who) + ": TILED layout with diagnostics needs photoC, photoC_NO3, CaCO3_form and bSi_form "
                "among them (the column integrals read them; the engine keeps no scratch in the caller's index space)");
  return OBGC_OK;
}

// entry points whose scratch lives in the caller's index space do not take the column-block layout
int no_tiled(obgc_handle* h, const obgc_grid& g, const char* who) {
  if (g.layout != OBGC_LAYOUT_TILED) return OBGC_OK;
  return fail(h, OBGC_ERR_UNSUPPORTED, std::string(who) + ": the TILED layout is served by obgc_bgc_source_sink, obgc_fused_step, the "
              "*_surface_fluxes, obgc_apply_tendencies and the accumulate family only");
}

// bytes spanned by `nplanes` planes of a (k,column) array
size_t span_kc(const GridView& v, int nplanes) {
  const long long last = v.at(v.nlev - 1, v.ncol > 0 ? v.ncol - 1 : 0) + 1;
  return (size_t)((long long)(nplanes - 1) * v.plane + last) * sizeof(double);
}
size_t span_c(const GridView& v, int nplanes) {
  return (size_t)((long long)(nplanes - 1) * v.ldc + v.ncol) * sizeof(double);
}

int Stager::add_kc(const void* p, const GridView& v, int nplanes, bool in, bool out) {
  const int i = add(p, span_kc(v, nplanes), in, out);
  if (i >= 0) items[(size_t)i].nplanes = nplanes;
  return i;
}

// `chunked`: the (k,column) arrays travel per column chunk (host_pipeline below); only the small arrays move here
int stage_in(obgc_handle* h, Stager& st, bool chunked = false) {
  if (!st.host) {
    for (auto& it : st.items) it.dev = it.host;
    return OBGC_OK;
  }
  int rc = arena_reserve(h, h->arena_used + st.need);
  if (rc != OBGC_OK) return rc;
  size_t off = h->arena_used;
  for (auto& it : st.items) {
    it.dev = h->arena + off;
    off += align_up(it.bytes);
    if (it.in && !(chunked && it.nplanes > 0))
      HIP_TRY(h, hipMemcpyAsync(it.dev, it.host, it.bytes, hipMemcpyHostToDevice, h->s_main));
  }
  h->arena_used = off;
  return OBGC_OK;
}

int stage_out(obgc_handle* h, Stager& st, bool chunked = false) {
  if (!st.host) return OBGC_OK;
  for (auto& it : st.items)
    if (it.out && !(chunked && it.nplanes > 0))
      HIP_TRY(h, hipMemcpyAsync(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost, h->s_main));
  HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}

// Byte range of columns [c0, c1) of plane n of a staged reference-layout array (contiguous: the column stride is the
// leading dimension).  The last range ends with the array, which may stop short of a full leading dimension.
inline void chunk_range(const Stage& it, const GridView& v, int n, int c0, int c1, size_t& begin, size_t& len) {
  begin = (size_t)((long long)n * v.plane + (long long)c0 * v.sc) * sizeof(double);
  size_t end = (size_t)((long long)n * v.plane + (long long)c1 * v.sc) * sizeof(double);
  if (end > it.bytes) end = it.bytes;
  len = end > begin ? end - begin : 0;
}

int finish(obgc_handle* h, bool force_sync) {
  if (force_sync || !(h->flags & OBGC_ASYNC)) HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}

}  // namespace

int obgc::handle_device(const obgc_handle* h) { return h->device; }

// ============================================================================ parameters ==

extern "C" void obgc_bgc_params_derive(obgc_bgc_params* p) {   // BGC_mod.F90:271-321
  enum { spC = 16, spChl, spFe, spCaCO3, diatC, diatChl, diatFe, diatSi, phaeoC, phaeoChl, phaeoFe, diazC, diazChl, diazFe };
  for (int a = 0; a < OBGC_AUTOTROPH_CNT; ++a) {
    obgc_autotroph& at = p->autotrophs[a];
    const int ai = a + 1;
    if (ai == p->sp_ind) { at.Chl_ind = p->ind[spChl]; at.C_ind = p->ind[spC]; at.Fe_ind = p->ind[spFe]; }
    else if (ai == p->diat_ind) { at.Chl_ind = p->ind[diatChl]; at.C_ind = p->ind[diatC]; at.Fe_ind = p->ind[diatFe]; }
    else if (ai == p->diaz_ind) { at.Chl_ind = p->ind[diazChl]; at.C_ind = p->ind[diazC]; at.Fe_ind = p->ind[diazFe]; }
    else if (ai == p->phaeo_ind) { at.Chl_ind = p->ind[phaeoChl]; at.C_ind = p->ind[phaeoC]; at.Fe_ind = p->ind[phaeoFe]; }
    at.Si_ind = (at.kSiO3 > 0.0) ? p->ind[diatSi] : 0;
    at.CaCO3_ind = (at.imp_calcifier || at.exp_calcifier) ? p->ind[spCaCO3] : 0;
  }
}

extern "C" void obgc_bgc_params_init(obgc_bgc_params* p, const int* ind, double T0K) {
  std::memset(p, 0, sizeof(*p));
  for (int i = 0; i < OBGC_BGC_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->sp_ind = 1; p->diat_ind = 2; p->diaz_ind = 3; p->phaeo_ind = 4;
  p->T0_Kelvin_BGC = T0K;
  // namelist defaults, BGC_parms.F90:524-541
  p->parm_Fe_bioavail = 1.0; p->parm_o2_min = 4.0; p->parm_o2_min_delta = 2.0; p->parm_kappa_nitrif = 0.06 * kDps;
  p->parm_nitrif_par_lim = 1.0; p->parm_z_mort_0 = 0.1 * kDps; p->parm_z_mort2_0 = 0.4 * kDps;
  p->parm_labile_ratio = 0.85; p->parm_POMbury = 1.4; p->parm_BSIbury = 0.65; p->parm_fe_scavenge_rate0 = 3.0;
  p->parm_f_prod_sp_CaCO3 = 0.055; p->parm_POC_diss = 88.0e2; p->parm_SiO2_diss = 250.0e2; p->parm_CaCO3_diss = 150.0e2;
  const double sz[4] = {130.0e2, 290.0e2, 670.0e2, 1700.0e2}, sv[4] = {1.0, 3.0, 5.0, 9.0};
  for (int i = 0; i < 4; ++i) { p->parm_scalelen_z[i] = sz[i]; p->parm_scalelen_vals[i] = sv[i]; }
  // autotroph table, BGC_parms.F90:543-697.  Columns: kFe kPO4 kDOP kNO3 kNH4 kSiO3 Qp gQfe_0 gQfe_min alphaPI PCref
  // thetaN_max loss_thres loss_thres2 temp_thres temp_thresS temp_thresN temp_optN temp_optS mort mort2 agg_rate_max
  // agg_rate_min z_umax_0 z_grz graze_zoo graze_poc graze_doc loss_poc f_zoo_detr
  const double sp[30] = {0.04e-3, 0.01, 0.26, 0.1, 0.01, 0.0, 0.00855, 20.0e-6, 3.0e-6, 0.6 * kDps, 5.5 * kDps, 2.5, 0.04,
                         0.0, -20.0, -20.0, -20.0, 50.0, 50.0, 0.12 * kDps, 0.001 * kDps, 0.9, 0.01, 3.3 * kDps, 1.05, 0.3,
                         0.0, 0.15, 0.0, 0.15};
  const double diat[30] = {0.06e-3, 0.05, 0.9, 0.5, 0.05, 0.8, 0.00855, 20.0e-6, 3.0e-6, 0.465 * kDps, 5.5 * kDps, 4.0, 0.04,
                           0.0, -20.0, 10.0, 35.0, 16.3, 5.0, 0.12 * kDps, 0.001 * kDps, 0.9, 0.02, 3.23 * kDps, 1.0, 0.3,
                           0.42, 0.15, 0.0, 0.2};
  const double diaz[30] = {0.04e-3, 0.02, 0.09, 1.0, 0.15, 0.0, 0.002735, 60.0e-6, 12.0e-6, 0.4 * kDps, 0.7 * kDps, 2.5, 0.022,
                           0.001, 14.0, -20.0, -20.0, 50.0, 50.0, 0.15 * kDps, 0.0, 0.0, 0.0, 0.6 * kDps, 1.2, 0.3, 0.05,
                           0.15, 0.0, 0.15};
  const double phaeo[30] = {0.075e-3, 0.05, 0.9, 0.7, 0.05, 0.0, 0.00855, 20.0e-6, 3.0e-6, 0.77 * kDps, 5.5 * kDps, 2.5, 0.04,
                            0.0, -20.0, 10.0, 35.0, 16.3, 5.0, 0.12 * kDps, 0.001 * kDps, 0.9, 0.02, 3.23 * kDps, 1.0, 0.3,
                            0.42, 0.15, 0.0, 0.2};
  set_autotroph(p->autotrophs[0], 0, 1, 1, 1, sp);
  set_autotroph(p->autotrophs[1], 0, 0, 2, 1, diat);
  set_autotroph(p->autotrophs[2], 1, 0, 3, 1, diaz);
  set_autotroph(p->autotrophs[3], 0, 0, 2, 2, phaeo);   // grazed with the diatoms; quasi-MMRT temperature curve
  obgc_bgc_params_derive(p);
}

extern "C" void obgc_dms_params_init(obgc_dms_params* p, const int* ind) {
  std::memset(p, 0, sizeof(*p));
  for (int i = 0; i < OBGC_DMS_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->k_S_p_base = 0.1 * kDps; p->zooC_avg = 0.3; p->mort = 0.0; p->k_conv = 1.0 * kDps; p->k_S_z = 0.1 * kDps;
  p->B_preexp = 0.1; p->B_exp = 0.5; p->k_S_B = 30.0 * kDps; p->k_bkgnd = 0.01 * kDps; p->j_dms_perI = 0.005 * kDps;
  p->inject_scale = 1.0; p->T_cryo_hi = 1.0; p->T_cryo_lo = -1.0; p->T_lo = 15.0; p->T_hi = 20.0;
  p->Min_cyano_frac = 0.0; p->Max_cyano_frac = 0.5; p->Min_yld = 0.2; p->Max_yld = 0.7; p->G_phaeo_S = 0.4;
  p->Sp_ref = 0.1; p->Stress_mult = 10.0; p->R = 0.137; p->Rs2n_diat = 0.01; p->Rs2n_phaeo = 0.3; p->Rs2n_cocco = 0.1;
  p->Rs2n_cyano = 0.0; p->Rs2n_eukar = 0.1; p->Rs2n_diaz = 0.0; p->f_qsw_par_DMS = 0.45;
}

extern "C" void obgc_macros_params_init(obgc_macros_params* p, const int* ind) {
  std::memset(p, 0, sizeof(*p));
  for (int i = 0; i < OBGC_MACROS_NTRACER; ++i) p->ind[i] = ind ? ind[i] : i + 1;
  p->f_prot = 0.6; p->f_poly = 0.2; p->f_lip = 0.2; p->k_C_p_base = kDps * 0.1; p->zooC_avg = 0.3; p->mort = 0.0;
  p->k_prot_bac = kDps * 0.1; p->k_poly_bac = kDps * 0.01; p->k_lip_bac = kDps * 1.0; p->inject_scale = 1.0;
}

extern "C" const char* obgc_tracer_name(int module, int slot, int which) {
  const int n = module == 0 ? 30 : module == 1 ? 14 : module == 2 ? 8 : 0;
  if (slot < 1 || slot > n || which < 0 || which > 2) return nullptr;
  const int i = slot - 1;
  if (which == 2) {
    if (module == 0) {
      if (i == 8) return "meq/m^3";                                  // ALK, BGC_mod.F90:324
      if (i == 17 || i == 21 || i == 25 || i == 28) return "mg/m^3";  // the four Chl tracers, 325-328
    }
    return "mmol/m^3";
  }
  if (module == 0) return which == 0 ? kBgcShort[i] : kBgcLong[i];
  if (module == 1) return which == 0 ? kDmsShort[i] : kDmsLong[i];
  return which == 0 ? kMacShort[i] : kMacLong[i];
}

// ================================================================================ handle ==

extern "C" int obgc_abi_version(void) { return OBGC_ABI_VERSION; }
extern "C" int obgc_sizeof_bgc_params(void) { return (int)sizeof(obgc_bgc_params); }
extern "C" int obgc_sizeof_dms_params(void) { return (int)sizeof(obgc_dms_params); }
extern "C" int obgc_sizeof_macros_params(void) { return (int)sizeof(obgc_macros_params); }
extern "C" int obgc_sizeof_bgc_diag(void) { return (int)sizeof(obgc_bgc_diag); }

extern "C" int obgc_create(obgc_handle** out, int device_id, unsigned flags) {
  if (!out) return OBGC_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return OBGC_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return OBGC_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "libobgc: device %d is %s; this library carries gfx950 (MI355X) code objects only\n", device_id,
                 prop.gcnArchName);
    return OBGC_ERR_NO_DEVICE;
  }
  obgc_handle* h = new obgc_handle();
  h->device = device_id;
  h->flags = flags;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->s_main, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->s_aux2, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return OBGC_ERR_HIP;
  }
  if (noalloc_init_device(device_id) != hipSuccess) {   // address of the fixed scratch array on THIS GPU (obgc_noalloc.hip)
    delete h;
    return OBGC_ERR_HIP;
  }
  hipEvent_t* evs[] = {&h->ev_begin, &h->ev_end, &h->ev_c0, &h->ev_c1, &h->ev_e0, &h->ev_e1, &h->ev_t0, &h->ev_t1, &h->ev_aux_done, &h->ev_aux2_done, &h->ev_k0, &h->ev_k1};
  for (auto e : evs)
    if (hipEventCreate(e) != hipSuccess) { delete h; return OBGC_ERR_HIP; }
  if (const char* e = std::getenv("OBGC_SCHED")) h->sched = std::atoi(e);
  if (const char* e = std::getenv("OBGC_COLUMN_KERNEL")) h->column = std::atoi(e);
  if (const char* e = std::getenv("OBGC_COLUMN_MIN_CELLS")) h->column_min_cells = std::atoll(e);
  if (const char* e = std::getenv("OBGC_KFAST_HYBRID")) h->kfast_hybrid = std::atoi(e);
  if (const char* e = std::getenv("OBGC_KFAST_DIRECT")) h->kfast_direct = std::atoi(e);
  if (const char* e = std::getenv("OBGC_SWEEP_LANES")) h->sweep_lanes = std::atoi(e);
  if (const char* e = std::getenv("OBGC_SPLIT_SWEEP")) h->split_sweep = std::atoi(e);
  if (const char* e = std::getenv("OBGC_SPLIT_MAX_CELLS")) h->split_max_cells = std::atoll(e);
  if (const char* e = std::getenv("OBGC_HOST_CHUNKS")) h->host_chunks = std::atoi(e);
  if (const char* e = std::getenv("OBGC_HOST_THREADS")) h->host_threads = std::atoi(e);
  if (const char* e = std::getenv("OBGC_HOST_PIPELINE_MIN_MB")) h->host_pipeline_min_bytes = (size_t)std::atoll(e) << 20;
  if (const char* e = std::getenv("OBGC_KFAST_TILE")) { const int t = std::atoi(e); if (t >= 64) h->kfast_tile = t; }
  obgc_bgc_params_init(&h->bgc, nullptr, 273.15);
  obgc_dms_params_init(&h->dms, nullptr);
  obgc_macros_params_init(&h->macros, nullptr);
  *out = h;
  return OBGC_OK;
}

extern "C" int obgc_destroy(obgc_handle* h) {
  if (!h) return OBGC_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->s_main);
  (void)hipStreamSynchronize(h->s_aux);
  if (h->s_aux2) (void)hipStreamSynchronize(h->s_aux2);
  if (h->arena) (void)hipFree(h->arena);
  hipEvent_t evs[] = {h->ev_begin, h->ev_end, h->ev_c0, h->ev_c1, h->ev_e0, h->ev_e1, h->ev_t0, h->ev_t1, h->ev_aux_done, h->ev_aux2_done, h->ev_k0, h->ev_k1};
  for (auto e : evs)
    if (e) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(h->s_main);
  (void)hipStreamDestroy(h->s_aux);
  if (h->s_aux2) (void)hipStreamDestroy(h->s_aux2);
  if (h->s_up) (void)hipStreamDestroy(h->s_up);
  if (h->s_down) (void)hipStreamDestroy(h->s_down);
  delete h;
  return OBGC_OK;
}

extern "C" int obgc_sync(obgc_handle* h) {
  if (!h) return OBGC_ERR_INVALID;
  HIP_TRY(h, hipStreamSynchronize(h->s_aux));
  HIP_TRY(h, hipStreamSynchronize(h->s_aux2));
  HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}

extern "C" const char* obgc_last_error(const obgc_handle* h) { return h ? h->err.c_str() : "null handle"; }
extern "C" void* obgc_stream(obgc_handle* h) { return h ? (void*)h->s_main : nullptr; }

extern "C" int obgc_set_bgc_params(obgc_handle* h, const obgc_bgc_params* p) {
  if (!h || !p) return OBGC_ERR_INVALID;
  for (int i = 0; i < OBGC_BGC_NTRACER; ++i)
    if (p->ind[i] < 1 || p->ind[i] > OBGC_BGC_NTRACER) return fail(h, OBGC_ERR_INVALID, "BGC tracer slot out of 1..30");
  if (p->lrest_po4 || p->lrest_no3 || p->lrest_sio3)
    return fail(h, OBGC_ERR_UNSUPPORTED, "nutrient restoring (lrest_*) is dead code in the reference and not provided");
  // (the Fortran modules push their parameters before every call: the reciprocal table survives an unchanged set)
  if (std::memcmp(&h->bgc, p, sizeof *p) != 0) h->bgc_recip_valid = false;
  h->bgc = *p;
  return OBGC_OK;
}
extern "C" int obgc_set_dms_params(obgc_handle* h, const obgc_dms_params* p) {
  if (!h || !p) return OBGC_ERR_INVALID;
  for (int i = 0; i < OBGC_DMS_NTRACER; ++i)
    if (p->ind[i] < 1 || p->ind[i] > OBGC_DMS_NTRACER) return fail(h, OBGC_ERR_INVALID, "DMS tracer slot out of 1..14");
  h->dms = *p;
  return OBGC_OK;
}
extern "C" int obgc_set_macros_params(obgc_handle* h, const obgc_macros_params* p) {
  if (!h || !p) return OBGC_ERR_INVALID;
  for (int i = 0; i < OBGC_MACROS_NTRACER; ++i)
    if (p->ind[i] < 1 || p->ind[i] > OBGC_MACROS_NTRACER) return fail(h, OBGC_ERR_INVALID, "MACROS tracer slot out of 1..8");
  h->macros = *p;
  return OBGC_OK;
}

extern "C" int obgc_last_schedule(obgc_handle* h) { return h ? h->last_sched : -1; }

extern "C" double obgc_last_kernel_ms(obgc_handle* h, int which) {
  if (!h || !h->timed_valid) return -1.0;
  if (hipEventSynchronize(h->ev_end) != hipSuccess) return -1.0;
  float ms = -1.0f;
  hipError_t e = hipSuccess;
  if (which == 0) e = hipEventElapsedTime(&ms, h->ev_begin, h->ev_end);
  else if (which == 1) e = hipEventElapsedTime(&ms, h->ev_c0, h->ev_c1);
  else if (which == 2) e = hipEventElapsedTime(&ms, h->ev_e0, h->ev_e1);
  else if (which == 3) { if (!h->tr_valid) return 0.0; e = hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1); }
  else if (which == 4) { if (!h->cell_valid) return 0.0; e = hipEventElapsedTime(&ms, h->ev_k0, h->ev_k1); }
  else return -1.0;
  return e == hipSuccess ? (double)ms : -1.0;
}


namespace {


// bytes of engine scratch obgc_bgc_source_sink needs for the view the kernels run on
// `column`: the column-block kernel keeps the light and the hand-off on the chip; only the chlorophyll profile of the 100 m
// integral goes through memory
// whether the column-block kernel solves the carbonate system in its own wavefronts (schedule 3) for this view
bool column_fuses(const obgc_handle* h, const GridView& kv, const obgc_bgc_diag* diag, bool want_zsat) {
  const bool have_diag = diag != nullptr;
  // column blocks: the engine keeps nothing of its own in the caller's index space, so the saturation-horizon kernel can only
  // run beside the column kernel on the caller's own CO3 / saturation arrays; without them the search runs inside the kernel
  if (kv.tiled && want_zsat && !(diag->CO3 && diag->co3_sat_calc && diag->co3_sat_arag)) return true;
  const bool large = (long long)kv.nlev * kv.ncol > 10000000LL;
  return h->sched == 2 || (h->sched < 0 && large && have_diag);
}

size_t bgc_scratch_bytes(const obgc_handle* h, const GridView& kv, const obgc_bgc_diag* diag, bool want_zsat, bool workspace_has_all,
                         bool column = false, bool split = false) {
  const size_t kc1 = span_kc(kv, 1);
  size_t bytes = 0;
  // (the column-block kernel with the carbonate solve inside searches the saturation horizons itself)
  if (want_zsat && !(column && column_fuses(h, kv, diag, want_zsat))) {
    if (!diag->CO3) bytes += align_up(kc1);
    if (!diag->co3_sat_calc) bytes += align_up(kc1);
    if (!diag->co3_sat_arag) bytes += align_up(kc1);
  }
  const long long span1 = (long long)(kc1 / sizeof(double));
  const long long splane = (span1 + 31) & ~31LL;
  const int nslots = column ? 0 : (diag ? (int)S_NDIAG : (int)S_NBASE);
  bytes += align_up((size_t)nslots * splane * sizeof(double));
  if (split) bytes += align_up((size_t)(diag ? (int)W_NDIAG : (int)W_NBASE) * splane * sizeof(double));
  if (diag) {
    const size_t zbytes = align_up((size_t)(3 * kv.plane + span1) * sizeof(double));
    const int nfb = (diag->photoC ? 0 : 1) + (diag->photoC_NO3 ? 0 : 1) + (diag->CaCO3_form ? 0 : 1) + (diag->bSi_form ? 0 : 1);
    bytes += nfb * zbytes;
  }
  (void)workspace_has_all;
  return bytes;
}

// carve the scratch of one launch set out of the (already reserved) arena and hook it into K
void bgc_carve_scratch(obgc_handle* h, BgcKernelArgs& K, bool want_zsat, bool column = false, bool split = false) {
  const GridView& kv = K.g;
  const size_t kc1 = span_kc(kv, 1);
  auto carve = [&](size_t bytes) -> double* {
    double* p = (double*)(h->arena + h->arena_used);
    h->arena_used += align_up(bytes);
    return p;
  };
  if (want_zsat && !(column && column_fuses(h, kv, K.have_diag ? &K.diag : nullptr, want_zsat))) {
    K.co3_out = K.diag.CO3 ? K.diag.CO3 : carve(kc1);
    K.sat_calc_out = K.diag.co3_sat_calc ? K.diag.co3_sat_calc : carve(kc1);
    K.sat_arag_out = K.diag.co3_sat_arag ? K.diag.co3_sat_arag : carve(kc1);
  }
  const long long span1 = (long long)(kc1 / sizeof(double));
  const long long splane = (span1 + 31) & ~31LL;
  if (column) {
    K.scratch = nullptr;
    K.chl100 = nullptr;
    K.colint = K.have_diag;
  } else {
    const int nslots = K.have_diag ? (int)S_NDIAG : (int)S_NBASE;
    K.scratch = carve((size_t)nslots * splane * sizeof(double));
    K.chl100 = K.have_diag ? K.scratch + (long long)S_CHL100 * splane : nullptr;
  }
  K.split = split ? carve((size_t)(K.have_diag ? (int)W_NDIAG : (int)W_NBASE) * splane * sizeof(double)) : nullptr;
  K.splane = splane;
  if (K.have_diag) {
    const size_t zbytes = (size_t)(3 * kv.plane + span1) * sizeof(double);
    K.z_photoC = K.diag.photoC ? K.diag.photoC : carve(zbytes);
    K.z_photoC_NO3 = K.diag.photoC_NO3 ? K.diag.photoC_NO3 : carve(zbytes);
    K.z_CaCO3_form = K.diag.CaCO3_form ? K.diag.CaCO3_form : carve(zbytes);
    K.z_bSi_form = K.diag.bSi_form ? K.diag.bSi_form : carve(zbytes);
  }
}

// s_main: attenuation -> PAR scan -> cell kernel -> particle sweep.
// s_aux : carbonate cells -> saturation-horizon scan.
// s_aux2: column integrals of the cell kernel's production profiles (diagnostics only).
// OBGC_SCHED picks when the carbonate kernel is released: 1 (default) when the cell kernel ends, so that its
// fp64 work overlaps the memory-bound sweep and the cell kernel has the register file to itself; 0 at t = 0;
// 2 = solved inside the cell kernel's wavefronts (measured slower: two waves per SIMD cannot feed the VALU).
// The native layout's default: the column-block kernel (attenuation, scan, cell terms and sweep in one launch) on s_main,
// then the saturation-horizon scan and the column integrals, which read what it wrote.  The carbonate system is solved
//   * by bgc_carbonate_kernel beside it on s_aux from the start of the call (schedule 4): the column kernel leaves VALU
//     issue slots idle while it waits on memory.  Default up to 10 M cells and whenever no diagnostics are written
//     (2.79 against 3.02 ms at 100 k columns x 60; 23.5 against 23.7 ms at 1 M x 80 without diagnostics);
//   * inside the column kernel's wavefronts at the end of every chunk (schedule 3): default above 10 M cells with
//     diagnostics, where the two tie (38.0-38.2 against 37.6-38.0 ms at 1 M x 80, 26.1 against 26.3 ms at 1 M x 60) and
//     this form is one launch plus its column scans and leaves s_aux idle.
// OBGC_SCHED = 0 forces schedule 4, 1 starts the carbonate kernel when the column kernel ends, 2 forces schedule 3.
int bgc_launch_column(obgc_handle* h, const BgcKernelArgs& K, bool want_zsat, const FusedDmArgs* F) {
  const int fuse = column_fuses(h, K.g, K.have_diag ? &K.diag : nullptr, want_zsat) ? 1 : 0;
  h->last_sched = fuse ? 3 : 4;
  FusedParams fp;
  fp.dms = h->dms;
  fp.macros = h->macros;
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_k0, h->s_main));
  launch_bgc_column(K, h->bgc, F, F ? &fp : nullptr, h->s_main, fuse);
  HIP_TRY(h, hipEventRecord(h->ev_k1, h->s_main));
  h->cell_valid = true;
  HIP_TRY(h, hipStreamWaitEvent(h->s_aux, (fuse || h->sched == 1) ? h->ev_k1 : h->ev_e0, 0));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_aux));
  if (!fuse) launch_bgc_carbonate(K, h->bgc, h->s_aux);
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_aux));
  if (want_zsat && !fuse) launch_bgc_zsat(K, h->s_aux);
  HIP_TRY(h, hipEventRecord(h->ev_aux_done, h->s_aux));
  HIP_TRY(h, hipStreamWaitEvent(h->s_aux2, h->ev_k1, 0));   // zint reads the column kernel's profiles
  launch_bgc_zint(K, h->bgc, h->s_aux2);
  HIP_TRY(h, hipEventRecord(h->ev_aux2_done, h->s_aux2));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux_done, 0));
  HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux2_done, 0));
  return OBGC_OK;
}

// the reciprocals of the shared divisors for the handle's parameters (BgcKernelArgs::recip), computed when they changed
inline int bgc_recips(obgc_handle* h, BgcKernelArgs& K) {
  if (!h->bgc_recip_valid) {
    HIP_TRY(h, bgc_recips_compute(h->bgc, h->bgc_recip, h->s_main));
    h->bgc_recip_valid = true;
  }
  std::memcpy(K.recip, h->bgc_recip, sizeof K.recip);
  return OBGC_OK;
}

// every launch set starts here or in bgc_launch_hybrid: the scratch it was given must lie inside the reserved arena
inline int arena_check(obgc_handle* h) {
  if (h->arena_used > h->arena_bytes) return fail(h, OBGC_ERR_INVALID, "internal: scratch carved beyond the reserved arena");
  return OBGC_OK;
}

int bgc_launch(obgc_handle* h, const BgcKernelArgs& K, bool want_zsat, const FusedDmArgs* F = nullptr) {
  if (int rc = arena_check(h)) return rc;
  // the scratch was carved for one form or the other (bgc_carve_scratch): the column-block kernel has no hand-off planes
  if (!K.g.kfast && K.scratch == nullptr) return bgc_launch_column(h, K, want_zsat, F);
  if (K.scratch == nullptr) return fail(h, OBGC_ERR_INVALID, "internal: the separate kernels were launched without their hand-off planes");
  // Large batches with diagnostics: the sweep (one wave per SIMD) is the longest kernel of the last phase and a
  // carbonate kernel beside it only slows it down (13.5 ms against 6.9 ms alone at 1 M columns), so the carbonate system
  // is solved inside the cell kernel's wavefronts instead: -4 % at 200 k - 1 M columns, +3 % at 100 k (measured).
  // Without diagnostics the large batches do best with the carbonate kernel started first (-4 % at 1 M columns, nothing at 100 k).
  const bool large = (long long)K.g.nlev * K.g.ncol > 10000000LL;
  const int sched = h->sched >= 0 ? h->sched : (large ? (K.have_diag ? 2 : 0) : 1);
  h->last_sched = sched;
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  if (F) {
    FusedParams fp;
    fp.dms = h->dms;
    fp.macros = h->macros;
    launch_fused_ecosys_cells(K, h->bgc, *F, fp, h->s_main, h->ev_k0, h->ev_k1, sched == 2);
  } else {
    launch_bgc_ecosys_cells(K, h->bgc, h->s_main, h->ev_k0, h->ev_k1, sched == 2);
  }
  h->cell_valid = true;
  HIP_TRY(h, hipStreamWaitEvent(h->s_aux, sched == 0 ? h->ev_e0 : h->ev_k1, 0));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_aux));
  if (sched != 2) launch_bgc_carbonate(K, h->bgc, h->s_aux);
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_aux));
  if (want_zsat) launch_bgc_zsat(K, h->s_aux);
  HIP_TRY(h, hipEventRecord(h->ev_aux_done, h->s_aux));
  HIP_TRY(h, hipStreamWaitEvent(h->s_aux2, h->ev_k1, 0));   // zint reads the cell kernel's profiles
  launch_bgc_zint(K, h->bgc, h->s_aux2);
  HIP_TRY(h, hipEventRecord(h->ev_aux2_done, h->s_aux2));
  if (K.split) launch_bgc_split_sweep(K, h->bgc, h->s_main);
  else launch_bgc_ecosys_sweep(K, h->bgc, h->s_main, h->sweep_lanes);
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux_done, 0));
  HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux2_done, 0));
  return OBGC_OK;
}

// Reference layout (level-fastest) on the device, hybrid scheme.  The cell-parallel kernels (attenuation, cell,
// carbonate) are layout-agnostic -- their lanes run along whatever axis is contiguous -- so they work on the caller's
// arrays directly.  Only the column kernels (PAR scan, sweep, column integrals, saturation horizons) need
// column-fastest rows; the ~50 planes they read and the ~40 they write are transposed through LDS tiles, one
// table-driven launch per batch, instead of all 200 planes of the call.  Per column tile of `tile` columns.
//   K : the caller's view (kfast) with engine scratch carved in the same index space
//   W : a dense column-fastest workspace view of one tile (own scratch), filled/drained here
int bgc_launch_hybrid(obgc_handle* h, const BgcKernelArgs& K, BgcKernelArgs W, bool want_zsat, int tile, int c_begin = 0,
                      int c_end = -1) {
  if (int rc = arena_check(h)) return rc;
  const GridView& v = K.g;
  const obgc_bgc_params& P = h->bgc;
  const long long wl = W.g.sk;
  const int staged[10] = {P.ind[1], P.ind[3], P.ind[4], P.ind[2], P.ind[0], P.ind[14], P.ind[13], P.ind[6], P.ind[8], P.ind[5]};
  if (c_end < 0) c_end = v.ncol;
  for (int c0 = c_begin; c0 < c_end; c0 += tile) {
    const int nc = (c_end - c0 < tile) ? (c_end - c0) : tile;
    const long long off = (long long)c0 * v.sc;
    // ---- the caller's view of this tile
    BgcKernelArgs Kt = K;
    Kt.g.ncol = nc;
    Kt.tracers += off; Kt.temp += off; Kt.salt += off; Kt.zmid += off; Kt.dz += off; Kt.zbot += off; Kt.fesed += off;
    Kt.tend += off; Kt.ph_prev += off; Kt.ph_prev_alt += off; Kt.scratch += off;
    if (Kt.chl100) Kt.chl100 += off;
    Kt.lat += c0; Kt.kmax += c0; Kt.dust += c0; Kt.sw += c0;
    if (Kt.co3_out) { Kt.co3_out += off; Kt.sat_calc_out += off; Kt.sat_arag_out += off; }
    if (Kt.z_photoC) { Kt.z_photoC += off; Kt.z_photoC_NO3 += off; Kt.z_CaCO3_form += off; Kt.z_bSi_form += off; }
#define X(n) if (Kt.diag.n) Kt.diag.n += off;
    OBGC_BGC_DIAG_KC(X)
    OBGC_BGC_DIAG_KCA(X)
#undef X
#define X(n) if (Kt.diag.n) Kt.diag.n += c0;
    OBGC_BGC_DIAG_CA(X)
    OBGC_BGC_DIAG_C(X)
#undef X
    // ---- the workspace view of this tile: per-column arrays are the caller's
    BgcKernelArgs Wt = W;
    Wt.g.ncol = nc;
    Wt.lat = Kt.lat; Wt.kmax = Kt.kmax; Wt.dust = Kt.dust; Wt.sw = Kt.sw;
#define X(n) Wt.diag.n = Kt.diag.n;
    OBGC_BGC_DIAG_CA(X)
    OBGC_BGC_DIAG_C(X)
#undef X
    auto kscr = [&](int slot) { return Kt.scratch + (long long)slot * Kt.splane; };
    auto wscr = [&](int slot) { return Wt.scratch + (long long)slot * Wt.splane; };
    auto ktr = [&](int slot1) { return Kt.tracers + (long long)(slot1 - 1) * v.plane; };
    auto wtr = [&](int slot1) { return const_cast<double*>(Wt.tracers) + (long long)(slot1 - 1) * Wt.g.plane; };

    HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
    // inputs of the column kernels that do not depend on this call's kernels
    PlaneTable tin; tin.n = 0;
    tin.add(Kt.dz, const_cast<double*>(Wt.dz)); tin.add(Kt.zbot, const_cast<double*>(Wt.zbot));
    tin.add(Kt.fesed, const_cast<double*>(Wt.fesed)); tin.add(Kt.temp, const_cast<double*>(Wt.temp));
    tin.add(Kt.zmid, const_cast<double*>(Wt.zmid));
    tin.add(ktr(P.ind[1]), wtr(P.ind[1])); tin.add(ktr(P.ind[4]), wtr(P.ind[4])); tin.add(ktr(P.ind[5]), wtr(P.ind[5]));
    launch_transpose_table_k2c(tin, v.nlev, nc, v.sc, wl, h->s_main);
    if (c0 == c_begin) HIP_TRY(h, hipEventRecord(h->ev_t0, h->s_main));
    // light: attenuation per cell (caller layout) -> scan per column (workspace) -> PAR entering each cell
    launch_bgc_att(Kt, P, h->s_main);
    PlaneTable t1; t1.n = 0; t1.add(kscr(S_ATT), wscr(S_ATT));
    launch_transpose_table_k2c(t1, v.nlev, nc, v.sc, wl, h->s_main);
    launch_bgc_parscan(Wt, h->s_main);
    PlaneTable t2; t2.n = 0; t2.add(wscr(S_PARIN), kscr(S_PARIN));
    launch_transpose_table_c2k(t2, v.nlev, nc, wl, v.sc, nullptr, h->s_main);
    HIP_TRY(h, hipEventRecord(h->ev_k0, h->s_main));
    launch_bgc_cell(Kt, P, h->s_main, 0);
    HIP_TRY(h, hipEventRecord(h->ev_k1, h->s_main));
    h->cell_valid = true;
    h->last_sched = 1;
    // hand-off of the cell kernel: production terms, parked tendencies (+ DIC_ALT_CO2's plane, whose zero fill must
    // survive), with diagnostics the local integrands and the four production profiles
    PlaneTable th; th.n = 0;
    th.add(kscr(S_PRODPOC), wscr(S_PRODPOC)); th.add(kscr(S_PRODCA), wscr(S_PRODCA)); th.add(kscr(S_PRODSI), wscr(S_PRODSI));
    th.add(kscr(S_PFELOC), wscr(S_PFELOC));
    for (int i = 0; i < 10; ++i) th.add(Kt.tend + (long long)(staged[i] - 1) * v.plane, Wt.tend + (long long)(staged[i] - 1) * Wt.g.plane);
    if (K.have_diag) {
      for (int sl = S_OC0; sl < S_NDIAG; ++sl) th.add(kscr(sl), wscr(sl));
      for (int a = 0; a < OBGC_AUTOTROPH_CNT; ++a) {
        th.add(Kt.z_photoC + a * v.plane, Wt.z_photoC + a * Wt.g.plane);
        th.add(Kt.z_photoC_NO3 + a * v.plane, Wt.z_photoC_NO3 + a * Wt.g.plane);
        th.add(Kt.z_CaCO3_form + a * v.plane, Wt.z_CaCO3_form + a * Wt.g.plane);
        th.add(Kt.z_bSi_form + a * v.plane, Wt.z_bSi_form + a * Wt.g.plane);
      }
    }
    launch_transpose_table_k2c(th, v.nlev, nc, v.sc, wl, h->s_main);
    HIP_TRY(h, hipEventRecord(h->ev_t1, h->s_main));
    // carbonate on the caller's layout beside the sweep (not beside the bandwidth-bound transposes above, which its
    // small workgroups would starve); its three profiles feed the saturation-horizon scan
    HIP_TRY(h, hipStreamWaitEvent(h->s_aux, h->ev_t1, 0));
    HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_aux));
    launch_bgc_carbonate(Kt, P, h->s_aux);
    HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_aux));
    if (want_zsat) {
      PlaneTable tz; tz.n = 0;
      tz.add(Kt.co3_out, Wt.co3_out); tz.add(Kt.sat_calc_out, Wt.sat_calc_out); tz.add(Kt.sat_arag_out, Wt.sat_arag_out);
      launch_transpose_table_k2c(tz, v.nlev, nc, v.sc, wl, h->s_aux);
      launch_bgc_zsat(Wt, h->s_aux);
    }
    HIP_TRY(h, hipEventRecord(h->ev_aux_done, h->s_aux));
    HIP_TRY(h, hipStreamWaitEvent(h->s_aux2, h->ev_t1, 0));
    launch_bgc_zint(Wt, P, h->s_aux2);
    HIP_TRY(h, hipEventRecord(h->ev_aux2_done, h->s_aux2));
    launch_bgc_ecosys_sweep(Wt, P, h->s_main, h->sweep_lanes);
    HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
    // results of the sweep: 11 tendencies + its level-dependent diagnostics, active cells only
    PlaneTable to; to.n = 0;
    for (int i = 0; i < 10; ++i) to.add(Wt.tend + (long long)(staged[i] - 1) * Wt.g.plane, Kt.tend + (long long)(staged[i] - 1) * v.plane);
    to.add(Wt.tend + (long long)(P.ind[7] - 1) * Wt.g.plane, Kt.tend + (long long)(P.ind[7] - 1) * v.plane);
    if (K.have_diag) {
#define X(n) if (Kt.diag.n) to.add(Wt.diag.n, Kt.diag.n);
      OBGC_SWEEP_DIAG_KC(X)
#undef X
    }
    launch_transpose_table_c2k(to, v.nlev, nc, wl, v.sc, Kt.kmax, h->s_main);
    HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux_done, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->s_main, h->ev_aux2_done, 0));
  }
  return OBGC_OK;
}

}  // namespace

// Host-memory callers, reference layout: the call is bound by the PCIe link (1.8 GB up, 7.8 GB down at cfg2 with all
// diagnostics against 5 ms of kernels), and the link is full duplex.  The columns are cut into chunks; this thread
// uploads chunk i+1 while the GPU works on chunk i and a second thread downloads the results of chunk i-1 -- columns
// are independent, so a chunk's outputs are final as soon as its kernels are.  A column range of one plane is one
// contiguous piece of the reference layout.  Two threads because copies from pageable memory block their caller.
int host_pipeline(obgc_handle* h, Stager& st, const GridView& v, const BgcKernelArgs& K, const BgcKernelArgs& W, bool want_zsat,
                  int tile_cols, int nchunk) {
  if (!h->s_up) HIP_TRY(h, hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking));
  if (!h->s_down) HIP_TRY(h, hipStreamCreateWithFlags(&h->s_down, hipStreamNonBlocking));
  int chunk = (v.ncol + nchunk - 1) / nchunk;
  chunk = (chunk + 63) / 64 * 64;
  nchunk = (v.ncol + chunk - 1) / chunk;
  std::vector<hipEvent_t> ev_up((size_t)nchunk, nullptr), ev_done((size_t)nchunk, nullptr);
  hipError_t ce = hipSuccess;
  for (int i = 0; i < nchunk && ce == hipSuccess; ++i) {
    ce = hipEventCreateWithFlags(&ev_up[(size_t)i], hipEventDisableTiming);
    if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ev_done[(size_t)i], hipEventDisableTiming);
  }
  auto destroy_events = [&] {
    for (auto e : ev_up) if (e) (void)hipEventDestroy(e);
    for (auto e : ev_done) if (e) (void)hipEventDestroy(e);
  };
  if (ce != hipSuccess) { destroy_events(); return fail(h, OBGC_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(ce)); }

  std::mutex m;
  std::condition_variable cv;
  int ready = 0;            // chunks whose kernels (and ev_done) are enqueued
  bool stop = false;
  hipError_t down_err = hipSuccess;
  auto download_chunk = [&](int i, hipStream_t stream) -> hipError_t {
    const int c0 = i * chunk, c1 = (c0 + chunk < v.ncol) ? c0 + chunk : v.ncol;
    hipError_t e = hipStreamWaitEvent(stream, ev_done[(size_t)i], 0);
    for (size_t k = 0; k < st.items.size() && e == hipSuccess; ++k) {
      const Stage& it = st.items[k];
      if (!it.out || it.nplanes <= 0) continue;
      for (int n = 0; n < it.nplanes && e == hipSuccess; ++n) {
        size_t b, len;
        chunk_range(it, v, n, c0, c1, b, len);
        if (len) e = hipMemcpyAsync((char*)it.host + b, (char*)it.dev + b, len, hipMemcpyDeviceToHost, stream);
      }
    }
    return e;
  };
  std::thread downloader;
  bool threaded = h->host_threads != 0;
  try {
    if (threaded) downloader = std::thread([&] {
      hipError_t e = hipSetDevice(h->device);
      for (int i = 0; i < nchunk && e == hipSuccess; ++i) {
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return ready > i || stop; });
          if (ready <= i) break;   // the producer gave up
        }
        e = download_chunk(i, h->s_down);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(h->s_down);
      down_err = e;
    });
  } catch (...) {
    threaded = false;   // no second thread to be had: same chunks, downloads in line
  }

  int rc = OBGC_OK;
  hipError_t e = hipSuccess;
  for (int i = 0; i < nchunk && e == hipSuccess && rc == OBGC_OK; ++i) {
    const int c0 = i * chunk, c1 = (c0 + chunk < v.ncol) ? c0 + chunk : v.ncol;
    for (size_t k = 0; k < st.items.size() && e == hipSuccess; ++k) {
      const Stage& it = st.items[k];
      if (!it.in || it.nplanes <= 0) continue;
      for (int n = 0; n < it.nplanes && e == hipSuccess; ++n) {
        size_t b, len;
        chunk_range(it, v, n, c0, c1, b, len);
        if (len) e = hipMemcpyAsync((char*)it.dev + b, (const char*)it.host + b, len, hipMemcpyHostToDevice, h->s_up);
      }
    }
    if (e == hipSuccess) e = hipEventRecord(ev_up[(size_t)i], h->s_up);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->s_main, ev_up[(size_t)i], 0);
    if (e != hipSuccess) break;
    rc = bgc_launch_hybrid(h, K, W, want_zsat, tile_cols, c0, c1);
    if (rc != OBGC_OK) break;
    e = hipEventRecord(ev_done[(size_t)i], h->s_main);
    if (e != hipSuccess) break;
    if (!threaded) {
      e = download_chunk(i, h->s_main);
      continue;
    }
    {
      std::lock_guard<std::mutex> lk(m);
      ready = i + 1;
    }
    cv.notify_one();
  }
  {
    std::lock_guard<std::mutex> lk(m);
    stop = true;
  }
  cv.notify_one();
  if (threaded) downloader.join();
  (void)hipStreamSynchronize(h->s_main);
  destroy_events();
  if (rc != OBGC_OK) return rc;
  if (e != hipSuccess) return fail(h, OBGC_ERR_HIP, std::string("host pipeline (upload): ") + hipGetErrorString(e));
  if (down_err != hipSuccess) return fail(h, OBGC_ERR_HIP, std::string("host pipeline (download): ") + hipGetErrorString(down_err));
  return OBGC_OK;
}

// ======================================================================== BGC_SourceSink ==

extern "C" int obgc_bgc_source_sink(obgc_handle* h, const obgc_bgc_source_sink_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (!a->tracers || !a->temp || !a->salt || !a->zmid || !a->dz || !a->zbot || !a->lat || !a->kmax || !a->fesedflux ||
      !a->dust_flux_in || !a->sw_flux || !a->tend || !a->ph_prev || !a->ph_prev_alt)
    return fail(h, OBGC_ERR_INVALID, "obgc_bgc_source_sink: null array");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0) return OBGC_OK;
  if (v.tiled) {
    if (int rc = tiled_ok(h, a->diag, "obgc_bgc_source_sink")) return rc;
  }
  h->arena_used = 0;
  h->timed_valid = false;
  h->tr_valid = false;
  h->cell_valid = false;

  // ---- stage caller arrays (no-op for device memory) ----
  Stager st(h, a->grid.memspace == OBGC_MEM_HOST);
  const int i_tr = st.add_kc(a->tracers, v, OBGC_BGC_NTRACER, true, false);
  const int i_T = st.add_kc(a->temp, v, 1, true, false), i_S = st.add_kc(a->salt, v, 1, true, false);
  const int i_zm = st.add_kc(a->zmid, v, 1, true, false), i_dz = st.add_kc(a->dz, v, 1, true, false);
  const int i_zb = st.add_kc(a->zbot, v, 1, true, false), i_fs = st.add_kc(a->fesedflux, v, 1, true, false);
  const int i_lat = st.add(a->lat, v.ncol * sizeof(double), true, false);
  const int i_km = st.add(a->kmax, v.ncol * sizeof(int), true, false);
  const int i_du = st.add(a->dust_flux_in, v.ncol * sizeof(double), true, false);
  const int i_sw = st.add(a->sw_flux, v.ncol * sizeof(double), true, false);
  // the host path copies whole strided spans back, so outputs with gaps (ld > extent) are read in first
  const bool gaps = st.host && ((v.kfast ? v.sc != v.nlev : v.sk != v.ncol) || v.plane != (v.kfast ? v.sc * v.ncol : v.sk * v.nlev));
  const int i_td = st.add_kc(a->tend, v, OBGC_BGC_NTRACER, gaps, true);
  const int i_pp = st.add_kc(a->ph_prev, v, 1, true, true), i_pa = st.add_kc(a->ph_prev_alt, v, 1, true, true);
  obgc_bgc_diag dd;
  std::memset(&dd, 0, sizeof dd);
  std::vector<std::pair<double**, int>> dslots;
  if (a->diag) {
    const obgc_bgc_diag& D = *a->diag;
    const bool gaps_c = st.host && v.ldc != v.ncol;
    // every (k,column) diagnostic is written in full (values or the zero fill), so only padding has to be read in first;
    // the three members neither the reference nor the engine touches (POC_ACCUM, DONr_remin, DOPr_remin) do not travel at all
#define X(n)                                                                                          \
  if (D.n) {                                                                                          \
    const bool untouched = &D.n == &D.POC_ACCUM || &D.n == &D.DONr_remin || &D.n == &D.DOPr_remin;   \
    dslots.push_back({&dd.n, st.add_kc(D.n, v, 1, gaps && !untouched, !untouched)});                  \
  }
    OBGC_BGC_DIAG_KC(X)
#undef X
#define X(n) if (D.n) dslots.push_back({&dd.n, st.add_kc(D.n, v, 4, gaps, true)});
    OBGC_BGC_DIAG_KCA(X)
#undef X
#define X(n) if (D.n) dslots.push_back({&dd.n, st.add(D.n, span_c(v, 4), gaps_c, true)});
    OBGC_BGC_DIAG_CA(X)
#undef X
#define X(n) if (D.n) dslots.push_back({&dd.n, st.add(D.n, v.ncol * sizeof(double), false, true)});
    OBGC_BGC_DIAG_C(X)
#undef X
  }
  const bool want_zsat = a->diag && (a->diag->zsatcalc || a->diag->zsatarag);
  // The reference layout is transposed through LDS tiles into a native workspace, one column tile at a
  // time, so that every kernel runs on coalesced column-fastest rows (OBGC_KFAST_DIRECT=1 keeps the
  // generic-stride path for A/B measurements).
  const bool tiled = v.kfast && !h->kfast_direct;
  const bool hybrid = tiled && h->kfast_hybrid;
  const int tile_cols = tiled ? (v.ncol < h->kfast_tile ? v.ncol : h->kfast_tile) : v.ncol;
  GridView kv = v;                       // the view the kernels see
  if (tiled) {
    kv.kfast = 0; kv.ncol = tile_cols; kv.sk = tile_cols; kv.sc = 1; kv.plane = (long long)tile_cols * v.nlev;
  }
  size_t work = 0;
  const size_t tplane = (size_t)tile_cols * v.nlev * sizeof(double);
  int n_kc_out = 0, n_kca_out = 0;
  if (tiled) {
    if (a->diag) {
#define X(n) if (a->diag->n) ++n_kc_out;
      OBGC_BGC_DIAG_KC(X)
#undef X
#define X(n) if (a->diag->n) ++n_kca_out;
      OBGC_BGC_DIAG_KCA(X)
#undef X
    }
    work = align_up(tplane * (size_t)(30 + 6 + 2 + 30 + n_kc_out + 4 * n_kca_out));
  }
  obgc_bgc_diag wdiag0;                   // hybrid: the workspace owns every level-dependent plane it uses
  std::memset(&wdiag0, 0, sizeof wdiag0);
  const bool column = !tiled && use_column_kernel(h, v);   // generic strides (OBGC_KFAST_DIRECT=1) keep the separate kernels
  const bool split = !tiled && use_split_sweep(h, v, column);
  size_t scratch = bgc_scratch_bytes(h, kv, a->diag, want_zsat, tiled, column, split);
  // the tile workspace runs whichever kernel form the handle selects: reserve for the larger of the two
  if (tiled && !hybrid) scratch = std::max(scratch, bgc_scratch_bytes(h, kv, a->diag, want_zsat, tiled, h->column != 0));
  if (hybrid) {
    work = align_up(tplane * (size_t)(30 + 5 + 30 + 27));
    scratch = bgc_scratch_bytes(h, v, a->diag, want_zsat, false) + bgc_scratch_bytes(h, kv, a->diag ? &wdiag0 : nullptr, want_zsat, true);
  }
  int rc = arena_reserve(h, (st.host ? st.need : 0) + work + scratch);
  if (rc != OBGC_OK) return rc;
  // host memory in the reference layout: overlap the two directions of the link and the kernels, chunk by chunk
  const bool chunked = st.host && hybrid && h->host_chunks > 1 && v.ncol >= 64 * h->host_chunks && st.need >= h->host_pipeline_min_bytes;
  rc = stage_in(h, st, chunked);
  if (rc != OBGC_OK) return rc;
  for (auto& s : dslots) *s.first = st.dev<double>(s.second);

  BgcKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  K.tracers = st.dev<const double>(i_tr);
  K.temp = st.dev<const double>(i_T); K.salt = st.dev<const double>(i_S); K.zmid = st.dev<const double>(i_zm);
  K.dz = st.dev<const double>(i_dz); K.zbot = st.dev<const double>(i_zb); K.fesed = st.dev<const double>(i_fs);
  K.lat = st.dev<const double>(i_lat); K.kmax = st.dev<const int>(i_km); K.dust = st.dev<const double>(i_du);
  K.sw = st.dev<const double>(i_sw);
  K.tend = st.dev<double>(i_td); K.ph_prev = st.dev<double>(i_pp); K.ph_prev_alt = st.dev<double>(i_pa);
  K.diag = dd;
  K.have_diag = a->diag ? 1 : 0;
  K.alt_co2_use_eco = a->alt_co2_use_eco;
  rc = bgc_recips(h, K);
  if (rc != OBGC_OK) return rc;

  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  if (!tiled) {
    bgc_carve_scratch(h, K, want_zsat, column, split);
    rc = bgc_launch(h, K, want_zsat);
    if (rc != OBGC_OK) return rc;
  } else if (hybrid) {
    bgc_carve_scratch(h, K, want_zsat);               // scratch of the cell-parallel kernels, caller's index space
    auto carve = [&](size_t bytes) -> double* {
      double* p = (double*)(h->arena + h->arena_used);
      h->arena_used += align_up(bytes);
      return p;
    };
    double* w = carve(work);
    auto take = [&](int nplanes) -> double* { double* p = w; w += (tplane / sizeof(double)) * nplanes; return p; };
    BgcKernelArgs W = K;
    W.g = kv;
    W.tracers = take(30);
    W.dz = take(1); W.zbot = take(1); W.fesed = take(1); W.temp = take(1); W.zmid = take(1);
    W.salt = nullptr;
    W.tend = take(30);
    W.ph_prev = W.ph_prev_alt = nullptr;
    W.co3_out = W.sat_calc_out = W.sat_arag_out = nullptr;
    W.z_photoC = W.z_photoC_NO3 = W.z_CaCO3_form = W.z_bSi_form = nullptr;
    std::memset(&W.diag, 0, sizeof W.diag);
    bgc_carve_scratch(h, W, want_zsat);               // W.diag is empty here: profiles and CO3/sat planes come from the arena
    if (a->diag) {
#define X(n) if (dd.n) W.diag.n = take(1);
      OBGC_SWEEP_DIAG_KC(X)
#undef X
    }
    rc = chunked ? host_pipeline(h, st, v, K, W, want_zsat, tile_cols, h->host_chunks) : bgc_launch_hybrid(h, K, W, want_zsat, tile_cols);
    if (rc != OBGC_OK) return rc;
    h->tr_valid = true;
  } else {
    // ---- native workspace (dense, column-fastest) ----
    auto carve = [&](size_t bytes) -> double* {
      double* p = (double*)(h->arena + h->arena_used);
      h->arena_used += align_up(bytes);
      return p;
    };
    double* w = carve(work);
    auto take = [&](int nplanes) -> double* { double* p = w; w += (tplane / sizeof(double)) * nplanes; return p; };
    double* w_tr = take(30);
    double *w_T = take(1), *w_S = take(1), *w_zm = take(1), *w_dz = take(1), *w_zb = take(1), *w_fs = take(1);
    double *w_pp = take(1), *w_pa = take(1), *w_td = take(30);
    BgcKernelArgs T = K;
    T.g = kv;
    T.tracers = w_tr; T.temp = w_T; T.salt = w_S; T.zmid = w_zm; T.dz = w_dz; T.zbot = w_zb; T.fesed = w_fs;
    T.tend = w_td; T.ph_prev = w_pp; T.ph_prev_alt = w_pa;
    struct Out { double* caller; double* ws; int planes; };
    std::vector<Out> outs;
    outs.push_back({K.tend, w_td, 30});
    outs.push_back({K.ph_prev, w_pp, 1});
    outs.push_back({K.ph_prev_alt, w_pa, 1});
    if (a->diag) {
#define X(n) if (dd.n) { T.diag.n = take(1); outs.push_back({dd.n, T.diag.n, 1}); }
      OBGC_BGC_DIAG_KC(X)
#undef X
#define X(n) if (dd.n) { T.diag.n = take(4); outs.push_back({dd.n, T.diag.n, 4}); }
      OBGC_BGC_DIAG_KCA(X)
#undef X
      // never written by the engine (nor by the reference): nothing to bring back
      for (auto it = outs.begin(); it != outs.end();)
        if (it->caller == dd.POC_ACCUM || it->caller == dd.DONr_remin || it->caller == dd.DOPr_remin) it = outs.erase(it);
        else ++it;
    }
    bgc_carve_scratch(h, T, want_zsat, h->column != 0);   // (sized for the separate kernels, which need more)
    const size_t arena_mark = h->arena_used;
    (void)arena_mark;
    HIP_TRY(h, hipEventRecord(h->ev_t0, h->s_main));
    for (int c0 = 0; c0 < v.ncol; c0 += tile_cols) {
      const int nc = (v.ncol - c0 < tile_cols) ? (v.ncol - c0) : tile_cols;
      BgcKernelArgs Tt = T;
      Tt.g.ncol = nc;                      // a short last tile keeps the workspace strides of a full one
      Tt.lat = K.lat + c0; Tt.kmax = K.kmax + c0; Tt.dust = K.dust + c0; Tt.sw = K.sw + c0;
#define X(n) if (dd.n) Tt.diag.n = dd.n + c0;
      OBGC_BGC_DIAG_CA(X)
      OBGC_BGC_DIAG_C(X)
#undef X
      const long long off = (long long)c0 * v.sc;
      const long long wl = kv.sk, wp = kv.plane;
      launch_transpose_k2c(K.tracers + off, w_tr, v.nlev, nc, v.sc, v.plane, wl, wp, 30, h->s_main);
      const double* ins[8] = {K.temp, K.salt, K.zmid, K.dz, K.zbot, K.fesed, K.ph_prev, K.ph_prev_alt};
      double* wins[8] = {w_T, w_S, w_zm, w_dz, w_zb, w_fs, w_pp, w_pa};
      for (int i = 0; i < 8; ++i) launch_transpose_k2c(ins[i] + off, wins[i], v.nlev, nc, v.sc, 0, wl, 0, 1, h->s_main);
      if (c0 == 0) HIP_TRY(h, hipEventRecord(h->ev_t1, h->s_main));
      rc = bgc_launch(h, Tt, want_zsat);
      if (rc != OBGC_OK) return rc;
      for (const Out& o : outs)
        launch_transpose_c2k(o.ws, o.caller + off, v.nlev, nc, wl, wp, v.sc, v.plane, o.planes, h->s_main);
    }
    h->tr_valid = true;
  }
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;

  rc = stage_out(h, st, chunked);
  if (rc != OBGC_OK) return rc;
  return finish(h, st.host);
}

// ============================================================================ fused step ==
// BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink of one batch in one call (BASELINE config "BGC + DMS + MACROS fused
// tendencies").  The kernels are those of obgc_bgc_source_sink with the other two modules' cell terms folded into the
// attenuation, scan and cell kernels: one light pass, the shared tracers read once.

extern "C" int obgc_fused_step(obgc_handle* h, const obgc_fused_step_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  const obgc_bgc_source_sink_args& b = a->bgc;
  if (!grid_ok(b.grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (!a->dms && !a->prot) return obgc_bgc_source_sink(h, &a->bgc);   // nothing to fuse
  if (b.grid.memspace != OBGC_MEM_DEVICE || b.grid.layout == OBGC_LAYOUT_KFAST)
    return fail(h, OBGC_ERR_UNSUPPORTED,
                "obgc_fused_step: device pointers in the native column-fastest or column-block layout only; for host memory or the "
                "reference layout call obgc_bgc_source_sink, obgc_dms_source_sink and obgc_macros_source_sink");
  if (b.grid.layout == OBGC_LAYOUT_TILED) {
    if (int rc = tiled_ok(h, b.diag, "obgc_fused_step")) return rc;
  }
  if (b.grid.ncol == 0) return OBGC_OK;
  if (!b.tracers || !b.temp || !b.salt || !b.zmid || !b.dz || !b.zbot || !b.lat || !b.kmax || !b.fesedflux || !b.dust_flux_in ||
      !b.sw_flux || !b.tend || !b.ph_prev || !b.ph_prev_alt)
    return fail(h, OBGC_ERR_INVALID, "obgc_fused_step: null BGC array");
  if (a->dms && (!a->dmsp || !a->SST || !a->dms_tend || !a->dmsp_tend)) return fail(h, OBGC_ERR_INVALID, "obgc_fused_step: null DMS array");
  if (a->prot && (!a->poly || !a->lip || !a->prot_tend || !a->poly_tend || !a->lip_tend))
    return fail(h, OBGC_ERR_INVALID, "obgc_fused_step: null MACROS array");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(b.grid);
  h->arena_used = 0;
  h->timed_valid = false;
  h->tr_valid = false;
  h->cell_valid = false;
  const bool want_zsat = b.diag && (b.diag->zsatcalc || b.diag->zsatarag);
  const size_t kc1 = span_kc(v, 1);
  // (the column-block kernel needs none of it: DMS's light runs through LDS beside BGC's)
  const bool column = use_column_kernel(h, v);
  const size_t dms_scratch = (a->dms && !column) ? 3 * align_up(kc1) + align_up((size_t)v.ncol * sizeof(int)) : 0;
  const bool split = use_split_sweep(h, v, column);
  int rc = arena_reserve(h, bgc_scratch_bytes(h, v, b.diag, want_zsat, false, column, split) + dms_scratch);
  if (rc != OBGC_OK) return rc;

  BgcKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  K.tracers = b.tracers;
  K.temp = b.temp; K.salt = b.salt; K.zmid = b.zmid; K.dz = b.dz; K.zbot = b.zbot; K.fesed = b.fesedflux;
  K.lat = b.lat; K.kmax = b.kmax; K.dust = b.dust_flux_in; K.sw = b.sw_flux;
  K.tend = b.tend; K.ph_prev = b.ph_prev; K.ph_prev_alt = b.ph_prev_alt;
  if (b.diag) K.diag = *b.diag;
  K.have_diag = b.diag ? 1 : 0;
  K.alt_co2_use_eco = b.alt_co2_use_eco;
  rc = bgc_recips(h, K);
  if (rc != OBGC_OK) return rc;
  bgc_carve_scratch(h, K, want_zsat, column, split);

  FusedDmArgs F;
  std::memset(&F, 0, sizeof F);
  F.dms = a->dms; F.dmsp = a->dmsp; F.SST = a->SST; F.dms_tend = a->dms_tend; F.dmsp_tend = a->dmsp_tend;
  F.prot = a->prot; F.poly = a->poly; F.lip = a->lip; F.prot_tend = a->prot_tend; F.poly_tend = a->poly_tend; F.lip_tend = a->lip_tend;
  if (a->dms && a->dms_diag) { F.ddiag = *a->dms_diag; F.have_ddiag = 1; }
  if (a->prot && a->macros_diag) { F.mdiag = *a->macros_diag; F.have_mdiag = 1; }
  F.f_qsw_par_dms = h->dms.f_qsw_par_DMS;
  F.all_flagged = (h->dms.f_qsw_par_DMS != 0.45) ? 1 : 0;   // BGC's f_qsw_par is the constant 0.45 (BGC_parms.F90:454-455)
  if (a->dms && !column) {
    auto carve = [&](size_t bytes) -> void* {
      void* p = h->arena + h->arena_used;
      h->arena_used += align_up(bytes);
      return p;
    };
    F.d_att = (double*)carve(kc1); F.d_kpardz = (double*)carve(kc1); F.d_parin = (double*)carve(kc1);
    F.flag = (int*)carve((size_t)v.ncol * sizeof(int));   // cleared by the first kernel of the launch set
  }
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  rc = bgc_launch(h, K, want_zsat, &F);
  if (rc != OBGC_OK) return rc;
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  return finish(h, false);
}

// *_SurfaceFluxes read the surface row of a few tracers (BGC_mod.F90:2816-2822, DMS_mod.F90:834).  A host-memory caller
// therefore does not ship the (k,column,n) array: the rows are gathered on the host into `rows` (slot-major, dense in
// the column) and land in a device array that the kernel addresses as a one-level, column-fastest tracer array.
struct SurfaceRows {
  std::vector<double> rows;   // alive until the call has synchronised
  int item = -1;              // Stager item of the device copy (ntracer x ncol doubles, only the gathered slots are filled)
};

void gather_surface_rows(Stager& st, const GridView& v, const double* tracers, int ntracer, const int* slots, int nslots,
                         SurfaceRows& out) {
  out.rows.resize((size_t)nslots * (size_t)v.ncol);
  for (int i = 0; i < nslots; ++i) {
    const double* src = tracers + (long long)(slots[i] - 1) * v.plane;
    double* dst = out.rows.data() + (size_t)i * (size_t)v.ncol;
    for (int c = 0; c < v.ncol; ++c) dst[c] = src[v.at(0, c)];
  }
  out.item = st.add(out.rows.data(), (size_t)ntracer * (size_t)v.ncol * sizeof(double), false, false);   // device room only
}

int upload_surface_rows(obgc_handle* h, const Stager& st, const GridView& v, const int* slots, int nslots, const SurfaceRows& r) {
  double* dev = st.dev<double>(r.item);
  for (int i = 0; i < nslots; ++i)
    HIP_TRY(h, hipMemcpyAsync(dev + (size_t)(slots[i] - 1) * (size_t)v.ncol, r.rows.data() + (size_t)i * (size_t)v.ncol,
                              (size_t)v.ncol * sizeof(double), hipMemcpyHostToDevice, h->s_main));
  return OBGC_OK;
}

inline GridView surface_rows_view(GridView v) {
  v.kfast = 0; v.nlev = 1; v.sc = 1; v.sk = v.ncol; v.plane = v.ncol;   // ldc stays the caller's
  return v;
}

// ===================================================================== BGC_SurfaceFluxes ==

extern "C" int obgc_bgc_surface_fluxes(obgc_handle* h, const obgc_bgc_surface_fluxes_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (!a->tracers || !a->surfacePressure || !a->windSpeedSquared10m || !a->atmCO2 || !a->atmCO2_ALT_CO2 ||
      !a->surfaceDepth || !a->SST || !a->SSS || !a->iceFraction || !a->surface_pH || !a->surface_pH_alt_co2 ||
      !a->depositionFlux || !a->riverFlux || !a->gasFlux || !a->seaIceFlux || !a->netFlux)
    return fail(h, OBGC_ERR_INVALID, "obgc_bgc_surface_fluxes: null array");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0) return OBGC_OK;
  h->arena_used = 0;
  h->timed_valid = false;
  Stager st(h, a->grid.memspace == OBGC_MEM_HOST);
  const size_t cb = v.ncol * sizeof(double), fb = span_c(v, OBGC_BGC_NTRACER);
  const int surf_slots[6] = {h->bgc.ind[6], h->bgc.ind[7], h->bgc.ind[8], h->bgc.ind[0], h->bgc.ind[2], h->bgc.ind[5]};
  SurfaceRows srows;
  int i_tr;
  if (st.host) { gather_surface_rows(st, v, a->tracers, OBGC_BGC_NTRACER, surf_slots, 6, srows); i_tr = srows.item; }
  else i_tr = st.add(a->tracers, span_kc(v, OBGC_BGC_NTRACER), true, false);
  const int i_p = st.add(a->surfacePressure, cb, true, false), i_w = st.add(a->windSpeedSquared10m, cb, true, false);
  const int i_c1 = st.add(a->atmCO2, cb, true, false), i_c2 = st.add(a->atmCO2_ALT_CO2, cb, true, false);
  const int i_sd = st.add(a->surfaceDepth, cb, true, false), i_t = st.add(a->SST, cb, true, false);
  const int i_s = st.add(a->SSS, cb, true, false), i_ice = st.add(a->iceFraction, cb, true, true);
  const int i_ph = st.add(a->surface_pH, cb, true, true), i_pa = st.add(a->surface_pH_alt_co2, cb, true, true);
  const int i_de = st.add(a->depositionFlux, fb, true, true), i_ri = st.add(a->riverFlux, fb, true, true);
  const int i_ga = st.add(a->gasFlux, fb, true, true), i_si = st.add(a->seaIceFlux, fb, true, true);
  const int i_ne = st.add(a->netFlux, fb, st.host && v.ldc != v.ncol, true);
  obgc_bgc_flux_diag dd;
  std::memset(&dd, 0, sizeof dd);
  std::vector<std::pair<double**, int>> dslots;
  if (a->diag) {
#define X(n) if (a->diag->n) dslots.push_back({&dd.n, st.add(a->diag->n, cb, false, true)});
    OBGC_BGC_FLUX_DIAG(X)
#undef X
  }
  int rc = stage_in(h, st);
  if (rc != OBGC_OK) return rc;
  for (auto& s : dslots) *s.first = st.dev<double>(s.second);
  SurfKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  if (st.host) {
    rc = upload_surface_rows(h, st, v, surf_slots, 6, srows);
    if (rc != OBGC_OK) return rc;
    K.g = surface_rows_view(v);
  }
  K.tracers = st.dev<const double>(i_tr);
  K.surfacePressure = st.dev<const double>(i_p); K.wind2 = st.dev<const double>(i_w); K.atmCO2 = st.dev<const double>(i_c1);
  K.atmCO2_alt = st.dev<const double>(i_c2); K.surfaceDepth = st.dev<const double>(i_sd); K.SST = st.dev<const double>(i_t);
  K.SSS = st.dev<const double>(i_s); K.iceFraction = st.dev<double>(i_ice); K.surface_pH = st.dev<double>(i_ph);
  K.surface_pH_alt = st.dev<double>(i_pa); K.depo = st.dev<double>(i_de); K.river = st.dev<double>(i_ri);
  K.gas = st.dev<double>(i_ga); K.seaice = st.dev<double>(i_si); K.net = st.dev<double>(i_ne);
  K.diag = dd;
  K.have_diag = a->diag ? 1 : 0;
  K.lcalc_o2 = a->lcalc_O2_gas_flux;
  K.lcalc_co2 = a->lcalc_CO2_gas_flux;
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  launch_bgc_surface(K, h->bgc, h->s_main);
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  rc = stage_out(h, st);
  if (rc != OBGC_OK) return rc;
  return finish(h, st.host);
}

// =================================================================================== DMS ==

extern "C" int obgc_dms_source_sink(obgc_handle* h, const obgc_dms_source_sink_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (!a->tracers || !a->dz || !a->kmax || !a->SST || !a->sw_flux || !a->tend)
    return fail(h, OBGC_ERR_INVALID, "obgc_dms_source_sink: null array");
  if (int rc = no_tiled(h, a->grid, "obgc_dms_source_sink")) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0) return OBGC_OK;
  h->arena_used = 0;
  h->timed_valid = false;
  Stager st(h, a->grid.memspace == OBGC_MEM_HOST);
  const size_t kc1 = span_kc(v, 1), cb = v.ncol * sizeof(double);
  const bool gaps = st.host && ((v.kfast ? v.sc != v.nlev : v.sk != v.ncol) || v.plane != (v.kfast ? v.sc * v.ncol : v.sk * v.nlev));
  const int i_tr = st.add(a->tracers, span_kc(v, OBGC_DMS_NTRACER), true, false);
  const int i_dz = st.add(a->dz, kc1, true, false), i_km = st.add(a->kmax, v.ncol * sizeof(int), true, false);
  const int i_t = st.add(a->SST, cb, true, false), i_sw = st.add(a->sw_flux, cb, true, false);
  const int i_td = st.add(a->tend, span_kc(v, OBGC_DMS_NTRACER), gaps, true);
  obgc_dms_diag dd;
  std::memset(&dd, 0, sizeof dd);
  std::vector<std::pair<double**, int>> dslots;
  if (a->diag) {
#define X(n) if (a->diag->n) dslots.push_back({&dd.n, st.add(a->diag->n, kc1, true, true)});
    OBGC_DMS_DIAG_KC(X)
#undef X
  }
  // light hand-off: three planes in the caller's index space (+ two dense column-fastest ones for the reference layout)
  const size_t dense = (size_t)v.nlev * v.ncol * sizeof(double);
  const size_t scratch = 3 * align_up(kc1) + (v.kfast ? 2 * align_up(dense) : 0);
  int rc = arena_reserve(h, (st.host ? st.need : 0) + scratch);
  if (rc != OBGC_OK) return rc;
  rc = stage_in(h, st);
  if (rc != OBGC_OK) return rc;
  for (auto& s : dslots) *s.first = st.dev<double>(s.second);
  auto carve = [&](size_t bytes) -> double* {
    double* p = (double*)(h->arena + h->arena_used);
    h->arena_used += align_up(bytes);
    return p;
  };
  DmsKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  K.tracers = st.dev<const double>(i_tr); K.dz = st.dev<const double>(i_dz); K.kmax = st.dev<const int>(i_km);
  K.SST = st.dev<const double>(i_t); K.sw = st.dev<const double>(i_sw); K.tend = st.dev<double>(i_td);
  K.diag = dd;
  K.have_diag = a->diag ? 1 : 0;
  K.s_att = carve(kc1); K.s_kpardz = carve(kc1); K.s_parin = carve(kc1);
  if (v.kfast) { K.w_att = carve(dense); K.w_parin = carve(dense); }
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  launch_dms(K, h->dms, h->s_main);
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  rc = stage_out(h, st);
  if (rc != OBGC_OK) return rc;
  return finish(h, st.host);
}

extern "C" int obgc_dms_surface_fluxes(obgc_handle* h, const obgc_dms_surface_fluxes_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (!a->lcalc_DMS_gas_flux) return OBGC_OK;   // DMS_mod.F90:846
  if (!a->tracers || !a->surfacePressure || !a->windSpeedSquared10m || !a->SST || !a->iceFraction || !a->netFlux)
    return fail(h, OBGC_ERR_INVALID, "obgc_dms_surface_fluxes: null array");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0) return OBGC_OK;
  h->arena_used = 0;
  h->timed_valid = false;
  Stager st(h, a->grid.memspace == OBGC_MEM_HOST);
  const size_t cb = v.ncol * sizeof(double);
  const int surf_slots[1] = {h->dms.ind[0]};   // seaSurfaceDMS is the only tracer value read (DMS_mod.F90:834)
  SurfaceRows srows;
  int i_tr;
  if (st.host) { gather_surface_rows(st, v, a->tracers, OBGC_DMS_NTRACER, surf_slots, 1, srows); i_tr = srows.item; }
  else i_tr = st.add(a->tracers, span_kc(v, OBGC_DMS_NTRACER), true, false);
  const int i_p = st.add(a->surfacePressure, cb, true, false), i_w = st.add(a->windSpeedSquared10m, cb, true, false);
  const int i_t = st.add(a->SST, cb, true, false), i_ice = st.add(a->iceFraction, cb, true, true);
  const int i_ne = st.add(a->netFlux, span_c(v, OBGC_DMS_NTRACER), true, true);   // only two rows are written
  obgc_dms_flux_diag dd;
  std::memset(&dd, 0, sizeof dd);
  std::vector<std::pair<double**, int>> dslots;
  if (a->diag) {
#define X(n) if (a->diag->n) dslots.push_back({&dd.n, st.add(a->diag->n, cb, false, true)});
    OBGC_DMS_FLUX_DIAG(X)
#undef X
  }
  int rc = stage_in(h, st);
  if (rc != OBGC_OK) return rc;
  for (auto& s : dslots) *s.first = st.dev<double>(s.second);
  DmsSurfKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  if (st.host) {
    rc = upload_surface_rows(h, st, v, surf_slots, 1, srows);
    if (rc != OBGC_OK) return rc;
    K.g = surface_rows_view(v);
  }
  K.tracers = st.dev<const double>(i_tr); K.surfacePressure = st.dev<const double>(i_p); K.wind2 = st.dev<const double>(i_w);
  K.SST = st.dev<const double>(i_t); K.iceFraction = st.dev<double>(i_ice); K.net = st.dev<double>(i_ne);
  K.diag = dd;
  K.have_diag = a->diag ? 1 : 0;
  launch_dms_surface(K, h->dms, h->s_main);
  HIP_TRY(h, hipGetLastError());
  rc = stage_out(h, st);
  if (rc != OBGC_OK) return rc;
  return finish(h, st.host);
}

// ================================================================================ MACROS ==

extern "C" int obgc_macros_source_sink(obgc_handle* h, const obgc_macros_source_sink_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (!a->tracers || !a->kmax || !a->tend) return fail(h, OBGC_ERR_INVALID, "obgc_macros_source_sink: null array");
  if (int rc = no_tiled(h, a->grid, "obgc_macros_source_sink")) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0) return OBGC_OK;
  h->arena_used = 0;
  h->timed_valid = false;
  Stager st(h, a->grid.memspace == OBGC_MEM_HOST);
  const size_t kc1 = span_kc(v, 1);
  const bool gaps = st.host && ((v.kfast ? v.sc != v.nlev : v.sk != v.ncol) || v.plane != (v.kfast ? v.sc * v.ncol : v.sk * v.nlev));
  const int i_tr = st.add(a->tracers, span_kc(v, OBGC_MACROS_NTRACER), true, false);
  const int i_km = st.add(a->kmax, v.ncol * sizeof(int), true, false);
  const int i_td = st.add(a->tend, span_kc(v, OBGC_MACROS_NTRACER), gaps, true);
  obgc_macros_diag dd;
  std::memset(&dd, 0, sizeof dd);
  std::vector<std::pair<double**, int>> dslots;
  if (a->diag) {
#define X(n) if (a->diag->n) dslots.push_back({&dd.n, st.add(a->diag->n, kc1, true, true)});
    OBGC_MACROS_DIAG_KC(X)
#undef X
  }
  int rc = stage_in(h, st);
  if (rc != OBGC_OK) return rc;
  for (auto& s : dslots) *s.first = st.dev<double>(s.second);
  MacrosKernelArgs K;
  std::memset(&K, 0, sizeof K);
  K.g = v;
  K.tracers = st.dev<const double>(i_tr); K.kmax = st.dev<const int>(i_km); K.tend = st.dev<double>(i_td);
  K.diag = dd;
  K.have_diag = a->diag ? 1 : 0;
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  launch_macros(K, h->macros, h->s_main);
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  rc = stage_out(h, st);
  if (rc != OBGC_OK) return rc;
  return finish(h, st.host);
}

// ============================================================================ resident tracers ==

extern "C" int obgc_apply_tendencies(obgc_handle* h, const obgc_apply_tendencies_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  if (!grid_ok(a->grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (a->grid.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute; arrays may be NULL
  if (a->grid.memspace != OBGC_MEM_DEVICE)
    return fail(h, OBGC_ERR_UNSUPPORTED, "obgc_apply_tendencies: tracers must be resident on the device (OBGC_MEM_DEVICE)");
  if (a->ntracer < 0 || !a->tracers || !a->tend || !a->kmax || (a->net_flux && !a->dz))
    return fail(h, OBGC_ERR_INVALID, "obgc_apply_tendencies: null array");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(a->grid);
  if (v.ncol == 0 || a->ntracer == 0) return OBGC_OK;
  h->timed_valid = false;
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  launch_apply_tendencies(v, a->ntracer, a->tracers, a->tend, a->net_flux, a->dz, a->kmax, a->dt, a->clamp_nonneg, h->s_main);
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  return finish(h, false);
}

namespace {

// Shared tail of the *_accumulate entry points: `planes` are (k,column) arrays, `vecs` are (column) arrays, both lists of
// (instantaneous, mean) device pointers.  Tables of AccumTable::kMax pointers per launch.
struct AccumList {
  std::vector<const double*> psrc, vsrc;
  std::vector<double*> pdst, vdst;
  void plane(const double* s, double* d) { psrc.push_back(s); pdst.push_back(d); }
  void vec(const double* s, double* d) { vsrc.push_back(s); vdst.push_back(d); }
};

int accumulate_run(obgc_handle* h, const obgc_grid* grid, const AccumList& L, double weight, int reset, const char* who) {
  if (!grid_ok(*grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (grid->memspace != OBGC_MEM_DEVICE)
    return fail(h, OBGC_ERR_UNSUPPORTED, std::string(who) + ": diagnostics must be resident on the device (OBGC_MEM_DEVICE)");
  HIP_TRY(h, hipSetDevice(h->device));
  const GridView v = make_view(*grid);
  if (v.ncol == 0) return OBGC_OK;
  h->timed_valid = false;
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  size_t ip = 0, iv = 0;
  while (ip < L.psrc.size() || iv < L.vsrc.size()) {   // planes first, then vectors, kMax pointers per table
    AccumTable t;
    t.nplane = t.nvec = 0;
    while (ip < L.psrc.size() && t.nplane < AccumTable::kMax) { t.src[t.nplane] = L.psrc[ip]; t.dst[t.nplane] = L.pdst[ip]; ++t.nplane; ++ip; }
    while (iv < L.vsrc.size() && t.nplane + t.nvec < AccumTable::kMax) {
      t.src[t.nplane + t.nvec] = L.vsrc[iv]; t.dst[t.nplane + t.nvec] = L.vdst[iv]; ++t.nvec; ++iv;
    }
    launch_accumulate(v, t, weight, reset, h->s_main);
  }
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  return finish(h, false);
}

}  // namespace

extern "C" int obgc_bgc_diag_accumulate(obgc_handle* h, const obgc_grid* grid, const obgc_bgc_diag* inst, const obgc_bgc_diag* mean,
                                        double weight, int reset) {
  if (!h || !grid || !inst || !mean) return OBGC_ERR_INVALID;
  if (!grid_ok(*grid)) return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  const GridView v = make_view(*grid);
  AccumList L;
#define KC(n) if (inst->n && mean->n) L.plane(inst->n, mean->n);
#define KCA(n) if (inst->n && mean->n) for (int a = 0; a < OBGC_AUTOTROPH_CNT; ++a) L.plane(inst->n + a * v.plane, mean->n + a * v.plane);
#define CA(n) if (inst->n && mean->n) for (int a = 0; a < OBGC_AUTOTROPH_CNT; ++a) L.vec(inst->n + a * v.ldc, mean->n + a * v.ldc);
#define CC(n) if (inst->n && mean->n) L.vec(inst->n, mean->n);
  OBGC_BGC_DIAG_KC(KC) OBGC_BGC_DIAG_KCA(KCA) OBGC_BGC_DIAG_CA(CA) OBGC_BGC_DIAG_C(CC)
#undef KCA
#undef CA
#undef CC
  return accumulate_run(h, grid, L, weight, reset, "obgc_bgc_diag_accumulate");
}

extern "C" int obgc_dms_diag_accumulate(obgc_handle* h, const obgc_grid* grid, const obgc_dms_diag* inst, const obgc_dms_diag* mean,
                                        double weight, int reset) {
  if (!h || !grid || !inst || !mean) return OBGC_ERR_INVALID;
  AccumList L;
  OBGC_DMS_DIAG_KC(KC)
  return accumulate_run(h, grid, L, weight, reset, "obgc_dms_diag_accumulate");
}

extern "C" int obgc_macros_diag_accumulate(obgc_handle* h, const obgc_grid* grid, const obgc_macros_diag* inst,
                                           const obgc_macros_diag* mean, double weight, int reset) {
  if (!h || !grid || !inst || !mean) return OBGC_ERR_INVALID;
  AccumList L;
  OBGC_MACROS_DIAG_KC(KC)
#undef KC
  return accumulate_run(h, grid, L, weight, reset, "obgc_macros_diag_accumulate");
}

// Any other resident arrays (flux diagnostics, tendencies, model fields): explicit pointer lists.
extern "C" int obgc_accumulate_arrays(obgc_handle* h, const obgc_grid* grid, int nplanes, const double* const* inst_planes,
                                      double* const* mean_planes, int nvecs, const double* const* inst_vecs, double* const* mean_vecs,
                                      double weight, int reset) {
  if (!h || !grid || nplanes < 0 || nvecs < 0 || (nplanes > 0 && (!inst_planes || !mean_planes)) ||
      (nvecs > 0 && (!inst_vecs || !mean_vecs)))
    return OBGC_ERR_INVALID;
  AccumList L;
  for (int i = 0; i < nplanes; ++i) {
    if (!inst_planes[i] || !mean_planes[i]) return fail(h, OBGC_ERR_INVALID, "obgc_accumulate_arrays: null plane pointer");
    L.plane(inst_planes[i], mean_planes[i]);
  }
  for (int i = 0; i < nvecs; ++i) {
    if (!inst_vecs[i] || !mean_vecs[i]) return fail(h, OBGC_ERR_INVALID, "obgc_accumulate_arrays: null vector pointer");
    L.vec(inst_vecs[i], mean_vecs[i]);
  }
  return accumulate_run(h, grid, L, weight, reset, "obgc_accumulate_arrays");
}

// ================================================================================= synth ==

extern "C" int obgc_synth_fill(obgc_handle* h, const obgc_synth_args* a) {
  if (!h || !a || a->nlev < 1 || a->ncol < 0) return OBGC_ERR_INVALID;
  if (a->tiled && a->ld < 64) return fail(h, OBGC_ERR_INVALID, "obgc_synth_fill: the TILED layout needs its level stride ld");
  HIP_TRY(h, hipSetDevice(h->device));
  launch_synth(*a, a->ind, h->s_main);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}

extern "C" int obgc_synth_fill_extra(obgc_handle* h, const obgc_synth_extra_args* a) {
  if (!h || !a || a->nlev < 1 || a->ncol < 0) return OBGC_ERR_INVALID;
  if (a->tiled ? a->ld < 64 : (a->ld != 0 && a->ld < a->ncol)) return fail(h, OBGC_ERR_INVALID, "bad leading dimension");
  if (a->ldc != 0 && a->ldc < a->ncol) return fail(h, OBGC_ERR_INVALID, "bad leading dimension");
  HIP_TRY(h, hipSetDevice(h->device));
  launch_synth_extra(*a, h->s_main);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}
