/* libobgc -- C ABI of the MI355X-native ocean-BGC column engine.
 *
 * This is the drop-in boundary for the hot path of E3SM-Project/Ocean-BGC: every
 * compute entry point below replaces one public Fortran subroutine of the
 * reference (which has no FFI layer of its own -- its boundary is the Fortran
 * module API).  The thin Fortran modules in ocean-bgc_amd/fortran/
 * keep the reference's module and procedure names and forward to these functions
 * through ISO_C_BINDING.
 *
 *   reference subroutine (file:line)                      C entry point
 *   BGC_parms_init      BGC_parms.F90:497   + BGC_init BGC_mod.F90:184   obgc_bgc_params_init
 *   BGC_SourceSink      BGC_mod.F90:340-342                               obgc_bgc_source_sink
 *   BGC_SurfaceFluxes   BGC_mod.F90:2706-2708                             obgc_bgc_surface_fluxes
 *   DMS_parms_init      DMS_parms.F90:203   + DMS_init DMS_mod.F90:73     obgc_dms_params_init
 *   DMS_SourceSink      DMS_mod.F90:156-157                               obgc_dms_source_sink
 *   DMS_SurfaceFluxes   DMS_mod.F90:778-780                               obgc_dms_surface_fluxes
 *   MACROS_parms_init   MACROS_parms.F90:143 + MACROS_init MACROS_mod.F90:72  obgc_macros_params_init
 *   MACROS_SourceSink   MACROS_mod.F90:137-138                            obgc_macros_source_sink
 *
 * Conventions
 *  - plain C: pointers, sizes, POD structs; no exceptions cross the boundary.
 *  - every function returns 0 (OBGC_OK) or a negative obgc_status; the reference
 *    itself cannot fail (its aborts are commented out, co2calc.F90:931-933, 993-995),
 *    so a non-zero status always means an engine problem (no GPU, bad argument,
 *    out of device memory), never a numerical one.  There is NO CPU fallback.
 *  - indices inside parameter structs are 1-based, like the reference's *_ind.
 *  - a call with grid.ncol == 0 is a no-op that returns OBGC_OK (the reference's column loop does not
 *    execute); its array pointers are not examined.
 *  - arrays are fp64; integers are 32-bit (BGC_r8 / BGC_i4, BGC_parms.F90:21-24).
 *  - the caller owns every array; the engine owns only scratch inside the handle and
 *    keeps no pointer after a call returns.  State carried between calls lives in the
 *    caller's arrays exactly as in the reference (PH_PREV_3D, PH_PREV_ALT_CO2_3D,
 *    surface_pH, surface_pH_alt_co2; BGC_parms.F90:151-152, 170-171).
 *  - Stream ordering: the engine runs on its own non-blocking HIP streams.  With device
 *    memory the caller must make sure the producers of the input arrays have completed
 *    (e.g. hipDeviceSynchronize / an event wait) before a call, and must not touch the
 *    arrays until the call has returned (synchronous handles) or obgc_sync has (OBGC_ASYNC).
 *  - one handle per host thread per GPU.  Calls are synchronous unless the handle was
 *    created with OBGC_ASYNC (then obgc_sync completes them).
 */
#ifndef OBGC_H
#define OBGC_H

#include "obgc_fields.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OBGC_ABI_VERSION 1
#define OBGC_AUTOTROPH_CNT 4   /* BGC_parms.F90:42-43 */
#define OBGC_BGC_NTRACER 30    /* BGC_mod.F90:117-118 */
#define OBGC_DMS_NTRACER 14    /* DMS_mod.F90:61-62 */
#define OBGC_MACROS_NTRACER 8  /* MACROS_mod.F90 MACROS_tracer_cnt */

typedef enum {
  OBGC_OK = 0,
  OBGC_ERR_INVALID = -1,     /* null / inconsistent argument */
  OBGC_ERR_NO_DEVICE = -2,   /* no usable gfx950 device */
  OBGC_ERR_HIP = -3,         /* a HIP runtime call failed (see obgc_last_error) */
  OBGC_ERR_ALLOC = -4,       /* device or pinned-host allocation failed */
  OBGC_ERR_UNSUPPORTED = -5
} obgc_status;

/* Memory layout of level-dependent arrays.
 *  KFAST   = the reference's Fortran order X(k, column[, n]): k fastest, then column,
 *            then tracer/autotroph plane (BGC_parms.F90:128-132).
 *  COLFAST = the engine's native order X[n][k][column]: column fastest.  A host that
 *            keeps its tracers resident on the GPU should use this one: all kernels
 *            then read and write fully coalesced rows.
 *  TILED   = column blocks: the columns are cut into blocks of 64 and element (k, c) of plane n lives at
 *              base + (c / 64) * ld * nlev + k * ld + n * plane + c % 64
 *            where `ld` (>= 64) is the distance between consecutive levels of one block and `plane` (default 64)
 *            the distance between the planes of one array.  Arrays that share `ld` interleave row by row in ONE
 *            allocation: give each array the address of its first 64-element row inside a common
 *            arena[nblocks][nlev][ld] (nblocks = ceil(ncol / 64)); a block's rows of one level are then a
 *            single contiguous run of ld * 8 bytes and a workgroup of the column kernel streams through ONE region of
 *            memory instead of ~240 planes (+13 % sustained bandwidth for this access pattern on MI355X).  DEVICE
 *            memory only; accepted by obgc_bgc_source_sink, obgc_fused_step, the two *_surface_fluxes,
 *            obgc_apply_tendencies, the *_diag_accumulate family and obgc_synth_fill*; with BGC diagnostics the four
 *            production profiles photoC, photoC_NO3, CaCO3_form and bSi_form must be among them. */
typedef enum { OBGC_LAYOUT_KFAST = 0, OBGC_LAYOUT_COLFAST = 1, OBGC_LAYOUT_TILED = 2 } obgc_layout;
typedef enum { OBGC_MEM_HOST = 0, OBGC_MEM_DEVICE = 1 } obgc_memspace;

enum { OBGC_ASYNC = 1u };      /* obgc_create flag: device-memory calls return after enqueue */

/* Shape of one call.  For KFAST `ld` is the element distance between consecutive
 * columns (>= nlev; the reference's numLevelsMax) and `plane` the distance between
 * tracer / autotroph planes (>= ld*ncol; ld*numColumnsMax when the host array is
 * larger than the batch).  For COLFAST `ld` is the distance between consecutive
 * levels (>= ncol) and `plane` the distance between planes (>= ld*nlev).
 * For TILED (`ld` is required) see the layout enumeration above.
 * plane == 0 selects the dense default. (column) and (column,n) arrays are always
 * column-fastest with plane stride `ldc` (>= ncol; 0 = ncol). */
typedef struct {
  int nlev;        /* numLevelsMax */
  int ncol;        /* numColumns   */
  int layout;      /* obgc_layout  */
  int memspace;    /* obgc_memspace of EVERY array pointer of the call */
  long long ld;
  long long plane;
  long long ldc;
} obgc_grid;

/* ---------------------------------------------------------------- parameters */

/* autotroph_type, BGC_parms.F90:51-79 (character names are host-side only) */
typedef struct {
#define X(n) int n;
  OBGC_AUTOTROPH_INTS(X)
#undef X
#define X(n) double n;
  OBGC_AUTOTROPH_REALS(X)
#undef X
} obgc_autotroph;

typedef struct {
  obgc_autotroph autotrophs[OBGC_AUTOTROPH_CNT];
  int ind[OBGC_BGC_NTRACER];                 /* tracer slots, order OBGC_BGC_TRACERS */
  int sp_ind, diat_ind, diaz_ind, phaeo_ind; /* BGC_parms.F90:114-118 */
#define X(n) double n;
  OBGC_BGC_PARMS(X)
#undef X
  double parm_scalelen_z[4], parm_scalelen_vals[4]; /* BGC_parms.F90:363-365 */
  double T0_Kelvin_BGC;                      /* BGC_parms.F90:45 (host-set global) */
  int lrest_po4, lrest_no3, lrest_sio3;      /* BGC_mod.F90:131-134; must be 0 (never set in the reference) */
} obgc_bgc_params;

typedef struct {
  int ind[OBGC_DMS_NTRACER];                 /* order OBGC_DMS_TRACERS */
#define X(n) double n;
  OBGC_DMS_PARMS(X)
#undef X
} obgc_dms_params;

typedef struct {
  int ind[OBGC_MACROS_NTRACER];              /* order OBGC_MACROS_TRACERS */
#define X(n) double n;
  OBGC_MACROS_PARMS(X)
#undef X
} obgc_macros_params;

/* Defaults of BGC_parms_init (BGC_parms.F90:497-699) followed by BGC_init's
 * derivation of Chl_ind/C_ind/Fe_ind/Si_ind/CaCO3_ind (BGC_mod.F90:271-321).
 * `ind` = 30 one-based slots in OBGC_BGC_TRACERS order, or NULL for 1..30. */
void obgc_bgc_params_init(obgc_bgc_params *p, const int *ind, double T0_Kelvin_BGC);
/* re-derive the autotroph tracer slots after editing ind[] / kSiO3 / *_calcifier */
void obgc_bgc_params_derive(obgc_bgc_params *p);
void obgc_dms_params_init(obgc_dms_params *p, const int *ind);       /* DMS_parms.F90:203-241 */
void obgc_macros_params_init(obgc_macros_params *p, const int *ind); /* MACROS_parms.F90:143-162 */
/* tracer metadata of BGC_init / DMS_init / MACROS_init (BGC_mod.F90:221-328,
 * DMS_mod.F90:101-143, MACROS_mod.F90:72-129). which: 0 short_name, 1 long_name, 2 units;
 * module: 0 BGC, 1 DMS, 2 MACROS; slot is 1-based in the module's OBGC_*_TRACERS order.
 * Returns a static string, or NULL when out of range. */
const char *obgc_tracer_name(int module, int slot, int which);

/* ------------------------------------------------------------------- handle */

typedef struct obgc_handle obgc_handle;

int obgc_abi_version(void);
int obgc_sizeof_bgc_params(void);
int obgc_sizeof_dms_params(void);
int obgc_sizeof_macros_params(void);
int obgc_sizeof_bgc_diag(void);
int obgc_create(obgc_handle **h, int device_id, unsigned flags);
int obgc_destroy(obgc_handle *h);
int obgc_sync(obgc_handle *h);
const char *obgc_last_error(const obgc_handle *h);
/* upload parameters (the reference's mutable module variables); call again whenever
 * the host changes one.  obgc_create installs the defaults. */
int obgc_set_bgc_params(obgc_handle *h, const obgc_bgc_params *p);
int obgc_set_dms_params(obgc_handle *h, const obgc_dms_params *p);
int obgc_set_macros_params(obgc_handle *h, const obgc_macros_params *p);
/* HIP stream (hipStream_t) the engine enqueues its last kernel of a call on */
void *obgc_stream(obgc_handle *h);
/* device time of the kernels of the last *_source_sink / *_surface_fluxes call in ms
 * (HIP events on the engine's own streams); `which`: 0 whole call, 1 carbonate
 * kernel, 2 the ecosystem kernels (PAR scan + cell kernel + particle sweep), 3 layout transposes,
 * 4 the ecosystem cell kernel alone */
double obgc_last_kernel_ms(obgc_handle *h, int which);
/* how the last obgc_bgc_source_sink (or obgc_fused_step) arranged its kernels: 0 the carbonate kernel runs from the
 * start beside the ecosystem kernels, 1 it is released when the cell kernel ends, 2 the carbonate system is solved
 * inside the cell kernel's wavefronts (one launch for both); 3 the column-block kernel (light, cell terms, particle
 * recurrence AND the carbonate system in one launch: the native layout's default above 10 M cells with diagnostics),
 * 4 the column-block kernel with the carbonate kernel beside it (the native layout's default otherwise, from 700 000
 * cells per call; smaller native-layout calls take the separate kernels, schedules 0-2, which spread a small batch over
 * the whole device); -1 before the first call.  Every arrangement gives the
 * same bits (measurement harnesses need to know which launch `which = 4` of obgc_last_kernel_ms timed).
 * Environment: OBGC_COLUMN_KERNEL=0 keeps the separate attenuation / scan / cell / sweep kernels (schedules 0-2) at every
 * size, =2 the column-block kernel at every size; OBGC_COLUMN_MIN_CELLS=<n> moves the 700 000; OBGC_SCHED=<n> forces an
 * arrangement. */
int obgc_last_schedule(obgc_handle *h);

/* --------------------------------------------------------- BGC_SourceSink */

/* BGC_diagnostics_type (BGC_parms.F90:192-321) as a struct of pointers.  NULL members
 * are skipped.  (k,column,autotroph) arrays use grid.plane between autotroph planes;
 * (column,autotroph) arrays use grid.ldc.  POC_ACCUM, DONr_remin, DOPr_remin are never
 * written (the reference never touches them either). */
typedef struct {
#define X(n) double *n;
  OBGC_BGC_DIAG_KC(X)
  OBGC_BGC_DIAG_KCA(X)
  OBGC_BGC_DIAG_CA(X)
  OBGC_BGC_DIAG_C(X)
#undef X
} obgc_bgc_diag;

typedef struct {
  obgc_grid grid;
  /* BGC_input_type, BGC_parms.F90:127-137 */
  const double *tracers;      /* BGC_tracers(k,column,30) */
  const double *temp, *salt;  /* PotentialTemperature, Salinity (k,column) */
  const double *zmid, *dz, *zbot; /* cell_center_depth, cell_thickness, cell_bottom_depth [cm] */
  const double *lat;          /* cell_latitude(column) */
  const int *kmax;            /* number_of_active_levels(column) */
  /* BGC_forcing_type, BGC_parms.F90:139-145 */
  const double *fesedflux;    /* FESEDFLUX(k,column) */
  const double *dust_flux_in; /* dust_FLUX_IN(column) */
  const double *sw_flux;      /* ShortWaveFlux_surface(column) */
  /* BGC_output_type, BGC_parms.F90:167-172 */
  double *tend;               /* BGC_tendencies(k,column,30), fully overwritten */
  double *ph_prev;            /* PH_PREV_3D(k,column), inout, 0 = no previous solution */
  double *ph_prev_alt;        /* PH_PREV_ALT_CO2_3D(k,column), inout */
  const obgc_bgc_diag *diag;  /* NULL = no diagnostics */
  int alt_co2_use_eco;        /* BGC_mod.F90:1741-1745 */
} obgc_bgc_source_sink_args;

int obgc_bgc_source_sink(obgc_handle *h, const obgc_bgc_source_sink_args *a);

/* ------------------------------------------------------ BGC_SurfaceFluxes */

typedef struct {
#define X(n) double *n;
  OBGC_BGC_FLUX_DIAG(X)
#undef X
} obgc_bgc_flux_diag;

typedef struct {
  obgc_grid grid;             /* shape of `tracers`; only level 1 is read (BGC_mod.F90:2816-2822) */
  const double *tracers;
  /* BGC_forcing_type (column) members, BGC_parms.F90:143-155 */
  const double *surfacePressure, *windSpeedSquared10m, *atmCO2, *atmCO2_ALT_CO2, *surfaceDepth, *SST, *SSS;
  double *iceFraction;        /* clamped to [0,1] IN PLACE (BGC_mod.F90:2837-2838) */
  double *surface_pH, *surface_pH_alt_co2; /* inout */
  /* (column,30) arrays, plane stride grid.ldc; the Fe rows of the first four are scaled
   * by parm_Fe_bioavail IN PLACE on every call (BGC_mod.F90:2828-2835) */
  double *depositionFlux, *riverFlux, *gasFlux, *seaIceFlux;
  double *netFlux;            /* out */
  const obgc_bgc_flux_diag *diag; /* NULL = none; the reference zero-fills all 14 first */
  int lcalc_O2_gas_flux, lcalc_CO2_gas_flux; /* BGC_parms.F90:162-164 */
} obgc_bgc_surface_fluxes_args;

int obgc_bgc_surface_fluxes(obgc_handle *h, const obgc_bgc_surface_fluxes_args *a);

/* -------------------------------------------------------------------- DMS */

typedef struct {
#define X(n) double *n;
  OBGC_DMS_DIAG_KC(X)
#undef X
} obgc_dms_diag;

typedef struct {
  obgc_grid grid;
  const double *tracers;      /* DMS_tracers(k,column,14), DMS_parms.F90:85-87 */
  const double *dz;           /* cell_thickness(k,column) */
  const int *kmax;
  const double *SST, *sw_flux; /* DMS_forcing_type (column), DMS_parms.F90:95-101 */
  double *tend;               /* DMS_tendencies(k,column,14) */
  const obgc_dms_diag *diag;  /* NULL = none.  NOT zero-filled (the reference does not either) */
} obgc_dms_source_sink_args;

int obgc_dms_source_sink(obgc_handle *h, const obgc_dms_source_sink_args *a);

/* obgc_dms_source_sink for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code under HIP graph capture -- which additionally needs a handle created with OBGC_ASYNC, since a
 * synchronous handle ends every call with a stream synchronisation, and captures obgc_stream(h)).  NO counterpart in
 * the reference.  Same arguments, same kernels and the same
 * bits as obgc_dms_source_sink, but the three-plane hand-off of the light pass lives in a fixed device array of the
 * library instead of the handle's growing arena, so the call performs no hipMalloc / hipFree:
 *   - DEVICE pointers in the native COLFAST layout only (anything else: OBGC_ERR_UNSUPPORTED);
 *   - at most OBGC_NOALLOC_MAX_CELLS (k,column) elements per plane, padding included (else OBGC_ERR_UNSUPPORTED);
 *   - the array is per GPU and shared by every handle on it: calls that overlap in time on one GPU race. */
#define OBGC_NOALLOC_MAX_CELLS (1LL << 20)
long long obgc_noalloc_max_cells(void);
int obgc_dms_source_sink_noalloc(obgc_handle *h, const obgc_dms_source_sink_args *a);

typedef struct {
#define X(n) double *n;
  OBGC_DMS_FLUX_DIAG(X)
#undef X
} obgc_dms_flux_diag;

typedef struct {
  obgc_grid grid;
  const double *tracers;
  const double *surfacePressure, *windSpeedSquared10m, *SST, *SSS;
  double *iceFraction;        /* clamped in place, DMS_mod.F90:858-859 */
  double *netFlux;            /* (column,14): only the dms and dmsp rows are written */
  const obgc_dms_flux_diag *diag;
  int lcalc_DMS_gas_flux;     /* 0: the call is a no-op, DMS_mod.F90:846 */
} obgc_dms_surface_fluxes_args;

int obgc_dms_surface_fluxes(obgc_handle *h, const obgc_dms_surface_fluxes_args *a);

/* ----------------------------------------------------------------- MACROS */

typedef struct {
#define X(n) double *n;
  OBGC_MACROS_DIAG_KC(X)
#undef X
} obgc_macros_diag;

typedef struct {
  obgc_grid grid;
  const double *tracers;      /* MACROS_tracers(k,column,8) */
  const int *kmax;
  double *tend;               /* MACROS_tendencies(k,column,8) */
  const obgc_macros_diag *diag;
} obgc_macros_source_sink_args;

int obgc_macros_source_sink(obgc_handle *h, const obgc_macros_source_sink_args *a);

/* ------------------------------------------------------------- fused step
 * BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink over one batch in ONE call: what a host does that runs all three
 * packages on the same columns every step (BGC_mod.F90:340, DMS_mod.F90:156, MACROS_mod.F90:137).  DMS reads twelve
 * tracers it shares with BGC (no3, doc, zooC, spC, spCaCO3, diatC, diazC, phaeoC and the four Chl; DMS_mod.F90:471-482),
 * MACROS five (zooC and the four C pools; MACROS_mod.F90:301-325): the host fills those slots of DMS_tracers /
 * MACROS_tracers with copies of the BGC tracers.  Here they are READ ONCE from bgc.tracers, the light attenuation that
 * BGC (BGC_mod.F90:907-924) and DMS (DMS_mod.F90:538-551) both compute from the same chlorophyll is computed once, and
 * the cell terms of all three run on one set of loads.  Only each package's OWN tracers, their tendencies and the
 * diagnostics are separate (k,column) planes, all in bgc.grid's index space.
 *   - dz, kmax and ShortWaveFlux_surface of DMS / MACROS are bgc.dz, bgc.kmax, bgc.sw_flux.
 *   - results are bit-identical to the three separate calls on DMS_tracers / MACROS_tracers filled that way; the twelve /
 *     five tendency planes of the shared tracers, which the reference leaves zero, are not written.
 *   - cells below kmax: the own tendencies are zero-filled, DMS / MACROS diagnostics are left alone (as the reference).
 *   - dms == NULL skips DMS, prot == NULL skips MACROS; both NULL is obgc_bgc_source_sink.
 *   - DEVICE pointers in the native COLFAST layout or in the TILED layout only (OBGC_ERR_UNSUPPORTED otherwise). */
typedef struct {
  obgc_bgc_source_sink_args bgc;
  const double *dms, *dmsp;               /* DMS_tracers(k,column,dms_ind / dmsp_ind) */
  const double *SST;                      /* DMS_forcing%SST(column) */
  double *dms_tend, *dmsp_tend;           /* the two non-zero planes of DMS_tendencies */
  const obgc_dms_diag *dms_diag;          /* NULL = none */
  const double *prot, *poly, *lip;        /* MACROS_tracers(k,column,prot_ind / poly_ind / lip_ind) */
  double *prot_tend, *poly_tend, *lip_tend;
  const obgc_macros_diag *macros_diag;    /* NULL = none */
} obgc_fused_step_args;

int obgc_fused_step(obgc_handle *h, const obgc_fused_step_args *a);

/* ------------------------------------------- resident tracers
 * NO reference counterpart: the reference returns tendencies and the host model owns the time
 * integration.  For a host that keeps its tracers on the GPU between steps this is the step right
 * after the path -- an explicit update that avoids moving 60 arrays over PCIe every step:
 *
 *   x(k,c,n) <- x(k,c,n) + dt * ( tend(k,c,n) + [k == 1] * net_flux(c,n) / dz(1,c) )      k <= kmax(c)
 *   x        <- max(x, 0)                                                  if clamp_nonneg
 *
 * Cells below kmax(c) are left untouched.  DEVICE pointers only (grid.memspace = OBGC_MEM_DEVICE);
 * both layouts.  net_flux is the (column, n) array *_SurfaceFluxes leaves in netFlux (NULL = none),
 * dz the cell thickness in the same length unit as the flux (cm for this package). */
typedef struct {
  obgc_grid grid;
  int ntracer;
  double *tracers;            /* (k,column,ntracer), updated in place */
  const double *tend;         /* (k,column,ntracer) */
  const double *net_flux;     /* (column,ntracer) with leading dimension grid.ldc, or NULL */
  const double *dz;           /* (k,column) */
  const int *kmax;
  double dt;
  int clamp_nonneg;
} obgc_apply_tendencies_args;

int obgc_apply_tendencies(obgc_handle *h, const obgc_apply_tendencies_args *a);

/* ------------------------------ a mesh block resident on the device
 * The device-side twin of ocean-bgc_amd/fortran/obgc_mpas_adapter.F90: a block of an unstructured-mesh ocean model
 * (tracer-fastest arrays in metres, one column per cell) whose BGC state -- native-layout tracers, tendencies,
 * PH_PREV_*, surface_pH* -- lives on the GPU.  Per step only the model's own arrays cross PCIe (tracers, T, S,
 * thickness up; thickness-weighted tendencies and surface fluxes down: 3.3 GB instead of 10 GB at 100k x 60), the
 * layout change and the unit conversions (m -> cm, rad -> deg, Pa -> atm, ...) run on the device, and the path itself
 * runs in its native column-fastest layout.  Same arithmetic as the Fortran adapter, so both give the same bits.
 * obgc_block_step takes HOST pointers in the model's layout, obgc_block_step_device DEVICE pointers in the same layout. */
typedef struct obgc_block obgc_block;
int obgc_block_create(obgc_handle *h, int nCells, int nVertLevels, obgc_block **out);
int obgc_block_destroy(obgc_block *b);

typedef struct {
  const double *ecosysTracers;       /* (30, nVertLevels, nCells)  mmol/m^3, tracer index fastest, slots as obgc_bgc_params.ind */
  const double *layerThickness;      /* (nVertLevels, nCells)      m */
  const double *zMid;                /* (nVertLevels, nCells)      m, negative downwards */
  const double *temperature, *salinity;   /* (nVertLevels, nCells) */
  const int *maxLevelCell;           /* (nCells) */
  const double *latCell;             /* (nCells) rad */
  const double *shortWaveHeatFlux;   /* (nCells) W/m^2 */
  const double *atmosphericPressure; /* (nCells) Pa */
  const double *iceFraction;         /* (nCells) */
  const double *windSpeedSquared10m; /* (nCells) m^2/s^2 */
  const double *atmosphericCO2, *atmosphericCO2_ALT_CO2;   /* (nCells) ppm */
  const double *dustFlux;            /* (nCells) kg/m^2/s */
  const double *ironSedimentFlux;    /* (nVertLevels, nCells) package units */
  const double *depositionFlux, *riverFlux, *seaIceFlux;   /* (nCells, 30) package units */
  double *ecosysTracersTend;         /* (30, nVertLevels, nCells): += layerThickness * tendency (accumulate != 0) or = (0) */
  double *ecosysSurfaceFlux;         /* (30, nCells) out: netFlux * 0.01 */
  int accumulate;
  /* Optional (NULL = not run): DMS and / or MACROS on the same cells in the same step.  The tracers the two modules share
   * with BGC (DMS_mod.F90:471-482, MACROS_mod.F90:301-325) are taken from ecosysTracers; only their own tracers travel.
   * The source-sink part then runs as ONE obgc_fused_step, and DMS_SurfaceFluxes follows BGC_SurfaceFluxes. */
  const double *dmsTracers;          /* (2, nVertLevels, nCells): dms, dmsp, tracer index fastest */
  double *dmsTracersTend;            /* (2, nVertLevels, nCells): += or = layerThickness * tendency, as ecosysTracersTend */
  double *dmsSurfaceFlux;            /* (2, nCells) out: DMS_SurfaceFluxes netFlux(dms, dmsp) * 0.01 */
  const double *macrosTracers;       /* (3, nVertLevels, nCells): prot, poly, lip */
  double *macrosTracersTend;         /* (3, nVertLevels, nCells) */
} obgc_block_step_args;

/* one BGC_SurfaceFluxes + BGC_SourceSink over the whole block (tendencies only; PH_PREV_* and surface_pH* persist
 * inside the block); with dmsTracers / macrosTracers also DMS_SurfaceFluxes and the fused BGC + DMS + MACROS step */
int obgc_block_step(obgc_block *b, const obgc_block_step_args *a);

/* The state a block carries between steps -- PH_PREV_3D, PH_PREV_ALT_CO2_3D (BGC_mod.F90:935-1001) and surface_pH,
 * surface_pH_alt_co2 (BGC_mod.F90:2816-2900), which the reference keeps in the host's arrays -- for the model's restart
 * file: HOST arrays in the model's order, (nVertLevels, nCells) level-fastest and (nCells), the arguments of
 * obgc_restart_write / obgc_restart_read.  A fresh block starts from zeros (no previous pH: the wide first bracket).
 * Both calls wait for the block's pending steps. */
int obgc_block_get_state(obgc_block *b, double *ph_prev, double *ph_prev_alt, double *surface_pH, double *surface_pH_alt);
int obgc_block_set_state(obgc_block *b, const double *ph_prev, const double *ph_prev_alt, const double *surface_pH,
                         const double *surface_pH_alt);

/* Diagnostics of the block's steps.  The block keeps its state in the native layout, so diagnostics are computed into
 * DEVICE arrays of that index space: COLFAST, nlev = nVertLevels, ncol = ld = ldc = nCells -- what an obgc_diag_set created
 * for such a grid owns (pass obgc_diag_set_bgc / _dms / _macros of the set; the host then accumulates, downloads or writes
 * through the set, in the reference's (k,column) order = the model's (nVertLevels, nCells)).  Every later step of the
 * block fills the non-NULL members; a NULL struct = none of that module (the default).  The structs are copied, the arrays
 * they point to must outlive the steps.  DMS / MACROS members are filled by steps that are given dmsTracers / macrosTracers. */
int obgc_block_set_diagnostics(obgc_block *b, const obgc_bgc_diag *bgc, const obgc_dms_diag *dms, const obgc_macros_diag *macros);

/* The same step for a model whose arrays already live on the handle's GPU (an OpenACC / OpenMP-offloaded or HIP ocean
 * model): every pointer of obgc_block_step_args is a DEVICE pointer, in the same tracer-fastest layout and units.  The
 * layout change reads the model's arrays and writes (or adds to) its tendency and flux arrays directly: nothing is staged,
 * nothing crosses PCIe, and a block that is only ever stepped this way reserves no staging memory.  The model's arrays
 * are not modified apart from the outputs (the package's in-place Fe scaling of the flux arrays works on copies).
 * Same kernels and arithmetic as obgc_block_step: same bits.  Runs on the handle's stream; returns after enqueue on a
 * handle created with OBGC_ASYNC (obgc_sync completes it), after completion otherwise.  The two forms may be mixed on
 * one block.  Host arrays are refused with OBGC_ERR_INVALID (the tracer and tendency arrays are looked up with
 * hipPointerGetAttributes) instead of being handed to a kernel. */
int obgc_block_step_device(obgc_block *b, const obgc_block_step_args *a);

/* --------------------------------------- diagnostics kept on the device
 * NO reference counterpart (the host model averages and writes diagnostics).  Diagnostics are 65 % of the bytes
 * the path produces; a host that only needs time means at its output interval keeps both the instantaneous
 * arrays and the running means on the GPU and downloads the means it selected when the interval ends:
 *
 *   mean.X <- (reset ? 0 : mean.X) + weight * inst.X        for every member X non-NULL in BOTH structs
 *
 * over all nlev x ncol cells of (k,column[,autotroph]) members (the package zero-fills below kmax) and over ncol
 * for (column[,autotroph]) members; padding of either layout is not touched.  weight = dt / interval.
 * One launch for all selected members.  DEVICE pointers only; both layouts. */
int obgc_bgc_diag_accumulate(obgc_handle *h, const obgc_grid *grid, const obgc_bgc_diag *inst,
                             const obgc_bgc_diag *mean, double weight, int reset);
/* the same for the 27 DMS and the 6 MACROS diagnostics (DMS_parms.F90:125-154, MACROS_parms.F90:105-113) ... */
int obgc_dms_diag_accumulate(obgc_handle *h, const obgc_grid *grid, const obgc_dms_diag *inst,
                             const obgc_dms_diag *mean, double weight, int reset);
int obgc_macros_diag_accumulate(obgc_handle *h, const obgc_grid *grid, const obgc_macros_diag *inst,
                                const obgc_macros_diag *mean, double weight, int reset);
/* ... and for any other resident arrays (flux diagnostics, tendencies, the host's own fields): HOST arrays of DEVICE
 * pointers, `nplanes` arrays of (k,column) in grid's layout and `nvecs` arrays of (column); every pointer non-NULL. */
int obgc_accumulate_arrays(obgc_handle *h, const obgc_grid *grid, int nplanes, const double *const *inst_planes,
                           double *const *mean_planes, int nvecs, const double *const *inst_vecs,
                           double *const *mean_vecs, double weight, int reset);

/* --------------------------------- diagnostics selected by name
 * NO reference counterpart.  The reference's host model reads a namelist of diagnostic names, averages those members over
 * its output interval and writes them.  An obgc_diag_set does that next to the fields: it owns the instantaneous arrays and
 * the running means of the SELECTED members only (names = the reference's members without the diag_ prefix, optionally
 * qualified "bgc.", "dms." or "macros."; BGC_parms.F90:192-321, DMS_parms.F90:125-154, MACROS_parms.F90:105-113), gives the
 * compute entry points pointer structs in which every other member is NULL -- so nothing else is written or even
 * zero-filled -- and returns or writes the selected fields in the reference's X(k,column[,autotroph]) order.
 *   - grid: DEVICE memory, COLFAST or TILED; the index space of the calls the structs are passed to.
 *   - COLFAST: the set allocates everything (inst_rows = mean_rows = NULL).  TILED: the (k,column[,a]) members are rows of the
 *     caller's arena -- obgc_diag_set_rows says how many -- starting at inst_rows / mean_rows (consecutive rows, in the order
 *     of `names`, four for an autotroph-dimensioned member); (column[,a]) members are always the set's own.
 *   - obgc_diag_set_bgc / _dms / _macros: the struct to pass as `diag` (NULL when no member of that module is selected).
 *   - obgc_diag_set_accumulate: mean <- (reset ? 0 : mean) + weight * inst for every selected member (one launch a module).
 *   - obgc_diag_set_download: one member (which = 0 the mean, 1 the instantaneous field) into a HOST array of
 *     obgc_diag_set_elems values; obgc_diag_set_write: all of them into a file (format: csrc/obgc_io.cpp, "OBGCDIA1"),
 *     read back by the host-only obgc_diag_file_* helpers.
 *   - obgc_diag_names_read_text: the names of a parameter file's [diagnostics] section (`bgc = CO3, pH_3D` ...), as
 *     `*count` NUL-terminated qualified names packed into `names` (a final empty string ends the list). */
typedef struct obgc_diag_set obgc_diag_set;
int obgc_diag_set_rows(int nnames, const char *const *names, int *rows, int *vectors, char *err, int errlen);
int obgc_diag_set_create(obgc_handle *h, const obgc_grid *grid, int nnames, const char *const *names, double *inst_rows,
                         double *mean_rows, obgc_diag_set **out);
int obgc_diag_set_destroy(obgc_diag_set *s);
int obgc_diag_set_count(const obgc_diag_set *s);
const obgc_bgc_diag *obgc_diag_set_bgc(const obgc_diag_set *s);
const obgc_dms_diag *obgc_diag_set_dms(const obgc_diag_set *s);
const obgc_macros_diag *obgc_diag_set_macros(const obgc_diag_set *s);
int obgc_diag_set_accumulate(obgc_diag_set *s, double weight, int reset);
long long obgc_diag_set_elems(const obgc_diag_set *s, const char *name);
int obgc_diag_set_download(obgc_diag_set *s, const char *name, int which, double *host, long long host_len);
int obgc_diag_set_write(obgc_diag_set *s, const char *path, int which);
int obgc_diag_names_read_text(const char *path, char *names, int nameslen, int *count, char *err, int errlen);
int obgc_diag_file_info(const char *path, int *nlev, int *ncol, int *nrec, int *which);
int obgc_diag_file_record(const char *path, int index, char *name, int namelen, int *module, int *kind, long long *nelem);
int obgc_diag_file_read(const char *path, const char *name, double *out, long long len);

/* ------------------------------- parameter files and restart state
 * Host-only helpers (no GPU needed).  The reference has no file format: its parameters are module variables.
 * Text format: sections [bgc], [autotroph N], [dms], [macros] (and [diagnostics], see above: checked, otherwise ignored
 * here); `name = value` with the reference's names; reals
 * are written as C99 hex floats (exact round trip), read as decimal or hex, always in the "C" locale.  Reading only
 * overrides the listed keys, so initialise the structs first (obgc_*_params_init); unknown keys are errors (message
 * in `err`).  Any of the three struct pointers may be NULL (section skipped on write, parsed but ignored on read).
 * A successful read ends with obgc_bgc_params_derive: the per-autotroph Chl/C/Fe/Si/CaCO3 slots always follow from
 * kSiO3 / *_calcifier / ind[] as read, whatever *_ind lines the file carries. */
int obgc_params_write_text(const char *path, const obgc_bgc_params *bgc, const obgc_dms_params *dms,
                           const obgc_macros_params *macros);
int obgc_params_read_text(const char *path, obgc_bgc_params *bgc, obgc_dms_params *dms, obgc_macros_params *macros,
                          char *err, int errlen);
/* Restart file of the only state carried between calls: PH_PREV_3D, PH_PREV_ALT_CO2_3D (nlev,ncol) and
 * surface_pH, surface_pH_alt_co2 (ncol), HOST arrays in the reference's order; 64-byte header "OBGCRST1" whose
 * order word (0 = level-fastest, the only one defined) is checked on read. */
int obgc_restart_write(const char *path, int nlev, int ncol, const double *ph_prev, const double *ph_prev_alt,
                       const double *surface_pH, const double *surface_pH_alt);
int obgc_restart_read_header(const char *path, int *nlev, int *ncol);
int obgc_restart_read(const char *path, int nlev, int ncol, double *ph_prev, double *ph_prev_alt, double *surface_pH,
                      double *surface_pH_alt);
/* The same file for HOST arrays described by a grid struct (either layout, leading dimension grid->ld; memspace must be
 * OBGC_MEM_HOST): a caller that keeps PH_PREV_* column-fastest -- the engine's native layout -- needs no transpose of
 * its own.  The file is the same level-fastest file either way. */
int obgc_restart_write_grid(const char *path, const obgc_grid *grid, const double *ph_prev, const double *ph_prev_alt,
                            const double *surface_pH, const double *surface_pH_alt);
int obgc_restart_read_grid(const char *path, const obgc_grid *grid, double *ph_prev, double *ph_prev_alt,
                           double *surface_pH, double *surface_pH_alt);

/* ------------------------------------------------ column ranges for several devices
 * No statement of the path reads another column (BGC_mod.F90:808-814 resets the only carried scalars), so N
 * devices take N contiguous column ranges and exchange nothing.  Host-only helper: bounds[0..nparts] with
 * bounds[0] = 0 <= ... <= bounds[nparts] = ncol; device p owns [bounds[p], bounds[p+1]).
 *   balance = 0  equal column counts
 *   balance = 1  equal numbers of ACTIVE cells, sum of max(kmax, 0): the cost of a range under ragged bathymetry
 *                (kmax = number_of_active_levels, HOST pointer; may be NULL for balance = 0) */
int obgc_partition_columns(const int *kmax, long long ncol, int nparts, int balance, long long *bounds);

/* ----------------------------------------------- workload utilities (bench/tests)
 * Deterministic synthetic columns, generated on the device in the
 * engine's native COLFAST layout for global columns [col0, col0+ncol).  Bit-identical
 * to ocean-bgc_amd/synth.py.  All pointers are DEVICE pointers; NULL skips a field.
 * `tracers` receives 30 planes placed by `ind` (1-based, OBGC_BGC_TRACERS order). */
typedef struct {
  int nlev, ncol;
  long long col0;
  long long ld, plane, ldc;
  const int *ind;             /* HOST pointer to 30 ints, or NULL for 1..30 */
  double *tracers, *temp, *salt, *zmid, *dz, *zbot, *fesedflux;
  double *lat, *dust_flux_in, *sw_flux;
  int *kmax;
  int tiled;                  /* 0: COLFAST strides (ld, plane); 1: OBGC_LAYOUT_TILED with the same two numbers */
} obgc_synth_args;
int obgc_synth_fill(obgc_handle *h, const obgc_synth_args *a);
/* The rest of the same workload: the (column) forcing of BGC_SurfaceFluxes / DMS_SurfaceFluxes
 * (BGC_parms.F90:143-155), the four (column,30) external flux arrays (plane stride ldc) and the own
 * tracers of DMS (DMS_parms.F90:85-87: dms, dmsp) and MACROS (prot, poly, lip) as (k,column) planes
 * with level stride ld.  DEVICE pointers; NULL skips a field.  Bit-identical to synth.py. */
typedef struct {
  int nlev, ncol;
  long long col0;
  long long ld, ldc;
  double *surfacePressure, *iceFraction, *windSpeedSquared10m, *atmCO2, *atmCO2_ALT_CO2, *surfaceDepth, *SST, *SSS;
  double *surface_pH, *surface_pH_alt_co2;            /* zero: no previous solution */
  double *depositionFlux, *riverFlux, *gasFlux, *seaIceFlux;
  double *dms, *dmsp, *prot, *poly, *lip;
  int tiled;                  /* as in obgc_synth_args, for the five (k,column) planes */
} obgc_synth_extra_args;
int obgc_synth_fill_extra(obgc_handle *h, const obgc_synth_extra_args *a);

#ifdef __cplusplus
}
#endif
#endif
