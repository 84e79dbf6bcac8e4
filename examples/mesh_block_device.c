/* A C "ocean model" whose arrays live on the GPU in the model's own conventions -- tracer index fastest, metres, radians,
 * Pa, one column per cell -- stepping its biogeochemistry through the resident mesh block of include/obgc.h:
 *
 *   obgc_block_create -> obgc_diag_set_create + obgc_block_set_diagnostics -> 3 x ( obgc_block_step_device,
 *   obgc_diag_set_accumulate ) -> tendencies, surface fluxes, the mean of a selected diagnostic, the state for a restart
 *
 * No Python, no PyTorch, nothing but the C ABI and the HIP runtime for device memory.  The model's fields are made from the
 * library's synthetic columns (generated on the device in the package's units, converted on the host the way
 * tests/test_mpas_adapter.py does it).  With a third argument the results are written as raw doubles;
 * tests/test_c_example.py reproduces them bit for bit through the Python host on host arrays.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "obgc.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, h ? obgc_last_error(h) : ""); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static double* dalloc(size_t n) {
  void* p = NULL;
  return hipMalloc(&p, n * sizeof(double)) == hipSuccess ? (double*)p : NULL;
}
static double* halloc(size_t n) { return (double*)malloc(n * sizeof(double)); }
/* device array -> fresh host array */
static double* fetch(const double* dev, size_t n) {
  double* p = halloc(n);
  if (p && hipMemcpy(p, dev, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { free(p); p = NULL; }
  return p;
}
/* host array -> fresh device array ("the model's") */
static double* push(const double* host, size_t n) {
  double* p = dalloc(n);
  if (p && hipMemcpy(p, host, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) p = NULL;
  return p;
}

int main(int argc, char** argv) {
  const int nc = argc > 1 ? atoi(argv[1]) : 4096, nz = argc > 2 ? atoi(argv[2]) : 60, nsteps = 3;
  const size_t kc = (size_t)nc * nz, c30 = 30 * (size_t)nc;
  obgc_handle* h = NULL;
  CHECK(obgc_create(&h, 0, 0));

  /* ---- 1. synthetic columns in the package's layout and units, on the device, then on the host ---- */
  double *p_tr = dalloc(30 * kc), *p_temp = dalloc(kc), *p_salt = dalloc(kc), *p_zmid = dalloc(kc), *p_dz = dalloc(kc), *p_zbot = dalloc(kc),
         *p_fesed = dalloc(kc), *p_lat = dalloc(nc), *p_dust = dalloc(nc), *p_sw = dalloc(nc);
  double *p_patm = dalloc(nc), *p_ice = dalloc(nc), *p_wind = dalloc(nc), *p_co2 = dalloc(nc), *p_co2alt = dalloc(nc);
  double *p_depo = dalloc(c30), *p_river = dalloc(c30), *p_seaice = dalloc(c30);
  int* p_kmax = NULL;
  HIP(hipMalloc((void**)&p_kmax, nc * sizeof(int)));
  if (!p_tr || !p_seaice) { fprintf(stderr, "device allocation failed\n"); return 1; }
  obgc_synth_args sy;
  memset(&sy, 0, sizeof sy);
  sy.nlev = nz; sy.ncol = nc; sy.col0 = 0;
  sy.tracers = p_tr; sy.temp = p_temp; sy.salt = p_salt; sy.zmid = p_zmid; sy.dz = p_dz; sy.zbot = p_zbot; sy.fesedflux = p_fesed;
  sy.lat = p_lat; sy.dust_flux_in = p_dust; sy.sw_flux = p_sw; sy.kmax = p_kmax;
  CHECK(obgc_synth_fill(h, &sy));
  obgc_synth_extra_args sx;
  memset(&sx, 0, sizeof sx);
  sx.nlev = nz; sx.ncol = nc; sx.col0 = 0;
  sx.surfacePressure = p_patm; sx.iceFraction = p_ice; sx.windSpeedSquared10m = p_wind; sx.atmCO2 = p_co2; sx.atmCO2_ALT_CO2 = p_co2alt;
  sx.depositionFlux = p_depo; sx.riverFlux = p_river; sx.seaIceFlux = p_seaice;
  CHECK(obgc_synth_fill_extra(h, &sx));
  CHECK(obgc_sync(h));
  double *tr = fetch(p_tr, 30 * kc), *temp = fetch(p_temp, kc), *salt = fetch(p_salt, kc), *zmid = fetch(p_zmid, kc), *dz = fetch(p_dz, kc),
         *fesed = fetch(p_fesed, kc), *lat = fetch(p_lat, nc), *dust = fetch(p_dust, nc), *patm = fetch(p_patm, nc), *wind = fetch(p_wind, nc);
  int* kmax = (int*)malloc(nc * sizeof(int));
  if (!tr || !wind || !kmax) { fprintf(stderr, "host allocation or download failed\n"); return 1; }
  HIP(hipMemcpy(kmax, p_kmax, nc * sizeof(int), hipMemcpyDeviceToHost));

  /* ---- 2. the model's arrays: (tracer, level, cell) with the tracer index fastest, (level, cell) with the level fastest ---- */
  const double RAD_TO_DEG = 180.0 / 3.14159265358979323846;
  double *m_tr = halloc(30 * kc), *m_h = halloc(kc), *m_zmid = halloc(kc), *m_t = halloc(kc), *m_s = halloc(kc), *m_fesed = halloc(kc);
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < nz; ++k) {
      const size_t pk = (size_t)k * nc + c, mk = (size_t)c * nz + k;
      for (int n = 0; n < 30; ++n) m_tr[mk * 30 + n] = tr[(size_t)n * kc + pk];
      m_h[mk] = dz[pk] / 100.0;          /* cm -> m */
      m_zmid[mk] = -zmid[pk] / 100.0;    /* positive-down cm -> negative-down m */
      m_t[mk] = temp[pk]; m_s[mk] = salt[pk]; m_fesed[mk] = fesed[pk];
    }
  for (int c = 0; c < nc; ++c) {
    lat[c] = lat[c] / RAD_TO_DEG;        /* degrees -> radians */
    patm[c] = patm[c] * 101325.0;        /* atm -> Pa */
    wind[c] = wind[c] / 1.0e4;           /* cm^2/s^2 -> m^2/s^2 */
    dust[c] = dust[c] * 10.0;            /* g/cm^2/s -> kg/m^2/s */
  }
  obgc_block_step_args a;
  memset(&a, 0, sizeof a);
  a.ecosysTracers = push(m_tr, 30 * kc); a.layerThickness = push(m_h, kc); a.zMid = push(m_zmid, kc);
  a.temperature = push(m_t, kc); a.salinity = push(m_s, kc); a.ironSedimentFlux = push(m_fesed, kc);
  a.maxLevelCell = p_kmax;
  a.latCell = push(lat, nc); a.atmosphericPressure = push(patm, nc); a.windSpeedSquared10m = push(wind, nc); a.dustFlux = push(dust, nc);
  a.shortWaveHeatFlux = p_sw; a.iceFraction = p_ice; a.atmosphericCO2 = p_co2; a.atmosphericCO2_ALT_CO2 = p_co2alt;
  a.depositionFlux = p_depo; a.riverFlux = p_river; a.seaIceFlux = p_seaice;     /* (nCells, 30) in Fortran order = [tracer][cell] */
  double *d_tend = dalloc(30 * kc), *d_sflux = dalloc(c30);
  a.ecosysTracersTend = d_tend; a.ecosysSurfaceFlux = d_sflux; a.accumulate = 0;
  if (!a.ecosysTracers || !a.dustFlux || !d_tend || !d_sflux) { fprintf(stderr, "device allocation or upload failed\n"); return 1; }

  /* ---- 3. the block, the diagnostics this model writes, the steps ---- */
  obgc_block* blk = NULL;
  CHECK(obgc_block_create(h, nc, nz, &blk));
  obgc_grid g;
  memset(&g, 0, sizeof g);
  g.nlev = nz; g.ncol = nc; g.layout = OBGC_LAYOUT_COLFAST; g.memspace = OBGC_MEM_DEVICE;
  const char* names[3] = {"CO3", "pH_3D", "photoC_TOT_zint"};
  obgc_diag_set* set = NULL;
  CHECK(obgc_diag_set_create(h, &g, 3, names, NULL, NULL, &set));
  CHECK(obgc_block_set_diagnostics(blk, obgc_diag_set_bgc(set), NULL, NULL));
  for (int s = 0; s < nsteps; ++s) {
    CHECK(obgc_block_step_device(blk, &a));
    CHECK(obgc_diag_set_accumulate(set, 1.0 / nsteps, s == 0));
  }
  CHECK(obgc_sync(h));

  /* ---- 4. what the model takes home ---- */
  double *tend = fetch(d_tend, 30 * kc), *sflux = fetch(d_sflux, c30);
  double *co3 = halloc(kc), *pcz = halloc(nc), *php = halloc(kc), *pha = halloc(kc), *sph = halloc(nc), *spha = halloc(nc);
  if (!tend || !sflux || !spha) { fprintf(stderr, "host allocation or download failed\n"); return 1; }
  CHECK(obgc_diag_set_download(set, "CO3", 0, co3, (long long)kc));                 /* the mean over the steps, (level, cell) */
  CHECK(obgc_diag_set_download(set, "photoC_TOT_zint", 0, pcz, (long long)nc));
  CHECK(obgc_block_get_state(blk, php, pha, sph, spha));                            /* for the model's restart file */
  double sum = 0.0, dsum = 0.0;
  for (size_t i = 0; i < 30 * kc; ++i) sum += tend[i] * (double)(1 + i % 7);
  for (size_t i = 0; i < kc; ++i) dsum += co3[i];
  printf("mesh_block_device cells=%d levels=%d steps=%d tendency checksum=%.17g mean CO3 sum=%.17g\n", nc, nz, nsteps, sum, dsum);
  if (argc > 3) {
    FILE* f = fopen(argv[3], "wb");
    int ok = f != NULL;
    ok = ok && fwrite(tend, sizeof(double), 30 * kc, f) == 30 * kc && fwrite(sflux, sizeof(double), c30, f) == c30;
    ok = ok && fwrite(co3, sizeof(double), kc, f) == kc && fwrite(pcz, sizeof(double), nc, f) == (size_t)nc;
    ok = ok && fwrite(php, sizeof(double), kc, f) == kc && fwrite(sph, sizeof(double), nc, f) == (size_t)nc;
    if (!ok) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    fclose(f);
  }
  CHECK(obgc_diag_set_destroy(set));
  CHECK(obgc_block_destroy(blk));
  CHECK(obgc_destroy(h));
  return 0;
}
