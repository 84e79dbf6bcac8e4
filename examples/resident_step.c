/* A C host that keeps its tracers on the GPU: no Python, no PyTorch, only the C ABI of
 * include/obgc.h and the HIP runtime for device memory.
 *
 *   synthetic columns on the device -> 3 x ( BGC_SurfaceFluxes -> BGC_SourceSink -> explicit update )
 *
 * prints a checksum and, with a third argument, writes the final tracers as raw doubles; tests/test_c_example.py
 * reproduces them bit for bit through the Python host.
 * build: see examples/Makefile (hipcc -x c only because of <hip/hip_runtime_api.h>).
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

int main(int argc, char** argv) {
  const int nc = argc > 1 ? atoi(argv[1]) : 4096, nz = argc > 2 ? atoi(argv[2]) : 60, nsteps = 3;
  const double dt = 900.0;
  obgc_handle* h = NULL;
  CHECK(obgc_create(&h, 0, 0));
  const size_t kc = (size_t)nc * nz;
  double *tr = dalloc(30 * kc), *tend = dalloc(30 * kc), *temp = dalloc(kc), *salt = dalloc(kc), *zmid = dalloc(kc), *dz = dalloc(kc),
         *zbot = dalloc(kc), *fesed = dalloc(kc), *php = dalloc(kc), *pha = dalloc(kc), *lat = dalloc(nc), *dust = dalloc(nc),
         *sw = dalloc(nc);
  double* col[11];   /* surfacePressure iceFraction wind2 atmCO2 atmCO2_alt surface_pH surface_pH_alt surfaceDepth SST SSS + spare */
  for (int i = 0; i < 11; ++i) col[i] = dalloc(nc);
  double *depo = dalloc(30 * (size_t)nc), *river = dalloc(30 * (size_t)nc), *gas = dalloc(30 * (size_t)nc), *ice = dalloc(30 * (size_t)nc),
         *net = dalloc(30 * (size_t)nc);
  int* kmax = NULL;
  HIP(hipMalloc((void**)&kmax, nc * sizeof(int)));
  if (!tr || !tend || !net) { fprintf(stderr, "device allocation failed\n"); return 1; }

  /* the deterministic synthetic workload, generated on the device in the native layout */
  obgc_synth_args sy;
  memset(&sy, 0, sizeof sy);
  sy.nlev = nz; sy.ncol = nc; sy.col0 = 0;
  sy.tracers = tr; sy.temp = temp; sy.salt = salt; sy.zmid = zmid; sy.dz = dz; sy.zbot = zbot; sy.fesedflux = fesed;
  sy.lat = lat; sy.dust_flux_in = dust; sy.sw_flux = sw; sy.kmax = kmax;
  CHECK(obgc_synth_fill(h, &sy));
  /* uniform surface forcing, no external fluxes, no previous pH */
  double* hostc = (double*)malloc(nc * sizeof(double));
  const double cval[10] = {1.0, 0.1, 6.4e5, 400.0, 280.0, 0.0, 0.0, 5.0, 0.0, 35.0};
  for (int i = 0; i < 10; ++i) {
    for (int c = 0; c < nc; ++c) hostc[c] = cval[i];
    HIP(hipMemcpy(col[i], hostc, nc * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP(hipMemcpy(col[8], temp, nc * sizeof(double), hipMemcpyDeviceToDevice));                                   /* SST = T(level 1, :) : native layout row 0 */
  HIP(hipMemset(php, 0, kc * sizeof(double))); HIP(hipMemset(pha, 0, kc * sizeof(double)));
  HIP(hipMemset(depo, 0, 30 * (size_t)nc * sizeof(double))); HIP(hipMemset(river, 0, 30 * (size_t)nc * sizeof(double)));
  HIP(hipMemset(gas, 0, 30 * (size_t)nc * sizeof(double))); HIP(hipMemset(ice, 0, 30 * (size_t)nc * sizeof(double)));
  HIP(hipDeviceSynchronize());

  obgc_grid g;
  memset(&g, 0, sizeof g);
  g.nlev = nz; g.ncol = nc; g.layout = OBGC_LAYOUT_COLFAST; g.memspace = OBGC_MEM_DEVICE;
  for (int s = 0; s < nsteps; ++s) {
    obgc_bgc_surface_fluxes_args sf;
    memset(&sf, 0, sizeof sf);
    sf.grid = g; sf.tracers = tr; sf.surfacePressure = col[0]; sf.iceFraction = col[1]; sf.windSpeedSquared10m = col[2];
    sf.atmCO2 = col[3]; sf.atmCO2_ALT_CO2 = col[4]; sf.surface_pH = col[5]; sf.surface_pH_alt_co2 = col[6]; sf.surfaceDepth = col[7];
    sf.SST = col[8]; sf.SSS = col[9];
    sf.depositionFlux = depo; sf.riverFlux = river; sf.gasFlux = gas; sf.seaIceFlux = ice; sf.netFlux = net;
    sf.lcalc_O2_gas_flux = 1; sf.lcalc_CO2_gas_flux = 1;
    CHECK(obgc_bgc_surface_fluxes(h, &sf));
    obgc_bgc_source_sink_args ss;
    memset(&ss, 0, sizeof ss);
    ss.grid = g; ss.tracers = tr; ss.temp = temp; ss.salt = salt; ss.zmid = zmid; ss.dz = dz; ss.zbot = zbot; ss.lat = lat; ss.kmax = kmax;
    ss.fesedflux = fesed; ss.dust_flux_in = dust; ss.sw_flux = sw; ss.tend = tend; ss.ph_prev = php; ss.ph_prev_alt = pha;
    ss.diag = NULL; ss.alt_co2_use_eco = 1;
    CHECK(obgc_bgc_source_sink(h, &ss));
    obgc_apply_tendencies_args ap;
    memset(&ap, 0, sizeof ap);
    ap.grid = g; ap.ntracer = 30; ap.tracers = tr; ap.tend = tend; ap.net_flux = net; ap.dz = dz; ap.kmax = kmax; ap.dt = dt;
    ap.clamp_nonneg = 1;
    CHECK(obgc_apply_tendencies(h, &ap));
  }
  CHECK(obgc_sync(h));
  double* back = (double*)malloc(30 * kc * sizeof(double));
  HIP(hipMemcpy(back, tr, 30 * kc * sizeof(double), hipMemcpyDeviceToHost));
  double sum = 0.0, wsum = 0.0;
  for (size_t i = 0; i < 30 * kc; ++i) { sum += back[i]; wsum += back[i] * (double)(i % 1009); }
  printf("resident_step columns=%d levels=%d steps=%d checksum=%.17g weighted=%.17g\n", nc, nz, nsteps, sum, wsum);
  if (argc > 3) {
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(back, sizeof(double), 30 * kc, f) != 30 * kc) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    fclose(f);
  }
  CHECK(obgc_destroy(h));
  return 0;
}
