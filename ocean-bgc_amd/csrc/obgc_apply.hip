// Explicit update of resident tracers (no reference counterpart -- the host model owns
// time integration).  Pure streaming: 16 B read + 8 B written per tracer value, HBM-bound.
//   x <- x + dt * (tend + [k == 0] * net_flux / dz)     for k < kmax(column);   optional clamp at zero.
#include "obgc_internal.h"

namespace obgc {

namespace {

constexpr int kBlock = 256;

// blockIdx.y = tracer plane; lanes run along the contiguous axis of the layout
__global__ __launch_bounds__(kBlock) void apply_tendencies_kernel(GridView g, double* __restrict__ x, const double* __restrict__ t,
                                                                 const double* __restrict__ nf, const double* __restrict__ dz,
                                                                 const int* __restrict__ kmax, double dt, int clamp) {
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  if (k >= kmax[c]) return;
  const long long n = blockIdx.y;
  const long long ix = n * g.plane + g.at(k, c);
  double s = t[ix];
  if (k == 0 && nf) s = s + nf[n * g.ldc + c] / dz[g.at(0, c)];
  double v = x[ix] + dt * s;
  if (clamp) v = dmax(v, 0.0);
  x[ix] = v;
}

}  // namespace

void launch_apply_tendencies(const GridView& g, int ntracer, double* x, const double* t, const double* nf, const double* dz,
                             const int* kmax, double dt, int clamp, hipStream_t s) {
  const long long ncell = (long long)g.nlev * g.ncol;
  if (ncell <= 0 || ntracer <= 0) return;
  const dim3 grid((unsigned)((ncell + kBlock - 1) / kBlock), (unsigned)ntracer);
  hipLaunchKernelGGL(apply_tendencies_kernel, grid, dim3(kBlock), 0, s, g, x, t, nf, dz, kmax, dt, clamp);
}

}  // namespace obgc
