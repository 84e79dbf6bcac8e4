// Explicit update of resident tracers (no reference counterpart -- the host model owns
// time integration).  Pure streaming: 16 B read + 8 B written per tracer value, HBM-bound.
//   x <- x + dt * (tend + [k == 0] * net_flux / dz)     for k < kmax(column);   optional clamp at zero.
#include <cstdint>

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

// Running means of many arrays in one launch: blockIdx.y picks the (source, destination) pair from a table that
// travels as a kernel argument (scalar loads), so ~160 small streaming updates cost one launch, not 160.
__global__ __launch_bounds__(kBlock) void accumulate_planes_kernel(GridView g, AccumTable t, double w, int reset) {
  const long long ncell = (long long)g.nlev * g.ncol;
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= ncell) return;
  int k, c;
  if (g.kfast) { k = (int)(idx % g.nlev); c = (int)(idx / g.nlev); }
  else { c = (int)(idx % g.ncol); k = (int)(idx / g.ncol); }
  const long long ix = g.at(k, c);
  const double* __restrict__ src = t.src[blockIdx.y];
  double* __restrict__ dst = t.dst[blockIdx.y];
  const double v = w * src[ix];
  dst[ix] = reset ? v : dst[ix] + v;
}

__global__ __launch_bounds__(kBlock) void accumulate_vectors_kernel(int ncol, AccumTable t, double w, int reset) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= ncol) return;
  const int e = t.nplane + blockIdx.y;
  const double v = w * t.src[e][c];
  t.dst[e][c] = reset ? v : t.dst[e][c] + v;
}

// column-fastest layout with even extents: two adjacent columns per lane, 16-byte accesses (see obgc_transpose.hip)
struct alignas(16) D2 { double x, y; };

__global__ __launch_bounds__(kBlock) void apply_tendencies_vec2_kernel(GridView g, double* __restrict__ x, const double* __restrict__ t,
                                                                      const double* __restrict__ nf, const double* __restrict__ dz,
                                                                      const int* __restrict__ kmax, double dt, int clamp) {
  const long long npair = (long long)g.nlev * (g.ncol / 2);
  const long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= npair) return;
  const int half = g.ncol / 2;
  const int c = (int)(idx % half) * 2, k = (int)(idx / half);
  const int km0 = kmax[c], km1 = kmax[c + 1];
  if (k >= km0 && k >= km1) return;
  const long long n = blockIdx.y;
  const long long ix = n * g.plane + g.at(k, c);
  const D2 tv = *reinterpret_cast<const D2*>(t + ix);
  D2 xv = *reinterpret_cast<const D2*>(x + ix);
  double s0 = tv.x, s1 = tv.y;
  if (k == 0 && nf) {
    const D2 dzv = *reinterpret_cast<const D2*>(dz + g.at(0, c));
    const D2 nv = *reinterpret_cast<const D2*>(nf + n * g.ldc + c);
    s0 = s0 + nv.x / dzv.x;
    s1 = s1 + nv.y / dzv.y;
  }
  double v0 = xv.x + dt * s0, v1 = xv.y + dt * s1;
  if (clamp) { v0 = dmax(v0, 0.0); v1 = dmax(v1, 0.0); }
  if (k < km0) xv.x = v0;
  if (k < km1) xv.y = v1;
  *reinterpret_cast<D2*>(x + ix) = xv;
}

}  // namespace

void launch_accumulate(const GridView& g, const AccumTable& t, double w, int reset, hipStream_t s) {
  const long long ncell = (long long)g.nlev * g.ncol;
  if (t.nplane > 0 && ncell > 0)
    hipLaunchKernelGGL(accumulate_planes_kernel, dim3((unsigned)((ncell + kBlock - 1) / kBlock), (unsigned)t.nplane), dim3(kBlock), 0, s,
                       g, t, w, reset);
  if (t.nvec > 0 && g.ncol > 0)
    hipLaunchKernelGGL(accumulate_vectors_kernel, dim3((unsigned)((g.ncol + kBlock - 1) / kBlock), (unsigned)t.nvec), dim3(kBlock), 0, s,
                       g.ncol, t, w, reset);
}

void launch_apply_tendencies(const GridView& g, int ntracer, double* x, const double* t, const double* nf, const double* dz,
                             const int* kmax, double dt, int clamp, hipStream_t s) {
  const long long ncell = (long long)g.nlev * g.ncol;
  if (ncell <= 0 || ntracer <= 0) return;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
  if (!g.kfast && g.ncol % 2 == 0 && g.sk % 2 == 0 && g.plane % 2 == 0 && g.ldc % 2 == 0 && al16(x) && al16(t) && (!nf || (al16(nf) && al16(dz)))) {
    const long long npair = (long long)g.nlev * (g.ncol / 2);
    const dim3 grid2((unsigned)((npair + kBlock - 1) / kBlock), (unsigned)ntracer);
    hipLaunchKernelGGL(apply_tendencies_vec2_kernel, grid2, dim3(kBlock), 0, s, g, x, t, nf, dz, kmax, dt, clamp);
    return;
  }
  const dim3 grid((unsigned)((ncell + kBlock - 1) / kBlock), (unsigned)ntracer);
  hipLaunchKernelGGL(apply_tendencies_kernel, grid, dim3(kBlock), 0, s, g, x, t, nf, dz, kmax, dt, clamp);
}

}  // namespace obgc
