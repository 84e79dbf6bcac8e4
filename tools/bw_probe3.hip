// Streaming probe for the column-block kernel's access pattern, to compare OUTPUT (and input) layouts without the
// arithmetic.  A workgroup of 256 lanes owns 64 adjacent columns and marches down them in chunks of 4 levels (wavefront =
// level, lane = column), as bgc_column_kernel does; every cell reads NIN values and writes NOUT values, 8 bytes each.
// 70 KB of LDS per workgroup hold the occupancy at the column kernel's two workgroups per CU.
//
//   layout 0  plane-major (the native layout):       X[plane][level][column]
//   layout 1  level-major tiles:                     X[level][block][plane][64]      block = 64 adjacent columns
//   layout 2  column-block-major tiles:              X[block][level][plane][64]      a workgroup's whole output is one run
//   probe_pairs: layout 2 with two planes interleaved per column, X[block][level][plane / 2][64][2]: 16-byte accesses
//
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe3.hip -o gpurun_out/bw_probe3 && gpurun_out/bw_probe3 [ncol] [nlev] [LDS KB per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static size_t g_shm_kb = 70;   // LDS per workgroup: 70 KB = two workgroups per CU (the column kernel's occupancy), 140 KB = one

template <int LAYOUT>
__device__ __forceinline__ long long addr(int p, int k, int b, int lane, int np, int nlev, long long nblk) {
  if (LAYOUT == 0) return ((long long)p * nlev + k) * (nblk * 64) + (long long)b * 64 + lane;
  if (LAYOUT == 1) return (((long long)k * nblk + b) * np + p) * 64 + lane;
  return (((long long)b * nlev + k) * np + p) * 64 + lane;
}

template <int NIN, int NOUT, int LIN, int LOUT, int WORK>
__global__ __launch_bounds__(256, 2) void probe(const double* __restrict__ in, double* __restrict__ out, int nlev, long long nblk) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  for (int k = w; k < nlev; k += 4) {
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc += in[addr<LIN>(p, k, b, lane, NIN, nlev, nblk)];
    if (acc == 12345.678) lds[threadIdx.x] = acc;   // keeps the LDS allocation alive
#pragma unroll
    for (int p = 0; p < NOUT; ++p) {
      double v = acc + p;
#pragma unroll
      for (int i = 0; i < WORK; ++i) v = v * 1.0000001 + 0.5;   // dependent fp64 work between the stores
      out[addr<LOUT>(p, k, b, lane, NOUT, nlev, nblk)] = v;
    }
  }
}

// two planes per access: a lane reads / writes the values of planes 2q and 2q + 1 of its column as one 16-byte access
template <int NIN, int NOUT>
__global__ __launch_bounds__(256, 2) void probe_pairs(const double2* __restrict__ in, double2* __restrict__ out, int nlev, long long nblk) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long b = blockIdx.x;
  for (int k = w; k < nlev; k += 4) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NIN / 2; ++q) { const double2 v = in[((b * nlev + k) * (NIN / 2) + q) * 64 + lane]; acc += v.x + v.y; }
    if (acc == 12345.678) lds[threadIdx.x] = acc;
#pragma unroll
    for (int q = 0; q < NOUT / 2; ++q) out[((b * nlev + k) * (NOUT / 2) + q) * 64 + lane] = make_double2(acc + 2 * q, acc + 2 * q + 1);
  }
}

// the same march with every workgroup starting its plane loops at a different plane (a rotation by a hash of the block
// index): whether workgroups that move through the planes in lock-step camp on the same channels
template <int NIN, int NOUT>
__global__ __launch_bounds__(256, 2) void probe_rot(const double* __restrict__ in, double* __restrict__ out, int nlev, long long nblk) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const unsigned h = (unsigned)b * 2654435761u;
  const int r_in = NIN ? (int)((h >> 8) % (unsigned)(NIN ? NIN : 1)) : 0, r_out = (int)((h >> 12) % (unsigned)NOUT);
  for (int k = w; k < nlev; k += 4) {
    double acc = 0.0;
    for (int q = 0; q < NIN; ++q) { int p = q + r_in; if (p >= NIN) p -= NIN; acc += in[addr<2>(p, k, b, lane, NIN, nlev, nblk)]; }
    if (acc == 12345.678) lds[threadIdx.x] = acc;
    for (int q = 0; q < NOUT; ++q) { int p = q + r_out; if (p >= NOUT) p -= NOUT; out[addr<2>(p, k, b, lane, NOUT, nlev, nblk)] = acc + p; }
  }
}

// the same march with the NEXT level's loads requested before the current level's stores are issued (their values wait in
// registers until the stores are out): what the load -> wait -> store order of a level costs
template <int NIN, int NOUT>
__global__ __launch_bounds__(256, 2) void probe_ahead(const double* __restrict__ in, double* __restrict__ out, int nlev, long long nblk) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  double v[NIN];
  if (w < nlev) {
#pragma unroll
    for (int p = 0; p < NIN; ++p) v[p] = in[addr<2>(p, w, b, lane, NIN, nlev, nblk)];
  }
  for (int k = w; k < nlev; k += 4) {
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < NIN; ++p) acc += v[p];
    if (acc == 12345.678) lds[threadIdx.x] = acc;
    if (k + 4 < nlev) {
#pragma unroll
      for (int p = 0; p < NIN; ++p) v[p] = in[addr<2>(p, k + 4, b, lane, NIN, nlev, nblk)];
    }
#pragma unroll
    for (int p = 0; p < NOUT; ++p) out[addr<2>(p, k, b, lane, NOUT, nlev, nblk)] = acc + p;
#pragma unroll
    for (int p = 0; p < NIN; ++p) asm volatile("" : "+v"(v[p]));   // the loaded values are not touched before the stores are out
  }
}

template <int NIN, int NOUT>
void run_ahead(const double* in, double* out, int ncol, int nlev) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long nblk = ncol / 64;
  const size_t shm = g_shm_kb * 1024;
  CK(hipFuncSetAttribute((const void*)probe_ahead<NIN, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int r = 0; r < 2; ++r) probe_ahead<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe_ahead<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * nblk * 64 * nlev * (NIN + NOUT);
  printf("in %3d out %3d, column blocks, next level's loads ahead of the stores  : %8.3f ms  %6.2f TB/s\n", NIN, NOUT, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

template <int NIN, int NOUT>
void run_rot(const double* in, double* out, int ncol, int nlev) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long nblk = ncol / 64;
  const size_t shm = g_shm_kb * 1024;
  CK(hipFuncSetAttribute((const void*)probe_rot<NIN, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int r = 0; r < 2; ++r) probe_rot<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe_rot<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * nblk * 64 * nlev * (NIN + NOUT);
  printf("in %3d out %3d, column blocks, plane order rotated per workgroup      : %8.3f ms  %6.2f TB/s\n", NIN, NOUT, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

template <int NIN, int NOUT>
void run_pairs(const double* in, double* out, int ncol, int nlev) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long nblk = ncol / 64;
  const size_t shm = g_shm_kb * 1024;
  CK(hipFuncSetAttribute((const void*)probe_pairs<NIN, NOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int r = 0; r < 2; ++r) probe_pairs<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>((const double2*)in, (double2*)out, nlev, nblk);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe_pairs<NIN, NOUT><<<(unsigned)nblk, 256, shm>>>((const double2*)in, (double2*)out, nlev, nblk);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * nblk * 64 * nlev * (NIN + NOUT);
  printf("in %3d out %3d, column blocks, two planes per 16-byte access : %8.3f ms  %6.2f TB/s\n", NIN, NOUT, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

template <int NIN, int NOUT, int LIN, int LOUT, int WORK>
void run(const double* in, double* out, int ncol, int nlev) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long nblk = ncol / 64;
  const size_t shm = g_shm_kb * 1024;
  CK(hipFuncSetAttribute((const void*)probe<NIN, NOUT, LIN, LOUT, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int r = 0; r < 2; ++r) probe<NIN, NOUT, LIN, LOUT, WORK><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe<NIN, NOUT, LIN, LOUT, WORK><<<(unsigned)nblk, 256, shm>>>(in, out, nlev, nblk);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * nblk * 64 * nlev * (NIN + NOUT);
  printf("in %3d (layout %d) out %3d (layout %d) work %d : %8.3f ms  %6.2f TB/s\n", NIN, LIN, NOUT, LOUT, WORK, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int ncol = (argc > 1) ? atoi(argv[1]) : 320000;   // a multiple of 64
  const int nlev = (argc > 2) ? atoi(argv[2]) : 80;
  if (argc > 3) g_shm_kb = (size_t)atoi(argv[3]);
  const long long n = (long long)ncol * nlev;
  double *in, *out;
  CK(hipMalloc(&in, 8 * n * 48)); CK(hipMalloc(&out, 8 * n * 200));
  CK(hipMemset(in, 0, 8 * n * 48)); CK(hipMemset(out, 0, 8 * n * 200));
  printf("%d columns x %d levels, %.1f GB, %zu KB of LDS per workgroup\n", ncol, nlev, 8.0 * n * 248 * 1e-9, g_shm_kb);
  run<43, 200, 0, 0, 0>(in, out, ncol, nlev);
  run<43, 200, 0, 1, 0>(in, out, ncol, nlev);
  run<43, 200, 0, 2, 0>(in, out, ncol, nlev);
  run<43, 200, 1, 1, 0>(in, out, ncol, nlev);
  run<43, 200, 2, 2, 0>(in, out, ncol, nlev);
  run<43, 200, 0, 0, 4>(in, out, ncol, nlev);
  run<43, 200, 0, 2, 4>(in, out, ncol, nlev);
  run<43, 200, 2, 2, 4>(in, out, ncol, nlev);
  run_ahead<43, 200>(in, out, ncol, nlev);
  run_rot<43, 200>(in, out, ncol, nlev);
  run_rot<0, 200>(in, out, ncol, nlev);
  run_pairs<44, 200>(in, out, ncol, nlev);
  run_pairs<0, 200>(in, out, ncol, nlev);
  run<43, 37, 0, 0, 0>(in, out, ncol, nlev);    // tendencies only
  run<43, 37, 2, 2, 0>(in, out, ncol, nlev);
  run<0, 200, 0, 0, 0>(in, out, ncol, nlev);    // writes only
  run<0, 200, 0, 2, 0>(in, out, ncol, nlev);
  run<43, 0, 0, 0, 0>(in, out, ncol, nlev);     // (the sum is never stored: the reads may be elided)
  return 0;
}
