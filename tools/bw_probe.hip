// Streaming probe with the access pattern of the ecosystem cell kernel: every lane owns one cell, reads NIN planes and
// writes NOUT planes (plane = ncol*nz doubles, cell index fastest).  Prints the sustained HBM rate, i.e. the practical
// ceiling for a kernel of this shape on the device it runs on.
//   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o gpurun_out/bw_probe && gpurun_out/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NIN, int NOUT, bool NT>
__global__ __launch_bounds__(256) void probe(const double* __restrict__ in, double* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
#pragma unroll
  for (int p = 0; p < NIN; ++p) acc += in[p * n + i];
#pragma unroll
  for (int p = 0; p < NOUT; ++p) {
    if (NT) __builtin_nontemporal_store(acc + p, &out[p * n + i]);
    else out[p * n + i] = acc + p;
  }
}

template <int NIN, int NOUT, bool NT>
void run(const double* in, double* out, long long n) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned grid = (unsigned)((n + 255) / 256);
  for (int w = 0; w < 2; ++w) probe<NIN, NOUT, NT><<<grid, 256>>>(in, out, n);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe<NIN, NOUT, NT><<<grid, 256>>>(in, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * n * (NIN + NOUT);
  printf("in %3d out %3d nt %d : %8.3f ms  %6.2f TB/s\n", NIN, NOUT, (int)NT, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long long n = (argc > 1) ? atoll(argv[1]) : 20LL * 1000 * 1000;
  double *in, *out;
  CK(hipMalloc(&in, 8 * n * 64)); CK(hipMalloc(&out, 8 * n * 192));
  CK(hipMemset(in, 0, 8 * n * 64)); CK(hipMemset(out, 0, 8 * n * 192));
  run<64, 0, false>(in, out, n);   // read only (one store of the sum is elided: NOUT = 0)
  run<1, 64, false>(in, out, n);
  run<1, 64, true>(in, out, n);
  run<32, 32, false>(in, out, n);
  run<48, 192, false>(in, out, n);
  run<48, 192, true>(in, out, n);
  run<48, 32, false>(in, out, n);
  return 0;
}
