// Store-path probe at the occupancy of the ecosystem kernels (2 waves per SIMD, forced with dynamic LDS): how fast can
// 192 planes be written when each wave may keep only 63 vector-memory instructions in flight?
//   mode 0: one 8-byte store per plane and lane (512 B per instruction)
//   mode 1: lane pairs swap values so that every lane stores 16 contiguous bytes: even lanes into plane p, odd lanes into
//           plane p+1 (1 KB per instruction, half the instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double swap_pair(double v) {   // value of lane ^ 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int NIN, int NOUT, int MODE, int WORK>
__global__ __launch_bounds__(256) void probe(const double* __restrict__ in, double* __restrict__ out, long long n) {
  extern __shared__ double hog[];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
#pragma unroll
  for (int p = 0; p < NIN; ++p) acc += in[p * n + i];
  const bool odd = threadIdx.x & 1;
#pragma unroll
  for (int p = 0; p < NOUT; p += 2) {
    double a = acc + p, b = acc * 0.5 + p;
#pragma unroll
    for (int w = 0; w < WORK; ++w) { a = a * 1.0000001 + b; b = b * 0.9999999 + a; }   // dependent fp64 work between stores
    if (MODE == 0) {
      out[p * n + i] = a;
      out[(p + 1) * n + i] = b;
    } else {
      const double give = odd ? a : b;
      const double got = swap_pair(give);
      double2 v;
      v.x = odd ? got : a;
      v.y = odd ? b : got;
      double* dst = out + (odd ? (p + 1) * n + (i - 1) : p * n + i);
      *reinterpret_cast<double2*>(dst) = v;
    }
  }
}

template <int NIN, int NOUT, int MODE, int WORK>
void run(const double* in, double* out, long long n, int lds_bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned grid = (unsigned)((n + 255) / 256);
  CK(hipFuncSetAttribute((const void*)probe<NIN, NOUT, MODE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  for (int w = 0; w < 2; ++w) probe<NIN, NOUT, MODE, WORK><<<grid, 256, lds_bytes>>>(in, out, n);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) probe<NIN, NOUT, MODE, WORK><<<grid, 256, lds_bytes>>>(in, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = 8.0 * n * (NIN + NOUT);
  printf("in %3d out %3d mode %d work %3d lds %6d : %8.3f ms  %6.2f TB/s\n", NIN, NOUT, MODE, WORK, lds_bytes, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long long n = (argc > 1) ? atoll(argv[1]) : 20LL * 1000 * 1000;
  double *in, *out;
  CK(hipMalloc(&in, 8 * n * 64)); CK(hipMalloc(&out, 8 * n * 192));
  CK(hipMemset(in, 0, 8 * n * 64)); CK(hipMemset(out, 0, 8 * n * 192));
  const int occ2 = 72 * 1024, occ8 = 16 * 1024;
  run<48, 192, 0, 0>(in, out, n, occ8);
  run<48, 192, 0, 0>(in, out, n, occ2);
  run<48, 192, 1, 0>(in, out, n, occ8);
  run<48, 192, 1, 0>(in, out, n, occ2);
  run<48, 192, 0, 20>(in, out, n, occ2);
  run<48, 192, 1, 20>(in, out, n, occ2);
  run<48, 192, 0, 40>(in, out, n, occ2);
  run<48, 192, 1, 40>(in, out, n, occ2);
  run<48, 192, 0, 40>(in, out, n, occ8);
  return 0;
}
