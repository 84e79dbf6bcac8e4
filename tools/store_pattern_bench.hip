// Write-bandwidth of the cell kernel's store pattern, without the kernel: P planes of N doubles, every lane stores its
// cell(s) into each plane.  Build: hipcc --offload-arch=gfx950 -O3 -o store_pattern_bench tools/store_pattern_bench.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) D2 { double x, y; };
struct alignas(32) D4 { double x, y, z, w; };

template <int WG>
__global__ __launch_bounds__(WG) void store8(double* base, long long n, long long plane, int P) {
  const long long i = (long long)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const double v = (double)i;
#pragma unroll 8
  for (int p = 0; p < P; ++p) base[(long long)p * plane + i] = v + p;
}

template <int WG>
__global__ __launch_bounds__(WG) void store8_nt(double* base, long long n, long long plane, int P) {
  const long long i = (long long)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const double v = (double)i;
#pragma unroll 8
  for (int p = 0; p < P; ++p) __builtin_nontemporal_store(v + p, &base[(long long)p * plane + i]);
}

template <int WG>
__global__ __launch_bounds__(WG) void store16(double* base, long long n, long long plane, int P) {
  const long long i = ((long long)blockIdx.x * WG + threadIdx.x) * 2;
  if (i >= n) return;
  const double v = (double)i;
#pragma unroll 8
  for (int p = 0; p < P; ++p) *reinterpret_cast<D2*>(&base[(long long)p * plane + i]) = D2{v + p, v - p};
}

template <int WG>
__global__ __launch_bounds__(WG) void store32(double* base, long long n, long long plane, int P) {
  const long long i = ((long long)blockIdx.x * WG + threadIdx.x) * 4;
  if (i >= n) return;
  const double v = (double)i;
#pragma unroll 8
  for (int p = 0; p < P; ++p) *reinterpret_cast<D4*>(&base[(long long)p * plane + i]) = D4{v + p, v - p, v, v * p};
}

// plane-major inside the workgroup: wave w of the group writes planes w, w+W, ... for ALL cells of the group (values
// would come through LDS in a real kernel): 16 B per lane, 1 KB contiguous per store instruction
template <int WG>
__global__ __launch_bounds__(WG) void store_planemajor(double* base, long long n, long long plane, int P) {
  constexpr int W = WG / 64;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long cell0 = (long long)blockIdx.x * (WG / 1);   // WG cells per group
  for (int p = w; p < P; p += W)
    for (int j = 0; j < WG; j += 128) {
      const long long i = cell0 + j + lane * 2;
      if (i < n) *reinterpret_cast<D2*>(&base[(long long)p * plane + i]) = D2{(double)i + p, (double)i - p};
    }
}

// the cell kernel's traffic without its arithmetic: R loads per lane first (one round trip), then P stores; dynamic LDS
// only limits how many workgroups fit a CU (40 KB per 128-thread group = two wavefronts per SIMD, the cell kernel's
// occupancy); `work` dependent fp64 fmas between stores stand in for the arithmetic
template <int WG>
__global__ __launch_bounds__(WG) void mix(const double* __restrict__ in, double* __restrict__ out, long long n, long long plane, int R,
                                          int P, int work) {
  extern __shared__ double lds[];
  const long long i = (long long)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
#pragma unroll 8
  for (int r = 0; r < R; ++r) acc += in[(long long)r * plane + i];
  if (acc == 12345.678) lds[threadIdx.x] = acc;   // keeps the allocation
  for (int p = 0; p < P; ++p) {
    for (int w = 0; w < work; ++w) acc = __builtin_fma(acc, 1.0000001, 1.0e-9);
    out[(long long)p * plane + i] = acc + p;
  }
}

// the same traffic from resident workgroups that walk the batches: the loads of the next batch are issued right behind
// the stores of this one, so their latency overlaps the draining of the stores instead of following it
template <int WG, int R>
__global__ __launch_bounds__(WG) void mix_persistent(const double* __restrict__ in, double* __restrict__ out, long long n, long long plane,
                                                     int P, int work) {
  extern __shared__ double lds[];
  const long long nb = (n + WG - 1) / WG;
  double v[R];
  long long b = blockIdx.x;
  if (b < nb) {
    const long long i = b * WG + threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = i < n ? in[(long long)r * plane + i] : 0.0;
  }
  for (; b < nb; b += gridDim.x) {
    const long long i = b * WG + threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += v[r];
    if (acc == 12345.678) lds[threadIdx.x] = acc;
    for (int p = 0; p < P; ++p) {
      for (int w = 0; w < work; ++w) acc = __builtin_fma(acc, 1.0000001, 1.0e-9);
      if (i < n) out[(long long)p * plane + i] = acc + p;
    }
    const long long bn = b + gridDim.x;
    if (bn < nb) {
      const long long in_i = bn * WG + threadIdx.x;
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = in_i < n ? in[(long long)r * plane + in_i] : 0.0;
    }
  }
}

int main() {
  const long long n = 6000000, plane = n;
  const int Pmax = 133;
  double* d;
  if (hipMalloc((void**)&d, sizeof(double) * plane * Pmax) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, int P, int cells_per_thread, int wg, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("%-22s P=%3d WG=%3d : %.3f ms  %.0f GB/s\n", name, P, wg, best, (double)n * P * 8 / best / 1e6);
    fflush(stdout);
  };
  for (int P : {1, 16, 133}) {
    time("8 B/lane", P, 1, 128, [&] { hipLaunchKernelGGL(store8<128>, dim3((n + 127) / 128), dim3(128), 0, 0, d, n, plane, P); });
    time("8 B/lane", P, 1, 256, [&] { hipLaunchKernelGGL(store8<256>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, plane, P); });
    time("8 B/lane", P, 1, 64, [&] { hipLaunchKernelGGL(store8<64>, dim3((n + 63) / 64), dim3(64), 0, 0, d, n, plane, P); });
    time("8 B/lane nontemporal", P, 1, 128, [&] { hipLaunchKernelGGL(store8_nt<128>, dim3((n + 127) / 128), dim3(128), 0, 0, d, n, plane, P); });
    time("16 B/lane", P, 2, 128, [&] { hipLaunchKernelGGL(store16<128>, dim3((n / 2 + 127) / 128), dim3(128), 0, 0, d, n, plane, P); });
    time("32 B/lane", P, 4, 128, [&] { hipLaunchKernelGGL(store32<128>, dim3((n / 4 + 127) / 128), dim3(128), 0, 0, d, n, plane, P); });
    time("plane-major 16 B", P, 1, 128, [&] { hipLaunchKernelGGL(store_planemajor<128>, dim3((n + 127) / 128), dim3(128), 0, 0, d, n, plane, P); });
    time("plane-major 16 B", P, 1, 256, [&] { hipLaunchKernelGGL(store_planemajor<256>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, plane, P); });
  }
  double* in;
  if (hipMalloc((void**)&in, sizeof(double) * plane * 35) != hipSuccess) return 1;
  hipMemset(in, 0, sizeof(double) * plane * 35);
  for (int ldsKB : {0, 20, 40}) {
    for (int work : {0, 8, 24}) {
      char name[64];
      snprintf(name, sizeof name, "35 ld + 133 st lds%2dK w%2d", ldsKB, work);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mix<128>, dim3((n + 127) / 128), dim3(128), (size_t)ldsKB * 1024, 0, in, d, n, plane, 35, 133, work);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%-30s : %.3f ms  %.0f GB/s (reads + writes)\n", name, best, (double)n * (35 + 133) * 8 / best / 1e6);
      fflush(stdout);
    }
  }
  for (int work : {0, 8, 16}) {
    for (int wgPerCU : {4, 8}) {
      char name[64];
      snprintf(name, sizeof name, "persistent %d WG/CU w%2d", wgPerCU, work);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix_persistent<128, 35>), dim3(256 * wgPerCU), dim3(128), (size_t)(160 / wgPerCU) * 1024, 0, in, d, n, plane, 133, work);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      printf("%-30s : %.3f ms  %.0f GB/s (reads + writes)\n", name, best, (double)n * (35 + 133) * 8 / best / 1e6);
      fflush(stdout);
    }
  }
  hipFree(in);
  hipFree(d);
  return 0;
}
