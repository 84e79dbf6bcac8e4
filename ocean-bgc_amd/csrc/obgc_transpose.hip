// LDS-tiled transposition between the reference's level-fastest arrays X(k,column,n)
// and the engine's column-fastest native order X[n][k][column].
//
// Pure data movement (16 B of traffic per element): both the global read and the global
// write of a wavefront are contiguous rows; the 64 x 64 fp64 tile is staged in LDS with one
// element of padding per row, so that the transposed ds_read_b64 of 32 lanes walks all 64
// banks (row stride 65 doubles = 130 dwords -> bank advance of 2 per lane) without
// conflicts.  One workgroup (256 lanes) per tile and plane.
#include <cstdint>

#include "obgc_internal.h"

namespace obgc {

namespace {

constexpr int TILE = 64;

// src(r, q) = src[r*src_ld + q]  ->  dst(q, r) = dst[q*dst_ld + r],  r < R, q < Q
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, int R,
                                                        int Q, long long src_ld, long long dst_ld, long long src_plane,
                                                        long long dst_plane) {
  __shared__ double tile[TILE][TILE + 1];
  const int q0 = blockIdx.x * TILE, r0 = blockIdx.y * TILE;
  src += (long long)blockIdx.z * src_plane;
  dst += (long long)blockIdx.z * dst_plane;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll
  for (int i = 0; i < TILE; i += 4) {
    const int r = r0 + ty + i, q = q0 + tx;
    if (r < R && q < Q) tile[ty + i][tx] = src[(long long)r * src_ld + q];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TILE; i += 4) {
    const int q = q0 + ty + i, r = r0 + tx;
    if (r < R && q < Q) dst[(long long)q * dst_ld + r] = tile[tx][ty + i];
  }
}

// The same tile transposition for a TABLE of unrelated planes (one launch, blockIdx.z picks the pair).  With `kmax`
// (level-fastest destination only: rows r = levels, q = columns) cells at or below the sea floor are not written,
// so a destination that already holds the reference's zero fill keeps it.
__global__ __launch_bounds__(256) void transpose_table_kernel(PlaneTable t, int R, int Q, long long src_ld, long long dst_ld,
                                                              const int* __restrict__ kmax) {
  __shared__ double tile[TILE][TILE + 1];
  __shared__ int kmx[TILE];
  const int q0 = blockIdx.x * TILE, r0 = blockIdx.y * TILE;
  const double* __restrict__ src = t.src[blockIdx.z];
  double* __restrict__ dst = t.dst[blockIdx.z];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (threadIdx.x < TILE) kmx[threadIdx.x] = (kmax && q0 + (int)threadIdx.x < Q) ? kmax[q0 + threadIdx.x] : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < TILE; i += 4) {
    const int r = r0 + ty + i, q = q0 + tx;
    if (r < R && q < Q) tile[ty + i][tx] = src[(long long)r * src_ld + q];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TILE; i += 4) {
    const int q = q0 + ty + i, r = r0 + tx;
    if (r < R && q < Q && r < kmx[ty + i]) dst[(long long)q * dst_ld + r] = tile[tx][ty + i];
  }
}

// 16 bytes per lane on both sides (global_load/store_dwordx4): half the vector-memory instructions of the kernel above,
// which is what bounds a low-arithmetic streaming kernel on this chip long before HBM does (8-byte accesses level off
// near 3-4 TB/s).  Needs even extents and leading dimensions and 16-byte aligned planes; the launcher checks.
struct alignas(16) D2 { double x, y; };

__global__ __launch_bounds__(256) void transpose_table_vec_kernel(PlaneTable t, int R, int Q, long long src_ld, long long dst_ld,
                                                                  const int* __restrict__ kmax) {
  __shared__ double tile[TILE][TILE + 1];
  __shared__ int kmx[TILE];
  const int q0 = blockIdx.x * TILE, r0 = blockIdx.y * TILE;
  const double* __restrict__ src = t.src[blockIdx.z];
  double* __restrict__ dst = t.dst[blockIdx.z];
  const int tx = (threadIdx.x & 31) * 2, ty = threadIdx.x >> 5;   // 32 pairs x 8 rows
  if (threadIdx.x < TILE) kmx[threadIdx.x] = (kmax && q0 + (int)threadIdx.x < Q) ? kmax[q0 + threadIdx.x] : 0x7fffffff;
#pragma unroll
  for (int i = 0; i < TILE; i += 8) {
    const int r = r0 + ty + i, q = q0 + tx;
    if (r < R && q < Q) {
      const D2 v = *reinterpret_cast<const D2*>(src + (long long)r * src_ld + q);
      tile[ty + i][tx] = v.x;
      tile[ty + i][tx + 1] = v.y;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TILE; i += 8) {
    const int q = q0 + ty + i, r = r0 + tx;
    if (r < R && q < Q) {
      D2 v;
      v.x = tile[tx][ty + i];
      v.y = tile[tx + 1][ty + i];
      const int lim = kmx[ty + i];
      double* o = dst + (long long)q * dst_ld + r;
      if (r + 1 < lim) *reinterpret_cast<D2*>(o) = v;
      else if (r < lim) *o = v.x;
    }
  }
}

void launch_table(const PlaneTable& t, int R, int Q, long long src_ld, long long dst_ld, const int* kmax, hipStream_t s) {
  if (R <= 0 || Q <= 0 || t.n <= 0) return;
  bool vec = (R % 2 == 0) && (Q % 2 == 0) && (src_ld % 2 == 0) && (dst_ld % 2 == 0) && R <= 65535 * TILE;
  for (int i = 0; i < t.n && vec; ++i) vec = (((uintptr_t)t.src[i] | (uintptr_t)t.dst[i]) & 15u) == 0;
  if (vec) {
    dim3 grid((Q + TILE - 1) / TILE, (R + TILE - 1) / TILE, t.n);
    hipLaunchKernelGGL(transpose_table_vec_kernel, grid, dim3(256), 0, s, t, R, Q, src_ld, dst_ld, kmax);
    return;
  }
  const int maxRows = 65535 * TILE;
  for (int r0 = 0; r0 < R; r0 += maxRows) {
    const int rr = (R - r0 < maxRows) ? (R - r0) : maxRows;
    PlaneTable tt = t;
    for (int i = 0; i < t.n; ++i) { tt.src[i] = t.src[i] + (long long)r0 * src_ld; tt.dst[i] = t.dst[i] + r0; }
    dim3 grid((Q + TILE - 1) / TILE, (rr + TILE - 1) / TILE, t.n);
    hipLaunchKernelGGL(transpose_table_kernel, grid, dim3(256), 0, s, tt, rr, Q, src_ld, dst_ld, kmax);
  }
}

void launch(const double* src, double* dst, int R, int Q, long long src_ld, long long dst_ld, long long src_plane,
            long long dst_plane, int nplanes, hipStream_t s) {
  if (R <= 0 || Q <= 0 || nplanes <= 0) return;
  // gridDim.y is limited to 65535 tiles = 4.19 M rows per launch
  const int maxRows = 65535 * TILE;
  for (int r0 = 0; r0 < R; r0 += maxRows) {
    const int rr = (R - r0 < maxRows) ? (R - r0) : maxRows;
    dim3 grid((Q + TILE - 1) / TILE, (rr + TILE - 1) / TILE, nplanes);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, src + (long long)r0 * src_ld, dst + r0, rr, Q, src_ld,
                       dst_ld, src_plane, dst_plane);
  }
}

}  // namespace

void launch_transpose_k2c(const double* src, double* dst, int nlev, int ncol, long long src_ld, long long src_plane,
                          long long dst_ld, long long dst_plane, int nplanes, hipStream_t s) {
  launch(src, dst, ncol, nlev, src_ld, dst_ld, src_plane, dst_plane, nplanes, s);   // rows = columns, q = level
}

void launch_transpose_c2k(const double* src, double* dst, int nlev, int ncol, long long src_ld, long long src_plane,
                          long long dst_ld, long long dst_plane, int nplanes, hipStream_t s) {
  launch(src, dst, nlev, ncol, src_ld, dst_ld, src_plane, dst_plane, nplanes, s);   // rows = levels, q = column
}

void launch_transpose_table_k2c(const PlaneTable& t, int nlev, int ncol, long long src_ld, long long dst_ld, hipStream_t s) {
  launch_table(t, ncol, nlev, src_ld, dst_ld, nullptr, s);   // rows = columns, q = level
}

void launch_transpose_table_c2k(const PlaneTable& t, int nlev, int ncol, long long src_ld, long long dst_ld, const int* kmax,
                                hipStream_t s) {
  // rows = levels: one launch never needs the row split (nlev is small), so the mask indexes columns q directly
  launch_table(t, nlev, ncol, src_ld, dst_ld, kmax, s);
}

}  // namespace obgc
