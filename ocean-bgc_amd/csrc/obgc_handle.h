// The engine handle and the error plumbing shared by the host-side translation units of libobgc.
#pragma once

#include <string>

#include "obgc_internal.h"

struct obgc_handle {
  int device = 0;
  unsigned flags = 0;
  hipStream_t s_main = nullptr;   // column-sweep kernels, copies, transposes
  hipStream_t s_aux = nullptr;    // cell-parallel carbonate kernel runs beside the sweep
  hipStream_t s_aux2 = nullptr;   // column integrals (diagnostics) run beside both
  hipStream_t s_up = nullptr, s_down = nullptr;   // host-memory path: uploads / downloads of column chunks (created on first use)
  int host_chunks = 4;    // OBGC_HOST_CHUNKS: column chunks of the pipelined host-memory path (0 or 1 = stage everything, then run)
  int host_threads = 1;   // OBGC_HOST_THREADS=0: no helper thread (downloads in line, as when thread creation fails)
  size_t host_pipeline_min_bytes = (size_t)64 << 20;   // OBGC_HOST_PIPELINE_MIN_MB: smaller calls are staged in one piece
  hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_e0 = nullptr, ev_e1 = nullptr,
             ev_t0 = nullptr, ev_t1 = nullptr, ev_aux_done = nullptr, ev_aux2_done = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
  bool timed_valid = false, tr_valid = false, cell_valid = false;
  int last_sched = -1;    // schedule the last obgc_bgc_source_sink / obgc_fused_step used (obgc_last_schedule)
  // the separate kernels' sweep as three launches (bgc_walk_kernel): 0 never, 1 for batches up to split_max_cells, 2 always
  // (OBGC_SPLIT_SWEEP, OBGC_SPLIT_MAX_CELLS)
  int split_sweep = 1;
  long long split_max_cells = 300000;
  int sweep_lanes = 0;    // experiment knob OBGC_SWEEP_LANES: 16 / 32 / 64 active lanes per wavefront in the sweep, 0 = auto
  int kfast_tile = 131072;   // columns per transposed tile of the reference-layout path (OBGC_KFAST_TILE)
  int kfast_direct = 0;   // experiment knob OBGC_KFAST_DIRECT=1: run the kernels on the reference layout through strides
  int kfast_hybrid = 1;   // OBGC_KFAST_HYBRID=0: transpose every plane (the first implementation) instead of only the column kernels' planes
  int sched = -1;  // OBGC_SCHED: 0 carbonate from the start, 1 carbonate released after the cell kernel, 2 carbonate inside the cell kernel's
                   // wavefronts, -1 (default) = 1, or for batches of more than 10 M cells 2 with diagnostics and 0 without (see bgc_launch)
  int column = 1;  // OBGC_COLUMN_KERNEL=0: native-layout calls use the separate attenuation / scan / cell / sweep kernels (the form the
                   // reference layout always takes) instead of the column-block kernel; 2: the column-block kernel whatever the size
  long long column_min_cells = 700000;   // OBGC_COLUMN_MIN_CELLS: native-layout calls below this many cells take the separate kernels
                                         // (use_column_kernel)
  obgc_bgc_params bgc;
  double bgc_recip[40] = {};   // BgcKernelArgs::recip for `bgc`, computed on first use after the parameters change
  bool bgc_recip_valid = false;
  obgc_dms_params dms;
  obgc_macros_params macros;
  // device scratch: one growing arena, carved per call
  char* arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  std::string err;
};

namespace obgc {

inline int fail(obgc_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, call)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(h, OBGC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

// One file-scope device array backs obgc_dms_source_sink_noalloc (obgc_noalloc.hip); obgc_create resolves its address
// on the handle's GPU.  Returns hipSuccess or the error of the symbol lookup.
hipError_t noalloc_init_device(int device);

}  // namespace obgc
