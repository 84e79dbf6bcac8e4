// obgc_dms_source_sink_noalloc: DMS_SourceSink (DMS_mod.F90:156-770) for callers that cannot allocate device memory
// -- tight serving loops, harnesses that forbid hipMalloc, code under HIP graph capture (there the handle must have been
// created with OBGC_ASYNC, because a synchronous handle ends the call with a stream synchronisation, which a capturing
// stream refuses, and the stream to capture is obgc_stream(h); no test of this library captures a graph).
//
// obgc_dms_source_sink keeps the hand-off of its light pass (exp(-KPARdz), KPARdz and the PAR entering each level: three
// fp64 planes in the caller's index space) in the handle's arena, which it grows with hipMalloc on demand.  Here the same
// three planes live in ONE file-scope device array of fixed size, so the entry point calls no hipMalloc, no hipFree and no
// allocating library routine; it launches exactly the kernels of the allocating entry point.
//
//   * limit: OBGC_NOALLOC_MAX_CELLS (k,column) elements per plane, padding of the leading dimension included;
//     larger calls are rejected with OBGC_ERR_UNSUPPORTED and a message;
//   * every GPU the library runs on holds its own copy of the array (3 x 8 MiB) until the process ends;
//   * every call on a GPU uses that GPU's array: calls that overlap in time on one GPU (two handles, OBGC_ASYNC) race;
//   * the array keeps the last call's contents.  Nothing in it has to start at zero: the attenuation kernel writes every
//     active cell of the first two planes before the scan reads them and the scan writes every active cell of the third
//     before the cell kernel reads it.  Cells below the sea floor ARE read by the cell kernel (one unconditional load
//     batch per lane, inside the call's span) and their values discarded: whatever an earlier call left there never
//     reaches an output.
#include <mutex>

#include "obgc_handle.h"

using namespace obgc;

namespace {

constexpr long long kMaxCells = OBGC_NOALLOC_MAX_CELLS;
constexpr int kPlanes = 3;
constexpr int kMaxDevices = 64;

// In host code this name is only a handle for hipGetSymbolAddress: the device address is looked up once per GPU
// (noalloc_init_device) and every launch below uses that pointer.
static __device__ __align__(256) double scratch[kPlanes * kMaxCells];

std::mutex g_mu;
double* g_scratch[kMaxDevices] = {};

}  // namespace

hipError_t obgc::noalloc_init_device(int device) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_scratch[device]) return hipSuccess;
  // the caller (obgc_create) has made `device` current: the lookup returns THIS GPU's copy of the array
  double* p = nullptr;
  const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!p) return hipErrorInvalidSymbol;
  g_scratch[device] = p;
  return hipSuccess;
}

extern "C" long long obgc_noalloc_max_cells(void) { return kMaxCells; }

extern "C" int obgc_dms_source_sink_noalloc(obgc_handle* h, const obgc_dms_source_sink_args* a) {
  if (!h || !a) return OBGC_ERR_INVALID;
  const obgc_grid& g = a->grid;
  if (g.nlev < 1 || g.ncol < 0 || (g.ld != 0 && g.ld < g.ncol) || (g.ldc != 0 && g.ldc < g.ncol))
    return fail(h, OBGC_ERR_INVALID, "bad obgc_grid");
  if (g.memspace != OBGC_MEM_DEVICE || g.layout != OBGC_LAYOUT_COLFAST)
    return fail(h, OBGC_ERR_UNSUPPORTED,
                "obgc_dms_source_sink_noalloc: device pointers in the native column-fastest layout only (host memory and the "
                "reference layout need staging room: use obgc_dms_source_sink)");
  if (g.ncol == 0) return OBGC_OK;   // numColumns = 0: the reference's loops do not execute
  if (!a->tracers || !a->dz || !a->kmax || !a->SST || !a->sw_flux || !a->tend)
    return fail(h, OBGC_ERR_INVALID, "obgc_dms_source_sink_noalloc: null array");
  const GridView v = make_view(g);
  const long long span = v.at(v.nlev - 1, v.ncol - 1) + 1;   // elements of one (k,column) plane, padding included
  if (span > kMaxCells)
    return fail(h, OBGC_ERR_UNSUPPORTED,
                "obgc_dms_source_sink_noalloc: " + std::to_string(span) + " (k,column) elements per plane exceed the fixed scratch of " +
                    std::to_string(kMaxCells) + "; use obgc_dms_source_sink");
  double* base = nullptr;
  if (h->device >= 0 && h->device < kMaxDevices) {
    std::lock_guard<std::mutex> lk(g_mu);   // obgc_create on another thread may be resolving another device's entry
    base = g_scratch[h->device];
  }
  if (!base) return fail(h, OBGC_ERR_HIP, "obgc_dms_source_sink_noalloc: the scratch array was not resolved on this device");
  HIP_TRY(h, hipSetDevice(h->device));
  h->timed_valid = false;
  DmsKernelArgs K = {};
  K.g = v;
  K.tracers = a->tracers; K.dz = a->dz; K.kmax = a->kmax; K.SST = a->SST; K.sw = a->sw_flux; K.tend = a->tend;
  if (a->diag) K.diag = *a->diag;
  K.have_diag = a->diag ? 1 : 0;
  K.s_att = base; K.s_kpardz = base + kMaxCells; K.s_parin = base + 2 * kMaxCells;
  K.w_att = K.w_parin = nullptr;
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->s_main));
  launch_dms(K, h->dms, h->s_main);
  HIP_TRY(h, hipEventRecord(h->ev_end, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_c1, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e0, h->s_main));
  HIP_TRY(h, hipEventRecord(h->ev_e1, h->s_main));
  HIP_TRY(h, hipGetLastError());
  h->timed_valid = true;
  if (!(h->flags & OBGC_ASYNC)) HIP_TRY(h, hipStreamSynchronize(h->s_main));
  return OBGC_OK;
}
