// Diagnostics selected BY NAME, averaged on the device and brought to the host at the output interval.
//
// The reference hands every diagnostic of its derived types to the host at every call (BGC_parms.F90:192-321: 99 members;
// DMS_parms.F90:125-154: 27; MACROS_parms.F90:105-113: 6) and the host model averages and writes the ones its namelist
// selects.  With the fields resident on the GPU that is 65 % of the bytes of a step for values nobody reads.  An
// obgc_diag_set is a selection by the reference's member names (without the diag_ prefix): it owns the instantaneous
// arrays and the running means of the selected members only, hands the kernels pointer structs in which every other
// member is NULL (a NULL member is neither computed into memory nor zero-filled), advances the means in one launch per
// module and returns or writes them in the reference's X(k,column[,autotroph]) order.  No counterpart in the reference.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "obgc_handle.h"

using namespace obgc;

namespace {

enum Kind { KC = 0, KCA = 1, CA = 2, C = 3 };

struct Member {
  std::string name;
  int module;   // 0 BGC, 1 DMS, 2 MACROS
  int kind;
  double **inst_slot, **mean_slot;   // members of the pointer structs below
  double *inst, *mean;
};

}  // namespace

struct obgc_diag_set {
  obgc_handle* h;
  obgc_grid grid;
  GridView v;
  obgc_bgc_diag bi, bm;
  obgc_dms_diag di, dm;
  obgc_macros_diag mi, mm;
  int nmod[3];
  std::vector<Member> members;
  double* dev;   // own allocation: dense planes (native layout) and the per-column members (both layouts)
};

namespace {

void set_err(char* err, int errlen, const std::string& msg) {
  if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
}

// "CO3", "bgc.CO3", "dms.DMS_R_TOTAL", "macros.PROT_S_TOTAL" -> member slots of the set's pointer structs
bool lookup(obgc_diag_set* s, const std::string& full, Member& m) {
  std::string mod, nm = full;
  const size_t dot = full.find('.');
  if (dot != std::string::npos) { mod = full.substr(0, dot); nm = full.substr(dot + 1); }
  if (nm.rfind("diag_", 0) == 0) nm = nm.substr(5);   // the reference's member names carry the prefix
  int hits = 0;
#define TRY(modname, modno, kindv, istruct, mstruct, n)                                  \
  if ((mod.empty() || mod == modname) && nm == #n) {                                     \
    m.name = std::string(modname) + "." + #n; m.module = modno; m.kind = kindv;          \
    m.inst_slot = &s->istruct.n; m.mean_slot = &s->mstruct.n; ++hits;                    \
  }
#define X(n) TRY("bgc", 0, KC, bi, bm, n)
  OBGC_BGC_DIAG_KC(X)
#undef X
#define X(n) TRY("bgc", 0, KCA, bi, bm, n)
  OBGC_BGC_DIAG_KCA(X)
#undef X
#define X(n) TRY("bgc", 0, CA, bi, bm, n)
  OBGC_BGC_DIAG_CA(X)
#undef X
#define X(n) TRY("bgc", 0, C, bi, bm, n)
  OBGC_BGC_DIAG_C(X)
#undef X
#define X(n) TRY("dms", 1, KC, di, dm, n)
  OBGC_DMS_DIAG_KC(X)
#undef X
#define X(n) TRY("macros", 2, KC, mi, mm, n)
  OBGC_MACROS_DIAG_KC(X)
#undef X
#undef TRY
  return hits == 1;
}

// elements one member occupies in the set's own allocation
long long own_elems(const GridView& v, int kind) {
  const long long kc1 = v.at(v.nlev - 1, v.ncol > 0 ? v.ncol - 1 : 0) + 1;
  switch (kind) {
    case KC: return v.tiled ? 0 : kc1;
    case KCA: return v.tiled ? 0 : 3 * v.plane + kc1;
    case CA: return 3 * v.ldc + v.ncol;
    default: return v.ncol;
  }
}
long long pad32(long long n) { return (n + 31) & ~31LL; }

// number of values of a member in the reference's order
long long ref_elems(const GridView& v, int kind) {
  const long long kc = (long long)v.nlev * v.ncol;
  return kind == KC ? kc : (kind == KCA ? 4 * kc : (kind == CA ? 4LL * v.ncol : (long long)v.ncol));
}

// one member, device -> host, into the reference's X(k,column[,a]) resp. X(column[,a]) order
int fetch(obgc_diag_set* s, const Member& m, int which, double* out) {
  obgc_handle* h = s->h;
  const GridView& v = s->v;
  const double* src = which ? m.inst : m.mean;
  if (m.kind == C) {
    HIP_TRY(h, hipMemcpy(out, src, (size_t)v.ncol * sizeof(double), hipMemcpyDeviceToHost));
    return OBGC_OK;
  }
  if (m.kind == CA) {
    HIP_TRY(h, hipMemcpy2D(out, (size_t)v.ncol * sizeof(double), src, (size_t)v.ldc * sizeof(double), (size_t)v.ncol * sizeof(double), 4,
                           hipMemcpyDeviceToHost));
    return OBGC_OK;
  }
  const int planes = m.kind == KCA ? 4 : 1;
  const long long nblk = (v.ncol + 63) / 64;
  std::vector<double> tmp((size_t)(v.tiled ? nblk * 64 : v.ncol) * (size_t)v.nlev);
  for (int a = 0; a < planes; ++a) {
    const double* p = src + (long long)a * v.plane;
    double* o = out + (size_t)a * (size_t)v.ncol * (size_t)v.nlev;
    if (v.tiled) {
      // rows of 64 columns, one per (block, level), a level stride apart -> tmp[block][level][64]
      HIP_TRY(h, hipMemcpy2D(tmp.data(), 64 * sizeof(double), p, (size_t)v.sk * sizeof(double), 64 * sizeof(double),
                             (size_t)(nblk * v.nlev), hipMemcpyDeviceToHost));
      for (int c = 0; c < v.ncol; ++c)
        for (int k = 0; k < v.nlev; ++k) o[(size_t)c * v.nlev + k] = tmp[((size_t)(c >> 6) * v.nlev + k) * 64 + (c & 63)];
    } else {
      HIP_TRY(h, hipMemcpy2D(tmp.data(), (size_t)v.ncol * sizeof(double), p, (size_t)v.sk * sizeof(double), (size_t)v.ncol * sizeof(double),
                             (size_t)v.nlev, hipMemcpyDeviceToHost));
      for (int c = 0; c < v.ncol; ++c)
        for (int k = 0; k < v.nlev; ++k) o[(size_t)c * v.nlev + k] = tmp[(size_t)k * v.ncol + c];
    }
  }
  return OBGC_OK;
}

const Member* find(const obgc_diag_set* s, const char* name) {
  if (!name) return nullptr;
  static obgc_diag_set probe{};   // lookup only takes addresses inside it: what is wanted here is the canonical name
  Member m;
  if (!lookup(&probe, name, m)) return nullptr;
  for (const Member& x : s->members)
    if (x.name == m.name) return &x;
  return nullptr;
}

struct FileHeader {
  char magic[8];
  int32_t nlev, ncol, nrec, which;
  int32_t reserved[10];
};
static_assert(sizeof(FileHeader) == 64, "diagnostics file header is 64 bytes");
struct RecordHeader {
  char name[40];
  int32_t module, kind, planes, reserved;
  int64_t nelem;
};
static_assert(sizeof(RecordHeader) == 64, "diagnostics record header is 64 bytes");

}  // namespace

extern "C" int obgc_diag_set_rows(int nnames, const char* const* names, int* rows, int* vectors, char* err, int errlen) {
  if (nnames < 0 || (nnames > 0 && !names)) return OBGC_ERR_INVALID;
  obgc_diag_set probe{};
  int r = 0, c = 0;
  for (int i = 0; i < nnames; ++i) {
    Member m;
    if (!names[i] || !lookup(&probe, names[i], m)) {
      set_err(err, errlen, std::string("unknown or ambiguous diagnostic name: ") + (names[i] ? names[i] : "(null)"));
      return OBGC_ERR_INVALID;
    }
    if (m.kind == KC) r += 1;
    else if (m.kind == KCA) r += 4;
    else c += (m.kind == CA ? 4 : 1);
  }
  if (rows) *rows = r;
  if (vectors) *vectors = c;
  return OBGC_OK;
}

extern "C" int obgc_diag_set_create(obgc_handle* h, const obgc_grid* grid, int nnames, const char* const* names, double* inst_rows,
                                    double* mean_rows, obgc_diag_set** out) {
  if (!h || !grid || !out || nnames < 0 || (nnames > 0 && !names)) return OBGC_ERR_INVALID;
  if (grid->memspace != OBGC_MEM_DEVICE || grid->layout == OBGC_LAYOUT_KFAST)
    return fail(h, OBGC_ERR_UNSUPPORTED, "obgc_diag_set_create: diagnostics stay on the device in the native or the column-block layout");
  const bool tiled = grid->layout == OBGC_LAYOUT_TILED;
  if (tiled && (grid->ld < 64 || (grid->plane != 0 && grid->plane != 64)))
    return fail(h, OBGC_ERR_INVALID, "obgc_diag_set_create: a column-block grid needs its level stride and plane = 64");
  obgc_diag_set* s = new obgc_diag_set{};
  s->h = h;
  s->grid = *grid;
  s->v = make_view(*grid);
  int rows = 0;
  for (int i = 0; i < nnames; ++i) {
    Member m;
    if (!names[i] || !lookup(s, names[i], m)) {
      const std::string bad = names[i] ? names[i] : "(null)";
      delete s;
      return fail(h, OBGC_ERR_INVALID, "obgc_diag_set_create: unknown or ambiguous diagnostic name: " + bad);
    }
    bool dup = false;
    for (const Member& x : s->members) dup = dup || x.name == m.name;
    if (dup) continue;
    s->members.push_back(m);
    s->nmod[m.module]++;
    if (m.kind == KC) rows += 1;
    if (m.kind == KCA) rows += 4;
  }
  if (tiled && rows > 0 && (!inst_rows || !mean_rows)) {
    delete s;
    return fail(h, OBGC_ERR_INVALID, "obgc_diag_set_create: in the column-block layout the caller places the " + std::to_string(rows) +
                                         " rows of the instantaneous fields and of the means in its arena (inst_rows, mean_rows)");
  }
  long long own = 0;
  for (const Member& m : s->members) own += 2 * pad32(own_elems(s->v, m.kind));
  if (hipSetDevice(h->device) != hipSuccess) { delete s; return fail(h, OBGC_ERR_HIP, "hipSetDevice failed"); }
  if (own > 0) {
    if (hipMalloc((void**)&s->dev, (size_t)own * sizeof(double)) != hipSuccess) {
      delete s;
      return fail(h, OBGC_ERR_ALLOC, "obgc_diag_set_create: device allocation of " + std::to_string(own * 8) + " bytes failed");
    }
    if (hipMemset(s->dev, 0, (size_t)own * sizeof(double)) != hipSuccess) {
      (void)hipFree(s->dev);
      delete s;
      return fail(h, OBGC_ERR_HIP, "obgc_diag_set_create: hipMemset failed");
    }
  }
  double* p = s->dev;
  long long row = 0;
  for (Member& m : s->members) {
    if (tiled && (m.kind == KC || m.kind == KCA)) {
      m.inst = inst_rows + row * 64;
      m.mean = mean_rows + row * 64;
      row += m.kind == KCA ? 4 : 1;
    } else {
      const long long n = pad32(own_elems(s->v, m.kind));
      m.inst = p;
      m.mean = p + n;
      p += 2 * n;
    }
    *m.inst_slot = m.inst;
    *m.mean_slot = m.mean;
  }
  *out = s;
  return OBGC_OK;
}

extern "C" int obgc_diag_set_destroy(obgc_diag_set* s) {
  if (!s) return OBGC_OK;
  if (s->dev) {
    (void)hipSetDevice(s->h->device);
    (void)hipFree(s->dev);
  }
  delete s;
  return OBGC_OK;
}

extern "C" const obgc_bgc_diag* obgc_diag_set_bgc(const obgc_diag_set* s) { return (s && s->nmod[0]) ? &s->bi : nullptr; }
extern "C" const obgc_dms_diag* obgc_diag_set_dms(const obgc_diag_set* s) { return (s && s->nmod[1]) ? &s->di : nullptr; }
extern "C" const obgc_macros_diag* obgc_diag_set_macros(const obgc_diag_set* s) { return (s && s->nmod[2]) ? &s->mi : nullptr; }
extern "C" int obgc_diag_set_count(const obgc_diag_set* s) { return s ? (int)s->members.size() : 0; }

extern "C" int obgc_diag_set_accumulate(obgc_diag_set* s, double weight, int reset) {
  if (!s) return OBGC_ERR_INVALID;
  int rc = OBGC_OK;
  if (s->nmod[0]) rc = obgc_bgc_diag_accumulate(s->h, &s->grid, &s->bi, &s->bm, weight, reset);
  if (rc == OBGC_OK && s->nmod[1]) rc = obgc_dms_diag_accumulate(s->h, &s->grid, &s->di, &s->dm, weight, reset);
  if (rc == OBGC_OK && s->nmod[2]) rc = obgc_macros_diag_accumulate(s->h, &s->grid, &s->mi, &s->mm, weight, reset);
  return rc;
}

extern "C" long long obgc_diag_set_elems(const obgc_diag_set* s, const char* name) {
  const Member* m = s ? find(s, name) : nullptr;
  return m ? ref_elems(s->v, m->kind) : -1;
}

extern "C" int obgc_diag_set_download(obgc_diag_set* s, const char* name, int which, double* host, long long host_len) {
  if (!s || !host) return OBGC_ERR_INVALID;
  const Member* m = find(s, name);
  if (!m) return fail(s->h, OBGC_ERR_INVALID, std::string("obgc_diag_set_download: not in the selection: ") + (name ? name : "(null)"));
  if (host_len != ref_elems(s->v, m->kind)) return fail(s->h, OBGC_ERR_INVALID, "obgc_diag_set_download: wrong length for " + m->name);
  HIP_TRY(s->h, hipSetDevice(s->h->device));
  int rc = obgc_sync(s->h);
  if (rc != OBGC_OK) return rc;
  return fetch(s, *m, which, host);
}

extern "C" int obgc_diag_set_write(obgc_diag_set* s, const char* path, int which) {
  if (!s || !path) return OBGC_ERR_INVALID;
  HIP_TRY(s->h, hipSetDevice(s->h->device));
  int rc = obgc_sync(s->h);
  if (rc != OBGC_OK) return rc;
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(s->h, OBGC_ERR_INVALID, std::string("obgc_diag_set_write: cannot open ") + path);
  FileHeader fh;
  std::memset(&fh, 0, sizeof fh);
  std::memcpy(fh.magic, "OBGCDIA1", 8);
  fh.nlev = s->v.nlev; fh.ncol = s->v.ncol; fh.nrec = (int32_t)s->members.size(); fh.which = which ? 1 : 0;
  bool ok = std::fwrite(&fh, sizeof fh, 1, f) == 1;
  std::vector<double> buf;
  for (const Member& m : s->members) {
    if (!ok) break;
    RecordHeader rh;
    std::memset(&rh, 0, sizeof rh);
    std::snprintf(rh.name, sizeof rh.name, "%s", m.name.c_str());
    rh.module = m.module; rh.kind = m.kind; rh.planes = (m.kind == KCA || m.kind == CA) ? 4 : 1;
    rh.nelem = ref_elems(s->v, m.kind);
    buf.resize((size_t)rh.nelem);
    rc = fetch(s, m, which, buf.data());
    if (rc != OBGC_OK) { std::fclose(f); return rc; }
    ok = std::fwrite(&rh, sizeof rh, 1, f) == 1 && std::fwrite(buf.data(), sizeof(double), buf.size(), f) == buf.size();
  }
  ok = (std::fclose(f) == 0) && ok;
  return ok ? OBGC_OK : fail(s->h, OBGC_ERR_INVALID, std::string("obgc_diag_set_write: short write to ") + path);
}
