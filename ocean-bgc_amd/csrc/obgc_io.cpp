// Parameter text files and the restart file of the only persistent state.
// Host-only code (no HIP): lets stand-alone drivers, benchmarks and regression suites run the engine with an
// explicit, versioned parameter set and carry PH_PREV_* / surface_pH* across runs.  The reference has no file
// format of its own (its parameters are Fortran module variables set by *_parms_init and edited by the host).
//
// Parameter file: UTF-8 text, '#' comments, sections [bgc], [autotroph N] (N = 1..4), [dms], [macros]; one
// `name = value` per line; names are the reference's (include/obgc_fields.h).  Reals are written as C99 hex floats
// (exact round trip) and read as decimal or hex, both in the "C" locale whatever the process locale is.  A file only
// overrides the keys it lists: the structs must hold defaults (obgc_*_params_init) before reading.  Unknown keys or
// sections are errors, not warnings.  The per-autotroph tracer slots (Chl_ind, C_ind, Fe_ind, Si_ind, CaCO3_ind) are
// derived quantities (BGC_mod.F90:271-321): they are written for the reader's information and re-derived after a read.
//
// A [diagnostics] section selects diagnostics by the reference's member names (BGC_parms.F90:192-321, DMS_parms.F90:125-154,
// MACROS_parms.F90:105-113; the diag_ prefix is optional): `bgc = CO3, pH_3D, photoC`, `dms = ...`, `macros = ...`; keys may
// repeat.  obgc_params_read_text checks the names and otherwise ignores the section; obgc_diag_names_read_text returns them.
//
// Diagnostics file (written by obgc_diag_set_write): little-endian binary, 64-byte header
//   char magic[8] = "OBGCDIA1"; int32 nlev, ncol, nrec, which (0 means, 1 instantaneous); int32 reserved[10];
// then nrec records: a 64-byte record header { char name[40] ("bgc.CO3"); int32 module, kind, planes, reserved; int64 nelem }
// followed by nelem fp64 values in the reference's order X(k,column[,autotroph]) resp. X(column[,autotroph]).
//
// Restart file: little-endian binary, 64-byte header
//   char magic[8] = "OBGCRST1"; int32 nlev, ncol; int32 order; int32 reserved[11];
// followed by four dense fp64 arrays: PH_PREV_3D(nlev,ncol), PH_PREV_ALT_CO2_3D(nlev,ncol), surface_pH(ncol),
// surface_pH_alt_co2(ncol).  `order` = 0: level-fastest, the reference's X(k,column) -- the only order ever written; any
// other value is rejected on read.  Callers that hold the state column-fastest (the engine's native layout) use the
// *_grid variants, which convert while writing / reading.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <locale.h>
#include <string>
#include <vector>

#include "obgc.h"

namespace {

void set_err(char* err, int errlen, const std::string& msg) {
  if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
}

std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

// the "C" locale, whatever the host program has set: a decimal comma must neither be written nor expected
locale_t c_locale() {
  static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
  return loc;
}

bool parse_real(const std::string& v, double& out) {
  errno = 0;
  char* end = nullptr;
  out = strtod_l(v.c_str(), &end, c_locale());
  return end != v.c_str() && *end == '\0' && errno == 0;
}

// "%a" of a double with a '.' as the radix character in every locale
std::string hexfloat(double x) {
  char buf[64];
  locale_t old = uselocale(c_locale());
  std::snprintf(buf, sizeof buf, "%a", x);
  uselocale(old);
  return buf;
}

bool parse_int(const std::string& v, int& out) {
  errno = 0;
  char* end = nullptr;
  long x = std::strtol(v.c_str(), &end, 10);
  out = (int)x;
  return end != v.c_str() && *end == '\0' && errno == 0;
}

bool set_bgc(obgc_bgc_params* p, const std::string& k, const std::string& v) {
#define X(n) if (k == #n) return parse_real(v, p->n);
  OBGC_BGC_PARMS(X)
#undef X
  if (k == "T0_Kelvin_BGC") return parse_real(v, p->T0_Kelvin_BGC);
  for (int i = 0; i < 4; ++i) {
    if (k == "parm_scalelen_z(" + std::to_string(i + 1) + ")") return parse_real(v, p->parm_scalelen_z[i]);
    if (k == "parm_scalelen_vals(" + std::to_string(i + 1) + ")") return parse_real(v, p->parm_scalelen_vals[i]);
  }
  return false;
}

bool set_autotroph(obgc_autotroph* a, const std::string& k, const std::string& v) {
#define X(n) if (k == #n) return parse_int(v, a->n);
  OBGC_AUTOTROPH_INTS(X)
#undef X
#define X(n) if (k == #n) return parse_real(v, a->n);
  OBGC_AUTOTROPH_REALS(X)
#undef X
  return false;
}

bool set_dms(obgc_dms_params* p, const std::string& k, const std::string& v) {
#define X(n) if (k == #n) return parse_real(v, p->n);
  OBGC_DMS_PARMS(X)
#undef X
  return false;
}

bool set_macros(obgc_macros_params* p, const std::string& k, const std::string& v) {
#define X(n) if (k == #n) return parse_real(v, p->n);
  OBGC_MACROS_PARMS(X)
#undef X
  return false;
}

// module (0 BGC, 1 DMS, 2 MACROS) of a diagnostic name, -1 if there is none; `mod` restricts the search
int diag_module(const std::string& mod, std::string nm) {
  if (nm.rfind("diag_", 0) == 0) nm = nm.substr(5);
  int found = -1, hits = 0;
#define X(n) if ((mod.empty() || mod == "bgc") && nm == #n) { found = 0; ++hits; }
  OBGC_BGC_DIAG_KC(X) OBGC_BGC_DIAG_KCA(X) OBGC_BGC_DIAG_CA(X) OBGC_BGC_DIAG_C(X)
#undef X
#define X(n) if ((mod.empty() || mod == "dms") && nm == #n) { found = 1; ++hits; }
  OBGC_DMS_DIAG_KC(X)
#undef X
#define X(n) if ((mod.empty() || mod == "macros") && nm == #n) { found = 2; ++hits; }
  OBGC_MACROS_DIAG_KC(X)
#undef X
  return hits == 1 ? found : -1;
}

// `bgc = a, b, c` of a [diagnostics] section -> "bgc.a", "bgc.b", ...; false on an unknown name
bool diag_list(const std::string& mod, const std::string& v, std::vector<std::string>* out, std::string& bad) {
  if (mod != "bgc" && mod != "dms" && mod != "macros") { bad = mod; return false; }
  size_t pos = 0;
  while (pos <= v.size()) {
    size_t comma = v.find(',', pos);
    if (comma == std::string::npos) comma = v.size();
    std::string nm = trim(v.substr(pos, comma - pos));
    pos = comma + 1;
    if (nm.empty()) continue;
    if (nm.rfind("diag_", 0) == 0) nm = nm.substr(5);
    if (diag_module(mod, nm) < 0) { bad = nm; return false; }
    if (out) out->push_back(mod + "." + nm);
  }
  return true;
}

struct DiagFileHeader {
  char magic[8];
  int32_t nlev, ncol, nrec, which;
  int32_t reserved[10];
};
struct DiagRecordHeader {
  char name[40];
  int32_t module, kind, planes, reserved;
  int64_t nelem;
};
static_assert(sizeof(DiagFileHeader) == 64 && sizeof(DiagRecordHeader) == 64, "diagnostics file headers are 64 bytes");

struct RestartHeader {
  char magic[8];
  int32_t nlev, ncol;
  int32_t order;          // 0 = level-fastest X(k,column); nothing else is defined
  int32_t reserved[11];
};
static_assert(sizeof(RestartHeader) == 64, "restart header is 64 bytes");
const char kMagic[8] = {'O', 'B', 'G', 'C', 'R', 'S', 'T', '1'};

}  // namespace

extern "C" int obgc_params_write_text(const char* path, const obgc_bgc_params* bgc, const obgc_dms_params* dms,
                                      const obgc_macros_params* macros) {
  if (!path) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "w");
  if (!f) return OBGC_ERR_INVALID;
  std::fprintf(f, "# obgc parameters v1 -- names as in the reference's *_parms modules; reals as C99 hex floats\n");
  if (bgc) {
    std::fprintf(f, "[bgc]\n");
#define X(n) std::fprintf(f, "%s = %s\n", #n, hexfloat(bgc->n).c_str());
    OBGC_BGC_PARMS(X)
#undef X
    std::fprintf(f, "T0_Kelvin_BGC = %s\n", hexfloat(bgc->T0_Kelvin_BGC).c_str());
    for (int i = 0; i < 4; ++i) std::fprintf(f, "parm_scalelen_z(%d) = %s\n", i + 1, hexfloat(bgc->parm_scalelen_z[i]).c_str());
    for (int i = 0; i < 4; ++i) std::fprintf(f, "parm_scalelen_vals(%d) = %s\n", i + 1, hexfloat(bgc->parm_scalelen_vals[i]).c_str());
    for (int a = 0; a < OBGC_AUTOTROPH_CNT; ++a) {
      std::fprintf(f, "[autotroph %d]\n", a + 1);
#define X(n) std::fprintf(f, "%s = %d\n", #n, bgc->autotrophs[a].n);
      OBGC_AUTOTROPH_INTS(X)
#undef X
#define X(n) std::fprintf(f, "%s = %s\n", #n, hexfloat(bgc->autotrophs[a].n).c_str());
      OBGC_AUTOTROPH_REALS(X)
#undef X
    }
  }
  if (dms) {
    std::fprintf(f, "[dms]\n");
#define X(n) std::fprintf(f, "%s = %s\n", #n, hexfloat(dms->n).c_str());
    OBGC_DMS_PARMS(X)
#undef X
  }
  if (macros) {
    std::fprintf(f, "[macros]\n");
#define X(n) std::fprintf(f, "%s = %s\n", #n, hexfloat(macros->n).c_str());
    OBGC_MACROS_PARMS(X)
#undef X
  }
  return std::fclose(f) == 0 ? OBGC_OK : OBGC_ERR_INVALID;
}

extern "C" int obgc_params_read_text(const char* path, obgc_bgc_params* bgc, obgc_dms_params* dms, obgc_macros_params* macros,
                                     char* err, int errlen) {
  if (!path) return OBGC_ERR_INVALID;
  std::ifstream f(path);
  if (!f) { set_err(err, errlen, std::string("cannot open ") + path); return OBGC_ERR_INVALID; }
  enum { NONE, BGC, AUTO, DMS, MACROS, DIAGS } sec = NONE;
  int autotroph = 0, lineno = 0;
  int rc = OBGC_OK;
  std::string line;
  while (rc == OBGC_OK && std::getline(f, line)) {   // whole lines of any length
    ++lineno;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    line = trim(line);
    if (line.empty()) continue;
    auto bad = [&](const std::string& why) {
      set_err(err, errlen, std::string(path) + ":" + std::to_string(lineno) + ": " + why);
      rc = OBGC_ERR_INVALID;
    };
    if (line.front() == '[') {
      if (line.back() != ']') { bad("unterminated section header"); break; }
      const std::string name = trim(line.substr(1, line.size() - 2));
      if (name == "bgc") sec = BGC;
      else if (name == "dms") sec = DMS;
      else if (name == "macros") sec = MACROS;
      else if (name == "diagnostics") sec = DIAGS;
      else if (name.rfind("autotroph", 0) == 0 && parse_int(trim(name.substr(9)), autotroph) && autotroph >= 1 &&
               autotroph <= OBGC_AUTOTROPH_CNT) sec = AUTO;
      else bad("unknown section [" + name + "]");
      continue;
    }
    const size_t eq = line.find('=');
    if (eq == std::string::npos) { bad("expected name = value"); break; }
    const std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
    bool ok = false;
    switch (sec) {
      case BGC: ok = !bgc || set_bgc(bgc, k, v); break;
      case AUTO: ok = !bgc || set_autotroph(&bgc->autotrophs[autotroph - 1], k, v); break;
      case DMS: ok = !dms || set_dms(dms, k, v); break;
      case MACROS: ok = !macros || set_macros(macros, k, v); break;
      case DIAGS: { std::string which; ok = diag_list(k, v, nullptr, which); } break;
      case NONE: bad("key before any section"); continue;
    }
    if (!ok) bad("unknown key or malformed value: " + k + " = " + v);
  }
  // kSiO3, imp_calcifier / exp_calcifier decide which autotroph owns a Si / CaCO3 pool (BGC_mod.F90:303-320): the
  // slots the kernels branch on always follow from the table just read, never from a stale *_ind line of the file
  if (rc == OBGC_OK && bgc) obgc_bgc_params_derive(bgc);
  return rc;
}

extern "C" int obgc_restart_write(const char* path, int nlev, int ncol, const double* ph_prev, const double* ph_prev_alt,
                                  const double* surface_pH, const double* surface_pH_alt) {
  if (!path || nlev < 1 || ncol < 0 || (ncol > 0 && (!ph_prev || !ph_prev_alt || !surface_pH || !surface_pH_alt)))
    return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "wb");
  if (!f) return OBGC_ERR_INVALID;
  RestartHeader h;
  std::memset(&h, 0, sizeof h);
  std::memcpy(h.magic, kMagic, 8);
  h.nlev = nlev;
  h.ncol = ncol;
  const size_t n3 = (size_t)nlev * (size_t)ncol, n2 = (size_t)ncol;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && std::fwrite(ph_prev, sizeof(double), n3, f) == n3 && std::fwrite(ph_prev_alt, sizeof(double), n3, f) == n3;
  ok = ok && std::fwrite(surface_pH, sizeof(double), n2, f) == n2 && std::fwrite(surface_pH_alt, sizeof(double), n2, f) == n2;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? OBGC_OK : OBGC_ERR_INVALID;
}

extern "C" int obgc_restart_read_header(const char* path, int* nlev, int* ncol) {
  if (!path || !nlev || !ncol) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) return OBGC_ERR_INVALID;
  RestartHeader h;
  const bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, kMagic, 8) == 0 && h.nlev >= 1 && h.ncol >= 0 &&
                  h.order == 0;
  std::fclose(f);
  if (!ok) return OBGC_ERR_INVALID;
  *nlev = h.nlev;
  *ncol = h.ncol;
  return OBGC_OK;
}

extern "C" int obgc_restart_read(const char* path, int nlev, int ncol, double* ph_prev, double* ph_prev_alt, double* surface_pH,
                                 double* surface_pH_alt) {
  int fl = 0, fc = 0;
  int rc = obgc_restart_read_header(path, &fl, &fc);
  if (rc != OBGC_OK) return rc;
  if (fl != nlev || fc != ncol || (ncol > 0 && (!ph_prev || !ph_prev_alt || !surface_pH || !surface_pH_alt))) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) return OBGC_ERR_INVALID;
  const size_t n3 = (size_t)nlev * (size_t)ncol, n2 = (size_t)ncol;
  bool ok = std::fseek(f, (long)sizeof(RestartHeader), SEEK_SET) == 0;
  ok = ok && std::fread(ph_prev, sizeof(double), n3, f) == n3 && std::fread(ph_prev_alt, sizeof(double), n3, f) == n3;
  ok = ok && std::fread(surface_pH, sizeof(double), n2, f) == n2 && std::fread(surface_pH_alt, sizeof(double), n2, f) == n2;
  // the file must end here
  ok = ok && std::fgetc(f) == EOF;
  std::fclose(f);
  return ok ? OBGC_OK : OBGC_ERR_INVALID;
}

namespace {

// (k,column) element of a host array described by an obgc_grid
inline size_t grid_at(const obgc_grid& g, int k, int c) {
  if (g.layout == OBGC_LAYOUT_KFAST) return (size_t)c * (size_t)(g.ld > 0 ? g.ld : g.nlev) + (size_t)k;
  return (size_t)k * (size_t)(g.ld > 0 ? g.ld : g.ncol) + (size_t)c;
}

bool grid_host_ok(const obgc_grid* g) {
  if (!g || g->nlev < 1 || g->ncol < 0 || g->memspace != OBGC_MEM_HOST) return false;
  if (g->layout != OBGC_LAYOUT_KFAST && g->layout != OBGC_LAYOUT_COLFAST) return false;
  const long long minld = g->layout == OBGC_LAYOUT_KFAST ? g->nlev : g->ncol;
  return g->ld == 0 || g->ld >= minld;
}

}  // namespace

extern "C" int obgc_restart_write_grid(const char* path, const obgc_grid* g, const double* ph_prev, const double* ph_prev_alt,
                                       const double* surface_pH, const double* surface_pH_alt) {
  if (!grid_host_ok(g) || (g->ncol > 0 && (!ph_prev || !ph_prev_alt))) return OBGC_ERR_INVALID;
  const size_t n3 = (size_t)g->nlev * (size_t)g->ncol;
  std::vector<double> a(n3), b(n3);
  for (int c = 0; c < g->ncol; ++c)
    for (int k = 0; k < g->nlev; ++k) {
      a[(size_t)c * g->nlev + k] = ph_prev[grid_at(*g, k, c)];
      b[(size_t)c * g->nlev + k] = ph_prev_alt[grid_at(*g, k, c)];
    }
  return obgc_restart_write(path, g->nlev, g->ncol, a.data(), b.data(), surface_pH, surface_pH_alt);
}

extern "C" int obgc_restart_read_grid(const char* path, const obgc_grid* g, double* ph_prev, double* ph_prev_alt, double* surface_pH,
                                      double* surface_pH_alt) {
  if (!grid_host_ok(g) || (g->ncol > 0 && (!ph_prev || !ph_prev_alt))) return OBGC_ERR_INVALID;
  const size_t n3 = (size_t)g->nlev * (size_t)g->ncol;
  std::vector<double> a(n3), b(n3);
  const int rc = obgc_restart_read(path, g->nlev, g->ncol, a.data(), b.data(), surface_pH, surface_pH_alt);
  if (rc != OBGC_OK) return rc;
  for (int c = 0; c < g->ncol; ++c)
    for (int k = 0; k < g->nlev; ++k) {
      ph_prev[grid_at(*g, k, c)] = a[(size_t)c * g->nlev + k];
      ph_prev_alt[grid_at(*g, k, c)] = b[(size_t)c * g->nlev + k];
    }
  return OBGC_OK;
}

// Static contiguous column ranges for `nparts` devices.  Columns are independent, so any cut is
// legal; with ragged bathymetry the cost of a range is its number of ACTIVE cells, not of columns.
//   balance = 0: equal column counts;  1: equal sums of max(kmax, 0) (prefix sum, cut at the first column whose
//   prefix reaches p/nparts of the total).
extern "C" int obgc_partition_columns(const int* kmax, long long ncol, int nparts, int balance, long long* bounds) {
  if (!bounds || nparts < 1 || ncol < 0 || (balance != 0 && balance != 1) || (balance == 1 && ncol > 0 && !kmax))
    return OBGC_ERR_INVALID;
  long long total = 0;
  if (balance == 1)
    for (long long c = 0; c < ncol; ++c) total += kmax[c] > 0 ? kmax[c] : 0;
  bounds[0] = 0;
  bounds[nparts] = ncol;
  if (balance == 0 || total == 0) {
    for (int p = 1; p < nparts; ++p) bounds[p] = (long long)(((__int128)ncol * p) / nparts);
    return OBGC_OK;
  }
  long long prefix = 0, c = 0;
  for (int p = 1; p < nparts; ++p) {
    const long long goal = (long long)(((__int128)total * p + nparts - 1) / nparts);   // ceil(total * p / nparts)
    while (c < ncol && prefix < goal) { prefix += kmax[c] > 0 ? kmax[c] : 0; ++c; }
    bounds[p] = c;
  }
  return OBGC_OK;
}

// ------------------------------------------------------------------ diagnostics: names from a parameter file, files
extern "C" int obgc_diag_names_read_text(const char* path, char* names, int nameslen, int* count, char* err, int errlen) {
  if (!path || !count || (nameslen > 0 && !names)) return OBGC_ERR_INVALID;
  std::ifstream f(path);
  if (!f) { set_err(err, errlen, std::string("cannot open ") + path); return OBGC_ERR_INVALID; }
  std::vector<std::string> out;
  bool in_sec = false;
  int lineno = 0;
  std::string line;
  while (std::getline(f, line)) {
    ++lineno;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    line = trim(line);
    if (line.empty()) continue;
    if (line.front() == '[') { in_sec = trim(line.substr(1, line.size() > 1 ? line.size() - 2 : 0)) == "diagnostics"; continue; }
    if (!in_sec) continue;
    const size_t eq = line.find('=');
    std::string bad;
    if (eq == std::string::npos || !diag_list(trim(line.substr(0, eq)), trim(line.substr(eq + 1)), &out, bad)) {
      set_err(err, errlen, std::string(path) + ":" + std::to_string(lineno) + ": unknown diagnostic or module: " + bad);
      return OBGC_ERR_INVALID;
    }
  }
  size_t need = 0;
  for (const std::string& n : out) need += n.size() + 1;
  *count = (int)out.size();
  if ((size_t)(nameslen > 0 ? nameslen : 0) < need + 1) {
    set_err(err, errlen, "name buffer too small: " + std::to_string(need + 1) + " bytes needed");
    return nameslen > 0 ? OBGC_ERR_INVALID : OBGC_OK;      // nameslen = 0: only counting
  }
  char* p = names;
  for (const std::string& n : out) { std::memcpy(p, n.c_str(), n.size() + 1); p += n.size() + 1; }
  *p = '\0';
  return OBGC_OK;
}

extern "C" int obgc_diag_file_info(const char* path, int* nlev, int* ncol, int* nrec, int* which) {
  if (!path) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) return OBGC_ERR_INVALID;
  DiagFileHeader h;
  const bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "OBGCDIA1", 8) == 0 && h.nlev >= 1 && h.ncol >= 0 && h.nrec >= 0;
  std::fclose(f);
  if (!ok) return OBGC_ERR_INVALID;
  if (nlev) *nlev = h.nlev;
  if (ncol) *ncol = h.ncol;
  if (nrec) *nrec = h.nrec;
  if (which) *which = h.which;
  return OBGC_OK;
}

namespace {
// positions the file at the data of record `index` (name == NULL) or of the record called `name`; fills `rh`
bool seek_record(FILE* f, int index, const char* name, DiagRecordHeader& rh) {
  DiagFileHeader h;
  if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "OBGCDIA1", 8) != 0) return false;
  for (int i = 0; i < h.nrec; ++i) {
    if (std::fread(&rh, sizeof rh, 1, f) != 1 || rh.nelem < 0) return false;
    rh.name[sizeof rh.name - 1] = '\0';
    if (name ? std::strcmp(rh.name, name) == 0 : i == index) return true;
    if (std::fseek(f, (long)(rh.nelem * (int64_t)sizeof(double)), SEEK_CUR) != 0) return false;
  }
  return false;
}
}  // namespace

extern "C" int obgc_diag_file_record(const char* path, int index, char* name, int namelen, int* module, int* kind, long long* nelem) {
  if (!path || index < 0) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) return OBGC_ERR_INVALID;
  DiagRecordHeader rh;
  const bool ok = seek_record(f, index, nullptr, rh);
  std::fclose(f);
  if (!ok) return OBGC_ERR_INVALID;
  if (name && namelen > 0) std::snprintf(name, (size_t)namelen, "%s", rh.name);
  if (module) *module = rh.module;
  if (kind) *kind = rh.kind;
  if (nelem) *nelem = rh.nelem;
  return OBGC_OK;
}

extern "C" int obgc_diag_file_read(const char* path, const char* name, double* out, long long len) {
  if (!path || !name || !out) return OBGC_ERR_INVALID;
  FILE* f = std::fopen(path, "rb");
  if (!f) return OBGC_ERR_INVALID;
  DiagRecordHeader rh;
  std::string full = name;
  if (full.find('.') == std::string::npos) {           // a bare member name: the module follows from it
    const int m = diag_module("", full);
    if (m >= 0) full = std::string(m == 0 ? "bgc." : (m == 1 ? "dms." : "macros.")) + (full.rfind("diag_", 0) == 0 ? full.substr(5) : full);
  }
  bool ok = seek_record(f, -1, full.c_str(), rh) && rh.nelem == len;
  ok = ok && std::fread(out, sizeof(double), (size_t)len, f) == (size_t)len;
  std::fclose(f);
  return ok ? OBGC_OK : OBGC_ERR_INVALID;
}
