"""Deterministic synthetic column generator (host / numpy twin of csrc/synth.hip).

Every value is a closed-form function of the
GLOBAL column index c, the level k and a field id, through splitmix64 hashes and
+,-,*,/ only (no transcendentals), so the numpy version here and the on-device
version produce bit-identical inputs, any shard of columns can be generated
independently, and the CPU baseline can be fed exactly what the GPU saw.

Branch coverage intended by the recipe (reference line numbers, BGC_mod.F90):
zero_mask 826-844, Fe-limited quotas 866-887, both KPAR regimes 915-919,
diaz temp_thres / phaeo quasi-mmrt both hemispheres 1074-1092, CaCO3 cold &
bloom modifiers 1259-1267, dark columns (nitrification 1554-1558), NO3==0
(1572), O2 bins (2311-2315), dust==0 (2081), all scalelength bins (2273-2286),
bottom-cell branches (2522-2631), land (kmax=0) and single-level columns.
"""
import numpy as np

SEED = np.uint64(0x0B6C2026)
_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)
_CELL_SALT = np.uint64(0x5DEECE66D1234567)

BGC_TRACERS = [
    "po4", "no3", "sio3", "nh4", "fe", "o2", "dic", "dic_alt_co2", "alk", "doc",
    "don", "dofe", "dop", "dopr", "donr", "zooC", "spC", "spChl", "spFe",
    "spCaCO3", "diatC", "diatChl", "diatFe", "diatSi", "phaeoC", "phaeoChl",
    "phaeoFe", "diazC", "diazChl", "diazFe",
]
DMS_TRACERS = [
    "dms", "dmsp", "no3", "doc", "zooC", "spC", "spCaCO3", "diatC", "diazC",
    "phaeoC", "spChl", "diatChl", "diazChl", "phaeoChl",
]
MACROS_TRACERS = ["prot", "poly", "lip", "zooC", "spC", "diatC", "diazC", "phaeoC"]


def splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _u01(h):
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def ucol(c, j):
    """uniform [0,1) for global column index array c (uint64) and stream id j < 64"""
    return _u01(splitmix64(SEED ^ (c * np.uint64(64) + np.uint64(j))))


def ucell(c, k, j):
    key = ((c * np.uint64(256) + k) * np.uint64(64) + np.uint64(j))
    return _u01(splitmix64(SEED ^ _CELL_SALT ^ key))


def grid(nz):
    k = np.arange(nz, dtype=np.float64)
    dz = 1000.0 * (1.0 + 0.05 * k)
    zbot = 1000.0 * ((k + 1.0) + 0.025 * (k * (k + 1.0)))
    zmid = zbot - 0.5 * dz
    return dz, zmid, zbot


def kmax_of(c, nz):
    c = np.asarray(c, dtype=np.int64)
    km = np.maximum(1, nz - (c % 7))
    km = np.where(c % 1000 == 999, 0, km)
    km = np.where(c % 1000 == 499, 1, km)
    return km.astype(np.int32)


def generate(ncol, nz, col0=0):
    """Returns a dict of named host arrays for columns [col0, col0+ncol).

    Level-dependent arrays have shape (ncol, nz) (C order == Fortran X(k,column)).
    """
    c = (np.arange(ncol, dtype=np.uint64) + np.uint64(col0))
    cc = c[:, None]
    kk = np.arange(nz, dtype=np.uint64)[None, :]
    dz1, zmid1, zbot1 = grid(nz)
    dz = np.broadcast_to(dz1, (ncol, nz)).copy()
    zmid = np.broadcast_to(zmid1, (ncol, nz)).copy()
    zbot = np.broadcast_to(zbot1, (ncol, nz)).copy()
    kmax = kmax_of(c, nz)

    def U(j):
        return ucol(c, j)[:, None]

    out = {"dz": dz, "zmid": zmid, "zbot": zbot, "kmax": kmax}
    out["lat"] = (-90.0 + 180.0 * ucol(c, 0))
    s = zmid / (zmid + 3.0e4)           # 0 at the surface -> 1 at depth
    q = 1.0 / (1.0 + zmid / 8.0e3)
    e = q * q                           # surface-intensified
    r = 1.0 / (1.0 + zmid / 4.0e4)
    u1 = U(1)
    out["temp"] = -1.8 + (4.0 + 26.0 * (u1 * u1)) * r
    out["salt"] = np.broadcast_to(33.0 + 3.0 * U(2), (ncol, nz)).copy()
    u3 = ucol(c, 3)
    out["sw"] = np.where(u3 < 0.1, 20.0 * u3, 300.0 * u3)
    u4 = ucol(c, 4)
    out["dust"] = np.where(u4 < 0.02, 0.0, 1.0e-12 * (0.1 + u4))
    kidx = np.arange(nz, dtype=np.int32)[None, :]
    out["fesed"] = np.where(kidx == (kmax[:, None] - 1), 2.3e-6 * U(14), 0.0)

    t = {}
    t["po4"] = (0.3 + 2.0 * s) * (0.5 + U(5))
    u6 = U(6)
    t["no3"] = np.where(u6 < 0.01, 0.0, (2.0 + 30.0 * s) * (0.5 + u6))
    t["sio3"] = (3.0 + 80.0 * s) * (0.5 + U(7))
    t["nh4"] = (0.2 * e) * (0.5 + U(8))
    t["fe"] = (2.0e-4 + 7.0e-4 * s) * (0.5 + U(9))
    u11 = U(11)
    o2f = np.where(u11 < 0.02, 0.01, np.where(u11 < 0.07, 0.12, 1.0))
    t["o2"] = ((250.0 - 220.0 * s) * (0.5 + U(10))) * o2f
    u12 = U(12)
    t["dic"] = (2000.0 + 250.0 * s) * (0.95 + 0.1 * u12)
    t["dic_alt_co2"] = t["dic"] * (1.0 - 0.02 * U(13))
    u15 = U(15)
    alkf = np.where(u15 < 0.005, 0.8, 1.0)
    t["alk"] = ((2300.0 + 80.0 * s) * (0.95 + 0.1 * u12)) * alkf
    t["doc"] = (5.0 + 60.0 * e) * (0.5 + U(16))
    t["don"] = (0.5 + 4.0 * e) * (0.5 + U(17))
    t["dofe"] = (1.0e-5 + 1.0e-4 * e) * (0.5 + U(18))
    t["dop"] = (0.02 + 0.2 * e) * (0.5 + U(19))
    t["dopr"] = 0.03 * (0.5 + U(20)) + 0.0 * s
    t["donr"] = 1.5 * (0.5 + U(21)) + 0.0 * s
    t["zooC"] = (0.002 + 1.0 * e) * (0.5 + U(22))
    t["spC"] = (0.01 + 3.0 * e) * (0.5 + U(23))
    t["diatC"] = (0.01 + 2.5 * e) * (0.5 + U(24))
    t["diazC"] = (0.001 + 0.3 * e) * (0.5 + U(25))
    t["phaeoC"] = (0.005 + 1.0 * e) * (0.5 + U(26))
    t["spChl"] = t["spC"] * (0.15 + 0.1 * U(27))
    t["diatChl"] = t["diatC"] * (0.15 + 0.1 * U(28))
    t["diazChl"] = t["diazC"] * (0.15 + 0.1 * U(29))
    t["phaeoChl"] = t["phaeoC"] * (0.15 + 0.1 * U(30))
    t["spFe"] = t["spC"] * (1.0e-5 + 3.0e-5 * U(31))
    t["diatFe"] = t["diatC"] * (1.0e-5 + 3.0e-5 * U(32))
    t["diazFe"] = t["diazC"] * (1.0e-5 + 3.0e-5 * U(33))
    t["phaeoFe"] = t["phaeoC"] * (1.0e-5 + 3.0e-5 * U(34))
    t["diatSi"] = t["diatC"] * (0.15 * (0.5 + U(35)))
    u36 = U(36)
    car = np.where(u36 < 0.03, 0.6, 0.05 * (0.5 + u36))
    t["spCaCO3"] = t["spC"] * car

    # per-cell specials
    h = ucell(cc, kk, 0)
    hs = (h * 400.0).astype(np.int64) % 4
    zm = h < 0.01
    for a, nm in enumerate(["spChl", "diatChl", "diazChl", "phaeoChl"]):
        t[nm] = np.where(zm & (hs == a), 0.0, t[nm])
    t["nh4"] = np.where((h >= 0.01) & (h < 0.015), -t["nh4"], t["nh4"])
    t["fe"] = np.where((h >= 0.015) & (h < 0.02), 0.0, t["fe"])
    out["tracers"] = t

    # surface forcing
    f = {}
    f["surfacePressure"] = 0.98 + 0.04 * ucol(c, 40)
    u41 = ucol(c, 41)
    f["iceFraction"] = np.where(u41 < 0.005, -0.1, np.where(u41 > 0.995, 1.2, 0.3 * u41))
    w = (3.0 + 12.0 * ucol(c, 42)) * 100.0
    f["windSpeedSquared10m"] = w * w
    f["atmCO2"] = np.full(ncol, 400.0)
    f["atmCO2_ALT_CO2"] = np.full(ncol, 280.0)
    f["surface_pH"] = np.zeros(ncol)
    f["surface_pH_alt_co2"] = np.zeros(ncol)
    f["surfaceDepth"] = np.full(ncol, dz1[0] / 200.0)
    f["SST"] = out["temp"][:, 0].copy()
    f["SSS"] = out["salt"][:, 0].copy()
    out["forcing"] = f
    n = np.arange(30, dtype=np.uint64)[None, :]
    for i, nm in enumerate(["depositionFlux", "riverFlux", "gasFlux", "seaIceFlux"]):
        out[nm] = np.ascontiguousarray((1.0e-9 * ucell(cc, n, 1 + i)).T)   # [n, ncol] == Fortran X(column, n)

    # DMS / MACROS own tracers
    out["dms"] = (2.0e-3 * e) * (0.5 + U(50)) + 1.0e-5
    out["dmsp"] = (5.0e-3 * e) * (0.5 + U(51)) + 1.0e-5
    out["prot"] = 1.0 * (0.5 + U(52)) + 0.0 * s
    out["poly"] = 2.0 * (0.5 + U(53)) + 0.0 * s
    out["lip"] = 0.1 * (0.5 + U(54)) + 0.0 * s
    return out


def default_indices(names=BGC_TRACERS):
    return {nm: i + 1 for i, nm in enumerate(names)}


def pack_tracers(named, names, indices=None):
    """-> array [n, ncol, nz] (C order == Fortran X(k,column,n)); indices 1-based"""
    if indices is None:
        indices = default_indices(names)
    first = named[names[0]]
    arr = np.zeros((len(names),) + first.shape, dtype=np.float64)
    for nm in names:
        arr[indices[nm] - 1] = named[nm]
    return arr


def bgc_tracer_array(syn, indices=None):
    return pack_tracers(syn["tracers"], BGC_TRACERS, indices)


def dms_tracer_array(syn, indices=None):
    named = dict(syn["tracers"])
    named["dms"] = syn["dms"]
    named["dmsp"] = syn["dmsp"]
    return pack_tracers(named, DMS_TRACERS, indices)


def macros_tracer_array(syn, indices=None):
    named = dict(syn["tracers"])
    for nm in ("prot", "poly", "lip"):
        named[nm] = syn[nm]
    return pack_tracers(named, MACROS_TRACERS, indices)
