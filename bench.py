#!/usr/bin/env python3
"""Headline benchmark: column-level tracer updates/s of the BGC column hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--diag full|none] [--ph cold|warm]

One "step" = one pass of the hot path over one batch of synthetic columns resident in HBM in the engine's native
column-fastest layout.  The default workload is the largest single-GPU configuration of BASELINE.json, cfg4:
1 M columns x 80 levels, BGC_SourceSink + DMS_SourceSink + MACROS_SourceSink (one fused call when the library offers it),
every diagnostic written like the reference does, a COLD pH state (PH_PREV = 0) at every timed step.
cfg2 = 100 k x 60 BGC only, cfg3 = 1 M x 60 BGC + BGC_SurfaceFluxes, cfg5 = 1 M x 100 per GPU, every routine.
One update = one prognostic tendency value of one active cell (30 BGC + 2 DMS + 3 MACROS).

N > 1: one process per GPU.  Started by a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the process is one
rank; started plainly as `python bench.py --gpus N` it starts its own N ranks -- before touching any GPU -- and relays
rank 0's line.  torch.distributed (RCCL) only provides the barrier and the max-over-ranks of the wall time: columns are
independent, rank r owns global columns [r*ncol, (r+1)*ncol), there is no data-path collective.  Scaling is WEAK (the
per-GPU batch is fixed) unless --scaling strong splits the configuration's columns over the ranks.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the step, timed with HIP events on the
engine's own streams; `cpu_baseline` times the reference arithmetic (oracle/_ref = the unmodified Fortran, else the C
restatement) on host cores over a bounded column sample of the same workload; `other_workloads` (N = 1 only) holds short
runs of cfg2, cfg3 and the warm-bracket variant of the headline workload.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (columns per GPU, levels, modules)
    "cfg2": (100_000, 60, ("bgc",)),
    "cfg3": (1_000_000, 60, ("bgc", "surf")),
    "cfg4": (1_000_000, 80, ("bgc", "dms", "macros")),
    "cfg5": (1_000_000, 100, ("bgc", "surf", "dms", "dms_surf", "macros")),
}
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s spec (6.3 TB/s is what a streaming copy reaches on this part)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64, FMA counted as 2

# ALGORITHMIC bytes per ACTIVE cell: every array of the routine touched once, fp64.
#   BGC_SourceSink: 38 inputs + 32 outputs = 560 B; + 130 diagnostic values = 1600 B
#   DMS_SourceSink: 14 tracers + dz in, 2 tendencies out = 136 B; + 27 diagnostics = 352 B
#   MACROS_SourceSink: 8 tracers in, 3 tendencies out = 88 B; + 6 diagnostics = 136 B
#   the three fused, shared tracers read once: 43 in + 37 out = 640 B; + 163 diagnostics = 1944 B
BYTES_BGC = {"none": 560, "full": 1600}
BYTES_CARBONATE = {"none": 88, "full": 168}      # the carbonate block's share of BYTES_BGC
BYTES_DMS = {"none": 136, "full": 352}
BYTES_MACROS = {"none": 88, "full": 136}
BYTES_FUSED = {"none": 640, "full": 1944}
# the ecosystem cell kernel alone: reads 27 tracers + T, dz, zmid, writes 19 final tendencies; with diagnostics it also
# writes the 72 per-autotroph and 21 of the (k,column) diagnostics and reads S and zbot(k-1)
BYTES_CELL = {"none": (27 + 3 + 19) * 8, "full": (27 + 5 + 19 + 72 + 21) * 8}
# the DMS / MACROS terms folded into it by the fused step: 2 + 3 own tracers in, 2 + 3 tendencies out (+ 27 + 6 diagnostics)
BYTES_CELL_DM = {"none": 10 * 8, "full": (10 + 33) * 8}


def shard(rank, world, ncol_per_gpu):
    """Static column-range partition: rank r owns global columns [r*ncol, (r+1)*ncol).  Weak scaling: the per-GPU
    batch does not depend on `world`."""
    return rank * ncol_per_gpu, ncol_per_gpu


def shard_strong(rank, world, total):
    """--scaling strong: the configuration's columns split over the ranks; the last rank takes what is left."""
    per = (total + world - 1) // world
    col0 = rank * per
    return col0, max(0, min(per, total - col0))


def reduce_results(dist, elapsed, updates, device):
    """max-over-ranks of the wall time, sum-over-ranks of the work; no data-path collective.
    -> (slowest rank's seconds, updates of all ranks, fastest rank's seconds)"""
    if dist is None:
        return elapsed, float(updates), elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(updates)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    lo = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float(t.item()), float(u.item()), float(lo.item())


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_worker(job):
    """Runs in a child process: reference arithmetic on one column slice; returns (updates, cold seconds, warm seconds)."""
    kind, col0, nc, nz, modules = job
    sys.path.insert(0, ROOT)
    import numpy as np
    from __graft_entry__ import load_pkg
    load_pkg()
    from ocean_bgc_amd import synth
    if kind == "reference":
        from oracle import refbind
        R = refbind.Ref()
    else:
        from oracle import orabind
        R = orabind.Oracle()
    syn = synth.generate(nc, nz, col0=col0)
    tr = synth.bgc_tracer_array(syn)
    z = np.zeros((nc, nz))
    cells = int(np.minimum(syn["kmax"], nz).clip(0).sum())
    f = syn["forcing"]
    dtr = synth.dms_tracer_array(syn) if ("dms" in modules or "dms_surf" in modules) else None
    mtr = synth.macros_tracer_array(syn) if "macros" in modules else None

    def one_pass(tr, php, pha, forcing):
        upd, sec = 0, 0.0
        t0 = time.perf_counter()
        res = R.bgc_source_sink(tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
                                syn["fesed"], syn["dust"], syn["sw"], php, pha)
        sec += res.get("seconds", time.perf_counter() - t0)
        upd += 30 * cells
        r2 = None
        if "surf" in modules:
            t0 = time.perf_counter()
            r2 = R.bgc_surface_fluxes(tr, forcing, syn["depositionFlux"], syn["riverFlux"], syn["gasFlux"], syn["seaIceFlux"])
            sec += r2.get("seconds", time.perf_counter() - t0)
        if "dms" in modules:
            t0 = time.perf_counter()
            r3 = R.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"])
            sec += r3.get("seconds", time.perf_counter() - t0)
            upd += 2 * cells
        if "dms_surf" in modules:
            t0 = time.perf_counter()
            r5 = R.dms_surface_fluxes(dtr, f)
            sec += r5.get("seconds", time.perf_counter() - t0)
        if "macros" in modules:
            t0 = time.perf_counter()
            r4 = R.macros_source_sink(mtr, syn["kmax"])
            sec += r4.get("seconds", time.perf_counter() - t0)
            upd += 3 * cells
        return upd, sec, res, r2

    upd, cold, res, r2 = one_pass(tr, z, z, f)
    # warm: the previous pH is the bracket centre and every tracer has moved a little, as in a running model
    tr2 = tr * (1.0 + 1.0e-3 * np.sin(np.arange(tr.size).reshape(tr.shape)))
    _, warm, _, _ = one_pass(tr2, res["ph_prev"], res["ph_prev_alt"], r2["forcing"] if r2 is not None else f)
    return upd, cold, warm


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(nz, modules, seconds_per_worker=3.0):   # measured: ~1.35 us / cell / core cold on the box's EPYC
    """Reference arithmetic on host cores: P processes (the reference keeps module state, co2calc.F90:65-67, so processes
    not threads), static column split of a bounded sample of the same workload; cold pass then warm pass."""
    from oracle import refbind
    kind = "reference" if refbind.available() else "port"
    if kind == "port":
        from oracle import orabind
        if not orabind.available():
            orabind.build()
    ncpu = os.cpu_count() or 1
    cores = max(1, min(16, ncpu))          # a one-GPU box's CPU share is 16 cores whatever the machine has
    per_cell_us = 1.2 * (1.0 + 0.15 * ("dms" in modules) + 0.05 * ("macros" in modules))     # ~2 us / cell / core (EPYC host)
    per = max(64, int(seconds_per_worker / (per_cell_us * 1e-6 * nz)))
    jobs = [(kind, i * per, per, nz, tuple(modules)) for i in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    upd = sum(r[0] for r in res)
    cold = max(r[1] for r in res)
    warm = max(r[2] for r in res)
    return {"value": upd / cold, "unit": "updates/s", "cores": cores, "kind": kind, "warm_value": upd / warm,
            "sample": "%d columns x %d levels (%d per process, %d processes of %d logical CPUs, %s), modules=%s, cold pH then "
                      "warm pH; slowest process %.2f s cold / %.2f s warm; per-core %.3g updates/s cold; pool wall %.1f s"
                      % (per * cores, nz, per, cores, ncpu, _cpu_model(), "+".join(modules), cold, warm, upd / cold / cores, wall)}


# ------------------------------------------------------------------------------------------------ GPU workload
class Workload:
    """Device-resident inputs and outputs of one configuration, generated on the device (bit-identical to synth.py)."""

    def __init__(self, eng, torch, L, name, ncol, nz, modules, col0, diag_on, ph, steps, fused, layout="tiled"):
        self.eng, self.torch, self.L = eng, torch, L
        self.name, self.ncol, self.nz, self.modules, self.diag_on, self.ph = name, ncol, nz, modules, diag_on, ph
        dev = torch.device("cuda", torch.cuda.current_device())
        f64 = dict(dtype=torch.float64, device=dev)
        E = torch.empty
        self.fused = bool(fused) and ("dms" in modules and "macros" in modules) and hasattr(eng, "fused_step")
        # the separate DMS / MACROS entry points keep scratch in the caller's index space: native layout only
        self.tiled = layout == "tiled" and (self.fused or not ("dms" in modules or "macros" in modules))
        # cold start per step: every timed step finds a zeroed pH pair of its own resident before the timed region when
        # device memory allows (two planes per step, at most 64 pairs); otherwise as many pairs as fit rotate and are
        # cleared again, by one fill, inside the timing
        with_own = "dms" in modules or "macros" in modules or "dms_surf" in modules
        with_dtr = ("dms" in modules or "dms_surf" in modules) and (not self.fused or "dms_surf" in modules)
        # level-dependent planes of the workload without the pH pairs
        nfixed = 30 + 6 + (5 if with_own else 0) + (14 if with_dtr else 0) + 30 + (5 if with_own else 0)
        if not self.fused:
            nfixed += (14 if "dms" in modules else 0) + (16 if "macros" in modules else 0)
        if diag_on:
            nfixed += len(L.BGC_DIAG_KC) - len(L.BGC_DIAG_KC_UNTOUCHED) + 4 * len(L.BGC_DIAG_KCA)
            nfixed += (len(L.DMS_DIAG_KC) if "dms" in modules else 0) + (len(L.MACROS_DIAG_KC) if "macros" in modules else 0)
        self.nbuf = 1
        if ph != "warm":
            plane = ((ncol + 63) // 64) * 64 * nz * 8
            free = torch.cuda.mem_get_info()[0]
            fit = int((0.85 * free / plane - nfixed) // 2)       # pairs that fit beside the workload's own arrays
            self.nbuf = max(1, min(steps, 64, max(fit, 8)))
        if self.tiled:
            # ONE arena for every level-dependent array: arena[block of 64 columns][level][row][64]; inputs first
            from ocean_bgc_amd.engine import TileArena
            nrows = nfixed + 2 * self.nbuf
            nrows += int(os.environ.get("OBGC_BENCH_TILE_PAD", "0"))      # experiment knob: unused rows at the end of every tile
            self.arena = TileArena(torch, dev, ncol, nz, nrows)
            KC = lambda n=1: self.arena.take(n, (ncol, nz) if n == 1 else None)
            self.gkw = dict(ld=self.arena.ld, plane=64)
            self.layout = L.OBGC_LAYOUT_TILED
            plane_of = lambda x, j: x.sub(j)
            view_of = lambda x, j: x.view()[:, :, j, :]
        else:
            KC = lambda n=1: E((nz, ncol) if n == 1 else (n, nz, ncol), **f64)
            self.gkw = {}
            self.layout = L.OBGC_LAYOUT_COLFAST
            plane_of = lambda x, j: x[j]
            view_of = plane_of
        self.tr = KC(30)
        self.kc = {nm: KC() for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
        self.cc = {nm: E((ncol,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
        self.kmax = E((ncol,), dtype=torch.int32, device=dev)
        eng.synth_fill(nz, ncol, col0, tracers=self.tr, kmax=self.kmax, tiled=self.tiled, **self.gkw, **self.kc, **self.cc)
        own = KC(5) if with_own else None
        dtr = KC(14) if with_dtr else None
        self.tend = KC(30)
        # adjacent planes, so that one fill clears them all
        self.ph_all = KC(2 * self.nbuf)
        self.phs = [(plane_of(self.ph_all, 2 * i), plane_of(self.ph_all, 2 * i + 1)) for i in range(self.nbuf)]
        self.rezero()
        self.diag = None
        if diag_on:
            d = {}
            for nm in L.BGC_DIAG_KC:
                if nm not in L.BGC_DIAG_KC_UNTOUCHED:
                    d[nm] = KC()
            for nm in L.BGC_DIAG_KCA:
                d[nm] = KC(4)
            for nm in L.BGC_DIAG_CA:
                d[nm] = E((4, ncol), **f64)
            for nm in L.BGC_DIAG_C:
                d[nm] = E((ncol,), **f64)
            self.diag = d
        # experiment switch (timings only, the byte counts no longer apply): the diagnostic code paths run but store nothing
        null_diag = diag_on and os.environ.get("OBGC_BENCH_NULLDIAG", "0") == "1"
        if null_diag:
            self.diag = {}
        self.surf = self.dms = self.macros = None
        extra = {}
        if "surf" in modules or "dms_surf" in modules:
            s = {nm: E((ncol,), **f64) for nm in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2", "atmCO2_ALT_CO2",
                                                  "surfaceDepth", "SST", "SSS", "surface_pH", "surface_pH_alt_co2")}
            flux = {nm: E((30, ncol), **f64) for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")}
            extra.update(s)
            extra.update(flux)
            self.surf = dict(forcing=s, flux=[flux[nm] for nm in ("depositionFlux", "riverFlux", "gasFlux", "seaIceFlux")],
                             flux0=None, ice0=None, net=E((30, ncol), **f64),
                             diag={nm: E((ncol,), **f64) for nm in L.BGC_FLUX_DIAG} if diag_on else None)
        if with_own:
            extra.update(dms=plane_of(own, 0), dmsp=plane_of(own, 1), prot=plane_of(own, 2), poly=plane_of(own, 3), lip=plane_of(own, 4))
        if extra:
            eng.synth_fill_extra(nz, ncol, col0, tiled=self.tiled, ld=self.gkw.get("ld", 0), **extra)
        if "dms" in modules or "dms_surf" in modules:
            if dtr is not None:     # the separate entry points read a 14-plane DMS tracer array
                view_of(dtr, 0).copy_(view_of(own, 0)); view_of(dtr, 1).copy_(view_of(own, 1))
                for j, nm in enumerate(L.DMS_TRACERS[2:], start=2):
                    view_of(dtr, j).copy_(view_of(self.tr, L.BGC_TRACERS.index(nm)))
            dtend = KC(2 if self.fused else 14)
            self.dms = dict(tr=dtr, own=[plane_of(own, 0), plane_of(own, 1)], tend=dtend,
                            tend_own=[plane_of(dtend, 0), plane_of(dtend, 1)] if self.fused else None,
                            diag=({} if null_diag else {nm: KC() for nm in L.DMS_DIAG_KC}) if diag_on else None,
                            sst=self.kc["temp"].surface_row() if self.tiled else self.kc["temp"][0].contiguous(),
                            net=E((14, ncol), **f64) if "dms_surf" in modules else None,
                            fdiag={nm: E((ncol,), **f64) for nm in L.DMS_FLUX_DIAG} if ("dms_surf" in modules and diag_on) else None)
        if "macros" in modules:
            mtr = None
            if not self.fused:
                mtr = E((8, nz, ncol), **f64)
                for j in range(3):
                    mtr[j].copy_(own[2 + j])
                for j, nm in enumerate(L.MACROS_TRACERS[3:], start=3):
                    mtr[j].copy_(self.tr[L.BGC_TRACERS.index(nm)])
            mtend = KC(3 if self.fused else 8)
            self.macros = dict(tr=mtr, own=[plane_of(own, 2 + j) for j in range(3)], tend=mtend,
                               tend_own=[plane_of(mtend, j) for j in range(3)] if self.fused else None,
                               diag=({} if null_diag else {nm: KC() for nm in L.MACROS_DIAG_KC}) if diag_on else None)
        self.active_cells = int(torch.clamp(self.kmax, 0, nz).sum().item())
        self.per_cell = 30 + (2 if "dms" in modules else 0) + (3 if "macros" in modules else 0)
        self.updates_per_step = self.active_cells * self.per_cell
        CF = self.layout
        kc, cc = self.kc, self.cc
        # the argument blocks of the calls are built once, as a compiled host would: the arrays do not move between steps
        if self.fused:
            self.prepared = [eng.fused_step_prepare(nz, ncol, CF, self.tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"],
                                                    cc["lat"], self.kmax, kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"],
                                                    self.tend, p[0], p[1], self.diag, True,
                                                    dms=self.dms["own"][0], dmsp=self.dms["own"][1], SST=self.dms["sst"],
                                                    dms_tend=self.dms["tend_own"][0], dmsp_tend=self.dms["tend_own"][1],
                                                    dms_diag=self.dms["diag"],
                                                    prot=self.macros["own"][0], poly=self.macros["own"][1], lip=self.macros["own"][2],
                                                    prot_tend=self.macros["tend_own"][0], poly_tend=self.macros["tend_own"][1],
                                                    lip_tend=self.macros["tend_own"][2], macros_diag=self.macros["diag"], **self.gkw)
                             for p in self.phs]
        else:
            self.prepared = [eng.bgc_source_sink_prepare(nz, ncol, CF, self.tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"],
                                                         cc["lat"], self.kmax, kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"],
                                                         self.tend, p[0], p[1], self.diag, True, **self.gkw)
                             for p in self.phs]
        self.kms = {"call": [], "carbonate": [], "ecosys": [], "cell": [], "dms": [], "macros": []}
        torch.cuda.synchronize()
        if self.surf is not None:
            # BGC_SurfaceFluxes scales the Fe rows of its flux arrays and clamps iceFraction IN PLACE: every step starts
            # from the generated values again (4 short device copies per step, inside the timing)
            self.surf["flux0"] = [x.clone() for x in self.surf["flux"]]
            self.surf["ice0"] = self.surf["forcing"]["iceFraction"].clone()

    def step(self, i, record):
        eng, L, torch = self.eng, self.L, self.torch
        CF = self.layout
        nz, ncol = self.nz, self.ncol
        if self.surf is not None:
            e = self.surf
            if self.ph == "cold":
                e["forcing"]["surface_pH"].zero_(); e["forcing"]["surface_pH_alt_co2"].zero_()
            for x, x0 in zip(e["flux"], e["flux0"]):
                x.copy_(x0)
            e["forcing"]["iceFraction"].copy_(e["ice0"])
            torch.cuda.synchronize()      # torch's stream -> engine's streams
            if "surf" in self.modules:     # the host model calls *_SurfaceFluxes before *_SourceSink
                eng.bgc_surface_fluxes(nz, ncol, CF, self.tr, e["forcing"], *e["flux"], e["net"], e["diag"], **self.gkw)
            if "dms_surf" in self.modules:
                d = self.dms
                eng.dms_surface_fluxes(nz, ncol, CF, d["tr"], e["forcing"], d["net"], d["fdiag"], **self.gkw)
        if self.fused:
            eng.fused_step_run(self.prepared[i % self.nbuf])
        else:
            eng.bgc_source_sink_run(self.prepared[i % self.nbuf])
        if record:
            k = self.kms
            k["call"].append(eng.last_kernel_ms(0)); k["carbonate"].append(eng.last_kernel_ms(1))
            k["ecosys"].append(eng.last_kernel_ms(2)); k["cell"].append(eng.last_kernel_ms(4))
        if not self.fused:
            if "dms" in self.modules:
                d = self.dms
                eng.dms_source_sink(nz, ncol, CF, d["tr"], self.kc["dz"], self.kmax, d["sst"], self.cc["sw_flux"], d["tend"], d["diag"])
                if record:
                    self.kms["dms"].append(eng.last_kernel_ms(0))
            if "macros" in self.modules:
                m = self.macros
                eng.macros_source_sink(nz, ncol, CF, m["tr"], self.kmax, m["tend"], m["diag"])
                if record:
                    self.kms["macros"].append(eng.last_kernel_ms(0))

    def rezero(self):
        self.ph_all.zero_()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, barrier):
        torch = self.torch
        torch.cuda.synchronize()              # inputs were produced on torch's stream / the engine's stream
        for i in range(warmup):
            self.step(i, False)
        if self.ph == "cold":
            self.rezero()
        elif warmup == 0:
            self.step(0, False)               # a warm measurement needs one untimed call to have a previous pH
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(i, True)                # engine calls are synchronous: each returns with its kernels done
            if self.ph == "cold" and (i + 1) % self.nbuf == 0 and i + 1 < steps:
                self.rezero()                 # (only when memory did not allow a pair per step) inside the timing
        barrier()
        return time.perf_counter() - t0

    def workload_name(self):
        return ("%s: %d columns x %d levels per GPU, %s%s, %s, diagnostics=%s, %s"
                % (self.name, self.ncol, self.nz, "+".join(self.modules), " (one fused call)" if self.fused else "",
                   "column-block layout (64-column blocks, one arena)" if self.tiled else "native column-fastest layout",
                   "full" if self.diag_on else "none",
                   "cold pH bracket every step" if self.ph == "cold" else "warm pH brackets (previous step)"))


def short_run(eng, torch, L, name, diag_on, ph, steps, warmup, fused, layout="tiled"):
    ncol, nz, modules = CONFIGS[name]
    w = Workload(eng, torch, L, name, ncol, nz, modules, 0, diag_on, ph, steps, fused, layout)
    el = w.timed(steps, warmup, torch.cuda.synchronize)
    out = {"workload": w.workload_name(), "value": w.updates_per_step * steps / el, "unit": "updates/s", "steps": steps,
           "ms_per_step": el / steps * 1e3}
    del w
    torch.cuda.empty_cache()
    return out


def end_to_end_host_call(eng, L, calls=3):
    """What a host pays that only swaps the link line: ONE BGC_SourceSink of cfg2 (100 000 columns x 60 levels, every
    diagnostic the reference's types carry) on pageable HOST arrays in the reference's level-fastest layout, staged over PCIe
    by the library (OBGC_MEM_HOST, the path of the Fortran drop-in modules).  Wall time of the call, cold pH; never `value`."""
    import numpy as np
    from ocean_bgc_amd import synth
    nc, nz, _ = CONFIGS["cfg2"]
    syn = synth.generate(nc, nz)
    tr = synth.bgc_tracer_array(syn)
    tend = np.zeros_like(tr)
    ph = np.zeros((2, nc, nz))
    diag = {}
    for nm in L.BGC_DIAG_KC:
        if nm not in L.BGC_DIAG_KC_UNTOUCHED:
            diag[nm] = np.zeros((nc, nz))
    for nm in L.BGC_DIAG_KCA:
        diag[nm] = np.zeros((4, nc, nz))
    for nm in L.BGC_DIAG_CA:
        diag[nm] = np.zeros((4, nc))
    for nm in L.BGC_DIAG_C:
        diag[nm] = np.zeros((nc,))
    kmax = np.asarray(syn["kmax"], dtype=np.int32)
    cells = int(np.minimum(kmax, nz).clip(0).sum())
    ts = []
    for _ in range(calls):
        ph[:] = 0.0
        t0 = time.perf_counter()
        eng.bgc_source_sink(nz, nc, L.OBGC_LAYOUT_KFAST, tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"],
                            kmax, syn["fesed"], syn["dust"], syn["sw"], tend, ph[0], ph[1], diag, True)
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:]) if len(ts) > 1 else ts[0]      # the first call also grows the staging arena
    up = sum(a.nbytes for a in (tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["fesed"], ph))
    down = tend.nbytes + ph.nbytes + sum(v.nbytes for v in diag.values())
    return {"workload": "cfg2: one BGC_SourceSink call on pageable host arrays, reference layout, every diagnostic, cold pH "
                        "(library stages over PCIe; Fortran drop-in path)",
            "ms_per_call": best * 1e3, "value": 30 * cells / best, "unit": "updates/s", "calls": calls,
            "bytes_up": up, "bytes_down": down, "link_GBps": (up + down) / best / 1e9,
            "note": "PCIe-inclusive; for comparison with other_workloads.cfg2 (the same call on resident arrays); never the headline"}


def roofline_of(w, eng, args, ROOT):
    """The dominant kernel of the step = the longest single launch."""
    import numpy as np
    dk = "full" if w.diag_on else "none"
    mean = lambda v: float(np.mean(v)) if v else 0.0
    k = w.kms
    t_car, t_eco, t_call, t_cell = mean(k["carbonate"]), mean(k["ecosys"]), mean(k["call"]), mean(k["cell"])
    sched = eng.last_schedule() if hasattr(eng, "last_schedule") else -1
    cand = {}
    b_cell = BYTES_CELL[dk] + (BYTES_CELL_DM[dk] if w.fused else 0)
    if sched == 2:      # the carbonate system is solved inside the cell kernel's wavefronts: one launch does both
        cand["bgc_cell_kernel(+carbonate%s)" % ("+dms+macros" if w.fused else "")] = (t_cell, b_cell + BYTES_CARBONATE[dk])
    elif sched == 3:    # the column-block kernel does the whole ecosystem (+ carbonate) in one launch
        cand["bgc_column_kernel(+carbonate%s)" % ("+dms+macros" if w.fused else "")] = (t_cell, (BYTES_FUSED[dk] if w.fused else BYTES_BGC[dk]))
    elif sched == 4:    # column-block kernel (light + cell terms + particle recurrence) with the carbonate kernel beside it
        cand["bgc_column_kernel%s" % ("(+dms+macros)" if w.fused else "")] = (
            t_cell, (BYTES_FUSED[dk] if w.fused else BYTES_BGC[dk]) - BYTES_CARBONATE[dk])
        cand["bgc_carbonate_kernel"] = (t_car, BYTES_CARBONATE[dk])
    else:
        cand["bgc_cell_kernel%s" % ("(+dms+macros)" if w.fused else "")] = (t_cell, b_cell)
        cand["bgc_carbonate_kernel"] = (t_car, BYTES_CARBONATE[dk])
    dom = max(cand, key=lambda n: cand[n][0])
    t_dom, b_dom = cand[dom]
    achieved = w.active_cells * b_dom / (t_dom * 1e-3) / 1e9 if t_dom > 0 else 0.0
    step_bytes = (BYTES_FUSED[dk] if ("dms" in w.modules and "macros" in w.modules) else
                  BYTES_BGC[dk] + (BYTES_DMS[dk] if "dms" in w.modules else 0) + (BYTES_MACROS[dk] if "macros" in w.modules else 0))
    t_step = t_call + mean(k["dms"]) + mean(k["macros"])
    # HBM bytes per launch and fp64 VALU utilisation from the rocprofv3 PMC passes of THIS kernel revision and workload
    # (profiles/pmc_traffic.json, profiles/valu.json, written by tools/pmc_summary.py); null when they belong to another
    traffic = valu = None
    try:
        rev = open(os.path.join(ROOT, "ocean-bgc_amd", "KERNEL_REV")).read().strip()
        key = "%s_%s" % (w.name, dk)
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tab.get("kernel_rev") == rev and not args.columns:
            base = dom.split("(")[0]
            traffic = tab["runs"][key][base]["total"]
        vt = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
        if vt.get("kernel_rev") == rev and not args.columns:
            valu = {kk: round(v["valu_busy"], 3) for kk, v in vt["runs"][key].items()}
    except Exception:
        pass
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "fp64_valu_busy": valu,
            "algorithmic_bytes_per_launch": w.active_cells * b_dom, "algorithmic_bytes_per_cell": b_dom,
            "active_cells_per_launch": w.active_cells, "avg_launch_ms": t_dom, "schedule": sched,
            "kernel_ms": {"bgc_carbonate_kernel": t_car, "bgc_cell_kernel": t_cell, "ecosystem_light+cell+sweep": t_eco,
                          "bgc_call_device_ms": t_call, "dms_call_device_ms": mean(k["dms"]), "macros_call_device_ms": mean(k["macros"])},
            "whole_step": {"algorithmic_bytes_per_cell": step_bytes, "device_ms": t_step,
                           "GBps": w.active_cells * step_bytes / (t_step * 1e-3) / 1e9 if t_step > 0 else 0.0,
                           "frac": (w.active_cells * step_bytes / (t_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_step > 0 else 0.0},
            "note": "frac prices the dominant kernel's ALGORITHMIC bytes against the 8 TB/s spec; whole_step does the same for "
                    "every kernel of the step together (device time of the calls); the carbonate solve is bound by fp64 VALU "
                    "issue, not by HBM"}


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus_without_hip():
    """GPUs the ranks will see, counted without loading a GPU runtime in THIS process: the visibility variables if set, else
    a short-lived child process asks torch (the kernel driver's topology files list every GPU of the host, also those a
    container cannot open, so they are no substitute)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             text=True, timeout=600).stdout.strip().splitlines()
        return int(out[-1]) if out else 0
    except Exception:
        return 0


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) and relay rank 0's line.
    The parent never touches a GPU and loads no GPU runtime (count_gpus_without_hip)."""
    n = args.gpus
    share = os.environ.get("OBGC_BENCH_SHARE_GPU", "0") == "1"
    have = count_gpus_without_hip()
    if have < 1 or (have < n and not share):
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible\n" % (n, have))
        return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OBGC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % ", ".join("rank %d rc %d" % b for b in bad))
        return 1
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None or json.loads(line).get("n_gpus") != n:
        sys.stderr.write("bench.py: rank 0 printed no result line for %d GPUs\n" % n)
        return 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)      # x ~41 ms: a timed region of 1.2 s
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--diag", default="full", choices=("full", "none"),
                    help="full: every diagnostic written like the reference does; none: tendencies + pH state only")
    ap.add_argument("--ph", default="cold", choices=("cold", "warm"),
                    help="cold (headline): PH_PREV_* = 0 at every step, full [6, 9] bracket; warm: the previous step's pH is "
                         "the bracket centre, as in a running model")
    ap.add_argument("--columns", type=int, default=0, help="override columns per GPU")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default, what a scaling run uses): the config's columns on EVERY GPU; strong: the config's "
                         "columns split over the GPUs")
    ap.add_argument("--separate", action="store_true", help="cfg4 / cfg5: three separate calls instead of the fused step")
    ap.add_argument("--layout", default="tiled", choices=("tiled", "colfast"),
                    help="tiled (default): blocks of 64 columns, every level-dependent array a set of rows of one arena "
                         "(OBGC_LAYOUT_TILED); colfast: one plane per field (OBGC_LAYOUT_COLFAST)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short cfg2 / cfg3 / warm-bracket runs of the N = 1 line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    # OBGC_BENCH_FORCE_DIST=1: a one-GPU run goes through everything an N-GPU rank does -- the RCCL process group bound to
    # its device, the barriers, the device all-reduces -- so that this code has run before a multi-GPU node meets it
    force_dist = os.environ.get("OBGC_BENCH_FORCE_DIST", "0") == "1"
    if env_world is None:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args))
        world, rank, local_rank = 1, 0, 0
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    else:
        world = int(env_world)
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    from __graft_entry__ import load_pkg
    load_pkg()
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # rehearsal on a 1-GPU box: OBGC_BENCH_SHARE_GPU=1 maps every rank onto GPU 0 and uses gloo (CPU tensors) for the
    # barrier / max-reduce (RCCL refuses two ranks on one device); the timed path is otherwise identical
    share = os.environ.get("OBGC_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    red_dev = dev
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            red_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ncol, nz, modules = CONFIGS[args.config]
    if args.columns:
        ncol = args.columns
    if args.scaling == "strong":
        col0, ncol = shard_strong(rank, world, ncol)
    else:
        col0, ncol = shard(rank, world, ncol)
    diag_on = args.diag == "full"
    eng = Engine(local_rank)
    w = Workload(eng, torch, L, args.config, ncol, nz, modules, col0, diag_on, args.ph, args.steps, not args.separate, args.layout)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed = w.timed(args.steps, args.warmup, barrier)
    elapsed, updates_all, fastest = reduce_results(dist, elapsed, w.updates_per_step, red_dev)
    value = updates_all * args.steps / elapsed

    if rank == 0:
        out = {"metric": "column-level tracer updates/sec", "value": value, "unit": "updates/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": w.workload_name(), "columns_per_gpu": ncol, "levels": nz,
                          "active_cells_per_gpu": w.active_cells, "updates_per_step_all_gpus": updates_all,
                          "parallelism": "column-range x%d, no collective" % world,
                          "columns_per_s": updates_all / w.updates_per_step * ncol * args.steps / elapsed if w.updates_per_step else 0.0,
                          "cells_per_s": updates_all / w.per_cell * args.steps / elapsed},
               "roofline": roofline_of(w, eng, args, ROOT)}
        if dist is not None:      # a slow GPU of the node shows here, not as a scaling loss of the path
            out["ranks"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                            "ms_per_step_slowest": elapsed / args.steps * 1e3, "ms_per_step_fastest": fastest / args.steps * 1e3}
        if world == 1 and not args.no_extras and not args.columns:
            del w
            torch.cuda.empty_cache()
            extras = {}
            for key, (name, ph, st, dg) in {"headline_warm_ph": (args.config, "warm" if args.ph == "cold" else "cold", 5, diag_on),
                                            "headline_other_diag_mode": (args.config, args.ph, 5, not diag_on),
                                            "cfg2": ("cfg2", "cold", 10, diag_on), "cfg3": ("cfg3", "cold", 5, diag_on),
                                            # one GPU's share of cfg5 (8 M columns over 8 GPUs): every routine of the path
                                            "cfg5_share_1gpu": ("cfg5", "cold", 4, diag_on)}.items():
                if not key.startswith("headline") and name == args.config:
                    continue
                try:
                    extras[key] = short_run(eng, torch, L, name, dg, ph, st, 2, not args.separate, args.layout)
                except Exception as exc:      # never lose the headline over an extra
                    extras[key] = {"failed": repr(exc)}
            out["other_workloads"] = extras
            try:
                out["end_to_end"] = end_to_end_host_call(eng, L)
            except Exception as exc:
                out["end_to_end"] = {"failed": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(nz, modules)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as exc:          # never lose the GPU number over the baseline leg
                out["cpu_baseline"] = {"value": None, "unit": "updates/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (exc,)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
