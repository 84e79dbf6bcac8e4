#!/usr/bin/env python3
"""Headline benchmark: column-level tracer updates/s of the BGC column hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3|cfg4|cfg5] [--diag full|none]

One "step" = one pass of the hot path over one batch of synthetic columns resident in HBM
in the engine's native column-fastest layout: BGC_SourceSink (cfg2), + BGC_SurfaceFluxes
(cfg3), + DMS_SourceSink + MACROS_SourceSink (cfg4), everything (cfg5).  Every timed step
starts from a COLD pH state (PH_PREV = 0) in freshly zeroed state buffers.
One update = one prognostic tendency value of one active cell (30 BGC + 2 DMS + 3 MACROS).

N > 1: one process per GPU (torch.distributed / RCCL used ONLY for the barrier and the
max-over-ranks of the wall time); columns are independent, so rank r owns global columns
[r*ncol, (r+1)*ncol) and there is no data-path collective.  Scaling is WEAK: the per-GPU
batch is fixed.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel of the step measured
with HIP events on the engine's own streams; `cpu_baseline` times the reference arithmetic
(oracle/_ref = the unmodified Fortran, else the C restatement) on host cores over a bounded
column sample of the same workload.
"""
import argparse
import json
import multiprocessing as mp
import os
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64, FMA counted as 2

# algorithmic bytes per ACTIVE cell (split per kernel in DESIGN.md)
BYTES_CARBONATE = {"none": 88, "full": 168}
BYTES_ECOSYS = {"none": 472, "full": 1432}     # PAR scan + cell kernel + particle sweep together
# the cell kernel alone: reads 27 tracers + T, dz, zmid, writes 19 final tendencies (+ 10 staged ones, 4 hand-off
# scratch values and the 3 light planes it reads back, all engine-internal, not algorithmic); with diagnostics
# it also writes the 72 per-autotroph and 21 of the (k,column) diagnostics and reads S and zbot(k-1)
BYTES_CELL = {"none": (27 + 3 + 19) * 8, "full": (27 + 5 + 19 + 72 + 21) * 8}
BYTES_DMS = {"none": 136, "full": 352}
BYTES_MACROS = {"none": 88, "full": 136}


def shard(rank, world, ncol_per_gpu):
    """Static column-range partition: rank r owns global columns
    [r*ncol, (r+1)*ncol).  Weak scaling: the per-GPU batch does not depend on `world`."""
    return rank * ncol_per_gpu, ncol_per_gpu


def reduce_results(dist, elapsed, updates, device):
    """max-over-ranks of the wall time, sum-over-ranks of the work; no data-path collective."""
    if dist is None:
        return elapsed, float(updates)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(updates)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def _cpu_worker(job):
    """Runs in a child process: reference arithmetic on one column slice; returns (updates, seconds)."""
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
    upd, sec = 0, 0.0
    t0 = time.perf_counter()
    res = R.bgc_source_sink(tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
                            syn["fesed"], syn["dust"], syn["sw"], z, z)
    sec += res.get("seconds", time.perf_counter() - t0)
    upd += 30 * cells
    f = syn["forcing"]
    if "surf" in modules:
        t0 = time.perf_counter()
        r2 = R.bgc_surface_fluxes(tr, f, syn["depositionFlux"], syn["riverFlux"], syn["gasFlux"], syn["seaIceFlux"])
        sec += r2.get("seconds", time.perf_counter() - t0)
    if "dms" in modules:
        dtr = synth.dms_tracer_array(syn)
        t0 = time.perf_counter()
        r3 = R.dms_source_sink(dtr, syn["dz"], syn["kmax"], f["SST"], syn["sw"])
        sec += r3.get("seconds", time.perf_counter() - t0)
        upd += 2 * cells
    if "macros" in modules:
        mtr = synth.macros_tracer_array(syn)
        t0 = time.perf_counter()
        r4 = R.macros_source_sink(mtr, syn["kmax"])
        sec += r4.get("seconds", time.perf_counter() - t0)
        upd += 3 * cells
    return upd, sec


def cpu_baseline(nz, modules, target_core_seconds=20.0):
    """Reference arithmetic on host cores: P processes (the reference keeps module state,
    co2calc.F90:65-67, so processes not threads), static column split of a bounded sample."""
    from oracle import refbind
    kind = "reference" if refbind.available() else "port"
    if kind == "port":
        from oracle import orabind
        if not orabind.available():
            orabind.build()
    cores = max(1, min(16, (os.cpu_count() or 1)))
    per_cell_us = 5.0 * (nz / 60.0) ** 0 * (1.0 + 0.15 * ("dms" in modules))     # ~5 us / cell / core
    ncol_total = int(target_core_seconds / (per_cell_us * 1e-6 * nz))
    per = max(64, ncol_total // cores)
    jobs = [(kind, i * per, per, nz, tuple(modules)) for i in range(cores)]
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    upd = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {"value": upd / slowest, "unit": "updates/s", "cores": cores, "kind": kind,
            "sample": "%d columns x %d levels (%d per process, %d processes), cold pH, modules=%s; per-core %.3g updates/s; pool wall %.1f s"
                      % (per * cores, nz, per, cores, "+".join(modules), upd / slowest / cores, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--diag", default="full", choices=("full", "none"),
                    help="full: all 96 diagnostics written like the reference does; none: tendencies + pH state only")
    ap.add_argument("--ph", default="cold", choices=("cold", "warm"),
                    help="cold (headline): PH_PREV_* = 0 at every step, full [6, 9] bracket; warm (secondary timing of "
                         "SURVEY.md 8d): the previous step's pH is the bracket centre, as in a running model")
    ap.add_argument("--columns", type=int, default=0, help="override columns per GPU")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak (default, what the driver's scaling run uses): the config's columns on EVERY GPU; strong: "
                         "the config's columns split over the GPUs (SURVEY.md 8d asks for both at cfg5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from __graft_entry__ import load_pkg
    load_pkg()
    from ocean_bgc_amd import _lib as L
    from ocean_bgc_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # rehearsal on a 1-GPU box: OBGC_BENCH_SHARE_GPU=1 maps every rank onto GPU 0 and uses gloo for the
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
    if world > 1:
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
        total = ncol
        ncol = (total + world - 1) // world          # the last rank takes what is left
        col0 = rank * ncol
        ncol = max(0, min(ncol, total - col0))
    else:
        col0, ncol = shard(rank, world, ncol)
    diag_on = args.diag == "full"
    eng = Engine(local_rank)
    f64 = dict(dtype=torch.float64, device=dev)

    # ---- inputs: generated on the device, native [plane][level][column] layout ----
    tr = torch.empty((30, nz, ncol), **f64)
    kc = {nm: torch.empty((nz, ncol), **f64) for nm in ("temp", "salt", "zmid", "dz", "zbot", "fesedflux")}
    cc = {nm: torch.empty((ncol,), **f64) for nm in ("lat", "dust_flux_in", "sw_flux")}
    kmax = torch.empty((ncol,), dtype=torch.int32, device=dev)
    eng.synth_fill(nz, ncol, col0, tracers=tr, kmax=kmax, **kc, **cc)
    tend = torch.empty((30, nz, ncol), **f64)
    # cold start per step: a zeroed pH state pair per step, rotated over at most 8 buffers
    nbuf = 1 if args.ph == "warm" else max(1, min(args.steps, 8))
    ph = [torch.zeros((2, nz, ncol), **f64) for _ in range(nbuf)]
    diag = None
    if diag_on:
        diag = {}
        for nm in L.BGC_DIAG_KC:
            if nm not in L.BGC_DIAG_KC_UNTOUCHED:
                diag[nm] = torch.empty((nz, ncol), **f64)
        for nm in L.BGC_DIAG_KCA:
            diag[nm] = torch.empty((4, nz, ncol), **f64)
        for nm in L.BGC_DIAG_CA:
            diag[nm] = torch.empty((4, ncol), **f64)
        for nm in L.BGC_DIAG_C:
            diag[nm] = torch.empty((ncol,), **f64)
    extra = {}
    if "surf" in modules:
        s = {nm: torch.empty((ncol,), **f64) for nm in ("surfacePressure", "iceFraction", "windSpeedSquared10m", "atmCO2",
                                                       "atmCO2_ALT_CO2", "surfaceDepth", "SST", "SSS")}
        u = torch.rand((ncol,), **f64)
        s["surfacePressure"].fill_(1.0); s["iceFraction"].copy_(0.3 * u); s["windSpeedSquared10m"].copy_(((3 + 12 * u) * 100) ** 2)
        s["atmCO2"].fill_(400.0); s["atmCO2_ALT_CO2"].fill_(280.0); s["surfaceDepth"].fill_(5.0)
        s["SST"].copy_(kc["temp"][0]); s["SSS"].copy_(kc["salt"][0])
        s["surface_pH"] = torch.zeros((ncol,), **f64); s["surface_pH_alt_co2"] = torch.zeros((ncol,), **f64)
        extra["surf"] = dict(forcing=s, flux=[torch.full((30, ncol), 1e-9, **f64) for _ in range(4)],
                             net=torch.empty((30, ncol), **f64),
                             diag={nm: torch.empty((ncol,), **f64) for nm in L.BGC_FLUX_DIAG} if diag_on else None)
    if "dms" in modules:
        dtr = torch.empty((14, nz, ncol), **f64)
        dtr[0].fill_(2e-3); dtr[1].fill_(5e-3)
        for j, nm in enumerate(L.DMS_TRACERS[2:], start=2):
            dtr[j].copy_(tr[L.BGC_TRACERS.index(nm)])
        extra["dms"] = dict(tr=dtr, tend=torch.empty((14, nz, ncol), **f64),
                            diag={nm: torch.empty((nz, ncol), **f64) for nm in L.DMS_DIAG_KC} if diag_on else None,
                            sst=kc["temp"][0].contiguous())
    if "macros" in modules:
        mtr = torch.empty((8, nz, ncol), **f64)
        mtr[0].fill_(1.0); mtr[1].fill_(2.0); mtr[2].fill_(0.1)
        for j, nm in enumerate(L.MACROS_TRACERS[3:], start=3):
            mtr[j].copy_(tr[L.BGC_TRACERS.index(nm)])
        extra["macros"] = dict(tr=mtr, tend=torch.empty((8, nz, ncol), **f64),
                               diag={nm: torch.empty((nz, ncol), **f64) for nm in L.MACROS_DIAG_KC} if diag_on else None)

    active_cells = int(torch.clamp(kmax, 0, nz).sum().item())
    per_cell = 30 + (2 if "dms" in modules else 0) + (3 if "macros" in modules else 0)
    updates_per_step = active_cells * per_cell
    CF = L.OBGC_LAYOUT_COLFAST
    kms = {"carbonate": [], "ecosys": [], "call": [], "cell": []}

    # the argument blocks of the calls are built once, as a compiled host would: the arrays do not move between steps
    prepared = [eng.bgc_source_sink_prepare(nz, ncol, CF, tr, kc["temp"], kc["salt"], kc["zmid"], kc["dz"], kc["zbot"], cc["lat"], kmax,
                                            kc["fesedflux"], cc["dust_flux_in"], cc["sw_flux"], tend, p[0], p[1], diag, True)
                for p in ph]

    def step(i, record):
        eng.bgc_source_sink_run(prepared[i % nbuf])
        if record:
            kms["call"].append(eng.last_kernel_ms(0)); kms["carbonate"].append(eng.last_kernel_ms(1))
            kms["ecosys"].append(eng.last_kernel_ms(2)); kms["cell"].append(eng.last_kernel_ms(4))
        if "surf" in extra:
            e = extra["surf"]
            e["forcing"]["surface_pH"].zero_(); e["forcing"]["surface_pH_alt_co2"].zero_()
            torch.cuda.synchronize()      # torch's stream -> engine's streams
            eng.bgc_surface_fluxes(nz, ncol, CF, tr, e["forcing"], *e["flux"], e["net"], e["diag"])
        if "dms" in extra:
            e = extra["dms"]
            eng.dms_source_sink(nz, ncol, CF, e["tr"], kc["dz"], kmax, e["sst"], cc["sw_flux"], e["tend"], e["diag"])
        if "macros" in extra:
            e = extra["macros"]
            eng.macros_source_sink(nz, ncol, CF, e["tr"], kmax, e["tend"], e["diag"])

    def rezero():
        for p in ph:
            p.zero_()
        torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()              # inputs were produced on torch's stream / the engine's stream
    for i in range(args.warmup):
        step(i, False)
    if args.ph == "cold":
        rezero()
    elif args.warmup == 0:
        step(0, False)                    # a warm measurement needs one untimed call to have a previous pH
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)                     # engine calls are synchronous: each returns with its kernels done
        if args.ph == "cold" and (i + 1) % nbuf == 0 and i + 1 < args.steps:
            rezero()                      # keeps every timed step cold; 2 small memsets per 8 steps, inside the timing
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, updates_all = reduce_results(dist, elapsed, updates_per_step, red_dev)
    value = updates_all * args.steps / elapsed

    if rank == 0:
        dk = args.diag
        mean = lambda v: float(np.mean(v)) if v else float("nan")
        t_car, t_eco, t_call, t_cell = mean(kms["carbonate"]), mean(kms["ecosys"]), mean(kms["call"]), mean(kms["cell"])
        # dominant kernel = the longest single launch: the ecosystem cell kernel or the carbonate kernel
        if t_cell >= t_car:
            dom, t_dom, b_dom = "bgc_cell_kernel", t_cell, BYTES_CELL[dk]
        else:
            dom, t_dom, b_dom = "bgc_carbonate_kernel", t_car, BYTES_CARBONATE[dk]
        achieved = active_cells * b_dom / (t_dom * 1e-3) / 1e9
        # HBM bytes per launch from rocprofv3 PMC passes of this kernel revision (tools/pmc_summary.py);
        # null when the committed table belongs to another revision / workload
        traffic = None
        try:
            tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            rev = open(os.path.join(ROOT, "ocean-bgc_amd", "KERNEL_REV")).read().strip()
            key = dom
            if tab.get("kernel_rev") == rev and not args.columns:
                traffic = tab["runs"]["%s_%s" % (args.config, dk)][key]["total"]
        except Exception:
            traffic = None
        # fp64 VALU issue utilisation of the kernels (SQ_ACTIVE_INST_VALU PMC pass of this kernel revision,
        # profiles/valu.json): SURVEY.md 8d asks for it next to the HBM fraction
        valu = None
        try:
            vt = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
            rev = open(os.path.join(ROOT, "ocean-bgc_amd", "KERNEL_REV")).read().strip()
            if vt.get("kernel_rev") == rev and not args.columns:
                valu = {k: round(v["valu_busy"], 3) for k, v in vt["runs"]["%s_%s" % (args.config, dk)].items()}
        except Exception:
            valu = None
        roofline = {"bound": "hbm", "kernel": dom, "fp64_valu_busy": valu, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": active_cells * b_dom,
                    "algorithmic_bytes_per_cell": b_dom, "active_cells_per_launch": active_cells,
                    "avg_launch_ms": t_dom,
                    "other_kernel_ms": {"bgc_carbonate_kernel": t_car, "bgc_cell_kernel": t_cell,
                                        "ecosystem_par+cell+sweep": t_eco, "call_device_ms": t_call},
                    "whole_call_GBps": active_cells * (BYTES_CARBONATE[dk] + BYTES_ECOSYS[dk]) / (t_call * 1e-3) / 1e9,
                    "note": "the cell kernel moves `traffic` bytes (hand-off + zero fill on top of the algorithmic "
                            "ones) at the HBM rate a write-heavy stream reaches; the carbonate kernel beside it is "
                            "bound by fp64 VALU issue (96 %); see DESIGN.md section 4"}
        out = {"metric": "column-level tracer updates/sec", "value": value, "unit": "updates/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s: %d columns x %d levels per GPU, %s, native column-fastest layout, "
                                      "diagnostics=%s, %s"
                                      % (args.config, ncol, nz, "+".join(modules), dk,
                                         "cold pH bracket every step" if args.ph == "cold" else "warm pH brackets (previous step)"),
                          "columns_per_gpu": ncol, "levels": nz, "active_cells_per_gpu": active_cells,
                          "updates_per_step_all_gpus": updates_all, "parallelism": "column-range x%d, no collective" % world,
                          "columns_per_s": ncol * world * args.steps / elapsed,
                          "cells_per_s": updates_all / per_cell * args.steps / elapsed},
               "roofline": roofline}
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
