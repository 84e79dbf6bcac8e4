"""world_size-2 `gloo` test of the N>1 path on CPU: the static column-range partition of bench.py,
its max/sum reductions, and shard invariance (a column's result does not depend on which rank or
batch offset it lands on -- there is no collective on the data path)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ncol, nz, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    import bench
    from ocean_bgc_amd import synth
    from oracle import orabind
    dist.init_process_group("gloo", rank=rank, world_size=world)
    col0, n = bench.shard(rank, world, ncol)
    syn = synth.generate(n, nz, col0=col0)
    O = orabind.Oracle()
    tr = synth.bgc_tracer_array(syn)
    z = np.zeros((n, nz))
    res = O.bgc_source_sink(tr, syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"],
                            syn["fesed"], syn["dust"], syn["sw"], z, z, diagnostics=False)
    elapsed, updates, fastest = bench.reduce_results(dist, 1.0 + rank, 30 * int(syn["kmax"].sum()), torch.device("cpu"))
    assert fastest == 1.0          # MIN over ranks: what shows a slow GPU next to the MAX
    q.put((rank, col0, n, res["tend"], res["ph_prev"], elapsed, updates, int(syn["kmax"].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_partition_and_reductions():
    world, ncol, nz = 2, 150, 60
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, ncol, nz, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # ranges tile the global column space without gaps or overlap
    assert [g[1] for g in got] == [0, ncol] and all(g[2] == ncol for g in got)
    # reductions: MAX of wall time, SUM of work
    assert all(g[5] == 2.0 for g in got)
    assert all(g[6] == 30.0 * (got[0][7] + got[1][7]) for g in got)
    # shard invariance: the two shards glued together equal one unsharded batch, bit for bit
    sys.path.insert(0, ROOT)
    from conftest import load_pkg
    load_pkg()
    from ocean_bgc_amd import synth
    from oracle import orabind
    syn = synth.generate(world * ncol, nz, col0=0)
    z = np.zeros((world * ncol, nz))
    ref = orabind.Oracle().bgc_source_sink(synth.bgc_tracer_array(syn), syn["temp"], syn["salt"], syn["zmid"], syn["dz"],
                                           syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"], syn["sw"], z, z,
                                           diagnostics=False)
    assert np.array_equal(np.concatenate([got[0][3], got[1][3]], axis=1), ref["tend"])
    assert np.array_equal(np.concatenate([got[0][4], got[1][4]], axis=0), ref["ph_prev"])
