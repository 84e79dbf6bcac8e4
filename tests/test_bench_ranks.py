"""bench.py's own launcher: `python bench.py --gpus N` without a torchrun-style launcher must start N ranks (never a
silent single-rank run), refuse when the GPUs are not there, and refuse a --gpus that contradicts WORLD_SIZE."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, has_gpu

BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "OBGC_BENCH_SHARE_GPU", "OBGC_BENCH_FORCE_DIST")}
    env.update(kw)
    return env


def test_gpus_flag_contradicting_world_size_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


@pytest.mark.skipif(has_gpu(), reason="the refusal on a machine without enough GPUs is checked where there is none")
def test_more_ranks_than_gpus_is_refused_before_anything_runs():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr and "{" not in out.stdout


@pytest.mark.gpu
def test_two_self_launched_ranks_share_the_one_gpu():
    """Rehearsal of the N > 1 path on a one-GPU box: both ranks map to GPU 0 (OBGC_BENCH_SHARE_GPU=1; gloo on CPU tensors
    for the barrier and the reductions); the line must say n_gpus = 2 and count both ranks' updates."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    cmd = [sys.executable, BENCH, "--gpus", "2", "--config", "cfg4", "--columns", "20000", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=_env(OBGC_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    one = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--config", "cfg4", "--columns", "20000", "--steps", "3", "--warmup",
                          "1", "--no-cpu-baseline", "--no-extras"], env=_env(), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    # rank 1 owns the NEXT 20 000 global columns (weak scaling): twice the columns, not the same ones twice
    assert r["config"]["updates_per_step_all_gpus"] > 1.9 * r1["config"]["updates_per_step_all_gpus"]
    # strong scaling splits the configuration's columns instead
    st = subprocess.run(cmd + ["--scaling", "strong"], env=_env(OBGC_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert st.returncode == 0, st.stderr[-3000:]
    rs = json.loads([ln for ln in st.stdout.splitlines() if ln.startswith("{")][-1])
    assert rs["n_gpus"] == 2 and rs["scaling"] == "strong"
    assert rs["config"]["updates_per_step_all_gpus"] == r1["config"]["updates_per_step_all_gpus"]


@pytest.mark.gpu
def test_one_rank_through_the_rccl_process_group():
    """OBGC_BENCH_FORCE_DIST=1: a one-GPU run takes the code of an N-GPU rank -- init_process_group("nccl", device_id=...), the
    barriers and the device all-reduces of the wall time and the work -- so that a multi-GPU node is not its first run."""
    if not has_gpu():
        pytest.fail("needs a GPU")
    cmd = [sys.executable, BENCH, "--gpus", "1", "--config", "cfg4", "--columns", "20000", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=_env(OBGC_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["value"] > 0
    assert r["ranks"]["backend"] == "nccl" and r["ranks"]["world_size"] == 1
    assert r["ranks"]["ms_per_step_fastest"] == pytest.approx(r["ranks"]["ms_per_step_slowest"]) == pytest.approx(r["ms_per_step"])
    plain = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-3000:]
    r0 = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    assert "ranks" not in r0 and r0["config"]["updates_per_step_all_gpus"] == r["config"]["updates_per_step_all_gpus"]
