"""The branches the data must hit: counters in the CPU restatement (oracle/obgc_oracle.c, BR(...)) assert
that the synthetic workload plus four hand-made columns (tests/crafted.py) drive every data-dependent arm of
BGC_SourceSink / co2calc the scope table lists; the hand-made columns are also pinned against the unmodified
reference (bit for bit) and, on the GPU, against the oracle at the parity bars.

One arm is unreachable for physical inputs and is excluded on purpose: the solver's "iterate stopped moving" exit
(co2calc.F90:969/975) needs |dx| >= xacc = 1e-10 together with soln + dx == soln, i.e. [H+] > 1e6 mol/kg.
"""
import numpy as np
import pytest

import parity as P
from conftest import has_gpu
from crafted import crafted_columns

UNREACHABLE = {"newton_exit_same"}


def _args(syn):
    return (syn["temp"], syn["salt"], syn["zmid"], syn["dz"], syn["zbot"], syn["lat"], syn["kmax"], syn["fesed"], syn["dust"],
            syn["sw"])


def test_workload_reaches_every_branch(oracle, synth):
    oracle.branch_reset()
    for nc, nz, col0 in ((6000, 60, 0), (1500, 100, 500000)):
        syn = synth.generate(nc, nz, col0=col0)
        tr = synth.bgc_tracer_array(syn)
        z = np.zeros((nc, nz))
        w = oracle.bgc_source_sink(tr, *_args(syn), z, z)
        if nz == 60:     # the secondary timing mode: warm brackets on perturbed tracers
            oracle.bgc_source_sink(tr * 1.001, *_args(syn), w["ph_prev"], w["ph_prev_alt"])
    generator_only = oracle.branch_hits()
    syn, php = crafted_columns(synth)
    oracle.bgc_source_sink(synth.bgc_tracer_array(syn), *_args(syn), php, php)
    hits = oracle.branch_hits()
    assert len(hits) >= 40
    missing = [k for k, v in hits.items() if v == 0 and k not in UNREACHABLE]
    assert not missing, missing
    # what the hand-made columns are for
    assert {k for k, v in generator_only.items() if v == 0} - UNREACHABLE == {"qa_def_remains", "poc_hflux_zero", "iron_clamp",
                                                                             "newton_bracket_grow"}


def test_crafted_columns_oracle_equals_reference(oracle, ref, synth):
    syn, php = crafted_columns(synth)
    tr = synth.bgc_tracer_array(syn)
    a = oracle.bgc_source_sink(tr, *_args(syn), php, php)
    b = ref.bgc_source_sink(tr, *_args(syn), php, php)
    assert np.array_equal(a["tend"], b["tend"]) and np.array_equal(a["ph_prev"], b["ph_prev"])
    for grp in ("kc", "kca", "ca", "c"):
        for nm in a[grp]:
            assert np.array_equal(a[grp][nm], b[grp][nm], equal_nan=True), (grp, nm)
    assert np.isfinite(a["tend"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["device_colfast", "device_kfast", "host"])
def test_crafted_columns_gpu_vs_oracle(pkg, oracle, synth, path):
    if not has_gpu():
        pytest.fail("needs a GPU")
    from ocean_bgc_amd.modules import ColumnEngine
    E = ColumnEngine(0, path=path)
    syn, php = crafted_columns(synth)
    tr = synth.bgc_tracer_array(syn)
    got = E.BGC_SourceSink(tr, *_args(syn), php, php)
    want = oracle.bgc_source_sink(tr, *_args(syn), php, php)
    rep = P.compare_bgc(got, want, dz=syn["dz"])
    assert rep.ok(), "\n".join(rep.fail[:10])
