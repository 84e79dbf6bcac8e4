#!/usr/bin/env python3
"""Instruction-class counts per kernel from hipcc -S output:  tools/isa_stats.py file.s [name substring ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s*; @", txt, re.M)]
for i, (pos, name) in enumerate(starts):
    if want and not any(w in name for w in want):
        continue
    end = txt.find(".Lfunc_end", pos)
    body = txt[pos:end]
    ins = collections.Counter()
    for ln in body.split("\n"):
        ln = ln.strip()
        if not ln or ln[0] in ".;/" or ln.endswith(":"):
            continue
        op = ln.split()[0]
        for pre, key in (("scratch_", "scratch"), ("flat_", "flat"), ("ds_", "ds"), ("global_load", "gload"), ("global_store", "gstore"),
                         ("buffer_", "buffer"), ("s_barrier", "barrier"), ("s_waitcnt", "waitcnt"), ("v_readlane", "lane_rw"),
                         ("v_writelane", "lane_rw"), ("v_accvgpr", "acc_mov"), ("s_load", "sload")):
            if op.startswith(pre):
                ins[key] += 1
                break
        else:
            ins["valu_f64" if (op.startswith("v_") and "f64" in op) else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other"] += 1
    short = re.sub(r"_ZN4obgc12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"EEvNS_13Bgc.*", "", short)
    print("%-34s total %6d " % (short[:34], sum(ins.values())), dict(sorted(ins.items())))
