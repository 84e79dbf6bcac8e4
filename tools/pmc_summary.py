#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV output) into the
per-launch HBM traffic table that bench.py reports as roofline.traffic.

    python tools/pmc_summary.py <out.json> <tag>=<fetch_dir>,<write_dir> ...

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it
is doubled.  The factor was re-calibrated for this engine's access pattern (8 B per lane,
column-fastest rows) with the MACROS kernel on 1M x 60 cells, whose byte count is known:
reads 3.844 GB -> FETCH_SIZE 2.040 GB (x1.88), writes 8.16 GB issued -> WRITE_SIZE 7.81 GB.
"""
import collections
import csv
import glob
import json
import sys

FETCH_CORRECTION = 2.0


def per_kernel(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}


def short(name):
    """bgc_column_kernel<true, true, 0>(...) in namespaces -> bgc_column_kernel"""
    import re
    m = re.search(r"(\w+_kernel)\b", name.replace("(anonymous namespace)::", ""))
    if not m or not m.group(1).startswith(("bgc_", "fused_", "dms_", "macros_", "transpose_", "apply_", "synth_")):
        return None
    return m.group(1)


def main():
    import os
    rev = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ocean-bgc_amd", "KERNEL_REV")).read().strip()
    out = {"kernel_rev": rev, "fetch_correction": FETCH_CORRECTION, "unit": "bytes per launch", "runs": {}}
    for spec in sys.argv[2:]:
        tag, dirs = spec.split("=")
        fd, wd = dirs.split(",")
        fe, wr = per_kernel(fd), per_kernel(wd)
        run = {}
        for k in fe:
            s = short(k)
            if s:
                run[s] = {"read": fe[k] * FETCH_CORRECTION, "write": wr.get(k, 0.0),
                          "total": fe[k] * FETCH_CORRECTION + wr.get(k, 0.0)}
        out["runs"][tag] = run
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
