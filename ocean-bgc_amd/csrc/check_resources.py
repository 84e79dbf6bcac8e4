#!/usr/bin/env python3
"""Build-time guard for the column-block kernel's register / scratch / LDS budget.

The kernel's speed hangs on two wavefronts per SIMD with no scratch memory: <= 256 VGPRs, no spills, <= 80 KB of LDS per
workgroup (two workgroups per CU).  That budget is reached with an internal LLVM switch (-mllvm -disable-machine-licm, see
the Makefile); if a compiler update drops or changes it, the build must fail here instead of shipping a kernel that
quietly spills.  Input: the remarks of  hipcc -Rpass-analysis=kernel-resource-usage  for obgc_ecosys.hip."""
import re
import sys

LIMITS = {"bgc_column_kernel": dict(vgprs=256, scratch=0, occupancy=2, lds=81920)}


def main(path):
    text = open(path, errors="replace").read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    seen, bad = 0, []
    for b in blocks:
        name = b.split()[0]
        key = next((k for k in LIMITS if k in name), None)
        if key is None:
            continue
        def num(label):
            m = re.search(re.escape(label) + r"\s*(\d+)", b)
            return int(m.group(1)) if m else None
        v, a, s, o, l = num("VGPRs:"), num("AGPRs:"), num("ScratchSize [bytes/lane]:"), num("Occupancy [waves/SIMD]:"), num("LDS Size [bytes/block]:")
        if None in (v, a, s, o, l):
            bad.append("%s: cannot read the resource remarks (compiler output changed?)" % name)
            continue
        seen += 1
        lim = LIMITS[key]
        if v + a > lim["vgprs"] or s > lim["scratch"] or o < lim["occupancy"] or l > lim["lds"]:
            bad.append("%s: VGPRs %d + AGPRs %d (<= %d), scratch %d B/lane (<= %d), occupancy %d (>= %d), LDS %d B (<= %d)"
                       % (name, v, a, lim["vgprs"], s, lim["scratch"], o, lim["occupancy"], l, lim["lds"]))
    if seen == 0:
        bad.append("no bgc_column_kernel instantiation found in %s" % path)
    if bad:
        sys.stderr.write("obgc: the column-block kernel left its resource budget:\n  " + "\n  ".join(bad) + "\n")
        return 1
    print("obgc: %d column-kernel instantiations inside the budget (<= 256 registers, no scratch, two workgroups per CU)" % seen)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
