import json,glob,sys
for f in sorted(glob.glob("gpurun_out/ab_*.json")):
    try: r=json.load(open(f))
    except Exception as e: print(f, "bad", e); continue
    ro=r["roofline"]
    print("%-28s %7.2f ms %.3e upd/s  %s %.2f ms frac %.3f whole %.3f sched %s" % (f.split("ab_")[1][:-5], r["ms_per_step"], r["value"], ro["kernel"], ro["avg_launch_ms"], ro["frac"], ro["whole_step"]["frac"], ro["schedule"]), {k:round(v,2) for k,v in ro["kernel_ms"].items() if v>0.02})
