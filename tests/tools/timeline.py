#!/usr/bin/env python3
"""Device timeline of the last timed job of a rocprofv3 trace of bench.py (kernels and copies in start order, offsets from the job's first upload copy):
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-sample-pairs 0 --keep
    python tests/tools/timeline.py gpurun_out/tl [n_units] > profiles/<tag>_timeline.txt"""
import csv, glob, os, sys
out = sys.argv[1]; n_units = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def rows(pattern):
    f = glob.glob(os.path.join(out, "**", pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "")) for r in rows("*kernel_trace.csv")]
c = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), "") for r in rows("*memory_copy_trace.csv")]
sweeps = sorted(x for x in k if "node_sweep<0>" in x[2])
job = sweeps[-n_units - 21:-21] if len(sweeps) > n_units + 21 else sweeps[-n_units:]
h2d = sorted(x for x in c if "HOST_TO_DEVICE" in x[2].upper())
t0 = min((x[0] for x in h2d if x[1] <= job[0][0] + 2_000_000 and job[0][0] - x[0] < 40_000_000), default=job[0][0])
t1 = job[-1][1] + 12_000_000
ev = sorted([x for x in k if t0 <= x[0] <= t1] + [x for x in c if t0 <= x[0] <= t1])
small = 0.0
for s, e, name, q in ev:
    d = (e - s) / 1e6
    if d < 0.05 and "node_sweep" not in name:
        small += d
        continue
    print("%8.2f ms  +%7.3f ms  q%-3s %s" % ((s - t0) / 1e6, d, q, name[:70]))
print("(%.2f ms of commands shorter than 0.05 ms not listed)" % small)
