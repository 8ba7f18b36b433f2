#!/usr/bin/env python3
"""Copy / kernel overlap from a rocprofv3 trace of bench.py (run on the GPU box):
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/ovl -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --cpu-sample-pairs 0 --keep
    python tests/tools/overlap.py gpurun_out/ovl > profiles/<tag>_overlap.txt
Reports, for the last job of the run: the host-to-device copies (the units' uploads), the kernels, how much of the copy time had a kernel of
ANOTHER unit's build running beside it, and the busy fractions of the job."""
import csv
import glob
import os
import sys

out = sys.argv[1]


def rows(pattern):
    f = glob.glob(os.path.join(out, "**", pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv")]
c = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), int(float(r.get("Bytes", r.get("Size", 0)) or 0))) for r in rows("*memory_copy_trace.csv")]      # (this rocprofv3 writes no byte counts: bench.py's upload_MB has them)
sweeps = sorted(x for x in k if "node_sweep<0>" in x[2])
assert sweeps, "no node sweep in the trace"
# the last job = the last n_units sweeps that are preceded by a gap; take the last 5 main sweeps (cfg3) or fewer
n_units = int(sys.argv[2]) if len(sys.argv) > 2 else 5
job_sweeps = sweeps[-n_units - 21:-21] if len(sweeps) > n_units + 21 else sweeps[-n_units:]      # (bench.py's section pass adds 21 builds of the largest unit behind the timed steps)
h2d_all = sorted(x for x in c if "HOST_TO_DEVICE" in x[2].upper() or "H2D" in x[2].upper())
t_lo = min((x[0] for x in h2d_all if x[1] <= job_sweeps[0][0] + 2_000_000 and job_sweeps[0][0] - x[0] < 25_000_000), default=job_sweeps[0][0])
t_hi = job_sweeps[-1][1] + 3_000_000
h2d = [x for x in h2d_all if t_lo <= x[0] <= t_hi]
kern = sorted(x for x in k if t_lo <= x[0] <= t_hi)


def union(iv):
    iv = sorted((a, b) for a, b, *_ in iv)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + (cur_b - cur_a if cur_b is not None else 0)


def overlap(a_iv, b_iv):
    b_iv = sorted((a, b) for a, b, *_ in b_iv)
    tot = 0
    for a0, a1, *_ in a_iv:
        for b0, b1 in b_iv:
            lo, hi = max(a0, b0), min(a1, b1)
            if hi > lo:
                tot += hi - lo
    return tot


span = (t_hi - t_lo) / 1e6
copy_ms, kern_ms = union(h2d) / 1e6, union(kern) / 1e6
both = overlap([(a, b) for a, b, *_ in h2d], kern) / 1e6
bytes_up = sum(x[3] for x in h2d)
print("job window %.1f ms (first upload copy of the job -> 3 ms after its last node sweep); %d host-to-device copies" % (span, len(h2d)))
print("host-to-device copies busy %.1f ms (%.0f %% of the window); kernels busy %.1f ms (%.0f %%)" % (copy_ms, 100 * copy_ms / span, kern_ms, 100 * kern_ms / span))
gaps = sorted(h2d)
idle = [(gaps[i + 1][0] - max(x[1] for x in gaps[:i + 1])) / 1e6 for i in range(len(gaps) - 1)]
print("idle PCIe between consecutive upload copies inside the window: %s ms" % ", ".join("%.1f" % g for g in idle if g > 0.5))
print("kernel time that ran BESIDE an upload copy: %.1f ms = %.0f %% of the kernel time (the kernels of unit n run while unit n+1 uploads)" % (both, 100 * both / kern_ms if kern_ms else 0))
for s0, s1, _ in job_sweeps:
    during = [x for x in h2d if x[0] < s1 and x[1] > s0]
    print("  node sweep %.2f ms at +%.1f ms: %d upload copies in flight beside it" % ((s1 - s0) / 1e6, (s0 - t_lo) / 1e6, len(during)))
