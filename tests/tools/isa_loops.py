"""Development aid (no GPU): instruction counts of a gfx950 kernel and of its loops, from the compiler's assembly.

    python tests/tools/isa_loops.py agx_k_node_sweep            # every kernel whose mangled name contains the text
    python tests/tools/isa_loops.py agx_k_node_sweep -D AGX_WP_POS=2

A loop = a backward branch (s_cbranch_* / s_branch to a label that lies above it); its body = the lines between the label and the branch, nested loops included.
Classes: VALU (v_*), SALU (s_* without memory, branch, wait), SMEM (s_load / s_buffer_load), VMEM (global_ / buffer_ / flat_ / scratch_), LDS (ds_*), branch, wait.
Static counts say what a loop ISSUES per trip when every branch in it falls through — they are not a time (HISTORY.md r04: variants with fewer instructions ran
slower because their loads no longer overlapped), but they tell at once whether a change removed anything from the loop that runs per tile-list entry.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "aligngraph_amd", "csrc", "agx_kernels.hip")


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_scratch")):
        return "SMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    return "other"


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    want = sys.argv[1]
    defs = []
    a = sys.argv[2:]
    while a:
        if a[0] == "-D" and len(a) > 1:
            defs.append("-D" + a[1]); a = a[2:]
        else:
            sys.exit(__doc__)
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + defs + [SRC, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    # what the kernel's descriptor says about its registers and its SCRATCH memory: a non-zero private segment means something that should be in registers is not (r05:
    # a record kept as a struct cost agx_k_tile_fill a factor of two, and every timing read while it was there was wrong)
    meta = {}
    blocks = "\n".join(text).split("  - .agpr_count:")[1:]
    for blk in blocks:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        meta[g("name")] = "vgpr %s, sgpr %s (spilled %s), LDS %s B, scratch (private segment) %s B" % (g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"))
    # kernels: from "<name>:" to s_endpgm
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):", text[i])
        if not (m and want in m.group(1)):
            i += 1
            continue
        name = m.group(1)
        body = []
        i += 1
        while i < len(text) and "s_endpgm" not in text[i]:
            body.append(text[i]); i += 1
        labels, ins = {}, []
        for line in body:
            ls = line.strip()
            lm = re.match(r"^(\.LBB\w+):", ls)
            if lm:
                labels[lm.group(1)] = len(ins)
                continue
            if not ls or ls.startswith((";", ".", "//")):
                continue
            op = ls.split()[0]
            ins.append((op, ls))
        total = collections.Counter(classify(op) for op, _ in ins)
        print("%s\n  %s\n  whole kernel: %d instructions  %s" % (name, meta.get(name, ""), len(ins), dict(total)))
        loops = []
        for at, (op, ls) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = ls.split()[-1]
                if tgt in labels and labels[tgt] <= at:
                    loops.append((labels[tgt], at, tgt))
        # loops that share a head label (several back edges to one label) are one loop: the widest extent
        ext = {}
        for lo, hi, tgt in loops:
            ext[tgt] = (lo, max(hi, ext.get(tgt, (lo, hi))[1]))
        spans = sorted(((lo, hi, tgt) for tgt, (lo, hi) in ext.items()), key=lambda t: (t[0], -t[1]))
        for lo, hi, tgt in spans:
            depth = sum(1 for l2, h2, _ in spans if l2 <= lo and hi <= h2) - 1
            inside = [(l2, h2) for l2, h2, _ in spans if lo <= l2 and h2 <= hi and (l2, h2) != (lo, hi)]
            covered = set()
            for l2, h2 in inside:
                covered.update(range(l2, h2 + 1))
            own = collections.Counter(classify(ins[j][0]) for j in range(lo, hi + 1) if j not in covered)
            print("  %sloop %-12s %5d instructions, %4d of its own  %s" % ("  " * depth, tgt, hi - lo + 1, sum(own.values()), dict(own)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
