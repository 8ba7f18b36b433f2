"""Golden checksums for configurations too large to commit as files — run in the build container only (needs /root/reference).

Inputs are regenerated from the seed (tools/agx_synth --threads: the same files for every thread count), replayed through the REAL
reference binary (-O2 build of /root/reference/AlignGraph/AlignGraph.cpp, oracle/Makefile; the README build agrees with it on every
small fixture) and the md5 + length of the three per-unit output files are written to tests/golden/big_md5.json together with the md5 of
the generated INPUT files (so that a test can tell "generator changed" from "engine differs").

The cases are described in tests/golden/big_cases.py (cfg2 = configs[1] at full size; batch2 = SURVEY §8(c)(vi), the batch boundary).
"""
import hashlib
import json
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import harness as H  # noqa: E402

from big_cases import CASES, generate  # noqa: E402

INPUTS = ("_genome.0.fa", "_contigs.fa", "_contigs_genome.0.psl", "_reads.fa", "_reads_genome.0.bowtie")


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    H.build()
    if not H.have_reference(True):
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    out_path = os.path.join(HERE, "big_md5.json")
    table = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for name, case in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        kw = case["synth"]
        run = generate(name, "/tmp/golden_big_" + name)
        meta = H.read_meta(run)
        t0 = time.time()
        outs, secs = H.run_reference(run, opt=True)
        assert len(outs) == 1
        if name == "batch2":                           # the fixture must pin the rule: without the batch boundary the output has to differ
            mine = H.run_oracle(os.path.join(run, "tmp"), 0, meta["k"], meta["insert_variation"], meta["coverage"])
            whole = H.run_oracle(os.path.join(run, "tmp"), 0, meta["k"], meta["insert_variation"], meta["coverage"], batch=2000000)
            assert all(mine[k2] == outs[0][k2] for k2 in mine), "oracle differs from the reference"
            assert whole["pre"] != outs[0]["pre"], "keeping the boundary pair does not change the output: the fixture pins nothing"
            print("   pre-extended records with / without the batch boundary:", outs[0]["pre"].count(b">"), whole["pre"].count(b">"))
        entry = {"synth": kw, "edit": case["edit"], "k": meta["k"], "insert_variation": meta["insert_variation"], "coverage": meta["coverage"],
                 "inputs_md5": {fn: md5_file(os.path.join(run, "tmp", fn)) for fn in INPUTS},
                 "reference_seconds_stages_1_5": round(secs, 1),
                 "expected": {key: {"md5": hashlib.md5(outs[0][key]).hexdigest(), "bytes": len(outs[0][key])} for key in ("initial", "pre", "extended")}}
        table[name] = entry
        print(name, "reference %.1f s (wall %.1f s)" % (secs, time.time() - t0), entry["expected"])
        json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
        shutil.rmtree(run)


if __name__ == "__main__":
    main()
