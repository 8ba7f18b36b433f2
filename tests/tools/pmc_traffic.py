#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a round's PMC pass (regenerate the file bench.py multiplies by every round it is used).

    python tests/tools/pmc_traffic.py profiles/r06_a cfg3        # reads profiles/r06_a_pmc_hbm.csv and profiles/r06_a_bench.json (tests/tools/profile.sh r06_a cfg3)

The node sweep's HBM bytes per tile-list entry = (FETCH_SIZE x 1.575 + WRITE_SIZE) of all agx_k_node_sweep<0> launches of the pass / the tile-list entries those launches swept.
The launches of a pass: (steps + warmup) jobs x the units of the configuration + the section pass's builds of the largest unit (bench.py: 21); a first build may be several
launches (r06: a window of tiles each), so the counters are summed over launches and divided by BUILDS.  Entries per unit = its hits x (entries / hits of the largest unit, which
the bench line carries).  The read factor 1.575 is r01's calibration for 4-byte-per-lane accesses (profiles/r01_f); the guide's factor for wide loads (2.0) gives the upper figure
that is also recorded."""
import csv
import json
import os
import sys

prefix, cfg = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = {r["kernel"]: r for r in csv.DictReader(open(prefix + "_pmc_hbm.csv"))}
sweep = next(r for k, r in rows.items() if "agx_k_node_sweep<0>" in k)
launches = int(sweep["launches"])
fetch, write = float(sweep["FETCH_SIZE_KB_per_launch"]) * 1024 * launches, float(sweep["WRITE_SIZE_KB_per_launch"]) * 1024 * launches
line = next(json.loads(l) for l in open(prefix + "_bench.json") if l.startswith("{"))
units = line["units"]
big = line["graph_largest_unit"]
per_hit = big["tile_entries"] / big["hits"]
jobs = 5                                   # profile.sh: --steps 4 --warmup 1
section_builds = 21                        # bench.py's section pass on the largest unit
entries = jobs * sum(u["hits"] for u in units.values()) * per_hit + section_builds * big["tile_entries"]
builds = jobs * len(units) + section_builds
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
tab = json.load(open(out_path)) if os.path.exists(out_path) else {}
per_entry = (fetch * 1.575 + write) / entries
tab.setdefault("node_sweep_bytes_per_tile_entry_by_config", {})[cfg] = round(per_entry, 2)
tab["node_sweep_bytes_per_tile_entry"] = round(per_entry, 2)
tab["_method_%s" % cfg] = ("%s_pmc_hbm.csv (bash tests/tools/profile.sh %s %s; tests/tools/pmc_traffic.py): %d launches of agx_k_node_sweep<0> = %d builds (first builds sweep a window of tiles per launch), "
                           "FETCH %.3f GB x 1.575 + WRITE %.3f GB per build over %.2f M tile-list entries per build; with the guide's factor 2.0 for wide loads: %.2f bytes per entry"
                           % (os.path.basename(prefix), os.path.basename(prefix), cfg, launches, builds, fetch / builds / 1e9, write / builds / 1e9, entries / builds / 1e6, (fetch * 2.0 + write) / entries))
# the same launches' instruction counters (profile.sh's SQ pass): vector and scalar instructions per list entry — what bench.py prints as roofline.issue, the roof the sweep is
# actually under (it is not an HBM-bound kernel)
sq_path = prefix + "_pmc_sq.csv"
if os.path.exists(sq_path):
    sq = next(r for r in csv.DictReader(open(sq_path)) if "agx_k_node_sweep<0>" in r["kernel"])
    n_sq = int(sq["launches"])
    tab.setdefault("node_sweep_insts_per_tile_entry_by_config", {})[cfg] = {
        "valu": round(float(sq["SQ_INSTS_VALU_per_launch"]) * n_sq / entries, 2), "salu": round(float(sq["SQ_INSTS_SALU_per_launch"]) * n_sq / entries, 2),
        "lds": round(float(sq["SQ_INSTS_LDS_per_launch"]) * n_sq / entries, 2), "vmem": round(float(sq["SQ_INSTS_VMEM_per_launch"]) * n_sq / entries, 2), "from": os.path.basename(sq_path)}
tab["_regenerated_from"] = os.path.basename(prefix)
json.dump(tab, open(out_path, "w"), indent=1)
print(json.dumps({k: tab[k] for k in ("node_sweep_bytes_per_tile_entry_by_config", "_method_%s" % cfg)}, indent=1))
