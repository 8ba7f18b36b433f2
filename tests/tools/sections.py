#!/usr/bin/env python3
"""Section times of a unit's first builds (exclusive builds with section events): python tests/tools/sections.py <tmp_dir> [unit] [reps]
AGX_LIB_PATH selects another build of the library (kernel A/B experiments)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aligngraph_amd as A  # noqa: E402
tmp, unit, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 6
keys = ("ms_upload_dev", "ms_prep", "ms_bin", "ms_node_sweep", "ms_node_big", "ms_edge_fast", "ms_edge_slow", "ms_compact", "ms_build_span")
acc = dict.fromkeys(keys, 0.0)
with A.Unit(k=5, insert_variation=50, coverage=5, flags=A.AGX_FLAG_TIME_SECTIONS) as u:
    u.load_files(tmp, unit)
    for i in range(reps + 1):
        try:
            u.upload(); u.build()
        except A.AgxError as e:
            print("build failed:", e)
        if i:
            st = u.stats()
            for k in keys:
                acc[k] += st[k] / reps
print(os.environ.get("AGX_LIB_PATH", "default"), {k: round(v, 3) for k, v in acc.items()})
