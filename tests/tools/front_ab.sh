#!/bin/bash
# kernel A/B of a build's front on the GPU box: one 30 Mb unit of the cfg3 shape (5.2 M pairs), section times of exclusive first builds and rocprofv3 kernel times per library variant
#   bash tests/tools/front_ab.sh base aligngraph_amd/libagx_v4.so ...
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
python - <<'PY'
import sys; sys.path.insert(0, "tools")
import agx_data as D
D.synth("/tmp/ab_run", seed=1000, chroms="30427671", pairs=5110000, L=100, k=5, coverage=5, sam_seq=0, threads=16)
PY
for lib in "$@"; do
  if [ "$lib" = base ]; then unset AGX_LIB_PATH; tag=base; else export AGX_LIB_PATH=$PWD/$lib; tag=$(basename $lib .so); fi
  python tests/tools/sections.py /tmp/ab_run/tmp 0 8 2>/dev/null | tail -1
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ab/$tag -o r --output-format csv -- python tests/tools/sections.py /tmp/ab_run/tmp 0 8 > gpurun_out/ab/$tag.log 2>&1
  python - <<PY
import csv
rows = list(csv.reader(open("gpurun_out/ab/$tag/r_kernel_stats.csv")))
print("$tag", {r[0].split("(")[0][-22:]: round(float(r[3]) / 1e3, 1) for r in rows[1:14]})
PY
done
