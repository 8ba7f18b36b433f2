#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repo root (gpurun), output under gpurun_out/<tag>/.
#   bash tests/tools/profile.sh r02_a [config]        (config: cfg2 (default) | cfg3 | ...: bench.py --config)
# 1. bench.py (defaults of the config) -> <tag>_bench.json   2. kernel trace + stats   3./4. FETCH_SIZE and WRITE_SIZE in their own passes
# 5. SQ instruction mix in its own pass.  Counter passes never share a run with a trace domain other than --kernel-trace.
set -u
tag=${1:-prof}; cfg=${2:-cfg2}; out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --config $cfg --steps 4 --warmup 1 --cpu-sample-pairs 0 --keep"
timeout 400 python bench.py --config $cfg --keep > $out/${tag}_bench.json 2> $out/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o r --output-format csv -- $B > $out/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o r --output-format csv -- $B > $out/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o r --output-format csv -- $B > $out/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $out/sq -o r --output-format csv -- $B > $out/sq.log 2>&1
python tests/tools/pmc_summary.py $out $tag
ls $out
