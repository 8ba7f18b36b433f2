#!/bin/bash
# On the GPU box: generate the cfg3 inputs and time the fast loaders for one unit and for all five side by side at several thread counts.
set -e
mkdir -p gpurun_out build
g++ -O3 -std=c++17 -pthread -o build/loadbench tests/tools/loadbench.cpp aligngraph_amd/csrc/agx_host.cpp aligngraph_amd/csrc/agx_walk.cpp aligngraph_amd/csrc/agx_load.cpp
g++ -O2 -std=c++17 -pthread -o build/agx_synth tools/agx_synth.cpp
uname -r; cat /sys/kernel/mm/transparent_hugepage/enabled
build/agx_synth --out /tmp/lb --seed 1000 --chroms 30427671,19698289,23459830,18585056,26975502 --part 1 --pairs 20000000 --L 100 --k 5 --coverage 5 --sam-seq 0 --threads 32 > /dev/null
NODE_CPUS=$(cat /sys/devices/system/node/node1/cpulist)
for T in ${THREADS:-8 16 32 64}; do echo "== one unit, $T threads"; AGX_LOAD_THREADS=$T AGX_LOAD_TIMING=1 taskset -c $NODE_CPUS build/loadbench /tmp/lb/tmp 0 5 2 2>&1 | grep "agx load\|fast," | tail -6; done
for T in ${THREADS5:-8 16 25}; do echo "== five units, $T threads each"; AGX_LOAD_THREADS=$T AGX_LOAD_TIMING=1 taskset -c $NODE_CPUS build/loadbench /tmp/lb/tmp 0,1,2,3,4 5 2 2>&1 | grep "agx load\|fast," | tail -24; done
echo "== five units, five PROCESSES, 12 threads each"
for u in 0 1 2 3 4; do AGX_LOAD_THREADS=12 taskset -c $NODE_CPUS build/loadbench /tmp/lb/tmp $u 5 3 2>&1 | grep "^fast, 1" | tail -1 & done; wait
