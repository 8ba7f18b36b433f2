# instruction-mix counters of the node sweep for experimental library builds: bash tests/tools/sq_ab.sh name1 name2 ...   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  out=$PWD/gpurun_out/sqab_$v; rm -rf $out; mkdir -p $out
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $out -o r --output-format csv -- bash tests/tools/sweep_ab.sh $v > $out/log.txt 2>&1
  grep "node sweep ms" $out/log.txt
  python - $out <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/r_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]
    if "node_sweep<0>" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); seen.add(row["Dispatch_Id"])
for k, c in acc.items():
    print(k[:40], "launches", len(seen), {m: round(v / len(seen) / 1e6, 1) for m, v in sorted(c.items())})
PY
done
