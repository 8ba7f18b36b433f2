for w in 4 6 8 default; do for warm in 400000 150000; do
  if [ $w = default ]; then unset AGX_WALK_SPLIT_WALKERS; else export AGX_WALK_SPLIT_WALKERS=$w; fi
  AGX_WALK_SPLIT_WARMUP=$warm AGX_WALK_TIMING=1 timeout 200 python bench.py --keep --steps 10 --warmup 3 --cpu-sample-pairs 0 > /tmp/b.json 2> /tmp/b.err
  python - <<PY
import json; d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
e=open("/tmp/b.err").read()
print("walkers $w warm $warm: ms", d["ms_per_step"], "cpu ms/step", d["host_cpu_ms_per_step"], "gave up", e.count("gave up"), "differ", e.count("states differ"))
PY
done; done
