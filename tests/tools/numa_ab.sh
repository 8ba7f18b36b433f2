# Development aid: does it matter which socket the host threads of a job run on?  (GPU box: two sockets, pinned memory near the GPU.)
lscpu | grep -i "numa\|socket\|model name" ; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
for f in /sys/bus/pci/devices/*/numa_node; do d=$(dirname $f); if [ -e $d/vendor ] && grep -q 0x1002 $d/vendor && grep -q "^0x03\|^0x12" $d/class; then echo "$d numa $(cat $f) class $(cat $d/class)"; fi; done
n0=$(cat /sys/devices/system/node/node0/cpulist); n1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 $n0 node1 $n1"
for cpus in all "$n0" "$n1"; do
  [ -z "$cpus" ] && continue
  if [ "$cpus" = all ]; then pre=""; else pre="taskset -c $cpus"; fi
  AGX_WALK_TIMING=1 $pre timeout 200 python bench.py --keep --steps 10 --warmup 3 --cpu-sample-pairs 0 > /tmp/o 2> /tmp/e
  python - <<PY
import json,re; d=json.loads(open("/tmp/o").read().strip().splitlines()[-1])
e=open("/tmp/e").read(); st=[float(x) for x in re.findall(r"stretch ([0-9.]+) ms\)", e)]
print("cpus $cpus: ms", d["ms_per_step"], "cpu", d["host_cpu_ms_per_step"], "stretch min/avg/max %.2f %.2f %.2f" % (min(st), sum(st)/len(st), max(st)))
PY
done
