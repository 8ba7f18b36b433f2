# timing of the node sweep with experimental library builds (build/variants/libagx_<name>.so): bash tests/tools/sweep_ab.sh name1 name2 ...
for v in "$@"; do
  L=$PWD/build/variants/libagx_$v.so; [ "$v" = base ] && L=$PWD/aligngraph_amd/libagx.so
  AGX_LIB_PATH=$L python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import aligngraph_amd as A, agx_data as D
extra = dict(kv.split("=") for kv in os.environ.get("AGX_AB_SYNTH", "").split(",") if kv)      # e.g. AGX_AB_SYNTH=read_indel=0,read_clip=0: another workload
run = "/tmp/sweep_ab_run" + "".join("_%s%s" % kv for kv in sorted(extra.items()))
if not os.path.exists(os.path.join(run, "synth_meta.txt")):
    D.synth(run, seed=1000, chroms="30427671", pairs=5100000, L=100, k=5, coverage=5, sam_seq=0, threads=16, **extra)
with A.Unit(k=5, insert_variation=50, coverage=5, device=0, flags=A.AGX_FLAG_TIME_SECTIONS) as u:
    u.load_files(os.path.join(run, "tmp"), 0)
    t = []
    for i in range(6):
        u.upload()
        try:
            u.build()
        except Exception as e:
            print("build error (experiment):", e); break
        st = u.stats(); t.append(st["ms_node_sweep"])
    print(os.path.basename(os.environ["AGX_LIB_PATH"]), "node sweep ms:", ["%.3f" % x for x in t], "entries", st["n_tile_entries"])
    print("   sections of the last build (ms):", {k: round(st[k], 3) for k in ("ms_prep", "ms_bin", "ms_node_sweep", "ms_node_big", "ms_edge_fast", "ms_edge_slow", "ms_compact", "ms_build_span")})
PY
done
