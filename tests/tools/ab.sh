#!/bin/bash
# kernel A/B on the GPU box: bench.py once per library variant given (paths relative to the repo root), prints the per-kernel times
for lib in "$@"; do
  if [ "$lib" = base ]; then unset AGX_LIB_PATH; else export AGX_LIB_PATH=$PWD/$lib; fi
  timeout 120 python bench.py --steps 96 --cpu-sample-pairs 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); b=j['breakdown_ms']
        print('$lib', 'ms/step %.3f'%j['ms_per_step'], 'sweep %.3f big %.3f prep %.3f bin %.3f edge %.3f/%.3f compact %.3f'%(b['ms_node_sweep'],b['ms_node_big'],b['ms_prep'],b['ms_bin'],b['ms_edge_fast'],b['ms_edge_slow'],b['ms_compact']), 'big_tiles', j['graph']['big_tiles'])
"
done
