# Development aid: cycle counts per section of the host walk on the bench unit (serial executor built with -DAGX_WALK_PROF; no GPU work).
# Usage (repo root): GRAFT_REPO_ROOT=$PWD bash tests/tests/tools/walk_profile.sh
set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, subprocess
sys.path.insert(0,'tools'); sys.path.insert(0,'tests')
import agx_data as D
run=D.synth('/tmp/wp_run', seed=1000, chroms="4600000", pairs=1000000, coverage=5)
from hostsim import sim
# profile build of the serial executor
subprocess.check_call(["g++","-O2","-std=c++17","-fPIC","-shared","-pthread","-DAGX_WALK_PROF","-o",sim.LIB]+sim.SRC)
os.environ['AGX_WALK_REPEAT']='6'; os.environ['AGX_WALK_TIMING']='1'
sim.run(run+'/tmp',0,5,50,5)
PY
