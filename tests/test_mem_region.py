"""Device-memory policy of the engine (aligngraph_amd/csrc/agx_mem.h: cache of whole blocks, per-device region, a unit's arena) against a made-up HIP runtime.

No reference counterpart (the reference keeps its graph in std::vector).  The policy only ever runs on a GPU box, where a mistake shows up as a job that takes seconds
instead of milliseconds (r04: 19.5 s per whole-human job before the region, 2.4 s per quarter-size job before small blocks came out of it) or as a unit that waits
for itself; tests/memsim/region_check.cpp replays those situations with counted driver calls and addresses that are never touched."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "memsim", "region_check.cpp")
ROCM_INC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")


@pytest.mark.skipif(not os.path.exists(os.path.join(ROCM_INC, "hip", "hip_runtime_api.h")), reason="no HIP headers")
def test_region_cache_and_arena_against_a_made_up_runtime(tmp_path):
    exe = str(tmp_path / "region_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I" + ROCM_INC, SRC, "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)      # (a policy that makes a small block wait for room hangs here)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"ok"), p.stdout[-400:]
