"""Host-side helpers that decide how many threads the loaders and walkers use (agx_host.cpp: usable_cpus / cgroup_cpu_quota).

No reference counterpart: the reference runs its units on a fixed four threads (AG:2293-2358); this engine sizes its loaders and walkers by the
CPUs the process may really use, which inside a container is the control group's quota, not the host's thread count.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))
import sim  # noqa: E402


def _tree(root, files):
    for rel, text in files.items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)


def _quota(tmp_path, cgroup_text, files):
    proc = tmp_path / "proc_cgroup"
    proc.write_text(cgroup_text)
    sysroot = tmp_path / "sys"
    sysroot.mkdir(exist_ok=True)
    _tree(sysroot, files)
    return sim.cgroup_quota(proc, sysroot)


def test_v2_quota_at_the_mount_root(tmp_path):
    assert _quota(tmp_path, "0::/\n", {"cpu.max": "1600000 100000\n"}) == 16


def test_v2_no_quota(tmp_path):
    assert _quota(tmp_path, "0::/\n", {"cpu.max": "max 100000\n"}) == 0
    assert _quota(tmp_path, "0::/\n", {}) == 0


def test_v2_nested_group_takes_the_tightest_ancestor(tmp_path):
    files = {"cpu.max": "max 100000\n", "a/cpu.max": "800000 100000\n", "a/b/cpu.max": "max 100000\n", "a/b/c/cpu.max": "1250000 100000\n"}
    assert _quota(tmp_path, "0::/a/b/c\n", files) == 8
    files["a/b/c/cpu.max"] = "250000 100000\n"
    assert _quota(tmp_path, "0::/a/b/c\n", files) == 3            # 2.5 CPUs rounds up


def test_v2_group_the_mount_does_not_show(tmp_path):
    # a container sees its own group as the mount's root: the path /proc names is not there
    assert _quota(tmp_path, "0::/kubepods/pod1/ctr\n", {"cpu.max": "400000 100000\n"}) == 4


def test_v1_quota(tmp_path):
    text = "4:memory:/x\n3:cpuset:/jobs\n2:cpuacct:/\n1:cpu:/jobs/j1\n0::/\n"
    files = {"cpu/cpu.cfs_quota_us": "-1\n", "cpu/cpu.cfs_period_us": "100000\n",
             "cpu/jobs/cpu.cfs_quota_us": "3200000\n", "cpu/jobs/cpu.cfs_period_us": "100000\n",
             "cpu/jobs/j1/cpu.cfs_quota_us": "-1\n", "cpu/jobs/j1/cpu.cfs_period_us": "100000\n"}
    assert _quota(tmp_path, text, files) == 32
    files["cpu/jobs/j1/cpu.cfs_quota_us"] = "150000\n"
    assert _quota(tmp_path, text, files) == 2


def test_v1_joint_controller_mount_and_hidden_group(tmp_path):
    text = "5:cpu,cpuacct:/docker/abc\n"
    files = {"cpu,cpuacct/cpu.cfs_quota_us": "1600000\n", "cpu,cpuacct/cpu.cfs_period_us": "100000\n"}
    assert _quota(tmp_path, text, files) == 16


def test_v1_and_v2_side_by_side_take_the_tighter(tmp_path):
    text = "1:cpu:/\n0::/g\n"
    files = {"cpu/cpu.cfs_quota_us": "2400000\n", "cpu/cpu.cfs_period_us": "100000\n", "g/cpu.max": "600000 100000\n"}
    assert _quota(tmp_path, text, files) == 6


def test_garbage_is_no_quota(tmp_path):
    assert _quota(tmp_path, "not a cgroup line\n\n0::/../../etc\n", {"cpu.max": "banana\n"}) == 0


def test_usable_cpus_is_within_the_affinity_mask():
    n = sim.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
