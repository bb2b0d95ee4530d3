"""CPU placement of the host threads that serve one GPU (swimm_amd/csrc/host/affinity.h): the plan is a pure function of
sysfs and the allowed CPUs, so it is checked here against fake sysfs trees -- an 8-GPU, two-socket, SMT-2 node like the
MI355X boxes, a tree that says nothing, a restricted cpuset -- and applied to this process for real."""
import os

import pytest

from swimm_amd import host


def make_sysfs(root, sockets=2, cores_per_socket=64, smt=2, gpus_per_socket=4, name_gpus=True):
    """fake /sys: CPUs 0..S*C-1 are the first hardware threads, S*C.. their siblings; GPU d hangs off socket d // gpus_per_socket"""
    ncore = sockets * cores_per_socket
    for cpu in range(ncore * smt):
        core = cpu % ncore
        d = os.path.join(root, "devices", "system", "cpu", f"cpu{cpu}", "topology")
        os.makedirs(d)
        open(os.path.join(d, "thread_siblings_list"), "w").write(",".join(str(core + k * ncore) for k in range(smt)) + "\n")
    bdfs = []
    for g in range(sockets * gpus_per_socket):
        bdf = f"0000:{0x0c + 0x10 * g:02x}:00.0"
        bdfs.append(bdf)
        if not name_gpus:
            continue
        s = g // gpus_per_socket
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        os.makedirs(d)
        lo = s * cores_per_socket
        open(os.path.join(d, "local_cpulist"), "w").write(",".join(f"{lo + k * ncore}-{lo + k * ncore + cores_per_socket - 1}" for k in range(smt)) + "\n")
    return bdfs, list(range(ncore * smt))


def test_eight_gpus_two_sockets_share_their_socket_by_whole_cores(tmp_path):
    bdfs, allowed = make_sysfs(str(tmp_path))
    plans = [host.affinity_plan(bdfs, d, allowed, str(tmp_path)) for d in range(8)]
    for d, p in enumerate(plans):
        s, k = d // 4, d % 4
        cores = list(range(s * 64 + k * 16, s * 64 + (k + 1) * 16))
        assert sorted(p) == cores + [c + 128 for c in cores], (d, host._cpulist(p))      # 16 cores of the device's socket, both hardware threads
    flat = [c for p in plans for c in p]
    assert len(flat) == len(set(flat)) == 256                                           # nobody sits on anybody else, nothing is left idle
    assert host._cpulist(plans[0]) == "0-15,128-143" and host._cpulist(plans[5]) == "80-95,208-223"


def test_two_ranks_of_eight_devices_only_count_the_devices_given(tmp_path):
    bdfs, allowed = make_sysfs(str(tmp_path))
    # a 2-GPU job on devices 0 and 1 (same socket): each gets half of that socket's cores
    p0, p1 = (host.affinity_plan(bdfs[:2], d, allowed, str(tmp_path)) for d in range(2))
    assert host._cpulist(p0) == "0-31,128-159" and host._cpulist(p1) == "32-63,160-191"
    # devices 0 and 4 (one per socket): each its whole socket
    q0, q1 = (host.affinity_plan([bdfs[0], bdfs[4]], d, allowed, str(tmp_path)) for d in range(2))
    assert host._cpulist(q0) == "0-63,128-191" and host._cpulist(q1) == "64-127,192-255"


def test_unknown_devices_get_an_even_share_of_the_allowed_cpus(tmp_path):
    bdfs, allowed = make_sysfs(str(tmp_path), name_gpus=False)
    plans = [host.affinity_plan(bdfs, d, allowed, str(tmp_path)) for d in range(8)]
    assert [len(p) for p in plans] == [32] * 8
    assert host._cpulist(plans[0]) == "0-15,128-143" and host._cpulist(plans[7]) == "112-127,240-255"
    # no sysfs at all (thread siblings unknown either): plain even split of the list
    plans = [host.affinity_plan(["", None, "x", ""], d, list(range(8)), str(tmp_path / "nothing")) for d in range(4)]
    assert plans == [[0, 1], [2, 3], [4, 5], [6, 7]]


def test_a_restricted_cpuset_is_respected(tmp_path):
    bdfs, _ = make_sysfs(str(tmp_path))
    allowed = list(range(0, 16)) + list(range(64, 72))          # what a container hands out
    plans = [host.affinity_plan(bdfs, d, allowed, str(tmp_path)) for d in range(8)]
    for p in plans:
        assert p and set(p) <= set(allowed)
    assert sorted(c for p in plans[:4] for c in p) == list(range(16)) and sorted(c for p in plans[4:] for c in p) == list(range(64, 72))
    # more devices than cores: they double up instead of getting nothing
    plans = [host.affinity_plan(bdfs[:4], d, [0, 1], str(tmp_path)) for d in range(4)]
    assert plans == [[0], [1], [0], [1]]


def test_apply_binds_this_thread_and_new_threads_inherit(tmp_path):
    import threading
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 2:
        pytest.skip("one CPU only")
    try:
        mine = host.affinity_plan(["", ""], 1, before, str(tmp_path))      # the second half of what we have
        host.affinity_apply(mine)
        assert sorted(os.sched_getaffinity(0)) == mine
        seen = []
        t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))
        t.start(); t.join()
        assert seen == [mine]
    finally:
        os.sched_setaffinity(0, before)
