"""bench.py's launcher-side placement logic against FAKE sysfs trees (no GPU, no HIP call): which CPUs rank r gets
(rank_cpu_sets / gpu_numa_nodes: the launcher's assumption, render-node order = HIP order) and what a rank does once it knows
its card's PCI address (verify_rank_placement: re-pin when the assumption was wrong).  The reference's one manager sits next
to its four PE arrays (bwa_mem_sw.v:162, batch_manager.v:343-348); SURVEY.md §8e names NUMA-local host threads as the
condition for >= 6x at 8 GPUs."""
import os

import pytest

import bench


def make_tree(root, node_cpus, gpu_nodes, bdfs=None):
    """node_cpus: {node: 'cpulist'}; gpu_nodes: NUMA node of GPU k (render node 128 + k); bdfs: its PCI address"""
    for node, cpus in node_cpus.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    (root / "class" / "drm").mkdir(parents=True)
    (root / "drivers" / "amdgpu").mkdir(parents=True)
    bdfs = bdfs or ["0000:%02x:00.0" % (0x11 + 0x10 * k) for k in range(len(gpu_nodes))]
    for k, (node, bdf) in enumerate(zip(gpu_nodes, bdfs)):
        dev = root / "bus" / "pci" / "devices" / bdf
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text("%d\n" % node)
        if node >= 0:
            (dev / "local_cpulist").write_text(node_cpus[node] + "\n")
        os.symlink(str(root / "drivers" / "amdgpu"), str(dev / "driver"))
        rd = root / "class" / "drm" / ("renderD%d" % (128 + k))
        rd.mkdir()
        os.symlink(str(dev), str(rd / "device"))
    return bdfs


def test_parse_and_print_cpu_lists():
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("") == []
    assert bench.cpu_list_str([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"


def test_eight_gpus_on_two_nodes(tmp_path):
    """4 + 4 cards on two nodes of 32 CPUs each: every rank gets 8 CPUs of ITS card's node, disjoint from the others"""
    make_tree(tmp_path, {0: "0-31", 1: "32-63"}, [0, 0, 0, 0, 1, 1, 1, 1])
    assert bench.gpu_numa_nodes(str(tmp_path)) == [0, 0, 0, 0, 1, 1, 1, 1]
    sets = bench.rank_cpu_sets(8, avail=range(64), sys_root=str(tmp_path))
    assert [len(s) for s in sets] == [8] * 8
    assert all(max(s) < 32 for s in sets[:4]) and all(min(s) >= 32 for s in sets[4:])
    flat = [c for s in sets for c in s]
    assert sorted(flat) == list(range(64))                      # disjoint and complete
    # fewer ranks than cards: ranks 0..1 share node 0's CPUs between them
    sets = bench.rank_cpu_sets(2, avail=range(64), sys_root=str(tmp_path))
    assert sets[0] == list(range(0, 16)) and sets[1] == list(range(16, 32))


def test_interleaved_cards_and_a_restricted_process(tmp_path):
    """cards alternate between the nodes, and the process may only use half of every node (a container's cpuset)"""
    make_tree(tmp_path, {0: "0-15", 1: "16-31"}, [0, 1, 0, 1])
    avail = list(range(0, 8)) + list(range(16, 24))
    sets = bench.rank_cpu_sets(4, avail=avail, sys_root=str(tmp_path))
    assert sets[0] == [0, 1, 2, 3] and sets[2] == [4, 5, 6, 7]
    assert sets[1] == [16, 17, 18, 19] and sets[3] == [20, 21, 22, 23]


def test_unknown_nodes_are_dealt_round_robin(tmp_path):
    """numa_node = -1 (the kernel does not say, e.g. a VM): the ranks are dealt over the nodes that exist"""
    make_tree(tmp_path, {0: "0-7", 1: "8-15"}, [-1, -1, -1, -1])
    sets = bench.rank_cpu_sets(4, avail=range(16), sys_root=str(tmp_path))
    assert all(s for s in sets)
    assert sorted(c for s in sets for c in s) == list(range(16))
    assert set(sets[0]) <= set(range(8)) and set(sets[1]) <= set(range(8, 16))


def test_fewer_cpus_than_ranks(tmp_path):
    """8 ranks, one node, 4 usable CPUs: nobody is left without a CPU (the sets overlap instead)"""
    make_tree(tmp_path, {0: "0-3"}, [0] * 8)
    sets = bench.rank_cpu_sets(8, avail=range(4), sys_root=str(tmp_path))
    assert all(s and set(s) <= {0, 1, 2, 3} for s in sets)


def test_no_sysfs_at_all(tmp_path):
    sets = bench.rank_cpu_sets(2, avail=range(8), sys_root=str(tmp_path / "nothing"))
    assert sets == [[0, 1, 2, 3], [4, 5, 6, 7]]


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="needs Linux affinity calls")
def test_a_rank_re_pins_itself_when_hip_order_differs_from_render_node_order(tmp_path):
    """The launcher gave rank 0 the CPUs of render node 128's card; HIP's device 0 turns out to be the card on the OTHER
    node.  The rank re-pins onto its card's node, taking its share of what the process was allowed before the launcher
    narrowed it; a rank that already sits next to its card is left alone."""
    me = sorted(os.sched_getaffinity(0))
    if len(me) < 4:
        pytest.skip("needs 4 CPUs")
    half = len(me) // 2
    lo, hi = me[:half], me[half:]
    bdfs = make_tree(tmp_path, {0: bench.cpu_list_str(lo), 1: bench.cpu_list_str(hi)}, [0, 1])
    try:
        os.sched_setaffinity(0, set(lo))                        # the launcher's (wrong) guess: node 0
        node, repinned, cpus = bench.verify_rank_placement(bdfs[1], me, (0, 1), sys_root=str(tmp_path))
        assert (node, repinned) == (1, True) and cpus == hi and sorted(os.sched_getaffinity(0)) == hi
        node, repinned, cpus = bench.verify_rank_placement(bdfs[1], me, (0, 1), sys_root=str(tmp_path))
        assert (node, repinned) == (1, False) and cpus == hi     # now it is where its card is
        # two ranks whose cards share node 1: each takes its half of the node's CPUs
        os.sched_setaffinity(0, set(lo))
        node, repinned, cpus = bench.verify_rank_placement(bdfs[1], me, (1, 2), sys_root=str(tmp_path))
        assert repinned and cpus == hi[len(hi) // 2:len(hi) // 2 * 2]
        # nothing known about the card: leave the rank alone
        os.sched_setaffinity(0, set(lo))
        node, repinned, cpus = bench.verify_rank_placement("0000:ff:00.0", me, (0, 1), sys_root=str(tmp_path))
        assert (node, repinned) == (-1, False) and cpus == lo
    finally:
        os.sched_setaffinity(0, set(me))
