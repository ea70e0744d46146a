"""watsor_amd/numa.py: where a detector process pins itself (no GPU needed: a fake sysfs tree and the parsing rules)."""
import os

from watsor_amd import numa


def test_cpulist_format():
    assert numa.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert numa.parse_cpulist("") == set()
    assert numa.parse_cpulist("5") == {5}


def test_unknown_device_pins_nothing():
    before = os.sched_getaffinity(0)
    info = numa.pin_to_gpu_node(63)                    # no such GPU here (and no GPU at all in the build container)
    assert info["pinned"] is False and info["cpus"] == 0 and info["numa_node"] == -1
    assert os.sched_getaffinity(0) == before


def test_pins_to_the_local_cpus_of_the_gpu(tmp_path, monkeypatch):
    """A fake /sys/bus/pci/devices entry for the device: the process ends up on (its allowed subset of) local_cpulist."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        import pytest
        pytest.skip("one CPU: nothing to narrow")
    local = allowed[: len(allowed) // 2]
    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "numa_node").write_text("1\n")
    (d / "local_cpulist").write_text(",".join(str(c) for c in local) + ",4095\n")     # (a CPU this process may not use is ignored)
    monkeypatch.setattr(numa, "device_pci_bus_id", lambda device: "0000:c1:00.0")
    try:
        info = numa.pin_to_gpu_node(0, sysfs=str(tmp_path))
        assert info == dict(device=0, pci="0000:c1:00.0", numa_node=1, cpus=len(local), pinned=True, previous=allowed)
        assert os.sched_getaffinity(0) == set(local)
        again = numa.pin_to_gpu_node(0, sysfs=str(tmp_path))                          # already there: nothing to do
        assert again["pinned"] is False and again["cpus"] == len(local)
        # ... and the caller gets its affinity back (a detector created in bench / smoke / a Thread delegate must not leave it pinned)
        assert numa.restore_affinity(again) is False                                  # (nothing to give back for a pin that did nothing)
        assert numa.restore_affinity(info) is True and os.sched_getaffinity(0) == set(allowed)
        assert numa.restore_affinity(info) is False
    finally:
        os.sched_setaffinity(0, allowed)


def test_frame_memory_node_report():
    """`report_arena_nodes`: where the frame memory a detector binds lives (`move_pages(2)` as a query) against the GPU's node --
    one warning naming the cameras whose frames would cross the socket interconnect on every DMA."""
    import numpy as np

    class Log:
        def __init__(self):
            self.lines = []

        def warning(self, msg):
            self.lines.append(msg)

    a = np.ones(1 << 16, np.uint8)                                   # touched: resident on some node (or -1 without NUMA support)
    node = numa.memory_node(a.ctypes.data)
    assert node >= -1
    log = Log()
    same = numa.report_arena_nodes({"cam0": a.ctypes.data}, node, log)
    assert same["remote"] == [] and not log.lines and same["by_node"] == {node: ["cam0"]}
    if node >= 0:
        other = numa.report_arena_nodes({"cam0": a.ctypes.data, "cam1": a.ctypes.data + 4096}, node + 1, log)
        assert other["remote"] == ["cam0", "cam1"] and len(log.lines) == 1 and "cam0" in log.lines[0]
    unknown = numa.report_arena_nodes({"cam0": a.ctypes.data}, -1, log)               # GPU node unknown: nothing is called remote
    assert unknown["remote"] == []
