"""Where a rank's host threads run: next to its GPU.

One process per GPU.  A rank's host side -- the submitting thread, the retiring thread, the copy
pool and the pinned staging buffers they touch first -- belongs on the NUMA node its GPU hangs off: on the two-socket hosts of
this pool a 1080p frame goes up at 26 GB/s from the far socket and at 53.7 GB/s from the near one
(profiles/r03_numa_pinned_copy_probe.txt), and eight unbound ranks would each measure their placement luck.

`bind_rank` must run BEFORE the process makes its first HIP call (threads the runtime starts and pages the process touches
afterwards inherit the mask): the GPU's PCI address therefore comes from the KFD topology in sysfs, read the way ROCr reads it
(`enumerate_gpus_sysfs`; only if that fails from a short-lived CHILD process that initialises HIP in this one's stead), the node
and its CPUs from the PCI device's sysfs entry, and the mask is set with os.sched_setaffinity -- never a re-exec.
Everything degrades to "not bound" with the reason recorded: a bench line must say what happened on every rank, not guess.

The reference has no counterpart (one GPU, one process: nu_scaler_core/src/gpu/detector.rs:136-165 picks the adapter)."""
from __future__ import annotations

import os
import subprocess
import sys
from typing import Dict, List, Optional, Sequence

_PCI_QUERY = r"""
import ctypes, json, os, sys
out = []
try:
    hip = None
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        raise OSError("libamdhip64.so not found")
    n = ctypes.c_int(0)
    if hip.hipGetDeviceCount(ctypes.byref(n)) != 0:
        raise OSError("hipGetDeviceCount failed")
    for i in range(n.value):
        buf = ctypes.create_string_buffer(64)
        out.append(buf.value.decode().lower() if hip.hipDeviceGetPCIBusId(buf, 64, i) == 0 else None)  # sysfs: lower case
    print(json.dumps({"bdf": out}))
except Exception as e:
    print(json.dumps({"bdf": out, "error": "%s: %s" % (type(e).__name__, e)}))
"""


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def cgroup_cpu_quota() -> Optional[float]:
    """CPUs' worth of CFS quota of this process's cgroup (cpu.max / cpu.cfs_quota_us), None = unlimited or unreadable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else int(q) / int(p)
    except (OSError, ValueError, IndexError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read().split()[0])
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read().split()[0])
        return q / p if q > 0 else None
    except (OSError, ValueError, IndexError):
        return None


def usable_cpus() -> int:
    """CPUs this process may keep busy: its affinity mask capped by the cgroup's CPU quota (a GPU box of this pool gives a job
    16 CPUs' worth of the host's 256 hardware threads)."""
    n = len(os.sched_getaffinity(0))
    q = cgroup_cpu_quota()
    return n if q is None else max(1, min(n, int(q + 0.5)))


def _index_list(text: Optional[str]) -> Optional[List[int]]:
    """'0,2,3' -> [0, 2, 3]; None if unset; raises ValueError on anything else (UUIDs: the caller asks HIP instead)."""
    if text is None or text.strip() == "":
        return None
    return [int(p) for p in text.split(",") if p.strip() != ""]


def enumerate_gpus_sysfs(sysfs: str = "/sys", dev: str = "/dev", environ=None) -> Dict:
    """PCI addresses of the HIP devices in HIP's order WITHOUT touching HIP or starting a process: the KFD topology
    (/sys/class/kfd/kfd/topology/nodes/N/properties: GPU nodes have simd_count > 0, in the order ROCr enumerates them; domain
    and location_id give the PCI address), filtered as ROCr filters it (a node whose /dev/dri/renderD<minor> this process may
    not open does not exist for it: how a job sees one GPU of eight), then ROCR_VISIBLE_DEVICES and HIP_VISIBLE_DEVICES as index
    lists.  {"bdf": [...]} or {"bdf": [], "error": why} -- then the caller may fall back to query_gpu_pci()."""
    env = os.environ if environ is None else environ
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError as e:
        return {"bdf": [], "error": f"KFD topology unreadable: {e}"}
    gpus: List[str] = []
    for n in nodes:
        props: Dict[str, int] = {}
        try:
            with open(os.path.join(base, str(n), "properties")) as f:
                for line in f:
                    parts = line.split()
                    if len(parts) == 2 and parts[1].lstrip("-").isdigit():
                        props[parts[0]] = int(parts[1])
        except PermissionError:
            continue  # the device cgroup hides the properties of a GPU that is not this job's (seen on the 1-GPU boxes: EPERM)
        except OSError as e:
            return {"bdf": [], "error": f"KFD node {n}: {e}"}
        if props.get("simd_count", 0) <= 0:
            continue  # a CPU node
        minor = props.get("drm_render_minor", -1)
        if minor >= 0 and not os.access(os.path.join(dev, "dri", f"renderD{minor}"), os.R_OK | os.W_OK):
            continue  # not this job's GPU
        loc = props.get("location_id", 0)
        gpus.append(f"{props.get('domain', 0):04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    if not gpus:
        return {"bdf": [], "error": "no accessible GPU node in the KFD topology"}
    try:
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):  # HIP's list indexes what ROCr's list left
            idx = _index_list(env.get(var))
            if idx is not None:
                gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
        if env.get("CUDA_VISIBLE_DEVICES") and not env.get("HIP_VISIBLE_DEVICES"):
            idx = _index_list(env.get("CUDA_VISIBLE_DEVICES"))
            gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
    except ValueError:
        return {"bdf": [], "error": "a *_VISIBLE_DEVICES variable is not an index list"}
    return {"bdf": gpus, "how": "sysfs (KFD topology)"}


def query_gpu_pci(timeout: float = 60.0) -> Dict:
    """PCI addresses ('0000:d9:00.0') of the HIP devices in HIP's own order, honouring HIP_/ROCR_VISIBLE_DEVICES exactly as
    this process will -- asked of a child process, so that this one has still made no HIP call."""
    try:
        res = subprocess.run([sys.executable, "-c", _PCI_QUERY], capture_output=True, text=True, timeout=timeout)
        import json

        for line in reversed(res.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"bdf": [], "error": f"no answer from the query child (rc {res.returncode})"}
    except Exception as e:  # a missing interpreter, a timeout: placement is best effort
        return {"bdf": [], "error": f"{type(e).__name__}: {e}"}


def pci_numa(bdf: str, sysfs: str = "/sys") -> Dict:
    """NUMA node and local CPUs of a PCI device from sysfs; node -1 / None when the platform does not say."""
    base = os.path.join(sysfs, "bus", "pci", "devices", bdf)
    info: Dict = {"bdf": bdf, "numa_node": None, "local_cpus": []}
    try:
        with open(os.path.join(base, "numa_node")) as f:
            info["numa_node"] = int(f.read().strip())
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(base, "local_cpulist")) as f:
            info["local_cpus"] = parse_cpulist(f.read())
    except (OSError, ValueError):
        pass
    return info


def _core_of(cpu: int, sysfs: str) -> int:
    """The lowest sibling of `cpu`'s physical core (so that SMT siblings are dealt out together); the CPU itself if unknown."""
    try:
        with open(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{cpu}", "topology", "thread_siblings_list")) as f:
            return parse_cpulist(f.read())[0]
    except (OSError, ValueError, IndexError):
        return cpu


def plan_binding(devices: Sequence[Dict], local_rank: int, local_world: int, mask: Sequence[int],
                 quota: Optional[float] = None, sysfs: str = "/sys", device_of_rank: Optional[Sequence[int]] = None) -> Dict:
    """Pure planning step (no side effects; tests drive it with made-up topologies).

    devices[i] = pci_numa() of HIP device i; rank r uses device device_of_rank[r] (default: r).  The ranks whose GPUs share
    a node share that node's CPUs (those of them in `mask`), dealt out in whole physical cores, contiguous runs, in rank
    order; a rank whose GPU's node is unknown, or whose node has fewer cores in the mask than ranks, keeps the mask it has.
    cpus_per_rank = what the rank may keep busy: its slice, capped by its share of the cgroup quota."""
    mask = sorted(set(int(c) for c in mask))
    in_mask = set(mask)
    dor = list(device_of_rank) if device_of_rank is not None else list(range(max(local_world, local_rank + 1)))
    share = max(1, len(mask) // max(1, local_world))
    if quota is not None:
        share = max(1, min(share, int(quota / max(1, local_world) + 0.5)))
    plan = {"bound": False, "why_not": None, "gpu_bdf": None, "numa_node": None, "cpus": mask, "cpus_per_rank": share,
            "ranks_on_node": None}
    if not (0 <= local_rank < len(dor)) or not (0 <= dor[local_rank] < len(devices)):
        plan["why_not"] = f"no PCI address for the HIP device of local rank {local_rank} ({len(devices)} reported)"
        return plan
    dev = devices[dor[local_rank]]
    plan["gpu_bdf"], plan["numa_node"] = dev.get("bdf"), dev.get("numa_node")
    if dev.get("numa_node") is None or dev.get("numa_node") < 0 or not dev.get("local_cpus"):
        plan["why_not"] = "sysfs gives no NUMA node / local CPUs for this GPU"
        return plan
    local = [c for c in dev["local_cpus"] if c in in_mask]
    if not local:
        plan["why_not"] = "none of the GPU's local CPUs is in this process's affinity mask"
        return plan
    peers = [r for r in range(min(local_world, len(dor)))
             if 0 <= dor[r] < len(devices) and devices[dor[r]].get("numa_node") == dev["numa_node"]]
    if local_rank not in peers:
        peers = sorted(peers + [local_rank])
    plan["ranks_on_node"] = len(peers)
    cores: Dict[int, List[int]] = {}
    for c in local:
        cores.setdefault(_core_of(c, sysfs), []).append(c)
    core_ids = sorted(cores)
    if len(core_ids) < len(peers):
        plan["why_not"] = f"{len(core_ids)} local cores in the mask for {len(peers)} ranks on node {dev['numa_node']}"
        return plan
    k = peers.index(local_rank)
    per = len(core_ids) // len(peers)
    mine = core_ids[k * per:(k + 1) * per]
    cpus = sorted(c for core in mine for c in cores[core])
    plan.update(bound=True, cpus=cpus)
    cap = len(cpus)
    if quota is not None:
        cap = max(1, min(cap, int(quota / max(1, local_world) + 0.5)))
    plan["cpus_per_rank"] = cap
    return plan


def thread_budget(cpus_per_rank: int) -> Dict[str, int]:
    """Host threads of one rank inside its CPU share: the submitting and the retiring thread of the host path always exist;
    the copy pool gets what is left, up to the 6 workers it has by default (nus_copy.cpp); OpenMP teams (the checker's) take
    the whole share -- they never run beside the host path."""
    return {"copy_threads": max(0, min(6, int(cpus_per_rank) - 2)), "omp_threads": max(1, int(cpus_per_rank))}


def bind_rank(device_index: int, local_world: int, sysfs: str = "/sys", apply: bool = True, slot: Optional[int] = None) -> Dict:
    """Bind this process to the NUMA node of HIP device `device_index` (see the module text) and size its host threads: sets
    the affinity mask, NUS_COPY_THREADS (read once by the library's copy pool when it starts) and OMP_NUM_THREADS.  Returns the
    report that goes into the bench line.  Call before the first HIP call of the process.  `slot`: the rank's position among
    the node's ranks when several ranks share ONE device (a rehearsal on a 1-GPU box); default: rank r uses device r."""
    # sysfs first: no process is started and nothing initialises HIP (eight ranks asking eight short-lived children to open
    # every GPU of the node is sixteen processes on the GPUs for a moment); the child only where the topology cannot be read
    q = enumerate_gpus_sysfs(sysfs)
    if not q.get("bdf"):
        q2 = query_gpu_pci()
        q = q2 if q2.get("bdf") else {"bdf": [], "error": f"{q.get('error')}; {q2.get('error')}"}
    devices = [pci_numa(b, sysfs) if b else {"bdf": None, "numa_node": None, "local_cpus": []} for b in q.get("bdf", [])]
    mask = sorted(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    if slot is None:
        plan = plan_binding(devices, device_index, local_world, mask, quota, sysfs)
    else:
        plan = plan_binding(devices, slot, local_world, mask, quota, sysfs, device_of_rank=[device_index] * max(local_world, slot + 1))
    if q.get("error") and not plan["why_not"]:
        plan["why_not"] = q["error"]
    if apply and plan["bound"]:
        try:
            os.sched_setaffinity(0, plan["cpus"])
        except OSError as e:
            plan.update(bound=False, why_not=f"sched_setaffinity: {e}", cpus=mask)
    budget = thread_budget(plan["cpus_per_rank"])
    from_env = _apply_thread_budget(budget) if apply else {}
    return {"bound": plan["bound"], "why_not": plan["why_not"], "gpu_bdf": plan["gpu_bdf"], "numa_node": plan["numa_node"],
            "cpus": format_cpulist(plan["cpus"]), "n_cpus_in_mask": len(plan["cpus"]), "cpus_per_rank": plan["cpus_per_rank"],
            "ranks_on_node": plan["ranks_on_node"], "cgroup_cpu_quota": quota, "local_world": local_world, **budget, **from_env,
            "_replan": {"devices": devices, "mask": mask, "slot": slot, "device_index": device_index}}


from .launch import LAUNCHER_OMP_MARK  # set by launch_ranks when IT chose OMP_NUM_THREADS (torchrun's default is 1)


def _apply_thread_budget(budget: Dict[str, int]) -> Dict:
    """Write the rank's thread counts into the environment -- unless the operator has: a NUS_COPY_THREADS or OMP_NUM_THREADS that
    was set by hand wins (the launcher's own default for OMP_NUM_THREADS, marked with LAUNCHER_OMP_MARK, does not count as
    one), is reported as such, and `budget` is updated to what will actually be used."""
    rep = {}
    user_copy = os.environ.get("NUS_COPY_THREADS")
    if user_copy is not None and os.environ.get("NUS_COPY_THREADS_FROM_PLACEMENT") != "1":
        try:
            budget["copy_threads"] = max(0, int(user_copy))
            rep["copy_threads_from_env"] = True
        except ValueError:
            user_copy = None
    if not rep.get("copy_threads_from_env"):
        os.environ["NUS_COPY_THREADS"] = str(budget["copy_threads"])
        os.environ["NUS_COPY_THREADS_FROM_PLACEMENT"] = "1"  # a second bind_rank (a re-plan) may overwrite its own value
    user_omp = os.environ.get("OMP_NUM_THREADS")
    if user_omp is not None and os.environ.get(LAUNCHER_OMP_MARK) != "1":
        try:
            budget["omp_threads"] = max(1, int(user_omp.split(",")[0]))
            rep["omp_threads_from_env"] = True
        except ValueError:
            user_omp = None
    if not rep.get("omp_threads_from_env"):
        os.environ["OMP_NUM_THREADS"] = str(budget["omp_threads"])
        os.environ[LAUNCHER_OMP_MARK] = "1"
    return rep


def verify_after_init(place: Dict, hip_bdf: Optional[str], sysfs: str = "/sys", apply: bool = True) -> Dict:
    """Second half of bind_rank, AFTER the process has initialised HIP: the binding was planned from the KFD topology in sysfs,
    now HIP itself says where the rank's device sits (`hip_bdf`, '0000:d9:00.0').  Equal: verified.  Different (the sysfs order
    was not HIP's order on this box): the affinity is planned again from HIP's address and applied -- sched_setaffinity is as
    legal now as before; threads the runtime has started keep the old mask, the submitting / retiring threads and the copy pool
    (started by the first host call) get the new one -- and the report says so loudly (`rebound_after_init`).  Updates and
    returns `place`; the private planning fields are removed."""
    st = place.pop("_replan", None) or {}
    devices, mask0, slot, device_index = st.get("devices"), st.get("mask"), st.get("slot"), st.get("device_index")
    planned = place.get("gpu_bdf")
    if not hip_bdf:
        place["gpu_bdf_verified"] = None
        return place
    hip_bdf = hip_bdf.lower()
    place["gpu_bdf_by_hip"] = hip_bdf
    place["gpu_bdf_verified"] = bool(planned) and planned == hip_bdf
    if place["gpu_bdf_verified"] or devices is None or device_index is None or not place.get("bound"):
        return place
    # planned for another device than the one HIP gave this rank: plan again with HIP's word for this rank's entry
    devs = list(devices)
    while len(devs) <= device_index:
        devs.append({"bdf": None, "numa_node": None, "local_cpus": []})
    devs[device_index] = pci_numa(hip_bdf, sysfs)
    lw = int(place.get("local_world") or 1)
    if slot is None:
        plan = plan_binding(devs, device_index, lw, mask0, place.get("cgroup_cpu_quota"), sysfs)
    else:
        plan = plan_binding(devs, slot, lw, mask0, place.get("cgroup_cpu_quota"), sysfs, device_of_rank=[device_index] * max(lw, slot + 1))
    place["rebound_after_init"] = {"planned_bdf": planned, "planned_cpus": place.get("cpus"), "ok": False}
    if plan["bound"] and apply:
        try:
            os.sched_setaffinity(0, plan["cpus"])
            place["rebound_after_init"]["ok"] = True
            place.update(gpu_bdf=hip_bdf, numa_node=plan["numa_node"], cpus=format_cpulist(plan["cpus"]),
                         n_cpus_in_mask=len(plan["cpus"]), cpus_per_rank=plan["cpus_per_rank"], ranks_on_node=plan["ranks_on_node"])
        except OSError as e:
            place["rebound_after_init"]["error"] = f"sched_setaffinity: {e}"
    elif not plan["bound"]:
        place["rebound_after_init"]["error"] = plan["why_not"]
    return place
