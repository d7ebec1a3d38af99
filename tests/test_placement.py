"""Per-rank CPU placement (nu_scaler_amd/placement.py): the planning step on made-up topologies (no GPU, no real sysfs), the
cpulist helpers, and that bind_rank degrades to "not bound" with a reason on a box without a HIP device."""
import json
import os

import pytest


def _fake_sysfs(root, gpus, siblings):
    """gpus: {bdf: (node, cpulist)}; siblings: {cpu: 'a,b'}"""
    for bdf, (node, cpulist) in gpus.items():
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "numa_node"), "w").write(f"{node}\n")
        open(os.path.join(d, "local_cpulist"), "w").write(cpulist + "\n")
    for cpu, sib in siblings.items():
        d = os.path.join(root, "devices", "system", "cpu", f"cpu{cpu}", "topology")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "thread_siblings_list"), "w").write(sib + "\n")


def test_cpulist_round_trip(nsc):
    from nu_scaler_amd import placement as p

    assert p.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert p.format_cpulist([11, 10, 8, 3, 2, 1, 0]) == "0-3,8,10-11"
    assert p.parse_cpulist("") == [] and p.format_cpulist([]) == ""


def test_plan_eight_gpus_two_sockets(nsc, tmp_path):
    """The host shape of this pool (profiles/r03_numa_pinned_copy_probe.txt): 2 x 64 cores with SMT, GPUs 0-3 on node 0,
    4-7 on node 1.  Every rank gets a disjoint run of whole cores (both hardware threads) of ITS GPU's node."""
    from nu_scaler_amd import placement as p

    gpus = {f"0000:{0x10 + 8 * i:02x}:00.0": (0 if i < 4 else 1, "0-63,128-191" if i < 4 else "64-127,192-255") for i in range(8)}
    _fake_sysfs(str(tmp_path), gpus, {c: f"{c % 128},{c % 128 + 128}" for c in range(256)})
    devices = [p.pci_numa(b, str(tmp_path)) for b in sorted(gpus)]
    seen = set()
    for r in range(8):
        plan = p.plan_binding(devices, r, 8, range(256), quota=None, sysfs=str(tmp_path))
        assert plan["bound"] and plan["numa_node"] == (0 if r < 4 else 1) and plan["ranks_on_node"] == 4
        cpus = plan["cpus"]
        assert len(cpus) == 32 and plan["cpus_per_rank"] == 32
        assert all(((c % 128) + 128 if c < 128 else c - 128) in cpus for c in cpus), "SMT siblings stay together"
        assert set(cpus) <= set(devices[r]["local_cpus"]) and not (set(cpus) & seen)
        seen |= set(cpus)
    # a CPU quota for the whole job caps what a rank may keep busy, not where it runs
    plan = p.plan_binding(devices, 5, 8, range(256), quota=16.0, sysfs=str(tmp_path))
    assert plan["bound"] and len(plan["cpus"]) == 32 and plan["cpus_per_rank"] == 2
    assert p.thread_budget(2) == {"copy_threads": 0, "omp_threads": 2}
    assert p.thread_budget(16) == {"copy_threads": 6, "omp_threads": 16}
    assert p.thread_budget(5) == {"copy_threads": 3, "omp_threads": 5}


def test_plan_one_gpu_box_and_rehearsal(nsc, tmp_path):
    """One GPU of a shared host: the rank takes the whole local CPU list; two rehearsal ranks on that ONE device split it."""
    from nu_scaler_amd import placement as p

    _fake_sysfs(str(tmp_path), {"0000:d9:00.0": (1, "64-127,192-255")}, {})
    devices = [p.pci_numa("0000:d9:00.0", str(tmp_path))]
    plan = p.plan_binding(devices, 0, 1, range(256), quota=16.0, sysfs=str(tmp_path))
    assert plan["bound"] and p.format_cpulist(plan["cpus"]) == "64-127,192-255" and plan["cpus_per_rank"] == 16
    a = p.plan_binding(devices, 0, 2, range(256), None, str(tmp_path), device_of_rank=[0, 0])
    b = p.plan_binding(devices, 1, 2, range(256), None, str(tmp_path), device_of_rank=[0, 0])
    assert a["bound"] and b["bound"] and not (set(a["cpus"]) & set(b["cpus"])) and len(a["cpus"]) == len(b["cpus"]) == 64


def test_plan_degrades_with_a_reason(nsc, tmp_path):
    from nu_scaler_amd import placement as p

    _fake_sysfs(str(tmp_path), {"0000:05:00.0": (-1, "0-7")}, {})
    unknown = [p.pci_numa("0000:05:00.0", str(tmp_path))]
    plan = p.plan_binding(unknown, 0, 1, range(8))
    assert not plan["bound"] and "NUMA" in plan["why_not"] and plan["cpus"] == list(range(8)) and plan["cpus_per_rank"] == 8
    plan = p.plan_binding([], 0, 2, range(8), quota=4.0)
    assert not plan["bound"] and "no PCI address" in plan["why_not"] and plan["cpus_per_rank"] == 2
    far = [{"bdf": "0000:05:00.0", "numa_node": 1, "local_cpus": [64, 65]}]
    plan = p.plan_binding(far, 0, 1, range(8))
    assert not plan["bound"] and "affinity mask" in plan["why_not"]


def test_bind_rank_without_a_gpu_reports_and_changes_nothing(nsc, monkeypatch):
    """No HIP device in this container: the query child says so, the mask stays, the thread budget is still set."""
    import torch

    from nu_scaler_amd import placement as p

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    monkeypatch.delenv("NUS_COPY_THREADS", raising=False)  # (both are restored to what they were when the test ends)
    monkeypatch.setenv("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS", "0"))
    if os.environ["OMP_NUM_THREADS"] == "0":
        monkeypatch.delenv("OMP_NUM_THREADS")
    before = os.sched_getaffinity(0)
    rep = p.bind_rank(0, 2)
    assert os.sched_getaffinity(0) == before
    assert rep["bound"] is False and rep["why_not"] and rep["local_world"] == 2
    assert rep["cpus_per_rank"] == max(1, min(len(before) // 2, int((p.cgroup_cpu_quota() or 1e9) / 2 + 0.5)))
    assert os.environ["NUS_COPY_THREADS"] == str(rep["copy_threads"])
    assert os.environ["OMP_NUM_THREADS"] == str(rep["omp_threads"])
    json.dumps(rep)  # the report goes into a bench line as it is


def test_bind_rank_respects_thread_counts_the_operator_set(nsc, monkeypatch):
    """NUS_COPY_THREADS / OMP_NUM_THREADS set by hand win over the rank's budget and are reported as such; the
    launcher's own OMP_NUM_THREADS default (marked) and the values an earlier bind_rank wrote do not count as the operator's."""
    from nu_scaler_amd import placement as p

    for k in ("NUS_COPY_THREADS", "OMP_NUM_THREADS", p.LAUNCHER_OMP_MARK, "NUS_COPY_THREADS_FROM_PLACEMENT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("NUS_COPY_THREADS", "1")
    monkeypatch.setenv("OMP_NUM_THREADS", "3")
    rep = p.bind_rank(0, 1, sysfs="/nonexistent")
    assert rep["copy_threads"] == 1 and rep["omp_threads"] == 3 and rep["copy_threads_from_env"] and rep["omp_threads_from_env"]
    assert os.environ["NUS_COPY_THREADS"] == "1" and os.environ["OMP_NUM_THREADS"] == "3"
    # the launcher's default: re-sized
    monkeypatch.delenv("NUS_COPY_THREADS")
    monkeypatch.setenv("OMP_NUM_THREADS", "4")
    monkeypatch.setenv(p.LAUNCHER_OMP_MARK, "1")
    rep = p.bind_rank(0, 1, sysfs="/nonexistent")
    assert "omp_threads_from_env" not in rep and os.environ["OMP_NUM_THREADS"] == str(rep["omp_threads"]) == str(rep["cpus_per_rank"])
    first = os.environ["NUS_COPY_THREADS"]
    rep2 = p.bind_rank(0, 2, sysfs="/nonexistent")  # a second plan may overwrite what the first one wrote
    assert "copy_threads_from_env" not in rep2 and os.environ["NUS_COPY_THREADS"] == str(rep2["copy_threads"])
    assert int(first) >= int(os.environ["NUS_COPY_THREADS"])
    # apply=False touches nothing
    monkeypatch.setenv("NUS_COPY_THREADS", "5")
    monkeypatch.delenv("NUS_COPY_THREADS_FROM_PLACEMENT")
    p.bind_rank(0, 1, sysfs="/nonexistent", apply=False)
    assert os.environ["NUS_COPY_THREADS"] == "5"


def test_verify_after_init_replans_from_the_address_hip_reports(nsc, tmp_path, monkeypatch):
    """The sysfs order was not HIP's on this (made-up) box: the rank planned for the GPU on node 0, HIP says its device is the one
    on node 1 -- the binding is planned again from HIP's address, applied, and the report says so."""
    from nu_scaler_amd import placement as p

    mask = sorted(os.sched_getaffinity(0))
    if len(mask) < 4:
        pytest.skip("needs four CPUs")
    half = len(mask) // 2
    sysfs = tmp_path / "sys"
    for bdf, node, cpus in (("0000:05:00.0", 0, mask[:half]), ("0000:85:00.0", 1, mask[half:])):
        d = sysfs / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        (d / "local_cpulist").write_text(p.format_cpulist(cpus) + "\n")
    devices = [p.pci_numa("0000:05:00.0", str(sysfs)), p.pci_numa("0000:85:00.0", str(sysfs))]
    applied = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: applied.append(sorted(cpus)))
    plan = p.plan_binding(devices, 0, 2, mask, None, str(sysfs))
    place = {"bound": True, "gpu_bdf": "0000:05:00.0", "numa_node": 0, "cpus": p.format_cpulist(plan["cpus"]), "local_world": 2,
             "cgroup_cpu_quota": None, "_replan": {"devices": devices, "mask": mask, "slot": None, "device_index": 0}}
    same = p.verify_after_init(dict(place, _replan=dict(place["_replan"])), "0000:05:00.0", str(sysfs))
    assert same["gpu_bdf_verified"] is True and "rebound_after_init" not in same and "_replan" not in same and not applied
    moved = p.verify_after_init(place, "0000:85:00.0", str(sysfs))
    assert moved["gpu_bdf_verified"] is False and moved["rebound_after_init"]["ok"] and moved["numa_node"] == 1
    assert applied and set(applied[-1]) <= set(mask[half:]) and moved["gpu_bdf"] == "0000:85:00.0"
    assert moved["rebound_after_init"]["planned_bdf"] == "0000:05:00.0"
    json.dumps(moved)


def test_enumerate_gpus_from_the_kfd_topology(nsc, tmp_path):
    """HIP's device order without HIP: GPU nodes of the KFD topology whose render node the process may open, then the
    *_VISIBLE_DEVICES index lists (ROCr's first, HIP's on what is left)."""
    from nu_scaler_amd import placement as p

    sysfs, dev = tmp_path / "sys", tmp_path / "dev"
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (dev / "dri").mkdir(parents=True)
    # two CPU nodes, then four GPUs; the job may open the render nodes of GPUs 1, 2 and 3 only
    specs = [(0, 0, -1, 0), (0, 0, -1, 0), (1216, 0x0500, 128, 0), (1216, 0xa400, 129, 0), (1216, 0xd900, 130, 0), (1216, 0x1508, 131, 1)]
    for n, (simd, loc, minor, domain) in enumerate(specs):
        d = nodes / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\nlocation_id {loc}\ndomain {domain}\n"
                                      f"drm_render_minor {minor}\nname_is_text x\n")
        if minor in (129, 130, 131):
            (dev / "dri" / f"renderD{minor}").write_text("")
    got = p.enumerate_gpus_sysfs(str(sysfs), str(dev), environ={})
    assert got["bdf"] == ["0000:a4:00.0", "0000:d9:00.0", "0001:15:01.0"]
    assert p.enumerate_gpus_sysfs(str(sysfs), str(dev), environ={"ROCR_VISIBLE_DEVICES": "2,0"})["bdf"] == ["0001:15:01.0", "0000:a4:00.0"]
    assert p.enumerate_gpus_sysfs(str(sysfs), str(dev), environ={"ROCR_VISIBLE_DEVICES": "2,0", "HIP_VISIBLE_DEVICES": "1"})["bdf"] == ["0000:a4:00.0"]
    assert "index list" in p.enumerate_gpus_sysfs(str(sysfs), str(dev), environ={"HIP_VISIBLE_DEVICES": "GPU-abc"})["error"]
    assert p.enumerate_gpus_sysfs(str(tmp_path / "nothing"), str(dev), environ={})["bdf"] == []


def test_bind_rank_for_all_eight_slots_of_the_node_the_driver_will_use(nsc, tmp_path, monkeypatch):
    """the shape of the first 8-GPU contact -- eight GPUs in the KFD topology behind two CPU nodes, 2 x 64
    cores with SMT (256 hardware threads in the mask), a cgroup quota of 16 CPUs for the whole job -- driven through bind_rank
    itself for every slot: each rank is planned onto 32 CPUs of ITS GPU's node (whole cores, disjoint from every other rank),
    may keep 2 of them busy, and so runs a ZERO-worker copy pool (the submitting and the retiring thread are its two)."""
    from nu_scaler_amd import placement as p

    sysfs = tmp_path / "sys"
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    bdfs = []
    for n in range(10):  # two CPU nodes, then the eight GPUs in ROCr's order
        d = nodes / str(n)
        d.mkdir(parents=True)
        if n < 2:
            (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\ndrm_render_minor -1\n")
            continue
        g = n - 2
        bus = 0x05 + 0x20 * g
        bdfs.append(f"0000:{bus:02x}:00.0")
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\ndrm_render_minor -1\n")
    _fake_sysfs(str(sysfs), {b: (0 if i < 4 else 1, "0-63,128-191" if i < 4 else "64-127,192-255") for i, b in enumerate(bdfs)},
                {c: f"{c % 128},{c % 128 + 128}" for c in range(256)})
    monkeypatch.setattr(p, "cgroup_cpu_quota", lambda: 16.0)
    monkeypatch.setattr(p.os, "sched_getaffinity", lambda pid: set(range(256)))
    for k in ("NUS_COPY_THREADS", "OMP_NUM_THREADS", p.LAUNCHER_OMP_MARK, "NUS_COPY_THREADS_FROM_PLACEMENT"):
        monkeypatch.setenv(k, "registered-for-restore")  # (so that what _apply_thread_budget writes below is undone afterwards)
        monkeypatch.delenv(k)
    seen = set()
    for r in range(8):
        rep = p.bind_rank(r, 8, sysfs=str(sysfs), apply=False)
        assert rep["bound"] and rep["why_not"] is None and rep["gpu_bdf"] == bdfs[r], rep
        assert rep["numa_node"] == (0 if r < 4 else 1) and rep["ranks_on_node"] == 4 and rep["local_world"] == 8
        cpus = p.parse_cpulist(rep["cpus"])
        assert len(cpus) == 32 == rep["n_cpus_in_mask"] and not (set(cpus) & seen)
        assert all(((c + 128) if c < 128 else (c - 128)) in cpus for c in cpus), "SMT siblings stay together"
        seen |= set(cpus)
        assert rep["cgroup_cpu_quota"] == 16.0 and rep["cpus_per_rank"] == 2
        assert rep["copy_threads"] == 0 and rep["omp_threads"] == 2
        json.dumps({k: v for k, v in rep.items() if k != "_replan"})
        # what apply=True writes for the library's copy pool of this rank (read once, when the pool starts)
        budget = p.thread_budget(rep["cpus_per_rank"])
        assert p._apply_thread_budget(budget) == {} and os.environ["NUS_COPY_THREADS"] == "0" and os.environ["OMP_NUM_THREADS"] == "2"
    assert seen == set(range(256))
