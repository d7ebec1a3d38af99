#!/bin/bash
# where the GPU sits among the host's NUMA nodes, and what the pinned-copy ceilings / the host path look like from CPUs near and far
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"; done
lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)" 
python3 - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("torch device 0 pci:", getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None))
PY
cpus=$(cat /sys/class/drm/card*/device/local_cpulist 2>/dev/null | head -1)
echo "== pcie probe, default affinity"; python3 tools/pcie_probe.py 2>&1 | grep -v amdgpu
if [ -n "$cpus" ]; then echo "== pcie probe, taskset -c $cpus (GPU-local)"; taskset -c $cpus python3 tools/pcie_probe.py 2>&1 | grep -v amdgpu; fi
for set in 0-15 64-79 128-143 192-207; do echo "== pcie probe, taskset -c $set"; taskset -c $set python3 tools/pcie_probe.py 2>&1 | grep -E "4k_out|memcpy"; done
