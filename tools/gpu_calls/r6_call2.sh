#!/bin/bash
# round 6, call 2: the bench with telemetry + pre-warm (does amdsmi answer on this box?), then the whole GPU suite (timing)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/r6b_amdsmi_probe.txt
import amdsmi, json
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
for h in hs[:1]:
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: v for k, v in m.items() if any(s in k for s in ("clk", "power", "temp", "throttle", "activity"))})
    try: print("bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
    except Exception as e: print("bdf err", e)
    try: print("cap", amdsmi.amdsmi_get_power_cap_info(h))
    except Exception as e: print("cap err", e)
import torch
print("pci", torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else "n/a")
PY
timeout 1200 python bench.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/r6b_bench_telemetry.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r6b_gpu_suite.txt
