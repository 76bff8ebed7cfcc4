#!/usr/bin/env python3
"""profiles/rNN_traffic_3d.json from tools/pmc_3d.sh summaries (what bench.py's 3-D lines read as roofline.traffic):
    python tools/traffic_3d_json.py profiles/r04_traffic_3d.json "nvsmall half2=gpurun_out/r04q/pmc_c5/summary.json" "nvsmall=.../summary.json"
Per model key: launch name -> {kernel, us_under_pmc, fetch_bytes_x2, write_bytes} at batch 1 (FETCH_SIZE in KB x 1024 x 2: gfx950 tallies
wide reads at half, MI355X_MICROARCH.md; WRITE_SIZE x 1024; one counter per rocprofv3 run, no trace domain beside --kernel-trace)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out = {"sources_sha16": bench.kernel_sources_sha16(),
       "method": "tools/pmc_3d.sh: rocprofv3 --kernel-trace --pmc <one group per run> -- python tools/iso_3d.py <model> --mark; the last of 3 passes; "
                 "dispatches segmented per launch by the executor's launch-trace hash kernel; fetch_bytes_x2 = FETCH_SIZE KB x 1024 x 2, "
                 "write_bytes = WRITE_SIZE KB x 1024; batch 1 unless the key says otherwise (keys '... batch N': per launch of N samples)"}
for spec in sys.argv[2:]:
    key, path = spec.split("=", 1)
    rows = json.load(open(path))
    out[key] = {r["name"]: {k: r[k] for k in ("kernel", "us_under_pmc", "fetch_bytes_x2", "write_bytes", "mfma_busy_cycles", "busy_cu_cycles",
                                               "lds_bank_conflict", "lds_idx_active") if k in r} for r in rows if r.get("name")}
# a launch of several dispatches (the factored first Conv3D) is only as good as the dispatches rocprofv3 reported counters for
for key, layers in out.items():
    if isinstance(layers, dict):
        for name, e in layers.items():
            if isinstance(e, dict) and "fold_" in e.get("kernel", "") and "fold_combine" not in e["kernel"]:
                e["incomplete"] = "the combining pass (fold_combine_kernel) is missing from the counter output of this run: durations and bytes cover the other dispatches only"
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], {k: len(v) for k, v in out.items() if isinstance(v, dict)})
