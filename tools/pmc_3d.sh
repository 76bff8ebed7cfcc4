#!/bin/bash
export RT_DEV_KNOBS=1
# Hardware counters of every launch of a 3-D model (one counter group per rocprofv3 run, never together with a trace domain):
#   tools/pmc_3d.sh <outdir> [model] [--half2] [--batch=N]     -> <outdir>/summary.json + a table on stdout
# Per launch position of the LAST pass: kernel, duration (kernel trace), FETCH_SIZE (x2: gfx950 calibration for wide reads, see
# MI355X_MICROARCH.md), WRITE_SIZE, MFMA busy cycles, LDS bank conflicts.
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- python $ROOT/tools/iso_3d.py "$@" --mark > $OUT/g$i.log 2>&1      # (iso_3d.py: 3 passes)
done <<'GRPS'
FETCH_SIZE
WRITE_SIZE
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM
GRPS
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
names = None
for line in open(out + "/g1.log"):
    if line.startswith("launches per pass:"):
        names = eval(line.split(":", 1)[1])
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/g*/p_counter_collection.csv")):
    per = collections.defaultdict(list)          # dispatch id -> ...
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = int(r["Dispatch_Id"])
        e = disp.setdefault(d, {"kernel": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # the kernel trace of the same run gives the duration
    dur = {}
    kt = f.replace("p_counter_collection.csv", "p_kernel_trace.csv")
    try:
        for r in csv.DictReader(open(kt)):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    except Exception:
        pass
    # iso_3d.py --mark: every launch is followed by one rt::hash_words_kernel dispatch (the executor's launch trace), so the dispatches
    # between two of them are ONE launch (a Conv3DTranspose of the phase form is two: one per depth class)
    segs, cur = [], []
    for d in disp:
        if "rt::" not in disp[d]["kernel"] and not disp[d]["kernel"].startswith("_ZN2rt"):      # (rocprofv3 leaves _Float16 template instantiations mangled)
            continue
        if "hash_words_kernel" in disp[d]["kernel"]:
            segs.append(cur); cur = []
        else:
            cur.append(d)
    assert names and len(segs) % len(names) == 0, (len(segs), names)
    segs = segs[-len(names):]                   # the last pass
    for pos, ds in enumerate(segs):
        key = (pos, names[pos])
        e = rows.setdefault(key, {"kernel": " + ".join(sorted(set(disp[d]["kernel"].split("(")[0].split("<")[0].replace("void ", "") for d in ds)))})
        acc = {}
        for d in ds:
            for k, v in disp[d].items():
                if k != "kernel":
                    acc[k] = acc.get(k, 0.0) + v
        e.update(acc)
        if all(d in dur for d in ds) and ds:
            e.setdefault("us", []).append(sum(dur[d] for d in ds))
res = []
for (pos, name), e in rows.items():
    us = sum(e["us"]) / len(e["us"]) if e.get("us") else None
    fetch = 2 * e.get("FETCH_SIZE", 0) * 1024 if "FETCH_SIZE" in e else None      # KB -> bytes, x2 (gfx950: wide reads are tallied at half)
    write = e.get("WRITE_SIZE", 0) * 1024 if "WRITE_SIZE" in e else None
    res.append({"launch": pos, "name": name, "kernel": e["kernel"], "us_under_pmc": us, "fetch_bytes_x2": fetch, "write_bytes": write,
                "mfma_busy_cycles": e.get("SQ_VALU_MFMA_BUSY_CYCLES"), "busy_cu_cycles": e.get("SQ_BUSY_CU_CYCLES"),
                "lds_bank_conflict": e.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active": e.get("SQ_LDS_IDX_ACTIVE"),
                "wave_cycles": e.get("SQ_WAVE_CYCLES"), "wait_inst_any": e.get("SQ_WAIT_INST_ANY"), "wait_any": e.get("SQ_WAIT_ANY"), "insts_vmem": e.get("SQ_INSTS_VMEM")})
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print("launches of a pass:", names)
print("%-3s %-26s %-34s %9s %10s %10s %8s %12s" % ("#", "", "kernel", "us", "fetch MB", "write MB", "TB/s", "mfma busy %"))
for r in res:
    mb = lambda v: "%.1f" % (v / 1e6) if v is not None else "-"
    tbs = "%.2f" % (((r["fetch_bytes_x2"] or 0) + (r["write_bytes"] or 0)) / (r["us_under_pmc"] * 1e-6) / 1e12) if r["us_under_pmc"] else "-"
    busy = "%.0f" % (100.0 * r["mfma_busy_cycles"] / (4 * r["busy_cu_cycles"])) if r.get("mfma_busy_cycles") and r.get("busy_cu_cycles") else "-"
    print("%-3d %-26s %-34s %9s %10s %10s %8s %12s" % (r["launch"], r["name"][:26], r["kernel"][:34], "%.1f" % r["us_under_pmc"] if r["us_under_pmc"] else "-", mb(r["fetch_bytes_x2"]), mb(r["write_bytes"]), tbs, busy))
PY
