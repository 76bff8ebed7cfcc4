import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
        sc = d.get("single_context", {})
        print("%-40s %8.1f pairs/s  ms/step %.4f  single ctx %7.1f  sync %.4f ms  parity %.3g  ctxdiff %s launches %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], sc.get("value", 0), sc.get("synchronous_execute", {}).get("ms_per_pair", 0), d.get("parity_max_abs_err") or -1, d.get("contexts_max_abs_diff"), d["config"].get("launches_per_step")))
    except Exception as e:
        print(f, "ERR", e)
