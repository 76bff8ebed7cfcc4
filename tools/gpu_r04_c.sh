#!/bin/bash
export RT_DEV_KNOBS=1 RT_WINO_IL8=1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04c; mkdir -p $O
RACE_SHOW=10 RT_VARIANT_DIR=tools/build/expA_p4 timeout 300 python tools/race_locate.py 500 6 exact 1 > $O/locate_src_p4.txt 2>&1
python - <<PY
import json
for line in open("$O/locate_src_p4.txt"):
    if line.startswith('{"pass"'):
        d = json.loads(line)
        o = d["output"]
        print(d["name"], "words", o["differing_words"], "k4", o.get("set_k4"), "i", o.get("set_i"), "ab", o.get("set_a"), o.get("set_b"))
        for s in o.get("sources", [])[:4]:
            print("   ", s["at"], "<-", [(h["delta_words"], h["pos"]) for h in s["same_bits_in_good_tensor_at"]][:3])
PY
