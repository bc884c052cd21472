#!/bin/bash
# usage: tools/pmc_traffic.sh <out_json> <command...>
# HBM traffic per kernel from rocprofv3 PMC counters, one counter per pass (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, no
# other trace domains), aggregated per kernel name: average KiB per dispatch.
out=$1; shift
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/hs_pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/hs_pmc_$c -o run -- "$@" > /tmp/hs_pmc_$c.log 2>&1 || { tail -5 /tmp/hs_pmc_$c.log; exit 1; }
done
python - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
out, cmd = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/hs_pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        a = acc[r["Kernel_Name"].split("(")[0].replace("void ", "")]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in acc.items():
        res[k][c + "_KiB"] = round(v / n, 3); res[k]["dispatches"] = n
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace (separate passes) -- {cmd}, MI355X",
           "unit": "KiB per dispatch (TCC_EA0 request counters x 64 B); FETCH_SIZE may under-count wide coalesced reads by 2x on gfx950 "
                   "(MI355X_MICROARCH.md HBM section) - uncalibrated",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE_KiB", 0) + kv[1].get("WRITE_SIZE_KiB", 0)))[:14]:
    print(f'{k[:60]:60s} fetch {v.get("FETCH_SIZE_KiB", 0):10.1f} KiB  write {v.get("WRITE_SIZE_KiB", 0):10.1f} KiB  x{v["dispatches"]}')
PY
