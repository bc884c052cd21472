#!/bin/bash
# usage: tools/pmc_sq.sh <out_json> <command...>
# Where the wave cycles of every kernel go, from one rocprofv3 PMC pass over the SQ block (8 slots; --kernel-trace only):
#   SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY  (quad-cycles,
#   MI355X_MICROARCH.md "rocprofv3 PMC slots"), plus the VALU / LDS instruction counts. Averages per dispatch.
out=$1; shift
export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"
rm -rf /tmp/hs_pmc_sq
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/hs_pmc_sq -o run -- "$@" > /tmp/hs_pmc_sq.log 2>&1 || { tail -5 /tmp/hs_pmc_sq.log; exit 1; }
python - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
out, cmd = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/hs_pmc_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    a = acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, cs in acc.items():
    v = {c: x[0] / x[1] for c, x in cs.items()}
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    res[k] = {"dispatches": next(iter(cs.values()))[1], "waves": round(v.get("SQ_WAVES", 0)), "wave_quad_cycles": round(wc),
              "parked_frac": round(v.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stall_frac": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
              "active_frac": round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "lds_issue_stall_frac": round(v.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
              "valu_insts_per_wave": round(v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1), 1),
              "lds_insts_per_wave": round(v.get("SQ_INSTS_LDS", 0) / max(v.get("SQ_WAVES", 1), 1), 1)}
json.dump({"source": f"rocprofv3 --pmc <8 SQ counters> --kernel-trace -- {cmd}, MI355X; averages per dispatch; cycle counters in quad-cycles",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["wave_quad_cycles"])[:14]:
    print(f'{k[:44]:44s} waves {v["waves"]:6d} parked {v["parked_frac"]:.2f} stall {v["issue_stall_frac"]:.2f} (lds {v["lds_issue_stall_frac"]:.2f}) active {v["active_frac"]:.2f} '
          f'valu/wave {v["valu_insts_per_wave"]:8.0f} lds/wave {v["lds_insts_per_wave"]:7.0f}')
PY
