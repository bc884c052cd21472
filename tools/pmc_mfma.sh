#!/bin/bash
# usage: tools/pmc_mfma.sh <out_json> <command...>
# MFMA evidence for the factorisation kernels (BASELINE.json north_star: "MFMA utilisation against gfx950 peak"): one rocprofv3 PMC pass
# (--kernel-trace only) with the matrix-core counters of the SQ block next to the VALU instruction count and the wave cycles:
#   SQ_INSTS_VALU_MFMA_F64     fp64 MFMA instructions issued        SQ_INSTS_MFMA          MFMA instructions of any type
#   SQ_VALU_MFMA_BUSY_CYCLES   cycles the matrix pipe was busy       SQ_INSTS_VALU          VALU instructions (MFMA included)
#   SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES / SQ_WAVES                    denominators
# Run it once with the default library (k_band_factor_la: VALU register tiles) and once with HS_DEBUG_FLAGS=131072 (k_band_factor_mfma:
# v_mfma_f64_16x16x4_f64 on the trailing window); tools/round_measurements.sh does both.
out=$1; shift
export TMPDIR=/tmp
C="SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
rm -rf /tmp/hs_pmc_mfma
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/hs_pmc_mfma -o run -- "$@" > /tmp/hs_pmc_mfma.log 2>&1 || { tail -5 /tmp/hs_pmc_mfma.log; exit 1; }
python - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections, os
out, cmd = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/hs_pmc_mfma/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    a = acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, cs in acc.items():
    v = {c: x[0] / x[1] for c, x in cs.items()}
    busy = max(v.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0)
    res[k] = {"dispatches": next(iter(cs.values()))[1], "waves": round(v.get("SQ_WAVES", 0)), "valu_insts": round(v.get("SQ_INSTS_VALU", 0)),
              "mfma_insts": round(v.get("SQ_INSTS_MFMA", 0)), "mfma_f64_insts": round(v.get("SQ_INSTS_VALU_MFMA_F64", 0)),
              "mfma_busy_cycles": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)), "busy_cu_cycles": round(busy),
              "mfma_busy_over_cu_busy": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / busy, 4)}
json.dump({"source": f"rocprofv3 --pmc {os.environ.get('HS_PMC_NOTE', '<7 SQ counters>')} --kernel-trace -- {cmd} (HS_DEBUG_FLAGS={os.environ.get('HS_DEBUG_FLAGS', '0')}), MI355X; averages per dispatch",
           "kernels": res}, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["busy_cu_cycles"])[:12]:
    print(f'{k[:46]:46s} waves {v["waves"]:6d} valu {v["valu_insts"]:9d} mfma {v["mfma_insts"]:7d} (f64 {v["mfma_f64_insts"]:7d}) mfma busy {v["mfma_busy_cycles"]:8d} / cu busy {v["busy_cu_cycles"]:9d} = {v["mfma_busy_over_cu_busy"]:.4f}')
PY
