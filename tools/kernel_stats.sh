#!/bin/bash
# usage: tools/kernel_stats.sh <out_csv> <command...>   — rocprofv3 kernel-trace stats of a command, top kernels printed
out=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/hs_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs_prof -o p -- "$@" > /tmp/hs_prof_cmd.log 2>&1
f=$(find /tmp/hs_prof -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then tail -20 /tmp/hs_prof_cmd.log; exit 1; fi
mkdir -p "$(dirname "$out")"; cp "$f" "$out"
python - "$out" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):5.1f} %')
PY
