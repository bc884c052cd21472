#!/bin/bash
# usage: tools/iteration_timeline.sh [config]   — start/end of every kernel of one LM iteration (rocprofv3 kernel trace), gaps and overlaps
export TMPDIR=/tmp
rm -rf /tmp/hs_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/hs_tl -o t -- python bench.py --config ${1:-1} --steps 3 --warmup 2 --no-cpu-baseline > /tmp/hs_tl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/hs_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last complete iteration but one: from a k_build_visual (k_linearize_visual on the record path) to the next
idx = [i for i, n in enumerate(names) if "k_build_visual" in n] or [i for i, n in enumerate(names) if "k_linearize_visual" in n]
# (side-stream kernels of the iteration may start before the build: include what runs up to 40 us before it)
i0 = idx[-3]
t0 = int(rows[i0]["Start_Timestamp"])
while i0 > 0 and int(rows[i0 - 1]["Start_Timestamp"]) > t0 - 40000 and "k_commit" not in names[i0 - 1] and "k_pack_decision" not in names[i0 - 1]:
    i0 -= 1
prev_end = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:idx[-2]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:60]}')
    prev_end = max(prev_end, e)
PY
