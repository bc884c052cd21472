#!/bin/bash
# usage (GPU box): bash tools/replay_timeline.sh [seconds=6.0] [imu=1] [order=4] — start / end of every kernel of ONE LM iteration late in the replay
# (rocprofv3 kernel trace of hyperslam_amd/host/replay), gaps and overlaps between the two streams
export TMPDIR=/tmp
rm -rf /tmp/hs_rtl
rocprofv3 --kernel-trace --output-format csv -d /tmp/hs_rtl -o t -- hyperslam_amd/host/replay ${1:-6.0} ${2:-1} ${3:-4} > /tmp/hs_rtl.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/hs_rtl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_build_visual" in n]
i0 = idx[-3]  # third build from the end: iteration 3 of the last optimize()
t0 = int(rows[i0]["Start_Timestamp"])
while i0 > 0 and int(rows[i0 - 1]["Start_Timestamp"]) > t0 - 40000 and "k_pack_decision" not in names[i0 - 1] and "k_update_visual" not in names[i0 - 1]:
    i0 -= 1
prev_end = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:idx[-2]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:60]}')
    prev_end = max(prev_end, e)
PY
