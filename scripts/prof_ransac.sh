#!/bin/bash
# per-kernel durations of the pose chain alone (2000 hypotheses x 25k points):  bash scripts/prof_ransac.sh
R=$PWD; O=$R/gpurun_out/ransac_prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 scripts/ransac_only.py > $O/run.log 2> $O/trace.log
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
    print("%-14s g%-8s start %8.1f us  dur %7.1f us" % ((m.group(1) if m else "?")[:14], r.get("Grid_Size", "?"), (s - t0) / 1e3, (e - s) / 1e3))
PY
cat $O/run.log | grep -v amdgpu
