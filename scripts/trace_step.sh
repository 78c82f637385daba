#!/bin/bash
# kernel timeline of one bench step (start / duration / gap to the previous kernel's end): bash scripts/trace_step.sh
set -e
R=$PWD; O=$R/gpurun_out/step_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --no-cpu --no-frame-loop --steps 30 > $O/run.log 2> $O/trace.log
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lk = [i for i, r in enumerate(rows) if "k_lk" in r["Kernel_Name"]]
i0, i1 = lk[-50], lk[-48]                      # two whole steps among the 60 untimed steps in front of the roofline leg (same schedule as the timed ones; one tracking launch a step since round 3)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    import re
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
    name = (m.group(1) if m else r["Kernel_Name"])[:28] + " g" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    gap = "" if prev_end is None else "%7.1f" % ((s - prev_end) / 1e3)
    print("%-36s start %8.1f us  dur %7.1f us  gap %s  q%s" % (name, (s - t0) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "?")))
    prev_end = max(prev_end or e, e)
PY
cat $O/timeline.txt
