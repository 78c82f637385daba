#!/bin/bash
# A/B of library builds on one box: scripts/ab_libs.sh <rounds> lib1.so lib2.so ...  (bench headline + k_lk launch time)
set -e
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    VO_HIP_LIB="$PWD/$lib" timeout -k 10 120 python bench.py --no-cpu --no-frame-loop --steps 100 > gpurun_out/ab_tmp.json
    python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "k_lk %.4f" % d["roofline"]["launch_ms"],
      " ".join("%s %.4f" % (k.split("+")[0][:6], v) for k, v in d["stage_ms"].items()), flush=True)
PY
  done
done
