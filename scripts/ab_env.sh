#!/bin/bash
# A/B of an environment switch on one box: scripts/ab_env.sh <rounds> VAR   (bench headline, k_lk launch time, stages)
set -e
rounds=$1; var=$2
for r in $(seq 1 "$rounds"); do
  for val in "" 1; do
    if [ -n "$val" ]; then export "$var=$val"; else unset "$var"; fi
    timeout -k 10 120 python bench.py --no-cpu --no-frame-loop --steps 100 > gpurun_out/ab_tmp.json
    python - "$var=${val:-unset}" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "k_lk %.4f" % d["roofline"]["launch_ms"],
      " ".join("%s %.4f" % (k.split("+")[0][:6], v) for k, v in d["stage_ms"].items()), flush=True)
PY
  done
done
