#!/bin/bash
# bench.py over keypoints per frame: bash scripts/keypoint_sweep.sh > gpurun_out/keypoint_sweep.txt
set -e
echo "keypoints/frame  ms/step  frames/s  Mkeypoints/s  roofline.frac  stage_ms (each stage alone)"
for n in 2000 5000 10000 20000 30000 40000 50000; do
  timeout -k 10 120 python bench.py --keypoints $n --no-cpu --no-frame-loop --steps 100 > gpurun_out/sweep_tmp.json 2> /dev/null
  python - $n <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep_tmp.json"))
print("%7d  %.4f  %7.1f  %6.2f  %.3f  %s" % (int(sys.argv[1]), d["ms_per_step"], d["frames_per_s"], d["value"] / 1e6, d["roofline"]["frac"],
      " ".join("%s=%.3f" % (k.split("+")[0][:8], v) for k, v in d["stage_ms"].items())), flush=True)
PY
done
