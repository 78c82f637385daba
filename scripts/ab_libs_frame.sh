#!/bin/bash
# A/B of library builds on one box, frame-loop leg included: scripts/ab_libs_frame.sh <rounds> lib1.so lib2.so ...
set -e
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    VO_HIP_LIB="$PWD/$lib" timeout -k 10 200 python bench.py --no-cpu --steps 100 > gpurun_out/ab_tmp.json
    python - "$lib" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json")); f = d["frame_loop"]
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "select %.4f" % d["stage_ms"]["select"], "frame %.4f" % f["ms_per_frame"],
      "frame_2k %.4f" % f["at_2k_landmarks"]["ms_per_frame"], flush=True)
PY
  done
done
