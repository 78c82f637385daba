#!/bin/bash
# A/B on one box: the bench step with and without the lane schedule (pose chain beside candidate tracking)
set -e
for i in 1 2; do
  VO_NO_LANE=1 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu --no-frame-loop | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no lane  ', round(d['ms_per_step'],4), 'ms/step')"
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu --no-frame-loop | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with lane', round(d['ms_per_step'],4), 'ms/step')"
done
