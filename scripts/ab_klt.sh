#!/bin/bash
# A/B of k_lk alone on one box: scripts/ab_klt.sh <rounds> <label>=<lib>[:scores] ...   (50k keypoints, REPS launches each, default 300: the clocks settle after ~50)
# e.g. scripts/ab_klt.sh 3 r2=ab/libvo_r2.so:scores new=visual_odometry_amd/libvo_hip.so:scores noscore=visual_odometry_amd/libvo_hip.so
set -e
rounds=$1; shift
for r in $(seq 1 "$rounds"); do
  for spec in "$@"; do
    label=${spec%%=*}; rest=${spec#*=}; lib=${rest%%:*}
    if [[ "$rest" == *:scores ]]; then sc=1; else sc=; fi
    echo -n "$label "
    VO_KLT_SCORES=$sc VO_HIP_LIB="$PWD/$lib" timeout -k 10 120 python scripts/klt_only.py 50000 ${REPS:-300} 2>&1 | grep -v amdgpu.ids
  done
done
