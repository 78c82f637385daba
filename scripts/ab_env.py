"""A/B of the bench step under environment switches, alternating, on one box:  python scripts/ab_env.py VO_NO_LANE [reps]"""
import json, os, subprocess, sys
var = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for i in range(reps):
    for val in ("1", ""):
        env = dict(os.environ)
        if val:
            env[var] = val
        else:
            env.pop(var, None)
        r = subprocess.run([sys.executable, "bench.py", "--steps", "200", "--warmup", "20", "--no-cpu", "--no-frame-loop"], env=env,
                           capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{var}={val or 'unset':5s} {d['ms_per_step']:.4f} ms/step  stages {json.dumps({k: round(v, 4) for k, v in d['stage_ms'].items()})}", flush=True)
