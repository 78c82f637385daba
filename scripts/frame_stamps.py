"""Where a frame's time goes on the GPU, from device-clock stamps (VO_FRAME_STAMPS, vo_frame.hip):
    python scripts/frame_stamps.py [frames in flight] [landmarks] [new per frame]
Stamps (100 MHz device clock): lane 0 start, 1 landmark track done, 2 tracked compaction, 3 pose chain, 4 pose tail;
ctx stream 5 start, 6 candidate track done, 7 pose event passed, 8 merge + compactions, 9 detect event passed, 10 selection
filter, 11 append + record; side stream 12 start, 13 pyramid of the frame after the next, 14 its detect half."""
import os, sys, subprocess
out = "gpurun_out/stamps.txt"
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    sys.path.insert(0, "scripts")
    import runpy
    try:
        runpy.run_path("scripts/frame_only.py", run_name="__main__")
    finally:
        import gc; gc.collect()
    sys.exit(0)
depth = sys.argv[1] if len(sys.argv) > 1 else "2"
nl = sys.argv[2] if len(sys.argv) > 2 else "300"
nn = sys.argv[3] if len(sys.argv) > 3 else "100"
os.makedirs("gpurun_out", exist_ok=True)
env = dict(os.environ, VO_FRAME_STAMPS=out)
r = subprocess.run([sys.executable, "scripts/frame_stamps.py", "--child", "24", nl, nn, "1", depth], env=env, capture_output=True, text=True)
print(r.stdout[-400:], r.stderr[-400:])
import numpy as np
if os.environ.get("VO_FRAME_STAMP_MASK"):
    rows = np.loadtxt(out)
    rows = rows[(rows[:, 0] >= 8) & (rows[:, 0] < 22)]
    mask = int(os.environ["VO_FRAME_STAMP_MASK"], 0)
    ids = [k for k in range(15) if (mask >> k) & 1]
    nm = ["L.start", "L.klt", "L.tracked", "L.pose", "L.tail", "M.start", "M.klt", "M.poseEv", "M.merge", "M.detEv", "M.filter", "M.append", "S.start", "S.pyr", "S.detect"]
    T = rows[:, 1:16] / 100.0
    print("period %.1f us" % np.median(np.diff(T[:, ids[0]])))
    for a in ids:
        for b in ids:
            if a != b:
                d = T[:, b] - T[:, a]
                dn = T[1:, b] - T[:-1, a]
                print("%-9s -> %-9s same frame %7.1f us   next frame %7.1f us" % (nm[a], nm[b], np.median(d), np.median(dn)))
    sys.exit(0)
rows = np.loadtxt(out)
rows = rows[(rows[:, 0] >= 8) & (rows[:, 0] < 22)]   # steady state of the last pass (frames are numbered per pass; the file keeps the last)
names = ["L.start", "L.klt", "L.tracked", "L.pose", "L.tail", "M.start", "M.klt", "M.poseEv", "M.merge", "M.detEv", "M.filter", "M.append",
         "S.start", "S.pyr", "S.detect"]
T = rows[:, 1:16] / 100.0               # us
per = np.diff(T[:, 0])
print("frame period (lane start to lane start): median %.1f us" % np.median(per))
ref = T[:, 0:1]
med = np.median(T - ref, axis=0)
for n, m in zip(names, med):
    print("%-10s %8.1f us after the frame's lane start" % (n, m))
# the loop: previous frame's M.merge (ev_state) -> this frame's L.start
gap = T[1:, 0] - T[:-1, 8]
print("M.merge(k) -> L.start(k+1): median %.1f us" % np.median(gap))
print("S.pyr(k) [pyramid of frame k+1... recorded in frame k-1] -> L.start(k+1): median %.1f us" % np.median(T[2:, 0] - T[:-2, 13]))
print("M.append(k) -> M.start(k+1): median %.1f us" % np.median(T[1:, 5] - T[:-1, 11]))

ev = []
for f in range(4, 7):
    for k, n in enumerate(names):
        ev.append((T[f, k] - T[4, 0], "f%d %s" % (int(rows[f, 0]), n), k))
for t, n, k in sorted(ev):
    print("%8.1f  %s%s" % (t, " " * 22 * (0 if k < 5 else (1 if k < 12 else 2)), n))
