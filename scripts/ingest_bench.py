"""Ingest rate of the decode pool (vo_ingest_*) on KITTI-shaped PNG frames, alone and feeding the
device-resident frame loop.  Frames are synthetic (texture warped along a
track, synth.plane_sequence), written as PNG with Paeth-filtered rows (the filter photographic PNGs mostly
use, and the costliest to undo).

    python scripts/ingest_bench.py [n_files] [threads ...]
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import numpy as np

from visual_odometry_amd import ingest, synth
from visual_odometry_amd.frame import FrameLoop

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16]
K = synth.KITTI_K
frames, poses, lift = synth.plane_sequence(16, K, nv=376, nu=1241)
tmp = tempfile.mkdtemp(prefix="vo_ingest_")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ingest as oi                                    # fixture writer only (png_encode)
blobs = [oi.png_encode(f, [4]) for f in frames]
paths = []
for i in range(n_files):
    p = os.path.join(tmp, "%06d.png" % i)
    with open(p, "wb") as f:
        f.write(blobs[(i % 30) if (i % 30) < 16 else 30 - (i % 30)])   # 0..15..1: a there-and-back track
    paths.append(p)
print(f"{n_files} files of {len(blobs[0]) / 1e3:.0f} kB, 1241 x 376 gray")

for t in threads:
    fs = ingest.FrameStream(paths, threads=t)
    fs.ctx.sync()
    t0 = time.perf_counter()
    n = 0
    for img in fs:
        n += 1
    fs.ctx.sync()
    dt = time.perf_counter() - t0
    st = fs.stats()
    print(f"pool alone   threads {st.threads:2d} depth {st.depth:2d}: {n / dt:7.0f} frames/s "
          f"({1e3 * st.decode_seconds / n:.2f} ms of worker time per frame)")
    fs.close()

# Malaga-shaped colour JPEGs (800 x 600, 4:2:0): Huffman decoding on the host, the rest on the device
try:
    import io
    from PIL import Image                                          # fixture writer only
    rgb = np.stack([synth.texture(600, 800, 31 + c, margin=0)[:600, :800] for c in range(3)], -1).astype(np.uint8)
    jp = []
    for i in range(8):
        b = io.BytesIO()
        Image.fromarray(np.roll(rgb, 3 * i, 1)).save(b, "JPEG", quality=90, subsampling=2)
        jp.append(b.getvalue())
    jpaths = []
    for i in range(n_files):
        p = os.path.join(tmp, "m%06d.jpg" % i)
        with open(p, "wb") as f:
            f.write(jp[i % 8])
        jpaths.append(p)
    print(f"{n_files} JPEG files of {len(jp[0]) / 1e3:.0f} kB, 800 x 600 colour 4:2:0")
    t0 = time.perf_counter()
    for b in jp * 4:
        ingest.imdecode(b)
    print(f"host-only decode (imread): {1e3 * (time.perf_counter() - t0) / 32:.2f} ms per file on one thread")
    for t in threads:
        fs = ingest.FrameStream(jpaths, threads=t)
        fs.ctx.sync()
        t0 = time.perf_counter()
        n = sum(1 for _ in fs)
        fs.ctx.sync()
        dt = time.perf_counter() - t0
        st = fs.stats()
        print(f"jpeg pool    threads {st.threads:2d} depth {st.depth:2d}: {n / dt:7.0f} frames/s "
              f"({1e3 * st.decode_seconds / n:.2f} ms of worker time per frame)")
        fs.close()
    for p in jpaths:
        os.remove(p)
except ImportError:
    print("Pillow missing: no JPEG leg")

# feeding the frame loop: state from the oracle-free product path (selection on frame 0 + the plane's lift)
from visual_odometry_amd import modules
for t in threads:
    loop = FrameLoop(K, frames[0].shape, maxNumLandmarks=300)
    kp = modules.selectCandidateKeypoints(frames[0], np.zeros((0, 2)), MinDistance=7, CandidatesToKeep=300) - 1.0
    st = modules._new_state()
    st.keypoints, st.landmarks, st.reproError = kp.astype(np.float64), lift(kp[:, 0], kp[:, 1]), np.zeros(len(kp))
    st.candidate_first_poses = np.zeros((0, 4, 3))
    loop.set_state(st, frames[0])
    fs = ingest.FrameStream(paths[1:], threads=t, hold=2)      # two frames in hand: the current one and the one announced
    t0 = time.perf_counter()
    n = 0
    ok = 0
    img = next(fs, None)
    while img is not None:
        nxt = next(fs, None)
        pose, status, info = loop.processFrame(img, nxt)
        ok += status == 0
        n += 1
        img = nxt
    dt = time.perf_counter() - t0
    print(f"pool + processFrame threads {t:2d}: {n / dt:7.0f} frames/s ({ok}/{n} poses ok, {info['n_landmarks']} landmarks at the end)")
    fs.close(); loop.close()
for p in paths:
    os.remove(p)
os.rmdir(tmp)
