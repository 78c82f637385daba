"""Frame ingest on the device, through the C ABI: vo_rgb2gray against the oracle's N11
definition (bit-exact), and the decode pool (vo_ingest_*) against the oracle's imread + rgb2gray of the
same files: order, contents, errors in the middle of a sequence, early close, and the hand-over of its
device images to the frame loop."""
import numpy as np
import pytest

from oracle import ingest as oi
from visual_odometry_amd import _lib, ingest, synth
from visual_odometry_amd.frame import FrameLoop

pytestmark = pytest.mark.gpu


def textured(rng, h, w, c=None):
    a = rng.integers(0, 256, (h, w) if c is None else (h, w, c)).astype(np.float64)
    for ax in (0, 1):
        a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax) + np.roll(a, 2, ax)) / 4
    return a.astype(np.uint8)


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (37, 53), (376, 1241), (480, 640)])
def test_rgb2gray_bit_exact(ctx, shape):
    rng = np.random.default_rng(shape[0])
    rgb = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    exp = oi.rgb2gray(rgb)
    assert np.array_equal(ingest.rgb2gray(rgb), exp)                       # interleaved rows (numpy)
    planes = np.ascontiguousarray(rgb.transpose(2, 1, 0))                   # MATLAB's H x W x 3 column-major memory
    out = np.zeros(shape[::-1], np.uint8)
    ctx.call("vo_rgb2gray", _lib.ptr(planes), shape[0], shape[1], 1, 0, _lib.ptr(out))
    assert np.array_equal(out.T, exp)


def test_rgb2gray_every_colour_of_a_slice_and_device_tensors(ctx):
    import torch
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 5), indexing="ij")
    rgb = np.stack([r, g, b], -1).astype(np.uint8).reshape(256, -1, 3)
    exp = oi.rgb2gray(rgb)
    assert np.array_equal(ingest.rgb2gray(rgb), exp)
    d = torch.from_numpy(rgb).cuda()
    assert np.array_equal(ingest.rgb2gray(d).cpu().numpy(), exp)
    odd = d.reshape(-1)[3:3 + 3 * 1001].reshape(7, 143, 3).contiguous()      # byte path: odd size
    assert np.array_equal(ingest.rgb2gray(odd).cpu().numpy(), oi.rgb2gray(odd.cpu().numpy()))
    with pytest.raises(ValueError):
        ingest.rgb2gray(np.zeros((4, 4), np.uint8))
    with pytest.raises(_lib.VoError):
        ctx.call("vo_rgb2gray", None, 4, 4, 0, 0, None)
    ctx.call("vo_rgb2gray", None, 0, 4, 0, 0, None)                          # empty image: nothing to do


def write_sequence(tmp_path, rng, n, shape=(48, 64)):
    paths, expect = [], []
    for i in range(n):
        colour = i % 3 == 1
        img = textured(rng, shape[0], shape[1], 3 if colour else None)
        p = tmp_path / ("f%04d.%s" % (i, "ppm" if i % 5 == 4 else "png"))
        if i % 5 == 4:
            img = textured(rng, shape[0], shape[1], 3)
            p.write_bytes(b"P6 %d %d 255\n" % (shape[1], shape[0]) + img.tobytes())
        else:
            p.write_bytes(oi.png_encode(img, None if i % 2 else [4]))
        paths.append(str(p))
        expect.append(oi.to_gray(oi.imread(p.read_bytes())))
    return paths, expect


@pytest.mark.parametrize("threads,depth", [(1, 2), (3, 4), (8, 0)])
def test_pool_delivers_every_frame_in_order(ctx, tmp_path, threads, depth):
    paths, expect = write_sequence(tmp_path, np.random.default_rng(11), 23)
    fs = ingest.FrameStream(paths, threads=threads, depth=depth)
    seen = 0
    for k, img in enumerate(fs):
        assert img.index == k and img.shape == (48, 64) and img.channels == 1
        assert np.array_equal(fs.read(), expect[k]), k
        seen += 1
    assert seen == 23
    st = fs.stats()
    assert st.threads == min(threads, 23) and st.depth >= 2 and st.decode_seconds > 0
    with pytest.raises(StopIteration):
        next(fs)
    fs.close()
    fs.close()


def test_pool_keep_rgb_mixed_sizes_and_slot_reuse(ctx, tmp_path):
    rng = np.random.default_rng(12)
    paths, expect = [], []
    for i, shape in enumerate([(10, 12), (40, 30), (5, 7), (64, 64), (9, 9), (33, 17), (40, 30)] * 3):
        img = textured(rng, shape[0], shape[1], 3 if i % 2 else None)
        p = tmp_path / ("m%03d.png" % i)
        p.write_bytes(oi.png_encode(img))
        paths.append(str(p)); expect.append(img)
    fs = ingest.FrameStream(paths, threads=2, depth=2, keep_rgb=True)      # two slots: each is reused ten times
    for k, img in enumerate(fs):
        assert img.shape == expect[k].shape
        assert np.array_equal(fs.read(), expect[k]), k
    fs.close()


def test_pool_reports_a_bad_file_and_moves_on(ctx, tmp_path):
    paths, expect = write_sequence(tmp_path, np.random.default_rng(13), 6)
    good = open(paths[2], "rb").read()
    open(paths[2], "wb").write(good[:len(good) // 2])
    paths[4] = str(tmp_path / "missing.png")
    fs = ingest.FrameStream(paths, threads=2, depth=3)
    got = []
    for k in range(6):
        try:
            next(fs)
            assert np.array_equal(fs.read(), expect[k])
            got.append(k)
        except _lib.VoError as e:
            assert ("png:" in str(e) and "f0002" in str(e)) if k == 2 else ("cannot open" in str(e))
    assert got == [0, 1, 3, 5]
    with pytest.raises(StopIteration):
        next(fs)
    fs.close()


def test_pool_closes_early_and_handles_no_files(ctx, tmp_path):
    paths, _ = write_sequence(tmp_path, np.random.default_rng(14), 12)
    fs = ingest.FrameStream(paths, threads=4, depth=4)
    next(fs)
    fs.close()                                                             # workers are mid-sequence
    empty = ingest.FrameStream([])
    assert len(empty) == 0
    with pytest.raises(StopIteration):
        next(empty)
    empty.close()
    with pytest.raises(_lib.VoError):
        fs.read()


def test_stream_feeds_the_frame_loop_like_host_images(ctx, tmp_path):
    """processFrame on the pool's device images == processFrame on the same images uploaded from the host."""
    K = np.array([[300.0, 0, 160.0], [0, 300.0, 120.0], [0, 0, 1.0]])
    frames, poses, lift = synth.plane_sequence(6, K, nv=240, nu=320)
    paths = []
    for i, f in enumerate(frames):
        p = tmp_path / ("%06d.png" % i)
        p.write_bytes(oi.png_encode(np.repeat(f[:, :, None], 3, 2) if i % 2 else f))   # gray and (gray-valued) colour files
        paths.append(str(p))
    from oracle import oracle, vo_frame
    u, v, _, _ = oracle.select_candidates(frames[0], np.zeros(0, np.float32), np.zeros(0, np.float32), min_distance=7, keep=120)
    st = vo_frame.new_state()
    st.keypoints = np.stack([u, v], 1).astype(np.float64)
    st.landmarks = lift(st.keypoints[:, 0], st.keypoints[:, 1])
    st.reproError = np.zeros(len(u))
    a, b = FrameLoop(K, frames[0].shape, maxNumLandmarks=150), FrameLoop(K, frames[0].shape, maxNumLandmarks=150)
    a.set_state(st, frames[0]); b.set_state(st, frames[0])
    fs = ingest.FrameStream(paths[1:], threads=2, depth=2)
    for k, img in enumerate(fs, start=1):
        pa, sa, ia = a.processFrame(img)
        pb, sb, ib = b.processFrame(frames[k])
        assert sa == sb and ia == ib and np.array_equal(pa, pb), k
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa.landmarks, sb.landmarks) and np.array_equal(sa.candidate_keypoints, sb.candidate_keypoints)
    fs.close(); a.close(); b.close()


JPEG_DIR = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "jpeg")


@pytest.mark.parametrize("keep_rgb", [False, True])
def test_pool_decodes_jpeg_fixtures_on_the_device(ctx, keep_rgb):
    """Committed JPEG files (gray, 4:4:4, 4:2:2, 4:2:0, restart intervals, optimised tables): host Huffman decoding,
    then IDCT + upsampling + colour conversion (+ rgb2gray) on the GPU, bit-exact against the oracle's N12 / N11."""
    import os
    exp = np.load(os.path.join(JPEG_DIR, "jpeg_expected.npz"), allow_pickle=False)
    names = sorted(exp.files)
    fs = ingest.FrameStream([os.path.join(JPEG_DIR, n + ".jpg") for n in names] * 2, threads=2, depth=3, keep_rgb=keep_rgb)
    for k, img in enumerate(fs):
        e = exp[names[k % len(names)]]
        assert np.array_equal(oi.jpeg_decode(open(fs.paths[k], "rb").read()), e)       # the oracle itself, on the box
        want = e if (keep_rgb or e.ndim == 2) else oi.rgb2gray(e)
        assert img.shape == want.shape, names[k % len(names)]
        assert np.array_equal(fs.read(), want), names[k % len(names)]
    fs.close()


def test_pool_jpeg_at_malaga_size_and_mixed_with_png(ctx, tmp_path):
    """800 x 600 colour JPEGs (the Malaga extract's shape) and odd sizes: the device back half equals the host decoder
    (itself pinned to the oracle and to Pillow in tests/test_ingest.py) byte for byte; PNG files in between make the
    slots change roles."""
    PIL_Image = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(21)
    paths, want_gray, want_rgb = [], [], []
    for i, (shape, sub, q) in enumerate([((600, 800, 3), 2, 85), ((600, 800, 3), 1, 92), ((601, 799, 3), 2, 70),
                                         ((600, 800), 0, 80), ((50, 70, 3), 0, 95), ((600, 800, 3), 0, 60)]):
        img = textured(rng, *shape)
        b = io.BytesIO()
        PIL_Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub, restart_marker_rows=(2 if i == 1 else 0))
        p = tmp_path / ("j%02d.jpg" % i)
        p.write_bytes(b.getvalue())
        dec = ingest.imdecode(b.getvalue())
        assert np.array_equal(dec, np.asarray(PIL_Image.open(io.BytesIO(b.getvalue()))))
        paths.append(str(p)); want_rgb.append(dec); want_gray.append(oi.to_gray(dec))
        png = textured(rng, 40, 60, 3)
        p = tmp_path / ("p%02d.png" % i)
        p.write_bytes(oi.png_encode(png))
        paths.append(str(p)); want_rgb.append(png); want_gray.append(oi.rgb2gray(png))
    for keep_rgb, want in ((False, want_gray), (True, want_rgb)):
        fs = ingest.FrameStream(paths, threads=3, depth=2, keep_rgb=keep_rgb)
        for k, img in enumerate(fs):
            assert img.shape == want[k].shape, k
            assert np.array_equal(fs.read(), want[k]), k
        fs.close()


def device_bytes(ctx, d):
    """a DeviceImage's pixels on the host (hipMemcpy of the runtime the library is linked to)."""
    import ctypes as C
    ctx.sync()
    out = np.empty(d.shape, np.uint8)
    ctx.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert ctx.lib.hipMemcpy(_lib.ptr(out), C.c_void_p(d.data_ptr()), out.size, 2) == 0      # 2 = device to host
    return out


def test_two_frames_in_the_consumers_hands(ctx, tmp_path):
    """hold = 2 (what FrameLoop.processFrame(curr, next) needs): the frame handed out before the current one stays
    valid and unchanged while the pool decodes on; with more frames than slots every slot is reused."""
    rng = np.random.default_rng(71)
    imgs = [rng.integers(0, 256, (60, 80), dtype=np.uint8) for _ in range(12)]
    paths = []
    for i, im in enumerate(imgs):
        p = tmp_path / ("h%02d.png" % i)
        p.write_bytes(oi.png_encode(im))
        paths.append(str(p))
    fs = ingest.FrameStream(paths, threads=2, depth=3, hold=2)
    prev = next(fs)
    for k in range(1, len(imgs)):
        cur = next(fs)
        assert np.array_equal(fs.read(), imgs[k])                      # the current one, through the pool
        assert np.array_equal(device_bytes(ctx, prev), imgs[k - 1]), k  # the one before it is still intact
        assert np.array_equal(device_bytes(ctx, cur), imgs[k])
        prev = cur
    assert next(fs, None) is None
    fs.close()
    with pytest.raises(_lib.VoError):
        ingest.FrameStream(paths, threads=1, depth=3, hold=3)          # hold <= depth - 1
