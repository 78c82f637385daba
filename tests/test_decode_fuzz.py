"""The host decoders on damaged files, under the address and undefined-behaviour sanitizers (host code only:
vo_decode.cpp and vo_jpeg.cpp built with g++ beside tests/fuzz/decode_fuzz.cpp, no HIP, no GPU).  The files
a run reads come from disk and are not trusted: whatever is in them, the decoders return VO_OK or an error code."""
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import ingest as oi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "visual_odometry_amd", "csrc")
JPEG_DIR = os.path.join(ROOT, "tests", "golden", "jpeg")


def textured(rng, *shape):
    a = rng.integers(0, 256, shape).astype(np.float64)
    for ax in (0, 1):
        a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax)) / 3
    return a.astype(np.uint8)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("fuzz") / "decode_fuzz")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-static-libasan", "-static-libubsan", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           os.path.join(ROOT, "tests", "fuzz", "decode_fuzz.cpp"), os.path.join(CSRC, "vo_decode.cpp"),
           os.path.join(CSRC, "vo_jpeg.cpp"), "-lz", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        pytest.skip("sanitizer build not possible here: " + r.stderr[-300:])
    return exe


def seeds(tmp_path):
    rng = np.random.default_rng(1)
    files = {
        "gray.png": oi.png_encode(textured(rng, 40, 52), [0, 1, 2, 3, 4]),      # every row filter
        "rgb.png": oi.png_encode(textured(rng, 24, 31, 3), [4, 3, 2, 1]),
        "rgba.png": oi.png_encode(textured(rng, 12, 9, 4)),
        "gray.pgm": b"P5\n# c\n21 17\n255\n" + textured(rng, 17, 21).tobytes(),
        "wide.ppm": b"P6 13 9 1023\n" + rng.integers(0, 1024, (9, 13, 3)).astype(">u2").tobytes(),
    }
    try:
        from PIL import Image
        b = io.BytesIO(); Image.fromarray(textured(rng, 30, 33, 3)).quantize(16).save(b, "PNG", bits=4)
        files["palette.png"] = b.getvalue()
        b = io.BytesIO(); Image.fromarray(rng.integers(0, 65536, (19, 23)).astype(np.uint16)).save(b, "PNG")
        files["sixteen.png"] = b.getvalue()
    except ImportError:
        pass
    paths = []
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    paths += sorted(os.path.join(JPEG_DIR, f) for f in os.listdir(JPEG_DIR) if f.endswith(".jpg"))
    return paths


@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_files_end_in_a_return_code(harness, tmp_path, seed):
    r = subprocess.run([harness, "1200", str(seed)] + seeds(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    decoded, refused = [int(w) for w in r.stdout.split() if w.isdigit()]
    assert decoded > 0 and refused > 0        # both outcomes were exercised


def _segments(jpg):
    """(marker, start, end) of the header segments of a baseline JPEG up to and including SOS."""
    pos, out = 2, []
    while pos < len(jpg):
        assert jpg[pos] == 0xFF
        m = jpg[pos + 1]
        ln = int.from_bytes(jpg[pos + 2:pos + 4], "big")
        out.append((m, pos, pos + 2 + ln))
        pos += 2 + ln
        if m == 0xDA:
            break
    return out


def crafted_bad_jpegs(tmp_path):
    """Files a fuzzer finds slowly: (1) a second SOF0 four times the size behind the first -- the header the caller
    sizes `out` from is the first, a decoder that lets the second one win writes past it; (2) the scan
    cut in the middle and closed with EOI -- the rest of the frame would decode from zero fill; (3) a stray non-RST
    marker in the middle of the scan."""
    good = open(os.path.join(JPEG_DIR, "gray_q75.jpg"), "rb").read()
    segs = _segments(good)
    sof = next(s for s in segs if s[0] == 0xC0)
    sos = next(s for s in segs if s[0] == 0xDA)
    seg = bytearray(good[sof[1]:sof[2]])
    h, w = int.from_bytes(seg[5:7], "big"), int.from_bytes(seg[7:9], "big")
    seg[5:7] = (2 * h).to_bytes(2, "big"); seg[7:9] = (2 * w).to_bytes(2, "big")
    two_sof = good[:sof[2]] + bytes(seg) + good[sof[2]:]
    body = good[sos[2]:-2]
    early_eoi = good[:sos[2]] + body[:len(body) // 2] + b"\xff\xd9"
    stray = good[:sos[2]] + body[:len(body) // 3] + b"\xff\xe1\x00\x02" + body[len(body) // 3:] + b"\xff\xd9"
    paths = []
    for name, data in (("two_sof.jpg", two_sof), ("early_eoi.jpg", early_eoi), ("stray_marker.jpg", stray)):
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    return paths


def test_crafted_bad_jpegs_are_refused(harness, tmp_path):
    r = subprocess.run([harness, "refuse"] + crafted_bad_jpegs(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "more than one frame header" in r.stdout and r.stdout.count("ends before the last MCU") == 2, r.stdout
