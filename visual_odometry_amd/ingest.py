"""Frame ingest over vo_image_* / vo_rgb2gray / vo_ingest_* (include/vo_hip.h).

Mirrors what the reference's frame loops do per frame -- `imread`, `rgb2gray`, `im2uint8`
(main.m:15-30,52-59) -- and its dataset loaders, same names and members:

    dataLoaderKitti(path)    utils/dataLoaderKitti.m    <path>/00/image_0/%06d.png, <path>/poses/00.txt
    dataLoaderParking(path)  utils/dataLoaderParking.m  <path>/images/img_%05d.png, K.txt, poses.txt
    dataLoaderMalaga(path)   utils/dataLoaderMalaga.m   every other file of ..._rectified_800x600_Images
    loadGeneralData(ds)      utils/loadGeneralData.m

`next()` / `retrieveFrame()` return host arrays like the reference; `loader.stream()` is the device path:
a FrameStream whose worker threads decode ahead and upload on their own copy stream, yielding device
images that FrameLoop.processFrame takes as they are.

The decoders are native (libvo_hip.so): PNG, baseline JPEG (Malaga's extract) and binary PGM/PPM.  In a
FrameStream a JPEG's Huffman decoding runs on the worker threads and everything after it on the GPU.
There is no fallback decoder.
"""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np

from . import _lib
from ._lib import VoError, ptr
from .modules import CameraParams

KITTI_K = np.array([[7.188560000000e+02, 0, 6.071928000000e+02],
                    [0, 7.188560000000e+02, 1.852157000000e+02],
                    [0, 0, 1]])                                    # utils/dataLoaderKitti.m:50-52
MALAGA_K = np.array([[621.18428, 0, 404.0076],
                     [0, 621.18428, 309.05989],
                     [0, 0, 1]])                                   # utils/dataLoaderMalaga.m:49-51
MALAGA_DIR = "malaga-urban-dataset-extract-07_rectified_800x600_Images"


class IngestOpts(C.Structure):
    _fields_ = [("threads", C.c_int), ("depth", C.c_int), ("keep_rgb", C.c_int)]


def _check(lib, rc, what):
    if rc != 0:
        raise VoError(f"{what} failed: {lib.vo_strerror(rc).decode()}")


def imdecode(data):
    """imread + im2uint8 of an encoded file held in memory -> (H, W) or (H, W, 3) uint8.  Host only."""
    lib = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    H, W, ch, bd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _check(lib, lib.vo_image_info(ptr(buf), C.c_size_t(len(buf)), C.byref(H), C.byref(W), C.byref(ch), C.byref(bd)),
           "vo_image_info")
    out = np.empty((H.value, W.value) if ch.value == 1 else (H.value, W.value, 3), np.uint8)
    _check(lib, lib.vo_image_decode(ptr(buf), C.c_size_t(len(buf)), ptr(out), C.c_size_t(out.size)), "vo_image_decode")
    return out


def imread(path):
    with open(path, "rb") as f:
        return imdecode(f.read())


def rgb2gray(img, ctx=None):
    """(H, W, 3) uint8 host array or device tensor -> (H, W) uint8 of the same kind, on the GPU."""
    ctx = ctx or _lib.Context.default()
    if isinstance(img, np.ndarray):
        a = np.ascontiguousarray(img, np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("rgb2gray: H x W x 3 image expected")
        out = np.empty(a.shape[:2], np.uint8)
        ctx.call("vo_rgb2gray", ptr(a), a.shape[0], a.shape[1], 0, 0, ptr(out))
        return out
    import torch
    if img.dim() != 3 or img.shape[2] != 3 or img.dtype != torch.uint8 or not img.is_contiguous():
        raise ValueError("rgb2gray: contiguous H x W x 3 uint8 tensor expected")
    out = torch.empty(img.shape[:2], dtype=torch.uint8, device=img.device)
    ctx.call("vo_rgb2gray", ptr(img), img.shape[0], img.shape[1], 0, 1, ptr(out))
    return out


def im2uint8(img):
    a = np.asarray(img)
    if a.dtype == np.uint8:
        return a
    if a.dtype == np.uint16:
        return ((a.astype(np.uint32) + 128) // 257).astype(np.uint8)
    raise TypeError("im2uint8: uint8 or uint16 input")


class DeviceImage:
    """A frame in HBM owned by a FrameStream; valid until the stream's next step.  Quacks like the device
    tensors FrameLoop takes (shape / is_contiguous / data_ptr)."""

    def __init__(self, p, H, W, ch, index):
        self._p, self.index, self.channels = p, index, ch
        self.shape = (H, W) if ch == 1 else (H, W, ch)

    def data_ptr(self):
        return self._p

    def is_contiguous(self):
        return True


class FrameStream:
    """Decode pool over a list of files -> device images in order (vo_ingest_*)."""

    def __init__(self, paths, threads=0, depth=0, keep_rgb=False, ctx=None, hold=1):
        """hold: how many of the frames handed out stay valid (2: the current one and the one after it, which is
        what FrameLoop.processFrame(curr, next_img) needs)."""
        self.ctx = ctx or _lib.Context.default()
        self.paths = [os.fspath(p) for p in paths]
        arr = (C.c_char_p * len(self.paths))(*[p.encode() for p in self.paths])
        self.h = C.c_void_p()
        opts = IngestOpts(int(threads), int(depth), int(bool(keep_rgb)))
        self.ctx.call("vo_ingest_open", arr, len(self.paths), C.byref(opts), C.byref(self.h))
        self.lib = self.ctx.lib
        if int(hold) != 1:
            self.ctx.check(self.lib.vo_ingest_set_hold(self.h, int(hold)), "vo_ingest_set_hold")
        self.lib.vo_ingest_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.current = None

    def __len__(self):
        return len(self.paths)

    def __iter__(self):
        return self

    def __next__(self):
        if not self.h:
            raise StopIteration
        p, H, W, ch, idx = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.vo_ingest_next(self.h, C.byref(p), C.byref(H), C.byref(W), C.byref(ch), C.byref(idx))
        if rc == 1:                                                   # VO_END
            self.current = None
            raise StopIteration
        self.ctx.check(rc, "vo_ingest_next")
        self.current = DeviceImage(p.value, H.value, W.value, ch.value, idx.value)
        return self.current

    def read(self):
        """The current frame as a host array (synchronises the stream)."""
        if self.current is None:
            raise VoError("FrameStream.read: no current frame")
        out = np.empty(self.current.shape, np.uint8)
        self.ctx.check(self.lib.vo_ingest_read(self.h, ptr(out), C.c_size_t(out.size)), "vo_ingest_read")
        return out

    def stats(self):
        s, t, d = C.c_double(), C.c_int(), C.c_int()
        self.ctx.check(self.lib.vo_ingest_stats(self.h, C.byref(s), C.byref(t), C.byref(d)), "vo_ingest_stats")
        return SimpleNamespace(decode_seconds=s.value, threads=t.value, depth=d.value)

    def close(self):
        if self.h:
            self.lib.vo_ingest_close(self.h)
            self.h = None
            self.current = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _load_ground_truth(path):
    """`load(path)` then columns [end-8 end] (utils/loadGeneralData.m:13-14,36-37): x and z of a 3x4 pose row."""
    g = np.loadtxt(path, dtype=np.float64, ndmin=2)
    return g[:, [g.shape[1] - 9, g.shape[1] - 1]]


class _Loader:
    """Members and methods of the reference's dataLoader* handle classes (index is zero-based there too)."""

    def __init__(self, path):
        self.dataset_path = os.fspath(path)
        self.index = 0
        self.finished = False

    def _path(self, index):
        raise NotImplementedError

    def _gt(self, index):
        return self.ground_truth_data[index] if len(self.ground_truth_data) else np.zeros((0,))

    def next(self):
        if not self.index <= self.last_frame:                         # dataLoaderKitti.m:28
            raise AssertionError("all frames were already loaded")
        img = imread(self._path(self.index))
        gt = self._gt(self.index)
        self.index += 1
        if self.index > self.last_frame:
            self.finished = True
        return img, gt

    def retrieveFrame(self, index):
        if not index <= self.last_frame:
            raise AssertionError("index out of range")
        # the reference returns the ground truth of the loader's CURRENT index here (dataLoaderKitti.m:42)
        return imread(self._path(index)), self._gt(min(self.index, self.last_frame))

    def reset(self, index=0):
        if not index <= self.last_frame:
            raise AssertionError("Index out of range")
        self.index = int(index)
        self.finished = False

    def paths(self, start=None, stop=None):
        start = self.index if start is None else int(start)
        stop = self.last_frame + 1 if stop is None else int(stop)
        return [self._path(i) for i in range(start, stop)]

    def stream(self, start=None, stop=None, **kw):
        """Device path: frames [start, stop) as a FrameStream (gray device images)."""
        return FrameStream(self.paths(start, stop), **kw)


class dataLoaderKitti(_Loader):
    """utils/dataLoaderKitti.m reads sequence 00 only (:21,47); `sequence` lets the sharded runner take the others
    (BASELINE configs[4]); K stays the reference's sequence-00 matrix unless `K` is given."""

    def __init__(self, path, last_frame=4540, sequence="00", K=None):
        super().__init__(path)
        self.camParams = CameraParams(KITTI_K if K is None else K)
        self.ground_truth_data = _load_ground_truth(os.path.join(self.dataset_path, "poses", "%s.txt" % sequence))
        self.last_frame = int(last_frame)                            # dataLoaderKitti.m:49
        self.img_file_format = os.path.join(self.dataset_path, sequence, "image_0", "%06d.png")

    def _path(self, index):
        return self.img_file_format % index


class dataLoaderParking(_Loader):
    def __init__(self, path, last_frame=598):
        super().__init__(path)
        K = np.loadtxt(_strip_commas(os.path.join(self.dataset_path, "K.txt")), ndmin=2)
        self.camParams = CameraParams(K)
        self.ground_truth_data = _load_ground_truth(os.path.join(self.dataset_path, "poses.txt"))
        self.last_frame = int(last_frame)                            # dataLoaderParking.m:51
        self.img_file_format = os.path.join(self.dataset_path, "images", "img_%05d.png")

    def _path(self, index):
        return self.img_file_format % index


class dataLoaderMalaga(_Loader):
    def __init__(self, path):
        super().__init__(path)
        self.camParams = CameraParams(MALAGA_K)
        self.ground_truth_data = np.zeros((0, 2))
        d = os.path.join(self.dataset_path, MALAGA_DIR)
        names = sorted(os.listdir(d))
        self.left_images_dirs = [os.path.join(d, n) for n in names[0::2]]   # images(3:2:end), :56
        self.last_frame = len(self.left_images_dirs) - 1

    def _path(self, index):
        return self.left_images_dirs[index]

    def _gt(self, index):
        return np.zeros((0,))


def _strip_commas(path):
    """MATLAB's `load` reads comma- or blank-separated numbers (the parking K.txt ends its rows with commas)."""
    import io
    with open(path) as f:
        return io.StringIO(f.read().replace(",", " "))


def loadGeneralData(ds, kitti_path="data/kitti", malaga_path="data/malaga-urban-dataset-extract-07",
                    parking_path="data/parking"):
    """utils/loadGeneralData.m: ds 0 KITTI, 1 Malaga, 2 parking -> (camParams, ground_truth, last_frame)."""
    if ds == 0:
        gt, last, K = _load_ground_truth(os.path.join(kitti_path, "poses", "00.txt")), 4540, KITTI_K.copy()
    elif ds == 1:
        n = len(sorted(os.listdir(os.path.join(malaga_path, MALAGA_DIR)))[0::2])
        gt, last, K = np.zeros((0, 2)), n, MALAGA_K.copy()          # :25: the count itself, not count - 1
    elif ds == 2:
        K = np.loadtxt(_strip_commas(os.path.join(parking_path, "K.txt")), ndmin=2)
        gt, last = _load_ground_truth(os.path.join(parking_path, "poses.txt")), 598
    else:
        raise AssertionError("ds must be 0, 1 or 2")
    return CameraParams(K), gt, last
