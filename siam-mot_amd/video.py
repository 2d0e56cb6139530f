"""Frame iterators feeding the tracker.

Mirrors demos/video_iterator.py:9-138 — ``ImageFolderIterator`` (:99-125), ``DecordVideoIterator`` (:9-42),
``CV2VideoIterator`` (:45-82) and ``build_video_iterator`` (:128-138): an iterator object is CALLED to obtain a
generator of ``(frame_idx, frame)`` with ``frame`` an RGB uint8 ``[H, W, 3]`` array, ``len()`` is the number of
frames that will be produced, ``frame_idxs`` selects (and sorts) a subset.

The reference decodes image folders with cv2 and videos with decord / cv2 (+ ffmpeg for the rotation tag).  None of
the three is a dependency here: the folder reader uses Pillow (the library the reference's own transform starts from,
demo_inference.py:77), and the two container readers import their decoder lazily and fail loudly when it is not
installed — there is no silent substitute for a video decoder.

What the iterators hand out goes to ``FramePreprocessor`` (preprocess.py) as bytes: the resize / normalise runs on the
GPU, and ``prefetch`` decodes the next frame on a worker thread while the current one is being tracked.
"""
import glob
import os
import queue
import threading

import numpy as np


class ImageFolderIterator(object):
    """A video stored as a folder of JPEG frames (video_iterator.py:99-125): files ``*.jpg`` in sorted order."""

    def __init__(self, video_folder, frame_idxs=None, pattern="*.jpg"):
        if not os.path.isdir(video_folder):
            raise FileNotFoundError("ImageFolderIterator: %s is not a directory" % video_folder)
        self.vr = sorted(glob.glob(os.path.join(video_folder, pattern)))
        if frame_idxs is None:
            self._frame_idxs = np.arange(len(self.vr))
        else:
            self._frame_idxs = sorted(frame_idxs)

    def __len__(self):
        return len(self._frame_idxs)

    def video_len(self):
        return len(self.vr)

    def read(self, frame_idx):
        """RGB uint8 ``[H,W,3]`` (the reference reads BGR with cv2 and flips it: same bytes for baseline JPEG
        up to the decoder's IDCT; both are libjpeg builds)."""
        from PIL import Image
        with Image.open(self.vr[frame_idx]) as im:
            return np.array(im.convert("RGB"))          # a writable copy (torch.from_numpy wants one)

    def __call__(self):
        for idx in range(len(self)):
            frame_idx = int(self._frame_idxs[idx])
            yield frame_idx, self.read(frame_idx)


class ArrayVideoIterator(object):
    """Frames already in memory (``[T,H,W,3]`` uint8 array or a list of arrays): the same contract, used by the
    synthetic benchmark streams and tests."""

    def __init__(self, frames, frame_idxs=None):
        self.vr = frames
        self._frame_idxs = np.arange(len(frames)) if frame_idxs is None else sorted(frame_idxs)

    def __len__(self):
        return len(self._frame_idxs)

    def video_len(self):
        return len(self.vr)

    def __call__(self):
        for idx in range(len(self)):
            frame_idx = int(self._frame_idxs[idx])
            yield frame_idx, np.asarray(self.vr[frame_idx])


def _rotate(frame, rotation):
    """video_iterator.py:39-40: undo the container's rotation tag."""
    if rotation > 0:
        frame = np.rot90(frame, k=(-(rotation // 90)) % 4)
    return frame


def check_rotation(video_file):
    """video_iterator.py:85-93 (ffmpeg.probe); 0 when ffmpeg-python is not installed."""
    try:
        import ffmpeg
    except ImportError:
        return 0
    meta = ffmpeg.probe(video_file)
    tags = meta["streams"][0].get("tags", {})
    return int(tags["rotate"]) if "rotate" in tags else 0


class DecordVideoIterator(object):
    """video_iterator.py:9-42; needs the ``decord`` package."""

    def __init__(self, video_file, frame_idxs=None):
        try:
            from decord import VideoReader, cpu
        except ImportError as e:
            raise RuntimeError("DecordVideoIterator: the 'decord' video decoder is not installed (%s); decode the "
                               "video to a JPEG folder and use ImageFolderIterator" % e)
        self.vr = VideoReader(video_file, ctx=cpu(0))
        self._rotation = check_rotation(video_file)
        self._frame_idxs = np.arange(len(self.vr)) if frame_idxs is None else sorted(frame_idxs)

    def __len__(self):
        return len(self._frame_idxs)

    def video_len(self):
        return len(self.vr)

    def __call__(self):
        for idx in range(len(self)):
            frame_idx = int(self._frame_idxs[idx])
            yield frame_idx, _rotate(self.vr[frame_idx].asnumpy(), self._rotation)


class CV2VideoIterator(object):
    """video_iterator.py:45-82; needs ``cv2``."""

    def __init__(self, video_file, frame_idxs=None):
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError("CV2VideoIterator: OpenCV is not installed (%s); decode the video to a JPEG folder "
                               "and use ImageFolderIterator" % e)
        self._cv2 = cv2
        vr = cv2.VideoCapture(video_file)
        assert vr.isOpened(), "Cannot open the video file: {}".format(video_file)
        self.vr = vr
        self._rotation = check_rotation(video_file)
        n = int(vr.get(cv2.CAP_PROP_FRAME_COUNT))
        self._frame_idxs = np.arange(n) if frame_idxs is None else sorted(frame_idxs)

    def __len__(self):
        return len(self._frame_idxs)

    def __call__(self):
        for idx in range(len(self)):
            frame_idx = int(self._frame_idxs[idx])
            self.vr.set(self._cv2.CAP_PROP_POS_FRAMES, frame_idx)
            ok, frame = self.vr.read()
            if not ok:
                break
            yield frame_idx, _rotate(frame, self._rotation)[:, :, ::-1]      # BGR -> RGB


def build_video_iterator(video_path, video_decode="decord"):
    """video_iterator.py:128-138."""
    if os.path.isdir(video_path):
        return ImageFolderIterator(video_path)
    if video_decode == "decord":
        return DecordVideoIterator(video_path)
    return CV2VideoIterator(video_path)


def prefetch(frame_generator, depth=2):
    """Run ``frame_generator`` (what calling an iterator returns) on a worker thread, ``depth`` frames ahead:
    JPEG decode of frame t+1 overlaps the GPU work of frame t (the reference decodes inline)."""
    q = queue.Queue(maxsize=max(1, depth))
    end = object()
    stop = threading.Event()

    def put(item):
        """Blocking put that gives up when the consumer is gone (it stopped early: an exception in the tracker, a
        ``break``): the worker must not sit in ``q.put`` for ever holding the decoder and ``depth`` frames."""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for item in frame_generator:
                if not put(item):
                    break
            else:
                put(end)
        except BaseException as e:          # surface decoder errors on the consumer side
            put(e)
        finally:
            close = getattr(frame_generator, "close", None)
            if close is not None:
                close()
    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item = q.get()
            if item is end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:                                # generator closed or exhausted: release the worker
        stop.set()
