"""Flow file IO with the reference's names (pytorch/raft_utils/frame_utils.py): host-side data formats either side of the
path.  Pure numpy; nothing here touches the device.

  readFlow / writeFlow      Middlebury .flo (frame_utils.py:14-33, 74-103): float32 tag 202021.25, int32 width, int32
                            height, then height x width interleaved (u, v) float32, all little-endian
  readPFM                   frame_utils.py:36-71
  readFlowKITTI / writeFlowKITTI / readDispKITTI   16-bit PNG triplets (frame_utils.py:106-125); they need OpenCV, which the
                            reference imports at module level -- here it is imported on first use
  read_gen                  dispatch on the file extension (frame_utils.py:128-142)
"""
import re
from os.path import splitext

import numpy as np

TAG_FLOAT = 202021.25
TAG_CHAR = np.array([TAG_FLOAT], np.float32)


def readFlow(fn):
    """[H,W,2] float32, or None (with the reference's message) when the tag is wrong."""
    with open(fn, "rb") as f:
        magic = np.frombuffer(f.read(4), "<f4")
        if magic.size != 1 or magic[0] != np.float32(TAG_FLOAT):
            print("Magic number incorrect. Invalid .flo file")
            return None
        w, h = (int(v) for v in np.frombuffer(f.read(8), "<i4"))
        data = np.frombuffer(f.read(8 * w * h), "<f4")
    return np.resize(data, (h, w, 2))


def writeFlow(filename, uv, v=None):
    """uv: [H,W,2], or the u plane with v given separately."""
    if v is None:
        uv = np.asarray(uv)
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u, v = np.asarray(uv), np.asarray(v)
    assert u.shape == v.shape
    h, w = u.shape
    with open(filename, "wb") as f:
        f.write(TAG_CHAR.astype("<f4").tobytes())
        f.write(np.array([w, h], "<i4").tobytes())
        f.write(np.stack([u, v], axis=-1).astype("<f4").tobytes())


def readPFM(file):
    with open(file, "rb") as f:
        header = f.readline().rstrip()
        if header == b"PF":
            color = True
        elif header == b"Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        m = re.match(rb"^(\d+)\s(\d+)\s$", f.readline())
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    return np.flipud(np.reshape(data, (height, width, 3) if color else (height, width)))


def _cv2():
    import cv2
    cv2.setNumThreads(0)
    cv2.ocl.setUseOpenCL(False)
    return cv2


def readFlowKITTI(filename):
    cv2 = _cv2()
    raw = cv2.imread(filename, cv2.IMREAD_ANYDEPTH | cv2.IMREAD_COLOR)[:, :, ::-1].astype(np.float32)
    return (raw[:, :, :2] - 2 ** 15) / 64.0, raw[:, :, 2]


def readDispKITTI(filename):
    cv2 = _cv2()
    disp = cv2.imread(filename, cv2.IMREAD_ANYDEPTH) / 256.0
    return np.stack([-disp, np.zeros_like(disp)], -1), disp > 0.0


def writeFlowKITTI(filename, uv):
    cv2 = _cv2()
    uv = 64.0 * uv + 2 ** 15
    out = np.concatenate([uv, np.ones([uv.shape[0], uv.shape[1], 1])], axis=-1).astype(np.uint16)
    cv2.imwrite(filename, out[..., ::-1])


def read_gen(file_name, pil=False):
    ext = splitext(file_name)[-1]
    if ext in (".png", ".jpeg", ".ppm", ".jpg"):
        from PIL import Image
        return Image.open(file_name)
    if ext in (".bin", ".raw"):
        return np.load(file_name)
    if ext == ".flo":
        return readFlow(file_name).astype(np.float32)
    if ext == ".pfm":
        flow = readPFM(file_name).astype(np.float32)
        return flow if flow.ndim == 2 else flow[:, :, :-1]
    return []
