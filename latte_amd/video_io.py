"""Video hand-off without third-party encoders.

The reference writes ``.mp4`` with imageio (sample/sample.py:124-126, sample_ddp.py:174-176: ``imageio.mimwrite(path,
video, fps=8, quality=9)``); neither imageio nor an H.264 encoder exists offline, so the drivers here write the same
uint8 ``[F, H, W, 3]`` RGB frames as an UNCOMPRESSED AVI (RIFF, 24-bit DIB frames) that every player and ffmpeg reads,
or as ``.npy``.  Pure Python + numpy; host-side plumbing only.
"""
import struct

import numpy as np


def _chunk(tag: bytes, payload: bytes) -> bytes:
    pad = b"\x00" if len(payload) & 1 else b""
    return tag + struct.pack("<I", len(payload)) + payload + pad


def _list(kind: bytes, payload: bytes) -> bytes:
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


def write_avi(path, video, fps=8):
    """video: uint8 array / tensor [F, H, W, 3] (RGB).  Writes an uncompressed 24-bit AVI."""
    v = np.asarray(video.cpu() if hasattr(video, "cpu") else video)
    if v.dtype != np.uint8 or v.ndim != 4 or v.shape[3] != 3:
        raise ValueError("write_avi expects uint8 [F, H, W, 3]")
    f, h, w, _ = v.shape
    stride = (w * 3 + 3) & ~3                                   # DIB rows are padded to 4 bytes
    frame_bytes = stride * h
    frames = []
    for i in range(f):
        bgr = v[i, ::-1, :, ::-1]                               # bottom-up rows, BGR byte order
        if stride != w * 3:
            buf = np.zeros((h, stride), dtype=np.uint8)
            buf[:, : w * 3] = bgr.reshape(h, w * 3)
            frames.append(buf.tobytes())
        else:
            frames.append(np.ascontiguousarray(bgr).tobytes())
    usec = int(round(1e6 / fps))
    avih = struct.pack("<14I", usec, frame_bytes * fps, 0, 0x10, f, 0, 1, frame_bytes, w, h, 0, 0, 0, 0)   # 0x10: HASINDEX
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"DIB ", 0, 0, 0, 0, 1, int(fps), 0, f, frame_bytes, 0xFFFFFFFF, 0,
                       0, 0, w, h)
    strf = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, frame_bytes, 0, 0, 0, 0)                        # BITMAPINFOHEADER
    hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", strf)))
    movi_payload = b"".join(_chunk(b"00db", fr) for fr in frames)
    movi = _list(b"movi", movi_payload)
    idx, off = [], 4                                            # offsets are relative to the 'movi' fourcc
    for fr in frames:
        idx.append(struct.pack("<4sIII", b"00db", 0x10, off, len(fr)))
        off += 8 + len(fr) + (len(fr) & 1)
    body = b"AVI " + hdrl + movi + _chunk(b"idx1", b"".join(idx))
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_avi(path):
    """Inverse of ``write_avi`` for files it wrote (tests / round trips): -> (uint8 [F, H, W, 3], fps)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("not an AVI file")
    p = data.index(b"avih") + 8
    usec, _, _, _, f, _, _, _, w, h = struct.unpack_from("<10I", data, p)
    stride = (w * 3 + 3) & ~3
    q = data.index(b"movi") + 4
    out = np.empty((f, h, w, 3), dtype=np.uint8)
    for i in range(f):
        tag, n = data[q:q + 4], struct.unpack_from("<I", data, q + 4)[0]
        if tag != b"00db" or n != stride * h:
            raise ValueError("unexpected chunk in movi list")
        rows = np.frombuffer(data, dtype=np.uint8, count=n, offset=q + 8).reshape(h, stride)[:, : w * 3].reshape(h, w, 3)
        out[i] = rows[::-1, :, ::-1]
        q += 8 + n + (n & 1)
    return out, 1e6 / usec
