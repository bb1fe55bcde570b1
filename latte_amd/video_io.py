"""Video hand-off without third-party video encoders.

The reference writes ``.mp4`` with imageio (sample/sample.py:124-126, sample_ddp.py:174-176: ``imageio.mimwrite(path,
video, fps=8, quality=9)``); neither imageio nor an H.264 encoder exists offline.  The drivers here write the same uint8
``[F, H, W, 3]`` RGB frames

* as ``.mp4`` (``write_mp4``): an ISO base-media file whose one video track holds Motion-JPEG samples (sample entry ``mp4v``
  with ``esds`` objectTypeIndication 0x6C = ISO/IEC 10918-1, the form ffmpeg muxes and VLC / ffmpeg play), the frames
  JPEG-encoded by Pillow -- the reference's file name and container, a different (intra-only) codec inside;
* as an UNCOMPRESSED AVI (``write_avi``: RIFF, 24-bit DIB frames) where exact pixels matter or Pillow is absent.

Pure Python + numpy (+ Pillow for the JPEG frames); host-side plumbing only.
"""
import io
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


# ------------------------------------------------------------------------------------------------ .mp4 (Motion-JPEG samples)
def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", len(payload) + 8) + kind + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag: int, payload: bytes) -> bytes:
    n = len(payload)
    if n >= 1 << 21:
        raise ValueError("descriptor too large")
    return bytes([tag, 0x80 | (n >> 14) & 0x7F, 0x80 | (n >> 7) & 0x7F, n & 0x7F]) + payload   # 3-byte-extended length form


_MATRIX = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def write_mp4(path, video, fps=8, quality=95):
    """video: uint8 array / tensor [F, H, W, 3] (RGB) -> ``.mp4`` with one Motion-JPEG video track (``fps`` samples per second).

    ``quality`` is Pillow's JPEG quality (imageio's ``quality=9`` of the reference is its near-top setting; 95 with 4:4:4 chroma
    keeps the frames within ~1 grey level RMS of the uint8 input)."""
    from PIL import Image
    v = np.asarray(video.cpu() if hasattr(video, "cpu") else video)
    if v.dtype != np.uint8 or v.ndim != 4 or v.shape[3] != 3:
        raise ValueError("write_mp4 expects uint8 [F, H, W, 3]")
    f, h, w, _ = v.shape
    if f == 0 or w >= 1 << 16 or h >= 1 << 16:
        raise ValueError("write_mp4: empty video or frame larger than 65535 pixels")
    frames = []
    for i in range(f):
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(v[i]), "RGB").save(buf, format="JPEG", quality=int(quality), subsampling=0)
        frames.append(buf.getvalue())
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2mp41")
    mdat = _box(b"mdat", b"".join(frames))
    first = len(ftyp) + 8                                        # file offset of the first sample
    scale, delta = int(round(fps * 1000)), 1000                  # media time scale: 1000 ticks per frame
    dur = f * delta
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, scale, dur) + struct.pack(">IH", 0x10000, 0x100) + b"\0" * 10 +
                 _MATRIX + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur) + b"\0" * 8 + struct.pack(">hhhh", 0, 0, 0, 0) +
                 _MATRIX + struct.pack(">II", w << 16, h << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, scale, dur, 0x55C4, 0))            # language 'und'
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, b"\0" * 8)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    peak = max(len(x) for x in frames)
    rate = int(sum(len(x) for x in frames) * 8 * fps / f)
    dcd = _descr(0x04, bytes([0x6C, 0x11]) + struct.pack(">I", peak)[1:] + struct.pack(">II", rate, rate))   # JPEG, visual stream
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) + dcd + _descr(0x06, b"\x02")))
    entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HHIIIH", w, h, 0x480000, 0x480000, 0, 1) +
             b"\0" * 32 + struct.pack(">Hh", 24, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"mp4v", entry))
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, f, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, f, 1))                                # one chunk holding every sample
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, f) + b"".join(struct.pack(">I", len(x)) for x in frames))
    stco = _full(b"stco", 0, 0, struct.pack(">II", 1, first))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = _box(b"minf", vmhd + dinf + stbl)
    trak = _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))
    with open(path, "wb") as fh:
        fh.write(ftyp + mdat + _box(b"moov", mvhd + trak))


def _find_box(data, kind, start, end):
    p = start
    while p + 8 <= end:
        n, k = struct.unpack_from(">I4s", data, p)
        if n < 8:
            raise ValueError("malformed box")
        if k == kind:
            return p + 8, p + n
        p += n
    raise ValueError("box %r not found" % kind)


def read_mp4(path):
    """Inverse of ``write_mp4`` for files it wrote (tests / round trips): -> (uint8 [F, H, W, 3] decoded frames, fps)."""
    from PIL import Image
    with open(path, "rb") as fh:
        data = fh.read()
    if data[4:8] != b"ftyp":
        raise ValueError("not an ISO base-media file")
    a, b = _find_box(data, b"moov", 0, len(data))
    a, b = _find_box(data, b"trak", a, b)
    a, b = _find_box(data, b"mdia", a, b)
    m0, _ = _find_box(data, b"mdhd", a, b)
    scale = struct.unpack_from(">I", data, m0 + 12)[0]
    a, b = _find_box(data, b"minf", a, b)
    a, b = _find_box(data, b"stbl", a, b)
    t0, _ = _find_box(data, b"stts", a, b)
    delta = struct.unpack_from(">I", data, t0 + 12)[0]
    z0, _ = _find_box(data, b"stsz", a, b)
    f = struct.unpack_from(">I", data, z0 + 8)[0]
    sizes = struct.unpack_from(">%dI" % f, data, z0 + 12)
    c0, _ = _find_box(data, b"stco", a, b)
    off = struct.unpack_from(">I", data, c0 + 8)[0]
    frames = []
    for n in sizes:
        frames.append(np.asarray(Image.open(io.BytesIO(data[off:off + n])).convert("RGB")))
        off += n
    return np.stack(frames), scale / delta
