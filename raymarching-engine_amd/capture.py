"""PNG capture of a presented frame -- the reference's `canvas.toDataURL("image/png")`
(client/src/index.tsx:470-476).  Host-side only: the pixels come from the present pass on the
GPU (rm_present, display.frag), this module packs them into a PNG container with the standard
library (zlib).  Framebuffer rows are in GL order (row 0 = bottom); a PNG starts at the top."""
from __future__ import annotations

import base64
import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(rgba8: np.ndarray, bottom_up: bool = True, level: int = 6) -> bytes:
    """rgba8: uint8 [H, W, 4] (as returned by Framebuffer.present).  8-bit RGBA, no interlace, filter 0."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("encode_png: expected a uint8 array of shape [H, W, 4]")
    if bottom_up:
        a = a[::-1]
    h, w = a.shape[:2]
    if h < 1 or w < 1:
        raise ValueError("encode_png: empty image")
    rows = np.empty((h, 1 + w * 4), np.uint8)
    rows[:, 0] = 0  # filter type None
    rows[:, 1:] = a.reshape(h, w * 4)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
            + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b""))


def decode_png(data: bytes) -> np.ndarray:
    """Inverse of encode_png for the files it writes (RGBA8, filter 0 only); rows top-down.  For tests."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if crc != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError("bad CRC in chunk %r" % tag)
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 6, 0):
                raise ValueError("decode_png only reads what encode_png writes")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    if raw[:, 0].any():
        raise ValueError("decode_png only reads filter type 0")
    return raw[:, 1:].reshape(h, w, 4).copy()


def to_data_url(rgba8: np.ndarray) -> str:
    """The string `canvas.toDataURL("image/png")` returns."""
    return "data:image/png;base64," + base64.b64encode(encode_png(rgba8)).decode("ascii")


def save_png(framebuffer, samples: int, path: str) -> None:
    """Present `framebuffer` (rm_present: DoF blur, 1/samples, gamma) and write it as a PNG."""
    with open(path, "wb") as f:
        f.write(encode_png(framebuffer.present(samples)))
