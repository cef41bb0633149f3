"""PNG encoding of the RgbImage a render returns.

The reference ends with `image.write_to(&mut file, ImageFormat::Png)` (main.rs:581-582): an 8-bit RGB, non-interlaced PNG
of the row-major, top-row-first buffer.  A PNG decoder returns exactly the bytes that were encoded, whatever filter and
compression level the encoder chose, so pixel parity with the reference's file is parity of the RGB8 buffer.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_rgb8(rgb: np.ndarray, level: int = 6) -> bytes:
    """rgb: [H, W, 3] uint8 -> PNG file bytes (colour type 2, bit depth 8, filter 0 on every row)."""
    rgb = np.ascontiguousarray(rgb)
    if rgb.dtype != np.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError("expected an [H, W, 3] uint8 image")
    h, w, _ = rgb.shape
    if h == 0 or w == 0:
        raise ValueError("empty image")
    rows = np.zeros((h, 1 + 3 * w), dtype=np.uint8)  # leading filter-type byte per scanline
    rows[:, 1:] = rgb.reshape(h, 3 * w)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
    return _SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b"")


def decode_rgb8(data: bytes) -> np.ndarray:
    """Inverse of encode_rgb8 for 8-bit RGB, non-interlaced files (all five scanline filters)."""
    if data[:8] != _SIGNATURE:
        raise ValueError("not a PNG file")
    pos, idat, w = 8, b"", None
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if crc != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError("chunk checksum mismatch")
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError("only 8-bit RGB non-interlaced files are supported")
        elif tag == b"IDAT":
            idat += body
        elif tag == b"IEND":
            break
        pos += 12 + n
    if w is None:
        raise ValueError("missing IHDR")
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), dtype=np.uint8)
    prev = np.zeros(3 * w, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(3 * w, dtype=np.int32)
            for i in range(3 * w):
                a = cur[i - 3] if i >= 3 else 0
                b = prev[i]
                c = prev[i - 3] if i >= 3 else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                elif f == 4:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("unknown scanline filter")
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)


def write_rgb8(path: str, rgb: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(encode_rgb8(rgb))
