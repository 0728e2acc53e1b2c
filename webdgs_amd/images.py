"""Ground-truth image ingest, mirroring ``src/utils/load-images.ts:11-56``.

The reference filters a file list to ``.jpg/.jpeg/.png``, orders it with ``localeCompare(numeric, base sensitivity)``
(load-images.ts:12-17), decodes with ``createImageBitmap`` and uploads each into an ``rgba8unorm`` texture
(``createTextureFromImage``, 42-56); a file that fails to decode is logged and dropped (31-34).  Here decoding is
Pillow's (PNG/JPEG, host side) with a dependency-free fallback for 8-bit non-interlaced PNGs, and the "texture" is a
``width*height*4``-byte device buffer -- the layout every kernel of the hot path consumes.
"""
from __future__ import annotations

import dataclasses
import os
import re
import struct
import sys
import zlib
from typing import Optional, Sequence

import numpy as np

_EXTENSIONS = (".jpg", ".jpeg", ".png")


@dataclasses.dataclass
class LoadedImage:
    """``LoadedImage`` (load-images.ts:1-8): ``texture`` is the rgba8 device buffer (``None`` when loaded host-only)."""

    name: str
    file: str
    bitmap: np.ndarray  # [H, W, 4] uint8
    width: int
    height: int
    texture: Optional[object] = None


def naturalKey(name: str):
    """Ordering of ``a.localeCompare(b, undefined, {numeric: true, sensitivity: 'base'})`` for ASCII names: case-insensitive,
    digit runs compared by value."""
    return [(0, int(tok), "") if tok.isdigit() else (1, 0, tok.lower()) for tok in re.split(r"(\d+)", name) if tok != ""]


def _paeth(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def decodePNG(data: bytes) -> np.ndarray:
    """8-bit, non-interlaced PNG (grey, grey+alpha, RGB, RGBA, palette) -> ``[H, W, 4]`` uint8."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, palette, trns = 8, [], None, None
    width = height = depth = ctype = interlace = None
    while pos + 8 <= len(data):
        (length,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if tag == b"IHDR":
            width, height, depth, ctype, _comp, _filt, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    packed = depth in (1, 2, 4) and ctype in (0, 3)  # sub-byte samples exist for grey and palette images only
    if width is None or interlace != 0 or not (depth == 8 or packed):
        raise ValueError("unsupported PNG (need non-interlaced, at most 8 bits per sample)")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = (width * depth + 7) // 8 if packed else width * channels
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(height, stride + 1)
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    for y in range(height):
        ft, line = int(raw[y, 0]), raw[y, 1:].copy()
        if ft == 1 or ft == 3 or ft == 4:  # Sub / Average / Paeth depend on the already-decoded left neighbour: per-pixel walk
            cur = np.zeros(stride + channels, np.uint8)
            up = np.concatenate([np.zeros(channels, np.uint8), prev])
            for x in range(0, stride, channels):
                left, above, upleft = cur[x:x + channels], up[x + channels:x + 2 * channels], up[x:x + channels]
                if ft == 1:
                    pred = left
                elif ft == 3:
                    pred = ((left.astype(np.uint16) + above) >> 1).astype(np.uint8)
                else:
                    pred = _paeth(left, above, upleft)
                cur[x + channels:x + 2 * channels] = line[x:x + channels] + pred
            line = cur[channels:]
        elif ft == 2:
            line = line + prev
        elif ft != 0:
            raise ValueError(f"bad PNG filter {ft}")
        out[y] = line
        prev = line
    if packed:  # samples are packed MSB first; grey levels scale to 0..255
        bits = np.unpackbits(out, axis=1)[:, :width * depth].reshape(height, width, depth)
        vals = (bits * (1 << np.arange(depth - 1, -1, -1, dtype=np.uint8))).sum(axis=2).astype(np.uint8)
        if ctype == 0:
            vals = (vals.astype(np.uint16) * 255 // ((1 << depth) - 1)).astype(np.uint8)
        out = vals
    px = out.reshape(height, width, channels)
    rgba = np.full((height, width, 4), 255, np.uint8)
    if ctype == 0:
        rgba[..., :3] = px
    elif ctype == 2:
        rgba[..., :3] = px
    elif ctype == 3:
        if palette is None:
            raise ValueError("palette PNG without PLTE")
        rgba[..., :3] = palette[px[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, np.uint8)
            alpha[:len(trns)] = trns
            rgba[..., 3] = alpha[px[..., 0]]
    elif ctype == 4:
        rgba[..., :3] = px[..., :1]
        rgba[..., 3] = px[..., 1]
    else:
        rgba[...] = px
    return rgba


def decodeImage(data: bytes, name: str = "") -> np.ndarray:
    """PNG or JPEG bytes -> ``[H, W, 4]`` uint8 RGBA (opaque alpha when the file has none), like an ImageBitmap upload."""
    try:
        import io

        from PIL import Image
    except ImportError:
        if data[:8] == b"\x89PNG\r\n\x1a\n":
            return decodePNG(data)
        raise RuntimeError(f"{name or 'image'}: JPEG decoding needs Pillow, which is not installed")
    with Image.open(io.BytesIO(data)) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGBA"), np.uint8))


def createTextureFromImage(device, image: np.ndarray):
    """``createTextureFromImage`` (load-images.ts:42-56): rgba8 rows, top to bottom, into a device buffer."""
    a = np.ascontiguousarray(image, np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("createTextureFromImage expects [H, W, 4] uint8")
    return device.bufferFrom(a, "gt image")


def loadImages(files: Sequence[str], device=None) -> list:
    """``loadImages(files, device)``: ``files`` are paths; ``device=None`` keeps the images on the host only."""
    names = [f for f in files if os.path.basename(f).lower().endswith(_EXTENSIONS)]
    names.sort(key=lambda f: naturalKey(os.path.basename(f)))
    out = []
    for path in names:
        try:
            with open(path, "rb") as fh:
                bitmap = decodeImage(fh.read(), os.path.basename(path))
            tex = createTextureFromImage(device, bitmap) if device is not None else None
            out.append(LoadedImage(os.path.basename(path), path, bitmap, bitmap.shape[1], bitmap.shape[0], tex))
        except Exception as e:  # load-images.ts:31-34: log and drop
            print(f"Failed to load image {os.path.basename(path)}: {e}", file=sys.stderr)
    return out
