"""On-disk formats next to the path: PFM depth maps as eval.py:176-180 writes them
(datasets/depth_utils.py:5-69).  Host-side byte work; PNG/GIF (imageio, cv2) stay out of scope."""
from __future__ import annotations

import sys

import numpy as np


def save_pfm(filename, image, scale=1):
    """depth_utils.py:43-69: float32 (H,W), (H,W,1) or (H,W,3); rows stored bottom-up; the sign of the scale line
    carries the byte order (negative = little-endian)."""
    image = np.asarray(image)
    if image.dtype != np.float32:
        raise Exception('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        tag = b'PF\n'
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        tag = b'Pf\n'
    else:
        raise Exception('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    order = image.dtype.byteorder
    little = order == '<' or (order == '=' and sys.byteorder == 'little')
    header = tag + f'{image.shape[1]} {image.shape[0]}\n'.encode() + (b'%f\n' % (-scale if little else scale))
    with open(filename, 'wb') as f:
        f.write(header)
        f.write(np.ascontiguousarray(image[::-1]).tobytes())


def read_pfm(filename):
    """depth_utils.py:5-40 -> (data (H,W[,3]) float32 top-down, scale)."""
    with open(filename, 'rb') as f:
        tag = f.readline().decode('utf-8').rstrip()
        if tag not in ('PF', 'Pf'):
            raise Exception('Not a PFM file.')
        dims = f.readline().decode('utf-8').split()
        if len(dims) != 2 or not all(d.isdigit() for d in dims):
            raise Exception('Malformed PFM header.')
        width, height = int(dims[0]), int(dims[1])
        scale = float(f.readline().rstrip())
        endian = '<' if scale < 0 else '>'
        data = np.frombuffer(f.read(), dtype=endian + 'f4')
    shape = (height, width, 3) if tag == 'PF' else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


# ---------------------------------------------------------------------------------------------------------
# PNG / GIF as eval.py:185-194 writes them through imageio (`imwrite(f'{i:03d}.png', uint8 (H,W,3))`,
# `mimsave(f'{scene}.gif', frames, fps=30)`).  imageio / PIL are absent here, so the files are produced by the formats'
# published algorithms with the standard library only (PNG: RFC 2083, zlib deflate + CRC-32; GIF89a: LZW).  The pixels
# -- not the bytes -- are what a decoder must get back: an encoder is free in its compression choices.
# ---------------------------------------------------------------------------------------------------------
def to_uint8(img):
    """eval.py:185 `(img_pred*255).astype(np.uint8)` (truncation, values expected in [0, 1])."""
    return (np.asarray(img) * 255).astype(np.uint8)


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def imwrite_png(filename, image):
    """8-bit PNG of a uint8 (H,W) / (H,W,1) grey, (H,W,3) RGB or (H,W,4) RGBA array (filter type 0 on every row)."""
    import struct
    import zlib
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise TypeError("imwrite_png expects uint8 (use to_uint8 for eval.py's conversion)")
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError("image must be (H,W), (H,W,1), (H,W,3) or (H,W,4)")
    h, w, c = a.shape
    color_type = {1: 0, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], 1).tobytes()
    with open(filename, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)))
        f.write(_png_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_png_chunk(b"IEND", b""))


def imread_png(filename):
    """Decoder for the files imwrite_png writes (8-bit, non-interlaced, filter 0..4) -> uint8 (H,W[,C])."""
    import struct
    import zlib
    data = open(filename, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError("PNG chunk CRC mismatch")
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace or ctype not in (0, 2, 6):
        raise NotImplementedError("only 8-bit non-interlaced grey / RGB / RGBA")
    c = {0: 1, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.uint8)
    for y in range(h):
        ft, row = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(w * c, np.int32)
        if ft == 0:
            rec = row
        elif ft == 2:
            rec = (row + prev) & 255
        else:                                               # sub / average / paeth need the left neighbour: serial
            rec = np.zeros(w * c, np.int32)
            for x in range(w * c):
                a_ = rec[x - c] if x >= c else 0
                b_, c_ = prev[x], (prev[x - c] if x >= c else 0)
                if ft == 1:
                    pr = a_
                elif ft == 3:
                    pr = (a_ + b_) // 2
                else:
                    p = a_ + b_ - c_
                    pa, pb, pc = abs(p - a_), abs(p - b_), abs(p - c_)
                    pr = a_ if (pa <= pb and pa <= pc) else (b_ if pb <= pc else c_)
                rec[x] = (row[x] + pr) & 255
        out[y] = rec
    out = out.reshape(h, w, c)
    return out[:, :, 0] if c == 1 else out


def _gif_lzw(indices: np.ndarray, min_code: int = 8) -> bytes:
    """Variable-width LZW of GIF89a (codes LSB-first, dictionary reset at 4096 entries)."""
    clear, end = 1 << min_code, (1 << min_code) + 1
    out, acc, nbits = bytearray(), 0, 0

    def emit(code, width):
        nonlocal acc, nbits
        acc |= code << nbits
        nbits += width
        while nbits >= 8:
            out.append(acc & 255)
            acc >>= 8
            nbits -= 8
    table, nxt, width = {}, end + 1, min_code + 1
    emit(clear, width)
    data = indices.tolist()
    cur = data[0]
    for k in data[1:]:
        key = (cur, k)
        code = table.get(key)
        if code is not None:
            cur = code
            continue
        emit(cur, width)
        if nxt < 4096:
            table[key] = nxt
            if nxt == (1 << width) and width < 12:
                width += 1
            nxt += 1
        else:
            emit(clear, width)
            table, nxt, width = {}, end + 1, min_code + 1
        cur = k
    emit(cur, width)
    emit(end, width)
    if nbits:
        out.append(acc & 255)
    return bytes(out)


def _palette_332(frame: np.ndarray) -> np.ndarray:
    """uint8 (H,W,3) -> indices into the fixed 3-3-2 bit RGB palette (no dithering)."""
    f = frame.astype(np.uint16)
    return ((f[..., 0] >> 5) << 5 | (f[..., 1] >> 5) << 2 | (f[..., 2] >> 6)).astype(np.uint8)


PALETTE_332 = np.array([[(i >> 5) * 255 // 7, ((i >> 2) & 7) * 255 // 7, (i & 3) * 255 // 3] for i in range(256)], np.uint8)


def mimsave_gif(filename, frames, fps=30):
    """Animated GIF89a of uint8 (H,W,3) frames (eval.py:194 `imageio.mimsave(..., imgs, fps=30)`): global 3-3-2 palette,
    frame delay round(100/fps) hundredths of a second, infinite loop."""
    import struct
    frames = [np.asarray(f) for f in frames]
    if not frames or any(f.dtype != np.uint8 or f.ndim != 3 or f.shape != frames[0].shape or f.shape[2] != 3 for f in frames):
        raise ValueError("frames must be a non-empty list of uint8 (H,W,3) arrays of one size")
    h, w, _ = frames[0].shape
    delay = max(1, int(round(100.0 / fps)))
    with open(filename, "wb") as f:
        f.write(b"GIF89a" + struct.pack("<HHBBB", w, h, 0xF7, 0, 0) + PALETTE_332.tobytes())
        f.write(b"\x21\xFF\x0BNETSCAPE2.0\x03\x01\x00\x00\x00")                       # loop forever
        for fr in frames:
            f.write(b"\x21\xF9\x04\x00" + struct.pack("<H", delay) + b"\x00\x00")      # graphic control extension
            f.write(b"\x2C" + struct.pack("<HHHHB", 0, 0, w, h, 0) + b"\x08")
            lzw = _gif_lzw(_palette_332(fr).reshape(-1))
            for i in range(0, len(lzw), 255):
                blk = lzw[i:i + 255]
                f.write(bytes([len(blk)]) + blk)
            f.write(b"\x00")
        f.write(b"\x3B")


def mimread_gif(filename):
    """Decoder for the files mimsave_gif writes -> (list of uint8 (H,W,3) frames, delay in 1/100 s)."""
    import struct
    d = open(filename, "rb").read()
    if d[:6] != b"GIF89a":
        raise ValueError("not a GIF89a file")
    w, h, flags = struct.unpack("<HHB", d[6:11])
    pos = 13
    pal = np.frombuffer(d[pos:pos + 3 * (2 << (flags & 7))], np.uint8).reshape(-1, 3)
    pos += 3 * (2 << (flags & 7))
    frames, delay = [], None
    while d[pos] != 0x3B:
        if d[pos] == 0x21:
            if d[pos + 1] == 0xF9:
                delay = struct.unpack("<H", d[pos + 4:pos + 6])[0]
            pos += 2
            while d[pos]:
                pos += 1 + d[pos]
            pos += 1
            continue
        assert d[pos] == 0x2C
        fw, fh = struct.unpack("<HH", d[pos + 5:pos + 9])
        min_code = d[pos + 10]
        pos += 11
        buf = bytearray()
        while d[pos]:
            buf += d[pos + 1:pos + 1 + d[pos]]
            pos += 1 + d[pos]
        pos += 1
        clear, end = 1 << min_code, (1 << min_code) + 1
        table = {i: [i] for i in range(clear)}
        nxt, width, acc, nbits, out, prev = end + 1, min_code + 1, 0, 0, [], None
        for byte in buf:
            acc |= byte << nbits
            nbits += 8
            while nbits >= width:
                code = acc & ((1 << width) - 1)
                acc >>= width
                nbits -= width
                if code == clear:
                    table = {i: [i] for i in range(clear)}
                    nxt, width, prev = end + 1, min_code + 1, None
                    continue
                if code == end:
                    break
                entry = table[code] if code in table else prev + [prev[0]]
                out += entry
                if prev is not None and nxt < 4096:
                    table[nxt] = prev + [entry[0]]
                    nxt += 1
                    if nxt == (1 << width) and width < 12:
                        width += 1
                prev = entry
        frames.append(pal[np.array(out[:fw * fh], np.int64)].reshape(fh, fw, 3))
    return frames, delay
