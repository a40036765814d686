"""TEST INFRASTRUCTURE ONLY — a second, independent OpenEXR scanline decoder used to pin ``sceneego_amd/exr.py``.

The reference reads its depth maps with ``cv2.imread(path, IMREAD_ANYCOLOR | IMREAD_ANYDEPTH)``
(``/root/reference/dataset/demo_dataset.py:84``); neither OpenCV nor OpenEXR can be run in this image, so the product's
reader (``sceneego_amd/exr.py``) cannot be compared with the library the reference uses.  This module restates the file
format a second time, written separately from the product reader and in the most literal form the format description
allows — bit-at-a-time Huffman decoding against an explicit {(length, code) -> symbol} dictionary, the wavelet as the
published per-level loops on Python integers, byte-wise predictor for ZIP — so that the two decoders share no code and
no optimisation tricks.  Agreement of both on every demo depth map of the reference is the pin
(``tests/test_oracle_golden.py::test_exr_reader_matches_independent_decoder``); parity against OpenEXR itself remains
unpinned and is stated so in DESIGN.md.

Format notes (OpenEXR "File Layout" + the PIZ scheme: ImfPizCompressor / ImfHuf / ImfWav):
  file    = magic 0x01312f76, version word, attributes (name\\0 type\\0 int32 size, bytes) ... \\0, offset table (uint64 per
            chunk), chunks (int32 y, int32 size, bytes)
  chunk   = 1 (NONE, ZIPS), 16 (ZIP) or 32 (PIZ) scanlines; inside a chunk scanlines are stored one after another, each as
            channel after channel (alphabetical), little-endian samples
  ZIP     = zlib stream of: byte de-interleave (even bytes first half, odd bytes second half) then delta predictor
  PIZ     = uint16 min, max of the non-zero bitmap range, bitmap bytes, int32 Huffman length, Huffman block
            (uint32 im, iM, table bits, data bits, reserved; 6-bit code lengths with zero-run codes 59..62 / 63+8 bits;
            canonical codes, MSB first; symbol iM = "repeat the previous symbol <next 8 bits> times"); the decoded uint16s are
            wavelet coefficients per channel (whole channel block: ny x nx x words-per-sample), inverse 2-D wavelet per
            word plane (14-bit or 16-bit flavour by the bitmap population), then the bitmap's rank -> value table.
Only what the demo needs is covered: single-part scanline files, HALF / FLOAT / UINT samples, no sub-sampling.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _header(buf):
    magic, version = struct.unpack_from("<II", buf, 0)
    if magic != 20000630:
        raise ValueError("not an OpenEXR file")
    if version & 0x200 or version & 0x1000 or version & 0x800:
        raise NotImplementedError("only single-part scanline files")
    pos = 8
    attr = {}
    while True:
        name, pos = _cstr(buf, pos)
        if not name:
            break
        typ, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attr[name] = (typ, bytes(buf[pos:pos + size]))
        pos += size
    chans = []
    raw = attr["channels"][1]
    p = 0
    while raw[p] != 0:
        cname, p = _cstr(raw, p)
        ptype, = struct.unpack_from("<i", raw, p)
        xs, ys = struct.unpack_from("<ii", raw, p + 8)
        p += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError("sub-sampled channels")
        chans.append((cname, ptype))
    chans.sort()
    return attr, chans, pos


# ---------------------------------------------------------------------------------------------------------------------
def _unzip_block(data, want):
    raw = bytearray(zlib.decompress(data))
    if len(raw) != want:
        raise ValueError("ZIP block size mismatch")
    for i in range(1, len(raw)):                      # predictor
        raw[i] = (raw[i - 1] + raw[i] - 128) & 0xFF
    half = (len(raw) + 1) // 2
    out = bytearray(len(raw))
    out[0::2] = raw[:half]                            # re-interleave
    out[1::2] = raw[half:]
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------------------
class _Bits:
    """MSB-first bit source over a bytes object."""

    def __init__(self, data, pos):
        self.data, self.bitpos = data, pos * 8

    def take(self, n):
        v = 0
        for _ in range(n):
            byte = self.data[self.bitpos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.bitpos & 7))) & 1)
            self.bitpos += 1
        return v


def _huffman(block, n_out):
    im, iM, _table_bits, n_bits, _ = struct.unpack_from("<IIIII", block, 0)
    src = _Bits(block, 20)
    lengths = {}
    s = im
    while s <= iM:
        l = src.take(6)
        if l == 63:
            s += src.take(8) + 6
        elif l >= 59:
            s += l - 59 + 2
        else:
            if l:
                lengths[s] = l
            s += 1
    # canonical codes: within a length, codes go up with the symbol value; the first code of length l is
    # (first code of length l+1 + number of codes of length l+1) >> 1
    count = [0] * 60
    for l in lengths.values():
        count[l] += 1
    first = [0] * 60
    c = 0
    for l in range(58, 0, -1):
        first[l] = c
        c = (c + count[l]) >> 1
    book = {}
    nxt = list(first)
    for sym in sorted(lengths):
        l = lengths[sym]
        book[(l, nxt[l])] = sym
        nxt[l] += 1
    # the bitstream starts at the next byte boundary after the table
    src = _Bits(block, (src.bitpos + 7) // 8)
    end = src.bitpos + n_bits
    out = []
    while src.bitpos < end and len(out) < n_out:
        l, code = 0, 0
        while True:
            code = (code << 1) | src.take(1)
            l += 1
            if (l, code) in book:
                sym = book[(l, code)]
                break
            if l > 58:
                raise ValueError("bad Huffman code")
        if sym == iM:                                  # run-length symbol
            rep = src.take(8)
            if not out:
                raise ValueError("run before first symbol")
            out.extend([out[-1]] * rep)
        else:
            out.append(sym)
    if len(out) != n_out:
        raise ValueError(f"Huffman block decoded to {len(out)} symbols, expected {n_out}")
    return out


def _s16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def _wdec14(l, h):
    ls, hs = _s16(l), _s16(h)
    ai = ls + (hs & 1) + (hs >> 1)
    return ai & 0xFFFF, (ai - hs) & 0xFFFF


def _wdec16(l, h):
    bb = (l - (h >> 1)) & 0xFFFF
    aa = (h + bb - 0x8000) & 0xFFFF
    return aa, bb


def _wavelet_inverse(buf, base, nx, ox, ny, oy, mx):
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        y = 0
        while y + p < ny:                              # rows that have a partner
            row = base + oy * y
            x = 0
            while x + p < nx:
                a = row + ox * x
                i00, i10 = dec(buf[a], buf[a + oy1])
                i01, i11 = dec(buf[a + ox1], buf[a + oy1 + ox1])
                buf[a], buf[a + ox1] = dec(i00, i01)
                buf[a + oy1], buf[a + oy1 + ox1] = dec(i10, i11)
                x += p2
            if x < nx:                                 # odd column
                a = row + ox * x
                buf[a], buf[a + oy1] = dec(buf[a], buf[a + oy1])
            y += p2
        if y < ny:                                     # odd row
            row = base + oy * y
            x = 0
            while x + p < nx:
                a = row + ox * x
                buf[a], buf[a + ox1] = dec(buf[a], buf[a + ox1])
                x += p2
        p2 = p
        p >>= 1
        _ = (oy2, ox2)


def _unpiz_block(data, chans, nx, ny):
    words = {0: 2, 1: 1, 2: 2}
    total = sum(nx * ny * words[t] for _, t in chans)
    mn, mx = struct.unpack_from("<HH", data, 0)
    pos = 4
    bitmap = bytearray(8192)
    if mn <= mx:
        bitmap[mn:mx + 1] = data[pos:pos + mx - mn + 1]
        pos += mx - mn + 1
    (hlen,) = struct.unpack_from("<i", data, pos)
    pos += 4
    coeff = _huffman(data[pos:pos + hlen], total)
    lut = [v for v in range(65536) if v == 0 or (bitmap[v >> 3] >> (v & 7)) & 1]
    top = len(lut) - 1
    base = 0
    starts = []
    for _, t in chans:
        starts.append(base)
        for j in range(words[t]):
            _wavelet_inverse(coeff, base + j, nx, words[t], ny, nx * words[t], top)
        base += nx * ny * words[t]
    vals = [lut[c] if c <= top else 0 for c in coeff]
    out = bytearray()
    for y in range(ny):                                # scanline-major, channel after channel
        for (_, t), st in zip(chans, starts):
            row = vals[st + y * nx * words[t]: st + (y + 1) * nx * words[t]]
            out += struct.pack(f"<{len(row)}H", *row)
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------------------
def read(path):
    """-> {channel name: 2-D numpy array} of a single-part scanline OpenEXR file (NONE / ZIPS / ZIP / PIZ)."""
    with open(path, "rb") as f:
        buf = f.read()
    attr, chans, pos = _header(buf)
    x0, y0, x1, y1 = struct.unpack("<iiii", attr["dataWindow"][1])
    nx, ny = x1 - x0 + 1, y1 - y0 + 1
    comp = attr["compression"][1][0]
    lines = {0: 1, 2: 1, 3: 16, 4: 32}.get(comp)
    if lines is None:
        raise NotImplementedError(f"compression {comp}")
    nchunks = (ny + lines - 1) // lines
    offsets = struct.unpack_from(f"<{nchunks}Q", buf, pos)
    size = {0: 4, 1: 2, 2: 4}
    dtype = {0: "<u4", 1: "<f2", 2: "<f4"}
    line_bytes = sum(nx * size[t] for _, t in chans)
    planes = {n: np.zeros((ny, nx), dtype=np.dtype(dtype[t])) for n, t in chans}
    for off in offsets:
        y, n = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8: off + 8 + n]
        rows = min(lines, y1 - y + 1)
        want = rows * line_bytes
        if n != want:                                  # a block that did not shrink is stored as is
            if comp in (2, 3):
                data = _unzip_block(data, want)
            elif comp == 4:
                data = _unpiz_block(data, chans, nx, rows)
        p = 0
        for r in range(rows):
            for name, t in chans:
                planes[name][y - y0 + r] = np.frombuffer(data, dtype=dtype[t], count=nx, offset=p)
                p += nx * size[t]
    return planes
