"""Minimal OpenEXR reader for the SceneEgo demo depth maps: single-part scanline files, HALF/FLOAT/UINT channels,
compression NONE, ZIPS/ZIP or PIZ.

The reference reads its depth maps with ``cv2.imread(path, IMREAD_ANYCOLOR | IMREAD_ANYDEPTH)``
(``dataset/demo_dataset.py:84``); OpenCV/OpenEXR are not available in this image, so the file format is decoded here
from the published OpenEXR specification ("OpenEXR File Layout", and the PIZ scheme of ImfPizCompressor / ImfHuf /
ImfWav: bitmap -> LUT, canonical Huffman with run-length symbol, 2-D Haar-like wavelet in 14- or 16-bit mode).
The demo files are 640x512, one HALF channel ``Y``, PIZ.  **Parity unpinned**: no OpenEXR/OpenCV decoder can be run in
this image to compare against.  ``tests/test_host_logic.py`` decodes the reference's own demo depth map
(``tests/golden/demo/img_001000.jpg.exr``, a data file) and checks what a wrong Huffman/wavelet/LUT stage could not
produce by accident: every Huffman chunk decodes to exactly its symbol count, all values are finite metric depths in
[0, 10.2] m, and the map is an exact 4x nearest up-sampling along y (rows 4k..4k+3 identical across all 16 independent
chunks); the NONE/ZIP paths are round-tripped against a writer in the test.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_PIXEL_SIZE = {0: 4, 1: 2, 2: 4}          # UINT, HALF, FLOAT
_PIXEL_DTYPE = {0: np.uint32, 1: np.float16, 2: np.float32}
_LINES_PER_CHUNK = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}   # NONE, RLE, ZIPS, ZIP, PIZ

HUF_ENCBITS = 16
HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1
SHORT_ZEROCODE_RUN = 59
LONG_ZEROCODE_RUN = 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN
BITMAP_SIZE = 8192


# ------------------------------------------------------------------------------------------------
# header
# ------------------------------------------------------------------------------------------------
def _parse_header(buf: bytes):
    if buf[:4] != b"\x76\x2f\x31\x01":
        raise ValueError("not an OpenEXR file")
    version = struct.unpack_from("<I", buf, 4)[0]
    if version & 0x200:
        raise NotImplementedError("tiled EXR files are not supported")
    if version & 0x1000:
        raise NotImplementedError("multi-part EXR files are not supported")
    pos = 8
    attrs = {}
    while True:
        end = buf.index(b"\0", pos)
        name = buf[pos:end].decode()
        pos = end + 1
        if name == "":
            break
        end = buf.index(b"\0", pos)
        typ = buf[pos:end].decode()
        pos = end + 1
        size = struct.unpack_from("<i", buf, pos)[0]
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    channels = []
    raw = attrs["channels"][1]
    p = 0
    while raw[p] != 0:
        end = raw.index(b"\0", p)
        cname = raw[p:end].decode()
        p = end + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", raw, p)
        p += 16
        channels.append((cname, ptype, xs, ys))
    channels.sort(key=lambda c: c[0])                     # channels are stored in alphabetical order
    xmin, ymin, xmax, ymax = struct.unpack("<iiii", attrs["dataWindow"][1])
    compression = attrs["compression"][1][0]
    line_order = attrs["lineOrder"][1][0]
    return dict(channels=channels, window=(xmin, ymin, xmax, ymax), compression=compression, line_order=line_order,
                data_start=pos)


# ------------------------------------------------------------------------------------------------
# PIZ: Huffman
# ------------------------------------------------------------------------------------------------
class _BitReader:
    __slots__ = ("data", "pos", "c", "lc")

    def __init__(self, data, pos):
        self.data, self.pos, self.c, self.lc = data, pos, 0, 0

    def get(self, nbits):
        while self.lc < nbits:
            self.c = ((self.c << 8) | self.data[self.pos]) & 0xFFFFFFFFFFFFFFFF
            self.pos += 1
            self.lc += 8
        self.lc -= nbits
        return (self.c >> self.lc) & ((1 << nbits) - 1)


def _huf_unpack_enc_table(data, pos, im, iM):
    hcode = np.zeros(HUF_ENCSIZE, dtype=np.int64)
    br = _BitReader(data, pos)
    i = im
    while i <= iM:
        l = br.get(6)
        hcode[i] = l
        if l == LONG_ZEROCODE_RUN:
            zerun = br.get(8) + SHORTEST_LONG_RUN
            hcode[i:i + zerun] = 0
            i += zerun
            continue
        if l >= SHORT_ZEROCODE_RUN:
            zerun = l - SHORT_ZEROCODE_RUN + 2
            hcode[i:i + zerun] = 0
            i += zerun
            continue
        i += 1
    return hcode, br.pos


def _huf_canonical_code_table(lengths):
    """Code lengths -> canonical codes (OpenEXR convention: codes of length l are consecutive, shorter codes are the
    numerically larger prefixes)."""
    n = np.bincount(lengths, minlength=59).astype(np.int64)
    c = 0
    start = np.zeros(59, dtype=np.int64)
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        start[i] = c
        c = nc
    codes = {}
    nxt = start.copy()
    for sym in np.nonzero(lengths)[0]:
        l = int(lengths[sym])
        codes[(l, int(nxt[l]))] = int(sym)
        nxt[l] += 1
    return codes


def _huf_uncompress(data: bytes, n_out: int) -> np.ndarray:
    if len(data) == 0:
        return np.zeros(n_out, dtype=np.uint16)
    im, iM, _table_len, nbits = struct.unpack_from("<IIII", data, 0)
    lengths, pos = _huf_unpack_enc_table(data, 20, im, iM)
    codes = _huf_canonical_code_table(lengths)
    rlc = iM
    min_len = min(l for l, _ in codes)
    out = np.zeros(n_out, dtype=np.uint16)
    o = 0
    # bit-serial decode (the demo maps are 327 680 symbols: ~1 s); MSB first
    total_bits = nbits
    byte_arr = data
    bitpos = pos * 8
    end_bit = bitpos + total_bits
    get = codes.get
    while bitpos < end_bit and o < n_out:
        code = 0
        l = 0
        sym = None
        while True:
            code = (code << 1) | ((byte_arr[bitpos >> 3] >> (7 - (bitpos & 7))) & 1)
            bitpos += 1
            l += 1
            if l >= min_len:
                sym = get((l, code))
                if sym is not None:
                    break
            if l > 58:
                raise ValueError("corrupt Huffman stream")
        if sym == rlc:
            cs = 0
            for _ in range(8):
                cs = (cs << 1) | ((byte_arr[bitpos >> 3] >> (7 - (bitpos & 7))) & 1)
                bitpos += 1
            out[o:o + cs] = out[o - 1]
            o += cs
        else:
            out[o] = sym
            o += 1
    if o != n_out:
        raise ValueError(f"Huffman stream decoded {o} of {n_out} values")
    return out


# ------------------------------------------------------------------------------------------------
# PIZ: wavelet
# ------------------------------------------------------------------------------------------------
def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    a = ai.astype(np.int16)
    b = (ai - hs).astype(np.int16)
    return a.view(np.uint16), b.view(np.uint16)


def _wdec16(l, h):
    m = l.astype(np.int32)
    d = h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - (1 << 15)) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(img: np.ndarray, mx: int) -> None:
    """In-place inverse wavelet of a [ny, nx] uint16 image (ImfWav wav2Decode, ox = 1, oy = nx)."""
    ny, nx = img.shape
    wdec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.arange(0)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.arange(0)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            px, p01, p10, p11 = img[Y, X], img[Y, X + p], img[Y + p, X], img[Y + p, X + p]
            i00, i10 = wdec(px, p10)
            i01, i11 = wdec(p01, p11)
            a, b = wdec(i00, i01)
            img[Y, X], img[Y, X + p] = a, b
            a, b = wdec(i10, i11)
            img[Y + p, X], img[Y + p, X + p] = a, b
        if nx & p and len(ys):            # odd column left over at this level
            x = len(xs) * p2
            a, b = wdec(img[ys, x], img[ys + p, x])
            img[ys, x], img[ys + p, x] = a, b
        if ny & p and len(xs):            # odd row left over at this level
            y = len(ys) * p2
            a, b = wdec(img[y, xs], img[y, xs + p])
            img[y, xs], img[y, xs + p] = a, b
        p2 = p
        p >>= 1


def _piz_decompress(block: bytes, channels, nx: int, ny: int) -> bytes:
    """One PIZ chunk -> raw scanline data (per line: channel after channel)."""
    sizes = [_PIXEL_SIZE[c[1]] // 2 for c in channels]            # 16-bit words per pixel
    n_words = sum(s * nx * ny for s in sizes)
    min_nz, max_nz = struct.unpack_from("<HH", block, 0)
    pos = 4
    bitmap = np.zeros(BITMAP_SIZE, dtype=np.uint8)
    if min_nz <= max_nz:
        cnt = max_nz - min_nz + 1
        bitmap[min_nz:min_nz + cnt] = np.frombuffer(block, dtype=np.uint8, count=cnt, offset=pos)
        pos += cnt
    bits = np.unpackbits(bitmap, bitorder="little").astype(bool)
    bits[0] = True                                                  # value 0 is always present
    lut = np.zeros(65536, dtype=np.uint16)
    present = np.nonzero(bits)[0].astype(np.uint16)
    lut[:len(present)] = present
    max_value = len(present) - 1
    (length,) = struct.unpack_from("<i", block, pos)
    pos += 4
    words = _huf_uncompress(block[pos:pos + length], n_words)
    off = 0
    planes = []
    for s in sizes:
        cnt = s * nx * ny
        plane = words[off:off + cnt].reshape(ny, nx * s).copy()
        off += cnt
        if s == 1:
            _wav2_decode(plane, max_value)
        else:                                   # 32-bit channels: the two 16-bit halves are transformed independently
            for j in range(s):
                sub = np.ascontiguousarray(plane[:, j::s])
                _wav2_decode(sub, max_value)
                plane[:, j::s] = sub
        planes.append(lut[plane])
    out = bytearray()
    for y in range(ny):
        for plane in planes:
            out += plane[y].astype("<u2").tobytes()
    return bytes(out)


def _zip_decompress(block: bytes) -> bytes:
    raw = np.frombuffer(zlib.decompress(block), dtype=np.uint8).astype(np.int32)
    # predictor
    d = raw.copy()
    d[1:] = raw[1:] - 128
    d = np.cumsum(d) & 0xFF
    d = d.astype(np.uint8)
    # de-interleave
    half = (len(d) + 1) // 2
    out = np.empty(len(d), dtype=np.uint8)
    out[0::2] = d[:half]
    out[1::2] = d[half:]
    return out.tobytes()


# ------------------------------------------------------------------------------------------------
def read_exr(path: str) -> dict:
    """Returns {channel name: float32 [H, W]} (UINT channels as float32 too)."""
    with open(path, "rb") as f:
        buf = f.read()
    hdr = _parse_header(buf)
    xmin, ymin, xmax, ymax = hdr["window"]
    W, H = xmax - xmin + 1, ymax - ymin + 1
    comp = hdr["compression"]
    if comp not in _LINES_PER_CHUNK:
        raise NotImplementedError(f"EXR compression {comp} is not supported (NONE, ZIPS, ZIP, PIZ are)")
    if any(c[2] != 1 or c[3] != 1 for c in hdr["channels"]):
        raise NotImplementedError("sub-sampled EXR channels are not supported")
    lpc = _LINES_PER_CHUNK[comp]
    n_chunks = (H + lpc - 1) // lpc
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, hdr["data_start"])
    bytes_per_line = sum(_PIXEL_SIZE[c[1]] for c in hdr["channels"]) * W
    planes = {c[0]: np.zeros((H, W), dtype=np.float32) for c in hdr["channels"]}
    for off in offsets:
        y0, size = struct.unpack_from("<ii", buf, off)
        block = buf[off + 8:off + 8 + size]
        ny = min(lpc, ymax - y0 + 1)
        expect = bytes_per_line * ny
        if comp == 0 or size == expect:          # stored uncompressed when compression did not help
            raw = block
        elif comp in (2, 3):
            raw = _zip_decompress(block)
        elif comp == 4:
            raw = _piz_decompress(block, hdr["channels"], W, ny)
        else:
            raise NotImplementedError("RLE")
        p = 0
        for r in range(ny):
            for name, ptype, _, _ in hdr["channels"]:
                cnt = W * _PIXEL_SIZE[ptype]
                row = np.frombuffer(raw, dtype=np.dtype(_PIXEL_DTYPE[ptype]).newbyteorder("<"), count=W, offset=p)
                planes[name][y0 - ymin + r] = row.astype(np.float32)
                p += cnt
    return planes


def read_depth_exr(path: str) -> np.ndarray:
    """First channel (alphabetical: 'Y' for the demo files; 'B' for BGR files, matching cv2's ``[:, :, 0]``)."""
    planes = read_exr(path)
    for pref in ("Y", "B", "Z", "R"):
        if pref in planes:
            return planes[pref]
    return planes[sorted(planes.keys())[0]]
