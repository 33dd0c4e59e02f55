"""Test infrastructure: a small baseline JPEG *encoder* for the stream shapes Pillow cannot write (sampling factors other than
4:4:4 / 4:2:2 / 4:2:0, non-interleaved scans, Huffman codes longer than the decoder's 9-bit look-up, 16-bit quantisation
tables, Adobe RGB).  Its output is decoded by libjpeg-turbo (through Pillow) for the expected pixels and by the library's
csrc/jpeg_decode.cpp for the test; how good the compression is does not matter.  Not part of the product."""
import numpy as np
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
AC_SYMBOLS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
DC_SYMBOLS = list(range(12))


def flat_table(symbols, length):
    """every symbol gets a code of `length` bits (canonical order): (bits[1..16], vals, {symbol: (code, length)})"""
    assert len(symbols) < (1 << length)
    bits = [0] * 17
    bits[length] = len(symbols)
    return bits[1:], list(symbols), {s: (i, length) for i, s in enumerate(symbols)}


def staircase_table(symbols):
    """a code with one symbol per length 2..15 and the rest at 16 bits: exercises the code lengths past the 9-bit look-up"""
    bits = [0] * 17
    codes = {}
    code = 0
    k = 0
    for length in range(1, 17):
        n = 0 if length == 1 else min(1, len(symbols) - k) if length < 16 else len(symbols) - k
        bits[length] = n
        for _ in range(n):
            codes[symbols[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    assert k == len(symbols) and bits[16] < (1 << 16) - 1
    return bits[1:], list(symbols), codes


class BitWriter:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1)); self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1 if self.n else 0

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def size_of(v):
    return int(abs(int(v))).bit_length()


def encode_block(bw, blk, pred, dc, ac):
    d = int(blk[0]) - pred
    s = size_of(d)
    bw.put(*dc[s])
    if s:
        bw.put(d if d >= 0 else d + (1 << s) - 1, s)
    run = 0
    for k in range(1, 64):
        v = int(blk[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*ac[0xF0]); run -= 16
        s = size_of(v)
        bw.put(*ac[(run << 4) | s])
        bw.put(v if v >= 0 else v + (1 << s) - 1, s)
        run = 0
    if run:
        bw.put(*ac[0x00])
    return int(blk[0])


def segment(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def encode(rgb, sampling=((2, 2), (1, 1), (1, 1)), q=8, interleaved=True, restart=0, long_codes=False, q16=False, colour="ycc", jfif=True):
    """rgb: HxWx3 (or HxW for one component) uint8.  sampling: (h, v) per component.  colour: "ycc" | "rgb" (Adobe transform 0)."""
    a = np.asarray(rgb, dtype=np.float64)
    if a.ndim == 2:
        planes = [a]; sampling = sampling[:1]
    elif colour == "ycc":
        r, g, b = a[..., 0], a[..., 1], a[..., 2]
        planes = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
    else:
        planes = [a[..., 0], a[..., 1], a[..., 2]]
    H, W = planes[0].shape
    hmax = max(h for h, _ in sampling); vmax = max(v for _, v in sampling)
    mcux = -(-W // (8 * hmax)); mcuy = -(-H // (8 * vmax))
    qt = np.full(64, q, dtype=np.int64)
    qt[0] = max(1, q // 2)
    if q16:
        qt[63] = 300
    coefs = []
    for p, (h, v) in zip(planes, sampling):
        fh, fv = hmax // h, vmax // v
        ph = np.pad(p, ((0, mcuy * 8 * vmax - H), (0, mcux * 8 * hmax - W)), mode="edge")
        ds = ph.reshape(mcuy * 8 * v, fv, mcux * 8 * h, fh).mean(axis=(1, 3))
        by, bx = mcuy * v, mcux * h
        blocks = ds.reshape(by, 8, bx, 8).transpose(0, 2, 1, 3) - 128.0
        c = dctn(blocks, axes=(2, 3), norm="ortho").reshape(by, bx, 64)
        c = np.rint(c / qt[None, None, :]).astype(np.int64)
        coefs.append(np.take(c, ZIGZAG, axis=2))
    # qt above is in natural order; DQT carries it in zigzag order
    dc_tab = flat_table(DC_SYMBOLS, 4) if not long_codes else staircase_table(DC_SYMBOLS)
    ac_tab = flat_table(AC_SYMBOLS, 8) if not long_codes else staircase_table(AC_SYMBOLS)
    out = bytearray(b"\xff\xd8")
    if jfif and colour == "ycc":
        out += segment(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")
    if colour == "rgb":
        out += segment(0xEE, b"Adobe\0\x64\0\0\0\0\0")                      # transform 0: the components are R, G, B
    qz = [int(qt[ZIGZAG[i]]) for i in range(64)]
    out += segment(0xDB, bytes([0x10]) + b"".join(x.to_bytes(2, "big") for x in qz)) if q16 else segment(0xDB, bytes([0]) + bytes(qz))
    n = len(planes)
    sof = bytes([8]) + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([n])
    for i, (h, v) in enumerate(sampling):
        sof += bytes([i + 1, (h << 4) | v, 0])
    out += segment(0xC0, sof)
    out += segment(0xC4, bytes([0x00]) + bytes(dc_tab[0]) + bytes(dc_tab[1]) + bytes([0x10]) + bytes(ac_tab[0]) + bytes(ac_tab[1]))
    if restart:
        out += segment(0xDD, restart.to_bytes(2, "big"))

    def scan(comps):
        hdr = bytes([len(comps)]) + b"".join(bytes([i + 1, 0x00]) for i in comps) + bytes([0, 63, 0])
        body = bytearray(segment(0xDA, hdr))
        bw = BitWriter(); pred = {i: 0 for i in comps}; count = 0; rst = 0
        if len(comps) > 1:
            units = [[(i, my * sampling[i][1] + v, mx * sampling[i][0] + h) for i in comps for v in range(sampling[i][1]) for h in range(sampling[i][0])]
                     for my in range(mcuy) for mx in range(mcux)]
        else:
            i = comps[0]; h, v = sampling[i]
            cw = -(-(-(-W * h // hmax)) // 8); ch = -(-(-(-H * v // vmax)) // 8)        # blocks that hold real samples
            units = [[(i, y, x)] for y in range(ch) for x in range(cw)]
        for u in units:
            if restart and count == restart:
                bw.flush(); body += bw.out; body += bytes([0xFF, 0xD0 + rst]); rst = (rst + 1) & 7
                bw = BitWriter(); pred = {i: 0 for i in comps}; count = 0
            for (i, y, x) in u:
                pred[i] = encode_block(bw, coefs[i][y, x], pred[i], dc_tab[2], ac_tab[2])
            count += 1
        bw.flush(); body += bw.out
        return body

    if interleaved or n == 1:
        out += scan(list(range(n)))
    else:
        for i in range(n):
            out += scan([i])
    out += b"\xff\xd9"
    return bytes(out)
