"""TEST INFRASTRUCTURE: a small FLAC *writer* (published format, https://xiph.org/flac/format.html) that exercises every decoder
path of ps_slm_amd/csrc/flac.hip -- CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC subframes, wasted bits, Rice and Rice2 residual
coding with partitions and escape partitions, independent / left-side / right-side / mid-side stereo, 8/16-bit block size fields,
frame CRC-8 / CRC-16 and the STREAMINFO MD5 of the PCM.  No FLAC encoder, decoder or file exists on the image, so the decoder is
tested by round trip against this writer (parity unpinned against libFLAC, stated in DESIGN.md)."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        v &= (1 << n) - 1
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def unary(self, q):
        self.bits.extend([0] * q + [1])

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def tobytes(self):
        assert len(self.bits) % 8 == 0
        a = np.array(self.bits, dtype=np.uint8).reshape(-1, 8)
        return bytes(np.packbits(a, axis=1).reshape(-1))


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(n):
    if n < 0x80:
        return bytes([n])
    out, first_bits = [], 6
    while n >= (1 << first_bits):
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        first_bits -= 1
    lead = (0xFF << (first_bits + 1)) & 0xFF
    return bytes([lead | n] + out[::-1])


def write_residual(bw, res, blocksize, order, rng, method, porder, escape_first):
    bw.put(method, 2)
    bw.put(porder, 4)
    pbits = 4 if method == 0 else 5
    idx = 0
    for p in range(1 << porder):
        cnt = (blocksize >> porder) - (order if p == 0 else 0)
        part = res[idx: idx + cnt]
        idx += cnt
        if escape_first and p == 0:
            nb = max(1, int(max(abs(int(x)) for x in part) if cnt else 0).bit_length() + 1)
            bw.put((1 << pbits) - 1, pbits)
            bw.put(nb, 5)
            for x in part:
                bw.put(int(x), nb)
            continue
        mean = float(np.mean(np.abs(part))) if cnt else 0.0
        k = min(max(0, int(np.log2(mean + 1))), (1 << pbits) - 2)
        bw.put(k, pbits)
        for x in part:
            x = int(x)
            u = (x << 1) if x >= 0 else ((-x) << 1) - 1
            bw.unary(u >> k)
            if k:
                bw.put(u & ((1 << k) - 1), k)
    assert idx == len(res)


def write_subframe(bw, x, bps, kind, rng, method=0, porder=0, escape_first=False, wasted=0):
    """x: int64 samples of one channel (after stereo decorrelation).  kind: 'constant' | 'verbatim' | ('fixed', order) | ('lpc', order)."""
    n = len(x)
    if wasted:
        assert all(int(v) % (1 << wasted) == 0 for v in x)
        x = x >> wasted
        bps -= wasted
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0, 6)
    elif kind == "verbatim":
        bw.put(1, 6)
    elif kind[0] == "fixed":
        bw.put(8 + kind[1], 6)
    else:
        bw.put(31 + kind[1], 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        bw.put(int(x[0]), bps)
    elif kind == "verbatim":
        for v in x:
            bw.put(int(v), bps)
    elif kind[0] == "fixed":
        order = kind[1]
        for v in x[:order]:
            bw.put(int(v), bps)
        coef = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        res = [int(x[i]) - sum(c * int(x[i - 1 - j]) for j, c in enumerate(coef)) for i in range(order, n)]
        write_residual(bw, res, n, order, rng, method, porder, escape_first)
    else:
        order = kind[1]
        prec, shift = 12, 9
        coef = [int(c) for c in rng.integers(-(1 << (prec - 3)), 1 << (prec - 3), order)]
        for v in x[:order]:
            bw.put(int(v), bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for c in coef:
            bw.put(c, prec)
        res = [int(x[i]) - (sum(c * int(x[i - 1 - j]) for j, c in enumerate(coef)) >> shift) for i in range(order, n)]
        write_residual(bw, res, n, order, rng, method, porder, escape_first)


def write_flac(samples, rate=16000, bps=16, blocksize=1024, seed=0, with_md5=True):
    """samples: int array [T] or [T, 2].  Returns the FLAC byte stream; frames rotate through the subframe kinds, residual methods,
    partition orders and (stereo) channel assignments."""
    rng = np.random.default_rng(seed)
    x = np.asarray(samples, dtype=np.int64)
    if x.ndim == 1:
        x = x[:, None]
    T, C = x.shape
    kinds = ["verbatim", ("fixed", 0), ("fixed", 1), ("fixed", 2), ("fixed", 3), ("fixed", 4), ("lpc", 1), ("lpc", 4), ("lpc", 8),
             ("lpc", 12), "constant"]
    frames = []
    for fno, start in enumerate(range(0, T, blocksize)):
        blk = x[start: start + blocksize]
        n = len(blk)
        hdr = BitWriter()
        hdr.put(0x3FFE, 14)
        hdr.put(0, 1)
        hdr.put(0, 1)                                   # fixed block size stream: frame number coded
        if n == blocksize and blocksize in (192, 576, 1152, 2304, 4608, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
            code = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5}.get(n) or (8 + int(np.log2(n // 256)))
            tail_bits = None
        else:
            code, tail_bits = (6, 8) if n <= 256 else (7, 16)
        hdr.put(code, 4)
        hdr.put({16000: 5, 8000: 4, 44100: 9, 48000: 10}.get(rate, 0), 4)
        ca = 0 if C == 1 else [1, 8, 9, 10][fno % 4]
        hdr.put(ca, 4)
        hdr.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps] if fno % 2 else 0, 3)     # alternately "from STREAMINFO"
        hdr.put(0, 1)
        hb = hdr.tobytes() + utf8_number(fno)
        if tail_bits:
            hb += (n - 1).to_bytes(tail_bits // 8, "big")
        hb += bytes([crc8(hb)])
        body = BitWriter()
        chans = [blk[:, c].copy() for c in range(C)]
        widths = [bps] * C
        if ca == 8:
            chans, widths = [chans[0], chans[0] - chans[1]], [bps, bps + 1]
        elif ca == 9:
            chans, widths = [chans[0] - chans[1], chans[1]], [bps + 1, bps]
        elif ca == 10:
            chans, widths = [(chans[0] + chans[1]) >> 1, chans[0] - chans[1]], [bps, bps + 1]
        for c, (ch, w) in enumerate(zip(chans, widths)):
            kind = kinds[(fno + 3 * c) % len(kinds)]
            wasted = 0
            if kind == "constant":
                if not (ch == ch[0]).all():
                    kind = ("fixed", 2)
            order = kind[1] if isinstance(kind, tuple) else 0
            if order > n:
                kind = "verbatim"
            porder = [0, 1, 2, 3][fno % 4]
            while porder and ((n >> porder) << porder != n or (n >> porder) <= order):
                porder -= 1
            if fno % 7 == 3 and (ch % 4 == 0).all() and w > 3:
                wasted = 2
            write_subframe(body, ch, w, kind, rng, method=fno % 2, porder=porder, escape_first=(fno % 5 == 4), wasted=wasted)
        body.align()
        frame = hb + body.tobytes()
        frames.append(frame + crc16(frame).to_bytes(2, "big"))
    nb = (bps + 7) // 8
    pcm = b"".join(int(v).to_bytes(nb, "little", signed=True) for v in x.reshape(-1))
    md5 = hashlib.md5(pcm).digest() if with_md5 else bytes(16)
    si = BitWriter()
    si.put(blocksize, 16)
    si.put(blocksize, 16)
    si.put(0, 24)
    si.put(0, 24)
    si.put(rate, 20)
    si.put(C - 1, 3)
    si.put(bps - 1, 5)
    si.put(T, 36)
    stream = b"fLaC" + bytes([0x00, 0, 0, 34]) + si.tobytes() + md5
    # a PADDING block after STREAMINFO (the decoder must walk the metadata chain), last-block flag set
    stream += bytes([0x81, 0, 0, 6]) + bytes(6)
    return stream + b"".join(frames)
