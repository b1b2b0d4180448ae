"""A small FLAC *encoder* for the tests of ``llark_flac_decode_host`` (no FLAC file and no encoder exist in this image): writes
streams that exercise every construct the decoder implements -- STREAMINFO with the MD5 signature, fixed and explicit block sizes,
CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC subframes, wasted bits, Rice partitions of any order with 4- and 5-bit parameters and
escape partitions, independent / left-side / right-side / mid-side stereo, CRC-8 and CRC-16 -- following the published format
(xiph.org FLAC format, RFC 9639).  Test infrastructure only."""
import hashlib
from typing import List, Optional, Sequence

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits: List[int] = []

    def put(self, value: int, n: int) -> None:
        value &= (1 << n) - 1
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def unary(self, zeros: int) -> None:
        self.bits.extend([0] * zeros)
        self.bits.append(1)

    def align(self) -> None:
        while len(self.bits) % 8:
            self.bits.append(0)

    def tobytes(self) -> bytes:
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    out = []
    n = 0
    while True:                                    # bytes of 6 payload bits, then a lead byte with n + 1 ones
        out.append(0x80 | (v & 0x3F))
        v >>= 6
        n += 1
        if v < (1 << (6 - n)):
            break
    lead = ((0xFF << (7 - n)) & 0xFF) | v
    return bytes([lead] + out[::-1])


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _rice_cost(res: np.ndarray, k: int) -> int:
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.int64)
    return int((u >> k).sum() + len(res) * (1 + k))


def write_residual(bw: BitWriter, res: np.ndarray, bs: int, order: int, porder: int, method: int = 0, escape_partition: Optional[int] = None):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    bw.put(method, 2)
    bw.put(porder, 4)
    parts = 1 << porder
    assert bs % parts == 0 and bs // parts >= order
    i = 0
    for pt in range(parts):
        cnt = bs // parts - (order if pt == 0 else 0)
        seg = res[i:i + cnt]
        i += cnt
        if escape_partition == pt:
            raw = max(1, int(max(int(abs(int(v))) for v in seg) if len(seg) else 0).bit_length() + 1)
            bw.put(esc, pbits)
            bw.put(raw, 5)
            for v in seg:
                bw.put(int(v), raw)
            continue
        k = min(range(0, esc), key=lambda kk: _rice_cost(seg, kk)) if len(seg) else 0
        bw.put(k, pbits)
        for v in seg:
            v = int(v)
            u = 2 * v if v >= 0 else -2 * v - 1
            bw.unary(u >> k)
            if k:
                bw.put(u & ((1 << k) - 1), k)


def write_subframe(bw: BitWriter, x: np.ndarray, bps: int, kind: str, order: int = 0, porder: int = 0, method: int = 0,
                   lpc: Optional[Sequence[int]] = None, lpc_prec: int = 12, lpc_shift: int = 0, escape_partition: Optional[int] = None):
    x = x.astype(np.int64)
    wasted = 0
    if kind != "constant" and x.any():
        while not (x & ((1 << (wasted + 1)) - 1)).any() and wasted < bps - 1:
            wasted += 1
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        order = len(lpc)
        code = 31 + order
    bw.put(0, 1)
    bw.put(code, 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
        x = x >> wasted
        bps -= wasted
    else:
        bw.put(0, 1)
    bs = len(x)
    if kind == "constant":
        assert (x == x[0]).all()
        bw.put(int(x[0]), bps)
    elif kind == "verbatim":
        for v in x:
            bw.put(int(v), bps)
    else:
        for v in x[:order]:
            bw.put(int(v), bps)
        coef = list(lpc) if kind == "lpc" else FIXED[order]
        if kind == "lpc":
            bw.put(lpc_prec - 1, 4)
            bw.put(lpc_shift, 5)
            for c in coef:
                bw.put(int(c), lpc_prec)
        while porder and (bs % (1 << porder) or (bs >> porder) < order):       # a short last block: the largest partition order that fits
            porder -= 1
        if escape_partition is not None:
            escape_partition = min(escape_partition, (1 << porder) - 1)
        res = np.zeros(bs - order, dtype=np.int64)
        for t in range(order, bs):
            pred = sum(int(coef[j]) * int(x[t - 1 - j]) for j in range(order))
            res[t - order] = int(x[t]) - (pred >> (lpc_shift if kind == "lpc" else 0))
        write_residual(bw, res, bs, order, porder, method, escape_partition)


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SIZE_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def write_frame(samples: np.ndarray, frame_no: int, sr: int, bps: int, stereo: str, sub, header_from_streaminfo: bool = False) -> bytes:
    """samples [bs][channels] ints; stereo in {"independent", "left_side", "right_side", "mid_side"}; sub(c) -> kwargs of write_subframe."""
    bs, ch = samples.shape
    bw = BitWriter()
    bw.put(0b11111111111110, 14)
    bw.put(0, 1)
    bw.put(0, 1)                                       # fixed block size stream: the coded number is the frame number
    bcode = BLOCK_CODES.get(bs, 6 if bs <= 256 else 7)
    bw.put(bcode, 4)
    rcode = 0 if header_from_streaminfo else RATE_CODES.get(sr, 13 if sr < 65536 else 12)
    bw.put(rcode, 4)
    bw.put({"independent": ch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[stereo], 4)
    bw.put(0 if header_from_streaminfo else SIZE_CODES[bps], 3)
    bw.put(0, 1)
    for b in utf8_number(frame_no):
        bw.put(b, 8)
    if bcode == 6:
        bw.put(bs - 1, 8)
    elif bcode == 7:
        bw.put(bs - 1, 16)
    if rcode == 12:
        bw.put(sr // 1000, 8)
    elif rcode == 13:
        bw.put(sr, 16)
    bw.put(crc8(bw.tobytes()), 8)
    x = samples.astype(np.int64)
    if stereo == "independent":
        chans = [(x[:, c], bps) for c in range(ch)]
    else:
        left, right = x[:, 0], x[:, 1]
        side = left - right
        if stereo == "left_side":
            chans = [(left, bps), (side, bps + 1)]
        elif stereo == "right_side":
            chans = [(side, bps + 1), (right, bps)]
        else:
            chans = [((left + right) >> 1, bps), (side, bps + 1)]
    for c, (data, b) in enumerate(chans):
        write_subframe(bw, data, b, **sub(c))
    bw.align()
    body = bw.tobytes()
    return body + crc16(body).to_bytes(2, "big")


def write_flac(samples: np.ndarray, sr: int, bps: int, block: int = 1024, stereo: str = "independent", sub=None, with_md5: bool = True,
               total_known: bool = True, header_from_streaminfo: bool = False, extra_blocks: Sequence[bytes] = (), id3: bool = False) -> bytes:
    """samples [n][channels] (or [n]) integers within bps bits -> a complete FLAC stream."""
    samples = np.asarray(samples)
    if samples.ndim == 1:
        samples = samples[:, None]
    n, ch = samples.shape
    sub = sub or (lambda c: dict(kind="fixed", order=2, porder=2))
    frames = [write_frame(samples[i:i + block], i // block, sr, bps, stereo, sub, header_from_streaminfo) for i in range(0, n, block)]
    nbytes = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in samples.reshape(-1))
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    bw = BitWriter()
    last = min(block, n - (n - 1) // block * block)
    bw.put(block, 16)                                  # min / max block size (the last block may be shorter)
    bw.put(block, 16)
    bw.put(min(len(f) for f in frames), 24)
    bw.put(max(len(f) for f in frames), 24)
    bw.put(sr, 20)
    bw.put(ch - 1, 3)
    bw.put(bps - 1, 5)
    bw.put(n if total_known else 0, 36)
    info = bw.tobytes() + md5
    blocks = [(0, info)] + [(4, b) for b in extra_blocks]          # type 4 = VORBIS_COMMENT-like payload the decoder must skip
    out = b""
    if id3:
        payload = b"\x00" * 37
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, len(payload)]) + payload
    out += b"fLaC"
    for i, (t, payload) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(payload).to_bytes(3, "big") + payload
    return out + b"".join(frames)
