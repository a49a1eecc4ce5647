"""An independently written FLAC frame decoder in pure Python (RFC 9639).

Second opinion for the C oracle on features no reference fixture exercises
(LPC, fixed orders 1-4, 5-bit Rice parameters, escape partitions, wasted bits,
more than two channels, 24-bit).  Written from the format description alone, with
a different structure from oracle/flac_oracle.c: the whole frame is one Python
integer and fields are cut out of it by shifting.  Small inputs only.
"""
import struct


class Bits:
    def __init__(self, data: bytes, start_byte: int = 0):
        self.n = len(data) * 8
        self.v = int.from_bytes(data, "big")
        self.p = start_byte * 8

    def u(self, k: int) -> int:
        if k == 0:
            return 0
        if self.p + k > self.n:
            raise EOFError("past the end of the frame")
        r = (self.v >> (self.n - self.p - k)) & ((1 << k) - 1)
        self.p += k
        return r

    def s(self, k: int) -> int:
        r = self.u(k)
        return r - (1 << k) if k and r >> (k - 1) else r

    def unary(self) -> int:
        q = 0
        while self.u(1) == 0:
            q += 1
        return q


def crc(data: bytes, poly: int, width: int) -> int:
    top, mask, c = 1 << (width - 1), (1 << width) - 1, 0
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


BLOCKSIZES = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192,
              14: 16384, 15: 32768}
SAMPLE_SIZES = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}
FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def parse_streaminfo(b: bytes) -> dict:
    mn, mx = struct.unpack(">HH", b[:4])
    w = int.from_bytes(b[10:18], "big")
    return dict(min_blocksize=mn, max_blocksize=mx, min_framesize=int.from_bytes(b[4:7], "big"),
                max_framesize=int.from_bytes(b[7:10], "big"), sample_rate=w >> 44, channels=((w >> 41) & 7) + 1,
                bps=((w >> 36) & 31) + 1, total_samples=w & ((1 << 36) - 1), md5=b[18:34])


def residual(br: Bits, blocksize: int, order: int):
    method = br.u(2)
    assert method < 2
    pbits = 5 if method else 4
    po = br.u(4)
    out = []
    for part in range(1 << po):
        count = (blocksize >> po) - (order if part == 0 else 0)
        k = br.u(pbits)
        if k == (1 << pbits) - 1:
            width = br.u(5)
            out += [br.s(width) for _ in range(count)]
        else:
            for _ in range(count):
                val = (br.unary() << k) | br.u(k)
                out.append((val >> 1) if not val & 1 else -((val + 1) >> 1))
    return out


def subframe(br: Bits, blocksize: int, bps: int):
    assert br.u(1) == 0
    kind = br.u(6)
    wasted = br.unary() + 1 if br.u(1) else 0
    bps -= wasted
    if kind == 0:
        out = [br.s(bps)] * blocksize
    elif kind == 1:
        out = [br.s(bps) for _ in range(blocksize)]
    elif 8 <= kind <= 12 or kind >= 32:
        order = kind - 8 if kind < 32 else kind - 31
        out = [br.s(bps) for _ in range(order)]
        if kind >= 32:
            precision = br.u(4) + 1
            assert precision != 16
            shift = br.s(5)
            assert shift >= 0
            coefs = [br.s(precision) for _ in range(order)]
        else:
            shift, coefs = 0, FIXED[order]
        for r in residual(br, blocksize, order):
            acc = sum(c * out[-1 - j] for j, c in enumerate(coefs))
            out.append(r + (acc >> shift))
    else:
        raise ValueError("reserved subframe type")
    return [x << wasted for x in out]


def decode_frame(frame: bytes, si: dict):
    """Returns (first_sample, [channel lists]) for one complete frame (sync code to CRC-16)."""
    assert frame[0] == 0xFF and frame[1] & 0xFE == 0xF8
    assert crc(frame, 0x8005, 16) == 0, "CRC-16"
    variable = frame[1] & 1
    br = Bits(frame, 2)
    bs_code, sr_code, ch, ss_code, rsv = br.u(4), br.u(4), br.u(4), br.u(3), br.u(1)
    assert rsv == 0 and bs_code != 0 and sr_code != 15 and ch <= 10 and ss_code != 3
    lead = br.u(8)
    nbytes = 0
    while lead & (0x80 >> nbytes):
        nbytes += 1
    number = lead & (0xFF >> (nbytes + 1)) if nbytes else lead
    for _ in range(max(0, nbytes - 1)):
        c = br.u(8)
        assert c >> 6 == 2
        number = (number << 6) | (c & 0x3F)
    if bs_code == 6:
        blocksize = br.u(8) + 1
    elif bs_code == 7:
        blocksize = br.u(16) + 1
    else:
        blocksize = BLOCKSIZES[bs_code]
    if sr_code == 12:
        br.u(8)
    elif sr_code in (13, 14):
        br.u(16)
    hdr_len = br.p // 8
    assert crc(frame[:hdr_len], 0x07, 8) == br.u(8), "CRC-8"
    bps = SAMPLE_SIZES[ss_code] if ss_code else si["bps"]
    nch = ch + 1 if ch < 8 else 2
    chans = []
    for c in range(nch):
        side = (ch == 8 and c == 1) or (ch == 9 and c == 0) or (ch == 10 and c == 1)
        chans.append(subframe(br, blocksize, bps + (1 if side else 0)))
    if ch == 8:
        chans[1] = [a - b for a, b in zip(chans[0], chans[1])]
    elif ch == 9:
        chans[0] = [a + b for a, b in zip(chans[0], chans[1])]
    elif ch == 10:
        left, right = [], []
        for m, s in zip(chans[0], chans[1]):
            m = (m << 1) | (s & 1)
            left.append((m + s) >> 1)
            right.append((m - s) >> 1)
        chans = [left, right]
    assert (br.p + 7) // 8 + 2 == len(frame), "frame length"
    first = number if variable else number * si["max_blocksize"]
    return first, chans


def decode_file(data: bytes):
    """Whole file -> (streaminfo dict, packed little-endian PCM bytes).  Frames are found by
    decoding them one after another (a frame ends where its subframes end)."""
    assert data[:4] == b"fLaC"
    pos, si = 4, None
    while True:
        last, kind, length = data[pos] >> 7, data[pos] & 0x7F, int.from_bytes(data[pos + 1:pos + 4], "big")
        if kind == 0:
            si = parse_streaminfo(data[pos + 4: pos + 4 + length])
        pos += 4 + length
        if last:
            break
    nbytes = (si["bps"] + 7) // 8
    pcm = bytearray(si["total_samples"] * si["channels"] * nbytes)
    while pos < len(data):
        # find this frame's end by trying successive sync codes until the CRC-16 closes
        end = pos + 2
        while True:
            nxt = data.find(b"\xff", end)
            if nxt < 0 or nxt + 1 >= len(data):
                end = len(data)
                break
            if data[nxt + 1] & 0xFE == 0xF8 and crc(data[pos:nxt], 0x8005, 16) == 0:
                end = nxt
                break
            end = nxt + 1
        first, chans = decode_frame(data[pos:end], si)
        o = first * si["channels"] * nbytes
        for n in range(len(chans[0])):
            for c in chans:
                pcm[o:o + nbytes] = (c[n] & ((1 << (8 * nbytes)) - 1)).to_bytes(nbytes, "little")
                o += nbytes
        pos = end
    return si, bytes(pcm)
