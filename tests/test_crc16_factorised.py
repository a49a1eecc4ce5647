"""The arithmetic of the table-free CRC-16 used by k_scan_tiles and k_crc16_spans (flac_amd/csrc/frame_scan.hpp:
crc16_fold15, crc16_words_q, crc16_times_x16, crc16_join; the wide accumulators; the join over a wave's lanes),
restated in Python word by word and checked against the oracle's bitwise CRC-16:
x^16 + x^15 + x^2 + 1 = (x + 1)(x^15 + x + 1), so the remainder is rebuilt from the message's parity and its
remainder modulo the trinomial, which folds with shifts and XORs only.  The constants tested here are the ones
the kernels use."""
import numpy as np

from util import oracle


def _fold15(w):
    h = w >> 15
    return (w & 0x7FFF) ^ h ^ (h << 1)


def piece_remainder(b16: bytes) -> int:
    words = [int.from_bytes(b16[4 * i: 4 * i + 4], "big") for i in range(4)]
    parity = sum(bin(w).count("1") for w in words) & 1
    a, b, c, d = (_fold15(w) for w in words)
    assert max(a, b, c, d) < 1 << 18
    # x^96 = x^12 + x^10 + x^8 + x^6, x^64 = x^8 + x^4, x^32 = x^4 + x^2 (mod x^15 + x + 1)
    t = (a << 12) ^ (a << 10) ^ (a << 8) ^ (a << 6) ^ (b << 8) ^ (b << 4) ^ (c << 4) ^ (c << 2) ^ d
    assert t < 1 << 30                       # fits the kernel's 32-bit registers
    t = _fold15(t)
    t = (t << 2) ^ (t << 1)                  # x^16 = x^2 + x
    t = _fold15(t)
    assert t < 1 << 15
    flip = (parity + bin(t).count("1")) & 1
    return t ^ (0x8003 if flip else 0)


def test_piece_remainder_equals_crc16():
    rng = np.random.default_rng(16)
    cases = [bytes(16), bytes([0xFF] * 16)]
    cases += [bytes([0] * i + [1 << b] + [0] * (15 - i)) for i in range(16) for b in range(8)]   # every single bit
    cases += [bytes(rng.integers(0, 256, size=16).astype(np.uint8)) for _ in range(3000)]
    for m in cases:
        assert piece_remainder(m) == oracle.crc16(m), m.hex()


# ---- wide accumulators (crc16_acc_step / crc16_acc_fold / crc16_acc_merge<6> + crc16_join of frame_scan.hpp), word by word ----

_M32 = 0xFFFFFFFF


def _alignbit(hi, lo, s):          # v_alignbit_b32
    return (((hi << 32) | lo) >> (s & 31)) & _M32


def _acc_step(A, piece):           # A[0] least significant; piece: 16 bytes in memory order
    W = [int.from_bytes(piece[4 * i: 4 * i + 4], "big") for i in range(4)]
    A[4] = _alignbit(A[4], A[3], 31)
    A[3] = _alignbit(A[3], A[2], 31) ^ W[0]
    A[2] = _alignbit(A[2], A[1], 31) ^ W[1]
    A[1] = _alignbit(A[1], A[0], 31) ^ W[2]
    A[0] = ((A[0] << 1) & _M32) ^ W[3]


def _fold128(t, lo, hi):           # x^128 = x^8 + x^2 + x
    return lo ^ (((t << 8) ^ (t << 2) ^ (t << 1)) & _M32), hi ^ (t >> 24) ^ (t >> 30) ^ (t >> 31)


def _acc_fold(A):
    A[0], A[1] = _fold128(A[4], A[0], A[1])
    A[4] = 0


def _acc_flush(n0, n1, n2, n3):    # newest step first; Horner with x^8192 = x^8 + x^4 + x
    h = list(n3) + [0]
    for n in (n2, n1, n0):
        new = [0] * 6
        for i in range(5, 0, -1):
            new[i] = (_alignbit(h[i], h[i - 1], 24) ^ _alignbit(h[i], h[i - 1], 28) ^ _alignbit(h[i], h[i - 1], 31)
                      ^ (n[i] if i < 5 else 0))
        new[0] = (((h[0] << 8) ^ (h[0] << 4) ^ (h[0] << 1)) & _M32) ^ n[0]
        h = new
    h[1], h[2] = _fold128(h[5], h[1], h[2])
    h[0], h[1] = _fold128(h[4], h[0], h[1])
    return piece_remainder(b"".join(w.to_bytes(4, "big") for w in (h[3], h[2], h[1], h[0])))


def lane_remainder(pieces):
    """What one lane of k_crc16_spans computes: the CRC-16 of its 16-byte pieces taken 1 KiB apart, the last one
    ending the message."""
    steps = len(pieces)
    acc = [[0] * 5 for _ in range(4)]
    for t, piece in enumerate(pieces):
        _acc_step(acc[(steps - 1 - t) & 3], piece)
        if (t & 127) == 127:
            for a in acc:
                _acc_fold(a)
    return _acc_flush(*acc)


def test_wide_accumulators_equal_crc16():
    rng = np.random.default_rng(17)
    for steps in list(range(1, 14)) + [127, 128, 129, 255, 256, 257, 700]:
        for kind in ("random", "ones"):
            if kind == "random":
                pieces = [bytes(rng.integers(0, 256, size=16).astype(np.uint8)) for _ in range(steps)]
            else:
                pieces = [bytes([0xFF] * 16)] * steps
            message = (b"".join(p + bytes(1008) for p in pieces[:-1])) + pieces[-1]
            assert lane_remainder(pieces) == oracle.crc16(message), (steps, kind)


# ---- the scan's flush: five-word merge and the join over the lanes of a wave (crc16_acc_merge<5>, crc16_wave_join) ----

def _merge_q(n0, n1, n2, n3, words):
    """Remainder modulo x^15 + x + 1 (times x^16) and parity of a lane's accumulators."""
    h = list(n3) + [0]
    for n in (n2, n1, n0):
        new = list(h)
        for i in range(words - 1, 0, -1):
            new[i] = (_alignbit(h[i], h[i - 1], 24) ^ _alignbit(h[i], h[i - 1], 28) ^ _alignbit(h[i], h[i - 1], 31)
                      ^ (n[i] if i < 5 else 0))
        new[0] = (((h[0] << 8) ^ (h[0] << 4) ^ (h[0] << 1)) & _M32) ^ n[0]
        h = new
    if words == 6:
        h[1], h[2] = _fold128(h[5], h[1], h[2])
    else:
        assert h[5] == 0 and h[4] < 1 << 27     # 131 + 24 bits
    h[0], h[1] = _fold128(h[4], h[0], h[1])
    parity = sum(bin(w).count("1") for w in h[:4]) & 1
    a, b, c, d = (_fold15(w) for w in (h[3], h[2], h[1], h[0]))
    t = (a << 12) ^ (a << 10) ^ (a << 8) ^ (a << 6) ^ (b << 8) ^ (b << 4) ^ (c << 4) ^ (c << 2) ^ d
    return _fold15(t), parity            # below 2^17, without the CRC's factor x^16


def _wave_join(q, parity):
    """q[l], parity[l] of 64 lanes; lane l stands 128 * (63 - l) bits above lane 63."""
    q = list(q)

    def shr(v, n):      # row_shr:n inside rows of 16 lanes, zero where there is no source
        return [v[l - n] if (l % 16) >= n else 0 for l in range(64)]
    for n, shifts, fold in ((1, (8, 2, 1), False), (2, (4, 1), True), (4, (8, 2), False), (8, (4, 2, 1), True)):
        t = shr(q, n)
        for l in range(64):
            w = q[l]
            for s in shifts:
                w ^= t[l] << s
            assert w < 1 << 32           # the kernel's registers
            q[l] = _fold15(w) if fold else w
    r = q[15]
    for row in range(1, 4):
        r = _fold15((r << 8) ^ (r << 4) ^ (r << 2)) ^ q[16 * row + 15]
    r = _fold15((r << 2) ^ (r << 1))     # times x^16
    assert r < 1 << 15
    flip = (sum(parity) + bin(r).count("1")) & 1
    return r ^ (0x8003 if flip else 0)


def test_scan_flush_equals_crc16():
    """A wave's worth of data: `chunks` consecutive KiB (at most 16, as between two cuts of the scan), accumulated
    by chunk index modulo 4 and flushed as the scan kernel does."""
    rng = np.random.default_rng(18)
    for chunks in (1, 2, 3, 4, 5, 7, 8, 13, 16):
        for start in (0, 1, 2, 3):           # chunk index (modulo 4) of the first chunk
            data = bytes(rng.integers(0, 256, size=1024 * chunks).astype(np.uint8))
            qs, ps = [], []
            for lane in range(64):
                acc = [[0] * 5 for _ in range(4)]
                for c in range(chunks):
                    _acc_step(acc[(start + c) & 3], data[1024 * c + 16 * lane: 1024 * c + 16 * lane + 16])
                k = (start + chunks - 1) & 3
                q, p = _merge_q(acc[k], acc[(k + 3) & 3], acc[(k + 2) & 3], acc[(k + 1) & 3], 5)
                q6, p6 = _merge_q(acc[k], acc[(k + 3) & 3], acc[(k + 2) & 3], acc[(k + 1) & 3], 6)
                assert (q, p) == (q6, p6)
                qs.append(q)
                ps.append(p)
            assert _wave_join(qs, ps) == oracle.crc16(data), (chunks, start)


# ---- the record scheme of k_scan_tiles / k_span_crc, on the CPU: cuts, zero padding, joins by powers of x ----

_P = 0x18005


def _gf_mulmod(a, b):
    r = 0
    for i in range(15, -1, -1):
        r <<= 1
        if r & 0x10000:
            r ^= _P
        if (b >> i) & 1:
            r ^= a
    return r & 0xFFFF


def _xpow8(n):                      # x^(8 n) modulo P
    r, base = 1, 0x100
    while n:
        if n & 1:
            r = _gf_mulmod(r, base)
        base = _gf_mulmod(base, base)
        n >>= 1
    return r


RANGE_LOG2 = 15   # kSegRangeLog2 of frame_scan.hpp
RANGE = 1 << RANGE_LOG2


def _scan_records(data, cuts):
    """What the scan leaves for one input: per range (RANGE bytes, one wave's share of a scan tile) the cuts inside it (offset in the range, CRC-16 of the
    bytes since the cut before or the range's start, zero-padded to the end of the cut's 1 KiB step) and the CRC-16
    from the last cut to the range's end."""
    ranges = []
    for base in range(0, len(data), RANGE):
        last, recs = base, []
        for p in sorted(c for c in cuts if base <= c < base + RANGE):
            pad = 1024 - (p & 1023)
            recs.append((p - base, oracle.crc16(data[last:p] + bytes(pad))))
            last = p
        ranges.append((recs, oracle.crc16(data[last:base + RANGE])))
    return ranges


def _span_crc(ranges, a, e):
    """k_span_crc's walk: CRC-16 of [a, e), both of them cuts, from the records."""
    rng_i = a >> RANGE_LOG2
    recs, end_crc = ranges[rng_i]
    i = [off for off, _ in recs].index(a & (RANGE - 1))
    crc, pos = 0, a
    while pos < e:
        if i + 1 < len(recs):
            i += 1
            off, v = recs[i]
            b = (rng_i << RANGE_LOG2) + off
            v = _gf_mulmod(v, _xpow8(32767 - (1024 - (off & 1023))))      # x has order 32767: takes the padding out
        else:
            b, v = (rng_i + 1) << RANGE_LOG2, end_crc
            if b < e:
                rng_i += 1
                recs, end_crc = ranges[rng_i]
                i = -1
        assert b <= e
        crc = _gf_mulmod(crc, _xpow8(b - pos)) ^ v if crc else v
        pos = b
    return crc


def test_record_walk_equals_crc16():
    rng = np.random.default_rng(19)
    data = bytes(rng.integers(0, 256, size=5 * RANGE).astype(np.uint8))
    cuts = sorted(set(int(c) for c in rng.integers(0, len(data), size=40)) | {0, 1023, 1024, RANGE, 3 * RANGE - 1, 3 * RANGE})
    ranges = _scan_records(data, cuts)
    for n, a in enumerate(cuts):
        for e in cuts[n + 1: n + 12]:
            assert _span_crc(ranges, a, e) == oracle.crc16(data[a:e]), (a, e)
