"""The arithmetic of the table-free CRC-16 piece remainder used by k_crc16_spans
(flac_amd/csrc/frame_scan.hpp, crc16_piece), restated in Python and checked against the oracle's
bitwise CRC-16: x^16 + x^15 + x^2 + 1 = (x + 1)(x^15 + x + 1), so the remainder is rebuilt from the
message's parity and its remainder modulo the trinomial, which folds with shifts and XORs only.
The constants tested here are the ones the kernel uses."""
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
