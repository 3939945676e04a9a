"""The arithmetic behind interpn_device.h::step_cell_fast, checked on the CPU with exact rationals.

The sweep kernel computes t = RN(a / b) for a divisor b that is the same for every point (a regular grid's
step) as   q0 = RN(a rb); r0 = RN(a - b q0); q1 = RN(q0 + r0 rb); r1 = a - b q1; t = RN(q1 + r1 rb)   with
rb = RN(1 / b) from the host, instead of the hardware's divide sequence, and claims the same bits
(Markstein's theorem on the final step of fma-based division).  Here every fma is evaluated exactly with
fractions.Fraction and rounded once (float(Fraction) is correctly rounded), and the result is compared with
Python's IEEE division on
  * random operands over the exponent range the kernel admits for the short form,
  * the hardest operands there are for a division's rounding: quotients that miss the midpoint of two
    neighbouring doubles by the smallest possible amount (a = (b M -+ 1) / 2^54 for the odd 54-bit M with
    b M = +-1 mod 2^54: a / b lies 2^-54 ulp from the midpoint M),
  * quotients next to powers of two, exact quotients, and the steps of the benchmark's grids.
No GPU, no oracle: this pins the algorithm, the GPU tests pin its implementation
(tests/test_gpu_parity.py::test_sweep_cell_index_and_t_without_divisions)."""
import math
import random
from fractions import Fraction

import numpy as np


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def div_short(a, b):
    rb = 1.0 / b
    q0 = a * rb
    r0 = fma(-b, q0, a)
    q1 = fma(r0, rb, q0)
    r1 = fma(-b, q1, a)
    assert Fraction(r1) == Fraction(a) - Fraction(b) * Fraction(q1), "the remainder of a faithful quotient is exact"
    return fma(r1, rb, q1)


def same(x, y):
    return x == y and math.copysign(1.0, x) == math.copysign(1.0, y)


def test_random_operands():
    rnd = random.Random(5)
    for _ in range(60_000):
        b = math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-128, 127)) * rnd.choice((-1.0, 1.0))
        a = math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-256, 255)) * rnd.choice((-1.0, 1.0))
        assert same(div_short(a, b), a / b), (a.hex(), b.hex())


def test_quotients_next_to_midpoints():
    """b M = +-1 (mod 2^54) for an odd 54-bit M: a = (b M -+ 1) / 2^54 is an integer and a / b misses the
    midpoint M (in units of half an ulp of the quotient) by 1 / b — nothing is closer."""
    rnd = random.Random(11)
    done = 0
    while done < 6000:
        B = rnd.getrandbits(53) | (1 << 52) | 1
        for sign in (1, -1):
            M = (sign * pow(B, -1, 1 << 54)) % (1 << 54)
            if M < (1 << 53):
                continue
            A = (B * M - sign) >> 54
            assert (A << 54) + sign == B * M
            if A >= (1 << 53) or A == 0:
                continue
            e = rnd.randint(-100, 100)
            a, b = math.ldexp(float(A), e), math.ldexp(float(B), -52)
            assert float(A) == A
            assert same(div_short(a, b), a / b), (a.hex(), b.hex())
            assert same(div_short(-a, b), -a / b)
            done += 1


def test_special_quotients_and_benchmark_steps():
    rnd = random.Random(3)
    steps = [2.0 / 63, 2.0 / 127, 2.0 / 31, 0.1, 1.0 / 3.0, 1.0, 0.5, 3.0, 1e-30, 3e30, math.ldexp(1.0, -128), math.ldexp(1.0, 128),
             float(np.nextafter(2.0, 1.0)), float(np.nextafter(1.0, 2.0))]  # (the last two: significands all ones / 1 + ulp)
    for b in steps:
        for k in range(0, 70):
            for a in (b * k, float(np.nextafter(b * k, np.inf)), float(np.nextafter(b * k, -np.inf)), b * (k + 0.5), b * (k + rnd.random())):
                if a == 0.0 or not (2.0 ** -256 <= abs(a) < 2.0 ** 256):
                    continue
                assert same(div_short(a, b), a / b), (a.hex(), b.hex())
        for _ in range(3000):
            a = math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-60, 60))
            assert same(div_short(a, b), a / b), (a.hex(), b.hex())
            q = math.ldexp(1.0, rnd.randint(-40, 40))  # quotients at and next to powers of two
            for qq in (q, float(np.nextafter(q, 0.0)), float(np.nextafter(q, np.inf))):
                a2 = qq * b
                if 2.0 ** -256 <= abs(a2) < 2.0 ** 256:
                    assert same(div_short(a2, b), a2 / b), (a2.hex(), b.hex())


def test_cell_index_from_the_reciprocal():
    """floor(RN(a0 / b)) = floor(RN(a0 rb)) whenever RN(a0 rb) is further than 2^-20 from every integer and below
    2^31 in magnitude (the condition under which the kernel keeps the short form)."""
    rnd = random.Random(8)
    checked = 0
    for _ in range(200_000):
        b = rnd.choice((2.0 / 63, 0.1, 1.0 / 3.0, 7.0, 1e-30, 3e30))
        k = rnd.randint(-5, 2 ** rnd.randint(1, 30))
        near = rnd.random() < 0.5
        a0 = b * (k + (rnd.choice((-1, 1)) * 2.0 ** rnd.uniform(-24, -16) if near else rnd.random()))
        qt = a0 * (1.0 / b)
        d = qt - math.floor(qt)
        if abs(d - 0.5) < 0.5 - 2.0 ** -20 and abs(qt) < 2.0 ** 31:
            assert math.floor(qt) == math.floor(a0 / b), (a0.hex(), b.hex())
            checked += 1
    assert checked > 100_000


# ---- f32: the same forms with p = 24 (rounding to f32 done exactly: float(Fraction) would round twice) ----
def rn32(fr):
    """Fraction -> nearest float32 (ties to even), exact; normal range only."""
    if fr == 0:
        return np.float32(0.0)
    sign = -1 if fr < 0 else 1
    fr = abs(fr)
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    if Fraction(2) ** e > fr:
        e -= 1
    assert -126 <= e <= 127
    scaled = fr / Fraction(2) ** (e - 23)  # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(sign * math.ldexp(float(n), e - 23))


def fma32(a, b, c):
    return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def div_short32(a, b):
    a, b = np.float32(a), np.float32(b)
    rb = np.float32(1.0) / b
    q0 = a * rb
    r0 = fma32(-b, q0, a)
    q1 = fma32(r0, rb, q0)
    r1 = fma32(-b, q1, a)
    assert Fraction(float(r1)) == Fraction(float(a)) - Fraction(float(b)) * Fraction(float(q1))
    return fma32(r1, rb, q1)


def test_f32_random_and_midpoint_operands():
    rnd = random.Random(21)
    for _ in range(20_000):
        b = np.float32(math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-16, 15)) * rnd.choice((-1.0, 1.0)))
        a = np.float32(math.ldexp(rnd.uniform(1.0, 2.0), rnd.randint(-24, 23)) * rnd.choice((-1.0, 1.0)))
        got, want = div_short32(a, b), a / b
        assert got == want and type(want) is np.float32, (float(a).hex(), float(b).hex())
    done = 0
    while done < 4000:  # quotients 2^-25 ulp from a midpoint: b M = +-1 (mod 2^25), M odd with 25 bits
        B = rnd.getrandbits(24) | (1 << 23) | 1
        for sign in (1, -1):
            M = (sign * pow(B, -1, 1 << 25)) % (1 << 25)
            if M < (1 << 24):
                continue
            A = (B * M - sign) >> 25
            if A >= (1 << 24) or A == 0:
                continue
            e = rnd.randint(-12, 12)
            a, b = np.float32(math.ldexp(float(A), e - 23)), np.float32(math.ldexp(float(B), -23))
            if not (2.0 ** -24 <= abs(float(a)) < 2.0 ** 24):
                continue
            assert div_short32(a, b) == a / b, (float(a).hex(), float(b).hex())
            done += 1


def test_f32_cell_index_from_the_reciprocal():
    rnd = random.Random(9)
    checked = 0
    for _ in range(100_000):
        b = np.float32(rnd.choice((2.0 / 63, 0.1, 1.0 / 3.0, 7.0, 2.0 / 159)))
        k = rnd.randint(-5, 2 ** rnd.randint(1, 19))
        near = rnd.random() < 0.5
        a0 = np.float32(float(b) * (k + (rnd.choice((-1, 1)) * 2.0 ** rnd.uniform(-22, -8) if near else rnd.random())))
        qt = a0 * (np.float32(1.0) / b)
        d = qt - np.floor(qt)
        margin = fma32(abs(qt), np.float32(2.0 ** -21), np.float32(2.0 ** -21))
        if np.float32(abs(np.float32(d - np.float32(0.5))) + margin) < np.float32(0.5):
            assert np.floor(qt) == np.floor(a0 / b), (float(a0).hex(), float(b).hex())
            checked += 1
    assert checked > 50_000


# ---- zero numerators (interpn_device.h::fast_numerator admits +0, not -0) ----
def fma_signed_zero(a, b, c):
    """fma with IEEE's sign rule for an exact zero sum under round-to-nearest: -0 only if both addends are -0."""
    exact = Fraction(a) * Fraction(b) + Fraction(c)
    if exact != 0:
        return float(exact)
    prod_negative = (math.copysign(1.0, a) * math.copysign(1.0, b)) < 0
    if Fraction(a) * Fraction(b) == 0 and Fraction(c) == 0:
        return -0.0 if (prod_negative and math.copysign(1.0, c) < 0) else 0.0
    return 0.0  # x + (-x)


def test_zero_numerators_of_the_short_division():
    """The rectilinear multicubic node divides value differences by a positive spacing ratio; equal neighbours give +0.
    With b > 0 the short form returns +0 = +0 / b; a numerator -0 would come back as +0 (wrong sign), which is why
    fast_numerator refuses it."""
    def short(a, b):
        rb = 1.0 / b
        q0 = a * rb
        r0 = fma_signed_zero(-b, q0, a)
        q1 = fma_signed_zero(r0, rb, q0)
        r1 = fma_signed_zero(-b, q1, a)
        return fma_signed_zero(r1, rb, q1)

    for b in (1.5, float.fromhex("0x1p-128"), float.fromhex("0x1.fffffffffffffp127"), 3.0e-7):
        assert same(short(0.0, b), 0.0 / b)
        assert not same(short(-0.0, b), -0.0 / b)  # (+0 where the division gives -0)
