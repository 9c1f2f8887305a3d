from fractions import Fraction as F
import random, math, struct
def rn(x, p):
    if x == 0: return F(0)
    s = -1 if x < 0 else 1; x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if F(2) ** e > x: e -= 1
    if F(2) ** (e + 1) <= x: e += 1
    sc = x / F(2) ** (e - p + 1)
    n, r = divmod(sc.numerator, sc.denominator)
    twice = 2 * r
    if twice > sc.denominator or (twice == sc.denominator and (n & 1)): n += 1
    return s * n * F(2) ** (e - p + 1)
def mark(a, b, p):
    a, b = F(a), F(b)
    y = rn(1 / b, p)
    q = rn(a * y, p)
    r = rn(a - b * q, p)
    assert r == a - b * q, "residual not exact"
    return rn(q + r * y, p)
# (1) f32: b / 60000 for every u16
bad = 0
for b0 in range(65536):
    if mark(b0, 60000, 24) != rn(F(b0, 60000), 24): bad += 1
print("u16 / 60000.0f mismatches:", bad)
# (2) f64: integer / 2000 for every u16
bad = 0
for b2 in range(65536):
    if mark(b2, 2000, 53) != rn(F(b2, 2000), 53): bad += 1
print("u16 / 2000.0 mismatches:", bad)
# (3) f64: (10 + v) / 20, v float32 -- random + near-midpoint candidates
def f32(x): return struct.unpack('f', struct.pack('f', x))[0]
random.seed(1); bad = 0; n = 0
for _ in range(300000):
    v = f32(random.uniform(-12, 12) * 10 ** random.uniform(-3, 1))
    a = rn(F(10) + F(v), 53)
    if mark(a, 20, 53) != rn(a / 20, 53): bad += 1
    n += 1
print("(10 + v) / 20.0 mismatches:", bad, "of", n)
# (4) generic a / b with a = (fi + s0) * 2 (float sum, exact in double, doubled), b = float param
bad = 0; n = 0
for _ in range(300000):
    fi = random.randrange(0, 4096); b0 = random.randrange(0, 65536)
    s0 = rn(F(b0, 60000), 24)
    a = rn(F(fi) + s0, 24) * 2
    b = F(f32(random.choice([512.0, 1024.0, 100.0, 333.3, 0.73, 1e-3, 57.0]) * random.uniform(0.5, 2.0)))
    if mark(a, b, 53) != rn(a / b, 53): bad += 1
    n += 1
print("generic a / param mismatches:", bad, "of", n)
