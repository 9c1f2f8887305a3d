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

# (5) f32, vectorised: x / 255 for every byte, and (fb + fg) + fr over 3 for ALL 256^3 colour triples (the preprocessing tile)
import numpy as np
def fma32(a, b, c):
    """RN32(a * b + c) for float32 arrays: the product is exact in double; the sum is formed in double and the rare values that
    land on a float32 rounding boundary after the double rounding are redone with exact rationals."""
    t = a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)
    out = t.astype(np.float32)
    bits = t.view(np.uint64) & np.uint64((1 << 29) - 1)
    tie = np.nonzero(bits == np.uint64(1 << 28))[0]
    for k in tie:
        out[k] = np.float32(float(rn(F(float(a[k])) * F(float(b[k])) + F(float(c[k])), 24)))
    return out
def mark32(a, b):
    b = np.float32(b); y = np.float32(1.0) / b
    q = a * y
    r = fma32(np.full_like(a, -b), q, a)
    assert np.array_equal(r.astype(np.float64), a.astype(np.float64) - np.float64(b) * q.astype(np.float64)), "residual not exact"
    return fma32(r, np.full_like(a, y), q)
x = np.arange(256, dtype=np.float32)
u = x / np.float32(255.0)
print("byte / 255.0f mismatches:", int((mark32(x, 255.0) != u).sum()))
bad = 0
for fr in range(256):
    s = (u[:, None] + u[None, :]).astype(np.float32) + u[fr]            # (fb + fg) + fr, float32 at each step
    s = s.reshape(-1).astype(np.float32)
    bad += int((mark32(s, 3.0) != s / np.float32(3.0)).sum())
print("((fb + fg) + fr) / 3.0f mismatches over 256^3 triples:", bad)

# (6) f32, x / 100.0f (depth_over_max) for EVERY significand: the sequence is invariant under scaling by powers of two while
# nothing overflows or underflows, so the 2^23 floats of [1, 2) stand for every normal numerator
bad = 0
for lo in range(0, 1 << 23, 1 << 20):
    a = (np.arange(lo, lo + (1 << 20), dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
    bad += int((mark32(a, 100.0) != (a.astype(np.float64) / 100.0).astype(np.float32)).sum())
print("x / 100.0f mismatches over all 2^23 significands (residual exact in every case):", bad)
