"""A second, deliberately naive implementation of the exactly-defined pieces of the path
(integer RNG, IEEE f32 arithmetic of rand(), Moeller-Trumbore, the AABB gate), used to
cross-check oracle/pt_oracle.c. Every f32 operation is done in exact rational arithmetic
and rounded once to binary32 (round-to-nearest-even), so fused multiply-adds are modelled
exactly. Pure Python: for a handful of rays only.

Follows reference src/raytrace.wgsl:60-99 and src/query.wgsl:95-147.
"""
from __future__ import annotations

import math
import struct
from fractions import Fraction

import numpy as np

M32 = 0xFFFFFFFF


def f32(x) -> float:
    """Round a Fraction / int / float exactly once to binary32; returns a Python float holding that value."""
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return x
        x = Fraction(x)
    x = Fraction(x)
    if x == 0:
        return 0.0
    sign = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    # now 2^e <= a < 2^(e+1)
    e = max(e, -126)
    scale = Fraction(2) ** (e - 23)
    q = a / scale
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    val = Fraction(n) * scale
    if val >= Fraction(2) ** 128:
        return sign * math.inf
    return sign * float(val)


def bits(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def fma(a, b, c):
    if any(math.isnan(v) or math.isinf(v) for v in (a, b, c)):
        return float(np.float32(np.float64(a) * np.float64(b) + np.float64(c)))
    return f32(Fraction(a) * Fraction(b) + Fraction(c))


def mul(a, b):
    return float(np.float32(a) * np.float32(b))


def add(a, b):
    return float(np.float32(a) + np.float32(b))


def sub(a, b):
    return float(np.float32(a) - np.float32(b))


def div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(np.float32(a) / np.float32(b))


def dot3(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], mul(a[0], b[0])))


def cross3(a, b):
    return (fma(a[1], b[2], -mul(a[2], b[1])), fma(a[2], b[0], -mul(a[0], b[2])), fma(a[0], b[1], -mul(a[1], b[0])))


# ---- RNG ---------------------------------------------------------------------------
def pcg3d(state):
    """src/raytrace.wgsl:60-78 on Python ints."""
    x, y, z = [(v * 1664525 + 1013904223) & M32 for v in state]
    x = (x + y * z) & M32
    y = (y + z * x) & M32
    z = (z + x * y) & M32
    x ^= x >> 16
    y ^= y >> 16
    z ^= z >> 16
    x = (x + y * z) & M32
    y = (y + z * x) & M32
    z = (z + x * y) & M32
    return (x, y, z)


def rand_from(v):
    """src/raytrace.wgsl:80-82 for one already drawn pcg3d() vector."""
    out = []
    for c in v:
        q = div(f32(Fraction(c)), 3141.592653589793)  # u32 -> f32 rounds to nearest even; literal rounds to f32 in div()
        out.append(abs(sub(q, float(math.floor(q)))))
    return tuple(out)


def u32_of_f32(f):
    if not (f > 0.0):
        return 0
    if f >= 4294967296.0:
        return M32
    return int(f)


def rng_setup(px, py, width, height, time):
    """src/raytrace.wgsl:97-99 + :210 with uv = ((px+0.5)/W, (py+0.5)/H)."""
    W, H = float(width), float(height)
    uvx, uvy = div(add(float(px), 0.5), W), div(add(float(py), 0.5), H)
    tp1 = add(float(np.float32(time)), 1.0)
    sx, sy = mul(mul(uvx, W), tp1), mul(mul(uvy, H), tp1)
    a, b = u32_of_f32(sx), u32_of_f32(sy)
    return (a, b, a ^ b)


# ---- geometry ------------------------------------------------------------------------
EPSILON = float(np.float32(4.88e-4))
T_MIN = float(np.float32(0.0001))


def moller_trumbore(pa, pb, pc, pos, dir_, best_t):
    """src/query.wgsl:95-111. Returns (accepted, t, u, v)."""
    e_ab = tuple(sub(pb[i], pa[i]) for i in range(3))
    e_ac = tuple(sub(pc[i], pa[i]) for i in range(3))
    n = cross3(e_ab, e_ac)
    ao = tuple(sub(pos[i], pa[i]) for i in range(3))
    dao = cross3(ao, dir_)
    det = -dot3(dir_, n)
    inv_det = div(1.0, det)
    t = mul(dot3(ao, n), inv_det)
    u = mul(dot3(e_ac, dao), inv_det)
    v = mul(-dot3(e_ab, dao), inv_det)
    w = sub(sub(1.0, u), v)
    rejected = det < EPSILON or t < T_MIN or t > best_t or u < 0.0 or v < 0.0 or w < 0.0
    return (not rejected), t, u, v


def hit_box(bmin, bmax, pos, dir_):
    """src/query.wgsl:127-147."""
    inv = [div(1.0, d) for d in dir_]
    tmin = [mul(sub(bmin[i], pos[i]), inv[i]) for i in range(3)]
    tmax = [mul(sub(bmax[i], pos[i]), inv[i]) for i in range(3)]

    def fmin(a, b):
        return b if math.isnan(a) else a if math.isnan(b) else min(a, b)

    def fmax(a, b):
        return b if math.isnan(a) else a if math.isnan(b) else max(a, b)

    t1 = [fmin(tmin[i], tmax[i]) for i in range(3)]
    t2 = [fmax(tmin[i], tmax[i]) for i in range(3)]
    near = fmax(fmax(t1[0], t1[1]), t1[2])
    far = fmin(fmin(t2[0], t2[1]), t2[2])
    if far >= near and far > 0.0:
        return near if near > 0.0 else 10000000.0
    return 0.0


def brute_force_identity_scene(positions, indices, bmin, bmax, pos, dir_):
    """hit_all for ONE object with identity transforms (so local ray == world ray bit for bit)."""
    best_t, best_tri, best_uv = 1000.0, None, (0.0, 0.0)
    if hit_box(bmin, bmax, pos, dir_) < T_MIN:
        return None, best_t, best_uv
    for i in range(len(indices) // 3):
        a, b, c = (tuple(float(x) for x in positions[indices[3 * i + k]]) for k in range(3))
        ok, t, u, v = moller_trumbore(a, b, c, pos, dir_, best_t)
        if ok:
            best_t, best_tri, best_uv = t, i, (u, v)
    return best_tri, best_t, best_uv
