"""Deterministic synthetic volumes of BASELINE.json's configs.

Each generator works on numpy (`xp=numpy`, CPU parity tests at small sizes) and on torch
(`xp=torch`, bench volumes generated directly in HBM), and can produce any z-range of the
volume so that each rank of a multi-GPU run builds only its own slab (+halo).

  sphere_sdf      config 3: float32 signed distance to an off-lattice sphere, iso 0.0
  marschner_lobb  config 4: float32 Marschner-Lobb test signal inside a one-voxel shell of 0, iso 0.5
  gradient_noise  config 5: uint8 integer-only lattice-gradient noise, iso 128

sphere_sdf and gradient_noise use only IEEE +,-,*,sqrt / integer arithmetic, so numpy and
torch (CPU or GPU) produce identical bytes.  marschner_lobb uses sin/cos, which are not
bit-portable: whoever generates it hands the same bytes to both sides of a comparison.
"""
import math

import numpy as np


class _NP:
    f64, f32, i64, u8 = np.float64, np.float32, np.int64, np.uint8

    @staticmethod
    def arange(a, b, dtype, device=None):
        return np.arange(a, b, dtype=dtype)

    sqrt, sin, cos, where, clip = np.sqrt, np.sin, np.cos, np.where, np.clip

    @staticmethod
    def astype(a, dt):
        return a.astype(dt)

    @staticmethod
    def zeros(shape, dtype, device=None):
        return np.zeros(shape, dtype=dtype)


def _tx():
    import torch

    class _TX:
        f64, f32, i64, u8 = torch.float64, torch.float32, torch.int64, torch.uint8

        @staticmethod
        def arange(a, b, dtype, device=None):
            return torch.arange(a, b, dtype=dtype, device=device)

        sqrt, sin, cos, where = torch.sqrt, torch.sin, torch.cos, torch.where

        @staticmethod
        def clip(a, lo, hi):
            return torch.clamp(a, lo, hi)

        @staticmethod
        def astype(a, dt):
            return a.to(dt)

        @staticmethod
        def zeros(shape, dtype, device=None):
            return torch.zeros(shape, dtype=dtype, device=device)

    return _TX


def _backend(xp):
    return _NP if xp is np or xp == "numpy" else _tx()


def sphere_sdf(n, z0=0, z1=None, xp=np, device=None):
    """f(x,y,z) = float(R - |p - c|), R = 0.4 n, c = ((n-1)/2 + 1/4, (n-1)/2 + 1/8, (n-1)/2 + 1/16)."""
    B = _backend(xp)
    z1 = n if z1 is None else z1
    c = (n - 1) / 2.0
    x = B.arange(0, n, B.f64, device) - (c + 0.25)
    y = B.arange(0, n, B.f64, device) - (c + 0.125)
    z = B.arange(z0, z1, B.f64, device) - (c + 0.0625)
    r2 = (z * z)[:, None, None] + (y * y)[None, :, None] + (x * x)[None, None, :]
    return B.astype(0.4 * n - B.sqrt(r2), B.f32)


def marschner_lobb(n, z0=0, z1=None, xp=np, device=None, alpha=0.25, fm=6.0, period=None):
    """Marschner-Lobb signal sampled at the centres of the inner (n-2)^3 voxels over [-1,1]^3, with
    a one-voxel shell of 0.0 (the reference's data convention, h:55-57).  `period`: if given, the
    volume repeats every `period` slices in z (stacked copies for weak-scaling runs)."""
    B = _backend(xp)
    z1 = n if z1 is None else z1
    m = n - 2

    def coord(i):
        return -1.0 + (2.0 * (i - 1.0) + 1.0) / m

    xi = B.arange(0, n, B.f64, device)
    zi = B.arange(z0, z1, B.f64, device)
    if period:
        zi = zi - period * B.astype(B.astype(zi / period, B.i64), B.f64)
    x, y, z = coord(xi), coord(xi), coord(zi)
    r = B.sqrt((y * y)[:, None] + (x * x)[None, :])
    pr = B.cos(2.0 * math.pi * fm * B.cos(math.pi * r / 2.0))
    rho = ((1.0 - B.sin(math.pi * z / 2.0))[:, None, None] + alpha * (1.0 + pr)[None, :, :]) / (2.0 * (1.0 + alpha))
    inx = (xi >= 1) & (xi <= n - 2)
    inz = (zi >= 1) & (zi <= n - 2)
    mask = inz[:, None, None] & inx[None, :, None] & inx[None, None, :]
    return B.astype(B.where(mask, rho, rho * 0.0), B.f32)


_M32 = 0xFFFFFFFF


def _hash32(ix, iy, iz, seed):
    h = ((ix * 73856093) ^ (iy * 19349663) ^ (iz * 83492791) ^ seed) & _M32
    h = h ^ (h >> 16)
    h = (h * 0x7FEB352D) & _M32
    h = h ^ (h >> 15)
    h = (h * 0x46CA68B) & _M32
    h = h ^ (h >> 16)
    return h


def _fade_q16(t):
    # 6t^5 - 15t^4 + 10t^3 with t in Q16, integer only
    t2 = (t * t) >> 16
    t3 = (t2 * t) >> 16
    return (t3 * (((t * (6 * t - (15 << 16))) >> 16) + (10 << 16))) >> 16


def gradient_noise(nx, ny, nz, z0=0, z1=None, xp=np, device=None, seed=740, base_period=128, octaves=3):
    """uint8 Perlin-style lattice-gradient noise, integer arithmetic only (identical bytes from
    numpy and torch on any device)."""
    B = _backend(xp)
    z1 = nz if z1 is None else z1
    X = B.arange(0, nx, B.i64, device)[None, None, :]
    Y = B.arange(0, ny, B.i64, device)[None, :, None]
    Z = B.arange(z0, z1, B.i64, device)[:, None, None]
    total = None
    for o in range(octaves):
        P = max(base_period >> o, 2)
        ix, iy, iz = X // P, Y // P, Z // P
        tx, ty, tz = ((X - ix * P) << 16) // P, ((Y - iy * P) << 16) // P, ((Z - iz * P) << 16) // P
        fx, fy, fz = _fade_q16(tx), _fade_q16(ty), _fade_q16(tz)

        def corner(cx, cy, cz):
            h = _hash32(ix + cx, iy + cy, iz + cz, seed + 1013 * o)
            # 12 edge directions of the cube, picked by the hash (Perlin's improved-noise set)
            k = h % 12
            dx, dy, dz = tx - (cx << 16), ty - (cy << 16), tz - (cz << 16)
            u = B.where(k < 8, dx, dy)          # first component
            v = B.where(k < 4, dy, dz)          # second component
            su = B.where((k & 1) == 0, u, -u)
            sv = B.where((k & 2) == 0, v, -v)
            return su + sv

        def lerp(a, b, f):
            return a + (((b - a) * f) >> 16)

        x00 = lerp(corner(0, 0, 0), corner(1, 0, 0), fx)
        x10 = lerp(corner(0, 1, 0), corner(1, 1, 0), fx)
        x01 = lerp(corner(0, 0, 1), corner(1, 0, 1), fx)
        x11 = lerp(corner(0, 1, 1), corner(1, 1, 1), fx)
        val = lerp(lerp(x00, x10, fy), lerp(x01, x11, fy), fz) >> o
        total = val if total is None else total + val
    # total is Q16 in about [-1.75, 1.75]: map to 0..255 around 128
    out = B.clip(128 + ((total * 110) >> 16), 0, 255)
    return B.astype(out, B.u8)
