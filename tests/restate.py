"""Python restatements of the reference's projection control flow (txx:439-474 and the two compiled-out branches) in plain
IEEE doubles over the two primitives tests/test_oracle.py pins (I5 interpolation, I6 gradient).  TEST INFRASTRUCTURE ONLY:
an independent second statement that the C++ oracle -- and, through an arbitrary value function, the host walk of the
drop-in filter -- is held to."""
import numpy as np


def f32(v):
    return float(np.float32(v))


def py_normal(oracle, vol, p):
    n = vol.shape[::-1]
    lo, hi, d = [], [], []
    for k in range(3):
        b = np.floor(p[k])
        d.append(p[k] - b)
        lo.append(int(min(max(b, 0), n[k] - 1)))
        hi.append(int(min(max(b + 1, 0), n[k] - 1)))
    acc, total = [0.0, 0.0, 0.0], 0.0
    for counter in range(8):
        overlap, ni = 1.0, []
        for k in range(3):
            if counter & (1 << k):
                ni.append(hi[k]); overlap *= d[k]
            else:
                ni.append(lo[k]); overlap *= 1.0 - d[k]
        if overlap:
            g = oracle.gradient_at_index(vol, tuple(ni))
            for k in range(3):
                acc[k] += overlap * float(g[k])
            total += overlap
        if total == 1.0:
            break
    nrm = [f32(a) for a in acc]
    norm = float(np.sqrt(np.float64(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2])))
    with np.errstate(all="ignore"):
        return [f32(np.float64(v) / np.float64(norm)) for v in nrm]


def py_default_walk(oracle, vol, value_fn, iso, v, thr, step, relax, max_steps):
    """txx:439-474: normal from the gradient image of `vol` (txx:451-452), value from value_fn(point) -- the
    interpolator's Evaluate (txx:455), whatever it computes.  Returns (final vertex, loop passes)."""
    v = [float(c) for c in v]
    number_of_steps, passes = 0, 0
    while True:
        passes += 1
        nrm = py_normal(oracle, vol, v)
        value = value_fn(tuple(v))
        if abs(value - iso) < thr:                              # txx:456-460
            break
        sign = 1.0 if value < iso else -1.0                     # txx:463
        with np.errstate(all="ignore"):
            v = [f32(v[k] + (nrm[k] * sign * step)) for k in range(3)]   # txx:464-467
        step *= relax                                           # txx:468
        number_of_steps += 1
        if number_of_steps - 1 > max_steps:                     # txx:469: numberOfSteps++ > max
            break
    return v, passes


def split_quads(points, quads):
    """txx:286-321 on flat buffers: two triangles per quad along the shorter diagonal (squared distances in double from the
    float coordinates), ties to the first form."""
    p = points.astype(np.float64)
    d02 = ((p[quads[:, 2]] - p[quads[:, 0]]) ** 2)
    d13 = ((p[quads[:, 3]] - p[quads[:, 1]]) ** 2)
    d02 = (d02[:, 0] + d02[:, 1]) + d02[:, 2]
    d13 = (d13[:, 0] + d13[:, 1]) + d13[:, 2]
    first = d02 >= d13
    a = np.stack([quads[:, 0], quads[:, 1], quads[:, 3], quads[:, 1], quads[:, 2], quads[:, 3]], 1)
    b = np.stack([quads[:, 0], quads[:, 1], quads[:, 2], quads[:, 0], quads[:, 2], quads[:, 3]], 1)
    return np.where(first[:, None], a, b).reshape(-1, 3)


def blend_field(n=14):
    """The float volume, and the second (smoothed) one, of the host-walk tests; nothing here depends on libm."""
    z, y, x = np.meshgrid(*(np.arange(n, dtype=np.float64),) * 3, indexing="ij")
    c = (n - 1) / 2
    r2 = (x - c - 0.25) ** 2 + (y - c - 0.125) ** 2 + (z - c + 0.3) ** 2
    vol = (((n * 0.33) ** 2 - r2) / n + 0.2 * ((x * 7 + y * 3 + z * 5) % 4 - 1.5)).astype(np.float32)
    pad = np.pad(vol.astype(np.float64), 1, mode="edge")
    smooth = np.zeros_like(pad[1:-1, 1:-1, 1:-1])
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                smooth += pad[dz:dz + n, dy:dy + n, dx:dx + n]
    return vol, (smooth / 27.0).astype(np.float32)


def blend_value(oracle, vol, smooth):
    """Evaluate() of itk/tests/host_walk.cxx's BlendInterpolator: 0.25 * linear(image) + 0.75 * linear(second image)."""
    return lambda p: 0.25 * oracle.interpolate(vol, p) + 0.75 * oracle.interpolate(smooth, p)
