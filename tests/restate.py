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


# ---- USE_GRADIENT_RECURSIVE_GAUSSIAN (h:21; txx:488-491): ITK's recursive Gaussian gradient, second restatement -------------

def _deriche(sigma, spacing, order):
    """Coefficients of ITK's RecursiveGaussianImageFilter (Deriche, fourth order) for smoothing (order 0) or the first
    derivative (order 1, NormalizeAcrossScale on); numpy float64 throughout."""
    f = np.float64
    A1, B1, W1, L1 = (f(1.3530), f(-0.6724)), (f(1.8151), f(-3.4327)), f(0.6681), f(-1.3932)
    A2, B2, W2, L2 = (f(-0.3531), f(0.6724)), (f(0.0902), f(0.6100)), f(2.0787), f(-1.3732)
    sd = f(sigma) / f(spacing)
    c1, c2, s1, s2 = np.cos(W1 / sd), np.cos(W2 / sd), np.sin(W1 / sd), np.sin(W2 / sd)
    e1, e2 = np.exp(L1 / sd), np.exp(L2 / sd)
    D4 = e1 * e1 * e2 * e2
    D3 = f(-2) * c1 * e1 * e2 * e2
    D3 = D3 + f(-2) * c2 * e2 * e1 * e1
    D2 = f(4) * c2 * c1 * e1 * e2
    D2 = D2 + (e1 * e1 + e2 * e2)
    D1 = f(-2) * (e2 * c2 + e1 * c1)
    SD = f(1.0) + D1 + D2 + D3 + D4
    DD = D1 + f(2) * D2 + f(3) * D3 + f(4) * D4
    a1, b1, a2, b2 = A1[order], B1[order], A2[order], B2[order]
    N0 = a1 + a2
    N1 = e2 * (b2 * s2 - (a2 + f(2) * a1) * c2)
    N1 = N1 + e1 * (b1 * s1 - (a1 + f(2) * a2) * c1)
    N2 = (a1 + a2) * c2 * c1
    N2 = N2 - (b1 * c2 * s1 + b2 * c1 * s2)
    N2 = N2 * (f(2) * e1 * e2)
    N2 = N2 + (a2 * e1 * e1 + a1 * e2 * e2)
    N3 = e2 * e1 * e1 * (b2 * s2 - a2 * c2)
    N3 = N3 + e1 * e2 * e2 * (b1 * s1 - a1 * c1)
    SN = N0 + N1 + N2 + N3
    DN = N1 + f(2) * N2 + f(3) * N3
    if order == 0:
        alpha = f(2) * SN / SD - N0
        N = [N0 / alpha, N1 / alpha, N2 / alpha, N3 / alpha]
        M = [N[1] - D1 * N[0], N[2] - D2 * N[0], N[3] - D3 * N[0], -D4 * N[0]]
    else:
        alpha = f(2) * (SN * DD - DN * SD) / (SD * SD)
        alpha = alpha * f(1.0)
        k = f(sigma) / alpha
        N = [N0 * k, N1 * k, N2 * k, N3 * k]
        M = [-(N[1] - D1 * N[0]), -(N[2] - D2 * N[0]), -(N[3] - D3 * N[0]), D4 * N[0]]
    D = [D1, D2, D3, D4]
    sn, sm, sdd = N[0] + N[1] + N[2] + N[3], M[0] + M[1] + M[2] + M[3], f(1.0) + D1 + D2 + D3 + D4
    BN = [d * sn / sdd for d in D]
    BM = [d * sm / sdd for d in D]
    return N, D, M, BN, BM


def _deriche_axis(vol64, axis, sigma, spacing, order):
    """One separable pass of the recursive filter along `axis` of a float64 array [z, y, x] (axis: 0 = x), all lines at
    once; returns float32 (the filter's images between the passes are float)."""
    N, D, M, BN, BM = _deriche(sigma, spacing, order)
    d = np.moveaxis(vol64, 2 - axis, 0)                      # the filtered axis first
    ln = d.shape[0]
    s = np.empty_like(d)
    v1 = d[0]
    s[0] = v1 * N[0] + v1 * N[1] + v1 * N[2] + v1 * N[3]
    s[1] = d[1] * N[0] + v1 * N[1] + v1 * N[2] + v1 * N[3]
    s[2] = d[2] * N[0] + d[1] * N[1] + v1 * N[2] + v1 * N[3]
    s[3] = d[3] * N[0] + d[2] * N[1] + d[1] * N[2] + v1 * N[3]
    s[0] -= v1 * BN[0] + v1 * BN[1] + v1 * BN[2] + v1 * BN[3]
    s[1] -= s[0] * D[0] + v1 * BN[1] + v1 * BN[2] + v1 * BN[3]
    s[2] -= s[1] * D[0] + s[0] * D[1] + v1 * BN[2] + v1 * BN[3]
    s[3] -= s[2] * D[0] + s[1] * D[1] + s[0] * D[2] + v1 * BN[3]
    for i in range(4, ln):
        s[i] = d[i] * N[0] + d[i - 1] * N[1] + d[i - 2] * N[2] + d[i - 3] * N[3]
        s[i] -= s[i - 1] * D[0] + s[i - 2] * D[1] + s[i - 3] * D[2] + s[i - 4] * D[3]
    out = s.copy()
    v2 = d[ln - 1]
    s[ln - 1] = v2 * M[0] + v2 * M[1] + v2 * M[2] + v2 * M[3]
    s[ln - 2] = d[ln - 1] * M[0] + v2 * M[1] + v2 * M[2] + v2 * M[3]
    s[ln - 3] = d[ln - 2] * M[0] + d[ln - 1] * M[1] + v2 * M[2] + v2 * M[3]
    s[ln - 4] = d[ln - 3] * M[0] + d[ln - 2] * M[1] + d[ln - 1] * M[2] + v2 * M[3]
    s[ln - 1] -= v2 * BM[0] + v2 * BM[1] + v2 * BM[2] + v2 * BM[3]
    s[ln - 2] -= s[ln - 1] * D[0] + v2 * BM[1] + v2 * BM[2] + v2 * BM[3]
    s[ln - 3] -= s[ln - 2] * D[0] + s[ln - 1] * D[1] + v2 * BM[2] + v2 * BM[3]
    s[ln - 4] -= s[ln - 3] * D[0] + s[ln - 2] * D[1] + s[ln - 1] * D[2] + v2 * BM[3]
    for i in range(ln - 4, 0, -1):
        s[i - 1] = d[i] * M[0] + d[i + 1] * M[1] + d[i + 2] * M[2] + d[i + 3] * M[3]
        s[i - 1] -= s[i] * D[0] + s[i + 1] * D[1] + s[i + 2] * D[2] + s[i + 3] * D[3]
    out += s
    return np.moveaxis(out.astype(np.float32), 0, 2 - axis)


def recursive_gaussian_gradient(vol, spacing=(1.0, 1.0, 1.0)):
    """itk::GradientRecursiveGaussianImageFilter with sigma = max spacing (txx:489) on a [z, y, x] volume with the
    identity direction: float64 [z, y, x, 3].  Per component: derivative along its axis first, then smoothing along the
    other axes in ascending order, float32 images in between, divided by the spacing at the end."""
    sigma = max(spacing)
    grad = np.empty(vol.shape + (3,), dtype=np.float64)
    for dim in range(3):
        img = _deriche_axis(vol.astype(np.float64), dim, sigma, spacing[dim], 1)
        for ax in range(3):
            if ax != dim:
                img = _deriche_axis(img.astype(np.float64), ax, sigma, spacing[ax], 0)
        grad[..., dim] = img.astype(np.float64) / np.float64(spacing[dim])
    return grad


def py_normal_from_gradient_image(grad, p, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0)):
    """VectorLinearInterpolate of a double gradient image at physical point p + Normalize(), all in double (the
    recursive-Gaussian variant's GradientPixelType is CovariantVector<double,3>); identity direction."""
    n = grad.shape[2::-1]
    lo, hi, d = [], [], []
    for k in range(3):
        ci = (p[k] - origin[k]) * (1.0 / spacing[k]) if spacing[k] != 1.0 else (p[k] - origin[k])
        b = np.floor(ci)
        d.append(ci - b)
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
            g = grad[ni[2], ni[1], ni[0]]
            for k in range(3):
                acc[k] += overlap * float(g[k])
            total += overlap
        if total == 1.0:
            break
    norm = float(np.sqrt(np.float64(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2])))
    with np.errstate(all="ignore"):
        return [float(np.float64(a) / np.float64(norm)) for a in acc]


def py_default_walk_normal_fn(normal_fn, value_fn, iso, v, thr, step, relax, max_steps):
    """txx:439-474 with the normal from normal_fn(point) (a double vector) instead of the float gradient image."""
    v = [float(c) for c in v]
    number_of_steps, passes = 0, 0
    while True:
        passes += 1
        nrm = normal_fn(tuple(v))
        value = value_fn(tuple(v))
        if abs(value - iso) < thr:
            break
        sign = 1.0 if value < iso else -1.0
        with np.errstate(all="ignore"):
            v = [f32(v[k] + (nrm[k] * sign * step)) for k in range(3)]
        step *= relax
        number_of_steps += 1
        if number_of_steps - 1 > max_steps:
            break
    return v, passes
