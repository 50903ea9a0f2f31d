"""oracle/indep/second_source.py — SECOND SOURCES of geometric primitives in double precision.        *** TEST INFRASTRUCTURE ***

Independent derivations (numpy f64, no wt/ header, no line of the restated algorithms) of quantities the physics headers compute, for
tests/test_second_source.py.  Where the other risky primitives have theirs: Fresnel coefficients and the Mueller matrix of a diagonal Jones
matrix — tests/test_kat.py (complex closed forms, Kronecker construction); UTD Ds / Dh — tests/test_kat_utd.py (Sommerfeld's exact half-plane
solution); the Fraunhofer edge sum (alpha_1 / alpha_2 / Psi / ASF, fsd.hpp:65-146) — fraunhofer_boundary_integral and polygon_fourier_integral
below (quadrature of the integrals the closed forms are the antiderivatives of).

cone_tri_min_z: the closest distance along the axis at which an elliptic cone  x^2 + (e y)^2 <= (z tan_alpha + x0)^2  meets a triangle
inside a z-slab (what intersect_cone_tri returns; the reference: include/wt/math/intersect/cone.hpp:550-626 via cone-plane and cone-edge
intersections in 3-D).  Here: a small convex programme in the triangle's barycentric plane — minimise the linear function z(u, v) over
{u, v >= 0, u + v <= 1} ∩ {g(u, v) <= 0} ∩ {zmin <= z <= zmax}, g the cone's quadratic form restricted to the plane — solved by enumerating
the KKT candidates: vertices, edge/cone and edge/slab crossings, the stationary points of z on g = 0 (Lagrange), and the slab's near plane."""
import numpy as np


def _quad_roots(a, b, c):
    if abs(a) < 1e-300:
        return [] if abs(b) < 1e-300 else [-c / b]
    D = b * b - 4 * a * c
    if D < 0:
        return []
    s = np.sqrt(D)
    q = -0.5 * (b + (s if b >= 0 else -s))
    r = [q / a]
    if abs(q) > 1e-300:
        r.append(c / q)
    return r


def cone_tri_min_z(P, tan_alpha, x0, e, zmin, zmax):
    """P: 3 x 3 triangle vertices in the cone's local frame (z along the axis, x the major axis).  Returns the minimal z or None."""
    P = np.asarray(P, np.float64)
    scale = max(1.0, float(np.abs(P).max()))
    A0, E1, E2 = P[0], P[1] - P[0], P[2] - P[0]
    w = np.array([1.0, e, 0.0])

    def point(u, v):
        return A0 + u * E1 + v * E2

    def g_of(p):
        r = p[2] * tan_alpha + x0
        return p[0] ** 2 + (e * p[1]) ** 2 - r * r

    def feasible(u, v, tol=1e-9):
        if u < -tol or v < -tol or u + v > 1 + tol:
            return None
        p = point(u, v)
        r = p[2] * tan_alpha + x0
        if r < -tol * scale or p[2] < zmin - tol * scale or p[2] > zmax + tol * scale:
            return None
        if g_of(p) > 1e-9 * scale * scale:
            return None
        return p[2]

    cands = []
    # vertices
    for (u, v) in ((0, 0), (1, 0), (0, 1)):
        cands.append((u, v))
    # edges: (u, v) = q0 + t dq
    for q0, dq in (((0.0, 0.0), (1.0, 0.0)), ((0.0, 0.0), (0.0, 1.0)), ((1.0, 0.0), (-1.0, 1.0))):
        a = point(*q0)
        d = point(q0[0] + dq[0], q0[1] + dq[1]) - a
        # g(a + t d) = (ax + t dx)^2 + e^2 (ay + t dy)^2 - ((az + t dz) ta + x0)^2
        ra, rd = a[2] * tan_alpha + x0, d[2] * tan_alpha
        qa = d[0] ** 2 + (e * d[1]) ** 2 - rd * rd
        qb = 2 * (a[0] * d[0] + e * e * a[1] * d[1] - ra * rd)
        qc = a[0] ** 2 + (e * a[1]) ** 2 - ra * ra
        ts = _quad_roots(qa, qb, qc)
        for zz in (zmin, zmax):
            if np.isfinite(zz) and abs(d[2]) > 1e-300:
                ts.append((zz - a[2]) / d[2])
        for t in ts:
            if -1e-12 <= t <= 1 + 1e-12:
                cands.append((q0[0] + t * dq[0], q0[1] + t * dq[1]))
    # g restricted to the plane: g(q) = q^T H q + 2 h^T q + c0
    def quad_form(Ea, Eb):
        return Ea[0] * Eb[0] + e * e * Ea[1] * Eb[1] - (Ea[2] * tan_alpha) * (Eb[2] * tan_alpha)
    r0 = A0[2] * tan_alpha + x0
    H = np.array([[quad_form(E1, E1), quad_form(E1, E2)], [quad_form(E1, E2), quad_form(E2, E2)]])
    h = np.array([A0[0] * E1[0] + e * e * A0[1] * E1[1] - r0 * E1[2] * tan_alpha, A0[0] * E2[0] + e * e * A0[1] * E2[1] - r0 * E2[2] * tan_alpha])
    c0 = A0[0] ** 2 + (e * A0[1]) ** 2 - r0 * r0
    zg = np.array([E1[2], E2[2]])
    # Lagrange: H q + h = mu zg, g(q) = 0
    if abs(np.linalg.det(H)) > 1e-14 * (np.abs(H).max() ** 2 + 1e-300) and np.abs(zg).max() > 1e-300:
        Hi = np.linalg.inv(H)
        qh, qz = -Hi @ h, Hi @ zg                      # q = qh + mu qz
        a2 = qz @ H @ qz
        b2 = 2 * (qh @ H @ qz + h @ qz)
        c2 = qh @ H @ qh + 2 * h @ qh + c0
        for mu in _quad_roots(a2, b2, c2):
            q = qh + mu * qz
            cands.append((q[0], q[1]))
    best = None
    for (u, v) in cands:
        z = feasible(u, v)
        if z is not None and (best is None or z < best):
            best = z
    # the slab's near plane: the triangle's section at z = zmin may cross the cone's disk although no candidate above lies on it
    if np.isfinite(zmin) and np.abs(zg).max() > 1e-300 and (best is None or best > zmin):
        # points of the (u, v) triangle with z = zmin: a segment; g along it is a quadratic
        pts = []
        for q0, dq in (((0.0, 0.0), (1.0, 0.0)), ((0.0, 0.0), (0.0, 1.0)), ((1.0, 0.0), (-1.0, 1.0))):
            za = point(*q0)[2]
            zb = point(q0[0] + dq[0], q0[1] + dq[1])[2]
            if abs(zb - za) > 1e-300:
                t = (zmin - za) / (zb - za)
                if -1e-12 <= t <= 1 + 1e-12:
                    pts.append(np.array([q0[0] + t * dq[0], q0[1] + t * dq[1]]))
        if len(pts) >= 2:
            qa_, qb_ = pts[0], pts[-1]
            for cand in pts[1:]:
                if np.linalg.norm(cand - qa_) > np.linalg.norm(qb_ - qa_):
                    qb_ = cand
            dq = qb_ - qa_
            a3 = dq @ H @ dq
            b3 = 2 * (qa_ @ H @ dq + h @ dq)
            c3 = qa_ @ H @ qa_ + 2 * h @ qa_ + c0
            ts = [0.0, 1.0]
            if abs(a3) > 1e-300:
                ts.append(min(1.0, max(0.0, -b3 / (2 * a3))))
            gmin = min(a3 * t * t + b3 * t + c3 for t in ts)
            if gmin <= 1e-9 * scale * scale and zmin * tan_alpha + x0 >= 0:
                best = zmin
    return best


def cone_contains(p, tan_alpha, x0, e, zmin, zmax):
    """A point (local frame) inside the cone within the slab."""
    r = p[2] * tan_alpha + x0
    return zmin <= p[2] <= zmax and r >= 0 and p[0] ** 2 + (e * p[1]) ** 2 <= r * r


# ---- Fraunhofer aperture: the edge sum of fsd.hpp:65-146 against the integrals it is the closed form of -------------------------------------
_GL_X, _GL_W = np.polynomial.legendre.leggauss(64)
_GL_X, _GL_W = (_GL_X + 1) / 2, _GL_W / 2


def fraunhofer_boundary_integral(segments, xi):
    """The far-field amplitude of an aperture given by boundary segments, as the LINE integral Stokes' theorem turns the Fourier integral of
    the aperture into:  B(xi) = sum_j (xi x e_j) / |xi|^2  *  int_0^1 c_j(t) exp(-i xi . (a_j + t e_j)) dt,  c_j(t) = ca_j + t (cb_j - ca_j)
    the (linearly interpolated) field amplitude along segment j from a_j to a_j + e_j.  64-point Gauss-Legendre per segment, f64.  The
    reference evaluates the same integral in closed form per segment — alpha_2 (the sinc term) for the mean amplitude, alpha_1 (its
    derivative) for the slope, times |e|^2 and the phase of the segment's midpoint (fsd.hpp:65-121) — and squares the sum: ASF = |B|^2 / (2 pi)^2.
    segments: iterable of (a[2], e[2], ca, cb)."""
    xi = np.asarray(xi, np.float64)
    B = 0j
    for a, e, ca, cb in segments:
        a, e = np.asarray(a, np.float64), np.asarray(e, np.float64)
        ph = np.exp(-1j * ((a[0] + _GL_X * e[0]) * xi[0] + (a[1] + _GL_X * e[1]) * xi[1]))
        B += (xi[0] * e[1] - xi[1] * e[0]) / (xi[0] ** 2 + xi[1] ** 2) * np.sum(_GL_W * (ca + _GL_X * (cb - ca)) * ph)
    return B


def polygon_fourier_integral(P, xi, n=40):
    """int_P exp(-i xi . x) d^2x over a simple polygon (vertices P[n,2]) by fan triangulation from P[0] and a Duffy-transformed tensor
    Gauss-Legendre rule per triangle (signed areas: any simple polygon).  The physics behind the edge sum: for a uniformly lit aperture
    |B(xi)| equals |this| — the Fraunhofer pattern IS the Fourier transform of the aperture."""
    P = np.asarray(P, np.float64)
    xi = np.asarray(xi, np.float64)
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = (g + 1) / 2, w / 2
    U, V = np.meshgrid(g, g, indexing="ij")
    W = np.outer(w, w) * (1 - U)
    S, T = U, V * (1 - U)
    tot = 0j
    for i in range(1, len(P) - 1):
        a, b, c = P[0], P[i], P[i + 1]
        J = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])
        X = a[0] + S * (b[0] - a[0]) + T * (c[0] - a[0])
        Y = a[1] + S * (b[1] - a[1]) + T * (c[1] - a[1])
        tot += J * np.sum(W * np.exp(-1j * (xi[0] * X + xi[1] * Y)))
    return tot
