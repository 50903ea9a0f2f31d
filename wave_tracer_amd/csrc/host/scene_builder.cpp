// wave_tracer_amd — host-side scene baking (see scene_builder.h for the reference map).
#include "scene_builder.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <complex>
#include <cstring>
#include <functional>
#include <map>
#include <numeric>
#include <queue>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "../wt/beam.h"
#include "../wt/fsd.h"
#include "spectra_data.h"

namespace wth {
using namespace wt;

static inline dvec3 operator+(dvec3 a, dvec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline dvec3 operator-(dvec3 a, dvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline dvec3 operator*(dvec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline double ddot(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline dvec3 dcross(dvec3 a, dvec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline double dlen(dvec3 a) { return std::sqrt(ddot(a, a)); }
static inline dvec3 dnorm(dvec3 a) { return a * (1.0 / dlen(a)); }
static inline vec3 tof(dvec3 a) { return vec3{(float)a.x, (float)a.y, (float)a.z}; }

// ---------------------------------------------------------------------------------------------------------
// transforms
xform_t xform_t::identity() {
    xform_t t{};
    for (int i = 0; i < 16; ++i) t.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return t;
}
xform_t xform_t::translate(double x, double y, double z) {
    xform_t t = identity();
    t.m[3] = x;
    t.m[7] = y;
    t.m[11] = z;
    return t;
}
xform_t xform_t::scale(double x, double y, double z) {
    xform_t t = identity();
    t.m[0] = x;
    t.m[5] = y;
    t.m[10] = z;
    return t;
}
xform_t xform_t::rotate(double ax, double ay, double az, double a) {
    const dvec3 d = dnorm({ax, ay, az});
    const double c = std::cos(a), s = std::sin(a), x = d.x, y = d.y, z = d.z;
    xform_t t = identity();
    t.m[0] = c + x * x * (1 - c);
    t.m[1] = x * y * (1 - c) - z * s;
    t.m[2] = x * z * (1 - c) + y * s;
    t.m[4] = y * x * (1 - c) + z * s;
    t.m[5] = c + y * y * (1 - c);
    t.m[6] = y * z * (1 - c) - x * s;
    t.m[8] = z * x * (1 - c) - y * s;
    t.m[9] = z * y * (1 - c) + x * s;
    t.m[10] = c + z * z * (1 - c);
    return t;
}
// include/wt/math/transform/transform.hpp:198-213: columns = (l, u, d, origin)
xform_t xform_t::lookat(dvec3 origin, dvec3 target, dvec3 up) {
    const dvec3 d = dnorm(target - origin);
    const dvec3 l = dnorm(dcross(up, d));
    const dvec3 u = dcross(d, l);
    xform_t t = identity();
    t.m[0] = l.x; t.m[1] = u.x; t.m[2] = d.x; t.m[3] = origin.x;
    t.m[4] = l.y; t.m[5] = u.y; t.m[6] = d.y; t.m[7] = origin.y;
    t.m[8] = l.z; t.m[9] = u.z; t.m[10] = d.z; t.m[11] = origin.z;
    return t;
}
xform_t xform_t::from_rows(const double r[16]) {
    xform_t t;
    std::memcpy(t.m, r, sizeof(t.m));
    return t;
}
xform_t xform_t::operator*(const xform_t& o) const {
    xform_t r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += m[i * 4 + k] * o.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
dvec3 xform_t::point(dvec3 p) const {
    const double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    const double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    const double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    const double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    return {x / w, y / w, z / w};
}
dvec3 xform_t::vector(dvec3 v) const {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
dvec3 xform_t::normal(dvec3 n) const {
    // inverse transpose of the upper 3x3 = cofactor matrix / det
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const dvec3 r{(e * i - f * h) * n.x + (f * g - d * i) * n.y + (d * h - e * g) * n.z,
                  (c * h - b * i) * n.x + (a * i - c * g) * n.y + (b * g - a * h) * n.z,
                  (b * f - c * e) * n.x + (c * d - a * f) * n.y + (a * e - b * d) * n.z};
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    return dnorm(r * (det < 0 ? -1.0 : 1.0));
}

static bool invert4(const double* a, double* inv) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = a[i * 4 + j];
            m[i][j + 4] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(m[r][c]) > std::fabs(m[p][c])) p = r;
        if (m[p][c] == 0) return false;
        if (p != c)
            for (int j = 0; j < 8; ++j) std::swap(m[p][j], m[c][j]);
        const double d = 1.0 / m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] *= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = m[r][c];
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][j + 4];
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// procedural meshes (src/mesh/*.cpp)
mesh_t mesh_rectangle(dvec3 p, dvec3 x, dvec3 y) {   // rectangle.cpp:23-67 with tessellation 1
    mesh_t m;
    m.verts = {p, p + x, p + x + y, p + y};
    m.uvs = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    m.tris = {{0, 1, 2}, {2, 3, 0}};
    return m;
}
mesh_t mesh_rectangle_scaled(double s) { return mesh_rectangle({-s / 2, -s / 2, 0}, {s, 0, 0}, {0, s, 0}); }
mesh_t mesh_cube(double length) {   // cube.cpp
    static const double pos[24][3] = {{1, -1, -1}, {1, -1, 1},  {-1, -1, 1}, {-1, -1, -1}, {1, 1, -1},  {-1, 1, -1}, {-1, 1, 1},  {1, 1, 1},
                                      {1, -1, -1}, {1, 1, -1},  {1, 1, 1},   {1, -1, 1},   {1, -1, 1},  {1, 1, 1},   {-1, 1, 1},  {-1, -1, 1},
                                      {-1, -1, 1}, {-1, 1, 1},  {-1, 1, -1}, {-1, -1, -1}, {1, 1, -1},  {1, -1, -1}, {-1, -1, -1}, {-1, 1, -1}};
    static const double ns[6][3] = {{0, -1, 0}, {0, 1, 0}, {1, 0, 0}, {0, 0, 1}, {-1, 0, 0}, {0, 0, -1}};
    static const float uvs[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
    static const uint32_t tids[12][3] = {{0, 1, 2},    {3, 0, 2},    {4, 5, 6},    {7, 4, 6},    {8, 9, 10},   {11, 8, 10},
                                         {12, 13, 14}, {15, 12, 14}, {16, 17, 18}, {19, 16, 18}, {20, 21, 22}, {23, 20, 22}};
    mesh_t m;
    for (int i = 0; i < 24; ++i) {
        m.verts.push_back({pos[i][0] * length / 2, pos[i][1] * length / 2, pos[i][2] * length / 2});
        m.normals.push_back({ns[i / 4][0], ns[i / 4][1], ns[i / 4][2]});
        m.uvs.push_back({uvs[i % 4][0], uvs[i % 4][1]});
    }
    for (auto& t : tids) m.tris.push_back({t[0], t[1], t[2]});
    return m;
}
static void icosahedron(std::vector<dvec3>& v, std::vector<std::array<uint32_t, 3>>& t) {   // icosahedron.cpp
    const double a = 1.0, b = 1.0 / 1.6180339887498948482;
    v = {{0, b, -a}, {b, a, 0}, {-b, a, 0}, {0, b, a}, {0, -b, a}, {-a, 0, b}, {0, -b, -a}, {a, 0, -b}, {a, 0, b}, {-a, 0, -b}, {b, -a, 0}, {-b, -a, 0}};
    for (auto& p : v) p = dnorm(p);
    t = {{2, 1, 0}, {1, 2, 3}, {5, 4, 3},  {4, 8, 3},  {7, 6, 0}, {6, 9, 0}, {11, 10, 4}, {10, 11, 6}, {9, 5, 2},  {5, 9, 11},
         {8, 7, 1}, {7, 8, 10}, {2, 5, 3}, {8, 1, 3},  {9, 2, 0}, {1, 7, 0}, {11, 9, 6},  {7, 10, 6},  {5, 11, 4}, {10, 8, 4}};
}
static void ico_subdivide(dvec3 p0, dvec3 p1, dvec3 p2, int rec, const std::function<void(dvec3, dvec3, dvec3)>& emit) {
    if (rec == 0) {
        emit(dnorm(p0), dnorm(p1), dnorm(p2));
        return;
    }
    const dvec3 p01 = (p0 + p1) * 0.5, p02 = (p0 + p2) * 0.5, p12 = (p1 + p2) * 0.5;
    ico_subdivide(p0, p01, p02, rec - 1, emit);
    ico_subdivide(p01, p1, p12, rec - 1, emit);
    ico_subdivide(p01, p12, p02, rec - 1, emit);
    ico_subdivide(p02, p12, p2, rec - 1, emit);
}
static std::array<float, 2> sphere_uv(dvec3 n) {
    return {(float)(std::atan2(n.z, n.x) * 0.15915494309189535), (float)(std::asin(std::max(-1.0, std::min(1.0, n.y))) / 3.14159265358979 + .5)};
}
mesh_t mesh_sphere(dvec3 centre, double r, int tessellation) {   // sphere.cpp
    std::vector<dvec3> iv;
    std::vector<std::array<uint32_t, 3>> it;
    icosahedron(iv, it);
    const int recursion = (int)(std::max(0.0, std::log2(double(tessellation) / 3.0)) + .5);
    mesh_t m;
    for (auto& t : it)
        ico_subdivide(iv[t[0]], iv[t[1]], iv[t[2]], recursion, [&](dvec3 n0, dvec3 n1, dvec3 n2) {
            const uint32_t i = (uint32_t)m.verts.size();
            const dvec3 ns[3] = {n0, n1, n2};
            for (auto& n : ns) {
                m.verts.push_back(n * r + centre);
                m.normals.push_back(n);
                m.uvs.push_back(sphere_uv(n));
            }
            m.tris.push_back({i, i + 1, i + 2});
        });
    return m;
}
// stand-in for the LFS-missing scanned meshes (dragon / bunny): a radially perturbed ico-sphere ("blob")
mesh_t mesh_blob(double r, int recursion, double bump, int freq, uint32_t seed) {
    std::vector<dvec3> iv;
    std::vector<std::array<uint32_t, 3>> it;
    icosahedron(iv, it);
    const double ph0 = (seed % 97) * 0.13, ph1 = (seed % 89) * 0.29, ph2 = (seed % 83) * 0.41;
    auto P = [&](dvec3 n) {
        const double s = std::sin(freq * n.x + ph0) * std::sin(freq * n.y + ph1) * std::sin(freq * n.z + ph2) +
                         0.5 * std::sin(2.3 * freq * n.x + ph1) * std::cos(1.7 * freq * n.y + ph2);
        return n * (r * (1.0 + bump * s));
    };
    auto N = [&](dvec3 n) {
        // numerical normal of the perturbed surface
        const dvec3 ax = std::fabs(n.x) > 0.9 ? dvec3{0, 1, 0} : dvec3{1, 0, 0};
        const dvec3 t = dnorm(dcross(ax, n)), b = dcross(n, t);
        const double h = 1e-4;
        const dvec3 du = P(dnorm(n + t * h)) - P(dnorm(n - t * h));
        const dvec3 dv = P(dnorm(n + b * h)) - P(dnorm(n - b * h));
        dvec3 nn = dnorm(dcross(du, dv));
        if (ddot(nn, n) < 0) nn = nn * -1.0;
        return nn;
    };
    mesh_t m;
    for (auto& t : it)
        ico_subdivide(iv[t[0]], iv[t[1]], iv[t[2]], recursion, [&](dvec3 n0, dvec3 n1, dvec3 n2) {
            const uint32_t i = (uint32_t)m.verts.size();
            const dvec3 ns[3] = {n0, n1, n2};
            for (auto& n : ns) {
                m.verts.push_back(P(n));
                m.normals.push_back(N(n));
                m.uvs.push_back(sphere_uv(n));
            }
            m.tris.push_back({i, i + 1, i + 2});
        });
    return m;
}
mesh_t mesh_blob_geodesic(double r, int n, double bump, int freq, uint32_t seed) {
    std::vector<dvec3> iv;
    std::vector<std::array<uint32_t, 3>> it;
    icosahedron(iv, it);
    const double ph0 = (seed % 97) * 0.13, ph1 = (seed % 89) * 0.29, ph2 = (seed % 83) * 0.41;
    auto P = [&](dvec3 d) {
        const double s = std::sin(freq * d.x + ph0) * std::sin(freq * d.y + ph1) * std::sin(freq * d.z + ph2) +
                         0.5 * std::sin(2.3 * freq * d.x + ph1) * std::cos(1.7 * freq * d.y + ph2);
        return d * (r * (1.0 + bump * s));
    };
    auto N = [&](dvec3 d) {
        const dvec3 ax = std::fabs(d.x) > 0.9 ? dvec3{0, 1, 0} : dvec3{1, 0, 0};
        const dvec3 t = dnorm(dcross(ax, d)), b = dcross(d, t);
        const double h = 1e-4;
        dvec3 nn = dnorm(dcross(P(dnorm(d + t * h)) - P(dnorm(d - t * h)), P(dnorm(d + b * h)) - P(dnorm(d - b * h))));
        if (ddot(nn, d) < 0) nn = nn * -1.0;
        return nn;
    };
    mesh_t m;
    for (auto& t : it) {
        const dvec3 A = iv[t[0]], B = iv[t[1]], C = iv[t[2]];
        auto at = [&](int i, int j) { return dnorm(A * (double(n - i - j) / n) + B * (double(i) / n) + C * (double(j) / n)); };
        auto put = [&](dvec3 d0, dvec3 d1, dvec3 d2) {
            const uint32_t k = (uint32_t)m.verts.size();
            const dvec3 ds[3] = {d0, d1, d2};
            for (auto& d : ds) {
                m.verts.push_back(P(d));
                m.normals.push_back(N(d));
                m.uvs.push_back(sphere_uv(d));
            }
            m.tris.push_back({k, k + 1, k + 2});
        };
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n - i; ++j) {
                put(at(i, j), at(i + 1, j), at(i, j + 1));
                if (i + j < n - 1) put(at(i + 1, j), at(i + 1, j + 1), at(i, j + 1));
            }
    }
    return m;
}
mesh_t mesh_star_plate(double R, double thickness, int s) {
    mesh_t m;
    const double ri = R / std::sqrt(3.0);
    dvec3 ring[12];
    for (int k = 0; k < 12; ++k) {
        const double a = M_PI / 6 * k, rr = (k % 2 == 0) ? R : ri;
        ring[k] = {0, rr * std::cos(a), rr * std::sin(a)};
    }
    const double hx = thickness / 2;
    auto face_tri = [&](dvec3 a, dvec3 b, dvec3 c, double x, bool flip) {   // s^2 sub-triangles
        auto at = [&](int i, int j) { return a * (double(s - i - j) / s) + b * (double(i) / s) + c * (double(j) / s) + dvec3{x, 0, 0}; };
        auto put = [&](dvec3 p0, dvec3 p1, dvec3 p2) {
            const uint32_t k = (uint32_t)m.verts.size();
            m.verts.insert(m.verts.end(), {p0, flip ? p2 : p1, flip ? p1 : p2});
            m.uvs.insert(m.uvs.end(), {{0, 0}, {1, 0}, {0, 1}});
            m.tris.push_back({k, k + 1, k + 2});
        };
        for (int i = 0; i < s; ++i)
            for (int j = 0; j < s - i; ++j) {
                put(at(i, j), at(i + 1, j), at(i, j + 1));
                if (i + j < s - 1) put(at(i + 1, j), at(i + 1, j + 1), at(i, j + 1));
            }
    };
    for (int side = 0; side < 2; ++side) {
        const double x = side ? hx : -hx;
        const bool flip = side == 0;   // the +x face is counter-clockwise seen from +x
        for (int k = 0; k < 12; k += 2) face_tri(ring[(k + 11) % 12], ring[k], ring[(k + 1) % 12], x, flip);          // the six tips
        for (int k = 1; k < 12; k += 2) face_tri({0, 0, 0}, ring[k], ring[(k + 2) % 12], x, flip);                     // the inner hexagon
    }
    for (int k = 0; k < 12; ++k)   // rim: s quads per outline edge, matching the faces' subdivision (no T-junctions)
        for (int i = 0; i < s; ++i) {
            const dvec3 pa = ring[k] * (double(s - i) / s) + ring[(k + 1) % 12] * (double(i) / s);
            const dvec3 pb = ring[k] * (double(s - i - 1) / s) + ring[(k + 1) % 12] * (double(i + 1) / s);
            const dvec3 a0 = pa + dvec3{-hx, 0, 0}, a1 = pa + dvec3{hx, 0, 0}, b0 = pb + dvec3{-hx, 0, 0}, b1 = pb + dvec3{hx, 0, 0};
            const uint32_t q = (uint32_t)m.verts.size();
            m.verts.insert(m.verts.end(), {a0, b0, b1, a1});
            m.uvs.insert(m.uvs.end(), {{0, 0}, {1, 0}, {1, 1}, {0, 1}});
            m.tris.push_back({q, q + 1, q + 2});
            m.tris.push_back({q + 2, q + 3, q});
        }
    return m;
}
mesh_t mesh_cylinder(dvec3 p0, dvec3 p1, double radius, int tess) {
    const dvec3 axis = dnorm(p1 - p0);
    const dvec3 ax = std::fabs(axis.x) > 0.9 ? dvec3{0, 1, 0} : dvec3{1, 0, 0};
    const dvec3 t = dnorm(dcross(ax, axis)), b = dcross(axis, t);
    // src/mesh/cylinder.cpp: an open tube (no caps), two shared vertices per azimuth step
    mesh_t m;
    const uint32_t verts = 2 * (uint32_t)tess;
    for (int i = 0; i < tess; ++i) {
        const double a0 = 2 * M_PI * i / tess;
        const dvec3 n0 = t * std::cos(a0) + b * std::sin(a0);
        m.verts.push_back(p0 + n0 * radius);
        m.verts.push_back(p1 + n0 * radius);
        m.normals.push_back(n0);
        m.normals.push_back(n0);
        m.uvs.push_back({(float)i / tess, 0});
        m.uvs.push_back({(float)i / tess, 1});
        const uint32_t i0 = 2 * (uint32_t)i, i1 = i0 + 1, i2 = (i0 + 2) % verts, i3 = i2 + 1;
        m.tris.push_back({i0, i2, i1});
        m.tris.push_back({i1, i2, i3});
    }
    return m;
}
// The reference's procedural lens (src/mesh/lens.cpp:19-199; box.xml:253-266 "dragon_lens"): two spherical (or planar) faces of
// curvature radius radius/Rk around the +x axis and, when the edge thickness is positive, a cylindrical rim.  Vertex rings sit at
// heights radius * (i/T)^0.8 (lens.cpp:56,83), T rings of T azimuth steps per curved face, one ring for a planar face.
mesh_t mesh_lens(dvec3 centre, double radius, double R1c, double R2c, double thickness, int T) {
    mesh_t m;
    const double inf = std::numeric_limits<double>::infinity();
    const double R1 = R1c != 0 ? radius / R1c : inf, R2 = R2c != 0 ? radius / R2c : inf;
    const bool c1 = std::isfinite(R1), c2 = std::isfinite(R2);
    auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
    const double x1 = c1 ? sgn(R1) * std::sqrt(R1 * R1 - radius * radius) : 0.0;    // centres of curvature on the axis
    const double x2 = c2 ? -sgn(R2) * std::sqrt(R2 * R2 - radius * radius) : 0.0;
    double ET = x1 - x2 - (c1 ? R1 : 0.0) - (c2 ? R2 : 0.0) + thickness;            // edge thickness
    if (thickness == 0 && R1c <= 0 && R2c <= 0) ET += radius / 1000;                 // faces of a double-concave lens must not touch
    struct face_t {
        uint32_t start;
        int rings;
    };
    auto add_face = [&](bool curved, double R, double xc, double side, double shift) {
        face_t f{(uint32_t)m.verts.size(), curved ? T : 1};
        m.verts.push_back({0, 0, 0});   // pole (set by the caller)
        m.normals.push_back({side, 0, 0});
        m.uvs.push_back({0, 0});
        for (int i = 0; i < f.rings; ++i) {
            const double h = radius * std::min(1.0, std::pow((i + 1) / double(f.rings), 0.8));
            for (int j = 0; j < T; ++j) {
                const double phi = 2 * M_PI * j / T;
                const dvec3 cp{0, std::cos(phi) * h, std::sin(phi) * h};
                dvec3 pnt, n{side, 0, 0};
                if (curved) {
                    n = dnorm(cp - dvec3{xc, 0, 0});
                    if (R < 0) n = n * -1.0;
                    pnt = dvec3{xc, 0, 0} + n * R + dvec3{shift, 0, 0};
                } else
                    pnt = cp + dvec3{shift, 0, 0};
                m.verts.push_back(pnt);
                m.normals.push_back(n);
                m.uvs.push_back({(float)(i + 1) / (T + 1), (float)j / T});
            }
        }
        return f;
    };
    const face_t L = add_face(c1, R1, x1, -1.0, 0.0);
    m.verts[L.start] = {x1 - (c1 ? R1 : 0.0), 0, 0};                                  // lens.cpp:51
    const face_t Rf = add_face(c2, R2, x2, +1.0, ET);
    m.verts[Rf.start] = {x2 + (c2 ? R2 : 0.0) + ET, 0, 0};                            // lens.cpp:78
    const uint32_t E = (uint32_t)m.verts.size();
    if (ET > 0)
        for (int j = 0; j < T; ++j) {
            const double phi = 2 * M_PI * j / T;
            const dvec3 n{0, std::cos(phi), std::sin(phi)};
            m.verts.push_back(n * radius);
            m.verts.push_back(n * radius + dvec3{ET, 0, 0});
            m.normals.push_back(n);
            m.normals.push_back(n);
            m.uvs.push_back({0, (float)j / T});
            m.uvs.push_back({1, (float)j / T});
        }
    // fans around the poles, quad strips between rings; the two faces wind oppositely (outward normals)
    auto tri_face = [&](const face_t& f, bool flip) {
        auto put = [&](uint32_t a, uint32_t b, uint32_t c) { m.tris.push_back(flip ? std::array<uint32_t, 3>{a, c, b} : std::array<uint32_t, 3>{a, b, c}); };
        for (int i = 0; i < f.rings; ++i)
            for (int j = 0; j < T; ++j) {
                const int jp = j > 0 ? j - 1 : T - 1;
                const uint32_t ring = f.start + 1 + (uint32_t)i * T, inner = ring - T;
                if (i == 0)
                    put(f.start, ring + j, ring + jp);
                else {
                    put(inner + jp, inner + j, ring + jp);
                    put(ring + jp, inner + j, ring + j);
                }
            }
    };
    tri_face(L, false);
    tri_face(Rf, true);
    if (ET > 0)
        for (int j = 0; j < T; ++j) {
            const uint32_t p0 = E + (j > 0 ? 2 * j - 2 : 2 * T - 2), p1 = p0 + 1, q0 = E + 2 * j, q1 = q0 + 1;
            m.tris.push_back({p1, p0, q0});
            m.tris.push_back({q1, p1, q0});
        }
    for (auto& v : m.verts) v = v + centre;
    return m;
}
// triangular prism: apex angle `angle`, side length `length` (extrusion along z), apex height `height`
mesh_t mesh_prism(double length, double height, double angle) {
    const double hw = height * std::tan(angle / 2);
    const dvec3 A0{-hw, 0, -length / 2}, B0{hw, 0, -length / 2}, C0{0, height, -length / 2};
    const dvec3 A1{-hw, 0, length / 2}, B1{hw, 0, length / 2}, C1{0, height, length / 2};
    mesh_t m;
    auto quad = [&](dvec3 a, dvec3 b, dvec3 c, dvec3 d) {
        const uint32_t k = (uint32_t)m.verts.size();
        m.verts.insert(m.verts.end(), {a, b, c, d});
        m.uvs.insert(m.uvs.end(), {{0, 0}, {1, 0}, {1, 1}, {0, 1}});
        m.tris.push_back({k, k + 1, k + 2});
        m.tris.push_back({k + 2, k + 3, k});
    };
    auto tri = [&](dvec3 a, dvec3 b, dvec3 c) {
        const uint32_t k = (uint32_t)m.verts.size();
        m.verts.insert(m.verts.end(), {a, b, c});
        m.uvs.insert(m.uvs.end(), {{0, 0}, {1, 0}, {.5f, 1}});
        m.tris.push_back({k, k + 1, k + 2});
    };
    quad(A0, A1, B1, B0);   // bottom (normal -y)
    quad(B0, B1, C1, C0);   // right face
    quad(C0, C1, A1, A0);   // left face
    tri(A0, B0, C0);        // -z cap
    tri(A1, C1, B1);        // +z cap
    return m;
}

// ---------------------------------------------------------------------------------------------------------
static inline float wavelen_mm_to_k(float lambda_mm) { return kTwoPi / lambda_mm; }

scene_builder_t::scene_builder_t() {
    std::memset(&sc_, 0, sizeof(sc_));
    sc_.opts.max_depth = 16;
    sc_.opts.MIS = 1;
    sc_.opts.RR = 1;
    sc_.opts.FSD = 1;
    sc_.opts.sensor_direct = 1;
    sc_.opts.emitter_direct = 1;
    sc_.opts.force_ray_tracing = 0;
}

int scene_builder_t::spectrum_const(float re, float im) {
    spectrum_t s{};
    s.type = SPEC_CONST;
    s.c_re = re;
    s.c_im = im;
    s.kmin = 0;
    s.kmax = WT_INF;
    spectra_.push_back(s);
    return (int)spectra_.size() - 1;
}
int scene_builder_t::spectrum_discrete(float wavelength_mm, float value) {
    spectrum_t s{};
    s.type = SPEC_DISCRETE;
    s.kmin = s.kmax = wavelen_mm_to_k(wavelength_mm);
    s.c_re = value;
    spectra_.push_back(s);
    return (int)spectra_.size() - 1;
}
static const int kSpecKnots = 2048;
int scene_builder_t::spectrum_from_wavelength_table(const float* vre, const float* vim, int n, float lmin_nm, float lstep_nm) {
    const float lmax_nm = lmin_nm + lstep_nm * (n - 1);
    spectrum_t s{};
    s.type = SPEC_TABLE;
    s.kmin = wavelen_mm_to_k(lmax_nm * 1e-6f);
    s.kmax = wavelen_mm_to_k(lmin_nm * 1e-6f);
    s.offset = (uint32_t)spectra_data_.size();
    s.count = kSpecKnots;
    s.is_complex = vim ? 1 : 0;
    auto sample = [&](const float* v, double k) {
        const double lnm = (2 * M_PI / k) * 1e6;
        double x = (lnm - lmin_nm) / lstep_nm;
        x = std::max(0.0, std::min(double(n - 1), x));
        const int l = std::min(n - 2, (int)x);
        const double f = x - l;
        return (float)(v[l] * (1 - f) + v[l + 1] * f);
    };
    for (int pass = 0; pass < (vim ? 2 : 1); ++pass)
        for (int i = 0; i < kSpecKnots; ++i) {
            const double k = s.kmin + (double(s.kmax) - s.kmin) * i / (kSpecKnots - 1);
            spectra_data_.push_back(sample(pass == 0 ? vre : vim, k));
        }
    spectra_.push_back(s);
    return (int)spectra_.size() - 1;
}
// spectrum::rgb_t (include/wt/spectrum/rgb.hpp:90-92): RGB_to_spectral::uplift (RGB_to_spectral.hpp:27-84, A. Weidlich's variant of
// Smits' uplift): ten 34-nm bins over 380..720 nm, 0 outside.  Baked on 1-nm steps (the step edges are smoothed by the k-knot
// resampling of SPEC_TABLE spectra).
int scene_builder_t::spectrum_rgb(float r, float g, float bl) {
    static const float white_I[11] = {1.0000f, 1.0000f, 0.9999f, 0.9993f, 0.9992f, 0.9998f, 1.0000f, 1.0000f, 1.0000f, 1.0000f, 0.f};
    static const float cyan_I[11] = {0.9710f, 0.9426f, 1.0007f, 1.0007f, 1.0007f, 1.0007f, 0.1564f, 0.0000f, 0.0000f, 0.0000f, 0.f};
    static const float magenta_I[11] = {1.0000f, 1.0000f, 0.968f, 0.22295f, 0.0000f, 0.0458f, 0.8369f, 1.0000f, 1.0000f, 0.9959f, 0.f};
    static const float yellow_I[11] = {0.0001f, 0.0000f, 0.1088f, 0.6651f, 1.0000f, 1.0000f, 0.9996f, 0.9586f, 0.9685f, 0.9840f, 0.f};
    static const float red_I[11] = {0.1012f, 0.0515f, 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.8325f, 1.0149f, 1.0149f, 1.014f, 0.f};
    static const float green_I[11] = {0.0000f, 0.0000f, 0.0273f, 0.7937f, 1.0000f, 0.9418f, 0.1719f, 0.0000f, 0.0000f, 0.0025f, 0.f};
    static const float blue_I[11] = {1.0000f, 1.0000f, 0.8916f, 0.3323f, 0.0000f, 0.0000f, 0.0003f, 0.0369f, 0.0483f, 0.0496f, 0.f};
    const int N = 341;
    std::vector<float> v(N);
    for (int i = 0; i < N; ++i) {
        const float lambda = 380.f + (float)i;
        const int bin = std::min(10, (int)((lambda - 380.f) / (720.f - 380.f) * 10.f));
        float I = 0.f;
        if (r <= g && r <= bl) {
            I += white_I[bin] * r;
            if (g <= bl) {
                I += cyan_I[bin] * (g - r);
                I += blue_I[bin] * (bl - g);
            } else {
                I += cyan_I[bin] * (bl - r);
                I += green_I[bin] * (g - bl);
            }
        } else if (g <= r && g <= bl) {
            I += white_I[bin] * g;
            if (r <= bl) {
                I += magenta_I[bin] * (r - g);
                I += blue_I[bin] * (bl - r);
            } else {
                I += magenta_I[bin] * (bl - g);
                I += red_I[bin] * (r - bl);
            }
        } else {
            I += white_I[bin] * bl;
            if (r <= g) {
                I += yellow_I[bin] * (r - bl);
                I += green_I[bin] * (g - r);
            } else {
                I += yellow_I[bin] * (g - bl);
                I += red_I[bin] * (r - g);
            }
        }
        v[i] = I;
    }
    return spectrum_from_wavelength_table(v.data(), nullptr, N, 380.f, 1.f);
}
// src/spectrum/blackbody.cpp:30-56 + colourspace/blackbody.hpp:24-49.  Planck's spectral radiance per mm of wavelength, TIMES 1e-10
// ("to make values more inline with emitter db quantities", blackbody.hpp:47-48: without that factor a blackbody area emitter
// outshines a database-spectrum spot by ten orders of magnitude and the emitter selection never picks the spot), sampled at the
// reference's knots — 8 nm steps from 8 nm, 8 nm + lambda / 100 beyond 800 nm — and interpolated linearly in WAVENUMBER between
// them, as the reference's piecewise-linear approximation does.  Baked over the wavelength range the bundled sensors see.
int scene_builder_t::spectrum_blackbody(float T, float scale) {
    const double c = 299792458.0, h = 6.62607015e-34, kB = 1.380649e-23;
    const double c1 = 2 * h * c * c, c2 = h * c / kB;
    auto planck = [&](double lnm) {
        const double l = lnm * 1e-9;
        const double Le = c1 / (std::pow(l, 5) * (std::exp(c2 / (l * T)) - 1.0));   // W/m^2/sr per m of wavelength
        return (double)(float)(Le * 1e-3) * 1e-10 * scale;                          // per mm, (f_t)ret * 1e-10, x scale
    };
    // knots over the reference's whole range, 10 nm .. 5 mm (blackbody.cpp:95-98: "hardcoded max & min"), from max(8 nm, 10 nm - 8 nm)
    std::vector<double> knot_l, knot_v;
    for (double l = 8.0; l <= 5e6 + 8.0;) {
        knot_l.push_back(l);
        knot_v.push_back(planck(l));
        l += l < 800.0 ? 8.0 : 8.0 + l / 100.0;
    }
    std::vector<float> v(SPD_N);
    size_t seg = 0;
    for (int i = 0; i < SPD_N; ++i) {
        const double l = SPD_LAMBDA_MIN_NM + SPD_LAMBDA_STEP_NM * i;
        while (seg + 2 < knot_l.size() && knot_l[seg + 1] < l) ++seg;
        const double k = 1.0 / l, k0 = 1.0 / knot_l[seg], k1 = 1.0 / knot_l[seg + 1];   // linear in k (the 2 pi cancels)
        const double f = (k - k0) / (k1 - k0);
        v[i] = (float)(knot_v[seg] * (1 - f) + knot_v[seg + 1] * f);
    }
    const int id = spectrum_from_wavelength_table(v.data(), nullptr, SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
    auto& support = spectrum_support_knots_[id];
    for (size_t j = knot_l.size(); j-- > 0;) support.push_back({2.0 * M_PI / (knot_l[j] * 1e-6), knot_v[j]});   // k in 1/mm, ascending
    return id;
}
// Share of an emission spectrum's k-integral that lies inside [k_lo, k_hi]:  power(range) / power(all wavenumbers)  of
// scene_build_sensor_sampling_data.cpp:79-80 (the spectrum is piecewise linear in k over its support, 0 outside).
double scene_builder_t::emission_fraction_in_range(int spectrum, double k_lo, double k_hi) const {
    std::vector<std::pair<double, double>> tmp;
    const std::vector<std::pair<double, double>>* kn = nullptr;
    const auto it = spectrum_support_knots_.find(spectrum);
    if (it != spectrum_support_knots_.end())
        kn = &it->second;
    else {
        const spectrum_t& sp = spectra_[spectrum];
        if (sp.type != SPEC_TABLE) return 1.0;
        const int N = 8192;
        for (int i = 0; i < N; ++i) {
            const double k = sp.kmin + (double(sp.kmax) - sp.kmin) * i / (N - 1);
            tmp.push_back({k, (double)spectrum_eval(spectrum, (float)k)});
        }
        kn = &tmp;
    }
    double all = 0, in = 0;
    for (size_t i = 0; i + 1 < kn->size(); ++i) {
        const double k0 = (*kn)[i].first, k1 = (*kn)[i + 1].first, v0 = (*kn)[i].second, v1 = (*kn)[i + 1].second;
        if (!(k1 > k0)) continue;
        all += .5 * (v0 + v1) * (k1 - k0);
        const double a = std::max(k0, k_lo), b = std::min(k1, k_hi);
        if (b > a) {
            const double va = v0 + (v1 - v0) * (a - k0) / (k1 - k0), vb = v0 + (v1 - v0) * (b - k0) / (k1 - k0);
            in += .5 * (va + vb) * (b - a);
        }
    }
    return all > 0 ? in / all : 1.0;
}
int scene_builder_t::spectrum_named(const std::string& name) {
    struct ent {
        const char* n;
        const float* re;
        const float* im;
    };
    static const ent tbl[] = {{"Al", SPD_IOR_Al_n, SPD_IOR_Al_k},   {"Au", SPD_IOR_Au_n, SPD_IOR_Au_k},   {"SF5", SPD_IOR_SF5_n, SPD_IOR_SF5_k},
                              {"SF11", SPD_IOR_SF11_n, SPD_IOR_SF11_k}, {"BK7", SPD_IOR_BK7_n, SPD_IOR_BK7_k}, {"Ag", SPD_IOR_Ag_n, SPD_IOR_Ag_k},
                              {"Cu", SPD_IOR_Cu_n, SPD_IOR_Cu_k},   {"CFL2534", SPD_EMISSION_CFL2534, nullptr}, {"CMF_X", SPD_CMF_X, nullptr},
                              {"CMF_Y", SPD_CMF_Y, nullptr},        {"CMF_Z", SPD_CMF_Z, nullptr}};
    for (auto& e : tbl)
        if (name == e.n) return spectrum_from_wavelength_table(e.re, e.im, SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
    throw std::runtime_error("unknown spectrum " + name);
}
float scene_builder_t::spectrum_eval(int id, float k) const {
    scene_t tmp = sc_;
    tmp.spectra = spectra_.data();
    tmp.spectra_data = spectra_data_.data();
    return spectrum_f(tmp, id, k);
}

static texture_t texture_base(int32_t type) {
    texture_t t{};
    t.type = type;
    t.rgba[3] = 1.f;
    t.col1 = t.col2 = -1;
    t.m[0] = t.m[3] = 1.f;
    t.scale = 1.f;
    t.uwrap = t.vwrap = WRAP_REPEAT;
    return t;
}
int scene_builder_t::add_texture_constant(float r, float g, float b, float a) {
    texture_t t = texture_base(TEX_CONSTANT);
    t.rgba[0] = r;
    t.rgba[1] = g;
    t.rgba[2] = b;
    t.rgba[3] = a;
    textures_.push_back(t);
    return (int)textures_.size() - 1;
}
int scene_builder_t::add_texture_checkerboard(int tex1, int tex2) {
    if (tex1 < 0 || tex2 < 0 || tex1 >= (int)textures_.size() || tex2 >= (int)textures_.size()) throw std::runtime_error("checkerboard: unknown nested texture");
    // (a checkerboard's cells are looked up through texture_rgba, wt/scene.h, which knows constants, checkerboards and bitmaps: an expression
    // there would render as its default colour without a word)
    if (texture_is_function(tex1) || texture_is_function(tex2))
        throw std::runtime_error("checkerboard: function / mix textures as cell colours are not supported (only constant, checkerboard and bitmap textures)");
    texture_t t = texture_base(TEX_CHECKERBOARD);
    t.col1 = tex1;
    t.col2 = tex2;
    textures_.push_back(t);
    return (int)textures_.size() - 1;
}
int scene_builder_t::add_texture_bitmap(uint32_t width, uint32_t height, uint32_t channels, const float* texels, uint32_t filter, uint32_t uwrap, uint32_t vwrap) {
    if (!width || !height || channels < 1 || channels > 4 || !texels) throw std::runtime_error("bitmap texture: width, height > 0 and 1..4 channels expected");
    texture_t t = texture_base(TEX_BITMAP);
    t.width = width;
    t.height = height;
    t.channels = channels;
    t.offset = (uint32_t)texture_data_.size();
    if (filter > 2u) throw std::runtime_error("bitmap texture: filter 0 (nearest), 1 (bilinear) or 2 (bicubic) expected");
    t.bilinear = filter;
    t.uwrap = uwrap;
    t.vwrap = vwrap;
    texture_data_.insert(texture_data_.end(), texels, texels + (size_t)width * height * channels);
    textures_.push_back(t);
    return (int)textures_.size() - 1;
}
int scene_builder_t::add_texture_function(const std::vector<float>& program) {
    if (program.empty() || program.size() % 2) throw std::runtime_error("function texture: empty or malformed program");
    std::vector<float> flat;
    int depth = 0, max_depth = 0;
    for (size_t i = 0; i < program.size(); i += 2) {
        const int op = (int)program[i];
        if (op == TOP_TEX) {
            const int id = (int)program[i + 1];
            if (id < 0 || id >= (int)textures_.size()) throw std::runtime_error("function texture: unknown nested texture");
            const texture_t& n = textures_[id];
            if (n.type == TEX_FUNCTION) {   // inline the nested program (the device evaluates no function inside a function)
                if (n.m[0] != 1.f || n.m[1] != 0.f || n.m[2] != 0.f || n.m[3] != 1.f || n.t[0] != 0.f || n.t[1] != 0.f)
                    throw std::runtime_error("function texture: a transformed function texture cannot be nested (transform its operands)");
                flat.insert(flat.end(), texture_data_.begin() + n.offset, texture_data_.begin() + n.offset + n.width);
                if (n.scale != 1.f) {
                    flat.push_back((float)TOP_CONST);
                    flat.push_back(n.scale);
                    flat.push_back((float)TOP_MUL);
                    flat.push_back(0.f);
                }
                continue;
            }
        }
        flat.push_back(program[i]);
        flat.push_back(program[i + 1]);
    }
    for (size_t i = 0; i < flat.size(); i += 2) {   // stack discipline (the device's evaluation stack holds 12 values)
        const int op = (int)flat[i];
        const bool unary = op == TOP_NEG || (op >= TOP_ABS && op <= TOP_ATAN) || op == TOP_NOT;
        depth += op <= TOP_TEX ? 1 : (unary ? 0 : (op == TOP_MIX ? -2 : -1));
        if (op > TOP_NOT || op < 0 || depth < 1) throw std::runtime_error("function texture: malformed program");
        max_depth = std::max(max_depth, depth);
    }
    if (depth != 1) throw std::runtime_error("function texture: the program does not leave exactly one value");
    if (max_depth > 12) throw std::runtime_error("function texture: expression too deep (more than 12 intermediate values)");
    texture_t t = texture_base(TEX_FUNCTION);
    t.offset = (uint32_t)texture_data_.size();
    t.width = (uint32_t)flat.size();
    texture_data_.insert(texture_data_.end(), flat.begin(), flat.end());
    textures_.push_back(t);
    return (int)textures_.size() - 1;
}
void scene_builder_t::texture_set_transform(int tex, const float M[4], const float tr[2]) {
    texture_t& t = textures_.at(tex);
    for (int i = 0; i < 4; ++i) t.m[i] = M[i];
    t.t[0] = tr[0];
    t.t[1] = tr[1];
}
void scene_builder_t::texture_set_scale(int tex, float scale) { textures_.at(tex).scale = scale; }
void scene_builder_t::texture_compose_transform(int tex, const float M[4], const float tr[2]) {
    texture_t& t = textures_.at(tex);
    const float a[4] = {t.m[0], t.m[1], t.m[2], t.m[3]}, ta[2] = {t.t[0], t.t[1]};
    t.m[0] = a[0] * M[0] + a[1] * M[2];
    t.m[1] = a[0] * M[1] + a[1] * M[3];
    t.m[2] = a[2] * M[0] + a[3] * M[2];
    t.m[3] = a[2] * M[1] + a[3] * M[3];
    t.t[0] = a[0] * tr[0] + a[1] * tr[1] + ta[0];
    t.t[1] = a[2] * tr[0] + a[3] * tr[1] + ta[1];
}

int scene_builder_t::add_material(const material_t& m) {
    materials_.push_back(m);
    return (int)materials_.size() - 1;
}

// src/mesh/mesh.cpp:27-87 (tris_from_indices) + surface_differentials.hpp
int scene_builder_t::add_shape(const mesh_t& mesh, const xform_t& to_world, int material, bool face_normals) {
    shape_rec_t rec{};
    rec.material = material;
    rec.emitter = -1;
    rec.tri_begin = (uint32_t)wtris_.size();
    const uint32_t shape_idx = (uint32_t)shape_recs_.size();
    for (const auto& t : mesh.tris) {
        dvec3 a = to_world.point(mesh.verts[t[0]]), b = to_world.point(mesh.verts[t[1]]), c = to_world.point(mesh.verts[t[2]]);
        const dvec3 cr = dcross(b - a, c - a);
        if (dlen(cr) == 0) continue;
        dvec3 gn = dnorm(cr);
        wtri_t w{};
        w.has_uv = !mesh.uvs.empty();
        std::array<float, 2> uv[3] = {{0, 0}, {0, 0}, {0, 0}};
        if (w.has_uv)
            for (int i = 0; i < 3; ++i) uv[i] = mesh.uvs[t[i]];
        dvec3 n1 = gn, n2 = gn, n3 = gn;
        if (!mesh.normals.empty() && !face_normals) {
            n1 = to_world.normal(mesh.normals[t[0]]);
            n2 = to_world.normal(mesh.normals[t[1]]);
            n3 = to_world.normal(mesh.normals[t[2]]);
            if (ddot(n1, gn) < 0 && ddot(n2, gn) < 0 && ddot(n3, gn) < 0) {
                std::swap(a, b);
                std::swap(uv[0], uv[1]);
                std::swap(n1, n2);
                gn = gn * -1.0;
            }
        }
        w.a = tof(a);
        w.b = tof(b);
        w.c = tof(c);
        w.n = tof(gn);
        w.n0 = tof(n1);
        w.n1 = tof(n2);
        w.n2 = tof(n3);
        w.uv0 = {uv[0][0], uv[0][1]};
        w.uv1 = {uv[1][0], uv[1][1]};
        w.uv2 = {uv[2][0], uv[2][1]};
        w.shape = shape_idx;
        w.shape_tri = (uint32_t)wtris_.size() - rec.tri_begin;
        wtris_.push_back(w);
    }
    rec.tri_count = (uint32_t)wtris_.size() - rec.tri_begin;
    if (rec.tri_count == 0) throw std::runtime_error("shape without geometry");
    shape_recs_.push_back(rec);
    return (int)shape_idx;
}

int scene_builder_t::add_emitter_spot(const xform_t& to_world, int spectrum, float scale, float cutoff, float falloff, float extent_m, float pse_scale) {
    emitter_t e{};
    e.type = EMIT_SPOT;
    e.spectrum = spectrum;
    e.scale = scale;
    e.phase_space_extent_scale = pse_scale;
    e.position = tof(to_world.point({0, 0, 0}));
    e.frame.t = tof(dnorm(to_world.vector({1, 0, 0})));
    e.frame.b = tof(dnorm(to_world.vector({0, 1, 0})));
    e.frame.n = tof(dnorm(to_world.vector({0, 0, 1})));
    e.cutoff = cutoff;
    e.falloff = falloff;
    e.cos_cutoff = std::cos(cutoff);
    e.cos_falloff = std::cos(falloff);
    e.recp_cutoff_range = 1.f / (cutoff - falloff);
    e.max_tan_alpha = std::tan(falloff);
    e.extent = extent_m;
    e.shape = -1;
    emitters_.push_back(e);
    return (int)emitters_.size() - 1;
}
int scene_builder_t::add_emitter_directional(dvec3 dir_to_emitter, int spectrum, float scale, float solid_angle_sr, float pse_scale) {
    emitter_t e{};
    e.type = EMIT_DIRECTIONAL;
    e.spectrum = spectrum;
    e.scale = scale;
    e.phase_space_extent_scale = pse_scale;
    e.frame = build_orthogonal_frame(tof(dnorm(dir_to_emitter)));
    // directional.hpp:86: tan(acos(1 - solid_angle / 2 pi))
    e.tan_alpha_at_target = std::tan(std::acos(1.f - solid_angle_sr * kInvTwoPi));
    e.shape = -1;
    emitters_.push_back(e);   // position / target / far are set in finalize() from the world AABB (scene.cpp:50-58)
    return (int)emitters_.size() - 1;
}
int scene_builder_t::add_emitter_point(dvec3 position, int spectrum, float scale, float extent_m, float pse_scale) {
    emitter_t e{};
    e.type = EMIT_POINT;
    e.spectrum = spectrum;
    e.scale = scale;
    e.phase_space_extent_scale = pse_scale;
    e.position = tof(position);
    e.frame = frame_t{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    e.extent = extent_m;
    e.shape = -1;
    emitters_.push_back(e);
    return (int)emitters_.size() - 1;
}
// ITU-R P.2040-2 table 3 (src/spectrum/util/spectrum_from_ITU.cpp:26-52,78-200): eps_r = a f^b, sigma = c f^d [S/m] (f in GHz),
// IOR = sqrt(eps_r - i sigma / (eps0 omega)).  Outside a material's frequency range the reference returns 0.
int scene_builder_t::spectrum_itu(const std::string& material, float wavelength_mm) {
    struct itu_t {
        const char* name;
        double a, b, c, d, fmin_ghz, fmax_ghz;
    };
    static const itu_t table[] = {{"vacuum", 1, 0, 0, 0, 0, 1e30},
                                  {"concrete", 5.24, 0, 0.0462, 0.7822, 1, 100},
                                  {"brick", 3.91, 0, 0.0238, 0.16, 1, 40},
                                  {"plasterboard", 2.73, 0, 0.0085, 0.9395, 1, 100},
                                  {"wood", 1.99, 0, 0.0047, 1.0718, 0.001, 100},
                                  {"chipboard", 2.58, 0, 0.0217, 0.7800, 1, 100},
                                  {"plywood", 2.71, 0, 0.33, 0, 1, 40},
                                  {"marble", 7.074, 0, 0.0055, 0.9262, 1, 60},
                                  {"metal", 1, 0, 1e7, 0, 1, 100}};
    const double c0 = 299792458.0, mu0 = 1.25663706212e-6;
    const double f_hz = c0 / (wavelength_mm * 1e-3), f_ghz = f_hz * 1e-9;
    for (const auto& t : table) {
        if (material != t.name) continue;
        if (f_ghz < t.fmin_ghz || f_ghz > t.fmax_ghz) return spectrum_const(0.f, 0.f);
        const double eps_r = t.a * (t.b != 0 ? std::pow(f_ghz, t.b) : 1.0);
        const double sigma = t.c * (t.d != 0 ? std::pow(f_ghz, t.d) : 1.0);
        const double eps0 = 1.0 / (mu0 * c0 * c0);
        const double omega = 2.0 * M_PI * f_hz;
        const std::complex<double> ior = std::sqrt(std::complex<double>(eps_r, -sigma / (eps0 * omega)));
        return spectrum_const((float)ior.real(), (float)ior.imag());
    }
    throw std::runtime_error("unknown ITU material " + material);
}
// new_order[i] = the current index of the emitter that becomes emitter i (the reference lists a scene's free emitters first, ordered
// by element id, then the area emitters in shape order: src/scene/loader/loader.cpp:272-310)
void scene_builder_t::permute_emitters(const std::vector<int>& new_order) {
    if (new_order.size() != emitters_.size()) throw std::runtime_error("permute_emitters: wrong length");
    std::vector<emitter_t> out(emitters_.size());
    std::vector<int> where(emitters_.size(), -1);
    for (size_t i = 0; i < new_order.size(); ++i) {
        const int from = new_order[i];
        if (from < 0 || (size_t)from >= emitters_.size() || where[from] >= 0) throw std::runtime_error("permute_emitters: not a permutation");
        out[i] = emitters_[from];
        where[from] = (int)i;
    }
    emitters_.swap(out);
    for (auto& r : shape_recs_)
        if (r.emitter >= 0) r.emitter = where[r.emitter];
}
int scene_builder_t::add_emitter_area(int shape, int spectrum, float scale, float pse_scale) {
    emitter_t e{};
    e.type = EMIT_AREA;
    e.spectrum = spectrum;
    e.scale = scale;
    e.phase_space_extent_scale = pse_scale;
    e.shape = shape;
    e.frame = frame_t{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    emitters_.push_back(e);
    shape_recs_[shape].emitter = (int)emitters_.size() - 1;
    return (int)emitters_.size() - 1;
}

// An area emitter whose radiance is a bitmap texture (src/emitter/area.cpp:295-340, 153-216).  The emitter's spectrum is the texture's mean
// spectrum (src/texture/bitmap.cpp:45-57: the mean texel, uplifted if the image is RGB, flat otherwise), which the emitter selection and the
// spectral sampling use; positions are drawn per triangle from the luminance of the texture on a barycentric grid of
// ceil(min(resolution, 512) x longest uv edge) cells a side.  Every accumulation is in single precision, in the reference's order.
int scene_builder_t::add_emitter_area_textured(int shape, int tex, float scale, float pse_scale) {
    const texture_t t = textures_.at(tex);
    if (t.type != TEX_BITMAP) throw std::runtime_error("(area emitter loader) the radiance texture must provide mean_spectrum(): a bitmap (constant textures: add_emitter_area)");
    const shape_rec_t& r = shape_recs_.at(shape);
    // texture2d_t::compute_texture_data (bitmap/texture2d.hpp:46-61): the mean texel
    float mean[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t y = 0; y < t.height; ++y)
        for (uint32_t x = 0; x < t.width; ++x)
            for (uint32_t c = 0; c < t.channels && c < 4; ++c) mean[c] += texture_data_[t.offset + ((size_t)y * t.width + x) * t.channels + c];
    for (float& m : mean) m /= float(t.width * t.height);
    const bool rgb = t.channels >= 3;
    const int spec = rgb ? spectrum_rgb(mean[0] * t.scale, mean[1] * t.scale, mean[2] * t.scale) : spectrum_const(mean[0] * t.scale);
    const int ei = add_emitter_area(shape, spec, scale, pse_scale);

    // working resolution: transform_t::resolution over bitmap_t::resolution (texture/transform.hpp:89-94, bitmap.hpp:70-72), at most 512
    const float r0x = std::max(1.f, float(t.width)), r0y = std::max(1.f, float(t.height));
    const float rrx = t.m[0] * (1.f / r0x) + t.m[1] * (1.f / r0y), rry = t.m[2] * (1.f / r0x) + t.m[3] * (1.f / r0y);
    const float res = std::min(std::max(std::max(1.f, 1.f / rrx), std::max(1.f, 1.f / rry)), 512.f);

    scene_t view{};   // texture lookups through the device's own code (wt/scene.h)
    view.textures = textures_.data();
    view.n_textures = (uint32_t)textures_.size();
    view.texture_data = texture_data_.data();
    const uint32_t T = r.tri_count;
    std::vector<float> tab(T + 1 + 4 * (size_t)T, 0.f), powers(T, 0.f);
    float total_I = 0.f;
    for (uint32_t i = 0; i < T; ++i) {
        const wtri_t& w = wtris_[r.tri_begin + i];
        float* h = &tab[T + 1 + 4 * (size_t)i];
        const size_t off = tab.size();
        if (!w.has_uv) {   // a single empty cell
            h[0] = 0.f, h[1] = 0.f, h[2] = 0.f, h[3] = float(off);
            tab.push_back(0.f);
            tab.push_back(1.f);
            continue;
        }
        const float tarea = .5f * length(cross(w.c - w.a, w.b - w.a));
        const float max_uv_dist = std::max(length(w.uv1 - w.uv0), std::max(length(w.uv2 - w.uv0), length(w.uv2 - w.uv1)));
        const uint32_t texels = (uint32_t)std::ceil(res * max_uv_dist);
        const float step = 1.f / float(texels);
        const size_t cells = (size_t)texels * (texels + 1) / 2;
        if (texels == 0 || off + cells + 1 >= (size_t(1) << 24))
            throw std::runtime_error("(area emitter) the per-triangle sampling tables of this radiance texture need 2^24 words or more (or a triangle has no uv extent)");
        tab.resize(off + cells + 1);
        float* cdf = &tab[off];
        h = &tab[T + 1 + 4 * (size_t)i];
        cdf[0] = 0.f;
        float I = 0.f;
        size_t n = 0;
        for (uint32_t b = 0; b < texels; ++b)
            for (uint32_t a = 0; a <= b; ++a) {
                const float alpha = float(a) * step + .5f * step, beta = 1.f - (float(b) * step + .5f * step);
                const vec2 uv = alpha * w.uv0 + beta * w.uv1 + std::max(0.f, 1.f - alpha - beta) * w.uv2;
                const rgba_t c = texture_rgba(view, tex, uv);
                const float lum = std::max(0.f, (rgb ? .2126f * c.r + .7152f * c.g + .0722f * c.b : c.r));   // colourspace::luminance (BT.709)
                I += lum;
                cdf[n + 1] = cdf[n] + std::max(0.f, lum);   // discrete_distribution_t: accumulate, then normalise (discrete_distribution.hpp:60-84)
                ++n;
            }
        const float sum = cdf[cells];
        if (sum > 0.f) {
            const float recp = 1.f / sum;
            for (size_t k = 0; k <= cells; ++k) cdf[k] *= recp;
        } else
            cdf[cells] = 1.f;
        powers[i] = I / float(cells) * tarea;
        total_I += powers[i];
        h[0] = float(texels), h[1] = step, h[2] = 1.f / (step * step * tarea), h[3] = float(off);
    }
    if (!(total_I > 0.f)) std::fprintf(stderr, "wtgpu: (area emitter) radiance texture is zero everywhere\n");
    tab[0] = 0.f;
    for (uint32_t i = 0; i < T; ++i) tab[i + 1] = tab[i] + std::max(0.f, powers[i]);
    if (tab[T] > 0.f) {
        const float recp = 1.f / tab[T];
        for (uint32_t i = 0; i <= T; ++i) tab[i] *= recp;
    } else
        tab[T] = 1.f;
    emitter_t& e = emitters_[ei];
    e.radiance_tex = 1 + tex;
    e.tab = (uint32_t)texture_data_.size();
    e.tab_words = (uint32_t)tab.size();
    texture_data_.insert(texture_data_.end(), tab.begin(), tab.end());
    return ei;
}

// include/wt/sensor/sensor/perspective.hpp:103-155
void scene_builder_t::set_sensor_perspective(const xform_t& to_world, double fov, uint32_t w, uint32_t h, float pse_scale, bool rt_only) {
    sensor_t& s = sc_.sensor;
    s.type = SENSOR_PERSPECTIVE;
    s.width = w;
    s.height = h;
    s.ray_trace_only = rt_only;
    s.position = tof(to_world.point({0, 0, 0}));
    s.frame.t = tof(dnorm(to_world.vector({1, 0, 0})));
    s.frame.b = tof(dnorm(to_world.vector({0, 1, 0})));
    s.frame.n = tof(dnorm(to_world.vector({0, 0, 1})));
    const double znear = 0.01;
    const double hh = 1.0 / std::tan(fov / 2), ww = hh / (double(w) / double(h));
    xform_t P{};
    P.m[0] = ww;
    P.m[5] = hh;
    P.m[11] = znear;   // row 2, col 3
    P.m[14] = 1.0;     // row 3, col 2
    const xform_t V = xform_t::scale(.5 * (w - 1.0), .5 * (h - 1.0), 1) * xform_t::translate(1, 1, 0) * xform_t::scale(-1, -1, 1);
    const xform_t cam = V * P;
    double inv[16];
    if (!invert4(cam.m, inv)) throw std::runtime_error("singular camera matrix");
    for (int i = 0; i < 16; ++i) {
        s.cam[i] = (float)cam.m[i];
        s.inv_cam[i] = (float)inv[i];
    }
    auto pos = [&](double fx, double fy) {
        const double x = inv[0] * fx + inv[1] * fy + inv[2] + inv[3], y = inv[4] * fx + inv[5] * fy + inv[6] + inv[7];
        const double z = inv[8] * fx + inv[9] * fy + inv[10] + inv[11], ww2 = inv[12] * fx + inv[13] * fy + inv[14] + inv[15];
        return dvec3{x / ww2, y / ww2, z / ww2};
    };
    const dvec3 p00 = pos(0, 0);
    s.ddir_dx = tof(pos(1, 0) - p00);
    s.ddir_dy = tof(pos(0, 1) - p00);
    const double ex = dlen(pos(w, 0) - p00), ey = dlen(pos(0, h) - p00);
    s.sensor_area = (float)(ex * ey);
    s.element_extent_x = (float)(ex / w);
    s.sourcing_tan_alpha = (float)((ex / w) / znear);
    s.phase_space_extent_scale = pse_scale;
    s.requested_tan_alpha = -1.f;
}
// src/sensor/virtual_plane_sensor.cpp:31-63
void scene_builder_t::set_sensor_virtual_plane(const xform_t& to_world, double ex, double ey, uint32_t w, uint32_t h, float tan_alpha) {
    sensor_t& s = sc_.sensor;
    s.type = SENSOR_VIRTUAL_PLANE;
    s.width = w;
    s.height = h;
    s.ray_trace_only = 0;
    const dvec3 t = dnorm(to_world.vector({1, 0, 0})), b = dnorm(to_world.vector({0, 1, 0})), n = dnorm(to_world.vector({0, 0, 1}));
    s.frame = frame_t{tof(t), tof(b), tof(n)};
    ex *= dlen(to_world.vector({1, 0, 0}));
    ey *= dlen(to_world.vector({0, 1, 0}));
    const dvec3 centre = to_world.point({0, 0, 0});
    s.origin = tof(centre - t * (ex / 2) - b * (ey / 2));
    s.position = tof(centre);
    s.extent = {(float)ex, (float)ey};
    s.element_extent = {(float)(ex / w), (float)(ey / h)};
    s.recp_area = (float)(1.0 / (ex * ey));
    s.requested_tan_alpha = tan_alpha;
    s.phase_space_extent_scale = 1.f;
}
void scene_builder_t::set_film_rfilter_scale(float s) { rfilter_scale_ = s; }
void scene_builder_t::set_response_rgb(const float white_xyz[3]) {
    response_is_rgb_ = true;
    // include/wt/spectrum/colourspace/RGB/RGB.hpp (CIE RGB, reference white E) + whitepoint.hpp (Bradford CAT)
    const double M[9] = {2.3706743, -0.9000405, -0.4706338, -0.5138850, 1.4253036, 0.0885814, 0.0052982, -0.0146949, 1.0093968};
    const double MA[9] = {0.8951, 0.2664, -0.1614, -0.7502, 1.7135, 0.0367, 0.0389, -0.0685, 1.0296};
    const double iMA[9] = {0.9869929, -0.1470543, 0.1599627, 0.4323053, 0.5183603, 0.0492912, -0.0085287, 0.0400428, 0.9684867};
    auto mul3 = [](const double* A, const double* B, double* C) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
                C[i * 3 + j] = s;
            }
    };
    const double sw[3] = {1, 1, 1}, dw[3] = {white_xyz[0], white_xyz[1], white_xyz[2]};
    double rs[3], rd[3];
    for (int i = 0; i < 3; ++i) {
        rs[i] = MA[i * 3] * sw[0] + MA[i * 3 + 1] * sw[1] + MA[i * 3 + 2] * sw[2];
        rd[i] = MA[i * 3] * dw[0] + MA[i * 3 + 1] * dw[1] + MA[i * 3 + 2] * dw[2];
    }
    double D[9] = {rd[0] / rs[0], 0, 0, 0, rd[1] / rs[1], 0, 0, 0, rd[2] / rs[2]}, T1[9], CAT[9], C[9];
    mul3(D, MA, T1);
    mul3(iMA, T1, CAT);
    const bool same = dw[0] == 1 && dw[1] == 1 && dw[2] == 1;
    if (same)
        std::memcpy(C, M, sizeof(C));
    else
        mul3(M, CAT, C);
    // f(channel,k) = max(0, (C * xyz(k))[channel])  (src/sensor/response/RGB.cpp:24-34)
    std::vector<float> ch[3], sens(SPD_N);
    for (int c = 0; c < 3; ++c) ch[c].resize(SPD_N);
    for (int i = 0; i < SPD_N; ++i) {
        const double x = SPD_CMF_X[i], y = SPD_CMF_Y[i], z = SPD_CMF_Z[i];
        for (int c = 0; c < 3; ++c) ch[c][i] = (float)std::max(0.0, C[c * 3] * x + C[c * 3 + 1] * y + C[c * 3 + 2] * z);
        sens[i] = (float)(x + y + z);   // multichannel sensitivity = sum of channels (multichannel.hpp:65-67)
    }
    sc_.sensor.channels = 3;
    for (int c = 0; c < 3; ++c) sc_.sensor.response_spec[c] = spectrum_from_wavelength_table(ch[c].data(), nullptr, SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
    sensitivity_spec_ = spectrum_from_wavelength_table(sens.data(), nullptr, SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
}
void scene_builder_t::set_response_mono_discrete(float wavelength_mm) {
    response_is_rgb_ = false;
    mono_lambda_mm_ = wavelength_mm;
    sc_.sensor.channels = 1;
    sc_.sensor.response_spec[0] = spectrum_discrete(wavelength_mm, 1.f);
    sensitivity_spec_ = sc_.sensor.response_spec[0];
}
void scene_builder_t::set_integrator(const integrator_opts_t& o) { sc_.opts = o; }
void scene_builder_t::set_fsd_lut_resolution(uint32_t n_theta, uint32_t m) {
    lut_n_theta_ = n_theta;
    lut_m_ = m;
}

// ---------------------------------------------------------------------------------------------------------
// BVH: binned SAH binary build + collapse of 3 binary levels into an 8-wide node
namespace {
struct aabb_t {
    float mn[3], mx[3];
    void reset() {
        for (int i = 0; i < 3; ++i) {
            mn[i] = WT_INF;
            mx[i] = -WT_INF;
        }
    }
    void grow(const vec3& p) {
        const float v[3] = {p.x, p.y, p.z};
        for (int i = 0; i < 3; ++i) {
            mn[i] = std::min(mn[i], v[i]);
            mx[i] = std::max(mx[i], v[i]);
        }
    }
    void grow(const aabb_t& o) {
        for (int i = 0; i < 3; ++i) {
            mn[i] = std::min(mn[i], o.mn[i]);
            mx[i] = std::max(mx[i], o.mx[i]);
        }
    }
    float area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
    }
};
struct bnode_t {
    aabb_t box;
    int left = -1, right = -1;
    uint32_t first = 0, count = 0;
    bool leaf = false;
};
}   // namespace

void scene_builder_t::build_bvh() {
    const uint32_t N = (uint32_t)wtris_.size();
    std::vector<uint32_t> order(N);
    std::iota(order.begin(), order.end(), 0u);
    std::vector<aabb_t> tb(N);
    std::vector<std::array<float, 3>> cen(N);
    for (uint32_t i = 0; i < N; ++i) {
        tb[i].reset();
        tb[i].grow(wtris_[i].a);
        tb[i].grow(wtris_[i].b);
        tb[i].grow(wtris_[i].c);
        for (int a = 0; a < 3; ++a) cen[i][a] = .5f * (tb[i].mn[a] + tb[i].mx[a]);
    }
    std::vector<bnode_t> bn;
    bn.reserve(2 * N);
    const float C_INT = 1.f, C_TRAV = 1.f;   // relative costs of the binary builder
    const uint32_t MAX_LEAF = 4;
    constexpr int MAX_BINS = 256;
    const int BINS = getenv("WTGPU_BVH_BINS") ? std::min(MAX_BINS, std::max(4, atoi(getenv("WTGPU_BVH_BINS")))) : 32;
    // knobs (read when a scene is baked): ranges of at most this many triangles become leaves without a SAH test, and how binary
    // subtrees are gathered into 8-wide nodes (see below)
    // (measured on the 283 K-triangle workload, ms per pass: 3 binary levels per node, leaves <= 2: 202.0; SAH-optimal grouping: 199; with
    // ranges of <= 4 triangles kept as leaves: 190.0 — `k_trace_heavy` 205 -> 189 ms, `k_trace` 185 -> 173 ms stream-summed)
    const uint32_t force_leaf = getenv("WTGPU_BVH_FORCE_LEAF") ? (uint32_t)std::max(1, atoi(getenv("WTGPU_BVH_FORCE_LEAF"))) : 4u;
    std::function<int(uint32_t, uint32_t, uint32_t)> build = [&](uint32_t first, uint32_t count, uint32_t depth) -> int {
        const int id = (int)bn.size();
        bn.emplace_back();
        bn[id].first = first;
        bn[id].count = count;
        bn[id].box.reset();
        aabb_t cb;
        cb.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            bn[id].box.grow(tb[order[i]]);
            cb.grow(vec3{cen[order[i]][0], cen[order[i]][1], cen[order[i]][2]});
        }
        bvh_max_depth_ = std::max(bvh_max_depth_, depth);
        if (count <= std::min(force_leaf, 7u)) {
            bn[id].leaf = true;
            return id;
        }
        float best = WT_INF;
        int best_axis = -1, best_bin = -1;
        const float pa = bn[id].box.area();
        for (int ax = 0; ax < 3; ++ax) {
            const float lo = cb.mn[ax], hi = cb.mx[ax];
            if (!(hi > lo)) continue;
            aabb_t bb[MAX_BINS];
            uint32_t bc[MAX_BINS] = {0};
            for (int b = 0; b < BINS; ++b) bb[b].reset();
            const float sc = BINS / (hi - lo);
            for (uint32_t i = first; i < first + count; ++i) {
                int b = (int)((cen[order[i]][ax] - lo) * sc);
                b = std::min(BINS - 1, std::max(0, b));
                bb[b].grow(tb[order[i]]);
                bc[b]++;
            }
            float la[MAX_BINS], ra[MAX_BINS];
            uint32_t lc[MAX_BINS], rc[MAX_BINS];
            aabb_t acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = 0; b < BINS; ++b) {
                if (bc[b]) acc.grow(bb[b]);
                c += bc[b];
                la[b] = acc.area();
                lc[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = BINS - 1; b >= 0; --b) {
                if (bc[b]) acc.grow(bb[b]);
                c += bc[b];
                ra[b] = acc.area();
                rc[b] = c;
            }
            for (int b = 0; b < BINS - 1; ++b) {
                if (lc[b] == 0 || rc[b + 1] == 0) continue;
                const float cost = C_TRAV + C_INT * (la[b] * lc[b] + ra[b + 1] * rc[b + 1]) / std::max(pa, 1e-30f);
                if (cost < best) {
                    best = cost;
                    best_axis = ax;
                    best_bin = b;
                }
            }
        }
        uint32_t mid;
        if (best_axis < 0 || (best >= C_INT * count && count <= MAX_LEAF)) {
            if (count <= MAX_LEAF) {
                bn[id].leaf = true;
                return id;
            }
            // degenerate centroids: median split by index
            mid = first + count / 2;
        } else {
            const float lo = cb.mn[best_axis], hi = cb.mx[best_axis], sc = BINS / (hi - lo);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = (int)((cen[t][best_axis] - lo) * sc);
                b = std::min(BINS - 1, std::max(0, b));
                return b <= best_bin;
            });
            mid = (uint32_t)(it - order.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        const int l = build(first, mid - first, depth + 1);
        const int r = build(mid, first + count - mid, depth + 1);
        bn[id].left = l;
        bn[id].right = r;
        return id;
    };
    int root = N ? build(0, N, 0) : -1;

    // ---- optional: insertion-based optimisation of the binary tree (Bittner, Hapala, Havran 2013, "Fast insertion-based optimization of
    // bounding volume hierarchies"): inner nodes are taken out — largest boxes first — and their two subtrees re-inserted where they
    // enlarge the tree's surface area least (branch-and-bound search).  Only the TOPOLOGY changes; the leaves (<= 4 triangles) stay, the
    // triangle order is re-linearised afterwards so that every subtree still owns one contiguous range.  WTGPU_BVH_REINSERT=<passes>
    // (default 2; 283 K-triangle cornell stand-in: summed surface area of the inner nodes 100 % -> 70.9 %, +1 s of build time; a pass of
    // the headline workload 100.6 -> 96.9 ms; 0 / 1 / 2 / 4 passes: 100.6 / 97.7 / 96.9 / 96.8); like every tree knob it changes how
    // fast a query is answered, not the answer.
    const int reinsert_passes = getenv("WTGPU_BVH_REINSERT") ? std::max(0, atoi(getenv("WTGPU_BVH_REINSERT"))) : 2;
    if (reinsert_passes > 0 && root >= 0 && !bn[root].leaf) {
        const int M = (int)bn.size();
        std::vector<int> parent(M, -1);
        for (int i = 0; i < M; ++i)
            if (!bn[i].leaf) parent[bn[i].left] = parent[bn[i].right] = i;
        auto merged = [](const aabb_t& a, const aabb_t& b) {
            aabb_t r = a;
            r.grow(b);
            return r;
        };
        auto refit_up = [&](int n) {
            for (; n >= 0; n = parent[n]) bn[n].box = merged(bn[bn[n].left].box, bn[bn[n].right].box);
        };
        double sa_before = 0;
        for (int i = 0; i < M; ++i)
            if (!bn[i].leaf) sa_before += bn[i].box.area();
        struct cand_t {
            float cost;   // induced cost so far (lower bound of the total)
            int node;
            bool operator<(const cand_t& o) const { return cost > o.cost; }
        };
        // where to insert subtree `x` (detached): the node whose replacement by a new parent {node, x} costs least
        auto best_position = [&](int x) {
            const aabb_t xb = bn[x].box;
            const float xa = xb.area();
            std::priority_queue<cand_t> pq;
            pq.push(cand_t{0.f, root});
            float best = WT_INF;
            int best_node = root;
            while (!pq.empty()) {
                const cand_t c = pq.top();
                pq.pop();
                if (c.cost + xa >= best) break;   // every remaining candidate is at least this expensive
                const float direct = merged(bn[c.node].box, xb).area();
                const float total = c.cost + direct;
                if (total < best) {
                    best = total;
                    best_node = c.node;
                }
                if (!bn[c.node].leaf) {
                    const float induced = c.cost + direct - bn[c.node].box.area();   // enlargement of this node if x goes below it
                    if (induced + xa < best) {
                        pq.push(cand_t{induced, bn[c.node].left});
                        pq.push(cand_t{induced, bn[c.node].right});
                    }
                }
            }
            return best_node;
        };
        // attaches detached subtree `x` as the sibling of `at`, using the free inner node `fresh`
        auto insert_at = [&](int x, int at, int fresh) {
            const int g = parent[at];
            bn[fresh].leaf = false;
            bn[fresh].left = at;
            bn[fresh].right = x;
            parent[fresh] = g;
            if (g >= 0) {
                if (bn[g].left == at)
                    bn[g].left = fresh;
                else
                    bn[g].right = fresh;
            }
            parent[at] = parent[x] = fresh;
            if (g < 0) root = fresh;   // inserted beside the root: the searches and the `p == root` guards below must see the new one
            refit_up(fresh);
        };
        for (int pass = 0; pass < reinsert_passes; ++pass) {
            std::vector<int> inner;
            for (int i = 0; i < M; ++i)
                if (!bn[i].leaf && i != root && parent[i] != root && parent[i] >= 0) inner.push_back(i);
            std::sort(inner.begin(), inner.end(), [&](int a, int b) { return bn[a].box.area() > bn[b].box.area(); });
            inner.resize(inner.size() / (pass == 0 ? 1 : 2));   // later passes: the larger half
            for (int n : inner) {
                const int p = parent[n];
                if (bn[n].leaf || p < 0 || p == root || parent[p] < 0) continue;   // (the tree changed under the list)
                const int g = parent[p];
                const int sib = bn[p].left == n ? bn[p].right : bn[p].left;
                const int l = bn[n].left, r = bn[n].right;
                // take n and p out: g adopts n's sibling; l and r are detached; n and p become the two free inner nodes
                if (bn[g].left == p)
                    bn[g].left = sib;
                else
                    bn[g].right = sib;
                parent[sib] = g;
                refit_up(g);
                parent[l] = parent[r] = -1;
                parent[n] = parent[p] = -1;
                const int first_x = bn[l].box.area() >= bn[r].box.area() ? l : r, second_x = first_x == l ? r : l;
                insert_at(first_x, best_position(first_x), n);
                insert_at(second_x, best_position(second_x), p);
            }
        }
        // (the root may not have moved: insertions beside the root are never chosen by the list above, but keep it robust)
        int new_root = root;
        while (parent[new_root] >= 0) new_root = parent[new_root];
        // re-linearise: triangle ranges in depth-first order
        std::vector<uint32_t> new_order;
        new_order.reserve(N);
        std::function<void(int, uint32_t)> relin = [&](int n, uint32_t depth) {
            bvh_max_depth_ = std::max(bvh_max_depth_, depth);
            if (bn[n].leaf) {
                const uint32_t f = (uint32_t)new_order.size();
                for (uint32_t i = 0; i < bn[n].count; ++i) new_order.push_back(order[bn[n].first + i]);
                bn[n].first = f;
                return;
            }
            relin(bn[n].left, depth + 1);
            relin(bn[n].right, depth + 1);
            bn[n].first = bn[bn[n].left].first;
            bn[n].count = bn[bn[n].left].count + bn[bn[n].right].count;
        };
        bvh_max_depth_ = 0;
        relin(new_root, 0);
        order.swap(new_order);
        double sa_after = 0;
        for (int i = 0; i < M; ++i)
            if (!bn[i].leaf) sa_after += bn[i].box.area();
        if (getenv("WTGPU_BVH_VERBOSE")) fprintf(stderr, "[wtgpu bvh] reinsertion: %d passes, inner surface area %.6g -> %.6g (%.1f %%)\n", reinsert_passes, sa_before, sa_after, 100.0 * sa_after / sa_before);
        root = new_root;
    }

    // triangles in BVH order
    tri_geo_.resize(N);
    tri_meta_.resize(N);
    tri_shade_.resize(N);
    std::vector<uint32_t> tuid_of(N);
    for (uint32_t i = 0; i < N; ++i) {
        const wtri_t& w = wtris_[order[i]];
        tuid_of[order[i]] = i;
        tri_geo_[i] = tri_geo_t{w.a, w.b, w.c, w.n};
        tri_meta_[i] = tri_meta_t{w.shape, w.shape_tri, {kInvalid, kInvalid, kInvalid}};
        tri_shade_t sh{};
        sh.n0 = w.n0;
        sh.n1 = w.n1;
        sh.n2 = w.n2;
        sh.uv0 = w.uv0;
        sh.uv1 = w.uv1;
        sh.uv2 = w.uv2;
        sh.has_uv = w.has_uv;
        // surface_differentials_for_triangle (mesh/surface_differentials.hpp:19-52)
        {
            const vec3 dp02 = w.a - w.c, dp12 = w.b - w.c;
            const vec2 duv02 = w.uv0 - w.uv2, duv12 = w.uv1 - w.uv2;
            const float det = diff_prod(duv02.x, duv12.y, duv02.y, duv12.x);
            const vec3 ng = w.n;
            if (std::fabs(det) < 1e-10f) {
                if (std::fabs(ng.x) > std::fabs(ng.y))
                    sh.dpdu = vec3{-ng.z, 0, ng.x} / std::sqrt(sqr(ng.x) + sqr(ng.z));
                else
                    sh.dpdu = vec3{0, ng.z, -ng.y} / std::sqrt(sqr(ng.y) + sqr(ng.z));
            } else {
                const float rd = 1.f / det;
                sh.dpdu = vec3{diff_prod(duv12.y, dp02.x, duv02.y, dp12.x), diff_prod(duv12.y, dp02.y, duv02.y, dp12.y),
                               diff_prod(duv12.y, dp02.z, duv02.y, dp12.z)} * rd;
            }
        }
        tri_shade_[i] = sh;
    }
    // shape -> tuid map and area cdfs (src/scene/shape.cpp:35-56)
    shapes_.clear();
    shape_tri_tuid_.clear();
    shape_tri_cdf_.clear();
    for (size_t s = 0; s < shape_recs_.size(); ++s) {
        const auto& r = shape_recs_[s];
        shape_t sh{};
        sh.material = r.material;
        sh.emitter = r.emitter;
        sh.tri_offset = (uint32_t)shape_tri_tuid_.size();
        sh.tri_count = r.tri_count;
        std::vector<double> areas(r.tri_count);
        double total = 0;
        for (uint32_t t = 0; t < r.tri_count; ++t) {
            const wtri_t& w = wtris_[r.tri_begin + t];
            shape_tri_tuid_.push_back(tuid_of[r.tri_begin + t]);
            areas[t] = 0.5 * length(cross(w.b - w.a, w.c - w.a));
            total += areas[t];
        }
        sh.surface_area = (float)total;
        sh.recp_surface_area = (float)(1.0 / total);
        double acc = 0;
        shape_tri_cdf_.push_back(0.f);
        for (uint32_t t = 0; t < r.tri_count; ++t) {
            acc += areas[t];
            shape_tri_cdf_.push_back(t + 1 == r.tri_count ? 1.f : (float)(acc / total));
        }
        shapes_.push_back(sh);
    }

    // collapse into 8-wide nodes (src/ads/bvh8w_constructor.cpp:27-103)
    nodes_.clear();
    leaves_.clear();
    if (root < 0) return;
    std::function<void(int, int, std::vector<int>&)> extract = [&](int n, int depth, std::vector<int>& out) {
        if (bn[n].leaf || depth == 0) {
            out.push_back(n);
            return;
        }
        extract(bn[n].left, depth - 1, out);
        extract(bn[n].right, depth - 1, out);
    };
    // ---- which binary subtrees become the children of an 8-wide node.
    // collapse mode 0: three binary levels per node (src/ads/bvh8w_constructor.cpp:27-103): 4.5 of the 8 slots used on average in the
    //   283 K-triangle scene; mode 1: greedy (open the inner child with the largest box until 8); mode 2 (default): the SAH-optimal
    //   grouping by dynamic programming (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through compressed
    //   wide BVHs", §3.1): C(n, j) = cheapest way to cover the subtree of n with at most j roots,
    //       C(n, 1) = min( leaf: A(n) P(n) c_prim  [P(n) <= max leaf],  inner: A(n) c_node + D(n, 8) ),
    //       D(n, j) = min_k C(left, k) + C(right, j - k),        C(n, j >= 2) = min( C(n, 1), D(n, j) ),
    //   which also decides where small subtrees are better kept as ONE leaf of up to `wide_max_leaf` triangles (their triangles are
    //   contiguous).  The reference itself notes "bvh8w: create better trees" as a TODO; the tree only changes how fast a query is
    //   answered, never the answer (same triangles, same order).
    const int collapse_mode = getenv("WTGPU_BVH_COLLAPSE") ? atoi(getenv("WTGPU_BVH_COLLAPSE")) : 2;
    const uint32_t wide_max_leaf = getenv("WTGPU_BVH_MAX_LEAF") ? (uint32_t)std::min(7, std::max(1, atoi(getenv("WTGPU_BVH_MAX_LEAF")))) : 4u;
    const float c_node = 1.f, c_prim = getenv("WTGPU_BVH_CPRIM") ? (float)atof(getenv("WTGPU_BVH_CPRIM")) : .3f;
    struct dp_t {
        float c[9];        // c[j], j = 1..8
        uint8_t k[9];      // k[j]: 0 = one root (see `as_leaf`), else the left share of D(n, j)
        uint8_t as_leaf;   // C(n, 1) is the leaf alternative
    };
    std::vector<dp_t> dp;
    if (collapse_mode == 2) {
        dp.resize(bn.size());
        std::function<void(int)> solve = [&](int n) {
            dp_t& d = dp[n];
            const float A = bn[n].box.area();
            if (bn[n].leaf) {
                for (int j = 1; j <= 8; ++j) {
                    d.c[j] = A * float(bn[n].count) * c_prim;
                    d.k[j] = 0;
                }
                d.as_leaf = 1;
                return;
            }
            const int l = bn[n].left, r = bn[n].right;
            solve(l);
            solve(r);
            float D[9];
            uint8_t K[9];
            for (int j = 2; j <= 8; ++j) {
                D[j] = WT_INF;
                K[j] = 1;
                for (int k = 1; k < j; ++k) {
                    const float v = dp[l].c[k] + dp[r].c[j - k];
                    if (v < D[j]) {
                        D[j] = v;
                        K[j] = (uint8_t)k;
                    }
                }
            }
            const float inner = A * c_node + D[8];
            const float leafc = bn[n].count <= wide_max_leaf ? A * float(bn[n].count) * c_prim : WT_INF;
            d.as_leaf = leafc <= inner ? 1 : 0;
            d.c[1] = d.as_leaf ? leafc : inner;
            d.k[1] = 0;
            for (int j = 2; j <= 8; ++j) {
                if (D[j] < d.c[1]) {
                    d.c[j] = D[j];
                    d.k[j] = K[j];
                } else {
                    d.c[j] = d.c[1];
                    d.k[j] = 0;
                }
            }
            // the children of n, should n become an inner 8-wide node (the root always does)
            d.k[0] = K[8];
        };
        solve(root);
    }
    // child list of the wide node rooted at binary node n: (binary node, is a leaf of the wide tree)
    std::function<void(int, int, std::vector<std::pair<int, bool>>&)> gather = [&](int m, int j, std::vector<std::pair<int, bool>>& out) {
        const dp_t& d = dp[m];
        if (bn[m].leaf || d.k[j] == 0) {
            out.push_back({m, bn[m].leaf || d.as_leaf != 0});
            return;
        }
        gather(bn[m].left, d.k[j], out);
        gather(bn[m].right, j - d.k[j], out);
    };
    std::function<uint32_t(int)> emit = [&](int bnode) -> uint32_t {
        const uint32_t idx = (uint32_t)nodes_.size();
        nodes_.emplace_back();
        std::vector<std::pair<int, bool>> chl;
        if (bn[bnode].leaf)
            chl.push_back({bnode, true});
        else if (collapse_mode == 2) {
            gather(bn[bnode].left, dp[bnode].k[0], chl);
            gather(bn[bnode].right, 8 - dp[bnode].k[0], chl);
        } else {
            std::vector<int> ch;
            if (collapse_mode == 0)
                extract(bnode, 3, ch);
            else {
                ch = {bn[bnode].left, bn[bnode].right};
                while (ch.size() < 8) {
                    int best = -1;
                    float best_area = -1.f;
                    for (size_t i = 0; i < ch.size(); ++i)
                        if (!bn[ch[i]].leaf && bn[ch[i]].box.area() > best_area) {
                            best_area = bn[ch[i]].box.area();
                            best = (int)i;
                        }
                    if (best < 0) break;
                    const int n = ch[best];
                    ch[best] = bn[n].left;                      // keeps the left-to-right (triangle) order of the children
                    ch.insert(ch.begin() + best + 1, bn[n].right);
                }
            }
            for (int c : ch) chl.push_back({c, bn[c].leaf});
        }
        bvh8_node_t nd{};
        nd.tris_start = bn[bnode].first;
        nd.tris_count = bn[bnode].count;
        for (int i = 0; i < 8; ++i) {
            nd.child[i] = 0;
            nd.minx[i] = nd.miny[i] = nd.minz[i] = WT_INF;
            nd.maxx[i] = nd.maxy[i] = nd.maxz[i] = -WT_INF;
        }
        std::vector<std::pair<int, int>> todo;
        for (size_t c = 0; c < chl.size(); ++c) {
            const bnode_t& b = bn[chl[c].first];
            nd.minx[c] = b.box.mn[0];
            nd.miny[c] = b.box.mn[1];
            nd.minz[c] = b.box.mn[2];
            nd.maxx[c] = b.box.mx[0];
            nd.maxy[c] = b.box.mx[1];
            nd.maxz[c] = b.box.mx[2];
            if (chl[c].second) {
                leaves_.push_back(bvh8_leaf_t{b.first, b.count});
                if (b.count == 0 || b.count > 7 || b.first >= (1u << 28)) throw std::runtime_error("BVH leaf does not fit the by-value child reference");
                nd.child[c] = -(int32_t)((b.first << 3) | b.count);   // leaf named by value (wt/scene.h)
            } else
                todo.push_back({(int)c, chl[c].first});
        }
        nodes_[idx] = nd;
        for (auto& t : todo) {
            const uint32_t ci = emit(t.second);
            nodes_[idx].child[t.first] = (int32_t)ci + 1;
        }
        return idx;
    };
    emit(root);
}

// include/wt/ads/edge_classification.hpp:31-238 — neighbours are found by exact vertex-position equality
void scene_builder_t::build_edges() {
    edges_.clear();
    struct key_t {
        uint32_t v[6];
        bool operator<(const key_t& o) const { return std::memcmp(v, o.v, sizeof(v)) < 0; }
    };
    auto mk = [](vec3 p, vec3 q) {
        key_t k;
        uint32_t a[3], b[3];
        std::memcpy(a, &p, 12);
        std::memcpy(b, &q, 12);
        // canonicalise -0 -> +0
        for (int i = 0; i < 3; ++i) {
            if (a[i] == 0x80000000u) a[i] = 0;
            if (b[i] == 0x80000000u) b[i] = 0;
        }
        if (std::memcmp(a, b, 12) > 0) std::swap_ranges(a, a + 3, b);
        std::memcpy(k.v, a, 12);
        std::memcpy(k.v + 3, b, 12);
        return k;
    };
    struct ref_t {
        uint32_t tuid;
        int side;
    };
    std::map<key_t, std::vector<ref_t>> emap;
    const uint32_t N = (uint32_t)tri_geo_.size();
    auto verts = [&](uint32_t t, int side, vec3& a, vec3& b, vec3& c) {
        const tri_geo_t& g = tri_geo_[t];
        if (side == 0) { a = g.a; b = g.b; c = g.c; }
        else if (side == 1) { a = g.b; b = g.c; c = g.a; }
        else { a = g.c; b = g.a; c = g.b; }
    };
    for (uint32_t t = 0; t < N; ++t)
        for (int s = 0; s < 3; ++s) {
            vec3 a, b, c;
            verts(t, s, a, b, c);
            emap[mk(a, b)].push_back({t, s});
        }
    auto edge_for = [&](uint32_t t1, int s1, bool has2, uint32_t t2, int s2, edge_t& out) -> bool {
        vec3 a, b, c1;
        verts(t1, s1, a, b, c1);
        vec3 n1 = tri_geo_[t1].n;
        vec3 n2 = has2 ? tri_geo_[t2].n : -n1;
        const vec3 e = normalize(b - a);
        const vec3 m = (a + b) / 2.f;
        vec3 tt1{0, 0, 1}, tt2{0, 0, 1};
        if (has2) {
            vec3 a2, b2, c2;
            verts(t2, s2, a2, b2, c2);
            const bool concave1 = dot(n1, c2 - m) > 0.f;
            const bool concave2 = dot(n2, c1 - m) > 0.f;
            if (concave1 != concave2) return false;   // inconsistent normals
            if (concave1 && concave2) {
                n1 = -n1;
                n2 = -n2;
            }
            tt2 = cross(n2, e);
            if (dot(tt2, c2 - m) < 0.f) tt2 = -tt2;
        }
        tt1 = cross(n1, e);
        if (dot(tt1, c1 - m) < 0.f) tt1 = -tt1;
        if (!has2) tt2 = tt1;
        const float alpha = std::max(0.f, kPi - std::acos(clampf(dot(n1, n2), -1.f, 1.f)));
        if (alpha > 160.f / 180.f * kPi) return false;
        out = edge_t{a, b, e, n1, tt1, n2, tt2, alpha, t1, has2 ? t2 : kInvalid};
        return true;
    };
    // deterministic ids: triangles in tuid order, sides ab, bc, ca; the least tuid creates the shared edge
    for (uint32_t t = 0; t < N; ++t)
        for (int s = 0; s < 3; ++s) {
            vec3 a, b, c;
            verts(t, s, a, b, c);
            auto& refs = emap[mk(a, b)];
            // refs are sorted by (tuid, side) by construction
            if (refs.size() >= 2) {
                if (refs[0].tuid != t || refs[0].side != s) continue;   // only the first reference creates
                if (refs[0].tuid == refs[1].tuid) continue;             // degenerate
                edge_t e;
                if (edge_for(refs[0].tuid, refs[0].side, true, refs[1].tuid, refs[1].side, e)) {
                    const uint32_t id = (uint32_t)edges_.size();
                    edges_.push_back(e);
                    tri_meta_[refs[0].tuid].edge[refs[0].side] = id;
                    tri_meta_[refs[1].tuid].edge[refs[1].side] = id;
                }
            } else {
                edge_t e;
                if (edge_for(t, s, false, 0, 0, e)) {
                    const uint32_t id = (uint32_t)edges_.size();
                    edges_.push_back(e);
                    tri_meta_[t].edge[s] = id;
                }
            }
        }
}

// src/scene/scene_build_sensor_sampling_data.cpp:40-150: emitter selection pmf ∝ ∫ emission x sensitivity, and
// per-emitter spectral sampling distribution ∝ emission(k) x sensitivity(k)
void scene_builder_t::build_sampling_tables() {
    kdists_.clear();
    kdist_data_.clear();
    std::vector<double> powers;
    const spectrum_t sens = spectra_[sensitivity_spec_];
    const int NK = 4096;
    // the sensitivity spectrum's own support (sensitivity_spectrum().wavenumber_range(): the CMF data cover 390..830 nm, the baked table
    // is zero-padded to 340..840 nm)
    double sens_support_kmin = sens.kmin, sens_support_kmax = sens.kmax;
    if (sens.type == SPEC_TABLE) {
        const int N = 4096;
        int first = -1, last = -1;
        for (int i = 0; i < N; ++i) {
            const double k = sens.kmin + (double(sens.kmax) - sens.kmin) * i / (N - 1);
            if (spectrum_eval(sensitivity_spec_, (float)k) > 0.f) {
                if (first < 0) first = i;
                last = i;
            }
        }
        if (first >= 0) {
            sens_support_kmin = sens.kmin + (double(sens.kmax) - sens.kmin) * std::max(0, first - 1) / (N - 1);
            sens_support_kmax = sens.kmin + (double(sens.kmax) - sens.kmin) * std::min(N - 1, last + 1) / (N - 1);
        }
    }
    for (auto& e : emitters_) {
        const spectrum_t es = spectra_[e.spectrum];
        double geom = 1.0;
        if (e.type == EMIT_SPOT)
            geom = kTwoPi * (1.0 - .5 * (e.cos_cutoff + e.cos_falloff));   // spot_solid_angle
        else if (e.type == EMIT_POINT)
            geom = 4.0 * M_PI;   // point.hpp:60-69
        else if (e.type == EMIT_DIRECTIONAL)
            geom = e.target_area;   // directional.hpp:101-103: irradiance x target area
        else {
            double area = 0;
            const auto& r = shape_recs_[e.shape];
            for (uint32_t t = 0; t < r.tri_count; ++t) {
                const wtri_t& w = wtris_[r.tri_begin + t];
                area += 0.5 * length(cross(w.b - w.a, w.c - w.a));
            }
            geom = area * M_PI;
        }
        kdist_t kd{};
        double power = 0;
        if (sens.type == SPEC_DISCRETE || es.type == SPEC_DISCRETE) {
            const float k0 = sens.type == SPEC_DISCRETE ? sens.kmin : es.kmin;
            kd.discrete = 1;
            kd.kmin = kd.kmax = k0;
            const double sv = sens.type == SPEC_DISCRETE ? 1.0 : spectrum_eval(sensitivity_spec_, k0);
            power = sv * spectrum_eval(e.spectrum, k0) * e.scale * geom;
            if (sens.type == SPEC_DISCRETE && es.type == SPEC_DISCRETE && sens.kmin != es.kmin) power = 0;
        } else {
            float kmin = sens.kmin, kmax = sens.kmax;
            if (es.type == SPEC_TABLE) {
                kmin = std::max(kmin, es.kmin);
                kmax = std::min(kmax, es.kmax);
            }
            kd.discrete = 0;
            kd.kmin = kmin;
            kd.kmax = kmax;
            kd.offset = (uint32_t)kdist_data_.size();
            kd.count = NK;
            std::vector<double> pdf(NK), cdf(NK);
            const double dk = (double(kmax) - kmin) / (NK - 1);
            for (int i = 0; i < NK; ++i) {
                const float k = (float)(kmin + dk * i);
                pdf[i] = std::max(0.f, spectrum_eval(sensitivity_spec_, k)) * std::max(0.f, spectrum_eval(e.spectrum, k));
            }
            cdf[0] = 0;
            for (int i = 1; i < NK; ++i) cdf[i] = cdf[i - 1] + .5 * (pdf[i] + pdf[i - 1]) * dk;
            const double total = cdf[NK - 1];
            // The reference's weight is R0 x power(sensitivity range) with R0 = ∫ s^ e^ dk of the two spectra NORMALISED over their own
            // supports (product_distribution.hpp:39-43): ∝ ∫ s e dk x [∫_range e dk / ∫_all e dk] x geometry; 1 / ∫ s is common to
            // all emitters.  The bracket is 1 for a lamp spectrum inside the visible band and ~0.3 for a blackbody tabulated from
            // 10 nm to 5 mm.
            const double in_range = emission_fraction_in_range(e.spectrum, sens_support_kmin, sens_support_kmax);
            emitter_in_range_.resize(&e - emitters_.data() + 1, 1.0);
            emitter_in_range_.back() = in_range;
            power = total * e.scale * geom * in_range;
            for (int i = 0; i < NK; ++i) kdist_data_.push_back(total > 0 ? (float)(pdf[i] / total) : 0.f);
            for (int i = 0; i < NK; ++i) kdist_data_.push_back(total > 0 ? (i == NK - 1 ? 1.f : (float)(cdf[i] / total)) : 0.f);
        }
        e.k_dist = (int)kdists_.size();
        kdists_.push_back(kd);
        powers.push_back(std::max(0.0, power));
    }
    double total = 0;
    for (double p : powers) total += p;
    if (!(total > 0)) throw std::runtime_error("no overlap between emitters' spectra and the sensor sensitivity");
    emitter_cdf_.assign(1, 0.f);
    double acc = 0;
    for (size_t i = 0; i < emitters_.size(); ++i) {
        emitters_[i].select_pmf = (float)(powers[i] / total);
        acc += powers[i];
        emitter_cdf_.push_back(i + 1 == emitters_.size() ? 1.f : (float)(acc / total));
    }
}

// Regenerates the inverse-CDF tables of chi_e*|alpha_1|^2 and chi_e*|alpha_2|^2 (the reference ships them as
// Git-LFS binaries that are absent here).  Layout and use: fsd_lut.hpp:27-69.
//
// Mask argument.  The only numbers the reference holds about its tables are the lobe powers PA1 = 0.00493610757945... and
// PA2 = 0.218997893980... (fsd.hpp:59-61, "power contained in chi_e x |alpha|^2").  With the mask evaluated at zeta itself the
// two integrals are 0.0048274 / 0.16279 (2 % / 26 % low).  Solving  int chi_e(c |zeta|^2) |alpha_j|^2 dzeta = PAj  for the mask
// constant c INDEPENDENTLY for j = 1 and j = 2 gives c = 3.527899 and c = 3.527895: one and the same constant,
// c = (17/4) * 0.830092714835359 = 3.527894, i.e. the tables' mask is chi_e(sqrt(17)/2 * zeta).  With it both constants are
// reproduced to 1e-6 relative (tests/test_kat.py::test_fsd_lut_mask_reproduces_reference_lobe_powers, scipy quadrature), so
// the proposal these tables draw from and the normalisation 1/PAj the sampler applies (fsd.h: fsd_Pj) agree like in the reference.
void scene_builder_t::build_fsd_lut() {
    const uint32_t M = lut_m_, NT = lut_n_theta_;
    const int J = 8000;
    const double r0 = 1e-4, Rmax = 4e4;   // the power beyond |zeta| = 4e4 is 1e-5 (alpha_1) / 1e-4 (alpha_2: 1/r tail along the y axis) of the lobe's
    std::vector<double> rs(J), lw(J);
    for (int j = 0; j < J; ++j) rs[j] = r0 * std::pow(Rmax / r0, double(j) / (J - 1));
    const double dl = std::log(Rmax / r0) / (J - 1);
    auto chi_e = [](double r2) {
        const double t = 1 + kFsdLutMaskScale2 * 0.830092714835359 * r2;
        return std::max(0.0, 1 - (3 / (t * t) - 2 / (t * t * t)));
    };
    auto sinc = [](double x) { return std::fabs(x) < 1e-8 ? 1.0 : std::sin(x) / x; };
    auto a1 = [&](double x, double y) { return x == 0 ? 0.0 : (1 / (2 * M_PI)) * y / (x * (x * x + y * y)) * (std::cos(x / 2) - sinc(x / 2)); };
    auto a2 = [&](double x, double y) { return x == 0 ? 0.0 : (1 / (2 * M_PI)) * y / (x * x + y * y) * sinc(x / 2); };
    for (int which = 0; which < 2; ++which) {
        std::vector<float>& lut = which == 0 ? lut1_ : lut2_;
        std::vector<float>& lutt = which == 0 ? lut_theta1_ : lut_theta2_;
        lut.assign((size_t)M * M, 0.f);
        std::vector<double> marg(M);
        std::vector<double> cdf(J);
        for (uint32_t ti = 0; ti < M; ++ti) {
            // the last row (theta = pi/2, x = 0) is a removable singularity of alpha: evaluate just inside
            double th = (double(ti) / (M - 1)) * (M_PI / 2);
            if (ti == M - 1) th = (double(ti) - 0.25) / (M - 1) * (M_PI / 2);
            if (ti == 0) th = 0.25 / (M - 1) * (M_PI / 2);
            const double c = std::cos(th), s = std::sin(th);
            cdf[0] = 0;
            double prev = 0;
            for (int j = 0; j < J; ++j) {
                const double r = rs[j];
                const double a = which == 0 ? a1(r * c, r * s) : a2(r * c, r * s);
                const double f = chi_e(r * r) * a * a * r * r;   // density * r (polar) * r (d ln r)
                if (j > 0) cdf[j] = cdf[j - 1] + .5 * (f + prev) * dl;
                prev = f;
            }
            marg[ti] = cdf[J - 1];
            // invert onto M uniform u values
            int j = 0;
            for (uint32_t ui = 0; ui < M; ++ui) {
                const double target = (double(ui) / (M - 1)) * cdf[J - 1];
                while (j < J - 2 && cdf[j + 1] < target) ++j;
                const double d = cdf[j + 1] - cdf[j];
                const double f = d > 0 ? (target - cdf[j]) / d : 0.0;
                lut[(size_t)ti * M + ui] = (float)(rs[j] * std::pow(rs[j + 1] / rs[j], std::min(1.0, std::max(0.0, f))));
            }
            lut[(size_t)ti * M + 0] = 0.f;
        }
        // theta marginal -> inverse cdf
        std::vector<double> tc(M);
        tc[0] = 0;
        const double dth = (M_PI / 2) / (M - 1);
        for (uint32_t i = 1; i < M; ++i) tc[i] = tc[i - 1] + .5 * (marg[i] + marg[i - 1]) * dth;
        lutt.assign(NT, 0.f);
        uint32_t j = 0;
        for (uint32_t ui = 0; ui < NT; ++ui) {
            const double target = (double(ui) / (NT - 1)) * tc[M - 1];
            while (j < M - 2 && tc[j + 1] < target) ++j;
            const double d = tc[j + 1] - tc[j];
            const double f = d > 0 ? (target - tc[j]) / d : 0.0;
            lutt[ui] = (float)((j + std::min(1.0, std::max(0.0, f))) * dth);
        }
        // 4 quadrants: total power, compare with PA1/PA2 (fsd.hpp:59-61)
        lut_power_[which] = 4.0 * tc[M - 1];
    }
}

const scene_t& scene_builder_t::finalize() {
    if (finalized_) return sc_;
    if (sensitivity_spec_ < 0) throw std::runtime_error("sensor response not set");
    build_bvh();
    build_edges();
    // per-child "subtree has classified edges" masks (every subtree owns a contiguous triangle range)
    {
        std::vector<uint32_t> pre(tri_meta_.size() + 1, 0);
        for (size_t t = 0; t < tri_meta_.size(); ++t) {
            const tri_meta_t& m = tri_meta_[t];
            pre[t + 1] = pre[t] + ((m.edge[0] != kInvalid || m.edge[1] != kInvalid || m.edge[2] != kInvalid) ? 1u : 0u);
        }
        for (auto& nd : nodes_) {
            nd.edge_mask = 0;
            for (int c = 0; c < 8; ++c) {
                const int32_t cp = nd.child[c];
                if (cp == 0) continue;
                uint32_t t0, cnt;
                if (cp < 0) {
                    t0 = (uint32_t)(-cp) >> 3;
                    cnt = (uint32_t)(-cp) & 7u;
                } else {
                    t0 = nodes_[cp - 1].tris_start;
                    cnt = nodes_[cp - 1].tris_count;
                }
                if (pre[t0 + cnt] - pre[t0] > 0) nd.edge_mask |= 1u << c;
            }
        }
    }
    // infinite emitters need the world AABB (src/scene/scene.cpp:50-58, directional_t::set_world_aabb, directional.hpp:46-75)
    {
        vec3 mn{WT_INF, WT_INF, WT_INF}, mx{-WT_INF, -WT_INF, -WT_INF};
        for (auto& g : tri_geo_) {
            mn = vmin(mn, vmin(g.a, vmin(g.b, g.c)));
            mx = vmax(mx, vmax(g.a, vmax(g.b, g.c)));
        }
        for (auto& e : emitters_) {
            if (e.type != EMIT_DIRECTIONAL) continue;
            e.position = (mn + mx) / 2.f;
            const vec3 pr = (mx - mn) / 2.f;
            float r2 = 0.f, zmax = 0.f;
            for (int yz = 0; yz < 4; ++yz) {
                const vec3 c = to_local(e.frame, vec3{-pr.x, (yz & 2) ? pr.y : -pr.y, (yz & 1) ? pr.z : -pr.z});
                r2 = std::max(r2, c.x * c.x + c.y * c.y);
                zmax = std::max(zmax, std::fabs(c.z));
            }
            e.target_radius = std::sqrt(r2);
            e.target_area = kPi * r2;
            e.far_dist = 1.01f * zmax;
        }
    }
    build_sampling_tables();
    if (sc_.opts.FSD && !sc_.opts.force_ray_tracing && sc_.opts.integrator == INTEGRATOR_BDPT) build_fsd_lut();   // plt_path diffracts with UTD

    sensor_t& s = sc_.sensor;
    s.rfilter_sigma = .25f * rfilter_scale_;
    s.rf_radius = (int)(uint32_t)(std::ceil(s.rfilter_sigma * 3.f) + .5f);
    if (s.rf_radius > 2) s.rf_radius = 2;

    sc_.tri_geo = tri_geo_.data();
    sc_.tri_meta = tri_meta_.data();
    sc_.tri_shade = tri_shade_.data();
    sc_.n_tris = (uint32_t)tri_geo_.size();
    sc_.edges = edges_.data();
    sc_.n_edges = (uint32_t)edges_.size();
    sc_.nodes = nodes_.data();
    sc_.n_nodes = (uint32_t)nodes_.size();
    sc_.leaves = leaves_.data();
    sc_.n_leaves = (uint32_t)leaves_.size();
    vec3 mn{WT_INF, WT_INF, WT_INF}, mx{-WT_INF, -WT_INF, -WT_INF};
    for (auto& g : tri_geo_) {
        mn = vmin(mn, vmin(g.a, vmin(g.b, g.c)));
        mx = vmax(mx, vmax(g.a, vmax(g.b, g.c)));
    }
    sc_.world_min = mn;
    sc_.world_max = mx;
    sc_.shapes = shapes_.data();
    sc_.n_shapes = (uint32_t)shapes_.size();
    sc_.shape_tri_tuid = shape_tri_tuid_.data();
    sc_.shape_tri_cdf = shape_tri_cdf_.data();
    sc_.materials = materials_.data();
    sc_.n_materials = (uint32_t)materials_.size();
    sc_.spectra = spectra_.data();
    sc_.n_spectra = (uint32_t)spectra_.size();
    sc_.spectra_data = spectra_data_.data();
    sc_.textures = textures_.data();
    sc_.n_textures = (uint32_t)textures_.size();
    sc_.texture_data = texture_data_.data();
    sc_.emitters = emitters_.data();
    sc_.n_emitters = (uint32_t)emitters_.size();
    sc_.emitter_cdf = emitter_cdf_.data();
    sc_.kdists = kdists_.data();
    sc_.kdist_data = kdist_data_.data();
    sc_.lut.n_theta = (uint32_t)lut_theta1_.size();
    sc_.lut.m = lut1_.empty() ? 0 : lut_m_;
    sc_.lut.icdf_theta1 = lut_theta1_.data();
    sc_.lut.icdf_theta2 = lut_theta2_.data();
    sc_.lut.icdf1 = lut1_.data();
    sc_.lut.icdf2 = lut2_.data();
    finalized_ = true;
    return sc_;
}

std::string scene_builder_t::stats() const {
    std::ostringstream o;
    o << "{\"tris\": " << tri_geo_.size() << ", \"edges\": " << edges_.size() << ", \"nodes8\": " << nodes_.size() << ", \"leaves\": " << leaves_.size()
      << ", \"shapes\": " << shapes_.size() << ", \"emitters\": " << emitters_.size() << ", \"binary_depth\": " << bvh_max_depth_
      << ", \"fsd_lut_power\": [" << lut_power_[0] << ", " << lut_power_[1] << "], \"emitter_list\": [";
    static const char* names[] = {"spot", "area", "point", "directional"};
    for (size_t i = 0; i < emitters_.size(); ++i) {
        const emitter_t& e = emitters_[i];
        o << (i ? ", " : "") << "{\"type\": \"" << names[e.type & 3] << "\", \"cutoff_deg\": " << e.cutoff * 180.0 / M_PI << ", \"shape\": " << e.shape
          << ", \"select_pmf\": " << e.select_pmf << ", \"in_range_fraction\": " << (i < emitter_in_range_.size() ? emitter_in_range_[i] : 1.0)
          << ", \"falloff_deg\": " << e.falloff * 180.0 / M_PI << ", \"scale\": " << e.scale << ", \"phase_space_extent_scale\": " << e.phase_space_extent_scale
          << ", \"position\": [" << e.position.x << ", " << e.position.y << ", " << e.position.z << "], \"direction\": [" << e.frame.n.x << ", " << e.frame.n.y
          << ", " << e.frame.n.z << "]}";
    }
    o << "]}";
    return o.str();
}

}   // namespace wth
