// wave_tracer_amd — elliptic cone + ray/cone intersection primitives.
//
// Reference: include/wt/math/shapes/elliptic_cone.hpp:30-333, src/math/elliptic_cone.cpp:19-145,
//            include/wt/math/intersect/cone.hpp:38-128,170-258,550-626,
//            include/wt/math/intersect/ray.hpp:147-180, include/wt/math/intersect/misc.hpp,
//            include/wt/math/intersect/clip.hpp:35-83,
//            include/wt/math/intersect/cone_intersection_tolerance.hpp:23-41
#pragma once
#include "core.h"
#include "gauss.h"

namespace wt {

constexpr float kMajorAxisToZScale = 2.f;   // beam/beam_generic.hpp:50 (interaction-region depth = 2 x major axis)

// inclusive range (include/wt/math/range.hpp)
struct range_t {
    float min, max;
};
WT_HD range_t range_positive() { return {0.f, WT_INF}; }
WT_HD range_t range_all() { return {-WT_INF, WT_INF}; }
WT_HD range_t range_null() { return {WT_INF, -WT_INF}; }
WT_HD bool contains(const range_t& r, float pt) { return (pt < r.max && r.min < pt) || pt == r.min || pt == r.max; }
WT_HD bool empty(const range_t& r) {
    if (r.min == r.max && !finitef(r.min)) return true;
    return r.min > r.max;
}
WT_HD range_t grow(const range_t& r, float e) { return {r.min - e, r.max + e}; }
WT_HD range_t rand_(const range_t& a, const range_t& b) { return {fmaxf_(a.min, b.min), fminf_(a.max, b.max)}; }
WT_HD float centre(const range_t& r) { return (r.max + r.min) / 2.f; }
WT_HD float length(const range_t& r) { return r.max - r.min; }

struct ray_t {
    vec3 o, d;
};

// elliptic_cone_t (elliptic_cone.hpp:30-48): containment  x^2+(e*y)^2 <= (z*tan_alpha + x0)^2
struct cone_t {
    vec3 o, d;      // central ray
    vec3 x;         // tangent (local x axis, major axis direction)
    float x0;       // initial major-axis length
    float tan_alpha;
    float e;           // major/minor (>=1)
    float one_over_e;  // minor/major
    float z_apex;      // -x0/tan_alpha, or -inf for a degenerate ray
};

WT_HD float cone_z_apex(float x0, float ta) { return (x0 != 0.f || ta != 0.f) ? -x0 / ta : -WT_INF; }
WT_HD vec3 cone_y(const cone_t& c) { return cross(c.d, c.x); }
WT_HD frame_t cone_frame(const cone_t& c) { return frame_t{c.x, cone_y(c), c.d}; }
WT_HD bool cone_is_ray(const cone_t& c) { return c.tan_alpha == 0.f && c.x0 == 0.f; }
WT_HD vec2 cone_axes(const cone_t& c, float z) {
    const float r = c.tan_alpha * z + c.x0;
    return {r, r * c.one_over_e};
}
// public ctor #2 (elliptic_cone.hpp:64-81)
WT_HD cone_t make_cone(vec3 o, vec3 d, vec3 x, float tan_alpha, float eccentricity, float x0) {
    cone_t c;
    c.o = o;
    c.d = d;
    c.x = x;
    c.x0 = x0;
    c.one_over_e = sqrtf(fmaxf_(0.f, 1.f - sqr(eccentricity)));
    c.e = 1.f / c.one_over_e;
    c.tan_alpha = tan_alpha;
    c.z_apex = cone_z_apex(x0, tan_alpha);
    return c;
}
// public ctor #1 (elliptic_cone.hpp:50-54): isotropic
WT_HD cone_t make_cone_iso(vec3 o, vec3 d, float tan_alpha, float x0) {
    return make_cone(o, d, build_orthogonal_frame(d).t, tan_alpha, 0.f, x0);
}
// private ctor (elliptic_cone.hpp:313-330)
WT_HD cone_t make_cone_raw(vec3 o, vec3 d, vec3 x, float x0, float tan_alpha, float one_over_e, float e) {
    cone_t c;
    c.o = o;
    c.d = d;
    c.x = x;
    c.x0 = x0;
    c.tan_alpha = tan_alpha;
    c.one_over_e = one_over_e;
    c.e = e;
    c.z_apex = cone_z_apex(x0, tan_alpha);
    return c;
}
WT_HD void cone_set_x0(cone_t& c, float x0) {
    c.x0 = x0;
    c.z_apex = cone_z_apex(x0, c.tan_alpha);
}
WT_HD bool cone_contains_local(const cone_t& c, vec3 p, const range_t& range) {
    return contains(range, p.z) && c.z_apex <= p.z && sqr(p.x) + sqr(c.e * p.y) <= sqr(p.z * c.tan_alpha + c.x0);
}
// elliptic_cone.hpp:210-218
WT_HD vec2 cone_project_local(const cone_t& c, vec3 p, float z) {
    const vec2 xy{p.x, p.y};
    const float scale = (c.tan_alpha * z + c.x0) / fabsf(c.tan_alpha * p.z + c.x0);
    return (c.x0 == 0.f && c.tan_alpha == 0.f) ? xy : xy * scale;
}

// ---- ray primitives (ray.hpp) -----------------------------------------------------------------
struct ray_tri_hit_t {
    float dist;
    float bx, by;   // barycentric_t::bary = (1-(u+v), u): weights of vertices a and b
};
// Möller–Trumbore, scalar variant (ray.hpp:147-180)
WT_HD bool intersect_ray_tri(vec3 ro, vec3 rd, vec3 a, vec3 b, vec3 c, const range_t& range, ray_tri_hit_t& out) {
    const vec3 ray = ro - a;
    const vec3 e1 = b - a;
    const vec3 e2 = c - a;
    const vec3 crs = cross(rd, e2);
    float det = dot(e1, crs);
    if (det == 0.f) return false;
    const float sdet = det >= 0.f ? 1.f : -1.f;
    det *= sdet;
    const vec3 q = cross(ray, e1);
    const float qe2 = sdet * dot(q, e2);
    const float bx = sdet * dot(ray, crs);
    const float by = sdet * dot(rd, q);
    if (bx >= 0.f && by >= 0.f && bx + by <= det && contains(range_t{det * range.min, det * range.max}, qe2)) {
        const float recp_det = 1.f / det;
        out.dist = qe2 * recp_det;
        const float bux = bx * recp_det, buy = by * recp_det;
        out.bx = 1.f - (bux + buy);
        out.by = bux;
        return true;
    }
    return false;
}
// 8-wide variant semantics used by the BVH leaf test (ray.hpp:192-236): the divide happens first,
// barycentrics are tested as (1-(u+v))>=0, u>=0, v>=0, and range is tested on z.
WT_HD bool intersect_ray_tri_wide(vec3 ro, vec3 rd, vec3 a, vec3 b, vec3 c, const range_t& range, ray_tri_hit_t& out) {
    const vec3 ray = ro - a;
    const vec3 e1 = b - a;
    const vec3 e2 = c - a;
    const vec3 crs = cross(rd, e2);
    const float det = dot(e1, crs);
    const float recp_det = 1.f / det;
    bool valid = det != 0.f;
    const vec3 q = cross(ray, e1);
    const float qe2 = dot(q, e2);
    const float betax = dot(ray, crs);
    const float betay = dot(rd, q);
    const float z = qe2 * recp_det;
    const float baryy = betax * recp_det;
    const float baryz = betay * recp_det;
    const float baryx = 1.f - (baryy + baryz);
    valid = valid && baryx >= 0.f && baryy >= 0.f && baryz >= 0.f && z >= range.min && z <= range.max;
    out.dist = z;
    out.bx = baryx;
    out.by = baryy;
    return valid;
}
// test_ray_tri 8-wide (ray.hpp:100-135) — any-hit
WT_HD bool test_ray_tri_wide(vec3 ro, vec3 rd, vec3 a, vec3 b, vec3 c, const range_t& range) {
    const vec3 ray = ro - a;
    const vec3 e1 = b - a;
    const vec3 e2 = c - a;
    const vec3 crs = cross(rd, e2);
    const float det = dot(e1, crs);
    const float recp_det = 1.f / det;
    const vec3 q = cross(ray, e1);
    const float qe2 = dot(q, e2);
    const float betax = dot(ray, crs);
    const float betay = dot(rd, q);
    const float z = qe2 * recp_det;
    return det != 0.f && (betax * recp_det) >= 0.f && (betay * recp_det) >= 0.f && ((betax + betay) * recp_det) <= 1.f &&
           z >= range.min && z <= range.max;
}

// intersect_line_plane with plane z=const in local frame (ray.hpp:30-52 specialised)
WT_HD bool intersect_line_zplane(vec3 p0, vec3 p1, float zplane, float& t) {
    const float dn = p1.z - p0.z;
    if (dn == 0.f) return false;
    t = (zplane - p0.z) / dn;
    return true;
}
// intersect_edge_plane for plane z=zp, n=(0,0,1) (misc.hpp:142-158)
WT_HD bool intersect_edge_zplane(vec3 p0, vec3 p1, float zp, vec3& out) {
    const float d0 = zp - p0.z;
    const float d1 = zp - p1.z;
    const vec3 E = p1 - p0;
    const float EN = E.z;
    if (signf(d0) == signf(d1) || EN == 0.f) return false;
    const float d = d0 / EN;
    if (d >= 0.f && 1.f >= d) {
        out = p0 + d * E;
        return true;
    }
    return false;
}

// edge–ellipse (axis aligned) (misc.hpp:77-118)
struct edge_ellipse_t {
    int points;
    vec2 u1, u2;
    float t1, t2;
};
WT_HD edge_ellipse_t intersect_edge_ellipse(vec2 point0, vec2 point1, float rx, float ry) {
    edge_ellipse_t ret{0, {0, 0}, {0, 0}, 0.f, 0.f};
    const vec2 scale{rx, ry};
    const vec2 rs{1.f / rx, 1.f / ry};
    const vec2 p0 = point0 * rs, p1 = point1 * rs;
    const vec2 d = p1 - p0;
    const float a = dot(d, d);
    const float b = 2.f * dot(p0, d);
    const float c = dot(p0, p0) - 1.f;
    const float det2 = b * b - 4.f * a * c;
    if (det2 <= 0.f || a == 0.f) return ret;
    const float recp_a = 1.f / a;
    const float det = sqrtf(det2);
    float t1 = 0.5f * (-b - signf(b) * det) * recp_a;
    float t2 = t1 == 0.f ? -b * recp_a : c * recp_a / t1;
    if (t1 > t2) {
        const float t = t1;
        t1 = t2;
        t2 = t;
    }
    const bool u1v = (t1 >= 0.f && 1.f >= t1);
    const bool u2v = (t2 >= 0.f && 1.f >= t2);
    ret.t1 = t1;
    ret.t2 = t2;
    if (!u1v && !u2v) return ret;
    if (u1v && u2v) {
        ret.points = 2;
        ret.u1 = (p0 + t1 * d) * scale;
        ret.u2 = (p0 + t2 * d) * scale;
        return ret;
    }
    ret.points = 1;
    ret.u1 = (u1v ? p0 + t1 * d : p0 + t2 * d) * scale;
    ret.t1 = u1v ? t1 : t2;
    ret.t2 = u1v ? t2 : t1;
    return ret;
}

// is_point_in_triangle, 3-D variant (include/wt/math/util.hpp:88-107)
WT_HD bool is_point_in_triangle3(vec3 p, vec3 a, vec3 b, vec3 c) {
    const vec3 v0 = b - a, v1 = c - a, u = p - a;
    const float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1);
    const float d20 = dot(u, v0), d21 = dot(u, v1);
    const float d = diff_prod(d00, d11, d01, d01);
    const float sgn = d > 0.f ? 1.f : -1.f;
    const float alpha = diff_prod(d11, d20, d01, d21);
    const float beta = diff_prod(d00, d21, d01, d20);
    return sgn * alpha >= 0.f && sgn * beta >= 0.f && sgn * (alpha + beta) <= sgn * d;
}
// 2-D variant (util.hpp:69-82)
WT_HD bool is_point_in_triangle2(vec2 p, vec2 a, vec2 b, vec2 c) {
    const float s1 = diff_prod(p.x - b.x, a.y - b.y, a.x - b.x, p.y - b.y);
    const float s2 = diff_prod(p.x - c.x, b.y - c.y, b.x - c.x, p.y - c.y);
    const float s3 = diff_prod(p.x - a.x, c.y - a.y, c.x - a.x, p.y - a.y);
    const bool neg = s1 < 0 || s2 < 0 || s3 < 0;
    const bool pos = s1 > 0 || s2 > 0 || s3 > 0;
    return !(neg && pos);
}

// ---- cone ∩ edge, local-frame, segment variant with clip planes (cone.hpp:38-128) --------------
struct cone_edge_t {
    vec3 p0, p1;
    range_t range;
    int pts;
};
WT_HD bool intersect_cone_edge_local(const cone_t& cone, vec3 p0in, vec3 p1in, const range_t& range, cone_edge_t& ret) {
    vec3 lp0 = p0in, lp1 = p1in;
    const bool p0closer = lp1.z > lp0.z;
    if (!p0closer) {
        const vec3 t = lp0;
        lp0 = lp1;
        lp1 = t;
    }
    const vec3 p = lp0, l = lp1 - lp0;
    const float x0 = cone.x0, ta = cone.tan_alpha, e = cone.e;

    const float cs = p.z * ta + x0;
    const float epy = e * p.y;
    const float ely = e * l.y;
    const float lzta = l.z * ta;

    const float c = sqr(p.x) + diff_prod(epy, epy, cs, cs);
    const float b = 2.f * eft_dot(vec3{p.x, epy, -lzta}, vec3{l.x, ely, cs});
    const float a = sqr(l.x) + diff_prod(ely, ely, lzta, lzta);

    const float D = b * b - 4.f * a * c;
    if (D < 0.f) return false;

    const float sqrtD = sqrtf(D);
    float t1 = b >= 0.f ? (-b - sqrtD) / (2.f * a) : (-b + sqrtD) / (2.f * a);
    float t2 = (-b / a) - t1;

    const float zapex = cone.z_apex;
    if (p.z + t1 * l.z <= zapex) t1 = WT_INF;
    if (p.z + t2 * l.z < zapex) t2 = WT_INF;

    if (t2 < t1) {
        const float t = t1;
        t1 = t2;
        t2 = t;
    }
    float z1 = t1 < WT_INF ? p.z + t1 * l.z : -WT_INF;
    float z2 = t2 < WT_INF ? p.z + t2 * l.z : WT_INF;

    if (z1 > range.max || z2 < range.min || (!finitef(z1) && !finitef(z2))) return false;

    if (range.min > zapex && z1 < range.min) {
        float tmin;
        if (intersect_line_zplane(p, p + l, range.min, tmin)) {
            t1 = tmin;
            z1 = range.min;
        }
    }
    if (z2 > range.max) {
        float tmax;
        if (intersect_line_zplane(p, p + l, range.max, tmax)) {
            t2 = tmax;
            z2 = range.max;
        }
    }

    const vec3 base = p0closer ? p0in : p1in;
    const vec3 dir = p0closer ? p1in - p0in : p0in - p1in;
    bool has1 = false, has2 = false;
    vec3 v1{0, 0, 0}, v2{0, 0, 0};
    if (t1 >= 0.f && 1.f >= t1) {
        v1 = base + t1 * dir;
        has1 = true;
    } else
        z1 = z2;
    if (t2 >= 0.f && 1.f >= t2) {
        v2 = base + t2 * dir;
        has2 = true;
    } else
        z2 = z1;
    if (!has1 && !has2) return false;

    ret.range = {z1, z2};
    ret.pts = (has1 && has2) ? 2 : 1;
    ret.p0 = has1 ? v1 : v2;
    ret.p1 = v2;
    return true;
}

// ---- cone ∩ plane, local frame (cone.hpp:170-258) ------------------------------------------------
struct cone_plane_t {
    range_t range;
    vec3 near, far;
};
WT_HD vec3 closest_point_plane_plane_(float z, vec2 u, vec3 n, float d) {
    float x0, y0;
    if (fabsf(n.y) > fabsf(n.x)) {
        y0 = (d - n.z * z) / n.y;
        x0 = n.x != 0.f ? (d - n.z * z - n.y * y0) / n.x : 0.f;
    } else {
        x0 = (d - n.z * z) / n.x;
        y0 = n.y != 0.f ? (d - n.z * z - n.x * x0) / n.y : 0.f;
    }
    const float s = x0 * u.x + y0 * u.y;
    return vec3{s * u.x, s * u.y, z};
}
WT_HD cone_plane_t intersect_cone_plane_local(const cone_t& cone, vec3 n, float d, const range_t& range) {
    const float x0 = cone.x0;
    const float e = cone.one_over_e;
    const float v_denom2 = sqr(n.x) + sqr(e * n.y);
    const vec2 v = v_denom2 > 0.f ? vec2{n.x, e * n.y} / sqrtf(v_denom2) : vec2{0.f, 0.f};
    const vec2 u = v * vec2{1.f, e};
    const float nu = n.x * u.x + n.y * u.y;

    const float zapex = cone.z_apex;
    float z01 = (d - x0 * nu) / (n.z + cone.tan_alpha * nu);
    float z02 = (d + x0 * nu) / (n.z - cone.tan_alpha * nu);

    const bool has_z01 = z01 >= zapex && !(z01 != z01);
    const bool has_z02 = z02 >= zapex && !(z02 != z02);
    if (!has_z01) z01 = WT_INF;
    if (!has_z02) z02 = WT_INF;
    const vec3 inf3{WT_INF, WT_INF, WT_INF};
    vec3 p1 = inf3, p2 = inf3;
    if (has_z01) {
        const float r = z01 * cone.tan_alpha + x0;
        p1 = vec3{r * u.x, r * u.y, z01};
    }
    if (has_z02) {
        const float r = z02 * cone.tan_alpha + x0;
        p2 = vec3{-r * u.x, -r * u.y, z02};
    }
    if (z01 > z02) {
        const float t = z01;
        z01 = z02;
        z02 = t;
        const vec3 tp = p1;
        p1 = p2;
        p2 = tp;
    }
    range_t rng{z01, z02};
    const bool is_empty = (!has_z01 && !has_z02) || empty(rand_(rng, range));
    cone_plane_t ret;
    if (is_empty) {
        ret.range = range_null();
        ret.near = inf3;
        ret.far = inf3;
        return ret;
    }
    if (finitef(rng.min)) {
        if (rng.min < range.min) {
            p1 = closest_point_plane_plane_(range.min, v, n, d);
            rng.min = range.min;
        }
    }
    const bool has_infinite = has_z01 != has_z02;
    if (finitef(rng.max) || has_infinite) {
        if (rng.max > range.max) {
            p2 = closest_point_plane_plane_(range.max, v, n, d);
            rng.max = range.max;
        }
    }
    ret.range = rng;
    ret.near = p1;
    ret.far = p2;
    return ret;
}
// world-space variant (in_local=false): n,d given in world, points returned in world
WT_HD cone_plane_t intersect_cone_plane_world(const cone_t& cone, vec3 n, float d, const range_t& range) {
    const frame_t f = cone_frame(cone);
    d -= dot(cone.o, n);
    const vec3 ln = to_local(f, n);
    cone_plane_t r = intersect_cone_plane_local(cone, ln, d, range);
    if (!empty(r.range)) {
        if (finitef(r.near.z)) r.near = cone.o + to_world(f, r.near);
        if (finitef(r.far.z)) r.far = cone.o + to_world(f, r.far);
    }
    return r;
}

// ---- cone ∩ triangle: closest z (cone.hpp:550-626) ----------------------------------------------
struct cone_tri_hit_t {
    float dist;
    vec3 p;
};
// Squared distance from the origin to the segment [a,b] (2-D).
WT_HD float dist2_origin_segment2(vec2 a, vec2 b) {
    const vec2 ab = b - a;
    const float l2 = dot(ab, ab);
    const float t = l2 > 0.f ? clampf(-dot(a, ab) / l2, 0.f, 1.f) : 0.f;
    const vec2 p = a + t * ab;
    return dot(p, p);
}
// TRUE only if no point of the (local-frame) triangle `vs` can lie inside the cone at any z <= zhi: every point the exact
// tests of intersect_cone_tri accept lies in the cone's cross-section at its own z, whose radius in the metric
// sqrt(x^2+(e y)^2) is at most r(zhi) = zhi tan_alpha + x0 (tan_alpha >= 0).  So if the distance from the axis to the
// triangle's projection in that metric exceeds r(zhi), by a margin far above the exact tests' own tolerances, the triangle
// cannot intersect.  Conservative: a FALSE here decides nothing.
WT_HD bool cone_tri_laterally_outside(const cone_t& cone, const vec3 vs[3], float zhi) {
    const float r_hi = zhi * cone.tan_alpha + cone.x0;
    if (!(r_hi >= 0.f) || !finitef(r_hi)) return false;
    const vec2 p0{vs[0].x, cone.e * vs[0].y}, p1{vs[1].x, cone.e * vs[1].y}, p2{vs[2].x, cone.e * vs[2].y};
    // origin inside the projected triangle?
    const float c0 = p0.x * p1.y - p0.y * p1.x, c1 = p1.x * p2.y - p1.y * p2.x, c2 = p2.x * p0.y - p2.y * p0.x;
    if ((c0 >= 0.f && c1 >= 0.f && c2 >= 0.f) || (c0 <= 0.f && c1 <= 0.f && c2 <= 0.f)) return false;
    const float d2 = fminf_(dist2_origin_segment2(p0, p1), fminf_(dist2_origin_segment2(p1, p2), dist2_origin_segment2(p2, p0)));
    const float lim = r_hi * 1.002f + 1e-7f * (fabsf(p0.x) + fabsf(p0.y) + fabsf(p1.x) + fabsf(p1.y) + fabsf(p2.x) + fabsf(p2.y));
    return d2 > lim * lim;
}
// Smallest sphere around a triangle (centre xyz, radius w): the circumsphere of an acute triangle, otherwise the sphere on its longest edge;
// computed in double and rounded OUTWARDS (the radius is the largest float distance from the float centre to a vertex, plus a relative 1e-6).
WT_HD void tri_bounding_sphere(vec3 a, vec3 b, vec3 c, float out[4]) {
    const double ax = a.x, ay = a.y, az = a.z;
    const double abx = b.x - ax, aby = b.y - ay, abz = b.z - az, acx = c.x - ax, acy = c.y - ay, acz = c.z - az;
    const double d00 = abx * abx + aby * aby + abz * abz, d01 = abx * acx + aby * acy + abz * acz, d11 = acx * acx + acy * acy + acz * acz;
    // circumcentre a + s ab + t ac; outside the triangle (s < 0, t < 0 or s + t > 1) for obtuse triangles: then the midpoint of the longest edge
    const double den = 2.0 * (d00 * d11 - d01 * d01);
    double s = 0.5, t = 0.0;   // degenerate: midpoint of ab, the radius below covers every vertex anyway
    if (den > 0.0) {
        s = d11 * (d00 - d01) / den;
        t = d00 * (d11 - d01) / den;
        if (s < 0.0) { s = 0.0; t = 0.5; }                       // obtuse at b: the longest edge is ac
        else if (t < 0.0) { s = 0.5; t = 0.0; }                  // the longest edge is ab
        else if (s + t > 1.0) { s = 0.5; t = 0.5; }              // the longest edge is bc
    }
    const float cx = (float)(ax + s * abx + t * acx), cy = (float)(ay + s * aby + t * acy), cz = (float)(az + s * abz + t * acz);
    double r2 = 0.0;
    const vec3 vs[3] = {a, b, c};
    for (int i = 0; i < 3; ++i) {
        const double dx = (double)vs[i].x - cx, dy = (double)vs[i].y - cy, dz = (double)vs[i].z - cz;
        const double q = dx * dx + dy * dy + dz * dz;
        r2 = q > r2 ? q : r2;
    }
    const double r = sqrt(r2);
    out[0] = cx;
    out[1] = cy;
    out[2] = cz;
    out[3] = (float)(r * (1.0 + 1e-6)) + 1e-30f;
    if ((double)out[3] < r) out[3] = nextafterf(out[3], WT_INF);
}
// Pre-filter on a triangle's BOUNDING SPHERE (centre c, radius r, both rounded outwards by whoever built them): FALSE only if no point of the
// sphere — hence of the triangle — lies inside the cone within `range`, i.e. only if cone_tri_maybe / intersect_cone_tri would say no too.
// Every point of the sphere has z in [zc - r, zc + r] and a Euclidean distance from the axis of at least l - r (l: that of the centre); the
// cone's cross-section at z, an ellipse of semi-axes (R(z), R(z) / e) with e >= 1, lies inside the circle of radius R(z) <= R(zhi), zhi =
// min(zc + r, range.max).  The margin covers the rounding of l (components of |v| eps each) many times over.  16 B and ~20 operations per
// triangle against 36 B and ~150 for cone_tri_maybe: the wave-cooperative queries run it first (wt/coop.h).
WT_HD bool cone_sphere_maybe(const cone_t& cone, vec3 c, float r, const range_t& range) {
    if (cone_is_ray(cone)) return true;
    const vec3 v = c - cone.o;
    const float zc = dot(v, cone.d);
    if (zc + r < range.min || zc - r > range.max) return false;
    const float zhi = fminf_(zc + r, range.max);
    const float r_hi = zhi * cone.tan_alpha + cone.x0;
    if (!(r_hi >= 0.f) || !finitef(r_hi)) return true;
    const vec3 w = v - zc * cone.d;
    const float l2 = dot(w, w);
    const float lim = r_hi * 1.002f + r + 4e-6f * (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
    return !(l2 > lim * lim);
}
// The two conservative rejections of intersect_cone_tri on their own: FALSE only if intersect_cone_tri(cone,a,b,c,..,range) would
// return false as well (used by the wave-cooperative traversal to filter candidates before the exact test).
WT_HD bool cone_tri_maybe(const cone_t& cone, vec3 a, vec3 b, vec3 c, const range_t& range) {
    if (cone_is_ray(cone)) return true;
    const frame_t frame = cone_frame(cone);
    const vec3 o = cone.o;
    const vec3 vs[3] = {to_local(frame, a - o), to_local(frame, b - o), to_local(frame, c - o)};
    const float closest_z = fminf_(vs[0].z, fminf_(vs[1].z, vs[2].z));
    const float farthest_z = fmaxf_(vs[0].z, fmaxf_(vs[1].z, vs[2].z));
    if (farthest_z < range.min || closest_z > range.max) return false;
    return !cone_tri_laterally_outside(cone, vs, fminf_(farthest_z, range.max));
}
// WT_SECOND_SOURCE (CPU only; oracle/indep/prims2.cpp, libindep2.so): independent derivations in double precision take the place of a handful of
// primitives, so that whole renders can be compared between the two sources (tests/test_second_source.py).
#if defined(WT_SECOND_SOURCE) && !defined(__HIP_DEVICE_COMPILE__)
#define WT_SS_ACTIVE 1
extern "C" int ss_intersect_cone_tri(const float o[3], const float d[3], const float x[3], float x0, float tan_alpha, float e, const float a[3], const float b[3],
                                     const float c[3], float zmin, float zmax, float* dist);
#else
#define WT_SS_ACTIVE 0
#endif
#ifdef WT_PROFILE_CONE_TRI
inline unsigned long long g_cone_tri_exits[8] = {0};
#define WT_CT_EXIT(i) (g_cone_tri_exits[i]++)
#else
#define WT_CT_EXIT(i) ((void)0)
#endif
// any_hit = true: only the boolean matters (out.dist is some hit distance inside `range`, not the closest): a contained vertex
// decides immediately.
template <bool any_hit = false>
WT_HD bool intersect_cone_tri(const cone_t& cone, vec3 a, vec3 b, vec3 c, vec3 n, const range_t& range, cone_tri_hit_t& out) {
    if (cone_is_ray(cone)) {
        ray_tri_hit_t h;
        if (intersect_ray_tri(cone.o, cone.d, a, b, c, range, h)) {
            out.dist = h.dist;
            out.p = cone.o + h.dist * cone.d;
            return true;
        }
        return false;
    }
#if WT_SS_ACTIVE
    {
        float dist = 0.f;
        if (!ss_intersect_cone_tri(&cone.o.x, &cone.d.x, &cone.x.x, cone.x0, cone.tan_alpha, cone.e, &a.x, &b.x, &c.x, range.min, range.max, &dist)) return false;
        out.dist = dist;
        out.p = cone.o + dist * cone.d;
        return true;
    }
#endif
    const frame_t frame = cone_frame(cone);
    const vec3 o = cone.o;
    const vec3 vs[3] = {to_local(frame, a - o), to_local(frame, b - o), to_local(frame, c - o)};

    const float closest_z = fminf_(vs[0].z, fminf_(vs[1].z, vs[2].z));
    const float farthest_z = fmaxf_(vs[0].z, fmaxf_(vs[1].z, vs[2].z));
    if (farthest_z < range.min || closest_z > range.max) {
        WT_CT_EXIT(0);
        return false;
    }
    // Cheap conservative rejection (not in the reference; result-preserving): 97 % of the candidate triangles a BVH leaf hands
    // over miss the cone, 3/4 by the slab test above and most of the rest laterally, and the exact tests below cost ~20x more.
    if (cone_tri_laterally_outside(cone, vs, fminf_(farthest_z, range.max))) {
        WT_CT_EXIT(6);
        return false;
    }
    const vec3 ln = to_local(frame, n);

    bool cont[3];
    for (int i = 0; i < 3; ++i) cont[i] = cone_contains_local(cone, vs[i], range);
    if (any_hit) {
        for (int i = 0; i < 3; ++i)
            if (cont[i]) {
                out.dist = vs[i].z;
                out.p = to_world(frame, vs[i]) + o;
                WT_CT_EXIT(1);
                return true;
            }
    }

    for (int i = 0; i < 3; ++i) {
        if (cont[i] && vs[i].z == closest_z) {
            out.dist = closest_z;
            out.p = to_world(frame, vs[i]) + o;
            WT_CT_EXIT(1);
            return true;
        }
    }
    const cone_plane_t icp = intersect_cone_plane_local(cone, ln, dot(vs[0], ln), range);
    if (!empty(icp.range)) {
        if (is_point_in_triangle3(icp.near, vs[0], vs[1], vs[2])) {
            out.dist = icp.range.min;
            out.p = to_world(frame, icp.near) + o;
            WT_CT_EXIT(2);
            return true;
        }
    }
    bool has = false;
    vec3 p{0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3;
        const vec3 ea = vs[i], eb = vs[j];
        if (cont[i] && cont[j]) continue;
        if (ea.z > range.max && eb.z > range.max) continue;
        if (ea.z < range.min && eb.z < range.min) continue;
        cone_edge_t ce;
        if (intersect_cone_edge_local(cone, ea, eb, range, ce) && (!has || p.z > ce.p0.z)) {
            p = ce.p0;
            has = true;
        }
    }
    if (!has) {
        WT_CT_EXIT((cont[0] || cont[1] || cont[2]) ? 5 : 4);
        return false;
    }
    out.dist = p.z;
    out.p = to_world(frame, p) + o;
    WT_CT_EXIT(3);
    return true;
}

// ---- clip triangle to z-slab (clip.hpp:35-83) ---------------------------------------------------
struct clip_tri_t {
    vec3 vs[5];
    int tris;
};
WT_HD void clip_tri_get(const clip_tri_t& c, int idx, vec3& a, vec3& b, vec3& cc) {
    if (idx == 0) {
        a = c.vs[0];
        b = c.vs[1];
        cc = c.vs[2];
    } else if (idx == 1) {
        a = c.vs[2];
        b = c.vs[0];
        cc = c.vs[c.tris == 2 ? 3 : 4];
    } else {
        a = c.vs[4];
        b = c.vs[2];
        cc = c.vs[3];
    }
}
WT_HD clip_tri_t clip_triangle_z(vec3 a, vec3 b, vec3 c, const range_t& zr) {
    const vec3 tri[3] = {a, b, c};
    int cls[3];
    for (int i = 0; i < 3; ++i) cls[i] = tri[i].z > zr.max ? +1 : (tri[i].z < zr.min ? -1 : 0);
    clip_tri_t ret;
    int idx = 0;
    for (int i = 0; i < 3; ++i) {
        const int next = i == 2 ? 0 : i + 1;
        if (cls[i] == 0 && idx < 5) ret.vs[idx++] = tri[i];
        if (cls[next] != cls[i]) {
            const float zp = cls[i] == -1 ? zr.min : ((cls[i] == 1 || cls[next] == 1) ? zr.max : zr.min);
            vec3 pt;
            if (!intersect_edge_zplane(tri[i], tri[next], zp, pt)) pt = cls[i] != 0 ? tri[i] : tri[next];
            if (idx < 5) ret.vs[idx++] = pt;
            if (cls[next] != 0 && cls[i] != 0) {
                const float zp2 = cls[next] == 1 ? zr.max : zr.min;
                vec3 pt2;
                if (!intersect_edge_zplane(tri[i], tri[next], zp2, pt2)) pt2 = tri[next];
                if (idx < 5) ret.vs[idx++] = pt2;
            }
        }
    }
    ret.tris = idx < 3 ? 0 : (idx == 3 ? 1 : (idx == 4 ? 2 : 3));
    return ret;
}

// find_closest_triangle's footprint integral of ONE triangle given in the beam's local frame (plt_bdpt_detail.hpp:391-416): the part
// of the triangle inside the interaction region's z-slab, projected along the envelope onto the cross-section at the slab's centre.
// A triangle that lies wholly inside the slab — nearly all of a large region's — is what clip_triangle_z returns for it (one
// triangle, the same vertices): integrated directly, without the clipper's 5-vertex array (scratch traffic on the device).
WT_HD float region_local_triangle_flux(const cone_t& envelope, const range_t& izr, float csz, vec2 sigma, vec3 la, vec3 lb, vec3 lc) {
    const bool inside = la.z >= izr.min && la.z <= izr.max && lb.z >= izr.min && lb.z <= izr.max && lc.z >= izr.min && lc.z <= izr.max;
    if (inside)
        return wavefront_integrate_triangle(sigma, cone_project_local(envelope, la, csz), cone_project_local(envelope, lb, csz),
                                            cone_project_local(envelope, lc, csz));
    const clip_tri_t ct = clip_triangle_z(la, lb, lc, izr);
    float flux = 0.f;
    for (int t = 0; t < ct.tris; ++t) {
        vec3 a, b, c;
        clip_tri_get(ct, t, a, b, c);
        flux += wavefront_integrate_triangle(sigma, cone_project_local(envelope, a, csz), cone_project_local(envelope, b, csz),
                                             cone_project_local(envelope, c, csz));
    }
    return flux;
}

// cone_intersection_tolerance.hpp:23-41
WT_HD float cone_intersection_tolerance(vec3 origin, vec3 a, vec3 b, vec3 c) {
    const float c0 = 4e-7f, c1 = 1e-6f, c2 = 1e-6f;
    const vec3 mn = vmin(a, vmin(b, c)), mx = vmax(a, vmax(b, c));
    const float obj_extent = 2.f * fmaxf_(max_element(vabs(mn)), max_element(vabs(mx)));
    const vec3 ao = vabs(origin);
    const vec3 obj_err = (c0 + c2) * ao + vec3{c1 * obj_extent, c1 * obj_extent, c1 * obj_extent};
    const vec3 wrd_err = (c1 + c2) * ao;
    return max_element(obj_err + wrd_err);
}

// ---- cone re-sourcing through a footprint ellipse / ellipsoid (src/math/elliptic_cone.cpp) -------
WT_HD cone_t cone_through_ellipse(vec3 x, vec3 y, vec3 n, vec3 ro, vec3 rd, float tan_alpha, float* self_intersection_distance) {
    const bool xz = x.x == 0.f && x.y == 0.f && x.z == 0.f;
    const bool yz = y.x == 0.f && y.y == 0.f && y.z == 0.f;
    if (xz && yz) {
        if (self_intersection_distance) *self_intersection_distance = 0.f;
        return make_cone_raw(ro, rd, build_orthogonal_frame(rd).t, 0.f, tan_alpha, 1.f, 1.f);
    }
    const frame_t of = build_orthogonal_frame(rd);
    const vec3 xl = to_local(of, x), yl = to_local(of, y);
    const svd_t svd = svd2(mkmat2(vec2{xl.x, xl.y}, vec2{yl.x, yl.y}));
    vec2 X{svd.Ucos, -svd.Usin};
    float lX = fabsf(svd.sigma1), lY = fabsf(svd.sigma2);
    if (lX < lY) {
        const float t = lX;
        lX = lY;
        lY = t;
        X = vec2{svd.Usin, svd.Ucos};
    }
    // NB: the reference takes e = sqrt(lX/lY) here (elliptic_cone.cpp:66), kept verbatim.
    const float e = lY > 0.f ? sqrtf(lX / lY) : 1.f;
    const vec3 wx = to_world(of, X);
    const cone_t cone = make_cone_raw(ro, rd, wx, lX, tan_alpha, 1.f / e, e);
    if (self_intersection_distance) {
        const cone_plane_t cp = intersect_cone_plane_world(cone, n, dot(n, ro), range_t{0.f, WT_INF});
        *self_intersection_distance = empty(cp.range) ? 0.f : cp.range.max;
    }
    return cone;
}

WT_HD cone_t cone_through_ellipsoid(vec3 axes, const frame_t& axes_frame, vec3 ro, vec3 rd, float tan_alpha) {
    const vec3 wolocal = to_local(axes_frame, rd);
    const frame_t frame = build_orthogonal_frame(wolocal);
    const vec3 t = axes;
    const vec3 nn = normalize(t * wolocal);
    const frame_t fc = build_orthogonal_frame(nn);
    const vec3 t1 = t * fc.t;
    const vec3 t2 = t * fc.b;
    // frame.to_local(pqvec2_t) — only xy components of t,b are used (frame.hpp:22-27)
    const vec2 c0 = to_local2(frame, vec2{t1.x, t1.y});
    const vec2 c1 = to_local2(frame, vec2{t2.x, t2.y});
    const mat2 A = mkmat2(c0, c1);
    if (A.c0x * A.c1y == A.c1x * A.c0y) return make_cone_raw(ro, rd, build_orthogonal_frame(rd).t, 0.f, tan_alpha, 1.f, 1.f);
    const svd_t svd = svd2(A);
    vec2 X{svd.Ucos, -svd.Usin};
    float lX = fabsf(svd.sigma1), lY = fabsf(svd.sigma2);
    if (lX < lY) {
        const float tt = lX;
        lX = lY;
        lY = tt;
        X = vec2{svd.Usin, svd.Ucos};
    }
    const float e = lY > 0.f ? sqrtf(lX / lY) : 1.f;
    const vec3 X3 = normalize(to_world(frame, X));
    return make_cone_raw(ro, rd, to_world(axes_frame, X3), lX, tan_alpha, 1.f / e, e);
}

}   // namespace wt
