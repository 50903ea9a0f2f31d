// wave_tracer_amd — Wavefront OBJ reader: what src/mesh/obj_loader.cpp:26-140 builds from a file through
// tinyobjloader — every face corner becomes its own vertex (position, normal unless face normals are requested, uv), triangles (i,
// i+1, i+2); faces with more than three corners are fan-triangulated like tinyobjloader's default `triangulate` does for convex
// polygons; a file that gives normals or uvs for some corners and not for others is rejected like the reference does.
// Material groups: the shape's `mtl` attribute keeps the faces of one material (obj_loader.cpp:66-73).  As tinyobjloader does, `mtllib` files are
// looked up beside the OBJ file and read for their `newmtl` names (the rest of the line), `usemtl <word>` selects one of them for the faces that
// follow, and a name no library defines — or no `usemtl` at all — is "no material" (-1).  The reference's filter, restated with its quirk: a face
// is dropped when it has no material and `mtl` is EMPTY, or when it has a material of another name; faces without a material pass any non-empty
// `mtl`.  Nothing else of a material library is read (the reference takes its BSDFs from the scene file).  Tested with generated files
// (tests/test_xml_scene.py).
#include <cmath>
#include <fstream>
#include <map>
#include <cstdio>
#include <sstream>
#include <stdexcept>

#include "scene_builder.h"

namespace wth {

mesh_t load_obj(const std::string& path, bool face_normals, double scale, const std::string* mtl) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("(obj loader) cannot open " + path);
    std::vector<dvec3> pos, nrm;
    std::vector<std::array<float, 2>> tex;
    mesh_t m;
    bool first = true, has_n = false, has_uv = false;
    std::map<std::string, int> materials;   // name -> id, from the mtllib files met so far
    int cur_mtl = -1;
    std::vector<std::string> mtl_names;
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
    std::string line;
    size_t lineno = 0;
    auto fail = [&](const std::string& w) { throw std::runtime_error("(obj loader) " + path + ":" + std::to_string(lineno) + ": " + w); };
    while (std::getline(f, line)) {
        ++lineno;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string kw;
        if (!(ls >> kw) || kw[0] == '#') continue;
        if (kw == "v") {
            dvec3 p;
            if (!(ls >> p.x >> p.y >> p.z)) fail("v: three coordinates expected");
            pos.push_back(p);
        } else if (kw == "vn") {
            dvec3 n;
            if (!(ls >> n.x >> n.y >> n.z)) fail("vn: three components expected");
            nrm.push_back(n);
        } else if (kw == "vt") {
            float u = 0, v = 0;
            if (!(ls >> u)) fail("vt: coordinates expected");
            ls >> v;
            tex.push_back({u, v});
        } else if (kw == "mtllib") {
            std::string lib;
            while (ls >> lib) {
                std::ifstream mf(dir + "/" + lib);
                if (!mf) {
                    std::fprintf(stderr, "(obj loader) %s: material library %s not found\n", path.c_str(), lib.c_str());   // a warning in tinyobjloader too
                    continue;
                }
                std::string ml;
                while (std::getline(mf, ml)) {
                    if (!ml.empty() && ml.back() == '\r') ml.pop_back();
                    const size_t b = ml.find_first_not_of(" \t");
                    if (b == std::string::npos || ml.compare(b, 7, "newmtl ") != 0) continue;
                    const size_t nb = ml.find_first_not_of(" \t", b + 7);
                    const std::string name = nb == std::string::npos ? std::string() : ml.substr(nb);
                    if (!materials.count(name)) {
                        materials[name] = (int)mtl_names.size();
                        mtl_names.push_back(name);
                    }
                }
            }
        } else if (kw == "usemtl") {
            std::string name;
            ls >> name;
            const auto it = materials.find(name);
            cur_mtl = it == materials.end() ? -1 : it->second;
        } else if (kw == "f") {
            if (mtl && ((cur_mtl == -1 && mtl->empty()) || (cur_mtl >= 0 && mtl_names[cur_mtl] != *mtl))) continue;   // obj_loader.cpp:68-73
            struct corner_t {
                long v, t, n;
            };
            std::vector<corner_t> cs;
            std::string tok;
            while (ls >> tok) {
                corner_t c{0, 0, 0};
                const size_t s1 = tok.find('/');
                c.v = std::stol(tok.substr(0, s1));
                if (s1 != std::string::npos) {
                    const size_t s2 = tok.find('/', s1 + 1);
                    const std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) c.t = std::stol(t);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = std::stol(tok.substr(s2 + 1));
                }
                // 1-based; negative = relative to the end
                auto fix = [&](long i, size_t count) -> long { return i > 0 ? i - 1 : (i < 0 ? (long)count + i : -1); };
                c.v = fix(c.v, pos.size());
                c.t = fix(c.t, tex.size());
                c.n = fix(c.n, nrm.size());
                if (c.v < 0 || c.v >= (long)pos.size()) fail("f: vertex index out of range");
                if (c.t >= (long)tex.size() || c.n >= (long)nrm.size()) fail("f: index out of range");
                cs.push_back(c);
            }
            if (cs.size() < 3) fail("f: at least three corners expected");
            for (size_t k = 1; k + 1 < cs.size(); ++k) {
                const corner_t tri[3] = {cs[0], cs[k], cs[k + 1]};
                for (const corner_t& c : tri) {
                    if (first) {
                        has_n = !face_normals && c.n >= 0;
                        has_uv = c.t >= 0;
                        first = false;
                    }
                    if (!face_normals && has_n != (c.n >= 0)) fail("OBJ file is missing normal data from some vertices. This is unsupported.");
                    if (has_uv != (c.t >= 0)) fail("OBJ file is missing uv data from some vertices. This is unsupported.");
                    const dvec3 p = pos[c.v];
                    m.verts.push_back({p.x * scale, p.y * scale, p.z * scale});
                    if (has_n) {
                        const dvec3 n = nrm[c.n];
                        const double l = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
                        m.normals.push_back(l > 0 ? dvec3{n.x / l, n.y / l, n.z / l} : dvec3{0, 0, 1});
                    }
                    if (has_uv) m.uvs.push_back(tex[c.t]);
                }
                const uint32_t i = (uint32_t)m.verts.size() - 3;
                m.tris.push_back({i, i + 1, i + 2});
            }
        }
        // o, g, s, ...: ignored
    }
    if (m.tris.empty() && mtl) throw std::runtime_error("(obj loader) " + path + ": No faces found for supplied 'mtl' (\"" + *mtl + "\")");   // (a warning and an empty mesh there)
    if (m.tris.empty()) throw std::runtime_error("(obj loader) " + path + ": no faces");
    return m;
}

}   // namespace wth
