// wave_tracer_amd — reader of the reference's XML scene format ("next" row), baking through scene_builder_t.
// Loads AS SHIPPED: scenes/diffraction_simple/{double_slits,double_slits_and_reflectors}.xml (+ bits/geometry.xml), scenes/cornell-box/
// {box,sphere_polarization}.xml (Git-LFS pointer files in place of meshes / bitmaps: the bundled stand-ins, host/scenes.cpp:
// asset_standin_mesh); with -Dwtgpu_missing_assets=skip (absent meshes left out) also room, bike, kitchen, etoile, munich, sponza_day
// and veach_mis.  Not a general loader: what is unsupported fails with a message that names it.
//
// Reference behaviour restated (nothing is copied; the reference parses with pugixml, which is not available here):
//   * <default name value> defines with command-line overrides ("-Dname=value"), textual "$name" substitution in attribute values,
//     "\$" for a literal dollar (src/scene/loader/xml/loader.cpp, src/main.cpp:805-928);
//   * attribute values are arithmetic / boolean expressions (+ - * / comparisons && || !, true / false / pi, sin cos tan asin acos atan
//     atan2 sqrt abs exp log pow min max round floor ceil), optionally followed by a unit ("($S-.0001) mm", ".001°", "5750K", "10GHz"
//     where a wavelength is expected), comma-separated for vectors, "a .. b" for ranges, "(re,imi)" for complex constants;
//   * <include path> is relative to the including file and is spliced in place;
//   * elements with <boolean name="enabled" value=…/> evaluating to false are skipped (integrators, sensors, emitters, shapes);
//   * <ref id=…/> stands for a named BSDF (inside shapes and wrappers) or a shared top-level texture / spectrum;
//   * node vocabulary: integrator (plt_bdpt | plt_path + direction); sensor (virtual_plane | perspective with fov_axis; polarimetric
//     attribute) with film (array, rfilter_scale; response monochromatic / discrete line or RGB with white point);
//     emitter (spot, point, directional with lookat or a general transform, area on a shape) in the reference loader's order;
//     bsdf (twosided, scale with a constant / spectrum / texture, mask, normalmap, composite bins, diffuse, dielectric, surface_spm with
//     dirac / fractal / gaussian profile, extIOR, reflection_scale, transmission_scale);
//     spectrum (constant, complex constant, discrete line, rgb, blackbody, piecewise_linear, gaussian, composite bins, named IOR /
//     emission database entries, ITU-R P.2040 materials);
//     texture (constant, checkerboard, scale, transform, bitmap from PNG / PFM with filter, wrap modes and colour encoding);
//     shape (rectangle, cube, sphere, cylinder, prism, lens, ply / obj: host/ply_loader.cpp, host/obj_loader.cpp) with general to_world
//     transforms (matrix / rotate / scale / translate / lookat).
//   Not handled: spatially varying emitter radiance textures (constant ones are), sensor masks.
// Spectral resolution at bake time (as in host/scenes.cpp): composite BSDFs / spectra take the bin that contains the sensor's
// sensitivity range; an emitter whose spectrum has no line-for-line overlap with the sensor's (a continuous spectrum against a
// monochromatic sensor: the reference integrates it over a 2e-6 relative band, scene_build_sensor_sampling_data.cpp:55-60, ~1e-9
// of a line emitter's power; a far-infrared line against an RGB sensor: zero) is not added.
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "scene_builder.h"

namespace wth {
using namespace wt;

namespace {

// ---------------------------------------------------------------------------------------------- tiny XML DOM
struct xnode_t {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<xnode_t> kids;
    const std::string* attr(const char* n) const {
        for (auto& a : attrs)
            if (a.first == n) return &a.second;
        return nullptr;
    }
    std::string get(const char* n, const std::string& def = "") const {
        const std::string* a = attr(n);
        return a ? *a : def;
    }
    // <type name="n" value=…> child
    const xnode_t* named(const char* n) const {
        for (auto& k : kids)
            if (k.get("name") == n) return &k;
        return nullptr;
    }
    const xnode_t* child(const char* element) const {
        for (auto& k : kids)
            if (k.name == element) return &k;
        return nullptr;
    }
};

struct xml_parser_t {
    const std::string& s;
    size_t i = 0;
    std::string file;
    explicit xml_parser_t(const std::string& text, std::string f) : s(text), file(std::move(f)) {}
    [[noreturn]] void fail(const std::string& what) const {
        size_t line = 1;
        for (size_t k = 0; k < i && k < s.size(); ++k) line += s[k] == '\n';
        throw std::runtime_error(file + ":" + std::to_string(line) + ": " + what);
    }
    void skip_ws() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }
    bool starts(const char* t) const { return s.compare(i, std::strlen(t), t) == 0; }
    void skip_misc() {   // whitespace, text, comments, processing instructions, doctype
        for (;;) {
            while (i < s.size() && s[i] != '<') ++i;
            if (i >= s.size()) return;
            if (starts("<!--")) {
                const size_t e = s.find("-->", i + 4);
                if (e == std::string::npos) fail("unterminated comment");
                i = e + 3;
            } else if (starts("<?")) {
                const size_t e = s.find("?>", i + 2);
                if (e == std::string::npos) fail("unterminated processing instruction");
                i = e + 2;
            } else if (starts("<!")) {
                const size_t e = s.find('>', i);
                if (e == std::string::npos) fail("unterminated declaration");
                i = e + 1;
            } else
                return;
        }
    }
    static std::string unescape(const std::string& v) {
        std::string o;
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] == '&') {
                static const std::pair<const char*, char> ent[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
                bool done = false;
                for (auto& e : ent)
                    if (v.compare(k, std::strlen(e.first), e.first) == 0) {
                        o += e.second;
                        k += std::strlen(e.first) - 1;
                        done = true;
                        break;
                    }
                if (done) continue;   // a bare '&' (the shipped scenes write "a && b" unescaped) is kept
            }
            o += v[k];
        }
        return o;
    }
    std::string ident() {
        const size_t b = i;
        while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
        if (i == b) fail("name expected");
        return s.substr(b, i - b);
    }
    // parses one element at s[i] == '<'
    int depth = 0;   // nesting of the element being parsed (a stack overflow could not be caught like the exceptions the C-ABI translates)
    struct depth_guard_t {
        int& d;
        explicit depth_guard_t(int& d_) : d(d_) { ++d; }
        ~depth_guard_t() { --d; }
    };
    xnode_t element() {
        const depth_guard_t guard(depth);
        if (depth > 256) fail("elements nested deeper than 256 levels");
        xnode_t n;
        ++i;
        n.name = ident();
        for (;;) {
            skip_ws();
            if (i >= s.size()) fail("unterminated element <" + n.name + ">");
            if (starts("/>")) {
                i += 2;
                return n;
            }
            if (s[i] == '>') {
                ++i;
                break;
            }
            const std::string an = ident();
            skip_ws();
            if (i >= s.size() || s[i] != '=') fail("'=' expected after attribute " + an);
            ++i;
            skip_ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("quoted value expected for attribute " + an);
            const char q = s[i++];
            const size_t e = s.find(q, i);
            if (e == std::string::npos) fail("unterminated attribute value");
            n.attrs.emplace_back(an, unescape(s.substr(i, e - i)));
            i = e + 1;
        }
        for (;;) {
            skip_misc();
            if (i >= s.size()) fail("missing </" + n.name + ">");
            if (starts("</")) {
                i += 2;
                const std::string cn = ident();
                if (cn != n.name) fail("</" + cn + "> closes <" + n.name + ">");
                skip_ws();
                if (i >= s.size() || s[i] != '>') fail("'>' expected");
                ++i;
                return n;
            }
            n.kids.push_back(element());
        }
    }
    // a document or a fragment: every top-level element
    std::vector<xnode_t> top_level() {
        std::vector<xnode_t> v;
        for (;;) {
            skip_misc();
            if (i >= s.size()) return v;
            v.push_back(element());
        }
    }
};

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
std::string dir_of(const std::string& path) {
    const size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

// ---------------------------------------------------------------------------------------------- expressions and quantities
struct expr_t {
    const std::string& s;
    size_t i = 0;
    explicit expr_t(const std::string& t) : s(t) {}
    [[noreturn]] void fail(const std::string& w) const { throw std::runtime_error("expression \"" + s + "\": " + w); }
    void ws() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }
    bool eat(const char* t) {
        ws();
        const size_t n = std::strlen(t);
        if (s.compare(i, n, t) == 0) {
            i += n;
            return true;
        }
        return false;
    }
    int paren_depth = 0;
    double primary() {
        ws();
        if (i >= s.size()) fail("operand expected");
        if (s[i] == '(') {
            ++i;
            if (++paren_depth > 256) fail("parentheses nested deeper than 256 levels");
            const double v = lor();
            --paren_depth;
            if (!eat(")")) fail("')' expected");
            return v;
        }
        if (std::isalpha((unsigned char)s[i])) {
            const size_t b = i;
            while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
            const std::string w = s.substr(b, i - b);
            if (w == "true") return 1.0;
            if (w == "false") return 0.0;
            if (w == "pi") return M_PI;
            static const std::pair<const char*, double (*)(double)> f1[] = {
                {"sin", [](double a) { return std::sin(a); }},     {"cos", [](double a) { return std::cos(a); }},   {"tan", [](double a) { return std::tan(a); }},
                {"asin", [](double a) { return std::asin(a); }},   {"acos", [](double a) { return std::acos(a); }}, {"atan", [](double a) { return std::atan(a); }},
                {"sqrt", [](double a) { return std::sqrt(a); }},   {"abs", [](double a) { return std::fabs(a); }},  {"exp", [](double a) { return std::exp(a); }},
                {"log", [](double a) { return std::log(a); }},     {"round", [](double a) { return std::round(a); }},
                {"floor", [](double a) { return std::floor(a); }}, {"ceil", [](double a) { return std::ceil(a); }}};
            static const std::pair<const char*, double (*)(double, double)> f2[] = {{"min", [](double a, double c) { return std::min(a, c); }},
                                                                                     {"max", [](double a, double c) { return std::max(a, c); }},
                                                                                     {"pow", [](double a, double c) { return std::pow(a, c); }},
                                                                                     {"atan2", [](double a, double c) { return std::atan2(a, c); }}};
            for (auto& f : f1)
                if (w == f.first) {
                    if (!eat("(")) fail("'(' expected after " + w);
                    const double a = lor();
                    if (!eat(")")) fail("')' expected");
                    return f.second(a);
                }
            for (auto& f : f2)
                if (w == f.first) {
                    if (!eat("(")) fail("'(' expected after " + w);
                    const double a = lor();
                    if (!eat(",")) fail("',' expected in " + w + "(a, b)");
                    const double c = lor();
                    if (!eat(")")) fail("')' expected");
                    return f.second(a, c);
                }
            fail("unknown identifier " + w);
        }
        const char* b = s.c_str() + i;
        char* e = nullptr;
        const double v = std::strtod(b, &e);
        if (e == b) fail("number expected");
        i += (size_t)(e - b);
        return v;
    }
    double unary() {
        ws();
        if (eat("-")) return -unary();
        if (eat("+")) return unary();
        if (i < s.size() && s[i] == '!' && !(i + 1 < s.size() && s[i + 1] == '=')) {
            ++i;
            return unary() == 0.0 ? 1.0 : 0.0;
        }
        return primary();
    }
    double mul() {
        double v = unary();
        for (;;) {
            if (eat("*"))
                v *= unary();
            else if (eat("/"))
                v /= unary();
            else
                return v;
        }
    }
    double add() {
        double v = mul();
        for (;;) {
            ws();
            if (eat("+"))
                v += mul();
            else if (i < s.size() && s[i] == '-') {
                ++i;
                v -= mul();
            } else
                return v;
        }
    }
    double cmp() {
        const double a = add();
        if (eat("==")) return a == add() ? 1.0 : 0.0;
        if (eat("!=")) return a != add() ? 1.0 : 0.0;
        if (eat("<=")) return a <= add() ? 1.0 : 0.0;
        if (eat(">=")) return a >= add() ? 1.0 : 0.0;
        if (eat("<")) return a < add() ? 1.0 : 0.0;
        if (eat(">")) return a > add() ? 1.0 : 0.0;
        return a;
    }
    double land() {
        double v = cmp();
        while (eat("&&")) {
            const double r = cmp();
            v = (v != 0.0 && r != 0.0) ? 1.0 : 0.0;
        }
        return v;
    }
    double lor() {
        double v = land();
        while (eat("||")) {
            const double r = land();
            v = (v != 0.0 || r != 0.0) ? 1.0 : 0.0;
        }
        return v;
    }
    // evaluates a prefix of the string; `i` is left behind it
    double prefix() { return lor(); }
};

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}
double eval_number(const std::string& s) {
    expr_t e(s);
    const double v = e.prefix();
    e.ws();
    if (e.i != s.size()) e.fail("trailing characters");
    return v;
}

// The same grammar, COMPILED instead of evaluated: the postfix program of a function texture (wt/scene.h: texture_function) over named variables
// (src/texture/function.cpp:38-107: the nested textures by name, then u, v, k).  `vars`: name -> (opcode, argument).
struct expr_compiler_t {
    const std::string& s;
    const std::vector<std::pair<std::string, std::pair<int, float>>>& vars;
    std::vector<float>& out;
    size_t i = 0;
    int depth = 0;
    expr_compiler_t(const std::string& t, const std::vector<std::pair<std::string, std::pair<int, float>>>& v, std::vector<float>& o) : s(t), vars(v), out(o) {}
    [[noreturn]] void fail(const std::string& w) const { throw std::runtime_error("function \"" + s + "\": " + w); }
    void emit(int op, float arg = 0.f) {
        out.push_back((float)op);
        out.push_back(arg);
    }
    void ws() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    }
    bool eat(const char* t) {
        ws();
        const size_t n = std::strlen(t);
        if (s.compare(i, n, t) == 0) {
            i += n;
            return true;
        }
        return false;
    }
    void primary() {
        ws();
        if (i >= s.size()) fail("operand expected");
        if (s[i] == '(') {
            ++i;
            if (++depth > 64) fail("parentheses nested too deeply");
            lor();
            --depth;
            if (!eat(")")) fail("')' expected");
            return;
        }
        if (std::isalpha((unsigned char)s[i]) || s[i] == '_') {
            const size_t b = i;
            while (i < s.size() && (std::isalnum((unsigned char)s[i]) || s[i] == '_')) ++i;
            const std::string w = s.substr(b, i - b);
            static const std::pair<const char*, int> f1[] = {{"sin", TOP_SIN}, {"cos", TOP_COS}, {"tan", TOP_TAN}, {"asin", TOP_ASIN}, {"acos", TOP_ACOS}, {"atan", TOP_ATAN},
                                                             {"sqrt", TOP_SQRT}, {"abs", TOP_ABS}, {"exp", TOP_EXP}, {"log", TOP_LOG}, {"round", TOP_ROUND}, {"floor", TOP_FLOOR},
                                                             {"ceil", TOP_CEIL}};
            static const std::pair<const char*, int> f2[] = {{"min", TOP_MIN}, {"max", TOP_MAX}, {"pow", TOP_POW}, {"atan2", TOP_ATAN2}};
            ws();
            const bool call = i < s.size() && s[i] == '(';
            if (call) {
                for (auto& f : f1)
                    if (w == f.first) {
                        eat("(");
                        lor();
                        if (!eat(")")) fail("')' expected");
                        emit(f.second);
                        return;
                    }
                for (auto& f : f2)
                    if (w == f.first) {
                        eat("(");
                        lor();
                        if (!eat(",")) fail("',' expected in " + w + "(a, b)");
                        lor();
                        if (!eat(")")) fail("')' expected");
                        emit(f.second);
                        return;
                    }
                if (w == "mix") {   // mix(a, b, t)
                    eat("(");
                    lor();
                    if (!eat(",")) fail("',' expected in mix(a, b, t)");
                    lor();
                    if (!eat(",")) fail("',' expected in mix(a, b, t)");
                    lor();
                    if (!eat(")")) fail("')' expected");
                    emit(TOP_MIX);
                    return;
                }
                fail("unknown function " + w);
            }
            for (auto& v : vars)
                if (v.first == w) {
                    emit(v.second.first, v.second.second);
                    return;
                }
            if (w == "pi") return emit(TOP_CONST, (float)M_PI);
            if (w == "true") return emit(TOP_CONST, 1.f);
            if (w == "false") return emit(TOP_CONST, 0.f);
            fail("unknown variable " + w);
        }
        const char* b = s.c_str() + i;
        char* e = nullptr;
        const double v = std::strtod(b, &e);
        if (e == b) fail("number expected");
        i += (size_t)(e - b);
        emit(TOP_CONST, (float)v);
    }
    void unary() {
        ws();
        if (eat("-")) {
            unary();
            return emit(TOP_NEG);
        }
        if (eat("+")) return unary();
        if (i < s.size() && s[i] == '!' && !(i + 1 < s.size() && s[i + 1] == '=')) {
            ++i;
            unary();
            return emit(TOP_NOT);
        }
        primary();
    }
    void mul() {
        unary();
        for (;;) {
            if (eat("*")) {
                unary();
                emit(TOP_MUL);
            } else if (eat("/")) {
                unary();
                emit(TOP_DIV);
            } else
                return;
        }
    }
    void add() {
        mul();
        for (;;) {
            ws();
            if (eat("+")) {
                mul();
                emit(TOP_ADD);
            } else if (i < s.size() && s[i] == '-') {
                ++i;
                mul();
                emit(TOP_SUB);
            } else
                return;
        }
    }
    void cmp() {
        add();
        static const std::pair<const char*, int> ops[] = {{"==", TOP_EQ}, {"!=", TOP_NE}, {"<=", TOP_LE}, {">=", TOP_GE}, {"<", TOP_LT}, {">", TOP_GT}};
        for (auto& o : ops)
            if (eat(o.first)) {
                add();
                return emit(o.second);
            }
    }
    void land() {
        cmp();
        while (eat("&&")) {
            cmp();
            emit(TOP_AND);
        }
    }
    void lor() {
        land();
        while (eat("||")) {
            land();
            emit(TOP_OR);
        }
    }
    void compile() {
        lor();
        ws();
        if (i != s.size()) fail("trailing characters");
    }
};

enum dim_e { DIM_NONE, DIM_LENGTH, DIM_ANGLE, DIM_TEMPERATURE };
struct quantity_t {
    double raw;      // the number as written
    double factor;   // unit -> SI (metres, radians, kelvin)
    bool degrees;
    dim_e dim;
    // (degrees: v * pi / 180 in this order, lengths: v * 1e-3 etc. — the operation order of host/scenes.cpp, so that a scene read from
    // its XML is bit-identical to the hand-written builder)
    double si() const { return degrees ? raw * M_PI / 180.0 : raw * factor; }
    double in_mm() const { return factor == 1e-3 ? raw : raw * factor * 1e3; }
};
quantity_t parse_quantity(const std::string& text) {
    const std::string s = trim(text);
    expr_t e(s);
    const double v = e.prefix();
    const std::string unit = trim(s.substr(e.i));
    static const struct {
        const char* u;
        double f;
        dim_e d;
    } units[] = {{"", 1, DIM_NONE},        {"m", 1, DIM_LENGTH},          {"mm", 1e-3, DIM_LENGTH},       {"cm", 1e-2, DIM_LENGTH},
                 {"um", 1e-6, DIM_LENGTH}, {"\xC2\xB5m", 1e-6, DIM_LENGTH}, {"nm", 1e-9, DIM_LENGTH},       {"km", 1e3, DIM_LENGTH},
                 {"rad", 1, DIM_ANGLE},    {"K", 1, DIM_TEMPERATURE}};
    if (unit == "\xC2\xB0" || unit == "deg") return {v, M_PI / 180.0, true, DIM_ANGLE};
    for (auto& u : units)
        if (unit == u.u) return {v, u.f, false, u.d};
    throw std::runtime_error("quantity \"" + text + "\": unknown unit \"" + unit + "\"");
}
quantity_t parse_q(const std::string& text, dim_e want, const char* what) {
    const quantity_t q = parse_quantity(text);
    if (q.dim != want) throw std::runtime_error(std::string(what) + " \"" + text + "\": wrong or missing unit");
    return q;
}
double parse_dim(const std::string& text, dim_e want, const char* what) { return parse_q(text, want, what).si(); }
// a wavelength, or a frequency that stands for one (parse_quantity.hpp:308-315: "…Hz" -> freq_to_wavelen)
quantity_t parse_wavelength(const std::string& text, const char* what) {
    if (text.find("Hz") == std::string::npos) return parse_q(text, DIM_LENGTH, what);
    const std::string s = trim(text);
    expr_t e(s);
    const double v = e.prefix();
    const std::string unit = trim(s.substr(e.i));
    static const std::pair<const char*, double> units[] = {{"Hz", 1.0}, {"kHz", 1e3}, {"MHz", 1e6}, {"GHz", 1e9}, {"THz", 1e12}};
    for (auto& u : units)
        if (unit == u.first) {
            if (!(v > 0)) throw std::runtime_error(std::string(what) + " \"" + text + "\": a positive frequency expected");
            return {299792458.0 / (v * u.second) * 1e3, 1e-3, false, DIM_LENGTH};   // millimetres (the operation order of host/scenes.cpp)
        }
    throw std::runtime_error(std::string(what) + " \"" + text + "\": unknown frequency unit \"" + unit + "\"");
}
// splits at top-level commas
std::vector<std::string> split_list(const std::string& s) {
    std::vector<std::string> out;
    int depth = 0;
    std::string cur;
    for (char c : s) {
        if (c == '(') ++depth;
        if (c == ')') --depth;
        if (c == ',' && depth == 0) {
            out.push_back(cur);
            cur.clear();
        } else
            cur += c;
    }
    out.push_back(cur);
    return out;
}
dvec3 parse_point(const std::string& s, dim_e dim, const char* what) {
    const auto p = split_list(s);
    if (p.size() != 3) throw std::runtime_error(std::string(what) + " \"" + s + "\": three components expected");
    return {parse_dim(p[0], dim, what), parse_dim(p[1], dim, what), parse_dim(p[2], dim, what)};
}
// "a .. b" (lengths) -> metres
bool parse_length_range(const std::string& s, double& lo, double& hi) {
    const size_t p = s.find("..");
    if (p == std::string::npos) return false;
    lo = parse_dim(s.substr(0, p), DIM_LENGTH, "range");
    hi = parse_dim(s.substr(p + 2), DIM_LENGTH, "range");
    return true;
}
// "(re,imi)" | "re"
void parse_complex(const std::string& text, double& re, double& im) {
    const std::string s = trim(text);
    re = im = 0;
    if (s.size() > 2 && s.front() == '(' && s.back() == ')') {
        const auto p = split_list(s.substr(1, s.size() - 2));
        if (p.size() == 2) {
            std::string ims = trim(p[1]);
            if (!ims.empty() && (ims.back() == 'i' || ims.back() == 'j')) {
                ims.pop_back();
                re = eval_number(p[0]);
                im = eval_number(ims);
                return;
            }
        }
    }
    re = eval_number(s);
}

// ---------------------------------------------------------------------------------------------- the loader
// an asset that the checkout does not hold: a Git-LFS pointer (a short text file) in place of the binary
static bool is_lfs_pointer(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.good()) return false;
    char head[24] = {0};
    f.read(head, 23);
    return std::string(head).rfind("version https://git-lfs", 0) == 0;
}

struct loader_t {
    std::map<std::string, std::string> defs;
    scene_builder_t& b;
    std::vector<xnode_t> items;   // the <scene>'s children with includes spliced in, substituted
    // sensor sensitivity: a line (mono) or the visible band
    bool mono = false;
    double line_m = 0, line_mm = 0;    // monochromatic sensor: wavelength [m], [mm]
    double band_lo = 0, band_hi = 0;   // RGB sensor: sensitivity band [m]
    std::map<std::string, int> materials;
    std::string base_dir;   // directory of the scene file: relative asset paths resolve against it
    int mesh_detail = 1;    // tessellation of the procedural stand-ins for Git-LFS assets (scene_params_t::mesh_detail)

    explicit loader_t(scene_builder_t& bb) : b(bb) {}

    std::string subst(const std::string& v) const {
        std::string o;
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] == '\\' && i + 1 < v.size() && v[i + 1] == '$') {   // "\$": a literal dollar (regular expressions in sensor masks)
                o += '$';
                ++i;
            } else if (v[i] == '$') {
                size_t e = i + 1;
                while (e < v.size() && (std::isalnum((unsigned char)v[e]) || v[e] == '_')) ++e;
                const std::string n = v.substr(i + 1, e - i - 1);
                const auto it = defs.find(n);
                if (it == defs.end()) throw std::runtime_error("undefined $" + n);
                o += it->second;
                i = e - 1;
            } else
                o += v[i];
        }
        return o;
    }
    void subst_tree(xnode_t& n) const {
        for (auto& a : n.attrs) a.second = subst(a.second);
        for (auto& k : n.kids) subst_tree(k);
    }
    void splice(std::vector<xnode_t>&& nodes, const std::string& dir, int depth = 0) {
        if (depth > 8) throw std::runtime_error("<include> nested deeper than 8 levels (a cycle?)");
        for (auto& n : nodes) {
            if (n.name == "default") {
                const std::string name = n.get("name");
                if (!defs.count(name)) defs[name] = subst(n.get("value"));   // command-line defines win
            } else if (n.name == "include") {
                const std::string path = dir + "/" + subst(n.get("path"));
                const std::string text = read_file(path);
                xml_parser_t p(text, path);
                splice(p.top_level(), dir_of(path), depth + 1);
            } else {
                subst_tree(n);
                items.push_back(std::move(n));
            }
        }
    }
    static bool enabled(const xnode_t& n) {
        const xnode_t* e = n.named("enabled");
        return !e || eval_number(e->get("value", "true")) != 0.0;
    }
    // x= y= z= attributes or value="a, b, c" (value="s": all three); missing components take `def`
    static dvec3 read_vec3(const xnode_t& n, dim_e dim, double def, const char* what) {
        if (n.attr("value")) {
            const auto p = split_list(n.get("value"));
            if (p.size() == 1) {
                const double v = parse_dim(p[0], dim, what);
                return {v, v, v};
            }
            return parse_point(n.get("value"), dim, what);
        }
        auto c = [&](const char* a) { return n.attr(a) ? parse_dim(n.get(a), dim, what) : def; };
        return {c("x"), c("y"), c("z")};
    }
    // src/math/transform_loader.cpp:60-143: a <lookat> (exclusive), or matrix / rotate / translate / scale applied in file order (each
    // multiplies from the left)
    xform_t to_world(const xnode_t& n, dvec3 default_up) const {
        const xnode_t* t = n.named("to_world");
        if (!t) return xform_t::identity();
        if (t->name == "ref") {   // <ref name="to_world" id=…/>: a shared top-level <transform id=…> (loader.cpp:168-181; scenes/objects/objects.xml)
            const std::string id = t->get("id");
            t = nullptr;
            for (auto& it : items)
                if (it.name == "transform" && it.get("id") == id) t = &it;
            if (!t) throw std::runtime_error("<ref name=\"to_world\" id=\"" + id + "\">: no shared transform of that id");
        }
        if (const xnode_t* la = t->child("lookat")) {
            const dvec3 o = parse_point(la->get("origin"), DIM_LENGTH, "lookat origin"), tg = parse_point(la->get("target"), DIM_LENGTH, "lookat target");
            const dvec3 up = la->attr("up") ? parse_point(la->get("up"), DIM_NONE, "lookat up") : default_up;
            return xform_t::lookat(o, tg, up);
        }
        xform_t M = xform_t::identity();
        for (auto& op : t->kids) {
            if (op.name == "matrix") {
                const auto v = split_list(op.get("value"));
                if (v.size() != 16) throw std::runtime_error("<matrix>: 16 values expected");
                double r[16];
                for (int i = 0; i < 16; ++i) r[i] = parse_quantity(v[i]).si();   // (the translation column carries length units)
                M = xform_t::from_rows(r) * M;
            } else if (op.name == "rotate") {
                const dvec3 ax = read_vec3(op, DIM_NONE, 0.0, "rotate axis");
                M = xform_t::rotate(ax.x, ax.y, ax.z, parse_dim(op.get("angle"), DIM_ANGLE, "rotate angle")) * M;
            } else if (op.name == "translate") {
                const dvec3 tr = read_vec3(op, DIM_LENGTH, 0.0, "translate");
                M = xform_t::translate(tr.x, tr.y, tr.z) * M;
            } else if (op.name == "scale") {
                const dvec3 sc = read_vec3(op, DIM_NONE, 1.0, "scale");
                M = xform_t::scale(sc.x, sc.y, sc.z) * M;
            } else
                throw std::runtime_error("<transform>: unsupported element <" + op.name + ">");
        }
        return M;
    }
    bool in_sensitivity(double lo, double hi) const {   // does [lo,hi] (metres) contain the sensor's sensitivity range?
        return mono ? (lo <= line_m && line_m <= hi) : (lo <= band_lo && band_hi <= hi);
    }
    // picks the <bin> whose wavelength_range holds the sensor's sensitivity (composite BSDFs and spectra, bsdf/composite.hpp:26-140)
    const xnode_t* pick_bin(const xnode_t& comp) const {
        for (auto& k : comp.kids) {
            if (k.name != "bin") continue;
            double lo, hi;
            if (!parse_length_range(k.get("wavelength_range"), lo, hi)) throw std::runtime_error("<bin>: wavelength_range expected");
            if (in_sensitivity(lo, hi)) return &k;
        }
        return nullptr;
    }
    // a spectrum node without its `scale` child (the emitters hand their scale to the builder separately, emitter_t::scale); the <bin>s of
    // piecewise_linear / composite spectra and nested spectra stay
    static xnode_t without_scale(const xnode_t& n) {
        xnode_t c = n;
        c.kids.erase(std::remove_if(c.kids.begin(), c.kids.end(), [](const xnode_t& k) { return k.get("name") == "scale"; }), c.kids.end());
        return c;
    }
    // real-valued spectrum node -> builder spectrum id.  `emitter` = the spectrum of an emitter: -2 when it has no overlap with the sensor's
    // sensitivity (a continuous spectrum under a monochromatic sensor carries no power at the line: the emitter is dropped, see the header of
    // this file).  Everything else (IORs, reflectances, scale factors) is baked as a continuous spectrum and evaluated at the sample's
    // wavenumber on the device, whatever the sensor; -2 only for a composite spectrum without a bin for the sensor.
    int spectrum(const xnode_t& n, bool emitter = false) {
        if (n.get("type") == "composite") {
            const xnode_t* bin = pick_bin(n);
            if (!bin) return -2;
            const xnode_t* s = bin->child("spectrum");
            if (!s) throw std::runtime_error("composite spectrum: <bin> without <spectrum>");
            return spectrum(*s, emitter);
        }
        double scale = 1.0;
        if (const xnode_t* sc = n.named("scale")) scale = eval_number(sc->get("value"));
        if (n.get("type") == "discrete") {
            const quantity_t q = parse_wavelength(n.get("wavelength"), "discrete spectrum wavelength");
            const double wl = q.si();
            const double val = n.attr("value") ? eval_number(n.get("value")) : 1.0;
            if (emitter && (mono ? std::fabs(wl - line_m) > 1e-9 * line_m : !(band_lo <= wl && wl <= band_hi))) return -2;
            return b.spectrum_discrete((float)q.in_mm(), (float)(val * scale));
        }
        if (n.attr("constant")) {
            double re, im;
            parse_complex(n.get("constant"), re, im);
            return b.spectrum_const((float)(re * scale), (float)(im * scale));
        }
        if (n.attr("rgb")) {
            const auto c = split_list(n.get("rgb"));
            if (c.size() != 3) throw std::runtime_error("rgb spectrum: three components expected");
            if (mono && emitter) return -2;   // RGB uplift is defined over 380..720 nm
            return b.spectrum_rgb((float)eval_number(c[0]), (float)eval_number(c[1]), (float)eval_number(c[2]));
        }
        // named database spectra (spectrum_from_db.cpp): the tables baked into the library (Al, Au, Ag, Cu, SF5, SF11, BK7; the CFL emission
        // spectrum), else a file <data dir>/ior/<name>.yml resp. <data dir>/emission/<name>.yml (data dir: $WTGPU_DATA_DIR, or "data" next to
        // the scene file, or ../../data as in the reference's checkout)
        auto data_file = [&](const char* sub, const std::string& name) {
            const std::string rel = std::string("/") + sub + "/" + name + ".yml";
            if (const char* env = getenv("WTGPU_DATA_DIR")) return std::string(env) + rel;
            // "data" next to the scene file, else the reference checkout's layout (scenes/<name>/x.xml, data/ beside scenes/)
            const std::string local = base_dir + "/data" + rel, checkout = base_dir + "/../../data" + rel;
            return !std::ifstream(local).good() && std::ifstream(checkout).good() ? checkout : local;
        };
        // tabulated spectra cannot fold a scale: the emitters pass theirs separately (emitter_t::scale), elsewhere it must be 1
        if ((n.attr("material") || n.attr("emitter")) && scale != 1.0) throw std::runtime_error("<spectrum>: a scale on a database spectrum is supported for emitters only");
        if (n.attr("material")) {
            const std::string m = n.get("material");
            for (const char* baked : {"Al", "Au", "Ag", "Cu", "SF5", "SF11", "BK7"})
                if (m == baked) return b.spectrum_named(m);
            return b.spectrum_ior_from_file(data_file("ior", m));
        }
        if (n.attr("emitter")) {
            const std::string e = n.get("emitter");
            if (mono && emitter) return -2;
            if (e == "2534_CFL_Tensor_Twister") return b.spectrum_named("CFL2534");
            return b.spectrum_emission_from_file(data_file("emission", e));
        }
        if (n.attr("ITU")) {   // ITU-R P.2040 building materials (src/spectrum/util/spectrum_from_ITU.cpp): radio frequencies, line sensors only
            if (!mono) throw std::runtime_error("<spectrum ITU=…>: supported for monochromatic sensors only");
            if (scale != 1.0) throw std::runtime_error("<spectrum ITU=…>: scale is not supported");
            return b.spectrum_itu(n.get("ITU"), (float)line_mm);
        }
        // piecewise_linear (<bin wavelength= value=/> knots, linear in WAVENUMBER between them, 0 outside: piecewise_linear.cpp:45-80) and
        // gaussian (value x exp(-(k - k0)^2 / 2 s^2) with s = k(mean) - k(mean + stddev), cut at 10 s: gaussian.cpp:60-77), baked on the
        // library's 0.5-nm table over 340..840 nm
        if (n.get("type") == "piecewise_linear" || n.get("type") == "gaussian") {
            if (mono && emitter) return -2;   // continuous spectrum x line sensor
            const int N = 1001;
            const double l0 = 340.0, dl = 0.5;
            std::vector<float> v(N, 0.f);
            if (n.get("type") == "gaussian") {
                const double mean = parse_wavelength(n.get("wavelength"), "gaussian spectrum wavelength").si(), sd = parse_wavelength(n.get("stddev"), "gaussian spectrum stddev").si();
                const double val = n.attr("value") ? eval_number(n.get("value")) : 1.0;
                if (val < 0 || sd < 0) throw std::runtime_error("(gaussian spectrum loader) a non-negative value and standard deviation must be provided");
                const double k0 = 2 * M_PI / mean, sk = k0 - 2 * M_PI / (mean + sd);
                for (int i = 0; i < N; ++i) {
                    const double k = 2 * M_PI / ((l0 + dl * i) * 1e-9), d = k - k0;
                    v[i] = sk > 0 && std::fabs(d) <= 10 * sk ? (float)(val * scale * std::exp(-.5 * d * d / (sk * sk))) : 0.f;
                }
            } else {
                std::vector<std::pair<double, double>> knots;   // (k, value)
                for (auto& c : n.kids)
                    if (c.name == "bin") {
                        const double wl = parse_wavelength(c.get("wavelength"), "piecewise_linear bin").si(), val = eval_number(c.get("value"));
                        if (!(wl > 0) || val < 0) throw std::runtime_error("(piecewise_linear spectrum loader) wavelength must be positive and value must be non-negative");
                        knots.push_back({2 * M_PI / wl, val * scale});
                    }
                if (knots.size() < 2) throw std::runtime_error("(piecewise_linear spectrum loader) at least 2 spectrum values must be provided");
                std::sort(knots.begin(), knots.end());
                for (int i = 0; i < N; ++i) {
                    const double k = 2 * M_PI / ((l0 + dl * i) * 1e-9);
                    if (k < knots.front().first || k > knots.back().first) continue;
                    size_t j = 0;
                    while (j + 2 < knots.size() && knots[j + 1].first < k) ++j;
                    const double f = (k - knots[j].first) / (knots[j + 1].first - knots[j].first);
                    v[i] = (float)(knots[j].second * (1 - f) + knots[j + 1].second * f);
                }
            }
            return b.spectrum_from_wavelength_table(v.data(), nullptr, N, (float)l0, (float)dl);
        }
        if (n.attr("blackbody")) {
            if (mono && emitter) return -2;   // continuous spectrum x line sensor: see the header of this file
            return b.spectrum_blackbody((float)parse_dim(n.get("blackbody"), DIM_TEMPERATURE, "blackbody"), (float)scale);
        }
        std::string desc = "<" + n.name;
        for (auto& at : n.attrs) desc += " " + at.first + "=\"" + at.second + "\"";
        throw std::runtime_error(desc + ">: unsupported kind of spectrum");
    }
    // an index of refraction: must resolve (the device reads a missing spectrum as 1, i.e. an index-matched interface)
    int ior_spectrum(const xnode_t& n, const char* what) {
        const int s = spectrum(n);
        if (s < 0) throw std::runtime_error(std::string(what) + ": the spectrum has no value at the sensor's wavelengths (a composite without a matching bin?)");
        return s;
    }
    // ---- textures (src/texture/texture_loader.cpp:30-62): constant, checkerboard (colour1 / colour2: a texture or a constant spectrum,
    // defaults 0 and 1), scale (constant `scale` spectrum x nested texture), transform (<matrix value="a,b,c,d"/>, <translate value="x,y"/>
    // on the uv of a nested texture), bitmap (<path>: a PNG (host/png_loader.cpp) or PFM file, colour_encoding, gamma, filter_type
    // nearest | bilinear | bicubic (the default, as in the reference), wrap_mode[_u|_v] black | white | clamp | repeat | mirror).  Luminance (wavelength-independent) values; RGB bitmaps
    // only for normal maps.  Returns the texture id.
    static float const_of(const xnode_t& sp, const char* what) {
        if (!sp.attr("constant")) throw std::runtime_error(std::string(what) + ": a constant spectrum is expected here");
        return (float)eval_number(sp.get("constant"));
    }
    int texture_or_constant(const xnode_t& n) { return n.name == "spectrum" ? b.add_texture_constant(const_of(n, "texture colour"), const_of(n, "texture colour"), const_of(n, "texture colour")) : texture(n); }
    static uint32_t wrap_of(const std::string& w) {
        static const char* names[] = {"black", "white", "clamp", "repeat", "mirror"};
        for (uint32_t i = 0; i < 5; ++i)
            if (w == names[i]) return i;
        throw std::runtime_error("unknown wrap mode \"" + w + "\"");
    }
    int texture(const xnode_t& n) {
        std::string type = n.get("type");
        if (type.empty() && n.attr("bitmap")) type = "bitmap";
        if (type.empty() && n.attr("function")) type = "function";   // <texture name=… function="…"> (src/texture/texture_loader.cpp:46-53 shorthands)
        if (type == "constant") {
            const xnode_t* sp = n.child("spectrum");
            if (!sp) throw std::runtime_error("(constant texture loader) A nested real spectrum must be provided");
            const float v = const_of(*sp, "constant texture");
            return b.add_texture_constant(v, v, v);
        }
        if (type == "checkerboard") {
            const xnode_t *c1 = n.named("colour1"), *c2 = n.named("colour2");
            const int t1 = c1 ? texture_or_constant(*c1) : b.add_texture_constant(0.f, 0.f, 0.f);
            const int t2 = c2 ? texture_or_constant(*c2) : b.add_texture_constant(1.f, 1.f, 1.f);
            return b.add_texture_checkerboard(t1, t2);
        }
        if (type == "scale") {
            const xnode_t *sc = n.named("scale"), *in = n.child("texture");
            if (!sc || !in) throw std::runtime_error("scale texture: a `scale` spectrum and a nested texture expected");
            const int t = texture(*in);
            b.texture_set_scale(t, b.texture_scale(t) * const_of(*sc, "scale texture"));
            return t;
        }
        if (type == "transform") {
            const xnode_t* in = n.child("texture");
            if (!in) throw std::runtime_error("(transform texture loader) A nested texture must be provided");
            float M[4] = {1, 0, 0, 1}, T[2] = {0, 0};
            if (const xnode_t* m = n.child("matrix")) {
                const auto v = split_list(m->get("value"));
                if (v.size() != 4) throw std::runtime_error("transform texture: <matrix> needs 4 values");
                for (int i = 0; i < 4; ++i) M[i] = (float)eval_number(v[i]);
            }
            if (const xnode_t* tr = n.child("translate")) {
                const auto v = split_list(tr->get("value"));
                if (v.size() != 2) throw std::runtime_error("transform texture: <translate> needs 2 values");
                T[0] = (float)eval_number(v[0]);
                T[1] = (float)eval_number(v[1]);
            }
            const int t = texture(*in);
            b.texture_compose_transform(t, M, T);
            return t;
        }
        if (type == "bitmap") {
            const xnode_t* pth = n.child("path");
            std::string file = pth ? pth->get("value") : n.get("bitmap");
            if (file.empty()) throw std::runtime_error("(bitmap texture loader) path must be provided");
            if (file[0] != '/') file = base_dir + "/" + file;
            uint32_t bilinear = 2u;   // the filter: bicubic unless the scene says otherwise, like the reference (texture2d_storage.hpp:73)
            if (const xnode_t* f = n.named("filter_type")) {
                const std::string v = f->get("value");
                if (v != "nearest" && v != "bilinear" && v != "bicubic") throw std::runtime_error("bitmap filter_type \"" + v + "\" is not supported (nearest | bilinear | bicubic)");
                bilinear = v == "nearest" ? 0u : (v == "bilinear" ? 1u : 2u);
            }
            uint32_t uw = WRAP_REPEAT, vw = WRAP_REPEAT;
            if (const xnode_t* w = n.named("wrap_mode")) uw = vw = wrap_of(w->get("value"));
            if (const xnode_t* w = n.named("wrap_mode_u")) uw = wrap_of(w->get("value"));
            if (const xnode_t* w = n.named("wrap_mode_v")) vw = wrap_of(w->get("value"));
            // a Git-LFS pointer in place of the image (scenes/cornell-box/textures/tiles2.png in the reference's checkout): mid-grey,
            // as the bundled cornell_box scene uses
            if (is_lfs_pointer(file)) {
                std::fprintf(stderr, "wtgpu: %s is a Git-LFS pointer: mid-grey stands in for the image\n", file.c_str());
                return b.add_texture_constant(.5f, .5f, .5f);
            }
            uint32_t W = 0, H = 0, C = 0;
            std::string ext = file.size() >= 4 ? file.substr(file.size() - 4) : std::string();
            for (char& c : ext) c = (char)std::tolower((unsigned char)c);
            std::vector<float> px;
            if (ext == ".png") {   // colour_encoding = linear | sRGB | gamma, gamma = g (src/texture/bitmap.cpp:81-123)
                int enc = 0;
                double gamma = 2.2;
                if (const xnode_t* ce = n.named("colour_encoding")) {
                    const std::string v = ce->get("value");
                    enc = v == "linear" ? 1 : v == "sRGB" ? 2 : v == "gamma" ? 3 : -1;
                    if (enc < 0) throw std::runtime_error("bitmap colour_encoding \"" + v + "\": linear | sRGB | gamma expected");
                }
                if (const xnode_t* g = n.named("gamma")) {
                    if (enc != 0 && enc != 3) throw std::runtime_error("(bitmap texture loader) 'gamma' can only be set for 'gamma' colour encoding");
                    gamma = eval_number(g->get("value"));
                    enc = 3;
                }
                px = load_png(file, W, H, C, enc, gamma);
            } else if (ext == ".pfm")
                px = load_pfm(file, W, H, C);
            else if (ext == ".exr") {   // linear floats (src/bitmap/texture2d_loader.cpp:195-200, load2d.cpp:38-75)
                if (const xnode_t* ce = n.named("colour_encoding"))
                    if (ce->get("value") != "linear") throw std::runtime_error("(bitmap loader) " + file + ": an EXR image is read as linear; colour_encoding \"" + ce->get("value") + "\" is not applied to float texels here");
                px = load_exr(file, W, H, C);
            } else
                throw std::runtime_error("(bitmap loader) " + file + ": PNG (8 / 16 bit), EXR and PFM files only");
            return b.add_texture_bitmap(W, H, C, px.data(), bilinear, uw, vw);
        }
        if (type == "function") {   // src/texture/function.cpp:38-124: nested NAMED textures are the variables, then u, v, k; the expression inline
            // (function="...") or as a <function value="..."/> child
            std::vector<std::pair<std::string, std::pair<int, float>>> vars;
            for (const xnode_t& c : n.kids)
                if (c.name == "texture" || (c.name == "ref" && deref(&c)->name == "texture")) {
                    const std::string name = c.get("name");
                    if (name.empty()) throw std::runtime_error("(function texture loader) Nested texture must be given a name");
                    vars.push_back({name, {TOP_TEX, (float)texture(*deref(&c))}});
                }
            vars.push_back({"u", {TOP_U, 0.f}});
            vars.push_back({"v", {TOP_V, 0.f}});
            vars.push_back({"k", {TOP_K, 0.f}});
            std::string fn = n.get("function");
            if (const xnode_t* f = n.child("function")) fn = f->get("value");
            if (fn.empty()) throw std::runtime_error("(function texture loader) No function 'function' provided");
            std::vector<float> prog;
            expr_compiler_t(fn, vars, prog).compile();
            return b.add_texture_function(prog);
        }
        if (type == "mix") {   // src/texture/mix.cpp:34-62: texture1, texture2, mix (each a texture; constants as spectra are accepted too)
            const xnode_t *t1 = deref(n.named("texture1")), *t2 = deref(n.named("texture2")), *m = deref(n.named("mix"));
            if (!t1 || !t2 || !m) throw std::runtime_error("(mix texture loader) Nested textures 'texture1', 'texture2' and 'mix' must be provided");
            std::vector<float> prog;
            for (const xnode_t* t : {t1, t2, m}) {
                prog.push_back((float)TOP_TEX);
                prog.push_back((float)texture_or_constant(*t));
            }
            prog.push_back((float)TOP_MIX);
            prog.push_back(0.f);
            return b.add_texture_function(prog);
        }
        throw std::runtime_error("(texture loader) texture type \"" + type + "\" is not supported");
    }

    // extIOR, reflection_scale, transmission_scale of the two interface BSDFs (src/bsdf/dielectric.cpp:94-97, surface_spm.cpp:225-229);
    // the scales are constants here
    void interface_extras(const xnode_t& n, material_t& out) {
        if (const xnode_t* e = n.named("extIOR")) out.ext_ior_spec = ior_spectrum(*e, "extIOR");
        if (const xnode_t* r = n.named("reflection_scale")) out.refl_scale = const_of(*r, "reflection_scale");
        if (const xnode_t* t = n.named("transmission_scale")) out.trans_scale = const_of(*t, "transmission_scale");
    }
    // <ref name=… id=…/> in place of a spectrum / texture: the shared top-level element of that id (loader.cpp:168-181)
    const xnode_t* deref(const xnode_t* n) const {
        if (!n || n->name != "ref") return n;
        const std::string id = n->get("id");
        for (auto& it : items)
            if ((it.name == "texture" || it.name == "spectrum") && it.get("id") == id) return &it;
        throw std::runtime_error("<ref id=\"" + id + "\">: no shared texture or spectrum of that id");
    }
    // what a wrapper wraps: a nested <bsdf>, or <ref id=…/> = a copy of a named BSDF's record.  FALSE: spectrally empty.
    bool nested(const xnode_t& n, bool two_sided, material_t& out, const char* what) {
        if (const xnode_t* in = n.child("bsdf")) return material(*in, two_sided, out);
        if (const xnode_t* r = n.child("ref")) {
            const auto it = materials.find(r->get("id"));
            if (it == materials.end()) return false;
            out = b.material(it->second);
            if (two_sided) out.two_sided = 1;
            return true;
        }
        throw std::runtime_error(std::string(what) + " without a nested <bsdf> or <ref id=…/>");
    }
    // bsdf node -> material (two_sided accumulated from the wrappers)
    bool material(const xnode_t& n, bool two_sided, material_t& out) {
        const std::string type = n.get("type");
        if (type == "twosided") return nested(n, true, out, "twosided bsdf");
        if (type.empty() && n.attr("scale")) {   // scale wrapper (bsdf/scale.hpp) with a constant
            if (!nested(n, two_sided, out, "scale bsdf")) return false;
            out.scale *= (float)eval_number(n.get("scale"));
            return true;
        }
        if (type == "scale") {   // <bsdf type="scale"><spectrum name="scale" constant=…/><bsdf …/></bsdf> (src/bsdf/scale.cpp:50-57)
            const xnode_t* sc = deref(n.named("scale"));
            if (!sc) throw std::runtime_error("scale bsdf: a `scale` spectrum or texture expected");
            if (!nested(n, two_sided, out, "scale bsdf")) return false;
            if (sc->name == "texture") {   // a texture (luminance): material_t::scale_tex
                if (out.scale_tex) throw std::runtime_error("scale bsdf: nested textured scales are not supported");
                out.scale_tex = 1 + (uint32_t)texture(*sc);
            } else if (sc->attr("constant"))
                out.scale *= const_of(*sc, "scale bsdf");
            else {   // a spectrum (rgb uplift, tables …): evaluated per wavenumber on the device (material_t::scale_spec)
                if (out.scale_spec) throw std::runtime_error("scale bsdf: nested spectral scales are not supported");
                const int sp = spectrum(*sc);
                if (sp == -2) return false;
                out.scale_spec = 1 + (uint32_t)sp;
            }
            return true;
        }
        if (type == "dielectric") {
            const xnode_t* ior = n.named("IOR");
            if (!ior) throw std::runtime_error("dielectric bsdf: IOR expected");
            out = mat_dielectric(ior_spectrum(*ior, "dielectric IOR"));
            out.two_sided = two_sided;
            interface_extras(n, out);
            return true;
        }
        if (type == "composite") {
            const xnode_t* bin = pick_bin(n);
            if (!bin) return false;
            const xnode_t* in = bin->child("bsdf");
            if (!in) throw std::runtime_error("composite bsdf: <bin> without <bsdf>");
            return material(*in, two_sided, out);
        }
        if (type == "diffuse") {
            const xnode_t* r = deref(n.named("reflectance"));
            if (!r) throw std::runtime_error("diffuse bsdf: reflectance expected");
            if (r->name == "texture") {
                // reflectance = spectrum x luminance texture: a `scale` texture's spectrum carries the wavelength dependence
                // (scenes/cornell-box/box.xml:34-41: scale 0.35 x bitmap), anything else is a grey texture
                if (r->get("type") == "scale" || r->attr("scale")) {
                    const xnode_t *sc = r->named("scale"), *in = r->child("texture");
                    if (!sc || !in) throw std::runtime_error("scale texture: a `scale` spectrum and a nested texture expected");
                    const int s = spectrum(*sc);
                    if (s == -2) return false;
                    out = mat_diffuse(s, 1.f, two_sided);
                    out.refl_tex = 1 + (uint32_t)texture(*in);
                } else {
                    out = mat_diffuse(b.spectrum_const(1.f), 1.f, two_sided);
                    out.refl_tex = 1 + (uint32_t)texture(*r);
                }
                return true;
            }
            const int s = spectrum(*r);
            if (s == -2) return false;
            out = mat_diffuse(s, 1.f, two_sided);
            return true;
        }
        if (type == "mask") {   // src/bsdf/mask.cpp:94-124: a `mask` texture (or constant spectrum) and a nested bsdf
            const xnode_t* mk = deref(n.named("mask"));
            if (!mk) throw std::runtime_error("(mask bsdf loader) a real 'mask' spectrum must be provided");
            material_t inner{};
            if (!nested(n, false, inner, "(mask bsdf loader) 'mask' bsdf")) return false;
            const int nested = b.add_material(inner);
            out = mat_mask(nested, mk->name == "spectrum" ? const_of(*mk, "mask") : 1.f, two_sided);
            if (mk->name == "texture") out.mask_tex = 1 + (uint32_t)texture(*mk);
            return true;
        }
        if (type == "normalmap") {   // bsdf/normalmap.hpp: the nested bsdf with a perturbed shading frame
            const xnode_t* nm = n.child("texture");   // the texture child needs no name (src/bsdf/normalmap.cpp:43-45)
            if (!nm) throw std::runtime_error("normalmap bsdf: a texture and a nested bsdf expected");
            if (!nested(n, two_sided, out, "normalmap bsdf")) return false;
            const int nt = texture(*nm);
            if (b.texture_is_function(nt)) throw std::runtime_error("normalmap bsdf: a function / mix texture as the normal map is not supported (its RGB lookup knows constant, checkerboard and bitmap textures)");
            out.normal_tex = 1 + (uint32_t)nt;
            for (const char* name : {"flip_tangent", "flip"})
                if (const xnode_t* f = n.named(name)) out.normal_flip = eval_number(f->get("value")) != 0.0 ? 1u : 0u;
            return true;
        }
        if (type == "surface_spm") {
            const xnode_t* ior = n.named("IOR");
            if (!ior) throw std::runtime_error("surface_spm bsdf: IOR expected");
            const int s = ior_spectrum(*ior, "surface_spm IOR");
            // surface profiles (src/interaction/surface_profile/{dirac,fractal,gaussian}.cpp): the roughness-parametrised forms with a
            // constant roughness (T / sigma_h resp. sigma textures are not supported)
            bool fractal = false, gaussian = false;
            float roughness = 0.f, gamma = 3.f, gauss_sigma = 0.f;
            uint32_t rough_tex = 0;   // a textured roughness (fractal.cpp:94-123: `roughness` is a texture; a constant spectrum is the constant texture)
            if (const xnode_t* sp = n.child("surface_profile")) {
                const std::string pt = sp->get("type");
                if (pt == "fractal" || pt == "gaussian") {
                    (pt == "fractal" ? fractal : gaussian) = true;
                    const xnode_t *ro = sp->named("roughness"), *sg = pt == "gaussian" ? sp->named("sigma") : nullptr;
                    if (sg) {   // explicit rms roughness [1/length] (gaussian.cpp:44-61: either `roughness` or `sigma`)
                        if (ro) throw std::runtime_error("(gaussian surface_profile loader) Either 'roughness' or 'sigma' must be provided");
                        const std::string v = trim(sg->get("value"));
                        static const std::pair<const char*, double> units[] = {{"1/mm", 1.0}, {"1/um", 1e3}, {"1/m", 1e-3}, {"1/cm", 1e-1}, {"1/nm", 1e6}};
                        bool ok = false;
                        for (auto& u : units) {
                            const std::string suf(u.first);
                            if (v.size() > suf.size() && v.compare(v.size() - suf.size(), suf.size(), suf) == 0) {
                                gauss_sigma = (float)(eval_number(v.substr(0, v.size() - suf.size())) * u.second);
                                ok = true;
                                break;
                            }
                        }
                        if (!ok || !(gauss_sigma > 0.f)) throw std::runtime_error("gaussian profile: sigma \"" + v + "\": a positive value in 1/mm, 1/um, 1/m, 1/cm or 1/nm expected");
                    } else {
                        ro = deref(ro);
                        if (!ro) throw std::runtime_error(pt + " profile: `roughness` expected");
                        if (ro->name == "texture") {
                            rough_tex = 1 + (uint32_t)texture(*ro);
                            roughness = 1.f;   // (placeholder: evaluated per interaction, wt/bsdf.h: material_resolve)
                        } else {
                            if (!ro->attr("constant")) throw std::runtime_error(pt + " profile: a constant `roughness` spectrum or a texture is expected");
                            roughness = (float)eval_number(ro->get("constant"));
                        }
                    }
                    if (const xnode_t* g = sp->named("gamma")) gamma = (float)eval_number(g->get("value"));
                } else if (pt != "dirac")
                    throw std::runtime_error("surface_profile type \"" + pt + "\" is not supported (dirac | fractal | gaussian)");
            }
            out = mat_spm(s, fractal, roughness, gamma, two_sided, 1.f);
            out.rough_tex = rough_tex;
            if (gaussian) {
                out.profile = PROFILE_GAUSSIAN;
                out.gauss_sigma = gauss_sigma;
            }
            interface_extras(n, out);
            return true;
        }
        throw std::runtime_error("bsdf type \"" + type + "\" is not supported by the minimal reader");
    }

    void load(const std::string& path, const scene_params_t& prm) {
        const std::string text = read_file(path);
        xml_parser_t p(text, path);
        std::vector<xnode_t> top = p.top_level();
        if (top.size() != 1 || top[0].name != "scene") throw std::runtime_error(path + ": a single <scene> element expected");
        base_dir = dir_of(path);
        mesh_detail = prm.mesh_detail;
        splice(std::move(top[0].kids), base_dir);

        // ---- integrator
        integrator_opts_t o{};
        o.max_depth = 1024;
        o.MIS = o.RR = o.FSD = o.sensor_direct = o.emitter_direct = 1;
        for (auto& n : items) {
            if (n.name != "integrator" || !enabled(n)) continue;
            const std::string type = n.get("type");
            if (type == "plt_bdpt")
                o.integrator = INTEGRATOR_BDPT;
            else if (type == "plt_path") {
                const xnode_t* d = n.named("direction");
                if (!d) throw std::runtime_error("(plt_path integrator loader) 'direction' must be specified");
                const std::string dir = d->get("value");
                if (dir != "forward" && dir != "backward") throw std::runtime_error("plt_path: direction forward | backward expected");
                o.integrator = dir == "forward" ? INTEGRATOR_PATH_FORWARD : INTEGRATOR_PATH_BACKWARD;
            } else
                throw std::runtime_error("integrator type \"" + type + "\" is not supported");
            if (const xnode_t* a = n.named("max_depth")) o.max_depth = (int32_t)eval_number(a->get("value"));
            if (const xnode_t* a = n.named("MIS")) o.MIS = eval_number(a->get("value")) != 0.0;
            if (const xnode_t* a = n.named("FSD")) o.FSD = eval_number(a->get("value")) != 0.0;
            if (const xnode_t* a = n.named("russian_roulette")) o.RR = eval_number(a->get("value")) != 0.0;
            if (const xnode_t* a = n.named("sensor_direct_sampling")) o.sensor_direct = eval_number(a->get("value")) != 0.0;
            if (const xnode_t* a = n.named("emitter_direct_sampling")) o.emitter_direct = eval_number(a->get("value")) != 0.0;
        }
        apply_opts(prm, o);
        b.set_integrator(o);
        if (prm.lut_m) b.set_fsd_lut_resolution(prm.lut_n_theta, prm.lut_m);

        // ---- <sampler> (src/scene/loader/loader.cpp:192-199, src/sampler/sampler_loader.cpp:24-33): optional, at most one, of type independent /
        // uniform / sobolld.  Every type is served by the library's counter-based streams (Philox-4x32-10 keyed by seed, pixel, sample, stream:
        // DESIGN.md section 5) — sample sequences are not the reference's for any of them, `sobolld`'s low-discrepancy points included.
        {
            const xnode_t* smp = nullptr;
            for (auto& n : items)
                if (n.name == "sampler" && enabled(n)) {
                    if (smp) throw std::runtime_error("only one sampler must be provided");
                    smp = &n;
                }
            if (smp) {
                const std::string t = smp->get("type");
                if (t != "independent" && t != "uniform" && t != "sobolld") throw std::runtime_error("sampler type \"" + t + "\" is not recognised");
            }
        }

        // ---- the enabled sensor (exactly one: one wtgpu_scene renders one film)
        const xnode_t* sensor = nullptr;
        for (auto& n : items)
            if (n.name == "sensor" && enabled(n)) {
                if (sensor) throw std::runtime_error("more than one enabled sensor: select one with -D defines");
                sensor = &n;
            }
        if (!sensor) throw std::runtime_error("no enabled sensor");
        const xnode_t* film = sensor->child("film");
        if (!film || film->get("type") != "array") throw std::runtime_error("sensor: <film type=\"array\"> expected");
        const xnode_t *fw = film->named("width"), *fh = film->named("height");
        if (!fw || !fh) throw std::runtime_error("film: width and height expected");
        const double Wd = eval_number(fw->get("value")), Hd = eval_number(fh->get("value"));
        if (!(Wd >= 1.0 && Wd <= 65536.0) || !(Hd >= 0.0 && Hd <= 65536.0)) throw std::runtime_error("film: width / height out of range (1..65536)");
        const uint32_t W = (uint32_t)Wd, H = std::max(1u, (uint32_t)Hd);
        const xnode_t* resp = film->child("response");
        if (!resp) throw std::runtime_error("film: <response> expected");
        if (resp->get("type") == "monochromatic") {
            const xnode_t* sp = resp->child("spectrum");
            if (!sp || sp->get("type") != "discrete") throw std::runtime_error("monochromatic response: discrete spectrum expected");
            mono = true;
            const quantity_t q = parse_wavelength(sp->get("wavelength"), "response wavelength");
            line_m = q.si();
            line_mm = q.in_mm();
        } else if (resp->get("type") == "RGB") {
            band_lo = 380e-9;
            band_hi = 720e-9;
        } else
            throw std::runtime_error("response type \"" + resp->get("type") + "\" is not supported");
        const std::string stype = sensor->get("type");
        if (stype == "virtual_plane") {
            const xnode_t* ext = sensor->named("extent");
            if (!ext) throw std::runtime_error("virtual_plane sensor: extent expected");
            const auto e2 = split_list(ext->get("value"));
            if (e2.size() != 2) throw std::runtime_error("virtual_plane sensor: extent needs two components");
            float tan_alpha = -1.f;
            if (const xnode_t* a = sensor->named("alpha")) tan_alpha = (float)std::tan(parse_dim(a->get("value"), DIM_ANGLE, "alpha"));
            b.set_sensor_virtual_plane(to_world(*sensor, {0, 1, 0}), parse_dim(e2[0], DIM_LENGTH, "extent"), parse_dim(e2[1], DIM_LENGTH, "extent"), W, H, tan_alpha);
        } else if (stype == "perspective") {
            const xnode_t* fov = sensor->named("fov");
            if (!fov) throw std::runtime_error("perspective sensor: fov expected");
            bool rto = false;
            if (const xnode_t* r = sensor->named("ray_trace_only")) rto = eval_number(r->get("value")) != 0.0;
            float pse = 1.f;
            if (const xnode_t* r = sensor->named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
            // `fov` is the VERTICAL field of view unless fov_axis = "x" (src/sensor/perspective.cpp:65,113-137)
            double fov_y = parse_dim(fov->get("value"), DIM_ANGLE, "fov");
            if (const xnode_t* ax = sensor->named("fov_axis")) {
                const std::string v = ax->get("value");
                if (v != "x" && v != "y") throw std::runtime_error("(perspective sensor loader) unsupported 'fov_axis'");
                if (v == "x") fov_y = 2.0 * std::atan(std::tan(fov_y / 2) / (double(W) / double(H)));
            }
            b.set_sensor_perspective(to_world(*sensor, {0, 1, 0}), fov_y, W, H, pse, rto);
        } else
            throw std::runtime_error("sensor type \"" + stype + "\" is not supported");
        if (const xnode_t* r = film->named("rfilter_scale")) b.set_film_rfilter_scale((float)eval_number(r->get("value")));
        // <sensor … polarimetric="true"> (src/sensor/sensor_loader.cpp:28-38: the sensor records Stokes vectors); the parameter overrides
        bool polarimetric = sensor->attr("polarimetric") && eval_number(sensor->get("polarimetric")) != 0.0;
        if (prm.polarimetric > 0) polarimetric = true;
        if (polarimetric) b.set_sensor_polarimetric(true);
        if (mono)
            b.set_response_mono_discrete((float)line_mm);
        else {
            std::string wp = "D65";
            if (const xnode_t* w = resp->named("white_point")) wp = w->get("value");
            const float D50[3] = {0.96422f, 1.00000f, 0.82521f}, D65[3] = {0.95047f, 1.00000f, 1.08883f};
            const float E[3] = {1.f, 1.f, 1.f}, D55[3] = {0.95682f, 1.00000f, 0.92149f};
            if (wp != "D50" && wp != "D55" && wp != "D65" && wp != "E") throw std::runtime_error("white point \"" + wp + "\" is not supported");
            b.set_response_rgb(wp == "D50" ? D50 : wp == "D55" ? D55 : wp == "E" ? E : D65);
        }

        // ---- emitters, materials, shapes, in file order
        // Emitter order (it decides which random numbers select which emitter): the reference lists the free emitters ordered by
        // element id — unnamed elements are numbered "__unnamed_$<n>" over the enabled top-level elements, compared as strings —
        // then the area emitters in shape order (src/scene/loader/loader.cpp:131-133,272-310)
        uint32_t n_emitters = 0, unnamed_ids = 0;
        struct emitter_key_t {
            int cls;
            std::string id;
            int index;
        };
        std::vector<emitter_key_t> emitter_keys;
        for (auto& n : items) {
            std::string element_id = n.get("id");
            if (enabled(n) && element_id.empty()) element_id = "__unnamed_$" + std::to_string(++unnamed_ids);
            if (n.name == "emitter") {
                if (!enabled(n)) continue;
                const std::string type = n.get("type");
                if (type == "spot") {
                    const xnode_t* sp = n.named("radiant_intensity");
                    if (!sp) throw std::runtime_error("spot emitter: radiant_intensity expected");
                    double scale = 1.0;
                    if (const xnode_t* sc = sp->named("scale")) scale = eval_number(sc->get("value"));
                    const xnode_t unscaled = without_scale(*sp);   // the builder takes the scale separately (emitter_t::scale)
                                        const int s = spectrum(unscaled, true);
                    if (s == -2) continue;
                    const xnode_t *bw = n.named("beam_width"), *co = n.named("cutoff_angle");
                    if (!co) throw std::runtime_error("spot emitter: cutoff_angle expected");
                    float pse = 1.f;
                    if (const xnode_t* r = n.named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
                    const double cutoff = parse_dim(co->get("value"), DIM_ANGLE, "cutoff_angle");
                    b.add_emitter_spot(to_world(n, {0, 1, 0}), s, (float)scale, (float)cutoff,
                                       bw ? (float)parse_dim(bw->get("value"), DIM_ANGLE, "beam_width") : (float)(cutoff * .75) /* spot.cpp:120-121 */, -1.f, pse);
                    emitter_keys.push_back({0, element_id, (int)n_emitters});
                    ++n_emitters;
                } else if (type == "point") {   // src/emitter/point.cpp: position, radiant_intensity, optional spatial_extent
                    const xnode_t* sp = n.named("radiant_intensity");
                    if (!sp) throw std::runtime_error("point emitter: radiant_intensity expected");
                    double scale = 1.0;
                    if (const xnode_t* sc = sp->named("scale")) scale = eval_number(sc->get("value"));
                    const xnode_t unscaled = without_scale(*sp);
                                        const int s = spectrum(unscaled, true);
                    if (s == -2) continue;
                    const xnode_t* pos = n.named("position");
                    if (!pos) throw std::runtime_error("point emitter: position expected");
                    float pse = 1.f, extent = -1.f;
                    if (const xnode_t* r = n.named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
                    if (const xnode_t* r = n.named("spatial_extent")) extent = (float)parse_dim(r->get("value"), DIM_LENGTH, "spatial_extent");
                    b.add_emitter_point(read_vec3(*pos, DIM_LENGTH, 0.0, "position"), s, (float)scale, extent, pse);
                    emitter_keys.push_back({0, element_id, (int)n_emitters});
                    ++n_emitters;
                } else if (type == "directional") {
                    const xnode_t* sp = n.named("irradiance");
                    if (!sp) throw std::runtime_error("directional emitter: irradiance expected");
                    double scale = 1.0;
                    if (const xnode_t* sc = sp->named("scale")) scale = eval_number(sc->get("value"));
                    const xnode_t unscaled = without_scale(*sp);   // the builder takes the scale separately (emitter_t::scale)
                                        const int s = spectrum(unscaled, true);
                    if (s == -2) continue;
                    const xnode_t* t = n.named("to_world");
                    const xnode_t* la = t ? t->child("lookat") : nullptr;
                    // dir = to_world (0, 0, -1) (src/emitter/directional.cpp:115-120) is the direction TO the emitter: with a lookat (local
                    // +z from origin towards target) that is origin - target
                    dvec3 to_emitter{0, 0, -1};
                    if (la) {
                        const dvec3 og = parse_point(la->get("origin"), DIM_LENGTH, "lookat origin"), tg = parse_point(la->get("target"), DIM_LENGTH, "lookat target");
                        to_emitter = {og.x - tg.x, og.y - tg.y, og.z - tg.z};
                    } else if (t)
                        to_emitter = to_world(n, {0, 1, 0}).vector({0, 0, -1});
                    float sa = 6.794e-5f, pse = 1.f;   // the sun's solid angle (directional.cpp default)
                    if (const xnode_t* r = n.named("solid_angle")) {
                        std::string v = trim(r->get("value"));
                        if (v.size() > 2 && v.compare(v.size() - 2, 2, "sr") == 0) v = v.substr(0, v.size() - 2);
                        sa = (float)eval_number(v);
                        if (!(sa > 0.f)) throw std::runtime_error("(directional emitter loader) 'solid_angle' cannot be vanishing or negative");
                    }
                    if (const xnode_t* r = n.named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
                    b.add_emitter_directional(to_emitter, s, (float)scale, sa, pse);
                    emitter_keys.push_back({0, element_id, (int)n_emitters});
                    ++n_emitters;
                } else
                    throw std::runtime_error("emitter type \"" + type + "\" is not supported by the minimal reader");
            } else if (n.name == "bsdf") {
                const std::string id = n.get("id");
                material_t m{};
                if (id.empty()) throw std::runtime_error("top-level <bsdf> without id");
                if (material(n, false, m)) materials[id] = b.add_material(m);
            } else if (n.name == "shape") {
                if (!enabled(n)) continue;
                auto pt = [&](const char* name, bool required = true, dvec3 def = {0, 0, 0}) {
                    const xnode_t* q = n.named(name);
                    if (!q) {
                        if (required) throw std::runtime_error(std::string("shape: point ") + name + " expected");
                        return def;
                    }
                    return read_vec3(*q, DIM_LENGTH, 0.0, name);
                };
                auto len = [&](const char* name, double def) {
                    const xnode_t* q = n.named(name);
                    return q ? parse_dim(q->get("value"), DIM_LENGTH, name) : def;
                };
                auto num = [&](const char* name, double def) {
                    const xnode_t* q = n.named(name);
                    return q ? eval_number(q->get("value")) : def;
                };
                // the shape's material: a nested <bsdf> or <ref id=…>
                int mat = -1;
                if (const xnode_t* inl = n.child("bsdf")) {
                    material_t m{};
                    if (!material(*inl, false, m)) throw std::runtime_error("shape: its bsdf has no bin for the sensor's sensitivity");
                    mat = b.add_material(m);
                } else {
                    const xnode_t* ref = nullptr;   // <ref id=…/> or <ref name="bsdf" id=…/> (not the <ref name="to_world" …/> of a shared transform)
                    for (auto& k : n.kids)
                        if (k.name == "ref" && (k.get("name").empty() || k.get("name") == "bsdf")) {
                            ref = &k;
                            break;
                        }
                    if (!ref) throw std::runtime_error("(shape loader) no bsdf found");
                    const auto it = materials.find(ref->get("id"));
                    if (it == materials.end()) throw std::runtime_error("shape refers to unknown or spectrally empty material \"" + ref->get("id") + "\"");
                    mat = it->second;
                }
                // defaults: src/scene/shape.cpp:196-380
                const std::string type = n.get("type");
                // (a bound, so that a typing error ends in a message and not in 10^12 triangles; the reference has none)
                auto tess = [&](int def) {
                    const double t = num("tessellation", def);
                    if (!(t >= 3 && t <= 4096)) throw std::runtime_error("(shape loader) " + type + ": tessellation " + std::to_string(t) + " — 3 … 4096 expected");
                    return (int)t;
                };
                xform_t M = to_world(n, {0, 1, 0});
                bool face_normals = false;
                if (const xnode_t* fn = n.named("face_normals")) face_normals = eval_number(fn->get("value")) != 0.0;
                mesh_t mesh;
                if (type == "rectangle") {
                    if (n.named("p"))
                        mesh = mesh_rectangle(pt("p"), pt("x"), pt("y"));
                    else
                        mesh = mesh_rectangle_scaled(len("length", 2.0));
                } else if (type == "cube")
                    mesh = mesh_cube(len("length", 2.0));
                else if (type == "sphere")
                    mesh = mesh_sphere(pt("center", false), len("radius", 1e-3), tess(32));
                else if (type == "cylinder")
                    mesh = mesh_cylinder(pt("p0"), pt("p1"), len("radius", 1e-3), tess(32));
                else if (type == "prism") {
                    const xnode_t* a = n.named("angle");
                    mesh = mesh_prism(len("length", 1.0), len("height", 1.0), a ? parse_dim(a->get("value"), DIM_ANGLE, "angle") : M_PI / 2);
                } else if (type == "lens")
                    mesh = mesh_lens(pt("center", false), len("radius", 1e-3), num("R1", 0), num("R2", 0), len("thickness", 0.0), tess(50));
                else if (type == "ply" || type == "obj") {
                    const xnode_t* pth = n.child("path");
                    if (!pth) throw std::runtime_error(type + " shape: <path value=…/> expected");
                    const std::string file = pth->get("value");
                    const std::string full = file.size() && file[0] == '/' ? file : base_dir + "/" + file;
                    xform_t Ms = M;
                    bool fns = face_normals;
                    if (is_lfs_pointer(full) && asset_standin_mesh(file, mesh_detail, mesh, Ms, fns)) {
                        std::fprintf(stderr, "wtgpu: %s is a Git-LFS pointer: using the procedural stand-in\n", full.c_str());
                        M = Ms;   // the stand-in is placed in world space (the asset's model units are unknown)
                        face_normals = fns;
                    } else if (is_lfs_pointer(full)) {
                        // -Dwtgpu_missing_assets=skip: leave such shapes out (to inspect what else a scene file describes)
                        const auto sk = defs.find("wtgpu_missing_assets");
                        if (sk != defs.end() && sk->second == "skip") {
                            std::fprintf(stderr, "wtgpu: %s is a Git-LFS pointer: shape skipped\n", full.c_str());
                            continue;
                        }
                        throw std::runtime_error(full + " is a Git-LFS pointer file: the asset is absent from this checkout and has no bundled stand-in");
                    }
                    else {
                        const xnode_t* mt = n.named("mtl");   // <string name="mtl" value=…/>: the faces of one OBJ material (src/scene/shape.cpp:358-391)
                        if (mt && type == "ply") throw std::runtime_error("(shape loader) ply shape do not support 'mtl'");
                        const std::string mtl_name = mt ? mt->get("value") : std::string();
                        mesh = type == "ply" ? load_ply(full, face_normals, len("scale", 1.0)) : load_obj(full, face_normals, len("scale", 1.0), mt ? &mtl_name : nullptr);
                    }
                } else
                    throw std::runtime_error("shape type \"" + type + "\" is not supported by the minimal reader");
                const int shape = b.add_shape(mesh, M, mat, face_normals);
                if (const xnode_t* em = n.child("emitter")) {   // area emitter on this shape (src/scene/shape.cpp:150-168)
                    if (em->get("type") != "area") throw std::runtime_error("(shape loader) emitter must be an area emitter");
                    const xnode_t* sp = em->named("radiance");
                    if (!sp) throw std::runtime_error("area emitter: radiance expected");
                    double scale = 1.0;
                    int spec = -2, radiance_tex = -1;
                    if (sp->name == "texture" || (sp->name == "ref" && deref(sp)->name == "texture")) {
                        // A radiance TEXTURE (area.hpp:103-116: radiance->f({uv, k}).x times the emitter's own `scale`).  A texture that is the same
                        // everywhere — a constant, or the mid-grey that stands in for an image missing from the checkout — makes a uniform emitter
                        // with the colour's spectrum (positions drawn uniformly over the shape, which is what the reference's tables reduce to).
                        // A bitmap makes a spatially varying emitter with per-triangle texel tables (src/emitter/area.cpp:153-260).  Checkerboard
                        // and function textures provide no mean_spectrum() in the reference either (area.cpp:322-323 throws).
                        const int t = texture(*deref(sp));
                        if (const xnode_t* sc = em->named("scale")) scale = eval_number(sc->get("value"));
                        float rgb[3];
                        if (b.texture_constant_rgb(t, rgb)) {
                            // a luminance texture (every constant texture is one) is wavelength independent — area.hpp: scale * radiance->f({uv, k}).x —
                            // at ANY wavenumber: a flat spectrum, not the RGB uplift, which is zero outside 380-720 nm (an IR / radio sensor would see a
                            // dark emitter); only a genuinely coloured stand-in (an RGB bitmap's mean colour) is uplifted
                            spec = (rgb[0] == rgb[1] && rgb[1] == rgb[2]) ? b.spectrum_const(rgb[0]) : b.spectrum_rgb(rgb[0], rgb[1], rgb[2]);
                        } else if (b.texture_is_bitmap(t))
                            radiance_tex = t;
                        else
                            throw std::runtime_error("(area emitter loader) the radiance texture must provide mean_spectrum().");
                    } else {
                        if (const xnode_t* sc = sp->named("scale")) scale = eval_number(sc->get("value"));
                        const xnode_t unscaled = without_scale(*sp);
                        spec = spectrum(unscaled, true);
                    }
                    if (radiance_tex >= 0) {
                        float pse = 1.f;
                        if (const xnode_t* r = em->named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
                        b.add_emitter_area_textured(shape, radiance_tex, (float)scale, pse);
                        emitter_keys.push_back({1, std::string(), (int)n_emitters});
                        ++n_emitters;
                    } else
                    if (spec != -2) {
                        float pse = 1.f;
                        if (const xnode_t* r = em->named("phase_space_extent_scale")) pse = (float)eval_number(r->get("value"));
                        b.add_emitter_area(shape, spec, (float)scale, pse);
                        emitter_keys.push_back({1, std::string(), (int)n_emitters});
                        ++n_emitters;
                    }
                }
            }
        }
        if (!n_emitters) throw std::runtime_error("(scene) no emitters overlap the sensor's sensitivity");
        std::stable_sort(emitter_keys.begin(), emitter_keys.end(), [](const emitter_key_t& x, const emitter_key_t& y) {
            return x.cls != y.cls ? x.cls < y.cls : x.cls == 0 ? x.id < y.id : x.index < y.index;
        });
        std::vector<int> order;
        bool reorder = false;
        for (size_t i = 0; i < emitter_keys.size(); ++i) {
            order.push_back(emitter_keys[i].index);
            reorder |= emitter_keys[i].index != (int)i;
        }
        if (reorder) b.permute_emitters(order);
    }
};

}   // namespace

// `defines`: "name=value" strings (the reference's -D command-line defines).  prm: res (if non-zero) becomes the define "res" unless
// given explicitly; max_depth / fsd / mis / rr / force_ray_tracing override the integrator element; lut_*: table resolution.
void build_scene_from_xml(const std::string& path, const std::vector<std::string>& defines, const scene_params_t& prm, scene_builder_t& b) {
    loader_t L(b);
    for (auto& d : defines) {
        const size_t e = d.find('=');
        if (e == std::string::npos || e == 0) throw std::runtime_error("define \"" + d + "\": name=value expected");
        L.defs[d.substr(0, e)] = d.substr(e + 1);
    }
    if (prm.res && !L.defs.count("res")) L.defs["res"] = std::to_string(prm.res);
    L.load(path, prm);
    b.finalize();
}

}   // namespace wth
