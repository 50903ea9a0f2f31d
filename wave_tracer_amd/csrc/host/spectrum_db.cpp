// wave_tracer_amd — spectrum database files: the refractiveindex.info-style YAML the reference ships under data/ior
// and data/emission and reads in src/spectrum/util/spectrum_from_db.cpp:83-140 — a DATA list whose entries are
//   type: tabulated nk | tabulated n | tabulated k   with a `data: |` block of "wavelength[um] value [value]" rows (piecewise linear), or
//   type: formula 1 | formula 2                       with `coefficients:` (Sellmeier: n^2 = 1 + A + sum_i B_i l^2 / (l^2 - C_i); formula 1
//                                                     squares the C_i)
// read with a line-oriented parser (no YAML library here) and resampled on the library's uniform wavelength grid (340..840 nm, 0.5 nm)
// exactly like tools/bake_spectra.py bakes the bundled tables, so a file read at run time and the same file baked give the same spectrum.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "scene_builder.h"
#include "spectra_data.h"

namespace wth {

namespace {

struct db_entry_t {
    std::string type;
    std::vector<std::array<double, 3>> rows;   // tabulated: wavelength [um], v1, v2
    std::vector<double> coeff;
};

std::string strip(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

std::vector<db_entry_t> read_db(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("(spectrum db) cannot open " + path);
    std::vector<db_entry_t> out;
    std::string line;
    bool in_data_list = false, in_block = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const std::string t = strip(line);
        if (t.empty() || t[0] == '#') continue;
        if (!std::isspace((unsigned char)line[0])) {   // a top-level key
            in_data_list = t.rfind("DATA:", 0) == 0;
            in_block = false;
            continue;
        }
        if (!in_data_list) continue;
        std::string u = t;
        if (u.rfind("- ", 0) == 0) {   // a new list entry
            out.emplace_back();
            in_block = false;
            u = strip(u.substr(2));
        }
        if (out.empty()) continue;
        if (u.rfind("type:", 0) == 0) {
            out.back().type = strip(u.substr(5));
            in_block = false;
        } else if (u.rfind("data:", 0) == 0) {
            in_block = true;   // "data: |" — the rows follow
        } else if (u.rfind("coefficients:", 0) == 0) {
            std::istringstream ls(u.substr(13));
            double v;
            while (ls >> v) out.back().coeff.push_back(v);
            in_block = false;
        } else if (u.find(':') != std::string::npos && !std::isdigit((unsigned char)u[0]) && u[0] != '-' && u[0] != '.') {
            in_block = false;   // another key (wavelength_range, ...): ignored
        } else if (in_block) {
            std::istringstream ls(u);
            std::array<double, 3> r{0, 0, 0};
            if (!(ls >> r[0] >> r[1])) throw std::runtime_error("(spectrum db) " + path + ": malformed data row: " + u);
            ls >> r[2];
            out.back().rows.push_back(r);
        }
    }
    for (auto& e : out) std::sort(e.rows.begin(), e.rows.end());
    if (out.empty()) throw std::runtime_error("(spectrum db) " + path + ": no DATA entries");
    return out;
}

// numpy.interp: clamped (left/right given explicitly)
double interp(const std::vector<std::array<double, 3>>& rows, int col, double x, double left, double right) {
    if (rows.empty()) return left;
    if (x < rows.front()[0]) return left;
    if (x > rows.back()[0]) return right;
    size_t hi = 1;
    while (hi < rows.size() && rows[hi][0] < x) ++hi;
    if (hi >= rows.size()) return rows.back()[col];
    const auto &a = rows[hi - 1], &b = rows[hi];
    const double t = b[0] > a[0] ? (x - a[0]) / (b[0] - a[0]) : 0.0;
    return a[col] + (b[col] - a[col]) * t;
}

}   // namespace

// complex IOR n + i k from a data/ior-style file
int scene_builder_t::spectrum_ior_from_file(const std::string& path) {
    const auto db = read_db(path);
    std::vector<float> n(SPD_N, 1.f), k(SPD_N, 0.f);
    for (const db_entry_t& d : db) {
        for (int i = 0; i < SPD_N; ++i) {
            const double lum = (SPD_LAMBDA_MIN_NM + SPD_LAMBDA_STEP_NM * i) * 1e-3;   // um
            if (d.type == "formula 1" || d.type == "formula 2") {
                double c[7] = {0, 0, 0, 0, 0, 0, 0};
                for (size_t j = 0; j < d.coeff.size() && j < 7; ++j) c[j] = d.coeff[j];
                double C1 = c[2], C2 = c[4], C3 = c[6];
                if (d.type == "formula 1") {
                    C1 *= C1;
                    C2 *= C2;
                    C3 *= C3;
                }
                const double l2 = lum * lum;
                const double n2 = 1 + c[0] + c[1] * l2 / (l2 - C1) + c[3] * l2 / (l2 - C2) + c[5] * l2 / (l2 - C3);
                n[i] = (float)std::sqrt(std::max(n2, 0.0));
            } else if (d.type == "tabulated nk") {
                n[i] = (float)interp(d.rows, 1, lum, d.rows.front()[1], d.rows.back()[1]);
                k[i] = (float)interp(d.rows, 2, lum, d.rows.front()[2], d.rows.back()[2]);
            } else if (d.type == "tabulated n") {
                n[i] = (float)interp(d.rows, 1, lum, d.rows.front()[1], d.rows.back()[1]);
            } else if (d.type == "tabulated k") {
                k[i] = (float)interp(d.rows, 1, lum, d.rows.front()[1], d.rows.back()[1]);
            } else
                throw std::runtime_error("(spectrum db) " + path + ": unsupported entry type \"" + d.type + "\"");
        }
    }
    return spectrum_from_wavelength_table(n.data(), k.data(), SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
}
// real emission / sensitivity spectrum from a data/emission-style file (wavelengths in nm, zero outside the table)
int scene_builder_t::spectrum_emission_from_file(const std::string& path) {
    const auto db = read_db(path);
    for (const db_entry_t& d : db)
        if (d.type.rfind("tabulated", 0) == 0) {
            std::vector<float> v(SPD_N);
            for (int i = 0; i < SPD_N; ++i) v[i] = (float)interp(d.rows, 1, SPD_LAMBDA_MIN_NM + SPD_LAMBDA_STEP_NM * i, 0.0, 0.0);
            return spectrum_from_wavelength_table(v.data(), nullptr, SPD_N, SPD_LAMBDA_MIN_NM, SPD_LAMBDA_STEP_NM);
        }
    throw std::runtime_error("(spectrum db) " + path + ": no tabulated entry");
}

}   // namespace wth
