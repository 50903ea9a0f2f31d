// wave_tracer_amd — host-side scene baking: procedural meshes -> world-space triangles -> binned-SAH BVH ->
// 8-wide collapse -> wedge/edge table -> spectral tables -> emitter/sensor sampling tables -> flattened wt::scene_t.
//
// This replaces, for the hot path's needs only, the reference's scene loader + ADS constructor (which cannot be
// built here: tinybvh/pugixml/... are absent):
//   src/ads/bvh_constructor.cpp:123-251 (SAH build; tinybvh replaced by an own binned-SAH builder),
//   src/ads/bvh8w_constructor.cpp:27-103,153-268 (collapse 3 binary levels into one 8-wide node),
//   include/wt/ads/edge_classification.hpp:31-238 (edges), src/mesh/*.cpp (procedural shapes),
//   src/scene/scene_build_sensor_sampling_data.cpp:40-150 (emitter x sensitivity tables).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../wt/scene.h"

namespace wth {

// mask of the Fraunhofer iCDF tables = chi_e(kFsdLutMaskScale2 * chi * |zeta|^2): the value that reproduces the reference's PA1 and PA2
// (fsd.hpp:59-61), see scene_builder.cpp: build_fsd_lut
constexpr double kFsdLutMaskScale2 = 17.0 / 4.0;

struct dvec3 {
    double x, y, z;
};
struct xform_t {   // row-major 4x4, column-vector convention: p' = M p
    double m[16];
    static xform_t identity();
    static xform_t translate(double x, double y, double z);
    static xform_t scale(double x, double y, double z);
    static xform_t rotate(double ax, double ay, double az, double angle_rad);
    static xform_t lookat(dvec3 origin, dvec3 target, dvec3 up);
    static xform_t from_rows(const double r[16]);
    xform_t operator*(const xform_t& o) const;
    dvec3 point(dvec3 p) const;
    dvec3 vector(dvec3 v) const;
    dvec3 normal(dvec3 n) const;   // inverse-transpose for normals (normalised)
};

struct mesh_t {
    std::vector<dvec3> verts;
    std::vector<dvec3> normals;     // empty: face normals
    std::vector<std::array<float, 2>> uvs;   // empty: no uv
    std::vector<std::array<uint32_t, 3>> tris;
};
mesh_t mesh_rectangle(dvec3 p, dvec3 x, dvec3 y);
mesh_t mesh_rectangle_scaled(double length);
mesh_t mesh_cube(double length);
mesh_t mesh_sphere(dvec3 centre, double r, int tessellation);
mesh_t mesh_blob(double r, int recursion, double bump, int freq, uint32_t seed);
// the same surface on a geodesic grid of frequency n (20 n^2 triangles: any count, not only 20 x 4^k)
mesh_t mesh_blob_geodesic(double r, int n, double bump, int freq, uint32_t seed);
// hexagram ("star of David") plate in the yz-plane, thickness along x, 2 x 12 s^2 + 24 s triangles, face normals
mesh_t mesh_star_plate(double outer_radius, double thickness, int s);
mesh_t mesh_cylinder(dvec3 p0, dvec3 p1, double radius, int tessellation);
mesh_t mesh_prism(double length, double height, double angle_rad);
// `lens` shape (src/mesh/lens.cpp): optical axis +x; R1, R2 = face curvatures in units of 1/radius (0 planar, +-1 half sphere, > 0 convex)
mesh_t mesh_lens(dvec3 centre, double radius, double R1c, double R2c, double thickness, int tessellation);
// PLY file (host/ply_loader.cpp; src/mesh/ply_loader.cpp:22-98): positions x `scale`, vertex normals unless face_normals, uvs, triangles
mesh_t load_ply(const std::string& path, bool face_normals, double scale);
// Wavefront OBJ (host/obj_loader.cpp; src/mesh/obj_loader.cpp:26-140): one vertex per face corner, polygons fan-triangulated
mesh_t load_obj(const std::string& path, bool face_normals, double scale, const std::string* mtl = nullptr);   // mtl: keep the faces of this material
// Portable float map (PF: RGB, Pf: grey; little or big endian), rows returned from the image's TOP (the file stores them bottom-up)
std::vector<float> load_pfm(const std::string& path, uint32_t& width, uint32_t& height, uint32_t& channels);
// PNG, bit depth 8 or 16 (host/png_loader.cpp; src/bitmap/load2d.cpp:200-300): normalised, linearised floats, rows from the top.
// encoding: 0 = the file's default (8 bit sRGB, 16 bit linear), 1 linear, 2 sRGB, 3 gamma
std::vector<float> load_png(const std::string& path, uint32_t& width, uint32_t& height, uint32_t& channels, int encoding, double gamma);
// OpenEXR scan-line files (host/exr_loader.cpp; src/bitmap/load2d.cpp:38-75): the data window, linear, through half precision; 4 channels (RGBA) or 1 (Y)
std::vector<float> load_exr(const std::string& path, uint32_t& width, uint32_t& height, uint32_t& channels);

class scene_builder_t {
public:
    scene_builder_t();
    // spectra
    int spectrum_const(float re, float im = 0.f);
    int spectrum_discrete(float wavelength_mm, float value);
    int spectrum_from_wavelength_table(const float* values_re, const float* values_im, int n, float lmin_nm, float lstep_nm);
    int spectrum_blackbody(float T, float scale);
    int spectrum_rgb(float r, float g, float b);   // RGB uplift (include/wt/spectrum/colourspace/RGB/RGB_to_spectral.hpp), 380..720 nm
    int spectrum_named(const std::string& name);
    // spectrum database files (host/spectrum_db.cpp; src/spectrum/util/spectrum_from_db.cpp:83-140): data/ior/*.yml, data/emission/*.yml
    int spectrum_ior_from_file(const std::string& path);
    int spectrum_emission_from_file(const std::string& path);   // "Al","Au","SF5","SF11","BK7","Ag","Cu","CFL2534","CMF_X/Y/Z"
    // real-valued spectrum evaluation on the host (for baking)
    float spectrum_eval(int id, float k) const;

    int add_material(const wt::material_t& m);
    // textures (include/wt/texture/*.hpp); ids index scene_t::textures (materials store id + 1)
    int add_texture_constant(float r, float g, float b, float a = 1.f);
    int add_texture_checkerboard(int tex1, int tex2);
    // float texels, rows from the image's top, 1..4 channels; wrap: wt::WRAP_*
    int add_texture_bitmap(uint32_t width, uint32_t height, uint32_t channels, const float* texels, uint32_t filter /* 0 nearest, 1 bilinear, 2 bicubic */, uint32_t uwrap, uint32_t vwrap);
    // function / mix textures (texture/function.hpp, texture/mix.hpp): a postfix program of (wt::TOP_*, argument) pairs; TOP_TEX arguments are
    // texture ids — a nested FUNCTION texture is inlined (its own transform / scale must be the identity: wrap its operands instead)
    int add_texture_function(const std::vector<float>& program);
    bool texture_is_function(int tex) const { return textures_.at(tex).type == wt::TEX_FUNCTION; }
    bool texture_is_bitmap(int tex) const { return textures_.at(tex).type == wt::TEX_BITMAP; }
    void texture_set_transform(int tex, const float M[4], const float t[2]);   // uv' = M uv + t (texture/transform.hpp)
    void texture_set_scale(int tex, float scale);                               // texture/scale.hpp with a constant scale
    float texture_scale(int tex) const { return textures_.at(tex).scale; }
    // TRUE (and its colour, scale included) if the texture is the same everywhere: a constant, whatever its transform
    bool texture_constant_rgb(int tex, float rgb[3]) const {
        const wt::texture_t& t = textures_.at(tex);
        if (t.type != wt::TEX_CONSTANT) return false;
        for (int c = 0; c < 3; ++c) rgb[c] = t.rgba[c] * t.scale;
        return true;
    }
    // wraps `tex` in another transform texture: uv' = M_tex (M uv + t) + t_tex
    void texture_compose_transform(int tex, const float M[4], const float t[2]);
    wt::material_t& material(int id) { return materials_[id]; }
    int add_shape(const mesh_t& mesh, const xform_t& to_world, int material, bool face_normals = false);
    int add_emitter_spot(const xform_t& to_world, int spectrum, float scale, float cutoff_rad, float falloff_rad, float extent_m, float pse_scale);
    int add_emitter_area(int shape, int spectrum, float scale, float pse_scale);
    // radiance = scale x the BITMAP texture `tex` over the shape's uv (per-triangle sampling tables: src/emitter/area.cpp:153-260)
    int add_emitter_area_textured(int shape, int tex, float scale, float pse_scale);
    int add_emitter_point(dvec3 position, int spectrum, float scale, float extent_m, float pse_scale);
    void permute_emitters(const std::vector<int>& new_order);   // new_order[i]: current index of the emitter that becomes emitter i
    // directional (sun-like) emitter: `dir_to_emitter`, irradiance spectrum, solid angle subtended at the target (default: the sun's)
    int add_emitter_directional(dvec3 dir_to_emitter, int spectrum, float scale, float solid_angle_sr, float pse_scale);
    // ITU-R P.2040 material IOR at one wavelength (src/spectrum/util/spectrum_from_ITU.cpp): a constant complex spectrum
    int spectrum_itu(const std::string& material, float wavelength_mm);

    void set_sensor_perspective(const xform_t& to_world, double fov_rad, uint32_t w, uint32_t h, float pse_scale, bool ray_trace_only);
    void set_sensor_virtual_plane(const xform_t& to_world, double extent_x, double extent_y, uint32_t w, uint32_t h, float tan_alpha);
    void set_film_rfilter_scale(float s);
    void set_sensor_polarimetric(bool on) { sc_.sensor.polarimetric = on ? 1u : 0u; }
    // response: "RGB" (CIE colourspace, given white point XYZ) or monochromatic discrete line
    void set_response_rgb(const float white_xyz[3]);
    void set_response_mono_discrete(float wavelength_mm);
    void set_integrator(const wt::integrator_opts_t& o);
    void set_fsd_lut_resolution(uint32_t n_theta, uint32_t m);

    // builds everything; the returned scene points into this builder's storage
    const wt::scene_t& finalize();
    const wt::scene_t& scene() const { return sc_; }
    std::string stats() const;

private:
    struct shape_rec_t {
        int material;
        int emitter;
        std::vector<uint32_t> tri_first;   // index of first world tri
        uint32_t tri_begin, tri_count;
    };
    struct wtri_t {
        wt::vec3 a, b, c, n;
        wt::vec3 n0, n1, n2;
        wt::vec2 uv0, uv1, uv2;
        uint32_t has_uv;
        uint32_t shape, shape_tri;
    };
    void build_bvh();
    void build_edges();
    void build_sampling_tables();
    void build_fsd_lut();

    std::vector<wtri_t> wtris_;   // world triangles in insertion order
    std::vector<shape_rec_t> shape_recs_;
    // flattened storage
    std::vector<wt::tri_geo_t> tri_geo_;
    std::vector<wt::tri_meta_t> tri_meta_;
    std::vector<wt::tri_shade_t> tri_shade_;
    std::vector<wt::edge_t> edges_;
    std::vector<wt::bvh8_node_t> nodes_;
    std::vector<wt::bvh8_leaf_t> leaves_;
    std::vector<wt::shape_t> shapes_;
    std::vector<uint32_t> shape_tri_tuid_;
    std::vector<float> shape_tri_cdf_;
    std::vector<wt::material_t> materials_;
    std::vector<wt::spectrum_t> spectra_;
    std::vector<float> spectra_data_;
    std::vector<wt::texture_t> textures_;
    std::vector<float> texture_data_;
    std::vector<wt::emitter_t> emitters_;
    std::vector<float> emitter_cdf_;
    std::vector<wt::kdist_t> kdists_;
    std::vector<float> kdist_data_;
    std::vector<float> lut_theta1_, lut_theta2_, lut1_, lut2_;
    uint32_t lut_n_theta_ = 2048, lut_m_ = 3072;   // the reference's table sizes (fsd_lut.hpp:27: Nsamples, Msamples); 2 x 37.7 MB
    float rfilter_scale_ = 1.f;
    bool response_is_rgb_ = true;
    float mono_lambda_mm_ = 0.f;
    int sensitivity_spec_ = -1;
    wt::scene_t sc_;
    bool finalized_ = false;
    uint32_t bvh_max_depth_ = 0;
    double lut_power_[2] = {0, 0};
    // spectra whose support exceeds the baked table (blackbody: 8 nm .. 5 mm): (k [1/mm] ascending, value) knots of the reference's
    // piecewise-linear spectrum, for the emitter-selection weights (build_sampling_tables)
    std::map<int, std::vector<std::pair<double, double>>> spectrum_support_knots_;
    double emission_fraction_in_range(int spectrum, double k_lo, double k_hi) const;
    std::vector<double> emitter_in_range_;   // per emitter: that share (reported by stats())

public:
    double fsd_lut_power(int which) const { return lut_power_[which]; }
};

// bundled scenes (host/scenes.cpp)
struct scene_params_t {
    uint32_t res;
    int32_t max_depth;     // <0: scene default
    int32_t fsd;           // <0: default; 0/1
    int32_t mis, rr;       // <0: default
    int32_t force_ray_tracing;
    int32_t mesh_detail;   // 0: low-poly stand-ins (tests), 1: full stand-in tessellation
    uint32_t lut_n_theta, lut_m;
    uint32_t debug_only_s, debug_only_t;
    int32_t polarimetric;  // >0: polarimetric sensor (Stokes film)
    uint32_t crop_of;      // 0: off; else the film is the central res x res crop of a crop_of x crop_of film
};
// names: "double_slits", "cornell_box", "furnace" (diffuse box test scene), "white_furnace", "etoile" (plt_path forward + UTD),
// "furnace_path" (plt_path backward in the furnace scene)
bool build_named_scene(const std::string& name, const scene_params_t& p, scene_builder_t& b);
// material constructors / parameter overrides shared by the bundled scenes and the XML reader (host/scenes.cpp)
wt::material_t mat_diffuse(int refl_spec, float tex_scale, bool two_sided);
wt::material_t mat_spm(int ior_spec, bool fractal, float roughness, float gamma, bool two_sided, float scale);
wt::material_t mat_mask(int nested, float alpha, bool two_sided);
wt::material_t mat_dielectric(int ior_spec);
void apply_opts(const scene_params_t& p, wt::integrator_opts_t& o);
// Procedural stand-ins for the reference's Git-LFS assets that are absent from its checkout: `file` as written
// in scenes/cornell-box/box.xml.  The stand-in comes with its own to_world (the asset's model units are unknown).  FALSE: no stand-in.
bool asset_standin_mesh(const std::string& file, int mesh_detail, mesh_t& mesh, xform_t& to_world, bool& face_normals);
// minimal reader of the reference's XML scene format (host/xml_scene.cpp): `defines` = "name=value" (-D of the reference's CLI)
void build_scene_from_xml(const std::string& path, const std::vector<std::string>& defines, const scene_params_t& p, scene_builder_t& b);

}   // namespace wth
