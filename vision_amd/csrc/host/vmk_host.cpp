// vmk_host.cpp — C++ host: Vision JSON scene -> flat vmk tables (include/vmk.h).
//
// Mirrors Vision's scene front-end (plugin category/type names, per-type defaults, JSON spelling) but, instead of
// instantiating DSL-emitting plugin objects, each "plugin" here only ENCODES its parameters into SoA tables.
// Reference call stack being re-expressed (Vision `src/`):
//   JsonImporter::read_file -> SceneDesc::from_json             importers/json/importer.cpp:16-23, base/import/scene_desc.cpp:37-62
//   Scene::init (light_sampler, spectrum, materials, sensor, shapes, integrator, sampler)   base/mgr/scene.cpp:16-35
//   Scene::prepare (tidy_up, fill_instances, lights.prepare, materials.prepare)             base/mgr/scene.cpp:79-91,165-187
//   Geometry::update_instances / upload                                                      base/mgr/geometry.cpp:20-34,64-71
#include "../../../include/vmk_host.h"
#include "json.h"
#include "image_codec.h"
#include "exr.h"
#include "rgb2spec_opt.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

namespace vmk {

static thread_local std::string g_error;
struct HostError : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] static void fail(const std::string &m) { throw HostError(m); }

// ------------------------------------------------------------------------------------------------
// small double-precision matrix helpers (column-major 4x4, m[col*4+row] like ocarina float4x4)
// ------------------------------------------------------------------------------------------------
struct Mat4 {
    double m[16];
    static Mat4 identity() { Mat4 r{}; for (int i = 0; i < 4; ++i) r.m[i * 4 + i] = 1.0; return r; }
    double &at(int row, int col) { return m[col * 4 + row]; }
    double at(int row, int col) const { return m[col * 4 + row]; }
};
static Mat4 operator*(const Mat4 &a, const Mat4 &b) {
    Mat4 r{};
    for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) { double s = 0; for (int k = 0; k < 4; ++k) s += a.at(rr, k) * b.at(k, c); r.at(rr, c) = s; }
    return r;
}
static Mat4 translation(double x, double y, double z) { Mat4 r = Mat4::identity(); r.at(0, 3) = x; r.at(1, 3) = y; r.at(2, 3) = z; return r; }
static Mat4 scale(double x, double y, double z) { Mat4 r = Mat4::identity(); r.at(0, 0) = x; r.at(1, 1) = y; r.at(2, 2) = z; return r; }
static double radians(double d) { return d * M_PI / 180.0; }
static Mat4 rotation_x(double deg) { double c = std::cos(radians(deg)), s = std::sin(radians(deg)); Mat4 r = Mat4::identity(); r.at(1, 1) = c; r.at(1, 2) = -s; r.at(2, 1) = s; r.at(2, 2) = c; return r; }
static Mat4 rotation_y(double deg) { double c = std::cos(radians(deg)), s = std::sin(radians(deg)); Mat4 r = Mat4::identity(); r.at(0, 0) = c; r.at(0, 2) = s; r.at(2, 0) = -s; r.at(2, 2) = c; return r; }
static Mat4 rotation_z(double deg) { double c = std::cos(radians(deg)), s = std::sin(radians(deg)); Mat4 r = Mat4::identity(); r.at(0, 0) = c; r.at(0, 1) = -s; r.at(1, 0) = s; r.at(1, 1) = c; return r; }
static Mat4 inverse(const Mat4 &a) {
    double inv[16], det; const double *m = a.m;
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) fail("singular transform matrix");
    Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = inv[i] / det;
    return r;
}
struct V3 { double x, y, z; };
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 normalize(V3 a) { double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return {a.x / l, a.y / l, a.z / l}; }
// ocarina look_at<H>: returns the camera-to-world frame with columns (right, up, forward, position); consistent
// with Sensor::update_mat / camera_to_world_rotation (sensor.cpp:73-78,153-162): right = fwd x up, up' = right x fwd.
static Mat4 look_at(V3 pos, V3 target, V3 up) {
    V3 fwd = normalize(target - pos);
    V3 right = normalize(cross(fwd, up));
    V3 up2 = cross(right, fwd);
    Mat4 r = Mat4::identity();
    r.at(0, 0) = right.x; r.at(1, 0) = right.y; r.at(2, 0) = right.z;
    r.at(0, 1) = up2.x; r.at(1, 1) = up2.y; r.at(2, 1) = up2.z;
    r.at(0, 2) = fwd.x; r.at(1, 2) = fwd.y; r.at(2, 2) = fwd.z;
    r.at(0, 3) = pos.x; r.at(1, 3) = pos.y; r.at(2, 3) = pos.z;
    return r;
}
static V3 json_v3(const Json &j, V3 d) { if (!j.is_array() || j.size() < 3) return d; return {j.at(0).as_double(0), j.at(1).as_double(0), j.at(2).as_double(0)}; }
// TransformDesc::init (node_desc.cpp:31-58)
static Mat4 parse_transform(const Json &ps) {
    if (ps.is_null()) return Mat4::identity();
    std::string type = ps["type"].as_string("matrix4x4");
    const Json &param = ps["param"];
    if (type == "look_at") return look_at(json_v3(param["position"], {0, 0, 0}), json_v3(param["target_pos"], {0, 0, 1}), json_v3(param["up"], {0, 1, 0}));
    if (type == "Euler") {
        V3 p = json_v3(param["position"], {0, 0, 0});
        return translation(p.x, p.y, p.z) * rotation_x(param["pitch"].as_double(0)) * rotation_z(param["roll"].as_double(0)) * rotation_y(param["yaw"].as_double(0));
    }
    if (type == "trs") {
        V3 t = json_v3(param["t"], {0, 0, 0}), s = json_v3(param["s"], {1, 1, 1});
        const Json &r = param["r"]; // (axis xyz, angle degrees)
        double ax = 1, ay = 0, az = 0, ang = 0;
        if (r.is_array() && r.size() >= 4) { ax = r.at(0).as_double(1); ay = r.at(1).as_double(0); az = r.at(2).as_double(0); ang = r.at(3).as_double(0); }
        V3 a = normalize({ax, ay, az}); double c = std::cos(radians(ang)), sn = std::sin(radians(ang));
        Mat4 R = Mat4::identity();
        R.at(0, 0) = c + a.x * a.x * (1 - c); R.at(0, 1) = a.x * a.y * (1 - c) - a.z * sn; R.at(0, 2) = a.x * a.z * (1 - c) + a.y * sn;
        R.at(1, 0) = a.y * a.x * (1 - c) + a.z * sn; R.at(1, 1) = c + a.y * a.y * (1 - c); R.at(1, 2) = a.y * a.z * (1 - c) - a.x * sn;
        R.at(2, 0) = a.z * a.x * (1 - c) - a.y * sn; R.at(2, 1) = a.z * a.y * (1 - c) + a.x * sn; R.at(2, 2) = c + a.z * a.z * (1 - c);
        return translation(t.x, t.y, t.z) * R * scale(s.x, s.y, s.z);
    }
    if (type == "matrix4x4") {
        const Json &mm = param["matrix4x4"];
        Mat4 r = Mat4::identity();
        if (mm.is_array() && mm.size() == 4) for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) r.at(rr, c) = mm.at(c).at(rr).as_double(rr == c ? 1.0 : 0.0);
        return r;
    }
    fail("transform type error " + type);
}

// ------------------------------------------------------------------------------------------------
// images
// ------------------------------------------------------------------------------------------------
struct Image { uint32_t w{0}, h{0}, channels{0}; bool is_float{false}; std::vector<uint8_t> u8; std::vector<float> f32; };
static std::map<std::string, Image> g_images;

static bool ends_with(const std::string &s, const std::string &suf) { return s.size() >= suf.size() && std::equal(suf.rbegin(), suf.rend(), s.rbegin(), [](char a, char b) { return std::tolower(a) == std::tolower(b); }); }
static std::string dir_of(const std::string &p) { size_t i = p.find_last_of('/'); return i == std::string::npos ? "." : p.substr(0, i); }
static std::string join_path(const std::string &dir, const std::string &fn) { if (!fn.empty() && fn[0] == '/') return fn; return dir + "/" + fn; }
static bool file_exists(const std::string &p) { std::ifstream f(p, std::ios::binary); return (bool) f; }

// Radiance .hdr (RGBE, new-style RLE) — native decoder
static bool load_hdr(const std::string &path, Image &img) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line; bool fmt = false;
    std::getline(f, line);
    if (line.rfind("#?", 0) != 0) return false;
    while (std::getline(f, line) && !line.empty()) if (line.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) fmt = true;
    (void) fmt;
    std::getline(f, line);
    int w = 0, h = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2) return false;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536) return false; // malformed header: no allocation from it
    img.w = w; img.h = h; img.channels = 3; img.is_float = true; img.f32.assign((size_t) w * h * 4, 1.f);
    std::vector<uint8_t> scan((size_t) w * 4);
    for (int y = 0; y < h; ++y) {
        uint8_t hd[4]; f.read((char *) hd, 4);
        if (!f) return false;
        if (hd[0] == 2 && hd[1] == 2 && ((hd[2] << 8) | hd[3]) == w && w >= 8 && w < 32768) {
            for (int c = 0; c < 4; ++c) {
                int x = 0;
                while (x < w) {
                    uint8_t n; f.read((char *) &n, 1);
                    if (!f || n == 0) return false; // truncated file / zero-length run: the loop would never advance
                    if (n > 128) { uint8_t v; f.read((char *) &v, 1); if (!f) return false; n -= 128; while (n-- && x < w) scan[(size_t) x++ * 4 + c] = v; }
                    else { while (n-- && x < w) { uint8_t v; f.read((char *) &v, 1); if (!f) return false; scan[(size_t) x++ * 4 + c] = v; } }
                }
            }
        } else {
            std::memcpy(scan.data(), hd, 4);
            f.read((char *) scan.data() + 4, (std::streamsize) ((size_t) w * 4 - 4));
            if (!f) return false;
        }
        for (int x = 0; x < w; ++x) {
            const uint8_t *p = &scan[(size_t) x * 4];
            float sc = p[3] ? std::ldexp(1.f, (int) p[3] - 136) : 0.f;
            float *o = &img.f32[((size_t) y * w + x) * 4];
            o[0] = p[0] * sc; o[1] = p[1] * sc; o[2] = p[2] * sc; o[3] = 1.f;
        }
    }
    return true;
}

// Seeded procedural environment used when the scene's HDRI is stripped from the reference checkout
// (classroom: spaichingen_hill_2k.exr).  Closed form, no RNG: clear-sky gradient + sun disc +
// ground, lat-long 2048x1024, row 0 = zenith.  Documented in DESIGN.md §"stand-in assets".
static void procedural_sky(Image &img) {
    const uint32_t W = 2048, H = 1024;
    img.w = W; img.h = H; img.channels = 3; img.is_float = true; img.f32.resize((size_t) W * H * 4);
    const double sun_el = radians(32.0), sun_az = radians(250.0);
    const double sx = std::cos(sun_el) * std::cos(sun_az), sy = std::cos(sun_el) * std::sin(sun_az), sz = std::sin(sun_el);
    for (uint32_t y = 0; y < H; ++y) {
        double theta = (y + 0.5) / H * M_PI;
        double cz = std::cos(theta), sn = std::sin(theta);
        for (uint32_t x = 0; x < W; ++x) {
            double phi = (x + 0.5) / W * 2.0 * M_PI;
            double dx = sn * std::cos(phi), dy = sn * std::sin(phi), dz = cz;
            double r, g, b;
            if (dz >= 0) {
                double t = std::pow(1.0 - dz, 3.0);
                r = 0.0022 * (1 - t) + 0.0085 * t; g = 0.0040 * (1 - t) + 0.0090 * t; b = 0.0085 * (1 - t) + 0.0095 * t;
            } else { r = 0.0016; g = 0.0014; b = 0.0011; }
            double cs = dx * sx + dy * sy + dz * sz;
            double ang = std::acos(std::min(1.0, std::max(-1.0, cs)));
            double sun = 6.0 * std::exp(-(ang * ang) / (2 * 0.02 * 0.02)) + 0.02 * std::exp(-(ang * ang) / (2 * 0.25 * 0.25));
            r += sun * 1.0; g += sun * 0.93; b += sun * 0.80;
            float *o = &img.f32[((size_t) y * W + x) * 4];
            o[0] = (float) r; o[1] = (float) g; o[2] = (float) b; o[3] = 1.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// scene builder
// ------------------------------------------------------------------------------------------------
struct MetalIor { const char *name; float eta[3]; float k[3]; };
struct MetalSpd { std::string name; std::vector<float> eta, k; };
static const MetalIor kMetals[] = {
#include "metal_ior_rgb.inl"
};

struct AliasBuild { std::vector<float> prob; std::vector<uint32_t> alias; std::vector<float> func; float integral{0}; };
// AliasTable::build (render_core/warper/alias.h:86-122)
static AliasBuild build_alias(std::vector<float> weights) {
    AliasBuild out;
    size_t n = weights.size();
    double sum = 0.0; for (float w : weights) sum += (double) w;
    double ratio = (double) n / sum;
    std::vector<uint32_t> over, under;
    out.prob.resize(n); out.alias.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        float p = (float) ((double) weights[i] * ratio);
        out.prob[i] = p; out.alias[i] = i;
        (p > 1.0f ? over : under).push_back(i);
    }
    while (!over.empty() && !under.empty()) {
        uint32_t o = over.back(), u = under.back();
        over.pop_back(); under.pop_back();
        out.prob[o] -= 1.0f - out.prob[u];
        out.alias[u] = o;
        if (out.prob[o] > 1.0f) over.push_back(o);
        else if (out.prob[o] < 1.0f) under.push_back(o);
    }
    for (uint32_t i : over) { out.prob[i] = 1.0f; out.alias[i] = i; }
    for (uint32_t i : under) { out.prob[i] = 1.0f; out.alias[i] = i; }
    out.integral = (float) (sum / (double) n);
    out.func = std::move(weights);
    return out;
}

struct SlotSpec { int dim; std::vector<float> def; };

struct HostScene {
    std::string scene_dir;
    vmk_host_options opt{};
    // tables
    std::vector<vmk_tri_pos> tri_pos;
    std::vector<vmk_tri_attr> tri_attr;
    std::vector<vmk_instance> instances;
    std::vector<vmk_material> materials;
    std::vector<std::string> material_names;
    std::vector<vmk_medium> mediums;
    std::vector<std::string> medium_names;
    std::vector<vmk_light> lights;
    std::vector<vmk_texture> textures;
    std::vector<uint8_t> tex_data;
    std::map<std::string, uint32_t> tex_index;
    std::vector<float> alias_prob, alias_func;
    std::vector<uint32_t> alias_idx;
    std::vector<float> luts;
    // hero spectrum (render_core/spectrum/hero.cpp): tabulated spectra pool, CIE tables, metal (eta, k) curves, sRGB uplift table
    bool hero{false};
    uint32_t spectrum_dimension{3}; // wavelengths per path of spectrum/hero (3 or 4)
    std::string data_dir;            // directory of the albedo-table blob: spectra.bin / srgb2spec.bin live next to it
    std::vector<float> spd;          // vmk_scene.spd_data
    std::vector<float> cie_raw;      // X, Y, Z, D65 at 1 nm (4 x 471)
    std::vector<MetalSpd> metal_spd;
    std::vector<float> rgb2spec;
    vmk_scene scene{};
    vmk_render_params params{};
    uint32_t output_spp{0};
    std::string output_fn;
    std::string description;
    std::vector<std::string> image_paths;
    bool list_only{false};
    double bmin[3] = {1e300, 1e300, 1e300}, bmax[3] = {-1e300, -1e300, -1e300};

    void describe(const std::string &cat, const std::string &type, const std::string &name) { description += cat + "/" + type + " " + name + "\n"; }

    // ---- textures (ImagePool::load_texture image_pool.cpp:13-35) ----
    uint32_t obtain_texture(const std::string &fn, std::string color_space, bool allow_procedural) {
        std::string path = join_path(scene_dir, fn);
        if (color_space.empty()) color_space = (ends_with(fn, ".exr") || ends_with(fn, ".hdr")) ? "linear" : "srgb";
        std::string key = path + "|" + color_space;
        auto it = tex_index.find(key);
        if (it != tex_index.end()) return it->second;
        image_paths.push_back(path);
        if (list_only) { tex_index[key] = 0; return 0; }
        Image local;
        const Image *img = nullptr;
        auto reg = g_images.find(path);
        if (reg != g_images.end()) img = &reg->second;
        else if (ends_with(fn, ".hdr") && load_hdr(path, local)) img = &local;
        else if (ends_with(fn, ".exr") && file_exists(path)) { // native OpenEXR (exr.h): scanline, NONE / RLE / ZIPS / ZIP / PIZ, half / float
            std::ifstream fi(path, std::ios::binary);
            std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
            vmk_exr::ImageF dec = vmk_exr::decode(bytes);
            if (!dec.error.empty()) fail("image '" + path + "': " + dec.error + " (decode it in the caller and hand the pixels to vmk_host_register_image)");
            local.w = dec.w; local.h = dec.h; local.channels = dec.channels; local.is_float = true; local.f32.swap(dec.px);
            img = &local;
        }
        else if ((ends_with(fn, ".png") || ends_with(fn, ".jpg") || ends_with(fn, ".jpeg")) && file_exists(path)) { // native decode (image_codec.h)
            std::ifstream fi(path, std::ios::binary);
            std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
            vmk_img::Decoded dec = ends_with(fn, ".png") ? vmk_img::decode_png(bytes) : vmk_img::decode_jpeg(bytes);
            if (!dec.error.empty()) fail("image '" + path + "': " + dec.error + " (decode it in the caller and hand the pixels to vmk_host_register_image)");
            local.w = dec.w; local.h = dec.h; local.channels = dec.channels; local.is_float = false; local.u8.swap(dec.px);
            img = &local;
        }
        else if (allow_procedural && opt.procedural_env) { procedural_sky(local); img = &local; describe("image", "procedural_sky", fn + " (stand-in: file missing)"); }
        else if (opt.missing_assets == 1 && !file_exists(path)) { // declared stand-in for an asset stripped from the checkout: 1x1 mid-grey
            local.w = local.h = 1; local.channels = 3; local.is_float = false; local.u8 = {128, 128, 128, 255};
            img = &local; describe("image", "constant_grey", fn + " (stand-in: file missing)");
        }
        else fail("image '" + path + "' is neither registered (vmk_host_register_image) nor decodable natively" + (file_exists(path) ? "" : " (file missing)"));
        vmk_texture t{};
        while (tex_data.size() % 16) tex_data.push_back(0);
        t.offset = tex_data.size(); t.width = img->w; t.height = img->h; t.channels = img->channels;
        size_t n = (size_t) img->w * img->h;
        if (img->is_float) {
            t.format = VMK_TEX_RGBA32F;
            size_t off = tex_data.size(); tex_data.resize(off + n * 16);
            float *dst = (float *) (tex_data.data() + off);
            if (img->f32.size() == n * 4) std::memcpy(dst, img->f32.data(), n * 16);
            else for (size_t i = 0; i < n; ++i) for (int c = 0; c < 4; ++c) dst[i * 4 + c] = c < (int) img->channels ? img->f32[i * img->channels + c] : (c == 3 ? 1.f : (img->channels == 1 ? img->f32[i] : 0.f));
        } else {
            t.format = color_space == "linear" ? VMK_TEX_RGBA8_LINEAR : VMK_TEX_RGBA8_SRGB;
            size_t off = tex_data.size(); tex_data.resize(off + n * 4);
            uint8_t *dst = tex_data.data() + off;
            for (size_t i = 0; i < n; ++i) for (int c = 0; c < 4; ++c) dst[i * 4 + c] = c < (int) img->channels ? img->u8[i * img->channels + c] : (c == 3 ? 255 : (img->channels == 1 ? img->u8[i] : 0));
        }
        uint32_t id = (uint32_t) textures.size();
        textures.push_back(t);
        tex_index[key] = id;
        return id;
    }

    // ---- tabulated spectra ("spd" shader node, SPD::init spd.cpp:50-53) ----
    vmk_slot add_spd(const std::vector<float> &func) {
        vmk_slot sl{};
        uint32_t off = (uint32_t) spd.size(), n = (uint32_t) func.size();
        spd.insert(spd.end(), func.begin(), func.end());
        std::memcpy(&sl.v[0], &off, 4); std::memcpy(&sl.v[1], &n, 4);
        sl.v[2] = 471.f / (float) n; // static_cast<float>(cie_sample_count) / func.size()
        sl.tex = VMK_SLOT_SPD;
        return sl;
    }
    void load_spectra() {
        std::string path = join_path(data_dir, "spectra.bin");
        std::ifstream f(path, std::ios::binary);
        if (!f) fail("spectrum/hero: cannot open '" + path + "' (CIE observer, D65 and metal curves; tools/make_spectra.py)");
        uint32_t hdr[3];
        f.read((char *) hdr, sizeof(hdr));
        if (!f || hdr[0] != 0x44505356u || hdr[1] != 1u || hdr[2] != 471u) fail("bad spectra blob '" + path + "'");
        cie_raw.resize(4 * 471);
        f.read((char *) cie_raw.data(), (std::streamsize) (cie_raw.size() * 4));
        uint32_t n_metals = 0;
        f.read((char *) &n_metals, 4);
        if (!f || n_metals > 64) fail("spectra blob truncated");
        for (uint32_t i = 0; i < n_metals; ++i) {
            char name[17] = {0}; uint32_t n = 0;
            f.read(name, 16); f.read((char *) &n, 4);
            if (!f || n < 2 || n > 4096) fail("spectra blob: bad metal record");
            MetalSpd m; m.name = name; m.eta.resize(n); m.k.resize(n);
            f.read((char *) m.eta.data(), (std::streamsize) n * 4); f.read((char *) m.k.data(), (std::streamsize) n * 4);
            if (!f) fail("spectra blob truncated");
            metal_spd.push_back(std::move(m));
        }
    }
    void load_rgb2spec() {
        std::string path = join_path(data_dir, "srgb2spec.bin");
        std::ifstream f(path, std::ios::binary);
        if (!f) fail("spectrum/hero: cannot open '" + path + "' (sRGB uplift table; python __graft_entry__.py builds it with vmk_host_build_rgb2spec)");
        uint32_t hdr[3];
        f.read((char *) hdr, sizeof(hdr));
        const size_t n = (size_t) 3 * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * 4;
        if (!f || hdr[0] != 0x53325256u || hdr[1] != 1u || hdr[2] != VMK_RGB2SPEC_RES) fail("bad sRGB uplift table '" + path + "'");
        rgb2spec.resize(n);
        f.read((char *) rgb2spec.data(), (std::streamsize) (n * 4));
        if (!f) fail("sRGB uplift table truncated");
    }
    // HeroWavelengthSpectrum ctor (hero.cpp:243-252): the CIE tables downsampled to every 5th 1 nm sample (spd.cpp:27-33,95-109)
    void init_hero() {
        hero = true;
        load_spectra();
        load_rgb2spec();
        const uint32_t interval = 5;
        const float factor = (830.0f - 360.0f) / (float) interval;
        const uint32_t n = (uint32_t) std::ceil(factor);
        for (int t = 0; t < 4; ++t) {
            std::vector<float> samples(n);
            for (uint32_t x = 0; x < n; ++x) samples[x] = cie_raw[(size_t) t * 471 + x * interval];
            vmk_slot sl = add_spd(samples);
            std::memcpy(&scene.spd_cie[t], &sl.v[0], 4);
            scene.spd_cie_count = n; scene.spd_cie_interval = sl.v[2];
        }
        { // densely_sampled_spectrum_integral(5, cie::Y) spd.cpp:15-25
            const float *Y = cie_raw.data() + 471;
            float sum = 0.0f, tt = (float) interval;
            float nf = (830.0f - 360.0f) / tt;
            uint32_t nn = (uint32_t) nf + 1u;
            for (uint32_t i = 0; i < nn - 1u; ++i) sum += 0.5f * (Y[i * interval] + Y[(i + 1u) * interval]);
            scene.cie_y_integral = sum * tt;
        }
    }

    // ---- slots: SlotDesc::init / ShaderNodeDesc::init (node_desc.cpp:128-151,307-335), swizzle shader_node.cpp:242-273 ----
    static uint32_t channel_mask(const std::string &channels, int dim) {
        uint32_t sw = 0;
        for (int i = 0; i < 3; ++i) {
            char ch = i < (int) channels.size() ? (char) std::tolower(channels[i]) : (dim == 1 ? channels[0] : "xyz"[i]);
            uint32_t idx = (ch == 'x' || ch == 'r') ? 0 : (ch == 'y' || ch == 'g') ? 1 : (ch == 'z' || ch == 'b') ? 2 : 3;
            sw |= idx << (2 * i);
        }
        return sw;
    }
    vmk_slot parse_slot(const Json &param, const std::string &key, int dim, std::vector<float> def, bool env_image = false) {
        return parse_slot_json(param[key], key, dim, def, env_image);
    }
    // a slot description: a bare value / array, {"channels", "node"}, or a node object (number / constant / image / multiply)
    // The TOPOLOGY of the slot parsed last, as a string: what ShaderNodeSlot::compute_topology_hash (shader_node.cpp:187-189) hashes —
    // channel mask, dimension, the node's own topology: a number node's is (value count, min, max) (number.cpp:18), an image node's
    // the reduction over its inner slots, which no scene file can change (image.cpp:14-36: the file name and the scale are values, not
    // topology), a binary node's that of its operands (math.cpp:71-73).  LightSampler::tidy_up groups the lights by it.
    std::string slot_topology;
    vmk_slot parse_slot_json(const Json &ps, const std::string &key, int dim, std::vector<float> def, bool env_image = false) {
        vmk_slot sl{}; sl.tex = VMK_INVALID;
        std::string channels = dim == 1 ? "x" : "xyz";
        const Json *node = &ps;
        if (ps.is_object() && ps.contains("channels")) { channels = ps["channels"].as_string(channels); node = &ps["node"]; }
        std::vector<float> value = def;
        std::string fn, color_space; float tex_scale = 1.f; bool is_image = false;
        float num_min = 0.f, num_max = 1.f; // NumberArray's range (number.cpp:24-25; ShaderNodeDesc::init node_desc.cpp:322-327)
        std::string node_kind = "number";
        if (node->is_null()) { node_kind = "none"; }
        else if (node->is_array()) value = node->as_float_vector();
        else if (node->is_number()) value = {node->as_float(0.f)};
        else if (node->is_object()) {
            const Json *p = node;
            std::string type;
            if (!node->contains("param")) type = (*node)["type"].as_string("image");
            else { type = (*node)["type"].as_string(); p = &(*node)["param"]; }
            if (type == "image") { is_image = true; fn = (*p)["fn"].as_string(); color_space = (*p)["color_space"].as_string(); tex_scale = (*p)["scale"].as_float(1.f); }
            else if (type == "number" || type == "constant") {
                const Json &v = (*p)["value"]; if (!v.is_null()) value = v.as_float_vector();
                if (type == "number") { num_min = (*p)["min"].as_float(0.f); num_max = (*p)["max"].as_float(1.f); }
                node_kind = type;
            }
            else if (type == "multiply") {
                vmk_slot r = parse_multiply(*p, key, channels, dim, env_image);
                slot_topology = "slot(" + std::to_string(channel_mask(channels, dim)) + "," + std::to_string(dim) + "," + slot_topology + ")";
                return r;
            }
            else fail("shader node type '" + type + "' (slot '" + key + "') is outside the hot-path scope (number / constant / image / multiply of an image and a constant)");
        }
        if (is_image) {
            uint32_t id = obtain_texture(fn, color_space, env_image);
            sl.v[0] = tex_scale; sl.v[1] = sl.v[2] = 0.f;
            sl.tex = (id & 0xffffu) | (channel_mask(channels, dim) << 16);
            slot_topology = "slot(" + std::to_string(channel_mask(channels, dim)) + "," + std::to_string(dim) + ",image)";
            return sl;
        }
        if (value.empty()) value = def;
        {   // (a scalar given for a 3-channel slot is replicated BEFORE the node is built, node_desc.cpp:137-143, so its value count is dim)
            size_t count = (dim > 1 && value.size() == 1 && !node->is_array()) ? (size_t) dim : value.size();
            char buf[96]; snprintf(buf, sizeof buf, "%s(%zu,%g,%g)", node_kind.c_str(), count, (double) num_min, (double) num_max);
            slot_topology = "slot(" + std::to_string(channel_mask(channels, dim)) + "," + std::to_string(dim) + "," + buf + ")";
        }
        if ((int) channels.size() > 1 && value.size() == 1) value = std::vector<float>(channels.size(), value[0]); // scalar broadcast
        float c[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < value.size() && i < 4; ++i) c[i] = value[i];
        uint32_t sw = channel_mask(channels, dim);
        sl.v[0] = c[sw & 3u]; sl.v[1] = dim == 1 ? 0.f : c[(sw >> 2) & 3u]; sl.v[2] = dim == 1 ? 0.f : c[(sw >> 4) & 3u];
        return sl;
    }
    // "multiply" (render_core/shadernode/math.cpp:34-92: BinaryOpNode — its `op_` is never set, so every binary node multiplies):
    // lhs * rhs per channel, then the outer slot's channel selection.  Encoded for constant x constant (folded here, in float32 like the
    // kernel would) and image x constant (VMK_SLOT_TINTED: the constant rides in the slot's v[0..2]); the image's own "scale" must be 1
    // for the latter (the slot has no room for both, and (t * s) * c is not t * (s * c) in float32).
    vmk_slot parse_multiply(const Json &p, const std::string &key, const std::string &channels, int dim, bool env_image) {
        vmk_slot l = parse_slot_json(p["lhs"], key + ".lhs", 3, {1.f, 1.f, 1.f}, env_image);
        const std::string topo_l = slot_topology;
        vmk_slot r = parse_slot_json(p["rhs"], key + ".rhs", 3, {1.f, 1.f, 1.f}, env_image);
        slot_topology = "multiply(" + topo_l + "," + slot_topology + ")";
        const bool li = l.tex != VMK_INVALID, ri = r.tex != VMK_INVALID;
        if (li && ri) fail("slot '" + key + "': a multiply node of two images is outside the hot-path scope (image x constant, constant x constant)");
        if ((li && (l.tex & VMK_SLOT_TINTED)) || (ri && (r.tex & VMK_SLOT_TINTED))) fail("slot '" + key + "': nested multiply nodes over an image are outside the hot-path scope");
        const uint32_t sw = channel_mask(channels, dim); // output channel k reads operand channel (sw >> 2k) & 3
        const int n_out = dim == 1 ? 1 : 3;
        vmk_slot out{}; out.tex = VMK_INVALID;
        uint32_t tex_sw = 0;
        for (int k = 0; k < 3; ++k) {
            const uint32_t j = k < n_out ? (sw >> (2 * k)) & 3u : 0u;
            if (k < n_out && j > 2u) fail("slot '" + key + "': channel 'w' of a multiply node is outside the hot-path scope");
            if (!li && !ri) out.v[k] = k < n_out ? l.v[j] * r.v[j] : 0.f;
            else {
                const vmk_slot &img = li ? l : r, &c = li ? r : l;
                if (img.v[0] != 1.f) fail("slot '" + key + "': an image with a scale other than 1 inside a multiply node is outside the hot-path scope");
                out.v[k] = k < n_out ? c.v[j] : 0.f;
                tex_sw |= (((img.tex >> 16) >> (2 * j)) & 3u) << (2 * k);
            }
        }
        if (li || ri) out.tex = ((li ? l.tex : r.tex) & 0xffffu) | (tex_sw << 16) | VMK_SLOT_TINTED;
        return out;
    }

    // ---- materials (Scene::load_materials scene.cpp:111-118; initialize_slots of each plugin) ----
    uint32_t add_material(const Json &desc) {
        std::string type = desc["type"].as_string("diffuse");
        std::string name = desc["name"].as_string();
        const Json &p = desc["param"];
        vmk_material m{};
        for (auto &s : m.slot) s.tex = VMK_INVALID;
        m.child0 = m.child1 = VMK_INVALID;
        m.normal.tex = VMK_INVALID;
        if (p.contains("normal")) { // Material::initialize_slots material.cpp:312-316: VS_INIT_SLOT_NO_DEFAULT(normal, Number)
            if (type == "mix" || type == "add") fail("material '" + name + "': a normal slot on mix / add has no effect in the reference (only its children build lobes) and is not accepted");
            m.normal = parse_slot(p, "normal", 3, {0.f, 0.f, 1.f});
            if (m.normal.tex != VMK_INVALID && (m.normal.tex & VMK_SLOT_TINTED)) fail("material '" + name + "': a multiply node in the normal slot is outside the hot-path scope");
            m.flags |= VMK_MATF_HAS_NORMAL;
        }
        bool remap = p["remapping_roughness"].as_bool(true);
        if (remap) m.flags |= VMK_MATF_REMAP_ROUGHNESS;
        if (type == "diffuse") { // diffuse.cpp:41-48
            m.type = VMK_MAT_DIFFUSE;
            m.slot[0] = parse_slot(p, "color", 3, {0.5f, 0.5f, 0.5f});
            if (p.contains("sigma")) { m.flags |= VMK_MATF_HAS_SIGMA; m.slot[1] = parse_slot(p, "sigma", 1, {0.5f}); }
        } else if (type == "mirror") { // mirror.cpp:35-41
            m.type = VMK_MAT_MIRROR;
            m.slot[0] = parse_slot(p, "color", 3, {1, 1, 1}); m.slot[1] = parse_slot(p, "roughness", 1, {0.001f}); m.slot[2] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "metal") { // metal.cpp:59-65,104-129 (srgb: SPD evaluated at the 3 peak wavelengths on the host)
            m.type = VMK_MAT_METAL;
            std::string mname = p["material_name"].as_string();
            const MetalIor *mi = &kMetals[0]; // names[0] when not found (sorted map order in the reference: "Ag")
            for (auto &k : kMetals) if (mname == k.name) mi = &k;
            m.slot[0].v[0] = mi->eta[0]; m.slot[0].v[1] = mi->eta[1]; m.slot[0].v[2] = mi->eta[2];
            m.slot[1].v[0] = mi->k[0]; m.slot[1].v[1] = mi->k[1]; m.slot[1].v[2] = mi->k[2];
            if (hero) { // is_complete(): the measured curves travel as "spd" nodes (metal.cpp:113-117)
                const MetalSpd *ms = &metal_spd.at(0);
                for (auto &k : metal_spd) if (k.name == "Ag") ms = &k; // names[0] of the sorted map when the name is unknown
                for (auto &k : metal_spd) if (mname == k.name) ms = &k;
                m.slot[0] = add_spd(ms->eta); m.slot[1] = add_spd(ms->k);
            }
            m.slot[2] = parse_slot(p, "roughness", 1, {0.01f}); m.slot[3] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "glass") { // glass.cpp:189-196,216-233
            m.type = VMK_MAT_GLASS;
            m.slot[0] = parse_slot(p, "color", 3, {1, 1, 1});
            std::string gname = p["material_name"].as_string();
            if (gname.empty()) m.slot[1] = parse_slot(p, "ior", 1, {1.5f});
            else { // Sellmeier at rgb_spectrum_peak_wavelengths.x (glass.cpp:104-134,226-228)
                auto sellmeier = [&](float lambda_nm) {
                    float lambda = lambda_nm / 1000.f; float l2 = lambda * lambda, f;
                    if (gname == "LASF9") f = 2.00029547f * l2 / (l2 - 0.0121426017f) + 0.298926886f * l2 / (l2 - 0.0538736236f) + 1.80691843f * l2 / (l2 - 156.530829f);
                    else f = 1.03961212f * l2 / (l2 - 0.00600069867f) + 0.231792344f * l2 / (l2 - 0.0200179144f) + 1.01046945f * l2 / (l2 - 103.560653f);
                    return std::sqrt(f + 1.f);
                };
                m.slot[1].v[0] = sellmeier(602.785f);
                if (hero) { // is_complete(): the curve tabulated by SPD::to_list (spd.h:41-47), is_dispersive() (glass.cpp:221-224,234)
                    std::vector<float> lst;
                    for (float lambda = 360.0f; lambda < 830.0f; lambda += 5.f) lst.push_back(sellmeier(lambda));
                    m.slot[1] = add_spd(lst);
                    m.flags |= VMK_MATF_DISPERSIVE;
                }
            }
            m.slot[2] = parse_slot(p, "roughness", 1, {0.5f}); m.slot[3] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "substrate") { // substrate.cpp:117-124
            m.type = VMK_MAT_SUBSTRATE;
            m.slot[0] = parse_slot(p, "color", 3, {1, 1, 1}); m.slot[1] = parse_slot(p, "spec", 3, {0.05f, 0.05f, 0.05f});
            m.slot[2] = parse_slot(p, "roughness", 1, {0.5f}); m.slot[3] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "principled_bsdf") { // principled_bsdf.cpp:281-307
            m.type = VMK_MAT_PRINCIPLED;
            m.slot[VMK_P_COLOR] = parse_slot(p, "color", 3, {1, 1, 1}); m.slot[VMK_P_METALLIC] = parse_slot(p, "metallic", 1, {0.f});
            m.slot[VMK_P_IOR] = parse_slot(p, "ior", 1, {1.5f}); m.slot[VMK_P_ROUGHNESS] = parse_slot(p, "roughness", 1, {0.5f});
            m.slot[VMK_P_SPEC_TINT] = parse_slot(p, "spec_tint", 3, {1, 1, 1}); m.slot[VMK_P_ANISOTROPIC] = parse_slot(p, "anisotropic", 1, {0.f});
            m.slot[VMK_P_OPACITY] = parse_slot(p, "opcacity", 1, {1.f});
            m.slot[VMK_P_SHEEN_WEIGHT] = parse_slot(p, "sheen_weight", 1, {0.f}); m.slot[VMK_P_SHEEN_ROUGHNESS] = parse_slot(p, "sheen_roughness", 1, {0.5f});
            m.slot[VMK_P_SHEEN_TINT] = parse_slot(p, "sheen_tint", 3, {1, 1, 1});
            m.slot[VMK_P_COAT_WEIGHT] = parse_slot(p, "coat_weight", 1, {0.f}); m.slot[VMK_P_COAT_ROUGHNESS] = parse_slot(p, "coat_roughness", 1, {0.2f});
            m.slot[VMK_P_COAT_IOR] = parse_slot(p, "coat_ior", 1, {1.5f}); m.slot[VMK_P_COAT_TINT] = parse_slot(p, "coat_tint", 3, {1, 1, 1});
            m.slot[VMK_P_SSS_WEIGHT] = parse_slot(p, "subsurface_weight", 1, {0.3f}); m.slot[VMK_P_SSS_RADIUS] = parse_slot(p, "subsurface_radius", 3, {1, 1, 1});
            m.slot[VMK_P_SSS_SCALE] = parse_slot(p, "subsurface_scale", 1, {0.2f}); m.slot[VMK_P_TRANS_WEIGHT] = parse_slot(p, "transmission_weight", 1, {0.f});
        } else if (type == "plastic") { // plastic.cpp:88-96
            m.type = VMK_MAT_PLASTIC;
            m.slot[0] = parse_slot(p, "color", 3, {1, 1, 1}); m.slot[1] = parse_slot(p, "spec", 3, {0.05f, 0.05f, 0.05f});
            m.slot[2] = parse_slot(p, "ior", 1, {1.3f});
            m.slot[3] = parse_slot(p, "roughness", 1, {0.5f}); m.slot[4] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "metallic") { // metallic.cpp:24-34
            m.type = VMK_MAT_METALLIC;
            m.slot[0] = parse_slot(p, "color", 3, {1, 1, 1}); m.slot[1] = parse_slot(p, "edge_tint", 3, {1, 1, 1});
            m.slot[2] = parse_slot(p, "roughness", 1, {0.5f}); m.slot[3] = parse_slot(p, "anisotropic", 1, {0.f});
        } else if (type == "mix" || type == "add") { // mix.cpp:27-41, add.cpp:17-27
            m.type = type == "mix" ? VMK_MAT_MIX : VMK_MAT_ADD;
            uint32_t c0 = add_material(p["mat0"]), c1 = add_material(p["mat1"]);
            // LobeSet::flatten (lobe.cpp:534-562) merges a lobe-set child's lobes into the parent's list: one principled_bsdf child next to a
            // single-lobe child is carried (the reference's own cbox/cbox.json); two lobe-set children, or nested mix / add, are not
            const uint32_t t0 = materials[c0].type, t1 = materials[c1].type;
            const bool ok0 = VMK_MAT_IS_SINGLE_LOBE(t0) || t0 == VMK_MAT_PRINCIPLED, ok1 = VMK_MAT_IS_SINGLE_LOBE(t1) || t1 == VMK_MAT_PRINCIPLED;
            if (!ok0 || !ok1 || (t0 == VMK_MAT_PRINCIPLED && t1 == VMK_MAT_PRINCIPLED)) fail("material '" + name + "': " + type + " of two lobe-set materials (principled_bsdf / mix / add) is outside the hot-path scope (one principled_bsdf child next to a single-lobe child is supported)");
            if ((materials[c0].flags | materials[c1].flags) & VMK_MATF_HAS_NORMAL) fail("material '" + name + "': normal maps on the children of " + type + " are outside the hot-path scope");
            m.child0 = c0; m.child1 = c1;
            if (type == "mix") m.slot[0] = parse_slot(p, "frac", 1, {0.5f});
        } else fail("material type '" + type + "' (" + name + ") is outside the hot-path scope (SURVEY.md §2)");
        describe("material", type, name);
        materials.push_back(m);
        material_names.push_back(name);
        return (uint32_t) materials.size() - 1;
    }

    // ---- meshes ----
    struct Vtx { float p[3], n[3], uv[2]; };
    struct Mesh { std::vector<Vtx> v; std::vector<uint32_t> idx; };

    static Mesh make_quad(const Json &p) { // quad.cpp:21-50
        Mesh m; float w = p["width"].as_float(1.f) / 2, h = p["height"].as_float(1.f) / 2;
        float P[4][3] = {{w, 0, h}, {w, 0, -h}, {-w, 0, h}, {-w, 0, -h}};
        float UV[4][2] = {{1, 1}, {1, 0}, {0, 1}, {0, 0}};
        float dp02[3] = {P[0][0] - P[2][0], P[0][1] - P[2][1], P[0][2] - P[2][2]}, dp12[3] = {P[1][0] - P[2][0], P[1][1] - P[2][1], P[1][2] - P[2][2]};
        float ng[3] = {dp02[1] * dp12[2] - dp02[2] * dp12[1], dp02[2] * dp12[0] - dp02[0] * dp12[2], dp02[0] * dp12[1] - dp02[1] * dp12[0]};
        for (int i = 0; i < 4; ++i) { Vtx v{}; std::memcpy(v.p, P[i], 12); std::memcpy(v.n, ng, 12); std::memcpy(v.uv, UV[i], 8); m.v.push_back(v); }
        m.idx = {0, 1, 2, 2, 1, 3};
        return m;
    }
    static Mesh make_sphere(const Json &p) { // sphere.cpp:20-88: latitude / longitude tessellation, same vertex and triangle order
        Mesh m;
        const float radius = p["radius"].as_float(1.f);
        const uint32_t theta_div = std::max(2u, p["sub_div"].as_uint(60u)), phi_div = 2 * theta_div;
        const float Pi = 3.14159265358979323846f, TwoPi = 6.28318530717958647692f;
        auto push = [&](float x, float y, float z, float u, float v) {
            Vtx vt{}; vt.p[0] = x; vt.p[1] = y; vt.p[2] = z; vt.uv[0] = u; vt.uv[1] = v;
            float l = std::sqrt(x * x + y * y + z * z);
            vt.n[0] = x / l; vt.n[1] = y / l; vt.n[2] = z / l;
            m.v.push_back(vt);
        };
        push(0.f, radius, 0.f, 0.f, 0.f);
        for (uint32_t i = 1; i < theta_div; ++i) {
            float v = float(i) / theta_div, theta = Pi * v, y = radius * std::cos(theta), r = radius * std::sin(theta);
            push(r, y, 0.f, 0.f, v);
            for (uint32_t j = 1; j < phi_div; ++j) {
                float u = float(j) / phi_div, phi = u * TwoPi;
                push(std::cos(phi) * r, y, std::sin(phi) * r, u, v);
            }
        }
        push(0.f, -radius, 0.f, 0.f, 1.f);
        auto tri = [&](uint32_t a, uint32_t b, uint32_t c) { m.idx.push_back(a); m.idx.push_back(b); m.idx.push_back(c); };
        for (uint32_t i = 0; i < phi_div; ++i) tri(0, (i + 1) % phi_div + 1, i + 1);
        for (uint32_t i = 0; i + 2 < theta_div; ++i) {
            uint32_t vs = 1 + i * phi_div;
            for (uint32_t j = 0; j < phi_div; ++j, ++vs) {
                if (j != phi_div - 1) { tri(vs, vs + 1, vs + phi_div); tri(vs + 1, vs + phi_div + 1, vs + phi_div); }
                else { tri(vs, vs + 1 - phi_div, vs + phi_div); tri(vs + 1 - phi_div, vs + 1, vs + phi_div); }
            }
        }
        const uint32_t ve = (uint32_t) m.v.size() - 1;
        for (uint32_t i = 0; i < phi_div; ++i) tri(ve, ve - ((1 + i) % phi_div + 1), ve - (i + 1));
        return m;
    }
    static Mesh make_cube(const Json &p) { // cube.cpp:21-72
        float x = p["x"].as_float(1.f), y = p["y"].as_float(1.f), z = p["z"].as_float(1.f);
        y = y == 0 ? x : y; z = z == 0 ? y : z; x /= 2.f; y /= 2.f; z /= 2.f;
        const float P[24][3] = {{-x, -y, z}, {x, -y, z}, {-x, y, z}, {x, y, z}, {-x, y, -z}, {x, y, -z}, {-x, -y, -z}, {x, -y, -z},
                                {-x, y, z}, {x, y, z}, {-x, y, -z}, {x, y, -z}, {-x, -y, z}, {x, -y, z}, {-x, -y, -z}, {x, -y, -z},
                                {x, -y, z}, {x, y, z}, {x, y, -z}, {x, -y, -z}, {-x, -y, z}, {-x, y, z}, {-x, y, -z}, {-x, -y, -z}};
        const float N[6][3] = {{0, 0, 1}, {0, 0, -1}, {0, 1, 0}, {0, -1, 0}, {1, 0, 0}, {-1, 0, 0}};
        const float UV[24][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {0, 1}, {1, 1}, {0, 0}, {1, 0}, {0, 1}, {1, 1}, {0, 0}, {1, 0},
                                 {0, 1}, {1, 1}, {0, 0}, {1, 0}, {0, 1}, {1, 1}, {1, 0}, {0, 0}, {0, 1}, {1, 1}, {1, 0}, {0, 0}};
        Mesh m;
        for (int i = 0; i < 24; ++i) { Vtx v{}; std::memcpy(v.p, P[i], 12); std::memcpy(v.n, N[i / 4], 12); std::memcpy(v.uv, UV[i], 8); m.v.push_back(v); }
        m.idx = {0, 1, 3, 0, 3, 2, 6, 5, 7, 4, 5, 6, 10, 9, 11, 8, 9, 10, 13, 14, 15, 13, 12, 14, 18, 17, 19, 17, 16, 19, 21, 22, 23, 20, 21, 23};
        return m;
    }
    // Wavefront OBJ (model.cpp:23-36 -> AssimpParser, importers/assimp_parser.cpp:231-341): unique vertex per distinct
    // (v, vt, vn) triple in first-use order (aiProcess_JoinIdenticalVertices), polygons fan-triangulated
    // (assimp_parser.cpp:284-295), flip_uv (v -> 1-v, default true), flat normals when the file has none.
    Mesh load_obj(const std::string &path, bool flip_uv, bool smooth) {
        std::ifstream f(path);
        if (!f) fail("cannot open mesh '" + path + "'");
        std::vector<std::array<float, 3>> P, N; std::vector<std::array<float, 2>> T;
        Mesh m; std::map<std::array<int, 3>, uint32_t> uniq;
        std::string line;
        bool gen_normals = false;
        std::vector<std::array<int, 3>> poly;
        while (std::getline(f, line)) {
            const char *s = line.c_str();
            while (*s == ' ' || *s == '\t') ++s;
            if (s[0] == 'v' && s[1] == ' ') { std::array<float, 3> v{}; std::sscanf(s + 2, "%f %f %f", &v[0], &v[1], &v[2]); P.push_back(v); }
            else if (s[0] == 'v' && s[1] == 'n') { std::array<float, 3> v{}; std::sscanf(s + 3, "%f %f %f", &v[0], &v[1], &v[2]); N.push_back(v); }
            else if (s[0] == 'v' && s[1] == 't') { std::array<float, 2> v{}; std::sscanf(s + 3, "%f %f", &v[0], &v[1]); T.push_back(v); }
            else if (s[0] == 'f' && s[1] == ' ') {
                poly.clear();
                const char *q = s + 2;
                while (*q) {
                    while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
                    if (!*q) break;
                    int vi = 0, ti = 0, ni = 0; char *e;
                    if (*q == '#') break; // trailing comment
                    vi = (int) std::strtol(q, &e, 10);
                    if (e == q) fail("malformed face record in '" + path + "': '" + line + "'"); // strtol consumed nothing: no progress
                    q = e;
                    if (*q == '/') { ++q; if (*q != '/') { ti = (int) std::strtol(q, &e, 10); q = e; } if (*q == '/') { ++q; ni = (int) std::strtol(q, &e, 10); q = e; } }
                    if (vi < 0) vi = (int) P.size() + vi + 1;
                    if (ti < 0) ti = (int) T.size() + ti + 1;
                    if (ni < 0) ni = (int) N.size() + ni + 1;
                    poly.push_back({vi, ti, ni});
                }
                if (poly.size() < 3) continue;
                for (size_t k = 1; k + 1 < poly.size(); ++k) {
                    std::array<int, 3> tri[3] = {poly[0], poly[k], poly[k + 1]};
                    if (tri[0][0] == tri[1][0] || tri[1][0] == tri[2][0] || tri[0][0] == tri[2][0]) continue; // aiProcess_FindDegenerates
                    for (auto &c : tri) {
                        std::array<int, 3> key = c;
                        if (key[2] == 0) { gen_normals = true; key[2] = -(int) (m.idx.size() / 3) - 1; } // flat: unique per face
                        auto it = uniq.find(key);
                        uint32_t id;
                        if (it == uniq.end()) {
                            Vtx v{};
                            if (c[0] < 1 || c[0] > (int) P.size()) fail("bad vertex index in '" + path + "'");
                            std::memcpy(v.p, P[c[0] - 1].data(), 12);
                            if (c[2] >= 1 && c[2] <= (int) N.size()) std::memcpy(v.n, N[c[2] - 1].data(), 12);
                            if (c[1] >= 1 && c[1] <= (int) T.size()) { v.uv[0] = T[c[1] - 1][0]; v.uv[1] = flip_uv ? 1.f - T[c[1] - 1][1] : T[c[1] - 1][1]; }
                            for (float &x : v.n) if (!std::isfinite(x)) { v.n[0] = v.n[1] = v.n[2] = 0.f; break; }
                            for (float &x : v.uv) if (!std::isfinite(x)) { v.uv[0] = v.uv[1] = 0.f; break; }
                            id = (uint32_t) m.v.size(); m.v.push_back(v); uniq[key] = id;
                        } else id = it->second;
                        m.idx.push_back(id);
                    }
                }
            }
        }
        if (gen_normals) { // aiProcess_GenNormals
            (void) smooth;
            for (size_t t = 0; t + 2 < m.idx.size(); t += 3) {
                Vtx &a = m.v[m.idx[t]], &b = m.v[m.idx[t + 1]], &c = m.v[m.idx[t + 2]];
                float e1[3] = {b.p[0] - a.p[0], b.p[1] - a.p[1], b.p[2] - a.p[2]}, e2[3] = {c.p[0] - a.p[0], c.p[1] - a.p[1], c.p[2] - a.p[2]};
                float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (l > 0) for (float &x : n) x /= l;
                for (Vtx *v : {&a, &b, &c}) if (v->n[0] == 0 && v->n[1] == 0 && v->n[2] == 0) std::memcpy(v->n, n, 12);
            }
        }
        return m;
    }

    // float32 o2w.apply_point exactly as compute_surface_interaction evaluates it per hit (geometry.cpp:94-96)
    static void apply_point(const float *o2w, const float *p, float *out) {
        for (int r = 0; r < 3; ++r) out[r] = o2w[0 * 4 + r] * p[0] + o2w[1 * 4 + r] * p[1] + o2w[2 * 4 + r] * p[2] + o2w[3 * 4 + r];
    }
    uint32_t add_instance(const Mesh &mesh, const Mat4 &o2w, uint32_t mat_id, uint32_t inside_medium, uint32_t outside_medium) {
        vmk_instance inst{};
        inst.mat_id = mat_id; inst.light_id = VMK_INVALID;
        inst.inside_medium = inside_medium; inst.outside_medium = outside_medium;
        inst.tri_offset = (uint32_t) tri_pos.size(); inst.tri_count = (uint32_t) (mesh.idx.size() / 3);
        for (int i = 0; i < 16; ++i) inst.o2w[i] = (float) o2w.m[i];
        Mat4 inv = inverse(o2w); // normal matrix = transpose(inverse(M3x3)): n2w(row r, col c) = inv(c, r)
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) inst.n2w[c * 3 + r] = (float) inv.at(c, r);
        uint32_t inst_id = (uint32_t) instances.size();
        for (uint32_t t = 0; t < inst.tri_count; ++t) {
            const Vtx &a = mesh.v[mesh.idx[3 * t]], &b = mesh.v[mesh.idx[3 * t + 1]], &c = mesh.v[mesh.idx[3 * t + 2]];
            vmk_tri_pos tp{}; vmk_tri_attr ta{};
            apply_point(inst.o2w, a.p, tp.p0); apply_point(inst.o2w, b.p, tp.p1); apply_point(inst.o2w, c.p, tp.p2);
            tp.inst = inst_id; tp.prim = t;
            std::memcpy(ta.n0, a.n, 12); std::memcpy(ta.n1, b.n, 12); std::memcpy(ta.n2, c.n, 12);
            std::memcpy(ta.uv0, a.uv, 8); std::memcpy(ta.uv1, b.uv, 8); std::memcpy(ta.uv2, c.uv, 8);
            for (const float *pp : {tp.p0, tp.p1, tp.p2}) for (int k = 0; k < 3; ++k) { bmin[k] = std::min(bmin[k], (double) pp[k]); bmax[k] = std::max(bmax[k], (double) pp[k]); }
            tri_pos.push_back(tp); tri_attr.push_back(ta);
        }
        instances.push_back(inst);
        return inst_id;
    }
    uint32_t append_alias(const AliasBuild &a) {
        uint32_t off = (uint32_t) alias_prob.size();
        alias_prob.insert(alias_prob.end(), a.prob.begin(), a.prob.end());
        alias_idx.insert(alias_idx.end(), a.alias.begin(), a.alias.end());
        alias_func.insert(alias_func.end(), a.func.begin(), a.func.end());
        return off;
    }

    // Light::Light + initialize_slots (light.cpp:10-24): colour normalised so max component <= 1, factor folded into scale
    std::string light_topology; // of the light initialised last: Light::compute_topology_hash = color_.topology_hash() (light.h:114-116)
    void init_light_color(vmk_light &l, const Json &p, bool env) {
        l.scale = p["scale"].as_float(1.f);
        l.color = parse_slot(p, "color", 3, {0.5f, 0.5f, 0.5f}, env);
        light_topology = slot_topology;
        if (l.color.tex == VMK_INVALID) { // NumberArray::normalize (number.cpp:38-47)
            float mx = std::max(l.color.v[0], std::max(l.color.v[1], l.color.v[2]));
            if (!(mx < 1.f)) { for (float &c : l.color.v) c = c / mx; l.scale = l.scale * mx; }
        }
    }

    void load(const std::string &json_path) {
        std::ifstream f(json_path);
        if (!f) fail("cannot open scene '" + json_path + "'");
        std::stringstream ss; ss << f.rdbuf();
        Json root;
        try { root = Json::parse(ss.str()); } catch (std::exception &e) { fail(std::string("scene json: ") + e.what()); }
        scene_dir = dir_of(json_path);

        // ---- render_setting / spectrum / mediums (scene_desc.cpp:37-53, node_desc.cpp:371-376) ----
        const Json &rs = root["render_setting"];
        params.ray_offset_factor = rs["ray_offset_factor"].as_float(1.f);
        double min_world_radius = rs["min_world_radius"].as_double(10.0);
        const Json &spec = root["spectrum"];
        std::string spec_type = spec["type"].as_string("srgb");
        if (opt.spectrum == 1) spec_type = "srgb"; else if (opt.spectrum == 2 || opt.spectrum == 3) spec_type = "hero";
        if (spec_type != "srgb" && spec_type != "hero") fail("spectrum/" + spec_type + " is outside the hot-path scope (srgb, hero)");
        spectrum_dimension = 3;
        if (spec_type == "hero") { // HeroWavelengthSpectrum::dimension_ (hero.cpp:240); srgb.cpp is always 3
            spectrum_dimension = opt.spectrum == 3 ? 4u : spec["param"]["dimension"].as_uint(3);
            if (spectrum_dimension != 3 && spectrum_dimension != 4) fail("spectrum/hero with dimension " + std::to_string(spectrum_dimension) + " is outside the hot-path scope (megakernel instances exist for 3 and 4 wavelengths per path)");
        }
        if (spec_type == "hero" && !list_only) init_hero();
        describe("spectrum", spec_type, spectrum_dimension == 4 ? "dimension 4" : "");
        // mediums (scene_desc.cpp:26-35, MediumDesc::init node_desc.cpp:182-197, homogeneous.cpp:20-24)
        uint32_t global_medium = VMK_INVALID;
        params.process_mediums = 0; params.camera_medium = VMK_INVALID;
        if (root.contains("mediums")) {
            const Json &md = root["mediums"];
            bool process = md["process"].as_bool(true);
            if (!opt.mediums) { if (process && md["list"].size()) describe("medium", "ignored", "vmk_host_options.mediums == 0: rendered as the non-fog variant"); }
            else if (process) {
                for (auto &m : md["list"].arr) {
                    std::string type = m["type"].as_string("homogeneous");
                    if (type != "homogeneous") fail("medium/" + type + " is outside the hot-path scope (homogeneous)");
                    const Json &mp = m["param"];
                    if (!mp["medium_name"].as_string("").empty()) fail("medium: measured 'medium_name' tables are outside the hot-path scope");
                    vmk_medium vm{};
                    auto read3 = [&](const Json &j, float *out) { // SlotDesc: scalar -> broadcast, array -> value
                        const Json &v = j.contains("value") ? j["value"] : j;
                        if (v.is_array()) for (int k = 0; k < 3; ++k) out[k] = v.at(std::min<size_t>(k, v.size() - 1)).as_float(0.f);
                        else for (int k = 0; k < 3; ++k) out[k] = v.as_float(0.f);
                    };
                    read3(mp["sigma_a"], vm.sigma_a); read3(mp["sigma_s"], vm.sigma_s);
                    float g = mp["g"].contains("value") ? mp["g"]["value"].as_float(0.f) : mp["g"].as_float(0.f);
                    vm.g = std::min(0.99f, std::max(-0.99f, g));
                    vm.scale = mp["scale"].contains("value") ? mp["scale"]["value"].as_float(1.f) : mp["scale"].as_float(1.f);
                    mediums.push_back(vm); medium_names.push_back(m["name"].as_string());
                    describe("medium", "homogeneous", m["name"].as_string());
                }
                params.process_mediums = mediums.empty() ? 0u : 1u;
                global_medium = medium_id(md["global"].as_string(""));
            }
        }

        // ---- materials ----
        for (auto &md : root["materials"].arr) add_material(md);

        // ---- light_sampler descs first (lights listed in JSON precede shape emissions: LightSampler ctor lightsampler.cpp:15-27) ----
        const Json &lsd = root["light_sampler"];
        std::string ls_type = lsd["type"].as_string("uniform");
        if (ls_type != "uniform" && ls_type != "power") fail("lightsampler/" + ls_type + " is outside the hot-path scope (uniform, power)");
        describe("lightsampler", ls_type, "");
        params.light_sampler = ls_type == "power" ? 1u : 0u;
        params.env_separate = lsd["param"]["env_separate"].as_bool(false) ? 1u : 0u;
        params.env_prob = std::min(0.99f, std::max(0.01f, lsd["param"]["env_prob"].as_float(0.5f)));
        struct PendingLight { vmk_light l; int order; std::string topo; };
        std::vector<PendingLight> pending;
        for (auto &ld : lsd["param"]["lights"].arr) {
            std::string type = ld["type"].as_string("area");
            const Json &p = ld["param"];
            if (type == "spherical") { // spherical.cpp:33-44
                vmk_light l{}; l.type = VMK_LIGHT_SPHERICAL; l.inst_id = VMK_INVALID;
                init_light_color(l, p, true);
                bool flip_u = p["flip_u"].as_bool(false);
                Mat4 o2w = parse_transform(p["o2w"]) * (rotation_x(-90) * scale(1.0, flip_u ? 1.0 : -1.0, 1.0));
                Mat4 w2o = inverse(o2w);
                Mat4 w2o_f{}; for (int i = 0; i < 16; ++i) w2o_f.m[i] = (double) (float) w2o.m[i];
                Mat4 o2w_back = inverse(w2o_f); // spherical.cpp:114 evaluates inverse(*w2o_) per sample
                for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { l.w2o[c * 3 + r] = (float) w2o.at(r, c); l.o2w[c * 3 + r] = (float) o2w_back.at(r, c); }
                pending.push_back({l, 1, "spherical|" + light_topology});
                describe("light", "spherical", ld["name"].as_string());
            } else if (type == "point" || type == "spot") { // point.cpp:19-30, spot.cpp:19-38
                vmk_light l{}; l.type = type == "point" ? VMK_LIGHT_POINT : VMK_LIGHT_SPOT; l.inst_id = VMK_INVALID;
                init_light_color(l, p, false);
                auto read3 = [&](const char *key, float dx, float dy, float dz, float *out) {
                    const Json &v = p[key];
                    out[0] = v.is_array() && v.size() > 0 ? v.at(0).as_float(dx) : dx;
                    out[1] = v.is_array() && v.size() > 1 ? v.at(1).as_float(dy) : dy;
                    out[2] = v.is_array() && v.size() > 2 ? v.at(2).as_float(dz) : dz;
                };
                read3("position", 0.f, 0.f, 0.f, l.position);
                if (l.type == VMK_LIGHT_SPOT) {
                    const float deg = 3.14159265358979323846f / 180.f;
                    float angle = std::min(89.f, std::max(1.f, p["angle"].as_float(45.f))) * deg;
                    // spot.cpp:31: radians(clamp(desc["falloff"], 0, angle_.hv())) — the clamp's upper bound is the angle already in RADIANS
                    float falloff = std::min(std::max(p["falloff"].as_float(10.f), 0.f), angle) * deg;
                    float d[3]; read3("direction", 0.f, 0.f, 1.f, d);
                    float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    for (int k = 0; k < 3; ++k) l.direction[k] = d[k] / len;
                    l.cos_angle = std::cos(angle);
                    l.cos_falloff_start = std::cos(std::max(0.f, angle - falloff));
                }
                pending.push_back({l, 0, type + "|" + light_topology});
                describe("light", type, ld["name"].as_string());
            } else if (type == "projector") { // projector.cpp:31-49
                vmk_light l{}; l.type = VMK_LIGHT_PROJECTOR; l.inst_id = VMK_INVALID;
                init_light_color(l, p, false);
                if (l.color.tex != VMK_INVALID && (l.color.tex & VMK_SLOT_TINTED)) fail("light/projector: a multiply node as the projected image is outside the hot-path scope");
                const float deg = 3.14159265358979323846f / 180.f;
                const float angle_y = std::min(89.f, std::max(1.f, p["angle"].as_float(45.f))) * deg;
                float ratio = p["ratio"].as_float(1.f);
                if (ratio == 0.f) { // the image's aspect ratio (:45-48)
                    if (l.color.tex == VMK_INVALID) fail("light/projector: ratio 0 needs an image colour");
                    const vmk_texture &t = textures[l.color.tex & 0xffffu];
                    ratio = (float) t.width / (float) t.height;
                }
                const float tan_y = std::tan(angle_y); // (the reference evaluates tan(*angle_y_) per sample on the device: ocarina's tan, unpinned, App. B)
                l.tan_xy[0] = ratio * tan_y; l.tan_xy[1] = tan_y;
                Mat4 o2w = parse_transform(p["o2w"]);
                Mat4 o2w_f{}; for (int i = 0; i < 16; ++i) o2w_f.m[i] = (double) (float) o2w.m[i]; // o2w_ is stored as float4x4; inverse(*o2w_) per sample (:100)
                Mat4 w2o = inverse(o2w_f);
                for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) l.w2o4[c * 4 + r] = (float) w2o.at(r, c);
                for (int k = 0; k < 3; ++k) l.position[k] = (float) o2w_f.at(k, 3);                           // position() = o2w_[3].xyz (:91)
                for (int k = 0; k < 3; ++k) l.direction[k] = (float) o2w_f.at(k, 2);                          // direction() = transform_vector(o2w_, (0, 0, 1)) (:95-97)
                pending.push_back({l, 0, type + "|" + light_topology});
                describe("light", type, ld["name"].as_string());
            } else if (type == "area") {
                fail("stand-alone light/area (own quad geometry, area.cpp:56-71) is outside the hot-path scope; use shape.param.emission");
            } else {
                if (opt.drop_unsupported_lights) { describe("light", type, "DROPPED (outside hot-path scope)"); continue; }
                fail("light/" + type + " is outside the hot-path scope (area, spherical, point, spot, projector)");
            }
        }

        // ---- shapes (Scene::load_shapes / add_shape scene.cpp:120-163) ----
        for (auto &sd : root["shapes"].arr) {
            std::string type = sd["type"].as_string();
            const Json &p = sd["param"];
            Mesh mesh;
            if (type == "quad") mesh = make_quad(p);
            else if (type == "cube") mesh = make_cube(p);
            else if (type == "sphere") mesh = make_sphere(p);
            else if (type == "model") {
                std::string fn = p["fn"].as_string();
                if (p["swap_handed"].as_bool(false) || p["subdiv_level"].as_uint(0)) fail("shape/model swap_handed/subdiv_level are outside the hot-path scope");
                if (!ends_with(fn, ".obj")) fail("shape/model: only Wavefront .obj is supported ('" + fn + "')");
                if (!list_only && opt.missing_assets == 1 && !file_exists(join_path(scene_dir, fn))) { // stripped mesh: skipped, and said so
                    describe("shape", "model_skipped", sd["name"].as_string() + " " + fn + " (stand-in: file missing, shape omitted)");
                    continue;
                }
                mesh = list_only ? Mesh{} : load_obj(join_path(scene_dir, fn), p["flip_uv"].as_bool(true), p["smooth"].as_bool(false));
            } else fail("shape/" + type + " is outside the hot-path scope (quad/cube/sphere/model)");
            describe("shape", type, sd["name"].as_string());
            std::string mat_name = p["material"].as_string();
            uint32_t mat_id = VMK_INVALID;
            for (uint32_t i = 0; i < material_names.size(); ++i) if (material_names[i] == mat_name) { mat_id = i; break; } // find_if: first match
            // ShapeGroup::post_init shape.cpp:246-271: explicit {inside, outside} names or the global medium on both sides
            uint32_t m_in = global_medium, m_out = global_medium;
            if (params.process_mediums && p.contains("medium")) { m_in = medium_id(p["medium"]["inside"].as_string("")); m_out = medium_id(p["medium"]["outside"].as_string("")); }
            if (!params.process_mediums) m_in = m_out = VMK_INVALID;
            uint32_t inst_id = add_instance(mesh, parse_transform(p["transform"]), mat_id, m_in, m_out);
            if (p.contains("emission")) { // ShapeDesc::init node_desc.cpp:66-68 -> light/area with inst_id
                const Json &em = p["emission"];
                std::string etype = em["type"].as_string("area");
                if (etype != "area") fail("emission type '" + etype + "' unsupported");
                vmk_light l{}; l.type = VMK_LIGHT_AREA; l.inst_id = inst_id;
                init_light_color(l, em["param"], false);
                l.two_sided = em["param"]["two_sided"].as_bool(false) ? 1u : 0u;
                pending.push_back({l, 0, "area|" + light_topology});
                describe("light", "area", sd["name"].as_string());
            }
        }

        // ---- LightSampler::tidy_up (lightsampler.cpp:64-74): std::sort of the lights by `lights_.topology_index(light)` ----
        // The index comes from ocarina's Polymorphic container (absent from the checkout; restated from its use): a light's topology is
        // its class together with Light::compute_topology_hash() = color_.topology_hash() (light.h:114-116; the slot's hash covers the
        // channel mask, the dimension and the node's topology, shader_node.cpp:187-189 — see slot_topology above), and a topology's
        // index is its rank of FIRST APPEARANCE among the lights in the order Scene::init registers them: the light sampler's own
        // list (scene.cpp:16-35 loads the sampler first), then one area light per emissive shape in shape order (scene.cpp:120-163).
        // Lights of one topology keep their registration order: the comparator never orders them, and every std::sort in use
        // (MSVC <= 32 elements, libstdc++ <= 16) finishes such a range with a stable insertion sort; the shipped scenes have <= 13
        // lights.  Beyond 16 lights the reference's order is whatever its standard library's introsort leaves — implementation
        // defined there, the registration order here, and said so in the scene description.
        {
            std::vector<std::string> first_seen; // topology -> index of first appearance
            auto index_of = [&](const std::string &t) { return (size_t) (std::find(first_seen.begin(), first_seen.end(), t) - first_seen.begin()); };
            for (auto &pl : pending) if (index_of(pl.topo) == first_seen.size()) first_seen.push_back(pl.topo);
            std::stable_sort(pending.begin(), pending.end(), [&](const PendingLight &a, const PendingLight &b) { return index_of(a.topo) < index_of(b.topo); });
            if (pending.size() > 16 && first_seen.size() > 1)
                describe("lightsampler", "tidy_up", std::to_string(pending.size()) + " lights of " + std::to_string(first_seen.size()) + " topologies: beyond 16 the reference's order is std::sort-implementation-defined; registration order kept");
        }
        double ext[3] = {bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2]};
        double aabb_radius = tri_pos.empty() ? 0.0 : 0.5 * std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
        float world_diameter = (float) (std::max(aabb_radius, min_world_radius) * 2.0); // scene.h:108-109
        scene.env_light = VMK_INVALID;
        for (auto &pl : pending) {
            vmk_light l = pl.l;
            uint32_t light_id = (uint32_t) lights.size();
            if (l.type == VMK_LIGHT_AREA) { // AreaLight::prepare (area.cpp:170-176): alias table over triangle areas
                vmk_instance &inst = instances[l.inst_id];
                inst.light_id = light_id;
                std::vector<float> areas;
                for (uint32_t t = 0; t < inst.tri_count; ++t) {
                    const vmk_tri_pos &tp = tri_pos[inst.tri_offset + t];
                    float e1[3] = {tp.p1[0] - tp.p0[0], tp.p1[1] - tp.p0[1], tp.p1[2] - tp.p0[2]}, e2[3] = {tp.p2[0] - tp.p0[0], tp.p2[1] - tp.p0[1], tp.p2[2] - tp.p0[2]};
                    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    areas.push_back(0.5f * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]));
                }
                if (areas.empty()) fail("emissive shape without triangles");
                AliasBuild a = build_alias(areas);
                l.alias_offset = append_alias(a); l.alias_count = (uint32_t) areas.size(); l.alias_integral = a.integral;
            } else if (l.type == VMK_LIGHT_POINT || l.type == VMK_LIGHT_SPOT || l.type == VMK_LIGHT_PROJECTOR) {
                // delta lights carry no tables
            } else { // SphericalMap::prepare (spherical.cpp:198-212) + AliasTable2D::build (alias2d.cpp:32-68)
                scene.env_light = light_id;
                l.world_diameter = world_diameter;
                uint32_t rx = 1, ry = 1; std::vector<float> weights;
                if (l.color.tex == VMK_INVALID || list_only) weights.push_back(1.f);
                else { // calculate_weights (spherical.cpp:170-196) incl. its `v = idx / res.y + 0.5; theta = v / res.x` quirk
                    const vmk_texture &t = textures[l.color.tex & 0xffffu];
                    rx = t.width; ry = t.height; weights.resize((size_t) rx * ry);
                    for (uint32_t idx = 0; idx < rx * ry; ++idx) {
                        float v = (float) (idx / ry) + 0.5f;
                        float theta = v / (float) rx;
                        float sinTheta = std::sin(3.14159265358979323846f * theta);
                        float lum;
                        if (t.format == VMK_TEX_RGBA32F) { const float *px = (const float *) (tex_data.data() + t.offset) + (size_t) idx * 4; lum = 0.212671f * px[0] + 0.715160f * px[1] + 0.072169f * px[2]; }
                        else { const uint8_t *px = tex_data.data() + t.offset + (size_t) idx * 4; lum = 0.212671f * (px[0] / 255.f) + 0.715160f * (px[1] / 255.f) + 0.072169f * (px[2] / 255.f); }
                        weights[idx] = lum * sinTheta;
                    }
                }
                std::vector<float> marginal; std::vector<AliasBuild> rows;
                for (uint32_t v = 0; v < ry; ++v) { rows.push_back(build_alias(std::vector<float>(weights.begin() + (size_t) v * rx, weights.begin() + (size_t) (v + 1) * rx))); marginal.push_back(rows.back().integral); }
                AliasBuild mg = build_alias(marginal);
                l.alias_offset = append_alias(mg); l.alias_count = ry; l.alias_integral = mg.integral;
                l.cond_offset = (uint32_t) alias_prob.size();
                for (auto &r : rows) append_alias(r);
                l.res_x = rx; l.res_y = ry;
            }
            lights.push_back(l);
        }
        if (lights.empty() && !list_only) fail("scene has no light inside the hot-path scope (area / spherical / point / spot)");
        // ---- PowerLightSampler::prepare (power.cpp:35-52): alias table over luminance(power()) ----
        scene.light_alias_offset = VMK_INVALID; scene.light_alias_integral = 0.f;
        if (params.light_sampler == 1 && !list_only) {
            std::vector<float> weights;
            for (const vmk_light &l : lights) {
                // Light::average() = colour average * scale (light.h:80-83); an image colour averages its texels
                float avg[3] = {l.color.v[0], l.color.v[1], l.color.v[2]};
                if (l.color.tex != VMK_INVALID) {
                    const vmk_texture &t = textures[l.color.tex & 0xffffu];
                    double acc[3] = {0, 0, 0};
                    for (size_t i = 0; i < (size_t) t.width * t.height; ++i) {
                        if (t.format == VMK_TEX_RGBA32F) { const float *px = (const float *) (tex_data.data() + t.offset) + i * 4; for (int k = 0; k < 3; ++k) acc[k] += px[k]; }
                        else { const uint8_t *px = tex_data.data() + t.offset + i * 4; for (int k = 0; k < 3; ++k) acc[k] += px[k] / 255.0; }
                    }
                    for (int k = 0; k < 3; ++k) avg[k] = (float) (acc[k] / ((double) t.width * t.height)) * l.color.v[0];
                }
                for (float &c : avg) c *= l.scale;
                float f = 0.f;
                const float pi = 3.14159265358979323846f;
                if (l.type == VMK_LIGHT_AREA) { // area.cpp:87-89
                    float area = alias_func_sum(l);
                    f = (l.two_sided ? 2.f : 1.f) * area * pi;
                } else if (l.type == VMK_LIGHT_SPHERICAL) { // spherical.cpp:56-59; weighs nothing when sampled separately
                    f = params.env_separate ? 0.f : pi * (l.world_diameter / 2.f) * (l.world_diameter / 2.f);
                } else if (l.type == VMK_LIGHT_POINT) f = 4.f * pi; // point.cpp:33-35
                else if (l.type == VMK_LIGHT_PROJECTOR) { // projector.cpp:59-89: twice the solid angle of a spherical triangle over 4 pi
                    const float ratio = l.tan_xy[0] / l.tan_xy[1];
                    const float y = std::sqrt(1.f / (ratio * ratio + 1.f)), x = ratio * y, z = std::sqrt(x * x + y * y);
                    auto cr = [](const float *a, const float *b, float *o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; float n = o[0] * o[0] + o[1] * o[1] + o[2] * o[2]; if (n > 0.f) { n = std::sqrt(n); o[0] /= n; o[1] /= n; o[2] /= n; } };
                    const float p0[3] = {x, y, z}, p1[3] = {x, -y, z}, p2[3] = {-x, -y, z};
                    float c01[3], c12[3], c20[3]; cr(p0, p1, c01); cr(p1, p2, c12); cr(p2, p0, c20);
                    auto ang = [](const float *a, const float *b) { float d = -(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]); return std::acos(std::min(1.f, std::max(-1.f, d))); };
                    const float solid = std::fabs(ang(c01, c12) + ang(c12, c20) + ang(c20, c01) - pi);
                    f = (2.f * solid) / (4.f * pi);
                } else { // spot.cpp:48-50 (angles in radians)
                    float angle = std::acos(l.cos_angle), start = std::acos(l.cos_falloff_start);
                    f = 2.f * pi * (1.f - .5f * (angle * 2.f + (angle - start)));
                }
                weights.push_back(0.212671f * avg[0] * f + 0.715160f * avg[1] * f + 0.072169f * avg[2] * f); // luminance(power())
            }
            AliasBuild a = build_alias(weights);
            scene.light_alias_offset = append_alias(a); scene.light_alias_integral = a.integral;
        }

        // ---- sensor (sensor.cpp:17-25,58-78,153-162; thin_lens.cpp:16-20) ----
        const Json &cam = root["camera"];
        std::string cam_type = cam["type"].as_string("thin_lens");
        if (cam_type != "thin_lens" && cam_type != "pinhole") fail("sensor/" + cam_type + " is outside the hot-path scope");
        describe("sensor", cam_type, cam["param"]["name"].as_string());
        const Json &cp = cam["param"];
        if (params.process_mediums) params.camera_medium = cp.contains("medium") ? medium_id(cp["medium"].as_string("")) : global_medium; // photosensory.cpp:16-32
        const Json &fbp = root["pipeline"]["param"]["frame_buffer"]["param"];
        uint32_t W = 1280, H = 720; // frame_buffer.cpp:18 default
        if (fbp["resolution"].is_array() && fbp["resolution"].size() == 2) { W = fbp["resolution"].at(0).as_uint(W); H = fbp["resolution"].at(1).as_uint(H); }
        if (opt.width && opt.height) { W = opt.width; H = opt.height; }
        params.width = W; params.height = H;
        {
            double fov_y = std::min(120.0, std::max(15.0, cp["fov_y"].as_double(20.0))); // sensor.h:20-21,99-111
            Mat4 m = parse_transform(cp["transform"]);
            double pitch = std::atan2(m.at(2, 1), m.at(1, 1)) * 180.0 / M_PI; // m[1][2], m[1][1]
            double yaw = std::atan2(m.at(0, 2), m.at(0, 0)) * 180.0 / M_PI;   // m[2][0], m[0][0]
            Mat4 c2w = translation(m.at(0, 3), m.at(1, 3), m.at(2, 3)) * (scale(1, 1, -1) * rotation_y(yaw) * rotation_x(-pitch));
            for (int i = 0; i < 16; ++i) params.c2w[i] = (float) c2w.m[i];
            // FrameBuffer::update_screen_window (frame_buffer.cpp:93-102), Sensor::update_resolution / update_raster
            double ratio = (double) W / (double) H;
            double lx = -1, ux = 1, ly = -1, uy = 1;
            if (ratio > 1.0) { lx = -ratio; ux = ratio; } else { ly = -1.0 / ratio; uy = 1.0 / ratio; }
            Mat4 screen_to_raster = scale(W, H, 1) * scale(1.0 / (ux - lx), 1.0 / -(uy - ly), 1.0) * translation(-lx, -uy, 0.0);
            double n = 0.01, fz = 1000.0, inv_tan = 1.0 / std::tan(radians(fov_y) / 2.0);
            Mat4 persp{}; persp.at(0, 0) = 1; persp.at(1, 1) = 1; persp.at(2, 2) = fz / (fz - n); persp.at(2, 3) = -fz * n / (fz - n); persp.at(3, 2) = 1;
            Mat4 camera_to_screen = scale(inv_tan, inv_tan, 1) * persp;
            Mat4 raster_to_sensor = inverse(camera_to_screen) * inverse(screen_to_raster);
            for (int i = 0; i < 16; ++i) params.raster_to_sensor[i] = (float) raster_to_sensor.m[i];
            params.focal_distance = cam_type == "thin_lens" ? cp["focal_distance"].as_float(5.f) : 5.f;
            params.lens_radius = cam_type == "thin_lens" ? cp["lens_radius"].as_float(0.f) : 0.f;
        }
        // ---- filter (filter.h:36-39; radius is read as a scalar — an array takes its first element) ----
        {
            const Json &fd = cp["filter"];
            std::string ft = fd["type"].as_string("gaussian");
            const Json &rj = fd["param"]["radius"];
            float radius = rj.is_array() ? rj.at(0).as_float(0.5f) : rj.as_float(0.5f);
            params.filter_radius[0] = params.filter_radius[1] = radius;
            describe("filter", ft, "");
            if (ft == "box") params.filter_type = VMK_FILTER_BOX;
            else if (ft == "triangle") params.filter_type = VMK_FILTER_TRIANGLE;
            else if (ft == "gaussian" || ft == "mitchell" || ft == "sinc") {
                // FittedCurveFilter: FilterSampler::build (fitted_curve.h:37-58) tabulates |f| on a 20x20 grid over one
                // quadrant and importance-samples it with an alias-2D warper; the sample weight (lut / pdf, signed for the
                // negative lobes of mitchell / sinc) is not consumed by the accumulate path (frame_buffer.cpp:117-126).
                params.filter_type = VMK_FILTER_TABLE;
                std::function<float(float, float)> eval;
                if (ft == "gaussian") { // gaussian.cpp:20-42
                    float sigma = fd["param"]["sigma"].as_float(1.f);
                    auto gaussian = [](float x, float mu, float sg) { return 1.f / std::sqrt(2 * 3.14159265358979323846f * sg * sg) * std::exp(-(x - mu) * (x - mu) / (2 * sg * sg)); };
                    float ex = gaussian(radius, 0, sigma);
                    eval = [=](float px, float py) { return std::max(0.f, gaussian(px, 0, sigma) - ex) * std::max(0.f, gaussian(py, 0, sigma) - ex); };
                } else if (ft == "mitchell") { // mitchell.cpp:18-49
                    float mb = fd["param"]["b"].as_float(1.f / 3.f), mc = fd["param"]["c"].as_float(1.f / 3.f);
                    auto m1d = [=](float x) {
                        x = std::fabs(x);
                        if (x <= 1) return ((12 - 9 * mb - 6 * mc) * x * x * x + (-18 + 12 * mb + 6 * mc) * x * x + (6 - 2 * mb)) * (1.f / 6.f);
                        if (x <= 2) return ((-mb - 6 * mc) * x * x * x + (6 * mb + 30 * mc) * x * x + (-12 * mb - 48 * mc) * x + (8 * mb + 24 * mc)) * (1.f / 6.f);
                        return 0.f;
                    };
                    eval = [=](float px, float py) { return m1d(2 * px / radius) * m1d(2 * py / radius); };
                } else { // sinc.cpp:16-33, math/util.h:57-66
                    float tau = fd["param"]["tau"].as_float(3.f);
                    auto sinc = [](float x) { x *= 3.14159265358979323846f; return 1.f + x * x == 1.f ? 1.f : std::sin(x) / x; };
                    auto wsinc = [=](float x) { return std::fabs(x) > radius ? 0.f : sinc(x) * sinc(x / tau); };
                    eval = [=](float px, float py) { return wsinc(px) * wsinc(py) * 4.f; };
                }
                const int N = VMK_FILTER_TABLE_SIZE;
                std::vector<float> func((size_t) N * N);
                for (int i = 0; i < N * N; ++i) {
                    int x = i % N, y = i / N;
                    func[i] = std::fabs(eval((x + 0.5f) / N * radius, (y + 0.5f) / N * radius));
                }
                std::vector<float> marginal; std::vector<AliasBuild> rows;
                for (int v = 0; v < N; ++v) { rows.push_back(build_alias(std::vector<float>(func.begin() + v * N, func.begin() + (v + 1) * N))); marginal.push_back(rows.back().integral); }
                AliasBuild mg = build_alias(marginal);
                for (int v = 0; v < N; ++v) {
                    params.filter_marginal_prob[v] = mg.prob[v]; params.filter_marginal_alias[v] = mg.alias[v]; params.filter_marginal_func[v] = mg.func[v];
                    for (int u = 0; u < N; ++u) { params.filter_cond_prob[v * N + u] = rows[v].prob[u]; params.filter_cond_alias[v * N + u] = rows[v].alias[u]; params.filter_cond_func[v * N + u] = rows[v].func[u]; }
                }
                params.filter_marginal_integral = mg.integral;
            } else fail("filter/" + ft + " is outside the hot-path scope (box / triangle / gaussian / mitchell / sinc)");
        }
        // ---- integrator (integrator.cpp:59-66) ----
        {
            const Json &ig = root["integrator"];
            std::string it = ig["type"].as_string("pt");
            if (it != "pt") fail("integrator/" + it + " is outside the hot-path scope (pt only)");
            describe("integrator", "pt", "");
            const Json &ip = ig["param"];
            params.max_depth = ip["max_depth"].as_uint(16); params.min_depth = ip["min_depth"].as_uint(5);
            params.rr_threshold = ip["rr_threshold"].as_float(1.f); params.mis_mode = (uint32_t) ip["mis_mode"].as_int(0);
            if (ip["separate"].as_bool(false)) describe("integrator", "pt", "note: separate=true has identical arithmetic on this backend");
            if (opt.max_depth >= 0) params.max_depth = (uint32_t) opt.max_depth;
            if (opt.min_depth >= 0) params.min_depth = (uint32_t) opt.min_depth;
        }
        std::string samp = root["sampler"]["type"].as_string("independent");
        if (samp != "independent") fail("sampler/" + samp + " is outside the hot-path scope");
        describe("sampler", "independent", "");
        // ---- frame buffer / tone mapper / output (frame_buffer.cpp:15-26, node_desc.cpp:214-220,360-369) ----
        params.exposure = fbp["exposure"].as_float(1.f);
        std::string tm = fbp["tone_mapper"]["type"].as_string("linear");
        params.tone_mapper = tm == "aces" ? 1u : (tm == "reinhard" ? 2u : 0u);
        describe("pipeline", root["pipeline"]["type"].as_string("fixed"), "");
        describe("framebuffer", "normal", "");
        describe("tonemapper", tm, "");
        output_spp = root["output"]["spp"].as_uint(0);
        output_fn = root["output"]["fn"].as_string("output.png");
    }

    void load_luts() {
        std::string path = opt.lut_path ? opt.lut_path : "";
        if (path.empty()) fail("vmk_host_options.lut_path not set (albedo tables, vision_amd/data/luts.bin)");
        std::ifstream f(path, std::ios::binary);
        if (!f) fail("cannot open albedo-table blob '" + path + "' (generate with tools/make_luts.py)");
        // header: magic 'VLUT', version, counts[7] (floats per table; 0 = absent)
        uint32_t hdr[9];
        f.read((char *) hdr, sizeof(hdr));
        if (!f || hdr[0] != 0x54554c56u || hdr[1] != 1u) fail("bad albedo-table blob '" + path + "'");
        const uint32_t N = VMK_LUT_RES;
        const uint32_t expect[7] = {N * N, N * N * N * 2, N * N * N * 2, N * N * N, N * N * N, N * N * 4, N * N * 4};
        size_t total = 0;
        for (int i = 0; i < 7; ++i) { if (hdr[2 + i] != 0 && hdr[2 + i] != expect[i]) fail("albedo-table blob: unexpected size"); total += hdr[2 + i]; }
        luts.resize(total);
        f.read((char *) luts.data(), (std::streamsize) (total * 4));
        if (!f) fail("albedo-table blob truncated");
        const float **dst[7] = {&scene.luts.pure_reflection, &scene.luts.dielectric, &scene.luts.dielectric_inv, &scene.luts.specular, &scene.luts.coat, &scene.luts.sheen_approx, &scene.luts.sheen_volume};
        size_t off = 0;
        for (int i = 0; i < 7; ++i) { *dst[i] = hdr[2 + i] ? luts.data() + off : nullptr; off += hdr[2 + i]; }
        for (int i = 0; i < 5; ++i) if (!*dst[i]) fail("albedo-table blob lacks a required table");
    }

    float alias_func_sum(const vmk_light &l) const { // an area light's table holds its triangle areas: surface_area()
        float sum = 0.f;
        for (uint32_t i = 0; i < l.alias_count; ++i) sum += alias_func[l.alias_offset + i];
        return sum;
    }
    uint32_t medium_id(const std::string &name) const {
        for (uint32_t i = 0; i < medium_names.size(); ++i) if (medium_names[i] == name) return i;
        return VMK_INVALID;
    }

    void finalize() {
        scene.abi_version = VMK_ABI_VERSION;
        scene.n_mediums = (uint32_t) mediums.size(); scene.mediums = mediums.data();
        scene.n_tris = (uint32_t) tri_pos.size(); scene.n_instances = (uint32_t) instances.size(); scene.n_materials = (uint32_t) materials.size();
        scene.n_lights = (uint32_t) lights.size(); scene.n_textures = (uint32_t) textures.size(); scene.n_alias = (uint32_t) alias_prob.size();
        scene.tri_pos = tri_pos.data(); scene.tri_attr = tri_attr.data(); scene.instances = instances.data(); scene.materials = materials.data();
        scene.lights = lights.data(); scene.textures = textures.data(); scene.tex_data = tex_data.data(); scene.tex_bytes = tex_data.size();
        scene.alias_prob = alias_prob.data(); scene.alias_idx = alias_idx.data(); scene.alias_func = alias_func.data();
        for (int k = 0; k < 3; ++k) { scene.world_min[k] = (float) bmin[k]; scene.world_max[k] = (float) bmax[k]; }
        scene.spectrum = hero ? VMK_SPECTRUM_HERO : VMK_SPECTRUM_SRGB;
        scene.spectrum_dimension = spectrum_dimension;
        scene.rgb2spec = hero ? rgb2spec.data() : nullptr; scene.spd_data = hero ? spd.data() : nullptr; scene.n_spd = (uint32_t) spd.size();
        // sheen needs the LTC tables; without them only sheen_weight == 0 (constant) is accepted
        if (!scene.luts.sheen_approx)
            for (auto &m : materials) if (m.type == VMK_MAT_PRINCIPLED && (m.slot[VMK_P_SHEEN_WEIGHT].tex != VMK_INVALID || m.slot[VMK_P_SHEEN_WEIGHT].v[0] != 0.f))
                fail("principled_bsdf with sheen_weight != 0 needs the LTC sheen tables (absent from the albedo-table blob)");
    }
};

}// namespace vmk

using namespace vmk;

struct vmk_host_scene { HostScene hs; };

extern "C" {

const char *vmk_host_last_error(void) { return g_error.c_str(); }

int vmk_host_register_image(const char *path, uint32_t width, uint32_t height, uint32_t channels, int is_float, const void *pixels) {
    if (!path || !pixels || !width || !height || channels < 1 || channels > 4) { g_error = "vmk_host_register_image: bad argument"; return VMK_ERR_ARG; }
    Image img; img.w = width; img.h = height; img.channels = channels; img.is_float = is_float != 0;
    size_t n = (size_t) width * height * channels;
    if (is_float) img.f32.assign((const float *) pixels, (const float *) pixels + n);
    else img.u8.assign((const uint8_t *) pixels, (const uint8_t *) pixels + n);
    g_images[path] = std::move(img);
    return VMK_OK;
}
void vmk_host_clear_images(void) { g_images.clear(); }

int vmk_host_list_images(const char *json_path, char *buf, uint32_t buf_bytes) {
    try {
        HostScene hs; hs.list_only = true; hs.opt.procedural_env = 1; hs.opt.drop_unsupported_lights = 1;
        hs.load(json_path);
        std::string out;
        for (auto &p : hs.image_paths) out += p + "\n";
        if (out.size() + 1 > buf_bytes) { g_error = "vmk_host_list_images: buffer too small"; return VMK_ERR_ARG; }
        std::memcpy(buf, out.c_str(), out.size() + 1);
        return (int) out.size();
    } catch (std::exception &e) { g_error = e.what(); return VMK_ERR_UNSUPPORTED; }
}

int vmk_host_load_scene(const char *json_path, const vmk_host_options *opt, vmk_host_scene **out) {
    if (!json_path || !out) { g_error = "vmk_host_load_scene: bad argument"; return VMK_ERR_ARG; }
    auto *h = new vmk_host_scene();
    try {
        if (opt) h->hs.opt = *opt; else { h->hs.opt.max_depth = -1; h->hs.opt.min_depth = -1; }
        std::string lut = opt && opt->lut_path ? opt->lut_path : "";
        h->hs.opt.lut_path = lut.empty() ? nullptr : lut.c_str();
        h->hs.data_dir = dir_of(lut);
        h->hs.load_luts();
        h->hs.load(json_path);
        h->hs.finalize();
        h->hs.opt.lut_path = nullptr;
    } catch (std::exception &e) { g_error = e.what(); delete h; return VMK_ERR_UNSUPPORTED; }
    *out = h;
    return VMK_OK;
}
void vmk_host_free_scene(vmk_host_scene *scene) { delete scene; }
const vmk_scene *vmk_host_scene_tables(const vmk_host_scene *scene) { return scene ? &scene->hs.scene : nullptr; }
const vmk_render_params *vmk_host_render_params(const vmk_host_scene *scene) { return scene ? &scene->hs.params : nullptr; }
uint32_t vmk_host_output_spp(const vmk_host_scene *scene) { return scene ? scene->hs.output_spp : 0; }
const char *vmk_host_output_fn(const vmk_host_scene *scene) { return scene ? scene->hs.output_fn.c_str() : ""; }
const char *vmk_host_describe(const vmk_host_scene *scene) { return scene ? scene->hs.description.c_str() : ""; }

// ---- image files without a scene: Image::load / Image::save_image of the reference (ocarina; image_pool.cpp:23-28, pipeline.cpp:190-198) ----
static bool read_file(const std::string &path, std::vector<uint8_t> &bytes) {
    std::ifstream fi(path, std::ios::binary);
    if (!fi) return false;
    bytes.assign((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
    return true;
}
int vmk_host_load_image(const char *path, uint32_t *width, uint32_t *height, uint32_t *channels, int *is_float, void **pixels) {
    if (!path || !width || !height || !channels || !is_float || !pixels) { g_error = "vmk_host_load_image: bad argument"; return VMK_ERR_ARG; }
    try {
        const std::string fn = path;
        std::vector<uint8_t> bytes;
        const void *src = nullptr; size_t n_bytes = 0;
        Image hdr; vmk_img::Decoded dec; vmk_exr::ImageF ex;
        if (ends_with(fn, ".hdr")) {
            if (!load_hdr(fn, hdr)) throw std::runtime_error("cannot decode '" + fn + "' as Radiance .hdr");
            *width = hdr.w; *height = hdr.h; *channels = hdr.f32.size() == (size_t) hdr.w * hdr.h * 4 ? 4u : hdr.channels; *is_float = 1; src = hdr.f32.data(); n_bytes = hdr.f32.size() * 4;
        } else {
            if (!read_file(fn, bytes)) throw std::runtime_error("cannot open '" + fn + "'");
            if (ends_with(fn, ".exr")) {
                ex = vmk_exr::decode(bytes);
                if (!ex.error.empty()) throw std::runtime_error(fn + ": " + ex.error);
                *width = ex.w; *height = ex.h; *channels = ex.channels; *is_float = 1; src = ex.px.data(); n_bytes = ex.px.size() * 4;
            } else if (ends_with(fn, ".png") || ends_with(fn, ".jpg") || ends_with(fn, ".jpeg")) {
                dec = ends_with(fn, ".png") ? vmk_img::decode_png(bytes) : vmk_img::decode_jpeg(bytes);
                if (!dec.error.empty()) throw std::runtime_error(fn + ": " + dec.error);
                *width = dec.w; *height = dec.h; *channels = dec.channels; *is_float = 0; src = dec.px.data(); n_bytes = dec.px.size();
            } else throw std::runtime_error("'" + fn + "': no decoder for this container (.png .jpg .hdr .exr)");
        }
        void *out = std::malloc(n_bytes ? n_bytes : 1);
        if (!out) throw std::runtime_error("out of memory");
        std::memcpy(out, src, n_bytes);
        *pixels = out;
        return VMK_OK;
    } catch (std::exception &e) { g_error = std::string("vmk_host_load_image: ") + e.what(); return VMK_ERR_ARG; }
}
void vmk_host_free_image(void *pixels) { std::free(pixels); }

int vmk_host_final_picture_mode(const char *fn) { // Pipeline::final_picture (pipeline.cpp:337-340): gamma unless the name ends with exr / hdr
    const std::string f = fn ? fn : "";
    return (ends_with(f, "exr") || ends_with(f, "hdr")) ? 2 : 1;
}
int vmk_host_save_image(const char *path, uint32_t width, uint32_t height, const float *rgba) {
    if (!path || !rgba || !width || !height) { g_error = "vmk_host_save_image: bad argument"; return VMK_ERR_ARG; }
    try {
        const std::string fn = path;
        std::vector<uint8_t> bytes;
        if (ends_with(fn, ".exr")) bytes = vmk_exr::encode(width, height, rgba, 3, false, true); // float32 B G R, ZIP
        else if (ends_with(fn, ".hdr")) bytes = vmk_img::encode_hdr(width, height, rgba);
        else if (ends_with(fn, ".png")) {
            std::vector<uint8_t> px((size_t) width * height * 3);
            for (size_t i = 0; i < (size_t) width * height; ++i) for (int c = 0; c < 3; ++c) {
                float v = rgba[i * 4 + c];
                v = v != v ? 0.f : std::min(1.f, std::max(0.f, v));
                px[i * 3 + c] = (uint8_t) (v * 255.f + 0.5f);
            }
            bytes = vmk_img::encode_png(width, height, 3, px.data());
        } else throw std::runtime_error("'" + fn + "': no encoder for this container (.png .exr .hdr)");
        std::ofstream fo(fn, std::ios::binary);
        if (!fo) throw std::runtime_error("cannot write '" + fn + "'");
        fo.write((const char *) bytes.data(), (std::streamsize) bytes.size());
        if (!fo) throw std::runtime_error("short write to '" + fn + "'");
        return VMK_OK;
    } catch (std::exception &e) { g_error = std::string("vmk_host_save_image: ") + e.what(); return VMK_ERR_ARG; }
}

int vmk_host_build_rgb2spec(const char *spectra_path, const char *out_path, uint32_t threads) {
    if (!spectra_path || !out_path) { g_error = "vmk_host_build_rgb2spec: bad argument"; return VMK_ERR_ARG; }
    std::ifstream f(spectra_path, std::ios::binary);
    uint32_t hdr[3] = {0, 0, 0};
    f.read((char *) hdr, sizeof(hdr));
    std::vector<float> cie(4 * 471);
    f.read((char *) cie.data(), (std::streamsize) (cie.size() * 4));
    if (!f || hdr[0] != 0x44505356u || hdr[1] != 1u || hdr[2] != 471u) { g_error = std::string("vmk_host_build_rgb2spec: bad spectra blob '") + spectra_path + "'"; return VMK_ERR_ARG; }
    const size_t n = (size_t) 3 * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * 4;
    std::vector<float> table(n);
    rgb2spec::optimise(cie.data(), table.data(), threads);
    std::ofstream o(out_path, std::ios::binary);
    const uint32_t ohdr[3] = {0x53325256u, 1u, VMK_RGB2SPEC_RES}; // 'VR2S'
    o.write((const char *) ohdr, sizeof(ohdr));
    o.write((const char *) table.data(), (std::streamsize) (n * 4));
    if (!o) { g_error = std::string("vmk_host_build_rgb2spec: cannot write '") + out_path + "'"; return VMK_ERR_ARG; }
    return VMK_OK;
}

}// extern "C"
