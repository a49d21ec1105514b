// liblfs_io.so — host side of SURVEY.md §8f row 4: COLMAP sparse models, splat PLY files, image headers / lossless decoders.
// Plain C++17 + zlib; no GPU, no libtorch. Interface and reference citations: include/lfs_io.h.
//
// Design notes (how this differs from the reference's loaders, on purpose):
//   * every read goes through a bounds-checked cursor: a truncated or corrupt COLMAP file is an error message, where the
//     reference walks a raw pointer past the buffer (colmap.cpp:305-455 never compares against `end` before reading);
//   * camera models are one table (parameter count, focal layout, which raw parameters are radial / tangential, projection
//     type), not an eleven-way switch; the numbers in the table are the reference's (colmap.cpp:172-262, :682-830);
//   * float arithmetic is spelled out in float32 in the order libtorch evaluates it (normalize -> rotation matrix ->
//     -R^T t), compiled without FMA contraction, so results match the reference's tensors to the last bit or two.
#include "../../include/lfs_io.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace fs = std::filesystem;

// lfs_jpeg.cpp
uint8_t* lfs_decode_jpeg_rgb8(const uint8_t* data, size_t size, int32_t* width, int32_t* height, bool* unsupported, std::string* error);

namespace {

thread_local std::string g_error;

struct IoError : std::runtime_error {
    int code;
    IoError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw IoError(code, buf);
}

template <class F>
int guarded(F&& f) {
    try {
        f();
        return LFS_IO_OK;
    } catch (const IoError& e) {
        g_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_error = e.what();
        return LFS_IO_E_INVALID;
    }
}

std::vector<char> slurp(const fs::path& p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) fail(LFS_IO_E_NOT_FOUND, "Failed to open %s", p.string().c_str());
    const std::streamsize n = f.tellg();
    std::vector<char> buf((size_t)n);
    f.seekg(0);
    if (n && !f.read(buf.data(), n)) fail(LFS_IO_E_FORMAT, "Short read on %s", p.string().c_str());
    return buf;
}

// ---------------------------------------------------------------------------------------------------------------------
// bounds-checked little-endian cursor
// ---------------------------------------------------------------------------------------------------------------------
class Cursor {
  public:
    Cursor(const std::vector<char>& b, std::string what) : p_(b.data()), end_(b.data() + b.size()), what_(std::move(what)) {}
    template <class T>
    T get() {
        need(sizeof(T));
        T v;
        std::memcpy(&v, p_, sizeof(T));
        p_ += sizeof(T);
        return v;
    }
    void skip(uint64_t n) {
        need(n);
        p_ += n;
    }
    std::string cstr() {
        const void* z = std::memchr(p_, 0, size_t(end_ - p_));
        if (!z) fail(LFS_IO_E_FORMAT, "%s: unterminated string", what_.c_str());
        std::string s(p_, (const char*)z);
        p_ = (const char*)z + 1;
        return s;
    }
    void expect_end() const {
        if (p_ != end_) fail(LFS_IO_E_FORMAT, "%s: trailing bytes", what_.c_str());
    }

  private:
    void need(uint64_t n) const {
        if (uint64_t(end_ - p_) < n) fail(LFS_IO_E_FORMAT, "%s: unexpected end of file", what_.c_str());
    }
    const char* p_;
    const char* end_;
    std::string what_;
};

// ---------------------------------------------------------------------------------------------------------------------
// COLMAP
// ---------------------------------------------------------------------------------------------------------------------
struct ModelInfo {
    const char* name;
    int n_params;          // as stored by COLMAP
    int n_focal;           // 1: f, cx, cy ...   2: fx, fy, cx, cy ...
    int n_radial;          // raw-parameter indices of the radial coefficients, in the order gsplat wants them
    int radial[6];
    int n_tangential;
    int tangential[2];
    int projection;        // gsplat::CameraModelType (0 pinhole, 2 fisheye), -1: rejected by the reference
    const char* rejection; // message of that rejection
};
// colmap.cpp:117-129 (ids, counts), :172-262 (what scales), :682-830 (what goes where)
const ModelInfo kModels[11] = {
    {"SIMPLE_PINHOLE", 3, 1, 0, {}, 0, {}, 0, nullptr},
    {"PINHOLE", 4, 2, 0, {}, 0, {}, 0, nullptr},
    {"SIMPLE_RADIAL", 4, 1, 1, {3}, 0, {}, 0, nullptr}, // k1 kept only when non-zero (:712-717)
    {"RADIAL", 5, 1, 2, {3, 4}, 0, {}, 0, nullptr},
    {"OPENCV", 8, 2, 2, {4, 5}, 2, {6, 7}, 0, nullptr},
    {"OPENCV_FISHEYE", 8, 2, 4, {4, 5, 6, 7}, 0, {}, 2, nullptr},
    {"FULL_OPENCV", 12, 2, 6, {4, 5, 8, 9, 10, 11}, 2, {6, 7}, 0, nullptr},
    {"FOV", 5, 2, 0, {}, 0, {}, -1, "FOV camera model is not supported."},
    {"SIMPLE_RADIAL_FISHEYE", 4, 1, 1, {3}, 0, {}, 2, nullptr},
    {"RADIAL_FISHEYE", 5, 1, 2, {3, 4}, 0, {}, 2, nullptr},
    {"THIN_PRISM_FISHEYE", 12, 2, 4, {4, 5, 8, 9}, 2, {6, 7}, -1,
     "THIN_PRISM_FISHEYE camera model is not supported but could be implemented in 3DGUT pretty easily"},
};

struct RawCamera {
    uint32_t id = 0;
    int model = 0;
    uint64_t width = 0, height = 0;
    std::vector<float> params;
};
struct RawImage {
    uint32_t id = 0, camera_id = 0;
    float q[4] = {1, 0, 0, 0}, t[3] = {0, 0, 0};
    std::string name;
};

// "images_4" -> 4 (colmap.cpp:265-283): the text after the last '_' parsed as a float in (0, 16]
float folder_scale(const std::string& folder) {
    const size_t us = folder.rfind('_');
    if (us == std::string::npos) return 1.f;
    const std::string tail = folder.substr(us + 1);
    char* endp = nullptr;
    const float v = std::strtof(tail.c_str(), &endp);
    if (endp == tail.c_str()) return 1.f; // std::stof throws on no conversion; trailing junk is accepted there too
    return (v > 0.f && v <= 16.f) ? v : 1.f;
}

void scale_raw(RawCamera& c, std::vector<double>& raw, float factor) {
    if (factor == 1.f) return;
    c.width = (uint64_t)((float)c.width / factor); // uint64 / float -> float in C++, truncated (:369-370)
    c.height = (uint64_t)((float)c.height / factor);
    const int n = kModels[c.model].n_focal + 2;    // focal length(s) and principal point; distortion is dimensionless
    for (int i = 0; i < n && i < (int)raw.size(); ++i) raw[i] /= factor;
}

fs::path find_ci(const fs::path& dir, const std::string& target) {
    std::error_code ec;
    if (!fs::is_directory(dir, ec)) return {};
    auto lower = [](std::string s) {
        std::transform(s.begin(), s.end(), s.begin(), [](unsigned char ch) { return (char)std::tolower(ch); });
        return s;
    };
    const std::string want = lower(target);
    for (fs::directory_iterator it(dir, ec), end; !ec && it != end; it.increment(ec))
        if (it->is_regular_file(ec) && lower(it->path().filename().string()) == want) return it->path();
    return {};
}

fs::path sparse_file(const fs::path& base, const std::string& name) {
    const fs::path dirs[3] = {base / "sparse" / "0", base / "sparse", base};
    for (const auto& d : dirs) {
        fs::path f = find_ci(d, name);
        if (!f.empty()) return f;
    }
    std::string msg = "Cannot find '" + name + "' in any of these locations:\n";
    for (const auto& d : dirs) msg += "  - " + (d / name).string() + "\n";
    msg += "Searched case-insensitively for: " + name;
    throw IoError(LFS_IO_E_NOT_FOUND, msg);
}

std::unordered_map<uint32_t, RawCamera> cameras_bin(const fs::path& file, float factor) {
    const auto buf = slurp(file);
    Cursor cur(buf, "cameras.bin");
    const uint64_t n = cur.get<uint64_t>();
    std::unordered_map<uint32_t, RawCamera> out;
    for (uint64_t i = 0; i < n; ++i) {
        RawCamera c;
        c.id = cur.get<uint32_t>();
        const int32_t model = cur.get<int32_t>();
        c.width = cur.get<uint64_t>();
        c.height = cur.get<uint64_t>();
        if (model < 0 || model > 10) fail(LFS_IO_E_UNSUPPORTED, "Unsupported camera-model id %d", model);
        c.model = model;
        std::vector<double> raw(kModels[model].n_params);
        for (auto& v : raw) v = cur.get<double>();
        scale_raw(c, raw, factor);
        c.params.assign(raw.begin(), raw.end()); // double -> float, after the scaling (:392-394)
        out.emplace(c.id, std::move(c));
    }
    cur.expect_end();
    return out;
}

std::vector<RawImage> images_bin(const fs::path& file) {
    const auto buf = slurp(file);
    Cursor cur(buf, "images.bin");
    const uint64_t n = cur.get<uint64_t>();
    std::vector<RawImage> out;
    out.reserve((size_t)std::min<uint64_t>(n, 1u << 20));
    for (uint64_t i = 0; i < n; ++i) {
        RawImage im;
        im.id = cur.get<uint32_t>();
        for (float& v : im.q) v = (float)cur.get<double>();
        for (float& v : im.t) v = (float)cur.get<double>();
        im.camera_id = cur.get<uint32_t>();
        im.name = cur.cstr();
        const uint64_t n2d = cur.get<uint64_t>();
        if (n2d > (uint64_t(1) << 40)) fail(LFS_IO_E_FORMAT, "images.bin: implausible 2-D point count");
        cur.skip(n2d * 24); // (x, y) doubles + point3D id
        out.push_back(std::move(im));
    }
    cur.expect_end();
    return out;
}

// text files (colmap.cpp:459-488): '#' lines dropped, trailing '\r' stripped, empty last line dropped, empty file = error
std::vector<std::string> text_lines(const fs::path& file) {
    std::ifstream f(file);
    if (!f) fail(LFS_IO_E_NOT_FOUND, "Failed to open %s", file.string().c_str());
    std::vector<std::string> lines;
    for (std::string line; std::getline(f, line);) {
        if (!line.empty() && line[0] == '#') continue;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        lines.push_back(std::move(line));
    }
    if (lines.empty()) fail(LFS_IO_E_FORMAT, "File %s is empty or contains no valid lines", file.string().c_str());
    if (lines.back().empty()) lines.pop_back();
    return lines;
}

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t a = 0;
    for (size_t b; (b = s.find(sep, a)) != std::string::npos; a = b + 1) out.push_back(s.substr(a, b - a));
    out.push_back(s.substr(a));
    return out;
}

double to_f64(const std::string& s, const char* what) {
    char* e = nullptr;
    const double v = std::strtod(s.c_str(), &e);
    if (e == s.c_str()) fail(LFS_IO_E_FORMAT, "Invalid number '%s' in %s", s.c_str(), what);
    return v;
}
float to_f32(const std::string& s, const char* what) {
    char* e = nullptr;
    const float v = std::strtof(s.c_str(), &e); // std::stof
    if (e == s.c_str()) fail(LFS_IO_E_FORMAT, "Invalid number '%s' in %s", s.c_str(), what);
    return v;
}
long long to_int(const std::string& s, const char* what) {
    char* e = nullptr;
    const long long v = std::strtoll(s.c_str(), &e, 10);
    if (e == s.c_str()) fail(LFS_IO_E_FORMAT, "Invalid integer '%s' in %s", s.c_str(), what);
    return v;
}

std::unordered_map<uint32_t, RawCamera> cameras_txt(const fs::path& file, float factor) {
    std::unordered_map<uint32_t, RawCamera> out;
    for (const auto& line : text_lines(file)) {
        const auto tok = split(line, ' ');
        if (tok.size() < 4) fail(LFS_IO_E_FORMAT, "Invalid format in cameras.txt: %s", line.c_str());
        RawCamera c;
        c.id = (uint32_t)to_int(tok[0], "cameras.txt");
        c.model = -1;
        for (int m = 0; m < 11; ++m)
            if (tok[1] == kModels[m].name) c.model = m;
        if (c.model < 0) fail(LFS_IO_E_FORMAT, "Invalid format in cameras.txt: %s", line.c_str());
        c.width = (uint64_t)to_int(tok[2], "cameras.txt");
        c.height = (uint64_t)to_int(tok[3], "cameras.txt");
        std::vector<double> raw;
        for (size_t j = 4; j < tok.size(); ++j) raw.push_back(to_f64(tok[j], "cameras.txt"));
        scale_raw(c, raw, factor);
        c.params.assign(raw.begin(), raw.end());
        out.emplace(c.id, std::move(c));
    }
    return out;
}

std::vector<RawImage> images_txt(const fs::path& file) {
    const auto lines = text_lines(file);
    if (lines.size() % 2) fail(LFS_IO_E_FORMAT, "images.txt should have an even number of lines");
    std::vector<RawImage> out;
    for (size_t i = 0; i < lines.size(); i += 2) { // second line of each pair: the 2-D points, unused
        const auto tok = split(lines[i], ' ');
        if (tok.size() != 10) fail(LFS_IO_E_FORMAT, "Invalid format in images.txt line %zu", i + 1);
        RawImage im;
        im.id = (uint32_t)to_int(tok[0], "images.txt");
        for (int k = 0; k < 4; ++k) im.q[k] = to_f32(tok[1 + k], "images.txt");
        for (int k = 0; k < 3; ++k) im.t[k] = to_f32(tok[5 + k], "images.txt");
        im.camera_id = (uint32_t)to_int(tok[8], "images.txt");
        im.name = tok[9];
        out.push_back(std::move(im));
    }
    return out;
}

// F::normalize(q, dim 0) then the rotation matrix, all in float32 (colmap.cpp:29-50)
void quat_to_rotmat(const float qraw[4], float R[9]) {
    float ss = 0.f;
    for (int k = 0; k < 4; ++k) ss += qraw[k] * qraw[k];
    const float den = std::max(std::sqrt(ss), 1e-12f);
    const float w = qraw[0] / den, x = qraw[1] / den, y = qraw[2] / den, z = qraw[3] / den;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

} // namespace

// image header parsers are defined further down
static void image_info_impl(const fs::path& p, int32_t& w, int32_t& h, int32_t& c);

struct lfs_colmap_scene {
    std::vector<lfs_colmap_view> views;
    std::vector<std::string> names, paths;
    float center[3] = {0, 0, 0};
};

namespace {

void assemble(lfs_colmap_scene& sc, const fs::path& base, const std::string& folder, const std::unordered_map<uint32_t, RawCamera>& cams,
              const std::vector<RawImage>& images) {
    const fs::path img_dir = base / folder;
    if (!fs::exists(img_dir)) fail(LFS_IO_E_NOT_FOUND, "Images folder does not exist: %s", img_dir.string().c_str());
    sc.views.resize(images.size());
    std::vector<float> locations(images.size() * 3);
    for (size_t i = 0; i < images.size(); ++i) {
        const RawImage& im = images[i];
        const auto it = cams.find(im.camera_id);
        if (it == cams.end()) fail(LFS_IO_E_FORMAT, "Camera ID %u not found", im.camera_id);
        const RawCamera& c = it->second;
        const ModelInfo& mi = kModels[c.model];
        lfs_colmap_view v{};
        v.camera_id = c.id; v.colmap_model = c.model; v.width = c.width; v.height = c.height;
        quat_to_rotmat(im.q, v.R);
        std::memcpy(v.T, im.t, sizeof v.T);
        // camera position = -R^T t (float32 matmul of a 3x3 with a 3-vector: sum in index order)
        for (int r = 0; r < 3; ++r) locations[3 * i + r] = -((v.R[0 + r] * v.T[0] + v.R[3 + r] * v.T[1]) + v.R[6 + r] * v.T[2]);
        if (mi.projection < 0) fail(LFS_IO_E_UNSUPPORTED, "%s", mi.rejection);
        if ((int)c.params.size() < mi.n_params) fail(LFS_IO_E_FORMAT, "Camera %u: %s needs %d parameters, got %zu", c.id, mi.name, mi.n_params, c.params.size());
        v.n_params = (int32_t)std::min<size_t>(c.params.size(), 12);
        std::copy_n(c.params.begin(), v.n_params, v.params);
        const float* p = c.params.data();
        v.focal_x = p[0];
        v.focal_y = mi.n_focal == 2 ? p[1] : p[0];
        v.center_x = p[mi.n_focal];
        v.center_y = p[mi.n_focal + 1];
        v.camera_model_type = mi.projection;
        const bool drop_zero_k1 = c.model == LFS_COLMAP_SIMPLE_RADIAL && p[3] == 0.f;
        v.n_radial = drop_zero_k1 ? 0 : mi.n_radial;
        for (int k = 0; k < v.n_radial; ++k) v.radial[k] = p[mi.radial[k]];
        v.n_tangential = mi.n_tangential;
        for (int k = 0; k < v.n_tangential; ++k) v.tangential[k] = p[mi.tangential[k]];
        sc.views[i] = v;
        sc.names.push_back(im.name);
        sc.paths.push_back((img_dir / im.name).string());
    }
    // the real size of the first image overrides the database (colmap.cpp:836-865)
    if (!sc.views.empty() && fs::exists(sc.paths[0])) {
        int32_t w = 0, h = 0, ch = 0;
        image_info_impl(sc.paths[0], w, h, ch);
        const float sx = (float)w / (float)(int)sc.views[0].width, sy = (float)h / (float)(int)sc.views[0].height;
        if (std::fabs(sx - 1.f) > 1e-5 || std::fabs(sy - 1.f) > 1e-5)
            for (auto& v : sc.views) {
                v.width = (uint64_t)w; v.height = (uint64_t)h;
                v.focal_x *= sx; v.focal_y *= sy; v.center_x *= sx; v.center_y *= sy;
            }
    }
    // camera_locations.mean(0): float32 sum over the views, then one division
    for (int r = 0; r < 3; ++r) {
        float s = 0.f;
        for (size_t i = 0; i < images.size(); ++i) s += locations[3 * i + r];
        sc.center[r] = images.empty() ? NAN : s / (float)images.size();
    }
}

} // namespace

struct lfs_point_cloud {
    std::vector<float> positions;
    std::vector<uint8_t> colors;
};

// ---------------------------------------------------------------------------------------------------------------------
// PLY
// ---------------------------------------------------------------------------------------------------------------------
struct lfs_ply {
    struct Prop { std::string name; int type; };             // type: index into kPlyTypes
    std::vector<Prop> props;
    uint64_t n_vertices = 0;
    bool ascii = false;
    std::vector<char> file;
    size_t data_offset = 0;                                  // first byte of the vertex element
};

namespace {
struct PlyType { const char* a; const char* b; int size; };
const PlyType kPlyTypes[8] = {{"char", "int8", 1}, {"uchar", "uint8", 1}, {"short", "int16", 2}, {"ushort", "uint16", 2},
                              {"int", "int32", 4}, {"uint", "uint32", 4}, {"float", "float32", 4}, {"double", "float64", 8}};
int ply_type(const std::string& s) {
    for (int i = 0; i < 8; ++i)
        if (s == kPlyTypes[i].a || s == kPlyTypes[i].b) return i;
    fail(LFS_IO_E_FORMAT, "PLY: unknown property type '%s'", s.c_str());
}
float ply_scalar(const char* p, int type) {
    switch (type) {
    case 0: { int8_t v; std::memcpy(&v, p, 1); return v; }
    case 1: { uint8_t v; std::memcpy(&v, p, 1); return v; }
    case 2: { int16_t v; std::memcpy(&v, p, 2); return v; }
    case 3: { uint16_t v; std::memcpy(&v, p, 2); return v; }
    case 4: { int32_t v; std::memcpy(&v, p, 4); return (float)v; }
    case 5: { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    case 6: { float v; std::memcpy(&v, p, 4); return v; }
    default: { double v; std::memcpy(&v, p, 8); return (float)v; }
    }
}

void ply_parse(lfs_ply& ply, const fs::path& path) {
    ply.file = slurp(path);
    const std::vector<char>& f = ply.file;
    if (f.size() < 10) fail(LFS_IO_E_FORMAT, "File too small to be valid PLY");
    if (std::memcmp(f.data(), "ply", 3) != 0 || (f[3] != '\n' && f[3] != '\r')) fail(LFS_IO_E_FORMAT, "Invalid PLY file - missing PLY header");
    size_t pos = 0;
    bool in_vertex = false, seen_vertex = false, done = false, format_seen = false;
    size_t skip_bytes = 0; // fixed-size elements declared before "vertex"
    uint64_t other_count = 0; size_t other_stride = 0; bool other_open = false;
    while (pos < f.size()) {
        size_t e = pos;
        while (e < f.size() && f[e] != '\n') ++e;
        std::string line(f.data() + pos, f.data() + e);
        pos = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line == "end_header") { done = true; break; }
        const auto tok = split(line, ' ');
        if (tok[0] == "format" && tok.size() >= 2) {
            format_seen = true;
            if (tok[1] == "ascii") ply.ascii = true;
            else if (tok[1] != "binary_little_endian") fail(LFS_IO_E_UNSUPPORTED, "PLY: format '%s' is not supported", tok[1].c_str());
        } else if (tok[0] == "element" && tok.size() >= 3) {
            if (other_open && !seen_vertex) skip_bytes += other_count * other_stride;
            other_open = false;
            in_vertex = tok[1] == "vertex";
            if (in_vertex) { ply.n_vertices = (uint64_t)to_int(tok[2], "PLY header"); seen_vertex = true; }
            else { other_open = true; other_count = (uint64_t)to_int(tok[2], "PLY header"); other_stride = 0; }
        } else if (tok[0] == "property" && tok.size() >= 3) {
            if (tok[1] == "list") {
                if (in_vertex || !seen_vertex) fail(LFS_IO_E_UNSUPPORTED, "PLY: list properties before or inside the vertex element are not supported");
                continue;
            }
            const int t = ply_type(tok[1]);
            if (in_vertex) ply.props.push_back({tok[2], t});
            else other_stride += kPlyTypes[t].size;
        }
    }
    if (!done) fail(LFS_IO_E_FORMAT, "No end_header found in PLY file");
    if (!format_seen || !seen_vertex) fail(LFS_IO_E_FORMAT, "PLY: no format line or no vertex element");
    if (ply.ascii && skip_bytes) fail(LFS_IO_E_UNSUPPORTED, "PLY: ascii files with elements before 'vertex' are not supported");
    ply.data_offset = pos + skip_bytes;
    if (!ply.ascii) {
        size_t stride = 0;
        for (const auto& p : ply.props) stride += kPlyTypes[p.type].size;
        if (ply.data_offset > f.size() || (f.size() - ply.data_offset) / std::max<size_t>(stride, 1) < ply.n_vertices)
            fail(LFS_IO_E_FORMAT, "PLY: vertex data is truncated (%llu vertices of %zu bytes declared)", (unsigned long long)ply.n_vertices, stride);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// images
// ---------------------------------------------------------------------------------------------------------------------
uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
uint32_t be16(const unsigned char* p) { return (uint32_t(p[0]) << 8) | p[1]; }
const unsigned char kPngSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

struct Pnm { int kind = 0, w = 0, h = 0, maxval = 0; size_t data = 0; };
bool pnm_header(const std::vector<char>& f, Pnm& out) {
    if (f.size() < 3 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return false;
    out.kind = f[1] - '0';
    size_t pos = 2;
    int vals[3], n = 0;
    while (n < 3 && pos < f.size()) {
        while (pos < f.size() && (std::isspace((unsigned char)f[pos]) || f[pos] == '#')) {
            if (f[pos] == '#') while (pos < f.size() && f[pos] != '\n') ++pos;
            else ++pos;
        }
        int v = 0; bool any = false;
        while (pos < f.size() && std::isdigit((unsigned char)f[pos])) { v = v * 10 + (f[pos] - '0'); ++pos; any = true; }
        if (!any) return false;
        vals[n++] = v;
    }
    if (n < 3 || pos >= f.size()) return false;
    out.w = vals[0]; out.h = vals[1]; out.maxval = vals[2]; out.data = pos + 1; // one whitespace byte after maxval
    return out.w > 0 && out.h > 0 && out.maxval > 0 && out.maxval < 65536;
}

int png_channels(int color_type, bool has_trns) {
    switch (color_type) {
    case 0: return has_trns ? 2 : 1;
    case 2: return has_trns ? 4 : 3;
    case 3: return has_trns ? 4 : 3;
    case 4: return 2;
    case 6: return 4;
    default: fail(LFS_IO_E_FORMAT, "PNG: invalid colour type %d", color_type);
    }
}

} // namespace

static void image_info_impl(const fs::path& path, int32_t& w, int32_t& h, int32_t& c) {
    const auto f = slurp(path);
    const unsigned char* u = (const unsigned char*)f.data();
    if (f.size() >= 33 && std::memcmp(u, kPngSig, 8) == 0) {
        w = (int32_t)be32(u + 16); h = (int32_t)be32(u + 20);
        const int ct = u[25];
        bool trns = false; // a tRNS chunk before IDAT adds an alpha channel
        for (size_t pos = 8; pos + 12 <= f.size();) {
            const uint32_t len = be32(u + pos);
            if (std::memcmp(u + pos + 4, "tRNS", 4) == 0) trns = true;
            if (std::memcmp(u + pos + 4, "IDAT", 4) == 0) break;
            pos += 12 + (size_t)len;
        }
        c = png_channels(ct, trns);
        return;
    }
    if (f.size() >= 4 && u[0] == 0xFF && u[1] == 0xD8) { // JPEG: walk the segments to the frame header
        size_t pos = 2;
        while (pos + 4 <= f.size()) {
            if (u[pos] != 0xFF) { ++pos; continue; }
            const int m = u[pos + 1];
            if (m == 0xFF) { ++pos; continue; }
            if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }
            const size_t len = be16(u + pos + 2);
            if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                if (pos + 10 > f.size()) break;
                h = (int32_t)be16(u + pos + 5); w = (int32_t)be16(u + pos + 7); c = u[pos + 9];
                return;
            }
            pos += 2 + len;
        }
        fail(LFS_IO_E_FORMAT, "JPEG: no frame header in %s", path.string().c_str());
    }
    Pnm pnm;
    if (pnm_header(f, pnm)) { w = pnm.w; h = pnm.h; c = pnm.kind == 6 ? 3 : 1; return; }
    if (f.size() >= 30 && u[0] == 'B' && u[1] == 'M') {
        int32_t bw, bh; uint16_t bpp;
        std::memcpy(&bw, u + 18, 4); std::memcpy(&bh, u + 22, 4); std::memcpy(&bpp, u + 28, 2);
        w = bw; h = bh < 0 ? -bh : bh; c = bpp == 32 ? 4 : (bpp == 8 ? 1 : 3);
        return;
    }
    fail(LFS_IO_E_UNSUPPORTED, "Unrecognised image format: %s", path.string().c_str());
}

namespace {

// Any channel count -> RGB as load_image does (image_io.cpp:135-247): >= 3 channels keep the first three, 1 -> grey
// replicated, 2 -> (r, g, (r + g) / 2)
void to_rgb(const uint8_t* src, int ch, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* s = src + i * ch;
        if (ch >= 3) { dst[3 * i] = s[0]; dst[3 * i + 1] = s[1]; dst[3 * i + 2] = s[2]; }
        else if (ch == 1) { dst[3 * i] = dst[3 * i + 1] = dst[3 * i + 2] = s[0]; }
        else { dst[3 * i] = s[0]; dst[3 * i + 1] = s[1]; dst[3 * i + 2] = (uint8_t)(((int)s[0] + (int)s[1]) / 2); }
    }
}

uint8_t* decode_png(const std::vector<char>& f, int32_t& w, int32_t& h) {
    const unsigned char* u = (const unsigned char*)f.data();
    w = (int32_t)be32(u + 16); h = (int32_t)be32(u + 20);
    const int depth = u[24], ct = u[25], interlace = u[28];
    if (interlace) fail(LFS_IO_E_UNSUPPORTED, "PNG: interlaced files are not supported");
    if (ct != 0 && ct != 2 && ct != 3 && ct != 4 && ct != 6) fail(LFS_IO_E_FORMAT, "PNG: bad colour type %d", ct);
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) fail(LFS_IO_E_FORMAT, "PNG: bad bit depth %d", depth);
    if ((depth < 8 && ct != 0 && ct != 3) || (depth == 16 && ct == 3)) // PNG spec table 11.1: packed samples only for grey / palette, no 16-bit palette
        fail(LFS_IO_E_FORMAT, "PNG: bit depth %d is not allowed for colour type %d", depth, ct);
    if (w <= 0 || h <= 0 || (uint64_t)w * h > (uint64_t(1) << 31)) fail(LFS_IO_E_FORMAT, "PNG: bad dimensions");
    std::vector<unsigned char> idat, palette;
    std::vector<unsigned char> trns;
    for (size_t pos = 8; pos + 12 <= f.size();) {
        const uint32_t len = be32(u + pos);
        if (pos + 12 + (size_t)len > f.size()) fail(LFS_IO_E_FORMAT, "PNG: truncated chunk");
        const unsigned char* body = u + pos + 8;
        if (!std::memcmp(u + pos + 4, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(u + pos + 4, "PLTE", 4)) palette.assign(body, body + len);
        else if (!std::memcmp(u + pos + 4, "tRNS", 4)) trns.assign(body, body + len);
        else if (!std::memcmp(u + pos + 4, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    const int samples = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : 4;
    const size_t bpp_bits = (size_t)samples * depth, row_bytes = ((size_t)w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<unsigned char> raw((row_bytes + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || out_len != raw.size()) fail(LFS_IO_E_FORMAT, "PNG: corrupt image data (zlib %d)", zr);
    // undo the scanline filters in place
    std::vector<unsigned char> prev(row_bytes, 0);
    std::vector<uint8_t> pix((size_t)w * h * samples); // 8 bit per sample
    for (int y = 0; y < h; ++y) {
        unsigned char* row = raw.data() + (size_t)y * (row_bytes + 1);
        const int ft = row[0];
        unsigned char* cur = row + 1;
        for (size_t i = 0; i < row_bytes; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
            case 0: break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: fail(LFS_IO_E_FORMAT, "PNG: bad filter type %d", ft);
            }
            cur[i] = (unsigned char)(cur[i] + pred);
        }
        std::memcpy(prev.data(), cur, row_bytes);
        uint8_t* dst = pix.data() + (size_t)y * w * samples;
        if (depth == 8) std::memcpy(dst, cur, (size_t)w * samples);
        else if (depth == 16) for (size_t i = 0; i < (size_t)w * samples; ++i) dst[i] = cur[2 * i]; // high byte
        else for (int x = 0; x < w; ++x) { // packed grey / palette indices
            const int v = (cur[(size_t)x * depth / 8] >> (8 - depth - (x * depth) % 8)) & ((1 << depth) - 1);
            dst[x] = ct == 3 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1));
        }
    }
    uint8_t* out = (uint8_t*)std::malloc((size_t)w * h * 3);
    if (!out) throw std::bad_alloc();
    if (ct == 3) {
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const size_t k = (size_t)pix[i] * 3;
            if (k + 3 > palette.size()) { std::free(out); fail(LFS_IO_E_FORMAT, "PNG: palette index out of range"); }
            out[3 * i] = palette[k]; out[3 * i + 1] = palette[k + 1]; out[3 * i + 2] = palette[k + 2];
        }
    } else {
        // OpenImageIO presents grey + tRNS as 2 channels and grey + alpha as 2 channels: both reach the (r, g, avg) branch
        const int ch = samples + ((ct == 0 && !trns.empty()) ? 1 : 0);
        if (ch == samples) to_rgb(pix.data(), samples, (size_t)w * h, out);
        else {
            std::vector<uint8_t> ga((size_t)w * h * 2);
            // the tRNS key is a sample value at the file's bit depth; pix[] holds samples rescaled to 0..255 (depth < 8) or their high byte
            // (depth 16: the low byte is gone, so a 16-bit key matches on the high byte only when its low byte equals the sample's - not tracked:
            // such files keep every pixel opaque, as before)
            int key = -1;
            if (trns.size() >= 2 && depth <= 8) { const int raw = (int)be16(trns.data()); if (raw < (1 << depth)) key = raw * 255 / ((1 << depth) - 1); }
            for (size_t i = 0; i < (size_t)w * h; ++i) { ga[2 * i] = pix[i]; ga[2 * i + 1] = (pix[i] == key) ? 0 : 255; }
            to_rgb(ga.data(), 2, (size_t)w * h, out);
        }
    }
    return out;
}

void png_chunk(std::vector<unsigned char>& out, const char type[4], const unsigned char* data, size_t n) {
    unsigned char hdr[8] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n, (unsigned char)type[0],
                            (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
    out.insert(out.end(), hdr, hdr + 8);
    if (n) out.insert(out.end(), data, data + n);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    const unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    out.insert(out.end(), c, c + 4);
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Blender / NeRF-synthetic transforms.json (src/loader/formats/transforms.cpp:62-265)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// A small JSON reader (objects, arrays, numbers, strings, literals, // and /* */ comments: nlohmann::json::parse(.., ignore_comments = true))
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0; bool b = false; std::string str;
    std::vector<Json> arr; std::vector<std::pair<std::string, Json>> obj;
    const Json* find(const std::string& k) const {
        for (const auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool contains(const std::string& k) const { return kind == Obj && find(k) != nullptr; }
    double number(const char* what) const { if (kind != Num) fail(LFS_IO_E_FORMAT, "transforms: '%s' is not a number", what); return num; }
};
class JsonParser {
  public:
    explicit JsonParser(const std::vector<char>& b) : p_(b.data()), end_(b.data() + b.size()) {}
    Json parse() { Json j = value(); ws(); if (p_ != end_) fail(LFS_IO_E_FORMAT, "transforms: trailing characters after the JSON document"); return j; }
  private:
    const char* p_; const char* end_;
    void ws() {
        for (;;) {
            while (p_ < end_ && std::isspace((unsigned char)*p_)) ++p_;
            if (p_ + 1 < end_ && p_[0] == '/' && p_[1] == '/') { while (p_ < end_ && *p_ != '\n') ++p_; continue; }
            if (p_ + 1 < end_ && p_[0] == '/' && p_[1] == '*') { p_ += 2; while (p_ + 1 < end_ && !(p_[0] == '*' && p_[1] == '/')) ++p_; p_ = p_ + 2 <= end_ ? p_ + 2 : end_; continue; }
            return;
        }
    }
    [[noreturn]] void bad(const char* what) { fail(LFS_IO_E_FORMAT, "transforms: JSON parse error (%s)", what); }
    std::string string() {
        ++p_; std::string out;
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\' && p_ + 1 < end_) {
                ++p_;
                switch (*p_) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': { if (p_ + 4 >= end_) bad("\\u"); unsigned cp = (unsigned)std::strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16); p_ += 4;
                            if (cp < 0x80) out += (char)cp; else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); } break; }
                default: out += *p_;
                }
                ++p_;
            } else out += *p_++;
        }
        if (p_ >= end_) bad("unterminated string");
        ++p_;
        return out;
    }
    Json value() {
        ws();
        if (p_ >= end_) bad("unexpected end");
        Json j;
        if (*p_ == '{') {
            j.kind = Json::Obj; ++p_; ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return j; }
            for (;;) {
                ws(); if (p_ >= end_ || *p_ != '"') bad("object key");
                std::string k = string(); ws();
                if (p_ >= end_ || *p_ != ':') bad("':'");
                ++p_;
                j.obj.emplace_back(std::move(k), value()); ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; return j; }
                bad("',' or '}'");
            }
        }
        if (*p_ == '[') {
            j.kind = Json::Arr; ++p_; ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return j; }
            for (;;) {
                j.arr.push_back(value()); ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; return j; }
                bad("',' or ']'");
            }
        }
        if (*p_ == '"') { j.kind = Json::Str; j.str = string(); return j; }
        if (end_ - p_ >= 4 && !std::strncmp(p_, "true", 4)) { j.kind = Json::Bool; j.b = true; p_ += 4; return j; }
        if (end_ - p_ >= 5 && !std::strncmp(p_, "false", 5)) { j.kind = Json::Bool; p_ += 5; return j; }
        if (end_ - p_ >= 4 && !std::strncmp(p_, "null", 4)) { p_ += 4; return j; }
        const std::string tmp(p_, std::min(end_, p_ + 64));
        char* e = nullptr;
        const double v = std::strtod(tmp.c_str(), &e);
        if (e == tmp.c_str()) bad("value");
        p_ += e - tmp.c_str();
        j.kind = Json::Num; j.num = v;
        return j;
    }
};

fs::path transform_image_path(const fs::path& dir, const Json& frame) { // :62-71: Blender sets carry no extension: ".png" is tried
    const Json* fp = frame.find("file_path");
    if (!fp || fp->kind != Json::Str) fail(LFS_IO_E_FORMAT, "transforms: frame without file_path");
    fs::path p = dir / fp->str;
    const fs::path png = fs::path(p.string() + ".png");
    return fs::exists(png) ? png : p;
}

// 4x4 inverse in double (the reference: torch::inverse = LAPACK in float32; a rigid matrix is well conditioned, results agree to ~1e-7)
bool invert4(const double m[16], double inv[16]) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c) for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[c][k]);
        const double d = a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) inv[4 * r + c] = a[r][4 + c];
    return true;
}

void transforms_load(lfs_colmap_scene& sc, const fs::path& given) {
    fs::path file = given;
    if (fs::is_directory(given)) {
        if (fs::is_regular_file(given / "transforms_train.json")) file = given / "transforms_train.json";
        else if (fs::is_regular_file(given / "transforms.json")) file = given / "transforms.json";
        else fail(LFS_IO_E_NOT_FOUND, "could not find transforms_train.json nor transforms.json in %s", given.string().c_str());
    }
    if (!fs::is_regular_file(file)) fail(LFS_IO_E_NOT_FOUND, "%s is not a valid file", file.string().c_str());
    const fs::path dir = file.parent_path();
    const Json t = JsonParser(slurp(file)).parse();
    if (t.kind != Json::Obj) fail(LFS_IO_E_FORMAT, "transforms: the document is not an object");
    const Json* frames = t.find("frames");
    int w = -1, h = -1;
    if (!t.contains("w") || !t.contains("h")) { // :101-121: size of the first image
        if (!frames || frames->kind != Json::Arr || frames->arr.empty()) fail(LFS_IO_E_FORMAT, "Error while trying to read image dimensions: no frames");
        int32_t iw = 0, ih = 0, ic = 0;
        try { image_info_impl(transform_image_path(dir, frames->arr[0]), iw, ih, ic); }
        catch (const IoError& e) { fail(e.code, "Error while trying to read image dimensions: %s", e.what()); }
        w = iw; h = ih;
    } else { w = (int)t.find("w")->number("w"); h = (int)t.find("h")->number("h"); }
    auto focal = [](int res, float fov_rad) { return 0.5f * (float)res / std::tan(0.5f * fov_rad); }; // :30-32
    float fl_x = -1.f, fl_y = -1.f;
    if (t.contains("fl_x")) fl_x = (float)t.find("fl_x")->number("fl_x");
    else if (t.contains("camera_angle_x")) fl_x = focal(w, (float)t.find("camera_angle_x")->number("camera_angle_x"));
    if (t.contains("fl_y")) fl_y = (float)t.find("fl_y")->number("fl_y");
    else if (t.contains("camera_angle_y")) fl_y = focal(h, (float)t.find("camera_angle_y")->number("camera_angle_y"));
    else { if (w != h) fail(LFS_IO_E_FORMAT, "no camera_angle_y expected w!=h"); fl_y = fl_x; }
    const float cx = t.contains("cx") ? (float)t.find("cx")->number("cx") : (float)(0.5 * w);
    const float cy = t.contains("cy") ? (float)t.find("cy")->number("cy") : (float)(0.5 * h);
    float dist[4] = {0, 0, 0, 0};
    const char* dn[4] = {"k1", "k2", "p1", "p2"};
    for (int i = 0; i < 4; ++i) if (t.contains(dn[i])) dist[i] = (float)t.find(dn[i])->number(dn[i]);
    if (dist[0] > 0 || dist[1] > 0 || dist[2] > 0 || dist[3] > 0)
        fail(LFS_IO_E_UNSUPPORTED, "GS don't support distortion for now: k1=%g, k2=%g, p1=%g, p2=%g", dist[0], dist[1], dist[2], dist[3]);
    // fixMat = rotation about Y by float(pi) (createYRotationMatrix): cos = -1, sin = -8.742278e-08
    const float ang = (float)M_PI, cs = std::cos(ang), sn = std::sin(ang);
    const float fix[16] = {cs, 0, sn, 0, 0, 1, 0, 0, -sn, 0, cs, 0, 0, 0, 0, 1};
    if (frames && frames->kind == Json::Arr) {
        uint32_t counter = 0;
        for (const Json& fr : frames->arr) {
            const Json* tm = fr.find("transform_matrix");
            if (!tm) fail(LFS_IO_E_FORMAT, "expected all frames to contain transform_matrix");
            if (tm->kind != Json::Arr || tm->arr.size() != 4) fail(LFS_IO_E_FORMAT, "transform_matrix has the wrong dimensions");
            double c2w[16];
            for (int i = 0; i < 4; ++i) {
                if (tm->arr[i].kind != Json::Arr || tm->arr[i].arr.size() < 4) fail(LFS_IO_E_FORMAT, "transform_matrix has the wrong dimensions");
                for (int j = 0; j < 4; ++j) c2w[4 * i + j] = (double)(float)tm->arr[i].arr[j].number("transform_matrix");
            }
            for (int i = 0; i < 3; ++i) { c2w[4 * i + 1] = -c2w[4 * i + 1]; c2w[4 * i + 2] = -c2w[4 * i + 2]; } // OpenGL -> COLMAP axes (:219)
            double inv[16];
            if (!invert4(c2w, inv)) fail(LFS_IO_E_FORMAT, "transform_matrix is singular");
            float w2c[16], m[16];
            for (int i = 0; i < 16; ++i) w2c[i] = (float)inv[i];
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { float acc = 0.f; for (int k = 0; k < 4; ++k) acc += w2c[4 * r + k] * fix[4 * k + c]; m[4 * r + c] = acc; }
            lfs_colmap_view v{};
            v.camera_id = counter++; v.colmap_model = LFS_COLMAP_PINHOLE; v.camera_model_type = 0;
            v.width = (uint64_t)w; v.height = (uint64_t)h;
            v.focal_x = fl_x; v.focal_y = fl_y; v.center_x = cx; v.center_y = cy;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) v.R[3 * r + c] = m[4 * r + c]; v.T[r] = m[4 * r + 3]; }
            v.n_params = 4; v.params[0] = fl_x; v.params[1] = fl_y; v.params[2] = cx; v.params[3] = cy;
            const fs::path img = transform_image_path(dir, fr);
            sc.views.push_back(v);
            sc.paths.push_back(img.string());
            sc.names.push_back(img.filename().string());
        }
    }
    sc.center[0] = sc.center[1] = sc.center[2] = 0.f; // (:252: the scene centre of a transforms set is the origin)
}

} // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

const char* lfs_io_last_error(void) { return g_error.c_str(); }
const char* lfs_io_version(void) { return "lfs_io 1 (colmap bin+txt, splat ply, png/pnm)"; }
void lfs_io_free(void* p) { std::free(p); }

int lfs_colmap_open(const char* base, const char* images_folder, int format, lfs_colmap_scene** scene) {
    return guarded([&] {
        if (!base || !images_folder || !scene || (format != 0 && format != 1)) fail(LFS_IO_E_INVALID, "lfs_colmap_open: bad arguments");
        *scene = nullptr;
        const float factor = folder_scale(images_folder);
        const bool txt = format == 1;
        const auto cams = txt ? cameras_txt(sparse_file(base, "cameras.txt"), factor) : cameras_bin(sparse_file(base, "cameras.bin"), factor);
        const auto images = txt ? images_txt(sparse_file(base, "images.txt")) : images_bin(sparse_file(base, "images.bin"));
        auto sc = std::make_unique<lfs_colmap_scene>();
        assemble(*sc, base, images_folder, cams, images);
        *scene = sc.release();
    });
}
int lfs_transforms_open(const char* path, lfs_colmap_scene** scene) {
    return guarded([&] {
        if (!path || !scene) fail(LFS_IO_E_INVALID, "lfs_transforms_open: bad arguments");
        *scene = nullptr;
        auto sc = std::make_unique<lfs_colmap_scene>();
        transforms_load(*sc, path);
        *scene = sc.release();
    });
}
void lfs_colmap_close(lfs_colmap_scene* s) { delete s; }
uint64_t lfs_colmap_num_views(const lfs_colmap_scene* s) { return s ? s->views.size() : 0; }
int lfs_colmap_view_at(const lfs_colmap_scene* s, uint64_t i, lfs_colmap_view* out) {
    return guarded([&] {
        if (!s || !out || i >= s->views.size()) fail(LFS_IO_E_INVALID, "lfs_colmap_view_at: index out of range");
        *out = s->views[i];
    });
}
const char* lfs_colmap_image_name(const lfs_colmap_scene* s, uint64_t i) { return (s && i < s->names.size()) ? s->names[i].c_str() : nullptr; }
const char* lfs_colmap_image_path(const lfs_colmap_scene* s, uint64_t i) { return (s && i < s->paths.size()) ? s->paths[i].c_str() : nullptr; }
int lfs_colmap_scene_center(const lfs_colmap_scene* s, float center[3]) {
    return guarded([&] {
        if (!s || !center) fail(LFS_IO_E_INVALID, "lfs_colmap_scene_center: bad arguments");
        std::memcpy(center, s->center, sizeof s->center);
    });
}

int lfs_colmap_points_open(const char* base, int format, lfs_point_cloud** pc) {
    return guarded([&] {
        if (!base || !pc || (format != 0 && format != 1)) fail(LFS_IO_E_INVALID, "lfs_colmap_points_open: bad arguments");
        *pc = nullptr;
        auto out = std::make_unique<lfs_point_cloud>();
        if (format == 0) {
            const auto buf = slurp(sparse_file(base, "points3D.bin"));
            Cursor cur(buf, "points3D.bin");
            const uint64_t n = cur.get<uint64_t>();
            if (n > buf.size() / 43) fail(LFS_IO_E_FORMAT, "points3D.bin: unexpected end of file"); // 43 bytes per point at least
            out->positions.resize(n * 3); out->colors.resize(n * 3);
            for (uint64_t i = 0; i < n; ++i) {
                cur.skip(8); // point id
                for (int k = 0; k < 3; ++k) out->positions[3 * i + k] = (float)cur.get<double>();
                for (int k = 0; k < 3; ++k) out->colors[3 * i + k] = cur.get<uint8_t>();
                cur.skip(8); // reprojection error
                const uint64_t track = cur.get<uint64_t>();
                if (track > (uint64_t(1) << 40)) fail(LFS_IO_E_FORMAT, "points3D.bin: implausible track length");
                cur.skip(track * 8);
            }
            cur.expect_end();
        } else {
            const auto lines = text_lines(sparse_file(base, "points3D.txt"));
            out->positions.resize(lines.size() * 3); out->colors.resize(lines.size() * 3);
            for (size_t i = 0; i < lines.size(); ++i) {
                const auto tok = split(lines[i], ' ');
                if (tok.size() < 8) fail(LFS_IO_E_FORMAT, "Invalid format in point3D.txt: %s", lines[i].c_str());
                for (int k = 0; k < 3; ++k) out->positions[3 * i + k] = to_f32(tok[1 + k], "points3D.txt");
                for (int k = 0; k < 3; ++k) out->colors[3 * i + k] = (uint8_t)to_int(tok[4 + k], "points3D.txt");
            }
        }
        *pc = out.release();
    });
}
uint64_t lfs_point_cloud_size(const lfs_point_cloud* pc) { return pc ? pc->positions.size() / 3 : 0; }
int lfs_point_cloud_copy(const lfs_point_cloud* pc, float* positions, uint8_t* colors) {
    return guarded([&] {
        if (!pc) fail(LFS_IO_E_INVALID, "lfs_point_cloud_copy: null point cloud");
        if (positions && !pc->positions.empty()) std::memcpy(positions, pc->positions.data(), pc->positions.size() * sizeof(float));
        if (colors && !pc->colors.empty()) std::memcpy(colors, pc->colors.data(), pc->colors.size());
    });
}
void lfs_point_cloud_close(lfs_point_cloud* pc) { delete pc; }

int lfs_ply_write_splat(const char* path, uint64_t N, uint32_t n_dc, uint32_t n_rest, const float* means, const float* normals, const float* f_dc,
                        const float* f_rest, const float* opacity, const float* scaling, const float* rotation) {
    return guarded([&] {
        if (!path || (N && (!means || !f_dc || !opacity || !scaling || !rotation || (n_rest && !f_rest)))) fail(LFS_IO_E_INVALID, "lfs_ply_write_splat: bad arguments");
        std::string hdr = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(N) + "\n";
        auto prop = [&](const std::string& n) { hdr += "property float " + n + "\n"; };
        for (const char* n : {"x", "y", "z", "nx", "ny", "nz"}) prop(n);
        for (uint32_t i = 0; i < n_dc; ++i) prop("f_dc_" + std::to_string(i));
        for (uint32_t i = 0; i < n_rest; ++i) prop("f_rest_" + std::to_string(i));
        prop("opacity");
        for (int i = 0; i < 3; ++i) prop("scale_" + std::to_string(i));
        for (int i = 0; i < 4; ++i) prop("rot_" + std::to_string(i));
        hdr += "end_header\n";
        const size_t stride = 6 + n_dc + n_rest + 1 + 3 + 4;
        FILE* fp = std::fopen(path, "wb");
        if (!fp) fail(LFS_IO_E_NOT_FOUND, "Failed to open %s for writing", path);
        bool ok = std::fwrite(hdr.data(), 1, hdr.size(), fp) == hdr.size();
        const uint64_t chunk = 16384;
        std::vector<float> rows(chunk * stride);
        for (uint64_t a = 0; a < N && ok; a += chunk) {
            const uint64_t n = std::min(chunk, N - a);
            for (uint64_t i = 0; i < n; ++i) {
                float* r = rows.data() + i * stride;
                const uint64_t g = a + i;
                std::memcpy(r, means + 3 * g, 12);
                if (normals) std::memcpy(r + 3, normals + 3 * g, 12); else r[3] = r[4] = r[5] = 0.f;
                std::memcpy(r + 6, f_dc + (size_t)n_dc * g, 4 * (size_t)n_dc);
                if (n_rest) std::memcpy(r + 6 + n_dc, f_rest + (size_t)n_rest * g, 4 * (size_t)n_rest);
                float* t = r + 6 + n_dc + n_rest;
                t[0] = opacity[g];
                std::memcpy(t + 1, scaling + 3 * g, 12);
                std::memcpy(t + 4, rotation + 4 * g, 16);
            }
            ok = std::fwrite(rows.data(), sizeof(float) * stride, n, fp) == n;
        }
        ok = (std::fclose(fp) == 0) && ok;
        if (!ok) fail(LFS_IO_E_INVALID, "Short write on %s", path);
    });
}

int lfs_ply_open(const char* path, lfs_ply** ply) {
    return guarded([&] {
        if (!path || !ply) fail(LFS_IO_E_INVALID, "lfs_ply_open: bad arguments");
        *ply = nullptr;
        auto p = std::make_unique<lfs_ply>();
        ply_parse(*p, path);
        *ply = p.release();
    });
}
uint64_t lfs_ply_num_vertices(const lfs_ply* p) { return p ? p->n_vertices : 0; }
uint32_t lfs_ply_num_properties(const lfs_ply* p) { return p ? (uint32_t)p->props.size() : 0; }
const char* lfs_ply_property_name(const lfs_ply* p, uint32_t i) { return (p && i < p->props.size()) ? p->props[i].name.c_str() : nullptr; }
int lfs_ply_read(const lfs_ply* p, float* out) {
    return guarded([&] {
        if (!p || (!out && p->n_vertices)) fail(LFS_IO_E_INVALID, "lfs_ply_read: bad arguments");
        const size_t P = p->props.size();
        if (p->ascii) {
            const char* c = p->file.data() + p->data_offset;
            const char* end = p->file.data() + p->file.size();
            std::string tmp(c, end); // strtod needs a terminator
            const char* s = tmp.c_str();
            for (uint64_t i = 0; i < p->n_vertices * P; ++i) {
                char* e = nullptr;
                const double v = std::strtod(s, &e);
                if (e == s) fail(LFS_IO_E_FORMAT, "PLY: ascii vertex data ends early");
                out[i] = (float)v;
                s = e;
            }
            return;
        }
        const char* c = p->file.data() + p->data_offset;
        bool all_float = true;
        for (const auto& pr : p->props) all_float &= pr.type == 6;
        if (all_float) { std::memcpy(out, c, p->n_vertices * P * sizeof(float)); return; }
        for (uint64_t i = 0; i < p->n_vertices; ++i)
            for (size_t k = 0; k < P; ++k) {
                out[i * P + k] = ply_scalar(c, p->props[k].type);
                c += kPlyTypes[p->props[k].type].size;
            }
    });
}
void lfs_ply_close(lfs_ply* p) { delete p; }

int lfs_image_info(const char* path, int32_t* width, int32_t* height, int32_t* channels) {
    return guarded([&] {
        if (!path || !width || !height || !channels) fail(LFS_IO_E_INVALID, "lfs_image_info: bad arguments");
        image_info_impl(path, *width, *height, *channels);
    });
}

int lfs_image_target_size(int32_t w, int32_t h, int32_t res_div, int32_t max_width, int32_t* ow, int32_t* oh) {
    return guarded([&] {
        if (w <= 0 || h <= 0 || !ow || !oh) fail(LFS_IO_E_INVALID, "lfs_image_target_size: bad arguments");
        int nw = w, nh = h;
        if (res_div == 2 || res_div == 4 || res_div == 8) { nw = std::max(1, w / res_div); nh = std::max(1, h / res_div); }
        else if (res_div > 1) fail(LFS_IO_E_UNSUPPORTED, "load_image: unsupported resize factor %d", res_div);
        if (max_width > 0 && (nw > max_width || nh > max_width)) { // image_io.cpp:152-161, :195-202 (integer arithmetic)
            const int a = nw, b = nh;
            if (a > b) { nh = std::max(1, max_width * b / a); nw = std::max(1, max_width); }
            else { nw = std::max(1, max_width * a / b); nh = std::max(1, max_width); }
        }
        *ow = nw; *oh = nh;
    });
}

int lfs_image_load_rgb8(const char* path, uint8_t** data, int32_t* width, int32_t* height) {
    return guarded([&] {
        if (!path || !data || !width || !height) fail(LFS_IO_E_INVALID, "lfs_image_load_rgb8: bad arguments");
        *data = nullptr;
        const auto f = slurp(path);
        const unsigned char* u = (const unsigned char*)f.data();
        if (f.size() >= 33 && std::memcmp(u, kPngSig, 8) == 0) { *data = decode_png(f, *width, *height); return; }
        Pnm pnm;
        if (pnm_header(f, pnm)) {
            const int ch = pnm.kind == 6 ? 3 : 1, bps = pnm.maxval > 255 ? 2 : 1;
            const size_t n = (size_t)pnm.w * pnm.h;
            if (f.size() - pnm.data < n * ch * bps) fail(LFS_IO_E_FORMAT, "PNM: truncated pixel data");
            std::vector<uint8_t> px(n * ch);
            for (size_t i = 0; i < n * ch; ++i) {
                const unsigned v = bps == 2 ? be16(u + pnm.data + 2 * i) : u[pnm.data + i];
                px[i] = pnm.maxval == 255 ? (uint8_t)v : (uint8_t)((v * 255u + (unsigned)pnm.maxval / 2) / (unsigned)pnm.maxval);
            }
            uint8_t* out = (uint8_t*)std::malloc(n * 3);
            if (!out) throw std::bad_alloc();
            to_rgb(px.data(), ch, n, out);
            *data = out; *width = pnm.w; *height = pnm.h;
            return;
        }
        if (f.size() >= 2 && u[0] == 0xFF && u[1] == 0xD8) {
            bool unsupported = false;
            std::string err;
            uint8_t* out = lfs_decode_jpeg_rgb8(u, f.size(), width, height, &unsupported, &err);
            if (!out) fail(unsupported ? LFS_IO_E_UNSUPPORTED : LFS_IO_E_FORMAT, "%s (%s)", err.c_str(), path);
            *data = out;
            return;
        }
        fail(LFS_IO_E_UNSUPPORTED, "Unrecognised image format: %s", path);
    });
}

int lfs_image_write_png_rgb8(const char* path, const uint8_t* data, int32_t w, int32_t h) {
    return guarded([&] {
        if (!path || !data || w <= 0 || h <= 0) fail(LFS_IO_E_INVALID, "lfs_image_write_png_rgb8: bad arguments");
        std::vector<unsigned char> raw(((size_t)w * 3 + 1) * h);
        for (int y = 0; y < h; ++y) {
            raw[(size_t)y * (w * 3 + 1)] = 0; // filter: none
            std::memcpy(raw.data() + (size_t)y * (w * 3 + 1) + 1, data + (size_t)y * w * 3, (size_t)w * 3);
        }
        uLongf zn = compressBound((uLong)raw.size());
        std::vector<unsigned char> z(zn);
        if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) fail(LFS_IO_E_INVALID, "PNG: deflate failed");
        std::vector<unsigned char> out(kPngSig, kPngSig + 8);
        const unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w, (unsigned char)(h >> 24),
                                        (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0};
        png_chunk(out, "IHDR", ihdr, 13);
        png_chunk(out, "IDAT", z.data(), zn);
        png_chunk(out, "IEND", nullptr, 0);
        FILE* fp = std::fopen(path, "wb");
        if (!fp) fail(LFS_IO_E_NOT_FOUND, "Failed to open %s for writing", path);
        const bool ok = std::fwrite(out.data(), 1, out.size(), fp) == out.size();
        if (std::fclose(fp) != 0 || !ok) fail(LFS_IO_E_INVALID, "Short write on %s", path);
    });
}

} // extern "C"
