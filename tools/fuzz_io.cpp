// Mutation fuzzer for the untrusted-input parsers of liblfs_io (PNG / JPEG / PNM / BMP headers, PLY, COLMAP bin / txt, transforms json).
// Build and run (AddressSanitizer + UBSan; not part of the product):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude tools/fuzz_io.cpp \
//       lichtfeld-studio_amd/csrc_host/lfs_io.cpp lichtfeld-studio_amd/csrc_host/lfs_jpeg.cpp -lz -o /tmp/fuzz_io
//   /tmp/fuzz_io <iterations> <seed> <work dir> file1 file2 ...
// Each seed file is mutated (byte flips, truncation, splices, length-field pokes) and fed to every entry point that could open it; any
// crash / sanitizer report is a bug, error returns are the expected outcome.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "lfs_io.h"
namespace fs = std::filesystem;

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const fs::path& p, const std::vector<uint8_t>& d) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    f.write((const char*)d.data(), (std::streamsize)d.size());
}

static void mutate(std::vector<uint8_t>& d, std::mt19937_64& rng) {
    if (d.empty()) return;
    const int kind = (int)(rng() % 7);
    auto pos = [&]() { return (size_t)(rng() % d.size()); };
    switch (kind) {
    case 0: for (int k = 0, n = 1 + (int)(rng() % 8); k < n; ++k) d[pos()] ^= (uint8_t)(1u << (rng() % 8)); break;
    case 1: for (int k = 0, n = 1 + (int)(rng() % 4); k < n; ++k) d[pos()] = (uint8_t)rng(); break;
    case 2: d.resize(pos()); break;                                            // truncate
    case 3: { size_t a = pos(), n = (size_t)(rng() % 64); if (a + n > d.size()) n = d.size() - a; d.erase(d.begin() + (long)a, d.begin() + (long)(a + n)); break; }
    case 4: { size_t a = pos(), b = pos(), n = (size_t)(rng() % 64); for (size_t i = 0; i < n && a + i < d.size() && b + i < d.size(); ++i) d[a + i] = d[b + i]; break; }
    case 5: { static const uint8_t poke[] = {0x00, 0xff, 0x7f, 0x80, 0x01, 0xfe}; size_t a = pos(); for (int k = 0; k < 4 && a + k < d.size(); ++k) d[a + k] = poke[rng() % 6]; break; }
    default: { size_t a = pos(); std::vector<uint8_t> ins(1 + rng() % 32); for (auto& x : ins) x = (uint8_t)rng(); d.insert(d.begin() + (long)a, ins.begin(), ins.end()); break; }
    }
}

static void run_image(const fs::path& p) {
    int32_t w = 0, h = 0, c = 0;
    lfs_image_info(p.string().c_str(), &w, &h, &c);
    uint8_t* data = nullptr;
    if (lfs_image_load_rgb8(p.string().c_str(), &data, &w, &h) == LFS_IO_OK) {
        volatile uint8_t sink = 0;
        if (w > 0 && h > 0) sink = data[(size_t)w * h * 3 - 1];   // the whole buffer must be there
        (void)sink;
        lfs_io_free(data);
    }
}
static void run_ply(const fs::path& p) {
    lfs_ply* ply = nullptr;
    if (lfs_ply_open(p.string().c_str(), &ply) != LFS_IO_OK) return;
    const uint64_t n = lfs_ply_num_vertices(ply);
    const uint32_t np = lfs_ply_num_properties(ply);
    for (uint32_t i = 0; i < np; ++i) (void)lfs_ply_property_name(ply, i);
    if (n * np < (1u << 24)) {
        std::vector<float> out((size_t)(n * np) + 1);
        lfs_ply_read(ply, out.data());
    }
    lfs_ply_close(ply);
}
static void run_colmap(const fs::path& base, int format) {
    lfs_colmap_scene* sc = nullptr;
    if (lfs_colmap_open(base.string().c_str(), "images", format, &sc) == LFS_IO_OK) {
        lfs_colmap_view v;
        for (uint64_t i = 0, n = lfs_colmap_num_views(sc); i < n; ++i) { lfs_colmap_view_at(sc, i, &v); (void)lfs_colmap_image_name(sc, i); (void)lfs_colmap_image_path(sc, i); }
        float c[3];
        lfs_colmap_scene_center(sc, c);
        lfs_colmap_close(sc);
    }
    lfs_point_cloud* pc = nullptr;
    if (lfs_colmap_points_open(base.string().c_str(), format, &pc) == LFS_IO_OK) {
        const uint64_t n = lfs_point_cloud_size(pc);
        std::vector<float> xyz(3 * n + 1);
        std::vector<uint8_t> rgb(3 * n + 1);
        lfs_point_cloud_copy(pc, xyz.data(), rgb.data());
        lfs_point_cloud_close(pc);
    }
}
static void run_transforms(const fs::path& p) {
    lfs_colmap_scene* sc = nullptr;
    if (lfs_transforms_open(p.string().c_str(), &sc) == LFS_IO_OK) lfs_colmap_close(sc);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s iterations seed workdir files...\n", argv[0]); return 2; }
    const long iters = atol(argv[1]);
    std::mt19937_64 rng((uint64_t)atoll(argv[2]));
    const fs::path work = argv[3];
    fs::create_directories(work / "scene" / "sparse" / "0");
    fs::create_directories(work / "scene" / "images");
    std::vector<std::pair<std::string, std::vector<uint8_t>>> seeds;
    for (int i = 4; i < argc; ++i) seeds.push_back({fs::path(argv[i]).filename().string(), slurp(argv[i])});
    long ok = 0;
    for (long it = 0; it < iters; ++it) {
        const auto& [name, bytes] = seeds[rng() % seeds.size()];
        std::vector<uint8_t> d = bytes;
        for (int k = 0, n = 1 + (int)(rng() % 3); k < n; ++k) mutate(d, rng);
        const std::string ext = fs::path(name).extension().string();
        if (ext == ".ply") { spit(work / "m.ply", d); run_ply(work / "m.ply"); }
        else if (ext == ".json") { spit(work / "transforms.json", d); run_transforms(work / "transforms.json"); }
        else if (ext == ".bin" || ext == ".txt") {
            // mutate one file of the triple, keep the others pristine
            for (const auto& [n2, b2] : seeds)
                if (fs::path(n2).extension() == ext) spit(work / "scene" / "sparse" / "0" / n2, n2 == name ? d : b2);
            run_colmap(work / "scene", ext == ".txt");
        } else { spit(work / ("m" + ext), d); run_image(work / ("m" + ext)); }
        ++ok;
    }
    printf("fuzz_io: %ld mutated inputs, no crash\n", ok);
    return 0;
}
