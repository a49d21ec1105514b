// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. Stand-in for the reference's include/core/image_io.hpp (OpenImageIO, not in this image): the COLMAP reader only asks
// for the size of the first image (ref_colmap_shim.cpp answers from the PNG / PNM header); core/camera.cpp also names load_image / free_image in
// load_and_get_image, which no test calls (ref_raster_shim.cpp defines them as throwing).
#pragma once
#include <filesystem>
#include <tuple>
std::tuple<int, int, int> get_image_info(std::filesystem::path p);
std::tuple<unsigned char*, int, int, int> load_image(std::filesystem::path p, int res_div = -1, int max_width = 3840);
void free_image(unsigned char* image);
