// ORACLE/_ref - TEST INFRASTRUCTURE ONLY. Stand-in for the reference's include/core/image_io.hpp (OpenImageIO, not in this image): the COLMAP reader only asks
// for the size of the first image; ref_colmap_shim.cpp answers from the PNG / PNM header.
#pragma once
#include <filesystem>
#include <tuple>
std::tuple<int, int, int> get_image_info(std::filesystem::path p);
