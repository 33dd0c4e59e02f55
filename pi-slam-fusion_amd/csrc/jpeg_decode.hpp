// jpeg_decode.hpp -- cv::imread's JPEG leg for the file driver (backup/map2dfusion.cpp:129-132); see jpeg_decode.cpp
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pf {
void set_error(const std::string& msg);          // fusion_map.cpp: what pf_last_error() returns
bool jpeg_info(const uint8_t* data, size_t len, int* rows, int* cols, int* comps);
// 8-bit BGR, rows x cols as jpeg_info reports them, `stride` bytes per row
bool jpeg_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols, size_t stride);
bool read_file_bytes(const char* filename, std::vector<uint8_t>& out);
bool read_image_file(const char* filename, std::vector<uint8_t>& bgr, int* rows, int* cols);
}  // namespace pf
