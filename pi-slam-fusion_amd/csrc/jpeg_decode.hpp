// jpeg_decode.hpp -- cv::imread's JPEG leg for the file driver (backup/map2dfusion.cpp:129-132); see jpeg_decode.cpp
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pf {
void set_error(const std::string& msg);          // fusion_map.cpp: what pf_last_error() returns (per thread)
const char* last_error();
bool jpeg_info(const uint8_t* data, size_t len, int* rows, int* cols, int* comps);
// 8-bit BGR, rows x cols as jpeg_info reports them, `stride` bytes per row
bool jpeg_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols, size_t stride);

// The entropy stage alone (parse + Huffman, sequential or progressive): what the device back end (jpeg_device.hip) starts from.
struct JpegComponent {
    int h, v;                    // sampling factors
    int bw, bh;                  // blocks held: whole MCUs
    int w, ht;                   // real samples (libjpeg's downsampled_width / _height)
    uint16_t q[64];              // quantiser, natural order
    size_t coef_off;             // first coefficient of the component in the store, in int16 units
};
struct JpegFrame {
    int rows, cols, ncomp, hmax, vmax, mcux, mcuy;
    bool ycc;                    // the three components are Y, Cb, Cr (else R, G, B)
    JpegComponent c[3];
    size_t coef_count;           // int16 values the store must hold
};
// geometry only (quantisers are not final before the scans have been read)
bool jpeg_frame_info(const uint8_t* data, size_t len, JpegFrame& f);
// coefficients of every block, natural order, 64 per block, blocks row-major per component, components one after another in `store`
bool jpeg_entropy_decode(const uint8_t* data, size_t len, JpegFrame& f, int16_t* store, size_t store_cap);
struct HuffParPlan;            // jpeg_huff_par.hpp
// the plan of a parallel Huffman pass over the stream's one scan, and the scan's bytes with the stuffing removed; false when the stream is not of that kind
// (seg_end: where the ends of the restart intervals' segments go; nullptr: streams with a restart interval are refused)
bool jpeg_scan_plan(const uint8_t* data, size_t len, JpegFrame& f, HuffParPlan& P, uint8_t* bits, size_t cap, size_t* nbytes, std::vector<uint32_t>* seg_end = nullptr);
// PNG (png_decode.cpp): cv::imread's default conversion to three 8-bit channels, BGR; non-interlaced files
bool png_info(const uint8_t* data, size_t len, int* rows, int* cols);
bool png_decode_bgr(const uint8_t* data, size_t len, uint8_t* bgr, int rows, int cols, size_t stride);
bool read_file_bytes(const char* filename, std::vector<uint8_t>& out);
bool read_image_file(const char* filename, std::vector<uint8_t>& bgr, int* rows, int* cols);
}  // namespace pf
