// Baseline JPEG -> 8-bit grey image.  The reference decodes the compressed frames it is handed with OpenCV
// (cv::imdecode(..., IMREAD_GRAYSCALE) for LpSlamImageFormat_8UC1_JPEPG frames, /root/reference/src/Manager/SlamManager.cpp:1139-1146;
// cv::imdecode(..., IMREAD_UNCHANGED) for the records its recorder wrote with cv::imencode(".jpg"), src/Manager/ReplayEngine.cpp:123,
// src/Manager/RecordEngine.cpp:93), i.e. with libjpeg: Huffman-coded sequential DCT, 8 bits, the "islow" integer inverse DCT.  This is
// that decoder for the grey output the trackers consume: a one-component file gives its samples, a YCbCr file its luma plane (what
// libjpeg delivers for JCS_GRAYSCALE output, the IMREAD_GRAYSCALE path).  Bit for bit libjpeg's samples (tests/test_jpeg_cpu.py compares
// with Pillow = libjpeg-turbo).  Not supported, as stated errors: progressive / arithmetic / lossless / 12-bit files.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "core.h"

namespace LpSlam {

bool decode_jpeg_gray(const uint8_t* data, size_t size, GrayImage& out, std::string* why = nullptr);
// what cv::imencode(".jpg", grey) writes (LpSlamManager::compressImage, src/InterfaceImpl/LpSlamManager.cpp:133-152): one component,
// baseline, Annex K tables scaled for `quality` (OpenCV's default: 95), libjpeg's islow forward DCT and rounding
bool encode_jpeg_gray(const GrayImage& img, int quality, std::vector<uint8_t>& out);
inline bool looks_like_jpeg(const uint8_t* data, size_t size) { return size >= 4 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF; }

}  // namespace LpSlam
