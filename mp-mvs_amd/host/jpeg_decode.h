// jpeg_decode.h -- dependency-free JPEG reader for the image-ingest row.
//
// The reference loads its inputs with cv::imread (src/PatchMatch.cpp:877 IMREAD_GRAYSCALE,
// :324 IMREAD_COLOR), i.e. through libjpeg.  No JPEG library headers exist in this image, so
// the decoder is written out here; its arithmetic follows libjpeg's published algorithms
// (integer "islow" inverse DCT, "fancy" triangle chroma upsampling, 16-bit fixed-point
// YCbCr->RGB), so the pixels equal libjpeg's bit for bit -- which matters because the 8-bit
// texture format and the NCC costs depend on the exact grey values.
#ifndef MPMVS_HOST_JPEG_DECODE_H_
#define MPMVS_HOST_JPEG_DECODE_H_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

// channels = 1: luminance plane (what IMREAD_GRAYSCALE returns for a YCbCr or grey file);
// channels = 3: interleaved B,G,R (cv::Vec3b order).  Supports 8-bit Huffman-coded baseline,
// extended-sequential and progressive files with 1 or 3 components, restart intervals, any
// sampling factors up to 4; an EXIF orientation tag is applied, as cv::imread does.  Returns false and fills err otherwise.
bool DecodeJpeg(const uint8_t* data, size_t size, int channels, std::vector<uint8_t>& pixels, int& width, int& height, std::string& err);
bool DecodeJpegFile(const std::string& path, int channels, std::vector<uint8_t>& pixels, int& width, int& height, std::string& err);

#endif
