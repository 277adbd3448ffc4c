// Reader for the on-disk format either side of the hot path: the ASL / EuRoC MAV layout
//     <root>/mav0/cam0/data.csv      "#timestamp [ns],filename" then "1403636579763555584,1403636579763555584.png"
//     <root>/mav0/cam0/data/*.png    8-bit grayscale 752x480
// as the reference reads it (src/legacy/EuRoCReader.cpp:23-27, 70-108: skip header/comment lines, split at the
// first comma, trim, sort by timestamp; :277-309: imread(IMREAD_GRAYSCALE)). OpenCV's imgcodecs is replaced by a
// dependency-free PNG decoder (zlib inflate only): 8-bit, non-interlaced, colour types 0 (gray), 2 (RGB), 4 (gray+a),
// 6 (RGBA); colour is converted with OpenCV's BGR2GRAY fixed-point weights. IMU / ground truth are out of scope.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace aria::io {

struct AslImage {
    double timestamp = 0.0;          // seconds (EuRoCReader::parseTimestamp: ns * 1e-9)
    std::string path;
};

class AslSequence {
public:
    // dataset_path may be the sequence root (containing mav0/) or the mav0 directory itself
    bool load(const std::string& dataset_path);
    std::size_t size() const { return images_.size(); }
    const AslImage& at(std::size_t i) const { return images_[i]; }
    // Decodes image i to 8-bit grayscale, row-major, tightly packed. Throws std::runtime_error on a bad file.
    void read(std::size_t i, std::vector<std::uint8_t>& gray, int& width, int& height) const;

private:
    std::vector<AslImage> images_;
};

// PNG -> 8-bit grayscale (see header comment). Throws std::runtime_error.
void decode_png_gray(const std::vector<std::uint8_t>& file, std::vector<std::uint8_t>& gray, int& width, int& height);

}  // namespace aria::io
