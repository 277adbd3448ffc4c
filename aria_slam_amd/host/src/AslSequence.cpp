// See aria_hip/AslSequence.hpp.
#include "aria_hip/AslSequence.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace aria::io {

namespace {

inline std::uint32_t be32(const std::uint8_t* p) { return ((std::uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool file_exists(const std::string& p) {
    std::ifstream f(p);
    return f.good();
}

}  // namespace

void decode_png_gray(const std::vector<std::uint8_t>& file, std::vector<std::uint8_t>& gray, int& width, int& height) {
    static const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("png: bad signature");
    std::size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<std::uint8_t> idat;
    bool done = false;
    while (!done && pos + 12 <= file.size()) {
        const std::uint32_t len = be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if (pos + 12 + len > file.size()) throw std::runtime_error("png: truncated chunk");
        const std::uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) throw std::runtime_error("png: bad IHDR");
            w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384) throw std::runtime_error("png: bad size");
    if (depth != 8 || interlace != 0) throw std::runtime_error("png: only 8-bit non-interlaced images are supported");
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: throw std::runtime_error("png: unsupported colour type");
    }
    const std::size_t stride = (std::size_t)w * ch;
    std::vector<std::uint8_t> raw((stride + 1) * (std::size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error("png: inflate failed");
    // undo the per-scanline filters in place (PNG spec section 9)
    std::vector<std::uint8_t> prev(stride, 0), cur(stride);
    gray.resize((std::size_t)w * h);
    for (int y = 0; y < h; y++) {
        const std::uint8_t* src = &raw[(stride + 1) * y];
        const int ft = src[0];
        for (std::size_t i = 0; i < stride; i++) {
            const int a = i >= (std::size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (std::size_t)ch ? prev[i - ch] : 0;
            int v = src[1 + i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error("png: bad filter type");
            }
            cur[i] = (std::uint8_t)v;
        }
        std::uint8_t* g = &gray[(std::size_t)y * w];
        if (ch <= 2) {
            for (int x = 0; x < w; x++) g[x] = cur[(std::size_t)x * ch];
        } else {
            // cv::cvtColor RGB -> gray, 8-bit path: (R*4899 + G*9617 + B*1868 + 8192) >> 14
            for (int x = 0; x < w; x++) {
                const std::uint8_t* p = &cur[(std::size_t)x * ch];
                g[x] = (std::uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
            }
        }
        prev.swap(cur);
    }
    width = w;
    height = h;
}

bool AslSequence::load(const std::string& dataset_path) {
    images_.clear();
    std::string cam = dataset_path + "/mav0/cam0";
    if (!file_exists(cam + "/data.csv")) cam = dataset_path + "/cam0";        // path already points at mav0
    std::ifstream file(cam + "/data.csv");
    if (!file.is_open()) return false;
    std::string line;
    std::getline(file, line);                                                 // header (EuRoCReader.cpp:78-79)
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;                         // :82
        std::stringstream ss(line);
        std::string ts, name;
        std::getline(ss, ts, ',');
        std::getline(ss, name, ',');
        const auto b = name.find_first_not_of(" \t");
        if (b == std::string::npos) continue;
        name.erase(0, b);
        name.erase(name.find_last_not_of(" \t\r\n") + 1);                     // :90-91
        AslImage img;
        img.timestamp = std::strtod(ts.c_str(), nullptr) * 1e-9;              // nanoseconds -> seconds
        img.path = cam + "/data/" + name;
        images_.push_back(img);
    }
    std::stable_sort(images_.begin(), images_.end(), [](const AslImage& a, const AslImage& b) { return a.timestamp < b.timestamp; });
    return !images_.empty();                                                  // :105
}

void AslSequence::read(std::size_t i, std::vector<std::uint8_t>& gray, int& width, int& height) const {
    const AslImage& im = images_.at(i);
    std::ifstream f(im.path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + im.path);
    std::vector<std::uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    decode_png_gray(bytes, gray, width, height);
}

}  // namespace aria::io

// ---- plain-C hooks so tests (ctypes) can exercise the reader without a C++ harness ----
extern "C" {

int aria_asl_decode_png_gray(const std::uint8_t* bytes, std::size_t n, std::uint8_t* out, std::size_t cap, int* width, int* height) {
    try {
        std::vector<std::uint8_t> file(bytes, bytes + n), gray;
        int w = 0, h = 0;
        aria::io::decode_png_gray(file, gray, w, h);
        if (width) *width = w;
        if (height) *height = h;
        if (gray.size() > cap) return -5;
        std::memcpy(out, gray.data(), gray.size());
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// Writes up to cap timestamps (seconds) in reader order; returns the number of images or -1.
int aria_asl_list(const char* dataset_path, double* timestamps, int cap, char* first_path, int first_path_cap) {
    aria::io::AslSequence s;
    if (!s.load(dataset_path)) return -1;
    for (std::size_t i = 0; i < s.size() && (int)i < cap; i++) timestamps[i] = s.at(i).timestamp;
    if (first_path && first_path_cap > 0) std::snprintf(first_path, (std::size_t)first_path_cap, "%s", s.at(0).path.c_str());
    return (int)s.size();
}

}  // extern "C"
