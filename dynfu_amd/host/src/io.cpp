// io.cpp — depth PNG reader, PCD / VTK writers, sequence listing (dfa_host/io.hpp).
#include <dfa_host/io.hpp>

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>

namespace dfa {

PolygonMesh convertToMesh(const std::vector<PointXYZ>& triangles) {  // kinfu.cpp:236-260
    PolygonMesh mesh;
    if (triangles.empty()) return mesh;
    mesh.cloud.points = triangles;
    mesh.polygons.resize(triangles.size() / 3);
    for (size_t i = 0; i < mesh.polygons.size(); ++i)
        mesh.polygons[i] = {(uint32_t)(i * 3 + 0), (uint32_t)(i * 3 + 2), (uint32_t)(i * 3 + 1)};
    return mesh;
}

namespace io {
namespace {

[[noreturn]] void bad(const std::string& what) { throw Error(1 /* DFA_ERR_INVALID */, "depth png: " + what); }

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)), v.push_back((uint8_t)(x >> 16)), v.push_back((uint8_t)(x >> 8)), v.push_back((uint8_t)x);
}

const uint8_t PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

int paeth(int a, int b, int c) {  // PNG specification, filter type 4
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at)));
}

std::string fmt_g(float v, int digits) {  // operator<< of a float on a classic-locale stream with this precision
    if (std::isnan(v)) return "nan";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*g", digits, (double)v);
    return buf;
}

}  // namespace

DepthImage decodeDepthPng(const uint8_t* b, size_t n) {
    if (n < 8 || std::memcmp(b, PNG_SIG, 8) != 0) bad("not a PNG file");
    size_t at = 8;
    bool have_ihdr = false, have_iend = false;
    uint32_t width = 0, height = 0;
    int depth = 0;
    std::vector<uint8_t> idat;
    while (!have_iend) {
        if (at + 12 > n) bad("truncated file");
        const uint32_t len = be32(b + at);
        if (len > n - at - 12) bad("chunk runs past the end of the file");
        const uint8_t* type = b + at + 4;
        const uint8_t* data = b + at + 8;
        if ((uint32_t)crc32(0L, type, (uInt)(len + 4)) != be32(data + len)) bad("chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13 || have_ihdr) bad("bad IHDR");
            width = be32(data), height = be32(data + 4), depth = data[8];
            if (data[9] != 0) bad("not a greyscale image (colour type " + std::to_string(data[9]) + ")");
            if (depth != 16 && depth != 8) bad("bit depth " + std::to_string(depth) + " (a depth frame has 16)");
            if (data[10] != 0 || data[11] != 0) bad("unknown compression / filter method");
            if (data[12] != 0) bad("interlaced images are not supported");
            if (width == 0 || height == 0 || width > 16384 || height > 16384) bad("unreasonable dimensions");
            have_ihdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (!have_ihdr) bad("IDAT before IHDR");
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            have_iend = true;
        } else if (!(type[0] & 0x20)) {
            bad(std::string("unknown critical chunk ") + std::string((const char*)type, 4));
        }
        at += 12 + (size_t)len;
    }
    if (!have_ihdr || idat.empty()) bad("no image data");
    const size_t bpp = (size_t)depth / 8, stride = (size_t)width * bpp;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf got = (uLongf)raw.size();
    const int rc = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (rc != Z_OK || got != raw.size()) bad("inflate failed or wrong image size");
    DepthImage img;
    img.cols = (int)width, img.rows = (int)height;
    img.data.resize((size_t)width * height);
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t* line = raw.data() + (stride + 1) * y;
        const int ft         = line[0];
        if (ft > 4) bad("unknown filter type");
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? up : ft == 3 ? (a + up) / 2 : paeth(a, up, c);
            cur[i] = (uint8_t)(line[1 + i] + pred);
        }
        uint16_t* out = img.data.data() + (size_t)width * y;
        for (uint32_t x = 0; x < width; ++x)
            out[x] = depth == 16 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : (uint16_t)cur[x];
        prev.swap(cur);
    }
    return img;
}

DepthImage readDepthPng(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) bad("cannot open " + path);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return decodeDepthPng(bytes.data(), bytes.size());
}

std::vector<uint8_t> encodeDepthPng(const uint16_t* px, int cols, int rows) {
    if (cols <= 0 || rows <= 0) bad("empty image");
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)cols * 2 + 1) * rows);
    for (int y = 0; y < rows; ++y) {
        raw.push_back(0);  // filter type None
        for (int x = 0; x < cols; ++x) raw.push_back((uint8_t)(px[(size_t)y * cols + x] >> 8)), raw.push_back((uint8_t)px[(size_t)y * cols + x]);
    }
    std::vector<uint8_t> z(compressBound((uLong)raw.size()));
    uLongf zn = (uLongf)z.size();
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) bad("deflate failed");
    z.resize(zn);
    std::vector<uint8_t> out(PNG_SIG, PNG_SIG + 8), ihdr;
    put32(ihdr, (uint32_t)cols), put32(ihdr, (uint32_t)rows);
    const uint8_t tail[5] = {16, 0, 0, 0, 0};  // 16 bits, greyscale, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr), chunk(out, "IDAT", z), chunk(out, "IEND", {});
    return out;
}

void writeDepthPng(const std::string& path, const uint16_t* px, int cols, int rows) {
    const std::vector<uint8_t> bytes = encodeDepthPng(px, cols, rows);
    std::ofstream f(path, std::ios::binary);
    if (!f.write((const char*)bytes.data(), (std::streamsize)bytes.size())) bad("cannot write " + path);
}

SequenceFiles listSequence(const std::string& dir) {
    namespace fs = std::filesystem;
    if (!fs::exists(dir)) throw Error(1, "Directory '" + dir + "' does not exist");  // demo.cpp:40-43
    if (!fs::exists(dir + "/depth") || !fs::exists(dir + "/color"))
        throw Error(1, "Directory should contain 'color' and 'depth' directories");   // demo.cpp:45-48
    SequenceFiles s;
    for (const auto& e : fs::directory_iterator(dir + "/depth"))
        if (e.is_regular_file()) s.depths.push_back(e.path().string());
    for (const auto& e : fs::directory_iterator(dir + "/color"))
        if (e.is_regular_file()) s.images.push_back(e.path().string());
    std::sort(s.depths.begin(), s.depths.end());  // demo.cpp:53-54
    std::sort(s.images.begin(), s.images.end());
    return s;
}

std::string pcdAsciiString(const PointCloud<PointXYZ>& cloud) {
    // pcl::PCDWriter::generateHeader + writeASCII for PointXYZ: the padding field is not a named field; an unorganised
    // cloud has width = points, height = 1; default viewpoint
    const size_t n = cloud.size();
    std::string s = "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n";
    s += "WIDTH " + std::to_string(n) + "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " + std::to_string(n) + "\nDATA ascii\n";
    for (const PointXYZ& p : cloud.points) s += fmt_g(p.x, 8) + " " + fmt_g(p.y, 8) + " " + fmt_g(p.z, 8) + "\n";
    return s;
}

std::string vtkMeshString(const PolygonMesh& mesh) {
    // pcl::io::saveVTKFile(file, PolygonMesh, precision = 5)
    const size_t n = mesh.cloud.size();
    std::string s = "# vtk DataFile Version 3.0\nvtk output\nASCII\nDATASET POLYDATA\nPOINTS " + std::to_string(n) + " float\n";
    for (const PointXYZ& p : mesh.cloud.points) s += fmt_g(p.x, 5) + " " + fmt_g(p.y, 5) + " " + fmt_g(p.z, 5) + "\n";
    s += "\nVERTICES " + std::to_string(n) + " " + std::to_string(2 * n) + "\n";
    for (size_t i = 0; i < n; ++i) s += "1 " + std::to_string(i) + "\n";
    size_t values = mesh.polygons.size();
    for (const auto& poly : mesh.polygons) values += poly.size();
    s += "\nPOLYGONS " + std::to_string(mesh.polygons.size()) + " " + std::to_string(values) + "\n";
    for (const auto& poly : mesh.polygons) {
        s += std::to_string(poly.size());
        for (uint32_t v : poly) s += " " + std::to_string(v);
        s += "\n";
    }
    return s;
}

static void write_text(const std::string& path, const std::string& text) {
    std::ofstream f(path, std::ios::binary);
    if (!f.write(text.data(), (std::streamsize)text.size())) throw Error(1, "could not save to " + path);
}
void savePCDFileASCII(const std::string& path, const PointCloud<PointXYZ>& cloud) { write_text(path, pcdAsciiString(cloud)); }
void saveVTKFile(const std::string& path, const PolygonMesh& mesh) { write_text(path, vtkMeshString(mesh)); }

}  // namespace io
}  // namespace dfa
