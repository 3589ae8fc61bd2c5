// test_host_io.cpp — the data formats either side of the path (dfa_host/io.hpp; reference: src/apps/demo.cpp).
//   test_host_io                      CPU tests (codec round trips, malformed files, writer texts, listing order)
//   test_host_io decode IN.png OUT    decode a PNG written by someone else (the pytest wrapper uses PIL) to raw u16
//   test_host_io encode W H OUT.png   write the ramp image the wrapper then reads back with PIL
//   test_host_io sequence DIR         GPU: run DynFusion over the PNG sequence in DIR (runSequence), print the report
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>

#include <dfa_host/io.hpp>

#include "minitest.hpp"

using namespace dfa;

static std::vector<uint16_t> ramp(int cols, int rows) {
    std::vector<uint16_t> px((size_t)cols * rows);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) px[(size_t)y * cols + x] = (uint16_t)(y * 257 + x * 3 + ((x * y) % 7 == 0 ? 0 : 500));
    return px;
}

TEST(DepthPng, RoundTripAllSizes) {
    std::mt19937 rng(1);
    for (auto wh : {std::pair<int, int>{1, 1}, {7, 3}, {640, 480}, {33, 200}}) {
        std::vector<uint16_t> px((size_t)wh.first * wh.second);
        for (auto& v : px) v = (uint16_t)rng();
        px[0] = 0, px.back() = 65535;
        const auto bytes = io::encodeDepthPng(px.data(), wh.first, wh.second);
        const auto img   = io::decodeDepthPng(bytes.data(), bytes.size());
        ASSERT_EQ(img.cols, wh.first);
        ASSERT_EQ(img.rows, wh.second);
        ASSERT_TRUE(img.data == px);
    }
}

TEST(DepthPng, MalformedFilesAreRejectedWithAReason) {
    const auto px = ramp(16, 8);
    auto good     = io::encodeDepthPng(px.data(), 16, 8);
    auto expect_throw = [](std::vector<uint8_t> b, const char* needle) {
        try {
            io::decodeDepthPng(b.data(), b.size());
        } catch (const Error& e) {
            ASSERT_TRUE(std::string(e.what()).find(needle) != std::string::npos);
            return;
        }
        ASSERT_TRUE(false && "no exception");
    };
    expect_throw({1, 2, 3}, "not a PNG");
    auto b = good;
    b[40] ^= 0x10;  // inside IDAT: the chunk CRC no longer matches
    expect_throw(b, "CRC");
    b = good;
    b.resize(b.size() - 20);
    expect_throw(b, "past the end");
    b = good;
    b[8 + 8 + 9] = 2;  // colour type RGB (CRC of IHDR then fails first: either reason is a rejection)
    try {
        io::decodeDepthPng(b.data(), b.size());
        ASSERT_TRUE(false);
    } catch (const Error&) {
    }
    try {
        io::readDepthPng("/nonexistent/depth.png");
        ASSERT_TRUE(false);
    } catch (const Error& e) {
        ASSERT_TRUE(std::string(e.what()).find("cannot open") != std::string::npos);
    }
}

TEST(Writers, PcdAsciiText) {
    PointCloud<PointXYZ> c;
    c.push_back(PointXYZ(0.5f, -1.25f, 3.f));
    c.push_back(PointXYZ(0.1f, 1e-5f, 123456.789f));
    c.push_back(PointXYZ(std::nanf(""), 0.f, -0.f));
    const std::string want =
        "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
        "WIDTH 3\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA ascii\n"
        "0.5 -1.25 3\n0.1 9.9999997e-06 123456.79\nnan 0 -0\n";
    ASSERT_TRUE(io::pcdAsciiString(c) == want);
    ASSERT_TRUE(io::pcdAsciiString(PointCloud<PointXYZ>()).find("POINTS 0\nDATA ascii\n") != std::string::npos);
}

TEST(Writers, VtkMeshTextAndTriangleOrder) {
    std::vector<PointXYZ> tri = {PointXYZ(0, 0, 0), PointXYZ(1, 0, 0), PointXYZ(0, 1, 0),
                                 PointXYZ(0, 0, 1), PointXYZ(1.23456789f, 0, 1), PointXYZ(0, 1, 1)};
    const PolygonMesh m = convertToMesh(tri);
    ASSERT_EQ(m.polygons.size(), (size_t)2);
    ASSERT_TRUE((m.polygons[1] == std::vector<uint32_t>{3, 5, 4}));  // kinfu.cpp:252-256: (3i, 3i+2, 3i+1)
    const std::string want =
        "# vtk DataFile Version 3.0\nvtk output\nASCII\nDATASET POLYDATA\nPOINTS 6 float\n"
        "0 0 0\n1 0 0\n0 1 0\n0 0 1\n1.2346 0 1\n0 1 1\n"
        "\nVERTICES 6 12\n1 0\n1 1\n1 2\n1 3\n1 4\n1 5\n"
        "\nPOLYGONS 2 8\n3 0 2 1\n3 3 5 4\n";
    ASSERT_TRUE(io::vtkMeshString(m) == want);
    ASSERT_TRUE(convertToMesh({}).polygons.empty());
}

TEST(Sequence, ListingIsSortedAndChecksTheLayout) {
    namespace fs = std::filesystem;
    const fs::path dir = fs::temp_directory_path() / "dfa_io_listing";
    fs::remove_all(dir);
    try {
        io::listSequence(dir.string());
        ASSERT_TRUE(false);
    } catch (const Error& e) {
        ASSERT_TRUE(std::string(e.what()).find("does not exist") != std::string::npos);
    }
    fs::create_directories(dir / "depth");
    try {
        io::listSequence(dir.string());
        ASSERT_TRUE(false);
    } catch (const Error& e) {
        ASSERT_TRUE(std::string(e.what()).find("'color' and 'depth'") != std::string::npos);
    }
    fs::create_directories(dir / "color");
    for (const char* n : {"frame-000010.depth.png", "frame-000002.depth.png", "frame-000001.depth.png"}) std::ofstream(dir / "depth" / n) << "x";
    for (const char* n : {"b.png", "a.png", "c.png"}) std::ofstream(dir / "color" / n) << "x";
    const auto s = io::listSequence(dir.string());
    ASSERT_EQ(s.depths.size(), (size_t)3);
    ASSERT_TRUE(s.depths[0].find("000001") != std::string::npos && s.depths[2].find("000010") != std::string::npos);
    ASSERT_TRUE(s.images[0].find("a.png") != std::string::npos);
    fs::remove_all(dir);
}

#ifdef DFA_WITH_DYNFUSION
#include <dynfu/dyn_fusion.hpp>
static int run_sequence(const char* dir) {
    DynFuParams p   = DynFuParams::defaultParams();
    const auto first = io::readDepthPng(io::listSequence(dir).depths.at(0));
    auto& k          = p.kinfuParams;
    k.cols = first.cols, k.rows = first.rows;
    k.intr = kfusion::Intr(k.intr.fx * first.cols / 640.f, k.intr.fy * first.rows / 480.f, first.cols / 2 - 0.5f, first.rows / 2 - 0.5f);
    p.intr = k.intr;
    const int dim = std::getenv("DFA_SEQ_DIM") ? std::atoi(std::getenv("DFA_SEQ_DIM")) : 64;
    k.volume_dims = dfa::Vec3i(dim, dim, dim);
    if (std::getenv("DFA_SEQ_NORTHSTAR")) p.north_star = true, p.epsilon = 0.05f;  // dev: the 6-DoF mode
    DynFusion dynfu(p);
    dynfu.nodeStep = dim >= 256 ? 128 : 64;
    const SequenceReport r = runSequence(dynfu, dir);
    std::printf("frames %d saved %d dynfu_ms %.2f nodes %zu canonical_vertices %zu mesh_polygons %zu\n", r.frames, r.saved, r.dynfu_ms,
                dynfu.getWarpfield()->getNodes().size(), dynfu.getCanonicalWarpedToLive()->size(),
                dynfu.getMesh()->polygons.size());
    if (std::getenv("DFA_SEQ_FRAME_MS")) {  // dev: per-frame time inside operator() (tools/host_sequence_timing.py)
        std::printf("frame_ms");
        for (double ms : r.frame_ms) std::printf(" %.3f", ms);
        std::printf("\n");
    }
    return 0;
}
#endif

int main(int argc, char** argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "decode")) {
        const auto img = io::readDepthPng(argv[2]);
        std::ofstream f(argv[3], std::ios::binary);
        const int32_t hdr[2] = {img.cols, img.rows};
        f.write((const char*)hdr, sizeof hdr);
        f.write((const char*)img.data.data(), (std::streamsize)(img.data.size() * 2));
        return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "encode")) {
        const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
        const auto px = ramp(w, h);
        io::writeDepthPng(argv[4], px.data(), w, h);
        return 0;
    }
    if (argc >= 2 && !std::strcmp(argv[1], "bench")) {  // host-side cost of the formats at VGA
        const auto px = ramp(640, 480);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> bytes;
        for (int i = 0; i < 50; ++i) bytes = io::encodeDepthPng(px.data(), 640, 480);
        const auto t1 = std::chrono::steady_clock::now();
        size_t sum = 0;
        for (int i = 0; i < 50; ++i) sum += io::decodeDepthPng(bytes.data(), bytes.size()).data[i];
        const auto t2 = std::chrono::steady_clock::now();
        PointCloud<PointXYZ> c;
        for (int i = 0; i < 262144; ++i) c.push_back(PointXYZ(i * 1e-3f, 0.5f, -i * 3e-4f));
        const std::string pcd = io::pcdAsciiString(c);
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::printf("VGA depth png: encode %.2f ms, decode %.2f ms (%zu bytes); PCD of 262144 points %.1f ms (%zu bytes) [%zu]\n",
                    ms(t0, t1) / 50, ms(t1, t2) / 50, bytes.size(), ms(t2, t3), pcd.size(), sum);
        return 0;
    }
#ifdef DFA_WITH_DYNFUSION
    if (argc >= 3 && !std::strcmp(argv[1], "sequence")) return run_sequence(argv[2]);
#endif
    return mt::run_all(argc, argv);
}
