// sequence_bench — the reference's own timed region on the adaptor classes: time inside DynFusion::operator()
// (src/apps/demo.cpp:90-95: upload outside, operator() inside), frame by frame, over a sequence of raw depth frames.
//
//   sequence_bench FRAMES.u16 W H N DIM ref|northstar [epsilon] [node_step]
//
// FRAMES.u16: N frames of W x H little-endian uint16 millimetres, back to back (bench.py writes its synthetic
// sequence there).  Prints one JSON object: per-frame milliseconds (the device is synchronised inside the timed region,
// so nothing of a frame hides behind the next one), node / vertex counts, the north-star energies.  No files are
// written: the demo's PCD / VTK output sits outside its timed call (demo.cpp:115-118).
#include <dynfu/dyn_fusion.hpp>

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s FRAMES.u16 W H N DIM ref|northstar [epsilon] [node_step]\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]), dim = std::atoi(argv[5]);
    const bool ns = !std::strcmp(argv[6], "northstar");
    std::vector<uint16_t> all((size_t)W * H * N);
    std::ifstream f(argv[1], std::ios::binary);
    if (!f.read((char*)all.data(), (std::streamsize)(all.size() * 2))) {
        std::fprintf(stderr, "cannot read %zu bytes from %s\n", all.size() * 2, argv[1]);
        return 2;
    }
    try {
        DynFuParams p = DynFuParams::defaultParams();  // dyn_fusion.cpp:6-31
        auto& k       = p.kinfuParams;
        k.cols = W, k.rows = H;
        k.intr        = kfusion::Intr(k.intr.fx * W / 640.f, k.intr.fy * H / 480.f, W / 2 - 0.5f, H / 2 - 0.5f);
        p.intr        = k.intr;
        k.volume_dims = dfa::Vec3i(dim, dim, dim);
        p.north_star  = ns;
        if (argc > 7) p.epsilon = (float)std::atof(argv[7]);
        else if (ns) p.epsilon = 0.05f;
        DynFusion dynfu(p);
        dynfu.nodeStep = argc > 8 ? std::atoi(argv[8]) : 128;  // dyn_fusion.cpp:151
        kfusion::cuda::Depth depth;
        std::vector<double> ms;
        std::vector<size_t> nodes;
        for (int i = 0; i < N; ++i) {
            depth.upload(all.data() + (size_t)i * W * H, (size_t)W * sizeof(uint16_t), H, W);  // demo.cpp:90
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            dynfu(depth);  // demo.cpp:92-95
            (void)hipDeviceSynchronize();
            ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            nodes.push_back(dynfu.getWarpfield()->getNodes().size());
        }
        std::printf("{\"mode\": \"%s\", \"dim\": %d, \"width\": %d, \"height\": %d, \"frames\": %d, \"canonical_vertices\": %zu, "
                    "\"live_vertices_last\": %zu, \"nodes_first\": %zu, \"nodes_last\": %zu, \"epsilon\": %g, \"node_step\": %d",
                    ns ? "northstar" : "ref", dim, W, H, N, dynfu.getCanonicalWarpedToLive()->size(),
                    dynfu.getLiveFrame() ? dynfu.getLiveFrame()->size() : (size_t)0, nodes.front(), nodes.back(), p.epsilon,
                    dynfu.nodeStep);
        if (ns)
            std::printf(", \"cost_first\": %.6g, \"cost_last\": %.6g, \"valid_rows\": %lld", dynfu.northStarInitialCost(),
                        dynfu.northStarFinalCost(), dynfu.northStarValidRows());
        std::printf(", \"frame_ms\": [");
        for (size_t i = 0; i < ms.size(); ++i) std::printf("%s%.4f", i ? ", " : "", ms[i]);
        std::printf("]}\n");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "sequence_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
