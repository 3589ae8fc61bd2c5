// test_tsdf_classify.cpp — CPU model of the run-classified TSDF sweep (dynfu_amd/csrc/tsdf_classify.hpp).
//
// The integrate kernels of tsdf.hip decide per run of U voxels of a column whether the whole run is skipped, is
// updated with tsdf == 1, or takes the reference's per-voxel path.  This program walks volumes exactly as those
// kernels do — same chunking, same running `vc += zstep`, same run ends, same header — with the per-voxel path of
// the FULL runs restated in IEEE arithmetic, and compares the result with oracle/tsdf_oracle.c bit for bit.  A run
// that is wrongly called SKIP or FRONT shows up as a voxel that differs.  No GPU involved.
//
//   test_tsdf_classify              run the test cases
//   test_tsdf_classify stats DIM    class fractions of the bench scene at DIM^3 (dev aid)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../dynfu_amd/csrc/tsdf_classify.hpp"
#include "minitest.hpp"

extern "C" {
#include "../../oracle/oracle.h"
}

namespace {

struct Scene {
    int cols, rows;
    float fx, fy, cx, cy;
    std::vector<uint16_t> dists;  // fp16 bits, dense rows
};

struct Volume {
    int X, Y, Z;
    float voxel[3];
    float trunc;
    int max_weight;
    float vol2cam[12];  // 9 rotation (row-major) + 3 translation
};

struct Stats {
    long runs[3] = {0, 0, 0};
    long wave_full = 0, waves = 0;  // 64-lane x segments with at least one FULL run
    long chunks_skipped = 0, columns_chunks = 0;  // (column, z chunk) pairs the chunk-level rule skipped / all of them
    std::vector<uint8_t>* classes = nullptr;  // [run][y][x] when set (stats mode)
};

float h2f(uint32_t bits) { return orc_half_to_float((uint16_t)bits); }
float rcp_host(float x) { return 1.0f / x; }

std::vector<uint32_t> build_tiles(const Scene& s) {
    const int T = 1 << dfa::TSDF_TILE_SHIFT;
    const int tc = (s.cols + T - 1) / T, tr = (s.rows + T - 1) / T;
    std::vector<uint32_t> tiles((size_t)tc * tr);
    for (int ty = 0; ty < tr; ++ty)
        for (int tx = 0; tx < tc; ++tx) {
            uint32_t lo = 0xffffu, hi = 0u;
            for (int y = ty * T; y < std::min(s.rows, ty * T + T); ++y)
                for (int x = tx * T; x < std::min(s.cols, tx * T + T); ++x) {
                    const uint32_t v = dfa::tile_bounds_of_pixel(s.dists[(size_t)y * s.cols + x]);
                    lo = std::min(lo, v & 0xffffu), hi = std::max(hi, v >> 16);
                }
            tiles[(size_t)ty * tc + tx] = lo | (hi << 16);
        }
    return tiles;
}

float dot3(float ax, float ay, float az, float bx, float by, float bz) { return fmaf(az, bz, fmaf(ay, by, ax * bx)); }

// integrate_voxel of tsdf.hip in IEEE arithmetic (== the oracle's per-voxel body)
uint32_t exact_voxel(const Scene& s, const Volume& v, float x, float y, float z, uint32_t old) {
    if (!(z > 0.f)) return old;
    const float coox = fmaf(s.fx, x / z, s.cx), cooy = fmaf(s.fy, y / z, s.cy);
    if (!(coox >= 0.f && cooy >= 0.f && coox < (float)s.cols && cooy < (float)s.rows)) return old;
    const float Dp = h2f(s.dists[(size_t)(int)cooy * s.cols + (int)coox]);
    if (Dp == 0.f) return old;
    const float sdf = Dp - sqrtf(dot3(x, y, z, x, y, z));
    if (!(sdf >= -v.trunc)) return old;
    const float tsdf      = fminf(1.f, sdf * (1.f / v.trunc));
    const int wp          = (int)(old >> 16);
    const float tp        = h2f(old & 0xffffu);
    const float tn        = fmaf(tp, (float)wp, tsdf) / (float)(wp + 1);
    const int wn          = wp + 1 < v.max_weight ? wp + 1 : v.max_weight;
    return (uint32_t)orc_float_to_half(tn) | ((uint32_t)wn << 16);
}

float extent_of(const Volume& v) {
    float m = 0.f;
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {(c & 1) ? v.voxel[0] * v.X : 0.f, (c & 2) ? v.voxel[1] * v.Y : 0.f, (c & 4) ? v.voxel[2] * v.Z : 0.f};
        for (int r = 0; r < 3; ++r)
            m = std::max(m, std::fabs(v.vol2cam[3 * r] * p[0] + v.vol2cam[3 * r + 1] * p[1] + v.vol2cam[3 * r + 2] * p[2] + v.vol2cam[9 + r]));
    }
    return m;
}

// the kernel's walk: columns, z chunks with replayed additions, runs of U slices, per-voxel tail
void model_integrate(bool fused, const Scene& s, const Volume& v, std::vector<uint32_t>& vol, int zchunk, int U, Stats* st) {
    const std::vector<uint32_t> tiles = build_tiles(s);
    const float zs[3] = {v.vol2cam[2] * v.voxel[2], v.vol2cam[5] * v.voxel[2], v.vol2cam[8] * v.voxel[2]};
    const dfa::RunConsts c =
        dfa::make_run_consts(tiles.data(), s.cols, s.rows, s.fx, s.fy, s.cx, s.cy, v.trunc, zs, U, extent_of(v));
    // the chunk-level rule (a column's whole z chunk skipped at once): margins of Z running additions
    const dfa::RunConsts cc =
        dfa::make_run_consts(tiles.data(), s.cols, s.rows, s.fx, s.fy, s.cx, s.cy, v.trunc, zs, v.Z, extent_of(v));
    const size_t slice = (size_t)v.X * v.Y;
    // FRONT on a cleared voxel: tsdf 1, weight min(1, max_weight)
    const uint32_t front_const = (uint32_t)orc_float_to_half(1.0f) | ((uint32_t)(1 < v.max_weight ? 1 : v.max_weight) << 16);
    std::vector<uint8_t> seg_full;
    for (int z0 = 0; z0 < v.Z; z0 += zchunk) {
        const int z1 = std::min(z0 + zchunk, v.Z);
        for (int y = 0; y < v.Y; ++y) {
            seg_full.assign((size_t)((v.X + 63) / 64) * ((z1 - z0 + U - 1) / U), 0);
            for (int x = 0; x < v.X; ++x) {
                const float vx = (float)x * v.voxel[0], vy = (float)y * v.voxel[1];
                float px = dot3(v.vol2cam[0], v.vol2cam[1], v.vol2cam[2], vx, vy, 0.f) + v.vol2cam[9];
                float py = dot3(v.vol2cam[3], v.vol2cam[4], v.vol2cam[5], vx, vy, 0.f) + v.vol2cam[10];
                float pz = dot3(v.vol2cam[6], v.vol2cam[7], v.vol2cam[8], vx, vy, 0.f) + v.vol2cam[11];
                uint32_t* p = vol.data() + (size_t)x + (size_t)v.X * y + slice * z0;
                if (dfa::chunk_skipped(px, py, pz, zs, z0, z1, cc, rcp_host, h2f)) {  // every voxel of the chunk is left alone
                    if (st) st->chunks_skipped++, st->columns_chunks++, st->runs[dfa::RUN_SKIP] += (z1 - z0) / U;
                    if (fused)
                        for (int z = z0; z < z1; ++z, p += slice) *p = 0u;
                    continue;
                }
                if (st) st->columns_chunks++;
                for (int i = 0; i < z0; ++i) px += zs[0], py += zs[1], pz += zs[2];
                int z = z0;
                dfa::RunEnd a = dfa::run_end(px, py, pz, c, rcp_host);
                for (; z + U <= z1; z += U) {
                    const dfa::RunEnd b = dfa::run_end(px + c.stepU[0], py + c.stepU[1], pz + c.stepU[2], c, rcp_host);
                    const int cls       = dfa::classify_run(a, b, c, h2f);
                    a                   = b;
                    if (st) {
                        st->runs[cls]++;
                        if (st->classes) (*st->classes)[((size_t)(z / U) * v.Y + y) * v.X + x] = (uint8_t)cls;
                        if (cls == dfa::RUN_FULL) seg_full[(size_t)(x / 64) * ((z1 - z0 + U - 1) / U) + (z - z0) / U] = 1;
                    }
                    for (int u = 0; u < U; ++u, p += slice) {
                        const uint32_t old = fused ? 0u : *p;
                        uint32_t nv;
                        if (cls == dfa::RUN_FULL) nv = exact_voxel(s, v, px, py, pz, old);
                        else if (cls == dfa::RUN_SKIP) nv = old;
                        else if (fused) nv = front_const;
                        else {  // FRONT on a live voxel: the update of :86-87 with tsdf = 1
                            const int wp   = (int)(old >> 16);
                            const float tn = fmaf(h2f(old & 0xffffu), (float)wp, 1.0f) / (float)(wp + 1);
                            nv             = (uint32_t)orc_float_to_half(tn) | ((uint32_t)(wp + 1 < v.max_weight ? wp + 1 : v.max_weight) << 16);
                        }
                        *p = nv;
                        px += zs[0], py += zs[1], pz += zs[2];
                    }
                }
                for (; z < z1; ++z, p += slice) {
                    *p = exact_voxel(s, v, px, py, pz, fused ? 0u : *p);
                    px += zs[0], py += zs[1], pz += zs[2];
                }
            }
            if (st)
                for (uint8_t f : seg_full) st->waves++, st->wave_full += f;
        }
    }
}

// sphere (radius 0.5 at (0,0,1.5)) in front of the plane z = 2.5, depth in mm -> dists (the bench scene without bulge)
Scene make_scene(int cols, int rows, float focal, int border) {
    Scene s;
    s.cols = cols, s.rows = rows, s.fx = s.fy = focal, s.cx = cols / 2 - 0.5f, s.cy = rows / 2 - 0.5f;
    std::vector<uint16_t> depth((size_t)cols * rows, 0);
    for (int y = border; y < rows - border; ++y)
        for (int x = border; x < cols - border; ++x) {
            double d[3] = {(x - s.cx) / s.fx, (y - s.cy) / s.fy, 1.0};
            const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + 1.0);
            d[0] /= n, d[1] /= n, d[2] /= n;
            const double b = d[2] * 1.5, disc = b * b - (1.5 * 1.5 - 0.25);
            double zhit = 2.5;
            if (disc > 0) zhit = (b - std::sqrt(disc)) * d[2];
            depth[(size_t)y * cols + x] = (uint16_t)std::lround(zhit * 1000.0);
        }
    s.dists.resize((size_t)cols * rows);
    orc_compute_dists(depth.data(), cols * 2, s.dists.data(), cols * 2, cols, rows, s.fx, s.fy, s.cx, s.cy);
    return s;
}

Volume make_volume(int dim, float size, float tx, float ty, float tz) {
    Volume v;
    v.X = v.Y = v.Z = dim;
    v.voxel[0] = v.voxel[1] = v.voxel[2] = size / (float)dim;
    v.trunc      = std::max(0.04f, 2.1f * v.voxel[0]);
    v.max_weight = 64;
    const float id[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, tx, ty, tz};
    std::memcpy(v.vol2cam, id, sizeof id);
    return v;
}

void rotate(Volume& v, float ax, float ay, float az) {  // vol2cam <- Rz Ry Rx about the volume centre, keeps the centre's image
    const float cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    const float R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, sz * cy, sz * sy * sx + cz * cx,
                        sz * sy * cx - cz * sx, -sy, cy * sx, cy * cx};
    const float c[3] = {v.voxel[0] * v.X / 2, v.voxel[1] * v.Y / 2, v.voxel[2] * v.Z / 2};
    float centre[3];
    for (int r = 0; r < 3; ++r) centre[r] = c[r] + v.vol2cam[9 + r];
    for (int i = 0; i < 9; ++i) v.vol2cam[i] = R[i];
    for (int r = 0; r < 3; ++r) v.vol2cam[9 + r] = centre[r] - (R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2]);
}

long compare(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    long bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    return bad;
}

// fused sweep and two accumulating sweeps, model vs oracle
void check(const Scene& s, const Volume& v, int zchunk, int U, Stats* st = nullptr) {
    const size_t n = (size_t)v.X * v.Y * v.Z;
    std::vector<uint32_t> ref(n, 0u), mod(n, 0xdeadbeefu);
    orc_tsdf_integrate(s.dists.data(), s.cols * 2, s.cols, s.rows, ref.data(), v.X, v.Y, v.Z, v.voxel, v.trunc, v.max_weight,
                       v.vol2cam, s.fx, s.fy, s.cx, s.cy, 8);
    model_integrate(true, s, v, mod, zchunk, U, st);
    ASSERT_EQ(compare(ref, mod), 0L);
    // second and third sweep of the same frame accumulate (weights 2, 3; the max_weight clamp when it is small)
    for (int rep = 0; rep < 2; ++rep) {
        orc_tsdf_integrate(s.dists.data(), s.cols * 2, s.cols, s.rows, ref.data(), v.X, v.Y, v.Z, v.voxel, v.trunc,
                           v.max_weight, v.vol2cam, s.fx, s.fy, s.cx, s.cy, 8);
        model_integrate(false, s, v, mod, zchunk, U, nullptr);
        ASSERT_EQ(compare(ref, mod), 0L);
    }
}

}  // namespace

TEST(TsdfClassify, BenchSceneIdentityPose) {
    const Scene s = make_scene(320, 240, 262.5f, 5);
    Volume v      = make_volume(128, 3.f, -1.5f, -1.5f, 0.5f);
    Stats st;
    check(s, v, 128, 8, &st);
    Stats quarter;  // (chunks of a quarter column, as the kernel cuts a 512^3 volume)
    check(s, v, 32, 8, &quarter);
    check(s, v, 128, 4);
    const double tot = st.runs[0] + st.runs[1] + st.runs[2];
    std::printf("    runs: skip %.1f %%  front %.1f %%  full %.1f %%; wave segments with a full run %.1f %%; (column, chunk) pairs "
                "skipped whole %.1f %%\n", 100 * st.runs[0] / tot, 100 * st.runs[1] / tot, 100 * st.runs[2] / tot,
                100.0 * st.wave_full / st.waves, 100.0 * quarter.chunks_skipped / quarter.columns_chunks);
    ASSERT_TRUE(quarter.chunks_skipped > 0.3 * quarter.columns_chunks);  // (half of the volume is outside the frustum)
    ASSERT_TRUE(st.runs[dfa::RUN_FULL] < 0.25 * tot);  // the point of the exercise
}

TEST(TsdfClassify, RotatedCameras) {
    const Scene s = make_scene(320, 240, 262.5f, 5);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> ang(-0.6f, 0.6f);
    for (int i = 0; i < 6; ++i) {
        Volume v = make_volume(96, 3.f, -1.5f, -1.5f, 0.5f);
        rotate(v, ang(rng), ang(rng), ang(rng));
        check(s, v, 32 + 32 * (i & 1), i < 4 ? 8 : 4);
    }
}

TEST(TsdfClassify, CameraInsideTheVolume) {
    // the camera plane z = 0 cuts the volume: runs with z <= 0, z ~ 0 and huge projections
    const Scene s = make_scene(320, 240, 262.5f, 0);
    Volume v      = make_volume(96, 3.f, -1.5f, -1.5f, -1.0f);
    check(s, v, 96, 8);
    rotate(v, 0.3f, -0.5f, 0.2f);
    check(s, v, 32, 8);
    Volume w = make_volume(64, 0.5f, -0.25f, -0.25f, 0.0f);  // close-up: several pixels per voxel, runs span many tiles
    check(s, w, 64, 8);
    check(s, w, 64, 4);
}

TEST(TsdfClassify, HolesAndSpecialHalves) {
    Scene s = make_scene(320, 240, 262.5f, 5);
    std::mt19937 rng(11);
    std::uniform_int_distribution<int> px(0, s.cols - 1), py(0, s.rows - 1), kind(0, 8);
    for (int i = 0; i < 400; ++i) {  // patches of invalid / odd values: 0, -0, -2, NaN, +inf, subnormal, -subnormal, -0.01, -inf
        const int x0 = px(rng), y0 = py(rng), w = 1 + px(rng) % 12, h = 1 + py(rng) % 12;
        static const uint16_t vals[9] = {0x0000, 0x8000, 0xc000, 0x7e00, 0x7c00, 0x0001, 0x8001, 0xa11f, 0xfc00};
        const uint16_t val            = vals[kind(rng)];
        for (int y = y0; y < std::min(s.rows, y0 + h); ++y)
            for (int x = x0; x < std::min(s.cols, x0 + w); ++x) s.dists[(size_t)y * s.cols + x] = val;
    }
    Volume v = make_volume(128, 3.f, -1.5f, -1.5f, 0.5f);
    check(s, v, 64, 8);
    rotate(v, -0.2f, 0.4f, 0.1f);
    check(s, v, 128, 8);
    check(s, v, 128, 4);
    // a volume that starts AT the camera with a thick band: voxels within trunc - |Dp| of the camera see the small
    // negative pixels (the reference only skips Dp == 0, tsdf_volume.cu:74)
    Volume n = make_volume(64, 0.6f, -0.3f, -0.3f, 0.01f);
    n.trunc  = 0.1f;
    check(s, n, 64, 8);
    check(s, n, 64, 4);
}

TEST(TsdfClassify, RaggedDimsOddImageSmallWeights) {
    Scene s = make_scene(173, 131, 140.f, 3);  // partial tiles on the right and bottom edges
    Volume v = make_volume(64, 3.f, -1.5f, -1.5f, 0.5f);
    v.X = 70, v.Y = 45, v.Z = 51;  // tail runs, partial wave segments
    v.max_weight = 2;
    check(s, v, 20, 8);
    v.max_weight = 0;
    check(s, v, 51, 8);
    v.max_weight = 1;
    v.trunc      = 0.2f;  // thick band
    check(s, v, 16, 8);
}

TEST(TsdfClassify, FarAndLargeScenes) {
    // volumes far from the origin: the margins scale with the extent
    const Scene s = make_scene(320, 240, 262.5f, 5);
    Volume v      = make_volume(64, 30.f, -15.f, -15.f, 0.5f);  // 47 cm voxels: trunc is clamped up to 2.1 voxels
    check(s, v, 64, 8);
    Volume w = make_volume(96, 3.f, 40.f, -1.5f, 60.f);  // entirely out of the frustum, 70 m away
    check(s, w, 96, 8);
}

int main(int argc, char** argv) {
    if (argc > 2 && !std::strcmp(argv[1], "stats")) {
        const int dim = std::atoi(argv[2]);
        const Scene s = dim > 512 ? make_scene(1280, 720, 1050.f, 5) : make_scene(640, 480, 525.f, 5);
        Volume v      = make_volume(dim, 3.f, -1.5f, -1.5f, 0.5f);
        for (int U : {4, 8}) {
            Stats st;
            std::vector<uint8_t> classes((size_t)(dim / U) * dim * dim);
            st.classes = &classes;
            std::vector<uint32_t> mod((size_t)dim * dim * dim);
            model_integrate(true, s, v, mod, 128, U, &st);
            for (int wx : {64, 32, 16, 8}) {  // wave footprint wx x (64 / wx) columns
                const int wy = 64 / wx;
                long full = 0, n = 0;
                for (int r = 0; r < dim / U; ++r)
                    for (int y0 = 0; y0 < dim; y0 += wy)
                        for (int x0 = 0; x0 < dim; x0 += wx) {
                            bool f = false;
                            for (int y = y0; y < y0 + wy; ++y)
                                for (int x = x0; x < x0 + wx; ++x) f |= classes[((size_t)r * dim + y) * dim + x] == dfa::RUN_FULL;
                            full += f, ++n;
                        }
                std::printf("   wave %2d x %d: %.2f %% of the wave runs hold a full lane\n", wx, wy, 100.0 * full / n);
            }
            const double tot = st.runs[0] + st.runs[1] + st.runs[2];
            std::printf("dim %d U %d: skip %.2f %% front %.2f %% full %.2f %%; wave segments with a full run %.2f %%\n", dim, U,
                        100 * st.runs[0] / tot, 100 * st.runs[1] / tot, 100 * st.runs[2] / tot, 100.0 * st.wave_full / st.waves);
        }
        return 0;
    }
    return mt::run_all(argc, argv);
}
