// tsdf_classify.hpp — conservative classification of short z-runs of voxels for the TSDF sweeps (tsdf.hip).
//
// A lane of the integrate kernel marches one voxel column over z (tsdf_volume.cu:62-93).  Most voxels of a sweep
// never reach the arithmetic that matters: half of a 512^3 volume projects outside the image, a third lies far behind
// the surface, most of the rest far in front of it (tsdf == 1 exactly).  This header decides, for a run of U
// consecutive voxels of one column at once and WITHOUT the per-voxel divisions / gather / square root, whether
//   RUN_SKIP   every voxel of the run fails one of the reference's tests (:70 frustum, :74 Dp == 0 | z <= 0, :79 sdf),
//   RUN_FRONT  every voxel of the run is updated with tsdf == 1.0f exactly (:80 min(1, sdf / trunc)),
//   RUN_FULL   anything else: the run takes the reference's per-voxel path unchanged.
// The decision is conservative — a run is only called SKIP / FRONT when that outcome is certain under stated
// margins — so the volume stays bit-identical to the per-voxel evaluation (tests/cpp/test_tsdf_classify.cpp sweeps
// this on the CPU against oracle/tsdf_oracle.c; tests/test_gpu_tsdf.py on the GPU).
//
// Geometry of the argument.  The voxels of a run lie (to within `pos_eps`, the rounding of the running `vc += zstep`)
// on the segment [a, b] between the run's first voxel a and b = a + U*zstep.  With z >= zmin > 0 on both ends:
//   * the projection u(p) = fx*x/z + cx of a point moving along a segment is monotone, so every pixel coordinate of
//     the run lies between the projections of the ends (+- a margin for pos_eps and the approximate arithmetic here);
//   * |p| is convex along the segment: max at an end, min >= min(|a|, |b|) - |b - a| / 2.
// The depth ("dists") values a run can meet are bounded by a min / max table over 8x8-pixel tiles (built per call by
// dists_tiles_kernel): runs whose pixel box spans more than 2x2 tiles are RUN_FULL.
//
// Plain float arithmetic, host + device: the CPU model test compiles this header with g++.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DFA_HD __host__ __device__ __forceinline__
#else
#define DFA_HD static inline
#endif

namespace dfa {

constexpr int TSDF_TILE_SHIFT = 3;  // 8 x 8 pixel tiles
enum { RUN_SKIP = 0, RUN_FRONT = 1, RUN_FULL = 2 };

// One tile of the dists image: low 16 bits = a lower bound, high 16 bits = an upper bound (fp16 bit patterns; non-negative
// fp16 values order like their bit patterns) of what its pixels can contribute.  Per pixel (lo, hi):
//   valid (positive, finite or +inf)          (bits, bits)
//   the reference skips it whatever the voxel (0, 0)        +-0 (tsdf_volume.cu:74 `Dp == 0`), NaN and -inf (sdf is NaN / -inf:
//                                                           :79 fails)
//   negative and finite                       (0, +inf)     NOT skipped by :74 — a voxel nearer to the camera than
//                                                           trunc - |Dp| passes :79 — so a run that can meet one is never
//                                                           called SKIP (upper bound +inf) nor FRONT (lower bound 0):
//                                                           it takes the per-voxel path.  compute_dists never produces
//                                                           one; a caller's own dists image may.
DFA_HD uint32_t tile_bounds_of_pixel(uint32_t half_bits) {
    const uint32_t mag = half_bits & 0x7fffu;
    if ((half_bits & 0x8000u) == 0u) return mag != 0u && mag <= 0x7c00u ? (half_bits | (half_bits << 16)) : 0u;
    return mag != 0u && mag < 0x7c00u ? (0x7c00u << 16) : 0u;
}

struct RunConsts {
    const uint32_t* tiles;  // [trows][tcols]
    int tcols;
    float fx, fy, cx, cy;
    float colsf, rowsf, colsm1f, rowsm1f;
    float zmin;     // ends nearer to the camera plane than this are never classified
    float zneg;     // both ends below -zneg: every voxel has z < 0
    float pix_c0x, pix_c0y, pix_c1;  // pixel margin = c0 + c1 * (|u_a| + |u_b|)
    float skip_k;   // trunc + half run length + distance margin
    float front_k;  // trunc + distance margin
    float stepU[3];  // U * zstep
};

// projection and squared range of one end of a run
struct RunEnd {
    float u, v, d2, z;
};

template <class Rcp>
DFA_HD RunEnd run_end(float x, float y, float z, const RunConsts& c, Rcp rcp) {
    RunEnd e;
    const float r = rcp(fmaxf(z, c.zmin));  // 1-ulp reciprocal is enough; unused when z < zmin
    e.u  = fmaf(c.fx, x * r, c.cx);
    e.v  = fmaf(c.fy, y * r, c.cy);
    e.d2 = fmaf(z, z, fmaf(y, y, x * x));
    e.z  = z;
    return e;
}

template <class HalfToFloat>
DFA_HD int classify_run(const RunEnd& a, const RunEnd& b, const RunConsts& c, HalfToFloat h2f) {
    if (!(fminf(a.z, b.z) >= c.zmin)) return fmaxf(a.z, b.z) < -c.zneg ? RUN_SKIP : RUN_FULL;
    const float mu   = fmaf(c.pix_c1, fabsf(a.u) + fabsf(b.u), c.pix_c0x);
    const float mv   = fmaf(c.pix_c1, fabsf(a.v) + fabsf(b.v), c.pix_c0y);
    const float umin = fminf(a.u, b.u) - mu, umax = fmaxf(a.u, b.u) + mu;
    const float vmin = fminf(a.v, b.v) - mv, vmax = fmaxf(a.v, b.v) + mv;
    if (umax < 0.f || vmax < 0.f || umin >= c.colsf || vmin >= c.rowsf) return RUN_SKIP;  // :70 for every voxel
    // pixels the run can land on: [ix0, ix1] x [iy0, iy1], clamped to the image (truncation == floor: non-negative)
    const int tx0 = (int)fmaxf(umin, 0.f) >> TSDF_TILE_SHIFT, tx1 = (int)fminf(umax, c.colsm1f) >> TSDF_TILE_SHIFT;
    const int ty0 = (int)fmaxf(vmin, 0.f) >> TSDF_TILE_SHIFT, ty1 = (int)fminf(vmax, c.rowsm1f) >> TSDF_TILE_SHIFT;
    if (tx1 - tx0 > 1 || ty1 - ty0 > 1) return RUN_FULL;
    const uint32_t* r0 = c.tiles + (long)ty0 * c.tcols;
    const uint32_t* r1 = c.tiles + (long)ty1 * c.tcols;
    const uint32_t t00 = r0[tx0], t01 = r0[tx1], t10 = r1[tx0], t11 = r1[tx1];
    const uint32_t lo01 = (t00 & 0xffffu) < (t01 & 0xffffu) ? (t00 & 0xffffu) : (t01 & 0xffffu);
    const uint32_t lo23 = (t10 & 0xffffu) < (t11 & 0xffffu) ? (t10 & 0xffffu) : (t11 & 0xffffu);
    const uint32_t lo   = lo01 < lo23 ? lo01 : lo23;
    const uint32_t hi01 = t00 > t01 ? t00 : t01;  // the high halves decide the order of the words
    const uint32_t hi23 = t10 > t11 ? t10 : t11;
    const uint32_t hi   = (hi01 > hi23 ? hi01 : hi23) >> 16;
    if (hi == 0u) return RUN_SKIP;  // no valid pixel in reach: :74 for every voxel
    const float max_dp = h2f(hi), min_dp = h2f(lo);
    const float dmin2 = fminf(a.d2, b.d2), dmax2 = fmaxf(a.d2, b.d2);
    const float sk = max_dp + c.skip_k;
    if (dmin2 > (sk * sk) * 1.0001f) return RUN_SKIP;  // :79 sdf < -trunc for every voxel and pixel in reach
    const float fr = min_dp - c.front_k;
    if (umin >= 0.f && vmin >= 0.f && umax < c.colsf && vmax < c.rowsf && fr > 0.f && dmax2 * 1.0001f < fr * fr)
        return RUN_FRONT;  // every pixel valid, sdf > trunc: tsdf == 1
    return RUN_FULL;
}

// Host side: the margins for one sweep.  `extent` bounds |component| of every voxel position in camera space
// (the 8 volume corners through vol2cam); `zstep` is the per-slice step (tsdf_volume.cu:58).
inline RunConsts make_run_consts(const uint32_t* tiles, int cols, int rows, float fx, float fy, float cx, float cy,
                                 float trunc, const float zstep[3], int U, float extent) {
    RunConsts c;
    c.tiles = tiles;
    c.tcols = (cols + (1 << TSDF_TILE_SHIFT) - 1) >> TSDF_TILE_SHIFT;
    c.fx = fx, c.fy = fy, c.cx = cx, c.cy = cy;
    c.colsf = (float)cols, c.rowsf = (float)rows, c.colsm1f = (float)(cols - 1), c.rowsm1f = (float)(rows - 1);
    // a voxel position is the sum of up to Z roundings of magnitude <= ulp(extent)/2 relative to the straight line
    // through the run's first voxel only over the U additions inside the run: U half-ulps, doubled for the end b
    // computed here with one multiply-add, doubled again for slack
    const float pos_eps = extent * (float)(4 * (U + 2)) * 5.9604645e-08f;  // 2^-24
    c.zmin = fmaxf(0.05f, 1000.f * pos_eps);
    c.zneg = fmaxf(1e-4f, 16.f * pos_eps);
    const float kz = pos_eps / c.zmin;
    c.pix_c0x = kz * (fabsf(fx) + fabsf(cx)) + 1e-3f + 2e-6f * fabsf(cx);
    c.pix_c0y = kz * (fabsf(fy) + fabsf(cy)) + 1e-3f + 2e-6f * fabsf(cy);
    c.pix_c1  = kz + 2e-6f;
    const float len = sqrtf(zstep[0] * zstep[0] + zstep[1] * zstep[1] + zstep[2] * zstep[2]) * (float)U;
    const float dm  = 1e-3f + 1e-4f * trunc + 8.f * pos_eps;
    c.skip_k  = trunc + 0.5f * len * 1.001f + dm;
    c.front_k = trunc + dm;
    for (int i = 0; i < 3; ++i) c.stepU[i] = zstep[i] * (float)U;
    return c;
}

// A whole z chunk of a column at once: true only when EVERY voxel of the slices [z0, z1) of the column is skipped (the run rule
// above applied to the chunk's ends).  `p0` is the column's position at slice 0; the ends are taken from it with one
// multiply-add each instead of the z0 running additions the sweep replays, so `cc` must carry the margins of Z additions:
// make_run_consts(..., U = Z, ...) (its run length, the whole column, bounds the chunk's).  Half of a 512^3 volume lies outside
// the frustum: a wave whose 64 columns all answer true has nothing to classify, replay or — in a sweep that leaves skipped
// voxels alone — touch.
template <class Rcp, class HalfToFloat>
DFA_HD bool chunk_skipped(float p0x, float p0y, float p0z, const float zstep[3], int z0, int z1, const RunConsts& cc, Rcp rcp,
                          HalfToFloat h2f) {
    const float fa = (float)z0, fb = (float)z1;  // (the last voxel is z1 - 1: one step of slack)
    const RunEnd a = run_end(fmaf(fa, zstep[0], p0x), fmaf(fa, zstep[1], p0y), fmaf(fa, zstep[2], p0z), cc, rcp);
    const RunEnd b = run_end(fmaf(fb, zstep[0], p0x), fmaf(fb, zstep[1], p0y), fmaf(fb, zstep[2], p0z), cc, rcp);
    return classify_run(a, b, cc, h2f) == RUN_SKIP;
}

}  // namespace dfa
