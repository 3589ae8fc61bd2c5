// tsdf.hip — gfx950 kernels for the TSDF seam: compute_dists, clear, integrate,
// fused clear+integrate, raycast (points / depth).
//
// What they compute is kfusion's src/kfusion/cuda/tsdf_volume.cu + imgproc.cu:233-245
// (cited per kernel); how they are laid out is MI355X-specific:
//   * the volume is x-fastest, 4 B / voxel; one lane owns one voxel COLUMN and marches it over z, a 256-thread
//     block covers 64 x 4 columns, z is cut into chunks (grid.z) until >= 4 096 workgroups are in flight;
//   * the reference's running `vc += zstep` (tsdf_volume.cu:64) is kept bit-for-bit: a chunk that starts at
//     slice z0 replays the z0 additions in registers first rather than using z0*zstep;
//   * the sweeps classify runs of 8 voxels of a column at once (tsdf_classify.hpp: skipped / in front of the
//     surface / per-voxel arithmetic) from one projection and four min-max tiles of the depth image — most of a
//     volume never sees a division, a gather or a square root, and the fused clear+integrate sweep runs at the
//     speed of its stores;
//   * the depth ("dists") image (600 KiB at VGA) and its tile table (19 KiB) stay in L2 / L1 while the volume
//     streams past them.
// Every kernel is HBM-bound integer/half work; nothing here is GEMM-shaped, no MFMA.
#include <hip/hip_runtime.h>

#include <mutex>
#include <map>

#include "device_math.hpp"
#include "dev_switch.hpp"
#include "kernels.hpp"
#include "tsdf_classify.hpp"

namespace dfa {

// ------------------------------------------------------------------------------------------
// compute_dists — imgproc.cu:233-245.  2 pixels per lane (one dword in, one dword out).
__global__ __launch_bounds__(256) void compute_dists_kernel(const uint16_t* __restrict__ depth, int depth_step,
                                                            uint16_t* __restrict__ dists, int dists_step, int cols,
                                                            int rows, float finvx, float finvy, float cx, float cy) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const uint16_t* drow = (const uint16_t*)((const char*)depth + (size_t)y * depth_step);
    uint16_t* orow       = (uint16_t*)((char*)dists + (size_t)y * dists_step);
    float xl             = ((float)x - cx) * finvx;
    float yl             = ((float)y - cy) * finvy;
    float lambda         = sqrtf(fmaf(yl, yl, xl * xl) + 1.f);
    orow[x]              = (uint16_t)float_to_half_bits(((float)drow[x] * lambda) * 0.001f);
}

// ------------------------------------------------------------------------------------------
// clear — tsdf_volume.cu:11-22: pack_tsdf(0.f, 0) == 0.  Grid-stride 16-byte stores.
__global__ __launch_bounds__(256) void clear_kernel(uint4* __restrict__ vol4, size_t n4, uint32_t* __restrict__ tail,
                                                    int ntail) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        vol4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0u;
}

// The store pattern of the fused sweep's zero fill (a lane per voxel column, a wave stores one 256-byte x-row segment per
// slice and marches over z): measured faster than the grid-stride 16-byte stores above at 512^3 and 1024^3 (DESIGN.md
// 4.1), so clear() uses it for volumes with whole 64-voxel rows.
__global__ __launch_bounds__(256) void clear_columns_kernel(uint32_t* __restrict__ vol, int X, int Y, int Z, int zchunk) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= X || y >= Y) return;
    const int z0 = blockIdx.z * zchunk, z1 = min(Z, z0 + zchunk);
    const size_t slice = (size_t)X * Y;
    uint32_t* p        = vol + (size_t)z0 * slice + (size_t)y * X + x;
    int z              = z0;
    for (; z + 8 <= z1; z += 8, p += 8 * slice) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u * slice] = 0u;
    }
    for (; z < z1; ++z, p += slice) *p = 0u;
}

// ------------------------------------------------------------------------------------------
// integrate — tsdf_volume.cu:43-96
struct IntegrateArgs {
    const uint16_t* dists;
    int dists_step, cols, rows;
    uint32_t* vol;
    int X, Y, Z;
    float vsx, vsy, vsz;
    float trunc, trunc_inv;
    int max_weight;
    Aff3 vol2cam;
    float fx, fy, cx, cy;
    int zchunk;
    uint8_t* occ;  // occupancy map (kernels.hpp: OccDims) or null
    int ox, oy;
    int occ_known;  // fused sweep: the map describes the volume as it is NOW — a box without weights that gets none is not written
    int chunk_rule;  // 1: a wave first asks whether its whole z chunk is skipped (tsdf_classify.hpp: chunk_skipped)
    int nt;      // DFA_TSDF_NT=1 (A/B): non-temporal stores in the fused sweep
    int ablate;  // -DDFA_DEV_ABLATE builds only (DFA_TSDF_ABLATE): 1 every run SKIP, 2 FULL runs filled like FRONT, 3 no classification
};

// x / z and y / z, correctly rounded.  hipcc expands an fp32 division into
//   s0 = div_scale(den), s1 = div_scale(num), r = rcp(s0), e = fma(-s0, r, 1), r1 = fma(e, r, r), q = s1 * r1,
//   e2 = fma(-s0, q, s1), q1 = fma(e2, r1, q), e3 = fma(-s0, q1, s1), div_fixup(div_fmas(e3, r1, q1))
// (11 instructions; div_scale / div_fmas / div_fixup only act on operands near the ends of the exponent range).  For
// operands well inside the range the two quotients share everything that depends on z alone — the same instructions
// on the same values, hence the same bits, in 13 instead of 22; anything else takes the plain divisions.
__device__ __forceinline__ void div_xy_by_z(float x, float y, float z, float& qx, float& qy) {
    const float big = fmaxf(fmaxf(fabsf(x), fabsf(y)), z);
    if (big <= 1048576.f && z >= 9.5367431640625e-07f) {  // 2^20, 2^-20 (z > 0 here)
        const float r  = __builtin_amdgcn_rcpf(z);
        const float r1 = fmaf(fmaf(-z, r, 1.0f), r, r);
        float q        = x * r1;
        q              = fmaf(fmaf(-z, q, x), r1, q);
        qx             = fmaf(fmaf(-z, q, x), r1, q);
        q              = y * r1;
        q              = fmaf(fmaf(-z, q, y), r1, q);
        qy             = fmaf(fmaf(-z, q, y), r1, q);
    } else {
        qx = x / z, qy = y / z;
    }
}

// One voxel of one slice, first half (tsdf_volume.cu:65-80): false when the reference leaves the voxel alone, else the
// truncated signed distance of this frame.
__device__ __forceinline__ bool voxel_tsdf(const IntegrateArgs& a, f3 vc, float& tsdf) {
    // :74 `vc.z <= 0` is tested first here: the reference tests it after the (side-effect
    // free) projection and texture fetch, the outcome is the same and NaN/inf never form.
    if (!(vc.z > 0.f)) return false;
    // Projector (device.hpp:40-45): correctly rounded divisions stand in for __fdividef
    float qx, qy;
    div_xy_by_z(vc.x, vc.y, vc.z, qx, qy);
    const float coox = fmaf(a.fx, qx, a.cx);
    const float cooy = fmaf(a.fy, qy, a.cy);
    if (!(coox >= 0.f && cooy >= 0.f && coox < (float)a.cols && cooy < (float)a.rows)) return false;  // :70
    // :73 point-sampled, un-normalised texture fetch == texel (floor x, floor y); coordinates
    // are non-negative here so the truncating convert is the floor
    const int px         = (int)coox;
    const int py         = (int)cooy;
    const uint16_t* drow = (const uint16_t*)((const char*)a.dists + (size_t)py * a.dists_step);
    const float Dp       = half_bits_to_float(drow[px]);
    if (Dp == 0.f) return false;                    // :74
    // Voxels far behind the surface (a third of the volume) leave before the correctly rounded square root: when
    // |vc|^2 exceeds (Dp + trunc)^2 by more than 1e-5 relative, sqrt exceeds Dp + trunc by 5e-6 relative — two orders
    // above the rounding of the three operations below, so the test of :79 fails for certain.
    const float d2  = dot(vc, vc);
    const float lim = Dp + a.trunc;
    if (d2 > lim * lim * 1.00001f) return false;
    const float sdf = Dp - sqrtf(d2);               // :77
    if (!(sdf >= -a.trunc)) return false;           // :79
    tsdf = fminf(1.f, sdf * a.trunc_inv);           // :80
    return true;
}

// second half (:82-90): running average with the voxel's previous state (`old` packed; 0 when the clear is fused)
template <bool FUSED_CLEAR>
__device__ __forceinline__ uint32_t voxel_update(const IntegrateArgs& a, uint32_t old, float tsdf) {
    int weight_prev;
    float tsdf_prev;
    if (FUSED_CLEAR) {
        weight_prev = 0;
        tsdf_prev   = 0.f;
    } else {
        weight_prev = (int)(old >> 16);
        tsdf_prev   = unpack_tsdf(old);
    }
    const float tsdf_new = fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1);  // :86
    const int weight_new = min(weight_prev + 1, a.max_weight);                                   // :87
    return pack_tsdf(tsdf_new, weight_new);
}

// One voxel of one slice (tsdf_volume.cu:65-91).  `old` is the packed voxel (0 when the clear
// is fused); returns the packed voxel after the update and sets `changed`.
template <bool FUSED_CLEAR>
__device__ __forceinline__ uint32_t integrate_voxel(const IntegrateArgs& a, f3 vc, uint32_t old, bool& changed) {
    float tsdf;
    if (!voxel_tsdf(a, vc, tsdf)) return old;
    changed = true;
    return voxel_update<FUSED_CLEAR>(a, old, tsdf);
}

template <int VX>
struct VoxVec;
template <>
struct VoxVec<4> {
    uint4 v;
    __device__ __forceinline__ uint32_t get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
    __device__ __forceinline__ void set(int i, uint32_t x) {
        if (i == 0) v.x = x;
        else if (i == 1) v.y = x;
        else if (i == 2) v.z = x;
        else v.w = x;
    }
    __device__ __forceinline__ void load(const uint32_t* p) { v = *(const uint4*)p; }
    __device__ __forceinline__ void store(uint32_t* p) const { *(uint4*)p = v; }
    __device__ __forceinline__ void zero() { v = make_uint4(0u, 0u, 0u, 0u); }
};
template <>
struct VoxVec<1> {
    uint32_t v;
    __device__ __forceinline__ uint32_t get(int) const { return v; }
    __device__ __forceinline__ void set(int, uint32_t x) { v = x; }
    __device__ __forceinline__ void load(const uint32_t* p) { v = *p; }
    __device__ __forceinline__ void store(uint32_t* p) const { *p = v; }
    __device__ __forceinline__ void zero() { v = 0u; }
};

// block = (64, 4): a wave spans 64*VX voxels in x, the block 4 rows in y; grid.z = z chunks.
template <bool FUSED_CLEAR, int VX>
__global__ __launch_bounds__(256) void integrate_kernel(const IntegrateArgs a) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * VX;
    const int y  = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= a.X || y >= a.Y) return;
    const int z0 = blockIdx.z * a.zchunk;
    const int z1 = min(z0 + a.zchunk, a.Z);

    // :58
    const f3 zstep = mk3(a.vol2cam.m[2], a.vol2cam.m[5], a.vol2cam.m[8]) * a.vsz;
    const f3 t     = mk3(a.vol2cam.t[0], a.vol2cam.t[1], a.vol2cam.t[2]);
    f3 vc[VX];
#pragma unroll
    for (int v = 0; v < VX; ++v) {
        const f3 vx = mk3((float)(x0 + v) * a.vsx, (float)y * a.vsy, 0.f);  // :60
        vc[v]       = mulR(a.vol2cam, vx) + t;                              // :61
    }
    // replay the z0 running additions of :64 so the chunk starts on the reference's value
    for (int i = 0; i < z0; ++i) {
#pragma unroll
        for (int v = 0; v < VX; ++v) vc[v] = vc[v] + zstep;
    }

    const size_t slice = (size_t)a.X * a.Y;
    uint32_t* ptr      = a.vol + (size_t)x0 + (size_t)a.X * y + slice * z0;

    constexpr int U = 4;  // slices in flight per lane
    int z           = z0;
    for (; z + U <= z1; z += U, ptr += slice * U) {
        VoxVec<VX> cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (FUSED_CLEAR) cur[u].zero();
            else cur[u].load(ptr + slice * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool changed = FUSED_CLEAR;  // the fused sweep writes every voxel
#pragma unroll
            for (int v = 0; v < VX; ++v) {
                cur[u].set(v, integrate_voxel<FUSED_CLEAR>(a, vc[v], cur[u].get(v), changed));
                vc[v] = vc[v] + zstep;  // :64, also for skipped voxels
            }
            if (changed) cur[u].store(ptr + slice * u);
        }
    }
    for (; z < z1; ++z, ptr += slice) {
        VoxVec<VX> cur;
        if (FUSED_CLEAR) cur.zero();
        else cur.load(ptr);
        bool changed = FUSED_CLEAR;
#pragma unroll
        for (int v = 0; v < VX; ++v) {
            cur.set(v, integrate_voxel<FUSED_CLEAR>(a, vc[v], cur.get(v), changed));
            vc[v] = vc[v] + zstep;
        }
        if (changed) cur.store(ptr);
    }
}

// ------------------------------------------------------------------------------------------
// Run-classified sweeps (tsdf_classify.hpp).  Same result as integrate_kernel, voxel for voxel; what changes is the
// work: per run of U slices a lane projects ONE point (the far end of the run; the near end is the previous run's
// far end), looks up four min / max tiles of the dists image and knows whether its U voxels are all skipped, all
// updated with tsdf == 1, or need the reference's per-voxel arithmetic.  At C2 (512^3, U = 8) 83 % of the runs are
// skipped, 10 % are in front of the surface, 7 % take the per-voxel path; a wave does when one of its lanes does.
//   * fused clear + integrate: skipped runs store zeros, front runs a constant — the sweep becomes a store stream;
//   * read + write sweep: skipped runs touch no memory at all.
// A wave covers WX x (64 / WX) columns, a block 64 x 4.  Measured (tools/tsdf_sweep.py, fused sweep, 512^3 / 1024^3):
//   per-voxel kernel 0.222 / 1.174 ms; U = 4, WX = 64: 0.157 / 0.883; U = 8, WX = 64: 0.137 / 0.825;
//   U = 8, WX = 32: 0.117 / 0.721 (128-byte row segments, and lanes that agree more often: 14 % instead of 19 % of
//   the wave runs hold a lane on the per-voxel path); U = 8, WX = 16: 0.161 / 1.236 (64-byte segments: half cache
//   lines); the same loop storing zeros only: 0.105 / 0.773.
constexpr int RUN_U = 8;

__device__ __forceinline__ float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float half_bits_to_float_u(uint32_t b) { return half_bits_to_float(b); }

// min / max tiles of the dists image: one wave per 8 x 8 tile
__global__ __launch_bounds__(256) void dists_tiles_kernel(const uint16_t* __restrict__ dists, int dists_step, int cols,
                                                          int rows, uint32_t* __restrict__ tiles, int tcols, int ntiles) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int lane = threadIdx.x & 63;
    const int x = (tile % tcols) * 8 + (lane & 7), y = (tile / tcols) * 8 + (lane >> 3);
    uint32_t lo = 0xffffu, hi = 0u;
    if (x < cols && y < rows) {
        const uint16_t* drow = (const uint16_t*)((const char*)dists + (size_t)y * dists_step);
        const uint32_t t = tile_bounds_of_pixel(drow[x]);
        lo = t & 0xffffu, hi = t >> 16;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
    if (lane == 0) tiles[tile] = lo | (hi << 16);
}

template <bool FUSED_CLEAR, int WX, int U>
__global__ __launch_bounds__(256) void integrate_runs_kernel(const IntegrateArgs a, const RunConsts rc, const RunConsts rcc,
                                                             const uint32_t front_const) {
    const int lane  = threadIdx.x, wave = threadIdx.y;  // block (64, 4)
    // the block's 64 x 4 columns, WX x (64 / WX) per wave: waves side by side in x, then stacked in y
    constexpr int WAVES_X = 64 / WX, WAVE_ROWS = 64 / WX;
    const int x = blockIdx.x * 64 + (wave % WAVES_X) * WX + (lane % WX);
    const int y = blockIdx.y * 4 + (wave / WAVES_X) * WAVE_ROWS + lane / WX;
    if (x >= a.X || y >= a.Y) return;
    const int z0 = blockIdx.z * a.zchunk;
    const int z1 = min(z0 + a.zchunk, a.Z);

    const f3 zstep = mk3(a.vol2cam.m[2], a.vol2cam.m[5], a.vol2cam.m[8]) * a.vsz;                  // :58
    const f3 vx    = mk3((float)x * a.vsx, (float)y * a.vsy, 0.f);                                   // :60
    f3 vc          = mulR(a.vol2cam, vx) + mk3(a.vol2cam.t[0], a.vol2cam.t[1], a.vol2cam.t[2]);      // :61
    // (all of a chunk's map bytes up front, as a bit mask: a load per run in front of the decision is a memory round trip per run)
    constexpr int WAVE_ROWS_ = 64 / WX;
    const size_t occ_layer = (size_t)a.ox * a.oy;
    // bit r of (clean_hi : clean): the box of the chunk's r-th run held zeros on entry (fused sweep over a known map); 128 runs
    // are a whole column of a 1024^3 volume, runs beyond are not known to be clean
    unsigned long long clean = 0ull, clean_hi = 0ull;
    const int nruns          = (z1 - z0) / U;
    if (FUSED_CLEAR && a.occ && a.occ_known) {
        const uint8_t* occ_old = a.occ + (size_t)(x / WX) + (size_t)a.ox * ((size_t)(y / WAVE_ROWS_) + (size_t)a.oy * (size_t)(z0 / U));
#pragma unroll 8
        for (int r = 0; r < min(nruns, 64); ++r) clean |= (unsigned long long)(occ_old[(size_t)r * occ_layer] == 0) << r;
#pragma unroll 8
        for (int r = 64; r < min(nruns, 128); ++r) clean_hi |= (unsigned long long)(occ_old[(size_t)r * occ_layer] == 0) << (r - 64);
    }
    // The chunk as a whole first (tsdf_classify.hpp, chunk_skipped): when every column of the wave skips every voxel of it —
    // half of a volume lies outside the frustum — there is nothing to classify or replay; the accumulating sweep leaves such
    // voxels alone anyway, the fused sweep may when the map says they are zeros already (and the chunk has no tail).
    auto ones = [](int n) { return n >= 64 ? ~0ull : n <= 0 ? 0ull : (1ull << n) - 1ull; };
    if (a.chunk_rule && (!FUSED_CLEAR || (a.occ && a.occ_known && nruns <= 128 && nruns * U == z1 - z0 && clean == ones(nruns) &&
                                          clean_hi == ones(nruns - 64)))) {
        const float zs[3] = {zstep.x, zstep.y, zstep.z};
        const bool skip   = chunk_skipped(vc.x, vc.y, vc.z, zs, z0, z1, rcc, rcp_approx, half_bits_to_float_u);
        if (__ballot(!skip) == 0ull) return;  // (a skipped chunk leaves the map's bytes as they are: nothing gained a weight)
    }
    for (int i = 0; i < z0; ++i) vc = vc + zstep;  // replay :64 up to the chunk's first slice

    const size_t slice = (size_t)a.X * a.Y;
    uint32_t* ptr      = a.vol + (size_t)x + (size_t)a.X * y + slice * z0;
    const f3 stepU     = mk3(rc.stepU[0], rc.stepU[1], rc.stepU[2]);

    int z      = z0;
    RunEnd end = run_end(vc.x, vc.y, vc.z, rc, rcp_approx);
    // occupancy map: a byte per (this wave's WX x (64 / WX) columns) x (run of 8 slices) — the launcher passes it only for
    // WX = 32, U = 8 and chunks that start on a multiple of 8.  Written by the first live lane of the wave.
    uint8_t* occ_cell = nullptr;
    if (a.occ) {
        const unsigned long long live = __ballot(1);
        if (lane == (int)__ffsll((long long)live) - 1)
            occ_cell = a.occ + (size_t)(x / WX) + (size_t)a.ox * ((size_t)(y / WAVE_ROWS) + (size_t)a.oy * (size_t)(z0 / U));
    }
    // The fused sweep over a volume whose map is KNOWN to describe it (dfa_tsdf_clear_integrate_known_occ): a box whose byte
    // is 0 holds 32 x 2 x 8 zeros, and when none of its runs gets a weight now either, storing those zeros again is the
    // 5/6 of the sweep's traffic that changes nothing (`clean`, read above).
    for (; z + U <= z1; z += U, ptr += slice * U) {
        const bool was_clean = (clean & 1ull) != 0ull;
        clean = (clean >> 1) | (clean_hi << 63), clean_hi >>= 1;
        const f3 far     = vc + stepU;
        int cls          = RUN_SKIP;
#ifdef DFA_DEV_ABLATE
        if (a.ablate != 3) {
            const RunEnd nxt = run_end(far.x, far.y, far.z, rc, rcp_approx);
            cls              = classify_run(end, nxt, rc, half_bits_to_float_u);
            end              = nxt;
            if (a.ablate == 1) cls = RUN_SKIP;
            if (a.ablate == 2 && cls == RUN_FULL) cls = RUN_FRONT;
        }
#else
        {
            const RunEnd nxt = run_end(far.x, far.y, far.z, rc, rcp_approx);
            cls              = classify_run(end, nxt, rc, half_bits_to_float_u);
            end              = nxt;
        }
#endif
        bool untouched = false;  // (wave-uniform) a box of zeros that stays one
        if (a.occ) {  // (uniform) bit 0: a run of the box was not SKIP (SKIP runs are the only ones that leave, or find, no
                      // weight); bit 1: a run was FULL — the only runs that can leave a NEGATIVE distance (FRONT runs write +1)
            const unsigned mark = (__ballot(cls != RUN_SKIP) != 0ull ? 1u : 0u) | (__ballot(cls == RUN_FULL) != 0ull ? 2u : 0u);
            untouched           = FUSED_CLEAR && was_clean && mark == 0u;
            if (occ_cell) {
                if (FUSED_CLEAR) {
                    if (!untouched) *occ_cell = (uint8_t)mark;
                } else if (mark) *occ_cell = (uint8_t)(*occ_cell | mark);  // (one writer per byte and launch)
                occ_cell += occ_layer;
            }
        }
        f3 p[U];  // the run's voxel positions by the reference's running addition (:64)
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = vc, vc = vc + zstep;
        if (FUSED_CLEAR) {
            if (untouched) continue;
            uint32_t out[U];
            const uint32_t fill = cls == RUN_FRONT ? front_const : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) out[u] = fill;
            if (cls == RUN_FULL) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool changed;
                    out[u] = integrate_voxel<true>(a, p[u], 0u, changed);
                }
            }
            if (a.nt) {
#pragma unroll
                for (int u = 0; u < U; ++u) __builtin_nontemporal_store(out[u], ptr + slice * u);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) ptr[slice * u] = out[u];
            }
        } else if (cls != RUN_SKIP) {
            uint32_t cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = ptr[slice * u];
            if (cls == RUN_FRONT) {
#pragma unroll
                for (int u = 0; u < U; ++u) ptr[slice * u] = voxel_update<false>(a, cur[u], 1.0f);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool changed    = false;
                    const uint32_t v = integrate_voxel<false>(a, p[u], cur[u], changed);
                    if (changed) ptr[slice * u] = v;
                }
            }
        }
    }
    if (occ_cell && z < z1) *occ_cell = 3;  // (the slices of a tail: marked without looking)
    for (; z < z1; ++z, ptr += slice) {  // tail shorter than a run: per voxel
        bool changed     = FUSED_CLEAR;
        const uint32_t v = integrate_voxel<FUSED_CLEAR>(a, vc, FUSED_CLEAR ? 0u : *ptr, changed);
        if (changed) *ptr = v;
        vc = vc + zstep;
    }
}

// ------------------------------------------------------------------------------------------
// raycast — tsdf_volume.cu:128-337
struct RaycastArgs {
    const uint32_t* vol;
    int X, Y, Z;
    float vsx, vsy, vsz;        // voxel size
    float vix, viy, viz;        // 1 / voxel size          (:362)
    float sx, sy, sz;           // volume size = voxel*dims (:359)
    float gdx, gdy, gdz;        // gradient delta           (:361)
    float time_step;            // trunc * step_factor      (:360)
    Aff3 cam2vol;
    Mat3 Rinv;
    float finvx, finvy, cx, cy;  // Reprojector
    int cols, rows;
};

__device__ __forceinline__ float qnan() { return __uint_as_float(0x7fffffffu); }  // temp_utils.hpp:22

// Work counters of one raycast (dfa_tsdf_raycast_tally, a measurement entry point: SURVEY 8(d) prices the raycast by
// rays x steps x 4 B + hits x 64 x 4 B and by the number of distinct voxels touched).  The product kernels are the
// TALLY = false instantiations: no counter exists in them.
struct RayTally {
    unsigned long long* counts;  // [0] rays that enter the box, [1] nearest-voxel fetches of the march, [2] hits,
                                 // [3] voxel fetches of the trilinear samples (8 per sample inside the volume)
    uint32_t* touched;           // one bit per voxel (X*Y*Z / 32 words, zeroed by the caller) or null
    unsigned int entered, march, hits, tri;
    __device__ __forceinline__ void touch(size_t voxel) const {
        if (touched) atomicOr(&touched[voxel >> 5], 1u << (voxel & 31));
    }
};

// :187-193 nearest voxel (round-half-even).  The clamp is memory safety only: rays are kept
// inside [0, size - voxel] by the slab test.
// IDX32: the volume has at most 2^32 voxels (every BASELINE size; 1024^3 = 2^30): the voxel index is formed in 32-bit
// arithmetic and widened once — 4 vector instructions for the address instead of 12.
template <bool IDX32>
__device__ __forceinline__ size_t voxel_index(const RaycastArgs& a, int x, int y, int z) {
    if constexpr (IDX32) return (size_t)((uint32_t)x + (uint32_t)a.X * ((uint32_t)y + (uint32_t)a.Y * (uint32_t)z));
    else return (size_t)x + (size_t)a.X * y + (size_t)a.X * a.Y * z;
}

template <bool TALLY = false, bool IDX32 = false>
__device__ __forceinline__ float fetch_tsdf(const RaycastArgs& a, f3 p, RayTally* tally = nullptr) {
    int x = (int)rintf(p.x * a.vix);
    int y = (int)rintf(p.y * a.viy);
    int z = (int)rintf(p.z * a.viz);
    x     = min(max(x, 0), a.X - 1);
    y     = min(max(y, 0), a.Y - 1);
    z     = min(max(z, 0), a.Z - 1);
    const size_t voxel = voxel_index<IDX32>(a, x, y, z);
    if constexpr (TALLY) {
        tally->march++;
        tally->touch(voxel);
    }
    return unpack_tsdf(a.vol[voxel]);
}

// :146-171 trilinear interpolation, voxel centres at integer coordinates.  The eight fetches are UNCONDITIONAL (from
// voxel 0 when the sample lies outside the interpolation range, the result then replaced by the reference's NaN): no
// branch separates the samples of a hit, so the two samples of the crossing go out as one batch of 16 fetches and the six
// of the normal as one of 48 — two memory round trips per hit where a branch per sample made eight.
template <bool TALLY = false>
__device__ __forceinline__ float interpolate(const RaycastArgs& a, f3 cf, RayTally* tally = nullptr) {
    const bool inside = cf.x >= 0.f && cf.x < (float)(a.X - 1) && cf.y >= 0.f && cf.y < (float)(a.Y - 1) && cf.z >= 0.f &&
                        cf.z < (float)(a.Z - 1);
    const f3 c = inside ? cf : mk3(0.f, 0.f, 0.f);
    const int gx = (int)c.x, gy = (int)c.y, gz = (int)c.z;  // floor of a non-negative value
    const float fa = c.x - (float)gx, fb = c.y - (float)gy, fc = c.z - (float)gz;
    const size_t sy = (size_t)a.X, sz = (size_t)a.X * a.Y;
    const uint32_t* b = a.vol + (size_t)gx + sy * gy + sz * gz;
    if constexpr (TALLY) {
        if (inside) {
            tally->tri += 8;
            const size_t v0 = (size_t)(b - a.vol);
            for (int c8 = 0; c8 < 8; ++c8) tally->touch(v0 + (c8 & 1) + (c8 & 2 ? sy : 0) + (c8 & 4 ? sz : 0));
        }
    }
    const float v000 = unpack_tsdf(b[0]), v001 = unpack_tsdf(b[sz]);
    const float v010 = unpack_tsdf(b[sy]), v011 = unpack_tsdf(b[sy + sz]);
    const float v100 = unpack_tsdf(b[1]), v101 = unpack_tsdf(b[1 + sz]);
    const float v110 = unpack_tsdf(b[1 + sy]), v111 = unpack_tsdf(b[1 + sy + sz]);
    float tsdf = 0.f;
    tsdf = fmaf((v000 * (1.f - fa)) * (1.f - fb), (1.f - fc), tsdf);
    tsdf = fmaf((v001 * (1.f - fa)) * (1.f - fb), fc, tsdf);
    tsdf = fmaf((v010 * (1.f - fa)) * fb, (1.f - fc), tsdf);
    tsdf = fmaf((v011 * (1.f - fa)) * fb, fc, tsdf);
    tsdf = fmaf((v100 * fa) * (1.f - fb), (1.f - fc), tsdf);
    tsdf = fmaf((v101 * fa) * (1.f - fb), fc, tsdf);
    tsdf = fmaf((v110 * fa) * fb, (1.f - fc), tsdf);
    tsdf = fmaf((v111 * fa) * fb, fc, tsdf);
    return inside ? tsdf : qnan();
}

// :320-336
template <bool TALLY = false>
__device__ __forceinline__ f3 compute_normal(const RaycastArgs& a, f3 p, RayTally* tally = nullptr) {
    const f3 vi = mk3(a.vix, a.viy, a.viz);
    f3 n;
    const float Fx1 = interpolate<TALLY>(a, mk3(p.x + a.gdx, p.y, p.z) * vi, tally);
    const float Fx2 = interpolate<TALLY>(a, mk3(p.x - a.gdx, p.y, p.z) * vi, tally);
    n.x             = (Fx1 - Fx2) / a.gdx;
    const float Fy1 = interpolate<TALLY>(a, mk3(p.x, p.y + a.gdy, p.z) * vi, tally);
    const float Fy2 = interpolate<TALLY>(a, mk3(p.x, p.y - a.gdy, p.z) * vi, tally);
    n.y             = (Fy1 - Fy2) / a.gdy;
    const float Fz1 = interpolate<TALLY>(a, mk3(p.x, p.y, p.z + a.gdz) * vi, tally);
    const float Fz2 = interpolate<TALLY>(a, mk3(p.x, p.y, p.z - a.gdz) * vi, tally);
    n.z             = (Fz1 - Fz2) / a.gdz;
    return normalized(n);
}

// march steps whose voxels are requested together (measured at 512^3 / VGA and 1024^3 / 720p: 1 step 0.082 / 0.219 ms,
// 2: 0.063 / 0.158, 4: 0.057 / 0.140, 6: 0.060 / 0.143, 8: 0.062 / 0.148 in the first batched form)
#ifndef DFA_RAY_BATCH  // (compile-time A/B: tools/ab_variant.sh rb8 tsdf.hip -DDFA_RAY_BATCH=8)
#define DFA_RAY_BATCH 4
#endif
constexpr int RAY_BATCH = DFA_RAY_BATCH;

// shared body of the two TsdfRaycaster::operator() overloads (:195-318)
template <bool TALLY = false, bool IDX32 = false>
__device__ __forceinline__ bool cast_ray(const RaycastArgs& a, int x, int y, f3& vertex_cam, f3& normal_cam,
                                         RayTally* tally = nullptr) {
    const f3 ray_org = mk3(a.cam2vol.t[0], a.cam2vol.t[1], a.cam2vol.t[2]);
    const f3 pix     = mk3((1.f * ((float)x - a.cx)) * a.finvx, (1.f * ((float)y - a.cy)) * a.finvy, 1.f);
    const f3 ray_dir = normalized(mulR(a.cam2vol, pix));
    const f3 box_max = mk3(a.sx - a.vsx, a.sy - a.vsy, a.sz - a.vsz);  // :213
    // intersect (:128-144), including the reference's asymmetric max/min
    const f3 invR = mk3(1.f / ray_dir.x, 1.f / ray_dir.y, 1.f / ray_dir.z);
    const f3 tbot = invR * (mk3(0.f, 0.f, 0.f) - ray_org);
    const f3 ttop = invR * (box_max - ray_org);
    const f3 tmn  = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    const f3 tmx  = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    float tmin    = fmaxf(fmaxf(tmn.x, tmn.y), fmaxf(tmn.x, tmn.z));
    float tmax    = fminf(fminf(tmx.x, tmx.y), fminf(tmx.x, tmx.z));
    tmin          = fmaxf(0.f, tmin);  // :219
    if (!(tmin < tmax)) return false;  // :220
    if constexpr (TALLY) tally->entered++;
    tmax -= a.time_step;
    const f3 vstep  = ray_dir * a.time_step;
    f3 next         = ray_org + ray_dir * tmin;
    float tsdf_next = fetch_tsdf<TALLY, IDX32>(a, next, tally);
    const f3 vi     = mk3(a.vix, a.viy, a.viz);
    // The march (:222-256) in batches of RAY_BATCH steps: the positions of the next RAY_BATCH samples — the same running
    // `next += vstep` additions — are computed and their voxels requested TOGETHER (a memory round trip per batch instead
    // of per step), as are the running `tcurr += time_step` sums.  Whether ANY of the batch's steps ends the march — the
    // reference's two sign tests (:234, :237) or its loop condition — takes a few compares; only a batch that holds an
    // event is then walked step by step, in the reference's order, to find the first one.  The fetches behind the exit
    // are speculative (clamped addresses, at most RAY_BATCH - 1 per ray) and their values unused.
    bool hit = false;
    f3 hit_curr = next, hit_next = next;
    float hit_t = 0.f;
    if (!(tmin < tmax)) return false;  // the loop condition before the first step
    for (float tcurr = tmin;;) {
        f3 pos[RAY_BATCH];
        float val[RAY_BATCH], tc[RAY_BATCH + 1];
        pos[0] = next + vstep;
        tc[0]  = tcurr;
#pragma unroll
        for (int j = 1; j < RAY_BATCH; ++j) pos[j] = pos[j - 1] + vstep;
#pragma unroll
        for (int j = 0; j < RAY_BATCH; ++j) tc[j + 1] = tc[j] + a.time_step;
#pragma unroll
        for (int j = 0; j < RAY_BATCH; ++j) val[j] = fetch_tsdf<false, IDX32>(a, pos[j]);  // (tallied below, per step taken)
        bool event = false;
#pragma unroll
        for (int j = 0; j < RAY_BATCH; ++j) {
            const float c = j ? val[j - 1] : tsdf_next, n = val[j];
            event |= (c < 0.f && n > 0.f) || (c > 0.f && n < 0.f) || !(tc[j + 1] < tmax);
        }
        if (event) {
#pragma unroll
            for (int j = 0; j < RAY_BATCH; ++j) {
                const float c = j ? val[j - 1] : tsdf_next, n = val[j];
                if constexpr (TALLY) (void)fetch_tsdf<true, IDX32>(a, pos[j], tally);
                if (c < 0.f && n > 0.f) break;  // :234
                if (c > 0.f && n < 0.f) {       // :237
                    hit      = true;
                    hit_curr = j ? pos[j - 1] : next, hit_next = pos[j], hit_t = tc[j];
                    break;
                }
                if (!(tc[j + 1] < tmax)) break;  // the loop condition
            }
            break;
        }
        if constexpr (TALLY)
            for (int j = 0; j < RAY_BATCH; ++j) (void)fetch_tsdf<true, IDX32>(a, pos[j], tally);
        next = pos[RAY_BATCH - 1], tsdf_next = val[RAY_BATCH - 1], tcurr = tc[RAY_BATCH];
    }
    if (hit) {
        const float Ft   = interpolate<TALLY>(a, hit_curr * vi, tally);
        const float Ftdt = interpolate<TALLY>(a, hit_next * vi, tally);
        const float Ts   = hit_t - (a.time_step * Ft) / (Ftdt - Ft);  // :241
        const f3 vertex  = ray_org + ray_dir * Ts;
        const f3 normal  = compute_normal<TALLY>(a, vertex, tally);
        const float prod = normal.x * normal.y * normal.z;
        if (prod == prod) {  // :246 !isnan
            if constexpr (TALLY) tally->hits++;
            normal_cam = mul(a.Rinv, normal);
            vertex_cam = mul(a.Rinv, vertex - ray_org);
            return true;
        }
    }
    return false;
}

// A wave covers an 8x8 pixel tile (rays of a tile walk neighbouring voxels -> shared cache lines); a 256-thread block
// covers 16x16 pixels.  Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: within every round of
// 64 tiles XCD i takes 8 CONSECUTIVE tiles — neighbours along x share their 64-byte voxel lines (16 voxels = ~32 pixels
// at 1.5 m and 512^3), so the line is fetched into one L2 instead of two to eight (HBM fetch 107 -> 69 MB per VGA launch
// at 512^3, L2 hit rate 21 -> 48 %, 0.051 -> 0.042 ms; contiguous bands per XCD fetch even less — 50 MB — but leave the XCDs
// with unequal work: slower).  The last, partial round keeps the identity order.
constexpr int RAY_XCD_GROUP = 8;
__device__ __forceinline__ void tile_pixel(int& x, int& y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const int full = nb / (8 * RAY_XCD_GROUP) * (8 * RAY_XCD_GROUP);
    int t = b;
    if (b < full) {
        const int round = b / (8 * RAY_XCD_GROUP), r = b % (8 * RAY_XCD_GROUP);
        t = round * 8 * RAY_XCD_GROUP + (r & 7) * RAY_XCD_GROUP + (r >> 3);
    }
    const int bx = t % gridDim.x, by = t / gridDim.x;
    x = bx * 16 + (wave & 1) * 8 + (lane & 7);
    y = by * 16 + (wave >> 1) * 8 + (lane >> 3);
}

template <bool IDX32>
__global__ __launch_bounds__(256) void raycast_points_kernel(const RaycastArgs a, float* __restrict__ points,
                                                             int points_step, float* __restrict__ normals,
                                                             int normals_step) {
    int x, y;
    tile_pixel(x, y);
    if (x >= a.cols || y >= a.rows) return;
    float4* prow = (float4*)((char*)points + (size_t)y * points_step);
    float4* nrow = (float4*)((char*)normals + (size_t)y * normals_step);
    f3 v, n;
    if (cast_ray<false, IDX32>(a, x, y, v, n)) {
        prow[x] = make_float4(v.x, v.y, v.z, 0.f);  // :312-313
        nrow[x] = make_float4(n.x, n.y, n.z, 0.f);
    } else {
        const float q = qnan();
        prow[x] = nrow[x] = make_float4(q, q, q, q);  // :267
    }
}

template <bool IDX32>
__global__ __launch_bounds__(256) void raycast_depth_kernel(const RaycastArgs a, uint16_t* __restrict__ depth,
                                                            int depth_step, float* __restrict__ normals,
                                                            int normals_step) {
    int x, y;
    tile_pixel(x, y);
    if (x >= a.cols || y >= a.rows) return;
    uint16_t* drow = (uint16_t*)((char*)depth + (size_t)y * depth_step);
    float4* nrow   = (float4*)((char*)normals + (size_t)y * normals_step);
    f3 v, n;
    if (cast_ray<false, IDX32>(a, x, y, v, n)) {
        nrow[x]  = make_float4(n.x, n.y, n.z, 0.f);  // :250
        float mm = v.z * 1000.f;                     // :251, saturating truncation
        mm       = mm < 0.f ? 0.f : (mm > 65535.f ? 65535.f : mm);
        drow[x]  = (uint16_t)(int)mm;
    } else {
        const float q = qnan();
        drow[x]       = 0;  // :204-205
        nrow[x]       = make_float4(q, q, q, q);
    }
}

// the same rays with the work counted instead of the images written (measurement only)
__global__ __launch_bounds__(256) void raycast_tally_kernel(const RaycastArgs a, unsigned long long* __restrict__ counts,
                                                            uint32_t* __restrict__ touched) {
    int x, y;
    tile_pixel(x, y);
    RayTally t{counts, touched, 0u, 0u, 0u, 0u};
    if (x < a.cols && y < a.rows) {
        f3 v, n;
        cast_ray<true>(a, x, y, v, n, &t);
    }
    unsigned int part[4] = {t.entered, t.march, t.hits, t.tri};
    for (int c = 0; c < 4; ++c) {
        unsigned int s = part[c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&counts[c], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------
// host-side launchers (called by the C ABI, capi.cpp)

static inline hipError_t launch_status() { return hipGetLastError(); }

hipError_t launch_compute_dists(const uint16_t* depth, int depth_step, uint16_t* dists, int dists_step, int cols,
                                int rows, float fx, float fy, float cx, float cy, hipStream_t s) {
    dim3 block(64, 4), grid((cols + 63) / 64, (rows + 3) / 4);
    // host wrapper passes finv = 1/f (imgproc.cu:252)
    compute_dists_kernel<<<grid, block, 0, s>>>(depth, depth_step, dists, dists_step, cols, rows, 1.f / fx, 1.f / fy,
                                                cx, cy);
    return launch_status();
}

static int pick_zchunk(int X, int Y, int Z, int vx, bool fused_clear);

hipError_t launch_tsdf_clear(uint32_t* vol, int X, int Y, int Z, hipStream_t s) {
    const size_t n = (size_t)X * Y * Z;
    const bool linear = dev_env("DFA_TSDF_CLEAR_LINEAR") != nullptr;  // A/B (development builds): the grid-stride 16-byte stores
    if (!linear && X % 64 == 0 && Z >= 32 && (((uintptr_t)vol & 255) == 0)) {
        const int zchunk = pick_zchunk(X, Y, Z, 1, true);
        dim3 block(64, 4), grid(X / 64, (Y + 3) / 4, (Z + zchunk - 1) / zchunk);
        clear_columns_kernel<<<grid, block, 0, s>>>(vol, X, Y, Z, zchunk);
        return launch_status();
    }
    // head/tail so the 16-byte stores are aligned whatever pointer the caller passes
    size_t head = ((16 - ((uintptr_t)vol & 15)) & 15) / 4;
    if (head > n) head = n;
    if (head) {
        hipError_t e = hipMemsetAsync(vol, 0, head * 4, s);
        if (e != hipSuccess) return e;
    }
    uint32_t* body  = vol + head;
    const size_t nb = n - head;
    const size_t n4 = nb / 4;
    const int ntail = (int)(nb - n4 * 4);
    size_t blocks   = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // 16 blocks per CU, grid-stride the rest
    if (blocks == 0) blocks = 1;
    clear_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((uint4*)body, n4, body + n4 * 4, ntail);
    return launch_status();
}

// Scratch for the tile table of one sweep (19 KiB at VGA).  The C ABI's caller owns every buffer it passes and the
// kernels allocate nothing it can see; this table is internal, so it is cached per stream: work on one stream is
// ordered, so a sweep never overwrites the table of a sweep still running, whatever the host threads do.  Grows with
// hipMalloc (hipFree of the old block waits for the device).
static hipError_t tile_scratch(hipStream_t s, size_t bytes, uint32_t** out) {
    struct Block {
        uint32_t* p = nullptr;
        size_t cap  = 0;
    };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Block> cache;  // (device, stream): the null stream exists on every device
    int dev = 0;
    hipError_t de = hipGetDevice(&dev);
    if (de != hipSuccess) return de;
    std::lock_guard<std::mutex> lock(mu);
    Block& b = cache[std::make_pair(dev, s)];
    if (b.cap < bytes) {
        if (b.p) (void)hipFree(b.p);
        b.p = nullptr, b.cap = 0;
        hipError_t e = hipMalloc((void**)&b.p, bytes);
        if (e != hipSuccess) return e;
        b.cap = bytes;
    }
    *out = b.p;
    return hipSuccess;
}

// z-chunk heuristic: enough workgroups to fill 256 CUs several times over, while keeping the
// replayed-additions prologue (z0 adds per chunk) a small fraction of a chunk's work.
static int pick_zchunk(int X, int Y, int Z, int vx, bool fused_clear) {
    const long columns_wg = (long)((X + 64 * vx - 1) / (64 * vx)) * ((Y + 3) / 4);
    int zchunk            = Z;
    // Chunks no shorter than 32 slices.  Measured with one voxel per lane (tools/tsdf_kernels.py): the read+write sweep
    // wants >= 16 workgroups per CU (512^3: 0.265 ms with 4 chunks, 0.386 unsplit); the fused sweep has no loads to hide
    // and is as fast with 4 per CU (512^3 unsplit 0.24 ms, 4 chunks 0.23) — and then half of the chip's wave slots stay
    // free for the kernels of other streams: the per-frame sweep runs beside the solve, whose short full-chip kernels
    // otherwise wait for resident sweep waves to retire (C2 frame 0.969 -> 0.957 ms, C3 4.27 -> 4.10).
    const long want = fused_clear ? 1024 : 4096;
    while (columns_wg * ((Z + zchunk - 1) / zchunk) < want && zchunk > 32) zchunk /= 2;
    zchunk = (zchunk + 3) & ~3;
    if (const char* e = dev_env("DFA_TSDF_ZCHUNK")) {  // (development builds: the tests of the z-chunk independence)
        int v = atoi(e);
        if (v > 0) zchunk = v;
    }
    return zchunk;
}

hipError_t launch_tsdf_integrate(bool fused_clear, const uint16_t* dists, int dists_step, int cols, int rows,
                                 uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                 int max_weight, const float vol2cam[12], float fx, float fy, float cx, float cy,
                                 uint8_t* occ, bool occ_known, hipStream_t s) {
    IntegrateArgs a;
    a.occ_known = occ && occ_known && fused_clear ? 1 : 0;
    a.chunk_rule = 0;
    a.dists = dists, a.dists_step = dists_step, a.cols = cols, a.rows = rows;
    a.vol = vol, a.X = X, a.Y = Y, a.Z = Z;
    const OccDims od = occ_dims(X, Y, Z);
    a.occ = nullptr, a.ox = od.ox, a.oy = od.oy;
    // a sweep that does not keep the map (the per-voxel sweeps below, other wave shapes / run lengths of development builds)
    // marks everything: the map stays a superset of the voxels with a weight
    auto occ_all = [&]() -> hipError_t { return occ ? hipMemsetAsync(occ, 3, od.bytes(), s) : hipSuccess; };
    a.vsx = voxel_size[0], a.vsy = voxel_size[1], a.vsz = voxel_size[2];
    a.trunc      = trunc_dist;
    a.trunc_inv  = 1.f / trunc_dist;  // tsdf_volume.cu:106
    a.max_weight = max_weight;
    for (int i = 0; i < 9; ++i) a.vol2cam.m[i] = vol2cam[i];
    for (int i = 0; i < 3; ++i) a.vol2cam.t[i] = vol2cam[9 + i];
    a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy;
#ifdef DFA_DEV_ABLATE  // development builds only (-DDFA_DEV_ABLATE): values 1 / 2 write WRONG volumes by design
    a.ablate = dev_env_int("DFA_TSDF_ABLATE", 0);
#else
    a.ablate = 0;
#endif
    a.nt     = dev_env_int("DFA_TSDF_NT", 0);
    // The run-classified sweep.  Development builds (-DDFA_DEV_AB): DFA_TSDF_LEGACY=1 runs the per-voxel sweep (every voxel
    // through the projection; the round-1 kernel) for A/B timings, DFA_TSDF_WAVE=16 gives a wave a 16 x 4 patch of columns.
    const bool legacy = dev_env("DFA_TSDF_LEGACY") != nullptr;
    if (!legacy) {
        const int tcols = (cols + 7) >> TSDF_TILE_SHIFT, trows = (rows + 7) >> TSDF_TILE_SHIFT;
        uint32_t* tiles = nullptr;
        hipError_t e    = tile_scratch(s, (size_t)tcols * trows * sizeof(uint32_t), &tiles);
        if (e != hipSuccess) return e;
        dists_tiles_kernel<<<(tcols * trows + 3) / 4, 256, 0, s>>>(dists, dists_step, cols, rows, tiles, tcols, tcols * trows);
        // bound of |component| over every voxel position in camera space: the 8 corners of the volume
        float extent = 0.f;
        for (int c = 0; c < 8; ++c) {
            const float p[3] = {(c & 1) ? a.vsx * X : 0.f, (c & 2) ? a.vsy * Y : 0.f, (c & 4) ? a.vsz * Z : 0.f};
            for (int r = 0; r < 3; ++r)
                extent = fmaxf(extent, fabsf(vol2cam[3 * r] * p[0] + vol2cam[3 * r + 1] * p[1] + vol2cam[3 * r + 2] * p[2] + vol2cam[9 + r]));
        }
        const float zstep[3] = {vol2cam[2] * a.vsz, vol2cam[5] * a.vsz, vol2cam[8] * a.vsz};
        if (extent == extent && extent < 1e30f) {  // finite poses only; anything else takes the per-voxel sweep
            const int run_u = dev_env_int("DFA_TSDF_RUN", RUN_U);
            const RunConsts rc = make_run_consts(tiles, cols, rows, fx, fy, cx, cy, trunc_dist, zstep, run_u == 8 ? 8 : 4, extent);
            const uint32_t front_const = 0x3c00u | ((uint32_t)(max_weight < 1 ? max_weight : 1) << 16);  // (1.0h, min(1, max_weight))
            // >= 4 096 workgroups for either sweep: the tile look-ups of a run are dependent loads that only occupancy
            // hides (512^3 fused: 0.147 ms unsplit = 1 024 workgroups, 0.117 ms with z-chunks of 128 slices)
            a.zchunk = pick_zchunk(X, Y, Z, 1, false);
            if (occ) a.zchunk = (a.zchunk + 7) & ~7;  // chunks of whole runs: a byte of the map has one writer
            dim3 block(64, 4), grid((X + 63) / 64, (Y + 3) / 4, (Z + a.zchunk - 1) / a.zchunk);
#ifdef DFA_DEV_AB
            if (occ && (dev_env_int("DFA_TSDF_WAVE", 32) != 32 || run_u != 8)) {
                const hipError_t oe = occ_all();
                if (oe != hipSuccess) return oe;
            } else
#endif
                a.occ = occ;
            // the chunk-level rule: margins of Z running additions (DFA_TSDF_NO_CHUNK_RULE=1 in development builds: A/B)
            const RunConsts rcc = make_run_consts(tiles, cols, rows, fx, fy, cx, cy, trunc_dist, zstep, Z, extent);
            a.chunk_rule = dev_env("DFA_TSDF_NO_CHUNK_RULE") ? 0 : 1;
#define DFA_RUNS(F, W, UU) integrate_runs_kernel<F, W, UU><<<grid, block, 0, s>>>(a, rc, rcc, front_const)
#ifdef DFA_DEV_AB
            const int wave_x = dev_env_int("DFA_TSDF_WAVE", 32);
#define DFA_RUNS_W(F, UU) (wave_x == 16 ? DFA_RUNS(F, 16, UU) : wave_x == 32 ? DFA_RUNS(F, 32, UU) : DFA_RUNS(F, 64, UU))
            if (run_u != RUN_U) {
                if (fused_clear) DFA_RUNS_W(true, RUN_U == 8 ? 4 : 8);
                else DFA_RUNS_W(false, RUN_U == 8 ? 4 : 8);
            } else {
                if (fused_clear) DFA_RUNS_W(true, RUN_U);
                else DFA_RUNS_W(false, RUN_U);
            }
#undef DFA_RUNS_W
#else
            static_assert(RUN_U == 8 || RUN_U == 4, "run length of the classified sweep");
            if (fused_clear) DFA_RUNS(true, 32, RUN_U);  // a wave = 32 x 2 voxel columns, runs of RUN_U voxels
            else DFA_RUNS(false, 32, RUN_U);
#endif
#undef DFA_RUNS
            return launch_status();
        }
    }
    // One voxel per lane.  Four consecutive voxels per lane (16-byte accesses; DFA_TSDF_VX4=1, the first design) lose
    // everywhere: a lane then walks its four voxels one after the other, each with its own early exits, and a wave
    // waits for its slowest lane four times per slice (fused sweep 0.289 -> 0.229 ms at 512^3, 1.49 -> 1.38 ms at
    // 1024^3, 0.061 -> 0.044 ms at 256^3); a wave's 256-byte stores are wide enough for HBM.
#ifdef DFA_DEV_AB
    const bool vec4 = dev_env("DFA_TSDF_VX4") && (X % 4 == 0) && (((uintptr_t)vol & 15) == 0);
#else
    constexpr bool vec4 = false;
#endif
    const int vx    = vec4 ? 4 : 1;
    {
        const hipError_t oe = occ_all();
        if (oe != hipSuccess) return oe;
    }
    a.zchunk        = pick_zchunk(X, Y, Z, vx, fused_clear);
    dim3 block(64, 4), grid((X + 64 * vx - 1) / (64 * vx), (Y + 3) / 4, (Z + a.zchunk - 1) / a.zchunk);
#ifdef DFA_DEV_AB
    if (vec4) {
        if (fused_clear) integrate_kernel<true, 4><<<grid, block, 0, s>>>(a);
        else integrate_kernel<false, 4><<<grid, block, 0, s>>>(a);
        return launch_status();
    }
#endif
    if (fused_clear) integrate_kernel<true, 1><<<grid, block, 0, s>>>(a);
    else integrate_kernel<false, 1><<<grid, block, 0, s>>>(a);
    return launch_status();
}

static RaycastArgs make_raycast_args(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3],
                                     float trunc_dist, const float cam2vol[12], const float Rinv[9], float fx,
                                     float fy, float cx, float cy, float step_factor, float delta_factor, int cols,
                                     int rows) {
    RaycastArgs a;
    a.vol = vol, a.X = X, a.Y = Y, a.Z = Z;
    a.vsx = voxel_size[0], a.vsy = voxel_size[1], a.vsz = voxel_size[2];
    // tsdf_volume.cu:359-362 (host, plain float arithmetic)
    a.sx = voxel_size[0] * (float)X, a.sy = voxel_size[1] * (float)Y, a.sz = voxel_size[2] * (float)Z;
    a.time_step = trunc_dist * step_factor;
    a.gdx = voxel_size[0] * delta_factor, a.gdy = voxel_size[1] * delta_factor, a.gdz = voxel_size[2] * delta_factor;
    a.vix = 1.f / voxel_size[0], a.viy = 1.f / voxel_size[1], a.viz = 1.f / voxel_size[2];
    for (int i = 0; i < 9; ++i) a.cam2vol.m[i] = cam2vol[i], a.Rinv.m[i] = Rinv[i];
    for (int i = 0; i < 3; ++i) a.cam2vol.t[i] = cam2vol[9 + i];
    a.finvx = 1.f / fx, a.finvy = 1.f / fy, a.cx = cx, a.cy = cy;
    a.cols = cols, a.rows = rows;
    return a;
}

// Normals of surface points (marching-cubes vertices) from the TSDF gradient: the raycaster's own compute_normal
// (tsdf_volume.cu:320-336 — central differences of the trilinear interpolant, `delta_factor` voxels apart) applied
// to points given in the volume's metric frame.  The reference leaves the extracted mesh without normals
// (dyn_fusion.cpp:80-88 "temporary workaround until normals are computed via mc"): this is SURVEY 8f rank 2.
__global__ __launch_bounds__(256) void vertex_normals_kernel(const RaycastArgs a, const float4* __restrict__ points, int n,
                                                             float4* __restrict__ normals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = points[i];
    const f3 nn    = compute_normal(a, mk3(p.x, p.y, p.z));  // NaN where a sample leaves the interpolation range
    normals[i]     = make_float4(nn.x, nn.y, nn.z, 0.f);
}

hipError_t launch_vertex_normals(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float delta_factor,
                                 const float* points, int n, float* normals, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const float id12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, id9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    RaycastArgs a = make_raycast_args(vol, X, Y, Z, voxel_size, 1.f, id12, id9, 1.f, 1.f, 0.f, 0.f, 1.f, delta_factor, 0, 0);
    vertex_normals_kernel<<<(n + 255) / 256, 256, 0, s>>>(a, (const float4*)points, n, (float4*)normals);
    return launch_status();
}

hipError_t launch_raycast_points(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3],
                                 float trunc_dist, const float cam2vol[12], const float Rinv[9], float fx, float fy,
                                 float cx, float cy, float step_factor, float delta_factor, float* points,
                                 int points_step, float* normals, int normals_step, int cols, int rows,
                                 hipStream_t s) {
    RaycastArgs a = make_raycast_args(vol, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy,
                                      step_factor, delta_factor, cols, rows);
    dim3 block(256), grid((cols + 15) / 16, (rows + 15) / 16);
    // (development builds: DFA_RAY_IDX64=1 runs the 64-bit voxel index of volumes beyond 2^32 voxels on any volume — the tests)
    if ((uint64_t)X * Y * Z <= (1ull << 32) && !dev_env("DFA_RAY_IDX64")) raycast_points_kernel<true><<<grid, block, 0, s>>>(a, points, points_step, normals, normals_step);
    else raycast_points_kernel<false><<<grid, block, 0, s>>>(a, points, points_step, normals, normals_step);
    return launch_status();
}

hipError_t launch_raycast_tally(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                                float step_factor, float delta_factor, int cols, int rows, unsigned long long* counts,
                                uint32_t* touched, hipStream_t s) {
    RaycastArgs a = make_raycast_args(vol, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy,
                                      step_factor, delta_factor, cols, rows);
    dim3 block(256), grid((cols + 15) / 16, (rows + 15) / 16);
    raycast_tally_kernel<<<grid, block, 0, s>>>(a, counts, touched);
    return launch_status();
}

hipError_t launch_raycast_depth(const uint32_t* vol, int X, int Y, int Z, const float voxel_size[3], float trunc_dist,
                                const float cam2vol[12], const float Rinv[9], float fx, float fy, float cx, float cy,
                                float step_factor, float delta_factor, uint16_t* depth, int depth_step,
                                float* normals, int normals_step, int cols, int rows, hipStream_t s) {
    RaycastArgs a = make_raycast_args(vol, X, Y, Z, voxel_size, trunc_dist, cam2vol, Rinv, fx, fy, cx, cy,
                                      step_factor, delta_factor, cols, rows);
    dim3 block(256), grid((cols + 15) / 16, (rows + 15) / 16);
    if ((uint64_t)X * Y * Z <= (1ull << 32) && !dev_env("DFA_RAY_IDX64")) raycast_depth_kernel<true><<<grid, block, 0, s>>>(a, depth, depth_step, normals, normals_step);
    else raycast_depth_kernel<false><<<grid, block, 0, s>>>(a, depth, depth_step, normals, normals_step);
    return launch_status();
}

}  // namespace dfa
