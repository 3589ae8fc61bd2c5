// mc.hip — marching cubes over the packed TSDF volume for gfx950.
//
// Reference semantics: src/kfusion/cuda/marching_cubes.cu (computeCubeIndex :35-73, OccupiedVoxels
// :75-141, computeOffsetsAndTotalVertices :165-181, TrianglesGenerator :183-257) and the host
// driver src/kfusion/marching_cubes.cpp:20-61.  The reference hard-codes a 128^3 volume
// (include/kfusion/internal.hpp:74, marching_cubes.cu:147,283-285); here the dimensions are
// arguments.
//
// MI355X design.  The reference compacts the occupied voxels with one global atomicAdd per warp
// and z slice (the voxel order of its output therefore changes from run to run), scans with
// thrust and synchronises with the host three times.  Here the output order is DEFINED — ascending
// linear voxel index — and nothing leaves the device:
//   1. count sweep: HBM-bound streaming read of the volume.  A lane owns 4 consecutive x voxels
//      (one 16-byte load per row), a wave one 256-voxel row segment, a thread marches along z and
//      keeps the two rows of slice z in registers while it loads slice z + 1; the x + 1 neighbour
//      comes from the next lane by a wave shuffle.  Output: vertices per segment.
//   2. exclusive scan of the segment counts (chunked, two small kernels).
//   3. emit: persistent waves look for segments with vertices (3 % of them for a surface), stage
//      such a segment's four rows in LDS, and then every lane produces one VERTEX at a time (its
//      cube found by a binary search over the cube offsets): 64 consecutive float4 points
//      {x, y, z, 1} (store_point :255-257) per wave store.
// The case tables are arguments (device pointers), as in kfusion::device::bindTextures (:14-19).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "device_math.hpp"
#include "dev_switch.hpp"
#include "kernels.hpp"

namespace dfa {

namespace {

struct McArgs {
    const uint32_t* vol;
    int X, Y, Z;
    int nseg;    // row segments per row: ceil(X / (64 * VX))
    int zchunk;  // count sweep: slices per workgroup
    float csx, csy, csz;
    const int32_t* tri;
    const int32_t* nverts;
    const uint8_t* occ;  // occupancy map of the volume (kernels.hpp: OccDims: a byte per 32 x 2 x 8 voxels) or null
    int ox, oy, oz;
};

// the VX voxels a lane owns in one row plus the voxel after them; zeros (weight 0 -> "no cube",
// marching_cubes.cu:38-60) wherever the row or the voxel does not exist
template <int VX>
struct Row {
    uint32_t v[VX + 1];
};

template <int VX>
__device__ __forceinline__ Row<VX> load_row(const McArgs& a, int x0, int y, int z) {
    Row<VX> r;
#pragma unroll
    for (int i = 0; i <= VX; ++i) r.v[i] = 0u;
    const bool row_ok = y < a.Y && z < a.Z;
    const uint32_t* p = a.vol + (size_t)a.X * ((size_t)y + (size_t)a.Y * (size_t)z);
    if (row_ok && x0 < a.X) {
        if (VX == 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(p + x0);
            r.v[0] = q.x, r.v[1] = q.y, r.v[2] = q.z, r.v[3] = q.w;
        } else {
            r.v[0] = p[x0];
        }
    }
    // the neighbour's first voxel; the last lane of the wave reads it from memory
    const uint32_t next = __shfl_down(r.v[0], 1, 64);
    if ((threadIdx.x & 63) == 63) {
        if (row_ok && x0 + VX < a.X) r.v[VX] = p[x0 + VX];
    } else {
        r.v[VX] = next;
    }
    return r;
}

// NR consecutive rows y .. y + NR - 1 of slice z.  The voxel after the wave's last one is fetched for
// all NR rows by ONE load instruction (lane r reads row r's) and handed to lane 63 by a readlane —
// the sweep is bound by vector-memory instruction issue, not by bytes.
template <int VX, int NR>
__device__ __forceinline__ void load_rows(const McArgs& a, int x0, int y, int z, Row<VX> (&out)[NR]) {
    const int lane = threadIdx.x & 63;
    const bool z_ok = z < a.Z;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        Row<VX>& o = out[r];
#pragma unroll
        for (int i = 0; i <= VX; ++i) o.v[i] = 0u;
        const uint32_t* p = a.vol + (size_t)a.X * ((size_t)(y + r) + (size_t)a.Y * (size_t)z);
        if (z_ok && y + r < a.Y && x0 < a.X) {
            if (VX == 4) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + x0);
                o.v[0] = q.x, o.v[1] = q.y, o.v[2] = q.z, o.v[3] = q.w;
            } else {
                o.v[0] = p[x0];
            }
        }
    }
    const int xend = (x0 - lane * VX) + 64 * VX;  // first voxel of the next segment
    uint32_t extra = 0u;
    if (lane < NR && z_ok && y + lane < a.Y && xend < a.X)
        extra = a.vol[(size_t)xend + (size_t)a.X * ((size_t)(y + lane) + (size_t)a.Y * (size_t)z)];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const uint32_t next = __shfl_down(out[r].v[0], 1, 64);
        const uint32_t last = __shfl(extra, r, 64);
        out[r].v[VX]        = lane == 63 ? last : next;
    }
}

// bit c of `neg`: f < 0 (isoValue = 0, internal.hpp:72); returns false if any weight is 0
__device__ __forceinline__ int cube_case(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4, uint32_t c5,
                                         uint32_t c6, uint32_t c7) {
    const uint32_t c[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
    bool valid          = true;
    int ci              = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        valid = valid && (c[k] >> 16) != 0u;
        ci |= (int)(half_bits_to_float(c[k] & 0xffffu) < 0.f) << k;  // :63-71
    }
    return valid ? ci : 0;
}

// cube i (0..VX-1) of the lane from the four rows (y,z) (y+1,z) (y,z+1) (y+1,z+1); corner order :37-60
template <int VX>
__device__ __forceinline__ int lane_cube(const Row<VX>& a0, const Row<VX>& a1, const Row<VX>& b0, const Row<VX>& b1,
                                         int i) {
    return cube_case(a0.v[i], a0.v[i + 1], a1.v[i + 1], a1.v[i], b0.v[i], b0.v[i + 1], b1.v[i + 1], b1.v[i]);
}

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------- 1. count
// wave total of per-lane triangle counts in [0, 31]: five ballots + scalar popcounts, no
// cross-lane dependency chain in the z loop
__device__ __forceinline__ int wave_sum_5bit(int n) {
    int s = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) s += __popcll(__ballot((n >> b) & 1)) << b;
    return s;
}

constexpr int MC_ROWS = 2;  // rows of cubes per wave: 3 row loads per slice serve 2 rows (y+1 shared)

template <int VX>
__global__ __launch_bounds__(256) void mc_count_kernel(const McArgs a, int32_t* __restrict__ seg_count) {
    __shared__ uint8_t ntri_lds[256];  // triangles per case
    __shared__ int wave_work[4];
    const int seg = blockIdx.x, x0 = (seg * 64 + threadIdx.x) * VX;
    const int y = (blockIdx.y * 4 + threadIdx.y) * MC_ROWS;
    const int z0 = blockIdx.z * a.zchunk, z1 = min(z0 + a.zchunk, a.Z - 1);
    // With an occupancy map: a cube with triangles has a non-zero weight at all eight corners (:38-60) and a negative
    // distance at one of them, so a slice pair (z, z + 1) has such cubes for this wave only if the map shows weights in BOTH
    // slices' layers over the wave's footprint — its 64 VX voxels and the one after them in x (2 VX + 1 boxes), rows
    // y .. y + MC_ROWS (two rows of boxes) — and a possibly negative distance in one of them.  Everything else is never
    // loaded: all but the band around the surface.  One byte per lane and two ballots per layer of 8 slices.
    auto layer_marks = [&](int z8) -> unsigned {  // bit 0: weights somewhere in the footprint; bit 1: negative distances possible
        if (!a.occ) return 3u;
        constexpr int NBX = 2 * VX + 1;
        const int lane = threadIdx.x, by = y / 2 + lane / NBX, bx = seg * 2 * VX + lane % NBX;
        unsigned v = 0u;
        if (lane < 2 * NBX && by < a.oy && bx < a.ox && z8 < a.oz) v = a.occ[((size_t)z8 * a.oy + by) * a.ox + bx];
        return (__ballot(v & 1u) != 0ull ? 1u : 0u) | (__ballot(v & 2u) != 0ull ? 2u : 0u);
    };
    // what the wave's chunk can hold at all (the layers of its first and last slice: chunks with a map are one layer long),
    // before anything else is loaded: most workgroups of a sparse volume end here
    bool work = y < a.Y && z0 < z1;
    if (work && a.occ) {
        const unsigned m0 = layer_marks(z0 / 8), m1 = z1 / 8 != z0 / 8 ? layer_marks(z1 / 8) : m0;
        // (the steps inside the first layer need weights and a negative distance there; the step into the last layer both layers)
        work = a.zchunk > 8 || (m0 & 3u) == 3u || ((m0 & m1 & 1u) && ((m0 | m1) & 2u));
    }
    if (threadIdx.x == 0) wave_work[threadIdx.y] = work;
    __syncthreads();
    if (!(wave_work[0] | wave_work[1] | wave_work[2] | wave_work[3])) return;  // (the whole workgroup)
    {
        const int t  = threadIdx.y * 64 + threadIdx.x;
        const int nv = a.nverts[t];
        ntri_lds[t]  = (t == 0 || t == 255) ? 0 : (uint8_t)(min(max(nv, 0), 15) / 3);  // :99
    }
    __syncthreads();
    if (!work) return;  // whole wave
    int cur8 = z0 / 8;
    unsigned cur = layer_marks(cur8);
    bool lo_valid = false;
    Row<VX> lo[MC_ROWS + 1];
    for (int z = z0; z < z1; ++z) {
        const int nxt8 = (z + 1) / 8;
        const unsigned nxt = nxt8 == cur8 ? cur : layer_marks(nxt8);
        const bool need = (cur & nxt & 1u) && ((cur | nxt) & 2u);  // (wave-uniform)
        cur = nxt, cur8 = nxt8;
        if (!need) {
            lo_valid = false;
            continue;
        }
        if (!lo_valid) load_rows<VX, MC_ROWS + 1>(a, x0, y, z, lo);
        lo_valid = true;
        Row<VX> hi[MC_ROWS + 1];
        load_rows<VX, MC_ROWS + 1>(a, x0, y, z + 1, hi);
#pragma unroll
        for (int r = 0; r < MC_ROWS; ++r) {
            int n = 0;
#pragma unroll
            for (int i = 0; i < VX; ++i) n += ntri_lds[lane_cube<VX>(lo[r], lo[r + 1], hi[r], hi[r + 1], i)];
            n = wave_sum_5bit(n);  // VX * 5 <= 20 triangles per lane
            if (threadIdx.x == 0 && n && y + r < a.Y) seg_count[((size_t)z * a.Y + y + r) * a.nseg + seg] = 3 * n;
        }
#pragma unroll
        for (int r = 0; r <= MC_ROWS; ++r) lo[r] = hi[r];
    }
}

// ------------------------------------------------------------------------------- 2. scan
constexpr int SCAN_CHUNK = 8192;  // entries per workgroup

__global__ __launch_bounds__(256) void scan_sum_kernel(const int32_t* __restrict__ in, long n,
                                                       int32_t* __restrict__ chunk_sums) {
    __shared__ int sh[4];
    const long base = (long)blockIdx.x * SCAN_CHUNK;
    int sum         = 0;
    for (long i = base + threadIdx.x; i < min(base + SCAN_CHUNK, n); i += 256) sum += in[i];
    sum = wave_sum_int(sum);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// out[i] = sum of in[0..i), out[n] = total; in and out may alias
__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* in, long n, int32_t* out,
                                                         const int32_t* __restrict__ chunk_sums,
                                                         int32_t* __restrict__ total) {
    __shared__ int sh[4], sh2[4];
    const long base = (long)blockIdx.x * SCAN_CHUNK;
    int before      = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) before += chunk_sums[c];
    before = wave_sum_int(before);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = before;
    constexpr int PER = SCAN_CHUNK / 256;
    const long first  = base + (long)threadIdx.x * PER;
    // (a thread's 32 entries are 128 contiguous bytes: 16-byte accesses where they all exist and the arrays allow it)
    const bool whole = first + PER <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    int loc[PER], sum = 0;
    if (whole) {
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) {
            const int4 c4 = reinterpret_cast<const int4*>(in + first)[q];
            loc[4 * q] = c4.x, loc[4 * q + 1] = c4.y, loc[4 * q + 2] = c4.z, loc[4 * q + 3] = c4.w;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) sum += loc[j];
    } else {
#pragma unroll
        for (int j = 0; j < PER; ++j) loc[j] = first + j < n ? in[first + j] : 0, sum += loc[j];
    }
    int incl       = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sh2[wave] = incl;
    __syncthreads();
    int off = sh[0] + sh[1] + sh[2] + sh[3] + incl - sum;
    for (int w = 0; w < wave; ++w) off += sh2[w];
    if (whole) {
        int st[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) st[j] = off, off += loc[j];
#pragma unroll
        for (int q = 0; q < PER / 4; ++q) reinterpret_cast<int4*>(out + first)[q] = make_int4(st[4 * q], st[4 * q + 1], st[4 * q + 2], st[4 * q + 3]);
        if (first + PER == n) {
            out[n] = off;
            if (total) *total = off;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (first + j < n) {
            out[first + j] = off;
            off += loc[j];
            if (first + j == n - 1) {
                out[n] = off;
                if (total) *total = off;
            }
        }
    }
}

// ------------------------------------------------------------------------------- 3. emit
// vertex_interp (:193-199) for one axis-aligned edge; only the coordinate along the edge moves but
// the reference interpolates all three, and p0 + t * 0 may round differently from p0 only for
// non-finite t — kept literal.
__device__ __forceinline__ f3 vertex_interp(f3 p0, f3 p1, float f0, float f1) {
    const float t = (0.f - f0) / (f1 - f0 + 1e-15f);
    return mk3(p0.x + t * (p1.x - p0.x), p0.y + t * (p1.y - p0.y), p0.z + t * (p1.z - p0.z));
}

// per-wave staging of one segment in LDS: the half-float tsdf of the four rows (64 * VX + 1 voxels
// each) and, per cube, its case and the offset of its first vertex inside the segment
template <int VX>
struct SegStage {
    static constexpr int W = 64 * VX + 1;
    uint16_t f[4][W + 3];
    uint16_t start[64 * VX];
    uint8_t ci[64 * VX];
};

// one segment, the whole wave cooperating (wave-uniform arguments).  Lanes first own cubes (case
// and vertex count, wave prefix sum), then own VERTICES: vertex t of the segment finds its cube by
// a binary search over the cube offsets, its edge in the case table, and interpolates — 64
// consecutive float4 stores per wave instead of lanes looping over their own cubes' vertices.
template <int VX>
__device__ __forceinline__ void emit_segment(const McArgs& a, const uint8_t* __restrict__ tri_lds,
                                             const uint8_t* __restrict__ nv_lds, SegStage<VX>& st, long s, int begin,
                                             int count, float4* __restrict__ out, int max_vertices) {
    const int seg = (int)(s % a.nseg);
    const long yz = s / a.nseg;
    const int y = (int)(yz % a.Y), z = (int)(yz / a.Y);
    const int lane = threadIdx.x & 63;
    const int x0   = (seg * 64 + lane) * VX;
    Row<VX> r[4];  // (y,z) (y+1,z) (y,z+1) (y+1,z+1)
    r[0] = load_row<VX>(a, x0, y, z), r[1] = load_row<VX>(a, x0, y + 1, z);
    r[2] = load_row<VX>(a, x0, y, z + 1), r[3] = load_row<VX>(a, x0, y + 1, z + 1);
    int ci[VX], mine = 0;
#pragma unroll
    for (int i = 0; i < VX; ++i) {
        ci[i] = lane_cube<VX>(r[0], r[1], r[2], r[3], i);
        mine += nv_lds[ci[i]];
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    int off = incl - mine;
#pragma unroll
    for (int i = 0; i < VX; ++i) {
        st.ci[lane * VX + i]    = (uint8_t)ci[i];
        st.start[lane * VX + i] = (uint16_t)off;
        off += nv_lds[ci[i]];
#pragma unroll
        for (int q = 0; q < 4; ++q) st.f[q][lane * VX + i] = (uint16_t)(r[q].v[i] & 0xffffu);
    }
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st.f[q][64 * VX] = (uint16_t)(r[q].v[VX] & 0xffffu);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = lane; t < count; t += 64) {
        // last cube whose first vertex is at or before t (cubes without vertices share the offset of
        // the next cube that has some, so the last one is the owner)
        int lo = 0, hi = 64 * VX;  // answer in [lo, hi)
#pragma unroll
        for (int step = 0; step < (VX == 4 ? 8 : 6); ++step) {
            const int mid = (lo + hi) >> 1;
            if ((int)st.start[mid] <= t) lo = mid;
            else hi = mid;
        }
        const int c  = lo;
        const int cs = st.ci[c];
        const int e  = tri_lds[cs * 16 + (t - (int)st.start[c])];
        // edge e joins corners (e0, e1): :233-244; corner k sits at (dx, dy, dz) = bits of 0x66, 0xCC, k >> 2
        const int e0 = e < 8 ? e : e - 8;
        const int e1 = e < 4 ? ((e + 1) & 3) : e < 8 ? 4 + ((e + 1) & 3) : e - 4;
        const int dx0 = (0x66 >> e0) & 1, dy0 = (0xCC >> e0) & 1, dz0 = e0 >> 2;
        const int dx1 = (0x66 >> e1) & 1, dy1 = (0xCC >> e1) & 1, dz1 = e1 >> 2;
        const int x   = seg * 64 * VX + c;
        const float f0 = half_bits_to_float(st.f[dy0 + 2 * dz0][c + dx0]);
        const float f1 = half_bits_to_float(st.f[dy1 + 2 * dz1][c + dx1]);
        // getNodeCoo :183-191: (i + 0.5) * cell_size
        const f3 p0 = mk3(((float)(x + dx0) + 0.5f) * a.csx, ((float)(y + dy0) + 0.5f) * a.csy, ((float)(z + dz0) + 0.5f) * a.csz);
        const f3 p1 = mk3(((float)(x + dx1) + 0.5f) * a.csx, ((float)(y + dy1) + 0.5f) * a.csy, ((float)(z + dz1) + 0.5f) * a.csz);
        const f3 p  = vertex_interp(p0, p1, f0, f1);
        if (begin + t < max_vertices) out[begin + t] = make_float4(p.x, p.y, p.z, 1.0f);
    }
    __builtin_amdgcn_wave_barrier();  // the next segment overwrites the staging area
}

// Persistent waves.  97 % of the segments of a surface volume are empty, and the others cluster in
// space, so wave w looks at the segments w, w + W, w + 2W, ... (W = waves of the grid): lane l reads
// the offsets of segment w + (64 i + l) W, a ballot finds the ones with vertices and the wave emits
// them one after the other — neighbouring segments land on different waves.
template <int VX>
__global__ __launch_bounds__(256) void mc_emit_kernel(const McArgs a, const int32_t* __restrict__ seg_off,
                                                      float4* __restrict__ out, int max_vertices, long nsegs_total) {
    // case tables in LDS: a vertex costs one dependent table read, and lanes emit up to 4 x 15 of them
    __shared__ uint8_t tri_lds[256 * 16], nv_lds[256];
    __shared__ SegStage<VX> stage[4];
    {
        const int t = threadIdx.y * 64 + threadIdx.x;
        for (int j = 0; j < 16; ++j) tri_lds[t * 16 + j] = (uint8_t)(a.tri[t * 16 + j] & 15);
        nv_lds[t] = (t == 0 || t == 255) ? 0 : (uint8_t)(3 * (min(max(a.nverts[t], 0), 15) / 3));
    }
    __syncthreads();
    const long nwaves = (long)gridDim.x * 4;
    const long w      = (long)blockIdx.x * 4 + threadIdx.y;
    const int lane    = threadIdx.x;
    for (long base = w; base < nsegs_total; base += 64 * nwaves) {
        const long s = base + (long)lane * nwaves;
        int begin = 0, end = 0;
        if (s < nsegs_total) begin = seg_off[s], end = seg_off[s + 1];
        unsigned long long todo = __ballot(end > begin && begin < max_vertices);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int b = __shfl(begin, l, 64);
            emit_segment<VX>(a, tri_lds, nv_lds, stage[threadIdx.y], base + (long)l * nwaves, b, __shfl(end, l, 64) - b, out,
                             max_vertices);
        }
    }
}

}  // namespace

long mc_segments(int X, int Y, int Z, bool vec4) {
    const int vx = vec4 ? 4 : 1;
    return (long)((X + 64 * vx - 1) / (64 * vx)) * Y * Z;
}
long mc_scan_chunks(long nsegs) { return (nsegs + SCAN_CHUNK - 1) / SCAN_CHUNK; }

hipError_t launch_marching_cubes(const uint32_t* vol, int X, int Y, int Z, const float cell_size[3],
                                 const int32_t* tri_table, const int32_t* num_verts_table, float* out_points,
                                 int max_vertices, int32_t* total_vertices, int32_t* seg_off, int32_t* chunk_sums,
                                 const uint8_t* occ, hipStream_t s) {
    const bool vec4 = (X % 4 == 0) && (((uintptr_t)vol & 15) == 0);
    const int vx    = vec4 ? 4 : 1;
    McArgs a;
    a.vol = vol, a.X = X, a.Y = Y, a.Z = Z;
    const OccDims od = occ_dims(X, Y, Z);
    a.occ = occ, a.ox = od.ox, a.oy = od.oy, a.oz = od.oz;
    a.nseg = (X + 64 * vx - 1) / (64 * vx);
    a.csx = cell_size[0], a.csy = cell_size[1], a.csz = cell_size[2];
    a.tri = tri_table, a.nverts = num_verts_table;
    const long nsegs = (long)a.nseg * Y * Z;
    // z chunks: >= 2048 workgroups when the volume allows, chunks of at least 16 slices
    const long columns = (long)a.nseg * ((Y + 4 * MC_ROWS - 1) / (4 * MC_ROWS));
    int zchunk         = Z;
    while (columns * ((Z + zchunk - 1) / zchunk) < 2048 && zchunk > 16) zchunk = (zchunk + 1) / 2;
    // With an occupancy map the sweep is no longer a stream: the few waves that have slices to load walk them as a chain of
    // dependent round trips (a slice per trip).  One layer of the map per workgroup: 4 x as many workgroups at 512^3, most
    // of which return after one look at the map, and chains of 8 trips instead of 32 (count sweep 109 -> see profiles/r05).
    if (occ) zchunk = 8;
    a.zchunk = zchunk;
    // only segments with vertices are written by the count sweep (an unconditional 4-byte store per
    // wave and slice costs 25 %: the sweep is bound by vector-memory instruction issue)
    hipError_t e = hipMemsetAsync(seg_off, 0, sizeof(int32_t) * (size_t)(nsegs + 1), s);
    if (e != hipSuccess) return e;
    dim3 block(64, 4), grid(a.nseg, (Y + 4 * MC_ROWS - 1) / (4 * MC_ROWS), (Z + zchunk - 1) / zchunk);
    if (vec4) mc_count_kernel<4><<<grid, block, 0, s>>>(a, seg_off);
    else mc_count_kernel<1><<<grid, block, 0, s>>>(a, seg_off);
    const int chunks = (int)mc_scan_chunks(nsegs);
    scan_sum_kernel<<<chunks, 256, 0, s>>>(seg_off, nsegs, chunk_sums);
    scan_apply_kernel<<<chunks, 256, 0, s>>>(seg_off, nsegs, seg_off, chunk_sums, total_vertices);
    if (out_points && max_vertices > 0) {
        long want = 8192;
        if (const char* e = dev_env("DFA_MC_EMIT_BLOCKS")) want = std::max(1L, atol(e));  // (development builds)
        const unsigned eblocks = (unsigned)std::min<long>((nsegs + 3) / 4, want);
        if (vec4) mc_emit_kernel<4><<<eblocks, block, 0, s>>>(a, seg_off, (float4*)out_points, max_vertices, nsegs);
        else mc_emit_kernel<1><<<eblocks, block, 0, s>>>(a, seg_off, (float4*)out_points, max_vertices, nsegs);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------- default case tables
// A marching-cubes case table derived from first principles (host code, runs once): for every
// sign configuration the crossed edges are joined face by face — two crossings of a face by one
// segment, four crossings (the ambiguous face) by the two segments that cut off the INSIDE
// corners, a rule that depends on the face's own corners only and therefore matches across
// neighbouring cubes — the segments close into loops, and every loop is fan-triangulated from its
// lowest edge id.  Corner / edge numbering and winding as the reference's kernels expect
// (corners marching_cubes.cu:37-60, edges :233-244, case 1 = {0, 8, 3}).  It is NOT the
// hand-made table the reference compiles in (src/kfusion/marching_cubes.cpp:86-343): pass that one
// to dfa_marching_cubes for output identical to the reference's; this one is for callers that
// have none.  At most 5 triangles per case, as the 16-wide rows require.
void mc_default_tables(int32_t tri_table[256 * 16], int32_t num_verts_table[256]) {
    static const int corner_of_edge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                              {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    // corner cycles of the six faces, counter-clockwise seen from outside the cube
    static const int face[6][4] = {{0, 3, 2, 1}, {4, 5, 6, 7}, {0, 1, 5, 4}, {3, 7, 6, 2}, {0, 4, 7, 3}, {1, 2, 6, 5}};
    int edge_id[8][8];
    for (auto& r : edge_id)
        for (int& v : r) v = -1;
    for (int e = 0; e < 12; ++e)
        edge_id[corner_of_edge[e][0]][corner_of_edge[e][1]] = edge_id[corner_of_edge[e][1]][corner_of_edge[e][0]] = e;
    for (int cs = 0; cs < 256; ++cs) {
        int next[12];
        for (int& v : next) v = -1;
        for (const auto& fc : face) {
            int ce[4], leaving[4], n = 0;  // crossings met on the ccw walk: edge, 1 if the walk leaves the inside
            for (int j = 0; j < 4; ++j) {
                const int p = fc[j], q = fc[(j + 1) & 3];
                const int ip = (cs >> p) & 1, iq = (cs >> q) & 1;
                if (ip != iq) ce[n] = edge_id[p][q], leaving[n] = ip, ++n;
            }
            // segments run from an "enter" crossing to the following "leave" crossing, so that each
            // one cuts off the inside corners between them
            for (int j = 0; j < n; ++j)
                if (!leaving[j]) next[ce[j]] = ce[(j + 1) % n];
        }
        int32_t* row = tri_table + cs * 16;
        for (int j = 0; j < 16; ++j) row[j] = -1;
        int nv = 0;
        bool used[12] = {false};
        for (int start = 0; start < 12; ++start) {
            if (next[start] < 0 || used[start]) continue;
            int loop[12], len = 0;
            for (int e = start; !used[e]; e = next[e]) used[e] = true, loop[len++] = e;
            // reverse the walk (winding of case 1 = {0, 8, 3}); `start` is the loop's lowest edge id
            int rl[12];
            rl[0] = loop[0];
            for (int j = 1; j < len; ++j) rl[j] = loop[len - j];
            for (int j = 1; j + 1 < len; ++j) row[nv++] = rl[0], row[nv++] = rl[j], row[nv++] = rl[j + 1];
        }
        num_verts_table[cs] = nv;
    }
}

}  // namespace dfa
