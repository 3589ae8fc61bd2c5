// The north-star assembly's moment accumulation S_ab += (h_a h_b) l l^T (8 x 8 per record) in two forms, on rows staged in
// LDS exactly as s6_assemble2_kernel stages them (80-byte records: l 8 floats | h 8 floats | 4), one 256-thread workgroup
// per CU, 4 workgroups' worth of waves per SIMD as in the kernel:
//   VALU  a lane per work unit owns the 36 distinct entries of its moment (16 packed + 4 scalar FMAs + 4 packed multiplies
//         per record, 2 ds_read_b128 + 2 ds_read_b32) — the kernel's inner loop;
//   MFMA  v_mfma_f32_16x16x4_f32: a wave takes 8 records of ONE block per instruction — rows i = (half, component) of A hold
//         c l of record 4 half + k, columns j of B hold l of the same record; the two 8 x 8 diagonal blocks of the 16 x 16
//         result are the block's moment, the two off-diagonal ones are products of different records (discarded): half of
//         the instruction's 2 048 FLOP are used, and the moment's symmetry (36 of 64 entries) cannot be used at all.
// Prints cycles per record and SIMD for both.   hipcc --offload-arch=gfx950 -O3 tools/microbench_moment.hip -o /tmp/mm && /tmp/mm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int R = 352, RS = 80, REP = 64, NREC = 512;  // staged rows; bytes per row; repetitions; records per unit / wave-group
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 4) void valu_kernel(const uint32_t* __restrict__ recs, float* __restrict__ out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char rows[];
    for (int i = threadIdx.x; i < R * RS / 4; i += 256) ((float*)rows)[i] = 0.001f * (float)((i * 37) % 101);
    __syncthreads();
    v2f me[10], mo[6];
    float md[4];
    for (int e = 0; e < 10; ++e) me[e] = v2f{0.f, 0.f};
    for (int e = 0; e < 6; ++e) mo[e] = v2f{0.f, 0.f};
    for (int e = 0; e < 4; ++e) md[e] = 0.f;
    const long long t0 = clock64();
    for (int rep = 0; rep < REP; ++rep)
        for (int g = 0; g < NREC; g += 8) {
            uint32_t rb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) rb[q] = recs[(size_t)(g + q) * 256 + threadIdx.x];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t pr = rb[q];
                const char* rp    = rows + RS * (pr >> 8);
                const float4 la = reinterpret_cast<const float4*>(rp)[0], lb = reinterpret_cast<const float4*>(rp)[1];
                const float ho = *reinterpret_cast<const float*>(rp + 32 + ((pr >> 2) & 28u));
                const float hj = *reinterpret_cast<const float*>(rp + 32 + ((pr << 2) & 28u));
                const float cf = ho * hj;
                const v2f L[4] = {v2f{la.x, la.y}, v2f{la.z, la.w}, v2f{lb.x, lb.y}, v2f{lb.z, lb.w}};
                const v2f cc   = v2f{cf, cf};
                v2f F[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) F[t] = cc * L[t];
                constexpr int eo[4] = {0, 4, 7, 9}, oo[4] = {0, 3, 5, 6};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v2f fe = v2f{F[t].x, F[t].x}, fo = v2f{F[t].y, F[t].y};
#pragma unroll
                    for (int u = 0; u < 4 - t; ++u) me[eo[t] + u] = __builtin_elementwise_fma(fe, L[t + u], me[eo[t] + u]);
                    md[t] = fmaf(F[t].y, L[t].y, md[t]);
#pragma unroll
                    for (int u = 0; u < 3 - t; ++u) mo[oo[t] + u] = __builtin_elementwise_fma(fo, L[t + 1 + u], mo[oo[t] + u]);
                }
            }
        }
    const long long t1 = clock64();
    float acc = 0.f;
    for (int e = 0; e < 10; ++e) acc += me[e].x + me[e].y;
    for (int e = 0; e < 6; ++e) acc += mo[e].x + mo[e].y;
    for (int e = 0; e < 4; ++e) acc += md[e];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// recs8: per wave and step the 8 records of the step (one block): the lane reads the id of ITS record (4 half + k)
__global__ __launch_bounds__(256, 4) void mfma_kernel(const uint32_t* __restrict__ recs, float* __restrict__ out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char rows[];
    for (int i = threadIdx.x; i < R * RS / 4; i += 256) ((float*)rows)[i] = 0.001f * (float)((i * 37) % 101);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int comp = lane & 7, half = (lane >> 3) & 1, kk = lane >> 4;
    v4f acc[4];  // four independent accumulators (four blocks in flight: the dependent-accumulator latency is 40 cycles)
    for (int e = 0; e < 4; ++e) acc[e] = v4f{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    // a wave consumes 8 records per MFMA; the same number of records per SIMD as the VALU kernel: 64 lanes x NREC / 8 steps
    for (int rep = 0; rep < REP; ++rep)
        for (int g = 0; g < NREC * 8; g += 4) {
            uint32_t pr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pr[q] = recs[((size_t)(g + q) * 4 + wave) * 8 % ((size_t)NREC * 256) + 4 * half + kk];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* rp = rows + RS * (pr[q] >> 8);
                const float l  = *reinterpret_cast<const float*>(rp + 4 * comp);
                const float ho = *reinterpret_cast<const float*>(rp + 32 + ((pr[q] >> 2) & 28u));
                const float hj = *reinterpret_cast<const float*>(rp + 32 + ((pr[q] << 2) & 28u));
                acc[q]         = __builtin_amdgcn_mfma_f32_16x16x4f32((ho * hj) * l, l, acc[q], 0, 0, 0);
            }
        }
    const long long t1 = clock64();
    float a = 0.f;
    for (int e = 0; e < 4; ++e) a += acc[e].x + acc[e].y + acc[e].z + acc[e].w;
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    std::vector<uint32_t> h((size_t)NREC * 256);
    uint32_t x = 12345u;
    for (auto& v : h) {
        x = x * 1664525u + 1013904223u;
        v = ((x >> 8) % R) << 8 | ((x >> 4) & 7u) << 4 | (x & 7u);
    }
    uint32_t* recs;
    float* out;
    long long* cyc;
    const int nblk = 256 * 4;  // four workgroups per CU, as the kernel runs
    hipMalloc(&recs, h.size() * 4);
    hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, sizeof(float) * 256 * nblk);
    hipMalloc(&cyc, sizeof(long long) * nblk);
    for (int form = 0; form < 2; ++form) {
        for (int warm = 0; warm < 2; ++warm) {
            if (form == 0) valu_kernel<<<nblk, 256, R * RS>>>(recs, out, cyc);
            else mfma_kernel<<<nblk, 256, R * RS>>>(recs, out, cyc);
        }
        hipDeviceSynchronize();
        std::vector<long long> c(nblk);
        hipMemcpy(c.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
        double s = 0;
        for (long long v : c) s += (double)v;
        // per workgroup: 256 lanes x NREC records x REP; a SIMD holds one wave of each of 4 workgroups -> records per SIMD in
        // the measured time = 4 x 64 x NREC x REP
        const double per_wg = s / nblk, recs_per_simd = 4.0 * 64.0 * NREC * REP;
        printf("%s: %.0f cycles per workgroup, %.2f cycles per record and SIMD (4 waves per SIMD)\n", form == 0 ? "VALU (lane per unit, 36 entries)" : "MFMA 16x16x4 f32 (8 records per instruction)",
               per_wg, per_wg / (recs_per_simd / 4.0) );
    }
    return 0;
}
