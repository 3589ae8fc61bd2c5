// LDS gather throughput of one 1024-thread workgroup: random 4-byte reads (what the register-resident PCG does per
// non-zero) against layouts that replicate the vector so that a lane's copy index selects its banks.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_lds_gather.hip -o /tmp/lds_gather && /tmp/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 1024, NG = 32, REP = 2000;

// mode R: element i of copy c at word i*R + c, lane reads copy (lane % R); R = 1 is the plain layout
template <int R, int WIDTH /* bytes per read: 4 or 16 */>
__global__ __launch_bounds__(NT) void gather_kernel(const uint16_t* __restrict__ cols, float* __restrict__ out, int D,
                                                    long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p = (float*)smem;
    const int words = D * R * (WIDTH / 4);
    for (int i = threadIdx.x; i < words; i += NT) p[i] = (float)(i % 97);
    uint32_t off[NG];
    for (int q = 0; q < NG; ++q) {
        const int col = cols[q * NT + threadIdx.x];
        off[q]        = WIDTH == 4 ? (uint32_t)(col * R + (threadIdx.x % R)) * 4u : (uint32_t)col * (uint32_t)WIDTH;
    }
    __syncthreads();
    float acc          = 0.f;
    const long long t0 = clock64();
    for (int rep = 0; rep < REP; ++rep) {
#pragma unroll
        for (int q = 0; q < NG; q += 4) {
            uint32_t o0 = off[q], o1 = off[q + 1], o2 = off[q + 2], o3 = off[q + 3];
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            if (WIDTH == 4) {
                acc += *(const float*)(smem + o0) + *(const float*)(smem + o1) + *(const float*)(smem + o2) +
                       *(const float*)(smem + o3);
            } else if (WIDTH == 8) {  // 8-byte elements (value + pad): ds_read_b64 banks on (a / 4) mod 64
                const float2 a = *(const float2*)(smem + o0), b = *(const float2*)(smem + o1), c = *(const float2*)(smem + o2),
                             d = *(const float2*)(smem + o3);
                acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
            } else {
                const float4 a = *(const float4*)(smem + o0), b = *(const float4*)(smem + o1),
                             c = *(const float4*)(smem + o2), d = *(const float4*)(smem + o3);
                acc += a.x + a.w + b.y + b.w + c.z + c.w + d.x + d.w;
            }
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    out[threadIdx.x]   = acc;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

// NGLOB of the NG gathers per sweep go through the vector-memory path (L1-resident 8 KB array in global memory),
// the rest through LDS: do the two pipes overlap?
template <int NGLOB>
__global__ __launch_bounds__(NT) void mixed_kernel(const uint16_t* __restrict__ cols, const float* __restrict__ gp,
                                                   float* __restrict__ out, int D, long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* p = (float*)smem;
    for (int i = threadIdx.x; i < D; i += NT) p[i] = (float)(i % 97);
    uint32_t off[NG];
    for (int q = 0; q < NG; ++q) off[q] = (uint32_t)cols[q * NT + threadIdx.x] * 4u;
    __syncthreads();
    float acc          = 0.f;
    const char* gbase  = (const char*)gp;
    const long long t0 = clock64();
    for (int rep = 0; rep < REP; ++rep) {
        float g[NGLOB > 0 ? NGLOB : 1];
#pragma unroll
        for (int q = 0; q < NGLOB; ++q) {
            uint32_t o = off[q];
            asm volatile("" : "+v"(o));
            g[q] = *(const float*)(gbase + o);
        }
#pragma unroll
        for (int q = NGLOB; q < NG; q += 4) {
            uint32_t o0 = off[q], o1 = off[q + 1], o2 = off[q + 2], o3 = off[q + 3];
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            acc += *(const float*)(smem + o0) + *(const float*)(smem + o1) + *(const float*)(smem + o2) +
                   *(const float*)(smem + o3);
        }
#pragma unroll
        for (int q = 0; q < NGLOB; ++q) acc += g[q];
        __syncthreads();
    }
    const long long t1 = clock64();
    out[threadIdx.x]   = acc;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int NGLOB>
void run_mixed(const uint16_t* cols, const float* gp, float* out, long long* cyc, int D) {
    mixed_kernel<NGLOB><<<1, NT, D * 4>>>(cols, gp, out, D, cyc);
    mixed_kernel<NGLOB><<<1, NT, D * 4>>>(cols, gp, out, D, cyc);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("random: %2d of %d gathers via global/L1, rest LDS: %.0f clk per sweep  err=%d\n", NGLOB, NG, (double)h / REP,
           (int)hipGetLastError());
}

template <int R, int WIDTH>
void run(const char* name, const uint16_t* cols, float* out, long long* cyc, int D) {
    const size_t sh = (size_t)D * R * WIDTH;
    hipFuncSetAttribute((const void*)gather_kernel<R, WIDTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    gather_kernel<R, WIDTH><<<1, NT, sh>>>(cols, out, D, cyc);
    hipEventRecord(a);
    gather_kernel<R, WIDTH><<<1, NT, sh>>>(cols, out, D, cyc);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / REP / (NG * (NT / 64));
    printf("%-34s %8.3f ms  %6.2f clk per wave-instruction  (%.0f clk per sweep of %d reads x 16 waves)  err=%d\n", name, ms,
           per, (double)h / REP, NG, (int)hipGetLastError());
}

// index tables from a file ([n][NG][NT] uint16, e.g. the slot -> column assignments of a real normal matrix): names on the command line
static void run_tables(const char* path, int argc, char** argv, float* out, long long* cyc, int D) {
    FILE* f = fopen(path, "rb");
    if (!f) return;
    std::vector<uint16_t> t(NG * NT);
    uint16_t* d;
    hipMalloc(&d, t.size() * 2);
    for (int i = 0; fread(t.data(), 2, t.size(), f) == t.size(); ++i) {
        hipMemcpy(d, t.data(), t.size() * 2, hipMemcpyHostToDevice);
        run<1, 4>(2 + i < argc ? argv[2 + i] : "table", d, out, cyc, D);
    }
    fclose(f);
}

int main(int argc, char** argv) {
    const int D = 2048;
    std::vector<uint16_t> cols(NG * NT), lin(NG * NT);
    srand(1);
    for (auto& c : cols) c = rand() % D;
    for (int q = 0; q < NG; ++q)
        for (int t = 0; t < NT; ++t) lin[q * NT + t] = (t + 37 * q) % D;
    uint16_t *d_cols, *d_lin;
    float* out;
    long long* cyc;
    hipMalloc(&d_cols, cols.size() * 2), hipMalloc(&d_lin, cols.size() * 2), hipMalloc(&out, NT * 4), hipMalloc(&cyc, 8);
    hipMemcpy(d_cols, cols.data(), cols.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_lin, lin.data(), lin.size() * 2, hipMemcpyHostToDevice);
    if (argc > 1) {
        run_tables(argv[1], argc, argv, out, cyc, D);
        return 0;
    }
    run<1, 4>("b32 consecutive lanes (no conflicts)", d_lin, out, cyc, D);
    run<1, 4>("b32 random", d_cols, out, cyc, D);
    run<2, 4>("b32 random, 2 copies", d_cols, out, cyc, D);
    run<4, 4>("b32 random, 4 copies", d_cols, out, cyc, D);
    run<8, 4>("b32 random, 8 copies", d_cols, out, cyc, D);
    run<16, 4>("b32 random, 16 copies", d_cols, out, cyc, D);
    run<1, 8>("b64 consecutive lanes", d_lin, out, cyc, D);
    run<1, 8>("b64 random (8-byte elements)", d_cols, out, cyc, D);
    run<1, 16>("b128 consecutive lanes", d_lin, out, cyc, D);
    run<1, 16>("b128 random", d_cols, out, cyc, D);
    float* gp;
    hipMalloc(&gp, D * 4);
    hipMemset(gp, 0, D * 4);
    run_mixed<0>(d_cols, gp, out, cyc, D);
    run_mixed<4>(d_cols, gp, out, cyc, D);
    run_mixed<8>(d_cols, gp, out, cyc, D);
    run_mixed<12>(d_cols, gp, out, cyc, D);
    run_mixed<16>(d_cols, gp, out, cyc, D);
    run_mixed<32>(d_cols, gp, out, cyc, D);
    return 0;
}
