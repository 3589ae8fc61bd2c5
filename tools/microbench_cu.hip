// dev tool: single-workgroup phase costs on one CU (barrier, DPP reduction, LDS gather) and the
// shader clock a lone workgroup actually gets.   hipcc --offload-arch=gfx950 -O3 -o mb microbench_cu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../dynfu_amd/csrc/device_math.hpp"
using namespace dfa;

template <int NT>
__global__ __launch_bounds__(NT) void k_phases(int iters, int mode, const int* cols, float* out, long long* cyc) {
    __shared__ float4 p_s[2048];
    __shared__ float red[2][16];
    for (int i = threadIdx.x; i < 2048; i += NT) p_s[i] = make_float4(i, 1.f, 2.f, 0.f);
    int c[16];
    for (int q = 0; q < 16; ++q) c[q] = cols[q * NT + threadIdx.x];
    __syncthreads();
    long long t0 = clock64(), w0 = wall_clock64();
    float acc = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
            __syncthreads();
        } else if (mode == 1) {
            float w = wave_total(acc);
            if ((threadIdx.x & 63) == 0) red[it & 1][threadIdx.x >> 6] = w;
            __syncthreads();
            float tot = 0.f;
            for (int i = 0; i < NT / 64; ++i) tot += red[it & 1][i];
            acc = tot * 1e-9f + 1.f;
        } else if (mode == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float4 pc = p_s[(c[q] + it) & 2047];
                acc = fmaf(pc.x, 1e-9f, acc);
            }
        } else if (mode == 3) {
            for (int q = 0; q < 256; ++q) acc = fmaf(acc, 1.000001f, 1e-9f);
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = w1 - w0;
}

template <int NT>
void run(const char* name) {
    int* cols;
    float* out;
    long long* cyc;
    hipMalloc(&cols, 16 * NT * 4);
    hipMalloc(&out, NT * 4);
    hipMalloc(&cyc, 16);
    std::vector<int> h(16 * NT);
    unsigned s = 12345;
    for (auto& v : h) s = s * 1664525u + 1013904223u, v = (s >> 8) & 2047;
    hipMemcpy(cols, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* modes[] = {"barrier", "block_sum(dpp+1 barrier)", "16 random ds_read_b128 / thread", "256 dependent fma"};
    for (int mode = 0; mode < 4; ++mode) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            hipEventRecord(e0);
            k_phases<NT><<<1, NT>>>(iters, mode, cols, out, cyc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long c[2];
            hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            if (rep == 1)
                printf("%s NT=%d %-34s %8.3f us/iter  %8.1f shader-cycles/iter  clock %.0f MHz (wall100MHz ticks %lld)\n", name, NT,
                       modes[mode], ms * 1e3 / iters, (double)c[0] / iters, (double)c[0] / (ms * 1e3), c[1]);
        }
    }
}

int main() {
    run<1024>("A");
    run<512>("B");
    run<256>("C");
    return 0;
}
