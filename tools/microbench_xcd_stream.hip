// What would a ONE-XCD persistent form of the north-star PCG (s6_pcg_step_kernel) pay per iteration just to read its matrix?
// The team PCG of reference mode (solve.hip: pcg_team_kernel) keeps the matrix in the registers of the 32 CUs that share an
// L2.  The 6 x 6-block matrix of the north-star solve is 4.5 MB at C2, 18.7 MB at C3, 37 MB at C4 — more than 32 CUs' registers
// (16 MiB) and more than the XCD's 4 MiB L2 from C3 on — so a one-XCD form re-reads it every iteration through ONE XCD's path to
// the Infinity Cache.  This measures that: the workgroups that land on XCD 0 (XCC_ID census, 1024 threads each) stream a buffer
// of the matrix's size `passes` times with 16-byte loads; the others leave.  Compared with all XCDs streaming the same buffer
// (what the launched kernel does: every XCD reads its eighth).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_xcd_stream.hip -o /tmp/xs && /tmp/xs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

template <bool ONE_XCD>
__global__ __launch_bounds__(1024) void k(const float4* __restrict__ m, size_t n4, int passes, unsigned* members, float* sink) {
    __shared__ unsigned me_sh;
    if (ONE_XCD && xcc_id() != 0) return;
    if (threadIdx.x == 0) me_sh = atomicAdd(members, 1u);
    __syncthreads();
    const unsigned me = me_sh, nmem = ONE_XCD ? 32u : gridDim.x;  // (a full team: 32 workgroups of 1024 threads on the XCD)
    if (me >= nmem) return;
    float acc = 0.f;
    for (int p = 0; p < passes; ++p) {
        // member `me` reads its contiguous share, four loads in flight per thread
        const size_t share = (n4 + nmem - 1) / nmem, b = me * share, e = b + share < n4 ? b + share : n4;
        for (size_t i = b + threadIdx.x; i < e; i += 4 * 1024) {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = m[i + q * 1024 < e ? i + q * 1024 : e - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].w;
        }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main() {
    const double sizes_mb[] = {4.5, 18.7, 37.0};
    const char* names[]     = {"C2", "C3", "C4"};
    unsigned* members;
    float* sink;
    hipMalloc(&members, 4), hipMalloc(&sink, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int c = 0; c < 3; ++c) {
        const size_t n4 = (size_t)(sizes_mb[c] * 1e6 / 16);
        float4* m;
        hipMalloc(&m, n4 * 16);
        hipMemset(m, 0, n4 * 16);
        const int passes = 50;
        for (int one = 1; one >= 0; --one)
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(members, 0, 4);
                hipEventRecord(e0);
                if (one) k<true><<<256, 1024>>>(m, n4, passes, members, sink);
                else k<false><<<256, 1024>>>(m, n4, passes, members, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned h;
                hipMemcpy(&h, members, 4, hipMemcpyDeviceToHost);
                if (rep)
                    printf("%s matrix %.1f MB, %s: %u workgroups took part, %.2f us per pass, %.0f GB/s\n", names[c], sizes_mb[c],
                           one ? "ONE XCD (32 x 1024 threads)" : "all XCDs (256 x 1024 threads)", h, ms * 1e3 / passes,
                           sizes_mb[c] * 1e6 / (ms * 1e-3 / passes) / 1e9);
            }
        hipFree(m);
    }
    return 0;
}
