// LDS atomic throughput of one CU (one 256-thread workgroup per CU, 256 workgroups): cycles per wave instruction of
// ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_rtn_f32 / plain ds_write_b32 / ds_read + VALU add + ds_write (racy), for
// NADDR distinct addresses per wave instruction (64 = conflict-free, 8 = eight lanes per address, 1 = all lanes on one
// word) — what the reference-mode assembly's LDS hash (solve.hip: assemble_kernel) pays per column.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int NT = 256, REP = 4000, NQ = 8;

template <int MODE>
__global__ __launch_bounds__(NT) void k(int naddr, float* __restrict__ out, long long* __restrict__ cycles) {
    __shared__ __attribute__((aligned(16))) unsigned long long mem[1024];
    for (int i = threadIdx.x; i < 1024; i += NT) mem[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t off[NQ];
    for (int q = 0; q < NQ; ++q) {
        // word index: `naddr` distinct words per wave instruction, spread over the banks, different per q and wave
        const int a = (lane % naddr) * (64 / naddr) + ((q * 7 + wave * 13) % (64 / naddr == 0 ? 1 : 64 / naddr));
        off[q]      = (uint32_t)((a + 64 * q) % 512);
    }
    __syncthreads();
    float acc          = 0.f;
    const float v      = 1.0f + 1e-3f * lane;
    const long long t0 = clock64();
    for (int rep = 0; rep < REP; ++rep) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint32_t o = off[q];
            asm volatile("" : "+v"(o));
            if (MODE == 0) atomicAdd((float*)mem + o, v);
            else if (MODE == 1) atomicAdd((unsigned int*)mem + o, 3u);
            else if (MODE == 2) atomicAdd(mem + o, 3ull);
            else if (MODE == 3) acc += atomicAdd((float*)mem + o, v);
            else if (MODE == 4) ((volatile float*)mem)[o] = v;
            else if (MODE == 5) { float x = ((volatile float*)mem)[o]; ((volatile float*)mem)[o] = x + v; }
            else if (MODE == 6) acc += __int_as_float(atomicCAS((int*)mem + o, -1, lane));
            else if (MODE == 7) acc += (float)((volatile uint8_t*)mem)[o * 4 + (lane & 3)];                       // ds_read_u8
            else if (MODE == 8) ((volatile uint8_t*)mem)[o * 4 + (lane & 3)] = (uint8_t)lane;                      // ds_write_b8
            else if (MODE == 9) { volatile uint16_t* h = (volatile uint16_t*)mem + 2 * o + (lane & 1); *h = (uint16_t)(*h + 1); }  // 16-bit read-modify-write
            else if (MODE == 10) acc += ((volatile float*)mem)[o];                                                 // ds_read_b32
            else if (MODE == 11) { float4 x; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"((o & 127u) * 16u)); acc += x.x + x.w; }          // ds_read_b128
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    out[blockIdx.x * NT + threadIdx.x] = acc + ((float*)mem)[threadIdx.x];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, sizeof(float) * NT * 256);
    hipMalloc(&cyc, sizeof(long long) * 256);
    const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_add_rtn_f32", "ds_write_b32", "read+add+write", "ds_cmpst_rtn_b32",
                           "ds_read_u8", "ds_write_b8", "u16 read+add+write", "ds_read_b32", "ds_read_b128"};
    const int naddrs[] = {64, 32, 16, 8, 4, 2, 1};
    printf("cycles per wave instruction (4 waves of one workgroup per CU issuing back to back; 256 workgroups)\n%-18s", "addresses/instr:");
    for (int n : naddrs) printf("%8d", n);
    printf("\n");
    for (int mode = 0; mode < 12; ++mode) {
        printf("%-18s", names[mode]);
        for (int n : naddrs) {
            auto launch = [&](int m) {
                switch (m) {
                    case 0: k<0><<<256, NT>>>(n, out, cyc); break;
                    case 1: k<1><<<256, NT>>>(n, out, cyc); break;
                    case 2: k<2><<<256, NT>>>(n, out, cyc); break;
                    case 3: k<3><<<256, NT>>>(n, out, cyc); break;
                    case 4: k<4><<<256, NT>>>(n, out, cyc); break;
                    case 5: k<5><<<256, NT>>>(n, out, cyc); break;
                    case 6: k<6><<<256, NT>>>(n, out, cyc); break;
                    case 7: k<7><<<256, NT>>>(n, out, cyc); break;
                    case 8: k<8><<<256, NT>>>(n, out, cyc); break;
                    case 9: k<9><<<256, NT>>>(n, out, cyc); break;
                    case 10: k<10><<<256, NT>>>(n, out, cyc); break;
                    default: k<11><<<256, NT>>>(n, out, cyc); break;
                }
            };
            launch(mode);
            launch(mode);
            hipDeviceSynchronize();
            long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0;
            for (long long x : h) s += (double)x;
            // per CU: 4 waves x REP x NQ wave instructions in (s / 256) cycles
            printf("%8.1f", (s / 256.0) / (4.0 * REP * NQ));
        }
        printf("\n");
    }
    return 0;
}
