// microbench: cost of a software grid barrier on MI355X (all workgroups co-resident), with and without an
// agent-scope release/acquire fence around it.  usage: ./microbench_gridbarrier [blocks] [barriers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Barrier { unsigned int count, gen; };

template <bool FENCE>
__device__ bool grid_sync(Barrier* b, unsigned nblocks, int* abort_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default for device code
        const unsigned gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 22)) {  // ~ a second: give up instead of hanging the GPU
                    *abort_flag = 1;
                    break;
                }
            }
        }
        if (FENCE) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return *abort_flag == 0;
}

template <bool FENCE>
__global__ __launch_bounds__(256) void k(Barrier* b, int n, float* data, int* abort_flag) {
    float v = 0.f;
    for (int i = 0; i < n; ++i) {
        data[blockIdx.x * 256 + threadIdx.x] = v + 1.f;  // some traffic the fence has to publish
        if (!grid_sync<FENCE>(b, gridDim.x, abort_flag)) return;
        v += data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    data[blockIdx.x * 256 + threadIdx.x] = v;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 1000;
    Barrier* b; float* data; int* ab;
    hipMalloc(&b, sizeof(Barrier)); hipMemset(b, 0, sizeof(Barrier));
    hipMalloc(&data, sizeof(float) * 256 * blocks); hipMemset(data, 0, sizeof(float) * 256 * blocks);
    hipMalloc(&ab, 4); hipMemset(ab, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fence = 0; fence < 2; ++fence) {
        for (int rep = 0; rep < 2; ++rep) {
            void* args[] = {&b, (void*)&n, &data, &ab};
            hipEventRecord(e0);
            hipError_t e = fence ? hipLaunchCooperativeKernel((void*)k<true>, dim3(blocks), dim3(256), args, 0, 0)
                                 : hipLaunchCooperativeKernel((void*)k<false>, dim3(blocks), dim3(256), args, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int h; hipMemcpy(&h, ab, 4, hipMemcpyDeviceToHost);
            if (rep) printf("blocks %d fence %d: %s, %.3f us per barrier, abort %d\n", blocks, fence, hipGetErrorString(e), ms * 1e3 / n, h);
        }
    }
    return 0;
}
