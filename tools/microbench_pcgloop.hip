// microbench: one iteration of a PERSISTENT multi-workgroup PCG — every workgroup publishes its slice of three vectors
// (agent-scope relaxed stores = global_store ... sc1, write-through), a software grid barrier (bounded spin), then every
// lane gathers NG x 3 entries of the vectors published by the others (agent-scope relaxed loads = sc1) — against the
// 7.2 - 8.5 us that one launch per iteration costs (s6_pcg_step_kernel).  usage: ./microbench_pcgloop [blocks] [iters] [NG]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Barrier { unsigned int count, gen; };

__device__ bool grid_sync(Barrier* b, unsigned nblocks, int* abort_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (++spins > (1L << 24)) {
                    *abort_flag = 1;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// RING = 1: every iteration publishes into buffers nobody has read before (u + it * n ...): no cache anywhere can hold a
// stale copy of them, so the gathers are PLAIN loads (L2-served inside an XCD).  Relies on caches being invalidated at the
// kernel boundary and on nothing fetching a line before its barrier — measured for the record, not used by the library.
template <int NT, int RING>
__global__ __launch_bounds__(NT) void k(Barrier* b, int iters, int ng, float* u0, float* m0, float* t0, int n, const int* cols,
                                        int* abort_flag, float* out, int* bad) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float* u = u0 + (RING ? (size_t)it * n : 0);
        float* m = m0 + (RING ? (size_t)it * n : 0);
        float* t = t0 + (RING ? (size_t)it * n : 0);
        if (gid < n) {  // publish this lane's entries
            __hip_atomic_store(&u[gid], acc + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&m[gid], acc + 2.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t[gid], acc + 3.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_sync(b, gridDim.x, abort_flag)) return;
        float s = 0.f;
        for (int q = 0; q < ng; ++q) {
            const int c = cols[(size_t)gid * ng + q];
            float a, bb, cc;
            if (RING) {
                a = u[c], bb = m[c], cc = t[c];
            } else {
                a  = __hip_atomic_load(&u[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bb = __hip_atomic_load(&m[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cc = __hip_atomic_load(&t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // published this iteration: (x + 1, x + 2, x + 3) of the SAME x, and never 0 (the buffers start at 0)
            if (!(bb - a == 1.f && cc - a == 2.f) && !(a > 1e6f)) atomicAdd(bad, 1);
            s += a + bb + cc;
        }
        acc = (float)(it + 1);  // the same in every lane: what every gather must see next iteration is (it+2, it+3, it+4)
        if (ng > 0 && s != (float)ng * (3.f * (float)it + 6.f)) atomicAdd(bad, 1);
        // (a second barrier is not needed in the real kernel: vectors ping-pong between two buffers)
    }
    if (gid < n) out[gid] = acc;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 64, iters = argc > 2 ? atoi(argv[2]) : 500, ng = argc > 3 ? atoi(argv[3]) : 15;
    const int ring = argc > 4 ? atoi(argv[4]) : 0;
    constexpr int NT = 1024;
    const int n = blocks * NT;
    Barrier* b; float *u, *m, *t, *out; int *ab, *cols, *bad;
    hipMalloc(&b, sizeof(Barrier)); hipMemset(b, 0, sizeof(Barrier));
    const size_t vb = 4 * (size_t)n * (ring ? iters : 1);
    hipMalloc(&u, vb); hipMalloc(&m, vb); hipMalloc(&t, vb); hipMalloc(&out, 4 * n);
    hipMemset(u, 0, vb); hipMemset(m, 0, vb); hipMemset(t, 0, vb);
    hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipMalloc(&ab, 4); hipMemset(ab, 0, 4);
    int* hc = (int*)malloc(sizeof(int) * (size_t)n * ng);
    srand(1);
    for (size_t i = 0; i < (size_t)n * ng; ++i) hc[i] = rand() % n;
    hipMalloc(&cols, sizeof(int) * (size_t)n * ng); hipMemcpy(cols, hc, sizeof(int) * (size_t)n * ng, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        int nn = n, it = iters, g = ng;
        void* args[] = {&b, &it, &g, &u, &m, &t, &nn, &cols, &ab, &out, &bad};
        hipMemset(u, 0, vb); hipMemset(m, 0, vb); hipMemset(t, 0, vb);  // (and the ring is written again by the next launch)
        hipEventRecord(e0);
        hipError_t e = ring ? hipLaunchCooperativeKernel((void*)k<NT, 1>, dim3(blocks), dim3(NT), args, 0, 0)
                            : hipLaunchCooperativeKernel((void*)k<NT, 0>, dim3(blocks), dim3(NT), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int h, hb; hipMemcpy(&h, ab, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        if (rep) printf("blocks %d x %d threads, %d gathers x 3, ring %d: %s, %.3f us per iteration, abort %d, wrong values %d\n", blocks, NT, ng, ring, hipGetErrorString(e), ms * 1e3 / iters, h, hb);
    }
    return 0;
}
