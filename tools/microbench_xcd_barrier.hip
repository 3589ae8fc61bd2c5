// A persistent kernel confined to ONE XCD: what does an iteration (publish a vector, barrier, gather a neighbour's part)
// cost when all participating workgroups share an L2?  Workgroups are dealt to the 8 XCDs by the dispatcher; each reads its
// XCC_ID (s_getreg) and only those on XCD 0 take part — the membership is COUNTED (census), never assumed, so the protocol
// does not depend on the placement.  Exchange: relaxed agent-scope stores (write through to the shared L2) + a workgroup-
// scope release (s_waitcnt vmcnt(0), no L2 write-back), an L2 atomic barrier, relaxed agent-scope loads (sc1: bypass the
// vector L1, hit the L2).  Compared with the same loop over ALL workgroups with agent-scope release / acquire fences
// (L2 write-back + invalidate: what a chip-wide persistent PCG pays).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_xcd_barrier.hip -o /tmp/xb && /tmp/xb [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Ctl {
    unsigned seen, members, count, gen, abort_flag, pad[3];
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned want_ne_or_eq, bool until_equal, Ctl* c) {
    long spins = 0;
    for (;;) {
        const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (until_equal ? v == want_ne_or_eq : v != want_ne_or_eq) return true;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 21)) {
            __hip_atomic_store(&c->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (__hip_atomic_load(&c->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
}

template <bool ONE_XCD, bool FLAGS = false>
__global__ __launch_bounds__(256) void k(Ctl* c, int n, float* data, float* out, long long* cyc, unsigned* flags) {
    __shared__ unsigned me_sh, nmem_sh, ok_sh;
    const bool member = !ONE_XCD || xcc_id() == 0;
    if (threadIdx.x == 0) {
        unsigned me = 0xffffffffu;
        if (member) me = __hip_atomic_fetch_add(&c->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&c->seen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1;
        if (member) ok = spin_until(&c->seen, gridDim.x, true, c);  // every workgroup has started: the census is final
        me_sh = me, ok_sh = ok, nmem_sh = __hip_atomic_load(&c->members, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!member || !ok_sh) return;
    const unsigned me = me_sh, nmem = nmem_sh;
    float v = 0.f;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        float* mine = data + (size_t)(i & 1) * 256 * 256 + (size_t)me * 256 + threadIdx.x;
        if (ONE_XCD) {
            __hip_atomic_store(mine, v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the stores have left this CU (write-through L1)
        } else {
            *mine = v + 1.f;
        }
        __syncthreads();
        if (FLAGS) {
            // barrier without read-modify-writes: every member publishes its round in a word of its own; the first wave polls
            // all of them with one load per lane (members <= 64)
            if (threadIdx.x == 0) __hip_atomic_store(&flags[me * 16], (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) {
                long spins = 0;
                for (;;) {
                    const unsigned f = threadIdx.x < nmem ? __hip_atomic_load(&flags[threadIdx.x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
                    if (__all((int)(f >= (unsigned)(i + 1)))) break;
                    if (++spins > (1L << 21)) {
                        ok_sh = 0;
                        __hip_atomic_store(&c->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        } else
        if (threadIdx.x == 0) {
            if (!ONE_XCD) __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope: L2 write-back
            const unsigned gen = __hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nmem - 1) {
                __hip_atomic_store(&c->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&c->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (!spin_until(&c->gen, gen, false, c)) {
                ok_sh = 0;
            }
            if (!ONE_XCD) __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope: L1 + L2 invalidate
        }
        __syncthreads();
        if (!ok_sh) return;
        const float* theirs = data + (size_t)(i & 1) * 256 * 256 + (size_t)((me + 1) % nmem) * 256 + threadIdx.x;
        v += ONE_XCD ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *theirs;
        __syncthreads();  // (the next iteration overwrites `mine`: everybody must have read... the barrier of the next round orders it)
    }
    const long long t1 = clock64();
    out[(size_t)me * 256 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[me] = t1 - t0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000, blocks = 256;
    Ctl* c;
    float *data, *out;
    long long* cyc;
    hipMalloc(&c, sizeof(Ctl));
    hipMalloc(&data, sizeof(float) * 2 * 256 * blocks);
    unsigned* flags;
    hipMalloc(&flags, sizeof(unsigned) * 16 * blocks);
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int one = 2; one >= 0; --one)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(c, 0, sizeof(Ctl));
            hipMemset(data, 0, sizeof(float) * 2 * 256 * blocks);
            hipMemset(flags, 0, sizeof(unsigned) * 16 * blocks);
            hipMemset(out, 0, sizeof(float) * 256 * blocks);
            hipEventRecord(e0);
            if (one == 2) k<true, true><<<blocks, 256>>>(c, n, data, out, cyc, flags);
            else if (one) k<true><<<blocks, 256>>>(c, n, data, out, cyc, flags);
            else k<false><<<blocks, 256>>>(c, n, data, out, cyc, flags);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            Ctl h;
            hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
            float o[256];
            hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
            // every round adds the neighbour's value of that round: v_i = sum_{j<i} (v_j + 1) for all members alike
            double expect = 0;
            for (int i = 0; i < n && i < 30; ++i) expect += expect + 1;  // (doubles every round: only small n is checked exactly)
            if (rep)
                printf("%s: members %u of %u workgroups, abort %u, %.3f us per iteration (kernel %.3f ms / %d)%s\n",
                       one == 2 ? "one XCD, flag barrier (no read-modify-write), sc1 exchange" : one ? "one XCD, atomic-counter barrier, sc1 exchange" : "all XCDs, agent-scope release / acquire",
                       h.members, h.seen, h.abort_flag, ms * 1e3 / n, ms, n, n <= 30 ? (o[0] == (float)expect ? "  values OK" : "  VALUES WRONG") : "");
        }
    return 0;
}
