// device.cpp — hipMalloc-backed containers and error translation of the host adaptors.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <dfa_host/device.hpp>
#include <dfa_host/types.hpp>

#include "../../../include/dynfu_amd.h"

namespace dfa {

// The adaptors pass dfa_solve_params / dfa_solve6_params / ..._stats by pointer: a libdynfu_amd.so built from another
// header would read or write past them.  Checked once, on the first checked call.
static void require_matching_abi() {
    static const bool ok = [] {
        if (dfa_abi_version() != DFA_ABI_VERSION)
            throw Error(DFA_ERR_INVALID, "libdynfu_amd.so implements ABI version " + std::to_string(dfa_abi_version()) +
                                             ", the host adaptors were built against " + std::to_string(DFA_ABI_VERSION));
        const size_t mine[] = {sizeof(dfa_solve_params), sizeof(dfa_solve_stats), sizeof(dfa_solve_timing),
                               sizeof(dfa_solve6_params), sizeof(dfa_solve6_stats), sizeof(dfa_solve6_timing)};
        for (int id = 0; id < 6; ++id)
            if (dfa_abi_struct_size(id) != mine[id])
                throw Error(DFA_ERR_INVALID, "libdynfu_amd.so and the host adaptors disagree on the size of struct " + std::to_string(id));
        return true;
    }();
    (void)ok;
}

bool host_switch(const char* name) {
    static std::mutex mu;
    static std::map<std::string, bool> seen;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(name);
    if (it == seen.end()) it = seen.emplace(name, std::getenv(name) != nullptr).first;
    return it->second;
}

void check(int rc, const char* where) {
    require_matching_abi();
    if (rc != DFA_OK) throw Error(rc, std::string(where) + ": " + dfa_last_error());
}

static void hip_check(hipError_t e, const char* where) {
    if (e != hipSuccess) throw Error(DFA_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e));
}

namespace {
// The reference's classes allocate per call (a CombinedSolver, its images and every staging array live for one frame);
// hipMalloc / hipFree cost ~0.1 ms each and hipFree waits for the device.  Blocks are therefore recycled: sizes are
// rounded up to a power of two and released blocks parked per (device, size), up to 2 GiB per process, beyond which
// they really are freed.  Safe because the adaptor classes do everything on the default stream (as the reference does):
// a block released while a kernel still reads it is next touched by its new owner's upload or kernel on that same
// stream, i.e. behind that kernel; hipMemcpy itself is synchronous.
struct BlockPool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> idle;
    size_t parked = 0;
    static size_t round_up(size_t n) {
        size_t r = 256;
        while (r < n) r <<= 1;
        return r;
    }
    void* take(size_t bytes, size_t& cap) {
        cap     = round_up(bytes);
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lock(mu);
            auto& v = idle[{dev, cap}];
            if (!v.empty()) {
                void* p = v.back();
                v.pop_back();
                parked -= cap;
                return p;
            }
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) {  // out of memory with blocks parked: release them and retry once
            trim();
            e = hipMalloc(&p, cap);
        }
        if (e != hipSuccess) throw dfa::Error(DFA_ERR_HIP, std::string("DeviceMemory::create: ") + hipGetErrorString(e));
        return p;
    }
    void give(void* p, size_t cap) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> lock(mu);
            if (parked + cap <= ((size_t)2 << 30)) {
                idle[{dev, cap}].push_back(p);
                parked += cap;
                return;
            }
        }
        (void)hipFree(p);
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& kv : idle)
            for (void* p : kv.second) (void)hipFree(p);
        idle.clear();
        parked = 0;
    }
};
BlockPool& pool() {
    static BlockPool* p = new BlockPool();  // never destroyed: blocks may be released after main() returns
    return *p;
}
}  // namespace

void DeviceMemory::create(size_t bytes) {
    if (bytes == size_ && data_) return;
    void* p    = nullptr;
    size_t cap = 0;
    if (bytes) p = pool().take(bytes, cap);
    data_ = std::shared_ptr<void>(p, [cap](void* q) {
        if (q) pool().give(q, cap);
    });
    size_ = bytes;
}
void DeviceMemory::upload(const void* host, size_t bytes) {
    create(bytes);
    if (bytes) hip_check(hipMemcpy(data_.get(), host, bytes, hipMemcpyHostToDevice), "DeviceMemory::upload");
}
void DeviceMemory::download(void* host, size_t bytes) const {
    if (bytes) hip_check(hipMemcpy(host, data_.get(), bytes, hipMemcpyDeviceToHost), "DeviceMemory::download");
}
void copy2d_h2d(void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, int rows) {
    hip_check(hipMemcpy2D(dst, dstep, src, sstep, width_bytes, rows, hipMemcpyHostToDevice), "DeviceArray2D::upload");
}
void copy2d_d2h(void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, int rows) {
    hip_check(hipMemcpy2D(dst, dstep, src, sstep, width_bytes, rows, hipMemcpyDeviceToHost),
              "DeviceArray2D::download");
}
void device_synchronize() { hip_check(hipDeviceSynchronize(), "device_synchronize"); }

}  // namespace dfa

void dfa_host_copy_from_device(void* dst, const void* src, size_t bytes) {
    if (bytes && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        throw dfa::Error(DFA_ERR_HIP, "copy from device failed");
}
