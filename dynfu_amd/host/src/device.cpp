// device.cpp — hipMalloc-backed containers and error translation of the host adaptors.
#include <hip/hip_runtime.h>

#include <dfa_host/device.hpp>
#include <dfa_host/types.hpp>

#include "../../../include/dynfu_amd.h"

namespace dfa {

void check(int rc, const char* where) {
    if (rc != DFA_OK) throw Error(rc, std::string(where) + ": " + dfa_last_error());
}

static void hip_check(hipError_t e, const char* where) {
    if (e != hipSuccess) throw Error(DFA_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e));
}

void DeviceMemory::create(size_t bytes) {
    if (bytes == size_ && data_) return;
    void* p = nullptr;
    if (bytes) hip_check(hipMalloc(&p, bytes), "DeviceMemory::create");
    data_ = std::shared_ptr<void>(p, [](void* q) {
        if (q) (void)hipFree(q);
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
