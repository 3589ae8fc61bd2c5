// device.hpp — device memory containers for the host adaptors: the slice of
// kfusion::cuda::DeviceMemory / DeviceArray / DeviceArray2D (include/kfusion/cuda/device_memory.hpp,
// device_array.hpp) the TSDF and solver seams need — ref-counted hipMalloc blobs, typed 1-D
// arrays, pitched 2-D images with upload / download, and swap.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>

#include "types.hpp"

namespace dfa {

class DeviceMemory {  // kfusion::cuda::DeviceMemory ("CudaData")
public:
    DeviceMemory() : size_(0) {}
    explicit DeviceMemory(size_t bytes) { create(bytes); }
    // non-owning wrapper of user memory (device_memory.cpp:54-56)
    DeviceMemory(void* ptr, size_t bytes) : data_(ptr, [](void*) {}), size_(bytes) {}
    void create(size_t bytes);  // no-op when the size is unchanged (device_memory.cpp:90-104)
    void release() {
        data_.reset();
        size_ = 0;
    }
    void swap(DeviceMemory& o) {
        data_.swap(o.data_);
        std::swap(size_, o.size_);
    }
    bool empty() const { return !data_; }
    // no other handle (copy of this object, a CudaData taken from its owner) refers to the same storage
    bool unique() const { return data_.use_count() <= 1; }
    size_t sizeBytes() const { return size_; }
    template <class T>
    T* ptr() {
        return (T*)data_.get();
    }
    template <class T>
    const T* ptr() const {
        return (const T*)data_.get();
    }
    void upload(const void* host, size_t bytes);
    void download(void* host, size_t bytes) const;

private:
    std::shared_ptr<void> data_;  // ref-counted like the reference's refcount_
    size_t size_;
};

template <class T>
class DeviceArray : public DeviceMemory {
public:
    DeviceArray() {}
    explicit DeviceArray(size_t n) : DeviceMemory(n * sizeof(T)) {}
    DeviceArray(T* ptr, size_t n) : DeviceMemory(ptr, n * sizeof(T)) {}  // non-owning (device_array.hpp)
    void create(size_t n) { DeviceMemory::create(n * sizeof(T)); }
    void upload(const std::vector<T>& v) {
        create(v.size());
        DeviceMemory::upload(v.data(), v.size() * sizeof(T));
    }
    void download(std::vector<T>& v) const {
        v.resize(size());
        DeviceMemory::download(v.data(), v.size() * sizeof(T));
    }
    size_t size() const { return sizeBytes() / sizeof(T); }
    T* ptr() { return DeviceMemory::ptr<T>(); }
    const T* ptr() const { return DeviceMemory::ptr<T>(); }
};

// pitched image: rows x cols elements, step() bytes per row (multiple of 256 like cudaMallocPitch)
template <class T>
class DeviceArray2D {
public:
    DeviceArray2D() : rows_(0), cols_(0), step_(0) {}
    DeviceArray2D(int rows, int cols) { create(rows, cols); }
    void create(int rows, int cols) {
        if (rows == rows_ && cols == cols_ && !mem_.empty()) return;
        rows_ = rows, cols_ = cols;
        step_ = ((size_t)cols * sizeof(T) + 255) & ~(size_t)255;
        mem_.create(step_ * (size_t)rows);
    }
    void upload(const T* host, size_t host_step, int rows, int cols);
    void download(T* host, size_t host_step) const;
    void upload(const std::vector<T>& v, int cols) { upload(v.data(), (size_t)cols * sizeof(T), (int)(v.size() / cols), cols); }
    void download(std::vector<T>& v, int& cols) const {
        cols = cols_;
        v.resize((size_t)rows_ * cols_);
        download(v.data(), (size_t)cols_ * sizeof(T));
    }
    void swap(DeviceArray2D& o) {
        mem_.swap(o.mem_);
        std::swap(rows_, o.rows_), std::swap(cols_, o.cols_), std::swap(step_, o.step_);
    }
    bool empty() const { return mem_.empty(); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    size_t step() const { return step_; }
    T* ptr() { return mem_.ptr<T>(); }
    const T* ptr() const { return mem_.ptr<T>(); }

private:
    DeviceMemory mem_;
    int rows_, cols_;
    size_t step_;
};

void copy2d_h2d(void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, int rows);
void copy2d_d2h(void* dst, size_t dstep, const void* src, size_t sstep, size_t width_bytes, int rows);

template <class T>
void DeviceArray2D<T>::upload(const T* host, size_t host_step, int rows, int cols) {
    create(rows, cols);
    copy2d_h2d(mem_.ptr<void>(), step_, host, host_step, (size_t)cols * sizeof(T), rows);
}
template <class T>
void DeviceArray2D<T>::download(T* host, size_t host_step) const {
    copy2d_d2h(host, host_step, mem_.ptr<void>(), step_, (size_t)cols_ * sizeof(T), rows_);
}

void device_synchronize();  // kfusion::cuda::waitAllDefaultStream

// A/B switches of the host adaptors (DFA_HOST_*): read ONCE per process, never on a frame's call path.
bool host_switch(const char* name);

}  // namespace dfa
