// vector.hxx -- vector_t<T, space>: std::vector on the host, an hipMalloc-backed RAII array on the
// device.  Mirrors the reference's include/sparsify.me/containers/vector.hxx:18-23, which aliases
// thrust::host_vector / thrust::device_vector; this build carries its own 100-line device vector
// with the subset of that interface the reference's drivers use (size, resize, data().get(),
// push_back, assignment across spaces) so the library does not depend on a vendor template library.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include <sparsify.me/containers/memory.hxx>

namespace sparsifyme {

template <typename T>
using host_vector = std::vector<T>;

// Thin pointer wrapper so that `v.data().get()` (the thrust idiom the drivers use) works.
template <typename T>
struct device_ptr {
  T* p = nullptr;
  T* get() const { return p; }
  operator T*() const { return p; }
};

template <typename T>
class device_vector {
  static_assert(std::is_trivially_copyable<T>::value, "device_vector holds trivially copyable types");

 public:
  using value_type = T;
  device_vector() = default;
  explicit device_vector(std::size_t n) { resize(n); }
  device_vector(const host_vector<T>& h) { *this = h; }
  device_vector(const device_vector& o) { *this = o; }
  device_vector(device_vector&& o) noexcept : ptr_(o.ptr_), size_(o.size_), cap_(o.cap_) { o.ptr_ = nullptr; o.size_ = o.cap_ = 0; }
  ~device_vector() { release(); }

  device_vector& operator=(const host_vector<T>& h) {
    resize_uninitialised(h.size());
    if (size_) check(hipMemcpy(ptr_, h.data(), size_ * sizeof(T), hipMemcpyHostToDevice));
    return *this;
  }
  device_vector& operator=(const device_vector& o) {
    if (this == &o) return *this;
    resize_uninitialised(o.size_);
    if (size_) check(hipMemcpy(ptr_, o.ptr_, size_ * sizeof(T), hipMemcpyDeviceToDevice));
    return *this;
  }
  device_vector& operator=(device_vector&& o) noexcept {
    if (this != &o) { release(); ptr_ = o.ptr_; size_ = o.size_; cap_ = o.cap_; o.ptr_ = nullptr; o.size_ = o.cap_ = 0; }
    return *this;
  }

  // new elements are zero-filled, as thrust value-initialises them
  void resize(std::size_t n) {
    const std::size_t old = size_;
    resize_uninitialised(n, /*keep=*/true);
    if (n > old) check(hipMemset(ptr_ + old, 0, (n - old) * sizeof(T)));
  }
  void push_back(const T& v) {
    resize_uninitialised(size_ + 1, /*keep=*/true);
    check(hipMemcpy(ptr_ + size_ - 1, &v, sizeof(T), hipMemcpyHostToDevice));
  }
  std::size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  device_ptr<T> data() const { return device_ptr<T>{ptr_}; }

  host_vector<T> to_host() const {
    host_vector<T> h(size_);
    if (size_) check(hipMemcpy(h.data(), ptr_, size_ * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }

 private:
  static void check(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e));
  }
  void release() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    size_ = cap_ = 0;
  }
  void resize_uninitialised(std::size_t n, bool keep = false) {
    if (n > cap_) {
      const std::size_t ncap = keep && cap_ ? (n > 2 * cap_ ? n : 2 * cap_) : n;
      T* np = nullptr;
      check(hipMalloc(reinterpret_cast<void**>(&np), ncap * sizeof(T)));
      if (keep && size_) check(hipMemcpy(np, ptr_, size_ * sizeof(T), hipMemcpyDeviceToDevice));
      if (ptr_) (void)hipFree(ptr_);
      ptr_ = np;
      cap_ = ncap;
    }
    size_ = n;
  }
  T* ptr_ = nullptr;
  std::size_t size_ = 0, cap_ = 0;
};

// host <- device assignment helper (thrust allows `host_vector = device_vector`)
template <typename T>
inline void assign(host_vector<T>& dst, const device_vector<T>& src) { dst = src.to_host(); }
template <typename T>
inline void assign(device_vector<T>& dst, const host_vector<T>& src) { dst = src; }
template <typename T>
inline void assign(device_vector<T>& dst, const device_vector<T>& src) { dst = src; }
template <typename T>
inline void assign(host_vector<T>& dst, const host_vector<T>& src) { dst = src; }

template <typename type_t, memory_space_t space>
using vector_t = std::conditional_t<space == memory_space_t::host, host_vector<type_t>, device_vector<type_t>>;

}  // namespace sparsifyme
