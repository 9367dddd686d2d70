// ssrlcv_amd/host/Memory.hpp -- ptr::{base,device,host,value} and the error macros of include/Memory.cuh, on HIP.
// Same public surface and ownership rules (std::shared_ptr with cudaMalloc / cudaMallocHost / new[] deleters,
// include/Memory.cuh:96-285); device memory comes from the C ABI (ssrlcv_hip_malloc / _free).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <type_traits>
#include "Logger.hpp"
#include "ssrlcv_hip.h"

namespace ssrlcv {

// CudaSafeCall / CudaCheckError (include/Memory.cuh:33-74): log and exit(-1) on any device-API failure.
inline void __hipSafeCall(int status, const char* file, const int line) {
  if (status != 0) {
    logger.err.printf("hipSafeCall() failed at %s:%i : %s", file, line, ssrlcv_hip_status_string(status));
    std::exit(-1);
  }
}
#define HipSafeCall(err) ::ssrlcv::__hipSafeCall((err), __FILE__, __LINE__)
#define HipCheckError() ::ssrlcv::__hipSafeCall(ssrlcv_hip_device_synchronize(), __FILE__, __LINE__)
// source compatibility with reference call sites
#define CudaSafeCall(err) HipSafeCall(err)
#define CudaCheckError() HipCheckError()

template <typename T> struct device_delete { void operator()(T* p) const { HipSafeCall(ssrlcv_hip_free(p)); } };
template <typename T> struct host_pinned_delete { void operator()(T* p) const { HipSafeCall(ssrlcv_hip_host_free(p)); } };
template <typename T> struct host_unpinned_delete { void operator()(T* p) const { delete[] p; } };

namespace ptr {

template <typename T>
class base {
 protected:
  std::shared_ptr<T> ptr;

 public:
  base() noexcept : ptr(nullptr) {}
  base(std::nullptr_t) noexcept : ptr(nullptr) {}
  base(const base& r) : ptr(r.ptr) {}
  base<T>& operator=(const base& r) { this->ptr = r.ptr; return *this; }
  T* get() const noexcept { return ptr.get(); }
  T& operator*() const noexcept { return ptr.operator*(); }
  T* operator->() const noexcept { return ptr.operator->(); }
  explicit operator bool() const noexcept { return ptr.operator bool(); }
  operator std::shared_ptr<T>() const noexcept { return ptr; }
  void clear() { ptr.reset(); }
  virtual void set(long, bool = false) {}
  virtual ~base() {}
};

template <typename T>
class device : public base<T> {
 public:
  using base<T>::ptr;
  using base<T>::set;
  void set(long n, bool = false) {
    void* tmp = nullptr;
    HipSafeCall(ssrlcv_hip_malloc(&tmp, (size_t)n * sizeof(T)));
    ptr.reset((T*)tmp, device_delete<T>());
  }
  device(long n) { set(n); }
  device() {}
  device(const device& r) : base<T>() { this->ptr = r.ptr; }
  device(std::nullptr_t) noexcept {}
  device& operator=(const device& r) { this->ptr = r.ptr; return *this; }
  // NOTE: upstream exposes operator[] on device pointers too (host dereference of device memory); not provided.
};

template <typename T>
class host : public base<T> {
 public:
  using base<T>::ptr;
  using base<T>::set;
  void set(long n, bool pinned = false) {
    if (pinned) {
      void* tmp = nullptr;
      HipSafeCall(ssrlcv_hip_host_malloc(&tmp, (size_t)n * sizeof(T)));
      ptr.reset((T*)tmp, host_pinned_delete<T>());
    } else {
      ptr.reset(new T[n], host_unpinned_delete<T>());
    }
  }
  host() {}
  host(std::nullptr_t) noexcept {}
  host(const host& r) : base<T>() { this->ptr = r.ptr; }
  host(long n, bool pinned = false) { set(n, pinned); }
  host& operator=(const host& r) { this->ptr = r.ptr; return *this; }
  T& operator[](std::ptrdiff_t idx) const { return ptr.get()[idx]; }
};

template <typename T>
class value : public base<T> {
 public:
  using base<T>::ptr;
  value(std::nullptr_t) noexcept {}
  value() {}
  value(const value& r) : base<T>() { this->ptr = r.ptr; }
  value& operator=(const value& r) { this->ptr = r.ptr; return *this; }
  template <typename... Args> void construct(Args&&... args) { ptr = std::make_shared<T>(std::forward<Args>(args)...); }
  template <typename Arg, typename... Args,
            typename = typename std::enable_if<
                std::is_constructible<T, Arg, Args...>::value &&
                (sizeof...(Args) != 0 || !std::is_same<typename std::remove_cv<typename std::remove_reference<Arg>::type>::type, value<T>>::value)>::type>
  value(Arg&& arg, Args&&... args) { ptr = std::make_shared<T>(arg, std::forward<Args>(args)...); }
};

template <typename T, typename U> bool operator==(const base<T>& l, const base<U>& r) noexcept { return (const void*)l.get() == (const void*)r.get(); }
template <typename T, typename U> bool operator!=(const base<T>& l, const base<U>& r) noexcept { return (const void*)l.get() != (const void*)r.get(); }
template <typename T> bool operator==(const base<T>& l, std::nullptr_t) noexcept { return l.get() == nullptr; }
template <typename T> bool operator==(std::nullptr_t, const base<T>& r) noexcept { return nullptr == r.get(); }
template <typename T> bool operator!=(const base<T>& l, std::nullptr_t) noexcept { return l.get() != nullptr; }
template <typename T> bool operator!=(std::nullptr_t, const base<T>& r) noexcept { return nullptr != r.get(); }

}  // namespace ptr
}  // namespace ssrlcv
