// ssrlcv_amd/host/Memory.hpp -- ptr::{base,device,host,value} and the error macros of include/Memory.cuh, on HIP.
// Same public surface and ownership rules (std::shared_ptr with cudaMalloc / cudaMallocHost / new[] deleters,
// include/Memory.cuh:96-285); device memory comes from the C ABI (ssrlcv_hip_malloc / _free).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <type_traits>
#include <vector>
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

// ABI check (round 6): the mirror was compiled against SSRLCV_HIP_ABI_VERSION of include/ssrlcv_hip.h; a library built from
// another revision of that header is refused before the first call that touches the device (every allocation of the
// mirror goes through detail::blockCache(), whose constructor calls this).  Same convention as a failed device call:
// logger.err + exit(-1).
inline void requireAbi() {
  static const bool ok = [] {
    const int got = ssrlcv_hip_abi_version();
    if (got != SSRLCV_HIP_ABI_VERSION) {
      logger.err.printf("libssrlcv_hip reports ABI version %d (%s), these headers are version %d: rebuild one of them", got,
                        ssrlcv_hip_version(), SSRLCV_HIP_ABI_VERSION);
      std::exit(-1);
    }
    return true;
  }();
  (void)ok;
}

// Size-class cache of device and pinned-host blocks.  The reference's Unity<T> allocates and frees on every memory-state
// change (setMemoryState is "hard": it frees the side it leaves), which its factories do around every call -- image
// up, image back to its origin state, exact-size result lists.  cudaMalloc / cudaMallocHost of 16..50 MB cost
// 0.1..3 ms each (pinning pages is the slow one): 3.9 of the 10.2 ms a generateFeatures call took on a 4096^2 image
// whose pixels start on the host.  Freed blocks are kept per size class (multiples of 256 B up to 1 MiB, of 1 MiB
// above) up to a byte budget and handed out again; contents are never assumed.  SSRLCV_NO_BLOCK_CACHE=1 turns it off.
namespace detail {
class BlockCache {
  std::mutex mtx;
  std::map<size_t, std::vector<void*>> freeList[3];  // [0] device, [1] pinned host, [2] pageable host (round 5)
  size_t cached[3] = {0, 0, 0};
  const size_t budget[3] = {(size_t)24 << 30, (size_t)2 << 30, (size_t)1 << 30};  // (pageable blocks: 1 GB; trimBlockCache() releases them)
  bool enabled;

 public:
  BlockCache() : enabled(std::getenv("SSRLCV_NO_BLOCK_CACHE") == nullptr) { requireAbi(); }
  ~BlockCache() {  // process exit: the runtime may already be gone, so return codes are not checked
    for (int k = 0; k < 3; ++k)
      for (auto& e : freeList[k])
        for (void* p : e.second) (void)release(k, p);
  }
  static int acquire(int kind, void** p, size_t bytes) {
    if (kind == 0) return ssrlcv_hip_malloc(p, bytes);
    if (kind == 1) return ssrlcv_hip_host_malloc(p, bytes);
    *p = std::malloc(bytes);  // pageable: what `new T[n]` rests on (16-byte aligned, like operator new[])
    return *p ? 0 : 2;        // 2 = hipErrorOutOfMemory
  }
  static int release(int kind, void* p) {
    if (kind == 0) return ssrlcv_hip_free(p);
    if (kind == 1) return ssrlcv_hip_host_free(p);
    std::free(p);
    return 0;
  }
  static size_t classOf(size_t bytes) {
    const size_t g = bytes <= ((size_t)1 << 20) ? 256 : ((size_t)1 << 20);
    return (bytes + g - 1) / g * g;
  }
  void* take(int kind, size_t bytes, size_t& cls) {
    cls = classOf(bytes ? bytes : 1);
    if (enabled) {
      std::lock_guard<std::mutex> lock(mtx);
      auto it = freeList[kind].find(cls);
      if (it != freeList[kind].end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        cached[kind] -= cls;
        return p;
      }
    }
    void* p = nullptr;
    int rc = acquire(kind, &p, cls);
    if (rc != 0 && enabled) {  // out of memory with blocks parked in the cache: release them and try once more
      trim(kind);
      rc = acquire(kind, &p, cls);
    }
    __hipSafeCall(rc, __FILE__, __LINE__);
    return p;
  }
  void give(int kind, void* p, size_t cls) {
    if (!p) return;
    if (enabled) {
      std::lock_guard<std::mutex> lock(mtx);
      if (cached[kind] + cls <= budget[kind]) {
        freeList[kind][cls].push_back(p);
        cached[kind] += cls;
        return;
      }
    }
    __hipSafeCall(release(kind, p), __FILE__, __LINE__);
  }
  void trim(int kind) {
    std::lock_guard<std::mutex> lock(mtx);
    for (auto& e : freeList[kind])
      for (void* p : e.second) (void)release(kind, p);
    freeList[kind].clear();
    cached[kind] = 0;
  }
};
inline BlockCache& blockCache() {
  static BlockCache c;
  return c;
}
}  // namespace detail

// Returns every parked block of the cache to the system: kind 0 device, 1 pinned host, 2 pageable host, -1 all of them.
// (The cache otherwise gives blocks back only when an allocation fails.)
inline void trimBlockCache(int kind = -1) {
  for (int k = 0; k < 3; ++k)
    if (kind < 0 || kind == k) detail::blockCache().trim(k);
}

template <typename T> struct device_delete {
  size_t cls;
  void operator()(T* p) const { detail::blockCache().give(0, p, cls); }
};
template <typename T> struct host_pinned_delete {
  size_t cls;
  void operator()(T* p) const { detail::blockCache().give(1, p, cls); }
};
// Pageable host arrays: `new T[n]` upstream.  Blocks of 1 MiB and more come from the cache too (a fresh 45 MB array is
// mmap'ed, page-faulted by its first writer and unmapped again on every generateFeatures + transferMemoryTo(cpu): 5 of the
// 7.3 ms that leg took in round 4); the elements are default-constructed in place unless the caller is about to overwrite
// all of them (construct = false: transferMemoryTo's device-to-host copy), and destroyed when T needs it.
template <typename T> struct host_unpinned_delete {
  size_t n, cls;  // cls == 0: a plain new[] array
  void operator()(T* p) const {
    if (cls == 0) { delete[] p; return; }
    if (!std::is_trivially_destructible<T>::value)
      for (size_t i = n; i-- > 0;) p[i].~T();
    detail::blockCache().give(2, p, cls);
  }
};

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
    size_t cls = 0;
    void* tmp = detail::blockCache().take(0, (size_t)n * sizeof(T), cls);
    ptr.reset((T*)tmp, device_delete<T>{cls});
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
  void set(long n, bool pinned = false) { alloc(n, pinned, true); }
  // the same allocation for a caller that overwrites every element next (Unity<T>::transferMemoryTo's device-to-host copy)
  void setForOverwrite(long n, bool pinned = false) { alloc(n, pinned, false); }

 private:
  // The cached pageable blocks (and with them the "do not construct what is about to be overwritten" path) only serve
  // element types for which raw storage IS a valid array: trivially copyable (a device-to-host memcpy makes the objects)
  // and trivially destructible (the deleter has nothing to run over elements nobody constructed).  Everything on the hot
  // path is (float, KeyPoint, MultiMatch, Feature<SIFT_Descriptor>, float3 ...); anything else takes plain new[] / delete[].
  static constexpr bool kRawStorageOk = std::is_trivially_copyable<T>::value && std::is_trivially_destructible<T>::value;
  void alloc(long n, bool pinned, bool construct) {
    if (pinned) {
      size_t cls = 0;
      void* tmp = detail::blockCache().take(1, (size_t)n * sizeof(T), cls);
      ptr.reset((T*)tmp, host_pinned_delete<T>{cls});
    } else if (kRawStorageOk && (size_t)n * sizeof(T) >= ((size_t)1 << 20) && alignof(T) <= 16) {
      size_t cls = 0;
      T* tmp = (T*)detail::blockCache().take(2, (size_t)n * sizeof(T), cls);
      if (construct && !std::is_trivially_default_constructible<T>::value)
        for (long i = 0; i < n; ++i) ::new ((void*)(tmp + i)) T;
      ptr.reset(tmp, host_unpinned_delete<T>{(size_t)n, cls});
    } else {
      ptr.reset(new T[n], host_unpinned_delete<T>{(size_t)n, 0});
    }
  }

 public:
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
