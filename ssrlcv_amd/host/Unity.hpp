// ssrlcv_amd/host/Unity.hpp -- the Unity<T> host/device container of include/Unity.cuh on HIP (drop-in surface).
//
// Same public members (host, device), MemoryState machine {null,cpu,gpu,both} with `fore` tracking, constructors,
// transferMemoryTo (soft, ends `both`), setMemoryState (hard, frees the other side), setData (takes ownership of the
// caller's shared pointer, so aliasing through pixels->setData(other->device, ...) works as upstream relies on,
// src/FeatureFactory.cu:27,34,37), resize, clear, zeroOut, remove, sort, checkpoint / checkpoint constructor (same
// on-disk format, include/Unity.cuh:467-531,924-971) and the exception family (:77-133).
//
// Differences, all off the hot path: remove()/sort() take host function pointers and run on the host copy (upstream
// hands device function pointers to thrust); unified/pinned states beyond `pinned` host allocation are not supported
// (upstream: "only types null,cpu,gpu,both are supported right now", :39).
#pragma once
#include <type_traits>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <typeinfo>
#include "Memory.hpp"

namespace ssrlcv {

typedef enum MemoryState { null = 0, cpu = 1, gpu = 2, both = 3, unified = 4, nc = 10 } MemoryState;

inline std::string memoryStateToString(MemoryState state) {
  switch (state) {
    case null: return "null";
    case cpu: return "cpu";
    case gpu: return "gpu";
    case both: return "both";
    case unified: return "unified";
    case nc: return "no change (this should only be used to help with data manipulation methods)";
    default:
      logger.err << "ERROR: unknown MemoryState when calling memoryStateToString()";
      std::exit(-1);
  }
}

struct UnityException : std::exception {
  std::string msg;
  UnityException() : msg("Unknown Unity Exception") {}
  UnityException(std::string m) : msg("Unity Exception: " + m) {}
  virtual const char* what() const throw() { return msg.c_str(); }
};
struct IllegalUnityTransition : public UnityException {
  std::string msg;
  IllegalUnityTransition() : msg("Illegal Unity memory transfer") {}
  IllegalUnityTransition(std::string m) : msg("Illegal Unity memory transfer: " + m) {}
  virtual const char* what() const throw() { return msg.c_str(); }
};
struct NullUnityException : public UnityException {
  std::string msg;
  NullUnityException() : msg("Illegal attempt to use null set Unity") {}
  NullUnityException(std::string m) : msg("Illegal attempt to use null set Unity: " + m) {}
  virtual const char* what() const throw() { return msg.c_str(); }
};
struct CheckpointException : public UnityException {
  std::string msg;
  CheckpointException() : msg("Error in writing checkpoint") {}
  CheckpointException(std::string m) : msg("Checkpoint Error: " + m) {}
  virtual const char* what() const throw() { return msg.c_str(); }
};

template <class T>
class Unity {
 private:
  MemoryState state = null;
  MemoryState fore = null;
  bool pinned = false;
  unsigned long numElements = 0;

  size_t bytes() const { return numElements * sizeof(T); }
  void h2d() { HipSafeCall(ssrlcv_hip_memcpy(device.get(), host.get(), bytes(), 0)); }
  void d2h() { HipSafeCall(ssrlcv_hip_memcpy(host.get(), device.get(), bytes(), 1)); }

 public:
  typedef bool (*comp_ptr)(const T& a, const T& b);
  typedef bool (*pred_ptr)(const T& a);

  ptr::host<T> host;
  ptr::device<T> device;

  Unity() {}
  Unity(ptr::host<T> data, unsigned long n, MemoryState s, bool pin = false) { setData(data, n, s, pin); }
  Unity(ptr::device<T> data, unsigned long n, MemoryState s, bool pin = false) { setData(data, n, s, pin); }
  Unity(std::nullptr_t, unsigned long n, MemoryState s, bool pin = false) { setData(nullptr, n, s, pin); }

  // exact copy (include/Unity.cuh:421-440)
  Unity(ptr::value<Unity<T>> copy) {
    if (copy == nullptr) return;
    if (copy->getMemoryState() == null || copy->size() == 0) throw NullUnityException("cannot copy a null Unity<T>");
    setData(nullptr, copy->size(), copy->getMemoryState());
    fore = copy->getFore();
    state = copy->getMemoryState();
    if (state == cpu || state == both) std::memcpy(host.get(), copy->host.get(), bytes());
    if (state == gpu || state == both) HipSafeCall(ssrlcv_hip_memcpy(device.get(), copy->device.get(), bytes(), 2));
  }
  // copy_if (:441-465); the predicate runs on the host here
  Unity(ptr::value<Unity<T>> copy, pred_ptr predicate) {
    if (copy->getMemoryState() == null || copy->size() == 0) throw NullUnityException("cannot copy_if a null Unity<T>");
    MemoryState origin = copy->getMemoryState();
    ptr::host<T> src = copy->host;
    if (copy->getFore() == gpu || origin == gpu) {
      src = ptr::host<T>(copy->size());
      HipSafeCall(ssrlcv_hip_memcpy(src.get(), copy->device.get(), copy->size() * sizeof(T), 1));
    }
    unsigned long kept = 0;
    for (unsigned long i = 0; i < copy->size(); ++i) kept += predicate(src.get()[i]) ? 1 : 0;
    if (kept == 0) return;  // upstream resize(0) clears
    ptr::host<T> out(kept);
    unsigned long k = 0;
    for (unsigned long i = 0; i < copy->size(); ++i)
      if (predicate(src.get()[i])) out.get()[k++] = src.get()[i];
    setData(out, kept, cpu);
    if (origin != cpu) setMemoryState(origin);
  }
  // checkpoint reader (:466-531)
  Unity(std::string path, bool pin = false) {
    std::ifstream cp(path.c_str(), std::ifstream::binary);
    if (!cp.is_open()) throw CheckpointException("cannot open for read: " + path);
    const std::type_info& t_info = typeid(T);
    std::string name_str;
    getline(cp, name_str);
    if (name_str != std::string(t_info.name()))
      throw CheckpointException("names of type T do not match up in Unity checkpoint reader");
    size_t hash_code;
    char eol;
    cp.read((char*)&hash_code, sizeof(size_t));
    cp.read(&eol, 1);
    if (hash_code != t_info.hash_code())
      throw CheckpointException("hash_codes of type T do not match up in Unity checkpoint reader");
    MemoryState origin;
    cp.read((char*)&origin, sizeof(MemoryState));
    cp.read((char*)&numElements, sizeof(unsigned long));
    cp.read(&eol, 1);
    if (origin == null) throw CheckpointException("read origin in Unity checkpoint header shows null");
    pinned = pin;
    host.set(numElements, pin);
    state = cpu;
    fore = cpu;
    cp.read((char*)host.get(), (std::streamsize)bytes());
    bool ok = cp.good();
    cp.close();
    if (!ok) throw CheckpointException("could not successfully read checkpoint " + path);
    if (state != origin) setMemoryState(origin);
    logger.info << "Unity created from checkpoint " + path;
  }
  ~Unity() {
    if (state != null) clear();
  }

  unsigned long size() { return numElements; }
  MemoryState getMemoryState() { return state; }
  MemoryState getFore() { return fore; }
  bool isPinned() { return pinned; }

  // :540-579
  void resize(unsigned long resizeLength) {
    if (state == null) throw NullUnityException("cannot resize and empty Unity");
    if (resizeLength == 0U) { clear(); return; }
    if (state > 3) throw UnityException("please implement resize for newly supported MemoryState = " + memoryStateToString(state));
    unsigned long toCopy = std::min(resizeLength, numElements);
    if (state == cpu || state == both) {
      ptr::host<T> repl;
      repl.set(resizeLength, pinned);
      std::memcpy(repl.get(), host.get(), toCopy * sizeof(T));
      host = repl;
    }
    if (state == gpu || state == both) {
      ptr::device<T> repl(resizeLength);
      if (resizeLength > numElements) {  // the tail is filled from a default-constructed host array upstream
        ptr::host<T> fill(resizeLength);
        HipSafeCall(ssrlcv_hip_memcpy(repl.get(), fill.get(), resizeLength * sizeof(T), 0));
      }
      HipSafeCall(ssrlcv_hip_memcpy(repl.get(), device.get(), toCopy * sizeof(T), 2));
      device = repl;
    }
    numElements = resizeLength;
  }
  // :580-608
  void clear(MemoryState s = both) {
    if (s == null) { logger.warn << "WARNING: Unity<T>::clear(ssrlcv::null) does nothing"; return; }
    if (s != both && state != both && state != s) {
      logger.warn << "WARNING: Attempt to clear null memory in location " << memoryStateToString(s) << "...action prevented";
      return;
    }
    if (state == null) { logger.warn << "WARNING: Attempt to clear null (empty) Unity...action prevented"; return; }
    if (state > 3) throw IllegalUnityTransition("unknown memory state in clear() (supported states = both, cpu & gpu)");
    if (s == cpu || (s == both && host != nullptr)) host.clear();
    if (s == gpu || (s == both && device != nullptr)) device.clear();
    fore = (s == both) ? null : (s == cpu) ? gpu : cpu;
    state = fore;
  }
  // :609-641: elements become T() (a value-initialised array), NOT all-zero bytes
  void zeroOut(MemoryState s = both) {
    if (s == null) throw NullUnityException("cannot zero out an empty unity with state null");
    if (s > 3) throw IllegalUnityTransition("unknown memory state in zeroOut() (supported states = both, cpu & gpu)");
    if (s != both && state != both && s != state)
      throw IllegalUnityTransition(std::string("cannot zero out ") +
                                   ((state == cpu) ? "device because state == cpu" : "this->host because state == gpu"));
    if (s == cpu || (s == both && host != nullptr)) {
      if (!pinned) {
        host = ptr::host<T>();
        host.set(numElements);
        for (unsigned long i = 0; i < numElements; ++i) host.get()[i] = T();
      } else {
        for (unsigned long i = 0; i < numElements; ++i) host.get()[i] = T();
      }
    }
    if (s == gpu || (s == both && device != nullptr)) {
      if (s == both && host != nullptr) {
        h2d();
      } else if (std::is_trivially_default_constructible<T>::value && std::is_trivially_copyable<T>::value && !std::is_member_pointer<T>::value) {
        // value-initialised PODs are zero bytes: the same device content without building the array on the host and
        // copying it over (6 MB per image pair for a uint2_pair list of 4 x 10^5 queries, every matcher call).  (T() of a
        // trivially default-constructible type is zero-initialisation; the one scalar whose zero value is not all-zero bytes,
        // a pointer to member, is excluded -- a struct holding one takes the general branch's semantics only by accident,
        // and no type on this path does.)
        HipSafeCall(ssrlcv_hip_memset(device.get(), 0, bytes()));
      } else {
        T* z = new T[numElements]();
        HipSafeCall(ssrlcv_hip_memcpy(device.get(), z, bytes(), 0));
        delete[] z;
      }
    }
    fore = s;
  }
  // :646-665
  void setMemoryState(MemoryState s) {
    if (s == state) {
      logger.warn << "WARNING: hard setting of memory state to same memory state does nothing: " << memoryStateToString(state);
      return;
    }
    if (state == null) throw NullUnityException("Cannot setMemoryState of a null Unity");
    if (s == null) { clear(); return; }
    if (s == both) {
      if (fore == cpu) transferMemoryTo(gpu);
      else if (fore == gpu) transferMemoryTo(cpu);
      return;
    }
    if (fore != s) transferMemoryTo(s);
    if (s == cpu) clear(gpu);
    else if (s == gpu) clear(cpu);
  }
  void pin() {
    if (pinned) { logger.warn << "WARNING: attempt to pin already pinned Unity<T> does nothing"; return; }
    if (state != gpu) {
      ptr::host<T> p(numElements, true);
      std::memcpy(p.get(), host.get(), bytes());
      host = p;
    }
    pinned = true;
  }
  void unpin() {
    if (!pinned) { logger.warn << "WARNING: attempt to unpin nonpinned Unity<T> does nothing"; return; }
    if (state != gpu) {
      ptr::host<T> p(numElements);
      std::memcpy(p.get(), host.get(), bytes());
      host = p;
    }
    pinned = false;
  }
  // :702-790
  void setData(ptr::host<T> data, unsigned long n, MemoryState s, bool pin = false) {
    if (s == null) throw NullUnityException("cannot use null as state of T* data in Unity<T>::setData");
    if (s == gpu) throw IllegalUnityTransition("cannot fill Unity gpu with host data");
    if (n == 0) throw IllegalUnityTransition("cannot fill Unity with T* data, numElements = 0");
    if (data != nullptr && ((void*)data.get() == (void*)host.get() || (void*)data.get() == (void*)device.get()))
      throw UnityException("cannot use Unity<T>::setData where T* data is this->host or this->device");
    if (s != cpu && s != both)
      throw IllegalUnityTransition("currently no support for Unity<T>::setData with T* data at MemoryState = " + memoryStateToString(s));
    if (state != null) clear();
    numElements = n;
    state = cpu;
    fore = cpu;
    pinned = pin;
    host = data;
    if (s == both) setMemoryState(both);
  }
  void setData(ptr::device<T> data, unsigned long n, MemoryState s, bool pin = false) {
    if (s == null) throw NullUnityException("cannot use null as state of T* data in Unity<T>::setData");
    if (s == cpu) throw IllegalUnityTransition("cannot fill Unity cpu with device data");
    if (n == 0) throw IllegalUnityTransition("cannot fill Unity with T* data, numElements = 0");
    if (data != nullptr && ((void*)data.get() == (void*)host.get() || (void*)data.get() == (void*)device.get()))
      throw UnityException("cannot use Unity<T>::setData where T* data is this->host or this->device");
    if (s != gpu && s != both)
      throw IllegalUnityTransition("currently no support for Unity<T>::setData with T* data at MemoryState = " + memoryStateToString(s));
    if (state != null) clear();
    numElements = n;
    state = gpu;
    fore = gpu;
    pinned = pin;
    device = data;
    if (s == both) setMemoryState(both);
  }
  void setData(std::nullptr_t, unsigned long n, MemoryState s, bool pin = false) {
    if (s == null) throw NullUnityException("cannot use null as state of T* data in Unity<T>::setData");
    if (n == 0) throw IllegalUnityTransition("cannot fill Unity with T* data, numElements = 0");
    if (s > 3) throw IllegalUnityTransition("attempt to instantiate unkown MemoryState fron nullptr (supported states = both, cpu & gpu)");
    if (state != null) clear();
    numElements = n;
    pinned = pin;
    if (s == gpu || s == both) {
      // the device buffer starts as a value-initialised T[] (:778-782 -> zeroOut(gpu)): Feature() gives parent = -1
      device.set(n);
      state = gpu;
      fore = gpu;
      zeroOut(gpu);
      if (s == both) setMemoryState(both);
    } else {
      state = cpu;
      fore = cpu;
      host.set(n, pin);  // plain new T[n]: PODs stay uninitialised, like upstream
      if (pin) zeroOut(cpu);
    }
  }
  // :796-818
  void setFore(MemoryState s) {
    if (state == null) throw NullUnityException("attempt to Unity<T>::setFore(MemoryState state) when this->state == null");
    if (fore == s) { logger.warn << "WARNING: Unity<T>::setFore(MemoryState state) when state == this->fore does nothing"; return; }
    if (s == both) {
      logger.warn << "ERROR: cannot set fore to both manually:" << "\tuse setMemoryState(both) or transferMemoryTo((this->fore == gpu) ? cpu : gpu)";
      std::exit(-1);
    }
    if (state != both && state != s) {
      if (state == cpu) throw IllegalUnityTransition("attempt to Unity<T>::setFore(MemoryState state) to gpu when this->device == nullptr");
      throw IllegalUnityTransition("attempt to Unity<T>::setFore(MemoryState state) to cpu when this->host == nullptr");
    }
    fore = s;
  }
  // :819-854
  void transferMemoryTo(MemoryState s) {
    if (state == null || bytes() == 0) throw NullUnityException("thrown in Unity<T>::transferMemoryTo()");
    if (s == null) throw IllegalUnityTransition("Cannot transfer unity memory to null");
    if (s > 3) throw IllegalUnityTransition("unsupported memory destination in Unity<T>::transferMemoryTo (supported states = both, cpu & gpu)");
    if (fore == s) {
      logger.warn << "WARNING: transfering memory to location of fore does nothing: " << memoryStateToString(s);
      return;
    }
    if (state != both) {
      if (state == cpu && device == nullptr) device.set(numElements);
      else if (state == gpu && host == nullptr) host.setForOverwrite(numElements, pinned);  // every element is overwritten by d2h() below
      state = both;
    }
    if (fore == cpu) h2d();
    else if (fore == gpu) d2h();
    fore = both;
  }
  // :855-881 (predicate evaluated on the host copy)
  void remove(pred_ptr predicate, MemoryState destination = nc) {
    if (state == null || numElements == 0) throw NullUnityException("cannot remove anything from an already null Unity<T>");
    if (destination == nc) destination = state;
    if (fore == gpu || state == gpu) transferMemoryTo(cpu);
    T* h = host.get();
    unsigned long kept = std::remove_if(h, h + numElements, predicate) - h;
    fore = cpu;
    if (state == both) fore = cpu;
    if (kept == 0) {
      logger.warn << "Unity<T>::remove(bool(*validate)(const T&)) led to all elements being removed (data cleared)";
      clear();
      return;
    }
    if (state == both) { clear(gpu); }
    if (kept != numElements) resize(kept);
    if (destination != state) setMemoryState(destination);
  }
  // :882-922
  void sort(bool greater = false, MemoryState destination = nc) {
    if (greater) sort_impl([](const T& a, const T& b) { return b < a; }, destination);
    else sort_impl([](const T& a, const T& b) { return a < b; }, destination);
  }
  void sort(comp_ptr comparator, MemoryState destination = nc) { sort_impl(comparator, destination); }
  // :923-971
  void checkpoint(int id, std::string dirPath = "./") {
    if (state == null) throw NullUnityException("cannot write a checkpoint with a null Unity<T>");
    const std::type_info& ti = typeid(T);
    size_t hash_code = ti.hash_code();
    const char* name = ti.name();
    char eol = '\n';
    std::string pathToFile = dirPath + std::to_string(id) + "_" + name + ".uty";
    std::ofstream cp(pathToFile.c_str(), std::ofstream::binary);
    MemoryState origin = state;
    if (fore == gpu) transferMemoryTo(cpu);
    if (!cp.is_open()) throw CheckpointException("could not open for writing: " + pathToFile);
    cp.write(name, std::strlen(name));
    cp.write(&eol, sizeof(char));
    cp.write((char*)&hash_code, sizeof(size_t));
    cp.write(&eol, sizeof(char));
    cp.write((char*)&origin, sizeof(MemoryState));
    cp.write((char*)&numElements, sizeof(unsigned long));
    cp.write(&eol, sizeof(char));
    cp.write((char*)host.get(), (std::streamsize)bytes());
    cp.close();
    if (!cp.good()) throw CheckpointException("could not write Unity<T> checkpoint: " + pathToFile);
    logger.info << "checkpoint " + pathToFile + " successfully written";
    if (state != origin) setMemoryState(origin);
  }
  void printInfo() {
    std::cout << "numElements = " << numElements << " state = " << memoryStateToString(state);
    if (pinned && (state == cpu || state == both)) std::cout << " (pinned)";
    std::cout << " fore = " << memoryStateToString(fore) << " type = " << typeid(T).name() << "\n";
  }

 private:
  template <typename Cmp>
  void sort_impl(Cmp cmp, MemoryState destination) {
    if (state == null || numElements == 0) throw NullUnityException("cannot sort a null Unity<T>");
    MemoryState origin = (destination == nc) ? state : destination;
    if (fore == gpu || state == gpu) transferMemoryTo(cpu);
    std::stable_sort(host.get(), host.get() + numElements, cmp);
    if (state == both) { fore = cpu; clear(gpu); }
    if (origin != state) setMemoryState(origin);
  }
};

}  // namespace ssrlcv
