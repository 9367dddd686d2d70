// ssrlcv_amd/host/Image.hpp -- ssrlcv::Image with its Camera / PushbroomCamera (include/Image.cuh:35-130).
// Member order and types follow upstream so that sizeof(Image) == 240 and the raw `.cpimg` checkpoint
// (Image::checkpoint = memcpy of the object, src/Image.cu:274-303) can be read: only the POD prefix is taken from the
// file; filePath and the pixels pointer are re-initialised (they are process-local in the dump).
// File decoding (png/jpg/tiff), params.csv parsing and the fundamental-matrix helpers are out of scope.
#pragma once
#include <cstring>
#include <fstream>
#include <string>
#include "Unity.hpp"
#include "cuda_vec_types.hpp"

namespace ssrlcv {

class Image {
 public:
  struct Camera {
    float3 cam_pos;
    float3 cam_rot;
    float2 fov;
    float foc;
    float2 dpix;
    long long int timeStamp;
    float3 ecef_offset;
    bool no_rot = false;
    uint2 size;
    Camera() : cam_pos{0, 0, 0}, cam_rot{0, 0, 0}, fov{0, 0}, foc(0), dpix{0, 0}, timeStamp(0), ecef_offset{0, 0, 0}, size{0, 0} {}
    Camera(uint2 size) : Camera() { this->size = size; }
    Camera(uint2 size, float3 cam_pos, float3 cam_rot) : Camera() {
      this->size = size;
      this->cam_pos = cam_pos;
      this->cam_rot = cam_rot;
    }
  };
  struct PushbroomCamera {
    float3 start_pos;
    float3 end_pos;
    float2 projection_center;
    float axis_radius;
    float roll;
    float altitude;
    float foc;
    float fov;
    float gsd;
    float2 dpix;
    uint2 size;
  };

  std::string filePath;
  int id;
  uint2 size;
  unsigned int colorDepth;
  Camera camera;
  PushbroomCamera pushbroom;
  bool isPushbroom;
  ptr::value<Unity<unsigned char>> pixels;

  Image() : id(-1), size{0, 0}, colorDepth(0), pushbroom(), isPushbroom(false) {}
  Image(uint2 size, unsigned int colorDepth, ptr::value<Unity<unsigned char>> pixels)
      : id(-1), size(size), colorDepth(colorDepth), pushbroom(), isPushbroom(false), pixels(pixels) {
    camera.size = size;
  }
  // `.cpimg` checkpoint constructor (src/Image.cu:48-64 with ignorePixels): POD members only
  Image(std::string path, int id = -1, bool /*ignorePixels*/ = true) : Image() {
    std::ifstream in(path.c_str(), std::ifstream::binary);
    char raw[240];
    in.read(raw, sizeof raw);
    if (!in.good()) throw CheckpointException("could not read image checkpoint " + path);
    std::memcpy(&this->id, raw + 32, sizeof(int));
    std::memcpy(&this->size, raw + 40, sizeof(uint2));
    std::memcpy(&this->colorDepth, raw + 48, sizeof(unsigned int));
    std::memcpy(&this->camera, raw + 56, sizeof(Camera));
    std::memcpy(&this->pushbroom, raw + 136, sizeof(PushbroomCamera));
    std::memcpy(&this->isPushbroom, raw + 208, sizeof(bool));
    if (id != -1) this->id = id;
    filePath = path;
  }
  // Image::setFloatVector / getFloatVector (src/Image.cu:400-472): {pos xyz, rot xyz, fov xy, foc, dpix xy}
  void setFloatVector(ptr::value<Unity<float>> params) {
    float* p = params->host.get();
    switch (params->size()) {
      case 11: camera.dpix.y = p[10];  // fallthrough
      case 10: camera.dpix.x = p[9];   // fallthrough
      case 9: camera.foc = p[8];       // fallthrough
      case 8: camera.fov.y = p[7];     // fallthrough
      case 7: camera.fov.x = p[6];     // fallthrough
      case 6: camera.cam_rot.z = p[5]; // fallthrough
      case 5: camera.cam_rot.y = p[4]; // fallthrough
      case 4: camera.cam_rot.x = p[3]; // fallthrough
      case 3: camera.cam_pos.z = p[2]; // fallthrough
      case 2: camera.cam_pos.y = p[1]; // fallthrough
      case 1: camera.cam_pos.x = p[0]; // fallthrough
      default: break;
    }
  }
  ptr::value<Unity<float>> getFloatVector(int len) {
    ptr::value<Unity<float>> out(nullptr, (unsigned long)len, cpu);
    float all[11] = {camera.cam_pos.x, camera.cam_pos.y, camera.cam_pos.z, camera.cam_rot.x, camera.cam_rot.y,
                     camera.cam_rot.z, camera.fov.x, camera.fov.y, camera.foc, camera.dpix.x, camera.dpix.y};
    for (int i = 0; i < len && i < 11; ++i) out->host.get()[i] = all[i];
    return out;
  }
};

// convertToBW (src/Image.cu:665-690): pixels end up single-channel, in their origin memory state
inline void convertToBW(ptr::value<Unity<unsigned char>> pixels, unsigned int colorDepth) {
  if (colorDepth == 1) {
    logger.warn << "Pixels are already bw";
    return;
  }
  if (colorDepth < 2 || colorDepth > 4) {
    logger.err.printf("ERROR colorDepth of %u is not supported", colorDepth);  // generateBW's default branch traps
    std::exit(-1);
  }
  MemoryState origin = pixels->getMemoryState();
  if (origin != gpu) pixels->setMemoryState(gpu);
  unsigned long numPixels = pixels->size() / colorDepth;
  ptr::device<unsigned char> bwPixels_device((long)numPixels);
  HipSafeCall(ssrlcv_hip_convert_to_bw(pixels->device.get(), colorDepth, bwPixels_device.get(), numPixels, nullptr));
  HipCheckError();
  pixels->setData(bwPixels_device, numPixels, gpu);
  if (origin != gpu) pixels->setMemoryState(origin);
}

static_assert(sizeof(Image::Camera) == sizeof(ssrlcv_camera), "Image::Camera must be 80 B");
static_assert(sizeof(Image::PushbroomCamera) == sizeof(ssrlcv_pushbroom), "Image::PushbroomCamera must be 72 B");
static_assert(sizeof(Image) == 240, "Image layout drifted from the reference's .cpimg dump");

}  // namespace ssrlcv
