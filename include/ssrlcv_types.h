/*
 * include/ssrlcv_types.h -- plain-old-data layouts shared by the C-ABI (ssrlcv_hip.h), the HIP kernels and the
 * C++ host mirror of the reference API.  Every struct is byte-compatible with the reference type it replaces, so a
 * Unity<T>::device pointer of the reference can be handed to the C-ABI unchanged and the reference's .uty
 * checkpoints can be read/written directly.
 *
 *   reference type                                   file:line                              here
 *   float2/float3/float4/uint2 (CUDA vector types)   cuda vector_types.h                    ssrlcv_float2 ...
 *   ssrlcv::Image::Camera            (80 B)          include/Image.cuh:40-57                ssrlcv_camera
 *   ssrlcv::Image::PushbroomCamera   (72 B)          include/Image.cuh:62-79                ssrlcv_pushbroom
 *   ssrlcv::Feature<SIFT_Descriptor> (152 B)         include/Feature.cuh:31-94              ssrlcv_sift_feature
 *   ScaleSpace::SSKeyPoint           (32 B)          include/FeatureFactory.cuh:64-75       ssrlcv_sskeypoint
 *   ssrlcv::KeyPoint (16 B), MultiMatch (8 B)        include/MatchFactory.cuh:32-46         ssrlcv_keypoint, ssrlcv_multimatch
 *   ssrlcv::Match (40 B), DMatch (48 B)              include/MatchFactory.cuh:52-65         ssrlcv_match, ssrlcv_dmatch
 *   ssrlcv::uint2_pair (16 B)                        include/MatchFactory.cuh:23-26         ssrlcv_uint2_pair
 *   ssrlcv::Bundle (12 B), Bundle::Line (24 B)       include/PointCloudFactory.cuh:25-40    ssrlcv_bundle, ssrlcv_line
 */
#ifndef SSRLCV_TYPES_H
#define SSRLCV_TYPES_H
#include <stdint.h>
#include <stddef.h>

#if defined(__cplusplus)
#define SSRLCV_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define SSRLCV_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct { float x, y; } __attribute__((aligned(8))) ssrlcv_float2;
typedef struct { float x, y, z; } ssrlcv_float3;
typedef struct { float x, y, z, w; } __attribute__((aligned(16))) ssrlcv_float4;
typedef struct { uint32_t x, y; } __attribute__((aligned(8))) ssrlcv_uint2;

typedef struct {
  ssrlcv_float3 cam_pos;
  ssrlcv_float3 cam_rot;
  ssrlcv_float2 fov;
  float foc;
  ssrlcv_float2 dpix;
  long long timeStamp;
  ssrlcv_float3 ecef_offset;
  uint8_t no_rot;
  ssrlcv_uint2 size;
} ssrlcv_camera;

typedef struct {
  ssrlcv_float3 start_pos;
  ssrlcv_float3 end_pos;
  ssrlcv_float2 projection_center;
  float axis_radius;
  float roll;
  float altitude;
  float foc;
  float fov;
  float gsd;
  ssrlcv_float2 dpix;
  ssrlcv_uint2 size;
} ssrlcv_pushbroom;

typedef struct {
  int parent;
  ssrlcv_float2 loc;
  float sigma;
  float theta;
  uint8_t values[128];
} ssrlcv_sift_feature;

typedef struct {
  int octave;
  int blur;
  ssrlcv_float2 loc;
  float intensity;
  float sigma;
  float theta;
  uint8_t discard;
} ssrlcv_sskeypoint;

typedef struct { int parentId; ssrlcv_float2 loc; } ssrlcv_keypoint;
typedef struct { uint32_t numKeyPoints; int index; } ssrlcv_multimatch;
typedef struct { uint8_t invalid; ssrlcv_keypoint keyPoints[2]; } ssrlcv_match;
typedef struct { uint8_t invalid; ssrlcv_keypoint keyPoints[2]; float distance; } ssrlcv_dmatch;
typedef struct { ssrlcv_uint2 a, b; } ssrlcv_uint2_pair;
typedef struct { float roll, pitch, yaw, x, y, z; } ssrlcv_pose;  /* ssrlcv::Pose (include/PoseEstimator.cuh:21-28) */
typedef struct { ssrlcv_float3 vec, pnt; } ssrlcv_line;
typedef struct { uint32_t numLines; int index; uint8_t invalid; } ssrlcv_bundle;

SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_camera) == 80, "Image::Camera layout");
SSRLCV_STATIC_ASSERT(offsetof(ssrlcv_camera, fov) == 24 && offsetof(ssrlcv_camera, foc) == 32 &&
                     offsetof(ssrlcv_camera, dpix) == 40 && offsetof(ssrlcv_camera, timeStamp) == 48 &&
                     offsetof(ssrlcv_camera, ecef_offset) == 56 && offsetof(ssrlcv_camera, no_rot) == 68 &&
                     offsetof(ssrlcv_camera, size) == 72, "Image::Camera offsets");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_pushbroom) == 72, "Image::PushbroomCamera layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_sift_feature) == 152 && offsetof(ssrlcv_sift_feature, loc) == 8 &&
                     offsetof(ssrlcv_sift_feature, values) == 24, "Feature<SIFT_Descriptor> layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_sskeypoint) == 32 && offsetof(ssrlcv_sskeypoint, discard) == 28, "SSKeyPoint layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_keypoint) == 16 && sizeof(ssrlcv_multimatch) == 8, "KeyPoint/MultiMatch layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_match) == 40 && sizeof(ssrlcv_dmatch) == 48 &&
                     offsetof(ssrlcv_dmatch, distance) == 40, "Match/DMatch layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_uint2_pair) == 16, "uint2_pair layout");
SSRLCV_STATIC_ASSERT(sizeof(ssrlcv_line) == 24 && sizeof(ssrlcv_bundle) == 12, "Bundle layout");

#endif
