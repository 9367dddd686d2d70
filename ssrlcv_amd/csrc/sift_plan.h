// ssrlcv_amd/csrc/sift_plan.h -- host-side description of the SIFT workspace (shared by pyramid.hip / keypoints.hip).
//
// HBM layout of one image's workspace (all offsets 256-byte aligned, sizes for a W x H u8 input):
//   octave o (0..3) works at (2W >> o) x (2H >> o); P_o pixels, sum P = 5.3125 W H
//   in0        f32  P_0        octave-0 input (2x bilinear upsample of the u8 image)
//   in1        f32  P_1        input of octaves 1..3 (2x2 bin of the previous octave's un-normalised level 3); reused
//   gauss[o][6] f32 6 sum P    the six gaussian levels of every octave (un-normalised, kept: they ARE the scale space the
//                              key-point stage reads)
//   (the DoG levels are NOT materialised: ssrlcv_hip_sift_build_dog's last pass per octave forms them in registers for
//    the extrema search and their min / max, every later consumer evaluates N(level b+1) - N(level b) at the pixels it
//    samples -- the same float operations, so the same values, without 20 bytes per pixel written and read back)
//   flags[o]   u8   sum P      extremum bits per pixel: bit k = extremum of DoG level k+1 (k = 0..2), bit 4+k = the same
//                              and past the first removeNoise (src/FeatureFactory.cu:484)
//   polar[o]   f32x2 3 sum P   gradient magnitude / direction of DoG levels 1..3 (shared by orientation + descriptors)
//   minmax     f32  4 x (6+5) x 2   per level {min,max} (gaussian, DoG)
//   key points: per octave two ping-pong SSKeyPoint lists (capacity cap_o), theta lists, counters, index tables
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>
#include "ssrlcv_hip.h"

namespace svp {

constexpr int kOctaves = 4;
constexpr int kGauss = 6;
constexpr int kDog = 5;
constexpr int kMaxTaps = 129;
constexpr int kMaxOrient = 4;
// thresholds of SIFT_FeatureFactory::generateFeatures (src/SIFT_FeatureFactory.cu:58-59); the first removeNoise runs at
// 0.8 x the noise threshold (src/FeatureFactory.cu:484)
constexpr float kNoiseThreshold = 0.01f;
constexpr float kEdgeThreshold = 12.1f;
constexpr int kNoiseFlagShift = 4;  // flags: bit k raw extremum of level k+1, bit kNoiseFlagShift + k = past the first removeNoise

// Device-resident per-octave key-point bookkeeping (mirrors Octave::extrema / extremaBlurIndices,
// include/FeatureFactory.cuh:107-123).
struct OctaveState {
  int idx[kDog];        // extremaBlurIndices
  int n;                // extrema->size(); 0 == nullptr
  int hasExtrema;       // extrema != nullptr
  int overflow;         // set when a list outgrew its capacity
  int stale[kDog];      // scratch for bookkeeping kernels
  int pad[3];
};

// Polar tables (float2 {|grad|, atan2} per pixel of DoG levels 1..3): the descriptor kernel indexes them the way the
// reference indexes its W*H gradient array, with the FLAT index round(y) * W + round(x) (src/SIFT_FeatureFactory.cu:507),
// and a rotated window may reach one column left / right of the level (the flat index then lands in the neighbouring
// row, as upstream) or one row below it and column -1 of row 0 (upstream: a read outside the array, undefined).  Each
// level's table therefore carries one zero entry in front and W + 1 behind: an outside read returns a zero gradient,
// which votes nothing -- the definition the oracle uses too.  Entry of flat index i of level l: l * stride + 1 + i.
__host__ __device__ inline size_t polar_level_stride(uint32_t w, uint32_t h) { return (size_t)w * h + w + 2; }

struct OctavePlan {
  uint32_t w, h;
  float pixelWidth;
  float sigma[kGauss];
  int taps[kGauss];
  float weights[kGauss][kMaxTaps];
  uint32_t cap;            // key-point list capacity
  size_t off_flags;
  size_t off_polar;        // float2 {|grad|, atan2} of the normalised DoG levels 1..3 (3 * polar_level_stride * 8 bytes)
  size_t off_kpA, off_kpB; // SSKeyPoint ping-pong lists
  size_t off_theta;        // cap * kMaxOrient floats
  size_t off_thetaCnt;     // cap uint32 (number of orientations per key point)
  size_t off_part;         // partition workspace (uint32 words)
  size_t off_featBase;     // uint32: first feature index of this octave
  size_t off_descConst;    // cap x 32 bytes: per-key-point constants of the descriptor kernel (k_desc_consts)
};

// Side streams and events of one plan.  build_dog runs the HBM-bound DoG kernel of octave o beside the FMA-bound
// convolutions of octave o+1 (on `table`); describe runs octave 0's key-point chain on the caller's stream, the three
// short chains of octaves 1-3 one after the other on `chain` and the polar tables on `table`.  Everything is forked
// from and joined back into the caller's stream, so the calls keep their stream-ordered semantics.  The two side
// streams belong to the device, not to the plan (every plan on a device shares them): with the caller's stream that
// makes three queues, below the four hardware queues the runtime multiplexes streams onto - with one stream per octave
// and plan the chains were measured waiting for each other behind a shared queue.  Starting octave 0's chain and the
// polar tables earlier, beside the convolutions of octaves 1-3, was measured too: the convolutions lose as much as the
// chain gains (every one of these kernels fills the chip on its own), so the order below is kept.
// One call per plan may be in flight at a time.
constexpr unsigned kDogMaxBlocks = 16384;               // grid bound of the streaming DoG kernel (256-thread blocks)
constexpr unsigned kDogMaxWaves = kDogMaxBlocks * 4;

constexpr int kSampleGroups = 4;  // describe: the (octave, blur segment) groups the orientation / descriptor kernels are pipelined over

struct PlanAsync {
  hipStream_t chain, table, chain2;  // chain2: the list chains of octaves 2-3 in describe (octave 1's runs on `chain`)
  hipStream_t polar;                 // round 5: the polar tables of a fused extract, started from inside build_dog
  hipEvent_t fork;
  hipEvent_t groupFork, groupExpanded[kSampleGroups], groupReady[kSampleGroups];  // describe: pipelined sampling groups
  hipEvent_t join[kOctaves + 1], convDone[kOctaves], dogDone[kOctaves], polarDone[kOctaves];
  hipEvent_t binDone[kOctaves];              // build_dog: level 3 of octave o and its 2x2 bin are complete
  hipEvent_t expandFork, expandJoin[2];      // describe: the orientation-expansion partitions of octaves 1..3 on the side streams
  hipEvent_t levelDone[kOctaves][kGauss];  // build_dog: gaussian level b of octave o complete (split DoG schedule)
};

}  // namespace svp

struct ssrlcv_sift_plan {
  uint32_t W, H;
  // makeBinnable (src/Image.cu:966-995) as ScaleSpace::ScaleSpace calls it (src/FeatureFactory.cu:364-376):
  // padMode 0 = sizes already binnable; 1 = even sizes, zero border of (padX, padY) pixels added to the input before the
  // 2x upsample (multiples of 2^3); 2 = an odd side, border added to the upsampled image (multiples of 2^5)
  int padMode;
  uint32_t padX, padY;
  size_t off_pad;      // mode 1: padded u8 input; mode 2: un-padded upsampled f32 image
  ssrlcv_sift_params params;
  svp::OctavePlan oct[svp::kOctaves];
  size_t off_in0, off_in1, off_in2;
  size_t off_gauss[svp::kOctaves][svp::kGauss];  // every octave has its own levels: octave o + 1 is convolved while
                                                 // DoG(o) still reads octave o, octave o + 2 beside levels 4-5 of o + 1
  size_t off_minmax;   // floats: [oct][kGauss + kDog][2]
  size_t off_state;    // OctaveState[kOctaves]
  size_t off_extremaCounts;  // scratch for the pixel-domain partition
  size_t off_dogPartial;     // per-wave {min, max} partials of the streaming DoG kernel: float[2 * kDog][kDogMaxWaves]
  size_t off_groups;         // 4 KB: range tables and control words of the pipelined sampling groups (keypoints.hip)
  size_t total;
  uint32_t maxFeatures;
  int stopStage;
  // which of an octave's two list buffers holds the current key-point list (it ping-pongs once per compaction); written
  // by every key-point stage, read by the next one and by ssrlcv_sift_plan_keypoints.  Host-side state of the (plan,
  // workspace) pair the stages run on: one extraction at a time per plan.
  mutable uint8_t listInB[svp::kOctaves];
  // ssrlcv_hip_sift_extract (both stages in one call): build_dog queues the polar tables of octave o on the `polar` side
  // stream as soon as that octave's DoG min / max exist (they are the first thing the key-point stage needs and the small
  // octaves' launches leave the chip half idle at the end of the scale-space stage); describe then joins them instead of
  // launching them.  The stand-alone stage calls keep their stream-ordered contract: nothing of theirs is left in flight.
  mutable int fusedCall;          // set by extract around its two stage calls
  mutable int polarInFlight;      // build_dog has queued the polar tables (events polarDone[])
  mutable int chain0InFlight;     // bit o: build_dog has queued octave o's list chain on `chain2` (see SSRLCV_EARLY_CHAIN0)
  mutable hipEvent_t stageEvent;  // nullable: recorded by extract on the caller's stream between the two stages
  mutable svp::PlanAsync* async;  // created on first use (needs a device); see svp::plan_async
  mutable int asyncState;         // 0 not tried, 1 ready, -1 serial (SSRLCV_SIFT_SERIAL set or creation failed)
};

namespace svp {
// -> the plan's side streams, or nullptr when the calls must run serially on the caller's stream
PlanAsync* plan_async(const ssrlcv_sift_plan* plan);
int stream_priority_mode();  // developer build: SSRLCV_PRIO (0 in the release build)
// k_polar (keypoints.hip) for ONE octave on `st`: the gradient tables of its DoG levels 1..3
void launch_polar_octave(const ssrlcv_sift_plan* plan, char* ws, int octave, hipStream_t st);
// the list chain (S8 tail - S12, keypoints.hip) of ONE octave on `st`, up to the plan's stop stage
int launch_chain_octave(const ssrlcv_sift_plan* plan, char* ws, int octave, hipStream_t st);
}  // namespace svp
