// ssrlcv_amd/csrc/filter.hip -- the filters between triangulation and bundle adjustment on the device (SURVEY.md section 8f
// item 1): what PointCloudFactory::linearCutoffFilter (src/PointCloudFactory.cu:3500-3644) and
// deterministicStatisticalFilter (:3070-3275) do on the HOST around their triangulation kernels upstream:
//   * the statistical cutoff -- sigma x the standard deviation of every sampleJump-th bundle error, two sequential float
//     sums (:3121-3156) -- and
//   * the rebuild of the MatchSet without the bundles the cutoff triangulation flagged (:3159-3272, :3517-3644),
// so that a flow that keeps its MatchSet on the GPU (config[4]: 3.1 M multi-matches) filters it there: upstream copies the
// bundles and errors to the host, loops, and copies a rebuilt MatchSet back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_math.h"
#include "scan_lookback.h"
#include "ssrlcv_hip.h"

namespace {

// ---- sigma * sqrt(variance of errors[0], errors[jump], errors[2 jump], ...) in the reference's float order.
// Float addition does not associate, so the two sums are what one host thread computes upstream: a sequential chain.
// One block: 256 threads fetch the next 1024 samples (strided gathers, all in flight) while thread 0 adds the 1024 before
// them out of LDS, four per ds_read_b128 -- the chain costs one dependent v_add_f32 per sample (310 000 samples of a
// 3.1 M-bundle set: ~1.2 ms; the reference's 13 k-bundle fixtures: microseconds).
constexpr int kChunk = 1024;
template <bool SQUARES>
__device__ __forceinline__ float sequential_sample_sum(const float* __restrict__ errors, uint32_t samples, uint32_t jump, float mean,
                                                       float* s_buf /* [2][kChunk] */) {
  float sum = 0.0f;  // thread 0's
  float regs[kChunk / 256];
  auto fetch = [&](uint32_t chunk) {
#pragma unroll
    for (int j = 0; j < kChunk / 256; ++j) {
      const uint32_t k = chunk * kChunk + j * 256 + threadIdx.x;
      regs[j] = k < samples ? errors[(size_t)k * jump] : 0.0f;
    }
  };
  const uint32_t chunks = (samples + kChunk - 1) / kChunk;
  if (chunks) fetch(0);
  for (uint32_t c = 0; c < chunks; ++c) {
    float* buf = s_buf + (c & 1) * kChunk;
#pragma unroll
    for (int j = 0; j < kChunk / 256; ++j) buf[j * 256 + threadIdx.x] = regs[j];
    __syncthreads();  // chunk c is in LDS; the other half was consumed before the previous barrier
    if (c + 1 < chunks) fetch(c + 1);
    if (threadIdx.x == 0) {
      const uint32_t left = samples - c * kChunk, m = left < (uint32_t)kChunk ? left : (uint32_t)kChunk;
      uint32_t k = 0;
      for (; k + 4 <= m; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(buf + k);
        if (SQUARES) {
          sum += (v.x - mean) * (v.x - mean);
          sum += (v.y - mean) * (v.y - mean);
          sum += (v.z - mean) * (v.z - mean);
          sum += (v.w - mean) * (v.w - mean);
        } else {
          sum += v.x;
          sum += v.y;
          sum += v.z;
          sum += v.w;
        }
      }
      for (; k < m; ++k) sum += SQUARES ? (buf[k] - mean) * (buf[k] - mean) : buf[k];
    }
    // no barrier here: the next iteration writes the OTHER half, and its barrier orders this read before the write after it
  }
  __shared__ float s_sum;
  __syncthreads();
  if (threadIdx.x == 0) s_sum = sum;
  __syncthreads();
  return s_sum;
}
__global__ __launch_bounds__(256) void k_sample_cutoff(const float* __restrict__ errors, uint32_t n, uint32_t jump, float sigma,
                                                       float* __restrict__ cutoff) {
  __shared__ __attribute__((aligned(16))) float s_buf[2 * kChunk];
  const uint32_t samples = (uint32_t)((int)(n - (n % jump)) / (int)jump);  // :3124
  const float sample_sum = sequential_sample_sum<false>(errors, samples, jump, 0.0f, s_buf);
  const float sample_mean = sample_sum / (float)samples;                    // :3130 (float / size_t)
  const float squared_sum = sequential_sample_sum<true>(errors, samples, jump, sample_mean, s_buf);
  const float variance = squared_sum / (float)samples;                      // :3142
  if (threadIdx.x == 0) *cutoff = sigma * sqrtf(variance);                  // :3156
}

// ---- the MatchSet without the flagged bundles, order kept, re-indexed: one pass (scan_lookback.h) over three sums --
// bundles kept, their lines, and ALL lines (upstream's k_bundle, k_adjust, k_keypnt of :3253-3268; the two-view loop of
// :3206-3213 is the same with two lines per bundle).
constexpr int kItems = 4;
__global__ __launch_bounds__(svs::kThreads) void k_filter_matchset(const ssrlcv_bundle* __restrict__ bundles, const ssrlcv_keypoint* __restrict__ kpIn,
                                                                   uint32_t n, ssrlcv_multimatch* __restrict__ mmOut,
                                                                   ssrlcv_keypoint* __restrict__ kpOut, uint32_t* __restrict__ counts,
                                                                   svs::TileScan<3> ts) {
  constexpr uint32_t kTile = svs::kThreads * kItems;
  for (uint32_t tile = svs::next_tile(ts.counter); tile < ts.numTiles; tile = svs::next_tile(ts.counter)) {
    const uint32_t base = tile * kTile + threadIdx.x * kItems;
    uint32_t lines[kItems];
    bool keep[kItems];
    uint32_t mine[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      lines[j] = 0;
      keep[j] = false;
      if (base + j < n) {
        const ssrlcv_bundle b = bundles[base + j];
        lines[j] = b.numLines;
        keep[j] = !b.invalid;
      }
      mine[0] += keep[j] ? 1u : 0u;
      mine[1] += keep[j] ? lines[j] : 0u;
      mine[2] += lines[j];
    }
    uint32_t excl[3], total[3], prefix[3];
    svs::block_exclusive<3>(mine, excl, total);
    svs::tile_prefix<3>(ts, tile, total, prefix);
    uint32_t k_bundle = prefix[0] + excl[0], k_adjust = prefix[1] + excl[1], k_keypnt = prefix[2] + excl[2];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      if (keep[j]) {
        ssrlcv_multimatch m;
        m.numKeyPoints = lines[j];
        m.index = (int)k_adjust;
        mmOut[k_bundle] = m;
        for (uint32_t l = 0; l < lines[j]; ++l) kpOut[k_adjust + l] = kpIn[k_keypnt + l];
        k_adjust += lines[j];
        ++k_bundle;
      }
      k_keypnt += lines[j];
    }
    if (tile == ts.numTiles - 1 && threadIdx.x == 0) {
      counts[0] = prefix[0] + total[0];  // bundles kept
      counts[1] = prefix[1] + total[1];  // key points kept
      counts[2] = prefix[2] + total[2];  // key points of the input (all lines)
    }
  }
}

inline uint32_t filter_tiles(uint32_t n) { return (n + svs::kThreads * kItems - 1) / (svs::kThreads * kItems); }

// ---- the two-view bundles of one image pair out of an N-view MatchSet, as a two-camera MatchSet (round 5) -----------------------
// BundleAdjustTwoView works on a 2-view MatchSet; in the N-view flows (config[3]) the sweep is run on the bundles of the
// first image pair.  One look-back pass: a multi-match is taken when it has exactly two key points and they belong to
// images a and b (in that order, which is the order generateMatchesExhaustive writes them); output j = {2, 2 j} with the
// two key points re-labelled 0 / 1, order kept.
__global__ __launch_bounds__(svs::kThreads) void k_select_pair(const ssrlcv_multimatch* __restrict__ mm, const ssrlcv_keypoint* __restrict__ kp,
                                                               uint32_t n, uint32_t numKeyPoints, int imageA, int imageB,
                                                               ssrlcv_multimatch* __restrict__ mmOut, ssrlcv_keypoint* __restrict__ kpOut,
                                                               uint32_t* __restrict__ count, svs::TileScan<1> ts) {
  constexpr uint32_t kTile = svs::kThreads * kItems;
  for (uint32_t tile = svs::next_tile(ts.counter); tile < ts.numTiles; tile = svs::next_tile(ts.counter)) {
    const uint32_t base = tile * kTile + threadIdx.x * kItems;
    bool keep[kItems];
    uint32_t first[kItems];
    uint32_t mine[1] = {0};
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      keep[j] = false;
      first[j] = 0;
      if (base + j < n) {
        const ssrlcv_multimatch m = mm[base + j];
        first[j] = (uint32_t)m.index;
        keep[j] = m.numKeyPoints == 2u && m.index >= 0 && first[j] + 1u < numKeyPoints && kp[first[j]].parentId == imageA &&
                  kp[first[j] + 1u].parentId == imageB;
      }
      mine[0] += keep[j] ? 1u : 0u;
    }
    uint32_t excl[1], total[1], prefix[1];
    svs::block_exclusive<1>(mine, excl, total);
    svs::tile_prefix<1>(ts, tile, total, prefix);
    uint32_t at = prefix[0] + excl[0];
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      if (!keep[j]) continue;
      ssrlcv_keypoint k0 = kp[first[j]], k1 = kp[first[j] + 1u];
      k0.parentId = 0;
      k1.parentId = 1;
      kpOut[2u * at] = k0;
      kpOut[2u * at + 1u] = k1;
      ssrlcv_multimatch m;
      m.numKeyPoints = 2u;
      m.index = (int)(2u * at);
      mmOut[at] = m;
      ++at;
    }
    if (tile == ts.numTiles - 1 && threadIdx.x == 0) *count = prefix[0] + total[0];
  }
}

}  // namespace

extern "C" {

int ssrlcv_hip_error_sample_cutoff(const float* errors, uint32_t numErrors, uint32_t sampleJump, float sigma, float* cutoff,
                                   ssrlcv_stream_t stream) {
  if (!errors || !cutoff || sampleJump == 0 || numErrors > 0x7fffffffu) return SSRLCV_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_sample_cutoff, dim3(1), dim3(256), 0, (hipStream_t)stream, errors, numErrors, sampleJump, sigma, cutoff);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

size_t ssrlcv_hip_filter_workspace_bytes(uint32_t numBundles) { return svs::workspace_bytes<3>(filter_tiles(numBundles ? numBundles : 1)); }

size_t ssrlcv_hip_select_pair_workspace_bytes(uint32_t numMatches) { return svs::workspace_bytes<1>(filter_tiles(numMatches ? numMatches : 1)); }

int ssrlcv_hip_select_pair_bundles(const ssrlcv_multimatch* matches, const ssrlcv_keypoint* keyPoints, uint32_t numMatches,
                                   uint32_t numKeyPoints, int imageA, int imageB, ssrlcv_multimatch* matchesOut,
                                   ssrlcv_keypoint* keyPointsOut, uint32_t* count, void* workspace, size_t workspaceBytes,
                                   ssrlcv_stream_t stream) {
  if (!count) return SSRLCV_ERR_INVALID_ARG;
  const hipStream_t st = (hipStream_t)stream;
  if (numMatches == 0) return (int)hipMemsetAsync(count, 0, 4, st);
  if (!matches || !keyPoints || !matchesOut || !keyPointsOut || !workspace) return SSRLCV_ERR_INVALID_ARG;
  const uint32_t tiles = filter_tiles(numMatches);
  if (workspaceBytes < svs::workspace_bytes<1>(tiles)) return SSRLCV_ERR_WORKSPACE;
  SSRLCV_HIP_TRY(hipMemsetAsync(workspace, 0, svs::workspace_bytes<1>(tiles), st));
  const svs::TileScan<1> ts = svs::make_tile_scan<1>(workspace, tiles);
  const unsigned blocks = tiles < 2048u ? tiles : 2048u;
  hipLaunchKernelGGL(k_select_pair, dim3(blocks), dim3(svs::kThreads), 0, st, matches, keyPoints, numMatches, numKeyPoints, imageA, imageB,
                     matchesOut, keyPointsOut, count, ts);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

int ssrlcv_hip_filter_matchset(const ssrlcv_bundle* bundles, const ssrlcv_keypoint* keyPoints, uint32_t numBundles,
                               ssrlcv_multimatch* matchesOut, ssrlcv_keypoint* keyPointsOut, uint32_t* counts, void* workspace,
                               size_t workspaceBytes, ssrlcv_stream_t stream) {
  if (!counts) return SSRLCV_ERR_INVALID_ARG;
  const hipStream_t st = (hipStream_t)stream;
  if (numBundles == 0) return (int)hipMemsetAsync(counts, 0, 12, st);
  if (!bundles || !keyPoints || !matchesOut || !keyPointsOut || !workspace) return SSRLCV_ERR_INVALID_ARG;
  const uint32_t tiles = filter_tiles(numBundles);
  if (workspaceBytes < svs::workspace_bytes<3>(tiles)) return SSRLCV_ERR_WORKSPACE;
  SSRLCV_HIP_TRY(hipMemsetAsync(workspace, 0, svs::workspace_bytes<3>(tiles), st));
  const svs::TileScan<3> ts = svs::make_tile_scan<3>(workspace, tiles);
  const unsigned blocks = tiles < 2048u ? tiles : 2048u;
  hipLaunchKernelGGL(k_filter_matchset, dim3(blocks), dim3(svs::kThreads), 0, st, bundles, keyPoints, numBundles, matchesOut, keyPointsOut,
                     counts, ts);
  SSRLCV_LAUNCH_CHECK();
  return SSRLCV_OK;
}

}  // extern "C"
