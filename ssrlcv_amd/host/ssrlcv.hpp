// ssrlcv_amd/host/ssrlcv.hpp -- umbrella header of the C++ host mirror of the reference API (hot path only).
#pragma once
#include "Unity.hpp"
#include "Feature.hpp"
#include "Image.hpp"
#include "SIFT_FeatureFactory.hpp"
#include "MatchFactory.hpp"
#include "PointCloudFactory.hpp"
#include "io_util.hpp"
#include "Pipeline.hpp"
