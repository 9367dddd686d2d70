// ssrlcv_amd/host/io_util.hpp -- the one IO routine the stage flow needs: the simple ASCII PLY writer
// (src/io_util.cpp:740-754, called by doTriangulation src/Pipeline.cu:276).  Same header lines, vertex format and
// default float formatting; `dir` defaults to the reference's "out/".  Image decoding, tinyply and CSV writers are out
// of scope.
#pragma once
#include <fstream>
#include <string>
#include "Unity.hpp"
#include "cuda_vec_types.hpp"

namespace ssrlcv {
inline void writePLY(std::string filename, ptr::value<Unity<float3>> points, std::string dir = "out/") {
  MemoryState origin = points->getMemoryState();
  if (origin == gpu || points->getFore() == gpu) points->transferMemoryTo(cpu);
  std::ofstream of;
  of.open(dir + filename + ".ply");
  of << "ply\nformat ascii 1.0\n";
  of << "comment author: SSRLCV simple PLY writer (MI355X build)\n";
  of << "element vertex " << points->size() << "\n";
  of << "property float x\nproperty float y\nproperty float z\n";
  of << "end_header\n";
  float3* p = points->host.get();
  for (unsigned long i = 0; i < points->size(); i++) of << p[i].x << " " << p[i].y << " " << p[i].z << "\n";
  of.close();
  if (origin == gpu) points->setMemoryState(gpu);
}
}  // namespace ssrlcv
