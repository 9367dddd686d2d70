// ssrlcv_amd/host/matrix_util.hpp -- the few host-side 3x3 helpers the stage glue and the pose estimator call
// (reference src/matrix_util.cu:52-62 multiply, :102-110 transpose, :244-250 getAxisRotations, :257-267
// getRotationMatrix, :314-327 rotatePoint, :339-356 rotatePointArbitrary; src/cuda_vec_util.cu normalize), and the
// symmetric pseudo-inverse both second-order solvers need (cusolverDnSgesvd upstream).
#pragma once
#include <cmath>
#include <vector>
#include "cuda_vec_types.hpp"

namespace ssrlcv {

inline void multiply(const float (&A)[3][3], const float (&B)[3][3], float (&C)[3][3]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float entry = 0;
      for (int z = 0; z < 3; ++z) entry += A[r][z] * B[z][c];
      C[r][c] = entry;
    }
}
inline void transpose(const float (&M)[3][3], float (&M_out)[3][3]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M_out[r][c] = M[c][r];
}
inline float3 getAxisRotations(const float (&R)[3][3]) {
  float x = atan2f(R[2][1], R[2][2]);
  float y = atan2f(-R[2][0], (R[2][2] / cosf(x)));
  float z = atan2f(R[1][0], R[0][0]);
  return {x, y, z};
}
inline void getRotationMatrix(const float3& angle, float (&R)[3][3]) {
  R[0][0] = cosf(angle.z) * cosf(angle.y);
  R[0][1] = cosf(angle.z) * sinf(angle.y) * sinf(angle.x) - sinf(angle.z) * cosf(angle.x);
  R[0][2] = cosf(angle.z) * sinf(angle.y) * cosf(angle.x) + sinf(angle.z) * sinf(angle.x);
  R[1][0] = sinf(angle.z) * cosf(angle.y);
  R[1][1] = sinf(angle.z) * sinf(angle.y) * sinf(angle.x) + cosf(angle.z) * cosf(angle.x);
  R[1][2] = sinf(angle.z) * sinf(angle.y) * cosf(angle.x) - cosf(angle.z) * sinf(angle.x);
  R[2][0] = -1 * sinf(angle.y);
  R[2][1] = cosf(angle.y) * sinf(angle.x);
  R[2][2] = cosf(angle.y) * cosf(angle.x);
}
inline float3 matrixMulVector(float3 x, const float (&A)[3][3]) {
  float t[3] = {x.x, x.y, x.z}, b[3];
  for (int r = 0; r < 3; ++r) {
    float val = 0;
    for (int c = 0; c < 3; ++c) val += A[r][c] * t[c];
    b[r] = val;
  }
  return {b[0], b[1], b[2]};
}
inline float3 rotatePoint(float3 point, float3 angle) {
  float R[3][3];
  getRotationMatrix(angle, R);
  return matrixMulVector(point, R);
}
inline float3 rotatePointArbitrary(float3 point, float3 axis, float angle) {
  float R[3][3];
  float k = (1 - cosf(angle));
  float m = sqrtf(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
  axis = {axis.x / m, axis.y / m, axis.z / m};
  R[0][0] = axis.x * axis.x * k + cosf(angle);
  R[0][1] = axis.x * axis.y * k - axis.z * sinf(angle);
  R[0][2] = axis.x * axis.z * k + axis.y * sinf(angle);
  R[1][0] = axis.x * axis.y * k + axis.z * sinf(angle);
  R[1][1] = axis.y * axis.y * k + cosf(angle);
  R[1][2] = axis.y * axis.z * k - axis.x * sinf(angle);
  R[2][0] = axis.x * axis.z * k - axis.y * sinf(angle);
  R[2][1] = axis.y * axis.z * k + axis.x * sinf(angle);
  R[2][2] = axis.z * axis.z * k + cosf(angle);
  return matrixMulVector(point, R);
}

// V S' U^T of an N x N matrix (row-major) as calculateImageHessianInverse builds it (src/PointCloudFactory.cu:1511-1824:
// cusolverDnSgesvd, then `if (S[i] >= 0.0001) S[i] = 1.0 / S[i]` (:1698) -- a singular value BELOW the cutoff is not
// zeroed, it keeps its own small value as the factor -- then V S' U^T by two cublasSgemm): one-sided Jacobi SVD in
// double on the host.  No reference fixture reaches the cuSOLVER results it replaces; held to the oracle's independent
// restatement and a numpy fixture (tests/test_pinv.py).
inline std::vector<float> pseudoInverse(const float* M, int N, double cutoff = 1e-4) {
  std::vector<double> U(N * N), Vm(N * N, 0.0);
  for (int i = 0; i < N * N; ++i) U[i] = M[i];
  for (int i = 0; i < N; ++i) Vm[i * N + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        double a = 0, b = 0, c = 0;
        for (int r = 0; r < N; ++r) { a += U[r * N + p] * U[r * N + p]; b += U[r * N + q] * U[r * N + q]; c += U[r * N + p] * U[r * N + q]; }
        off += c * c;
        if (std::fabs(c) <= 1e-30 * std::sqrt(a * b) || c == 0.0) continue;
        double zeta = (b - a) / (2.0 * c);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int r = 0; r < N; ++r) {
          double up = U[r * N + p], uq = U[r * N + q];
          U[r * N + p] = cs * up - sn * uq;
          U[r * N + q] = sn * up + cs * uq;
          double vp = Vm[r * N + p], vq = Vm[r * N + q];
          Vm[r * N + p] = cs * vp - sn * vq;
          Vm[r * N + q] = sn * vp + cs * vq;
        }
      }
    if (off < 1e-60) break;
  }
  std::vector<double> out(N * N, 0.0);
  for (int k = 0; k < N; ++k) {
    double s = 0;
    for (int r = 0; r < N; ++r) s += U[r * N + k] * U[r * N + k];
    s = std::sqrt(s);
    if (s == 0.0) continue;
    const double f = s >= cutoff ? 1.0 / s : s;  // :1698
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) out[i * N + j] += Vm[i * N + k] * (U[j * N + k] / s) * f;  // V S' U^T
  }
  std::vector<float> inv(N * N);
  for (int i = 0; i < N * N; ++i) inv[i] = (float)out[i];
  return inv;
}

}  // namespace ssrlcv
