// Derivative stencils of the wind-field operators (shared by elementwise.hip and physics_loss.hip).
#pragma once
#include "common.h"

namespace {

// Wind-field derivatives (see windsr_hip.h).  Row i of the derivative operator along one axis with
// coordinates c[0..n): interior  d_i = a_i f_{i-1} + b_i f_i + c_i f_{i+1},
//   a_i = -hr^2/den, b_i = (hr^2 - hl^2)/den, c_i = hl^2/den, hl = c_i - c_{i-1}, hr = c_{i+1} - c_i,
//   den = hl*hr*(hl + hr);  ends: (f_1 - f_0)/(c_1 - c_0), (f_{n-1} - f_{n-2})/(c_{n-1} - c_{n-2}).
struct Row3 { float a, b, c; };
template <class Coord>
__device__ __forceinline__ Row3 deriv_row(const Coord& co, int i, int n) {
  Row3 r;
  if (n < 2) { r.a = r.b = r.c = 0.f; return r; }
  if (i == 0) { const float h = co(1) - co(0); r.a = 0.f; r.b = -1.f / h; r.c = 1.f / h; return r; }
  if (i == n - 1) { const float h = co(n - 1) - co(n - 2); r.a = -1.f / h; r.b = 1.f / h; r.c = 0.f; return r; }
  const float hl = co(i) - co(i - 1), hr = co(i + 1) - co(i);
  const float den = hl * hr * (hl + hr);
  r.a = -(hr * hr) / den; r.b = (hr * hr - hl * hl) / den; r.c = (hl * hl) / den;
  return r;
}
struct Lin { const float* p; long s; __device__ float operator()(int i) const { return p[(long)i * s]; } };

}  // namespace
