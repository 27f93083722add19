// Fused statistics and gradient of the generator's content losses (reference
// GAN_models/wind_field_GAN_3D.py:377-432 with process_data.py:273-313 and :773-814):
//
//   pix        = mean |HR - SR|  (l1)  or  mean (HR - SR)^2  (l2)
//   xy_grad    = mse(J_sr[:6] / n_xy,  J_hr[:6] / n_xy)            J = the 9-channel Jacobian stack of
//   z_grad     = mse(J_sr[6:] / n_z,   J_hr[6:] / n_z)                 calculate_gradient_of_wind_field
//   div        = mse(div3_hr / n_d3,   div3_sr / n_d3)             div3 = J0 + J4 + J8
//   xy_div     = mse(div2_hr / n_d2,   div2_sr / n_d2)             div2 = J0 + J4
//   n_*        = max(max_hr, max_sr / 100)                         batch-global maxima (|.| except the z one)
//
// Every normaliser is a scalar, so mse(a/n, b/n) = sum (a-b)^2 / (n^2 N): ONE pass over HR, SR and the level
// heights produces the 6 sums and the 8 maxima (14 numbers; the Jacobians are never written to memory), and
// the scalar algebra that follows stays in the caller.  The backward pass takes the gradients of the loss
// with respect to the 6 sums (a 6-float device array: no host round trip) and produces d loss / d SR:
// a residual pass (9 channels) followed by the adjoint of the derivative stencils + the pixel term.
// Reductions are two-pass and atomic-free: per-workgroup partial rows, then one small kernel (deterministic).
//
// All tensors planar fp32 (B, 3, X, Y, Z) / (B, 1, X, Y, Z): the network boundary layout.  HBM-bound:
// forward reads 7 floats per voxel (28 B), backward reads 7 + writes 9, then reads 9 + 7 and writes 3.
#include "common.h"
#include "stencil.h"

namespace {

constexpr int PL_BLOCK = 256;
constexpr int PL_ROWS = 1024;   // max workgroups of the statistics pass = rows of the partial table
constexpr int PL_NS = 6, PL_NM = 8, PL_N = PL_NS + PL_NM;

__device__ __forceinline__ float nanmax(float m, float v) { return (m != m || v != v) ? __builtin_nanf("") : fmaxf(m, v); }

// 3 derivatives x 3 components of one field at voxel (x, y, z); p points at component 0 of that voxel
struct Jac { float j[9]; };
__device__ __forceinline__ Jac jacobian(const float* __restrict__ p, long vol, long plane, int Z, int x, int y, int z, int X,
                                        int Y, const Row3& wx, const Row3& wy, const Row3& wz) {
  Jac o;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* fp = p + (long)c * vol;
    const float f0 = *fp;
    o.j[c] = (x > 0 ? wx.a * fp[-plane] : 0.f) + wx.b * f0 + (x < X - 1 ? wx.c * fp[plane] : 0.f);
    o.j[3 + c] = (y > 0 ? wy.a * fp[-Z] : 0.f) + wy.b * f0 + (y < Y - 1 ? wy.c * fp[Z] : 0.f);
    o.j[6 + c] = (z > 0 ? wz.a * fp[-1] : 0.f) + wz.b * f0 + (z < Z - 1 ? wz.c * fp[1] : 0.f);
  }
  return o;
}

__global__ __launch_bounds__(PL_BLOCK) void physics_stats_kernel(const float* __restrict__ hr, const float* __restrict__ sr,
                                                                const float* __restrict__ xs, const float* __restrict__ ys,
                                                                const float* __restrict__ zc, float* __restrict__ part, int B,
                                                                int X, int Y, int Z) {
  const long plane = (long)Y * Z, vol = (long)X * plane, total = (long)B * vol;
  float s[PL_NS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float ninf = -__builtin_inff();
  float m[PL_NM] = {0.f, ninf, 0.f, 0.f, 0.f, ninf, 0.f, 0.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    long r = i / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X);
    const long b = r / X;
    const long sp = (long)x * plane + (long)y * Z + z;
    const Row3 wx = deriv_row(Lin{xs, 1}, x, X), wy = deriv_row(Lin{ys, 1}, y, Y);
    const Row3 wz = deriv_row(Lin{zc + b * vol + (long)x * plane + (long)y * Z, 1}, z, Z);
    const float* hp = hr + b * 3 * vol + sp;
    const float* qp = sr + b * 3 * vol + sp;
    const Jac jh = jacobian(hp, vol, plane, Z, x, y, z, X, Y, wx, wy, wz);
    const Jac js = jacobian(qp, vol, plane, Z, x, y, z, X, Y, wx, wy, wz);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float d = js.j[c] - jh.j[c];
      if (c < 6) {
        s[0] += d * d;
        m[0] = nanmax(m[0], fabsf(jh.j[c]));
        m[4] = nanmax(m[4], fabsf(js.j[c]));
      } else {
        s[1] += d * d;
        m[1] = nanmax(m[1], jh.j[c]);  // no abs: the reference takes the signed maximum here (:780-781)
        m[5] = nanmax(m[5], js.j[c]);
      }
    }
    const float h2 = jh.j[0] + jh.j[4], s2 = js.j[0] + js.j[4];
    const float h3 = h2 + jh.j[8], s3 = s2 + js.j[8];
    s[2] += (h3 - s3) * (h3 - s3);
    s[3] += (h2 - s2) * (h2 - s2);
    m[2] = nanmax(m[2], fabsf(h3)); m[6] = nanmax(m[6], fabsf(s3));
    m[3] = nanmax(m[3], fabsf(h2)); m[7] = nanmax(m[7], fabsf(s2));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = hp[(long)c * vol] - qp[(long)c * vol];
      s[4] += fabsf(d);
      s[5] += d * d;
    }
  }
  // wave reduction (64 lanes), then across the 4 waves through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < PL_NS; ++k) s[k] += __shfl_xor(s[k], off, 64);
#pragma unroll
    for (int k = 0; k < PL_NM; ++k) m[k] = nanmax(m[k], __shfl_xor(m[k], off, 64));
  }
  __shared__ float sh[PL_BLOCK / 64][PL_N];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < PL_NS; ++k) sh[wave][k] = s[k];
#pragma unroll
    for (int k = 0; k < PL_NM; ++k) sh[wave][PL_NS + k] = m[k];
  }
  __syncthreads();
  if (threadIdx.x < PL_N) {
    const int k = threadIdx.x;
    float a = sh[0][k];
    for (int w = 1; w < PL_BLOCK / 64; ++w) a = k < PL_NS ? a + sh[w][k] : nanmax(a, sh[w][k]);
    part[(long)blockIdx.x * PL_N + k] = a;
  }
}

// rows -> 14 statistics (sums in double: up to 1024 partial rows of up to ~1e5 terms each)
__global__ __launch_bounds__(64 * PL_N) void physics_stats_final_kernel(const float* __restrict__ part, int rows,
                                                                       float* __restrict__ out) {
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;  // one wave per statistic
  if (k < PL_NS) {
    double a = 0.0;
    for (int r = lane; r < rows; r += 64) a += (double)part[(long)r * PL_N + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) out[k] = (float)a;
  } else {
    float a = part[k];  // row 0
    for (int r = lane; r < rows; r += 64) a = nanmax(a, part[(long)r * PL_N + k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a = nanmax(a, __shfl_xor(a, off, 64));
    if (lane == 0) out[k] = a;
  }
}

// residual: r[c] = d loss / d J_sr[c] given coef = d loss / d {s_xy, s_z, s_div3, s_div2, ., .}
__global__ __launch_bounds__(PL_BLOCK) void physics_residual_kernel(const float* __restrict__ hr, const float* __restrict__ sr,
                                                                   const float* __restrict__ xs, const float* __restrict__ ys,
                                                                   const float* __restrict__ zc, const float* __restrict__ coef,
                                                                   float* __restrict__ res, int B, int X, int Y, int Z) {
  const long plane = (long)Y * Z, vol = (long)X * plane, total = (long)B * vol;
  const float g0 = 2.f * coef[0], g1 = 2.f * coef[1], g2 = 2.f * coef[2], g3 = 2.f * coef[3];
  // A coefficient that is EXACTLY zero means the term is not in the loss (the reference drops the four physics terms
  // from the total when one of them is not finite, wind_field_GAN_3D.py:434-445: autograd then never visits them) - it
  // must not turn a non-finite Jacobian entry into 0 * inf = NaN in the generator's gradient.
  const bool on0 = g0 != 0.f, on1 = g1 != 0.f, on2 = g2 != 0.f, on3 = g3 != 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    long r = i / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X);
    const long b = r / X;
    const long sp = (long)x * plane + (long)y * Z + z;
    const Row3 wx = deriv_row(Lin{xs, 1}, x, X), wy = deriv_row(Lin{ys, 1}, y, Y);
    const Row3 wz = deriv_row(Lin{zc + b * vol + (long)x * plane + (long)y * Z, 1}, z, Z);
    const Jac jh = jacobian(hr + b * 3 * vol + sp, vol, plane, Z, x, y, z, X, Y, wx, wy, wz);
    const Jac js = jacobian(sr + b * 3 * vol + sp, vol, plane, Z, x, y, z, X, Y, wx, wy, wz);
    float o[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) o[c] = (c < 6 ? on0 : on1) ? (c < 6 ? g0 : g1) * (js.j[c] - jh.j[c]) : 0.f;
    const float d2 = (js.j[0] + js.j[4]) - (jh.j[0] + jh.j[4]);
    const float d3 = d2 + (js.j[8] - jh.j[8]);
    const float t3 = on2 ? g2 * d3 : 0.f;
    const float t = t3 + (on3 ? g3 * d2 : 0.f);
    o[0] += t; o[4] += t; o[8] += t3;
    float* rp = res + b * 9 * vol + sp;
#pragma unroll
    for (int c = 0; c < 9; ++c) rp[(long)c * vol] = o[c];
  }
}

// dsr = adjoint of the derivative stencils applied to the residual + the pixel term
__global__ __launch_bounds__(PL_BLOCK) void physics_adjoint_kernel(const float* __restrict__ res, const float* __restrict__ hr,
                                                                  const float* __restrict__ sr, const float* __restrict__ xs,
                                                                  const float* __restrict__ ys, const float* __restrict__ zc,
                                                                  const float* __restrict__ coef, float* __restrict__ dsr,
                                                                  int B, int X, int Y, int Z) {
  const long plane = (long)Y * Z, vol = (long)X * plane, total = (long)B * 3 * vol;
  const float c1 = coef[4], c2 = 2.f * coef[5];
  // (no physics term in the loss: the residuals are zeros, and the stencil weights - 1 / dz of two equal levels - must not
  // meet them)
  const bool phys = coef[0] != 0.f || coef[1] != 0.f || coef[2] != 0.f || coef[3] != 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    long r = i / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X); r /= X;
    const int c = (int)(r % 3);
    const long b = r / 3;
    const long gb = ((b * 9) * vol) + (long)x * plane + (long)y * Z + z;
    float acc = 0.f;
    if (phys) {
    {
      const float* gp = res + gb + (long)(0 + c) * vol;
      const Lin co{xs, 1};
      acc += deriv_row(co, x, X).b * gp[0];
      if (x > 0) acc += deriv_row(co, x - 1, X).c * gp[-plane];
      if (x < X - 1) acc += deriv_row(co, x + 1, X).a * gp[plane];
    }
    {
      const float* gp = res + gb + (long)(3 + c) * vol;
      const Lin co{ys, 1};
      acc += deriv_row(co, y, Y).b * gp[0];
      if (y > 0) acc += deriv_row(co, y - 1, Y).c * gp[-Z];
      if (y < Y - 1) acc += deriv_row(co, y + 1, Y).a * gp[Z];
    }
    {
      const float* gp = res + gb + (long)(6 + c) * vol;
      const Lin co{zc + (b * vol + (long)x * plane + (long)y * Z), 1};
      acc += deriv_row(co, z, Z).b * gp[0];
      if (z > 0) acc += deriv_row(co, z - 1, Z).c * gp[-1];
      if (z < Z - 1) acc += deriv_row(co, z + 1, Z).a * gp[1];
    }
    }
    const float d = sr[i] - hr[i];
    acc += (c1 != 0.f ? c1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f) + (c2 != 0.f ? c2 * d : 0.f);
    dsr[i] = acc;
  }
}

inline int pl_grid(long n) {
  long g = (n + PL_BLOCK - 1) / PL_BLOCK;
  if (g > PL_ROWS) g = PL_ROWS;
  return g < 1 ? 1 : (int)g;
}

}  // namespace

extern "C" int64_t wsr_physics_loss_workspace_floats(void) { return (int64_t)PL_ROWS * PL_N; }

extern "C" int wsr_physics_loss_stats(const float* hr, const float* sr, const float* xs, const float* ys, const float* zc,
                                      float* stats, float* workspace, int32_t B, int32_t X, int32_t Y, int32_t Z,
                                      void* stream) {
  if (!hr || !sr || !xs || !ys || !zc || !stats || !workspace || B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return WSR_EINVAL;
  const int grid = pl_grid((long)B * X * Y * Z);
  hipLaunchKernelGGL(physics_stats_kernel, dim3(grid), dim3(PL_BLOCK), 0, as_stream(stream), hr, sr, xs, ys, zc, workspace,
                     B, X, Y, Z);
  WSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(physics_stats_final_kernel, dim3(1), dim3(64 * PL_N), 0, as_stream(stream), workspace, grid, stats);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_physics_loss_bwd(const float* hr, const float* sr, const float* xs, const float* ys, const float* zc,
                                    const float* coef, float* residual, float* dsr, int32_t B, int32_t X, int32_t Y,
                                    int32_t Z, void* stream) {
  if (!hr || !sr || !xs || !ys || !zc || !coef || !residual || !dsr || B <= 0 || X <= 0 || Y <= 0 || Z <= 0)
    return WSR_EINVAL;
  const long nvox = (long)B * X * Y * Z;
  long g = (nvox + PL_BLOCK - 1) / PL_BLOCK;
  if (g > 65535 * 16) g = 65535 * 16;
  hipLaunchKernelGGL(physics_residual_kernel, dim3((unsigned)g), dim3(PL_BLOCK), 0, as_stream(stream), hr, sr, xs, ys, zc,
                     coef, residual, B, X, Y, Z);
  WSR_LAUNCH_CHECK();
  long g3 = (3 * nvox + PL_BLOCK - 1) / PL_BLOCK;
  if (g3 > 65535 * 16) g3 = 65535 * 16;
  hipLaunchKernelGGL(physics_adjoint_kernel, dim3((unsigned)g3), dim3(PL_BLOCK), 0, as_stream(stream), residual, hr, sr,
                     xs, ys, zc, coef, dsr, B, X, Y, Z);
  WSR_LAUNCH_CHECK();
  return 0;
}
