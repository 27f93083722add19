// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16 family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/windsr_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define WSR_LAUNCH_CHECK()                        \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

// ---- element traits ---------------------------------------------------------
// A "piece" is 16 bytes of consecutive channels: 4 fp32 or 8 bf16.
struct F32 {
  using elem = float;
  static constexpr int EPP = 4;
  static constexpr int ID = WSR_F32;
};
struct BF16 {
  using elem = unsigned short;
  static constexpr int EPP = 8;
  static constexpr int ID = WSR_BF16;
};

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}

template <class T> __device__ __forceinline__ float ldf(const typename T::elem* p);
template <> __device__ __forceinline__ float ldf<F32>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<BF16>(const unsigned short* p) { return bf2f(*p); }
template <class T> __device__ __forceinline__ void stf(typename T::elem* p, float v);
template <> __device__ __forceinline__ void stf<F32>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<BF16>(unsigned short* p, float v) { *p = f2bf(v); }

// 4 consecutive elements <-> float4
template <class T> __device__ __forceinline__ float4 ld4(const typename T::elem* p);
template <> __device__ __forceinline__ float4 ld4<F32>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<BF16>(const unsigned short* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <class T> __device__ __forceinline__ void st4(typename T::elem* p, float4 v);
template <> __device__ __forceinline__ void st4<F32>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<BF16>(unsigned short* p, float4 v) {
  uint2 u;
  u.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
  u.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

// One 64-byte K-chunk of a 16x16 output tile.  Lane (r = lane&15, g = lane>>4)
// holds the 16 bytes at K-offset 16g of row r for both operands.
//   bf16: one v_mfma_f32_16x16x32_bf16 (k = 8g + j)
//   f32 : four v_mfma_f32_16x16x4_f32; MFMA j contracts k = {4g + j}, the same
//         permutation on both operands, so the sum over the chunk is exact.
template <class T> __device__ __forceinline__ void mma_chunk(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma_chunk<BF16>(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<F32>(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// XCD-aware block remap: consecutive logical tiles land on the same XCD (shared
// L2) although the dispatcher deals workgroups round-robin over the 8 XCDs.
// Bijective for any grid size (cdna guide 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- tuning switches ------------------------------------------------------------------------------------
// Environment switches (WSR_*) are tuning / A-B aids.  Reading them with getenv on every launch cost 3-7 scans of
// the environment per launch (~2 000 launches per step: milliseconds of host time on the launch-bound real-data
// shapes).  Each call site now caches its value; wsr_reload_env() (C ABI) bumps the generation so that a process
// that changes the environment at run time (the tests do) sees the new values.
// Thread-safe: the generation and every call site's (generation, value) pair are single atomics - a race costs a
// second getenv, never a torn read (entry points may be called from the autograd thread and the main thread at once).
#include <atomic>
#include <climits>
#include <cstdlib>
extern std::atomic<int> g_wsr_env_gen;  // elementwise.hip
struct WsrEnvCache {
  std::atomic<unsigned long long> gv{~0ull};  // (generation << 32) | value bits; INT_MIN: variable not set
};
static inline int wsr_env_lookup(WsrEnvCache& c, const char* name) {
  const int gen = g_wsr_env_gen.load(std::memory_order_acquire);
  unsigned long long gv = c.gv.load(std::memory_order_relaxed);
  if ((int)(gv >> 32) != gen) {
    const char* e = getenv(name);
    const int val = e ? atoi(e) : INT_MIN;
    gv = ((unsigned long long)(unsigned)gen << 32) | (unsigned)val;
    c.gv.store(gv, std::memory_order_relaxed);
  }
  return (int)(unsigned)(gv & 0xFFFFFFFFull);
}
// value of an integer switch, or INT_MIN when it is not set
#define WSR_ENV_RAW(name) ([]() -> int { static WsrEnvCache c_; return wsr_env_lookup(c_, name); }())
#define WSR_ENV_SET(name) (WSR_ENV_RAW(name) != INT_MIN)
#define WSR_ENV_INT(name, dflt) (WSR_ENV_SET(name) ? WSR_ENV_RAW(name) : (dflt))

static inline int conv_geom_ok(const wsr_conv_t* c);
// the same for a conv whose input channels >= c0 live in a second tensor (ABI 8): the window of the FIRST tensor only
// has to hold channels [0, c0)
static inline int conv_geom_ok_split(const wsr_conv_t* c, int c0) {
  if (!c || c0 <= 0 || c0 >= c->Cin) return 0;
  wsr_conv_t t = *c;
  t.Cin = c0;
  return conv_geom_ok(&t);
}
static inline int conv_geom_ok(const wsr_conv_t* c) {
  if (!c) return 0;
  if (c->B <= 0 || c->Xi <= 0 || c->Yi <= 0 || c->Zi <= 0 || c->Xo <= 0 || c->Yo <= 0 || c->Zo <= 0) return 0;
  if (c->Cin <= 0 || c->Cout <= 0 || c->in_off < 0 || c->out_off < 0) return 0;
  if (c->in_off + c->Cin > c->in_ctot || c->out_off + c->Cout > c->out_ctot) return 0;
  if (c->KX <= 0 || c->KY <= 0 || c->KZ <= 0 || c->KX * c->KY * c->KZ > 125) return 0;
  if (c->sx <= 0 || c->sy <= 0 || c->sz <= 0 || c->px < 0 || c->py < 0 || c->pz < 0) return 0;
  if (c->dtype != WSR_F32 && c->dtype != WSR_BF16) return 0;
  if (c->lat) {  // parity conv of a sub-pixel up-sampling conv: same-size, stride 1, uneven pads (see wsr_conv_t)
    if ((c->lat != 2 && c->lat != 3) || c->upsample_xy || (c->sx | c->sy | c->sz) != 1) return 0;
    if (c->lat == 3 && c->lat_phases) return 0;  // (lat = 3: the INPUT sits on the lattice - filter gradients only)
    if ((unsigned)c->lat_ox > 1u || (unsigned)c->lat_oy > 1u || (c->lat_phases != 0 && c->lat_phases != 4)) return 0;
    if (c->Xo != c->Xi || c->Yo != c->Yi || c->px >= c->KX || c->py >= c->KY) return 0;
    if (c->lat_phases == 4 && (c->px < 1 || c->py < 1 || c->lat_ox || c->lat_oy)) return 0;
    if (c->lat_mz != 0 && c->lat_mz != 1 && c->lat_mz != 2) return 0;
    if (c->lat_oz < 0 || c->lat_oz >= (c->lat_mz > 1 ? c->lat_mz : 1)) return 0;
    return c->Zo == c->Zi && c->pz < c->KZ;
  }
  if (c->lat_ox | c->lat_oy | c->lat_phases | c->lat_mz | c->lat_oz) return 0;
  const int ux = c->upsample_xy ? 2 : 1;
  if ((c->Xi * ux + 2 * c->px - c->KX) / c->sx + 1 != c->Xo) return 0;
  if ((c->Yi * ux + 2 * c->py - c->KY) / c->sy + 1 != c->Yo) return 0;
  if ((c->Zi + 2 * c->pz - c->KZ) / c->sz + 1 != c->Zo) return 0;
  return 1;
}
