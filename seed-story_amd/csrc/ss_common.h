// Common device/host helpers for libseedstory_hip.so (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes; blocks are multiples of 64 threads;
//   * activations are row-major [rows, features] in the "model dtype" T
//     (float, bf16 or fp16 — the reference runs fp16/bf16, src/inference/gen_george.py:19,
//     and fp32 is the CPU-parity mode);
//   * all reductions / accumulations are fp32; values are rounded to T exactly where the
//     reference's torch graph rounds them (cited per kernel);
//   * no kernel allocates; workspaces are passed in by the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/seedstory_hip.h"

namespace ss {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };
struct fp8_t { uint8_t v; };   // OCP e4m3fn (gfx950's FP8; NOT MI300X's fnuz): GEMM operand type only, never an output

// ---- scalar conversions ---------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// round-to-nearest-even fp32 -> bf16 (matches torch's float->bfloat16 cast; NaN stays a quiet NaN):
// gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32), one VALU op per PAIR of values
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_bits(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
    _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)b);
    return (float)h;
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE
    return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

template <typename T> struct Tr;
template <> struct Tr<float> {
    static constexpr int kDtype = SS_F32;
    static constexpr int kVec = 4;  // elements per 16-byte access
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }  // round to T and back
};
template <> struct Tr<bf16_t> {
    static constexpr int kDtype = SS_BF16;
    static constexpr int kVec = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_bits_to_f32(p->v); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = (uint16_t)f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};
template <> struct Tr<f16_t> {
    static constexpr int kDtype = SS_F16;
    static constexpr int kVec = 8;
    static __device__ __forceinline__ float ld(const f16_t* p) { return f16_bits_to_f32(p->v); }
    static __device__ __forceinline__ void st(f16_t* p, float v) { p->v = (uint16_t)f32_to_f16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
};

// ---- 16-byte vectors of T <-> fp32 ------------------------------------------------------
// A "pack" is one uint4 (16 bytes) = Tr<T>::kVec elements.
template <typename T> __device__ __forceinline__ void unpack(const uint4& u, float* f);
template <> __device__ __forceinline__ void unpack<float>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> __device__ __forceinline__ void unpack<bf16_t>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack<f16_t>(const uint4& u, float* f) {
    f[0] = f16_bits_to_f32(u.x & 0xffffu); f[1] = f16_bits_to_f32(u.x >> 16);
    f[2] = f16_bits_to_f32(u.y & 0xffffu); f[3] = f16_bits_to_f32(u.y >> 16);
    f[4] = f16_bits_to_f32(u.z & 0xffffu); f[5] = f16_bits_to_f32(u.z >> 16);
    f[6] = f16_bits_to_f32(u.w & 0xffffu); f[7] = f16_bits_to_f32(u.w >> 16);
}
template <typename T> __device__ __forceinline__ uint4 pack(const float* f);
template <> __device__ __forceinline__ uint4 pack<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack<bf16_t>(const float* f) {
    return make_uint4(f32x2_to_bf16x2_bits(f[0], f[1]), f32x2_to_bf16x2_bits(f[2], f[3]),
                      f32x2_to_bf16x2_bits(f[4], f[5]), f32x2_to_bf16x2_bits(f[6], f[7]));
}
template <> __device__ __forceinline__ uint4 pack<f16_t>(const float* f) {
    return make_uint4(f32_to_f16_bits(f[0]) | (f32_to_f16_bits(f[1]) << 16),
                      f32_to_f16_bits(f[2]) | (f32_to_f16_bits(f[3]) << 16),
                      f32_to_f16_bits(f[4]) | (f32_to_f16_bits(f[5]) << 16),
                      f32_to_f16_bits(f[6]) | (f32_to_f16_bits(f[7]) << 16));
}

// dot of two packs, fp32 accumulate.  bf16 uses v_dot2c_f32_bf16 (no unpacking).
template <typename T> __device__ __forceinline__ float dot_pack(const uint4& a, const uint4& b, float acc);
template <> __device__ __forceinline__ float dot_pack<float>(const uint4& a, const uint4& b, float acc) {
    acc = fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
    acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
    acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
    acc = fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
    return acc;
}
template <> __device__ __forceinline__ float dot_pack<bf16_t>(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
    return acc;
}
template <> __device__ __forceinline__ float dot_pack<f16_t>(const uint4& a, const uint4& b, float acc) {
    float fa[8], fb[8];
    unpack<f16_t>(a, fa); unpack<f16_t>(b, fb);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(fa[i], fb[i], acc);
    return acc;
}

// ---- wave / block reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum; `red` is >= 16 floats of LDS; every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// non-temporal 16-byte load for data that is streamed exactly once (decode weights)
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, const uint4& v) { *reinterpret_cast<uint4*>(p) = v; }

// ---- host side ------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define SS_HIP(expr)                                                   \
    do {                                                               \
        int _rc = ::ss::check_hip((expr), #expr);                      \
        if (_rc) return _rc;                                           \
    } while (0)
// Raise a kernel's dynamic-LDS limit: once per (kernel instantiation, device) — the attribute belongs to the DEVICE's copy of the
// code object, so a process that drives several devices must set it on each — with the return code checked (a part with
// less LDS fails here, by name, not later as a generic launch error).  `done` = the call site's static device mask.
inline int ensure_dyn_lds(const void* kern, size_t bytes, uint64_t& done) {
    int dev = 0;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done & bit)) {
        rc = check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (rc) return rc;
        done |= bit;
    }
    return 0;
}
#define SS_DYN_LDS(kern, bytes)                                        \
    do {                                                               \
        static uint64_t _done = 0;                                     \
        int _rc = ::ss::ensure_dyn_lds((const void*)(kern), (bytes), _done); \
        if (_rc) return _rc;                                           \
    } while (0)
#define SS_LAUNCH_CHECK(name)                                          \
    do {                                                               \
        int _rc = ::ss::check_hip(hipGetLastError(), name);            \
        if (_rc) return _rc;                                           \
    } while (0)
#define SS_REQUIRE(cond, ...)                                          \
    do {                                                               \
        if (!(cond)) {                                                 \
            ::ss::set_error(__VA_ARGS__);                              \
            return SS_EINVAL;                                          \
        }                                                              \
    } while (0)

// dtype dispatch: calls F<T>(args...) for the runtime dtype code
#define SS_DISPATCH(dtype, FN, ...)                                                    \
    ((dtype) == SS_BF16 ? FN<::ss::bf16_t>(__VA_ARGS__)                                \
     : (dtype) == SS_F32 ? FN<float>(__VA_ARGS__)                                      \
     : (dtype) == SS_F16 ? FN<::ss::f16_t>(__VA_ARGS__)                                \
                         : (::ss::set_error("unsupported dtype %d", (int)(dtype)), SS_EINVAL))

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline size_t dtype_size(int dt) { return dt == SS_F32 ? 4 : 2; }

int tuning_get(const char* key, int dflt);
void tuning_set(const char* key, int value);   // library-side counters readable through ss_get_tuning

}  // namespace ss
