// common.h -- shared host/device helpers for libtdrn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tdrn_hip.h"

#include <cstdlib>

namespace tdrn {

// Ablation switches (TDRN_CONV_ABLATE, TDRN_PW_ABLATE: skip the loads / the MFMAs / the stores of a kernel) make a launch return
// GARBAGE on purpose.  They exist only in developer builds (make EXTRA=-DTDRN_DEV_ABLATE OUT=... BUILD=..., scripts/dev/pw_ablate.sh):
// in the product library a stray environment variable cannot poison a result.
// What the product library DOES read from the environment (each once, cached per process; all listed in INTEGRATION.md "Environment
// knobs"): kernel-CHOICE and schedule switches -- TDRN_CONV_PATCH / _PP / _PP_SK / _PP_POOL / _PP_NMAJOR / _WS / _VARIANT, TDRN_PW1X1,
// TDRN_PW_NMAJOR, TDRN_PW_STAGGER, TDRN_DWPW, TDRN_DW_SLIDE / _STRIP, TDRN_FIRST_MFMA, TDRN_FUSE_FIRST, TDRN_IGEMM_BATCH_MINOR, TDRN_SPLITK(_REF),
// TDRN_PATCH_SMALL_BN, TDRN_DEFORM_TS / _SPLIT / _XCD, TDRN_SAMPLE_XCD, TDRN_YGEMM_V2 / _CT / _MULTI, TDRN_Y_TAP_MAJOR, TDRN_TS_RANGE_MB, TDRN_PATCH_TAIL,
// TDRN_STREAMS, TDRN_L2_EARLY, TDRN_LATE_SIDE, TDRN_MAIN_GRID, TDRN_SIDE_GRID, TDRN_CHAIN, TDRN_PLAN_DUMP.  Every one of them selects
// among implementations that the GPU tests hold to the same results (bit-identical where tests/test_gpu_pin16.py says so, fp32
// summation noise for the split-K / head3x3 choices): none can make a launch return garbage, which is the line between them and the
// ablation switches above.
inline int dev_ablate_env(const char *name)
{
#ifdef TDRN_DEV_ABLATE
    const char *e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct bf16_t { unsigned short v; };   // storage-only tags for the 16-bit element types
struct f16_t { unsigned short v; };

template <typename T> struct elem_traits;
template <> struct elem_traits<float> { static constexpr int bytes = 4; static constexpr int per16 = 4; };
template <> struct elem_traits<bf16_t> { static constexpr int bytes = 2; static constexpr int per16 = 8; };
template <> struct elem_traits<f16_t> { static constexpr int bytes = 2; static constexpr int per16 = 8; };

// ---- scalar conversions (round-to-nearest-even; NaN stays NaN via the hardware cvt) ----------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950 (keeps NaN a NaN)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }

template <typename T> __device__ __forceinline__ float to_f32(const T &);
template <> __device__ __forceinline__ float to_f32<float>(const float &x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(const bf16_t &x) { return bf16_to_f32(x.v); }
template <> __device__ __forceinline__ float to_f32<f16_t>(const f16_t &x) { return f16_to_f32(x.v); }
template <typename T> __device__ __forceinline__ T from_f32(float);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return bf16_t{f32_to_bf16(x)}; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return f16_t{f32_to_f16(x)}; }

// 16 bytes of T <-> fp32 lanes.  N = elem_traits<T>::per16 (4 or 8).
template <typename T> __device__ __forceinline__ void unpack16(const u32x4 &raw, float *out);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4 &raw, float *out)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(raw[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4 &raw, float *out)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = __uint_as_float(raw[i] << 16);
        out[2 * i + 1] = __uint_as_float(raw[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const u32x4 &raw, float *out)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = f16_to_f32((unsigned short)(raw[i] & 0xffffu));
        out[2 * i + 1] = f16_to_f32((unsigned short)(raw[i] >> 16));
    }
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float *in);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float *in)
{
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(in[i]);
    return r;
}
// two floats -> one dword of 16-bit elements with ONE packed convert (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32; the
// element-wise form costs two converts, a shift and an or per pair)
__device__ __forceinline__ unsigned pack2_bf16(float a, float b)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float fl2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, bf2));
}
__device__ __forceinline__ unsigned pack2_f16(float a, float b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float fl2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, h2));
}
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b) { return pack2_bf16(a, b); }
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float a, float b) { return pack2_f16(a, b); }
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float *in)
{
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2_bf16(in[2 * i], in[2 * i + 1]);
    return r;
}
template <> __device__ __forceinline__ u32x4 pack16<f16_t>(const float *in)
{
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack2_f16(in[2 * i], in[2 * i + 1]);
    return r;
}

// ---- host-side helpers -----------------------------------------------------------------------
inline int hip_status(hipError_t e) { return e == hipSuccess ? TDRN_OK : (int)e; }
#define TDRN_HIP_TRY(expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return (int)_e;                \
    } while (0)
#define TDRN_TRY(expr)                                       \
    do {                                                     \
        int _s = (expr);                                     \
        if (_s != TDRN_OK) return _s;                        \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int dtype_bytes(int dt) { return dt == TDRN_F32 ? 4 : 2; }

// host fp32 -> bf16 / fp16 bit patterns (round to nearest even) for weight packing
unsigned short host_f32_to_bf16(float f);
unsigned short host_f32_to_f16(float f);

}  // namespace tdrn
