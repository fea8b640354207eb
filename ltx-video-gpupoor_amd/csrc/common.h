// common.h -- shared device helpers for libltxmi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ltxmi.h"

namespace ltxmi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;      // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;        // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;      // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

constexpr int WAVE = 64;

// ---- bf16 <-> f32 ------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// ---- activations ---------------------------------------------------------------
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + fast_exp2(-1.44269504f * x)); }
// gelu(tanh): 0.5 x (1 + tanh(u)) = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608f * (x + 0.044715f * x * x * x);
    return x * fast_rcp(1.0f + fast_exp2(-2.88539008f * u));
}

// two elements at a time in packed-f32 form (v_pk_mul/fma/add_f32): same arithmetic as gelu_tanh_f with the
// constants folded, x * 1/(1 + 2^(-x (c1 + c2 x^2))), c1 = 2 sqrt(2/pi) log2(e), c2 = 0.044715 c1
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_tanh_pk(float& a, float& b) {
    const f32x2 x = {a, b};
    const f32x2 u = x * x * f32x2{-0.10294324f, -0.10294324f} + f32x2{-2.30220820f, -2.30220820f};
    const f32x2 nz = x * u;
    const f32x2 d = f32x2{fast_exp2(nz[0]), fast_exp2(nz[1])} + f32x2{1.0f, 1.0f};
    const f32x2 r = x * f32x2{fast_rcp(d[0]), fast_rcp(d[1])};
    a = r[0];
    b = r[1];
}

// ---- wave reductions -----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_ptr)gsrc, (lds_void_ptr)lds_wave_base, 16, 0, 0);
}

// same through a buffer descriptor: per-lane byte offset voff (bounds-checked against the descriptor's
// num_records: out-of-range lanes deliver zeros) + wave-uniform byte offset soff
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, uint32_t voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

// ---- host-side error plumbing --------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Per-DEVICE launch state (a process may run the DiT on one GPU and the VAE on another):
//   reserve_lds: hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); `done` is the
//   kernel's own bit mask of devices (a function-local static of the launcher).  Returns LTXMI_OK or
//   LTXMI_ERR_LAUNCH with the error text set.
//   device_cu_count: CUs of the current device (cached per device), <= 0 on failure.
int reserve_lds(const void* kernel, int bytes, unsigned long long* done, const char* what);
int device_cu_count(const char* what);
// conv_direct.hip: direct 3x3x3 convolution; -1 = shape not taken (use the implicit GEMM)
int launch_conv3d_direct(const ltxmi_conv3d_args* a, hipStream_t stream);
int64_t conv3d_direct_workspace_bytes(const ltxmi_conv3d_args* a);   // conv_direct.hip: the workspace a call would like (0: none)
bool conv3d_direct_fuses_post_norm(const ltxmi_conv3d_args* a);     // conv_direct.hip: would this call apply post_norm in its epilogue

}  // namespace ltxmi

#define LTXMI_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            ltxmi::set_error(__VA_ARGS__);      \
            return (code);                      \
        }                                       \
    } while (0)
