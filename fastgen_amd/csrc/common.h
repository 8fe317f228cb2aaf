// Shared device/host helpers for the fastgen_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FG_WAVE 64

// ---- compute-dtype traits -------------------------------------------------------------------------
// KC = input channels per K-chunk; a chunk row is always 128 bytes in LDS (64 bf16 / 32 fp32).
template <typename T>
struct DT;
template <>
struct DT<__bf16> {
    static constexpr int KC = 64;
    static constexpr bool FAST = true;  // hardware exp2/rcp in the SiLU prologue and softmax
};
template <>
struct DT<float> {
    static constexpr int KC = 32;
    static constexpr bool FAST = false;  // accurate expf + IEEE division (fp32 parity mode)
};

// 8 consecutive K elements of one MFMA operand row, as held by one lane.
template <typename T>
struct Frag8;
template <>
struct Frag8<__bf16> {
    bf16x8 v;
};
template <>
struct Frag8<float> {
    f32x4 lo, hi;
};

// acc += A(32 x 16) * B(16 x 32) with this lane's 8-element slices of A and B.
// bf16: one v_mfma_f32_32x32x16_bf16.  fp32: eight v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); element j of
// lane half h is K index 8h+j on BOTH operands, so the pairing (j from half 0, j from half 1) per MFMA is consistent.
__device__ __forceinline__ void mma16(f32x16& acc, const Frag8<__bf16>& a, const Frag8<__bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x16& acc, const Frag8<float>& a, const Frag8<float>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[0], b.lo[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[1], b.lo[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[2], b.lo[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[3], b.lo[3], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[0], b.hi[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[1], b.hi[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[2], b.hi[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[3], b.hi[3], acc, 0, 0, 0);
}

// Fragment loads through a buffer resource: wave-uniform byte offset in an SGPR (soffset), lane part in one VGPR, so the
// load needs no vector address arithmetic (weights streamed from L2 in the MFMA loops).
typedef __attribute__((ext_vector_type(4))) unsigned fg_u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// 16-byte-aligned fragment loads (global or LDS; address space is inferred after inlining).
__device__ __forceinline__ Frag8<__bf16> load_frag(const __bf16* p) {
    Frag8<__bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
__device__ __forceinline__ Frag8<float> load_frag(const float* p) {
    Frag8<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    return f;
}
// lane_elems = lane * 8 (element offset of this lane inside a 512-element fragment)
__device__ __forceinline__ void load_frag_rsrc(Frag8<__bf16>& f, __amdgpu_buffer_rsrc_t r, int lane_elems, int elem_off) {
    f.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_elems * 2, elem_off * 2, 0));
}
__device__ __forceinline__ void load_frag_rsrc(Frag8<float>& f, __amdgpu_buffer_rsrc_t r, int lane_elems, int elem_off) {
    f.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_elems * 4, elem_off * 4, 0));
    f.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_elems * 4 + 16, elem_off * 4, 0));
}
__device__ __forceinline__ void store_frag(__bf16* p, const float (&x)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j];
    *reinterpret_cast<bf16x8*>(p) = v;
}
__device__ __forceinline__ void store_frag(float* p, const float (&x)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{x[4], x[5], x[6], x[7]};
}

// a loaded 8-element fragment widened to fp32
__device__ __forceinline__ void widen8(const Frag8<__bf16>& f, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)f.v[j];
}
__device__ __forceinline__ void widen8(const Frag8<float>& f, float (&v)[8]) {
    v[0] = f.lo[0]; v[1] = f.lo[1]; v[2] = f.lo[2]; v[3] = f.lo[3];
    v[4] = f.hi[0]; v[5] = f.hi[1]; v[6] = f.hi[2]; v[7] = f.hi[3];
}

// 4 consecutive activation elements <-> fp32
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const __bf16* p) {
    const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
}
// stores v in the activation dtype and returns the stored values widened back to fp32
__device__ __forceinline__ f32x4 store4(float* p, f32x4 v) {
    *reinterpret_cast<f32x4*>(p) = v;
    return v;
}
__device__ __forceinline__ f32x4 store4(__bf16* p, f32x4 v) {
    const bf16x4 q = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = q;
    return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
}

// raw (un-widened) 4-element activation vectors: lets a load stay in flight without a dependent conversion
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<__bf16> { typedef bf16x4 type; };
__device__ __forceinline__ f32x4 raw_load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ bf16x4 raw_load4(const __bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ f32x4 widen4(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 widen4(bf16x4 q) { return f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]}; }

// SiLU = x * sigmoid(x) (torch.nn.functional.silu, used at EDM/network.py:276,283,520-521,556).
template <bool FAST>
__device__ __forceinline__ float silu_f(float x) {
    if (FAST) {
        // x / (1 + 2^(-x*log2e)) with v_exp_f32 / v_rcp_f32 (about 1 ulp each)
        float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * x);
        return x * __builtin_amdgcn_rcpf(1.0f + e);
    } else {
        return x / (1.0f + expf(-x));
    }
}

// Row (in 32-pixel MFMA tile) of accumulator register i for lane half h: C/D map of the 32x32 MFMA shapes.
__device__ __forceinline__ constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
