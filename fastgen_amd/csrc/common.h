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

// Host side: the current device's slot (0..15) of the per-device caches the launchers keep (function attributes belong to the
// device's copy of the code object; the CU count to the device), or -1.
inline int fg_device_slot() {
    int dev = -1;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : -1;
}

// ---- compute-dtype traits -------------------------------------------------------------------------
// KC = input channels per K-chunk; a chunk row is always 128 bytes in LDS (64 bf16 / 32 fp32).
// ST = storage type of activation tensors, WT = element type of packed weights, WPARTS = weight planes per fragment,
// FRAG_BYTES = LDS bytes of one lane's 8-element A fragment (of one plane).
template <typename T>
struct DT;
template <>
struct DT<__bf16> {
    typedef __bf16 ST;
    typedef __bf16 WT;
    static constexpr int KC = 64;
    static constexpr int WPARTS = 1;
    static constexpr int FRAG_BYTES = 16;
    static constexpr bool FAST = true;  // hardware exp2/rcp in the SiLU prologue and softmax
};
template <>
struct DT<float> {
    typedef float ST;
    typedef float WT;
    static constexpr int KC = 32;
    static constexpr int WPARTS = 1;
    static constexpr int FRAG_BYTES = 32;
    static constexpr bool FAST = false;  // accurate expf + IEEE division (fp32 parity mode)
};
// Split-bf16 compute ("bf16x3", FG_DTYPE_BF16X3): tensors are fp32 in memory; every matrix operand is split into
// hi = bf16(x), lo = bf16(x - hi) — activations on their way into LDS, weights at pack time — and a product is evaluated as
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the bf16 matrix pipe with fp32 accumulation.  hi + lo carries 16-17 significant bits
// of x (|x - hi - lo| <= 2^-18 |x|), the dropped a_lo*b_lo term is <= 2^-18 |a b|: about 2^-17 relative per product, 64 times
// tighter than the TF32 arithmetic (2^-11) the reference runs on NVIDIA GPUs (fastgen/utils/scripts.py:43-45), at a third
// of the bf16 MFMA rate instead of the sixteenth that exact-fp32 MFMA runs at (gfx950 has no TF32 matrix instruction).
// A K-chunk is 32 input channels: [32 hi | 32 lo] = 128 bytes per pixel in LDS, the same pitch as the other modes.
struct bf16x3 {};
template <>
struct DT<bf16x3> {
    typedef float ST;
    typedef __bf16 WT;
    static constexpr int KC = 32;
    static constexpr int WPARTS = 2;
    static constexpr int FRAG_BYTES = 16;
    static constexpr bool FAST = true;  // v_exp / v_rcp are ~1 ulp: far inside the mode's 2^-17
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
template <>
struct Frag8<bf16x3> {
    bf16x8 hi, lo;
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

// bf16x3: small terms first, then the leading product (one accumulator chain: back-to-back MFMAs on one accumulator issue at
// full rate on gfx950)
__device__ __forceinline__ void mma16(f32x16& acc, const Frag8<bf16x3>& a, const Frag8<bf16x3>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
// x = hi + lo (+ <= 2^-18 |x|): round-to-nearest-even both times
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)x[j];
        lo[j] = (__bf16)(x[j] - (float)hi[j]);
    }
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
// bf16x3: hi plane at elem_off, lo plane 512 elements (one fragment of 64 lanes x 8) behind it
__device__ __forceinline__ void load_frag_rsrc(Frag8<bf16x3>& f, __amdgpu_buffer_rsrc_t r, int lane_elems, int elem_off) {
    f.hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_elems * 2, elem_off * 2, 0));
    f.lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_elems * 2, elem_off * 2 + 1024, 0));
}
// a packed weight fragment straight from memory (p: the fragment's first element for this lane)
template <typename T>
__device__ __forceinline__ Frag8<T> load_wfrag(const typename DT<T>::WT* p) {
    return load_frag(p);
}
template <>
__device__ __forceinline__ Frag8<bf16x3> load_wfrag<bf16x3>(const __bf16* p) {
    Frag8<bf16x3> f;
    f.hi = *reinterpret_cast<const bf16x8*>(p);
    f.lo = *reinterpret_cast<const bf16x8*>(p + 512);
    return f;
}
// A fragment of 16-deep MFMA step kk out of a pixel's K-chunk in LDS (p: the lane's pixel + lane-half offset)
template <typename T>
__device__ __forceinline__ Frag8<T> lds_read_a(const char* p, int kk);
template <>
__device__ __forceinline__ Frag8<__bf16> lds_read_a<__bf16>(const char* p, int kk) {
    Frag8<__bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p + kk * 32);
    return f;
}
template <>
__device__ __forceinline__ Frag8<float> lds_read_a<float>(const char* p, int kk) {
    Frag8<float> f;
    f.lo = *reinterpret_cast<const f32x4*>(p + kk * 64);
    f.hi = *reinterpret_cast<const f32x4*>(p + kk * 64 + 16);
    return f;
}
template <>
__device__ __forceinline__ Frag8<bf16x3> lds_read_a<bf16x3>(const char* p, int kk) {
    Frag8<bf16x3> f;
    f.hi = *reinterpret_cast<const bf16x8*>(p + kk * 32);
    f.lo = *reinterpret_cast<const bf16x8*>(p + 64 + kk * 32);
    return f;
}
// park 8 transformed channels (one octet of a pixel's K-chunk) in LDS; p: pixel + octet * FRAG_BYTES
template <typename T>
__device__ __forceinline__ void lds_store_a(char* p, const float (&x)[8]);
__device__ __forceinline__ void store_frag(__bf16* p, const float (&x)[8]);
__device__ __forceinline__ void store_frag(float* p, const float (&x)[8]);
template <>
__device__ __forceinline__ void lds_store_a<__bf16>(char* p, const float (&x)[8]) { store_frag(reinterpret_cast<__bf16*>(p), x); }
template <>
__device__ __forceinline__ void lds_store_a<float>(char* p, const float (&x)[8]) { store_frag(reinterpret_cast<float*>(p), x); }
template <>
__device__ __forceinline__ void lds_store_a<bf16x3>(char* p, const float (&x)[8]) {
    bf16x8 hi, lo;
    split8(x, hi, lo);
    *reinterpret_cast<bf16x8*>(p) = hi;
    *reinterpret_cast<bf16x8*>(p + 64) = lo;
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

// split-bf16 storage of 4 consecutive values: hi plane at p, lo plane lo_off elements behind it (the head-split q | k | v^T of
// the DiT qkv projection in the bf16x3 mode: the attention kernel then loads operand halves instead of splitting in registers)
__device__ __forceinline__ f32x4 store4_split(__bf16* p, size_t lo_off, f32x4 v) {
    bf16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
    *reinterpret_cast<bf16x4*>(p) = hi;
    *reinterpret_cast<bf16x4*>(p + lo_off) = lo;
    return v;
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
