// Shared device/host helpers for libs2p_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/s2p_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define S2P_WAVE 64

// ---- diagnostics are a separate build ----------------------------------------------------------------------
// Timing ablations (skip loads / skip MFMAs / clock stamps: outputs INVALID) and A/B environment switches exist only
// in libs2p_hip_diag.so (build.sh diag -> -DS2P_DIAG_BUILD), which tools/ select with S2P_LIB.  In the product
// library S2P_DIAGV() is the constant 0 (every ablation branch is compiled out of the kernels) and s2p_env_int()
// never reads the environment: a stray variable in a training job cannot change a kernel.
#ifdef S2P_DIAG_BUILD
#include <stdlib.h>
#define S2P_DIAGV(a) ((a).diag)
// run-time A/B switches of the diagnostics build (s2p_diag_set, misc.hip): tools/ab_step.py flips them between captures in ONE process
extern int s2p_diag_switch[32];
#define S2P_DIAG_SWITCH(k) (s2p_diag_switch[k])
static inline int s2p_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline int s2p_env_set(const char* name) { return getenv(name) ? 1 : 0; }
#else
#define S2P_DIAG_SWITCH(k) 0
#define S2P_DIAGV(a) 0
static inline int s2p_env_int(const char*, int dflt) { return dflt; }
static inline int s2p_env_set(const char*) { return 0; }
#endif

// ---- error plumbing ---------------------------------------------------------
void s2p_set_error(const char* fmt, ...);
#define S2P_FAIL(code, ...) do { s2p_set_error(__VA_ARGS__); return (code); } while (0)
// leaky-relu slopes above 1 are rejected by the forward entry points: the kernels compute  max(v, v * slope)  (lrelu_ns below)
#define S2P_CHECK_SLOPE(who, act, slope) do { if ((act) == S2P_ACT_LRELU && !((slope) <= 1.f)) \
    S2P_FAIL(-1, "%s: leaky-relu slope %g > 1 is not supported", who, (double)(slope)); } while (0)
#define S2P_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { s2p_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return -(int)e_ - 1000; } } while (0)

// ---- dtype traits -----------------------------------------------------------
template <typename T> struct DT;
template <> struct DT<float> { static constexpr int CE = 4; static constexpr int id = S2P_F32; };
template <> struct DT<__bf16> { static constexpr int CE = 8; static constexpr int id = S2P_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// 16-byte chunk <-> CE floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  u32x4 raw;
  // NB: bit_cast applied directly to a vector-element lvalue is miscompiled by hipcc 7.2 (always element 0):
  // go through a scalar temporary.
  __device__ __forceinline__ float get(int i) const { unsigned u = raw[i]; return __uint_as_float(u); }
  __device__ __forceinline__ void set(int i, float v) { unsigned u = __float_as_uint(v); raw[i] = u; }
  __device__ __forceinline__ void unpack(float (&v)[4]) const { for (int e = 0; e < 4; ++e) v[e] = get(e); }
  __device__ __forceinline__ void pack(const float (&v)[4]) { for (int e = 0; e < 4; ++e) set(e, v[e]); }
};
template <> struct Chunk<__bf16> {
  u32x4 raw;
  __device__ __forceinline__ float get(int i) const {
    unsigned w = raw[i >> 1];
    unsigned b = (i & 1) ? (w & 0xffff0000u) : (w << 16);
    return __builtin_bit_cast(float, b);
  }
  __device__ __forceinline__ void set(int i, float v) {
    __bf16 h = (__bf16)v;
    unsigned short u = __builtin_bit_cast(unsigned short, h);
    unsigned w = raw[i >> 1];
    raw[i >> 1] = (i & 1) ? ((w & 0x0000ffffu) | ((unsigned)u << 16)) : ((w & 0xffff0000u) | u);
  }
  // all eight at once.  pack(): one v_cvt_pk_bf16_f32 per PAIR (the same round-to-nearest-even as set(), which costs a conversion and
  // an insert per element: 1.5 VALU instructions instead of 0.5 -- round 5's instruction-mix counters)
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = get(e);
  }
  __device__ __forceinline__ void pack(const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16x2_t h = {(__bf16)v[2 * e], (__bf16)v[2 * e + 1]};
      raw[e] = __builtin_bit_cast(unsigned, h);
    }
  }
};

// ---- index arithmetic without integer division ---------------------------------------------------------------------------------
// m / d and the remainder for 0 <= m < 2^31, d > 0, from a double reciprocal and one correction step each way (the estimate m * (1/d)
// is within 2^-50 relative of the quotient, so its floor is off by at most one).  An integer division is ~25 (32-bit) to ~100 (64-bit)
// VALU instructions on gfx950; the pixel -> (image, row, column) splits of the implicit-GEMM prologues and of the element-wise
// kernels were as much VALU time as everything else those kernels do (round 5: DESIGN.md section 3.12).
__device__ __forceinline__ double s2p_rcp_f64(int d) {
  const double x = (double)d;
  double r = __builtin_amdgcn_rcp(x);                      // v_rcp_f64 is an estimate; two Newton steps reach 2^-52 from 2^-14
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ int divmod_rcp(int m, int d, double rcp, int& rem) {
  int q = (int)((double)m * rcp);
  int r = m - q * d;
  const int lo = r < 0 ? 1 : 0;                            // (selects, not branches)
  q -= lo; r += lo ? d : 0;
  const int hi = r >= d ? 1 : 0;
  q += hi; r -= hi ? d : 0;
  rem = r;
  return q;
}
// divisor of an element-wise kernel's index split: fast for indices below 2^31 (every launch of the S2P path), plain 64-bit
// division above.  `fast` is launch-uniform.
struct IdxDiv {
  int d; double r; bool fast;
  __device__ __forceinline__ IdxDiv(int d_, bool fast_) : d(d_), r(fast_ ? s2p_rcp_f64(d_) : 0.0), fast(fast_) {}
  __device__ __forceinline__ long long split(long long idx, int& rem) const {
    if (fast) return divmod_rcp((int)idx, d, r, rem);
    const long long q = idx / d; rem = (int)(idx - q * d); return q;
  }
};

// v > 0 ? v : v * ns for ns <= 1 (relu: 0, leaky relu: its slope, none: 1) as one multiply and one maximum instead of multiply,
// compare and select.  The entry points reject slopes above 1 (s2p_check_slope).
// (the instruction itself: __builtin_fmaxf puts a canonicalising v_max_f32 x, x in front whenever it cannot prove its operand is one)
__device__ __forceinline__ float lrelu_ns(float v, float ns) {
  const float m = v * ns;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
  return r;
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
  switch (act) {
    case S2P_ACT_RELU: return v > 0.f ? v : 0.f;
    case S2P_ACT_LRELU: return v > 0.f ? v : v * slope;
    case S2P_ACT_TANH: return tanhf(v);
    case S2P_ACT_SWISH: return v / (1.f + expf(-v));        // x * sigmoid(x)  (gaussian_ensemble.py:9-11)
    default: return v;
  }
}
// derivative given the activation OUTPUT (relu / lrelu preserve sign; tanh: 1-y^2)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  switch (act) {
    case S2P_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case S2P_ACT_LRELU: return y > 0.f ? 1.f : slope;
    case S2P_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// One LDS element = one ds_read_b32: a relaxed atomic load, which the compiler may neither merge with its neighbours nor widen
// (nor feed to the SLP vectoriser as part of a packed pair).  Used by the diagnostics build's LDS-staged linear kernels and the
// weight-pack kernel's odd-pitch tile.
__device__ __forceinline__ float lds_ld(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- hand-issued LDS-DMA (buffer_load_dwordx4 ... lds) ---------------------------------------------------------
// Issued from inline asm so that hipcc does not see the LDS write: with the builtin, hipcc inserts `s_waitcnt vmcnt(0)`
// before the next LDS read of the staged array, which serialises a multi-stage pipeline.  The caller counts the
// DMAs it has in flight and places `s_waitcnt vmcnt(N)` + `s_barrier` itself (cdna_hip_programming.md 5.7).
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 s2p_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));      // stride 0
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ unsigned s2p_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)((__attribute__((address_space(3))) const char*)p);
}
// lds_dst: wave-uniform LDS byte address of this wave's 1-KiB piece; voffset: this lane's byte offset into the buffer
// (an offset >= the descriptor's size returns zeros).  M0 is written in the same statement that consumes it and is declared
// clobbered: the compiler may keep nothing of its own in M0 across a DMA.
__device__ __forceinline__ void s2p_dma16(i32x4 rsrc, unsigned lds_dst, int voffset) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :: "v"(voffset), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
#define S2P_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// thin-Cout direct convolutions (thin_conv.hip)
bool s2p_thin_applicable(const s2p_conv_desc* d);
int s2p_thin_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                 hipStream_t st);
int s2p_thin_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, int cin_real, void* ws, size_t ws_bytes, hipStream_t st);
size_t s2p_thin_wgrad_ws_bytes(const s2p_conv_desc* d, int cin_real);
// the stem's weight gradient (thin input 3 -> 64, 7x7) on the row-streaming kernel with exchanged operands (thin_conv.hip)
bool s2p_stem_wgrad_applicable(const s2p_conv_desc* d, int cin_real, int cout_real);
size_t s2p_stem_wgrad_ws_bytes(const s2p_conv_desc* d, int cin_real);
int s2p_stem_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real, void* ws, size_t ws_bytes, hipStream_t st);
size_t s2p_channel_sum_ws_bytes(int64_t pixels, int C);
int s2p_channel_sum_det(int dtype, const void* dy, int64_t pixels, int C, int pitch, float* db, void* ws, size_t ws_bytes, void* stream);
// row-streaming thin-Cout kernels (thin_rows.hip)
bool s2p_thin_rows_applicable(const s2p_conv_desc* d);
int s2p_thin_rows_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                      float slope, hipStream_t st);
bool s2p_thin_rows_dgrad_applicable(const s2p_conv_desc* d, int cout_pad);
int s2p_thin_rows_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* dx, int cout_pad, hipStream_t st);
// thin-input (Cin <= 8) forward convs (thin_rows.hip)
bool s2p_thin_cin_fwd_applicable(const s2p_conv_desc* d, int act, int epi);
int s2p_thin_cin_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                     hipStream_t st);
// 7x7 convs with <= 4 real channels on the thin side, 4-channel-pitch padded copy + 14-step MFMA kernel (thin_rows.hip)
bool s2p_thin4_fwd_applicable(const s2p_conv_desc* d, int act, int epi);
size_t s2p_thin4_fwd_ws_bytes(const s2p_conv_desc* d);
int s2p_thin4_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope, void* ws,
                  size_t ws_bytes, hipStream_t st);
bool s2p_thin4_dgrad_applicable(const s2p_conv_desc* d, int cout_pad);
size_t s2p_thin4_dgrad_ws_bytes(const s2p_conv_desc* d);
int s2p_thin4_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* dx, int cout_pad, void* ws, size_t ws_bytes, hipStream_t st);
// PatchGAN logit heads, Cout = 1 (wgrad_head.hip)
// wgrad_slabg.hip: padded-raster weight gradient of the strided / 4x4 convolutions (fixed-order partial reduce; needs its workspace)
bool s2p_wgrad_slabg_supported(const s2p_conv_desc* d, int cin_real, int cout_real);
size_t s2p_wgrad_slabg_workspace(const s2p_conv_desc* d, int cin_real, int cout_real);
int s2p_wgrad_slabg(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real, int cout_real,
                    void* workspace, size_t workspace_bytes, hipStream_t st);
bool s2p_head_wgrad_supported(const s2p_conv_desc* d, int cin_real, int cout_real);
size_t s2p_head_wgrad_workspace(const s2p_conv_desc* d);
bool s2p_head_fwd_applicable(const s2p_conv_desc* d);
int s2p_head_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                 hipStream_t st);
int s2p_head_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real,
                   void* workspace, size_t workspace_bytes, hipStream_t st);


// out[i] += sum_b part[b * stride + i]  (b < nblk, i < n) in a FIXED order: wave w of the 16 adds the blocks w, w + 16, ...
// (64 consecutive i per workgroup: coalesced), then wave 0 adds the 16 wave sums in wave order -- bitwise reproducible, and
// the loads of a wave are independent (a single thread summing 200..500 partials one after the other took 40-50 us).
static __global__ __launch_bounds__(1024) void s2p_partial_reduce_kernel(const float* part, int nblk, long long stride, int n, float* out) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n) {
    int b = w;
    for (; b + 48 < nblk; b += 64) {                      // four loads in flight
      const float v0 = part[(long long)b * stride + i], v1 = part[(long long)(b + 16) * stride + i];
      const float v2 = part[(long long)(b + 32) * stride + i], v3 = part[(long long)(b + 48) * stride + i];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < nblk; b += 16) s += part[(long long)b * stride + i];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    out[i] += t;
  }
}
static inline void s2p_partial_reduce(const float* part, int nblk, long long stride, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(s2p_partial_reduce_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, st, part, nblk, stride, n, out);
}
