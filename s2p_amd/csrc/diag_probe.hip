// Diagnostics build only (libs2p_hip_diag.so): instruction-level probes of the co-residency hazard.
// This file is compiled WITH packed fp32 instructions enabled (the product sources are not: build.sh) -- it exists to execute them.
#include "s2p_common.h"
#ifdef S2P_DIAG_BUILD

// Diagnostics build only: single-instruction VALU victims (round 4; the same chains as tests/tools/pk_probe.hip) to be run beside the
// REAL conv kernels from Python (tests/tools/repro_valu_probe.py).  Every lane repeats one instruction form on exactly
// representable data; out[gid] = 1 where the result is wrong, val[2 gid ..] = what it got.
//   0 v_fma_f32   1 v_pk_fma_f32   2 v_pk_add_f32   3 v_pk_mul_f32   4 v_pk_fma_f32 with an SGPR-pair operand   5 v_pk_mov_b32 chain
//   6 v_mov_b64 chain   7 v_pk_fma_f16   8 v_fma_f64   9 v_pk_mul_f32 with op_sel (cross-half)   10 v_rcp_f32 + 1 wait state + use
typedef __attribute__((ext_vector_type(2))) float pr_f32x2;
template <int V>
__global__ __launch_bounds__(64) void valu_probe_kernel(int iters, int* out, float* val) {
  const int gid = blockIdx.x * 64 + threadIdx.x;
  float expect = (float)iters, got = 0.f, got2 = (float)iters;
  const float one = 1.0f;
  if constexpr (V == 0) {
    float acc = 0.f;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(one));
    }
    got = acc;
  } else if constexpr (V == 1 || V == 2 || V == 4) {
    pr_f32x2 acc = {0.f, 0.f}; const pr_f32x2 o2 = {1.f, 1.f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if constexpr (V == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(o2));
        if constexpr (V == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(o2));
        if constexpr (V == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(o2), "v"(o2));
      }
    }
    got = acc[0]; got2 = acc[1];
  } else if constexpr (V == 3 || V == 9) {
    pr_f32x2 x = {1.f, 1.f}, s = {0.f, 0.f}; const pr_f32x2 two = {2.f, 2.f}, half = {0.5f, 0.5f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (V == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(two));
        else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(two));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[0]) : "v"(x[0]));
        if constexpr (V == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(half));
        else asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(x) : "v"(half));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[1]) : "v"(x[1]));
      }
    }
    got = s[0]; got2 = s[1] * 2.0f;
  } else if constexpr (V == 5 || V == 6) {
    pr_f32x2 a = {0.f, 0.f}, b = {0.f, 0.f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(one));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[1]) : "v"(one));
        if constexpr (V == 5) {
          asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(b) : "v"(a));
          asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(a) : "v"(b));
        } else {
          asm volatile("v_mov_b64 %0, %1" : "=v"(b) : "v"(a));
          asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(b));
        }
      }
    }
    got = a[0]; got2 = a[1];
  } else if constexpr (V == 7) {
    float tot = 0.f;
    for (int i = 0; i < iters; i += 1024) {
      unsigned acc = 0u; const unsigned o2 = 0x3c003c00u;
      const int n = iters - i < 1024 ? iters - i : 1024;
      for (int j = 0; j < n; j += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(o2));
      }
      const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(acc & 0xffff)), hi = __builtin_bit_cast(_Float16, (unsigned short)(acc >> 16));
      tot += 0.5f * ((float)lo + (float)hi);
    }
    got = tot;
  } else if constexpr (V == 8) {
    double acc = 0.0; const double o = 1.0;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(acc) : "v"(o));
    }
    got = (float)acc;
  } else if constexpr (V == 10) {
    float x = 2.f, acc = 0.f;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0\n\tv_add_f32 %1, %1, %0" : "+v"(x), "+v"(acc));
    }
    expect = (float)iters * 1.25f; got = acc; got2 = expect;
  }
  out[gid] = (got != expect || got2 != expect) ? 1 : 0;
  val[2 * gid] = got; val[2 * gid + 1] = got2;
}
extern "C" __attribute__((visibility("default"))) int s2p_diag_valu_probe(int kind, int blocks, int iters, int* out, float* val, void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define PR_CASE(V) case V: hipLaunchKernelGGL(valu_probe_kernel<V>, dim3(blocks), dim3(64), 0, st, iters, out, val); break;
  switch (kind) { PR_CASE(0) PR_CASE(1) PR_CASE(2) PR_CASE(3) PR_CASE(4) PR_CASE(5) PR_CASE(6) PR_CASE(7) PR_CASE(8) PR_CASE(9) PR_CASE(10)
    default: S2P_FAIL(-1, "s2p_diag_valu_probe: unknown kind %d", kind); }
#undef PR_CASE
  S2P_CHECK_LAUNCH("valu_probe_kernel");
  return 0;
}
#endif
