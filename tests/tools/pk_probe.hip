// Synthetic probe for the co-residency wrong-result hazard.  Round 4 located the damage of the real victims
// (the SSIM kernel, the old LDS-staged linear kernels) in LANES 48..63 of plain VALU results, with or without LDS, with or without
// transcendental instructions -- in code where hipcc had formed PACKED fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
// This program runs, on one stream, a long "aggressor" kernel made of ONE instruction kind and, on a second stream, a "victim"
// kernel that repeats ONE VALU instruction form on exactly representable data (acc += 1 * 1: the result is the iteration count;
// a lost write leaves the accumulator LOW), then counts the lanes whose result is wrong, by 16-lane group.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_probe tests/tools/pk_probe.hip && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

enum { V_FMA = 0, V_PK_FMA, V_PK_ADD, V_PK_MUL, V_RCP, V_FMA_F64, V_PK_FMA_F16, V_MOV_B64, NVICT };
static const char* VN[NVICT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_rcp_f32 + s_nop 0 + v_add_f32", "v_fma_f64",
                                "v_pk_fma_f16", "v_pk_mov_b32 chain"};

// out[gid] = 1 if this lane's result is wrong; iters is a multiple of 16
template <int V>
__global__ __launch_bounds__(64) void victim(int iters, int* out, float* val) {
  const int gid = blockIdx.x * 64 + threadIdx.x;
  float expect = (float)iters, got = 0.f, got2 = (float)iters;
  const float one = 1.0f;
  if constexpr (V == V_FMA) {
    float acc = 0.f;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(one));
    }
    got = acc;
  } else if constexpr (V == V_PK_FMA) {
    f32x2 acc = {0.f, 0.f}; const f32x2 o2 = {1.f, 1.f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(o2));
    }
    got = acc[0]; got2 = acc[1];
  } else if constexpr (V == V_PK_ADD) {
    f32x2 acc = {0.f, 0.f}; const f32x2 o2 = {1.f, 1.f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(o2));
    }
    got = acc[0]; got2 = acc[1];
  } else if constexpr (V == V_PK_MUL) {
    // x <- x * 2 and x <- x * 0.5 alternate; a lost write breaks the balance.  Sum the x values: 16 iterations add 8 * (2 + 1)
    f32x2 x = {1.f, 1.f}, s = {0.f, 0.f}; const f32x2 two = {2.f, 2.f}, half = {0.5f, 0.5f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(two));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[0]) : "v"(x[0]));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(half));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[1]) : "v"(x[1]));
      }
    }
    expect = (float)iters; got = s[0]; got2 = s[1] * 2.0f;      // s0 = 8 * 2 per 16 iterations, s1 = 8 * 1
  } else if constexpr (V == V_RCP) {
    // x alternates 2 -> 0.5 -> 2 (v_rcp_f32 of a power of two is exact); acc adds x right behind the compiler's ONE wait state
    float x = 2.f, acc = 0.f;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0\n\ts_nop 0\n\tv_add_f32 %1, %1, %0" : "+v"(x), "+v"(acc));
    }
    expect = (float)iters * 1.25f; got = acc;
  } else if constexpr (V == V_FMA_F64) {
    double acc = 0.0; const double o = 1.0;
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(acc) : "v"(o));
    }
    got = (float)acc;
  } else if constexpr (V == V_PK_FMA_F16) {
    // half precision: count to 2048 at most exactly -> run the chain in blocks of 1024 and move the block into an fp32 total
    float tot = 0.f;
    for (int i = 0; i < iters; i += 1024) {
      unsigned acc = 0u; const unsigned o2 = 0x3c003c00u;       // {1.0h, 1.0h}
      const int n = iters - i < 1024 ? iters - i : 1024;
      for (int j = 0; j < n; j += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(o2));
      }
      const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(acc & 0xffff)), hi = __builtin_bit_cast(_Float16, (unsigned short)(acc >> 16));
      tot += 0.5f * ((float)lo + (float)hi);
    }
    got = tot;
  } else if constexpr (V == V_MOV_B64) {
    // a chain of 64-bit register moves carrying a counter: b <- a + 1 (fp32, low half), a <- b via v_pk_mov_b32
    f32x2 a = {0.f, 0.f}, b = {0.f, 0.f};
    for (int i = 0; i < iters; i += 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(one));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[1]) : "v"(one));
        asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(b) : "v"(a));
        asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(a) : "v"(b));
      }
    }
    got = a[0]; got2 = a[1];
  }
  out[gid] = (got != expect || got2 != expect) ? 1 : 0;
  if (val) { val[2 * gid] = got; val[2 * gid + 1] = got2; }
}

enum { A_NONE = 0, A_MFMA32, A_MFMA16, A_DSTR, A_DS128, A_DMA, A_VALU, A_MFMA32_DSTR, A_MFMA32_F32, NAGG };
static const char* AN[NAGG] = {"(quiet)", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "ds_read_b64_tr_b16", "ds_read_b128",
                               "buffer_load_dwordx4 ... lds", "v_fma_f32", "mfma 32x32x16 + ds_read_b64_tr_b16", "v_mfma_f32_32x32x2_f32"};

// 256 threads, 72 KiB of LDS, ~128 VGPRs: two workgroups per CU like the slab weight-gradient kernel, registers left for the victim
template <int A>
__global__ __launch_bounds__(256, 2) void aggressor(int iters, float* sink, const float* src) {
  __shared__ __attribute__((aligned(1024))) char smem[72 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 72 * 1024 / 4; i += 256) ((unsigned*)smem)[i] = 0x3f803f80u + i;
  __syncthreads();
  f32x16 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float keep = 0.f;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const bf16x8 fa = __builtin_bit_cast(bf16x8, (i32x4){0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80 + lane});
  if constexpr (A == A_MFMA32 || A == A_MFMA32_DSTR) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fa, acc[t], 0, 0, 0);
      if constexpr (A == A_MFMA32_DSTR) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(smem + ((i * 2048 + lane * 128) & 0xffff)));
        keep += (float)v[0];
      }
    }
  } else if constexpr (A == A_MFMA32_F32) {
    for (int i = 0; i < iters / 2; ++i) {
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, (float)lane, acc[t], 0, 0, 0);
    }
  } else if constexpr (A == A_MFMA16) {
    f32x4 a4[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) a4[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 12; ++t) a4[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fa, a4[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) keep += a4[t][0] + a4[t][3];
  } else if constexpr (A == A_DSTR) {
    for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(smem + ((i * 2048 + t * 16384 + lane * 128) & 0xffff)));
        keep += (float)v[0];
      }
    }
  } else if constexpr (A == A_DS128) {
    for (int i = 0; i < iters * 4; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 v = *(const f32x4*)(smem + ((i * 2048 + t * 16384 + lane * 32) & 0xffff));
        keep += v[0];
      }
    }
  } else if constexpr (A == A_DMA) {
    const unsigned long long ga = (unsigned long long)src;
    i32x4 r; r[0] = __builtin_amdgcn_readfirstlane((int)(ga & 0xffffffffull)); r[1] = __builtin_amdgcn_readfirstlane((int)((ga >> 32) & 0xffffull));
    r[2] = 1 << 24; r[3] = 0x00020000;
    const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)smem);
    const unsigned wbase = __builtin_amdgcn_readfirstlane(lds0 + (tid >> 6) * 16384);
    for (int i = 0; i < iters / 2; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const unsigned dst = wbase + t * 1024;
        const int voff = ((blockIdx.x * 7 + i * 8 + t) & 1023) * 16384 + lane * 16;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(r), "s"(dst) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    keep += ((float*)smem)[tid];
  } else if constexpr (A == A_VALU) {
    float a0 = (float)lane, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    for (int i = 0; i < iters * 8; ++i) {
      asm volatile("v_fma_f32 %0, %0, %4, %1\n\tv_fma_f32 %1, %1, %4, %2\n\tv_fma_f32 %2, %2, %4, %3\n\tv_fma_f32 %3, %3, %4, %0"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0.5f));
    }
    keep += a0 + a1 + a2 + a3;
  }
#pragma unroll
  for (int t = 0; t < 6; ++t) keep += acc[t][0] + acc[t][5] + acc[t][15];
  if (keep == 12345.678f) sink[0] = keep;          // keeps everything alive
}

template <int V> static void launch_victim(int blocks, int iters, int* out, float* val, hipStream_t st) {
  hipLaunchKernelGGL(victim<V>, dim3(blocks), dim3(64), 0, st, iters, out, val);
}
template <int A> static void launch_aggr(int iters, float* sink, const float* src, hipStream_t st) {
  hipLaunchKernelGGL(aggressor<A>, dim3(512), dim3(256), 0, st, iters, sink, src);
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int VBLOCKS = 4096, VITERS = 8192, AITERS = 40000;
  int* out; float* val; float* sink; float* src;
  CHECK(hipMalloc(&out, VBLOCKS * 64 * sizeof(int))); CHECK(hipMalloc(&val, VBLOCKS * 64 * 2 * sizeof(float)));
  CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&src, 1 << 24)); CHECK(hipMemset(src, 0, 1 << 24));
  std::vector<int> h(VBLOCKS * 64);
  std::vector<float> hv(VBLOCKS * 64 * 2);
  void (*vict[NVICT])(int, int, int*, float*, hipStream_t) = {launch_victim<V_FMA>, launch_victim<V_PK_FMA>, launch_victim<V_PK_ADD>, launch_victim<V_PK_MUL>,
      launch_victim<V_RCP>, launch_victim<V_FMA_F64>, launch_victim<V_PK_FMA_F16>, launch_victim<V_MOV_B64>};
  void (*aggr[NAGG])(int, float*, const float*, hipStream_t) = {nullptr, launch_aggr<A_MFMA32>, launch_aggr<A_MFMA16>, launch_aggr<A_DSTR>, launch_aggr<A_DS128>,
      launch_aggr<A_DMA>, launch_aggr<A_VALU>, launch_aggr<A_MFMA32_DSTR>, launch_aggr<A_MFMA32_F32>};
  hipEvent_t e0, e1, e2, e3;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&e3));
  printf("%-36s | %-36s | wrong lanes of %d, by 16-lane group [0-15 16-31 32-47 48-63] | aggressor ms, victim ms, first wrong value\n",
         "aggressor (stream 0)", "victim (stream 1)", VBLOCKS * 64);
  for (int a = 0; a < NAGG; ++a) {
    for (int v = 0; v < NVICT; ++v) {
      CHECK(hipMemsetAsync(out, 0xff, VBLOCKS * 64 * sizeof(int), s1));
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipEventRecord(e0, s0));
      if (aggr[a]) aggr[a](AITERS, sink, src, s0);
      CHECK(hipEventRecord(e1, s0));
      CHECK(hipEventRecord(e2, s1));
      for (int rep = 0; rep < 1; ++rep) vict[v](VBLOCKS, VITERS, out, val, s1);
      CHECK(hipEventRecord(e3, s1));
      CHECK(hipDeviceSynchronize());
      float ma = 0.f, mv = 0.f;
      CHECK(hipEventElapsedTime(&ma, e0, e1)); CHECK(hipEventElapsedTime(&mv, e2, e3));
      CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(int), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hv.data(), val, hv.size() * sizeof(float), hipMemcpyDeviceToHost));
      int grp[4] = {0, 0, 0, 0}, bad = 0, first = -1;
      for (size_t i = 0; i < h.size(); ++i) if (h[i] != 0) { ++bad; ++grp[(i & 63) >> 4]; if (first < 0) first = (int)i; }
      printf("%-36s | %-36s | %7d  [%6d %6d %6d %6d] | %7.2f %7.2f", AN[a], VN[v], bad, grp[0], grp[1], grp[2], grp[3], ma, mv);
      if (first >= 0) printf("  lane %d of wave %d: %.9g / %.9g", first & 63, first >> 6, hv[2 * first], hv[2 * first + 1]);
      printf("\n");
      fflush(stdout);
    }
  }
  return 0;
}
