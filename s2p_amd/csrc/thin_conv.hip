// Thin-Cout convolutions (Cout <= 4: the generator's 64->3 output conv, the PatchGAN 512->1 heads), bf16.
// A 32-row MFMA tile would be >90 % empty here and the GEMM has almost no parallelism in N, so these layers are
// written as direct convolutions instead:
//   fwd  : a pixel is spread over LPP lanes (one 16-byte channel chunk each); per tap one 16-B global load and
//          Cout x 4 packed bf16 dot products (v_dot2c_f32_bf16) against weights held in LDS; lanes of a pixel are
//          combined with wave shuffles; one 16-B store per pixel.
//   wgrad: a thread owns one (tap, channel chunk) pair and streams a stripe of pixels, keeping Cout x 8 fp32
//          accumulators in registers; one atomic per accumulator per workgroup at the end.
// dgrad of these layers has K = taps*8 and a wide N, which the MFMA gather kernel already handles well.
#include "s2p_common.h"

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct ThinArgs {
  const __bf16* x; const __bf16* w; const float* bias; const __bf16* dy; __bf16* y; float* dw;
  int N, H, W, Cin, x_pitch, Ho, Wo, Cout, y_pitch, KH, KW, stride, pad, reflect, act;
  float slope;
  int cin_real, M;
  int rows_per_block;
};

__device__ __forceinline__ float dot8(u32x4 a, u32x4 b, float c) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, (unsigned)a[i]), __builtin_bit_cast(bf16x2, (unsigned)b[i]), c, false);
  return c;
}

// LPP lanes per pixel (power of two >= Cin/8)
template <int LPP>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const ThinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];       // weights [Cout][T][Cin] bf16
  const int T = a.KH * a.KW;
  const int nchunk = a.Cin / 8;
  const int wbytes = a.Cout * T * a.Cin * 2;
  for (int i = threadIdx.x * 16; i < wbytes; i += 256 * 16) *(u32x4*)(smem + i) = *(const u32x4*)((const char*)a.w + i);
  __syncthreads();
  constexpr int PPW = 64 / LPP;                                      // pixels per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = lane % LPP, slot = lane / LPP;
  const bool chok = ch < nchunk;
  const int HoWo = a.Ho * a.Wo;
  for (int m0 = (blockIdx.x * 4 + wave) * PPW; m0 < a.M; m0 += gridDim.x * 4 * PPW) {
    const int m = m0 + slot;
    const bool mok = m < a.M;
    const int mm = mok ? m : 0;
    const int n = mm / HoWo, rr = mm - n * HoWo, oy = rr / a.Wo, ox = rr - oy * a.Wo;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (mok && chok) {
      const __bf16* xb = a.x + (size_t)n * a.H * a.W * a.x_pitch + ch * 8;
      int t = 0;
      for (int ky = 0; ky < a.KH; ++ky) {
        int iy = oy * a.stride + ky - a.pad;
        if (a.reflect) iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
        const bool yok = iy >= 0 && iy < a.H;
        for (int kx = 0; kx < a.KW; ++kx, ++t) {
          int ix = ox * a.stride + kx - a.pad;
          if (a.reflect) ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
          if (!yok || ix < 0 || ix >= a.W) continue;
          const u32x4 xv = *(const u32x4*)(xb + ((size_t)iy * a.W + ix) * a.x_pitch);
          const char* wp = smem + (t * a.Cin + ch * 8) * 2;
#pragma unroll
          for (int co = 0; co < 4; ++co)
            if (co < a.Cout) acc[co] = dot8(xv, *(const u32x4*)(wp + co * T * a.Cin * 2), acc[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int o = LPP / 2; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o, 64);
    if (mok && ch == 0) {
      Chunk<__bf16> c; c.raw = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < a.Cout) c.set(co, act_fwd(acc[co] + (a.bias ? a.bias[co] : 0.f), a.act, a.slope));
      *(u32x4*)(a.y + (size_t)m * a.y_pitch) = c.raw;               // channels Cout..7 are written as zeros
    }
  }
}

// thread <-> (tap, chunk); block = 256 such pairs x one stripe of GEMM pixels (output pixels of the conv)
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinArgs a) {
  const int T = a.KH * a.KW, nchunk = a.Cin / 8;
  const int pair = blockIdx.x * 256 + threadIdx.x;
  const bool pok = pair < T * nchunk;
  const int t = pok ? pair / nchunk : 0, ch = pok ? pair - t * nchunk : 0;
  const int ky = t / a.KW, kx = t - ky * a.KW;
  float acc[4][8];
#pragma unroll
  for (int co = 0; co < 4; ++co)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[co][e] = 0.f;
  const int m_begin = blockIdx.y * a.rows_per_block;
  int m_end = m_begin + a.rows_per_block; if (m_end > a.M) m_end = a.M;
  const int HoWo = a.Ho * a.Wo;
  int n = m_begin / HoWo, rr = m_begin - n * HoWo, oy = rr / a.Wo, ox = rr - oy * a.Wo;
  for (int m = m_begin; m < m_end; ++m) {
    int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
    if (a.reflect) {
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
    }
    if (pok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
      Chunk<__bf16> xv, dv;
      xv.raw = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch + ch * 8);
      dv.raw = *(const u32x4*)(a.dy + (size_t)m * a.y_pitch);          // same address for the whole block: broadcast
      float xf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xf[e] = xv.get(e);
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < a.Cout) {
          const float d = dv.get(co);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[co][e] += d * xf[e];
        }
    }
    if (++ox == a.Wo) { ox = 0; if (++oy == a.Ho) { oy = 0; ++n; } }
  }
  if (pok) {
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < a.Cout)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int ci = ch * 8 + e;
          if (ci < a.cin_real) atomicAdd(a.dw + ((size_t)co * T + t) * a.cin_real + ci, acc[co][e]);
        }
  }
}

static void fill_args(ThinArgs& a, const s2p_conv_desc* d) {
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_pitch = d->x_pitch; a.Ho = d->Ho; a.Wo = d->Wo;
  a.Cout = d->Cout; a.y_pitch = d->y_pitch; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.reflect = d->reflect; a.M = d->N * d->Ho * d->Wo;
}

bool s2p_thin_applicable(const s2p_conv_desc* d) {
  return d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->Cout <= 4 && d->Cin % 8 == 0 &&
         d->Cin / 8 <= 64 && d->y_pitch % 8 == 0 && (long long)d->Cout * d->KH * d->KW * d->Cin * 2 <= 64 * 1024;
}

int s2p_thin_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                 hipStream_t st) {
  ThinArgs a{};
  fill_args(a, d);
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y; a.act = act; a.slope = slope;
  const int nchunk = d->Cin / 8;
  int lpp = 1; while (lpp < nchunk) lpp <<= 1;
  const int ppw = 64 / lpp;
  int blocks = cdiv(a.M, 4 * ppw); if (blocks > 256 * 8) blocks = 256 * 8;
  const size_t lds = (size_t)d->Cout * d->KH * d->KW * d->Cin * 2;
#define THIN_LAUNCH(L) hipLaunchKernelGGL(thin_fwd_kernel<L>, dim3(blocks), dim3(256), lds, st, a)
  switch (lpp) {
    case 1: THIN_LAUNCH(1); break; case 2: THIN_LAUNCH(2); break; case 4: THIN_LAUNCH(4); break;
    case 8: THIN_LAUNCH(8); break; case 16: THIN_LAUNCH(16); break; case 32: THIN_LAUNCH(32); break;
    default: THIN_LAUNCH(64); break;
  }
#undef THIN_LAUNCH
  S2P_CHECK_LAUNCH("thin_fwd_kernel");
  return 0;
}

int s2p_thin_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, int cin_real, hipStream_t st) {
  ThinArgs a{};
  fill_args(a, d);
  a.x = (const __bf16*)x; a.dy = (const __bf16*)dy; a.dw = dw; a.cin_real = cin_real;
  const int pairs = d->KH * d->KW * (d->Cin / 8);
  const int gx = cdiv(pairs, 256);
  int gy = cdiv(2048, gx); if (gy > cdiv(a.M, 64)) gy = cdiv(a.M, 64); if (gy < 1) gy = 1;
  a.rows_per_block = cdiv(a.M, gy); gy = cdiv(a.M, a.rows_per_block);
  hipLaunchKernelGGL(thin_wgrad_kernel, dim3(gx, gy), dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("thin_wgrad_kernel");
  return 0;
}
