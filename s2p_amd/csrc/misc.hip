// Small HBM/latency-bound kernels of the S2P path: positional encoding, pooling, nearest resize, layout
// conversion, loss reductions (+ gradient seeds), fused Adam, weight packing, reflect-pad adjoint.
#include "s2p_common.h"
#include <string.h>

// ---- error plumbing ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void s2p_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* s2p_last_error(void) { return g_err; }
extern "C" int s2p_version(void) { return S2P_VERSION; }

static inline int grid_for(long long total, int cap = 4096) {
  long long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b;
}
#define GRID_STRIDE(idx, total) \
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (total); idx += (long long)gridDim.x * 256)

// ---- positional encoding ---------------------------------------------------------------------------
__global__ void posenc_kernel(const float* s, int N, int S, int L, float* out, int pitch) {
  long long total = (long long)N * pitch;
  GRID_STRIDE(idx, total) {
    int n = (int)(idx / pitch), j = (int)(idx - (long long)n * pitch);
    float v = 0.f;
    if (j < S) v = s[n * S + j];
    else if (j < S * (1 + 2 * L)) {
      int b = (j - S) / S, i = (j - S) - b * S;   // block b: 2k -> sin, 2k+1 -> cos
      int k = b >> 1;
      float arg = s[n * S + i] * (float)(1 << k);
      v = (b & 1) ? cosf(arg) : sinf(arg);
    }
    out[idx] = v;
  }
}
extern "C" int s2p_posenc_fwd(const float* state, int N, int S, int L, float* out, int out_pitch, void* stream) {
  if (!state || !out) S2P_FAIL(-1, "s2p_posenc_fwd: null pointer");
  if (out_pitch < S * (1 + 2 * L) || L > 30) S2P_FAIL(-1, "s2p_posenc_fwd: bad pitch/L");
  hipLaunchKernelGGL(posenc_kernel, dim3(grid_for((long long)N * out_pitch)), dim3(256), 0, (hipStream_t)stream,
                     state, N, S, L, out, out_pitch);
  S2P_CHECK_LAUNCH("posenc_kernel");
  return 0;
}

// ---- pooling -----------------------------------------------------------------------------------------
// One thread per (pixel, 16-byte channel chunk) when C is a whole number of chunks (the NHWC pitch of the image tensors: 8), else per
// element (CH = 1): the window geometry and the index split are shared by the chunk's channels (round 1 ran one thread per element).
template <typename T, int CH>
__global__ void avgpool_fwd_kernel(const T* x, int N, int H, int W, int C, T* y, int Ho, int Wo) {
  const int cpr = C / CH;
  long long total = (long long)N * Ho * Wo * cpr;
  const bool fast = total < (1ll << 31);
  const IdxDiv dC(cpr, fast), dW(Wo, fast), dH(Ho, fast);
  GRID_STRIDE(idx, total) {
    int c, ox, oy;
    long long p = dC.split(idx, c);
    p = dW.split(p, ox); int n = (int)dH.split(p, oy);
    float s[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) s[e] = 0.f;
    int cnt = 0;
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky; if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx; if (ix < 0 || ix >= W) continue;
        const T* src = x + (((size_t)n * H + iy) * W + ix) * C + c * CH;
        if constexpr (CH == 1) s[0] += to_f32(src[0]);
        else { Chunk<T> v; v.raw = *(const u32x4*)src;
#pragma unroll
          for (int e = 0; e < CH; ++e) s[e] += v.get(e); }
        ++cnt;
      }
    }
    const float fc = (float)cnt;
    if constexpr (CH == 1) y[idx] = from_f32<T>(s[0] / fc);
    else { Chunk<T> o;
#pragma unroll
      for (int e = 0; e < CH; ++e) s[e] = s[e] / fc;
      o.pack(s); *(u32x4*)(y + (size_t)idx * CH) = o.raw; }
  }
}
template <typename T, int CH>
__global__ void avgpool_bwd_kernel(const T* dy, int N, int H, int W, int C, T* dx, int Ho, int Wo, int accumulate) {
  const int cpr = C / CH;
  long long total = (long long)N * H * W * cpr;
  const bool fast = total < (1ll << 31);
  const IdxDiv dC(cpr, fast), dW(W, fast), dH(H, fast);
  GRID_STRIDE(idx, total) {
    int c, ix, iy;
    long long p = dC.split(idx, c);
    p = dW.split(p, ix); int n = (int)dH.split(p, iy);
    float s[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) s[e] = 0.f;
    for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {     // oy*2-1 <= iy <= oy*2+1
      if (oy < 0 || oy >= Ho) continue;
      int y0 = oy * 2 - 1, cy = (y0 < 0 ? 2 : 3) - ((y0 + 2 >= H) ? (y0 + 3 - H) : 0);
      for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        int x0 = ox * 2 - 1, cx = (x0 < 0 ? 2 : 3) - ((x0 + 2 >= W) ? (x0 + 3 - W) : 0);
        const float fc = (float)(cy * cx);
        const T* src = dy + (((size_t)n * Ho + oy) * Wo + ox) * C + c * CH;
        if constexpr (CH == 1) s[0] += to_f32(src[0]) / fc;
        else { Chunk<T> v; v.raw = *(const u32x4*)src;
#pragma unroll
          for (int e = 0; e < CH; ++e) s[e] += v.get(e) / fc; }
      }
    }
    if constexpr (CH == 1) {
      if (accumulate) s[0] += to_f32(dx[idx]);
      dx[idx] = from_f32<T>(s[0]);
    } else {
      Chunk<T> o;
      if (accumulate) { o.raw = *(const u32x4*)(dx + (size_t)idx * CH);
#pragma unroll
        for (int e = 0; e < CH; ++e) s[e] += o.get(e); }
      o.pack(s); *(u32x4*)(dx + (size_t)idx * CH) = o.raw;
    }
  }
}
extern "C" int s2p_avgpool3x3s2_fwd(int dtype, const void* x, int N, int H, int W, int C, void* y, void* stream) {
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int ce = dtype == S2P_F32 ? 4 : 8;
  const bool chunks = C % ce == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  long long total = (long long)N * Ho * Wo * (chunks ? C / ce : C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2P_F32) {
    if (chunks) hipLaunchKernelGGL((avgpool_fwd_kernel<float, 4>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, N, H, W, C, (float*)y, Ho, Wo);
    else hipLaunchKernelGGL((avgpool_fwd_kernel<float, 1>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, N, H, W, C, (float*)y, Ho, Wo);
  } else if (dtype == S2P_BF16) {
    if (chunks) hipLaunchKernelGGL((avgpool_fwd_kernel<__bf16, 8>), dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)x, N, H, W, C, (__bf16*)y, Ho, Wo);
    else hipLaunchKernelGGL((avgpool_fwd_kernel<__bf16, 1>), dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)x, N, H, W, C, (__bf16*)y, Ho, Wo);
  } else S2P_FAIL(-1, "s2p_avgpool3x3s2_fwd: bad dtype");
  S2P_CHECK_LAUNCH("avgpool_fwd_kernel");
  return 0;
}
extern "C" int s2p_avgpool3x3s2_bwd(int dtype, const void* dy, int N, int H, int W, int C, void* dx, int accumulate, void* stream) {
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int ce = dtype == S2P_F32 ? 4 : 8;
  const bool chunks = C % ce == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
  long long total = (long long)N * H * W * (chunks ? C / ce : C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2P_F32) {
    if (chunks) hipLaunchKernelGGL((avgpool_bwd_kernel<float, 4>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, N, H, W, C, (float*)dx, Ho, Wo, accumulate);
    else hipLaunchKernelGGL((avgpool_bwd_kernel<float, 1>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)dy, N, H, W, C, (float*)dx, Ho, Wo, accumulate);
  } else if (dtype == S2P_BF16) {
    if (chunks) hipLaunchKernelGGL((avgpool_bwd_kernel<__bf16, 8>), dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)dy, N, H, W, C, (__bf16*)dx, Ho, Wo, accumulate);
    else hipLaunchKernelGGL((avgpool_bwd_kernel<__bf16, 1>), dim3(grid_for(total)), dim3(256), 0, st, (const __bf16*)dy, N, H, W, C, (__bf16*)dx, Ho, Wo, accumulate);
  } else S2P_FAIL(-1, "s2p_avgpool3x3s2_bwd: bad dtype");
  S2P_CHECK_LAUNCH("avgpool_bwd_kernel");
  return 0;
}

// max-pool 2x2 stride 2 (floor).  16-byte chunks along C.
template <typename T>
__global__ void maxpool_fwd_kernel(const T* x, int N, int H, int W, int C, T* y, int Ho, int Wo) {
  constexpr int CE = DT<T>::CE;
  int cpr = C / CE;
  long long total = (long long)N * Ho * Wo * cpr;
  const bool fast = total < (1ll << 31);
  const IdxDiv dC(cpr, fast), dW(Wo, fast), dH(Ho, fast);
  GRID_STRIDE(idx, total) {
    int ch, ox, oy;
    long long p = dC.split(idx, ch);
    p = dW.split(p, ox); int n = (int)dH.split(p, oy);
    const T* b = x + (((size_t)n * H + oy * 2) * W + ox * 2) * C + ch * CE;
    Chunk<T> v00, v01, v10, v11, o;
    v00.raw = *(const u32x4*)b; v01.raw = *(const u32x4*)(b + C);
    v10.raw = *(const u32x4*)(b + (size_t)W * C); v11.raw = *(const u32x4*)(b + (size_t)W * C + C);
    float ov[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) ov[e] = fmaxf(fmaxf(v00.get(e), v01.get(e)), fmaxf(v10.get(e), v11.get(e)));
    o.pack(ov);
    *(u32x4*)(y + (size_t)idx * CE) = o.raw;
  }
}
// dx = route dy to the first arg-max of each window, times relu'(x) (x = the pooled tensor's input = relu output);
// rows/cols of x not covered by a window (odd H/W) get zero.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* dy, const T* x, int N, int H, int W, int C, T* dx, int Ho, int Wo) {
  constexpr int CE = DT<T>::CE;
  int cpr = C / CE;
  long long total = (long long)N * H * W * cpr;
  const bool fast = total < (1ll << 31);
  const IdxDiv dC(cpr, fast), dW(W, fast), dH(H, fast);
  GRID_STRIDE(idx, total) {
    int ch, ix, iy;
    long long p = dC.split(idx, ch);
    p = dW.split(p, ix); int n = (int)dH.split(p, iy);
    int oy = iy >> 1, ox = ix >> 1;
    Chunk<T> o; o.raw = (u32x4){0u, 0u, 0u, 0u};
    if (oy < Ho && ox < Wo) {
      const T* b = x + (((size_t)n * H + oy * 2) * W + ox * 2) * C + ch * CE;
      Chunk<T> v[4], d;
      v[0].raw = *(const u32x4*)b; v[1].raw = *(const u32x4*)(b + C);
      v[2].raw = *(const u32x4*)(b + (size_t)W * C); v[3].raw = *(const u32x4*)(b + (size_t)W * C + C);
      d.raw = *(const u32x4*)(dy + ((((size_t)n * Ho + oy) * Wo + ox) * C + ch * CE));
      const int me = (iy & 1) * 2 + (ix & 1);
      if constexpr (sizeof(T) == 2) {
        // bf16: two elements per instruction on the raw bit patterns.  Negative inputs are clamped to +0 first (signed 16-bit maximum
        // with 0) -- a window whose maximum is not positive passes no gradient anyway -- and non-negative bf16 values order like their
        // bit patterns as unsigned integers.  This position receives the gradient iff it equals the window maximum m, no EARLIER
        // position does (first arg-max, as the scan  m = v0; if (vk > m) take k  resolves ties), and m > 0.  A 16-bit lane that is
        // zero becomes the mask 0xffff through  min(z, 1) - 1.  ~14 VALU instructions per element instead of ~26 with the unpack,
        // compare / select scan and pack (the kernel is VALU-bound: round 5's instruction-mix counters).
        typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
        typedef __attribute__((ext_vector_type(2))) short s16x2;
        const u16x2 one = {1, 1}, zero = {0, 0}, ones = {0xffff, 0xffff};
        const s16x2 szero = {0, 0};
        const bool b0 = me > 0, b1 = me > 1, b2 = me > 2;              // position k lies before this one
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u16x2 a[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned w = v[k].raw[j];
            a[k] = __builtin_bit_cast(u16x2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), szero));
          }
          const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(a[0], a[1]), __builtin_elementwise_max(a[2], a[3]));
          const u16x2 mine = me == 0 ? a[0] : (me == 1 ? a[1] : (me == 2 ? a[2] : a[3]));
          const u16x2 eq = __builtin_elementwise_min(m - mine, one) - one;                     // 0xffff where mine == m
          const u16x2 d0 = b0 ? m - a[0] : ones, d1 = b1 ? m - a[1] : ones, d2 = b2 ? m - a[2] : ones;
          const u16x2 first = zero - __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_min(d0, d1), d2), one);   // 0xffff where every earlier one differs
          const u16x2 pos = zero - __builtin_elementwise_min(m, one);                         // 0xffff where m > 0
          const unsigned dw = d.raw[j];
          o.raw[j] = dw & __builtin_bit_cast(unsigned, eq) & __builtin_bit_cast(unsigned, first) & __builtin_bit_cast(unsigned, pos);
        }
      } else {
        float ov[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          float m = v[0].get(e); int am = 0;
#pragma unroll
          for (int k = 1; k < 4; ++k) { float t = v[k].get(e); if (t > m) { m = t; am = k; } }
          ov[e] = (am == me && m > 0.f) ? d.get(e) : 0.f;
        }
        o.pack(ov);
      }
    }
    *(u32x4*)(dx + (size_t)idx * CE) = o.raw;
  }
}
extern "C" int s2p_maxpool2x2_fwd(int dtype, const void* x, int N, int H, int W, int C, void* y, void* stream) {
  int Ho = H / 2, Wo = W / 2, ce = dtype == S2P_F32 ? 4 : 8;
  if (C % ce) S2P_FAIL(-1, "s2p_maxpool2x2_fwd: C must be a multiple of %d", ce);
  long long total = (long long)N * Ho * Wo * (C / ce);
  if (dtype == S2P_F32) hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, N, H, W, C, (float*)y, Ho, Wo);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, N, H, W, C, (__bf16*)y, Ho, Wo);
  else S2P_FAIL(-1, "s2p_maxpool2x2_fwd: bad dtype");
  S2P_CHECK_LAUNCH("maxpool_fwd_kernel");
  return 0;
}
extern "C" int s2p_maxpool2x2_bwd(int dtype, const void* dy, const void* x, int N, int H, int W, int C, void* dx, void* stream) {
  int Ho = H / 2, Wo = W / 2, ce = dtype == S2P_F32 ? 4 : 8;
  if (C % ce) S2P_FAIL(-1, "s2p_maxpool2x2_bwd: C must be a multiple of %d", ce);
  long long total = (long long)N * H * W * (C / ce);
  if (dtype == S2P_F32) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, N, H, W, C, (float*)dx, Ho, Wo);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (const __bf16*)x, N, H, W, C, (__bf16*)dx, Ho, Wo);
  else S2P_FAIL(-1, "s2p_maxpool2x2_bwd: bad dtype");
  S2P_CHECK_LAUNCH("maxpool_bwd_kernel");
  return 0;
}

// ---- nearest resize (F.interpolate mode='nearest': src = floor(dst * in/out)) -------------------------
template <typename T>
__global__ void resize_kernel(const T* x, int N, int H, int W, int C, T* y, int Ho, int Wo) {
  long long total = (long long)N * Ho * Wo * C;
  float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  GRID_STRIDE(idx, total) {
    int c = (int)(idx % C); long long p = idx / C;
    int ox = (int)(p % Wo); p /= Wo; int oy = (int)(p % Ho); int n = (int)(p / Ho);
    int iy = (int)floorf(oy * sy), ix = (int)floorf(ox * sx);
    if (iy > H - 1) iy = H - 1; if (ix > W - 1) ix = W - 1;
    y[idx] = x[(((size_t)n * H + iy) * W + ix) * C + c];
  }
}
extern "C" int s2p_resize_nearest(int dtype, const void* x, int N, int H, int W, int C, void* y, int Ho, int Wo, void* stream) {
  long long total = (long long)N * Ho * Wo * C;
  if (dtype == S2P_F32) hipLaunchKernelGGL(resize_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, N, H, W, C, (float*)y, Ho, Wo);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(resize_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, N, H, W, C, (__bf16*)y, Ho, Wo);
  else S2P_FAIL(-1, "s2p_resize_nearest: bad dtype");
  S2P_CHECK_LAUNCH("resize_kernel");
  return 0;
}

// ---- layout ------------------------------------------------------------------------------------------
// One thread per PIXEL: the C plane reads of consecutive threads are consecutive floats, the pixel's channels leave as whole 16-byte
// chunks (zero_pad: every chunk of the pitch; otherwise the C channels at c_off, element by element).  Round 1 ran one thread per
// ELEMENT with two 64-bit divisions each: 9 launches a step, 10 us each, VALU-bound.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, int N, int C, int H, int W, T* y, int pitch, int c_off, int zero_pad) {
  constexpr int CE = DT<T>::CE;
  const int HW = H * W;
  const long long total = (long long)N * HW;
  const IdxDiv dP(HW, total < (1ll << 31));
  GRID_STRIDE(p, total) {
    int hw;
    const long long n = dP.split(p, hw);
    const float* xp = x + (size_t)n * C * HW + hw;
    T* yp = y + (size_t)p * pitch;
    if (zero_pad && pitch % CE == 0) {
      for (int cb = 0; cb < pitch; cb += CE) {
        float v[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) { const int c = cb + e - c_off; v[e] = (c >= 0 && c < C) ? xp[(size_t)c * HW] : 0.f; }
        Chunk<T> o; o.pack(v);
        *(u32x4*)(yp + cb) = o.raw;
      }
    } else if (zero_pad) {
      for (int j = 0; j < pitch; ++j) { const int c = j - c_off; yp[j] = from_f32<T>((c >= 0 && c < C) ? xp[(size_t)c * HW] : 0.f); }
    } else {
      for (int j = 0; j < C; ++j) yp[c_off + j] = from_f32<T>(xp[(size_t)j * HW]);
    }
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, int pitch, int c_off, int N, int C, int H, int W, float* y, int accumulate) {
  long long total = (long long)N * C * H * W;
  const bool fast = total < (1ll << 31);
  const IdxDiv dP(H * W, fast), dC(C, fast);
  GRID_STRIDE(idx, total) {
    int hw, c;
    long long q = dP.split(idx, hw);
    int n = (int)dC.split(q, c);
    float v = to_f32(x[((size_t)n * H * W + hw) * pitch + c_off + c]);
    y[idx] = accumulate ? y[idx] + v : v;
  }
}
extern "C" int s2p_nchw_to_nhwc(int dtype, const float* x, int N, int C, int H, int W, void* y, int y_pitch, int c_off, int zero_pad, void* stream) {
  if (c_off + C > y_pitch) S2P_FAIL(-1, "s2p_nchw_to_nhwc: channels exceed pitch");
  long long total = (long long)N * H * W;                                   // one thread per pixel
  if (dtype == S2P_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, (float*)y, y_pitch, c_off, zero_pad);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, (__bf16*)y, y_pitch, c_off, zero_pad);
  else S2P_FAIL(-1, "s2p_nchw_to_nhwc: bad dtype");
  S2P_CHECK_LAUNCH("nchw_to_nhwc_kernel");
  return 0;
}
extern "C" int s2p_nhwc_to_nchw(int dtype, const void* x, int x_pitch, int c_off, int N, int C, int H, int W, float* y, int accumulate, void* stream) {
  long long total = (long long)N * C * H * W;
  if (dtype == S2P_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_pitch, c_off, N, C, H, W, y, accumulate);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, x_pitch, c_off, N, C, H, W, y, accumulate);
  else S2P_FAIL(-1, "s2p_nhwc_to_nchw: bad dtype");
  S2P_CHECK_LAUNCH("nhwc_to_nchw_kernel");
  return 0;
}

template <typename S, typename D>
__global__ void cast_kernel(const S* s, D* d, long long n) {
  GRID_STRIDE(idx, n) d[idx] = from_f32<D>(to_f32(s[idx]));
}
extern "C" int s2p_cast(int sd, const void* src, int dd, void* dst, int64_t n, void* stream) {
  dim3 g(grid_for(n)), b(256); hipStream_t st = (hipStream_t)stream;
  if (sd == S2P_F32 && dd == S2P_BF16) hipLaunchKernelGGL((cast_kernel<float, __bf16>), g, b, 0, st, (const float*)src, (__bf16*)dst, (long long)n);
  else if (sd == S2P_BF16 && dd == S2P_F32) hipLaunchKernelGGL((cast_kernel<__bf16, float>), g, b, 0, st, (const __bf16*)src, (float*)dst, (long long)n);
  else if (sd == S2P_F32 && dd == S2P_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, (long long)n);
  else if (sd == S2P_BF16 && dd == S2P_BF16) hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), g, b, 0, st, (const __bf16*)src, (__bf16*)dst, (long long)n);
  else S2P_FAIL(-1, "s2p_cast: bad dtype");
  S2P_CHECK_LAUNCH("cast_kernel");
  return 0;
}

// ---- reflect-pad adjoint ---------------------------------------------------------------------------------
template <typename T>
__global__ void reflect_fold_kernel(const T* dxp, int N, int H, int W, int C, int pad, T* dx) {
  // one 16-byte channel chunk of one interior pixel per thread: sums the (up to 3 x 3) padded positions that
  // reflect onto it.  C is the NHWC pitch, a multiple of the chunk size.
  constexpr int CE = DT<T>::CE;
  const int cpr = C / CE;
  const long long total = (long long)N * H * W * cpr;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const bool fast = total < (1ll << 31);
  const IdxDiv dC(cpr, fast), dW(W, fast), dH(H, fast);
  GRID_STRIDE(idx, total) {
    int ch, x, y;
    long long p = dC.split(idx, ch);
    p = dW.split(p, x); const int n = (int)dH.split(p, y);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + pad; if (y >= 1 && y <= pad) ys[ny++] = pad - y; if (y <= H - 2 && y >= H - 1 - pad) ys[ny++] = 2 * H - 2 - y + pad;
    xs[nx++] = x + pad; if (x >= 1 && x <= pad) xs[nx++] = pad - x; if (x <= W - 2 && x >= W - 1 - pad) xs[nx++] = 2 * W - 2 - x + pad;
    float s[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) s[e] = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        Chunk<T> v; v.raw = *(const u32x4*)(dxp + (((size_t)n * Hp + ys[a]) * Wp + xs[b]) * C + ch * CE);
#pragma unroll
        for (int e = 0; e < CE; ++e) s[e] += v.get(e);
      }
    Chunk<T> o;
    o.pack(s);
    *(u32x4*)(dx + (size_t)idx * CE) = o.raw;
  }
}
extern "C" int s2p_reflect_pad_bwd(int dtype, const void* dxp, int N, int H, int W, int C, int pad, void* dx, void* stream) {
  const int ce = dtype == S2P_F32 ? 4 : 8;
  if (C % ce) S2P_FAIL(-1, "s2p_reflect_pad_bwd: C (the NHWC pitch) must be a multiple of %d", ce);
  long long total = (long long)N * H * W * (C / ce);
  if (dtype == S2P_F32) hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dxp, N, H, W, C, pad, (float*)dx);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(reflect_fold_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dxp, N, H, W, C, pad, (__bf16*)dx);
  else S2P_FAIL(-1, "s2p_reflect_pad_bwd: bad dtype");
  S2P_CHECK_LAUNCH("reflect_fold_kernel");
  return 0;
}

// ---- losses ------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_atomic_add(float v, float* out) {
  __shared__ float part[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}
// 16-byte chunks when every pointer is 16-byte aligned (the NHWC activations always are), scalar tail otherwise
__device__ __forceinline__ bool aligned16(const void* p) { return ((unsigned long long)p & 15ull) == 0ull; }
template <typename T>
__global__ void l1_loss_kernel(const T* a, const T* b, long long count, float scale, float* loss, T* grad, int accumulate) {
  constexpr int CE = DT<T>::CE;
  const bool vec = aligned16(a) && aligned16(b) && (!grad || aligned16(grad));
  const long long nch = vec ? count / CE : 0;
  float s = 0.f;
  GRID_STRIDE(ci, nch) {
    Chunk<T> av, bv, gv;
    av.raw = *(const u32x4*)(a + ci * CE); bv.raw = *(const u32x4*)(b + ci * CE);
    if (grad && accumulate) gv.raw = *(const u32x4*)(grad + ci * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float d = av.get(e) - bv.get(e);
      s += fabsf(d);
      float gsgn = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
      if (grad && accumulate) gsgn += gv.get(e);
      gv.set(e, gsgn);
    }
    if (grad) *(u32x4*)(grad + ci * CE) = gv.raw;
  }
  for (long long idx = nch * CE + (long long)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (long long)gridDim.x * 256) {
    float d = to_f32(a[idx]) - to_f32(b[idx]);
    s += fabsf(d);
    if (grad) {
      float gsgn = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
      if (accumulate) gsgn += to_f32(grad[idx]);
      grad[idx] = from_f32<T>(gsgn);
    }
  }
  block_atomic_add(s * scale, loss);
}
extern "C" int s2p_l1_loss(int dtype, const void* a, const void* b, int64_t count, float scale, float* loss_out, void* grad_a, int accumulate, void* stream) {
  if (!a || !b || !loss_out) S2P_FAIL(-1, "s2p_l1_loss: null pointer");
  // every workgroup ends in ONE atomicAdd on the same loss word, and same-address atomics retire at ~13 ns each on
  // MI355X (2048 workgroups = a 29 us floor): 512 workgroups of 16-byte loads still saturate HBM
  static const int l1_cap = s2p_env_int("S2P_L1_BLOCKS", 512);
  dim3 g(grid_for(count / 4, l1_cap));
  if (dtype == S2P_F32) hipLaunchKernelGGL(l1_loss_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (long long)count, scale, loss_out, (float*)grad_a, accumulate);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(l1_loss_kernel<__bf16>, g, dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b, (long long)count, scale, loss_out, (__bf16*)grad_a, accumulate);
  else S2P_FAIL(-1, "s2p_l1_loss: bad dtype");
  S2P_CHECK_LAUNCH("l1_loss_kernel");
  return 0;
}
// several L1 terms in ONE launch (the 8 feature-matching maps of the two PatchGAN scales, the 5 VGG taps): blockIdx.y = job.
// Every job needs 16-byte aligned pointers and a count that is a multiple of the chunk (NHWC activations are both).
// The jobs differ by up to 35x in size (VGG relu1_1 .. relu5_1): each gets workgroups in proportion to its size (first[j] =
// its first workgroup of the 1-D grid), and a thread keeps four chunk pairs in flight.
struct L1Multi { const void* a[S2P_L1_MAX_JOBS]; const void* b[S2P_L1_MAX_JOBS]; void* g[S2P_L1_MAX_JOBS];
                 long long count[S2P_L1_MAX_JOBS]; float scale[S2P_L1_MAX_JOBS]; float* loss[S2P_L1_MAX_JOBS];
                 int first[S2P_L1_MAX_JOBS + 1]; int n_jobs; };
template <typename T>
__global__ __launch_bounds__(256) void l1_multi_kernel(const L1Multi m) {
  constexpr int CE = DT<T>::CE;
  int j = 0;
  while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
  const int blk = blockIdx.x - m.first[j], nblk = m.first[j + 1] - m.first[j];
  const T* a = (const T*)m.a[j]; const T* b = (const T*)m.b[j]; T* grad = (T*)m.g[j];
  const long long nch = m.count[j] / CE;
  const float scale = m.scale[j];
  const long long stride = (long long)nblk * 256;
  float s = 0.f;
  for (long long c0 = (long long)blk * 256 + threadIdx.x; c0 < nch; c0 += 4 * stride) {
    Chunk<T> av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long ci = c0 + u * stride;
      av[u].raw = (u32x4){0u, 0u, 0u, 0u}; bv[u].raw = av[u].raw;
      if (ci < nch) { av[u].raw = *(const u32x4*)(a + ci * CE); bv[u].raw = *(const u32x4*)(b + ci * CE); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long ci = c0 + u * stride;
      if (ci >= nch) break;
      Chunk<T> gv;
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const float d = av[u].get(e) - bv[u].get(e);
        s += fabsf(d);
        gv.set(e, d > 0.f ? scale : (d < 0.f ? -scale : 0.f));
      }
      if (grad) *(u32x4*)(grad + ci * CE) = gv.raw;
    }
  }
  block_atomic_add(s * scale, m.loss[j]);
}
extern "C" int s2p_l1_loss_multi(int dtype, const s2p_l1_job* jobs, int n_jobs, void* stream) {
  if (!jobs || n_jobs < 1 || n_jobs > S2P_L1_MAX_JOBS) S2P_FAIL(-1, "s2p_l1_loss_multi: 1..%d jobs", S2P_L1_MAX_JOBS);
  const int ce = dtype == S2P_F32 ? 4 : 8;
  L1Multi m{};
  m.n_jobs = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const s2p_l1_job& q = jobs[j];
    if (!q.a || !q.b || !q.loss_out || q.count <= 0) S2P_FAIL(-1, "s2p_l1_loss_multi: job %d: null pointer / empty", j);
    if (((unsigned long long)q.a | (unsigned long long)q.b | (unsigned long long)q.grad_a) & 15ull || q.count % ce)
      S2P_FAIL(-1, "s2p_l1_loss_multi: job %d: pointers must be 16-byte aligned and count a multiple of %d", j, ce);
    m.a[j] = q.a; m.b[j] = q.b; m.g[j] = q.grad_a; m.count[j] = q.count; m.scale[j] = q.scale; m.loss[j] = q.loss_out;
    // one workgroup per 256 threads x 4 chunks x 4 rounds; same-address atomics retire at ~13 ns each: <= 512 workgroups per job
    static const int cap = s2p_env_int("S2P_L1_BLOCKS", 512);
    long long nb = (q.count / ce + 4095) / 4096; if (nb > cap) nb = cap; if (nb < 1) nb = 1;
    m.first[j + 1] = m.first[j] + (int)nb;
  }
  dim3 g(m.first[n_jobs]);
  if (dtype == S2P_F32) hipLaunchKernelGGL(l1_multi_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, m);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(l1_multi_kernel<__bf16>, g, dim3(256), 0, (hipStream_t)stream, m);
  else S2P_FAIL(-1, "s2p_l1_loss_multi: bad dtype");
  S2P_CHECK_LAUNCH("l1_multi_kernel");
  return 0;
}
template <typename T>
__global__ void hinge_kernel(const T* x, long long count, int mode, float scale, float* loss, T* grad) {
  float s = 0.f;
  GRID_STRIDE(idx, count) {
    const float v = to_f32(x[idx]);
    const float sgn = mode == 0 ? 1.f : -1.f;           // d(1 + sgn*v)/dv
    const float t = 1.f + sgn * v;
    const bool on = t > 0.f;
    const float l = mode == 2 ? -v : (on ? t : 0.f);
    const float gr = mode == 2 ? -scale : (on ? sgn * scale : 0.f);
    s += l;
    if (grad) grad[idx] = from_f32<T>(gr);
  }
  block_atomic_add(s * scale, loss);
}
extern "C" int s2p_hinge_loss(int dtype, const void* x, int64_t count, int mode, float scale, float* loss_out, void* grad_x, void* stream) {
  if (!x || !loss_out || mode < 0 || mode > 2) S2P_FAIL(-1, "s2p_hinge_loss: bad argument");
  dim3 g(grid_for(count, 256));                         // same-address atomic per workgroup: keep the grid small
  if (dtype == S2P_F32) hipLaunchKernelGGL(hinge_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long long)count, mode, scale, loss_out, (float*)grad_x);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(hinge_kernel<__bf16>, g, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (long long)count, mode, scale, loss_out, (__bf16*)grad_x);
  else S2P_FAIL(-1, "s2p_hinge_loss: bad dtype");
  S2P_CHECK_LAUNCH("hinge_kernel");
  return 0;
}

// hinge terms straight on an NHWC logit map [pixels][pitch] whose channel 0 is the logit (the PatchGAN heads write pitch 8):
// grad_x gets the gradient in channel 0 and zeros in the padding channels -- no NHWC -> dense -> NHWC round trip
template <typename T>
__global__ void hinge_strided_kernel(const T* x, long long pixels, int pitch, int mode, float scale, float* loss, T* grad) {
  constexpr int CE = DT<T>::CE;
  float s = 0.f;
  GRID_STRIDE(p, pixels) {
    const float v = to_f32(x[p * pitch]);
    const float sgn = mode == 0 ? 1.f : -1.f;
    const float t = 1.f + sgn * v;
    const bool on = t > 0.f;
    s += mode == 2 ? -v : (on ? t : 0.f);
    if (grad) {
      const float gr = mode == 2 ? -scale : (on ? sgn * scale : 0.f);
      for (int c0 = 0; c0 < pitch; c0 += CE) {
        Chunk<T> c; c.raw = (u32x4){0u, 0u, 0u, 0u};
        if (c0 == 0) c.set(0, gr);
        *(u32x4*)(grad + p * pitch + c0) = c.raw;
      }
    }
  }
  block_atomic_add(s * scale, loss);
}
extern "C" int s2p_hinge_loss_strided(int dtype, const void* x, int64_t pixels, int pitch, int mode, float scale, float* loss_out,
                                      void* grad_x, void* stream) {
  if (!x || !loss_out || mode < 0 || mode > 2) S2P_FAIL(-1, "s2p_hinge_loss_strided: bad argument");
  const int ce = dtype == S2P_F32 ? 4 : 8;
  if (pitch < ce || pitch % ce) S2P_FAIL(-1, "s2p_hinge_loss_strided: pitch must be a multiple of %d", ce);
  dim3 g(grid_for(pixels, 256));                        // same-address atomic per workgroup: keep the grid small
  if (dtype == S2P_F32) hipLaunchKernelGGL(hinge_strided_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long long)pixels, pitch, mode, scale, loss_out, (float*)grad_x);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(hinge_strided_kernel<__bf16>, g, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (long long)pixels, pitch, mode, scale, loss_out, (__bf16*)grad_x);
  else S2P_FAIL(-1, "s2p_hinge_loss_strided: bad dtype");
  S2P_CHECK_LAUNCH("hinge_strided_kernel");
  return 0;
}

// ---- fused Adam over a flat fp32 buffer (28 B / parameter of HBM traffic) ----------------------------------
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n4, long long n, float lr_bc1,
                            float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
  GRID_STRIDE(idx, n4) {
    long long i = idx * 4;
    if (i + 4 <= n) {
      f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gg = gv[e] * gscale;
        mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
        vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
        pv[e] -= lr_bc1 * mv[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
      }
      *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
    } else {
      for (long long k = i; k < n; ++k) {
        float gg = g[k] * gscale;
        m[k] = beta1 * m[k] + (1.f - beta1) * gg;
        v[k] = beta2 * v[k] + (1.f - beta2) * gg * gg;
        p[k] -= lr_bc1 * m[k] / (sqrtf(v[k]) * inv_sqrt_bc2 + eps);
      }
    }
  }
}
extern "C" int s2p_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || step < 1) S2P_FAIL(-1, "s2p_adam_step: bad argument");
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) S2P_FAIL(-1, "s2p_adam_step: buffers must be 16-byte aligned");
  double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  long long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, (long long)n,
                     (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  S2P_CHECK_LAUNCH("adam_kernel");
  return 0;
}

// graph-capturable variant: the step counter lives in device memory (a captured hipGraph replays the same
// kernel arguments, so the bias corrections must be derived on the device).
__global__ void adam_tick_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n4, long long n, float lr,
                                float beta1, float beta2, float eps, const int* step, float gscale) {
  // the bias corrections once per workgroup (wave 0, through LDS): two powf calls are ~300 VALU instructions, and with one
  // 16-byte group per thread every wave of the launch used to run them -- a third of the kernel's instructions (round 5)
  __shared__ float sc[2];
  if (threadIdx.x < 64) {
    const int t = *step;
    const float bc1 = 1.f - powf(beta1, (float)t), bc2 = 1.f - powf(beta2, (float)t);
    if (threadIdx.x == 0) { sc[0] = lr / bc1; sc[1] = rsqrtf(bc2); }
  }
  __syncthreads();
  const float lr_bc1 = sc[0], inv_sqrt_bc2 = sc[1];
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  // p -= lr_bc1 * m / (sqrt(v) * inv_sqrt_bc2 + eps) with the hardware square root and reciprocal (1 ulp each; the update is
  // ~1e-4 of the parameter, so the parameter moves by < 1e-11 relative against the correctly rounded forms, which cost ~25 VALU
  // instructions per element more)
  auto upd = [&](float gg0, float& mm, float& vv, float& pp) {
    const float gg = gg0 * gscale;
    mm = __builtin_fmaf(beta1, mm, omb1 * gg);
    vv = __builtin_fmaf(beta2, vv, omb2 * gg * gg);
    const float den = __builtin_fmaf(__builtin_amdgcn_sqrtf(vv), inv_sqrt_bc2, eps);
    pp = __builtin_fmaf(-lr_bc1 * mm, __builtin_amdgcn_rcpf(den), pp);
  };
  GRID_STRIDE(idx, n4) {
    long long i = idx * 4;
    if (i + 4 <= n) {                       // 16-byte accesses: 28 B / parameter at HBM speed (4-byte accesses ran at 2.6 TB/s)
      const f32x4 g4 = *(const f32x4*)(g + i);
      f32x4 m4 = *(const f32x4*)(m + i), v4 = *(const f32x4*)(v + i), p4 = *(const f32x4*)(p + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float mm = m4[e], vv = v4[e], pp = p4[e];
        upd(g4[e], mm, vv, pp);
        m4[e] = mm; v4[e] = vv; p4[e] = pp;
      }
      *(f32x4*)(m + i) = m4; *(f32x4*)(v + i) = v4; *(f32x4*)(p + i) = p4;
      continue;
    }
    for (long long k = i; k < n; ++k) {
      float mm = m[k], vv = v[k], pp = p[k];
      upd(g[k], mm, vv, pp);
      m[k] = mm; v[k] = vv; p[k] = pp;
    }
  }
}
extern "C" int s2p_adam_step_dev_part(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int* step_dev, float grad_scale, int tick, void* stream) {
  if (!p || !g || !m || !v || !step_dev) S2P_FAIL(-1, "s2p_adam_step_dev: bad argument");
  if (n <= 0) return 0;
  long long n4 = (n + 3) / 4;
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, (long long)n,
                     lr, beta1, beta2, eps, (const int*)step_dev, grad_scale);
  S2P_CHECK_LAUNCH("adam_dev_kernel");
  return 0;
}
extern "C" int s2p_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int* step_dev, float grad_scale, void* stream) {
  return s2p_adam_step_dev_part(p, g, m, v, n, lr, beta1, beta2, eps, step_dev, grad_scale, 1, stream);
}

// ---- weight packing: fp32 channels-last master [R][T][C] -> compute dtype, both GEMM orientations ----------
// fwd: dst_fwd[r][t][c] (row length T*Cpad, zeros for c >= C);  bwd: dst_bwd[c][t][r_off + r] (row length
// T*Rrow; rows c >= C and columns outside [r_off, r_off+R) are never written: the caller pre-zeroes them once).
// One pass over the master: per tap t, a 64 x 64 (r x c) tile is read ONCE with 16-byte loads along c (the master's contiguous
// axis); the forward operand goes out from the registers (8-byte bf16 stores along c), the transposed backward operand through a
// 64 x 65 LDS tile (8-byte stores along r).  (Rounds 1-2 read the master twice, the second time through 32 x 32 tiles with two
// barriers per 1024 elements: 69 + 58 us per step for G + D.)
template <typename T>
__device__ void pack_one(const s2p_pack_job& j, int part, int nparts, float (*tile)[65]) {
  const int tid = threadIdx.x;
  T* df = (T*)j.dst_fwd;
  T* db = (T*)j.dst_bwd;
  const int Cw = df ? (j.Cpad > j.C ? j.Cpad : j.C) : j.C;       // columns to cover (the forward operand's zero padding included)
  const int tr = (j.R + 63) / 64, tc = (Cw + 63) / 64;
  const long long ntiles = (long long)tr * tc * j.T;
  const int rr = tid >> 4, cq = tid & 15;                    // 16 rows x 16 four-column chunks per pass, 4 passes
  const bool vec_src = (j.C & 3) == 0;                       // master rows start 16-byte aligned and hold whole float4s
  const bool vec_bwd = db && (j.Rrow & 3) == 0 && (j.r_off & 3) == 0;
  for (long long q = part; q < ntiles; q += nparts) {
    const int t = (int)(q % j.T); const long long rc = q / j.T;
    const int r0 = (int)(rc % tr) * 64, c0 = (int)(rc / tr) * 64;
    if (db) __syncthreads();                                 // previous tile's transposed reads are done
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + rr + 16 * k, c = c0 + 4 * cq;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < j.R && c < j.C) {
        const float* sp = j.src + ((long long)r * j.T + t) * j.C + c;
        if (vec_src) v = *(const f32x4*)sp;
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (c + e < j.C) v[e] = sp[e];
        }
      }
      if (df && r < j.R && c < j.Cpad) {                     // Cpad is a multiple of the 16-byte chunk: c + 3 < Cpad
        T* dp = df + ((long long)r * j.T + t) * j.Cpad + c;
        if constexpr (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
          *(bf16x4_t*)dp = (bf16x4_t){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        } else {
          *(f32x4*)dp = v;
        }
      }
      if (db) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[rr + 16 * k][4 * cq + e] = v[e];
      }
    }
    if (!db) continue;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + rr + 16 * k, r = r0 + 4 * cq;       // output row c, four consecutive r
      if (c >= j.C || r >= j.R) continue;
      T* dp = db + ((long long)c * j.T + t) * j.Rrow + j.r_off + r;
      float w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = lds_ld(&tile[4 * cq + e][rr + 16 * k]);   // single-dword reads of the odd-pitch tile
      if (vec_bwd && r + 3 < j.R) {
        if constexpr (sizeof(T) == 2) {
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
          *(bf16x4_t*)dp = (bf16x4_t){(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
        } else {
          *(f32x4*)dp = (f32x4){w[0], w[1], w[2], w[3]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (r + e < j.R) dp[e] = from_f32<T>(w[e]);
      }
    }
  }
}
__global__ __launch_bounds__(256) void pack_kernel(const s2p_pack_job* jobs) {
  __shared__ float tile[64][65];
  const s2p_pack_job j = jobs[blockIdx.y];
  // the grid is sized for the largest job (one 64 x 64 tile per workgroup and pass); a smaller job uses only as many workgroups as
  // it has tiles and the rest return at once (the jobs of one network differ by 4 orders of magnitude in size)
  const int Cw = j.dst_fwd ? (j.Cpad > j.C ? j.Cpad : j.C) : j.C;
  long long need = (long long)((j.R + 63) / 64) * ((Cw + 63) / 64) * j.T;
  if (need > (long long)gridDim.x) need = gridDim.x;
  if (need < 1) need = 1;
  if ((long long)blockIdx.x >= need) return;
  if (j.dtype == S2P_F32) pack_one<float>(j, blockIdx.x, (int)need, tile);
  else pack_one<__bf16>(j, blockIdx.x, (int)need, tile);
}
extern "C" int s2p_pack_weights(const s2p_pack_job* jobs, int n_jobs, int max_elems, void* stream) {
  if (!jobs || n_jobs <= 0) S2P_FAIL(-1, "s2p_pack_weights: bad argument");
  int parts = (max_elems + 4095) / 4096; if (parts < 1) parts = 1; if (parts > 2048) parts = 2048;     // ~ one 64 x 64 tile per workgroup
  hipLaunchKernelGGL(pack_kernel, dim3(parts, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
  S2P_CHECK_LAUNCH("pack_kernel");
  return 0;
}

// ---- activation backward / scaling -----------------------------------------------------------------------
template <typename T>
__global__ void act_bwd_kernel(const T* dy, const T* y, long long n, int act, float slope, T* dx) {
  GRID_STRIDE(idx, n) dx[idx] = from_f32<T>(to_f32(dy[idx]) * act_grad_from_out(to_f32(y[idx]), act, slope));
}
extern "C" int s2p_act_bwd(int dtype, const void* dy, const void* y, int64_t n, int act, float slope, void* dx, void* stream) {
  if (dtype == S2P_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)y, (long long)n, act, slope, (float*)dx);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(act_bwd_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (const __bf16*)y, (long long)n, act, slope, (__bf16*)dx);
  else S2P_FAIL(-1, "s2p_act_bwd: bad dtype");
  S2P_CHECK_LAUNCH("act_bwd_kernel");
  return 0;
}
// x *= *scale (device scalar, fp32) -- applies an upstream scalar grad_output without a host sync
template <typename T>
__global__ void scale_kernel(T* x, long long n, const float* scale) {
  float s = *scale;
  GRID_STRIDE(idx, n) x[idx] = from_f32<T>(to_f32(x[idx]) * s);
}
extern "C" int s2p_scale(int dtype, void* x, int64_t n, const float* scale, void* stream) {
  if (dtype == S2P_F32) hipLaunchKernelGGL(scale_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (float*)x, (long long)n, scale);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(scale_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (__bf16*)x, (long long)n, scale);
  else S2P_FAIL(-1, "s2p_scale: bad dtype");
  S2P_CHECK_LAUNCH("scale_kernel");
  return 0;
}

// ---- elementwise add and channel-slice copy ---------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* out, long long n) {
  constexpr int CE = DT<T>::CE;
  const long long nch = (aligned16(a) && aligned16(b) && aligned16(out)) ? n / CE : 0;
  GRID_STRIDE(ci, nch) {
    Chunk<T> av, bv;
    av.raw = *(const u32x4*)(a + ci * CE); bv.raw = *(const u32x4*)(b + ci * CE);
#pragma unroll
    for (int e = 0; e < CE; ++e) av.set(e, av.get(e) + bv.get(e));
    *(u32x4*)(out + ci * CE) = av.raw;
  }
  for (long long idx = nch * CE + (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256)
    out[idx] = from_f32<T>(to_f32(a[idx]) + to_f32(b[idx]));
}
extern "C" int s2p_add(int dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (dtype == S2P_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)out, (long long)n);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(add_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a, (const __bf16*)b, (__bf16*)out, (long long)n);
  else S2P_FAIL(-1, "s2p_add: bad dtype");
  S2P_CHECK_LAUNCH("add_kernel");
  return 0;
}
// dst[p][dst_off + c] (+)= src[p][src_off + c], c < C, p < pixels  (torch.cat along channels / its backward slice)
template <typename T>
__global__ void copy_channels_kernel(const T* src, int sp, int so, T* dst, int dp, int d_o, int C, long long pixels, int accumulate) {
  long long total = pixels * C;
  GRID_STRIDE(idx, total) {
    long long p = idx / C; int c = (int)(idx - p * C);
    float v = to_f32(src[p * sp + so + c]);
    if (accumulate) v += to_f32(dst[p * dp + d_o + c]);
    dst[p * dp + d_o + c] = from_f32<T>(v);
  }
}
extern "C" int s2p_copy_channels(int dtype, const void* src, int src_pitch, int src_off, void* dst, int dst_pitch, int dst_off,
                                 int C, int64_t pixels, int accumulate, void* stream) {
  long long total = (long long)pixels * C;
  if (dtype == S2P_F32) hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)src, src_pitch, src_off, (float*)dst, dst_pitch, dst_off, C, (long long)pixels, accumulate);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(copy_channels_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, src_pitch, src_off, (__bf16*)dst, dst_pitch, dst_off, C, (long long)pixels, accumulate);
  else S2P_FAIL(-1, "s2p_copy_channels: bad dtype");
  S2P_CHECK_LAUNCH("copy_channels_kernel");
  return 0;
}

// ---- dataset image formats (bulk augmentation caller) --------------------------------
// uint8 NHWC [P][C] frames (rlkit/torch/slac/algo.py:189-190 reads them NHWC) <-> NHWC compute dtype in [-1,1].
//   in : v = u8 / 127.5 - 1      (zero-padded to `pitch` channels)
//   out: u8 = clamp(round((v + 1) * 127.5), 0, 255)   -- exact inverse on the 256 representable values
template <typename T>
__global__ void u8_to_nhwc_kernel(const unsigned char* x, long long pixels, int C, T* y, int pitch) {
  long long total = pixels * pitch;
  GRID_STRIDE(idx, total) {
    long long p = idx / pitch; int c = (int)(idx - p * pitch);
    float v = c < C ? (float)x[p * C + c] / 127.5f - 1.f : 0.f;      // true division: 255 -> exactly 1.0
    y[idx] = from_f32<T>(v);
  }
}
template <typename T>
__global__ void nhwc_to_u8_kernel(const T* x, int pitch, long long pixels, int C, unsigned char* y) {
  long long total = pixels * C;
  GRID_STRIDE(idx, total) {
    long long p = idx / C; int c = (int)(idx - p * C);
    float v = (to_f32(x[p * pitch + c]) + 1.f) * 127.5f;
    v = rintf(v);
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    y[idx] = (unsigned char)v;
  }
}
extern "C" int s2p_u8_to_nhwc(int dtype, const void* x, int64_t pixels, int C, void* y, int y_pitch, void* stream) {
  if (!x || !y || C > y_pitch) S2P_FAIL(-1, "s2p_u8_to_nhwc: bad argument");
  long long total = (long long)pixels * y_pitch;
  if (dtype == S2P_F32) hipLaunchKernelGGL(u8_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x, (long long)pixels, C, (float*)y, y_pitch);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(u8_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x, (long long)pixels, C, (__bf16*)y, y_pitch);
  else S2P_FAIL(-1, "s2p_u8_to_nhwc: bad dtype");
  S2P_CHECK_LAUNCH("u8_to_nhwc_kernel");
  return 0;
}
extern "C" int s2p_nhwc_to_u8(int dtype, const void* x, int x_pitch, int64_t pixels, int C, void* y, void* stream) {
  if (!x || !y || C > x_pitch) S2P_FAIL(-1, "s2p_nhwc_to_u8: bad argument");
  long long total = (long long)pixels * C;
  if (dtype == S2P_F32) hipLaunchKernelGGL(nhwc_to_u8_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, x_pitch, (long long)pixels, C, (unsigned char*)y);
  else if (dtype == S2P_BF16) hipLaunchKernelGGL(nhwc_to_u8_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, x_pitch, (long long)pixels, C, (unsigned char*)y);
  else S2P_FAIL(-1, "s2p_nhwc_to_u8: bad dtype");
  S2P_CHECK_LAUNCH("nhwc_to_u8_kernel");
  return 0;
}

// ---- ensemble dynamics head (N2) ------------------------------------------------------------------------------
// one thread per sample: E*2D raw values -> soft-clamped std, local-mode mean, member pick + de-normalisation and the
// two uncertainty reductions of state_transition_rollout.py:199-204.  E*D <= 7*32 values are kept in registers/local.
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus (threshold 20)
__global__ void ensemble_head_kernel(const float* raw, int rp, const float* xin, int xp, int B, int E, int D,
                                     const float* mn, const float* mx, float* mean, float* sd, const int* pick,
                                     const float* om, const float* os, float rm, float rs, float* nobs, float* rew,
                                     float* dis, float* ale) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int Do = D - 1;
  float avg[32];
  for (int d = 0; d < Do; ++d) avg[d] = 0.f;
  float amax = 0.f;
  const int pk = pick ? pick[b] : 0;
  for (int e = 0; e < E; ++e) {
    const float* r = raw + (size_t)b * rp + e * 2 * D;
    float s2 = 0.f;
    for (int d = 0; d < D; ++d) {
      float mu = r[d] + (d < Do ? xin[(size_t)b * xp + d] : 0.f);          // 'local' mode: obs part is a delta
      float ls = r[D + d];
      ls = mx[d] - softplus_f(mx[d] - ls);                                   // soft_clamp upper, then lower
      ls = mn[d] + softplus_f(ls - mn[d]);
      float s = expf(ls);
      s2 += s * s;
      if (d < Do) avg[d] += mu;
      if (mean) mean[((size_t)e * B + b) * D + d] = mu;
      if (sd) sd[((size_t)e * B + b) * D + d] = s;
      if (pick && e == pk) { if (d < Do) nobs[(size_t)b * Do + d] = mu * os[d] + om[d]; else rew[b] = mu * rs + rm; }
    }
    amax = fmaxf(amax, sqrtf(s2));
  }
  if (ale) ale[b] = amax;
  if (dis) {
    float dmax = 0.f;
    for (int e = 0; e < E; ++e) {
      const float* r = raw + (size_t)b * rp + e * 2 * D;
      float s2 = 0.f;
      for (int d = 0; d < Do; ++d) { float df = r[d] + xin[(size_t)b * xp + d] - avg[d] / (float)E; s2 += df * df; }
      dmax = fmaxf(dmax, sqrtf(s2));
    }
    dis[b] = dmax;
  }
}
extern "C" int s2p_ensemble_head(const float* raw, int raw_pitch, const float* xin, int x_pitch, int B, int E, int D,
                                 const float* min_logstd, const float* max_logstd, float* mean, float* std,
                                 const int32_t* pick, const float* obs_mean, const float* obs_std, float rew_mean,
                                 float rew_std, float* next_obs, float* reward, float* disagreement, float* aleatoric,
                                 void* stream) {
  if (!raw || !xin || !min_logstd || !max_logstd || D < 2 || D > 33 || E < 1) S2P_FAIL(-1, "s2p_ensemble_head: bad argument");
  if (pick && (!obs_mean || !obs_std || !next_obs || !reward)) S2P_FAIL(-1, "s2p_ensemble_head: pick needs outputs");
  hipLaunchKernelGGL(ensemble_head_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, raw, raw_pitch, xin, x_pitch,
                     B, E, D, min_logstd, max_logstd, mean, std, (const int*)pick, obs_mean, obs_std, rew_mean, rew_std,
                     next_obs, reward, disagreement, aleatoric);
  S2P_CHECK_LAUNCH("ensemble_head_kernel");
  return 0;
}

#ifdef S2P_DIAG_BUILD
// Diagnostics build only: an LDS "canary".  Every workgroup (256 threads) fills `words` dwords of LDS with a pattern, then for
// `spins` rounds sleeps and re-reads all of it with single-dword reads.  A word that no longer holds its pattern was overwritten
// by SOMEBODY ELSE (the workgroup itself never writes again): out[0] counts them, out[1 + 2 i] / out[2 + 2 i] record the
// dword offset and the value found for the first few (tests/tools/repro_canary.py runs it beside the LDS-DMA kernels).
__global__ __launch_bounds__(256) void lds_canary_kernel(int words, int spins, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned cz[];
  const int tid = threadIdx.x;
  for (int i = tid; i < words; i += 256) cz[i] = 0xA5000000u ^ (unsigned)i;
  __syncthreads();
  for (int s = 0; s < spins; ++s) {
    __builtin_amdgcn_s_sleep(64);
    for (int i = tid; i < words; i += 256) {
      const unsigned v = __hip_atomic_load(&cz[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (v != (0xA5000000u ^ (unsigned)i)) {
        const unsigned k = atomicAdd(out, 1u);
        if (k < 16) { out[1 + 2 * k] = (unsigned)i; out[2 + 2 * k] = v; }
        cz[i] = 0xA5000000u ^ (unsigned)i;               // re-arm
      }
    }
  }
}
extern "C" __attribute__((visibility("default"))) int s2p_diag_lds_canary(int blocks, int words, int spins, unsigned* out, void* stream) {
  hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(256), (size_t)words * 4, (hipStream_t)stream, words, spins, out);
  S2P_CHECK_LAUNCH("lds_canary_kernel");
  return 0;
}
// Diagnostics build only: a VGPR / EXEC "canary" (round 4).  Every wave fills NR vector registers with a lane- and register-
// dependent pattern, then for `spins` rounds sleeps and re-checks every register (each value passes through an opaque asm
// statement, so the compiler compares the REGISTER, not a recomputed constant) and the EXEC mask.  A register that no longer
// holds its pattern was written by something other than this wave's program.  out[0] counts events; the first 64 are logged
// as {register index, lane, value found, HW_ID, GPR_ALLOC, spin, EXEC_LO, EXEC_HI}.
template <int NR>
__global__ __launch_bounds__(64) void vgpr_canary_kernel(int spins, unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned r[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { r[i] = 0xC0DE0000u ^ ((unsigned)i << 8) ^ lane; asm volatile("" : "+v"(r[i])); }
  const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), gpr = __builtin_amdgcn_s_getreg((31 << 11) | 5);
  for (int s = 0; s < spins; ++s) {
    __builtin_amdgcn_s_sleep(32);
    const unsigned long long ex = __builtin_amdgcn_read_exec();
    if (ex != ~0ull && lane == (unsigned)__builtin_ctzll(ex)) {
      const unsigned k = atomicAdd(out, 1u);
      if (k < 64) { unsigned* o = out + 1 + 8 * k; o[0] = 0xffffu; o[1] = lane; o[2] = 0; o[3] = hw_id; o[4] = gpr; o[5] = (unsigned)s; o[6] = (unsigned)ex; o[7] = (unsigned)(ex >> 32); }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      asm volatile("" : "+v"(r[i]));
      const unsigned want = 0xC0DE0000u ^ ((unsigned)i << 8) ^ lane;
      if (r[i] != want) {
        const unsigned k = atomicAdd(out, 1u);
        if (k < 64) { unsigned* o = out + 1 + 8 * k; o[0] = (unsigned)i; o[1] = lane; o[2] = r[i]; o[3] = hw_id; o[4] = gpr; o[5] = (unsigned)s; o[6] = (unsigned)ex; o[7] = (unsigned)(ex >> 32); }
        r[i] = want;                                    // re-arm
        asm volatile("" : "+v"(r[i]));
      }
    }
  }
}
extern "C" __attribute__((visibility("default"))) int s2p_diag_vgpr_canary(int blocks, int regs, int spins, unsigned* out, void* stream) {
  if (regs <= 32) hipLaunchKernelGGL(vgpr_canary_kernel<24>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, spins, out);
  else if (regs <= 64) hipLaunchKernelGGL(vgpr_canary_kernel<56>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, spins, out);
  else hipLaunchKernelGGL(vgpr_canary_kernel<104>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, spins, out);
  S2P_CHECK_LAUNCH("vgpr_canary_kernel");
  return 0;
}

// run-time A/B switches (diagnostics build only): 0 = no generalised plane kernel (conv_planeg.hip), 1 = the tiled 7x7 thin weight
// gradient instead of the row-streaming one, 2 = no generalised plane kernel for 3x3 convs (VGG conv4_x),
// 3 = no 4-channel-pitch kernel for the 7x7 thin convs (stem forward, output-conv dgrad), 4 = no row-band form of the generalised plane kernel,
// 5 = no padded-raster weight gradient of the 4x4 layers (wgrad_slabg.hip), 6 = that kernel also for the 3x3 stride-2 layers,
// 8 = no 192-pixel row bands for the short-K layers (VGG conv1_2 / conv2_x take the 448-pixel bands),
// 9 = no register-resident InstanceNorm forward for the large planes (reduce + apply instead),
// 13 = the VALU / shuffle form of the PatchGAN logit-head forward instead of the MFMA one,
// 14 = the stem's weight gradient on the implicit GEMM instead of the row-streaming kernel with exchanged operands,
// round 5: 7 = no gamma|beta staging under the K loop (conv_plane GST), 10 = sub-pixel phases of the 64-row conv_dma tile in blockIdx.z,
// 11 = 64-channel slabs for the 42x42 InstanceNorm forward, 12 = the round-2 logit-head weight gradient, 15 = row bands dealt round-robin
// over the XCDs, 16 = one (not two) workgroups per CU as the K-split target of the 4x4 weight gradients, 17 = the PAIR launch's group in
// blockIdx.y, 18 = the slab weight gradients' (n, row, column) address state instead of the tabulated padded raster, 19 = the same for
// the implicit-GEMM weight gradient's gathered operand (wgrad_dma_kernel), 20 = one image per plane in conv_planeg (no stacked planes)
int s2p_diag_switch[32] = {0};
extern "C" __attribute__((visibility("default"))) int s2p_diag_set(int key, int value) {
  if (key < 0 || key >= 32) return -1;
  s2p_diag_switch[key] = value;
  return 0;
}
#endif
