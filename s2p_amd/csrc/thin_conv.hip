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
#include <type_traits>

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
  // U pixels per trip: their x / dY loads are all issued before the first FMA (one dependent load per iteration left the
  // kernel latency-bound: ~56 iterations x one L2/HBM latency each)
  constexpr int U = 8;
  for (int m0 = m_begin; m0 < m_end; m0 += U) {
    u32x4 xr[U], dr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + u;
      xr[u] = (u32x4){0u, 0u, 0u, 0u}; dr[u] = (u32x4){0u, 0u, 0u, 0u};
      if (m < m_end) {
        const int n = m / HoWo, rr = m - n * HoWo, oy = rr / a.Wo, ox = rr - oy * a.Wo;
        int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        if (a.reflect) {
          iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
        }
        if (pok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          xr[u] = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch + ch * 8);
          dr[u] = *(const u32x4*)(a.dy + (size_t)m * a.y_pitch);          // same address for the whole block: broadcast
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Chunk<__bf16> xv, dv;
      xv.raw = xr[u]; dv.raw = dr[u];
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

static bool tiled_applicable(const s2p_conv_desc* d);
int s2p_thin_tiled_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                       float slope, hipStream_t st);
int s2p_thin_tiled_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, int cin_real, void* ws, size_t ws_bytes, hipStream_t st);
static int tiled_wgrad_blocks(const s2p_conv_desc* d);

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
  static const int no_head = s2p_env_set("S2P_NO_HEAD_FWD");          // A/B switch (diagnostics build only)
  if (!no_head && s2p_head_fwd_applicable(d)) return s2p_head_fwd(d, x, w, bias, y, act, slope, st);
  static const int no_rows = s2p_env_set("S2P_NO_THIN_ROWS");         // A/B switch (diagnostics build only)
  if (!no_rows && act != S2P_ACT_SWISH && s2p_thin_rows_applicable(d)) return s2p_thin_rows_fwd(d, x, w, bias, y, act, slope, st);
  if (tiled_applicable(d)) return s2p_thin_tiled_fwd(d, x, w, bias, y, act, slope, st);
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

// scratch of the atomics-free form (tiled kernel only: per-workgroup partial tiles)
size_t s2p_thin_wgrad_ws_bytes(const s2p_conv_desc* d, int cin_real) {
  if (!tiled_applicable(d)) return 0;
  return (size_t)tiled_wgrad_blocks(d) * d->Cout * d->KH * d->KW * cin_real * sizeof(float);
}

int s2p_thin_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, int cin_real, void* ws, size_t ws_bytes,
                   hipStream_t st) {
  if (tiled_applicable(d)) return s2p_thin_tiled_wgrad(d, x, dy, dw, cin_real, ws, ws_bytes, st);
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

// ================================================================================================
// Spatially tiled MFMA kernels for thin-Cout, large-kernel, stride-1 convs (the generator's 7x7 64->3 output
// conv at 84x84 .. 256x256).  A workgroup owns a 16x8 output tile of one image and stages the input halo tile
// ((16+K-1) x (8+K-1) positions x Cin) ONCE in LDS; all K*K taps then read LDS, so the 49-fold re-read of the
// activation that an implicit GEMM does through L2 disappears.  The contraction runs on v_mfma_f32_16x16x32_bf16
// with the thin dimension (Cout <= 4, padded to 16) as one MFMA side: even at 3/16 utilisation the matrix core
// is ~8x faster than the packed-VALU direct form.
//   fwd  : D[pixel][co]  = sum_{tap,ci} X[pixel+tap][ci] * W[co][tap][ci]      A = halo rows (b128 reads), B = W
//   wgrad: D[co][ci]@tap = sum_pixel  dY[pixel][co] * X[pixel+tap][ci]        A = dY^T (LDS [co][pixel]),
//                                                                              B = halo via ds_read_b64_tr_b16
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int TW = 16, TH = 8;

struct TileArgs {
  const __bf16* x; const __bf16* w; const float* bias; const __bf16* dy; __bf16* y; float* dw;
  int N, H, W, Cin, x_pitch, Ho, Wo, Cout, y_pitch, K, pad, reflect, act;
  float slope;
  int cin_real, tiles_x, tiles_y, ntiles;
  float* part;                 // weight gradient: per-workgroup partial tiles [gridDim.x][Cout][T][cin_real] (nullptr: fp32 atomics)
};

// Halo tile -> LDS.  All global loads of a thread are issued before the first LDS store (up to MAXL 16-byte loads in
// flight per thread) so the phase costs about one memory latency instead of one per chunk.
template <int MAXL>
__device__ __forceinline__ void load_halo(const TileArgs& a, char* xt, int xs, int n, int oy0, int ox0, int HW_, int HH) {
  const int nch = a.Cin / 8;
  const int total = HW_ * HH * nch;
  u32x4 v[MAXL];
  int dst[MAXL];
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    const int idx = threadIdx.x + i * blockDim.x;
    v[i] = (u32x4){0u, 0u, 0u, 0u};
    dst[i] = -1;
    if (idx < total) {
      int pos = idx / nch, ch = idx - pos * nch;
      int hy = pos / HW_, hx = pos - hy * HW_;
      int iy = oy0 + hy - a.pad, ix = ox0 + hx - a.pad;
      bool ok = true;
      if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
        iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);        // positions that belong to clipped edge-tile pixels
        ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      } else {
        ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      }
      if (ok) v[i] = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch + ch * 8);
      dst[i] = pos * xs + ch * 16;
    }
  }
#pragma unroll
  for (int i = 0; i < MAXL; ++i)
    if (dst[i] >= 0) *(u32x4*)(xt + dst[i]) = v[i];
  // tiles larger than MAXL * blockDim chunks (not reached by the shapes admitted in tiled_applicable)
  for (int idx = threadIdx.x + MAXL * blockDim.x; idx < total; idx += blockDim.x) {
    int pos = idx / nch, ch = idx - pos * nch;
    int hy = pos / HW_, hx = pos - hy * HW_;
    int iy = oy0 + hy - a.pad, ix = ox0 + hx - a.pad;
    bool ok = true;
    if (a.reflect) {
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
    } else {
      ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    }
    u32x4 t = {0u, 0u, 0u, 0u};
    if (ok) t = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch + ch * 8);
    *(u32x4*)(xt + pos * xs + ch * 16) = t;
  }
}

// register-lean variant (one chunk in flight per thread): the wgrad kernel has no VGPRs to spare for staging
__device__ __forceinline__ void load_halo_simple(const TileArgs& a, char* xt, int xs, int n, int oy0, int ox0, int HW_, int HH) {
  const int nch = a.Cin / 8;
  for (int idx = threadIdx.x; idx < HW_ * HH * nch; idx += blockDim.x) {
    int pos = idx / nch, ch = idx - pos * nch;
    int hy = pos / HW_, hx = pos - hy * HW_;
    int iy = oy0 + hy - a.pad, ix = ox0 + hx - a.pad;
    bool ok = true;
    if (a.reflect) {
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
    } else {
      ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    }
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch + ch * 8);
    *(u32x4*)(xt + pos * xs + ch * 16) = v;
  }
}

// KS / CS: compile-time kernel size / Cin (0 = run-time): the 49 x 2 (tap, 32-channel) steps of the 7x7 64->3 conv are
// then straight-line code whose LDS reads the compiler can hoist over the MFMAs
template <int KS, int CS>
__global__ __launch_bounds__(256) void thin_tiled_fwd_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int aK = KS > 0 ? KS : a.K, aCin = CS > 0 ? CS : a.Cin;
  const int T = aK * aK, HW_ = TW + aK - 1, HH = TH + aK - 1;
  const int xs = aCin * 2 + 16;                         // halo position stride (padded: conflict-free b128 reads)
  char* xt = smem;
  char* wt = xt + HW_ * HH * xs;                         // [Cout][T][Cin] bf16
  char* ot = wt + a.Cout * T * aCin * 2;                // [128 pixels][8] bf16 output staging
  const int tile = blockIdx.x;
  const int n = tile / (a.tiles_x * a.tiles_y), tr = tile - n * a.tiles_x * a.tiles_y;
  const int oy0 = (tr / a.tiles_x) * TH, ox0 = (tr % a.tiles_x) * TW;
  const int wbytes = a.Cout * T * aCin * 2;
  for (int i = threadIdx.x * 16; i < wbytes; i += 256 * 16) *(u32x4*)(wt + i) = *(const u32x4*)((const char*)a.w + i);
  for (int i = threadIdx.x * 16; i < TW * TH * 16; i += 256 * 16) *(u32x4*)(ot + i) = (u32x4){0u, 0u, 0u, 0u};
  load_halo<10>(a, xt, xs, n, oy0, ox0, HW_, HH);
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;             // A: pixel row l15, k group kg ; B: column (co) l15
  f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bf16x8 zero8 = {};
  const int nck = aCin / 32;
#pragma unroll
  for (int ky = 0; ky < aK; ++ky)
#pragma unroll
    for (int kx = 0; kx < aK; ++kx) {
      const int t = ky * aK + kx;
#pragma unroll
      for (int ck = 0; ck < nck; ++ck) {
        bf16x8 bfrag = zero8;
        if (l15 < a.Cout) bfrag = *(const bf16x8*)(wt + ((l15 * T + t) * aCin + ck * 32 + kg * 8) * 2);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int ry = wave * 2 + rr;
          const bf16x8 afrag = *(const bf16x8*)(xt + ((ry + ky) * HW_ + l15 + kx) * xs + (ck * 32 + kg * 8) * 2);
          acc[rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, acc[rr], 0, 0, 0);
        }
      }
    }
  // D: col = lane & 15 (co), row = (lane >> 4) * 4 + reg (pixel within the 16-pixel tile row)
  if (l15 < a.Cout) {
    const float b = a.bias ? a.bias[l15] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int pix = (wave * 2 + rr) * TW + kg * 4 + e;
        *(__bf16*)(ot + pix * 16 + l15 * 2) = (__bf16)act_fwd(acc[rr][e] + b, a.act, a.slope);
      }
  }
  __syncthreads();
  if (threadIdx.x < TW * TH) {
    int ry = threadIdx.x / TW, rx = threadIdx.x - ry * TW;
    int oy = oy0 + ry, ox = ox0 + rx;
    if (oy < a.Ho && ox < a.Wo)
      *(u32x4*)(a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_pitch) = *(const u32x4*)(ot + threadIdx.x * 16);
  }
}

// 512 threads (8 waves); persistent over tiles; each wave owns (tap, 16-channel group) pairs w, w+8, ...
template <int MAXP, int KS = 0, int CS = 0>          // KS / CS: compile-time kernel size / Cin (0 = run-time)
__global__ __launch_bounds__(512) void thin_tiled_wgrad_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int aK = KS > 0 ? KS : a.K, aCin = CS > 0 ? CS : a.Cin;
  const int T = aK * aK, HW_ = TW + aK - 1, HH = TH + aK - 1;
  const int xs = aCin * 2 + 16;
  char* xt = smem;
  char* dyt = xt + HW_ * HH * xs;                        // [4 co][128 pixels] bf16, transposed dY tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int q = (lane >> 2) & 3, p = lane & 3;           // transposed-read lane roles inside the 16-lane group
  const int ncg = aCin / 16, npairs = T * ncg;
  f32x4_t acc[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16x8 zero8 = {};
  typedef __attribute__((ext_vector_type(8))) short s16x8;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_x * a.tiles_y), tr = tile - n * a.tiles_x * a.tiles_y;
    const int oy0 = (tr / a.tiles_x) * TH, ox0 = (tr % a.tiles_x) * TW;
    __syncthreads();                                     // previous tile fully consumed
    load_halo_simple(a, xt, xs, n, oy0, ox0, HW_, HH);
    if (threadIdx.x < TW * TH) {
      int ry = threadIdx.x / TW, rx = threadIdx.x - ry * TW;
      int oy = oy0 + ry, ox = ox0 + rx;
      Chunk<__bf16> d; d.raw = (u32x4){0u, 0u, 0u, 0u};
      if (oy < a.Ho && ox < a.Wo) d.raw = *(const u32x4*)(a.dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_pitch);
#pragma unroll
      for (int co = 0; co < 4; ++co) *(unsigned short*)(dyt + (co * TW * TH + threadIdx.x) * 2) = (unsigned short)((d.raw[co >> 1] >> ((co & 1) * 16)) & 0xffff);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {                     // 4 x 32 pixels = two tile rows each
      bf16x8 afrag = zero8;                              // A[row = co][k = 8*kg + j] = dY[pixel 32*kk + 8*kg + j][co]
      if (l15 < a.Cout) afrag = *(const bf16x8*)(dyt + (l15 * TW * TH + 32 * kk + 8 * kg) * 2);
      // this lane group's 8 pixels: tile row ry = 2*kk + (kg >> 1), columns 8*(kg & 1) .. +7
      const int ry = 2 * kk + (kg >> 1), px0 = 8 * (kg & 1);
#pragma unroll
      for (int i = 0; i < MAXP; ++i) {
        const int pr = wave + 8 * i;
        if (pr < npairs) {                                // wave-uniform
          const int t = pr / ncg, cg = pr - t * ncg;
          const int ky = t / aK, kx = t - ky * aK;
          const char* base = xt + ((ry + ky) * HW_ + px0 + kx + q) * xs + (cg * 16 + 4 * p) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * xs));
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, v), acc[i], 0, 0, 0);
        }
      }
    }
  }
  // D: col = lane & 15 (ci within the group), row = (lane >> 4) * 4 + reg (co): rows < Cout live in lanes 0..15
  if (kg == 0) {
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int pr = wave + 8 * i;
      if (pr < npairs) {
        const int t = pr / ncg, cg = pr - t * ncg;
        const int ci = cg * 16 + l15;
        if (ci < a.cin_real)
#pragma unroll
          for (int co = 0; co < 4; ++co)
            if (co < a.Cout) {
              const size_t o = ((size_t)co * T + t) * a.cin_real + ci;
              if (a.part) a.part[(size_t)blockIdx.x * ((size_t)a.Cout * T * a.cin_real) + o] = acc[i][co];
              else atomicAdd(a.dw + o, acc[i][co]);
            }
      }
    }
  }
}

// ---- 7x7 stride-1 pad-3 weight gradient with Cin = 64 and Cout <= 4 (the generator's output conv), row-streaming form (round 4) ------
// The tiled kernel above re-reads x 2.4 times (16x8 tiles with a 6-pixel halo), loads each halo synchronously and spends an MFMA
// per (tap, 16 channels) with 3 of its 16 rows used: 125 us, alone on the chip at the start of the generator's backward.  Here
//   * a workgroup owns a BAND of 21-23 output rows of one image over the full width and streams the input rows once through an
//     8-row LDS ring (1.29x instead of 2.4x; the next row's global loads are in flight while the current one is computed);
//   * the MFMA's thin side carries (kx, co) -- 21 of 32 rows used instead of 3 of 16: for a fixed ky,
//         dW[co][ky][kx][ci] = sum_{p'} dY[row][p' - kx][co] * X[row + ky][p'][ci]
//     so A[m = (kx, co)][k = p'] is dY shifted by kx (seven shifted planar copies of each dY row are kept in LDS, so every
//     fragment is one aligned 16-byte read) and B[k = p'][ci] is the halo row through ds_read_b64_tr_b16, one k-chunk = 32 columns;
//   * wave (m tile, 16-channel group) keeps the accumulators of all seven ky: an input row is read from LDS once and used by up
//     to seven output rows.
constexpr int R7_XC = 96, R7_XS = 144, R7_MR = 24;
// rows per band: the fewest bands of at most 23 rows, equal in length (84 rows: 4 x 21; the stem's padded 90 rows: 4 x 23 -- 64 images x 4
// bands are ONE round of the 256 CUs; 5 bands of 21 were two rounds: 96 us against 52)
static inline int r7_nbands(int H) { return (H + 22) / 23; }
static inline int r7_band(int H) { const int nb = r7_nbands(H); return (H + nb - 1) / nb; }
struct Rows7Args {
  const __bf16* x; const __bf16* dy; float* dw; float* part;
  int N, H, W, x_pitch, y_pitch, Cout, cin_real, reflect, nbands;
  // SWAPPED use (the stem 3 -> 64: thin INPUT, wide OUTPUT gradient): `dy` is the reflect- / zero-padded input on the (H, W) grid =
  // image + 6, `x` is the 64-channel output gradient of (Hw, Ww) = the image, sitting at offset woff = 3 inside that grid (zeros
  // around it); the result is the stem's gradient with both taps flipped and the channel roles exchanged (swap != 0):
  //   G[c3][ky][kx][c64] = sum_p thin[p][c3] * wide[p + (ky - 3, kx - 3)][c64] = dW_stem[c64][6 - ky][6 - kx][c3]
  int Hw, Ww, woff, swap;
  int band;                                              // rows per band (r7_band)
};

__global__ __launch_bounds__(512) void thin_rows7_wgrad_kernel(const Rows7Args a) {
  __shared__ __attribute__((aligned(16))) char xring[8 * R7_XC * R7_XS];
  __shared__ __attribute__((aligned(16))) __bf16 dyc[8 * R7_MR * R7_XC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int q = (lane >> 2) & 3, p = lane & 3;           // transposed-read lane roles inside the 16-lane group
  const int cg = wave & 3, mt = wave >> 2;
  const int n = blockIdx.x / a.nbands, band = blockIdx.x - n * a.nbands;
  const int r0 = band * a.band;
  const int nrows = a.H - r0 < a.band ? a.H - r0 : a.band;
  const int M = 7 * a.Cout;
  for (int i = tid; i < 8 * R7_XC * R7_XS / 16; i += 512) ((u32x4*)xring)[i] = (u32x4){0u, 0u, 0u, 0u};
  for (int i = tid; i < 8 * R7_MR * R7_XC * 2 / 16; i += 512) ((u32x4*)dyc)[i] = (u32x4){0u, 0u, 0u, 0u};
  f32x4_t acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) acc[k] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int ncols = a.W + 6;                             // halo columns: image columns -3 .. W + 2
  u32x4 xv[2], dv;
  auto load_row = [&](int h) {                           // halo row h of x and output row h of dY -> registers
    int iy = r0 - 3 + h - a.woff;                          // row of the wide tensor (its own grid: Hw x Ww at offset woff)
    bool rok = true;
    if (a.reflect) { iy = iy < 0 ? -iy : (iy >= a.Hw ? 2 * a.Hw - 2 - iy : iy); iy = iy < 0 ? 0 : (iy >= a.Hw ? a.Hw - 1 : iy); }
    else rok = iy >= 0 && iy < a.Hw;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, col = idx >> 3, ch = idx & 7;
      xv[i] = (u32x4){0u, 0u, 0u, 0u};
      if (col < ncols) {
        int ix = col - 3 - a.woff;
        bool ok = rok;
        if (a.reflect) { ix = ix < 0 ? -ix : (ix >= a.Ww ? 2 * a.Ww - 2 - ix : ix); ix = ix < 0 ? 0 : (ix >= a.Ww ? a.Ww - 1 : ix); }
        else ok = ok && ix >= 0 && ix < a.Ww;
        if (ok) xv[i] = *(const u32x4*)(a.x + (((size_t)n * a.Hw + iy) * a.Ww + ix) * a.x_pitch + ch * 8);
      }
    }
    // dY: four threads per pixel (each scatters two of the seven shifted copies below; round 4 gave a pixel's 21 two-byte stores to
    // ONE thread, i.e. to the first two waves, and the other six waited for them at every row's barrier)
    dv = (u32x4){0u, 0u, 0u, 0u};
    if ((tid >> 2) < a.W && h < nrows) dv = *(const u32x4*)(a.dy + (((size_t)n * a.H + r0 + h) * a.W + (tid >> 2)) * a.y_pitch);
  };
  auto store_row = [&](int h) {
    const int slot = h & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, col = idx >> 3, ch = idx & 7;
      if (col < ncols) *(u32x4*)(xring + (slot * R7_XC + col) * R7_XS + ch * 16) = xv[i];
    }
    if ((tid >> 2) < a.W && h < nrows) {
      const int col = tid >> 2, part = tid & 3;
      for (int co = 0; co < a.Cout; ++co) {
        const unsigned short v = (unsigned short)((dv[co >> 1] >> ((co & 1) * 16)) & 0xffff);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kx = part + 4 * u;
          if (kx < 7) *(unsigned short*)(dyc + ((slot * R7_MR + kx * a.Cout + co) * R7_XC + col + kx)) = v;
        }
      }
    }
  };
  __syncthreads();                                       // the rings are zero
  load_row(0);
  store_row(0);
  __syncthreads();
  const int nsteps = nrows + 6;
  const int m = 16 * mt + l15;
  const bf16x8 zero8 = {};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  // A fragments of rows m >= M are zeros.  With M < R7_MR every ring slot has a never-written (zero) row M: such lanes READ that row
  // instead of selecting zeros behind each of the 21 loads of a row step (4 v_cndmask each: 84 of the step's ~160 VALU instructions;
  // round 5's instruction-mix counters: 10 VALU instructions per MFMA in this kernel)
  const bool zrow = M < R7_MR;                           // (launch-uniform)
  const int mr = (m < M || !zrow) ? m : M;
  auto row_steps = [&](auto guardc) {
  constexpr bool GUARD = decltype(guardc)::value;
  for (int h = 0; h < nsteps; ++h) {
    if (h + 1 < nsteps) load_row(h + 1);
    // B fragments of halo row h: [k = 32 c + 8 kg + j][ci = 16 cg + l15]
    bf16x8 bfr[3];
    const char* xrow = xring + ((h & 7) * R7_XC) * R7_XS;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const char* base = xrow + (32 * c + 8 * kg + q) * R7_XS + (cg * 16 + 4 * p) * 2;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * R7_XS));
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[c] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int r = h - ky;                              // output row (in the band) that pairs with halo row h under tap row ky
      if (r >= 0 && r < nrows) {                         // workgroup-uniform
        const __bf16* drow = dyc + ((r & 7) * R7_MR + mr) * R7_XC + 8 * kg;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          bf16x8 afr = *(const bf16x8*)(drow + 32 * c);
          if constexpr (GUARD) { if (m >= M) afr = zero8; }
          acc[ky] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[c], acc[ky], 0, 0, 0);
        }
      }
    }
    if (h + 1 < nsteps) store_row(h + 1);                // slot (h + 1) & 7 held row h - 7: nobody reads it any more
    __syncthreads();
  }
  };
  if (zrow) row_steps(std::integral_constant<bool, false>{}); else row_steps(std::integral_constant<bool, true>{});
  // D: col = lane & 15 (ci within the group), row = (lane >> 4) * 4 + reg -> m = (kx, co)
  const int ci = cg * 16 + l15;
  if (ci < a.cin_real) {
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int mm = 16 * mt + 4 * kg + e;
        if (mm < M) {
          const int kx = mm / a.Cout, co = mm - kx * a.Cout;
          const size_t o = a.swap ? ((size_t)ci * 49 + (6 - ky) * 7 + (6 - kx)) * a.Cout + co          // dW_stem[c64][6 - ky][6 - kx][c3]
                                  : ((size_t)co * 49 + ky * 7 + kx) * a.cin_real + ci;
          if (a.part) a.part[(size_t)blockIdx.x * ((size_t)a.Cout * 49 * a.cin_real) + o] = acc[ky][e];
          else atomicAdd(a.dw + o, acc[ky][e]);
        }
      }
  }
}

static bool rows7_applicable(const s2p_conv_desc* d) {
  return d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->Cout <= 4 && d->stride == 1 && d->KH == 7 && d->KW == 7 &&
         d->pad == 3 && d->Cin == 64 && d->x_pitch == 64 && d->y_pitch == 8 && d->Ho == d->H && d->Wo == d->W && d->W + 6 <= R7_XC &&
         d->W <= 512 && d->H >= 7;
}
static int rows7_blocks(const s2p_conv_desc* d) { return d->N * r7_nbands(d->H); }

static bool tiled_applicable(const s2p_conv_desc* d) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->Cout <= 4 && d->stride == 1 && d->KH == d->KW &&
        d->KH >= 3 && d->KH <= 7 && d->Cin % 32 == 0 && d->Cin <= 128 && d->y_pitch == 8 && d->x_pitch == d->Cin))
    return false;
  const int HW_ = TW + d->KH - 1, HH = TH + d->KH - 1;
  const long long lds = (long long)HW_ * HH * (d->Cin * 2 + 16) + (long long)d->Cout * d->KH * d->KW * d->Cin * 2 + TW * TH * 16;
  const int pairs = d->KH * d->KW * (d->Cin / 16);
  return lds <= 64 * 1024 && pairs <= 8 * 32 && d->Ho * d->Wo >= 1024;
}

static void fill_tile_args(TileArgs& a, const s2p_conv_desc* d) {
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_pitch = d->x_pitch; a.Ho = d->Ho; a.Wo = d->Wo;
  a.Cout = d->Cout; a.y_pitch = d->y_pitch; a.K = d->KH; a.pad = d->pad; a.reflect = d->reflect;
  a.tiles_x = cdiv(d->Wo, TW); a.tiles_y = cdiv(d->Ho, TH); a.ntiles = d->N * a.tiles_x * a.tiles_y;
}

int s2p_thin_tiled_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                       float slope, hipStream_t st) {
  TileArgs a{};
  fill_tile_args(a, d);
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y; a.act = act; a.slope = slope;
  const int HW_ = TW + d->KH - 1, HH = TH + d->KH - 1;
  const size_t lds = (size_t)HW_ * HH * (d->Cin * 2 + 16) + (size_t)d->Cout * d->KH * d->KW * d->Cin * 2 + TW * TH * 16;
  static const int no_static = s2p_env_set("S2P_NO_THIN_STATIC");
  if (d->KH == 7 && d->Cin == 64 && !no_static) hipLaunchKernelGGL((thin_tiled_fwd_kernel<7, 64>), dim3(a.ntiles), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((thin_tiled_fwd_kernel<0, 0>), dim3(a.ntiles), dim3(256), lds, st, a);
  S2P_CHECK_LAUNCH("thin_tiled_fwd_kernel");
  return 0;
}

static int tiled_wgrad_blocks(const s2p_conv_desc* d) {      // partial tiles the caller's workspace must hold
  TileArgs a{};
  fill_tile_args(a, d);
  const int tiled = a.ntiles < 512 ? a.ntiles : 512;
  if (!rows7_applicable(d)) return tiled;
#ifdef S2P_DIAG_BUILD
  return rows7_blocks(d) > tiled ? rows7_blocks(d) : tiled;   // the diagnostics build can switch between the two kernels at run time
#else
  return rows7_blocks(d);
#endif
}

int s2p_thin_tiled_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, int cin_real, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  TileArgs a{};
  fill_tile_args(a, d);
  a.x = (const __bf16*)x; a.dy = (const __bf16*)dy; a.dw = dw; a.cin_real = cin_real;
  const size_t need = s2p_thin_wgrad_ws_bytes(d, cin_real);
  a.part = (ws && need > 0 && ws_bytes >= need) ? (float*)ws : nullptr;
  if (rows7_applicable(d) && !S2P_DIAG_SWITCH(1)) {
    Rows7Args r{a.x, a.dy, dw, a.part, d->N, d->H, d->W, d->x_pitch, d->y_pitch, d->Cout, cin_real, d->reflect, r7_nbands(d->H),
                d->H, d->W, 0, 0, r7_band(d->H)};
    const int blocks = rows7_blocks(d);
    hipLaunchKernelGGL(thin_rows7_wgrad_kernel, dim3(blocks), dim3(512), 0, st, r);
    S2P_CHECK_LAUNCH("thin_rows7_wgrad_kernel");
    if (a.part) {
      const int n = d->Cout * 49 * cin_real;
      s2p_partial_reduce(a.part, blocks, n, n, dw, st);
      S2P_CHECK_LAUNCH("s2p_partial_reduce_kernel(thin rows7 wgrad)");
    }
    return 0;
  }
  const int HW_ = TW + d->KH - 1, HH = TH + d->KH - 1;
  const size_t lds = (size_t)HW_ * HH * (d->Cin * 2 + 16) + 4 * TW * TH * 2;
  const int pairs = d->KH * d->KW * (d->Cin / 16);
  int blocks = a.ntiles < 512 ? a.ntiles : 512;
  static const int no_static = s2p_env_set("S2P_NO_THIN_STATIC");
  if (d->KH == 7 && d->Cin == 64 && !no_static) hipLaunchKernelGGL((thin_tiled_wgrad_kernel<25, 7, 64>), dim3(blocks), dim3(512), lds, st, a);
  else if (pairs <= 8 * 13) hipLaunchKernelGGL(thin_tiled_wgrad_kernel<13>, dim3(blocks), dim3(512), lds, st, a);
  else if (pairs <= 8 * 25) hipLaunchKernelGGL(thin_tiled_wgrad_kernel<25>, dim3(blocks), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(thin_tiled_wgrad_kernel<32>, dim3(blocks), dim3(512), lds, st, a);
  S2P_CHECK_LAUNCH("thin_tiled_wgrad_kernel");
  if (a.part) {
    const int n = d->Cout * d->KH * d->KW * cin_real;
    s2p_partial_reduce(a.part, blocks, n, n, dw, st);          // dw += the workgroups' partial tiles, in a fixed order
    S2P_CHECK_LAUNCH("s2p_partial_reduce_kernel(thin tiled wgrad)");
  }
  return 0;
}

// ---- the STEM's weight gradient (3 -> 64, 7x7, stride 1, pad 3: thin INPUT) on the row-streaming kernel with the operands exchanged ----
// dW[co][ky][kx][ci] = sum_p dY[p][co] * Xpad[p + (ky, kx)][ci]  =  sum_P Xpad[P][ci] * dY[P - (ky, kx)][co]   (P on the padded grid):
// the same sum with the THIN tensor (the padded image, 3 channels) at the centre position and the WIDE one (dY, 64 channels) read at
// the negated tap -- thin_rows7_wgrad_kernel's form with flipped taps (Rows7Args::swap).  The implicit GEMM ran this layer with a
// half-empty A tile, 8-channel pads for 3 channels and 128 K splits: 101 us + reduce at 78 TFLOP/s, the last conv launch of the generator's
// backward.  A small kernel writes the padded (reflect / zero) 8-channel-pitch copy of the image first.
struct Pad8Args { const __bf16* x; __bf16* y; int N, H, W, x_pitch, pad, reflect; };
__global__ __launch_bounds__(256) void pad8_kernel(const Pad8Args a) {
  const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
  const long long total = (long long)a.N * Hp * Wp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / ((long long)Hp * Wp));
    const int r = (int)(i - (long long)n * Hp * Wp);
    int iy = r / Wp - a.pad, ix = r % Wp - a.pad;
    bool ok = true;
    if (a.reflect) {
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy); ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
    } else ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *(const u32x4*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch);
    *(u32x4*)(a.y + (size_t)i * 8) = v;
  }
}

bool s2p_stem_wgrad_applicable(const s2p_conv_desc* d, int cin_real, int cout_real) {
  return d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->KH == 7 && d->KW == 7 && d->stride == 1 && d->pad == 3 &&
         d->Cin == 8 && d->x_pitch == 8 && cin_real >= 1 && cin_real <= 4 && d->Cout == 64 && cout_real == 64 && d->y_pitch == 64 &&
         d->Ho == d->H && d->Wo == d->W && d->W + 12 <= R7_XC && d->H >= 7 && (!d->reflect || (d->H >= 4 && d->W >= 4)) &&
         !S2P_DIAG_SWITCH(14);
}
static inline size_t stem_align(size_t b) { return (b + 255) / 256 * 256; }
size_t s2p_stem_wgrad_ws_bytes(const s2p_conv_desc* d, int cin_real) {
  const size_t padded = stem_align((size_t)d->N * (d->H + 6) * (d->W + 6) * 8 * sizeof(__bf16));
  const size_t parts = stem_align((size_t)d->N * r7_nbands(d->H + 6) * cin_real * 49 * 64 * sizeof(float));
  return padded + parts + s2p_channel_sum_ws_bytes((int64_t)d->N * d->H * d->W, 64);      // (+ the bias gradient's partial sums)
}
int s2p_stem_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real, void* ws, size_t ws_bytes,
                   hipStream_t st) {
  if (!ws || ws_bytes < s2p_stem_wgrad_ws_bytes(d, cin_real)) S2P_FAIL(-1, "s2p_stem_wgrad: workspace of %zu bytes needed", s2p_stem_wgrad_ws_bytes(d, cin_real));
  const int Hp = d->H + 6, Wp = d->W + 6;
  __bf16* xpad = (__bf16*)ws;
  const size_t padded = stem_align((size_t)d->N * Hp * Wp * 8 * sizeof(__bf16));
  const size_t parts = stem_align((size_t)d->N * r7_nbands(Hp) * cin_real * 49 * 64 * sizeof(float));
  float* part = (float*)((char*)ws + padded);
  if (db) {
    int rc = s2p_channel_sum_det(d->dtype, dy, (int64_t)d->N * d->H * d->W, 64, d->y_pitch, db, (char*)ws + padded + parts,
                                 s2p_channel_sum_ws_bytes((int64_t)d->N * d->H * d->W, 64), st);
    if (rc) return rc;
  }
  Pad8Args p{(const __bf16*)x, xpad, d->N, d->H, d->W, d->x_pitch, 3, d->reflect};
  const long long total = (long long)d->N * Hp * Wp;
  int pb = (int)((total + 255) / 256); if (pb > 4096) pb = 4096;
  hipLaunchKernelGGL(pad8_kernel, dim3(pb), dim3(256), 0, st, p);
  S2P_CHECK_LAUNCH("pad8_kernel");
  const int nbands = r7_nbands(Hp), blocks = d->N * nbands;
  // wide := dY (64 channels, image grid at offset 3 of the padded grid, zeros around), thin := the padded image (Cout field = its channels)
  Rows7Args r{(const __bf16*)dy, xpad, dw, part, d->N, Hp, Wp, d->y_pitch, 8, cin_real, 64, 0, nbands, d->H, d->W, 3, 1, r7_band(Hp)};
  hipLaunchKernelGGL(thin_rows7_wgrad_kernel, dim3(blocks), dim3(512), 0, st, r);
  S2P_CHECK_LAUNCH("thin_rows7_wgrad_kernel(stem)");
  const int n = cin_real * 49 * 64;
  s2p_partial_reduce(part, blocks, n, n, dw, st);
  S2P_CHECK_LAUNCH("s2p_partial_reduce_kernel(stem wgrad)");
  return 0;
}
