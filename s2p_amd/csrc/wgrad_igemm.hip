// Weight-gradient implicit GEMM for gfx950:  dW[a][(t,c)] += sum_m A[m][a] * B[gather(m,t)][c]
//   conv2d        : A = dY (its own pixel grid = output grid), B = X gathered at (qy*s + ky - pad)
//   conv_transpose: A = X  (input grid),                       B = dY gathered at (qy*s + ky - pad)
// The reduction axis (pixels) is the NON-contiguous axis of both NHWC operands, so the MFMA fragments are
// read from LDS with the gfx950 transposed read ds_read_b64_tr_b16 (bf16) or as single dwords (fp32: the
// v_mfma_f32_32x32x2_f32 operand is one float per lane, so no transpose is needed).  Tiles are staged exactly
// as they lie in HBM ([pixel][channel], 16-B chunks along channels, coalesced) with rows padded 256 -> 320 B
// so that the four k-rows of one transposed read fall on disjoint banks.  Split-K over the pixel axis with
// fp32 atomics whose wave-instruction shape is two 128-B row segments (the full-rate shape on this chip).
#include "s2p_common.h"

struct WgradArgs {
  const void* A; const void* B; float* dW;
  int M, Qh, Qw;                // pixel grid of A
  int Ca, a_pitch, a_gstride;   // A channels (multiple of CE), pitch
  int Ca_real;                  // rows of dW actually written
  int Hi, Wi, Cb, b_pitch, b_gstride;   // gathered tensor
  int Cb_real;
  int istride, reflect;
  int T, NB;                    // taps, NB = T*Cb
  long long dw_gstride; int dw_row;   // dW row length = T*Cb_real
  int splitk, steps_per_split;
  int na_tiles, nb_tiles;
  int tap[64];
};

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int BT = 256 / (int)sizeof(T);   // tile width in channels: 128 (bf16) / 64 (fp32)
  constexpr int BKP = 32;                    // pixels per K step
  constexpr int RS = 320;                    // LDS row stride (bytes)
  constexpr int TT = BT / 2 / 32;            // MFMA tiles per wave per dim: 2 (bf16) / 1 (fp32)
  constexpr int TILE = BKP * RS;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [buf][A|B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = blockIdx.x;
  const int a_tile = bid % a.na_tiles, b_tile = bid / a.na_tiles;
  const int g = blockIdx.y;
  const int split = blockIdx.z;

  const int cc = tid & 15;          // chunk column (16 chunks = 256 B)
  const int pr0 = tid >> 4;         // pixel row 0..15 (+16)
  const int QQ = a.Qh * a.Qw;

  // A side
  const int a_ch = a_tile * BT + cc * CE;
  const bool a_ok = a_ch < a.Ca;
  const T* Ag = (const T*)a.A + (size_t)g * a.a_gstride + a_ch;
  // B side: fixed (tap, channel) for this thread
  const int nb = b_tile * BT + cc * CE;
  const bool b_ok = nb < a.NB;
  int bt = 0, bc = 0;
  if (b_ok) { bt = nb / a.Cb; bc = nb - bt * a.Cb; }
  const int ti = a.tap[bt];
  const int tdy = (int)(signed char)(ti & 0xff), tdx = (int)(signed char)((ti >> 8) & 0xff);
  const T* Bg = (const T*)a.B + (size_t)g * a.b_gstride + bc;

  const int step0 = split * a.steps_per_split;
  int nsteps = a.steps_per_split;
  {
    int total = (a.M + BKP - 1) / BKP;
    if (step0 + nsteps > total) nsteps = total - step0;
  }
  if (nsteps <= 0) return;   // uniform per block

  // pixel coordinates of this thread's two rows, advanced incrementally by BKP per step
  int pm[2], pn[2], py[2], px[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = step0 * BKP + pr0 + 16 * i;
    pm[i] = m;
    int n = m / QQ, rr = m - n * QQ, qy = rr / a.Qw, qx = rr - qy * a.Qw;
    pn[i] = n; py[i] = qy; px[i] = qx;
  }

  u32x4 regA[2], regB[2];
  auto load_global = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bool mok = pm[i] < a.M;
      u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
      if (mok && a_ok) va = *(const u32x4*)(Ag + (size_t)pm[i] * a.a_pitch);
      int iy = py[i] * a.istride + tdy, ix = px[i] * a.istride + tdx;
      if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.Hi ? 2 * a.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.Wi ? 2 * a.Wi - 2 - ix : ix);
      }
      if (mok && b_ok && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi)
        vb = *(const u32x4*)(Bg + ((size_t)(pn[i] * a.Hi + iy) * a.Wi + ix) * a.b_pitch);
      regA[i] = va; regB[i] = vb;
      // advance by BKP pixels
      pm[i] += BKP; px[i] += BKP;
      while (px[i] >= a.Qw) { px[i] -= a.Qw; if (++py[i] >= a.Qh) { py[i] = 0; ++pn[i]; } }
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * 2 * TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int row = pr0 + 16 * i;
      *(u32x4*)(base + row * RS + cc * 16) = regA[i];
      *(u32x4*)(base + TILE + row * RS + cc * 16) = regB[i];
    }
  };

  f32x16 acc[TT][TT];
#pragma unroll
  for (int i = 0; i < TT; ++i)
#pragma unroll
    for (int j = 0; j < TT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wa0 = (wave >> 1) * (TT * 32), wb0 = (wave & 1) * (TT * 32);

  load_global();
  store_lds(0);
  __syncthreads();

  for (int kt = 0; kt < nsteps; ++kt) {
    const bool more = kt + 1 < nsteps;
    if (more) load_global();
    const char* At = smem + (kt & 1) * 2 * TILE;
    const char* Bt = At + TILE;
    if constexpr (sizeof(T) == 2) {
      // lane l: 16-lane group gq = l>>4 : channel block 16*(gq&1), k half hh = gq>>1 ; inside the group lane
      // 4q+p supplies row q, columns 4p..4p+3; lane i receives column i, element q = row q.
      const int gq = lane >> 4, gg = gq & 1, hh = gq >> 1, q = (lane >> 2) & 3, p = lane & 3;
      const int rowoff = (8 * hh + q) * RS + (16 * gg + 4 * p) * 2;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[TT], bf[TT];
#pragma unroll
        for (int i = 0; i < TT; ++i) {
          const char* ptr = At + s * 16 * RS + rowoff + (wa0 + 32 * i) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr + 4 * RS));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[i] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < TT; ++j) {
          const char* ptr = Bt + s * 16 * RS + rowoff + (wb0 + 32 * j) * 2;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr + 4 * RS));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bf[j] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < TT; ++i)
#pragma unroll
          for (int j = 0; j < TT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int s = 0; s < BKP / 2; ++s) {
        float av = *(const float*)(At + (2 * s + h) * RS + (wa0 + r) * 4);
        float bv = *(const float*)(Bt + (2 * s + h) * RS + (wb0 + r) * 4);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0][0], 0, 0, 0);
      }
    }
    if (more) store_lds((kt + 1) & 1);
    __syncthreads();
  }

  // ---- split-K accumulation: lanes <-> consecutive b (contiguous floats), registers <-> rows a --------
  const int r = lane & 31, h = lane >> 5;
  float* dWg = a.dW + (size_t)g * a.dw_gstride;
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    int n = b_tile * BT + wb0 + 32 * j + r;
    if (n >= a.NB) continue;
    int t = n / a.Cb, c = n - t * a.Cb;
    if (c >= a.Cb_real) continue;
    int col = t * a.Cb_real + c;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int arow = a_tile * BT + wa0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (arow < a.Ca_real) atomicAdd(dWg + (size_t)arow * a.dw_row + col, acc[i][j][e]);
      }
  }
}

extern "C" int s2p_conv2d_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw,
                                int cin_real, int cout_real, int64_t dw_gstride, int splitk, void* stream) {
  if (!d || !x || !dy || !dw) S2P_FAIL(-1, "s2p_conv2d_wgrad: null pointer");
  if (d->dtype != S2P_F32 && d->dtype != S2P_BF16) S2P_FAIL(-1, "s2p_conv2d_wgrad: bad dtype");
  const int ce = d->dtype == S2P_F32 ? 4 : 8;
  if (d->Cin % ce || d->x_pitch % ce || d->y_pitch % ce || d->x_gstride % ce || d->y_gstride % ce)
    S2P_FAIL(-1, "s2p_conv2d_wgrad: channel counts / pitches must be multiples of %d", ce);
  const int T = d->KH * d->KW;
  if (T > 64) S2P_FAIL(-2, "s2p_conv2d_wgrad: more than 64 taps");
  const int cout_pad = (d->Cout + ce - 1) / ce * ce;
  if (cout_pad > d->y_pitch) S2P_FAIL(-1, "s2p_conv2d_wgrad: dy pitch < padded Cout");
  if (s2p_thin_applicable(d) && cout_real == d->Cout) return s2p_thin_wgrad(d, x, dy, dw, cin_real, (hipStream_t)stream);
  WgradArgs a{};
  a.dW = dw;
  a.istride = d->stride; a.reflect = d->reflect; a.T = T;
  if (!d->transposed) {        // A = dY on the output grid, B = X gathered
    a.A = dy; a.B = x;
    a.Qh = d->Ho; a.Qw = d->Wo; a.M = d->N * d->Ho * d->Wo;
    a.Ca = cout_pad; a.a_pitch = d->y_pitch; a.a_gstride = d->y_gstride; a.Ca_real = cout_real;
    a.Hi = d->H; a.Wi = d->W; a.Cb = d->Cin; a.b_pitch = d->x_pitch; a.b_gstride = d->x_gstride;
    a.Cb_real = cin_real;
  } else {                     // A = X on the input grid, B = dY gathered at iy*s + ky - pad
    a.A = x; a.B = dy;
    a.Qh = d->H; a.Qw = d->W; a.M = d->N * d->H * d->W;
    a.Ca = d->Cin; a.a_pitch = d->x_pitch; a.a_gstride = d->x_gstride; a.Ca_real = cin_real;
    a.Hi = d->Ho; a.Wi = d->Wo; a.Cb = cout_pad; a.b_pitch = d->y_pitch; a.b_gstride = d->y_gstride;
    a.Cb_real = cout_real;
  }
  a.NB = T * a.Cb; a.dw_row = T * a.Cb_real; a.dw_gstride = dw_gstride;
  for (int ky = 0; ky < d->KH; ++ky)
    for (int kx = 0; kx < d->KW; ++kx)
      a.tap[ky * d->KW + kx] = (((kx - d->pad) & 0xff) << 8) | ((ky - d->pad) & 0xff);
  const int BT = d->dtype == S2P_F32 ? 64 : 128;
  a.na_tiles = cdiv(a.Ca, BT); a.nb_tiles = cdiv(a.NB, BT);
  const int total_steps = cdiv(a.M, 32);
  if (splitk <= 0) {   // aim at ~2 blocks per CU
    int tiles = a.na_tiles * a.nb_tiles * d->groups;
    splitk = (512 + tiles - 1) / tiles;
    if (splitk > total_steps) splitk = total_steps;
    if (splitk < 1) splitk = 1;
  }
  a.steps_per_split = cdiv(total_steps, splitk);
  a.splitk = cdiv(total_steps, a.steps_per_split);
  dim3 grid(a.na_tiles * a.nb_tiles, d->groups, a.splitk);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == S2P_F32) hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad_kernel<__bf16>, grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("wgrad_kernel");
  return 0;
}
