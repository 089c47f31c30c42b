// Row-streaming MFMA kernels for the thin-Cout, large-kernel, stride-1 convolution of the generator's output layer
// (7x7, 64 -> 3, reflect pad, 84x84 .. 256x256), bf16.
//
// The spatially tiled kernel (thin_conv.hip) puts Cout (3, padded to 16) on one MFMA side: 19 % of the matrix core does
// work and every input position is re-read from LDS once per tap (49 x): the launch is LDS-bound at ~110 us for a
// 58 MB input.  Here the (kx, co) PAIRS are the MFMA rows instead (7 x 3 = 21 of 32), the padded input COLUMNS are the
// MFMA columns and the contraction runs over ci:
//     R[oy][(kx,co)][pc] = sum_{ky, ci} W[co][ky][kx][ci] * Xp[oy + ky][pc][ci]          (Xp = padded input)
//     y[oy][ox][co]      = act(bias[co] + sum_kx R[oy][(kx,co)][ox + kx])
// A wave owns 32 padded columns and walks DOWN the image: an input row is loaded ONCE, straight from global memory into
// B fragments (lane = column, 16 B of ci), and multiplied into KS rotating accumulators, one per output row it
// contributes to (ky = 0..KS-1); after its KS-th contribution an accumulator is complete, the 7-term shifted sum over
// kx is taken across lanes with ds_bpermute (a wave's 32 columns yield 32-(KS-1) outputs: neighbouring waves overlap by
// KS-1 columns, so waves never exchange data and the loop has no barrier), and the accumulator is reused for the next
// output row.  No input halo in LDS, no re-reads: each input element is fetched (KS-1+band)/band times (row bands); the
// weights sit in LDS in fragment order.
#include "s2p_common.h"

struct RowsArgs {
  const __bf16* x; const __bf16* w; const float* bias; __bf16* y;
  int N, H, W, x_pitch, Ho, Wo, y_pitch, pad, reflect, act;
  float slope;
  int band, nbands, nstrips, strip_out, nq;     // rows per band, bands per image, column strips, outputs per strip, row groups
  int flip;                                     // taps read in reverse order (dgrad: the adjoint's kernel is the flipped one)
  unsigned img_bytes;                           // bytes of one image of x
  int diag;                                     // diagnostics build: 1 = no in-loop loads, 2 = no emission (timing only)
};

// tanh for a bf16 result: 1 - 2 / (1 + e^2x)  (v_exp + v_rcp; saturates correctly at +-inf)
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

template <int KS, int CO, int CIN>
__global__ __launch_bounds__(256, 1) void thin_rows_fwd_kernel(const RowsArgs a) {
  constexpr int NS = CIN / 16;                  // k16 steps per input row
  constexpr int NJ = KS * CO;                   // MFMA rows in use: j = kx * CO + co
  constexpr int OPW = 32 - (KS - 1);            // outputs per wave
  constexpr int T = KS * KS;
  static_assert(NJ <= 32 && CO <= 8, "KS * Cout must fit the 32 MFMA rows, Cout one 16-byte chunk");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Wf = smem;                              // [KS][NS] fragments of 64 lanes x 16 B

  int b = blockIdx.x;
  const int strip = b % a.nstrips; b /= a.nstrips;
  const int bandi = b % a.nbands; const int n = b / a.nbands;
  const int oy0 = bandi * a.band, ox0 = strip * a.strip_out;

  for (int f = tid; f < KS * NS * 64; f += blockDim.x) {
    const int l = f & 63, frag = f >> 6, ky = frag / NS, s = frag - ky * NS;
    const int j = l & 31, hh = l >> 5;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (j < NJ) {
      const int kx = j / CO, co = j - kx * CO;
      const int t = a.flip ? T - 1 - (ky * KS + kx) : ky * KS + kx;
      v = *(const u32x4*)(a.w + ((size_t)(co * T + t) * CIN + s * 16 + hh * 8));
    }
    *(u32x4*)(Wf + (size_t)f * 16) = v;
  }
  __syncthreads();

  const int nl = lane & 31, h = lane >> 5;
  const int pc = wave * OPW + nl;               // padded column inside the strip (waves overlap by KS-1 columns)
  int ix = ox0 + pc - a.pad;
  bool colok = true;
  if (a.reflect) {
    ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
    ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);             // columns past the strip's last output: any valid address
  } else {
    colok = ix >= 0 && ix < a.W;
  }
  const unsigned OOB = 0x80000000u;
  const unsigned colbyte = colok ? (unsigned)((ix * a.x_pitch + h * 8) * 2) : OOB;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * a.H * a.W * a.x_pitch), 0,
                                                                 a.img_bytes, 0x00020000);
  auto load_row = [&](int i, u32x4 (&dst)[NS]) {
    int r = oy0 - a.pad + i;
    bool ok = colok;
    if (a.reflect) {
      r = r < 0 ? -r : (r >= a.H ? 2 * a.H - 2 - r : r);
      r = r < 0 ? 0 : (r >= a.H ? a.H - 1 : r);               // rows past the band's last output (never emitted)
    } else {
      ok = ok && r >= 0 && r < a.H;
      if (!ok) r = 0;
    }
    const unsigned off = ok ? (unsigned)(r * a.W * a.x_pitch * 2) + colbyte : OOB;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      dst[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(off + s * 32), 0, 0));
  };

  f32x16 acc[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const f32x16 zero = acc[0];

  u32x4 cur[NS], nx1[NS], nx2[NS];
  load_row(0, cur); load_row(1, nx1); load_row(2, nx2);
  float bias_r[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) bias_r[co] = a.bias ? a.bias[co] : 0.f;
  const int ox = ox0 + wave * OPW + nl;         // output column of this lane (lanes nl < OPW of the lower half store)
  const bool store_lane = h == 0 && nl < OPW && ox < a.Wo;

  // relu / lrelu / none are  v > 0 ? v : v * ns ; tanh is selected per value (both forms are a few instructions)
  const bool is_tanh = a.act == S2P_ACT_TANH;
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  __bf16* yimg = a.y + (size_t)n * a.Ho * a.Wo * a.y_pitch;
  // shifted sum over kx of a finished accumulator + bias + activation + store of output row `orel` of the band.
  // Branch-free (the store is predicated): the compiler is free to interleave it with the MFMAs of the next input row.
  auto emit = [&](const f32x16& d, int orel) {
    float o[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] = bias_r[co];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        // row j = kx*CO + co of D lives in register (j&3) + 4*(j>>3) of the lanes of half (j>>2)&1, column = lane & 31
        const int j = kx * CO + co;
        const int src = (((nl + kx) & 31) | (((j >> 2) & 1) << 5)) * 4;
        const float v = d[(j & 3) + 4 * (j >> 3)];
        o[co] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v)));
      }
    Chunk<__bf16> c; c.raw = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float t = fast_tanh(o[co]), l = o[co] > 0.f ? o[co] : o[co] * ns;
      c.set(co, is_tanh ? t : l);
    }
    const bool ok = store_lane && orel >= 0 && orel < a.band && oy0 + orel < a.Ho;
    if (ok) *(u32x4*)(yimg + ((size_t)(oy0 + orel) * a.Wo + ox) * a.y_pitch) = c.raw;
  };

  for (int q = 0; q < a.nq; ++q) {
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int i = q * KS + u;
      u32x4 nx3[NS];
      if (S2P_DIAGV(a) != 1) load_row(i + 3, nx3);
      else {
#pragma unroll
        for (int s = 0; s < NS; ++s) nx3[s] = cur[s];
      }
      // slot u was completed by the previous input row (output row i - KS) and is restarted by this row's ky = 0 MFMA
      const f32x16 done = acc[u];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bf16x8 bf = __builtin_bit_cast(bf16x8, cur[s]);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int slot = (u - ky + KS) % KS;                 // static: output row i - ky
          const bf16x8 af = *(const bf16x8*)(Wf + ((ky * NS + s) * 64 + lane) * 16);
          if (ky == 0 && s == 0) acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, zero, 0, 0, 0);
          else acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[slot], 0, 0, 0);
        }
      }
      if (S2P_DIAGV(a) != 2) emit(done, i - KS);
#pragma unroll
      for (int s = 0; s < NS; ++s) { cur[s] = nx1[s]; nx1[s] = nx2[s]; nx2[s] = nx3[s]; }
    }
  }
  emit(acc[0], a.nq * KS - KS);                                // the row completed by the last input row
}

// instantiated shapes: the generator's output conv (7x7, 64 -> 3) and, as the adjoint of VGG conv1_1 (3x3, 3 -> 64), a
// 3x3 64 -> 8 conv with flipped taps
static bool rows_shape(int KS, int cout, int cin) { return (KS == 7 && cout == 3 && cin == 64) || (KS == 3 && cout == 8 && cin == 64); }

bool s2p_thin_rows_applicable(const s2p_conv_desc* d) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->stride == 1 && d->KH == d->KW && d->y_pitch == 8 &&
        d->x_pitch % 8 == 0))
    return false;
  if (!rows_shape(d->KH, d->Cout, d->Cin)) return false;
  if (d->reflect && (d->pad >= d->H || d->pad >= d->W)) return false;
  if ((long long)d->H * d->W * d->x_pitch * 2 >= (1ll << 31)) return false;
  return d->Ho >= 1 && d->Wo >= 1;
}

// geometry in "forward" terms: x [N,H,W,x_pitch] (CIN channels read), y [N,Ho,Wo,8]
static int rows_launch(int KS, int cout, int cin, int N, int H, int W, int x_pitch, int Ho, int Wo, int pad, int reflect,
                       int flip, const void* x, const void* w, const float* bias, void* y, int act, float slope, hipStream_t st) {
  RowsArgs a{};
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
  a.N = N; a.H = H; a.W = W; a.x_pitch = x_pitch; a.Ho = Ho; a.Wo = Wo; a.y_pitch = 8;
  a.pad = pad; a.reflect = reflect; a.act = act; a.slope = slope; a.flip = flip;
  a.img_bytes = (unsigned)((long long)H * W * x_pitch * 2);
  static const int diag = s2p_env_int("S2P_DIAG", 0);
  a.diag = diag;
  const int opw = 32 - (KS - 1);
  int nw = cdiv(Wo, opw);
  if (nw > 4) nw = 4;
  a.strip_out = opw * nw;
  a.nstrips = cdiv(Wo, a.strip_out);
  // row bands: one workgroup per CU when the batch allows (a band re-reads KS-1 halo rows)
  static const int target = s2p_env_int("S2P_ROWS_WGS", 256);   // diagnostics build only
  int nb = cdiv(target, (long long)N * a.nstrips);
  if (nb < 1) nb = 1;
  a.band = cdiv(Ho, nb);
  if (a.band < KS) a.band = KS < Ho ? KS : Ho;
  a.nbands = cdiv(Ho, a.band);
  a.nq = cdiv(a.band + KS - 1, KS);
  const int NS = cin / 16;
  const size_t lds = (size_t)KS * NS * 1024;
  const dim3 grid((unsigned)(N * a.nbands * a.nstrips));
  if (KS == 7) hipLaunchKernelGGL((thin_rows_fwd_kernel<7, 3, 64>), grid, dim3(64 * nw), lds, st, a);
  else hipLaunchKernelGGL((thin_rows_fwd_kernel<3, 8, 64>), grid, dim3(64 * nw), lds, st, a);
  S2P_CHECK_LAUNCH("thin_rows_fwd_kernel");
  return 0;
}

int s2p_thin_rows_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                      float slope, hipStream_t st) {
  return rows_launch(d->KH, d->Cout, d->Cin, d->N, d->H, d->W, d->x_pitch, d->Ho, d->Wo, d->pad, d->reflect, 0, x, w, bias, y, act,
                     slope, st);
}

// dgrad of a stride-1 conv with a thin INPUT (Cin <= 8, e.g. VGG conv1_1: 3 -> 64): dx = conv(dy, flipped kernel), i.e. the
// same row-streaming kernel run over dy with w_bwd ([Cin_pad][T][Cout_pad]) as its forward weight and the taps reversed
bool s2p_thin_rows_dgrad_applicable(const s2p_conv_desc* d, int cout_pad) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && !d->reflect && d->stride == 1 && d->KH == d->KW &&
        d->x_pitch == 8 && d->Cin == 8 && d->y_pitch % 8 == 0))
    return false;
  if (!rows_shape(d->KH, 8, cout_pad)) return false;
  if ((long long)d->Ho * d->Wo * d->y_pitch * 2 >= (1ll << 31)) return false;
  return d->pad <= d->KH - 1;
}

int s2p_thin_rows_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* dx, int cout_pad, hipStream_t st) {
  return rows_launch(d->KH, 8, cout_pad, d->N, d->Ho, d->Wo, d->y_pitch, d->H, d->W, d->KH - 1 - d->pad, 0, 1, dy, w_bwd, nullptr, dx,
                     S2P_ACT_NONE, 0.f, st);
}

// ================================================================================================================
// Thin-INPUT convolutions with few taps (Cin <= 8 after padding, 3x3 / 4x4: VGG conv1_1, the 3 -> 1536 conditioning conv,
// the PatchGAN first layer), forward, bf16.  The implicit GEMM has K = taps x 8; one v_mfma_f32_32x32x16_bf16 step
// is TWO taps of one pixel column: the B fragment of lane (pixel, half) is the 16-byte pixel of tap 2s + half, loaded
// straight from global memory (32 consecutive pixels of a row = 512 contiguous bytes: coalesced, and the 7 MB input stays
// in L1 / L2 across its T re-reads) -- no LDS staging, no gather index tables.  The weights (64 output channels x K) sit
// in LDS in fragment order.  A wave owns 32 consecutive output pixels x 64 channels and transposes them through a private LDS strip so
// that every store instruction writes full 128-byte rows.
struct CinArgs {
  const __bf16* x; const __bf16* w; const float* bias; __bf16* y;
  int N, H, W, Ho, Wo, Cout, y_pitch, stride, pad, reflect, act;
  float slope;
  int M, tiles, x_bytes;
};

constexpr int CIN_SRS = 144;                 // staging-strip row: 64 channels x 2 B + 16
template <int KS>
__global__ __launch_bounds__(256) void thin_cin_fwd_kernel(const CinArgs a) {
  constexpr int T = KS * KS, NK = (T + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];         // [2 co tiles][NK] fragments of 64 lanes x 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co_base = blockIdx.y * 64;
  // weights -> fragment order.  The source is read linearly (16-byte chunk i = co * T + t of this block's 64 channels);
  // the odd half of the last K step (T odd) and channels past Cout stay zero.
  if (T & 1)
    for (int f = tid; f < 2 * 32; f += 256)
      *(u32x4*)(smem + ((size_t)((f >> 5) * NK + NK - 1) * 64 + 32 + (f & 31)) * 16) = (u32x4){0u, 0u, 0u, 0u};
  for (int i = tid; i < 64 * T; i += 256) {
    const int cl = i / T, t = i - cl * T, co = co_base + cl;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < a.Cout) v = *(const u32x4*)(a.w + ((size_t)co_base * T + i) * 8);
    *(u32x4*)(smem + ((size_t)((cl >> 5) * NK + (t >> 1)) * 64 + (t & 1) * 32 + (cl & 31)) * 16) = v;
  }
  __syncthreads();
  const int nl = lane & 31, h = lane >> 5;
  const unsigned OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  const int HoWo = a.Ho * a.Wo;
  const double rcpHW = s2p_rcp_f64(HoWo), rcpW = s2p_rcp_f64(a.Wo);
  // bias of this lane's 32 channels, once per workgroup (it was re-loaded for every tile: 32 dependent global loads in each epilogue)
  float bb[2][16];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co_base + ct * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
      bb[ct][i] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
  static_assert(NK <= 8, "all K steps of a tile are loaded as one batch");
  // every K step's pixel load of a tile is issued as ONE batch, and the batch of the wave's NEXT tile goes out before the current
  // tile's MFMAs and epilogue (a wave walks ~7 tiles: with the loads issued at the top of each tile the kernel was a chain of
  // load -> MFMA -> epilogue latencies, 58 us for the 3 -> 1536 conditioning conv against 15 us of HBM time for its output)
  auto issue_tile = [&](int tile, u32x4 (&bq)[NK], bool& mok, int& m) {
    m = tile * 32 + nl;
    mok = m < a.M;
    const int mm = mok ? m : 0;
    int rr, ox;
    const int n = divmod_rcp(mm, HoWo, rcpHW, rr), oy = divmod_rcp(rr, a.Wo, rcpW, ox);      // (two integer divisions per tile: ~50 of its ~300 VALU instructions)
    unsigned rowoff[KS], coloff[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      int iy = oy * a.stride + k - a.pad, ix = ox * a.stride + k - a.pad;
      bool yok = mok, xok = true;
      if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
      } else {
        yok = yok && iy >= 0 && iy < a.H; xok = ix >= 0 && ix < a.W;
      }
      rowoff[k] = yok ? (unsigned)(((n * a.H + iy) * a.W) * 16) : OOB;
      coloff[k] = xok ? (unsigned)(ix * 16) : OOB;
    }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const int t0 = 2 * s, t1 = 2 * s + 1 < T ? 2 * s + 1 : T - 1;           // the odd half of the last step has zero weights
      const unsigned o0 = rowoff[t0 / KS] | coloff[t0 % KS], o1 = rowoff[t1 / KS] | coloff[t1 % KS];
      const unsigned s0 = rowoff[t0 / KS] + coloff[t0 % KS], s1 = rowoff[t1 / KS] + coloff[t1 % KS];
      const unsigned off = h ? ((o1 & OOB) ? OOB : s1) : ((o0 & OOB) ? OOB : s0);
      bq[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0));
    }
  };
  const int tstride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  u32x4 cur[NK], nxt[NK];
  bool mok = false, mok_n = false;
  int m = 0, m_n = 0;
  if (tile < a.tiles) issue_tile(tile, cur, mok, m);
  for (; tile < a.tiles; tile += tstride) {
    const bool more = tile + tstride < a.tiles;                  // wave-uniform
    if (more) issue_tile(tile + tstride, nxt, mok_n, m_n);
    const char* wf = smem;
    asm volatile("" : "+v"(wf));                         // opaque per tile: the fragment reads stay in the loop
    f32x16 acc[2];                                       // start from the bias: the epilogue's 32 adds are the initialisation
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = bb[ct][e];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const bf16x8 bf = __builtin_bit_cast(bf16x8, cur[s]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const bf16x8 af = *(const bf16x8*)(wf + ((size_t)(ct * NK + s) * 64 + lane) * 16);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[ct], 0, 0, 0);
      }
    }
    // bias + activation, pack to bf16: pk[ct][q] = channels co_base + 32 ct + 8 q + 4 h + (0..3) of this lane's pixel
    u32x2 pk[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = lrelu_ns(acc[ct][4 * q + e], ns);
        }
        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        pk[ct][q] = __builtin_bit_cast(u32x2, o);
      }
    // transpose through the wave's private strip (32 pixel rows of 128 + 16 bytes) so that a store instruction writes FULL 128-byte
    // rows -- 8 lanes per pixel, 8 pixels per instruction (16-byte pieces 32 bytes apart wrote these 58-87-MB outputs at 2.3-2.8 TB/s).
    // DS operations of one wave execute in order: no wait between the passes.
    {
      char* strip = smem + 2 * NK * 1024 + wave * (32 * CIN_SRS);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x2*)(strip + nl * CIN_SRS + (32 * ct + 8 * q + 4 * h) * 2) = pk[ct][q];
      __builtin_amdgcn_wave_barrier();
      const int m0 = tile * 32, ch = lane & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 3) + 8 * r;
        const u32x4 out = *(const u32x4*)(strip + row * CIN_SRS + ch * 16);
        if (m0 + row < a.M && co_base + ch * 8 < a.Cout) *(u32x4*)(a.y + (size_t)(m0 + row) * a.y_pitch + co_base + ch * 8) = out;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // the prefetched batch becomes the current one
#pragma unroll
    for (int s = 0; s < NK; ++s) cur[s] = nxt[s];
    mok = mok_n; m = m_n;
  }
}

bool s2p_thin_cin_fwd_applicable(const s2p_conv_desc* d, int act, int epi) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->Cin == 8 && d->x_pitch == 8 && d->KH == d->KW))
    return false;
  if (!(d->KH == 3 || d->KH == 4)) return false;      // (7x7: 25 K steps -- the register-staged gather kernel is faster there)
  if (epi != S2P_EPI_STORE || !(act == S2P_ACT_NONE || act == S2P_ACT_RELU || act == S2P_ACT_LRELU)) return false;
  if (d->Cout % 8 || d->Cout < 32 || d->y_pitch % 8) return false;
  if (d->reflect && (d->pad >= d->H || d->pad >= d->W)) return false;
  return (long long)d->N * d->H * d->W * 16 < (1ll << 31) && (long long)d->N * d->Ho * d->Wo < (1ll << 31) - 64;
}

int s2p_thin_cin_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                     hipStream_t st) {
  CinArgs a{};
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_pitch = d->y_pitch;
  a.stride = d->stride; a.pad = d->pad; a.reflect = d->reflect; a.act = act; a.slope = slope;
  a.M = d->N * d->Ho * d->Wo; a.tiles = cdiv(a.M, 32);
  a.x_bytes = (int)((long long)d->N * d->H * d->W * 16);
  const int T = d->KH * d->KW, NK = (T + 1) / 2;
  const int ncb = cdiv(d->Cout, 64);
  int gx = cdiv(a.tiles, 4);
  const int cap = 768 / ncb > 32 ? 768 / ncb : 32;            // ~3 workgroups per CU: each stages 2*NK KiB of weights once
  if (gx > cap) gx = cap;
  const dim3 grid(gx, ncb);
  const size_t lds = (size_t)2 * NK * 1024 + 4 * 32 * CIN_SRS;      // weight fragments + one staging strip per wave
  if (d->KH == 4) hipLaunchKernelGGL(thin_cin_fwd_kernel<4>, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(thin_cin_fwd_kernel<3>, grid, dim3(256), lds, st, a);
  S2P_CHECK_LAUNCH("thin_cin_fwd_kernel");
  return 0;
}


// ================================================================================================================
// 7x7 stride-1 convolutions with at most FOUR real input channels (the generator's stem 3 -> 64, and -- over dY with flipped taps --
// the dgrad of its 64 -> 3 output conv), forward, bf16 (round 4).  The implicit-GEMM kernels contract 8 channels per tap (the 16-byte
// chunk of a thin tensor: 5 of them zero): 13 MFMA k-steps per tile, 62 us for the stem against 10 us of HBM time for its output.
// Here the input is first copied into a PRE-PADDED tensor with a 4-channel pitch ([N][Ho + 6][Wo + 8][4], 8 bytes per pixel;
// reflect or zero padding is resolved by that copy: 4 MB), in which the four taps kx .. kx + 3 of a pixel row are 32 CONTIGUOUS
// bytes: one v_mfma_f32_32x32x16_bf16 step contracts four taps (the lane halves take two each, 16 bytes at an 8-byte-aligned
// address straight from global memory), a tap row is two steps (kx = 7 has zero weights), the conv 14 steps of half the
// K -- and no border logic in the kernel.  Otherwise the kernel is thin_cin_fwd_kernel: weights in LDS in fragment order, a wave
// owns 32 consecutive output pixels x 64 channels, the lane halves swap 8-byte pieces so that every store is a full chunk.
struct Thin4Args {
  const __bf16* xp; const __bf16* w; const float* bias; __bf16* y;
  int N, Ho, Wo, Hp, Wp, Cout, y_pitch, act, w_row, w_tap, flip;      // w[(co * w_row + tap * w_tap + ch)]: forward [Cout][49][8], dgrad use [Cin][49][cout_pad]
  float slope;
  int M, tiles, xp_bytes;
  const u32x4* wfrag;                                                 // weights in MFMA fragment order (pad4_kernel writes them)
};
struct Pad4Args {
  const __bf16* x; __bf16* xp; int N, H, W, x_pitch, Hp, Wp, pad, reflect;
  // the blocks behind the first nb_pad ones pack the weights into fragment order: [co block of 64][2 co tiles][14 steps][64 lanes] x 16 B
  int nb_pad; const __bf16* w; u32x4* wfrag; int Cout, w_row, w_tap, flip, ncb;
};

// xp[n][yp][xq][0..3] = x[n][yp - pad][xq - pad][0..3] (reflected or zero outside; columns >= W + 2 pad are zero)
// ... and, in the blocks behind those, the weights in fragment order (round 5): every thin4_fwd workgroup used to gather its 28 KB of
// fragments from the packed weights with 2-byte loads -- 56 dependent-latency loads per thread in front of the first MFMA of each of
// the 1 024 workgroups, most of the kernel's 48 us; now one block per 256 fragments does it once and the workgroups copy 16-byte pieces.
__global__ __launch_bounds__(256) void pad4_kernel(const Pad4Args a) {
  if ((int)blockIdx.x >= a.nb_pad) {
    constexpr int NK = 14;
    const int f = ((int)blockIdx.x - a.nb_pad) * 256 + threadIdx.x;
    if (f >= a.ncb * 2 * NK * 64) return;
    // A[m = co][k = 8 hh + e] of step (ky, j): tap kx = 4 j + 2 hh + (e >> 2), channel e & 3
    const int ln = f & 63, s = (f >> 6) % NK, ct = (f / (64 * NK)) & 1, cb = f / (2 * 64 * NK);
    const int co = cb * 64 + ct * 32 + (ln & 31), hh = ln >> 5, ky = s >> 1, j = s & 1;
    unsigned short e8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kx = 4 * j + 2 * hh + (e >> 2), ch = e & 3;
      unsigned short v = 0;
      if (co < a.Cout && kx < 7) {
        const int t = ky * 7 + kx, tap = a.flip ? 48 - t : t;
        v = *(const unsigned short*)(a.w + (size_t)co * a.w_row + (size_t)tap * a.w_tap + ch);
      }
      e8[e] = v;
    }
    a.wfrag[f] = (u32x4){(unsigned)e8[0] | ((unsigned)e8[1] << 16), (unsigned)e8[2] | ((unsigned)e8[3] << 16),
                         (unsigned)e8[4] | ((unsigned)e8[5] << 16), (unsigned)e8[6] | ((unsigned)e8[7] << 16)};
    return;
  }
  const long long total = (long long)a.N * a.Hp * a.Wp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)a.nb_pad * 256) {
    const int xq = (int)(i % a.Wp);
    const long long r = i / a.Wp;
    const int yp = (int)(r % a.Hp), n = (int)(r / a.Hp);
    int iy = yp - a.pad, ix = xq - a.pad;
    bool ok = xq < a.W + 2 * a.pad;
    if (a.reflect) {
      iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
      ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    } else {
      ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    }
    u32x2 v = {0u, 0u};
    if (ok) v = *(const u32x2*)(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_pitch);
    *(u32x2*)(a.xp + (size_t)i * 4) = v;
  }
}

// Round 5: as in thin_cin_fwd_kernel, the next tile's 14 loads are issued before the current tile's MFMAs, and the 32 x 64 output tile
// goes through a wave-private LDS strip so that every store instruction writes full 128-byte rows (the lane halves' 16-byte pieces,
// 128 bytes apart, wrote the stem's 58-MB output at 1.3 TB/s: 48 us).
__global__ __launch_bounds__(512) void thin4_fwd_kernel(const Thin4Args a) {      // 8 waves: one workgroup per CU stages the 28 KB of fragments once
  constexpr int NK = 14;                                              // K steps: (ky, half row): taps kx = 4 j .. 4 j + 3
  __shared__ __attribute__((aligned(16))) char smem[2 * NK * 1024 + 8 * 32 * CIN_SRS];   // [2 co tiles][NK] fragments of 64 lanes x 16 B + a strip per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co_base = blockIdx.y * 64;
  // this 64-channel block's weight fragments (pad4_kernel packed them): [2 co tiles][NK][64 lanes] x 16 B
  for (int f = tid; f < 2 * NK * 64; f += 512) *(u32x4*)(smem + (size_t)f * 16) = a.wfrag[(size_t)blockIdx.y * (2 * NK * 64) + f];
  __syncthreads();
  const int nl = lane & 31, h = lane >> 5;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.xp, 0, a.xp_bytes, 0x00020000);
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  const bool tanh_act = a.act == S2P_ACT_TANH;
  const int HoWo = a.Ho * a.Wo;
  float bb[2][16];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co_base + ct * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
      bb[ct][i] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
  auto issue_tile = [&](int tile, u32x4 (&bq)[NK]) {
    const int m = tile * 32 + nl;
    const int mm = m < a.M ? m : 0;
    const int n = mm / HoWo, rr = mm - n * HoWo, oy = rr / a.Wo, ox = rr - oy * a.Wo;
    // byte offset of (row oy + ky, column ox + 2 h): + ky * Wp * 8, + j * 32
    const unsigned base = (unsigned)((((size_t)n * a.Hp + oy) * a.Wp + ox + 2 * h) * 8);
#pragma unroll
    for (int s = 0; s < NK; ++s)
      bq[s] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(base + (unsigned)((s >> 1) * a.Wp * 8 + (s & 1) * 32)), 0, 0));
  };
  const int tstride = gridDim.x * 8;
  int tile = blockIdx.x * 8 + wave;
  u32x4 cur[NK], nxt[NK];
  if (tile < a.tiles) issue_tile(tile, cur);
  for (; tile < a.tiles; tile += tstride) {
    const bool more = tile + tstride < a.tiles;                  // wave-uniform
    if (more) issue_tile(tile + tstride, nxt);
    const char* wf = smem;
    asm volatile("" : "+v"(wf));                         // opaque per tile: the fragment reads stay in the loop
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const bf16x8 bf = __builtin_bit_cast(bf16x8, cur[s]);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const bf16x8 af = *(const bf16x8*)(wf + ((size_t)(ct * NK + s) * 64 + lane) * 16);
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[ct], 0, 0, 0);
      }
    }
    // bias + activation, pack to bf16: pk[ct][q] = channels co_base + 32 ct + 8 q + 4 h + (0..3) of this lane's pixel
    u32x2 pk[2][4];
    auto pack = [&](auto f) {                               // (the activation is chosen by ONE uniform branch per tile: as a select the
#pragma unroll                                              //  compiler evaluated tanhf for all 32 outputs of every lane, used or not)
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = f(acc[ct][4 * q + e] + bb[ct][4 * q + e]);
          const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          pk[ct][q] = __builtin_bit_cast(u32x2, o);
        }
    };
    if (tanh_act) pack([](float t) { return tanhf(t); });
    else pack([ns](float t) { return lrelu_ns(t, ns); });
    // transpose through the wave's private strip (32 pixel rows of 128 + 16 bytes): 8 lanes per pixel, 8 pixels per store instruction.
    // DS operations of one wave execute in order: no wait between the passes.
    {
      char* strip = smem + 2 * NK * 1024 + wave * (32 * CIN_SRS);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) *(u32x2*)(strip + nl * CIN_SRS + (32 * ct + 8 * q + 4 * h) * 2) = pk[ct][q];
      __builtin_amdgcn_wave_barrier();
      const int m0 = tile * 32, ch = lane & 7;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 3) + 8 * r;
        const u32x4 out = *(const u32x4*)(strip + row * CIN_SRS + ch * 16);
        if (m0 + row < a.M && co_base + ch * 8 < a.Cout) *(u32x4*)(a.y + (size_t)(m0 + row) * a.y_pitch + co_base + ch * 8) = out;
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int s = 0; s < NK; ++s) cur[s] = nxt[s];
  }
}

// forward use: 7x7 stride 1 pad 3 (reflect or zero), <= 4 real input channels in an 8-pitch tensor, Cout a multiple of 8
bool s2p_thin4_fwd_applicable(const s2p_conv_desc* d, int act, int epi) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->Cin == 8 && d->x_pitch == 8 && d->KH == 7 && d->KW == 7 &&
        d->stride == 1 && d->pad == 3 && d->cin_real >= 1 && d->cin_real <= 4))
    return false;
  if (epi != S2P_EPI_STORE || act == S2P_ACT_SWISH) return false;
  if (d->Cout % 8 || d->Cout < 32 || d->y_pitch % 8 || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->reflect && (d->H < 4 || d->W < 4)) return false;
  return (long long)d->N * (d->H + 6) * (d->W + 8) * 8 < (1ll << 31) && (long long)d->N * d->H * d->W < (1ll << 31) - 64;
}
static size_t thin4_frag_bytes(int cout) { return (size_t)cdiv(cout, 64) * 2 * 14 * 1024; }
size_t s2p_thin4_fwd_ws_bytes(const s2p_conv_desc* d) { return (size_t)d->N * (d->H + 6) * (d->W + 8) * 8 + 16 + thin4_frag_bytes(d->Cout); }

static int thin4_launch(Thin4Args& a, Pad4Args& p, hipStream_t st) {
  const long long total = (long long)p.N * p.Hp * p.Wp;
  int pb = (int)((total + 255) / 256); if (pb > 4096) pb = 4096;
  const int ncb = cdiv(a.Cout, 64);
  // the padded copy is followed (16-byte aligned: total * 8 bytes) by the weight fragments
  p.nb_pad = pb; p.w = a.w; p.wfrag = (u32x4*)((char*)p.xp + (((size_t)total * 8 + 15) & ~(size_t)15));
  p.Cout = a.Cout; p.w_row = a.w_row; p.w_tap = a.w_tap; p.flip = a.flip; p.ncb = ncb;
  a.wfrag = p.wfrag;
  hipLaunchKernelGGL(pad4_kernel, dim3(pb + cdiv(ncb * 2 * 14 * 64, 256)), dim3(256), 0, st, p);
  S2P_CHECK_LAUNCH("pad4_kernel");
  a.M = a.N * a.Ho * a.Wo; a.tiles = cdiv(a.M, 32);
  a.xp_bytes = (int)(total * 8);
  int gx = cdiv(a.tiles, 8);
  const int cap = 256 / ncb > 32 ? 256 / ncb : 32;               // one 8-wave workgroup per CU (180 VGPRs: two waves per SIMD)
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(thin4_fwd_kernel, dim3(gx, ncb), dim3(512), 0, st, a);
  S2P_CHECK_LAUNCH("thin4_fwd_kernel");
  return 0;
}

int s2p_thin4_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope, void* ws,
                  size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < s2p_thin4_fwd_ws_bytes(d)) S2P_FAIL(-1, "s2p_conv2d_fwd: this 7x7 thin-input conv needs %zu bytes of workspace (s2p_conv2d_fwd_workspace)", s2p_thin4_fwd_ws_bytes(d));
  Pad4Args p{(const __bf16*)x, (__bf16*)ws, d->N, d->H, d->W, d->x_pitch, d->H + 6, d->W + 8, 3, d->reflect};
  Thin4Args a{};
  a.xp = (const __bf16*)ws; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo; a.Hp = p.Hp; a.Wp = p.Wp; a.Cout = d->Cout; a.y_pitch = d->y_pitch; a.act = act; a.slope = slope;
  a.w_row = 49 * 8; a.w_tap = 8; a.flip = 0;
  return thin4_launch(a, p, st);
}

// dgrad use: the 7x7 stride-1 conv has <= 4 OUTPUT channels (dY is the thin tensor, pitch 8); the produced tensor dx has Cin channels.
// Reflect-padded convs produce the PADDED grid (H + 6) x (W + 6) (the full correlation: dY zero-padded by 6), zero-padded ones H x W.
bool s2p_thin4_dgrad_applicable(const s2p_conv_desc* d, int cout_pad) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->KH == 7 && d->KW == 7 && d->stride == 1 && d->pad == 3 &&
        d->Cout >= 1 && d->Cout <= 4 && cout_pad == 8 && d->y_pitch == 8 && d->Ho == d->H && d->Wo == d->W))
    return false;
  if (d->Cin % 8 || d->Cin < 32 || d->x_pitch % 8) return false;
  const int e = d->reflect ? 6 : 0;
  return (long long)d->N * (d->H + e + 6) * (d->W + e + 8) * 8 < (1ll << 31) && (long long)d->N * (d->H + e) * (d->W + e) < (1ll << 31) - 64;
}
size_t s2p_thin4_dgrad_ws_bytes(const s2p_conv_desc* d) {
  const int e = d->reflect ? 6 : 0;
  return (size_t)d->N * (d->H + e + 6) * (d->W + e + 8) * 8 + 16 + thin4_frag_bytes(d->Cin);
}
int s2p_thin4_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* dx, int cout_pad, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < s2p_thin4_dgrad_ws_bytes(d)) S2P_FAIL(-1, "s2p_conv2d_dgrad: this 7x7 thin-output conv needs %zu bytes of workspace (s2p_conv2d_dgrad_workspace)", s2p_thin4_dgrad_ws_bytes(d));
  const int e = d->reflect ? 6 : 0;                   // produced grid: (H + e) x (W + e); dY padded by 3 + e / 2 ... = (6 | 3) on every side
  const int pad = d->reflect ? 6 : 3;
  Pad4Args p{(const __bf16*)dy, (__bf16*)ws, d->N, d->H, d->W, d->y_pitch, d->H + e + 6, d->W + e + 8, pad, 0};
  Thin4Args a{};
  a.xp = (const __bf16*)ws; a.w = (const __bf16*)w_bwd; a.bias = nullptr; a.y = (__bf16*)dx;
  a.N = d->N; a.Ho = d->H + e; a.Wo = d->W + e; a.Hp = p.Hp; a.Wp = p.Wp; a.Cout = d->Cin; a.y_pitch = d->x_pitch; a.act = S2P_ACT_NONE; a.slope = 0.f;
  a.w_row = 49 * cout_pad; a.w_tap = cout_pad; a.flip = 1;
  return thin4_launch(a, p, st);
}
