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
  static_assert(NJ <= 32, "KS * Cout must fit the 32 MFMA rows");
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
      v = *(const u32x4*)(a.w + ((size_t)(co * T + ky * KS + kx) * CIN + s * 16 + hh * 8));
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
  const float bias_r[4] = {a.bias && CO > 0 ? a.bias[0] : 0.f, a.bias && CO > 1 ? a.bias[1] : 0.f,
                           a.bias && CO > 2 ? a.bias[2] : 0.f, a.bias && CO > 3 ? a.bias[3] : 0.f};
  const int ox = ox0 + wave * OPW + nl;         // output column of this lane (lanes nl < OPW of the lower half store)
  const bool store_lane = h == 0 && nl < OPW && ox < a.Wo;

  // relu / lrelu / none are  v > 0 ? v : v * ns ; tanh is selected per value (both forms are a few instructions)
  const bool is_tanh = a.act == S2P_ACT_TANH;
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  __bf16* yimg = a.y + (size_t)n * a.Ho * a.Wo * a.y_pitch;
  // shifted sum over kx of a finished accumulator + bias + activation + store of output row `orel` of the band.
  // Branch-free (the store is predicated): the compiler is free to interleave it with the MFMAs of the next input row.
  auto emit = [&](const f32x16& d, int orel) {
    float o[4] = {bias_r[0], bias_r[1], bias_r[2], bias_r[3]};
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

static bool rows_shape(const s2p_conv_desc* d) { return d->KH == 7 && d->Cout == 3 && d->Cin == 64; }

bool s2p_thin_rows_applicable(const s2p_conv_desc* d) {
  if (!(d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && d->stride == 1 && d->KH == d->KW && d->y_pitch == 8 &&
        d->x_pitch % 8 == 0))
    return false;
  if (!rows_shape(d)) return false;
  if (d->reflect && (d->pad >= d->H || d->pad >= d->W)) return false;
  if ((long long)d->H * d->W * d->x_pitch * 2 >= (1ll << 31)) return false;
  return d->Ho >= 1 && d->Wo >= 1;
}

int s2p_thin_rows_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act,
                      float slope, hipStream_t st) {
  RowsArgs a{};
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.x_pitch = d->x_pitch; a.Ho = d->Ho; a.Wo = d->Wo; a.y_pitch = d->y_pitch;
  a.pad = d->pad; a.reflect = d->reflect; a.act = act; a.slope = slope;
  a.img_bytes = (unsigned)((long long)d->H * d->W * d->x_pitch * 2);
  const int KS = d->KH;
  const int opw = 32 - (KS - 1);
  int nw = cdiv(d->Wo, opw);
  if (nw > 4) nw = 4;
  a.strip_out = opw * nw;
  a.nstrips = cdiv(d->Wo, a.strip_out);
  // row bands: one workgroup per CU when the batch allows (a band re-reads KS-1 halo rows)
  static const int diag = s2p_env_int("S2P_DIAG", 0);
  a.diag = diag;
  static const int target = s2p_env_int("S2P_ROWS_WGS", 256);   // diagnostics build only
  int nb = cdiv(target, (long long)d->N * a.nstrips);
  if (nb < 1) nb = 1;
  a.band = cdiv(d->Ho, nb);
  if (a.band < KS) a.band = KS < d->Ho ? KS : d->Ho;
  a.nbands = cdiv(d->Ho, a.band);
  a.nq = cdiv(a.band + KS - 1, KS);
  const int NS = d->Cin / 16;
  const size_t lds = (size_t)KS * NS * 1024;
  const dim3 grid((unsigned)(d->N * a.nbands * a.nstrips));
  hipLaunchKernelGGL((thin_rows_fwd_kernel<7, 3, 64>), grid, dim3(64 * nw), lds, st, a);
  S2P_CHECK_LAUNCH("thin_rows_fwd_kernel");
  return 0;
}
