// Weight gradient of the PatchGAN logit heads (Cout = 1, 4x4, stride 1: 512 -> 1 on 128 x 13 x 13 and 128 x 8 x 8 maps).
//
// The generic thin kernel keeps (tap, channel chunk) per thread and therefore re-reads the activation once per tap
// (16 x 22 MB through L2: 204 us for 0.7 GFLOP).  Here the ACTIVATION is stationary: a thread owns one 16-byte channel
// chunk of one input pixel, loads it ONCE, and updates all KH*KW tap accumulators from the (tiny, cache-resident) dY map:
//     dW[0][ky][kx][ci] = sum_{n,iy,ix} x[n,iy,ix,ci] * dY[n, iy + pad - ky, ix + pad - kx]
// A workgroup folds its pixel rows through LDS and stores ONE partial [T][Cin] to the workspace; a second tiny kernel adds
// the partials in workgroup order (and the bias gradient = sum of dY): no atomics, bitwise reproducible.
#include "s2p_common.h"

struct HeadArgs {
  const __bf16* x; const __bf16* dy; float* dw; float* db; float* part;
  int N, H, W, Cin, x_pitch, Ho, Wo, y_pitch, KH, KW, pad;
  int cin_real, nchunk, rpb, npix, iters, G;
  int wpi;                          // streaming form: workgroups per image
};

template <int KH_, int KW_>        // compile-time kernel size (4x4 for the PatchGAN heads; 0 = run-time, up to 16 taps)
__global__ __launch_bounds__(256) void head_wgrad_kernel(const HeadArgs a) {
  constexpr int MAXT = KH_ > 0 ? KH_ * KW_ : 16;
  __shared__ float red[256 * 8];
  const int KW = KW_ > 0 ? KW_ : a.KW;
  const int T = KH_ > 0 ? KH_ * KW_ : a.KH * a.KW;
  const int ch = threadIdx.x % a.nchunk, row = threadIdx.x / a.nchunk;
  float acc[MAXT][8];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  const int HW = a.H * a.W;
  for (int it = 0; it < a.iters; ++it) {
    const int pi = (it * a.G + blockIdx.x) * a.rpb + row;
    if (pi >= a.npix || row >= a.rpb) continue;
    const int n = pi / HW, rr = pi - n * HW, iy = rr / a.W, ix = rr - iy * a.W;
    Chunk<__bf16> xv;
    xv.raw = *(const u32x4*)(a.x + (size_t)pi * a.x_pitch + ch * 8);
    const __bf16* dyn = a.dy + (size_t)n * a.Ho * a.Wo * a.y_pitch;
    // all tap values of dY first (clamped addresses + a validity select: no branches, 16 loads in flight), then the FMAs
    float d[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int ky = t / KW, kx = t - ky * KW;                 // compile-time when KW_ > 0
      const int oy = iy + a.pad - ky, ox = ix + a.pad - kx;
      const bool ok = t < T && oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
      const int oyc = oy < 0 ? 0 : (oy >= a.Ho ? a.Ho - 1 : oy), oxc = ox < 0 ? 0 : (ox >= a.Wo ? a.Wo - 1 : ox);
      const float v = (float)dyn[(size_t)(oyc * a.Wo + oxc) * a.y_pitch];
      d[t] = ok ? v : 0.f;
    }
    float xf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[e] = xv.get(e);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[t][e] = __builtin_fmaf(d[t], xf[e], acc[t][e]);
  }
  // fold the rpb pixel rows of the workgroup, one tap at a time (rpb * Cin floats of LDS), in row order
  float* part = a.part + (size_t)blockIdx.x * T * a.Cin;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < T) {
      __syncthreads();
      if (row < a.rpb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(row * a.nchunk + ch) * 8 + e] = acc[t][e];
      }
      __syncthreads();
      for (int c = threadIdx.x; c < a.Cin; c += 256) {
        float s = 0.f;
        for (int r = 0; r < a.rpb; ++r) s += red[r * a.Cin + c];
        part[t * a.Cin + c] = s;
      }
    }
  }
}

// Streaming form (round 5; Cin = 512, 4x4).  The kernel above keeps ONE 16-byte load in flight per thread (128 accumulators: one
// wave per SIMD), 8 KB per CU -- 0.8 TB/s on a 22-MB activation.  Here the activation reaches the thread through LDS-DMA instead:
// a wave owns every fourth pixel of its workgroup's range, a pixel's 512 channels are ONE 1-KB DMA piece (lane l = channels
// 8 l .. 8 l + 7), and the wave keeps R = 16 pieces in flight in a private ring (no VGPRs, no barrier: each lane reads back the 16
// bytes it fetched itself, behind a counted vmcnt).  The logit-gradient map of the image sits in LDS (fp32); a workgroup is a
// contiguous pixel range of one image, folds its four waves in wave order and stores one partial.  Same partial / reduce scheme.
template <int R>
__global__ __launch_bounds__(256, 2) void head_wgrad_stream_kernel(const HeadArgs a) {
  constexpr int T = 16, KW = 4;
  __shared__ __attribute__((aligned(1024))) char ring[4 * R * 1024];
  __shared__ float hmap[400];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x / a.wpi, part_i = blockIdx.x - n * a.wpi;
  const int HW = a.H * a.W, nmap = a.Ho * a.Wo;
  const int per = (HW + a.wpi - 1) / a.wpi, p0 = part_i * per, p1 = p0 + per < HW ? p0 + per : HW;
  for (int i = tid; i < nmap; i += 256) hmap[i] = (float)a.dy[((size_t)n * nmap + i) * a.y_pitch];
  __syncthreads();
  const i32x4 rs = s2p_make_rsrc(a.x + (size_t)n * HW * a.x_pitch, (unsigned)HW * (unsigned)a.x_pitch * 2u);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(ring)) + (unsigned)wave * (R * 1024);
  const int npx = p1 - p0 - wave > 0 ? (p1 - p0 - wave + 3) >> 2 : 0;      // pixels p0 + wave + 4 i of this wave
  auto issue = [&](int i) {                                      // (beyond the range: an out-of-range offset, zeros, same count)
    const int p = p0 + wave + 4 * i;
    s2p_dma16(rs, ring_lds + (unsigned)((i % R) * 1024), i < npx ? p * a.x_pitch * 2 + lane * 16 : (int)0x80000000);
  };
#pragma unroll
  for (int i = 0; i < R; ++i) issue(i);
  float acc[T][8];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  for (int i = 0; i < npx; ++i) {
    S2P_WAIT_VMCNT(R - 1);                                       // the oldest piece (pixel i) has landed
    Chunk<__bf16> xv;
    xv.raw = *(const u32x4*)(ring + (wave * R + (i % R)) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // ... and is in registers before its slot is refilled
    issue(i + R);
    const int p = p0 + wave + 4 * i, iy = p / a.W, ix = p - iy * a.W;
    float d[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ky = t / KW, kx = t - ky * KW;
      const int oy = iy + a.pad - ky, ox = ix + a.pad - kx;
      const bool ok = oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo;
      d[t] = ok ? hmap[oy * a.Wo + ox] : 0.f;
    }
    float xf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[e] = xv.get(e);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[t][e] = __builtin_fmaf(d[t], xf[e], acc[t][e]);
  }
  S2P_WAIT_VMCNT(0);
  __syncthreads();                                               // every ring is quiet: it becomes the fold area [4 waves][512]
  float* red = (float*)ring;
  float* part = a.part + (size_t)blockIdx.x * T * a.Cin;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave * 512 + lane * 8 + e] = acc[t][e];
    __syncthreads();
    for (int c = tid; c < 512; c += 256) part[t * a.Cin + c] = ((red[c] + red[512 + c]) + red[1024 + c]) + red[1536 + c];
  }
}

// dw[t][ci] += sum over the G partials (workgroup order).  64 outputs per workgroup; the G partials of an output are
// summed by 16 threads (a sixteenth each, 8 loads in flight) and combined in segment order: the result does not depend
// on timing, and the kernel is not a chain of G dependent L2 round trips.
__global__ __launch_bounds__(1024) void head_wgrad_reduce_kernel(const HeadArgs a) {
  const int T = a.KH * a.KW;
  constexpr int NSEG = 16;
  __shared__ float red[1024];
  if (blockIdx.x < gridDim.x - 1) {
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;                        // (tap, ci) over the padded channel count
    const int per = (a.G + NSEG - 1) / NSEG;
    const int g0 = sg * per, g1 = g0 + per < a.G ? g0 + per : a.G;
    float s = 0.f;
    if (i < T * a.Cin) {
      const float* p = a.part + i;
      const size_t st = (size_t)T * a.Cin;
      int g = g0;
      for (; g + 8 <= g1; g += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(g + u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; g < g1; ++g) s += p[(size_t)g * st];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (sg == 0 && i < T * a.Cin) {
      const int t = i / a.Cin, c = i - t * a.Cin;
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < NSEG; ++k) tot += red[64 * k + o];
      if (c < a.cin_real) a.dw[t * a.cin_real + c] += tot;
    }
    return;
  }
  // last workgroup: bias gradient = sum of the dY map, fixed order
  if (!a.db) return;
  const long long cnt = (long long)a.N * a.Ho * a.Wo;
  float s = 0.f;
  for (long long p = threadIdx.x; p < cnt; p += 1024) s += (float)a.dy[(size_t)p * a.y_pitch];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 1024; ++k) t += red[k];
    a.db[0] += t;
  }
}

bool s2p_head_wgrad_supported(const s2p_conv_desc* d, int cin_real, int cout_real) {
  if (d->dtype != S2P_BF16 || d->transposed || d->reflect || d->groups != 1 || d->stride != 1) return false;
  if (d->Cout != 1 || cout_real != 1 || d->KH * d->KW > 16) return false;
  const int nchunk = d->Cin / 8;
  if (d->Cin % 8 || nchunk < 1 || nchunk > 256 || 256 % nchunk) return false;
  if (cin_real > d->Cin) return false;
  return true;
}

// streaming form: Cin = 512, 4x4, the logit map of an image fits the LDS table; workgroups per image so that ~512 run
static int head_stream_wpi(const s2p_conv_desc* d) {
  if (S2P_DIAG_SWITCH(12) || d->Cin != 512 || d->KH != 4 || d->KW != 4 || d->Ho * d->Wo > 400 || d->x_pitch % 8) return 0;
  if ((long long)d->H * d->W * d->x_pitch * 2 >= (1ll << 31)) return 0;
  int wpi = (512 + d->N - 1) / d->N;
  const int HW = d->H * d->W;
  if (wpi > (HW + 7) / 8) wpi = (HW + 7) / 8;                     // at least 8 pixels per workgroup
  return wpi < 1 ? 1 : wpi;
}

static int head_groups(const s2p_conv_desc* d) {
  if (const int wpi = head_stream_wpi(d)) return d->N * wpi;
  const int rpb = 256 / (d->Cin / 8);
  const long long npix = (long long)d->N * d->H * d->W;
  long long g = (npix + (long long)rpb * 8 - 1) / ((long long)rpb * 8);           // >= 8 pixels per thread
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return (int)g;
}

size_t s2p_head_wgrad_workspace(const s2p_conv_desc* d) {
  return (size_t)head_groups(d) * d->KH * d->KW * d->Cin * sizeof(float);
}

int s2p_head_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real,
                   void* workspace, size_t workspace_bytes, hipStream_t st) {
  HeadArgs a{};
  a.x = (const __bf16*)x; a.dy = (const __bf16*)dy; a.dw = dw; a.db = db; a.part = (float*)workspace;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_pitch = d->x_pitch; a.Ho = d->Ho; a.Wo = d->Wo;
  a.y_pitch = d->y_pitch; a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.cin_real = cin_real;
  a.nchunk = d->Cin / 8; a.rpb = 256 / a.nchunk; a.npix = d->N * d->H * d->W;
  a.G = head_groups(d);
  a.iters = cdiv(a.npix, (long long)a.G * a.rpb);
  if (!workspace || workspace_bytes < s2p_head_wgrad_workspace(d)) S2P_FAIL(-1, "s2p_head_wgrad: workspace too small");
  a.wpi = head_stream_wpi(d);
  if (a.wpi) hipLaunchKernelGGL((head_wgrad_stream_kernel<16>), dim3(a.G), dim3(256), 0, st, a);
  else if (a.KH == 4 && a.KW == 4) hipLaunchKernelGGL((head_wgrad_kernel<4, 4>), dim3(a.G), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((head_wgrad_kernel<0, 0>), dim3(a.G), dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("head_wgrad_kernel");
  const int T = a.KH * a.KW;
  hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(T * a.Cin, 64) + 1), dim3(1024), 0, st, a);
  S2P_CHECK_LAUNCH("head_wgrad_reduce_kernel");
  return 0;
}

// ================================================================================================================
// Forward of the PatchGAN logit heads (Cout = 1, 4x4, stride 1, 512 input channels).  The generic thin kernel spreads
// an OUTPUT pixel over the lanes and re-reads the activation once per tap (16 x through L2: 26 us for an 11 MB map).
// Here the ACTIVATION is stationary: a wave walks one input row, a lane owns one 16-byte channel chunk (64 lanes = the
// 1 KiB of a pixel, one coalesced load) and holds its chunk of all 16 tap weights in registers; per pixel 16 partial dot
// products are reduced over the wave by a transposing butterfly (15 + 2 shuffles, not 16 x 6) and lane t adds tap t's
// total to the output cell it belongs to (oy = iy + pad - ky, ox = ix + pad - kx) in the wave's LDS patch -- in pixel
// order, so the sum is reproducible.  A workgroup is one image (one wave per input row); after a barrier each output
// pixel adds its <= 4 row patches, the bias and the activation.
struct HeadFwdArgs {
  const __bf16* x; const __bf16* w; const float* bias; __bf16* y;
  int N, H, W, x_pitch, Ho, Wo, y_pitch, pad, act;
  float slope;
};

__global__ __launch_bounds__(1024) void head_fwd_kernel(const HeadFwdArgs a) {
  constexpr int KS = 4, T = 16, MAXW = 24;
  __shared__ float patch[16][KS][MAXW];                  // [input row][ky][ox]
  const int tid = threadIdx.x, lane = tid & 63;
  const int iy = __builtin_amdgcn_readfirstlane(tid >> 6);     // one wave per input row
  const int n = blockIdx.x;
  for (int i = tid; i < 16 * KS * MAXW; i += blockDim.x) (&patch[0][0][0])[i] = 0.f;
  u32x4 wv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) wv[t] = *(const u32x4*)(a.w + (size_t)t * 512 + lane * 8);
  __syncthreads();
  const __bf16* xrow = a.x + ((size_t)(n * a.H + iy) * a.W) * a.x_pitch + lane * 8;
  const int kyl = (lane & 15) >> 2, kxl = lane & 3;      // the tap this lane ends up holding
  u32x4 xv = *(const u32x4*)xrow;
  for (int ix = 0; ix < a.W; ++ix) {
    const u32x4 cur = xv;
    if (ix + 1 < a.W) xv = *(const u32x4*)(xrow + (size_t)(ix + 1) * a.x_pitch);
    float d[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(2))) __bf16, (unsigned)cur[i]),
                                            __builtin_bit_cast(__attribute__((ext_vector_type(2))) __bf16, (unsigned)wv[t][i]), s, false);
      d[t] = s;
    }
    // transposing butterfly over lane bits 3..0: afterwards lane l holds tap (l & 15) summed over the 16 lanes that share
    // its bits 5..4; then two plain steps over bits 4, 5
#define S2P_BFLY(B)                                                                         \
    {                                                                                        \
      const bool up = (lane >> (B)) & 1;                                                     \
      _Pragma("unroll") for (int i = 0; i < (1 << (B)); ++i) {                               \
        const float send = up ? d[i] : d[i + (1 << (B))];                                    \
        const float keep = up ? d[i + (1 << (B))] : d[i];                                    \
        d[i] = keep + __shfl_xor(send, 1 << (B), 64);                                        \
      }                                                                                      \
    }
    S2P_BFLY(3) S2P_BFLY(2) S2P_BFLY(1) S2P_BFLY(0)
#undef S2P_BFLY
    float v = d[0];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) {
      const int ox = ix + a.pad - kxl;
      if (ox >= 0 && ox < a.Wo) patch[iy][kyl][ox] += v;
    }
  }
  __syncthreads();
  for (int o = tid; o < a.Ho * a.Wo; o += blockDim.x) {
    const int oy = o / a.Wo, ox = o - oy * a.Wo;
    float s = a.bias ? a.bias[0] : 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int r = oy + ky - a.pad;
      if (r >= 0 && r < a.H) s += patch[r][ky][ox];
    }
    Chunk<__bf16> c; c.raw = (u32x4){0u, 0u, 0u, 0u};
    c.set(0, act_fwd(s, a.act, a.slope));
    *(u32x4*)(a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.y_pitch) = c.raw;
  }
}

// The same forward on the matrix cores (round 4).  P[pixel][tap] = sum_ci x[pixel][ci] * w[tap][ci] is a 169 x 16 x 512 GEMM per
// image: a wave owns 16 consecutive input pixels, its A fragments (16 pixels x 32 channels) are plain 16-byte global loads in the
// MFMA's own operand layout (lane (q, l15): pixel l15, channels 8 q .. 8 q + 7), the B fragments (32 channels x 16 taps) come
// straight from the 16-KB weight array -- all 2 x 16 loads of a wave are in flight before the first of its 16
// v_mfma_f32_16x16x32_bf16.  The per-tap products go to an LDS table [pixel][tap]; after one barrier each OUTPUT pixel adds its
// <= 16 entries (oy = iy + pad - ky, ox = ix + pad - kx) in tap order: reproducible, no shuffles.  18 -> 6 us at 64 x 13 x 13.
__global__ __launch_bounds__(1024) void head_fwd_mfma_kernel(const HeadFwdArgs a) {
  constexpr int KS = 4, T = 16, NS = 16;                    // 4x4 taps, 16 K steps of 32 channels (Cin = 512)
  __shared__ float P[256][T + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, l15 = lane & 15;
  const int n = blockIdx.x, HW = a.H * a.W;
  int p = tile * 16 + l15;
  const bool pok = p < HW;
  if (!pok) p = HW - 1;
  const __bf16* xp = a.x + ((size_t)n * HW + p) * a.x_pitch + 8 * q;
  const __bf16* wp = a.w + (size_t)l15 * 512 + 8 * q;
  bf16x8 xa[NS], wb[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) xa[s] = *(const bf16x8*)(xp + 32 * s);
#pragma unroll
  for (int s = 0; s < NS; ++s) wb[s] = *(const bf16x8*)(wp + 32 * s);
  typedef __attribute__((ext_vector_type(4))) float f32x4h;
  f32x4h acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[s], wb[s], acc, 0, 0, 0);
  // D[row = 4 q + e][col = l15]: pixel tile * 16 + 4 q + e, tap l15
#pragma unroll
  for (int e = 0; e < 4; ++e) P[tile * 16 + 4 * q + e][l15] = acc[e];
  __syncthreads();
  for (int o = tid; o < a.Ho * a.Wo; o += blockDim.x) {
    const int oy = o / a.Wo, ox = o - oy * a.Wo;
    float sum = a.bias ? a.bias[0] : 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int iy = oy - a.pad + ky, ix = ox - a.pad + kx;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) sum += P[iy * a.W + ix][ky * KS + kx];
      }
    Chunk<__bf16> c; c.raw = (u32x4){0u, 0u, 0u, 0u};
    c.set(0, act_fwd(sum, a.act, a.slope));
    *(u32x4*)(a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.y_pitch) = c.raw;
  }
}

bool s2p_head_fwd_applicable(const s2p_conv_desc* d) {
  return d->dtype == S2P_BF16 && d->groups == 1 && !d->transposed && !d->reflect && d->stride == 1 && d->Cout == 1 &&
         d->KH == 4 && d->KW == 4 && d->Cin == 512 && d->x_pitch % 8 == 0 && d->y_pitch == 8 && d->H <= 16 && d->Wo <= 24 &&
         d->H >= 1 && d->W >= 1;
}

int s2p_head_fwd(const s2p_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int act, float slope,
                 hipStream_t st) {
  HeadFwdArgs a{};
  a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.bias = bias; a.y = (__bf16*)y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.x_pitch = d->x_pitch; a.Ho = d->Ho; a.Wo = d->Wo; a.y_pitch = d->y_pitch;
  a.pad = d->pad; a.act = act; a.slope = slope;
  if (!S2P_DIAG_SWITCH(13) && d->H * d->W <= 256) {
    hipLaunchKernelGGL(head_fwd_mfma_kernel, dim3(d->N), dim3(64 * cdiv(d->H * d->W, 16)), 0, st, a);
    S2P_CHECK_LAUNCH("head_fwd_mfma_kernel");
    return 0;
  }
  hipLaunchKernelGGL(head_fwd_kernel, dim3(d->N), dim3(64 * d->H), 0, st, a);
  S2P_CHECK_LAUNCH("head_fwd_kernel");
  return 0;
}
