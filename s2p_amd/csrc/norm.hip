// InstanceNorm + MAT/SPADE modulation for NHWC tensors on gfx950.  HBM-bound: every tensor is touched with
// 16-byte chunks along the contiguous channel axis; per-(n,c) reductions are done per thread over a pixel
// stripe, then across stripes through LDS (fixed order), then across the workgroups of a plane by the CONSUMER,
// again in a fixed order.  No atomics anywhere: results are bitwise reproducible run to run.
//
//   stats  : opaque fp32 buffer: a 4-word header {S, rows per split, 0, 0} followed by [N][C][S][2] per-split partial
//            moments {mean_b, M2_b}.  A split accumulates sum(x-K), sum((x-K)^2) about a pivot K = its first pixel
//            (no cancellation for |mean| >> std), converts to (mean_b, M2_b); consumers merge the S partials with
//            Chan's formula.  The header makes the buffer self-describing: a batch prefix (the first n images) is a
//            valid stats buffer for that prefix whatever split count the consumer's own launch would pick.
//   sums   : opaque fp32 buffer [N][C][S][4] of per-split backward sums, summed by the consumer in split order.
//   y  = act( xhat * (1 + g_img + g_st) + (b_img + b_st) ),   xhat = (x - mean) * rstd
#include "s2p_common.h"
#include <type_traits>

struct NormArgs {
  const void* x; const void* da; const void* gb; const float* gbst; const float* stats; float* sums;
  void* y; void* dgb; float* dgbst;
  const void* res; int res_pitch;   // fused backward only: dx += res (the ResBlk skip gradient), same dtype as dx
  int N, HW, C, x_pitch, da_pitch, gb_pitch, gbst_pitch, y_pitch, dgb_pitch, dgbst_pitch;
  int act; float slope, eps;
  int psplit, rows_per_split;       // geometry of THIS launch
  int q_split;                      // geometry of the backward-sum partials (bwd apply merges q_split entries)
};
constexpr int STATS_HDR = 4;        // header words in front of the partial moments

// split geometry of the reductions: aim at >= 1024 workgroups, at least 128 pixels per split
static inline int stats_splits(int N, int HW, int C) {
  int slabs = cdiv(C, 64);
  int ps = cdiv(1024, slabs * N);
  int maxps = cdiv(HW, 128); if (ps > maxps) ps = maxps; if (ps < 1) ps = 1;
  int rows = cdiv(HW, ps);
  return cdiv(HW, rows);
}

// merge the per-split partial moments of channel c of image n (fixed order: bitwise reproducible)
__device__ __forceinline__ void mean_rstd(const NormArgs& a, int n, int c, float& mean, float& rstd) {
  const int S = ((const int*)a.stats)[0], rows = ((const int*)a.stats)[1];
  const float* p = a.stats + STATS_HDR + ((size_t)n * a.C + c) * S * 2;
  const float inv = 1.f / (float)a.HW;
  const float m0 = p[0];                               // merge about the first split's mean: the differences are O(std)
  float m = 0.f;
  for (int b = 1; b < S; ++b) {
    int nb = a.HW - b * rows; if (nb > rows) nb = rows;
    m += (float)nb * (p[2 * b] - m0);
  }
  m = m0 + m * inv;
  float M2 = 0.f;
  for (int b = 0; b < S; ++b) {
    int nb = a.HW - b * rows; if (nb > rows) nb = rows;
    const float d = p[2 * b] - m;
    M2 += p[2 * b + 1] + (float)nb * d * d;
  }
  mean = m;
  rstd = 1.f / sqrtf(M2 * inv + a.eps);
}

// the modulated value, written ONCE so that forward, backward-reduce and backward-apply round identically (the
// backward re-derives the activation mask from it)
__device__ __forceinline__ float mat_value(float x, float mean, float rstd, float gg, float bb, float& xh) {
  xh = (x - mean) * rstd;
  return __builtin_fmaf(xh, gg, bb);
}

// ------------------------------------------------------------------------------------------------
// reductions: block = (channel slab of 64, image n, pixel split).  MODE 0: moments. MODE 1: backward sums.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void in_reduce_kernel(const NormArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int CS = 64;                 // channels per block
  constexpr int NCH = CS / CE;           // chunk columns: 8 (bf16) / 16 (fp32)
  constexpr int PR = 256 / NCH;          // pixel rows in flight: 32 / 16
  constexpr int NQ = MODE == 0 ? 2 : 4;
  __shared__ float red[PR][CS + 1];
  __shared__ float cst[4][CS];
  const int tid = threadIdx.x, cc = tid % NCH, pr = tid / NCH;
  const int n = blockIdx.y, c0 = blockIdx.x * CS + cc * CE;
  const bool cok = c0 < a.C;
  const int p_begin = blockIdx.z * a.rows_per_split;
  int p_end = p_begin + a.rows_per_split;
  if (p_end > a.HW) p_end = a.HW;

  float q[NQ][CE];
#pragma unroll
  for (int k = 0; k < NQ; ++k)
#pragma unroll
    for (int e = 0; e < CE; ++e) q[k][e] = 0.f;

  float mean[CE], rstd[CE], gs[CE], bs[CE];
  // activation selectors resolved once per thread (a per-element switch on a kernel argument is scalar-branch bound)
  const float gneg = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  const T* xb = (const T*)a.x + (size_t)n * a.HW * a.x_pitch + c0;
  if (MODE == 1) {
    // per-(n, channel) constants: computed by 64 threads once per workgroup and shared through LDS (a thread walks
    // only a few pixels, so 4*CE global loads + CE rsqrt per thread would cost as much as its payload)
    if (tid < CS) {
      const int c = blockIdx.x * CS + tid;
      float m = 0.f, r = 0.f, g1 = 1.f, b1 = 0.f;
      if (c < a.C) {
        mean_rstd(a, n, c, m, r);
        if (a.gbst) { g1 = 1.f + a.gbst[(size_t)n * a.gbst_pitch + c]; b1 = a.gbst[(size_t)n * a.gbst_pitch + a.C + c]; }
      }
      cst[0][tid] = m; cst[1][tid] = r; cst[2][tid] = g1; cst[3][tid] = b1;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      mean[e] = cst[0][cc * CE + e]; rstd[e] = cst[1][cc * CE + e]; gs[e] = cst[2][cc * CE + e]; bs[e] = cst[3][cc * CE + e];
    }
  } else {
    // pivot = the split's first pixel (every stripe of the workgroup reads the same 16 bytes)
    Chunk<T> kv; kv.raw = (u32x4){0u, 0u, 0u, 0u};
    if (cok) kv.raw = *(const u32x4*)(xb + (size_t)p_begin * a.x_pitch);
#pragma unroll
    for (int e = 0; e < CE; ++e) mean[e] = kv.get(e);
    if (pr == 0) {
#pragma unroll
      for (int e = 0; e < CE; ++e) cst[0][cc * CE + e] = mean[e];
    }
  }
  if (cok) {
    for (int p = p_begin + pr; p < p_end; p += PR) {
      Chunk<T> xv; xv.raw = *(const u32x4*)(xb + (size_t)p * a.x_pitch);
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e < CE; ++e) { float v = xv.get(e) - mean[e]; q[0][e] += v; q[1][e] = __builtin_fmaf(v, v, q[1][e]); }
      } else {
        Chunk<T> dv; dv.raw = *(const u32x4*)((const T*)a.da + ((size_t)n * a.HW + p) * a.da_pitch + c0);
        Chunk<T> gv, bv;
        if (a.gb) {
          const T* gp = (const T*)a.gb + ((size_t)n * a.HW + p) * a.gb_pitch + c0;
          gv.raw = *(const u32x4*)gp; bv.raw = *(const u32x4*)(gp + a.C);
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          float gg = gs[e] + (a.gb ? gv.get(e) : 0.f);
          float bb = bs[e] + (a.gb ? bv.get(e) : 0.f);
          float xh;
          float yv = mat_value(xv.get(e), mean[e], rstd[e], gg, bb, xh);
          float dy = dv.get(e) * (yv > 0.f ? 1.f : gneg);
          float dxh = dy * gg;
          q[0][e] += dxh; q[1][e] += dxh * xh; q[2][e] += dy * xh; q[3][e] += dy;
        }
      }
    }
  }
  // cross-stripe reduction through LDS, one quantity at a time, in stripe order
  float tot[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < CE; ++e) red[pr][cc * CE + e] = q[k][e];
    __syncthreads();
    float s = 0.f;
    if (tid < CS) {
#pragma unroll 4
      for (int i = 0; i < PR; ++i) s += red[i][tid];
    }
    tot[k] = s;
  }
  if (tid < CS) {
    const int c = blockIdx.x * CS + tid;
    if (c < a.C) {
      if constexpr (MODE == 0) {
        const float nb = (float)(p_end - p_begin);
        const float d = tot[0] / nb;                       // mean_b - pivot
        float M2 = tot[1] - tot[0] * d;
        float* o = (float*)a.stats + STATS_HDR + (((size_t)n * a.C + c) * a.psplit + blockIdx.z) * 2;
        o[0] = cst[0][tid] + d; o[1] = M2 > 0.f ? M2 : 0.f;
        if (c == 0 && n == 0 && blockIdx.z == 0) *(i32x4*)a.stats = (i32x4){a.psplit, a.rows_per_split, 0, 0};
      } else {
        float* o = a.sums + (((size_t)n * a.C + c) * a.psplit + blockIdx.z) * 4;
        *(f32x4*)o = (f32x4){tot[0], tot[1], tot[2], tot[3]};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise passes.  MODE 0: forward apply.  MODE 1: backward apply (dx, dgamma_img, dbeta_img).
// Same thread geometry as the reductions: a thread owns one 16-byte channel chunk of one image and walks a pixel
// stripe, so every per-(n,c) constant (mean, rstd, state gamma/beta, backward sums) is computed ONCE per workgroup and
// the loop body is pure streaming: 16-B loads, a few FMAs per element, 16-B stores.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void in_apply_kernel(const NormArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int CS = 64, NCH = CS / CE, PR = 256 / NCH;
  const int tid = threadIdx.x, cc = tid % NCH, pr = tid / NCH;
  const int n = blockIdx.y, c0 = blockIdx.x * CS + cc * CE;
  const int p_begin = blockIdx.z * a.rows_per_split;
  int p_end = p_begin + a.rows_per_split;
  if (p_end > a.HW) p_end = a.HW;
  const float invHW = 1.f / (float)a.HW;
  // per-(n, channel) constants once per workgroup through LDS (see in_reduce_kernel)
  __shared__ float cst[6][CS];
  if (tid < CS) {
    const int c = blockIdx.x * CS + tid;
    float m = 0.f, r = 0.f, g1 = 1.f, b1 = 0.f, q1 = 0.f, q2 = 0.f;
    if (c < a.C) {
      mean_rstd(a, n, c, m, r);
      if (a.gbst) { g1 = 1.f + a.gbst[(size_t)n * a.gbst_pitch + c]; b1 = a.gbst[(size_t)n * a.gbst_pitch + a.C + c]; }
      if (MODE == 1) {
        const float* sm = a.sums + ((size_t)n * a.C + c) * a.q_split * 4;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        for (int b = 0; b < a.q_split; ++b) { t0 += sm[4 * b]; t1 += sm[4 * b + 1]; t2 += sm[4 * b + 2]; t3 += sm[4 * b + 3]; }
        q1 = t0 * invHW; q2 = t1 * invHW;
        // gradient of the per-sample state affine (gamma_st | beta_st): one workgroup per (n, slab) stores it
        if (a.dgbst && blockIdx.z == 0) {
          a.dgbst[(size_t)n * a.dgbst_pitch + c] = t2;
          a.dgbst[(size_t)n * a.dgbst_pitch + a.C + c] = t3;
        }
      }
    }
    cst[0][tid] = m; cst[1][tid] = r; cst[2][tid] = g1; cst[3][tid] = b1; cst[4][tid] = q1; cst[5][tid] = q2;
  }
  __syncthreads();
  if (c0 >= a.C) return;
  // activation selectors resolved once per thread: relu / lrelu / none are  v > 0 ? v : v * ns
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  const bool act_generic = a.act == S2P_ACT_TANH || a.act == S2P_ACT_SWISH;
  const bool g_tanh = a.act == S2P_ACT_TANH;
  float mean[CE], rstd[CE], gs[CE], bs[CE], s1[CE], s2[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    const int k = cc * CE + e;
    mean[e] = cst[0][k]; rstd[e] = cst[1][k]; gs[e] = cst[2][k]; bs[e] = cst[3][k];
    s1[e] = cst[4][k]; s2[e] = cst[5][k];
  }
  const size_t img = (size_t)n * a.HW;
  const T* xb = (const T*)a.x + img * a.x_pitch + c0;
  const T* gbb = a.gb ? (const T*)a.gb + img * a.gb_pitch + c0 : nullptr;
  const T* dab = MODE == 1 ? (const T*)a.da + img * a.da_pitch + c0 : nullptr;
  T* yb = (T*)a.y + img * a.y_pitch + c0;
  T* dgb = (MODE == 1 && a.dgb) ? (T*)a.dgb + img * a.dgb_pitch + c0 : nullptr;
  for (int p = p_begin + pr; p < p_end; p += PR) {
    Chunk<T> xv; xv.raw = *(const u32x4*)(xb + (size_t)p * a.x_pitch);
    Chunk<T> gv, bv, dv;
    if (gbb) { gv.raw = *(const u32x4*)(gbb + (size_t)p * a.gb_pitch); bv.raw = *(const u32x4*)(gbb + (size_t)p * a.gb_pitch + a.C); }
    if (MODE == 1) dv.raw = *(const u32x4*)(dab + (size_t)p * a.da_pitch);
    Chunk<T> o0, o1, o2;
    float v0[CE], v1[CE], v2[CE];                     // packed pairwise at the end (Chunk::pack: one conversion per two elements)
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float gg = gs[e] + (gbb ? gv.get(e) : 0.f);
      float bb = bs[e] + (gbb ? bv.get(e) : 0.f);
      float xh;
      float yv = mat_value(xv.get(e), mean[e], rstd[e], gg, bb, xh);
      if (MODE == 0) {
        v0[e] = lrelu_ns(yv, ns);
      } else {
        const float dvv = dv.get(e);
        float dy = yv > 0.f ? dvv : dvv * ns;
        if (g_tanh) dy = dvv * (1.f - yv * yv);
        float dxh = dy * gg;
        v0[e] = rstd[e] * (dxh - s1[e] - xh * s2[e]);
        v1[e] = dy * xh;
        v2[e] = dy;
      }
    }
    o0.pack(v0);
    if (MODE == 1) { o1.pack(v1); o2.pack(v2); }
    if (MODE == 0 && act_generic) {                   // tanh / swish: rare, one uniform branch per chunk
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        float gg = gs[e] + (gbb ? gv.get(e) : 0.f);
        float bb = bs[e] + (gbb ? bv.get(e) : 0.f);
        float xh;
        o0.set(e, act_fwd(mat_value(xv.get(e), mean[e], rstd[e], gg, bb, xh), a.act, a.slope));
      }
    }
    *(u32x4*)(yb + (size_t)p * a.y_pitch) = o0.raw;
    if (MODE == 1 && dgb) {
      *(u32x4*)(dgb + (size_t)p * a.dgb_pitch) = o1.raw;
      *(u32x4*)(dgb + (size_t)p * a.dgb_pitch + a.C) = o2.raw;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused forward for small planes (HW * 16 B * MAXP fits a thread's registers: the 21x21 MAT norms of the ResBlks and
// the small PatchGAN maps): ONE workgroup of 1024 threads owns a whole (image, 64-channel slab) plane, loads every
// chunk of x (and gamma / beta) it needs up front (196 KB in flight per CU), computes the exact two-pass statistics
// (mean, then centred second moment) with wave shuffles + LDS, and applies the modulation from registers: x is read
// once instead of twice and the statistics pass is not a separate launch.  It also writes the statistics buffer in
// the usual self-describing format (one split), so the backward kernels consume it unchanged.
// TH = 256 is the form for the SMALL planes (PatchGAN 7x7 .. 13x13 maps, plain InstanceNorm: GB = false): 8 chunks per thread,
// planes of <= 256 (bf16) / 128 (fp32) pixels.  A 1024-thread workgroup leaves most of its lanes without a pixel there and only
// two workgroups fit a CU, so load, reduce and store phases hardly overlap; eight 256-thread workgroups per CU do.
// MP > 0 (round 4): the LARGE-plane form for the plain InstanceNorms of the encoder / decoder (42x42 x 128, 84x84 x 64 channels): MP
// chunks per thread, the plane of an (image, CS-channel slab) still in the registers of ONE workgroup -- 512 threads (a 256-VGPR
// budget: at 1024 threads / 128 VGPRs the compiler spilled 50 of them), 112 VGPRs of payload at MP = 28: CS = 64 holds 1792
// pixels, CS = 16 (two 16-byte chunks per pixel; the four slabs of an image run side by side on one XCD and share its 128-byte
// lines in L2) 7168.  One read of x instead of the reduce + apply pair's two: 40.8 -> 36.4 us at 84x84 x 64 alone (the second
// read was an infinity-cache hit), step -0.06 ms.  Half-width slabs at 14 chunks per thread (two workgroups per CU) were measured:
// 42x42 16.9 vs 21.3 us, 84x84 51 vs 36 us, step +0.06 ms -- not kept.
template <typename T, int CS, int TH = 1024, bool GB = true, int MP = 0>   // CS = channels per workgroup: 64 (full 128-byte bf16 lines), 32 or 16
__global__ __launch_bounds__(TH, (MP > 14 || TH == 1024) ? 1 : 4) void in_fused_fwd_kernel(const NormArgs a) {     // <= 128 VGPRs but for the 28-chunk form: MP 14 two workgroups per CU, TH 256 four
  constexpr int CE = DT<T>::CE;
  constexpr int NCH = CS / CE, PR = TH / NCH, MAXP = MP ? MP : (TH == 1024 ? (DT<T>::CE == 8 ? 512 : 256) / PR : 8);   // TH 1024: planes of <= 512 (bf16) / 256 (fp32) pixels; TH 512 the same with 8 chunks per thread
  constexpr int RPW = 64 / NCH, NW = TH / 64, MG = GB ? MAXP : 1;
  __shared__ float red[NW][CS];
  __shared__ float cst[4][CS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = tid % NCH, pr = tid / NCH;
  // grid = (images, slabs), images fastest: the workgroups of neighbouring slabs of one image are N apart in dispatch
  // order, i.e. on the same XCD when N % 8 == 0 -- with CS = 32 the two halves of a 128-byte line meet in one L2
  const int n = blockIdx.x, slab = blockIdx.y, c0 = slab * CS + cc * CE;
  const bool cok = c0 < a.C;
  const size_t img = (size_t)n * a.HW;
  const T* xb = (const T*)a.x + img * a.x_pitch + c0;
  const T* gbb = (GB && a.gb) ? (const T*)a.gb + img * a.gb_pitch + c0 : nullptr;
  Chunk<T> xv[MAXP], gv[MG], bv[MG];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int p = pr + k * PR;
    xv[k].raw = (u32x4){0u, 0u, 0u, 0u};
    if (GB) { gv[k % MG].raw = xv[k].raw; bv[k % MG].raw = xv[k].raw; }
    if (cok && p < a.HW) {
      xv[k].raw = *(const u32x4*)(xb + (size_t)p * a.x_pitch);
      if (GB && gbb) { gv[k % MG].raw = *(const u32x4*)(gbb + (size_t)p * a.gb_pitch); bv[k % MG].raw = *(const u32x4*)(gbb + (size_t)p * a.gb_pitch + a.C); }
    }
  }
  // sum over the pixel rows of a wave (lanes that share cc differ in the lane bits above log2(NCH)), then over waves
  auto plane_sum = [&](float (&v)[CE], int slot) {
#pragma unroll
    for (int e = 0; e < CE; ++e) {
#pragma unroll
      for (int o = NCH; o < 64; o <<= 1) v[e] += __shfl_xor(v[e], o, 64);
    }
    __syncthreads();                                                   // red[] free again
    if (lane < NCH) {
#pragma unroll
      for (int e = 0; e < CE; ++e) red[wave][lane * CE + e] = v[e];
    }
    __syncthreads();
    if (tid < CS) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w][tid];
      cst[slot][tid] = t;
    }
    __syncthreads();
  };
  (void)RPW;
  const float inv = 1.f / (float)a.HW;
  float acc[CE], mean[CE], rstd[CE];
  // KEEP (the plane fits the registers unpacked: every form but the large-plane one): the chunks are unpacked ONCE and the second pass
  // leaves the centred values for the third; the large-plane form (MP > 0) unpacks in every pass (its payload must stay packed)
  constexpr bool KEEP = MP == 0;
  float xf[KEEP ? MAXP : 1][CE];
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) xv[k].unpack(xf[k]);
  }
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    acc[e] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) acc[e] += KEEP ? xf[KEEP ? k : 0][e] : xv[k].get(e);      // missing pixels were loaded as zeros
  }
  plane_sum(acc, 0);
  // large-plane form: the payload must stay PACKED between the passes (the compiler would otherwise keep all MAXP * CE unpacked
  // floats of a pass alive for the next one: 224 VGPRs, 40 of them spilled)
  auto keep_packed = [&]() {
    if constexpr (MP > 0) {
#pragma unroll
      for (int k = 0; k < MAXP; ++k) asm volatile("" : "+v"(xv[k].raw));
    }
  };
  keep_packed();
#pragma unroll
  for (int e = 0; e < CE; ++e) mean[e] = cst[0][cc * CE + e] * inv;
#pragma unroll
  for (int e = 0; e < CE; ++e) acc[e] = 0.f;
  // (workgroup-uniform) every thread owns a pixel in each of its first MAXP - 1 slots: only the last slot needs the bounds select
  const bool full_rows = a.HW >= (MAXP - 1) * PR + PR;
  auto second_moment = [&](auto fullc) {
    constexpr bool FULL = decltype(fullc)::value;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const bool in = pr + k * PR < a.HW;
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const float d = (KEEP ? xf[KEEP ? k : 0][e] : xv[k].get(e)) - mean[e];
        if constexpr (KEEP) xf[KEEP ? k : 0][e] = d;
        if (FULL && k < MAXP - 1) acc[e] = __builtin_fmaf(d, d, acc[e]);
        else acc[e] += in ? d * d : 0.f;
      }
    }
  };
  if constexpr (MP > 14) {                                              // (two copies of the pass only where it is 224 elements long)
    if (full_rows) second_moment(std::integral_constant<bool, true>{}); else second_moment(std::integral_constant<bool, false>{});
  } else second_moment(std::integral_constant<bool, false>{});
  plane_sum(acc, 1);
  keep_packed();
  if (tid < CS) {
    const int c = slab * CS + tid;
    if (c < a.C) {
      float* o = (float*)a.stats + STATS_HDR + ((size_t)n * a.C + c) * 2;
      o[0] = cst[0][tid] * inv; o[1] = cst[1][tid];
      if (c == 0 && n == 0) *(i32x4*)a.stats = (i32x4){1, a.HW, 0, 0};
      cst[2][tid] = a.gbst ? 1.f + a.gbst[(size_t)n * a.gbst_pitch + c] : 1.f;
      cst[3][tid] = a.gbst ? a.gbst[(size_t)n * a.gbst_pitch + a.C + c] : 0.f;
    }
  }
  __syncthreads();
  if (!cok) return;
  float gs[CE], bs[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    rstd[e] = 1.f / sqrtf(cst[1][cc * CE + e] * inv + a.eps);
    gs[e] = cst[2][cc * CE + e]; bs[e] = cst[3][cc * CE + e];
  }
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);   // none / relu / lrelu only (host)
  T* yb = (T*)a.y + img * a.y_pitch + c0;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int p = pr + k * PR;
    if (p >= a.HW) break;
    Chunk<T> o0;
    float ov[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      float gg = gs[e] + ((GB && gbb) ? gv[k % MG].get(e) : 0.f);
      float bb = bs[e] + ((GB && gbb) ? bv[k % MG].get(e) : 0.f);
      // mat_value's arithmetic, bit for bit: ((x - mean) * rstd) * gg + bb
      const float xh = (KEEP ? xf[KEEP ? k : 0][e] : xv[k].get(e) - mean[e]) * rstd[e];
      ov[e] = lrelu_ns(__builtin_fmaf(xh, gg, bb), ns);
    }
    o0.pack(ov);
    *(u32x4*)(yb + (size_t)p * a.y_pitch) = o0.raw;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused backward for small planes, same geometry as in_fused_fwd_kernel: one workgroup of 1024 threads owns an (image,
// 64-channel slab) plane, loads x, dL/dy, gamma, beta once (64 VGPRs of 16-byte chunks per thread), forms the four plane
// sums (wave shuffles + LDS, fixed order) and then writes dx, d(gamma_img | beta_img) and the state-affine gradient from
// the registers: one launch and one read of every tensor instead of the reduce + apply pair (each tensor read twice).
// Per-channel constants stay in LDS and are re-read per use (keeps the kernel under the 128-VGPR budget of 16 waves / CU).
template <typename T, int CS, int TH = 1024, bool GB = true>      // TH = 256, GB = false: the small-plane form (see in_fused_fwd_kernel)
__global__ __launch_bounds__(TH) void in_fused_bwd_kernel(const NormArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int NCH = CS / CE, PR = TH / NCH, MAXP = TH == 1024 ? (DT<T>::CE == 8 ? 512 : 256) / PR : 8;
  constexpr int NW = TH / 64, MG = GB ? MAXP : 1;
  typedef typename std::conditional<(MAXP * CE > 32), unsigned long long, unsigned>::type mask_t;
  __shared__ float red[4][NW][CS];
  __shared__ __attribute__((aligned(16))) float cst[6][CS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = tid % NCH, pr = tid / NCH;
  const int n = blockIdx.x, slab = blockIdx.y, c0 = slab * CS + cc * CE;       // images fastest: see in_fused_fwd_kernel
  const bool cok = c0 < a.C;
  // workgroup-uniform bases (SGPRs) + 32-bit per-lane element offsets: 64-bit per-lane addresses for 4 pixels x 7 tensors
  // would not fit the register budget (the plane of one image is far below 2^31 elements)
  const size_t img = (size_t)n * a.HW;
  const int cb0 = slab * CS;
  const T* xb = (const T*)a.x + img * a.x_pitch + cb0;
  const T* dab = (const T*)a.da + img * a.da_pitch + cb0;
  const T* gbb = (GB && a.gb) ? (const T*)a.gb + img * a.gb_pitch + cb0 : nullptr;
  const int lc = cc * CE;
  Chunk<T> xv[MAXP], dv[MAXP], gv[MG], bv[MG];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int p = pr + k * PR;
    xv[k].raw = (u32x4){0u, 0u, 0u, 0u}; dv[k].raw = xv[k].raw;
    if (GB) { gv[k % MG].raw = xv[k].raw; bv[k % MG].raw = xv[k].raw; }
    if (cok && p < a.HW) {
      xv[k].raw = *(const u32x4*)(xb + (p * a.x_pitch + lc));
      dv[k].raw = *(const u32x4*)(dab + (p * a.da_pitch + lc));
      if (GB && gbb) { gv[k % MG].raw = *(const u32x4*)(gbb + (p * a.gb_pitch + lc)); bv[k % MG].raw = *(const u32x4*)(gbb + (p * a.gb_pitch + lc + a.C)); }
    }
  }
  if (tid < CS) {
    const int c = slab * CS + tid;
    float m = 0.f, r = 0.f, g1 = 1.f, b1 = 0.f;
    if (c < a.C) {
      mean_rstd(a, n, c, m, r);
      if (a.gbst) { g1 = 1.f + a.gbst[(size_t)n * a.gbst_pitch + c]; b1 = a.gbst[(size_t)n * a.gbst_pitch + a.C + c]; }
    }
    cst[0][tid] = m; cst[1][tid] = r; cst[2][tid] = g1; cst[3][tid] = b1;
  }
  __syncthreads();
  const float gneg = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  mask_t posmask = 0;
  // ---- pass 1: plane sums.  Pixels beyond HW were loaded as zeros: their dL/dy is 0, so they add nothing.
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    const int ch = cc * CE + e;
    const float m = cst[0][ch], r = cst[1][ch], g1 = cst[2][ch], b1 = cst[3][ch];
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const float gg = g1 + ((GB && gbb) ? gv[k % MG].get(e) : 0.f), bb = b1 + ((GB && gbb) ? bv[k % MG].get(e) : 0.f);
      float xh;
      const float yv = mat_value(xv[k].get(e), m, r, gg, bb, xh);
      const bool pos = yv > 0.f;
      posmask |= pos ? ((mask_t)1 << (k * CE + e)) : (mask_t)0;           // pass 2 takes the branch from here: beta is dead after pass 1
      const float dy = dv[k].get(e) * (pos ? 1.f : gneg);
      const float dxh = dy * gg;
      q0 += dxh; q1 += dxh * xh; q2 += dy * xh; q3 += dy;
    }
#pragma unroll
    for (int o = NCH; o < 64; o <<= 1) {
      q0 += __shfl_xor(q0, o, 64); q1 += __shfl_xor(q1, o, 64); q2 += __shfl_xor(q2, o, 64); q3 += __shfl_xor(q3, o, 64);
    }
    if (lane < NCH) { red[0][wave][ch] = q0; red[1][wave][ch] = q1; red[2][wave][ch] = q2; red[3][wave][ch] = q3; }
    __builtin_amdgcn_sched_barrier(0);        // one channel at a time: interleaving the unrolled channels spills (128-VGPR budget)
  }
  __syncthreads();
  if (tid < CS) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t0 += red[0][w][tid]; t1 += red[1][w][tid]; t2 += red[2][w][tid]; t3 += red[3][w][tid]; }
    const float inv = 1.f / (float)a.HW;
    cst[4][tid] = t0 * inv; cst[5][tid] = t1 * inv;
    const int c = slab * CS + tid;
    if (a.dgbst && c < a.C) {
      a.dgbst[(size_t)n * a.dgbst_pitch + c] = t2;
      a.dgbst[(size_t)n * a.dgbst_pitch + a.C + c] = t3;
    }
  }
  __syncthreads();
  if (!cok) return;
  // ---- pass 2: outputs, pixel by pixel, four channels at a time (constants re-read from LDS as float4)
  T* yb = (T*)a.y + img * a.y_pitch + cb0;
  T* dgb = a.dgb ? (T*)a.dgb + img * a.dgb_pitch + cb0 : nullptr;
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int p = pr + k * PR;
    if (p >= a.HW) break;
    Chunk<T> o0, o1, o2;
    float v0[CE], v1[CE], v2[CE];                        // packed pairwise below (Chunk::pack)
#pragma unroll
    for (int h4 = 0; h4 < CE / 4; ++h4) {
      const int ch = cc * CE + 4 * h4;
      const f32x4 m4 = *(const f32x4*)&cst[0][ch], r4 = *(const f32x4*)&cst[1][ch], g4 = *(const f32x4*)&cst[2][ch],
                  s14 = *(const f32x4*)&cst[4][ch], s24 = *(const f32x4*)&cst[5][ch];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = 4 * h4 + j;
        const float gg = g4[j] + ((GB && gbb) ? gv[k % MG].get(e) : 0.f);
        const float xh = (xv[k].get(e) - m4[j]) * r4[j];
        const float dy = dv[k].get(e) * (((posmask >> (k * CE + e)) & (mask_t)1) ? 1.f : gneg);
        const float dxh = dy * gg;
        v0[e] = r4[j] * (dxh - s14[j] - xh * s24[j]);
        v1[e] = dy * xh;
        v2[e] = dy;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    o0.pack(v0); o1.pack(v1); o2.pack(v2);
    if (a.res) {                                         // skip-connection gradient folded into the store (added to the ROUNDED dx, as before)
      Chunk<T> rv; rv.raw = *(const u32x4*)((const T*)a.res + img * a.res_pitch + cb0 + (p * a.res_pitch + lc));
#pragma unroll
      for (int e = 0; e < CE; ++e) v0[e] = o0.get(e) + rv.get(e);
      o0.pack(v0);
    }
    *(u32x4*)(yb + (p * a.y_pitch + lc)) = o0.raw;
    if (dgb) {
      *(u32x4*)(dgb + (p * a.dgb_pitch + lc)) = o1.raw;
      *(u32x4*)(dgb + (p * a.dgb_pitch + lc + a.C)) = o2.raw;
    }
  }
}

// per-channel sum over pixels (bias gradient)
// part == nullptr: the pixel blocks add to db with fp32 atomics; otherwise block y stores its sums to part[y][Cs] (Cs = C rounded
// up to 64) and channel_sum_reduce_kernel adds them in block order: bitwise reproducible
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* dy, long long pixels, int C, int pitch,
                                                          float* db, int rows_per_block, float* part) {
  constexpr int CE = DT<T>::CE;
  constexpr int CS = 64, NCH = CS / CE;
  __shared__ float red[256][CE + 1];
  const int tid = threadIdx.x;
  // chunks of this 64-channel slab that hold real channels (a 3-channel image tensor has one): the threads are spread over
  // (pixel row, chunk) so that thin tensors keep all 256 threads loading
  int nch = (C - (int)blockIdx.x * CS + CE - 1) / CE; if (nch > NCH) nch = NCH;
  int npow = 1; while (npow < nch) npow <<= 1;           // power of two: 256 / npow pixel rows per pass
  const int cc = tid % npow, pr = tid / npow, PR = 256 / npow;
  const int c0 = blockIdx.x * CS + cc * CE;
  long long p0 = (long long)blockIdx.y * rows_per_block, p1 = p0 + rows_per_block;
  if (p1 > pixels) p1 = pixels;
  float q[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) q[e] = 0.f;
  if (cc < nch) {
    long long p = p0 + pr;
    if (c0 + CE <= C) {
      // four rows in flight per thread (one dependent load per pass left the kernel latency-bound: 2.2 TB/s on the 87-MB gradient
      // of the conditioning conv); the sums are added in row order
      for (; p + 3 * PR < p1; p += 4 * PR) {
        Chunk<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].raw = *(const u32x4*)(dy + (size_t)(p + u * PR) * pitch + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < CE; ++e) q[e] += v[u].get(e);
      }
    }
    for (; p < p1; p += PR) {
      if (c0 + CE <= C) {
        Chunk<T> v; v.raw = *(const u32x4*)(dy + (size_t)p * pitch + c0);
#pragma unroll
        for (int e = 0; e < CE; ++e) q[e] += v.get(e);
      } else {
        for (int e = 0; e < CE; ++e) if (c0 + e < C) q[e] += to_f32(dy[(size_t)p * pitch + c0 + e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < CE; ++e) red[tid][e] = q[e];
  __syncthreads();
  if (tid < CS) {
    const int ch = tid / CE, e = tid % CE;               // channel tid of the slab lives in chunk ch, element e
    float s = 0.f;
    if (ch < nch)
      for (int i = 0; i < PR; ++i) s += red[i * npow + ch][e];
    int c = blockIdx.x * CS + tid;
    if (part) part[(size_t)blockIdx.y * (gridDim.x * CS) + c] = s;
    else if (c < C) atomicAdd(db + c, s);
  }
}

// ------------------------------------------------------------------------------------------------
static int norm_check(const char* who, int dtype, int C, int p0, int p1, int p2) {
  if (dtype != S2P_F32 && dtype != S2P_BF16) S2P_FAIL(-1, "%s: bad dtype", who);
  int ce = dtype == S2P_F32 ? 4 : 8;
  if (C % ce || p0 % ce || p1 % ce || p2 % ce) S2P_FAIL(-1, "%s: C / pitches must be multiples of %d", who, ce);
  return 0;
}

template <int MODE>
static int launch_reduce(int dtype, NormArgs& a, hipStream_t st) {
  int slabs = cdiv(a.C, 64);
  a.psplit = stats_splits(a.N, a.HW, a.C); a.rows_per_split = cdiv(a.HW, a.psplit);
  a.q_split = a.psplit;
  dim3 grid(slabs, a.N, a.psplit);
  if (dtype == S2P_F32) hipLaunchKernelGGL((in_reduce_kernel<float, MODE>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((in_reduce_kernel<__bf16, MODE>), grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("in_reduce_kernel");
  return 0;
}

template <int MODE>
static int launch_apply(int dtype, NormArgs& a, hipStream_t st) {
  int slabs = cdiv(a.C, 64);
  a.q_split = stats_splits(a.N, a.HW, a.C);             // geometry of the backward-sum partials of the same call
  int ps = cdiv(1024, slabs * a.N);                     // aim at >= 1024 workgroups
  int maxps = cdiv(a.HW, 64); if (ps > maxps) ps = maxps; if (ps < 1) ps = 1;
  a.psplit = ps; a.rows_per_split = cdiv(a.HW, ps);
  dim3 grid(slabs, a.N, cdiv(a.HW, a.rows_per_split));
  if (dtype == S2P_F32) hipLaunchKernelGGL((in_apply_kernel<float, MODE>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((in_apply_kernel<__bf16, MODE>), grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("in_apply_kernel");
  return 0;
}

extern "C" int64_t s2p_in_stats_floats(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  return STATS_HDR + (int64_t)N * C * stats_splits(N, HW, C) * 2;
}

extern "C" int64_t s2p_in_bwd_sums_floats(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  return (int64_t)N * C * stats_splits(N, HW, C) * 4;
}

extern "C" int s2p_in_stats(int dtype, const void* x, int N, int HW, int C, int pitch, float eps, float* stats,
                            void* stream) {
  int rc = norm_check("s2p_in_stats", dtype, C, pitch, 0, 0); if (rc) return rc;
  if (!x || !stats || N <= 0 || HW <= 0) S2P_FAIL(-1, "s2p_in_stats: null pointer / empty problem");
  NormArgs a{}; a.x = x; a.stats = stats; a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.eps = eps;
  return launch_reduce<0>(dtype, a, (hipStream_t)stream);
}

extern "C" int s2p_in_apply_fwd(int dtype, const void* x, int N, int HW, int C, int pitch, const float* stats,
                                const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch, int act,
                                float slope, float eps, void* y, int y_pitch, void* stream) {
  int rc = norm_check("s2p_in_apply_fwd", dtype, C, pitch, gb_pitch, y_pitch); if (rc) return rc;
  S2P_CHECK_SLOPE("s2p_in_apply_fwd", act, slope);
  NormArgs a{}; a.x = x; a.stats = stats; a.gb = gb_img; a.gbst = gb_st; a.y = y;
  a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.gb_pitch = gb_pitch; a.gbst_pitch = gb_st_pitch;
  a.y_pitch = y_pitch; a.act = act; a.slope = slope; a.eps = eps;
  return launch_apply<0>(dtype, a, (hipStream_t)stream);
}

// the 256-thread form of the fused kernels: plain InstanceNorm (no gamma/beta maps) on planes of <= 256 (bf16) / 128 (fp32) pixels
// (returns the workgroup size: 256, 512 for up to twice those planes, or 0: the 1024-thread form)
static inline int small_plane(int dtype, int HW, const void* gb_img) {
  static const int off = s2p_env_set("S2P_NORM_NO_SMALL");          // A/B switches (diagnostics build only)
  static const int no512 = s2p_env_set("S2P_NORM_NO_512");
  if (off || gb_img) return 0;
  const int lim = dtype == S2P_F32 ? 128 : 256;
  return HW <= lim ? 256 : (HW <= 2 * lim && !no512 ? 512 : 0);
}

// statistics + apply in one call: one fused launch for small planes, otherwise the two-kernel path
extern "C" int s2p_in_norm_fwd(int dtype, const void* x, int N, int HW, int C, int pitch, const void* gb_img, int gb_pitch,
                               const float* gb_st, int gb_st_pitch, int act, float slope, float eps, void* y, int y_pitch,
                               float* stats, void* stream) {
  int rc = norm_check("s2p_in_norm_fwd", dtype, C, pitch, gb_pitch, y_pitch); if (rc) return rc;
  S2P_CHECK_SLOPE("s2p_in_norm_fwd", act, slope);
  if (!x || !y || !stats || N <= 0 || HW <= 0) S2P_FAIL(-1, "s2p_in_norm_fwd: null pointer / empty problem");
  const int maxhw = dtype == S2P_F32 ? 256 : 512;
  const bool simple_act = act == S2P_ACT_NONE || act == S2P_ACT_RELU || act == S2P_ACT_LRELU;   // tanh / swish: two-kernel path
  if (dtype == S2P_BF16 && HW > maxhw && simple_act && !gb_img && S2P_DIAG_SWITCH(9) != 1 && !s2p_env_set("S2P_NO_FUSED_NORM")) {
    // large planes, plain InstanceNorm: the register-resident form (see in_fused_fwd_kernel, MP)
    // planes of <= 1792 pixels: 64-channel slabs at 28 chunks per thread, or -- when those are fewer workgroups than CUs (42x42 x 128 at
    // N 64: 128) -- 32-channel slabs at 14 chunks (twice the workgroups, 56 VGPRs of payload: round 4 measured 16.9 vs 21.3 us on that
    // plane and rejected the half slabs only because the SAME switch also halved the 84x84 planes' slabs, 51 vs 36 us)
    int cs = (HW <= 28 * 64 && C % 64 == 0) ? 64 : ((HW <= 28 * 256 && C % 16 == 0) ? 16 : 0);
    if (cs == 64 && HW <= 14 * 128 && C % 32 == 0 && N * (C / 64) < 256 && !S2P_DIAG_SWITCH(11)) cs = 32;
    if (cs && N * (C / cs) >= 64) {        // (a handful of workgroups: the two-kernel path spreads over more CUs)
      NormArgs a{}; a.x = x; a.stats = stats; a.gbst = gb_st; a.y = y;
      a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.gbst_pitch = gb_st_pitch; a.y_pitch = y_pitch; a.act = act; a.slope = slope; a.eps = eps;
      const dim3 lg(N, C / cs);
      if (cs == 64) hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 64, 512, false, 28>), lg, dim3(512), 0, (hipStream_t)stream, a);
      else if (cs == 32) hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 32, 512, false, 14>), lg, dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 16, 512, false, 28>), lg, dim3(512), 0, (hipStream_t)stream, a);
      S2P_CHECK_LAUNCH("in_fused_fwd_kernel (large planes)");
      return 0;
    }
  }
  if (HW > maxhw || !simple_act || s2p_env_set("S2P_NO_FUSED_NORM")) {
    rc = s2p_in_stats(dtype, x, N, HW, C, pitch, eps, stats, stream); if (rc) return rc;
    return s2p_in_apply_fwd(dtype, x, N, HW, C, pitch, stats, gb_img, gb_pitch, gb_st, gb_st_pitch, act, slope, eps, y, y_pitch, stream);
  }
  NormArgs a{}; a.x = x; a.stats = stats; a.gb = gb_img; a.gbst = gb_st; a.y = y;
  a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.gb_pitch = gb_pitch; a.gbst_pitch = gb_st_pitch;
  a.y_pitch = y_pitch; a.act = act; a.slope = slope; a.eps = eps;
  // 32-channel slabs (twice the workgroups, two per CU) were measured and lost: 24.2 vs 20.5 us on the MAT norms -- the
  // 64-byte half-line accesses cost more than the extra overlap buys.  Kept selectable in the diagnostics build only.
  const bool half = s2p_env_set("S2P_NORM_CS32") && C % 32 == 0;
  dim3 grid(N, cdiv(C, half ? 32 : 64));
  hipStream_t st = (hipStream_t)stream;
  if (const int th = small_plane(dtype, HW, gb_img)) {
    const dim3 sg(N, cdiv(C, 64));
    if (th == 256) { if (dtype == S2P_F32) hipLaunchKernelGGL((in_fused_fwd_kernel<float, 64, 256, false>), sg, dim3(256), 0, st, a);
                     else hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 64, 256, false>), sg, dim3(256), 0, st, a); }
    else { if (dtype == S2P_F32) hipLaunchKernelGGL((in_fused_fwd_kernel<float, 64, 512, false>), sg, dim3(512), 0, st, a);
           else hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 64, 512, false>), sg, dim3(512), 0, st, a); }
    S2P_CHECK_LAUNCH("in_fused_fwd_kernel (small planes)");
    return 0;
  }
  if (dtype == S2P_F32) { if (half) hipLaunchKernelGGL((in_fused_fwd_kernel<float, 32>), grid, dim3(1024), 0, st, a);
                          else hipLaunchKernelGGL((in_fused_fwd_kernel<float, 64>), grid, dim3(1024), 0, st, a); }
  else { if (half) hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 32>), grid, dim3(1024), 0, st, a);
         else hipLaunchKernelGGL((in_fused_fwd_kernel<__bf16, 64>), grid, dim3(1024), 0, st, a); }
  S2P_CHECK_LAUNCH("in_fused_fwd_kernel");
  return 0;
}

extern "C" int s2p_in_bwd_reduce(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C,
                                 int pitch, const float* stats, const void* gb_img, int gb_pitch,
                                 const float* gb_st, int gb_st_pitch, int act, float slope, float eps,
                                 float* sums, void* stream) {
  int rc = norm_check("s2p_in_bwd_reduce", dtype, C, pitch, gb_pitch, da_pitch); if (rc) return rc;
  NormArgs a{}; a.x = x; a.da = da; a.stats = stats; a.gb = gb_img; a.gbst = gb_st; a.sums = sums;
  a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.da_pitch = da_pitch; a.gb_pitch = gb_pitch;
  a.gbst_pitch = gb_st_pitch; a.act = act; a.slope = slope; a.eps = eps;
  return launch_reduce<1>(dtype, a, (hipStream_t)stream);
}

extern "C" int s2p_in_bwd_apply(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C,
                                int pitch, const float* stats, const void* gb_img, int gb_pitch,
                                const float* gb_st, int gb_st_pitch, int act, float slope, float eps,
                                const float* sums, void* dx, int dx_pitch, void* dgb_img, int dgb_pitch,
                                float* dgb_st, int dgb_st_pitch, void* stream) {
  int rc = norm_check("s2p_in_bwd_apply", dtype, C, pitch, gb_pitch, da_pitch); if (rc) return rc;
  if (dx_pitch % (dtype == S2P_F32 ? 4 : 8) || dgb_pitch % (dtype == S2P_F32 ? 4 : 8))
    S2P_FAIL(-1, "s2p_in_bwd_apply: bad output pitch");
  if (dgb_st && dgb_st_pitch < 2 * C) S2P_FAIL(-1, "s2p_in_bwd_apply: dgb_st pitch < 2*C");
  NormArgs a{}; a.x = x; a.da = da; a.stats = stats; a.gb = gb_img; a.gbst = gb_st; a.sums = (float*)sums;
  a.y = dx; a.dgb = dgb_img; a.dgbst = dgb_st; a.dgbst_pitch = dgb_st_pitch;
  a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.da_pitch = da_pitch; a.gb_pitch = gb_pitch;
  a.gbst_pitch = gb_st_pitch; a.y_pitch = dx_pitch; a.dgb_pitch = dgb_pitch; a.act = act; a.slope = slope;
  a.eps = eps;
  return launch_apply<1>(dtype, a, (hipStream_t)stream);
}

// reduce + apply in one call: one fused launch for small planes (relu / lrelu / no activation), else the two kernels
extern "C" int s2p_in_norm_bwd(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C, int pitch,
                               const float* stats, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                               int act, float slope, float eps, float* sums, void* dx, int dx_pitch, void* dgb_img,
                               int dgb_pitch, float* dgb_st, int dgb_st_pitch, void* stream) {
  return s2p_in_norm_bwd_res(dtype, da, da_pitch, x, N, HW, C, pitch, stats, gb_img, gb_pitch, gb_st, gb_st_pitch, act, slope, eps,
                             sums, dx, dx_pitch, dgb_img, dgb_pitch, dgb_st, dgb_st_pitch, nullptr, 0, stream);
}

extern "C" int s2p_in_norm_bwd_res(int dtype, const void* da, int da_pitch, const void* x, int N, int HW, int C, int pitch,
                                   const float* stats, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                                   int act, float slope, float eps, float* sums, void* dx, int dx_pitch, void* dgb_img,
                                   int dgb_pitch, float* dgb_st, int dgb_st_pitch, const void* res, int res_pitch, void* stream) {
  int rc = norm_check("s2p_in_norm_bwd", dtype, C, pitch, gb_pitch, da_pitch); if (rc) return rc;
  if (res && (res_pitch != dx_pitch || dx_pitch != C)) S2P_FAIL(-1, "s2p_in_norm_bwd_res: res must have the layout of dx (pitch == C)");
  const int maxhw = dtype == S2P_F32 ? 256 : 512;
  const bool simple_act = act == S2P_ACT_NONE || act == S2P_ACT_RELU || act == S2P_ACT_LRELU;
  if (HW > maxhw || !simple_act || s2p_env_set("S2P_NO_FUSED_NORM")) {
    if (!sums) S2P_FAIL(-1, "s2p_in_norm_bwd: the two-kernel path needs the `sums` workspace");
    rc = s2p_in_bwd_reduce(dtype, da, da_pitch, x, N, HW, C, pitch, stats, gb_img, gb_pitch, gb_st, gb_st_pitch, act, slope, eps,
                           sums, stream);
    if (rc) return rc;
    rc = s2p_in_bwd_apply(dtype, da, da_pitch, x, N, HW, C, pitch, stats, gb_img, gb_pitch, gb_st, gb_st_pitch, act, slope, eps,
                          sums, dx, dx_pitch, dgb_img, dgb_pitch, dgb_st, dgb_st_pitch, stream);
    if (rc || !res) return rc;
    return s2p_add(dtype, dx, res, dx, (int64_t)N * HW * C, stream);          // two-kernel path: the residual is a third pass
  }
  if (dx_pitch % (dtype == S2P_F32 ? 4 : 8) || dgb_pitch % (dtype == S2P_F32 ? 4 : 8))
    S2P_FAIL(-1, "s2p_in_norm_bwd: bad output pitch");
  if (dgb_st && dgb_st_pitch < 2 * C) S2P_FAIL(-1, "s2p_in_norm_bwd: dgb_st pitch < 2*C");
  NormArgs a{}; a.x = x; a.da = da; a.stats = stats; a.gb = gb_img; a.gbst = gb_st;
  a.y = dx; a.dgb = dgb_img; a.dgbst = dgb_st; a.dgbst_pitch = dgb_st_pitch;
  a.N = N; a.HW = HW; a.C = C; a.x_pitch = pitch; a.da_pitch = da_pitch; a.gb_pitch = gb_pitch;
  a.gbst_pitch = gb_st_pitch; a.y_pitch = dx_pitch; a.dgb_pitch = dgb_pitch; a.act = act; a.slope = slope; a.eps = eps;
  a.res = res; a.res_pitch = res_pitch;
  const bool half = s2p_env_set("S2P_NORM_CS32") && C % 32 == 0;
  dim3 grid(N, cdiv(C, half ? 32 : 64));
  hipStream_t st = (hipStream_t)stream;
  if (const int th = dgb_img ? 0 : small_plane(dtype, HW, gb_img)) {
    const dim3 sg(N, cdiv(C, 64));
    if (th == 256) { if (dtype == S2P_F32) hipLaunchKernelGGL((in_fused_bwd_kernel<float, 64, 256, false>), sg, dim3(256), 0, st, a);
                     else hipLaunchKernelGGL((in_fused_bwd_kernel<__bf16, 64, 256, false>), sg, dim3(256), 0, st, a); }
    else { if (dtype == S2P_F32) hipLaunchKernelGGL((in_fused_bwd_kernel<float, 64, 512, false>), sg, dim3(512), 0, st, a);
           else hipLaunchKernelGGL((in_fused_bwd_kernel<__bf16, 64, 512, false>), sg, dim3(512), 0, st, a); }
    S2P_CHECK_LAUNCH("in_fused_bwd_kernel (small planes)");
    return 0;
  }
  if (dtype == S2P_F32) { if (half) hipLaunchKernelGGL((in_fused_bwd_kernel<float, 32>), grid, dim3(1024), 0, st, a);
                          else hipLaunchKernelGGL((in_fused_bwd_kernel<float, 64>), grid, dim3(1024), 0, st, a); }
  else { if (half) hipLaunchKernelGGL((in_fused_bwd_kernel<__bf16, 32>), grid, dim3(1024), 0, st, a);
         else hipLaunchKernelGGL((in_fused_bwd_kernel<__bf16, 64>), grid, dim3(1024), 0, st, a); }
  S2P_CHECK_LAUNCH("in_fused_bwd_kernel");
  return 0;
}

static void channel_sum_geom(int64_t pixels, int C, int& nb, int& rows) {
  // pixel blocks: ~1 500 workgroups over (channel slabs x blocks), at least 128 pixels each, at most 256 blocks
  const int slabs = (C + 63) / 64;
  nb = (1536 + slabs - 1) / slabs;
  if ((int64_t)nb * 128 > pixels) nb = (int)(pixels / 128);
  if (nb > 256) nb = 256; if (nb < 1) nb = 1;
  rows = (int)((pixels + nb - 1) / nb);
  nb = (int)((pixels + rows - 1) / rows);
}
// scratch bytes of the atomics-free form (s2p_channel_sum_det)
size_t s2p_channel_sum_ws_bytes(int64_t pixels, int C) {
  if (pixels <= 0 || C <= 0) return 0;
  int nb, rows; channel_sum_geom(pixels, C, nb, rows);
  return (size_t)nb * cdiv(C, 64) * 64 * sizeof(float);
}
// db[c] += sum over pixels of dy[p][c].  ws == nullptr (or too small): fp32 atomics; otherwise partial sums + a fixed-order reduce.
int s2p_channel_sum_det(int dtype, const void* dy, int64_t pixels, int C, int pitch, float* db, void* ws, size_t ws_bytes,
                        void* stream) {
  if (dtype != S2P_F32 && dtype != S2P_BF16) S2P_FAIL(-1, "s2p_channel_sum: bad dtype");
  int ce = dtype == S2P_F32 ? 4 : 8;
  if (pitch % ce) S2P_FAIL(-1, "s2p_channel_sum: pitch must be a multiple of %d", ce);
  if (pixels <= 0) return 0;
  int nb, rows; channel_sum_geom(pixels, C, nb, rows);
  dim3 grid(cdiv(C, 64), nb);
  float* part = (ws && ws_bytes >= s2p_channel_sum_ws_bytes(pixels, C) && nb > 1) ? (float*)ws : nullptr;
  if (dtype == S2P_F32)
    hipLaunchKernelGGL(channel_sum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                       (long long)pixels, C, pitch, db, rows, part);
  else
    hipLaunchKernelGGL(channel_sum_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy,
                       (long long)pixels, C, pitch, db, rows, part);
  S2P_CHECK_LAUNCH("channel_sum_kernel");
  if (part) {
    s2p_partial_reduce(part, nb, (long long)grid.x * 64, C, db, (hipStream_t)stream);
    S2P_CHECK_LAUNCH("s2p_partial_reduce_kernel(channel_sum)");
  }
  return 0;
}

extern "C" int s2p_channel_sum(int dtype, const void* dy, int64_t pixels, int C, int pitch, float* db,
                               void* stream) {
  return s2p_channel_sum_det(dtype, dy, pixels, C, pitch, db, nullptr, 0, stream);
}
