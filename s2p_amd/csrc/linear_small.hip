// The state path: positional encoding -> 4 x (Linear + LeakyReLU) -> per-norm affine
// Linear(256 -> 12*2C), all fp32, M = batch (64) rows.  0.6 MFLOP per image: these layers are LATENCY-bound, and the
// generic implicit-GEMM conv kernel spends ~45 us per layer on them (2 workgroups, a 16..23-step staged K loop).
// Here a layer is one short launch of many small workgroups, on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 operands,
// fp32 accumulate), operands straight from global memory into the MFMA registers -- NO LDS at all (round 3; the LDS-staged
// VALU form these replaced took ~20 us per layer, and it was the kernel family the LDS co-residency hazard of DESIGN.md
// section 4 was found on: a kernel without LDS accesses is outside that class by construction):
//   forward / dgrad : y[M][N] = f(x)[M][Kr] . W[N][Kr]^T (+ bias, activation)   a wave owns a 16 x 16 tile, a workgroup the
//                     four 16-row tiles of a 64-row block (they share the W rows); optional split of the reduction Kr over
//                     blockIdx.z with partial tiles in a workspace and a fixed-order reduce (the 6144-deep dgrad of the
//                     affine layer);
//   wgrad (+ bias)  : dW[N][K] += sum_m dpre[m][n] x[m][k]                       a wave owns 16 n x 16 k, loop over the batch;
// where f / dpre fold the LeakyReLU derivative of the layer's saved OUTPUT into the operand staging, so the backward of a
// layer is two short launches (wgrad + bias grad; dgrad) -- no act_bwd pass, no atomics, fixed summation order.
// (One launch per layer, not one per chain: a layer needs every output of the previous one, and on this chip a kernel
// boundary (~1.5 us, hipGraph) is cheaper than an in-kernel grid barrier (~4-7 us).)
#include "s2p_common.h"

struct LinArgs {
  const float* x; const float* xact; const float* w; const float* bias; float* y; float* part;
  // backward extras
  const float* dy; const float* yact; const float* xin; float* dw; float* db;
  int M, Kr, N, x_pitch, xact_pitch, w_row, y_pitch, n_store;
  int act, in_act; float slope;
  int ksplit, k_per_split;
  // wgrad
  int K, xin_pitch, dy_pitch, yact_pitch, dw_row, k_real;
};

__device__ __forceinline__ float lin_actgrad(float yv, int act, float slope) {
  return act == S2P_ACT_LRELU ? (yv > 0.f ? 1.f : slope) : (act == S2P_ACT_RELU ? (yv > 0.f ? 1.f : 0.f) : 1.f);
}

// y tile of  x'[M][Kr] . W[N][Kr]^T,  x' = x * act'(xact) when xact != nullptr.
// MFMA 16x16x4 f32 operand layout: lane l = 16 j + i holds A[row i][k j] and B[k j][col i]; the result register r of lane l is
// D[row 4 j + r][col i].  A k-chunk of 16 is loaded as ONE float4 per lane and operand (lane group j takes k = 16 t + 4 j .. + 3)
// and consumed by four MFMAs (MFMA e uses component e: the same k permutation on both operands, so the sum is the plain dot
// product).  U chunks are loaded before the first MFMA: 2 U (3 U with xact) independent 16-byte loads in flight per lane.
#ifndef S2P_LIN_U
#define S2P_LIN_U 8
#endif
__global__ __launch_bounds__(256) void lin_fwd_kernel(const LinArgs a) {
  constexpr int U = S2P_LIN_U;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, j = lane >> 4;
  const int nb = blockIdx.x * 16, mb = blockIdx.y * 64 + wave * 16, z = blockIdx.z;
  if (mb >= a.M) return;                                   // (wave-uniform)
  const int k0 = z * a.k_per_split, k1 = k0 + a.k_per_split < a.Kr ? k0 + a.k_per_split : a.Kr;
  const int m = mb + i, n = nb + i;
  const bool mok = m < a.M, nok = n < a.N;
  const float* xr = a.x + (size_t)(mok ? m : 0) * a.x_pitch;
  const float* ar = a.xact ? a.xact + (size_t)(mok ? m : 0) * a.xact_pitch : nullptr;
  const float* wr = a.w + (size_t)(nok ? n : 0) * a.w_row;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kc = k0; kc < k1; kc += 16 * U) {
    f32x4 xv[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kc + 16 * u + 4 * j;                   // Kr, k_per_split and the pitches are multiples of 4: a float4 is in or out
      xv[u] = (mok && k < k1) ? *(const f32x4*)(xr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
      wv[u] = (nok && k < k1) ? *(const f32x4*)(wr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ar && mok && k < k1) {
        const f32x4 yv = *(const f32x4*)(ar + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[u][e] *= lin_actgrad(yv[e], a.in_act, a.slope);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][e], wv[u][e], acc, 0, 0, 0);
  }
  if (n >= a.n_store) return;
  const float b = (a.bias && nok && a.ksplit <= 1) ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mo = mb + 4 * j + r;
    if (mo >= a.M) continue;
    if (a.ksplit > 1) { a.part[((size_t)z * a.M + mo) * a.n_store + n] = acc[r]; continue; }   // partial tile -> workspace [z][M][n_store]
    float v = nok ? acc[r] + b : 0.f;
    v = a.act == S2P_ACT_LRELU ? (v > 0.f ? v : v * a.slope) : (a.act == S2P_ACT_RELU ? (v > 0.f ? v : 0.f) : v);
    a.y[(size_t)mo * a.y_pitch + n] = v;
  }
}

// y[m][n] = sum over the K splits, in split order
__global__ __launch_bounds__(256) void lin_splitk_reduce_kernel(const LinArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)a.M * a.n_store) return;
  const int m = (int)(i / a.n_store), n = (int)(i - (long long)m * a.n_store);
  float s = 0.f;
  for (int z = 0; z < a.ksplit; ++z) s += a.part[((size_t)z * a.M + m) * a.n_store + n];
  a.y[(size_t)m * a.y_pitch + n] = s;
}

// dW[N][K] += dpre^T x, db[N] += sum_m dpre   (dpre = dy * act'(y)).  D[n][k] = sum_m A[n][m] B[m][k]: per MFMA (4 batch rows)
// a lane loads ONE dpre value and ONE x value (16 lanes = 64 contiguous bytes of a row); the bias gradient is a second MFMA
// against a column of ones (no cross-lane reduction).  All of a 64-row batch's loads are in flight before the first MFMA.
__global__ __launch_bounds__(256) void lin_wgrad_kernel(const LinArgs a) {
  constexpr int U = 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, j = lane >> 4;
  const int kt_n = (a.K + 63) / 64;
  const int nb = (blockIdx.x / kt_n) * 16, kb = (blockIdx.x % kt_n) * 64 + wave * 16;
  if (kb >= a.K) return;                                   // (wave-uniform)
  const int n = nb + i, k = kb + i;
  const bool nok = n < a.N, kok = k < a.K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
  const bool want_b = a.db && kb == 0;
  for (int m0 = 0; m0 < a.M; m0 += 4 * U) {
    float dv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + 4 * u + j;
      const bool mok = m < a.M;
      dv[u] = (mok && nok) ? a.dy[(size_t)m * a.dy_pitch + n] : 0.f;
      if (a.yact && mok && nok) dv[u] *= lin_actgrad(a.yact[(size_t)m * a.yact_pitch + n], a.act, a.slope);
      xv[u] = (mok && kok) ? a.xin[(size_t)m * a.xin_pitch + k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u], xv[u], acc, 0, 0, 0);
      if (want_b) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u], 1.f, accb, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int no = nb + 4 * j + r;
    if (no >= a.N) continue;
    if (k < a.k_real) a.dw[(size_t)no * a.dw_row + k] += acc[r];
    if (want_b && i == 0) a.db[no] += accb[r];
  }
}

#ifdef S2P_DIAG_BUILD
// ---- diagnostics build only: the LDS-staged VALU kernels of rounds 2-3 (S2P_LIN_LDS=1 selects them; tools/bench_lin.py times them
// against the MFMA kernels above).  They are where the co-residency wrong-result hazard was first seen (lanes 48..63, ~1 % of a
// partial sum, beside an LDS-DMA conv kernel).  Rounds 2-3 read that as an LDS effect of the MERGED reads hipcc formed on the
// odd-pitch tiles and bisected read forms (S2P_LIN_LDS_MODE 1..4: plain loads at pitch 65 / 68, element-wise lds_ld() on one tile
// or the other).  Round 4 showed the cause to be PACKED fp32 instructions with an op_sel swizzle: the merged
// reads only put the operands in adjacent registers, which let the SLP vectoriser pack the FMAs; lds_ld() blocked that.  With the
// library built without packed fp32 every mode is safe; the modes stay as the record of that bisection.
#ifndef S2P_LIN_LDS_MODE
#define S2P_LIN_LDS_MODE 0
#endif
constexpr int LIN_LD = S2P_LIN_LDS_MODE == 2 ? 68 : 65;
__device__ __forceinline__ float lin_ld_x(const float* p) { return (S2P_LIN_LDS_MODE == 0 || S2P_LIN_LDS_MODE == 4) ? lds_ld(p) : *p; }
__device__ __forceinline__ float lin_ld_w(const float* p) { return (S2P_LIN_LDS_MODE == 0 || S2P_LIN_LDS_MODE == 3) ? lds_ld(p) : *p; }

// y tile [64 rows][16 cols] of  x'[M][Kr] . W[N][Kr]^T,  x' = x * act'(xact) when xact != nullptr
__device__ __forceinline__ void lin_gemm_tile(const LinArgs& a, int mb, int nb, int k0, int k1, float (&acc)[4], float* xs, float* ws) {
  constexpr int KC = 64, LD = LIN_LD;
  const int t = threadIdx.x, mq = t & 15, c = t >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = 0.f;
  for (int kc = k0; kc < k1; kc += KC) {
    __syncthreads();
    // stage x' : 64 rows x 64 k  (4 float4 per thread), W : 16 rows x 64 k (1 float4 per thread)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i, r = idx >> 4, q = idx & 15;
      const int m = mb + r, k = kc + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.M && k < k1) {
        v = *(const f32x4*)(a.x + (size_t)m * a.x_pitch + k);
        if (a.xact) {
          const f32x4 yv = *(const f32x4*)(a.xact + (size_t)m * a.xact_pitch + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= lin_actgrad(yv[e], a.in_act, a.slope);
        }
        if (k + 4 > k1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e >= k1) v[e] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) xs[r * LD + q * 4 + e] = v[e];
    }
    {
      const int r = t >> 4, q = t & 15;
      const int n = nb + r, k = kc + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < a.N && k < k1) {
        v = *(const f32x4*)(a.w + (size_t)n * a.w_row + k);
        if (k + 4 > k1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e >= k1) v[e] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ws[r * LD + q * 4 + e] = v[e];
    }
    __syncthreads();
    // element-wise LDS reads: see lds_ld() above
#pragma unroll 8
    for (int k = 0; k < KC; ++k) {
      const float wv = lin_ld_w(ws + c * LD + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(lin_ld_x(xs + (4 * mq + i) * LD + k), wv, acc[i]);
    }
  }
}

__global__ __launch_bounds__(256) void lin_fwd_lds_kernel(const LinArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[64 * LIN_LD], ws[16 * LIN_LD];
  const int nb = blockIdx.x * 16, mb = blockIdx.y * 64, z = blockIdx.z;
  const int k0 = z * a.k_per_split, k1 = k0 + a.k_per_split < a.Kr ? k0 + a.k_per_split : a.Kr;
  float acc[4];
  lin_gemm_tile(a, mb, nb, k0, k1, acc, xs, ws);
  const int t = threadIdx.x, mq = t & 15, c = t >> 4, n = nb + c;
  if (n >= a.n_store) return;
  if (a.ksplit > 1) {                                     // partial tile -> workspace [z][M][n_store]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + 4 * mq + i;
      if (m < a.M) a.part[((size_t)z * a.M + m) * a.n_store + n] = acc[i];
    }
    return;
  }
  const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mb + 4 * mq + i;
    if (m < a.M) {
      float v = n < a.N ? acc[i] + b : 0.f;
      v = a.act == S2P_ACT_LRELU ? (v > 0.f ? v : v * a.slope) : (a.act == S2P_ACT_RELU ? (v > 0.f ? v : 0.f) : v);
      a.y[(size_t)m * a.y_pitch + n] = v;
    }
  }
}

// dW[N][K] += dpre^T x, db[N] += sum_m dpre   (dpre = dy * act'(y))
__global__ __launch_bounds__(256) void lin_wgrad_lds_kernel(const LinArgs a) {
  constexpr int LDX = LIN_LD, LDD = S2P_LIN_LDS_MODE == 2 ? 20 : 17;   // pitches of the x / dpre tiles (mode 2: 16-byte-aligned rows)
  __shared__ __attribute__((aligned(16))) float xs[64 * LDX], ws[64 * LDD];
  const int t = threadIdx.x;
  // ---- wgrad tile: 16 outputs n x 64 inputs k; dpre staged [m][16], x staged [m][64]; thread = (4 k, 1 n) ----------
  const int kt_n = (a.K + 63) / 64;
  const int nb = (blockIdx.x / kt_n) * 16, kb = (blockIdx.x % kt_n) * 64;
  const int kq = t & 15, c = t >> 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float accb = 0.f;
  float* ds = ws;                                          // [64 m][16 n] (+1 pad)
  for (int mb = 0; mb < a.M; mb += 64) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // x : 64 rows x 64 k
      const int idx = t + 256 * i, r = idx >> 4, q = idx & 15;
      const int m = mb + r, k = kb + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.M && k < a.K) v = *(const f32x4*)(a.xin + (size_t)m * a.xin_pitch + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) xs[r * LDX + q * 4 + e] = v[e];
    }
    {                                                      // dpre : 64 rows x 16 n
      const int r = t >> 2, q = t & 3;
      const int m = mb + r, n = nb + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < a.M && n < a.N) {
        v = *(const f32x4*)(a.dy + (size_t)m * a.dy_pitch + n);
        if (a.yact) {
          const f32x4 yv = *(const f32x4*)(a.yact + (size_t)m * a.yact_pitch + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= lin_actgrad(yv[e], a.act, a.slope);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) ds[r * LDD + q * 4 + e] = v[e];
    }
    __syncthreads();
    const int mlim = a.M - mb < 64 ? a.M - mb : 64;
    for (int m = 0; m < mlim; ++m) {
      const float d = lin_ld_w(ds + m * LDD + c);            // element-wise LDS reads: see lds_ld()
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(d, lin_ld_x(xs + m * LDX + 4 * kq + i), acc[i]);
      accb += d;
    }
  }
  const int n = nb + c;
  if (n < a.N) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb + 4 * kq + i;
      if (k < a.k_real) a.dw[(size_t)n * a.dw_row + k] += acc[i];
    }
    if (a.db && kb == 0 && kq == 0) a.db[n] += accb;
  }
}

#endif  // S2P_DIAG_BUILD

#ifdef S2P_DIAG_BUILD
static inline bool lin_use_lds() { static const int v = s2p_env_set("S2P_LIN_LDS"); return v != 0; }
#define LIN_LAUNCH_FWD(grid, st, args) do { if (lin_use_lds()) hipLaunchKernelGGL(lin_fwd_lds_kernel, grid, dim3(256), 0, st, args); \
                                            else hipLaunchKernelGGL(lin_fwd_kernel, grid, dim3(256), 0, st, args); } while (0)
#define LIN_LAUNCH_WGRAD(grid, st, args) do { if (lin_use_lds()) hipLaunchKernelGGL(lin_wgrad_lds_kernel, grid, dim3(256), 0, st, args); \
                                              else hipLaunchKernelGGL(lin_wgrad_kernel, grid, dim3(256), 0, st, args); } while (0)
#else
#define LIN_LAUNCH_FWD(grid, st, args) hipLaunchKernelGGL(lin_fwd_kernel, grid, dim3(256), 0, st, args)
#define LIN_LAUNCH_WGRAD(grid, st, args) hipLaunchKernelGGL(lin_wgrad_kernel, grid, dim3(256), 0, st, args)
#endif

static int lin_check(const char* who, int M, int K, int N, int xp) {
  if (M <= 0 || K <= 0 || N <= 0) S2P_FAIL(-1, "%s: empty problem", who);
  if (K % 4 || xp % 4) S2P_FAIL(-1, "%s: K and pitches must be multiples of 4 floats", who);
  return 0;
}

// y[M][y_pitch] = act(x[M][K] . w[N][w_row]^T + bias); columns [N, n_store) are written as zeros (channel padding)
extern "C" int s2p_linear_fwd(const float* x, int M, int K, int x_pitch, const float* w, int w_row, const float* bias, int N,
                              int act, float slope, float* y, int y_pitch, int n_store, void* stream) {
  int rc = lin_check("s2p_linear_fwd", M, K, N, x_pitch); if (rc) return rc;
  if (!x || !w || !y || w_row % 4 || n_store < N || n_store > y_pitch) S2P_FAIL(-1, "s2p_linear_fwd: bad arguments");
  LinArgs a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.M = M; a.Kr = K; a.N = N; a.x_pitch = x_pitch; a.w_row = w_row;
  a.y_pitch = y_pitch; a.n_store = n_store; a.act = act; a.slope = slope; a.ksplit = 1; a.k_per_split = K;
  LIN_LAUNCH_FWD(dim3(cdiv(n_store, 16), cdiv(M, 64), 1), (hipStream_t)stream, a);
  S2P_CHECK_LAUNCH("lin_fwd_kernel");
  return 0;
}

extern "C" size_t s2p_linear_bwd_workspace(int M, int K, int N) {
  const int ks = N >= 2048 ? cdiv(N, 512) : 1;
  return ks > 1 ? (size_t)ks * M * ((K + 3) / 4 * 4) * sizeof(float) : 0;
}

// Backward of y = act(x . w^T + b) given dy = dL/dy and the layer's OUTPUT y (act != NONE):
//   dw[N][dw_row] += dpre^T x (columns < k_real), db[N] += sum_m dpre, dx[M][dx_pitch] = dpre . w  (dx may be NULL),
// dpre = dy * act'(y).  w_bwd: [K][wb_row] = the transpose of w (row k holds w[:, k]).
extern "C" int s2p_linear_bwd(const float* x, int x_pitch, const float* dy, int dy_pitch, const float* y, int y_pitch, int M,
                              int K, int k_real, int N, const float* w_bwd, int wb_row, int act, float slope, float* dw,
                              int dw_row, float* db, float* dx, int dx_pitch, void* workspace, size_t workspace_bytes,
                              void* stream) {
  int rc = lin_check("s2p_linear_bwd", M, K, N, x_pitch); if (rc) return rc;
  if (!x || !dy || !dw || dy_pitch % 4 || N % 4) S2P_FAIL(-1, "s2p_linear_bwd: bad arguments (N and pitches must be multiples of 4)");
  if (act != S2P_ACT_NONE && (!y || y_pitch % 4)) S2P_FAIL(-1, "s2p_linear_bwd: the activation output is needed");
  if (dx && (!w_bwd || wb_row % 4)) S2P_FAIL(-1, "s2p_linear_bwd: dx needs w_bwd");
  hipStream_t st = (hipStream_t)stream;
  LinArgs a{};
  a.M = M; a.act = act; a.slope = slope;
  // wgrad part
  a.xin = x; a.xin_pitch = x_pitch; a.dy = dy; a.dy_pitch = dy_pitch; a.yact = act != S2P_ACT_NONE ? y : nullptr;
  a.yact_pitch = y_pitch; a.dw = dw; a.dw_row = dw_row; a.db = db; a.K = K; a.k_real = k_real; a.N = N;
  const int wg_blocks = cdiv(N, 16) * cdiv(K, 64);
  LIN_LAUNCH_WGRAD(dim3(wg_blocks), st, a);
  S2P_CHECK_LAUNCH("lin_wgrad_kernel");
  if (!dx) return 0;
  // dgrad: "x" = dy (with the activation derivative folded in), reduction over N, output columns = the K inputs
  const int ks = N >= 2048 ? cdiv(N, 512) : 1;
  LinArgs g = a;
  g.x = dy; g.x_pitch = dy_pitch; g.xact = a.yact; g.xact_pitch = y_pitch; g.in_act = act; g.w = w_bwd; g.w_row = wb_row;
  g.bias = nullptr; g.y = dx; g.y_pitch = dx_pitch; g.Kr = N; g.act = S2P_ACT_NONE;
  const int kcols = (K + 3) / 4 * 4;
  g.n_store = kcols <= dx_pitch ? kcols : K;
  g.N = K;
  if (ks == 1) {
    g.ksplit = 1; g.k_per_split = g.Kr;
    LIN_LAUNCH_FWD(dim3(cdiv(g.n_store, 16), cdiv(M, 64), 1), st, g);
    S2P_CHECK_LAUNCH("lin_fwd_kernel(dgrad)");
    return 0;
  }
  const size_t need = (size_t)ks * M * g.n_store * sizeof(float);
  if (!workspace || workspace_bytes < need) S2P_FAIL(-1, "s2p_linear_bwd: workspace of %zu bytes needed", need);
  g.part = (float*)workspace; g.ksplit = ks; g.k_per_split = cdiv(g.Kr, ks); g.k_per_split = (g.k_per_split + 63) / 64 * 64;
  g.ksplit = cdiv(g.Kr, g.k_per_split);
  LIN_LAUNCH_FWD(dim3(cdiv(g.n_store, 16), cdiv(M, 64), g.ksplit), st, g);
  S2P_CHECK_LAUNCH("lin_fwd_kernel(dgrad, split)");
  hipLaunchKernelGGL(lin_splitk_reduce_kernel, dim3(cdiv((long long)M * g.n_store, 256)), dim3(256), 0, st, g);
  S2P_CHECK_LAUNCH("lin_splitk_reduce_kernel");
  return 0;
}
