// Implicit-GEMM 2-D convolution family for gfx950 (MI355X), NHWC activations.
//
// One "gather GEMM" kernel computes  D[co][m] = sum_{t,c} Wp[co][wt(t)][c] * X[n, qy*is+dy(t), qx*is+dx(t), c]
// for GEMM pixel m=(n,qy,qx) and stores D at output pixel (n, qy*os+oy0, qx*os+ox0).  With the tap table
// (dy,dx,wt) chosen on the host this one kernel is
//   * conv2d forward (any KHxKW / stride / zero or reflect pad / groups),
//   * conv_transpose2d forward and strided-conv dgrad (one launch per output phase: sub-pixel decomposition,
//     so no MFMA lane ever multiplies a structural zero),
//   * stride-1 conv dgrad and conv_transpose dgrad (plain gather with the transposed packed weight).
// MFMA: A operand = packed weights (rows = co, K contiguous), B operand = gathered pixels (K contiguous in NHWC),
// v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32 parity path).  LDS rows are 64 B of K
// padded to 80 B (conflict-free ds_read_b128); global->register->LDS staging, double buffered, one barrier
// per K step; epilogue goes through LDS so every global store is a full 16-B chunk along the channel axis.
#include "s2p_common.h"
#include "conv_plane.h"
#include "conv_planeg.h"
#include <type_traits>

#define MAX_TAPS 64

constexpr int PHASE_TAPS = 16, MAX_PHASES = 4;
struct GatherArgs {
  const void* x; const void* w; const float* bias; const void* aux; const void* aux2; void* y;
  int M, Hi, Wi, Qh, Qw;
  int Cin, x_pitch, x_gstride;
  int Cout, Cst, y_pitch, y_gstride;
  int Ho, Wo;
  int istride, ostride, oy0, ox0;
  int T, Ktot, w_row;
  long long w_gstride;
  int reflect, act, epi, gact;
  float slope, gslope;
  int npix_tiles, nco_tiles;
  int halo_lo, halo_hi;        // halo kernel: pixels needed before / after the tile in raster order
  int diag;                    // timing-only ablation (S2P_DIAG env): 1 = skip in-loop loads, 2 = skip MFMAs
  int splitk, ksteps;          // generic fp32 path only: split-K over blockIdx.z with fp32 atomics into a zeroed y
  // bf16 LDS-DMA kernel, launches that cannot fill the chip: K split over blockIdx.z into `psplit` slices of `psteps` K
  // steps; each slice stores its fp32 partial tile to part[z][co][m] (m padded to part_m) and conv_part_reduce_kernel
  // applies bias / activation / epilogue to the sum (fixed order: no atomics).  ws / ws_bytes: caller's scratch.
  float* part; int psplit, psteps, part_m;
  void* ws; size_t ws_bytes;
  size_t* plan;                // non-null: dry run -- report the scratch bytes this launch would use, launch nothing
  const PlaneMat* mat; int* mat_done;   // optional fused InstanceNorm + MAT epilogue (plane-resident kernel only): *mat_done = 1 when applied
  unsigned x_bytes, w_bytes;   // fast path: buffer-descriptor sizes of the gathered tensor / packed weights (per group view)
  int tap[MAX_TAPS];   // (wt << 16) | ((dx & 0xff) << 8) | (dy & 0xff)
  // merged sub-pixel phases (strided dgrad / transposed fwd on the LDS-DMA kernel): blockIdx.z selects a record that
  // overrides the per-phase fields above, so the s*s short GEMMs of one layer are ONE launch
  int nphase, phase_fast;
  struct Phase { int Qh, Qw, M, oy0, ox0, T, Ktot, npix_tiles; int tap[PHASE_TAPS]; } ph[MAX_PHASES];
};

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

template <typename T> struct Mma;
template <> struct Mma<__bf16> { static constexpr int BK = 32; };
template <> struct Mma<float> { static constexpr int BK = 16; };

template <typename T, int BCO, int BPIX, int TCO, int TPIX>
__device__ __forceinline__ void conv_epilogue(const GatherArgs& a, f32x16 (&acc)[TCO][TPIX], char* smem, const int* rowoff,
                                              int g, int co_base, int wco0, int wpix0, int r, int h, int tid) {
  constexpr int CE = DT<T>::CE;
  constexpr int ERS = BCO * (int)sizeof(T) + 16;
  // ---- epilogue: bias + activation in registers, transpose through LDS, 16-B stores ----------
  // The activation / epilogue selectors are kernel arguments; they are resolved ONCE per wave here (uniform
  // selects of a negative-side slope), never per element: a per-element switch costs ~6 us per launch in scalar
  // branches and instruction fetch.  relu / lrelu / none all are  v > 0 ? v : v * ns  with ns = 0 / slope / 1.
  const float* bias = (a.bias && (blockIdx.z == 0 || a.nphase > 0)) ? a.bias + (size_t)g * a.Cout : nullptr;   // z: split-K slice or phase
  const bool act_generic = (a.act == S2P_ACT_TANH || a.act == S2P_ACT_SWISH);
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  if (bias) {
    // dword buffer loads: channels past Cout read as zero through the descriptor's range check (round 1 guarded every one of the
    // 16 loads with a branch: ~170 instructions of a tile whose K loop is 100-400)
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, a.Cout * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int co = co_base + wco0 + 32 * i + 8 * q4 + 4 * h;
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, co * 4, e * 4, 0));
#pragma unroll
        for (int j = 0; j < TPIX; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q4 + e] += bv[e];
      }
  }
  if (act_generic) {
#pragma unroll                                           // (static register indices: a rolled loop would spill acc)
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPIX; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float t = acc[i][j][e];
          acc[i][j][e] = a.act == S2P_ACT_TANH ? tanhf(t) : t / (1.f + expf(-t));
        }
  }
#pragma unroll
  for (int i = 0; i < TCO; ++i) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      int col = wco0 + 32 * i + 8 * q4 + 4 * h;     // local co of element e=0
#pragma unroll
      for (int j = 0; j < TPIX; ++j) {
        int prow_l = wpix0 + 32 * j + r;
        char* dst = smem + prow_l * ERS + col * (int)sizeof(T);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][4 * q4 + e];
          v[e] = act_generic ? t : lrelu_ns(t, ns);
        }
        if constexpr (sizeof(T) == 2) {
          bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *(bf16x4*)dst = o;
        } else {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)dst = o;
        }
      }
    }
  }
  __syncthreads();
  constexpr int CPR = BCO / CE;
  T* yg = (T*)a.y + (size_t)g * a.y_gstride;
  const T* auxg = a.aux ? (const T*)a.aux + (size_t)g * a.y_gstride : nullptr;
  const T* aux2g = a.aux2 ? (const T*)a.aux2 + (size_t)g * a.y_gstride : nullptr;
  // MUL_ACTGRAD factor from the saved activation output x:  tanh: 1 - x^2;  else  x > 0 ? 1 : gneg
  const bool epi_add = a.epi == S2P_EPI_ADD;
  const bool g_tanh = a.gact == S2P_ACT_TANH;
  const float gneg = a.gact == S2P_ACT_RELU ? 0.f : (a.gact == S2P_ACT_LRELU ? a.gslope : 1.f);
  for (int idx = tid; idx < BPIX * CPR; idx += 256) {
    int row = idx / CPR, ch = idx - row * CPR;
    int off = rowoff[row];
    int co0 = co_base + ch * CE;
    if (off < 0 || co0 >= a.Cst) continue;
    Chunk<T> c;
    c.raw = *(const u32x4*)(smem + row * ERS + ch * 16);
    size_t go = (size_t)off * a.y_pitch + co0;
    bool full = co0 + CE <= a.Cst;
    if (a.epi != S2P_EPI_STORE) {
      Chunk<T> x, x2;
      x2.raw = (u32x4){0u, 0u, 0u, 0u};
      if (full) {
        x.raw = *(const u32x4*)(auxg + go);
        if (aux2g) x2.raw = *(const u32x4*)(aux2g + go);
      } else {
        x.raw = (u32x4){0u, 0u, 0u, 0u};
        for (int e = 0; e < CE; ++e) if (co0 + e < a.Cst) {
          x.set(e, to_f32(auxg[go + e]));
          if (aux2g) x2.set(e, to_f32(aux2g[go + e]));
        }
      }
      float ov[CE];
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        float v = c.get(e), xv = x.get(e);
        float f = g_tanh ? 1.f - xv * xv : (xv > 0.f ? 1.f : gneg);
        ov[e] = epi_add ? v + xv : (v + x2.get(e)) * f;
      }
      c.pack(ov);
    }
    if constexpr (sizeof(T) == 4) {
      if (a.splitk > 1) {
        for (int e = 0; e < CE; ++e) if (co0 + e < a.Cst) atomicAdd((float*)yg + go + e, c.get(e));
        continue;
      }
    }
    if (S2P_DIAGV(a) == 6 && c.raw[0] != 0x12345678u) continue;
    if (full) *(u32x4*)(yg + go) = c.raw;
    else for (int e = 0; e < CE; ++e) if (co0 + e < a.Cst) yg[go + e] = from_f32<T>(c.get(e));
  }
}

template <typename T, int BCO, int BPIX, int WCO, int WPIX>
__global__ __launch_bounds__(256, 2) void conv_gather_kernel(const GatherArgs a) {
  constexpr int CE = DT<T>::CE;
  constexpr int BK = Mma<T>::BK;          // 64 bytes of K per row
  constexpr int RS = 80;                  // LDS row stride (bytes)
  constexpr int TCO = BCO / WCO / 32, TPIX = BPIX / WPIX / 32;
  constexpr int NLW = (BCO + 63) / 64, NLP = BPIX / 64;
  constexpr int STAGE = (BCO + BPIX) * RS;
  constexpr int ERS = BCO * (int)sizeof(T) + 16;      // epilogue row stride
  constexpr int EPI = BPIX * ERS;
  constexpr int MAIN = (2 * STAGE > EPI ? 2 * STAGE : EPI);
  __shared__ __attribute__((aligned(16))) char smem[MAIN + BPIX * 4 + MAX_TAPS * 4];
  int* rowoff = (int*)(smem + MAIN);
  int* taps = rowoff + BPIX;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (observed round-robin); give each XCD a
  // contiguous run of tiles so blocks that share a pixel tile (all co tiles of it) hit the same L2.
  int nblk = a.npix_tiles * a.nco_tiles;
  int bid = blockIdx.x;
  {
    int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int co_tile = bid % a.nco_tiles, pix_tile = bid / a.nco_tiles;
  const int co_base = co_tile * BCO, pix_base = pix_tile * BPIX;

  if (tid < MAX_TAPS) taps[tid] = a.tap[tid];
  const int QQ = a.Qh * a.Qw;
  const double rcpQQ = s2p_rcp_f64(QQ), rcpQw = s2p_rcp_f64(a.Qw);
  if (tid < BPIX) {
    int m = pix_base + tid;
    int off = -1;
    if (m < a.M) {
      int rr, qx;
      const int n = divmod_rcp(m, QQ, rcpQQ, rr), qy = divmod_rcp(rr, a.Qw, rcpQw, qx);
      off = ((n * a.Ho + qy * a.ostride + a.oy0) * a.Wo + qx * a.ostride + a.ox0);
    }
    rowoff[tid] = off;
  }

  // ---- per-thread staging assignment -------------------------------------------------------
  const int jc = tid & 3;          // 16-byte chunk column inside the 64-byte K row
  const int r0 = tid >> 2;         // row 0..63 (+64*i)
  const int nk_all = (a.Ktot + BK - 1) / BK;
  const int kt0 = a.splitk > 1 ? (int)blockIdx.z * a.ksteps : 0;
  const int kt1 = a.splitk > 1 ? (kt0 + a.ksteps < nk_all ? kt0 + a.ksteps : nk_all) : nk_all;
  int kk = kt0 * BK + jc * CE;     // linear k of this thread's chunk
  int k_tap, k_c;                  // (tap, channel) of this thread's chunk, advanced by BK per step
  k_tap = kk / a.Cin; k_c = kk - k_tap * a.Cin;
  // pixel rows handled by this thread
  int p_py[NLP], p_px[NLP], p_base[NLP];
  bool p_ok[NLP];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    int m = pix_base + r0 + 64 * i;
    p_ok[i] = m < a.M;
    int mm = p_ok[i] ? m : 0;
    int rr, qx;
    const int n = divmod_rcp(mm, QQ, rcpQQ, rr), qy = divmod_rcp(rr, a.Qw, rcpQw, qx);
    p_py[i] = qy * a.istride; p_px[i] = qx * a.istride; p_base[i] = n * a.Hi;
  }
  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;

  u32x4 regW[NLW], regP[NLP];
  const int nk = kt1 > kt0 ? kt1 - kt0 : 0;

  __syncthreads();   // taps visible

  auto load_global = [&]() {
    const bool kok = kk < a.Ktot;
    const int ti = taps[k_tap < MAX_TAPS ? k_tap : 0];
    const int dy = (int)(signed char)(ti & 0xff), dx = (int)(signed char)((ti >> 8) & 0xff), wt = ti >> 16;
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
      int row = r0 + 64 * i;
      int co = co_base + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < BCO && co < a.Cout && kok)
        v = *(const u32x4*)(wg + (size_t)co * a.w_row + wt * a.Cin + k_c);
      regW[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      int iy = p_py[i] + dy, ix = p_px[i] + dx;
      if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.Hi ? 2 * a.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.Wi ? 2 * a.Wi - 2 - ix : ix);
      }
      bool ok = p_ok[i] && kok && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *(const u32x4*)(xg + ((size_t)(p_base[i] + iy) * a.Wi + ix) * a.x_pitch + k_c);
      regP[i] = v;
    }
    // advance to the next K step
    kk += BK; k_c += BK;
    while (k_c >= a.Cin) { k_c -= a.Cin; ++k_tap; }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
      int row = r0 + 64 * i;
      if (row < BCO) *(u32x4*)(base + row * RS + jc * 16) = regW[i];
    }
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      int row = BCO + r0 + 64 * i;
      *(u32x4*)(base + row * RS + jc * 16) = regP[i];
    }
  };

  f32x16 acc[TCO][TPIX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPIX; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wco0 = (wave / WPIX) * (TCO * 32);
  const int wpix0 = (wave % WPIX) * (TPIX * 32);

  if (nk > 0) {
    load_global();
    store_lds(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_global();
    const char* base = smem + (kt & 1) * STAGE;
    const char* wrow = base + (wco0 + r) * RS + h * 16;
    const char* prow = base + (BCO + wpix0 + r) * RS + h * 16;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[TCO], bf[TPIX];
#pragma unroll
        for (int i = 0; i < TCO; ++i) af[i] = *(const bf16x8*)(wrow + i * 32 * RS + s * 32);
#pragma unroll
        for (int j = 0; j < TPIX; ++j) bf[j] = *(const bf16x8*)(prow + j * 32 * RS + s * 32);
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
          for (int j = 0; j < TPIX; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 af[TCO], bf[TPIX];
#pragma unroll
        for (int i = 0; i < TCO; ++i) af[i] = *(const f32x4*)(wrow + i * 32 * RS + s * 32);
#pragma unroll
        for (int j = 0; j < TPIX; ++j) bf[j] = *(const f32x4*)(prow + j * 32 * RS + s * 32);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPIX; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_lds((kt + 1) & 1);
    __syncthreads();
  }

  conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
}

// ------------------------------------------------------------------------------------------------
// Fast path (bf16, Cin % 64 == 0, zero padding, tensors < 2 GiB): the bulk of the FLOPs.
//   * BK = 64: one K step = one tap x 64 channels, so the tap is block-uniform (scalar) and steps are twice as long
//     (16 MFMAs per wave between barriers);
//   * global loads are buffer loads with a hardware range check: an invalid gather (padding, row >= M, co >= Cout)
//     is expressed as an out-of-range offset that returns zeros -- no exec-mask branches, no per-load predicates;
//   * validity of (pixel row, tap) is a 64-bit mask computed once per row; per step it costs a shift and a select;
//   * all offsets are 32-bit byte offsets (host checks the tensors are < 2 GiB).
template <int BCO, int BPIX, int WCO, int WPIX>
__global__ __launch_bounds__(256, 2) void conv_fast_kernel(const GatherArgs a) {
  typedef __bf16 T;
  constexpr int CE = 8, BK = 64, RS = 144;           // 128 B of K per row + 16 B pad (conflict-free ds_read_b128)
  constexpr int TCO = BCO / WCO / 32, TPIX = BPIX / WPIX / 32;
  constexpr int NLW = BCO / 32, NLP = BPIX / 32;
  constexpr int STAGE = (BCO + BPIX) * RS;
  constexpr int ERS = BCO * 2 + 16;
  constexpr int EPI = BPIX * ERS;
  constexpr int MAIN = (2 * STAGE > EPI ? 2 * STAGE : EPI);
  __shared__ __attribute__((aligned(16))) char smem[MAIN + BPIX * 4];
  int* rowoff = (int*)(smem + MAIN);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;
  int nblk = a.npix_tiles * a.nco_tiles;
  int bid = blockIdx.x;
  {
    int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int co_tile = bid % a.nco_tiles, pix_tile = bid / a.nco_tiles;
  const int co_base = co_tile * BCO, pix_base = pix_tile * BPIX;
  const int QQ = a.Qh * a.Qw;
  const double rcpQQ = s2p_rcp_f64(QQ), rcpQw = s2p_rcp_f64(a.Qw);
  if (tid < BPIX) {
    int m = pix_base + tid;
    int off = -1;
    if (m < a.M) {
      int rr, qx;
      const int n = divmod_rcp(m, QQ, rcpQQ, rr), qy = divmod_rcp(rr, a.Qw, rcpQw, qx);
      off = ((n * a.Ho + qy * a.ostride + a.oy0) * a.Wo + qx * a.ostride + a.ox0);
    }
    rowoff[tid] = off;
  }

  const int jc = tid & 7, r0 = tid >> 3;            // 16-B chunk column (8 per row), row 0..31 (+32*i)
  const unsigned OOB = 0x80000000u;
  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  unsigned p_byte[NLP];
  unsigned long long p_mask[NLP];
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    int m = pix_base + r0 + 32 * i;
    unsigned long long mask = 0ull;
    unsigned byte = 0u;
    if (m < a.M) {
      int rr, qx;
      const int n = divmod_rcp(m, QQ, rcpQQ, rr), qy = divmod_rcp(rr, a.Qw, rcpQw, qx);
      int py = qy * a.istride, px = qx * a.istride;
      byte = (unsigned)(((n * a.Hi + py) * a.Wi + px) * a.x_pitch * 2 + jc * 16);
      for (int t = 0; t < a.T; ++t) {
        int ti = a.tap[t];
        int iy = py + (int)(signed char)(ti & 0xff), ix = px + (int)(signed char)((ti >> 8) & 0xff);
        if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) mask |= 1ull << t;
      }
    }
    p_byte[i] = byte; p_mask[i] = mask;
  }
  unsigned w_byte[NLW];
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    int co = co_base + r0 + 32 * i;
    w_byte[i] = co < a.Cout ? (unsigned)(co * a.w_row * 2 + jc * 16) : OOB;
  }

  u32x4 regW[NLW], regP[NLP];
  const int nk = a.Ktot / BK;
  int tap = 0, c0 = 0;                                // block-uniform K position

  auto load_global = [&]() {
    const int ti = a.tap[tap];
    const int dy = (int)(signed char)(ti & 0xff), dx = (int)(signed char)((ti >> 8) & 0xff), wt = ti >> 16;
    const int toff = ((dy * a.Wi + dx) * a.x_pitch + c0) * 2;
    const int woff = (wt * a.Cin + c0) * 2;
#pragma unroll
    for (int i = 0; i < NLW; ++i)
      regW[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(w_byte[i] == OOB ? OOB : w_byte[i] + (unsigned)woff), 0, 0);
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      bool ok = (p_mask[i] >> tap) & 1ull;
      regP[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? p_byte[i] + (unsigned)toff : OOB), 0, 0);
    }
    c0 += BK;
    if (c0 >= a.Cin) { c0 = 0; ++tap; }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE + r0 * RS + jc * 16;
#pragma unroll
    for (int i = 0; i < NLW; ++i) *(u32x4*)(base + 32 * i * RS) = regW[i];
#pragma unroll
    for (int i = 0; i < NLP; ++i) *(u32x4*)(base + (BCO + 32 * i) * RS) = regP[i];
  };

  f32x16 acc[TCO][TPIX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPIX; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wco0 = (wave / WPIX) * (TCO * 32);
  const int wpix0 = (wave % WPIX) * (TPIX * 32);

  if (nk > 0) { load_global(); store_lds(0); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_global();
    const char* base = smem + (kt & 1) * STAGE;
    const char* wrow = base + (wco0 + r) * RS + h * 16;
    const char* prow = base + (BCO + wpix0 + r) * RS + h * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[TCO], bf[TPIX];
#pragma unroll
      for (int i = 0; i < TCO; ++i) af[i] = *(const bf16x8*)(wrow + i * 32 * RS + s * 32);
#pragma unroll
      for (int j = 0; j < TPIX; ++j) bf[j] = *(const bf16x8*)(prow + j * 32 * RS + s * 32);
#pragma unroll
      for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPIX; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_lds((kt + 1) & 1);
    __syncthreads();
  }
  conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
}

// ------------------------------------------------------------------------------------------------
template <typename T, int BCO, int BPIX, int WCO, int WPIX>
static int launch_cfg(GatherArgs& a, int groups, hipStream_t st) {
  if (a.plan) return 0;
  a.npix_tiles = cdiv(a.M, BPIX);
  a.nco_tiles = cdiv(a.Cst, BCO);
  int splits = 1;
  a.splitk = 1; a.ksteps = 0;
  if (sizeof(T) == 4 && a.T == 1 && a.act == S2P_ACT_NONE && a.epi == S2P_EPI_STORE) {   // 1x1 only: spatial convs stay atomics-free (bitwise reproducible)
    // latency-bound fp32 linear layers (tiny M, long K, a handful of tiles): split K, accumulate with fp32 atomics
    const int nk = cdiv(a.Ktot, Mma<T>::BK);
    const int tiles = a.npix_tiles * a.nco_tiles * groups;
    if (tiles < 64 && nk >= 32) {
      splits = 256 / tiles; if (splits > nk / 8) splits = nk / 8; if (splits < 1) splits = 1;
      if (splits > 1) {
        a.ksteps = cdiv(nk, splits); splits = cdiv(nk, a.ksteps); a.splitk = splits;
        // y must start from zero: clear exactly the region this launch owns (rows of Cst channels)
        if (a.ostride != 1 || groups != 1 || a.Cst != a.y_pitch) { a.splitk = 1; splits = 1; }
        else if (hipMemsetAsync(a.y, 0, (size_t)a.M * a.y_pitch * sizeof(T), st) != hipSuccess) { a.splitk = 1; splits = 1; }
      }
    }
  }
  dim3 grid(a.npix_tiles * a.nco_tiles, groups, splits);
  hipLaunchKernelGGL((conv_gather_kernel<T, BCO, BPIX, WCO, WPIX>), grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("conv_gather_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the fast path: `buffer_load_dwordx4 ... lds` writes the staged tile straight into LDS, so the
// register->LDS `ds_write_b128` pass (13 cycles per wave-instruction; with two co-resident workgroups it made the
// kernel LDS-bound) disappears, and so do the 32 staging VGPRs.  An LDS-DMA wave-instruction writes 1 KiB linearly
// (8 rows of 128 B), so rows cannot be padded; bank conflicts are avoided with an XOR swizzle of the 16-byte chunk
// index, chunk' = chunk ^ ((row >> 1) & 7), applied on the per-lane global SOURCE address and again on the
// fragment read (conflict-free for the 16-lane groups of ds_read_b128 with 128-byte rows).
template <int BCO, int BPIX, int WCO, int WPIX>
__global__ __launch_bounds__(256, 2) void conv_dma_kernel(const GatherArgs a) {
  typedef __bf16 T;
  constexpr int BK = 64, RS = 128;
  constexpr int TCO = BCO / WCO / 32, TPIX = BPIX / WPIX / 32;
  constexpr int NI = (BCO + BPIX) / 32;              // DMA instructions per wave per K step (8 rows each, 4 waves)
  constexpr int NIW = BCO / 32;                      // the first NIW of them stage weight rows
  constexpr int STAGE = (BCO + BPIX) * RS;
  constexpr int ERS = BCO * 2 + 16;
  constexpr int EPI = BPIX * ERS;
  constexpr int MAIN = (2 * STAGE > EPI ? 2 * STAGE : EPI);
  __shared__ __attribute__((aligned(1024))) char smem[MAIN + BPIX * 4];
  int* rowoff = (int*)(smem + MAIN);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;
  // per-phase problem fields (block-uniform)
  int pQh = a.Qh, pQw = a.Qw, pM = a.M, poy0 = a.oy0, pox0 = a.ox0, pT = a.T, pKtot = a.Ktot, pnpt = a.npix_tiles;
  const int* ptap = a.tap;
  int bid = blockIdx.x;
  if (a.nphase > 0) {
    // merged sub-pixel phases.  phase_fast: the phase is the FASTEST index inside an XCD's share of the 1-D grid (blocks b, b + 8,
    // ... share an XCD under the observed round-robin placement: speed only), so the phases of a pixel tile -- which gather the SAME
    // input rows -- run side by side on one XCD and the rows come from HBM once; with the phase in blockIdx.z (round 1) each phase
    // was a pass of its own over the input (4.05 x the input in L2 misses on the decoder's transposed conv)
    int pz = blockIdx.z;
    if (a.phase_fast) { const int xcd = bid & 7, k = bid >> 3; pz = k % a.nphase; bid = ((k / a.nphase) << 3) | xcd; }
    const GatherArgs::Phase& P = a.ph[pz];
    pQh = P.Qh; pQw = P.Qw; pM = P.M; poy0 = P.oy0; pox0 = P.ox0; pT = P.T; pKtot = P.Ktot; pnpt = P.npix_tiles;
    ptap = P.tap;
  }
  int nblk = pnpt * a.nco_tiles;
  if (bid >= nblk) return;                            // phases differ in size; the grid is sized for the largest
  {
    int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int co_tile = bid % a.nco_tiles, pix_tile = bid / a.nco_tiles;
  const int co_base = co_tile * BCO, pix_base = pix_tile * BPIX;
  const int QQ = pQh * pQw;
  const double rcpQQ = s2p_rcp_f64(QQ), rcpQw = s2p_rcp_f64(pQw);
  auto split_pixel = [&](int m, int& n, int& qy, int& qx) {
    int rr;
    n = divmod_rcp(m, QQ, rcpQQ, rr);
    qy = divmod_rcp(rr, pQw, rcpQw, qx);
  };
  if (tid < BPIX) {
    int m = pix_base + tid;
    int off = -1;
    if (m < pM) {
      int n, qy, qx;
      split_pixel(m, n, qy, qx);
      off = ((n * a.Ho + qy * a.ostride + poy0) * a.Wo + qx * a.ostride + pox0);
    }
    rowoff[tid] = off;
  }

  const unsigned OOB = 0x80000000u;
  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  // lane -> (row inside the 8-row piece, physical chunk); instruction i of wave w stages rows (4i + w)*8 .. +7
  const int lrow = lane >> 3, pc = lane & 7;
  unsigned w_byte[NIW];
  // pixel rows: byte offset of (n, py, px, chunk) and the pixel's (py, px); a tap's bounds test is two unsigned compares at issue
  // time (round 1 kept a 64-bit tap mask per row, built in a loop over the taps here)
  unsigned p_byte[NI - NIW];
  int p_y[NI - NIW], p_x[NI - NIW];
#pragma unroll
  for (int i = 0; i < NIW; ++i) {
    int row = (4 * i + wave) * 8 + lrow;
    int c = pc ^ ((row >> 1) & 7);
    int co = co_base + row;
    w_byte[i] = co < a.Cout ? (unsigned)(co * a.w_row * 2 + c * 16) : OOB;
  }
#pragma unroll
  for (int i = 0; i < NI - NIW; ++i) {
    int row = (4 * (i + NIW) + wave) * 8 + lrow;      // stage row (>= BCO)
    int c = pc ^ ((row >> 1) & 7);
    int m = pix_base + row - BCO;
    unsigned byte = 0u;
    int py = -0x40000000, px = 0;                     // rows past the end of the problem: no tap is in bounds
    if (m < pM) {
      int n, qy, qx;
      split_pixel(m, n, qy, qx);
      py = qy * a.istride; px = qx * a.istride;
      byte = (unsigned)(((n * a.Hi + py) * a.Wi + px) * a.x_pitch * 2 + c * 16);
    }
    p_byte[i] = byte; p_y[i] = py; p_x[i] = px;
  }

  int nk = pKtot / BK;
  int tap = 0, c0 = 0;                                // block-uniform K position
  if (a.psplit > 1) {                                 // this workgroup's K slice (blockIdx.z; never combined with phases)
    const int kt0 = (int)blockIdx.z * a.psteps;
    const int kt1 = kt0 + a.psteps < nk ? kt0 + a.psteps : nk;
    const int k0 = kt0 * BK;
    tap = k0 / a.Cin; c0 = k0 - tap * a.Cin;
    nk = kt1 - kt0;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;

  auto issue = [&](int buf) {
    const int ti = ptap[tap];
    const int dy = (int)(signed char)(ti & 0xff), dx = (int)(signed char)((ti >> 8) & 0xff), wt = ti >> 16;
    const int toff = ((dy * a.Wi + dx) * a.x_pitch + c0) * 2;
    const int woff = (wt * a.Cin + c0) * 2;
    char* base = smem + buf * STAGE + wave * (8 * RS);
#pragma unroll
    for (int i = 0; i < NIW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(base + i * (32 * RS)), 16,
                                               (int)(w_byte[i] == OOB ? OOB : w_byte[i] + (unsigned)woff), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI - NIW; ++i) {
      bool ok = (unsigned)(p_y[i] + dy) < (unsigned)a.Hi && (unsigned)(p_x[i] + dx) < (unsigned)a.Wi;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(base + (i + NIW) * (32 * RS)), 16,
                                               (int)(ok ? p_byte[i] + (unsigned)toff : OOB), 0, 0, 0);
    }
    c0 += BK;
    if (c0 >= a.Cin) { c0 = 0; ++tap; }
  };

  f32x16 acc[TCO][TPIX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPIX; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wco0 = (wave / WPIX) * (TCO * 32);
  const int wpix0 = (wave % WPIX) * (TPIX * 32);
  const int sw = (r >> 1) & 7;                        // read-side swizzle (tile bases are multiples of 16 rows)

  if (nk > 0) issue(0);
  __syncthreads();                                    // hipcc drains vmcnt before the barrier (LDS-DMA in flight)
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk && S2P_DIAGV(a) != 1) issue((kt + 1) & 1);
    const char* base = smem + (kt & 1) * STAGE;
    const char* wrow = base + (wco0 + r) * RS;
    const char* prow = base + (BCO + wpix0 + r) * RS;
    if (S2P_DIAGV(a) != 2)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = ((2 * s + h) ^ sw) * 16;
      bf16x8 af[TCO], bf[TPIX];
#pragma unroll
      for (int i = 0; i < TCO; ++i) af[i] = *(const bf16x8*)(wrow + i * 32 * RS + ch);
#pragma unroll
      for (int j = 0; j < TPIX; ++j) bf[j] = *(const bf16x8*)(prow + j * 32 * RS + ch);
#pragma unroll
      for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TPIX; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (a.psplit > 1) {
    // partial tile, [co][m]: for a fixed register the 32 lanes of a half store 32 consecutive pixels
    float* P = a.part + (size_t)blockIdx.z * a.Cst * a.part_m;
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < TPIX; ++j) {
        const int m = pix_base + wpix0 + 32 * j + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = co_base + wco0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (co < a.Cst) P[(size_t)co * a.part_m + m] = acc[i][j][e];
        }
      }
    return;
  }
  conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
}

// y[m][co] = epilogue(bias[co] + sum_z part[z][co][m]) for the K-split launches of conv_dma_kernel: the same bias /
// activation / residual / producer-activation-gradient semantics as conv_epilogue, element by element.  One thread per
// (pixel, 8-channel chunk), pixels fastest: the partial reads are coalesced, the 16-B stores land in L2.
__global__ __launch_bounds__(64) void conv_part_reduce_kernel(const GatherArgs a) {
  typedef __bf16 T;
  // one thread per (4 consecutive pixels, 8-channel chunk), pixel groups fastest: 16-B partial loads, coalesced.  The launch is
  // small (e.g. 25 k threads) and latency-bound: one-wave workgroups spread it over every CU, and the slices are loaded four at
  // a time (32 loads in flight per thread) but still added in slice order
  const int m4n = a.part_m >> 2;                         // part_m is a multiple of 128
  const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
  const int nch = (a.Cst + 7) / 8;
  if (idx >= (long long)m4n * nch) return;
  const int ch = (int)(idx / m4n), m0 = (int)(idx - (long long)ch * m4n) * 4;
  if (m0 >= a.M) return;
  const int co0 = ch * 8;
  f32x4 v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float bv = (a.bias && co0 + e < a.Cout) ? a.bias[co0 + e] : 0.f;
    v[e] = (f32x4){bv, bv, bv, bv};
  }
  const size_t zs = (size_t)a.Cst * a.part_m;
  const float* P0 = a.part + (size_t)co0 * a.part_m + m0;
  int z = 0;
  for (; z + 4 <= a.psplit; z += 4) {
    f32x4 t[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        t[u][e] = (co0 + e < a.Cst) ? *(const f32x4*)(P0 + (z + u) * zs + (size_t)e * a.part_m) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += t[u][e];
  }
  for (; z < a.psplit; ++z) {
    const float* P = P0 + z * zs;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (co0 + e < a.Cst) v[e] += *(const f32x4*)(P + (size_t)e * a.part_m);
  }
  const float ns = a.act == S2P_ACT_RELU ? 0.f : (a.act == S2P_ACT_LRELU ? a.slope : 1.f);
  const bool act_generic = (a.act == S2P_ACT_TANH || a.act == S2P_ACT_SWISH);
  const bool full = co0 + 8 <= a.Cst;
  const bool epi_add = a.epi == S2P_EPI_ADD;
  const bool g_tanh = a.gact == S2P_ACT_TANH;
  const float gneg = a.gact == S2P_ACT_RELU ? 0.f : (a.gact == S2P_ACT_LRELU ? a.gslope : 1.f);
  T* y = (T*)a.y;
  const T* aux = (const T*)a.aux; const T* aux2 = (const T*)a.aux2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + i;
    if (m >= a.M) break;
    Chunk<T> c;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = v[e][i];
      float w = lrelu_ns(t, ns);
      if (act_generic) w = a.act == S2P_ACT_TANH ? tanhf(t) : t / (1.f + expf(-t));      // (a uniform branch: as a select, tanhf AND expf ran for every element)
      wv[e] = w;
    }
    c.pack(wv);                                          // rounded to bf16 here, as the fused epilogue does before epi
    const size_t go = (size_t)m * a.y_pitch + co0;        // same grid, stride 1: output pixel index == GEMM pixel index
    if (a.epi != S2P_EPI_STORE) {
      Chunk<T> x, x2;
      x.raw = (u32x4){0u, 0u, 0u, 0u}; x2.raw = (u32x4){0u, 0u, 0u, 0u};
      if (full) {
        x.raw = *(const u32x4*)(aux + go);
        if (aux2) x2.raw = *(const u32x4*)(aux2 + go);
      } else {
        for (int e = 0; e < 8; ++e) if (co0 + e < a.Cst) {
          x.set(e, to_f32(aux[go + e]));
          if (aux2) x2.set(e, to_f32(aux2[go + e]));
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xv = x.get(e);
        const float f = g_tanh ? 1.f - xv * xv : (xv > 0.f ? 1.f : gneg);
        wv[e] = epi_add ? c.get(e) + xv : (c.get(e) + x2.get(e)) * f;
      }
      c.pack(wv);
    }
    if (full) *(u32x4*)(y + go) = c.raw;
    else for (int e = 0; e < 8; ++e) if (co0 + e < a.Cst) y[go + e] = from_f32<T>(c.get(e));
  }
}

// K-split plan of a single-phase bf16 launch (1 = no split): only launches of <= 160 workgroups, >= 16 K steps per slice
static int conv_split_plan(int nwg, int nk) {
  static const int min_steps = s2p_env_int("S2P_SPLIT_MIN_STEPS", 16);      // A/B switch (diagnostics build only)
  static const int max_wg = s2p_env_int("S2P_SPLIT_MAX_WG", 160);
  if (nwg > max_wg || nk < 2 * min_steps) return 1;
  int S = 384 / nwg;
  if (S > nk / min_steps) S = nk / min_steps;
  if (S > 16) S = 16;
  return S < 2 ? 1 : S;
}

// ------------------------------------------------------------------------------------------------
// Halo-resident variant for stride-1 "same" convolutions (output grid == gathered grid: ResBlk / VGG / gamma-beta
// convs and their dgrads).  In the implicit GEMM every tap re-fetches a shifted copy of the same pixels from L2
// (9x for 3x3) and the kernel is bound by L2->LDS bytes.  Here, for each 64-channel slab, the pixel tile PLUS its
// halo (the contiguous pixel range [tile_start - halo_lo, tile_end + halo_hi) in raster order, which also covers
// neighbouring rows and images) is staged ONCE and stays resident in LDS while all taps are swept; only the 16 KiB
// weight stage is streamed per tap.  Per 3x3 slab: 22 + 9*16 KiB instead of 9*32 KiB (-42 % L2 traffic).
// Tap validity (zero padding, image/row borders) is a per-pixel bit mask; an invalid (pixel, tap) reads a zero row.
template <int NPOS_CAP, bool DBUF, int TS = 0, bool PIPE = false>   // TS: static tap count (9 = 3x3, taps unrolled) or 0 = run-time taps
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const GatherArgs a) {
  typedef __bf16 T;
  constexpr int BCO = 128, BPIX = 128, WPIX = 2, TCO = 2, TPIX = 2;
  constexpr int BK = 64, RS = 128;
  constexpr int WSTAGE = BCO * RS;                     // 16 KiB weight stage
  constexpr int HALO = (NPOS_CAP + 1) * RS;            // + one all-zero row per halo buffer (target of invalid taps, TS variant)
  constexpr int ERS = BCO * 2 + 16;
  constexpr int EPI = BPIX * ERS;
  constexpr int NHB = DBUF ? 2 : 1;                    // halo buffers (double-buffered when two workgroups still fit a CU)
  constexpr int MAIN0 = NHB * HALO + 2 * WSTAGE + 1024;      // + zero rows
  constexpr int MAIN = MAIN0 > EPI ? MAIN0 : EPI;
  __shared__ __attribute__((aligned(1024))) char smem[MAIN + BPIX * 4];
  int* rowoff = (int*)(smem + MAIN);
  char* hbase = smem;
  char* wbase = smem + NHB * HALO;
  char* zrow = smem + NHB * HALO + 2 * WSTAGE;
  // diagnostics build, S2P_DIAG=9: the launch runs normally and stamps s_memrealtime (100 MHz) at entry / loop start / loop
  // end / after the epilogue into `aux` (a debug buffer of its own: 4 x u64 per workgroup; epi must be STORE)
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
  if (S2P_DIAGV(a) == 9) tl0 = __builtin_amdgcn_s_memrealtime();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int g = blockIdx.y;
  int nblk = a.npix_tiles * a.nco_tiles;
  int bid = blockIdx.x;
  {
    int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int co_tile = bid % a.nco_tiles, pix_tile = bid / a.nco_tiles;
  const int co_base = co_tile * BCO, pix_base = pix_tile * BPIX;
  const int QQ = a.Qh * a.Qw;
  if (tid < BPIX) {
    int m = pix_base + tid;
    rowoff[tid] = m < a.M ? m : -1;                   // same grid: output pixel index == GEMM pixel index
  }
  if (tid < 64) *(u32x4*)(zrow + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (tid < 8 * NHB) *(u32x4*)(hbase + (tid >> 3) * HALO + NPOS_CAP * RS + (tid & 7) * 16) = (u32x4){0u, 0u, 0u, 0u};

  const unsigned OOB = 0x80000000u;
  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  const int lrow = lane >> 3, pc = lane & 7;
  // weight DMA: instruction i of wave w stages rows (4i + w)*8 .. +7 of the 128-row stage
  unsigned w_byte[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (4 * i + wave) * 8 + lrow;
    int c = pc ^ ((row >> 1) & 7);
    int co = co_base + row;
    w_byte[i] = co < a.Cout ? (unsigned)(co * a.w_row * 2 + c * 16) : OOB;
  }
  // halo DMA: piece j (8 positions) for j = wave, wave + 4, ...; position p <-> pixel pix_base - halo_lo + p
  const int npos = BPIX + a.halo_lo + a.halo_hi;
  constexpr int NHP = (NPOS_CAP / 8 + 3) / 4;
  unsigned h_byte[NHP];
#pragma unroll
  for (int i = 0; i < NHP; ++i) {
    int pos = (4 * i + wave) * 8 + lrow;
    int c = pc ^ ((pos >> 1) & 7);
    long long gpix = (long long)pix_base - a.halo_lo + pos;
    h_byte[i] = (pos < npos && gpix >= 0 && gpix < a.M) ? (unsigned)(gpix * a.x_pitch * 2 + c * 16) : OOB;
  }
  // tap-validity masks of the pixels this lane feeds to the MFMA (B operand columns)
  const int wco0 = (wave / WPIX) * (TCO * 32);
  const int wpix0 = (wave % WPIX) * (TPIX * 32);
  unsigned long long vmask[TPIX];
  const double hrcpQQ = s2p_rcp_f64(QQ), hrcpQw = s2p_rcp_f64(a.Qw);
#pragma unroll
  for (int j = 0; j < TPIX; ++j) {
    int m = pix_base + wpix0 + 32 * j + r;
    unsigned long long mask = 0ull;
    if (m < a.M) {
      int rr, qx;
      const int n = divmod_rcp(m, QQ, hrcpQQ, rr), qy = divmod_rcp(rr, a.Qw, hrcpQw, qx);
      (void)n;
      for (int t0 = 0; t0 < a.T; t0 += 8) {            // 8 tap words per round: the scalar loads go out back to back
        int tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tv[u] = a.tap[(t0 + u) < MAX_TAPS ? t0 + u : MAX_TAPS - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          int iy = qy + (int)(signed char)(tv[u] & 0xff), ix = qx + (int)(signed char)((tv[u] >> 8) & 0xff);
          if (t0 + u < a.T && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) mask |= 1ull << (t0 + u);
        }
      }
    }
    vmask[j] = mask;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nslab = a.Cin / BK;
  const int nk = nslab * a.T;

  auto issue_w = [&](int buf, int tapword, int c0) {
    const int wt = tapword >> 16;
    const int woff = (wt * a.Cin + c0) * 2;
    char* base = wbase + buf * WSTAGE + wave * (8 * RS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(base + i * (32 * RS)), 16,
                                               (int)(w_byte[i] == OOB ? OOB : w_byte[i] + (unsigned)woff), 0, 0, 0);
  };
  auto issue_halo = [&](int hb, int c0) {
    char* base = hbase + hb * HALO + wave * (8 * RS);
#pragma unroll
    for (int i = 0; i < NHP; ++i)
      if ((4 * i + wave) * 8 < npos)                   // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(base + i * (32 * RS)), 16,
                                                 (int)(h_byte[i] == OOB ? OOB : h_byte[i] + (unsigned)(c0 * 2)), 0, 0, 0);
  };

  f32x16 acc[TCO][TPIX];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TPIX; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int swr = (r >> 1) & 7;                        // weight rows: tile bases are multiples of 16 rows

  // one linear pipeline over (slab, tap): weights are always one step ahead; the next slab's halo is prefetched
  // into the other halo buffer during the current slab (DBUF) or loaded behind a barrier at the slab boundary.
  if (S2P_DIAGV(a) == 5) return;                              // timing ablation: index set-up only
  if constexpr (TS > 0 && PIPE && DBUF) {
    // Software-pipelined static-tap form.  The weight stage is split into BK=32 HALF-stages (4 x 8 KiB in the same
    // 32 KiB as two full stages): the DMA of half-step h+3 is issued (inline asm, so the waits can be counted) while
    // half-step h computes, the fragments of half-step h+1 are read from LDS into a second register set under the MFMAs
    // of half-step h, and the only wait before the raw s_barrier is vmcnt(N) for the half-stage needed two steps later.
    constexpr int WHALF = BCO * 64;                     // 8 KiB: 128 rows x 64 B, chunk' = chunk ^ ((row >> 2) & 3)
    int pk[TS][TPIX];
    int wtv[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      const int ti = a.tap[t];
      wtv[t] = ti >> 16;
      const int toff = (int)(signed char)(ti & 0xff) * a.Wi + (int)(signed char)((ti >> 8) & 0xff);
#pragma unroll
      for (int j = 0; j < TPIX; ++j) {
        const int hp = wpix0 + 32 * j + r + a.halo_lo + toff;
        const bool ok = (vmask[j] >> t) & 1ull;
        pk[t][j] = ok ? hp * RS + ((hp >> 1) & 7) * 16 : NPOS_CAP * RS;
      }
    }
    const int ch16[4] = {(0 + h) * 16, (2 + h) * 16, (4 + h) * 16, (6 + h) * 16};
    const i32x4 wrs = s2p_make_rsrc(wg, a.w_bytes);
    const i32x4 xrs = s2p_make_rsrc(xg, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u);
    const unsigned wb_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(wbase));
    const unsigned hb_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(hbase));
    // weight half-stage DMA: piece p = 16 rows x 64 B; wave w issues pieces w and w + 4
    unsigned wv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 16 * (wave + 4 * q) + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      const int co = co_base + row;
      wv[q] = co < a.Cout ? (unsigned)(co * a.w_row * 2 + c * 16) : OOB;
    }
    auto issue_wh = [&](int hs, int wt, int kofs) {
      const unsigned dst = wb_lds + (unsigned)((hs & 3) * WHALF + wave * 1024);
      const unsigned woff = (unsigned)((wt * a.Cin + kofs) * 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        s2p_dma16(wrs, dst + q * 4096, (int)(wv[q] == OOB ? OOB : wv[q] + woff));
    };
    // halo DMA: every wave issues exactly NHP pieces (the last ones repeat its last real piece), so vmcnt counts are fixed
    constexpr int NHPc = (NPOS_CAP / 8 + 3) / 4;
    const int last_piece = (npos + 7) / 8 - 1;
    unsigned hv[NHPc]; int hpc[NHPc];
#pragma unroll
    for (int i = 0; i < NHPc; ++i) {
      int pc8 = 4 * i + wave; if (pc8 > last_piece) pc8 = last_piece - ((last_piece - wave) & 3);   // this wave's last real piece
      if (pc8 < 0) pc8 = wave <= last_piece ? wave : 0;
      hpc[i] = pc8;
      const int pos = pc8 * 8 + lrow;
      const int c = pc ^ ((pos >> 1) & 7);
      const long long gpix = (long long)pix_base - a.halo_lo + pos;
      hv[i] = (pos < npos && gpix >= 0 && gpix < a.M) ? (unsigned)(gpix * a.x_pitch * 2 + c * 16) : OOB;
    }
    auto issue_halo_p = [&](int hbuf, int c0) {
      const unsigned dst = hb_lds + (unsigned)(hbuf * HALO);
#pragma unroll
      for (int i = 0; i < NHPc; ++i)
        s2p_dma16(xrs, dst + (unsigned)(hpc[i] * 1024), (int)(hv[i] == OOB ? OOB : hv[i] + (unsigned)(c0 * 2)));
    };
    const int arow[TCO] = {(wco0 + r) * 64, (wco0 + 32 + r) * 64};
    const int asw = (r >> 2) & 3;
    bf16x8 FA[2][2][TCO], FB[2][2][TPIX];
    auto read_frags = [&](auto bufc, int hs, auto tc, auto halfc, const char* hbp) {
      constexpr int buf = decltype(bufc)::value, t = decltype(tc)::value, half = decltype(halfc)::value;
      const char* wbp = wbase + (hs & 3) * WHALF;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int i = 0; i < TCO; ++i) FA[buf][s2][i] = *(const bf16x8*)(wbp + arow[i] + (((2 * s2 + h) ^ asw) * 16));
#pragma unroll
        for (int j = 0; j < TPIX; ++j) FB[buf][s2][j] = *(const bf16x8*)(hbp + (pk[t][j] ^ ch16[2 * half + s2]));
      }
    };
    const int nhs = nslab * TS * 2;
    issue_halo_p(0, 0);
    issue_wh(0, wtv[0], 0);
    issue_wh(1, wtv[0], 32);
    if (nhs > 2) { issue_wh(2, wtv[1], 0); S2P_WAIT_VMCNT(2); } else { S2P_WAIT_VMCNT(0); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (S2P_DIAGV(a) == 3) return;
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (S2P_DIAGV(a) == 8) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    if (S2P_DIAGV(a) == 9) tl1 = __builtin_amdgcn_s_memrealtime();
    read_frags(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, hbase);
    int hs = 0;
    for (int slab = 0; slab < nslab; ++slab) {
      const char* hb = hbase + (slab & 1) * HALO;
      const char* hb_next = hbase + ((slab + 1) & 1) * HALO;
      const bool more_slabs = slab + 1 < nslab;
      static_for<0, 2 * TS>([&](auto uc) {               // u = 2 * tap + half (compile time: register sets and taps are static)
        constexpr int u = decltype(uc)::value;
        // (1) DMA of half-step hs + 3
        constexpr int u3 = u + 3;
        bool issued = false, halo_issued = false;
        if constexpr (u3 < 2 * TS) {
          issue_wh(hs + 3, wtv[u3 >> 1], slab * BK + (u3 & 1) * 32); issued = true;
        } else {
          if (more_slabs) { issue_wh(hs + 3, wtv[(u3 - 2 * TS) >> 1], (slab + 1) * BK + ((u3 - 2 * TS) & 1) * 32); issued = true; }
        }
        if constexpr (u == 0) {
          if (more_slabs) { issue_halo_p((slab + 1) & 1, (slab + 1) * BK); halo_issued = true; }
        }
        // (2) fragments of half-step hs + 1 into the other register set
        if constexpr (u + 1 < 2 * TS) {
          read_frags(std::integral_constant<int, (u + 1) & 1>{}, hs + 1, std::integral_constant<int, (u + 1) / 2>{},
                     std::integral_constant<int, (u + 1) & 1>{}, hb);
        } else {
          if (more_slabs)
            read_frags(std::integral_constant<int, 0>{}, hs + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, hb_next);
        }
        // (3) the MFMAs of half-step hs from the current register set
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPIX; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[u & 1][s2][i], FB[u & 1][s2][j], acc[i][j], 0, 0, 0);
        // (4) half-stage hs + 2 must have landed for every wave before the next step reads it
        if (issued) { if (halo_issued) S2P_WAIT_VMCNT(2 + NHPc); else S2P_WAIT_VMCNT(2); }
        else S2P_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ++hs;
      });
    }
    if (S2P_DIAGV(a) == 8) {
      const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (tid == 0 && g == 0) {
        unsigned long long* o = (unsigned long long*)a.y + (size_t)blockIdx.x * 2;
        o[0] = c1 - st_c0; o[1] = r1 - st_r0;
      }
      if (acc[0][0][0] == 12345.678f) ((float*)a.y)[7] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (S2P_DIAGV(a) == 4) { if (acc[0][0][0] == 12345.678f) ((float*)a.y)[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1]; return; }
    if (S2P_DIAGV(a) == 9) {
      tl2 = __builtin_amdgcn_s_memrealtime();
      GatherArgs b = a; b.aux = nullptr;
      conv_epilogue<T, BCO, BPIX, TCO, TPIX>(b, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && g == 0 && a.aux) {
        unsigned long long* o = (unsigned long long*)a.aux + (size_t)blockIdx.x * 4;
        o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = __builtin_amdgcn_s_memrealtime();
      }
      return;
    }
    conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
    return;
  } else if constexpr (TS > 0) {
    // Static-tap form (3x3): everything that depends only on (lane, tap) is computed ONCE -- the halo row a lane reads
    // for tap t (or the zero row when the tap falls outside the image), with the row's XOR swizzle folded into the low
    // bits, so a fragment address in the loop is  halo_base + (pk[t][j] ^ chunk_offset)  -- and the taps are unrolled:
    // no tap decode, validity test, address select or kernarg load per K step (they cost ~470 of ~1500 cycles a step).
    int pk[TS][TPIX];
    int wtv[TS];
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      const int ti = a.tap[t];
      wtv[t] = ti >> 16;
      const int toff = (int)(signed char)(ti & 0xff) * a.Wi + (int)(signed char)((ti >> 8) & 0xff);
#pragma unroll
      for (int j = 0; j < TPIX; ++j) {
        const int hp = wpix0 + 32 * j + r + a.halo_lo + toff;
        const bool ok = (vmask[j] >> t) & 1ull;
        pk[t][j] = ok ? hp * RS + ((hp >> 1) & 7) * 16 : NPOS_CAP * RS;
      }
    }
    const int ch16[4] = {(0 + h) * 16, (2 + h) * 16, (4 + h) * 16, (6 + h) * 16};
    issue_halo(0, 0);
    issue_w(0, wtv[0] << 16, 0);
    __syncthreads();
    if (S2P_DIAGV(a) == 3) return;
    unsigned long long st_c0 = 0, st_r0 = 0;
    if (S2P_DIAGV(a) == 8) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    int kt = 0;
    for (int slab = 0; slab < nslab; ++slab) {
      const char* hb = hbase + (DBUF ? (slab & 1) * HALO : 0);
      const bool more_slabs = slab + 1 < nslab;
#pragma unroll
      for (int t = 0; t < TS; ++t, ++kt) {
        if (S2P_DIAGV(a) != 1) {
          if (t + 1 < TS) issue_w((kt + 1) & 1, wtv[t + 1] << 16, slab * BK);
          else if (more_slabs) issue_w((kt + 1) & 1, wtv[0] << 16, (slab + 1) * BK);
          if (DBUF && t == 0 && more_slabs) issue_halo((slab + 1) & 1, (slab + 1) * BK);
        }
        const char* wrow = wbase + (kt & 1) * WSTAGE + (wco0 + r) * RS;
        if (S2P_DIAGV(a) != 2) {
          bf16x8 af[4][TCO], bf[4][TPIX];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int i = 0; i < TCO; ++i) af[s4][i] = *(const bf16x8*)(wrow + i * 32 * RS + (((2 * s4 + h) ^ swr) * 16));
#pragma unroll
            for (int j = 0; j < TPIX; ++j) bf[s4][j] = *(const bf16x8*)(hb + (pk[t][j] ^ ch16[s4]));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
              for (int j = 0; j < TPIX; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s4][i], bf[s4][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
      }
      if (!DBUF && more_slabs) {                        // single halo buffer: reload it now that nobody reads it
        issue_halo(0, (slab + 1) * BK);
        __syncthreads();
      }
    }
    if (S2P_DIAGV(a) == 8) {
      const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (tid == 0 && g == 0) {
        unsigned long long* o = (unsigned long long*)a.y + (size_t)blockIdx.x * 2;
        o[0] = c1 - st_c0; o[1] = r1 - st_r0;
      }
      if (acc[0][0][0] == 12345.678f) ((float*)a.y)[7] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
      return;
    }
    if (S2P_DIAGV(a) == 4) { if (acc[0][0][0] == 12345.678f) ((float*)a.y)[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1]; return; }
    conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
    return;
  }
  // tap words travel one step ahead of their use (a kernarg scalar load waited for on the spot stalls every step)
  int tw_cur = a.tap[0];
  int tw_next = a.tap[a.T > 1 ? 1 : 0];
  issue_halo(0, 0);
  issue_w(0, tw_cur, 0);
  __syncthreads();                                     // hipcc drains vmcnt before the barrier
  if (S2P_DIAGV(a) == 3) return;                              // timing ablation: prologue only
  int slab = 0, tap = 0;
  // S2P_DIAG=8 (diagnostic build of the launch, output invalid): stamp shader clock and 100 MHz wall clock around the
  // main loop; the host tool derives the in-kernel clock and cycles per K step (diagnostics build)
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (S2P_DIAGV(a) == 8) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int kt = 0; kt < nk; ++kt) {
    const int c0 = slab * BK;
    int ntap = tap + 1, nslab_i = slab;
    if (ntap == a.T) { ntap = 0; ++nslab_i; }
    int n2tap = ntap + 1;
    if (n2tap == a.T) n2tap = 0;
    const int tw_next2 = a.tap[n2tap];
    if (kt + 1 < nk && S2P_DIAGV(a) != 1) issue_w((kt + 1) & 1, tw_next, nslab_i * BK);
    if (DBUF && tap == 0 && slab + 1 < nslab && S2P_DIAGV(a) != 1) issue_halo((slab + 1) & 1, c0 + BK);
    const char* hb = hbase + (DBUF ? (slab & 1) * HALO : 0);
    const int ti = tw_cur;
    const int toff = (int)(signed char)(ti & 0xff) * a.Wi + (int)(signed char)((ti >> 8) & 0xff);
    const char* wrow = wbase + (kt & 1) * WSTAGE + (wco0 + r) * RS;
    const char* prow[TPIX];
    int psw[TPIX];
#pragma unroll
    for (int j = 0; j < TPIX; ++j) {
      const int hp = wpix0 + 32 * j + r + a.halo_lo + toff;
      const bool ok = (vmask[j] >> tap) & 1ull;
      prow[j] = ok ? hb + hp * RS : zrow;
      psw[j] = ok ? (hp >> 1) & 7 : 0;
    }
    if (S2P_DIAGV(a) != 2) {
      // all 16 fragment reads of the step are issued before its first MFMA (the waits become counted lgkmcnt(N)):
      // reading per sub-step exposes the LDS latency four times per step
      bf16x8 af[4][TCO], bf[4][TPIX];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TCO; ++i) af[s][i] = *(const bf16x8*)(wrow + i * 32 * RS + (((2 * s + h) ^ swr) * 16));
#pragma unroll
        for (int j = 0; j < TPIX; ++j) bf[s][j] = *(const bf16x8*)(prow[j] + (((2 * s + h) ^ psw[j]) * 16));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
          for (int j = 0; j < TPIX; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    tap = ntap;
    tw_cur = tw_next; tw_next = tw_next2;
    if (nslab_i != slab) {
      slab = nslab_i;
      if (!DBUF && slab < nslab) {                     // single halo buffer: reload it now that nobody reads it
        issue_halo(0, slab * BK);
        __syncthreads();
      }
    }
  }
  (void)nk;
  if (S2P_DIAGV(a) == 8) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && g == 0) {
      unsigned long long* o = (unsigned long long*)a.y + (size_t)blockIdx.x * 2;
      o[0] = c1 - st_c0; o[1] = r1 - st_r0;
    }
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[7] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
    return;
  }
  if (S2P_DIAGV(a) == 4) { if (acc[0][0][0] == 12345.678f) ((float*)a.y)[0] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1]; return; }
  conv_epilogue<T, BCO, BPIX, TCO, TPIX>(a, acc, smem, rowoff, g, co_base, wco0, wpix0, r, h, tid);
}

template <int BCO, int BPIX, int WCO, int WPIX>
static int launch_fast(GatherArgs& a, int groups, hipStream_t st) {
  a.npix_tiles = cdiv(a.M, BPIX);
  a.nco_tiles = cdiv(a.Cst, BCO);
  dim3 grid(a.npix_tiles * a.nco_tiles, groups);
  static const int no_dma = s2p_env_set("S2P_NO_LDS_DMA");
  static const int diag = s2p_env_int("S2P_DIAG", 0);
  a.diag = diag;
  static const int no_halo = s2p_env_set("S2P_NO_HALO");        // A/B switch: plain LDS-DMA kernel
  // launches that cannot fill the chip (<= 160 workgroups with a long K: PatchGAN 256->512 4x4 on 7x7 / 12x12 maps, VGG conv4_1 /
  // conv5_1 on 10x10 / 5x5 maps): K split over blockIdx.z + fixed-order reduce with the epilogue, when the caller passed a
  // scratch buffer.  One workgroup alone on a CU is bound by its LDS-DMA issue rate (~0.7 us per 32 KiB K step) whichever
  // kernel runs it, so the K loop is spread over the idle CUs instead -- this takes precedence over the halo-resident kernel.
  static const int no_split = s2p_env_set("S2P_NO_CONV_SPLITK");
  int S_plan = 1;
  if (!no_dma && !no_split && groups == 1 && a.nphase == 0 && a.ostride == 1 && a.oy0 == 0 && a.ox0 == 0 && a.Qh == a.Ho &&
      a.Qw == a.Wo && (a.plan || a.ws))
    S_plan = conv_split_plan((int)grid.x, a.Ktot / 64);
  // small planes (21x21 ResBlk / VGG conv3 / gamma-beta layers and their dgrads): one workgroup per (image, 64-channel slab),
  // the whole padded plane resident in LDS (conv_plane.hip)
  static const int no_plane = s2p_env_set("S2P_NO_PLANE");          // A/B switch (diagnostics build only)
  if (!no_plane && !no_dma && a.nphase == 0 && a.T == 9 && a.istride == 1 && a.ostride == 1 && a.oy0 == 0 && a.ox0 == 0 &&
      a.Qh == a.Hi && a.Qw == a.Wi && a.Ho == a.Qh && a.Wo == a.Qw && a.M % (a.Qh * a.Qw) == 0) {
    PlaneArgs p{};
    bool ok = true;
    for (int t = 0; t < 9; ++t) p.wt[t] = -1;
    for (int t = 0; t < 9 && ok; ++t) {
      const int dy = (int)(signed char)(a.tap[t] & 0xff), dx = (int)(signed char)((a.tap[t] >> 8) & 0xff);
      if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || p.wt[(dy + 1) * 3 + dx + 1] >= 0) ok = false;
      else p.wt[(dy + 1) * 3 + dx + 1] = a.tap[t] >> 16;
    }
    if (ok) {
      p.x = a.x; p.w = a.w; p.bias = a.bias; p.aux = a.aux; p.aux2 = a.aux2; p.y = a.y;
      p.N = a.M / (a.Qh * a.Qw); p.H = a.Qh; p.W = a.Qw;
      p.Cin = a.Cin; p.x_pitch = a.x_pitch; p.x_gstride = a.x_gstride;
      p.Cout = a.Cout; p.Cst = a.Cst; p.y_pitch = a.y_pitch; p.y_gstride = a.y_gstride;
      p.w_row = a.w_row; p.w_gstride = a.w_gstride;
      p.act = a.act; p.epi = a.epi; p.gact = a.gact; p.slope = a.slope; p.gslope = a.gslope;
      p.x_bytes = a.x_bytes; p.w_bytes = a.w_bytes;
      if (s2p_conv_plane_applicable(p)) {
        const bool fuse = a.mat && groups == 1 && a.act == S2P_ACT_NONE && a.epi != S2P_EPI_MUL_ACTGRAD && (!a.mat->xn || a.epi == S2P_EPI_STORE);
        if (fuse) *a.mat_done = 1;
        if (a.plan) return 0;                              // no scratch
        if (fuse) {
          p.xn = a.mat->xn; p.xn_pitch = a.mat->xn_pitch; p.dgb = a.mat->dgb; p.dgb_pitch = a.mat->dgb_pitch;
          p.dgbst = a.mat->dgbst; p.dgbst_pitch = a.mat->dgbst_pitch; p.res = a.mat->res; p.res_pitch = a.mat->res_pitch;
          p.y2 = a.mat->y2; p.y2_pitch = a.mat->y2_pitch; p.gb = a.mat->gb; p.gb_pitch = a.mat->gb_pitch;
          p.gbst = a.mat->gbst; p.gbst_pitch = a.mat->gbst_pitch; p.stats = a.mat->stats;
          p.n_act = a.mat->act; p.n_slope = a.mat->slope; p.eps = a.mat->eps;
        }
        return s2p_conv_plane_launch(p, groups, st);
      }
    }
  }
  // other small planes -- the PatchGAN 4x4 layers and their stride-1 dgrads (with the InstanceNorm that follows (forward) / precedes
  // (backward) the conv in the epilogue), VGG conv4_x on 10x10 maps: the generalised plane-resident kernel (conv_planeg.hip)
  if (!no_plane && !S2P_DIAG_SWITCH(0) && !(a.T == 9 && S2P_DIAG_SWITCH(2)) && !no_dma && groups == 1 && a.nphase == 0 && a.ostride == 1 && a.oy0 == 0 && a.ox0 == 0 && a.Qh == a.Ho &&
      a.Qw == a.Wo && a.M % (a.Qh * a.Qw) == 0 && !a.reflect) {
    PlaneGProblem pr{a.M / (a.Qh * a.Qw), a.Hi, a.Wi, a.Ho, a.Wo, a.Cin, a.Cout, a.Cst, a.x_pitch, a.y_pitch, a.istride, a.T, a.tap, a.mat != nullptr};
    PlaneGArgs p{};
    if (s2p_conv_planeg_setup(pr, p)) {
      const bool fuse = a.mat && p.nbands == 1 && p.gimg == 1 && !a.mat->gb && a.act == S2P_ACT_NONE && a.epi != S2P_EPI_MUL_ACTGRAD;
      if (fuse) *a.mat_done = 1;
      if (a.plan) return 0;                                // no scratch
      p.x = a.x; p.w = a.w; p.bias = a.bias; p.aux = a.aux; p.aux2 = a.aux2; p.y = a.y;
      p.N = pr.N; p.Cin = a.Cin; p.x_pitch = a.x_pitch; p.x_gstride = a.x_gstride;
      p.Cout = a.Cout; p.Cst = a.Cst; p.y_pitch = a.y_pitch; p.y_gstride = a.y_gstride;
      p.w_row = a.w_row; p.w_gstride = a.w_gstride;
      p.act = a.act; p.epi = a.epi; p.gact = a.gact; p.slope = a.slope; p.gslope = a.gslope;
      p.x_bytes = a.x_bytes; p.w_bytes = a.w_bytes;
      if (fuse) {
        p.xn = a.mat->xn; p.xn_pitch = a.mat->xn_pitch; p.dgb = a.mat->dgb; p.dgb_pitch = a.mat->dgb_pitch;
        p.dgbst = a.mat->dgbst; p.dgbst_pitch = a.mat->dgbst_pitch; p.res = a.mat->res; p.res_pitch = a.mat->res_pitch;
        p.y2 = a.mat->y2; p.y2_pitch = a.mat->y2_pitch; p.gbst = a.mat->gbst; p.gbst_pitch = a.mat->gbst_pitch; p.stats = a.mat->stats;
        p.n_act = a.mat->act; p.n_slope = a.mat->slope; p.eps = a.mat->eps;
      }
      return s2p_conv_planeg_launch(p, groups, st);
    }
  }
  if constexpr (BCO == 128 && BPIX == 128) {
    if (S_plan <= 1 && !no_dma && !no_halo && a.istride == 1 && a.ostride == 1 && a.Qh == a.Hi && a.Qw == a.Wi && a.Ho == a.Qh &&
        a.Wo == a.Qw && a.T >= 4) {
      if (a.plan) return 0;                              // halo-resident kernels: no scratch
      int lo = 0, hi = 0;
      for (int t = 0; t < a.T; ++t) {
        int off = (int)(signed char)(a.tap[t] & 0xff) * a.Wi + (int)(signed char)((a.tap[t] >> 8) & 0xff);
        if (-off > lo) lo = -off;
        if (off > hi) hi = off;
      }
      a.halo_lo = lo; a.halo_hi = hi;
      const int npos = BPIX + lo + hi;
      static const int extra_lds = s2p_env_int("S2P_HALO_EXTRA_LDS", 0);   // occupancy experiment
      static const int no_ts = s2p_env_set("S2P_NO_STATIC_TAPS");
      static const int pipe = (s2p_env_set("S2P_NO_HALO_PIPE") ? 0 : 1);         // A/B switch: software-pipelined variant (default on)
      if (a.T == 9 && !no_ts && pipe && npos <= 176 && a.Cin % 64 == 0) {
        hipLaunchKernelGGL((conv_halo_kernel<176, true, 9, true>), grid, dim3(256), extra_lds, st, a); S2P_CHECK_LAUNCH("conv_halo_kernel"); return 0;
      }
      if (a.T == 9 && !no_ts) {
        if (npos <= 176) { hipLaunchKernelGGL((conv_halo_kernel<176, true, 9>), grid, dim3(256), extra_lds, st, a); S2P_CHECK_LAUNCH("conv_halo_kernel"); return 0; }
        if (npos <= 320) { hipLaunchKernelGGL((conv_halo_kernel<320, false, 9>), grid, dim3(256), 0, st, a); S2P_CHECK_LAUNCH("conv_halo_kernel"); return 0; }
      }
      if (npos <= 176) { hipLaunchKernelGGL((conv_halo_kernel<176, true>), grid, dim3(256), extra_lds, st, a); S2P_CHECK_LAUNCH("conv_halo_kernel"); return 0; }
      if (npos <= 320) { hipLaunchKernelGGL((conv_halo_kernel<320, false>), grid, dim3(256), 0, st, a); S2P_CHECK_LAUNCH("conv_halo_kernel"); return 0; }
    }
  }
  if (S_plan > 1) {
    const int nk = a.Ktot / 64;
    {
      a.psteps = cdiv(nk, S_plan); a.psplit = cdiv(nk, a.psteps);
      a.part_m = a.npix_tiles * BPIX;
      const size_t need = (size_t)a.psplit * a.Cst * a.part_m * sizeof(float);
      if (a.plan) { if (need > *a.plan) *a.plan = need; return 0; }
      if (a.psplit > 1 && need <= a.ws_bytes) {
        a.part = (float*)a.ws;
        grid.z = a.psplit;
        hipLaunchKernelGGL((conv_dma_kernel<BCO, BPIX, WCO, WPIX>), grid, dim3(256), 0, st, a);
        S2P_CHECK_LAUNCH("conv_dma_kernel(split)");
        const long long n = (long long)(a.part_m / 4) * ((a.Cst + 7) / 8);
        hipLaunchKernelGGL(conv_part_reduce_kernel, dim3(cdiv(n, 64)), dim3(64), 0, st, a);
        S2P_CHECK_LAUNCH("conv_part_reduce_kernel");
        return 0;
      }
      a.psplit = 1; a.part = nullptr;
    }
  }
  if (a.plan) return 0;
  if (no_dma) hipLaunchKernelGGL((conv_fast_kernel<BCO, BPIX, WCO, WPIX>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_dma_kernel<BCO, BPIX, WCO, WPIX>), grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("conv_fast_kernel");
  return 0;
}

template <typename T>
static int launch_gather(GatherArgs& a, int groups, long long x_elems, hipStream_t st) {
  if (a.M <= 0) return 0;
  if constexpr (sizeof(T) == 2) {
    const long long xb = x_elems * 2, wb = (long long)a.Cout * a.w_row * 2;
    if (a.Cin % 64 == 0 && !a.reflect && a.T > 0 && xb < (1ll << 31) && wb < (1ll << 31) && a.Cst > 32) {
      a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
      if (a.Cst > 64) return launch_fast<128, 128, 2, 2>(a, groups, st);
      return launch_fast<64, 128, 2, 2>(a, groups, st);
    }
  }
  if (a.Cst > 64) return launch_cfg<T, 128, 128, 2, 2>(a, groups, st);
  if (a.Cst > 32) return launch_cfg<T, 64, 128, 2, 2>(a, groups, st);
  return launch_cfg<T, 32, 256, 1, 4>(a, groups, st);
}

static int pack_tap(int dy, int dx, int wt) { return (wt << 16) | ((dx & 0xff) << 8) | (dy & 0xff); }

// caller's scratch for K-split launches (ws may be null), or a dry run that only reports the bytes needed (plan)
struct Scratch { void* ws; size_t bytes; size_t* plan; const PlaneMat* mat; int* mat_done; };

// geometry of one generic problem: gathered tensor (Hi,Wi,Ci,xpitch,xg), produced tensor (Ho,Wo,Co,Cst,ypitch,yg)
struct Geo {
  int N, Hi, Wi, Ci, xp, xg, Ho, Wo, Co, Cst, yp, yg, KH, KW, stride, pad, reflect, groups;
  long long w_gstride; int w_row;
};

// Elements from the gathered tensor's base to the end of its last group (the buffer descriptors' extent; the kernels take
// g * xg off for group g).  Groups either share the pixel rows (xg < xp: channel slices, the extent is one tensor) or are whole
// tensors xg elements apart (group-major: [groups][N][H][W][xp]).
static inline long long gathered_elems(const Geo& G) {
  const long long one = (long long)G.N * G.Hi * G.Wi * G.xp;
  return (G.groups > 1 && G.xg >= G.xp) ? one + (long long)(G.groups - 1) * G.xg : one;
}

// "gather" orientation: out(oy) = sum_k in(oy*stride + k - pad)   (conv fwd, convT dgrad)
template <typename T>
static int run_gather(const Geo& G, const void* x, const void* w, const float* bias, const void* aux,
                      const void* aux2, void* y,
                      int act, float slope, int epi, int gact, float gslope, hipStream_t st, const Scratch& sc) {
  GatherArgs a{};
  a.ws = sc.ws; a.ws_bytes = sc.bytes; a.plan = sc.plan; a.mat = sc.mat; a.mat_done = sc.mat_done;
  a.x = x; a.w = w; a.bias = bias; a.aux = aux; a.aux2 = aux2; a.y = y;
  a.Hi = G.Hi; a.Wi = G.Wi; a.Qh = G.Ho; a.Qw = G.Wo; a.M = G.N * G.Ho * G.Wo;
  a.Cin = G.Ci; a.x_pitch = G.xp; a.x_gstride = G.xg;
  a.Cout = G.Co; a.Cst = G.Cst; a.y_pitch = G.yp; a.y_gstride = G.yg;
  a.Ho = G.Ho; a.Wo = G.Wo; a.istride = G.stride; a.ostride = 1; a.oy0 = 0; a.ox0 = 0;
  a.T = G.KH * G.KW; a.Ktot = a.T * G.Ci; a.w_row = G.w_row; a.w_gstride = G.w_gstride;
  a.reflect = G.reflect; a.act = act; a.epi = epi; a.slope = slope; a.gact = gact; a.gslope = gslope;
  if (a.T > MAX_TAPS) S2P_FAIL(-2, "conv: more than %d taps", MAX_TAPS);
  for (int ky = 0; ky < G.KH; ++ky)
    for (int kx = 0; kx < G.KW; ++kx) a.tap[ky * G.KW + kx] = pack_tap(ky - G.pad, kx - G.pad, ky * G.KW + kx);
  return launch_gather<T>(a, G.groups, gathered_elems(G), st);
}

// "scatter" orientation expressed per output phase: out(oy) = sum_k in((oy + pad - k)/stride)
// (conv_transpose fwd, strided/unstrided conv dgrad)
template <typename T>
static int run_scatter(const Geo& G, const void* x, const void* w, const float* bias, const void* aux,
                       const void* aux2, void* y,
                       int act, float slope, int epi, int gact, float gslope, hipStream_t st, const Scratch& sc) {
  const int s = G.stride;
  if constexpr (sizeof(T) == 2) {
    // all s*s phases in ONE launch of the LDS-DMA kernel (blockIdx.z = phase) when that kernel applies
    static const int no_merge = (s2p_env_set("S2P_NO_PHASE_MERGE") || s2p_env_set("S2P_NO_LDS_DMA"));
    const long long xb = gathered_elems(G) * 2, wb = (long long)G.Co * G.w_row * 2;
    if (!no_merge && s * s <= MAX_PHASES && s > 1 && G.Ci % 64 == 0 && xb < (1ll << 31) && wb < (1ll << 31) && G.Cst > 32) {
      GatherArgs a{};
      a.x = x; a.w = w; a.bias = bias; a.aux = aux; a.aux2 = aux2; a.y = y;
      a.Hi = G.Hi; a.Wi = G.Wi;
      a.Cin = G.Ci; a.x_pitch = G.xp; a.x_gstride = G.xg;
      a.Cout = G.Co; a.Cst = G.Cst; a.y_pitch = G.yp; a.y_gstride = G.yg;
      a.Ho = G.Ho; a.Wo = G.Wo; a.istride = 1; a.ostride = s;
      a.w_row = G.w_row; a.w_gstride = G.w_gstride;
      a.reflect = 0; a.act = act; a.epi = epi; a.slope = slope; a.gact = gact; a.gslope = gslope;
      a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
      static const int diag = s2p_env_int("S2P_DIAG", 0);
      a.diag = diag;
      bool ok = true;
      int np = 0, max_npt = 0;
      for (int py = 0; py < s && ok; ++py)
        for (int px = 0; px < s && ok; ++px) {
          GatherArgs::Phase& P = a.ph[np];
          P.Qh = (G.Ho - py + s - 1) / s; P.Qw = (G.Wo - px + s - 1) / s;
          if (P.Qh <= 0 || P.Qw <= 0) continue;
          P.M = G.N * P.Qh * P.Qw; P.oy0 = py; P.ox0 = px;
          int t = 0;
          for (int ky = 0; ky < G.KH; ++ky) {
            if ((py + G.pad - ky) % s != 0) continue;
            for (int kx = 0; kx < G.KW; ++kx) {
              if ((px + G.pad - kx) % s != 0) continue;
              if (t >= PHASE_TAPS) { ok = false; break; }
              P.tap[t++] = pack_tap((py + G.pad - ky) / s, (px + G.pad - kx) / s, ky * G.KW + kx);
            }
            if (!ok) break;
          }
          if (t == 0) ok = false;                      // a phase without taps still has to store zeros / bias: generic path
          P.T = t; P.Ktot = t * G.Ci;
          P.npix_tiles = cdiv(P.M, 128);
          if (P.npix_tiles > max_npt) max_npt = P.npix_tiles;
          ++np;
        }
      if (ok && np > 0) {
        if (sc.plan) return 0;
        a.nphase = np;
        const int BCO = a.Cst > 64 ? 128 : 64;
        a.nco_tiles = cdiv(a.Cst, BCO);
        dim3 grid(max_npt * a.nco_tiles, G.groups, np);
        // phase-fastest 1-D grid (see conv_dma_kernel) for the 64-row weight tile: HBM reads of the decoder's 128 -> 64 transposed conv
        // 117 -> 30 MB (= its input once) at the same duration, the 64 -> 128 stride-2 dgrad 55 -> 33 MB and 6 % faster; the 128-row tile
        // keeps the phase in blockIdx.z (measured: 41 -> 48 us on the 128 -> 256 stride-2 dgrad with the phases interleaved)
        a.phase_fast = (BCO == 64 && !S2P_DIAG_SWITCH(10)) ? 1 : 0;
        if (a.phase_fast) grid = dim3(((max_npt * a.nco_tiles + 7) / 8) * 8 * np, G.groups, 1);
        if (BCO == 128) hipLaunchKernelGGL((conv_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_dma_kernel<64, 128, 2, 2>), grid, dim3(256), 0, st, a);
        S2P_CHECK_LAUNCH("conv_dma_kernel(phases)");
        return 0;
      }
    }
  }
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      GatherArgs a{};
      a.ws = sc.ws; a.ws_bytes = sc.bytes; a.plan = sc.plan; a.mat = sc.mat; a.mat_done = sc.mat_done;
      a.x = x; a.w = w; a.bias = bias; a.aux = aux; a.aux2 = aux2; a.y = y;
      a.Hi = G.Hi; a.Wi = G.Wi;
      a.Qh = (G.Ho - py + s - 1) / s; a.Qw = (G.Wo - px + s - 1) / s;
      if (a.Qh <= 0 || a.Qw <= 0) continue;
      a.M = G.N * a.Qh * a.Qw;
      a.Cin = G.Ci; a.x_pitch = G.xp; a.x_gstride = G.xg;
      a.Cout = G.Co; a.Cst = G.Cst; a.y_pitch = G.yp; a.y_gstride = G.yg;
      a.Ho = G.Ho; a.Wo = G.Wo; a.istride = 1; a.ostride = s; a.oy0 = py; a.ox0 = px;
      a.w_row = G.w_row; a.w_gstride = G.w_gstride;
      a.reflect = 0; a.act = act; a.epi = epi; a.slope = slope; a.gact = gact; a.gslope = gslope;
      int t = 0;
      for (int ky = 0; ky < G.KH; ++ky) {
        if ((py + G.pad - ky) % s != 0) continue;
        for (int kx = 0; kx < G.KW; ++kx) {
          if ((px + G.pad - kx) % s != 0) continue;
          if (t >= MAX_TAPS) S2P_FAIL(-2, "conv: more than %d taps", MAX_TAPS);
          a.tap[t++] = pack_tap((py + G.pad - ky) / s, (px + G.pad - kx) / s, ky * G.KW + kx);
        }
      }
      a.T = t; a.Ktot = t * G.Ci;
      int rc = launch_gather<T>(a, G.groups, gathered_elems(G), st);
      if (rc) return rc;
    }
  return 0;
}

static int check_desc(const s2p_conv_desc* d, const char* who) {
  if (!d) S2P_FAIL(-1, "%s: null desc", who);
  if (d->dtype != S2P_F32 && d->dtype != S2P_BF16) S2P_FAIL(-1, "%s: bad dtype %d", who, d->dtype);
  int ce = d->dtype == S2P_F32 ? 4 : 8;
  if (d->Cin % ce || d->x_pitch % ce || d->y_pitch % ce || d->x_gstride % ce || d->y_gstride % ce)
    S2P_FAIL(-1, "%s: Cin/pitches must be multiples of %d elements (Cin=%d xp=%d yp=%d)", who, ce, d->Cin,
             d->x_pitch, d->y_pitch);
  if (d->groups < 1 || d->stride < 1 || d->KH < 1 || d->KW < 1) S2P_FAIL(-1, "%s: bad geometry", who);
  if (d->x_pitch < d->Cin || d->y_pitch < d->Cout) S2P_FAIL(-1, "%s: pitch smaller than channels", who);
  if ((long long)d->N * d->Ho * d->Wo >= (1ll << 31) / 512 * 64) { /* offsets are 32-bit pixel indices */ }
  if (d->reflect && (d->pad >= d->H || d->pad >= d->W)) S2P_FAIL(-1, "%s: reflect pad >= size", who);
  return 0;
}

static int conv_fwd_impl(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias, const void* aux,
                         void* y, int act, float slope, int epi, const Scratch& sc, void* stream) {
  int rc = check_desc(d, "s2p_conv2d_fwd");
  if (rc) return rc;
  S2P_CHECK_SLOPE("s2p_conv2d_fwd", act, slope);
  if (sc.mat) S2P_CHECK_SLOPE("s2p_conv2d_fwd_mat", sc.mat->act, sc.mat->slope);
  if (!sc.plan) {
    if (!x || !w_fwd || (!y && !sc.mat)) S2P_FAIL(-1, "s2p_conv2d_fwd: null pointer");      // (y == NULL: s2p_conv2d_fwd_mat checked that the launch is fused)
    if (epi != S2P_EPI_STORE && !aux) S2P_FAIL(-1, "s2p_conv2d_fwd: epi needs aux");
  }
  hipStream_t st = (hipStream_t)stream;
  int ce = d->dtype == S2P_F32 ? 4 : 8;
  if (epi == S2P_EPI_STORE && s2p_thin_applicable(d)) return sc.plan ? 0 : s2p_thin_fwd(d, x, w_fwd, bias, y, act, slope, st);
  if (!S2P_DIAG_SWITCH(3) && s2p_thin4_fwd_applicable(d, act, epi)) {      // 7x7, <= 4 real input channels (the generator's stem)
    if (sc.plan) { const size_t need = s2p_thin4_fwd_ws_bytes(d); if (need > *sc.plan) *sc.plan = need; return 0; }
    if (sc.ws && sc.bytes >= s2p_thin4_fwd_ws_bytes(d)) return s2p_thin4_fwd(d, x, w_fwd, bias, y, act, slope, sc.ws, sc.bytes, st);
  }
  static const int no_cin = s2p_env_set("S2P_NO_THIN_CIN");          // A/B switch (diagnostics build only)
  if (!no_cin && s2p_thin_cin_fwd_applicable(d, act, epi)) return sc.plan ? 0 : s2p_thin_cin_fwd(d, x, w_fwd, bias, y, act, slope, st);
  Geo G{d->N, d->H, d->W, d->Cin, d->x_pitch, d->x_gstride, d->Ho, d->Wo, d->Cout,
        /*Cst*/ d->groups == 1 ? ((d->Cout + ce - 1) / ce * ce <= d->y_pitch ? (d->Cout + ce - 1) / ce * ce : d->Cout)
                               : d->Cout,
        d->y_pitch, d->y_gstride, d->KH, d->KW, d->stride, d->pad, d->reflect, d->groups,
        (long long)d->Cout * d->KH * d->KW * d->Cin, d->KH * d->KW * d->Cin};
  if (d->transposed) {
    if (d->reflect) S2P_FAIL(-1, "s2p_conv2d_fwd: reflect + transposed unsupported");
    return d->dtype == S2P_F32 ? run_scatter<float>(G, x, w_fwd, bias, aux, nullptr, y, act, slope, epi, 0, 0.f, st, sc)
                               : run_scatter<__bf16>(G, x, w_fwd, bias, aux, nullptr, y, act, slope, epi, 0, 0.f, st, sc);
  }
  return d->dtype == S2P_F32 ? run_gather<float>(G, x, w_fwd, bias, aux, nullptr, y, act, slope, epi, 0, 0.f, st, sc)
                             : run_gather<__bf16>(G, x, w_fwd, bias, aux, nullptr, y, act, slope, epi, 0, 0.f, st, sc);
}

extern "C" int s2p_conv2d_fwd(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                              const void* aux, void* y, int act, float slope, int epi, void* stream) {
  return conv_fwd_impl(d, x, w_fwd, bias, aux, y, act, slope, epi, Scratch{nullptr, 0, nullptr}, stream);
}
extern "C" int s2p_conv2d_fwd_ws(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                                 const void* aux, void* y, int act, float slope, int epi, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return conv_fwd_impl(d, x, w_fwd, bias, aux, y, act, slope, epi, Scratch{workspace, workspace_bytes, nullptr}, stream);
}
extern "C" int s2p_conv2d_fwd_mat(const s2p_conv_desc* d, const void* x, const void* w_fwd, const float* bias, const void* aux,
                                  void* y, int epi, const void* gb_img, int gb_pitch, const float* gb_st, int gb_st_pitch,
                                  int act, float slope, float eps, void* y_mat, int y_mat_pitch, float* stats, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!d || !y_mat || !stats) S2P_FAIL(-1, "s2p_conv2d_fwd_mat: null pointer");
  // y == NULL: the conv output itself is not wanted (a forward pass without a backward: it is written only for the backward's sake) --
  // allowed where conv and norm are ONE launch, which then skips that store
  if (!y && !s2p_conv2d_mat_is_fused(d, 0, gb_img != nullptr))
    S2P_FAIL(-1, "s2p_conv2d_fwd_mat: y == NULL needs the fused launch (s2p_conv2d_mat_is_fused)");
  if (act != S2P_ACT_NONE && act != S2P_ACT_RELU && act != S2P_ACT_LRELU) S2P_FAIL(-1, "s2p_conv2d_fwd_mat: activation must be none / relu / lrelu");
  if (d->groups != 1 || d->transposed) S2P_FAIL(-1, "s2p_conv2d_fwd_mat: groups == 1, not transposed");
  PlaneMat m{y_mat, y_mat_pitch, gb_img, gb_pitch, gb_st, gb_st_pitch, stats, act, slope, eps, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0};
  int done = 0;
  int rc = conv_fwd_impl(d, x, w_fwd, bias, aux, y, S2P_ACT_NONE, 0.f, epi, Scratch{workspace, workspace_bytes, nullptr, &m, &done}, stream);
  if (rc || done) return rc;
  if (!y) S2P_FAIL(-1, "s2p_conv2d_fwd_mat: y == NULL but the launch was not fused");
  // shapes the plane-resident kernel does not take: the conv above + the norm as its own launch(es)
  return s2p_in_norm_fwd(d->dtype, y, d->N, d->Ho * d->Wo, d->Cout, d->y_pitch, gb_img, gb_pitch, gb_st, gb_st_pitch, act, slope,
                         eps, y_mat, y_mat_pitch, stats, stream);
}
extern "C" size_t s2p_conv2d_fwd_workspace(const s2p_conv_desc* d, int epi) {
  size_t need = 0;
  if (conv_fwd_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, S2P_ACT_NONE, 0.f, epi, Scratch{nullptr, 0, &need}, nullptr)) return 0;
  return need;
}

// dgrad: gathered tensor = dy (grid Ho x Wo, channels Cout), produced tensor = dx (grid H x W, channels Cin).
// With reflect padding the produced grid is the PADDED one, (H+2p) x (W+2p): fold it with s2p_reflect_pad_bwd.
static int conv_dgrad_impl(const s2p_conv_desc* d, const void* dy, const void* w_bwd, const void* aux, const void* aux2,
                           void* dx, int epi, int aux_act, float slope, const Scratch& sc, void* stream) {
  int rc = check_desc(d, "s2p_conv2d_dgrad");
  if (rc) return rc;
  if (!sc.plan) {
    if (!dy || !w_bwd || !dx) S2P_FAIL(-1, "s2p_conv2d_dgrad: null pointer");
    if (epi != S2P_EPI_STORE && !aux) S2P_FAIL(-1, "s2p_conv2d_dgrad: epi needs aux");
  }
  hipStream_t st = (hipStream_t)stream;
  int ce = d->dtype == S2P_F32 ? 4 : 8;
  int cout_pad = (d->Cout + ce - 1) / ce * ce;       // channels of dy actually gathered
  if (cout_pad > d->y_pitch) S2P_FAIL(-1, "s2p_conv2d_dgrad: dy pitch %d < padded Cout %d", d->y_pitch, cout_pad);
  if (!S2P_DIAG_SWITCH(3) && epi == S2P_EPI_STORE && s2p_thin4_dgrad_applicable(d, cout_pad)) {      // 7x7, <= 4 output channels (the generator's output conv)
    if (sc.plan) { const size_t need = s2p_thin4_dgrad_ws_bytes(d); if (need > *sc.plan) *sc.plan = need; return 0; }
    if (sc.ws && sc.bytes >= s2p_thin4_dgrad_ws_bytes(d)) return s2p_thin4_dgrad(d, dy, w_bwd, dx, cout_pad, sc.ws, sc.bytes, st);
  }
  // thin input (<= 8 channels), stride 1: the adjoint is a thin-Cout conv over dy (row-streaming kernel, thin_rows.hip)
  if (epi == S2P_EPI_STORE && s2p_thin_rows_dgrad_applicable(d, cout_pad))
    return sc.plan ? 0 : s2p_thin_rows_dgrad(d, dy, w_bwd, dx, cout_pad, st);
  int H = d->H, W = d->W, pad = d->pad;
  if (d->reflect) { H += 2 * pad; W += 2 * pad; pad = 0; }
  Geo G{d->N, d->Ho, d->Wo, cout_pad, d->y_pitch, d->y_gstride, H, W, d->Cin, d->Cin,
        d->x_pitch, d->x_gstride, d->KH, d->KW, d->stride, pad, 0, d->groups,
        (long long)d->Cin * d->KH * d->KW * cout_pad, d->KH * d->KW * cout_pad};
  if (d->transposed)   // adjoint of a scatter is a gather
    return d->dtype == S2P_F32 ? run_gather<float>(G, dy, w_bwd, nullptr, aux, aux2, dx, S2P_ACT_NONE, 0.f, epi, aux_act, slope, st, sc)
                               : run_gather<__bf16>(G, dy, w_bwd, nullptr, aux, aux2, dx, S2P_ACT_NONE, 0.f, epi, aux_act, slope, st, sc);
  return d->dtype == S2P_F32 ? run_scatter<float>(G, dy, w_bwd, nullptr, aux, aux2, dx, S2P_ACT_NONE, 0.f, epi, aux_act, slope, st, sc)
                             : run_scatter<__bf16>(G, dy, w_bwd, nullptr, aux, aux2, dx, S2P_ACT_NONE, 0.f, epi, aux_act, slope, st, sc);
}

extern "C" int s2p_conv2d_dgrad(const s2p_conv_desc* d, const void* dy, const void* w_bwd, const void* aux,
                                const void* aux2, void* dx, int epi, int aux_act, float slope, void* stream) {
  return conv_dgrad_impl(d, dy, w_bwd, aux, aux2, dx, epi, aux_act, slope, Scratch{nullptr, 0, nullptr}, stream);
}
extern "C" int s2p_conv2d_dgrad_ws(const s2p_conv_desc* d, const void* dy, const void* w_bwd, const void* aux,
                                   const void* aux2, void* dx, int epi, int aux_act, float slope, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return conv_dgrad_impl(d, dy, w_bwd, aux, aux2, dx, epi, aux_act, slope, Scratch{workspace, workspace_bytes, nullptr}, stream);
}
extern "C" int s2p_conv2d_dgrad_mat(const s2p_conv_desc* d, const void* dy, const void* w_bwd, void* d_mid, const void* aux, const void* xn,
                                    int xn_pitch, const float* stats, const void* gb_img, int gb_pitch, const float* gb_st,
                                    int gb_st_pitch, int act, float slope, float eps, float* sums, void* dxn, int dxn_pitch,
                                    void* dgb_img, int dgb_pitch, float* dgb_st, int dgb_st_pitch, const void* res, int res_pitch,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !xn || !stats || !dxn) S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: null pointer");
  if (act != S2P_ACT_NONE && act != S2P_ACT_RELU && act != S2P_ACT_LRELU) S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: activation must be none / relu / lrelu");
  if (d->groups != 1 || d->transposed || d->reflect) S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: groups == 1, not transposed, zero padding");
  if (d->Cin != d->x_pitch) S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: the produced tensor must be dense (x_pitch %d != Cin %d)", d->x_pitch, d->Cin);
  PlaneMat m{dxn, dxn_pitch, gb_img, gb_pitch, gb_st, gb_st_pitch, const_cast<float*>(stats), act, slope, eps,
             xn, xn_pitch, dgb_img, dgb_pitch, dgb_st, dgb_st_pitch, res, res_pitch};
  if (!d_mid || !sums) {                  // scratch of the two-launch form left out: the caller relies on the fused kernel
    if (!s2p_conv2d_mat_is_fused(d, 1, gb_img != nullptr))
      S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: this shape runs as two launches and needs d_mid and sums (s2p_conv2d_mat_is_fused)");
  }
  int done = 0;
  int rc = conv_dgrad_impl(d, dy, w_bwd, aux, nullptr, d_mid ? d_mid : dxn, aux ? S2P_EPI_ADD : S2P_EPI_STORE, S2P_ACT_NONE, 0.f,
                           Scratch{workspace, workspace_bytes, nullptr, &m, &done}, stream);
  if (rc || done) return rc;
  if (!d_mid || !sums) S2P_FAIL(-1, "s2p_conv2d_dgrad_mat: not fused for these arguments (an aux gradient fuses only on the 4x4 family): d_mid and sums are required");
  // shapes the plane-resident kernels do not take: the dgrad above wrote d_mid; the norm backward as its own launch(es)
  return s2p_in_norm_bwd_res(d->dtype, d_mid, d->x_pitch, xn, d->N, d->H * d->W, d->Cin, xn_pitch, stats, gb_img, gb_pitch, gb_st,
                             gb_st_pitch, act, slope, eps, sums, dxn, dxn_pitch, dgb_img, dgb_pitch, dgb_st, dgb_st_pitch, res,
                             res_pitch, stream);
}
extern "C" int s2p_conv2d_mat_is_fused(const s2p_conv_desc* d, int dgrad, int has_gb) {
  if (!d || d->groups != 1 || d->transposed || d->reflect) return 0;
  static const char dummy[16] = {0};
  PlaneMat m{};
  m.y2 = (void*)dummy; m.gb = has_gb ? dummy : nullptr; m.xn = dgrad ? dummy : nullptr;
  int done = 0; size_t need = 0;
  const int rc = dgrad ? conv_dgrad_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, S2P_EPI_STORE, S2P_ACT_NONE, 0.f, Scratch{nullptr, 0, &need, &m, &done}, nullptr)
                       : conv_fwd_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, S2P_ACT_NONE, 0.f, S2P_EPI_STORE, Scratch{nullptr, 0, &need, &m, &done}, nullptr);
  return rc == 0 && done ? 1 : 0;
}
extern "C" size_t s2p_conv2d_dgrad_workspace(const s2p_conv_desc* d) {
  size_t need = 0;
  if (conv_dgrad_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, S2P_EPI_STORE, S2P_ACT_NONE, 0.f, Scratch{nullptr, 0, &need}, nullptr)) return 0;
  return need;
}
