// Padded-raster weight gradient of the PatchGAN 4x4 convolutions (round 4): stride 2 (64 -> 128, 128 -> 256 on both scales) and
// stride 1 / pad 2 (256 -> 512).
//
//   dW[a][t][b] += sum over positions (n, r, c) of the A grid:  A[n, r, c][a] * B[n, s*r + dy_t, s*c + dx_t][b]
//   (conv: A = dY, B = X;  transposed conv: A = X, B = dY -- the same sum, see wgrad_igemm.hip)
//
// Why: in the implicit GEMM (wgrad_dma_kernel) a 128 x 128 output tile is ONE tap x 128 channels, so every tap re-streams its own
// gathered copy of B and every tile re-streams A: the 90-GFLOP 256 -> 512 layer runs at 540 TFLOP/s.  wgrad_slab.hip removes that
// for stride-1 3x3 convs with a padded raster on which a tap is a row shift of ONE resident window.  This file is the same kernel
// with the three things the 4x4 layers need:
//   * STRIDE 2 as parity classes.  dy_t = 2*a_t + py_t: tap t reads the parity sub-plane (py, px) of B at the stride-1 shift
//     (a_t, b_t).  The taps of one parity form a TILE CLASS: its workgroups stage the window of that sub-plane (the space-to-depth
//     is done by the LDS-DMA's per-lane source address: pixel (2r + py, 2c + px)) and keep one accumulator tile per tap of the
//     class -- four classes of 4 taps for 4x4 stride 2.  The 16 taps of a 4x4 stride-1 conv are two classes of 8 (ky < 2, ky >= 2).
//   * A and B live on DIFFERENT grids (13x13 against 12x12 for the pad-2 layer; 22x22 against the parity planes of 43x43): the
//     common raster is Hp x Wp with Hp = max(Ha + amax, Hsub - amin) (Wp likewise), every shifted read that leaves B's grid lands on
//     a position where B's DMA returns zeros, every pad position of A contributes 0.
//   * The tap count of a workgroup is a run-time property of its class (the body is instantiated for 4 and 8 taps and selected by
//     a wave-uniform branch).  Partial tiles go to a slab with plain stores, a second kernel adds them in a fixed order: no atomics.
// Staging, swizzle, fragment reads (`ds_read_b64_tr_b16`), the 3-stage LDS-DMA pipeline with counted vmcnt: as in wgrad_slab.hip.
// The 3x3 stride-2 convs of the encoder / decoder (classes of 4 / 2 / 2 / 1 taps) were measured on this kernel too: 47-53 us against
// 45-48 us on the implicit GEMM -- a stride-2 gather touches a quarter of B per tap, so there the re-streaming is cheap and the
// short-tap classes are bound by the L2 -> LDS bytes of their windows; they stay on wgrad_dma_kernel.
#include "s2p_common.h"
#include <type_traits>

template <int B, int E, typename F>
__device__ __forceinline__ void sg_static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); sg_static_for<B + 1, E>(f); }
}

constexpr int SG_MAX_CLS = 4, SG_MAX_T = 8;
struct WgSlabGArgs {
  const void* A; const void* B; float* dW; float* db;
  float* slab; float* slabb;
  int N, Ha, Wa, Hb, Wb, Hp, Wp, bs;
  int a_pitch, b_pitch;
  int co_tiles, tiles_per_cls;                   // Ca / 64, (Ca / 64) * (Cb / 64)
  int dw_row, Cb;                                // floats per dW row (taps * Cb)
  int ncls;
  int cls_T[SG_MAX_CLS], cls_py[SG_MAX_CLS], cls_px[SG_MAX_CLS], cls_hneg[SG_MAX_CLS];
  int cls_S[SG_MAX_CLS], cls_bps[SG_MAX_CLS];    // K splits of the class, raster blocks per split
  int cls_wg0[SG_MAX_CLS + 1];                   // first workgroup of the class (cumulative)
  int cls_red0[SG_MAX_CLS + 1];                  // first workgroup of the class in the reduce launch (tiles_per_cls * T each)
  long long cls_slab0[SG_MAX_CLS];               // first float of the class's partial tiles in `slab`
  int toff[SG_MAX_CLS][SG_MAX_T];                // raster offset a_t * Wp + b_t
  int wt[SG_MAX_CLS][SG_MAX_T];                  // tap index in dW
  int nblocks, total_wgs;
  unsigned a_bytes, b_bytes;
  int tbl;                                       // DMA offsets from the tabulated padded raster (sg_body)
};

// TBL: DMA source offsets from a tabulated padded raster (one table per operand at LDS address 0: wgrad_slab.hip, round 5)
constexpr int SG_TBL_MAX = 768;
template <int T, int NXI, bool TBL>
__device__ __forceinline__ void sg_body(const WgSlabGArgs& a, char* smem, const int cls, const int tile, const int split) {
  constexpr int RS = 128, NST = 3;
  constexpr int ASTG = 64 * RS;                 // 8 KiB: 64 positions x 64 A channels
  constexpr int WROWS = NXI * 32;
  constexpr int XSTG = WROWS * RS;              // B window: 64 positions + the class's halo, 64 B channels
  constexpr int STG = ASTG + XSTG;
  constexpr int NDMA = 2 + NXI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co_t = tile % a.co_tiles, ci_s = tile / a.co_tiles;
  const int bps = a.cls_bps[cls];
  const int b0 = split * bps;
  int b1 = b0 + bps; if (b1 > a.nblocks) b1 = a.nblocks;
  const int nblk = b1 - b0;                                   // >= 1 by construction
  const int py = a.cls_py[cls], px = a.cls_px[cls], hneg = a.cls_hneg[cls];

  const unsigned OOB = 0x80000000u;
  const i32x4 ar = s2p_make_rsrc(a.A, a.a_bytes);
  const i32x4 br = s2p_make_rsrc(a.B, a.b_bytes);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(s2p_lds_addr(smem));

  // ---- DMA geometry: a piece = 8 rows x 128 B; every lane keeps the raster coordinates (n, r, c) of the NDMA rows it stages
  const int lrow = lane >> 3, pch = lane & 7;
  const int a_cbyte = (co_t * 64) * 2, b_cbyte = (ci_s * 64) * 2;
  int pn[NDMA], prr[NDMA], pc[NDMA];
  int cb[NDMA];
  // TBL: q4 = 4 * (position inside its image's padded raster), noffc = image offset + chunk offset (bytes), tv = the table entry of q
  unsigned q4[NDMA], noffc[NDMA], tv[NDMA];
  const int HpWp = a.Hp * a.Wp;
  unsigned* const ptab = (unsigned*)(smem - 2 * SG_TBL_MAX * 4);       // [A | B] in front of the stages (wgrad_slabg_kernel)
  if constexpr (TBL) {
    for (int q = tid; q < HpWp; q += 256) {
      const int r = q / a.Wp, c = q - r * a.Wp;
      ptab[q] = (c < a.Wa && r < a.Ha) ? (unsigned)(r * a.Wa + c) * (unsigned)(a.a_pitch * 2) : 0xc0000000u;
      const int sy = r * a.bs + py, sx = c * a.bs + px;          // the parity sub-plane's pixel in the full-resolution tensor
      ptab[SG_TBL_MAX + q] = (sy < a.Hb && sx < a.Wb) ? (unsigned)(sy * a.Wb + sx) * (unsigned)(a.b_pitch * 2) : 0xc0000000u;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int row = (4 * (i < 2 ? i : i - 2) + wave) * 8 + lrow;
    int pos = b0 * 64 + row - (i < 2 ? 0 : hneg);
    int nadj = 0;
    if (pos < 0) { pos += a.Hp * a.Wp; nadj = -1; }            // pos >= -hneg > -Hp*Wp
    cb[i] = (i < 2 ? a_cbyte : b_cbyte) + ((pch ^ (((row >> 1) & 1) << 2)) * 16);
    if constexpr (TBL) {
      const int n = pos / HpWp, q = pos - n * HpWp;
      q4[i] = (unsigned)q * 4u;
      noffc[i] = (unsigned)((n + nadj) * (i < 2 ? a.Ha * a.Wa * a.a_pitch * 2 : a.Hb * a.Wb * a.b_pitch * 2) + cb[i]);
      tv[i] = ptab[(i < 2 ? 0 : SG_TBL_MAX) + q];
    } else {
      const int q1 = pos / a.Wp;
      pc[i] = pos - q1 * a.Wp;
      const int n = q1 / a.Hp;
      prr[i] = q1 - n * a.Hp;
      pn[i] = n + nadj;
    }
  }
  const int a_pitch2 = a.a_pitch * 2, b_pitch2 = a.b_pitch * 2;
  const int adv_q = 64 / a.Wp, adv_c = 64 - adv_q * a.Wp, adv_n = adv_q / a.Hp, adv_r = adv_q - adv_n * a.Hp;
  const int tadv_n = 64 / HpWp;
  const unsigned tadv_q4 = (unsigned)(64 - tadv_n * HpWp) * 4u, HpWp4 = (unsigned)HpWp * 4u;
  const unsigned img_a = (unsigned)(a.Ha * a.Wa) * (unsigned)a_pitch2, img_b = (unsigned)(a.Hb * a.Wb) * (unsigned)b_pitch2;
  auto issue_one = [&](auto ic, unsigned base) {
    constexpr int i = decltype(ic)::value;
    if constexpr (TBL) {
      const unsigned toff = tv[i] + noffc[i];
      if (i < 2) s2p_dma16(ar, base + (4 * i + wave) * 1024, (int)toff);
      else s2p_dma16(br, base + ASTG + (4 * (i - 2) + wave) * 1024, (int)toff);
      const unsigned img = i < 2 ? img_a : img_b;
      const unsigned qa = q4[i] + tadv_q4;
      const bool wrap = qa >= HpWp4;
      q4[i] = wrap ? qa - HpWp4 : qa;
      noffc[i] += (unsigned)tadv_n * img + (wrap ? img : 0u);
      tv[i] = *(const unsigned*)((const char*)(ptab + (i < 2 ? 0 : SG_TBL_MAX)) + q4[i]);      // consumed a whole block later
      return;
    }
    int off;
    if constexpr (i < 2) {
      const bool ok = pc[i] < a.Wa && prr[i] < a.Ha && (unsigned)pn[i] < (unsigned)a.N;
      const int pix = __mul24(__mul24(pn[i], a.Ha) + prr[i], a.Wa) + pc[i];
      off = ok ? __mul24(pix, a_pitch2) + cb[i] : (int)OOB;
      s2p_dma16(ar, base + (4 * i + wave) * 1024, off);
    } else {
      const int sy = prr[i] * a.bs + py, sx = pc[i] * a.bs + px;     // the parity sub-plane's pixel in the full-resolution tensor
      const bool ok = sy < a.Hb && sx < a.Wb && (unsigned)pn[i] < (unsigned)a.N;
      const int pix = __mul24(__mul24(pn[i], a.Hb) + sy, a.Wb) + sx;
      off = ok ? __mul24(pix, b_pitch2) + cb[i] : (int)OOB;
      s2p_dma16(br, base + ASTG + (4 * (i - 2) + wave) * 1024, off);
    }
    int c = pc[i] + adv_c, r = prr[i] + adv_r, n = pn[i] + adv_n;
    const bool cw = c >= a.Wp;
    c = cw ? c - a.Wp : c; r += cw ? 1 : 0;
    const bool rw = r >= a.Hp;
    r = rw ? r - a.Hp : r; n += rw ? 1 : 0;
    pc[i] = c; prr[i] = r; pn[i] = n;
  };
  auto issue = [&](int stage) {
    const unsigned base = lds0 + stage * STG;
    sg_static_for<0, NDMA>([&](auto ic) { issue_one(ic, base); });
  };

  // ---- fragment geometry (ds_read_b64_tr_b16), as in wgrad_slab.hip
  const int gq = lane >> 4, gg = gq & 1, hh = gq >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int wa = wave >> 1, wb = wave & 1;                     // wave tile: A channels [32wa, +32) x B channels [32wb, +32), every tap
  const int a_lane = (8 * hh + q) * RS + (((4 * wa + 2 * gg + (p >> 1)) ^ ((q >> 1) << 2)) * 16) + 8 * (p & 1);
  int b_lane[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int r0 = hneg + a.toff[cls][t] + 8 * hh + q;         // window row of this lane for k = 0, half 0
    b_lane[t] = ASTG + r0 * RS + (((4 * wb + 2 * gg + (p >> 1)) ^ (((r0 >> 1) & 1) << 2)) * 16) + 8 * (p & 1);
  }

  f32x16 acc[T];
  f32x16 accb;
#pragma unroll
  for (int e = 0; e < 16; ++e) accb[e] = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const bool do_bias = a.db != nullptr && cls == 0 && ci_s == 0 && wb == 0;         // wave-uniform
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  issue(0);
  if (nblk > 1) { issue(1); S2P_WAIT_VMCNT(NDMA); } else { S2P_WAIT_VMCNT(0); }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  constexpr int NSTEP = 4 * T;
  constexpr int LA = NSTEP < 4 ? NSTEP : 4, RING = LA + 1;
  constexpr int TA = T - 1 - LA >= 0 ? T - 1 - LA : 0;         // tap step behind which the next substep's A fragment is read
  auto read_frag = [&](const char* ptr) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ptr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ptr + 4 * RS));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto main_loop = [&](auto biasc) {
    constexpr bool BIAS = decltype(biasc)::value;
    int stage = 0;
    for (int kb = 0; kb < nblk; ++kb) {
      int st2 = stage + 2; if (st2 >= NST) st2 -= NST;
      const bool more = kb + 2 < nblk;
      const unsigned dbase = lds0 + st2 * STG;
      const char* sb = smem + stage * STG;
      bf16x8 AF[2], BF[RING];
      AF[0] = read_frag(sb + a_lane);
      sg_static_for<0, LA>([&](auto vc) {
        constexpr int v = decltype(vc)::value;
        BF[v % RING] = read_frag(sb + b_lane[v % T] + (v / T) * 16 * RS);
      });
      sg_static_for<0, NSTEP>([&](auto uc) {
        constexpr int u = decltype(uc)::value, s_ = u / T, t = u % T;
        if constexpr (BIAS && t == 0) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[s_ & 1], ones, accb, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[s_ & 1], BF[u % RING], acc[t], 0, 0, 0);
        constexpr int v = u + LA;
        if constexpr (v < NSTEP) BF[v % RING] = read_frag(sb + b_lane[v % T] + (v / T) * 16 * RS);
        if constexpr (t == TA && s_ < 3) AF[(s_ + 1) & 1] = read_frag(sb + a_lane + (s_ + 1) * 16 * RS);
        // DMA i of block kb + 2 goes out behind step (i * NSTEP) / NDMA + 1 (the last step at the latest)
        sg_static_for<0, NDMA>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int at = (i * NSTEP) / NDMA + 1 < NSTEP ? (i * NSTEP) / NDMA + 1 : NSTEP - 1;
          if constexpr (u == at) { if (more) issue_one(ic, dbase); }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if (more) S2P_WAIT_VMCNT(NDMA); else S2P_WAIT_VMCNT(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (++stage == NST) stage = 0;
    }
  };
  if (do_bias) main_loop(std::integral_constant<bool, true>{});
  else main_loop(std::integral_constant<bool, false>{});

  // ---- epilogue: the partial tile [64 A rows][T][64 B channels] of this (tile, split) with plain stores
  const int r = lane & 31, h = lane >> 5;
  const int S = a.cls_S[cls];
  float* sl = a.slab + a.cls_slab0[cls] + ((size_t)tile * S + split) * (64 * T * 64);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h;
      sl[(row * T + t) * 64 + 32 * wb + r] = acc[t][e];
    }
  if (do_bias && r == 0) {
    float* sb2 = a.slabb + ((size_t)co_t * S + split) * 64;
#pragma unroll
    for (int e = 0; e < 16; ++e) sb2[32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h] = accb[e];
  }
}

template <int NXI>
__global__ __launch_bounds__(256, 2) void wgrad_slabg_kernel(const WgSlabGArgs a) {
  constexpr int STG = 64 * 128 + NXI * 32 * 128;
  constexpr int TOFF = 2 * SG_TBL_MAX * 4;                    // the two offset tables at LDS address 0 (a multiple of 1 KiB), the stages behind them
  __shared__ __attribute__((aligned(1024))) char lds_all[TOFF + 3 * STG];
  char* const smem = lds_all + TOFF;
  // flattened workgroup id, spread so that workgroups b, b+8, ... (one XCD) hold consecutive ids: the tiles of one (class, split)
  // stream the same A / B rows and share that XCD's L2
  int f;
  {
    const int nw = a.total_wgs, q8 = nw >> 3, r8 = nw & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    f = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  int cls = 0;
#pragma unroll
  for (int c = 1; c < SG_MAX_CLS; ++c) cls += (c < a.ncls && f >= a.cls_wg0[c]) ? 1 : 0;
  cls = __builtin_amdgcn_readfirstlane(cls);
  const int rem = f - a.cls_wg0[cls];
  const int split = rem / a.tiles_per_cls, tile = rem - split * a.tiles_per_cls;
  const int T = a.cls_T[cls];
  if (a.tbl) {
    if (T == 8) sg_body<8, NXI, true>(a, smem, cls, tile, split);
#ifdef S2P_DIAG_BUILD
    else if (T == 2) sg_body<2, NXI, true>(a, smem, cls, tile, split);
    else if (T == 1) sg_body<1, NXI, true>(a, smem, cls, tile, split);
#endif
    else sg_body<4, NXI, true>(a, smem, cls, tile, split);
    return;
  }
  if (T == 8) sg_body<8, NXI, false>(a, smem, cls, tile, split);
#ifdef S2P_DIAG_BUILD
  else if (T == 2) sg_body<2, NXI, false>(a, smem, cls, tile, split);
  else if (T == 1) sg_body<1, NXI, false>(a, smem, cls, tile, split);
#endif
  else sg_body<4, NXI, false>(a, smem, cls, tile, split);
}

// dW[tile] += sum over the S partial tiles of the class, in a fixed order (bitwise reproducible).  These layers have few tiles and
// many splits (8 tiles x 64 splits for the 64 -> 128 convs), so the reduce is spread as well: one workgroup per (class, tile, tap,
// group of 4 rows) = 64 float4 outputs, each summed by FOUR threads (a quarter of the splits each, eight loads in flight) whose
// partial sums are combined in segment order.
__global__ __launch_bounds__(256) void wgrad_slabg_reduce_kernel(const WgSlabGArgs a) {
  __shared__ f32x4 red[4][64];
  __shared__ float redb[4][64];
  const int blk = blockIdx.x >> 4, rq = blockIdx.x & 15;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < SG_MAX_CLS; ++c) cls += (c < a.ncls && blk >= a.cls_red0[c]) ? 1 : 0;
  const int T = a.cls_T[cls], S = a.cls_S[cls];
  const int rem = blk - a.cls_red0[cls];
  const int tile = rem / T, t = rem - tile * T;
  const int co_t = tile % a.co_tiles, ci_s = tile / a.co_tiles;
  const size_t chunk = (size_t)64 * T * 64;
  const float* sl = a.slab + a.cls_slab0[cls] + (size_t)tile * S * chunk;
  const int seg = threadIdx.x >> 6, j = threadIdx.x & 63;
  const int per = (S + 3) >> 2, k0 = seg * per, k1 = k0 + per < S ? k0 + per : S;
  const int qd = j & 15, row = rq * 4 + (j >> 4);
  const size_t so = ((size_t)(row * T + t) * 16 + qd) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(sl + (size_t)(k + u) * chunk + so);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < k1; ++k) s += *(const f32x4*)(sl + (size_t)k * chunk + so);
  red[seg][j] = s;
  const bool bias = cls == 0 && t == 0 && rq == 0 && ci_s == 0 && a.db != nullptr;          // workgroup-uniform
  if (bias) {
    const float* sb = a.slabb + (size_t)co_t * S * 64;
    float b = 0.f;
    for (int kk = k0; kk < k1; ++kk) b += sb[kk * 64 + j];
    redb[seg][j] = b;
  }
  __syncthreads();
  if (seg == 0) {
    const f32x4 tot = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
    float* o = a.dW + (size_t)(co_t * 64 + row) * a.dw_row + a.wt[cls][t] * a.Cb + ci_s * 64 + qd * 4;
    *(f32x4*)o = *(const f32x4*)o + tot;
    if (bias) a.db[co_t * 64 + j] += ((redb[0][j] + redb[1][j]) + redb[2][j]) + redb[3][j];
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------------------
static int sg_ncu() {
  static int ncu = 0;
  if (!ncu) { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev); ncu = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }
  return ncu;
}

// fills everything but the pointers; false: the layer is outside this kernel's scope.  `nxi` = window rows / 32 (3 or 4)
static bool sg_plan(const s2p_conv_desc* d, int cin_real, int cout_real, WgSlabGArgs& a, int& nxi, size_t& ws_floats) {
  if (!d || d->dtype != S2P_BF16 || d->reflect || d->groups != 1 || d->KH != d->KW) return false;
  if (d->Cin % 64 || d->Cout % 64 || cin_real != d->Cin || cout_real != d->Cout) return false;
  const int K = d->KH, s = d->stride, pad = d->pad;
  const bool k4 = K == 4 && pad == 2 && (s == 2 || (s == 1 && !d->transposed));
  const bool k3 = K == 3 && pad == 1 && s == 2 && S2P_DIAG_SWITCH(6);      // diagnostics build only: the encoder / decoder convs
  if (!k4 && !k3) return false;
  int Ca, Cb;
  if (!d->transposed) { a.Ha = d->Ho; a.Wa = d->Wo; a.Hb = d->H; a.Wb = d->W; Ca = d->Cout; Cb = d->Cin; a.a_pitch = d->y_pitch; a.b_pitch = d->x_pitch; }
  else { a.Ha = d->H; a.Wa = d->W; a.Hb = d->Ho; a.Wb = d->Wo; Ca = d->Cin; Cb = d->Cout; a.a_pitch = d->x_pitch; a.b_pitch = d->y_pitch; }
  a.N = d->N; a.bs = s; a.Cb = Cb; a.dw_row = K * K * Cb;
  a.co_tiles = Ca / 64; a.tiles_per_cls = a.co_tiles * (Cb / 64);
  // taps -> (class, stride-1 shift)
  int ta[16], tb[16], tc[16];
  int amin = 0, amax = 0, bmin = 0, bmax = 0;
  for (int ky = 0; ky < K; ++ky)
    for (int kx = 0; kx < K; ++kx) {
      const int t = ky * K + kx, dy = ky - pad, dx = kx - pad;
      if (s == 2) {
        const int py = dy & 1, px = dx & 1;
        ta[t] = (dy - py) >> 1; tb[t] = (dx - px) >> 1; tc[t] = py * 2 + px;
      } else { ta[t] = dy; tb[t] = dx; tc[t] = ky >> 1; }
      amin = ta[t] < amin ? ta[t] : amin; amax = ta[t] > amax ? ta[t] : amax;
      bmin = tb[t] < bmin ? tb[t] : bmin; bmax = tb[t] > bmax ? tb[t] : bmax;
    }
  const int Hsub = s == 2 ? (a.Hb + 1) / 2 : a.Hb, Wsub = s == 2 ? (a.Wb + 1) / 2 : a.Wb;
  a.Hp = a.Ha + amax > Hsub - amin ? a.Ha + amax : Hsub - amin;
  a.Wp = a.Wa + bmax > Wsub - bmin ? a.Wa + bmax : Wsub - bmin;
  const long long npos = (long long)a.N * a.Hp * a.Wp;
  if (npos >= (1 << 23) || (long long)a.N * a.Hb * a.Wb >= (1 << 23) || (long long)a.N * a.Ha * a.Wa >= (1 << 23)) return false;
  const long long ab = (long long)a.N * a.Ha * a.Wa * a.a_pitch * 2, bb = (long long)a.N * a.Hb * a.Wb * a.b_pitch * 2;
  if (ab >= (1ll << 31) || bb >= (1ll << 31)) return false;
  a.a_bytes = (unsigned)ab; a.b_bytes = (unsigned)bb;
  a.tbl = (a.Hp * a.Wp <= SG_TBL_MAX && ab <= (1ll << 29) && bb <= (1ll << 29) && !S2P_DIAG_SWITCH(18)) ? 1 : 0;
  a.nblocks = cdiv(npos, 64);
  // classes in order of decreasing tap count (the long workgroups start first)
  const int ncand = s == 2 ? 4 : 2;
  int order[4] = {0, 1, 2, 3}, cnt[4] = {0, 0, 0, 0};
  for (int t = 0; t < K * K; ++t) ++cnt[tc[t]];
  for (int i = 0; i < ncand; ++i)
    for (int j = i + 1; j < ncand; ++j)
      if (cnt[order[j]] > cnt[order[i]]) { int o = order[i]; order[i] = order[j]; order[j] = o; }
  a.ncls = 0;
  int span = 0;
  for (int i = 0; i < ncand; ++i) {
    const int c = order[i];
    if (!cnt[c]) continue;
    if (cnt[c] != 4 && cnt[c] != 8 && !(k3 && (cnt[c] == 1 || cnt[c] == 2))) return false;
    const int k = a.ncls++;
    a.cls_T[k] = cnt[c]; a.cls_py[k] = s == 2 ? c >> 1 : 0; a.cls_px[k] = s == 2 ? c & 1 : 0;
    int lo = 0, hi = 0, n = 0;
    for (int t = 0; t < K * K; ++t)
      if (tc[t] == c) {
        const int off = ta[t] * a.Wp + tb[t];
        a.toff[k][n] = off; a.wt[k][n] = t; ++n;
        lo = off < lo ? off : lo; hi = off > hi ? off : hi;
      }
    a.cls_hneg[k] = -lo;
    span = 64 + hi - lo > span ? 64 + hi - lo : span;
  }
  if (span > 128) return false;
  nxi = span <= 96 ? 3 : 4;
  // K splits per class, in proportion to its taps: ~2 workgroups per CU in total
  // K splits: ~2 workgroups per CU in total, the same number for every class (a block costs a class of 4 taps as much as one of
  // 8: these launches are bound by the DMA round trips of a block, not by its MFMAs)
  const double unit = (S2P_DIAG_SWITCH(16) ? 1.0 : 2.0) * sg_ncu() / ((double)a.tiles_per_cls * a.ncls);
  int wg = 0, red = 0;
  size_t fl = 0;
  for (int k = 0; k < a.ncls; ++k) {
    int S = (int)(unit + 0.5);
    if (S > 128) S = 128;
    if (S > a.nblocks) S = a.nblocks;
    if (S < 1) S = 1;
    a.cls_bps[k] = cdiv(a.nblocks, S);
    a.cls_S[k] = cdiv(a.nblocks, a.cls_bps[k]);
    a.cls_wg0[k] = wg; wg += a.tiles_per_cls * a.cls_S[k];
    a.cls_red0[k] = red; red += a.tiles_per_cls * a.cls_T[k];
    a.cls_slab0[k] = (long long)fl; fl += (size_t)a.tiles_per_cls * a.cls_S[k] * 64 * a.cls_T[k] * 64;
  }
  for (int k = a.ncls; k <= SG_MAX_CLS; ++k) { a.cls_wg0[k] = wg; a.cls_red0[k] = red; }
  a.total_wgs = wg;
  ws_floats = fl + (size_t)a.co_tiles * a.cls_S[0] * 64;        // + the bias partials of class 0
  return true;
}

bool s2p_wgrad_slabg_supported(const s2p_conv_desc* d, int cin_real, int cout_real) {
  if (S2P_DIAG_SWITCH(5)) return false;
  WgSlabGArgs a{}; int nxi; size_t fl;
  return sg_plan(d, cin_real, cout_real, a, nxi, fl);
}

size_t s2p_wgrad_slabg_workspace(const s2p_conv_desc* d, int cin_real, int cout_real) {
  WgSlabGArgs a{}; int nxi; size_t fl;
  if (!sg_plan(d, cin_real, cout_real, a, nxi, fl)) return 0;
  return fl * sizeof(float);
}

int s2p_wgrad_slabg(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db, int cin_real, int cout_real,
                    void* workspace, size_t workspace_bytes, hipStream_t st) {
  WgSlabGArgs a{}; int nxi; size_t fl;
  if (!sg_plan(d, cin_real, cout_real, a, nxi, fl)) S2P_FAIL(-2, "s2p_wgrad_slabg: unsupported geometry");
  if (!workspace || workspace_bytes < fl * sizeof(float))
    S2P_FAIL(-1, "s2p_wgrad_slabg: workspace of %zu bytes needed, got %zu", fl * sizeof(float), workspace_bytes);
  a.A = d->transposed ? x : dy; a.B = d->transposed ? dy : x;
  a.dW = dw; a.db = db;
  a.slab = (float*)workspace;
  a.slabb = a.slab + (fl - (size_t)a.co_tiles * a.cls_S[0] * 64);
  if (nxi == 3) hipLaunchKernelGGL(wgrad_slabg_kernel<3>, dim3(a.total_wgs), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad_slabg_kernel<4>, dim3(a.total_wgs), dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("wgrad_slabg_kernel");
  hipLaunchKernelGGL(wgrad_slabg_reduce_kernel, dim3(a.cls_red0[a.ncls] * 16), dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("wgrad_slabg_reduce_kernel");
  return 0;
}
