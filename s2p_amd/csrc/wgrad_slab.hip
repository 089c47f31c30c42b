// Batched weight gradient of stride-1 "same" convolutions (ResBlk 3x3 convs, the grouped gamma/beta 3x3 convs) on
// gfx950:   dW[co][t][ci] (+)= sum_p dY[p][co] * X[p + (dy_t, dx_t)][ci],   db[co] (+)= sum_p dY[p][co].
//
// Why a second wgrad kernel.  In the generic implicit GEMM (wgrad_igemm.hip) a 128x128 output tile holds ONE tap, so
// every tap re-streams its own shifted copy of X and every tile re-streams dY: 64 FLOP per L2->LDS byte, and the
// 256x2304 output of a ResBlk conv (36 tiles) only fills the chip through split-K 16 with fp32 atomics (16x write
// amplification, run-to-run last-bit noise).  Here
//   * both operands are addressed on a PADDED RASTER: positions k = (n*Hp + r)*Wp + c with Hp = H + pad, Wp = W + pad;
//     pad rows / columns are zero (the LDS-DMA's out-of-range offset returns zeros), so a tap is a pure row shift
//     k -> k + dy*Wp + dx and image borders need no masks (a shifted read that leaves the image lands on a pad
//     position; a pad position of dY contributes 0).  Cost: (Hp*Wp)/(H*W) = 1.10 more MFMA work at 21x21;
//   * a workgroup (4 waves) owns a 64(co) x 9 taps x 64(ci) output tile: one dY stage (64 positions x 64 co) and one X
//     window (64 positions + halo, 64 ci) feed all nine taps -> 214 FLOP per L2->LDS byte, and each wave keeps
//     9 accumulator tiles (32 co x 32 ci per tap) so an A fragment is reused nine times;
//   * several layers (jobs) share one launch: 12 ResBlk convs x 16 tiles x S K-splits fill the chip with S = 2..4
//     instead of 16, partial tiles go to a slab with plain stores and a second tiny kernel adds them up in a fixed
//     order: no atomics, bitwise reproducible, dW written once.
// MFMA fragments come from `ds_read_b64_tr_b16` ([position][channel] staging, as the tensors lie in HBM); 128-byte LDS
// rows with the 16-byte chunk index XOR-ed by ((row >> 1) & 1) << 2 (on the DMA source address and on the read) are
// conflict-free for the four rows x 64 bytes a half-wave touches.  3-stage LDS-DMA pipeline, counted vmcnt, raw barrier.
#include "s2p_common.h"
#include <type_traits>

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

constexpr int WS_MAX_JOBS = 16;
struct WgSlabArgs {
  const void* A[WS_MAX_JOBS]; const void* B[WS_MAX_JOBS]; float* dW[WS_MAX_JOBS]; float* db[WS_MAX_JOBS];
  float* slab; float* slabb;
  int njobs;
  int N, H, W, Hp, Wp, Kp;
  int a_pitch, b_pitch, Cout, Cin;
  int co_tiles, tiles_per_job;
  int S, blocks_per_split, nblocks;
  int halo, wrows;
  int toff[9];
  unsigned a_bytes, b_bytes;
  int total_wgs;
};

// TBL: the padded raster of one image (Hp * Wp <= WS_TBL_MAX positions) is tabulated in LDS once per workgroup, one table per
// operand -- position -> byte offset of the pixel inside its image, 0xc0000000 where the position is padding -- and a DMA's source
// offset is  table[q] + (image offset + chunk offset):  6 VALU instructions per DMA (advance q with one wrap, one add) and one
// 4-byte LDS read issued a whole block ahead, instead of ~19 for the (n, row, column) state with its bounds tests and two
// multiplies.  Round 5's instruction-mix counters: the kernel spent 3.7 VALU instructions per MFMA, nearly all of them on these
// addresses, and 81 % of its time is VALU + MFMA issue.  A padding position adds up to an offset in [2^31 + 2^29, 2^32 - 2^29) and
// an image index outside [0, N) to one below 0 or beyond the tensor: either way the buffer range check returns zeros (the host
// keeps the tensors below 2^29 bytes on this path).
constexpr int WS_TBL_MAX = 768;
template <int WROWS, bool TBL>
__global__ __launch_bounds__(256, 2) void wgrad_slab_kernel(const WgSlabArgs a) {
  constexpr int T = 9, RS = 128, NST = 3;
  constexpr int ASTG = 64 * RS;                 // 8 KiB: 64 positions x 64 co
  constexpr int XSTG = WROWS * RS;              // X window: 64 positions + halo both sides, 64 ci
  constexpr int STG = ASTG + XSTG;
  constexpr int NXI = WROWS / 32;               // X DMA instructions per wave per block (8 rows each, 4 waves)
  constexpr int NDMA = 2 + NXI;
  // the tables sit at LDS address 0 (their byte offsets fit a ds_read's immediate: no address arithmetic), the stages behind them
  constexpr int TOFF = TBL ? 2 * WS_TBL_MAX * 4 : 0;           // [A | B], a multiple of 1 KiB
  __shared__ __attribute__((aligned(1024))) char lds_all[TOFF + NST * STG];
  char* const smem = lds_all + TOFF;
  unsigned* const ptab = (unsigned*)lds_all;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // flattened workgroup id, job-major, spread so that workgroups b, b+8, ... (one XCD) hold consecutive ids: the tiles
  // of one job stream the same dY / X rows and share that XCD's L2
  int f;
  {
    const int nw = a.total_wgs, q8 = nw >> 3, r8 = nw & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    f = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int per_job = a.tiles_per_job * a.S;
  const int job = f / per_job;
  const int rem = f - job * per_job;
  const int split = rem / a.tiles_per_job, tile = rem - split * a.tiles_per_job;
  const int co_t = tile % a.co_tiles, ci_s = tile / a.co_tiles;
  const int b0 = split * a.blocks_per_split;
  int b1 = b0 + a.blocks_per_split; if (b1 > a.nblocks) b1 = a.nblocks;
  const int nblk = b1 - b0;                                   // >= 1 by construction

  const unsigned OOB = 0x80000000u;
  const i32x4 ar = s2p_make_rsrc(a.A[job], a.a_bytes);
  const i32x4 br = s2p_make_rsrc(a.B[job], a.b_bytes);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(s2p_lds_addr(smem));

  // ---- DMA geometry: a piece = 8 rows x 128 B; lane -> (row in piece, physical chunk).  Every lane keeps the padded-
  //      raster coordinates (n, r, c) of the NDMA rows it stages and advances them by 64 positions per block with a few
  //      branch-free adds / selects (a division per DMA would cost more issue slots than the block's MFMAs).
  const int lrow = lane >> 3, pch = lane & 7;
  const int a_cbyte = (co_t * 64) * 2, b_cbyte = (ci_s * 64) * 2;
  int pn[NDMA], prr[NDMA], pc[NDMA];
  int cb[NDMA];                                                // chunk byte offset (swizzled) + channel base
  // TBL: q4 = 4 * (position inside its image's padded raster), noffc = image offset + chunk offset (bytes), tv = the table entry of q
  unsigned q4[NDMA], noffc[NDMA], tv[NDMA];
  const int HpWp = a.Hp * a.Wp;
  if constexpr (TBL) {
    for (int q = tid; q < HpWp; q += 256) {
      const int r = q / a.Wp, c = q - r * a.Wp;
      const bool in = c < a.W && r < a.H;
      ptab[q] = in ? (unsigned)(r * a.W + c) * (unsigned)(a.a_pitch * 2) : 0xc0000000u;
      ptab[WS_TBL_MAX + q] = in ? (unsigned)(r * a.W + c) * (unsigned)(a.b_pitch * 2) : 0xc0000000u;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int row = (4 * (i < 2 ? i : i - 2) + wave) * 8 + lrow;
    int pos = b0 * 64 + row - (i < 2 ? 0 : a.halo);
    int nadj = 0;
    if (pos < 0) { pos += a.Hp * a.Wp; nadj = -1; }            // pos >= -halo > -Hp*Wp
    cb[i] = (i < 2 ? a_cbyte : b_cbyte) + ((pch ^ (((row >> 1) & 1) << 2)) * 16);
    if constexpr (TBL) {
      const int n = pos / HpWp, q = pos - n * HpWp;
      q4[i] = (unsigned)q * 4u;
      noffc[i] = (unsigned)((n + nadj) * (a.H * a.W * (i < 2 ? a.a_pitch : a.b_pitch) * 2) + cb[i]);
      tv[i] = ptab[(i < 2 ? 0 : WS_TBL_MAX) + q];
    } else {
      const int q1 = pos / a.Wp;
      pc[i] = pos - q1 * a.Wp;
      const int n = q1 / a.Hp;
      prr[i] = q1 - n * a.Hp;
      pn[i] = n + nadj;
    }
  }
  const int a_pitch2 = a.a_pitch * 2, b_pitch2 = a.b_pitch * 2;
  // 64 positions = adv_n images + adv_r rows + adv_c columns (block-uniform scalars)
  const int adv_q = 64 / a.Wp, adv_c = 64 - adv_q * a.Wp, adv_n = adv_q / a.Hp, adv_r = adv_q - adv_n * a.Hp;
  // TBL: 64 positions = tadv_n images + tadv_q positions; HpWp4 = 4 Hp Wp
  const int tadv_n = 64 / HpWp;
  const unsigned tadv_q4 = (unsigned)(64 - tadv_n * HpWp) * 4u, HpWp4 = (unsigned)HpWp * 4u;
  const unsigned img_a = (unsigned)(a.H * a.W) * (unsigned)a_pitch2, img_b = (unsigned)(a.H * a.W) * (unsigned)b_pitch2;
  // one DMA (index i of this wave's NDMA per block) of the block the coordinate state points at, then advance that state
  auto issue_one = [&](auto ic, unsigned base) {
    constexpr int i = decltype(ic)::value;
    if constexpr (TBL) {
      const unsigned off = tv[i] + noffc[i];
      if (i < 2) s2p_dma16(ar, base + (4 * i + wave) * 1024, (int)off);
      else s2p_dma16(br, base + ASTG + (4 * (i - 2) + wave) * 1024, (int)off);
      const unsigned img = i < 2 ? img_a : img_b;
      const unsigned qa = q4[i] + tadv_q4;
      const bool wrap = qa >= HpWp4;
      q4[i] = wrap ? qa - HpWp4 : qa;
      noffc[i] += (unsigned)tadv_n * img + (wrap ? img : 0u);
      tv[i] = *(const unsigned*)((const char*)(ptab + (i < 2 ? 0 : WS_TBL_MAX)) + q4[i]);      // consumed a whole block later
    } else {
    const bool ok = pc[i] < a.W && prr[i] < a.H && (unsigned)pn[i] < (unsigned)a.N;
    const int pix = __mul24(__mul24(pn[i], a.H) + prr[i], a.W) + pc[i];               // < 2^24 (host-checked)
    int off = __mul24(pix, i < 2 ? a_pitch2 : b_pitch2) + cb[i];
    off = ok ? off : (int)OOB;
    if (i < 2) s2p_dma16(ar, base + (4 * i + wave) * 1024, off);
    else s2p_dma16(br, base + ASTG + (4 * (i - 2) + wave) * 1024, off);
    int c = pc[i] + adv_c, r = prr[i] + adv_r, n = pn[i] + adv_n;
    const bool cw = c >= a.Wp;
    c = cw ? c - a.Wp : c; r += cw ? 1 : 0;
    const bool rw = r >= a.Hp;
    r = rw ? r - a.Hp : r; n += rw ? 1 : 0;
    pc[i] = c; prr[i] = r; pn[i] = n;
    }
  };
  auto issue = [&](int stage) {
    const unsigned base = lds0 + stage * STG;
    static_for<0, NDMA>([&](auto ic) { issue_one(ic, base); });
  };

  // ---- fragment geometry (ds_read_b64_tr_b16): 16-lane group gq: channel block 16*(gq&1), k half gq>>1; inside the
  //      group lane 4q+p supplies row q, columns 4p..4p+3 -------------------------------------------------------------
  const int gq = lane >> 4, gg = gq & 1, hh = gq >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int wa = wave >> 1, wb = wave & 1;                     // wave tile: co [32wa, +32) x ci [32wb, +32) for every tap
  const int a_lane = (8 * hh + q) * RS + (((4 * wa + 2 * gg + (p >> 1)) ^ ((q >> 1) << 2)) * 16) + 8 * (p & 1);
  int b_lane[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int r0 = a.halo + a.toff[t] + 8 * hh + q;            // window row of this lane for k = 0, half 0
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
  const bool do_bias = a.db[job] != nullptr && ci_s == 0 && wb == 0;         // wave-uniform
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  // ---- 3-stage pipeline over the blocks of this split ----------------------------------------------------------------
  issue(0);
  if (nblk > 1) { issue(1); S2P_WAIT_VMCNT(NDMA); } else { S2P_WAIT_VMCNT(0); }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // Inside a block the 36 (substep, tap) MFMAs of a wave run as one software pipeline: the two transposed reads of the B
  // fragment used LA steps later and, once per substep, the next A fragment are issued right behind each MFMA, so an
  // MFMA never waits on a read issued less than ~LA x 32 cycles earlier; the block's DMAs (for block kb + 2) are spread
  // over the steps instead of being issued as one burst in front of the first read.
  constexpr int LA = 4, RING = LA + 1, NSTEP = 4 * T;
  auto read_frag = [&](const char* ptr) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ptr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ptr + 4 * RS));
    // (a concatenation, not eight element inserts: the inserts cost four v_mov_b32 per fragment -- 4.7 VALU instructions per MFMA
    // in this loop, round 5's instruction-mix counters)
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
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
      static_for<0, LA>([&](auto vc) {
        constexpr int v = decltype(vc)::value;
        BF[v % RING] = read_frag(sb + b_lane[v % T] + (v / T) * 16 * RS);
      });
      static_for<0, NSTEP>([&](auto uc) {
        constexpr int u = decltype(uc)::value, s_ = u / T, t = u % T;
        if constexpr (BIAS && t == 0) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[s_ & 1], ones, accb, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[s_ & 1], BF[u % RING], acc[t], 0, 0, 0);
        constexpr int v = u + LA;
        if constexpr (v < NSTEP) BF[v % RING] = read_frag(sb + b_lane[v % T] + (v / T) * 16 * RS);
        if constexpr (t == T - 1 - LA && s_ < 3) AF[(s_ + 1) & 1] = read_frag(sb + a_lane + (s_ + 1) * 16 * RS);
        // DMA i of block kb + 2 goes out behind step (i * NSTEP) / NDMA + 1
        static_for<0, NDMA>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (u == (i * NSTEP) / NDMA + 1) { if (more) issue_one(ic, dbase); }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      // block kb+1 must have landed (for every wave) before anyone reads it; block kb+2 may stay in flight
      if (more) S2P_WAIT_VMCNT(NDMA); else S2P_WAIT_VMCNT(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (++stage == NST) stage = 0;
    }
  };
  if (do_bias) main_loop(std::integral_constant<bool, true>{});
  else main_loop(std::integral_constant<bool, false>{});

  // ---- epilogue: lanes <-> consecutive ci (contiguous floats), registers <-> co rows --------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int tile_g = job * a.tiles_per_job + tile;
  if (a.S == 1) {
    float* dW = a.dW[job];
    const int row_len = T * a.Cin;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co_t * 64 + 32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h;
        float* o = dW + (size_t)co * row_len + t * a.Cin + ci_s * 64 + 32 * wb + r;
        *o += acc[t][e];
      }
    if (do_bias && r == 0) {
      float* db = a.db[job];
#pragma unroll
      for (int e = 0; e < 16; ++e) db[co_t * 64 + 32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h] += accb[e];
    }
  } else {
    float* sl = a.slab + ((size_t)tile_g * a.S + split) * (64 * T * 64);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h;
        sl[(row * T + t) * 64 + 32 * wb + r] = acc[t][e];
      }
    if (do_bias && r == 0) {
      float* sb2 = a.slabb + ((size_t)tile_g * a.S + split) * 64;
#pragma unroll
      for (int e = 0; e < 16; ++e) sb2[32 * wa + (e & 3) + 8 * (e >> 2) + 4 * h] = accb[e];
    }
  }
}

// dW[tile] += sum over the S partial tiles, in split order (fixed order: bitwise reproducible).  One workgroup per
// (tile, tap): 64 rows x 64 ci = 1024 float4, four per thread, all S partial loads of a thread in flight together.
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const WgSlabArgs a) {
  constexpr int T = 9;
  const int tile_g = blockIdx.x / T, t = blockIdx.x - tile_g * T;
  const int job = tile_g / a.tiles_per_job, tile = tile_g - job * a.tiles_per_job;
  const int co_t = tile % a.co_tiles, ci_s = tile / a.co_tiles;
  const float* sl = a.slab + (size_t)tile_g * a.S * (64 * T * 64);
  float* dW = a.dW[job];
  const int row_len = T * a.Cin;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = threadIdx.x + 256 * u;                       // float4 index inside the [64 rows][16 quads] tap plane
    const int qd = i & 15, row = i >> 4;
    const size_t so = ((size_t)(row * T + t) * 16 + qd) * 4;
    f32x4 s = *(const f32x4*)(sl + so);
    for (int k = 1; k < a.S; ++k) s += *(const f32x4*)(sl + (size_t)k * (64 * T * 64) + so);
    float* o = dW + (size_t)(co_t * 64 + row) * row_len + t * a.Cin + ci_s * 64 + qd * 4;
    *(f32x4*)o = *(const f32x4*)o + s;
  }
  if (t == 0 && a.db[job] != nullptr && ci_s == 0 && threadIdx.x < 64) {
    const float* sb = a.slabb + (size_t)tile_g * a.S * 64;
    float s = sb[threadIdx.x];
    for (int k = 1; k < a.S; ++k) s += sb[k * 64 + threadIdx.x];
    a.db[job][co_t * 64 + threadIdx.x] += s;
  }
}

static int ws_window_rows(int halo) { return (64 + 2 * halo + 31) / 32 * 32; }

static bool ws_supported(const s2p_conv_desc* d, int n_jobs, int cin_real, int cout_real) {
  if (d->dtype != S2P_BF16 || d->transposed || d->reflect || d->groups != 1) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
  if (d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin % 64 || d->Cout % 64 || cin_real != d->Cin || cout_real != d->Cout) return false;
  if (n_jobs < 1 || n_jobs > WS_MAX_JOBS) return false;
  const int halo = (d->W + 1) + 1;
  if (ws_window_rows(halo) > 256) return false;
  const long long ab = (long long)d->N * d->H * d->W * d->y_pitch * 2, bb = (long long)d->N * d->H * d->W * d->x_pitch * 2;
  if (ab >= (1ll << 31) || bb >= (1ll << 31)) return false;
  if ((long long)d->N * (d->H + 1) * (d->W + 1) >= (1 << 23)) return false;      // positions must be exact in float
  return true;
}

static int ws_splits(const s2p_conv_desc* d, int n_jobs) {
  static int ncu = 0;
  if (!ncu) { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev); ncu = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }
  const int tiles = n_jobs * (d->Cout / 64) * (d->Cin / 64);
  const int nblocks = cdiv((long long)d->N * (d->H + 1) * (d->W + 1), 64);
  const int forced = s2p_env_int("S2P_WGRAD_SLAB_SPLITS", 0);
  int S = forced > 0 ? forced : cdiv(3 * ncu, tiles);          // ~3 workgroups per CU (two resident, one queued)
  if (S > 8) S = 8;
  if (S > nblocks) S = nblocks;
  if (S < 1) S = 1;
  const int bps = cdiv(nblocks, S);
  return cdiv(nblocks, bps);
}

extern "C" size_t s2p_conv2d_wgrad_batched_workspace(const s2p_conv_desc* d, int n_jobs, int cin_real, int cout_real) {
  if (d && n_jobs >= 1 && s2p_head_wgrad_supported(d, cin_real, cout_real)) return (size_t)n_jobs * s2p_head_wgrad_workspace(d);
  if (!d) return 0;
  if (!ws_supported(d, n_jobs, cin_real, cout_real)) return s2p_conv2d_wgrad_workspace(d, cin_real, cout_real);   // jobs run one after the other
  const int S = ws_splits(d, n_jobs);
  if (S == 1) return 0;
  const size_t tiles = (size_t)n_jobs * (d->Cout / 64) * (d->Cin / 64);
  return tiles * S * (64 * 9 * 64 + 64) * sizeof(float);
}

extern "C" int s2p_conv2d_wgrad_batched(const s2p_conv_desc* d, const s2p_wgrad_job* jobs, int n_jobs, int cin_real,
                                        int cout_real, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !jobs || n_jobs < 1) S2P_FAIL(-1, "s2p_conv2d_wgrad_batched: null pointer / no jobs");
  for (int j = 0; j < n_jobs; ++j)
    if (!jobs[j].x || !jobs[j].dy || !jobs[j].dw) S2P_FAIL(-1, "s2p_conv2d_wgrad_batched: job %d has a null pointer", j);
  if (s2p_head_wgrad_supported(d, cin_real, cout_real)) {
    // PatchGAN logit heads (Cout = 1): activation-stationary kernel + fixed-order partial reduce (csrc/wgrad_head.hip)
    const size_t per = s2p_head_wgrad_workspace(d);
    if (!workspace || workspace_bytes < per * n_jobs) S2P_FAIL(-1, "s2p_conv2d_wgrad_batched: workspace of %zu bytes needed", per * n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
      int rc = s2p_head_wgrad(d, jobs[j].x, jobs[j].dy, jobs[j].dw, jobs[j].db, cin_real, (char*)workspace + j * per, per,
                              (hipStream_t)stream);
      if (rc) return rc;
    }
    return 0;
  }
  if (!ws_supported(d, n_jobs, cin_real, cout_real)) {
    // geometry outside the slab kernel's scope (other taps / strides / dtypes): one generic launch per job, stream-ordered
    // on the same workspace
    for (int j = 0; j < n_jobs; ++j) {
      int rc = s2p_conv2d_wgrad_ws(d, jobs[j].x, jobs[j].dy, jobs[j].dw, jobs[j].db, cin_real, cout_real, 0, 0, workspace,
                                   workspace_bytes, stream);
      if (rc) return rc;
    }
    return 0;
  }
  WgSlabArgs a{};
  a.njobs = n_jobs;
  for (int j = 0; j < n_jobs; ++j) { a.A[j] = jobs[j].dy; a.B[j] = jobs[j].x; a.dW[j] = jobs[j].dw; a.db[j] = jobs[j].db; }
  a.N = d->N; a.H = d->H; a.W = d->W; a.Hp = d->H + 1; a.Wp = d->W + 1; a.Kp = d->N * a.Hp * a.Wp;
  a.a_pitch = d->y_pitch; a.b_pitch = d->x_pitch; a.Cout = d->Cout; a.Cin = d->Cin;
  a.co_tiles = d->Cout / 64; a.tiles_per_job = a.co_tiles * (d->Cin / 64);
  a.nblocks = cdiv(a.Kp, 64);
  a.S = ws_splits(d, n_jobs);
  a.blocks_per_split = cdiv(a.nblocks, a.S);
  a.halo = a.Wp + 1; a.wrows = ws_window_rows(a.halo);
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) a.toff[ky * 3 + kx] = (ky - 1) * a.Wp + (kx - 1);
  a.a_bytes = (unsigned)((long long)d->N * d->H * d->W * d->y_pitch * 2);
  a.b_bytes = (unsigned)((long long)d->N * d->H * d->W * d->x_pitch * 2);
  const int tiles = n_jobs * a.tiles_per_job;
  a.total_wgs = tiles * a.S;
  if (a.S > 1) {
    const size_t need = (size_t)tiles * a.S * (64 * 9 * 64 + 64) * sizeof(float);
    if (!workspace || workspace_bytes < need)
      S2P_FAIL(-1, "s2p_conv2d_wgrad_batched: workspace of %zu bytes needed (s2p_conv2d_wgrad_batched_workspace), got %zu", need, workspace_bytes);
    a.slab = (float*)workspace;
    a.slabb = a.slab + (size_t)tiles * a.S * (64 * 9 * 64);
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(a.total_wgs);
  // tabulated padded raster (see the kernel): images of <= WS_TBL_MAX padded positions, tensors of <= 2^29 bytes, image offsets that
  // stay exact in 32 bits; switch 18 of the diagnostics build selects the (n, row, column) state everywhere
  const bool tbl = a.Hp * a.Wp <= WS_TBL_MAX && a.a_bytes <= (1u << 29) && a.b_bytes <= (1u << 29) && !S2P_DIAG_SWITCH(18);
  if (a.wrows <= 128) { if (tbl) hipLaunchKernelGGL((wgrad_slab_kernel<128, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_slab_kernel<128, false>), grid, dim3(256), 0, st, a); }
  else if (a.wrows <= 192) { if (tbl) hipLaunchKernelGGL((wgrad_slab_kernel<192, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_slab_kernel<192, false>), grid, dim3(256), 0, st, a); }
  else { if (tbl) hipLaunchKernelGGL((wgrad_slab_kernel<256, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((wgrad_slab_kernel<256, false>), grid, dim3(256), 0, st, a); }
  S2P_CHECK_LAUNCH("wgrad_slab_kernel");
  if (a.S > 1) {
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(tiles * 9), dim3(256), 0, st, a);
    S2P_CHECK_LAUNCH("wgrad_slab_reduce_kernel");
  }
  return 0;
}
