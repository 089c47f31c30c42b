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
  const void* A; const void* B; float* dW; float* db;
  unsigned a_bytes, b_bytes;
  int diag;
  int groups, xcd_map;          // DMA kernel: 1-D grid, all tiles of one (group, split) unit on one XCD
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
  // deterministic mode (DMA kernel): every (split, group) unit stores its partial tile sums to `part`
  // [unit][part_rows][part_cols] (and bias partials to part_b [unit][part_rows]); wgrad_part_reduce_kernel adds them to
  // dW / db in unit order.  part == nullptr: fp32 atomics straight into dW / db.
  float* part; float* part_b; int part_rows, part_cols;
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

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16, tensors < 2 GiB): both operand tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds`
// (no register staging, no ds_write pass, out-of-range rows / columns / padding taps come back as zeros from the
// buffer range check).  Rows are 256 B unpadded; the 16-byte chunk index is XOR-swizzled with
// ((row & 3) << 2) | ((row >> 2) & 3) on the source address and on the transposed read, which keeps
// ds_read_b64_tr_b16 conflict-free (one image serves the 32x32x16 A and B fragments).
// Workgroups of B-tile 0 also accumulate the bias gradient  db[a] += sum_m A[m][a]  with one extra MFMA against a
// ones fragment per A fragment, so dY is not read a second time by a separate reduction kernel.
// TBL (round 5): the gathered operand's source offsets come from two small tables in LDS per tap -- output row -> byte offset of the
// tap's input row inside its image, output column -> byte offset of the tap's input pixel inside its row, 0xc0000000 where the tap
// falls outside (reflection folded in) -- for the (at most two: Cb % 64 == 0) taps the workgroup's 128 B columns touch.  A gathered
// row costs 11 VALU instructions and two 4-byte LDS reads per step (advance the (row, column) state with one wrap each, add the two
// entries and the image offset), an A row one add, instead of ~45 with the bounds / reflection tests, three multiplies and a `while`
// loop per row: the kernel spent 12 VALU instructions per MFMA.  An entry of 0xc0000000 (or two: the sum wraps to 0x80000000) and an
// image index outside [0, N) put the offset beyond the buffer: the LDS-DMA's range check supplies the zeros (tensors <= 2^29 bytes on
// this path).  A full position table (wgrad_slab.hip's scheme) was tried first: 3 528 entries built by every workgroup for the 23 K
// steps it runs cost more than they saved, and their 14 KB took the third workgroup off a CU.
constexpr int WD_TBL_DIM = 128;                        // output rows / columns the tables hold
template <int BKP, int NST = 3, bool FUSE_DB = true, bool TBL = false>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradArgs a) {
  typedef __bf16 T;
  constexpr int BT = 128, RS = 256, TT = 2;
  constexpr int NP = BKP / 16;                        // DMA pieces (4 rows) per wave per operand per step
  constexpr int TILE = BKP * RS;                      // 8 KiB per operand
  // NST pipeline stages: 3 = two K steps of LDS-DMA in flight behind the MFMAs; 2 = one (32 KiB LDS: 4 workgroups per CU)
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * TILE];
  __shared__ unsigned ptab[TBL ? 2 * 2 * WD_TBL_DIM : 1];       // [tap slot][rows | columns]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping: workgroups b, b+8, ... share an XCD (and its 4 MiB L2).  All tiles of one (group, split)
  // unit read the SAME pixel rows, so a unit is kept on one XCD: its rows stream into that L2 once and serve
  // every tile of the unit, instead of every XCD streaming every unit from the Infinity Cache.
  const int tiles = a.na_tiles * a.nb_tiles;
  int g, split, tile;
  {
    // the (unit, tile) items in unit-major order are cut into 8 contiguous runs, one per XCD: whole units when the unit
    // count is a multiple of 8, otherwise a unit's tiles are shared by at most two XCDs
    const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
    const int total = tiles * a.groups * a.splitk, per = (total + 7) >> 3;
    const int item = xcd * per + k;
    if (item >= total) return;                          // whole workgroup: filler of the last run
    const int u = item / tiles;
    tile = item - u * tiles;
    g = u / a.splitk; split = u - g * a.splitk;
  }
  const int a_tile = tile % a.na_tiles, b_tile = tile / a.na_tiles;
  const int QQ = a.Qh * a.Qw;
  const unsigned OOB = 0x80000000u;

  // DMA geometry: piece = 4 rows x 256 B; wave w, instruction i (0,1) stages rows (4i + w)*4 .. +3 of each operand
  const int lrow = lane >> 4, pc = lane & 15;
  const int swz = (lrow << 2) | (wave & 3);           // ((row & 3) << 2) | ((row >> 2) & 3) for row = (4i+w)*4 + lrow
  const int c = pc ^ swz;                             // logical 16-byte chunk this lane fetches
  const int a_ch = a_tile * BT + c * 8;
  const bool a_ok = a_ch < a.Ca;
  const int nb = b_tile * BT + c * 8;
  const bool b_ok = nb < a.NB;
  int bt = 0, bc = 0;
  if (b_ok) { bt = nb / a.Cb; bc = nb - bt * a.Cb; }
  const int ti = a.tap[bt];
  const int tdy = (int)(signed char)(ti & 0xff), tdx = (int)(signed char)((ti >> 8) & 0xff);
  const T* Ag = (const T*)a.A + (size_t)g * a.a_gstride;
  const T* Bg = (const T*)a.B + (size_t)g * a.b_gstride;
  const i32x4 ar = s2p_make_rsrc(Ag, a.a_bytes - (unsigned)g * (unsigned)a.a_gstride * 2u);
  const i32x4 br = s2p_make_rsrc(Bg, a.b_bytes - (unsigned)g * (unsigned)a.b_gstride * 2u);

  const int step0 = split * a.steps_per_split;
  int nsteps = a.steps_per_split;
  {
    int total = (a.M + BKP - 1) / BKP;
    if (step0 + nsteps > total) nsteps = total - step0;
  }
  if (nsteps <= 0) return;

  int pm[NP], pn[NP], py[NP], px[NP];
  unsigned noffc[NP], ao[NP];                               // TBL state (see the kernel's header)
  int oyv[NP], oxv[NP];
  unsigned tvy[NP], tvx[NP];
  const int bt0 = (b_tile * BT) / a.Cb;                     // first tap of this workgroup's B columns
  if constexpr (TBL) {
    for (int e = tid; e < 2 * 2 * WD_TBL_DIM; e += 256) {
      const int sl = e / (2 * WD_TBL_DIM), r2 = e - sl * (2 * WD_TBL_DIM), isx = r2 / WD_TBL_DIM, o = r2 - isx * WD_TBL_DIM;
      unsigned v = 0xc0000000u;
      if (bt0 + sl < a.T && o < (isx ? a.Qw : a.Qh)) {
        const int tw = a.tap[bt0 + sl];
        const int lim = isx ? a.Wi : a.Hi;
        int iv = o * a.istride + (int)(signed char)(isx ? ((tw >> 8) & 0xff) : (tw & 0xff));
        if (a.reflect) iv = iv < 0 ? -iv : (iv >= lim ? 2 * lim - 2 - iv : iv);
        if (iv >= 0 && iv < lim) v = (unsigned)iv * (unsigned)((isx ? 1 : a.Wi) * a.b_pitch * 2);
      }
      ptab[e] = v;
    }
    __syncthreads();
  }
  const unsigned* const ytab = ptab + (b_ok && bt > bt0 ? 2 * WD_TBL_DIM : 0);      // (bt - bt0 is 0 or 1: Cb % 64 == 0)
  const unsigned* const xtab = ytab + WD_TBL_DIM;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    int m = step0 * BKP + (4 * i + wave) * 4 + lrow;
    pm[i] = m;
    if constexpr (TBL) {
      const int n = m / QQ, rr = m - n * QQ, qy = rr / a.Qw, qx = rr - qy * a.Qw;
      oyv[i] = qy; oxv[i] = qx;
      noffc[i] = (unsigned)n * (unsigned)(a.Hi * a.Wi * a.b_pitch * 2) + (unsigned)(bc * 2);
      ao[i] = a_ok ? (unsigned)((m * a.a_pitch + a_ch) * 2) : OOB;      // rows beyond M lie beyond the buffer (a_bytes = M * pitch)
      tvy[i] = ytab[qy]; tvx[i] = xtab[qx];
    } else {
      int n = m / QQ, rr = m - n * QQ, qy = rr / a.Qw, qx = rr - qy * a.Qw;
      pn[i] = n; py[i] = qy; px[i] = qx;
    }
  }
  // BKP positions = tadv_n images + tadv_y rows + tadv_x columns (launch-uniform)
  const int tadv_q = BKP / a.Qw, tadv_x = BKP - tadv_q * a.Qw, tadv_n = tadv_q / a.Qh, tadv_y = tadv_q - tadv_n * a.Qh;
  const unsigned img_b = (unsigned)(a.Hi * a.Wi) * (unsigned)(a.b_pitch * 2), a_step = (unsigned)(BKP * a.a_pitch * 2);
  auto issue = [&](int buf) {
    const unsigned base = s2p_lds_addr(smem) + buf * 2 * TILE + wave * (4 * RS);
    if constexpr (TBL) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const unsigned bo = b_ok ? tvy[i] + tvx[i] + noffc[i] : OOB;
        s2p_dma16(ar, base + i * (16 * RS), (int)ao[i]);
        s2p_dma16(br, base + TILE + i * (16 * RS), (int)bo);
        ao[i] += a_step;
        int x = oxv[i] + tadv_x;
        const bool cw = x >= a.Qw;
        x = cw ? x - a.Qw : x;
        int y = oyv[i] + tadv_y + (cw ? 1 : 0);
        const bool rw = y >= a.Qh;
        y = rw ? y - a.Qh : y;
        noffc[i] += (unsigned)tadv_n * img_b + (rw ? img_b : 0u);
        oxv[i] = x; oyv[i] = y;
        tvy[i] = ytab[y]; tvx[i] = xtab[x];                        // consumed a whole step later
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool mok = pm[i] < a.M;
      unsigned ao = (mok && a_ok) ? (unsigned)((pm[i] * a.a_pitch + a_ch) * 2) : OOB;
      int iy = py[i] * a.istride + tdy, ix = px[i] * a.istride + tdx;
      if (a.reflect) {
        iy = iy < 0 ? -iy : (iy >= a.Hi ? 2 * a.Hi - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= a.Wi ? 2 * a.Wi - 2 - ix : ix);
      }
      const bool bok = mok && b_ok && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
      unsigned bo = bok ? (unsigned)((((pn[i] * a.Hi + iy) * a.Wi + ix) * a.b_pitch + bc) * 2) : OOB;
      s2p_dma16(ar, base + i * (16 * RS), (int)ao);
      s2p_dma16(br, base + TILE + i * (16 * RS), (int)bo);
      pm[i] += BKP; px[i] += BKP;
      while (px[i] >= a.Qw) { px[i] -= a.Qw; if (++py[i] >= a.Qh) { py[i] = 0; ++pn[i]; } }
    }
  };

  f32x16 acc[TT][TT];
  f32x16 accb[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < TT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  const int wa0 = (wave >> 1) * (TT * 32), wb0 = (wave & 1) * (TT * 32);
  const bool do_bias = FUSE_DB && a.db != nullptr && b_tile == 0 && wb0 == 0;          // wave-uniform
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

  // transposed-read lane geometry (see wgrad_kernel): group gq = lane>>4 -> channel block 16*(gq&1), k half gq>>1
  const int gq = lane >> 4, gg = gq & 1, hh = gq >> 1, q = (lane >> 2) & 3, p = lane & 3;

  // 3-stage pipeline: while step kt computes, the DMAs of steps kt+1 and kt+2 are in flight.  The only waits are a
  // counted vmcnt (all but the newest stage's 2*NP DMA instructions) and a raw s_barrier -- never vmcnt(0) in the loop.
  issue(0);
  if (NST == 3 && nsteps > 1) issue(1);
  if (NST == 3 && nsteps > 1) S2P_WAIT_VMCNT(2 * NP);
  else S2P_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int stage = 0;
  for (int kt = 0; kt < nsteps; ++kt) {
    int st2 = stage + (NST - 1); if (st2 >= NST) st2 -= NST;
    if (kt + (NST - 1) < nsteps && S2P_DIAGV(a) != 1) issue(st2);
    const char* At = smem + stage * 2 * TILE;
    const char* Bt = At + TILE;
    if (S2P_DIAGV(a) != 2)
#pragma unroll
    for (int s = 0; s < BKP / 16; ++s) {
      s16x4 av[2][TT], bv[2][TT];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = s * 16 + 8 * hh + q + 4 * half;
        const int rs = ((row & 3) << 2) | ((row >> 2) & 3);
        const char* ra = At + row * RS + 8 * (p & 1);
        const char* rb = Bt + row * RS + 8 * (p & 1);
#pragma unroll
        for (int i = 0; i < TT; ++i) {
          const int c0 = (wa0 + 32 * i + 16 * gg) / 8 + (p >> 1);
          av[half][i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ra + 16 * (c0 ^ rs)));
        }
#pragma unroll
        for (int j = 0; j < TT; ++j) {
          const int c0 = (wb0 + 32 * j + 16 * gg) / 8 + (p >> 1);
          bv[half][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(rb + 16 * (c0 ^ rs)));
        }
      }
      bf16x8 af[TT], bf[TT];
#pragma unroll
      for (int i = 0; i < TT; ++i) {
        s16x8 t = {av[0][i][0], av[0][i][1], av[0][i][2], av[0][i][3], av[1][i][0], av[1][i][1], av[1][i][2], av[1][i][3]};
        af[i] = __builtin_bit_cast(bf16x8, t);
        s16x8 u = {bv[0][i][0], bv[0][i][1], bv[0][i][2], bv[0][i][3], bv[1][i][0], bv[1][i][1], bv[1][i][2], bv[1][i][3]};
        bf[i] = __builtin_bit_cast(bf16x8, u);
      }
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int j = 0; j < TT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < TT; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
    }
    // stage kt+1 must have landed (for every wave) before anyone reads it; stage kt+2 may stay in flight
    if (NST == 3 && kt + 2 < nsteps) S2P_WAIT_VMCNT(2 * NP);
    else S2P_WAIT_VMCNT(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);                 // no LDS read may be scheduled above the wait + barrier
    if (++stage == NST) stage = 0;
  }

  const int r = lane & 31, h = lane >> 5;
  float* dWg = a.dW + (size_t)g * a.dw_gstride;
  if (S2P_DIAGV(a) == 3) return;
  if (a.part) {
    // plain stores of the partial tile: for a fixed register the 32 lanes of a half write 32 consecutive columns
    const size_t unit = (size_t)split * a.groups + g;
    float* P = a.part + unit * a.part_rows * a.part_cols;
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int n = b_tile * BT + wb0 + 32 * j + r;
      if (n >= a.NB) continue;
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int arow = a_tile * BT + wa0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (arow < a.Ca) P[(size_t)arow * a.part_cols + n] = acc[i][j][e];
        }
    }
    if (do_bias && r == 0) {
      float* PB = a.part_b + unit * a.part_rows;
#pragma unroll
      for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int arow = a_tile * BT + wa0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (arow < a.Ca) PB[arow] = accb[i][e];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    int n = b_tile * BT + wb0 + 32 * j + r;
    if (n >= a.NB) continue;
    int t = n / a.Cb, cch = n - t * a.Cb;
    if (cch >= a.Cb_real) continue;
    int col = t * a.Cb_real + cch;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int arow = a_tile * BT + wa0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (arow < a.Ca_real) atomicAdd(dWg + (size_t)arow * a.dw_row + col, acc[i][j][e]);
      }
  }
  if (do_bias && r == 0) {
    float* dbg = a.db + (size_t)g * a.Ca_real;
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int arow = a_tile * BT + wa0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (arow < a.Ca_real) atomicAdd(dbg + arow, accb[i][e]);
      }
  }
}

// dW[g][arow][t][c] += sum over the split units' partial tiles, db likewise, in unit order (no atomics: bitwise
// reproducible).  64 outputs per workgroup; with many units (NSEG = 16) an output's partials are summed by 16 threads (a
// sixteenth each, 8 loads in flight) and combined in segment order.
template <int NSEG>
__global__ __launch_bounds__(64 * NSEG) void wgrad_part_reduce_kernel(const WgradArgs a, int n_out, int n_bias) {
  __shared__ float red[64 * NSEG];
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long long idx = (long long)blockIdx.x * 64 + o;           // [0, n_out): dW elements, then [n_out, n_out + n_bias): db
  const int S = a.splitk;
  const int per = (S + NSEG - 1) / NSEG;
  const int k0 = sg * per, k1 = k0 + per < S ? k0 + per : S;
  float s = 0.f;
  float* dst = nullptr;
  if (idx < (long long)n_out + n_bias) {
    const float* p; size_t stride;
    if (idx < n_out) {
      const int per_g = a.Ca_real * a.dw_row;
      const int g = (int)(idx / per_g), rem = (int)(idx - (long long)g * per_g);
      const int arow = rem / a.dw_row, col = rem - arow * a.dw_row;
      const int t = col / a.Cb_real, c = col - t * a.Cb_real;
      p = a.part + ((size_t)g * a.part_rows + arow) * a.part_cols + t * a.Cb + c;
      stride = (size_t)a.groups * a.part_rows * a.part_cols;
      dst = a.dW + (size_t)g * a.dw_gstride + (size_t)arow * a.dw_row + col;
    } else {
      const int b = (int)(idx - n_out), g = b / a.Ca_real, arow = b - g * a.Ca_real;
      p = a.part_b + (size_t)g * a.part_rows + arow;
      stride = (size_t)a.groups * a.part_rows;
      dst = a.db + b;
    }
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += p[(size_t)k * stride];
  }
  if (NSEG == 1) { if (dst) *dst += s; return; }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && dst) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NSEG; ++k) tot += red[64 * k + o];
    *dst += tot;
  }
}

// split count of the LDS-DMA kernels: enough (unit, tile) workgroups to fill the chip; every extra unit costs one more
// partial tile set to store and to reduce
static void wgrad_dma_splits(WgradArgs& a, int groups, int target) {
  const int tiles = a.na_tiles * a.nb_tiles;
  const int total = cdiv(a.M, 32);
  int U = cdiv(target, tiles);
  if (U < 1) U = 1;
  int sk = cdiv(U, groups);
  if (sk > total) sk = total;
  if (sk < 1) sk = 1;
  a.steps_per_split = cdiv(total, sk); a.splitk = cdiv(total, a.steps_per_split);
  a.groups = groups;
}

static bool wgrad_dma_ok(const s2p_conv_desc* d) {
  if (d->dtype != S2P_BF16) return false;
  const long long pa = d->transposed ? (long long)d->N * d->H * d->W * d->x_pitch : (long long)d->N * d->Ho * d->Wo * d->y_pitch;
  const long long pb = d->transposed ? (long long)d->N * d->Ho * d->Wo * d->y_pitch : (long long)d->N * d->H * d->W * d->x_pitch;
  return pa * 2 < (1ll << 31) && pb * 2 < (1ll << 31);
}

// fills the geometry fields shared by the launch and the workspace query; returns false for shapes outside the DMA kernels
static bool wgrad_dma_plan(const s2p_conv_desc* d, int cin_real, int cout_real, WgradArgs& a, bool& dense) {
  const int ce = 8;
  const int T = d->KH * d->KW;
  const int cout_pad = (d->Cout + ce - 1) / ce * ce;
  a.istride = d->stride; a.reflect = d->reflect; a.T = T;
  if (!d->transposed) {        // A = dY on the output grid, B = X gathered
    a.Qh = d->Ho; a.Qw = d->Wo; a.M = d->N * d->Ho * d->Wo;
    a.Ca = cout_pad; a.a_pitch = d->y_pitch; a.a_gstride = d->y_gstride; a.Ca_real = cout_real;
    a.Hi = d->H; a.Wi = d->W; a.Cb = d->Cin; a.b_pitch = d->x_pitch; a.b_gstride = d->x_gstride;
    a.Cb_real = cin_real;
  } else {                     // A = X on the input grid, B = dY gathered at iy*s + ky - pad
    a.Qh = d->H; a.Qw = d->W; a.M = d->N * d->H * d->W;
    a.Ca = d->Cin; a.a_pitch = d->x_pitch; a.a_gstride = d->x_gstride; a.Ca_real = cin_real;
    a.Hi = d->Ho; a.Wi = d->Wo; a.Cb = cout_pad; a.b_pitch = d->y_pitch; a.b_gstride = d->y_gstride;
    a.Cb_real = cout_real;
  }
  a.NB = T * a.Cb; a.dw_row = T * a.Cb_real;
  a.na_tiles = cdiv(a.Ca, 128); a.nb_tiles = cdiv(a.NB, 128);
  dense = a.a_pitch <= 1024 && a.b_pitch <= 1024;
  if (!wgrad_dma_ok(d) || s2p_env_set("S2P_NO_LDS_DMA")) return false;
  if (!dense && s2p_env_set("S2P_NO_WGRAD_WIDE_DMA")) return false;
  wgrad_dma_splits(a, d->groups, dense ? s2p_env_int("S2P_WGRAD_BLOCKS", 512) : 512);
  a.part_rows = a.na_tiles * 128; a.part_cols = a.nb_tiles * 128;
  return true;
}

static inline size_t ws_align(size_t b) { return (b + 255) / 256 * 256; }
// scratch of the K-split partial tiles of the LDS-DMA kernels (0: no split, or another kernel runs the layer)
static size_t wgrad_dma_ws_bytes(const s2p_conv_desc* d, int cin_real, int cout_real, bool* dense_out) {
  WgradArgs a{};
  bool dense = true;
  if (!wgrad_dma_plan(d, cin_real, cout_real, a, dense)) return 0;
  if (dense_out) *dense_out = dense;
  if (a.splitk * a.groups <= 1) return 0;
  const size_t units = (size_t)a.splitk * a.groups;
  return units * a.part_rows * ((size_t)a.part_cols + 1) * sizeof(float);
}
// bias gradient as a separate channel-sum pass: its scratch sits behind the weight-gradient scratch
static size_t wgrad_bias_ws_bytes(const s2p_conv_desc* d, int cout_real) {
  return s2p_channel_sum_ws_bytes((int64_t)d->N * d->Ho * d->Wo, cout_real * d->groups);
}

extern "C" size_t s2p_conv2d_wgrad_workspace(const s2p_conv_desc* d, int cin_real, int cout_real) {
  if (!d || d->dtype != S2P_BF16 || d->KH * d->KW > 64) return 0;
  if (s2p_thin_applicable(d) && cout_real == d->Cout) {
    const size_t t = s2p_thin_wgrad_ws_bytes(d, cin_real);
    return t ? ws_align(t) + wgrad_bias_ws_bytes(d, cout_real) : 0;
  }
  // strided / 4x4 layers: padded-raster slab kernel (wgrad_slabg.hip); the scratch serves either kernel
  size_t sg = s2p_wgrad_slabg_workspace(d, cin_real, cout_real);
  if (s2p_stem_wgrad_applicable(d, cin_real, cout_real)) { const size_t t = s2p_stem_wgrad_ws_bytes(d, cin_real); if (t > sg) sg = t; }
  bool dense = true;
  WgradArgs a{};
  if (!wgrad_dma_plan(d, cin_real, cout_real, a, dense)) return sg;
  const size_t w = wgrad_dma_ws_bytes(d, cin_real, cout_real, nullptr);
  const size_t g = dense ? w : ws_align(w) + wgrad_bias_ws_bytes(d, cout_real);      // dense: fused bias gradient
  return g > sg ? g : sg;
}

extern "C" int s2p_conv2d_wgrad(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db,
                                int cin_real, int cout_real, int64_t dw_gstride, int splitk, void* stream) {
  return s2p_conv2d_wgrad_ws(d, x, dy, dw, db, cin_real, cout_real, dw_gstride, splitk, nullptr, 0, stream);
}

extern "C" int s2p_conv2d_wgrad_ws(const s2p_conv_desc* d, const void* x, const void* dy, float* dw, float* db,
                                   int cin_real, int cout_real, int64_t dw_gstride, int splitk, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (!d || !x || !dy || !dw) S2P_FAIL(-1, "s2p_conv2d_wgrad: null pointer");
  if (d->dtype != S2P_F32 && d->dtype != S2P_BF16) S2P_FAIL(-1, "s2p_conv2d_wgrad: bad dtype");
  const int ce = d->dtype == S2P_F32 ? 4 : 8;
  if (d->Cin % ce || d->x_pitch % ce || d->y_pitch % ce || d->x_gstride % ce || d->y_gstride % ce)
    S2P_FAIL(-1, "s2p_conv2d_wgrad: channel counts / pitches must be multiples of %d", ce);
  const int T = d->KH * d->KW;
  if (T > 64) S2P_FAIL(-2, "s2p_conv2d_wgrad: more than 64 taps");
  const int cout_pad = (d->Cout + ce - 1) / ce * ce;
  if (cout_pad > d->y_pitch) S2P_FAIL(-1, "s2p_conv2d_wgrad: dy pitch < padded Cout");
  if (db && d->transposed) S2P_FAIL(-1, "s2p_conv2d_wgrad: bias gradient is undefined for the transposed form");
  if (s2p_thin_applicable(d) && cout_real == d->Cout) {
    // with the scratch of s2p_conv2d_wgrad_workspace: per-workgroup partial tiles / partial channel sums + fixed-order
    // reduces (no atomics); without it fp32 atomics
    const size_t tw = s2p_thin_wgrad_ws_bytes(d, cin_real);
    const bool det = tw > 0 && workspace && workspace_bytes >= s2p_conv2d_wgrad_workspace(d, cin_real, cout_real);
    if (db) {
      int rc = s2p_channel_sum_det(d->dtype, dy, (int64_t)d->N * d->Ho * d->Wo, cout_real, d->y_pitch, db,
                                   det ? (char*)workspace + ws_align(tw) : nullptr, det ? wgrad_bias_ws_bytes(d, cout_real) : 0, stream);
      if (rc) return rc;
    }
    return s2p_thin_wgrad(d, x, dy, dw, cin_real, det ? workspace : nullptr, det ? tw : 0, (hipStream_t)stream);
  }
  hipStream_t st = (hipStream_t)stream;
  if (workspace && s2p_stem_wgrad_applicable(d, cin_real, cout_real) && workspace_bytes >= s2p_stem_wgrad_ws_bytes(d, cin_real))
    return s2p_stem_wgrad(d, x, dy, dw, db, cin_real, workspace, workspace_bytes, st);
  if (workspace && s2p_wgrad_slabg_supported(d, cin_real, cout_real) &&
      workspace_bytes >= s2p_wgrad_slabg_workspace(d, cin_real, cout_real))
    return s2p_wgrad_slabg(d, x, dy, dw, db, cin_real, cout_real, workspace, workspace_bytes, st);
  {
    // bf16, tensors < 2 GiB: LDS-DMA kernels.  With a workspace (s2p_conv2d_wgrad_workspace) the split units store
    // partial tiles and a second kernel adds them in unit order: no atomics, bitwise reproducible, and no 16-fold
    // read-modify-write traffic on dW; without one the units add to dW with fp32 atomics.
    WgradArgs a{};
    bool dense = true;
    if (d->dtype == S2P_BF16 && wgrad_dma_plan(d, cin_real, cout_real, a, dense)) {
      a.dW = dw; a.dw_gstride = dw_gstride;
      a.A = d->transposed ? x : dy; a.B = d->transposed ? dy : x;
      for (int ky = 0; ky < d->KH; ++ky)
        for (int kx = 0; kx < d->KW; ++kx)
          a.tap[ky * d->KW + kx] = (((kx - d->pad) & 0xff) << 8) | ((ky - d->pad) & 0xff);
      a.a_bytes = (unsigned)((long long)d->N * a.Qh * a.Qw * a.a_pitch * 2);
      a.b_bytes = (unsigned)((long long)d->N * a.Hi * a.Wi * a.b_pitch * 2);
      static const int diag = s2p_env_int("S2P_DIAG", 0);
      a.diag = diag;
      // Large-pitch (grouped / channel-sliced) operands: these launches are L2-latency-bound and live on occupancy, so
      // they take the 2-stage DMA kernel without the fused bias accumulators (126 VGPRs, 32 KiB LDS: 4 workgroups per CU;
      // the 3-stage + fused-bias form is 146 VGPRs / 48 KiB = 3 per CU and measured 930 us on the 12-group gamma/beta
      // wgrad, the register-staged kernel 670 us, this form 521 us).  The bias gradient is a separate channel-sum pass.
      const size_t wneed = wgrad_dma_ws_bytes(d, cin_real, cout_real, nullptr);
      const size_t need = s2p_conv2d_wgrad_workspace(d, cin_real, cout_real);
      const bool have_ws = workspace && need > 0 && workspace_bytes >= need;
      if (!dense && db) {
        int rc = s2p_channel_sum_det(d->dtype, dy, (int64_t)d->N * d->Ho * d->Wo, cout_real * d->groups, d->y_pitch, db,
                                     have_ws ? (char*)workspace + ws_align(wneed) : nullptr, have_ws ? wgrad_bias_ws_bytes(d, cout_real) : 0,
                                     stream);
        if (rc) return rc;
      }
      a.db = dense ? db : nullptr;              // dense: fused bias gradient (groups: db is [groups][Cout])
      const int units = a.groups * a.splitk;
      const bool det = units > 1 && have_ws && wneed > 0;
      if (det) {
        a.part = (float*)workspace;
        a.part_b = a.part + (size_t)units * a.part_rows * a.part_cols;
      }
      const int tiles = a.na_tiles * a.nb_tiles;
      dim3 grid1(8 * cdiv((long long)tiles * units, 8));
      // gathered-operand offsets from LDS tables (see the kernel): two taps per B tile at most, an output grid the tables hold,
      // tensors below 2^29 bytes; switch 19 of the diagnostics build keeps the (n, row, column) state
      const bool tbl = dense && a.Cb % 64 == 0 && a.Qh <= WD_TBL_DIM && a.Qw <= WD_TBL_DIM && a.a_bytes <= (1u << 29) && a.b_bytes <= (1u << 29) &&
                       !S2P_DIAG_SWITCH(19);
      if (tbl) hipLaunchKernelGGL((wgrad_dma_kernel<32, 3, true, true>), grid1, dim3(256), 0, st, a);
      else if (dense) hipLaunchKernelGGL(wgrad_dma_kernel<32>, grid1, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((wgrad_dma_kernel<32, 2, false>), grid1, dim3(256), 0, st, a);
      S2P_CHECK_LAUNCH("wgrad_dma_kernel");
      if (det) {
        const long long n_out = (long long)a.groups * a.Ca_real * a.dw_row;
        const int n_bias = a.db ? a.groups * a.Ca_real : 0;
        if (n_out >= (1ll << 31)) S2P_FAIL(-2, "s2p_conv2d_wgrad: dW too large");
        const int nblk = cdiv(n_out + n_bias, 64);
        if (a.splitk > 128) hipLaunchKernelGGL(wgrad_part_reduce_kernel<16>, dim3(nblk), dim3(1024), 0, st, a, (int)n_out, n_bias);
        else if (a.splitk > 32) hipLaunchKernelGGL(wgrad_part_reduce_kernel<4>, dim3(nblk), dim3(256), 0, st, a, (int)n_out, n_bias);
        else hipLaunchKernelGGL(wgrad_part_reduce_kernel<1>, dim3(nblk), dim3(64), 0, st, a, (int)n_out, n_bias);
        S2P_CHECK_LAUNCH("wgrad_part_reduce_kernel");
      }
      return 0;
    }
  }
  // fp32 (exact MFMA path) and bf16 tensors beyond the buffer-descriptor range: register-staged kernel, fp32 atomics
  WgradArgs a{};
  a.dW = dw; a.db = nullptr;
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
  if (db) {
    int rc = s2p_channel_sum(d->dtype, dy, (int64_t)d->N * d->Ho * d->Wo, cout_real * d->groups, d->y_pitch, db, stream);
    if (rc) return rc;
  }
  if (d->dtype == S2P_F32) hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(wgrad_kernel<__bf16>, grid, dim3(256), 0, st, a);
  S2P_CHECK_LAUNCH("wgrad_kernel");
  return 0;
}
