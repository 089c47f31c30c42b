// Plane-resident 3x3 "same" convolution for small feature maps (the 21x21 ResBlk / VGG conv3 / gamma-beta layers and their
// stride-1 dgrads), bf16, NHWC, gfx950.
//
// Decomposition: ONE workgroup = one image x one 64-channel slab of the output (N * Cout/64 workgroups: 256 for the
// 64 x 256-channel ResBlk conv = one per CU, no tile quantisation) and it keeps the image's WHOLE zero-padded input plane
// of the current 32-channel half-slab in LDS:
//   * the plane is stored on a padded raster, position p(y,x) = (y+1)*WP + x + 1 with one zero column per row and a zero
//     row above / below, so a tap (dy,dx) is the constant position shift dy*WP + dx -- no validity masks, no per-tap
//     address arithmetic: every fragment read is  base_register + immediate;
//   * LDS images are [16-ch pair][position or co][2 x 16 B]: 32-byte rows.  For a 16-lane group of ds_read_b128 (8 rows
//     at k-chunk q, 8 rows at q+1: the v_mfma_f32_16x16x32_bf16 operand map) consecutive rows then fall on 16 distinct
//     16-B slots for ANY start row, i.e. the reads are conflict-free at every tap shift without an XOR swizzle;
//   * 8 waves = 2 per SIMD.  Waves w and w+4 own the SAME 64 co x 112 px output tile (28 accumulator tiles of 16x16) and
//     split K: wave set 0 takes the even K steps (one step = one tap x 32 channels = 28 MFMAs), set 1 the odd ones, so one
//     wave's LDS-DMA issue, fragment reads and barrier wait run under its partner's MFMAs; the two partial accumulators are
//     added through LDS once, after the loop.  One s_barrier per PAIR of K steps (~900 MFMA cycles per SIMD);
//   * weights stream through a 6-stage ring of 4-KiB stages (one stage = one K step), two pair-steps ahead, by LDS-DMA
//     (buffer_load ... lds) with counted vmcnt; the next half-slab's plane is fetched while the current one is swept.
// Per workgroup and K step the LDS carries 11 KiB of fragment reads for 28 MFMAs (the 128x128-tile kernels: 16 KiB for
// 16 larger MFMAs of twice the cycles) and 4 KiB of weight writes for 448 px (there: 16 KiB for 128 px).
#include "s2p_common.h"
#include "conv_plane.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4v;

template <int B, int E, typename F>
__device__ __forceinline__ void pl_static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); pl_static_for<B + 1, E>(f); }
}

// DMA with the uniform part of the source offset in an SGPR (soffset).  An invalid lane carries voffset = 0x80000000 and returns
// zeros whatever the scalar part is.  The scalar part COUNTS in the range check on gfx950 (measured in round 4: a group offset
// beyond num_records in soffset zero-filled valid lanes), so num_records must cover base + voffset + soffset of every valid lane:
// here the scalar part is a channel offset inside a pixel row / a tap offset inside a weight row, both inside the records.
__device__ __forceinline__ void pl_dma16(i32x4 rsrc, unsigned lds_dst, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :: "v"(voffset), "s"(rsrc), "s"(lds_dst), "s"(soffset) : "memory", "m0");
}

namespace {
constexpr int PL_NPOS = 512;                 // padded-raster positions per plane buffer
constexpr int PL_CPS = PL_NPOS * 32;         // bytes between the two 16-channel pairs of a half-slab
constexpr int PL_PBUF = 2 * PL_CPS;          // one half-slab plane buffer (32 KiB)
constexpr int PL_WST = 4096;                 // one weight stage: [2 pairs][64 co][32 B]
constexpr int PL_RING = 6;
constexpr int PL_ERS = 144;                  // epilogue staging row: 64 co x 2 B + 16
}

// DIAG: timing ablations, instantiated only in the diagnostics build; a bit mask: 1 no in-loop DMA, 2 no MFMAs, 4 no fragment
// reads, 8 no K loop, 16 clock stamps around the K loop, 64 coalesced (wrong) DMA sources; outputs are invalid for DIAG != 0.
// MAT: the epilogue also InstanceNorm-alises the output plane it owns and applies the MAT / SPADE modulation + activation
// (one more output tensor + the statistics buffer of norm.hip): conv -> IN -> modulate -> LeakyReLU in one launch.
// GST (with MAT 1 / 2, gamma|beta maps present, Cin >= 256): the workgroup's gamma|beta rows -- the one operand of the norm tail
// that no wave needs before the tail -- are fetched by LDS-DMA UNDER the K loop instead of by register loads behind it: the tail
// then only computes and stores (the reads were 2/3 of the forward tail's HBM traffic and arrived cold, with every MFMA idle).
// No VGPR is spent on them (the two register-prefetch forms of round 3 spilled or only warmed L2).  The whole 160 KB are used:
//   [0, 32K)    plane buffer 0; from pair-step 4 of the LAST iteration (its last fragment read is in pair-step 3): rows 288..383
//   [32K, 88K)  plane buffer 1 + weight ring; behind the loop: accumulator exchange in TWO rounds of 56 KB (one co block per wave
//               and round), then the [pixel][co] staging rows at a 128-byte pitch with an XOR chunk swizzle, then reduce scratch
//   [88K, 160K) gamma|beta rows 0..287, fetched during the three iterations before the last one (one piece per wave in
//               pair-steps 2, 3, 4)
// A row is 256 B, [gamma 128 B | beta 128 B] on even rows and [beta | gamma] on odd ones (the DMA's per-lane source address does
// the swap): the tail's 16-lane ds_read_b128 groups then touch 16 distinct 16-byte slots.  Rows 384..440 (the seventh row of
// each thread) still come through registers, loaded right behind the loop.  DMA accounting: the pieces are younger than every
// weight / plane piece a later pair-step waits for only in the last iteration's tail, where the counted waits let them stay in
// flight until the tail's first use (one vmcnt(0) + barrier there).
template <int PB, int WP, int DIAG, int MAT = 0, int GST = 0>   // MAT: 0 plain conv, 1 + forward norm, 2 + backward of the norm that FED this dgrad's forward conv
__global__ __launch_bounds__(512) void conv_plane_kernel(const PlaneArgs a) {
  typedef __bf16 T;
  static_assert(GST == 0 || (MAT != 0 && DIAG == 0 && PB == 7), "gamma|beta staging: fused-norm product kernels only");
  constexpr int BPIX = 4 * PB * 16;
  constexpr int NT = 4 * PB;                                   // accumulator tiles per wave
  constexpr int MERGE = 4 * NT * 1024;
  constexpr int MAIN = 2 * PL_PBUF + PL_RING * PL_WST;
  constexpr int EPI = BPIX * PL_ERS;
  constexpr int SMEM = GST ? 163840 : (MERGE > MAIN ? (MERGE > EPI ? MERGE : EPI) : (MAIN > EPI ? MAIN : EPI));
  constexpr int XOFF = GST ? PL_PBUF : 0;                      // GST: exchange / staging / scratch region = buffer 1 + ring (56 KB)
  constexpr int GB_B = MAIN;                                   // GST: gamma|beta rows 0..287
  constexpr int GB_ROWS_B = (163840 - MAIN) / 256, GB_ROWS = GB_ROWS_B + 96;      // 288, 384 (= 6 x 64 rows; 96 rows in plane buffer 0)
  static_assert(!GST || (GB_ROWS_B == 288 && BPIX * 128 == MAIN - PL_PBUF), "LDS map of the staged form");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* const pbase = smem;
  char* const wbase = smem + 2 * PL_PBUF;

  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};               // DIAG & 128: s_memrealtime (100 MHz) at the phase boundaries -> aux
  if constexpr ((DIAG & 128) != 0) ph[0] = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = wave >> 2, wq = wave & 3;
  const int q = lane >> 4, l15 = lane & 15;
  const int g = blockIdx.y;
  // workgroup -> (image, co slab): the nco slabs of one image run back to back on ONE XCD (blocks b, b+8, ... share an
  // XCD under the observed round-robin placement: speed only), so the image's plane comes from HBM once per XCD.
  int img, cs;
  {
    const int bid = blockIdx.x, nco = a.nco;
    if ((a.N & 7) == 0) { const int xcd = bid & 7, k = bid >> 3; cs = k % nco; img = (k / nco) * 8 + xcd; }
    else { cs = bid % nco; img = bid / nco; }
  }
  const int co_base = cs * 64;
  const int HW = a.H * a.W;

  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  const i32x4 xrs = s2p_make_rsrc(xg, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u);
  const i32x4 wrs = s2p_make_rsrc(wg, a.w_bytes);
  const unsigned p_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(pbase));
  const unsigned w_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(wbase));
  const unsigned OOB = 0x80000000u;

  // ---- DMA source offsets ---------------------------------------------------------------------------------------
  // plane: 32 pieces per half-slab (piece = 32 positions x one 16-channel pair); wave w issues pieces w, w+8, w+16, w+24
  int hv[4]; unsigned hdst[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ii = wave + 8 * k;
    const int cp = ii & 1, pg = ii >> 1;
    const int pos = 32 * pg + (lane >> 1);
    const int pm = pos - 1;
    const int yy = pm / WP - 1, xx = pm % WP;
    const bool ok = pos >= 1 && yy >= 0 && yy < a.H && xx < a.W;
    hv[k] = ok ? (int)((((unsigned)(img * a.H + yy) * a.W + xx) * a.x_pitch) * 2u + (2 * cp + (lane & 1)) * 16) : (int)OOB;
    hdst[k] = (unsigned)(cp * PL_CPS + pg * 1024);
    if constexpr ((DIAG & 64) != 0) hv[k] = (int)(((unsigned)img * a.H * a.W * a.x_pitch) * 2u + ii * 1024 + lane * 16);   // timing only: coalesced source
  }
  // weights: 4 pieces per stage (piece = 32 co x one pair); wave w issues piece (w & 3) of the stage its set consumes
  int wv; unsigned wdst;
  {
    const int cp = wave & 1, cohalf = (wave >> 1) & 1;
    const int co = co_base + 32 * cohalf + (lane >> 1);
    wv = co < a.Cout ? (int)((unsigned)co * a.w_row * 2u + (2 * cp + (lane & 1)) * 16) : (int)OOB;
    wdst = (unsigned)(cp * 2048 + cohalf * 1024);
    if constexpr ((DIAG & 64) != 0) wv = (int)((unsigned)co_base * a.w_row * 2u + (wave & 3) * 1024 + lane * 16);            // timing only: coalesced source
  }
  int wto[9];                                                   // byte offset of geometric tap t inside a packed weight row
#pragma unroll
  for (int t = 0; t < 9; ++t) wto[t] = a.wt[t] * a.Cin * 2;

  // GST: gamma|beta pieces.  Piece pi = rows 4 pi .. 4 pi + 3; lane l lands at byte 16 l of the piece: row l >> 4, 128-byte half
  // (l >> 3) & 1, chunk l & 7; the half holds beta iff half != row parity.  Wave w issues pieces w + 8 j.
  i32x4 grs = xrs; int gvo = 0;
  if constexpr (GST != 0) {
    grs = s2p_make_rsrc(a.gb, (unsigned)a.N * (unsigned)HW * (unsigned)a.gb_pitch * 2u);
    const int rl = lane >> 4, isb = ((lane >> 3) ^ rl) & 1;
    gvo = (int)((((unsigned)img * HW + rl) * a.gb_pitch + isb * a.Cout + co_base + (lane & 7) * 8) * 2u);
  }
  auto issue_g = [&](int pi) {
    if constexpr (GST != 0) {
      const unsigned dst = pi < GB_ROWS_B / 4 ? (unsigned)(GB_B + pi * 1024) : (unsigned)((pi - GB_ROWS_B / 4) * 1024);
      pl_dma16(grs, p_lds + dst, gvo, pi * 4 * a.gb_pitch * 2);     // (host: H * W >= 384, every staged row exists)
    }
  };
  const int nhs = a.Cin / 32;                                   // half-slabs (host guarantees Cin % 64 == 0)
  auto issue_plane2 = [&](int buf, int hs, int k0) {            // two of this wave's four pieces of half-slab hs
    const int so = hs * 64;
#pragma unroll
    for (int k = k0; k < k0 + 2; ++k)
      pl_dma16(xrs, p_lds + (unsigned)(buf * PL_PBUF) + hdst[k], hs < nhs ? hv[k] : (int)OOB, so);
  };
  auto issue_w = [&](int stage, int tap_off, int hs) {
    pl_dma16(wrs, w_lds + (unsigned)(stage * PL_WST) + wdst, hs < nhs ? wv : (int)OOB, tap_off + hs * 64);
  };

  // ---- prologue: plane of half-slab 0 + the first three weight stages of this wave's set --------------------------------
  issue_plane2(0, 0, 0);
  issue_plane2(0, 0, 2);
  issue_w(set, set ? wto[1] : wto[0], 0);                       // K steps set, set + 2, set + 4 (taps of half-slab 0)
  issue_w(set + 2, set ? wto[3] : wto[2], 0);
  issue_w(set + 4, set ? wto[5] : wto[4], 0);
  // ---- fragment read bases (computed while the prologue DMA is in flight) ---------------------------------------------
  int bB[PB];
  {
    const float rw = 1.0f / (float)a.W;                        // exact floor(m / W) for m < 512: (m + 0.5) / W is >= 0.5 / W away from an integer
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      int m = (wq * PB + j) * 16 + l15;
      if (m >= HW) m = HW - 1;                                 // padding columns of the last block: computed, never stored
      const int y = (int)(((float)m + 0.5f) * rw), x = m - y * a.W;
      bB[j] = (q >> 1) * PL_CPS + (q & 1) * 16 + (y * WP + x) * 32;
    }
  }
  const int bA = (q >> 1) * 2048 + l15 * 32 + (q & 1) * 16;
  int bAw = (int)(2 * PL_PBUF) + bA;                            // ring base + lane part: opaque, so that every fragment read is this
  if constexpr (GST != 0) {                                     // register + a 16-bit immediate (the three loop bodies of the staged
    asm volatile("" : "+v"(bAw));                               // form otherwise hoist one address register per (stage, block))
#pragma unroll
    for (int j = 0; j < PB; ++j) asm volatile("" : "+v"(bB[j]));
  }
  f32x4v acc[4][PB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

  // Software pipeline, identical in both wave sets: in pair-step U a wave runs the 28 MFMAs of its K step u on fragments
  // that are already in registers, and BETWEEN them issues the 11 fragment reads of its next step (u + 2) and its LDS-DMA
  // pieces (weights of step u + 6 into the ring stage step u just released; its share of the next plane).  The MFMAs of
  // the two waves of a SIMD interleave on the matrix pipe, so each wave has a ~16-cycle slot after each of its MFMAs for one
  // of those instructions.  Set 0 runs the even K steps, set 1 the odd ones (ring stages of the same parity).
  auto read_a = [&](auto uc, auto ic, bf16x8 (&fa)[4]) {
    constexpr int u = decltype(uc)::value % 18, i = decltype(ic)::value;
    if constexpr ((DIAG & 4) == 0) fa[i] = *(const bf16x8*)(smem + (u % PL_RING) * PL_WST + i * 512 + bAw);
  };
  auto read_b = [&](auto uc, auto jc, bf16x8 (&fb)[PB]) {
    constexpr int u = decltype(uc)::value % 18, j = decltype(jc)::value;
    constexpr int t = u % 9, hsl = u / 9;
    if constexpr ((DIAG & 4) == 0) fb[j] = *(const bf16x8*)(pbase + hsl * PL_PBUF + ((t / 3) * WP + (t % 3)) * 32 + bB[j]);
  };
  auto mfma4 = [&](auto jc, bf16x8 (&fa)[4], bf16x8 (&fb)[PB]) {
    constexpr int j = decltype(jc)::value;
    if constexpr ((DIAG & 2) != 0) {                            // timing ablation: no MFMAs (the operands stay live)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i]));
      asm volatile("" :: "v"(fb[j]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  // top of pair-step U: everything this wave issued before the previous pair-step has landed (what it issued IN the previous
  // pair-step -- one weight piece, plus two plane pieces after U = 0, 1, 5, 6 -- may still be in flight), its own LDS
  // reads have returned (another wave's DMA may overwrite what they read once the barrier is passed), then the barrier
  // GST: ONE loop body and NO branch in it (a second copy of the body made the register allocator spill accumulators, wave-uniform
  // branches inside it made it spill the DMA offsets and reload them behind vmcnt(0)); what differs between iterations is chosen
  // by scalar selects, and every iteration issues the same number of DMAs per pair-step, so the counted waits are compile-time
  // constants.  Every iteration issues one gamma|beta piece per wave in pair-steps 2, 3, 4: rows 0..287 over the three iterations
  // before the last; earlier iterations and the last one re-issue a piece that is already there (same bytes to the same place).
  // In the LAST iteration the four plane slots of pair-steps 5, 6 -- dummies without the staging, which would zero plane buffer 0
  // -- carry rows 288..383 into that buffer (three pieces + one re-issue).
  auto sync_top = [&](auto Uc) {
    constexpr int UP = (decltype(Uc)::value + 8) % 9;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (UP == 0 || UP == 1 || UP == 5 || UP == 6) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (GST != 0 && (UP == 2 || UP == 3 || UP == 4)) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  const int nloop = (DIAG & 8) ? 0 : nhs;
  unsigned long long st_c0 = 0, st_r0 = 0;                      // DIAG & 16: shader-clock / 100 MHz stamps around the K loop -> y
  auto run = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
    typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
    typedef std::integral_constant<int, 6> I6;
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");            // plane 0 and the stage of the first step have landed ...
    __builtin_amdgcn_s_barrier();                               // ... for every wave
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fa[4], fb[PB];                                       // fragments of the step being computed
    {
      typedef std::integral_constant<int, SET> uc;
      if constexpr ((DIAG & 4) != 0) {                          // timing ablation: no fragment reads (lane-dependent junk operands)
        const bf16x8 junk = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)bA * 2654435761u, 0x3f80bf80u, 0x3f803f80u, (unsigned)lane * 97u + 0x3f000000u});
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = junk;
#pragma unroll
        for (int j = 0; j < PB; ++j) fb[j] = junk;
      }
      pl_static_for<0, 4>([&](auto ic) { read_a(uc{}, ic, fa); });
      pl_static_for<0, PB>([&](auto jc) { read_b(uc{}, jc, fb); });
    }
    if constexpr ((DIAG & 16) != 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    if constexpr ((DIAG & 128) != 0) ph[1] = __builtin_amdgcn_s_memrealtime();
    for (int k2 = 0; k2 < nloop; k2 += 2) {                     // two half-slabs = 18 K steps = 9 pair-steps per iteration
      const int left = (nloop - k2) >> 1;                       // GST: iterations left, this one included (host: at least 4 in all)
      const bool last = GST != 0 && left == 1;
      int gj = (4 - left) * 3;                                  // GST: first of this iteration's three pieces w + 8 j (clamped: a re-issue)
      gj = gj < 0 ? 0 : (gj > 6 ? 6 : gj);
      // GST, pair-steps 5 / 6: plane piece k of the next half-slab, or (last iteration) gamma|beta piece 72 + w + 8 min(k, 2)
      auto issue_p0 = [&](int k) {
        if constexpr (GST != 0) {
          const int pi = GB_ROWS_B / 4 + wave + 8 * (k > 2 ? 2 : k);
          i32x4 rs;
#pragma unroll
          for (int c = 0; c < 4; ++c) rs[c] = last ? grs[c] : xrs[c];
          pl_dma16(rs, p_lds + (last ? (unsigned)((pi - GB_ROWS_B / 4) * 1024) : hdst[k]), last ? gvo : hv[k],
                   last ? pi * 4 * a.gb_pitch * 2 : (k2 + 2) * 64);
        }
      };
      pl_static_for<0, 9>([&](auto Uc) {
        constexpr int U = decltype(Uc)::value;
        constexpr int u = 2 * U + SET;
        typedef std::integral_constant<int, u + 2> un;           // the step whose fragments are fetched now
        sync_top(Uc);
        bf16x8 na[4], nb[PB];
        if constexpr ((DIAG & 4) != 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) na[i] = fa[i];
#pragma unroll
          for (int j = 0; j < PB; ++j) nb[j] = fb[j];
        }
        // seven groups of four MFMAs; the pinned order puts one or two LDS / DMA instructions behind each group
        read_a(un{}, I0{}, na); read_a(un{}, I1{}, na);
        mfma4(I0{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        read_a(un{}, I2{}, na); read_a(un{}, I3{}, na);
        mfma4(I1{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        read_b(un{}, I0{}, nb); read_b(un{}, I1{}, nb);
        if constexpr ((DIAG & 1) == 0) {
          constexpr int u2 = u + 6;
          issue_w(u2 % PL_RING, wto[u2 % 9], k2 + u2 / 9);
        }
        mfma4(I2{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        read_b(un{}, I2{}, nb); read_b(un{}, I3{}, nb);
        if constexpr ((DIAG & 1) == 0) {
          if constexpr (U == 0) issue_plane2(1, k2 + 1, 0);     // buffer 1 was last read in pair-step 7 of the previous iteration
          if constexpr (U == 1) issue_plane2(1, k2 + 1, 2);
          if constexpr (GST != 0 && U >= 2 && U <= 4) issue_g(wave + 8 * (gj + U - 2));
          if constexpr (GST == 0) {
            if constexpr (U == 5) issue_plane2(0, k2 + 2, 0);   // buffer 0 was last read in pair-step 3
            if constexpr (U == 6) issue_plane2(0, k2 + 2, 2);
          } else {
            if constexpr (U == 5) { issue_p0(0); issue_p0(1); }
            if constexpr (U == 6) { issue_p0(2); issue_p0(3); }
          }
        }
        mfma4(I3{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        read_b(un{}, I4{}, nb); read_b(un{}, I5{}, nb);
        mfma4(I4{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        read_b(un{}, I6{}, nb);
        mfma4(I5{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
        mfma4(I6{}, fa, fb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = na[i];
#pragma unroll
        for (int j = 0; j < PB; ++j) fb[j] = nb[j];
      });
    }
  };
  if (set == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
  S2P_WAIT_VMCNT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr ((DIAG & 128) != 0) ph[2] = __builtin_amdgcn_s_memrealtime();
  // GST: what the tail still takes through registers -- the seventh row group's gamma | beta chunks and, in the backward form, the norm
  // input xn -- is requested HERE: the loop's fragment registers are dead, and the loads' (cold) latency runs under the accumulator
  // exchange and the staging pass instead of in front of the first plane sum
  constexpr int T_MAXR = BPIX / 64, T_KG0 = GST ? 6 : 0;
  Chunk<T> pre_g[T_MAXR - T_KG0], pre_b[T_MAXR - T_KG0], pre_x[MAT == 2 ? T_MAXR : 1], pre_a[MAT == 1 ? T_MAXR : 1];
  if constexpr (GST != 0) {
    const int ch = tid & 7, r0 = tid >> 3, lc = co_base + ch * 8;
    const T* gbb = (const T*)a.gb + (size_t)img * HW * a.gb_pitch + lc;
#pragma unroll
    for (int k = T_KG0; k < T_MAXR; ++k) {
      const int row = r0 + 64 * k;
      pre_g[k - T_KG0].raw = (u32x4){0u, 0u, 0u, 0u}; pre_b[k - T_KG0].raw = pre_g[k - T_KG0].raw;
      if (row < HW) {
        pre_g[k - T_KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch);
        pre_b[k - T_KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch + a.Cout);
      }
    }
    if constexpr (MAT == 1) {                                  // ... and the residual / producer tensor of the conv's own epilogue (conv_1 + skip)
      if (a.epi != S2P_EPI_STORE) {
        const T* ab = (const T*)a.aux + (size_t)g * a.y_gstride + (size_t)img * HW * a.y_pitch + lc;
#pragma unroll
        for (int k = 0; k < T_MAXR; ++k) {
          const int row = r0 + 64 * k;
          pre_a[k].raw = (u32x4){0u, 0u, 0u, 0u};
          if (row < HW) pre_a[k].raw = *(const u32x4*)(ab + (size_t)row * a.y_pitch);
        }
      }
    }
    if constexpr (MAT == 2) {
      const T* xb = (const T*)a.xn + (size_t)img * HW * a.xn_pitch + lc;
#pragma unroll
      for (int k = 0; k < T_MAXR; ++k) {
        const int row = r0 + 64 * k;
        pre_x[k].raw = (u32x4){0u, 0u, 0u, 0u};
        if (row < HW) pre_x[k].raw = *(const u32x4*)(xb + (size_t)row * a.xn_pitch);
      }
    }
  }
  if constexpr ((DIAG & 16) != 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && g == 0) { unsigned long long* o = (unsigned long long*)a.y + (size_t)blockIdx.x * 2; o[0] = c1 - st_c0; o[1] = r1 - st_r0; }
    float chk = 0.f;                                            // keeps every accumulator (hence every MFMA) alive
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < PB; ++j) chk += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (chk == 12345.678f) ((float*)a.y)[7] = chk;
    return;
  }

  // ---- add the two partial accumulators of a wave pair through LDS: the pair exchanges halves (set 0 ends up with the sums
  //      of co blocks 0-1, set 1 with co blocks 2-3), so all eight waves share the epilogue -----------------------------------
  // staging row of pixel px, 16-byte chunk c (8 channels): 144-byte rows, or (GST) 128-byte rows with the chunk index XOR px & 7
  auto srow = [&](int px, int c) -> char* {
    if constexpr (GST != 0) return smem + XOFF + px * 128 + ((c ^ (px & 7)) << 4);
    else return smem + px * PL_ERS + c * 16;
  };
  auto finish = [&](auto ibc) {
    constexpr int IB = decltype(ibc)::value;                    // first co block this wave keeps; it hands over the other two
    if constexpr (GST == 0) {
      char* mb = smem + (size_t)(wq * NT) * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4v*)(mb + ((2 - IB + i) * PB + j) * 1024) = acc[2 - IB + i][j];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[IB + i][j] += *(const f32x4v*)(mb + ((IB + i) * PB + j) * 1024);
      __syncthreads();                                          // the staging rows below overlap the exchange area
    } else {
      // two rounds of one co block per wave: a wave pair shares 2 x 7 KB; set 0 writes slot 1 and reads slot 0, set 1 the reverse
      char* mb = smem + XOFF + (size_t)(wq * 2 * PB) * 1024 + lane * 16;
      constexpr int WS = IB == 0 ? 1 : 0, RS = 1 - WS;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < PB; ++j) *(f32x4v*)(mb + (WS * PB + j) * 1024) = acc[2 - IB + i][j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[IB + i][j] += *(const f32x4v*)(mb + (RS * PB + j) * 1024);
        __syncthreads();                                        // (the last one: the staging rows below overlap the exchange area)
      }
    }
    if constexpr ((DIAG & 128) != 0) ph[3] = __builtin_amdgcn_s_memrealtime();
    // bias + activation in registers, then [pixel][co] staging rows (transpose through LDS)
    const float* bias = a.bias ? a.bias + (size_t)g * a.Cout + co_base : nullptr;
    float bv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[i][e] = bias ? bias[16 * (IB + i) + 4 * q + e] : 0.f;
    // the activation selector is resolved ONCE (a uniform branch around the whole pass), never per element
    auto stage_out = [&](auto f) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const int px = (wq * PB + j) * 16 + l15;
          const f32x4v v = acc[IB + i][j];
          bf16x4 o = {(__bf16)f(v[0] + bv[i][0]), (__bf16)f(v[1] + bv[i][1]), (__bf16)f(v[2] + bv[i][2]), (__bf16)f(v[3] + bv[i][3])};
          *(bf16x4*)(srow(px, 2 * (IB + i) + (q >> 1)) + (q & 1) * 8) = o;
        }
    };
    if (a.act == S2P_ACT_TANH) stage_out([](float v) { return tanhf(v); });
    else if (a.act == S2P_ACT_SWISH) stage_out([](float v) { return v / (1.f + expf(-v)); });
    else if (a.act == S2P_ACT_NONE) stage_out([](float v) { return v; });        // (dgrads, gamma/beta conv: the pass is VALU-bound)
    else {
      const float ns = a.act == S2P_ACT_RELU ? 0.f : a.slope;                    // relu / lrelu
      stage_out([ns](float v) { return lrelu_ns(v, ns); });
    }
  };
  if (set == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 2>{});
  __syncthreads();
  if constexpr ((DIAG & 128) != 0) ph[4] = __builtin_amdgcn_s_memrealtime();
  T* yg = (T*)a.y + (size_t)g * a.y_gstride;
  const T* auxg = a.aux ? (const T*)a.aux + (size_t)g * a.y_gstride : nullptr;
  const T* aux2g = a.aux2 ? (const T*)a.aux2 + (size_t)g * a.y_gstride : nullptr;
  const bool epi_add = a.epi == S2P_EPI_ADD;
  const bool g_tanh = a.gact == S2P_ACT_TANH;
  const float gneg = a.gact == S2P_ACT_RELU ? 0.f : (a.gact == S2P_ACT_LRELU ? a.gslope : 1.f);
  // one (pixel row, 8-channel chunk) item of the output: staged value (+ residual / producer-activation-gradient epilogue)
  auto out_chunk = [&](int row, int ch, size_t go, const Chunk<T>* pre = nullptr) {      // pre: aux chunk already in registers
    Chunk<T> c;
    c.raw = *(const u32x4*)srow(row, ch);
    if (a.epi != S2P_EPI_STORE) {
      Chunk<T> x, x2;
      if (pre) x = *pre; else x.raw = *(const u32x4*)(auxg + go);
      x2.raw = (u32x4){0u, 0u, 0u, 0u};
      if (aux2g) x2.raw = *(const u32x4*)(aux2g + go);
      float ov[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = c.get(e), xv = x.get(e);
        const float f = g_tanh ? 1.f - xv * xv : (xv > 0.f ? 1.f : gneg);
        ov[e] = epi_add ? v + xv : (v + x2.get(e)) * f;
      }
      c.pack(ov);
    }
    return c;
  };
  if constexpr (MAT == 0) {
    for (int idx = tid; idx < HW * 8; idx += 512) {
      const int row = idx >> 3, ch = idx & 7;
      const size_t go = ((size_t)img * HW + row) * a.y_pitch + co_base + ch * 8;
      *(u32x4*)(yg + go) = out_chunk(row, ch, go).raw;
    }
  } else if constexpr (MAT == 1) {
    // ---- fused InstanceNorm + MAT modulation (norm.hip: in_fused_fwd_kernel) on the plane this workgroup owns ------------
    // Thread (row lane r = tid >> 3, chunk ch = tid & 7) holds the rows r, r + 64, ... of its 8 channels: the conv output
    // (as stored: bf16) stays in registers, the statistics are the exact two-pass ones (mean, then centred second moment),
    // summed in a fixed order (lanes, then waves), and the modulated tensor is written from the same registers.
    constexpr int MAXR = BPIX / 64;
    const int ch = tid & 7, r0 = tid >> 3;
    const T* gbb = a.gb ? (const T*)a.gb + (size_t)img * HW * a.gb_pitch + co_base + ch * 8 : nullptr;
    constexpr int KG0 = GST ? GB_ROWS / 64 : 0;                  // rows r0 + 64 k, k < KG0: gamma | beta are staged in LDS
    static_assert(!GST || KG0 == T_KG0, "row groups staged in LDS");
    Chunk<T> xv[MAXR], gv[MAXR - KG0], bv[MAXR - KG0];
#pragma unroll
    for (int k = KG0; k < MAXR; ++k) {                           // gamma / beta first: their latency runs under the rest
      const int row = r0 + 64 * k;
      if constexpr (GST != 0) { gv[k - KG0] = pre_g[k - KG0]; bv[k - KG0] = pre_b[k - KG0]; }     // (requested behind the loop)
      else {
        gv[k - KG0].raw = (u32x4){0u, 0u, 0u, 0u}; bv[k - KG0].raw = gv[k - KG0].raw;
        if (gbb && row < HW) {
          gv[k - KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch);
          bv[k - KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch + a.Cout);
        }
      }
    }
    // GST: LDS address of this thread's gamma chunk of row r0 (beta: ^ 128); row r0 + 64 k is 64 rows = 16 KB further (r0 + 64 k
    // keeps r0's parity), rows from 288 on live in plane buffer 0
    auto gst_row = [&](int k) -> int {
      const int row = r0 + 64 * k;
      return (row < GB_ROWS_B ? GB_B + row * 256 : (row - GB_ROWS_B) * 256) + ((r0 & 1) << 7) + ch * 16;
    };
    // The plane stays in registers UNPACKED from here on (56 floats: the accumulators are dead): round 4 kept it packed and paid
    // three unpack passes; with the centred values kept from the second pass, the maximum form of the activation and paired
    // bf16 conversions the tail is ~330 VALU instructions per wave shorter (of 1 320)
    float xf[MAXR][8];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      xv[k].raw = (u32x4){0u, 0u, 0u, 0u};
      if (row < HW) {
        const size_t go = ((size_t)img * HW + row) * a.y_pitch + co_base + ch * 8;
        if constexpr (GST != 0) xv[k] = out_chunk(row, ch, go, &pre_a[k]); else xv[k] = out_chunk(row, ch, go);
        if (a.y) *(u32x4*)(yg + go) = xv[k].raw;                  // (y == NULL: the caller keeps only the modulated tensor -- a forward without a backward)
      }
      xv[k].unpack(xf[k]);                                        // rows beyond HW hold zeros
    }
    __syncthreads();                                            // the staging rows are dead: LDS is scratch from here on
    float* red = (float*)(smem + XOFF);                         // [8 waves][64]
    float* cst = red + 8 * 64;                                  // [4][64]: plane sum / M2, then 1 + gamma_st, beta_st
    auto plane_sum = [&](float (&v)[8], int slot) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) v[e] += __shfl_xor(v[e], o, 64);
      }
      __syncthreads();
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * 64 + lane * 8 + e] = v[e];
      }
      __syncthreads();
      if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w * 64 + tid];
        cst[slot * 64 + tid] = t;
      }
      __syncthreads();
    };
    const float inv = 1.f / (float)HW;
    float sacc[8], mean[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sacc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) sacc[e] += xf[k][e];
    }
    plane_sum(sacc, 0);
    // rows r0 + 64 k with k < KFULL exist in every thread (GST: the host requires HW >= 384): no bounds select on their squares
    constexpr int KFULL = GST ? 6 : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = cst[ch * 8 + e] * inv;
      sacc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const float d = xf[k][e] - mean[e];
        xf[k][e] = d;                                             // the centred value is what the last pass needs
        if (k < KFULL) sacc[e] = __builtin_fmaf(d, d, sacc[e]);
        else sacc[e] += (r0 + 64 * k < HW) ? d * d : 0.f;
      }
    }
    plane_sum(sacc, 1);
    if (tid < 64) {
      const int c = co_base + tid;
      float* o = a.stats + 4 + ((size_t)img * a.Cout + c) * 2;     // norm.hip format: 4-word header, then [N][C][1 split]{mean, M2}
      o[0] = cst[tid] * inv; o[1] = cst[64 + tid];
      if (c == 0 && img == 0) *(i32x4*)a.stats = (i32x4){1, HW, 0, 0};
      cst[2 * 64 + tid] = a.gbst ? 1.f + a.gbst[(size_t)img * a.gbst_pitch + c] : 1.f;
      cst[3 * 64 + tid] = a.gbst ? a.gbst[(size_t)img * a.gbst_pitch + a.Cout + c] : 0.f;
    }
    __syncthreads();
    float rstd[8], gs[8], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      rstd[e] = 1.f / sqrtf(cst[64 + ch * 8 + e] * inv + a.eps);
      gs[e] = cst[2 * 64 + ch * 8 + e]; bs[e] = cst[3 * 64 + ch * 8 + e];
    }
    const float nns = a.n_act == S2P_ACT_RELU ? 0.f : (a.n_act == S2P_ACT_LRELU ? a.n_slope : 1.f);   // none / relu / lrelu (host)
    T* y2 = (T*)a.y2 + (size_t)img * HW * a.y2_pitch + co_base + ch * 8;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      if (row >= HW) break;
      Chunk<T> o0, gk, bk;
      if (k < KG0) { const int go = gst_row(k); gk.raw = *(const u32x4*)(smem + go); bk.raw = *(const u32x4*)(smem + (go ^ 128)); }
      else { gk = gv[k < KG0 ? 0 : k - KG0]; bk = bv[k < KG0 ? 0 : k - KG0]; }
      float ov[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = gs[e] + gk.get(e), bb = bs[e] + bk.get(e);
        const float xh = xf[k][e] * rstd[e];                      // (x - mean) * rstd
        const float yv = __builtin_fmaf(xh, gg, bb);              // (norm.hip: mat_value)
        ov[e] = lrelu_ns(yv, nns);
      }
      o0.pack(ov);
      *(u32x4*)(y2 + (size_t)row * a.y2_pitch) = o0.raw;
    }
  }
  if constexpr (MAT == 2) {
    // ---- fused backward of InstanceNorm + MAT modulation + activation (norm.hip: in_fused_bwd_kernel) ---------------------
    // This launch is the dgrad of the conv that CONSUMED the norm's output, so the staged plane is dL/d(norm output) for the
    // (image, 64-channel slab) this workgroup owns; it never goes to HBM.  Thread (row lane r0 = tid >> 3, chunk ch = tid & 7)
    // loads the norm INPUT xn, gamma and beta of its rows, forms the four plane sums (lanes, then waves: fixed order) and
    // writes dL/d(xn) (+ the skip gradient `res`), d(gamma_img | beta_img) and the state-affine gradient.
    constexpr int MAXR = BPIX / 64;
    const int ch = tid & 7, r0 = tid >> 3, lc = co_base + ch * 8;
    const T* xb = (const T*)a.xn + (size_t)img * HW * a.xn_pitch + lc;
    const T* gbb = a.gb ? (const T*)a.gb + (size_t)img * HW * a.gb_pitch + lc : nullptr;
    constexpr int KG0 = GST ? GB_ROWS / 64 : 0;                  // rows r0 + 64 k, k < KG0: gamma | beta are staged in LDS
    Chunk<T> xv[MAXR], gv[MAXR - KG0], bv[MAXR - KG0], dv[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      xv[k].raw = (u32x4){0u, 0u, 0u, 0u}; dv[k].raw = xv[k].raw;
      if (k >= KG0) { gv[k < KG0 ? 0 : k - KG0].raw = xv[k].raw; bv[k < KG0 ? 0 : k - KG0].raw = xv[k].raw; }
      if constexpr (GST != 0) {                                  // xn and the seventh row group's gamma | beta: requested behind the loop
        xv[k] = pre_x[k];
        if (k >= KG0) { gv[k < KG0 ? 0 : k - KG0] = pre_g[k < KG0 ? 0 : k - KG0]; bv[k < KG0 ? 0 : k - KG0] = pre_b[k < KG0 ? 0 : k - KG0]; }
        if (row < HW) dv[k].raw = *(const u32x4*)srow(row, ch);
      } else if (row < HW) {
        xv[k].raw = *(const u32x4*)(xb + (size_t)row * a.xn_pitch);
        if (k >= KG0 && gbb) { gv[k < KG0 ? 0 : k - KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch); bv[k < KG0 ? 0 : k - KG0].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch + a.Cout); }
        dv[k].raw = *(const u32x4*)srow(row, ch);                // rows beyond HW stay zero: they add nothing to the sums
      }
    }
    auto gst_row = [&](int k) -> int {                          // (see the forward tail)
      const int row = r0 + 64 * k;
      return (row < GB_ROWS_B ? GB_B + row * 256 : (row - GB_ROWS_B) * 256) + ((r0 & 1) << 7) + ch * 16;
    };
    // gamma | beta chunk of this thread's row r0 + 64 k: from the staged rows (a masked row reads some other row: its dv is zero)
    auto load_gb = [&](int k, Chunk<T>& gk, Chunk<T>& bk) {
      if (k < KG0) { const int go = gst_row(k); gk.raw = *(const u32x4*)(smem + go); bk.raw = *(const u32x4*)(smem + (go ^ 128)); }
      else { gk = gv[k < KG0 ? 0 : k - KG0]; bk = bv[k < KG0 ? 0 : k - KG0]; }
    };
    __syncthreads();                                            // the staging rows are dead: LDS is scratch from here on
    float* red = (float*)(smem + XOFF);                         // [4 sums][8 waves][64]
    float* cst = red + 4 * 8 * 64;                              // [6][64]: mean, rstd, 1 + gamma_st, beta_st, s1 / HW, s2 / HW
    if (tid < 64) {
      const int c = co_base + tid;
      // merge the per-split partial moments (norm.hip: mean_rstd; S = 1 when a fused forward kernel wrote them)
      const int S = ((const int*)a.stats)[0], rows = ((const int*)a.stats)[1];
      const float* pm = a.stats + 4 + ((size_t)img * a.Cout + c) * S * 2;
      const float inv = 1.f / (float)HW;
      const float m0 = pm[0];
      float m = 0.f;
      for (int b = 1; b < S; ++b) { int nb = HW - b * rows; if (nb > rows) nb = rows; m += (float)nb * (pm[2 * b] - m0); }
      m = m0 + m * inv;
      float M2 = 0.f;
      for (int b = 0; b < S; ++b) { int nb = HW - b * rows; if (nb > rows) nb = rows; const float dd = pm[2 * b] - m; M2 += pm[2 * b + 1] + (float)nb * dd * dd; }
      cst[tid] = m; cst[64 + tid] = 1.f / sqrtf(M2 * inv + a.eps);
      cst[2 * 64 + tid] = a.gbst ? 1.f + a.gbst[(size_t)img * a.gbst_pitch + c] : 1.f;
      cst[3 * 64 + tid] = a.gbst ? a.gbst[(size_t)img * a.gbst_pitch + a.Cout + c] : 0.f;
    }
    __syncthreads();
    const float gneg = a.n_act == S2P_ACT_RELU ? 0.f : (a.n_act == S2P_ACT_LRELU ? a.n_slope : 1.f);
    // ---- pass 1: the four plane sums.  Rows outermost (one read of the gamma | beta chunk per row); every sum adds its rows in
    // ascending order.  The normalised input xh and the gradient dy behind the activation stay in registers for pass 2 (112 floats:
    // the accumulators are dead) -- round 4 recomputed both there from the packed chunks and a 64-bit mask of the activation
    // branches: 40 VALU instructions per element over the two passes, now ~23
    float xhf[MAXR][8], dyf[MAXR][8];
    {
      float mm[8], rr[8], g1[8], b1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int cl = ch * 8 + e; mm[e] = cst[cl]; rr[e] = cst[64 + cl]; g1[e] = cst[128 + cl]; b1[e] = cst[192 + cl]; }
      float q0[8], q1[8], q2[8], q3[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { q0[e] = 0.f; q1[e] = 0.f; q2[e] = 0.f; q3[e] = 0.f; }
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        Chunk<T> gk, bk;
        load_gb(k, gk, bk);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gg = g1[e] + gk.get(e), bb = b1[e] + bk.get(e);
          const float xh = (xv[k].get(e) - mm[e]) * rr[e];
          const float yv = __builtin_fmaf(xh, gg, bb);          // (norm.hip: mat_value -- the forward's rounding)
          const float dvv = dv[k].get(e);
          const float dy = yv > 0.f ? dvv : dvv * gneg;
          const float dxh = dy * gg;
          xhf[k][e] = xh; dyf[k][e] = dy;
          q0[e] += dxh; q1[e] += dxh * xh; q2[e] += dy * xh; q3[e] += dy;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cl = ch * 8 + e;
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          q0[e] += __shfl_xor(q0[e], o, 64); q1[e] += __shfl_xor(q1[e], o, 64); q2[e] += __shfl_xor(q2[e], o, 64); q3[e] += __shfl_xor(q3[e], o, 64);
        }
        if (lane < 8) { red[(0 * 8 + wave) * 64 + cl] = q0[e]; red[(1 * 8 + wave) * 64 + cl] = q1[e]; red[(2 * 8 + wave) * 64 + cl] = q2[e]; red[(3 * 8 + wave) * 64 + cl] = q3[e]; }
      }
    }
    __syncthreads();
    if (tid < 64) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { t0 += red[(0 * 8 + w) * 64 + tid]; t1 += red[(1 * 8 + w) * 64 + tid]; t2 += red[(2 * 8 + w) * 64 + tid]; t3 += red[(3 * 8 + w) * 64 + tid]; }
      const float inv = 1.f / (float)HW;
      cst[4 * 64 + tid] = t0 * inv; cst[5 * 64 + tid] = t1 * inv;
      const int c = co_base + tid;
      if (a.dgbst) {
        a.dgbst[(size_t)img * a.dgbst_pitch + c] = t2;
        a.dgbst[(size_t)img * a.dgbst_pitch + a.Cout + c] = t3;
      }
    }
    __syncthreads();
    // ---- pass 2: outputs
    T* dxo = (T*)a.y2 + (size_t)img * HW * a.y2_pitch + lc;
    T* dgo = a.dgb ? (T*)a.dgb + (size_t)img * HW * a.dgb_pitch + lc : nullptr;
    const T* rsb = a.res ? (const T*)a.res + (size_t)img * HW * a.res_pitch + lc : nullptr;
    float rr2[8], g12[8], s1v[8], s2v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int cl = ch * 8 + e; rr2[e] = cst[64 + cl]; g12[e] = cst[128 + cl]; s1v[e] = cst[256 + cl]; s2v[e] = cst[320 + cl]; }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      if (row >= HW) break;
      Chunk<T> o0, o1, o2, gk, bk;
      load_gb(k, gk, bk);
      float v0[8], v1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = g12[e] + gk.get(e);
        const float xh = xhf[k][e], dy = dyf[k][e];
        const float dxh = dy * gg;
        v0[e] = rr2[e] * (dxh - s1v[e] - xh * s2v[e]);
        v1[e] = dy * xh;
      }
      if (rsb) {                                                // skip-connection gradient folded into the store (fp32 add, one rounding)
        Chunk<T> rv; rv.raw = *(const u32x4*)(rsb + (size_t)row * a.res_pitch);
        // (round 4 rounded dx to bf16 before the add: o0.get(e) + rv.get(e); kept, bit for bit)
        o0.pack(v0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v0[e] = o0.get(e) + rv.get(e);
      }
      o0.pack(v0); o1.pack(v1); o2.pack(dyf[k]);
      *(u32x4*)(dxo + (size_t)row * a.y2_pitch) = o0.raw;
      if (dgo) {
        *(u32x4*)(dgo + (size_t)row * a.dgb_pitch) = o1.raw;
        *(u32x4*)(dgo + (size_t)row * a.dgb_pitch + a.Cout) = o2.raw;
      }
    }
  }
  if constexpr ((DIAG & 128) != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ph[5] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && g == 0 && a.aux && a.epi == S2P_EPI_STORE) {
      unsigned long long* o = (unsigned long long*)a.aux + (size_t)blockIdx.x * 8;
#pragma unroll
      for (int k = 0; k < 6; ++k) o[k] = ph[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// PAIR form for launches with several tiles per CU (the grouped gamma/beta conv: 24 tiles per CU; VGG conv3 at N 128: 2): one
// workgroup computes TWO neighbouring 64-channel slabs of the same image from ONE resident plane.  Wave set s (waves 4s..4s+3)
// owns slab 2*pair + s and runs EVERY K step on it with its own 6-stage weight ring; both sets read the same plane buffers.
// Against two single-slab workgroups this halves the plane traffic and the prologues, and there is no K split, hence no
// accumulator exchange; the LDS fragment traffic per MFMA is unchanged (11 reads per 28 MFMAs per wave and step).
template <int PB, int WP, int DIAG>
__global__ __launch_bounds__(512) void conv_plane_pair_kernel(const PlaneArgs a) {
  typedef __bf16 T;
  constexpr int BPIX = 4 * PB * 16;
  constexpr int RINGB = PL_RING * PL_WST;                      // one set's weight ring
  constexpr int MAIN = 2 * PL_PBUF + 2 * RINGB;
  constexpr int EPI = BPIX * PL_ERS;                           // one set's staging rows
  constexpr int SMEM = 2 * EPI > MAIN ? 2 * EPI : MAIN;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* const pbase = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = wave >> 2, wq = wave & 3;
  const int q = lane >> 4, l15 = lane & 15;
  int g = blockIdx.y;
  int img, cpair;
  {
    const int bid = blockIdx.x, np = a.nco >> 1;
    if (a.gxcd) {
      // grouped launch, GROUPS on XCDs (round 5).  With the group in blockIdx.y every XCD works on the same group at the same time and
      // fetches that group's weights into its own L2: 8 copies of all the weights per launch (113 MB of the gamma/beta conv's 200 MB of
      // reads).  Here the launch is cut into (group, half of the images) units, unit u runs on XCD u % 8 (blocks b, b + 8, ... share an
      // XCD under the observed round-robin placement: speed only), an XCD's units one after the other: 3 groups' weights per XCD
      // instead of 12, every activation slice still read by exactly one XCD.
      const int L = blockIdx.y * gridDim.x + bid, xcd = L & 7, j = L >> 3;
      const int per_unit = (a.N >> 1) * np;
      const int u = xcd + 8 * (j / per_unit), r = j % per_unit;
      g = u >> 1; cpair = r % np; img = (u & 1) * (a.N >> 1) + r / np;
    } else if ((a.N & 7) == 0) { const int xcd = bid & 7, k = bid >> 3; cpair = k % np; img = (k / np) * 8 + xcd; }
    else { cpair = bid % np; img = bid / np; }
  }
  const int co_base = (2 * cpair + set) * 64;                   // this wave set's slab
  const int HW = a.H * a.W;
  char* const wbase = smem + 2 * PL_PBUF + set * RINGB;         // this wave set's ring
  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  const i32x4 xrs = s2p_make_rsrc(xg, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u);
  const i32x4 wrs = s2p_make_rsrc(wg, a.w_bytes);
  const unsigned p_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(pbase));
  const unsigned w_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(wbase));
  const unsigned OOB = 0x80000000u;
  // plane: 32 pieces per half-slab, wave w issues pieces w, w + 8, w + 16, w + 24 (one per K step 0..3 / 9..12)
  int hv[4]; unsigned hdst[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ii = wave + 8 * k;
    const int cp = ii & 1, pg = ii >> 1;
    const int pos = 32 * pg + (lane >> 1);
    const int pm = pos - 1;
    const int yy = pm / WP - 1, xx = pm % WP;
    const bool ok = pos >= 1 && yy >= 0 && yy < a.H && xx < a.W;
    hv[k] = ok ? (int)((((unsigned)(img * a.H + yy) * a.W + xx) * a.x_pitch) * 2u + (2 * cp + (lane & 1)) * 16) : (int)OOB;
    hdst[k] = (unsigned)(cp * PL_CPS + pg * 1024);
  }
  // weights: 4 pieces per stage and set; wave w issues piece (w & 3) of ITS set's stage
  int wv; unsigned wdst;
  {
    const int cp = wave & 1, cohalf = (wave >> 1) & 1;
    const int co = co_base + 32 * cohalf + (lane >> 1);
    wv = co < a.Cout ? (int)((unsigned)co * a.w_row * 2u + (2 * cp + (lane & 1)) * 16) : (int)OOB;
    wdst = (unsigned)(cp * 2048 + cohalf * 1024);
  }
  int wto[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wto[t] = a.wt[t] * a.Cin * 2;
  const int nhs = a.Cin / 32;
  auto issue_plane1 = [&](int buf, int hs, int k) {
    pl_dma16(xrs, p_lds + (unsigned)(buf * PL_PBUF) + hdst[k], hs < nhs ? hv[k] : (int)OOB, hs * 64);
  };
  auto issue_w = [&](int stage, int tap_off, int hs) {
    pl_dma16(wrs, w_lds + (unsigned)(stage * PL_WST) + wdst, hs < nhs ? wv : (int)OOB, tap_off + hs * 64);
  };
  // prologue: the plane of half-slab 0, the weight stages of K steps 0, 1, 2
#pragma unroll
  for (int k = 0; k < 4; ++k) issue_plane1(0, 0, k);
  issue_w(0, wto[0], 0);
  issue_w(1, wto[1], 0);
  issue_w(2, wto[2], 0);
  int bB[PB];
  {
    const float rw = 1.0f / (float)a.W;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      int m = (wq * PB + j) * 16 + l15;
      if (m >= HW) m = HW - 1;
      const int y = (int)(((float)m + 0.5f) * rw), x = m - y * a.W;
      bB[j] = (q >> 1) * PL_CPS + (q & 1) * 16 + (y * WP + x) * 32;
    }
  }
  const int bA = (q >> 1) * 2048 + l15 * 32 + (q & 1) * 16;
  f32x4v acc[4][PB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  auto read_a = [&](auto uc, auto ic, bf16x8 (&fa)[4]) {
    constexpr int u = decltype(uc)::value % 18, i = decltype(ic)::value;
    fa[i] = *(const bf16x8*)(wbase + (u % PL_RING) * PL_WST + i * 512 + bA);
  };
  auto read_b = [&](auto uc, auto jc, bf16x8 (&fb)[PB]) {
    constexpr int u = decltype(uc)::value % 18, j = decltype(jc)::value;
    constexpr int t = u % 9, hsl = u / 9;
    fb[j] = *(const bf16x8*)(pbase + hsl * PL_PBUF + ((t / 3) * WP + (t % 3)) * 32 + bB[j]);
  };
  auto mfma4 = [&](auto jc, bf16x8 (&fa)[4], bf16x8 (&fb)[PB]) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
  typedef std::integral_constant<int, 4> I4; typedef std::integral_constant<int, 5> I5;
  typedef std::integral_constant<int, 6> I6;
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");              // plane 0 and the stage of step 0 have landed ...
  __builtin_amdgcn_s_barrier();                                 // ... for every wave
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 fa[4], fb[PB];
  pl_static_for<0, 4>([&](auto ic) { read_a(I0{}, ic, fa); });
  pl_static_for<0, PB>([&](auto jc) { read_b(I0{}, jc, fb); });
  // K step u of an 18-step iteration (two half-slabs): MFMAs of step u on the fragments in registers; between them the
  // fragment reads of step u + 1, the weight piece of step u + 3 (ring stage (u + 3) % 6, last read in step u - 4) and, in
  // steps 0..3 / 9..12, one piece of the plane of the next-but-one / next half-slab (buffer 1 is free from step 17 of the
  // previous iteration and needed by the reads of step 8; buffer 0 is free from step 8 and needed in step 17).
  for (int k2 = 0; k2 < nhs; k2 += 2) {
    pl_static_for<0, 18>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      typedef std::integral_constant<int, u + 1> un;
      constexpr int up = (u + 17) % 18;                         // previous step: what it issued may still be in flight
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (up <= 3 || (up >= 9 && up <= 12)) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 na[4], nb[PB];
      read_a(un{}, I0{}, na); read_a(un{}, I1{}, na);
      mfma4(I0{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_a(un{}, I2{}, na); read_a(un{}, I3{}, na);
      mfma4(I1{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_b(un{}, I0{}, nb); read_b(un{}, I1{}, nb);
      { constexpr int u3 = u + 3; issue_w(u3 % PL_RING, wto[u3 % 9], k2 + u3 / 9); }
      mfma4(I2{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_b(un{}, I2{}, nb); read_b(un{}, I3{}, nb);
      if constexpr (u <= 3) issue_plane1(1, k2 + 1, u);
      if constexpr (u >= 9 && u <= 12) issue_plane1(0, k2 + 2, u - 9);
      mfma4(I3{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_b(un{}, I4{}, nb); read_b(un{}, I5{}, nb);
      mfma4(I4{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_b(un{}, I6{}, nb);
      mfma4(I5{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(I6{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = na[i];
#pragma unroll
      for (int j = 0; j < PB; ++j) fb[j] = nb[j];
    });
  }
  S2P_WAIT_VMCNT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- epilogue: every wave stages its 28 tiles into its set's [pixel][co] rows; then 16-B stores of both slabs --------
  {
    char* stg = smem + set * EPI;
    const float* bias = a.bias ? a.bias + (size_t)g * a.Cout + co_base : nullptr;
    float bv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[i][e] = bias ? bias[16 * i + 4 * q + e] : 0.f;
    auto stage_out = [&](auto f) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const int px = (wq * PB + j) * 16 + l15;
          const f32x4v v = acc[i][j];
          bf16x4 o = {(__bf16)f(v[0] + bv[i][0]), (__bf16)f(v[1] + bv[i][1]), (__bf16)f(v[2] + bv[i][2]), (__bf16)f(v[3] + bv[i][3])};
          *(bf16x4*)(stg + px * PL_ERS + (16 * i + 4 * q) * 2) = o;
        }
    };
    if (a.act == S2P_ACT_TANH) stage_out([](float v) { return tanhf(v); });
    else if (a.act == S2P_ACT_SWISH) stage_out([](float v) { return v / (1.f + expf(-v)); });
    else if (a.act == S2P_ACT_NONE) stage_out([](float v) { return v; });        // (dgrads, gamma/beta conv: the pass is VALU-bound)
    else {
      const float ns = a.act == S2P_ACT_RELU ? 0.f : a.slope;
      stage_out([ns](float v) { return lrelu_ns(v, ns); });
    }
  }
  __syncthreads();
  {
    T* yg = (T*)a.y + (size_t)g * a.y_gstride;
    const T* auxg = a.aux ? (const T*)a.aux + (size_t)g * a.y_gstride : nullptr;
    const T* aux2g = a.aux2 ? (const T*)a.aux2 + (size_t)g * a.y_gstride : nullptr;
    const bool epi_add = a.epi == S2P_EPI_ADD;
    const bool g_tanh = a.gact == S2P_ACT_TANH;
    const float gneg = a.gact == S2P_ACT_RELU ? 0.f : (a.gact == S2P_ACT_LRELU ? a.gslope : 1.f);
    for (int idx = tid; idx < HW * 16; idx += 512) {            // (row, 16 chunks of 8 channels: the two slabs side by side)
      const int row = idx >> 4, ch16 = idx & 15, sl = ch16 >> 3, ch = ch16 & 7;
      Chunk<T> c;
      c.raw = *(const u32x4*)(smem + sl * EPI + row * PL_ERS + ch * 16);
      const size_t go = ((size_t)img * HW + row) * a.y_pitch + (2 * cpair + sl) * 64 + ch * 8;
      if (a.epi != S2P_EPI_STORE) {
        Chunk<T> x, x2;
        x.raw = *(const u32x4*)(auxg + go);
        x2.raw = (u32x4){0u, 0u, 0u, 0u};
        if (aux2g) x2.raw = *(const u32x4*)(aux2g + go);
        float ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = c.get(e), xv = x.get(e);
          const float f = g_tanh ? 1.f - xv * xv : (xv > 0.f ? 1.f : gneg);
          ov[e] = epi_add ? v + xv : (v + x2.get(e)) * f;
        }
        c.pack(ov);
      }
      *(u32x4*)(yg + go) = c.raw;
    }
  }
}

bool s2p_conv_plane_applicable(const PlaneArgs& a) {
  if (a.H > 21 || a.W > 21 || a.H < 3 || a.W < 3) return false;      // (H+2)*22 + 2 <= 512 positions
  const int HW = a.H * a.W;
  if (HW > 448 || HW <= 320) return false;                              // four waves x 7 blocks of 16 px; smaller planes waste the tile
  if (a.Cin % 64 || a.Cout % 64 || a.Cst != a.Cout) return false;
  if (a.y_pitch % 8 || a.x_pitch % 8) return false;
  return true;
}

int s2p_conv_plane_launch(PlaneArgs& a, int groups, hipStream_t st) {
  a.nco = a.Cout / 64;
  dim3 grid(a.N * a.nco, groups);
#ifdef S2P_DIAG_BUILD
  static const int diag = s2p_env_int("S2P_DIAG", 0);
#define PL_DIAG_CASE(D) if (diag == D) { hipLaunchKernelGGL((conv_plane_kernel<7, 22, D>), grid, dim3(512), 0, st, a); S2P_CHECK_LAUNCH("conv_plane_kernel(diag)"); return 0; }
  PL_DIAG_CASE(1) PL_DIAG_CASE(2) PL_DIAG_CASE(4) PL_DIAG_CASE(5) PL_DIAG_CASE(8) PL_DIAG_CASE(16) PL_DIAG_CASE(17) PL_DIAG_CASE(20) PL_DIAG_CASE(21) PL_DIAG_CASE(64) PL_DIAG_CASE(128)
#undef PL_DIAG_CASE
#endif
  // several tiles per CU and an even slab count: two slabs per workgroup from one resident plane (no accumulator exchange)
  if (!a.y2 && (a.nco & 1) == 0 && (long long)a.N * a.nco * groups > 256 && !s2p_env_set("S2P_NO_PLANE_PAIR")) {
    dim3 gp(a.N * (a.nco / 2), groups);
    // (group, image half) units on XCDs: 2 * groups units over 8 XCDs, whole units per XCD, the grid's x extent a multiple of 8
    a.gxcd = (groups > 1 && (2 * groups) % 8 == 0 && a.N % 2 == 0 && gp.x % 8 == 0 && !S2P_DIAG_SWITCH(17)) ? 1 : 0;
    hipLaunchKernelGGL((conv_plane_pair_kernel<7, 22, 0>), gp, dim3(512), 0, st, a);
    S2P_CHECK_LAUNCH("conv_plane_pair_kernel");
    return 0;
  }
  // gamma|beta maps present and at least four loop iterations (Cin >= 256): the maps' rows are staged by LDS-DMA under the K loop
  const bool gst = a.y2 && a.gb && a.Cin >= 256 && a.H * a.W >= 384 && (long long)a.N * a.H * a.W * a.gb_pitch * 2 < (1ll << 31) && !S2P_DIAG_SWITCH(7);
  if (a.y2 && a.xn) {
    if (gst) hipLaunchKernelGGL((conv_plane_kernel<7, 22, 0, 2, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_plane_kernel<7, 22, 0, 2>), grid, dim3(512), 0, st, a);
    S2P_CHECK_LAUNCH("conv_plane_kernel(mat bwd)"); return 0;
  }
  if (a.y2) {
    if (gst) hipLaunchKernelGGL((conv_plane_kernel<7, 22, 0, 1, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_plane_kernel<7, 22, 0, 1>), grid, dim3(512), 0, st, a);
    S2P_CHECK_LAUNCH("conv_plane_kernel(mat)"); return 0;
  }
  hipLaunchKernelGGL((conv_plane_kernel<7, 22, 0>), grid, dim3(512), 0, st, a);
  S2P_CHECK_LAUNCH("conv_plane_kernel");
  return 0;
}
