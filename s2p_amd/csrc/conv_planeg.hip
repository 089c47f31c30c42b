// Generalised plane-resident convolution for small feature maps, bf16, NHWC, gfx950 (round 4): the PatchGAN discriminator's 4x4
// layers -- stride 1 (256 -> 512 on 12x12 / 7x7 maps, and their dgrads) and stride 2 (64 -> 128, 128 -> 256) -- with the
// InstanceNorm + LeakyReLU that follows them (forward) or precedes them (backward) in the epilogue.
//
// It is conv_plane.hip's design (read that file's header first) with three things made general:
//   * the tap rectangle TY x TX and the padded row width WP (a run-time value: a tap is  base_row_register[ty] + immediate(tx));
//   * the K-step schedule: 2 * TY*TX steps per iteration (two half-slabs), a weight ring of RING stages (RING divides the steps of
//     an iteration), the plane prefetch windows derived from TY*TX at compile time;
//   * stride 2 as a 2x2 stride-1 convolution over the four PARITY sub-planes of the input: in(2y + dy, 2x + dx), dy = 2(a-1) + py.
//     The LDS-DMA's per-lane source address does the space-to-depth; a "half-slab" is one parity x 32 channels.
#include "s2p_common.h"
#include "conv_planeg.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4v;

template <int B, int E, typename F>
__device__ __forceinline__ void pg_static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); pg_static_for<B + 1, E>(f); }
}
__device__ __forceinline__ void pg_dma16(i32x4 rsrc, unsigned lds_dst, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :: "v"(voffset), "s"(rsrc), "s"(lds_dst), "s"(soffset) : "memory", "m0");
}

#ifndef PG_OCC4_ALL
#define PG_OCC4_ALL 0      // 1 (timing builds): also the 4x4 / parity-form 192-pixel shapes are limited to 128 VGPRs (two workgroups per CU; spills)
#endif
namespace {
constexpr int PG_WST = 4096;                 // one weight stage: [2 pairs][64 co][32 B]
constexpr int PG_ERS = 144;                  // epilogue staging row: 64 co x 2 B + 16

// plane prefetch schedule of an iteration of T pair-steps (compile time).  Buffer 0 holds the even half-slabs (steps [0, T) of an
// iteration), buffer 1 the odd ones (steps [T, 2T)); the fragments of step v are read in pair-step (v - 2) / 2.
//   T >= 6: buffer 1 <- half-slab k2 + 1 in pair-steps [0, (T-2)/2 - 2], buffer 0 <- half-slab k2 + 2 in [(T-3)/2 + 1, T - 3];
//   T == 4: the windows are single pair-steps: buffer 0 <- k2 + 2 in pair-step 1, buffer 1 <- k2 + 3 in pair-step 3 (the prologue
//           loads half-slabs 0 AND 1).
// PPW pieces per wave and plane go out two per pair-step (all of them in the one slot when T == 4).
template <int T, int PPW> struct PgSched {
  static constexpr bool EARLY1 = T >= 6;
  static constexpr int W1_END = EARLY1 ? (T - 2) / 2 - 2 : -1;
  static constexpr int W0_BEG = (T - 3) / 2 + 1, W0_END = T - 3;
  static constexpr int PER1 = EARLY1 ? (PPW + W1_END) / (W1_END + 1) : PPW;              // pieces per pair-step (2 at least: the 3x3 kernel's pace)
  static constexpr int PER0 = EARLY1 ? (PPW + (W0_END - W0_BEG)) / (W0_END - W0_BEG + 1) : PPW;
  static constexpr int Q1 = PER1 < 2 ? 2 : PER1, Q0 = PER0 < 2 ? 2 : PER0;
  static constexpr int NS1 = (PPW + Q1 - 1) / Q1, NS0 = (PPW + Q0 - 1) / Q0;             // pair-steps used per plane
  static constexpr int P0_U0 = EARLY1 ? W0_END + 1 - NS0 : 1;
  static_assert(!EARLY1 || (NS1 <= W1_END + 1 && P0_U0 >= W0_BEG), "plane prefetch does not fit its window");
  // buffer (-1: none), first piece, piece count issued in pair-step U
  static constexpr int buf(int U) {
    if (!EARLY1) return U == 1 ? 0 : (U == 3 ? 1 : -1);
    if (U < NS1) return 1;
    if (U >= P0_U0 && U < P0_U0 + NS0) return 0;
    return -1;
  }
  static constexpr int k0(int U) { return !EARLY1 ? 0 : (U < NS1 ? Q1 * U : Q0 * (U - P0_U0)); }
  static constexpr int cnt(int U) {
    if (buf(U) < 0) return 0;
    if (!EARLY1) return PPW;
    const int q = buf(U) == 1 ? Q1 : Q0;
    return PPW - k0(U) < q ? PPW - k0(U) : q;
  }
  static constexpr int hs_ahead(int U) { return !EARLY1 ? (U == 1 ? 2 : 3) : (buf(U) == 1 ? 1 : 2); }
};
}  // namespace

// MAT: 0 plain conv, 1 + InstanceNorm (+ MAT modulation) + activation of the output plane, 2 + backward of the norm that FED this
// dgrad's forward conv.  GB: gamma / beta maps may be present (false: plain InstanceNorm, no registers spent on them).
template <int TY, int TX, int PB, int NPB, int RING, int MAT, bool S2D, bool GB>
__global__ __launch_bounds__(512, ((PB <= 3 && (TY == 3 || PG_OCC4_ALL)) ? 4 : 2)) void conv_planeg_kernel(const PlaneGArgs a) {
  typedef __bf16 T;
  constexpr int NTAP = TY * TX, NSTEP = 2 * NTAP;
  static_assert(NSTEP % RING == 0 && RING % 2 == 0 && RING >= 6, "ring must divide the steps of an iteration");
  static_assert(NPB % 128 == 0, "a plane buffer is a whole number of pieces per wave");
  constexpr int CPS = NPB * 32;                                // bytes between the two 16-channel pairs of a half-slab
  constexpr int PBUF = 2 * CPS;
  constexpr int PPW = NPB / 128;                               // plane pieces per wave and half-slab
  typedef PgSched<NTAP, PPW> Sched;
  constexpr int BPIX = 4 * PB * 16;
  constexpr int NT = 4 * PB;
  constexpr int MERGE = 4 * NT * 1024;
  constexpr int MAIN = 2 * PBUF + RING * PG_WST;
  constexpr int EPI = BPIX * PG_ERS;
  constexpr int SMEM = MERGE > MAIN ? (MERGE > EPI ? MERGE : EPI) : (MAIN > EPI ? MAIN : EPI);
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char* const pbase = smem;
  char* const wbase = smem + 2 * PBUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = wave >> 2, wq = wave & 3;
  const int q = lane >> 4, l15 = lane & 15;
  const int g = blockIdx.y;
  // workgroup -> (plane = image x row band, co slab): the nco slabs of one plane run back to back on ONE XCD, and so do ALL the
  // bands of an image (blocks b, b + 8, ... share an XCD under the observed round-robin placement: speed only): neighbouring
  // bands read each other's halo rows, which are then L2 hits instead of a second and third HBM read (round 4 dealt the bands
  // round-robin over the XCDs: VGG conv1_2 as 2-row bands read its input 3.0 times from HBM)
  int img, band, cs;
  {
    const int bid = blockIdx.x, nco = a.nco;
    if (a.img_xcd) {
      const int xcd = bid & 7, k = bid >> 3;
      cs = k % nco; band = (k / nco) % a.nbands; img = (k / (nco * a.nbands)) * 8 + xcd;
    } else if ((((a.N / a.gimg) * a.nbands) & 7) == 0) {          // (planes: N / gimg images-per-plane groups x bands)
      const int xcd = bid & 7, k = bid >> 3; cs = k % nco; const int pl = (k / nco) * 8 + xcd; img = pl / a.nbands; band = pl - img * a.nbands;
    } else { cs = bid % nco; const int pl = bid / nco; img = pl / a.nbands; band = pl - img * a.nbands; }
  }
  const int r0 = band * a.R;                                   // first produced row of this band (0 when the plane is the whole image)
  const int co_base = cs * 64;
  const int rows_here = a.Ho - r0 < a.R ? a.Ho - r0 : a.R;
  // gimg > 1 (round 5, planes of a few dozen pixels without a fused norm: VGG conv5_1 on 5 x 5 maps): `img` counts PLANES of gimg
  // consecutive images stacked in one padded raster, Hst raster rows apart (an image's rows + the zero row it shares with the next);
  // their produced pixels are consecutive in the NHWC tensor, so everything behind the K loop sees one plane of gimg * Ho * Wo pixels
  const int G = a.gimg;
  const int HW = G > 1 ? G * a.Ho * a.Wo : rows_here * a.Wo;  // pixels this workgroup produces
  const size_t pix0 = (size_t)img * G * a.Ho * a.Wo + (size_t)r0 * a.Wo;      // index of its first pixel in the produced tensor

  const T* xg = (const T*)a.x + (size_t)g * a.x_gstride;
  const T* wg = (const T*)a.w + (size_t)g * a.w_gstride;
  const i32x4 xrs = s2p_make_rsrc(xg, a.x_bytes - (unsigned)g * (unsigned)a.x_gstride * 2u);
  const i32x4 wrs = s2p_make_rsrc(wg, a.w_bytes);
  const unsigned p_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(pbase));
  const unsigned w_lds = __builtin_amdgcn_readfirstlane(s2p_lds_addr(wbase));
  const unsigned OOB = 0x80000000u;
  const int ncg = a.Cin / 32;                                  // 32-channel groups
  const int nhs = S2D ? 4 * ncg : ncg;                         // half-slabs (even: Cin % 64 == 0)

  // ---- plane DMA: 2 * NPB / 32 pieces per half-slab (piece = 32 positions x one 16-channel pair); wave w issues pieces w + 8 k.
  //      hv: source offset of this lane's 16 bytes (parity 0 in the parity form), hm: bit pp set where parity pp's pixel exists.
  int hv[PPW]; unsigned hdst[PPW]; int hm[PPW];
  {
    const float rwp = 1.0f / (float)a.WP, rhst = 1.0f / (float)a.Hst;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int ii = wave + 8 * k;
      const int cp = ii & 1, pg = ii >> 1;
      const int pos = 32 * pg + (lane >> 1);
      const int row = (int)(((float)pos + 0.5f) * rwp), col = pos - row * a.WP;      // exact for pos < 2^20
      const int gi = (int)(((float)row + 0.5f) * rhst);       // image of the stacked raster (0 when gimg == 1: Hst is huge then)
      const int yi = row - gi * a.Hst - a.PT + r0, xi = col - a.PL;      // (bands: stride-1 form only -- produced row r reads gathered rows r - PT ..)
      const bool in = yi >= 0 && yi < a.Hs && xi >= 0 && xi < a.Ws && gi < G;
      const int simg = img * G + gi;
      int m = 0;
      if constexpr (S2D) {
        if (in) m = 1 | (2 * xi + 1 < a.W ? 2 : 0) | (2 * yi + 1 < a.H ? 4 : 0) | ((2 * xi + 1 < a.W && 2 * yi + 1 < a.H) ? 8 : 0);
        hv[k] = (int)((((unsigned)(simg * a.H + 2 * yi) * a.W + 2 * xi) * a.x_pitch) * 2u + (2 * cp + (lane & 1)) * 16);
      } else {
        m = in ? 1 : 0;
        hv[k] = (int)((((unsigned)(simg * a.H + yi) * a.W + xi) * a.x_pitch) * 2u + (2 * cp + (lane & 1)) * 16);
      }
      hm[k] = m;
      hdst[k] = (unsigned)(cp * CPS + pg * 1024);
    }
  }
  // weights: 4 pieces per stage (piece = 32 co x one pair); wave w issues piece (w & 3) of the stage its set consumes
  int wv; unsigned wdst;
  {
    const int cp = wave & 1, cohalf = (wave >> 1) & 1;
    const int co = co_base + 32 * cohalf + (lane >> 1);
    wv = co < a.Cout ? (int)((unsigned)co * a.w_row * 2u + (2 * cp + (lane & 1)) * 16) : (int)OOB;
    wdst = (unsigned)(cp * 2048 + cohalf * 1024);
  }
  const int cin2 = a.Cin * 2;
  // half-slab hs -> source offset (bytes, uniform) of its channels (and parity) in the gathered tensor
  auto plane_soff = [&](int hs) {
    if constexpr (S2D) { const int pp = hs / ncg, cg = hs - pp * ncg; return (((pp >> 1) * a.W + (pp & 1)) * a.x_pitch) * 2 + cg * 64; }
    else return hs * 64;
  };
  auto issue_plane = [&](int buf, int hs, int k) {
    int v = (int)OOB;
    if (hs < nhs) {
      if constexpr (S2D) { const int pp = hs / ncg; v = ((hm[k] >> pp) & 1) ? hv[k] : (int)OOB; }
      else v = hm[k] ? hv[k] : (int)OOB;
    }
    pg_dma16(xrs, p_lds + (unsigned)(buf * PBUF) + hdst[k], v, hs < nhs ? plane_soff(hs) : 0);
  };
  // weights of raster tap t of half-slab hs
  auto issue_w = [&](int stage, int t, int hs) {
    int so = 0;
    if (hs < nhs) {
      if constexpr (S2D) { const int pp = hs / ncg, cg = hs - pp * ncg; so = a.wt[pp * 4 + t] * cin2 + cg * 64; }
      else so = a.wt[t] * cin2 + hs * 64;
    }
    pg_dma16(wrs, w_lds + (unsigned)(stage * PG_WST) + wdst, hs < nhs ? wv : (int)OOB, so);
  };

  // ---- prologue: plane of half-slab 0 (and 1 when the schedule prefetches buffer 1 a whole iteration ahead) + the first RING / 2
  //      weight stages of this wave's set
#pragma unroll
  for (int k = 0; k < PPW; ++k) issue_plane(0, 0, k);
  if constexpr (!Sched::EARLY1) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) issue_plane(1, 1, k);
  }
#pragma unroll
  for (int i = 0; i < RING / 2; ++i) { const int u = set + 2 * i; issue_w(u % RING, u % NTAP, u / NTAP); }
  // ---- fragment read bases ---------------------------------------------------------------------------------------------
  int bB[PB][TY];
  {
    const float rw = 1.0f / (float)a.Wo, rho = G > 1 ? 1.0f / (float)a.Ho : 0.f;     // (one image: yt < Ho or a band's row -> gi = 0)
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      int m = (wq * PB + j) * 16 + l15;
      if (m >= HW) m = HW - 1;                                 // padding columns of the last block: computed, never stored
      const int yt = (int)(((float)m + 0.5f) * rw), x = m - yt * a.Wo;             // row over all images of the plane
      const int gi = (int)(((float)yt + 0.5f) * rho), y = yt - gi * a.Ho + gi * a.Hst;   // -> raster row of the stacked planes
#pragma unroll
      for (int ty = 0; ty < TY; ++ty) bB[j][ty] = (q >> 1) * CPS + (q & 1) * 16 + ((y + ty) * a.WP + x) * 32;
    }
  }
  const int bA = (q >> 1) * 2048 + l15 * 32 + (q & 1) * 16;
  f32x4v acc[4][PB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

  auto read_a = [&](auto vc, auto ic, bf16x8 (&fa)[4]) {
    constexpr int v = decltype(vc)::value % NSTEP, i = decltype(ic)::value;
    fa[i] = *(const bf16x8*)(wbase + (v % RING) * PG_WST + i * 512 + bA);
  };
  auto read_b = [&](auto vc, auto jc, bf16x8 (&fb)[PB]) {
    constexpr int v = decltype(vc)::value % NSTEP, j = decltype(jc)::value;
    constexpr int t = v % NTAP, hsl = v / NTAP;
    fb[j] = *(const bf16x8*)(pbase + hsl * PBUF + (t % TX) * 32 + bB[j][t / TX]);
  };
  auto mfma4 = [&](auto jc, bf16x8 (&fa)[4], bf16x8 (&fb)[PB]) {
    constexpr int j = decltype(jc)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  // top of pair-step U: everything this wave issued before the previous pair-step has landed (what it issued IN the previous pair-step
  // -- one weight piece plus that pair-step's plane pieces -- may still be in flight), its own LDS reads have returned, then the barrier
  auto sync_top = [&](auto Uc) {
    constexpr int UP = (decltype(Uc)::value + NTAP - 1) % NTAP;
    constexpr int INFL = 1 + Sched::cnt(UP);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(INFL) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NRD = 4 + PB, RPS = (NRD + PB - 1) / PB;      // fragment reads of a step, reads issued in front of each MFMA group
  constexpr int SLOT_W = PB > 2 ? 2 : PB - 1, SLOT_P = PB > 3 ? 3 : PB - 1;
  auto run = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RING / 2 - 2) : "memory");     // the plane(s) and this set's first two stages have landed ...
    __builtin_amdgcn_s_barrier();                                             // ... for every wave
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fa[4], fb[PB];
    {
      typedef std::integral_constant<int, SET> vc;
      pg_static_for<0, 4>([&](auto ic) { read_a(vc{}, ic, fa); });
      pg_static_for<0, PB>([&](auto jc) { read_b(vc{}, jc, fb); });
    }
    for (int k2 = 0; k2 < nhs; k2 += 2) {                       // two half-slabs = NSTEP K steps = NTAP pair-steps per iteration
      pg_static_for<0, NTAP>([&](auto Uc) {
        constexpr int U = decltype(Uc)::value;
        constexpr int u = 2 * U + SET;
        typedef std::integral_constant<int, u + 2> vn;           // the step whose fragments are fetched now
        sync_top(Uc);
        bf16x8 na[4], nb[PB];
        pg_static_for<0, PB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          pg_static_for<j * RPS, ((j + 1) * RPS < NRD ? (j + 1) * RPS : NRD)>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r < 4) read_a(vn{}, std::integral_constant<int, r>{}, na);
            else read_b(vn{}, std::integral_constant<int, r - 4>{}, nb);
          });
          if constexpr (j == SLOT_W) { constexpr int u2 = u + RING; issue_w(u2 % RING, u2 % NTAP, k2 + u2 / NTAP); }
          if constexpr (j == SLOT_P && Sched::cnt(U) > 0) {
#pragma unroll
            for (int k = 0; k < Sched::cnt(U); ++k) issue_plane(Sched::buf(U), k2 + Sched::hs_ahead(U), Sched::k0(U) + k);
          }
          mfma4(jc, fa, fb);
          __builtin_amdgcn_sched_barrier(0);
        });
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

  // ---- add the two partial accumulators of a wave pair through LDS (set 0 keeps co blocks 0-1, set 1 co blocks 2-3), then
  //      bias + activation in registers and [pixel][co] staging rows (transpose through LDS) -------------------------------------
  auto finish = [&](auto ibc) {
    constexpr int IB = decltype(ibc)::value;
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
    __syncthreads();                                            // the staging rows below overlap the exchange area
    const float* bias = a.bias ? a.bias + (size_t)g * a.Cout + co_base : nullptr;
    float bv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[i][e] = bias ? bias[16 * (IB + i) + 4 * q + e] : 0.f;
    auto stage_out = [&](auto f) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const int px = (wq * PB + j) * 16 + l15;
          const f32x4v v = acc[IB + i][j];
          bf16x4 o = {(__bf16)f(v[0] + bv[i][0]), (__bf16)f(v[1] + bv[i][1]), (__bf16)f(v[2] + bv[i][2]), (__bf16)f(v[3] + bv[i][3])};
          *(bf16x4*)(smem + px * PG_ERS + (16 * (IB + i) + 4 * q) * 2) = o;
        }
    };
    if (a.act == S2P_ACT_TANH) stage_out([](float v) { return tanhf(v); });
    else if (a.act == S2P_ACT_SWISH) stage_out([](float v) { return v / (1.f + expf(-v)); });
    else if (a.act == S2P_ACT_NONE) stage_out([](float v) { return v; });
    else {
      const float ns = a.act == S2P_ACT_RELU ? 0.f : a.slope;
      stage_out([ns](float v) { return lrelu_ns(v, ns); });
    }
  };
  if (set == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 2>{});
  __syncthreads();
  T* yg = (T*)a.y + (size_t)g * a.y_gstride;
  const T* auxg = a.aux ? (const T*)a.aux + (size_t)g * a.y_gstride : nullptr;
  const T* aux2g = a.aux2 ? (const T*)a.aux2 + (size_t)g * a.y_gstride : nullptr;
  const bool epi_add = a.epi == S2P_EPI_ADD;
  const bool g_tanh = a.gact == S2P_ACT_TANH;
  const float gneg = a.gact == S2P_ACT_RELU ? 0.f : (a.gact == S2P_ACT_LRELU ? a.gslope : 1.f);
  // one (pixel row, 8-channel chunk) item of the output: staged value (+ residual / producer-activation-gradient epilogue)
  auto out_chunk = [&](int row, int ch, size_t go) {
    Chunk<T> c;
    c.raw = *(const u32x4*)(smem + row * PG_ERS + ch * 16);
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
    return c;
  };
  constexpr int MAXR = BPIX / 64;
  if constexpr (MAT == 0) {
    for (int idx = tid; idx < HW * 8; idx += 512) {
      const int row = idx >> 3, ch = idx & 7;
      const size_t go = (pix0 + row) * a.y_pitch + co_base + ch * 8;
      *(u32x4*)(yg + go) = out_chunk(row, ch, go).raw;
    }
  } else if constexpr (MAT == 1) {
    // ---- fused InstanceNorm (+ MAT modulation) + activation of the plane this workgroup owns (conv_plane.hip, MAT == 1) ---------
    const int ch = tid & 7, r0 = tid >> 3;
    const T* gbb = (GB && a.gb) ? (const T*)a.gb + pix0 * a.gb_pitch + co_base + ch * 8 : nullptr;
    Chunk<T> xv[MAXR], gv[GB ? MAXR : 1], bv[GB ? MAXR : 1];
    if constexpr (GB) {
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const int row = r0 + 64 * k;
        gv[k].raw = (u32x4){0u, 0u, 0u, 0u}; bv[k].raw = gv[k].raw;
        if (gbb && row < HW) {
          gv[k].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch);
          bv[k].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch + a.Cout);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      xv[k].raw = (u32x4){0u, 0u, 0u, 0u};
      if (row < HW) {
        const size_t go = (pix0 + row) * a.y_pitch + co_base + ch * 8;
        xv[k] = out_chunk(row, ch, go);
        if (a.y) *(u32x4*)(yg + go) = xv[k].raw;                  // (y == NULL: only the normalised tensor is wanted)
      }
    }
    // the plane unpacked ONCE, the centred values kept from the second pass for the third (conv_plane.hip, MAT == 1; round 5)
    float xf[MAXR][8];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) xv[k].unpack(xf[k]);
    __syncthreads();                                            // the staging rows are dead: LDS is scratch from here on
    float* red = (float*)smem;                                  // [8 waves][64]
    float* cst = (float*)smem + 8 * 64;                         // [4][64]: plane sum / M2, then 1 + gamma_st, beta_st
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
      for (int k = 0; k < MAXR; ++k) sacc[e] += xf[k][e];          // rows beyond HW hold zeros
    }
    plane_sum(sacc, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = cst[ch * 8 + e] * inv;
      sacc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const float d = xf[k][e] - mean[e];
        xf[k][e] = d;
        sacc[e] += (r0 + 64 * k < HW) ? d * d : 0.f;
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
    const float nns = a.n_act == S2P_ACT_RELU ? 0.f : (a.n_act == S2P_ACT_LRELU ? a.n_slope : 1.f);
    T* y2 = (T*)a.y2 + pix0 * a.y2_pitch + co_base + ch * 8;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      if (row >= HW) break;
      Chunk<T> o0;
      float ov[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = gs[e] + (GB ? gv[k].get(e) : 0.f), bb = bs[e] + (GB ? bv[k].get(e) : 0.f);
        const float xh = xf[k][e] * rstd[e];                      // (x - mean) * rstd
        const float yv = __builtin_fmaf(xh, gg, bb);              // (norm.hip: mat_value)
        ov[e] = lrelu_ns(yv, nns);
      }
      o0.pack(ov);
      *(u32x4*)(y2 + (size_t)row * a.y2_pitch) = o0.raw;
    }
  } else {
    // ---- fused backward of InstanceNorm (+ MAT modulation) + activation (conv_plane.hip, MAT == 2): the staged plane (+ the aux
    //      gradient of EPI_ADD, e.g. a feature-matching tap) is dL/d(norm output) for this (image, slab); it never goes to HBM --------
    const int ch = tid & 7, r0 = tid >> 3, lc = co_base + ch * 8;
    const T* xb = (const T*)a.xn + pix0 * a.xn_pitch + lc;
    const T* gbb = (GB && a.gb) ? (const T*)a.gb + pix0 * a.gb_pitch + lc : nullptr;
    Chunk<T> xv[MAXR], gv[GB ? MAXR : 1], bv[GB ? MAXR : 1], dv[MAXR];
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      xv[k].raw = (u32x4){0u, 0u, 0u, 0u}; dv[k].raw = xv[k].raw;
      if constexpr (GB) { gv[k].raw = xv[k].raw; bv[k].raw = xv[k].raw; }
      if (row < HW) {
        xv[k].raw = *(const u32x4*)(xb + (size_t)row * a.xn_pitch);
        if constexpr (GB) {
          if (gbb) { gv[k].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch); bv[k].raw = *(const u32x4*)(gbb + (size_t)row * a.gb_pitch + a.Cout); }
        }
        dv[k] = out_chunk(row, ch, (pix0 + row) * a.y_pitch + lc);      // rows beyond HW stay zero: they add nothing to the sums
      }
    }
    __syncthreads();                                            // the staging rows are dead: LDS is scratch from here on
    float* red = (float*)smem;                                  // [4 sums][8 waves][64]
    float* cst = (float*)smem + 4 * 8 * 64;                     // [6][64]: mean, rstd, 1 + gamma_st, beta_st, s1 / HW, s2 / HW
    if (tid < 64) {
      const int c = co_base + tid;
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
    const float nneg = a.n_act == S2P_ACT_RELU ? 0.f : (a.n_act == S2P_ACT_LRELU ? a.n_slope : 1.f);
    // the normalised input xh and the gradient dy behind the activation stay in registers for the output pass (conv_plane.hip,
    // MAT == 2; round 5: 40 -> ~23 VALU instructions per element over the two passes)
    float xhf[MAXR][8], dyf[MAXR][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cl = ch * 8 + e;
      const float m = cst[cl], r = cst[64 + cl], g1 = cst[128 + cl], b1 = cst[192 + cl];
      float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
      for (int k = 0; k < MAXR; ++k) {
        const float gg = g1 + (GB ? gv[k].get(e) : 0.f), bb = b1 + (GB ? bv[k].get(e) : 0.f);
        const float xh = (xv[k].get(e) - m) * r;
        const float yv = __builtin_fmaf(xh, gg, bb);            // (norm.hip: mat_value -- the forward's rounding)
        const float dvv = dv[k].get(e);
        const float dy = yv > 0.f ? dvv : dvv * nneg;
        const float dxh = dy * gg;
        xhf[k][e] = xh; dyf[k][e] = dy;
        q0 += dxh; q1 += dxh * xh; q2 += dy * xh; q3 += dy;
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        q0 += __shfl_xor(q0, o, 64); q1 += __shfl_xor(q1, o, 64); q2 += __shfl_xor(q2, o, 64); q3 += __shfl_xor(q3, o, 64);
      }
      if (lane < 8) { red[(0 * 8 + wave) * 64 + cl] = q0; red[(1 * 8 + wave) * 64 + cl] = q1; red[(2 * 8 + wave) * 64 + cl] = q2; red[(3 * 8 + wave) * 64 + cl] = q3; }
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
    T* dxo = (T*)a.y2 + pix0 * a.y2_pitch + lc;
    T* dgo = a.dgb ? (T*)a.dgb + pix0 * a.dgb_pitch + lc : nullptr;
    const T* rsb = a.res ? (const T*)a.res + pix0 * a.res_pitch + lc : nullptr;
    float rr2[8], g12[8], s1v[8], s2v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int cl = ch * 8 + e; rr2[e] = cst[64 + cl]; g12[e] = cst[128 + cl]; s1v[e] = cst[256 + cl]; s2v[e] = cst[320 + cl]; }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
      const int row = r0 + 64 * k;
      if (row >= HW) break;
      Chunk<T> o0, o1, o2;
      float v0[8], v1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = g12[e] + (GB ? gv[k].get(e) : 0.f);
        const float xh = xhf[k][e], dy = dyf[k][e];
        const float dxh = dy * gg;
        v0[e] = rr2[e] * (dxh - s1v[e] - xh * s2v[e]);
        v1[e] = dy * xh;
      }
      o0.pack(v0); o1.pack(v1); o2.pack(dyf[k]);
      if (rsb) {                                                // (added to the ROUNDED dx, as before)
        Chunk<T> rv; rv.raw = *(const u32x4*)(rsb + (size_t)row * a.res_pitch);
#pragma unroll
        for (int e = 0; e < 8; ++e) v0[e] = o0.get(e) + rv.get(e);
        o0.pack(v0);
      }
      *(u32x4*)(dxo + (size_t)row * a.y2_pitch) = o0.raw;
      if (dgo) {
        *(u32x4*)(dgo + (size_t)row * a.dgb_pitch) = o1.raw;
        *(u32x4*)(dgo + (size_t)row * a.dgb_pitch + a.Cout) = o2.raw;
      }
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
namespace {
struct PgShape { int TY, TX, PB, NPB, S2D; };
// instantiated shapes: 4x4 stride 1 on planes up to 64 / 192 produced pixels (the 256 -> 512 layers of both scales and their dgrads);
// 4x4 stride 2 pad 2 in the parity form (2x2 raster taps) on the same tile sizes (128 -> 256 of both scales, 64 -> 128 of the coarser);
// 3x3 stride 1 on planes of 65..128 produced pixels (VGG conv4_x on 10x10 maps and their dgrads: too small for conv_plane.hip's tile);
// 3x3 stride 1 on ROW BANDS (up to 448 pixels, 640 raster positions) of larger planes: VGG conv1_2 at 84x84 (5 rows per band),
// conv2_x at 42x42 (10 rows), the 64x64 ResBlk maps of the 256x256 rollout -- one workgroup per (image, band, 64-channel slab)
// With 64 input channels (18 K steps) the fixed part of a workgroup -- first plane fetch, accumulator exchange, staging,
// store -- is as long as its loop: those layers take 192-pixel bands (74 KB of LDS: two workgroups per CU, one's fixed part under the
// other's loop) instead of 448-pixel ones (112 KB, one per CU).
const PgShape PG_SHAPES[] = {{4, 4, 1, 128, 0}, {4, 4, 3, 256, 0}, {2, 2, 1, 128, 1}, {2, 2, 3, 256, 1}, {3, 3, 2, 256, 0}, {3, 3, 7, 640, 0},
                             {3, 3, 3, 384, 0}};
constexpr int PG_BAND_SHAPE = 5;             // the shape that takes ROW BANDS of planes too large for any whole-plane shape
constexpr int PG_BAND_SHAPE_SHORTK = 6;      // ... when the K loop is short (Cin = 64; measured: 128 -> 128 at 42x42 is 2 % faster on the large bands)
}  // namespace

static bool pg_pick_shape(PlaneGArgs& a, int TY, int TX, int s2d, int Cin, int N, bool want_mat) {
  const int HW = a.Ho * a.Wo;
  const int last = (a.Ho - 1 + TY - 1) * a.WP + (a.Wo - 1 + TX - 1);              // last raster position any tap reads
  a.shape = -1; a.R = a.Ho; a.nbands = 1; a.gimg = 1; a.Hst = 1 << 20;
  // several images per workgroup (see the kernel) for planes that fill less than half of the smallest tile: the largest power of two
  // G <= 4 that divides N, fits the tile and its raster, and leaves at least 16 planes; never with a fused norm (its statistics are
  // per image)
  if (!s2d && !want_mat && !S2P_DIAG_SWITCH(20)) {
    const int over = TY - 1 - a.PT > a.PT ? TY - 1 - a.PT : a.PT;                 // zero rows between two stacked images
    for (int G = 4; G >= 2; G >>= 1) {
      if (N % G || N / G < 16) continue;
      const int Hst = a.Hs + over, lastg = ((G - 1) * Hst + a.Ho - 1 + TY - 1) * a.WP + (a.Wo - 1 + TX - 1);
      for (int i = 0; i < (int)(sizeof(PG_SHAPES) / sizeof(PG_SHAPES[0])); ++i) {
        const PgShape& sh = PG_SHAPES[i];
        if (i == PG_BAND_SHAPE_SHORTK || sh.TY != TY || sh.TX != TX || sh.S2D != s2d) continue;
        if (2 * HW > 4 * sh.PB * 16) break;                                          // one image already fills half of this tile
        if (G * HW <= 4 * sh.PB * 16 && lastg < sh.NPB) { a.shape = i; a.gimg = G; a.Hst = Hst; return true; }
        break;                                                                      // (only the smallest tile of the tap rectangle)
      }
    }
  }
  for (int i = 0; i < (int)(sizeof(PG_SHAPES) / sizeof(PG_SHAPES[0])); ++i) {     // (ordered by tile size: the smallest that fits)
    const PgShape& s = PG_SHAPES[i];
    if (i != PG_BAND_SHAPE_SHORTK && s.TY == TY && s.TX == TX && s.S2D == s2d && HW <= 4 * s.PB * 16 && last < s.NPB) { a.shape = i; break; }
  }
  // a plane much smaller than the tile wastes the MFMAs: leave those to the generic kernels
  if (a.shape >= 0) return 2 * HW > 4 * PG_SHAPES[a.shape].PB * 16;
  // too large for a whole-plane tile: row bands of the stride-1 form (no fused norm: its statistics span the whole plane)
  if (s2d || S2P_DIAG_SWITCH(4)) return false;
  for (int pass = 0; pass < 2; ++pass) {
    const int bi = pass == 0 ? PG_BAND_SHAPE_SHORTK : PG_BAND_SHAPE;
    if (pass == 0 && (Cin > 64 || S2P_DIAG_SWITCH(8))) continue;
    const PgShape& b = PG_SHAPES[bi];
    if (b.TY != TY || b.TX != TX) continue;
    int R = (b.NPB - TX) / a.WP - (TY - 1);                                      // (R + TY - 1) * WP + TX <= NPB
    if (R * a.Wo > 4 * b.PB * 16) R = (4 * b.PB * 16) / a.Wo;
    if (R < 2 || 4 * R * a.Wo < 3 * 4 * b.PB * 16) continue;                      // at least 3/4 of the tile used
    a.shape = bi; a.R = R; a.nbands = (a.Ho + R - 1) / R;
    return true;
  }
  return false;
}

bool s2p_conv_planeg_setup(const PlaneGProblem& p, PlaneGArgs& a) {
  if (p.Cin % 64 || p.Cout % 64 || p.Cst != p.Cout || p.x_pitch % 8 || p.y_pitch % 8) return false;
  if (p.Ho < 1 || p.Wo < 1) return false;
  if (p.istride == 2) {
    // 4x4 stride 2 pad 2: in(2 y + dy, 2 x + dx), dy = 2 (a - 1) + py in [-2, 1] -> parity sub-plane (py, px), raster tap (a, b) of a 2x2
    // stride-1 correlation with one row / column of padding in front
    if (p.T != 16) return false;
    for (int t = 0; t < 16; ++t) a.wt[t] = -1;
    for (int t = 0; t < 16; ++t) {
      const int dy = (int)(signed char)(p.tap[t] & 0xff), dx = (int)(signed char)((p.tap[t] >> 8) & 0xff);
      if (dy < -2 || dy > 1 || dx < -2 || dx > 1) return false;
      const int py = dy & 1, px = dx & 1, ra = (dy + 2) >> 1, rb = (dx + 2) >> 1;
      const int k = (py * 2 + px) * 4 + ra * 2 + rb;
      if (a.wt[k] >= 0) return false;
      a.wt[k] = p.tap[t] >> 16;
    }
    a.H = p.Hi; a.W = p.Wi; a.Hs = (p.Hi + 1) / 2; a.Ws = (p.Wi + 1) / 2; a.Ho = p.Ho; a.Wo = p.Wo;
    if (p.Ho != p.Hi / 2 + 1 || p.Wo != p.Wi / 2 + 1) return false;
    a.PT = 1; a.PL = 1;
    const int PR = p.Wo - a.Ws > 0 ? p.Wo - a.Ws : 0;
    a.WP = a.Ws + (PR > 1 ? PR : 1);
    return pg_pick_shape(a, 2, 2, 1, p.Cin, p.N, p.want_mat);
  }
  if (p.istride != 1) return false;
  if (p.T < 1 || p.T > 16) return false;
  int dy0 = 127, dy1 = -127, dx0 = 127, dx1 = -127;
  for (int t = 0; t < p.T; ++t) {
    const int dy = (int)(signed char)(p.tap[t] & 0xff), dx = (int)(signed char)((p.tap[t] >> 8) & 0xff);
    if (dy < dy0) dy0 = dy; if (dy > dy1) dy1 = dy; if (dx < dx0) dx0 = dx; if (dx > dx1) dx1 = dx;
  }
  const int TY = dy1 - dy0 + 1, TX = dx1 - dx0 + 1;
  if (TY * TX != p.T || dy0 > 0 || dx0 > 0) return false;     // a full rectangle of taps, padding in front >= 0
  for (int t = 0; t < 16; ++t) a.wt[t] = -1;
  for (int t = 0; t < p.T; ++t) {
    const int dy = (int)(signed char)(p.tap[t] & 0xff), dx = (int)(signed char)((p.tap[t] >> 8) & 0xff);
    const int k = (dy - dy0) * TX + (dx - dx0);
    if (a.wt[k] >= 0) return false;
    a.wt[k] = p.tap[t] >> 16;
  }
  a.PT = -dy0; a.PL = -dx0;
  a.H = p.Hi; a.W = p.Wi; a.Hs = p.Hi; a.Ws = p.Wi; a.Ho = p.Ho; a.Wo = p.Wo;
  // columns / rows a tap reads beyond the gathered grid's last pixel; the raster row is W real columns + max(left pad, right
  // overhang) zero columns, which serve as the right pad of a row AND the left pad of the next one
  const int PR = p.Wo - p.Wi + dx1 > 0 ? p.Wo - p.Wi + dx1 : 0;
  a.WP = p.Wi + (a.PL > PR ? a.PL : PR);
  return pg_pick_shape(a, TY, TX, 0, p.Cin, p.N, p.want_mat);
}

template <int TY, int TX, int PB, int NPB, bool S2D>
static void pg_launch_shape(const PlaneGArgs& a, dim3 grid, hipStream_t st) {
  constexpr int RING = (2 * TY * TX) % 8 == 0 ? 8 : 6;          // divides the K steps of an iteration
  if (a.y2 && a.xn) hipLaunchKernelGGL((conv_planeg_kernel<TY, TX, PB, NPB, RING, 2, S2D, false>), grid, dim3(512), 0, st, a);
  else if (a.y2) hipLaunchKernelGGL((conv_planeg_kernel<TY, TX, PB, NPB, RING, 1, S2D, false>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((conv_planeg_kernel<TY, TX, PB, NPB, RING, 0, S2D, false>), grid, dim3(512), 0, st, a);
}

int s2p_conv_planeg_launch(PlaneGArgs& a, int groups, hipStream_t st) {
  a.nco = a.Cout / 64;
  a.img_xcd = (a.N % 8 == 0 && a.nbands > 1 && !S2P_DIAG_SWITCH(15)) ? 1 : 0;      // all bands of an image on one XCD (see the kernel)
  if (a.gb) S2P_FAIL(-1, "conv_planeg: gamma / beta maps are not instantiated for this kernel family");
  if (a.nbands > 1 && a.y2) S2P_FAIL(-1, "conv_planeg: the fused norm needs the whole plane in one workgroup");
  if (a.gimg > 1 && (a.y2 || a.nbands > 1 || a.N % a.gimg)) S2P_FAIL(-1, "conv_planeg: several images per plane exclude the fused norm and row bands");
  dim3 grid((a.N / a.gimg) * a.nbands * a.nco, groups);
  if (a.shape == 0) pg_launch_shape<4, 4, 1, 128, false>(a, grid, st);
  else if (a.shape == 1) pg_launch_shape<4, 4, 3, 256, false>(a, grid, st);
  else if (a.shape == 2) pg_launch_shape<2, 2, 1, 128, true>(a, grid, st);
  else if (a.shape == 3) pg_launch_shape<2, 2, 3, 256, true>(a, grid, st);
  else if (a.shape == 4) pg_launch_shape<3, 3, 2, 256, false>(a, grid, st);
  else if (a.shape == 5) pg_launch_shape<3, 3, 7, 640, false>(a, grid, st);
  else if (a.shape == 6) pg_launch_shape<3, 3, 3, 384, false>(a, grid, st);
  else S2P_FAIL(-1, "conv_planeg: no kernel instantiated for this shape (s2p_conv_planeg_setup decides)");
  S2P_CHECK_LAUNCH("conv_planeg_kernel");
  return 0;
}
