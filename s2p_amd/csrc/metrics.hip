// Image-fidelity metrics on device (the paper's PSNR / SSIM, `rebuttal.md:50`; no code in
// the reference => the definition is the published one, restated in oracle/metrics_oracle.py -- parity unpinned).
//   PSNR = 10 log10(R^2 / MSE)                                     per image, MSE over all C*H*W samples
//   SSIM (Wang et al. 2004): 11x11 Gaussian window (sigma 1.5, normalised), K1 = 0.01, K2 = 0.03, evaluated on the
//         (H-10) x (W-10) fully covered window positions of every channel plane, averaged per image.
// One fused pass over fp32 NCHW frames (the layout rollout() / Pix2PixModel.generated_to_nchw hand out).
//
// The kernel uses NO LDS and no cross-lane instruction (round 4).  A lane owns one output column x of a 64-column strip and
// walks down a segment of rows: per input row it forms the five horizontally filtered moments (a, b, aa, bb, ab) from 11 + 11
// global loads (neighbouring lanes read overlapping, cache-resident words), keeps the last 11 rows of them in a register
// window and applies the vertical filter from registers.  Round 3's kernel staged 26x26 patches and the horizontal pass in
// LDS; its SSIM sums came out ~1 % low whenever the slab weight-gradient kernel shared the CU -- that
// version is kept in the diagnostics build only, as the subject of the tile dump that located the loss.
#include "s2p_common.h"

constexpr int WIN = 11;
constexpr int SEG = 16;                         // output rows per wave (SEG + 10 input rows are filtered horizontally)

struct MetricArgs {
  const float* a; const float* b;
  int N, C, H, W, tiles_x, tiles_y;
  float c1, c2;
  float g[WIN];
  float* sq_sum; float* ssim_sum;
};

// DIV (diagnostics build): how the SSIM quotient is formed -- 0: the compiler's fp32 division (v_rcp_f32, ONE wait state, then the
// dependent v_fma); 1 / 4 / 5: v_rcp_f32, then 16 / 2 / 4 wait states before its first use, Newton-refined; 2: no transcendental
// instruction at all (integer seed + Newton steps).  tests/tools/repro_metrics_values.py runs them beside the slab kernel.
template <int DIV>
__device__ __forceinline__ float ssim_div(float n, float d) {
  if constexpr (DIV == 0) return n / d;
  else if constexpr (DIV == 2) {
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
#pragma unroll
    for (int i = 0; i < 4; ++i) r = r * __builtin_fmaf(-d, r, 2.f);
    float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
  } else {
    float r = __builtin_amdgcn_rcpf(d);
    if constexpr (DIV == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(r));
    if constexpr (DIV == 4) asm volatile("s_nop 1" : "+v"(r));
    if constexpr (DIV == 5) asm volatile("s_nop 3" : "+v"(r));
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
    float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
  }
}

template <bool MAP, int DIV = 0, bool CHK = false>
__global__ __launch_bounds__(64) void image_metrics_kernel(const MetricArgs a, float* map, unsigned* chk = nullptr) {
  const int lane = threadIdx.x;
  const int plane = blockIdx.y, n = plane / a.C;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int OW = a.W - (WIN - 1), OH = a.H - (WIN - 1);
  const int x = tx * 64 + lane;
  const bool col_ok = x < OW;
  const int xs = col_ok ? x : 0;                              // idle lanes read column 0 and contribute nothing
  const int y0 = ty * SEG;
  const int y1 = y0 + SEG < OH ? y0 + SEG : OH;               // output rows [y0, y1)
  const float* A = a.a + (size_t)plane * a.H * a.W + xs;
  const float* B = a.b + (size_t)plane * a.H * a.W + xs;
  // squared error: input pixel (r, c) is counted by the lane whose column is c at tap 0, columns c >= OW by the lane of
  // column OW - 1 at taps 1..10; rows are owned by the segment that starts at them, the last segment also owns the
  // trailing 10 rows
  const bool last_col = col_ok && x == OW - 1;
  const int own_r1 = (ty == a.tiles_y - 1) ? a.H : y1;
  float g[WIN];
#pragma unroll
  for (int k = 0; k < WIN; ++k) g[k] = a.g[k];
  float win[WIN][5];
#pragma unroll
  for (int i = 0; i < WIN; ++i)
#pragma unroll
    for (int s = 0; s < 5; ++s) win[i][s] = 0.f;
  float sq = 0.f, ss = 0.f;
  const int rows_in = y1 - y0 + (WIN - 1);
  // rows are consumed in groups of WIN so that the register window is indexed statically
  for (int rb = 0; rb < rows_in; rb += WIN) {
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
      const int r = rb + i;                                   // input row y0 + r goes into window slot i
      if (r < rows_in) {
        const float* ar = A + (size_t)(y0 + r) * a.W;
        const float* br = B + (size_t)(y0 + r) * a.W;
        float va[WIN], vb[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) { va[k] = ar[k]; vb[k] = br[k]; }
        if constexpr (CHK) {
          // diagnostics: the images hold their own element index (a) / index + 0.5 (b): every loaded value names the address it
          // came from.  Log {k | which << 8, lane, value bits, expected bits, HW_ID, row} for the first mismatches.
          const float base = (float)(plane * a.H * a.W + (y0 + r) * a.W + xs);
#pragma unroll
          for (int k = 0; k < WIN; ++k) {
            const float ea = base + (float)k, eb = ea + 0.5f;
            if (va[k] != ea || vb[k] != eb) {
              const unsigned n = atomicAdd(chk, 1u);
              if (n < 128) { unsigned* o = chk + 1 + 8 * n; o[0] = (unsigned)k | (va[k] != ea ? 0u : 256u); o[1] = (unsigned)lane;
                             o[2] = __float_as_uint(va[k] != ea ? va[k] : vb[k]); o[3] = __float_as_uint(va[k] != ea ? ea : eb);
                             o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[5] = (unsigned)(y0 + r); o[6] = (unsigned)plane; o[7] = (unsigned)tx; }
            }
          }
        }
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
          const float w = g[k];
          m0 += w * va[k]; m1 += w * vb[k]; m2 += w * va[k] * va[k]; m3 += w * vb[k] * vb[k]; m4 += w * va[k] * vb[k];
        }
        win[i][0] = m0; win[i][1] = m1; win[i][2] = m2; win[i][3] = m3; win[i][4] = m4;
        if (col_ok && y0 + r < own_r1) {
          const float d0 = va[0] - vb[0];
          sq += d0 * d0;
          if (last_col) {
#pragma unroll
            for (int k = 1; k < WIN; ++k) { const float d = va[k] - vb[k]; sq += d * d; }
          }
        }
        // the window now holds input rows r - 10 .. r (slot of row r - 10 + k is (i + 1 + k) % WIN): output row y0 + r - 10
        if (r >= WIN - 1 && col_ok) {
          float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
#pragma unroll
          for (int k = 0; k < WIN; ++k) {
            const float w = g[k];
            const int sl = (i + 1 + k) % WIN;
            v0 += w * win[sl][0]; v1 += w * win[sl][1]; v2 += w * win[sl][2]; v3 += w * win[sl][3]; v4 += w * win[sl][4];
          }
          const float sa = v2 - v0 * v0, sb = v3 - v1 * v1, cov = v4 - v0 * v1;
          const float sv = ssim_div<DIV>((2.f * v0 * v1 + a.c1) * (2.f * cov + a.c2), (v0 * v0 + v1 * v1 + a.c1) * (sa + sb + a.c2));
          ss += sv;
          if constexpr (MAP) map[((size_t)plane * OH + (y0 + r - (WIN - 1))) * OW + x] = sv;      // diagnostics build: per-position SSIM
        }
      }
    }
  }
  // per-image sums: fp32 atomics (reported values only; every lane adds its own partial -- no cross-lane instruction)
  if (col_ok) { atomicAdd(a.sq_sum + n, sq); atomicAdd(a.ssim_sum + n, ss); }
}

static int metrics_fill(MetricArgs& m, const float* a, const float* b, int N, int C, int H, int W, float data_range,
                        float* sq_err_sum, float* ssim_sum, int tile_w, int tile_h) {
  if (!a || !b || !sq_err_sum || !ssim_sum) S2P_FAIL(-1, "s2p_image_metrics: null pointer");
  if (N < 1 || C < 1 || H < WIN || W < WIN) S2P_FAIL(-1, "s2p_image_metrics: images must be at least %dx%d", WIN, WIN);
  if (!(data_range > 0.f)) S2P_FAIL(-1, "s2p_image_metrics: data_range must be positive");
  m.a = a; m.b = b; m.N = N; m.C = C; m.H = H; m.W = W;
  m.tiles_y = (H - (WIN - 1) + tile_h - 1) / tile_h; m.tiles_x = (W - (WIN - 1) + tile_w - 1) / tile_w;
  m.c1 = (0.01f * data_range) * (0.01f * data_range); m.c2 = (0.03f * data_range) * (0.03f * data_range);
  double s = 0.0, g[WIN];
  for (int k = 0; k < WIN; ++k) { const double d = k - (WIN - 1) / 2; g[k] = exp(-d * d / (2.0 * 1.5 * 1.5)); s += g[k]; }
  for (int k = 0; k < WIN; ++k) m.g[k] = (float)(g[k] / s);
  m.sq_sum = sq_err_sum; m.ssim_sum = ssim_sum;
  if ((long long)N * C > 65535) S2P_FAIL(-1, "s2p_image_metrics: more than 65535 planes in one call");
  return 0;
}

extern "C" int s2p_image_metrics(const float* a, const float* b, int N, int C, int H, int W, float data_range,
                              float* sq_err_sum, float* ssim_sum, void* stream) {
  MetricArgs m{};
  if (int rc = metrics_fill(m, a, b, N, C, H, W, data_range, sq_err_sum, ssim_sum, 64, SEG)) return rc;
  hipLaunchKernelGGL(image_metrics_kernel<false>, dim3(m.tiles_x * m.tiles_y, N * C), dim3(64), 0, (hipStream_t)stream, m, (float*)nullptr);
  S2P_CHECK_LAUNCH("image_metrics_kernel");
  return 0;
}

#ifdef S2P_DIAG_BUILD
// ---- diagnostics build only: round 3's LDS-staged kernel, with a dump of its LDS tiles --------------------------------------
// dump (optional): per workgroup [pa 26x27 | pb 26x27 | hm 5x26x17] floats copied out right after the barrier that publishes
// them; hw (optional): per workgroup {HW_ID, LDS_ALLOC, XCC_ID, 0}.  tests/tools/repro_metrics_dump.py runs it once quiet and once
// beside the slab weight-gradient kernel and diffs the tiles.
constexpr int TILE = 16, PATCH = TILE + WIN - 1;   // 26
constexpr int DUMP_FLOATS = 2 * PATCH * (PATCH + 1) + 5 * PATCH * (TILE + 1);

__global__ __launch_bounds__(256) void image_metrics_lds_kernel(const MetricArgs a, float* dump, unsigned* hw) {
  __shared__ float pa[PATCH][PATCH + 1], pb[PATCH][PATCH + 1];
  __shared__ float hm[5][PATCH][TILE + 1];            // horizontal pass: mu_a, mu_b, E[aa], E[bb], E[ab]
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int plane = blockIdx.y, n = plane / a.C;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int oy0 = ty * TILE, ox0 = tx * TILE;
  const float* A = a.a + (size_t)plane * a.H * a.W;
  const float* B = a.b + (size_t)plane * a.H * a.W;
  const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (hw && tid == 0) {
    hw[wg * 4 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    hw[wg * 4 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 6);      // HW_REG_LDS_ALLOC
    hw[wg * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    hw[wg * 4 + 3] = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)&pa[0][0]);
  }
  const int own_h = (ty == a.tiles_y - 1) ? a.H - oy0 : TILE, own_w = (tx == a.tiles_x - 1) ? a.W - ox0 : TILE;
  float sq = 0.f;
  for (int i = tid; i < PATCH * PATCH; i += 256) {
    const int r = i / PATCH, c = i - r * PATCH;
    const int y = oy0 + r, x = ox0 + c;
    float va = 0.f, vb = 0.f;
    if (y < a.H && x < a.W) { va = A[(size_t)y * a.W + x]; vb = B[(size_t)y * a.W + x]; }
    pa[r][c] = va; pb[r][c] = vb;
    if (r < own_h && c < own_w && y < a.H && x < a.W) { const float d = va - vb; sq += d * d; }
  }
  __syncthreads();
  if (dump) {
    float* d = dump + wg * DUMP_FLOATS;
    for (int i = tid; i < PATCH * (PATCH + 1); i += 256) { d[i] = (&pa[0][0])[i]; d[PATCH * (PATCH + 1) + i] = (&pb[0][0])[i]; }
  }
  for (int i = tid; i < PATCH * TILE; i += 256) {
    const int r = i / TILE, c = i - r * TILE;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      const float w = a.g[k], va = pa[r][c + k], vb = pb[r][c + k];
      m0 += w * va; m1 += w * vb; m2 += w * va * va; m3 += w * vb * vb; m4 += w * va * vb;
    }
    hm[0][r][c] = m0; hm[1][r][c] = m1; hm[2][r][c] = m2; hm[3][r][c] = m3; hm[4][r][c] = m4;
  }
  __syncthreads();
  if (dump) {
    float* d = dump + wg * DUMP_FLOATS + 2 * PATCH * (PATCH + 1);
    for (int i = tid; i < 5 * PATCH * (TILE + 1); i += 256) d[i] = (&hm[0][0][0])[i];
  }
  float ss = 0.f;
  {
    const int r = tid / TILE, c = tid - r * TILE;
    if (oy0 + r < a.H - (WIN - 1) && ox0 + c < a.W - (WIN - 1)) {
      float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
      for (int k = 0; k < WIN; ++k) {
        const float w = a.g[k];
        m0 += w * hm[0][r + k][c]; m1 += w * hm[1][r + k][c]; m2 += w * hm[2][r + k][c];
        m3 += w * hm[3][r + k][c]; m4 += w * hm[4][r + k][c];
      }
      const float va = m2 - m0 * m0, vb = m3 - m1 * m1, cov = m4 - m0 * m1;
      ss = ((2.f * m0 * m1 + a.c1) * (2.f * cov + a.c2)) / ((m0 * m0 + m1 * m1 + a.c1) * (va + vb + a.c2));
    }
  }
  sq = wave_sum(sq); ss = wave_sum(ss);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sq; red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(a.sq_sum + n, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(a.ssim_sum + n, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

extern "C" __attribute__((visibility("default"))) int s2p_diag_image_metrics_lds(
    const float* a, const float* b, int N, int C, int H, int W, float data_range, float* sq_err_sum, float* ssim_sum,
    float* dump, unsigned* hw, void* stream) {
  MetricArgs m{};
  if (int rc = metrics_fill(m, a, b, N, C, H, W, data_range, sq_err_sum, ssim_sum, TILE, TILE)) return rc;
  hipLaunchKernelGGL(image_metrics_lds_kernel, dim3(m.tiles_x * m.tiles_y, N * C), dim3(256), 0, (hipStream_t)stream, m, dump, hw);
  S2P_CHECK_LAUNCH("image_metrics_lds_kernel");
  return 0;
}
extern "C" __attribute__((visibility("default"))) int s2p_diag_image_metrics_dump_floats(void) { return DUMP_FLOATS; }
// the product kernel + a map of every SSIM value it adds: [N*C][H-10][W-10]
extern "C" __attribute__((visibility("default"))) int s2p_diag_image_metrics_map(
    const float* a, const float* b, int N, int C, int H, int W, float data_range, float* sq_err_sum, float* ssim_sum, float* map, int div_mode,
    void* stream) {
  MetricArgs m{};
  if (int rc = metrics_fill(m, a, b, N, C, H, W, data_range, sq_err_sum, ssim_sum, 64, SEG)) return rc;
  const dim3 grid(m.tiles_x * m.tiles_y, N * C);
  hipStream_t st = (hipStream_t)stream;
  if (div_mode == 1) hipLaunchKernelGGL((image_metrics_kernel<true, 1>), grid, dim3(64), 0, st, m, map);
  else if (div_mode == 2) hipLaunchKernelGGL((image_metrics_kernel<true, 2>), grid, dim3(64), 0, st, m, map);
  else if (div_mode == 4) hipLaunchKernelGGL((image_metrics_kernel<true, 4>), grid, dim3(64), 0, st, m, map);
  else if (div_mode == 5) hipLaunchKernelGGL((image_metrics_kernel<true, 5>), grid, dim3(64), 0, st, m, map);
  else if (div_mode == 100) hipLaunchKernelGGL((image_metrics_kernel<true, 0, true>), grid, dim3(64), 0, st, m, map + (size_t)N * C * (H - 10) * (W - 10), (unsigned*)map);
  else hipLaunchKernelGGL((image_metrics_kernel<true, 0>), grid, dim3(64), 0, st, m, map);
  S2P_CHECK_LAUNCH("image_metrics_kernel<map>");
  return 0;
}
#endif
