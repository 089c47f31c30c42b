// Image-fidelity metrics on device (the paper's PSNR / SSIM, `rebuttal.md:50`; no code in
// the reference => the definition is the published one, restated in oracle/metrics_oracle.py -- parity unpinned).
//   PSNR = 10 log10(R^2 / MSE)                                     per image, MSE over all C*H*W samples
//   SSIM (Wang et al. 2004): 11x11 Gaussian window (sigma 1.5, normalised), K1 = 0.01, K2 = 0.03, evaluated on the
//         (H-10) x (W-10) fully covered window positions of every channel plane, averaged per image.
// One fused pass over fp32 NCHW frames (the layout rollout() / Pix2PixModel.generated_to_nchw hand out): a workgroup
// stages a 26x26 input patch of both images in LDS, runs the separable filter for the five moments, reduces its
// 16x16 SSIM values and the squared error of the pixels it owns, and adds both into per-image accumulators.
#include "s2p_common.h"

constexpr int WIN = 11, TILE = 16, PATCH = TILE + WIN - 1;   // 26

struct MetricArgs {
  const float* a; const float* b;
  int N, C, H, W, tiles_x, tiles_y;
  float c1, c2;
  float g[WIN];
  float* sq_sum; float* ssim_sum;
};

__global__ __launch_bounds__(256) void image_metrics_kernel(const MetricArgs a) {
  __shared__ float pa[PATCH][PATCH + 1], pb[PATCH][PATCH + 1];
  __shared__ float hm[5][PATCH][TILE + 1];            // horizontal pass: mu_a, mu_b, E[aa], E[bb], E[ab]
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  const int plane = blockIdx.y, n = plane / a.C;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int oy0 = ty * TILE, ox0 = tx * TILE;
  const float* A = a.a + (size_t)plane * a.H * a.W;
  const float* B = a.b + (size_t)plane * a.H * a.W;
  // the last tile of a row / column also owns the trailing 10 input pixels (each input pixel is owned exactly once)
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

extern "C" int s2p_image_metrics(const float* a, const float* b, int N, int C, int H, int W, float data_range,
                                 float* sq_err_sum, float* ssim_sum, void* stream) {
  if (!a || !b || !sq_err_sum || !ssim_sum) S2P_FAIL(-1, "s2p_image_metrics: null pointer");
  if (N < 1 || C < 1 || H < WIN || W < WIN) S2P_FAIL(-1, "s2p_image_metrics: images must be at least %dx%d", WIN, WIN);
  if (!(data_range > 0.f)) S2P_FAIL(-1, "s2p_image_metrics: data_range must be positive");
  MetricArgs m{};
  m.a = a; m.b = b; m.N = N; m.C = C; m.H = H; m.W = W;
  m.tiles_y = (H - (WIN - 1) + TILE - 1) / TILE; m.tiles_x = (W - (WIN - 1) + TILE - 1) / TILE;
  m.c1 = (0.01f * data_range) * (0.01f * data_range); m.c2 = (0.03f * data_range) * (0.03f * data_range);
  double s = 0.0, g[WIN];
  for (int k = 0; k < WIN; ++k) { const double d = k - (WIN - 1) / 2; g[k] = exp(-d * d / (2.0 * 1.5 * 1.5)); s += g[k]; }
  for (int k = 0; k < WIN; ++k) m.g[k] = (float)(g[k] / s);
  m.sq_sum = sq_err_sum; m.ssim_sum = ssim_sum;
  if ((long long)N * C > 65535) S2P_FAIL(-1, "s2p_image_metrics: more than 65535 planes in one call");
  hipLaunchKernelGGL(image_metrics_kernel, dim3(m.tiles_x * m.tiles_y, N * C), dim3(256), 0, (hipStream_t)stream, m);
  S2P_CHECK_LAUNCH("image_metrics_kernel");
  return 0;
}
