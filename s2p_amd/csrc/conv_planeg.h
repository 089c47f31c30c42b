// Generalised plane-resident convolution (conv_planeg.hip, round 4): host-side argument block + entry points.
// Same decomposition as conv_plane.hip (one workgroup = one image x one 64-channel slab of the output, the image's zero-padded
// input plane of the current 32-channel half-slab resident in LDS, K split over wave pairs, weight ring by LDS-DMA), for
//   * any TY x TX tap rectangle on a stride-1 raster (the PatchGAN 4x4 pad-2 layers and their dgrads), and
//   * stride-2 4x4 convolutions as a 2x2 stride-1 convolution over the four PARITY sub-planes of the input (space-to-depth done
//     by the LDS-DMA's per-lane source addresses: a "half-slab" is then one parity x 32 channels),
// with the padded row width a run-time value and the InstanceNorm (+ LeakyReLU) forward / backward in the epilogue.
#pragma once
#include <hip/hip_runtime.h>

struct PlaneGArgs {
  const void* x; const void* w; const float* bias; const void* aux; const void* aux2; void* y;
  int N;
  int H, W;                          // gathered tensor's grid (full resolution)
  int Ho, Wo;                        // produced grid
  int WP, PT, PL;                    // padded raster: row width in positions, rows / columns of padding in front
  int Hs, Ws;                        // grid the raster holds: H, W (stride 1) or ceil(H/2), ceil(W/2) (parity sub-planes)
  int Cin, x_pitch, x_gstride;
  int Cout, Cst, y_pitch, y_gstride;
  int w_row;                         // elements per packed weight row ([Cout][tap][Cin])
  long long w_gstride;
  int wt[16];                        // packed-weight tap index: stride 1: of raster tap ty*TX + tx; parity form: of (parity pp, raster tap t) at [pp*4 + t]
  int act, epi, gact;
  float slope, gslope;
  unsigned x_bytes, w_bytes;
  // fused InstanceNorm forward of the OUTPUT plane (y2 != NULL, xn == NULL) / backward of the norm that fed this dgrad's forward
  // conv (xn != NULL): see conv_plane.h; gamma / beta maps are optional (plain InstanceNorm when NULL)
  void* y2; int y2_pitch;
  const void* gb; int gb_pitch;
  const float* gbst; int gbst_pitch;
  float* stats; int n_act; float n_slope, eps;
  const void* xn; int xn_pitch; void* dgb; int dgb_pitch; float* dgbst; int dgbst_pitch; const void* res; int res_pitch;
  int nco, img_xcd;               // set by the launcher: Cout / 64; all row bands of an image on one XCD
  int R, nbands;                     // produced rows per workgroup and bands per image (R = Ho, 1: the whole plane)
  int shape;                         // index of the instantiated tile shape (set by s2p_conv_planeg_setup)
  int gimg, Hst;                     // images stacked in one plane (1: Hst = 2^20) and raster rows from one image to the next
};

// which == 0: would the kernel take this problem?  (no launch)
struct PlaneGProblem {
  int N, Hi, Wi, Ho, Wo, Cin, Cout, Cst, x_pitch, y_pitch, istride, T;
  const int* tap;                    // GatherArgs::tap entries: (widx << 16) | ((dx & 0xff) << 8) | (dy & 0xff)
  bool want_mat;                     // the caller will ask for a fused norm: one image per plane
};
bool s2p_conv_planeg_setup(const PlaneGProblem& p, PlaneGArgs& a);      // fills the geometry fields of `a`; false: not applicable
int s2p_conv_planeg_launch(PlaneGArgs& a, int groups, hipStream_t st);
