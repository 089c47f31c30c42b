// Plane-resident 3x3 "same" convolution (conv_plane.hip): host-side argument block + entry points.
#pragma once
#include <hip/hip_runtime.h>

// optional fused InstanceNorm + MAT epilogue (see PlaneArgs)
struct PlaneMat {
  void* y2; int y2_pitch; const void* gb; int gb_pitch; const float* gbst; int gbst_pitch; float* stats; int act; float slope, eps;
  const void* xn; int xn_pitch; void* dgb; int dgb_pitch; float* dgbst; int dgbst_pitch; const void* res; int res_pitch;   // backward form
};

struct PlaneArgs {
  const void* x; const void* w; const float* bias; const void* aux; const void* aux2; void* y;
  int N, H, W;                       // output grid == input grid (stride 1, pad 1)
  int Cin, x_pitch, x_gstride;       // per-group input channels (multiple of 64), tensor pitch, channels between groups
  int Cout, Cst, y_pitch, y_gstride;
  int w_row;                         // elements per packed weight row ([Cout][tap][Cin])
  long long w_gstride;
  int wt[9];                         // packed-weight tap index of the geometric tap (dy+1)*3 + (dx+1)
  int act, epi, gact;
  float slope, gslope;
  unsigned x_bytes, w_bytes;
  // fused InstanceNorm + MAT modulation of the OUTPUT plane (y2 != NULL; groups must be 1): y2 = n_act(xhat * (1 + g_img + g_st)
  // + b_img + b_st) with the statistics of y (after the residual epilogue); stats is written in norm.hip's format
  void* y2; int y2_pitch;
  const void* gb; int gb_pitch;      // [N, HW, gb_pitch]: gamma at channel 0, beta at channel Cout (may be NULL)
  const float* gbst; int gbst_pitch; // fp32 [N][gbst_pitch]: gamma at [0, Cout), beta at [Cout, 2 Cout) (may be NULL)
  float* stats; int n_act; float n_slope, eps;
  // fused BACKWARD of that norm (xn != NULL; this launch is the dgrad of the conv the norm fed): the conv result is
  // dL/d(norm output) and is not stored (y unused); y2 = dL/d(xn) + res, dgb / dgbst = d(gamma | beta) of the image map /
  // the state affine; stats is read (the forward's)
  const void* xn; int xn_pitch; void* dgb; int dgb_pitch; float* dgbst; int dgbst_pitch; const void* res; int res_pitch;
  int nco, gxcd;                     // set by the launcher: Cout / 64; grouped launch with (group, image half) units dealt to the XCDs
  int diag;                          // timing ablations (diagnostics build only)
};

bool s2p_conv_plane_applicable(const PlaneArgs& a);
int s2p_conv_plane_launch(PlaneArgs& a, int groups, hipStream_t st);
