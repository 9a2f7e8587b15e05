// Launch parameters of the fused Bottleneck kernels (bottleneck.hip, bottleneck_asm.hip).
#pragma once
#include "cft_common.h"

struct Bneck128Params {
  const unsigned char* x;
  const unsigned char* w1;   // [128][kpad1]
  const unsigned char* w2;   // [128][kpad2], k = (kh*3 + kw)*128 + ci
  const unsigned char* w2s;  // the same weights as 36 stage images of 8 KiB (cft_bottleneck_pack_w2), or null
  const float* b1;
  const float* b2;
  unsigned char* y;
  int ldx, xoff, ldy, yoff, kpad1, kpad2;
  int H, W, tiles_x, tiles_y, ntiles, shortcut;
};

// bottleneck_asm.hip: the 128-channel Bottleneck on 16 x 16-pixel tiles with the hand-scheduled 3x3 loop (dtype CFT_BF16 / CFT_F16)
int bneck128_asm_launch(const Bneck128Params& p, int dtype, hipStream_t stream);
