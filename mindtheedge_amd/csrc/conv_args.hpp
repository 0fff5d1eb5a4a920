// Argument block of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm8.hip).
#pragma once
#include "common.hpp"

struct ConvArgs {
    const void* x; long ldx;        // input pixels, elements per pixel (>= Cin_p)
    const void* w;                  // [N][taps][Cin_p] packed weights, same element type as x
    const float* bias;              // [N] or null
    void* y; long ldy; int out_f32; // output pixels
    int B, H, W, Cin_p, N, KH, KW;
    long M;
    int splits; float* ws;          // split-K: split s STORES its partial sums into its own slab ws[s][M][N] (fp32); splitk_finish_kernel adds
                                    // the slabs in order -- no floating-point atomics, the result does not depend on the arrival order
    int accum;                      // 1: y += conv -- second gradient of a two-consumer activation.  Every tile form stages conv + bias in the OUTPUT type
                                    // (its LDS image / packed registers), then adds the old value in fp32 and rounds again: two roundings, the same in
                                    // all forms (they stay bit-identical with each other); only fp32 output (out_f32) rounds nothing
    int solo;                       // host-side hint (MTE_CONV_SOLO): nothing runs beside this launch on another stream
    // Sparse form (SAN branch, round 3): GEMM row m is pixel rows[m] of the dense NHWC maps, for m < *nrows (device-side count).  The
    // input map is zero-filled off the active set, so gathering a row's taps from it IS the sparse convolution's sum over active
    // neighbours; outputs are scattered to the same sites, nothing is written elsewhere.  Work scales with the active count: tiles past
    // *nrows return at once (the grid is sized for the dense capacity M, the count never visits the host).
    const int* rows; const int* nrows;
    // K order of the 8-phase kernels (conv_igemm8.hip; every other tile form runs tap-major): 0 = filter tap outer, input channels inner (the order
    // of the weight pack); 1 (round 5, Cin_p % 64 == 0) = 64-CHANNEL SLICE outer, taps inner -- the nine taps of a slice re-touch the lines the first
    // tap brought into the XCD's L2 (a slice of the resident tiles' inputs fits it; a whole tap sweep over all channels does not: 9 x the input was
    // fetched from the Infinity Cache).  A different summation order: results agree with the tap-major forms to fp32 rounding, not bit for bit.
    int kslice;
    // Un-shuffled output (round 6; mte_conv2d_igemm_unshuffle): the N = 4 C output channels are the packed depths d = 4 c + s of a pixel-unshuffled tensor and y is
    // that tensor BEFORE the unshuffle, [B][2H][2W][C] with pixel stride ldy: GEMM row (b, h, w), column d goes to y[b][2h + s / 2][2w + s % 2][c].  0 = off, else C.
    // Only the tile forms that stage their result in LDS write it (one workgroup holds a packed pixel's whole 32-depth groups); no K split.
    int unshuffle_c;
};

// conv_igemm8.hip: the 8-phase 256-row tile kernels (bf16, buffer-descriptor LDS-DMA).  bn = 256 or 128 output columns per tile.
// Returns MTE_ERR_UNSUPPORTED when the shape is outside what the kernel covers (the caller then takes the older tile forms).
__attribute__((visibility("hidden"))) int igemm8_launch(ConvArgs a, int bn, hipStream_t st);
