/* mte_kernels.h -- C ABI of libmte_hip.so, the gfx950 (MI355X) kernel library underneath mindtheedge_amd.
 *
 * The reference (liortalker/MindTheEdge) has no native code and no FFI: every op below replaces a PyTorch
 * module call of the reference's hot path.  Each entry cites the reference call site it stands in for
 * (paths relative to packnet_code/packnet_sfm/).  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted
 *   - caller owns and pre-allocates every buffer (including scratch); no hidden allocation, no host sync;
 *     kernels are enqueued on `stream` and are re-entrant
 *   - return value: 0 = MTE_OK, negative = error (never throws):
 *       -1 MTE_ERR_ARG  -2 MTE_ERR_LAUNCH  -3 MTE_ERR_UNSUPPORTED
 *   - activations are NHWC ("pixel-major"): element (b,y,x,c) at ((b*H + y)*W + x)*ld + c, where `ld`
 *     (elements per pixel) >= C lets a tensor be a channel slice of a wider buffer (decoder concat buffers)
 *   - dtype: 0 = bf16 (raw 16-bit), 1 = fp32.  Channel counts / ld / slice offsets are multiples of 8
 *   - loss maps are fp32 [B,H,W] (the reference's [B,1,H,W])
 */
#ifndef MTE_KERNELS_H
#define MTE_KERNELS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mte_stream_t; /* hipStream_t */

#define MTE_OK 0
#define MTE_ERR_ARG (-1)
#define MTE_ERR_LAUNCH (-2)
#define MTE_ERR_UNSUPPORTED (-3)
#define MTE_DT_BF16 0
#define MTE_DT_F32 1

/* ---- convolution: nn.Conv2d(k, stride 1) + ConstantPad2d(k//2)  (networks/layers/packnet/layers01.py:29-31,61,116-117)
 * y = conv(x, wpack) + bias.  wpack = [N][KH*KW][Cin_p] in `dtype` (see mte_pack_conv_weights).
 * Also the data-gradient: call with the "backward" pack and x := dy.
 * workspace (nullable): fp32 scratch of >= 2*B*H*W*N elements; when given, shapes with few output tiles and a long
 * reduction (pack4/pack5.conv at low resolution) are split along K over up to min(8, workspace_elems / (B*H*W*N)) workgroups
 * per tile.  Every split STORES its partial tile into its own [M][N] slab and a finish kernel adds the slabs in split order:
 * no floating-point atomics, the result is bit-reproducible and the workspace needs no clearing.
 * accumulate = 1: y += conv(x) (conv + bias rounded to the activation type, added to the old value in fp32, rounded again -- every tile form
 * alike): the second data gradient of an activation with two consumers
 * lands in the first one's buffer instead of going through a separate add.  `accumulate` is a bit set: MTE_CONV_ACCUMULATE (1) and
 * MTE_CONV_SOLO (2) = no kernel of another stream is expected to run beside this launch (forward pass, inference), so the 256 x 128
 * tile may use its 3-slot-ring variant that puts two workgroups on a CU and claims 148 of the 160 KB of LDS: measured 15-20 % faster
 * on the short-reduction layers when alone, but slower for the step when the weight-gradient stream needs LDS on the same CUs. */
#define MTE_CONV_ACCUMULATE 1
#define MTE_CONV_SOLO 2
int mte_conv2d_igemm(const void* x, long ldx, const void* wpack, const float* bias, void* y, long ldy, int out_f32,
                     int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype,
                     float* workspace, long workspace_elems, int accumulate, mte_stream_t stream);
/* Round 6: the data gradient of a folded pack layer (reference: networks/layers/packnet/layers01.py:214-250, PackLayerConv3d backward) written WITHOUT the
 * mte_pixel_shuffle pass behind it.  The N = 4 C output channels are the packed depths d = 4 c + s of the [B][H][W] packed grid; y is the un-shuffled tensor
 * [B][2H][2W][C] (pixel stride ldy): GEMM row (b, h, w), column d goes to y[b][2h + s / 2][2w + s % 2][c].  accumulate as in mte_conv2d_igemm (bit 0).  bf16.
 * Returns MTE_ERR_UNSUPPORTED where the launch would not take a tile form that stages its result in LDS (no K split, no 8-phase kernel): the caller then runs
 * mte_conv2d_igemm + mte_pixel_shuffle -- the results are bit-identical. */
int mte_conv2d_igemm_unshuffle(const void* x, long ldx, const void* wpack, void* y, long ldy,
                               int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype, int accumulate, mte_stream_t stream);
/* Library options.  MTE_OPT_GN_PREZEROED (0): when 1, the GroupNorm statistics / reduction / bias-gradient buffers handed to
 * mte_gn_stats and mte_gn_elu_bwd -- and, since round 4, that call's dgamma / dbeta -- are already zero (the caller clears one arena
 * per step with a single fill; a zeroed flat gradient buffer qualifies) and the library skips its own per-call fills. */
#define MTE_OPT_GN_PREZEROED 0
/* MTE_OPT_LOSS_PREZEROED (1): when 1, `work` of mte_edge_loss_multi_fwd and `sums` of mte_edge_loss_fwd are zero on entry (first
 * mte_edge_loss_work_elems / _sums_elems doubles; same arena) and the launch's own fill is skipped. */
#define MTE_OPT_LOSS_PREZEROED 1
/* MTE_OPT_HANDOFF_FENCES (2), round 5: when 1, the last-arriver hand-offs of the read-only reduction kernels (mte_gn_stats, the fused loss
 * forward) draw their arrival ticket behind an agent-scope RELEASE fence and the last arriver issues an agent-scope ACQUIRE before it reads
 * the other workgroups' records.  Off, the protocol is the one MI355X_MICROARCH.md's visibility table lists as measured-valid: records by
 * returning atomic exchanges (performed at the memory side), a drained wait, a relaxed agent-scope ticket, agent-scope atomic loads. */
#define MTE_OPT_HANDOFF_FENCES 2
/* MTE_OPT_WGRAD_SHARES_CHIP (3), round 5: 1 = the caller queues the weight-gradient launches (mte_conv2d_wgrad, mte_conv2d_patch_wgrad) on a stream of their own
 * beside the data-gradient chain; those kernels then aim for half a chip of workgroups, which is faster for the STEP (same box 23.59 -> 23.1 ms) although each launch
 * is slower on its own.  0 (default): one workgroup (group) per CU, the right geometry when nothing runs beside them. */
#define MTE_OPT_WGRAD_SHARES_CHIP 3
int mte_set_option(int option, int value);
/* Device error word, round 5.  Kernels whose workgroups wait for each other inside a launch (the GroupNorm cluster kernels) bound that wait;
 * a wait that gives up sets a flag in one word of pinned host memory instead of going on silently with incomplete statistics.
 * mte_device_error_init: allocate the word (once, outside stream capture).  mte_device_error_poll: -> flags set since the last poll
 * (MTE_DEVERR_*; 0 = none) and clears them; no synchronisation -- the host polls once per training step and raises.
 * (No reference counterpart: the reference's GroupNorm is one cuDNN call, layers01.py:32.) */
#define MTE_DEVERR_GN_CLUSTER_FWD 1
#define MTE_DEVERR_GN_CLUSTER_BWD 2
int mte_device_error_init(void);
int mte_device_error_poll(void);
/* weight gradient of the same conv into dw_stage = `stage_parts` x [N][KH*KW][Cin_p] fp32 (overwritten).  The reduction over
 * pixels is split over at most stage_parts workgroup groups; each one stores its PARTIAL gradient in its own part (plain stores;
 * *parts_out = parts written, mte_unpack_conv_wgrad adds them in part order: round 4 -- no floating-point atomics on the conv weight
 * gradient's path, the result does not depend on the order in which workgroups finish).  mte_unpack_conv_wgrad with parts > 32 uses
 * up to 32 further slabs BEHIND the parts as scratch: a stage of more than 32 parts is allocated with parts + 32 slabs. */
int mte_conv2d_wgrad(const void* x, long ldx, const void* dy, long ldy, float* dw_stage, int stage_parts, int* parts_out,
                     int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype, mte_stream_t stream);
/* 1 when mte_conv2d_wgrad takes this 3x3 bf16 layer with its nine-tap kernel (conv_wgrad9.hip: N % 128 == 0, Cin_p % 64 == 0, W % 32 == 0 or W % 16 == 0
 * and H even).  Round 5: measured faster than the LDS-patch weight gradient on every 128-output layer both take (128 -> 128 @96x320: 100.5 -> 84.9 us,
 * 192 -> 128: 147 -> 115), so the host asks this first. */
int mte_conv2d_wgrad_nine_tap(int H, int W, int Cin_p, int N, int KH, int KW, int dtype);
/* OIHW fp32 master weights -> forward pack [Cout][taps][Cin_p] and (optional) dgrad pack [Cin_p][taps rot180][Cout_p] */
int mte_pack_conv_weights(const float* w_oihw, void* wfwd, void* wbwd, int Cout, int Cin, int KH, int KW,
                          int Cin_p, int Cout_p, int dtype, mte_stream_t stream);
/* Every kernel-ready weight copy of `njobs` conv layers in ceil(njobs / 48) launches (the ~100 layers of the network after an
 * optimizer step: ~400 launch-bound per-layer launches otherwise).  `jobs` is a HOST array; per job, from the OIHW fp32 master `w`
 * [Cout][Cin][taps]: wf = forward pack [Cout][taps][Cin_p]; wb (nullable) = data-gradient pack [Cin_p][taps rot180][Cout]; pf / pb
 * (nullable, bf16 only) = the fragment-block packs of mte_conv2d_patch_fwd for wf / wb (mte_conv2d_patch_pack_elems(Cin_p, Cout, ..)
 * / (Cout, Cin_p, ..) elements).  end_* are scratch the library fills.  Same values as mte_pack_conv_weights + _patch_repack. */
typedef struct { const float* w; void* wf; void* wb; void* pf; void* pb; int Cout, Cin, taps, Cin_p; int end_f, end_b, end_pf, end_pb; } mte_pack_job;
int mte_pack_conv_weights_multi(const void* jobs, int njobs, int dtype, mte_stream_t stream);
/* dgrad pack derived from an existing forward pack */
int mte_pack_conv_weights_bwd(const void* wfwd, void* wbwd, int Cout, int KH, int KW, int Cin_p, int dtype, mte_stream_t stream);
/* sum of the `parts` partial stages -> OIHW fp32 gradient (drops channel padding).  The stage is scratch: with more than 32 parts
 * (LDS-patch weight gradient: one slab per workgroup) a parallel first level adds parts 1.. into part 0 before the transpose. */
int mte_unpack_conv_wgrad(float* dw_stage, int parts, float* dw_oihw, int Cout, int Cin, int KH, int KW, int Cin_p, mte_stream_t stream);
/* out[N] = column sums of y[M][N] (conv bias gradient) */
int mte_colsum(const void* y, long ld, long M, int N, float* out, int dtype, mte_stream_t stream);

/* ---- LDS-patch convolution for the high-resolution, few-channel layers (bf16, C_out <= 64, W % 32 == 0, k in {1,3,5,7}):
 * same math as mte_conv2d_igemm / mte_conv2d_wgrad, different tiling (the 8x32-pixel tile's input patch is staged once
 * per 32-channel slice and reused by all k*k taps).  *_supported and *_pack_elems are queries (they return a value). */
int mte_conv2d_patch_supported(int W, int Cin_p, int N, int KH, int KW, int dtype);
int mte_conv2d_patch_wgrad_supported(int W, int Cin_p, int N, int KH, int KW, int dtype);   /* + 3x3 layers with 65..128 output channels (weight gradient only) */
long mte_conv2d_patch_pack_elems(int Cin_p, int N, int KH, int KW);
int mte_conv2d_patch_repack(const void* wgeneric, void* wpatch, int Cin_p, int N, int KH, int KW, mte_stream_t stream);
int mte_conv2d_patch_fwd(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy,
                         int B, int H, int W, int Cin_p, int N, int KH, int KW, int accumulate, mte_stream_t stream);
int mte_conv2d_patch_wgrad(const void* x, long ldx, const void* dy, long lddy, float* dw_stage, int stage_parts, int* parts_out,
                           int B, int H, int W, int Cin_p, int N, int KH, int KW, mte_stream_t stream);
/* round 5 -- mte_conv2d_patch_fwd that also leaves the GroupNorm(16) statistics of what it stores (reference layers01.py:35-38: Conv2d -> GroupNorm(16) -> ELU;
 * the reference's GroupNorm is a cuDNN call that reads the conv output again): one record of 32 floats (sum, sum of squares per group, fp32 over the tile's
 * pixels) per output tile, written in the store loop from the bf16 values being stored -- with `accumulate` from the sums.  rec: at least
 * mte_conv2d_patch_fwd_gn_elems(B, H, W) floats; *tiles_per_sample_out records per sample were written (tile height depends on the kernel form).
 * mte_gn_stats_from_records (below) adds them in tile order into the buffer mte_gn_stats would have filled: the stand-alone statistics pass over y
 * (252 MB per full-resolution layer at B = 8) is not run.  N % 16 == 0, else MTE_ERR_UNSUPPORTED.  Bit-reproducible (no atomics). */
long mte_conv2d_patch_fwd_gn_elems(int B, int H, int W);
int mte_conv2d_patch_fwd_gn(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N, int KH, int KW,
                            int accumulate, float* rec, long rec_elems, int* tiles_per_sample_out, mte_stream_t stream);

/* ---- stem convolution: exactly 8 input channels (the zero-padded rgb image), C_out <= 32, k in {3,5,7}, W % 32 == 0, bf16
 * (encoder.pre_calc = Conv2D(3, 32, 5, 1): networks/depth/PackNetSAN01.py:27, layers01.py:29-31).  With 8 channels a pixel is one
 * 16-byte chunk, so the reduction runs over (tap, channel) and an MFMA covers two taps: a quarter of the MFMA work the 32-channel
 * slices of the LDS-patch kernels spend on it.  wf = the GENERIC forward pack [N][taps][8] of mte_pack_conv_weights (no special pack);
 * _wgrad fills dw_stage like mte_conv2d_patch_wgrad (one slab per workgroup, *parts_out slabs, summed by mte_unpack_conv_wgrad).
 * The image needs no data gradient. */
int mte_conv2d_stem_supported(int W, int Cin_p, int N, int KH, int KW, int dtype);
int mte_conv2d_stem_fwd(const void* x, long ldx, const void* wf, const float* bias, void* y, long ldy,
                        int B, int H, int W, int N, int KH, int KW, mte_stream_t stream);
int mte_conv2d_stem_wgrad(const void* x, long ldx, const void* dy, long lddy, float* dw_stage, int stage_parts, int* parts_out,
                          int B, int H, int W, int N, int KH, int KW, mte_stream_t stream);

/* ---- GroupNorm(16, C) + ELU, optionally over y1 + scale2[b,c]*y2 (residual tail with Dropout2d)
 *      (layers01.py:32-38 Conv2D; layers01.py:62-73 ResidualConv)
 * A statistics buffer is mte_gn_stats_elems(B) doubles: [B][16][2] (sum, sum of squares) -- what mte_gn_elu_fwd / _bwd read --
 * followed by the statistics pass's workspace (one arrival ticket per sample, which must be zero on entry -- cleared here unless
 * MTE_OPT_GN_PREZEROED -- and one record per workgroup).  The pass uses no floating-point atomics: per-thread sums, fixed wave
 * butterflies, waves in order, and the LAST workgroup of a sample adds the workgroup records in slot order, so the forward pass is
 * bit-reproducible run to run and under HIP-graph replay (round 3; the fp32 LDS / fp64 global atomics it replaces made two
 * forward passes of one frame differ by 1-2 % in inverse depth after ~60 bf16 layers). */
long mte_gn_stats_elems(int B);
int mte_gn_stats(const void* y1, long ld1, const void* y2, long ld2, const float* scale2, double* stats,
                 int B, int HW, int C, int dtype, mte_stream_t stream);
/* round 5: the same sums from the per-tile records a convolution left in its store loop (mte_conv2d_patch_fwd_gn): stats[b][group][2] = the
 * tiles_per_sample records of sample b added in tile order (fp64).  Only the first 32 B doubles of `stats` are written. */
int mte_gn_stats_from_records(const float* rec, int tiles_per_sample, double* stats, int B, mte_stream_t stream);
/* mte_gn_elu_fwd: stats_ready = 1: `stats` holds the sums of mte_gn_stats.  stats_ready = 0 (allowed where
 * mte_gn_fwd_is_single_pass(HW, C, y2 != NULL, dtype) returns 1: a (sample, group) slab fits one workgroup's registers): the
 * kernel loads each slab ONCE, computes its statistics on chip, normalises and stores `stats` in the same format as an
 * OUTPUT for the backward pass -- the forward is then one read + one write, and mte_gn_stats is not called.
 * mte_gn_elu_bwd takes the matching one-pass route by itself where the slab of (y, dz) fits (low-resolution layers).
 * (development knob 13 of the -DMTE_DEV build: 0 = streaming two-pass kernels everywhere.) */
int mte_gn_fwd_is_single_pass(int HW, int C, int has_y2, int dtype);
/* Round 4: the same route for slabs too large for one workgroup (512 channels at 24x80, 256 at 48x160, 128 at 96x320): a CLUSTER of
 * 2-8 workgroups holds the slab in registers and exchanges its partial sums through the record area of `stats` (fixed order:
 * bit-reproducible; the whole buffer must be zero at entry -- MTE_OPT_GN_PREZEROED callers guarantee it, otherwise the library clears it).
 * The cluster's size depends on the batch, so callers that know B ask mte_gn_fwd_is_single_pass_b; stats_ready = 0 is allowed wherever it
 * returns 1.  Replaces the same reference ops (nn.GroupNorm(16, C) + nn.ELU of Conv2D, layers01.py:32-38; residual tail :62-73).
 * (development knob 25 of the -DMTE_DEV build: 0 = no cluster kernels.) */
int mte_gn_fwd_is_single_pass_b(int B, int HW, int C, int has_y2, int dtype);
int mte_gn_elu_fwd(const void* y1, long ld1, const void* y2, long ld2, const float* scale2, double* stats, int stats_ready,
                   const float* gamma, const float* beta, void* z, long ldz,
                   int B, int HW, int C, float eps, int dtype, mte_stream_t stream);
/* mte_gn_elu_bwd: y2 = NULL with scale2 and d2 given (round 5, the backward of mte_gn_tail_fwd's outer norm): ONE input tensor, but the
 * gradient leaves twice -- d1 = dv and d2 = scale2[b,c] * dv -- and dbias, if asked, is the column sum of d2.  d2 without y2 AND without scale2 is
 * MTE_ERR_ARG (there is no factor to form it with). */
int mte_gn_elu_bwd(const void* dz, long lddz, const void* y1, long ld1, const void* y2, long ld2, const float* scale2,
                   const double* stats, const float* gamma, const float* beta, float* red,
                   void* d1, long ldd1, void* d2, long ldd2, float* dgamma, float* dbeta, float* dbias,
                   int B, int HW, int C, float eps, int dtype, mte_stream_t stream);
/* Round 5 -- the tail of a residual block in two launches: ResidualConv.forward's `self.activ(self.normalize(x_out + shortcut))` with
 * x_out = conv2's GroupNorm + ELU applied on the fly (layers01.py:35-38 and :69-73).
 *   y1 = conv2's CONVOLUTION output (+ bias), stats1 = its sums (mte_gn_stats), gamma1 / beta1 = conv2.normalize;
 *   y2 = the 1x1 shortcut's output, scale2 = Dropout2d's keep / (1 - p) per (sample, channel) or NULL;
 *   t  = ELU(GN(y1)) + scale2 * y2 -- an OUTPUT in the activation type, the tensor both backward norms need -- with its statistics in stats_t
 *        (a statistics buffer, tickets zero at entry; bit-reproducible like mte_gn_stats);  z = ELU(GN_t(t)) with gamma_t / beta_t = the block's normalize.
 * Against the four launches it replaces (mte_gn_elu_fwd of conv2, mte_gn_stats and mte_gn_elu_fwd over two tensors): 6 tensor passes instead of 8, and the
 * backward of the outer norm reads one tensor instead of two in both of its passes. */
int mte_gn_tail_fwd(const void* y1, long ld1, const double* stats1, const float* gamma1, const float* beta1,
                    const void* y2, long ld2, const float* scale2, void* t, long ldt, double* stats_t,
                    const float* gamma_t, const float* beta_t, void* z, long ldz,
                    int B, int HW, int C, float eps, int dtype, mte_stream_t stream);

/* ---- 3-D packing / unpacking stencils: packing + nn.Conv3d(1,4,3,pad 1) (+ view / PixelShuffle)
 *      (layers01.py:127-149, 214-248 PackLayerConv3d; 251-287 UnpackLayerConv3d).  H,W,C describe x. */
int mte_pack3d_fwd(const void* x, long ldx, const float* w3, const float* b3, void* out, long ldo,
                   int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_pack3d_bwd_data(const void* dout, long ldo, const float* w3, void* dx, long lddx,
                        int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_pack3d_bwd_weight(const void* x, long ldx, const void* dout, long ldo, float* dwb,
                          int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_unpack3d_fwd(const void* x, long ldx, const float* w3, const float* b3, void* out, long ldo,
                     int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_unpack3d_bwd_data(const void* dout, long ldo, const float* w3, void* dx, long lddx,
                          int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_unpack3d_bwd_weight(const void* x, long ldx, const void* dout, long ldo, float* dwb,
                            int B, int H, int W, int C, int dtype, mte_stream_t stream);

/* ---- conv3d folded into the pack convolution (see csrc/pack_fold.hip): exact away from the border; kernels.PackFoldedConvFn
 *      recomputes the k/2-pixel border bands with the unfolded kernels.  (layers01.py:241-247) */
int mte_fold_pack_weights(const float* W, const float* K3, const float* b, const float* b3, float* Wf, float* bf,
                          int Co, int D, int k, mte_stream_t stream);
int mte_unfold_pack_wgrad(const float* dWf, const float* dbf, const float* W, const float* K3, const float* b3,
                          float* dW, float* dk3b, int Co, int D, int k, int accumulate, mte_stream_t stream);
int mte_pixel_shuffle(const void* src, long lds_, void* dst, long ldd, int B, int H, int W, int C, int dir, int dtype, mte_stream_t stream);
int mte_copy_rect(const void* src, long lds_, int Hs, int Ws, int sy, int sx, void* dst, long ldd, int Hd, int Wd, int dy, int dx,
                  int B, int h, int w, int C, int mode, int dtype, mte_stream_t stream);
/* the same for up to 8 rectangles in ONE launch; ops = HOST array of n records (device pointers inside) */
typedef struct { const void* src; long lds_; int Hs, Ws, sy, sx; void* dst; long ldd; int Hd, Wd, dy, dx, h, w, mode; } mte_rect_op;
int mte_copy_rects(const void* ops, int n, int B, int C, int dtype, mte_stream_t stream);

/* ---- InvDepth head: sigmoid(conv3x3(x) + b) / min_depth  (layers01.py:99-123) */
int mte_invdepth_fwd(const void* x, long ldx, const float* w, const float* bias, float* out,
                     int B, int H, int W, int C, float min_depth, int dtype, mte_stream_t stream);
/* backward, two independent halves: _data writes dlogit [B,H,W] (fp32 scratch, also the input of _weight) and dx;
 * _weight writes dwb [C*9 + 1] = (dW OIHW, db): every workgroup stores one record of its sums into `records`
 * (mte_invdepth_bwd_weight_workspace_elems(C) floats of scratch) and a second small launch adds the records in a fixed order
 * (no atomics: the C*9+1 contended adds per workgroup cost as much as the stream itself on the 256-channel head) */
int mte_invdepth_bwd_data(const float* w, const float* inv_out, const float* dout, float* dlogit,
                          void* dx, long lddx, int B, int H, int W, int C, float min_depth, int dtype, mte_stream_t stream);
long mte_invdepth_bwd_weight_workspace_elems(int C);
int mte_invdepth_bwd_weight(const void* x, long ldx, const float* dlogit, float* dwb, float* records,
                            int B, int H, int W, int C, int dtype, mte_stream_t stream);

/* The inverse-depth channel of the decoder's iconv inputs as a rank-1 term (round 4).  iconv3 / iconv2 / iconv1 see
 * torch.cat((unpack, skip, nearest_up2(inv_depth)), 1) (reference PackNetSAN01.py:118-143): conv(cat(x, u)) = conv(x) + conv_1(u).  The one-channel part is a
 * 3x3 stencil of the low-resolution map; the GEMM kernels run on the other C-1 = 64 / 96 / 192 channels and ACCUMULATE onto it.  w: channel C-1 of the OIHW
 * weight (element (n, tap) at w[n * w_stride + tap]).  _fwd overwrites y [B,2h,2w,N]; _bwd_data: dinv [B,h,w] (+)=; _bwd_weight below. */
int mte_rank1_conv_fwd(const float* inv, const float* w, long w_stride, void* y, long ldy, int B, int h, int wl, int N, int dtype, mte_stream_t stream);
int mte_rank1_conv_bwd_data(const void* dy, long lddy, const float* w, long w_stride, float* dinv, int B, int h, int wl, int N, int accumulate, int dtype,
                            mte_stream_t stream);
int mte_upsample2_f32(const float* inv, float* out, int B, int h, int wl, mte_stream_t stream);
/* _fwd fused into the LDS-patch 3x3 forward (bf16, N <= 64): y = conv_3(x, wpatch) + bias + conv_1(nearest_up2(inv), w1) in ONE launch -- the store loop of a
 * tile adds the term from two small LDS tables (the map under the tile + halo, the 9 x N weights), so y is neither written first nor read back.  inv [B,H/2,W/2];
 * w1 / w1_stride as w / w_stride above.  _ok(...) = 1 when this form exists for the shape (otherwise: mte_rank1_conv_fwd, then the conv with accumulate = 1). */
int mte_conv2d_patch_fwd_rank1_ok(const float* bias, long ldx, int B, int H, int W, int Cin_p, int N);
int mte_conv2d_patch_fwd_rank1(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N,
                               const float* inv, const float* w1, long w1_stride, mte_stream_t stream);
/* y = conv_3x3(x, wpatch) + conv_1x1(x2, wpatch2) + bias in one launch of the LDS-patch kernel (bf16, N <= 64, W % 32 == 0): the second source's C2 channels are
 * further K-steps of every tile at the centre tap; wpatch2 = fragment-block pack of the 1x1 weights for the same N (mte_conv2d_patch_pack_elems(C2, N, 1, 1)).
 * The data gradient of a residual block's input (reference layers01.py:55-73: conv1 and the 1x1 shortcut read the same x): dx = conv3x3^T(dy1) + conv1x1^T(dy3)
 * without a stand-alone 1x1 launch and without an accumulating pass over dx. */
int mte_conv2d_patch_fwd_plus1x1(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N,
                                 const void* x2, long ldx2, const void* wpatch2, int C2, mte_stream_t stream);
/* its gradient with respect to the weight column: dw: element (n, tap) at dw[n * dw_stride + tap] (overwritten); records: mte_rank1_conv_bwd_records_elems(N)
 * floats of scratch.  Every record of dy is read once; no floating-point atomics (fixed-order sums: bit-reproducible). */
long mte_rank1_conv_bwd_records_elems(int N);
int mte_rank1_conv_bwd_weight(const void* dy, long lddy, const float* inv, float* dw, long dw_stride, float* records,
                              int B, int h, int wl, int N, int dtype, mte_stream_t stream);
/* ---- layout / wiring helpers (networks/depth/PackNetSAN01.py:92-143 cat + Upsample; models/model_utils.py:98-117 flip) */
int mte_nchw_to_nhwc(const float* src, void* dst, long ldd, int B, int C, int H, int W, int Cp, int flip_w, int dtype, mte_stream_t stream);
int mte_upsample_inv_fwd(const float* inv, void* dst, long ldd, int B, int h, int w, int dtype, mte_stream_t stream);
int mte_upsample_inv_bwd(const void* dsrc, long lds_, float* dinv, int B, int h, int w, int accumulate, int dtype, mte_stream_t stream);
int mte_copy_channels(const void* src, long lds_, void* dst, long ldd, long npix, int C, int dtype, mte_stream_t stream);
/* a [dW; db] record (conv3d weight gradients, the heads' weight gradients) into its two places of a flat gradient buffer in one launch:
 * dst0[0..n0) = src[0..n0), dst1[0..n1) = src[n0..n0+n1) */
int mte_split_record(const float* src, float* dst0, int n0, float* dst1, int n1, mte_stream_t stream);
/* out = a + b over NHWC channel-slice views: the summed gradient of an activation with two consumers (what autograd's
 * implicit accumulation does in the reference), one 16-byte-vectorised pass whatever the strides */
int mte_add_channels(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long npix, int C, int dtype, mte_stream_t stream);

/* ---- depth-edge loss: inv2depth + GradLayer + GradLoss('cross_entropy')
 *      (utils/depth.py:104-121; losses/grad_loss.py:20-31,65-95,122-219) */
/* All scales of SemiSupEdgeModel.compute_edge_loss_with_all_scales (models/SemiSupEdgeModel.py:164-198) in ONE forward and ONE
 * backward launch, optionally with the silog loss of scale 0 (models/SemiSupEdgeModel.py:144; losses/supervised_loss.py:57-69,
 * 155-216) fused in -- it reads the same full-resolution inverse depth.  Workgroup = 64x32-pixel tile of one (scale, sample); the
 * last workgroup to arrive adds the per-workgroup partial sums in a fixed order and computes alpha, the loss scalars and the
 * backward coefficients on the device (no reduce / finalize launches, no host sync, deterministic sums).
 *   scales : HOST array of nscales (<= 4) records; every map is fp32 [B,H,W]; normal / mask / gmap nullable; dpred = backward output
 *   work   : mte_edge_loss_work_elems() doubles (content on entry ignored; zero on entry under MTE_OPT_LOSS_PREZEROED)
 *   losses [nscales] <- weight * balanced BCE per scale;  coef [nscales][2B+1] <- backward coefficients
 *   gt_depth (nullable; metric depth, 0 = invalid, at the size of scale 0): fused silog -> silog_loss[1], silog_aux[2]
 *   backward: dpred_s <- gout[s] * d loss_s / d pred_s  (+ silog_gout[0] * d silog / d pred_0); gout / silog_gout: device, nullable = 1 */
typedef struct { const float* pred; const float* edge; const float* normal; const float* mask; float* gmap; float* dpred; int H; int W; } mte_edge_scale;
long mte_edge_loss_work_elems(const void* scales, int nscales, int B);
int mte_edge_loss_multi_fwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            float weight, float pos_to_neg, const float* gt_depth, double* work, float* losses, float* coef,
                            float* silog_loss, float* silog_aux, mte_stream_t stream);
int mte_edge_loss_multi_bwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            const float* coef, const float* gout, const float* gt_depth, const float* silog_aux, const float* silog_gout,
                            mte_stream_t stream);
/* single-scale entry points (GradLoss / GradLayer called on their own): the same kernels with one scale */
long mte_edge_loss_sums_elems(int B, int H, int W);   /* doubles `sums` must hold: B*6 + 4 results, the arrival ticket, the per-workgroup partial sums */
int mte_edge_loss_fwd(const float* pred, const float* edge, const float* normal, const float* mask, double* sums, float* gmap,
                      int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, mte_stream_t stream);
int mte_edge_loss_finalize(const double* sums, int B, long numel, float weight, float pos_to_neg, int has_mask,
                           float out_scale, float* loss_acc, float* loss_this, float* coef, mte_stream_t stream);
int mte_edge_loss_bwd(const float* pred, const float* edge, const float* normal, const float* mask, const float* coef, const float* gout,
                      float* dpred, int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, mte_stream_t stream);

/* F.interpolate(pred, size = label size, mode = 'bilinear') of GradLoss.forward (grad_loss.py:127; identity on the multi-scale
 * training path, where every scale is compared at its own resolution) and its adjoint; fp32 [B,h,w] -> [B,H,W] */
int mte_resize_bilinear_fwd(const float* x, float* y, int B, int h, int w, int H, int W, mte_stream_t stream);
int mte_resize_bilinear_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, mte_stream_t stream);

/* ---- silog supervised loss: depth2inv + sparse mask + SilogLoss (utils/depth.py:124-144; losses/supervised_loss.py:57-69,155-216) */
int mte_silog_fwd(const float* inv, const float* depth, long n, double* sums, float out_scale, float* loss_acc, float* loss_this, float* aux, mte_stream_t stream);
int mte_silog_bwd(const float* inv, const float* depth, const float* aux, const float* gout, float* dinv, long n, int accumulate, mte_stream_t stream);

/* ---- optimizer: torch.optim.Adam step over a flat fp32 buffer (models/model_wrapper.py:142-180; trainers/common_trainer.py:125) */
int mte_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  int step, float gscale, mte_stream_t stream);
/* the same update with hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read from DEVICE memory: no per-step host scalar in the launch,
 * so the training step can be replayed from a HIP graph (utils/graph.py::GraphedTrainStep) */
int mte_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps,
                      float gscale, mte_stream_t stream);

/* ---- validation depth metrics: fp32 [B,1,H,W] maps, no host synchronisation
 * mte_post_process_inv_depth = post_process_inv_depth (utils/depth.py:230-256); method 0 'mean', 1 'max', 2 'min'
 *   (fuse_inv_depth, utils/depth.py:202-227).  `inv_depth_flipped` is the network output on the mirrored image.
 * mte_depth_metrics = compute_depth_metrics (utils/depth.py:259-325): `pred` [B,1,h,w] is brought to the ground-truth
 *   resolution by scale_mode 0 'resize' (bilinear, align_corners=True) or 1 'top-center' (utils/depth.py:328-361);
 *   valid = min_depth < gt < max_depth (and the garg crop window); with use_gt_scale the prediction is multiplied by
 *   median(gt)/median(pred) over the valid pixels (torch.median = element of rank (n-1)/2); out7 receives
 *   (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) averaged over the B images (images without a valid pixel add 0).
 *   `workspace` needs mte_depth_metrics_workspace_bytes(B) bytes, 8-byte aligned; its content on entry is ignored. */
long mte_depth_metrics_workspace_bytes(int B);
int mte_depth_metrics(const float* gt, const float* pred, int B, int H, int W, int h, int w, int scale_mode, int garg_crop,
                      float min_depth, float max_depth, int use_gt_scale, void* workspace, long workspace_bytes,
                      float* out7, mte_stream_t stream);
int mte_post_process_inv_depth(const float* inv_depth, const float* inv_depth_flipped, float* out, int B, int H, int W,
                               int method, mte_stream_t stream);

/* ---- depth-edge annotation post-processing: fp32 [B,H,W] maps
 * mte_dee_sobel_nms: prob = pred*scale (the '/2' of infer_edge_estimation.py:192); 5x5 Sobel as cv2.Sobel(CV_64F, ksize=5);
 *   normals_u8 = uint8(((atan2(-sy,sx)*180/pi+180)/360)*255) (infer_edge_estimation.py:194-200) and/or
 *   nms = non_max_suppression(prob) (utils/tools.py:9-46; zero one-pixel frame).  Either output may be NULL.
 * hysteresis(img, t_low, t_high) (utils/tools.py:49-92) in three steps so that the host decides when to stop sweeping:
 *   mte_hysteresis_begin      classify into `state` (B*H*W bytes), reset `info` (B*4 ints)
 *   mte_hysteresis_propagate  run `sweeps` propagation sweeps; `flags` (sweeps+1 ints) is reset by the call and
 *                             flags[sweeps] != 0 afterwards means the last sweep still changed pixels: call again
 *   mte_hysteresis_finish     out = img * state/max(state), including the reference's treatment of the frame */
int mte_dee_sobel_nms(const float* pred, float scale, unsigned char* normals_u8, float* nms, int B, int H, int W, mte_stream_t stream);
int mte_hysteresis_begin(const float* img, unsigned char* state, int* info, int B, int H, int W, double t_low, double t_high,
                         mte_stream_t stream);
int mte_hysteresis_propagate(unsigned char* state, int* flags, int sweeps, int B, int H, int W, mte_stream_t stream);
int mte_hysteresis_finish(const float* img, const unsigned char* state, const int* info, float* out, int B, int H, int W,
                          mte_stream_t stream);

/* ---- chamfer edge metrics (edge half): chamfer_distance of utils/edge.py:19-64 (= edge.py:29-71), mask=None.
 * im_pred / im_gt: float [B,H,W] edge images on the 0..255 scale (a pixel is an edge iff v/255 > 0.5).  out[b] = (c_dist,
 * percentage) as doubles: mean exact Euclidean distance from the predicted edge pixels to the nearest ground-truth edge
 * pixel and the share of them closer than edge_to_edge_thresh (NaN when nothing is predicted, +inf distances when the
 * ground truth has no edge pixel -- scipy's result for that case is not reproduced).  dist_map (nullable) receives the
 * distance transform, cond_map (nullable) the reference's -1 / 0 / 1 map.  workspace: mte_chamfer_workspace_bytes(B,H,W). */
long mte_chamfer_workspace_bytes(int B, int H, int W);
int mte_chamfer_distance(const float* im_pred, const float* im_gt, int B, int H, int W, double edge_to_edge_thresh,
                         void* workspace, long workspace_bytes, double* out, float* dist_map, float* cond_map, mte_stream_t stream);

/* ---- Canny step of the validation edge metrics (models/model_wrapper.py:376-400).  PARITY UNPINNED: restates OpenCV's
 * published cv2.Canny (apertureSize 3, L1 gradient) -- see oracle/canny_oracle.py.
 * mte_canny_begin: vis = uint8(depth * (255 / max(depth))) per image (vis_u8 nullable, B*H*W bytes), 3x3 Sobel, non-maximum
 *   suppression, and classification for n_pairs (<= 4) threshold pairs `thresholds[2*p], thresholds[2*p+1]` (HOST ints)
 *   into state[n_pairs][B][H][W] bytes (0 none, 1 candidate, 2 edge).  max_ws: B unsigned ints of scratch.
 * mte_canny_propagate: hysteresis sweeps over the `maps` = n_pairs*B state maps; same flags protocol as mte_hysteresis_propagate.
 * mte_canny_finish: edges[maps][H][W] floats, 255 on edges, 0 elsewhere (the scale mte_chamfer_distance expects). */
int mte_canny_begin(const float* depth, int B, int H, int W, int n_pairs, const int* thresholds, unsigned* max_ws,
                    unsigned char* vis_u8, unsigned char* state, mte_stream_t stream);
 /* mte_resize_linear: cv2.resize(float32 [B,h,w] -> [B,H,W], INTER_LINEAR) as used in front of the Canny step
 * (models/model_wrapper.py:386-387); restated from OpenCV's resize.cpp, PARITY UNPINNED like the rest of this block. */
int mte_resize_linear(const float* src, int B, int h, int w, float* dst, int H, int W, mte_stream_t stream);
int mte_canny_propagate(unsigned char* state, int* flags, int sweeps, int maps, int H, int W, mte_stream_t stream);
int mte_canny_finish(const unsigned char* state, float* edges, int maps, int H, int W, mte_stream_t stream);

/* ---- sparse auxiliary (SAN) branch.  PARITY UNPINNED: dense-equivalent of the
 * MinkowskiEngine operators the reference uses (networks/layers/minkowski_encoder.py:11-132, minkowski.py:33-79); see
 * oracle/san_oracle.py.  Features are zero-filled NHWC activations (bf16 / fp32), the active set is a byte mask [B,H,W].
 * mte_sparsify_depth: mask = depth > 0, feat channel 0 = depth on the mask, channels 1..7 = 0 (feat has >= 8 channels/pixel)
 * mte_sparse_maxpool3s2: MinkowskiMaxPooling(3, stride 2): mask_out = any of the 2x2 block, value = max over the active
 *   cells of the centred 3x3 window; H and W even
 * mte_sparse_bn_relu: out = mask ? relu(batchnorm_eval(a [+ b] [+ c])) : 0   (b, c nullable)
 * mte_san_fuse: out = skip * w[0] + sparse + bias[0]   (networks/depth/PackNetSAN01.py:254-258) */
int mte_sparsify_depth(const float* depth, void* feat, long ldf, unsigned char* mask, int B, int H, int W, int dtype, mte_stream_t stream);
/* Round 3: the sparse convolutions as gather-GEMM-scatter over the ACTIVE sites.  mte_sparse_site_list: sites[0 .. *count) = raster-ordered
 * pixel indices of the active set (three small launches, no atomics; `count` stays in device memory).  mte_conv2d_igemm_sparse: the
 * implicit GEMM of mte_conv2d_igemm with GEMM row m = pixel sites[m]: the k x k taps of a site are gathered from the zero-filled dense
 * map (inactive neighbours contribute zeros = MinkowskiConvolution's sum over active neighbours), the result is scattered to the same
 * sites and nothing else is written; tiles beyond *count return at once, so the work is proportional to the active count (LiDAR: ~5 % at
 * the first level) without the count ever visiting the host.  Also the data gradient (backward pack, x := dy). */
long mte_sparse_site_list_workspace_elems(long npix);
int mte_sparse_site_list(const unsigned char* mask, long npix, int* sites, int* count, int* ws, mte_stream_t stream);
int mte_conv2d_igemm_sparse(const void* x, long ldx, const void* wpack, const float* bias, void* y, long ldy,
                            int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype,
                            const int* sites, const int* count, int accumulate, mte_stream_t stream);
int mte_sparse_maxpool3s2(const void* in, long ldi, const unsigned char* mask_in, void* out, long ldo, unsigned char* mask_out,
                          int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_sparse_bn_relu(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const unsigned char* mask,
                       const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       void* out, long ldo, long npix, int C, int dtype, mte_stream_t stream);
int mte_san_fuse(const void* skip, long ld_skip, const void* sparse, long ld_sparse, const float* w, const float* bias,
                 void* out, long ldo, long npix, int C, int dtype, mte_stream_t stream);
/* Training path of the branch (two-pass RGB / RGB+LiDAR step, networks/depth/PackNetSAN01.py:324-342; minkowski_encoder.py:27-84):
 * mte_sparse_bn_stats: MinkowskiBatchNorm in training mode = BatchNorm1d over the active points of the whole batch:
 *   sums[0..C) = sum x, sums[C..2C) = sum x^2 over the active pixels (x = a [+ b] [+ c]), sums[2C] = number of active pixels
 *   (fp64, zeroed here).  The caller forms mean / biased variance and runs mte_sparse_bn_relu with them.
 * mte_sparse_bn_relu_bwd: dy = dout * [out > 0];  dx = mask ? gamma * invstd * (dy - mean_active(dy) - xhat * mean_active(dy xhat)) : 0
 *   (the gradient of a, b and c alike; n_active: device pointer to the number of active pixels, i.e. mte_sparse_bn_stats' sums + 2C); sums[0..C) = sum dy (= dbeta), sums[C..2C) = sum dy * xhat (= dgamma); scratch zeroed here.
 * mte_sparse_maxpool3s2_bwd: a fine cell receives the gradient of every coarse cell whose window maximum it is (first maximum in
 *   row-major window order among the active cells).
 * mte_san_fuse_bwd: dskip = dout * w[0]; sums[0] = sum dout * skip (dw), sums[1] = sum dout (db); the sparse operand's gradient is dout.
 * mte_feat_l2: sum (nullable) = sum (a - b)^2;  db (nullable) = -2 (a - b) * inv_n * gscale[0]  -- the feature-matching loss between
 *   the RGB+LiDAR pass (a, detached) and the RGB pass (b) and its gradient (PackNetSAN01.py:340-342). */
int mte_sparse_bn_stats(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const unsigned char* mask,
                        double* sums, long npix, int C, int dtype, mte_stream_t stream);
int mte_sparse_bn_relu_bwd(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const void* out, long ldo,
                           const void* dout, long ldd, const unsigned char* mask, const float* gamma, const float* mean, const float* invstd,
                           const double* n_active, double* sums, void* dx, long ldx, long npix, int C, int dtype, mte_stream_t stream);
int mte_sparse_maxpool3s2_bwd(const void* in, long ldi, const unsigned char* mask_in, const void* dout, long ldd, void* din, long ldn,
                              int B, int H, int W, int C, int dtype, mte_stream_t stream);
int mte_san_fuse_bwd(const void* skip, long ld_skip, const void* dout, long ldd, const float* w, void* dskip, long ldk, double* sums,
                     long npix, int C, int dtype, mte_stream_t stream);
int mte_feat_l2(const void* a, long lda, const void* b, long ldb, double* sum, void* db, long ldg, const float* gscale, float inv_n,
                long npix, int C, int dtype, mte_stream_t stream);

/* ---- training-target preparation (data half)
 * mte_edge_target_from_u8:   dst = src / 255                      (datasets/augmentations.py:186-188,199-201)
 * mte_normal_target_from_u8: dst = (360 * (src/255) - 180) * pi/180   (datasets/gta_dataset.py:407-409,417-418), float64 inside
 * mte_resize_depth_preserve: resize_depth_preserve (datasets/augmentations.py:58-100) of float [B,h,w] sparse maps into
 *   [B,H,W]: valid = value > 0, target = (int(y*H/h), int(x*W/w)), last source pixel in raster order wins, zeros elsewhere.
 *   winner_ws: B*H*W ints of scratch. */
int mte_edge_target_from_u8(const unsigned char* src, float* dst, long n, mte_stream_t stream);
int mte_normal_target_from_u8(const unsigned char* src, float* dst, long n, mte_stream_t stream);
int mte_resize_depth_preserve(const float* src, int B, int h, int w, float* dst, int H, int W, int* winner_ws, mte_stream_t stream);

#ifdef MTE_DEV
/* Development knobs for same-box A/B measurements (tools/sweep.sh) and kernel-variant cross-checks (tests/).  NOT part of the
 * integration surface: exported only by libmte_hip_dev.so, the -DMTE_DEV build of the same sources (mindtheedge_amd/_build.py);
 * libmte_hip.so, the shipped library, has neither this entry point nor the ablation arms of the implicit-GEMM template.  key:
 *   0 igemm loader (1 buffer-descriptor LDS-DMA [default], 2 pointer LDS-DMA, 0 register staging)
 *   1 conv3d kernels (0 gather, 1 LDS-tiled, 2 + four-plane unpack data gradient [default]; 100/101 large/half-size tiles)
 *   2 / 3 GroupNorm launch geometry (min rows per thread / target workgroups)
 *   4 wgrad kernel (1 LDS-DMA ring [default], 0 register-staged)      6 igemm tiles (0 128x128, 1 + 256x128, 2 + 256x256, 3 + 192x96 [default])
 *   7 min tiles for the big igemm tiles (224)   8 wgrad 8/16-wave tiles (1)   9 wgrad workgroup target (512)
 *   11 patch conv: 0/1 tall 16x32 tiles, >= 100 = workgroup target of the patch wgrad (512)
 *   13 GroupNorm single-pass slab kernels (1 [default], 0 = streaming kernels only)
 *   14 GroupNorm second passes walk the samples in reverse order (Infinity-Cache reuse; 1 [default])
 *   15 implicit GEMM, 8-wave 256x128 tile: 0 [default] 4-slot LDS ring, 3-slot / two workgroups per CU for MTE_CONV_SOLO launches;
 *      1 = 6 slots, 3 = always 3 slots x 2 workgroups, 4 = always 4 slots
 *   19 implicit GEMM: two-workgroup 256x128 variant beside the
 *      weight-gradient stream for launches with at most this many K-steps and >= 512 tiles (72 [default], 0 = solo launches only)
 *   17 implicit GEMM main-loop ablation (tools/igemm_ablate.py): leave out 1 MFMAs | 2 in-loop LDS-DMA | 4 fragment reads; results are garbage */
int mte_debug_set(int key, int value);
#endif /* MTE_DEV */

#ifdef __cplusplus
}
#endif
#endif /* MTE_KERNELS_H */
