/*
 * fluidnet_ops.h -- argument slots of every FnetOp opcode (see fluidnet_hip.h for the record and
 * the entry points).  Documentation plus named slot constants for C/C++ hosts; the Python host
 * (fluidnet_amd/engine/plan.py) fills the same slots positionally.
 *
 * Conventions: "P" = P-tensor [B][H+2][W+2][ldc] of op.dtype (fluidnet_hip.h); a view is
 * (pointer, ldc, coff, C); every vectorised op needs ldc, coff and C multiples of 8 and the
 * pointer 16-byte aligned (checked on the host, FNET_E_BADARG otherwise).  pm = FNET_PAD_*;
 * act = FNET_ACT_*.  "(+=)" = the op accumulates into that buffer.  "|0" = pointer may be 0.
 * Npix = B*(H+2)*(W+2), Wp = W+2.  Reference citations are relative to the reference root.
 *
 * PACK_INPUT        <- torch.cat of the six input planes (ml/training/trainer.py:356) + the conv's own padding
 *   i: 0 B, 1 H, 2 W, 3 Cpad, 4 pm, 5..10 batch strides of the six planes (floats), 15 n (0: the six planes; n >= 1: p[0] is
 *      ONE contiguous [B,n,H,W] tensor with batch stride i[5] -- the nn.Module path, any in_channels <= Cpad)
 *   p: 0..5 planes fp32 (sample n at p + n*stride, H*W contiguous), 6 dst P[.,Cpad] (channels beyond the input's are zero)
 * CONV3X3           <- nn.Conv2d(k=3, padding=1) forward, and its data gradient on flipped weights
 *                      (ml/models/unet.py:55, 59, 92, 218; autograd's conv backward-data)
 *   i: 0 Npix, 1 Wp, 2 Cin, 3 ldc_in, 4 coff_in, 5 Cout, 6 ldc_out, 7 coff_out, 8 impl (FNET_CONV_*),
 *      9 mode (MFMA only): 0 = 3x3; 1 / 2 = nn.Conv2d(C, Cout, 3, stride=2, padding=1) (ml/models/unet.py:90-91) forward / data gradient
 *      as a 2x2-tap convolution over the space-to-depth view X'[Y][X][(py,px,c)] = Xpad[2Y+py][2X+px][c] of the full-resolution
 *      input: Npix, Wp describe the HALF-resolution padded space, 10 = h (half-resolution rows), 11 = C;
 *        mode 1: Cin = 4C, ldc_in / coff_in address the full-resolution source P, dst = half-resolution P (Cout channels);
 *        mode 2: src = half-resolution gradient P (Cin = the conv's Cout, ring zero), Cout = 4C, ldc_out / coff_out address the
 *                full-resolution destination P (C channels): every cell, ring included, is written once (a folded gradient);
 *        weights [4][rows][K] K-blocked, PREP_WEIGHTS kind 2
 *   p: 0 src P, 1 weights (same element type as src): impl REF [9][Cout][Cin]; impl MFMA K-blocked
 *      [ceil(Cin/16)][9][Cout][16] as written by PREP_WEIGHTS (MFMA: Cin % 8 == 0 suffices -- the last K stage may
 *      overrun 8 channels into the next pixel against the zero-padded weights), 2 bias fp32|0, 3 dst P (every pixel incl.
 *      ring is written; the ring is then overwritten by APPLY / PAD_RING)
 * WGRAD             <- autograd's conv backward-weight
 *   i: 0 Npix, 1 Wp, 2 Cin, 3 ldc_in, 4 coff_in, 5 Cout, 6 ldc_do, 7 coff_do, 8 impl, 9 Cin_w (row
 *      length of dWk), 10 B, 11 H, 12 capacity of p[3] in partials of 9*32*32 floats,
 *      13 mode (MFMA only): 1 = the stride-2 convolution in its 2x2 space-to-depth form (CONV3X3 mode 1): Npix, Wp, H describe the
 *      half-resolution dout, src is the full-resolution input of 14 = C channels, Cin = 4C, dWk = [4][Cout][Cin_w >= 4C]
 *   p: 0 src P (the conv's input), 1 dout P (gradient of the conv's output, ring = 0),
 *      2 dWk fp32 [9][Cout][Cin_w] (+=), 3 partial slab (MFMA: per-block partials + ordered reduce; REF: unused)
 * IN_STATS          <- InstanceNorm2d statistics (ml/models/unet.py:22-23, nn.InstanceNorm2d(affine=True))
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc, 5 coff, 6 floats available at p[2]; f: 0 eps
 *   p: 0 raw P, 1 stats float2[B][C] = (mean, 1/sqrt(var+eps)), 2 |0 scratch (bf16 plans: one
 *      coalesced pass into per-block partials + ordered finalize; fp32 plans / no scratch: exact
 *      two-pass kernel)
 * APPLY             <- norm affine -> activation -> Dropout2d [-> + residual], ring refill
 *                      (ml/models/unet.py:56-69, 86-89)
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc_raw, 5 coff_raw, 6 ldc_res, 7 coff_res, 8 ldc_dst, 9 coff_dst, 10 act,
 *      11 pm, 12 dropout site id, 13 dropout seed; f: 0 drop_p
 *   p: 0 raw P, 1 stats|0 (0 = no norm), 2 gamma|0, 3 beta|0, 4 residual P|0, 5 dst P
 * UPSAMPLE2X        <- nn.Upsample(scale_factor=2, bilinear align_corners=False | nearest)
 *                      (ml/models/unet.py:152-160)
 *   i: 0 B, 1 h, 2 w, 3 C, 4 ldc_s, 5 coff_s, 6 ldc_d, 7 coff_d, 8 mode (0 nearest, 1 bilinear), 9 pm
 *   p: 0 src P (h,w), 1 dst P (2h,2w)
 * SUBSAMPLE_BIAS    <- the stride-2 half of nn.Conv2d(stride=2) (ml/models/unet.py:92): picks
 *                      the even-centred pixels of a full-resolution CONV3X3 result and adds the bias
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc_s, 5 coff_s, 6 ldc_d, 7 coff_d, 8 pm; p: 0 src P (H,W), 1 bias|0, 2 dst P (H/2,W/2)
 * PAD_RING          <- F.pad(mode=...) in place
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc, 5 coff, 6 pm; p: 0 buf P
 * HEAD_FWD          <- 1x1 head conv + output activation (ml/models/unet.py:279, 294-303)
 *   i: 0 B, 1 H, 2 W, 3 C (real channels; need not be a multiple of 8: the tail up to the next 8 is zero padding),
 *      4 ldc, 5 coff, 6 Cout (<= 4), 7 FNET_OUT_*
 *   p: 0 src P, 1 Wh fp32 [Cout][C], 2 bh, 3 z fp32 [B,Cout,H,W] (pre-activation), 4 pred fp32 [B,Cout,H,W]
 * LOSS_REDUCE       <- PhysicsAwareLoss.forward sums (ml/training/physics_loss.py:210-278)
 *   i: 0 B, 1 H, 2 W, 3 target bstride, 4 in0 bstride, 5 emitter bstride, 6 collider bstride, 7 pm, 8 stencil,
 *      9 flags (bit0 divergence, bit1 gradient, bit2 emitter term enabled); f: 0 dx
 *   p: 0 pred [B,3,H,W], 1 target, 2 in0 (density_t plane), 3 emitter, 4 collider, 5 sums double[8] (zeroed by the caller)
 * LOSS_FINALIZE     <- weighted total + the reference's loss dict (physics_loss.py:262-278)
 *   i: 0 B, 1 H, 2 W, 3 flags, 4 world; f: 0 w_mse, 1 w_div, 2 w_grad, 3 w_emit, 4 rollout-step weight
 *   p: 0 sums, 1 |0 global fluid count double* (data parallel), 2 terms float[8], 3 |0 accum float[8] (+=)
 * LOSS_GRAD         <- d(total)/d(pred) (autograd of the above)
 *   i: 0..9 as LOSS_REDUCE, 10 world, 11 dnext bstride; f: 0 dx, 1 w_mse, 2 w_div, 3 w_grad, 4 w_emit, 5 step weight
 *   p: 0..4 as LOSS_REDUCE, 5 sums, 6 |0 global count, 7 |0 dnext [B,>=4,H,W] (gradient arriving from
 *      rollout step k+1's input, trainer.py:354-376), 8 |0 grad scale float*, 9 dpred out [B,3,H,W]
 * HEAD_BWD          <- adjoint of HEAD_FWD
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc, 5 coff, 6 ldc_d, 7 coff_d, 8 Cout, 9 FNET_OUT_*, 10 Cz (channels of dact written,
 *      0 = C rounded up to 8; channels C .. Cz-1 get zeros)
 *   p: 0 dpred, 1 z, 2 pred, 3 act P, 4 Wh, 5 dact out P (interior), 6 dWh (+=), 7 dbh (+=)
 * INACT_BWD_REDUCE  <- pass 1 of the norm/activation/dropout adjoint
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc_g1, 5 coff_g1, 6 ldc_g2, 7 coff_g2, 8 ldc_raw, 9 coff_raw, 10 ldc_G, 11 coff_G,
 *      12 fold1, 13 fold2 (1 = the gradient's ring is folded onto the interior per pm), 14 act, 15 pm,
 *      16 dropout site, 17 seed; f: 0 drop_p
 *   p: 0 g1 P, 1 g2 P|0, 2 raw P, 3 stats, 4 gamma, 5 beta, 6 G out P|0 (0 only with one source: pass 2 then reads g1
 *      itself), 7 S float2[B][C] (+=; zero first), 8 dgamma|0 (+=), 9 dbeta|0 (+=) (both or neither)
 * INACT_BWD_APPLY   <- pass 2: gradient w.r.t. the conv output
 *   i: 0 B, 1 H, 2 W, 3 C, 4 ldc_G, 5 coff_G, 6 ldc_raw, 7 coff_raw, 8 ldc_d, 9 coff_d, 10 act, 11 dropout site, 12 seed,
 *      13 fold (1 = p[0] is an unfolded gradient source: its ring is folded onto the interior here, rounded to the
 *      element type exactly as pass 1 would have stored it in G), 14 pm (used when fold = 1),
 *      15 weighted S (1: p[5] is NORM_COMBINE_BWD's output, BatchNorm2d / GroupNorm)
 *   f: 0 drop_p; p: 0 G (or the single source g1), 1 raw, 2 stats, 3 gamma, 4 beta, 5 S, 6 d_raw out P (ring = 0)
 * NORM_COMBINE      <- BatchNorm2d / GroupNorm (unet.py:20-28) on top of IN_STATS: pools the per-(n,c) (mean, rstd) over the batch
 *                      (kind 1, training; also updates the running statistics: momentum f[1], unbiased variance) or over the
 *                      channels of a group (kind 2), in place, every (n,c) receiving its pool's values; kind 3 = eval-mode
 *                      BatchNorm2d: statistics = the running ones (no IN_STATS needed)
 *                   i: 0 B, 1 C, 2 kind, 3 channels per group (kind 2), 4 H*W, 5 real C (0 = C; groups cover the real
 *                      channels of a zero-padded layer); f: 0 eps, 1 momentum
 *                   p: 0 stats float2[B][C], 1 running_mean|0, 2 running_var|0
 * NORM_COMBINE_BWD  i: 0 B, 1 C, 2 kind, 3 channels per group, 4 real C (0 = C); p: 0 S float2[B][C], 1 gamma, 2 Sw float2[B][C] out: the
 *                      gamma-weighted mean of S over the normalisation group (kind 3: zeros), read by INACT_BWD_APPLY with i[15] = 1
 * AFFINE_GRAD       i: 0 B, 1 C; p: 0 S float2[B][C], 1 dgamma (+=), 2 dbeta (+=)
 * AFFINE_GRAD_MULTI AFFINE_GRAD of up to 6 norm layers in one launch (the plan batches a backward pass's 26 layers into 5; slot 29 is the lane)
 *   i: 0 B, 1 n_units (<= 6), then per unit u at 2+4u: S offset (floats from p[0]), C, dgamma offset (-1: none), dbeta offset
 *      (floats from p[1])
 *   p: 0 S slab base, 1 flat gradient buffer base (+=)
 * FOLD_SUM          i: 0 B, 1 H, 2 W, 3 C, 4 ldc_g1, 5 coff_g1, 6 ldc_g2, 7 coff_g2, 8 ldc_d, 9 coff_d, 10 fold1, 11 fold2, 12 pm
 *                   p: 0 g1, 1 g2|0, 2 dst (interior = folded sum, ring = 0)
 * BIAS_GRAD         i: 0 B, 1 H, 2 W, 3 C, 4 ldc, 5 coff; p: 0 d P, 1 dbias (+=)
 * ZERO_INSERT       <- adjoint of SUBSAMPLE_BIAS; i: 0 B, 1 h, 2 w, 3 C, 4 ldc_s, 5 coff_s, 6 ldc_d, 7 coff_d
 *                   p: 0 src P (h,w), 1 dst P (2h,2w; whole padded domain written)
 * POOL2X            <- nn.AvgPool2d(2) / nn.MaxPool2d(2) of the pooled `down` variants (unet.py:93-94)
 *                   i: 0 B, 1 h, 2 w (output size), 3 C, 4 ldc_s, 5 coff_s, 6 ldc_d, 7 coff_d, 8 mode (0 avg, 1 max), 9 pm
 *                   p: 0 src P (2h,2w), 1 dst P (h,w; whole padded domain written, ring per pm)
 * POOL2X_BWD        i: 0 B, 1 h, 2 w, 3 C, 4 ldc_g, 5 coff_g, 6 ldc_x, 7 coff_x, 8 ldc_d, 9 coff_d, 10 fold, 11 mode, 12 pm
 *                   p: 0 g P (h,w), 1 x P (2h,2w) = POOL2X's input (max only: the gradient goes to the first maximum of the
 *                      window in row-major order, as torch's max_pool2d) | 0, 2 dst P (2h,2w; whole padded domain, ring 0)
 * DEPTH2SPACE       <- second half of nn.ConvTranspose2d(C, C, 2, stride=2) (unet.py:150): the first half is a CONV3X3 into 4C
 *                      channels with PREP_WEIGHTS kind 1 weights (phase q = 2a+b outermost, centre tap only)
 *                   i: 0 B, 1 h, 2 w, 3 C, 4 ldc_s, 5 coff_s, 6 ldc_d, 7 coff_d, 8 pm
 *                   p: 0 src P (h,w; 4C channels), 1 bias fp32 [C], 2 dst P (2h,2w; C channels; whole padded domain, ring per pm)
 * SPACE2DEPTH       <- its adjoint; i: 0 B, 1 h, 2 w, 3 C, 4 ldc_g, 5 coff_g, 6 ldc_d, 7 coff_d, 8 fold, 9 pm
 *                   p: 0 g P (2h,2w; C channels), 1 dst P (h,w; 4C channels; ring 0)
 * UPSAMPLE2X_BWD    i: 0 B, 1 h, 2 w, 3 C, 4 ldc_g, 5 coff_g, 6 ldc_d, 7 coff_d, 8 fold, 9 mode, 10 pm
 *                   p: 0 g P (2h,2w), 1 dst P (h,w interior)
 * UNPACK_GRAD       i: 0 B, 1 H, 2 W, 3 ldc, 4 nch, 5 pm; p: 0 d_in_pad P, 1 dx fp32 [B,nch,H,W]
 * ZERO              i: 0 bytes & 0x7fffffff, 1 bytes >> 31; p: 0 ptr
 * PREP_WEIGHTS      <- per-step weight re-layout (no reference counterpart: OIHW fp32 is the checkpoint format)
 *   i: 0 Cout, 1 Cin, 2 Cin_pad (K extent of Wf), 3 Cout_d (rows of Wd), 4 blocked flags (bit0: Wf K-blocked
 *      [Cin_pad/16][9][Cout_pad][16]; bit1: Wd likewise) -- set for the MFMA conv, clear for the REF conv,
 *      5 Cout_pad (rows of Wf, 0 = Cout), 6 Kd (K extent of Wd, 0 = Cout_pad); everything beyond the real Cout x Cin is zero:
 *      that is how channel counts that are not multiples of 8 / 16 run (zero-padded tensors, K overrun of the MFMA conv),
 *      7 kind: 0 conv3x3 (W [Cout][Cin][3][3]); 1 transposed 2x2 stride 2 (W [Cin][Cout][2][2] -> Wf [9][4*Cout][Cin_pad],
 *      Wd [9][Cout_d][Kd >= 4*Cout], centre tap only); 2 conv3x3 stride 2 in its 2x2 space-to-depth form (Cin = C, Cin_pad = Cout_d = 4C,
 *      flags = 3): Wf [4][Cout_pad][4C] with tap (dy,dx), K index (py,px,c) = W[co][c][2dy+py][2dx+px] (0 where an index is 3),
 *      Wd [(ty,tx)][4C][Kd] = Wf[(1-ty,1-tx)] transposed
 *   p: 0 W fp32 OIHW, 1 Wf [9][Cout_pad][Cin_pad], 2 |0 Wd [9][Cout_d][Kd] (taps flipped, in/out swapped: data gradient)
 * UNPREP_WGRAD      i: 0 Cout, 1 Cin, 2 Cin_w, 3 accumulate, 4 Cout_pad (rows of dWk, 0 = Cout), 5 kind (as PREP_WEIGHTS)
 *                   p: 0 dWk fp32 [9][Cout_pad][Cin_w], 1 dW fp32 OIHW (padded rows / columns dropped)
 * GRAD_NORM         <- clip_grad_norm_'s norm (ml/training/trainer.py:192-193)
 *   i: 0 n & 0x7fffffff, 1 n >> 31; p: 0 grads fp32, 1 sum of squares double* (+=; zero first)
 * ADAM              <- clip_grad_norm_'s scaling + torch.optim.Adam.step (no amsgrad, no weight decay; trainer.py:164-195)
 *   i: 0 n & 0x7fffffff, 1 n >> 31; f: 0 lr, 1 beta1, 2 beta2, 3 eps, 4 1-beta1^t, 5 sqrt(1-beta2^t),
 *      6 max_norm (<= 0: no clip), 7 grad scale; p: 0 params, 1 grads, 2 exp_avg, 3 exp_avg_sq, 4 |0 sum of squares
 * METRICS           <- the validation metrics of ml/training/metrics.py:7-97 in one pass (used by Trainer.validate,
 *                      ml/training/trainer.py:256-302); SSIM follows torchmetrics' published defaults (parity unpinned:
 *                      the library is not in the image)
 *   i: 0 B, 1 H, 2 W, 3 target bstride, 4 emitter bstride, 5 collider bstride, 6 pm, 7 stencil, 8 flags (bit0 SSIM); f: 0 dx
 *   p: 0 pred [B,3,H,W], 1 target, 2 emitter, 3 collider, 4 sums double[16] (zero first): [0..2] squared error per
 *      channel, [3],[4] |d/dx|,|d/dy| of the density error, [5] divergence^2, [6] vx^2+vy^2, [7] collider cells,
 *      [8] density in collider, [9] emitter cells, [10] density in emitter, [11] SSIM map sum over the cropped interior
 * GATHER_BATCH      <- FluidNPZSequenceDataset.__getitem__ + default collate + flip augmentation, from an HBM-resident
 *                      sequence store (ml/dataset/npz_sequence.py:349-433, ml/dataset/augmentations.py:20-35)
 *   i: 0 B, 1 H, 2 W, 3 K, 4 Km (mask frames, K or K+1), 5 mask_first (masks start at t + this), 6 Ttotal
 *   p: 0 frames fp32 [5][Ttotal][H][W] (density, velx, velz, emitter, collider; all sequences back to back, already
 *      normalised), 1 int32[B] global frame index of each sample's t, 2 int32[B] flip flags,
 *      3 x0 out [B,6,H,W], 4 y_seq out [B,K,3,H,W], 5 masks out [B,Km,2,H,W]
 * COPY_PLANES       <- the autoregressive state hand-off (ml/scripts/simple_infer.py:259-260)
 *   i: 0 B, 1 count, 2 src batch stride, 3 dst batch stride (floats); p: 0 src, 1 dst
 * STENCIL           <- compute_spatial_gradients / compute_divergence (physics_loss.py:19-91)
 *   i: 0 B, 1 H, 2 W, 3 pm, 4 stencil, 5 mode (0 gradients, 1 divergence); f: 0 dx, 1 dy; p: 0 f0, 1 f1|0, 2 out0, 3 out1|0
 */
#ifndef FLUIDNET_OPS_H
#define FLUIDNET_OPS_H

#include "fluidnet_hip.h"

/* named slots of the two ops a C++ host most often fills by hand */
enum FnetConvSlot {
    FNET_CONV_I_NPIX = 0, FNET_CONV_I_WP = 1, FNET_CONV_I_CIN = 2, FNET_CONV_I_LDC_IN = 3, FNET_CONV_I_COFF_IN = 4,
    FNET_CONV_I_COUT = 5, FNET_CONV_I_LDC_OUT = 6, FNET_CONV_I_COFF_OUT = 7, FNET_CONV_I_IMPL = 8,
    FNET_CONV_I_STATS_FLOATS = 9, FNET_CONV_I_H = 10,
    FNET_CONV_P_SRC = 0, FNET_CONV_P_WEIGHTS = 1, FNET_CONV_P_BIAS = 2, FNET_CONV_P_DST = 3, FNET_CONV_P_STATS = 4
};
enum FnetApplySlot {
    FNET_APPLY_I_B = 0, FNET_APPLY_I_H = 1, FNET_APPLY_I_W = 2, FNET_APPLY_I_C = 3, FNET_APPLY_I_LDC_RAW = 4,
    FNET_APPLY_I_COFF_RAW = 5, FNET_APPLY_I_LDC_RES = 6, FNET_APPLY_I_COFF_RES = 7, FNET_APPLY_I_LDC_DST = 8,
    FNET_APPLY_I_COFF_DST = 9, FNET_APPLY_I_ACT = 10, FNET_APPLY_I_PM = 11, FNET_APPLY_I_DROP_SITE = 12,
    FNET_APPLY_I_DROP_SEED = 13,
    FNET_APPLY_P_RAW = 0, FNET_APPLY_P_STATS = 1, FNET_APPLY_P_GAMMA = 2, FNET_APPLY_P_BETA = 3, FNET_APPLY_P_RES = 4,
    FNET_APPLY_P_DST = 5
};

#endif /* FLUIDNET_OPS_H */
