/*
 * hosrender.h -- C ABI of libhosrender.so: MI355X (gfx950) kernels for HOSNeRF's per-ray hot path.
 *
 * The reference (TencentARC/HOSNeRF) is 100% Python/torch and has no FFI of its own; its plugin
 * surface is the two nn.Module.forward signatures.  This library is the native
 * layer *below* that surface: hosnerf_amd/{mipnerf360,human_nerf,hosnerf}.py mirror the reference
 * modules and call these entry points through ctypes.  Each entry point cites the reference
 * lines (torch-op chain) it replaces.  Abbreviations:
 *   H: 3rd_Complete_HOSNeRF/src/model/mipnerf360/helper.py     M: .../mipnerf360/model.py
 *   N: 3rd_Complete_HOSNeRF/core/nets/human_nerf/network.py    U: .../core/utils/network_util.py
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + sizes + a hipStream_t (passed as void*); no torch types.
 *   - return 0 on success, a negative HOS_E_* code on bad arguments, or the positive hipError_t
 *     of a failed launch.  Never throws, never allocates, never synchronises: work is enqueued on
 *     `stream` and ordered with the caller's other work on that stream.
 *   - all tensors are fp32 row-major unless stated; "ld" = leading dimension in elements.
 *   - re-entrant per device; the caller owns all memory (workspaces are passed in).
 */
#ifndef HOSRENDER_H
#define HOSRENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOS_OK 0
#define HOS_E_ARG (-1)       /* null pointer / negative size                        */
#define HOS_E_ALIGN (-2)     /* pointer or leading dimension not 16-byte compatible */
#define HOS_E_SHAPE (-3)     /* unsupported shape (see the entry point)             */
#define HOS_E_NODEVICE (-4)  /* no HIP device visible                               */

typedef void* hos_stream_t; /* hipStream_t */

/* Library / device probes (no compute). */
int hos_version(void);                 /* 100*major + minor */
int hos_device_count(void);            /* >=0, or HOS_E_NODEVICE */
const char* hos_error_string(int code);

/* ------------------------------------------------------------------------------------------
 * Dense MLP contractions (fp32 MFMA v_mfma_f32_32x32x2_f32, exact-fp32 numerics).
 * Replaces every nn.Linear on the path: M:299-351 (PropMLP/NeRFMLP), canonical_mlps/
 * mlp_rgb_sigma.py:49-58, non_rigid_motion_mlps/mlp_offset.py:54-66.
 * All reduction dimensions must be multiples of 32 (buffers are zero-padded by the host side).
 * ------------------------------------------------------------------------------------------ */

/* Arithmetic mode of the three linear entry points on fp32 operands (hos_linear_fwd / _dgrad / _wgrad):
 *   HOS_GEMM_FP32   v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain (peak 157 TFLOP/s)
 *   HOS_GEMM_BF16X3 (library default) 16-bit hi/lo split of both operands on the fly -- fp16 pairs on the forward side (22
 *                   mantissa bits), bf16 pairs for gradients (8-bit exponent) -- 3 MFMAs per product, fp32 accumulate:
 *                   fp32-grade results (2.9e-5 RGB L-inf) at 1/3 of the 16-bit matrix rate (peak 833 TFLOP/s).
 * (The host side's own default is a third mode built on the same arithmetic: wide MLP trunks keep their activations as
 *  pre-split 16-bit planes and go through hos_linearp_*; see the planes section below.)
 * Layers with N <= 32 always use the fp32 kernel.
 * hos_set_gemm_mode        the process DEFAULT (set once at start-up);
 * hos_set_thread_gemm_mode an override for the CALLING THREAD only (-1 = follow the default): scoped switches ("this module
 *                          in exact fp32") use it, so host threads / modules pinned to different modes do not interfere;
 * hos_get_gemm_mode        the mode the calling thread's next launch will use. */
#define HOS_GEMM_MODE_FP32 0
#define HOS_GEMM_MODE_BF16X3 1
int hos_set_gemm_mode(int mode);
int hos_set_thread_gemm_mode(int mode);
int hos_get_gemm_mode(void);

/* Range guard of the split modes.  Forward activations travel as fp16 (hi, lo) pairs: exact to 2^-22 for |x| <= 65504; hi
 * saturates there and lo carries the residual up to 131008 (2^-11 relative); beyond that the pair saturates.  With a device word
 * registered here (caller-owned, NULL switches the reporting off) every forward epilogue that produces a hidden activation
 * (HOS_EPI_NONE / HOS_EPI_RELU, and the planes outputs) ORs 1 into it when |x| > 6e4, so the host can re-run the layer stack
 * in HOS_GEMM_MODE_FP32 (hosnerf_amd.ops.guarded_forward does).  Gradients use bf16 pairs (8-bit exponent) and need no guard. */
int hos_set_range_flag(unsigned int* flag);

/* epilogues for hos_linear_fwd */
#define HOS_EPI_NONE 0       /* C = acc + bias                                                */
#define HOS_EPI_RELU 1       /* C = relu(acc + bias)                                          */
#define HOS_EPI_DENSITY 2    /* N==1: aux[m] = softplus(acc + bias + density_bias)   (M:316)   */
#define HOS_EPI_RGB 3        /* C = sigmoid(acc+bias)*(1+2*pad) - pad               (M:345-346)*/
#define HOS_EPI_NERF_HEAD 4  /* cols < aux_col: C = acc+bias; col == aux_col: aux[m] = softplus(acc+bias+density_bias) */
#define HOS_EPI_SIGMOID_RELU4 5 /* N==4: cols 0..2 sigmoid, col 3 relu (N:539-540)             */
#define HOS_EPI_RESIDUAL 6   /* C = acc + bias + aux[m*aux_col + n]  (xyz + offset, mlp_offset.py:66; aux_col = ld of aux) */

/* `rows_dev` (several entry points below; may be NULL): the number of LIVE rows of a fixed-capacity buffer, read from device
 * memory by the kernel -- rows (whole row tiles for the GEMMs) at or past it are skipped.  Used for the cycle-consistency set
 * of the human branch, whose size is data dependent (hos_compact_rows): shapes stay static, the work follows the count.
 *
 * C[M,N] = epi( [A0 | A1][M, K0+K1] @ W[N, K0+K1]^T + bias[N] ).
 * A1 may be NULL (K1 = 0); K0, K1 multiples of 32; W row n starts at W + n*ldw.  */
int hos_linear_fwd(const float* A0, int lda0, int K0, const float* A1, int lda1, int K1,
                   const float* W, int ldw, const float* bias, float* C, int ldc,
                   int M, int N, int epilogue, float* aux, int aux_col, float p0, float p1,
                   const int32_t* rows_dev, hos_stream_t stream);

/* C[M,N] = A[M,K] . W[N,K]^T in exact fp32 MFMA for a small output with a long reduction (K % 32 == 0; the input gradients of
 * the volume decoder's ConvTranspose3d layers, deconv_vol_decoder.py:17-42): the reduction is split over ~512 workgroups that
 * accumulate into the zeroed C with fp32 atomics (sums in a launch-dependent order).  Replaces a library GEMM call. */
int hos_linear_fwd_splitk(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                          hos_stream_t stream);
/* Round 5: the same split with a FIXED summation order (bit-reproducible from call to call): partial tiles as plain stores into `ws`
 * (>= hos_linear_fwd_splitk_ws_floats(M, N, K) floats, 16-byte aligned, caller-owned), summed in slab order by a second launch that
 * also adds bias (may be NULL) and applies ReLU (relu != 0).  LPIPS' deep convolutions (third_parties/lpips/pretrained_networks.py:
 * 105-131) use it: with atomics a pre-activation within rounding of 0 flipped a ReLU / pooling winner from run to run. */
long long hos_linear_fwd_splitk_ws_floats(int M, int N, int K);
int hos_linear_fwd_splitk_det(const float* A, int lda, const float* W, int ldw, const float* bias, int relu, float* C, int ldc,
                              int M, int N, int K, float* ws, long long ws_floats, hos_stream_t stream);

/* dX[M,K] = (dY[M,Npad] @ W[Npad,K]) (* (Xact[M,K] > 0) if Xact != NULL).
 * Npad (the reduction dim = padded layer width) multiple of 32; rows >= N of W must be zero.
 * If accumulate != 0, dX += result (used for skip / multi-consumer activations). */
int hos_linear_dgrad(const float* dY, int lddy, const float* W, int ldw, int Npad,
                     const float* Xact, int ldx, float* dX, int lddx, int M, int K,
                     int accumulate, hos_stream_t stream);

/* dW[N,K] += dY[M,N]^T @ X[M,K]   and, if db != NULL, db[N] += column sums of dY.
 * M (the reduction dim = number of sample points) is arbitrary (tail rows are zero-filled).  Accumulates with fp32
 * atomics over `splits` partitions of M (splits <= 0: chosen by the library). */
int hos_linear_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int ldw,
                     float* db, int M, int N, int K, int splits, hos_stream_t stream);

/* Fused backward of one thin layer (N, K <= 128) over M rows -- replaces a hos_linear_wgrad + hos_linear_dgrad pair on the
 * same (dY, X): dX [M,K] = (dY . W) masked by X > 0 (relu_mask != 0; dX NULL: skip), dW [N,ldw] += dY^T . X, db [N] += column
 * sums of dY (NULL: skip).  dY [M, lddy] has its columns >= N zero up to the next multiple of 32; W points at the first
 * column of the K-slice of the layer's weight [N, ldw] (nn.Linear layout).  bf16 hi/lo x3 products like the split GEMMs.
 * ws / ws_floats: optional scratch (>= 256*(128*128+128) floats) for per-workgroup dW / db partials; NULL falls back to fp32 atomics.
 * Reference: autograd of nn.Linear + ReLU in mlp_offset.py:54-70 (non-rigid MLPs, 128 wide, M = rays x 128). */
int hos_linear_bwd_fused(const float* dY, int lddy, const float* X, int ldx, const float* W, int ldw,
                         float* dX, int lddx, float* dW, int lddw, float* db, int M, int N, int K,
                         int relu_mask, float* ws, int64_t ws_floats, const int32_t* rows_dev,
                         hos_stream_t stream);

/* Deferred slab reductions of hos_linear_bwd_fused / hos_linear_wgrad_tr.  Each of those calls ends in a second launch
 * that sums its per-workgroup partials (ws) into dW / db.  Between hos_mlp_bwd_defer(1) and hos_mlp_bwd_flush() the calls
 * of the calling thread only RECORD that reduction (each call then needs its own ws region of hos_mlp_bwd_ws_floats(M, N, K,
 * fused) floats; 0 = that call uses atomics and needs none); hos_mlp_bwd_flush runs all recorded reductions in one launch
 * per 16 and dW / db are complete after it; hos_mlp_bwd_defer(0) restores the immediate reduction.  A thin MLP's backward
 * is 6-8 such calls: 40 reduction launches per stage-2 step (0.72 ms) become 4. */
int hos_mlp_bwd_defer(int on);
int hos_mlp_bwd_flush(hos_stream_t stream);
long long hos_mlp_bwd_ws_floats(int M, int N, int K, int fused);

/* WGRAD of a layer up to 256 x 256 with the staging of hos_linear_bwd_fused (operands split once into LDS planes, transposed
 * LDS reads instead of in-register transposes): dW [N,ldw] += dY^T . X, db [N] += column sums (NULL: skip).  ws: optional
 * scratch (>= 256*(256*256+256) floats) for the per-workgroup dW / db partials (NULL: fp32 atomics).  Replaces hos_linear_wgrad for M >> N, K.
 * Reference: autograd of the 256-wide nn.Linear layers of CanonicalMLP, canonical_mlps/mlp_rgb_sigma.py:49-58. */
int hos_linear_wgrad_tr(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw, float* db,
                        int M, int N, int K, float* ws, int64_t ws_floats, hos_stream_t stream);

/* Thin layers over very many rows with the weight slice of every wave resident in registers (hos_thin.hip): N, K <= 256
 * (forward: K <= 320, the folded skip layer below).
 *   hos_thin_linear_fwd  : Y [M,N] = epi(X[:, :K] . W[:N, :K]^T + bias), epilogue HOS_EPI_NONE / HOS_EPI_RELU (fp16 hi/lo x3)
 *   hos_thin_linear_dgrad: dX [M,K] = (dY[:, :Npad] . W[:Npad, :K]) * [mask > 0]  (bf16 hi/lo x3; mask NULL: none); W and mask
 *                          may start at any column of their matrices (4-byte alignment: the h part of the skip layer's concat row
 *                          starts at column 127, mlp_rgb_sigma.py:53-55), dY and dX rows are 16-byte aligned
 *   relu_bits / mask_bits (optional, 1024 * ceil(M/32) bytes, opaque): the ReLU mask of the backward pass as ONE BIT per output
 *                          element, written by the forward launch and read by the dgrad launch of the layer above instead of
 *                          the fp32 activations (a third of that launch's HBM traffic); takes precedence over `mask`
 * Same results contract as hos_linear_fwd / hos_linear_dgrad in split mode.
 * Reference: CanonicalMLP, canonical_mlps/mlp_rgb_sigma.py:49-58 (256-wide Linear + ReLU chain at M = rays x 128). */
int hos_thin_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                        int M, int N, int K, int epilogue, void* relu_bits, hos_stream_t stream);
int hos_thin_linear_dgrad(const float* dY, int lddy, const float* W, int ldw, int Npad, const float* mask, int ldmask,
                          const void* mask_bits, float* dX, int lddx, int M, int K, hos_stream_t stream);

/* CanonicalMLP with the state embedding folded into biases.  The reference concatenates ONE state vector per call to every
 * point's Fourier features (core/nets/human_nerf/network.py:177-230 picks it by frame time and expands it over the points), so its columns of the input
 * layer W0 [n_out, nf + ne] and of the skip layer W5 [n_out, nf + ne + nh] act as per-call biases:
 *   hos_canonical_fold_pack    W0f [n_out, nfp] = W0[:, :nf] | 0,  b0f = b0 + W0[:, nf:nf+ne] . embed,
 *                              W5f [n_out, nfp + nh] = W5[:, :nf] | 0 | W5[:, nf+ne:],  b5f likewise  (nfp = nf rounded up to 4):
 *                              rows of 64 / 320 instead of 128 / 384 columns, the h part 16-byte aligned
 *   hos_canonical_fold_unfold  gradients of the folded layers (gW0f, db0, gW5f, db5: accumulated by the usual backward launches)
 *                              += into the reference-shaped gW0 / gb0 / gW5 / gb5, the state columns as db (x) embed, and
 *                              g_embed [ne] += W0[:, nf:nf+ne]^T db0 + W5[:, nf:nf+ne]^T db5  (fixed-order fp32 sums) */
int hos_canonical_fold_pack(const float* W0, int ld0, const float* b0, const float* W5, int ld5, const float* b5,
                            const float* embed, int n_out, int nf, int ne, int nh,
                            float* W0f, float* b0f, float* W5f, float* b5f, hos_stream_t stream);
int hos_canonical_fold_unfold(const float* gW0f, const float* db0, const float* gW5f, const float* db5,
                              const float* W0, int ld0, const float* W5, int ld5, const float* embed,
                              int n_out, int nf, int ne, int nh,
                              float* gW0, float* gb0, float* gW5, float* gb5, float* g_embed, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * "Planes" form of the same three contractions: operands are pre-split 16-bit hi/lo values (fp16 on the forward
 * side, bf16 for gradients), staged by LDS-DMA, 3 MFMAs per product.  The producer of a tensor does the split once;
 * the trunk of the MLPs (M:299-303 forward, its autograd backward) then runs without conversion work in the K loop.
 * Storage ("interleaved planes"): a matrix [R][ld], ld % 32 == 0, is ONE 16-bit array [R][ld/32][2][32] -- per row
 * and 32-column block the 32 hi values, then the 32 lo values (128 bytes); element (r,c) = hi + lo with
 * hi at r*2*ld + (c/32)*64 + c%32 and lo 32 elements further.  All `ld` arguments are logical column counts.
 * ------------------------------------------------------------------------------------------ */

/* fp32 [R][lds] -> planes.  dtype 0 = fp16, 1 = bf16.  Row-major planes out [R][ldo] (columns [C,ldo) zeroed)
 * and/or transposed planes outT [C][ldt] (columns [R,ldt) zeroed); either may be NULL. */
int hos_split_planes(const float* src, int lds, int R, int C, int dtype, void* out, int ldo,
                     void* outT, int ldt, hos_stream_t stream);

/* n <= 12 transposed bf16 splits in ONE launch: src[i] fp32 [R[i]][lds[i]] (C[i] columns) -> planes outT[i] [C[i]][ldt[i]],
 * columns [R[i], ldt[i]) zeroed -- the transposed weight planes hos_linearp_dgrad needs for every layer of an MLP.  The argument
 * arrays are HOST arrays of n entries, read during the call. */
int hos_split_planes_t_batch(int n, const float* const* src, const int* lds, const int* R, const int* C, void* const* outT,
                             const int* ldt, hos_stream_t stream);

/* fp32 [R][lds] -> fp16 planes [R][ld16] and/or bf16 planes [R][ldb] in one pass (padding columns zeroed). */
int hos_split_planes2(const float* src, int lds, int R, int C, void* p16, int ld16, void* pb, int ldb,
                      hos_stream_t stream);

/* Forward: acc[M,N] = [A | A1][M,K0+K1] @ W[N,K0+K1]^T (fp16 planes) + bias.
 *  - plane outputs (Y != NULL and/or Yb != NULL): relu?(acc) as fp16 planes [M][ldy] (input of the next layer)
 *    and/or as bf16 planes [M][ldyb] (weight-gradient operand); padding columns zeroed; with relu and Y, relu_bits
 *    (may be NULL) receives one bit per element, acc + bias > 0: uint32 [ceil(M/32)][ceil(ldy/64)][64], per 32-row x
 *    64-column block in the accumulator layout of the kernel (dword l + 32 h, bit 31 - (16 y + r) = row (r&3) + 8 (r>>2) + 4 h,
 *    column 32 y + l) -- the ReLU mask hos_linearp_dgrad reads back, opaque to everything else;
 *  - otherwise the fp32 epilogues of hos_linear_fwd (C/ldc/epilogue/aux/aux_col/p0). */
int hos_linearp_fwd(const void* A, int lda, int K0, const void* A1, int lda1, int K1, const void* W, int ldw,
                    const float* bias, int M, int N, int relu, void* Y, int ldy, void* Yb, int ldyb, void* relu_bits,
                    float* C, int ldc, int epilogue, float* aux, int aux_col, float p0, hos_stream_t stream);

/* One-column head on fp16 planes: out[M] = act( A[M, :K] . w[:K] + bias[0] + p0 ); A planes [M][lda], K % 32 == 0, w fp32 [K]
 * (not split: exact operand), bias a device scalar or NULL, act = torch.nn.Softplus if softplus != 0.  One pass over A instead of
 * a 128-wide GEMM tile per 256 rows: the density heads of the proposal MLPs (Linear(width, 1)) and the density column of the NeRF
 * MLP's head (mipnerf360/model.py:158-160, 325).  Sums in a fixed order. */
int hos_planes_rowdot(const void* A, int lda, int K, const float* w, const float* bias, float p0, int softplus, int64_t M,
                      float* out, hos_stream_t stream);

/* Round 5: the same two calls with ONE activation format, bf16 planes, for layer stacks whose outputs are only rendered (the NeRF MLP,
 * mipnerf360/model.py:354-362: no resampling depends on its densities).  A, A1, W and the plane output Y are bf16 planes; Y's ReLU
 * bit mask (relu_bits) and Y itself are what hos_linearp_dgrad / hos_linearp_wgrad read, so the layer writes its output once
 * (hos_linearp_fwd writes fp16 planes for the next layer AND bf16 planes for the weight gradient).  Products as in the backward
 * kernels: a_hi b_hi + a_hi b_lo + a_lo b_hi on bf16 pairs (2^-17 per product, fp32 accumulation); no fp16 range to guard. */
int hos_linearp_fwd_b(const void* A, int lda, int K0, const void* A1, int lda1, int K1, const void* W, int ldw,
                      const float* bias, int M, int N, int relu, void* Y, int ldy, void* relu_bits,
                      float* C, int ldc, int epilogue, float* aux, int aux_col, float p0, hos_stream_t stream);
int hos_planes_rowdot_b(const void* A, int lda, int K, const float* w, const float* bias, float p0, int softplus, int64_t M,
                        float* out, hos_stream_t stream);

/* Data gradient: dX[M,K] = dZ[M,Npad] @ WT[K,Npad]^T (bf16 planes; WT = transposed weight planes), masked by the
 * ReLU bit mask hos_linearp_fwd wrote next to the layer input (mask_bits != NULL, ldmask = that input's ld; see
 * relu_bits there) or by the fp16 planes of the layer input themselves (hi > 0, mask != NULL); bf16 planes [M][lddx]. */
int hos_linearp_dgrad(const void* dZ, int lddz, const void* WT, int ldwt, int Npad, const void* mask, int ldmask,
                      const void* mask_bits, int M, int K, void* dX, int lddx, hos_stream_t stream);

/* Weight gradient: dW[N,K] += dZ[M,N]^T @ X[M, x_col0 : x_col0+K] (both ROW-MAJOR bf16 planes; the
 * reduction-contiguous MFMA fragments are gathered from LDS with ds_read_b64_tr_b16), db[N] += column sums of dZ.
 * M % 32 == 0, x_col0 % 32 == 0.  The split-K partial tiles are written to `ws` (>= splits * N * round4(K) floats,
 * caller-owned scratch, may be NULL) and summed by a second launch; without a workspace they are accumulated with
 * fp32 atomics. */
int hos_linearp_wgrad(const void* dZ, int lddz, const void* X, int ldx, int x_col0, float* dW, int ldw, float* db,
                      int M, int N, int K, int splits, float* ws, long long ws_floats, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-frame ray set-up on the device (core/utils/camera_util.py:154-265 of stage 3).
 * Kinv9 / R9 / T3 / bounds6 are HOST pointers (camera scalars are passed to the kernel by value); the ray arrays
 * are device pointers.
 * ------------------------------------------------------------------------------------------ */

/* get_rays_from_KRT (C:154-183) and, when viewdirs / radii are non-NULL, get_rays_from_KRT_bkg (C:185-216):
 * rays_o, rays_d, viewdirs [H*W,3], radii [H*W] for an H x W image, row-major pixels (pixel p = row*W + col). */
int hos_camera_rays(const float* Kinv9, const float* R9, const float* T3, int H, int W, float* rays_o, float* rays_d,
                    float* viewdirs, float* radii, hos_stream_t stream);

/* rays_intersect_3d_bbox (C:219-265) for every ray: mask[p] = 1 iff exactly two of the six plane hits lie inside the
 * box (bounds6 = min xyz, max xyz; grown by 0.01), near/far [n] (0 where invalid).  Like the reference it clamps
 * direction components with magnitude < 1e-5 to 1e-5 IN PLACE in rays_d. */
int hos_rays_aabb(const float* rays_o, float* rays_d, int64_t n, const float* bounds6, float* near, float* far,
                  unsigned char* mask, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ConvTranspose3d(kernel 4, stride 2, padding 1) of the motion-weight volume decoder (network_util.py:21-59,
 * deconv_vol_decoder.py:17-42) = GEMM (hos_linear_*) + these two gathers; activations are channel-last [voxel][C].
 * ------------------------------------------------------------------------------------------ */

/* out[(2D)^3][Cout] = act(bias + taps of ycol[D^3][Cout*64]); act = LeakyReLU(leaky_slope) if leaky else identity. */
int hos_deconv3d_col2im(const float* ycol, const float* bias, int D, int Cout, float leaky_slope, int leaky,
                        float* out, hos_stream_t stream);

/* dycol[D^3][Cout*64] = gather of dpre[(2D)^3][Cout] (gradient w.r.t. the pre-activation output); zeros where a tap
 * leaves the output volume. */
int hos_deconv3d_im2col(const float* dpre, int D, int Cout, float* dycol, hos_stream_t stream);
/* Round 5, volume decoder sharded over the data-parallel ranks by INPUT channel (deconv_vol_decoder.py:17-42 replicated in the
 * reference, run.py:173-190 DDP): hos_bias_lrelu: y [M, N] = LeakyReLU?(y + bias [N]) in place, behind the all-reduce of the ranks'
 * partial pre-activations; hos_shard_interleave: out [M, world * cs], out[m][r * cs + c] = parts[r][m][c] -- the all-gathered
 * per-rank input-gradient slices in channel order. */
int hos_bias_lrelu(float* y, const float* bias, long long M, int N, float leaky_slope, int leaky, hos_stream_t stream);
int hos_shard_interleave(const float* parts, int world, int M, int cs, float* out, hos_stream_t stream);

/* Backward of the block's LeakyReLU and its bias gradient in one pass over the output gradient g [R, C] (channel-last, R =
 * (2D)^3): dpre = g * (out > 0 ? 1 : leaky_slope) if leaky != 0 (else g is already the pre-activation gradient and dpre is not
 * written), db [C] += column sums of it (NULL: skip).  deconv_vol_decoder.py:17-42. */
int hos_deconv3d_dpre(const float* g, const float* out, long long R, int C, float leaky_slope, int leaky, float* dpre, float* db,
                      hos_stream_t stream);
/* Weight gradient of the decoder's first layers (1, 8, 64 input voxels): gW [K, ldw] += x[M, :K]^T . dy[M, :N], M <= 64,
 * exact fp32, one read-add-write pass over the weights.  Reference: deconv_vol_decoder.py:34-42 (ConvTranspose3d autograd). */
int hos_outer_accum(const float* x, int ldx, const float* dy, int lddy, float* gW, int ldw, int M, int K, int N,
                    hos_stream_t stream);
/* One row against a long weight stream: y [N] = act(x [K] . W [K, ldw][:, :N] + bias[n % bias_mod]) (act: identity or LeakyReLU with
 * `leaky_slope`), exact fp32 with a fixed summation order (16-row slabs, then the slabs in ascending order); ws: scratch of
 * hos_gemv_ws_floats(K, N) floats.  The first ConvTranspose3d of the volume decoder sees ONE voxel (network_util.py:21-59,
 * deconv_vol_decoder.py:34-42): with its 8 live taps stored tap-major, x . Wc viewed as [8, Cout] + bias IS its output. */
long long hos_gemv_ws_floats(int K, int N);
int hos_gemv_rowvec(const float* x, const float* W, int ldw, int K, int N, const float* bias, int bias_mod,
                    float leaky_slope, int leaky, float* ws, float* y, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Background branch, per-ray kernels (one wavefront per ray).
 * ------------------------------------------------------------------------------------------ */

/* Proposal resampling: replaces H:187-194 (max_dilate_weights) + M:469-482 (trim, logits) +
 * H:373-399 (sample_intervals -> sample -> invert_cdf -> sorted_interp) + H:169-174 (s_to_t).
 * n_prev == 1 selects the level-0 path (t=[0,1], w=1, no dilation).
 *   sdist_prev [B,n_prev+1], w_prev [B,n_prev]       previous level histogram
 *   u_base [S]        torch.linspace grid of H:354 / H:363 (host supplies it; bit-exact u)
 *   jitter [B] or NULL, jitter_scale = max_jitter of H:360 (train); NULL -> u = u_base (eval)
 *   outputs: sdist [B,S+1], tdist [B,S+1], bin_idx [B,S] int32 (optional, may be NULL):
 *            index of the CDF knot left of every sample centre (bit-exact target of SURVEY B4).
 *   anneal = bias(train_frac, anneal_slope) of M:459-460; if train_frac_dev != NULL it is evaluated on the device from
 *   *train_frac_dev instead (a step captured in a hipGraph anneals like an eager one; `anneal` is then ignored). */
int hos_resample(const float* sdist_prev, const float* w_prev, int n_prev, int B, int S,
                 float dilation, float anneal, const float* train_frac_dev, float anneal_slope,
                 float resample_padding, const float* u_base, const float* jitter, float jitter_scale,
                 float near_, float far_, float* sdist, float* tdist, int32_t* bin_idx,
                 hos_stream_t stream);

/* Conical-frustum cast + contraction + lift + integrated positional encoding (+ state embedding
 * columns + zero pad): replaces H:279-339, H:33-68 (closed-form Jacobian), H:71-89, M:295-296.
 *   X [B*S, ldx]: cols [0,504) IPE (level-major, 21 dirs; sin part then sin(.+pi/2) part),
 *                 [504,568) = state embedding, [568,ldx) = 0.   basis [3,21] row-major. */
int hos_encode_ipe(const float* tdist, const float* rays_o, const float* rays_d, const float* radii,
                   const float* basis, const float* embed, int B, int S, float* X, int ldx,
                   hos_stream_t stream);

/* Same encoder, rows written directly as the interleaved 16-bit planes of the planes GEMMs (fp16 planes p16 [P][ld]
 * and, if pb != NULL, bf16 planes): the MLP trunk then needs no fp32 copy of the encoding.  ld % 32 == 0, ld >= 568. */
int hos_encode_ipe_planes(const float* tdist, const float* rays_o, const float* rays_d, const float* radii,
                          const float* basis, const float* embed, int B, int S, void* p16, void* pb, int ld,
                          hos_stream_t stream);

/* View-direction encoding (H:93-100, deg 0..4, identity appended = 27) broadcast over samples
 * into columns [col0, col0+27) of Xv [B*S, ldx]; columns [col0+27, ldx) are zeroed (M:330-335). */
int hos_encode_viewdirs(const float* viewdirs, int B, int S, float* Xv, int ldx, int col0,
                        hos_stream_t stream);

/* compute_alpha_weights (H:235-261):  weights [B,S]  from density [B,S], tdist [B,S+1], dirs [B,3]. */
int hos_alpha_weights_fwd(const float* density, const float* tdist, const float* dirs, int B, int S,
                          int opaque_background, float* weights, hos_stream_t stream);
/* g_density [B,S] from g_weights [B,S] (recomputes alpha/trans). */
int hos_alpha_weights_bwd(const float* g_weights, const float* density, const float* tdist,
                          const float* dirs, int B, int S, int opaque_background, float* g_density,
                          hos_stream_t stream);

/* volumetric_rendering (H:265-275): rgb [B,3] = sum w*c + clip(1-sum w,0)*bg. */
int hos_volrender_fwd(const float* rgbs, const float* weights, int B, int S, float bg, float* rgb,
                      hos_stream_t stream);
int hos_volrender_bwd(const float* g_rgb, const float* rgbs, const float* weights, int B, int S,
                      float bg, float* g_rgbs, float* g_weights, hos_stream_t stream);

/* Stage-1 losses (M1:611-627 -> H:136-149).  Per-ray partial losses are written to loss_ray [B]
 * (the host sums / means them); the backward kernels take the upstream scale g (dLoss/dmean / B).
 *   interlevel: c [B,Sc+1], w [B,Sc] (detached NeRF histogram), cp [B,Sp+1], wp [B,Sp].
 *   idx_lo/idx_hi [B,Sc+1] int32 optional outputs (bit-exact vs H:109-114). */
int hos_interlevel_fwd(const float* c, const float* w, const float* cp, const float* wp, int B,
                       int Sc, int Sp, float* loss_ray, int32_t* idx_lo, int32_t* idx_hi,
                       hos_stream_t stream);
int hos_interlevel_bwd(const float* c, const float* w, const float* cp, const float* wp, int B,
                       int Sc, int Sp, float scale, float* g_wp, hos_stream_t stream);
int hos_distortion_fwd(const float* t, const float* w, int B, int S, float* loss_ray,
                       hos_stream_t stream);
int hos_distortion_bwd(const float* t, const float* w, int B, int S, float scale, float* g_w,
                       hos_stream_t stream);

/* Activation derivatives that turn (g_density, g_rgb) into pre-activation gradients, written
 * into the zero-padded dY buffers the dgrad/wgrad kernels consume.
 *   dz_density[m*ld_dd + col_dd] = g_density[m] * (1 - exp(-density[m]))     (softplus')
 *   dz_rgb[m*ld_dr + c]          = g_rgb[m,c] * (1+2pad) * s*(1-s), s=(rgb+pad)/(1+2pad)  */
int hos_head_grad(const float* g_density, const float* density, const float* g_rgb, const float* rgb,
                  int P, float rgb_padding, float* dz_density, int ld_dd, int col_dd,
                  float* dz_rgb, int ld_dr, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Human-object branch, per-sample kernels (N: = core/nets/human_nerf/network.py of stage 3).
 * ------------------------------------------------------------------------------------------ */

/* Ray samples + backward LBS warp: replaces N:409-424 (_get_samples_along_ray, _stratified_sampling),
 * N:451 (pts = o + d z) and N:304-355 (_sample_motion_fields: 26 rigid maps, 26 single-channel
 * grid_sample taps, weight-normalised blend).
 *   t_vals [N] = torch.linspace(0,1,N) (host supplies it);  t_rand [B*N] uniform draws or NULL (cfg.perturb == 0)
 *   R [K,3,3], T [K,3] backward motion basis;  vol [>=K, V,V,V] motion-weight volume (z,y,x order)
 *   outputs: z_vals [B*N] (may be NULL), pts [B*N,3] (may be NULL), x_skel [B*N,3], mask [B*N]. */
int hos_human_sample_warp(const float* rays_o, const float* rays_d, const float* near_, const float* far_,
                          const float* t_vals, const float* t_rand, const float* R, const float* T,
                          const float* vol, int V, const float* bbox_min, const float* bbox_scale,
                          int B, int N, int K, float* z_vals, float* pts, float* x_skel, float* mask,
                          hos_stream_t stream);

/* Forward LBS (N:357-399): one K-channel tap at each canonical point, blend of the K forward maps.
 * vol_cl is the volume in channel-LAST layout [V,V,V,CL] (CL >= K, multiple of 4). */
int hos_lbs_forward(const float* cnl_pts, const float* R_fwd, const float* T_fwd, const float* vol_cl,
                    int V, int CL, const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                    float* x_deform, const int32_t* rows_dev, hos_stream_t stream);

/* Hann-windowed positional encoding of the non-rigid MLP (embedders/hannw_fourier.py:15-71) written
 * as the MLP's first-layer input row  E[p] = [cond(cond_size) | w_j sin(2^j x), w_j cos(2^j x) | 0]
 * (mlp_offset.py:55) and optionally the features alone into PE [P, ldpe] for the skip concat (:59-60). */
int hos_embed_hannw(const float* x, const float* band_w, int num_freqs, const float* cond, int cond_size,
                    int64_t P, float* E, int lde, float* PE, int ldpe, const int32_t* rows_dev, hos_stream_t stream);

/* Canonical-MLP input row  E[p] = [x, sin(2^j x), cos(2^j x) (j<num_freqs) | state embedding | 0]
 * (embedders/fourier.py:11-57, N:248-249); E2 (optional) receives the same 3+6F+state_size columns
 * (the skip-concat buffer of mlp_rgb_sigma.py:52-53). */
int hos_embed_fourier(const float* x, int num_freqs, const float* state, int state_size, int64_t P,
                      float* E, int lde, float* E2, int lde2, hos_stream_t stream);

/* Backward of hos_human_sample_warp w.r.t. the motion-weight volume (atomics into g_vol [K,V,V,V]) and the
 * backward motion basis (g_R [K,9], g_T [K,3], accumulated with atomics -- caller zeroes them):
 * the autograd of N:318-340 incl. grid_sample's gradient w.r.t. the grid (bone transforms move the tap). */
int hos_human_sample_warp_bwd(const float* pts, const float* R, const float* T, const float* vol, int V,
                              const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                              const float* g_x_skel, const float* g_mask, float* g_vol, float* g_R, float* g_T,
                              float* scratch, const float* fwd_x_skel, const float* fwd_mask, hos_stream_t stream);
/*   scratch: optional [P,2] floats; with it (and V^3 floats fitting LDS) the volume gradient is accumulated per bone in LDS
 *   by a second kernel instead of scattered global atomics per point (NULL: single kernel).
 *   fwd_x_skel [P,3] / fwd_mask [P]: optional, the outputs hos_human_sample_warp wrote for these points; with them the kernel
 *   does not re-evaluate the K x 8 taps of the forward pass per point (both NULL: recomputed, bit-identical result). */
/* Backward of hos_lbs_forward: g_cnl [P,3] (written), g_vol_cl / g_R / g_T accumulated with atomics. */
int hos_lbs_forward_bwd(const float* cnl_pts, const float* R_fwd, const float* T_fwd, const float* vol_cl,
                        int V, int CL, const float* bbox_min, const float* bbox_scale, int64_t P, int K,
                        const float* g_x_deform, float* g_cnl, float* g_vol_cl, float* g_R, float* g_T,
                        const int32_t* rows_dev, hos_stream_t stream);
/* Backward of both positional embedders w.r.t. x: feature gradients are read from dA[:, colA:] (+ dB[:, colB:]
 * if not NULL); band_w NULL = plain Fourier (weights 1); identity != 0 = features start with x itself. */
int hos_embed_bwd(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda,
                  int colA, const float* dB, int ldb, int colB, int64_t P, float* g_x, int accumulate,
                  const int32_t* rows_dev, hos_stream_t stream);
/* out[p,c] = src[p*lds+col0+c] * (mask_src[p*ldm+mcol0+c] > 0), c < width (mask_src may be NULL). */
int hos_slice_mask(const float* src, int lds, int col0, const float* mask_src, int ldm, int mcol0, int64_t P,
                   int width, float* out, int ldo, const int32_t* rows_dev, hos_stream_t stream);
/* out[p, :] = [src[p*lds+col0 .. +width) | 0 ...] over the WHOLE [P, ldo] row (width <= 4, ldo % 4 == 0): a [P,3] gradient widened
 * to the zero-padded operand row of hos_linear_bwd_fused without a fill (autograd of the offset head, mlp_offset.py:63-70). */
int hos_slice_pad(const float* src, int lds, int col0, int64_t P, int width, float* out, int ldo,
                  const int32_t* rows_dev, hos_stream_t stream);
/* dz[p, 0..3] = g * (sigmoid' | relu') evaluated from the activated outputs (N:539-540); the whole [P, ldz] row is written
 * (columns 4.. zero; ldz % 4 == 0, 16-byte aligned pointers). */
int hos_rgbsigma_grad(const float* g_rgbsigma, const float* rgbsigma, int64_t P, float* dz, int ldz,
                      hos_stream_t stream);

/* NeRF-style composite `_raw2outputs` (M:73-99; S2 form N2:273-299 with the activations applied by
 * the MLP epilogue): dists = [dz..., last_dist]*|d|; alpha = (1-exp(-sigma*dists))*mask;
 * w = alpha * cumprod([1, 1-alpha+1e-10])[:-1]; rgb = sum w c (+ (1-sum w)*bgcolor/255 if bgcolor).
 * rgbs / sigma are strided views (element (ray,s) at [(ray*S+s)*ld]) so a packed [B,S,4] buffer works.
 * mask / bgcolor / acc / weights / depth may be NULL. */
int hos_raw2outputs_fwd(const float* rgbs, int rgb_ld, const float* sigma, int sigma_ld, const float* z_vals,
                        const float* rays_d, const float* mask, const float* bgcolor, float last_dist,
                        int B, int S, float* rgb, float* acc, float* weights, float* depth, hos_stream_t stream);
/* gradients w.r.t. rgbs / sigma / mask from g_rgb [B,3] (+ optional g_weights [B,S]). */
int hos_raw2outputs_bwd(const float* g_rgb, const float* g_weights, const float* rgbs, int rgb_ld,
                        const float* sigma, int sigma_ld, const float* z_vals, const float* rays_d,
                        const float* mask, const float* bgcolor, float last_dist, int B, int S,
                        float* g_rgbs, int g_rgb_ld, float* g_sigma, int g_sigma_ld, float* g_mask,
                        hos_stream_t stream);

/* Stage-3 composite of one ray batch (M:1524-1596, the inline block of LitMipNeRF360.training_step):
 *   C1  z_h = mean_xyz((A[p,1] - o_b)/(d_b + 1e-10))   (first non-tiny component when *tiny_d_flag != 0)
 *   C2  fg = sum(mask) > thre_fg; stable z-sort of [bkg tdist[:-1] (Sb) | human z_h (Sh)] -> total_order
 *   C3  masked _raw2outputs over the merged samples (fg rays) or the Sb background samples (bg rays)
 * Outputs: rgb [B,3]; idx_fg [B] (0/1); total_order [B,Sb+Sh] int32 (-1 on bg rays) -- the bit-exact
 * index target; human_weights_sorted [B,Sh] = composite weights of the human samples in sorted order
 * (rows of bg rays are 0) = `human_weights_onlyfg` of M:1588 after row selection; z_human [B,Sh].
 * All optional outputs may be NULL. */
int hos_merge_composite_fwd(const float* bkg_tdist, const float* bkg_rgb, const float* bkg_density,
                            const float* human_rgbsigma, const float* newsmpl_pts, const float* pts_mask,
                            const float* rays_o_bkg, const float* rays_d_bkg, const float* smpl_to_world,
                            const int32_t* tiny_d_flag, int B, int Sb, int Sh, float thre_fg,
                            float* rgb, int32_t* idx_fg, int32_t* total_order, float* human_weights_sorted,
                            float* z_human, hos_stream_t stream);
int hos_merge_composite_bwd(const float* g_rgb, const float* g_human_weights_sorted,
                            const float* bkg_tdist, const float* bkg_rgb, const float* bkg_density,
                            const float* human_rgbsigma, const float* newsmpl_pts, const float* pts_mask,
                            const float* rays_o_bkg, const float* rays_d_bkg, const float* smpl_to_world,
                            const int32_t* tiny_d_flag, int B, int Sb, int Sh, float thre_fg,
                            float* g_bkg_rgb, float* g_bkg_density, float* g_human_rgbsigma, float* g_pts_mask,
                            hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-frame prologue of the human-object branch, F frames per launch (the current
 * frame and, for the flow set, the previous one).  One workgroup per frame: latency-bound bookkeeping
 * on 26 joints that the reference spreads over ~120 torch launches per step.
 * ------------------------------------------------------------------------------------------ */
/* BodyPoseRefiner (pose_decoders/mlp_delta_body_pose.py:14-73; trunk 75->256->256->256, heads 256->256->75,
 * ReLU) + RodriguesModule (U:66-92, theta = sqrt(1e-5 + |r|^2)) + N:589-605: Rs_out[0] = Rs[0],
 * Rs_out[i] = Rs[i] dR[i-1], Ts_out[i] = Ts[i] + dT[i-1].
 * posevec [F,75], Rs [F,K,3,3], Ts [F,K,3]; K = 26, width = 256.
 * weights14 / grads14: HOST arrays of 14 device pointers in the order (weight, bias) of
 *   block_mlps.0, block_mlps.2, block_mlps.4, block_mlps_dstR.0, block_mlps_dstR.2, block_mlps_dstT.0,
 *   block_mlps_dstT.2  (weights row-major [out,in], contiguous).
 * saved: [F, hos_pose_refine_saved_floats()] activations kept for the backward pass.
 * The backward ACCUMULATES (+=) the parameter gradients into grads14 (every element owned by one thread, no atomics);
 * workspace: F * hos_pose_refine_workspace_floats() floats (the per-layer output gradients between its two launches). */
long long hos_pose_refine_saved_floats(void);
long long hos_pose_refine_workspace_floats(void);
int hos_pose_refine_fwd(const float* posevec, const float* Rs, const float* Ts, const float* const* weights14,
                        int F, int K, int width, float* Rs_out, float* Ts_out, float* saved, hos_stream_t stream);
int hos_pose_refine_bwd(const float* g_Rs_out, const float* g_Ts_out, const float* posevec, const float* Rs,
                        const float* saved, const float* const* weights14, float* const* grads14, int F, int K, int width,
                        float* workspace, hos_stream_t stream);

/* MotionBasisComputer.forward (U:134-174): G_dst = kinematic chain of [R_i|T_i] over the SMPL tree (U:100-103);
 * backward bases [R_bwd|T_bwd] = G_cnl G_dst^-1 (observation -> canonical, used by the backward LBS warp N:304-355),
 * forward bases [R_fwd|T_fwd] = G_dst G_cnl^-1 (N:357-399).  cnl_gtfms [K,4,4] row-major (shared by the frames).
 * The affine inverses are closed form (adjugate / determinant) -- the function torch.inverse computes by LU.
 * The backward takes the gradients w.r.t. the four outputs (any may be NULL) and writes g_dst_Rs / g_dst_Ts. */
int hos_motion_basis_fwd(const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int F, int K,
                         float* R_bwd, float* T_bwd, float* R_fwd, float* T_fwd, hos_stream_t stream);
int hos_motion_basis_bwd(const float* g_R_bwd, const float* g_T_bwd, const float* g_R_fwd, const float* g_T_fwd,
                         const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int F, int K,
                         float* g_dst_Rs, float* g_dst_Ts, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Forward of a whole non-rigid motion MLP in one launch, activations on chip across the layers (hos_chain.hip):
 * `NonRigidMotionMLP` / `NonRigidForwardMLP`, non_rigid_motion_mlps/mlp_offset.py:16-70 -- [cond 75 | hann 36] -> 5 x (128,
 * ReLU) with the hann features re-concatenated before Linear #4 -> 3, xyz = x + offset.  fp16 (hi, lo) x3 products like the split
 * GEMMs; every hidden activation is written once (fp32 [P, ldact], the backward pass reads them), none is read back.
 *   hos_mlp_chain_pack    weights7 / ldw7 / biases7: HOST arrays over the 7 linear layers (weights fp32 [128 (3), ld], nn.Linear
 *                         layout, zero-padded reduction: ld >= 128, the skip layer [h 128 | hann 36 | 0] with ld >= 192) ->
 *                         chain_planes (hos_mlp_chain_weight_bytes() bytes: per 32-wide output block and 16-wide reduction
 *                         step the MFMA A fragments, reduction index permuted to the accumulator layout) and aux
 *                         (hos_mlp_chain_aux_floats() floats: biases + the last layer).  Once per optimiser step.
 *   hos_mlp_chain128_fwd  E [P, lde >= 128] first-layer rows, PE [P, ldpe >= 64] hann features, x [P,3]; acts6: HOST array of the
 *                         6 output buffers; xyz [P,3].  rows_dev as for hos_linear_fwd.
 * Folded form (E == NULL): the condition code (mlp_offset.py:55 -- the pose vector, one per FRAME) is the same in every row of a
 * launch, so W0[:, :ncond] . cond enters as a bias and the first layer reduces over the hann features alone:
 *   hos_mlp_chain_pack_fold    as hos_mlp_chain_pack (layer 0: ld >= ncond + nfeat) + cond [ncond] on the device, nfeat <= 64 ->
 *                              planes / aux of the folded chain and w0h [128, 64] = fp32 copy of W0[:, ncond : ncond + nfeat]
 *                              (16-byte aligned rows: the W operand of the first layer's hos_linear_bwd_fused, X = PE)
 *   hos_mlp_chain128_fwd       with E == NULL: PE is read once per row and serves layer 0 and the skip layer
 *   hos_mlp_chain_unfold_grad  after the backward pass: gW0[:, ncond:] += gw0h, gW0[:, :ncond] += db (x) cond, gb0 += db, where
 *                              gw0h [128, 64] / db [128] are the (zeroed) gradient buffers the folded first layer accumulated into */
long long hos_mlp_chain_weight_bytes(void);
long long hos_mlp_chain_aux_floats(void);
int hos_mlp_chain_pack(const float* const* weights7, const int* ldw7, const float* const* biases7, void* chain_planes,
                       float* aux, hos_stream_t stream);
int hos_mlp_chain_pack_fold(const float* const* weights7, const int* ldw7, const float* const* biases7, const float* cond,
                            int ncond, int nfeat, void* chain_planes, float* aux, float* w0h, hos_stream_t stream);
int hos_mlp_chain_unfold_grad(const float* gw0h, const float* db, const float* cond, int ncond, int nfeat,
                              float* gW0, int ldw, float* gb0, hos_stream_t stream);
int hos_mlp_chain128_fwd(const float* E, int lde, const float* PE, int ldpe, const float* x, const void* chain_planes,
                         const float* aux, float* const* acts6, int ldact, float* xyz, int64_t P,
                         const int32_t* rows_dev, hos_stream_t stream);

/* Backward of a GROUP of consecutive thin layers of that MLP in one launch (hos_mlpbwd.hip, chain_bwd_kernel): the gradient with
 * respect to a layer's output stays in LDS between the layers of the group -- it replaces the chain of hos_linear_bwd_fused calls
 * (autograd of core/nets/human_nerf/non_rigid_motion_mlps/mlp_offset.py:54-70), each of which wrote dX to HBM for the next one.
 *   cfg 0: {offset head [3 (32) <- 128] -> chain, layer 5 [128 <- 128] -> HBM}
 *   cfg 1: {skip concat's hann columns [128 <- 64] -> HBM, layer 4 [128 <- 128] -> chain, layer 3 -> HBM}
 *   cfg 2: {layer 2 -> chain, layer 1 -> chain, folded layer 0 [128 <- 64] -> HBM};  cfg 3: the same with an unfolded layer 0 [128 <- 128]
 *   cfg 4, 5: the MLP as two groups of four steps (register-pressure experiment)
 * Per step s (HOST arrays of hos_mlp_chain_bwd_steps(cfg) entries): X[s] [M, ldx[s]] the layer's input rows (ReLU mask where the
 * step has one), images[s] the weight as packed by hos_mlp_chain_bwd_pack (n <= 8 images per launch: HOST arrays of (cfg, step, W + col0,
 * ldw, N, K, image)) into hos_mlp_chain_bwd_image_bytes(cfg, s) bytes (once per optimiser step), dXout[s] [M, lddx[s]] for "-> HBM" steps (NULL otherwise),
 * dW[s] [N[s], lddw[s]] +=, db[s] [N[s]] += (NULL: no bias gradient).  dZ [M, lddz] enters the first step; rows_dev as in
 * hos_mlp_chain128_fwd.  ws: hos_mlp_chain_bwd_ws_floats(cfg, M) floats of slab workspace; the slab reductions are deferrable like
 * those of hos_linear_bwd_fused (hos_mlp_bwd_defer / hos_mlp_bwd_flush). */
int hos_mlp_chain_bwd_steps(int cfg);
long long hos_mlp_chain_bwd_image_bytes(int cfg, int step);
long long hos_mlp_chain_bwd_ws_floats(int cfg, int M);
int hos_mlp_chain_bwd_pack(int n, const int* cfg, const int* step, const float* const* W, const int* ldw, const int* N, const int* K,
                           void* const* image, hos_stream_t stream);
int hos_mlp_chain_bwd(int cfg, const float* dZ, int lddz, int M, const int32_t* rows_dev, const float* const* X, const int* ldx,
                      const void* const* images, float* const* dXout, const int* lddx, float* const* dW, const int* lddw,
                      float* const* db, const int* N, const int* K, float* ws, int64_t ws_floats, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Cycle-consistency set (N:505-536): the sample points with fg_likelihood_mask > 0.005.
 * The reference selects them by boolean indexing (data-dependent shape = a host round trip per step);
 * this is an order-preserving stream compaction into fixed-capacity buffers, the count stays on the device.
 *   sel[j]   = j-th index i (ascending, = torch.nonzero order) with mask[i] > thr, -1 for j >= *count
 *   out_a[j] = src_a[sel[j]], out_b[j] = src_b[sel[j]]  ([P,3] each, rows >= *count zero); any of sel / out_a /
 *   out_b may be NULL.  workspace: hos_compact_workspace_ints() int32, zeroed once by the caller.  P <= 4 Mi. */
long long hos_compact_workspace_ints(void);
int hos_compact_rows(const float* mask, float thr, const float* src_a, const float* src_b, int64_t P, int32_t* count,
                     int32_t* sel, float* out_a, float* out_b, int32_t* workspace, hos_stream_t stream);
/* The gather's gradient: dst [P,3] = 0, then dst[sel[j]] = src[j] for j < *count. */
int hos_scatter_rows(const float* src, const int32_t* sel, const int32_t* count, int64_t P, float* dst, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training losses of the human-object stages, values and gradients on the
 * device, no host round trip, deterministic summation order.
 * Replaces `get_loss` M:1690-1716 (stage 3) and 2nd_State_Conditional_Human-Object/src/model/mipnerf360/
 * model.py:918-944 (stage 2) minus their LPIPS term: `img2mse` + `_unpack_imgs` (M:36, M:41-50), `flow_func`
 * M:1680-1688 with `img2mae` M:61-71, and the cycle term M:1707-1709.
 *   mse   = (sum_rays |rgb - target|^2 + mse_const) / mse_count     (mse_const: the patch pixels outside the
 *           ray mask, which `_unpack_imgs` fills with the background colour; 0 in stage 3)
 *   flow  = sum |uv(pts_prev) - ray_grid[:, :2] - ray_grid[:, 2:4]| * weights * M / (S * sum_b M_b + 1e-8) / 2,
 *           M_b = ray_grid[b,4] (* fg[b] if fg != NULL: the stage-3 row selection M:1704); pts_prev NULL -> 0
 *   cycle = mean over the first n rows of |observe - deform|^2 / 2, n = min(n_cyc, *n_cyc_dev) (n_cyc_dev may
 *           be NULL); n == 0 -> 0 (the reference's single-point fallback N:534-536 gives the same 0)
 * out8 = {total, mse, flow, cycle, 1/flow-denominator, 1/n, sum_b M_b, n}.
 * workspace: hos_train_losses_workspace_floats() floats, zeroed once by the caller (re-armed by the kernel). */
long long hos_train_losses_workspace_floats(void);
int hos_train_losses_fwd(const float* rgb, const float* target, long long n_rays, float mse_const, float mse_count,
                         const float* pts_prev, const float* weights, const float* ray_grid, const int32_t* fg,
                         const float* cam_prev, const float* intrinsics_prev, int S,
                         const float* observe, const float* deform, long long n_cyc, const int32_t* n_cyc_dev,
                         float w_mse, float w_flow, float w_cycle, float* workspace, float* out8,
                         hos_stream_t stream);
/* Gradients of `total` (scaled by *g_total, 1 if NULL) w.r.t. rgb [n_rays,3], pts_prev [n_rays,S,3],
 * weights [n_rays,S] and deform [n_cyc,3] (rows >= n are zeroed); any output may be NULL. */
int hos_train_losses_bwd(const float* g_total, const float* out8, const float* rgb, const float* target, long long n_rays,
                         float mse_count, const float* pts_prev, const float* weights, const float* ray_grid,
                         const int32_t* fg, const float* cam_prev, const float* intrinsics_prev, int S,
                         const float* observe, const float* deform, long long n_cyc, const int32_t* n_cyc_dev,
                         float w_mse, float w_flow, float w_cycle,
                         float* g_rgb, float* g_pts_prev, float* g_weights, float* g_deform, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser over flat buffers (torch.optim.Adam semantics, M1:536-539; PL norm clipping,
 * S1/run.py:155 gradient_clip_val).
 * ------------------------------------------------------------------------------------------ */
/* sumsq[0] += sum g^2  (caller zeroes sumsq). */
int hos_sumsq(const float* g, int64_t n, float* sumsq, hos_stream_t stream);
/* Adam step; if sumsq != NULL and max_norm > 0 the gradient is scaled by
 * min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) first; grad_scale multiplies g before that
 * (1/world_size for DDP averaging). */
int hos_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale, const float* sumsq,
                  float max_norm, hos_stream_t stream);

/* Same update with the per-step scalars read from device memory: hyper = {lr, 1-beta1^t, 1/sqrt(1-beta2^t)}.
 * Lets a whole training step be captured once in a hipGraph and replayed (the host only refreshes 12 bytes). */
int hos_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                      float beta1, float beta2, float eps, float grad_scale, const float* sumsq,
                      float max_norm, hos_stream_t stream);

/* ---- round 4: the remaining torch / library launches of a captured training step as library kernels ---------------------- */
/* Head of the motion-weight volume decoder: y [N] = LeakyReLU(W [N, ldw] . x [K] + b [N]) -- `block_mlp` of ConvDecoder3D on the
 * learned constant embedding (network_util.py:21-30, deconv_vol_decoder.py:36-37; was F.linear -> a library GEMM + leaky_relu). */
int hos_rowdot_lrelu_fwd(const float* x, const float* W, int ldw, const float* b, int N, int K, float slope, float* y,
                         hos_stream_t stream);
/* its backward: gW [N, ldgw] += d x^T, gb [N] += d (NULL: skip), gx [K] += W^T d (NULL: skip), d = g * (y > 0 ? 1 : slope). */
int hos_rowdot_lrelu_bwd(const float* g, const float* y, const float* x, const float* W, int ldw, int N, int K, float slope,
                         float* gW, int ldgw, float* gb, float* gx, hos_stream_t stream);
/* Tail of the decoder: vol [C, V3] = softmax over c of (z [V3, C] + log prior [C, V3]), C <= 32 (deconv_vol_decoder.py:38-42:
 * `F.softmax(decoded_weights + torch.log(motion_weights_priors), dim=1)`); backward gz [V3, C] = vol (g - sum_c g vol). */
int hos_volume_softmax_fwd(const float* z, const float* prior, int C, long long V3, float* vol, hos_stream_t stream);
int hos_volume_softmax_bwd(const float* g_vol, const float* vol, int C, long long V3, float* gz, hos_stream_t stream);
/* vol_cl [V3, 32] = the first Kb channels of vol [C, V3], channel-last and zero padded (the forward warp N:357-399 taps all bones at
 * one position); hos_volume_pair_bwd: g [C, V3] = g_vol (NULL: 0) + the channel-major scatter of g_cl [V3, 32] (NULL: 0). */
int hos_volume_channel_last(const float* vol, int Kb, long long V3, float* vol_cl, hos_stream_t stream);
int hos_volume_pair_bwd(const float* g_vol, const float* g_cl, int C, int Kb, long long V3, float* g, hos_stream_t stream);
/* n <= 8 buffers in one launch: dst[s][0 .. count[s]) = src[s] (src == NULL or src[s] == NULL: zeros).  Fills of accumulation
 * targets, stacks of per-frame tensors.  The tables are read during the call.  (plumbing; no reference counterpart) */
int hos_copy_or_zero_n(int n, float* const* dst, const float* const* src, const long long* count, hos_stream_t stream);
/* Diagnostics: buf[slot] = the constant 100 MHz device counter (s_memrealtime) when this point of the stream is reached; a kernel
 * node, so it can be captured inside a step's graph (scripts/diag_overlap.py: the un-profiled timeline of the two-stream step). */
int hos_debug_stamp(long long* buf, int slot, hos_stream_t stream);
/* Returns AND CLEARS the runtime's sticky last-error code (0 = none).  Every entry point reports a launch through hipGetLastError(), so
 * an error left behind by somebody else's failed call on this thread -- a hipGraph capture that was invalidated by an operation that
 * cannot be captured (a torch.distributed collective, an allocation) -- would be reported by the NEXT launch of this library although
 * that launch succeeded.  A caller that recovers from a failed capture calls this once before it goes on eagerly.  (plumbing) */
int hos_clear_last_error(void);
/* out[i] = sum_k src[k][i], n <= 8: the gradient of a tensor that feeds several consumers in one pass. */
int hos_add_n(int n, const float* const* src, long long count, float* out, hos_stream_t stream);
/* flag[0] = any |x[i]| < thr (M:1526, the tiny-direction test of the stage-3 re-projection, kept on the device). */
int hos_any_abs_below(const float* x, long long n, float thr, int32_t* flag, hos_stream_t stream);
/* gb [N] += db (NULL: skip); g_embed [E <= 64] += db [N] . W [N, ldw][:, c0 : c0 + E]: gradient of the per-call state embedding
 * through the first / skip layer of a MipNeRF360MLP (M:295-296). */
int hos_state_embed_grad(const float* db, const float* W, int ldw, int c0, int N, int E, float* gb, float* g_embed, hos_stream_t stream);
/* hos_embed_bwd added to a residual cotangent: g_x [P, 3] = res + d(features)/dx (rows past *rows_dev: res alone); g_x may be
 * uninitialised (mlp_offset.py:66-70, xyz = x + offset). */
int hos_embed_bwd_res(const float* x, const float* band_w, int num_freqs, int identity, const float* dA, int lda, int colA,
                      const float* dB, int ldb, int colB, int64_t P, const float* res, float* g_x, const int32_t* rows_dev,
                      hos_stream_t stream);
/* hos_head_grad that also writes the zero padding of its operand rows: columns (col_dd, ld_dd) of dz_density and [3, ld_dr) of dz_rgb. */
int hos_head_grad_padded(const float* g_density, const float* density, const float* g_rgb, const float* rgb, int P, float rgb_padding,
                         float* dz_density, int ld_dd, int col_dd, float* dz_rgb, int ld_dr, hos_stream_t stream);
/* Gradient norm + Adam of a whole training step in TWO launches, whatever the number of flat buffers / learning-rate ranges.
 * hos_sumsq_partials: partial[0 .. hos_sumsq_blocks()) = per-block sums of squares over n <= 32 spans (count % 4 == 0), fixed order,
 * nothing to zero first (`Trainer(gradient_clip_val=..., "norm")`, S1/run.py:155, 3rd_.../run.py:188-189).
 * hos_adam_multi: torch.optim.Adam (M1:536-569, optimizer.py:19-60) over n <= 32 spans; span s takes {lr, 1-beta1^t, 1/sqrt(1-beta2^t)}
 * from device memory hyper[s] (graph replay) or, if NULL, from lr[s] / step; clip coefficient min(max_norm / (sqrt(sum partial) *
 * |grad_scale| + 1e-6), 1) when partial != NULL; guard (the word of hos_set_range_flag, NULL: off): non-zero -> no parameter is
 * touched by THIS launch, skipped[0] is incremented and the word is cleared again (by the launch's last workgroup; skipped[1] is its
 * ticket scratch, both words zero-initialised by the caller) -- a forward whose activations left the exact fp16 hi/lo range never
 * reaches Adam, and the steps behind it are not lost.  With several ranks, MAX-reduce the word before this call. */
int hos_sumsq_blocks(void);
int hos_sumsq_partials(int n, const float* const* g, const long long* count, float* partial, hos_stream_t stream);
int hos_adam_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* count,
                   const float* const* hyper, const float* lr, int step, float beta1, float beta2, float eps, float grad_scale,
                   const float* partial, float max_norm, const unsigned int* guard, unsigned int* skipped, hos_stream_t stream);
/* Lazily updated spans (round 6).  The reference's optimiser is torch.optim.Adam under Lightning, whose zero_grad() sets gradients to
 * None: a parameter that took no part in a step -- the state embeddings of the states the step's frame is not in (M:224-296,
 * N:179-246), the pose decoder before its kick-in iteration (N:589-605) -- is SKIPPED (no moment decay, no movement) and its bias
 * corrections count ITS OWN updates.  hos_adam_lazy_prepare (one launch, n <= 32 spans, after the gradient exchange): state[s]
 * {t, active, 1-beta1^t, 1/sqrt(1-beta2^t), 2 scratch words, 2 unused} (8 floats, zero-initialised once by the caller; spans
 * float4-aligned with count % 4 == 0) -- active = the span's gradient is not
 * identically zero (and the range-guard word, if given, is clear); then t += 1.  hos_adam_multi_lazy = hos_adam_multi with lazy[s]
 * (NULL: a plain span) = that row: an inactive span is not touched, an active one uses its own corrections.  lazy_row[s] > 0 (NULL
 * table / 0: one row for the span): the span is consecutive ROWS of that many floats with consecutive 8-float state rows at lazy[s]
 * (the [n_states, 64] block of state embeddings as ONE span, whatever the number of states). */
int hos_adam_lazy_prepare(int n, const float* const* g, const long long* count, float* const* state, float beta1, float beta2,
                          const unsigned int* guard, hos_stream_t stream);
int hos_adam_multi_lazy(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* count,
                        const float* const* hyper, const float* const* lazy, const int* lazy_row, const float* lr, int step, float beta1, float beta2,
                        float eps, float grad_scale, const float* partial, float max_norm, const unsigned int* guard,
                        unsigned int* skipped, hos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LPIPS term of the stage-2 / stage-3 training loss (hos_lpips.hip): `LPIPS(net='vgg')` of third_parties/lpips/lpips.py:22-122 as
 * src/model/mipnerf360/model.py:1664-1678 calls it on the unpacked P x P patches (weight 1.0 in configs/default.yaml:97-101), VGG-16
 * frozen, eval mode.  The 3 x 3 convolutions are im2col + hos_linear_fwd (bias + ReLU epilogue) and, backward, hos_linear_dgrad +
 * col2im (no weight gradient exists); everything is channel-last [image, y, x, channel]:
 *   hos_lpips_prep        out = (2 x - 1 - shift_c) / scale_c for n_pixels RGB pixels in [0, 1]        (lpips.py:124-131, model.py:1661)
 *   hos_im2col3x3         col [NI*H*W, ld >= 9 C] <- in [NI, H, W, C], kernel 3, padding 1, column tap * C + c, padding columns zero
 *   hos_col2im3x3         dx [NI, H, W, C] <- dcol, times [relu_src > 0] (NULL: no mask): the input gradient of that convolution
 *   hos_maxpool2x2_fwd/bwd  MaxPool2d(2, 2); bwd routes to the first maximum of a window (torch's rule) and applies [in > 0]
 *   hos_lpips_head_fwd    part (hos_lpips_part_floats(Np) floats, zeroed by the caller) += coef * sum_pixels sum_c w_c (f0_c / R0 -
 *                         f1_c / R1)^2 per pair i (prediction i = image i, target i = image Np + i of feats [2 Np * HW, C]) and pixel
 *                         chunk; R = sqrt(sum f^2 + 1e-10) + 1e-10   (lpips.py:92-100)
 *   hos_lpips_head_bwd    g_feats [Np * HW, C] (+)= gscale[0] * coef * d/d f0, times [f0 > 0]
 *   hos_lpips_finish      out[0] = sum of part, in index order
 *   hos_bias_relu         y = relu(y + bias) in place: the epilogue behind hos_linear_fwd_splitk for the deep, few-pixel convolutions
 *   hos_unpack_patches_fwd/bwd  model.py:41-50 `_unpack_imgs`: img[p] = idx[p] >= 0 ? rgb[idx[p]] : bgcolor * bg_scale;  backward
 *                         g_rgb[idx[p]] = g_img[p] * (s0, s1, s2) per channel (the caller zeroes g_rgb) */
int hos_lpips_prep(const float* x01, int64_t n_pixels, float* out, hos_stream_t stream);
int hos_im2col3x3(const float* in, int NI, int H, int W, int C, float* col, int ld, hos_stream_t stream);
int hos_col2im3x3(const float* dcol, int ld, int NI, int H, int W, int C, const float* relu_src, float* dx, hos_stream_t stream);
int hos_maxpool2x2_fwd(const float* in, int NI, int H, int W, int C, float* out, hos_stream_t stream);
int hos_maxpool2x2_bwd(const float* g_out, const float* in, int NI, int H, int W, int C, float* g_in, hos_stream_t stream);
int hos_lpips_head_fwd(const float* feats, const float* lin_w, int Np, int HW, int C, float coef, float* part, hos_stream_t stream);
int hos_lpips_head_bwd(const float* feats, const float* lin_w, int Np, int HW, int C, float coef, const float* gscale,
                       int accumulate, float* g_feats, hos_stream_t stream);
int hos_lpips_finish(const float* part, int Np, float* out, hos_stream_t stream);
int hos_lpips_part_floats(int Np);
int hos_bias_relu(float* y, const float* bias, int64_t M, int N, hos_stream_t stream);
int hos_unpack_patches_fwd(const float* rgb, const int32_t* idx, const float* bgcolor, float bg_scale, int64_t n_pixels, float* img,
                           hos_stream_t stream);
int hos_unpack_patches_bwd(const float* g_img, const int32_t* idx, int64_t n_pixels, float s0, float s1, float s2, float* g_rgb,
                           hos_stream_t stream);

/* Tail of the stage-1 loss (1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-514) in one launch each way:
 *   out4 = [m_data sqrt(mse + pad^2) + m_inter (sum inter0 + sum inter1) / (B Sc) + m_dist mean(dist), mse, interlevel, distortion]
 * from rgb / target [B,3] and the per-ray terms of hos_interlevel_fwd (inter0 / inter1 [B], either may be NULL) and hos_distortion_fwd
 * (dist [B]); the backward writes g_rgb [B,3] and the (constant) gradients of the per-ray terms, scaled by gout[0]. */
int hos_stage1_loss_fwd(const float* rgb, const float* target, int B, const float* inter0, const float* inter1, int Sc, const float* dist,
                        float m_data, float m_inter, float m_dist, float charb_padding, float* out4, hos_stream_t stream);
int hos_stage1_loss_bwd(const float* rgb, const float* target, int B, int Sc, const float* fwd_out4, const float* gout, float m_data,
                        float m_inter, float m_dist, float charb_padding, float* g_rgb, float* g_inter, float* g_dist, hos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOSRENDER_H */
