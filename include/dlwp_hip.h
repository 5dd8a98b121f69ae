/*
 * dlwp_hip.h -- C ABI of libdlwp_hip.so: the MI355X (gfx950) forecast-step hot path of DLWP.
 *
 * The reference (jweyn/DLWP) has NO native code and NO FFI: its drop-in boundary is a Python name registry
 * (DLWP/model/models.py:97-103 -> DLWP/util.py:82-93) on top of third-party Keras/TensorFlow ops.  Every entry point
 * below therefore replaces the Keras/TF op(s) that a reference layer resolves to; the reference call site is cited
 * per function.  The Python host layer (dlwp_amd/) binds these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns int: 0 ok, DLWP_EINVAL (-1) bad argument, DLWP_EUNSUPPORTED (-2), DLWP_EHIP (-3) HIP
 *     runtime error, DLWP_ERCCL (-4) RCCL error; dlwp_last_error() returns a thread-local message for the last non-zero return.
 *   - all tensor pointers are CALLER-OWNED DEVICE memory (e.g. torch-ROCm storage), dense, row-major, NCHW unless
 *     stated; the library never allocates, frees or retains them (rollout objects excepted: buffers captured in a
 *     rollout graph must outlive it).
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it.  No host synchronisation inside.
 *     The Winograd convolutions keep their transformed filters in ONE scratch buffer of the handle (32 MB, allocated at the
 *     first use): launches that share a handle must be ordered on one stream at a time (use one handle per stream
 *     otherwise, or dlwp_conv2d_prepare into caller memory).  A rollout graph prepares into caller memory as well
 *     (dlwp_rollout_workspace_bytes).  Kernel-selection switches live in the handle (dlwp_set_option).
 *   - dtype: the STORAGE type of activation tensors.  DLWP_F32 everywhere; the forward convolutions and
 *     dlwp_maxpool2_fwd also take DLWP_BF16 and, for the convolutions, DLWP_DTYPE_IO(in, out) with different input and
 *     output storage (config 4: bf16 activations between the layers, fp32 state at the model boundary).  Weights, biases
 *     and all arithmetic (MFMA accumulate, activation) are fp32; fp32 -> bf16 rounds to nearest even.
 */
#ifndef DLWP_HIP_H
#define DLWP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLWP_OK            0
#define DLWP_EINVAL       (-1)
#define DLWP_EUNSUPPORTED (-2)
#define DLWP_EHIP         (-3)
#define DLWP_ERCCL        (-4)

#define DLWP_F32  0
#define DLWP_BF16 1
#define DLWP_BF16_O8 2   /* bfloat16 in channel OCTETS: (N, C/8, H, W, 8) -- a pixel's 8 consecutive channels are 16 contiguous
                          * bytes, the unit the bf16 matrix-core kernels stage and multiply.  Only inside DLWP_DTYPE_IO(in, out)
                          * of dlwp_conv2d_fwd / _prepared / dlwp_convlstm_conv_fwd, for layers dlwp_conv2d_supports_dtype
                          * accepts; channel counts and channel windows must be whole octets.  With an O8 OUTPUT,
                          * dlwp_convlstm_conv_fwd reads z_add in octets and keeps the float32 cell state (c_prev, c_out) as
                          * (N, F/8, H, W, 8) float32.  NCHW stays the layout at the model boundary.                        */
#define DLWP_DTYPE_IO(in, out) (0x10000 | (in) | ((out) << 8))   /* input / output storage of one launch */
#define DLWP_DTYPE_IN(d)  (((d) & 0x10000) ? ((d) & 0xff) : (d))
#define DLWP_DTYPE_OUT(d) (((d) & 0x10000) ? (((d) >> 8) & 0xff) : (d))
#define DLWP_COMPUTE_BF16 0x20000   /* OR-ed into a convolution's dtype: a float32-stored INPUT may be rounded to bfloat16
                                     * so that the layer runs on the bf16 matrix cores (config 4's first layers)        */

/* per-axis halo modes */
#define DLWP_PAD_ZERO 0   /* keras.layers.ZeroPadding2D                                  (examples/train.py:163)   */
#define DLWP_PAD_WRAP 1   /* DLWP.custom.PeriodicPadding2D                               (DLWP/custom.py:139-214)  */
#define DLWP_PAD_EDGE 2   /* DLWP.custom.FillPadding2D (pole rows)                       (DLWP/custom.py:309-402)  */
#define DLWP_PAD_REFLECT   3   /* DLWP.custom.TFPadding2D(mode='REFLECT'):   tf.pad mirror WITHOUT the border element  */
#define DLWP_PAD_SYMMETRIC 4   /* DLWP.custom.TFPadding2D(mode='SYMMETRIC'): tf.pad mirror WITH it (custom.py:527-600) */

#define DLWP_ACT_LINEAR 0
#define DLWP_ACT_TANH   1
#define DLWP_ACT_RELU   2

/* how the conv loader reads its stored input tensor */
#define DLWP_SRC_DIRECT    0
#define DLWP_SRC_UPSAMPLE2 1   /* fused keras UpSampling2D(2) in front of the conv       (examples/train.py:191,201) */
#define DLWP_SRC_MAXPOOL2  2   /* fused keras MaxPooling2D(2) in front of the conv       (examples/train.py:171,181) */

typedef struct dlwp_handle* dlwp_handle_t;

typedef struct { int n, c, h, w; } dlwp_shape4;

typedef struct {
  int top, bottom, left, right;   /* asymmetric amounts                                                    */
  int mode_h, mode_w;             /* DLWP_PAD_* per axis; corners follow from applying the two axes in turn */
} dlwp_pad2d;

typedef struct {
  int cout, kh, kw, dil_h, dil_w; /* stride 1, 'valid' after the implicit halo                              */
  dlwp_pad2d halo;                /* fused Periodic/Zero/Fill padding in front of the conv; all-zero = none */
  int act;                        /* DLWP_ACT_* applied in the epilogue after the bias                      */
  int in_c_off, in_c_total;       /* read channels [off, off+xs.c) of a buffer that has in_c_total channels (slice_layer) */
  int out_c_off, out_c_total;     /* write channels [off, off+cout) of a buffer with out_c_total channels (concatenate)   */
  int src_mode;                   /* DLWP_SRC_*: xs describes the STORED tensor; the conv sees it transformed */
  int out_pool;                   /* 1: the epilogue also applies MaxPooling2D(2) -- y is (n, cout, ho/2, wo/2) (forward /
                                   * inference only; ask dlwp_conv2d_supports_out_pool first).  2: 2x2 SUM instead of the
                                   * max (linear activation, no bias; DLWP_EUNSUPPORTED when no kernel instance has it):
                                   * the adjoint of UpSampling2D(2), see dlwp_conv2d_bwd_data_stored                   */
  int out_d2s;                    /* 1: the cout = 4 F channels are the 2x2 PHASES of F fields, phase-major (channel
                                   * (2a + b) F + f = field f at rows 2i + a, columns 2j + b): the epilogue stores them
                                   * interleaved -- y is (n, out_c_total, 2 ho, 2 wo), channels [out_c_off, out_c_off + F) --
                                   * which is dlwp_depth_to_space2 without the pass (inference; ask
                                   * dlwp_conv2d_supports_out_d2s first; not with out_pool)                             */
  int lstm_f, lstm_rec_act;       /* lstm_f = F > 0: this convolution completes the gate pre-activations of a ConvLSTM2D step
                                   * (cout = 4 F, gates i f c o) and its epilogue applies the cell update of
                                   * dlwp_convlstm_gates -- see dlwp_convlstm_conv_fwd, the only entry that takes such a
                                   * descriptor; act = the cell activation, lstm_rec_act = 0 hard_sigmoid | 1 sigmoid; the
                                   * output window (out_c_off, out_c_total) is that of h (F channels).  0: plain Conv2D     */
} dlwp_conv2d;

/* ---- library ---------------------------------------------------------------------------------------------------- */
int         dlwp_version(void);
const char* dlwp_last_error(void);
/* Diagnostics: `text` (NULL: nothing) is written to stdout, followed by a newline, if the process dies of SIGABRT / SIGSEGV / SIGBUS -- the
 * HSA runtime abort()s a process on a GPU memory fault.  bench.py parks its result line here before it runs the one-shot exchange
 * between real GPUs for the first time, and clears it afterwards.  At most 512 KB; earlier handlers are chained.                 */
int         dlwp_set_crash_message(const char* text);
/* One idle wave that occupies `stream`'s hardware queue for `microseconds` (at most 100 000): the probe with which the host layer
 * finds out whether two streams really run beside each other (the runtime multiplexes streams onto a few hardware queues).       */
int         dlwp_spin(dlwp_handle_t h, int microseconds, void* stream);
int         dlwp_create(dlwp_handle_t* h, int device);   /* one handle per device; thread-compatible */
int         dlwp_destroy(dlwp_handle_t h);
int         dlwp_device_info(dlwp_handle_t h, int* cu_count, int* lds_bytes, char* arch, size_t arch_len);
/* Per-handle switches: nothing about kernel selection is process-global, so handles (threads, streams) never see each
 * other's settings.  A new handle starts from the environment (DLWP_WINOGRAD=0, DLWP_BF16_MFMA=0 switch the families off).
 * `previous` (nullable) receives the old value.                                                                       */
#define DLWP_OPT_WINOGRAD           0  /* 3x3 layers with whole channel tiles: Winograd F(2x2,3x3) (1, default) or the direct
                                        * implicit GEMM (0)                                                                */
#define DLWP_OPT_BF16_MFMA          1  /* layers whose INPUT is stored as bf16 (even width, >= 12 channels, no pooling fused
                                        * in): multiply on the bf16 matrix cores with bf16-rounded weights (1, default)     */
#define DLWP_OPT_FORCE_CONV_CONFIG  2  /* tuning sweeps / tests: run this forward tile configuration (-1 = heuristic)      */
#define DLWP_OPT_FORCE_WGRAD_CONFIG 3  /* ... this weight-gradient configuration                                           */
#define DLWP_OPT_WINO_PAIRS         4  /* Winograd on narrow maps (22x45): two samples side by side in one virtual row (1,
                                        * default) or the wide + narrow launch pair (0); same bits either way              */
#define DLWP_OPT_WGRAD_FILL         5  /* weight gradient: how many waves per CU the number of partial-sum splits aims at, in
                                        * eighths of the CU's 16 wave slots (default 4 = half a complement; r2 used 16: measured on
                                        * the six weight gradients of the config-2 U-Net, 8 / 64 samples: 0.306 / 1.291 ms at 16,
                                        * 0.251 / 1.172 ms at 4, 0.263 / 1.820 ms at 2, tools/bench_reduce_jobs.py).  Every
                                        * split writes a slab the size of the weight tensor: fewer splits = less slab traffic    */
#define DLWP_OPT_FEW_STREAM         6  /* 3x3 layers of at most 4 input channels under a pooling epilogue (the first layer):
                                        * the streaming kernel that keeps the weights in registers and walks over the samples
                                        * (csrc/conv_fwd_few.hip) -- 1 (default): from 8 tiles per workgroup on, 0: never,
                                        * 2: whenever the layer qualifies.  Same bits as the direct family's instance.  r4: the
                                        * same switch selects the streaming form of the 16-output-channel Winograd blocks
                                        * (csrc/conv_fwd_wino2s.hip: at most 32 input channels, filters resident in LDS; the
                                        * restated output layer) -- the bits of the position-split instance it stands in for   */
#define DLWP_OPT_SPLITK             7  /* Winograd layers on SMALL grids (a launch under one round of resident workgroups: an
                                        * ensemble share of 1 ... 8 members, 8 training samples per rank): the input channels are
                                        * divided over several workgroups per output tile; the last one to arrive sums the partial
                                        * tiles in index order (deterministic), adds the bias, activates, pools and stores
                                        * (csrc/conv_fwd_k3d1s.hip).  0 (default since r5): never -- a sample's bits do not depend on
                                        * its batch size at all (a member rolls out to the same bits alone, in a
                                        * batch, on 1 GPU or as a shard of 8); the one operating point that gains is 1 ... 2
                                        * members of the 88 x 180 grid (+6 %).  1 (DLWP_SPLITK=1 in the environment):
                                        * by the rule of dlwp_conv2d_split_count; k >= 2: k
                                        * workgroups per tile wherever the layer is eligible (tuning sweeps, tests).  A split
                                        * launch differs from the unsplit one by float32 round-off (another association of the
                                        * same sum); two launches with the SAME split count give the same bits               */
#define DLWP_OPT_WINO_XLOADER       8  /* how the Winograd forward kernel of the 8 x 32 tile fetches its input tile -- bit 0: as
                                        * image-aligned column PAIRS (one 8-byte load per two elements) on float32 plain sources of
                                        * even width under zero / periodic column halos; bit 1: an UP-SAMPLED float32 source at
                                        * source resolution (one load per source element, written to the 2 x 2 tile slots it
                                        * replicates into; zero / periodic / edge halos); bit 2: EDGE PAIRS -- on a map whose
                                        * last 8-row tile is at most half used (44 rows) the blocks of that tile row take the
                                        * valid rows of two neighbouring column tiles each: one block in eighteen less.  7
                                        * (default): all; 0: as before r5 (DLWP_WINO_XLOADER in the environment).  The same
                                        * values reach the same patch positions: identical bits whatever the setting            */
int         dlwp_set_option(dlwp_handle_t h, int option, int value, int* previous);
/* the defaults themselves: what handles created AFTERWARDS start from, and what the handle-less host logic (planner hints
 * called with a NULL handle, e.g. on a machine without a GPU) uses.  Existing handles are not touched.                 */
int         dlwp_set_default_option(int option, int value, int* previous);

/* ---- halo padding: DLWP.custom.PeriodicPadding2D.call (custom.py:191-214), FillPadding2D.call (custom.py:359-402),
 *      keras ZeroPadding2D.  Generic [outer, H, W, inner] view: NCHW -> outer=N*C, inner=1; NHWC -> outer=N, inner=C.
 *      fwd: y[outer, H+t+b, W+l+r, inner];  bwd: dx = adjoint (halo folded back: wrap adds to the periodic image,
 *      edge adds to the border, zero drops).                                                                          */
int dlwp_pad2d_fwd(dlwp_handle_t, const void* x, void* y, int outer, int h, int w, int inner, dlwp_pad2d p,
                   int dtype, void* stream);
int dlwp_pad2d_bwd(dlwp_handle_t, const void* dy, void* dx, int outer, int h, int w, int inner, dlwp_pad2d p,
                   int dtype, void* stream);

/* ---- keras Conv2D(filters, k, dilation_rate, padding='valid', activation, data_format='channels_first')
 *      (examples/train.py:164-169 ... 214-219) with the halo, UpSampling2D/MaxPooling2D-in-front, bias, activation,
 *      slice_layer (custom.py:675-692) and concatenate fused.  Weights: Keras HWIO (kh,kw,cin,cout); bias (cout).
 *      xs = stored input (n, cin, h, w).  Output (n, cout, ho, wo) with
 *        hin = h (direct) | 2h (upsample2) | h/2 (maxpool2);  ho = hin + top + bottom - dil_h*(kh-1);  same for w.   */
int dlwp_conv2d_out_shape(dlwp_shape4 xs, const dlwp_conv2d* cd, dlwp_shape4* ys);
int dlwp_conv2d_fwd(dlwp_handle_t, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                    const dlwp_conv2d* cd, int dtype, void* stream);
/* y AND its MaxPooling2D(2) image y_pool (n, out_c_total, ho/2, wo/2) from one launch: the training forward of a Conv2D under a
 * pooling layer (examples/train.py:164-181: the backward pass needs y, the next layer the pooled tensor).  float32, no other
 * epilogue option in cd; prepared: as dlwp_conv2d_fwd_prepared (built for the same xs / cd with out_pool = 1 -- the instance is
 * chosen as for the pooling epilogue), or NULL.  DLWP_EUNSUPPORTED where the layer's kernel has no such epilogue:
 * dlwp_conv2d_fwd + dlwp_maxpool2_fwd.                                                                                       */
int dlwp_conv2d_fwd_pool2(dlwp_handle_t, const void* x, const void* w, const void* prepared, const void* bias, void* y,
                          void* y_pool, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, void* stream);
/* Weight-stationary use (model.predict over many batches, the rollout of DLWP/model/models.py:263-310): some kernel
 * families read the weights in a prepared layout (Winograd G g G^T, packed-N expansion, bf16 arrangement), which
 * dlwp_conv2d_fwd builds in the handle's scratch before every launch (~5 us).  dlwp_conv2d_prepare builds it once into
 * caller-owned device memory of dlwp_conv2d_prepared_bytes() bytes (0: the layer's kernel reads HWIO directly; then
 * `prepared` may be NULL); it is valid for exactly this (xs incl. xs.n, cd, dtype) and until the weights change.        */
size_t dlwp_conv2d_prepared_bytes(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
int dlwp_conv2d_prepare(dlwp_handle_t, const void* w, void* prepared, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype,
                        void* stream);
int dlwp_conv2d_fwd_prepared(dlwp_handle_t, const void* x, const void* w, const void* prepared, const void* bias, void* y,
                             dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, void* stream);
/* same contract, one thread per output element on the vector ALU: any kernel size; used as the in-library cross-check */
int dlwp_conv2d_fwd_direct(dlwp_handle_t, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int dtype, void* stream);

/* tuning hooks (tools/tune_conv.py, tests): enumerate the compiled MFMA tile configurations, ask which one the heuristic
 * picks (-1 = direct kernel); DLWP_OPT_FORCE_CONV_CONFIG forces one.  info9 = {ks, dil, th, tw, waves,
 * frags_per_wave (0: Winograd instance), cout_frags (< 0: packed-N instance for cout <= 16/-cout_frags), channel_chunk,
 * pooled_loader (2: bf16-MFMA instance, only for inputs stored as bf16)}.
 * Not part of the drop-in surface.                                                                                  */
int dlwp_conv2d_num_configs(void);
int dlwp_conv2d_config_info(int i, int* info9, int* lds_bytes);
int dlwp_conv2d_config_flags(int i);        /* bit 0: Winograd instance whose 16-position case splits the positions over two
                                              * waves per tile fragment (conv_fwd_wino2_kernel.h: 2 x waves x 64 threads);
                                              * bit 1: bf16 instance with the ConvLSTM2D cell update in its epilogue
                                              * (dlwp_conv2d.lstm_f; takes only such layers);
                                              * bit 2: position-split instance evaluated in the 32-channel kernel's order of
                                              * operations (same bits as that kernel): for layers with whole 32-channel tiles on
                                              * small grids, where the plain split instances (bit 0 alone) are not offered;
                                              * bit 3 / bit 4: bf16 instance whose input / output is stored in channel octets
                                              * (DLWP_BF16_O8) -- it takes launches with exactly that storage;
                                              * bit 5: Winograd instance with a compiled split-K variant (DLWP_OPT_SPLITK)   */
/* Host logic (no device work; handle nullable = default options): 1 when dlwp_conv2d_fwd(xs, cd, dtype) multiplies with
 * bf16-rounded weights (the bf16-MFMA family), else 0: what a caller comparing against an fp32-weight computation needs to
 * know. */
int dlwp_conv2d_uses_bf16_weights(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
/* Planner hint (host logic, handle nullable): 1 when a convolution of this geometry behind a MaxPooling2D runs faster with
 * the pooled tensor materialised by dlwp_maxpool2_fwd (the Winograd family has no pooled loader) than with the pooling
 * fused into the direct kernel's loader; 0 otherwise. */
int dlwp_conv2d_prefers_unfused_pool(dlwp_handle_t, int cin, int cout, int kh, int kw, int dil_h, int dil_w);
/* Planner hint (host logic, handle nullable): 1 when a compiled kernel can apply a following MaxPooling2D(2) in the epilogue
 * of this convolution (cd->out_pool = 1): the pre-pooling tensor is then never written. */
int dlwp_conv2d_supports_out_pool(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd);
/* 1 when a compiled matrix-core instance runs this layer with the storage codes of `dtype` (a DLWP_DTYPE_IO value; the question
 * matters for DLWP_BF16_O8, which only the bf16 matrix-core family reads and writes).  Host logic; h may be NULL.            */
int dlwp_conv2d_supports_dtype(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
/* ... and 1 when a compiled kernel can store the 2x2 phase channels of this convolution interleaved (cd->out_d2s = 1). */
int dlwp_conv2d_supports_out_d2s(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd);
int dlwp_conv2d_pick_config(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd);
/* Measurement hook (bench.py's roofline): what dlwp_conv2d_fwd(xs, cd, dtype) launches -- one kernel, or two when a Winograd
 * layer hands its ragged last column tile to a narrower instance -- and the matrix-core work each launch EXECUTES: the
 * padded GEMM volume its MFMA instructions multiply, 2 FLOP per multiply-add, tile / channel padding included and with
 * Winograd's 16 (9) multiplies per 2x2 outputs instead of the direct 36.  For the fp32 families this is exactly
 * SQ_INSTS_MFMA x 2048 (v_mfma_f32_16x16x4_f32) of a rocprofv3 --pmc pass.  config: index for dlwp_conv2d_config_info
 * (-1: the one-thread-per-output vector kernel, no matrix work; -2 / -3: the streaming kernels of DLWP_OPT_FEW_STREAM -- few
 * input channels under a pooling epilogue / 16-output-channel Winograd blocks --, whose grid is a number of persistent
 * workgroups, not of tiles).  out2 must hold 2 entries.                                                                   */
typedef struct {
  int config, grid, block_threads;
  double matrix_flops;
  int bf16_matrix;            /* 1: v_mfma_f32_16x16x32_bf16 (peak 2.5 PFLOP/s), 0: fp32 matrix cores (157.3 TFLOP/s) */
  int x_loader;               /* Winograd 8 x 32 instances (DLWP_OPT_WINO_XLOADER): 0 element by element, 1 column pairs, 2 source resolution;
                               * + 4: an edge-pair launch (the last tile row's blocks take two column tiles each)                       */
} dlwp_launch_info;
int dlwp_conv2d_launch_info(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype, dlwp_launch_info* out2,
                            int* n_launches);
/* The split regime of dlwp_conv2d_fwd(xs, cd, dtype) on this handle (DLWP_OPT_SPLITK): how many workgroups share the input
 * channels of one output tile -- 1: unsplit.  Two launches of a layer give the same bits for a sample exactly when their split
 * counts are equal (the count follows from the layer and xs.n: every batch size from the chip's first full round of workgroups
 * on is unsplit); across counts the results differ by float32 round-off.                                                     */
int dlwp_conv2d_split_count(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);

/* ---- backward of the fused Conv2D: the two halves of the Keras train step behind DLWPNeuralNet.fit / fit_generator
 *      (DLWP/model/models.py:188-228).  dz = dL/d(pre-activation), (n, out_c_total, ho, wo) window [out_c_off,+cout).
 *      bwd_data : dx = dL/d(input after the src transform, before the halo), (n, cin, hin, win); for DLWP_SRC_DIRECT it
 *                 may be the channel window [in_c_off,+cin) of an in_c_total-channel buffer; for the pooled / up-sampled
 *                 loaders the caller finishes with dlwp_maxpool2_bwd / dlwp_upsample2_bwd.
 *      bwd_weight: dw (kh,kw,cin,cout) HWIO, deterministic (fixed-order slab reduction); accumulate != 0 adds to dw.
 *      Workspace: caller-owned device memory of at least dlwp_conv2d_bwd_workspace(pass 0 = data, 1 = weight) bytes.   */
int dlwp_conv2d_bwd_workspace(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int pass, size_t* bytes);
int dlwp_conv2d_bwd_data(dlwp_handle_t, const void* dz, const void* w, void* dx, dlwp_shape4 xs, const dlwp_conv2d* cd,
                         int dtype, void* ws, size_t ws_bytes, void* stream);
/* DLWP_SRC_UPSAMPLE2 layers only: dx = dL/d(STORED tensor), (n, cin, xs.h, xs.w) -- bwd_data followed by
 * dlwp_upsample2_bwd in one kernel (the dense gradient is never written).  DLWP_EUNSUPPORTED when the layer has no kernel
 * instance with the summing epilogue; the caller then takes the two-call route.                                        */
int dlwp_conv2d_bwd_data_stored(dlwp_handle_t, const void* dz, const void* w, void* dx, dlwp_shape4 xs,
                                const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, void* stream);
int dlwp_conv2d_bwd_weight(dlwp_handle_t, const void* x, const void* dz, void* dw, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int accumulate, int dtype, void* ws, size_t ws_bytes, void* stream);
/* dlwp_conv2d_bwd_data where the layer's input x is the ACTIVATION OUTPUT of the layer in front (Conv2D -> Conv2D, the decoder of
 * examples/train.py:191-219): dx <- (data gradient) * act'(x) and db_in (xs.c floats, nullable) <- the per-channel sums of that
 * product -- the front layer's dlwp_act_bwd_bias_grad -- from the data gradient's own store phase.  prepared: as
 * dlwp_conv2d_bwd_data_prepared, or NULL (then w is flipped here).  act_in: DLWP_ACT_TANH / RELU.  Workspace:
 * dlwp_conv2d_bwd_workspace(pass = 3).  DLWP_EUNSUPPORTED where the gradient's convolution does not run on the instance with that
 * store phase (8 x 32 Winograd tiles, whole 32-channel tiles of xs.c, plain source, 'same' halo): keep the two calls.            */
int dlwp_conv2d_bwd_data_act(dlwp_handle_t, const void* dz, const void* w, const void* prepared, void* dx, dlwp_shape4 xs,
                             const dlwp_conv2d* cd, const void* x, int act_in, void* db_in, int dtype, void* ws, size_t ws_bytes,
                             void* stream);
/* The weight AND bias gradient of a layer whose only reader is MaxPooling2D(2) and whose data gradient nobody needs (the first
 * layer of the reference's networks, examples/train.py:159-170), from the layer's output y (laid out like dz above) and the
 * POOLED tensor's gradient dpool (n, cout, Ho/2, Wo/2): what dlwp_pool_act_bwd_bias_grad + dlwp_conv2d_bwd_weight compute, without
 * the gradient tensor in between (it is formed in the weight-gradient kernel's loader; ties as dlwp_maxpool2_bwd).  act:
 * DLWP_ACT_LINEAR / TANH / RELU; db nullable.  Workspace: dlwp_conv2d_bwd_workspace(pass = 2).  DLWP_EUNSUPPORTED where no
 * streaming instance fits (3x3, at most 4 input channels): the caller keeps the two calls.                                   */
int dlwp_conv2d_bwd_weight_pooled(dlwp_handle_t, const void* x, const void* y, const void* dpool, void* dw, void* db,
                                  dlwp_shape4 xs, const dlwp_conv2d* cd, int act, int accumulate, int dtype, void* ws,
                                  size_t ws_bytes, void* stream);
int dlwp_conv2d_wgrad_num_configs(void);                                   /* tuning hooks, as for the forward */
int dlwp_conv2d_wgrad_config_info(int i, int* info6, int* lds_bytes);      /* {ks, dil, th, tw, cout_frags (< 0: packed-N
                                                                             * instance for cout <= -cout_frags), waves} */
int dlwp_conv2d_wgrad_config_form(int i, int* cin_block, int* form);      /* input channels per workgroup; form 0 direct, 1
                                                                             * Winograd, 3 channel-block Winograd, 4 the
                                                                             * streaming form for <= 4 input channels */
int dlwp_conv2d_wgrad_pick_config(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd);   /* the heuristic's choice, -1: none */

/* ---- DLWP.custom.RowConnected2D.call / row_conv2d (DLWP/custom.py:825-837, 840-896): a Conv2D whose filters are shared
 *      along a row only -- output row r is the 'valid' convolution of input rows [r, r + kh) with its own kernel w[r]
 *      (custom.py:879-888: one K.conv2d per row slice + concatenate); the optional last layer of the functional U-Net
 *      (examples/train_functional.py:191-196).  The descriptor is dlwp_conv2d with dilation 1, DLWP_SRC_DIRECT -- or, for the
 *      forward and the weight gradient, DLWP_SRC_UPSAMPLE2: the loaders resolve a keras UpSampling2D(2) in front -- and a
 *      plain epilogue (halo, channel windows, bias, activation as for dlwp_conv2d_fwd; output shape: dlwp_conv2d_out_shape).
 *      w: (ho, kh, kw, cin, cout) -- custom.py:800-805; bias: the stored (ho, 1, cout) array of custom.py:812, which
 *      K.bias_add (Keras 2.2, tensorflow backend) reshapes to (1, cout, ho, 1) for channels_first: channel co, row r
 *      receives flat element co * ho + r (nullable).  float32 only; stride 1 (the reference's call sites).
 *      bwd_data : dx (n, cin, h, w) dense <- dL/dx (DLWP_SRC_DIRECT descriptors; behind an up-sampling the caller describes
 *                 the up-sampled tensor and finishes with dlwp_upsample2_bwd); the halo's adjoint is applied through a padded
 *                 temporary of dlwp_rowconv2d_bwd_workspace() bytes (0 without a halo).
 *      bwd_weight: dw (ho, kh, kw, cin, cout) and db (nullable; the stored (ho, 1, cout) layout) from x and dz =
 *                 dL/d(pre-activation), summed over samples and columns in a fixed order (bit-reproducible); accumulate != 0
 *                 adds to dw / db.
 *      _fwd_direct: one thread per output on the vector ALU, the in-library cross-check.
 *      dlwp_rowconv2d_uses_matrix_cores (host logic): 1 when pass 0 = forward | 1 = data | 2 = weight gradient of this
 *      geometry runs on the MFMA kernels, 0 when it takes the vector-ALU route (LDS footprint).                           */
int dlwp_rowconv2d_fwd(dlwp_handle_t, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                       const dlwp_conv2d* cd, int dtype, void* stream);
int dlwp_rowconv2d_fwd_direct(dlwp_handle_t, const void* x, const void* w, const void* bias, void* y, dlwp_shape4 xs,
                              const dlwp_conv2d* cd, int dtype, void* stream);
int dlwp_rowconv2d_uses_matrix_cores(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int pass);
int dlwp_rowconv2d_bwd_workspace(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, size_t* bytes);
int dlwp_rowconv2d_bwd_data(dlwp_handle_t, const void* dz, const void* w, void* dx, dlwp_shape4 xs, const dlwp_conv2d* cd,
                            int dtype, void* ws, size_t ws_bytes, void* stream);
int dlwp_rowconv2d_bwd_weight(dlwp_handle_t, const void* x, const void* dz, void* dw, void* db, dlwp_shape4 xs,
                              const dlwp_conv2d* cd, int accumulate, int dtype, void* stream);

/* ---- the rest of the train step: Keras 'mse' loss + 'mae' metric (examples/train.py:240, train_functional.py:285),
 *      activation backward, bias gradient, Keras-2.2-form Adam (restated by the reference at DLWP/custom.py:34-40) and
 *      SGD on flat parameter buffers.  Reductions use fixed trees: bit-reproducible.
 *      dlwp_mse_mae: out2[0] = mean((yp-yt)^2), out2[1] = mean(|yp-yt|) (device floats); dy (nullable) =
 *      loss_weight * 2*(yp-yt)/n.  ws >= dlwp_mse_mae_workspace() bytes.                                                */
int    dlwp_act_bwd(dlwp_handle_t, const void* y, const void* dy, void* dz, size_t n, int act, int dtype, void* stream);
size_t dlwp_bias_grad_workspace(int c);
int    dlwp_bias_grad(dlwp_handle_t, const void* dz, void* db, int n, int c, int c_off, int c_total, int hw, void* ws,
                      size_t ws_bytes, int dtype, void* stream);
/* dlwp_act_bwd on channels [c_off, c_off+c) of a (n, c_total, hw) tensor and dlwp_bias_grad of the result in one pass:
 * dz = dy * act'(y) (dz may alias dy), db[c] = sum of dz over (n, hw), by a fixed (bit-reproducible) reduction tree.       */
int    dlwp_act_bwd_bias_grad(dlwp_handle_t, const void* y, const void* dy, void* dz, void* db, int n, int c, int c_off,
                              int c_total, int hw, int act, void* ws, size_t ws_bytes, int dtype, void* stream);
/* dlwp_maxpool2_bwd + dlwp_act_bwd + dlwp_bias_grad in one pass, for a Conv2D whose only reader is MaxPooling2D(2)
 * (DLWP/model/models.py:188-228 train step of the U-Net encoders): y (n,c,h,w) the conv's output, dp (n,c,h/2,w/2) the
 * pooled tensor's gradient, dz (n,c,h,w) <- dL/d(pre-activation), db[c] (nullable) <- sum of dz.  ws as dlwp_bias_grad.  */
int    dlwp_pool_act_bwd_bias_grad(dlwp_handle_t, const void* y, const void* dp, void* dz, void* db, dlwp_shape4 ys, int act,
                                   void* ws, size_t ws_bytes, int dtype, void* stream);
size_t dlwp_mse_mae_workspace(dlwp_handle_t);
int    dlwp_mse_mae(dlwp_handle_t, const void* y_pred, const void* y_true, size_t n, void* out2, void* dy,
                    float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream);
/* dlwp_mse_mae taken on the PHASE channels of a restated output layer (dlwp_phase_weights: y_phase (n, 4f, h, w), output
 * pixel (2i+a, 2j+b) of field co in channel (2a+b) f + co) against the target y_true (n, f, 2h, 2w): out2 as above;
 * dz_phase (nullable, (n, 4f, h, w)) = what dlwp_space_to_depth2 makes of dy; db4f (nullable, 4f floats) = its sums over
 * (n, h, w), the bias gradient of a linear layer.  One pass instead of dlwp_depth_to_space2 + dlwp_mse_mae +
 * dlwp_space_to_depth2 + dlwp_bias_grad (the train step of examples/train.py's 5x5 output layer, DLWP/model/models.py:188-228). */
size_t dlwp_mse_mae_phase_workspace(int f);
int    dlwp_mse_mae_phase(dlwp_handle_t, const void* y_phase, const void* y_true, int n, int f, int h, int w, void* out2,
                          void* dz_phase, void* db4f, float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream);
/* the reference's custom losses, on the device (values + gradient, no host round trip):
 *   kind 0: mean((w (yp-yt))^2)                           -- latitude_weighted_loss(mse) (DLWP/custom.py:956-991)
 *   kind 1: regularizer - ACC,  ACC = mean(PT)/sqrt(mean(P^2) mean(T^2)), P = w*yp - mean, T = w*yt - mean
 *           -- anomaly_correlation_loss(mean, regularize_mean, reverse=True) (custom.py:1036-1088), optionally wrapped
 *           in latitude_weighted_loss (examples/train.py:224-234).  regularize: 0 none, 1 'mse', 2 'mae', 3 'global'
 *           (|(mean(T') - mean(P'))/mean(T')| over everything), 4 'spatial' (the same per (sample, channel) plane, averaged),
 *           P' = w*yp, T' = w*yt.
 *   y: (n, c, h, w);  mean: (c, h, w) or null;  row_weights: (h) or null.
 *   stats7 (device) <- {loss, mse, mae, S_pt, S_pp, S_tt, regularizer value};  dy (nullable) <- loss_weight * dL/dyp.   */
size_t dlwp_loss_workspace(dlwp_handle_t, int n, int c);
int    dlwp_loss_custom(dlwp_handle_t, const void* y_pred, const void* y_true, int n, int c, int h, int w,
                        const void* mean, const void* row_weights, int kind, int regularize, void* stats7, void* dy,
                        float loss_weight, void* ws, size_t ws_bytes, int dtype, void* stream);
int    dlwp_adam_keras(dlwp_handle_t, void* p, void* m, void* v, const void* g, size_t n, float lr, float beta_1,
                       float beta_2, float epsilon, float decay, long long iteration, float grad_scale, void* stream);
/* dlwp_adam_keras for a training step captured as a hipGraph: the step number is read from (and advanced in) device memory
 * -- *iteration_dev plays `iteration`, lr_t_scratch is one device float the update reads its step size from.              */
int    dlwp_adam_keras_dev(dlwp_handle_t, void* p, void* m, void* v, const void* g, size_t n, float lr, float beta_1,
                           float beta_2, float epsilon, float decay, long long* iteration_dev, float* lr_t_scratch,
                           float grad_scale, void* stream);
int    dlwp_sgd_keras(dlwp_handle_t, void* p, void* vel, const void* g, size_t n, float lr, float momentum, float decay,
                      long long iteration, float grad_scale, void* stream);
int    dlwp_axpby(dlwp_handle_t, const void* x, void* y, size_t n, float a, float b, void* stream);   /* y = a*x + b*y */
/* count <= 8 contiguous float32 copies dsts[i] <- srcs[i] (floats[i] elements each) in ONE launch: the batch and its targets
 * on their way into the fixed buffers a captured training step reads (a device-to-device memcpy each before: ~5 us apiece). */
int    dlwp_copy_many(dlwp_handle_t, const void* const* srcs, void* const* dsts, const size_t* floats, int count, void* stream);

/* ---- a training step's weight-side helpers in ONE launch each (csrc/batch.hip).  Keras / TF run one kernel per op
 *      (the train step behind DLWP/model/models.py:188-228); at the 8 samples per GPU of an 8-way data-parallel config-3
 *      step, 32 of 64 launches were sub-5-us helpers on weight-sized tensors.
 *   dlwp_prepare_begin .. dlwp_prepare_flush: dlwp_conv2d_prepare and dlwp_conv2d_bwd_data_prepare in between only RECORD
 *      their work (Winograd filter transforms, packed-N expansions, flipped / transposed kernels); flush builds all of it
 *      with one kernel on `stream`.  (bf16 arrangements are not batched: they run at once.)
 *   dlwp_reductions_begin .. dlwp_reductions_flush: the FINAL sums of dlwp_conv2d_bwd_weight (over its slabs),
 *      dlwp_bias_grad / dlwp_act_bwd_bias_grad / dlwp_pool_act_bwd_bias_grad (over their partials) and dlwp_mse_mae in
 *      between are recorded; their outputs (dw, db, out2) are undefined and their workspaces must stay untouched -- one
 *      workspace per call -- until flush sums everything with one kernel (fixed order: deterministic).  Two recorded sums
 *      into the same tensor (accumulate) flush in between.  At most 24 jobs per launch; more flush early.
 *   The modes live on the handle: one thread per handle while they are on.                                              */
int    dlwp_prepare_begin(dlwp_handle_t);
int    dlwp_prepare_flush(dlwp_handle_t, void* stream);
int    dlwp_reductions_begin(dlwp_handle_t);
int    dlwp_reductions_flush(dlwp_handle_t, void* stream);

/* ---- two INDEPENDENT launches of a training step as ONE kernel launch (r4; csrc/conv_pair.hip).  On grids under one round of
 *      workgroups (8 samples per GPU: every launch of a step) a layer's weight gradient and its data gradient -- both read the
 *      layer's pre-activation gradient, neither reads the other's output -- each leave most CUs idle; issued back to back they cost
 *      the sum of their lifetimes, on two queues more.  Between dlwp_pair_begin and dlwp_pair_end ONE
 *      dlwp_conv2d_bwd_weight and ONE dlwp_conv2d_bwd_data[_stored / _prepared] (in either order) hand their launch over instead of
 *      issuing it; dlwp_pair_end(stream) issues both on `stream` -- as one grid whose first blocks run the weight-gradient
 *      body and whose other blocks run the data-gradient body when a fused instance is compiled for the two kernel
 *      instances (the channel-block Winograd weight gradient on 4 x 32 tiles beside the 8 x 32 Winograd forward instance, 16- or
 *      9-position), one after the other otherwise.  Same bodies, same bits.  The CALLER states the independence; calls the
 *      mode does not cover (another family, a padded-gradient data gradient, a second call of the same kind) run at once.
 *      dlwp_pair_begin on a pair that was never ended (the caller's step raised) issues what that pair holds, launch by launch,
 *      and opens the new one.  dlwp_pair_fused_count: pairs this handle has issued as one launch so far (tests and tools).  */
int    dlwp_pair_begin(dlwp_handle_t);
int    dlwp_pair_end(dlwp_handle_t, void* stream);
long long dlwp_pair_fused_count(dlwp_handle_t);
/* The data gradient's operand -- the flipped / transposed kernel, followed by its Winograd / packed-N form when the
 * gradient's convolution runs on such an instance -- depends on the weights only: build it once per step
 * (dlwp_conv2d_bwd_data_prepare into `prepared`, dlwp_conv2d_bwd_data_prepared_bytes large; batched between
 * dlwp_prepare_begin / _flush) and pass it to dlwp_conv2d_bwd_data_prepared, which then launches the gradient's convolution
 * only.  stored: 0 = dlwp_conv2d_bwd_data semantics, 1 = dlwp_conv2d_bwd_data_stored.  The workspace may be smaller by the
 * kernel's size (kh kw cin cout floats, rounded up to 256 bytes).                                                          */
size_t dlwp_conv2d_bwd_data_prepared_bytes(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int stored);
int    dlwp_conv2d_bwd_data_prepare(dlwp_handle_t, const void* w, void* prepared, dlwp_shape4 xs, const dlwp_conv2d* cd,
                                    int stored, void* stream);
int    dlwp_conv2d_bwd_data_prepared(dlwp_handle_t, const void* dz, const void* prepared, void* dx, dlwp_shape4 xs,
                                     const dlwp_conv2d* cd, int dtype, void* ws, size_t ws_bytes, int stored, void* stream);

/* ---- keras MaxPooling2D(2) / UpSampling2D(2) standalone (examples/train.py:171,181,191,201) ------------------------ */
int dlwp_maxpool2_fwd (dlwp_handle_t, const void* x, void* y, dlwp_shape4 xs, int dtype, void* stream);
int dlwp_maxpool2_bwd (dlwp_handle_t, const void* x, const void* dy, void* dx, dlwp_shape4 xs, int dtype, void* stream);
int dlwp_upsample2_fwd(dlwp_handle_t, const void* x, void* y, dlwp_shape4 xs, int dtype, void* stream);
int dlwp_upsample2_bwd(dlwp_handle_t, const void* dy, void* dx, dlwp_shape4 xs, int dtype, void* stream);

/* ---- slice_layer (custom.py:675-692) / keras concatenate(axis=1): copy c channels between buffers of different
 *      channel counts.  src (n, src_c_total, h, w) channels [src_off, src_off+c) -> dst channels [dst_off, dst_off+c). */
int dlwp_copy_channels(dlwp_handle_t, const void* src, void* dst, int n, int c, int hw, int src_c_off, int src_c_total,
                       int dst_c_off, int dst_c_total, int dtype, void* stream);

/* ---- predict_timeseries bookkeeping (DLWP/model/models.py:294-300, 448-451): series (T, N, time_dim, V, H*W) ->
 *      (T*time_dim, N, V, H*W).                                                                                        */
int dlwp_series_merge_time(dlwp_handle_t, const void* series, void* out, int t, int n, int time_dim, int v, int hw,
                           int dtype, void* stream);

/* ---- Conv2D on a 2x nearest-neighbour up-sampled tensor, restated on the tensor itself (the decoder layers
 *      UpSampling2D -> padding -> Conv2D of examples/train.py:191-219, Azure/train_tf.py:247-268): the k x k taps fall on few
 *      distinct source pixels, so each of the 4 output phases is a small kernel of summed weights on the low-resolution
 *      tensor; the 4 phases run as ONE convolution with 4*cout channels + a depth-to-space interleave (phase.hip).
 *      dlwp_phase_geometry: per axis, the window [lo, hi] of source offsets and its size k2 for kernel size k and the
 *      top / left halo `pad` of the up-sampled tensor.  dlwp_phase_weights: w (kh,kw,cin,cout), bias -> w2
 *      (kh2,kw2,cin,4*cout), b2 (4*cout, nullable) with column (2a+b)*cout + co for output phase (a, b).                */
int dlwp_phase_geometry(int k, int pad, int* k2, int* lo, int* hi);
int dlwp_phase_weights(dlwp_handle_t, const void* w, const void* bias, void* w2, void* b2, int kh, int kw, int cin, int cout,
                       int pad_top, int pad_left, int dtype, void* stream);
int dlwp_depth_to_space2(dlwp_handle_t, const void* src, void* dst, int n, int f, int h, int w, int c_off, int c_total,
                         int dtype, void* stream);
/* their adjoints, for the training step: dW (+)= the gather of dW2 over the phases (db likewise, nullable pair), and the
 * inverse interleave (n, c_total, 2h, 2w)[c_off:+f] -> (n, 4f, h, w) of a gradient.                                     */
int dlwp_phase_weights_bwd(dlwp_handle_t, const void* dw2, const void* db2, void* dw, void* db, int kh, int kw, int cin,
                           int cout, int pad_top, int pad_left, int accumulate, int dtype, void* stream);
int dlwp_space_to_depth2(dlwp_handle_t, const void* src, void* dst, int n, int f, int h, int w, int c_off, int c_total,
                         int dtype, void* stream);

/* ---- ConvLSTM2D cell update (keras ConvLSTM2DCell.call; call sites examples/train.py:148-155,
 *      examples/train_functional.py:207-219).  zx / zh: (n, 4F, h*w) gate pre-activations i | f | c | o from the input
 *      convolution (+bias) and the recurrent 'same' convolution of h_{t-1}; zh and c_prev may be NULL on the first step
 *      (h = c = 0).   c = rec(z_f)*c_prev + rec(z_i)*act(z_c);  h = rec(z_o)*act(c).   c_out: dense (n, F, h*w); h is
 *      written to channels [h_c_off, +F) of an h_c_total-channel buffer (the return_sequences output (T*F, h, w)).
 *      act: DLWP_ACT_*; rec_act: 0 = hard_sigmoid (Keras default), 1 = sigmoid.  dtype: DLWP_F32, or
 *      DLWP_DTYPE_IO(z, h) = storage of the gate pre-activations zx / zh and of h (DLWP_F32 | DLWP_BF16 each); the cell
 *      state c and the arithmetic are float32.                                                                        */
/* A ConvLSTM2D step with the cell update in the convolution's epilogue (bfloat16 inference, BASELINE config 4): the 4F gate
 * pre-activations z = conv(x; w) + bias + z_add are NOT stored; the epilogue writes c_out (float32, (n, F, ho, wo)) and h_t
 * (channels [cd->out_c_off, +F) of an out_c_total-channel buffer, storage DLWP_DTYPE_OUT(dtype)).  z_add: the other
 * convolution's stored pre-activations, bfloat16 (n, 4F, ho, wo), or NULL (first step: no recurrent term); c_prev: float32
 * (n, F, ho, wo) or NULL.  cd->lstm_f = F, cd->cout = 4F.  Runs on the bf16 matrix-core instances with 64-channel blocks
 * (4 gates x 16 hidden channels per block): ask dlwp_convlstm_conv_supported first (needs wo % 4 == 0 and a layer the bf16
 * family covers).  prepared: NULL, or the buffer dlwp_conv2d_prepare filled for (xs, cd, dtype).
 * Replaces: one of the two convolutions of a step + dlwp_convlstm_gates (keras ConvLSTM2DCell.call).                      */
int dlwp_convlstm_conv_fwd(dlwp_handle_t, const void* x, const void* w, const void* prepared, const void* bias,
                           const void* z_add, const void* c_prev, void* c_out, void* h_out, dlwp_shape4 xs,
                           const dlwp_conv2d* cd, int dtype, void* stream);
int dlwp_convlstm_conv_supported(dlwp_handle_t, dlwp_shape4 xs, const dlwp_conv2d* cd, int dtype);
/* One ConvLSTM2D step t >= 1 in ONE launch (r3; keras ConvLSTM2DCell.call, the recurrent front end of examples/train.py:144-157):
 * z = conv_h(h_{t-1}) + conv_x(x_t) + bias, then the cell update -- the input convolution's 4 F pre-activations are neither
 * stored nor read back.  cd_h: the recurrent convolution (3x3, 'same' zero halo 1, dilation 1, lstm_f = F <= 24, its channel
 * windows on the h sequence: in = h_{t-1}, out = h_t), xs_h = (n, F, H, W); cd_x: the input convolution (3x3, dilation 2, halo 2
 * of any mode, its channel window on the float32 state), xs_x = (n, Cx <= 8, H, W).  dtype = DLWP_DTYPE_IO(DLWP_BF16_O8,
 * DLWP_BF16_O8): h in octets; c_prev / c_out float32 octets (N, F/8, H, W, 8).  w_h / w_x: the two HWIO kernels (ignored when
 * `prepared` -- dlwp_convlstm_step_prepare's output, dlwp_convlstm_step_prepared_bytes large -- is given); bias: (4 F).       */
int    dlwp_convlstm_step_supported(dlwp_handle_t, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                    const dlwp_conv2d* cd_x, int dtype);
size_t dlwp_convlstm_step_prepared_bytes(dlwp_handle_t, dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x,
                                         const dlwp_conv2d* cd_x, int dtype);
int    dlwp_convlstm_step_prepare(dlwp_handle_t, const void* w_h, const void* w_x, void* prepared, dlwp_shape4 xs_h,
                                  const dlwp_conv2d* cd_h, dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype, void* stream);
int    dlwp_convlstm_step_fwd(dlwp_handle_t, const void* h_in, const void* x_in, const void* w_h, const void* w_x,
                              const void* prepared, const void* bias, const void* c_prev, void* c_out, void* h_out,
                              dlwp_shape4 xs_h, const dlwp_conv2d* cd_h, dlwp_shape4 xs_x, const dlwp_conv2d* cd_x, int dtype,
                              void* stream);
int dlwp_convlstm_gates(dlwp_handle_t, const void* zx, const void* zh, const void* c_prev, void* c_out, void* h_out,
                        int n, int f, int hw, int h_c_off, int h_c_total, int act, int rec_act, int dtype, void* stream);
/* backward of the cell update (one step of back-propagation through time behind DLWPNeuralNet.fit on the recurrent
 * model): zx, zh, c_prev as given to the forward, c = the c_out it produced, dh = dL/dh_t read from channels
 * [h_c_off, +F) of an h_c_total-channel gradient buffer, dc_in = dL/dc_t arriving from step t+1 (NULL on the last step).
 * Writes dz (n, 4F, h*w) = dL/d zx = dL/d zh and dc_prev = dL/dc_{t-1} (NULL allowed when there is no previous step). */
int dlwp_convlstm_gates_bwd(dlwp_handle_t, const void* zx, const void* zh, const void* c_prev, const void* c,
                            const void* dh, const void* dc_in, void* dz, void* dc_prev, int n, int f, int hw,
                            int h_c_off, int h_c_total, int act, int rec_act, int dtype, void* stream);

/* ---- rollout: the N-step predict_timeseries loop (DLWP/model/models.py:277-293, 439-447) captured as ONE hipGraph.
 *      A plan is an array of dlwp_op describing one model call; buffer index >= 0 = caller scratch buffer,
 *      DLWP_BUF_STATE_IN = the call's input state, DLWP_BUF_OUT(o) = output o of the call.  Call t reads
 *      state0 (t == 0) or slot t*n_outputs-1 of `series` and writes slots [t*n_outputs, (t+1)*n_outputs).          */
#define DLWP_BUF_STATE_IN (-1)
#define DLWP_BUF_OUT(o)   (-2 - (o))
#define DLWP_OP_CONV2D     0
#define DLWP_OP_PAD2D      1
#define DLWP_OP_MAXPOOL2   2
#define DLWP_OP_UPSAMPLE2  3
#define DLWP_OP_COPYCH     4
#define DLWP_OP_LSTM_GATES 5
#define DLWP_OP_PHASE_WEIGHTS 6   /* src = kernel buffer, w = -1 | bias buffer index in b, dst = w2 buffer, aux[0] = b2 buffer |
                                   * -1000; conv = {cout, kh, kw, halo.top, halo.left}, xs.c = cin.  Run ONCE at the head of
                                   * the rollout graph (the weights do not change inside a launch). */
#define DLWP_OP_DEPTH2SPACE 7     /* src (n, 4F, h, w) -> dst window [conv.out_c_off, +F) of conv.out_c_total channels at
                                   * (2h, 2w); xs = (n, F, h, w) */
#define DLWP_OP_ROWCONV2D  8      /* dlwp_rowconv2d_fwd: src, dst, w, b, xs, conv as for DLWP_OP_CONV2D (float32 buffers) */
typedef struct {
  int kind;                 /* DLWP_OP_*                                                                  */
  int src, dst;             /* buffer indices                                                              */
  int w, b;                 /* weight / bias buffer indices (conv only)                                    */
  dlwp_shape4 xs;           /* stored input shape of this op                                               */
  dlwp_conv2d conv;         /* DLWP_OP_CONV2D; for DLWP_OP_COPYCH: in_c_off/in_c_total/out_c_off/out_c_total */
  dlwp_pad2d pad;           /* DLWP_OP_PAD2D (NHWC: xs = (n,1,h,w) and conv.in_c_total = channels)          */
  int aux[4];               /* DLWP_OP_LSTM_GATES: src = zx, dst = h buffer (window conv.out_c_off/out_c_total), xs =
                             * (n, F, h, w), aux = {zh | -1000, c_prev | -1000, c_out, rec_act + 256 * (h buffer stored as
                             * bfloat16) + 512 * (zx / zh stored as bfloat16)}, conv.act = activation.
                             * DLWP_OP_CONV2D / DLWP_OP_MAXPOOL2: aux[0] = storage dtype of this op's tensors (DLWP_F32,
                             * DLWP_BF16 or DLWP_DTYPE_IO(in, out)); the rollout's own dtype describes state and series.
                             * DLWP_OP_CONV2D with conv.lstm_f > 0 (dlwp_convlstm_conv_fwd): dst = h buffer, aux[1..3] =
                             * {z_add | -1000, c_prev | -1000, c_out}                                                  */
  /* DLWP_OP_CONV2D with conv.lstm_f > 0 and src2 != -1000: a whole ConvLSTM2D step (dlwp_convlstm_step_fwd) -- conv = the
   * recurrent convolution, src = dst = the h buffer, w / b = recurrent kernel / the layer's bias; src2 = the float32 state the
   * input convolution conv2 reads xs2_c channels of, w2 = its kernel.  Every other op: src2 = -1000.                       */
  int src2, w2, xs2_c;
  dlwp_conv2d conv2;
} dlwp_op;
#define DLWP_BUF_NONE (-1000)
typedef struct dlwp_rollout* dlwp_rollout_t;
/* workspace: caller-owned device memory of dlwp_rollout_workspace_bytes() bytes that must outlive the rollout -- the
 * prepared weights of the Winograd / packed-N / bf16 layers live there, rebuilt by the first kernels of every launch (the
 * weights may change between launches).  The library allocates no device memory for a rollout.                        */
size_t dlwp_rollout_workspace_bytes(dlwp_handle_t, const dlwp_op* plan, int n_ops, int groups);
int dlwp_rollout_create(dlwp_handle_t, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers,
                        const void* state0, void* series, size_t slot_elems, int calls, int n_outputs, int dtype,
                        void* workspace, size_t workspace_bytes, dlwp_rollout_t* out);
/* The same with the members (xs.n of the ops) split into `groups` equal parts captured as parallel graph branches:
 * members are independent, so an ensemble can run as several chains whose kernels fill the gaps each other's launches
 * leave.  buffer_sample_bytes[i] = bytes ONE member occupies in scratch buffer i (0 for weight / bias buffers); the member
 * count must be a multiple of groups.  groups = 1 is dlwp_rollout_create.                                               */
int dlwp_rollout_create_grouped(dlwp_handle_t, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers,
                                const size_t* buffer_sample_bytes, int groups, const void* state0, void* series,
                                size_t slot_elems, int calls, int n_outputs, int dtype, void* workspace,
                                size_t workspace_bytes, dlwp_rollout_t* out);
/* `dtype` of the two create calls may carry DLWP_ROLLOUT_PREPARED: the graph is a later TIME SLICE of a rollout (state0 = the
 * previous slice's last series slot) that shares `workspace` with the first slice and is always launched behind it on the same
 * stream -- it does not prepare the weights again.                                                                          */
#define DLWP_ROLLOUT_PREPARED 0x100
/* ---- a rollout whose model outputs are NOT its next inputs: TimeSeriesEstimator.predict (DLWP/model/extensions.py:206-240, the
 *      forecast examples/validate.py:191-205 runs: inputs = 2 time steps x (variables + insolation), outputs = 2 x variables) and
 *      the step_sequence branch of DLWPNeuralNet.predict_timeseries (DLWP/model/models.py:280-290).  Between two model calls the
 *      reference rebuilds the input on the host (reindex of the sample axis by k, impute, insolation of the rows past the data,
 *      .loc assignment of the predicted channels); dlwp_state_feedback is that update as ONE launch on device-resident tensors:
 *      for every row i < rows and state channel c, in the reference's order of precedence
 *          src[c] = -1 - j              -> out[i, j]                                   (the model predicts this channel)
 *          i >= rows - tail, sol[c] >= 0 -> sol[i - (rows - tail), sol[c]]              (insolation, known for any time)
 *          i >= rows - tail, mean        -> mean[c]                                     (impute=True)
 *          i + shift < rows              -> old_state[i + shift, src[c]]                (the data of the later start time)
 *          otherwise                     -> NaN                                         (ran out of data)
 *      old_state / new_state (rows, state_c, hw), out (rows, out_c, hw), sol (tail, sol_planes, hw) | NULL, mean (state_c, hw) |
 *      NULL; float32, planes are copied bit for bit.  new_state must not alias old_state.                                      */
#define DLWP_FB_MAX_CHANNELS 128
typedef struct {
  int rows, state_c, out_c, hw;
  int shift;                        /* k = es + interval - 1: series steps the window advances per model call (0: none)      */
  int tail;                         /* min(es, rows): trailing rows that receive the insolation / the mean state             */
  int sol_planes;                   /* planes per tail row of the insolation block (the input time steps)                    */
  int src[DLWP_FB_MAX_CHANNELS];    /* per state channel: >= 0 channel of the old state; -1 - j: channel j of the output     */
  int sol[DLWP_FB_MAX_CHANNELS];    /* per state channel: plane of the tail row's insolation block, or -1                    */
} dlwp_feedback;
int dlwp_state_feedback(dlwp_handle_t, const void* old_state, const void* out, void* new_state, const void* sol,
                        const void* mean, const dlwp_feedback* fb, int dtype, void* stream);
/* The rollout graph with that update between its calls: call t reads state[t & 1] (state_a = the initial state, written by the
 * caller before every launch; state_b = scratch of the same size), writes its series slot (slot_elems = rows * out_c * hw), and
 * the feedback launch behind it builds state[(t + 1) & 1] with the insolation block sol + t * tail * sol_planes * hw
 * (sol: (calls of the whole rollout - 1, tail, sol_planes, hw) | NULL).  One output per call, one member chain (rows exchange
 * data).  A TIME SLICE of a rollout (one graph per model call, so that a finished slot can leave for the host under the next
 * call): the graph holds calls [first_call, first_call + calls), `series` points at the slot of first_call, feed_last != 0
 * puts the feedback behind its last call into the graph too (every slice but the last); DLWP_ROLLOUT_PREPARED as above.        */
int dlwp_rollout_create_fed(dlwp_handle_t, const dlwp_op* plan, int n_ops, void* const* buffers, int n_buffers,
                            void* state_a, void* state_b, void* series, size_t slot_elems, int calls, int first_call,
                            int feed_last, const dlwp_feedback* fb, const void* sol, const void* mean, int dtype,
                            void* workspace, size_t workspace_bytes, dlwp_rollout_t* out);
/* ---- the returned series of such a forecast (DLWP/model/extensions.py:243-302): one model call's output (n, t, c, hw) -> its
 *      block of the result, time first (kept, n, c, hw) -- the transpose + reshape of :262-263 and the [:, :, :es] cut of :260 --
 *      or (n, kept, c, hw) with keep_time_dim; destination channel j <- source channel perm[j] (NULL: identity): the sorted
 *      (variable, level) order xarray's unstack returns (:298-302).                                                          */
int dlwp_series_arrange(dlwp_handle_t, const void* src, void* dst, int n, int t, int c, int hw, int kept, const int* perm,
                        int time_major, int dtype, void* stream);
int dlwp_rollout_launch(dlwp_rollout_t, void* stream);
int dlwp_rollout_destroy(dlwp_rollout_t);

/* ---- host side of the DataGenerator feed (keras.utils.Sequence batches assembled by worker processes under
 *      fit_generator(use_multiprocessing=True), DLWP/model/models.py:216-228, generators.py:137-159): dst[i] = src[rows[i]],
 *      rows of row_bytes bytes, on `threads` host threads -- straight into the pinned staging buffer of the H2D copy.
 *      Host memory only; no device work, no handle.                                                                          */
int dlwp_host_gather_rows(void* dst, const void* src, const long long* rows, long long n_rows, size_t row_bytes,
                          long long src_rows, int threads);
/* ---- the training step as a library object: replaces the per-step Python launch loop behind keras Model.train_on_batch as
 *      DLWPNeuralNet.fit / fit_generator drive it (DLWP/model/models.py:188-228; examples/train.py:262-263).
 *      A model's step is ~30 launches through this ABI.  Between dlwp_train_step_record_begin and dlwp_train_step_create every
 *      launch-type entry point the RECORDING THREAD calls on the handle is executed as usual and appended to a tape: a closure over
 *      its arguments (descriptors copied; device pointers as given -- the caller keeps that memory alive and in place).  Streams are
 *      recorded as lanes (lane 0 = main_stream), fork / join edges between them go through dlwp_stream_wait.  The step object
 *      replays the tape with ONE C call: DLWP_STEP_LANES issues the launches one by one (lane 0 on the given stream, the others on
 *      the step's own side streams), DLWP_STEP_GRAPH replays one hipGraph captured from that sequence on a single stream (built at
 *      the first launch in that mode), DLWP_STEP_GRAPH_BRANCHES one with the lanes as graph branches.  What varies from step to
 *      step must live in device memory (the Adam step number: dlwp_adam_keras_dev); collectives stay outside.
 *      in_dst / in_floats: the (at most 8) device buffers the recorded launches read the batch and its targets from;
 *      dlwp_train_step_launch(srcs != NULL) copies srcs[i] there first, in one launch.                                        */
#define DLWP_STEP_LANES          0
#define DLWP_STEP_GRAPH          1
#define DLWP_STEP_GRAPH_BRANCHES 2
#define DLWP_STEP_LANES_RECORDED 3  /* as DLWP_STEP_LANES, the side lanes on the streams the step was RECORDED on (the caller keeps them
                                     * alive): streams created later may share a hardware queue with the main stream            */
typedef struct dlwp_train_step* dlwp_train_step_t;
int dlwp_train_step_record_begin(dlwp_handle_t, void* main_stream);
int dlwp_train_step_record_abort(dlwp_handle_t);
int dlwp_stream_wait(dlwp_handle_t, void* waiter_stream, void* signaler_stream);   /* waiter waits for what signaler holds now */
int dlwp_train_step_create(dlwp_handle_t, int n_in, void* const* in_dst, const size_t* in_floats, dlwp_train_step_t* out);
int dlwp_train_step_info(dlwp_train_step_t, int* n_launches, int* n_lanes, int* n_waits);
int dlwp_train_step_launch(dlwp_train_step_t, const void* const* srcs, int mode, void* stream);
int dlwp_train_step_destroy(dlwp_train_step_t);

/* ---- data parallel training: replaces keras.utils.multi_gpu_model (DLWP/model/models.py:104-109, 365-372; batch =
 *      n_gpu x batch, Azure/train_tf.py:163-164).  One process per GPU; every rank holds the whole (< 1 MB) weight set and
 *      trains on its rows of the global batch; the ONE flat fp32 gradient buffer is summed by ONE RCCL all-reduce over
 *      xGMI per step, enqueued on the training stream right behind the last weight-gradient kernel, and dlwp_adam_keras
 *      (grad_scale = 1/world) consumes it on the same stream.  RCCL is bound with dlopen at the first dlwp_comm_* call
 *      (the process' already-loaded librccl.so.1 if there is one), so single-GPU use never touches it.
 *      dlwp_comm_unique_id: id == NULL -> *id_bytes = size needed (128); otherwise rank 0 fills id and the CALLER ships
 *      the bytes to the other ranks (any channel: the launcher's TCP store, MPI, a file).  dlwp_comm_init_rank is
 *      collective over all `world` ranks; `device` is the HIP device of the calling rank.  Buffers are device memory;
 *      both collectives are in place and asynchronous on `stream`.                                                    */
typedef struct dlwp_comm* dlwp_comm_t;
int dlwp_comm_unique_id(void* id, size_t* id_bytes);
int dlwp_comm_init_rank(dlwp_comm_t* comm, int device, int world, int rank, const void* unique_id, size_t id_bytes);
int dlwp_comm_info(dlwp_comm_t, int* world, int* rank, int* rccl_version);
int dlwp_allreduce_sum_f32(dlwp_comm_t, void* flat, size_t n, void* stream);
int dlwp_broadcast_f32(dlwp_comm_t, void* flat, size_t n, int root, void* stream);   /* replicas start identical */
int dlwp_comm_destroy(dlwp_comm_t);

/* ---- a one-shot all-reduce of our own for the same exchange, fused with the Keras-form Adam update (csrc/xchg.hip; SURVEY 5 /
 *      8e: the step's 756 KB buffer is latency-bound -- a ring pays 2 (W - 1) hops, this ONE).  Every rank owns a region of device
 *      memory (header with one flag per rank + two payload buffers, step parity) that its peers map through hipIpcMemHandle (xGMI
 *      peers of one node; on a one-GPU test box two processes of the same device); the region is UNCACHED device memory where the
 *      runtime exports an IPC handle for it (dlwp_xchg_info: memory_kind 2; 1 fine-grained, 0 plain).  One launch per step and
 *      rank, a persistent grid of at most one workgroup per CU whatever n: publish the flat buffer, raise the flag in every peer's
 *      header, wait for the peers' flags (bounded: 2 s; then the launch leaves p / m / v alone, writes NaN over flat[n_params:]
 *      -- the loss table -- and dlwp_xchg_status reports the time-out, for good: the exchange is dead afterwards), read all W
 *      buffers, sum them IN RANK ORDER -- identical bits on every rank -- and update the rank's own parameters.
 *      n (floats, a multiple of 4) is fixed at creation.  RCCL (dlwp_comm_*) stays the default
 *      transport; dlwp_amd/parallel.py takes this one with DLWP_ALLREDUCE=oneshot.
 *      create (every rank) -> exchange the 64-byte handles through any host channel -> connect (world x 64 bytes, rank order).   */
typedef struct dlwp_xchg* dlwp_xchg_t;
int dlwp_xchg_create(dlwp_handle_t, int world, int rank, size_t n_floats, void* ipc_handle_out_64_bytes, dlwp_xchg_t* out);
int dlwp_xchg_connect(dlwp_xchg_t, const void* handles);
int dlwp_xchg_allreduce_sum_f32(dlwp_xchg_t, void* flat, size_t n, void* stream);
int dlwp_xchg_allreduce_adam(dlwp_xchg_t, void* flat, size_t n_params, size_t n, void* p, void* m, void* v, float lr, float beta_1,
                             float beta_2, float epsilon, float decay, long long iteration, float grad_scale, void* stream);
int dlwp_xchg_status(dlwp_xchg_t, int* timed_out);
int dlwp_xchg_info(dlwp_xchg_t, int* memory_kind, int* max_blocks);
int dlwp_xchg_destroy(dlwp_xchg_t);

#ifdef __cplusplus
}
#endif
#endif /* DLWP_HIP_H */
