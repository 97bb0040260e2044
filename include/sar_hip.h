/*
 * sar_hip.h -- C ABI of libsar_hip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * skeleton-action-recognition hot path (ST-GCN training, VirtualRadar spectrograms).
 *
 * The reference (itskalvik/skeleton-action-recognition) is pure Python and has no FFI of its own:
 * every entry point below replaces the framework op(s) the reference invokes at the cited
 * file:line.  Conventions shared by every function:
 *   - extern "C", plain C types; every pointer is a DEVICE pointer owned by the caller unless the
 *     parameter is documented as host memory.  The library never allocates or frees device memory.
 *   - the last argument is the hipStream_t (passed as void*) the work is enqueued on; calls are
 *     asynchronous with respect to the host and re-entrant on distinct streams.
 *   - return value: 0 = OK, negative = argument error (SAR_E_*), positive = hipError_t.
 *     sar_last_error_string() returns a thread-local description of the last failure.
 *
 * Activation layout ("CN"): an activation with C channels over B sequences of T frames x V joints
 * is a row-major matrix [C][ld], ld >= B*T*V, column n = (b*T + t)*V + v.  (The reference's NCHW
 * tensor (B,C,T,V) is this matrix with the batch axis moved inside the row.)
 * Weight layouts are the reference's Keras HWIO layouts: graph conv (Cin, K*F) -- channel k*F+f as
 * in models/gcn.py:207 --, temporal conv (Kt, Cin, F), residual conv (Cin, F), logits (Cin, classes).
 */
#ifndef SAR_HIP_H
#define SAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAR_E_ARG (-1)      /* invalid argument (null pointer, bad size, unsupported shape) */
#define SAR_E_UNSUP (-2)    /* valid but unsupported configuration (e.g. adjacency denser than 4 nnz/column) */

typedef void* sar_stream_t;

int sar_version(void);
const char* sar_last_error_string(void);   /* thread-local */
/* (the diagnostic entry points sar_debug_* are declared in include/sar_hip_debug.h and exist only in a `make DEBUG=1`
 * build of the library: they are not part of the product ABI) */

/* ------------------------------------------------------------------------------------------------
 * Caller-owned launch context.  The library keeps NO global mutable state: everything is asynchronous on the stream the
 * caller passes and re-entrant from several host threads on distinct streams.  A few operators are faster when they fan
 * out over several streams -- today the 3x3 / stride-2 data gradient of sar_conv2d_gemm_f32, whose four parity-class
 * launches are individually too small to fill the chip.  The streams and events that takes are owned by a sar_context
 * the CALLER creates (one per host thread and device) and hands in through sar_conv2d_desc.ctx; with ctx == NULL the same
 * launches run one after the other on the caller's stream.  A context is created for the CURRENT device
 * (hipGetDevice); used while another device is current it is ignored (caller's stream only).  Whatever happens inside
 * the call -- including a failing launch -- the side streams are joined back into the caller's stream before it returns.
 * ------------------------------------------------------------------------------------------------ */
typedef struct sar_context sar_context;
int sar_context_create(sar_context** out);   /* 3 non-blocking side streams + 4 events; 0 or a hipError_t */
int sar_context_destroy(sar_context* ctx);   /* waits for nothing: synchronise the caller's stream first */

/* A caller-owned stream confined to the compute units whose bits are set in mask[nwords] (bit i of word j = CU 32 j + i;
 * hipExtStreamCreateWithCUMask).  The engines put their weight-gradient stream on such a stream so that the short kernels of the main
 * chain are not queued behind its long-lived workgroups (sar_amd/ops.py: shared_side_stream).  No reference counterpart
 * (MirroredStrategy / DataParallel leave stream placement to the framework, main_gnn.py:257-258). */
int sar_stream_create_cu_mask(const uint32_t* mask, int nwords, sar_stream_t* out);
int sar_stream_destroy(sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Fused conv-GEMM on the CN layout (fp32 MFMA, v_mfma_f32_32x32x2_f32).
 *
 *   out[m, n] = sum_tap sum_c W[tap][c][m] * OP_tap(pro(src))[c, n]  (+ bias term) ; epilogue
 *
 * mode SAR_CONV_GRAPH    : OP_k(x)[c,(b,t,w)] = sum_v x[c,(b,t,v)] * A[k,v,w]  (A given as <=4-nnz
 *                          column gather lists).  With W = the GraphConvTD kernel this is
 *                          models/gcn.py:199-209 (1x1 Conv2D + reshape + einsum 'nkctv,kvw->nctw')
 *                          in the reordered form  sum_k W_k (x A_k) + sum_k b_k colsum(A_k);
 *                          with A^T lists and W^T it is that layer's data gradient.
 * mode SAR_CONV_TEMPORAL : OP_tap(x)[c,(b,to,v)] = x[c,(b, to*stride + tap - pad, v)] (zero outside
 *                          [0,T_src)): Conv2D(F,[Kt,1],strides=[s,1],'same'), models/stgcn.py:29-36
 *                          (pad = TF-SAME begin pad), and with taps=1,pad=0 the strided 1x1
 *                          residual conv, models/stgcn.py:47-54.
 *                          transposed=1 gives the data gradient of that conv: out frame t gathers
 *                          src frame (t + pad - tap)/stride when divisible.
 * pro(x) = x                                   if pro_scale == NULL
 *        = relu?(x * pro_scale[c] + pro_shift[c])  (BatchNormalization+ReLU of models/stgcn.py:27-28
 *          folded into the operand load; padding stays exactly 0 as in the reference)
 * epilogue (epi):
 *   SAR_EPI_NONE   store
 *   SAR_EPI_STATS  store + per-row partial (sum, sum of squares) -> partials (train-mode BN statistics)
 *   SAR_EPI_MASK   val = (aux*aux_scale[m]+aux_shift[m] > 0) ? val : 0 ; store ;
 *                  partial (sum val, sum val*(aux-aux_mean[m]))   (ReLU+BN backward reductions, centred)
 *   SAR_EPI_ADD    val += aux ; store               (residual gradient accumulation)
 * partials layout: [M][nparts][2] with nparts = sar_conv_gemm_nparts(desc).
 * Limits (checked, SAR_E_ARG otherwise): ld_out, ld_aux < 2^22 columns, T_src*V < 2^28, the weight tensor within
 * 2^28 floats (the kernels address rows with 32-bit byte offsets through buffer descriptors); V <= 64.
 * ------------------------------------------------------------------------------------------------ */
enum { SAR_CONV_GRAPH = 0, SAR_CONV_TEMPORAL = 1 };
enum { SAR_EPI_NONE = 0, SAR_EPI_STATS = 1, SAR_EPI_MASK = 2, SAR_EPI_ADD = 3,
       /* graph data gradient only, round 4: out = gate(acc + aux), where gate keeps channel c of column n iff -- sar_conv_gemm_cn8 (bf16):
        * bit (c & 7) of aux_mask[(c >> 3) * ld_aux2 + n]; sar_conv_gemm_f32: bit (n & 3) of aux_mask[c * (ld_aux2 / 4) + (n >> 2)],
        * aux2 = [M][ld_aux2] floats, ld_aux2 % 4 == 0, M % 8 == 0 -- is set (the ReLU mask a block tail wrote), and the BatchNorm-backward sums
        * of the tail that PRODUCED aux2 are reduced from the same accumulators: partials[m][part] = (sum out, sum out (aux2 - aux_mean)).
        * It replaces, for the block below, the masked-gradient write of bn_add_relu_bwd_apply and the whole bn_add_relu_bwd_reduce
        * pass (models/stgcn.py:37,62-63 backward). */
       SAR_EPI_ADD_GATE = 4 };

/* g_flags: the caller asserts that EVERY gather weight g_wt is exactly representable in bfloat16 (the NTU adjacency of
 * graph/ntu_rgb_d.py holds 0.25 / 0.5 / 1).  The bf16 (CN8) kernels then apply a dense adjacency slice on the matrix cores
 * (z_k = x . A_k as a 16 x 16 x 32 MFMA per frame, exact products, fp32 accumulation) instead of in the vector ALU; without
 * the flag they keep the fp32 gather.  Ignored by the fp32 kernels. */
#define SAR_GRAPH_WT_BF16_EXACT 1
/* g_flags: the caller asserts that at most n = (g_flags >> SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff of the 3 V gather lists are
 * anything else than {one entry of weight 1} or empty (NTU 'spatial' adjacency, graph/tools.py:11-30: 2 of 75 forward, 8 of 75
 * transposed).  The bf16 (CN8) graph convolution then reads the operand of such a trivial list straight from the raw tile at
 * the listed joint and builds only the n remaining lists per frame (csrc/conv_graph_cn8.hip), bit-identical with the kernel
 * that builds every gathered tile.  A count that is too small makes the excess lists read as empty (wrong result, no fault). */
#define SAR_GRAPH_FEW_DENSE 4
/* g_flags: the caller asserts that slice 0's gather lists are the identity ({(w, 1.0)} for every joint w: the self-links of the
 * 'spatial' strategy, graph/tools.py:22-30).  The split graph weight gradient then stages the raw tile once (csrc/conv_wgrad_split.hip). */
#define SAR_GRAPH_SLICE0_IDENTITY 8
#define SAR_GRAPH_FEW_DENSE_SHIFT 8
/* g_flags (sar_conv_gemm_split, GRAPH, SAR_SPLIT_F16X3A): take the one-tile-per-workgroup kernel of round 5 instead of the persistent
 * LDS-DMA kernel (bit-identical results; kept for A/B measurements and as the cross-check of tests/test_gpu_split.py) */
#define SAR_GRAPH_ONE_TILE_WG 16
/* g_flags (GRAPH, epilogues SAR_EPI_ADD / SAR_EPI_ADD_GATE): `aux` holds EVEN output frames only -- [M][ld_aux] with column
 * (b * Ta + t / 2) * V + v for even t, Ta = (T_out + 1) / 2; odd output frames add nothing.  The skip gradient that reaches a block
 * through its strided 1x1 residual convolution (models/stgcn.py:47-56, stride 2) is non-zero on even frames only: the engines
 * compute it as a dense 1x1 product on the To frames and never materialise (or re-read) the zeros. */
#define SAR_GRAPH_AUX_EVEN_FRAMES 32
typedef struct sar_conv_desc {
  int32_t mode;        /* SAR_CONV_* */
  int32_t transposed;  /* TEMPORAL only */
  int32_t B, V;        /* sequences (N*M), joints */
  int32_t T_src, T_out;/* frames of src / out */
  int32_t Kc, M;       /* src channels (reduction), out channels */
  int32_t taps, stride, pad;
  int32_t pro_relu;
  int32_t epi;
  int32_t nz[3];       /* GRAPH: gather-list length per adjacency slice (<=4) */
  int32_t g_flags;     /* GRAPH: SAR_GRAPH_* bits, 0 = none */
  int32_t reserved0;   /* 0 */
  const float* src; int64_t ld_src;
  float* out; int64_t ld_out;
  const float* W;          /* element (tap, c, m) at tap*w_stride_tap + c*w_stride_c + m (m contiguous) */
  int64_t w_stride_tap, w_stride_c;
  const float* bias;       /* TEMPORAL: [M]; GRAPH: [taps][M] (scaled by colsum(A_k)[w]); may be NULL */
  const float* pro_scale; const float* pro_shift;   /* [Kc] or NULL */
  const int32_t* g_idx;    /* GRAPH: [taps][V][4] source joint of each gather entry */
  const float* g_wt;       /* GRAPH: [taps][V][4] weight of each gather entry */
  const float* g_colsum;   /* GRAPH: [taps][V] column sums of A_k (bias term) */
  const float* aux; int64_t ld_aux;                  /* epilogue operand [M][ld_aux] */
  const float* aux_scale; const float* aux_shift;    /* [M] (SAR_EPI_MASK) */
  const float* aux_mean;   /* [M] or NULL: centre of the second MASK reduction */
  float* partials;         /* [M][nparts][2] (SAR_EPI_STATS / SAR_EPI_MASK / SAR_EPI_ADD_GATE) */
  const void* aux2; int64_t ld_aux2;                 /* SAR_EPI_ADD_GATE: tensor of the second reduction: CN8 [ceil(M/8)][ld_aux2] units / fp32 [M][ld_aux2] */
  const unsigned char* aux_mask;                     /* SAR_EPI_ADD_GATE: gate bytes: CN8 [ceil(M/8)][ld_aux2] / fp32 [M][ld_aux2 / 4] */
} sar_conv_desc;

/* sizeof(sar_conv_desc) (which=0) / sizeof(sar_wgrad_desc) (1) / sizeof(sar_conv2d_desc) (2) as compiled: lets a binding verify its mirror */
int sar_struct_size(int which);
int sar_conv_gemm_nparts(const sar_conv_desc* d);                 /* host query, no GPU work */
int sar_conv_gemm_f32(const sar_conv_desc* d, sar_stream_t s);
/* The same operators with bf16 MFMA operands: W and the src-side operand -- pro(src) for
 * TEMPORAL, z_k = pro(src) . A_k (formed in fp32) for GRAPH -- are rounded to bfloat16 (nearest-even) as they are
 * staged, products are exact, accumulation / bias / epilogue / BatchNorm sums are fp32; src, out and W stay fp32 in
 * memory.  Needs M % 8 == 0.  workspace: sar_conv_gemm_bf16_workspace_bytes(d) bytes, 16-byte aligned (holds the packed
 * bf16 weights, rewritten by every call).  Partial-sum layout: sar_conv_gemm_nparts. */
int64_t sar_conv_gemm_bf16_workspace_bytes(const sar_conv_desc* d);
int sar_conv_gemm_bf16(const sar_conv_desc* d, void* workspace, sar_stream_t s);
/* Packing many weight tensors in ONE launch (a training step packs every layer, in both orientations, from the flat
 * parameter buffer): item i reads element (tap, c, m) at base[src_off + tap*st + c*sc + m*sm] -- a data gradient is the
 * same tensor with sc and sm exchanged, no transposed copy needed -- and writes the image sar_conv_gemm_bf16 expects
 * at out + 16*dst_unit bytes (taps * G * M units, G = 2*ceil(Kc/16)).  A descriptor with W == NULL makes
 * sar_conv_gemm_bf16 take `workspace` as such an already packed image.  `items` is a DEVICE array. */
typedef struct sar_pack_item {
  int64_t src_off, st, sc, sm, dst_unit;
  int32_t taps, Kc, M, G;
} sar_pack_item;
int sar_pack_weights_bf16_batch(const float* base, const sar_pack_item* items, int nitems, int64_t max_units, void* out,
                                sar_stream_t s);

/* fp32-ACCURATE results on the bf16 / fp16 matrix pipe ("split" arithmetic, csrc/conv_gemm_split.hip; fp32 MFMA runs at 1/16 of
 * that pipe's rate).  Same operator, descriptor, epilogues, fp32 storage and partial-sum contract as sar_conv_gemm_f32 --
 * models/stgcn.py:29-36 and its data gradient, under the fp32 parity tolerances.  `arith`:
 *   SAR_SPLIT_BF16X6  every fp32 operand = three bfloat16 terms (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 24
 *                     significant bits, remainders exact), the six cross products with i + j <= 2, each exact in the fp32
 *                     accumulator of v_mfma_f32_32x32x16_bf16 (dropped products <= 3 * 2^-27 |a b|).  No range limits.
 *   SAR_SPLIT_F16X3A  (the product arithmetic of the f32_split engine) fp16 terms of the operands scaled by powers of two, three
 *                     products, half the matrix work of X6: the well-conditioned operand (W; in the weight gradient the
 *                     BatchNorm-ed src) as w0 = fp16(s w), w1 = fp16(s w - w0) and w0 2^-11, the wide-range operand (activations,
 *                     gradients) as x0 = fp16(t x) and x1' = fp16((t x - x0) 2^11); acc += w0 x0 + w1 x0 + (w0 2^-11) x1'
 *                     (dropped: w1 x1 <= 2^-24 |w x|).  x keeps 22 significant bits over 2^29 of dynamic range below its
 *                     bound.  The scale of each operand comes from an
 *                     UPPER BOUND of its magnitudes that the caller keeps in device memory (`src_bound` for pro(src), `w_bound`
 *                     for W: the BITS of a non-negative float; sar_amax_f32 / sar_bn_bound_f32 / sar_affine_bound_f32 /
 *                     sar_pack_weights_split_batch produce them without a host sync): s = 2^(14 - floor(log2(bound))).  Values
 *                     beyond the bound saturate at the fp16 maximum (a stale bound gives a wrong, finite result).  A cell that
 *                     holds the bits of Inf / NaN -- the producers raise cells by UNSIGNED maxima of the value bits, so a
 *                     non-finite element, BatchNorm parameter or weight ends up there -- makes EVERY output of the launch NaN:
 *                     what the fp32 kernels and the reference's framework ops propagate from a non-finite operand, instead of
 *                     finite values out of the clamp (csrc/split_scale.h; tests/test_gpu_split.py).
 *   X1 / X3 / X9 / F16X3 / F16X3S (two symmetric fp16 terms: loses the low term of elements 2^18 below the bound -- not enough
 *   for gradient tensors) are measured data points (tools/split_probe.py, profiles/r05_split_probe_*; 9-tap temporal only).
 * Built for (i) the 9-tap TEMPORAL operator at V = 25, stride 1 / 2, 8 <= Kc <= 256, M % 8 == 0 (forward, and transposed = its
 * data gradient), every arithmetic; and (ii) the GRAPH operator (models/gcn.py:199-209 and, with the transposed tables, its data
 * gradient) at V = 25, taps = 3, T_src == T_out, 16 <= Kc <= 256 with Kc % 16 == 0, M % 8 == 0, no folded prologue, gather tables
 * flagged SAR_GRAPH_FEW_DENSE with at most 16 non-trivial lists, in SAR_SPLIT_BF16X6 / SAR_SPLIT_F16X3A only (F16X3A: two kernels
 * with bit-identical results -- persistent workgroups with the raw source by LDS-DMA for the data gradients, one tile per
 * workgroup for the forward launches; SAR_GRAPH_ONE_TILE_WG forces the latter); and (iii, round 6) the 1-tap TEMPORAL operator -- the
 * strided 1x1 residual convolution of models/stgcn.py:47-54, and, with W^T, the dense 1x1 product of its data gradient -- in its
 * forward form (transposed = 0) at V = 25, stride 1 / 2, pad 0, Kc >= 16, M % 8 == 0, no folded prologue, epilogues NONE / STATS /
 * MASK / ADD, in SAR_SPLIT_BF16X6 / SAR_SPLIT_F16X3A only.  Anything else returns SAR_E_UNSUP
 * (sar_conv_gemm_split_nparts too): the caller keeps sar_conv_gemm_f32 for it.  `packed` = the weight term images written by
 * sar_pack_weights_split_batch (same items as sar_pack_weights_bf16_batch, but G = ceil(Kc / 8), and an item occupies
 * sar_conv_gemm_split_workspace_bytes / 16 units = terms * taps * G * M); item_amax[nitems] receives each item's amax bits (fp16
 * arithmetics; = the item's w_bound).  Partial sums: [M][sar_conv_gemm_split_nparts][2]. */
enum { SAR_SPLIT_BF16X1 = 1, SAR_SPLIT_BF16X3 = 3, SAR_SPLIT_BF16X6 = 6, SAR_SPLIT_BF16X9 = 9, SAR_SPLIT_F16X3 = 103, SAR_SPLIT_F16X3S = 104, SAR_SPLIT_F16X3A = 105 };
int64_t sar_conv_gemm_split_workspace_bytes(const sar_conv_desc* d, int arith);
int sar_conv_gemm_split_nparts(const sar_conv_desc* d);
int sar_pack_weights_split_batch(const float* base, const sar_pack_item* items, int nitems, int64_t max_units, int arith,
                                 void* out, uint32_t* item_amax, sar_stream_t s);
int sar_conv_gemm_split(const sar_conv_desc* d, int arith, const void* packed, const uint32_t* src_bound, const uint32_t* w_bound,
                        sar_stream_t s);
/* Operand bounds for the fp16 arithmetics.  A cell is one uint32 in device memory holding the bits of a non-negative float;
 * every function below RAISES it (atomic max: order-independent, deterministic), the caller zeroes it first.
 *   sar_amax_f32:          cell = max(cell, max |x|) over a [C][n] matrix with row stride ld
 *   sar_bn_bound_f32:      cell = max(cell, max_c |gamma_c| sqrt(count - 1) + |beta_c|): bounds |gamma (x - mean) rstd + beta| for
 *                          ANY data normalised by its own batch statistics over `count` samples (Samuelson's inequality) -- the
 *                          train-mode BatchNormalization + ReLU of models/stgcn.py:27-28 folded into an operand load
 *   sar_affine_bound_f32:  cell = max(cell, max_c |scale_c| * src_cell + max_c |shift_c|): any folded affine (eval-mode BN) */
int sar_amax_f32(const float* x, int C, int64_t n, int64_t ld, uint32_t* cell, sar_stream_t s);
int sar_bn_bound_f32(const float* gamma, const float* beta, int C, double count, uint32_t* cell, sar_stream_t s);
int sar_affine_bound_f32(const float* scale, const float* shift, int C, const uint32_t* src_cell, uint32_t* cell, sar_stream_t s);

/* Weight gradient of the same operator (reduction over all positions n):
 *   dW[tap][c][m] = sum_n dout[m, n] * OP_tap(pro(src))[c, n]        (tf.GradientTape of the conv,
 *   dbias                                                             main_gnn.py:233)
 * Written as per-split partial slabs slab[split][wsize + bsize] (deterministic; no atomics), element
 * (tap,c,m) at tap*w_stride_tap + c*w_stride_c + m, bias partials at wsize + [..]; reduce with
 * sar_slab_reduce_f32.  bias slab: TEMPORAL [M]; GRAPH [taps][M] (= sum_n dout*colsum(A_k)[w(n)]). */
typedef struct sar_wgrad_desc {
  int32_t mode;
  int32_t B, V;
  int32_t T_src, T_out;     /* frames of src / dout */
  int32_t Kc, M;
  int32_t taps, stride, pad;
  int32_t pro_relu;
  int32_t nz[3];
  int32_t nsplit;           /* number of slabs (grid.x) */
  int32_t g_flags;          /* GRAPH: SAR_GRAPH_* bits, 0 = none */
  const float* src; int64_t ld_src;
  const float* dout; int64_t ld_dout;
  const float* pro_scale; const float* pro_shift;
  const int32_t* g_idx; const float* g_wt; const float* g_colsum;
  int64_t w_stride_tap, w_stride_c;  /* element strides inside the weight tensor */
  int64_t wsize, bsize;              /* floats per slab = wsize + bsize */
  float* slab;                       /* [nsplit][wsize + bsize] */
} sar_wgrad_desc;

int sar_conv_wgrad_f32(const sar_wgrad_desc* d, sar_stream_t s);
/* The same slabs for the 9-tap TEMPORAL operator with bf16 MFMA operands (both operands rounded to bfloat16 as they
 * are staged, exact products, fp32 accumulation; the bias gradient is summed from the fp32 values).  Built for V = 25
 * at stride 1, and at stride 2 with pad 3 (TF-SAME padding of an even T); other shapes: sar_conv_wgrad_f32. */
int sar_conv_wgrad_bf16(const sar_wgrad_desc* d, sar_stream_t s);
/* out[i] = sum_s slab[s*slab_stride + i] (i < n), summed in split order. */
int sar_slab_reduce_f32(const float* slab, int nsplit, int64_t slab_stride, int64_t n, float* out, sar_stream_t s);
/* The same reduction for a BATCH of weight gradients in ONE launch (round 6: a train step reduced its 20-22 slab sets by as many
 * launches behind their weight-gradient kernels -- 3-4 % of the fp32-storage steps, tools/skip_probe.py): items = DEVICE table,
 * every out[i] is summed exactly as sar_slab_reduce_f32 sums it (bit-identical); max_n = the largest n of the batch.  The engines
 * issue one call per gradient bucket (main_gnn.py:234,239: the slice's all-reduce follows it). */
typedef struct sar_slab_item {
  const float* slab;        /* [nsplit][slab_stride] */
  float* out;               /* [n] */
  int64_t slab_stride, n;
  int32_t nsplit, reserved;
} sar_slab_item;
int sar_slab_reduce_batch_f32(const sar_slab_item* items, int nitems, int64_t max_n, sar_stream_t s);
/* The weight / bias gradient of the same operator in the split arithmetics SAR_SPLIT_BF16X6 / SAR_SPLIT_F16X3A
 * (csrc/conv_wgrad_split.hip; SAR_SPLIT_BF16X6 / SAR_SPLIT_F16X3A): same descriptor and slab contract as sar_conv_wgrad_f32 (slabs reduced by sar_slab_reduce_f32),
 * src_bound / dout_bound = bound cells of pro(src) / dout for the fp16 arithmetic.  Built for the 9-tap TEMPORAL operator at
 * V = 25, stride 1 or stride 2 with pad 3 and T_src = 2 T_out, 8 <= Kc <= 256, and for the GRAPH operator at V = 25, 16 <= Kc <= 256
 * without a folded prologue, and (round 6) for the 1-tap TEMPORAL operator (the residual 1x1 convolution's weight / bias gradient) at
 * V = 25, pad 0, T_src = stride T_out, Kc >= 16, no folded prologue (wk 1; tiles of 128 flat positions, 64 x 128-channel blocks);
 * other shapes: SAR_E_UNSUP (keep sar_conv_wgrad_f32).  nsplit must be a multiple of the wk
 * that sar_conv_wgrad_split_blocks reports -- TEMPORAL: 1, or 2 at M <= 64 (two wave pairs of a workgroup split a tile's k-steps and
 * write two slabs); GRAPH: 1 at M >= 256, 2 at M = 128, 4 at M <= 64 (the waves that do not own an M block split the k-steps);
 * that query returns the weight blocks per slab group (or SAR_E_UNSUP) and the positions per tile: a launch has
 * (nsplit / wk) * blocks workgroups, two resident per CU. */
int sar_conv_wgrad_split_blocks(const sar_wgrad_desc* d, int arith, int* wk, int* tile_positions);
int sar_conv_wgrad_split(const sar_wgrad_desc* d, int arith, const uint32_t* src_bound, const uint32_t* dout_bound, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Batch-norm plumbing (Keras BatchNormalization(axis=1), models/stgcn.py:27,37,56,111).
 * partials: [C][nparts][2] (sum, sum of squares); statistics reduced in fp64.
 *   mean = S1/count ; var = S2/count - mean^2 (biased) ; rstd = 1/sqrt(var+eps)
 *   scale = gamma*rstd ; shift = beta - mean*scale
 *   running_mean = momentum*rm + (1-momentum)*mean ; running_var likewise with var*count/(count-1)
 *   when unbiased_running != 0 (fused NCHW path) -- skipped when running_mean == NULL.
 * ------------------------------------------------------------------------------------------------ */
int sar_bn_finalize_f32(const float* partials, int nparts, int C, double count, float eps, float momentum,
                        int unbiased_running, const float* gamma, const float* beta,
                        float* running_mean, float* running_var,
                        float* mean, float* rstd, float* scale, float* shift, sar_stream_t s);
/* inference affine from the moving statistics (training=False path, main_gnn.py:207). */
int sar_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, int C, float* scale, float* shift, sar_stream_t s);
/* Backward reductions -> coefficients.  partials [C][nparts][2] = (sum dz, sum dz*x).
 *   dgamma = rstd*(S2 - mean*S1) ; dbeta = S1 ; and  dx = k1*dz + k2*x + k3  with
 *   k1 = gamma*rstd, k2 = -gamma*rstd^2*b, k3 = gamma*rstd*(mean*rstd*b - a), a = S1/count, b = dgamma/count.
 * partial_stride: floats between consecutive channels; partial_step: floats between consecutive parts. */
int sar_bn_bwd_finalize_f32(const float* partials, int nparts, int64_t chan_stride, int64_t part_stride,
                            int off1, int off2, int centered, int C, double count,
                            const float* gamma, const float* mean, const float* rstd,
                            float* dgamma, float* dbeta, float* k1, float* k2, float* k3, sar_stream_t s);

/* data_bn, models/stgcn.py:142-147: x (N,C,T,V,M) contiguous -> CN activation [C][ld] with
 * b = n*M+m; BN channel = v*C+c, statistics over (n,m,t).  Optional fused joint->bone transform
 * (data_gen/gen_bone_data.py:36-41): bone_parent[v] = v2 (or -1 for none) subtracts joint v2.  motion != 0 feeds
 * the motion stream (data_gen/gen_motion_data.py:24-27): frame t+1 minus frame t of the joint / bone data, last frame 0. */
int sar_data_bn_stats_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                          float* partials /* [V*C][N][2] */, sar_stream_t s);
int sar_data_bn_apply_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                          const float* scale, const float* shift, float* out, int64_t ld_out, sar_stream_t s);
/* backward: partials [V*C][N][2] = (sum dy, sum dy*(x_raw - mean[ch])) from dy in CN layout (mean may be NULL). */
int sar_data_bn_bwd_reduce_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                               const float* dy, int64_t ld_dy, const float* mean, float* partials, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Block tail, models/stgcn.py:37,62-63:  y = relu(u*sc[c]+sh[c] + res)
 *   res_kind 0: none ; 1: identity (res = r) ; 2: conv residual (res = r*rsc[c]+rsh[c])
 * ------------------------------------------------------------------------------------------------ */
int sar_bn_add_relu_fwd_f32(const float* u, const float* sc, const float* sh, int res_kind, const float* r,
                            const float* rsc, const float* rsh, float* y, int C, int64_t n, int64_t ld, sar_stream_t s);
/* backward pass 1: dz = (y>0)?dy:0 ; partials[C][nparts][4] = (sum dz, sum dz*(u-mu[c]), sum dz*(r-mr[c]), 0);
 * mu / mr (the batch means) may be NULL = 0 */
int sar_bn_add_relu_bwd_reduce_f32(const float* dy, const float* y, const float* u, const float* r,
                                   const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n, int64_t ld, sar_stream_t s);
/* The block tail with a 1-bit ReLU mask (fp32 rows of 4-element groups: n and ld multiples of 4, 16-byte aligned pointers, else
 * SAR_E_UNSUP): the forward also writes mask[c][i / 4] (one byte per float4 of y, bit j = element j > 0; ld / 4 bytes per row)
 * and the two backward passes read it instead of y: the same results bit for bit, a third less HBM traffic in the reduce. */
int sar_bn_add_relu_fwd_mask_f32(const float* u, const float* scale, const float* shift, int res_kind, const float* r,
                                 const float* res_scale, const float* res_shift, float* y, void* mask, int C, int64_t n, int64_t ld,
                                 sar_stream_t s);
int sar_bn_add_relu_bwd_reduce_mask_f32(const float* dy, const void* mask, const float* u, const float* r,
                                        const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                        sar_stream_t s);
int sar_bn_add_relu_bwd_apply_mask_f32(const float* dy, const void* mask, const float* u, const float* r,
                                       const float* k1, const float* k2, const float* k3,
                                       const float* rk1, const float* rk2, const float* rk3,
                                       float* du, float* dr, float* dz_out, int C, int64_t n, int64_t ld, sar_stream_t s);
/* The same reduction (gradient of `relu(bn(u) + res)`, models/stgcn.py:37,56,62-63 and models/resnet18.py:46-63 under
 * main_gnn.py:233 / loss.backward()) with the BatchNorm-backward finalisation folded in ("tail"): the LAST workgroup of every channel to
 * finish (an integer ticket per channel, agent-scope release / acquire; no float atomics, every sum in a fixed order) adds the
 * channel's partials in fp64 and writes what sar_bn_bwd_finalize_f32(centered) would: dgamma, dbeta, k1, k2, k3 of the
 * block's BatchNorm and, when r != NULL, of the residual branch's -- one small dependent launch (or two) less on the critical
 * chain of every block.  ticket: C (f32) / ceil(C/8) (cn8) ints, zero before the first use; the kernel leaves them zero. */
typedef struct sar_bn_tail {
  int32_t* ticket;
  double count;                                   /* elements per channel (N*T*V) */
  const float* gamma; const float* rstd;          /* BatchNorm of u (its mean is the reduce call's mu) */
  float* dgamma; float* dbeta; float* k1; float* k2; float* k3;
  const float* rgamma; const float* rrstd;        /* BatchNorm of the residual branch (r != NULL), mean = mr */
  float* rdgamma; float* rdbeta; float* rk1; float* rk2; float* rk3;
} sar_bn_tail;
int sar_bn_add_relu_bwd_reduce_tail_f32(const float* dy, const float* y, const float* u, const float* r,
                                        const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                        const sar_bn_tail* tail, sar_stream_t s);
/* backward pass 2: du = k1*dz+k2*u+k3 ; dr = rk1*dz+rk2*r+rk3 (if dr != NULL) ; dz_out = dz (if != NULL) */
int sar_bn_add_relu_bwd_apply_f32(const float* dy, const float* y, const float* u, const float* r,
                                  const float* k1, const float* k2, const float* k3,
                                  const float* rk1, const float* rk2, const float* rk3,
                                  float* du, float* dr, float* dz_out, int C, int64_t n, int64_t ld, sar_stream_t s);
/* The same passes with an operand-bound by-product for the split arithmetic (cells: see sar_amax_f32): amax_y / amax_du /
 * amax_out is RAISED to the largest |value| the pass wrote to y / du / out -- what a separate sar_amax_f32 pass over that tensor
 * would give, without reading it again (28 such passes cost 2.1 ms of a 39 ms f32_split step). */
int sar_bn_add_relu_fwd_mask_amax_f32(const float* u, const float* scale, const float* shift, int res_kind, const float* r,
                                      const float* res_scale, const float* res_shift, float* y, void* mask, uint32_t* amax_y,
                                      int C, int64_t n, int64_t ld, sar_stream_t s);
int sar_bn_add_relu_bwd_apply_mask_amax_f32(const float* dy, const void* mask, const float* u, const float* r,
                                            const float* k1, const float* k2, const float* k3,
                                            const float* rk1, const float* rk2, const float* rk3,
                                            float* du, float* dr, float* dz_out, uint32_t* amax_du, uint32_t* amax_dr, int C, int64_t n,
                                            int64_t ld, sar_stream_t s);   /* amax_dr (may be NULL): the same for dr -- the operand of the residual branch's dense 1x1 data gradient (round 6) */
int sar_affine2_amax_f32(const float* a, const float* b, const float* k1, const float* k2, const float* k3,
                         float* out, uint32_t* amax_out, int C, int64_t n, int64_t ld, sar_stream_t s);
/* generic row-affine: out = k1[c]*a + k2[c]*b + k3[c]  (BN backward apply: dg from dz1 and g) */
int sar_affine2_f32(const float* a, const float* b, const float* k1, const float* k2, const float* k3,
                    float* out, int C, int64_t n, int64_t ld, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Head, models/stgcn.py:153-158: GlobalAveragePooling2D over (T,V), mean over the M persons,
 * 1x1 Conv2D logits.  y: CN [C][ld], B = N*Mp sequences of TV positions.
 * ------------------------------------------------------------------------------------------------ */
int sar_pool_fwd_f32(const float* y, int64_t ld, int C, int B, int TV, int Mp, float* feat /* [N][C] */, sar_stream_t s);
int sar_fc_fwd_f32(const float* feat, const float* W /* [C][K] */, const float* bias, int N, int C, int K,
                   float* logits /* [N][K] */, sar_stream_t s);
/* main_gnn.py:224-226: loss_i = CE(logits_i, label_i); dlogits = (softmax - onehot)*inv_global_batch.
 * loss_sum[0] receives sum_i loss_i * inv_global_batch (written, not accumulated). */
int sar_softmax_ce_f32(const float* logits, const int64_t* labels, int N, int K, float inv_global_batch,
                       float* loss_sum, float* dlogits, float* probs /* may be NULL */, sar_stream_t s);
/* dW[c][k] = sum_n feat[n][c] dlogits[n][k], dbias[k] = sum_n dlogits[n][k], dfeat = dlogits W^T.  dW and dbias may both be NULL (only
 * dfeat, the launch on the backward chain), or dfeat (only the parameter gradients: they feed nothing but the optimizer). */
int sar_fc_bwd_f32(const float* feat, const float* W, const float* dlogits, int N, int C, int K,
                   float* dW, float* dbias, float* dfeat, sar_stream_t s);
/* dy[c,(b,tv)] = dfeat[b/Mp][c] / (Mp*TV) */
int sar_pool_bwd_f32(const float* dfeat, int64_t ld, int C, int B, int TV, int Mp, float* dy, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Optimizer, main_gnn.py:312-314 (tf.keras SGD momentum 0.9 nesterov):
 *   v <- momentum*v - lr*g ; w <- w + momentum*v - lr*g      over a flat parameter buffer.
 * lr is read from device memory (lr_dev[0]) so a captured graph can be replayed across LR changes.
 * ------------------------------------------------------------------------------------------------ */
int sar_sgd_nesterov_f32(float* w, float* v, const float* g, int64_t n, const float* lr_dev, float momentum,
                         sar_stream_t s);
/* batched 2-D transpose: in [batch][R][Cc] -> out [batch][Cc][R] (weight re-layout for data gradients) */
int sar_transpose_f32(const float* in, float* out, int batch, int R, int Cc, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * VirtualRadar, layers/virtual_radar.py:79-134.
 * sar_vr_signal_f32  (:93-123): x (B,3,T,V,M) contiguous -> complex baseband z[b][t] (re, im planes).
 *   edges given as src/dst joint index arrays (E entries); loc = radar_location (3 floats, device),
 *   wavelength (1 float, device).  fp32 arithmetic in the reference's op order; accurate sin/cos.
 * sar_stft_logmag_f32 (:124-133 + nnAudio 0.1.1 STFT): reflect-pad n_fft/2, periodic-Hann windowed
 *   DFT of length n_fft every hop samples, out[b][(k + n_fft/2) % n_fft][f] = log(|Z[k,f]| + 1e-6).
 *   If out_cols > 0 only the frames  f = floor(j*F/out_cols), j < out_cols  are produced (nearest
 *   F.interpolate of models/resnet.py:26 fused as a column select) and out is [B][n_fft][out_cols].
 * ------------------------------------------------------------------------------------------------ */
int sar_vr_signal_f32(const float* x, int B, int T, int V, int M, const int32_t* e_src, const int32_t* e_dst, int E,
                      const float* loc, const float* wavelength, float* z_re, float* z_im, sar_stream_t s);
int sar_stft_logmag_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                        const float* window /* [n_fft] */, int out_cols, float* out, sar_stream_t s);
/* Backward of the two stages, for training radar_location / wavelength (layers/virtual_radar.py:46-52 with
 * train_* = True; main_spectrogram.py:133-136 unfreezes `radar_loc*` after --loc-train-epoch): torch autograd
 * of :93-133 in the reference.
 * sar_stft_logmag_bwd_f32: dout [B][n_fft][F or out_cols] (gradient of sar_stft_logmag_f32's output) ->
 *   dz_re, dz_im [B][T] (cotangent of the complex signal).  Recomputes the spectrum from z; workspace of
 *   sar_stft_logmag_bwd_workspace_floats(...) floats; fixed summation order (no atomics).
 * sar_vr_signal_bwd_f32: partials[nparts][4] with sum over parts = d loss / d (loc_x, loc_y, loc_z, wavelength)
 *   = sum_{b,t} Re(conj(dz) * dz/dp); nparts = sar_vr_signal_bwd_nparts(B, T).  Forward-mode tangents of the
 *   geometry / RCS / phase, contracted with dz per (clip, frame). */
int64_t sar_stft_logmag_bwd_workspace_floats(int B, int T, int n_fft, int hop);
int sar_stft_logmag_bwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                            const float* window, int out_cols, const float* dout, float* workspace,
                            float* dz_re, float* dz_im, sar_stream_t s);
/* Trainable Fourier kernels -- layers/virtual_radar.py:71-76 `train_stft_kernel` (nnAudio STFT(trainable=True): the
 * conv1d kernels wcos / wsin [n_fft][1][n_fft], window folded in, are Parameters).  Same output as sar_stft_logmag_f32 but
 * the transform is the matrix product with the CURRENT kernels:
 *   Z_re[k] = sum_n fr[n] wcos[k][n] + fi[n] wsin[k][n],  Z_im[k] = sum_n fi[n] wcos[k][n] - fr[n] wsin[k][n].
 * wcosT / wsinT are the [n][k] transposes (sar_transpose_f32).  n_fft <= 1024.
 * sar_stft_kernels_bwd_f32: dw[2][n_fft][n_fft] = (d wcos, d wsin) and, when dz_re / dz_im != NULL, the cotangent of the
 * signal for sar_vr_signal_bwd_f32.  workspace: sar_stft_kernels_bwd_workspace_floats floats; the kernel gradient is
 * reduced over nsplit slabs in a fixed order (no atomics). */
int sar_stft_kernels_fwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop, const float* wcosT,
                             const float* wsinT, int out_cols, float* out, sar_stream_t s);
int64_t sar_stft_kernels_bwd_workspace_floats(int B, int T, int n_fft, int hop, int nsplit);
int sar_stft_kernels_bwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop, const float* wcos,
                             const float* wsin, const float* wcosT, const float* wsinT, int out_cols, const float* dout,
                             float* workspace, int nsplit, float* dw, float* dz_re, float* dz_im, sar_stream_t s);
int sar_vr_signal_bwd_nparts(int B, int T);
int sar_vr_signal_bwd_f32(const float* x, int B, int T, int V, int M, const int32_t* e_src, const int32_t* e_dst,
                          int E, const float* loc, const float* wavelength, const float* dz_re, const float* dz_im,
                          float* partials, sar_stream_t s);

/* Frame-rate up-sampling of utils.py:134-140 (Dataset.pad_frames, what main_spectrogram.py feeds the radar:
 * gaussian_filter1d(sigma) along T, then interp1d 'cubic' evaluated at np.linspace(0, 1, P*T), P = num_pad_frames = 250)
 * fused with the radar signal: the (B,3,P*T,V,M) tensor is never materialised.
 * sar_upsample_prepare_f64: x (B,3,T,V,M) -> per-interval cubic pieces coef [B][T-1][3][V*M][4] (float64;
 *   sar_upsample_coef_doubles elements).  weights = the radius+1 one-sided Gaussian weights w[0..radius] (float64,
 *   device; scipy's _gaussian_kernel1d, normalised), workspace of sar_upsample_workspace_bytes(...) bytes.  T >= 5.
 * sar_vr_signal_upsampled_f32 / _bwd_f32: sar_vr_signal_f32 / sar_vr_signal_bwd_f32 on the up-sampled clip (P*T frames;
 *   nparts = sar_vr_signal_bwd_nparts(B, P*T)), each frame evaluated from coef in float64 and rounded to float32. */
int64_t sar_upsample_workspace_bytes(int B, int T, int V, int M);
int64_t sar_upsample_coef_doubles(int B, int T, int V, int M);
int sar_upsample_prepare_f64(const float* x, int B, int T, int V, int M, const double* weights, int radius,
                             void* workspace, double* coef, sar_stream_t s);
int sar_vr_signal_upsampled_f32(const double* coef, int B, int T, int P, int V, int M, const int32_t* e_src,
                                const int32_t* e_dst, int E, const float* loc, const float* wavelength, float* z_re,
                                float* z_im, sar_stream_t s);
int sar_vr_signal_upsampled_bwd_f32(const double* coef, int B, int T, int P, int V, int M, const int32_t* e_src,
                                    const int32_t* e_dst, int E, const float* loc, const float* wavelength,
                                    const float* dz_re, const float* dz_im, float* partials, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * ResNet-18 of the spectrogram path, models/resnet18.py:131-254 (torch Conv2d bias=False / BatchNorm2d /
 * MaxPool2d / Linear) on the CN layout: an image batch (B,C,H,W) is the matrix [C][B*H*W].
 *
 * sar_conv2d_gemm_f32: out[m,(b,ho,wo)] = sum_{kh,kw,c} W[tap][c][m] * pro(src)[c,(b, ho*s+kh-pad, wo*s+kw-pad)]
 *   (nn.Conv2d, models/resnet18.py:5-23,159-164); transposed=1 is its data gradient (out at the conv's INPUT
 *   resolution, src = gradient at its output resolution).  Supported: 3x3 and 1x1, stride 1|2; the 7x7/2 stem
 *   (1 input channel, models/resnet18.py:159-164) forward.  W is the repacked weight: element (tap=kh*KW+kw, c, m)
 *   at tap*w_stride_tap + c*w_stride_c + m.  pro / epi / aux / partials as in sar_conv_gemm_f32.
 *   The data gradient of a 3x3 / stride 2 / pad 1 layer whose input is exactly twice its output (every stride-2 layer of
 *   resnet18) runs as four parity-class launches (output pixels (y&1, x&1) use 1, 2, 2, 4 of the 9 taps): 9 tap products per
 *   4 pixels instead of 36.  flags & SAR_C2D_AUX_EVEN_PIXELS (that geometry, epi = ADD): aux is a COMPACT tensor
 *   [M][B*H_src*W_src] added to the output pixels (2i, 2j) only -- the data gradient of the block's parallel 1x1 / stride 2
 *   down-sampling convolution (models/resnet18.py:92-100), which is non-zero exactly there and is computed at the small
 *   resolution by an ordinary stride-1 call.
 * sar_conv2d_wgrad_f32: dW[tap][c][m] = sum_n dout[m,n] * OP_tap(pro(src))[c,n] as per-split slabs laid out
 *   (tap, c, m) with m contiguous: slab[split][taps*Kc*M]; reduce with sar_slab_reduce_f32.
 * ------------------------------------------------------------------------------------------------ */
#define SAR_C2D_AUX_EVEN_PIXELS 1
typedef struct sar_conv2d_desc {
  int32_t transposed;
  int32_t B, Kc, M;
  int32_t H_src, W_src, H_out, W_out;
  int32_t KH, KW, stride, pad;
  int32_t pro_relu, epi;
  int32_t nsplit;          /* wgrad only */
  int32_t flags;           /* SAR_C2D_* bits, 0 = none */
  const float* src; int64_t ld_src;
  float* out; int64_t ld_out;                        /* gemm: output activations; wgrad: unused */
  const float* dout; int64_t ld_dout;                /* wgrad only */
  const float* W; int64_t w_stride_tap, w_stride_c;  /* gemm only */
  const float* pro_scale; const float* pro_shift;
  const float* aux; int64_t ld_aux; const float* aux_scale; const float* aux_shift; const float* aux_mean;
  float* partials;         /* gemm: [M][nparts][2] */
  float* slab;             /* wgrad: [nsplit][KH*KW*Kc*M] */
  sar_context* ctx;        /* optional side streams for launches that fan out (see sar_context); NULL = caller's stream only */
} sar_conv2d_desc;

int sar_conv2d_nparts(const sar_conv2d_desc* d);
int sar_conv2d_gemm_f32(const sar_conv2d_desc* d, sar_stream_t s);
int sar_conv2d_wgrad_f32(const sar_conv2d_desc* d, sar_stream_t s);
/* The 3x3 / stride-1 / pad-1 convolutions of the BasicBlocks (models/resnet18.py:5-14,37-61) and their data gradients with fp32
 * results on the fp16 / bf16 matrix pipe: the split arithmetic of sar_conv_gemm_split (SAR_SPLIT_F16X3A / SAR_SPLIT_BF16X6 only) on
 * the sar_conv2d_desc operator -- same descriptor, fp32 storage, epilogues (NONE / STATS / MASK / ADD) and partial-sum contract
 * ([M][sar_conv2d_gemm_split_nparts][2]) as sar_conv2d_gemm_f32; csrc/conv2d_split.hip.  `packed` = the term images written by
 * sar_pack_weights_split_batch from an item with taps = 9 (tap = kh * 3 + kw), G = ceil(Kc / 8).  A descriptor with transposed = 1
 * (stride 1: the data gradient is a convolution of dout with mirrored taps) takes the image of THAT view of the weights: item
 * element (tap, c, m) = W[8 - tap][m][c] (a negative tap stride, sc and sm exchanged) -- the kernel itself does not look at
 * `transposed`.  src_bound / w_bound: as sar_conv_gemm_split.  Built for H_src = H_out, W_src = W_out, 8 <= Kc <= 512, M % 8 == 0,
 * windows of at most 512 staged pixels ((rows per tile + 2) (W + 2), tiles of 256 output pixels), flags == 0; anything else
 * returns SAR_E_UNSUP (the nparts query too) and the caller keeps sar_conv2d_gemm_f32.  d->W and d->ctx are ignored. */
int64_t sar_conv2d_gemm_split_workspace_bytes(const sar_conv2d_desc* d, int arith);
 /* Small feature maps (a launch of <= 1 workgroup per CU with a long contraction: 8x8 x 512 channels) divide the channel stages among
 * up to 8 workgroups per output tile when the caller provides a workspace in d->slab (sar_conv2d_gemm_split_slab_bytes(d) bytes, 0 = no
 * K-split planned for this shape; 16-byte aligned; d->slab == NULL: one workgroup per tile): raw fp32 partial tiles, then a second
 * launch that sums them in a fixed order and applies the epilogue.  The partial-sum count (nparts) depends on it: query
 * sar_conv2d_gemm_split_nparts with d->slab set as it will be in the call. */
int64_t sar_conv2d_gemm_split_slab_bytes(const sar_conv2d_desc* d);
int sar_conv2d_gemm_split_nparts(const sar_conv2d_desc* d);
int sar_conv2d_gemm_split(const sar_conv2d_desc* d, int arith, const void* packed, const uint32_t* src_bound,
                          const uint32_t* w_bound, sar_stream_t s);
/* Weight gradient of the same convolutions on the split arithmetic (csrc/conv_wgrad_split.hip instantiated for images: the batch is
 * one flat sequence of B H W positions, a tap a shift of it; the image borders are handled by masking one element of a dout fragment /
 * redirecting a src fragment to a zero row).  Same descriptor fields and slab contract as sar_conv2d_wgrad_f32 (slab[nsplit][9 Kc M] in
 * (tap, c, m) order, summed by sar_slab_reduce_f32 in slab order); src_bound = bound of pro(src) (the well-conditioned operand: three term
 * images), dout_bound = bound of dout (two).  Built for 3x3 / stride 1 / pad 1, W in {8, 16, 32, 64}, Kc >= 8, M >= 8, B H W < 2^22;
 * anything else: SAR_E_UNSUP.  sar_conv2d_wgrad_split_blocks: host query -- workgroups per slab group (return value), *wk = slabs a
 * group writes (nsplit must be a multiple), *tile_positions = positions per tile (256). */
int sar_conv2d_wgrad_split_blocks(const sar_conv2d_desc* d, int arith, int* wk, int* tile_positions);
int sar_conv2d_wgrad_split(const sar_conv2d_desc* d, int arith, const uint32_t* src_bound, const uint32_t* dout_bound, sar_stream_t s);
/* Data gradient of a ONE-input-channel conv (the 7x7/2 stem, models/resnet18.py:159): dx[b][h][w] = sum_{kh,kw,m}
 * dout[m][(b, (h+pad-kh)/s, (w+pad-kw)/s)] * w_packed[kh*KW+kw][m] over the taps that divide evenly.  Needed only
 * when the image itself depends on trainable parameters (VirtualRadar location / wavelength). */
int sar_conv2d_stem_dgrad_f32(const float* dout, int64_t ld_dout, const float* w_packed, int B, int H, int W,
                              int H_out, int W_out, int M, int KH, int KW, int stride, int pad, float* dx,
                              sar_stream_t s);

/* out[i][j][k] (contiguous [d0][d1][d2]) = in[i*s0 + j*s1 + k*s2]: weight repacking OIHW <-> (tap, c, m). */
int sar_permute3_f32(const float* in, float* out, int d0, int d1, int d2, int64_t s0, int64_t s1, int64_t s2,
                     sar_stream_t s);
/* The same for many tensors in ONE launch (a training step re-packs every conv weight of the resnet into both operand
 * layouts, and every weight gradient back into OIHW): item i copies d0*d1*d2 elements, out[dst_off + (a*d1 + b)*d2 + c] =
 * in[src_off + a*s0 + b*s1 + c*s2].  `items` is a DEVICE array; max_elems = the largest d0*d1*d2. */
typedef struct sar_permute_item {
  int64_t src_off, dst_off, s0, s1, s2;
  int32_t d0, d1, d2, reserved;
} sar_permute_item;
int sar_permute3_batch_f32(const float* in, float* out, const sar_permute_item* items, int nitems, int64_t max_elems,
                           sar_stream_t s);
/* stem tail, models/resnet18.py:236-239: y = MaxPool2d(3,2,1)(relu(x*scale[c]+shift[c])); x [C][B*H*W] -> y [C][B*Ho*Wo]. */
int sar_bn_relu_maxpool_fwd_f32(const float* x, const float* scale, const float* shift, float* y, int C, int B, int H,
                                int W, int64_t ld_x, int64_t ld_y, sar_stream_t s);
/* its backward: dz[c,n] = gradient w.r.t. the BatchNorm output (ReLU mask applied, max routed to the first
 * maximal element of each window like torch); partials[C][nparts][2] = (sum dz, sum dz*(x-mean[c])). */
int sar_bn_relu_maxpool_bwd_nparts(int B, int H, int W);   /* partial sums per channel written by the backward */
int sar_bn_relu_maxpool_bwd_f32(const float* x, const float* scale, const float* shift, const float* mean,
                                const float* dy, float* dz, float* partials, int nparts, int C, int B, int H, int W,
                                int64_t ld_x, int64_t ld_y, sar_stream_t s);
/* torch.optim.Adam (main_spectrogram.py:106) over flat buffers; step_dev[0] = t (>= 1), lr_dev[0] = lr. */
int sar_adam_f32(float* w, float* m, float* v, const float* g, int64_t n, const float* lr_dev, const float* step_dev,
                 float beta1, float beta2, float eps, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Dense (trainable) adjacency, SURVEY.md 8(f)-4: the contraction of models/gcn.py:207-208 / 236-237 (AdjGraphConv) with
 * an arbitrary A (K, V, V) in device memory, fp32 CN layout, V <= 32, K <= 8.  y is the 3F-channel output of the 1x1
 * convolution (channel k*F + m; sar_conv_gemm_f32, TEMPORAL, taps = 1 -- bias included, as in the reference).
 *   fwd       out[m, (t,w)]      = sum_k sum_v y[k*F + m, (t,v)] * A[k, v, w]        (+ partials[F][nparts][2] = per-row
 *                                   (sum, sum of squares) per tile when partials != NULL; nparts = sar_graph_dense_nparts;
 *                                   + add[m, (t,w)] when add != NULL)
 *   bwd_data  dy[k*F + m, (t,v)] = sum_w dout[m, (t,w)] * A[k, v, w]
 *   dA        dA[k, v, w]        = sum_{m, frames} y[k*F + m, (f,v)] * dout[m, (f,w)]  (slab: sar_graph_dense_dadj_slab_floats
 *                                   floats of scratch; partial blocks are summed in a fixed order: deterministic)
 * nframes = B*T frames of V joints (the adjacency acts inside a frame).
 * ------------------------------------------------------------------------------------------------ */
int sar_graph_dense_nparts(int64_t nframes);
int sar_graph_dense_fwd_f32(const float* y, int64_t ld_y, const float* A, float* out, int64_t ld_out, int K, int F, int V,
                            int64_t nframes, float* partials, const float* add, int64_t ld_add, sar_stream_t s);
int sar_graph_dense_bwd_data_f32(const float* dout, int64_t ld_dout, const float* A, float* dy, int64_t ld_dy, int K, int F,
                                 int V, int64_t nframes, sar_stream_t s);
int64_t sar_graph_dense_dadj_slab_floats(int K, int F, int V, int nsplit);
int sar_graph_dense_dadj_f32(const float* y, int64_t ld_y, const float* dout, int64_t ld_dout, int K, int F, int V,
                           int64_t nframes, int nsplit, float* slab, float* dA, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Graph isomorphism convolution, SURVEY.md 8(f)-4 (models/gcn.py:112-163 GraphIsoConvTD as used by models/stgin.py:24-25):
 * fp32 CN layout; the K branch MLPs of a layer are stacked along the channel axis (row k*C + c).
 *   sar_gin_adjacency_f32   table[k][a][b] = A[k][b][a] (k < Km1), table[Km1] = (1 + eps[0]) I  -- models/gcn.py:150-153
 *                           A_ = concat(A, diag(1 + epsilon)), transposed for sar_graph_dense_bwd_data_f32 (x . A_k) and
 *                           sar_graph_dense_fwd_f32 (its gradient);  scale[0:C] = 1 + eps[0] (the prologue of the self slice);
 *                           slice_scale[0:Km1+1] = (1, .., 1, 1 + eps[0]) when != NULL; table may be NULL
 *   sar_graph_gather_*_f32  the same contractions for a FIXED sparse adjacency given as gather lists (idx / wt [K][V][4],
 *                           sar_amd/graph_tables.py -- what the ST-GCN kernels fold into their operand loads):
 *                           expand: out[k*F + m, (t,v)] = sum_j wt[k][v][j] * in[m, (t, idx[k][v][j])]
 *                           sum:    out[m, (t,w)] = sum_k scale[k] * sum_j wt[k][w][j] * in[k*F + m, (t, idx[k][w][j])] (+ add)
 *                           n = frames * V columns; HBM-bound (every tensor moved once).  nz_host: HOST array [K], entries per
 *                           joint used by slice k (1 for an identity slice; NULL = 4): idx / wt beyond it are not read
 *   sar_gin_sum_fwd_f32     s[c, n] = sum_k relu(a[k*C + c, n] * scale[k*C + c] + shift[k*C + c])  -- the last BN + ReLU of
 *                           every branch and tf.reduce_sum (models/gcn.py:139-142,160); partials[C][nparts][2] = (sum s,
 *                           sum s^2) per workgroup for the BatchNorm that follows (models/stgin.py:28), nparts = sar_gin_nparts(n)
 *   sar_gin_bwd_reduce_f32  dz = ds[c] where a*scale+shift > 0 else 0; partials[K*C][nparts][2] = (sum dz, sum dz*(a - mean))
 *   sar_gin_bwd_apply_f32   da[k*C + c] = k1*dz + k2*a + k3  (k1..k3 from sar_bn_bwd_finalize_f32; da may alias a)
 *   sar_gin_eps_grad_f32    the self slice's first convolution is W . ((1 + eps) x): given G = dout . x^T (weight gradient
 *                           taken on the un-scaled x), deps[0] = <G, W> and G *= (1 + eps[0]) in place
 * ------------------------------------------------------------------------------------------------ */
int sar_gin_nparts(int64_t n);
int sar_gin_adjacency_f32(const float* A, int Km1, int V, const float* eps, float* table, float* scale, int C, float* slice_scale,
                          sar_stream_t s);
int sar_gin_sum_fwd_f32(const float* a, int64_t ld_a, const float* scale, const float* shift, int K, int C, int64_t n, float* s_out,
                        int64_t ld_s, float* partials, sar_stream_t s);
int sar_gin_bwd_reduce_f32(const float* ds, int64_t ld_ds, const float* a, int64_t ld_a, const float* scale, const float* shift,
                           const float* mean, int K, int C, int64_t n, float* partials, sar_stream_t s);
int sar_gin_bwd_apply_f32(const float* ds, int64_t ld_ds, const float* a, int64_t ld_a, const float* scale, const float* shift,
                          const float* k1, const float* k2, const float* k3, int K, int C, int64_t n, float* da, int64_t ld_da,
                          sar_stream_t s);
int sar_gin_eps_grad_f32(float* G, const float* W, int64_t n, const float* eps, float* deps, sar_stream_t s);
int sar_graph_gather_sum_f32(const float* in, int64_t ld_in, const int32_t* idx, const float* wt, const int32_t* nz_host,
                             const float* scale, int K, int F, int V, int64_t n, float* out, int64_t ld_out, const float* add,
                             int64_t ld_add, sar_stream_t s);
int sar_graph_gather_expand_f32(const float* in, int64_t ld_in, const int32_t* idx, const float* wt, const int32_t* nz_host, int K,
                                int F, int V, int64_t n, float* out, int64_t ld_out, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * bf16 configuration (bf16 activations in HBM, bf16 MFMA operands, fp32 accumulation, fp32
 * BatchNorm statistics, fp32 master weights).
 *
 * Activation layout "CN8": an activation with C channels over n columns (column = (b*T + t)*V + v as in the CN layout)
 * is G = ceil(C/8) planes of ld >= n 16-byte units; unit (g, col) = the 8 bfloat16 channels 8g..8g+7 of that column,
 * channels >= C are zero.  Byte address of channel c, column n: 16*((c/8)*ld + n) + 2*(c%8).  All CN8 pointers are
 * 16-byte aligned; `ld` counts units per plane.  (This is the k-innermost MFMA operand image: csrc/cn8.h.)
 *
 * sar_conv_gemm_cn8: the operators of sar_conv_gemm_f32 (same descriptor; src / out / aux are CN8 tensors passed through
 * the float* fields, ld_* in units; W is ignored) with the weights given as the packed bf16 image of
 * sar_pack_weights_bf16_batch ([taps][2*ceil(Kc/16)][M] units).  Products of bf16 operands are exact, accumulation is
 * fp32, BatchNorm partial sums (sar_conv_gemm_cn8_nparts per row) are taken from the fp32 accumulators BEFORE the
 * result is rounded to bfloat16 (nearest even).  SAR_EPI_MASK decides on the stored (bf16) aux value.  GRAPH mode
 * forms z_k = src . A_k in fp32 from the bf16 src and rounds it once; no prologue in GRAPH mode.
 * ------------------------------------------------------------------------------------------------ */
int sar_conv_gemm_cn8_nparts(const sar_conv_desc* d);
int sar_conv_gemm_cn8(const sar_conv_desc* d, const void* packed_w, sar_stream_t s);
/* Weight / bias gradients with CN8 src / dout (passed through the float* fields of sar_wgrad_desc, ld_* in units): the
 * operators and the slab contract of sar_conv_wgrad_f32, bf16 products accumulated in fp32, bias sums in fp32 from the
 * stored values.  Built for V = 25: GRAPH (3 slices; slice0_identity != 0 promises that slice 0's gather list is
 * {(v, 1.0)}, i.e. A_0 = I, and lets the kernel read that operand straight from the staged tile), TEMPORAL with 9 taps
 * (stride 1; stride 2 with the TF-SAME pads 3 (even T) or 4 (odd T)) and 1 tap (stride 1 or 2, pad 0).  Tiles are sar_conv_wgrad_cn8_tile_frames(mode)
 * frames of one sequence; nsplit slabs as for sar_conv_wgrad_f32. */
int sar_conv_wgrad_cn8_tile_frames(int mode);
int sar_conv_wgrad_cn8(const sar_wgrad_desc* d, int slice0_identity, sar_stream_t s);
/* Block tail and BatchNorm-backward passes on CN8 tensors: semantics and partial layouts of the *_f32 functions of the
 * same name (C = channels, n = columns, ld = units per plane). */
int sar_bn_add_relu_fwd_cn8(const void* u, const float* scale, const float* shift, int res_kind, const void* r,
                            const float* res_scale, const float* res_shift, void* y, int C, int64_t n, int64_t ld,
                            sar_stream_t s);
int sar_bn_add_relu_bwd_reduce_cn8(const void* dy, const void* y, const void* u, const void* r, const float* mean_u,
                                   const float* mean_r, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                   sar_stream_t s);
/* The block tail with a 1-bit ReLU mask: the forward also writes mask[g][col] (one byte per unit, bit j = stored channel 8 g + j
 * is > 0; ld bytes per plane), and the two backward passes read that byte instead of the 16-byte unit of y -- the same
 * results bit for bit, 115 MB less HBM traffic per pass at the NTU shapes (models/stgcn.py:37,62-63 and their gradients). */
int sar_bn_add_relu_fwd_mask_cn8(const void* u, const float* scale, const float* shift, int res_kind, const void* r,
                                 const float* res_scale, const float* res_shift, void* y, void* mask, int C, int64_t n, int64_t ld,
                                 sar_stream_t s);
int sar_bn_add_relu_bwd_reduce_mask_cn8(const void* dy, const void* mask, const void* u, const void* r, const float* mean_u,
                                        const float* mean_r, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                        sar_stream_t s);
int sar_bn_add_relu_bwd_apply_mask_cn8(const void* dy, const void* mask, const void* u, const void* r, const float* k1,
                                       const float* k2, const float* k3, const float* rk1, const float* rk2, const float* rk3,
                                       void* du, void* dr, void* dz_out, int C, int64_t n, int64_t ld, sar_stream_t s);
int sar_bn_add_relu_bwd_reduce_tail_cn8(const void* dy, const void* y, const void* u, const void* r, const float* mean_u,
                                        const float* mean_r, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                        const sar_bn_tail* tail, sar_stream_t s);
int sar_bn_add_relu_bwd_apply_cn8(const void* dy, const void* y, const void* u, const void* r, const float* k1,
                                  const float* k2, const float* k3, const float* rk1, const float* rk2, const float* rk3,
                                  void* du, void* dr, void* dz_out, int C, int64_t n, int64_t ld, sar_stream_t s);
int sar_affine2_cn8(const void* a, const void* b, const float* k1, const float* k2, const float* k3, void* out, int C,
                    int64_t n, int64_t ld, sar_stream_t s);
/* data_bn (models/stgcn.py:136-147) writing / reading the (C <= 8)-channel CN8 input tensor (one plane) */
int sar_data_bn_apply_cn8(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                          const float* scale, const float* shift, void* out, int64_t ld_out, sar_stream_t s);
int sar_data_bn_bwd_reduce_cn8(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                               const void* dy, int64_t ld_dy, const float* mean, float* partials, sar_stream_t s);
/* GlobalAveragePooling2D + mean over bodies (models/stgcn.py:153-156) on a CN8 tensor, and its gradient */
int sar_pool_fwd_cn8(const void* y, int64_t ld, int C, int B, int TV, int Mp, float* feat, sar_stream_t s);
int sar_pool_bwd_cn8(const float* dfeat, int64_t ld, int C, int B, int TV, int Mp, void* dy, sar_stream_t s);
/* layout conversion fp32 CN [C][ld] <-> CN8 (boundary of the bf16 engine, tests) */
int sar_cn_to_cn8(const float* x, int64_t ld_x, void* out, int64_t ld_out, int C, int64_t n, sar_stream_t s);
int sar_cn8_to_cn(const void* x, int64_t ld_x, float* out, int64_t ld_out, int C, int64_t n, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Box calibration (csrc/box_probe.hip; bench.py's "box" object -- no reference counterpart, measurement only): what the box
 * this process landed on sustains, so that a line's `frac` (against the guide's peaks) can be read next to `frac_of_box`.
 *   sar_box_mfma        dense loop of one matrix instruction: kind 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_32x32x16_f16,
 *                       2 = v_mfma_f32_32x32x16_bf16; `blocks` workgroups of four waves, iters x 32 instructions per wave, operands
 *                       in registers.  sink: [blocks * 256] floats (written).  clocks: NULL or [blocks][2] uint32 -- per workgroup
 *                       the shader-clock cycles (s_memtime) and the 100 MHz ticks (s_memrealtime) it lived: their ratio x 100 MHz
 *                       is the clock the chip HELD under this load.
 *   sar_box_mfma_flops  floating-point operations one such launch executes (host arithmetic, no device work).
 *   sar_box_copy_f32    dst[i] = src[i], float4 units, n % 4 == 0, 16-byte aligned: 8 n bytes of HBM traffic per launch.
 * ------------------------------------------------------------------------------------------------ */
int sar_box_mfma(int kind, int blocks, int iters, float* sink, uint32_t* clocks, sar_stream_t s);
int64_t sar_box_mfma_flops(int kind, int blocks, int iters);
int sar_box_copy_f32(const float* src, float* dst, int64_t n, sar_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Host-side input helpers (HOST pointers, no stream, no device work): what tf.data.TFRecordDataset's native reader does
 * for main_gnn.py:159-194 -- the reference's clips are tf.train.Example records written by
 * data_gen/gen_tfrecord_data.py:25-33,76-85.
 *   sar_crc32c          CRC-32C (Castagnoli, reflected 0x82F63B78, init/xorout 0xFFFFFFFF); SSE4.2 crc32 instruction when
 *                       the CPU has it, slice-by-8 tables otherwise (sar_crc32c_sw = always the tables: self-test).
 *   sar_masked_crc32c   TFRecord's mask: rotr(crc, 15) + 0xA282EAD8.
 *   sar_tfrecord_index  walks a whole shard held in memory (uint64 length | uint32 masked crc(length) | data |
 *                       uint32 masked crc(data)) and returns the number of records, writing the payload offset / length
 *                       of the first max_records of them (offsets / lengths may be NULL to count only).
 *                       verify: 0 = framing only, 1 = + length CRCs, 2 = + data CRCs.
 *                       Error: -(code + 4*record_index), code 2 = truncated header, 3 = corrupt length CRC,
 *                       4 = truncated record, 5 = corrupt data CRC; -1 = bad arguments.
 * ------------------------------------------------------------------------------------------------ */
uint32_t sar_crc32c(const void* data, int64_t n);
uint32_t sar_crc32c_sw(const void* data, int64_t n);
uint32_t sar_masked_crc32c(const void* data, int64_t n);
int64_t sar_tfrecord_index(const void* file, int64_t nbytes, int verify, int64_t* offsets, int64_t* lengths,
                           int64_t max_records);

#ifdef __cplusplus
}
#endif
#endif /* SAR_HIP_H */
