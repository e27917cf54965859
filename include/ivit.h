/*
 * ivit.h - C ABI of the MI355X-native ViT forward engine (libivit.so, gfx950).
 *
 * This is the drop-in boundary under the reference's Python operator API.  The reference
 * (0Marble/interactive-vit) has no native code: every node's arithmetic is the call
 * `sub(x)` into torch's CPU kernels at main/context.py:79-88 (Model.compute), reached from
 * Context.compute (main/context.py:143-147) <- views.compute (main/views.py:30-42).  The ViT model
 * plugin (interactive_vit_amd/models/vit.py, modelled on static/models/vgg16.py:10-62) replaces that
 * one call with the entry points below; SURVEY.md 8(b) lists what a C-ABI replacement must export.
 *
 * Conventions
 *   - plain C types only; no Python or torch objects cross this boundary;
 *   - every function returns 0 on success, non-zero on failure; the message of the last failure
 *     on the calling thread is ivit_last_error().  The Python shim raises Exception(message), which
 *     views.compute maps to HTTP 400 (main/views.py:40-42) - the reference's error convention;
 *   - the caller owns in/out buffers and keeps them alive for the call; the engine owns its device
 *     memory (bf16 weights, activation workspaces);
 *   - an engine may be used from several host threads: calls on one handle are serialised inside;
 *   - tensors are row-major, float32 at the boundary (the wire format is f32: main/message.py:41-59).
 *
 * Stages (the nodes of the plugin, SURVEY.md 8(a2')); L = config.layers:
 *     0            transform      [B,3,S,S]  -> [B,3,S,S]   (x-mean)/std per channel
 *     1            conv_proj      [B,3,S,S]  -> [B,Np,D]    unfold + GEMM + bias
 *     2            tokens         [B,Np,D]   -> [B,N,D]     class token, + position embedding
 *     3 .. 3+L-1   encoder.layers.i  [B,N,D] -> [B,N,D]     residual-inclusive transformer block
 *     3+L          encoder.ln     [B,N,D]    -> [B,N,D]
 *     4+L          cls            [B,N,D]    -> [B,D]
 *     5+L          heads          [B,D]      -> [B,classes]
 * A range [begin,end) of consecutive stages runs fused: intermediate activations stay on the device
 * in the engine's internal layout (fp32 residual stream, bf16 GEMM operands).
 */
#ifndef IVIT_H
#define IVIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVIT_ABI_VERSION 10

#define IVIT_PRECISION_BF16 0   /* bf16 GEMM operands (default) */
#define IVIT_PRECISION_FP8  1   /* encoder GEMMs on e4m3 weights + activations (BASELINE config 5); needs ivit_fp8_calibrate */
#define IVIT_PRECISION_F16  2   /* IEEE f16 GEMM operands on v_mfma_f32_16x16x32_f16 - the bf16 rate on gfx950, 11 significant
                                   bits: every node within 1e-3 of the PLAIN f32 forward (the reference's sub(x), main/context.py:79-88,
                                   returns f32), where bf16 operand rounding alone costs 2e-3.  Range 6.5e4: meant for LayerNorm-ed
                                   ViT activations; accumulation, statistics and the residual stream stay f32 as in the other modes */

#define IVIT_PRECISION_F16X 3   /* the TOLERANCE mode: IVIT_PRECISION_F16 with split-operand GEMMs where f16 operand rounding is the larger part of the
                                   distance from the f32 forward (tools/f16_error_terms.py: the weights are 55 % of it): MLP up and MLP down on hi + lo
                                   WEIGHT pairs (W = hi + lo, two MFMA passes over the same activations, one f32 accumulator), the out-projection on
                                   pairs of both operands (three passes).  Logits of ViT-B/16, ViT-L/16-384 and ViT-H/14 within 1e-3 of the CPU f32
                                   forward with >= 30 % margin (6.3e-4 ... 6.9e-4; profiles/r04_f16x_split_sets.txt).  IVIT_F16X_PROJ=0 (measurement
                                   knob) drops the out-projection split: 7.1e-4 by emulation, 8.6e-4 measured - inside 1e-3, not by 20 %.
                                   lo = rn16(w - hi) is stored unscaled: for |w| < 2^-3 it is an f16 subnormal (step 2^-24), so hi + lo carries ~19
                                   significant bits for the N(0, 0.02^2)-scale weights of the tests, not 22; the MFMA keeps subnormal operands
                                   (tests/test_gpu_parity.py: test_split_weight_low_parts_survive).
                                   Both f16 modes run the patch embedding and the classifier head on hi + lo pairs of both operands (0.7 % of the FLOPs) */

#define IVIT_PRECISION_FP8M 4   /* e4m3 on the MLP pair only (round 4: the configuration the fp8 error budget names, profiles/r03_fp8_error_terms.txt): MLP up /
                                   down - 53 % of the FLOPs - on the 2x-rate scaled MFMA with the scales of IVIT_PRECISION_FP8 (ivit_fp8_calibrate); QKV,
                                   attention and out-projection stay bf16 (LayerNorm kernels) */

typedef struct ivit_engine ivit_engine;

typedef struct ivit_config {
    int32_t image;      /* S */
    int32_t patch;      /* p */
    int32_t dim;        /* D */
    int32_t heads;      /* H, D % H == 0, head dim in {64, 80, 128} or any multiple of 16 <= 128 */
    int32_t layers;     /* L */
    int32_t mlp;        /* hidden width of the MLP */
    int32_t classes;
    float   ln_eps;
    int32_t device;     /* HIP device ordinal */
    int32_t max_batch;  /* workspaces are sized for this many images per call */
    int32_t precision;  /* IVIT_PRECISION_* (BF16, FP8, F16, F16X, FP8M) */
} ivit_config;

/* ABI / build introspection (no GPU needed). */
int         ivit_abi_version(void);
const char* ivit_build_info(void);
const char* ivit_last_error(void);

/* Stage bookkeeping (pure host arithmetic, no GPU needed).
 * ivit_stage_count: 6 + L.  ivit_stage_shape: per-image element shape of a stage's input
 * (which = 0) or output (which = 1); returns ndim and writes up to 3 dims. */
int ivit_stage_count(const ivit_config* cfg);
int ivit_stage_shape(const ivit_config* cfg, int stage, int which, int64_t dims[3]);

/* Patch bookkeeping oracle hook (host arithmetic): flat offset into a [3,S,S] image of unfold
 * element (patch n, column k), the same function the device unfold kernel is compiled from.
 * Replaces nothing in the reference (it has no unfold); pins "bit-exact patch index". */
int64_t ivit_unfold_offset(int32_t image, int32_t patch, int32_t n, int32_t k);

/* Lifecycle. */
int  ivit_create(const ivit_config* cfg, ivit_engine** out);
void ivit_destroy(ivit_engine* e);

/* Weights: torchvision VisionTransformer state-dict names (see interactive_vit_amd/weights.py),
 * float32 host arrays; converted to bf16 (matrices) / kept f32 (biases, LN, pos, cls) on device.
 * ivit_weights_ready fails and names the first missing tensor if the set is incomplete. */
int ivit_set_weight(ivit_engine* e, const char* name, const float* host, const int64_t* shape, int ndim);
int ivit_weights_ready(ivit_engine* e);

/* 1 when a forward of `batch` images folds every LayerNorm of the encoder into the GEMM that consumes it, else 0.
 * The fold is the default on the 16-bit data paths (IVIT_FOLD_LN=0 in the environment at ivit_create disables it) at every
 * batch size since round 5 (calls large enough for the one-per-CU 256x256 tile - ViT-L / ViT-H batches - fold the row
 * statistics once per row with a small kernel; IVIT_FOLD_LN=3 restores rounds 3-4's rule, which kept the LayerNorm kernels
 * there); the e4m3 paths and a weight set that trips the guard (ivit_ln_fold_calibrate) keep the kernels.  Both forms are
 * 16-bit evaluations of the same f32 contract with different rounding points; the parity tests ask which one the oracle's
 * rounding-aware mode has to mirror. */
int ivit_ln_fold(const ivit_engine* e, int batch);

/* 0 when the MLP of a forward of `batch` images runs as two GEMM launches (MLP up + GELU, MLP down + residual); else the form of the fused MLP
 * kernel it runs as (csrc/mlp_fused_kernel.h: LayerNorm fold, up, GELU, down, residual and the next layer's statistics in one launch; round 5):
 * 1 plain 16-bit weights; 2 both weight matrices as hi / lo pairs (IVIT_PRECISION_F16X with IVIT_F16X_MLP2=1), 3 the up weight only
 * (IVIT_PRECISION_F16X's default).  Every form is BIT-IDENTICAL to the two launches on the same operands.  The fused kernel takes the calls whose
 * 64-row workgroups fill the chip (dim 512 / 768, LayerNorm fold on; IVIT_FUSED_MLP=0 disables it). */
int ivit_fused_mlp(const ivit_engine* e, int batch);

/* Which GEMMs of this engine multiply hi / lo pairs of f16 values (IVIT_PRECISION_F16 / F16X; 0 for the other precisions): bit 0 patch embedding and
 * classifier head (both operands), bit 1 the out-projection (both operands; IVIT_F16X_PROJ=0 drops it), bit 2 the MLP-up weight, bit 3 the MLP-down
 * weight (by default only where the fused MLP kernel cannot run - dim other than 512 / 768; IVIT_F16X_MLP2=0 / 1 overrides). */
int ivit_split_set(const ivit_engine* e);

/* Calibration and guard of the LayerNorm fold for a weight set.  The folded GEMM multiplies a 16-bit copy of the residual-stream rows that is not
 * centred per row, so its operand-rounding noise is rms(copy) / std(x) times the unfolded form's (which rounds (x - mean) rstd) - a factor
 * sqrt(1 + (mean/std)^2) for the plain copy rn16(x): small for the seeded weights of the tests, not guaranteed for a user-supplied checkpoint
 * (reference plugins load whatever state dict they are given, static/models/vgg16.py:12-14), whose rows carry channel-constant offsets and outlier
 * channels.  Those are constant across rows, so this entry (round 5) removes them before the rounding: it runs one forward of `batch` sample images
 * (`in`: device f32 [B,3,S,S] in [0,1]) with LayerNorm kernels, takes at every LayerNorm input of every layer the per-channel mean over the rows as
 * that input's CENTRE vector m, and from then on the engine's copies are rn16(x - m) and the folded GEMMs add d = W' m back in their epilogues:
 *     LN(x) W^T + b = rstd (rn16(x - m) W'^T + d - mean s) + c        (statistics mean / rstd are those of x itself; m = 0 reproduces rounds 3-4).
 * The guard statistic is the noise factor as sqrt(factor^2 - 1) - for the plain copy |mean| / std - max over the rows of a LayerNorm input.  A
 * population mean does not suit every row (a class-token row that lacks the patch rows' common offset gets WORSE when that offset is subtracted), so
 * each of the 2 * layers inputs keeps its vector only where the centred copies' statistic is below the plain copies' - and, on the bf16 data path,
 * only where the plain copies' statistic exceeds 0.25 (the centred epilogues cost ~2 % of a ViT-B/16 step; rows that are nearly centred anyway gain
 * nothing measurable in bf16; the f16 data paths and IVIT_FOLD_CENTRE=2 centre wherever it lowers the statistic) - otherwise that input stays on the
 * plain copy (vector zero).  *max_ratio (optional) receives the maximum over the inputs of the statistic of the copy each one uses, and the fold
 * is kept for this engine only if it is <= threshold (0.5: at most 12 % more rounding noise than the LayerNorm kernels); otherwise every later call
 * uses the LayerNorm kernels (ivit_ln_fold then answers 0).
 * IVIT_FOLD_CENTRE=0 in the environment at ivit_create keeps the plain copy (guard on |mean| / std, as before).  ivit_set_weight drops the vectors
 * (plain copy again until the next calibration).  Synchronises `stream`. */
int ivit_ln_fold_calibrate(ivit_engine* e, int batch, const void* in, float threshold, float* max_ratio, void* stream);

/* The centre vectors of the last calibration, for a checker that mirrors the engine's rounding points: out (optional, room for 2 * layers * dim
 * floats) receives m of LN1 of layer 0, LN2 of layer 0, LN1 of layer 1, ... (zeros when the copies are not centred); *centred (optional) says whether
 * they are in use, *plain_ratio (optional) receives the guard statistic of the PLAIN copy at that calibration (what rounds 3-4 guarded on). */
int ivit_ln_fold_centres(ivit_engine* e, float* out, int64_t capacity, int* centred, float* plain_ratio);

/* Replaces: Model.compute -> sub(x)  (reference main/context.py:79-88) for a run of nodes.
 * Host-buffer form (interactive path: the request tensors are CPU f32, main/message.py:58):
 * copies `in` to the device, runs stages [begin,end), copies the last stage's output to `out`.
 * out_capacity is in floats; fails if the result does not fit. */
int ivit_forward_host(ivit_engine* e, int stage_begin, int stage_end, int batch,
                      const float* in, float* out, int64_t out_capacity);

/* Chained form of ivit_forward_host for a request that walks a chain of nodes (Context.compute calls
 * one node after the other and hands node k's output tensor to node k+1 by reference,
 * main/context.py:143-147, main/graph.py:22-29).  The f32 output of every host
 * call also stays on the device; *out_token names it.  A call that passes that token as in_token -
 * the caller asserting that `in` still holds exactly the bytes it was given back then - skips the
 * upload and consumes the resident copy (bit-identical: both are the same f32 values).  A stale,
 * foreign or zero token falls back to uploading `in`, which must therefore always be valid.
 * Any later host call invalidates older tokens. */
int ivit_forward_host_chained(ivit_engine* e, int stage_begin, int stage_end, int batch,
                              const float* in, float* out, int64_t out_capacity,
                              uint64_t in_token, uint64_t* out_token);

/* Asynchronous form of ivit_forward_host_chained (Response collects every node's output only at the END of
 * the request, main/message.py:77-83): the call enqueues upload (or takes the resident copy), kernels and the D2H copy of the
 * result into `out` - which must be page-locked host memory that stays alive - and returns WITHOUT waiting; *ticket names the
 * call.  `out` may be read only after ivit_host_wait(e, ticket) (or after the wait for any LATER ticket of this engine: host
 * calls complete in order).  So node k's copy runs behind node k+1's kernels, and one wait at the end of a chain covers it. */
int ivit_forward_host_async(ivit_engine* e, int stage_begin, int stage_end, int batch,
                            const float* in, float* out, int64_t out_capacity,
                            uint64_t in_token, uint64_t* out_token, uint64_t* ticket);
int ivit_host_wait(ivit_engine* e, uint64_t ticket);

/* The `<model>:preprocess` node: what the reference's model plugin does to a raw image with its weights'
 * preset (static/models/vgg16.py:40-42, `weights.transforms()`): `in` is [batch,3,height,width] f32 in
 * [0,1], any size; the shorter side is resized to image*256/224 (antialiased bilinear, ATen's
 * upsample_bilinear2d_aa weights), the centre image x image square is cropped and normalised with the
 * ImageNet mean / std - i.e. the result is what the `transform` stage gives for an image x image input and
 * feeds `conv_proj`.  Host form: *out_token (optional) names the resident copy as in
 * ivit_forward_host_chained.  Device form: enqueued on `stream`, no synchronisation. */
int ivit_preprocess_host(ivit_engine* e, int batch, const float* in, int height, int width,
                         float* out, int64_t out_capacity, uint64_t* out_token);
int ivit_preprocess(ivit_engine* e, int batch, const void* in, int height, int width, void* out, void* stream);

/* Device-pointer form (benchmark / chained nodes): `in` and `out` are device f32 buffers on the
 * engine's device, work is enqueued on `stream` (a hipStream_t; NULL = the null stream) and the
 * call returns without synchronising.  cls_out (optional, may be NULL) receives the [B,D] f32
 * class-token features after encoder.ln when the range ends at or after stage 4+L. */
int ivit_forward_device(ivit_engine* e, int stage_begin, int stage_end, int batch,
                        const void* in, void* out, void* cls_out, void* stream);

/* Attention probabilities of encoder layer `layer` for a residual-stream input [B,N,D] (f32, device):
 * softmax(q k^T / sqrt(dh)) as f32 [B, heads, N, N] into `out` (device).  Backs the
 * `<model>:encoder.layers.<i>.attn` node (attention maps the client's MultiView
 * can display, multi_view.js:53-66); the reference has no counterpart. */
int ivit_attention_map(ivit_engine* e, int layer, int batch, const void* in, void* out, void* stream);
/* host-buffer form of the same (CPU f32 in / out; out_capacity in floats) */
int ivit_attention_map_host(ivit_engine* e, int layer, int batch, const float* in, float* out, int64_t out_capacity);

/* The layer node with its attention map as a SECOND output channel (Response ships every channel of every node,
 * main/message.py:80-83; io() is the operator's to override, main/context.py:94-96): `out` = the residual-inclusive block as
 * ivit_forward_device(stage 3 + layer) gives it, `attn` = f32 [B, heads, N, N] from the q|k|v tensor that same call computed - bit for
 * bit the map of the `.attn` inspector node, without its second LayerNorm + QKV GEMM.  Backs `<model>:encoder.layers.<i>.with_attn`. */
int ivit_layer_with_attn(ivit_engine* e, int layer, int batch, const void* in, void* out, void* attn, void* stream);
int ivit_layer_with_attn_host(ivit_engine* e, int layer, int batch, const float* in, float* out, int64_t out_capacity, float* attn, int64_t attn_capacity);

/* fp8 data path (IVIT_PRECISION_FP8).  Policy (the reference has none - stated in DESIGN.md): OCP e4m3fn,
 * saturating; weights quantised per OUTPUT ROW (scale = row amax / 448) from their bf16 copies;
 * activations feeding the four encoder GEMMs of every layer (LN1 out, attention out, LN2 out, GELU
 * out) quantised per TENSOR with static scales = amax / 448 measured by one bf16 calibration forward
 * of `batch` images (`in`: device f32 [B,3,S,S] in [0,1]).  Patch embedding and classifier head stay
 * bf16; accumulation, LN/softmax statistics and the residual stream stay f32; q|k|v stay bf16.
 * ivit_fp8_scales copies the L*4 activation scales (layer-major: h1, att, h2, u) to host memory. */
int ivit_fp8_calibrate(ivit_engine* e, int batch, const void* in, void* stream);
int ivit_fp8_scales(ivit_engine* e, float* out, int capacity);

/* Inspection entry used by the parity tests: the bf16 unfold image [B*Np, Kpad] the patch GEMM
 * consumes, widened to f32 into `out` (device, B*Np*K floats, padding columns dropped).
 * normalise != 0 applies the transform first (the fused path), 0 unfolds the input as is. */
int ivit_debug_unfold(ivit_engine* e, int batch, const void* in, void* out, int normalise, void* stream);

/* Multi-GPU (the reference has no counterpart): images shard by batch over the GPUs of one node, one
 * process and one engine per GPU, weights replicated; the ONLY data-path collective is one all-gather per batch of every
 * rank's packed [b_local, classes + D] f32 block (logits | class-token features), issued by the engine through RCCL over
 * xGMI on the caller's stream.  librccl.so is loaded on demand by these entries.
 *   ivit_comm_unique_id: rank 0 creates the 128-byte id and the host distributes it to the other ranks (any channel);
 *   ivit_comm_init:      every rank binds its engine to the communicator (collective: all ranks must call it);
 *   ivit_allgather_cls:  recv[r * floats_per_rank ...] = rank r's send block, equal block sizes, enqueued on `stream`,
 *                        no synchronisation.  The communicator is destroyed with the engine. */
int ivit_comm_unique_id(void* id128);
int ivit_comm_init(ivit_engine* e, const void* id128, int rank, int world);
int ivit_allgather_cls(ivit_engine* e, const void* send, void* recv, int64_t floats_per_rank, void* stream);

/* The multi-GPU step without copy kernels (round 4).  ivit_forward_device_packed runs stages [stage_begin, end of model) and writes row b of
 * `packed` (device f32, row_stride floats per image, a multiple of 4 >= classes + dim) as [logits | class-token features] - the head GEMM and
 * the final LayerNorm store with that row stride - so the block ivit_allgather_rows sends is produced in place.
 * ivit_shard_layout: the shard rule (rank r of `world` holds rows [begin, begin + rows) of `total`, the first total % world ranks one row
 * more; padded_rows = the largest shard) - host arithmetic, no GPU.
 * ivit_allgather_rows: ONE ncclAllGather of every rank's [rows_local, row_floats] block into recv [total_rows, row_floats] in image order;
 * ragged shards (total_rows % world != 0) are padded to padded_rows inside the engine for the collective and compacted afterwards (staging sized at
 * ivit_comm_init for max_batch packed rows per rank; wider rows grow it behind a device synchronise).  A rank whose shard is empty (total_rows <
 * world) passes rows_local = 0 and may pass send = NULL; it still takes part in the collective.  A non-zero return on ANY rank (argument check,
 * allocation) means that rank did not enter the collective: its peers block in theirs - treat it as fatal for the communicator. */
int ivit_forward_device_packed(ivit_engine* e, int stage_begin, int batch, const void* in, void* packed, int64_t row_stride, void* stream);
int ivit_shard_layout(int64_t total, int world, int rank, int64_t* begin, int64_t* rows, int64_t* padded_rows);
int ivit_allgather_rows(ivit_engine* e, const void* send, int64_t rows_local, int64_t row_floats, int64_t total_rows, void* recv, void* stream);

/* Inspection entries used by the per-GEMM parity tests.  An encoder layer is seven steps: 1 LayerNorm 1 (with the
 * LayerNorm fold: the 16-bit operand copy of x the QKV GEMM consumes), 2 QKV projection, 3 attention, 4 out-projection
 * (+ residual), 5 LayerNorm 2 (fold: nothing new - the copy the out-projection left), 6 MLP up (+ GELU), 7 MLP down
 * (+ residual).  ivit_debug_layer_tap runs layer `layer` on `in` (device f32 [B,N,D]) up to and including step `tap` and
 * copies that step's output AS STORED - e4m3 bytes, bf16 / f16 halves or f32, rows of *row_bytes bytes, *elem_bytes per
 * element - to `out` (device): the exact operand bytes the next GEMM multiplies, so that a test can gate every GEMM on
 * identical quantised inputs.  ivit_debug_weight_fp8 copies a quantised matrix of the fp8 path (which: 0 in_proj,
 * 1 out_proj, 2 mlp.0, 3 mlp.3; [rows][*ld] bytes) and its per-row scales to HOST memory. */
int ivit_debug_layer_tap(ivit_engine* e, int layer, int batch, const void* in, int tap, void* out, int64_t out_capacity_bytes,
                         int64_t* row_bytes, int* elem_bytes, void* stream);
int ivit_debug_weight_fp8(ivit_engine* e, int layer, int which, void* out_bytes, int64_t out_capacity_bytes, float* out_rowscale,
                          int rowscale_capacity, int* rows, int* cols, int* ld);

/* Per-kernel-class device timing (HIP events on the launch stream), for bench.py's roofline line.
 * While enabled every launch of a class is bracketed by an event pair; ivit_profile_read
 * synchronises, then returns accumulated milliseconds, launch count and algorithmic FLOPs and
 * bytes of class `cls` since ivit_profile_reset.  Class names: ivit_profile_class_name(i). */
int         ivit_profile_enable(ivit_engine* e, int on);
int         ivit_profile_reset(ivit_engine* e);
int         ivit_profile_class_count(void);
const char* ivit_profile_class_name(int cls);
int         ivit_profile_read(ivit_engine* e, int cls, double* ms, int64_t* launches,
                              double* flops, double* bytes);
/* The same measurements per launch site, named "role:kernel" - e.g. "mlp1:ivit_gemm_bf16_160x128x64_lf",
 * "qkv:ivit_gemm_bf16_256x256x64_stag", "attention:attention" - for the per-kernel roofline list of bench.py
 * and for the tests that assert WHICH kernel a configuration dispatches (the tile is picked per shape).
 * ivit_profile_kernel_count: distinct sites seen since ivit_profile_reset (-1: null engine);
 * ivit_profile_kernel_read: copies the name (NUL-terminated, truncated to name_capacity) and the totals of site `index`. */
int         ivit_profile_kernel_count(ivit_engine* e);
int         ivit_profile_kernel_read(ivit_engine* e, int index, char* name, int name_capacity, double* ms,
                                     int64_t* launches, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* IVIT_H */
