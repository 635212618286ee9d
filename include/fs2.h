/*
 * fs2.h -- C ABI of libfs2_hip.so: the MI355X (gfx950) FastSpeech2 mel-generation forward pass.
 *
 * The reference (rishikksh20/FastSpeech2) has no FFI: its hot path sits behind a Python
 * torch.nn.Module, `fastspeech.FeedForwardTransformer` (reference fastspeech.py:28).  This header is
 * the boundary a maintainer of the reference would bind (ctypes stub in INTEGRATION.md) to replace the
 * tensor work inside `_forward` (fastspeech.py:169-243) -- encoder/decoder FFT blocks
 * (core/encoder.py:46-71,185-204; core/attention.py:30-74; core/modules.py:237-248), duration / pitch /
 * energy predictors (core/duration_modeling/duration_predictor.py:64-86; core/variance_predictor.py:39-60,
 * 154-159), length regulator (core/duration_modeling/length_regulator.py:38-95) and Postnet
 * (core/modules.py:350-359) -- with hand-written HIP kernels.
 *
 * Conventions: plain pointers and sizes only (no torch types); every function returns 0 or a negative
 * FS2_ERR_* code and never throws; `fs2_last_error` gives the message.  "device" pointers are HIP device
 * memory on the handle's device, "host" pointers are ordinary host memory.  All launches go to the
 * caller's hipStream_t (passed as void*); no function allocates device memory except fs2_create /
 * fs2_load_weights (weight storage).  A handle is bound to one device and is not thread-safe.
 *
 * A forward pass is two calls, because the number of mel frames is data dependent:
 *   fs2_encode  : phoneme ids -> encoder -> duration predictor -> frame counts (olens, device)
 *   (caller reads olens back: the one unavoidable host sync of the path)
 *   fs2_decode  : length regulator -> pitch/energy -> decoder -> mel projection -> Postnet
 */
#ifndef FS2_H_
#define FS2_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS2_OK 0
#define FS2_ERR_ARG (-1)         /* bad argument / shape                                   */
#define FS2_ERR_HIP (-2)         /* a HIP runtime call failed                              */
#define FS2_ERR_STATE (-3)       /* call order violated (e.g. decode before encode)        */
#define FS2_ERR_WEIGHT (-4)      /* a required tensor is missing or has the wrong shape    */
#define FS2_ERR_WORKSPACE (-5)   /* workspace too small                                    */
#define FS2_ERR_UNSUPPORTED (-6) /* configuration outside what the kernels implement       */

/* arithmetic modes of the GEMM-shaped kernels */
#define FS2_PREC_FP32 0   /* f32-input MFMA (v_mfma_f32_16x16x4_f32), exact fp32            */
#define FS2_PREC_BF16X3 1 /* split-bf16: hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 */
#define FS2_PREC_BF16 2   /* plain bf16 inputs, fp32 accumulate                             */
/* mixed modes: as FS2_PREC_BF16X3, except that the FFN convolution w_1 (the dominant kernel) runs on fp16
 * operands (v_mfma_f32_16x16x32_f16): activations split hi + lo, weights rounded to fp16 once (2 MFMAs per
 * fragment pair), or both operands rounded once (1 MFMA).  Measured error / speed: BASELINE.md section 4. */
#define FS2_PREC_MIX_F16X2 3
#define FS2_PREC_MIX_F16X1 4
/* as FS2_PREC_BF16X3, except that the 9-tap FFN convolution computes a.w = ah.wh (fp16 MFMA) + ra.wh + ah.rw (block-scaled
 * fp8 MFMA, v_mfma_scale_f32_16x16x128_f8f6f4): ~2.2 MFMA-equivalents per product at split-bf16-class accuracy */
#define FS2_PREC_MIX_MX 5
/* as FS2_PREC_MIX_MX with both cross terms of the decoder's FFN convolution in block-scaled fp4 (e2m1; one E8M0 scale per 16-channel block of a frame / of a weight row and tap):
 * 1.5 MFMA-equivalents per product.  Applies where the decoder's activations travel as planes only (big frame-level regimes: gemm_row4_bf16
 * produces the operand with its row scales); everywhere else the mode IS FS2_PREC_MIX_MX.  Measured error / speed: BASELINE.md section 4. */
#define FS2_PREC_MIX_MX4 6

typedef struct fs2_handle fs2_handle;

/* ABI revision of this header.  The four structs that are the argument of an entry point -- fs2_config, fs2_encode_io,
 * fs2_decode_io, fs2_op_gemm_args -- start with `struct_size`, which the library compares with its own sizeof: a caller
 * built against a different revision gets FS2_ERR_ARG (and a message naming both sizes) instead of fields read at the
 * wrong offsets.  Two structs carry no size field: `fs2_batch`, embedded in the io structs and so covered by their check
 * and passed alone only to the workspace-size / row-capacity queries, and `fs2_tensor_desc`, the array elements handed to
 * fs2_load_weights.  Their layout is frozen within a revision: any change to them bumps FS2_ABI_VERSION, and a binding
 * compares its own FS2_ABI_VERSION with fs2_abi_version() before the first call (fastspeech2_amd/_lib.py: lib();
 * csrc/fs2_torch_op.cpp: check_abi()). */
#define FS2_ABI_VERSION 4
int32_t fs2_abi_version(void);

/* Model hyper-parameters: the hp.model / hp.data fields FeedForwardTransformer.__init__ reads
 * (reference fastspeech.py:53-160). */
typedef struct fs2_config {
    uint32_t struct_size;               /* = sizeof(fs2_config): checked by fs2_create (FS2_ERR_ARG on mismatch), so a
                                           binding compiled against another revision of this header is rejected, not misread */
    int32_t idim, odim;                 /* phoneme symbols (68), mel bins (80)               */
    int32_t adim, aheads, elayers, eunits;
    int32_t ddim, dlayers, dunits;
    int32_t ffn_kernel;                 /* positionwise_conv_kernel_size; 1 for "linear"      */
    int32_t dur_layers, dur_chans, dur_kernel;   /* duration predictor (from hp)              */
    int32_t var_layers, var_chans, var_kernel;   /* pitch/energy predictors (hard-wired 2/256/3,
                                                    reference variance_predictor.py:125,198)   */
    int32_t n_bins;                     /* 256 quantisation levels                            */
    int32_t postnet_layers, postnet_chans, postnet_filts, use_batch_norm;
    int32_t use_scaled_pos_enc;
    int32_t reduction_factor;           /* r in [1, 8]: feat_out emits r mel frames per decoder frame; the mel outputs
                                           (before / after) then hold Lmax * r frames and after_packed r * sum(olens) rows */
    int32_t device;                     /* HIP device ordinal                                 */
    int32_t decoder_input_layer;        /* 1: Linear -> LN -> ReLU -> +pe (fastspeech.py:120-135, encoder.py:118-125);
                                           0: +pe only (the TorchScript twin, utils/fastspeech2_script.py:112-127;
                                           needs ddim == adim)                                */
    /* FFT-block variants (reference core/encoder.py:53-71,201-202; hp.model.{encoder,decoder}_{normalize_before,concat_after}) */
    int32_t enc_normalize_before, dec_normalize_before;   /* LayerNorm in front of the sub-layers + after_norm at the end */
    int32_t enc_concat_after, dec_concat_after;           /* x + concat_linear(cat(x, self_attn(x))) instead of x + self_attn(x) */
} fs2_config;

/* One reference-layout tensor (state_dict entry), fp32, resident on the device. */
typedef struct fs2_tensor_desc {
    const char *name;   /* e.g. "decoder.encoders_.0.feed_forward.w_1.weight" */
    const void *data;   /* device pointer, contiguous, float32                 */
    int32_t ndim;
    int64_t shape[4];
} fs2_tensor_desc;

/* Host-side description of a batch (lengths are host data: the reference also reads them on the host,
 * utils/util.py:263-264). */
typedef struct fs2_batch {
    int32_t B;              /* utterances                                                      */
    int32_t Tmax;           /* padded phoneme length of xs                                      */
    const int64_t *ilens;   /* host [B] phoneme counts                                          */
    int32_t compat_padded;  /* 0: per-utterance semantics (batch invariant; what inference()
                               computes).  1: reproduce the reference's padded-batch numerics
                               (conv / unmasked attention see the pad rows)      */
    int32_t precision;      /* FS2_PREC_*                                                      */
    /* Kernel-variant regime (ABI 4).  Some launches exist in variants that sum in different orders (LayerNorm fused into the
     * row-complete GEMM or not, deterministic split-K, the 32- or the 64-query attention kernel); which one runs is a function
     * of the SIZE of the batch only -- of these two numbers, never of a capacity or of the data.  0 / 0 = this call's own batch
     * (sum of ilens, B).  A caller that runs a SHARD of a larger batch (one rank of the multi-GPU split, a batch cut into
     * pieces) passes the WHOLE batch's numbers on every piece: every utterance is then computed by exactly the kernels the
     * one-call run of the whole batch uses, and its result is bit-identical to that run's (SURVEY.md section 8e's criterion;
     * the per-utterance semantics of reference fastspeech.py:169-243).  Both numbers must be given together. */
    int64_t regime_tokens;      /* phonemes of the batch the variants are chosen for (0: sum of ilens)  */
    int32_t regime_utterances;  /* utterances of that batch (0: B)                                      */
} fs2_batch;

typedef struct fs2_encode_io {
    uint32_t struct_size;     /* = sizeof(fs2_encode_io); fs2_encode returns FS2_ERR_ARG on mismatch         */
    fs2_batch batch;
    const int64_t *xs;        /* device [B, Tmax] phoneme ids (0 = pad)                         */
    const int64_t *ds;        /* device [B, Tmax] durations to use (teacher forcing / override),
                                 or NULL: use the predicted durations                           */
    float *d_log;             /* device [B, Tmax] log-domain predictor output, pads = 0, or NULL */
    int64_t *d_int;           /* device [B, Tmax] clamp(round(exp(y)-1),0), pads = 0, or NULL    */
    int64_t *olens;           /* device [B] frames per utterance = sum of the durations used
                                 (an all-zero row counts as all ones, length_regulator.py:86-88);
                                 -1 for an utterance whose row of xs (all Tmax positions, the padding behind
                                 ilens included) holds a phoneme id outside [0, idim): the
                                 reference's nn.Embedding raises there (fastspeech.py:65-67), a caller
                                 of the host-driven layout must too (fs2_decode refuses the value), the
                                 device-driven layout reports FS2_OVF_BAD_ID                     */
    float *enc_out;           /* device [B, Tmax, adim] encoder output, pads = 0, or NULL        */
    void *workspace;          /* device, fs2_token_workspace_bytes(); must stay alive and
                                 untouched until the matching fs2_decode returns                 */
    size_t workspace_bytes;
    float duration_alpha;     /* length-regulator speed control (length_regulator.py:57-59): the durations
                                 used become round(d * alpha); 0 or 1 = unchanged.  d_int stays unscaled */
} fs2_encode_io;

typedef struct fs2_decode_io {
    uint32_t struct_size;     /* = sizeof(fs2_decode_io); fs2_decode returns FS2_ERR_ARG on mismatch         */
    fs2_batch batch;
    const int64_t *olens;     /* HOST [B]: the values fs2_encode wrote to its device olens       */
    int32_t Lmax;             /* padded frame length of the outputs (>= max olens)               */
    int32_t masked;           /* compat_padded only: 1 = decoder attention / predictor outputs
                                 masked by olens (teacher-forced `_forward`), 0 = no masks
                                 (inference)                                                     */
    const float *es;          /* device [B, es_stride] energies to quantise, or NULL: predict    */
    const float *ps;          /* device [B, ps_stride] pitches to quantise,  or NULL: predict    */
    int32_t es_stride, ps_stride;
    float *before;            /* device [B, Lmax, odim] mel before Postnet (pads = 0 unless compat) */
    float *after;             /* device [B, Lmax, odim] mel after Postnet (may be NULL when after_packed
                                 is given: the sharded path ships only the packed form)             */
    float *e_out, *p_out;     /* device [B, Lmax] predictor outputs (or NULL)                     */
    int32_t *qe, *qp;         /* device [B, Lmax] bucket indices actually embedded (or NULL)      */
    int32_t *lr_index;        /* device [B, Lmax] phoneme index of every frame, -1 at pads (or NULL) */
    float *dec_out;           /* device [B, Lmax, ddim] decoder output (or NULL)                  */
    void *token_workspace;    /* the workspace given to fs2_encode                               */
    void *workspace;          /* device, fs2_frame_workspace_bytes()                              */
    size_t workspace_bytes;
    float *after_packed;      /* device [r * sum(olens), odim] (r = reduction_factor): the valid frames of
                                 `after`, utterances back to back in batch order (what the multi-GPU
                                 all-gather ships), or NULL                                                   */
    /* Device-driven frame layout (no host read-back of the frame counts between fs2_encode and fs2_decode):
     * set olens = NULL and give capacities instead.  The frame counts are taken from the device copy fs2_encode
     * left in the token workspace, the packed-row layout and the attention work list are built by a kernel, grids
     * are sized for the capacities and the surplus tiles exit at once.  Lmax is then the per-utterance capacity of
     * the padded outputs.  status (device int32[8], required in this mode) receives
     * {total rows used, attention work items, overflow flags, longest utterance, valid frames, waves of this call's decoder
     * attention that left attn_w32's fast path (see fs2_get_counter), 0, 0}; overflow
     * flags != 0 (FS2_OVF_*) means a capacity was too small and the outputs are invalid -- before / after /
     * after_packed are then filled with NaN: rerun with larger capacities or with host olens.  after_packed,
     * if given, must hold r * row_capacity rows in this mode. */
    int64_t row_capacity;     /* 0 = host-driven layout (olens required)                                      */
    int32_t *status;
} fs2_decode_io;

#define FS2_OVF_ROWS 1        /* packed rows needed > row_capacity                                           */
#define FS2_OVF_LMAX 2        /* an utterance is longer than Lmax                                            */
#define FS2_OVF_PE 4          /* an utterance is longer than the decoder's positional table                  */
#define FS2_OVF_EMPTY 8       /* an utterance has no frames                                                  */
#define FS2_OVF_BAD_ID 16     /* an utterance holds a phoneme id outside [0, idim): fs2_encode left the frame count -1
                                 for it (the reference's torch.nn.Embedding raises, fastspeech.py:65-67)                */

/* rows to reserve for fs2_decode's device-driven layout given an estimate of the total frame count (alignment
 * and gap rows of the packed layout included) */
int64_t fs2_row_capacity(const fs2_batch *batch, int64_t total_frames_bound);
size_t fs2_frame_workspace_bytes_cap(const fs2_handle *h, const fs2_batch *batch, int64_t row_capacity, int32_t lmax_capacity);

/* lifecycle (replaces FeedForwardTransformer.__init__ / .to(device) / load_state_dict,
 * reference fastspeech.py:37-167, inference.py:156-166).  Every function that takes a handle runs on the
 * handle's device and restores the caller's current HIP device before returning. */
int fs2_create(const fs2_config *cfg, fs2_handle **out);
void fs2_destroy(fs2_handle *h);
const char *fs2_last_error(const fs2_handle *h); /* h may be NULL: last creation error */
int fs2_load_weights(fs2_handle *h, const fs2_tensor_desc *tensors, int32_t n, void *stream);

/* forward pass (replaces FeedForwardTransformer._forward, reference fastspeech.py:169-243) */
size_t fs2_token_workspace_bytes(const fs2_handle *h, const fs2_batch *batch);
int fs2_encode(fs2_handle *h, void *stream, const fs2_encode_io *io);
size_t fs2_frame_workspace_bytes(const fs2_handle *h, const fs2_batch *batch, const int64_t *olens_host);
int fs2_decode(fs2_handle *h, void *stream, const fs2_decode_io *io);

/* Per-launch timing: while profiling is on, every kernel launch is bracketed by hipEvents recorded on
 * the caller's stream; records accumulate until fs2_set_profiling is called again (which clears them).
 * fs2_get_profile waits for the events and fills names/ms/flops/bytes (algorithmic work of each launch)
 * up to `cap`; returns the number of records. */
int fs2_set_profiling(fs2_handle *h, int32_t on);
int fs2_set_profile_filter(fs2_handle *h, const char *name); /* NULL / "": every launch; else only launches of that name */
int fs2_get_profile(fs2_handle *h, const char **names, float *ms, double *flops, double *bytes, int32_t cap);

/* ---- single operators, exported for per-kernel parity tests (all pointers device unless noted) ---- */

/* y = epilogue(conv1d_k(x) or linear(x)); x:[R,C] rows packed with zero gap rows already in place;
 * w: reference layout [N,C,k] (k=1: [N,C]); epilogue order: +bias, +resid, relu_pre, LayerNorm(eps),
 * act_post (0 none,1 relu,2 tanh), dot with dot_w (+dot_b) -> dot_out[R].  row_valid[R] (int32, may be NULL)
 * marks rows to be written as zeros when 0.  (reference modules.py:237-248, encoder.py:60-69) */
typedef struct fs2_op_gemm_args {
    uint32_t struct_size;     /* = sizeof(fs2_op_gemm_args) */
    int32_t R, C, N, ktaps, precision;
    const float *x, *w, *bias, *resid;
    int32_t relu_pre;
    const float *ln_gamma, *ln_beta;
    float ln_eps;
    int32_t act_post;
    const float *dot_w, *dot_b;
    float *dot_out;
    float *y;
    const int32_t *row_valid;
} fs2_op_gemm_args;
int fs2_op_conv_gemm(void *stream, const fs2_op_gemm_args *a);

/* scaled-dot-product self-attention over packed sequences: qkv [R, 3*D] (q | k | v, head-major inside
 * each), ctx [R, D].  seq_start/len/klen: HOST [B]; in the bf16 modes seq_start must be a multiple of 8.  Keys >= klen
 * never reach the output whatever their rows hold (NaN included).  Query rows >= klen produce zeros when mask_q.
 * (reference core/attention.py:47-70) */
int fs2_op_attention(void *stream, const float *qkv, float *ctx, int32_t D, int32_t heads, int32_t B,
                     const int32_t *seq_start, const int32_t *seq_len, const int32_t *seq_klen,
                     int32_t mask_q, int32_t precision);

/* length regulator on a padded batch: hs [B,Tmax,D], ds [B,Tmax] (i64), ilens HOST [B] ->
 * out [B,Lmax,D] (pads 0), index [B,Lmax] (-1 pads), olens device [B].  Lmax must be >= max olens.
 * alpha > 0: speed control, durations become round(d * alpha) (1 = unchanged).
 * (reference length_regulator.py:38-95, utils/util.py:91-104) */
int fs2_op_length_regulate(void *stream, const float *hs, const int64_t *ds, const int64_t *ilens_host,
                           int32_t B, int32_t Tmax, int32_t D, int32_t Lmax, float alpha, float *out,
                           int32_t *index, int64_t *olens);

/* dst [B, Lout, W] <- packed rows src [sum(lens), W] (utterance b = rows [starts[b], starts[b]+lens[b])), zero
 * padded; starts/lens: HOST [B].  Inverse of fs2_decode_io.after_packed; replaces utils/util.py:91-104 pad_2d_tensor. */
int fs2_op_unpack_rows(void *stream, const float *src, int32_t W, int32_t B, const int32_t *starts, const int32_t *lens,
                       int32_t Lout, float *dst);

/* same with DEVICE starts / lens (no host copy, no synchronisation): the sync-free multi-GPU gather */
int fs2_op_unpack_rows_dev(void *stream, const float *src, int32_t W, int32_t B, const int32_t *starts_dev, const int32_t *lens_dev,
                           int32_t Lout, float *dst);

/* dst [W, N] <- src [N, W]^T : packed mel frames [sum L, 80] -> vocoder layout [80, sum L] (the reference does
 * mel.transpose + np.concatenate on the host, inference.py:173-178; MelGAN takes [1, 80, L], utils/plot.py:96-105) */
int fs2_op_transpose(void *stream, const float *src, int64_t N, int32_t W, float *dst);

/* idx[i] = bucketize(x[i], bins[nb]) (right=False, NaN -> nb)  (variance_predictor.py:158,231) */
int fs2_op_bucketize(void *stream, const float *x, int64_t n, const float *bins, int32_t nb, int32_t *idx);

/* d[i] = clamp(round_half_even(exp(d_log[i]) - 1), 0) as int64: the duration predictor's inference post-op
 * (duration_predictor.py:77-81) on a flat array.  NaN -> 0; exp overflow saturates to INT64_MAX (torch's
 * .long() of +inf is implementation defined). */
int fs2_op_duration(void *stream, const float *d_log, int64_t n, int64_t *d);

/* Kernel-choice switches for A/B measurements and tests ("FS2_BM", "FS2_ROW8", "FS2_QKV8", "FS2_NOSPLITK",
 * "FS2_F32_ROWS", "FS2_MT8", "FS2_FUSE_VAR", "FS2_BAL", "FS2_ATTN_W32", "FS2_ROW4", "FS2_MT4", "FS2_QKV4", "FS2_FFN2_MX", "FS2_POST_MX"; -1 = automatic).  Their initial values come from the environment variables of the same
 * names, read once when the library is first used; the launch path never reads the environment. */
int fs2_set_option(const char *name, int32_t value);

/* Reads a switch back (the value fs2_set_option / the environment left: -1 = automatic), or one of the read-only facts about
 * THIS binary a benchmark line should carry:
 *   "FS2_AUDIT_CLEAN"   1 if a clean ISA-audit record of exactly this binary (libfs2_hip.audit.json next to it, tied to the
 *                       file's SHA-256) was found when the library was first used, else 0.  The kernels whose accumulators
 *                       are literal registers (attn_w32, gemm_row4_bf16) run only in an audited binary: without the record
 *                       FS2_ATTN_W32 / FS2_ROW4 / FS2_QKV4 start at 0 for EVERY consumer of the library (ctypes, the
 *                       TorchScript op, a C program) and can only be switched on by an explicit fs2_set_option;
 *   "attn_w32_active" / "row4_active" / "qkv4_active"   1 if that kernel is allowed to run (its switch is not 0).
 * Unknown name: FS2_ERR_ARG. */
int fs2_get_option(const char *name, int32_t *value);

/* Cumulative event counters of a handle (device memory, read back with one blocking 8-byte copy behind `stream`):
 *   "attn_slow_path_waves"  waves of attn_w32 that left the fast path because a row's probabilities, relative to its FIRST key
 *                           tile's maximum, summed beyond 2^60 (peaked attention of a trained model: reference
 *                           core/attention.py:55-62 has no such notion, its softmax is one formula) and were recomputed by the
 *                           plain fp32 two-pass loop: correct, but ~30 x slower per wave.  0 on flat (random-init) attention.
 * reset != 0 zeroes the counter after reading it. */
int fs2_get_counter(fs2_handle *h, void *stream, const char *name, int64_t *value, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* FS2_H_ */
