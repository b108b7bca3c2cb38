/* howl_hip.h -- C ABI of libhowl_hip.so: the MI355X (gfx950) implementation of Howl's audio hot path.
 *
 * The reference (castorini/howl) has no FFI: its hot path is a chain of stock torch / torchaudio ops
 * reached through Python modules.  This header is the seam inserted *beneath* those modules; every entry
 * point names the reference call site(s) it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns 0 (HOWL_OK) or a negative HOWL_E_* code; howl_last_error() gives the message
 *     (thread-local).  No function allocates, frees, synchronises the device or takes ownership: the caller
 *     (PyTorch's caching allocator on the host side) owns every pointer, including workspaces.
 *   - all pointers are DEVICE pointers unless marked "host"; all tensors are contiguous fp32 unless noted.
 *   - every call takes the hipStream_t to enqueue on and is re-entrant.  All work is launched on that stream and on no
 *     other (the calls can be captured into a hipGraph); the library owns no streams.
 */
#ifndef HOWL_HIP_H
#define HOWL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define HOWL_FB_COLS 48  /* column count of one packed filterbank BANK: (260, 48) fp32, zero padded -- what one pass of the mel
                          * contraction covers (12 groups of 4 bins) */
#define HOWL_MAX_MELS 96 /* mel bins supported: up to two banks.  BASELINE configs use 40, the reference's stock NUM_MELS is 80
                          * (howl/settings.py:32): more than 48 bins run as two passes over the spectrum, bins [0, lo) and [lo, M),
                          * lo = 4 * ceil(M / 8) (80 -> 40 + 40) */
/* A packed bank is (260, 48) row-major floats followed by the same matrix in the fragment order of the mel
 * contraction: the banded LDS image (17 bin slots x 64 lanes x 4 mel groups), every (slot, group) fragment
 * (17 x 12 x 64 lanes), and 32 int32 words of flags.  A packed filterbank is howl_fb_packed_floats(M) floats: one bank
 * (M <= 48) or two, back to back. */
#define HOWL_FB_PACKED_FLOATS (260 * HOWL_FB_COLS + 17 * 64 * 4 + 17 * (HOWL_FB_COLS / 4) * 64 + 32)

int howl_version(int* major, int* minor);
const char* howl_last_error(void);

/* Optional kernel timing for bench.py's roofline leg: while enabled, the dominant kernels (tags "logmel", "conv0_fwd",
 * "conv3x3_fwd", "conv3x3_dgrad", "wgrad"; "lstm_fwd", "lstm_bwd", "gemm"; "mb_gemm", "mb_sweep") are bracketed by hipEvents
 * recorded on their launch stream.  howl_profile_read synchronises those events and returns the summed duration and
 * launch count for one tag; howl_profile_read_work also returns the summed algorithmic work (FLOPs for the matrix
 * kernels, HBM bytes for the "mb_sweep" activation sweeps) the call sites attached to those launches. */
int howl_profile_enable(int on);
int howl_profile_read(const char* tag, double* total_ms, int* count, int reset);
int howl_profile_read_work(const char* tag, double* total_ms, int* count, double* work, int reset);

/* Releases what the library keeps for the life of the process (the HIP events of howl_profile_enable).  Call once before
 * the HIP runtime goes away (howl_amd/lib.py registers it with atexit). */
int howl_shutdown(void);

/* ---------------------------------------------------------------------------------------------------
 * Frontend: howl/data/transform/transform.py:234-296 (StandardAudioTransform) and the torchaudio
 * MelSpectrogram / ComputeDeltas it calls; howl/data/transform/operator.py:119-146 (ZmuvTransform).
 * ------------------------------------------------------------------------------------------------- */

/* (257, M) mel filterbank -> packed operand of howl_logmel_fwd (howl_fb_packed_floats(M) floats; the first 260 x 48 of a bank
 * are its zero-padded columns themselves).
 * Replaces the `.to(device)` of the CPU-built matrix at transform.py:435-443. */
size_t howl_fb_packed_floats(int M);
int howl_fb_pack(const float* fb, int M, float* fbp, hipStream_t stream);

/* Build the packed filterbank on the device from the M+2 triangle corner frequencies (host struct, passed
 * by value to the kernel: no H2D copy, no sync).  The corner points carry the VTLP warp of
 * transform.py:394-401; the triangle arithmetic is transform.py:402-409. */
typedef struct {
    float f[HOWL_MAX_MELS + 2];
} HowlMelPoints;
int howl_fb_from_points(const HowlMelPoints* pts /* host */, int M, float nyquist, float* fbp, hipStream_t stream);

/* Fused reflect-pad + Hann + rFFT-512 (hop 200) + |.|^2 + mel contraction + log(x + log_eps) [+ ZMUV].
 *   pcm (B, L) row stride ld (rows may overlap: ld < L gives strided windows of one clip);  T = 1 + L/200 frames;  zmuv = {mean, std} device pair or NULL
 *   layout 0: out (B, M, T)  -- the reference's `mels_only` tensor (transform.py:275-277)
 *   layout 1: out (B, T, M)  -- the (time, frequency) layout res8 consumes (cnn.py:128-129), no permute needed
 * Replaces transform.py:249-254 + :275 (MelSpectrogram, add_(1e-7).log_()) and operator.py:145-146. */
int howl_logmel_fwd(const float* pcm, int B, int L, long ld, const float* fbp, int M, float log_eps,
                    const float* zmuv, float* out, int layout, hipStream_t stream);

/* (B, M, T) raw log-mels -> (B, 3, M, T) = stack(log-mel, deltas, accels), each optionally ZMUV-normalised.
 * Replaces transform.py:278-280 (ComputeDeltas x2 + torch.stack) and operator.py:145-146. */
int howl_deltas_fwd(const float* logmel, int B, int M, int T, const float* zmuv, float* out3, hipStream_t stream);

/* Running scalar mean / mean-of-squares update over n elements: operator.py:126-135 (`ZmuvTransform.update`).
 * scratch2: 2 doubles of device scratch. */
int howl_zmuv_update(const float* x, size_t n, float* total, float* mean, float* mean2, double* scratch2,
                     hipStream_t stream);
/* The same with a mask expanded to x's shape (operator.py:128-130): sums run over x*mask, the element count is the sum of the
 * caller's UNEXPANDED mask = (sum of the expanded mask) * count_scale, count_scale = mask.numel() / x.numel() before the
 * broadcast (1 for a full-shape mask) -- taken on the device, no host sync.  scratch3: 3 doubles of device scratch. */
int howl_zmuv_update_masked(const float* x, const float* mask, size_t n, double count_scale, float* total, float* mean,
                            float* mean2, double* scratch3, hipStream_t stream);
/* pair = {mean, sqrt(mean2 - mean^2)}: operator.py:141-143 (`ZmuvTransform.std`). */
int howl_zmuv_pair(const float* mean, const float* mean2, float* pair, hipStream_t stream);

/* out = (x - pair[0]) / pair[1] elementwise (in place allowed): operator.py:145-146 (`ZmuvTransform.forward`). */
int howl_zmuv_apply(const float* x, size_t n, const float* pair, float* out, hipStream_t stream);

/* Collate + waveform augmentation on the device: out[b] = pad(noise(timeshift(bank[idx[b]][:src_len[b]]))), (B, Lout).
 * Replaces, for device-resident clips, the DataLoader-worker chain truncate_length -> TimeshiftTransform ->
 * NoiseTransform -> batchify (operator.py:73-86, transform.py:120-196).  Per-sample parameters are drawn on the host with
 * the reference's RNG protocol (shift / head-or-tail, white sigma, salt-pepper prob; 0 disables); the noise samples
 * themselves come from a counter-based generator keyed by (seed, b, n), so they match torch's only in distribution. */
int howl_collate_augment(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                         const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed, int B,
                         int Lout, float* out, hipStream_t stream);
/* The same with DatasetMixer (transform.py:199-231) in front of the chain, as train.py:218 composes it: before the time
 * shift, sample n of clip b becomes x*(1-alpha[b]) + bg[bg_idx[b]][bg_off[b] + n]*alpha[b] (alpha 0 = not mixed, 1 =
 * replaced).  The background clip, its offset (rand.randint(len, bg_len) - len) and alpha are host draws in the
 * reference's order.  bg == NULL: identical to howl_collate_augment. */
int howl_collate_augment_mix(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                             const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed,
                             const float* bg, long bg_ld, const int* bg_idx, const int* bg_off, const float* alpha, int B,
                             int Lout, float* out, hipStream_t stream);

/* The same chain feeding WakeWordFrameBatchifier (train.py:211-225 composes Timeshift -> Noise -> batchifier): row b holds
 * the samples [shift[b], src_len[b]) (from_head = 1) of the mixed / noised clip at columns dst_off[b].., zeros on both
 * sides, i.e. a window of the augmented clip padded on the side tensorize_audio_data(rand_append=True) drew
 * (batchifier.py:108-115, operator.py:104-107).  dst_off == NULL: identical to howl_collate_augment_mix. */
int howl_collate_augment_window(const float* bank, long bank_ld, const int* idx, const int* src_len, const int* shift,
                                const int* from_head, const float* sigma, const float* sp_prob, unsigned long long seed,
                                const float* bg, long bg_ld, const int* bg_idx, const int* bg_off, const float* alpha,
                                const int* dst_off, int B, int Lout, float* out, hipStream_t stream);

/* Frame-window gather: out (B, Lout) row b = zeros except out[b][dst_off[b] + n] = bank[idx[b]][start[b] + n], n < len[b].
 * Replaces, for device-resident clips, the window cut of WakeWordFrameBatchifier.__call__ (batchifier.py:56-118:
 * `ex.audio_data[..., a:b]`) and the zero padding of tensorize_audio_data(rand_append=True, max_length=window)
 * (operator.py:89-109: zeros before or after the samples); which window, and which side, are host draws in the
 * reference's order (howl_amd/data/transform/batchifier.py).  Requires dst_off[b] + len[b] <= Lout. */
int howl_gather_windows(const float* bank, long bank_ld, const int* idx, const int* start, const int* len,
                        const int* dst_off, int B, int Lout, float* out, hipStream_t stream);

/* SpecAugment masks with host-drawn parameters (per sample; width <= 0 = no mask): transform.py:309-327.
 * x is a (B,C,M,T) view with element strides (sb,sc,sm,st), masked in place. */
int howl_specaug_mask(float* x, int B, int C, int M, int T, long sb, long sc, long sm, long st, const int* f0,
                      const int* f, const int* t0, const int* t, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * res8 classifier: howl/model/cnn.py:113-145 (Res8.forward) and its autograd backward, as driven by
 * training/run/pretrain_gsc.py:126-133 and training/run/train.py:288-302.
 * Parameter tensors keep the reference's state_dict shapes (howl/workspace.py:31-67 round-trips them).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* conv0_w;           /* conv0.weight (45,1,3,3) */
    const float* conv_w[6];         /* conv{1..6}.weight (45,45,3,3) */
    float* bn_running_mean[6];      /* bn{1..6}.running_mean (45): updated in place when training */
    float* bn_running_var[6];       /* bn{1..6}.running_var (45) */
    long long* bn_num_batches[6];   /* bn{1..6}.num_batches_tracked: int64 scalars */
    const float* out_w;             /* output.weight (C,45) */
    const float* out_b;             /* output.bias (C) */
} HowlRes8Params;

typedef struct {
    float* conv0_w;
    float* conv_w[6];
    float* out_w;
    float* out_b;
} HowlRes8Grads;

/* activations the backward pass needs; all caller-allocated */
typedef struct {
    float* s[7];       /* s[0] = avgpool(relu(conv0(x))); s[i] = layer i's pre-BatchNorm output; each howl_res8_saved_floats(B,T,M) floats = (B,45,T/3,M/4) for T <= 83.
                        * Values are >= 0 by construction; for i = 2,4,6 the sign bit carries the ReLU mask of conv_i
                        * (the backward needs it), so readers take |s|.  M = 40: the reference's NCHW layout; M = 80: the
                        * library's own -- utterance b is the two blocks 2b, 2b+1 of (45,T/3,10): pooled columns 0..9, 10..19;
                        * T > 83: nr = ceil(T/3 / 27) row strips of Hs = ceil(T/3 / nr) pooled rows, block (b * nr + r) * (M/40) + c. */
    float* bn_stats;   /* (6, 2, 48): per layer {mean[48], rstd[48]} used by this forward */
    float* pooled;     /* (B, 48): spatial mean of BN6's output */
    unsigned short* mask0; /* (B,45,T/3,M/4) uint16, laid out like s[0]: ReLU pattern of conv0's 3x4 pre-pool window (bit 4i+j); NULL in eval */
} HowlRes8Saved;

/* M = NUM_MELS (howl/settings.py:32): 40 (every res8 preset, envs/res8.env) or 80 (the stock default).  The 3x3 kernels keep
 * 10 pooled columns of one utterance on chip; at 80 bins an utterance runs as two such strips that fetch each other's edge
 * column.  howl_res8_workspace_bytes(B, T) is the M = 40 size. */
size_t howl_res8_workspace_bytes(int B, int T);
size_t howl_res8_workspace_bytes_mels(int B, int T, int M);
/* Floats of ONE saved activation tensor (HowlRes8Saved.s[i]; mask0 has as many uint16).  B*45*(T/3)*(M/4) up to 83 frames; longer
 * utterances (cnn.py:127-145 takes any T) run as row strips of equal height, the last one padded, so the tensors are a little
 * larger: always size them with this. */
size_t howl_res8_saved_floats(int B, int T, int M);
/* Eval mode (training = 0) needs less: the workspace only up to the pooled sums (no backward buffers), and layer i reads
 * s[i-1], s[i-2] while it writes s[i], so THREE activation buffers in rotation (s[i] = buffer i mod 3) may stand in for the
 * seven; mask0 is still written.  howl_res8_row_strips(T): the row strips a T-frame input runs as (howl_res8_fwd / _bwd take
 * up to 1024 = 82,944 frames). */
size_t howl_res8_eval_workspace_bytes_mels(int B, int T, int M);
int howl_res8_row_strips(int T);

/* feat: log-mel features, element (b, t, m) at feat[b*sb + t*st + m*sm] (so both the (B,T,M) model layout and
 * channel 0 of the reference's (B,3,M,T) tensor are accepted; replaces x[:, :1].permute(0,1,3,2), cnn.py:128-129).
 * Any T >= 3 (cnn.py:127-145), M = 40 or 80; sizes from howl_res8_workspace_bytes_mels / howl_res8_saved_floats.
 * training != 0: BatchNorm uses batch statistics and updates the running buffers; else uses the running buffers.
 * logits: (B, C). */
int howl_res8_fwd(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  int training, const HowlRes8Saved* saved, float* logits, void* ws, size_t ws_bytes,
                  hipStream_t stream);

/* Inference-mode forward for inputs longer than the kernels' on-chip map (T > 83 frames, up to ~2,500): the clip is cut
 * into overlapping 27-row windows whose interiors tile it exactly (the six 3x3 convolutions spread a window's padding 7 pooled
 * rows inwards), the windows run as a virtual batch and only the final spatial mean sees them together.  Same arguments
 * as howl_res8_fwd without `training` (running statistics are used) and without saved activations; results equal
 * cnn.py:127-145 on the whole clip (ConvertedStaticModel's first window, base.py:52-62, and engine clips > 1 s). */
size_t howl_res8_long_workspace_bytes(int B, int T);          /* M = 40 */
size_t howl_res8_long_workspace_bytes_mels(int B, int T, int M);
int howl_res8_fwd_long(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       float* logits, void* ws, size_t ws_bytes, hipStream_t stream);

/* backward of a training-mode forward: dlogits (B,C) -> parameter gradients (overwritten, not accumulated).
 * Replaces loss.backward() through cnn.py:127-145 (pretrain_gsc.py:131, train.py:294).  Per layer, the data gradient and
 * the weight gradient share one launch (half of the CUs each); summation orders are fixed: repeated calls are bit-identical. */
int howl_res8_bwd(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                  const HowlRes8Saved* saved, const float* dlogits, const HowlRes8Grads* grads, void* ws,
                  size_t ws_bytes, hipStream_t stream);
/* The same pass in two calls for data-parallel steps (same arguments, part 1 then part 2; part 0 = howl_res8_bwd):
 * after part 1 grads->conv_w[0..5], out_w, out_b are final, so their all-reduce can run under part 2 (conv0's weight
 * gradient -> grads->conv0_w).  Same launches and the same bits as the single call, plus one small fold launch. */
int howl_res8_bwd_part(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* saved, const float* dlogits, const HowlRes8Grads* grads, void* ws,
                       size_t ws_bytes, int part, hipStream_t stream);

/* The same step as an argument of a backward call that ends in a fold of ALL of the model's gradients (howl_res8_bwd_xent, howl_seq_lstm_bwd): the
 * fold applies it to each gradient element as it writes it (one launch and one pass over the gradients fewer).  p / g / m / v: the
 * flat buffers of n floats the call's gradient pointers lie in. */
typedef struct {
    float* p;
    float* g;
    float* m;
    float* v;
    size_t n;
    float lr, beta1, beta2, eps, weight_decay;
    int step;
    float grad_scale;
} HowlAdamW;

/* The training step's forward + nn.CrossEntropyLoss() (pretrain_gsc.py:126-133) with the loss inside the forward's last launch:
 * howl_res8_fwd (training mode) that also writes, per utterance, nll[b] = logsumexp(logits[b]) - logits[b][labels[b]] and
 * dlogits[b] = (softmax(logits[b]) - onehot) / B, and leaves the pooled gradient in the workspace; C <= 64.  The backward
 * pass that follows is howl_res8_bwd_xent: howl_res8_bwd_part (same `part` convention, same workspace) without its first
 * launch, and with loss[0] = mean_b nll[b].  Same arithmetic in the same order as howl_res8_fwd + howl_xent_fwd_bwd +
 * howl_res8_bwd (bit-identical logits, loss and gradients), two launches fewer. */
int howl_res8_fwd_xent(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* saved, const long long* labels, float* logits, float* nll, float* dlogits,
                       void* ws, size_t ws_bytes, hipStream_t stream);
int howl_res8_bwd_xent(const HowlRes8Params* prm, const float* feat, long sb, long st, long sm, int B, int T, int M, int C,
                       const HowlRes8Saved* saved, const float* dlogits, const float* nll, float* loss,
                       const HowlRes8Grads* grads, void* ws, size_t ws_bytes, int part,
                       const HowlAdamW* adamw /* NULL: gradients only; else part must be 0 */, hipStream_t stream);

/* mean cross-entropy over (B,C) logits with int64 labels and its gradient (dlogits may be NULL):
 * nn.CrossEntropyLoss() at pretrain_gsc.py:95,131 / train.py:251,293. */
int howl_xent_fwd_bwd(const float* logits, const long long* labels, int B, int C, float* loss, float* dlogits,
                      hipStream_t stream);

/* Fused log_softmax + CTC loss and its gradient w.r.t. the logits:
 *     scores = log_softmax(model(...), -1); loss = nn.CTCLoss(blank)(scores, targets, input_lengths, target_lengths)
 * at training/run/train.py:250-256,291-296 (sequence objective, envs/seq-lstm.env), reduction "mean", zero_infinity False.
 * logits: (T, B, C) addressed as t*st_t + b*st_b + c (so the model's (B,T,C) buffer can be passed as its (T,B,C) view);
 * targets: (B, >= max_target_length) int64, row stride tgt_stride; input_lengths / target_lengths: (B) int64, device.
 * nll: (B) per-utterance negative log likelihood; loss: (1) mean_b nll_b / max(target_length_b, 1), or NULL to leave the
 * mean to howl_head_bwd's HowlCtcMean rider (one launch fewer);
 * dlogits (nullable): d loss / d logits, addressed as t*dst_t + b*dst_b + c, rows t >= input_length_b are zero.
 * Range: howl_ctc_supported(T, C, max_target_length) -- T <= 8192 frames, C <= 64, targets <= 31 labels; outside it the call
 * returns HOWL_E_ARG (the host side raises: the training path has no vendor fallback).  Up to 128 frames an utterance's
 * rows stay in LDS; longer ones (whole clips: AudioSequenceBatchifier, howl/data/transform/batchifier.py:14-34, with the loss
 * over all their frames, train.py:291-296) are walked in windows of 128 frames and, when the gradient is wanted, keep their
 * alpha rows in `workspace`: howl_ctc_workspace_floats(T, B) floats (0 up to 128 frames; workspace may then be NULL). */
int howl_ctc_supported(int T, int C, int max_target_length);
size_t howl_ctc_workspace_floats(int T, int B);
int howl_ctc_loss(const float* logits, long st_t, long st_b, int T, int B, int C, const long long* targets, long tgt_stride,
                  int max_target_length, const long long* input_lengths, const long long* target_lengths, int blank,
                  float* nll, float* loss, float* dlogits, long dst_t, long dst_b, float* workspace, size_t workspace_floats,
                  hipStream_t stream);

/* Fused AdamW over one flat buffer (torch.optim.AdamW semantics; pretrain_gsc.py:93,133, train.py:256,302).
 * step counts from 1; grad_scale multiplies g on the fly (1/world_size after a sum all-reduce). */
int howl_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, float grad_scale, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * LSTM classifiers: howl/model/rnn.py:41-91 (SequentialLstm "seq-lstm", SimpleLstm "lstm"):
 * pack_padded_sequence + nn.LSTM(40,128) (rnn.py:65-66,88) and the Linear-ReLU-Linear head (rnn.py:44-48,71,91).
 * Batch-major internal layout: x (B,T,M) log-mels; lengths int64 (B) or NULL (= T for all).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* w_ih; /* lstm.weight_ih_l0 (512, M) */
    const float* w_hh; /* lstm.weight_hh_l0 (512, 128) */
    const float* b_ih; /* lstm.bias_ih_l0 (512) */
    const float* b_hh; /* lstm.bias_hh_l0 (512) */
} HowlLstmParams;

typedef struct {
    float* w_ih;
    float* w_hh;
    float* b_ih;
    float* b_hh;
} HowlLstmGrads;

typedef struct {
    float* gx;     /* (B,T,512) input projection x W_ih^T + b_ih + b_hh; may be NULL when howl_lstm_needs_gx() says 0 */
    float* gates;  /* (B,T,512) gate activations i,f,g,o */
    float* c;      /* (B,T,128) cell states */
    float* hseq;   /* (B,T+1,128): hseq[b][0] = h0, hseq[b][t+1] = h_t (zero for t >= length, as pad_packed_sequence) */
    float* dgates; /* (B,T,512) gate pre-activation gradients (backward scratch) */
    int t_out;     /* number of steps to run = max(lengths) (host int; the reference syncs for it too) */
    int x_frames;  /* frames per utterance in x's buffer (>= T): row (b, t) of x starts at (b * x_frames + t) * M floats, so
                      that the first T frames of a longer feature buffer are used in place; 0 = T */
} HowlLstmSaved;

size_t howl_lstm_workspace_bytes(int B, int T);
/* 1 when howl_lstm_fwd on this shape writes the (B,T,512) projection buffer saved->gx, 0 when the recurrence multiplies
   x_t W_ih^T itself (four-sequence kernel, M = 40) and gx may be NULL.  x_frames as in HowlLstmSaved (0 = T). */
size_t howl_lstm_needs_gx(const HowlLstmParams* p, int B, int T, int M, int x_frames);
/* h0/c0: (B,128) initial state or NULL (zeros) -- the streaming carry of rnn.py:62,67-68; hT/cT: final state (B,128). */
int howl_lstm_fwd(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths,
                  const float* h0, const float* c0, const HowlLstmSaved* saved, float* hT, float* cT, void* ws,
                  size_t ws_bytes, hipStream_t stream);
/* howl_lstm_fwd + the frontend of the NEXT batch (howl_logmel_fwd's arguments): where the recurrence leaves CUs idle (four
 * sequences per workgroup: (B + 3) / 4 workgroups <= half of the device) the frontend runs as additional blocks of the
 * recurrence's launch -- independent work, read by later calls only -- and as its own launch behind the recurrence otherwise;
 * either way next->out holds the features when the call's work has run.  The training loop's one-batch look-ahead
 * (FusedTrainer.step_sequence(next_audio=...)). */
typedef struct {
    const float* pcm;
    int B, L;
    long ld;
    const float* fbp;
    int M;
    float log_eps;
    const float* zmuv;
    float* out;
    int layout;
} HowlLogmelArgs;
int howl_lstm_fwd_next(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths,
                       const float* h0, const float* c0, const HowlLstmSaved* saved, float* hT, float* cT, void* ws,
                       size_t ws_bytes, const HowlLogmelArgs* next, hipStream_t stream);
/* dy: (B,T,128) gradient w.r.t. the padded outputs or NULL; dhT/dcT: (B,128) gradient w.r.t. the final state or NULL. */
int howl_lstm_bwd(const HowlLstmParams* p, const float* x, int B, int T, int M, const long long* lengths,
                  const float* c0, const HowlLstmSaved* saved, const float* dy, const float* dhT, const float* dcT,
                  const HowlLstmGrads* grads, void* ws, size_t ws_bytes, hipStream_t stream);

/* The classifier head of both LSTM models as one unit: Linear(n_in, n_hid) - ReLU - Linear(n_hid, n_out)
 * (replaces `self.dnn` of howl/model/rnn.py:44-48 applied at rnn.py:71 / rnn.py:91).  x has `rows` rows of n_in contiguous
 * floats; row r lives at (r / rows_inner) * s_outer + (r % rows_inner) * s_inner (elements), so (B,T+1,128) slices of the
 * hidden-state buffer are usable in place.
 * With n_out <= 8 and n_hid <= 256 the second layer runs as vector kernels on the hidden activations: its backward is ONE
 * pass that yields dz1 = (y1 > 0) * (dy2 W2), dW2, db2 and db1, folded with the first layer's split-K slabs in one launch;
 * other shapes take the library's generic GEMM path.  Same results either way up to fp32 summation order. */
typedef struct {
    const float* w1; /* dnn[0].weight (n_hid, n_in) */
    const float* b1; /* dnn[0].bias (n_hid) */
    const float* w2; /* dnn[2].weight (n_out, n_hid) */
    const float* b2; /* dnn[2].bias (n_out) */
} HowlHeadParams;
typedef struct {
    float* w1;
    float* b1;
    float* w2;
    float* b2;
} HowlHeadGrads;
/* Optional rider of howl_head_bwd in the sequence step: the batch mean of a CTC loss whose own mean launch was left out
 * (howl_ctc_loss with loss = NULL): loss[0] = mean_b nll[b] / max(target_lengths[b], 1), the same arithmetic as
 * howl_ctc_loss's, taken by one extra block of the head's backward launch. */
typedef struct {
    const float* nll;                 /* (B) per-utterance negative log-likelihoods from howl_ctc_loss */
    const long long* target_lengths;  /* (B) */
    int B;
    float* loss;                      /* (1) */
} HowlCtcMean;
size_t howl_head_workspace_bytes(int n_in, int n_hid, int n_out);
/* y1 (rows, n_hid) = relu(x W1^T + b1) [kept for the backward], y2 (rows, n_out) = y1 W2^T + b2. */
int howl_head_fwd(const HowlHeadParams* p, const float* x, int rows_inner, long s_outer, long s_inner, int rows, int n_in,
                  int n_hid, int n_out, float* y1, float* y2, hipStream_t stream);
/* dy2 (rows, n_out) contiguous -> grads, dx (rows, n_in) contiguous (NULL to skip); dz1 (rows, n_hid) is scratch the caller
 * provides (it holds the gradient at the hidden pre-activations afterwards). */
int howl_head_bwd(const HowlHeadParams* p, const float* x, int rows_inner, long s_outer, long s_inner, int rows, int n_in,
                  int n_hid, int n_out, const float* y1, const float* dy2, float* dz1, float* dx, const HowlHeadGrads* grads,
                  const HowlCtcMean* ctc_mean /* NULL: none */, void* ws, size_t ws_bytes, hipStream_t stream);
/* Backward of the whole sequence model (SequentialLstm, rnn.py:60-71: dnn(lstm(x))) in one call: howl_head_bwd on the hidden
 * states h_1 .. h_T read in place from saved->hseq, then howl_lstm_bwd on the gradient it leaves in dhs (B, T, 128) -- the same
 * arithmetic, but the three wide weight gradients (dnn[0].weight, W_ih, W_hh) run as ONE launch and all slab folds as one
 * (12 launches per seq-lstm training step instead of 15).  Requires saved->t_out == T; head_ws / ws as for the two calls.
 * adamw != NULL: the optimiser step on the flat buffers is part of the call (inside the fold when every gradient pointer lies in
 * adamw->g and together they cover it; as howl_adamw_step's launch behind it otherwise). */
/* Round 6: head forward + log_softmax / CTC + the head's backward over the rows, of the sequence model's training step, in ONE
 * launch -- dnn(rnn_seq) (rnn.py:71), log_softmax + CTCLoss(blank) and their backward (train.py:250-256,291-296) for batches whose
 * utterances fit on chip (howl_seq_head_ctc_supported: 128 -> 256 -> <= 8 labels, T <= ~80 frames, targets <= 8 labels, >= 2048
 * rows): a workgroup owns whole utterances, the hidden activations never leave LDS.  x: hidden rows, row (b, t) at
 * x + b * s_outer + t * s_inner; y2 (B, T, n_out) logits; nll (B); dz1 (B T, n_hid) and dhs (B, T, n_in) as howl_head_bwd leaves
 * them; the per-workgroup partial sums of dW2 / db1 / db2 stay in head_ws (howl_head_workspace_bytes) for howl_seq_lstm_bwd /
 * howl_head_bwd called with y1 = dy2 = NULL, which folds them, takes the first layer's weight gradient and (HowlCtcMean) the
 * batch mean of nll.  Same bits as howl_head_fwd + howl_ctc_loss + howl_head_bwd for y2, nll, dz1, dhs. */
int howl_seq_head_ctc_supported(int B, int T, int n_in, int n_hid, int n_out, int max_target_length);
int howl_seq_head_ctc(const HowlHeadParams* p, const float* x, long s_outer, long s_inner, int B, int T, int n_in, int n_hid, int n_out,
                      const long long* targets, long tgt_stride, int max_target_length, const long long* input_lengths,
                      const long long* target_lengths, int blank, float* y2, float* nll, float* dz1, float* dhs, void* head_ws,
                      size_t head_ws_bytes, hipStream_t stream);
int howl_seq_lstm_bwd(const HowlHeadParams* head, int n_hid, int n_out, const float* y1, const float* dy2, float* dz1, float* dhs,
                      const HowlHeadGrads* head_grads, const HowlCtcMean* ctc_mean /* NULL: none */, void* head_ws,
                      size_t head_ws_bytes, const HowlLstmParams* p, const float* x, int B, int T, int M,
                      const long long* lengths, const float* c0, const HowlLstmSaved* saved, const HowlLstmGrads* grads, void* ws,
                      size_t ws_bytes, const HowlAdamW* adamw /* NULL: gradients only */, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * MobileNetClassifier, registry name "mobilenet" (replaces howl/model/cnn.py:15-29: downsample conv/BN/ReLU/pool +
 * torchvision mobilenet_v2 + Linear(1280, num_labels); BASELINE configs[4]).
 * The layer table is owned by the library: layer 0 is `downsample`, layer 1 `features[0]`, then the 17 inverted
 * residual blocks, last `features[18]`.  Parameters live in ONE flat float buffer (per layer: conv weight in PyTorch's
 * own shape, [conv bias], BN weight, BN bias; then classifier weight (num_labels,1280) and bias) and the BatchNorm
 * running statistics in a second one (per layer: running_mean, running_var); gradients use the parameter layout.
 * State-dict key of layer i: feat < 0: "downsample.0" / "downsample.1"; sub < 0: "model.features.{feat}.0" / ".1";
 * wrapped: "model.features.{feat}.conv.{sub}.0" / ".{sub}.1"; else "model.features.{feat}.conv.{sub}" / ".conv.{sub+1}".
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
    int kind;    /* 0 dense 3x3, 1 pointwise 1x1, 2 depthwise 3x3 */
    int cin, cout, stride, pad_h, pad_w;
    int act;     /* 0 none, 1 ReLU6, 2 ReLU */
    int bias;    /* convolution has a bias (downsample only) */
    int pool;    /* MaxPool2d((1,2)) after the activation (downsample only) */
    int res_src; /* >= 0: output adds the output of that layer (inverted-residual skip) */
    int feat, sub, wrapped;
    long long w_off, b_off, gamma_off, beta_off; /* float offsets in the parameter buffer (b_off = -1: none) */
    long long rmean_off, rvar_off;               /* float offsets in the BN buffer */
} HowlMbLayer;
size_t howl_mobilenet_num_layers(void);
int howl_mobilenet_layer(int i, HowlMbLayer* out);
size_t howl_mobilenet_param_floats(int num_labels);
size_t howl_mobilenet_buffer_floats(void);
size_t howl_mobilenet_workspace_bytes(int B, int M, int T, int num_labels);
/* n keep flags (1.0 with probability 1 - p, else 0.0) from the counter-based device generator keyed by `seed`: the
 * drop_mask operand of howl_mobilenet_fwd / _bwd, replacing nn.Dropout's torch.rand chain (cnn.py:22 via torchvision's classifier). */
int howl_dropout_mask(float* mask, size_t n, float p, unsigned long long seed, hipStream_t stream);

/* x: element (b, mel, t) at x[b*sb + mel*sm + t*st] (the log-Mel channel of the (B,3,M,T) features).  training != 0:
 * batch statistics (running buffers updated, momentum 0.1) and, if drop_mask (B,1280 of 0/1) is given, dropout with
 * kept activations scaled by drop_scale = 1/(1-p).  The workspace keeps what howl_mobilenet_bwd needs: pass the SAME
 * workspace, inputs and mask to it.  logits: (B, num_labels). */
int howl_mobilenet_fwd(const float* params, float* buffers, int num_labels, const float* x, long sb, long sm, long st, int B,
                       int M, int T, int training, const float* drop_mask, float drop_scale, float* logits, void* ws,
                       size_t ws_bytes, hipStream_t stream);
/* dlogits (B, num_labels) -> grads (parameter layout, every entry written). */
int howl_mobilenet_bwd(const float* params, int num_labels, const float* x, long sb, long sm, long st, int B, int M, int T,
                       const float* drop_mask, float drop_scale, const float* dlogits, float* grads, void* ws, size_t ws_bytes,
                       hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOWL_HIP_H */
