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
 *   - every call takes the hipStream_t to enqueue on and is re-entrant.
 */
#ifndef HOWL_HIP_H
#define HOWL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

#define HOWL_MAX_MELS 48 /* mel bins supported by the MFMA contraction (3 tiles of 16); BASELINE configs use 40 */
#define HOWL_FB_COLS 48  /* column count of a packed filterbank: (260, 48) fp32, zero padded */
#define HOWL_FB_PACKED_FLOATS (260 * HOWL_FB_COLS)

int howl_version(int* major, int* minor);
const char* howl_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Frontend: howl/data/transform/transform.py:234-296 (StandardAudioTransform) and the torchaudio
 * MelSpectrogram / ComputeDeltas it calls; howl/data/transform/operator.py:119-146 (ZmuvTransform).
 * ------------------------------------------------------------------------------------------------- */

/* (257, M) mel filterbank -> packed (260, 48) operand of howl_logmel_fwd.
 * Replaces the `.to(device)` of the CPU-built matrix at transform.py:435-443. */
int howl_fb_pack(const float* fb, int M, float* fbp, hipStream_t stream);

/* Build the packed filterbank on the device from the M+2 triangle corner frequencies (host struct, passed
 * by value to the kernel: no H2D copy, no sync).  The corner points carry the VTLP warp of
 * transform.py:394-401; the triangle arithmetic is transform.py:402-409. */
typedef struct {
    float f[HOWL_MAX_MELS + 2];
} HowlMelPoints;
int howl_fb_from_points(const HowlMelPoints* pts /* host */, int M, float nyquist, float* fbp, hipStream_t stream);

/* Fused reflect-pad + Hann + rFFT-512 (hop 200) + |.|^2 + mel contraction + log(x + log_eps) [+ ZMUV].
 *   pcm (B, L) row stride ld;  T = 1 + L/200 frames;  zmuv = {mean, std} device pair or NULL
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
/* pair = {mean, sqrt(mean2 - mean^2)}: operator.py:141-143 (`ZmuvTransform.std`). */
int howl_zmuv_pair(const float* mean, const float* mean2, float* pair, hipStream_t stream);

/* SpecAugment masks with host-drawn parameters (per sample; width <= 0 = no mask): transform.py:309-327. */
int howl_specaug_mask(float* x, int B, int C, int M, int T, const int* f0, const int* f, const int* t0,
                      const int* t, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOWL_HIP_H */
