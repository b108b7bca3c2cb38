"""``stride`` of ``howl/utils/audio_utils.py:26-49`` (window generator used by FrameInferenceEngine)."""
import torch


def stride(audio_data: torch.Tensor, window_ms: int, stride_ms: int, sample_rate: int, drop_incomplete: bool = True):
    chunk_sz = int(window_ms / 1000 * sample_rate)
    stride_sz = int(stride_ms / 1000 * sample_rate)
    curr_idx = 0
    while curr_idx < audio_data.size(-1):
        sliced = audio_data[..., curr_idx: curr_idx + chunk_sz]
        if sliced.size(-1) != chunk_sz and drop_incomplete:
            return
        yield sliced
        curr_idx += stride_sz


def stride_starts(num_samples: int, window_ms: int, stride_ms: int, sample_rate: int):
    """Start offsets of the complete windows ``stride`` yields (for the batched engine)."""
    chunk_sz = int(window_ms / 1000 * sample_rate)
    stride_sz = int(stride_ms / 1000 * sample_rate)
    starts, curr = [], 0
    while curr < num_samples and curr + chunk_sz <= num_samples:
        starts.append(curr)
        curr += stride_sz
    return starts, chunk_sz
