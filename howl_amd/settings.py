"""Environment-variable settings, mirroring ``howl/settings.py:1-170`` (same class / field / env-var names).

The reference uses pydantic-v1 ``BaseSettings`` (field ``num_mels`` <- env ``NUM_MELS`` etc., read when the settings
object is first constructed).  This is a dependency-free restatement with the same lazy behaviour and ``reset()``.
"""
import json
import multiprocessing
import os
from typing import List, get_type_hints

__all__ = ["SETTINGS", "HowlSettings"]


class _EnvSettings:
    """Fields are class annotations with defaults; an upper-cased env var of the same name overrides the default."""

    def __init__(self, **overrides):
        hints = get_type_hints(type(self))
        for name, typ in hints.items():
            default = getattr(type(self), name, None)
            value = overrides.get(name, default)
            env = os.environ.get(name.upper(), os.environ.get(name))
            if name not in overrides and env is not None:
                value = self._parse(env, typ)
            if isinstance(value, list):
                value = list(value)
            setattr(self, name, value)

    @staticmethod
    def _parse(text, typ):
        if typ is bool:
            return text.strip().lower() in ("1", "true", "yes", "on")
        if typ is int:
            return int(text)
        if typ is float:
            return float(text)
        if typ is str:
            return text
        return json.loads(text)  # List[...] fields are JSON, as pydantic parses complex env values

    def dict(self):
        return {k: getattr(self, k) for k in get_type_hints(type(self))}


class ResourceSettings(_EnvSettings):
    cpu_count: int = max(multiprocessing.cpu_count() // 2, 1)


class CacheSettings(_EnvSettings):
    cache_size: int = 128144


class AudioSettings(_EnvSettings):
    sample_rate: int = 16000
    use_mono: bool = True


class AudioTransformSettings(_EnvSettings):
    num_fft: int = 512
    num_mels: int = 80
    sample_rate: int = 16000
    hop_length: int = 200
    use_meyda_spectrogram: bool = False


class InferenceEngineSettings(_EnvSettings):
    inference_weights: List[float] = None
    inference_sequence: List[int] = [0]
    inference_window_ms: float = 2000
    smoothing_window_ms: float = 50
    tolerance_window_ms: float = 500
    inference_threshold: float = 0


class TrainingSettings(_EnvSettings):
    seed: int = 0
    vocab: List[str] = ["fire"]
    num_epochs: int = 10
    num_labels: int = 2
    learning_rate: float = 1e-3
    device: str = "cuda:0"
    batch_size: int = 16
    lr_decay: float = 0.955
    max_window_size_seconds: float = 0.75
    eval_window_size_seconds: float = 0.75
    eval_stride_size_seconds: float = 0.063
    weight_decay: float = 0
    convert_static: bool = False
    objective: str = "frame"
    token_type: str = "word"
    phone_dictionary: str = None
    use_noise_dataset: bool = False
    noise_dataset_path: str = None


class DatasetSettings(_EnvSettings):
    dataset_path: str = None


KEY_TO_SETTINGS_CLASS = {
    "_audio": AudioSettings,
    "_audio_transform": AudioTransformSettings,
    "_inference_engine": InferenceEngineSettings,
    "_dataset": DatasetSettings,
    "_cache": CacheSettings,
    "_training": TrainingSettings,
    "_resource": ResourceSettings,
}


class HowlSettings:
    """Lazy container (``settings.py:80-157``): each group is built from the environment on first access."""

    def __init__(self):
        for key in KEY_TO_SETTINGS_CLASS:
            setattr(self, key, None)

    def _get(self, key):
        if getattr(self, key) is None:
            setattr(self, key, KEY_TO_SETTINGS_CLASS[key]())
        return getattr(self, key)

    resource = property(lambda self: self._get("_resource"))
    audio = property(lambda self: self._get("_audio"))
    audio_transform = property(lambda self: self._get("_audio_transform"))
    inference_engine = property(lambda self: self._get("_inference_engine"))
    dataset = property(lambda self: self._get("_dataset"))
    cache = property(lambda self: self._get("_cache"))
    training = property(lambda self: self._get("_training"))

    def reset(self):
        for key, cls in KEY_TO_SETTINGS_CLASS.items():
            setattr(self, key, cls())

    def __repr__(self):
        rep = "Howl Settings:\n"
        for key in KEY_TO_SETTINGS_CLASS:
            grp = getattr(self, key)
            rep += f"\t{key}:" + (" None\n" if grp is None else "\n" + "".join(f"\t\t{k}: {v}\n" for k, v in grp.dict().items()))
        return rep


SETTINGS = HowlSettings()
