"""MI355X-native reward-scoring forward path for LLaVA-Reward (Phi-3.5-V).

Drop-in surface (mirrors eval/reward_adaptor_loader.py of the reference):
    load_reward_adaptor, inference_process_phi3v, preference_compute, and a model object whose
    ``custom_forward`` runs the hand-written HIP path through the C-ABI in ``include/llava_reward_hip.h``.
"""
from . import synth  # noqa: F401
from .model import RewardModel  # noqa: F401
from .reward_adaptor_loader import inference_process_phi3v, load_reward_adaptor, preference_compute  # noqa: F401
from .preprocess import (batch_inference_process_phi3v_device, collate_rows, hd_transform_batch,  # noqa: F401
                         inference_process_phi3v_device, zero_pad_sequences)
from . import trainer_shim  # noqa: F401
from .scoring import score_candidates  # noqa: F401

__all__ = ["synth", "RewardModel", "load_reward_adaptor", "inference_process_phi3v", "preference_compute",
           "hd_transform_batch", "inference_process_phi3v_device", "batch_inference_process_phi3v_device", "collate_rows",
           "zero_pad_sequences", "trainer_shim", "score_candidates"]
