from .language_model import (LlavaLlamaForCausalLM, LlavaConfig, VCoderLlavaLlamaForCausalLM, VCoderLlavaConfig,
                             VCoderDSLlavaLlamaForCausalLM, VCoderDSLlavaConfig)
from .projector import build_vision_projector, build_seg_projector, build_depth_projector
from .vision_tower import build_vision_tower
from .builder import load_pretrained_model
