"""Placeholder ids / tokens of the reference prompt format (values fixed by vcoder_llava/constants.py:1-12;
they are part of the checkpoint/prompt contract, not an implementation choice)."""
LOGDIR = "."
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
SEG_TOKEN_INDEX = -300
DEFAULT_SEG_TOKEN = "<seg>"
DEPTH_TOKEN_INDEX = -400
DEFAULT_DEPTH_TOKEN = "<depth>"
