"""vcoder_amd — MI355X-native VCoder / VCoder-DS LLaVA-1.5 inference hot path (gfx950 HIP kernels behind a C ABI).

Importing the package never touches the GPU; the first model construction loads libvcoder_hip.so and fails loudly
if it (or a GPU) is missing."""
__version__ = "0.1.0"
