"""``LattePipeline`` name kept for ``sample/sample_t2x.py`` (sample/pipeline_latte.py:100-115).

The text-to-video family (LatteT2V + T5 cross-attention + AutoencoderKLTemporalDecoder + diffusers
schedulers) is SURVEY.md §8(f) rank 2 — the next row after the class-conditional / unconditional
sampling path — and depends on diffusers 0.24.0, which is not vendored in the reference.  The class
exists so imports resolve; constructing it fails loudly instead of silently falling back."""
from ._lib import LatteError


class LattePipeline:
    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None):
        raise LatteError(
            "LattePipeline (Latte-1 text-to-video) is not part of the MI355X engine yet: the accelerated path is "
            "sample.py / sample_ddp.py (Latte_models + create_diffusion + VAE decode). See DESIGN.md, 'out of scope'.")
