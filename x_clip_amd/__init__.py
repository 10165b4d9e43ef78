"""x_clip_amd -- MI355X-native (gfx950) drop-in for the lucidrains/x-clip contrastive-training path.

    from x_clip_amd import CLIP, TextTransformer        # mirrors `from x_clip import CLIP, TextTransformer`

Python host code (this package) over a C-ABI HIP library (include/xclip.h, built by `python -m x_clip_amd.build`).
"""
from .clip import CLIP, TextTransformer, VisionTransformer  # noqa: F401

__all__ = ["CLIP", "TextTransformer", "VisionTransformer"]
