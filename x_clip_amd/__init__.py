"""x_clip_amd -- MI355X-native (gfx950) drop-in for the lucidrains/x-clip contrastive-training path.

    from x_clip_amd import CLIP, TextTransformer        # mirrors `from x_clip import CLIP, TextTransformer`

Python host code (this package) over a C-ABI HIP library (include/xclip.h, built by `python -m x_clip_amd.build`).
"""
from .clip import CLIP, TextTransformer, VisionTransformer  # noqa: F401


def set_batch_invariant(on: bool = True) -> bool:
    """Make a sample's activations independent of what shares its batch, bit for bit (evaluation pipelines that compare per-sample
    embeddings across batch sizes).  By default the last partial round of a persistent GEMM launch runs as a split-K problem of its own
    (csrc/xclip_api.hip gemm2_tail_cut): which rows form that tail depends on the batch size, so up to ~16 samples of a batch are summed in
    another order (a few bf16 ulps).  With this switch on no forward / input-gradient product is cut, at the price of that last round
    (vision tower N = 512 products +12 ... +18 %, text +1 ... +2 %).  Process-wide; -> the previous setting."""
    from . import ops
    was = ops.BATCH_INVARIANT_GEMM
    ops.BATCH_INVARIANT_GEMM = bool(on)
    return was


__all__ = ["CLIP", "TextTransformer", "VisionTransformer", "set_batch_invariant"]
