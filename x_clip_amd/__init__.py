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
    (vision tower N = 512 products +12 ... +18 %, text +1 ... +2 %).  The second batch-size dependent choice goes with it: the
    latency-built 64 x 64 kernel (gemm_small.h) takes a product by its ROW COUNT (M a multiple of 64, at most 64 tiles, 2 M N K under
    xclip_gemm_small_limit) and sums its k-blocks in another order than the 256 x 256 kernels -- a pooled-layer or latent product at B = 1024
    would take it and at B = 1000 or 2048 would not -- so the switch also sets that limit to 0 (every product on the 256 x 256 kernels, whose
    per-tile summation order does not depend on M) and restores it when switched off.  Process-wide; -> the previous setting."""
    from . import ops
    global _small_limit_saved
    was = ops.BATCH_INVARIANT_GEMM
    ops.BATCH_INVARIANT_GEMM = bool(on)
    if on and not was:
        _small_limit_saved = ops.gemm_small_limit(0)
    elif was and not on and _small_limit_saved is not None:
        ops.gemm_small_limit(_small_limit_saved)
        _small_limit_saved = None
    return was


_small_limit_saved = None


__all__ = ["CLIP", "TextTransformer", "VisionTransformer", "set_batch_invariant"]
