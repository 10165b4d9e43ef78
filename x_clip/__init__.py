"""Import-name alias: lets code written against the reference package run unchanged on the MI355X path.

    from x_clip import CLIP, TextTransformer            # resolves to x_clip_amd.CLIP / x_clip_amd.TextTransformer
    from x_clip.distributed import all_gather           # x_clip_amd.distributed
    from x_clip.visual_ssl import SimSiam, SimCLR       # x_clip_amd.visual_ssl
    from x_clip.mlm import MLM                          # x_clip_amd.mlm
    from x_clip.tokenizer import tokenizer              # x_clip_amd.tokenizer (needs a BPE merges file: see that module)

Nothing is implemented here: every name is the object of the same name in `x_clip_amd`, whose constructor keywords, forward
signature, state_dict keys and assertion messages mirror the reference.  Put the repository root in front
of a site-packages install of the reference on `sys.path` (or do not install the reference) for this alias to take effect.
"""
import sys as _sys

import x_clip_amd as _impl
from x_clip_amd import distributed as _distributed, mlm as _mlm, tokenizer as _tokenizer, visual_ssl as _visual_ssl

for _name in getattr(_impl, "__all__", [n for n in dir(_impl) if not n.startswith("_")]):
    globals()[_name] = getattr(_impl, _name)

_sys.modules[__name__ + ".distributed"] = _distributed
_sys.modules[__name__ + ".mlm"] = _mlm
_sys.modules[__name__ + ".visual_ssl"] = _visual_ssl
_sys.modules[__name__ + ".tokenizer"] = _tokenizer
distributed, mlm, visual_ssl, tokenizer = _distributed, _mlm, _visual_ssl, _tokenizer
