"""`from pcdet.utils.spconv_utils import spconv` boundary (reference pcdet/utils/spconv_utils.py:1-34):
here `spconv` IS the libspx operator package."""
from typing import Set

import torch.nn as nn

import spx as spconv


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    """Names of every sparse-conv weight (for the spconv 1.x -> 2.x layout fix at checkpoint load)."""
    found: Set[str] = set()
    for name, child in model.named_children():
        new_prefix = "%s.%s" % (prefix, name) if prefix != "" else name
        if isinstance(child, spconv.conv.SparseConvolution):
            found.add(new_prefix + ".weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found


def replace_feature(out, new_features):
    if "replace_feature" in out.__dir__():
        return out.replace_feature(new_features)
    out.features = new_features
    return out
