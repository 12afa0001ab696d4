"""Row sharding of one frame over the GPUs of a node.

Every pixel-sample depends only on its own pixel (raymarcher.frag:226,350-351,382),
so the frame shards with no exchange between samples.  Contiguous row blocks
would be unbalanced (sky rows are ~10x cheaper than rows through the fractal),
so the frame is cut into stripes of `stripe_rows` rows dealt round-robin:
part g holds the stripes k with k % parts == g, packed in ascending order
(rm_fb_create_striped).  Once per presented frame the colour plane is gathered
to rank 0 and the stripes are put back in image order.
"""
from __future__ import annotations

import numpy as np

STRIPE_ROWS = 8  # = the height of a wave's 8x8 pixel tile


def owned_rows(height: int, parts: int, part: int, stripe_rows: int = STRIPE_ROWS) -> np.ndarray:
    """Image rows held by `part`, in the order they are packed."""
    rows = np.arange(height)
    return rows[(rows // stripe_rows) % parts == part]


def row_counts(height: int, parts: int, stripe_rows: int = STRIPE_ROWS):
    return [len(owned_rows(height, parts, p, stripe_rows)) for p in range(parts)]


def assemble(parts_data, height: int, stripe_rows: int = STRIPE_ROWS):
    """Put gathered per-part planes (each [>= rows_p, W, C], numpy or torch) back
    into image order -> [height, W, C]."""
    n = len(parts_data)
    first = parts_data[0]
    if isinstance(first, np.ndarray):
        out = np.empty((height,) + first.shape[1:], first.dtype)
        for p, data in enumerate(parts_data):
            rows = owned_rows(height, n, p, stripe_rows)
            out[rows] = data[: len(rows)]
        return out
    import torch

    out = torch.empty((height,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    for p, data in enumerate(parts_data):
        rows = torch.as_tensor(owned_rows(height, n, p, stripe_rows), device=first.device)
        out.index_copy_(0, rows, data[: len(rows)])
    return out
