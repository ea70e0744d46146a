"""SSD anchor table baked into the engine file.

The reference never builds anchors itself -- they are constants inside the TF graph / the TensorRT
`GridAnchor` plugin the model was exported with -- but its own training config spells the generator
out (`watsor/test/model/prepare.py:108-119`: 6 layers, scales 0.2 .. 0.95, aspect ratios
1, 2, 0.5, 3, 0.3333) and SURVEY.md Appendix B.2 gives the TF-OD-API construction
(`ssd_anchor_generator` + `reduce_boxes_in_lowest_layer`).  Anchors are stored in the
centre/size form the box decoder consumes.
"""
from __future__ import annotations

import numpy as np


def ssd_box_specs(levels=6, min_scale=0.2, max_scale=0.95, aspect_ratios=(1.0, 2.0, 0.5, 3.0, 0.3333)):
    """[(scale, aspect ratio), ...] per feature map, as `create_ssd_anchors(reduce_boxes_in_lowest_layer=True)` lists them: what the
    graph's MultipleGridAnchorGenerator holds as constants (`watsor_amd/frozen_graph.py` checks an imported graph against it)."""
    scales = [min_scale + (max_scale - min_scale) * k / (levels - 1) for k in range(levels)] + [1.0]
    specs = []
    for k in range(levels):
        if k == 0:
            specs.append([(0.1, 1.0), (scales[0], 2.0), (scales[0], 0.5)])
        else:
            specs.append([(scales[k], r) for r in aspect_ratios] + [(float(np.sqrt(scales[k] * scales[k + 1])), 1.0)])
    return specs


def ssd_anchor_table(grid_sizes, anchors_per_location, min_scale=0.2, max_scale=0.95,
                     aspect_ratios=(1.0, 2.0, 0.5, 3.0, 0.3333)) -> np.ndarray:
    """float32 [A, 4] rows (y_center, x_center, height, width), ordered layer, row, column, anchor."""
    f32 = np.float32
    specs = ssd_box_specs(len(grid_sizes), min_scale, max_scale, aspect_ratios)
    table = []
    for k, grid in enumerate(grid_sizes):
        spec = specs[k]
        assert len(spec) == anchors_per_location[k]
        sc = np.array([s for s, _ in spec], f32)
        rs = np.sqrt(np.array([r for _, r in spec], f32)).astype(f32)
        hh = (sc / rs).astype(f32)                       # anchor heights
        ww = (sc * rs).astype(f32)                       # anchor widths
        step = f32(1.0 / grid)
        centre = (np.arange(grid, dtype=f32) * step + f32(0.5 * (1.0 / grid))).astype(f32)
        cy, cx, _ = np.meshgrid(centre, centre, np.zeros(len(spec), f32), indexing="ij")
        h = np.broadcast_to(hh, cy.shape)
        w = np.broadcast_to(ww, cy.shape)
        # TF materialises corner boxes first and the decoder turns them back into centre/size
        y0 = (cy - f32(0.5) * h).astype(f32); y1 = (cy + f32(0.5) * h).astype(f32)
        x0 = (cx - f32(0.5) * w).astype(f32); x1 = (cx + f32(0.5) * w).astype(f32)
        bh = (y1 - y0).astype(f32); bw = (x1 - x0).astype(f32)
        yc = (y0 + bh / f32(2.0)).astype(f32); xc = (x0 + bw / f32(2.0)).astype(f32)
        table.append(np.stack([yc, xc, bh, bw], -1).reshape(-1, 4))
    return np.ascontiguousarray(np.concatenate(table, 0), dtype=np.float32)
