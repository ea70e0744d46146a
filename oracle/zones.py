"""Oracle (test infrastructure): detection zones as *polygons*, the way the reference defines them.

Restates `get_alpha_channel`, `find_contours`, `contours_key` of `watsor/filter/mask.py:62-88` and the
shapely `Polygon.intersects` call of `mask.py:45-54` without OpenCV / shapely (neither is installed
here; both are un-vendored pip dependencies of the reference, `setup.py:38-39`):

  * thresh = 255 where alpha == 255 (mask.py:85)
  * findContours(RETR_EXTERNAL): outermost borders only -> holes and anything nested inside them belong
    to the enclosing contour; CHAIN_APPROX_SIMPLE only drops collinear points, the polygon is unchanged
  * contours_key: (cx, cy) = int(m10/m00), int(m01/m00) with cv2.moments of the *contour* (Green's
    formula over the vertex list), key = cx^2 + cy^2; sorted() is stable
  * Polygon(contour).intersects(Polygon(box)): closed sets sharing at least one point -- done here in
    exact integer arithmetic.

Pinned by `watsor/test/test_filter.py:38-74` (mask known-answer test) and by the zone statistics of the
reference's `config/porch.png` recorded in SURVEY.md §4.  Degenerate (zero-area) contours raise
ZeroDivisionError like the reference (mask.py:80).  Tie order between equal keys is OpenCV's contour
order, which cannot be observed here: UNPINNED detail, equal keys are avoided in tests.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

Point = Tuple[int, int]

# 8 neighbours, clockwise on screen (y down), starting east
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, 1, 1, 1, 0, -1, -1, -1)


def read_alpha(filename, width=None, height=None) -> np.ndarray:
    """get_alpha_channel (mask.py:62-75) with PIL; same assertion messages."""
    from PIL import Image
    try:
        img = Image.open(filename)
        img.load()
    except Exception:
        img = None
    assert img is not None, "Error reading mask file {}".format(filename)
    arr = np.array(img)
    assert arr.ndim == 3 and arr.shape[2] == 4, "Mask image {} is not of 32 bit color".format(filename)
    if width is not None and height is not None:
        assert arr.shape[0] == height and arr.shape[1] == width, \
            "The size of mask image {} doesn't match {}x{}".format(filename, width, height)
    return np.ascontiguousarray(arr[:, :, 3])


def filled_components(alpha: np.ndarray):
    """(labels int32 (H,W), n): outermost 8-connected alpha==255 regions with their interiors filled."""
    from scipy import ndimage
    fg = alpha == 255
    filled = ndimage.binary_fill_holes(fg)                      # background invaded 4-connectedly from outside
    lab, n = ndimage.label(filled, structure=np.ones((3, 3), bool))
    return lab.astype(np.int32), int(n)


def outer_border(mask: np.ndarray) -> List[Point]:
    """Closed 8-connected outer border of the single region in `mask` as a vertex list (x, y)."""
    h, w = mask.shape
    ys, xs = np.nonzero(mask)
    first = np.lexsort((xs, ys))[0]
    sx, sy = int(xs[first]), int(ys[first])          # top-most, then left-most pixel

    def inside(x, y):
        return 0 <= x < w and 0 <= y < h and mask[y, x]

    pts = [(sx, sy)]
    if not any(inside(sx + _DX[d], sy + _DY[d]) for d in range(8)):
        return pts
    cx, cy, back = sx, sy, 4                          # the west neighbour of the start is background
    for _ in range(8 * mask.size + 16):
        nxt = None
        for k in range(1, 9):
            d = (back + k) % 8
            if inside(cx + _DX[d], cy + _DY[d]):
                nxt = d
                break
        nx, ny = cx + _DX[nxt], cy + _DY[nxt]
        if (cx, cy) == (sx, sy) and len(pts) > 1 and (nx, ny) == pts[1]:
            pts.pop()                                 # closing visit of the start duplicates pts[0]
            return pts
        prev = (nxt + 7) % 8
        bx, by = cx + _DX[prev], cy + _DY[prev]       # last background pixel looked at
        back = next(d for d in range(8) if (nx + _DX[d], ny + _DY[d]) == (bx, by))
        cx, cy = nx, ny
        pts.append((cx, cy))
    raise RuntimeError("border tracing did not close")


def contour_moments(pts: List[Point]):
    """cv::moments for a contour (contourMoments): m00, m10, m01 by Green's formula, in double."""
    a00 = a10 = a01 = 0.0
    xi_1, yi_1 = float(pts[-1][0]), float(pts[-1][1])
    for x, y in pts:
        xi, yi = float(x), float(y)
        dxy = xi_1 * yi - xi * yi_1
        a00 += dxy
        a10 += dxy * (xi_1 + xi)
        a01 += dxy * (yi_1 + yi)
        xi_1, yi_1 = xi, yi
    m00, m10, m01 = a00 * 0.5, a10 / 6.0, a01 / 6.0
    if a00 < 0:
        m00, m10, m01 = -m00, -m10, -m01
    return m00, m10, m01


def contours_key(pts: List[Point]) -> int:
    """mask.py:78-81."""
    m00, m10, m01 = contour_moments(pts)
    center = (int(m10 / m00), int(m01 / m00))         # ZeroDivisionError for a degenerate contour
    return center[0] * center[0] + center[1] * center[1]


def zone_polygons(alpha: np.ndarray) -> List[List[Point]]:
    """find_contours (mask.py:84-88): contour polygons sorted by contours_key (stable)."""
    lab, n = filled_components(alpha)
    contours = [outer_border(lab == i) for i in range(1, n + 1)]
    contours.reverse()                                 # OpenCV returns the last-found contour first
    return sorted(contours, key=contours_key)


def zone_fill(alpha: np.ndarray) -> np.ndarray:
    """uint8 [nz,H,W]: lattice points of each zone polygon, zones in the reference's order.

    Built from the polygons (even-odd scanline fill + the border itself), i.e. independently of the
    component labelling the product uses."""
    polys = zone_polygons(alpha)
    h, w = alpha.shape
    out = np.zeros((len(polys), h, w), np.uint8)
    for zi, poly in enumerate(polys):
        xs = [p[0] for p in poly]
        ys = [p[1] for p in poly]
        for y in range(min(ys), max(ys) + 1):
            for x in range(min(xs), max(xs) + 1):
                if point_in_polygon(x, y, poly):
                    out[zi, y, x] = 1
    return out


# ---- exact integer geometry (stands in for shapely/GEOS) -----------------------------------------

def _orient(ax, ay, bx, by, cx, cy) -> int:
    v = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    return (v > 0) - (v < 0)


def _on_segment(ax, ay, bx, by, px, py) -> bool:
    return (_orient(ax, ay, bx, by, px, py) == 0 and min(ax, bx) <= px <= max(ax, bx)
            and min(ay, by) <= py <= max(ay, by))


def segments_intersect(a, b, c, d) -> bool:
    """Closed segments ab and cd share a point."""
    o1 = _orient(*a, *b, *c)
    o2 = _orient(*a, *b, *d)
    o3 = _orient(*c, *d, *a)
    o4 = _orient(*c, *d, *b)
    if o1 != o2 and o3 != o4:
        return True
    return (_on_segment(*a, *b, *c) or _on_segment(*a, *b, *d) or _on_segment(*c, *d, *a)
            or _on_segment(*c, *d, *b))


def point_in_polygon(px: int, py: int, poly: List[Point]) -> bool:
    """Closed polygon (boundary included), non-zero winding, exact for integer coordinates."""
    n = len(poly)
    if n == 1:
        return (px, py) == poly[0]
    wn = 0
    for i in range(n):
        ax, ay = poly[i]
        bx, by = poly[(i + 1) % n]
        if _on_segment(ax, ay, bx, by, px, py):
            return True
        if ay <= py:
            if by > py and _orient(ax, ay, bx, by, px, py) > 0:
                wn += 1
        elif by <= py and _orient(ax, ay, bx, by, px, py) < 0:
            wn -= 1
    return wn != 0


def box_intersects_polygon(x_min: int, y_min: int, x_max: int, y_max: int, poly: List[Point]) -> bool:
    """Polygon([(x_min,y_min),(x_max,y_min),(x_max,y_max),(x_min,y_max)]).intersects(Polygon(poly))."""
    xl, xh = min(x_min, x_max), max(x_min, x_max)
    yl, yh = min(y_min, y_max), max(y_min, y_max)
    for (x, y) in poly:                                    # a zone vertex inside the closed box
        if xl <= x <= xh and yl <= y <= yh:
            return True
    corners = [(x_min, y_min), (x_max, y_min), (x_max, y_max), (x_min, y_max)]
    for (x, y) in corners:                                 # a box corner inside the closed zone polygon
        if point_in_polygon(x, y, poly):
            return True
    n = len(poly)
    for i in range(n):                                     # boundaries cross
        a, b = poly[i], poly[(i + 1) % n]
        if max(a[0], b[0]) < xl or min(a[0], b[0]) > xh or max(a[1], b[1]) < yl or min(a[1], b[1]) > yh:
            continue
        for j in range(4):
            if segments_intersect(a, b, corners[j], corners[(j + 1) % 4]):
                return True
    return False
