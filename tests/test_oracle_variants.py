"""The two steps whose form differs between exporter generations, as oracle twins: the coordinate rule of
ResizeBilinear (legacy vs half_pixel_centers) and the order of clipping and NMS.  Known answers worked out by hand."""
import numpy as np

from oracle import postprocess as post
from oracle import preprocess as pre

F32 = np.float32


def test_half_pixel_interpolation_weights_known_answers():
    # 4 -> 2: scale 2.  legacy: src = 0, 2 (exactly on pixels 0 and 2); half-pixel: src = 0.5, 2.5 (between pixels)
    lo, up, w = pre.interpolation_weights(2, 4)
    assert lo.tolist() == [0, 2] and up.tolist() == [0, 2] and w.tolist() == [0.0, 0.0]
    lo, up, w = pre.interpolation_weights(2, 4, half_pixel_centers=True)
    assert lo.tolist() == [0, 2] and up.tolist() == [1, 3] and w.tolist() == [0.5, 0.5]
    # 2 -> 4 (upscaling): half-pixel src = -0.25, 0.25, 0.75, 1.25: clamped below, lerp measured from floor(src)
    lo, up, w = pre.interpolation_weights(4, 2, half_pixel_centers=True)
    assert lo.tolist() == [0, 0, 0, 1] and up.tolist() == [0, 1, 1, 1]
    np.testing.assert_array_equal(w, np.array([0.75, 0.25, 0.75, 0.25], F32))
    img = np.array([[[0, 0, 0], [100, 100, 100]]], np.uint8).repeat(2, 0)                 # 2 x 2, columns 0 and 100
    out = pre.resize_bilinear(img, 4, 4, half_pixel_centers=True)[0, :, 0]
    np.testing.assert_array_equal(out, np.array([0, 25, 75, 100], F32))
    out = pre.resize_bilinear(img, 4, 4)[0, :, 0]                                       # legacy: src = 0, .5, 1, 1.5
    np.testing.assert_array_equal(out, np.array([0, 50, 100, 100], F32))


def _frame(boxes, scores):
    """boxes [n,4] decoded, scores [n] of class 0 -> the arguments of the NMS functions (2 classes)."""
    s = np.zeros((len(boxes), 2), F32)
    s[:, 0] = scores
    return np.asarray(boxes, F32), s


def test_clip_order_known_answers():
    # A sticks far out of the image, B lies inside: IoU(A, B) = 0.25 unclipped -- but A clipped IS B's twin (IoU 1)
    A, B = [-1.0, -1.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]
    boxes, scores = _frame([A, B], [0.9, 0.8])
    b1, s1, c1, n1 = post.multiclass_nms(boxes, scores)
    assert n1 == 1 and s1[0] == F32(0.9) and b1[0].tolist() == [0, 0, 1, 1]                # clipped first: B is suppressed
    b2, s2, c2, n2 = post.multiclass_nms_clip_after(boxes, scores)
    assert n2 == 2 and s2[:2].tolist() == [F32(0.9), F32(0.8)] and b2[0].tolist() == [0, 0, 1, 1] and b2[1].tolist() == [0, 0, 1, 1]
    # P lies entirely outside: clip-first drops it before the NMS; clip-after lets it suppress its neighbour Q, then prunes it
    P, Q = [1.2, 0.0, 1.6, 0.4], [1.15, 0.0, 1.6, 0.4]                                     # IoU(P, Q) = 0.889; Q, too, is outside
    R = [0.9, 0.0, 1.6, 0.4]                                                               # overlaps P by 0.571 < 0.6: survives, clips to [0.9, 1]
    boxes, scores = _frame([P, Q, R], [0.9, 0.8, 0.7])
    b1, s1, c1, n1 = post.multiclass_nms(boxes, scores)
    assert n1 == 1 and s1[0] == F32(0.7)                                                   # P and Q have no area after clipping
    b2, s2, c2, n2 = post.multiclass_nms_clip_after(boxes, scores)
    assert n2 == 1 and s2[0] == F32(0.7) and np.allclose(b2[0], [0.9, 0.0, 1.0, 0.4])
    # ... and a box that clip-first keeps but clip-after loses to an outsider
    O, I = [-0.5, 0.0, 0.3, 0.5], [-0.45, 0.0, 0.3, 0.5]                                    # IoU 0.9375 unclipped; both reach into the image
    boxes, scores = _frame([O, I], [0.9, 0.8])
    assert post.multiclass_nms(boxes, scores)[3] == 1 and post.multiclass_nms_clip_after(boxes, scores)[3] == 1
    # per-class cap counts what the NMS selected, pruned or not
    far = [[2.0 + i, 0.0, 2.5 + i, 0.5] for i in range(3)]                                 # three selected boxes outside the image
    boxes, scores = _frame(far + [[0.1, 0.1, 0.4, 0.4], [0.5, 0.5, 0.9, 0.9]], [0.9, 0.8, 0.7, 0.6, 0.5])
    b, s, c, n = post.multiclass_nms_clip_after(boxes, scores, max_per_class=4)
    assert n == 1 and s[0] == F32(0.6)                                                     # the fifth box found its class full
    b, s, c, n = post.multiclass_nms(boxes, scores, max_per_class=4)
    assert n == 2                                                                          # clip-first never saw the outsiders
