"""CPU oracle for the Watsor detection hot path  --  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy / torch-CPU fp32, plain C for the byte and
integer stages), the algorithm that the reference's CPU detector plugin runs for one
frame (`watsor/detection/tensorflow_cpu.py:74-121`) together with the per-camera
filters that run right behind it (`watsor/filter/{confidence,area,mask}.py`).

Who may use it: `tests/`, `__graft_entry__.smoke()` and two legs of `bench.py` -- `cpu_baseline`
(the reported baseline) and `parity` (the live check of the timed engine's scores against the
north star's 1e-3, requested by the round-1 review) -- as the *checker* or the *reported
baseline*, never as the thing that is shipped or measured.  Nothing under `watsor_amd/` imports this package; the product
path fails loudly when the HIP library is missing instead of falling back to it.

Pinning status (see DESIGN.md "Oracle"):

* detector numerics (resize, network, decode, NMS): **parity unpinned**.  The
  arithmetic of the reference lives in TensorFlow and in a model file that are both
  absent from `/root/reference` (un-vendored pip dependency `tensorflow`, unpinned in
  `setup.py:51-53`; model `ssd_mobilenet_v2_coco_2018_03_29`, URL only in
  `README.md:450`).  The reference's only test of the path (`watsor/test/test_detect.py:28-77`)
  asserts a detection *count*.  The restatement follows the published TF1 Object
  Detection API graph semantics and the reference's own SSD
  config (`watsor/test/model/prepare.py:19-150`); the backbone is cross-checked
  against the independent `transformers` MobileNetV2 (tests/test_oracle_witness.py).
  Route to pinned: `tests/golden/make_tf_golden.py` (run where TensorFlow and the model exist)
  + `tests/test_tf_golden.py` (consumes the vectors when present).
* row fill / int truncation / struct ABI: pinned by `tensorflow_cpu.py:79-90` and the
  ctypes layout of `watsor/stream/share.py:11-32` (checked against the real structs
  when `/root/reference` is importable, tests/test_abi.py).
* YUV 4:2:0 -> RGB24 in front of the resize (oracle/yuv.py; only for frames handed over as NV12 / I420, which the
  reference itself never does -- `watsor/config/schema.py:161`): the formula is pinned by BT.601 known answers
  (tests/test_yuv_oracle.py); **unpinned** against ffmpeg's swscale, which would do this conversion on the host.
* confidence / area / mask filters: pinned by the known-answer tests of
  `watsor/test/test_filter.py:14-74` (tests/test_filters_oracle.py) and by fixtures
  generated from the reference's own `ConfidenceFilter` / `AreaFilter`
  (tests/golden/make_filter_golden.py).
"""
