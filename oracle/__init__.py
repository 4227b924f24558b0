"""CPU oracle for the YOLOv2 hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a CPU restatement (numpy for the bit-exact integer/index work,
torch-CPU for the floating-point network math whose arithmetic lives in the
third-party dependency PyTorch — the reference pins `torch<=0.3.1`,
requirements.txt:5; here torch 2.10 CPU kernels) of the algorithms of
ruiminshen/yolo2-pytorch on the path named by BASELINE.json:north_star.  Every
function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import anything from here, and only as the checker / reported baseline.
The product package (`yolo2-pytorch_amd/`) never imports `oracle`; its GPU path
raises if the HIP library is missing.

Pinning (how we know the restatement is the reference's algorithm):
  * IoU: the reference's own known-answer tests (utils/iou/torch.py:79-113,
    179-213, 255-289; utils/iou/numpy.py:108-142) are restated in
    tests/test_oracle.py and must pass on the oracle.
  * Darknet forward / decode / NMS / filter+postprocess / loss: no reference
    test or stored vector pins these, so they are pinned against OUTPUTS OF
    THE REFERENCE ITSELF run in the build container: oracle/make_golden.py
    loads the reference files by path (oracle/refload.py), runs them on seeded
    inputs and writes tests/golden/*.npz; tests/test_oracle.py checks the
    oracle against those fixtures (bit-exact for indices, fp32 tolerance for
    tensors).  The loss needs six line-level edits to run on torch 2.10
    (0.3 mask semantics); make_golden.py applies them to
    a patched in-memory copy of the reference source, never to a file.
"""
