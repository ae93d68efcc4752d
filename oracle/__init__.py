"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32 / numpy) restatement of the multiyolov5 hot path
(Model.forward_once, Detect, SegMask*, ComputeLoss, SegmentationLosses,
OhemCELoss, non_max_suppression, seg argmax), used solely as the parity
checker for the HIP path.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package.  Nothing under `multiyolov5_amd/`
imports it; the product path raises if the HIP library is missing.

Pinning: the reference (`/root/reference`, pure Python/PyTorch) ships no
tests, golden vectors or weights.  The restatement is
therefore pinned against outputs of the reference itself, generated in the
build container by `oracle/make_golden.py` (which imports the reference
through `oracle/ref_shim.py`) and committed under `tests/golden/`.
`torchvision.ops.nms` (requirements.txt:11, `torchvision>=0.8.1`, call site
utils/general.py:493) is absent from the reference tree and from this image:
its greedy algorithm is restated in `oracle/nms_ref.py` -- that one function
is "parity unpinned" (no reference test pins it); everything around it is
pinned by running the reference's own `non_max_suppression` with the
restated kernel injected.
"""
