"""oracle/ -- CPU restatement of the reference's dual-stream hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU / numpy code of our own, the algorithm of
Ye-zixiao/Double-YOLO-Kaist's forward / loss / decode / NMS path so that the HIP product can be
checked against it.  Every function cites the reference file:line it follows.

Pinning: the reference is Python and imports in the build container, so the oracle is pinned by
golden vectors generated from the *reference itself* (tests/golden/make_golden.py, fixtures
under tests/golden/); `pytest -m "not gpu"` re-checks the oracle against them.  One boundary is
"parity unpinned": `torchvision.ops.nms` (reference build_utils/utils.py:448) is a third-party
dependency that is neither vendored nor version-pinned by the reference (requirements.txt does
not list torchvision) and is absent from this image; oracle/nms.py restates its documented
algorithm (greedy, descending score, suppress IoU > thr, IoU = inter / (a1 + a2 - inter)) and is
anchored on hand-computed known-answer tests instead.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (double-yolo-kaist_amd/) never does.
"""
