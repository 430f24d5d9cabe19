"""Import the reference (read-only, /root/reference) in THIS container to generate golden vectors.

Only used by tests/golden/make_golden.py; never on the GPU box (the reference does not travel).
Two absent third-party modules are stubbed so that `models.py` / `build_utils/utils.py` import:
cv2 (only `setNumThreads` is touched at import, utils.py:21) and torchvision (only
`torchvision.ops.nms`, utils.py:448, for which oracle.nms.nms_numpy supplies the documented
torchvision semantics).
"""
import os
import sys
import types

REF = "/root/reference"


def import_reference():
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.nms import nms_numpy

    cv2 = types.ModuleType("cv2")
    cv2.cv2 = cv2
    cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = sys.modules["cv2.cv2"] = cv2
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")

    def _nms(boxes, scores, thr):
        keep = nms_numpy(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(thr))
        return torch.as_tensor(keep, dtype=torch.long)

    ops.nms = _nms
    tv.ops = ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = ops
    # the reference's modules are top-level (`models`, `build_utils`): make sure ours are not shadowing
    for k in list(sys.modules):
        if k == "models" or k.startswith("build_utils") or k.startswith("other_utils"):
            del sys.modules[k]
    sys.path = [p for p in sys.path if not p.rstrip("/").endswith("double-yolo-kaist_amd")]
    sys.path.insert(0, REF)
    os.chdir(REF)  # cfg paths in the reference are relative
    import models as ref_models
    import build_utils.utils as ref_utils
    import build_utils.parse_config as ref_parse
    import other_utils.metrics as ref_metrics
    return ref_models, ref_utils, ref_parse, ref_metrics
