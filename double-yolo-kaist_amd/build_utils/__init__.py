"""Hot-path half of the reference's `build_utils` package (parse_config, layers, utils, torch_utils).

The data pipeline and drawing helpers of the reference (kaist_dataset, img_utils, snowflake,
draw_box_utils) are out of scope; when this package is used as a drop-in next to a checkout of the
reference, set DYK_REFERENCE_ROOT=<checkout> and those sub-modules resolve from there.
"""
import os as _os

_ref = _os.environ.get("DYK_REFERENCE_ROOT")
if _ref and _os.path.isdir(_os.path.join(_ref, "build_utils")):
    __path__.append(_os.path.join(_ref, "build_utils"))
