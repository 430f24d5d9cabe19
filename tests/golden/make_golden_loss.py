"""Second half of the golden generator: targets / loss / nms / ap / step fixtures (see make_golden.py)."""


def run(what, ref_models, ref_utils, ref_parse, ref_metrics, HERE):
    pass
