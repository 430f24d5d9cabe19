"""Host logic: cfg parsing pinned against the reference parser's output (golden G1)."""
import json
import os

import pytest

from helpers import CFGS, GOLDEN, golden_sections, same_sections


@pytest.mark.parametrize("name", CFGS)
def test_materialized_cfg_parses_to_golden(name, tmp_path):
    from build_utils.parse_config import materialize_cfg, parse_model_cfg
    path = materialize_cfg(name, out_dir=str(tmp_path))
    got = parse_model_cfg(path)
    assert same_sections(got, golden_sections(name))
    assert got[0]["type"] == "net"


def test_quirks_match_reference():
    """text-level quirks: ';width' keys in [net], floats stay strings, spaces in anchors, int lists,
    conv default batch_normalize=0 -- expected output produced by the reference parser."""
    from build_utils.parse_config import parse_model_cfg, sections_from_json
    got = parse_model_cfg(os.path.join(GOLDEN, "quirks.cfg"))
    want = sections_from_json(os.path.join(GOLDEN, "parse_quirks.json"))
    assert same_sections(got, want)
    assert got[0][";width"] == 608 and got[0]["hue"] == ".1" and got[0]["scales"] == ".1,.1"
    assert got[3]["weights_type"] == "1.0" and got[2]["batch_normalize"] == 0
    assert got[6]["anchors"].shape == (4, 2) and got[6]["anchors"].dtype.name == "float64"


def test_errors(tmp_path):
    from build_utils.parse_config import parse_model_cfg
    with pytest.raises(FileNotFoundError):
        parse_model_cfg(str(tmp_path / "missing.cfg"))
    p = tmp_path / "x.txt"
    p.write_text("[net]\n")
    with pytest.raises(FileNotFoundError):
        parse_model_cfg(str(p))
    q = tmp_path / "bad.cfg"
    q.write_text("[net]\nfoo=1\n[convolutional]\nbogus_key=3\n")
    with pytest.raises(ValueError):
        parse_model_cfg(str(q))
    r = tmp_path / "ok.cfg"
    r.write_text("[net]\nanything_goes=1\n[convolutional]\nfilters=8\n")
    assert parse_model_cfg(str(r))[0]["anything_goes"] == 1      # [net] is exempt from the key check


def test_parse_data_cfg(tmp_path):
    from build_utils.parse_config import parse_data_cfg
    with open(os.path.join(GOLDEN, "parse_data_cfg.json")) as f:
        want = json.load(f)
    p = tmp_path / "k.data"
    p.write_text("# comment\n\n" + "\n".join("%s = %s" % kv for kv in want.items()) + "\n")
    assert parse_data_cfg(str(p)) == want
