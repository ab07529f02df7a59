"""HOCON-subset reader (recmv/hocon.py) against the config schema of the reference."""
import glob
from pathlib import Path

import pytest

from recmv.hocon import ConfigException, ConfigFactory, ConfigMissingException

REPO = Path(__file__).resolve().parent.parent


def test_schema_accessors_on_shipped_config():
    c = ConfigFactory.parse_file(str(REPO / "configs" / "synthetic" / "people_snapshot_like.conf"))
    assert c.get_int('train.sample_pix_num') == 2048
    assert c.get_float('loss_coarse.grad_weight') == 1.0                 # quoted "1." coerced
    assert c.get_float('loss_coarse.pc_weight.laplacian_weight') == -10.0
    assert c.get_bool('train.opt_camera.quat') is False and c.get_bool('train.opt_pose') is True
    assert c.get_list('train.scheduler.milestones') == [10, 30, 80, 120]
    assert c.get_string('mlp_deformer.type') == "MLPTranslator"
    assert c.get_string('loss_coarse.fl_visible_method') == "zbuff"
    sub = c.get_config('loss_coarse')
    assert sub.get_float('def_regu.c') == 0.5 and 'def_regu' in sub and 'nope' not in sub
    assert 'loss_coarse.pc_weight.def_consistent' in c and 'loss_extra' not in c
    for stage in ('coarse', 'medium', 'fine'):             # every stage train.py switches to has its loss block
        assert c.get_float(f'loss_{stage}.pc_weight.curve_aware_weight') > 0
    assert c.get_int('loss_fine.sample_pix_num') == 6144
    assert c.get_int('train.coarse.point_render.remesh_intersect') == 30
    with pytest.raises(ConfigMissingException):
        c.get_int('train.missing')
    assert c.get_int('train.missing', 7) == 7


def test_syntax_corner_cases():
    c = ConfigFactory.parse_string('''
        # comment
        a { b = 1, c : "2.5" }   // trailing comment
        a { d = [1, 2
                 3] }
        a.e = true
        s = hello world
        dup = 1
        dup = 2
        neg = -0.001
    ''')
    assert c.get_int('a.b') == 1 and c.get_float('a.c') == 2.5 and c.get_list('a.d') == [1, 2, 3]
    assert c.get_bool('a.e') is True and c.get_string('s') == "hello world" and c.get_int('dup') == 2
    assert c.get_float('neg') == -0.001
    with pytest.raises(ConfigException):
        ConfigFactory.parse_string('a { b = 1')
    with pytest.raises(ConfigException):
        ConfigFactory.parse_string('a = [1, 2')


def test_parses_every_reference_config_if_mounted():
    files = glob.glob('/root/reference/configs/**/*.conf', recursive=True)
    if not files:
        pytest.skip("reference not mounted")
    for f in files:
        c = ConfigFactory.parse_file(f)
        assert c.get_int('sdf_net.multires') == 6 and c.get_int('train.sample_pix_num') > 0
        for stage in ('coarse', 'medium', 'fine'):
            assert c.get_int(f'train.{stage}.point_render.remesh_intersect') in (30, 60, 120)
            assert c.get_float(f'loss_{stage}.color_weight') > 0
