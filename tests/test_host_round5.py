"""Round-5 host-side fixes, CPU only: the cached module walk of ops.fast_parameters follows a changing module tree,
the YAML loader accepts anchors + merge keys as OmegaConf's does, the trainer's mode helper re-applies a mode a child lost,
tune_for_throughput(False) restores what was in effect before."""
import torch
import yaml

from dxmi_hip import ops


def test_fast_parameters_follows_the_module_tree():
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2))
    assert [id(p) for p in ops.fast_parameters(net)] == [id(p) for p in net.parameters()]
    net.register_parameter("log_betas", torch.nn.Parameter(torch.zeros(5)))          # late registration on an existing module: read live
    assert [id(p) for p in ops.fast_parameters(net)] == [id(p) for p in net.parameters()]
    net[2] = torch.nn.Linear(4, 7)                                                    # a REPLACED submodule: the cached list is stale
    net.add_module("extra", torch.nn.Linear(7, 1))
    got = None
    for _ in range(70):             # the fingerprint is re-checked every 64th call
        got = [id(p) for p in ops.fast_parameters(net)]
    assert got == [id(p) for p in net.parameters()]
    # a parameter shared between two modules is yielded once, as module.parameters() does
    a, b = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    b.weight = a.weight
    both = torch.nn.ModuleList([a, b])
    assert [id(p) for p in ops.fast_parameters(both)] == [id(p) for p in both.parameters()]


def test_yaml_anchors_and_merge_keys_load():
    import dxmi_config
    text = """
base: &base
  lr: 1e-7
  T: 10
sampler:
  <<: *base
  T: 4
other: {<<: *base}
"""
    cfg = yaml.load(text, Loader=dxmi_config._Loader)
    assert cfg["sampler"] == {"lr": 1e-7, "T": 4} and cfg["other"] == {"lr": 1e-7, "T": 10}
    try:
        yaml.load("a: 1\na: 2\n", Loader=dxmi_config._Loader)
        raise AssertionError("duplicate keys must stay an error")
    except yaml.constructor.ConstructorError:
        pass


def test_set_mode_reapplies_a_mode_a_child_lost():
    from models.DxMI.trainer import _set_mode
    net = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.Dropout(0.5), torch.nn.Sequential(torch.nn.Dropout(0.1)))
    _set_mode(net, True)
    assert all(m.training for m in net.modules())
    net[1].eval()                                   # somebody toggles a child alone: the root flag still says "training"
    _set_mode(net, True)
    assert all(m.training for m in net.modules())
    _set_mode(net, False)
    assert not any(m.training for m in net.modules())
    net.train()                                     # ... or the root, behind the trainer's back
    _set_mode(net, False)
    assert not any(m.training for m in net.modules())
    _set_mode(net, False)                           # nothing changed: no walk needed, still consistent
    assert not any(m.training for m in net.modules())


def test_tune_for_throughput_restores_previous_values(monkeypatch):
    state = {"conv_ws_min_tiles": 7, "conv_sm_mask": 5}          # e.g. environment overrides in effect

    def fake_set(name, value):
        old = state[name]
        state[name] = int(value)
        return old
    monkeypatch.setattr(ops, "set_tuning", fake_set)
    ops.tune_for_throughput(True)
    assert state == {"conv_ws_min_tiles": 96, "conv_sm_mask": 13}
    with ops.throughput_tuning():
        assert state == {"conv_ws_min_tiles": 96, "conv_sm_mask": 13}
    assert state == {"conv_ws_min_tiles": 96, "conv_sm_mask": 13}
    ops.tune_for_throughput(False)
    assert state == {"conv_ws_min_tiles": 7, "conv_sm_mask": 5}
    ops.tune_for_throughput(False)                                   # unbalanced: nothing to restore, nothing touched
    assert state == {"conv_ws_min_tiles": 7, "conv_sm_mask": 5}
