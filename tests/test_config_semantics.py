"""OverridableConfig / set_config_overrides (judo_amd/config.py) behave as the reference's (judo/config.py:12-96): the behaviours its own test file pins
(/root/reference/tests/test_config.py:78-310 -- registry bookkeeping, the warning for an unknown field, the TypeError for a non-dataclass, reset against no-reset, fields
without defaults, default factories, switching keys, instance independence), restated on the build's classes."""

import warnings
from dataclasses import dataclass, field
from typing import Any

import pytest

from judo_amd.config import _OVERRIDE_REGISTRY, OverridableConfig, set_config_overrides


@dataclass
class Plain(OverridableConfig):
    gain: int = 10
    label: str = "stock"
    flag: bool = True


@dataclass
class Required(OverridableConfig):
    name: str
    count: int
    ratio: float = 0.5


@dataclass
class Factories(OverridableConfig):
    items: list[int] = field(default_factory=list)
    table: dict[str, Any] = field(default_factory=dict)
    fixed: int = 100


class NoDataclass:
    pass


@pytest.fixture(autouse=True)
def _fresh_registry():
    saved = {k: {kk: dict(vv) for kk, vv in v.items()} for k, v in _OVERRIDE_REGISTRY.items()}
    for cls in (Plain, Required, Factories):
        _OVERRIDE_REGISTRY.pop(cls, None)
    yield
    for cls in (Plain, Required, Factories):
        _OVERRIDE_REGISTRY.pop(cls, None)
    assert {k: v for k, v in _OVERRIDE_REGISTRY.items()} == saved  # the shipped per-task tables are untouched


def test_registry_bookkeeping():
    set_config_overrides("a", Plain, {"gain": 100, "label": "A"})
    assert _OVERRIDE_REGISTRY[Plain]["a"] == {"gain": 100, "label": "A"}
    set_config_overrides("b", Plain, {"label": "B"})                       # a second key on the same class
    assert set(_OVERRIDE_REGISTRY[Plain]) == {"a", "b"} and _OVERRIDE_REGISTRY[Plain]["b"] == {"label": "B"}
    set_config_overrides("a", Plain, {"gain": 150, "flag": False})        # updating a key keeps what it does not name
    assert _OVERRIDE_REGISTRY[Plain]["a"] == {"gain": 150, "label": "A", "flag": False}
    set_config_overrides("empty", Plain, {})                               # an empty set of values still registers the key
    assert _OVERRIDE_REGISTRY[Plain]["empty"] == {}


def test_unknown_field_warns_and_the_rest_registers():
    with pytest.warns(UserWarning, match="Field 'missing' not found in class 'Plain'"):
        set_config_overrides("a", Plain, {"missing": 1, "gain": 50})
    assert _OVERRIDE_REGISTRY[Plain]["a"] == {"gain": 50}


def test_non_dataclass_is_a_type_error():
    with pytest.raises(TypeError, match="Provided class NoDataclass is not a dataclass."):
        set_config_overrides("a", NoDataclass, {"x": 1})


def test_instances_register_their_class_without_clobbering_it():
    cfg = Plain()
    assert (cfg.gain, cfg.label) == (10, "stock") and Plain in _OVERRIDE_REGISTRY
    _OVERRIDE_REGISTRY[Plain] = {"preset": {"gain": 123}}
    Plain()
    assert _OVERRIDE_REGISTRY[Plain] == {"preset": {"gain": 123}}


def test_override_with_and_without_reset():
    set_config_overrides("p", Plain, {"gain": 100, "flag": False})
    cfg = Plain(); cfg.label = "edited"
    cfg.set_override("p")
    assert (cfg.gain, cfg.flag, cfg.label) == (100, False, "stock")        # unnamed field back to its default
    set_config_overrides("q", Plain, {"gain": 7})
    cfg.label, cfg.flag = "mine", False
    cfg.set_override("q", reset_to_defaults=False)
    assert (cfg.gain, cfg.flag, cfg.label) == (7, False, "mine")           # unnamed fields keep their values
    cfg.set_override("nobody")                                             # unknown key: everything to defaults
    assert (cfg.gain, cfg.flag, cfg.label) == (10, True, "stock")
    cfg.gain = 999
    cfg.set_override("nobody", reset_to_defaults=False)
    assert cfg.gain == 999


def test_fields_without_defaults_are_left_alone_with_a_warning():
    set_config_overrides("e", Required, {"ratio": 1.5})
    cfg = Required(name="x", count=77)
    with pytest.warns(UserWarning) as rec:
        cfg.set_override("e")
    assert (cfg.ratio, cfg.name, cfg.count) == (1.5, "x", 77)
    msgs = [str(w.message) for w in rec]
    assert any("Field 'name' has no default value to reset to and no override for key 'e'" in m for m in msgs)
    assert any("Field 'count' has no default value to reset to and no override for key 'e'" in m for m in msgs)
    set_config_overrides("e", Required, {"ratio": 2.5})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cfg.set_override("e", reset_to_defaults=False)                     # no reset asked for: nothing to warn about
    assert len(w) == 0 and (cfg.ratio, cfg.name, cfg.count) == (2.5, "x", 77)


def test_switching_keys_back_and_forth():
    set_config_overrides("dev", Plain, {"gain": 1, "label": "dev"})
    set_config_overrides("prod", Plain, {"gain": 1000, "label": "prod", "flag": False})
    cfg = Plain()
    for key, want in (("dev", (1, "dev", True)), ("prod", (1000, "prod", False)), ("dev", (1, "dev", True))):
        cfg.set_override(key)
        assert (cfg.gain, cfg.label, cfg.flag) == want


def test_default_factories_give_fresh_objects_and_overrides_are_shared():
    a = Factories(); a.items.append(1); a.table["k"] = "v"
    shared = [10, 20]
    set_config_overrides("f", Factories, {"items": shared, "fixed": 200})
    a.set_override("f")
    assert a.items is shared and a.table == {} and a.fixed == 200
    b = Factories(); b.items.append(99); b.set_override("f")
    assert b.items is shared                                               # the registered object itself, on every instance
    set_config_overrides("g", Factories, {"fixed": 300})
    a.set_override("g")
    assert a.items == [] and a.items is not shared and a.fixed == 300


def test_instances_are_independent():
    set_config_overrides("m", Plain, {"gain": 555})
    one, two = Plain(), Plain()
    one.set_override("m")
    assert (one.gain, two.gain) == (555, 10)
    two.set_override("m")
    assert two.gain == 555
