"""The recipe front-end (dspeed_amd/recipe.py) on the CPU: the forms a recipe entry can take, database look-ups and the order of
evaluation -- the behaviour of the reference's build_processing_chain (src/dspeed/processing_chain.py:2476-2651), restated as cases."""
import pytest

from dspeed_amd.errors import ProcessingChainError
from dspeed_amd.recipe import Recipe, defined_names, variables_in


def one(node, db=None, key="x"):
    return Recipe({key: node}, db).entries[0]


def test_forms_of_an_entry():
    e = one({"function": "trap_filter", "module": "dspeed.processors", "args": ["wf", "10*us", "x"]})
    assert (e.function, e.module, e.args) == ("trap_filter", "dspeed.processors", ["wf", "10*us", "x"]) and e.needs == ["wf"]
    e = one("dspeed.processors.trap_filter(wf, 10*us, 2*us, x)")
    assert (e.function, e.module, e.args) == ("trap_filter", "dspeed.processors", ["wf", "10*us", "2*us", "x"])
    e = one({"function": "dspeed.processors.min_max", "args": ["wf", "a", "b", "c", "d"]}, key="a, b, c, d")
    assert (e.function, e.module) == ("min_max", "dspeed.processors") and e.targets == ("a", "b", "c", "d") and e.needs == ["wf"]
    e = one({"function": "pole_zero(wf, db.tau, x)", "module": "dspeed.processors"}, {"tau": "60*us"})
    assert (e.function, e.module, e.args) == ("pole_zero", "dspeed.processors", ["wf", "60*us", "x"])
    e = one("numpy.amax(wf, 1, x, signature='(n),()->()', types=['fi->f'])")
    assert e.module == "numpy" and e.function == "amax" and e.args[-2:] == ["signature='(n),()->()'", "types=['fi->f']"]
    for text in ("np.pi", "round(tp, 16*ns)", "a + 2*b", "wf[10:20]", "-a", "a if b else c"):
        e = one(text)
        assert e.module is None and e.args == [text] and e.function == text, text
    assert one("round(tp, 16*ns)").needs == ["tp"] and one("a + 2*b").needs == ["a", "b"]


def test_form_errors_keep_the_reference_texts():
    with pytest.raises(ProcessingChainError, match="Module specified twice for parameter x"):
        one({"function": "m.f", "module": "m", "args": []})
    with pytest.raises(ProcessingChainError, match="Module specified twice for parameter x"):
        one({"function": "m.f(a)", "module": "m"})
    with pytest.raises(ProcessingChainError, match="Cannot specify arguments if function is expr for parameter x"):
        one({"function": "f(a)", "module": "m", "args": ["a"]})
    with pytest.raises(ProcessingChainError, match="Cannot specify arguments if function is expr for parameter x"):
        one({"function": "a + b", "args": ["a"]})
    with pytest.raises(ProcessingChainError, match="Could not find module for parameter x"):
        one({"function": "f", "args": ["a"]})
    with pytest.raises(ProcessingChainError, match="Could not find module for parameter x"):
        one("f(a, b)")
    with pytest.raises(ProcessingChainError, match="Could not find args for parameter x"):
        one({"function": "f", "module": "m"})
    with pytest.raises(ProcessingChainError):
        one({"module": "m", "args": []})


def test_database_values():
    db = {"pz": {"tau": "60*us", "n": 3}, "list": [1, 2]}
    e = one({"function": "f", "module": "m", "args": ["db.pz.tau", "db.pz.n", "2*db.pz.n + 1", "db.list", 7]}, db)
    assert e.args == ["60*us", 3, "2*3 + 1", [1, 2], 7]  # a whole-argument reference keeps the value's type
    e = one({"function": "f", "module": "m", "args": ["db.missing.tau", "db.pz.tau"], "defaults": {"db.missing.tau": "5*us"}}, db)
    assert e.args == ["5*us", "60*us"]
    with pytest.raises(ProcessingChainError, match="did not find db.missing.tau in database, and could not find default value."):
        one({"function": "f", "module": "m", "args": ["db.missing.tau"]}, db)
    with pytest.raises(ProcessingChainError, match="did not find db.pz.tau.deeper"):
        one({"function": "f", "module": "m", "args": ["db.pz.tau.deeper + 1"]}, db)
    assert one("m.f(wf, db.pz.tau, x)", db).needs == ["wf"]  # substituted values are not dependencies


def test_order_of_evaluation():
    procs = {
        "wf_blsub": "m.bl_subtract(waveform, baseline, wf_blsub)",
        "wf_pz": "m.pole_zero(wf_blsub, db.tau, wf_pz)",
        "tp_min, tp_max, wf_min, wf_max": "m.min_max(wf_pz, tp_min, tp_max, wf_min, wf_max)",
        "wf_trap": "m.trap_filter(wf_pz, 10*us, 3*us, wf_trap)",
        "E": "m.fixed_time_pickoff(wf_trap, tp_max + 2*us, 'l', E)",
        "unused": "m.trap_filter(wf_blsub, 1*us, 1*us, unused)",
    }
    r = Recipe(procs, {"tau": "60*us"})
    order, inputs, computed, copied = r.plan(["E", "wf_max", "timestamp"])
    keys = [e.key for e in order]
    assert keys == ["wf_blsub", "wf_pz", "wf_trap", "tp_min, tp_max, wf_min, wf_max", "E"]  # depth first, E's needs in the order written
    assert inputs == ["waveform", "baseline"] and computed == ["E", "wf_max"] and copied == ["timestamp"]
    assert r.defined_by["tp_max"] is r.defined_by["wf_min"] is r.defined_by["tp_min, tp_max, wf_min, wf_max"]
    # explicit prereqs replace the ones read from the arguments
    order, inputs, *_ = Recipe({"a": {"function": "f", "module": "m", "args": ["x", "a"], "prereqs": ["b"]}, "b": "m.g(y, b)"}).plan(["a"])
    assert [e.key for e in order] == ["b", "a"] and inputs == ["y"]
    with pytest.raises(ProcessingChainError, match="Circular references detected for parameter 'a'"):
        Recipe({"a": "m.f(b, a)", "b": "m.f(c, b)", "c": "m.f(a, c)"}).plan(["a"])
    # a diamond is not a cycle, and nothing is listed twice
    order, *_ = Recipe({"a": "m.f(x, a)", "b": "m.f(a, b)", "c": "m.f(a, c)", "d": "m.f(b, c, d)"}).plan(["d", "c"])
    assert [e.key for e in order] == ["a", "b", "c", "d"]


def test_names():
    assert defined_names("a, b  c,d") == ("a", "b", "c", "d")
    assert variables_in("wf_out(len(wf_in)-10, 'f', period=wf_in.period)") == ["wf_out", "wf_in"]
    assert variables_in("round(tp_0 + 10*us, wf.period)") == ["tp_0", "wf"]
    assert variables_in("np.pi * a") == ["a"] and variables_in("'l'") == [] and variables_in("not python (") == []
