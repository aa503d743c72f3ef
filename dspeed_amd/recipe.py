"""The recipe front-end: what a dspeed JSON / YAML recipe *says*, before anything is translated for the device.

A recipe maps ``"name"`` (or ``"name_a, name_b"`` for processors with several outputs) to either a string -- ``"module.func(arg, ...)"``
or an expression of the argument language -- or a mapping with ``function`` / ``module`` / ``args`` / ``defaults`` / ``unit`` /
``prereqs``.  The behaviour follows the reference's ``build_processing_chain`` (src/dspeed/processing_chain.py:2476-2651: the
accepted forms and their error texts, ``db.a.b`` look-ups with ``defaults``, dependencies taken from the argument strings, outputs
resolved depth first with a cycle check); the structure is this package's own:

* ``Entry`` -- one recipe entry after parsing (what it defines, what it calls, its argument strings, what it needs);
* ``Recipe`` -- all entries, indexed by every name they define; ``Recipe.plan(outputs)`` walks the dependency graph with an explicit
  stack and returns the entries in an order in which everything is defined before it is used, plus the names nobody defines (the
  columns of the input table);
* the *form* of a ``function`` string is decided once, by the kind of its syntax-tree root, through the table ``_FORMS``.
"""
from __future__ import annotations

import ast
import re
from collections.abc import Callable, Mapping
from dataclasses import dataclass, field
from functools import reduce

from .errors import ProcessingChainError

#: functions of the argument language: ``round(x, 16*ns)`` as a recipe entry is an expression, not a processor call (reference :729-751)
LANGUAGE_CALLS = ("round", "floor", "ceil", "trunc", "len", "float", "int", "where", "isnan", "isfinite", "astype", "loadlh5")
#: module prefixes whose bare attributes are inline constants (``np.pi``)
CONSTANT_MODULES = ("np", "numpy")
#: names that are units, not variables
UNIT_NAMES = ("ns", "us", "ms", "s")

_DB_REF = re.compile(r"(?![^\w_.])db\.[\w_.]+")
_NAME_SPLIT = re.compile(r"[,\s]+")


def defined_names(key: str) -> tuple[str, ...]:
    """``"a, b"`` / ``"a b"`` -> ("a", "b")"""
    return tuple(part for part in _NAME_SPLIT.split(key) if part)


@dataclass
class Entry:
    key: str                      # as written in the recipe
    targets: tuple                # the variables it defines
    function: str                 # processor name, or the expression text for inline entries
    module: str | None            # None: an expression of the argument language
    args: list
    needs: list = field(default_factory=list)
    extra: dict = field(default_factory=dict)   # the remaining fields of the node (unit, kwargs, defaults, ...)

    # the builder reads entries the way it read the reference-style dicts
    def __getitem__(self, name):
        if name in ("function", "module", "args"):
            return getattr(self, name)
        if name == "prereqs":
            return self.needs
        return self.extra[name]

    def __contains__(self, name):
        return name in ("function", "module", "args", "prereqs") or name in self.extra

    def get(self, name, default=None):
        return self[name] if name in self else default

    def as_dict(self) -> dict:
        return {"function": self.function, "module": self.module, "args": list(self.args), "prereqs": list(self.needs), **self.extra}


# ----------------------------------------------------------------------------------------------------------------------------------
# the forms a "function" string can take.  Each handler receives (text, tree, given) where `given` says which of module / args the node
# already carries, and returns (function, module | _KEEP, args | _KEEP).
# ----------------------------------------------------------------------------------------------------------------------------------
_KEEP = object()


def _text_of(text: str, node: ast.AST) -> str:
    return text[node.col_offset:node.end_col_offset]


def _call_arguments(text: str, call: ast.Call) -> list[str]:
    return [_text_of(text, a) for a in (*call.args, *call.keywords)]


def _form_plain_name(key, text, tree, given):
    return text, _KEEP, _KEEP  # "trap_filter" with module and args beside it


def _form_dotted(key, text, tree, given):
    owner = _text_of(text, tree.value)
    if owner in CONSTANT_MODULES and "args" not in given:
        return text, None, [text]  # np.pi: a constant of the language
    if "module" in given:
        raise ProcessingChainError(f"Module specified twice for parameter {key}")
    return tree.attr, owner, _KEEP


def _form_call(key, text, tree, given):
    if "args" in given:
        raise ProcessingChainError(f"Cannot specify arguments if function is expr for parameter {key}")
    callee = tree.func
    if isinstance(callee, ast.Attribute):
        if "module" in given:
            raise ProcessingChainError(f"Module specified twice for parameter {key}")
        return callee.attr, _text_of(text, callee.value), _call_arguments(text, tree)
    if isinstance(callee, ast.Name):
        if callee.id in LANGUAGE_CALLS and "module" not in given:
            return text, None, [text]  # round(tp_0, 16*ns): an expression
        return callee.id, _KEEP, _call_arguments(text, tree)
    return _form_expression(key, text, tree, given)


def _form_expression(key, text, tree, given):
    if "args" in given:
        raise ProcessingChainError(f"Cannot specify arguments if function is expr for parameter {key}")
    if "module" in given:
        raise ProcessingChainError(f"Module specified twice for parameter {key}")
    return text, None, [text]


_FORMS: dict[type, Callable] = {ast.Name: _form_plain_name, ast.Attribute: _form_dotted, ast.Call: _form_call}


# ----------------------------------------------------------------------------------------------------------------------------------
def variables_in(arg: str) -> list[str]:
    """The variable names an argument string mentions, declarations ``name(shape, 'f')`` first, in order of appearance, once each."""
    try:
        tree = ast.parse(arg, mode="eval")
    except SyntaxError:
        return []
    declared = [n.func.id for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id not in LANGUAGE_CALLS]
    skip = set(declared) | set(UNIT_NAMES) | set(LANGUAGE_CALLS) | set(CONSTANT_MODULES)
    plain = [n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and n.id not in skip]
    return list(dict.fromkeys(declared + plain))


class Recipe:
    """All entries of one recipe with database values filled in."""

    def __init__(self, processors: Mapping, database: Mapping | None = None):
        self.database = database or {}
        self.entries: list[Entry] = []
        self.defined_by: dict[str, Entry] = {}
        for key, node in processors.items():
            entry = self._parse(key, node)
            self.entries.append(entry)
            self.defined_by[key] = entry
            for name in entry.targets:
                self.defined_by[name] = entry

    # ---- one entry
    def _parse(self, key: str, node) -> Entry:
        fields = {"function": node} if isinstance(node, str) else dict(node)
        if "function" not in fields:
            raise ProcessingChainError(f"no function for parameter {key}")
        text = fields.pop("function")
        tree = ast.parse(text, mode="eval").body
        function, module, args = _FORMS.get(type(tree), _form_expression)(key, text, tree, fields)
        if module is _KEEP:
            if "module" not in fields:
                raise ProcessingChainError(f"Could not find module for parameter {key}")
            module = fields["module"]
        if args is _KEEP:
            if "args" not in fields:
                raise ProcessingChainError(f"Could not find args for parameter {key}")
            args = list(fields["args"])
        fields.pop("module", None)
        fields.pop("args", None)
        needs = fields.pop("prereqs", None)
        targets = defined_names(key)
        args = [self._with_database(a, fields.get("defaults")) for a in args]
        if needs is None:
            mentioned = (name for a in args if isinstance(a, str) for name in variables_in(a))
            needs = [n for n in dict.fromkeys(mentioned) if n not in targets]
        return Entry(key, targets, function, module, args, list(needs), fields)

    # ---- database values
    def lookup(self, ref: str, defaults=None):
        """``db.a.b`` -> database["a"]["b"], else defaults["db.a.b"]"""
        try:
            return reduce(lambda level, part: level[part], ref.split(".")[1:], self.database)
        except (KeyError, TypeError, IndexError):
            pass
        try:
            return defaults[ref]
        except (KeyError, TypeError):
            raise ProcessingChainError(f"did not find {ref} in database, and could not find default value.") from None

    def _with_database(self, arg, defaults):
        if not isinstance(arg, str):
            return arg
        if _DB_REF.fullmatch(arg):
            return self.lookup(arg, defaults)  # the value keeps its type (a number, a list, a string)
        return _DB_REF.sub(lambda m: str(self.lookup(m.group(0), defaults)), arg)

    # ---- order of evaluation
    def plan(self, outputs) -> tuple[list[Entry], list[str], list[str], list[str]]:
        """-> (entries in an order that defines before it uses, names nobody defines = input columns, outputs the recipe computes, outputs
        that are copied from the input).  Depth first from each requested output, dependencies in the order the entry lists them."""
        ordered: list[Entry] = []
        finished: set[int] = set()
        inputs: list[str] = []
        computed, copied = [], []
        for out in outputs:
            root = self.defined_by.get(out)
            if root is None:
                copied.append(out)
                continue
            computed.append(out)
            if id(root) in finished:
                continue
            # explicit stack of (entry, iterator over what it needs); `open_ids` = entries whose dependencies are being walked
            stack = [(root, iter(root.needs))]
            open_ids = {id(root)}
            while stack:
                entry, pending = stack[-1]
                name = next(pending, None)
                if name is None:
                    stack.pop()
                    open_ids.discard(id(entry))
                    finished.add(id(entry))
                    ordered.append(entry)
                    continue
                dep = self.defined_by.get(name)
                if dep is None:
                    if name not in inputs:
                        inputs.append(name)
                elif id(dep) in open_ids:
                    raise ProcessingChainError(f"Circular references detected for parameter '{name}'")
                elif id(dep) not in finished:
                    stack.append((dep, iter(dep.needs)))
                    open_ids.add(id(dep))
        return ordered, inputs, computed, copied
