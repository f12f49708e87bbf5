"""Compile single definitions of the reference's source files, in the build container, into a namespace that holds torch symbols only.

Used by tests/golden/make_ref_leaf_golden.py (leaf functions) and tests/golden/make_ref_wiring_golden.py (block bodies and the wiring
methods). The reference cannot be imported here (deepspeed / diffusers / peft / ipdb are absent: ordinary ModuleNotFoundError), but many of its
definitions are self-contained PyTorch once their `self` / `module` / `attn` arguments are attribute bags of plain torch modules and callables.

What this module guarantees about a compiled definition:
  * the source file is the one this repo was written against: its sha256 is pinned below and a mismatch refuses to run (the reference is
    untrusted public content; a changed file would otherwise execute unreviewed code in the build container);
  * the namespace holds `torch`, `torch.nn`, `torch.nn.functional`, `typing.List`, whatever reference definition the caller compiled the same
    way and passes in `extra`, and tripwires (`not_taken`) - nothing else;
  * `__builtins__` of that namespace is the short whitelist BUILTINS (len, range, dict, zip, ... - no __import__, open, eval, exec, getattr);
  * every free name of the definition is one of the above, or the call fails before anything runs.
Annotations and decorators are dropped (they name typing symbols the namespace does not hold); the body is untouched. Nothing of the reference's
text is stored anywhere: the fixtures hold tensors only.
"""
from __future__ import annotations

import ast
import builtins
import hashlib
import os
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

REF = os.environ.get("UNIGEN_REFERENCE", "/root/reference")
ALLOWED = {"torch": torch, "nn": nn, "F": F, "List": List}
SHA256 = {
    "src/UniGenUtils.py": "b459854c104b4c01543accb6add1d4f8e398b060bd0d58e812728682ab67b224",
    "src/UniGenTransformer.py": "7c626da9c4201e36c06f020eb6d9a16cf89b0eff6fea154140306c277fc1bd4c",
    "src/UniGenPipeline.py": "ce2a1bbe35d5e927af22f8a361d9dbd2de94fbf178f08d75ccd0cfb27a1a02ab",
}
BUILTINS = {n: getattr(builtins, n) for n in (
    "len", "range", "enumerate", "zip", "dict", "list", "tuple", "int", "float", "bool", "sum", "min", "max", "abs", "isinstance", "hasattr",
    "ValueError", "AssertionError", "TypeError", "KeyError")}


def available() -> bool:
    return os.path.isdir(REF)


def _source(path: str) -> str:
    with open(os.path.join(REF, path), "rb") as f:
        raw = f.read()
    got = hashlib.sha256(raw).hexdigest()
    if path not in SHA256 or got != SHA256[path]:
        raise SystemExit(f"{path}: sha256 {got} is not the pinned {SHA256.get(path)}: review the file, then update tests/golden/ref_harness.py")
    return raw.decode()


def _find(tree: ast.AST, name: str, cls: str | None = None) -> ast.FunctionDef:
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    return next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)


def _tripwire(name: str):
    class _Trip:
        def __call__(self, *a, **k):
            raise AssertionError(f"the fixture reached `{name}`, a third-party symbol the namespace does not provide")

        def __getattr__(self, attr):
            raise AssertionError(f"the fixture reached `{name}.{attr}`, a third-party symbol the namespace does not provide")
    return _Trip()


def compile_reference_function(path: str, name: str, cls: str | None = None, extra: dict | None = None, not_taken: tuple = ()):
    """Compile one function definition of a reference file in a namespace of torch symbols only -> (function, first line). `not_taken`:
    third-party names that only occur on a branch the fixture's arguments never take - bound to a tripwire that raises if the branch is
    entered after all (not to an implementation)."""
    tree = ast.parse(_source(path))
    fn = _find(tree, name, cls)
    for a in fn.args.args + fn.args.kwonlyargs + [x for x in (fn.args.vararg, fn.args.kwarg) if x is not None]:
        a.annotation = None
    fn.returns = None
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(ALLOWED)
    ns.update(extra or {})
    ns.update({n: _tripwire(n) for n in not_taken})
    ns["__builtins__"] = dict(BUILTINS)
    loads = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
    local = {a.arg for a in fn.args.args + fn.args.kwonlyargs} | ({fn.args.kwarg.arg} if fn.args.kwarg else set()) | \
            ({fn.args.vararg.arg} if fn.args.vararg else set()) | \
            {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
    unknown = {n for n in loads - local if n not in ns and n not in BUILTINS}
    assert not unknown, f"{name}: reaches names outside torch and the builtin whitelist: {unknown}"
    # no import statement, no dunder attribute access, no nested definition that could smuggle either
    for n in ast.walk(fn):
        assert not isinstance(n, (ast.Import, ast.ImportFrom, ast.Global, ast.Nonlocal)), f"{name}: {type(n).__name__} in a reference body"
        assert not (isinstance(n, ast.Attribute) and n.attr.startswith("__")), f"{name}: dunder attribute `{n.attr}` in a reference body"
    exec(compile(mod, f"<{path}:{fn.lineno} {name}>", "exec"), ns)
    return ns[name], fn.lineno


def bag(**kw) -> nn.Module:
    """An attribute bag: what the reference's code sees as `self` / `module` / `attn`. Plain torch modules, tensors and callables only."""
    m = nn.Module()
    for k, v in kw.items():
        if callable(v) and not isinstance(v, nn.Module):
            object.__setattr__(m, k, v)
        else:
            setattr(m, k, v)
    return m


def lin(i: int, o: int, g: torch.Generator, std: float = 0.3) -> nn.Linear:
    """nn.Linear with bf16-representable seeded parameters."""
    m = nn.Linear(i, o)
    with torch.no_grad():
        m.weight.copy_((torch.randn(o, i, generator=g) * std / i ** 0.5).bfloat16().float())
        m.bias.copy_((torch.randn(o, generator=g) * 0.1).bfloat16().float())
    return m
