"""bench.py is what the driver runs at round end, on a box nobody can debug on: two static checks that need no GPU.
(i) no function or method is longer than 200 lines (r04 / r05: one ~700-line main(); r06: `BenchRun`, one method per leg);
(ii) every global name a function of bench.py loads exists in the module (or is a builtin, or one of the two modules `BenchRun.__init__`
    imports into the module namespace) -- the class of mistake a mechanical split of main() makes and only a full GPU run would find."""
import ast
import builtins
import dis
import os
import types

import bench

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")


def test_no_function_is_longer_than_200_lines():
    tree = ast.parse(open(SRC).read())
    long = [(n.end_lineno - n.lineno + 1, n.name) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.end_lineno - n.lineno + 1 > 200]
    assert not long, long
    names = {n.name for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    for leg in ("leg_c2", "leg_one_engine", "assemble_headline", "leg_c3_uniform", "leg_c3_chained", "leg_c4", "leg_latency", "leg_cpu_baseline", "emit", "run"):
        assert leg in names, leg


def test_every_global_a_function_loads_exists():
    late = {"torch", "capi"}                                  # BenchRun.__init__: `global torch, capi` + imports (GPU box only)
    bad = []

    def walk(code, path):
        for ins in dis.get_instructions(code):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in bench.__dict__ and not hasattr(builtins, ins.argval) and ins.argval not in late:
                bad.append((path, ins.argval))
        for c in code.co_consts:
            if isinstance(c, types.CodeType):
                walk(c, path + "." + c.co_name)
    for name, obj in list(bench.__dict__.items()):
        if isinstance(obj, types.FunctionType) and obj.__module__ == bench.__name__:
            walk(obj.__code__, name)
        elif isinstance(obj, type) and obj.__module__ == bench.__name__:
            for k, v in obj.__dict__.items():
                if isinstance(v, types.FunctionType):
                    walk(v.__code__, name + "." + k)
    assert not bad, sorted(set(bad))
