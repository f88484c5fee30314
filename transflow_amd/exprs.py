"""Compiler from the `polar` flow filter's user expressions to the postfix programs the GPU runs.

The reference evaluates two Python expressions of (t, r, a) with r, a float32 arrays [H, W]
(transflow/flow/filters.py:75-87, utils.parse_lambda_expression utils.py:409-414).  Here every
subtree that does not touch r or a stays Python: it is evaluated on the host for each frame (so
`math`, `random`, anything of t works as in the reference) and becomes a constant of the program;
the array part must be built from arithmetic, comparisons and the numpy functions in FUNCS, which is
what runs per pixel in `k_pp_polar`.  numpy's typing is kept: float32 arithmetic unless a
numpy.float64 scalar takes part (NEP 50: Python scalars are weak).  Transcendental functions differ
from numpy's by a few units in the last place.
"""
from __future__ import annotations

import ast
import math
import os
import random
import re

import numpy

# opcodes shared with flowops.hip (enum PolarOp)
OPS = ["push_r", "push_a", "push_const", "add", "sub", "mul", "div", "pow", "mod", "floordiv", "neg", "sin", "cos",
       "tan", "arcsin", "arccos", "arctan", "arctan2", "sqrt", "abs", "exp", "log", "log2", "log10", "minimum",
       "maximum", "floor", "ceil", "rint", "sign", "square", "hypot", "lt", "le", "gt", "ge", "eq", "ne", "where",
       "clip", "reciprocal", "not"]
OP = {name: i for i, name in enumerate(OPS)}
MAX_INSTR = 48
MAX_STACK = 12

BINOPS = {ast.Add: "add", ast.Sub: "sub", ast.Mult: "mul", ast.Div: "div", ast.Pow: "pow", ast.Mod: "mod",
          ast.FloorDiv: "floordiv"}
CMPOPS = {ast.Lt: "lt", ast.LtE: "le", ast.Gt: "gt", ast.GtE: "ge", ast.Eq: "eq", ast.NotEq: "ne"}
# numpy function name -> (opcode, number of arguments)
FUNCS = {"sin": ("sin", 1), "cos": ("cos", 1), "tan": ("tan", 1), "arcsin": ("arcsin", 1), "arccos": ("arccos", 1),
         "arctan": ("arctan", 1), "arctan2": ("arctan2", 2), "atan2": ("arctan2", 2), "asin": ("arcsin", 1),
         "acos": ("arccos", 1), "atan": ("arctan", 1), "sqrt": ("sqrt", 1), "abs": ("abs", 1),
         "absolute": ("abs", 1), "fabs": ("abs", 1), "exp": ("exp", 1), "log": ("log", 1), "log2": ("log2", 1),
         "log10": ("log10", 1), "minimum": ("minimum", 2), "maximum": ("maximum", 2), "floor": ("floor", 1),
         "ceil": ("ceil", 1), "rint": ("rint", 1), "round": ("rint", 1), "sign": ("sign", 1), "square": ("square", 1),
         "hypot": ("hypot", 2), "where": ("where", 3), "clip": ("clip", 3), "power": ("pow", 2),
         "reciprocal": ("reciprocal", 1), "add": ("add", 2), "subtract": ("sub", 2), "multiply": ("mul", 2),
         "divide": ("div", 2), "negative": ("neg", 1), "mod": ("mod", 2), "floor_divide": ("floordiv", 2),
         "less": ("lt", 2), "greater": ("gt", 2)}
SCOPE = {"math": math, "numpy": numpy, "random": random, "re": re, "os": os}   # the reference's utils module scope
F32, F64, BOOL = 0, 1, 2


class Unsupported(NotImplementedError):
    pass


def _uses_arrays(node) -> bool:
    return any(isinstance(n, ast.Name) and n.id in ("r", "a") for n in ast.walk(node))


class Program:
    """One compiled expression: `code` is a list of (opcode, f64 flag, host-constant index or -1);
    `consts` the Python code objects of the host subtrees (evaluated per frame with t)."""

    def __init__(self, text: str):
        self.text = text
        tree = ast.parse(text.strip(), mode="eval").body
        self.code: list = []
        self.consts: list = []
        self.scalar_only = not _uses_arrays(tree)
        self._whole = compile(ast.Expression(tree), "<polar>", "eval")
        if not self.scalar_only:
            self.dtype = self._emit(tree)
            if len(self.code) > MAX_INSTR:
                raise Unsupported(f"polar expression too long for the device ({len(self.code)} > {MAX_INSTR} steps)")
            self._check_stack()

    # ---- compilation ------------------------------------------------------------------------
    def _const(self, node):
        """A host subtree: dtype is known only per frame; recorded as 'weak unless numpy says otherwise'."""
        self.consts.append(compile(ast.Expression(node), "<polar>", "eval"))
        self.code.append(["push_const", None, len(self.consts) - 1])
        return ("const", len(self.code) - 1)

    def _emit(self, node):
        if not _uses_arrays(node):
            return self._const(node)
        if isinstance(node, ast.Name):
            self.code.append(["push_r" if node.id == "r" else "push_a", F32, -1])
            return F32
        if isinstance(node, ast.BinOp) and type(node.op) in BINOPS:
            op = BINOPS[type(node.op)]
            lt = self._emit(node.left)
            if op == "pow" and not _uses_arrays(node.right):
                # numpy's scalar-exponent fast paths are chosen per frame in values(); keep generic pow here
                pass
            rt = self._emit(node.right)
            return self._push_op(op, lt, rt)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            t = self._emit(node.operand)
            if isinstance(node.op, ast.USub):
                return self._push_op("neg", t)
            return t
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Invert):
            t = self._emit(node.operand)
            if t != BOOL:
                raise Unsupported("~ on a non-boolean array")
            return self._push_op("not", t)
        if isinstance(node, ast.Compare) and len(node.ops) == 1 and type(node.ops[0]) in CMPOPS:
            lt = self._emit(node.left)
            rt = self._emit(node.comparators[0])
            self._push_op(CMPOPS[type(node.ops[0])], lt, rt)
            return BOOL
        if isinstance(node, ast.Call) and not node.keywords:
            fn = node.func
            name = None
            if isinstance(fn, ast.Attribute) and isinstance(fn.value, ast.Name) and fn.value.id in ("numpy", "np"):
                name = fn.attr
            elif isinstance(fn, ast.Name) and fn.id == "abs":
                name = "abs"
            if name in FUNCS and FUNCS[name][1] == len(node.args):
                types = [self._emit(arg) for arg in node.args]
                return self._push_op(FUNCS[name][0], *types)
        raise Unsupported(f"polar expression not supported on the device: {ast.unparse(node)!r}")

    def _push_op(self, op, *types):
        # result type: float64 only when a float64 operand is strong; constants resolve per frame
        self.code.append([op, ("types", types), -1])
        if op in ("lt", "le", "gt", "ge", "eq", "ne", "not"):
            return BOOL
        return ("op", len(self.code) - 1)

    def _check_stack(self):
        depth = peak = 0
        arity = {"where": 3, "clip": 3}
        for op, _, _ in self.code:
            if op.startswith("push"):
                depth += 1
            else:
                n = arity.get(op, 2 if op in ("add", "sub", "mul", "div", "pow", "mod", "floordiv", "arctan2", "minimum",
                                              "maximum", "hypot", "lt", "le", "gt", "ge", "eq", "ne") else 1)
                depth -= n - 1
            peak = max(peak, depth)
        if peak > MAX_STACK:
            raise Unsupported(f"polar expression needs a deeper stack than the device has ({peak} > {MAX_STACK})")

    # ---- per frame ------------------------------------------------------------------------------
    def host_value(self, t):
        """The whole expression on the host (only valid when it does not use r or a)."""
        return eval(self._whole, SCOPE, {"t": t})

    def resolve(self, t):
        """[(opcode, f64 flag, immediate)] for this frame: host constants evaluated, types propagated
        the way numpy does (weak Python scalars, strong numpy scalars)."""
        vals = [eval(c, SCOPE, {"t": t}) for c in self.consts]
        kinds = {}      # instruction index -> dtype of its result: F32 / F64 / BOOL / 'weak'
        out = []

        def kind_of(tag):
            if tag in (F32, F64, BOOL):
                return tag
            return kinds[tag[1]]

        for i, (op, info, ci) in enumerate(self.code):
            if op == "push_const":
                v = vals[ci]
                if isinstance(v, numpy.ndarray):
                    raise Unsupported("array-valued sub-expression that does not come from r or a")
                if isinstance(v, numpy.generic) and v.dtype == numpy.float64:
                    kinds[i] = F64
                elif isinstance(v, (bool, int, float, numpy.bool_, numpy.integer, numpy.floating)):
                    kinds[i] = F32 if isinstance(v, numpy.float32) else "weak"
                else:
                    raise Unsupported(f"polar sub-expression of type {type(v).__name__}")
                out.append((OP[op], 1, float(v)))       # immediates travel as double; rounded at use
                continue
            if op in ("push_r", "push_a"):
                kinds[i] = F32
                out.append((OP[op], 0, 0.0))
                continue
            ks = [kind_of(tg) for tg in info[1]]
            if op == "where":
                ks = ks[1:]                              # the condition does not take part in the promotion
            wide = any(k == F64 for k in ks)
            if all(k in ("weak", BOOL) for k in ks):
                wide = True                              # Python scalars among themselves: float64 (cannot happen: folded)
            kinds[i] = BOOL if op in ("lt", "le", "gt", "ge", "eq", "ne", "not") else (F64 if wide else F32)
            out.append((OP[op], int(wide), 0.0))
        out = self._scalar_power_fast_paths(out)
        return out, (kinds[len(self.code) - 1] if self.code else "weak")

    @staticmethod
    def _scalar_power_fast_paths(steps):
        """numpy evaluates array ** scalar through a dedicated ufunc for a few exponents
        (2 -> square, 0.5 -> sqrt, -1 -> reciprocal, 1 -> the array itself)."""
        res = []
        for op, wide, imm in steps:
            if op == OP["pow"] and res and res[-1][0] == OP["push_const"] and res[-1][2] in (2.0, 0.5, -1.0, 1.0):
                e = res.pop()[2]
                if e != 1.0:
                    res.append((OP[{2.0: "square", 0.5: "sqrt", -1.0: "reciprocal"}[e]], wide, 0.0))
                continue
            res.append((op, wide, imm))
        return res


class PolarFilter:
    """filters.py:75-87 with its two expressions compiled for the device."""

    def __init__(self, expr_radius: str, expr_theta: str):
        self.radius, self.theta = Program(expr_radius), Program(expr_theta)

    @staticmethod
    def _steps(prog: Program, t):
        if prog.scalar_only:
            v = prog.host_value(t)
            if isinstance(v, numpy.ndarray):
                raise Unsupported("array-valued polar expression that does not come from r or a")
            kind = F64 if (isinstance(v, numpy.generic) and v.dtype == numpy.float64) else (
                F32 if isinstance(v, numpy.float32) else "weak")
            return [(OP["push_const"], 1, float(v))], kind
        return prog.resolve(t)

    def programs(self, t):
        """(radius steps, theta steps, wide_trig, wide_product) for frame time t."""
        sr, kr = self._steps(self.radius, t)
        st, kt = self._steps(self.theta, t)
        wide_trig = kt != F32            # numpy.sin of a float64 -- or of a bare Python scalar -- is a float64
        wide_product = wide_trig or kr == F64
        return sr, st, wide_trig, wide_product
