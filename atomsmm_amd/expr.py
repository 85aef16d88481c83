"""Host compiler for CustomIntegrator expressions that are not a kick or a move.

The reference's thermostat / stochastic propagators (src/atomsmm/propagators.py:276-827, 1108-2172) call
`addComputePerDof(variable, expression)`, `addComputeSum(variable, expression)` and `addComputeGlobal(variable,
expression)` with OpenMM's expression syntax: `^` for powers, functions (sqrt, exp, log, sin, ..., step, delta,
select, min, max), and auxiliary definitions after ';' (`vscaling*v; vscaling = exp(-a*dt)`).  Per-DOF and sum
expressions are compiled here into the postfix program that `amm_expr_eval` interprets on the GPU (csrc/expr.hip);
global expressions are evaluated on the host.  `gaussian` / `uniform` are ONE draw per evaluation and degree of
freedom, as in OpenMM (every occurrence inside one expression sees the same value).
"""
import ast
import functools
import math
import re

OPCODES = dict(
    CONST=0, GLOBAL=1, BUF=2, MASS=3, GAUSS=4, UNIFORM=5, LOAD=6, STORE=7, DEVG=8, OUT=9,
    ADD=10, SUB=11, MUL=12, DIV=13, NEG=14, POW=15, POWI=16,
    sqrt=20, exp=21, log=22, sin=23, cos=24, tan=25, asin=26, acos=27, atan=28, sinh=29, cosh=30, tanh=31, erf=32,
    erfc=33, abs=34, floor=35, ceil=36, step=37, delta=38, min=39, max=40, select=41, atan2=42, HORNER=43)
_ARITY = {'min': 2, 'max': 2, 'select': 3, 'atan2': 2}
MAX_LOCALS = 16


class ExpressionError(ValueError):
    pass


class NeedsValue(Exception):
    """A deferred global was used where its number is needed (anything but sums and multiples)."""


class Deferred:
    """A global of a host-walked step program whose number waits on device results: const + sum_k coef[k] * slot[k],
    slot[k] being scalars that kernels already enqueued will leave in a device buffer (deriv(energy, lambda) of an AFED
    program, integrators.py:735-737: the extended variable's velocity collects them, nothing else reads it until lambda itself
    moves).  Sums and multiples keep the form, so the host does not wait for the GPU at every deriv(); any other use raises
    NeedsValue and the engine reads the buffer once."""
    __slots__ = ('const', 'terms')

    def __init__(self, const=0.0, terms=None):
        self.const, self.terms = float(const), dict(terms or {})

    def _combined(self, other, sign):
        if isinstance(other, Deferred):
            terms = dict(self.terms)
            for k, c in other.terms.items():
                terms[k] = terms.get(k, 0.0) + sign * c
            return Deferred(self.const + sign * other.const, terms)
        return Deferred(self.const + sign * float(other), self.terms)

    def _scaled(self, factor):
        if isinstance(factor, Deferred):
            raise NeedsValue()
        factor = float(factor)
        return Deferred(self.const * factor, {k: c * factor for k, c in self.terms.items()})

    def __add__(self, other):
        return self._combined(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        return self._combined(other, -1.0)

    def __rsub__(self, other):
        return self._scaled(-1.0)._combined(other, 1.0)

    def __neg__(self):
        return self._scaled(-1.0)

    def __pos__(self):
        return self

    def __mul__(self, other):
        return self._scaled(other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Deferred):
            raise NeedsValue()
        return self._scaled(1.0 / float(other))

    def _needs_value(self, *args):
        raise NeedsValue()

    __rtruediv__ = __pow__ = __rpow__ = __float__ = __abs__ = __bool__ = _needs_value
    __lt__ = __le__ = __gt__ = __ge__ = __eq__ = __ne__ = _needs_value
    __hash__ = None

    def resolve(self, values):
        return self.const + sum(c * float(values[k]) for k, c in self.terms.items())


def split_definitions(text):
    """'a*v; a = exp(-g*dt); g = 2' -> ('a*v', {'a': 'exp(-g*dt)', 'g': '2'})."""
    parts = [p.strip() for p in text.split(';') if p.strip()]
    if not parts:
        raise ExpressionError('empty expression')
    defs = {}
    for p in parts[1:]:
        if '=' not in p:
            raise ExpressionError('auxiliary definition without "=": ' + p)
        name, rhs = p.split('=', 1)
        defs[name.strip()] = rhs.strip()
    return parts[0], defs


_KEYWORD = re.compile(r'\b(lambda)\b')       # a legal OpenMM variable name (tests/test_systems.py:165), a Python keyword


def _name(node_id):
    return node_id[:-len('__kw')] if node_id.endswith('__kw') else node_id


@functools.lru_cache(maxsize=4096)
def _parse(text):
    """Syntax tree of one expression; cached (a host-walked step program evaluates the same few dozen texts every step;
    nothing that walks the tree modifies it)."""
    try:
        tree = ast.parse(_KEYWORD.sub(r'\1__kw', text).replace('^', '**'), mode='eval').body
        for node in ast.walk(tree):
            if isinstance(node, ast.Name):
                node.id = _name(node.id)
        return tree
    except SyntaxError as exc:
        raise ExpressionError('cannot parse expression %r: %s' % (text, exc))


class Program:
    """Postfix program for amm_expr_eval: `code` (opcode | arg << 8), `consts`, and the names of the global values the
    launch must supply, in order (`globals_`)."""

    def __init__(self):
        self.code, self.consts, self.globals_ = [], [], []

    def emit(self, op, arg=0):
        self.code.append(OPCODES[op] | (int(arg) << 8))

    def const(self, value):
        value = float(value)
        for k, c in enumerate(self.consts):
            if c == value and math.copysign(1.0, c) == math.copysign(1.0, value):
                return k
        self.consts.append(value)
        return len(self.consts) - 1

    def global_index(self, name):
        if name not in self.globals_:
            self.globals_.append(name)
        return self.globals_.index(name)


def compile_per_dof(text, resolve):
    """Compile a per-DOF (or sum) expression.  `resolve(name)` returns ('buf', slot), ('mass',), ('global',) or None."""
    main, defs = split_definitions(text)
    prog = Program()
    local_of, in_progress = {}, set()

    def gen(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            prog.emit('CONST', prog.const(node.value))
        elif isinstance(node, ast.Name):
            name = node.id
            if name in defs:
                if name not in local_of:
                    if name in in_progress:
                        raise ExpressionError('circular auxiliary definition: ' + name)
                    if len(local_of) + len(in_progress) >= MAX_LOCALS:
                        raise ExpressionError('too many auxiliary definitions')
                    in_progress.add(name)
                    gen(_parse(defs[name]))
                    in_progress.discard(name)
                    local_of[name] = len(local_of)
                    prog.emit('STORE', local_of[name])
                prog.emit('LOAD', local_of[name])
            elif name == 'gaussian':
                prog.emit('GAUSS')
            elif name in ('uniform', 'random'):
                prog.emit('UNIFORM')
            else:
                kind = resolve(name)
                if kind is None:
                    raise ExpressionError('unknown symbol in per-DOF expression: ' + name)
                if kind[0] == 'buf':
                    prog.emit('BUF', kind[1])
                elif kind[0] == 'mass':
                    prog.emit('MASS')
                else:
                    prog.emit('GLOBAL', prog.global_index(name))
        elif isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            gen(node.operand)
            if isinstance(node.op, ast.USub):
                prog.emit('NEG')
        elif isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div, ast.Pow)):
            if isinstance(node.op, ast.Pow):
                e = node.right
                neg = isinstance(e, ast.UnaryOp) and isinstance(e.op, ast.USub)
                ev = e.operand if neg else e
                if isinstance(ev, ast.Constant) and float(ev.value) == int(ev.value) and abs(int(ev.value)) < 1 << 20:
                    gen(node.left)
                    prog.emit('POWI', -int(ev.value) if neg else int(ev.value))
                    return
            gen(node.left)
            gen(node.right)
            prog.emit({ast.Add: 'ADD', ast.Sub: 'SUB', ast.Mult: 'MUL', ast.Div: 'DIV', ast.Pow: 'POW'}[type(node.op)])
        elif isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and not node.keywords:
            fn = node.func.id
            if fn not in OPCODES or fn.isupper() or len(node.args) != _ARITY.get(fn, 1):
                raise ExpressionError('unsupported function call: %s/%d' % (fn, len(node.args)))
            for a in node.args:
                gen(a)
            prog.emit(fn)
        else:
            raise ExpressionError('unsupported syntax in expression: ' + ast.dump(node))

    gen(_parse(main))
    return prog


def compile_polynomial(coefficients, scale, shift, operand):
    """Scalar program (amm_expr_eval_scalar) for sum_k coefficients[k] * u^k with u = scale * operand + shift: the first local holds u,
    one HORNER word per coefficient (top <- top * u + c)."""
    prog = Program()
    if isinstance(operand, Deferred):
        prog.emit('CONST', prog.const(operand.const))
        for k, coef in sorted(operand.terms.items()):
            prog.emit('DEVG', k)
            if coef != 1.0:
                prog.emit('CONST', prog.const(coef))
                prog.emit('MUL')
            prog.emit('ADD')
    else:
        prog.emit('CONST', prog.const(float(operand)))
    prog.emit('CONST', prog.const(scale))
    prog.emit('MUL')
    prog.emit('CONST', prog.const(shift))
    prog.emit('ADD')
    prog.emit('STORE', 0)
    prog.emit('CONST', prog.const(coefficients[-1]))
    for c in reversed(coefficients[:-1]):
        prog.consts.append(float(c))          # (no search for equal constants: their order is the polynomial's)
        prog.emit('HORNER', len(prog.consts) - 1)
    return prog


def compile_scalar(text, env, rng=None, predicate=None, keep=None):
    """Compile a GLOBAL expression (addComputeGlobal) whose operands include deferred values (Deferred: const + sum coef * scalar[k],
    the scalars living in a device buffer) into a postfix program for amm_expr_eval_scalar: numbers become constants, a Deferred its
    linear form over DEVG operands, `deriv(energy, p)` what env['__deriv__'] returns, `gaussian` / `uniform` one host draw each.
    `predicate` / `keep` (values as in env): the program's value is select(predicate, expression, keep) -- a step inside an
    if-block whose condition waits on the device."""
    main, defs = split_definitions(text)
    prog = Program()
    local_of, in_progress, draws = {}, set(), {}

    def value(v):
        if isinstance(v, Deferred):
            prog.emit('CONST', prog.const(v.const))
            for k, coef in sorted(v.terms.items()):
                prog.emit('DEVG', k)
                if coef != 1.0:
                    prog.emit('CONST', prog.const(coef))
                    prog.emit('MUL')
                prog.emit('ADD')
        else:
            prog.emit('CONST', prog.const(float(v)))

    def gen(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            prog.emit('CONST', prog.const(node.value))
        elif isinstance(node, ast.Name):
            name = node.id
            if name in defs:
                if name not in local_of:
                    if name in in_progress:
                        raise ExpressionError('circular auxiliary definition: ' + name)
                    if len(local_of) + len(in_progress) >= MAX_LOCALS:
                        raise ExpressionError('too many auxiliary definitions')
                    in_progress.add(name)
                    gen(_parse(defs[name]))
                    in_progress.discard(name)
                    local_of[name] = len(local_of)
                    prog.emit('STORE', local_of[name])
                prog.emit('LOAD', local_of[name])
            elif name in ('gaussian', 'uniform', 'random'):
                key = 'gaussian' if name == 'gaussian' else 'uniform'
                if key not in draws:
                    if rng is None:
                        raise ExpressionError('random numbers in a global expression need a generator')
                    draws[key] = float(rng.standard_normal()) if key == 'gaussian' else float(rng.random())
                prog.emit('CONST', prog.const(draws[key]))
            elif name in env and not callable(env[name]):
                value(env[name])
            else:
                raise ExpressionError('unknown symbol in global expression: ' + name)
        elif isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            gen(node.operand)
            if isinstance(node.op, ast.USub):
                prog.emit('NEG')
        elif isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div, ast.Pow)):
            if isinstance(node.op, ast.Pow):
                e = node.right
                neg = isinstance(e, ast.UnaryOp) and isinstance(e.op, ast.USub)
                ev = e.operand if neg else e
                if isinstance(ev, ast.Constant) and float(ev.value) == int(ev.value) and abs(int(ev.value)) < 1 << 20:
                    gen(node.left)
                    prog.emit('POWI', -int(ev.value) if neg else int(ev.value))
                    return
            gen(node.left)
            gen(node.right)
            prog.emit({ast.Add: 'ADD', ast.Sub: 'SUB', ast.Mult: 'MUL', ast.Div: 'DIV', ast.Pow: 'POW'}[type(node.op)])
        elif isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == 'deriv' and len(node.args) == 2 \
                and all(isinstance(a, ast.Name) for a in node.args):
            if '__deriv__' not in env:
                raise ExpressionError('deriv() is not available in this context')
            value(env['__deriv__'](node.args[0].id, node.args[1].id))
        elif isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and not node.keywords:
            fn = node.func.id
            if fn not in OPCODES or fn.isupper() or len(node.args) != _ARITY.get(fn, 1):
                raise ExpressionError('unsupported function call: %s/%d' % (fn, len(node.args)))
            for a in node.args:
                gen(a)
            prog.emit(fn)
        else:
            raise ExpressionError('unsupported syntax in expression: ' + ast.dump(node))

    if predicate is not None:
        value(predicate)
    gen(_parse(main))
    if predicate is not None:
        value(keep)
        prog.emit('select')
    return prog


_HOST_FUNCS = dict(sqrt=math.sqrt, exp=math.exp, log=math.log, sin=math.sin, cos=math.cos, tan=math.tan, asin=math.asin,
                   acos=math.acos, atan=math.atan, sinh=math.sinh, cosh=math.cosh, tanh=math.tanh, erf=math.erf,
                   erfc=math.erfc, abs=abs, floor=math.floor, ceil=math.ceil, atan2=math.atan2, min=min, max=max,
                   step=lambda x: 1.0 if x >= 0 else 0.0, delta=lambda x: 1.0 if x == 0 else 0.0,
                   select=lambda c, a, b: a if c != 0 else b)


def eval_global(text, env, rng=None):
    """Value of a global expression (addComputeGlobal) for the variable values in `env`; `rng` (numpy Generator)
    supplies `gaussian` / `uniform` (one draw of each per evaluation)."""
    main, defs = split_definitions(text)
    cache, draws = {}, {}

    def value_of(name):
        if name in defs:
            if name not in cache:
                cache[name] = ev(_parse(defs[name]))
            return cache[name]
        if name in ('gaussian', 'uniform', 'random'):
            key = 'gaussian' if name == 'gaussian' else 'uniform'
            if key not in draws:
                if rng is None:
                    raise ExpressionError('random numbers in a global expression need a generator')
                draws[key] = float(rng.standard_normal()) if key == 'gaussian' else float(rng.random())
            return draws[key]
        if name in env:
            return env[name] if isinstance(env[name], Deferred) else float(env[name])
        raise ExpressionError('unknown symbol in global expression: ' + name)

    def ev(node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return float(node.value)
        if isinstance(node, ast.Name):
            return value_of(node.id)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = ev(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.BinOp):
            a, b = ev(node.left), ev(node.right)
            if isinstance(node.op, ast.Add):
                return a + b
            if isinstance(node.op, ast.Sub):
                return a - b
            if isinstance(node.op, ast.Mult):
                return a * b
            if isinstance(node.op, ast.Div):
                return a / b
            if isinstance(node.op, ast.Pow):
                return a ** b
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == 'deriv' and len(node.args) == 2 \
                and all(isinstance(a, ast.Name) for a in node.args):
            # deriv(energy, parameter): supplied by the engine (force kernels), integrators.py:735
            if '__deriv__' not in env:
                raise ExpressionError('deriv() is not available in this context')
            value = env['__deriv__'](node.args[0].id, node.args[1].id)
            return value if isinstance(value, Deferred) else float(value)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _HOST_FUNCS and not node.keywords:
            return float(_HOST_FUNCS[node.func.id](*[ev(a) for a in node.args]))
        raise ExpressionError('unsupported syntax in expression: ' + ast.dump(node))

    return ev(_parse(main))


@functools.lru_cache(maxsize=4096)
def symbols(text):
    """Names referenced by an expression (auxiliary definitions expanded, their own names removed)."""
    main, defs = split_definitions(text)
    found = set()
    for part in [main] + list(defs.values()):
        for node in ast.walk(_parse(part)):
            if isinstance(node, ast.Name):
                found.add(node.id)
    return frozenset(found - set(defs) - set(_HOST_FUNCS))
