"""Scalar programs of a host-walked step program (atomsmm_amd/expr.py: compile_scalar, compile_polynomial -> amm_expr_eval_scalar):
the words the device interprets, run here by a few lines of Python and compared with the host's own evaluation of the same
expressions (expr.eval_global) -- CustomIntegrator.addComputeGlobal semantics, integrators.py:701-737 for the texts."""
import math

import numpy as np
import pytest

from atomsmm_amd import expr as X

OP = {v: k for k, v in X.OPCODES.items()}
FUN = dict(sqrt=math.sqrt, exp=math.exp, log=math.log, sin=math.sin, cos=math.cos, tan=math.tan, abs=abs, floor=math.floor, ceil=math.ceil,
           step=lambda x: 1.0 if x >= 0 else 0.0, delta=lambda x: 1.0 if x == 0 else 0.0, tanh=math.tanh, sinh=math.sinh, cosh=math.cosh,
           erf=math.erf, erfc=math.erfc, asin=math.asin, acos=math.acos, atan=math.atan)


def run(code, consts, scalars):
    """What csrc/expr.hip: k_expr_scalar does, word by word."""
    st, loc = [], {}
    for word in code:
        op, arg = OP[word & 0xff], word >> 8
        if op == 'CONST':
            st.append(consts[arg])
        elif op == 'DEVG':
            st.append(scalars[arg])
        elif op == 'OUT':
            scalars[arg] = st.pop()
        elif op == 'LOAD':
            st.append(loc[arg])
        elif op == 'STORE':
            loc[arg] = st.pop()
        elif op == 'HORNER':
            st.append(st.pop() * loc[0] + consts[arg])
        elif op in ('ADD', 'SUB', 'MUL', 'DIV', 'POW', 'min', 'max', 'atan2'):
            b, a = st.pop(), st.pop()
            st.append({'ADD': a + b, 'SUB': a - b, 'MUL': a * b, 'DIV': a / b if b else float('nan'), 'POW': a ** b if op == 'POW' else 0.0,
                       'min': min(a, b), 'max': max(a, b), 'atan2': math.atan2(a, b)}[op])
        elif op == 'NEG':
            st.append(-st.pop())
        elif op == 'POWI':
            st.append(st.pop() ** arg)
        elif op == 'select':
            no, yes, cond = st.pop(), st.pop(), st.pop()
            st.append(yes if cond != 0.0 else no)
        else:
            st.append(FUN[op](st.pop()))
    assert not st


def evaluate(prog, scalars, dst=99):
    scalars = dict(scalars)
    run(prog.code + [X.OPCODES['OUT'] | (dst << 8)], prog.consts, scalars)
    return scalars[dst]


SCALARS = {3: -12.5, 7: 0.25, 11: 4.0}
ENV = {'dt': 0.004, '_m': 50.0, '_kT': 2.5, '_Q': 0.02, 'lam': X.Deferred(0.6, {7: 0.5}), '_v': X.Deferred(0.05, {3: -1e-5, 11: 2e-5}),
       '_v_eta': 0.3}
NUMBERS = {k: (v.resolve(SCALARS) if isinstance(v, X.Deferred) else v) for k, v in ENV.items()}


@pytest.mark.parametrize('text', [
    'lam + 0.5*dt*_v',
    '_v_eta + 0.5*dt*(_m*_v^2-_kT)/_Q',
    '_v*exp(-dt*_v_eta)',
    'select(step(lam-(0)),2,0)-lam',
    'a*_v + b; a = exp(-dt*g); b = sqrt(1-a*a); g = 2',
    'delta(((step(lam-(0))*step(1-lam))-(0)))',
    '-_v',
])
def test_scalar_program_equals_the_hosts_value(text):
    prog = X.compile_scalar(text, ENV)
    assert evaluate(prog, SCALARS) == pytest.approx(X.eval_global(text, NUMBERS), rel=1e-15, abs=1e-300)
    assert all((w & 0xff) != X.OPCODES['GLOBAL'] for w in prog.code)             # numbers travel as constants, deferred values as DEVG words


def test_predicated_assignment():
    """A step inside an if-block whose condition waits on the device: target <- select(condition, expression, target)."""
    for cond, expect in ((1.0, 'new'), (0.0, 'old')):
        scalars = dict(SCALARS)
        scalars[20] = cond
        prog = X.compile_scalar('-_v', ENV, None, X.Deferred(0.0, {20: 1.0}), ENV['_v'])
        want = -NUMBERS['_v'] if expect == 'new' else NUMBERS['_v']
        assert evaluate(prog, scalars) == pytest.approx(want, rel=1e-15)


def test_deriv_and_random_draws_are_bound_at_compile_time():
    env = dict(ENV)
    env['__deriv__'] = lambda what, name: X.Deferred(1.5, {11: 1.0})
    prog = X.compile_scalar('_v - 0.5*(dt/4)*deriv(energy,lam)/_m + 0*gaussian', env, np.random.default_rng(3))
    want = NUMBERS['_v'] - 0.5 * (0.004 / 4) * (1.5 + SCALARS[11]) / 50.0
    assert evaluate(prog, SCALARS) == pytest.approx(want, rel=1e-15)
    with pytest.raises(X.ExpressionError):
        X.compile_scalar('gaussian', ENV)                       # no generator: no draw


def test_polynomial_program():
    """The long-range correction's lambda-derivative while lambda is a device scalar: monomials in u = 2 lambda - 1, one HORNER word each."""
    rng = np.random.default_rng(1)
    coef = list(rng.normal(size=9))
    prog = X.compile_polynomial(coef, 2.0, -1.0, ENV['lam'])
    u = 2.0 * NUMBERS['lam'] - 1.0
    assert evaluate(prog, SCALARS) == pytest.approx(sum(c * u ** k for k, c in enumerate(coef)), rel=1e-13)
    assert sum((w & 0xff) == X.OPCODES['HORNER'] for w in prog.code) == len(coef) - 1
