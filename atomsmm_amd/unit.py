"""Minimal stand-in for `simtk.unit` (OpenMM's unit package), enough for AtomsMM-style scripts.

The reference mixes `unit.Quantity` and plain numbers in its constructors (forces.py:446 requires a
Quantity, forces.py:145-147 documents "Number or unit.Quantity").  This module provides Quantity
arithmetic in OpenMM's MD unit system (nm, ps, dalton, e, K, radian, kJ/mol), so that
`10*unit.angstroms`, `0.29/unit.angstroms`, `q/q.unit`, `q.value_in_unit(unit.nanometers)` and
comparisons behave as scripts expect.  Dimensions: (length, time, mass, charge, temperature, angle).
"""
import math

_DIMS = ('length', 'time', 'mass', 'charge', 'temperature', 'angle')


class Unit:
    __array_priority__ = 100

    def __init__(self, scale, dims, name=None):
        self.scale = float(scale)              # value of 1 <this unit> in MD base units
        self.dims = tuple(dims)
        self.name = name

    def __repr__(self):
        return 'Unit(%s)' % (self.name or ('%g x %s' % (self.scale, dict(zip(_DIMS, self.dims)))))

    def get_name(self):
        return self.name or repr(self)

    def is_compatible(self, other):
        return self.dims == other.dims

    def conversion_factor_to(self, other):
        if self.dims != other.dims:
            raise TypeError('incompatible units: %r and %r' % (self, other))
        return self.scale / other.scale

    def is_dimensionless(self):
        return all(d == 0 for d in self.dims)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale * other.scale, [a + b for a, b in zip(self.dims, other.dims)],
                        _join(self.name, other.name, '*'))
        if isinstance(other, Quantity):
            return Quantity(other._value, self * other.unit)
        return Quantity(other, self)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return Unit(self.scale / other.scale, [a - b for a, b in zip(self.dims, other.dims)],
                        _join(self.name, other.name, '/'))
        if isinstance(other, Quantity):
            return Quantity(1.0 / other._value, self / other.unit)
        return Quantity(1.0 / other, self)

    def __rtruediv__(self, other):
        inv = Unit(1.0 / self.scale, [-a for a in self.dims], _join('1', self.name, '/'))
        if isinstance(other, Quantity):
            return Quantity(other._value, other.unit * inv)
        return Quantity(other, inv)

    def __pow__(self, p):
        return Unit(self.scale ** p, [a * p for a in self.dims], '%s**%s' % (self.name, p) if self.name else None)

    def __eq__(self, other):
        return isinstance(other, Unit) and self.dims == other.dims and math.isclose(self.scale, other.scale, rel_tol=1e-14)

    def __hash__(self):
        return hash(self.dims)


def _join(a, b, op):
    return '%s%s%s' % (a, op, b) if a and b else None


class Quantity:
    __array_priority__ = 100

    def __init__(self, value, unit=None):
        if unit is None:
            unit = dimensionless
        self._value = value
        self.unit = unit

    def __repr__(self):
        return 'Quantity(value=%r, unit=%s)' % (self._value, self.unit.get_name())

    __str__ = __repr__

    def value_in_unit(self, unit):
        f = self.unit.conversion_factor_to(unit)
        return self._value * f if f != 1.0 else self._value

    def in_units_of(self, unit):
        return Quantity(self.value_in_unit(unit), unit)

    def _md(self):
        """Value in MD base units."""
        return self._value * self.unit.scale if self.unit.scale != 1.0 else self._value

    def _coerce(self, other):
        if isinstance(other, Quantity):
            return other.value_in_unit(self.unit)
        if self.unit.is_dimensionless():
            return other / self.unit.scale
        raise TypeError('cannot combine %r with a plain number' % (self,))

    def __add__(self, other):
        return Quantity(self._value + self._coerce(other), self.unit)

    __radd__ = __add__

    def __sub__(self, other):
        return Quantity(self._value - self._coerce(other), self.unit)

    def __rsub__(self, other):
        return Quantity(self._coerce(other) - self._value, self.unit)

    def __neg__(self):
        return Quantity(-self._value, self.unit)

    def __abs__(self):
        return Quantity(abs(self._value), self.unit)

    def _reduce(self):
        if self.unit.is_dimensionless():
            return self._value * self.unit.scale
        return self

    def __mul__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value * other._value, self.unit * other.unit)._reduce()
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit * other)._reduce()
        return Quantity(self._value * other, self.unit)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value / other._value, self.unit / other.unit)._reduce()
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit / other)._reduce()
        return Quantity(self._value / other, self.unit)

    def __rtruediv__(self, other):
        return Quantity(other / self._value, dimensionless / self.unit)

    def __pow__(self, p):
        return Quantity(self._value ** p, self.unit ** p)

    def sqrt(self):
        return Quantity(math.sqrt(self._value), self.unit ** 0.5)

    def _cmp(self, other):
        return self._value, self._coerce(other)

    def __lt__(self, other):
        a, b = self._cmp(other); return a < b

    def __le__(self, other):
        a, b = self._cmp(other); return a <= b

    def __gt__(self, other):
        a, b = self._cmp(other); return a > b

    def __ge__(self, other):
        a, b = self._cmp(other); return a >= b

    def __eq__(self, other):
        try:
            a, b = self._cmp(other)
        except TypeError:
            return False
        return a == b

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash((self._md() if not hasattr(self._value, '__len__') else id(self)))

    def __float__(self):
        if self.unit.is_dimensionless():
            return float(self._value * self.unit.scale)
        raise TypeError('only dimensionless quantities convert to float')

    def __len__(self):
        return len(self._value)

    def __getitem__(self, k):
        return Quantity(self._value[k], self.unit)

    def __iter__(self):
        for v in self._value:
            yield Quantity(v, self.unit)


def is_quantity(x):
    return isinstance(x, Quantity)


def md_value(x, expected=None):
    """Plain float/array in MD units (nm, ps, dalton, e, K, rad, kJ/mol) from a Quantity or a number.
    `expected` (a Unit) is used only to check dimensions of Quantities."""
    if isinstance(x, Quantity):
        if expected is not None and x.unit.dims != expected.dims:
            raise TypeError('expected a quantity compatible with %s, got %s' % (expected.get_name(), x.unit.get_name()))
        return x._md()
    return x


def _u(scale, length=0, time=0, mass=0, charge=0, temperature=0, angle=0, name=None):
    return Unit(scale, (length, time, mass, charge, temperature, angle), name)


dimensionless = _u(1.0, name='dimensionless')
nanometer = nanometers = _u(1.0, length=1, name='nanometer')
angstrom = angstroms = _u(0.1, length=1, name='angstrom')
picometer = picometers = _u(1e-3, length=1, name='picometer')
meter = meters = _u(1e9, length=1, name='meter')
picosecond = picoseconds = _u(1.0, time=1, name='picosecond')
femtosecond = femtoseconds = _u(1e-3, time=1, name='femtosecond')
nanosecond = nanoseconds = _u(1e3, time=1, name='nanosecond')
second = seconds = _u(1e12, time=1, name='second')
dalton = daltons = amu = amus = _u(1.0, mass=1, name='dalton')
elementary_charge = elementary_charges = _u(1.0, charge=1, name='elementary charge')
kelvin = kelvins = _u(1.0, temperature=1, name='kelvin')
radian = radians = _u(1.0, angle=1, name='radian')
degree = degrees = _u(math.pi / 180.0, angle=1, name='degree')
mole = moles = _u(1.0, name='mole')                      # amounts are folded into per-mole energies
kilojoule_per_mole = kilojoules_per_mole = _u(1.0, length=2, time=-2, mass=1, name='kilojoule/mole')
kilocalorie_per_mole = kilocalories_per_mole = _u(4.184, length=2, time=-2, mass=1, name='kilocalorie/mole')
kilojoule = kilojoules = kilojoule_per_mole              # (per mole implied, as in MD unit systems)
# physical constants as in simtk.unit of the OpenMM 7.x the reference's literals come from (CODATA 2006: kB = 1.3806504e-23
# J/K, NA = 6.02214179e23 /mol): tests/test_computers.py:33-37, :70-74 are met to 1e-12 with these and to 8e-5 only
# with the 2018 values
_NA = 6.02214179e23
_KB = 1.3806504e-23
atmosphere = atmospheres = _u(1.01325e5 * 1e-27 * _NA * 1e-3, length=-1, time=-2, mass=1, name='atmosphere')
bar = bars = _u(1e5 * 1e-27 * _NA * 1e-3, length=-1, time=-2, mass=1, name='bar')

# kB*NA = R in kJ/mol/K ; the reference forms it as this product (utils.py:16)
BOLTZMANN_CONSTANT_kB = Quantity(_KB * _NA * 1e-3, kilojoule_per_mole / kelvin)
AVOGADRO_CONSTANT_NA = 1.0
MOLAR_GAS_CONSTANT_R = BOLTZMANN_CONSTANT_kB
