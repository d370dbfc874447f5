"""Minimal stand-in for `pint`, used ONLY by oracle/make_goldens.py in the build container.

TEST INFRASTRUCTURE - never imported by the product path.

The reference tags its Scenario defaults with units (marlpde/parameters.py:16-48) and then
strips them again with ``.magnitude`` (parameters.py:99).  Only the magnitudes matter, so
units here are inert tokens that are closed under ``* / **``.
"""


class _Unit:
    def __mul__(self, other):
        return self

    __rtruediv__ = __truediv__ = __pow__ = __mul__

    def __rmul__(self, other):
        if isinstance(other, (int, float)):
            return Quantity(other)
        return self

    def __rtruediv__(self, other):
        if isinstance(other, (int, float)):
            return Quantity(other)
        return self


class Quantity:
    def __init__(self, magnitude):
        self.magnitude = magnitude

    def __mul__(self, other):
        if isinstance(other, _Unit):
            return self
        return Quantity(self.magnitude * _mag(other))

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, _Unit):
            return self
        return Quantity(self.magnitude / _mag(other))

    def __repr__(self):
        return f"Quantity({self.magnitude!r})"


def _mag(x):
    return x.magnitude if isinstance(x, Quantity) else x


class UnitRegistry:
    Quantity = Quantity

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Unit()
