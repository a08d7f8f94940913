"""Generates the plain `get_<name>` / `set_<name>` pairs that NSoL's solver
classes expose (the parameter-study code of the reference drives solvers
reflectively through them, solver_parameter_study.py:175-182)."""


def add_accessors(cls, names, setters=True):
    for name in names:
        attr = "_" + name

        def getter(self, _a=attr):
            return getattr(self, _a)
        getter.__name__ = "get_" + name
        setattr(cls, "get_" + name, getter)
        if setters:
            def setter(self, value, _a=attr):
                setattr(self, _a, value)
            setter.__name__ = "set_" + name
            setattr(cls, "set_" + name, setter)
    return cls
