"""Iteration observer (API of nsol/observer.py:21-161).

The solvers hand it a host copy of every iterate (`add_x`), user-supplied
measure callables are evaluated afterwards over that history.  Attaching an
observer therefore costs one device-to-host copy per iteration; the fused
solvers step iteration by iteration instead of enqueueing the whole run."""
import numpy as np

from ._accessors import add_accessors


class Observer(object):

    def __init__(self, name="Observer"):
        self._name = name
        self._computational_time = None
        self._x_list = []
        self._functions = {}        # measure name -> callable(x) -> float
        self._values = {}           # measure name -> np.ndarray over history

    def add_x(self, x):
        self._x_list.append(x)

    def clear_x_list(self):
        del self._x_list[:]

    def set_measures(self, measures_dic):
        for key, fn in measures_dic.items():
            self._functions[key] = fn
            self._values.setdefault(key, None)

    def get_measures(self):
        return self._values

    def compute_measures(self):
        history = self._x_list
        for key, fn in self._functions.items():
            self._values[key] = np.fromiter((fn(x) for x in history),
                                            dtype=float, count=len(history))


add_accessors(Observer, ["name", "computational_time"])
add_accessors(Observer, ["x_list"], setters=False)
